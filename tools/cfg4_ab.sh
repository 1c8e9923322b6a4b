#!/bin/bash
# same-box A/B of the cfg-4 decode fusions: bash tools/cfg4_ab.sh
cd $GRAFT_REPO_ROOT
run() { env "$@" python -c "
import torch, bench
print(bench.cfg4_env_latency(torch.device('cuda', 0)))" 2>/dev/null | tail -1 | cut -c1-150; }
for i in 1 2; do
echo "default            : $(run X=1)"
echo "no attn_out_cols   : $(run D4_ATTN_OUT_COLS=0)"
echo "no fused append    : $(run D4_TIME_ATTN_FUSED_APPEND=0)"
echo "neither            : $(run D4_ATTN_OUT_COLS=0 D4_TIME_ATTN_FUSED_APPEND=0)"
done
