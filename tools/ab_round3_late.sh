#!/bin/bash
# Same-box A/B of the late round-3 changes: each line = imagined steps/s and ms per step of `bench.py --steps 10 --warmup 3` (no CPU baseline, no secondary)
run() { echo -n "$1: "; env $2 timeout 400 python bench.py --no-cpu-baseline --no-secondary --steps 10 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], 'steps/s', d['ms_per_step'], 'ms;  rollout', d['generate_ms'], 'ms')"; }
run "HEAD                                          " "D4_NOP=1"
run "split-operand persistent form off             " "D4_GEMM_X3SK=0"
run "four-heads-per-wave time-layer kernels off    " "D4_TIME_ATTN_FEW=0 D4_KV_APPEND_LEGACY=1"
run "block-per-row pool mix only up to 64 rows     " "D4_POOL_MIX_ROWS_MAX=64"
run "all three off                                 " "D4_GEMM_X3SK=0 D4_TIME_ATTN_FEW=0 D4_KV_APPEND_LEGACY=1 D4_POOL_MIX_ROWS_MAX=64"
run "HEAD again                                    " "D4_NOP=1"
