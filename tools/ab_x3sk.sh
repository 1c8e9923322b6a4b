run() { echo "== $1"; env $2 timeout 400 python bench.py --no-cpu-baseline --no-secondary --steps 10 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], 'steps/s', d['ms_per_step'], 'ms')"; }
run "HEAD (every split-operand call persistent)" "D4_NOP=1"
run "half-tile rule only" "D4_GEMM_X3SK=1"
run "x3sk off" "D4_GEMM_X3SK=0"
run "HEAD (every split-operand call persistent)" "D4_NOP=1"
run "half-tile rule only" "D4_GEMM_X3SK=1"
run "x3sk off" "D4_GEMM_X3SK=0"
