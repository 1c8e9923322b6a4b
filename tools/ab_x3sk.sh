run() { echo "== $1"; env $2 timeout 400 python bench.py --no-cpu-baseline --no-secondary --steps 10 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], 'steps/s', d['ms_per_step'], 'ms')
for k,v in d['roofline']['all_gemm_configs_one_warmup_step'].items():
    if 'x3sk' in k or '<2, 2, 1, 1,' in k: print('   ', k[:60], v)"; }
run "HEAD (rule)" "D4_NOP=1"
run "+ SiLU-GLU output projection as half tiles" "D4_GEMM_X3SK=3"
run "x3sk off" "D4_GEMM_X3SK=0"
run "HEAD (rule)" "D4_NOP=1"
run "+ SiLU-GLU output projection as half tiles" "D4_GEMM_X3SK=3"
run "x3sk off" "D4_GEMM_X3SK=0"
