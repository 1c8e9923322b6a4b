timeout 300 python bench.py --no-cpu-baseline --no-secondary --steps 6 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step']); g=d['roofline']['glue_kernels_hbm']
for k in ('time_attn64_kernel','time_kv_append_kernel','pool_mix_kernel','small_attn_kernel','space_attn_kernel','assemble_kernel'): print(k, g[k]['frac_of_8tbs'], g[k]['avg_us'], g[k]['launches'])"
