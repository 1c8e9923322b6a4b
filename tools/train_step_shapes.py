"""Per-shape GEMM table of ONE dynamics training step at cfg 2's architecture (forward + backward, flow loss only), plus host-vs-device time.
Run with D4_GEMM_LOG=1 (the table goes to stderr).   python tools/train_step_shapes.py [B] [T]"""
import ctypes as C, sys, time
sys.path.insert(0, __file__.rsplit('/', 2)[0])
import torch
from dreamer4_amd import DynamicsWorldModel, _lib
from dreamer4_amd.synthetic import randomize_weights

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
T = int(sys.argv[2]) if len(sys.argv) > 2 else 16
torch.manual_seed(0)
m = randomize_weights(DynamicsWorldModel(dim=512, dim_latent=32, num_latent_tokens=32, depth=6, num_discrete_actions=4)).cuda()
g = torch.Generator(device='cuda').manual_seed(1)
lat = torch.randn(B, T, 32, 32, device='cuda', generator=g).clamp(-2, 2)
acts = torch.randint(0, 4, (B, T, 1), device='cuda', generator=g)
params = list(m.parameters())
lib = _lib.load()


def step():
    for p in params:
        p.grad = None
    m(latents=lat, discrete_actions=acts, generator=g, prob_shortcut_train=0.).backward()


for _ in range(3):
    step()
torch.cuda.synchronize()
n = 5
t0 = time.perf_counter()
for _ in range(n):
    step()
t1 = time.perf_counter()
torch.cuda.synchronize(); t2 = time.perf_counter()
print(f'flow-only step: host enqueue {1e3 * (t1 - t0) / n:.2f} ms, with sync {1e3 * (t2 - t0) / n:.2f} ms')
ncls = lib.d4_profile_classes()
lib.d4_profile_enable((1 << ncls) - 1)
step()
torch.cuda.synchronize()
lib.d4_profile_enable(0)
ms = (C.c_double * ncls)(); fl = (C.c_double * ncls)(); cnt = (C.c_int64 * ncls)()
_lib.check(lib.d4_profile_read(ms, fl, cnt, ncls))
tot_ms, tot_fl = sum(ms), sum(fl)
print(f'GEMM launches {sum(cnt)}, GEMM time {tot_ms:.2f} ms, {tot_fl / 1e12:.3f} TFLOP -> {tot_fl / tot_ms / 1e9:.1f} TF/s inside the GEMMs')
