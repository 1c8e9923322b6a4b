"""GPU experiment: the cfg-2 rollout as ONE B=256 call vs TWO concurrent B=128 calls on two HIP streams (two engines, same weights).
Question: do the launch ramps / tails of one half-batch hide behind the other's matrix work?  Median of 5 after warm passes."""
import sys, time; sys.path.insert(0, '/root/repo')
import torch
from dreamer4_amd import DynamicsWorldModel
from dreamer4_amd.synthetic import randomize_weights

CFG = dict(dim=512, dim_latent=32, num_latent_tokens=32, depth=6, num_discrete_actions=4)
if len(sys.argv) > 2 and sys.argv[2] == 'cfg5':       # BASELINE config 5 on the bf16 path, B = 128
    CFG = dict(dim=1024, dim_latent=32, num_latent_tokens=64, depth=12, num_continuous_actions=6, matmul_dtype='bf16')


def model():
    torch.manual_seed(0)
    return randomize_weights(DynamicsWorldModel(**CFG), terminal_bias=-10.).cuda()


def med(f, n=5):
    ts = []
    for _ in range(n):
        torch.cuda.synchronize(); t0 = time.perf_counter(); f(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return 1e3 * sorted(ts)[n // 2]


parts = int(sys.argv[1]) if len(sys.argv) > 1 else 2
B = 128 if 'matmul_dtype' in CFG else 256
one = model()
g = torch.Generator(device='cuda').manual_seed(1234)
kw = dict(return_for_policy_optimization=True)
for _ in range(2): one.generate(16, batch_size=B, generator=g, **kw)
print(f'one call  B={B}: {med(lambda: one.generate(16, batch_size=B, generator=g, **kw)):.2f} ms')

ms = [model() for _ in range(parts)]
ss = [torch.cuda.Stream() for _ in range(parts)]
gs = [torch.Generator(device='cuda').manual_seed(1234 + i) for i in range(parts)]


import threading


def split():                                   # generate() ends in a host sync (terminals.all()), so each half runs in its own thread
    def run(m_, s_, g_):
        with torch.cuda.stream(s_):
            m_.generate(16, batch_size=B // parts, generator=g_, **kw)
    th = [threading.Thread(target=run, args=a) for a in zip(ms, ss, gs)]
    for t in th: t.start()
    for t in th: t.join()


for _ in range(2): split()
print(f'{parts} streams B={B // parts} each: {med(split):.2f} ms')
# sequential on one stream for reference (what the split costs without overlap)
def seq():
    for m_, g_ in zip(ms, gs): m_.generate(16, batch_size=B // parts, generator=g_, **kw)
for _ in range(1): seq()
print(f'{parts} sequential B={B // parts}: {med(seq):.2f} ms')
