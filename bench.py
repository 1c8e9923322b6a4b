#!/usr/bin/env python
"""Headline benchmark of the imagination hot path (BASELINE.json): imagined latent steps / second
(whole job) + actor/critic step ms, dim=512 depth=6 H=15, on N MI355X of one node.

    python bench.py [--gpus N] [--steps K] [--warmup W]            (N = 1)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one DreamTrainer iteration (dreamer4/trainers.py:1422-1452) on synthetic inputs:
generate(H+1 frames, B=256 trajectories per GPU, 4 denoising steps + 1 clean step per frame)
-> learn_from_experience(ppo) -> clip+AdamW on the policy head, then the value head.
Weights: default init of that architecture under torch.manual_seed(0) with non-trivial head weights
(SURVEY.md 8d); rollout noise from generator seed 1234 + rank.  Everything is resident in HBM before
the timed region.  Rank 0 prints ONE JSON line.

  value        = N * B * (H+1) * K / wall     imagined steps per second over the WHOLE step (rollout + learner)
  roofline     = the dominant kernel (the GEMM class with the largest share of GPU time, found in an extra event-timed step before the
                 timed region): algorithmic flops of its launches / their summed HIP-event durations in an event-timed REPEAT of the timed
                 steps right after the timed region (events cannot ride inside replayed hipGraphs; the timed region is the default path),
                 against the 157.3 TFLOP/s fp32 matrix peak (2500 / 6 for a split-operand class)
  glue_kernels_hbm = the non-GEMM kernel classes of the rollout (attention cores, KV append, pool mix, ...): ALGORITHMIC bytes of
                 their launches in the extra event-timed step / their HIP-event durations, against the 8 TB/s HBM peak — measured live
                 in this process (d4_profile_glue_*), not replayed from a file
  cpu_baseline = the CPU oracle (oracle/restate.py, torch fp32) running the FULL workload once on 16 host threads, rank 0 at N=1
                 only; cpu_baseline_sharded = the same workload sharded by trajectory over 16 processes x 4 threads (the best sharded form measured; the fair
                 "all host cores" form: the path is thousands of small ops, one process cannot use 256 threads).  Reported
                 baselines, not the target.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')      # dmabuf IPC: RCCL between processes fails with hipIpcGetMemHandle otherwise (must precede HIP init)

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CFG2 = dict(dim=512, dim_latent=32, num_latent_tokens=32, depth=6, attn_heads=8, attn_dim_head=64,
            num_spatial_tokens=4, num_register_tokens=8, max_steps=64, multi_token_pred_len=8, num_discrete_actions=4)
B_LOCAL, HORIZON, NUM_STEPS = 256, 15, 4
PEAK_FP32_MFMA_TFLOPS = 157.3          # /opt/skills/guides/MI355X_MICROARCH.md "Peak FP32 (matrix)"
EVENT_STRIDE = 7                       # roofline pass: every 7th launch of the dominant GEMM configuration carries an event pair (7 is co-prime to the ~30-launch per-evaluation pattern, so every shape is sampled)
FLOP_PER_IMAGINED_STEP = 5.23e9        # GEMM flops per generated frame of one trajectory (cfg 2)


def build_model(device):
    from dreamer4_amd import DynamicsWorldModel
    from dreamer4_amd.synthetic import randomize_weights
    torch.manual_seed(0)
    m = DynamicsWorldModel(**CFG2)
    randomize_weights(m, seed=0, terminal_bias=-10.)     # heads non-trivial; terminal head ~never fires so all H+1 frames count
    return m.to(device)


def cpu_baseline(max_threads=16, full_budget_s=150.0, keep=None):
    """The CPU oracle (oracle/restate.py, torch fp32) on the host cores, rank 0 at N = 1 only: the SAME workload as the GPU step
    (B = 256 trajectories, H + 1 = 16 frames, 4 + 1 evaluations per frame, + learn_from_experience(ppo) with its backward) run
    ONCE in full when a short probe projects it to fit `full_budget_s`; otherwise a bounded sample of the same architecture /
    call shape (flagged in `sample`).  torch fp32 on the host: the path is thousands of small ops, so more than ~16 intra-op
    threads only adds synchronisation cost (256 threads ran it 400x slower than 8 on the MI355X host) — `cores` is the thread
    count actually used, `host_cores` what the box has."""
    from dreamer4_amd import DynamicsWorldModel
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    from util import make_noise, oracle_config, oracle_weights
    from dreamer4_amd.synthetic import randomize_weights
    from oracle import restate
    host = os.cpu_count() or 1
    cores = min(host, max_threads)
    torch.set_num_threads(cores)
    torch.manual_seed(0)
    m = randomize_weights(DynamicsWorldModel(**CFG2), seed=0, terminal_bias=-10.)
    cfg, W = oracle_config(m), oracle_weights(m)
    heads = ('policy_head', 'value_head', 'action_embedder.discrete_action_unembed')

    nz_full = make_noise(cfg, HORIZON + 1, B_LOCAL, 1234)         # the draws the GPU rollout of `parity_at_headline` was given (trajectory-major slices)

    def run(batch, frames):
        nz = {k: v[:frames, :batch].contiguous() for k, v in nz_full.items()}
        t0 = time.perf_counter()
        with torch.no_grad():
            exp = restate.generate(cfg, W, frames, batch_size=batch, noise=nz, num_steps=NUM_STEPS)
        Wg = {k: (v.clone().requires_grad_() if k.startswith(heads) else v) for k, v in W.items()}
        pl, vl = restate.learn_losses(cfg, Wg, exp, 'ppo')
        pl.backward(); vl.backward()
        if keep is not None and frames == HORIZON + 1:
            keep.update(exp=exp, noise=nz, cfg=cfg, batch=batch)
        return batch * exp['latents'].shape[1], time.perf_counter() - t0

    run(1, 1)                                           # warm-up (thread pool, allocator)
    steps, dt = run(16, 3)                              # probe: projects the full workload's cost
    full = B_LOCAL * (HORIZON + 1)
    projected = dt / steps * full
    if projected <= full_budget_s:
        steps, dt = run(B_LOCAL, HORIZON + 1)
        what = f'the full workload once: generate(B={B_LOCAL}, frames={HORIZON + 1}, num_steps={NUM_STEPS}) + learn(ppo)'
    else:
        batch = int(max(16, min(B_LOCAL, 30. / max(projected, 1e-3) * B_LOCAL)))
        steps, dt = run(batch, HORIZON + 1)
        what = (f'BOUNDED SAMPLE (the full workload projected to {projected:.0f} s): generate(B={batch}, frames={HORIZON + 1}, '
                f'num_steps={NUM_STEPS}) + learn(ppo)')
    return dict(value=steps / dt, unit='imagined steps/s', cores=cores, host_cores=host, kind='port',
                sample=f'oracle/restate.py, {what} at dim=512 depth=6: {steps} imagined steps in {dt:.1f} s, torch fp32, {cores} of {host} host threads')


def _cpu_shard_worker(args):
    """One trajectory shard of the CPU baseline (spawned process): generate(B_shard, H+1) + learn(ppo) with `threads` intra-op threads."""
    shard, batch, threads, frames = args
    import torch as _t
    _t.set_num_threads(threads)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    from util import make_noise, oracle_config, oracle_weights
    from dreamer4_amd import DynamicsWorldModel
    from dreamer4_amd.synthetic import randomize_weights
    from oracle import restate
    _t.manual_seed(0)
    m = randomize_weights(DynamicsWorldModel(**CFG2), seed=0, terminal_bias=-10.)
    cfg, W = oracle_config(m), oracle_weights(m)
    heads = ('policy_head', 'value_head', 'action_embedder.discrete_action_unembed')
    with _t.no_grad():                                   # warm-up (thread pool, allocator), untimed
        restate.generate(cfg, W, 1, batch_size=1, noise=make_noise(cfg, 1, 1, 1), num_steps=NUM_STEPS)
    nz = make_noise(cfg, frames, batch, 1234 + shard)
    t0 = time.perf_counter()
    with _t.no_grad():
        exp = restate.generate(cfg, W, frames, batch_size=batch, noise=nz, num_steps=NUM_STEPS)
    Wg = {k: (v.clone().requires_grad_() if k.startswith(heads) else v) for k, v in W.items()}
    pl, vl = restate.learn_losses(cfg, Wg, exp, 'ppo')
    pl.backward(); vl.backward()
    return batch * exp['latents'].shape[1], time.perf_counter() - t0


def cpu_baseline_sharded(procs=16, threads=4):
    """The same full workload (B = 256, H + 1 = 16 frames + learn) sharded by dream trajectory over `procs` processes x `threads`
    intra-op threads each (the path shards by trajectory, north_star; measured on the MI355X host, tools/cpu_shard_tune.py: 16 x 4 -> 128
    steps/s, 32 x 4 -> 86, 32 x 8 -> 33, 16 x 16 -> 29: small per-shard matrices do not feed more threads).  Wall time = the slowest shard, model construction excluded
    (each worker times its own generate + learn).  Per-shard advantage statistics (no cross-process reduce): baseline only."""
    import multiprocessing as mp
    host = os.cpu_count() or 1
    procs = max(1, min(procs, host // max(threads, 1)) or 1)
    per = B_LOCAL // procs
    ctx = mp.get_context('spawn')
    with ctx.Pool(procs) as pool:
        res = pool.map(_cpu_shard_worker, [(i, per, threads, HORIZON + 1) for i in range(procs)])
    steps = sum(r[0] for r in res)
    dt = max(r[1] for r in res)
    return dict(value=steps / dt, unit='imagined steps/s', cores=procs * threads, host_cores=host, kind='port',
                sample=f'oracle/restate.py, the full workload sharded by trajectory: {procs} processes x {threads} threads, each generate(B={per}, '
                       f'frames={HORIZON + 1}, num_steps={NUM_STEPS}) + learn(ppo): {steps} imagined steps, slowest shard {dt:.1f} s')


CFG4 = dict(dim=512, dim_latent=16, num_latent_tokens=4, num_spatial_tokens=4, depth=6, num_discrete_actions=4)
CFG5 = dict(dim=1024, dim_latent=32, num_latent_tokens=64, depth=12, num_continuous_actions=6)
PEAK_BF16_MFMA_TFLOPS = 2500.          # MI355X_MICROARCH.md "Peak BF16/FP16 MFMA" (dense)
FLOP_PER_IMAGINED_STEP_CFG5 = 33.9e9   # cfg 5, per generated frame of one trajectory


def measured_peaks(device, lib):
    """What THIS box sustains (csrc/peaks.hip; outside every timed region, < 1 s): float4 stream copy over 2 x 512 MiB (read + write GB/s) and bare MFMA streams
    on random operands.  Reported BESIDE the datasheet peaks every `frac` is priced against (8000 GB/s, 157.3 / 2500 TFLOP/s), never instead of them."""
    from dreamer4_amd import _lib
    buf = torch.empty(1 << 30, dtype=torch.uint8, device=device)
    hbm, f32, b16 = C.c_double(), C.c_double(), C.c_double()
    _lib.check(lib.d4_measure_peaks(_lib.ptr(buf), buf.numel(), C.byref(hbm), C.byref(f32), C.byref(b16), C.c_void_p(torch.cuda.current_stream(device).cuda_stream)))
    del buf
    torch.cuda.empty_cache()
    return dict(hbm_stream_copy_gbs=round(hbm.value, 1), mfma_f32_tflops=round(f32.value, 1), mfma_bf16_tflops=round(b16.value, 1),
                what='float4 stream copy through 2 x 512 MiB (read + write bytes / s); v_mfma_f32_16x16x4_f32 and v_mfma_f32_32x32x16_bf16 streams, 12 independent '
                     'accumulators, 2 waves per SIMD, random operands; datasheet: 8000 GB/s, 157.3 / 2500 TFLOP/s')


def cfg4_env_latency(device, horizon=50):
    """BASELINE config 4, secondary numbers (driver-timed because they are part of this process): dim 512 depth 6, 4 x 16 latents,
    4 discrete user-chosen actions, one generated frame per call with the KV cache carried, exactly the call
    DynamicsWorldModelWrapper.step makes (dreamer4/env.py:445-483).  ms per env step at B = 1 and B = 16 (second of two passes)."""
    from dreamer4_amd import DynamicsWorldModel
    from dreamer4_amd.synthetic import randomize_weights
    torch.manual_seed(0)
    m = randomize_weights(DynamicsWorldModel(**CFG4), terminal_bias=-10.).to(device)
    out = {}
    for B in (1, 16):
        g = torch.Generator(device=device).manual_seed(1)
        acts_all = torch.randint(0, 4, (B, horizon, 1), device=device, generator=g)
        for rep in range(2):
            lat = torch.zeros(B, 0, 4, 16, device=device); rew = torch.zeros(B, 0, device=device); tc = None
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for t in range(horizon):
                kw = dict(prompt_latents=lat, prompt_discrete_actions=acts_all[:, :t], prompt_rewards=rew) if t > 0 else {}
                e, tc = m.generate(t + 1, batch_size=B, return_rewards_per_frame=True, return_terminals=True, time_cache=tc,
                                   return_time_cache=True, generator=g, **kw)
                lat, rew = e.latents, e.rewards
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
        out[f'b{B}_ms_per_env_step'] = round(1e3 * dt / horizon, 3)
        out[f'b{B}_steps_per_sec'] = round(B * horizon / dt, 1)
    out['workload'] = f'cfg4: dim=512 depth=6 latents=4x16, horizon {horizon}, one generate() call per env step (prompt + carried time cache), rewards + terminals'
    return out


def train_flow_step(device, B=16, T=16, reps=4):
    """SURVEY.md 8(f-3), secondary numbers: one dynamics TRAINING step at config 2's architecture — DynamicsWorldModel.forward without
    signal levels (flow + shortcut losses, D4:6956-7003, 7335-7431) + backward through the HIP trunk blocks (dreamer4_amd/trunk_ops.py),
    B x T frames of 15 tokens.  The blocks keep their forward workspace (nothing is recomputed in the backward)."""
    from dreamer4_amd import DynamicsWorldModel
    from dreamer4_amd.synthetic import randomize_weights
    torch.manual_seed(0)
    m = randomize_weights(DynamicsWorldModel(**CFG2)).to(device)
    g = torch.Generator(device=device).manual_seed(1)
    lat = torch.randn(B, T, CFG2['num_latent_tokens'], CFG2['dim_latent'], device=device, generator=g).clamp(-2, 2)
    acts = torch.randint(0, 4, (B, T, 1), device=device, generator=g)
    out = {}
    for name, prob in (('flow_only', 0.), ('with_shortcut', 1.)):
        for rep in range(2):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(reps):
                for p in m.parameters():
                    p.grad = None
                m(latents=lat, discrete_actions=acts, generator=g, prob_shortcut_train=prob).backward()
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
        out[f'{name}_ms_per_step'] = round(1e3 * dt, 2)
        out[f'{name}_frames_per_sec'] = round(B * T / dt, 1)
    # GEMM work of one flow-only step: forward + dX + dW = 3 x the forward's 2MNK (66.5 MFLOP per token, SURVEY.md 8d, + the learned-query
    # pools; the blocks keep their forward workspace, so nothing is recomputed: round 2 executed 4 x) -> a lower bound on the executed flops,
    # against the fp32 matrix peak
    fwd_flop = 66.5e6 * B * T * 15 + 11.5e9 * (B * T) / 256.
    out['flow_only_gemm_tflops'] = round(3. * fwd_flop / (out['flow_only_ms_per_step'] * 1e-3) / 1e12, 1)
    out['flow_only_frac_of_fp32_matrix_peak'] = round(out['flow_only_gemm_tflops'] / PEAK_FP32_MFMA_TFLOPS, 3)
    out['workload'] = f'cfg2 architecture, training forward + backward, B={B} x T={T} frames x 15 tokens = {B * T * 15} token rows, fp32'

    return out


def cfg5_bf16(device, lib, B=128, frames=16, reps=2):
    """BASELINE config 5, secondary numbers: dim 1024 depth 12, 64 x 32 latents, 6 continuous (Beta) actions, B = 128 per GPU
    (1024 / 8), H = 15, trunk GEMMs on the bf16 MFMA path.  Both halves of BASELINE.json's metric: the rollout (imagined steps/s) and the
    actor/critic step (learn_from_experience(ppo) on the Beta head + both backward passes + clip/AdamW on both heads; fp32, as at config 2)."""
    from dreamer4_amd import DreamTrainer, DynamicsWorldModel, _lib
    from dreamer4_amd.synthetic import randomize_weights
    torch.manual_seed(0)
    m = randomize_weights(DynamicsWorldModel(**CFG5, matmul_dtype='bf16'), terminal_bias=-10.).to(device)
    g = torch.Generator(device=device).manual_seed(1234)
    gk = dict(return_for_policy_optimization=True, num_steps=NUM_STEPS, generator=g)
    m.generate(frames, batch_size=B, **gk)
    torch.cuda.synchronize()
    lib.d4_profile_bf16_enable(5)
    t0 = time.perf_counter()
    for _ in range(reps):
        e = m.generate(frames, batch_size=B, **gk)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    lib.d4_profile_bf16_enable(0)
    ms, fl, cnt = C.c_double(), C.c_double(), C.c_int64()
    _lib.check(lib.d4_profile_bf16_read(C.byref(ms), C.byref(fl), C.byref(cnt)))
    steps = B * e.latents.shape[1]
    ach = fl.value / max(ms.value, 1e-9) / 1e9
    # the second half of the metric: actor/critic step on this rollout (trainers.py:1430-1452)
    tr = DreamTrainer(m, batch_size=B, generate_timesteps=frames - 1, objective='ppo')
    tr.learn(e)                                            # warm-up: learner workspace, tile choices of the 2048-row head GEMMs
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(3):
        tr.learn(e)
    torch.cuda.synchronize()
    learn_ms = 1e3 * (time.perf_counter() - t1) / 3
    return dict(value=round(steps / dt, 1), unit='imagined steps/s', ms_per_rollout=round(1e3 * dt, 2), actor_critic_step_ms=round(learn_ms, 2),
                whole_step_steps_per_sec=round(steps / (dt + 1e-3 * learn_ms), 1),
                dtype='bf16 MFMA (fp32 accumulate / norms / softmax); learner fp32',
                workload=f'cfg5: dim=1024 depth=12 latents=64x32, 6 continuous actions, B={B}, H={frames - 1}, num_steps={NUM_STEPS}; rollout, then learn_from_experience(ppo) + clip/AdamW both heads',
                rollout_algorithmic_tflops=round(FLOP_PER_IMAGINED_STEP_CFG5 * steps / dt / 1e12, 1),
                roofline=dict(bound='mfma', kernel='gemm_bf16a_kernel / gemm_bf16_kernel (all bf16 trunk GEMMs: bf16 activation images by LDS-DMA where the activation has one)', achieved=round(ach, 1), peak=PEAK_BF16_MFMA_TFLOPS, unit='TFLOP/s',
                              frac=round(ach / PEAK_BF16_MFMA_TFLOPS, 4), launches_timed=int(cnt.value), event_stride=5,
                              avg_launch_us=round(1e3 * ms.value / max(cnt.value, 1), 2)))


def cfg2_fp16x2_mode(device, reps=3):
    """Secondary: the headline rollout (cfg 2, B = 256, 16 frames) in the OPT-IN matmul_dtype='fp32_fp16x2' mode (csrc/gemm_h2.hip: trunk GEMMs of >= 1.2 G
    multiply-adds on the fp16 matrix cores — two fp16 planes per operand under exact row scales, three products, fp32 accumulate).  NOT what `value` is
    measured on: the scheme fails the sharpest of the three fp32 criteria the default path is held to (profiles/r05_x3_products.txt)."""
    from dreamer4_amd import DynamicsWorldModel
    from dreamer4_amd.synthetic import randomize_weights
    torch.manual_seed(0)
    m = randomize_weights(DynamicsWorldModel(**CFG2, matmul_dtype='fp32_fp16x2'), seed=0, terminal_bias=-10.).to(device)
    g = torch.Generator(device=device).manual_seed(1234)
    gk = dict(return_for_policy_optimization=True, num_steps=NUM_STEPS, generator=g)
    for _ in range(2):
        m.generate(HORIZON + 1, batch_size=B_LOCAL, **gk)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        e = m.generate(HORIZON + 1, batch_size=B_LOCAL, **gk)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    return dict(rollout_steps_per_sec=round(B_LOCAL * e.latents.shape[1] / dt, 1), generate_ms=round(1e3 * dt, 2),
                dtype='fp32-class on the fp16 MFMA: 23-bit operand images (two fp16 planes, exact power-of-two row scales), three products, fp32 accumulate',
                note='opt-in mode, not the default and not `value`: error vs float64 0.45x the f32-input MFMA on this model\'s dot products, but single products carry up '
                     'to 2^-21 relative error (fp32: 2^-24) - it fails the wide-exponent criterion of tests/test_gpu_kernels.py (profiles/r05_x3_products.txt)',
                workload=f'cfg2 rollout only: B={B_LOCAL}, H={HORIZON}, num_steps={NUM_STEPS}')


def cfg5_error_vs_oracle(device, kept):
    """The bf16 engine at config 5 against the oracle rollout the CPU baseline just timed (same weights, same injected draws incl. the Beta sampler's, the
    bench's 16-frame horizon through the KV cache): max |difference| per tensor.  Outside every timed region; the oracle is the checker, never measured here."""
    from dreamer4_amd import DynamicsWorldModel
    from dreamer4_amd.synthetic import randomize_weights
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    from util import rollout_parity_continuous
    torch.manual_seed(0)
    m = randomize_weights(DynamicsWorldModel(**CFG5, matmul_dtype='bf16'), terminal_bias=-10.).to(device)
    nz = kept['noise']
    e = m.generate(kept['frames'], batch_size=kept['batch'], return_for_policy_optimization=True, num_steps=NUM_STEPS, noise=nz).cpu()
    rep = rollout_parity_continuous(e, kept['exp'])
    rep['what'] = (f"the bf16 engine's rollout of B={kept['batch']} x {kept['frames']} frames vs oracle/restate.py (fp32) under the same injected draws; a trajectory is "
                   "'tracked' while every sampled Beta action stays within track_tol of the oracle's (a rejection-sampling decision inside the bf16 error flips "
                   "otherwise and the trajectory takes another path); *_max_abs over the tracked trajectories, *_max_abs_all over all")
    return rep


def cfg5_cpu_baseline(max_threads=16, budget_s=25., keep=None):
    """The CPU oracle at config 5's architecture (dim 1024, depth 12, 6 Beta actions; torch fp32 on the host), rank 0 at N = 1 only: a BOUNDED
    SAMPLE of the workload — the full one (B = 128, 16 frames: 69 TFLOP) would take minutes on the host — sized by a short probe to about `budget_s`."""
    from dreamer4_amd import DynamicsWorldModel
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    from util import make_noise, oracle_config, oracle_weights
    from dreamer4_amd.synthetic import randomize_weights
    from oracle import restate
    host = os.cpu_count() or 1
    cores = min(host, max_threads)
    torch.set_num_threads(cores)
    torch.manual_seed(0)
    m = randomize_weights(DynamicsWorldModel(**CFG5), terminal_bias=-10.)
    cfg, W = oracle_config(m), oracle_weights(m)
    heads = ('policy_head', 'value_head', 'action_embedder.continuous_action_unembed')

    def run(batch, frames):
        nz = make_noise(cfg, frames, batch, 1234)
        t0 = time.perf_counter()
        with torch.no_grad():
            exp = restate.generate(cfg, W, frames, batch_size=batch, noise=nz, num_steps=NUM_STEPS)
        Wg = {k: (v.clone().requires_grad_() if k.startswith(heads) else v) for k, v in W.items()}
        pl, vl = restate.learn_losses(cfg, Wg, exp, 'ppo')
        pl.backward(); vl.backward()
        if keep is not None and frames == HORIZON + 1:
            keep.update(exp=exp, noise=nz, batch=batch, frames=frames)
        return batch * exp['latents'].shape[1], time.perf_counter() - t0

    run(1, 1)
    steps, dt = run(4, 2)
    batch = int(max(2, min(128, budget_s / max(dt / steps, 1e-6) / (HORIZON + 1))))
    steps, dt = run(batch, HORIZON + 1)
    return dict(value=round(steps / dt, 2), unit='imagined steps/s', cores=cores, host_cores=host, kind='port',
                sample=f'BOUNDED SAMPLE: oracle/restate.py at cfg5 (dim=1024 depth=12, 6 continuous actions), generate(B={batch}, frames={HORIZON + 1}, '
                       f'num_steps={NUM_STEPS}) + learn(ppo): {steps} imagined steps in {dt:.1f} s, torch fp32, {cores} of {host} host threads')


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=3)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-kernel-timing', action='store_true', help='skip the per-launch HIP-event timing of the GEMMs')
    ap.add_argument('--no-secondary', action='store_true', help='skip the secondary cfg4 (decode latency) / cfg5 (bf16) measurements')
    ap.add_argument('--allow-experiment-env', action='store_true', help='run although an experiment switch (dreamer4_amd/knobs.py) is set; it is reported in the JSON')
    args = ap.parse_args()

    from dreamer4_amd import DreamTrainer, _lib, parallel
    from dreamer4_amd.knobs import experiment_overrides
    env_over = experiment_overrides()
    assert not env_over or args.allow_experiment_env, (f'experiment switches set: {env_over} - the bench measures the product defaults (the ones the GPU tests '
                                                      'run under); unset them or pass --allow-experiment-env')
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        # bare `python bench.py --gpus N`: become the launcher — one process per GPU under torch.distributed.run (trainers.py:1388-1396 leaves
        # this to `accelerate launch`); rank 0 of the children prints the one JSON line on this process's stdout
        import socket
        with socket.socket() as sk:
            sk.bind(('127.0.0.1', 0))
            port = sk.getsockname()[1]
        sys.stdout.flush(); sys.stderr.flush()
        os.execv(sys.executable, [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={args.gpus}',
                                  '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__), *sys.argv[1:]])
    world = int(os.environ.get('WORLD_SIZE', '1'))
    assert world == args.gpus, f'--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}'
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    assert torch.cuda.is_available(), 'bench.py needs an MI355X; there is no CPU path'
    backend = os.environ.get('D4_BENCH_BACKEND', 'nccl')   # 'gloo': the N-rank code path on FEWER devices than ranks (tests/test_gpu_dp.py; RCCL wants a device per rank)
    dev_index = local_rank if backend == 'nccl' else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    device = torch.device('cuda', dev_index)
    parallel.init_from_env(backend)                      # 'nccl' = RCCL over xGMI
    if world > 1 or os.environ.get('D4_BENCH_STRICT_TUNE') == '1':
        # N ranks must not each time GEMM tile configurations on their own clock (different choices per rank, first-step skew): every shape
        # of this workload has to come from the shipped table dreamer4_amd/gemm_tune_default.txt, else the first step fails with the shape named
        os.environ['D4_GEMM_AUTOTUNE'] = 'strict'
    rank = parallel.rank()

    model = build_model(device)
    parallel.broadcast_parameters(model.parameters())
    trainer = DreamTrainer(model, batch_size=B_LOCAL, generate_timesteps=HORIZON, objective='ppo', seed=1234,
                           generate_kwargs=dict(return_for_policy_optimization=True, num_steps=NUM_STEPS))
    lib = _lib.load()

    gpu_headline, headline_error = None, None
    if world == 1 and not args.no_cpu_baseline:
        # parity at the headline size: the SAME rollout (B = 256, 16 frames) under injected draws with the INITIAL weights (before any optimiser step
        # moves the heads), compared at the end with the oracle run that `cpu_baseline` times anyway (outside the timed region; the oracle is the
        # checker here, never the thing measured)
        sys.path.insert(0, os.path.join(ROOT, 'tests'))
        from util import make_noise, oracle_config
        try:                                              # (a failing parity leg must not cost the headline line)
            nz = make_noise(oracle_config(model), HORIZON + 1, B_LOCAL, 1234)
            gpu_headline = model.generate(HORIZON + 1, batch_size=B_LOCAL, return_for_policy_optimization=True, num_steps=NUM_STEPS, noise=nz).cpu()
        except Exception as exc:                          # noqa: BLE001
            gpu_headline, headline_error = None, repr(exc)
    timing = not args.no_kernel_timing
    ncls = lib.d4_profile_classes()
    raw_names = [lib.d4_profile_class_name(i).decode() for i in range(ncls)]
    is_split = [n.startswith(('gemm_x3_kernel', 'gemm_x3sk_kernel')) for n in raw_names]     # fp32 GEMM on the bf16 matrix cores: split operands, 6 bf16 MFMA products per fp32 product
    names = [n + ((' (persistent 128 x 128 / 128 x 64 form)' if n.startswith('gemm_x3sk') else ', *>') + ' fp32 by split operands on the bf16 MFMA' if sp else ', *> fp32 MFMA')
             for n, sp in zip(raw_names, is_split)]
    # warm-up: first-use tile autotuning of every GEMM shape and the capture of the decode frames' hipGraphs happen here
    dom, dom2, warm_classes, warm_exec, glue_measured, fused_flops = None, None, None, (0., 0.), None, 0.
    for w in range(args.warmup):
        trainer.train_step()                             # the default path: tile autotuning at first use, the decode frames' hipGraphs captured
    if timing:
        # one EXTRA untimed step with every GEMM / glue class event-timed: finds the dominant class and measures the glue kernels.  Per-launch
        # events cannot ride inside a replayed hipGraph, so this step enqueues its frames eagerly — the same kernels by the same rules
        # (tests/test_gpu_generate.py::test_eager_and_graph_replayed_frames_run_the_same_kernels_bitwise), a different launch mechanism.
        torch.cuda.synchronize()
        lib.d4_profile_enable((1 << ncls) - 1)
        lib.d4_profile_glue_enable((1 << lib.d4_profile_glue_classes()) - 1)
        trainer.train_step()
        torch.cuda.synchronize()
        lib.d4_profile_enable(0)
        lib.d4_profile_glue_enable(0)
        ng = lib.d4_profile_glue_classes()
        gms = (C.c_double * ng)(); gby = (C.c_double * ng)(); gcnt = (C.c_int64 * ng)(); gfl = (C.c_double * ng)()
        _lib.check(lib.d4_profile_glue_read_flops(gfl, ng))
        _lib.check(lib.d4_profile_glue_read(gms, gby, gcnt, ng))
        fused_flops = sum(gfl[i] for i in range(ng))          # GEMM work done inside the per-frame fused kernels (frame_fused.hip)
        glue_measured = {lib.d4_profile_glue_class_name(i).decode(): dict(
            hbm_gbs=round(gby[i] / max(gms[i], 1e-9) / 1e6, 1), frac_of_8tbs=round(gby[i] / max(gms[i], 1e-9) / 1e6 / 8000., 3),
            avg_us=round(1e3 * gms[i] / gcnt[i], 2), launches=int(gcnt[i]), ms_per_step=round(gms[i], 2),
            algorithmic_mb_per_launch=round(gby[i] / gcnt[i] / 1e6, 2),
            **({'matrix_gflop_per_launch': round(gfl[i] / gcnt[i] / 1e9, 2)} if gfl[i] > 0 else {})) for i in range(ng) if gcnt[i]}
        ms = (C.c_double * ncls)(); fl = (C.c_double * ncls)(); cnt = (C.c_int64 * ncls)()
        _lib.check(lib.d4_profile_read(ms, fl, cnt, ncls))
        dom = max(range(ncls), key=lambda i: ms[i])
        # ... and the largest class of the OTHER instruction family (f32-input MFMA vs split operands on the bf16 MFMA): the two lead the table
        # within a fraction of a millisecond of each other, so which one is "dominant" flips from box to box — both are event-timed and reported
        others = [i for i in range(ncls) if is_split[i] != is_split[dom] and ms[i] > 0]
        dom2 = max(others, key=lambda i: ms[i]) if others else None
        warm_classes = {names[i]: dict(ms=round(ms[i], 2), tflops=round(fl[i] / max(ms[i], 1e-9) / 1e9, 2), launches=int(cnt[i]))
                        for i in range(ncls) if cnt[i]}
        warm_exec = (sum(fl[i] for i in range(ncls)), sum(ms[i] for i in range(ncls)))      # EXECUTED flops / GEMM time of one step
    torch.cuda.synchronize()

    # ---- the timed region: the DEFAULT product path (no per-launch events; decode frames replayed from hipGraphs where the engine's rule says so)
    gen_ms, learn_ms = [], []
    frames_total = 0
    parallel.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3 * args.steps)]
    for k in range(args.steps):
        ev[3 * k].record()
        dreams = trainer.generate()
        ev[3 * k + 1].record()
        trainer.learn(dreams)
        ev[3 * k + 2].record()
        frames_total += dreams.latents.shape[1]
    parallel.barrier()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    for k in range(args.steps):
        gen_ms.append(ev[3 * k].elapsed_time(ev[3 * k + 1]))
        learn_ms.append(ev[3 * k + 1].elapsed_time(ev[3 * k + 2]))

    # ---- roofline pass: the SAME steps repeated right after the timed region with a HIP-event pair on every EVENT_STRIDE-th launch of the dominant
    # GEMM class (and of the largest class of the other instruction family).  Outside `value`: events force the eager launch mechanism.
    ev_ms_per_step = None
    if timing:
        psteps = max(1, min(args.steps, 3))
        lib.d4_profile_enable(((1 << dom) | ((1 << dom2) if dom2 is not None else 0)) | (EVENT_STRIDE << 27))
        torch.cuda.synchronize()
        tp = time.perf_counter()
        for k in range(psteps):
            trainer.learn(trainer.generate())
        torch.cuda.synchronize()
        ev_ms_per_step = 1e3 * (time.perf_counter() - tp) / psteps
        lib.d4_profile_enable(0)

    wall_t = torch.tensor([wall], device=device, dtype=torch.float64)
    parallel.all_reduce_max_(wall_t)
    wall = float(wall_t.item())
    # rank skew in one line: min / max over ranks of each rank's mean rollout and learner time (MAX all-reduce of (x, -x))
    mine = torch.tensor([sum(gen_ms) / len(gen_ms), sum(learn_ms) / len(learn_ms)], device=device, dtype=torch.float64)
    skew = torch.cat([mine, -mine])
    parallel.all_reduce_max_(skew)
    skew = skew.tolist()
    per_rank = dict(generate_ms_min=round(-skew[2], 2), generate_ms_max=round(skew[0], 2),
                    actor_critic_step_ms_min=round(-skew[3], 2), actor_critic_step_ms_max=round(skew[1], 2))
    steps_done = world * B_LOCAL * frames_total              # every rank generates the same number of frames
    value = steps_done / wall

    roofline = None
    if timing:
        ms = (C.c_double * ncls)(); fl = (C.c_double * ncls)(); cnt = (C.c_int64 * ncls)()
        _lib.check(lib.d4_profile_read(ms, fl, cnt, ncls))
        ach = fl[dom] / (ms[dom] * 1e-3) / 1e12 if ms[dom] > 0 else 0.
        traffic = None     # HBM bytes per launch of the dominant class from the committed rocprofv3 PMC passes (separate runs; profiles/pmc_traffic.json):
        try:               # the class has an RMS and a non-RMS instantiation -> the one with the larger share of GPU time
            pj = json.load(open(os.path.join(ROOT, 'profiles', 'pmc_traffic.json')))
            stem = raw_names[dom]
            cands = [v for k, v in pj.get('kernels', {}).items() if k.startswith(stem)]
            if cands:
                best = max(cands, key=lambda v: v['launches'] * v['avg_us'])
                traffic = best['hbm_bytes_per_launch']
        except (OSError, ValueError, KeyError):
            pass
        glue = glue_measured
        # a split-operand launch executes 6 bf16 MFMA products per fp32 product: its matrix-pipe bound is the bf16 dense peak / 6
        peak = PEAK_BF16_MFMA_TFLOPS / 6. if is_split[dom] else PEAK_FP32_MFMA_TFLOPS
        roofline = dict(bound='mfma', achieved=round(ach, 2), peak=round(peak, 1), unit='TFLOP/s',
                        frac=round(ach / peak, 4), traffic=traffic, kernel=names[dom],
                        launches_timed=int(cnt[dom]), event_stride=EVENT_STRIDE, avg_launch_us=round(1e3 * ms[dom] / max(cnt[dom], 1), 2),
                        flops_per_launch=round(fl[dom] / max(cnt[dom], 1)),
                        measured_in=('an event-timed repeat of the timed steps, run right after the timed region in the same process (per-launch HIP events '
                                     'cannot ride inside replayed hipGraphs, so that pass enqueues the decode frames eagerly: the same kernels by the same rules, '
                                     'asserted bit-identical in tests/test_gpu_generate.py); `value` / `ms_per_step` are the default path without events'),
                        ms_per_step_event_timed_pass=round(ev_ms_per_step, 2),
                        traffic_source='profiles/pmc_traffic.json: builder-run rocprofv3 --pmc passes of this workload (separate runs), not measured in this process',
                        all_gemm_configs_one_warmup_step=warm_classes,
                        # algorithmic = SURVEY's 5.23 GFLOP per imagined step (what the reference would execute); executed = the 2MNK of
                        # the GEMM launches this engine actually makes (it drops ~26 % of the reference's work: agent row, compacted
                        # rows, pool value restructure) -> only the executed figure is a roofline fraction
                        rollout_algorithmic_tflops=round(FLOP_PER_IMAGINED_STEP * B_LOCAL * (HORIZON + 1) / (sum(gen_ms) / len(gen_ms) * 1e-3) / 1e12, 2))
        if dom2 is not None and ms[dom2] > 0:
            ach2 = fl[dom2] / (ms[dom2] * 1e-3) / 1e12
            peak2 = PEAK_BF16_MFMA_TFLOPS / 6. if is_split[dom2] else PEAK_FP32_MFMA_TFLOPS
            roofline['second_class'] = dict(kernel=names[dom2], bound='mfma', achieved=round(ach2, 2), peak=round(peak2, 1), unit='TFLOP/s', frac=round(ach2 / peak2, 4),
                                            launches_timed=int(cnt[dom2]), avg_launch_us=round(1e3 * ms[dom2] / max(cnt[dom2], 1), 2),
                                            note='the largest class of the other instruction family (f32-input MFMA vs split operands on the bf16 MFMA); the two '
                                                 'classes carry ~42 ms of the step each, so which is dominant flips from box to box')
        if warm_classes is not None:
            roofline['executed_gemm_tflop_per_step'] = round(warm_exec[0] / 1e12, 2)
            roofline['executed_tflops_inside_gemm_kernels'] = round(warm_exec[0] / max(warm_exec[1], 1e-9) / 1e9, 2)
            # whole-step figure: the GEMM launches' flops + the matrix work the per-frame fused kernels do in place of GEMM launches
            roofline['executed_gemm_tflop_per_step_incl_fused'] = round((warm_exec[0] + fused_flops) / 1e12, 2)
            roofline['executed_tflops_over_the_whole_step'] = round((warm_exec[0] + fused_flops) / (1e-3 * (sum(gen_ms) + sum(learn_ms)) / len(gen_ms)) / 1e12, 2)
            roofline['executed_frac_of_fp32_matrix_peak_whole_step'] = round(roofline['executed_tflops_over_the_whole_step'] / PEAK_FP32_MFMA_TFLOPS, 4)
            roofline['note'] = ('fp32 throughout; the gemm_x3_kernel classes are fp32 GEMMs run as 6 bf16 MFMA products per fp32 product (operands split exactly '
                                'into three bf16 numbers, fp32 accumulation; error vs float64 below the f32-input MFMA kernels\'), bound = 2500 / 6 TFLOP/s; '
                                'executed_frac_of_fp32_matrix_peak_whole_step prices every GEMM flop against the 157.3 TFLOP/s f32-input MFMA peak')
        if glue:
            roofline['glue_kernels_hbm'] = glue

    if rank != 0:
        return
    out = dict(
        metric='imagined latent steps/sec', value=round(value, 1), unit='imagined steps/s', n_gpus=world, steps=args.steps,
        warmup=args.warmup, ms_per_step=round(1e3 * wall / args.steps, 2), higher_is_better=True, scaling='weak',
        vs_baseline=None, dtype='f32', data='synthetic',
        config=dict(workload='cfg2 Moving-MNIST-latent shape: dim=512 depth=6 heads=8x64 latents=32x32 spatial=4 registers=8; '
                             'generate(H+1=16 frames, num_steps=4, time cache) + learn_from_experience(ppo) + clip/AdamW both heads',
                    global_batch=world * B_LOCAL, per_gpu_batch=B_LOCAL, horizon=HORIZON, num_steps=NUM_STEPS, parallelism=f'dp{world}'),
        generate_ms=round(sum(gen_ms) / len(gen_ms), 2), actor_critic_step_ms=round(sum(learn_ms) / len(learn_ms), 2),
        per_rank=per_rank, experiment_env=env_over,
        rollout_steps_per_sec=round(world * B_LOCAL * (HORIZON + 1) / (sum(gen_ms) / len(gen_ms) * 1e-3), 1),
        roofline=roofline,
    )
    if world == 1 and not args.no_secondary:
        del trainer, model
        torch.cuda.empty_cache()
        try:
            out['measured_peaks'] = measured_peaks(device, lib)
        except Exception as exc:                      # noqa: BLE001
            out['measured_peaks_error'] = repr(exc)
        for key, fn in (('cfg4_env_step', lambda: cfg4_env_latency(device)), ('cfg5_bf16', lambda: cfg5_bf16(device, lib)),
                        ('train_flow_step', lambda: train_flow_step(device)), ('cfg2_fp16x2_mode', lambda: cfg2_fp16x2_mode(device))):
            try:                                      # a failing secondary measurement must not cost the headline line
                out[key] = fn()
            except Exception as exc:                  # noqa: BLE001
                out[key + '_error'] = repr(exc)
    if world == 1 and not args.no_cpu_baseline:
        kept = {}
        out['cpu_baseline'] = cpu_baseline(keep=kept)
        out['speedup_vs_cpu_baseline'] = round(value / out['cpu_baseline']['value'], 1)
        if 'cfg5_bf16' in out:
            kept5 = {}
            try:
                out['cfg5_bf16']['cpu_baseline'] = cfg5_cpu_baseline(keep=kept5)
                out['cfg5_bf16']['speedup_vs_cpu_baseline'] = round(out['cfg5_bf16']['whole_step_steps_per_sec'] / out['cfg5_bf16']['cpu_baseline']['value'], 1)
            except Exception as exc:                  # noqa: BLE001
                out['cfg5_bf16']['cpu_baseline_error'] = repr(exc)
            if kept5:
                try:
                    out['cfg5_bf16']['error_vs_oracle'] = cfg5_error_vs_oracle(device, kept5)
                except Exception as exc:              # noqa: BLE001
                    out['cfg5_bf16']['error_vs_oracle'] = dict(error=repr(exc))
        if kept and gpu_headline is not None:
            from util import first_trajectories, rollout_parity
            b = kept['batch']
            try:
                rep = rollout_parity(first_trajectories(gpu_headline, b), kept['exp'], kept['noise'], kept['cfg'])
            except Exception as exc:                      # noqa: BLE001 - the headline line must still be printed
                rep = dict(error=repr(exc))
            rep['what'] = (f'the GPU rollout of the first {b} of the B={B_LOCAL} trajectories x {HORIZON + 1} frames vs oracle/restate.py under the same injected draws '
                           '(max |difference| per tensor over the trajectories whose sampling margins are well posed; integers must be equal there)')
            out['parity_at_headline'] = rep
        elif headline_error:
            out['parity_at_headline'] = dict(error=headline_error)
        if (os.cpu_count() or 1) >= 64:
            out['cpu_baseline_sharded'] = cpu_baseline_sharded()
    print(json.dumps(out))


if __name__ == '__main__':
    main()
