"""GPU: the opt-in bf16 MFMA path of the trunk (BASELINE config 5: "MFMA bf16").  bf16 weights + activations, fp32 accumulation,
norms, softmax and residual stream.  Stated error: the kernel alone is exact up to fp32 summation order against a reference
computed from the SAME bf16-rounded operands; the engine in bf16 stays within 3e-2 (absolute, on latents in [-1, 1], values and
agent embeddings of unit scale) of the engine in fp32 on identical weights and noise at the config-5 shape."""
import ctypes as C

import pytest
import torch

from dreamer4_amd import DynamicsWorldModel, _lib
from util import make_noise, oracle_config, randomize_weights, small_model

pytestmark = pytest.mark.gpu


def stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


@pytest.mark.parametrize('M,N,K,flags', [(128, 128, 64, 0), (3584, 512, 512, 0), (200, 300, 96, 1), (1920, 2752, 1024, 5), (45, 388, 32, 3),
                                         (1920, 1552, 1024, 1), (17, 64, 2752, 0), (8192, 1024, 32, 1)])
def test_gemm_bf16_kernel(M, N, K, flags):
    lib = _lib.load()
    g = torch.Generator(device='cuda').manual_seed(0)
    A = torch.randn(M, K, device='cuda', generator=g); W = torch.randn(N, K, device='cuda', generator=g)
    b = torch.randn(N, device='cuda', generator=g)
    swiglu = bool(flags & _lib.GEMM_SWIGLU)
    R = None if swiglu else torch.randn(M, N, device='cuda', generator=g)
    Wb = W.to(torch.bfloat16).contiguous()
    Nout = N // 2 if swiglu else N
    out = torch.full((M, Nout), float('nan'), device='cuda')
    _lib.check(lib.d4_gemm_bf16(_lib.ptr(A), K, _lib.ptr(Wb), K, _lib.ptr(out), Nout, _lib.ptr(b), _lib.ptr(R), N, M, N, K, flags, 1.1920929e-07, stream()))
    Ad, Wd = A.to(torch.bfloat16).double(), Wb.double()
    ref = Ad @ Wd.t()
    if flags & _lib.GEMM_RMS_ROWSCALE:
        ref = ref * torch.rsqrt(A.double().pow(2).mean(-1, keepdim=True) + 1.1920929e-07)      # fp32 norm of the UNROUNDED activations
    ref = ref + b.double()
    if flags & _lib.GEMM_SILU:
        ref = torch.nn.functional.silu(ref)
    if swiglu:
        r = ref.reshape(M, N // 64, 2, 32)
        ref = (r[:, :, 0] * torch.nn.functional.silu(r[:, :, 1])).reshape(M, N // 2)
    if R is not None:
        ref = ref + R.double()
    err = (out.double() - ref).abs().max().item()
    tol = 3e-6 * max(1., ref.abs().max().item()) * max(1., K / 256) ** 0.5
    assert err <= tol, f'M{M} N{N} K{K} flags{flags}: err {err:.3e} > {tol:.3e}'


@pytest.mark.parametrize('M,N,K,flags', [(128, 128, 64, 0), (1792, 1024, 512, 0), (200, 300, 128, 1), (1920, 2752, 1024, 5), (45, 388, 64, 3), (1792, 1552, 1024, 1),
                                         (17, 64, 2752, 0), (700, 520, 192, 1), (515, 1088, 320, 5), (130, 390, 128, 1)])
def test_gemm_bf16_with_bf16_activations_every_configuration(M, N, K, flags):
    """gemm_bf16a.hip (d4_gemm_bf16a): both operands bf16 in HBM, LDS-DMA ring, v_mfma_f32_16x16x32_bf16.  Every tile configuration against a
    float64 product of the SAME bf16 operands (row scale from the bf16 activations), all epilogues, partial tiles; the bf16 copy of the output
    equals the rounded fp32 output."""
    lib = _lib.load()
    g = torch.Generator(device='cuda').manual_seed(0)
    Ab = torch.randn(M, K, device='cuda', generator=g).to(torch.bfloat16).contiguous()
    Wb = torch.randn(N, K, device='cuda', generator=g).to(torch.bfloat16).contiguous()
    b = torch.randn(N, device='cuda', generator=g)
    swiglu = bool(flags & _lib.GEMM_SWIGLU)
    R = None if swiglu else torch.randn(M, N, device='cuda', generator=g)
    Nout = N // 2 if swiglu else N
    eps = 1.1920929e-07
    Ad = Ab.double()
    ref = (Ad * torch.rsqrt(Ad.pow(2).mean(-1, keepdim=True) + eps) if flags & _lib.GEMM_RMS_ROWSCALE else Ad) @ Wb.double().t() + b.double()
    if flags & _lib.GEMM_SILU:
        ref = torch.nn.functional.silu(ref)
    if swiglu:
        r = ref.reshape(M, N // 64, 2, 32)
        ref = (r[:, :, 0] * torch.nn.functional.silu(r[:, :, 1])).reshape(M, N // 2)
    if R is not None:
        ref = ref + R.double()
    tol = 3e-6 * max(1., ref.abs().max().item()) * max(1., K / 256) ** 0.5
    ran = 0
    for cfg in (-1, 0, 1, 2, 3, 4, 5, 6, 7):              # 6: the phased 256 x 256 form (gemm_bf16p.hip); odd and even k-tile counts, partial tiles in both directions
        out = torch.full((M, Nout), float('nan'), device='cuda'); outb = torch.zeros(M, Nout, device='cuda', dtype=torch.bfloat16)
        rc = lib.d4_gemm_bf16a(_lib.ptr(Ab), K, _lib.ptr(Wb), K, _lib.ptr(out), Nout, _lib.ptr(outb), _lib.ptr(b), _lib.ptr(R), N, M, N, K, flags, eps, cfg, stream())
        if rc != 0:
            assert swiglu and cfg in (1, 2, 4), lib.d4_last_error()       # the SiLU-GLU pairing needs a wave tile of 64 columns
            continue
        ran += 1
        assert (out.double() - ref).abs().max().item() <= tol, (cfg, (out.double() - ref).abs().max().item(), tol)
        assert torch.equal(outb, out.to(torch.bfloat16)), cfg
        if swiglu:          # the engine's form: only the bf16 image is written (the phased kernel stages it through LDS into 16-byte stores)
            only = torch.zeros(M, Nout, device='cuda', dtype=torch.bfloat16)
            _lib.check(lib.d4_gemm_bf16a(_lib.ptr(Ab), K, _lib.ptr(Wb), K, None, Nout, _lib.ptr(only), _lib.ptr(b), None, 0, M, N, K, flags, eps, cfg, stream()))
            assert torch.equal(only, outb), cfg
    assert ran >= 4
    with pytest.raises(_lib.D4Error, match='not supported'):
        _lib.check(lib.d4_gemm_bf16a(_lib.ptr(Ab), K, _lib.ptr(Wb), K, _lib.ptr(out), Nout, None, None, None, 0, M, N, 96, 0, eps, -1, stream()))


@pytest.mark.parametrize('frames,S,N,K,lo,hi,last', [(40, 15, 512, 128, 1, 5, 1), (37, 14, 320, 64, 1, 5, 0), (300, 15, 1024, 256, 1, 5, 1)])
def test_gemm_bf16a_row_compacted_second_output(frames, S, N, K, lo, hi, last):
    """The engine's second output of a residual-stream GEMM: the token rows the final attention pool / latent head read (tokens [lo, hi) of every frame of S
    tokens, plus the frame's last token when `last`), fp32 and its bf16 image, written by the same epilogue — every tile configuration (the phased
    256 x 256 form stages its epilogue through LDS: rows are re-assigned to lanes there) equals a row gather of the full output."""
    lib = _lib.load()
    g = torch.Generator(device='cuda').manual_seed(1)
    M = frames * S
    Ab = torch.randn(M, K, device='cuda', generator=g).to(torch.bfloat16).contiguous()
    Wb = (torch.randn(N, K, device='cuda', generator=g) / K ** 0.5).to(torch.bfloat16).contiguous()
    R = torch.randn(M, N, device='cuda', generator=g)
    keep = hi - lo + last
    rows = torch.tensor([f * S + t for f in range(frames) for t in list(range(lo, hi)) + ([S - 1] if last else [])], device='cuda')
    for cfg in (-1, 0, 1, 2, 3, 4, 5, 6, 7):
        out = torch.full((M, N), float('nan'), device='cuda'); outb = torch.zeros(M, N, device='cuda', dtype=torch.bfloat16)
        c2 = torch.full((frames * keep, N), float('nan'), device='cuda'); c2b = torch.zeros(frames * keep, N, device='cuda', dtype=torch.bfloat16)
        _lib.check(lib.d4_gemm_bf16a_compact(_lib.ptr(Ab), K, _lib.ptr(Wb), K, _lib.ptr(out), N, _lib.ptr(outb), None, _lib.ptr(R), N, M, N, K, 0, 1e-6,
                                             _lib.ptr(c2), _lib.ptr(c2b), N, S, lo, hi, last, cfg, stream()))
        torch.cuda.synchronize()
        ref = Ab.double() @ Wb.double().t() + R.double()
        assert (out.double() - ref).abs().max().item() <= 3e-6 * max(1., ref.abs().max().item()), cfg
        assert torch.equal(c2, out[rows]) and torch.equal(c2b, outb[rows]) and torch.equal(outb, out.to(torch.bfloat16)), cfg


@pytest.mark.parametrize('M,N,K,flags,batch', [(1792, 1024, 512, 0, 1), (1920, 2752, 1024, 5, 1), (1792, 64, 1024, 0, 4), (130, 256, 192, 1, 1)])
def test_gemm_bf16a_bf16_only_output_and_batches(M, N, K, flags, batch):
    """The bf16 engine's producer -> consumer hand-off: a GEMM whose fp32 output nobody reads writes only the bf16 image (C = null, Cb set),
    bit-identical to the image written next to an fp32 output; the per-head batched form (the attention pool's value projection: A head slices
    at stride K, outputs at stride N) gives each head's plain product.  d4_cvt_rows_bf16 (the conversion pass after non-GEMM producers) is
    round-to-nearest-even of strided rows."""
    lib = _lib.load()
    g = torch.Generator(device='cuda').manual_seed(0)
    A = torch.randn(M, batch * K, device='cuda', generator=g)
    Ab = A.to(torch.bfloat16).contiguous(); Wb = (torch.randn(batch * N, K, device='cuda', generator=g) / K ** 0.5).to(torch.bfloat16).contiguous()
    swiglu = bool(flags & _lib.GEMM_SWIGLU)
    Nout = (N // 2 if swiglu else N) * batch
    eps = 1.1920929e-07

    def run(with_c):
        out = torch.full((M, Nout), float('nan'), device='cuda'); outb = torch.zeros(M, Nout, device='cuda', dtype=torch.bfloat16)
        _lib.check(lib.d4_gemm_bf16a_batched(_lib.ptr(Ab), batch * K, _lib.ptr(Wb), K, _lib.ptr(out) if with_c else None, Nout, _lib.ptr(outb), None, None, 0,
                                             M, N, K, flags, eps, batch, K, N * K, Nout // batch, -1, stream()))
        torch.cuda.synchronize()
        return out, outb

    out, outb = run(True)
    _, only = run(False)
    assert torch.equal(outb, out.to(torch.bfloat16)) and torch.equal(only, outb)
    Ad, Wd = Ab.double().view(M, batch, K), Wb.double().view(batch, N, K)
    X = Ad * torch.rsqrt(Ad.pow(2).mean(-1, keepdim=True) + eps) if flags & _lib.GEMM_RMS_ROWSCALE else Ad
    ref = torch.einsum('mbk,bnk->mbn', X, Wd)
    if swiglu:
        r = ref.reshape(M, batch, N // 64, 2, 32)
        ref = (r[:, :, :, 0] * torch.nn.functional.silu(r[:, :, :, 1])).reshape(M, batch, N // 2)
    ref = ref.reshape(M, Nout)
    assert (out.double() - ref).abs().max().item() <= 2e-6 * max(1., ref.abs().max().item()) * (K / 64) ** 0.5
    # the conversion pass: strided rows, a width that is not a multiple of 4 included
    src = torch.randn(37, 100, device='cuda', generator=g)
    for cols in (100, 98, 64):
        dst = torch.zeros(37, 104, device='cuda', dtype=torch.bfloat16)
        _lib.check(lib.d4_cvt_rows_bf16(_lib.ptr(src), 100, _lib.ptr(dst), 104, 37, cols, stream()))
        torch.cuda.synchronize()
        assert torch.equal(dst[:, :cols], src[:, :cols].to(torch.bfloat16)) and not dst[:, cols:].any()


def _pair(kw, seed=0, dtypes=('fp32', 'bf16')):
    torch.manual_seed(seed)
    a = randomize_weights(DynamicsWorldModel(**kw, matmul_dtype=dtypes[0]), seed=seed)
    b = DynamicsWorldModel(**kw, matmul_dtype=dtypes[1])
    b.load_state_dict(a.state_dict())
    return a.cuda(), b.cuda()


def test_small_engine_bf16_tracks_fp32():
    kw = dict(dim=64, dim_latent=8, num_latent_tokens=6, depth=4, time_block_every=2, attn_heads=2, num_discrete_actions=4)
    a, b = _pair(kw)
    cfg = oracle_config(a)
    nz = make_noise(cfg, 4, 3, 7)
    ea = a.generate(4, batch_size=3, return_rewards_per_frame=True, return_agent_actions=True, return_log_probs_and_values=True, noise=nz)
    eb = b.generate(4, batch_size=3, return_rewards_per_frame=True, return_agent_actions=True, return_log_probs_and_values=True, noise=nz)
    d = (ea.latents - eb.latents).abs().max().item()
    assert 0. < d < 3e-2, d            # not bit-identical (it really ran in bf16), and close
    assert (ea.values - eb.values).abs().max().item() < 0.2           # values live on [-20, 20]


@pytest.mark.parametrize('kw,B', [(dict(dim=512, dim_latent=32, num_latent_tokens=32, depth=6, num_discrete_actions=4), 24),
                                  (dict(dim=192, dim_latent=16, num_latent_tokens=8, depth=3, time_block_every=3, attn_heads=3, num_discrete_actions=(3, 2)), 5)])
def test_bf16_engine_with_mixed_image_coverage_tracks_fp32(kw, B):
    """bf16 engines whose shapes do NOT let every GEMM take the bf16-activation kernel: at dim 512 the SiLU-GLU hidden is 1376 wide (K % 64 = 32: its
    consumer stays on the fp32-activation kernel, so the producer must keep writing the fp32 buffer next to the image), at dim 192 with 3 heads most
    contractions are not multiples of 64 — producer -> consumer hand-offs through images, conversion passes and fp32 buffers all in one rollout.
    Same bound against the fp32 engine as the small and the config-5 models."""
    a, b = _pair(kw)
    cfg = oracle_config(a)
    nz = make_noise(cfg, 4, B, 7)
    gk = dict(return_rewards_per_frame=True, return_agent_actions=True, return_log_probs_and_values=True, return_terminals=False)
    ea = a.generate(4, batch_size=B, noise=nz, **gk)
    eb = b.generate(4, batch_size=B, noise=nz, **gk)
    lat = (ea.latents - eb.latents).abs()
    assert 0. < lat.max().item() < 3e-2 and lat.mean().item() < 3e-3, (lat.max().item(), lat.mean().item())
    assert (ea.agent_embed - eb.agent_embed).abs().max().item() < 3e-2 * max(1., ea.agent_embed.abs().max().item())
    assert (ea.values - eb.values).abs().max().item() < 0.2


def test_config5_shape_bf16_vs_fp32_at_the_per_gpu_batch():
    """dim 1024, depth 12, 64 x 32 latents, 6 continuous (Beta) actions, B = 128 (= 1024 / 8 GPUs), 3 frames x (4 + 1) evaluations:
    the bf16 engine against the fp32 engine on identical weights and noise."""
    kw = dict(dim=1024, dim_latent=32, num_latent_tokens=64, depth=12, num_continuous_actions=6)
    a, b = _pair(kw)
    with torch.no_grad():
        for m in (a, b):
            m.action_embedder.continuous_action_unembed.mul_(30.)
    cfg = oracle_config(a)
    B, T = 128, 3
    nz = make_noise(cfg, T, B, 3)
    gk = dict(return_rewards_per_frame=True, return_agent_actions=True, return_log_probs_and_values=True, return_terminals=False)
    ea = a.generate(T, batch_size=B, noise=nz, **gk)
    eb = b.generate(T, batch_size=B, noise=nz, **gk)
    lat = (ea.latents - eb.latents).abs()
    emb = (ea.agent_embed - eb.agent_embed).abs()
    print(f'\\ncfg5 bf16 vs fp32: latents max {lat.max().item():.3e} mean {lat.mean().item():.3e} | agent_embed max {emb.max().item():.3e} '
          f'(scale {ea.agent_embed.abs().max().item():.2f}) | values max {(ea.values - eb.values).abs().max().item():.3e}')
    assert lat.max().item() < 3e-2 and lat.mean().item() < 3e-3
    assert emb.max().item() < 3e-2 * max(1., ea.agent_embed.abs().max().item())
    # frame 0's Beta parameters come from one evaluation chain: the sampled continuous actions agree to bf16 accuracy
    pa, pb = ea.old_action_unembeds.continuous[:, 0], eb.old_action_unembeds.continuous[:, 0]
    assert (pa - pb).abs().max().item() < 2e-2 * pa.abs().max().item()            # raw parameters of scale ~25: 1-2 % relative


def test_config5_bf16_engine_vs_oracle_at_the_bench_horizon():
    """BASELINE config 5 as bench.py times it — dim 1024, depth 12, 64 x 32 latents, 6 continuous (Beta) actions, the bf16 engine, 16 frames x (4 + 1)
    evaluations through the KV cache — against the fp32 CPU oracle (oracle/restate.py) under the same injected draws, Beta sampler included (B = 8: the
    oracle finishes in seconds).  No exactness claim in this mode; the STATED bounds after 16 chained frames, on the trajectories whose sampled actions
    stayed on the oracle's path (a rejection-sampling decision inside the bf16 error sends a trajectory down another path: reported, not compared):
    latents (clamped to [-1, 1]) 4e-2 max / 4e-3 mean, agent embedding 6e-2 of its scale, values (range +-20) 0.3, Beta log-probs 0.3, actions 5e-2."""
    from oracle import restate
    from util import oracle_weights, rollout_parity_continuous
    kw = dict(dim=1024, dim_latent=32, num_latent_tokens=64, depth=12, num_continuous_actions=6)
    torch.manual_seed(0)
    ref_m = randomize_weights(DynamicsWorldModel(**kw), terminal_bias=-10.)
    cfg, W = oracle_config(ref_m), oracle_weights(ref_m)
    m = DynamicsWorldModel(**kw, matmul_dtype='bf16')
    m.load_state_dict(ref_m.state_dict())
    B, T = 8, 16
    nz = make_noise(cfg, T, B, 1234)
    with torch.no_grad():
        ref = restate.generate(cfg, W, T, batch_size=B, noise=nz, num_steps=4)
    e = m.cuda().generate(T, batch_size=B, return_for_policy_optimization=True, num_steps=4, noise=nz).cpu()
    rep = rollout_parity_continuous(e, ref)
    print('\ncfg5 bf16 vs oracle, 16 frames:', {k: (round(v, 5) if isinstance(v, float) else v) for k, v in rep.items()})
    assert rep['frames_equal'] and rep['tracked_trajectories'] >= B // 2, rep
    assert rep['latents_max_abs'] < 4e-2 and rep['latents_mean_abs'] < 4e-3, rep
    assert rep['agent_embed_max_abs'] < 6e-2 * max(1., rep['agent_embed_scale']), rep
    assert rep['values_max_abs'] < 0.3 and rep['cont_logp_max_abs'] < 0.3 and rep['actions_cont_max_abs'] <= 5e-2, rep


@pytest.mark.parametrize('kw,B', [(dict(dim=512, dim_latent=32, num_latent_tokens=32, depth=6, num_discrete_actions=4), 24),
                                  (dict(dim=64, dim_latent=8, num_latent_tokens=6, depth=4, time_block_every=2, attn_heads=2, num_discrete_actions=4), 3),
                                  (dict(dim=1024, dim_latent=32, num_latent_tokens=64, depth=3, num_continuous_actions=6), 9)])
def test_bf16_wide_key_projection_equals_one_key_projection_per_pool(kw, B):
    """bf16 engine: a hidden is projected ONCE, when it is produced, onto the key weights of every later attention pool and the query weights of the
    pool it feeds (engine.hip: pool_block, one launch of N = (depth - p) * 256 per pool) instead of a query launch + a key launch over all 2p + 3
    hiddens per pool (D4:2143-2177 re-projects the whole stack in every pool).  Same products and row scales; only the QUERIES change arithmetic
    (rounded to bf16 like the keys): against the per-pool form (test hook) the rollout differs, and by no more than bf16 rounding of one operand."""
    lib = _lib.load()
    a, b = _pair(kw, dtypes=('bf16', 'bf16'))            # two engines: a captured decode graph keeps the launch sequence it was recorded with
    cfg = oracle_config(a)
    nz = make_noise(cfg, 4, B, 5)
    gk = dict(return_rewards_per_frame=True, return_agent_actions=True, return_log_probs_and_values=True, noise=nz)
    try:
        assert lib.d4_debug_switch(b'pool_wide_keys', 1) in (0, 1)
        ea = a.generate(4, batch_size=B, **gk)
        lib.d4_debug_switch(b'pool_wide_keys', 0)
        eb = b.generate(4, batch_size=B, **gk)
    finally:
        lib.d4_debug_switch(b'pool_wide_keys', 1)
    d = (ea.latents - eb.latents).abs().max().item()
    assert 0. < d < 1e-2, d                      # it took the other path (bf16 queries), and stays inside the mode's own noise (3e-2 against fp32)
    assert (ea.values - eb.values).abs().max().item() < 0.1


def test_fp32_default_with_split_operand_projections_equals_the_f32_mfma_engine_to_fp32_accuracy():
    """matmul_dtype='fp32' (default) runs the SiLU-GLU input projections as split-operand fp32 GEMMs on the bf16 matrix cores
    (csrc/gemm_x3.hip); 'fp32_mfma' keeps every GEMM on the f32-input MFMA.  Both are fp32 arithmetic: at config 2's architecture
    (B = 32: 448 token rows per evaluation, 6 frames x 5 evaluations) they agree to fp32 rounding noise and sample the same actions."""
    kw = dict(dim=512, dim_latent=32, num_latent_tokens=32, depth=6, num_discrete_actions=4)
    a, b = _pair(kw, dtypes=('fp32', 'fp32_mfma'))
    cfg = oracle_config(a)
    B, T = 32, 6
    nz = make_noise(cfg, T, B, 11)
    gk = dict(return_rewards_per_frame=True, return_agent_actions=True, return_log_probs_and_values=True, return_terminals=False)
    ea = a.generate(T, batch_size=B, noise=nz, **gk)
    eb = b.generate(T, batch_size=B, noise=nz, **gk)
    d = (ea.latents - eb.latents).abs().max().item()
    assert 0. < d < 2e-5, d                    # different summation order (it really took the other kernel), fp32 noise only
    assert torch.equal(ea.actions.discrete, eb.actions.discrete)
    assert (ea.values - eb.values).abs().max().item() < 2e-4
    assert (ea.agent_embed - eb.agent_embed).abs().max().item() < 2e-5 * max(1., ea.agent_embed.abs().max().item())


@pytest.mark.parametrize('B,T', [(2, 6), (24, 3)])
def test_fp16x2_mode_matches_the_oracle_like_the_default_mode(B, T):
    """matmul_dtype='fp32_fp16x2' (csrc/gemm_h2.hip, opt-in): the trunk's larger GEMMs on the fp16 matrix cores — operands as two fp16 planes under exact
    power-of-two row scales, three products, fp32 accumulate; row exponents of the residual-stream slabs computed once per evaluation.  At BASELINE
    config 2's architecture the rollout must match the CPU oracle within the SAME bounds as the default engine (2e-4, integers exact), it
    really runs other kernels (not bit-identical to the default engine), and the engines stay within 2e-5 of each other."""
    from oracle import restate
    from util import oracle_weights
    kw = dict(dim=512, dim_latent=32, num_latent_tokens=32, depth=6, num_discrete_actions=4)
    a, b = _pair(kw, dtypes=('fp32', 'fp32_fp16x2'))
    cfg, W = oracle_config(a), oracle_weights(a)
    nz = make_noise(cfg, T, B, 5)
    gk = dict(return_rewards_per_frame=True, return_agent_actions=True, return_log_probs_and_values=True, noise=nz)
    ea, eb = a.generate(T, batch_size=B, **gk), b.generate(T, batch_size=B, **gk)
    assert torch.equal(ea.actions.discrete, eb.actions.discrete)
    d = (ea.latents - eb.latents).abs().max().item()
    assert d < 2e-5, d
    if B * 14 >= 256:                  # (the family takes calls of >= 256 rows: below that the two engines run the same kernels, bit for bit)
        assert d > 0.
    assert (ea.agent_embed - eb.agent_embed).abs().max().item() < 5e-5 and (ea.values - eb.values).abs().max().item() < 2e-5
    if B <= 2:
        torch.set_num_threads(min(16, torch.get_num_threads()))
        ref = restate.generate(cfg, W, T, batch_size=B, noise=nz, return_terminals=False)
        for x, y in ((eb.latents, ref['latents']), (eb.agent_embed, ref['agent_embed']), (eb.values, ref['values']), (eb.log_probs.discrete, ref['log_probs'])):
            assert torch.allclose(x.cpu(), y, atol=2e-4, rtol=1e-4), (x.cpu() - y).abs().max().item()
        assert torch.equal(eb.actions.discrete.cpu(), ref['actions'])
