"""GPU: single-kernel parity through the C-ABI (the same launchers the engine uses)."""
import ctypes as C

import pytest
import torch

from dreamer4_amd import _lib

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def lib():
    assert torch.cuda.is_available(), 'these tests need the MI355X'
    return _lib.load()


def stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def run_gemm(lib, M, N, K, flags=0, bias=False, res=False, ta=False, tb=False, seed=0):
    g = torch.Generator(device='cuda').manual_seed(seed)
    Mp = (M + 3) // 4 * 4          # leading dimensions must be multiples of 4 floats (16-byte rows)
    A_full = torch.randn((K, Mp) if ta else (M, K), device='cuda', generator=g)
    A = A_full[:, :M] if ta else A_full
    W = torch.randn((K, N) if tb else (N, K), device='cuda', generator=g)
    b = torch.randn(N, device='cuda', generator=g) if bias else None
    R = torch.randn(M, N, device='cuda', generator=g) if res else None
    Nout = N // 2 if flags & _lib.GEMM_SWIGLU else N
    out = torch.full((M, Nout), float('nan'), device='cuda')
    f = flags | (_lib.GEMM_TRANS_A if ta else 0) | (_lib.GEMM_TRANS_B if tb else 0)
    _lib.check(lib.d4_gemm(_lib.ptr(A_full), A_full.shape[1], _lib.ptr(W), W.shape[1], _lib.ptr(out), Nout, _lib.ptr(b), _lib.ptr(R), N,
                           M, N, K, f, 1.1920929e-07, stream()))
    Ad = (A.t() if ta else A).double()
    Wd = (W if tb else W.t()).double()
    X = Ad * torch.rsqrt(Ad.pow(2).mean(-1, keepdim=True) + 1.1920929e-07) if flags & _lib.GEMM_RMS_ROWSCALE else Ad
    ref = X @ Wd
    if bias:
        ref = ref + b.double()
    if flags & _lib.GEMM_SILU:
        ref = torch.nn.functional.silu(ref)
    if flags & _lib.GEMM_SWIGLU:
        r = ref.reshape(M, N // 64, 2, 32)
        ref = (r[:, :, 0] * torch.nn.functional.silu(r[:, :, 1])).reshape(M, N // 2)
    if res:
        ref = ref + R.double()
    err = (out.double() - ref).abs().max().item()
    tol = 2e-6 * max(1., ref.abs().max().item()) * max(1., K / 256) ** 0.5
    assert err <= tol, f'M{M} N{N} K{K} flags{f}: err {err:.3e} > {tol:.3e}'
    return out


@pytest.mark.parametrize('M,N,K', [(128, 128, 32), (64, 64, 32), (200, 300, 72), (3, 4, 256), (45, 388, 64), (15, 2064, 512),
                                   (3840, 512, 512), (1, 255, 512), (50000, 512, 512), (96, 1024, 8), (14, 512, 1376), (16, 1552, 512), (8, 1024, 16),
                                   (4, 2048, 2048), (17, 512, 512)])
def test_gemm_nt(lib, M, N, K):
    run_gemm(lib, M, N, K)
    run_gemm(lib, M, N, K, flags=_lib.GEMM_RMS_ROWSCALE, bias=True)
    run_gemm(lib, M, N, K, flags=_lib.GEMM_SILU, bias=True, res=True)


@pytest.mark.parametrize('M,N,K', [(45, 192, 64), (3840, 2752, 512), (256, 128, 32), (15, 2752, 512), (1, 64, 2048), (16, 192, 1376)])
def test_gemm_swiglu_epilogue(lib, M, N, K):
    run_gemm(lib, M, N, K, flags=_lib.GEMM_RMS_ROWSCALE | _lib.GEMM_SWIGLU, bias=True)


@pytest.mark.parametrize('M,N,K,ta,tb', [(100, 200, 300, False, True), (256, 2048, 4096, True, True), (4, 2048, 4096, True, True),
                                         (255, 128, 30 * 4, True, True), (128, 64, 40, True, False)])
def test_gemm_transposed_operands(lib, M, N, K, ta, tb):
    run_gemm(lib, M, N, K, ta=ta, tb=tb)


def test_gemm_is_deterministic(lib):
    A = torch.randn(3840, 512, device='cuda'); W = torch.randn(1552, 512, device='cuda')
    outs = []
    for _ in range(2):
        o = torch.empty(3840, 1552, device='cuda')
        _lib.check(lib.d4_gemm(_lib.ptr(A), 512, _lib.ptr(W), 512, _lib.ptr(o), 1552, None, None, 0, 3840, 1552, 512, 1, 1e-7, stream()))
        outs.append(o)
    assert torch.equal(outs[0], outs[1])


@pytest.mark.parametrize('M,N,K,flags', [(3584, 1552, 512, 1), (3584, 512, 1376, 0), (3584, 2752, 512, 5), (1000, 300, 72, 3), (300, 520, 256, 1)])
def test_every_tile_configuration_gives_the_same_bits(lib, M, N, K, flags):
    """The tuner may pick any tile configuration per shape, so all of them must agree bit for bit (same k order, same MFMA,
    one canonical order for the folded RMSNorm's row sums)."""
    g = torch.Generator(device='cuda').manual_seed(3)
    A = torch.randn(M, K, device='cuda', generator=g); W = torch.randn(N, K, device='cuda', generator=g); b = torch.randn(N, device='cuda', generator=g)
    Nout = N // 2 if flags & _lib.GEMM_SWIGLU else N
    outs = []
    n = lib.d4_gemm_force_config(-1)
    try:
        for cfg in range(n):
            lib.d4_gemm_force_config(cfg)
            o = torch.full((M, Nout), float('nan'), device='cuda')
            _lib.check(lib.d4_gemm(_lib.ptr(A), K, _lib.ptr(W), K, _lib.ptr(o), Nout, _lib.ptr(b), None, 0, M, N, K, flags, 1e-6, stream()))
            outs.append(o)
    finally:
        lib.d4_gemm_force_config(-1)
    assert n >= 8
    for o in outs[1:]:
        assert torch.equal(o, outs[0])


@pytest.mark.parametrize('M,N,K', [(256, 2048, 2048), (128, 4096, 4096), (250, 1030, 1024), (37, 4100, 1152), (200, 1100, 1024)])
def test_gemm_few_rows_long_k_cut_inside_the_workgroup(lib, M, N, K):
    """gemm2_ksplit_kernel (the heads' hidden layers at rollout batch: few rows, K >= 1024): a workgroup of 16 waves owns a 32 x 64 tile, its four wave groups
    multiply one quarter of K each and the partial tiles are summed in LDS in group order.  Taken by a rule on the shape (the dispatcher, not the tuner);
    against float64 with bias / SiLU / residual, ragged rows and column counts that are not multiples of 64 or 4; and deterministic (same bits twice)."""
    cls = [lib.d4_profile_class_name(c).decode() for c in range(lib.d4_profile_classes())]
    ks = cls.index('gemm2_ksplit_kernel')
    for kw in (dict(), dict(flags=_lib.GEMM_SILU, bias=True), dict(bias=True, res=True)):
        lib.d4_profile_enable(1 << ks)
        a = run_gemm(lib, M, N, K, **kw)
        torch.cuda.synchronize()
        ms = (C.c_double * len(cls))(); fl = (C.c_double * len(cls))(); cnt = (C.c_int64 * len(cls))()
        _lib.check(lib.d4_profile_read(ms, fl, cnt, len(cls)))
        lib.d4_profile_enable(0)
        assert cnt[ks] == 1, 'the call did not take the k-split kernel'
        b = run_gemm(lib, M, N, K, **kw)
        assert torch.equal(a, b)
    # k-tile counts that do not divide by four (the first groups take one more tile, the others run their last iteration on zeros), forced through the
    # test hook on shapes the rule does not take
    lib.d4_gemm_force_config(199)
    try:
        for (m, n, k) in ((100, 520, 1376), (1024, 512, 1376), (33, 64, 160)):
            run_gemm(lib, m, n, k, bias=True, res=True)
    finally:
        lib.d4_gemm_force_config(-1)


@pytest.mark.parametrize('M,N,K', [(112, 64, 32), (3584, 512, 512), (200, 300, 64), (45, 388, 96), (1000, 255, 128), (3840, 1552, 512),
                                   (17, 20, 32), (3584, 512, 1376), (130, 129, 2048)])
def test_gemm_second_family_every_configuration(lib, M, N, K):
    """gemm2.hip (16x16x4 MFMA fed by the LDS-DMA ring): every configuration against fp64, partial tiles included, and all of
    them bit-identical to each other (the choice among them is made by timing)."""
    n1 = lib.d4_gemm_force_config(-1)
    n2 = sum(lib.d4_profile_class_name(c).decode().startswith('gemm2_kernel') for c in range(lib.d4_profile_classes()))
    assert n2 >= 4
    variants = [dict(), dict(flags=_lib.GEMM_RMS_ROWSCALE, bias=True), dict(flags=_lib.GEMM_SILU, bias=True, res=True)]
    ref = None
    try:
        for c in range(n2):
            lib.d4_gemm_force_config(100 + c)
            outs = [run_gemm(lib, M, N, K, **v) for v in variants]
            if ref is None:
                ref = outs
            for o, r in zip(outs, ref):
                assert torch.equal(o, r), f'configuration {c} differs'
    finally:
        lib.d4_gemm_force_config(-1)


@pytest.mark.parametrize('M,N,K', [(45, 192, 64), (3584, 2752, 512), (256, 128, 32), (300, 2752, 512)])
def test_gemm_second_family_swiglu(lib, M, N, K):
    n1 = lib.d4_gemm_force_config(-1)
    n2 = sum(lib.d4_profile_class_name(c).decode().startswith('gemm2_kernel') for c in range(lib.d4_profile_classes()))
    outs = []
    try:
        for c in range(n2):
            lib.d4_gemm_force_config(100 + c)      # configurations without the SiLU-GLU epilogue fall through to the default path
            outs.append(run_gemm(lib, M, N, K, flags=_lib.GEMM_RMS_ROWSCALE | _lib.GEMM_SWIGLU, bias=True))
    finally:
        lib.d4_gemm_force_config(-1)


def test_gemm_row_scale_is_identical_across_families(lib):
    """The folded RMSNorm's 1/rms uses one canonical summation order in both families: with W = I the outputs are x / rms(x)."""
    g = torch.Generator(device='cuda').manual_seed(5)
    A = torch.randn(300, 64, device='cuda', generator=g); W = torch.eye(64, device='cuda')
    outs = []
    try:
        for cfg in (4, 100, 104, 105):
            lib.d4_gemm_force_config(cfg)
            o = torch.empty(300, 64, device='cuda')
            _lib.check(lib.d4_gemm(_lib.ptr(A), 64, _lib.ptr(W), 64, _lib.ptr(o), 64, None, None, 0, 300, 64, 64, 1, 1e-6, stream()))
            outs.append(o)
    finally:
        lib.d4_gemm_force_config(-1)
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])


def split_planes(lib, W):
    """W [N][K] fp32 -> the three bf16 planes d4_gemm_split reads (d4_split_bf16x3)."""
    n = W.numel()
    plane = (n + 7) // 8 * 8
    W3 = torch.zeros(3 * plane, dtype=torch.bfloat16, device='cuda')
    _lib.check(lib.d4_split_bf16x3(_lib.ptr(W), _lib.ptr(W3), n, plane, stream()))
    return W3, plane


def test_split_bf16x3_is_an_exact_decomposition(lib):
    """Every fp32 number is the exact sum of its three bf16 planes (3 x 8 significant bits; round to nearest even at each level)."""
    g = torch.Generator(device='cuda').manual_seed(0)
    W = torch.randn(1000, 96, device='cuda', generator=g) * torch.exp(4 * torch.randn(1000, 96, device='cuda', generator=g))
    W3, plane = split_planes(lib, W)
    n = W.numel()
    parts = [W3[i * plane:i * plane + n].double() for i in range(3)]
    assert torch.equal((parts[0] + parts[1] + parts[2]).float().reshape(W.shape), W)
    assert torch.equal(parts[0].float().reshape(W.shape), W.to(torch.bfloat16).float())


@pytest.mark.parametrize('M,N,K,flags', [(3584, 2752, 512, 5), (3584, 512, 1376, 0), (1000, 300, 96, 1), (45, 388, 32, 3), (3840, 2064, 512, 1),
                                         (130, 129, 2048, 0), (257, 64, 64, 2)])
def test_gemm_split_operands_is_fp32_accurate(lib, M, N, K, flags):
    """gemm_x3.hip: fp32 GEMM on the bf16 matrix cores (operands split into three bf16 numbers, six products, fp32 accumulate).
    It must be an fp32 GEMM, not a reduced-precision one: its error against float64 may not exceed the f32-input MFMA kernels'
    (measured: about 0.4x), every tile configuration gives the same bits, and all epilogues (folded RMSNorm, bias, SiLU, SiLU-GLU,
    residual) and partial tiles agree with float64."""
    g = torch.Generator(device='cuda').manual_seed(5)
    A = torch.randn(M, K, device='cuda', generator=g); W = torch.randn(N, K, device='cuda', generator=g) / K ** 0.5
    b = torch.randn(N, device='cuda', generator=g)
    swiglu = bool(flags & _lib.GEMM_SWIGLU)
    R = None if swiglu else torch.randn(M, N, device='cuda', generator=g)
    W3, plane = split_planes(lib, W)
    Nout = N // 2 if swiglu else N
    eps = 1.1920929e-07
    Ad, Wd = A.double(), W.double()
    X = Ad * torch.rsqrt(Ad.pow(2).mean(-1, keepdim=True) + eps) if flags & _lib.GEMM_RMS_ROWSCALE else Ad
    ref = X @ Wd.t() + b.double()
    if flags & _lib.GEMM_SILU:
        ref = torch.nn.functional.silu(ref)
    if swiglu:
        r = ref.reshape(M, N // 64, 2, 32)
        ref = (r[:, :, 0] * torch.nn.functional.silu(r[:, :, 1])).reshape(M, N // 2)
    if R is not None:
        ref = ref + R.double()
    native = torch.full((M, Nout), float('nan'), device='cuda')
    _lib.check(lib.d4_gemm(_lib.ptr(A), K, _lib.ptr(W), K, _lib.ptr(native), Nout, _lib.ptr(b), _lib.ptr(R), N, M, N, K, flags, eps, stream()))
    outs = []
    for cfg in range(6):
        o = torch.full((M, Nout), float('nan'), device='cuda')
        rc = lib.d4_gemm_split(_lib.ptr(A), K, _lib.ptr(W3), plane, K, _lib.ptr(o), Nout, _lib.ptr(b), _lib.ptr(R), N, M, N, K, flags, eps, cfg, stream())
        if rc != 0:                       # SiLU-GLU needs a wave tile of two 32-column sub-tiles: three of the six configurations
            assert swiglu and cfg in (0, 1, 5), lib.d4_last_error()
            continue
        outs.append(o)
    assert len(outs) >= 3
    for o in outs[1:]:
        assert torch.equal(o, outs[0])
    rms = lambda x: (x.double() - ref).pow(2).mean().sqrt().item()
    e_split, e_native = rms(outs[0]), rms(native)
    # (+ one output rounding, 2^-24 / sqrt(3) of the result's rms: few-row shapes run natively on the VALU kernel, whose K-split tree sums
    # leave little more than that)
    slack = 6e-8 * ref.pow(2).mean().sqrt().item()
    assert e_split <= 1.05 * e_native + slack, f'split-operand error {e_split:.3e} exceeds the f32-input MFMA error {e_native:.3e}'
    tol = 3e-6 * max(1., ref.abs().max().item()) * max(1., K / 256) ** 0.5
    assert (outs[0].double() - ref).abs().max().item() <= tol


@pytest.mark.parametrize('M,N,K,flags', [(3584, 2752, 512, 5), (3584, 512, 1376, 0), (3584, 1552, 512, 1), (3840, 2752, 512, 5), (1000, 300, 96, 1),
                                         (130, 129, 2048, 0), (257, 64, 64, 2), (3584, 512, 1376, 32)])
def test_gemm_split_persistent_form_is_bit_identical(lib, M, N, K, flags):
    """gemm_x3sk.hip (d4_gemm_split config 6, the form the engine uses): the persistent form of the 128 x 128 split-operand kernel — one workgroup per CU,
    whole rounds of tiles, then a last partial round of at most half the workgroups as 128 x 64 half tiles (SiLU-GLU value / gate waves meeting through
    LDS).  Nothing crosses between workgroups and every element keeps its k order: BIT-IDENTICAL to the plain kernel on every epilogue, ragged edges and
    the accumulating form; repeated launches bit-identical.  (The k-cut of the last round that rounds 3-4 carried was removed in round 5.)"""
    g = torch.Generator(device='cuda').manual_seed(11)
    A = torch.randn(M, K, device='cuda', generator=g); W = torch.randn(N, K, device='cuda', generator=g) / K ** 0.5
    b = torch.randn(N, device='cuda', generator=g)
    swiglu = bool(flags & _lib.GEMM_SWIGLU)
    accumulate = bool(flags & 32)
    R = None if swiglu else torch.randn(M, N, device='cuda', generator=g)
    W3, plane = split_planes(lib, W)
    Nout = N // 2 if swiglu else N
    eps = 1.1920929e-07
    Ad, Wd = A.double(), W.double()
    X = Ad * torch.rsqrt(Ad.pow(2).mean(-1, keepdim=True) + eps) if flags & _lib.GEMM_RMS_ROWSCALE else Ad
    ref = X @ Wd.t() + b.double()
    if flags & _lib.GEMM_SILU:
        ref = torch.nn.functional.silu(ref)
    if swiglu:
        r = ref.reshape(M, N // 64, 2, 32)
        ref = (r[:, :, 0] * torch.nn.functional.silu(r[:, :, 1])).reshape(M, N // 2)
    if R is not None:
        ref = ref + R.double()
    C0 = torch.randn(M, Nout, device='cuda', generator=g) if accumulate else None
    if accumulate:
        ref = ref + C0.double()
    outs = []
    for cfg in (4, 6, 6):
        o = C0.clone() if accumulate else torch.full((M, Nout), float('nan'), device='cuda')
        _lib.check(lib.d4_gemm_split(_lib.ptr(A), K, _lib.ptr(W3), plane, K, _lib.ptr(o), Nout, _lib.ptr(b), _lib.ptr(R), N, M, N, K, flags, eps, cfg, stream()))
        outs.append(o)
    torch.cuda.synchronize()
    plain, sk = outs[0], outs[1]
    assert torch.isfinite(sk).all()
    assert torch.equal(sk, plain) and torch.equal(outs[2], sk)
    tol = 3e-6 * max(1., ref.abs().max().item()) * max(1., K / 256) ** 0.5
    assert (sk.double() - ref).abs().max().item() <= tol


def test_gemm_split_operands_wide_exponent_spread(lib):
    """gemm_x3.hip with operands whose magnitudes span 2^120 inside one row (|a|, |w| from 2^-60 to 2^60): every output must stay within
    fp32 rounding of float64 RELATIVE TO sum_k |a_k w_k| (the dropped a2.w3 + a3.w2 + a3.w3 terms are <= 2^-24 of each product, and no
    plane of a normal fp32 number of this range is a bf16 subnormal), and no worse than the f32-input MFMA on the same operands."""
    M, N, K = 256, 256, 512
    g = torch.Generator(device='cuda').manual_seed(11)

    def spread(r, c):
        mag = torch.exp2(torch.randint(-60, 61, (r, c), device='cuda', generator=g).float())
        return torch.randn(r, c, device='cuda', generator=g) * mag
    A, W = spread(M, K), spread(N, K)
    W3, plane = split_planes(lib, W)
    ref = A.double() @ W.double().t()
    scale = A.double().abs() @ W.double().abs().t()
    native = torch.full((M, N), float('nan'), device='cuda'); o = torch.full((M, N), float('nan'), device='cuda')
    _lib.check(lib.d4_gemm(_lib.ptr(A), K, _lib.ptr(W), K, _lib.ptr(native), N, None, None, N, M, N, K, 0, 0., stream()))
    _lib.check(lib.d4_gemm_split(_lib.ptr(A), K, _lib.ptr(W3), plane, K, _lib.ptr(o), N, None, None, N, M, N, K, 0, 0., 0, stream()))
    assert torch.isfinite(o).all()
    e_split = ((o.double() - ref).abs() / scale).max().item()
    e_native = ((native.double() - ref).abs() / scale).max().item()
    assert e_split <= 6e-7, e_split                        # fp32 summation of K = 512 terms, relative to the absolute-value sum (measured 3.8e-7)
    assert e_split <= 1.25 * e_native + 6e-8, (e_split, e_native)


def test_gemm_split_operands_non_finite_and_subnormal_operands(lib):
    """The documented edge behaviour of gemm_x3.hip: an infinite or NaN operand makes every output that depends on
    it NON-FINITE (NaN where the f32-input MFMA gives +-inf: a2 = bf16(inf - inf)) and leaves every other output untouched — a failure
    upstream is never turned into a finite number; fp32 subnormal operands count as zero (error <= K * 2^-126 * max|other operand|)."""
    M, N, K = 64, 128, 256
    g = torch.Generator(device='cuda').manual_seed(12)
    A = torch.randn(M, K, device='cuda', generator=g); W = torch.randn(N, K, device='cuda', generator=g)
    A[3, 7] = float('inf'); A[5, 9] = float('nan'); W[11, 20] = float('-inf')
    W3, plane = split_planes(lib, W)
    o = torch.zeros(M, N, device='cuda'); native = torch.zeros(M, N, device='cuda')
    _lib.check(lib.d4_gemm_split(_lib.ptr(A), K, _lib.ptr(W3), plane, K, _lib.ptr(o), N, None, None, N, M, N, K, 0, 0., 0, stream()))
    _lib.check(lib.d4_gemm(_lib.ptr(A), K, _lib.ptr(W), K, _lib.ptr(native), N, None, None, N, M, N, K, 0, 0., stream()))
    bad = torch.zeros(M, N, dtype=torch.bool, device='cuda'); bad[3] = True; bad[5] = True; bad[:, 11] = True
    assert (~torch.isfinite(o[bad])).all() and (~torch.isfinite(native[bad])).all()
    assert torch.isfinite(o[~bad]).all()
    ref = A.double() @ W.double().t()
    assert torch.allclose(o[~bad].double(), ref[~bad], atol=3e-5, rtol=1e-5)
    # subnormals: a row of A that holds only fp32 subnormals contributes (at most) its exact tiny product
    A2 = torch.randn(M, K, device='cuda', generator=g); A2[0] = 1e-40
    o2 = torch.zeros(M, N, device='cuda')
    _lib.check(lib.d4_gemm_split(_lib.ptr(A2), K, _lib.ptr(W3), plane, K, _lib.ptr(o2), N, None, None, N, M, N, K, 0, 0., 0, stream()))
    keep = torch.ones(N, dtype=torch.bool, device='cuda'); keep[11] = False
    assert o2[0, keep].abs().max().item() <= K * 1e-40 * 8.


def split2_planes(lib, W, ld=None):
    """W [N][K] fp32 -> the two fp16 planes (hi, lo 2^11) + inverse row scales d4_gemm_split2 reads (d4_split_f16x2)."""
    N, K = W.shape
    ld = ld or K
    plane = (N * ld + 7) // 8 * 8
    W2 = torch.zeros(2 * plane, dtype=torch.float16, device='cuda')
    inv = torch.zeros(N, device='cuda')
    _lib.check(lib.d4_split_f16x2(_lib.ptr(W), _lib.ptr(W2), N, K, ld, plane, _lib.ptr(inv), stream()))
    return W2, plane, inv


def test_split_f16x2_planes_are_the_documented_decomposition(lib):
    """d4_split_f16x2: per row an exact power-of-two scale that puts the largest magnitude into [2^14, 2^15); hi = fp16(w s), lo = fp16((w s - hi) 2^11);
    hi + lo 2^-11 reproduces w s to 2^-22 relative (elements within 2^27 of the row maximum), and inv_scale undoes s exactly."""
    g = torch.Generator(device='cuda').manual_seed(0)
    W = torch.randn(300, 96, device='cuda', generator=g) * torch.exp2(torch.randint(-8, 9, (300, 96), device='cuda', generator=g).float()) * torch.exp2(
        torch.randint(-30, 31, (300, 1), device='cuda', generator=g).float())
    W2, plane, inv = split2_planes(lib, W)
    n = W.numel()
    hi, lo = (W2[i * plane:i * plane + n].double().reshape(W.shape) for i in range(2))
    s = 1. / inv.double()[:, None]
    mx = (W.double().abs() * s).amax(dim=1)
    assert (mx >= 2. ** 14).all() and (mx < 2. ** 15).all()
    assert torch.equal(torch.log2(inv.double()).round(), torch.log2(inv.double()))                     # powers of two
    assert torch.equal(hi.float(), (W.double() * s).float().to(torch.float16).float())
    x = W.double() * s
    assert ((hi + lo / 2048. - x).abs() <= 2. ** -21 * x.abs() + 2. ** -35).all()


@pytest.mark.parametrize('M,N,K,flags', [(3584, 2752, 512, 5), (3584, 512, 1376, 0), (1000, 300, 96, 1), (45, 388, 32, 3), (3840, 2064, 512, 1),
                                         (130, 129, 2048, 0), (257, 64, 64, 2)])
def test_gemm_h2_is_fp32_accurate(lib, M, N, K, flags):
    """gemm_h2.hip: fp32 GEMM on the fp16 matrix cores (operands as two fp16 planes under exact power-of-two row scales, three products, fp32
    accumulate) under EXACTLY the criteria the bf16x3 / six-product scheme is held to (test_gemm_split_operands_is_fp32_accurate, same shapes,
    same bounds): its error against float64 may not exceed the f32-input MFMA kernels', every tile configuration gives the same bits, and all
    epilogues (folded RMSNorm, bias, SiLU, SiLU-GLU, residual) and partial tiles agree with float64."""
    g = torch.Generator(device='cuda').manual_seed(5)
    A = torch.randn(M, K, device='cuda', generator=g); W = torch.randn(N, K, device='cuda', generator=g) / K ** 0.5
    b = torch.randn(N, device='cuda', generator=g)
    swiglu = bool(flags & _lib.GEMM_SWIGLU)
    R = None if swiglu else torch.randn(M, N, device='cuda', generator=g)
    W2, plane, inv = split2_planes(lib, W)
    Nout = N // 2 if swiglu else N
    eps = 1.1920929e-07
    Ad, Wd = A.double(), W.double()
    X = Ad * torch.rsqrt(Ad.pow(2).mean(-1, keepdim=True) + eps) if flags & _lib.GEMM_RMS_ROWSCALE else Ad
    ref = X @ Wd.t() + b.double()
    if flags & _lib.GEMM_SILU:
        ref = torch.nn.functional.silu(ref)
    if swiglu:
        r = ref.reshape(M, N // 64, 2, 32)
        ref = (r[:, :, 0] * torch.nn.functional.silu(r[:, :, 1])).reshape(M, N // 2)
    if R is not None:
        ref = ref + R.double()
    native = torch.full((M, Nout), float('nan'), device='cuda')
    _lib.check(lib.d4_gemm(_lib.ptr(A), K, _lib.ptr(W), K, _lib.ptr(native), Nout, _lib.ptr(b), _lib.ptr(R), N, M, N, K, flags, eps, stream()))
    outs = []
    for cfg in range(7):
        o = torch.full((M, Nout), float('nan'), device='cuda')
        rc = lib.d4_gemm_split2(_lib.ptr(A), K, _lib.ptr(W2), plane, K, _lib.ptr(inv), _lib.ptr(o), Nout, _lib.ptr(b), _lib.ptr(R), N, M, N, K, flags, eps, cfg, None, stream())
        if rc != 0:                       # SiLU-GLU needs a wave tile of two 32-column sub-tiles: three of the seven configurations cannot
            assert swiglu and cfg in (0, 1, 5), lib.d4_last_error()
            continue
        outs.append(o)
    assert len(outs) >= 4
    for o in outs[1:]:
        assert torch.equal(o, outs[0])
    rms = lambda x: (x.double() - ref).pow(2).mean().sqrt().item()
    e_split, e_native = rms(outs[0]), rms(native)
    slack = 6e-8 * ref.pow(2).mean().sqrt().item()
    assert e_split <= 1.05 * e_native + slack, f'split-operand error {e_split:.3e} exceeds the f32-input MFMA error {e_native:.3e}'
    tol = 3e-6 * max(1., ref.abs().max().item()) * max(1., K / 256) ** 0.5
    assert (outs[0].double() - ref).abs().max().item() <= tol
    # the row exponents handed in by a producer give the same bits as the kernel's own prologue
    ex = torch.clamp(14 - torch.floor(torch.log2(A.abs().amax(dim=1).clamp(min=1e-45))), max=126).to(torch.int32)
    o = torch.full((M, Nout), float('nan'), device='cuda')
    _lib.check(lib.d4_gemm_split2(_lib.ptr(A), K, _lib.ptr(W2), plane, K, _lib.ptr(inv), _lib.ptr(o), Nout, _lib.ptr(b), _lib.ptr(R), N, M, N, K, flags, eps, 2, _lib.ptr(ex), stream()))
    assert torch.equal(o, outs[0])


def test_gemm_h2_wide_exponent_spread_is_where_it_is_not_fp32(lib):
    """gemm_h2.hip on the operands of test_gemm_split_operands_wide_exponent_spread (magnitudes spanning 2^120 inside one row: every output is
    essentially ONE product).  This is the criterion the fp16x2 scheme does NOT meet, and the reason it is an opt-in mode and not the default fp32
    path: a product of two 23-bit operand images with the lo.lo term dropped carries up to 2^-21 relative error where an fp32 product has 2^-24.
    Pinned here as measured on MI355X: it stays inside the ABSOLUTE bound of that test (<= 6e-7 of sum |a w|; measured 5.5e-7), every output is
    finite, but it is 1.2-2x the f32-input MFMA's error (measured 3.1e-7; the bf16x3 scheme 3.8e-7) — outside that test's `<= 1.25 native + 6e-8`.
    Four and five products (both images complete; W with a third plane) measure 5.0e-7 and 4.95e-7: the fp16 MFMA's own accumulation is the rest."""
    M, N, K = 256, 256, 512
    g = torch.Generator(device='cuda').manual_seed(11)

    def spread(r, c):
        mag = torch.exp2(torch.randint(-60, 61, (r, c), device='cuda', generator=g).float())
        return torch.randn(r, c, device='cuda', generator=g) * mag
    A, W = spread(M, K), spread(N, K)
    W2, plane, inv = split2_planes(lib, W)
    ref = A.double() @ W.double().t()
    scale = A.double().abs() @ W.double().abs().t()
    native = torch.full((M, N), float('nan'), device='cuda'); o = torch.full((M, N), float('nan'), device='cuda')
    _lib.check(lib.d4_gemm(_lib.ptr(A), K, _lib.ptr(W), K, _lib.ptr(native), N, None, None, N, M, N, K, 0, 0., stream()))
    _lib.check(lib.d4_gemm_split2(_lib.ptr(A), K, _lib.ptr(W2), plane, K, _lib.ptr(inv), _lib.ptr(o), N, None, None, N, M, N, K, 0, 0., 0, None, stream()))
    assert torch.isfinite(o).all()
    e_split = ((o.double() - ref).abs() / scale).max().item()
    e_native = ((native.double() - ref).abs() / scale).max().item()
    assert e_split <= 6e-7, e_split
    assert e_native < e_split <= 2. * e_native, (e_split, e_native)            # worse than fp32 on single products, by less than one bit


def test_gemm_h2_non_finite_and_subnormal_operands(lib):
    """gemm_h2.hip under the criteria of test_gemm_split_operands_non_finite_and_subnormal_operands, unchanged: an infinite or NaN operand makes every
    output that depends on it NON-FINITE and leaves every other output untouched; fp32 subnormal operands contribute at most their exact tiny product."""
    M, N, K = 64, 128, 256
    g = torch.Generator(device='cuda').manual_seed(12)
    A = torch.randn(M, K, device='cuda', generator=g); W = torch.randn(N, K, device='cuda', generator=g)
    A[3, 7] = float('inf'); A[5, 9] = float('nan'); W[11, 20] = float('-inf')
    W2, plane, inv = split2_planes(lib, W)
    o = torch.zeros(M, N, device='cuda'); native = torch.zeros(M, N, device='cuda')
    _lib.check(lib.d4_gemm_split2(_lib.ptr(A), K, _lib.ptr(W2), plane, K, _lib.ptr(inv), _lib.ptr(o), N, None, None, N, M, N, K, 0, 0., 0, None, stream()))
    _lib.check(lib.d4_gemm(_lib.ptr(A), K, _lib.ptr(W), K, _lib.ptr(native), N, None, None, N, M, N, K, 0, 0., stream()))
    bad = torch.zeros(M, N, dtype=torch.bool, device='cuda'); bad[3] = True; bad[5] = True; bad[:, 11] = True
    assert (~torch.isfinite(o[bad])).all() and (~torch.isfinite(native[bad])).all()
    assert torch.isfinite(o[~bad]).all()
    ref = A.double() @ W.double().t()
    assert torch.allclose(o[~bad].double(), ref[~bad], atol=3e-5, rtol=1e-5)
    A2 = torch.randn(M, K, device='cuda', generator=g); A2[0] = 1e-40
    o2 = torch.zeros(M, N, device='cuda')
    _lib.check(lib.d4_gemm_split2(_lib.ptr(A2), K, _lib.ptr(W2), plane, K, _lib.ptr(inv), _lib.ptr(o2), N, None, None, N, M, N, K, 0, 0., 0, None, stream()))
    keep = torch.ones(N, dtype=torch.bool, device='cuda'); keep[11] = False
    assert o2[0, keep].abs().max().item() <= K * 1e-40 * 8.
    assert torch.allclose(o2[1:][:, keep].double(), (A2.double() @ W.double().t())[1:][:, keep], atol=3e-5, rtol=1e-5)


@pytest.mark.parametrize('M1,M2,N1,N2,flags', [(3584, 3 * 3584, 256, 256, 1), (3584, 11 * 3584, 256, 256, 1), (1000, 4097, 272, 512, 0), (40, 5000, 256, 256, 1)])
def test_gemm_pair_is_bit_identical_to_two_launches(lib, M1, M2, N1, N2, flags):
    """gemm2_pair_kernel (the attention pool's query + key projections in one grid): every output bit equals the separate launches'."""
    K = 512
    g = torch.Generator(device='cuda').manual_seed(21)
    A1 = torch.randn(M1, K, device='cuda', generator=g); A2 = torch.randn(M2, K, device='cuda', generator=g)
    W1 = torch.randn(N1, K, device='cuda', generator=g) / K ** 0.5; W2 = torch.randn(N2, K, device='cuda', generator=g) / K ** 0.5
    eps = 1.1920929e-07
    r1 = torch.full((M1, N1), float('nan'), device='cuda'); r2 = torch.full((M2, N2), float('nan'), device='cuda')
    _lib.check(lib.d4_gemm(_lib.ptr(A1), K, _lib.ptr(W1), K, _lib.ptr(r1), N1, None, None, N1, M1, N1, K, flags, eps, stream()))
    _lib.check(lib.d4_gemm(_lib.ptr(A2), K, _lib.ptr(W2), K, _lib.ptr(r2), N2, None, None, N2, M2, N2, K, flags, eps, stream()))
    o1 = torch.full((M1, N1), float('nan'), device='cuda'); o2 = torch.full((M2, N2), float('nan'), device='cuda')
    _lib.check(lib.d4_gemm_pair(_lib.ptr(A1), K, _lib.ptr(W1), _lib.ptr(o1), N1, M1, N1, _lib.ptr(A2), K, _lib.ptr(W2), _lib.ptr(o2), N2, M2, N2, K, flags, eps, stream()))
    assert torch.equal(o1, r1) and torch.equal(o2, r2)
    X = A2.double() * torch.rsqrt(A2.double().pow(2).mean(-1, keepdim=True) + eps) if flags & 1 else A2.double()
    assert torch.allclose(o2.double(), X @ W2.double().t(), atol=2e-5, rtol=1e-5)


@pytest.mark.parametrize('M,N,K,lda,ldb,ldc,slices', [(2736, 512, 3840, 2736, 512, 512, 0), (512, 1368, 3840, 512, 1368, 1368, 0), (260, 512, 1000, 264, 520, 512, 0),
                                                      (4, 8, 5, 4, 8, 8, 0), (68, 132, 17, 68, 132, 140, 1), (512, 512, 4099, 512, 512, 512, 5),
                                                      (64, 64, 16, 64, 64, 64, 0), (512, 32, 8192, 512, 32, 32, 0)])
def test_weight_gradient_gemm_vs_float64_and_deterministic(lib, M, N, K, lda, ldb, ldc, slices):
    """d4_gemm_tn: C = A^T B with the contraction over the operands' rows (the gradient of a Linear's weight), read straight into the MFMA
    layout; ragged tiles, row counts that are not multiples of the 16-row step or the slice, padded leading dimensions, forced and ruled slice
    counts.  Error against float64 <= 4e-7 of sum |a b| (fp32 accumulation), columns outside [M][N] untouched, two runs bit-identical."""
    g = torch.Generator(device='cuda').manual_seed(5)
    A = torch.randn(K, lda, device='cuda', generator=g); B = torch.randn(K, ldb, device='cuda', generator=g)
    part = torch.empty(8 << 20, device='cuda')
    outs = []
    for _ in range(2):
        C_ = torch.full((M, ldc), float('nan'), device='cuda')
        _lib.check(lib.d4_gemm_tn(_lib.ptr(A), lda, _lib.ptr(B), ldb, _lib.ptr(C_), ldc, M, N, K, _lib.ptr(part), part.numel(), 0, slices, stream()))
        outs.append(C_)
    assert torch.equal(outs[0][:, :N], outs[1][:, :N])
    assert ldc == N or torch.isnan(outs[0][:, N:]).all()
    ref = A[:, :M].double().t() @ B[:, :N].double()
    scale = A[:, :M].double().abs().t() @ B[:, :N].double().abs()
    assert ((outs[0][:, :N].double() - ref).abs() / scale.clamp(min=1e-30)).max().item() <= 4e-7
    # no scratch: one slice, same tolerance
    C1 = torch.full((M, ldc), float('nan'), device='cuda')
    _lib.check(lib.d4_gemm_tn(_lib.ptr(A), lda, _lib.ptr(B), ldb, _lib.ptr(C1), ldc, M, N, K, None, 0, 0, 0, stream()))
    assert ((C1[:, :N].double() - ref).abs() / scale.clamp(min=1e-30)).max().item() <= 4e-7


def test_weight_gradient_gemm_rejects_widths_that_are_not_multiples_of_4(lib):
    A = torch.randn(16, 6, device='cuda'); C_ = torch.empty(6, 6, device='cuda')
    with pytest.raises(_lib.D4Error, match='multiples of 4'):
        _lib.check(lib.d4_gemm_tn(_lib.ptr(A), 6, _lib.ptr(A), 6, _lib.ptr(C_), 6, 6, 6, 16, None, 0, 0, 0, stream()))


def test_gemm_rejects_misaligned_operands(lib):
    A = torch.randn(8, 34, device='cuda')
    with pytest.raises(_lib.D4Error, match='multiples of 4'):
        _lib.check(lib.d4_gemm(_lib.ptr(A), 34, _lib.ptr(A), 34, _lib.ptr(A), 8, None, None, 0, 8, 8, 34, 0, 0., stream()))


@pytest.mark.parametrize('rows,D', [(4, 64), (7, 512), (256, 2048), (5, 8), (0, 64)])
def test_rmsnorm(lib, rows, D):
    x = torch.randn(rows, D, device='cuda'); g = torch.randn(D, device='cuda'); y = torch.zeros_like(x)
    _lib.check(lib.d4_rmsnorm(_lib.ptr(x), D, _lib.ptr(g), _lib.ptr(y), D, rows, D, 1.1920929e-07, stream()))
    ref = torch.nn.functional.rms_norm(x, (D,), g, eps=None)
    assert torch.allclose(y, ref, atol=2e-6, rtol=1e-5)


@pytest.mark.parametrize('bins', [255, 63, 20])
def test_hl_gauss_bins_to_scalar(lib, bins):
    logits = torch.randn(300, bins, device='cuda') * 3
    support = torch.linspace(-20., 20., bins + 1, device='cuda')
    centers = ((support[:-1] + support[1:]) / 2).contiguous()
    out = torch.empty(300, device='cuda')
    _lib.check(lib.d4_hl_gauss_scalar(_lib.ptr(logits), bins, _lib.ptr(centers), _lib.ptr(out), 300, bins, stream()))
    ref = (logits.softmax(-1) * centers).sum(-1)
    assert torch.allclose(out, ref, atol=1e-5)
    assert (out >= -20).all() and (out <= 20).all()           # reference test_hl_gauss_reward_encoder: values within range


def test_gae_matches_the_reference_fixture(lib):
    from util import load_golden, t
    g = load_golden('learn.npz')
    r, v = t(g['gae_rewards']).cuda(), t(g['gae_values']).cuda()
    lens = t(g['gae_lens']).cuda(); tr = t(g['gae_trunc']).to(torch.uint8).cuda(); te = t(g['gae_term']).to(torch.uint8).cuda()
    out = torch.empty_like(r)
    _lib.check(lib.d4_gae(_lib.ptr(r), _lib.ptr(v), _lib.ptr(lens), _lib.ptr(tr), _lib.ptr(te), 0.997, 0.95, r.shape[0], r.shape[1],
                          _lib.ptr(out), stream()))
    assert torch.allclose(out.cpu(), t(g['gae_returns']), atol=1e-6)


def test_torch_library_rmsnorm_and_linear_are_differentiable(lib):
    """torch.ops.d4hip.rmsnorm / linear carry registered autograd (d4_rmsnorm_backward, d4_gemm with transposed operands): gradients match
    torch's own autograd of F.rms_norm / F.linear; the fused-epilogue forms of `linear` refuse to be differentiated; flow_euler_step == the formula."""
    g = torch.Generator(device='cuda').manual_seed(3)
    x = torch.randn(5, 37, 64, device='cuda', generator=g); w = torch.randn(64, device='cuda', generator=g)
    W = torch.randn(96, 64, device='cuda', generator=g) / 8; b = torch.randn(96, device='cuda', generator=g)
    dy = torch.randn(5, 37, 96, device='cuda', generator=g)
    eps = 1.1920929e-07
    leaves = [t.clone().requires_grad_() for t in (x, w, W, b)]
    y = torch.ops.d4hip.linear(torch.ops.d4hip.rmsnorm(leaves[0], leaves[1], eps), leaves[2], leaves[3], None, 0, 0.)
    y.backward(dy)
    ref_leaves = [t.clone().double().requires_grad_() for t in (x, w, W, b)]
    yr = torch.nn.functional.linear(torch.nn.functional.rms_norm(ref_leaves[0], (64,), ref_leaves[1], eps), ref_leaves[2], ref_leaves[3])
    yr.backward(dy.double())
    assert torch.allclose(y.double(), yr, atol=1e-4, rtol=1e-4)
    for a, r, name in zip(leaves, ref_leaves, ('dx', 'd gamma', 'dW', 'db')):
        scale = r.grad.abs().max().item()
        assert (a.grad.double() - r.grad).abs().max().item() <= 2e-5 * max(scale, 1.), name
    xs = x.clone().requires_grad_()
    with pytest.raises(_lib.D4Error, match='differentiable only without'):
        torch.ops.d4hip.linear(xs, W, b, None, _lib.GEMM_SILU, 0.)
    pred = torch.randn_like(x)
    out = torch.ops.d4hip.flow_euler_step(x, pred, 0.75, 0.25)
    assert torch.allclose(out, x + (pred - x) / 0.75 * 0.25, atol=1e-5)


def test_torch_library_ops_run_the_hip_kernels_and_trace_without_graph_breaks(lib):
    x = torch.randn(37, 64, device='cuda'); w = torch.randn(64, device='cuda'); W = torch.randn(96, 64, device='cuda'); b = torch.randn(96, device='cuda')

    def f(x):
        h = torch.ops.d4hip.rmsnorm(x, w, 1.1920929e-07)
        return torch.ops.d4hip.linear(h, W, b, None, _lib.GEMM_SILU, 0.)

    ref = torch.nn.functional.silu(torch.nn.functional.rms_norm(x, (64,), w, eps=None) @ W.t() + b)
    assert torch.allclose(f(x), ref, atol=1e-4, rtol=1e-4)
    compiled = torch.compile(f, backend='eager', fullgraph=True)       # fullgraph: the custom ops are traceable (no graph break)
    assert torch.allclose(compiled(x), ref, atol=1e-4, rtol=1e-4)
    r = torch.randn(4, 7, device='cuda'); v = torch.randn(4, 7, device='cuda')
    lens = torch.tensor([7, 3, 5, 1], device='cuda'); tr = torch.tensor([True, False, False, True], device='cuda')
    out = torch.ops.d4hip.gae(r, v, lens, tr, ~tr, 0.997, 0.95)
    ref2 = torch.empty_like(r)
    tr8, te8 = tr.to(torch.uint8), (~tr).to(torch.uint8)               # (kept alive: a temporary would be freed before the launch)
    _lib.check(lib.d4_gae(_lib.ptr(r), _lib.ptr(v), _lib.ptr(lens), _lib.ptr(tr8), _lib.ptr(te8), 0.997, 0.95, 4, 7, _lib.ptr(ref2), stream()))
    assert torch.equal(out, ref2)


@pytest.mark.parametrize('sizes,temperature', [((4,), 1.), ((3, 5, 2), 0.7), ((17,), 1.3)])
def test_categorical_sample_logp_op_vs_oracle(lib, sizes, temperature):
    """torch.ops.d4hip.categorical_sample_logp (d4_categorical_sample_logp): MultiCategorical.sample + log_prob, D4:485-497, 1374-1376, 1422-1423, against
    oracle/restate.py on injected uniforms — indices bit-exact (rows whose top-2 score margin is below 1e-4 are excluded, none in practice)."""
    from oracle import restate
    cfg = restate.Config(dim=64, dim_latent=8, num_latent_tokens=4, num_discrete_actions=tuple(sizes))
    g = torch.Generator().manual_seed(17)
    logits = torch.randn(6, 37, sum(sizes), generator=g) * 2
    u = torch.rand(6, 37, sum(sizes), generator=g)
    ref_a = restate.sample_discrete(cfg, logits, u, temperature)
    ref_lp = restate.discrete_log_probs(cfg, logits, ref_a)
    acts, lps = torch.ops.d4hip.categorical_sample_logp(logits.cuda(), u.cuda(), list(sizes), temperature)
    sc = logits / temperature - torch.log((-torch.log(u.clamp(min=1e-20))).clamp(min=1e-20))
    ok, o = torch.ones(6, 37, dtype=torch.bool), 0
    for n in sizes:
        top = sc[..., o:o + n].topk(min(2, n), dim=-1).values
        if n > 1:
            ok &= (top[..., 0] - top[..., 1]) > 1e-4
        o += n
    assert ok.float().mean() > 0.99
    assert torch.equal(acts.cpu()[ok], ref_a[ok])
    assert torch.allclose(lps.cpu()[ok], ref_lp[ok], atol=1e-5)
    compiled = torch.compile(lambda l, uu: torch.ops.d4hip.categorical_sample_logp(l, uu, list(sizes), temperature), backend='eager', fullgraph=True)
    a2, _ = compiled(logits.cuda(), u.cuda())
    assert torch.equal(a2, acts)


@pytest.mark.parametrize('two_hot', [False, True])
def test_hl_gauss_ce_op_vs_oracle(lib, two_hot):
    """torch.ops.d4hip.hl_gauss_ce (d4_hl_gauss_ce): the value branch's loss D4:6254-6295 — HL-Gauss / symexp two-hot targets, cross entropy, masked mean —
    and its gradient, against autograd of oracle/restate.py's restatement."""
    from oracle import restate
    bins, vrange = 63, (-20., 20.)
    cfg = restate.Config(dim=64, dim_latent=8, num_latent_tokens=4, reward_encoder_type='symexp_two_hot' if two_hot else 'hl_gauss')
    g = torch.Generator().manual_seed(23)
    logits = torch.randn(5, 40, bins, generator=g) * 2
    returns = torch.randn(5, 40, generator=g) * (3. if two_hot else 9.)
    mask = torch.rand(5, 40, generator=g) > 0.3
    lr = logits.clone().requires_grad_()
    probs = restate.symexp_two_hot(returns, vrange, bins) if two_hot else restate.hl_gauss_to_probs(cfg, returns, vrange, bins)
    ref = (-(probs * lr.log_softmax(dim=-1)).sum(dim=-1))[mask].mean()
    ref.backward()
    if two_hot:
        support, vmin, vmax = restate.symexp_bin_values(vrange, bins), float(restate.symexp_bin_values(vrange, bins)[0]), float(restate.symexp_bin_values(vrange, bins)[-1])
    else:
        support, vmin, vmax = restate.hl_gauss_centers(vrange, bins)[0], vrange[0], vrange[1]
    sigma = cfg.hl_gauss_sigma_to_bin_ratio * (vrange[1] - vrange[0]) / bins
    lg = logits.cuda().requires_grad_()
    loss, _ = torch.ops.d4hip.hl_gauss_ce(lg, returns.cuda(), mask.cuda(), support.cuda(), vmin, vmax, sigma, cfg.hl_gauss_eps, two_hot)
    (2. * loss).backward()
    assert abs(loss.item() - ref.item()) < 1e-5 * max(1., abs(ref.item()))
    assert torch.allclose(lg.grad.cpu(), 2. * lr.grad, atol=1e-6, rtol=1e-4)
    # no mask = every row
    loss_all, _ = torch.ops.d4hip.hl_gauss_ce(logits.cuda(), returns.cuda(), None, support.cuda(), vmin, vmax, sigma, cfg.hl_gauss_eps, two_hot)
    ref_all = (-(probs * logits.log_softmax(dim=-1)).sum(dim=-1)).mean()
    assert abs(loss_all.item() - ref_all.item()) < 1e-5 * max(1., abs(ref_all.item()))


@pytest.mark.parametrize('objective,sizes', [(0, (4,)), (1, (4,)), (0, (3, 5, 2))])
def test_ppo_policy_loss_op_vs_oracle(lib, objective, sizes):
    """torch.ops.d4hip.ppo_policy_loss (d4_ppo_policy_loss): the policy branch's loss D4:6077-6242 for discrete actions — joint log-prob of the stored
    actions, PPO clipped surrogate / SPO against the behaviour log-probs, entropy bonus, masked mean — and its gradient, against autograd of
    oracle/restate.py's restatement (discrete_log_probs, masked_mean and the surrogate lines of learn_losses)."""
    from oracle import restate
    cfg = restate.Config(dim=64, dim_latent=8, num_latent_tokens=4, num_discrete_actions=tuple(sizes))
    g = torch.Generator().manual_seed(31)
    logits = torch.randn(5, 40, sum(sizes), generator=g) * 1.5
    actions = torch.stack([torch.randint(0, n, (5, 40), generator=g) for n in sizes], dim=-1)
    old_lp = restate.discrete_log_probs(cfg, logits + 0.3 * torch.randn(logits.shape, generator=g), actions)       # a nearby behaviour policy
    adv = torch.randn(5, 40, generator=g)
    mask = torch.rand(5, 40, generator=g) > 0.25
    clip, ent_w = cfg.ppo_eps_clip, 0.02
    lr = logits.clone().requires_grad_()
    lps, ents = restate.discrete_log_probs(cfg, lr, actions, with_entropy=True)
    lp, old = lps.sum(dim=-1), old_lp.sum(dim=-1)
    ratio = (lp - old).exp()
    if objective == 0:
        pl = -torch.min(ratio * adv, ratio.clamp(1. - clip, 1. + clip) * adv)
    else:
        pl = -(ratio * adv - (adv.abs() * (ratio - 1.).square()) / (2 * clip))
    ref = restate.masked_mean(pl, mask) + ent_w * restate.masked_mean(-ents.sum(dim=-1), mask)
    ref.backward()
    lg = logits.cuda().requires_grad_()
    loss, _ = torch.ops.d4hip.ppo_policy_loss(lg, actions.cuda(), old_lp.cuda(), adv.cuda(), mask.cuda(), list(sizes), objective, clip, ent_w)
    (3. * loss).backward()
    assert abs(loss.item() - ref.item()) < 1e-5 * max(1., abs(ref.item()))
    assert torch.allclose(lg.grad.cpu(), 3. * lr.grad, atol=1e-6, rtol=1e-3)
    compiled = torch.compile(lambda l: torch.ops.d4hip.ppo_policy_loss(l, actions.cuda(), old_lp.cuda(), adv.cuda(), None, list(sizes), objective, clip, ent_w)[0],
                             backend='eager', fullgraph=True)
    ref_all = pl.mean() + ent_w * (-ents.sum(dim=-1)).mean()
    assert abs(compiled(logits.cuda()).item() - ref_all.item()) < 1e-5 * max(1., abs(ref_all.item()))


def test_attn_pool_op_vs_oracle(lib):
    """dreamer4_amd.ops.attn_pool — Residual(AttentionPool), D4:2143-2177 + 1869, as a composition of the dispatcher op attn_block_cross — against
    oracle/restate.py's attention_pool on random weights, forward and the gradient with respect to the input tokens."""
    from oracle import restate
    from dreamer4_amd import ops
    D, L, rows, hp = 64, 5, 33, 256
    g = torch.Generator().manual_seed(41)
    r = lambda *s, sc=1.: torch.randn(*s, generator=g) * sc
    pre = 'p.'
    W = {pre + 'fn.attn.norm.weight': 1. + r(D, sc=0.1), pre + 'fn.attn.norm_context.weight': 1. + r(D, sc=0.1), pre + 'fn.attn.to_q.weight': r(hp, D, sc=D ** -0.5),
         pre + 'fn.attn.to_k.weight': r(hp, D, sc=D ** -0.5), pre + 'fn.attn.to_v.weight': r(hp, D, sc=D ** -0.5), pre + 'fn.attn.to_out.weight': r(D, hp, sc=hp ** -0.5),
         pre + 'fn.attn.to_gates.0.weight': r(4, D, sc=D ** -0.5), pre + 'fn.attn.k_heads_rmsnorm.gamma': r(4, 64, sc=0.2)}
    cfg = restate.Config(dim=D, dim_latent=8, num_latent_tokens=4)
    hid = r(L, rows, D)
    x = hid[-1].clone().requires_grad_()
    ref = restate.attention_pool(cfg, W, pre, x[None], [h[None] for h in hid])[0]
    ref.square().sum().backward()
    xg = hid[-1].cuda().requires_grad_()
    k = lambda n: W[pre + 'fn.attn.' + n].cuda()
    out = ops.attn_pool(xg, hid.cuda(), k('norm.weight'), k('norm_context.weight'), k('to_q.weight'), k('to_k.weight'), k('to_v.weight'), k('to_out.weight'),
                        k('to_gates.0.weight'), k('k_heads_rmsnorm.gamma'))
    out.square().sum().backward()
    assert torch.allclose(out.detach().cpu(), ref.detach(), atol=2e-5, rtol=1e-4)
    assert torch.allclose(xg.grad.cpu(), x.grad, atol=2e-4 * float(x.grad.abs().max()), rtol=1e-3)
