"""The ONE table of `D4_*` environment switches (csrc/*.hip read them with getenv at first use, a few Python modules with os.environ).

kind:
  'experiment'  an A/B switch of a measured decision.  The default IS the product; tests and bench.py refuse to run with one of these set
                (tests/conftest.py, bench.py), so the defaults the GPU tests run under are the defaults the bench runs under.  A test that needs
                the other arm sets it itself (monkeypatch) for one engine.
  'mode'        a legitimate runtime mode a deployment may choose (memory / traceability / multi-rank behaviour), numerically pinned by tests.
  'io'          paths and logging; no effect on results or kernel choice.
`tests/test_host.py::test_every_environment_switch_is_in_the_knob_table` greps the sources and fails on a switch missing here."""

KNOBS = {
    # --- launch mechanism / engine
    'D4_GRAPH_MAX_ROWS': ('4096', 'experiment', 'decode frames of <= this many token rows are replayed from hipGraphs (0: always enqueue eagerly); same kernels either way'),
    'D4_FRAME_FUSED': ('1', 'experiment', 'per-frame fused block tails (frame_fused.hip): 0 off, 1 on where the rule applies, 2 tails only'),
    'D4_ATTN_OUT_COLS': ('1', 'experiment', 'few frames (decode): within-frame attention recomputed inside the column-split output projection, one launch (0: two)'),
    'D4_BF16_ACT': ('1', 'experiment', 'bf16 engine: bf16 activation images between producer and consumer (0: fp32 activations into every bf16 GEMM, the round-2 form)'),
    'D4_BF16A_PAIR': ('1', 'experiment', 'bf16 engine: the attention pool query and key projections in one grid (gemm_bf16a_pair_kernel)'),
    'D4_BF16A_GROUPED': ('1', 'experiment', 'gemm_bf16a: grouped (L2-friendly) tile order (0: row-major)'),
    # --- fp32 GEMM dispatch
    'D4_GEMM_V2': ('1', 'experiment', 'LDS-DMA fp32 family gemm2.hip for K % 32 == 0 (0: register-staged family only)'),
    'D4_GEMM_X3': ('1', 'experiment', 'split-operand fp32 family on the bf16 matrix cores: 0 never, 1 by the shape rule, 2 every applicable call'),
    'D4_GEMM_X3SK': ('2', 'experiment', 'persistent split-operand form: 0 never, 1 half-tile calls only, 2 every call of >= 1024 rows, 3 also long-K output projections'),
    'D4_GEMM_PAIR': ('1', 'experiment', 'two independent projections in one grid (gemm2_pair_kernel)'),
    'D4_GEMM_SKINNY': ('1', 'experiment', 'few-row GEMM kernel (gemm_skinny.hip)'),
    'D4_SKINNY_MAXM': ('32', 'experiment', 'few-row kernel: row limit'),
    'D4_SKINNY_TILES': ('64', 'experiment', 'few-row kernel: taken below this many 64 x 64 tiles'),
    'D4_SKINNY_NW': ('0', 'experiment', 'few-row kernel: force the waves per block'),
    'D4_SKINNY_PAIR': ('1', 'experiment', 'two few-row projections in one launch'),
    'D4_GEMM_TN': ('1', 'experiment', 'weight gradients on gemm_tn_kernel (0: transposed-operand form of gemm_kernel)'),
    'D4_GEMM_DX_T': ('1', 'experiment', 'input gradients through a transposed weight image'),
    # --- glue kernels
    'D4_SPACE_ATTN_MFMA': ('1', 'experiment', 'within-frame / small cross attention on the matrix pipe (attn_mfma_kernel)'),
    'D4_TIME_ATTN_FEW': ('1', 'experiment', 'four-heads-per-wave time attention for histories of <= 16 keys'),
    'D4_TIME_ATTN_FUSED_APPEND': ('1', 'experiment', 'cached decode of one frame: KV append and time attention in one launch (0: two launches)'),
    'D4_TIME_ATTN_LEGACY': ('unset', 'experiment', 'one-key-per-reduction time attention'),
    'D4_KV_APPEND_LEGACY': ('unset', 'experiment', 'one-head-per-wave KV append'),
    'D4_POOL_MIX_ROWS': ('1', 'experiment', 'block-per-row pool mix for few rows'),
    'D4_POOL_MIX_ROWS_MAX': ('2048', 'experiment', 'row limit of the block-per-row pool mix'),
    # --- modes
    'D4_GEMM_AUTOTUNE': ('1', 'mode', '1: time a new GEMM shape at first use; 0: static choice; strict: fail on a shape missing from the shipped tile table (multi-rank bench)'),
    'D4_FORCE_PG': ('0', 'mode', '1: create the process group and issue every collective in a one-rank world (RCCL path on one GPU)'),
    'D4_TRUNK_SAVE_FORWARD': ('1', 'mode', 'training blocks keep their forward workspace (0: recompute in the backward, less memory)'),
    'D4_TRUNK_DISPATCHER': ('0', 'mode', '1: training blocks go through torch.ops.d4hip.* (torch.compile-traceable) instead of autograd.Function'),
    'D4_DP_BACKEND': ('gloo', 'mode', 'tests/dp_gpu_worker.py: process-group backend'),
    'D4_BENCH_BACKEND': ('nccl', 'mode', 'bench.py: process-group backend (gloo: run the N-rank path on fewer devices than ranks, tests only)'),
    'D4_BENCH_STRICT_TUNE': ('0', 'mode', 'bench.py: force D4_GEMM_AUTOTUNE=strict on one rank (the multi-rank setting)'),
    # --- io
    'D4_GEMM_TUNE_DEFAULT': ('dreamer4_amd/gemm_tune_default.txt', 'io', 'shipped shape -> tile table (set by dreamer4_amd._lib)'),
    'D4_GEMM_TUNE_CACHE': ('unset', 'io', 'file newly tuned shapes are appended to and read from'),
    'D4_GEMM_LOG': ('unset', 'io', 'per-shape GEMM tables / tuning log on stderr'),
}


def experiment_overrides(environ=None):
    """The experiment switches set in the environment (name -> value).  Empty in every valid test / bench run."""
    import os
    env = os.environ if environ is None else environ
    return {k: env[k] for k, (_, kind, _) in KNOBS.items() if kind == 'experiment' and k in env}
