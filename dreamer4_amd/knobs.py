"""The ONE table of `D4_*` environment switches (csrc/*.hip read them with getenv at first use, a few Python modules with os.environ).

kind:
  'experiment'  (none left: round 5) an A/B switch of a measured decision.  tests/conftest.py and bench.py still refuse to run with one set.
  'mode'        a legitimate runtime mode a deployment may choose (memory / traceability / multi-rank behaviour), numerically pinned by tests.
  'io'          paths and logging; no effect on results or kernel choice.
`tests/test_host.py::test_every_environment_switch_is_in_the_knob_table` greps the sources and fails on a switch missing here."""

KNOBS = {
    # --- modes
    'D4_GRAPH_MAX_ROWS': ('4096', 'mode', 'decode frames of <= this many token rows are replayed from hipGraphs (0: always enqueue eagerly); the same kernels by the same rules either way (asserted bitwise); read at engine creation'),
    'D4_GEMM_AUTOTUNE': ('1', 'mode', '1: time a new GEMM shape\'s tile configurations at first use (same bits whichever wins); 0: static choice; strict: fail on a shape missing from the shipped tile table (multi-rank bench).  Shapes first seen under stream capture or under 1e8 flops always take the static choice'),
    'D4_FORCE_PG': ('0', 'mode', '1: create the process group and issue every collective in a one-rank world (RCCL path on one GPU)'),
    'D4_TRUNK_SAVE_FORWARD': ('1', 'mode', 'training blocks keep their forward workspace (0: recompute in the backward, less memory)'),
    'D4_TRUNK_DISPATCHER': ('0', 'mode', '1: training blocks go through torch.ops.d4hip.* (torch.compile-traceable) instead of autograd.Function'),
    'D4_DP_BACKEND': ('gloo', 'mode', 'tests/dp_gpu_worker.py: process-group backend'),
    'D4_DP_SIZE': ('small', 'mode', 'tests/dp_gpu_worker.py: `headline` runs the two-rank equivalence at config 2\'s architecture, global batch 256 x 16 frames'),
    'D4_BENCH_BACKEND': ('nccl', 'mode', 'bench.py: process-group backend (gloo: run the N-rank path on fewer devices than ranks, tests only)'),
    'D4_BENCH_STRICT_TUNE': ('0', 'mode', 'bench.py: force D4_GEMM_AUTOTUNE=strict on one rank (the multi-rank setting)'),
    # --- io
    'D4_GEMM_TUNE_DEFAULT': ('dreamer4_amd/gemm_tune_default.txt', 'io', 'shipped shape -> tile table (set by dreamer4_amd._lib)'),
    'D4_GEMM_TUNE_CACHE': ('unset', 'io', 'file newly tuned shapes are appended to and read from'),
    'D4_GEMM_LOG': ('unset', 'io', 'per-shape GEMM tables / tuning log on stderr'),
}
# Round 5 retired the 24 `experiment` switches of rounds 2-4 (every one an A/B arm of a decision recorded in profiles/ as level-or-worse for two rounds):
# the losing arms are gone from csrc/, the three bit-identical fusions that tests still compare against their unfused forms are reached through C test
# hooks (d4_frame_fused_set, d4_debug_switch), never through the environment.


# The retired names: setting one no longer does anything, so a run that sets one is NOT the run its author thinks it is — tests/conftest.py and bench.py
# refuse to start (the same guard that used to catch a live experiment switch).
RETIRED = (
    'D4_ATTN_OUT_COLS', 'D4_BF16A_GROUPED', 'D4_BF16A_PAIR', 'D4_BF16_ACT', 'D4_BF16_DMA', 'D4_FRAME_FUSED', 'D4_GEMM_DX_T', 'D4_GEMM_PAIR', 'D4_GEMM_SKINNY',
    'D4_GEMM_TN', 'D4_GEMM_V2', 'D4_GEMM_X3', 'D4_GEMM_X3SK', 'D4_KV_APPEND_LEGACY', 'D4_POOL_MIX_ROWS', 'D4_POOL_MIX_ROWS_MAX', 'D4_SKINNY_MAXM', 'D4_SKINNY_NW',
    'D4_SKINNY_PAIR', 'D4_SKINNY_TILES', 'D4_SPACE_ATTN_MFMA', 'D4_TIME_ATTN_FEW', 'D4_TIME_ATTN_FUSED_APPEND', 'D4_TIME_ATTN_LEGACY',
)


def experiment_overrides(environ=None):
    """The experiment switches set in the environment (name -> value): live ones (none since round 5) and RETIRED names, which are silently ignored by the
    library and therefore must not be set.  Empty in every valid test / bench run."""
    import os
    env = os.environ if environ is None else environ
    live = {k: env[k] for k, (_, kind, _) in KNOBS.items() if kind == 'experiment' and k in env}
    live.update({k: env[k] + ' (retired: ignored by the library)' for k in RETIRED if k in env})
    return live
