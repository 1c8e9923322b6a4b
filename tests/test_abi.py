"""CPU: the C-ABI shared library loads and exports every symbol include/d4hip.h declares; engine
construction / validation logic runs without a GPU (no compute calls)."""
import ctypes as C
import os
import re

import pytest

from dreamer4_amd import _lib
from dreamer4_amd.build import build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def lib():
    build()
    return _lib.load()


def header_symbols():
    text = open(os.path.join(ROOT, 'include', 'd4hip.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(d4_[a-z0-9_]+)\s*\(', text)))


def test_every_declared_symbol_is_exported_and_bound(lib):
    syms = header_symbols()
    assert len(syms) >= 20
    for s in syms:
        assert hasattr(lib, s), f'{s} declared in d4hip.h but not exported by libd4hip.so'
    assert set(syms) == set(_lib.SYMBOLS), set(syms) ^ set(_lib.SYMBOLS)
    assert lib.d4_version() == 2


def _cfg(**over):
    c = _lib.Config()
    c.dim, c.dim_latent, c.num_latent_tokens, c.depth, c.time_block_every = 64, 8, 6, 4, 2
    c.attn_heads, c.attn_dim_head, c.attn_softclamp_value = 2, 64, 50.
    c.num_spatial_tokens, c.num_register_tokens, c.max_steps, c.num_tasks = 4, 8, 64, 0
    c.num_discrete_action_types = 1
    c.num_discrete_actions[0] = 4
    c.multi_token_pred_len, c.policy_head_mlp_depth, c.value_head_mlp_depth, c.terminal_mlp_depth = 8, 3, 3, 1
    c.predict_terminals, c.reward_num_bins, c.value_num_bins, c.pool_heads, c.pool_dim_head = 1, 255, 255, 4, 64
    c.max_batch, c.max_frames, c.max_parallel_frames, c.max_learn_rows = 4, 8, 1, 32
    for k, v in over.items():
        setattr(c, k, v)
    return c


def test_engine_create_and_workspace_size(lib):
    e = C.c_void_p()
    cfg = _cfg()
    _lib.check(lib.d4_engine_create(C.byref(cfg), C.byref(e)))
    small = lib.d4_engine_workspace_bytes(e)
    assert small > 1 << 20
    assert lib.d4_engine_cache_frames(e) == 0
    lib.d4_engine_destroy(e)
    cfg = _cfg(max_batch=64)
    _lib.check(lib.d4_engine_create(C.byref(cfg), C.byref(e)))
    assert lib.d4_engine_workspace_bytes(e) > small
    lib.d4_engine_destroy(e)


@pytest.mark.parametrize('over,frag', [
    (dict(attn_dim_head=48), 'attn_dim_head'),
    (dict(num_spatial_tokens=65), 'at most 64'),
    (dict(max_steps=48), 'power of two'),
    (dict(dim=66), 'multiples of 4'),
])
def test_unsupported_configs_fail_loudly(lib, over, frag):
    e = C.c_void_p()
    cfg = _cfg(**over)
    rc = lib.d4_engine_create(C.byref(cfg), C.byref(e))
    assert rc != 0
    assert frag in lib.d4_last_error().decode()
    with pytest.raises(_lib.D4Error):
        _lib.check(rc)


def test_prepare_without_weights_names_the_missing_key(lib):
    e = C.c_void_p()
    cfg = _cfg()
    _lib.check(lib.d4_engine_create(C.byref(cfg), C.byref(e)))
    rc = lib.d4_engine_prepare(e, None)
    assert rc != 0 and 'workspace' in lib.d4_last_error().decode()
    lib.d4_engine_destroy(e)
