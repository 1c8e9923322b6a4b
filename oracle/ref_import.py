"""Import the reference's dreamer4/dreamer4.py UNMODIFIED in the build container.

TEST INFRASTRUCTURE ONLY.  18 of the reference's third-party dependencies are
absent from this image; `oracle/shim/` holds stand-ins
for them (behavioural restatements for the ones on the imagination path, inert
stubs for the rest).  Nothing here travels to the GPU box: the only consumers
are `oracle/gen_golden.py` (writes tests/golden/*.npz) and the container-only
tests that cross-check `oracle/restate.py` against the reference.
"""
import importlib.util
import os
import sys

REFERENCE_ROOT = os.environ.get('D4_REFERENCE_ROOT', '/root/reference')
_SHIM = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'shim')

def reference_available():
    return os.path.isfile(os.path.join(REFERENCE_ROOT, 'dreamer4', 'dreamer4.py'))

_cached = None

# The stand-ins on the imagination path.  D4_ORACLE_REAL=1: a REAL installed package wins over its stand-in — the
# shim directory goes to the END of sys.path instead of the front, so `import x_mlps_pytorch` finds the site-packages copy first — and
# `third_party_sources()` reports, per package, which one the reference actually imported (gen_golden records it in every fixture).
SHIMMED = ('x_mlps_pytorch', 'hl_gauss_pytorch', 'discrete_continuous_embed_readout', 'assoc_scan', 'einx', 'torch_einops_utils')


def prefer_real():
    return os.environ.get('D4_ORACLE_REAL', '0') == '1'


def resolve_third_party():
    """Which of the stood-in packages would resolve to a real installation (outside oracle/shim) right now."""
    real = {}
    for name in SHIMMED:
        found = None
        for entry in sys.path:
            if os.path.abspath(entry or '.') == _SHIM:
                continue
            if os.path.isdir(os.path.join(entry or '.', name)) or os.path.isfile(os.path.join(entry or '.', name + '.py')):
                found = entry
                break
        real[name] = found
    return real


def third_party_sources():
    """After load_reference(): {'package': 'real' | 'shim'} by where the imported module lives."""
    out = {}
    for name in SHIMMED:
        mod = sys.modules.get(name)
        f = getattr(mod, '__file__', None) or ''
        out[name] = 'unloaded' if mod is None else ('shim' if os.path.abspath(f).startswith(_SHIM) else 'real')
    return out

def load_reference():
    """Returns the reference `dreamer4.dreamer4` module object."""
    global _cached
    if _cached is not None:
        return _cached
    assert reference_available(), f'{REFERENCE_ROOT} not present (GPU box?)'
    resolve_third_party()
    if _SHIM not in sys.path:
        sys.path.append(_SHIM) if prefer_real() else sys.path.insert(0, _SHIM)
    import _inert
    _inert.install()
    path = os.path.join(REFERENCE_ROOT, 'dreamer4', 'dreamer4.py')
    spec = importlib.util.spec_from_file_location('_d4_reference', path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules['_d4_reference'] = mod
    spec.loader.exec_module(mod)
    _cached = mod
    return mod
