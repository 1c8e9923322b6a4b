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

def load_reference():
    """Returns the reference `dreamer4.dreamer4` module object."""
    global _cached
    if _cached is not None:
        return _cached
    assert reference_available(), f'{REFERENCE_ROOT} not present (GPU box?)'
    if _SHIM not in sys.path:
        sys.path.insert(0, _SHIM)
    import _inert
    _inert.install()
    path = os.path.join(REFERENCE_ROOT, 'dreamer4', 'dreamer4.py')
    spec = importlib.util.spec_from_file_location('_d4_reference', path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules['_d4_reference'] = mod
    spec.loader.exec_module(mod)
    _cached = mod
    return mod
