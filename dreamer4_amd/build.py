"""Build libd4hip.so in-tree with plain hipcc for gfx950 (no torch headers, no JIT cache)."""
from __future__ import annotations

import glob
import os
import subprocess
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, 'csrc')
OUT = os.path.join(_HERE, 'libd4hip.so')


def sources():
    return sorted(glob.glob(os.path.join(CSRC, '*.hip')) + glob.glob(os.path.join(CSRC, '*.cpp')))


def _stale():
    if not os.path.isfile(OUT):
        return True
    deps = sources() + glob.glob(os.path.join(CSRC, '*.h')) + [os.path.join(_HERE, '..', 'include', 'd4hip.h')]
    t = os.path.getmtime(OUT)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    if not force and not _stale():
        return OUT
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    cmd = [hipcc, '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-shared', '-o', OUT, *sources()]
    if verbose:
        print(' '.join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True, cwd=CSRC)
    return OUT


if __name__ == '__main__':
    print(build(force='--force' in sys.argv, verbose=True))
