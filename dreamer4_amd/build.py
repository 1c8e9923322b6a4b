"""Build libd4hip.so in-tree with plain hipcc for gfx950 (no torch headers, no JIT cache).

Every csrc/*.hip / *.cpp is compiled to its own object (in parallel, only when it or a header changed) and
the objects are linked into dreamer4_amd/libd4hip.so.  Objects live under csrc/_obj/ (git-ignored)."""
from __future__ import annotations

import glob
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, 'csrc')
OBJ = os.path.join(CSRC, '_obj')
OUT = os.path.join(_HERE, 'libd4hip.so')
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC']


def sources():
    return sorted(glob.glob(os.path.join(CSRC, '*.hip')) + glob.glob(os.path.join(CSRC, '*.cpp')))


def _headers():
    return glob.glob(os.path.join(CSRC, '*.h')) + [os.path.join(_HERE, '..', 'include', 'd4hip.h')]


def _obj_of(src):
    return os.path.join(OBJ, os.path.basename(src) + '.o')


def _stale():
    if not os.path.isfile(OUT):
        return True
    t = os.path.getmtime(OUT)
    return any(os.path.getmtime(d) > t for d in sources() + _headers())


def build(force=False, verbose=False):
    if not force and not _stale():
        return OUT
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    os.makedirs(OBJ, exist_ok=True)
    hdr_t = max(os.path.getmtime(h) for h in _headers())
    todo = []
    for s in sources():
        o = _obj_of(s)
        if force or not os.path.isfile(o) or os.path.getmtime(o) < max(os.path.getmtime(s), hdr_t):
            todo.append(s)

    def compile_one(s):
        cmd = [hipcc, *FLAGS, '-c', s, '-o', _obj_of(s)]
        if verbose:
            print(' '.join(cmd), file=sys.stderr)
        subprocess.run(cmd, check=True, cwd=CSRC)

    with ThreadPoolExecutor(max_workers=min(len(todo), os.cpu_count() or 1) or 1) as pool:
        list(pool.map(compile_one, todo))
    for stale in set(glob.glob(os.path.join(OBJ, '*.o'))) - {_obj_of(s) for s in sources()}:
        os.remove(stale)                                   # a source file was removed / renamed
    cmd = [hipcc, '--offload-arch=gfx950', '-fPIC', '-shared', '-o', OUT, *[_obj_of(s) for s in sources()]]
    if verbose:
        print(' '.join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True, cwd=CSRC)
    return OUT


if __name__ == '__main__':
    print(build(force='--force' in sys.argv, verbose=True))
