"""TEST INFRASTRUCTURE ONLY: compile the unmodified kernel sources of dfq_amd/csrc with g++ against
the fiber-based HIP emulation in tests/emu/include, producing tests/emu/_build/libdfq_emu.so with the
same C ABI as the product library.  Used by the `-m "not gpu"` tests to run every kernel's logic on
the CPU of the build container.  The product package never loads this library."""
from __future__ import annotations

import glob
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
OUT = os.path.join(HERE, '_build', 'libdfq_emu.so')


def sources():
    src = sorted(glob.glob(os.path.join(ROOT, 'dfq_amd', 'csrc', '*.hip')))
    src += sorted(glob.glob(os.path.join(ROOT, 'dfq_amd', 'csrc', '*.cpp')))
    src.append(os.path.join(HERE, 'emu_runtime.cpp'))
    return src


def deps():
    d = sources()
    d += glob.glob(os.path.join(ROOT, 'dfq_amd', 'csrc', '*.hpp'))
    d += glob.glob(os.path.join(ROOT, 'include', '*.h'))
    d += glob.glob(os.path.join(HERE, 'include', 'hip', '*.h'))
    return d


def _fresh():
    if not os.path.exists(OUT):
        return False
    t = os.path.getmtime(OUT)
    return all(os.path.getmtime(f) <= t for f in deps())


def build(force=False):
    """Builds at most once even when several pytest-xdist workers ask at the same time: an exclusive lock around the check and
    the build, the library written under a temporary name and renamed into place (a worker never maps a half-written file)."""
    import fcntl
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    if not force and _fresh():
        return OUT
    with open(OUT + '.lock', 'w') as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        if not force and _fresh():
            return OUT
        _compile(OUT + '.tmp.{}'.format(os.getpid()))
        os.replace(OUT + '.tmp.{}'.format(os.getpid()), OUT)
    return OUT


def _compile(out):
    """every source to an object file of its own, in parallel (one g++ over all sources took 28 s), then one link"""
    from concurrent.futures import ThreadPoolExecutor
    flags = ['-O2', '-g', '-std=c++17', '-fPIC', '-ffp-contract=off', '-fno-fast-math', '-fno-strict-aliasing',
             '-DDFQ_GLOBAL_AS=', '-DDFQ_CONSTANT_AS=', '-Wall', '-Wno-unknown-pragmas', '-Wno-unused-function', '-Wno-unused-variable',
             '-I', os.path.join(HERE, 'include'), '-I', os.path.join(ROOT, 'include')]
    objdir = out + '.objs'
    os.makedirs(objdir, exist_ok=True)
    objs = [os.path.join(objdir, os.path.basename(s) + '.o') for s in sources()]

    def one(job):
        src, obj = job
        subprocess.run(['g++'] + flags + ['-c', '-x', 'c++', src, '-o', obj], check=True)
    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as pool:
        list(pool.map(one, zip(sources(), objs)))
    subprocess.run(['g++', '-shared', '-fPIC', '-o', out] + objs, check=True)
    for o in objs:
        os.remove(o)
    os.rmdir(objdir)


if __name__ == '__main__':
    print(build(force=True))
