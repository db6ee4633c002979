"""Build oracle/_ref/ : the UNMODIFIED reference's hot-path modules, byte-compiled from the sources where they lie.

TEST / MEASUREMENT INFRASTRUCTURE.  Run by __graft_entry__.build() in the build container (the only place
/root/reference exists).  The reference is Python, so "building" it is `py_compile`: the five modules of the path

    dfq.py, utils/__init__.py, utils/quantize.py, utils/layer_transform.py, utils/relation.py

are compiled straight from /root/reference into sourceless .pyc files under oracle/_ref/ (git-ignored, so no reference
code enters the history; NOT gpurun-ignored, so the directory travels to the GPU box like libdfq_hip.so does -- same
image, same interpreter, so the bytecode loads there).  Nothing of the product imports it: oracle/time_ref.py (the
`cpu_baseline.reference` leg of bench.py) is its only user -- it times the reference's own cross_layer_equalization /
bias_correction on the host cores of whatever box the bench runs on (north_star: "next to the
reference dfq.py CPU path timed on the same box's host cores").
"""
from __future__ import annotations

import os
import py_compile
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = '/root/reference'
OUT = os.path.join(ROOT, 'oracle', '_ref')
MODULES = ['dfq.py', 'utils/__init__.py', 'utils/quantize.py', 'utils/layer_transform.py', 'utils/relation.py']


def build(ref=REF, out=OUT):
    """Returns True when oracle/_ref is (re)built, False when the reference is not present (GPU box: keep what travelled)."""
    if not os.path.isfile(os.path.join(ref, 'dfq.py')):
        return False
    shutil.rmtree(out, ignore_errors=True)
    os.makedirs(os.path.join(out, 'utils'))
    for rel in MODULES:
        dst = os.path.join(out, rel[:-3] + '.pyc')          # sourceless layout: <name>.pyc next to where <name>.py would be
        py_compile.compile(os.path.join(ref, rel), cfile=dst, dfile='reference/' + rel, doraise=True, optimize=0)
    with open(os.path.join(out, 'PROVENANCE'), 'w') as f:
        f.write('py_compile of {} from {} by oracle/build_ref.py, python {}\n'.format(', '.join(MODULES), ref, sys.version.split()[0]))
    return True


if __name__ == '__main__':
    print('oracle/_ref built' if build() else 'no /root/reference here: oracle/_ref left as it is')
