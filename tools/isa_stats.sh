#!/bin/bash
# register / spill / scratch figures of every kernel of one source file (from the compiler's metadata)
# usage: tools/isa_stats.sh dfq_le_resident.hip [extra flags]
f=$1; shift
d=$(mktemp -d)
( cd $d && /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -I/root/repo/dfq_amd/csrc "$@" -save-temps -c /root/repo/dfq_amd/csrc/$f -o x.o 2>/dev/null )
s=$(ls $d/*gfx950*.s | head -1)
python3 - "$s" <<'PY'
import re, sys
txt = open(sys.argv[1]).read()
for m in re.finditer(r'- \.agpr_count:.*?\.wavefront_size:\s+\d+', txt, flags=re.S):
    blk = m.group(0)
    g = lambda k: (re.search(r'\.%s:\s+(\S+)' % k, blk) or [None, '?'])[1]
    print('{:40s} vgpr {:>4s} sgpr {:>4s} sgpr_spill {:>4s} vgpr_spill {:>4s} scratch {:>5s} lds {:>6s}'.format(
        g('name')[:40], g('vgpr_count'), g('sgpr_count'), g('sgpr_spill_count'), g('vgpr_spill_count'),
        g('private_segment_fixed_size'), g('group_segment_fixed_size')))
PY
cp $s /tmp/last_isa.s; rm -rf $d
