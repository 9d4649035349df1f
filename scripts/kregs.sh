#!/bin/bash
# usage: scripts/kregs.sh andi_amd/csrc/build/scan.o  — VGPR/SGPR/spill/LDS figures of every kernel in a hipcc object
set -e
obj=$(readlink -f ${1:-andi_amd/csrc/build/scan.o})
tmp=$(mktemp -d)
cp $obj $tmp/k.o
(cd $tmp && /opt/rocm/lib/llvm/bin/llvm-objdump --offloading k.o >/dev/null)
/opt/rocm/lib/llvm/bin/llvm-readelf --notes $tmp/k.o.0.hipv4-amdgcn-amd-amdhsa--gfx950 | python3 -c "
import sys,re
txt=sys.stdin.read()
for blk in txt.split('.agpr_count')[1:]:
    g=lambda k: (re.search(r'\.'+k+r':\s+(\S+)',blk) or [None,'?'])[1]
    print('%-64s vgpr %s sgpr %s spill %s lds %s scratch %s' % (g('name')[:64], g('vgpr_count'), g('sgpr_count'), g('vgpr_spill_count'), g('group_segment_fixed_size'), g('private_segment_fixed_size')))
"
rm -rf $tmp
