#!/bin/bash
# registers / scratch / LDS of the kernels of the built library, from the code object's metadata:  tools/kernel_regs.sh [pattern]
SO=${SO:-$(dirname $0)/../airspy-fmradion_amd/libfmradion_amd.so}
T=$(mktemp -d)
python3 - "$SO" "$T/co.elf" <<'PY'
import sys
d=open(sys.argv[1],'rb').read()
i=d.find(b'__CLANG_OFFLOAD_BUNDLE__')
import struct
n=struct.unpack_from('<Q',d,i+24)[0]
o=i+32
for _ in range(n):
    off,size,tl=struct.unpack_from('<QQQ',d,o); t=d[o+24:o+24+tl].decode(); o+=24+tl
    if 'gfx950' in t: open(sys.argv[2],'wb').write(d[i+off:i+off+size])
PY
/opt/rocm/lib/llvm/bin/llvm-readelf --notes $T/co.elf | python3 -c "
import sys,re
pat=sys.argv[1] if len(sys.argv)>1 else ''
txt=sys.stdin.read()
for blk in txt.split('- .agpr_count')[1:]:
    g=lambda k: (re.search(r'\.'+k+r':\s*(\S+)',blk) or [None,'?'])[1]
    name=g('name')
    if pat in name: print('%-90s vgpr %s agpr %s sgpr %s spill %s scratch %s lds %s' % (name[:90], g('vgpr_count'), blk.split()[0].strip(':'), g('sgpr_count'), g('vgpr_spill_count'), g('private_segment_fixed_size'), g('group_segment_fixed_size')))
" "$1"
rm -rf $T
