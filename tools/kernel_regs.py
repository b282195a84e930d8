"""VGPR / SGPR / LDS / scratch of the kernels in a hipcc object or shared library: pulls the gfx950 code object out of the
clang offload bundle(s) and reads its AMDGPU metadata notes.  usage: python tools/kernel_regs.py <file.o|.so> [name filter]"""
import re
import struct
import subprocess
import sys
import tempfile

data = open(sys.argv[1], 'rb').read()
flt = sys.argv[2] if len(sys.argv) > 2 else ''
magic = b'__CLANG_OFFLOAD_BUNDLE__'
pos = 0
while True:
    i = data.find(magic, pos)
    if i < 0:
        break
    pos = i + len(magic)
    n, = struct.unpack_from('<Q', data, pos)
    p = pos + 8
    for _ in range(n):
        off, size, tl = struct.unpack_from('<QQQ', data, p)
        triple = data[p + 24:p + 24 + tl].decode()
        p += 24 + tl
        if 'gfx950' not in triple or size == 0:
            continue
        with tempfile.NamedTemporaryFile(suffix='.co') as f:
            f.write(data[i + off:i + off + size])
            f.flush()
            txt = subprocess.run(['/opt/rocm/lib/llvm/bin/llvm-readelf', '--notes', f.name], capture_output=True, text=True).stdout
        for blk in txt.split('- .agpr_count')[1:]:
            name = re.search(r'\.name:\s+(\S+)', blk)
            if not name or flt not in name.group(1):
                continue
            g = lambda k: (re.search(r'\.%s:\s+(\d+)' % k, blk) or [None, '?'])[1]
            dem = subprocess.run(['c++filt', name.group(1)], capture_output=True, text=True).stdout.strip()
            print('vgpr %3s agpr %3s sgpr %3s lds %6s scratch %5s spill_v %3s  %s' % (g('vgpr_count'), re.match(r':\s+(\d+)', blk).group(1) if re.match(r':\s+(\d+)', blk) else '?',
                  g('sgpr_count'), g('group_segment_fixed_size'), g('private_segment_fixed_size'), g('vgpr_spill_count'), dem[:150]))
