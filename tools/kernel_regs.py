"""Register / LDS usage of the kernels in libfami_hip.so (reads the AMDGPU metadata notes of the embedded code objects).
usage: python tools/kernel_regs.py [substring ...]"""
import re
import struct
import subprocess
import sys
import tempfile
import os

so = sys.argv.pop(1) if len(sys.argv) > 1 and sys.argv[1].endswith(('.so', '.o')) else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'fami-pose_amd', 'libfami_hip.so')
data = open(so, 'rb').read()
magic = b'__CLANG_OFFLOAD_BUNDLE__'
pos, rows = 0, []
while True:
    i = data.find(magic, pos)
    if i < 0:
        break
    n = struct.unpack_from('<Q', data, i + 24)[0]
    off = i + 32
    for _ in range(n):
        o, sz, tl = struct.unpack_from('<QQQ', data, off)
        off += 24
        trip = data[off:off + tl].decode()
        off += tl
        if 'gfx950' in trip and sz > 0:
            with tempfile.NamedTemporaryFile(suffix='.o', delete=False) as f:
                f.write(data[i + o:i + o + sz])
            txt = subprocess.run(['/opt/rocm/lib/llvm/bin/llvm-readelf', '--notes', f.name], capture_output=True, text=True).stdout
            os.unlink(f.name)
            for e in re.split(r'\n\s*- \.agpr_count', txt)[1:]:
                g = lambda k: re.search(r'\.%s:\s+(\S+)' % k, e)
                rows.append((g('name').group(1), int(g('vgpr_count').group(1)), int(g('sgpr_count').group(1)),
                             int(g('group_segment_fixed_size').group(1)), int(g('private_segment_fixed_size').group(1))))
    pos = i + 24
keys = sys.argv[1:]
for nm, vg, sg, lds, scr in sorted(rows):
    if not keys or any(k in nm for k in keys):
        dem = subprocess.run(['c++filt', nm], capture_output=True, text=True).stdout.strip()
        print('%-90s vgpr %3d sgpr %3d lds %6d scratch %d' % (dem[:90], vg, sg, lds, scr))
