#!/usr/bin/env python3
"""Developer helper: per-kernel register / scratch / LDS usage of one csrc/*.hip file (hipcc -Rpass-analysis=kernel-resource-usage).
usage: tools/resusage.py spectral_kernels.hip [filter-substring] [extra hipcc flags...]"""
import re, subprocess, sys, os
src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 and not sys.argv[2].startswith('-') else ''
extra = [a for a in sys.argv[2:] if a.startswith('-')]
csrc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'neural-navier-stokes_amd', 'csrc')
exact = ['-ffp-contract=off'] if src.split('.')[0] in ('fd_kernels', 'sor_kernels', 'cheb_kernels', 'coarsen_kernels') else []
noslp = ['-fno-slp-vectorize'] if src.startswith(('spectral_k', 'spectral_b', 'spectral_s')) else []
cmd = ['/opt/rocm/bin/hipcc', '-O3', '-std=c++17', '-fPIC', '--offload-arch=gfx950', '-fvisibility=hidden', '-Rpass-analysis=kernel-resource-usage',
       '-c', os.path.join(csrc, src), '-o', '/dev/null'] + exact + noslp + extra
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None
rows = []
for line in out.splitlines():
    m = re.search(r'Function Name: (\S+)', line)
    if m:
        cur = dict(name=subprocess.run(['c++filt', m.group(1)], capture_output=True, text=True).stdout.strip()[:110]); rows.append(cur); continue
    for key, pat in (('vgpr', r' VGPRs: (\d+)'), ('agpr', r'AGPRs: (\d+)'), ('scratch', r'ScratchSize \[bytes/lane\]: (\d+)'), ('occ', r'Occupancy \[waves/SIMD\]: (\d+)'),
                     ('lds', r'LDS Size \[bytes/block\]: (\d+)'), ('sgpr', r' SGPRs: (\d+)')):
        m = re.search(pat, line)
        if m and cur is not None:
            cur[key] = int(m.group(1))
if not rows:
    print(out[-3000:])
for r in rows:
    if flt in r['name']:
        print('%-112s v%3d a%3d scratch %4d occ %d lds %6d' % (r['name'], r.get('vgpr', -1), r.get('agpr', -1), r.get('scratch', -1), r.get('occ', -1), r.get('lds', -1)))
