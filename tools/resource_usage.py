"""Registers / occupancy / scratch per kernel of one csrc file, from hipcc's -Rpass-analysis=kernel-resource-usage.
   python tools/resource_usage.py pair_kernels.hip [name-filter]"""
import os, re, subprocess, sys, tempfile
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
from admp_amd import build as B

src = os.path.join(B.CSRC, sys.argv[1])
flt = sys.argv[2] if len(sys.argv) > 2 else ''
with tempfile.TemporaryDirectory() as d:
    r = subprocess.run(['hipcc'] + B.FLAGS + ['-Rpass-analysis=kernel-resource-usage', '-c', src, '-o', os.path.join(d, 'x.o')],
                       capture_output=True, text=True)
rows, cur = [], None
for line in r.stderr.splitlines():
    m = re.search(r'remark:\s+(.*?) \[-Rpass', line)
    if not m:
        continue
    t = m.group(1).strip()
    if t.startswith('Function Name:') or t.startswith('Name:'):
        cur = {'name': t.split(':', 1)[1].strip()}
        rows.append(cur)
    elif cur is not None and ':' in t:
        k, v = t.split(':', 1)
        cur[k.strip()] = v.strip()
for c in rows:
    name = subprocess.run(['c++filt', c['name']], capture_output=True, text=True).stdout.strip()
    name = re.sub(r'\(.*', '', name).replace('void admp::', '')
    if flt and flt not in name:
        continue
    print('%-48s VGPR %4s AGPR %3s SGPR %3s occ %2s scratch %5s LDS %6s' % (
        name[:48], c.get('VGPRs', '?'), c.get('AGPRs', '?'), c.get('TotalSGPRs', c.get('SGPRs', '?')),
        c.get('Occupancy [waves/SIMD]', '?'), c.get('ScratchSize [bytes/lane]', '?'), c.get('LDS Size [bytes/block]', '?')))
