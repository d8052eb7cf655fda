#!/usr/bin/env python3
"""Compile one .hip file for gfx950 and print a per-kernel register / scratch / occupancy table.

usage: tools/kernel_resources.py admp_amd/csrc/pair_kernels.hip [filter-substring] [-- extra hipcc flags]
"""
import re
import subprocess
import sys

args = sys.argv[1:]
extra = []
if '--' in args:
    k = args.index('--')
    extra = args[k + 1:]
    args = args[:k]
src = args[0]
flt = args[1] if len(args) > 1 else ''
cmd = ['hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-munsafe-fp-atomics', '-fno-slp-vectorize',
       '-Rpass-analysis=kernel-resource-usage', '-c', src, '-o', '/dev/null'] + extra
out = subprocess.run(cmd, capture_output=True, text=True).stderr
rows, cur = [], None
for line in out.splitlines():
    if 'error' in line:
        print(line)
    m = re.search(r'remark:\s+(.*?)\s+\[-Rpass', line)
    if not m:
        continue
    t = m.group(1)
    if t.startswith('Function Name:'):
        name = subprocess.run(['c++filt', t.split(':', 1)[1].strip()], capture_output=True, text=True).stdout.strip()
        cur = {'name': re.sub(r'\(.*', '', name).replace('void admp::', '')}
        rows.append(cur)
    elif cur is not None and ':' in t:
        k, v = t.split(':', 1)
        cur[k.strip()] = v.strip()
print(f"{'kernel':58s} {'SGPR':>5s} {'VGPR':>5s} {'AGPR':>5s} {'scratch':>8s} {'occ':>4s} {'LDS':>7s}")
for r in rows:
    if flt in r['name'] and 'rocprim' not in r['name']:
        print(f"{r['name'][:58]:58s} {r.get('TotalSGPRs', '?'):>5s} {r.get('VGPRs', '?'):>5s} {r.get('AGPRs', '?'):>5s} "
              f"{r.get('ScratchSize [bytes/lane]', '?'):>8s} {r.get('Occupancy [waves/SIMD]', '?'):>4s} "
              f"{r.get('LDS Size [bytes/block]', '?'):>7s}")
