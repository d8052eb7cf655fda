#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / SQ_INSTS_VALU passes (one directory per counter and workload, written
by the gpurun command in profiles/README.md) into profiles/pmc_traffic.json and a per-kernel CSV.

HBM bytes per launch = 2 * FETCH_SIZE[KB] * 1024 + WRITE_SIZE[KB] * 1024
(MI355X_MICROARCH.md section HBM: counters are in KB; on gfx950 FETCH_SIZE reports half the bytes of 16 B/lane
reads -- doubled here; WRITE_SIZE is exact for 16 B/lane stores and float atomics)."""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, 'gpurun_out')
tag = sys.argv[2] if len(sys.argv) > 2 else 'r01'
out = {}
rows_out = []
for wl in ('S1', 'S2', 'S3'):
    per = collections.defaultdict(dict)
    for c in ('FETCH_SIZE', 'WRITE_SIZE', 'SQ_INSTS_VALU'):
        f = glob.glob(os.path.join(src, 'pmc_%s_%s' % (wl, c), '**', '*counter_collection.csv'), recursive=True)
        if not f:
            continue
        f.sort(key=os.path.getmtime, reverse=True)      # gpurun merges into gpurun_out/: the newest pass counts
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f[0])):
            if r['Counter_Name'] == c:
                agg[r['Kernel_Name']].append(float(r['Counter_Value']))
        for k, v in agg.items():
            per[k][c] = (sum(v) / len(v), len(v))
    mix = collections.defaultdict(lambda: collections.defaultdict(list))      # instruction classes of the pair kernel
    for f in sorted(glob.glob(os.path.join(src, 'pmc_%s_MIX' % wl, '**', '*counter_collection.csv'), recursive=True),
                    key=os.path.getmtime, reverse=True)[:1]:
        for r in csv.DictReader(open(f)):
            if 'k_pair_full' in r['Kernel_Name']:
                mix[r['Counter_Name']]['v'].append(float(r['Counter_Value']))
    if not per:
        continue
    out[wl] = {}
    if mix:
        avg = {c: sum(d['v']) / len(d['v']) for c, d in mix.items()}
        out[wl]['pair_full_valu_trans_insts_per_launch'] = int(avg.get('SQ_INSTS_VALU_TRANS_F32', 0) + avg.get('SQ_INSTS_VALU_TRANS_F64', 0))
        out[wl]['pair_full_valu_f64_insts_per_launch'] = int(avg.get('SQ_INSTS_VALU_FMA_F64', 0) + avg.get('SQ_INSTS_VALU_ADD_F64', 0) +
                                                             avg.get('SQ_INSTS_VALU_MUL_F64', 0))
        out[wl]['pair_full_valu_cvt_insts_per_launch'] = int(avg.get('SQ_INSTS_VALU_CVT', 0))
    for k, d in sorted(per.items()):
        fk = d.get('FETCH_SIZE', (0.0, 0))
        wk = d.get('WRITE_SIZE', (0.0, 0))
        hbm = 2 * fk[0] * 1024 + wk[0] * 1024
        vi = d.get('SQ_INSTS_VALU', (0.0, 0))
        rows_out.append([wl, k[:100], fk[1], round(fk[0], 1), round(wk[0], 1), int(hbm), int(vi[0])])
        short = k.split('(')[0].replace('void admp::', '').strip()
        if short.startswith('k_pair_full'):
            out[wl]['pair_full_bytes_per_launch'] = int(hbm)
            out[wl]['pair_full_fetch_kb_raw'] = round(fk[0], 1)
            out[wl]['pair_full_write_kb'] = round(wk[0], 1)
            if vi[1]:
                out[wl]['pair_full_valu_insts_per_launch'] = int(vi[0])      # wave-level VALU instructions (SQ_INSTS_VALU)
for wl in out:
    out[wl]['measured_at'] = tag
with open(os.path.join(ROOT, 'profiles', 'pmc_traffic.json'), 'w') as fh:
    json.dump(out, fh, indent=1)
with open(os.path.join(ROOT, 'profiles', '%s_pmc_per_kernel.csv' % tag), 'w') as fh:
    w = csv.writer(fh)
    w.writerow(['workload', 'kernel', 'launches', 'FETCH_SIZE_KB_avg', 'WRITE_SIZE_KB_avg', 'hbm_bytes_per_launch(2F+W)',
                'SQ_INSTS_VALU_avg'])
    w.writerows(rows_out)
print(json.dumps(out, indent=1))
