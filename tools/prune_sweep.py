#!/usr/bin/env python3
"""All-terms MD step (bench.md_all_terms) with the inner list pruned every m steps: python tools/prune_sweep.py [S3] [m ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
name = sys.argv[1] if len(sys.argv) > 1 else 'S3'
ms = [int(x) for x in sys.argv[2:]] or [1, 2, 5]
w = bench.make_workload(name)
f, a = bench.make_force(w)
fr = bench.ThermalFrames(w, torch.device('cuda', 0))
for m in ms:
    out = bench.md_all_terms(w, f, a, fr, 10, 2, prune=m)
    il = out.get('inner_list', {})
    print('prune every %d steps (inner rc + %.2f A): all terms %.3f ms (median %.3f) -> with inner list %.3f ms (median %.3f), %d pairs' % (
        m, bench.SKIN * m / 10.0, out['ms_per_step'], out['step_ms_min_median_max'][1], il.get('ms_per_step', float('nan')),
        il.get('step_ms_min_median_max', [0, float('nan')])[1], il.get('n_pairs_inner', 0)))
