#!/usr/bin/env python3
"""The timed S1 loop of bench.py alone (moving frames, warm-started SCF), for a kernel trace:
    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/s1trace -- python3 tools/s1_trace.py [steps]
    python tools/timeline.py gpurun_out/s1trace k_prepare_sites 2        # the last two complete steps"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
w = bench.make_workload(sys.argv[2] if len(sys.argv) > 2 else 'S1')
f, a = bench.make_force(w)
frames = bench.ThermalFrames(w, torch.device('cuda'))
dt, _, cyc = bench.run_timed(f, a, steps, 5, frames, only=False)
print('%s: %.4f ms/step, cycles %s' % (w['name'], dt / steps * 1e3, cyc))
