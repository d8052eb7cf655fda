import os, sys, time
sys.path.insert(0, '/root/repo')
import torch, bench
w = bench.make_workload('S3')
f, a = bench.make_force(w)
fr = bench.ThermalFrames(w, torch.device('cuda', 0))
print(bench.md_all_terms(w, f, a, fr, 5, 2))
print(bench.md_all_terms(w, f, a, fr, 10, 3))
