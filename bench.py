#!/usr/bin/env python3
"""Benchmark of the PME hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload S1|S2|S3] [--no-scale] [--no-cpu]

A "step" is one `ADMPPmeForce.get_forces` call (electrostatics incl. the induced-dipole SCF started from the
previous step's dipoles, fixed pair list) = one MD step's worth of the hot path.  The reference has no
integrator; ns/day is defined with dt = 1 fs:  ns/day = 0.0864 / t_step[s].

THE ATOMS MOVE in the timed region: every step evaluates the next frame of a deterministic thermal trajectory
(`thermal_frames`: every water translates and rotates rigidly with seeded Maxwell-Boltzmann velocities at 300 K,
1 fs between frames, frames resident in HBM before the timed region), so the SCF really has to follow the
geometry: `scf_field_evaluations_per_step` / `jacobi_updates_per_step` are measured, at the reference's default
threshold POL_CONV = 10 (admp/settings.py:29) for `value` and at 1e-2 in the `scf_tight` leg.  The old
static-geometry number (identical positions every step: the first SCF check always passes) is kept only as the
labelled upper bound `static_geometry`.  `md_all_terms` times what examples/md/nve_water.py does per step: PME +
dispersion PME + Tang-Toennies on the moving frames with a Verlet list (skin 1 A) rebuilt every 10 steps.

Default workload (N = 1): S1 = BASELINE.json configs[1] -- 1024 polarizable MPID waters, double precision,
on the seeded synthetic liquid box (the shipped water1024.pdb geometry has 0.67 A contacts on which the
reference's own Jacobi SCF diverges), plus an `at_scale` leg on S3 (1 048 575 atoms, f32).

N > 1 (`--gpus N`): the path shards, so the value is the SLAB-DECOMPOSED step of BASELINE configs[3]
(S3: 1 048 575 polarizable atoms, K = 256, f32) over the N GPUs -- x-slabs of the mesh and of space, RCCL issued by the
library itself (admp_amd/csrc/rccl_comm.hip), every rank keeping its home rows -- `scaling: "strong"`.  A 3072-atom box does
not strong-scale; the aggregate of N independent S1 replicas is kept only as the labelled extra `replicas_S1`.  When the
process was not started by a launcher (no WORLD_SIZE in the environment), `--gpus N` makes this process -- which has not
touched the GPU yet -- start N fresh child ranks (one per device, 127.0.0.1 rendezvous) and relay rank 0's line; a failed
rank makes the run exit non-zero.  ADMP_BENCH_BACKEND=gloo (+ ADMP_BENCH_SCALE=S2) is the functional rehearsal of several
ranks on one GPU (host-staged collectives, timings not representative).

One JSON line is printed by rank 0 (contract in the task statement), with `roofline` (real-space pair
kernel, measured with HIP events inside the timed region) and `cpu_baseline` (the float64 oracle timed on
the host cores on a frame of the same trajectory, SCF warm-started like the GPU step).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s spec, 6.29 measured copy)
DT_FS = 1.0
RC = 4.0
SKIN = 1.0                   # Verlet skin of the md_all_terms leg (examples/md/nve_water.py)
TEMP_K = 300.0
KB = 0.0083144626            # kJ/mol/K
MASS = (15.999, 1.008, 1.008)
EXCURSION = 10               # trajectory time runs 0 .. +10 .. -10 .. 0 fs (triangle wave): |dt| = 1 fs every step

WORKLOADS = {
    # name: (n_mol, precision, K (None = reference rule), description)
    'S1': (1024, 'double', None, 'water_pol_1024 (BASELINE configs[1]): 1024 polarizable MPID waters = 3072 atoms, '
                                 'seeded liquid box L=31.289 A, rc 4 A, f64'),
    'S2': (32768, 'single', 128, '98 304-atom polarizable water box (configs[2] size), K=128, rc 4 A, f32'),
    'S2ref': (32768, 'single', None, '98 304-atom polarizable water box with the mesh of the reference rule '
                                     '(admp/pme.py:146-172: K = 305 = 5 * 61, a Bluestein size), rc 4 A, f32'),
    'S3': (349525, 'single', 256, '1 048 575-atom polarizable water box (configs[3] size), K=256, rc 4 A, f32'),
}


def pair_kernel_bytes(n_pairs, n_atoms, wbytes, polarizable):
    """Algorithmic bytes of one energy+adjoint pass of the pair kernel:
    per pair 8 B of indices + j-side read nr*w + j-side accumulate nw*w + i-side (nr+nw)*w / nbar."""
    nr, nw = (17, 15) if polarizable else (12, 12)
    nbar = n_pairs / float(n_atoms)
    per_pair = 8 + (nr + nw) * wbytes + (nr + nw) * wbytes / nbar
    return per_pair * n_pairs, per_pair


def make_workload(name):
    from admp_amd import systems as S
    n_mol, prec, K, desc = WORKLOADS[name]
    pos, box = S.synthetic_water_box(n_mol, seed=20240)
    at, ai, cov = S.water_topology(n_mol)
    par = S.water_parameters(n_mol, polarizable=True)
    # pair list from the GPU cell list (admp_amd.neighbor); the host cKDTree builder gives the same set
    from admp_amd import settings
    from admp_amd.neighbor import NeighborList
    settings.PRECISION = prec
    pairs = NeighborList(box, RC).allocate(pos)
    return dict(name=name, n_mol=n_mol, prec=prec, K=K, desc=desc, pos=pos, box=box, at=at, ai=ai, cov=cov, par=par,
                pairs=pairs)


def frame_time(step_index):
    """Trajectory time (fs, integer) of timed/warm-up step number `step_index`: a triangle wave of amplitude EXCURSION, so
    that consecutive steps always differ by exactly 1 fs of motion while the excursion from the start stays bounded
    (ballistic rigid bodies would eventually overlap) and only 2*EXCURSION+1 distinct frames exist, whatever --steps is."""
    p = step_index % (4 * EXCURSION)
    if p <= EXCURSION:
        return p
    if p <= 3 * EXCURSION:
        return 2 * EXCURSION - p
    return p - 4 * EXCURSION


class ThermalFrames:
    """Deterministic thermal motion of the water box: every molecule translates with a Maxwell-Boltzmann centre-of-mass
    velocity (300 K, 18 amu: 0.0037 A/fs per component) and rotates rigidly about its centre of mass with a thermal
    angular velocity (omega = I^-1 L, L ~ N(0, kT I): the hydrogens move ~0.013 A/fs), seeded on the host.  frame(t) is
    the geometry at trajectory time t fs, built on the GPU in f64, cast to the workload's precision, cached."""

    def __init__(self, w, device, seed=11):
        import torch
        self.dtype = torch.float32 if w['prec'] == 'single' else torch.float64
        n_mol = w['n_mol']
        g = torch.Generator().manual_seed(seed)
        xi = torch.randn((n_mol, 6), generator=g, dtype=torch.float64).to(device)
        m = torch.tensor(MASS, dtype=torch.float64, device=device)
        mol = torch.as_tensor(w['pos'], dtype=torch.float64, device=device).reshape(n_mol, 3, 3)
        self.com = (mol * m[None, :, None]).sum(1) / m.sum()
        self.rel = mol - self.com[:, None, :]
        kt = KB * TEMP_K
        self.v = xi[:, :3] * float(np.sqrt(kt / float(m.sum()))) * 1e-2            # A/fs
        # principal axes of a C2v water: bisector, in-plane perpendicular, plane normal; omega_k ~ N(0, kT / I_k) about each
        rel = self.rel
        e1 = rel[:, 1] + rel[:, 2] - 2 * rel[:, 0]
        e1 = e1 / e1.norm(dim=1, keepdim=True)
        e3 = torch.linalg.cross(rel[:, 1] - rel[:, 0], rel[:, 2] - rel[:, 0])
        e3 = e3 / e3.norm(dim=1, keepdim=True)
        e2 = torch.linalg.cross(e3, e1)
        axes = torch.stack([e1, e2, e3], dim=1)                                    # (n_mol, 3 axes, 3)
        proj = torch.einsum('nak,nbk->nab', rel, axes)                             # atom coordinates on the axes
        r2 = (proj ** 2).sum(-1, keepdim=True)
        inertia = (m[None, :, None] * (r2 - proj ** 2)).sum(1)                     # (n_mol, 3) principal moments
        omega = (xi[:, 3:] * torch.sqrt(kt / inertia))[:, :, None] * axes          # (n_mol, 3, 3)
        omega = omega.sum(1) * 1e-2                                                # rad/fs
        self.wnorm = omega.norm(dim=1, keepdim=True)
        self.axis = omega / self.wnorm
        self.cache = {}
        self.rms_step = None

    def frame(self, t):
        import torch
        t = int(t)
        if t not in self.cache:
            th = (self.wnorm * float(t))[:, None, :]                               # (n_mol, 1, 1)
            n = self.axis[:, None, :]
            r = self.rel
            rot = r * torch.cos(th) + torch.linalg.cross(n.expand_as(r), r) * torch.sin(th) + \
                n * (n * r).sum(-1, keepdim=True) * (1 - torch.cos(th))
            f = (self.com + self.v * float(t))[:, None, :] + rot
            self.cache[t] = f.reshape(-1, 3).to(self.dtype).contiguous()
        return self.cache[t]

    def step_frame(self, k):
        return self.frame(frame_time(k))

    def describe(self):
        import torch
        d = (self.frame(1).double() - self.frame(0).double()).norm(dim=1)
        return ('rigid-body thermal motion at %g K, seeded: per 1 fs frame every atom moves %.4f A rms (max %.4f A); '
                'trajectory time = triangle wave of +-%d fs' % (TEMP_K, float((d ** 2).mean().sqrt()), float(d.max()),
                                                               EXCURSION))


def make_force(w, comm=None, outputs='replicated'):
    import torch
    from admp_amd import settings
    from admp_amd.pme import ADMPPmeForce
    settings.PRECISION = w['prec']
    if comm is None:
        f = ADMPPmeForce(w['box'], w['at'], w['ai'], w['cov'], RC, 1e-4, 2, lpol=True)
    else:
        from admp_amd.parallel import SlabPme
        f = SlabPme(comm, w['box'], w['at'], w['ai'], w['cov'], RC, 1e-4, 2, lpol=True, outputs=outputs)
    if w['K'] is not None:
        for k in ('K1', 'K2', 'K3'):
            f.update_env(k, w['K'])
    dt = torch.float32 if w['prec'] == 'single' else torch.float64
    dev = torch.device('cuda', torch.cuda.current_device())
    par = w['par']
    # inputs resident in HBM before the timed region
    args = dict(positions=torch.as_tensor(w['pos'], dtype=dt, device=dev), box=w['box'],
                pairs=w['pairs'].to(device=dev, dtype=torch.int32),
                Q_local=torch.as_tensor(par['Q_local'], dtype=dt, device=dev),
                pol=torch.as_tensor(par['pol'], dtype=dt, device=dev),
                tholes=torch.as_tensor(par['tholes'], dtype=dt, device=dev),
                mScales=par['mScales'], pScales=par['pScales'], dScales=par['dScales'])
    return f, args


def step(f, a, U, positions=None, pairs='fixed'):
    return f.get_forces(a['positions'] if positions is None else positions, a['box'],
                        a['pairs'] if pairs == 'fixed' else pairs, a['Q_local'], a['pol'], a['tholes'], a['mScales'],
                        a['pScales'], a['dScales'], U_init=U)


def run_timed(f, a, steps, warmup, frames=None, barrier=None, only='pair_full', thresh=None, predictor=False):
    """Timed region.  frames = ThermalFrames: step k evaluates frame k of the moving trajectory (None: the same geometry
    every step).  Only the roofline kernel is bracketed by HIP events (two event records per launch); the full
    per-kernel breakdown comes from `kernel_breakdown` afterwards, outside the timed region.
    predictor: start the SCF of step n from 2 U(n-1) - U(n-2) instead of U(n-1) (a labelled extra, not the headline: the
    reference's drivers pass the previous step's dipoles, examples/water_pol_1024/run_admp.py:139); the two small tensor
    operations are inside the timed region."""
    import torch
    from admp_amd import settings
    old = settings.POL_CONV
    if thresh is not None:
        settings.POL_CONV = thresh
    try:
        seq = [frames.step_frame(k) if frames is not None else None for k in range(warmup + steps)]   # resident in HBM
        U = None
        U1 = U0 = None            # predictor: the converged dipoles of the two previous steps

        def start():
            return (2.0 * U1 - U0) if (predictor and U0 is not None) else U
        for k in range(warmup):
            step(f, a, start(), seq[k])
            U = f.U_ind
            if predictor:
                U0, U1 = U1, U.clone()
        f.profile(only is not False, only=only if only else None)
        f.profile_reset()
        torch.cuda.synchronize()
        import gc
        gc.collect()
        gc.disable()               # (a generation-2 collection of the interpreter inside the region shows up as a 4 ms step:
        try:                       #  r04z, 0.206 instead of 0.186 ms per step over 200 steps)
            if barrier:
                barrier()
            t0 = time.perf_counter()
            updates = 0
            marks = [t0]
            for k in range(warmup, warmup + steps):
                step(f, a, start(), seq[k])
                U = f.U_ind
                updates += f.n_cycle
                if predictor:
                    U0, U1 = U1, U.clone()
                marks.append(time.perf_counter())      # (a call returns after its own host read of the energies)
            torch.cuda.synchronize()
            if barrier:
                barrier()
            dt = time.perf_counter() - t0
        finally:
            gc.enable()
        rep = f.profile_report()
        f.profile(False)
    finally:
        settings.POL_CONV = old
    # n_cycle = Jacobi updates done before the check that passed; every update is preceded by one field evaluation
    per = np.diff(np.asarray(marks)) * 1e3
    return dt, rep, {'jacobi_updates_per_step': round(updates / float(steps), 3),
                     'scf_field_evaluations_per_step': round(updates / float(steps) + 1.0, 3),
                     'step_ms_min_median_max': [round(float(per.min()), 4), round(float(np.median(per)), 4),
                                                round(float(per.max()), 4)]}


def kernel_breakdown(f, a, frames, first, steps=10):
    """(ms per step, launches per step, raw report) of every kernel label.  All launches are bracketed and the side stream is
    OFF for this pass (admp_set_option ADMP_OPT_SIDE_STREAM): an event pair around a kernel that shares the chip with another
    stream's kernels does not measure its duration (round-3 verdict: 29 us reported for a 14.7 us pair kernel), so the
    per-kernel figures -- and the roofline's launch time at S1 -- come from serialised launches and agree with the rocprofv3
    averages in profiles/; the steps of this pass are slower than the timed region's."""
    U = f.U_ind
    f.set_side_stream(False)
    f.profile(True)
    f.profile_reset()
    for k in range(steps):
        step(f, a, U, frames.step_frame(first + k) if frames is not None else None)
        U = f.U_ind
    rep = f.profile_report()
    f.profile(False)
    f.set_side_stream(True)
    return ({k: round(v[0] / steps, 5) for k, v in sorted(rep.items())},
            {k: round(v[1] / float(steps), 3) for k, v in sorted(rep.items())}, rep)


def time_list_rebuild(f, w, reps=3):
    """Neighbour search + compilation of the pair table (the step either side of the path), ms per rebuild."""
    import torch
    dt = torch.float32 if w['prec'] == 'single' else torch.float64
    p = torch.as_tensor(w['pos'], dtype=dt, device='cuda')
    f.update_neighbors(p, w['box'])
    best = float('inf')
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        f.update_neighbors(p, w['box'])          # cell-list search fused with the table build
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best * 1e3


VALU_PEAK_GINST = 1024 * 2.4 / 2.0     # wave64 VALU instructions per ns the chip can issue: 256 CUs x 4 SIMD-32s, a wave64
                                       # instruction issues over 2 cycles, 2.4 GHz (MI355X_MICROARCH.md:14-15,34,54) = 1228.8 G/s


def event_bracket_overhead_us(n=200):
    """What an EMPTY pair of HIP events reads on the library's stream (torch's current stream): the part of every event-timed
    launch duration that is not the kernel (median of n brackets, us).  rocprofv3's kernel durations do not contain it (it is
    not simply additive either: with a kernel in between, part of it overlaps the kernel's own dispatch)."""
    import torch
    a = [torch.cuda.Event(enable_timing=True) for _ in range(n)]
    b = [torch.cuda.Event(enable_timing=True) for _ in range(n)]
    torch.cuda.synchronize()
    for i in range(n):
        a[i].record()
        b[i].record()
    torch.cuda.synchronize()
    return float(np.median([a[i].elapsed_time(b[i]) for i in range(n)])) * 1e3


def roofline_of(rep, w, n_pairs, rep_serial=None, share=1.0):
    """Roofline of the dominant kernel, k_pair_full, in the contract's form: `achieved` = ALGORITHMIC bytes per launch
    (bytes per pair x pairs) / average launch duration, against the HBM peak; `traffic` = measured HBM bytes per
    launch (rocprofv3 PMC passes of this command, profiles/pmc_traffic.json).

    Launch duration: HIP events around every launch.  Up to 200 000 atoms the figure is taken from `rep_serial`, the breakdown
    pass right after the timed region (same process, same frames, side stream off): between 4096 and 200 000 atoms the kernel
    runs on a SIDE stream next to the mesh chain, so events in the timed region would bracket a kernel that shares the chip;
    below 4096 atoms (one stream) two event records per launch cost 6 % of a 0.19 ms step, so the timed region carries none and
    the pass after it measures the same launches.  Above 200 000 atoms the timed region's own events are used.

    The byte model is not the kernel's limiter (its partner rows come from L2 / MALL: measured traffic is a quarter of the
    algorithmic bytes, and at 1M atoms the algorithmic rate exceeds what HBM can deliver); `valu_view` gives the view that
    is: SQ_INSTS_VALU per launch / launch time against the wave-instruction issue rate of 1024 SIMDs.
    share: fraction of the pairs this rank evaluates (slab ranks: home atoms / atoms; the algorithmic bytes scale with it)."""
    wbytes = 4 if w['prec'] == 'single' else 8
    n_atoms = 3 * w['n_mol']
    total, per_pair = pair_kernel_bytes(n_pairs, n_atoms, wbytes, True)
    total *= share
    ms_t, cnt_t = rep.get('pair_full', (0.0, 0)) if rep else (0.0, 0)
    quiet = n_atoms <= int(os.environ.get('ADMP_OVERLAP_MAX', '200000')) and share == 1.0      # no events in the timed region
    concurrent = quiet and n_atoms >= int(os.environ.get('ADMP_OVERLAP_MIN', '4096'))           # ... because of the side stream
    src = rep_serial if (quiet and rep_serial and rep_serial.get('pair_full', (0, 0))[1]) else rep
    ms, cnt = src.get('pair_full', (0.0, 0)) if src else (0.0, 0)
    avg_s = (ms / cnt) * 1e-3 if cnt else float('nan')
    alg = total / avg_s / 1e9 if cnt else float('nan')
    traffic = tag = valu = n_trans = n_f64 = None
    tfile = os.path.join(ROOT, 'profiles', 'pmc_traffic.json')
    if os.path.exists(tfile) and share == 1.0:
        try:
            rec = json.load(open(tfile)).get(w['name'], {})
            traffic = rec.get('pair_full_bytes_per_launch')
            valu = rec.get('pair_full_valu_insts_per_launch')
            n_trans = rec.get('pair_full_valu_trans_insts_per_launch')
            n_f64 = rec.get('pair_full_valu_f64_insts_per_launch')
            tag = rec.get('measured_at')
        except Exception:
            traffic = None
    out = {'kernel': 'k_pair_full', 'bound': 'hbm', 'achieved': round(alg, 2), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
           'frac': round(alg / HBM_PEAK_GBS, 5), 'traffic': traffic,
           'avg_launch_us': round(avg_s * 1e6, 2), 'launches': int(cnt),
           'launch_time_source': (('HIP events, breakdown pass with the side stream off (the kernel shares the chip in the timed '
                                   'region)' if concurrent else
                                   'HIP events, breakdown pass right after the timed region (same single stream; two event '
                                   'records per launch inside the timed region cost 6 % of a step of this dispatch-bound workload)')
                                  if src is rep_serial and src is not rep else 'HIP events over the timed region'),
           'algorithmic_bytes_per_launch': int(total), 'bytes_per_pair': round(per_pair, 2),
           'note': 'achieved = algorithmic bytes / launch time (contract); the kernel is not HBM-bound: see traffic, valu_view'}
    try:
        ov = event_bracket_overhead_us()
        # an EMPTY event pair reads this much on the same stream: event timing of a 15 us kernel cannot agree with a
        # profiler's kernel duration better than a fraction of it (S1: events 15-17 us, rocprofv3 14.7); at S3 it is 2 %
        out['event_bracket_overhead_us'] = round(ov, 2)
    except Exception:
        pass
    if quiet and cnt_t:
        out['avg_launch_us_timed_region'] = round(ms_t / cnt_t * 1e3, 2)
    if concurrent:
        out['timed_region'] = 'no events inside the timed region: this kernel runs on a side stream next to the mesh chain there'
    elif quiet:
        out['timed_region'] = 'no events inside the timed region (they cost 6 % of a step here); one stream, as in the breakdown pass'


    if share != 1.0:
        out['pair_share_of_this_rank'] = round(share, 5)
    if valu and cnt:
        rate = valu / avg_s / 1e9
        vv = {'achieved': round(rate, 2), 'peak': VALU_PEAK_GINST, 'unit': 'G wave-instructions/s',
              'frac': round(rate / VALU_PEAK_GINST, 5), 'valu_insts_per_launch': int(valu)}
        if n_trans is not None and n_f64 is not None:
            # not every instruction issues in 2 cycles: transcendentals (exp / rcp / rsq / sqrt) take 8, f64 arithmetic 4
            # (MI355X_MICROARCH.md constants table; f64 vector peak = half of f32).  The issue cycles the kernel's instruction
            # mix needs per SIMD against the SIMD cycles of the launch (2.4 GHz):
            cyc = 2.0 * (valu - n_trans - n_f64) + 8.0 * n_trans + 4.0 * n_f64
            vv['issue_cycles_frac'] = round(cyc / 1024.0 / (avg_s * 2.4e9), 5)
            vv['mix'] = {'transcendental': int(n_trans), 'f64': int(n_f64)}
        out['valu_view'] = vv
    if traffic:
        out['traffic_source'] = 'profiles/pmc_traffic.json (rocprofv3 --pmc passes of this command%s), not collected in this run' % (
            ', ' + tag if tag else '')
        out['hbm_traffic_frac'] = round(traffic / avg_s / 1e9 / HBM_PEAK_GBS, 5) if cnt else None
    return out


TRANSFORM_PREFIXES = ('rocfft_', 'dft_', 'fftx_', 'kspace')
KSPACE_LABELS = ('kspace', 'dft_x_kspace', 'fftx_kspace')      # exactly one launch of these per mesh convolution


def recip_kernel_rooflines(kb, counts, w, grid):
    """Achieved algorithmic HBM rate of the spread / gather / convolution legs (SURVEY.md 8d formulas).  The labels are taken
    from the breakdown's own keys (every key with a transform prefix is consumed: a renamed kernel cannot drop out silently,
    as dft_zy_fwd / dft_yz_inv did in round 3) and the number of launches from the profiler's counts, not from an assumed
    number of SCF evaluations."""
    wb = 4 if w['prec'] == 'single' else 8
    na = 3 * w['n_mol']
    K3 = grid[0] * grid[1] * grid[2]
    Kh = grid[0] * grid[1] * (grid[2] // 2 + 1)
    out = {}

    def add(name, labels, nbytes, passes, note=None):
        ms_step = sum(kb.get(k, 0.0) for k in labels)
        if ms_step > 0 and passes > 0:
            ms = ms_step / passes
            gbs = nbytes / (ms * 1e-3) / 1e9
            out[name] = {'labels': sorted(labels), 'algorithmic_bytes': int(nbytes), 'passes_per_step': round(passes, 3),
                         'ms_per_pass': round(ms, 5), 'achieved_GBs': round(gbs, 1),
                         'frac_of_hbm_peak': round(gbs / HBM_PEAK_GBS, 4)}
            if note:
                out[name]['note'] = note
    # permanent multipoles + dipoles of all atoms (once per evaluation that spreads them); the SCF increments spread the
    # dipole changes of the polarizable sites only (a third of the atoms, 3 of the 15 words): their own line
    add('spread', ['spread'], na * 15 * wb + K3 * wb, counts.get('spread', 0.0))
    add('spread_scf_increment', ['spread_ind'], (na // 3) * 6 * wb + K3 * wb, counts.get('spread_ind', 0.0))
    if 'dft_spread_zy_fwd' in kb:
        # small double-precision systems on the direct-DFT plane kernels (round 4): no spread kernel and no charge mesh in
        # memory -- the forward transform's workgroups build their x planes in LDS from the site rows (dft_kernels.hip)
        out['spread'] = {'fused_into': 'dft_spread_zy_fwd', 'launches_per_step': counts.get('dft_spread_zy_fwd', 0.0),
                         'note': 'the spread is the first phase of the forward plane transform; its time is part of '
                                 'transforms+kspace, which moves no spread bytes through HBM'}
    add('gather', ['gather'], K3 * wb + na * 24 * wb, counts.get('gather', 0.0))
    tl = [k for k in kb if k.startswith(TRANSFORM_PREFIXES)]
    conv = sum(counts.get(k, 0.0) for k in KSPACE_LABELS)
    add('transforms+kspace', tl, 12 * Kh * 2 * wb + Kh * 5 * wb, conv,
        'one pass = one mesh convolution (forward transform, G multiply, inverse transform)')
    unused = [k for k in kb if k.startswith(TRANSFORM_PREFIXES) and k not in tl]
    assert not unused, unused
    return out


def f32_vs_f64_force_error(w, f32_force, a32, pos64):
    """Relative L2 difference of the single-precision gradient to the double-precision gradient of the same HIP path on
    the same inputs (both from a cold SCF start).  The oracle cannot run at this size; the f64 path is parity-tested
    against it at small sizes, so this is the f32 error bar of the large configuration (bar: 1e-2)."""
    import torch
    w64 = dict(w, prec='double')
    f64, a64 = make_force(w64)
    E64, G64 = step(f64, a64, None, pos64.double())
    E32, G32 = step(f32_force, a32, None, pos64.float())
    g64 = torch.as_tensor(G64).double()
    g32 = torch.as_tensor(G32).double().to(g64.device)
    err = float(torch.linalg.norm(g32 - g64) / torch.linalg.norm(g64))
    return {'force_rel_l2_f32_vs_f64': err, 'energy_rel_f32_vs_f64': float(abs(E32 - E64) / abs(E64)),
            'scf_cycles_f32_f64': [int(f32_force.n_cycle), int(f64.n_cycle)]}


def md_all_terms(w, f, a, frames, steps, warmup, rebuild=10, prune=5):
    """What one step of examples/md/nve_water.py costs on the moving frames: polarizable PME (SCF warm-started) +
    dispersion PME (pmax 10) + Tang-Toennies, every calculator on a Verlet list of rc + 1 A skin that is rebuilt on the
    GPU from the positions every `rebuild` steps (inside the timed region).  ms per step and per term.  Large systems get one
    more leg, `inner_list`: the same loop with the list pruned to rc + skin * prune / rebuild every `prune` steps
    (admp_prune_pairs: the multipolar kernels evaluate every listed pair, so a shorter list is less work; below ~200k atoms the
    pruning passes cost more than they save)."""
    import torch
    from admp_amd.disp_pme import ADMPDispPmeForce
    from admp_amd.pairwise import generate_pairwise_interaction, TT_damping_qq_c6_kernel, value_and_grad
    from admp_amd import settings
    settings.PRECISION = w['prec']      # the calculators built here take it from the module global (make_force of another
    dt = torch.float32 if w['prec'] == 'single' else torch.float64      # workload may have left it at the other precision)
    par = w['par']
    cl = torch.as_tensor(par['c_list'], dtype=dt, device='cuda')
    disp = ADMPDispPmeForce(w['box'], w['cov'], RC, 1e-4, 10)
    if w['K'] is not None:
        for k in ('K1', 'K2', 'K3'):
            disp.update_env(k, w['K'])
    tt_obj = generate_pairwise_interaction(TT_damping_qq_c6_kernel, w['cov'], static_args={})
    tt = value_and_grad(tt_obj)
    a_, b_, q_ = (torch.as_tensor(par[k], dtype=dt, device='cuda') for k in ('a_list', 'b_list', 'q_list'))
    c6 = cl[:, 0].contiguous()
    mS = par['mScales']
    seq = [frames.step_frame(k) for k in range(warmup + steps)]

    inner = RC + SKIN * prune / float(rebuild)
    use_prune = [False]

    def rebuild_lists(p):
        # one search + one compiled table for the three calculators (the reference's drivers hand them one `pairs` array)
        f.update_neighbors(p, w['box'], rc=RC + SKIN)
        for obj in (disp, tt_obj):
            obj.share_neighbors(f)

    def one(k, U, terms=(1, 1, 1)):
        p = seq[k]
        if k >= warmup and (k - warmup) % rebuild == 0:      # the timed region starts with a rebuild: one per `rebuild` steps
            rebuild_lists(p)
        if use_prune[0] and k >= warmup and (k - warmup) % prune == 0:      # (inner_list leg) every `prune` steps, also right
            f.prune_neighbors(p, w['box'], inner)                           # after a rebuild
        g = None
        if terms[0]:
            _, g = step(f, a, U, p, pairs=None)
        if terms[1]:
            _, g2 = disp.get_forces(p, w['box'], None, cl, mS)
            g = g2 if g is None else g.add_(g2)
        if terms[2]:
            _, g3 = tt(p, w['box'], None, mS, a_, b_, q_, c6)
            g = g3 if g is None else g.add_(g3)
        return f.U_ind if terms[0] else U

    def timed(terms):
        import gc
        U = None
        rebuild_lists(seq[0])
        for k in range(warmup):
            U = one(k, U, terms)
        torch.cuda.synchronize()
        gc.collect()
        gc.disable()                   # (a generation-2 collection inside a 5-step region would show up as a stall)
        try:
            t0 = time.perf_counter()
            marks = [t0]
            upd = 0
            for k in range(warmup, warmup + steps):
                U = one(k, U, terms)
                upd += f.n_cycle if terms[0] else 0
                marks.append(time.perf_counter())
            torch.cuda.synchronize()
            t1 = time.perf_counter()
        finally:
            gc.enable()
        per = np.diff(np.asarray(marks)) * 1e3
        return (t1 - t0) / steps * 1e3, upd / float(steps), [round(float(per.min()), 4), round(float(np.median(per)), 4),
                                                             round(float(per.max()), 4)]
    ms_all, upd, spread = timed((1, 1, 1))
    out = {'ms_per_step': round(ms_all, 4), 'ns_per_day': round(0.0864 / (ms_all * 1e-3), 3),
           'steps': steps, 'step_ms_min_median_max': spread,
           'jacobi_updates_per_step': round(upd, 3), 'n_pairs_with_skin': int(f.n_pairs),
           'list': 'rc %.1f + skin %.1f A, rebuilt every %d steps; the timed region of %d steps starts with a rebuild' % (
               RC, SKIN, rebuild, steps),
           'pme_ms': round(timed((1, 0, 0))[0], 4), 'dispersion_pme_pmax10_ms': round(timed((0, 1, 0))[0], 4),
           'tang_toennies_ms': round(timed((0, 0, 1))[0], 4)}
    if prune and 3 * w['n_mol'] >= 200000:
        use_prune[0] = True
        try:
            ms_p, upd_p, spread_p = timed((1, 1, 1))
            out['inner_list'] = {'note': 'EXTRA: the same loop on an inner list of rc + %.2f A pruned from the skin list every %d '
                                         'steps (admp_prune_pairs; energy drift of an NVE run is larger with it: the multipolar '
                                         'kernels see a set of pairs beyond rc that changes at every prune)' % (inner - RC, prune),
                                 'ms_per_step': round(ms_p, 4), 'ns_per_day': round(0.0864 / (ms_p * 1e-3), 3),
                                 'step_ms_min_median_max': spread_p, 'n_pairs_inner': int(f.n_pairs)}
        finally:
            use_prune[0] = False
    f.set_pairs(a['pairs'])          # back to the fixed rc list of the headline
    return out


def slab_one_rank_ms(w, frames, steps=5, warmup=2):
    """ms per step of SlabPme on a ONE-rank communicator (what a multi-GPU driver pays on top of the kernels when there is
    nobody to talk to), same moving frames as the plain calculator's leg."""
    from admp_amd.parallel import ThreadComm
    f1, a1 = make_force(w, ThreadComm(ThreadComm.World(1), 0))
    dt1, _, _ = run_timed(f1, a1, steps, warmup, frames, only=False)
    return round(dt1 / steps * 1e3, 4)


def cpu_baseline(w, pos_prev_U, pos_frame):
    """The float64 oracle (CPU restatement of the reference's algorithm) on one frame of the same trajectory, its SCF
    warm-started from the converged dipoles of the previous frame (what the timed GPU step does)."""
    import torch
    from oracle import admp_oracle as O
    from admp_amd.pme import setup_ewald_parameters
    kappa, K1, K2, K3 = setup_ewald_parameters(RC, 1e-4, w['box'])
    if w['K'] is not None:
        K1 = K2 = K3 = w['K']
    par = w['par']
    sysm = O.PmeSystem(w['at'], w['ai'], w['cov'], kappa, (K1, K2, K3), 2, True)
    t0 = time.perf_counter()
    r = O.pme_energy_and_grad(sysm, pos_frame, w['box'], w['pairs'].cpu().numpy(), par['Q_local'], par['mScales'], par['pol'],
                              par['tholes'], par['pScales'], U_init=pos_prev_U)
    dt = time.perf_counter() - t0
    return {'value': round(0.0864 * DT_FS / dt, 6), 'unit': 'ns/day', 'cores': int(torch.get_num_threads()),
            'kind': 'port', 'sample': '1 get_forces call on one frame of the %s trajectory, SCF warm-started from the previous '
            'frame\'s converged dipoles (%d field evaluations, as on the GPU), %.1f s wall; torch-CPU float64 restatement of '
            'the reference algorithm (the reference\'s JAX path is not installable offline)' % (w['name'], r['n_cycle'] + 1, dt)}, r


def reduce_max_seconds(dt, dist, device):
    """MAX over ranks of the timed region (every rank returns the same number)."""
    if dist is None:
        return float(dt)
    import torch
    t = torch.tensor([dt], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def aggregate_ns_per_day(t_step_s, world):
    """Whole-job throughput: `world` independent replicas each advance dt per step."""
    return world * 0.0864 * DT_FS / t_step_s


def launch_ranks(n, argv):
    """`python bench.py --gpus N` without a launcher: this process has not touched the GPU (torch is not even imported yet);
    it starts N fresh child ranks (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in their environment, one per device) and
    relays rank 0's JSON line.  Any rank failing makes the run fail: the others are ended and the exit code is non-zero."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(('127.0.0.1', 0))
        port = so.getsockname()[1]
    import threading
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY='0')
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    rc = 0
    try:
        live = list(procs)
        while live and rc == 0:            # a rank that fails ends the run: its peers would wait in a collective for ever
            time.sleep(0.2)
            for pr in list(live):
                code = pr.poll()
                if code is not None:
                    live.remove(pr)
                    rc = rc or code
    finally:
        for pr in procs:
            if pr.poll() is None:
                pr.kill()                  # (the exact processes started above)
        for pr in procs:
            pr.wait()
    reader.join(timeout=10)
    if rc == 0:
        sys.stdout.write(b''.join(chunks).decode(errors='replace'))
        sys.stdout.flush()
    else:
        sys.stderr.write('bench.py: a rank exited with code %d; no result line\n' % rc)
    return rc


def selftest_rank(rank, world):
    """ADMP_BENCH_SELFTEST (CPU test of the launcher, tests/test_distributed_cpu.py): rendezvous over gloo, one barrier, rank 0
    prints a line; 'failN' makes rank N exit non-zero before the barrier."""
    mode = os.environ['ADMP_BENCH_SELFTEST']
    if mode == 'fail%d' % rank:
        return 3
    import torch.distributed as dist
    dist.init_process_group('gloo')
    dist.barrier()
    if rank == 0:
        print(json.dumps({'selftest': True, 'n_gpus': world, 'master_port': os.environ.get('MASTER_PORT')}))
    dist.barrier()
    dist.destroy_process_group()
    return 0


def main_slab(opt, rank, world):
    """N > 1: the slab-decomposed step of BASELINE configs[3] over the N ranks (see the module docstring)."""
    import datetime
    import torch
    import torch.distributed as dist
    from admp_amd.parallel import make_comm
    ndev = max(torch.cuda.device_count(), 1)
    local = int(os.environ.get('LOCAL_RANK', str(rank))) % ndev      # rehearsal on one GPU: the ranks share it (gloo only)
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    backend = os.environ.get('ADMP_BENCH_BACKEND', 'nccl')           # 'gloo' = host-staged rehearsal on one GPU
    if backend == 'nccl':
        dist.init_process_group('nccl', device_id=dev, timeout=datetime.timedelta(seconds=300))
    else:
        dist.init_process_group(backend, timeout=datetime.timedelta(seconds=300))
    rdev = 'cuda' if backend == 'nccl' else 'cpu'
    name = os.environ.get('ADMP_BENCH_SCALE', 'S3')
    w = make_workload(name)
    comm = make_comm()
    outputs = os.environ.get('ADMP_BENCH_OUTPUTS', 'home')
    f, a = make_force(w, comm, outputs)
    # the rank's neighbour table from the positions (cell list on the GPU): only the rows near its slab are built
    # (ADMP_SLAB_ROWS; the same pairs as the workload's explicit rc list, which a rank would have to compile in full)
    n_pairs_all = int(len(w['pairs']))
    f.update_neighbors(a['positions'], w['box'])
    a['pairs'] = None
    frames = ThermalFrames(w, dev)
    dt, _, cyc = run_timed(f, a, opt.steps, opt.warmup, frames, dist.barrier, only=False)
    dt = reduce_max_seconds(dt, dist, rdev)
    t_step = dt / opt.steps
    # per-kernel and per-collective times of this rank (HIP events around every launch / collective: a pass of its own)
    comm.reset_stats()
    if hasattr(comm, 'profile'):
        comm.profile = True
    nb = min(opt.steps, 5)
    kb, counts, rep = kernel_breakdown(f, a, frames, opt.warmup + opt.steps, nb)
    if hasattr(comm, 'refresh_stats'):
        comm.refresh_stats()
    sent = {k: int(v / nb) for k, v in sorted(comm.bytes_sent.items())}
    coll_py = comm.report(nb)
    n_atoms = 3 * w['n_mol']
    # what the reference's calling convention (every caller receives the full gradient / dipole arrays) costs on top: one SUM
    # all-reduce of (Na, 3) per output array and call -- the same object with outputs='replicated', a short leg of its own
    replicated = None
    if not opt.no_extras and outputs == 'home':
        try:
            f.outputs = 'replicated'
            kr = min(opt.steps, 5)
            dtr, _, _ = run_timed(f, a, kr, 1, frames, dist.barrier, only=False)
            dtr = reduce_max_seconds(dtr, dist, rdev)
            replicated = {'note': 'EXTRA: the same step with outputs=\'replicated\' (full arrays on every rank, as the reference\'s '
                                  'single-device API returns them)', 'ms_per_step': round(dtr / kr * 1e3, 5),
                          'ns_per_day': round(0.0864 / (dtr / kr), 3), 'steps': kr}
        except Exception as e:
            replicated = {'error': repr(e)}
        finally:
            f.outputs = outputs
    # the labelled extra: N independent replicas of the single-GPU headline workload (what a 3072-atom box can do with N GPUs)
    replicas = None
    if not opt.no_extras:
        try:
            w1 = make_workload('S1')
            f1, a1 = make_force(w1)
            fr1 = ThermalFrames(w1, dev)
            k1 = min(opt.steps, 50)
            dt1, _, _ = run_timed(f1, a1, k1, opt.warmup, fr1, dist.barrier, only=False)
            dt1 = reduce_max_seconds(dt1, dist, rdev)
            replicas = {'note': 'EXTRA, not the metric: every rank steps an independent replica of S1 (3072 atoms, f64)',
                        'aggregate_ns_per_day': round(aggregate_ns_per_day(dt1 / k1, world), 3),
                        'ms_per_step': round(dt1 / k1 * 1e3, 5), 'scaling': 'weak'}
        except Exception as e:       # the extra must not take the metric down
            replicas = {'error': repr(e)}
    # the SAME workload on one GPU (rank 0 alone, the others wait): what `value` has to be compared with -- the driver's own
    # N = 1 run measures the S1 headline, another workload
    single = None
    if not opt.no_extras:
        try:
            if rank == 0:
                fs, as_ = make_force(w)
                ks = min(opt.steps, 20)
                dts, _, cycs = run_timed(fs, as_, ks, min(opt.warmup, 5), ThermalFrames(w, dev), only=False)
                single = {'note': 'the same workload and frames on ONE GPU (rank 0 alone): the strong-scaling reference of `value`',
                          'ms_per_step': round(dts / ks * 1e3, 5), 'ns_per_day': round(0.0864 * DT_FS / (dts / ks), 4), 'steps': ks}
                single.update(cycs)
                fs = as_ = None
                torch.cuda.empty_cache()
        except Exception as e:
            single = {'error': repr(e)}
        dist.barrier()
    if rank == 0:
        comm_ms = {k: v for k, v in kb.items() if k.startswith('comm_')}
        kern_ms = {k: v for k, v in kb.items() if not k.startswith('comm_')}
        cfg = {'workload': w['desc'], 'n_atoms': n_atoms, 'n_pairs': n_pairs_all, 'pme_grid': [f.K1, f.K2, f.K3],
               'kappa': round(float(f.kappa), 6), 'dt_fs': DT_FS, 'geometry': 'MOVING: ' + frames.describe(),
               'scf': 'Jacobi, warm-started from the previous step\'s dipoles, POL_CONV = 10 kJ/mol/(e A)',
               'pair_list': 'fixed during the timed steps; every rank holds the table rows near its slab '
                            '(%d of the %d pairs on rank 0)' % (int(f.n_pairs), n_pairs_all),
               'parallelism': 'x-slab decomposition of mesh and space over %d ranks (one process per GPU); collectives: %s; '
                              'outputs: %s rows' % (world, 'RCCL issued by the library (ncclAllReduce, grouped ncclSend/ncclRecv)'
                                                    if getattr(comm, 'native_rccl', False) else
                                                    'torch.distributed %s through the callback interface' % backend, outputs),
               'home_atoms_rank0': int(f.n_home), 'import_atoms_rank0': int(f.n_import)}
        cfg.update(cyc)
        out = {'metric': 'ns/day (electrostatics get_forces per 1 fs step, polarizable PME incl. SCF, moving atoms)',
               'value': round(0.0864 * DT_FS / t_step, 4), 'unit': 'ns/day', 'n_gpus': world, 'steps': opt.steps,
               'warmup': opt.warmup, 'ms_per_step': round(t_step * 1e3, 5), 'higher_is_better': True, 'scaling': 'strong',
               'vs_baseline': None, 'dtype': 'f32' if w['prec'] == 'single' else 'f64', 'data': 'synthetic', 'config': cfg,
               'roofline': roofline_of(rep, w, n_pairs_all, None, share=f.n_home / float(n_atoms)),
               'rank0_kernel_ms_per_step': kern_ms,
               'rank0_kernel_ms_sum': round(sum(kern_ms.values()), 4),
               'rank0_collective_ms_per_step': comm_ms if comm_ms else coll_py,
               'rank0_collective_ms_sum': round(sum((comm_ms if comm_ms else coll_py).values()), 4),
               'rank0_launches_per_step': counts,
               'rank0_bytes_sent_per_step': sent,
               'recip_kernels_rank0': None,
               'cpu_baseline': None}
        if single is not None:
            out['single_gpu_same_workload'] = single
            if 'ms_per_step' in single:
                out['speedup_vs_single_gpu'] = round(single['ms_per_step'] / (t_step * 1e3), 4)
        if replicated is not None:
            out['outputs_replicated'] = replicated
        if replicas is not None:
            out['replicas_S1'] = replicas
        print(json.dumps(out))
        sys.stdout.flush()
    dist.barrier()
    dist.destroy_process_group()
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    # defaults: the S1 loop settles within ~60 steps of a fresh process (clocks, caches, the SCF-form history: tools/s1_settle.py
    # -- steps 20-59 run 2 % slower than the steady state); 20 + 200 steps of 0.2 ms keep the default run short
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--workload', default='S1', choices=sorted(WORKLOADS))
    ap.add_argument('--no-scale', action='store_true', help='skip the extra 1M-atom measurement (N=1 only)')
    ap.add_argument('--no-cpu', action='store_true', help='skip the CPU baseline leg')
    ap.add_argument('--no-extras', action='store_true', help='headline only (no scf_tight / static / md_all_terms legs)')
    opt = ap.parse_args()

    if 'WORLD_SIZE' not in os.environ and opt.gpus > 1:
        return launch_ranks(opt.gpus, sys.argv[1:])           # (nothing in this process has touched the GPU)
    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if os.environ.get('ADMP_BENCH_SELFTEST'):
        return selftest_rank(rank, world)
    if world > 1:
        return main_slab(opt, rank, world)

    import torch
    torch.cuda.set_device(0)
    dev = torch.device('cuda', 0)
    w = make_workload(opt.workload)
    f, a = make_force(w)
    frames = ThermalFrames(w, dev)
    n_atoms = 3 * w['n_mol']

    # events around the roofline kernel inside the timed region only where they measure its duration (it runs alone above
    # 200 000 atoms); below, the kernel shares the chip with the mesh chain and is timed in the breakdown pass instead
    # (up to 200 000 atoms the timed region carries no events: from 4096 atoms on the kernel runs on a side stream next to the
    # mesh chain, and below that two event records per launch cost 6 % of a 0.19 ms step -- 0.2046 against 0.1925 ms at S1)
    side = n_atoms <= int(os.environ.get('ADMP_OVERLAP_MAX', '200000'))
    dt, rep, cyc = run_timed(f, a, opt.steps, opt.warmup, frames, only=False if side else 'pair_full')
    t_step = dt / opt.steps
    value = aggregate_ns_per_day(t_step, 1)
    last = opt.warmup + opt.steps            # index of the next frame of the trajectory
    U_last = f.U_ind
    kb, counts, rep_serial = kernel_breakdown(f, a, frames, last)
    head = dict(n_pairs=int(f.n_pairs), grid=[f.K1, f.K2, f.K3], kappa=round(float(f.kappa), 6),
                roofline=roofline_of(rep, w, f.n_pairs, rep_serial), kernels=kb, counts=counts)
    head['rebuild_ms'] = time_list_rebuild(f, w)
    f.set_pairs(a['pairs'])

    cfg = {'workload': w['desc'], 'n_atoms': n_atoms, 'n_pairs': head['n_pairs'],
           'pme_grid': head['grid'], 'kappa': head['kappa'], 'dt_fs': DT_FS,
           'geometry': 'MOVING: ' + frames.describe(),
           'scf': 'Jacobi, warm-started from the previous step\'s dipoles, POL_CONV = 10 kJ/mol/(e A) '
                  '(reference default, admp/settings.py:29)',
           'pair_list': 'fixed during the timed steps; GPU cell-list search + table compile = %.3f ms per rebuild '
                        '(value with a rebuild every 10 steps: %.2f ns/day)' % (
                            head['rebuild_ms'], 0.0864 / (t_step + head['rebuild_ms'] * 1e-4)),
           'parallelism': 'single GPU'}
    cfg.update(cyc)
    out = {
        'metric': 'ns/day (electrostatics get_forces per 1 fs step, polarizable PME incl. SCF, moving atoms)',
        'value': round(value, 4), 'unit': 'ns/day', 'n_gpus': 1, 'steps': opt.steps, 'warmup': opt.warmup,
        'ms_per_step': round(t_step * 1e3, 5), 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
        'dtype': 'f64' if w['prec'] == 'double' else 'f32', 'data': 'synthetic',
        'config': cfg,
        'roofline': head['roofline'],
        'kernel_ms_per_step': head['kernels'],
        'kernel_launches_per_step': head['counts'],
        'kernel_ms_note': 'breakdown pass: every launch bracketed by HIP events, side stream off (steps slower than the timed ones)',
        'recip_kernels': recip_kernel_rooflines(head['kernels'], head['counts'], w, head['grid']),
    }
    if not opt.no_extras:
        try:
            k2 = min(opt.steps, 20)
            dt2, _, cyc2 = run_timed(f, a, k2, opt.warmup, frames, only=False, thresh=1e-2)
            out['scf_tight'] = dict(thresh=1e-2, ms_per_step=round(dt2 / k2 * 1e3, 5),
                                    ns_per_day=round(0.0864 / (dt2 / k2), 3), steps=k2, **cyc2)
            dtp, _, cycp = run_timed(f, a, k2, opt.warmup, frames, only=False, predictor=True)
            dtq, _, cycq = run_timed(f, a, k2, opt.warmup, frames, only=False, thresh=1e-2, predictor=True)
            out['scf_predictor'] = dict(note='EXTRA, not the metric: SCF started from the linear extrapolation 2 U(n-1) - U(n-2) '
                                        'of the two previous steps instead of U(n-1)',
                                        ms_per_step=round(dtp / k2 * 1e3, 5), ns_per_day=round(0.0864 / (dtp / k2), 3),
                                        steps=k2, **cycp,
                                        thresh_1e_2=dict(ms_per_step=round(dtq / k2 * 1e3, 5), **cycq))
            dt0, _, cyc0 = run_timed(f, a, k2, opt.warmup, None, only=False)
            out['static_geometry'] = dict(note='UPPER BOUND, not the metric: identical positions every step, the first SCF '
                                          'check always passes (what round 1 reported as the headline)',
                                          ms_per_step=round(dt0 / k2 * 1e3, 5), ns_per_day=round(0.0864 / (dt0 / k2), 3),
                                          steps=k2, **cyc0)
        except Exception as e:
            out['scf_tight'] = {'error': repr(e)}
    if not opt.no_cpu and opt.workload == 'S1':
        # frame `last` on the GPU from the dipoles of frame last-1, then the oracle on the same frame from the same start
        p_next = frames.step_frame(last)
        E, G = step(f, a, U_last, p_next)
        ncyc_gpu = int(f.n_cycle)
        cb, ref = cpu_baseline(w, U_last.cpu().numpy() if hasattr(U_last, 'cpu') else U_last, p_next.cpu().numpy())
        out['cpu_baseline'] = cb
        Gh = G.cpu().numpy()
        out['force_rel_l2_vs_oracle'] = float(np.linalg.norm(Gh - ref['grad']) / np.linalg.norm(ref['grad']))
        out['energy_rel_vs_oracle'] = float(abs(E - ref['E']) / abs(ref['E']))
        out['scf_cycles_gpu_vs_oracle'] = [ncyc_gpu, int(ref['n_cycle'])]
    if not opt.no_extras:
        try:
            k3 = min(opt.steps, 20)
            out['md_all_terms'] = {w['name']: md_all_terms(w, f, a, frames, k3, opt.warmup)}
        except Exception as e:
            out['md_all_terms'] = {'error': repr(e)}
    if not opt.no_scale and opt.workload == 'S1':
        try:
            f = None
            frames = None
            torch.cuda.empty_cache()
            w3 = make_workload('S3')
            f3, a3 = make_force(w3)
            fr3 = ThermalFrames(w3, dev)
            dt3, rep3, cyc3 = run_timed(f3, a3, 5, 2, fr3)
            kb3, cn3, rs3 = kernel_breakdown(f3, a3, fr3, 7, 5)
            rb3 = time_list_rebuild(f3, w3)
            f3.set_pairs(a3['pairs'])
            sc = {'workload': w3['desc'], 'n_atoms': 3 * w3['n_mol'], 'n_pairs': int(f3.n_pairs),
                  'geometry': 'MOVING: ' + fr3.describe(),
                  'ms_per_step': round(dt3 / 5 * 1e3, 3), 'ns_per_day': round(0.0864 / (dt3 / 5), 3), 'dtype': 'f32',
                  'list_rebuild_ms': round(rb3, 3),
                  'ns_per_day_rebuild_every_10_steps': round(0.0864 / (dt3 / 5 + rb3 * 1e-4), 3),
                  'roofline': roofline_of(rep3, w3, f3.n_pairs, rs3),
                  'kernel_ms_per_step': kb3, 'kernel_launches_per_step': cn3,
                  'recip_kernels': recip_kernel_rooflines(kb3, cn3, w3, [f3.K1, f3.K2, f3.K3])}
            sc.update(cyc3)
            sc['slab_1rank_ms'] = slab_one_rank_ms(w3, fr3)
            if not opt.no_extras:
                dt0, _, cyc0 = run_timed(f3, a3, 5, 2, None, only=False)
                sc['static_geometry'] = dict(note='upper bound (identical positions every step)',
                                             ms_per_step=round(dt0 / 5 * 1e3, 3), ns_per_day=round(0.0864 / (dt0 / 5), 3))
                dt2, _, cyc2 = run_timed(f3, a3, 5, 2, fr3, only=False, thresh=1e-2)
                sc['scf_tight'] = dict(thresh=1e-2, ms_per_step=round(dt2 / 5 * 1e3, 3),
                                       ns_per_day=round(0.0864 / (dt2 / 5), 3), **cyc2)
                dtp, _, cycp = run_timed(f3, a3, 5, 3, fr3, only=False, predictor=True)
                dtq, _, cycq = run_timed(f3, a3, 5, 3, fr3, only=False, thresh=1e-2, predictor=True)
                sc['scf_predictor'] = dict(note='extra: SCF started from 2 U(n-1) - U(n-2)',
                                           ms_per_step=round(dtp / 5 * 1e3, 3), ns_per_day=round(0.0864 / (dtp / 5), 3), **cycp,
                                           thresh_1e_2=dict(ms_per_step=round(dtq / 5 * 1e3, 3), **cycq))
                sc['precision_check'] = f32_vs_f64_force_error(w3, f3, a3, fr3.frame(3))
                out.setdefault('md_all_terms', {})[w3['name']] = md_all_terms(w3, f3, a3, fr3, 10, 2)
            out['at_scale'] = sc
        except Exception as e:      # the headline line must still be printed
            out['at_scale'] = {'error': repr(e)}
        if not opt.no_extras:
            try:      # configs[2] size with the mesh the reference's own rule gives (K = 305: a Bluestein size for rocFFT)
                f3 = a3 = fr3 = None
                torch.cuda.empty_cache()
                w2 = make_workload('S2ref')
                f2, a2 = make_force(w2)
                fr2 = ThermalFrames(w2, dev)
                dtm, _, cycm = run_timed(f2, a2, 10, 3, fr2, only=False)
                dts, _, _ = run_timed(f2, a2, 10, 3, None, only=False)
                out['reference_rule_mesh'] = dict(
                    workload=w2['desc'], pme_grid=[f2.K1, f2.K2, f2.K3],
                    convolution='two-level (Good-Thomas) direct DFT, admp_amd/csrc/pfa_kernels.hip; the rocFFT (Bluestein) '
                                'leg of the same step: ADMP_PFA=0, see profiles/README.md',
                    ms_per_step=round(dtm / 10 * 1e3, 3), ns_per_day=round(0.0864 / (dtm / 10), 3),
                    static_geometry_ms_per_step=round(dts / 10 * 1e3, 3), **cycm)
            except Exception as e:
                out['reference_rule_mesh'] = {'error': repr(e)}
    print(json.dumps(out))
    return 0


if __name__ == '__main__':
    sys.exit(main())
