"""Multi-GPU evaluation of the PME path: 1-D slab decomposition along x, one process per GPU.

The reference is single-device; this module is the multi-GPU layer of the MI355X build:

    every rank holds all atoms' inputs, owns the mesh planes [X0, X1) and works on its "home" atoms
    (lowest B-spline stencil plane inside the slab).  Per evaluation:
      * real space   : pair kernel over the home rows only -- the i-grouped neighbour table evaluates each
                       pair from the row atom's side, so no force is ever written to a remote atom;
      * spread       : home atoms -> local mesh (slab + 5 ghost planes), ghost planes sent to the next rank;
      * 3-D FFT      : batched 2-D r2c on the owned planes, all-to-all transpose (RCCL over xGMI: every
                       GPU exchanges an equal block with each of the other N-1 peers), batched 1-D c2c
                       along x, k-space multiply on the y-slab, and back;
      * gather       : phi ghost planes fetched from the next rank, home atoms gathered;
      * SCF          : max|field| by a scalar MAX all-reduce, new dipoles by a SUM all-reduce;
      * result       : gradient and the four energies by SUM all-reduce.

`SlabPme` drives the staged C ABI (admp_stage_*, include/admp_hip.h) and is written against a small
communicator interface so that the same code runs over torch.distributed (`TorchComm`: nccl = RCCL, or
gloo with host staging) and over an in-process thread communicator (`ThreadComm`) used by the tests.
"""
import ctypes
import threading

import numpy as np
import torch

from . import _lib, settings
from .pme import ADMPPmeForce


# ------------------------------------------------------------------------------------------ communicators
class TorchComm:
    """torch.distributed backend.  nccl (= RCCL on ROCm) works on device tensors directly; with gloo the
    tensors are staged through host memory (functional testing on one GPU / CPU only)."""

    def __init__(self, group=None):
        import torch.distributed as dist
        self.dist = dist
        self.group = group
        self.rank = dist.get_rank(group)
        self.size = dist.get_world_size(group)
        self.native = dist.get_backend(group) == 'nccl'

    def _host(self, t):
        return t if self.native else t.cpu()

    def all_reduce_sum(self, t):
        if self.native:
            self.dist.all_reduce(t, group=self.group)
        else:
            h = t.cpu()
            self.dist.all_reduce(h, group=self.group)
            t.copy_(h)
        return t

    def all_reduce_max(self, x):
        t = torch.tensor([x], dtype=torch.float64, device='cuda' if self.native else 'cpu')
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX, group=self.group)
        return float(t.item())

    def all_to_all(self, recv, send):
        """recv[s] <- the block rank s addressed to this rank (send[t] goes to rank t)."""
        if self.native:
            self.dist.all_to_all(recv, send, group=self.group)
            return
        ops, hrecv = [], [torch.empty(r.shape, dtype=r.dtype) for r in recv]
        for s in range(self.size):
            if s == self.rank:
                hrecv[s].copy_(send[s])
                continue
            ops.append(self.dist.P2POp(self.dist.isend, send[s].cpu(), s, self.group))
            ops.append(self.dist.P2POp(self.dist.irecv, hrecv[s], s, self.group))
        if ops:
            for w in self.dist.batch_isend_irecv(ops):
                w.wait()
        for r, h in zip(recv, hrecv):
            r.copy_(h)

    def shift(self, send, recv, to_next):
        """ring shift: send to rank+1 (to_next) or rank-1, receive from the opposite neighbour."""
        if self.size == 1:
            recv.copy_(send)
            return
        dst = (self.rank + (1 if to_next else -1)) % self.size
        src = (self.rank - (1 if to_next else -1)) % self.size
        s = send.contiguous() if self.native else send.cpu().contiguous()
        r = recv if self.native else torch.empty(recv.shape, dtype=recv.dtype)
        ops = [self.dist.P2POp(self.dist.isend, s, dst, self.group), self.dist.P2POp(self.dist.irecv, r, src, self.group)]
        for w in self.dist.batch_isend_irecv(ops):
            w.wait()
        if not self.native:
            recv.copy_(r)


class ThreadComm:
    """In-process communicator: N python threads (one SlabPme each, all on the same GPU) exchange tensors
    through a shared mailbox guarded by a barrier.  Test infrastructure for the decomposition logic."""

    class World:
        def __init__(self, size):
            self.size = size
            self.barrier = threading.Barrier(size)
            self.box = [None] * size

    def __init__(self, world, rank):
        self.w, self.rank, self.size = world, rank, world.size

    def _exchange(self, item):
        torch.cuda.current_stream().synchronize()
        self.w.box[self.rank] = item
        self.w.barrier.wait()
        got = list(self.w.box)
        self.w.barrier.wait()
        return got

    def all_reduce_sum(self, t):
        parts = self._exchange(t.clone())
        t.copy_(torch.stack(parts).sum(dim=0))
        return t

    def all_reduce_max(self, x):
        return max(self._exchange(float(x)))

    def all_to_all(self, recv, send):
        got = self._exchange([s.clone() for s in send])
        for s in range(self.size):
            recv[s].copy_(got[s][self.rank])

    def shift(self, send, recv, to_next):
        got = self._exchange(send.clone())
        src = (self.rank - (1 if to_next else -1)) % self.size
        recv.copy_(got[src])


# ------------------------------------------------------------------------------------------ slab driver
def slab_bounds(K, size):
    """[(lo, hi)] of the `size` slabs of an axis with K points (same rule as the library: floor(s*K/size))."""
    return [((s * K) // size, ((s + 1) * K) // size) for s in range(size)]


class SlabPme(ADMPPmeForce):
    """ADMPPmeForce whose evaluation is spread over the ranks of `comm` (SPMD: every rank makes the same
    get_energy / get_forces call with the same full arrays and receives the same full results)."""

    GHOST = 5

    def __init__(self, comm, box, axis_type, axis_indices, covalent_map, rc, ethresh, lmax, lpol=False, device=None):
        self.comm = comm
        super().__init__(box, axis_type, axis_indices, covalent_map, rc, ethresh, lmax, lpol, device)

    def refresh_calculators(self):
        super().refresh_calculators()
        _lib.check(self._h, self._L.admp_slab_configure(self._h, self.comm.rank, self.comm.size), 'admp_slab_configure')
        self._bufs = None

    # -- helpers -------------------------------------------------------------------------------------------
    def _stage(self, what, a=None, b=None, iarg=0, dout=None):
        _lib.check(self._h, self._L.admp_stage(self._h, what, self._ptr(a), self._ptr(b), iarg, dout), 'admp_stage(%d)' % what)

    def _buffers(self):
        info = (ctypes.c_int64 * 11)()
        _lib.check(self._h, self._L.admp_slab_info(self._h, info), 'admp_slab_info')
        X0, X1, Y0, Y1, nloc, ghost, K0, K1, K2h = (int(v) for v in info[:9])
        key = (X0, X1, Y0, Y1, K0, K1, K2h, self._dtype)
        if self._bufs is None or self._bufs['key'] != key:
            K2 = int(self.K3)
            dev, dt = self._device, self._dtype
            nx, ny = X1 - X0, Y1 - Y0
            self._bufs = dict(key=key, nx=nx, ny=ny, K0=K0, K1=K1, K2h=K2h,
                              mesh=torch.empty((nloc, K1, K2), dtype=dt, device=dev),
                              spec=torch.empty((nx, K1, K2h, 2), dtype=dt, device=dev),
                              tbuf=torch.empty((K0, ny, K2h, 2), dtype=dt, device=dev),
                              ghost=torch.empty((ghost, K1, K2), dtype=dt, device=dev),
                              xs=slab_bounds(K0, self.comm.size), ys=slab_bounds(K1, self.comm.size))
        return self._bufs

    def _recip(self, scf):
        """spread -> distributed r2c -> G multiply (+ energy) -> distributed c2r; mesh then holds phi incl. ghosts."""
        B, c = self._buffers(), self.comm
        mesh, spec, tbuf, nx = B['mesh'], B['spec'], B['tbuf'], B['nx']
        self._stage(_lib.ST_SPREAD, mesh)
        if c.size > 1:
            # stencil overhang: my ghost planes are the next rank's first planes
            c.shift(mesh[nx:nx + self.GHOST], B['ghost'], to_next=True)
            mesh[:self.GHOST] += B['ghost']
            self._stage(_lib.ST_FFT_YZ, mesh, spec, 0)
            send = [spec[:, y0:y1].contiguous() for (y0, y1) in B['ys']]
            recv = [torch.empty((x1 - x0, B['ny'], B['K2h'], 2), dtype=spec.dtype, device=spec.device) for (x0, x1) in B['xs']]
            c.all_to_all(recv, send)
            torch.cat(recv, dim=0, out=tbuf)
            self._stage(_lib.ST_FFT_X, tbuf, None, 0)
            self._stage(_lib.ST_KSPACE, tbuf, None, 1 if scf else 0)
            self._stage(_lib.ST_FFT_X, tbuf, None, 1)
            send = [tbuf[x0:x1].contiguous() for (x0, x1) in B['xs']]
            recv = [torch.empty((nx, y1 - y0, B['K2h'], 2), dtype=spec.dtype, device=spec.device) for (y0, y1) in B['ys']]
            c.all_to_all(recv, send)
            torch.cat(recv, dim=1, out=spec)
            self._stage(_lib.ST_FFT_YZ, mesh, spec, 1)
            # phi halo: my ghost planes hold the next rank's first planes
            c.shift(mesh[:self.GHOST], B['ghost'], to_next=False)
            mesh[nx:nx + self.GHOST] = B['ghost']
        else:
            self._stage(_lib.ST_FFT_YZ, mesh, spec, 0)
            self._stage(_lib.ST_KSPACE, spec, None, 1 if scf else 0)
            self._stage(_lib.ST_FFT_YZ, mesh, spec, 1)

    # -- the evaluation --------------------------------------------------------------------------------------
    def _evaluate_on_stream(self, positions, box, pairs, Q_local, mScales, pol, tholes, pScales, dScales, U_init,
                            want_grad, want_dQ, maxiter, thresh):
        if want_dQ:
            raise NotImplementedError('dE/dQ_local is not assembled across ranks yet')
        L, h, na, c = self._L, self._h, self.n_atoms, self.comm
        self.set_pairs(pairs)
        pos = self._real(positions, (na, 3))
        Q = self._pad_Q(Q_local)
        boxh = self._host64(box, 9)
        mS = self._host64(mScales)
        ns = len(mS)
        pol_t = th_t = U = None
        pS = None
        if self.lpol:
            pol_t = self._real(pol, (na,))
            th_t = self._real(tholes, (na,))
            pS = self._host64(pScales, ns)
            U = (torch.zeros((na, 3), dtype=self._dtype, device=self._device) if U_init is None
                 else self._real(U_init, (na, 3)).clone())
        maxiter = settings.MAX_N_POL if maxiter is None else int(maxiter)
        thresh = settings.POL_CONV if thresh is None else float(thresh)
        nhome = ctypes.c_int(0)
        _lib.check(h, L.admp_stage_begin(h, self._ptr(pos), _lib.darr(boxh), self._ptr(Q), self._ptr(pol_t), self._ptr(th_t),
                                         ns, _lib.darr(mS), None if pS is None else _lib.darr(pS), self._ptr(U),
                                         ctypes.byref(nhome)), 'admp_stage_begin')
        self.n_home = int(nhome.value)
        B = self._buffers()
        mesh = B['mesh']
        phi_valid, cyc, flag, done = False, 0, True, False
        grad = torch.empty((na, 3), dtype=self._dtype, device=self._device)

        def jacobi(U):
            if c.size > 1:
                Unew = torch.zeros_like(U)
                self._stage(_lib.ST_JACOBI, Unew)
                c.all_reduce_sum(Unew)
                self._stage(_lib.ST_SET_U, Unew)
                return Unew
            self._stage(_lib.ST_JACOBI, U)
            return U

        if self.lpol:
            fm = (ctypes.c_double * 1)()
            start = 0
            if getattr(self, '_warm_regime', False):
                # steady-state regime (previous call passed its first check): first cycle with the full kernels, which
                # also deliver dE/dU -- if the check passes again the step is finished (engine.hip, `warm_regime`)
                self._stage(_lib.ST_PAIR_FULL, grad, None, 1)
                self._recip(scf=True)
                self._stage(_lib.ST_GATHER, mesh, grad, 1)
                self._stage(_lib.ST_FIELD_FINISH, dout=fm)
                if c.all_reduce_max(fm[0]) < thresh:
                    phi_valid = done = True
                else:
                    U = jacobi(U)
                    start = 1
            i = 0 if done else start
            for i in ([] if done else range(start, maxiter)):       # admp/pme.py:132-138
                self._stage(_lib.ST_PAIR_FIELD)
                self._recip(scf=True)
                self._stage(_lib.ST_GATHER_FIELD, mesh)
                self._stage(_lib.ST_FIELD_FINISH, dout=fm)
                if c.all_reduce_max(fm[0]) < thresh:
                    phi_valid = True
                    break
                U = jacobi(U)
            cyc = min(i, maxiter - 1)
            flag = (cyc != maxiter - 1)                    # admp/pme.py:139-143
            self._warm_regime = (cyc == 0)
        if not done:
            self._stage(_lib.ST_PAIR_FULL, grad)
            if not phi_valid:
                self._recip(scf=False)
            self._stage(_lib.ST_GATHER, mesh, grad)
        E = (ctypes.c_double * 4)()
        self._stage(_lib.ST_FINISH, grad if want_grad else None, None, 1 if phi_valid else 0, E)
        Et = torch.tensor(list(E), dtype=torch.float64, device=self._device)
        if c.size > 1:
            c.all_reduce_sum(Et)
            if want_grad:
                c.all_reduce_sum(grad)
        Eh = Et.cpu().tolist()
        torch.cuda.current_stream(self._device).synchronize()
        self.energy_parts = tuple(Eh)
        out = {'E': np.float64(sum(Eh))}
        if self.lpol:
            out['U'] = U
            out['flag'] = bool(flag)
            out['i'] = int(cyc)
        if want_grad:
            out['grad'] = grad
        return out
