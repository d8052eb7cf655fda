"""Multi-GPU evaluation of the PME path: 1-D slab decomposition along x, one process per GPU.

The reference is single-device; this module is the multi-GPU layer of the MI355X build.  Every rank makes
the SAME call with the same full input arrays (the reference's calling convention), owns the mesh planes [X0, X1) and
works on its "home" atoms (lowest B-spline stencil plane inside the slab).  Since round 3 the whole evaluation -- the
same kernels, the same SCF forms (speculative / chained / plain, incremental field updates, charge-only pair forms) as
on one GPU -- runs inside the library (`admp_pme_energy_grad` on a handle configured with `admp_slab_configure` +
`admp_set_comm`); the library calls back into the communicator below where ranks exchange data:

  * decomposition : every rank derives owner / home / import / export lists of the evaluation BY ITSELF from the replicated
                    inputs and the (symmetric) neighbour table -- no communication, one host read of the list lengths;
  * real space    : pair kernels over the home rows only -- the i-grouped table evaluates each pair from the row atom's
                    side, so nothing is written to a remote atom;
  * spread        : home atoms -> local mesh (slab + 5 ghost planes), ghost planes sent to the next rank (`shift`);
  * 3-D FFT       : batched 2-D r2c on the owned planes, all-to-all transpose (`all_to_all_v`; RCCL over xGMI: every GPU
                    exchanges a block with each of the other N-1 peers, all 7 links busy), x lines forward * G * inverse in
                    one kernel on the y rows the rank owns, and back;
  * gather        : phi ghost planes fetched from the next rank, home atoms gathered;
  * SCF           : max|field| reduced on the device (`all_reduce` MAX of one word per check); after a Jacobi step only
                    the dipole CHANGES of the imported atoms travel (`all_to_all_v` of n_import x 3 reals);
  * gradient      : contributions a rank made to atoms it does not own (local-frame adjoint of molecules that straddle a
                    slab face) go back to the owners over the same lists; the four energies by one SUM all-reduce.

The library returns the rank's home rows of gradient / dipoles / dE/dQ_local.  With outputs='home' (the default since round
4) nothing proportional to the number of atoms is ever sent -- each rank keeps its home rows (`home_atoms`, the other rows
are zero), the form an MD driver that keeps the atoms distributed uses.  outputs='replicated' gives every caller the full
arrays, as the reference's single-device API does, at the price of one SUM all-reduce of (Na, 3) per output array.

Communicators.  `RcclComm` (round 4, the production path): the library issues the collectives ITSELF -- ncclAllReduce and
grouped ncclSend / ncclRecv on the handle's stream (admp_amd/csrc/rccl_comm.hip) -- Python only distributes the 128-byte
unique id once, at construction.  `TorchComm` / `ThreadComm` implement the callback interface of include/admp_hip.h
(`all_reduce`, `all_to_all_v`, `shift` on flat device tensors) on torch.distributed (gloo with host staging: functional
rehearsal of several ranks on one GPU; nccl: the round-3 path, kept as the test double of the native one) and on in-process
threads (tests).  `make_comm()` picks RcclComm for an nccl process group and TorchComm otherwise.  All count the bytes they
send per label (`bytes_sent`) and time the collectives with device events (`report`; the native path through the library's
own profiler labels `comm_*`).  `SlabDispPme` / `SlabPairInteraction` decompose dispersion PME and the pair
potentials over the same slabs.
"""
import collections
import ctypes
import threading

import numpy as np
import torch

from . import _lib, settings
from .disp_pme import ADMPDispPmeForce
from .pairwise import _HipPairKernel, _PairInteraction
from .pme import ADMPPmeForce


# ------------------------------------------------------------------------------------------ communicators
class _CommBase:
    def _init_stats(self):
        self.bytes_sent = collections.defaultdict(int)     # label -> bytes this rank sent to OTHER ranks
        self.calls = collections.defaultdict(int)
        self.profile = False
        self._events = []
        self._ms = collections.defaultdict(float)

    def _count(self, label, nbytes):
        self.bytes_sent[label] += int(nbytes)
        self.calls[label] += 1

    def _timed(self, label):
        comm = self

        class _T:
            def __enter__(self_):
                if comm.profile:
                    self_.a = torch.cuda.Event(enable_timing=True)
                    self_.b = torch.cuda.Event(enable_timing=True)
                    self_.a.record()

            def __exit__(self_, *exc):
                if comm.profile:
                    self_.b.record()
                    comm._events.append((label, self_.a, self_.b))
        return _T()

    def report(self, steps=1):
        """{label: ms per step} of the collectives since the last call (device-event time: includes waiting for peers)."""
        torch.cuda.synchronize()
        for label, a, b in self._events:
            self._ms[label] += a.elapsed_time(b)
        self._events = []
        out = {k: round(v / max(steps, 1), 4) for k, v in sorted(self._ms.items())}
        self._ms = collections.defaultdict(float)
        return out

    def reset_stats(self):
        self.bytes_sent.clear()
        self.calls.clear()
        self._events = []
        self._ms = collections.defaultdict(float)


class TorchComm(_CommBase):
    """torch.distributed backend.  nccl (= RCCL on ROCm) works on device tensors directly; with gloo the
    tensors are staged through host memory (functional testing on one GPU / CPU only)."""

    def __init__(self, group=None):
        import torch.distributed as dist
        self.dist = dist
        self.group = group
        self.rank = dist.get_rank(group)
        self.size = dist.get_world_size(group)
        self.native = dist.get_backend(group) == 'nccl'
        self._init_stats()

    def all_reduce(self, t, op='sum', label='all_reduce'):
        """in place on a device tensor"""
        rop = self.dist.ReduceOp.SUM if op == 'sum' else self.dist.ReduceOp.MAX
        # ring all-reduce: every rank sends 2 (N-1)/N of the buffer
        self._count(label, 2 * (self.size - 1) * t.numel() * t.element_size() // max(self.size, 1))
        with self._timed(label):
            if self.native:
                self.dist.all_reduce(t, op=rop, group=self.group)
            else:
                h = t.cpu()
                self.dist.all_reduce(h, op=rop, group=self.group)
                t.copy_(h)
        return t

    def all_to_all_v(self, recv, send, rsplits, ssplits, label='all_to_all'):
        """flat 1-D buffers; segment t of `send` (ssplits[t] elements) goes to rank t, segment s of `recv` comes from s"""
        self._count(label, (sum(ssplits) - ssplits[self.rank]) * send.element_size())
        with self._timed(label):
            if self.native:
                self.dist.all_to_all_single(recv, send, output_split_sizes=list(rsplits), input_split_sizes=list(ssplits),
                                            group=self.group)
                return
            so = np.concatenate([[0], np.cumsum(ssplits)]).astype(np.int64)
            ro = np.concatenate([[0], np.cumsum(rsplits)]).astype(np.int64)
            hs = send.cpu()
            hr = torch.empty(recv.shape, dtype=recv.dtype)
            ops = []
            for p in range(self.size):
                if p == self.rank:
                    hr[ro[p]:ro[p + 1]] = hs[so[p]:so[p + 1]]
                    continue
                if ssplits[p]:
                    ops.append(self.dist.P2POp(self.dist.isend, hs[so[p]:so[p + 1]].contiguous(), p, self.group))
                if rsplits[p]:
                    ops.append(self.dist.P2POp(self.dist.irecv, hr[ro[p]:ro[p + 1]], p, self.group))
            if ops:
                for w in self.dist.batch_isend_irecv(ops):
                    w.wait()
            recv.copy_(hr)

    def abort(self):
        """This rank cannot go on (an exception inside a collective's callback): close its side of the group so that peers
        blocked in a collective fail promptly instead of waiting for the group's timeout (gloo: their pending operations see
        the connection closed; nccl: the communicator is aborted where torch offers it)."""
        d = self.dist
        try:
            ab = getattr(d.distributed_c10d, '_abort_process_group', None)
            if self.native and ab is not None:
                ab(self.group)
            else:
                d.destroy_process_group(self.group)
        except Exception:      # noqa: BLE001 -- already on an error path
            pass

    def shift(self, send, recv, to_next, label='shift'):
        """ring shift: send to rank+1 (to_next) or rank-1, receive from the opposite neighbour."""
        if self.size == 1:
            recv.copy_(send)
            return
        self._count(label, send.numel() * send.element_size())
        dst = (self.rank + (1 if to_next else -1)) % self.size
        src = (self.rank - (1 if to_next else -1)) % self.size
        with self._timed(label):
            s = send.contiguous() if self.native else send.cpu().contiguous()
            r = recv if self.native else torch.empty(recv.shape, dtype=recv.dtype)
            ops = [self.dist.P2POp(self.dist.isend, s, dst, self.group), self.dist.P2POp(self.dist.irecv, r, src, self.group)]
            for w in self.dist.batch_isend_irecv(ops):
                w.wait()
            if not self.native:
                recv.copy_(r)


class RcclComm(_CommBase):
    """Native communicator: the collectives of a decomposed handle are RCCL calls made by libadmp_hip itself
    (include/admp_hip.h admp_rccl_*), nothing re-enters Python during a step.  The 128-byte unique id is created on rank 0
    and distributed over an existing torch.distributed group (any backend), or handed in by the caller (`unique_id` with
    explicit `rank` / `size`: MPI, a file, ...)."""
    native_rccl = True

    def __init__(self, group=None, device=None, unique_id=None, rank=None, size=None):
        self._L = _lib.load()
        self._c = ctypes.c_void_p()
        self.device = torch.device('cuda', torch.cuda.current_device()) if device is None else torch.device(device)
        if unique_id is None:
            import torch.distributed as dist
            self.rank, self.size = dist.get_rank(group), dist.get_world_size(group)
            buf = ctypes.create_string_buffer(_lib.RCCL_ID_BYTES)
            if self.rank == 0:
                self._check(self._L.admp_rccl_unique_id(buf), 'admp_rccl_unique_id')
            box = [bytes(buf.raw) if self.rank == 0 else None]
            dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
            unique_id = box[0]
        else:
            self.rank, self.size = int(rank), int(size)
        if len(unique_id) != _lib.RCCL_ID_BYTES:
            raise ValueError('unique_id must be %d bytes' % _lib.RCCL_ID_BYTES)
        self._check(self._L.admp_rccl_create(ctypes.byref(self._c), self.device.index, unique_id, self.rank, self.size),
                    'admp_rccl_create')
        self._init_stats()

    def _check(self, rc, what):
        if rc != 0:
            raise _lib.AdmpHipError('%s failed (code %d): %s' % (what, rc, self._L.admp_rccl_last_error().decode()))

    def version(self):
        v = ctypes.c_int(0)
        self._check(self._L.admp_rccl_version(ctypes.byref(v)), 'admp_rccl_version')
        return int(v.value)

    def all_reduce(self, t, op='sum', label='all_reduce'):
        """in place on a contiguous device tensor, on the caller's current stream (the wrapper's `outputs='replicated'`)"""
        dt = {torch.float32: _lib.T_F32, torch.float64: _lib.T_F64, torch.int32: _lib.T_I32}[t.dtype]
        st = torch.cuda.current_stream(self.device).cuda_stream
        self._check(self._L.admp_rccl_all_reduce(self._c, ctypes.c_void_p(t.data_ptr()), t.numel(), dt,
                                                 _lib.OP_MAX if op == 'max' else _lib.OP_SUM, ctypes.c_void_p(st)),
                    'admp_rccl_all_reduce')
        return t

    def _pull_stats(self, reset=False):
        b = (ctypes.c_int64 * _lib.RCCL_NTAGS)()
        c = (ctypes.c_int64 * _lib.RCCL_NTAGS)()
        self._check(self._L.admp_rccl_stats(self._c, b, c, 1 if reset else 0), 'admp_rccl_stats')
        self.bytes_sent = collections.defaultdict(int, {_lib.TAGS[t]: int(b[t]) for t in _lib.TAGS if c[t]})
        self.calls = collections.defaultdict(int, {_lib.TAGS[t]: int(c[t]) for t in _lib.TAGS if c[t]})

    def refresh_stats(self):
        """bytes_sent / calls per label since the last reset, read from the library"""
        self._pull_stats()
        return self.bytes_sent

    def reset_stats(self):
        if getattr(self, '_c', None) is not None and self._c.value:
            self._pull_stats(reset=True)
        self.bytes_sent = collections.defaultdict(int)
        self.calls = collections.defaultdict(int)

    def report(self, steps=1):
        """collective times of the native path are the library profiler's `comm_*` labels (profile_report of the force)"""
        return {}

    def abort(self):
        if self._c.value:
            self._L.admp_rccl_abort(self._c)

    def close(self):
        if getattr(self, '_c', None) is not None and self._c.value:
            self._L.admp_rccl_destroy(self._c)
            self._c = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def make_comm(group=None, device=None):
    """The communicator of this process for the slab calculators: RCCL issued by the library itself when the process group's
    backend is nccl (one process per GPU), the host-staged TorchComm otherwise (gloo: rehearsals of several ranks on one GPU,
    CPU tests).  ADMP_COMM=torch forces the callback path on nccl as well (A/B against the native one)."""
    import os
    import torch.distributed as dist
    if dist.get_backend(group) == 'nccl' and os.environ.get('ADMP_COMM', 'rccl') != 'torch':
        return RcclComm(group, device)
    return TorchComm(group)


class ThreadComm(_CommBase):
    """In-process communicator: N python threads (one SlabPme each, all on the same GPU) exchange tensors
    through a shared mailbox guarded by a barrier.  Test infrastructure for the decomposition logic."""

    class World:
        def __init__(self, size, serialize=False):
            """serialize: only one rank at a time runs its compute segments (a token handed over at the collectives), so
            that kernels of different ranks never interleave on the one GPU -- per-kernel event timings then mean what
            they would on a GPU of the rank's own (tools/slab_rehearsal.py).  Every rank thread brackets its work with
            comm.begin() / comm.end()."""
            self.size = size
            self.barrier = threading.Barrier(size)
            self.box = [None] * size
            self.token = threading.Lock() if serialize else None

    def __init__(self, world, rank):
        self.w, self.rank, self.size = world, rank, world.size
        self._init_stats()

    def begin(self):
        if self.w.token is not None:
            self.w.token.acquire()

    def end(self):
        if self.w.token is not None:
            torch.cuda.current_stream().synchronize()
            self.w.token.release()

    def _exchange(self, item):
        torch.cuda.current_stream().synchronize()
        self.w.box[self.rank] = item
        if self.w.token is not None:
            self.w.token.release()
        self.w.barrier.wait()
        got = list(self.w.box)
        self.w.barrier.wait()
        if self.w.token is not None:
            self.w.token.acquire()
        return got

    def all_reduce(self, t, op='sum', label='all_reduce'):
        self._count(label, 2 * (self.size - 1) * t.numel() * t.element_size() // max(self.size, 1))
        parts = torch.stack(self._exchange(t.clone()))
        t.copy_(parts.sum(dim=0) if op == 'sum' else parts.max(dim=0).values)
        return t

    def all_to_all_v(self, recv, send, rsplits, ssplits, label='all_to_all'):
        self._count(label, (sum(ssplits) - ssplits[self.rank]) * send.element_size())
        got = self._exchange((send.clone(), list(ssplits)))
        off = 0
        for s in range(self.size):
            buf, sp = got[s]
            o = sum(sp[:self.rank])
            n = sp[self.rank]
            assert n == rsplits[s]
            recv[off:off + n] = buf[o:o + n]
            off += n

    def shift(self, send, recv, to_next, label='shift'):
        if self.size > 1:
            self._count(label, send.numel() * send.element_size())
        got = self._exchange(send.clone())
        src = (self.rank - (1 if to_next else -1)) % self.size
        recv.copy_(got[src])


# ------------------------------------------------------------------------------------------ slab driver
def slab_bounds(K, size):
    """[(lo, hi)] of the `size` slabs of an axis with K points (same rule as the library: floor(s*K/size))."""
    return [((s * K) // size, ((s + 1) * K) // size) for s in range(size)]


_TORCH_OF = {_lib.T_I32: (torch.int32, '<i4'), _lib.T_F32: (torch.float32, '<f4'), _lib.T_F64: (torch.float64, '<f8')}


class _DevArray:
    """raw device memory of the library as an object torch can wrap (CUDA array interface; works on ROCm builds)"""

    def __init__(self, ptr, count, typestr):
        self.__cuda_array_interface__ = {'shape': (int(count),), 'typestr': typestr, 'data': (int(ptr), False), 'version': 2}


class CommBinding:
    """The admp_comm callbacks (include/admp_hip.h) of one handle, bound to a communicator object.  The library hands over
    raw device pointers of its own buffers; they are wrapped as flat torch tensors (cached per pointer / length / type: the
    buffers are the same step after step) and passed to the communicator, which works on the caller's current stream --
    the stream the library runs on."""

    def __init__(self, comm, device):
        self.comm, self.device = comm, device
        self.error = None
        self._views = {}
        self._cbs = (_lib.ALL_REDUCE_FN(self._all_reduce), _lib.ALL_TO_ALL_V_FN(self._all_to_all_v), _lib.SHIFT_FN(self._shift))
        self.struct = _lib.AdmpComm(None, *self._cbs)

    def _view(self, ptr, count, dtype):
        key = (ptr, count, dtype)
        t = self._views.get(key)
        if t is None:
            td, ts = _TORCH_OF[dtype]
            if count == 0 or not ptr:
                t = torch.empty(0, dtype=td, device=self.device)
            else:
                t = torch.as_tensor(_DevArray(ptr, count, ts), device=self.device)
            if len(self._views) > 256:
                self._views.clear()
            self._views[key] = t
        return t

    def _guard(self, fn):
        try:
            fn()
            return 0
        except BaseException as e:      # never let an exception cross the C frame
            self.error = e
            return 1

    def _all_reduce(self, ctx, buf, count, dtype, op, tag):
        return self._guard(lambda: self.comm.all_reduce(self._view(buf, count, dtype), 'max' if op == _lib.OP_MAX else 'sum',
                                                        label=_lib.TAGS.get(tag, 'all_reduce')))

    def _all_to_all_v(self, ctx, send, scounts, recv, rcounts, dtype, tag):
        def run():
            n = self.comm.size
            sc, rc = [int(scounts[t]) for t in range(n)], [int(rcounts[t]) for t in range(n)]
            self.comm.all_to_all_v(self._view(recv, sum(rc), dtype), self._view(send, sum(sc), dtype), rc, sc,
                                   label=_lib.TAGS.get(tag, 'all_to_all'))
        return self._guard(run)

    def _shift(self, ctx, send, recv, count, dtype, to_next, tag):
        return self._guard(lambda: self.comm.shift(self._view(send, count, dtype), self._view(recv, count, dtype), bool(to_next),
                                                   label=_lib.TAGS.get(tag, 'shift')))


class _NoBinding:
    error = None


class _SlabMixin:
    """What the decomposed calculators share: binding the communicator to the handle, the home list of an evaluation and
    the assembly of per-atom outputs."""

    def _bind_comm(self, comm, outputs):
        if outputs not in ('replicated', 'home'):
            raise ValueError("outputs must be 'replicated' or 'home'")
        self.comm = comm
        self.outputs = outputs
        self.home_atoms = None
        self.n_home = 0
        self.n_import = 0
        if getattr(comm, 'native_rccl', False):      # the library talks to RCCL itself: no callbacks
            self._binding = _NoBinding()
            _lib.check(self._h, self._L.admp_set_comm_rccl(self._h, comm._c), 'admp_set_comm_rccl')
            return
        self._binding = CommBinding(comm, self._device)
        _lib.check(self._h, self._L.admp_slab_configure(self._h, comm.rank, comm.size), 'admp_slab_configure')
        _lib.check(self._h, self._L.admp_set_comm(self._h, ctypes.byref(self._binding.struct) if comm.size > 1 else None),
                   'admp_set_comm')

    def _checked(self, fn):
        """run a library call; an exception raised inside a communicator callback is re-raised as itself"""
        self._binding.error = None
        try:
            return fn()
        except _lib.AdmpHipError:
            # A rank that fails in the middle of a decomposed call leaves its peers inside a collective: release them (they
            # fail too, promptly) rather than let them wait -- for the group's timeout with torch.distributed, for ever with
            # raw RCCL.
            if self.comm.size > 1 and hasattr(self.comm, 'abort'):
                self.comm.abort()
            if self._binding.error is not None:
                raise self._binding.error
            raise

    def _fetch_home(self):
        if self.comm.size == 1:
            self.home_atoms = torch.arange(self.n_atoms, device=self._device)
            self.n_home, self.n_import = self.n_atoms, 0
            return
        home = torch.empty(self.n_atoms, dtype=torch.int32, device=self._device)
        nh, ni = ctypes.c_int(0), ctypes.c_int(0)
        _lib.check(self._h, self._L.admp_slab_home(self._h, self._ptr(home), ctypes.byref(nh), ctypes.byref(ni)), 'admp_slab_home')
        self.n_home, self.n_import = int(nh.value), int(ni.value)
        self.home_atoms = home[:self.n_home].to(torch.int64)

    def _assemble(self, x):
        """per-atom output of a decomposed call: the rank's home rows are valid.  'replicated': one sum over the ranks gives
        every caller the full array (the reference's API); 'home': the other rows are zeroed, nothing is sent."""
        if x is None or self.comm.size == 1:
            return x
        full = torch.zeros_like(x)
        full.index_copy_(0, self.home_atoms, x.index_select(0, self.home_atoms))
        if self.outputs == 'replicated':
            self.comm.all_reduce(full.view(-1), op='sum', label='replicate_outputs')
        return full


    def _assemble_any(self, x):
        """_assemble for a per-atom output in whatever container the base class returned it (torch tensor or numpy array)"""
        if x is None or self.comm.size == 1:
            return x
        if isinstance(x, torch.Tensor):
            return self._assemble(x.to(self._device))
        return self._assemble(torch.as_tensor(np.ascontiguousarray(x), device=self._device)).cpu().numpy()


class SlabPme(_SlabMixin, ADMPPmeForce):
    """ADMPPmeForce whose evaluation is spread over the ranks of `comm` (SPMD: every rank makes the same
    get_energy / get_forces call with the same full input arrays).  outputs='home' (default): only the rows of `home_atoms`
    are valid on a rank (no O(Na) communication); outputs='replicated': every rank receives the full gradient / dipoles."""

    def __init__(self, comm, box, axis_type, axis_indices, covalent_map, rc, ethresh, lmax, lpol=False, device=None,
                 outputs='home'):
        self._pending = (comm, outputs)
        super().__init__(box, axis_type, axis_indices, covalent_map, rc, ethresh, lmax, lpol, device)

    def refresh_calculators(self):
        if getattr(self, '_binding', None) is None:
            self._bind_comm(*self._pending)
        super().refresh_calculators()

    def _evaluate_on_stream(self, *args):
        out = self._checked(lambda: ADMPPmeForce._evaluate_on_stream(self, *args))
        self._fetch_home()
        for k in ('grad', 'U', 'dQ'):
            if k in out:
                out[k] = self._assemble(out[k])
        return out


    def get_energy_and_box_gradient(self, *a, **k):
        """(E, dE/dbox) on the slab ranks (round 4): pair / gather / frame sums over the rank's home rows, k-tensor sums over
        its rows of the transposed spectrum, the 24 sums added over the ranks -- every rank returns the full 3 x 3 gradient."""
        return self._checked(lambda: ADMPPmeForce.get_energy_and_box_gradient(self, *a, **k))

    def _at_U(self, *a, **k):
        """the bare calculators at caller-supplied dipoles (energy_fn / grad_U_fn / grad_pos_fn) on the slab ranks: the energy
        is global, the two gradients are per-atom outputs (home rows, assembled like the others)"""
        E, gpos, gU = self._checked(lambda: ADMPPmeForce._at_U(self, *a, **k))
        self._fetch_home()
        return E, self._assemble_any(gpos), self._assemble_any(gU)

    # parameter gradients on the slab ranks (round 4): class sums (dE/dmScales, dE/dpScales) are added over the ranks inside the
    # library -- every rank returns the full vector; per-atom ones (dE/dpol, dE/dtholes) are outputs like the gradient
    def get_mscale_gradient(self, *a, **k):
        return self._checked(lambda: ADMPPmeForce.get_mscale_gradient(self, *a, **k))

    def get_pscale_gradient(self, *a, **k):
        return self._checked(lambda: ADMPPmeForce.get_pscale_gradient(self, *a, **k))

    def get_pol_thole_gradients(self, *a, **k):
        dpol, dth = self._checked(lambda: ADMPPmeForce.get_pol_thole_gradients(self, *a, **k))
        self._fetch_home()
        return self._assemble_any(dpol), self._assemble_any(dth)


class SlabDispPme(_SlabMixin, ADMPDispPmeForce):
    """ADMPDispPmeForce decomposed over the same x-slabs (real-space pairs over the home rows, the C6 / C8 / C10 meshes
    through the distributed transform, gathers at the home atoms)."""

    def __init__(self, comm, box, covalent_map, rc, ethresh, pmax, device=None, outputs='home'):
        self._pending = (comm, outputs)
        super().__init__(box, covalent_map, rc, ethresh, pmax, device)

    def refresh_calculators(self):
        if getattr(self, '_binding', None) is None:
            self._bind_comm(*self._pending)
        super().refresh_calculators()

    def _evaluate(self, positions, box, pairs, c_list, mScales, want_grad=True):
        E, grad = self._checked(lambda: ADMPDispPmeForce._evaluate(self, positions, box, pairs, c_list, mScales, want_grad))
        self._fetch_home()
        return E, self._assemble(grad)


    def get_energy_and_box_gradient(self, *a, **k):
        """(E, dE/dbox) of dispersion PME on the slab ranks (round 4); every rank returns the full gradient"""
        return self._checked(lambda: ADMPDispPmeForce.get_energy_and_box_gradient(self, *a, **k))

    def get_mscale_gradient(self, *a, **k):
        return self._checked(lambda: ADMPDispPmeForce.get_mscale_gradient(self, *a, **k))

    def get_param_gradient(self, *a, **k):
        g = self._checked(lambda: ADMPDispPmeForce.get_param_gradient(self, *a, **k))
        self._fetch_home()
        return self._assemble_any(g)


class SlabPairInteraction(_SlabMixin, _PairInteraction):
    """generate_pairwise_interaction(kernel, covalent_map, static_args) for the kernels of libadmp_hip (Tang-Toennies),
    decomposed over x-slabs: every rank evaluates the rows of its home atoms; one SUM all-reduce of the energy."""

    def __init__(self, comm, kernel, covalent_map, static_args=None, outputs='home'):
        if not isinstance(kernel, _HipPairKernel):
            raise TypeError('the slab decomposition takes the named pair kernels of admp_amd.pairwise')
        super().__init__(kernel, covalent_map, static_args)
        self._bind_comm(comm, outputs)

    def _evaluate(self, positions, box, pairs, mScales, atomic_params, want_grad):
        E, grad = self._checked(lambda: _PairInteraction._evaluate(self, positions, box, pairs, mScales, atomic_params, True))
        self._fetch_home()
        return E, (self._assemble(grad) if want_grad else None)

    def get_energy_and_box_gradient(self, *a, **k):
        """(E, dE/dbox) on the slab ranks: the pair sums over the rank's home rows, added over the ranks"""
        return self._checked(lambda: _PairInteraction.get_energy_and_box_gradient(self, *a, **k))

    def get_mscale_gradient(self, *a, **k):
        return self._checked(lambda: _PairInteraction.get_mscale_gradient(self, *a, **k))

    def get_param_gradient(self, *a, **k):
        gs = self._checked(lambda: _PairInteraction.get_param_gradient(self, *a, **k))
        self._fetch_home()
        return tuple(self._assemble_any(g) for g in gs)
