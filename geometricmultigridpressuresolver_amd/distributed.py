"""Z-slab multi-GPU front end: one process per GPU, torch.distributed for rendezvous.

The orchestration (which operator needs which ghost plane, the collapse of the coarse tail to rank 0)
lives in the C++ library (csrc/mgps_solver.hip); this module only builds the transport it runs on:

* RcclComm        -- production: the library's own RCCL communicator (ncclSend/ncclRecv over xGMI on
                     the solver's stream); torch.distributed is used once, to ship the 128-byte
                     ncclUniqueId from rank 0 to the others.
* TorchDistComm   -- the same vtable filled with Python callbacks that stage through the host and use
                     torch.distributed point-to-point / collectives of ANY backend (gloo in the
                     test-suite, where two ranks may share one GPU, which RCCL does not allow).

The reference is single-process shared memory; nothing here has a counterpart there.
"""
import ctypes as C

import numpy as np
import torch
import torch.distributed as dist

from ._lib import check, lib
from .solver import GeometricMultigridPoissonSolver, _np_f32, _np_u8, _p, default_options

_EXCH = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t,
                    C.c_void_p, C.c_size_t, C.c_void_p)
_ALLR = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double), C.c_int, C.c_int)
_GATH = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p)
_DEST = C.CFUNCTYPE(None, C.c_void_p)
_GATHV = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t), C.c_int, C.c_void_p)
_SCATV = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t), C.c_void_p, C.c_size_t, C.c_int, C.c_void_p)
_ALLRD = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p)


class Xfer2(C.Structure):
    """mgps_xfer2: one message of mgps_comm.exchange2 in two segments"""

    _fields_ = [("ptr", C.c_void_p * 2), ("bytes", C.c_size_t * 2)]


_EXCH2 = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(Xfer2), C.POINTER(Xfer2), C.POINTER(Xfer2), C.POINTER(Xfer2), C.c_void_p)


class CommStruct(C.Structure):
    """mgps_comm (include/mgps.h)."""

    _fields_ = [
        ("struct_size", C.c_int),
        ("rank", C.c_int),
        ("size", C.c_int),
        ("user", C.c_void_p),
        ("exchange", _EXCH),
        ("allreduce", _ALLR),
        ("gather", _GATH),
        ("scatter", _GATH),
        ("destroy", _DEST),
        ("gatherv", _GATHV),
        ("scatterv", _SCATV),
        ("allreduce_device", _ALLRD),
        ("exchange2", _EXCH2),
    ]


class RcclComm:
    """The library's RCCL transport.  Needs an initialised torch.distributed process group (any
    backend) to broadcast the unique id."""

    def __init__(self, device=None, group=None):
        self.rank, self.size = dist.get_rank(group), dist.get_world_size(group)
        dev = torch.cuda.current_device() if device is None else int(device)
        ident = (C.c_ubyte * 128)()
        if self.rank == 0:
            check(lib().mgps_rccl_unique_id(ident))
        box = [bytes(ident)]
        dist.broadcast_object_list(box, src=0, group=group)
        ident = (C.c_ubyte * 128).from_buffer_copy(box[0])
        self.struct = CommStruct()
        check(lib().mgps_comm_create_rccl(C.byref(self.struct), self.rank, self.size, ident, dev))

    def selftest(self, floats=1 << 20):
        """send + receive to this rank itself through librccl (mgps_comm_rccl_selftest)"""
        lib().mgps_comm_rccl_selftest.argtypes = [C.c_void_p, C.c_size_t]
        check(lib().mgps_comm_rccl_selftest(C.byref(self.struct), floats))

    def selfbench(self, floats, reps=200):
        """microseconds per self send + receive group (mgps_comm_rccl_selfbench)"""
        us = C.c_double()
        lib().mgps_comm_rccl_selfbench.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.POINTER(C.c_double)]
        check(lib().mgps_comm_rccl_selfbench(C.byref(self.struct), floats, reps, C.byref(us)))
        return us.value

    def close(self):
        if self.struct is not None:
            lib().mgps_comm_destroy(C.byref(self.struct))
            self.struct = None


class TorchDistComm:
    """mgps_comm over torch.distributed with host staging (works with gloo; ranks may share a GPU)."""

    def __init__(self, group=None):
        self.group = group
        self.rank, self.size = dist.get_rank(group), dist.get_world_size(group)
        self._hip = C.CDLL("libamdhip64.so")  # the runtime torch already loaded
        self._hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        self._hip.hipStreamSynchronize.argtypes = [C.c_void_p]
        self.exchanges = 0
        self.bytes_sent = 0
        self._cb = (_EXCH(self._exchange), _ALLR(self._allreduce), _GATH(self._gather), _GATH(self._scatter))
        self._cbv = (_GATHV(self._gatherv), _SCATV(self._scatterv), _ALLRD(self._allreduce_device), _EXCH2(self._exchange2))
        self.struct = CommStruct(C.sizeof(CommStruct), self.rank, self.size, None, *self._cb, _DEST(), *self._cbv)
        self.device_allreduces = 0
        self.segmented_exchanges = 0
        self.drop_every = 0  # test hook: > 0 = what every N-th exchange receives is thrown away

    # -- staging helpers ---------------------------------------------------------------------------
    def _d2h(self, ptr, nbytes, stream):
        self._hip.hipStreamSynchronize(stream)
        t = torch.empty(nbytes, dtype=torch.uint8)
        assert self._hip.hipMemcpy(t.data_ptr(), ptr, nbytes, 2) == 0
        return t

    def _h2d(self, ptr, t):
        assert self._hip.hipMemcpy(ptr, t.data_ptr(), t.numel(), 1) == 0

    def _global(self, r):
        return r if self.group is None else dist.get_global_rank(self.group, r)

    # -- vtable ------------------------------------------------------------------------------------
    def _exchange(self, user, send_lo, send_lo_n, recv_lo, recv_lo_n, send_hi, send_hi_n, recv_hi, recv_hi_n, stream):
        try:
            self.exchanges += 1
            self.bytes_sent += (send_lo_n if send_lo else 0) + (send_hi_n if send_hi else 0)
            ops, recvs = [], []
            for send, sn, recv, rn, peer in ((send_lo, send_lo_n, recv_lo, recv_lo_n, self.rank - 1),
                                             (send_hi, send_hi_n, recv_hi, recv_hi_n, self.rank + 1)):
                if send and sn:
                    out = self._d2h(send, sn, stream)
                    ops.append(dist.P2POp(dist.isend, out, self._global(peer), self.group))
                if recv and rn:
                    inc = torch.empty(rn, dtype=torch.uint8)
                    ops.append(dist.P2POp(dist.irecv, inc, self._global(peer), self.group))
                    recvs.append((recv, inc))
            for w in dist.batch_isend_irecv(ops) if ops else []:
                w.wait()
            if self.drop_every and self.exchanges % self.drop_every == 0:
                return 0  # (test hook: the ghost data never arrive)
            for recv, inc in recvs:
                self._h2d(recv, inc)
            return 0
        except Exception as e:  # never let an exception cross the C boundary
            print("TorchDistComm.exchange failed:", e, flush=True)
            return 1

    def _exchange2(self, user, send_lo, recv_lo, send_hi, recv_hi, stream):
        """exchange with two segments per message (segment 0 of every message first, then segment 1)"""
        try:
            self.exchanges += 1
            self.segmented_exchanges += 1
            ops, recvs = [], []
            for q in range(2):
                for send, recv, peer in ((send_lo, recv_lo, self.rank - 1), (send_hi, recv_hi, self.rank + 1)):
                    if send and send.contents.bytes[q]:
                        self.bytes_sent += send.contents.bytes[q]
                        out = self._d2h(send.contents.ptr[q], send.contents.bytes[q], stream)
                        ops.append(dist.P2POp(dist.isend, out, self._global(peer), self.group))
                    if recv and recv.contents.bytes[q]:
                        inc = torch.empty(recv.contents.bytes[q], dtype=torch.uint8)
                        ops.append(dist.P2POp(dist.irecv, inc, self._global(peer), self.group))
                        recvs.append((recv.contents.ptr[q], inc))
            for w in dist.batch_isend_irecv(ops) if ops else []:
                w.wait()
            if self.drop_every and self.exchanges % self.drop_every == 0:
                return 0  # (test hook)
            for ptr, inc in recvs:
                self._h2d(ptr, inc)
            return 0
        except Exception as e:  # never let an exception cross the C boundary
            print("TorchDistComm.exchange2 failed:", e, flush=True)
            return 1

    def _allreduce(self, user, values, count, op):
        try:
            t = torch.tensor([values[i] for i in range(count)], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.SUM if op == 0 else dist.ReduceOp.MAX, group=self.group)
            for i in range(count):
                values[i] = float(t[i])
            return 0
        except Exception as e:
            print("TorchDistComm.allreduce failed:", e, flush=True)
            return 1

    def _allreduce_device(self, user, values_dev, count, op, stream):
        """device doubles, in place: staged through the host here (RCCL does it on the stream)"""
        try:
            self.device_allreduces += 1
            t = self._d2h(values_dev, 8 * count, stream).view(torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.SUM if op == 0 else dist.ReduceOp.MAX, group=self.group)
            self._h2d(values_dev, t.view(torch.uint8))
            return 0
        except Exception as e:
            print("TorchDistComm.allreduce_device failed:", e, flush=True)
            return 1

    def _gather(self, user, send, recv, nbytes, root, stream):
        try:
            mine = self._d2h(send, nbytes, stream)
            parts = [torch.empty(nbytes, dtype=torch.uint8) for _ in range(self.size)] if self.rank == root else None
            dist.gather(mine, parts, dst=self._global(root), group=self.group)
            if self.rank == root:
                self._h2d(recv, torch.cat(parts))
            return 0
        except Exception as e:
            print("TorchDistComm.gather failed:", e, flush=True)
            return 1

    def _scatter(self, user, send, recv, nbytes, root, stream):
        try:
            parts = None
            if self.rank == root:
                whole = self._d2h(send, nbytes * self.size, stream)
                parts = [p.contiguous() for p in whole.split(nbytes)]
            mine = torch.empty(nbytes, dtype=torch.uint8)
            dist.scatter(mine, parts, src=self._global(root), group=self.group)
            self._h2d(recv, mine)
            return 0
        except Exception as e:
            print("TorchDistComm.scatter failed:", e, flush=True)
            return 1

    def _gatherv(self, user, send, send_bytes, recv, counts, displs, root, stream):
        try:
            mine = self._d2h(send, send_bytes, stream) if send_bytes else torch.empty(0, dtype=torch.uint8)
            if self.rank == root:
                for r in range(self.size):
                    part = mine if r == root else torch.empty(counts[r], dtype=torch.uint8)
                    if r != root and counts[r]:
                        dist.recv(part, src=self._global(r), group=self.group)
                    if part.numel():
                        self._h2d(recv + displs[r], part)
            elif send_bytes:
                dist.send(mine, dst=self._global(root), group=self.group)
            return 0
        except Exception as e:
            print("TorchDistComm.gatherv failed:", e, flush=True)
            return 1

    def _scatterv(self, user, send, counts, displs, recv, recv_bytes, root, stream):
        try:
            if self.rank == root:
                for r in range(self.size):
                    if not counts[r]:
                        continue
                    part = self._d2h(send + displs[r], counts[r], stream)
                    if r == root:
                        self._h2d(recv, part)
                    else:
                        dist.send(part, dst=self._global(r), group=self.group)
            elif recv_bytes:
                mine = torch.empty(recv_bytes, dtype=torch.uint8)
                dist.recv(mine, src=self._global(root), group=self.group)
                self._h2d(recv, mine)
            return 0
        except Exception as e:
            print("TorchDistComm.scatterv failed:", e, flush=True)
            return 1

    def close(self):
        pass


def comm_preflight(comm, floats=1 << 16):
    """mgps_comm_preflight: rank-stamped data through every entry of the transport, verified on arrival.  Collective.  Returns the
    number of ranks the (device) all-reduce counted; raises when something arrives wrong."""
    seen = C.c_int()
    lib().mgps_comm_preflight.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(C.c_int)]
    check(lib().mgps_comm_preflight(C.byref(comm.struct), int(floats), C.byref(seen)))
    return seen.value


def slab_partition(labels, mg_levels, size, use_gauss_seidel, options=None):
    """mgps_slab_partition: the cuts of a `size`-rank slab run, balanced by active cells where the smoother allows it.
    Returns a list of size + 1 plane indices; rank r owns [cuts[r], cuts[r + 1])."""
    labels = _np_u8(labels)
    nz, ny, nx = labels.shape
    cuts = (C.c_int * (size + 1))()
    opt = options if options is not None else default_options()
    check(lib().mgps_slab_partition(nx, ny, nz, _p(labels), int(mg_levels), int(size), int(bool(use_gauss_seidel)), C.byref(opt), cuts))
    return [int(v) for v in cuts]


class SlabSolver(GeometricMultigridPoissonSolver):
    """GeometricMultigridPoissonSolver on one Z-slab per rank (mgps_create_slab).

    labels: the WHOLE solver grid (nz, ny, nx) uint8; weights: this rank's slab only -- wx, wy with
    nz/size planes, wz with nz/size + 1.  Grids of this solver hold the rank's owned planes; they
    come from new_grid()/to_device(), which surround them with the two ghost planes the exchange
    writes into."""

    def __init__(self, labels, slab_weights, mg_levels, use_gauss_seidel, comm, device=None, options=None, splits=None):
        """splits: the cuts (slab_partition); None = nz / size planes per rank.
        slab_weights: numpy arrays (mgps_create_slab_ranges) or float32 CUDA tensors (mgps_create_slab_device_weights: nothing of
        the weights crosses PCIe)."""
        labels = _np_u8(labels)
        on_device = all(isinstance(a, torch.Tensor) and a.is_cuda for a in slab_weights)
        if on_device:
            assert all(a.dtype == torch.float32 and a.is_contiguous() for a in slab_weights)
            w = list(slab_weights)
        else:
            w = [_np_f32(a) for a in slab_weights]
        nz, ny, nx = labels.shape
        if splits is None:
            assert nz % comm.size == 0, "nz must divide evenly over the ranks"
            splits = [nz // comm.size * r for r in range(comm.size + 1)]
        assert len(splits) == comm.size + 1
        nzl = splits[comm.rank + 1] - splits[comm.rank]
        assert tuple(w[0].shape) == (nzl, ny, nx + 1) and tuple(w[1].shape) == (nzl, ny + 1, nx) and tuple(w[2].shape) == (nzl + 1, ny, nx)
        self.splits = [int(v) for v in splits]
        opt = options if options is not None else default_options()
        if device is not None:
            opt.device = torch.device(device).index if not isinstance(device, int) else device
        self.comm = comm
        self.h = C.c_void_p()
        cuts = (C.c_int * (comm.size + 1))(*self.splits)
        if on_device:
            torch.cuda.synchronize()  # (the library reads the tensors on its own stream)
            ptr = [C.c_void_p(a.data_ptr()) for a in w]
            check(
                lib().mgps_create_slab_device_weights(
                    C.byref(self.h), nx, ny, nz, _p(labels), ptr[0], ptr[1], ptr[2], int(mg_levels),
                    int(bool(use_gauss_seidel)), C.byref(opt), C.byref(comm.struct), cuts,
                )
            )
        else:
            check(
                lib().mgps_create_slab_ranges(
                    C.byref(self.h), nx, ny, nz, _p(labels), _p(w[0]), _p(w[1]), _p(w[2]), int(mg_levels),
                    int(bool(use_gauss_seidel)), C.byref(opt), C.byref(comm.struct), cuts,
                )
            )
        self.shape = (nzl, ny, nx)
        self.global_shape = (nz, ny, nx)
        self.use_gauss_seidel = bool(use_gauss_seidel)
        dev_index = opt.device if opt.device >= 0 else torch.cuda.current_device()
        self.device = torch.device("cuda", dev_index)
        self.use_torch_stream()

    @property
    def distributed_levels(self):
        return lib().mgps_distributed_levels(self.h)

    def band_stage_form(self, level):
        """how the band stage of a distributed level runs: "boxes" (one launch per stage; on a cut level two list messages per
        stroke) or "passes" (a launch pair and an exchange per band pass)"""
        form = C.c_int()
        check(lib().mgps_band_stage_form(self.h, int(level), C.byref(form)), self.h)
        return {0: "passes", 1: "boxes"}[form.value]

    @property
    def exchange_count(self):
        lib().mgps_exchange_count.restype = C.c_int64
        return lib().mgps_exchange_count(self.h)

    def slab_range(self, level=0):
        z0, z1 = C.c_int(), C.c_int()
        check(lib().mgps_slab_range(self.h, level, C.byref(z0), C.byref(z1)), self.h)
        return z0.value, z1.value

    @property
    def ghost_planes(self):
        """planes every grid of this solver carries below and above its owned ones (mgps_ghost_planes)"""
        return int(lib().mgps_ghost_planes(self.h))

    def new_grid(self, level=0):
        nz, ny, nx = self.level_shape(level)
        g = self.ghost_planes
        padded = torch.zeros((nz + 2 * g, ny, nx), dtype=torch.float32, device=self.device)
        return padded[g:-g]  # contiguous view; the storage keeps the ghost planes alive

    def to_device(self, a, level=0):
        g = self.new_grid(level)
        g.copy_(torch.from_numpy(_np_f32(a)))
        return g

    def gather_global(self, grid, level=0):
        """All ranks' owned planes of `grid`, concatenated in z, as a numpy array on every rank."""
        group = getattr(self.comm, "group", None)
        local = grid.detach().contiguous()
        if dist.get_backend(group) != "nccl":
            local = local.cpu()
        planes = [(self.splits[r + 1] - self.splits[r]) >> level for r in range(self.comm.size)]
        most = max(planes)
        padded = torch.zeros((most,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        padded[: local.shape[0]] = local
        parts = [torch.empty_like(padded) for _ in range(self.comm.size)]
        dist.all_gather(parts, padded, group=group)
        return torch.cat([p[:n] for p, n in zip(parts, planes)]).cpu().numpy()
