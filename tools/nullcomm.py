"""A transport that moves nothing, for timing the compute side of ONE rank of a P-rank slab run on one GPU: exchanges, collectives
and the collapse return at once (the values in the ghost planes are meaningless, the kernels and launches are exactly those of
the real run).  One thing it does deliver: the collapse level's labels that rank 0 gathers at set-up -- coarsened on the host from
the global labels (outside any timed region) -- so that rank 0 builds the tail it would build in a real run."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from geometricmultigridpressuresolver_amd.distributed import _ALLR, _ALLRD, _DEST, _EXCH, _EXCH2, _GATH, _GATHV, _SCATV, CommStruct
from geometricmultigridpressuresolver_amd.solver import Hierarchy


class NullComm:
    def __init__(self, rank, size, labels=None, levels=None):
        self.rank, self.size = rank, size
        self.calls = 0
        self._labels, self._levels, self._coarse = labels, levels, None
        self._hip = C.CDLL("libamdhip64.so")
        self._hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]

        def exch(*a):
            self.calls += 1
            return 0

        def gather(user, send, recv, nbytes, root, stream):
            return self._serve(recv, nbytes * self.size)

        def gatherv(user, send, send_bytes, recv, counts, displs, root, stream):
            return self._serve(recv, sum(counts[r] for r in range(self.size)))

        self._cb = (_EXCH(exch), _ALLR(lambda *a: 0), _GATH(gather), _GATH(lambda *a: 0))
        self._cbv = (_GATHV(gatherv), _SCATV(lambda *a: 0), _ALLRD(lambda *a: 0), _EXCH2(exch))
        self.struct = CommStruct(C.sizeof(CommStruct), rank, size, None, *self._cb, _DEST(), *self._cbv)

    def prepare(self):
        """the coarse labels a set-up may ask for, made ahead of any timed region (rank 0 only)"""
        if self._labels is None or self.rank != 0 or self._coarse is not None:
            return
        hier = Hierarchy(self._labels, self._levels)
        self._coarse = {lev: np.ascontiguousarray(hier.level_labels(lev)) for lev in range(1, hier.levels)}
        hier.close()

    def _serve(self, recv, total):
        """the set-up's gather (a byte per cell of the collapse level; the cycle's gathers move floats): the collapse level's labels"""
        if not recv or self._labels is None:
            return 0
        self.prepare()
        for lab in self._coarse.values():
            if lab.nbytes == total:
                assert self._hip.hipMemcpy(recv, lab.ctypes.data, lab.nbytes, 1) == 0
                break
        return 0
