"""Gradient exchange between the data-parallel ranks: the RCCL communicator owned by liblasr (``lasr_comm_*`` in
include/lasr.h), i.e. what Lightning's DDP plugin does for the reference (conf/conf.yaml:30 ``accelerator: ddp``).

``torch.distributed`` is used for ONE thing here: carrying the 128-byte RCCL unique id from rank 0 to the other ranks
(any rendez-vous would do).  The collectives themselves are ``ncclAllReduce`` / ``ncclBroadcast`` calls made by the library
on its own side stream, ordered against the compute stream by events (no host synchronisation)."""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Sequence, Tuple

import torch

from . import _lib
from ._lib import call

ID_BYTES = 128


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


class Communicator:
    def __init__(self, unique_id: bytes, world: int, rank: int, device: torch.device):
        if len(unique_id) != ID_BYTES:
            raise ValueError("RCCL unique id must be %d bytes" % ID_BYTES)
        self.world, self.rank = int(world), int(rank)
        self.device = torch.device(device)
        if self.device.type != "cuda" or self.device.index is None:
            raise _lib.LasrError("Communicator needs an indexed GPU device (got %s)" % (device,))
        h = C.c_void_p()
        buf = C.create_string_buffer(unique_id, ID_BYTES)
        call("lasr_comm_init", C.byref(h), buf, ID_BYTES, self.world, self.rank, self.device.index)
        self._h = h

    # ---- construction ---------------------------------------------------------------------------------------------
    @staticmethod
    def new_unique_id() -> bytes:
        buf = C.create_string_buffer(ID_BYTES)
        call("lasr_comm_unique_id", buf, ID_BYTES)
        return bytes(buf.raw)

    @classmethod
    def single(cls, device) -> "Communicator":
        """a 1-rank communicator (tests the whole side-stream path on one GPU)"""
        return cls(cls.new_unique_id(), 1, 0, device)

    @classmethod
    def from_torch_distributed(cls, device, group=None) -> "Communicator":
        """rank / world / unique-id bootstrap from an initialised torch.distributed group (any backend)"""
        import torch.distributed as dist
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        box = [cls.new_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        return cls(box[0], world, rank, device)

    def close(self) -> None:
        if getattr(self, "_h", None):
            _lib.load().lasr_comm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- collectives (asynchronous; in place on f32 buffers) -----------------------------------------------------------
    @staticmethod
    def _check(t: torch.Tensor) -> None:
        if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
            raise TypeError("collectives run in place on contiguous f32 GPU buffers")

    def all_reduce(self, flat: torch.Tensor) -> None:
        """SUM over ranks of ``flat``, after everything enqueued so far on the current stream"""
        self._check(flat)
        call("lasr_comm_allreduce", self._h, flat.data_ptr(), flat.numel(), _stream())

    def all_reduce_ranges(self, flat: torch.Tensor, ranges: Sequence[Tuple[int, int]]) -> None:
        """one bucket = several [lo, hi) pieces of the flat gradient, one grouped RCCL launch"""
        self._check(flat)
        n = len(ranges)
        lo = (C.c_int64 * n)(*[r[0] for r in ranges])
        hi = (C.c_int64 * n)(*[r[1] for r in ranges])
        if any(h > flat.numel() for h in hi):
            raise ValueError("range outside the buffer")
        call("lasr_comm_allreduce_ranges", self._h, flat.data_ptr(), lo, hi, n, _stream())

    def broadcast(self, flat: torch.Tensor, root: int = 0) -> None:
        self._check(flat)
        call("lasr_comm_broadcast", self._h, flat.data_ptr(), flat.numel(), root, _stream())

    def wait(self) -> None:
        """the current stream waits for every collective issued so far (device-side; the host does not block)"""
        call("lasr_comm_wait", self._h, _stream())

    # ---- timing (bench.py's `comm` record) -----------------------------------------------------------------------------------
    def timing(self, on: bool) -> None:
        """bracket every collective (side stream) and every wait (consumer stream) with events from now on; eager launches only"""
        call("lasr_comm_timing", self._h, int(bool(on)))

    def timing_collect(self, max_recs: int = 4096):
        """([(us, bytes) per collective, in call order], [us each ``wait`` actually stalled the consumer stream])"""
        us = (C.c_double * max_recs)()
        by = (C.c_double * max_recs)()
        wt = (C.c_double * max_recs)()
        nc, nw = C.c_int32(0), C.c_int32(0)
        call("lasr_comm_timing_collect", self._h, max_recs, us, by, C.byref(nc), wt, C.byref(nw))
        return [(us[i], by[i]) for i in range(min(nc.value, max_recs))], [wt[i] for i in range(min(nw.value, max_recs))]
