"""Communicators of the multi-GPU exchange: ctypes wrapper of the dm_comm_* part of the C ABI (include/dismember_hip.h).

The reference's only collective is the in-process average of its worker threads' gradient buffers
(tdm/src/main/scala/com/mass/tdm/optim/LocalOptimizer.scala:164-187).  Here a worker is one process per GPU and the
exchange lives in the library: RCCL over xGMI between GPUs (transport "rccl"), or a TCP star with host staging when several
workers share one GPU / for CPU-only processes (transport "host").  No torch anywhere.
"""
import ctypes as C
import os

import numpy as np

from . import _native as N

TRANSPORTS = {"host": 0, "rccl": 1}


class CommError(RuntimeError):
    pass


class Comm:
    """One rank of a communicator.  `Comm.from_env()` reads the torchrun-style environment (RANK, WORLD_SIZE,
    MASTER_ADDR, MASTER_PORT); the rendezvous port defaults to MASTER_PORT + 29 so that it does not collide with the
    launcher's own store."""

    def __init__(self, nranks, rank, addr="127.0.0.1", port=29629, transport="rccl", device_id=0):
        self._c = C.c_void_p()
        self.transport = transport
        rc = N.lib().dm_comm_create_tcp(int(nranks), int(rank), addr.encode(), int(port), TRANSPORTS[transport],
                                        int(device_id), C.byref(self._c))
        if rc != 0:
            raise CommError("dm_comm_create_tcp failed (%d): %s" % (rc, (N.lib().dm_comm_last_error(None) or b"").decode()))
        self.rank, self.world = int(rank), int(nranks)

    @classmethod
    def from_env(cls, transport="rccl", device_id=None, port_offset=29):
        rank = int(os.environ.get("RANK", "0"))
        world = int(os.environ.get("WORLD_SIZE", "1"))
        local = int(os.environ.get("LOCAL_RANK", "0"))
        addr = os.environ.get("MASTER_ADDR", "127.0.0.1")
        port = int(os.environ.get("DM_COMM_PORT", int(os.environ.get("MASTER_PORT", "29600")) + port_offset))
        return cls(world, rank, addr, port, transport, local if device_id is None else device_id)

    def _chk(self, rc):
        if rc != 0:
            raise CommError("dm_comm error %d: %s" % (rc, (N.lib().dm_comm_last_error(self._c) or b"").decode()))

    def close(self):
        if getattr(self, "_c", None):
            N.lib().dm_comm_destroy(self._c)
            self._c = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def barrier(self):
        self._chk(N.lib().dm_comm_barrier(self._c))

    def allreduce(self, values, op="sum"):
        """In rank order on every rank (bit-identical results); values: float or sequence of floats."""
        v = np.atleast_1d(np.asarray(values, np.float64)).copy()
        self._chk(N.lib().dm_comm_allreduce_f64(self._c, v.ctypes.data_as(C.POINTER(C.c_double)), v.size, {"sum": 0, "max": 1}[op]))
        return v if np.ndim(values) else float(v[0])

    def all_gather_bytes(self, payload):
        """Every rank's bytes, as a list in rank order."""
        payload = bytes(payload)
        sizes = (C.c_uint64 * self.world)()
        buf = C.create_string_buffer(payload, len(payload)) if payload else None
        self._chk(N.lib().dm_comm_all_gather_v(self._c, buf, len(payload), None, 0, sizes))
        total = int(sum(sizes))
        out = C.create_string_buffer(max(total, 1))
        self._chk(N.lib().dm_comm_all_gather_v(self._c, buf, len(payload), out, total, sizes))
        res, o = [], 0
        for r in range(self.world):
            res.append(out.raw[o:o + int(sizes[r])])
            o += int(sizes[r])
        return res

    def all_gather_array(self, arr, axis0_concat=True):
        """Concatenate every rank's array along axis 0 (same dtype and trailing shape on all ranks)."""
        a = np.ascontiguousarray(arr)
        blocks = self.all_gather_bytes(a.tobytes())
        parts = [np.frombuffer(b, dtype=a.dtype).reshape((-1,) + a.shape[1:]) for b in blocks]
        return np.concatenate(parts, axis=0) if axis0_concat else parts


def make_clique(devices):
    """One process driving len(devices) GPUs (the reference's one-JVM shape): raw dm_comm_t handles of an
    ncclCommInitAll clique, rank i on devices[i].  Attach with Engine.attach_comm(handle)."""
    n = len(devices)
    devs = (C.c_int * n)(*[int(d) for d in devices])
    out = (C.c_void_p * n)()
    rc = N.lib().dm_comm_create_all(n, devs, out)
    if rc != 0:
        raise CommError("dm_comm_create_all failed (%d): %s" % (rc, (N.lib().dm_comm_last_error(None) or b"").decode()))
    return [C.c_void_p(out[i]) for i in range(n)]
