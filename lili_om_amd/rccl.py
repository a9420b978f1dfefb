"""RCCL for the sharded matcher loop: the two tiny all-reduces per iteration (correspondence counts, Gram record)
are enqueued by lili_s2m_iterate_sharded from C, directly between the kernels on the context's HIP stream, through the
`ncclAllReduce` of the librccl.so this process already has (PyTorch's).  This module only creates the communicator — the
library itself does not link RCCL.  One process per GPU; the ncclUniqueId travels over the existing torch.distributed group.

    comm = rccl.Communicator(rank, world, device=local_gpu)        # collective: every rank calls it
    matcher.iterate_sharded(slot, n, counts_ptr, gram_ptr, comm.allreduce_fn, comm.handle, ...)
"""
import ctypes as C
import os
import threading

NCCL_UNIQUE_ID_BYTES = 128
ncclInt32, ncclFloat64, ncclSum = 2, 8, 0


class _UniqueId(C.Structure):
    _fields_ = [("internal", C.c_byte * NCCL_UNIQUE_ID_BYTES)]


def find_library():
    """The RCCL PyTorch loaded (one RCCL per process), or LILI_RCCL_LIB."""
    p = os.environ.get("LILI_RCCL_LIB")
    if p:
        return p
    import torch
    cand = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
    if os.path.exists(cand):
        return cand
    return "librccl.so"


class Communicator:
    def __init__(self, rank, world, device=None, timeout_s=60.0):
        """device: HIP device ordinal of this rank (default: torch's current device).  The communicator is created in a helper
        thread (so that a stuck bootstrap cannot hang the caller); the HIP current device is per thread, hence set there too."""
        import torch
        import torch.distributed as dist
        if device is None:
            device = torch.cuda.current_device()
        self.lib = C.CDLL(find_library())
        L = self.lib
        L.ncclGetUniqueId.restype = C.c_int
        L.ncclGetUniqueId.argtypes = [C.POINTER(_UniqueId)]
        L.ncclCommInitRank.restype = C.c_int
        L.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, _UniqueId, C.c_int]
        L.ncclCommDestroy.restype = C.c_int
        L.ncclCommDestroy.argtypes = [C.c_void_p]
        L.ncclAllReduce.restype = C.c_int
        L.ncclAllReduce.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        uid = _UniqueId()
        box = [None]
        if rank == 0 and L.ncclGetUniqueId(C.byref(uid)) == 0:
            box = [bytes(uid.internal)]
        if world > 1:
            dist.broadcast_object_list(box, src=0)      # always reached by every rank, also when rank 0 has no id to offer
        if box[0] is None:
            raise RuntimeError("ncclGetUniqueId failed on rank 0")
        C.memmove(C.byref(uid), box[0], NCCL_UNIQUE_ID_BYTES)
        comm = C.c_void_p()
        result = {}

        def init():
            torch.cuda.set_device(device)
            result["rc"] = L.ncclCommInitRank(C.byref(comm), int(world), uid, int(rank))

        th = threading.Thread(target=init, daemon=True)
        th.start()
        th.join(timeout_s)
        if th.is_alive():
            raise TimeoutError("ncclCommInitRank did not return")
        if result.get("rc", 1) != 0 or not comm.value:
            raise RuntimeError(f"ncclCommInitRank failed ({result.get('rc')})")
        self.handle = comm.value
        self.allreduce_fn = C.cast(L.ncclAllReduce, C.c_void_p).value
        self.rank, self.world = rank, world

    def all_reduce(self, ptr, count, dtype, stream):
        """In-place sum (host-side convenience for tests)."""
        rc = self.lib.ncclAllReduce(C.c_void_p(ptr), C.c_void_p(ptr), count, dtype, ncclSum, C.c_void_p(self.handle), C.c_void_p(stream))
        if rc != 0:
            raise RuntimeError(f"ncclAllReduce failed ({rc})")

    def close(self):
        if getattr(self, "handle", None):
            self.lib.ncclCommDestroy(C.c_void_p(self.handle))
            self.handle = None
