"""Peer-to-peer all-reduce for the sharded matcher loop (include/lili_hip.h: lili_p2p_*): the mailbox
exchange lives in the library (lili_om_amd/csrc/lili_p2p.hip); this module only moves the 64-byte hipIpc handles between the
ranks over an existing torch.distributed group (gloo or RCCL) and hands the result to lili_s2m_iterate_sharded:

    comm = p2p.Communicator(ctx, rank, world)          # collective: every rank calls it
    matcher.iterate_sharded(slot, n, counts_ptr, gram_ptr, comm.allreduce_fn, comm.handle, ...)
"""
import ctypes as C

from . import api

HANDLE_BYTES = 64


class Communicator:
    def __init__(self, ctx, rank, world, dist=None):
        """Collective over `dist` (torch.distributed, default group).  Every rank takes part in the same two group operations (handle
        all-gather, barrier) whatever fails locally, and ALL ranks raise if any rank failed — a rank that cannot create or map a mailbox
        must not leave the others waiting in a collective."""
        self.lib = api.load_library()
        self.ctx = ctx
        self.handle = None
        self.rank, self.world = rank, world
        err = None
        mine = (C.c_ubyte * HANDLE_BYTES)()
        try:
            h = C.c_void_p()
            ctx._chk(self.lib.lili_p2p_create(ctx.h, int(rank), int(world), C.byref(h)))
            self.handle = h.value
            ctx._chk(self.lib.lili_p2p_handle(C.c_void_p(self.handle), mine))
        except Exception as e:          # noqa: BLE001
            err = e
        if world > 1:
            if dist is None:
                import torch.distributed as dist
            box = [None] * world
            dist.all_gather_object(box, None if err is not None else bytes(mine))
            if err is None and any(b is None for b in box):
                err = RuntimeError("lili_p2p: a peer could not create its mailbox")
            if err is None:
                try:
                    blob = b"".join(box)
                    assert len(blob) == world * HANDLE_BYTES
                    ctx._chk(self.lib.lili_p2p_connect(C.c_void_p(self.handle), C.c_char_p(blob)))
                except Exception as e:  # noqa: BLE001
                    err = e
            ok = [None] * world
            dist.all_gather_object(ok, err is None)   # doubles as the barrier: every rank has mapped every mailbox before the first store goes out
            if err is None and not all(ok):
                err = RuntimeError("lili_p2p: a peer could not map the mailboxes")
        if err is not None:
            self.close()
            raise err
        self.allreduce_fn = C.cast(self.lib.lili_p2p_allreduce, C.c_void_p).value

    def all_reduce(self, ptr, count, dtype, stream):
        """In-place sum on `stream` (host-side convenience for tests); dtype 2 = int32, 8 = f64."""
        rc = self.lib.lili_p2p_allreduce(C.c_void_p(ptr), C.c_void_p(ptr), count, dtype, 0, C.c_void_p(self.handle), C.c_void_p(stream))
        if rc != 0:
            raise RuntimeError("lili_p2p_allreduce failed")

    def status(self):
        return self.lib.lili_p2p_status(C.c_void_p(self.handle))

    def set_timeout(self, seconds):
        """How long an exchange waits for a peer (device time, default 10 s) before the communicator fails for good."""
        self.ctx._chk(self.lib.lili_p2p_set_timeout(C.c_void_p(self.handle), float(seconds)))

    def close(self):
        if getattr(self, "handle", None):
            self.lib.lili_p2p_destroy(C.c_void_p(self.handle))
            self.handle = None
