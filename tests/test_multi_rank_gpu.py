"""Two ranks running lili_s2m_iterate_sharded ITSELF: the C loop with its two collectives enqueued between the
kernels, queries block-sharded, map replicated — two processes that share GPU 0 (the pool has one GPU per box; the loop, the
kernels and the exchange are the ones an 8-GPU node runs, only the link is HBM instead of xGMI).
  * all-reduce backed by gloo (a host callback with ncclAllReduce's signature: stream sync, D2H, all_gather, sum in rank order, H2D);
  * all-reduce = lili_p2p_allreduce (hipIpc-mapped mailboxes, sums in rank order on the device): folded into the count kernel and
    the reduce + Gauss-Newton kernel (4 launches per iteration), and as separate launches (the generic lili_allreduce_fn path).
Both ranks must end with bit-identical poses, equal to the single-rank fused iterations up to the summation tree (<= 1e-12)."""
import ctypes as C
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

N_ITERS = 5
SEED = 27


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _scene():
    import lili_om_amd as L
    from lili_om_amd import synth
    room = synth.make_room(seed=SEED, n_query=6001, n_edge_query=301)
    P = L.make_params("rot")
    tb, qb = L.api.body_pose_from_lidar(room["t_true"], room["q_true"], P)
    t0, q0 = synth.perturbed_pose(tb, qb, np.random.default_rng(5), 0.05, 0.5)
    return room, P, t0, q0


def _matcher(ctx, room, P, lo_s, hi_s, lo_e, hi_e):
    import lili_om_amd as L
    m = L.ScanToMapMatcher(ctx, P)
    m.set_input_cloud(L.KIND_SURF, room["map_xyz"])
    m.set_input_cloud(L.KIND_EDGE, room["edge_map_xyz"])
    m.set_queries(0, L.KIND_SURF, room["q_xyz"][lo_s:hi_s])
    m.set_queries(0, L.KIND_EDGE, room["eq_xyz"][lo_e:hi_e])
    return m


def _worker(rank, world, port, mode, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch
    import torch.distributed as dist
    import lili_om_amd as L
    from lili_om_amd import sharding
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    room, P, t0, q0 = _scene()
    ctx = L.Context(0)
    lo_s, hi_s = sharding.shard_bounds(room["q_xyz"].shape[0], world, rank)
    lo_e, hi_e = sharding.shard_bounds(room["eq_xyz"].shape[0], world, rank)
    m = _matcher(ctx, room, P, lo_s, hi_s, lo_e, hi_e)
    mask = L.MASK_SURF | L.MASK_EDGE
    counts = torch.zeros(2, dtype=torch.int32, device="cuda")
    gram = torch.zeros(L.api.GRAM_DOUBLES, dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    calls = [0]
    comm = None
    if mode == "gloo":
        CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p)

        def host_allreduce(send, recv, count, dtype, op, comm_, stream):
            buf = counts if send == counts.data_ptr() else gram
            if send != recv or send != buf.data_ptr() or count != buf.numel() or op != 0:
                return 1
            ctx.sync()                                   # everything enqueued so far on the context's stream
            host = buf.cpu()
            parts = [torch.zeros_like(host) for _ in range(world)]
            dist.all_gather(parts, host)
            total = parts[0].clone()
            for r in range(1, world):
                total += parts[r]                        # rank order on every rank: identical bits
            buf.copy_(total)
            torch.cuda.synchronize()
            calls[0] += 1
            return 0

        cb = CB(host_allreduce)
        fn, handle = C.cast(cb, C.c_void_p).value, None
    else:
        from lili_om_amd import p2p
        comm = p2p.Communicator(ctx, rank, world, dist)
        fn, handle = comm.allreduce_fn, comm.handle
        if mode == "p2p_staged":                 # the generic path: one k_p2p_allreduce launch per collective (7 launches per iteration)
            ctx.set_option("p2p_fusion", 0)
    m.pose_set(0, t0, q0)
    dist.barrier()
    m.iterate_sharded(0, N_ITERS, counts.data_ptr(), gram.data_ptr(), fn, handle, kind_mask=mask)
    ctx.sync()
    t, q, st = m.pose_get(0)
    status = comm.status() if comm is not None else 0
    out[rank] = (t.copy(), q.copy(), int(st), gram.cpu().numpy().copy(), counts.cpu().numpy().copy(), calls[0], status)
    dist.barrier()
    if comm is not None:
        comm.close()
    ctx.close()
    dist.destroy_process_group()


def _single_rank():
    import lili_om_amd as L
    room, P, t0, q0 = _scene()
    ctx = L.Context(0)
    m = _matcher(ctx, room, P, 0, room["q_xyz"].shape[0], 0, room["eq_xyz"].shape[0])
    m.pose_set(0, t0, q0)
    m.iterate(0, N_ITERS, L.MASK_SURF | L.MASK_EDGE)
    ctx.sync()
    t, q, st = m.pose_get(0)
    ctx.close()
    assert st == 0 and np.abs(t - t0).max() > 1e-3
    return t, q


@pytest.mark.parametrize("mode", ["gloo", "p2p", "p2p_staged"])
def test_two_ranks_share_one_gpu(mode):
    import torch.multiprocessing as mp
    world = 2
    t_ref, q_ref = _single_rank()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), mode, out), nprocs=world, join=True)
    assert sorted(out.keys()) == [0, 1]
    for r in range(world):
        t, q, st, gram, counts, calls, status = out[r]
        assert st == 0 and status == 0
        if mode == "gloo":
            assert calls == 2 * N_ITERS                        # counts + Gram per iteration (ROT: count-scaled residuals)
        assert np.abs(t - t_ref).max() < 1e-12 and np.abs(q - q_ref).max() < 1e-12     # only the summation tree differs
        assert counts[0] > 2000 and counts[1] > 50             # GLOBAL counts on every rank
    a, b = out[0], out[1]
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])                    # the ranks agree bit for bit
    assert np.array_equal(a[3], b[3]) and np.array_equal(a[4], b[4])                    # ... on the reduced record as well
