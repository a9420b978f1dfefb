"""World-size-2 gloo test (CPU) of the sharded scan-to-map iteration: block-sharding the queries, all-reducing the
counts and the Gram record, and applying the same GN step on every rank reproduces the single-rank result.
The per-shard association / linearisation is done by the ORACLE here (the HIP kernels need a GPU); what is under
test is the sharding + reduction algebra that bench.py --gpus N uses (lili_om_amd/sharding.py)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from lili_om_amd import sharding, synth
import lili_om_amd as L


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _one_iteration(O, room, lo, hi, t, q, P, PO, reduce_fn):
    tree = O.KdTree(room["map_xyz"])
    Q2, T2 = L.api.assoc_transform(t, q, P)
    rs = O.associate_surf(tree, None, room["q_xyz"][lo:hi], None, Q2, T2, PO)
    counts = torch.tensor([rs["count"], 0], dtype=torch.int32)
    counts = reduce_fn("counts", counts)
    G, cost, n = O.linearize_surf(rs, t, q, PO, (1000.0, max(int(counts[0]), 1)))
    rec = torch.zeros(72, dtype=torch.float64)
    rec[:64] = torch.from_numpy(G.reshape(-1))
    rec[64], rec[65] = cost, n
    rec = reduce_fn("gram", rec)
    st, t2, q2, _ = L.api.gn_step_host(rec[:64].numpy(), t, q)
    return st, t2, q2, rec


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle as O
    room = synth.make_room(seed=21, n_query=3001, n_edge_query=10)
    P, PO = L.make_params("rot"), O.params("rot")
    tb, qb = L.api.body_pose_from_lidar(room["t_true"], room["q_true"], P)
    t, q = synth.perturbed_pose(tb, qb, np.random.default_rng(2), 0.1, 0.5)
    lo, hi = sharding.shard_bounds(room["q_xyz"].shape[0], world, rank)

    def red(kind, x):
        return sharding.allreduce_counts(dist, x) if kind == "counts" else sharding.allreduce_gram(dist, x)
    for _ in range(3):
        st, t, q, rec = _one_iteration(O, room, lo, hi, t, q, P, PO, red)
        assert st == 0
    out[rank] = (t.copy(), q.copy(), rec.numpy().copy())
    dist.destroy_process_group()


def test_two_rank_sharded_iteration_matches_single_rank(oracle):
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    # single-rank reference
    room = synth.make_room(seed=21, n_query=3001, n_edge_query=10)
    P, PO = L.make_params("rot"), oracle.params("rot")
    tb, qb = L.api.body_pose_from_lidar(room["t_true"], room["q_true"], P)
    t, q = synth.perturbed_pose(tb, qb, np.random.default_rng(2), 0.1, 0.5)
    for _ in range(3):
        st, t, q, rec = _one_iteration(oracle, room, 0, room["q_xyz"].shape[0], t, q, P, PO, lambda k, x: x)
    for r in range(world):
        tr, qr, recr = out[r]
        assert np.abs(tr - t).max() < 1e-12 and np.abs(qr - q).max() < 1e-12      # summation tree differs: ~1e-16 relative
        assert np.abs(recr[:64] - rec[:64].numpy()).max() <= 1e-12 * np.abs(rec[:64].numpy()).max()
        assert recr[65] == rec[65]
    assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1])   # ranks agree bit for bit


def test_shard_bounds_cover_everything_once():
    for n in (0, 1, 7, 200000, 200001):
        for world in (1, 2, 4, 8):
            seen = np.zeros(n, int)
            for r in range(world):
                lo, hi = sharding.shard_bounds(n, world, r)
                assert 0 <= lo <= hi <= n
                seen[lo:hi] += 1
            assert (seen == 1).all()


def test_project_gram_matches_host_gn_step(oracle):
    rng = np.random.default_rng(0)
    J = rng.normal(size=(100, 8))
    G = J.T @ J
    q = rng.normal(size=4); q /= np.linalg.norm(q)
    H, g = sharding.project_gram(G, q)
    d = np.linalg.solve(H, -g)
    st, t2, q2, delta = L.api.gn_step_host(G, np.zeros(3), q)
    assert st == 0 and np.allclose(delta, d, rtol=1e-9, atol=1e-12)


# ---- the sliding window across ranks (BASELINE configs[4]): K keyframes, every rank holds its shard of every keyframe's queries; per
# evaluation ONE all-reduce of the packed K x 2 counts and ONE of the packed K x 72 records
N_KF = 3


def _window_scene():
    room = synth.make_room(seed=23, n_query=2401, n_edge_query=10)
    P = L.make_params("rot")
    tb, qb = L.api.body_pose_from_lidar(room["t_true"], room["q_true"], P)
    poses = [synth.perturbed_pose(tb, qb, np.random.default_rng(40 + k), 0.08, 0.6) for k in range(N_KF)]
    # keyframe k sees its own third of the feature cloud
    qs = [room["q_xyz"][k::N_KF] for k in range(N_KF)]
    return room, P, poses, qs


def _window_evaluation(O, room, P, PO, poses, qs, world, rank, reduce_fn):
    tree = O.KdTree(room["map_xyz"])
    recs, counts = [], torch.zeros(2 * N_KF, dtype=torch.int32)
    for k in range(N_KF):
        lo, hi = sharding.shard_bounds(qs[k].shape[0], world, rank)
        Q2, T2 = L.api.assoc_transform(poses[k][0], poses[k][1], P)
        recs.append(O.associate_surf(tree, None, qs[k][lo:hi], None, Q2, T2, PO))
        counts[2 * k] = recs[-1]["count"]
    counts = reduce_fn(counts)                                     # ONE collective for the counts of all keyframes
    rows = []
    for k in range(N_KF):
        G, cost, n = O.linearize_surf(recs[k], poses[k][0], poses[k][1], PO, (1000.0, max(int(counts[2 * k]), 1)))
        rec = np.zeros(72)
        rec[:64], rec[64], rec[65] = G.reshape(-1), cost, n
        rows.append(rec)
    buf = torch.from_numpy(sharding.pack_window_records(rows))
    buf = reduce_fn(buf)                                           # ONE collective for the K records
    return sharding.unpack_window_records(buf.numpy(), N_KF)


def _window_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle as O
    room, P, poses, qs = _window_scene()
    calls = [0]

    def red(x):
        calls[0] += 1
        return sharding.allreduce_window(dist, x)
    res = _window_evaluation(O, room, P, O.params("rot"), poses, qs, world, rank, red)
    out[rank] = (res, calls[0])
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_window_evaluation_matches_single_rank(oracle, world):
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_window_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    room, P, poses, qs = _window_scene()
    ref = _window_evaluation(oracle, room, P, oracle.params("rot"), poses, qs, 1, 0, lambda x: x)
    for r in range(world):
        res, calls = out[r]
        assert calls == 2                                           # two collectives per evaluation whatever the number of keyframes
        for k in range(N_KF):
            G, cost, cnt = res[k]
            assert cnt == ref[k][2] and cnt[0] > 400
            assert np.abs(G - ref[k][0]).max() <= 1e-12 * np.abs(ref[k][0]).max() and abs(cost - ref[k][1]) <= 1e-12 * ref[k][1]
            assert np.array_equal(G, out[0][0][k][0])               # the ranks agree bit for bit (gloo reduces in one order for all)


# ---- slot-per-rank window (round 5: lili_s2m_*_window_gather): keyframe k lives on rank k mod world at FULL size; ONE exchange of K x 72 doubles per evaluation in
# which every record has a single non-zero contributor = an all-gather; no count exchange (the owner's count is the global one)
def _gather_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle as O
    room, P, poses, qs = _window_scene()
    PO = O.params("rot")
    owner = sharding.window_owners(N_KF, world)
    tree = O.KdTree(room["map_xyz"])
    rows = [None] * N_KF
    for k in range(N_KF):
        if owner[k] != rank:
            continue                                                # this rank holds neither the queries nor the records of keyframe k
        Q2, T2 = L.api.assoc_transform(poses[k][0], poses[k][1], P)
        rec = O.associate_surf(tree, None, qs[k], None, Q2, T2, PO)
        G, cost, n = O.linearize_surf(rec, poses[k][0], poses[k][1], PO, (1000.0, max(rec["count"], 1)))      # the owner's count IS the global count
        row = np.zeros(72)
        row[:64], row[64], row[65] = G.reshape(-1), cost, n
        rows[k] = row
    out[rank] = sharding.gather_window_records(dist, rows, owner, rank)
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_slot_per_rank_window_gathers_the_owners_records(oracle, world):
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_gather_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    room, P, poses, qs = _window_scene()
    ref = _window_evaluation(oracle, room, P, oracle.params("rot"), poses, qs, 1, 0, lambda x: x)
    assert sharding.window_owners(N_KF, world) == [k % world for k in range(N_KF)]
    for r in range(world):
        for k in range(N_KF):
            G, cost, cnt = out[r][k]
            assert cnt == ref[k][2] and cnt[0] > 400
            assert np.array_equal(G, ref[k][0]) and cost == ref[k][1]          # a gather: the single-rank evaluation's bits, on every rank

