"""Host-side sharding algebra of the multi-GPU scan-to-map iteration.

The data path is: every rank associates + linearises its block of the query array against the replicated map,
then ONE all-reduce(sum) of the 72-double Gram record (and, for the ROT residual scale num/N, one of the two
correspondence counters) makes every rank hold the same normal equations; each rank applies the same
Gauss-Newton update.  This module holds the pieces that do not depend on the device so that they can be tested
with gloo on CPU; bench.py drives the same functions with the HIP kernels in between.
"""
import numpy as np


def shard_bounds(n_items, world, rank):
    """Contiguous block shard [lo, hi) of rank `rank` (rank r gets [r*ceil(N/P), ...))."""
    per = (n_items + world - 1) // world
    lo = min(n_items, rank * per)
    return lo, min(n_items, lo + per)


def allreduce_counts(dist, counts):
    """In-place sum of the [n_surf, n_edge] tensor over ranks (ROT: scale = num / GLOBAL N)."""
    dist.all_reduce(counts)
    return counts


def allreduce_gram(dist, gram):
    """In-place sum of the 72-double record [64 Gram | cost | n_surf | n_edge | ...] over ranks."""
    dist.all_reduce(gram)
    return gram


def pack_window_records(records):
    """The Gram records of the K keyframes of a sliding window as ONE buffer of K x 72 doubles (BASELINE configs[4]: the reference evaluates
    the lidar blocks of all keyframes per solver evaluation, L/src/BackendFusion.cpp:919-992) — one all-reduce per evaluation instead of K."""
    return np.ascontiguousarray(np.concatenate([np.asarray(r, np.float64).reshape(72) for r in records]))


def unpack_window_records(buf, k):
    buf = np.asarray(buf, np.float64).reshape(k, 72)
    return [(buf[i, :64].reshape(8, 8).copy(), float(buf[i, 64]), (int(buf[i, 65]), int(buf[i, 66]))) for i in range(k)]


def allreduce_window(dist, tensor):
    """In-place sum over ranks of the packed window buffer (K x 72 doubles) or of the packed counts (K x 2 int32): ONE collective each."""
    dist.all_reduce(tensor)
    return tensor


def window_owners(n_slots, world):
    """Slot-per-rank window (lili_s2m_*_window_gather): keyframe i of the window lives on rank i mod world at full size."""
    return [i % world for i in range(n_slots)]


def gather_window_records(dist, records, owner, rank):
    """The K x 72 records of one evaluation in slot-per-rank mode: this rank contributes the records of the slots it owns (`records[i]` for owner[i] == rank, anything
    else is ignored) and zeros elsewhere; ONE all-reduce(sum) in which every record has a single non-zero contributor, i.e. an all-gather — the owner's bits on
    every rank (a -0.0 entry reads +0.0).  Returns the unpacked list."""
    import torch
    buf = np.zeros((len(owner), 72))
    for i, o in enumerate(owner):
        if o == rank:
            buf[i] = np.asarray(records[i], np.float64).reshape(72)
    t = torch.from_numpy(buf.reshape(-1))
    dist.all_reduce(t)
    return unpack_window_records(t.numpy(), len(owner))


def plus_jacobian(q):
    """ceres::QuaternionParameterization::ComputeJacobian for q = (w,x,y,z): 4x3."""
    w, x, y, z = q
    return np.array([[-x, -y, -z], [w, z, -y], [-z, w, x], [y, -x, w]], dtype=np.float64)


def project_gram(gram8, q):
    """(H 6x6, g 6) of the local parameterisation from the global 8x8 Gram: H = P^T G77 P, g = P^T G7r."""
    G = np.asarray(gram8, np.float64).reshape(8, 8)
    P = np.zeros((7, 6))
    P[:3, :3] = np.eye(3)
    P[3:, 3:] = plus_jacobian(q)
    return P.T @ G[:7, :7] @ P, P.T @ G[:7, 7]
