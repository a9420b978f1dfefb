"""CPU model of the association kernel's fast selection tier (Sel5K, lili_om_amd/csrc/lili_s2m.hip; DESIGN.md §4 "Two-tier
selection") — a line-by-line Python restatement of its bookkeeping (32-bit keys = distance bucket | 6-bit code, six held keys,
chunk table with re-pointing every eight chunks, resolution and exact re-ordering of the five winners, redo flag) checked
against brute force on adversarial inputs.  It tests the EXACTNESS ARGUMENT, not the GPU: whenever the model does not ask for
the exact selector, its five neighbours are the five smallest (distance, index) pairs within the bound; it asks rarely on
generic data and always on data where buckets collide around the fifth neighbour."""
import numpy as np
import pytest


def _bits(x):
    return int(np.float32(x).view(np.uint32))


class Sel5KModel:
    def __init__(self, bound):
        b = min(_bits(bound), _bits(np.float32(3.0e38)))
        self.bnd = np.uint32(b).view(np.float32)
        self.bb = b >> 6
        self.k = [(((self.bb + 1 + s) << 6) | 63) & 0xffffffff for s in range(6)]
        self.T = [0] * 16
        self.T[15] = -4
        self.tc = 0

    def push(self, key):                       # five med3 + one min == keep the six smallest keys sorted
        self.k = sorted(self.k + [key])[:6]

    def where(self, key):
        c = key & 63
        return self.T[c >> 2] + (c & 3)

    def repoint(self):
        jr = [self.where(k) for k in self.k]
        for s in range(6):
            self.T[8 + s] = jr[s]
            self.k[s] = (self.k[s] & ~63 & 0xffffffff) | (32 + 4 * s)

    def chunk(self, d2, j, end):
        """d2: distances of the candidates at array positions j .. j+3 (positions >= end are padding)."""
        if self.tc >= 8 and (self.tc & 7) == 0:
            self.repoint()
        self.T[self.tc & 7] = j
        code = (self.tc & 7) << 2
        self.tc += 1
        for s in range(4):
            u = _bits(d2[s]) if j + s < end else 0x7f800000
            self.push(((u & ~63) & 0xffffffff) | (code + s))

    def finish(self, dist_of, idx_of):
        redo = ((self.k[5] ^ self.k[4]) < 64) or ((self.k[4] >> 6) == self.bb)
        e = []
        for s in range(5):
            j = self.where(self.k[s])
            if j >= 0:
                e.append(((_bits(dist_of(j)) << 32) | idx_of(j), j))
            else:
                e.append(((_bits(self.bnd) << 32) | 0x7fffffff, -1))
        e.sort(key=lambda t: t[0])             # the kernel sorts only when some lane has an inversion; same result
        return redo, [np.uint32(k >> 32).view(np.float32) for k, _ in e], [j for _, j in e]


def _run(d2, idx, bound, rng):
    """Candidates arrive as runs of random lengths (rows), processed in chunks of four like the kernel's walk."""
    n = len(d2)
    sel = Sel5KModel(bound)
    pos = 0
    while pos < n:
        end = min(n, pos + int(rng.integers(1, 14)))
        for j in range(pos, end, 4):
            sel.chunk([d2[min(j + s, end - 1)] for s in range(4)], j, end)
        pos = end
    redo, dd, jj = sel.finish(lambda j: d2[j], lambda j: int(idx[j]))
    # brute force: the five smallest (distance, original index) among candidates with d <= bound; padded with (bound, -1)
    cand = sorted((( _bits(d2[j]), int(idx[j])), j) for j in range(n) if d2[j] <= np.float32(min(bound, 3.0e38)))
    want = [j for _, j in cand[:5]] + [-1] * max(0, 5 - len(cand))
    return redo, jj, want, dd


@pytest.mark.parametrize("seed", range(6))
def test_generic_data_is_exact_and_rarely_redone(seed):
    rng = np.random.default_rng(seed)
    redos = 0
    for trial in range(300):
        n = int(rng.integers(0, 90))
        d2 = rng.uniform(0.0, 2.5, n).astype(np.float32)
        idx = rng.permutation(10_000)[:n]
        bound = np.float32(1.0) if trial % 3 else np.float32(3.0e38)
        redo, got, want, dd = _run(d2, idx, bound, rng)
        if redo:
            redos += 1
            continue
        assert got == want, (trial, got, want)
        assert all(dd[s] <= dd[s + 1] for s in range(4))
    assert redos <= 3                                   # bucket = 64 ulps: collisions around the 5th/6th are rare on generic floats


@pytest.mark.parametrize("seed", range(4))
def test_colliding_buckets_are_flagged_never_wrong(seed):
    """Distances drawn from a handful of values a few ulps apart (lattice-like maps): either the model asks for the exact
    selector, or its answer is exactly the brute-force one — never a silent mistake."""
    rng = np.random.default_rng(100 + seed)
    flagged = 0
    for trial in range(400):
        n = int(rng.integers(5, 70))
        base = np.float32(rng.uniform(0.05, 0.9))
        pool = np.array([np.nextafter(base, np.float32(2), dtype=np.float32)] * 3 + [base] * 3 +
                        [np.float32(base * (1 + 1e-6 * k)) for k in range(4)] + [np.float32(rng.uniform(0, 2)) for _ in range(6)], np.float32)
        d2 = rng.choice(pool, n).astype(np.float32)
        idx = rng.permutation(5_000)[:n]
        redo, got, want, dd = _run(d2, idx, np.float32(1.0), rng)
        if redo:
            flagged += 1
        else:
            assert got == want, (trial, got, want)
    assert flagged > 50                                  # the adversarial pool really exercises the flag


def test_bound_bucket_and_long_walks():
    rng = np.random.default_rng(7)
    # a candidate in the bucket of the bound as 5th best must be flagged (it may lie beyond the bound)
    bound = np.float32(1.0)
    d2 = np.array([0.1, 0.2, 0.3, 0.4, np.nextafter(bound, np.float32(2), dtype=np.float32), 1.7], np.float32)
    redo, got, want, _ = _run(d2, np.arange(6), bound, rng)
    assert redo
    # more than 8 chunks: the re-pointed slots keep resolving to the right array positions
    for trial in range(50):
        n = int(rng.integers(120, 400))
        d2 = rng.uniform(0.0, 3.0, n).astype(np.float32)
        idx = rng.permutation(100_000)[:n]
        redo, got, want, _ = _run(d2, idx, np.float32(3.0e38), rng)
        assert redo or got == want
    # fewer than five candidates inside the bound: sentinels fill the tail
    d2 = np.array([0.5, 3.0, 0.25, 7.0], np.float32)
    redo, got, want, dd = _run(d2, np.arange(4), bound, rng)
    assert not redo and got == want == [2, 0, -1, -1, -1] and float(dd[4]) == 1.0


def test_candidate_just_beyond_the_bound_with_fewer_than_five_neighbours():
    """Documented corner: a candidate in the bucket of the bound but beyond it is not filtered by the fast tier.  With fewer than
    five neighbours inside the bound it can appear in the list, but the fifth entry is then a sentinel at the bound, so the
    reference's gate `d2[4] < gate` (bound = smallest f32 >= gate) rejects the query exactly as it does for the exact tier."""
    rng = np.random.default_rng(3)
    bound = np.float32(1.0)
    beyond = np.nextafter(bound, np.float32(2), dtype=np.float32)
    d2 = np.array([0.2, beyond, 0.4, 0.6], np.float32)
    redo, got, want, dd = _run(d2, np.arange(4), bound, rng)
    assert not redo and got[:3] == want[:3] == [0, 2, 3]
    assert -1 in got[3:] and float(dd[4]) >= float(bound)         # a sentinel at the bound is among the last two and the fifth distance is >= the bound:
                                                                     # the query fails the gate on both tiers
