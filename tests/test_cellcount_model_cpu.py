"""CPU model of the map index build's count pass on the NARROW table (k_cell_count_narrow, lili_om_amd/csrc/lili_s2m.hip; DESIGN.md §3 "K7 build,
round 4") — a wave-by-wave Python restatement of its bookkeeping: four 8-bit counters per 32-bit word, lanes of one WORD RUN sharing one atomic whose
return value carries the four old counts, a lane's rank = the old count of its cell + the lanes of its run before it in the same cell, the overflow
flag for a cell that would pass 255 — and of the row-wise single-pass scan that turns the byte table into `cell_start` (k_scan_lookback_t<true>).
It tests the ARGUMENT, not the GPU (that is tests/test_map_build_gpu.py): whatever order the waves' atomics are served in, the ranks of every
cell are a permutation of 0 .. count-1, the table equals the plain histogram, and the scatter positions `cell_start[cell] + rank` are a
permutation of 0 .. n-1 that keeps every point inside its cell's range.  Replaces kd_tree->setInputCloud, L/src/BackendFusion.cpp:839-840."""
import numpy as np
import pytest


def popc(x):
    return bin(x).count("1")


def wave_pass(cells, table, overflow):
    """One wave (<= 64 lanes, cell -1 = lane past the end) of k_cell_count_narrow against the shared byte table (uint32 words).  Returns the ranks."""
    n = len(cells)
    w = [c >> 2 if c >= 0 else -1 for c in cells]
    sub = [c & 3 for c in cells]
    head = [i == 0 or w[i] != w[i - 1] for i in range(n)]
    ranks = [0] * n
    i = 0
    while i < n:
        j = i + 1
        while j < n and not head[j]:
            j += 1                                        # lanes i .. j-1 are one word run
        if cells[i] >= 0:
            masks = [sum(1 << l for l in range(i, j) if sub[l] == s) for s in range(4)]
            add = sum(popc(masks[s]) << (8 * s) for s in range(4))
            old = int(table[w[i]])
            table[w[i]] = np.uint32((old + add) & 0xffffffff)  # the atomic
            for l in range(i, j):
                before = (old >> (8 * sub[l])) & 0xff
                ranks[l] = before + popc(masks[sub[l]] & ((1 << l) - 1))
                if before + ((add >> (8 * sub[l])) & 0xff) > 255:
                    overflow[0] = True
        i = j
    return ranks


def count_pass(cells, n_cells, rng, wave=64):
    """All waves of the pass in a RANDOM service order (the hardware promises none)."""
    table = np.zeros((n_cells + 3) // 4 + 1, np.uint32)
    ranks = np.zeros(len(cells), np.int64)
    overflow = [False]
    starts = list(range(0, len(cells), wave))
    rng.shuffle(starts)
    for s in starts:
        chunk = list(cells[s:s + wave]) + [-1] * (wave - len(cells[s:s + wave]))
        ranks[s:s + wave] = wave_pass(chunk, table, overflow)[: len(cells[s:s + wave])]
    return table, ranks, overflow[0]


def rowwise_scan(table_bytes, n_cells, tile_rows=16):
    """k_scan_lookback_t<true> of one tile sequence: a wave owns 16 rows of 256 cells, a lane 4 consecutive cells of every row; exclusive prefix."""
    out = np.zeros(n_cells + 1, np.int64)
    carry_tiles = 0
    per_wave = tile_rows * 256
    pos = 0
    while pos < n_cells:
        carry = 0
        for r in range(tile_rows):
            row = table_bytes[pos + 256 * r: pos + 256 * (r + 1)].astype(np.int64)
            row = np.concatenate([row, np.zeros(256 - row.size, np.int64)])
            s4 = row.reshape(64, 4).sum(1)
            inc = np.cumsum(s4)
            ex = carry + inc - s4
            for lane in range(64):
                base = carry_tiles + ex[lane]
                for k in range(4):
                    i = pos + 256 * r + 4 * lane + k
                    if i < n_cells:
                        out[i] = base
                    base += row[4 * lane + k]
            carry += inc[-1]
        carry_tiles += carry
        pos += per_wave
    out[n_cells] = carry_tiles
    return out


@pytest.mark.parametrize("order", ["voxel", "random", "one cell", "duplicates in a wave"])
def test_narrow_count_pass_gives_a_counting_sort(order):
    rng = np.random.default_rng(11)
    n_cells = 700
    if order == "voxel":                                   # long sorted stretches, the same cells visited by several stretches (voxel rows of one cell row)
        cells = np.concatenate([np.sort(rng.integers(0, n_cells, 300)) for _ in range(6)])
    elif order == "random":
        cells = rng.integers(0, n_cells, 1800)
    elif order == "one cell":
        cells = np.full(200, 37)
    else:                                                  # a cell appearing in two separate runs of one word run (c, c+1, c): the second run's ranks continue the first's
        cells = np.tile(np.array([8, 9, 8, 8, 10, 9, 11, 8]), 30)
    table, ranks, overflow = count_pass(list(cells), n_cells, rng)
    assert not overflow
    counts = np.bincount(cells, minlength=n_cells)
    bytes_ = table.view(np.uint8)[:n_cells]
    assert np.array_equal(bytes_, counts)                                   # the byte table is the histogram
    for c in np.unique(cells):
        assert sorted(ranks[cells == c]) == list(range(counts[c])), c       # ranks of a cell: 0 .. count-1, each once
    start = rowwise_scan(table.view(np.uint8), n_cells)
    assert np.array_equal(start[:-1], np.concatenate([[0], np.cumsum(counts)[:-1]])) and start[-1] == len(cells)
    pos = start[cells] + ranks
    assert sorted(pos) == list(range(len(cells)))                           # the scatter is a permutation ...
    assert np.all(pos >= start[cells]) and np.all(pos < start[cells + 1])   # ... that keeps every point inside its cell's range


def test_narrow_count_pass_flags_a_cell_beyond_255_points():
    rng = np.random.default_rng(12)
    cells = np.concatenate([np.full(255, 5), rng.integers(100, 200, 300)])
    rng.shuffle(cells)
    _, _, overflow = count_pass(list(cells), 256, rng)
    assert not overflow                                                     # 255 points in a cell still fit
    cells = np.concatenate([cells, [5]])
    _, _, overflow = count_pass(list(cells), 256, rng)
    assert overflow                                                         # the 256th raises the flag: lili_map_set repeats the build with 32-bit counters
