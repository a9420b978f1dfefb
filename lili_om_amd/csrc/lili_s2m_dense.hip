// Association on a map that is much DENSER than the gate radius (SURVEY §8d Config 2, variant B: 5 M points at a 0.05 m leaf — ~170 points per
// gate-sized cell, ~1 500 candidates in the inner 27 gate-sized cells).  lili_map_set gives such a map an index with cells sized from the measured
// point density (~3 points per cell, super-row layout) and k_associate_fine searches that one index in two steps inside the same wave:
//
//   step 1   one query per lane: the inner 27 fine cells (ONE contiguous run) with the payload-pool key selector Sel5P.  A query whose fifth neighbour
//            lies inside the radius its inner block covers completely (one cell + the gap to the nearest face of its own cell) is SETTLED — these five
//            are the global 5-NN — and is fitted, gated and stored right there.
//   step 2   the lanes that are not settled (too few points nearby: far from converged, map border, hole; a bucket tie of the key selector) are served
//            by the WHOLE wave, kFarL lanes per query, 64 / kFarL queries per round: rings of super-rows around the query's own — level m covers the
//            cells within 3 m + 1 of the query's cell with (2 m + 1)^2 contiguous runs — until the fifth neighbour lies inside the radius the level
//            covers completely or the level covers the reference's gate (`pointSearchSqDis[4] < gate`, L/src/BackendFusion.cpp:1615); exact
//            (distance, original index) selector, group reductions by DPP as in lili_s2m_coop.hip.
//
// Round 5: a lane the fine index could not settle repeated its search ALONE on the gate-sized index while the other 63 lanes of its wave waited (~1 500
// to 4 400 candidates, 25 KB of map per query): 46 us per launch when every query is settled, 515-590 us for the first launch of a registration from
// 0.1 m / 0.5 deg off, where 35 % of the queries (and a lane of every wave) are not (profiles/r06_2B_*).  Both searches are exact k-NN searches, so the
// records are the oracle's bit for bit either way (tests/test_dense_map_gpu.py).  The gate-sized index of a dense map is no longer searched.
//
// Reference behaviour replaced: findCorrespondingSurfFeatures / findCorrespondingCornerFeatures (L/src/BackendFusion.cpp:1531-1681,
// R/src/BackendFusion.cpp:1394-1520) on a pcl::KdTreeFLANN of a finely voxelised map (leaf sizes L/src/BackendFusion.cpp:1488-1511).
#include "lili_s2m_dev.h"

namespace lili {

// group reductions by DPP (as lili_s2m_coop.hip; the patterns stay inside a row of 16 lanes)
template <int CTRL> __device__ __forceinline__ unsigned dense_dpp_u32(unsigned v) {
    return (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, 0xf, 0xf, false);
}
template <int CTRL> __device__ __forceinline__ void dense_kmin_step(unsigned& lo, unsigned& hi, int& j) {
    const unsigned olo = dense_dpp_u32<CTRL>(lo), ohi = dense_dpp_u32<CTRL>(hi);
    const int oj = (int)dense_dpp_u32<CTRL>((unsigned)j);
    const unsigned long long a = ((unsigned long long)hi << 32) | lo, b = ((unsigned long long)ohi << 32) | olo;
    const bool take = b < a;
    lo = take ? olo : lo; hi = take ? ohi : hi; j = take ? oj : j;
}
struct PoolTab {      // Sel5P's pool: six slots per lane
    float4 p[6][kAssocBlock];
    int j[6][kAssocBlock];
};

__device__ __forceinline__ void store_record(const AssocArgs& A, int kind, int i, bool ok, const float4& r0, const float4& rb, double score) {
    A.rec0[i] = r0;
    if (kind == 0) reinterpret_cast<double*>(A.rec1)[i] = score; else reinterpret_cast<float4*>(A.rec1)[i] = rb;
    A.valid[i] = ok ? 1 : 0;
}

// ------------------------------------------------------------------------------------------------------------------------------------------------
// step 2: L lanes per query, rings of super-rows, one LEVEL per round
// ------------------------------------------------------------------------------------------------------------------------------------------------
constexpr int kFarL = 16;                          // lanes per query
constexpr int kFarG = kAssocBlock / kFarL;         // queries per round
constexpr int kFarU = 8;                           // candidates per lane and trip
constexpr int kFarRuns = 2 * kFarL;                // runs a group can publish per batch of blocks (two x segments per lane)

constexpr int kSerialMin = 16;                     // unsettled lanes of a wave from which level 1 is walked by the lanes themselves (see k_associate_fine)
union FarTab {
    struct { int run_b[kFarG][kFarRuns], run_n[kFarG][kFarRuns]; } grp;   // cooperative rounds: the group's non-empty runs of the batch: first position, length
    struct { int b[9][kAssocBlock], n[9][kAssocBlock]; } ser;            // lane-serial level 1: the lane's nine runs (first position, length)
};
__device__ __forceinline__ unsigned long long shfl_u64(unsigned long long v, int src) {
    return ((unsigned long long)(unsigned)__shfl((int)(unsigned)(v >> 32), src) << 32) | (unsigned)__shfl((int)(unsigned)v, src);
}

template <int L> __device__ __forceinline__ void group_kmin_d(unsigned& lo, unsigned& hi, int& j) {
    dense_kmin_step<0xB1>(lo, hi, j); dense_kmin_step<0x4E>(lo, hi, j); dense_kmin_step<0x141>(lo, hi, j);
    if constexpr (L >= 16) dense_kmin_step<0x140>(lo, hi, j);
}
template <int CTRL> __device__ __forceinline__ int dense_dpp_add(int v) { return v + __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, true); }
template <int L> __device__ __forceinline__ int group_sum_d(int v) {      // the sum over the lanes of a group, on every lane (same butterflies as the minimum)
    v = dense_dpp_add<0xB1>(v); v = dense_dpp_add<0x4E>(v); v = dense_dpp_add<0x141>(v);
    if constexpr (L >= 16) v = dense_dpp_add<0x140>(v);
    return v;
}

// One level of the exact 5-NN search of one query by the L lanes of its group (all 64 lanes of the wave call this; `live`, the query and K / J are
// uniform per group).  Level m = 1, 2, 3 ...: the blocks of 3 x 3 rows (a, b), |a|, |b| <= m, around the query's row — a block whose centre row has
// super-rows is ONE run per x segment, any other block (grid border, outside the focus box) is walked row by row on the base index by the lane that
// owns it — over the cells cx - R .. cx + R, R = 3 m + 1; a block the previous level has seen only adds its two side segments (level 1 sees all of its
// nine blocks whole).  Nothing is seen twice.  K / J: in = the five best of the levels before (or the bound, -1), out = the five best including this
// level — exact within the radius R * cell + (gap of the query to the nearest face of its own cell).  Returns true if the search is over: the fifth
// lies inside that radius (0.1 % conservative) or R >= r_max.
//   EXACT = false: every lane selects among its candidates with Sel5P (32-bit bucket keys, payload pool); a bucket tie in any lane makes the group
//                  repeat the level with EXACT = true (the caller does): 64-bit (distance, index) keys throughout.
template <int L, bool EXACT>
__device__ __forceinline__ bool far_level(const GridView& g, int m, bool live, int sub, int grp, float qx, float qy, float qz, int r_max,
                                          unsigned long long K[5], int J[5], FarTab& tab, PoolTab& pool, bool& tie) {
    tie = false;
    const bool in = live;
    const int cx = in ? cell_coord(qx, g.ox, g.inv_cell) : 0, cy = in ? cell_coord(qy, g.oy, g.inv_cell) : 0, cz = in ? cell_coord(qz, g.oz, g.inv_cell) : 0;
    const double c = g.cell;
    const double fxm = (double)qx - (g.ox + (double)cx * c), fxp = (g.ox + (double)(cx + 1) * c) - (double)qx;
    const double fym = (double)qy - (g.oy + (double)cy * c), fyp = (g.oy + (double)(cy + 1) * c) - (double)qy;
    const double fzm = (double)qz - (g.oz + (double)cz * c), fzp = (g.oz + (double)(cz + 1) * c) - (double)qz;
    const double gap0 = fmax(fmin(fmin(fmin(fxm, fxp), fmin(fym, fyp)), fmin(fzm, fzp)), 0.0);
    const int grp0 = grp * L;                                        // first lane of this group within the wave
    const unsigned gbits = (L >= 32) ? 0xffffffffu : ((1u << L) - 1u);
    const float W = __uint_as_float((unsigned)(K[4] >> 32));         // nothing beyond the fifth best so far (or the gate) can matter; ties at W enter and are ordered by the full key
    const int R = 3 * m + 1, Rp = R - 3, side = 2 * m + 1, nblk = side * side;
    // a query more than r_max cells outside the grid has no neighbour within the gate
    const bool go = in && !(cx < -r_max || cx > g.nx - 1 + r_max || cy < -r_max || cy > g.ny - 1 + r_max || cz < -r_max || cz > g.nz - 1 + r_max);
    Sel5 se_; Sel5P sp_;
    if constexpr (EXACT) se_.init(W); else sp_.init(W, &pool.p[0][threadIdx.x], &pool.j[0][threadIdx.x]);
    auto consider = [&](float4 p, int at, bool valid) {
        asm volatile("" : "+v"(p.w));
        const unsigned du = valid ? __float_as_uint(dist2(p, qx, qy, qz)) : 0x7f800000u;
        if constexpr (EXACT) { if (du <= se_.worst_bits()) se_.insert(__uint_as_float(du), p, at); }
        else sp_.push(du, p, at);
    };
    if (go) {
        for (int t0 = 0; t0 < nblk; t0 += L) {
            // every lane prepares ONE block: its x segments as runs of the unified array, with their lower distance bounds
            const int t = t0 + sub;
            const int a = t / side - m, b = t % side - m;
            const int Y = cy + 3 * a, Z = cz + 3 * b;
            const bool blk = t < nblk && Y + 1 >= 0 && Y - 1 < g.ny && Z + 1 >= 0 && Z - 1 < g.nz;
            const bool fresh = m == 1 || max(abs(a), abs(b)) == m;                       // not seen by the previous level
            const double gy = a == 0 ? 0.0 : fmax(a < 0 ? fym + (double)(-3 * a - 2) * c : fyp + (double)(3 * a - 2) * c, 0.0);
            const double gz = b == 0 ? 0.0 : fmax(b < 0 ? fzm + (double)(-3 * b - 2) * c : fzp + (double)(3 * b - 2) * c, 0.0);
            const double lbr = 0.999 * (gy * gy + gz * gz);
            // segment 0: the whole range (fresh) or the left side; segment 1: the right side of a block seen before
            int xs[2], xe[2]; float lb[2];
            if (fresh) { xs[0] = max(cx - R, 0); xe[0] = min(cx + R, g.nx - 1); lb[0] = (float)lbr; xs[1] = 0; xe[1] = -1; lb[1] = 0.f; }
            else {
                const double gl = fmax(fxm + (double)Rp * c, 0.0), gr = fmax(fxp + (double)Rp * c, 0.0);
                xs[0] = max(cx - R, 0); xe[0] = min(cx - Rp - 1, g.nx - 1); lb[0] = (float)(lbr + 0.999 * gl * gl);
                xs[1] = max(cx + Rp + 1, 0); xe[1] = min(cx + R, g.nx - 1); lb[1] = (float)(lbr + 0.999 * gr * gr);
            }
            const bool has9 = blk && g.cell_start9 && Y >= g.by0 && Y < g.by0 + g.bny && Z >= g.bz0 && Z < g.bz0 + g.bnz;
            int sb[2] = {0, 0}, sn[2] = {0, 0};
            bool slow[2] = {false, false};
#pragma unroll
            for (int s = 0; s < 2; s++) {
                if (!blk || xs[s] > xe[s] || lb[s] > W) continue;      // a run whose box lies beyond the fifth best so far is dropped
                if (has9 && xs[s] >= g.bx0 && xe[s] < g.bx0 + g.bnx) {
                    const int* row = g.cell_start9 + srow_index(g, g.bx0, Y, Z) - g.bx0;
                    sb[s] = row[xs[s]]; sn[s] = row[xe[s] + 1] - sb[s];
                } else slow[s] = true;
            }
            // The group's non-empty runs, compacted into its list in LDS (segment 0 of the lanes first, then segment 1); the candidates of the list are ONE
            // sequence of N, and lane l takes the contiguous piece [l q, (l + 1) q), q = ceil(N / L): eight consecutive positions per trip, all requested
            // together (run by run with two loads per lane in flight a round was a chain of 9 to 25 dependent memory round trips).
            const unsigned b0 = (unsigned)(__ballot(sn[0] > 0) >> grp0) & gbits, b1 = (unsigned)(__ballot(sn[1] > 0) >> grp0) & gbits;
            const int S = __popc(b0) + __popc(b1);
            if (sn[0] > 0) { const int k = __popc(b0 & ((1u << sub) - 1u)); tab.grp.run_b[grp][k] = sb[0]; tab.grp.run_n[grp][k] = sn[0]; }
            if (sn[1] > 0) { const int k = __popc(b0) + __popc(b1 & ((1u << sub) - 1u)); tab.grp.run_b[grp][k] = sb[1]; tab.grp.run_n[grp][k] = sn[1]; }
            const int N = group_sum_d<L>(sn[0] + sn[1]);
            __builtin_amdgcn_wave_barrier();      // (one wave per workgroup: LDS writes of the wave are in order with its reads; this only pins the compiler)
            if (N > 0) {
                const int q = (N + L - 1) / L, n_l = sub * q;
                int k = 0, pos = 0, endk = 0, acc = 0;
                for (int kk = 0; kk < S; kk++) {      // the run that holds this lane's first candidate
                    const int len = tab.grp.run_n[grp][kk];
                    if (n_l >= acc && n_l < acc + len) { k = kk; pos = tab.grp.run_b[grp][kk] + (n_l - acc); endk = tab.grp.run_b[grp][kk] + len; }
                    acc += len;
                }
                const int rem = max(min(q, N - n_l), 0);
                for (int c0 = 0; c0 < q; c0 += kFarU) {
                    int at[kFarU];
#pragma unroll
                    for (int u = 0; u < kFarU; u++) {
                        const bool v = c0 + u < rem;
                        at[u] = v ? pos : -1;
                        if (v) {
                            pos++;
                            if (pos == endk && k + 1 < S) { k++; pos = tab.grp.run_b[grp][k]; endk = pos + tab.grp.run_n[grp][k]; }
                        }
                    }
                    float4 pt[kFarU];
#pragma unroll
                    for (int u = 0; u < kFarU; u++) pt[u] = load_pt(g, max(at[u], 0));
#pragma unroll
                    for (int u = 0; u < kFarU; u++) consider(pt[u], at[u], at[u] >= 0);
                }
            }
            __builtin_amdgcn_wave_barrier();
            // rows of the base index, by this lane alone (a block at the grid border or outside the focus box)
#pragma unroll
            for (int s = 0; s < 2; s++) {
                if (!slow[s]) continue;
                for (int z = max(Z - 1, 0); z <= min(Z + 1, g.nz - 1); z++)
                    for (int y = max(Y - 1, 0); y <= min(Y + 1, g.ny - 1); y++) {
                        const int* cs = g.cell_start + (size_t)(z * g.ny + y) * g.nx;
                        const int rb = cs[xs[s]], re = cs[xe[s] + 1];
                        for (int j = rb; j < re; j++) consider(load_pt(g, j), j, true);
                    }
            }
        }
    }
    // this lane's five best of the level as exact keys, then the group's five best of [levels before | this level]
    Sel5 mine; mine.init(W);
    if constexpr (EXACT) mine = se_;
    else {
        Top5 t;
        const bool redo = sp_.finish(qx, qy, qz, t);
        tie = go && redo;
#pragma unroll
        for (int r = 0; r < 5; r++) {
            mine.k[r] = t.j[r] >= 0 ? (((unsigned long long)__float_as_uint(t.d[r]) << 32) | (unsigned)__float_as_int(t.p[r].w)) : mine.k[r];
            mine.j[r] = t.j[r];
        }
    }
    tie = __any(tie) && (((__ballot(tie) >> grp0) & gbits) != 0ull);      // uniform within the group
    unsigned long long k0 = mine.k[0], k1 = mine.k[1], k2 = mine.k[2], k3 = mine.k[3], k4 = mine.k[4];
    int j0 = mine.j[0], j1 = mine.j[1], j2 = mine.j[2], j3 = mine.j[3], j4 = mine.j[4];
    unsigned long long i0 = K[0], i1 = K[1], i2 = K[2], i3 = K[3], i4 = K[4];      // the list of the levels before (uniform within the group): one more sorted list
    int ij0 = J[0], ij1 = J[1], ij2 = J[2], ij3 = J[3], ij4 = J[4];
#pragma unroll
    for (int r = 0; r < 5; r++) {
        unsigned lo = (unsigned)k0, hi = (unsigned)(k0 >> 32);
        int jj = j0;
        group_kmin_d<L>(lo, hi, jj);
        const unsigned long long mk = ((unsigned long long)hi << 32) | lo;
        const bool ipop = i0 < mk;                  // the list of the levels before holds the smaller key (real keys are unique: the levels see disjoint cells)
        const bool pop = !ipop && k0 == mk;
        K[r] = ipop ? i0 : mk; J[r] = ipop ? ij0 : jj;
        k0 = pop ? k1 : k0; k1 = pop ? k2 : k1; k2 = pop ? k3 : k2; k3 = pop ? k4 : k3; k4 = pop ? ~0ull : k4;
        j0 = pop ? j1 : j0; j1 = pop ? j2 : j1; j2 = pop ? j3 : j2; j3 = pop ? j4 : j3; j4 = pop ? -1 : j4;
        i0 = ipop ? i1 : i0; i1 = ipop ? i2 : i1; i2 = ipop ? i3 : i2; i3 = ipop ? i4 : i3; i4 = ipop ? ~0ull : i4;
        ij0 = ipop ? ij1 : ij0; ij1 = ipop ? ij2 : ij1; ij2 = ipop ? ij3 : ij2; ij3 = ipop ? ij4 : ij3; ij4 = ipop ? -1 : ij4;
    }
    const float Wn = __uint_as_float((unsigned)(K[4] >> 32));
    const double margin = (double)R * c + gap0;
    return !go || Wn < (float)(0.999 * margin * margin) || R >= r_max;
}

// `kind`: 0 surf, 1 edge.  A.g = the fine index (whole-grid or focused super-rows), A.block_counts[b] = correspondences of workgroup b.
// r_max: fine cells step 2 has to reach for the gate ball.  qpw: queries per workgroup (wave) — 64, or 16 for a launch too small to give every SIMD a wave: the
// rounds of step 2 are a serial chain per wave, and a small launch is as long as its longest chain.
__global__ __launch_bounds__(kAssocBlock) void k_associate_fine(AssocArgs A, int r_max, int qpw, int kind, PoseArg pa, MatchParams P) {
    __shared__ PoolTab pool;
    __shared__ FarTab tab;
    const GridView& g = A.g;
    const int lane = (int)threadIdx.x;
    const int i = (int)blockIdx.x * qpw + lane;
    const bool live = lane < qpw && i < A.n_q;
    const float4 ql = A.queries[live ? i : 0];
    dq Q2; d3 T2;
    load_assoc_pose(pa, P, Q2, T2);
    const d3 pmd = qrot(Q2, d3{(double)ql.x, (double)ql.y, (double)ql.z}) + T2;      // transformPoint, L:695-711
    const float px = (float)pmd.x, py = (float)pmd.y, pz = (float)pmd.z;
    const float gate = gate_bound(kind == 0 ? P.kd_max_radius : P.edge_gate);
    const bool finite = isfinite(px) && isfinite(py) && isfinite(pz);
    const int cx = finite ? cell_coord(px, g.ox, g.inv_cell) : 0, cy = finite ? cell_coord(py, g.oy, g.inv_cell) : 0, cz = finite ? cell_coord(pz, g.oz, g.inv_cell) : 0;
    // the query's own super-row exists and its inner three cells lie in the box that has super-rows
    const bool inner9 = live && finite && g.cell_start9 && cx >= 0 && cx < g.nx && cy >= g.by0 && cy < g.by0 + g.bny && cz >= g.bz0 && cz < g.bz0 + g.bnz &&
                        max(cx - 1, 0) >= g.bx0 && min(cx + 1, g.nx - 1) < g.bx0 + g.bnx;
    Top5 nn; nn.aux = 0; nn.have = false;
#pragma unroll
    for (int k = 0; k < 5; k++) { nn.j[k] = -1; nn.d[k] = gate; }      // a non-finite query has no neighbours (its distances are NaN): an invalid record, as knn5_grid_sel leaves it
    bool settled = false;
    const bool exact_only = (P.debug & 32768) != 0;      // LILI_DEBUG bit 32768, the tier tests: every query through the exact selector of step 2
    if (inner9 && !exact_only) {
        const double c = g.cell;
        const double fxm = (double)px - (g.ox + (double)cx * c), fxp = (g.ox + (double)(cx + 1) * c) - (double)px;
        const double fym = (double)py - (g.oy + (double)cy * c), fyp = (g.oy + (double)(cy + 1) * c) - (double)py;
        const double fzm = (double)pz - (g.oz + (double)cz * c), fzp = (g.oz + (double)(cz + 1) * c) - (double)pz;
        const double margin = c + fmax(fmin(fmin(fmin(fxm, fxp), fmin(fym, fyp)), fmin(fzm, fzp)), 0.0);
        const float covered = fminf((float)(0.999 * margin * margin), gate);      // every point outside the inner block is at least `margin` away (0.1 % conservative, as knn5_grid_sel)
        Sel5P sel; sel.init(covered, &pool.p[0][lane], &pool.j[0][lane]);
        const int* row = g.cell_start9 + srow_index(g, g.bx0, cy, cz) - g.bx0;
        int cj = row[max(cx - 1, 0)];
        const int ce = row[min(cx + 1, g.nx - 1) + 1];
        auto fetch = [&](float4& p0, float4& p1, float4& p2, float4& p3, int& pj) {
            // unconditional loads (exact waits); a chunk requested past the run's end is never processed — it re-reads the run's last point instead of the line behind it
            pj = cj;
            const float4* q = (const float4*)((const char*)g.pts + ((unsigned)min(cj, max(ce - 1, 0)) << 4));
            p0 = q[0]; p1 = q[1]; p2 = q[2]; p3 = q[3];
            cj += 4;
        };
        float4 a0, a1, a2, a3, b0, b1, b2, b3;
        int aj = 0, bj = 0;
        fetch(a0, a1, a2, a3, aj);
        for (;;) {
            if (!(aj < ce)) break;
            fetch(b0, b1, b2, b3, bj);
            asm volatile("" : "+v"(a0.w), "+v"(a1.w), "+v"(a2.w), "+v"(a3.w));
            sel.chunk(a0, a1, a2, a3, aj, ce, px, py, pz);
            if (!(bj < ce)) break;
            fetch(a0, a1, a2, a3, aj);
            asm volatile("" : "+v"(b0.w), "+v"(b1.w), "+v"(b2.w), "+v"(b3.w));
            sel.chunk(b0, b1, b2, b3, bj, ce, px, py, pz);
        }
        Top5 t1;
        const bool redo = sel.finish(px, py, pz, t1);
        // settled: five candidates strictly inside the covered radius (which is the bound of the selection: a fifth best in the bound's bucket reports `redo`), no bucket tie
        settled = !redo && t1.j[4] >= 0 && t1.d[4] < covered;
        if (settled) nn = t1;
    }
    const bool far = live && finite && !settled;
    // step 2: the wave serves its unsettled lanes level by level, kFarG queries per round; group `grp` takes the grp-th raised bit of the round and gets the query from
    // the lane that owns it; the five best of a query wait in LDS between its levels
    unsigned long long mask = __ballot(far);
    if (mask) {
        const int sub = lane & (kFarL - 1), grp = lane / kFarL;
        unsigned long long oK[5]; int oJ[5];      // this lane's own query: the five best so far as exact (distance, original index) keys and their positions
#pragma unroll
        for (int r = 0; r < 5; r++) { oK[r] = ((unsigned long long)__float_as_uint(gate) << 32) | 0x7fffffffull; oJ[r] = -1; }
        int lvl = 1;                               // the next level this lane's query needs
        // Many unsettled lanes (the first association of a registration: two thirds of the wave): level 1 is walked BY THE LANES THEMSELVES — nine runs per lane, one
        // after the other with the next chunk in flight, Sel5P — and only what level 1 leaves open (0.7 % of the queries) goes through the cooperative rounds.  A
        // cooperative round costs ~1 500 wave-instructions for four queries whatever they need (block geometry, compaction, merge): eleven rounds per wave made the first
        // launch instruction-bound (51.9 M wave-instructions, 226 us); the lane-serial walk is ~6 000 for up to 64 queries.  Few unsettled lanes (settled launches: one lane
        // in one wave of eighteen): the cooperative round is the shorter chain.
        if (__popcll(mask) >= kSerialMin && !exact_only) {
            bool walk = false;
            const double c = g.cell;
            const double fxm = (double)px - (g.ox + (double)cx * c), fxp = (g.ox + (double)(cx + 1) * c) - (double)px;
            const double fym = (double)py - (g.oy + (double)cy * c), fyp = (g.oy + (double)(cy + 1) * c) - (double)py;
            const double fzm = (double)pz - (g.oz + (double)cz * c), fzp = (g.oz + (double)(cz + 1) * c) - (double)pz;
            const double gap0 = fmax(fmin(fmin(fmin(fxm, fxp), fmin(fym, fyp)), fmin(fzm, fzp)), 0.0);
            if (far) {
                // the nine blocks of level 1 as runs of the unified array (all 18 range words requested together); a block without super-rows (grid border, outside the
                // focus box) leaves the whole query to the cooperative rounds
                walk = g.cell_start9 != nullptr && !(cx < -r_max || cx > g.nx - 1 + r_max || cy < -r_max || cy > g.ny - 1 + r_max || cz < -r_max || cz > g.nz - 1 + r_max);
                const int xs = max(cx - 4, 0), xe = min(cx + 4, g.nx - 1);
                int rb[9], rn[9];
#pragma unroll
                for (int k = 0; k < 9; k++) {
                    const int a = k / 3 - 1, b = k % 3 - 1;
                    const int Y = cy + 3 * a, Z = cz + 3 * b;
                    const bool blk = Y + 1 >= 0 && Y - 1 < g.ny && Z + 1 >= 0 && Z - 1 < g.nz && xs <= xe;
                    const bool has9 = blk && Y >= g.by0 && Y < g.by0 + g.bny && Z >= g.bz0 && Z < g.bz0 + g.bnz && xs >= g.bx0 && xe < g.bx0 + g.bnx;
                    if (blk && !has9) walk = false;
                    const int* row = g.cell_start9 + (has9 && walk ? srow_index(g, g.bx0, Y, Z) - g.bx0 : 0);
                    const int b0 = has9 && walk ? row[xs] : 0, b1 = has9 && walk ? row[xe + 1] : 0;
                    rb[k] = b0; rn[k] = b1 - b0;
                }
#pragma unroll
                for (int k = 0; k < 9; k++) { tab.ser.b[k][lane] = rb[k]; tab.ser.n[k][lane] = walk ? rn[k] : 0; }
            }
            __builtin_amdgcn_wave_barrier();
            if (walk) {
                Sel5P sel; sel.init(gate, &pool.p[0][lane], &pool.j[0][lane]);
                int k = 0, cj = 0, ce = 0;
                auto fetch = [&](float4& p0, float4& p1, float4& p2, float4& p3, int& pj, int& pe) {
                    while (cj >= ce && k < 9) {      // the next run that holds points (no pruning by box distance: with the gate as the only bound nothing is dropped)
                        const int b = tab.ser.b[k][lane], n = tab.ser.n[k][lane];
                        k++;
                        if (n > 0) { cj = b; ce = b + n; }
                    }
                    pj = cj; pe = ce;
                    const float4* q = (const float4*)((const char*)g.pts + ((unsigned)min(cj, max(ce - 1, 0)) << 4));
                    p0 = q[0]; p1 = q[1]; p2 = q[2]; p3 = q[3];
                    cj += 4;
                };
                float4 a0, a1, a2, a3, b0, b1, b2, b3;
                int aj = 0, ae = 0, bj = 0, be = 0;
                fetch(a0, a1, a2, a3, aj, ae);
                for (;;) {
                    if (!(aj < ae)) break;
                    fetch(b0, b1, b2, b3, bj, be);
                    asm volatile("" : "+v"(a0.w), "+v"(a1.w), "+v"(a2.w), "+v"(a3.w));
                    sel.chunk(a0, a1, a2, a3, aj, ae, px, py, pz);
                    if (!(bj < be)) break;
                    fetch(a0, a1, a2, a3, aj, ae);
                    asm volatile("" : "+v"(b0.w), "+v"(b1.w), "+v"(b2.w), "+v"(b3.w));
                    sel.chunk(b0, b1, b2, b3, bj, be, px, py, pz);
                }
                Top5 t1;
                const bool redo = sel.finish(px, py, pz, t1);
                if (!redo) {      // (a bucket tie: the cooperative rounds repeat level 1 for this query)
                    const double margin = 4.0 * c + gap0;
                    const bool done = t1.d[4] < (float)(0.999 * margin * margin) || 4 >= r_max;
#pragma unroll
                    for (int r = 0; r < 5; r++) {
                        oK[r] = t1.j[r] >= 0 ? (((unsigned long long)__float_as_uint(t1.d[r]) << 32) | (unsigned)__float_as_int(t1.p[r].w)) : oK[r];
                        oJ[r] = t1.j[r];
                    }
                    lvl = 2;
                    if (done) { nn = t1; lvl = 0; }      // else: level 2 and beyond by the cooperative rounds
                }
            }
            __builtin_amdgcn_wave_barrier();
            mask = __ballot(far && lvl != 0);
        }
        const bool served = far && lvl == 0;      // settled by the lane-serial level 1: nn is final
        for (int m = 1; mask; m++) {
            unsigned long long todo = __ballot(far && lvl == m);
            while (todo) {
                // the first kFarG raised bits of the round: group r serves the r-th; an owner lane knows its rank among them
                const int rank = __popcll(todo & ((1ull << lane) - 1ull));
                const bool mine = ((todo >> lane) & 1ull) && rank < kFarG;
                unsigned long long mm = todo;
                int bit = -1;
#pragma unroll
                for (int r = 0; r < kFarG; r++) {
                    const int p = mm ? __builtin_ctzll(mm) : -1;
                    if (r == grp) bit = p;
                    if (mm) mm &= mm - 1ull;
                }
                todo = mm;
                const bool lv = bit >= 0;
                const int src = max(bit, 0);
                const float qx = __shfl(px, src), qy = __shfl(py, src), qz = __shfl(pz, src);
                unsigned long long K[5]; int J[5];
#pragma unroll
                for (int r = 0; r < 5; r++) { K[r] = shfl_u64(oK[r], src); J[r] = __shfl(oJ[r], src); }
                bool tie = false, done;
                if (exact_only) done = far_level<kFarL, true>(g, m, lv, sub, grp, qx, qy, qz, r_max, K, J, tab, pool, tie);
                else {
                    unsigned long long K0[5]; int J0[5];
#pragma unroll
                    for (int r = 0; r < 5; r++) { K0[r] = K[r]; J0[r] = J[r]; }
                    done = far_level<kFarL, false>(g, m, lv, sub, grp, qx, qy, qz, r_max, K, J, tab, pool, tie);
                    if (__any(tie)) {      // a bucket tie somewhere in the wave (rare): those groups repeat the level with exact keys, the others keep what they have
                        bool t2;
                        const bool d2 = far_level<kFarL, true>(g, m, lv && tie, sub, grp, qx, qy, qz, r_max, K0, J0, tab, pool, t2);
                        if (tie) {
                            done = d2;
#pragma unroll
                            for (int r = 0; r < 5; r++) { K[r] = K0[r]; J[r] = J0[r]; }
                        }
                    }
                }
                // back to the owner lanes: lane 0 of group `rank` holds the query's five best
                const int from = min(rank, kFarG - 1) * kFarL;
                const bool dn = __shfl((int)done, from) != 0;
#pragma unroll
                for (int r = 0; r < 5; r++) {
                    const unsigned long long rk = shfl_u64(K[r], from);
                    const int rj = __shfl(J[r], from);
                    if (mine) { oK[r] = rk; oJ[r] = rj; }
                }
                if (mine) lvl = dn ? 0 : m + 1;
            }
            mask = __ballot(far && lvl != 0);
        }
        if (far && !served) {
#pragma unroll
            for (int r = 0; r < 5; r++) { nn.d[r] = __uint_as_float((unsigned)(oK[r] >> 32)); nn.j[r] = oJ[r]; }
            nn.have = false;
        }
    }
    bool ok = false;
    if (live) {
        float4 r0 = make_float4(0.f, 0.f, 0.f, 0.f), rb = r0; double score = 0.0;
        store_debug_nn(g, nn, i, A.dbg_idx, A.dbg_d2);
        if (kind == 0) ok = surf_fit(g, P, nn, ql, px, py, pz, r0, score);
        else ok = edge_fit(g, P, nn, px, py, pz, r0, rb);
        store_record(A, kind, i, ok, r0, rb, score);
    }
    const int n_ok = __popcll(__ballot(ok));
    if (lane == 0) A.block_counts[blockIdx.x] = n_ok;
}

}  // namespace lili
