// Cooperative association for SMALL launches: L = 2 / 4 / 8 / 16 lanes of one wave serve ONE query.
//
// Why: k_associate_* (lili_s2m.hip) gives every query one lane, and that lane walks the query's whole dependent chain — range words,
// 6-7 trips of four candidates, winners, fit: ~11 us.  At 200 k queries the chip holds 3 waves per SIMD and the chains overlap; at the
// sizes the reference itself produces (1-3 k surf + 0.1-1 k edge features per keyframe, L/src/BackendFusion.cpp:1601-1681) or a rank's
// shard of an 8-way split (25 k) there is less than one wave per SIMD and the launch lasts exactly one chain.  Here the L lanes of a
// group load L * U consecutive candidates of the query's run per trip (one coalesced request, normally the whole inner block in ONE
// trip), keep their own exact top five, and the group's five best are extracted with DPP butterflies (quad_perm, row_half_mirror,
// row_mirror: no LDS, no barrier).  The shell of the 5x5x5 block and the nine-row walk of queries outside the super-row box are
// split over the lanes run by run.  The fit, the gates and the record stores are the functions of lili_s2m_dev.h, evaluated on the
// same five neighbours in the same (distance, original index) order — records, counts and debug rows are bit-identical to
// k_associate_* (tests/test_coop_gpu.py); the search is the exact one: every candidate the one-lane walk may prune
// lies beyond the group's current fifth best, which is an upper bound of the final one.
//
// Reference behaviour replaced: the same as lili_s2m.hip (findCorrespondingSurfFeatures L/src/BackendFusion.cpp:1601-1681,
// R/src/BackendFusion.cpp:1464-1520, L/src/LidarOdometry.cpp:352-413; findCorrespondingCornerFeatures L:1531-1599, R:1394-1462).
#include "lili_s2m_dev.h"

namespace lili {

constexpr int kCoopBlock = 256;

template <int CTRL> __device__ __forceinline__ unsigned dpp_u32(unsigned v) {
    return (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, 0xf, 0xf, false);
}
// one butterfly step of a group-wide minimum of (64-bit key, payload): afterwards both partners hold the smaller pair
template <int CTRL> __device__ __forceinline__ void kmin_step(unsigned& lo, unsigned& hi, int& j) {
    const unsigned olo = dpp_u32<CTRL>(lo), ohi = dpp_u32<CTRL>(hi);
    const int oj = (int)dpp_u32<CTRL>((unsigned)j);
    const unsigned long long a = ((unsigned long long)hi << 32) | lo, b = ((unsigned long long)ohi << 32) | olo;
    const bool take = b < a;
    lo = take ? olo : lo; hi = take ? ohi : hi; j = take ? oj : j;
}
// DPP controls: lane ^ 1, lane ^ 2 inside a quad; i <-> 7 - i inside 8 lanes; i <-> 15 - i inside a row of 16.  After the quad steps
// the four lanes of a quad agree, so the mirror steps pair equal halves: every lane of the group ends with the group's minimum.
template <int L> __device__ __forceinline__ void group_kmin(unsigned& lo, unsigned& hi, int& j) {
    kmin_step<0xB1>(lo, hi, j);
    if constexpr (L >= 4) kmin_step<0x4E>(lo, hi, j);
    if constexpr (L >= 8) kmin_step<0x141>(lo, hi, j);
    if constexpr (L >= 16) kmin_step<0x140>(lo, hi, j);
}
// The five smallest (distance, original index) keys of the L sorted per-lane lists of a group, in order, on every lane of the group:
// five rounds of [group minimum of the list heads, the owner pops].  Keys of real candidates are unique (distinct map points); the
// initial sentinels (bound, INT_MAX) are equal on all lanes and pop together — harmless, a lane holds five entries and pops at most five.
template <int L> __device__ __forceinline__ void group_top5(const Sel5& sel, unsigned long long K[5], int J[5]) {
    unsigned long long k0 = sel.k[0], k1 = sel.k[1], k2 = sel.k[2], k3 = sel.k[3], k4 = sel.k[4];
    int j0 = sel.j[0], j1 = sel.j[1], j2 = sel.j[2], j3 = sel.j[3], j4 = sel.j[4];
#pragma unroll
    for (int r = 0; r < 5; r++) {
        unsigned lo = (unsigned)k0, hi = (unsigned)(k0 >> 32);
        int jj = j0;
        group_kmin<L>(lo, hi, jj);
        const unsigned long long m = ((unsigned long long)hi << 32) | lo;
        K[r] = m; J[r] = jj;
        const bool pop = k0 == m;
        k0 = pop ? k1 : k0; k1 = pop ? k2 : k1; k2 = pop ? k3 : k2; k3 = pop ? k4 : k3; k4 = pop ? ~0ull : k4;
        j0 = pop ? j1 : j0; j1 = pop ? j2 : j1; j2 = pop ? j3 : j2; j3 = pop ? j4 : j3; j4 = pop ? -1 : j4;
    }
}

// candidates [beg, end) of ONE run, dealt to the L lanes of the group in stripes: lane `sub` takes beg + sub + L * (U * trip + u).
// Consecutive lanes read consecutive 16-byte points; the loads of a trip are independent.  Positions past the end re-read the run's
// last point and are masked by position.
template <int L, int U>
__device__ __forceinline__ void coop_scan_run(const GridView& g, Sel5& sel, int beg, int end, int sub, float qx, float qy, float qz) {
    if (!(beg < end)) return;
    const int last = end - 1;
    for (int base = beg + sub; base - sub < end; base += L * U) {
        float4 p[U];
#pragma unroll
        for (int u = 0; u < U; u++) p[u] = load_pt(g, min(base + L * u, last));
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int j = base + L * u;
            asm volatile("" : "+v"(p[u].w));
            const unsigned du = j < end ? __float_as_uint(dist2(p[u], qx, qy, qz)) : 0x7f800000u;
            if (du <= sel.worst_bits()) sel.insert(__uint_as_float(du), p[u], j);
        }
    }
}

// Exact 5-NN of one query by the L lanes of its group.  Returns the neighbours (positions in the cell-sorted array, f32 distances in the
// oracle's (d2, index) order) on EVERY lane of the group.  All 64 lanes of the wave must call this (the group reductions are DPP).
template <int L>
__device__ __forceinline__ void knn5_coop(const GridView& g, bool live, int sub, float qx, float qy, float qz, float bound, Top5& best) {
    constexpr int U = L >= 16 ? 2 : 4;
    Sel5 sel; sel.init(bound);
    const int R = g.reach;
    int cx = 0, cy = 0, cz = 0;
    bool in = live && isfinite(qx) && isfinite(qy) && isfinite(qz);
    if (in) {
        cx = cell_coord(qx, g.ox, g.inv_cell); cy = cell_coord(qy, g.oy, g.inv_cell); cz = cell_coord(qz, g.oz, g.inv_cell);
        // queries more than `reach` cells outside the grid cannot have a neighbour within the gate radius
        in = !(cx < -R || cx > g.nx - 1 + R || cy < -R || cy > g.ny - 1 + R || cz < -R || cz > g.nz - 1 + R);
    }
    const int x0 = max(cx - 1, 0), x1 = min(cx + 1, g.nx - 1);
    const bool inner9 = in && g.cell_start9 && cy >= g.by0 && cy < g.by0 + g.bny && cz >= g.bz0 && cz < g.bz0 + g.bnz &&
                        x0 >= g.bx0 && x1 < g.bx0 + g.bnx;
    // one row of the base index, cells xa..xb, if its box distance `lb` (0.1 % conservative) can still beat this lane's fifth best
    auto scan_base_row = [&](Sel5& s, int y, int z, int xa, int xb, float lb) {
        if (y < 0 || y >= g.ny || z < 0 || z >= g.nz || xa > xb || lb > s.worst()) return;
        const int* cs = g.cell_start + (size_t)(z * g.ny + y) * g.nx;
        scan_run(g, s, cs[xa], cs[xb + 1], qx, qy, qz);
    };
    if (inner9) {
        if (x0 <= x1) {      // super-row layout: the inner 27 cells are ONE run, dealt to the lanes in stripes
            const int* row = g.cell_start9 + srow_index(g, g.bx0, cy, cz) - g.bx0;
            coop_scan_run<L, U>(g, sel, row[x0], row[x1 + 1], sub, qx, qy, qz);
        }
    } else if (in && x0 <= x1) {
        // outside the super-row box: the nine rows of the inner block, one row per lane and round (centre, faces, diagonals)
        for (int n = sub; n < 9; n += L) {
            const int w = row_order(n);
            scan_base_row(sel, cy + w / 3 - 1, cz + w % 3 - 1, x0, x1, row_lower_bound(g, qy, qz, cy, cz, w / 3 - 1, w % 3 - 1));
        }
    }
    unsigned long long K[5]; int J[5];
    group_top5<L>(sel, K, J);
    if (R == 2) {
        const double c = g.cell;
        const double fxm = (double)qx - (g.ox + (double)cx * c), fxp = (g.ox + (double)(cx + 1) * c) - (double)qx;
        const double fym = (double)qy - (g.oy + (double)cy * c), fyp = (g.oy + (double)(cy + 1) * c) - (double)qy;
        const double fzm = (double)qz - (g.oz + (double)cz * c), fzp = (g.oz + (double)(cz + 1) * c) - (double)qz;
        const double margin = c + fmax(fmin(fmin(fmin(fxm, fxp), fmin(fym, fyp)), fmin(fzm, fzp)), 0.0);
        const float W = __uint_as_float((unsigned)(K[4] >> 32));      // the group's fifth best (or the bound): an upper bound of the final one
        const bool shell = in && !(W < (float)(0.999 * margin * margin));      // uniform within the group
        if (__any(shell)) {
            // Every lane restarts with the bound W (candidates beyond it can never matter; ties at W enter and are ordered by the full key);
            // lane 0 of the group carries the five found so far.  The shell's runs are dealt to the lanes one by one:
            //   r < 16      the (dy, dz) rows with |dy| = 2 or |dz| = 2, up to five cells, x-trimmed by the lane's own fifth best
            //   r >= 16     the two cells x = cx -+ 2 of each inner row — two super cells with the super-row layout, else 18 single cells
            Sel5 s2; s2.init(W);
            if (sub == 0) {
#pragma unroll
                for (int r = 0; r < 5; r++) { s2.k[r] = K[r]; s2.j[r] = J[r]; }
            }
            if (shell) {
                const bool side9 = inner9 && (cx - 2 < 0 || cx - 2 >= g.bx0) && (cx + 2 >= g.nx || cx + 2 < g.bx0 + g.bnx);
                const double g1m = fmax(fxm, 0.0), g1p = fmax(fxp, 0.0), g2m = fmax(fxm + c, 0.0), g2p = fmax(fxp + c, 0.0);
                const int nr = side9 ? 18 : 34;
                for (int r = sub; r < nr; r += L) {
                    if (r < 16) {
                        // kShellDy / kShellDz of lili_s2m_dev.h (nearest rows first), packed three bits per entry: a table indexed per lane would live in scratch
                        const int dy = (int)((0x900900659812ull >> (3 * r)) & 7ull) - 2, dz = (int)((0x8206599004a0ull >> (3 * r)) & 7ull) - 2;
                        const double gy = dy == 0 ? 0.0 : fmax(dy < 0 ? fym + (double)(-dy - 1) * c : fyp + (double)(dy - 1) * c, 0.0);
                        const double gz = dz == 0 ? 0.0 : fmax(dz < 0 ? fzm + (double)(-dz - 1) * c : fzp + (double)(dz - 1) * c, 0.0);
                        const double lbr = 0.999 * (gy * gy + gz * gz);
                        const float wv = s2.worst();
                        const int dl = (float)(lbr + 0.999 * g2m * g2m) > wv ? ((float)(lbr + 0.999 * g1m * g1m) > wv ? 0 : 1) : 2;
                        const int dr = (float)(lbr + 0.999 * g2p * g2p) > wv ? ((float)(lbr + 0.999 * g1p * g1p) > wv ? 0 : 1) : 2;
                        scan_base_row(s2, cy + dy, cz + dz, max(cx - dl, 0), min(cx + dr, g.nx - 1), (float)lbr);
                    } else if (side9) {
                        const int* row = g.cell_start9 + srow_index(g, g.bx0, cy, cz) - g.bx0;
                        const int xs = r == 16 ? cx - 2 : cx + 2;
                        const double gx = r == 16 ? g2m : g2p;
                        if (xs >= 0 && xs < g.nx && !((float)(0.999 * gx * gx) > s2.worst())) scan_run(g, s2, row[xs], row[xs + 1], qx, qy, qz);
                    } else {
                        const int n = (r - 16) >> 1, dy = n / 3 - 1, dz = n % 3 - 1;
                        const double gy = dy == 0 ? 0.0 : fmax(dy < 0 ? fym : fyp, 0.0), gz = dz == 0 ? 0.0 : fmax(dz < 0 ? fzm : fzp, 0.0);
                        const int xs = (r & 1) ? cx + 2 : cx - 2;
                        const double gx = (r & 1) ? g2p : g2m;
                        scan_base_row(s2, cy + dy, cz + dz, xs, (xs >= 0 && xs < g.nx) ? xs : xs - 1, (float)(0.999 * (gy * gy + gz * gz) + 0.999 * gx * gx));
                    }
                }
            }
            // groups that did not need the shell get their five back unchanged: lane 0 holds them and every other list starts at (W, INT_MAX)
            group_top5<L>(s2, K, J);
        }
    }
#pragma unroll
    for (int r = 0; r < 5; r++) { best.d[r] = __uint_as_float((unsigned)(K[r] >> 32)); best.j[r] = J[r]; }
    best.have = false; best.aux = 0;
}

// Blocks [0, E.nb) take the edge queries, the rest the surf queries (either may be absent); kCoopBlock / L queries per block.
// LIN: the lane that holds a group's record also linearises it (the flavours without count scaling, see k_associate_lin) and the block
// stores ONE partial for k_reduce_partials.
// cb_blocks > 0 (with LIN): the count-scaled flavour (ROT: residual weight = num / N with the scan's correspondence count N,
// R/src/BackendFusion.cpp:843,861) linearises in this launch as well — N exists only after the last workgroup has counted, so every
// workgroup adds its count and an arrival to ONE 64-bit word of the slot's state and waits until all cb_blocks workgroups have arrived
// (a relaxed agent-scope atomic and a poll: only the VALUE is exchanged, no other data, hence no fence).  Small grids only (<= 256
// workgroups, all resident); the wait is bounded — a workgroup that gives up contributes no rows and flags the slot.
template <int L, bool LIN>
__device__ __forceinline__ void assoc_coop_block(const AssocArgs& A /*the arguments of THIS workgroup's kind*/, const bool edge, const int vb /*workgroup index within the kind*/,
                                                 const int b /*workgroup index within the slot's launch*/, const PoseArg& pa, const MatchParams& P,
                                                 double* __restrict__ part_surf, double* __restrict__ part_edge, SlotState* cb_state, int cb_blocks) {
    // (the caller selects A: a select between two by-value kernel arguments stays an address select in the kernel-argument segment only when it is
    // written in the kernel itself — through this function's parameters it put both structs into scratch, 408 bytes per lane)
    constexpr int QPB = kCoopBlock / L;
    const int sub = (int)threadIdx.x & (L - 1);
    const int i = vb * QPB + ((int)threadIdx.x / L);
    const bool live = i < A.n_q;
    const float4 ql = A.queries[live ? i : 0];     // requested before the (dependent, scalar) pose loads: the two latencies overlap
    dq Q2; d3 T2;
    load_assoc_pose(pa, P, Q2, T2);
    const d3 pmd = qrot(Q2, d3{(double)ql.x, (double)ql.y, (double)ql.z}) + T2;   // transformPoint, L/src/BackendFusion.cpp:695-711
    const float px = (float)pmd.x, py = (float)pmd.y, pz = (float)pmd.z;
    Top5 nn;
    knn5_coop<L>(A.g, live, sub, px, py, pz, gate_bound(edge ? P.edge_gate : P.kd_max_radius), nn);
    const bool mine = live && sub == 0;        // the lane that owns the group's record
    LaneRec rec{};
    bool ok = false;
    if (live) {          // every lane of the group evaluates the fit (same instructions, no divergence inside the group); one stores
        if (mine) store_debug_nn(A.g, nn, i, A.dbg_idx, A.dbg_d2);
        if (edge) {
            float4 ra, rb;
            ok = edge_fit(A.g, P, nn, px, py, pz, ra, rb);
            if (mine) { A.rec0[i] = ra; reinterpret_cast<float4*>(A.rec1)[i] = rb; A.valid[i] = ok ? 1 : 0; }
            rec.r0 = ra; rec.r1 = rb;
        } else {
            float4 rn; double score;
            ok = surf_fit(A.g, P, nn, ql, px, py, pz, rn, score);
            if (mine) { A.rec0[i] = rn; reinterpret_cast<double*>(A.rec1)[i] = score; A.valid[i] = ok ? 1 : 0; }
            rec.r0 = rn; rec.score = score;
        }
        rec.ql = ql;
    }
    rec.ok = ok && mine;
    store_block_count<kCoopBlock>(rec.ok, A.block_counts, vb);
    if constexpr (LIN) {
        __shared__ __attribute__((aligned(16))) double lds[kCoopBlock * kRow];
        __shared__ unsigned long long s_counts;
        bool rows_ok = true;
        if (cb_blocks > 0) {
            if (threadIdx.x == 0) {
                const unsigned long long mine = (unsigned long long)(unsigned)A.block_counts[vb];          // store_block_count has just written it (same thread)
                __hip_atomic_fetch_add(&cb_state->cnt_word, (1ull << 48) | (edge ? mine << 24 : mine), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                unsigned long long w = 0ull;
                for (unsigned spins = 0; spins < (1u << 22); spins++) {
                    w = __hip_atomic_load(&cb_state->cnt_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if ((int)(w >> 48) >= cb_blocks) break;
                    __builtin_amdgcn_s_sleep(1);
                }
                s_counts = (int)(w >> 48) >= cb_blocks ? w : ~0ull;
            }
            __syncthreads();
            rows_ok = s_counts != ~0ull;
            if (!rows_ok && threadIdx.x == 0 && b == 0) cb_state->gn_status = 2;
        }
        dq Q; d3 T;
        load_body_pose(pa, Q, T);
        double Jr[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        double cost = 0.0;
        if (rec.ok && rows_ok) {
            if (edge) {
                double sw = (double)rec.r0.w;
                if (P.scale_edge_num > 0) sw = (double)__fdiv_rn(__fmul_rn(rec.r0.w, (float)P.scale_edge_num), (float)(int)((s_counts >> 24) & 0xffffffull));      // R:843, float arithmetic (lin_edge_body)
                cost = edge_lin_row(P, Q, T, rec.ql, rec.r0, rec.r1, sw, Jr);
            } else {
                double score = rec.score;
                if (P.scale_surf_num > 0) score = score * P.scale_surf_num / (double)(int)(s_counts & 0xffffffull);                                                 // R:861 (lin_surf_body)
                cost = surf_lin_row(P, Q, T, dq{P.q_lb_inv_jet[0], P.q_lb_inv_jet[1], P.q_lb_inv_jet[2], P.q_lb_inv_jet[3]}, rec.ql, rec.r0, score, Jr);
            }
        }
        if (!rows_ok) rec.ok = false;
        GramAcc ga; ga.init();
        ga.add_rows(Jr, cost, rec.ok, lds);
        ga.finish(lds, edge ? part_edge + (size_t)vb * kPartialStride : part_surf + (size_t)vb * kPartialStride);
    }
}
template <int L, bool LIN>
__global__ __launch_bounds__(kCoopBlock) void k_associate_coop(AssocArgs S, AssocArgs E, PoseArg pa, MatchParams P, double* __restrict__ part_surf, double* __restrict__ part_edge,
                                                               SlotState* cb_state, int cb_blocks) {
    const int b = (int)blockIdx.x;
    const bool edge = b < E.nb;
    const AssocArgs& A = edge ? E : S;
    assoc_coop_block<L, LIN>(A, edge, edge ? b : b - E.nb, b, pa, P, part_surf, part_edge, cb_state, cb_blocks);
}
// The association of EVERY keyframe of a sliding window in ONE launch (round 4; lili_s2m_associate_window): workgroup b belongs to the last slot
// whose first_block <= b and is that slot's workgroup b - first_block of k_associate_coop<L, false> — same lanes per query for all slots (the
// records do not depend on L), the two map views shared by all slots.  It was one launch per keyframe on forked streams (two event
// waits per slot cost more than the launches themselves at the sizes a keyframe has).
template <int L>
__global__ __launch_bounds__(kCoopBlock) void k_associate_coop_window(WinAssocArgs W, MatchParams P) {
    const int bid = (int)blockIdx.x;
    int i = 0;
#pragma unroll
    for (int k = 1; k < kWindowMaxSlots; k++) if (k < W.n && bid >= W.s[k].first_block) i = k;
    const WinAssocSlot& ws = W.s[i];
    const int b = bid - ws.first_block;
    const bool edge = b < ws.k[1].nb;
    const WinAssocKind& k = ws.k[edge ? 1 : 0];           // uniform per workgroup: scalar loads from the kernel-argument segment
    AssocArgs A{};
    A.queries = k.queries; A.n_q = k.n_q; A.g = W.g[edge ? 1 : 0];
    A.rec0 = k.rec0; A.rec1 = k.rec1; A.valid = k.valid;
    A.dbg_idx = k.dbg_idx; A.dbg_d2 = k.dbg_d2; A.block_counts = k.block_counts;
    A.nb = k.nb;
    assoc_coop_block<L, false>(A, edge, edge ? b : b - ws.k[1].nb, b, ws.pa, P, nullptr, nullptr, nullptr, 0);
}
template __global__ void k_associate_coop_window<2>(WinAssocArgs, MatchParams);
template __global__ void k_associate_coop_window<4>(WinAssocArgs, MatchParams);
template __global__ void k_associate_coop_window<8>(WinAssocArgs, MatchParams);
template __global__ void k_associate_coop_window<16>(WinAssocArgs, MatchParams);

// ================================================================================================
// Persistent outer iterations for small scans (round 3): n_iters x [re-associate, linearise, reduce, Gauss-Newton update] in ONE launch.
//
// Why: at the sizes the reference itself produces (1-3 k surf + 0.1-1 k edge features per keyframe) an outer iteration is two or three launches
// whose floor is ~4 us EACH in the kernel trace (an empty kernel costs 3.7-4.4 us there): 15.0 us per iteration for the front-end flavour at 2 k
// queries, 19.9 us for ROT, of which only ~6 us is the cooperative association + rows.  Here the workgroups of k_associate_coop stay
// resident for the whole registration: every iteration each workgroup associates its queries at the pose it holds in LDS, linearises its
// records, publishes its 40-double partial as key-tagged granules, the partials are summed in index order (one hop up to 16 workgroups, else
// group sums of 16 first) so that EVERY workgroup holds the same record bit for bit, and every workgroup applies the same Gauss-Newton step
// itself — nothing is broadcast, nothing goes back to the host.  The count-scaled ROT flavour exchanges the correspondence counts the same way
// before its rows.  Records, counts and debug rows are stored every iteration like the separate launches do; results equal lili_s2m_iterate's
// up to the partition of the Gram sum (poses <= 1e-10, tests/test_coop_gpu.py).  All workgroups must be resident (<= 256 of 256 threads);
// every wait is bounded and ends with gn_status = 2 instead of a hang.
// ================================================================================================
struct IterArgs {
    SlotState* state;
    double* part;                 // [2 parities][nb][kPartialStride]
    double* gsum;                 // [2 parities][ng][kPartialStride]
    double* cpart;                // [2 parities][nb + ng][4]            correspondence counts (surf, edge) as granules
    int nb, ng, n_iters, derive_assoc;
    unsigned long long launch;
};
struct IterShared {
    double vals[16][40];
    double cvals[16][2];
    double tot[40];
    double ctot[2];
    double full[64];
    double H[6][6], gv[6];
    double pose[7];
    double last_delta[6];
    int status;                   // 0 ok, 1 = normal matrix not positive definite (pose kept), 2 = an exchange gave up
    int n_updates;
};
// H d = g (no damping) on 42 lanes as in lili_s2m_lm.hip: returns false if a pivot is not positive or the step is not finite
__device__ __forceinline__ bool gn_solve_wave(const double (*H)[6], const double* g, double d[6]) {
    const int lane = threadIdx.x & 63;
    const int ri = lane / 7, cj = lane - 7 * ri;
    const bool in = lane < 42;
    double a = 0.0;
    if (in) a = cj < 6 ? H[ri][cj] : g[ri];
    bool okc = true;
    double pinv[6];
#pragma unroll
    for (int p = 0; p < 6; p++) {
        const double piv = __shfl(a, p * 7 + p);
        okc = okc && (piv > 0.0);
        pinv[p] = 1.0 / piv;
        const double rowp = __shfl(a, p * 7 + (in ? cj : 0));
        const double colp = __shfl(a, (in ? ri : 0) * 7 + p);
        if (in && ri > p) a -= (colp * pinv[p]) * rowp;
    }
#pragma unroll
    for (int p = 5; p >= 0; p--) {
        d[p] = __shfl(a, p * 7 + 6) * pinv[p];
        const double up = __shfl(a, (in ? ri : 0) * 7 + p);
        if (in && cj == 6 && ri < p) a -= up * d[p];
    }
#pragma unroll
    for (int i = 0; i < 6; i++) okc = okc && (d[i] == d[i]);
    return okc;
}
template <int L>
__global__ __launch_bounds__(kCoopBlock, 2) void k_iterate_coop(AssocArgs S, AssocArgs E, MatchParams P, IterArgs a) {      // (two waves per SIMD: <= 256 registers, two workgroups per CU)
    constexpr int QPB = kCoopBlock / L;
    __shared__ __attribute__((aligned(16))) double lds[kCoopBlock * kRow];
    __shared__ IterShared sh;
    const int b = (int)blockIdx.x;
    const bool edge = b < E.nb;
    const AssocArgs& A = edge ? E : S;
    const int vb = edge ? b : b - E.nb;
    const int sub = (int)threadIdx.x & (L - 1);
    const int i = vb * QPB + ((int)threadIdx.x / L);
    const bool live = i < A.n_q;
    const bool mine = live && sub == 0;
    const bool wave0 = threadIdx.x < 64;
    const bool scaled = P.scale_surf_num > 0 || P.scale_edge_num > 0;
    const float4 ql = A.queries[live ? i : 0];
    if (threadIdx.x < 7) sh.pose[threadIdx.x] = a.state->pose[threadIdx.x];
    if (threadIdx.x == 0) { sh.status = 0; sh.n_updates = 0; }
    for (int it = 0; it < a.n_iters; it++) {
        __syncthreads();
        if (sh.status == 2) break;                                        // uniform
        const int par = it & 1;
        const dq Q{sh.pose[3], sh.pose[4], sh.pose[5], sh.pose[6]};
        const d3 T{sh.pose[0], sh.pose[1], sh.pose[2]};
        dq Q2 = Q; d3 T2 = T;
        if (a.derive_assoc) {      // L/src/BackendFusion.cpp:929-930 (load_assoc_pose)
            Q2 = qmul(Q, dq{P.q_lb_inv[0], P.q_lb_inv[1], P.q_lb_inv[2], P.q_lb_inv[3]});
            T2 = T - qrot(Q2, d3{P.t_lb[0], P.t_lb[1], P.t_lb[2]});
        }
        const d3 pmd = qrot(Q2, d3{(double)ql.x, (double)ql.y, (double)ql.z}) + T2;
        const float px = (float)pmd.x, py = (float)pmd.y, pz = (float)pmd.z;
        Top5 nn;
        knn5_coop<L>(A.g, live, sub, px, py, pz, gate_bound(edge ? P.edge_gate : P.kd_max_radius), nn);
        LaneRec rec{};
        bool ok = false;
        if (live) {
            if (mine) store_debug_nn(A.g, nn, i, A.dbg_idx, A.dbg_d2);
            if (edge) {
                float4 ra, rb;
                ok = edge_fit(A.g, P, nn, px, py, pz, ra, rb);
                if (mine) { A.rec0[i] = ra; reinterpret_cast<float4*>(A.rec1)[i] = rb; A.valid[i] = ok ? 1 : 0; }
                rec.r0 = ra; rec.r1 = rb;
            } else {
                float4 rn; double score;
                ok = surf_fit(A.g, P, nn, ql, px, py, pz, rn, score);
                if (mine) { A.rec0[i] = rn; reinterpret_cast<double*>(A.rec1)[i] = score; A.valid[i] = ok ? 1 : 0; }
                rec.r0 = rn; rec.score = score;
            }
            rec.ql = ql;
        }
        rec.ok = ok && mine;
        store_block_count<kCoopBlock>(rec.ok, A.block_counts, vb);          // (ends with the block's count in A.block_counts[vb], written by thread 0)
        // ---- count-scaled flavour: the scan's correspondence counts, summed over the workgroups (round 2 it)
        double n_surf = 1.0, n_edge = 1.0;
        if (scaled) {
            const unsigned long long ckey = xchg_key(a.launch, 2 * it);
            double* cpart = a.cpart + (size_t)par * (a.nb + a.ng) * 4;
            if (threadIdx.x == 0) {
                const double c = (double)A.block_counts[vb];
                store_granule(cpart + (size_t)b * 4, edge ? 0.0 : c, ckey);
                store_granule(cpart + (size_t)b * 4 + 2, edge ? c : 0.0, ckey);
            }
            if (wave0) {
                bool okx = true;
                if (a.ng > 1) {
                    if (b % 16 == 0) {
                        okx = xchg_gather<2>(cpart + (size_t)b * 4, min(16, a.nb - b), ckey, sh.cvals, sh.ctot, 4);
                        if ((threadIdx.x & 63) < 2) store_granule(cpart + (size_t)(a.nb + b / 16) * 4 + 2 * (threadIdx.x & 63), sh.ctot[threadIdx.x & 63], ckey);
                    }
                    okx = xchg_gather<2>(cpart + (size_t)a.nb * 4, a.ng, ckey, sh.cvals, sh.ctot, 4) && okx;
                } else okx = xchg_gather<2>(cpart, a.nb, ckey, sh.cvals, sh.ctot, 4);
                if (!okx && threadIdx.x == 0) sh.status = 2;
            }
            __syncthreads();
            n_surf = sh.ctot[0]; n_edge = sh.ctot[1];
        }
        // ---- rows of this workgroup's records, Gram partial as granules (round 2 it + 1)
        double Jr[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        double cost = 0.0;
        if (rec.ok) {
            if (edge) {
                double sw = (double)rec.r0.w;
                if (P.scale_edge_num > 0) sw = (double)__fdiv_rn(__fmul_rn(rec.r0.w, (float)P.scale_edge_num), (float)(int)n_edge);      // R:843 (lin_edge_body)
                cost = edge_lin_row(P, Q, T, rec.ql, rec.r0, rec.r1, sw, Jr);
            } else {
                double score = rec.score;
                if (P.scale_surf_num > 0) score = score * P.scale_surf_num / (double)(int)n_surf;                                       // R:861 (lin_surf_body)
                cost = surf_lin_row(P, Q, T, dq{P.q_lb_inv_jet[0], P.q_lb_inv_jet[1], P.q_lb_inv_jet[2], P.q_lb_inv_jet[3]}, rec.ql, rec.r0, score, Jr);
            }
        }
        const unsigned long long key = xchg_key(a.launch, 2 * it + 1);
        double* part = a.part + (size_t)par * a.nb * kPartialStride;
        double* gsum = a.gsum + (size_t)par * a.ng * kPartialStride;
        GramAcc ga; ga.init();
        ga.add_rows(Jr, cost, rec.ok, lds);
        ga.finish(lds, part + (size_t)b * kPartialStride, key);
        if (wave0) {
            bool okx = true;
            if (a.ng > 1) {
                if (b % 16 == 0) {
                    okx = xchg_gather<40>(part + (size_t)b * kPartialStride, min(16, a.nb - b), key, sh.vals, sh.tot);
                    if ((threadIdx.x & 63) < 40) store_granule(gsum + (size_t)(b / 16) * kPartialStride + 2 * (threadIdx.x & 63), sh.tot[threadIdx.x & 63], key);
                }
                okx = xchg_gather<40>(gsum, a.ng, key, sh.vals, sh.tot) && okx;
            } else okx = xchg_gather<40>(part, a.nb, key, sh.vals, sh.tot);
            // ---- the Gauss-Newton step, identically in every workgroup: H = P^T G77 P, g = -P^T G7r, H d = g, x (+) d
            {
                const int lane = threadIdx.x;
                const int r = lane >> 3, c = lane & 7;
                const int aa = r < c ? r : c, bb = r < c ? c : r;
                sh.full[lane] = sh.tot[aa * 8 - aa * (aa - 1) / 2 + (bb - aa)];
            }
            LILI_WAVE_SYNC();
            {
                const int tid = threadIdx.x;
                const double x0 = sh.pose[3], x1 = sh.pose[4], x2 = sh.pose[5], x3 = sh.pose[6];
                auto jcol = [&](int c, double o[4]) {
                    o[0] = c == 0 ? -x1 : c == 1 ? -x2 : -x3;
                    o[1] = c == 0 ? x0 : c == 1 ? x3 : -x2;
                    o[2] = c == 0 ? -x3 : c == 1 ? x0 : x1;
                    o[3] = c == 0 ? x2 : c == 1 ? -x1 : x0;
                };
                if (tid < 42) {
                    const int ar = tid < 36 ? tid / 6 : tid - 36, bc = tid < 36 ? tid % 6 : 7;
                    double jb[4] = {0, 0, 0, 0}, ja[4] = {0, 0, 0, 0};
                    if (bc >= 3 && bc < 6) jcol(bc - 3, jb);
                    if (ar >= 3) jcol(ar - 3, ja);
                    const double* gram = sh.full;
                    auto Mrow = [&](int ii) -> double {
                        if (bc < 3 || bc == 7) return gram[ii * 8 + bc];
                        return ((gram[ii * 8 + 3] * jb[0] + gram[ii * 8 + 4] * jb[1]) + gram[ii * 8 + 5] * jb[2]) + gram[ii * 8 + 6] * jb[3];
                    };
                    double v;
                    if (ar < 3) v = Mrow(ar);
                    else v = ((ja[0] * Mrow(3) + ja[1] * Mrow(4)) + ja[2] * Mrow(5)) + ja[3] * Mrow(6);
                    if (tid < 36) sh.H[ar][bc] = v; else sh.gv[ar] = -v;
                }
            }
            LILI_WAVE_SYNC();
            double d[6];
            const bool solved = gn_solve_wave(sh.H, sh.gv, d);
            if (threadIdx.x == 0) {
                if (!okx) sh.status = 2;
                else if (!solved) sh.status = 1;
                else {
                    sh.status = 0;
                    sh.pose[0] += d[0]; sh.pose[1] += d[1]; sh.pose[2] += d[2];
                    const double nd2 = d[3] * d[3] + d[4] * d[4] + d[5] * d[5];
                    if (nd2 > 0.0) {
                        double sbd, cw;
                        if (nd2 < 0.25) sinc_cos_small(nd2, sbd, cw);
                        else {
                            double h2 = nd2; int k = 0;
                            while (h2 >= 0.25 && k < 60) { h2 *= 0.25; k++; }
                            double sc, cc;
                            sinc_cos_small(h2, sc, cc);
                            double sn = sc * sqrt(h2);
                            for (int q = 0; q < k; q++) { const double s2 = 2.0 * sn * cc, c2 = cc * cc - sn * sn; sn = s2; cc = c2; }
                            sbd = sn / sqrt(nd2); cw = cc;
                        }
                        const dq rq = qmul(dq{cw, sbd * d[3], sbd * d[4], sbd * d[5]}, dq{sh.pose[3], sh.pose[4], sh.pose[5], sh.pose[6]});
                        sh.pose[3] = rq.w; sh.pose[4] = rq.x; sh.pose[5] = rq.y; sh.pose[6] = rq.z;
                    }
#pragma unroll
                    for (int q = 0; q < 6; q++) sh.last_delta[q] = d[q];
                }
                sh.n_updates++;
            }
        }
    }
    __syncthreads();
    if (b == 0 && threadIdx.x == 0) {
        for (int q = 0; q < 7; q++) a.state->pose[q] = sh.pose[q];
        for (int q = 0; q < 6; q++) a.state->last_delta[q] = sh.last_delta[q];
        a.state->gn_status = sh.status;
        a.state->iters += sh.n_updates;
    }
}
template __global__ void k_iterate_coop<2>(AssocArgs, AssocArgs, MatchParams, IterArgs);
template __global__ void k_iterate_coop<4>(AssocArgs, AssocArgs, MatchParams, IterArgs);
template __global__ void k_iterate_coop<8>(AssocArgs, AssocArgs, MatchParams, IterArgs);
template __global__ void k_iterate_coop<16>(AssocArgs, AssocArgs, MatchParams, IterArgs);

#define LILI_COOP_INST(L) \
    template __global__ void k_associate_coop<L, false>(AssocArgs, AssocArgs, PoseArg, MatchParams, double*, double*, SlotState*, int); \
    template __global__ void k_associate_coop<L, true>(AssocArgs, AssocArgs, PoseArg, MatchParams, double*, double*, SlotState*, int);
LILI_COOP_INST(2)
LILI_COOP_INST(4)
LILI_COOP_INST(8)
LILI_COOP_INST(16)
#undef LILI_COOP_INST

}  // namespace lili
