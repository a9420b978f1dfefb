// Device-side building blocks of the scan-to-map matcher shared by the translation units that define its kernels
// (lili_s2m.hip: one query per lane; lili_s2m_coop.hip: several lanes per query; lili_s2m_lm.hip: the device Levenberg-Marquardt loop).
// Everything here is __device__ __forceinline__ or a type: exact 5-NN on the uniform-grid index, the plane / line fits with the
// reference's gates, residual rows + loss corrector, the f64-MFMA Gram accumulation.  Reference lines are cited at each function.
#pragma once
#include <type_traits>
#include "lili_kernels.h"
#include "lili_device_math.h"

namespace lili {

__device__ __forceinline__ int cell_coord(float v, double o, double inv_cell) {
    // f64 so that the covering argument of DESIGN.md §3 does not depend on f32 rounding of (v - o) / c
    return (int)floor(((double)v - o) * inv_cell);
}
__device__ __forceinline__ int cell_of(float4 p, const GridView& g) {
    if (!(isfinite(p.x) && isfinite(p.y) && isfinite(p.z))) return 0;   // never selected: its distance is NaN
    int cx = min(max(cell_coord(p.x, g.ox, g.inv_cell), 0), g.nx - 1);
    int cy = min(max(cell_coord(p.y, g.oy, g.inv_cell), 0), g.ny - 1);
    int cz = min(max(cell_coord(p.z, g.oz, g.inv_cell), 0), g.nz - 1);
    return (cz * g.ny + cy) * g.nx + cx;
}

// Super-rows: a second copy of the map in which the points of the 3x3 (y,z) rows around a row are stored together, sorted by
// x cell — "super cell" (x, y', z') holds the points of the nine cells (x, y'+dy, z'+dz) in the fixed order k = (dz+1)*3 + (dy+1), each in
// its base order.  The 27-cell neighbourhood of a query in cell (cx, cy, cz) is then ONE contiguous run: super cells cx-1..cx+1 of super-row
// (cy, cz) — two range words instead of eighteen, no row table, no per-row bounds, full chunks.  Super-rows exist for the cells of a box
// (bx0.., by0.., bz0..; the whole grid unless lili_map_focus names a region): queries elsewhere take the nine-row walk.  The copy lives behind
// the base points in the same array ([base | super-rows], at most 10 n entries); start9 holds positions in that array.
__device__ __forceinline__ size_t srow_index(const GridView& g, int x, int y, int z) {
    return ((size_t)(z - g.bz0) * (size_t)g.bny + (size_t)(y - g.by0)) * (size_t)g.bnx + (size_t)(x - g.bx0);
}

// ================================================================================================
// exact 5-NN inside the 27-cell neighbourhood
// ================================================================================================
struct Top5 {
    float d[5];
    int j[5];     // position in the cell-sorted array
    int aux;      // profiling only: chunks of four candidates this query went through
    float4 p[5];  // the five points themselves when `have` (the key selector has just loaded them to recompute the exact distances:
    bool have;    //  the fit then skips its own gather)
};
// FLANN L2_Simple on 3 floats (f32, x then y then z, no FMA)
__device__ __forceinline__ float dist2(float4 p, float qx, float qy, float qz) {
    float r = 0.f;
    float dx = qx - p.x; r += dx * dx;
    float dy = qy - p.y; r += dy * dy;
    float dz = qz - p.z; r += dz * dz;
    return r;
}
// Running 5 best as a sorted list of packed keys (f32 distance bits << 32 | original map index): squared
// distances are >= 0 so their bit patterns order like the values, and the low word makes the order the
// oracle's lexicographic (d2, index) — FLANN's own tie order is unspecified (App. B1).  Insertion is a
// 5-stage compare-exchange chain without branches: on a 64-lane wave some lane inserts at almost every
// candidate, so a branchy insertion path is executed (and stalls) nearly every iteration anyway.
// NaN distances have bit patterns above +inf and therefore never displace the initial (+inf, INT_MAX) keys.
struct Sel5 {
    unsigned long long k[5];
    int j[5];
    // `bound`: candidates enter only with d <= bound.  +inf gives the plain 5-NN; the association passes the gate of the
    // reference (`pointSearchSqDis[4] < gate`, rounded UP to f32), so that rows and shell cells beyond the gate are pruned
    // from the first candidate on — queries without 5 neighbours inside the gate then end with d[4] = bound >= gate, j = -1.
    __device__ __forceinline__ void init(float bound = 3.0e38f) {
        bound = fminf(bound, 3.0e38f);   // finite, so that the +inf of masked slots never qualifies
#pragma unroll
        for (int s = 0; s < 5; s++) { k[s] = ((unsigned long long)__float_as_uint(bound) << 32) | 0x7fffffffull; j[s] = -1; }
    }
    // Sorted insertion by rank: the five comparisons are independent (no compare-exchange chain), slot s takes its left
    // neighbour if the key ranks before s-1, the key itself if it ranks exactly at s, else keeps its value.
    __device__ __forceinline__ void insert(float d, float4 p, int jpos) {
        const unsigned long long key = ((unsigned long long)__float_as_uint(d) << 32) | (unsigned)__float_as_int(p.w);
        const bool c0 = key < k[0], c1 = key < k[1], c2 = key < k[2], c3 = key < k[3], c4 = key < k[4];
        k[4] = c3 ? k[3] : (c4 ? key : k[4]);  j[4] = c3 ? j[3] : (c4 ? jpos : j[4]);
        k[3] = c2 ? k[2] : (c3 ? key : k[3]);  j[3] = c2 ? j[2] : (c3 ? jpos : j[3]);
        k[2] = c1 ? k[1] : (c2 ? key : k[2]);  j[2] = c1 ? j[1] : (c2 ? jpos : j[2]);
        k[1] = c0 ? k[0] : (c1 ? key : k[1]);  j[1] = c0 ? j[0] : (c1 ? jpos : j[1]);
        k[0] = c0 ? key : k[0];                j[0] = c0 ? jpos : j[0];
    }
    __device__ __forceinline__ float worst() const { return __uint_as_float((unsigned)(k[4] >> 32)); }
    __device__ __forceinline__ bool final_tie() const { return false; }
    __device__ __forceinline__ unsigned worst_bits() const { return (unsigned)(k[4] >> 32); }
    __device__ __forceinline__ void to_top5(Top5& t) const {
#pragma unroll
        for (int s = 0; s < 5; s++) { t.d[s] = __uint_as_float((unsigned)(k[s] >> 32)); t.j[s] = j[s]; }
        t.have = false;
    }
};

__device__ __forceinline__ unsigned umed3(unsigned a, unsigned b, unsigned c) {   // no clang builtin for the integer median
    unsigned r;
    asm("v_med3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
// Key selector (default fast tier): ONE 32-bit key per candidate and no payload moves.
//   key = (distance bits & ~63) | code        code = ((chunk & 7) << 2) | slot-in-chunk      (a "bucket" = 64 f32 ulps of d^2)
// The SIX smallest keys are kept sorted with five v_med3_u32 and one v_min_u32 per candidate — no comparisons, no selects,
// no wave-level branch.  Where a key came from is recovered afterwards from a small per-lane LDS table
// (T[0..7] = first array position of the last eight chunks, T[8..13] = positions resolved so far, T[15] = -4 for the
// sentinel code 63): position = T[code >> 2] + (code & 3); every eight chunks the six held keys are re-pointed to T[8..13].
// Exactness: buckets are monotone in the distance, so every candidate outside the held six has a bucket >= the sixth's.
// If the sixth key's bucket differs from the fifth's, the five best are exactly the five smallest distances (as a set);
// their exact f32 distances are then recomputed from the re-loaded points and put into the oracle's (d2, index) order.
// If the buckets coincide, or the fifth lies in the bucket of the search bound, the query is repeated with the exact
// selector (about once in 1e5 queries on voxel-filtered maps; always on lattice ties).
struct Sel5K {
    unsigned k[6];
    int tc;          // chunks processed by this lane
    int* T;          // this lane's column of the chunk table (row stride ts ints)
    int ts;
    unsigned bb;     // bucket of the bound
    float bnd;
    __device__ __forceinline__ void init(float bound) {
        bnd = fminf(bound, 3.0e38f);
        bb = __float_as_uint(bnd) >> 6;
#pragma unroll
        for (int s = 0; s < 6; s++) k[s] = ((bb + 1u + (unsigned)s) << 6) | 63u;   // six distinct buckets above the bound
        tc = 0; T = nullptr; ts = 0;
    }
    __device__ __forceinline__ void attach(int* col, int stride) { T = col; ts = stride; T[15 * stride] = -4; }
    __device__ __forceinline__ float worst() const { return __uint_as_float(k[4] | 63u); }   // upper end of the bucket: pruning stays conservative
    __device__ __forceinline__ unsigned worst_bits() const { return k[4] | 63u; }
    __device__ __forceinline__ void to_top5(Top5& t) const {   // only meaningful right after init (early exits)
#pragma unroll
        for (int s = 0; s < 5; s++) { t.d[s] = bnd; t.j[s] = -1; }
        t.have = false;
    }
    __device__ __forceinline__ void push(unsigned key) {
        const unsigned m5 = umed3(k[4], k[5], key), m4 = umed3(k[3], k[4], key), m3 = umed3(k[2], k[3], key);
        const unsigned m2 = umed3(k[1], k[2], key), m1 = umed3(k[0], k[1], key);
        k[0] = min(k[0], key); k[1] = m1; k[2] = m2; k[3] = m3; k[4] = m4; k[5] = m5;
    }
    __device__ __forceinline__ int where(unsigned key) const { const unsigned c = key & 63u; return T[(int)(c >> 2) * ts] + (int)(c & 3u); }
    __device__ __forceinline__ void repoint() {
        int jr[6];
#pragma unroll
        for (int s = 0; s < 6; s++) jr[s] = where(k[s]);
#pragma unroll
        for (int s = 0; s < 6; s++) { T[(8 + s) * ts] = jr[s]; k[s] = (k[s] & ~63u) | (unsigned)(32 + 4 * s); }
    }
    __device__ __forceinline__ void chunk(float4 p0, float4 p1, float4 p2, float4 p3, int j, int end, float qx, float qy, float qz) {
        if (tc >= 8 && (tc & 7) == 0) repoint();
        T[(tc & 7) * ts] = j;
        const unsigned code = (unsigned)(tc & 7) << 2;
        tc++;
        const unsigned u0 = __float_as_uint(dist2(p0, qx, qy, qz));
        const unsigned u1 = j + 1 < end ? __float_as_uint(dist2(p1, qx, qy, qz)) : 0x7f800000u;
        const unsigned u2 = j + 2 < end ? __float_as_uint(dist2(p2, qx, qy, qz)) : 0x7f800000u;
        const unsigned u3 = j + 3 < end ? __float_as_uint(dist2(p3, qx, qy, qz)) : 0x7f800000u;
        push((u0 & ~63u) | code); push((u1 & ~63u) | (code + 1u)); push((u2 & ~63u) | (code + 2u)); push((u3 & ~63u) | (code + 3u));
    }
    // Resolves the five best, recomputes their exact distances and orders them by (d2, original index).
    // Returns true if the query has to be repeated with the exact selector.
    __device__ __forceinline__ bool finish(const GridView& g, float qx, float qy, float qz, Top5& t) const {
        const bool redo = ((k[5] ^ k[4]) < 64u) || ((k[4] >> 6) == bb);
        unsigned long long e[5];
        int jr[5];
        float4 pp[5];
#pragma unroll
        for (int s = 0; s < 5; s++) {
            jr[s] = where(k[s]);
            const bool real = jr[s] >= 0;
            const float4 p = g.pts[real ? jr[s] : 0];
            pp[s] = p;
            const unsigned du = real ? __float_as_uint(dist2(p, qx, qy, qz)) : __float_as_uint(bnd);
            const unsigned lo = real ? (unsigned)__float_as_int(p.w) : 0x7fffffffu;
            e[s] = ((unsigned long long)du << 32) | lo;
        }
        const bool unsorted = !(e[0] <= e[1] && e[1] <= e[2] && e[2] <= e[3] && e[3] <= e[4]);   // real keys are unique (distinct indices); equal keys are sentinels
        if (__any(unsorted)) {   // within-bucket inversion somewhere in the wave (rare): 9 compare-exchanges
#define LILI_CE(a, b) { const bool sw = e[b] < e[a]; const unsigned long long ea = e[a], eb = e[b]; const int ja = jr[a], jb = jr[b]; \
                        const float4 pa_ = pp[a], pb_ = pp[b]; \
                        e[a] = sw ? eb : ea; e[b] = sw ? ea : eb; jr[a] = sw ? jb : ja; jr[b] = sw ? ja : jb; \
                        pp[a].x = sw ? pb_.x : pa_.x; pp[a].y = sw ? pb_.y : pa_.y; pp[a].z = sw ? pb_.z : pa_.z; pp[a].w = sw ? pb_.w : pa_.w; \
                        pp[b].x = sw ? pa_.x : pb_.x; pp[b].y = sw ? pa_.y : pb_.y; pp[b].z = sw ? pa_.z : pb_.z; pp[b].w = sw ? pa_.w : pb_.w; }
            LILI_CE(0, 1) LILI_CE(3, 4) LILI_CE(2, 4) LILI_CE(2, 3) LILI_CE(0, 3) LILI_CE(0, 2) LILI_CE(1, 4) LILI_CE(1, 3) LILI_CE(1, 2)
#undef LILI_CE
        }
#pragma unroll
        for (int s = 0; s < 5; s++) { t.d[s] = __uint_as_float((unsigned)(e[s] >> 32)); t.j[s] = jr[s]; t.p[s] = pp[s]; }
        t.aux = tc;
        t.have = !redo;
        return redo;
    }
};
__device__ __forceinline__ void process_chunk(Sel5K& sel, float4 p0, float4 p1, float4 p2, float4 p3, int j, int end, float qx, float qy, float qz) {
    asm volatile("" : "+v"(p0.w), "+v"(p1.w), "+v"(p2.w), "+v"(p3.w));
    sel.chunk(p0, p1, p2, p3, j, end, qx, qy, qz);
}

// Key selector WITH a payload pool (round 6; the dense-map association k_associate_fine): the same 32-bit keys and the same five-median network as
// Sel5K, but the code of a key names one of SIX slots of a per-lane pool in LDS that holds the candidate itself (point + array position).  A
// candidate that enters the six held keys takes the slot — and the code — of the key it pushes out, so it enters iff its BUCKET is smaller than the sixth's (a candidate in
// the sixth's bucket stays out: everything outside the held six then has a bucket >= the sixth's, which is all the exactness argument needs).  Nothing is re-loaded from
// the map when the walk ends.  Why: on a map whose index does not fit L2 + Infinity Cache (5 M points at 0.05 m: 1.1 GB of super-rows) the lines of a lane's run have left
// the caches by the time Sel5K::finish gathers the five winners again — 7.6 of 46 us per 200 k-query launch (profiles/r06_2B_experiments.md).  Exactness as Sel5K:
// buckets are monotone in the distance; a tie of the fifth and sixth bucket, or a fifth in the bucket of the bound, reports `redo`.
struct Sel5P {
    unsigned k[6];
    unsigned bb;     // bucket of the bound
    float bnd;
    float4* P;       // this lane's column of the pool: slot s at P[s * STRIDE]
    int* J;
    int tc;
    static constexpr int STRIDE = 64;      // lanes per pool row (one wave per workgroup)
    __device__ __forceinline__ void init(float bound, float4* pcol, int* jcol) {
        bnd = fminf(bound, 3.0e38f);
        bb = __float_as_uint(bnd) >> 6;
        P = pcol; J = jcol;
#pragma unroll
        for (int s = 0; s < 6; s++) { k[s] = ((bb + 1u + (unsigned)s) << 6) | (unsigned)s; J[s * STRIDE] = -1; }   // six distinct buckets above the bound, slots without a candidate
        tc = 0;
    }
    __device__ __forceinline__ float worst() const { return __uint_as_float(k[4] | 63u); }
    __device__ __forceinline__ void push(unsigned u, float4 p, int j) {
        const unsigned old5 = k[5];
        const unsigned slot = old5 & 63u;                    // the slot that becomes free if this candidate enters
        const unsigned key = (u & ~63u) | slot;
        if (key < old5) { P[slot * STRIDE] = p; J[slot * STRIDE] = j; }      // same code: the comparison is one of the buckets
        const unsigned m5 = umed3(k[4], k[5], key), m4 = umed3(k[3], k[4], key), m3 = umed3(k[2], k[3], key);
        const unsigned m2 = umed3(k[1], k[2], key), m1 = umed3(k[0], k[1], key);
        k[0] = min(k[0], key); k[1] = m1; k[2] = m2; k[3] = m3; k[4] = m4; k[5] = m5;
    }
    // four consecutive candidates [j, j + 4) of a run ending at `end`
    __device__ __forceinline__ void chunk(float4 p0, float4 p1, float4 p2, float4 p3, int j, int end, float qx, float qy, float qz) {
        tc++;
        const unsigned u0 = j < end ? __float_as_uint(dist2(p0, qx, qy, qz)) : 0x7f800000u;
        const unsigned u1 = j + 1 < end ? __float_as_uint(dist2(p1, qx, qy, qz)) : 0x7f800000u;
        const unsigned u2 = j + 2 < end ? __float_as_uint(dist2(p2, qx, qy, qz)) : 0x7f800000u;
        const unsigned u3 = j + 3 < end ? __float_as_uint(dist2(p3, qx, qy, qz)) : 0x7f800000u;
        push(u0, p0, j); push(u1, p1, j + 1); push(u2, p2, j + 2); push(u3, p3, j + 3);
    }
    // The five best with exact distances in the oracle's (d2, original index) order.  Returns true if the query has to be repeated with the exact selector.
    __device__ __forceinline__ bool finish(float qx, float qy, float qz, Top5& t) const {
        const bool redo = ((k[5] ^ k[4]) < 64u) || ((k[4] >> 6) == bb);
        unsigned long long e[5];
        int jr[5];
        float4 pp[5];
#pragma unroll
        for (int s = 0; s < 5; s++) {
            const unsigned c = k[s] & 63u;
            jr[s] = J[c * STRIDE];
            const bool real = jr[s] >= 0;
            const float4 p = P[c * STRIDE];
            pp[s] = p;
            const unsigned du = real ? __float_as_uint(dist2(p, qx, qy, qz)) : __float_as_uint(bnd);
            const unsigned lo = real ? (unsigned)__float_as_int(p.w) : 0x7fffffffu;
            e[s] = ((unsigned long long)du << 32) | lo;
        }
        const bool unsorted = !(e[0] <= e[1] && e[1] <= e[2] && e[2] <= e[3] && e[3] <= e[4]);
        if (__any(unsorted)) {
#define LILI_CE(a, b) { const bool sw = e[b] < e[a]; const unsigned long long ea = e[a], eb = e[b]; const int ja = jr[a], jb = jr[b]; \
                        const float4 pa_ = pp[a], pb_ = pp[b]; \
                        e[a] = sw ? eb : ea; e[b] = sw ? ea : eb; jr[a] = sw ? jb : ja; jr[b] = sw ? ja : jb; \
                        pp[a].x = sw ? pb_.x : pa_.x; pp[a].y = sw ? pb_.y : pa_.y; pp[a].z = sw ? pb_.z : pa_.z; pp[a].w = sw ? pb_.w : pa_.w; \
                        pp[b].x = sw ? pa_.x : pb_.x; pp[b].y = sw ? pa_.y : pb_.y; pp[b].z = sw ? pa_.z : pb_.z; pp[b].w = sw ? pa_.w : pb_.w; }
            LILI_CE(0, 1) LILI_CE(3, 4) LILI_CE(2, 4) LILI_CE(2, 3) LILI_CE(0, 3) LILI_CE(0, 2) LILI_CE(1, 4) LILI_CE(1, 3) LILI_CE(1, 2)
#undef LILI_CE
        }
#pragma unroll
        for (int s = 0; s < 5; s++) { t.d[s] = __uint_as_float((unsigned)(e[s] >> 32)); t.j[s] = jr[s]; t.p[s] = pp[s]; }
        t.aux = tc;
        t.have = !redo;
        return redo;
    }
};

// Lower bound (conservative by 0.1 %) of the f32 squared distance from the query to any point of the cell row
// (cy+dy, cz+dz): the gap to the own cell's boundary in y and z.  Rows whose bound exceeds the current 5th best
// cannot contribute (a candidate enters only with d <= that value) — skipping them keeps the search exact.
__device__ __forceinline__ float row_lower_bound(const GridView& g, float qy, float qz, int cy, int cz, int dy, int dz) {
    const double c = g.cell;
    double gy = dy == 0 ? 0.0 : (dy < 0 ? (double)qy - (g.oy + (double)cy * c) : (g.oy + (double)(cy + 1) * c) - (double)qy);
    double gz = dz == 0 ? 0.0 : (dz < 0 ? (double)qz - (g.oz + (double)cz * c) : (g.oz + (double)(cz + 1) * c) - (double)qz);
    gy = fmax(gy, 0.0); gz = fmax(gz, 0.0);
    return (float)(0.999 * (gy * gy + gz * gz));
}
// visiting order of the 9 (dy,dz) rows: centre, faces, diagonals (w = (dy+1)*3 + (dz+1))
__device__ __forceinline__ int row_order(int n) { return n == 0 ? 4 : n == 1 ? 1 : n == 2 ? 3 : n == 3 ? 5 : n == 4 ? 7 : n == 5 ? 0 : n == 6 ? 2 : n == 7 ? 6 : 8; }

// One run of consecutive cell-sorted map points.  Four independent loads are in flight per trip (the search is bound
// by the length of its dependent-load chain, not by bandwidth); lanes past their run end re-load the run's last point
// and give it a NaN distance, whose key can never enter the selection.
__device__ __forceinline__ float4 load_pt(const GridView& g, int j) {   // 32-bit byte offset from the uniform base (map < 2^28 points)
    return *(const float4*)((const char*)g.pts + ((unsigned)j << 4));
}
// Four consecutive candidates [j, j+4) of a run ending at `end` (slots past the end were loaded from the run's last point
// and get +inf, which no selector accepts; NaN distances of non-finite map points likewise: fminf).
template <class SEL>
__device__ __forceinline__ void process_chunk(SEL& sel, float4 p0, float4 p1, float4 p2, float4 p3, int j, int end, float qx, float qy, float qz) {
    const int last = end - 1;
    asm volatile("" : "+v"(p0.w), "+v"(p1.w), "+v"(p2.w), "+v"(p3.w));   // keep each point ONE 16-byte load (no re-load of .w inside the branches)
    // distances as bit patterns (non-negative floats order like unsigned integers; a NaN distance — non-finite map point —
    // sorts above every bound and is never accepted); slots past the run end get +inf
    const unsigned u0 = __float_as_uint(dist2(p0, qx, qy, qz));
    const unsigned u1 = j + 1 < end ? __float_as_uint(dist2(p1, qx, qy, qz)) : 0x7f800000u;
    const unsigned u2 = j + 2 < end ? __float_as_uint(dist2(p2, qx, qy, qz)) : 0x7f800000u;
    const unsigned u3 = j + 3 < end ? __float_as_uint(dist2(p3, qx, qy, qz)) : 0x7f800000u;
    // each test is a wave-level skip of the selection code (taken if any lane qualifies)
    if (u0 <= sel.worst_bits()) sel.insert(__uint_as_float(u0), p0, j);
    if (u1 <= sel.worst_bits()) sel.insert(__uint_as_float(u1), p1, min(j + 1, last));
    if (u2 <= sel.worst_bits()) sel.insert(__uint_as_float(u2), p2, min(j + 2, last));
    if (u3 <= sel.worst_bits()) sel.insert(__uint_as_float(u3), p3, min(j + 3, last));
}
// One run of consecutive cell-sorted map points, four independent loads in flight per trip (shell phase).
template <class SEL>
__device__ __forceinline__ void scan_run(const GridView& g, SEL& sel, int beg, int end, float qx, float qy, float qz) {
    const int last = end - 1;
    for (int j = beg; j < end; j += 4) {
        float4 p0 = load_pt(g, j), p1 = load_pt(g, min(j + 1, last)), p2 = load_pt(g, min(j + 2, last)), p3 = load_pt(g, min(j + 3, last));
        process_chunk(sel, p0, p1, p2, p3, j, end, qx, qy, qz);
    }
}

// Per-thread table of the non-empty rows of the inner 3x3 block (LDS, one column per thread): run begin / end in the
// cell-sorted array and the row's lower distance bound.
template <int BS>
struct RowTabT {
    int b[9][BS];
    int e[9][BS];
    float lb[9][BS];
    int cj[16][BS];   // Sel5K: array positions of the last eight chunks / resolved positions / sentinel
};

// Exact 5-NN among the map points of the (2*reach+1)^3 cells around the query.  reach = 1: the 27 cells (9 runs).
// reach = 2 (cells of half the size): the inner 27 cells first — about 2.4x fewer candidates than 27 full-size
// cells — and the outer shell of the 5x5x5 block only if the current 5th-best distance does not rule it out:
// every point outside the inner block is at least `margin` away (margin = one cell + the query's gap to the nearest
// face of its own cell), so worst < 0.999 * margin^2 makes the shell irrelevant.  All bounds are conservative by
// 0.1 % against f32 rounding of the distances; ties (d == worst) never skip.
// Profiling aid, compiled only with -DLILI_PHASE_PROBE (tools/assoc_phases.sh builds such a copy of the library; the product
// build has none of it): s_memrealtime stamps (100 MHz) of one wave at the phase boundaries of the association.  `dep` is a value
// the phase produced, so that the stamp cannot be taken before it exists.
struct PhaseProbe { long long t[8]; };
#ifdef LILI_PHASE_PROBE
#define PHASE_STAMP(pp, k, dep) do { if (pp) { long long t_; asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) : "v"(dep) : "memory"); (pp)->t[k] = t_; } } while (0)
#else
#define PHASE_STAMP(pp, k, dep) do {} while (0)
#endif
__device__ __forceinline__ float gate_bound(double gate) {   // smallest f32 >= gate
    float gf = (float)gate;
    if ((double)gf < gate) gf = __uint_as_float(__float_as_uint(gf) + 1u);
    return gf;
}
// the 16 (dy, dz) rows of the 5x5x5 shell, nearest first (faces, then the rows next to them, then the corners), two batches of eight
__device__ constexpr int kShellDy[16] = {0, 0, -2, 2, -1, 1, -1, 1, -2, -2, 2, 2, -2, -2, 2, 2};
__device__ constexpr int kShellDz[16] = {-2, 2, 0, 0, -2, -2, 2, 2, -1, 1, -1, 1, -2, 2, -2, 2};
template <class SEL, class TAB>
__device__ __forceinline__ bool knn5_grid_sel(const GridView& g, TAB& tab, float qx, float qy, float qz, float bound, Top5& best, PhaseProbe* pp = nullptr) {
    SEL sel; sel.init(bound);
    sel.to_top5(best);
    if (!(isfinite(qx) && isfinite(qy) && isfinite(qz))) return false;
    if constexpr (std::is_same<SEL, Sel5K>::value) sel.attach(&tab.cj[0][threadIdx.x], (int)(sizeof(tab.cj[0]) / sizeof(int)));
    const int R = g.reach;
    int cx = cell_coord(qx, g.ox, g.inv_cell), cy = cell_coord(qy, g.oy, g.inv_cell), cz = cell_coord(qz, g.oz, g.inv_cell);
    // queries more than `reach` cells outside the grid cannot have a neighbour within the gate radius
    if (cx < -R || cx > g.nx - 1 + R || cy < -R || cy > g.ny - 1 + R || cz < -R || cz > g.nz - 1 + R) return false;
    // the query's inner block lies in the box that has super-rows (x0 > x1: nothing to search either way)
    const bool inner9 = g.cell_start9 && cy >= g.by0 && cy < g.by0 + g.bny && cz >= g.bz0 && cz < g.bz0 + g.bnz &&
                        max(cx - 1, 0) >= g.bx0 && min(cx + 1, g.nx - 1) < g.bx0 + g.bnx;
    if (inner9) {
        // Super-row layout: the inner 27 cells are ONE run of the unified array (two range words, full chunks, no row table).
        const int x0 = max(cx - 1, 0), x1 = min(cx + 1, g.nx - 1);
        if (x0 <= x1) {
            const int* row = g.cell_start9 + srow_index(g, g.bx0, cy, cz) - g.bx0;
            int cj = row[x0];
            const int ce = row[x1 + 1];
            PHASE_STAMP(pp, 2, ce);
            auto fetch = [&](float4& p0, float4& p1, float4& p2, float4& p3, int& pj) {
                // unconditional: no branch around the loads, so the waits the compiler inserts are exact.  Slots past the run's end — also the
                // whole chunk requested after the last one — read the following entries (the array has 8 entries of slack) and are masked by
                // position / never processed.
                pj = cj;
                const float4* q = (const float4*)((const char*)g.pts + ((unsigned)cj << 4));
                p0 = q[0]; p1 = q[1]; p2 = q[2]; p3 = q[3];
                cj += 4;
            };
            float4 a0, a1, a2, a3, b0, b1, b2, b3;
            int aj = 0, bj = 0;
            fetch(a0, a1, a2, a3, aj);
            for (;;) {
                if (!(aj < ce)) break;
                fetch(b0, b1, b2, b3, bj);
                process_chunk(sel, a0, a1, a2, a3, aj, ce, qx, qy, qz);
                if (!(bj < ce)) break;
                fetch(a0, a1, a2, a3, aj);
                process_chunk(sel, b0, b1, b2, b3, bj, ce, qx, qy, qz);
            }
        }
    } else {
        const int x0 = max(cx - 1, 0), x1 = min(cx + 1, g.nx - 1);
        if (x0 <= x1) {
            // All nine row ranges are fetched at once (18 independent loads); the non-empty rows go to this thread's
            // column of the LDS table in visiting order (centre, faces, diagonals) with their lower bounds.
            int rb[9], re[9];
#pragma unroll
            for (int n = 0; n < 9; n++) {
                const int w = row_order(n);
                const int y = cy + w / 3 - 1, z = cz + w % 3 - 1;
                const bool in = z >= 0 && z < g.nz && y >= 0 && y < g.ny;
                const int* cs = g.cell_start + (size_t)(min(max(z, 0), g.nz - 1) * g.ny + min(max(y, 0), g.ny - 1)) * g.nx;
                const int b = cs[x0], e = cs[x1 + 1];
                rb[n] = b; re[n] = in ? e : b;
            }
            const int tid = threadIdx.x;
            int cnt = 0;
#pragma unroll
            for (int n = 0; n < 9; n++) {
                const int w = row_order(n);
                if (rb[n] < re[n]) {
                    tab.b[cnt][tid] = rb[n]; tab.e[cnt][tid] = re[n];
                    tab.lb[cnt][tid] = row_lower_bound(g, qy, qz, cy, cz, w / 3 - 1, w % 3 - 1);
                    cnt++;
                }
            }
            // Each lane walks ITS OWN rows (a wave takes max-over-lanes of the summed trips, not the sum of per-row maxima)
            // and the four loads of the next chunk are issued before the current chunk is processed.  The row for the
            // next chunk is chosen with the 5th-best distance of one chunk ago: a stale (larger) value can only keep a
            // row that the fresh one would prune — extra candidates, never a missing one.
            PHASE_STAMP(pp, 2, cnt);                                        // row ranges loaded, table written
            int n = 0, cj = 0, ce = 0;
            auto fetch = [&](float wv, float4& p0, float4& p1, float4& p2, float4& p3, int& pj, int& pe) {
                while (cj >= ce && n < cnt) {
                    const int b = tab.b[n][tid], e = tab.e[n][tid];
                    const float lb = tab.lb[n][tid];
                    n++;
                    if (!(lb > wv)) { cj = b; ce = e; }
                }
                pj = cj; pe = ce;
                if (cj < ce) {
                    const int last = ce - 1;
                    p0 = load_pt(g, cj); p1 = load_pt(g, min(cj + 1, last)); p2 = load_pt(g, min(cj + 2, last)); p3 = load_pt(g, min(cj + 3, last));
                    cj += 4;
                }
            };
            // two chunk buffers in ping-pong, so that the loads of one are in flight while the other is processed
            float4 a0, a1, a2, a3, b0, b1, b2, b3;
            int aj = 0, ae = 0, bj = 0, be = 0;
            fetch(sel.worst(), a0, a1, a2, a3, aj, ae);
            for (;;) {
                if (!(aj < ae)) break;
                fetch(sel.worst(), b0, b1, b2, b3, bj, be);
                process_chunk(sel, a0, a1, a2, a3, aj, ae, qx, qy, qz);
                if (!(bj < be)) break;
                fetch(sel.worst(), a0, a1, a2, a3, aj, ae);
                process_chunk(sel, b0, b1, b2, b3, bj, be, qx, qy, qz);
            }
        }
    }
    PHASE_STAMP(pp, 3, sel.worst());                                        // inner 3x3x3 block walked
    if (R == 2) {
        const double c = g.cell;
        const double fxm = (double)qx - (g.ox + (double)cx * c), fxp = (g.ox + (double)(cx + 1) * c) - (double)qx;
        const double fym = (double)qy - (g.oy + (double)cy * c), fyp = (g.oy + (double)(cy + 1) * c) - (double)qy;
        const double fzm = (double)qz - (g.oz + (double)cz * c), fzp = (g.oz + (double)(cz + 1) * c) - (double)qz;
        const double margin = c + fmax(fmin(fmin(fmin(fxm, fxp), fmin(fym, fyp)), fmin(fzm, fzp)), 0.0);
        if (!(sel.worst() < (float)(0.999 * margin * margin))) {
            // super-row layout: the 18 single-cell runs x = cx -+ 2 of the nine inner rows are two runs (one super cell each)
            const bool side9 = inner9 && (cx - 2 < 0 || cx - 2 >= g.bx0) && (cx + 2 >= g.nx || cx + 2 < g.bx0 + g.bnx);
            if (side9) {
                const int* row = g.cell_start9 + srow_index(g, g.bx0, cy, cz) - g.bx0;
                const int xl = cx - 2, xr = cx + 2;
                if (xl >= 0 && xl < g.nx) { const double gx = fmax(fxm + c, 0.0); if (!((float)(0.999 * gx * gx) > sel.worst())) scan_run(g, sel, row[xl], row[xl + 1], qx, qy, qz); }
                if (xr >= 0 && xr < g.nx) { const double gx = fmax(fxp + c, 0.0); if (!((float)(0.999 * gx * gx) > sel.worst())) scan_run(g, sel, row[xr], row[xr + 1], qx, qy, qz); }
            }
            // How many lanes of the wave are here?  A handful (a converged pose: one lane in a few waves) is bound by the dependent round trips
            // of the 16 shell rows — the batched form below; many (the first iterations of a registration) are bound by instruction issue on
            // mostly idle lanes, where the row-by-row walk with its progressive pruning and x-trimming does less work.
            const bool few = __popcll(__ballot(1)) <= 2;
            if (side9 && few) {
                // The 16 rows of the shell, nearest first, in two batches of eight: the range words of a batch are requested TOGETHER (one
                // round trip instead of eight dependent ones; pruned and x-trimmed with the 5th best at that moment), parked in the lane's
                // columns of the row table (idle in this layout), and only the non-empty rows that still matter are scanned.  The second
                // batch sees the 5th best the first one left.
                const double g1m = fmax(fxm, 0.0), g1p = fmax(fxp, 0.0), g2m = fmax(fxm + c, 0.0), g2p = fmax(fxp + c, 0.0);
                const int tid = threadIdx.x;
#pragma unroll
                for (int batch = 0; batch < 2; batch++) {
                    const float wv = sel.worst();
                    int rb[8], re[8]; float rl[8];
#pragma unroll
                    for (int i = 0; i < 8; i++) {
                        const int dy = kShellDy[batch * 8 + i], dz = kShellDz[batch * 8 + i];
                        const int y = cy + dy, z = cz + dz;
                        const double gy = dy == 0 ? 0.0 : fmax(dy < 0 ? fym + (double)(-dy - 1) * c : fyp + (double)(dy - 1) * c, 0.0);
                        const double gz = dz == 0 ? 0.0 : fmax(dz < 0 ? fzm + (double)(-dz - 1) * c : fzp + (double)(dz - 1) * c, 0.0);
                        const double lbr = 0.999 * (gy * gy + gz * gz);
                        const int dl = (float)(lbr + 0.999 * g2m * g2m) > wv ? ((float)(lbr + 0.999 * g1m * g1m) > wv ? 0 : 1) : 2;
                        const int dr = (float)(lbr + 0.999 * g2p * g2p) > wv ? ((float)(lbr + 0.999 * g1p * g1p) > wv ? 0 : 1) : 2;
                        const int x0 = max(cx - dl, 0), x1 = min(cx + dr, g.nx - 1);
                        const bool keep = y >= 0 && y < g.ny && z >= 0 && z < g.nz && !((float)lbr > wv) && x0 <= x1;
                        const int* cs = g.cell_start + (size_t)(keep ? z * g.ny + y : 0) * g.nx;
                        const int b = cs[keep ? x0 : 0], e = cs[keep ? x1 + 1 : 0];
                        rb[i] = b; re[i] = keep ? e : b; rl[i] = (float)lbr;
                    }
#pragma unroll
                    for (int i = 0; i < 8; i++) { tab.b[i][tid] = rb[i]; tab.e[i][tid] = re[i]; tab.lb[i][tid] = rl[i]; }
                    for (int i = 0; i < 8; i++) {
                        const int b = tab.b[i][tid], e = tab.e[i][tid];
                        if (b < e && !(tab.lb[i][tid] > sel.worst())) scan_run(g, sel, b, e, qx, qy, qz);
                    }
                }
            } else
            for (int dz = -2; dz <= 2; dz++) {
                const int z = cz + dz;
                if (z < 0 || z >= g.nz) continue;
                const double gz = dz == 0 ? 0.0 : fmax(dz < 0 ? fzm + (double)(-dz - 1) * c : fzp + (double)(dz - 1) * c, 0.0);
                for (int dy = -2; dy <= 2; dy++) {
                    const int y = cy + dy;
                    if (y < 0 || y >= g.ny) continue;
                    const double gy = dy == 0 ? 0.0 : fmax(dy < 0 ? fym + (double)(-dy - 1) * c : fyp + (double)(dy - 1) * c, 0.0);
                    const double lbr = 0.999 * (gy * gy + gz * gz);
                    if ((float)lbr > sel.worst()) continue;
                    const int* cs = g.cell_start + (size_t)(z * g.ny + y) * g.nx;
                    if (dy == -2 || dy == 2 || dz == -2 || dz == 2) {            // a row of the shell: up to 5 cells,
                        // trimmed to the cells whose box distance (row gap + x gap) can still beat the 5th best
                        const float wv = sel.worst();
                        const double g1m = fmax(fxm, 0.0), g1p = fmax(fxp, 0.0), g2m = fmax(fxm + c, 0.0), g2p = fmax(fxp + c, 0.0);
                        const int dl = (float)(lbr + 0.999 * g2m * g2m) > wv ? ((float)(lbr + 0.999 * g1m * g1m) > wv ? 0 : 1) : 2;
                        const int dr = (float)(lbr + 0.999 * g2p * g2p) > wv ? ((float)(lbr + 0.999 * g1p * g1p) > wv ? 0 : 1) : 2;
                        const int x0 = max(cx - dl, 0), x1 = min(cx + dr, g.nx - 1);
                        if (x0 <= x1) scan_run(g, sel, cs[x0], cs[x1 + 1], qx, qy, qz);
                    } else if (!side9) {                                         // inner row: only its two outer cells are new
                        const int xl = cx - 2, xr = cx + 2;
                        if (xl >= 0 && xl < g.nx) { double gx = fmax(fxm + c, 0.0); if (!((float)(lbr + 0.999 * gx * gx) > sel.worst())) scan_run(g, sel, cs[xl], cs[xl + 1], qx, qy, qz); }
                        if (xr >= 0 && xr < g.nx) { double gx = fmax(fxp + c, 0.0); if (!((float)(lbr + 0.999 * gx * gx) > sel.worst())) scan_run(g, sel, cs[xr], cs[xr + 1], qx, qy, qz); }
                    }
                }
            }
        }
    }
    PHASE_STAMP(pp, 4, sel.worst());                                        // shell decided / walked
    if constexpr (std::is_same<SEL, Sel5K>::value) return sel.finish(g, qx, qy, qz, best);
    else { sel.to_top5(best); return sel.final_tie(); }
}
// Fast selection first; the rare queries with an exact distance tie that could matter are repeated with the exact
// (distance, original index) selector, so the result is always the oracle's.
template <class TAB>
__device__ __forceinline__ void knn5_grid(const GridView& g, TAB& tab, float qx, float qy, float qz, float bound, Top5& best, int dbg = 0, PhaseProbe* pp = nullptr) {
    if (dbg & 32768) { knn5_grid_sel<Sel5>(g, tab, qx, qy, qz, bound, best); return; }   // A/B: exact selector only
    const bool redo = knn5_grid_sel<Sel5K>(g, tab, qx, qy, qz, bound, best, pp);
    PHASE_STAMP(pp, 5, best.d[4]);                                          // five winners resolved (exact distances, order)
    if (redo) knn5_grid_sel<Sel5>(g, tab, qx, qy, qz, bound, best);
}

// Correspondence counting without atomics on a shared word (3128 same-address atomics cost ~40 us on
// MI355X): each block stores its own count; consumers add the <= few-thousand block counts themselves.
template <int BS>
__device__ __forceinline__ void store_block_count(bool ok, int* __restrict__ block_counts, int bid) {
    __shared__ int wave_cnt[BS / 64];
    unsigned long long bal = __ballot(ok);
    if ((threadIdx.x & 63) == 0) wave_cnt[threadIdx.x >> 6] = __popcll(bal);
    __syncthreads();
    if (threadIdx.x == 0) {
        int s = 0;
#pragma unroll
        for (int w = 0; w < BS / 64; w++) s += wave_cnt[w];
        block_counts[bid] = s;
    }
}
// Sum of the per-block counts of one association launch (every thread of the block gets the total).  In two steps so that a caller can put work
// between the loads and the barriers: sum_block_counts_begin returns this wave's sum (loads + shuffles), sum_block_counts_end the block's.
__device__ __forceinline__ int sum_block_counts_begin(const int* __restrict__ block_counts, int nb) {
    int s = 0;
    // four independent loads per trip (the plain strided loop serialises one L2 round trip per element)
    const int bd = blockDim.x;
    for (int b0 = threadIdx.x; b0 < nb; b0 += 4 * bd) {
        const int b1 = b0 + bd, b2 = b0 + 2 * bd, b3 = b0 + 3 * bd;
        const int v0 = block_counts[b0], v1 = b1 < nb ? block_counts[b1] : 0, v2 = b2 < nb ? block_counts[b2] : 0, v3 = b3 < nb ? block_counts[b3] : 0;
        s += (v0 + v1) + (v2 + v3);
    }
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    return s;
}
__device__ __forceinline__ int sum_block_counts_end(int wave_sum) {
    __shared__ int part[16];
    __shared__ int total;
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = wave_sum;
    __syncthreads();
    if (threadIdx.x == 0) { int t = 0; for (int w = 0; w < (int)(blockDim.x >> 6); w++) t += part[w]; total = t; }
    __syncthreads();
    return total;
}
__device__ __forceinline__ int sum_block_counts(const int* __restrict__ block_counts, int nb) { return sum_block_counts_end(sum_block_counts_begin(block_counts, nb)); }

// 16-byte granule {value, value ^ key}: one write-through store / two relaxed agent-scope (sc1) loads
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ unsigned long long launch_key(unsigned long long epoch) { return (epoch + 1ull) * 0x9E3779B97F4A7C15ull; }   // never 0 for epoch < 2^64 - 1
__device__ __forceinline__ void store_granule(double* g, double v, unsigned long long key) {
    const unsigned long long lo = (unsigned long long)__double_as_longlong(v), hi = lo ^ key;
    const u32x4 d = {(unsigned)lo, (unsigned)(lo >> 32), (unsigned)hi, (unsigned)(hi >> 32)};
    asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(g), "v"(d) : "memory");
}
__device__ __forceinline__ void load_granule(const double* g, unsigned long long& lo, unsigned long long& hi) {
    const unsigned long long* p = reinterpret_cast<const unsigned long long*>(g);
    lo = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    hi = __hip_atomic_load(p + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// The body pose of a slot as the kernel in front published it (gn_update_block, `pub_key`): lanes 0..6 of the wave read one granule each of the workgroup's copy — ONE
// coalesced 112-byte request per poll, agent scope — until all seven carry the key.  The wait is bounded; a wave that gives up raises SlotState::wait_failed (sticky;
// lili_s2m_pose_get reports it) and carries on with what the slot holds.
__device__ __forceinline__ void wait_published_pose(const SlotState* st, const double* pub, unsigned long long key, double pose[7]) {
    const int lane = threadIdx.x & 63;
    const double* g = pub + (size_t)(blockIdx.x & (kPubReplicas - 1)) * kPubStride + 2 * (lane < 7 ? lane : 0);
    unsigned long long lo = 0ull, hi = 0ull;
    bool ok = false;
    for (unsigned sweep = 0; sweep < (1u << 21); sweep++) {
        u32x4 d;
        asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(d) : "v"(g) : "memory");
        lo = ((unsigned long long)d.y << 32) | d.x; hi = ((unsigned long long)d.w << 32) | d.z;
        ok = lane >= 7 || ((lo ^ hi) == key);
        if (__all(ok)) break;
        __builtin_amdgcn_s_sleep(2);
    }
    if (!__all(ok)) {
        if (lane == 0) const_cast<SlotState*>(st)->wait_failed = 1ull;
        if (lane < 7) lo = (unsigned long long)__double_as_longlong(st->pose[lane]);
    }
#pragma unroll
    for (int k = 0; k < 7; k++) pose[k] = __longlong_as_double((long long)(((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(lo >> 32), k) << 32) |
                                                                            (unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)lo, k)));
}
__device__ __forceinline__ void load_assoc_pose(const PoseArg& pa, const MatchParams& P, dq& Q2, d3& T2) {
    if (pa.state) {
        double pp[7];
        // profiling aid (LILI_DEBUG bit 256, bench.py LILI_PHASES): workgroup 0 stamps its start and the moment it holds the pose, and snapshots the stamps the
        // reduction + GN kernel in front of it left (its start / record built), which that kernel's next launch overwrites
        const bool probe = (P.debug & 256) && blockIdx.x == 0 && threadIdx.x == 0;
        if (probe) const_cast<SlotState*>(pa.state)->tprof[5] = (long long)__builtin_amdgcn_s_memrealtime();
#ifdef LILI_OVERLAP_GN      // experiment build only (profiles/EXPERIMENTS.md: no gain in situ, so the polling path stays out of the product kernels)
        if (pa.wait_key) wait_published_pose(pa.state, pa.pub, pa.wait_key, pp);
        else
#endif
        {
#pragma unroll
            for (int k = 0; k < 7; k++) pp[k] = pa.state->pose[k];
        }
        if (probe) {
            SlotState* w = const_cast<SlotState*>(pa.state);
            long long t_; asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) : "v"(pp[0]) : "memory");
            w->tprof[6] = t_; w->tprof[13] = w->tprof[8]; w->tprof[14] = w->tprof[10];
        }
        const double* s = pp;
        dq Q{s[3], s[4], s[5], s[6]};
        d3 T{s[0], s[1], s[2]};
        if (pa.derive_assoc) {   // L/src/BackendFusion.cpp:929-930
            Q2 = qmul(Q, dq{P.q_lb_inv[0], P.q_lb_inv[1], P.q_lb_inv[2], P.q_lb_inv[3]});
            T2 = T - qrot(Q2, d3{P.t_lb[0], P.t_lb[1], P.t_lb[2]});
        } else { Q2 = Q; T2 = T; }
    } else {
        Q2 = dq{pa.q[0], pa.q[1], pa.q[2], pa.q[3]};
        T2 = d3{pa.t[0], pa.t[1], pa.t[2]};
    }
}

// ================================================================================================
// K5 / K6 — association.  The per-query fits are shared by the tiled (LDS) and the direct search path.
// surf records: rec_nd[i] = (w*nx, w*ny, w*nz, w*normInverse) as floats, rec_score[i] (f64), valid[i]
// edge records: rec_a[i] = (Ax, Ay, Az, s), rec_b[i] = (Bx, By, Bz, 0), valid[i]
// ================================================================================================
__device__ __forceinline__ void store_debug_nn(const GridView& g, const Top5& nn, int i, int* __restrict__ dbg_idx, float* __restrict__ dbg_d2) {
    if (!dbg_idx) return;
#pragma unroll
    for (int k = 0; k < 5; k++) {
        dbg_idx[(size_t)i * 5 + k] = nn.j[k] >= 0 ? __float_as_int(g.pts[nn.j[k]].w) : -1;
        dbg_d2[(size_t)i * 5 + k] = nn.d[k];
    }
}

// findCorrespondingSurfFeatures body after the kNN (L/src/BackendFusion.cpp:1613-1679 and variants)
__device__ __forceinline__ bool surf_fit(const GridView& g, const MatchParams& P, const Top5& nn, float4 ql, float px, float py, float pz,
                                         float4& rn, double& score) {
    rn = make_float4(0.f, 0.f, 0.f, 0.f);
    score = 0.0;
    if (!(nn.j[4] >= 0 && (double)nn.d[4] < P.kd_max_radius)) return false;   // L:1615
    float4 m[5];
#pragma unroll
    for (int k = 0; k < 5; k++) m[k] = nn.p[k];
    if (__any(!nn.have)) {           // exact-selector / tiled / debug paths: the points were not handed over
#pragma unroll
        for (int k = 0; k < 5; k++) if (!nn.have) m[k] = g.pts[nn.j[k]];
    }
    double sum_w = 0.0;
    double mx[5], my[5], mz[5], wk[5];
#pragma unroll
    for (int k = 0; k < 5; k++) { mx[k] = (double)m[k].x; my[k] = (double)m[k].y; mz[k] = (double)m[k].z; wk[k] = 1.0; }
    if (P.variant == 0) {   // Livox reflectivity weighting, L:1617-1638
        double w[5];
#pragma unroll
        for (int k = 0; k < 5; k++) {
            float diff = ql.w - g.aux[nn.j[k]];
            double tmp_w = (double)fabsf(diff);
            sum_w += tmp_w;
            w[k] = 1.0 / tmp_w;
        }
        if (sum_w > P.reflect_thres) return false;
#pragma unroll
        for (int k = 0; k < 5; k++) wk[k] = w[k] / sum_w;
    }
    double nv[3];
    bool fitted = false;
    if (!(P.debug & 16384)) {   // LILI_DEBUG bit 16384: always take the pivoted QR (A/B and parity of the two paths)
        if (P.variant == 0) {
            double w2[5];
#pragma unroll
            for (int k = 0; k < 5; k++) w2[k] = wk[k] * wk[k];
            fitted = plane_fit_centered<true>(mx, my, mz, w2, nv, (P.debug & 2048) ? 0.0 : 1e-7);
        } else fitted = plane_fit_centered<false>(mx, my, mz, wk, nv, (P.debug & 2048) ? 0.0 : 1e-7);
    }
    if (!fitted) {            // ill-conditioned or rank-deficient: Eigen's rank-revealing procedure (rare, wave-divergent)
        col5 c0, c1, c2, b;
#pragma unroll
        for (int k = 0; k < 5; k++) { c0.v[k] = wk[k] * mx[k]; c1.v[k] = wk[k] * my[k]; c2.v[k] = wk[k] * mz[k]; b.v[k] = -1.0 * wk[k]; }
        lstsq53(c0, c1, c2, b, nv);
    }
    double nn_ = sqrt(nv[0] * nv[0] + nv[1] * nv[1] + nv[2] * nv[2]);
    double normInverse = 1.0 / nn_;
    nv[0] *= normInverse; nv[1] *= normInverse; nv[2] *= normInverse;
    bool planeValid = true;
#pragma unroll
    for (int k = 0; k < 5; k++) {
        if (fabs(nv[0] * (double)m[k].x + nv[1] * (double)m[k].y + nv[2] * (double)m[k].z + normInverse) > P.surf_dist_thres) planeValid = false;
    }
    if (!planeValid) return false;
    // L:1661-1662: float pd, float weight; sqrt(sqrt()) on a float argument is the float overload
    float pd = (float)(nv[0] * (double)px + nv[1] * (double)py + nv[2] * (double)pz + normInverse);
    float r2 = px * px + py * py + pz * pz;
    float weight = (float)(1.0 - 0.9 * (double)fabsf(pd) / (double)sqrtf(sqrtf(r2)));
    if (!((double)weight > P.surf_weight_min)) return false;
    rn.x = (float)((double)weight * nv[0]); rn.y = (float)((double)weight * nv[1]); rn.z = (float)((double)weight * nv[2]);
    rn.w = (float)((double)weight * normInverse);
    if (P.variant == 0) score = P.lidar_const * ((double)weight + exp(-sum_w));   // L:1676
    else if (P.variant == 1) score = P.lidar_const * (double)weight;                // R:1515
    else score = 1.0;
    return true;
}

// findCorrespondingCornerFeatures body after the kNN (L/src/BackendFusion.cpp:1543-1596, R:1404-1458)
__device__ __forceinline__ bool edge_fit(const GridView& g, const MatchParams& P, const Top5& nn, float px, float py, float pz,
                                         float4& ra, float4& rb) {
    ra = make_float4(0.f, 0.f, 0.f, 0.f); rb = ra;
    if (!(nn.j[4] >= 0 && (double)nn.d[4] < P.edge_gate)) return false;   // L:1543
    d3 m[5]; d3 c{0, 0, 0};
#pragma unroll
    for (int k = 0; k < 5; k++) { float4 p = nn.p[k]; if (!nn.have) p = g.pts[nn.j[k]]; m[k] = d3{(double)p.x, (double)p.y, (double)p.z}; c = c + m[k]; }
    c = d3{c.x / 5.0, c.y / 5.0, c.z / 5.0};
    double a00 = 0, a01 = 0, a02 = 0, a11 = 0, a12 = 0, a22 = 0;
#pragma unroll
    for (int k = 0; k < 5; k++) {
        d3 z = m[k] - c;
        a00 += z.x * z.x; a01 += z.x * z.y; a02 += z.x * z.z; a11 += z.y * z.y; a12 += z.y * z.z; a22 += z.z * z.z;
    }
    double ev[3]; d3 vmin, vmax;
    eig3_sym(a00, a01, a02, a11, a12, a22, ev, vmin, vmax);
    if (!(ev[2] > 3.0 * ev[1])) return false;   // L:1575
    d3 u = canon_sign(vmax);
    d3 A = c + 0.1 * u, B = c - 0.1 * u;
    if (P.edge_dist_max > 0) {   // R:1437-1443
        d3 lp{(double)px, (double)py, (double)pz};
        d3 nu = cross3(lp - A, lp - B);
        d3 de = A - B;
        double dist = sqrt(dot3(nu, nu)) / sqrt(dot3(de, de));
        if (!(dist < P.edge_dist_max)) return false;
    }
    ra = make_float4((float)A.x, (float)A.y, (float)A.z, (float)P.lidar_const);
    rb = make_float4((float)B.x, (float)B.y, (float)B.z, 0.f);
    return true;
}

// What a lane of the association found, handed on in registers to the launch that linearises on the fly (k_associate_lin): the values
// are the ROUNDED ones the record arrays receive, so the two-launch path sees the same numbers.
struct LaneRec { bool ok; float4 ql, r0, r1; double score; };

// ================================================================================================
// Linearisation: residual + 1x7 global Jacobian per record, loss corrector, Gram reduction.
//
// Reduction scheme (deterministic, no float atomics): every wave stages its 64 rows [J0..J6, r, cost]
// in LDS; lane l < 36 owns Gram entry (a,b) of the upper triangle and sums row[q][a]*row[q][b] over
// q = 0..63 in order; lane 36 sums the cost column.  Waves of a block are then added in order and the
// block writes one 40-double partial; k_reduce_gn adds the partials in a fixed order.
// ================================================================================================
constexpr int kRow = 12;          // LDS row: [J0..J6, r | 1, cost, 0, 0]
constexpr int kLinBlock = 1024;   // linearisation block (16 waves; the launch covers the queries with <= 256 blocks)
// Gram accumulation on the f64 matrix cores.  Per wave, G += V^T V over its 64 rows v = [a | b | e] with a = (J0..J3),
// b = (J4, J5, J6, r), e = (1, cost, 0, 0), issued as 16 x v_mfma_f64_4x4x4_4b_f64: ONE instruction contracts four rows (k)
// into four independent 4x4 blocks — block 0: a a^T, block 1: a b^T, block 2: b b^T, block 3: e e^T (count and cost sum) —
// i.e. exactly the 36 + 2 numbers of the upper triangle, where round 1's 16x16x4 form computed a 16x16 tile of which 55
// entries were used (measured on MI355X, tools/probe_mfma.hip: 64 clocks per 16x16x4 against 20 per 4x4x4_4b, and the
// 16 operand reads of a wave were issued one by one in front of their MFMA).  Lane map of the instruction (probed, same file):
// operand lane l feeds row i (A) / column j (B) = l & 3 of block (l >> 2) & 3 at k = l >> 4; result lane o holds
// D_block[o >> 4][o & 3] of block (o >> 2) & 3.  This is a reduction, not a GEMM re-shaping of the path; the summation
// order is fixed by the instruction sequence, so results stay deterministic.
// profiling aid: thread 0 of one probe block stamps the constant 100 MHz clock into SlotState::tprof[slot]
__device__ __forceinline__ void tstamp(const SlotState* state, int debug, int probe_block, int slot) {
    if ((debug & 256) && (int)blockIdx.x == probe_block && threadIdx.x == 0)
        const_cast<SlotState*>(state)->tprof[slot] = (long long)__builtin_amdgcn_s_memrealtime();
}
// result lane of partial entry e: e < 36 = upper triangle of the 8x8 Gram (row-major), 36 = cost, 37 = count, 38 / 39 = always zero
__device__ __forceinline__ int gram_lane(int e) {
    if (e == 36) return 16 + 12;        // block 3, [1][0] = sum cost * 1
    if (e == 37) return 12;             // block 3, [0][0] = sum 1 * 1
    if (e >= 38) return 2 * 16 + 12 + 2;   // block 3, [2][2] = 0
    int a = 0, l = e;
    while (l >= 8 - a) { l -= 8 - a; a++; }
    const int b = a + l;
    if (b < 4) return 16 * a + b;                       // a a^T
    if (a < 4) return 16 * a + 4 + (b - 4);             // a b^T
    return 16 * (a - 4) + 8 + (b - 4);                  // b b^T
}
struct GramAcc {
    double acc;
    __device__ __forceinline__ void init() { acc = 0.0; }
    // Every wave stages and consumes ITS OWN 64 rows, so only wave-level ordering is needed here (LDS operations of
    // one wave execute in order; the fences keep the compiler from moving them) — no block barrier: fast waves do
    // their MFMAs while slow ones still wait for their records.  All lanes of the wave must call this.
    __device__ __forceinline__ void add_rows(const double Jr[8], double cost, bool ok, double* lds) {
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        double* rows = lds + wave * 64 * kRow;
        double* myrow = rows + lane * kRow;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");   // the previous tile's reads are done before the rows are overwritten
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int k = 0; k < 8; k++) myrow[k] = ok ? Jr[k] : 0.0;
        myrow[8] = ok ? 1.0 : 0.0;
        myrow[9] = ok ? cost : 0.0;
        myrow[10] = 0.0; myrow[11] = 0.0;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const int kq = lane >> 4, blk = (lane >> 2) & 3, c = lane & 3;
        const int ia = blk < 2 ? c : (blk == 2 ? 4 + c : 8 + c);
        const int ib = blk == 0 ? c : (blk == 3 ? 8 + c : 4 + c);
        const double* pa = rows + kq * kRow + ia;
        const double* pb = rows + kq * kRow + ib;
        double a[16], b[16];
#pragma unroll
        for (int s = 0; s < 16; s++) { a[s] = pa[4 * s * kRow]; b[s] = pb[4 * s * kRow]; }   // all 32 operand reads in flight before the first MFMA
#pragma unroll
        for (int s = 0; s < 16; s++) acc = __builtin_amdgcn_mfma_f64_4x4x4f64(a[s], b[s], acc, 0, 0, 0);
    }
    // block partial: 36 upper-triangle entries of the 8x8 Gram, [36] = cost, [37] = count
    // `key` != 0: the partial is PUBLISHED for the reducer block of the same launch (fused_tail) as 40 granules of 16 bytes,
    // {value bits, value bits ^ key}, each written by ONE write-through (sc1) 16-byte store.  The data is its own flag: a granule
    // whose halves satisfy hi == lo ^ key was written by THIS launch (key is unique per launch), so the reducer needs no ticket, no
    // fence and no drained-store wait on the producer side (MI355X_MICROARCH.md, inter-workgroup visibility, form R2).
    __device__ __forceinline__ void finish(double* lds, double* slot, unsigned long long key = 0ull) {
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        __syncthreads();                 // every wave is done with its row area before the LDS is reused for the wave results
        lds[wave * 64 + lane] = acc;     // this wave's four 4x4 result blocks
        __syncthreads();
        if (threadIdx.x < 40) {
            const int src = gram_lane(threadIdx.x);
            double s = 0.0;
            for (int w = 0; w < (int)(blockDim.x >> 6); w++) s += lds[w * 64 + src];
            if (key) store_granule(slot + 2 * threadIdx.x, s, key);
            else slot[threadIdx.x] = s;
        }
    }
};

__device__ __forceinline__ void load_body_pose(const PoseArg& pa, dq& Q, d3& T) {
    if (pa.state) { const double* s = pa.state->pose; T = d3{s[0], s[1], s[2]}; Q = dq{s[3], s[4], s[5], s[6]}; }
    else { T = d3{pa.t[0], pa.t[1], pa.t[2]}; Q = dq{pa.q[0], pa.q[1], pa.q[2], pa.q[3]}; }
}

// Residual, 1x7 Jacobian row and loss corrector of ONE correspondence (Jr[0..6] = robustified Jacobian, Jr[7] = residual; returns the robust
// cost) — shared by the linearisation launch and by the association launch that linearises on the fly (k_associate_lin).
//   surf: LidarPlaneNormFactor / LidarPlaneNormIncreFactor, L/include/factors/LidarKeyframeFactor.h:86-90, 118-128; `score` is the (count-scaled) weight
// In two parts since round 4: the geometry (rotations, Jacobian of the rotation) does not depend on the weight, and for the count-scaled flavour the weight
// exists only after the block has summed the association's block counts — surf_lin_geom runs while those counts travel, surf_lin_scale afterwards.  The
// same operations on the same operands as before: `score` multiplies finished values.
struct SurfLinGeom { d3 n; double base; double jq[4]; };       // base = n . pw + d  (the unweighted residual)
__device__ __forceinline__ SurfLinGeom surf_lin_geom(const MatchParams& P, const dq& Q, const d3& T, const dq& qlb_inv, float4 ql, float4 nd) {
    SurfLinGeom gm;
    d3 cp{(double)ql.x, (double)ql.y, (double)ql.z};
    gm.n = d3{(double)nd.x, (double)nd.y, (double)nd.z};
    d3 v;
    if (P.variant == 2) v = cp;   // LidarPlaneNormIncreFactor, LidarKeyframeFactor.h:118-128
    else v = qrot(qlb_inv, cp - d3{P.t_lb[0], P.t_lb[1], P.t_lb[2]});                                  // :86
    d3 pw = qrot(Q, v) + T;                                                                              // :87
    gm.base = dot3(gm.n, pw) + (double)nd.w;
    qrot_jac_row(Q, v, gm.n, gm.jq);
    return gm;
}
__device__ __forceinline__ double surf_lin_scale(const MatchParams& P, const SurfLinGeom& gm, double score, double Jr[8]) {
    if (P.variant == 2) score = 1.0;
    double r = score * gm.base;                                                                          // :90
    double J[7] = {score * gm.n.x, score * gm.n.y, score * gm.n.z, score * gm.jq[0], score * gm.jq[1], score * gm.jq[2], score * gm.jq[3]};
    const double cost = robustify(P.loss, P.loss_a, J, r, P.no_cost == 0);
#pragma unroll
    for (int k = 0; k < 7; k++) Jr[k] = J[k];
    Jr[7] = r;
    return cost;
}
__device__ __forceinline__ double surf_lin_row(const MatchParams& P, const dq& Q, const d3& T, const dq& qlb_inv, float4 ql, float4 nd, double score, double Jr[8]) {
    const SurfLinGeom gm = surf_lin_geom(P, Q, T, qlb_inv, ql, nd);
    return surf_lin_scale(P, gm, score, Jr);
}
//   edge: LidarEdgeFactor, LidarKeyframeFactor.h:38-44 (no extrinsic: SURVEY F6); `s` is the (count-scaled) weight
__device__ __forceinline__ double edge_lin_row(const MatchParams& P, const dq& Q, const d3& T, float4 ql, float4 fa, float4 fb, double s, double Jr[8]) {
    d3 cp{(double)ql.x, (double)ql.y, (double)ql.z};
    d3 Av{(double)fa.x, (double)fa.y, (double)fa.z}, B{(double)fb.x, (double)fb.y, (double)fb.z};
    d3 lp = qrot(Q, cp) + T;                    // :38
    d3 nu = cross3(lp - Av, lp - B);            // :40
    d3 de = Av - B;                             // :41
    double nn = sqrt(dot3(nu, nu)), dn = sqrt(dot3(de, de));
    double r = s * (nn / dn);                   // :43-44
    // d|nu|/dlp = nu^T [a-b]x / |nu| = (nu x (B - A))^T / |nu|
    d3 g = cross3(nu, B - Av);
    double k = s / (nn * dn);
    g = k * g;
    double jq[4];
    qrot_jac_row(Q, cp, g, jq);
    double J[7] = {g.x, g.y, g.z, jq[0], jq[1], jq[2], jq[3]};
    const double cost = robustify(P.loss, P.loss_a, J, r, P.no_cost == 0);
#pragma unroll
    for (int kk = 0; kk < 7; kk++) Jr[kk] = J[kk];
    Jr[7] = r;
    return cost;
}

// wave-level ordering of LDS traffic inside ONE wave (LDS operations of a wave execute in order; the fence keeps the compiler from
// moving them) — the tail of the reduction and the GN update run in a single wave, without s_barrier
#define LILI_WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); } while (0)

// sin(x)/x and cos(x) from x^2 for the small rotation of a Gauss-Newton step (|x| < 0.5: the series are truncated below 1e-19
// relative): 16 fused multiply-adds instead of two libm calls with argument reduction (~120 dependent f64 instructions at the
// end of the latency-bound update chain).  ceres::QuaternionParameterization::Plus takes sin / cos from libm, which is not
// correctly rounded either; the two agree to 1-2 ulp.
__device__ __forceinline__ void sinc_cos_small(double x2, double& sinc, double& c) {
    double s = -1.0 / 355687428096000.0;            // -1/17!
    s = __fma_rn(s, x2, 1.0 / 1307674368000.0);      //  1/15!
    s = __fma_rn(s, x2, -1.0 / 6227020800.0);        // -1/13!
    s = __fma_rn(s, x2, 1.0 / 39916800.0);           //  1/11!
    s = __fma_rn(s, x2, -1.0 / 362880.0);            // -1/9!
    s = __fma_rn(s, x2, 1.0 / 5040.0);
    s = __fma_rn(s, x2, -1.0 / 120.0);
    s = __fma_rn(s, x2, 1.0 / 6.0);
    sinc = __fma_rn(-s, x2, 1.0);
    double k = 1.0 / 6402373705728000.0;             //  1/18!
    k = __fma_rn(k, x2, -1.0 / 20922789888000.0);    // -1/16!
    k = __fma_rn(k, x2, 1.0 / 87178291200.0);        //  1/14!
    k = __fma_rn(k, x2, -1.0 / 479001600.0);         // -1/12!
    k = __fma_rn(k, x2, 1.0 / 3628800.0);            //  1/10!
    k = __fma_rn(k, x2, -1.0 / 40320.0);
    k = __fma_rn(k, x2, 1.0 / 720.0);
    k = __fma_rn(k, x2, -1.0 / 24.0);
    k = __fma_rn(k, x2, 0.5);
    c = __fma_rn(-k, x2, 1.0);
}

// Linearisation bodies: `bid` of `nb` virtual blocks of one kind (the combined surf + edge launch maps its grid onto both).
__device__ __forceinline__ void lin_surf_body(const LinArgs& A, int bid, const PoseArg& pa, const MatchParams& P, const SlotState* __restrict__ state,
                                              const int* __restrict__ n_global, double* lds, unsigned long long key) {
    const float4* __restrict__ queries = A.queries; const float4* __restrict__ rec_nd = A.rec0;
    const double* __restrict__ rec_score = reinterpret_cast<const double*>(A.rec1);
    const unsigned char* __restrict__ valid = A.valid;
    const int n_q = A.n_q;
    tstamp(state, P.debug, 100, 0);
    if ((P.debug & 512) && bid == 100 && threadIdx.x == 0) const_cast<SlotState*>(state)->tprof[15] = (long long)__builtin_amdgcn_s_memrealtime();
    GramAcc ga; ga.init();
    dq Q; d3 T;
    load_body_pose(pa, Q, T);
    const dq qlb_inv{P.q_lb_inv_jet[0], P.q_lb_inv_jet[1], P.q_lb_inv_jet[2], P.q_lb_inv_jet[3]};
    // N of R:861: this rank's count (sum of the association's block counts) or, when a multi-GPU caller has
    // all-reduced it, the global count in state->n_res
    // the first tile's records are requested before the count reduction below (which synchronises the block twice), and
    // unconditionally — one memory round trip instead of valid -> record
    const int BS = blockDim.x;
    const int i0 = bid * BS + threadIdx.x;
    const int i0c = min(i0, n_q - 1);
    unsigned char v0 = valid[i0c];
    float4 ql0 = queries[i0c], nd0 = rec_nd[i0c];
    double sc0 = rec_score[i0c];
    // ROT count scaling exactly as the reference writes it (R/src/BackendFusion.cpp:861, pinned by tests/test_reference_*.py against the reference text):
    // vec_surf_scores[i] * 1000 / vec_surf_res_cnt  =  (score * 1000.0) / (double)N — a multiply, then a true division.
    // The block counts are requested first, the geometry of the first tile (which does not depend on N) runs while they travel, the barriers of the
    // count sum come after it (round 4: they used to stand in front of all the arithmetic).
    const bool sum_counts = P.scale_surf_num > 0 && A.block_counts;
    const int wave_cnt = sum_counts ? sum_block_counts_begin(A.block_counts, A.n_bc) : 0;
    const bool ok0 = i0 < n_q && v0;
    SurfLinGeom g0;
    if (ok0) g0 = surf_lin_geom(P, Q, T, qlb_inv, ql0, nd0);
    double n_den = 1.0;
    if (P.scale_surf_num > 0) n_den = (double)(A.block_counts ? sum_block_counts_end(wave_cnt) : (n_global ? n_global[0] : state->n_res[0]));
    tstamp(state, P.debug, 100, 1);
    for (int base = bid * BS; base < n_q; base += A.nb * BS) {
        int i = base + threadIdx.x;
        const bool first = base == bid * BS;
        const int ic = min(i, n_q - 1);
        bool ok = first ? ok0 : (i < n_q && valid[ic]);
        double Jr[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        double cost = 0.0;
        if (ok) {
            double score = first ? sc0 : rec_score[i];
            if (P.scale_surf_num > 0) score = score * P.scale_surf_num / n_den;
            if (first) cost = surf_lin_scale(P, g0, score, Jr);
            else cost = surf_lin_row(P, Q, T, qlb_inv, queries[i], rec_nd[i], score, Jr);
        }
        tstamp(state, P.debug, 100, 2);
        ga.add_rows(Jr, cost, ok, lds);
        tstamp(state, P.debug, 100, 3);
        if ((P.debug & 512) && bid == 100 && (threadIdx.x & 63) == 0) const_cast<SlotState*>(state)->tprof[threadIdx.x >> 6] = (long long)__builtin_amdgcn_s_memrealtime();
    }
    ga.finish(lds, A.partials + (size_t)bid * kPartialStride, key);
    tstamp(state, P.debug, 100, 4);
}

__device__ __forceinline__ void lin_edge_body(const LinArgs& A, int bid, const PoseArg& pa, const MatchParams& P, const SlotState* __restrict__ state,
                                              const int* __restrict__ n_global, double* lds, unsigned long long key) {
    const float4* __restrict__ queries = A.queries; const float4* __restrict__ rec_a = A.rec0;
    const float4* __restrict__ rec_b = reinterpret_cast<const float4*>(A.rec1);
    const unsigned char* __restrict__ valid = A.valid;
    const int n_q = A.n_q;
    GramAcc ga; ga.init();
    dq Q; d3 T;
    load_body_pose(pa, Q, T);
    // R:843: points[i].intensity * 200 / vec_edge_res_cnt — float * int / int, i.e. FLOAT arithmetic (pinned by tests/test_reference_*.py against the reference text)
    float n_den = 1.0f;
    if (P.scale_edge_num > 0) n_den = (float)(A.block_counts ? sum_block_counts(A.block_counts, A.n_bc) : (n_global ? n_global[1] : state->n_res[1]));
    const float n_num = (float)P.scale_edge_num;
    for (int base = bid * blockDim.x; base < n_q; base += A.nb * blockDim.x) {
        int i = base + threadIdx.x;
        bool ok = i < n_q && valid[i];
        double Jr[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        double cost = 0.0;
        if (ok) {
            float4 ql = queries[i]; float4 fa = rec_a[i], fb = rec_b[i];
            double s = (double)fa.w;
            if (P.scale_edge_num > 0) s = (double)__fdiv_rn(__fmul_rn(fa.w, n_num), n_den);
            cost = edge_lin_row(P, Q, T, ql, fa, fb, s, Jr);
        }
        ga.add_rows(Jr, cost, ok, lds);
    }
    ga.finish(lds, A.partials + (size_t)bid * kPartialStride, key);
}


// ---- exchange between the workgroups of ONE persistent launch (lili_s2m_lm.hip, k_iterate_coop in lili_s2m_coop.hip): 16-byte granules
// {value, value ^ key}, key unique per (launch, round) — the data is its own flag (cdna_hip_programming.md Guideline 16, form R2).
__device__ __forceinline__ unsigned long long xchg_key(unsigned long long launch, int round) { return (launch * 4096ull + (unsigned long long)round + 1ull) * 0x9E3779B97F4A7C15ull; }
// ONE wave (lanes 0..63): waits until the `count` x NE granules at `src` (`stride` doubles per source) carry `key`, then out[e] = sum over the
// sources in index order.  `vals` = LDS scratch [>= count][NE].  Returns false if the (bounded) wait gave up.
template <int NE>
__device__ __forceinline__ bool xchg_gather(const double* src, int count, unsigned long long key, double (*vals)[NE], double* out, int stride = kPartialStride) {
    const int lane = threadIdx.x & 63;
    const int n = count * NE;
    bool ok = false;
    for (unsigned sweep = 0; sweep < (1u << 22); sweep++) {
        bool all = true;
        for (int g = lane; g < n; g += 64) {
            const int m = g / NE, e = g - m * NE;
            unsigned long long lo, hi;
            load_granule(src + (size_t)m * stride + 2 * e, lo, hi);
            all = all && ((lo ^ hi) == key);
            vals[m][e] = __longlong_as_double((long long)lo);
        }
        if (__all(all)) { ok = true; break; }
        __builtin_amdgcn_s_sleep(1);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (lane < NE) {
        double s = 0.0;
        for (int m = 0; m < count; m++) s += vals[m][lane];
        out[lane] = s;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    return ok;
}

}  // namespace lili
