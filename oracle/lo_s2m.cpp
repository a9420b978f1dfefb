// ORACLE — TEST INFRASTRUCTURE ONLY (see lo_math.h header).  PARITY: association records, residual-block rows, loss
// corrector and A/b assembly are pinned bit for bit against the reference's own functions compiled from their text
// (oracle/_ref: libref_backend_{L,R}, libref_lo, libref_factors, libref_marg — tests/test_reference_cpu.py); the kd-tree, QR and
// eigen-solver underneath restate third-party code and are checked against semantics only.
//
// CPU restatement of the reference's scan-to-map matcher:
//   * exact kNN-5 over the local map          — pcl::KdTreeFLANN::nearestKSearch call sites
//                                                L/src/BackendFusion.cpp:1541,1611 (semantics App. B1)
//   * findCorrespondingCornerFeatures          — L/src/BackendFusion.cpp:1531-1599, R/src/BackendFusion.cpp:1394-1462
//   * findCorrespondingSurfFeatures            — L/src/BackendFusion.cpp:1601-1681, R/src/BackendFusion.cpp:1464-1520,
//                                                L/src/LidarOdometry.cpp:352-413 (front-end)
//   * LidarEdgeFactor / LidarPlaneNormFactor / LidarPlaneNormIncreFactor evaluated with 7-partial
//     dual numbers exactly as ceres::AutoDiffCostFunction does — L/include/factors/LidarKeyframeFactor.h:12-139
//   * loss corrector + JtJ/Jtr accumulation    — L/src/MarginalizationFactor.cpp:3-29,44-70
//   * one Gauss-Newton step on the 6-dof local parameterisation (ceres::QuaternionParameterization)
// (L/ = /root/reference/LiLi-OM/, R/ = /root/reference/LiLi-OM-ROT/)
#include "lo_math.h"
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <atomic>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>
#include <pthread.h>
#include <sched.h>

using namespace lo;


// ------------------------------------------------------------------------------------------
// parameters (plain C struct; independent of the product's lili_s2m_params on purpose)
// ------------------------------------------------------------------------------------------
enum { LO_VARIANT_LIVOX = 0, LO_VARIANT_ROT = 1, LO_VARIANT_FRONTEND = 2 };
enum { LO_LOSS_NONE = 0, LO_LOSS_CAUCHY = 1, LO_LOSS_HUBER = 2 };
struct lo_params {
    int variant;
    int loss;              // Cauchy(1.0) back-end (L:845), Huber(0.1) front-end (LidarOdometry.cpp:507)
    double loss_a;
    double lidar_const;    // L config 20, R config 7.5
    double kd_max_radius;  // compared against a SQUARED distance
    double edge_gate;      // 1.0 hard-coded (L:1543)
    double surf_dist_thres;
    double reflect_thres;  // Livox only
    double surf_weight_min;  // 0.2 Livox (L:1665), 0.3 ROT (R:1504), 0.4 front-end (LidarOdometry.cpp:400)
    double edge_dist_max;    // ROT only: 0.1 (R:1443); <= 0 disables
    double q_lb[4];          // w,x,y,z
    double t_lb[3];
};

// ------------------------------------------------------------------------------------------
// exact kNN-5.  Distance = FLANN L2_Simple on 3 floats: result += diff*diff, f32, in x,y,z order.
// Order: ascending (d2, index) — FLANN's tie order is traversal-dependent (App. B1); we fix it.
// ------------------------------------------------------------------------------------------
static inline float d2f(const float* a, const float* b) {
    float r = 0.f;
    float d = a[0] - b[0]; r += d * d;
    d = a[1] - b[1]; r += d * d;
    d = a[2] - b[2]; r += d * d;
    return r;
}
struct Top5 {
    float d[5]; int i[5]; int n;
    Top5() : n(0) { for (int k = 0; k < 5; k++) { d[k] = INFINITY; i[k] = -1; } }
    inline bool better(float dd, int ii, int k) const { return dd < d[k] || (dd == d[k] && ii < i[k]); }
    inline void push(float dd, int ii) {
        if (n == 5 && !better(dd, ii, 4)) return;
        int k = n < 5 ? n : 4;
        while (k > 0 && better(dd, ii, k - 1)) { d[k] = d[k - 1]; i[k] = i[k - 1]; k--; }
        d[k] = dd; i[k] = ii;
        if (n < 5) n++;
    }
    inline float worst() const { return n < 5 ? INFINITY : d[4]; }
};

struct KdNode { int lo, hi; int left, right; int dim; float split; };
struct KdTree {
    const float* pts; int n;
    std::vector<int> order;
    std::vector<KdNode> nodes;
    int build(int lo, int hi) {
        int id = (int)nodes.size();
        nodes.push_back({lo, hi, -1, -1, -1, 0.f});
        if (hi - lo <= 12) return id;
        float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
        for (int k = lo; k < hi; k++) for (int d = 0; d < 3; d++) {
            float v = pts[3 * (size_t)order[k] + d];
            mn[d] = std::min(mn[d], v); mx[d] = std::max(mx[d], v);
        }
        int dim = 0; float ext = mx[0] - mn[0];
        for (int d = 1; d < 3; d++) if (mx[d] - mn[d] > ext) { ext = mx[d] - mn[d]; dim = d; }
        if (!(ext > 0)) return id;  // all identical: keep as a (large) leaf
        int mid = (lo + hi) / 2;
        const float* P = pts;
        std::nth_element(order.begin() + lo, order.begin() + mid, order.begin() + hi,
                         [P, dim](int a, int b) { return P[3 * (size_t)a + dim] < P[3 * (size_t)b + dim]; });
        float split = pts[3 * (size_t)order[mid] + dim];
        int l = build(lo, mid);
        int r = build(mid, hi);
        nodes[id].left = l; nodes[id].right = r; nodes[id].dim = dim; nodes[id].split = split;
        return id;
    }
    void search(int id, const float* q, Top5& best) const {
        const KdNode& nd = nodes[id];
        if (nd.left < 0) {
            for (int k = nd.lo; k < nd.hi; k++) { int ii = order[k]; best.push(d2f(q, pts + 3 * (size_t)ii), ii); }
            return;
        }
        // left holds values <= split, right holds values >= split
        float diff = q[nd.dim] - nd.split;
        int nearc = diff < 0 ? nd.left : nd.right;
        int farc = diff < 0 ? nd.right : nd.left;
        search(nearc, q, best);
        // Conservative bound on the computed f32 distance of any point on the far side: every such
        // point p has |p[dim]-q[dim]| >= |split-q[dim]|; f32 rounding is monotone, so its first
        // squared term is >= diff*diff and adding non-negative terms never decreases the sum.
        float bound = diff * diff;
        if (bound <= best.worst()) search(farc, q, best);  // '<=': equal distance may still win on index
    }
};

extern "C" void* lo_kdtree_build(const float* xyz, int n) {
    KdTree* t = new KdTree();
    t->pts = xyz; t->n = n;
    t->order.resize(n);
    for (int i = 0; i < n; i++) t->order[i] = i;
    t->nodes.reserve((size_t)n / 4 + 16);
    if (n > 0) t->build(0, n);
    return t;
}
extern "C" void lo_kdtree_free(void* t) { delete (KdTree*)t; }

static void knn5_one(const KdTree* t, const float* q, int* idx, float* d2) {
    Top5 best;
    if (t->n > 0) t->search(0, q, best);
    for (int k = 0; k < 5; k++) { idx[k] = best.i[k]; d2[k] = best.d[k]; }
}

// Persistent worker pool with dynamic chunks (round 5, VERDICT r4 #5: the all-core CPU figure spawned `nthreads` std::threads per call and gave each one a static
// n / nthreads-sized range — 9x on 64 cores).  Workers are created once (lazily, so they inherit the affinity mask the caller has set: bench.py pins the process to
// one socket's physical cores first; lo_pool_reset drops them when the mask changes) and pull chunks of kChunk queries from an atomic counter, so a thread that
// drew expensive queries (far rings walk more of the tree) does not hold the others up.  The caller's thread works too.
namespace {
constexpr int kChunk = 256;      // queries per chunk: 782 chunks for a 200 k-point scan = 12.2 per thread on 64 cores (1024-query chunks left a quarter of the cores idle in the last round)
struct Pool {
    std::vector<std::thread> th;
    std::mutex m;
    std::condition_variable cv, cv_done;
    const std::function<void(int)>* job = nullptr;
    int n_chunks = 0, use = 0, running = 0;
    std::atomic<int> next{0};
    unsigned long gen = 0;
    bool stop = false;
    void worker(int id, unsigned long seen) {      // seen: the generation at creation — a new worker must not take a finished job's stale generation for a new one
        for (;;) {
            const std::function<void(int)>* f;
            {
                std::unique_lock<std::mutex> lk(m);
                cv.wait(lk, [&] { return stop || gen != seen; });
                if (stop) return;
                seen = gen;
                if (id >= use) continue;
                f = job;
            }
            for (int c; (c = next.fetch_add(1, std::memory_order_relaxed)) < n_chunks;) (*f)(c);
            {
                std::lock_guard<std::mutex> lk(m);
                if (--running == 0) cv_done.notify_one();
            }
        }
    }
    void run(int chunks, int nthreads, const std::function<void(int)>& f) {
        const int helpers = std::max(0, std::min(nthreads, chunks) - 1);
        while ((int)th.size() < helpers) {
            const int id = (int)th.size(); const unsigned long g0 = gen;
            th.emplace_back([this, id, g0] { worker(id, g0); });
            // one core per worker: the (id + 1)-th CPU of the mask the CALLER runs under (bench.py: the physical cores of one socket) — unpinned workers were
            // seen to share cores for whole registrations (five runs of the same registration: 112 ... 310 it/s)
            cpu_set_t mask;
            if (sched_getaffinity(0, sizeof(mask), &mask) == 0) {
                const int ncpu = CPU_COUNT(&mask);
                int want = ncpu > 0 ? (id + 1) % ncpu : 0, seen = 0;
                for (int c = 0; c < CPU_SETSIZE; c++) if (CPU_ISSET(c, &mask)) { if (seen == want) { cpu_set_t one; CPU_ZERO(&one); CPU_SET(c, &one); pthread_setaffinity_np(th.back().native_handle(), sizeof(one), &one); break; } seen++; }
            }
        }
        {
            std::lock_guard<std::mutex> lk(m);
            job = &f; n_chunks = chunks; use = helpers; running = helpers; next.store(0, std::memory_order_relaxed); gen++;
        }
        cv.notify_all();
        for (int c; (c = next.fetch_add(1, std::memory_order_relaxed)) < chunks;) f(c);
        std::unique_lock<std::mutex> lk(m);
        cv_done.wait(lk, [&] { return running == 0; });
    }
    void reset() {
        { std::lock_guard<std::mutex> lk(m); stop = true; }
        cv.notify_all();
        for (auto& t : th) t.join();
        th.clear();
        stop = false;
    }
    ~Pool() { reset(); }
};
Pool& pool() { static Pool p; return p; }
}  // namespace
extern "C" void lo_pool_reset() { pool().reset(); }
extern "C" int lo_pool_threads() { return (int)pool().th.size(); }

template <class F> static void parallel_for(int n, int nthreads, F f) {
    if (nthreads <= 1 || n < 2 * kChunk) { f(0, n); return; }
    const int chunks = (n + kChunk - 1) / kChunk;
    const std::function<void(int)> job = [&](int c) { f(c * kChunk, std::min(n, (c + 1) * kChunk)); };
    pool().run(chunks, nthreads, job);
}

extern "C" void lo_knn5(void* tree, const float* q, int m, int* idx, float* d2, int nthreads) {
    const KdTree* t = (const KdTree*)tree;
    parallel_for(m, nthreads, [=](int lo, int hi) { for (int i = lo; i < hi; i++) knn5_one(t, q + 3 * (size_t)i, idx + 5 * (size_t)i, d2 + 5 * (size_t)i); });
}
extern "C" void lo_knn5_brute(const float* xyz, int n, const float* q, int m, int* idx, float* d2) {
    for (int i = 0; i < m; i++) {
        Top5 best;
        for (int j = 0; j < n; j++) best.push(d2f(q + 3 * (size_t)i, xyz + 3 * (size_t)j), j);
        for (int k = 0; k < 5; k++) { idx[5 * (size_t)i + k] = best.i[k]; d2[5 * (size_t)i + k] = best.d[k]; }
    }
}

// ------------------------------------------------------------------------------------------
// transformPoint — L/src/BackendFusion.cpp:695-711: f64 rotate+translate, stored to f32.
// ------------------------------------------------------------------------------------------
static inline void transform_point(const float* p, Q4 q, V3 t, float out[3]) {
    V3 o = qrot(q, V3{(double)p[0], (double)p[1], (double)p[2]}) + t;
    out[0] = (float)o.x; out[1] = (float)o.y; out[2] = (float)o.z;
}

// eigenvector sign is arbitrary in Eigen (App. B4) and only +-v enters the result symmetrically;
// we canonicalise (largest-|component| positive, first such on ties) so records can be compared.
static inline void canon_sign(double v[3]) {
    int k = 0; double m = std::fabs(v[0]);
    if (std::fabs(v[1]) > m) { m = std::fabs(v[1]); k = 1; }
    if (std::fabs(v[2]) > m) { k = 2; }
    if (v[k] < 0) { v[0] = -v[0]; v[1] = -v[1]; v[2] = -v[2]; }
}

// ------------------------------------------------------------------------------------------
// findCorrespondingSurfFeatures for queries [lo,hi).  Per-query outputs are written at the query's
// own index (valid[i] = 0/1); the reference's ordered lists are these rows with valid==1, in order.
// ------------------------------------------------------------------------------------------
struct SurfIO {
    const KdTree* tree; const float* map_xyz; const float* map_refl; int n_map;
    const float* q_xyz; const float* q_refl; Q4 q; V3 t; const lo_params* P;
    unsigned char* valid; int* nn_idx; float* nn_d2; float* rec_cp; float* rec_n; float* rec_d; double* rec_score;
};
static void surf_range(const SurfIO& io, int lo_i, int hi_i) {
    const lo_params& P = *io.P;
    for (int i = lo_i; i < hi_i; i++) {
        io.valid[i] = 0;
        const float* pl = io.q_xyz + 3 * (size_t)i;
        float pm[3];
        transform_point(pl, io.q, io.t, pm);
        int* idx = io.nn_idx + 5 * (size_t)i; float* d2 = io.nn_d2 + 5 * (size_t)i;
        knn5_one(io.tree, pm, idx, d2);
        if (io.n_map < 5) continue;  // reference indexes [4] out of bounds here (UB); we reject
        if (!((double)d2[4] < P.kd_max_radius)) continue;  // L:1615: float d2 promoted against the double parameter
        double A[5][3], b[5];
        double sum_w = 0;
        if (P.variant == LO_VARIANT_LIVOX) {
            // L:1617-1638 reflectivity weighting (IEEE inf/NaN behaviour kept, App. A6)
            double w[5];
            for (int j = 0; j < 5; j++) {
                double tmp_w = std::fabs((double)(io.q_refl[i] - io.map_refl[idx[j]]));
                // pt_in_map.curvature - map.curvature is a float subtraction promoted by fabs(double)
                sum_w += tmp_w;
                w[j] = 1.0 / tmp_w;
            }
            for (int j = 0; j < 5; j++) w[j] /= sum_w;
            if (sum_w > P.reflect_thres) continue;
            for (int j = 0; j < 5; j++) {
                const float* m = io.map_xyz + 3 * (size_t)idx[j];
                A[j][0] = w[j] * m[0]; A[j][1] = w[j] * m[1]; A[j][2] = w[j] * m[2];
                b[j] = -1.0 * w[j];
            }
        } else {
            for (int j = 0; j < 5; j++) {
                const float* m = io.map_xyz + 3 * (size_t)idx[j];
                A[j][0] = m[0]; A[j][1] = m[1]; A[j][2] = m[2]; b[j] = -1.0;
            }
        }
        double nv[3];
        lstsq_5x3_colpiv(A, b, nv);
        double nn = std::sqrt(nv[0] * nv[0] + nv[1] * nv[1] + nv[2] * nv[2]);
        double normInverse = 1 / nn;
        // Eigen normalize(): v /= norm()
        nv[0] /= nn; nv[1] /= nn; nv[2] /= nn;
        bool planeValid = true;
        for (int j = 0; j < 5; j++) {
            const float* m = io.map_xyz + 3 * (size_t)idx[j];
            if (std::fabs(nv[0] * m[0] + nv[1] * m[1] + nv[2] * m[2] + normInverse) > P.surf_dist_thres) { planeValid = false; break; }
        }
        if (!planeValid) continue;
        // L:1661-1662 — float pd, float weight; sqrt(sqrt(float)) resolves to the float overloads (App. A1)
        float pd = (float)(nv[0] * pm[0] + nv[1] * pm[1] + nv[2] * pm[2] + normInverse);
        float r2 = pm[0] * pm[0] + pm[1] * pm[1] + pm[2] * pm[2];
        float weight = (float)(1 - 0.9 * std::fabs(pd) / std::sqrt(std::sqrt(r2)));
        if (!((double)weight > P.surf_weight_min)) continue;  // L:1665: float weight promoted against the double literal
        io.valid[i] = 1;
        float* cp = io.rec_cp + 3 * (size_t)i; float* rn = io.rec_n + 3 * (size_t)i;
        cp[0] = pl[0]; cp[1] = pl[1]; cp[2] = pl[2];
        rn[0] = (float)(weight * nv[0]); rn[1] = (float)(weight * nv[1]); rn[2] = (float)(weight * nv[2]);
        io.rec_d[i] = (float)(weight * normInverse);
        if (P.variant == LO_VARIANT_LIVOX) io.rec_score[i] = P.lidar_const * (weight + std::exp(-sum_w));  // L:1676
        else if (P.variant == LO_VARIANT_ROT) io.rec_score[i] = P.lidar_const * weight;                    // R:1515
        else io.rec_score[i] = 1.0;                                                                        // front-end: no score
    }
}

extern "C" int lo_associate_surf(void* tree, const float* map_xyz, const float* map_refl, int n_map,
                      const float* q_xyz, const float* q_refl, int n_q,
                      const double pose_q[4], const double pose_t[3], const lo_params* P, int nthreads,
                      unsigned char* valid, int* nn_idx, float* nn_d2,
                      float* rec_cp, float* rec_n, float* rec_d, double* rec_score) {
    SurfIO io{(const KdTree*)tree, map_xyz, map_refl, n_map, q_xyz, q_refl,
              Q4{pose_q[0], pose_q[1], pose_q[2], pose_q[3]}, V3{pose_t[0], pose_t[1], pose_t[2]}, P,
              valid, nn_idx, nn_d2, rec_cp, rec_n, rec_d, rec_score};
    parallel_for(n_q, nthreads, [&io](int lo, int hi) { surf_range(io, lo, hi); });
    int cnt = 0; for (int i = 0; i < n_q; i++) cnt += valid[i];
    return cnt;
}

// ------------------------------------------------------------------------------------------
// findCorrespondingCornerFeatures
// ------------------------------------------------------------------------------------------
struct EdgeIO {
    const KdTree* tree; const float* map_xyz; int n_map; const float* q_xyz; Q4 q; V3 t; const lo_params* P;
    unsigned char* valid; int* nn_idx; float* nn_d2; float* rec_cp; float* rec_a; float* rec_b; float* rec_s;
};
static void edge_range(const EdgeIO& io, int lo_i, int hi_i) {
    const lo_params& P = *io.P;
    for (int i = lo_i; i < hi_i; i++) {
        io.valid[i] = 0;
        const float* pl = io.q_xyz + 3 * (size_t)i;
        float pm[3];
        transform_point(pl, io.q, io.t, pm);
        int* idx = io.nn_idx + 5 * (size_t)i; float* d2 = io.nn_d2 + 5 * (size_t)i;
        knn5_one(io.tree, pm, idx, d2);
        if (io.n_map < 5) continue;
        if (!((double)d2[4] < P.edge_gate)) continue;  // L:1543
        V3 c{0, 0, 0}; V3 nc[5];
        for (int j = 0; j < 5; j++) {
            const float* m = io.map_xyz + 3 * (size_t)idx[j];
            nc[j] = V3{(double)m[0], (double)m[1], (double)m[2]};
            c = c + nc[j];
        }
        c = V3{c.x / 5.0, c.y / 5.0, c.z / 5.0};
        double A[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
        for (int j = 0; j < 5; j++) {
            V3 z = nc[j] - c; double zz[3] = {z.x, z.y, z.z};
            for (int r = 0; r < 3; r++) for (int cc = 0; cc < 3; cc++) A[r][cc] = A[r][cc] + zz[r] * zz[cc];
        }
        double ev[3], evec[3][3];
        eig3_sym(A, ev, evec);
        if (!(ev[2] > 3 * ev[1])) continue;  // L:1575
        double u[3] = {evec[2][0], evec[2][1], evec[2][2]};
        canon_sign(u);
        V3 ud{u[0], u[1], u[2]};
        V3 ptA = c + 0.1 * ud, ptB = c - 0.1 * ud;
        if (P.edge_dist_max > 0) {  // R:1437-1443
            V3 lp{(double)pm[0], (double)pm[1], (double)pm[2]};
            V3 nu = cross(lp - ptA, lp - ptB);
            V3 de = ptA - ptB;
            double dist = norm(nu) / norm(de);
            if (!(dist < P.edge_dist_max)) continue;
        }
        io.valid[i] = 1;
        float* cp = io.rec_cp + 3 * (size_t)i; float* ra = io.rec_a + 3 * (size_t)i; float* rb = io.rec_b + 3 * (size_t)i;
        cp[0] = pl[0]; cp[1] = pl[1]; cp[2] = pl[2];
        ra[0] = (float)ptA.x; ra[1] = (float)ptA.y; ra[2] = (float)ptA.z;   // PointType x,y,z are float (L:1584-1589)
        rb[0] = (float)ptB.x; rb[1] = (float)ptB.y; rb[2] = (float)ptB.z;
        io.rec_s[i] = (float)P.lidar_const;                                  // pt_in_local.intensity = lidar_const (L:1581)
    }
}
extern "C" int lo_associate_edge(void* tree, const float* map_xyz, int n_map, const float* q_xyz, int n_q,
                      const double pose_q[4], const double pose_t[3], const lo_params* P, int nthreads,
                      unsigned char* valid, int* nn_idx, float* nn_d2,
                      float* rec_cp, float* rec_a, float* rec_b, float* rec_s) {
    EdgeIO io{(const KdTree*)tree, map_xyz, n_map, q_xyz,
              Q4{pose_q[0], pose_q[1], pose_q[2], pose_q[3]}, V3{pose_t[0], pose_t[1], pose_t[2]}, P,
              valid, nn_idx, nn_d2, rec_cp, rec_a, rec_b, rec_s};
    parallel_for(n_q, nthreads, [&io](int lo, int hi) { edge_range(io, lo, hi); });
    int cnt = 0; for (int i = 0; i < n_q; i++) cnt += valid[i];
    return cnt;
}

// ------------------------------------------------------------------------------------------
// Factors, evaluated the way ceres::AutoDiffCostFunction<F,1,3,4> does: parameters become Jets
// (t -> partials 0..2, q(w,x,y,z) -> partials 3..6), the functor body runs on Jets.
// ------------------------------------------------------------------------------------------
static inline JV3 jv3c(V3 v) { return {Jet7(v.x), Jet7(v.y), Jet7(v.z)}; }

// LidarKeyframeFactor.h:25-47  (q_l_b, t_l_b are constructed but never used: SURVEY F6)
static Jet7 edge_factor(const double t[3], const double q[4], V3 cp, V3 lpa, V3 lpb, double s) {
    JV3 jt{Jet7(t[0], 0), Jet7(t[1], 1), Jet7(t[2], 2)};
    JQ4 jq{Jet7(q[0], 3), Jet7(q[1], 4), Jet7(q[2], 5), Jet7(q[3], 6)};
    JV3 lp = jqrot(jq, jv3c(cp)) + jt;
    JV3 nu = jcross(lp - jv3c(lpa), lp - jv3c(lpb));
    JV3 de = jv3c(lpa) - jv3c(lpb);
    Jet7 r = jnorm(nu) / jnorm(de);
    r = r * Jet7(s);
    return r;
}
// LidarKeyframeFactor.h:78-92
static Jet7 plane_factor(const double t[3], const double q[4], V3 cp, V3 n, Q4 qlb, V3 tlb, double d, double score) {
    JV3 jt{Jet7(t[0], 0), Jet7(t[1], 1), Jet7(t[2], 2)};
    JQ4 jq{Jet7(q[0], 3), Jet7(q[1], 4), Jet7(q[2], 5), Jet7(q[3], 6)};
    // q_l_b.inverse() runs on Jets there (:86): Eigen's conjugate().coeffs() / squaredNorm() with ceres::Jet's operator/,
    // which multiplies by the reciprocal of the denominator (ceres/jet.h) — one ulp away from a plain division for the
    // non-unit q_lb of the configs; pinned by tests/golden/ref_factors.npz (the functor itself, compiled from the reference).
    double n2 = qlb.x * qlb.x + qlb.y * qlb.y + qlb.z * qlb.z + qlb.w * qlb.w;
    double gi = 1.0 / n2;
    Q4 qi = n2 > 0 ? Q4{qlb.w * gi, (-qlb.x) * gi, (-qlb.y) * gi, (-qlb.z) * gi} : Q4{0, 0, 0, 0};
    JQ4 jqi{Jet7(qi.w), Jet7(qi.x), Jet7(qi.y), Jet7(qi.z)};
    JV3 pw = jqrot(jqi, jv3c(cp) - jv3c(tlb));
    pw = jqrot(jq, pw) + jt;
    Jet7 r = Jet7(score) * (jdot(jv3c(n), pw) + Jet7(d));
    return r;
}
// LidarKeyframeFactor.h:118-128 (front-end; the reference's parameter order there is (q,t) —
// partial indices are still t:0..2, q:3..6 in this oracle, reordered by the caller if needed)
static Jet7 plane_incre_factor(const double t[3], const double q[4], V3 cp, V3 n, double d) {
    JV3 jt{Jet7(t[0], 0), Jet7(t[1], 1), Jet7(t[2], 2)};
    JQ4 jq{Jet7(q[0], 3), Jet7(q[1], 4), Jet7(q[2], 5), Jet7(q[3], 6)};
    JV3 pw = jqrot(jq, jv3c(cp)) + jt;
    return jdot(jv3c(n), pw) + Jet7(d);
}

// raw (un-robustified) residual + 1x7 Jacobian of one record; exposed for Jacobian tests
extern "C" void lo_eval_edge(const double t[3], const double q[4], const float cp[3], const float a[3], const float b[3], double s, double out[8]) {
    Jet7 r = edge_factor(t, q, V3{(double)cp[0], (double)cp[1], (double)cp[2]}, V3{(double)a[0], (double)a[1], (double)a[2]}, V3{(double)b[0], (double)b[1], (double)b[2]}, s);
    for (int k = 0; k < 7; k++) out[k] = r.v[k];
    out[7] = r.a;
}
extern "C" void lo_eval_plane(const double t[3], const double q[4], const float cp[3], const float n[3], float d, double score,
                   const double qlb[4], const double tlb[3], int frontend, double out[8]) {
    Jet7 r;
    V3 c{(double)cp[0], (double)cp[1], (double)cp[2]}, nn{(double)n[0], (double)n[1], (double)n[2]};
    if (frontend) r = plane_incre_factor(t, q, c, nn, (double)d);
    else r = plane_factor(t, q, c, nn, Q4{qlb[0], qlb[1], qlb[2], qlb[3]}, V3{tlb[0], tlb[1], tlb[2]}, (double)d, score);
    for (int k = 0; k < 7; k++) out[k] = r.v[k];
    out[7] = r.a;
}

// ceres::CauchyLoss / HuberLoss ::Evaluate(s, rho[3])
static inline void loss_eval(int loss, double a, double s, double rho[3]) {
    if (loss == LO_LOSS_CAUCHY) {
        double b = a * a, c = 1 / b;
        double sum = 1.0 + s * c, inv = 1.0 / sum;
        rho[0] = b * std::log(sum); rho[1] = std::max(2.2250738585072014e-308, inv); rho[2] = -c * (inv * inv);
    } else if (loss == LO_LOSS_HUBER) {
        double b = a * a;
        if (s > b) { double r = std::sqrt(s); rho[0] = 2.0 * a * r - b; rho[1] = std::max(2.2250738585072014e-308, a / r); rho[2] = -rho[1] / (2.0 * s); }
        else { rho[0] = s; rho[1] = 1.0; rho[2] = 0.0; }
    } else { rho[0] = s; rho[1] = 1.0; rho[2] = 0.0; }
}
// ResidualBlockInfo::Evaluate's corrector — L/src/MarginalizationFactor.cpp:44-70 (1 residual)
static inline void robustify(int loss, double a, double Jr[8], double* cost) {
    double sq_norm = Jr[7] * Jr[7];
    double rho[3];
    loss_eval(loss, a, sq_norm, rho);
    *cost = 0.5 * rho[0];
    if (loss == LO_LOSS_NONE) return;
    double sqrt_rho1 = std::sqrt(rho[1]);
    double residual_scaling, alpha_sq_norm;
    if (sq_norm == 0.0 || rho[2] <= 0.0) { residual_scaling = sqrt_rho1; alpha_sq_norm = 0.0; }
    else {
        double D = 1.0 + 2.0 * sq_norm * rho[2] / rho[1];
        double alpha = 1.0 - std::sqrt(D);
        residual_scaling = sqrt_rho1 / (1 - alpha);
        alpha_sq_norm = alpha / sq_norm;
    }
    for (int k = 0; k < 7; k++) Jr[k] = sqrt_rho1 * (Jr[k] - alpha_sq_norm * Jr[7] * (Jr[7] * Jr[k]));
    Jr[7] *= residual_scaling;
}

// ROT count scaling as the reference writes it (R/src/BackendFusion.cpp:843,861; pinned by oracle/_ref/libref_backend_R.so):
//   plane:  vec_surf_scores[idVec][i] * 1000 / vec_surf_res_cnt[idVec]      double * int / int  -> (score * 1000.0) / N in double
//   edge:   points[i].intensity * 200 / vec_edge_res_cnt[idVec]              float * int / int   -> (s * 200.0f) / (float)N in FLOAT
// scale_den > 0: `scale` is the numerator (1000 / 200) and scale_den the correspondence count; scale_den == 0: plain factor.
static inline double scaled_score(double score, double scale, int den) { return den > 0 ? score * scale / (double)den : score * scale; }
static inline double scaled_s(float s, double scale, int den) { return den > 0 ? (double)(s * (float)scale / (float)den) : (double)s * scale; }

// Gram accumulation G += [J r]^T [J r] over all valid records in query order (8x8 row-major, full),
// cost += 1/2 rho(r^2).
extern "C" void lo_linearize_surf(const unsigned char* valid, const float* rec_cp, const float* rec_n, const float* rec_d, const double* rec_score,
                       int n_q, const double t[3], const double q[4], const lo_params* P, double scale, int scale_den,
                       double gram[64], double* cost, int* count) {
    for (int k = 0; k < 64; k++) gram[k] = 0;
    double c = 0; int cnt = 0;
    for (int i = 0; i < n_q; i++) {
        if (!valid[i]) continue;
        double Jr[8];
        lo_eval_plane(t, q, rec_cp + 3 * (size_t)i, rec_n + 3 * (size_t)i, rec_d[i], scaled_score(rec_score[i], scale, scale_den), P->q_lb, P->t_lb,
                      P->variant == LO_VARIANT_FRONTEND, Jr);
        double ci; robustify(P->loss, P->loss_a, Jr, &ci);
        c += ci; cnt++;
        for (int a = 0; a < 8; a++) for (int b = 0; b < 8; b++) gram[a * 8 + b] += Jr[a] * Jr[b];
    }
    *cost = c; *count = cnt;
}
extern "C" void lo_linearize_edge(const unsigned char* valid, const float* rec_cp, const float* rec_a, const float* rec_b, const float* rec_s,
                       int n_q, const double t[3], const double q[4], const lo_params* P, double scale, int scale_den,
                       double gram[64], double* cost, int* count) {
    for (int k = 0; k < 64; k++) gram[k] = 0;
    double c = 0; int cnt = 0;
    for (int i = 0; i < n_q; i++) {
        if (!valid[i]) continue;
        double Jr[8];
        lo_eval_edge(t, q, rec_cp + 3 * (size_t)i, rec_a + 3 * (size_t)i, rec_b + 3 * (size_t)i, scaled_s(rec_s[i], scale, scale_den), Jr);
        double ci; robustify(P->loss, P->loss_a, Jr, &ci);
        c += ci; cnt++;
        for (int a = 0; a < 8; a++) for (int b = 0; b < 8; b++) gram[a * 8 + b] += Jr[a] * Jr[b];
    }
    *cost = c; *count = cnt;
}

// ------------------------------------------------------------------------------------------
// One Gauss-Newton step on the 6-dof local parameterisation.
// ceres::QuaternionParameterization: plus-Jacobian (4x3) and Plus(x, delta) = [cos|d|, sin|d|/|d| d] (x) x.
// Returns 0 on success, 1 if the 6x6 normal matrix is not positive definite (pose left unchanged).
// ------------------------------------------------------------------------------------------
extern "C" int lo_gn_step(const double gram[64], double t[3], double q[4], double delta_out[6]) {
    double Pm[7][6];
    for (int r = 0; r < 7; r++) for (int c = 0; c < 6; c++) Pm[r][c] = 0;
    Pm[0][0] = Pm[1][1] = Pm[2][2] = 1;
    const double x0 = q[0], x1 = q[1], x2 = q[2], x3 = q[3];
    Pm[3][3] = -x1; Pm[3][4] = -x2; Pm[3][5] = -x3;
    Pm[4][3] = x0;  Pm[4][4] = x3;  Pm[4][5] = -x2;
    Pm[5][3] = -x3; Pm[5][4] = x0;  Pm[5][5] = x1;
    Pm[6][3] = x2;  Pm[6][4] = -x1; Pm[6][5] = x0;
    double H[36], g[6];
    for (int a = 0; a < 6; a++) {
        for (int b = 0; b < 6; b++) {
            double s = 0;
            for (int i = 0; i < 7; i++) { double gi = 0; for (int j = 0; j < 7; j++) gi += gram[i * 8 + j] * Pm[j][b]; s += Pm[i][a] * gi; }
            H[a * 6 + b] = s;
        }
        double s = 0; for (int i = 0; i < 7; i++) s += Pm[i][a] * gram[i * 8 + 7];
        g[a] = -s;
    }
    if (!chol_solve(6, H, g)) return 1;
    for (int k = 0; k < 6; k++) if (!(g[k] == g[k])) return 1;
    if (delta_out) for (int k = 0; k < 6; k++) delta_out[k] = g[k];
    t[0] += g[0]; t[1] += g[1]; t[2] += g[2];
    double nd = std::sqrt(g[3] * g[3] + g[4] * g[4] + g[5] * g[5]);
    if (nd > 0.0) {
        double sbd = std::sin(nd) / nd;
        Q4 qd{std::cos(nd), sbd * g[3], sbd * g[4], sbd * g[5]};
        Q4 r = qmul(qd, Q4{q[0], q[1], q[2], q[3]});
        q[0] = r.w; q[1] = r.x; q[2] = r.y; q[3] = r.z;
    }
    return 0;
}

// small helpers exported for KATs
extern "C" int lo_eig3(const double A[9], double evals[3], double evecs[9]) {
    double M[3][3], V[3][3];
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) M[r][c] = A[3 * r + c];
    bool ok = eig3_sym(M, evals, V);
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) evecs[3 * r + c] = V[r][c];
    return ok ? 0 : 1;
}
extern "C" void lo_lstsq53(const double A[15], const double b[5], double x[3]) {
    double M[5][3];
    for (int r = 0; r < 5; r++) for (int c = 0; c < 3; c++) M[r][c] = A[3 * r + c];
    lstsq_5x3_colpiv(M, b, x);
}
extern "C" void lo_qrot(const double q[4], const double v[3], double out[3]) {
    V3 o = qrot(Q4{q[0], q[1], q[2], q[3]}, V3{v[0], v[1], v[2]});
    out[0] = o.x; out[1] = o.y; out[2] = o.z;
}
extern "C" void lo_loss(int loss, double a, double s, double rho[3]) { loss_eval(loss, a, s, rho); }


// Per-residual robustified rows [J(7) r] (test aid for the marginalisation assembly, L/src/MarginalizationFactor.cpp:3-71)
extern "C" void lo_rows_surf(const unsigned char* valid, const float* rec_cp, const float* rec_n, const float* rec_d, const double* rec_score,
                             int n_q, const double t[3], const double q[4], const lo_params* P, double scale, int scale_den, double* rows, int* count) {
    int cnt = 0;
    for (int i = 0; i < n_q; i++) {
        if (!valid[i]) continue;
        double* Jr = rows + 8 * (size_t)cnt;
        lo_eval_plane(t, q, rec_cp + 3 * (size_t)i, rec_n + 3 * (size_t)i, rec_d[i], scaled_score(rec_score[i], scale, scale_den), P->q_lb, P->t_lb,
                      P->variant == LO_VARIANT_FRONTEND, Jr);
        double ci; robustify(P->loss, P->loss_a, Jr, &ci);
        cnt++;
    }
    *count = cnt;
}
extern "C" void lo_rows_edge(const unsigned char* valid, const float* rec_cp, const float* rec_a, const float* rec_b, const float* rec_s,
                             int n_q, const double t[3], const double q[4], const lo_params* P, double scale, int scale_den, double* rows, int* count) {
    int cnt = 0;
    for (int i = 0; i < n_q; i++) {
        if (!valid[i]) continue;
        double* Jr = rows + 8 * (size_t)cnt;
        lo_eval_edge(t, q, rec_cp + 3 * (size_t)i, rec_a + 3 * (size_t)i, rec_b + 3 * (size_t)i, scaled_s(rec_s[i], scale, scale_den), Jr);
        double ci; robustify(P->loss, P->loss_a, Jr, &ci);
        cnt++;
    }
    *count = cnt;
}

// Multi-threaded variant of lo_linearize_surf for the all-cores CPU baseline: per-thread partial Grams over
// contiguous query ranges, added in thread order (the reference itself sums per-thread A, b after joining its 4
// marginalisation threads, L/src/MarginalizationFactor.cpp:159-174).
extern "C" void lo_linearize_surf_mt(const unsigned char* valid, const float* rec_cp, const float* rec_n, const float* rec_d, const double* rec_score,
                                     int n_q, const double t[3], const double q[4], const lo_params* P, double scale, int scale_den, int nthreads,
                                     double gram[64], double* cost, int* count) {
    if (nthreads < 1) nthreads = 1;
    // one partial per chunk of kChunk queries, added in chunk order: the record does not depend on the number of threads or on who drew which chunk
    const int chunks = std::max(1, (n_q + kChunk - 1) / kChunk);
    std::vector<double> G((size_t)chunks * 64, 0.0), C(chunks, 0.0);
    std::vector<int> N(chunks, 0);
    const std::function<void(int)> job = [&](int w) {
        const int lo = w * kChunk, hi = std::min(n_q, lo + kChunk);
        double* g = G.data() + (size_t)w * 64; double c = 0; int cnt = 0;
        for (int i = lo; i < hi; i++) {
            if (!valid[i]) continue;
            double Jr[8];
            lo_eval_plane(t, q, rec_cp + 3 * (size_t)i, rec_n + 3 * (size_t)i, rec_d[i], scaled_score(rec_score[i], scale, scale_den), P->q_lb, P->t_lb,
                          P->variant == LO_VARIANT_FRONTEND, Jr);
            double ci; robustify(P->loss, P->loss_a, Jr, &ci);
            c += ci; cnt++;
            for (int a = 0; a < 8; a++) for (int b = 0; b < 8; b++) g[a * 8 + b] += Jr[a] * Jr[b];
        }
        C[w] = c; N[w] = cnt;
    };
    if (nthreads <= 1 || chunks < 2) for (int w = 0; w < chunks; w++) job(w);
    else pool().run(chunks, nthreads, job);
    for (int k = 0; k < 64; k++) gram[k] = 0;
    double c = 0; int cnt = 0;
    for (int w = 0; w < chunks; w++) { for (int k = 0; k < 64; k++) gram[k] += G[(size_t)w * 64 + k]; c += C[w]; cnt += N[w]; }
    *cost = c; *count = cnt;
}

// ------------------------------------------------------------------------------------------
// One scan registration entirely in C (round 5, the CPU baseline of bench.py): `n_iters` outer iterations of the ROT / Livox / front-end surf matcher
// — association pose from the body pose (L/src/BackendFusion.cpp:929-930), findCorrespondingSurfFeatures over all queries, linearisation with the count scaling
// of R/src/BackendFusion.cpp:861 when scale_num > 0, one Gauss-Newton step — on the persistent pool, no allocation and no interpreter between the stages.
// The same functions as the per-stage entry points above; the record of every iteration is the chunk-ordered sum of lo_linearize_surf_mt.
// t, q: body pose, in / out.  Returns the number of iterations whose step was applied; counts_out (optional, n_iters ints) receives the correspondences per iteration.
// ------------------------------------------------------------------------------------------
extern "C" int lo_register_surf(void* tree, const float* map_xyz, int n_map, const float* q_xyz, int n_q, double t[3], double q[4], const lo_params* P,
                                double scale_num, int n_iters, int nthreads, int* counts_out) {
    std::vector<unsigned char> valid((size_t)n_q);
    std::vector<int> nn_idx((size_t)n_q * 5);
    std::vector<float> nn_d2((size_t)n_q * 5), cp((size_t)n_q * 3), nrm((size_t)n_q * 3), d((size_t)n_q);
    std::vector<double> score((size_t)n_q);
    int applied = 0;
    for (int it = 0; it < n_iters; it++) {
        Q4 Q{q[0], q[1], q[2], q[3]}, Q2 = Q; V3 T{t[0], t[1], t[2]}, T2 = T;
        if (P->variant != LO_VARIANT_FRONTEND) {
            const double n2 = P->q_lb[0] * P->q_lb[0] + P->q_lb[1] * P->q_lb[1] + P->q_lb[2] * P->q_lb[2] + P->q_lb[3] * P->q_lb[3];
            const Q4 qi{P->q_lb[0] / n2, -P->q_lb[1] / n2, -P->q_lb[2] / n2, -P->q_lb[3] / n2};
            Q2 = qmul(Q, qi);
            T2 = T - qrot(Q2, V3{P->t_lb[0], P->t_lb[1], P->t_lb[2]});
        }
        const double pq[4] = {Q2.w, Q2.x, Q2.y, Q2.z}, pt[3] = {T2.x, T2.y, T2.z};
        const int cnt = lo_associate_surf(tree, map_xyz, nullptr, n_map, q_xyz, nullptr, n_q, pq, pt, P, nthreads, valid.data(), nn_idx.data(), nn_d2.data(),
                                          cp.data(), nrm.data(), d.data(), score.data());
        if (counts_out) counts_out[it] = cnt;
        double gram[64], cost; int c2;
        lo_linearize_surf_mt(valid.data(), cp.data(), nrm.data(), d.data(), score.data(), n_q, t, q, P, scale_num > 0 ? scale_num : 1.0, scale_num > 0 ? std::max(cnt, 1) : 0, nthreads,
                             gram, &cost, &c2);
        if (lo_gn_step(gram, t, q, nullptr) == 0) applied++;
    }
    return applied;
}

