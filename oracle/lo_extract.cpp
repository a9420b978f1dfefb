// ORACLE — TEST INFRASTRUCTURE ONLY (see lo_math.h header).  PARITY: the literal mode of both extractors reproduces every cloud
// the reference's own Preprocessing.cpp publishes bit for bit (oracle/_ref: libref_rot, libref_livox — tests/test_reference_cpu.py);
// PCL's VoxelGrid and Eigen's quaternion / eigen-solver arithmetic are restated third-party code.
//
// CPU restatement of the reference's feature extractors:
//   * LOAM-style 16/32/64-ring extractor      — R/src/Preprocessing.cpp:120-177 (filters, deskew), :277-509
//   * pcl::VoxelGrid<PointXYZI>::applyFilter  — third-party (PCL >= 1.8 semantics, App. B2): call site R:502-508
//   * Livox Horizon 6 x 4000 grid extractor   — L/src/Preprocessing.cpp:72-127, :219-383
// (L/ = /root/reference/LiLi-OM/, R/ = /root/reference/LiLi-OM-ROT/)
//
// Float widths follow the reference expression by expression: `using namespace std`
// makes sqrt/atan/atan2/fabs on float arguments the FLOAT overloads; mixed int/float/double literals promote
// as C++ says.  Build: -ffp-contract=off, no fast-math.
#include "lo_math.h"
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <vector>

using namespace lo;

namespace {

struct P4 { float x, y, z, i; };

// atan / atan2 on float arguments.  mode 0: the float overloads of this libm (what the reference runs on a given machine);
// mode 2: glibc's fdlibm float routines restated (lo_math.h: fd_atanf / fd_atan2f, bit-identical to this image's libm on every
// input — the definition the GPU path uses by default); mode 1: the double function rounded to float (correctly rounded up
// to ~1e-9 probability; libm-independent, the GPU path's "rot_atan" = 1 option).
static inline float atan_f(float v, int mode) { return mode == 2 ? fd_atanf(v) : mode ? (float)std::atan((double)v) : std::atan(v); }
static inline float atan2_f(float y, float x, int mode) { return mode == 2 ? fd_atan2f(y, x) : mode ? (float)std::atan2((double)y, (double)x) : std::atan2(y, x); }

// undistortion — R/src/Preprocessing.cpp:153-177 (with_lb) and L/src/Preprocessing.cpp:104-127
static inline P4 undistort(P4 pt, Q4 q_imu, Q4 q_lb, bool with_lb) {
    double dt = 0.1;
    int line = int(pt.i);
    double dt_i = pt.i - line;          // float - int -> float, then widened
    double ratio_i = dt_i / dt;
    if (ratio_i >= 1.0) ratio_i = 1.0;
    Q4 q0{1, 0, 0, 0};
    Q4 q_si = qslerp(q0, ratio_i, q_imu);
    if (with_lb) q_si = qmul(qmul(q_lb, q_si), qinv(q_lb));
    V3 r = qrot(q_si, V3{(double)pt.x, (double)pt.y, (double)pt.z});
    return P4{(float)r.x, (float)r.y, (float)r.z, pt.i};
}

struct VoxIdx { unsigned idx; int pi; bool operator<(const VoxIdx& o) const { return idx < o.idx; } };

// pcl::VoxelGrid<PointXYZI>::applyFilter, downsample_all_data_ = true, no field limits, min_points_per_voxel_ = 0.
// stable = false: std::sort exactly as PCL (order of points inside a voxel = libstdc++ introsort's);
// stable = true : points inside a voxel are accumulated in input order (the GPU path's definition).
static void voxel_grid(const std::vector<P4>& in, float leaf, bool stable, std::vector<P4>& out, std::vector<int>* counts) {
    out.clear();
    if (counts) counts->clear();
    if (in.empty()) return;
    float inv = 1.0f / leaf;                       // inverse_leaf_size_ = Array4f::Ones() / leaf_size_
    float mn[3] = {in[0].x, in[0].y, in[0].z}, mx[3] = {in[0].x, in[0].y, in[0].z};
    for (const P4& p : in) {                       // getMinMax3D
        mn[0] = std::min(mn[0], p.x); mn[1] = std::min(mn[1], p.y); mn[2] = std::min(mn[2], p.z);
        mx[0] = std::max(mx[0], p.x); mx[1] = std::max(mx[1], p.y); mx[2] = std::max(mx[2], p.z);
    }
    int min_b[3], max_b[3], div_b[3];
    for (int k = 0; k < 3; k++) {
        min_b[k] = static_cast<int>(std::floor(mn[k] * inv));
        max_b[k] = static_cast<int>(std::floor(mx[k] * inv));
        div_b[k] = max_b[k] - min_b[k] + 1;
    }
    int mul[3] = {1, div_b[0], div_b[0] * div_b[1]};
    std::vector<VoxIdx> iv; iv.reserve(in.size());
    for (size_t i = 0; i < in.size(); i++) {
        const P4& p = in[i];
        int ijk0 = static_cast<int>(std::floor(p.x * inv) - static_cast<float>(min_b[0]));
        int ijk1 = static_cast<int>(std::floor(p.y * inv) - static_cast<float>(min_b[1]));
        int ijk2 = static_cast<int>(std::floor(p.z * inv) - static_cast<float>(min_b[2]));
        int idx = ijk0 * mul[0] + ijk1 * mul[1] + ijk2 * mul[2];
        iv.push_back(VoxIdx{static_cast<unsigned>(idx), (int)i});
    }
    if (stable) std::stable_sort(iv.begin(), iv.end()); else std::sort(iv.begin(), iv.end());
    size_t a = 0;
    while (a < iv.size()) {
        size_t b = a + 1;
        while (b < iv.size() && iv[b].idx == iv[a].idx) b++;
        // CentroidPoint<PointXYZI>: float accumulators, divided by the count
        float sx = 0, sy = 0, sz = 0, si = 0;
        for (size_t k = a; k < b; k++) { const P4& p = in[iv[k].pi]; sx += p.x; sy += p.y; sz += p.z; si += p.i; }
        float n = static_cast<float>(b - a);
        out.push_back(P4{sx / n, sy / n, sz / n, si / n});
        if (counts) counts->push_back((int)(b - a));
        a = b;
    }
}

}  // namespace

struct lo_rot_params {
    int n_scans;      // 16 / 32 / 64 (R:46, "line_num")
    int ds_rate;      // R/config/config_fr_iosb.yaml:13
    float ds_v;       // 0.6 (R:14)
    float near_thres; // 3.0 (R:281)
    int atan_mode;    // 0 = libm float overloads (literal), 1 = double function rounded to float, 2 = glibc fdlibm float routines restated
    int stable_sort;  // 0 = std::sort (literal), 1 = ties broken by index
};

// Outputs (all caller-allocated with capacity n):
//   full[n_full*4]        concatenated, deskewed ring cloud (x,y,z,intensity = ring + 0.1*relTime)   — /lidar_cloud_cutted
//   full_src[n_full]      index of each full-cloud point in the input
//   ring_start/ring_end   scanStartInd / scanEndInd per ring (size n_scans)
//   curvature/label[n_full]  (label: 2 sharp, 1 less sharp, -1 flat, 0 other; entries outside [5, n-5) are 0)
//   edge_idx[n_edge]      cornerPointsLessSharp as indices into full, in push order                  — /edge_features
//   sharp_idx, flat_idx   cornerPointsSharp / surfPointsFlat (built by the reference, not published)
//   surf[n_surf*4]        voxel-filtered less-flat points, ring after ring                            — /surf_features
//   surf_cnt[n_surf]      points per output voxel;  lessflat_idx[n_lessflat]: indices into full before the voxel filter
//   n_ties                number of equal-curvature neighbours seen inside sorted segments (std::sort is unstable there)
extern "C" int lo_extract_rot(const float* pts, int n, const double q_imu_[4], const double q_lb_[4], const lo_rot_params* P,
                              float* full, int* full_src, int* n_full, int* ring_start, int* ring_end,
                              float* curvature, int* label,
                              int* edge_idx, int* n_edge, int* sharp_idx, int* n_sharp, int* flat_idx, int* n_flat,
                              float* surf, int* surf_cnt, int* n_surf, int* lessflat_idx, int* n_lessflat, int* n_ties) {
    const int N_SCANS = P->n_scans;
    Q4 qIMU{q_imu_[0], q_imu_[1], q_imu_[2], q_imu_[3]}, q_lb{q_lb_[0], q_lb_[1], q_lb_[2], q_lb_[3]};
    *n_full = *n_edge = *n_sharp = *n_flat = *n_surf = *n_lessflat = *n_ties = 0;
    for (int i = 0; i < N_SCANS; i++) { ring_start[i] = 0; ring_end[i] = 0; }
    // removeNaNFromPointCloud + removeClosedPointCloud (R:280-281, :120-145)
    std::vector<P4> in; std::vector<int> src;
    in.reserve(n); src.reserve(n);
    float thres = P->near_thres;
    for (int i = 0; i < n; i++) {
        P4 p{pts[4 * i], pts[4 * i + 1], pts[4 * i + 2], pts[4 * i + 3]};
        if (!std::isfinite(p.x) || !std::isfinite(p.y) || !std::isfinite(p.z)) continue;
        if (p.x * p.x + p.y * p.y + p.z * p.z < thres * thres) continue;
        in.push_back(p); src.push_back(i);
    }
    int cloudSize = (int)in.size();
    if (cloudSize == 0) return 0;     // the reference would index points[0] here
    const int am = P->atan_mode;
    float startOri = -atan2_f(in[0].y, in[0].x, am);                                          // R:285
    float endOri = (float)(-atan2_f(in[cloudSize - 1].y, in[cloudSize - 1].x, am) + 2 * M_PI); // R:286-288
    if (endOri - startOri > 3 * M_PI) endOri = (float)(endOri - 2 * M_PI);                    // R:290-294
    else if (endOri - startOri < M_PI) endOri = (float)(endOri + 2 * M_PI);
    bool halfPassed = false;
    int count = cloudSize;
    std::vector<std::vector<P4>> scans(N_SCANS);
    std::vector<std::vector<int>> scans_src(N_SCANS);
    for (int i = 0; i < cloudSize; i++) {                                                     // R:308-372
        P4 point{in[i].x, in[i].y, in[i].z, (float)(0.1 * in[i].i)};
        // atan(float)*180 is a float product, / M_PI a double division, the result narrows to float (R:315)
        float angle = (float)((atan_f(point.z / std::sqrt(point.x * point.x + point.y * point.y), am) * 180) / M_PI);
        int scanID = 0;
        if (N_SCANS == 16) {
            scanID = int((angle + 15) / 2 + 0.5);
            if (scanID > (N_SCANS - 1) || scanID < 0) { count--; continue; }
        } else if (N_SCANS == 32) {
            scanID = int((angle + 92.0 / 3.0) * 3.0 / 4.0);
            if (scanID > (N_SCANS - 1) || scanID < 0) { count--; continue; }
        } else if (N_SCANS == 64) {
            if (angle >= -8.83) scanID = int((2 - angle) * 3.0 + 0.5);
            else scanID = N_SCANS / 2 + int((-8.83 - angle) * 2.0 + 0.5);
            if (angle > 2 || angle < -24.33 || scanID > 50 || scanID < 0) { count--; continue; }
        } else return -1;
        float ori = -atan2_f(point.y, point.x, am);
        if (!halfPassed) {
            if (ori < startOri - M_PI / 2) ori = (float)(ori + 2 * M_PI);
            else if (ori > startOri + M_PI * 3 / 2) ori = (float)(ori - 2 * M_PI);
            if (ori - startOri > M_PI) halfPassed = true;
        } else {
            ori = (float)(ori + 2 * M_PI);
            if (ori < endOri - M_PI * 3 / 2) ori = (float)(ori + 2 * M_PI);
            else if (ori > endOri + M_PI / 2) ori = (float)(ori - 2 * M_PI);
        }
        float relTime = (ori - startOri) / (endOri - startOri);
        point.i = (float)(scanID + 0.1 * relTime);
        P4 und = undistort(point, qIMU, q_lb, true);
        scans[scanID].push_back(und);
        scans_src[scanID].push_back(src[i]);
    }
    cloudSize = count;
    std::vector<P4> cloud; cloud.reserve(cloudSize);
    std::vector<int> csrc; csrc.reserve(cloudSize);
    for (int i = 0; i < N_SCANS; i++) {                                                       // R:378-382
        ring_start[i] = (int)cloud.size() + 5;
        cloud.insert(cloud.end(), scans[i].begin(), scans[i].end());
        csrc.insert(csrc.end(), scans_src[i].begin(), scans_src[i].end());
        ring_end[i] = (int)cloud.size() - 6;
    }
    *n_full = cloudSize;
    for (int i = 0; i < cloudSize; i++) {
        full[4 * i] = cloud[i].x; full[4 * i + 1] = cloud[i].y; full[4 * i + 2] = cloud[i].z; full[4 * i + 3] = cloud[i].i;
        full_src[i] = csrc[i]; curvature[i] = 0.f; label[i] = 0;
    }
    std::vector<float> curv(cloudSize, 0.f);
    std::vector<int> sortInd(cloudSize, 0), picked(cloudSize, 0), lab(cloudSize, 0);
    for (int i = 5; i < cloudSize - 5; i++) {                                                 // R:385-394, strictly left to right in f32
        float dX = cloud[i - 5].x + cloud[i - 4].x + cloud[i - 3].x + cloud[i - 2].x + cloud[i - 1].x - 10 * cloud[i].x + cloud[i + 1].x + cloud[i + 2].x + cloud[i + 3].x + cloud[i + 4].x + cloud[i + 5].x;
        float dY = cloud[i - 5].y + cloud[i - 4].y + cloud[i - 3].y + cloud[i - 2].y + cloud[i - 1].y - 10 * cloud[i].y + cloud[i + 1].y + cloud[i + 2].y + cloud[i + 3].y + cloud[i + 4].y + cloud[i + 5].y;
        float dZ = cloud[i - 5].z + cloud[i - 4].z + cloud[i - 3].z + cloud[i - 2].z + cloud[i - 1].z - 10 * cloud[i].z + cloud[i + 1].z + cloud[i + 2].z + cloud[i + 3].z + cloud[i + 4].z + cloud[i + 5].z;
        curv[i] = dX * dX + dY * dY + dZ * dZ;
        sortInd[i] = i; picked[i] = 0; lab[i] = 0;
    }
    auto gap2 = [&](int a, int b) {   // R:435-438: float differences and float sum, compared with the double 0.05
        float dX = cloud[a].x - cloud[b].x, dY = cloud[a].y - cloud[b].y, dZ = cloud[a].z - cloud[b].z;
        return dX * dX + dY * dY + dZ * dZ;
    };
    auto range2 = [&](int k) { return cloud[k].x * cloud[k].x + cloud[k].y * cloud[k].y + cloud[k].z * cloud[k].z; };
    int ne = 0, ns = 0, nf = 0, nsurf = 0, nlf = 0, ties = 0;
    for (int i = 0; i < N_SCANS; i++) {                                                       // R:401-509
        if (ring_end[i] - ring_start[i] < 6 || i % P->ds_rate != 0) continue;
        std::vector<P4> lessFlatScan;
        for (int j = 0; j < 6; j++) {
            int sp = ring_start[i] + (ring_end[i] - ring_start[i]) * j / 6;
            int ep = ring_start[i] + (ring_end[i] - ring_start[i]) * (j + 1) / 6 - 1;
            auto comp = [&](int a, int b) { return curv[a] < curv[b]; };
            if (P->stable_sort) std::stable_sort(sortInd.begin() + sp, sortInd.begin() + ep + 1, comp);
            else std::sort(sortInd.begin() + sp, sortInd.begin() + ep + 1, comp);
            for (int k = sp; k < ep; k++) if (curv[sortInd[k]] == curv[sortInd[k + 1]]) ties++;
            int largestPickedNum = 0;
            for (int k = ep; k >= sp; k--) {
                int ind = sortInd[k];
                if (picked[ind] == 0 && curv[ind] > 2.0) {
                    largestPickedNum++;
                    if (largestPickedNum <= 2) { lab[ind] = 2; sharp_idx[ns++] = ind; edge_idx[ne++] = ind; }
                    else if (largestPickedNum <= 10) { lab[ind] = 1; edge_idx[ne++] = ind; }
                    else break;
                    picked[ind] = 1;
                    for (int l = 1; l <= 5; l++) { if (gap2(ind + l, ind + l - 1) > 0.05) break; picked[ind + l] = 1; }
                    for (int l = -1; l >= -5; l--) { if (gap2(ind + l, ind + l + 1) > 0.05) break; picked[ind + l] = 1; }
                }
            }
            int smallestPickedNum = 0;
            for (int k = sp; k <= ep; k++) {
                int ind = sortInd[k];
                if (range2(ind) < 0.25) continue;
                if (picked[ind] == 0 && curv[ind] < 0.1) {
                    lab[ind] = -1; flat_idx[nf++] = ind;
                    smallestPickedNum++;
                    if (smallestPickedNum >= 4) break;             // before marking the neighbours (R:468-470)
                    picked[ind] = 1;
                    for (int l = 1; l <= 5; l++) { if (gap2(ind + l, ind + l - 1) > 0.05) break; picked[ind + l] = 1; }
                    for (int l = -1; l >= -5; l--) { if (gap2(ind + l, ind + l + 1) > 0.05) break; picked[ind + l] = 1; }
                }
            }
            for (int k = sp; k <= ep; k++) {
                if (range2(k) < 0.25) continue;
                if (lab[k] <= 0) { lessFlatScan.push_back(cloud[k]); lessflat_idx[nlf++] = k; }
            }
        }
        std::vector<P4> ds; std::vector<int> cnts;
        voxel_grid(lessFlatScan, P->ds_v, P->stable_sort != 0, ds, &cnts);
        for (size_t k = 0; k < ds.size(); k++) {
            surf[4 * nsurf] = ds[k].x; surf[4 * nsurf + 1] = ds[k].y; surf[4 * nsurf + 2] = ds[k].z; surf[4 * nsurf + 3] = ds[k].i;
            surf_cnt[nsurf] = cnts[k]; nsurf++;
        }
    }
    for (int i = 0; i < cloudSize; i++) { curvature[i] = curv[i]; label[i] = lab[i]; }
    *n_edge = ne; *n_sharp = ns; *n_flat = nf; *n_surf = nsurf; *n_lessflat = nlf; *n_ties = ties;
    return 0;
}

// stand-alone voxel filter (KATs)
// Pin of the fdlibm restatement against THIS libm on a pseudo-random sample (the exhaustive run is tools/check_fdlibm_atan.cpp):
// returns the number of inputs on which fd_atanf / fd_atan2f and atanf / atan2f differ in a bit (NaN results compare equal).
extern "C" long long lo_fd_atan_mismatches(long long n, unsigned long long seed) {
    unsigned long long x = seed * 0x9E3779B97F4A7C15ull + 1ull;
    auto next = [&]() { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return x; };
    long long bad = 0;
    for (long long i = 0; i < n; i++) {
        const unsigned long long r = next();
        float a, b;
        if (i & 1) { a = bitsf((uint32_t)r); b = bitsf((uint32_t)(r >> 32)); }                          // any bit patterns
        else { a = (float)((double)(r & 0xffffff) / 16777216.0 * 400.0 - 200.0); b = (float)((double)((r >> 32) & 0xffffff) / 16777216.0 * 400.0 - 200.0); }   // lidar coordinates
        const float l1 = std::atan(a), f1 = fd_atanf(a), l2 = std::atan2(a, b), f2 = fd_atan2f(a, b);
        if (fbits(l1) != fbits(f1) && !(l1 != l1 && f1 != f1)) bad++;
        if (fbits(l2) != fbits(f2) && !(l2 != l2 && f2 != f2)) bad++;
    }
    return bad;
}

extern "C" int lo_voxel_grid(const float* pts, int n, float leaf, int stable, float* out, int* counts) {
    std::vector<P4> in(n), o; std::vector<int> c;
    for (int i = 0; i < n; i++) in[i] = P4{pts[4 * i], pts[4 * i + 1], pts[4 * i + 2], pts[4 * i + 3]};
    voxel_grid(in, leaf, stable != 0, o, &c);
    for (size_t k = 0; k < o.size(); k++) { out[4 * k] = o[k].x; out[4 * k + 1] = o[k].y; out[4 * k + 2] = o[k].z; out[4 * k + 3] = o[k].i; counts[k] = c[k]; }
    return (int)o.size();
}

// =================================================================================================
// Livox Horizon extractor — L/src/Preprocessing.cpp:219-383
// =================================================================================================
struct lo_livox_params {
    double surf_thres;   // L/config/config_fr_iosb.yaml:5 (0.28)
    double edge_thres;   // :6 (4)
    float near_thres;    // 0.1 (L:226)
};

namespace {
struct PN { float x, y, z, nx, ny, nz, intensity, curvature; };   // pcl::PointXYZINormal payload
static inline void canon3(double v[3]) {
    int k = 0; double m = std::fabs(v[0]);
    if (std::fabs(v[1]) > m) { m = std::fabs(v[1]); k = 1; }
    if (std::fabs(v[2]) > m) { k = 2; }
    if (v[k] < 0) { v[0] = -v[0]; v[1] = -v[1]; v[2] = -v[2]; }
}
static inline double depth_of(const PN& p) { return (double)std::sqrt(p.x * p.x + p.y * p.y + p.z * p.z); }   // getDepth: float sqrt, widened (App. A1)
}  // namespace

// pts: n x 5 floats (x, y, z, intensity = line + 0.1 * t, curvature = 0.1 * reflectivity) — FormatConvert's layout
// (L/src/FormatConvert.cpp:14-23).  Outputs, 8 floats per point (x,y,z,nx,ny,nz,intensity,curvature):
//   cutted[n_cut*8]  every undistorted point with a valid line  (/lidar_cloud_cutted), cut_src its input index
//   edge[n_edge*8], edge_cell[n_edge] (line * 4000 + column)    (/edge_features)
//   surf[n_surf*8], surf_cell[n_surf]                           (/surf_features)
//   cell_src[24000]  input index of the point that won each grid cell, or -1
// The eigenvector sign (stored as normal / direction) is canonicalised: Eigen's is arbitrary (App. B4).
extern "C" int lo_extract_livox(const float* pts, int n, const double q_imu_[4], const lo_livox_params* P,
                                float* cutted, int* cut_src, int* n_cut, float* edge, int* edge_cell, int* n_edge,
                                float* surf, int* surf_cell, int* n_surf, int* cell_src) {
    const int N_SCANS = 6, H_SCANS = 4000;
    Q4 qIMU{q_imu_[0], q_imu_[1], q_imu_[2], q_imu_[3]};
    std::vector<PN> mat((size_t)N_SCANS * H_SCANS, PN{0, 0, 0, 0, 0, 0, 0, 0});   // PCL point ctor zero-initialises (L:238)
    auto M = [&](int k, int c) -> PN& { return mat[(size_t)k * H_SCANS + c]; };
    for (int c = 0; c < N_SCANS * H_SCANS; c++) cell_src[c] = -1;
    double t_interval = 0.1 / (H_SCANS - 1);
    int nc = 0;
    float thres = P->near_thres;
    for (int i = 0; i < n; i++) {
        P4 p{pts[5 * i], pts[5 * i + 1], pts[5 * i + 2], pts[5 * i + 3]};
        float curv = pts[5 * i + 4];
        if (!std::isfinite(p.x) || !std::isfinite(p.y) || !std::isfinite(p.z)) continue;       // L:225
        if (p.x * p.x + p.y * p.y + p.z * p.z < thres * thres) continue;                        // L:226
        int scan_id = (int)p.i;                                                                  // L:252
        if (scan_id < 0) continue;
        if (scan_id >= N_SCANS) continue;    // the reference would index mat[] out of bounds; Horizon lines are 0..5
        P4 u = undistort(p, qIMU, Q4{1, 0, 0, 0}, false);                                        // L:256
        PN pu{u.x, u.y, u.z, 0, 0, 0, u.i, curv};
        float* o = cutted + 8 * (size_t)nc;
        o[0] = pu.x; o[1] = pu.y; o[2] = pu.z; o[3] = 0; o[4] = 0; o[5] = 0; o[6] = pu.intensity; o[7] = pu.curvature;
        cut_src[nc++] = i;
        double dep = pu.x * pu.x + pu.y * pu.y + pu.z * pu.z;                                    // float expression widened (L:259)
        if (dep > 40000.0 || dep < 4.0 || pu.curvature < 0.05 || pu.curvature > 25.45) continue;
        int col = int(std::round((pu.intensity - scan_id) / t_interval));                        // L:262
        if (col >= H_SCANS || col < 0) continue;
        if (M(scan_id, col).curvature != 0) continue;                                            // first writer wins
        M(scan_id, col) = pu;
        cell_src[scan_id * H_SCANS + col] = i;
    }
    *n_cut = nc;
    int ne = 0, ns = 0;
    for (int i = 5; i < H_SCANS - 12; i = i + 6) {                                              // L:270
        V3 center{0, 0, 0};
        int num = 36;
        std::vector<V3> near_pts;
        for (int j = 0; j < 6; j++) for (int k = 0; k < N_SCANS; k++) {
            if (M(k, i + j).curvature <= 0) { num--; continue; }
            V3 pt{(double)M(k, i + j).x, (double)M(k, i + j).y, (double)M(k, i + j).z};
            center = center + pt; near_pts.push_back(pt);
        }
        if (num < 25) continue;
        center = V3{center.x / num, center.y / num, center.z / num};
        double A1[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
        for (size_t j = 0; j < near_pts.size(); j++) {
            V3 z = near_pts[j] - center; double zz[3] = {z.x, z.y, z.z};
            for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) A1[r][c] += zz[r] * zz[c];
        }
        double ev[3], evec[3][3];
        eig3_sym(A1, ev, evec);
        std::vector<int> idsx, idsy;
        for (int k = 0; k < N_SCANS; k++) {                                                      // L:302-331
            double max_s = 0; int idx = i;
            for (int j = 0; j < 6; j++) {
                if (M(k, i + j).curvature <= 0) continue;
                double g1 = depth_of(M(k, i + j - 4)) + depth_of(M(k, i + j - 3)) + depth_of(M(k, i + j - 2)) + depth_of(M(k, i + j - 1)) - 8 * depth_of(M(k, i + j)) +
                            depth_of(M(k, i + j + 1)) + depth_of(M(k, i + j + 2)) + depth_of(M(k, i + j + 3)) + depth_of(M(k, i + j + 4));
                g1 = g1 / (8 * depth_of(M(k, i + j)) + 1e-3);
                if (g1 > 0.06) { if (g1 > max_s) { max_s = g1; idx = i + j; } }
            }
            if (max_s != 0) { idsx.push_back(k); idsy.push_back(idx); }
        }
        V3 ce{0, 0, 0};
        std::vector<V3> near_e;
        for (size_t j = 0; j < idsx.size(); j++) {
            V3 pt{(double)M(idsx[j], idsy[j]).x, (double)M(idsx[j], idsy[j]).y, (double)M(idsx[j], idsy[j]).z};
            ce = ce + pt; near_e.push_back(pt);
        }
        double ne_d = (double)idsx.size();
        ce = V3{ce.x / ne_d, ce.y / ne_d, ce.z / ne_d};                                          // 0/0 = NaN when there is no candidate (App. A6)
        double AE[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
        for (size_t j = 0; j < near_e.size(); j++) {
            V3 z = near_e[j] - ce; double zz[3] = {z.x, z.y, z.z};
            for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) AE[r][c] += zz[r] * zz[c];
        }
        double eve[3], evece[3][3];
        eig3_sym(AE, eve, evece);
        if (eve[2] > P->edge_thres * eve[1] && idsx.size() > 3) {                                // L:353
            double u[3] = {evece[2][0], evece[2][1], evece[2][2]};
            canon3(u);
            for (size_t j = 0; j < idsx.size(); j++) {
                PN& m = M(idsx[j], idsy[j]);
                if (m.curvature <= 0 && m.intensity <= 0) continue;
                m.nx = (float)u[0]; m.ny = (float)u[1]; m.nz = (float)u[2];
                float* o = edge + 8 * (size_t)ne;
                o[0] = m.x; o[1] = m.y; o[2] = m.z; o[3] = m.nx; o[4] = m.ny; o[5] = m.nz; o[6] = m.intensity; o[7] = m.curvature;
                edge_cell[ne++] = idsx[j] * H_SCANS + idsy[j];
                m.curvature *= -1;
            }
        }
        if (ev[0] < P->surf_thres * ev[1]) {                                                     // L:367
            double u[3] = {evec[0][0], evec[0][1], evec[0][2]};
            canon3(u);
            for (int j = 0; j < 6; j++) for (int k = 0; k < N_SCANS; k++) {
                PN& m = M(k, i + j);
                if (m.curvature <= 0) continue;
                m.nx = (float)u[0]; m.ny = (float)u[1]; m.nz = (float)u[2];
                float* o = surf + 8 * (size_t)ns;
                o[0] = m.x; o[1] = m.y; o[2] = m.z; o[3] = m.nx; o[4] = m.ny; o[5] = m.nz; o[6] = m.intensity; o[7] = m.curvature;
                surf_cell[ns++] = k * H_SCANS + (i + j);
                m.curvature *= -1;
            }
        }
    }
    *n_edge = ne; *n_surf = ns;
    return 0;
}
