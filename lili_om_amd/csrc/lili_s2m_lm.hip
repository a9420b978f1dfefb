// Device Levenberg-Marquardt on fixed correspondences, ONE launch.
//
// What it replaces: the reference hands the residual blocks of a keyframe to ceres::Solve (L/src/BackendFusion.cpp:984-992: DENSE_QR,
// max_num_iterations = max_num_iter (15), everything else Ceres 2.0 defaults) — a trust-region loop that EVALUATES all
// blocks once per iteration and accepts or rejects the step by the robust cost.  lili_s2m_iterate_inner ran plain Gauss-Newton steps,
// one launch per evaluation (14.7 us at 200 k records); a caller that wanted Ceres' accept / reject had to come back to the host
// after every evaluation.  Here the whole loop is one persistent kernel:
//
//   evaluation   every workgroup linearises its share of the records at the candidate pose (lin_surf_body / lin_edge_body of
//                lili_s2m_dev.h: robustified rows, f64-MFMA Gram, block partial incl. the robust cost) and PUBLISHES the 40-double partial
//                as 16-byte granules {value, value ^ key}, key unique per (launch, evaluation): the data is its own flag
//                (cdna_hip_programming.md Guideline 16 form R2: one write-through store per granule, relaxed agent-scope loads, no fence)
//   exchange     by ONE polling wave per workgroup.  Up to 16 workgroups (8 k records): one hop — everybody adds all partials in index order.
//                More: two hops — the first workgroup of every group of 16 adds its members' partials in index order and publishes the
//                group sum; every workgroup then adds the <= 15 group sums in index order.  Fixed order,
//                so every workgroup holds the SAME 40 doubles bit for bit — no broadcast of the decision is needed: each workgroup
//                runs the trust-region step itself (6x6 LDL^T in one lane) and arrives at the same candidate pose.  Partial / group
//                buffers are double-buffered by the parity of the evaluation: a workgroup can start evaluation e + 1 only after it has
//                read every partial of e, so nobody overwrites what somebody still reads.
//   step logic   TrustRegionMinimizer + LevenbergMarquardtStrategy of Ceres 2.0 (the checker's restatement is what tests/test_lm_gpu.py compares with): Jacobi scaling from
//                the first Jacobian, D^2 = clamp(diag) / radius, model cost change, relative decrease rho, parameter / function /
//                gradient tolerances checked in Ceres' order, radius update radius / max(1/3, 1 - (2 rho - 1)^3) or halving with a
//                doubling divisor.  The linear system is solved on the 6x6 normal equations (the Gram is what the evaluation produces);
//                Ceres' DENSE_QR works on the stacked rows — the same minimiser up to rounding, which tests/test_lm_gpu.py bounds by
//                comparing every accept / reject decision and the final pose with the CPU restatement's loop on per-residual rows.
//
// All workgroups must be co-resident (<= one per CU: 512 threads at up to 256 VGPRs, 48 KB LDS); every wait is bounded and ends the launch with
// termination = LILI_LM_STALLED instead of hanging the GPU.
#include "lili_s2m_dev.h"
#include "../../include/lili_hip.h"

namespace lili {

constexpr int kLmThreads = 512;      // 8 waves: the launch may use 256 VGPRs per lane (1024-thread workgroups cap it at 128 and the loop spilled ~150 words)
constexpr int kLmGroup = 16;      // workgroups per group sum; up to this many workgroups exchange in ONE hop (32: measured slower, one wave polls 1280 granules)

struct LmArgs {
    LinArgs S, E;                 // records of the two kinds; S.nb / E.nb = workgroups of each kind (either may be 0)
    SlotState* state;
    double* part;                 // [2 parities][nb][kPartialStride]   block partials as granules
    double* gsum;                 // [2 parities][ng][kPartialStride]   group sums as granules
    int nb, ng;
    int max_iter;
    unsigned long long launch;    // host counter: makes the granule keys of this launch unique
    lili_lm_summary* summary;     // device copy, written by workgroup 0
    double function_tolerance, gradient_tolerance, parameter_tolerance;
    double initial_radius, max_radius, min_radius, min_relative_decrease, min_lm_diagonal, max_lm_diagonal;
};
struct WinLmArgs { LmArgs a[kWindowMaxSlots]; int first_block[kWindowMaxSlots]; int n; };      // must match lili_launch.h

struct LmShared {
    double vals[kLmGroup][40];
    double tot[40];            // sum of all partials of the current evaluation (upper triangle of the 8x8 Gram, [36] cost, [37] rows)
    double cur[40];            // the same at the accepted point x
    double full[64];           // symmetric 8x8 Gram of `cur`
    double H[6][6], gv[6];     // local-coordinate normal matrix P^T G P and gradient P^T G_7r at x
    double Hs[6][6], gs[6];    // the same in Jacobi-scaled coordinates (S H S, S g): what the trust-region step works on
    double x[7], xn[7];        // accepted point, candidate
    double scale[6];
    double cost, radius, decrease, model_change, step_norm;
    int it, n_ok, term, go;    // go: 1 = evaluate the candidate next, 0 = finished
    int counts[2];
    int n_invalid;             // consecutive invalid steps (model cost change <= 0)
    int stalled;               // a bounded wait gave up
    int take;                  // the candidate was accepted: `tot` becomes `cur`
    int max_iter;
    // the solver options, parked here so that they are not live in registers across the whole launch
    double function_tolerance, gradient_tolerance, parameter_tolerance;
    double max_radius, min_radius, min_relative_decrease, min_lm_diagonal, max_lm_diagonal;
};

// H = P^T G77 P, gv = P^T G7r at the quaternion xq (lanes 0..41 of ONE wave; gn_update_block of lili_s2m.hip builds the same)
__device__ __forceinline__ void lm_local_system(LmShared& sh) {
    const int tid = threadIdx.x & 63;
    const double x0 = sh.x[3], x1 = sh.x[4], x2 = sh.x[5], x3 = sh.x[6];
    const double* gram = sh.full;
    auto jcol = [&](int c, double o[4]) {
        o[0] = c == 0 ? -x1 : c == 1 ? -x2 : -x3;
        o[1] = c == 0 ? x0 : c == 1 ? x3 : -x2;
        o[2] = c == 0 ? -x3 : c == 1 ? x0 : x1;
        o[3] = c == 0 ? x2 : c == 1 ? -x1 : x0;
    };
    if (tid < 42) {
        const int a = tid < 36 ? tid / 6 : tid - 36, b = tid < 36 ? tid % 6 : 7;
        double jb[4] = {0, 0, 0, 0}, ja[4] = {0, 0, 0, 0};
        if (b >= 3 && b < 6) jcol(b - 3, jb);
        if (a >= 3) jcol(a - 3, ja);
        auto Mrow = [&](int i) -> double {
            if (b < 3 || b == 7) return gram[i * 8 + b];
            return ((gram[i * 8 + 3] * jb[0] + gram[i * 8 + 4] * jb[1]) + gram[i * 8 + 5] * jb[2]) + gram[i * 8 + 6] * jb[3];
        };
        double v;
        if (a < 3) v = Mrow(a);
        else v = ((ja[0] * Mrow(3) + ja[1] * Mrow(4)) + ja[2] * Mrow(5)) + ja[3] * Mrow(6);
        if (tid < 36) sh.H[a][b] = v; else sh.gv[a] = v;
    }
}
__device__ __forceinline__ void lm_tri_to_full(const double* tri, double* full) {      // lanes 0..63 of one wave
    const int lane = threadIdx.x & 63;
    const int r = lane >> 3, c = lane & 7;
    const int a = r < c ? r : c, b = r < c ? c : r;
    full[lane] = tri[a * 8 - a * (a - 1) / 2 + (b - a)];
}

// Jacobi-scaled system of the accepted point (lanes 0..41 of one wave, after lm_local_system and with sh.scale set)
__device__ __forceinline__ void lm_scaled_system(LmShared& sh) {
    const int tid = threadIdx.x & 63;
    if (tid < 36) sh.Hs[tid / 6][tid % 6] = sh.H[tid / 6][tid % 6] * sh.scale[tid / 6] * sh.scale[tid % 6];
    else if (tid < 42) sh.gs[tid - 36] = sh.gv[tid - 36] * sh.scale[tid - 36];
}
// The trust-region step from the accepted point, by ONE WAVE (all 64 lanes call it, control flow uniform).  Returns with sh.go = 1 and sh.xn =
// candidate, or sh.go = 0 (finished).  The 6x6 system (H_s + D^2) d = -g_s is eliminated on 42 lanes — lane (i, j) holds entry j of row i of the
// augmented matrix [A | b] — with the pivot taken by v_readlane and the pivot row / column through ds_bpermute: six elimination and six substitution
// steps of ~200 cycles instead of ~600 dependent f64 instructions on one lane (round 3 first version: ~3 us of the ~8.5 us an evaluation took).
// Every workgroup runs the same instruction sequence on the same bits, so the candidates agree bit for bit across workgroups.
__device__ __forceinline__ double lm_bcast(double v, int src_lane) { return __shfl(v, src_lane); }
__device__ __forceinline__ void lm_propose(LmShared& sh) {
    const int lane = threadIdx.x & 63;
    const int ri = lane / 7, cj = lane - 7 * ri;             // row / column of the augmented 6 x 7 matrix (lanes >= 42 idle along)
    const bool in = lane < 42;
    for (;;) {
        const int it = sh.it;
        if (it >= sh.max_iter) { if (lane == 0) { sh.term = LILI_LM_MAX_ITERATIONS; sh.go = 0; } return; }
        double gmax = 0.0;
#pragma unroll
        for (int i = 0; i < 6; i++) gmax = fmax(gmax, fabs(sh.gv[i]));
        if (gmax <= sh.gradient_tolerance) { if (lane == 0) { sh.term = LILI_LM_GRADIENT_TOLERANCE; sh.it = it + 1; sh.go = 0; } return; }      // (Ceres counts the iteration it stops in)
        const double radius = sh.radius;
        if (!(radius > sh.min_radius)) { if (lane == 0) { sh.term = LILI_LM_MIN_RADIUS; sh.go = 0; } return; }      // MinTrustRegionRadiusReached (after an invalid step)
        double a = 0.0;
        if (in) {
            if (cj < 6) {
                a = sh.Hs[ri][cj];
                if (cj == ri) a += fmin(fmax(a, sh.min_lm_diagonal), sh.max_lm_diagonal) / radius;      // D^2 = clamp(diag H_s) / radius
            } else a = -sh.gs[ri];
        }
        bool okc = true;
        double pinv[6];
#pragma unroll
        for (int p = 0; p < 6; p++) {
            const double piv = lm_bcast(a, p * 7 + p);
            okc = okc && (piv > 0.0);
            pinv[p] = 1.0 / piv;
            const double rowp = lm_bcast(a, p * 7 + (in ? cj : 0));      // A[p][my column]
            const double colp = lm_bcast(a, (in ? ri : 0) * 7 + p);      // A[my row][p]
            if (in && ri > p) a -= (colp * pinv[p]) * rowp;
        }
        double d[6];
#pragma unroll
        for (int p = 5; p >= 0; p--) {
            d[p] = lm_bcast(a, p * 7 + 6) * pinv[p];
            const double up = lm_bcast(a, (in ? ri : 0) * 7 + p);        // U[my row][p]
            if (in && cj == 6 && ri < p) a -= up * d[p];
        }
#pragma unroll
        for (int i = 0; i < 6; i++) okc = okc && (d[i] == d[i]);
        if (!okc) { if (lane == 0) { sh.term = LILI_LM_NUMERICAL_FAILURE; sh.go = 0; } return; }
        // model_cost_change = -d^T (g_s + H_s d / 2): row sums on six lanes, then a fixed-order sum
        double tr = 0.0;
        if (lane < 6) {
            double hd = 0.0;
#pragma unroll
            for (int j = 0; j < 6; j++) hd += sh.Hs[lane][j] * d[j];
            double dl = 0.0;
#pragma unroll
            for (int j = 0; j < 6; j++) dl = lane == j ? d[j] : dl;
            tr = dl * (sh.gs[lane] + 0.5 * hd);
        }
        double mc = 0.0;
#pragma unroll
        for (int i = 0; i < 6; i++) mc += lm_bcast(tr, i);
        mc = -mc;
        if (!(mc > 0.0)) {
            // not a descent step of the model = Ceres' INVALID step (TrustRegionMinimizer::HandleInvalidStep): the iteration counts, nothing is evaluated,
            // LevenbergMarquardtStrategy::StepIsInvalid halves the radius (the rejection divisor is left alone); max_num_consecutive_invalid_steps (5) of
            // them in a row end the solve with FAILURE
            const int n_inv = sh.n_invalid + 1;
            LILI_WAVE_SYNC();
            if (lane == 0) { sh.n_invalid = n_inv; sh.radius = radius * 0.5; sh.it = it + 1; }
            if (n_inv >= 5) { if (lane == 0) { sh.term = LILI_LM_NUMERICAL_FAILURE; sh.go = 0; } return; }
            LILI_WAVE_SYNC();
            continue;          // (the loop head checks max_iterations, then the radius, in FinalizeIterationAndCheckIfMinimizerCanContinue's order)
        }
        if (lane == 0) {
            double n2 = 0.0;
#pragma unroll
            for (int i = 0; i < 6; i++) { d[i] = d[i] * sh.scale[i]; n2 += d[i] * d[i]; }      // delta in the unscaled local coordinates
            sh.model_change = mc;
            sh.n_invalid = 0;
            sh.step_norm = sqrt(n2);
            // x (+) delta: ceres::QuaternionParameterization::Plus
            sh.xn[0] = sh.x[0] + d[0]; sh.xn[1] = sh.x[1] + d[1]; sh.xn[2] = sh.x[2] + d[2];
            const double nd2 = d[3] * d[3] + d[4] * d[4] + d[5] * d[5];
            if (nd2 > 0.0) {
                // sin(|d|) / |d| and cos(|d|): the series of sinc_cos_small below 0.5 rad, above it halve the angle first and double it back
                // (libm's sin / cos bring a Payne-Hanek reduction with a scratch table into the launch; a trust-region step never turns that far anyway)
                double sbd, cw;
                if (nd2 < 0.25) sinc_cos_small(nd2, sbd, cw);
                else {
                    double h2 = nd2; int k = 0;
                    while (h2 >= 0.25 && k < 60) { h2 *= 0.25; k++; }
                    double sc, c;
                    sinc_cos_small(h2, sc, c);
                    double sn = sc * sqrt(h2);
                    for (int i = 0; i < k; i++) { const double s2 = 2.0 * sn * c, c2 = c * c - sn * sn; sn = s2; c = c2; }
                    sbd = sn / sqrt(nd2); cw = c;
                }
                const dq r = qmul(dq{cw, sbd * d[3], sbd * d[4], sbd * d[5]}, dq{sh.x[3], sh.x[4], sh.x[5], sh.x[6]});
                sh.xn[3] = r.w; sh.xn[4] = r.x; sh.xn[5] = r.y; sh.xn[6] = r.z;
            } else { sh.xn[3] = sh.x[3]; sh.xn[4] = sh.x[4]; sh.xn[5] = sh.x[5]; sh.xn[6] = sh.x[6]; }
            sh.go = 1;
        }
        return;
    }
}

// one slot's solve, by the workgroups [0, a.nb) of that slot (b = the workgroup's index within the slot)
__device__ __forceinline__ void solve_lm_body(const LmArgs& a, const MatchParams& P, const int b, double* lds, LmShared& sh) {
    const bool surf = b < a.S.nb;
    const bool wave0 = threadIdx.x < 64;
    const bool boss = b == 0 && threadIdx.x == 0;
    // correspondence counts (ROT residual scale num / N): fixed for the whole solve, summed once from the association's block counts
    {
        const int n_s = (a.S.n_q > 0 && a.S.block_counts) ? sum_block_counts(a.S.block_counts, a.S.n_bc) : 0;
        __syncthreads();
        const int n_e = (a.E.n_q > 0 && a.E.block_counts) ? sum_block_counts(a.E.block_counts, a.E.n_bc) : 0;
        if (threadIdx.x == 0) {
            sh.counts[0] = n_s; sh.counts[1] = n_e;
            for (int i = 0; i < 7; i++) { sh.x[i] = a.state->pose[i]; sh.xn[i] = sh.x[i]; }
            sh.max_iter = a.max_iter;
            sh.function_tolerance = a.function_tolerance; sh.gradient_tolerance = a.gradient_tolerance; sh.parameter_tolerance = a.parameter_tolerance;
            sh.max_radius = a.max_radius; sh.min_radius = a.min_radius; sh.min_relative_decrease = a.min_relative_decrease;
            sh.min_lm_diagonal = a.min_lm_diagonal; sh.max_lm_diagonal = a.max_lm_diagonal;
            sh.radius = a.initial_radius; sh.decrease = 2.0; sh.it = 0; sh.n_ok = 0; sh.term = LILI_LM_MAX_ITERATIONS; sh.go = 1; sh.stalled = 0; sh.take = 0; sh.n_invalid = 0;
            sh.cost = 0.0; sh.model_change = 0.0; sh.step_norm = 0.0;
        }
        __syncthreads();
    }
    LinArgs S = a.S, E = a.E;
    S.block_counts = nullptr; E.block_counts = nullptr;         // the bodies then take N from n_global (= sh.counts)
    int n_log = 0;
    double cost0 = 0.0;
    for (int eval = 0;; eval++) {
        const int par = eval & 1;
        const unsigned long long key = xchg_key(a.launch, eval);
        double* part = a.part + (size_t)par * a.nb * kPartialStride;
        double* gsum = a.gsum + (size_t)par * a.ng * kPartialStride;
        PoseArg pa{};
        for (int i = 0; i < 3; i++) pa.t[i] = sh.xn[i];
        for (int i = 0; i < 4; i++) pa.q[i] = sh.xn[3 + i];
        pa.state = nullptr; pa.derive_assoc = 0;
        // ---- evaluation at the candidate: this workgroup's partial, published as granules
        S.partials = part; E.partials = part + (size_t)a.S.nb * kPartialStride;
        if (surf) lin_surf_body(S, b, pa, P, a.state, sh.counts, lds, key);
        else lin_edge_body(E, b - a.S.nb, pa, P, a.state, sh.counts, lds, key);
        // ---- exchange (wave 0): group sums, then the total, both in index order
        if (wave0) {
            bool ok = true;
            if (a.ng > 1) {
                if (b % kLmGroup == 0) {
                    const int cnt = min(kLmGroup, a.nb - b);
                    ok = xchg_gather<40>(part + (size_t)b * kPartialStride, cnt, key, sh.vals, sh.tot);
                    if ((threadIdx.x & 63) < 40) store_granule(gsum + (size_t)(b / kLmGroup) * kPartialStride + 2 * (threadIdx.x & 63), sh.tot[threadIdx.x & 63], key);
                }
                ok = xchg_gather<40>(gsum, a.ng, key, sh.vals, sh.tot) && ok;
            } else ok = xchg_gather<40>(part, a.nb, key, sh.vals, sh.tot);
            if (!ok && threadIdx.x == 0) sh.stalled = 1;
            // ---- step logic, identical in every workgroup
            if (eval == 0) {
                // first evaluation: x0 is the accepted point; Jacobi scaling 1 / (1 + sqrt(diag J^T J)) from this Jacobian, kept for the whole solve
                if (threadIdx.x < 40) sh.cur[threadIdx.x] = sh.tot[threadIdx.x];
                LILI_WAVE_SYNC();
                lm_tri_to_full(sh.cur, sh.full);
                LILI_WAVE_SYNC();
                lm_local_system(sh);
                LILI_WAVE_SYNC();
                if (threadIdx.x < 6) sh.scale[threadIdx.x] = 1.0 / (1.0 + sqrt(sh.H[threadIdx.x][threadIdx.x]));
                if (threadIdx.x == 0) { sh.cost = sh.cur[36]; cost0 = sh.cost; }
                LILI_WAVE_SYNC();
                lm_scaled_system(sh);
                LILI_WAVE_SYNC();
                if (threadIdx.x == 0 && sh.stalled) { sh.term = LILI_LM_STALLED; sh.go = 0; }
            } else {
                // the candidate's cost is known: accept or reject (Ceres checks both tolerances on the candidate first)
                int accepted = 0, stop = 0;
                if (threadIdx.x == 0) {
                    const double new_cost = sh.tot[36];
                    const double rho = (sh.cost - new_cost) / sh.model_change;
                    if (boss && a.summary && n_log < LILI_LM_MAX_LOG) {
                        lili_lm_iteration& L = a.summary->it[n_log];
                        L.cost = sh.cost; L.new_cost = new_cost; L.rho = rho; L.radius = sh.radius; L.step_norm = sh.step_norm; L.accepted = 0; L.iteration = sh.it;
                    }
                    const double xnorm = sqrt(sh.x[0] * sh.x[0] + sh.x[1] * sh.x[1] + sh.x[2] * sh.x[2] + sh.x[3] * sh.x[3] + sh.x[4] * sh.x[4] + sh.x[5] * sh.x[5] + sh.x[6] * sh.x[6]);
                    if (sh.stalled) { sh.term = LILI_LM_STALLED; stop = 1; }
                    else if (sh.step_norm <= sh.parameter_tolerance * (xnorm + sh.parameter_tolerance)) { sh.term = LILI_LM_PARAMETER_TOLERANCE; stop = 1; }
                    // Ceres returns from ParameterToleranceReached / FunctionToleranceReached BEFORE IsStepSuccessful / HandleSuccessfulStep
                    // (TrustRegionMinimizer::Minimize): the candidate that triggers a tolerance is never taken, x stays at the last accepted point
                    else if (fabs(sh.cost - new_cost) <= sh.function_tolerance * sh.cost) { sh.term = LILI_LM_FUNCTION_TOLERANCE; stop = 1; }
                    else if (rho > sh.min_relative_decrease) {
                        accepted = 1;
                        const double f = 2.0 * rho - 1.0;
                        sh.radius = fmin(sh.max_radius, sh.radius / fmax(1.0 / 3.0, 1.0 - f * f * f));
                        sh.decrease = 2.0;
                    } else {
                        // LevenbergMarquardtStrategy::StepRejected: no clamp; MinTrustRegionRadiusReached ends the solve (CONVERGENCE) once the radius is
                        // at or below min_trust_region_radius
                        sh.radius = sh.radius / sh.decrease; sh.decrease *= 2.0;
                        if (!(sh.radius > sh.min_radius)) { sh.term = sh.it + 1 >= sh.max_iter ? LILI_LM_MAX_ITERATIONS : LILI_LM_MIN_RADIUS; stop = 1; }      // (max iterations is checked first)
                    }
                    if (accepted) { for (int i = 0; i < 7; i++) sh.x[i] = sh.xn[i]; sh.cost = new_cost; sh.n_ok++; }
                    if (boss && a.summary && n_log < LILI_LM_MAX_LOG) a.summary->it[n_log].accepted = accepted;
                    n_log++;
                    sh.it++;
                    sh.go = stop ? 0 : 1;
                    sh.take = accepted;
                }
                LILI_WAVE_SYNC();
                const bool take = sh.take != 0;
                if (take && threadIdx.x < 40) sh.cur[threadIdx.x] = sh.tot[threadIdx.x];
                LILI_WAVE_SYNC();
                if (take) {
                    lm_tri_to_full(sh.cur, sh.full);
                    LILI_WAVE_SYNC();
                    lm_local_system(sh);
                    LILI_WAVE_SYNC();
                    lm_scaled_system(sh);
                }
            }
            LILI_WAVE_SYNC();
            if (sh.go) lm_propose(sh);      // the next candidate (or the end), from the accepted point — the whole wave, ONE call site
            LILI_WAVE_SYNC();
        }
        __syncthreads();
        if (!sh.go) break;
    }
    if (boss) {
        for (int i = 0; i < 7; i++) a.state->pose[i] = sh.x[i];
        a.state->gn_status = (sh.term == LILI_LM_STALLED || sh.term == LILI_LM_NUMERICAL_FAILURE) ? 1 : 0;
        a.state->iters += sh.n_ok;
        if (a.summary) {
            a.summary->iterations = sh.it; a.summary->successful_steps = sh.n_ok; a.summary->termination = sh.term;
            a.summary->initial_cost = cost0; a.summary->final_cost = sh.cost; a.summary->final_radius = sh.radius;
            a.summary->n_logged = n_log < LILI_LM_MAX_LOG ? n_log : LILI_LM_MAX_LOG;
            a.summary->n_surf = sh.counts[0]; a.summary->n_edge = sh.counts[1];
        }
    }
}

// persistent launch: grid = S.nb + E.nb workgroups of kLmThreads threads; dynamic LDS = kLmThreads * kRow doubles (Gram staging rows)
__global__ __launch_bounds__(kLmThreads) void k_solve_lm(LmArgs a, MatchParams P) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    __shared__ LmShared sh;
    solve_lm_body(a, P, (int)blockIdx.x, lds, sh);
}
// The solves of ALL keyframes of a sliding window in ONE launch (round 6): workgroup bid belongs to the last slot whose first_block <= bid and is that slot's workgroup
// bid - first_block of k_solve_lm — the slots' solves are independent (own records, own partial buffers, own keys) and run side by side, one slot's exchange under another's
// arithmetic.  It was one launch per slot on forked streams: 65 us of fork / join / launch overhead per window solve around 9 us per evaluation of all three keyframes.
__global__ __launch_bounds__(kLmThreads) void k_solve_lm_window(WinLmArgs W, MatchParams P) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    __shared__ LmShared sh;
    const int bid = (int)blockIdx.x;
    int i = 0;
#pragma unroll
    for (int k = 1; k < kWindowMaxSlots; k++) if (k < W.n && bid >= W.first_block[k]) i = k;
    solve_lm_body(W.a[i], P, bid - W.first_block[i], lds, sh);
}

}  // namespace lili
