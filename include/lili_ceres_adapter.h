// lili_ceres_adapter.h — header-only glue between ceres::Solve and liblili_hip.so.
//
// Meant for the reference's catkin workspace (needs <ceres/ceres.h>).  Ceres is not available in the build image of this
// repository; the header is nevertheless compiled and exercised here against a stand-in ceres::CostFunction /
// ceres::Problem together with the reference's own LidarKeyframeFactor.h (oracle/refshim/ref_seam.cpp, run by
// tests/test_reference_gpu.py::test_ceres_seam_batch_factor_equals_reference_blocks: one LidarBatchFactor gives the normal
// equations and cost of the reference's ~1 500 per-correspondence blocks to 1e-15).  Its algebra, lili_gram_to_factor(),
// lives in the library and is unit-tested in tests/test_abi_cpu.py::test_gram_to_factor_reproduces_normal_equations.
//
// What it replaces in the reference (L/ = LiLi-OM/):
//   for every correspondence:  new AutoDiffCostFunction<LidarEdgeFactor|LidarPlaneNormFactor,1,3,4>(...)
//                              problem.AddResidualBlock(cost, CauchyLoss(1.0), t_ptr, q_ptr)
//                                                          — L/src/BackendFusion.cpp:938-974
// by ONE cost function per keyframe:
//   problem.AddResidualBlock(new lili::LidarBatchFactor(ctx, slot, mask, params), nullptr, t_ptr, q_ptr);
//
// Contract mirrored from ceres::CostFunction: Evaluate(parameters, residuals, jacobians) with
// parameters[0] = t[3], parameters[1] = q[4] (w,x,y,z); jacobians / jacobians[i] may be NULL; row-major
// num_residuals x block_size; returning false makes Ceres reject the step.  The block has 9 residuals:
// 8 rows of the square-root factor of the robustified Gram and one zero-Jacobian row that pads the cost to
// sum 1/2 rho(r_i^2) (so Ceres' step-acceptance ratio sees the same cost as with per-point blocks); the loss is
// already applied per residual on the GPU (same corrector as L/src/MarginalizationFactor.cpp:44-70), hence
// loss_function = nullptr.  Ownership: ceres::Problem owns the factor (default options); the factor does not
// own the lili context.  Threading: Ceres' num_threads stays 1 (the reference never sets it); one context per
// thread otherwise.
#pragma once
#include <ceres/ceres.h>

#include <vector>

#include "lili_hip.h"

namespace lili {

class LidarBatchFactor : public ceres::SizedCostFunction<9, 3, 4> {
public:
    LidarBatchFactor(lili_ctx* ctx, int slot, int kind_mask, const lili_s2m_params& params)
        : ctx_(ctx), slot_(slot), mask_(kind_mask), params_(params) {}

    bool Evaluate(double const* const* parameters, double* residuals, double** jacobians) const override {
        double gram[64], cost = 0.0;
        int counts[2] = {0, 0};
        if (lili_s2m_linearize(ctx_, slot_, mask_, parameters[0], parameters[1], &params_, gram, &cost, counts) != LILI_OK)
            return false;
        double res[9], jac[63];
        if (lili_gram_to_factor(gram, cost, res, jac) != LILI_OK) return false;
        for (int i = 0; i < 9; ++i) residuals[i] = res[i];
        if (jacobians) {
            if (jacobians[0]) for (int r = 0; r < 9; ++r) for (int c = 0; c < 3; ++c) jacobians[0][r * 3 + c] = jac[r * 7 + c];
            if (jacobians[1]) for (int r = 0; r < 9; ++r) for (int c = 0; c < 4; ++c) jacobians[1][r * 4 + c] = jac[r * 7 + 3 + c];
        }
        return true;
    }

    // Lidar contribution to MarginalizationInfo's A, b (L/src/MarginalizationFactor.cpp:3-29 uses rightCols(3) of
    // the 1x4 quaternion Jacobian): rows/cols {0,1,2} and {4,5,6} of the Gram, b from column 7.
    static void MarginalizationBlocks(const double gram[64], double A[36], double b[6]) {
        const int idx[6] = {0, 1, 2, 4, 5, 6};
        for (int i = 0; i < 6; ++i) {
            for (int j = 0; j < 6; ++j) A[i * 6 + j] = gram[idx[i] * 8 + idx[j]];
            b[i] = gram[idx[i] * 8 + 7];
        }
    }

private:
    lili_ctx* ctx_;
    int slot_, mask_;
    lili_s2m_params params_;
};

// The lidar terms of ALL keyframes of the sliding window as ONE cost function (L/src/BackendFusion.cpp:919-980 adds the per-correspondence
// blocks keyframe by keyframe): parameter blocks (t_0, q_0, t_1, q_1, ...), 9 residuals per keyframe as in LidarBatchFactor, block-diagonal
// Jacobian.  One Evaluate is ONE lili_s2m_linearize_window call — the keyframes run concurrently on the GPU and the host synchronises once
// per evaluation instead of once per keyframe.  Numerically identical to one LidarBatchFactor per keyframe (oracle/refshim/ref_seam.cpp).
class LidarWindowFactor : public ceres::CostFunction {
public:
    LidarWindowFactor(lili_ctx* ctx, const std::vector<int>& slots, int kind_mask, const lili_s2m_params& params)
        : ctx_(ctx), slots_(slots), mask_(kind_mask), params_(params) {
        set_num_residuals(9 * (int)slots_.size());
        for (size_t k = 0; k < slots_.size(); ++k) { mutable_parameter_block_sizes()->push_back(3); mutable_parameter_block_sizes()->push_back(4); }
    }

    bool Evaluate(double const* const* parameters, double* residuals, double** jacobians) const override {
        const int K = (int)slots_.size(), rows = 9 * K;
        std::vector<double> t(3 * K), q(4 * K), gram(64 * K), cost(K);
        for (int k = 0; k < K; ++k) {
            for (int c = 0; c < 3; ++c) t[3 * k + c] = parameters[2 * k][c];
            for (int c = 0; c < 4; ++c) q[4 * k + c] = parameters[2 * k + 1][c];
        }
        if (linearize(t, q, gram, cost) != LILI_OK) return false;
        for (int k = 0; k < K; ++k) {
            double res[9], jac[63];
            if (lili_gram_to_factor(&gram[64 * k], cost[k], res, jac) != LILI_OK) return false;
            for (int i = 0; i < 9; ++i) residuals[9 * k + i] = res[i];
            if (!jacobians) continue;
            if (double* jt = jacobians[2 * k]) {
                for (int i = 0; i < rows * 3; ++i) jt[i] = 0.0;
                for (int r = 0; r < 9; ++r) for (int c = 0; c < 3; ++c) jt[(9 * k + r) * 3 + c] = jac[r * 7 + c];
            }
            if (double* jq = jacobians[2 * k + 1]) {
                for (int i = 0; i < rows * 4; ++i) jq[i] = 0.0;
                for (int r = 0; r < 9; ++r) for (int c = 0; c < 4; ++c) jq[(9 * k + r) * 4 + c] = jac[r * 7 + 3 + c];
            }
        }
        return true;
    }

    // The window SHARDED over the ranks of a node (BASELINE configs[4]; every rank holds its shard of every keyframe's features and runs the same
    // solver): Evaluate becomes ONE lili_s2m_linearize_window_sharded — the records of all keyframes summed over the ranks in one exchange of
    // K x 72 doubles, identical bits on every rank, so the ranks' solvers take identical steps.  allreduce / comm: lili_p2p_allreduce + the
    // lili_p2p*, or ncclAllReduce + the ncclComm_t; d_gram: device buffer of K x LILI_GRAM_DOUBLES doubles owned by the caller.  Call
    // lili_s2m_counts_window_sharded after the associations (ROT flavour) before the solve.
    void shard_over_ranks(lili_allreduce_fn allreduce, void* comm, double* d_gram) { allreduce_ = allreduce; comm_ = comm; d_gram_ = d_gram; owner_.clear(); }
    // The window with ONE WHOLE keyframe per rank (the split that scales, lili_hip.h "slot-per-rank window"): slot k lives on rank owner[k] with all its features, the
    // other ranks hold only its pose.  Evaluate becomes ONE lili_s2m_linearize_window_gather_at — every owner linearises its keyframe at the solver's parameter values,
    // one exchange of K x 72 doubles in which every record has a single contributor, identical bits on every rank.  No count exchange (the owner's count is the global one).
    void gather_over_ranks(lili_allreduce_fn allreduce, void* comm, double* d_gram, const std::vector<int>& owner, int rank) {
        allreduce_ = allreduce; comm_ = comm; d_gram_ = d_gram; owner_ = owner; rank_ = rank;
    }

private:
    int linearize(const std::vector<double>& t, const std::vector<double>& q, std::vector<double>& gram, std::vector<double>& cost) const {
        const int K = (int)slots_.size();
        if (d_gram_ && (int)owner_.size() == K)
            return lili_s2m_linearize_window_gather_at(ctx_, slots_.data(), K, mask_, t.data(), q.data(), &params_, owner_.data(), rank_, allreduce_, comm_, d_gram_, gram.data(), cost.data(), nullptr);
        if (d_gram_) return lili_s2m_linearize_window_sharded(ctx_, slots_.data(), K, mask_, t.data(), q.data(), &params_, allreduce_, comm_, d_gram_, gram.data(), cost.data(), nullptr);
        return lili_s2m_linearize_window(ctx_, slots_.data(), K, mask_, t.data(), q.data(), &params_, gram.data(), cost.data(), nullptr);
    }
    lili_ctx* ctx_;
    std::vector<int> slots_;
    int mask_;
    lili_s2m_params params_;
    lili_allreduce_fn allreduce_ = nullptr;
    void* comm_ = nullptr;
    double* d_gram_ = nullptr;
    std::vector<int> owner_;
    int rank_ = 0;
};

}  // namespace lili
