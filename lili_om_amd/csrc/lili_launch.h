// Internal: forward declarations of every kernel the host files launch (definitions: lili_s2m.hip, lili_s2m_coop.hip, lili_s2m_lm.hip) and the launch helper.
#pragma once
#include "lili_ctx.h"

#include <hip/hip_ext.h>

namespace lili {
// kernels (lili_s2m.hip)
__global__ void k_cloud_to_f4(const unsigned char*, int, int, int, float4*, unsigned*);
__global__ void k_bbox(const float4*, int, unsigned*);
__global__ void k_bbox_src(SrcCloud, int, unsigned*);
__global__ void k_cell_count(SrcCloud, int, GridView, int*, int*, unsigned long long*, int, float);
__global__ void k_scan_block_sums(const int*, int64_t, int*);
__global__ void k_scan_sums(int*, int);
__global__ void k_scan_apply(const int*, int64_t, const int*, int*);
template <bool NARROW> __global__ void k_scan_lookback_t(int*, const unsigned char*, int64_t, unsigned long long*, unsigned*);
__global__ void k_cell_count_narrow(SrcCloud, int, GridView, unsigned*, unsigned char*, unsigned long long*, int, float);
template <typename RankT> __global__ void k_scatter_t(SrcCloud, int, GridView, const RankT*, const int*, float4*, float*);
__global__ void k_start9(const int*, GridView, const int*, int*);
__global__ void k_rowtot9(const int*, GridView, int*);
template <int BS> __global__ void k_associate_lin(AssocArgs, AssocArgs, PoseArg, MatchParams, double*, double*);
__global__ void k_scatter9(const int*, GridView, const int*, float4*, float*, int);
template <int BS> __global__ void k_associate_surf(const float4*, int, GridView, PoseArg, MatchParams, float4*, double*, unsigned char*, int*, float*, int*);
template <int BS> __global__ void k_associate_edge(const float4*, int, GridView, PoseArg, MatchParams, float4*, float4*, unsigned char*, int*, float*, int*);
__global__ void k_associate_both(AssocArgs, AssocArgs, PoseArg, MatchParams);
__global__ void k_linearize(LinArgs, LinArgs, PoseArg, MatchParams, const SlotState*, const int*, FuseTail);
__global__ void k_reduce_partials(const double*, int, const double*, int, double*, SlotState*, int, P2PView, unsigned long long, double*, SlotState*);
__global__ void k_sum_counts(const int*, int, const int*, int, SlotState*, int*, P2PView);
__global__ void k_gn_update(const double*, SlotState*);
__global__ void k_pose_copy(SlotState*, const SlotState*);
__global__ void k_window_reduce(WindowArgs, double*, int, P2PView);
__global__ void k_linearize_window(WinLinArgs, MatchParams);
__global__ void k_window_gn(WindowArgs, const double*);
__global__ void k_window_counts(WindowArgs, int*, P2PView);
// lili_s2m_dense.hip: the association on a dense map (fine index)
__global__ void k_associate_fine(AssocArgs, int, int, int, PoseArg, MatchParams);
// lili_s2m_lm.hip: the Levenberg-Marquardt loop on fixed correspondences, one persistent launch
struct LmArgs {      // must match lili_s2m_lm.hip
    LinArgs S, E;
    SlotState* state;
    double* part;
    double* gsum;
    int nb, ng;
    int max_iter;
    unsigned long long launch;
    lili_lm_summary* summary;
    double function_tolerance, gradient_tolerance, parameter_tolerance;
    double initial_radius, max_radius, min_radius, min_relative_decrease, min_lm_diagonal, max_lm_diagonal;
};
struct WinLmArgs { LmArgs a[kWindowMaxSlots]; int first_block[kWindowMaxSlots]; int n; };      // must match lili_s2m_lm.hip
__global__ void k_solve_lm(LmArgs, MatchParams);
__global__ void k_solve_lm_window(WinLmArgs, MatchParams);
// lili_s2m_coop.hip: L lanes per query (small launches)
template <int L, bool LIN> __global__ void k_associate_coop(AssocArgs, AssocArgs, PoseArg, MatchParams, double*, double*, SlotState*, int);
template <int L> __global__ void k_associate_coop_window(WinAssocArgs, MatchParams);
struct IterArgs {      // must match lili_s2m_coop.hip
    SlotState* state;
    double* part;
    double* gsum;
    double* cpart;
    int nb, ng, n_iters, derive_assoc;
    unsigned long long launch;
};
template <int L> __global__ void k_iterate_coop(AssocArgs, AssocArgs, MatchParams, IterArgs);
}  // namespace lili

// A kernel launch that may drop the barrier against the kernels enqueued before it on the stream (`any_order`: hipExtAnyOrderLaunch — the AQL packet goes without the
// barrier bit, so it is dispatched as soon as the packets in front of it have been DISPATCHED, not completed).  Only the association that follows a reduction + GN
// kernel uses it (option "overlap_gn"): its waves poll for the pose that kernel publishes.  Where the runtime ignores the flag the launch is an ordinary one.
template <typename K, typename... A>
static inline void launch_k(hipStream_t stream, bool any_order, K kernel, dim3 grid, dim3 block, size_t lds, A... args) {
    if (any_order) hipExtLaunchKernelGGL(kernel, grid, block, (uint32_t)lds, stream, nullptr, nullptr, hipExtAnyOrderLaunch, args...);
    else hipLaunchKernelGGL(kernel, grid, block, lds, stream, args...);
}
