// Kernel-side declarations shared by the translation units of liblili_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "lili_p2p_dev.h"

namespace lili {

// Uniform-grid index over the local map (replaces the FLANN kd-tree).
// Points are stored cell-sorted as float4 (x, y, z, bitcast(original index)); cell_start has
// n_cells+1 entries; cells are x-fastest so the 3 x-neighbours of a cell are one contiguous run.
// With super-rows, `pts` / `aux` continue behind the n_points base entries with the super-row copy (same float4 layout).
struct GridView {
    const float4* pts;      // cell-sorted map points
    const float* aux;       // cell-sorted auxiliary float (Livox reflectivity) or nullptr
    const int* cell_start;  // [n_cells + 1]
    const int* cell_start9; // super-row index (k_start9 in lili_s2m.hip) or nullptr: [bnx*bny*bnz + 1] positions in the unified array
    int bx0, by0, bz0, bnx, bny, bnz;   // the box of cells that has super-rows (the whole grid, or the cells around lili_map_focus)
    double ox, oy, oz;      // grid origin
    double inv_cell;        // 1 / cell edge
    double cell;            // 1 / inv_cell (the value the pruning bounds use)
    int nx, ny, nz;
    int n_points;
    int reach;              // neighbourhood half-width in cells: reach * cell edge >= 1.01 * gate radius (1 or 2)
};

// A caller's point cloud as the map-index build reads it (round 4: no float4 copy of the map any more): device pointer, byte stride, byte offset of the
// auxiliary float or < 0.  `f4`: stride 16, 16-byte aligned, aux at offset 12 or absent — one 16-byte load per point.
struct SrcCloud {
    const unsigned char* p;
    int stride, aux_off;
    int f4;
};

// Per-slot state that lives in device memory so that outer iterations need no host round trip.
struct SlotState {
    double pose[7];   // body pose: t(3), q(w,x,y,z)
    int n_res[2];     // correspondences of the last associate: [surf, edge]
    int gn_status;    // 0 ok, 1 = normal matrix not positive definite (pose left unchanged)
    int iters;        // GN updates applied since pose_set
    double last_delta[6];
    long long tprof[16];   // profiling aid (LILI_DEBUG bit 256): s_memrealtime stamps (100 MHz) of one block per kernel
    unsigned int reserved_;
    unsigned long long cnt_word;   // k_associate_coop's in-launch count barrier: [0,24) surf correspondences, [24,48) edge, [48,64) workgroups arrived; k_reduce_partials zeroes it
    unsigned long long epoch;   // fused linearisation launches of this slot so far: launch_key(epoch) tags the granules of the next one
    // Round 5, option "overlap_gn": raised (sticky) by an association launch that gave up waiting for the pose the Gauss-Newton kernel in front of it publishes
    // (PosePub below).  Behind `epoch`: lili_s2m_pose_set does not touch it.
    unsigned long long wait_failed;
};
// The pose of the last Gauss-Newton update PUBLISHED for an association launch that is already running (option "overlap_gn"): seven 16-byte granules
// {value, value ^ key}, key unique per update (the data is its own flag, as the partials of the fused tail), in kPubReplicas copies 128 bytes apart — the 3 125 waves
// of the association poll, and all of them on ONE line queue up on one memory channel behind each other (and the publishing store behind them: measured, the overlap
// then gains nothing); workgroup b polls copy b mod kPubReplicas.
constexpr int kPubReplicas = 64;
constexpr int kPubStride = 16;      // doubles per copy: 7 granules + padding to 128 bytes

// Fused tail of a linearisation launch (see fused_tail in lili_s2m.hip)
struct FuseTail {
    int mode;                  // 0 = off, 1 = reduce to `out`, 2 = reduce + GN update, 3 = publish the partials as granules only
    const double* part_surf; int nb_surf;
    const double* part_edge; int nb_edge;
    double* out;
    SlotState* state;
    int debug;                 // LILI_DEBUG bits (256: phase stamps of the reducer block)
};


// Pose argument of a kernel: either by value (host-provided) or read from SlotState.
struct PoseArg {
    double t[3];
    double q[4];
    const SlotState* state;  // if non-null, the body pose is read from state->pose
    int derive_assoc;        // associate only: derive (Q2,T2) = (Q*q_lb^-1, T - Q2*t_lb) from the body pose
    const double* pub;             // the slot's kPubReplicas x kPubStride published-pose copies (wait_key != 0)
    unsigned long long wait_key;   // associate only, != 0: the launch may have started BEFORE the reduction + GN kernel in front of it has finished (no barrier between the two
                                   // dispatches): the pose is taken from state->pose_pub once all seven granules carry this key
};

struct MatchParams {     // device copy of lili_s2m_params (+ derived values)
    int variant, loss;
    double loss_a, lidar_const, kd_max_radius, edge_gate, surf_dist_thres, reflect_thres, surf_weight_min, edge_dist_max;
    double q_lb[4], t_lb[3];
    double q_lb_inv[4];   // Eigen inverse() of q_lb (conjugate / squared norm), computed once on the host: IEEE divisions, same bits
    double q_lb_inv_jet[4];   // the same inverse as the plane factor sees it on ceres::Jet (LidarKeyframeFactor.h:86): conjugate * (1 / n2)
    double scale_surf_num, scale_edge_num;
    int no_cost;   // 1: the robust cost value (slot 64 of the Gram record) is not computed — the fused iterate loops, whose record never leaves the library
    int debug;   // LILI_DEBUG env, tests and profiling builds only: 256 / 512 = clock stamps of the linearisation / reduction / association prologue into SlotState::tprof,
                 // 2048 = never fall back to the pivoted QR, 16384 = always the pivoted QR, 32768 = exact (d2, index) selector only (the tier tests force both tiers and
                 // compare bit for bit), 4096 = per-workgroup timestamps (only in a -DLILI_PHASE_PROBE build: tools/assoc_blocks.py, tools/assoc_phases.sh)
};

// Arguments of one kind (surf or edge) of the combined linearisation launch k_linearize.
struct LinArgs {
    const float4* queries; int n_q;
    const float4* rec0;            // surf: (w*n, w*d); edge: (A, s)
    const void* rec1;              // surf: double score[]; edge: float4 (B, 0)
    const unsigned char* valid;
    const int* block_counts; int n_bc;   // per-block correspondence counts of the association launch (nullptr: use n_global / state)
    double* partials;
    int nb;                        // linearisation blocks of this kind
};
// Arguments of one kind of the combined association launch k_associate_both.
struct AssocArgs {
    const float4* queries; int n_q; GridView g;
    float4* rec0; void* rec1; unsigned char* valid;
    int* dbg_idx; float* dbg_d2; int* block_counts;
    int nb;
};

// The slots of a sliding window handed to ONE reduction / count launch (k_window_reduce, k_window_counts in lili_s2m.hip)
constexpr int kWindowMaxSlots = 8;       // = LILI_MAX_SLOTS
struct WindowSlot {
    const double* part_surf; int nb_surf;      // block partials of the slot's last linearisation (k_linearize)
    const double* part_edge; int nb_edge;
    const int* bc_surf; int nbc_surf;          // per-block correspondence counts of the slot's last association
    const int* bc_edge; int nbc_edge;
    SlotState* state;
};
struct WindowArgs { WindowSlot s[kWindowMaxSlots]; int n; };
// The linearisation of EVERY slot of the window in ONE launch (k_linearize_window): slot i owns the blocks [first_block, first_block + S.nb + E.nb)
// of the grid and runs exactly the bodies k_linearize runs for it (same block geometry, same partial buffers) — one launch boundary per
// evaluation of the joint window instead of one per keyframe (the reference evaluates all keyframes per solver evaluation, L/src/BackendFusion.cpp:919-992).
struct WinLinSlot {
    LinArgs S, E;
    PoseArg pa;
    const SlotState* state;
    const int* n_global;           // all-reduced correspondence counts of a multi-GPU caller, else nullptr
    int first_block;
};
struct WinLinArgs { WinLinSlot s[kWindowMaxSlots]; int n; };
// The cooperative association of EVERY slot of the window in ONE launch (k_associate_coop_window in lili_s2m_coop.hip): per slot and kind what
// AssocArgs holds minus the map view, which all slots share.
struct WinAssocKind {
    const float4* queries; int n_q;
    float4* rec0; void* rec1; unsigned char* valid;
    int* dbg_idx; float* dbg_d2; int* block_counts;
    int nb;
};
struct WinAssocSlot { WinAssocKind k[2]; PoseArg pa; int first_block; };
struct WinAssocArgs { GridView g[2]; WinAssocSlot s[kWindowMaxSlots]; int n; };

constexpr int kPartialDoubles = 40;  // per-block partial: 36 upper-triangle Gram entries, cost, count, 2 spare
constexpr int kPartialStride = 80;   // doubles per block slot of the partial buffers: 40 plain doubles, or 40 16-byte granules {value, value ^ key}
constexpr int kBlock = 256;
constexpr int kAssocBlock = 64;   // association (one query per thread): one wave per workgroup, so the dispatcher balances SIMDs wave by wave

}  // namespace lili
