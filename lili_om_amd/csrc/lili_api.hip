// C-ABI of liblili_hip.so (include/lili_hip.h): context, buffers, launches.  Host code only; the
// kernels live in lili_s2m.hip; the map index is in lili_map.hip, the matcher in lili_match.hip.  There is no CPU fallback anywhere in this file:
// without a gfx950 device lili_ctx_create fails with LILI_E_NODEVICE.
#include "lili_launch.h"

#include <cmath>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

constexpr size_t kPinBytes = 64 * 1024;
// Round 5: the small reads of a synchronisation are gathered by ONE kernel that writes them into the page-locked scratch across PCIe (k_readback_gather), enqueued by
// lili_readback_finish right before it synchronises — a frame of the front-end pipeline issued eleven 4-us copy launches for its counts, boxes, flags and pose.  The
// sources are therefore read at FINISH time: every caller adds its items behind the kernels that produce them and enqueues nothing that rewrites them before the finish.
// An item on another stream than the context's, a large one, or a full table goes the old way (a copy enqueued at add time).
namespace lili {
constexpr int kGatherMax = 12;
struct GatherTable { const unsigned char* src[kGatherMax]; unsigned off[kGatherMax], bytes[kGatherMax]; int n; };
__global__ __launch_bounds__(256) void k_readback_gather(GatherTable t, unsigned char* __restrict__ dst) {
    for (int k = 0; k < t.n; k++) {
        const unsigned char* s = t.src[k];
        unsigned char* d = dst + t.off[k];
        const unsigned nb = t.bytes[k];
        if (((reinterpret_cast<uintptr_t>(s) | reinterpret_cast<uintptr_t>(d) | nb) & 3u) == 0) {
            for (unsigned i = threadIdx.x; i < nb / 4; i += 256) reinterpret_cast<unsigned*>(d)[i] = reinterpret_cast<const unsigned*>(s)[i];
        } else for (unsigned i = threadIdx.x; i < nb; i += 256) d[i] = s[i];
    }
}
}  // namespace lili
int lili_readback_add(lili_ctx* ctx, void* dst, const void* d_src, size_t bytes, hipStream_t stream) {
    if (bytes == 0) return LILI_OK;
    hipStream_t s = stream ? stream : ctx->stream;
    if (!ctx->h_pin) {
        HIPCHK(hipHostMalloc(reinterpret_cast<void**>(&ctx->h_pin), kPinBytes, hipHostMallocDefault));
        void* d = nullptr;
        if (hipHostGetDevicePointer(&d, ctx->h_pin, 0) != hipSuccess) { (void)hipGetLastError(); d = nullptr; }
        ctx->h_pin_dev = static_cast<unsigned char*>(d);
    }
    const size_t off = (ctx->h_pin_used + 15) & ~(size_t)15;
    if (off + bytes > kPinBytes) {        // (does not happen with the library's own reads; a large one goes the plain way)
        HIPCHK(hipMemcpyAsync(dst, d_src, bytes, hipMemcpyDeviceToHost, s));
        return LILI_OK;
    }
    const bool lazy = ctx->h_pin_dev && s == ctx->stream && bytes <= 16384 && ctx->readback_gather && (int)ctx->h_pin_lazy.size() < lili::kGatherMax;
    if (lazy) ctx->h_pin_lazy.push_back({d_src, off, bytes});
    else HIPCHK(hipMemcpyAsync(ctx->h_pin + off, d_src, bytes, hipMemcpyDeviceToHost, s));
    ctx->h_pin_items.push_back({dst, off, bytes});
    ctx->h_pin_used = off + bytes;
    return LILI_OK;
}
int lili_lazy_sources_clear_of(lili_ctx* ctx, const void* p, size_t bytes) {
    const char* a = static_cast<const char*>(p);
    for (const auto& it : ctx->h_pin_lazy) {
        const char* s = static_cast<const char*>(it.src);
        if (a < s + it.bytes && s < a + bytes) return ctx->fail(LILI_E_STATE, "internal: a fill of the scratch words would overwrite the source of a pending read-back");
    }
    return LILI_OK;
}
int lili_readback_finish(lili_ctx* ctx, hipStream_t stream) {
    hipStream_t s = stream ? stream : ctx->stream;
    hipError_t e = hipSuccess;
    if (ctx->extract_join_pending) {      // an extraction on its own stream (frame_extract_stream): its products and counts are read from here on
        ctx->extract_join_pending = false;
        e = hipStreamWaitEvent(ctx->stream, ctx->join_ev[lili_ctx::kExtractSide], 0);
    }
    if (e == hipSuccess && !ctx->h_pin_lazy.empty()) {      // (lazy items are always on the context's stream; a finish of another stream synchronises the context's as well)
        lili::GatherTable t{};
        t.n = (int)ctx->h_pin_lazy.size();
        for (int k = 0; k < t.n; k++) { t.src[k] = static_cast<const unsigned char*>(ctx->h_pin_lazy[k].src); t.off[k] = (unsigned)ctx->h_pin_lazy[k].off; t.bytes[k] = (unsigned)ctx->h_pin_lazy[k].bytes; }
        hipLaunchKernelGGL(lili::k_readback_gather, dim3(1), dim3(256), 0, ctx->stream, t, ctx->h_pin_dev);
        e = hipGetLastError();
        ctx->h_pin_lazy.clear();
        if (e == hipSuccess && s != ctx->stream) e = hipStreamSynchronize(ctx->stream);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e == hipSuccess) for (const auto& it : ctx->h_pin_items) std::memcpy(it.dst, ctx->h_pin + it.off, it.bytes);
    ctx->h_pin_items.clear();
    ctx->h_pin_used = 0;
    ctx->readback_gen++;
    HIPCHK(e);
    return LILI_OK;
}

// copies / converts a described cloud into a device float4 array (x, y, z, aux)
// The device-side address of a caller's PAGE-LOCKED host buffer (lili_host_alloc / hipHostMalloc / hipHostRegister), or nullptr: pageable memory, or not aligned to
// `align` bytes.  Asked on EVERY call that wants to let a kernel read or write the buffer across PCIe (0.06-0.16 us: tools/ptr_attr_cost.hip) — a cached "page-locked"
// would outlive the buffer: the address of a freed page-locked buffer can come back from malloc as pageable memory, and a kernel would fault on it.
void* lili_pinned_dev_ptr(const void* host, size_t align) {
    if (!host) return nullptr;
    hipPointerAttribute_t attr{};
    void* d = nullptr;
    if (hipPointerGetAttributes(&attr, host) == hipSuccess && attr.type == hipMemoryTypeHost && hipHostGetDevicePointer(&d, const_cast<void*>(host), 0) == hipSuccess &&
        d && (reinterpret_cast<uintptr_t>(d) & (align - 1)) == 0) return d;
    (void)hipGetLastError();      // pageable memory is unknown to the runtime: not an error of the call
    return nullptr;
}

int lili_ingest_cloud(lili_ctx* ctx, const lili_cloud* c, DevBuf& out_f4, unsigned* d_bbox) {
    ARGCHK(c && (c->n == 0 || c->data), "cloud: null data");
    ARGCHK(c->stride >= 12 && c->stride % 4 == 0, "cloud: stride must be a multiple of 4 and >= 12");
    ARGCHK(c->aux_offset < 0 || (size_t)c->aux_offset + 4 <= c->stride, "cloud: aux_offset outside the point");
    ARGCHK(c->n < (size_t)1 << 31, "cloud: too many points");
    if (c->n == 0) return LILI_OK;
    HIPCHK(out_f4.ensure(c->n * sizeof(float4)));
    const unsigned char* src = reinterpret_cast<const unsigned char*>(c->data);
    if (c->mem == LILI_MEM_HOST) {
        // Round 4: a PAGE-LOCKED cloud (a driver's DMA buffer, lili_host_alloc) is read by the conversion kernel where it lies, across PCIe — no copy-engine transfer,
        // no wait for its completion signal before the kernel may start.  Measured (tools/rot_host_time.py, tools/livox_timeline.py): a 1.15 MB Livox scan -28 us per call;
        // a 3.2 MB ROT scan +9 us (the kernel reads PCIe at ~41 GB/s, the copy engine at ~55) — but the call then takes the same time from the first scan on, where a
        // copy-engine upload next to a download of the same call ran 0.35-0.45 ms for its first fifty-odd calls (the bench's bimodal extract_rot).  Pageable memory:
        // one staged transfer of the rows as they are.
        if (void* d = lili_pinned_dev_ptr(c->data, 4)) src = static_cast<const unsigned char*>(d);
        else {
            HIPCHK(ctx->staging.ensure(c->n * c->stride));
            HIPCHK(hipMemcpyAsync(ctx->staging.p, c->data, c->n * c->stride, hipMemcpyHostToDevice, ctx->stream));
            src = ctx->staging.as<unsigned char>();
        }
    } else ARGCHK(c->mem == LILI_MEM_DEVICE, "cloud: bad mem");
    // with a bounding box: a bounded grid (grid-stride loop) so that the box costs a few thousand atomics, not one set per 256 points
    const int nb = d_bbox ? std::min(nblocks((int64_t)c->n, kBlock), 4096) : nblocks((int64_t)c->n, kBlock);
    hipLaunchKernelGGL(k_cloud_to_f4, dim3(nb), dim3(kBlock), 0, ctx->stream, src, (int)c->n, (int)c->stride, c->aux_offset, out_f4.as<float4>(), d_bbox);
    HIPCHK(hipGetLastError());
    return LILI_OK;
}

extern "C" {

int lili_abi_version(void) { return LILI_ABI_VERSION; }

void* lili_host_alloc(size_t bytes) {
    void* p = nullptr;
    if (bytes == 0 || hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    return p;
}
void lili_host_free(void* p) { if (p) (void)hipHostFree(p); }


int lili_ctx_create(lili_ctx** out, int device, void* stream) {
    if (!out) return LILI_E_ARG;
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0 || device < 0 || device >= count) return LILI_E_NODEVICE;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return LILI_E_NODEVICE;
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) return LILI_E_NODEVICE;   // kernels are built for gfx950 only
    if (hipSetDevice(device) != hipSuccess) return LILI_E_NODEVICE;
    lili_ctx* ctx = new (std::nothrow) lili_ctx();
    if (!ctx) return LILI_E_NOMEM;
    ctx->device = device;
    ctx->n_simd = prop.multiProcessorCount * 4;   // CDNA: four SIMDs per CU
    if (const char* e = std::getenv("LILI_FRAME_EXTRACT_STREAM")) ctx->frame_extract_stream = std::atoi(e) != 0;      // (A/B aid for whole-bench runs; lili_set_option "frame_extract_stream" otherwise)
    if (stream) ctx->stream = reinterpret_cast<hipStream_t>(stream);
    else {
        if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) { delete ctx; return LILI_E_HIP; }
        ctx->own_stream = true;
    }
    bool ok = ctx->states.ensure(sizeof(SlotState) * LILI_MAX_SLOTS) == hipSuccess &&
              ctx->gram.ensure(sizeof(double) * LILI_GRAM_DOUBLES * LILI_MAX_SLOTS) == hipSuccess &&
              ctx->misc.ensure(kMiscAlloc) == hipSuccess &&
              ctx->pose_pub.ensure(sizeof(double) * kPubReplicas * kPubStride * LILI_MAX_SLOTS) == hipSuccess &&
              hipMemsetAsync(ctx->pose_pub.p, 0, sizeof(double) * kPubReplicas * kPubStride * LILI_MAX_SLOTS, ctx->stream) == hipSuccess &&
              hipMemsetAsync(ctx->states.p, 0, sizeof(SlotState) * LILI_MAX_SLOTS, ctx->stream) == hipSuccess &&
              hipMemsetAsync(ctx->gram.p, 0, sizeof(double) * LILI_GRAM_DOUBLES * LILI_MAX_SLOTS, ctx->stream) == hipSuccess;
    if (!ok) { lili_ctx_destroy(ctx); return LILI_E_HIP; }
    *out = ctx;
    return LILI_OK;
}

void lili_ctx_destroy(lili_ctx* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    if (ctx->build_stream) { (void)hipStreamSynchronize(ctx->build_stream); (void)hipStreamDestroy(ctx->build_stream); }
    for (int k = 0; k < 2; k++) { if (ctx->build_done[k]) (void)hipEventDestroy(ctx->build_done[k]); if (ctx->main_mark[k]) (void)hipEventDestroy(ctx->main_mark[k]); }
    if (ctx->cloud_ready) (void)hipEventDestroy(ctx->cloud_ready);
    for (auto& m : ctx->map) { m.sorted_f.release(); m.aux_sorted_f.release(); m.cell_start_f.release(); m.cell_start9.release(); m.cell_start9_f.release(); m.row9.release(); m.sorted.release(); m.aux_sorted.release(); m.cell_start.release(); m.cell_tmp.release(); m.pt_cell.release(); m.block_sums.release(); }
    for (auto& s : ctx->slots) for (auto& k : s.k) { k.q.release(); k.rec0.release(); k.rec1.release(); k.valid.release(); k.dbg_idx.release(); k.dbg_d2.release(); k.partials.release(); k.partials_wave.release(); k.block_counts.release(); }
    if (ctx->h_records) { (void)hipHostFree(ctx->h_records); ctx->h_records = nullptr; }
    if (ctx->h_pin) { (void)hipHostFree(ctx->h_pin); ctx->h_pin = nullptr; }
    if (ctx->h_state_mirror) { (void)hipHostFree(ctx->h_state_mirror); ctx->h_state_mirror = nullptr; ctx->h_state_mirror_dev = nullptr; }
    ctx->states.release(); ctx->staging.release(); ctx->gram.release(); ctx->win_rec.release(); ctx->win_counts.release(); ctx->misc.release();
    if (ctx->ext_rot && ctx->ext_rot_free) ctx->ext_rot_free(ctx->ext_rot);
    if (ctx->ext_livox && ctx->ext_livox_free) ctx->ext_livox_free(ctx->ext_livox);
    if (ctx->ext_voxel && ctx->ext_voxel_free) ctx->ext_voxel_free(ctx->ext_voxel);
    for (auto& st : ctx->side) if (st) (void)hipStreamDestroy(st);
    if (ctx->fork_ev) (void)hipEventDestroy(ctx->fork_ev);
    if (ctx->extract_fork_ev) (void)hipEventDestroy(ctx->extract_fork_ev);
    for (auto& e : ctx->join_ev) if (e) (void)hipEventDestroy(e);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

const char* lili_last_error(const lili_ctx* ctx) { return ctx ? ctx->err.c_str() : "null context"; }

int lili_sync(lili_ctx* ctx) {
    if (!ctx) return LILI_E_ARG;
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return LILI_OK;
}

int lili_set_debug(lili_ctx* ctx, int keep_neighbors) {
    if (!ctx) return LILI_E_ARG;
    ctx->keep_nn = keep_neighbors != 0;
    return LILI_OK;
}

int lili_set_option(lili_ctx* ctx, const char* name, int value) {
    if (!ctx || !name) return LILI_E_ARG;
    if (std::strcmp(name, "grid_reach") == 0) { if (value != 1 && value != 2) return ctx->fail(LILI_E_ARG, "grid_reach must be 1 or 2"); ctx->grid_reach = value; return LILI_OK; }
    if (std::strcmp(name, "cell_pct") == 0) { if (value < 50 || value > 100) return ctx->fail(LILI_E_ARG, "cell_pct must be in 50..100"); ctx->cell_pct = value; return LILI_OK; }
    if (std::strcmp(name, "fuse_tail") == 0) { ctx->fuse_tail = value != 0; return LILI_OK; }   // any time: the block partition does not depend on it
    if (std::strcmp(name, "merge_kinds") == 0) { ctx->merge_kinds = value != 0; return LILI_OK; }
    if (std::strcmp(name, "fuse_lin") == 0) { ctx->fuse_lin = value != 0; return LILI_OK; }
    if (std::strcmp(name, "sort_fused_max_tiles") == 0) { ctx->sort_fused_max_tiles = std::max(0, value); return LILI_OK; }
    if (std::strcmp(name, "sort_fused_scan") == 0) { ctx->sort_fused_scan = value != 0; return LILI_OK; }      // radix passes of short sorts without the scan launch (0: three launches per pass, A/B)
    if (std::strcmp(name, "persistent_iterate") == 0) { ctx->persistent_iterate = value != 0; return LILI_OK; }      // small scans: lili_s2m_iterate* as one persistent launch per registration (0: launch by launch, A/B)
    if (std::strcmp(name, "count_barrier") == 0) { ctx->count_barrier = value != 0; return LILI_OK; }      // ROT small launches: association + count barrier + linearisation in one launch (0: three launches, A/B)
    if (std::strcmp(name, "assoc_lpq") == 0) {      // lanes per query of the association: 0 = by launch size, 1 = always one lane per query, 2 / 4 / 8 / 16 forced
        if (value != 0 && value != 1 && value != 2 && value != 4 && value != 8 && value != 16) return ctx->fail(LILI_E_ARG, "assoc_lpq must be 0 (auto), 1, 2, 4, 8 or 16");
        ctx->assoc_lpq = value; return LILI_OK;
    }
    if (std::strcmp(name, "fuse_lin_block") == 0) { if (value != 0 && value != kAssocBlock && value != kBlock) return ctx->fail(LILI_E_ARG, "fuse_lin_block must be 0 (auto), 64 or 256"); ctx->fuse_lin_block = value; return LILI_OK; }
    if (std::strcmp(name, "fine_grid") == 0) { ctx->fine_grid = value != 0; return LILI_OK; }
    if (std::strcmp(name, "sort_ride_hist") == 0) { ctx->sort_ride_hist = value != 0; return LILI_OK; }
    if (std::strcmp(name, "frame_guess_counts") == 0) { ctx->frame_guess_counts = value != 0; return LILI_OK; }
    if (std::strcmp(name, "frame_extract_stream") == 0) { ctx->frame_extract_stream = value != 0; return LILI_OK; }
    if (std::strcmp(name, "rot_fold") == 0) { ctx->rot_fold = value != 0; return LILI_OK; }
    if (std::strcmp(name, "rot_segment_wait") == 0) { ctx->rot_segment_wait = value != 0; return LILI_OK; }
    if (std::strcmp(name, "super_rows") == 0) { ctx->super_rows = value != 0; return LILI_OK; }   // takes effect at the next lili_map_set
    if (std::strcmp(name, "fine_occupancy") == 0) { if (value < 2) return ctx->fail(LILI_E_ARG, "fine_occupancy must be >= 2"); ctx->fine_occupancy = value; return LILI_OK; }
    if (std::strcmp(name, "scan_lookback") == 0) { ctx->scan_lookback = value != 0; return LILI_OK; }
    if (std::strcmp(name, "map_narrow_counts") == 0) { ctx->map_narrow_counts = value != 0; for (auto& b : ctx->spec_box) b.wide_counts = false; return LILI_OK; }   // 0: 32-bit cell counters always (A/B)
    if (std::strcmp(name, "map_guess_box") == 0) { ctx->map_guess_box = value != 0; for (auto& b : ctx->spec_box) { const bool w = b.wide_counts; b = SpecBox{}; b.wide_counts = w; } return LILI_OK; }   // 0: every lili_map_set measures its box first (A/B)
    if (std::strcmp(name, "localmap_super_rows") == 0) { ctx->localmap_super_rows = value != 0; return LILI_OK; }
    if (std::strcmp(name, "localmap_incremental") == 0) { ctx->localmap_incremental = value != 0; return LILI_OK; }
    if (std::strcmp(name, "sort_digit_bits") == 0) { if (value != 4 && value != 8) return ctx->fail(LILI_E_ARG, "sort_digit_bits must be 4 or 8"); ctx->sort_digit_bits = value; return LILI_OK; }
    if (std::strcmp(name, "rot_atan") == 0) { if (value != 1 && value != 2) return ctx->fail(LILI_E_ARG, "rot_atan must be 1 or 2"); ctx->rot_atan = value; return LILI_OK; }
    if (std::strcmp(name, "p2p_fusion") == 0) { ctx->no_p2p_fusion = value == 0; return LILI_OK; }
    if (std::strcmp(name, "readback_gather") == 0) { ctx->readback_gather = value != 0; return LILI_OK; }      // small device-to-host reads of one synchronisation in ONE gather launch (0: a copy launch each, A/B)
    if (std::strcmp(name, "voxel_guess_bits") == 0) { ctx->voxel_guess_bits = value != 0; return LILI_OK; }      // VoxelGrid of > 8192 points without the host round trip for its box (0: measured, A/B)
    if (std::strcmp(name, "voxel_small") == 0) { ctx->voxel_small = value != 0; return LILI_OK; }      // VoxelGrid of <= 8192 points in one single-workgroup launch (0: the general chain, A/B)
#ifdef LILI_OVERLAP_GN
    if (std::strcmp(name, "overlap_gn") == 0) { ctx->overlap_gn = value != 0; return LILI_OK; }
#endif      // lili_s2m_iterate*: the association behind a reduction + GN kernel starts without waiting for it (0: three barriers per iteration)
    if (std::strcmp(name, "max_cells") == 0) { if (value < 1) return ctx->fail(LILI_E_ARG, "max_cells must be positive"); ctx->max_cells = value; return LILI_OK; }
    return ctx->fail(LILI_E_ARG, std::string("unknown option ") + name);
}

}  // extern "C"
