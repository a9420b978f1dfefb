// Device side of the peer-to-peer exchange (lili_p2p.hip has the description and the host side).  ONE wave runs an exchange:
// lane l carries the 8-byte words l, l + 64, l + 128, ... of the record (W words per lane, count <= 64 W <= 640: the Gram records of a
// whole sliding window, 8 slots x 72 doubles, go out as ONE exchange), so the collective can sit inside another kernel's single-wave
// tail (the count kernel, the partial reduction + Gauss-Newton update) without block barriers.
#pragma once
#include <hip/hip_runtime.h>

namespace lili {

constexpr int kP2PMaxWorld = 16;
constexpr int kP2PSlotBytes = 8192;                 // 640 payload words + flag word, padded
constexpr int kP2PSlotWords = kP2PSlotBytes / 8;
constexpr int kP2PFlagWord = 1000;
constexpr int kP2PMaxCount = 640;                   // 8-byte words per record (one Gram record has 72, a window of 8 slots 576)
constexpr int kP2PMaxPerLane = kP2PMaxCount / 64;
constexpr int kP2PDeadWord = 2 * kP2PMaxWorld * kP2PSlotWords;   // behind the slots of a mailbox: set once a wait of its owner has given up

struct P2PView {
    unsigned long long* box[kP2PMaxWorld];          // mailbox of every rank (own included), [2 parities][kP2PMaxWorld sources][kP2PSlotWords]
    int rank, world;
    int* status;                                    // pinned host word: set to 1 when a wait gives up
    unsigned long long seq;                         // sequence number of THIS exchange (0 = no exchange: view unused)
    long long timeout_ticks;                        // how long a wait may last, in 100 MHz ticks of s_memrealtime (lili_p2p_set_timeout)
};

// The sticky failure word of this rank's mailbox (0 = healthy).  Kernels that end with an exchange request it at their START, so that its
// latency hides behind their own loads, and hand the value to p2p_exchange_wave: a rank whose communicator has failed must not publish
// another record (its peers would keep adding the records of a rank that no longer applies the updates — ADVICE r2).
__device__ __forceinline__ unsigned long long p2p_dead_word(const P2PView& v) {
    return v.seq ? __hip_atomic_load(v.box[v.rank] + kP2PDeadWord, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : 0ull;
}

// All 64 lanes of ONE wave.  w[i]: this rank's word `lane + 64 i` (ignored beyond count).  On return s[i] holds the sums over the ranks,
// added in rank order, as f64 (F64) or int32 in the low half of the word.  Returns false if a peer's record did not arrive within the
// communicator's timeout (status word set), or if the communicator had failed before (`was_dead`, the value of p2p_dead_word): then
// nothing is published.  A rank that gives up also raises the failure word in EVERY peer's mailbox, so that the peers — which may be
// waiting for this rank's next record — fail at their next look instead of after their own timeout.
template <bool F64, int W>
__device__ __forceinline__ bool p2p_exchange_words(const P2PView& v, int count, const unsigned long long (&w)[W], unsigned long long (&s)[W], unsigned long long was_dead) {
    static_assert(W >= 1 && W <= kP2PMaxPerLane, "words per lane");
    const int lane = threadIdx.x & 63;
    const int par = (int)(v.seq & 1ull);
    const int world = v.world;
    if (__any(was_dead != 0ull)) { if (lane == 0) *v.status = 1; return false; }
    // 1. my record into slot[rank] of every mailbox (own included): write-through system-scope stores
    for (int p = 0; p < world; p++) {
        unsigned long long* slot = v.box[p] + (size_t)(par * kP2PMaxWorld + v.rank) * kP2PSlotWords;
#pragma unroll
        for (int i = 0; i < W; i++) if (lane + 64 * i < count) __hip_atomic_store(slot + lane + 64 * i, w[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    // 2. once the wave's store counter has drained the payload is in the peers' memory; then the flags
    //    (MI355X_MICROARCH.md, inter-workgroup visibility: "sc1 payload -> asm vmcnt(0) -> sc1 flag", here at system scope)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    if (lane < world) {
        unsigned long long* slot = v.box[lane] + (size_t)(par * kP2PMaxWorld + v.rank) * kP2PSlotWords;
        __hip_atomic_store(slot + kP2PFlagWord, v.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    // 3. every source's flag in MY mailbox
    bool ok = true;
    unsigned long long* dead = v.box[v.rank] + kP2PDeadWord;
    if (lane < world) {
        const unsigned long long* flag = v.box[v.rank] + (size_t)(par * kP2PMaxWorld + lane) * kP2PSlotWords + kP2PFlagWord;
        const long long t0 = (long long)__builtin_amdgcn_s_memrealtime();      // 100 MHz
        for (unsigned spins = 0;; spins++) {
            if (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) == v.seq) break;
            // a peer that gave up has raised my failure word: look at it now and then (it lives in my own memory)
            if ((spins & 63u) == 63u && __hip_atomic_load(dead, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0ull) { ok = false; break; }
            if ((long long)__builtin_amdgcn_s_memrealtime() - t0 > v.timeout_ticks) { ok = false; break; }
            __builtin_amdgcn_s_sleep(2);
        }
    }
    if (!__all(ok)) {
        if (lane == 0) *v.status = 1;
        if (lane < world) __hip_atomic_store(v.box[lane] + kP2PDeadWord, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);      // mine and every peer's
        return false;
    }
    // 4. sums in rank order; the payload is read with system-scope loads too (they bypass L1 / L2: nothing stale to invalidate)
    const unsigned long long* base = v.box[v.rank] + (size_t)(par * kP2PMaxWorld) * kP2PSlotWords;
    if (F64) {
        double a[W];
#pragma unroll
        for (int i = 0; i < W; i++) a[i] = 0.0;
        for (int r = 0; r < world; r++) {
#pragma unroll
            for (int i = 0; i < W; i++)
                if (lane + 64 * i < count) a[i] += __longlong_as_double((long long)__hip_atomic_load(base + (size_t)r * kP2PSlotWords + lane + 64 * i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM));
        }
#pragma unroll
        for (int i = 0; i < W; i++) s[i] = (unsigned long long)__double_as_longlong(a[i]);
    } else {
        int a[W];
#pragma unroll
        for (int i = 0; i < W; i++) a[i] = 0;
        for (int r = 0; r < world; r++) {
#pragma unroll
            for (int i = 0; i < W; i++)
                if (lane + 64 * i < count) a[i] += (int)(unsigned)__hip_atomic_load(base + (size_t)r * kP2PSlotWords + lane + 64 * i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
#pragma unroll
        for (int i = 0; i < W; i++) s[i] = (unsigned long long)(unsigned)a[i];
    }
    return true;
}
// two words per lane (one Gram record, the two counters)
template <bool F64>
__device__ __forceinline__ bool p2p_exchange_wave(const P2PView& v, int count, unsigned long long w0, unsigned long long w1,
                                                  unsigned long long& s0, unsigned long long& s1, unsigned long long was_dead) {
    const unsigned long long w[2] = {w0, w1};
    unsigned long long s[2] = {0ull, 0ull};
    const bool ok = p2p_exchange_words<F64, 2>(v, count, w, s, was_dead);
    s0 = s[0]; s1 = s[1];
    return ok;
}

}  // namespace lili
