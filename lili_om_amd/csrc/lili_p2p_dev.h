// Device side of the peer-to-peer exchange (lili_p2p.hip has the description and the host side).  ONE wave runs an exchange:
// lane l carries the 8-byte words l and l + 64 of the record (count <= 96), so the collective can sit inside another kernel's
// single-wave tail (the count kernel, the partial reduction + Gauss-Newton update) without block barriers.
#pragma once
#include <hip/hip_runtime.h>

namespace lili {

constexpr int kP2PMaxWorld = 16;
constexpr int kP2PSlotBytes = 1024;                 // 96 payload words + flag word, padded
constexpr int kP2PSlotWords = kP2PSlotBytes / 8;
constexpr int kP2PFlagWord = 120;
constexpr int kP2PMaxCount = 96;                    // 8-byte words per record (the Gram record has 72)
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

// All 64 lanes of ONE wave.  w0 / w1: this rank's words `lane` and `lane + 64` (ignored beyond count).  On return s0 / s1 hold the
// sums over the ranks, added in rank order, as f64 (F64) or int32 in the low half of the word.  Returns false if a peer's record
// did not arrive within the communicator's timeout (status word set), or if the communicator had failed before (`was_dead`, the value
// of p2p_dead_word): then nothing is published.  A rank that gives up also raises the failure word in EVERY peer's mailbox, so that
// the peers — which may be waiting for this rank's next record — fail at their next look instead of after their own timeout.
template <bool F64>
__device__ __forceinline__ bool p2p_exchange_wave(const P2PView& v, int count, unsigned long long w0, unsigned long long w1,
                                                  unsigned long long& s0, unsigned long long& s1, unsigned long long was_dead) {
    const int lane = threadIdx.x & 63;
    const int par = (int)(v.seq & 1ull);
    const int world = v.world;
    if (__any(was_dead != 0ull)) { if (lane == 0) *v.status = 1; return false; }
    // 1. my record into slot[rank] of every mailbox (own included): write-through system-scope stores
    for (int p = 0; p < world; p++) {
        unsigned long long* slot = v.box[p] + (size_t)(par * kP2PMaxWorld + v.rank) * kP2PSlotWords;
        if (lane < count) __hip_atomic_store(slot + lane, w0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (lane + 64 < count) __hip_atomic_store(slot + lane + 64, w1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
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
        double a = 0.0, b = 0.0;
        for (int r = 0; r < world; r++) {
            if (lane < count) a += __longlong_as_double((long long)__hip_atomic_load(base + (size_t)r * kP2PSlotWords + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM));
            if (lane + 64 < count) b += __longlong_as_double((long long)__hip_atomic_load(base + (size_t)r * kP2PSlotWords + lane + 64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM));
        }
        s0 = (unsigned long long)__double_as_longlong(a); s1 = (unsigned long long)__double_as_longlong(b);
    } else {
        int a = 0, b = 0;
        for (int r = 0; r < world; r++) {
            if (lane < count) a += (int)(unsigned)__hip_atomic_load(base + (size_t)r * kP2PSlotWords + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            if (lane + 64 < count) b += (int)(unsigned)__hip_atomic_load(base + (size_t)r * kP2PSlotWords + lane + 64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        s0 = (unsigned long long)(unsigned)a; s1 = (unsigned long long)(unsigned)b;
    }
    return true;
}

}  // namespace lili
