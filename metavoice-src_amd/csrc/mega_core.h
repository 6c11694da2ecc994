// Shared pieces of the persistent decode-step kernel (decode_mega.hip; tools/experiments/decode_mega_lds.hip is a measured-slower
// loader-wave variant kept out of the product library): constants, the tagged-granule hand-off and the per-wave polls.
//
// Hand-off (cdna_hip_programming.md Guideline 16, recipe R2): every activation vector is exchanged as 8-byte granules
// {payload: 2 x bf16, tag: 32-bit epoch}, written with ONE agent-scope (sc1, write-through) store by the producing lane
// and re-read with agent-scope loads by every consumer until all tags match -- the data is the flag: no fence, no
// barrier object, no dependence on workgroup placement.  tag = epoch x 256 + layer x 8 + phase + 1 with epoch = the step counter
// (one per decode step: advanced by block 0 at the end of every step of a multi-step launch, by sample_kernel when the token is
// drawn by a launch of its own): strictly increasing, so a stale granule can never match.  Layer index n_layer tags the logits,
// phase code 0 of the NEXT epoch the control word {token, position, done} that ends a step.
// Every wait is bounded; a timeout raises a device flag that the host turns into an error (a plain launch gives no
// residency guarantee: if another kernel occupies CUs the grid may not be co-resident).
#pragma once
#include <type_traits>

#include "attn_core.h"
#include "gemv_core.h"

#define MG_THREADS 512
#define MG_WAVES 8
#define MG_DIM 2048
#define MG_INTER 5632
#define MG_HEADS 16
#define MG_SPIN_LIMIT (1 << 18)
#define MG_STAMP_SLOTS 24  // s_memtime stamps per layer (MV_MEGA_STAMPS=1, tools/mega_stamps.py)

typedef u64 gran_t;

__device__ __forceinline__ void gran_store(gran_t* g, uint32_t payload, uint32_t tag) {
    __hip_atomic_store(g, ((u64)tag << 32) | payload, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ u64 gran_load(const gran_t* g) { return __hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// optional phase stamps: needs `stamping`, `stamp_base` and `lane` in scope (MV_MEGA_STAMPS=1, tools/mega_stamps.py)
#define MG_STAMP(l, k) do { if (stamping && lane == 0) stamp_base[(l) * MG_STAMP_SLOTS + (k)] = __builtin_amdgcn_s_memtime(); } while (0)

// The first wave whose wait runs out records what it was waiting for next to the abort flag: [1] the tag (epoch * 256 + layer * 8 +
// phase + 1), [2] its workgroup, [3] its wave.  mv_run prints them with the fall-back notice.
__device__ __forceinline__ void mega_report_timeout(uint32_t* abort_flag, uint32_t tag) {
    if ((threadIdx.x & 63) == 0 && atomicCAS(abort_flag + 1, 0u, tag) == 0u) {
        abort_flag[2] = blockIdx.x;
        abort_flag[3] = threadIdx.x >> 6;
    }
}

// The poll loads are inline asm ending in `s_waitcnt vmcnt(0)`: every load the wave issued before them has landed when the statement
// ends (loads retire in order) -- but hipcc does not look inside asm statements, keeps the weight / K/V loads it issued itself on its
// scoreboard and puts a counted `s_waitcnt vmcnt(N)` in front of their first use, where N no longer means what it assumed (LDS-direct
// requests and the next poll's loads were issued in between): a wait for traffic the use does not depend on.  This compiler-visible
// wait right behind the poll costs nothing (nothing is outstanding) and clears the scoreboard.
__device__ __forceinline__ void mega_loads_have_landed() {
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0), expcnt / lgkmcnt untouched
}

// The abort flag as the polls take it: a pointer, or a callable returning it that is evaluated only when a wait runs out (the persistent
// kernel reads the pointer from its launch arguments there instead of keeping it live in two SGPRs through every layer).
template <typename AF>
__device__ __forceinline__ uint32_t* mega_abort_ptr(AF af) {
    if constexpr (std::is_pointer_v<AF>) return af;
    else return af();
}

// One wave waits until all of ITS granules carry `tag` (no block barrier inside: the loaders only mirror the barrier after).
// Returns the number of attempts that came back incomplete (0 = the first look saw everything; the instrumented build records it).
// (Round 5: with this return value in place of `void` the one-utterance bf16 step is 7.5 us shorter -- 647.0-647.6 against 654.5-655.2 us, int8
// 591-594 against 596-597, A B A B on one box and again on a second, identical tokens, profiles/r05q, r05r -- although no caller of the
// product build reads it: the retry loops compile to slightly different control flow and every poll of the layer ends ~0.1 us earlier
// (stamps).  A third form -- the compiler-visible wait of mega_loads_have_landed() behind the loop instead of inside it, the request
// addresses computed once in front of the loop -- measured 9 us LONGER than the `void` form (663.5-664.3).  The form below is the
// measured one; treat the loop's shape as part of the kernel's tuning, not as style.)
template <int PER, typename AF, typename Fetch, typename Sink>
__device__ __forceinline__ int wave_poll(uint32_t tag, AF abort_flag_of, int* s_abort, Fetch fetch, Sink sink) {
    for (int tries = 0;; ++tries) {
        u64 v[PER];
        bool have[PER];
#pragma unroll
        for (int k = 0; k < PER; ++k) have[k] = fetch(k, v[k]);
        bool ok = true;
#pragma unroll
        for (int k = 0; k < PER; ++k) ok = ok && (!have[k] || (uint32_t)(v[k] >> 32) == tag);
        if (__all(ok)) {
#pragma unroll
            for (int k = 0; k < PER; ++k)
                if (have[k]) sink(k, (uint32_t)v[k]);
            return tries;
        }
        if ((tries & 1023) == 1023) {  // a peer is missing (grid not co-resident?) or already gave up
            uint32_t* abort_flag = mega_abort_ptr(abort_flag_of);
            const bool peer = __hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u;
            if (peer || tries >= MG_SPIN_LIMIT) {
                if (!peer) mega_report_timeout(abort_flag, tag);
                __hip_atomic_store(abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                *(volatile int*)s_abort = 1;
                return tries;
            }
        }
        __builtin_amdgcn_s_sleep(2);
    }
}
// PER 16-byte agent-scope loads AND their wait in ONE asm statement: hipcc does not know that an inline-asm load completes later
// (cdna_hip_programming.md 5.7), so with the loads and the s_waitcnt in separate statements it is free to copy or spill an output
// register in between -- before the data has landed.  It did, once the kernel grew (256 VGPRs, spills): every tag compared stale
// and the wait ran out.  Early-clobber outputs: the address registers are read by later loads of the same statement.
template <int PER>
__device__ __forceinline__ void poll16_loads(u32x4 (&v)[PER], const gran_t* const (&p)[PER]) {
    static_assert(PER == 1 || PER == 2 || PER == 3 || PER == 4 || PER == 6 || PER == 8 || PER == 12, "add the instance");
    if constexpr (PER == 1)
        asm volatile("global_load_dwordx4 %0, %1, off sc1\n\t"
                     "s_waitcnt vmcnt(0)"
                     : "=&v"(v[0])
                     : "v"(p[0])
                     : "memory");
    else if constexpr (PER == 2)
        asm volatile("global_load_dwordx4 %0, %2, off sc1\n\t"
                     "global_load_dwordx4 %1, %3, off sc1\n\t"
                     "s_waitcnt vmcnt(0)"
                     : "=&v"(v[0]), "=&v"(v[1])
                     : "v"(p[0]), "v"(p[1])
                     : "memory");
    else if constexpr (PER == 3)
        asm volatile("global_load_dwordx4 %0, %3, off sc1\n\t"
                     "global_load_dwordx4 %1, %4, off sc1\n\t"
                     "global_load_dwordx4 %2, %5, off sc1\n\t"
                     "s_waitcnt vmcnt(0)"
                     : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2])
                     : "v"(p[0]), "v"(p[1]), "v"(p[2])
                     : "memory");
    else if constexpr (PER == 4)
        asm volatile("global_load_dwordx4 %0, %4, off sc1\n\t"
                     "global_load_dwordx4 %1, %5, off sc1\n\t"
                     "global_load_dwordx4 %2, %6, off sc1\n\t"
                     "global_load_dwordx4 %3, %7, off sc1\n\t"
                     "s_waitcnt vmcnt(0)"
                     : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3])
                     : "v"(p[0]), "v"(p[1]), "v"(p[2]), "v"(p[3])
                     : "memory");
    else if constexpr (PER == 6)
        asm volatile("global_load_dwordx4 %0, %6, off sc1\n\t"
                     "global_load_dwordx4 %1, %7, off sc1\n\t"
                     "global_load_dwordx4 %2, %8, off sc1\n\t"
                     "global_load_dwordx4 %3, %9, off sc1\n\t"
                     "global_load_dwordx4 %4, %10, off sc1\n\t"
                     "global_load_dwordx4 %5, %11, off sc1\n\t"
                     "s_waitcnt vmcnt(0)"
                     : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5])
                     : "v"(p[0]), "v"(p[1]), "v"(p[2]), "v"(p[3]), "v"(p[4]), "v"(p[5])
                     : "memory");
    else if constexpr (PER == 8)
        asm volatile("global_load_dwordx4 %0, %8, off sc1\n\t"
                     "global_load_dwordx4 %1, %9, off sc1\n\t"
                     "global_load_dwordx4 %2, %10, off sc1\n\t"
                     "global_load_dwordx4 %3, %11, off sc1\n\t"
                     "global_load_dwordx4 %4, %12, off sc1\n\t"
                     "global_load_dwordx4 %5, %13, off sc1\n\t"
                     "global_load_dwordx4 %6, %14, off sc1\n\t"
                     "global_load_dwordx4 %7, %15, off sc1\n\t"
                     "s_waitcnt vmcnt(0)"
                     : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7])
                     : "v"(p[0]), "v"(p[1]), "v"(p[2]), "v"(p[3]), "v"(p[4]), "v"(p[5]), "v"(p[6]), "v"(p[7])
                     : "memory");
    else if constexpr (PER == 12)
        asm volatile("global_load_dwordx4 %0, %12, off sc1\n\t"
                     "global_load_dwordx4 %1, %13, off sc1\n\t"
                     "global_load_dwordx4 %2, %14, off sc1\n\t"
                     "global_load_dwordx4 %3, %15, off sc1\n\t"
                     "global_load_dwordx4 %4, %16, off sc1\n\t"
                     "global_load_dwordx4 %5, %17, off sc1\n\t"
                     "global_load_dwordx4 %6, %18, off sc1\n\t"
                     "global_load_dwordx4 %7, %19, off sc1\n\t"
                     "global_load_dwordx4 %8, %20, off sc1\n\t"
                     "global_load_dwordx4 %9, %21, off sc1\n\t"
                     "global_load_dwordx4 %10, %22, off sc1\n\t"
                     "global_load_dwordx4 %11, %23, off sc1\n\t"
                     "s_waitcnt vmcnt(0)"
                     : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7]), "=&v"(v[8]), "=&v"(v[9]), "=&v"(v[10]), "=&v"(v[11])
                     : "v"(p[0]), "v"(p[1]), "v"(p[2]), "v"(p[3]), "v"(p[4]), "v"(p[5]), "v"(p[6]), "v"(p[7]), "v"(p[8]), "v"(p[9]), "v"(p[10]), "v"(p[11])
                     : "memory");
}

// The same with 16-byte agent-scope loads: two granules per lane per request, each 8-byte half validated by its own tag
// (profiles/r01g_handoff: 2.0 vs 2.3 us per idle 2048-granule hand-off).  index(k) = pair index of the lane's k-th request or -1.
template <int PER, typename AF, typename Index, typename Sink>
__device__ __forceinline__ int wave_poll16(const gran_t* g, uint32_t tag, AF abort_flag_of, int* s_abort, Index index, Sink sink) {
    for (int tries = 0;; ++tries) {
        u32x4 v[PER];
        int n[PER];
        const gran_t* p[PER];
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            n[k] = index(k);
            p[k] = g + 2 * max(n[k], 0);
        }
        poll16_loads<PER>(v, p);
        mega_loads_have_landed();
        bool ok = true;
#pragma unroll
        for (int k = 0; k < PER; ++k) ok = ok && (n[k] < 0 || (v[k].y == tag && v[k].w == tag));
        if (__all(ok)) {
#pragma unroll
            for (int k = 0; k < PER; ++k)
                if (n[k] >= 0) {
                    if constexpr (std::is_invocable_v<Sink, std::integral_constant<int, 0>, int, uint32_t, uint32_t>)
                        sink(k, n[k], v[k].x, v[k].z);   // sinks that keep the payload in registers index by the (unrolled) request number
                    else sink(n[k], v[k].x, v[k].z);
                }
            return tries;
        }
        if ((tries & 1023) == 1023) {
            uint32_t* abort_flag = mega_abort_ptr(abort_flag_of);
            const bool peer = __hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u;
            if (peer || tries >= MG_SPIN_LIMIT) {
                if (!peer) mega_report_timeout(abort_flag, tag);
                __hip_atomic_store(abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                *(volatile int*)s_abort = 1;
                return tries;
            }
        }
        __builtin_amdgcn_s_sleep(1);
    }
}
