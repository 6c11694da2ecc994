// Micro-benchmark for the one lever left on the persistent decode kernel's layer time (profiles/r02b_fold/NOTES.md, "edge model"):
// a SECOND phase of weight rows in flight.
//
// decode_mega.hip requests a phase's weight rows into registers after the previous phase's dots, then polls the phase's input: the
// poll returns, in order, behind the rows, so an exchange costs the issue stall of its rows plus the landing of the last one
// (tools/handoff_bench.hip: 2.99 / 3.43 / 3.65 us per 16 KB exchange with 0 / 32 / 64 KB per CU requested per round).
//
// Here the rows of round r+1 go straight into LDS (global_load_lds_dwordx4: no registers, 2 x 8 waves x NL KB ring) and are
// requested DURING round r's wait for its input: a poll attempt is ONE asm statement = {the wave's poll loads, CH row requests queued
// BEHIND them, s_waitcnt vmcnt(CH)}: the poll never waits for a row that was requested after it, a failed attempt has put CH more KB
// per wave in flight, and whatever is left is requested after the publish.  The rows are consumed from LDS a round later.
// Same dependency chain and checksum as handoff_bench.hip (a stale payload changes the checksum).
//
//   mode "reg": per-wave 16-byte polls, rows into registers after the publish (decode_mega.hip today)
//   mode "db" : rows of the next round into LDS, CH per poll attempt
//
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -o tools/handoff_db_bench.bin tools/handoff_db_bench.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef unsigned long long u64;
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

#define HB_THREADS 512
#define HB_WGS 256
#define HB_SPIN (1 << 20)

struct Args {
    u64* gran;            // [2][NG]
    const u32x4* w;       // weight stream (zero filled)
    size_t region;        // u32x4 elements per round region
    int nregions, rounds;
    uint32_t* out;        // [HB_WGS]
    uint32_t* abort_flag;
};

__host__ __device__ inline uint32_t mix(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
__device__ __forceinline__ void gran_store(u64* g, uint32_t payload, uint32_t tag) {
    __hip_atomic_store(g, ((u64)tag << 32) | payload, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// one row request: 64 lanes x 16 B of global memory -> 1 KB of LDS at `lds` (wave-uniform byte address), no registers
__device__ __forceinline__ void row_to_lds(uint32_t lds, const u32x4* g) {
    asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(lds), "v"(g) : "memory", "m0");
}

// ONE poll attempt of a wave: its two 16-byte poll loads, then CH row requests, then the wait that covers the polls only.
template <int CH>
__device__ __forceinline__ void poll_attempt(u32x4 (&v)[2], const u64* p0, const u64* p1, const uint32_t (&lds)[CH > 0 ? CH : 1],
                                             const u32x4* const (&g)[CH > 0 ? CH : 1]) {
    static_assert(CH == 0 || CH == 1 || CH == 2 || CH == 4 || CH == 8, "add the instance");
#define HB_POLL "global_load_dwordx4 %0, %2, off sc1\n\tglobal_load_dwordx4 %1, %3, off sc1\n\t"
#define HB_ROW(l, a) "s_mov_b32 m0, %" #l "\n\tglobal_load_lds_dwordx4 %" #a ", off\n\t"
    if constexpr (CH == 0)
        asm volatile(HB_POLL "s_waitcnt vmcnt(0)" : "=&v"(v[0]), "=&v"(v[1]) : "v"(p0), "v"(p1) : "memory");
    else if constexpr (CH == 1)
        asm volatile(HB_POLL HB_ROW(4, 5) "s_waitcnt vmcnt(1)" : "=&v"(v[0]), "=&v"(v[1]) : "v"(p0), "v"(p1), "s"(lds[0]), "v"(g[0]) : "memory", "m0");
    else if constexpr (CH == 2)
        asm volatile(HB_POLL HB_ROW(4, 5) HB_ROW(6, 7) "s_waitcnt vmcnt(2)"
                     : "=&v"(v[0]), "=&v"(v[1])
                     : "v"(p0), "v"(p1), "s"(lds[0]), "v"(g[0]), "s"(lds[1]), "v"(g[1])
                     : "memory", "m0");
    else if constexpr (CH == 4)
        asm volatile(HB_POLL HB_ROW(4, 5) HB_ROW(6, 7) HB_ROW(8, 9) HB_ROW(10, 11) "s_waitcnt vmcnt(4)"
                     : "=&v"(v[0]), "=&v"(v[1])
                     : "v"(p0), "v"(p1), "s"(lds[0]), "v"(g[0]), "s"(lds[1]), "v"(g[1]), "s"(lds[2]), "v"(g[2]), "s"(lds[3]), "v"(g[3])
                     : "memory", "m0");
    else
        asm volatile(HB_POLL HB_ROW(4, 5) HB_ROW(6, 7) HB_ROW(8, 9) HB_ROW(10, 11) HB_ROW(12, 13) HB_ROW(14, 15) HB_ROW(16, 17) HB_ROW(18, 19)
                     "s_waitcnt vmcnt(8)"
                     : "=&v"(v[0]), "=&v"(v[1])
                     : "v"(p0), "v"(p1), "s"(lds[0]), "v"(g[0]), "s"(lds[1]), "v"(g[1]), "s"(lds[2]), "v"(g[2]), "s"(lds[3]), "v"(g[3]), "s"(lds[4]),
                       "v"(g[4]), "s"(lds[5]), "v"(g[5]), "s"(lds[6]), "v"(g[6]), "s"(lds[7]), "v"(g[7])
                     : "memory", "m0");
#undef HB_POLL
#undef HB_ROW
}

// DB = false: rows into registers after the publish (NL x 16 B per lane), poll behind them.
// DB = true : rows of round r+1 into the LDS ring, CH per poll attempt of round r, the rest after the publish; consumed from LDS.
template <bool DB, int NL, int CH>
__global__ void __launch_bounds__(HB_THREADS, 2) db_kernel(const Args a) {
    constexpr int NG = 2048, PER_WG = NG / HB_WGS;
    static_assert(!DB || NL == 0 || NL % (CH > 0 ? CH : 1) == 0, "NL must be a multiple of CH");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t* xs = (uint32_t*)smem;                 // [NG]
    u32x4* ring = (u32x4*)(smem + NG * 4);          // DB: [2][8 waves][NL rows][64 lanes]
    __shared__ int s_flag;
    const int tid0 = threadIdx.x, b = blockIdx.x;
    int lane = tid0 & 63, wave = tid0 >> 6, tid = tid0;
    u32x4 wv[!DB && NL > 0 ? NL : 1];
#pragma unroll
    for (int j = 0; j < (!DB && NL > 0 ? NL : 1); ++j) wv[j] = u32x4{0, 0, 0, 0};
    auto grow = [&](int r, int j) { return a.w + (size_t)(r % a.nregions) * a.region + ((size_t)j * HB_WGS + b) * HB_THREADS + tid; };
    const uint32_t ring_base = (uint32_t)(uintptr_t)(lds_ptr_t)ring;
    auto lds_row = [&](int buf, int j) { return __builtin_amdgcn_readfirstlane(ring_base + (uint32_t)((((buf * 8 + wave) * NL + j) * 64) * 16)); };
    auto issue_regs = [&](int r) {
#pragma unroll
        for (int j = 0; j < NL; ++j)
            if (!DB) wv[j] = __builtin_nontemporal_load(grow(r, j));
    };
    auto issue_lds = [&](int r, int j0) {  // rows j0 .. NL-1 of round r
        for (int j = j0; j < NL; ++j) row_to_lds(lds_row(r & 1, j), grow(r, j));
    };
    uint32_t S = 0x1234567u, z = 0;
    auto publish = [&](int r) {
        if (tid < PER_WG) {
            const int idx = b * PER_WG + tid;
            gran_store(a.gran + (size_t)(r & 1) * NG + idx, mix(S + (uint32_t)idx + (uint32_t)r + 1u + z), (uint32_t)r + 1u);
        }
    };
    if (DB) issue_lds(0, 0); else issue_regs(0);
    publish(0);
    if (tid == 0) s_flag = 0;
    __syncthreads();

    for (int r = 0; r < a.rounds; ++r) {
        asm volatile("" : "+v"(lane), "+v"(wave));  // opaque ids: address arithmetic stays next to its use
        tid = wave * 64 + lane;
        const uint32_t tag = (uint32_t)r + 1u;
        const u64* buf = a.gran + (size_t)(r & 1) * NG;
        const u64 *p0 = buf + 2 * tid, *p1 = buf + 2 * (tid + HB_THREADS);  // 1024 pairs: two 16-byte polls per lane
        int issued = 0;  // rows of round r+1 requested so far
        for (int tries = 0;; ++tries) {
            u32x4 v[2];
            if (DB && CH > 0 && issued < NL && r + 1 < a.rounds) {
                uint32_t l[CH > 0 ? CH : 1];
                const u32x4* g[CH > 0 ? CH : 1];
#pragma unroll
                for (int c = 0; c < CH; ++c) { l[c] = lds_row((r + 1) & 1, issued + c); g[c] = grow(r + 1, issued + c); }
                poll_attempt<CH>(v, p0, p1, l, g);
                issued += CH;
            } else {
                const uint32_t l0[1] = {0};
                const u32x4* const g0[1] = {nullptr};
                poll_attempt<0>(v, p0, p1, l0, g0);
            }
            const bool ok = v[0].y == tag && v[0].w == tag && v[1].y == tag && v[1].w == tag;
            if (__all(ok)) {
                xs[2 * tid] = v[0].x; xs[2 * tid + 1] = v[0].z;
                xs[2 * (tid + HB_THREADS)] = v[1].x; xs[2 * (tid + HB_THREADS) + 1] = v[1].z;
                break;
            }
            if (tries > HB_SPIN) { a.abort_flag[0] = 1; s_flag = 1; break; }
            __builtin_amdgcn_s_sleep(1);
        }
        lds_barrier();
        if (s_flag) return;
        // ---- "dots": every wave reads the whole vector from LDS, then consumes this round's rows (zeros; the compiler cannot know)
        uint32_t acc = 0;
#pragma unroll
        for (int k = 0; k < NG / 4 / 64; ++k) {
            const u32x4 t = *(const u32x4*)(xs + 4 * (k * 64 + lane));
            acc += t.x + t.y + t.z + t.w;
        }
        if (DB) {
            // rows of round r were requested before this round's polls: in-order return means they have landed (the polls did)
#pragma unroll
            for (int j = 0; j < NL; ++j) {
                const u32x4 t = ring[(((r & 1) * 8 + wave) * NL + j) * 64 + lane];
                z |= t.x | t.y | t.z | t.w;
            }
        } else {
#pragma unroll
            for (int j = 0; j < NL; ++j) z |= wv[j].x | wv[j].y | wv[j].z | wv[j].w;
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) acc += __shfl_xor(acc, o);
        S = acc;
        if (r + 1 < a.rounds) publish(r + 1);   // publish BEFORE the remaining requests
        __builtin_amdgcn_sched_barrier(0);
        if (r + 1 < a.rounds) {
            if (DB) issue_lds(r + 1, issued); else issue_regs(r + 1);
        }
        __builtin_amdgcn_sched_barrier(0);
        lds_barrier();  // everyone has read xs before the next round overwrites it
    }
    if (tid == 0) a.out[b] = S + z;
}

static uint32_t host_expect(int rounds) {
    uint32_t S = 0x1234567u;
    for (int r = 0; r < rounds; ++r) {
        uint32_t acc = 0;
        for (int idx = 0; idx < 2048; ++idx) acc += mix(S + (uint32_t)idx + (uint32_t)r + 1u);
        S = acc;
    }
    return S;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)

template <bool DB, int NL, int CH>
static void run(Args a, uint32_t expect) {
    const size_t lds = 2048 * 4 + (DB ? (size_t)2 * 8 * NL * 1024 : 0);
    CK(hipFuncSetAttribute((const void*)db_kernel<DB, NL, CH>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float ms = 0;
    for (int pass = 0; pass < 2; ++pass) {  // pass 0 warms up (tags restart at 1: clear the granules in between)
        CK(hipMemset(a.gran, 0, sizeof(u64) * 2 * 2048));
        CK(hipMemset(a.abort_flag, 0, 4));
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((db_kernel<DB, NL, CH>), dim3(HB_WGS), dim3(HB_THREADS), lds, 0, a);
        CK(hipGetLastError());
        CK(hipEventRecord(e1));
        CK(hipDeviceSynchronize());
        CK(hipEventElapsedTime(&ms, e0, e1));
    }
    std::vector<uint32_t> out(HB_WGS);
    uint32_t ab = 0;
    CK(hipMemcpy(out.data(), a.out, 4 * HB_WGS, hipMemcpyDeviceToHost));
    CK(hipMemcpy(&ab, a.abort_flag, 4, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int i = 0; i < HB_WGS; ++i) bad += out[i] != expect;
    const double bytes = (double)NL * 1024 * 8 * HB_WGS, us = ms * 1e3 / a.rounds;
    printf("{\"mode\": \"%s\", \"rows_per_wave\": %d, \"rows_per_poll_attempt\": %d, \"us_per_round\": %.3f, \"stream_MB_per_round\": %.2f, "
           "\"stream_TBps\": %.3f, \"abort\": %u, \"wrong_wgs\": %d}\n",
           DB ? "db_lds" : "reg", NL, CH, us, bytes / 1e6, bytes / (us * 1e-6) / 1e12, ab, bad);
    fflush(stdout);
}

int main(int argc, char** argv) {
    const int rounds = argc > 1 ? atoi(argv[1]) : 2000;
    Args a{};
    a.region = (size_t)24 * HB_WGS * HB_THREADS;  // u32x4 per round: 50 MB
    a.nregions = 16;                              // 800 MB cycle: never L2 / MALL resident
    a.rounds = rounds;
    CK(hipMalloc(&a.w, a.region * a.nregions * sizeof(u32x4)));
    CK(hipMemset((void*)a.w, 0, a.region * a.nregions * sizeof(u32x4)));
    CK(hipMalloc(&a.gran, sizeof(u64) * 2 * 2048));
    CK(hipMalloc(&a.out, 4 * HB_WGS));
    CK(hipMalloc(&a.abort_flag, 4));
    const uint32_t e = host_expect(rounds);
    run<false, 0, 0>(a, e);
    run<false, 4, 0>(a, e);
    run<false, 8, 0>(a, e);
    run<true, 4, 0>(a, e);   // rows of the next round requested only after the publish (one round of prefetch distance, no metering)
    run<true, 4, 1>(a, e);
    run<true, 4, 2>(a, e);
    run<true, 4, 4>(a, e);
    run<true, 8, 0>(a, e);
    run<true, 8, 1>(a, e);
    run<true, 8, 2>(a, e);
    run<true, 8, 4>(a, e);
    run<true, 8, 8>(a, e);
    return 0;
}
