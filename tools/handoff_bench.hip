// Micro-benchmark behind the persistent decode kernel's hand-off design (profiles/r01g_handoff/NOTES.md).
//
// 256 resident workgroups x 512 threads run `rounds` all-to-all exchanges of one activation vector (NG 8-byte granules
// {payload, tag}, cdna_hip_programming.md Guideline 16) while every wave keeps `nl` 16-byte weight loads per lane in
// flight, exactly the situation of decode_mega.hip: weights of the NEXT phase are requested before the CURRENT phase's
// input is polled.  What is varied is WHO polls and THROUGH WHICH PATH:
//   mode 0: every wave polls its share with vector agent-scope loads (decode_mega.hip today).  Vector loads of a wave
//           return in order, so a poll queues behind that wave's own weight prefetch.
//   mode 1: NPW dedicated poller waves (no weight loads of their own) poll everything with vector loads.
//   mode 2: every wave polls its share through the SCALAR memory path (s_load_dwordx16 glc): separate counter (lgkmcnt),
//           separate queue from the vector memory pipeline.
// Output: microseconds per round for each (mode, NG, nl), plus a host-checked checksum (a stale payload changes it).
//
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -o tools/handoff_bench.bin tools/handoff_bench.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef unsigned long long u64;
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x16 __attribute__((ext_vector_type(16)));

#define HB_THREADS 512
#define HB_WGS 256
#define HB_SPIN (1 << 20)

struct Args {
    u64* gran;            // [2][NG]
    const u32x4* w;       // weight stream (zero filled)
    size_t region;        // u32x4 elements per round region
    int nregions, rounds, nl;
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
__device__ __forceinline__ u64 gran_load(const u64* g) { return __hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__device__ __forceinline__ uint32_t writelane(uint32_t val, int l, uint32_t old) {
    asm volatile("v_writelane_b32 %0, %1, %2" : "+v"(old) : "s"(val), "i"(l));
    return old;
}

template <int MODE, int NPW, int NG, int NL>
__global__ void __launch_bounds__(HB_THREADS, 2) hb_kernel(const Args a) {
    __shared__ __attribute__((aligned(16))) uint32_t xs[NG];
    __shared__ int s_flag;
    const int tid0 = threadIdx.x, b = blockIdx.x;
    int lane = tid0 & 63, wave = tid0 >> 6, tid = tid0;
    const bool poller = MODE == 1 ? wave < NPW : true;
    const bool streamer = MODE == 1 ? wave >= NPW : true;
    constexpr int PER_WG = NG / HB_WGS;  // 8 or 22 granules published per workgroup
    u32x4 wv[NL > 0 ? NL : 1];
#pragma unroll
    for (int j = 0; j < NL; ++j) wv[j] = u32x4{0, 0, 0, 0};
    auto issue = [&](int r) {
        if (!streamer) return;
        const u32x4* base = a.w + (size_t)(r % a.nregions) * a.region + (size_t)b * HB_THREADS + tid;
#pragma unroll
        for (int j = 0; j < NL; ++j) wv[j] = __builtin_nontemporal_load(base + (size_t)j * HB_WGS * HB_THREADS);
    };
    uint32_t S = 0x1234567u, z = 0;
    // publish this workgroup's granules of round r (needs S of the previous round: a true dependency chain)
    auto publish = [&](int r) {
        if (tid < PER_WG) {
            const int idx = b * PER_WG + tid;
            gran_store(a.gran + (size_t)(r & 1) * NG + idx, mix(S + (uint32_t)idx + (uint32_t)r + 1u + z), (uint32_t)r + 1u);
        }
    };
    constexpr bool PUBLISH_FIRST = MODE != 1 && (NPW & 1), NO_SLEEP = MODE != 1 && (NPW & 2);
    issue(0);
    publish(0);
    if (tid == 0) s_flag = 0;
    __syncthreads();

    for (int r = 0; r < a.rounds; ++r) {
        // opaque ids: keeps the address arithmetic next to its use instead of hoisted (and spilled) out of the loop
        asm volatile("" : "+v"(lane), "+v"(wave));
        tid = wave * 64 + lane;
        const uint32_t tag = (uint32_t)r + 1u;
        u64* buf = a.gran + (size_t)(r & 1) * NG;
        // ---- poll every granule of the round into LDS
        if (MODE == 0 || MODE == 3) {
            constexpr int PER = (NG + HB_THREADS - 1) / HB_THREADS;
            if (MODE == 3) {  // spin on ONE granule per producing workgroup first (2 KB instead of 16-45 KB per try)
                for (int tries = 0;; ++tries) {
                    bool ok = true;
                    if (tid < HB_WGS) ok = (uint32_t)(gran_load(buf + tid * PER_WG + PER_WG - 1) >> 32) == tag;
                    if (__syncthreads_and(ok)) break;
                    if (tries > HB_SPIN) { if (tid == 0) a.abort_flag[0] = 1; return; }
                    if (!NO_SLEEP) __builtin_amdgcn_s_sleep(1);
                }
            }
            for (int tries = 0;; ++tries) {
                u64 v[PER];
                bool ok = true;
#pragma unroll
                for (int k = 0; k < PER; ++k) {
                    const int n = tid + k * HB_THREADS;
                    if (n < NG) v[k] = gran_load(buf + n);
                }
#pragma unroll
                for (int k = 0; k < PER; ++k) {
                    const int n = tid + k * HB_THREADS;
                    if (n < NG) ok = ok && (uint32_t)(v[k] >> 32) == tag;
                }
                if (ok) {
#pragma unroll
                    for (int k = 0; k < PER; ++k) {
                        const int n = tid + k * HB_THREADS;
                        if (n < NG) xs[n] = (uint32_t)v[k];
                    }
                }
                if (__syncthreads_and(ok)) break;
                if (tries > HB_SPIN) { if (tid == 0) a.abort_flag[0] = 1; return; }
                if (!NO_SLEEP) __builtin_amdgcn_s_sleep(1);
            }
        } else if (MODE == 6) {
            // every wave spins on its own granules, no barrier inside the retry loop (decode_mega.hip since r01i); NPW & 4: keep
            // TWO staggered polls in flight so that a miss costs half a round trip instead of a whole one
            constexpr int PER = (NG + HB_THREADS - 1) / HB_THREADS;
            constexpr bool TWO = (NPW & 4) != 0;
            auto fetch = [&](u64 (&v)[PER]) {
#pragma unroll
                for (int k = 0; k < PER; ++k) {
                    const int n = tid + k * HB_THREADS;
                    v[k] = n < NG ? gran_load(buf + n) : ((u64)tag << 32);
                }
            };
            auto good = [&](const u64 (&v)[PER]) {
                bool ok = true;
#pragma unroll
                for (int k = 0; k < PER; ++k) ok = ok && (uint32_t)(v[k] >> 32) == tag;
                return __all(ok) != 0;
            };
            auto sink = [&](const u64 (&v)[PER]) {
#pragma unroll
                for (int k = 0; k < PER; ++k) {
                    const int n = tid + k * HB_THREADS;
                    if (n < NG) xs[n] = (uint32_t)v[k];
                }
            };
            u64 va[PER], vb[PER];
            fetch(va);
            for (int tries = 0;; ++tries) {
                if (TWO) { __builtin_amdgcn_s_sleep(6); fetch(vb); }
                if (good(va)) { sink(va); break; }
                if (TWO) {
                    __builtin_amdgcn_s_sleep(6);
                    fetch(va);
                    if (good(vb)) { sink(vb); break; }
                } else {
                    __builtin_amdgcn_s_sleep(1);
                    fetch(va);
                }
                if (tries > HB_SPIN) { a.abort_flag[0] = 1; s_flag = 1; break; }
            }
            lds_barrier();
            if (s_flag) return;
        } else if (MODE == 7) {
            // per-wave spin with 16-byte agent-scope loads: two granules per lane per request (each 8-byte half carries its own tag)
            constexpr int PAIRS = NG / 2, PER = (PAIRS + HB_THREADS - 1) / HB_THREADS;
            for (int tries = 0;; ++tries) {
                u32x4 v[PER];
#pragma unroll
                for (int k = 0; k < PER; ++k) {
                    const int n = min(tid + k * HB_THREADS, PAIRS - 1);
                    asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v[k]) : "v"(buf + 2 * n) : "memory");
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                bool ok = true;
#pragma unroll
                for (int k = 0; k < PER; ++k) ok = ok && v[k].y == tag && v[k].w == tag;
                if (__all(ok)) {
#pragma unroll
                    for (int k = 0; k < PER; ++k) {
                        const int n = tid + k * HB_THREADS;
                        if (n < PAIRS) { xs[2 * n] = v[k].x; xs[2 * n + 1] = v[k].z; }
                    }
                    break;
                }
                if (tries > HB_SPIN) { a.abort_flag[0] = 1; s_flag = 1; break; }
                __builtin_amdgcn_s_sleep(1);
            }
            lds_barrier();
            if (s_flag) return;
        } else if (MODE == 1) {
            if (poller) {
                constexpr int SHARE = NG / (NPW > 0 ? NPW : 1);  // granules per poller wave
                constexpr int PER = (SHARE + 63) / 64;  // loads per lane
                constexpr int BATCH = 32;
                const u64* g = buf + wave * SHARE;
                for (int k0 = 0; k0 < PER; k0 += BATCH) {
                    for (int tries = 0;; ++tries) {
                        u64 v[BATCH];
                        bool ok = true;
#pragma unroll
                        for (int k = 0; k < BATCH; ++k) {
                            const int n = (k0 + k) * 64 + lane;
                            if (k0 + k < PER && n < SHARE) v[k] = gran_load(g + n);
                        }
#pragma unroll
                        for (int k = 0; k < BATCH; ++k) {
                            const int n = (k0 + k) * 64 + lane;
                            if (k0 + k < PER && n < SHARE) ok = ok && (uint32_t)(v[k] >> 32) == tag;
                        }
                        if (__all(ok)) {
#pragma unroll
                            for (int k = 0; k < BATCH; ++k) {
                                const int n = (k0 + k) * 64 + lane;
                                if (k0 + k < PER && n < SHARE) xs[wave * SHARE + n] = (uint32_t)v[k];
                            }
                            break;
                        }
                        if (tries > HB_SPIN) { a.abort_flag[0] = 1; s_flag = 1; break; }
                        __builtin_amdgcn_s_sleep(1);
                    }
                }
            }
            lds_barrier();
            if (s_flag) return;
        } else {
            // scalar path: wave `wave` polls granules [wave * SHARE, + SHARE) in batches of 32 granules = 4 x 64 bytes
            constexpr int SHARE = NG / 8;
            const u64* g0 = buf + wave * SHARE;
            bool dead = false;
            for (int k0 = 0; k0 < SHARE; k0 += 32) {
                const u64 addr = (u64)(g0 + k0);
                const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)addr), hi = __builtin_amdgcn_readfirstlane((uint32_t)(addr >> 32));
                const u64 sp = ((u64)hi << 32) | lo;
                for (int tries = 0;; ++tries) {
                    u32x16 q0, q1, q2, q3;
                    asm volatile(
                        "s_load_dwordx16 %0, %4, 0x0 glc\n\t"
                        "s_load_dwordx16 %1, %4, 0x40 glc\n\t"
                        "s_load_dwordx16 %2, %4, 0x80 glc\n\t"
                        "s_load_dwordx16 %3, %4, 0xc0 glc\n\t"
                        "s_waitcnt lgkmcnt(0)"
                        : "=&s"(q0), "=&s"(q1), "=&s"(q2), "=&s"(q3)
                        : "s"(sp)
                        : "memory");
                    uint32_t bad = 0;
#pragma unroll
                    for (int i = 0; i < 8; ++i) bad |= (q0[2 * i + 1] ^ tag) | (q1[2 * i + 1] ^ tag) | (q2[2 * i + 1] ^ tag) | (q3[2 * i + 1] ^ tag);
                    if (SHARE - k0 < 32) {  // ragged tail (NG = 5632: 704 = 22 x 32 exactly; kept for generality)
                        bad = 0;
#pragma unroll
                        for (int i = 0; i < 8; ++i) {
                            if (k0 + i < SHARE) bad |= q0[2 * i + 1] ^ tag;
                            if (k0 + 8 + i < SHARE) bad |= q1[2 * i + 1] ^ tag;
                            if (k0 + 16 + i < SHARE) bad |= q2[2 * i + 1] ^ tag;
                            if (k0 + 24 + i < SHARE) bad |= q3[2 * i + 1] ^ tag;
                        }
                    }
                    if (bad == 0) {
                        uint32_t pv = 0;
#pragma unroll
                        for (int i = 0; i < 8; ++i) {
                            pv = writelane(q0[2 * i], i, pv);
                            pv = writelane(q1[2 * i], 8 + i, pv);
                            pv = writelane(q2[2 * i], 16 + i, pv);
                            pv = writelane(q3[2 * i], 24 + i, pv);
                        }
                        if (lane < 32 && k0 + lane < SHARE) xs[wave * SHARE + k0 + lane] = pv;
                        break;
                    }
                    if (tries > HB_SPIN) { dead = true; break; }
                    __builtin_amdgcn_s_sleep(1);
                }
                if (dead) break;
            }
            if (dead) { a.abort_flag[0] = 1; s_flag = 1; }
            lds_barrier();
            if (s_flag) return;
        }
        // ---- "dots": every wave reads the whole vector from LDS (16-byte reads, like the GEMV operand reads)
        uint32_t acc = 0;
#pragma unroll
        for (int k = 0; k < (NG / 4 + 63) / 64; ++k) {
            const int i4 = k * 64 + lane;
            if (i4 < NG / 4) {
                const u32x4 t = *(const u32x4*)(xs + 4 * i4);
                acc += t.x + t.y + t.z + t.w;
            }
        }
        // ---- consume the weights of this round (zeros; the compiler cannot know), then request the next round's
        if (streamer) {
#pragma unroll
            for (int j = 0; j < NL; ++j) z |= wv[j].x | wv[j].y | wv[j].z | wv[j].w;
        }
        __builtin_amdgcn_sched_barrier(0);
        if (!PUBLISH_FIRST) issue(r + 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) acc += __shfl_xor(acc, o);
        S = acc;
        if (r + 1 < a.rounds) publish(r + 1);
        __builtin_amdgcn_sched_barrier(0);
        if (PUBLISH_FIRST) issue(r + 1);
        __builtin_amdgcn_sched_barrier(0);
        lds_barrier();  // everyone has read xs before the next round overwrites it
    }
    if (tid == 0) a.out[b] = S + z;
}

// ---- mode 5: dedicated LOADER waves in barrier lock-step with the compute waves ---------------------------------------
// 8 compute waves + 4 loader waves (768 threads).  The loaders stream the weights of round r+1 straight into LDS
// (global_load_lds_dwordx4, no registers) while the compute waves hand off round r; the compute waves never have a weight
// load in flight, so their polls are never queued behind one and they are never blocked by memory back-pressure.
// The weight buffer is double-buffered in LDS: 2 x LPW lines of 1 KB per compute wave.
#define HL_THREADS 768
template <int NG, int LPW>
__global__ void __launch_bounds__(HL_THREADS, 1) hb_loader_kernel(const Args a) {
    __shared__ __attribute__((aligned(16))) uint32_t xs[NG];
    __shared__ __attribute__((aligned(16))) u32x4 ring[2][8 * LPW * 64];
    __shared__ int s_flag;
    const int tid0 = threadIdx.x, b = blockIdx.x;
    int lane = tid0 & 63, wave = tid0 >> 6;
    constexpr int PER_WG = NG / HB_WGS;
    constexpr int LPL = 8 * LPW / 4;  // lines per loader per round
    if (tid0 == 0) s_flag = 0;
    __syncthreads();
    if (wave >= 8) {
        // ================= loader: line j of a round = (compute wave j / LPW, chunk j % LPW); loader q takes j = 4 i + q
        const int q = wave - 8;
        auto issue = [&](int r) {
            const u32x4* base = a.w + (size_t)(r % a.nregions) * a.region + (size_t)b * (8 * LPW * 64);
#pragma unroll
            for (int i = 0; i < LPL; ++i) {
                const int j = 4 * i + q;
                __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(base + (size_t)j * 64 + lane),
                                                 (void __attribute__((address_space(3)))*)(&ring[r & 1][j * 64]), 16, 0, 0);
            }
        };
        issue(0);
        for (int r = 0; r < a.rounds; ++r) {
            asm volatile("" : "+v"(lane));
            if (r + 1 < a.rounds) {
                issue(r + 1);  // ring[(r+1)&1] was released by the closing barrier of round r-1
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPL) : "memory");  // round r's lines have landed in LDS
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            lds_barrier();  // B1
            if (s_flag) return;
            lds_barrier();  // B2
        }
        return;
    }
    // ================= compute waves
    int tid = tid0;
    uint32_t S = 0x1234567u, z = 0;
    auto publish = [&](int r) {
        if (tid < PER_WG) {
            const int idx = b * PER_WG + tid;
            gran_store(a.gran + (size_t)(r & 1) * NG + idx, mix(S + (uint32_t)idx + (uint32_t)r + 1u + z), (uint32_t)r + 1u);
        }
    };
    publish(0);
    for (int r = 0; r < a.rounds; ++r) {
        asm volatile("" : "+v"(lane), "+v"(wave));
        tid = wave * 64 + lane;
        const uint32_t tag = (uint32_t)r + 1u;
        const u64* buf = a.gran + (size_t)(r & 1) * NG;
        // every wave spins on its own granules (no barrier inside the retry loop: the loaders only mirror B1 / B2)
        constexpr int PER = (NG + HB_THREADS - 1) / HB_THREADS;
        for (int tries = 0;; ++tries) {
            u64 v[PER];
            bool ok = true;
#pragma unroll
            for (int k = 0; k < PER; ++k) {
                const int n = tid + k * HB_THREADS;
                if (n < NG) v[k] = gran_load(buf + n);
            }
#pragma unroll
            for (int k = 0; k < PER; ++k) {
                const int n = tid + k * HB_THREADS;
                if (n < NG) ok = ok && (uint32_t)(v[k] >> 32) == tag;
            }
            if (__all(ok)) {
#pragma unroll
                for (int k = 0; k < PER; ++k) {
                    const int n = tid + k * HB_THREADS;
                    if (n < NG) xs[n] = (uint32_t)v[k];
                }
                break;
            }
            if (tries > HB_SPIN) { a.abort_flag[0] = 1; s_flag = 1; break; }
            __builtin_amdgcn_s_sleep(1);
        }
        lds_barrier();  // B1: xs complete, this round's weight lines are in LDS
        if (s_flag) return;
        uint32_t acc = 0;
#pragma unroll
        for (int k = 0; k < (NG / 4 + 63) / 64; ++k) {
            const int i4 = k * 64 + lane;
            if (i4 < NG / 4) {
                const u32x4 t = *(const u32x4*)(xs + 4 * i4);
                acc += t.x + t.y + t.z + t.w;
            }
        }
#pragma unroll
        for (int c = 0; c < LPW; ++c) {
            const u32x4 t = ring[r & 1][(wave * LPW + c) * 64 + lane];
            z |= t.x | t.y | t.z | t.w;
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) acc += __shfl_xor(acc, o);
        S = acc;
        if (r + 1 < a.rounds) publish(r + 1);
        lds_barrier();  // B2: xs and ring[r&1] may be overwritten
    }
    if (tid == 0) a.out[b] = S + z;
}

template <int NG>
static uint32_t host_expect(int rounds) {
    uint32_t S = 0x1234567u;
    for (int r = 0; r < rounds; ++r) {
        uint32_t acc = 0;
        for (int idx = 0; idx < NG; ++idx) acc += mix(S + (uint32_t)idx + (uint32_t)r + 1u);
        S = acc;
    }
    return S;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)

template <int MODE, int NPW, int NG, int NL>
static void run(const char* name, Args a, uint32_t expect) {
    const int nl = NL;
    a.nl = nl;
    CK(hipMemset(a.gran, 0, sizeof(u64) * 2 * NG));
    CK(hipMemset(a.abort_flag, 0, 4));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    // warm-up launch (tags restart at 1: clear the granules in between)
    hipLaunchKernelGGL((hb_kernel<MODE, NPW, NG, NL>), dim3(HB_WGS), dim3(HB_THREADS), 0, 0, a);
    CK(hipDeviceSynchronize());
    CK(hipMemset(a.gran, 0, sizeof(u64) * 2 * NG));
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((hb_kernel<MODE, NPW, NG, NL>), dim3(HB_WGS), dim3(HB_THREADS), 0, 0, a);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<uint32_t> out(HB_WGS);
    uint32_t ab = 0;
    CK(hipMemcpy(out.data(), a.out, 4 * HB_WGS, hipMemcpyDeviceToHost));
    CK(hipMemcpy(&ab, a.abort_flag, 4, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int i = 0; i < HB_WGS; ++i) bad += out[i] != expect;
    const int nstream = MODE == 1 ? 8 - NPW : 8;
    const double bytes = (double)nl * 16 * 64 * nstream * HB_WGS;
    const double us = ms * 1e3 / a.rounds;
    printf("{\"mode\": \"%s\", \"granules\": %d, \"nl\": %d, \"us_per_round\": %.3f, \"stream_MB_per_round\": %.2f, \"stream_TBps\": %.3f, \"abort\": %u, \"wrong_wgs\": %d}\n",
           name, NG, nl, us, bytes / 1e6, bytes / (us * 1e-6) / 1e12, ab, bad);
    fflush(stdout);
}

template <int NG, int LPW>
static void run_loader(Args a, uint32_t expect) {
    a.nl = LPW;
    CK(hipMemset(a.gran, 0, sizeof(u64) * 2 * NG));
    CK(hipMemset(a.abort_flag, 0, 4));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((hb_loader_kernel<NG, LPW>), dim3(HB_WGS), dim3(HL_THREADS), 0, 0, a);
    CK(hipDeviceSynchronize());
    CK(hipMemset(a.gran, 0, sizeof(u64) * 2 * NG));
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((hb_loader_kernel<NG, LPW>), dim3(HB_WGS), dim3(HL_THREADS), 0, 0, a);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<uint32_t> out(HB_WGS);
    uint32_t ab = 0;
    CK(hipMemcpy(out.data(), a.out, 4 * HB_WGS, hipMemcpyDeviceToHost));
    CK(hipMemcpy(&ab, a.abort_flag, 4, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int i = 0; i < HB_WGS; ++i) bad += out[i] != expect;
    const double bytes = (double)LPW * 1024 * 8 * HB_WGS;
    const double us = ms * 1e3 / a.rounds;
    printf("{\"mode\": \"loader_waves_lds\", \"granules\": %d, \"nl\": %d, \"us_per_round\": %.3f, \"stream_MB_per_round\": %.2f, \"stream_TBps\": %.3f, \"abort\": %u, \"wrong_wgs\": %d}\n",
           NG, LPW, us, bytes / 1e6, bytes / (us * 1e-6) / 1e12, ab, bad);
    fflush(stdout);
}

int main(int argc, char** argv) {
    const int rounds = argc > 1 ? atoi(argv[1]) : 2000;
    Args a{};
    const size_t region = (size_t)24 * HB_WGS * HB_THREADS;  // u32x4 per round: 50 MB
    a.region = region;
    a.nregions = 16;                                          // 800 MB cycle: never L2 / MALL resident
    a.rounds = rounds;
    CK(hipMalloc(&a.w, region * a.nregions * sizeof(u32x4)));
    CK(hipMemset((void*)a.w, 0, region * a.nregions * sizeof(u32x4)));
    // argv[2] = granule memory kind: 0 plain hipMalloc (default), 1 fine-grained, 3 uncached (hipExtMallocWithFlags)
    const int gran_kind = argc > 2 ? atoi(argv[2]) : 0;
    if (gran_kind == 0) CK(hipMalloc(&a.gran, sizeof(u64) * 2 * 5632));
    else CK(hipExtMallocWithFlags((void**)&a.gran, sizeof(u64) * 2 * 5632, (unsigned)gran_kind));
    printf("{\"granule_memory\": %d}\n", gran_kind);
    CK(hipMalloc(&a.out, 4 * HB_WGS));
    CK(hipMalloc(&a.abort_flag, 4));
    const uint32_t e2048 = host_expect<2048>(rounds), e5632 = host_expect<5632>(rounds);
#define ALL(NL)                                                        \
    run<0, 1, 2048, NL>("vector_all_waves+publish_first", a, e2048);   \
    run<6, 1, 2048, NL>("per_wave_spin+publish_first", a, e2048);      \
    run<7, 1, 2048, NL>("per_wave_spin_16B+publish_first", a, e2048);  \
    run<0, 1, 5632, NL>("vector_all_waves+publish_first", a, e5632);   \
    run<6, 1, 5632, NL>("per_wave_spin+publish_first", a, e5632);      \
    run<7, 1, 5632, NL>("per_wave_spin_16B+publish_first", a, e5632);
    if (argc > 3) {  // the dedicated-poller and scalar-path variants (slow / no gain: profiles/r01g_handoff/NOTES.md)
        run<1, 1, 2048, 12>("vector_1_poller", a, e2048);
        run<1, 2, 2048, 12>("vector_2_pollers", a, e2048);
        run<2, 0, 2048, 12>("scalar_all_waves", a, e2048);
        run<1, 1, 5632, 12>("vector_1_poller", a, e5632);
        run<1, 2, 5632, 12>("vector_2_pollers", a, e5632);
        run<2, 0, 5632, 12>("scalar_all_waves", a, e5632);
    }
    ALL(0) ALL(4) ALL(8) ALL(12) ALL(24)
    run_loader<2048, 4>(a, e2048);
    run_loader<2048, 8>(a, e2048);
    run_loader<5632, 4>(a, e5632);
    run_loader<5632, 6>(a, e5632);
    return 0;
}
