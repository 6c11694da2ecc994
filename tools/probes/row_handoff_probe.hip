// The tag-free row hand-off of decode_mfma.hip in isolation: 256 workgroups (one per CU), a vector of 16 rows x 2048 bf16 columns, workgroup b
// owns columns 8 b .. 8 b + 7 of every row (16 bytes per row: 4-byte agent-scope stores by 64 threads, as the kernel's reduce threads do), every
// workgroup pulls all rows with agent-scope LDS-DMA loads and looks for the sentinel 0xFFFFFFFF.  Two buffers alternate by round; a slot is put
// back to the sentinel one round after it was read (by then every reader has published the next round's vector behind it).  Workgroups are
// deliberately SKEWED (a pseudo-random delay in front of the publish) so that first looks are often early and the retry path runs.
//   row_handoff_probe [rounds] [max_delay] [retry_flavour: 0 = sc1 LDS-DMA, 1 = sc0 sc1 LDS-DMA, 2 = atomic reads, 3 = sc1 register loads]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
#define ROWS 16
#define COLS 2048
#define SENT 0xFFFFFFFFu
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int SC>
__device__ void row_to_lds(unsigned lds, const void* g) {
    unsigned m0s;
    if (SC == 1) asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, off sc1\n\ts_mov_b32 m0, %0" : "=&s"(m0s) : "s"(lds), "v"(g) : "memory");
    else asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, off sc0 sc1\n\ts_mov_b32 m0, %0" : "=&s"(m0s) : "s"(lds), "v"(g) : "memory");
}

__global__ void __launch_bounds__(512, 1) cycle(unsigned short* buf0, unsigned short* buf1, int* out, int rounds, int max_delay, int flavour) {
    __shared__ __attribute__((aligned(16))) unsigned short xs[ROWS * COLS];   // 64 KB
    __shared__ int s_stop;
    const int b = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const unsigned xs_lds = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)xs;
    if (t == 0) s_stop = 0;
    __syncthreads();
    for (int k = 0; k < rounds; ++k) {
        unsigned short* cur = (k & 1) ? buf1 : buf0;
        unsigned short* prev = (k & 1) ? buf0 : buf1;
        // publish round k (value depends on round, row, column), after a skew
        unsigned h = (unsigned)(k * 2654435761u) ^ (unsigned)(b * 40503u);
        h ^= h >> 13; h *= 0x5bd1e995u; h ^= h >> 15;
        if (max_delay) for (int d = (int)(h % (unsigned)max_delay); d > 0; --d) __builtin_amdgcn_s_sleep(8);
        if (t < 64) {   // thread = (pair j, row r): columns 8 b + 2 j, + 1 of row r
            const int r = t % ROWS, j = t / ROWS;
            const unsigned lo = (unsigned)((k + r + 8 * b + 2 * j) & 0x7FFF), hi = (unsigned)((k + r + 8 * b + 2 * j + 1) & 0x7FFF);
            __hip_atomic_store((unsigned*)(cur + (size_t)r * COLS + 8 * b + 2 * j), lo | (hi << 16), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        // put round k - 1's slots back (every reader of round k - 1 published round k behind its read... of which this block has seen nothing yet:
        // so, as in the kernel, the reset of round k - 1 goes out only after THIS round's poll; here: reset round k - 1 after polling round k)
        // poll round k: wave w takes requests w, w + 8, ...: request q = (row q / 4, segment q % 4), 8 per wave
        unsigned pending = 0xFFu;
        int tries = 0;
        for (;; ++tries) {
            for (int i = 0; i < 8; ++i) {
                const int q = wave + 8 * i;
                if (pending >> i & 1u) {
                    const int r = q / 4, sg = q % 4;
                    const unsigned short* g = cur + (size_t)r * COLS + 512 * sg + lane * 8;
                    unsigned short* l = xs + (size_t)r * COLS + 512 * sg + lane * 8;
                    if (tries == 0 || flavour == 0) row_to_lds<1>(xs_lds + (unsigned)(r * COLS + 512 * sg) * 2u, g);
                    else if (flavour == 1) row_to_lds<2>(xs_lds + (unsigned)(r * COLS + 512 * sg) * 2u, g);
                    else if (flavour == 2) {
                        unsigned* gw = (unsigned*)g;
                        u32x4 v;
                        v.x = __hip_atomic_fetch_or(gw + 0, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        v.y = __hip_atomic_fetch_or(gw + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        v.z = __hip_atomic_fetch_or(gw + 2, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        v.w = __hip_atomic_fetch_or(gw + 3, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        *(u32x4*)l = v;
                    } else {
                        u32x4 v;
                        asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(g) : "memory");
                        *(u32x4*)l = v;
                    }
                }
            }
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            for (int i = 0; i < 8; ++i) {
                const int q = wave + 8 * i;
                if (pending >> i & 1u) {
                    const int r = q / 4, sg = q % 4;
                    const u32x4 v = *(const u32x4*)(xs + (size_t)r * COLS + 512 * sg + lane * 8);
                    const bool bad = v.x == SENT || v.y == SENT || v.z == SENT || v.w == SENT;
                    if (!__any(bad)) pending &= ~(1u << i);
                }
            }
            pending = __builtin_amdgcn_readfirstlane(pending);
            if (pending == 0u) break;
            if (tries > 100000 || *(volatile int*)&s_stop) {
                if (lane == 0 && atomicCAS(out, 0, 1) == 0) { out[1] = k; out[2] = b; out[3] = wave; out[4] = (int)pending; out[5] = tries; }
                *(volatile int*)&s_stop = 1;
                break;
            }
            __builtin_amdgcn_s_sleep(1);
        }
        if (lane == 0 && tries > 0) atomicAdd(out + 8, 1);          // polls that needed a repeat
        if (lane == 0 && tries > out[9]) atomicMax(out + 9, tries);
        __syncthreads();
        if (s_stop || __hip_atomic_load(out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;
        // verify what landed (every word of every row)
        for (int i = t; i < ROWS * COLS / 2; i += 512) {
            const int r = i / (COLS / 2), c = 2 * (i % (COLS / 2));
            const unsigned want = (unsigned)((k + r + c) & 0x7FFF) | ((unsigned)((k + r + c + 1) & 0x7FFF) << 16);
            if (((const unsigned*)xs)[i] != want && atomicCAS(out + 16, 0, 1) == 0) { out[17] = k; out[18] = b; out[19] = r; out[20] = c; out[21] = (int)((const unsigned*)xs)[i]; out[22] = (int)want; }
        }
        __syncthreads();
        // reset the previous round's slots of this block (its readers all published round k, which this block has just seen complete)
        if (k > 0 && t < 64) {
            const int r = t % ROWS, j = t / ROWS;
            __hip_atomic_store((unsigned*)(prev + (size_t)r * COLS + 8 * b + 2 * j), SENT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (the kernel's next poll does this)
    }
    if (t == 0 && b == 0) out[7] = rounds;
}

int main(int argc, char** argv) {
    const int rounds = argc > 1 ? atoi(argv[1]) : 100000, max_delay = argc > 2 ? atoi(argv[2]) : 8, flavour = argc > 3 ? atoi(argv[3]) : 0;
    unsigned short *b0, *b1; int* out;
    CHECK(hipMalloc((void**)&b0, ROWS * COLS * 2)); CHECK(hipMalloc((void**)&b1, ROWS * COLS * 2)); CHECK(hipMalloc((void**)&out, 4096));
    CHECK(hipMemset(b0, 0xFF, ROWS * COLS * 2)); CHECK(hipMemset(b1, 0xFF, ROWS * COLS * 2)); CHECK(hipMemset(out, 0, 4096));
    hipLaunchKernelGGL(cycle, dim3(256), dim3(512), 0, 0, b0, b1, out, rounds, max_delay, flavour);
    CHECK(hipDeviceSynchronize());
    int h[32];
    CHECK(hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost));
    printf("retry flavour %d, skew up to %d: ", flavour, max_delay);
    if (h[0]) printf("STUCK in round %d: block %d wave %d pending mask %x after %d attempts; ", h[1], h[2], h[3], (unsigned)h[4], h[5]);
    else printf("%d rounds completed; ", h[7]);
    printf("%d wave-polls needed a repeat (worst %d repeats)", h[8], h[9]);
    if (h[16]) printf("; WRONG DATA in round %d block %d row %d column %d: %08x instead of %08x", h[17], h[18], h[19], h[20], (unsigned)h[21], (unsigned)h[22]);
    printf("\n");
    return 0;
}
