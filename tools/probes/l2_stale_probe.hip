// Does an agent-scope (sc1) load see another XCD's write-through store when its OWN XCD's L2 already holds the line?
// Consumer (block 0, XCD 0) reads a line (value A), tells the producer (block 1, XCD 1), the producer stores B (sc1, write-through) and drains,
// the consumer re-reads with one of several load flavours and counts the attempts until it sees B -- with NO other traffic on the chip
// (the situation of a hand-off retry in a persistent kernel whose weights have all landed).  Signals travel as returning device-scope atomics.
//   hipcc --offload-arch=gfx950 -O2 tools/probes/l2_stale_probe.hip -o /tmp/l2_stale_probe && /tmp/l2_stale_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ unsigned sig_get(unsigned* p) { return __hip_atomic_fetch_add(p, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ void sig_set(unsigned* p, unsigned v) { __hip_atomic_exchange(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

template <int FLAVOUR>
__device__ unsigned load_word(const unsigned* p, unsigned* lds_slot) {
    unsigned v;
    if constexpr (FLAVOUR == 0) asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(p) : "memory");
    else if constexpr (FLAVOUR == 1) asm volatile("global_load_dword %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(p) : "memory");
    else if constexpr (FLAVOUR == 2) asm volatile("global_load_dword %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(p) : "memory");
    else if constexpr (FLAVOUR == 3) asm volatile("buffer_inv sc1\n\tglobal_load_dword %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(p) : "memory");
    else if constexpr (FLAVOUR == 4 || FLAVOUR == 5) {   // LDS-DMA, 16 bytes per lane; lane 0's first word is the probe
        const unsigned lds = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)lds_slot;
        unsigned m0s;
        if constexpr (FLAVOUR == 4)
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, off sc1\n\ts_mov_b32 m0, %0\n\ts_waitcnt vmcnt(0)"
                         : "=&s"(m0s) : "s"(__builtin_amdgcn_readfirstlane(lds)), "v"(p + (threadIdx.x & 63) * 4) : "memory");
        else
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, off sc0 sc1\n\ts_mov_b32 m0, %0\n\ts_waitcnt vmcnt(0)"
                         : "=&s"(m0s) : "s"(__builtin_amdgcn_readfirstlane(lds)), "v"(p + (threadIdx.x & 63) * 4) : "memory");
        v = *(volatile unsigned*)lds_slot;
    } else asm volatile("global_load_dword %0, %1, off nt sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(p) : "memory");
    return v;
}

// data: 64 lanes x 16 B = 1 KB region (8 lines); sig[0]: consumer has read A, sig[32]: producer has stored B; out[0] = attempts until B (or -1), out[1] = first value seen
template <int FIRST, int RETRY>
__global__ void probe(unsigned* data, unsigned* sig, int* out, unsigned A, unsigned B, int max_tries) {
    __shared__ __attribute__((aligned(16))) unsigned slot[256];
    const int lane = threadIdx.x;
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    if (blockIdx.x == 0) {
        const unsigned v0 = load_word<FIRST>(data, slot);   // the line is now in this XCD's L2 (if this flavour allocates)
        if (lane == 0) { out[1] = (int)v0; out[2] = (int)(xcc & 15); sig_set(sig, 1u); }
        if (lane == 0) while (sig_get(sig + 32) == 0u) __builtin_amdgcn_s_sleep(4);
        __syncthreads();
        int seen = -1;
        for (int i = 0; i < max_tries; ++i) {
            const unsigned v = load_word<RETRY>(data, slot);
            if (__builtin_amdgcn_readfirstlane(v) == B) { seen = i; break; }
            __builtin_amdgcn_s_sleep(8);
        }
        if (lane == 0) out[0] = seen;
    } else if (blockIdx.x == 1) {
        if (lane == 0) { while (sig_get(sig) == 0u) __builtin_amdgcn_s_sleep(4); }
        __syncthreads();
        // every lane rewrites its 16 bytes (the whole 1 KB region): write-through, then drain
        const u32x4 w = {B, B, B, B};
        asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" ::"v"(data + lane * 4), "v"(w) : "memory");
        __syncthreads();
        if (lane == 0) { out[3] = (int)(xcc & 15); sig_set(sig + 32, 1u); }
    }
}

template <int FIRST, int RETRY>
void run(const char* name, unsigned* data, unsigned* sig, int* out) {
    int worst = 0, never = 0, first_bad = 0, xc = -1, xp = -1;
    for (int rep = 0; rep < 20; ++rep) {
        const unsigned A = 0x1000u + rep, B = 0x2000u + rep;
        unsigned init[256];
        for (auto& x : init) x = A;
        CHECK(hipMemcpy(data, init, sizeof(init), hipMemcpyHostToDevice));
        CHECK(hipMemset(sig, 0, 64 * 4));
        CHECK(hipMemset(out, 0xFF, 16));
        hipLaunchKernelGGL((probe<FIRST, RETRY>), dim3(2), dim3(64), 0, 0, data, sig, out, A, B, 20000);
        CHECK(hipDeviceSynchronize());
        int h[4];
        CHECK(hipMemcpy(h, out, 16, hipMemcpyDeviceToHost));
        if ((unsigned)h[1] != A) ++first_bad;
        if (h[0] < 0) ++never; else if (h[0] > worst) worst = h[0];
        xc = h[2]; xp = h[3];
    }
    printf("%-64s consumer XCC %d producer XCC %d: never saw the new value in %2d / 20 runs (20000 attempts each), worst successful run %d attempts%s\n", name, xc, xp, never, worst,
           first_bad ? "  [first read did not return A!]" : "");
}

int main() {
    unsigned *data, *sig; int* out;
    CHECK(hipMalloc((void**)&data, 4096)); CHECK(hipMalloc((void**)&sig, 4096)); CHECK(hipMalloc((void**)&out, 64));
    run<0, 0>("first read sc1, retries sc1", data, sig, out);
    run<2, 0>("first read plain, retries sc1", data, sig, out);
    run<0, 1>("first read sc1, retries sc0 sc1", data, sig, out);
    run<2, 2>("first read plain, retries plain (expected stale: L1)", data, sig, out);
    run<2, 3>("first read plain, retries buffer_inv sc1 + plain", data, sig, out);
    run<4, 4>("first read LDS-DMA sc1, retries LDS-DMA sc1", data, sig, out);
    run<4, 5>("first read LDS-DMA sc1, retries LDS-DMA sc0 sc1", data, sig, out);
    run<4, 0>("first read LDS-DMA sc1, retries sc1 (registers)", data, sig, out);
    run<6, 6>("first read nt sc1, retries nt sc1", data, sig, out);
    return 0;
}
