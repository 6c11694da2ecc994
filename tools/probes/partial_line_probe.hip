// Lost updates under concurrent partial-line write-through stores from different XCDs?  The tag-free row hand-off of decode_mfma.hip has 8 workgroups
// (8 XCDs) publish 16-byte pieces of ONE 128-byte line with 4-byte agent-scope stores, every workgroup reads every line with agent-scope loads, and
// every piece is put back to a sentinel by its owner before the next round.  This probe runs that cycle with nothing else on the chip:
// per round every block (a) polls the whole region until every word carries the round's value, (b) grid barrier, (c) resets its own words to the
// sentinel, drains, (d) grid barrier, (e) publishes the next value into its own words.  A poll that never completes = a lost update (or a stale read).
//   hipcc --offload-arch=gfx950 -O2 tools/probes/partial_line_probe.hip -o partial_line_probe; ./partial_line_probe [blocks] [rounds] [mode]
//   mode 0: no barrier between (c) and (e) beyond the drain (the kernel's situation: reset, then much later publish)   1: publish straight after the poll (no reset at all)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
#define WORDS 4096          // 16 KB region = 128 lines
#define SENT 0xFFFFFFFFu

__device__ void grid_barrier(unsigned* ctr, unsigned nblocks, unsigned& phase) {
    __syncthreads();
    if (threadIdx.x == 0) {
        ++phase;
        __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        while (__hip_atomic_fetch_add(ctr, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < phase * nblocks) __builtin_amdgcn_s_sleep(2);
    }
    __syncthreads();
}
__device__ unsigned ld_sc1(const unsigned* p) {
    unsigned v;
    asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(p) : "memory");
    return v;
}
__device__ void st_sc1(unsigned* p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// owner of word w: 16-byte pieces dealt round-robin to the blocks
__device__ int owner(int w, int nblocks) { return (w >> 2) % nblocks; }

__global__ void cycle(unsigned* data, unsigned* ctr, int* out, int rounds, int mode) {
    const int nb = gridDim.x, b = blockIdx.x, t = threadIdx.x;
    unsigned phase = 0;
    __shared__ int s_bad;
    for (int k = 0; k < rounds; ++k) {
        const unsigned want = 0x100u + (unsigned)k;
        // (a) poll everything
        int tries = 0;
        for (;; ++tries) {
            if (t == 0) s_bad = 0;
            __syncthreads();
            int bad = 0;
            for (int w = t; w < WORDS; w += blockDim.x) bad += ld_sc1(data + w) != want;
            if (bad) atomicAdd(&s_bad, bad);
            __syncthreads();
            const int tot = s_bad;
            __syncthreads();
            if (tot == 0) break;
            if (tries > 200000) {   // stuck: record the round, the block and the first bad word + its content + its owner
                if (t == 0) {
                    int first = -1; unsigned val = 0;
                    for (int w = 0; w < WORDS && first < 0; ++w) { const unsigned v = ld_sc1(data + w); if (v != want) { first = w; val = v; } }
                    if (atomicCAS(out, 0, 1) == 0) { out[1] = k; out[2] = b; out[3] = first; out[4] = (int)val; out[5] = owner(first, nb); out[6] = tot; }
                }
                return;   // (the others run into their own time-out)
            }
            __builtin_amdgcn_s_sleep(4);
        }
        if (t == 0 && tries > out[8 + b]) out[8 + b] = tries;
        if (mode == 0) {
            grid_barrier(ctr, nb, phase);   // every reader is done
            for (int w = t; w < WORDS; w += blockDim.x) if (owner(w, nb) == b) st_sc1(data + w, SENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            grid_barrier(ctr, nb, phase);   // every reset has drained
        } else grid_barrier(ctr, nb, phase);
        for (int w = t; w < WORDS; w += blockDim.x) if (owner(w, nb) == b) st_sc1(data + w, want + 1u);
    }
    if (t == 0 && b == 0) out[7] = rounds;
}

int main(int argc, char** argv) {
    const int blocks = argc > 1 ? atoi(argv[1]) : 8, rounds = argc > 2 ? atoi(argv[2]) : 20000, mode = argc > 3 ? atoi(argv[3]) : 0;
    unsigned *data, *ctr; int* out;
    CHECK(hipMalloc((void**)&data, WORDS * 4)); CHECK(hipMalloc((void**)&ctr, 256)); CHECK(hipMalloc((void**)&out, 4096));
    unsigned* init = (unsigned*)malloc(WORDS * 4);
    for (int i = 0; i < WORDS; ++i) init[i] = 0x100u;
    CHECK(hipMemcpy(data, init, WORDS * 4, hipMemcpyHostToDevice));
    CHECK(hipMemset(ctr, 0, 256)); CHECK(hipMemset(out, 0, 4096));
    hipLaunchKernelGGL(cycle, dim3(blocks), dim3(256), 0, 0, data, ctr, out, rounds, mode);
    CHECK(hipDeviceSynchronize());
    int h[8 + 256];
    CHECK(hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost));
    int worst = 0;
    for (int i = 0; i < blocks; ++i) worst = h[8 + i] > worst ? h[8 + i] : worst;
    if (h[0]) printf("%d blocks, mode %d: STUCK in round %d: block %d never saw word %d (owner: block %d) change -- it reads 0x%x; %d words wrong\n", blocks, mode, h[1], h[2], h[3], h[5], (unsigned)h[4], h[6]);
    else printf("%d blocks, mode %d: %d rounds completed, no poll stuck; worst poll needed %d repeats\n", blocks, mode, h[7], worst);
    return 0;
}
