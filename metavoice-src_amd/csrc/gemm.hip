// Many-row (M >= 16) building blocks: MFMA bf16 GEMM with fused epilogues, row normalisation, multi-query attention.
//
// Used by (a) the first-stage PREFILL: `prefill()` runs `Transformer.forward` over all T prompt positions x 2 CFG rows
// (fam/llm/fast_inference_utils.py:123-132, fam/llm/fast_model.py:150-163) -- M = 2T rows, MFMA-bound side of the ridge
// (rows FLOP/B ~ 256), and (b) the second-stage non-causal GPT forward over (b, block_size) positions
// (fam/llm/model.py:195-314, fam/llm/layers/*.py) under bf16 autocast (fam/llm/inference.py:80-82).
//
// GEMM: C[M][N] = A[M][K] . W[N][K]^T, both operands K-contiguous, which is exactly the per-lane fragment shape of
// v_mfma_f32_16x16x32_bf16 (lane l: row l&15, 8 consecutive k at (l>>4)*8), so tiles go global -> registers -> LDS
// (16 B per lane, XOR-swizzled 16-byte chunks: <= 2-way conflicts on ds_read_b128) -> fragments with no transposes.
// 64x64x64 block tile, 4 waves (2x2), each wave 2x2 MFMA tiles; the next K-tile's global loads are issued before the
// current tile's MFMAs (register double buffer).  M = 2T ~ 256 only gives 4 row-tiles, so N/64 column tiles x 4 keep
// >= 128 blocks in flight and W is re-read 4x through L2, not HBM.
#include "common.h"
#include <atomic>

#include "kernels.h"

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));

#define GT 64  // block tile edge (M, N and K)

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }

// The epilogue of one output element (m, n): v = the fp32 dot product x[m] . w[n], v2 = x[m] . w2[n] (gated pairs).
__device__ __forceinline__ void gemm_epilogue(const GemmArgs& a, int m, int n, float v, float v2) {
    const bool gated = a.w2 != nullptr;
    float y = round_bf(v + (a.bias ? bf2f(a.bias[n]) : 0.f));  // the bf16 Linear output
    if (a.scale) y = round_bf(y * bf2f(a.scale[n]));          // int8 weight-only: the codes were the GEMM operand, the row scale follows
    if (gated) {  // bf16(bf16(silu(w1 x)) * bf16(w3 x))   (fast_model.py:237, layers.py:62-72)
        float u = round_bf(v2 + (a.bias2 ? bf2f(a.bias2[n]) : 0.f));
        if (a.scale2) u = round_bf(u * bf2f(a.scale2[n]));
        y = round_bf(round_bf(y / (1.0f + __expf(-y))) * u);
    } else if (a.act == GEMM_ACT_GELU) {
        y = round_bf(gelu_erf(y));
    }
    if (a.qkv_dim > 0) {  // first-stage prefill: q -> out, k/v -> cache[row][head][t][128]  (fast_model.py:209-218)
        const bf16_t o = f2bf(y);
        const int dim = a.qkv_dim;
        if (n < dim) {
            a.out_bf16[(size_t)m * dim + n] = o;
        } else {
            const int which = (n - dim) / dim, hd = (n - dim) % dim;
            const int row = m / a.seq_len, t = m % a.seq_len;
            bf16_t* cache = which ? a.vcache : a.kcache;
            cache[(((size_t)row * a.n_head + hd / 128) * a.cache_len + t) * 128 + hd % 128] = o;
        }
    } else if (a.out_f32) {  // second stage: fp32 residual stream, x = x + bf16(linear)  (combined.py:37-52)
        a.out_f32[(size_t)m * a.ldc + n] = (a.res_f32 ? a.res_f32[(size_t)m * a.ldc + n] : 0.f) + y;
    } else {  // first stage: bf16 residual stream (fast_model.py:179-180)
        a.out_bf16[(size_t)m * a.ldc + n] = a.res_bf16 ? f2bf(bf2f(a.res_bf16[(size_t)m * a.ldc + n]) + y) : f2bf(y);
    }
}

__global__ void __launch_bounds__(256) gemm_bf16_kernel(const GemmArgs a) {
    __shared__ __attribute__((aligned(16))) u32x4 lds[3][GT * 8];  // A, W, W2 tiles: [row][8 chunks of 16 B], chunk ^= row & 7
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int tile_m = blockIdx.y * GT, tile_n = blockIdx.x * GT;
    const bool gated = a.w2 != nullptr;
    const int M = a.M, N = a.N, K = a.K;

    // staging: thread owns chunk (tid & 7) of rows (tid >> 3) and (tid >> 3) + 32 of each tile
    const int sr = tid >> 3, sc = tid & 7;
    const bf16_t* ga0 = a.a + (size_t)min(tile_m + sr, M - 1) * a.lda + sc * 8;
    const bf16_t* ga1 = a.a + (size_t)min(tile_m + sr + 32, M - 1) * a.lda + sc * 8;
    const bf16_t* gw0 = a.w + (size_t)min(tile_n + sr, N - 1) * K + sc * 8;
    const bf16_t* gw1 = a.w + (size_t)min(tile_n + sr + 32, N - 1) * K + sc * 8;
    const bf16_t* gv0 = gated ? a.w2 + (size_t)min(tile_n + sr, N - 1) * K + sc * 8 : nullptr;
    const bf16_t* gv1 = gated ? a.w2 + (size_t)min(tile_n + sr + 32, N - 1) * K + sc * 8 : nullptr;
    const int s0 = sr * 8 + (sc ^ (sr & 7)), s1 = (sr + 32) * 8 + (sc ^ ((sr + 32) & 7));

    f32x4_t acc[2][2], acc2[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) { acc[i][j] = f32x4_t{0, 0, 0, 0}; acc2[i][j] = f32x4_t{0, 0, 0, 0}; }

    // K need only be a multiple of 8: 16-byte chunks past the end of a row are staged as zeros
    const u32x4 zero = {0u, 0u, 0u, 0u};
    // split-K: this block covers [kb, ke) of K (whole 64-column steps; the last range takes the remainder)
    const int ksteps = (K + GT - 1) / GT, per = (ksteps + a.splitk - 1) / a.splitk;
    const int kb = (int)blockIdx.z * per * GT, ke = min(K, (int)(blockIdx.z + 1) * per * GT);
    const bool in0 = kb + sc * 8 < ke;
    u32x4 ra0 = in0 ? load16(ga0 + kb) : zero, ra1 = in0 ? load16(ga1 + kb) : zero, rw0 = in0 ? load16(gw0 + kb) : zero,
          rw1 = in0 ? load16(gw1 + kb) : zero, rv0 = zero, rv1 = zero;
    if (gated && in0) { rv0 = load16(gv0 + kb); rv1 = load16(gv1 + kb); }
    for (int k0 = kb; k0 < ke; k0 += GT) {
        __syncthreads();  // everyone is done reading the previous tile
        lds[0][s0] = ra0; lds[0][s1] = ra1; lds[1][s0] = rw0; lds[1][s1] = rw1;
        if (gated) { lds[2][s0] = rv0; lds[2][s1] = rv1; }
        __syncthreads();
        if (k0 + GT < ke) {  // next tile's loads fly under this tile's MFMAs
            const bool in = k0 + GT + sc * 8 < ke;
            ra0 = in ? load16(ga0 + k0 + GT) : zero; ra1 = in ? load16(ga1 + k0 + GT) : zero;
            rw0 = in ? load16(gw0 + k0 + GT) : zero; rw1 = in ? load16(gw1 + k0 + GT) : zero;
            if (gated) { rv0 = in ? load16(gv0 + k0 + GT) : zero; rv1 = in ? load16(gv1 + k0 + GT) : zero; }
        }
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8_t af[2], bf[2], cf[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int r = wm * 32 + i * 16 + (lane & 15);
                af[i] = __builtin_bit_cast(bf16x8_t, lds[0][r * 8 + ((kk * 4 + (lane >> 4)) ^ (r & 7))]);
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int r = wn * 32 + j * 16 + (lane & 15);
                bf[j] = __builtin_bit_cast(bf16x8_t, lds[1][r * 8 + ((kk * 4 + (lane >> 4)) ^ (r & 7))]);
                if (gated) cf[j] = __builtin_bit_cast(bf16x8_t, lds[2][r * 8 + ((kk * 4 + (lane >> 4)) ^ (r & 7))]);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
                    if (gated) acc2[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], cf[j], acc2[i][j], 0, 0, 0);
                }
        }
    }

    // ---- epilogue: C fragment (i, j): column lane & 15, rows (lane >> 4) * 4 + reg ---------------------------------
    // split-K: this block's partial tile goes to the slab of its K range [range][M][N] (gated pairs: a second slab set behind it)
    float* slab = a.splitk > 1 ? a.slab + (size_t)blockIdx.z * M * N : nullptr;
    float* slab2 = a.splitk > 1 && gated ? a.slab + ((size_t)a.splitk + blockIdx.z) * M * N : nullptr;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = tile_n + wn * 32 + j * 16 + (lane & 15);
            if (n >= N) continue;
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int m = tile_m + wm * 32 + i * 16 + (lane >> 4) * 4 + reg;
                if (m >= M) continue;
                if (slab) {
                    slab[(size_t)m * N + n] = acc[i][j][reg];
                    if (slab2) slab2[(size_t)m * N + n] = acc2[i][j][reg];
                } else {
                    gemm_epilogue(a, m, n, acc[i][j][reg], acc2[i][j][reg]);
                }
            }
        }
}

// ---- few rows (a prefill: M = 2T ~ 256), big matrices: 64 columns x ALL rows per block -------------------------------------------
// With 64 x 64 tiles a prefill re-reads every weight tile M / 64 times through L2 and leaves each block a chain of short dependent
// K-steps: mv_begin spent 4.0 ms on 2.5 GB of weights (0.62 TB/s, round 3).  Here (cdna_hip_programming.md section 5, "M = 256
// projection GEMM") a block owns 64 output columns for a whole 256-row tile: the W tile is streamed once and shared by the block's 8
// waves (wave w: rows 32 w .. 32 w + 31, 2 x 4 MFMA tiles), the activations (1 MB at K = 2048: L2-resident, re-read by every column
// block) go through LDS in full 128-byte lines, two LDS buffers so that a K-step costs one barrier, the next step's global loads in
// flight under the current step's 16 MFMAs per wave; K is split over blockIdx.z until ~one block per CU is resident.  Same fragment
// mapping, same epilogues (gemm_epilogue) and the same fixed-order split-K combine as gemm_bf16_kernel.
#define SK_BM 256
// NT = 16-column MFMA tiles per wave = block columns / 16: 4 (64 columns) or 8 (128 columns: half the activation bytes per weight byte --
// the activation tile, re-read from L2 by every column block, is what bounds this shape: r04k, both tile shapes at 1.2 TB/s of weights)
template <int NT>
__global__ void __launch_bounds__(512) gemm_skinny_kernel(const GemmArgs a) {
    extern __shared__ __attribute__((aligned(16))) u32x4 sk_lds[];
    constexpr int BNS = NT * 16, WCH = BNS * 8;        // block columns; chunks of one W tile
    constexpr int BUF = SK_BM * 8 + 2 * WCH;           // chunks per buffer: A [256][8] + W [BNS][8] + W2 [BNS][8]
    constexpr int WPT = BNS / 64;                      // W rows staged per thread
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tile_m = blockIdx.y * SK_BM, tile_n = blockIdx.x * BNS;
    const bool gated = a.w2 != nullptr;
    const int M = a.M, N = a.N, K = a.K;
    // staging: thread owns chunk (tid & 7) of A rows (tid >> 3) + 64 i (i < 4) and of W rows (tid >> 3) + 64 j (j < WPT)
    const int sr = tid >> 3, sc = tid & 7;
    const bf16_t* ga[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) ga[i] = a.a + (size_t)min(tile_m + sr + 64 * i, M - 1) * a.lda + sc * 8;
    const bf16_t *gw[WPT], *gv[WPT];
#pragma unroll
    for (int j = 0; j < WPT; ++j) {
        gw[j] = a.w + (size_t)min(tile_n + sr + 64 * j, N - 1) * K + sc * 8;
        gv[j] = gated ? a.w2 + (size_t)min(tile_n + sr + 64 * j, N - 1) * K + sc * 8 : nullptr;
    }
    int sa[4], sw[WPT];
#pragma unroll
    for (int i = 0; i < 4; ++i) sa[i] = (sr + 64 * i) * 8 + (sc ^ ((sr + 64 * i) & 7));
#pragma unroll
    for (int j = 0; j < WPT; ++j) sw[j] = SK_BM * 8 + (sr + 64 * j) * 8 + (sc ^ ((sr + 64 * j) & 7));

    f32x4_t acc[2][NT], acc2[2][NT];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) { acc[i][j] = f32x4_t{0, 0, 0, 0}; acc2[i][j] = f32x4_t{0, 0, 0, 0}; }

    const u32x4 zero = {0u, 0u, 0u, 0u};
    const int ksteps = (K + GT - 1) / GT, per = (ksteps + a.splitk - 1) / a.splitk;
    const int kb = (int)blockIdx.z * per * GT, ke = min(K, (int)(blockIdx.z + 1) * per * GT);
    u32x4 ra[4], rw[WPT], rv[WPT];
    auto fetch = [&](int k0) {
        const bool in = k0 + sc * 8 < ke;
#pragma unroll
        for (int i = 0; i < 4; ++i) ra[i] = in ? load16(ga[i] + k0) : zero;
#pragma unroll
        for (int j = 0; j < WPT; ++j) {
            rw[j] = in ? load16(gw[j] + k0) : zero;
            rv[j] = gated && in ? load16(gv[j] + k0) : zero;
        }
    };
    auto stage = [&](int buf) {
        u32x4* L = sk_lds + buf * BUF;
#pragma unroll
        for (int i = 0; i < 4; ++i) L[sa[i]] = ra[i];
#pragma unroll
        for (int j = 0; j < WPT; ++j) {
            L[sw[j]] = rw[j];
            if (gated) L[sw[j] + WCH] = rv[j];
        }
    };
    if (kb < ke) {
        fetch(kb);
        stage(0);
    }
    __syncthreads();
    int buf = 0;
    for (int k0 = kb; k0 < ke; k0 += GT) {
        const bool more = k0 + GT < ke;
        if (more) fetch(k0 + GT);   // next step's loads fly under this step's MFMAs
        const u32x4* L = sk_lds + buf * BUF;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8_t af[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int r = wave * 32 + i * 16 + (lane & 15);
                af[i] = __builtin_bit_cast(bf16x8_t, L[r * 8 + ((kk * 4 + (lane >> 4)) ^ (r & 7))]);
            }
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int r = j * 16 + (lane & 15), idx = SK_BM * 8 + r * 8 + ((kk * 4 + (lane >> 4)) ^ (r & 7));
                const bf16x8_t bf = __builtin_bit_cast(bf16x8_t, L[idx]);
#pragma unroll
                for (int i = 0; i < 2; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bf, acc[i][j], 0, 0, 0);
                if (gated) {
                    const bf16x8_t cf = __builtin_bit_cast(bf16x8_t, L[idx + WCH]);
#pragma unroll
                    for (int i = 0; i < 2; ++i) acc2[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], cf, acc2[i][j], 0, 0, 0);
                }
            }
        }
        if (more) stage(buf ^ 1);   // the other buffer: last read before the barrier that ended the previous step
        __syncthreads();
        buf ^= 1;
    }
    // NT = 8 always leaves fp32 partial tiles (launch_gemm runs the combine kernel behind it even for one K range): 64 inlined copies of
    // the epilogue are past hipcc's full-unroll budget, and a loop it does not unroll indexes the accumulators dynamically -- they went
    // to scratch memory (528 bytes per lane)
    const bool to_slab = a.splitk > 1 || NT == 8;
    float* slab = to_slab ? a.slab + (size_t)blockIdx.z * M * N : nullptr;
    float* slab2 = to_slab && gated ? a.slab + ((size_t)a.splitk + blockIdx.z) * M * N : nullptr;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int n = tile_n + j * 16 + (lane & 15);
            if (n >= N) continue;
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int m = tile_m + wave * 32 + i * 16 + (lane >> 4) * 4 + reg;
                if (m >= M) continue;
                if (NT == 8 || slab) {
                    slab[(size_t)m * N + n] = acc[i][j][reg];
                    if (slab2) slab2[(size_t)m * N + n] = acc2[i][j][reg];
                } else {
                    gemm_epilogue(a, m, n, acc[i][j][reg], acc2[i][j][reg]);
                }
            }
        }
}
template <int NT> constexpr int sk_lds_bytes() { return 2 * (SK_BM * 8 + 2 * NT * 16 * 8) * 16; }

// split-K tail: partial tiles added in range order (fixed: the result does not depend on which block finished first), then the epilogue
__global__ void __launch_bounds__(256) gemm_combine_kernel(const GemmArgs a) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x, MN = (size_t)a.M * a.N;
    if (i >= MN) return;
    float v = 0.f, v2 = 0.f;
    for (int z = 0; z < a.splitk; ++z) v += a.slab[(size_t)z * MN + i];
    if (a.w2)
        for (int z = 0; z < a.splitk; ++z) v2 += a.slab[((size_t)a.splitk + z) * MN + i];
    gemm_epilogue(a, (int)(i / a.N), (int)(i % a.N), v, v2);
}

hipError_t launch_gemm(hipStream_t s, const GemmArgs& a_in) {
    GemmArgs a = a_in;
    if (a.M < 1 || a.N < 1 || a.K < 8 || a.K % 8 != 0 || a.lda % 8 != 0) return hipErrorInvalidValue;
    const int nt = (a.N + GT - 1) / GT, mt = (a.M + GT - 1) / GT, ksteps = (a.K + GT - 1) / GT;
    // split K (2, 4 or 8 ranges) while the tile grid alone is below ~4 blocks per CU (1024) and every range keeps >= 4 steps: a block's K-loop is a
    // chain of dependent load round trips, so what fills the memory system is the number of resident blocks
    a.splitk = 1;
    // few rows, big matrix (a prefill): 64 / 128 columns x all rows per block, K split until ~one block per CU is resident -- behind
    // MV_GEMM_SKINNY=1 (tests, A/B runs): measured on the prefill it is NOT faster than the 64 x 64 tiles (r04j-r04l: mv_begin 3.7-3.9 ms
    // at T = 128 with either, every GEMM launch ~21.6 us = 1.2 TB/s of weights whatever the tiling -- with one K-step of prefetch distance a
    // block is a chain of dependent load round trips and the resident blocks hold too few bytes in flight; 8.4 vs 7.9 ms at T = 512).
    const size_t per_range = (size_t)a.M * a.N * (a.w2 ? 2 : 1);   // floats of partial tiles one K range leaves in the slab
    const size_t max_ranges = a.slab && per_range ? a.slab_elems / per_range : 0;   // what the caller's slab holds
    if (a.M >= 96 && a.K >= 512 && a.N >= 512 && a.M <= 8 * SK_BM && getenv("MV_GEMM_SKINNY")) {
        const int mt256 = (a.M + SK_BM - 1) / SK_BM;
        const char* cols_env = getenv("MV_GEMM_SKINNY_COLS");   // A/B: 64 or 128 columns per block (default 128)
        // the 128-column instance ALWAYS leaves partial tiles (even with one range): it needs room for at least one range in the slab
        // (the gated w1|w3 GEMM of a long prompt -- N = 5632, M = 2 T >= ~1490 -- did not fit the prefill's fixed slab and wrote past it)
        const int cols = (cols_env ? atoi(cols_env) : 128) == 64 || max_ranges < 1 ? 64 : 128;
        const int ntc = (a.N + cols - 1) / cols;
        if (a.slab && !getenv("MV_GEMM_NO_SPLITK"))
            while (a.splitk < 8 && (size_t)a.splitk * 2 <= max_ranges && (size_t)ntc * mt256 * a.splitk * 2 <= 320 && ksteps / (a.splitk * 2) >= 4) a.splitk *= 2;
        static std::atomic<bool> raised_on[64];   // the attribute belongs to the DEVICE's copy of the function
        int dev = 0;
        (void)hipGetDevice(&dev);
        std::atomic<bool>& raised = raised_on[dev & 63];
        if (!raised) {
            hipError_t e = hipFuncSetAttribute((const void*)gemm_skinny_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, sk_lds_bytes<4>());
            if (e == hipSuccess) e = hipFuncSetAttribute((const void*)gemm_skinny_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, sk_lds_bytes<8>());
            if (e != hipSuccess) return e;
            raised = true;
        }
        if (cols == 64) hipLaunchKernelGGL(gemm_skinny_kernel<4>, dim3(ntc, mt256, a.splitk), dim3(512), sk_lds_bytes<4>(), s, a);
        else hipLaunchKernelGGL(gemm_skinny_kernel<8>, dim3(ntc, mt256, a.splitk), dim3(512), sk_lds_bytes<8>(), s, a);
        if (a.splitk > 1 || cols == 128) hipLaunchKernelGGL(gemm_combine_kernel, dim3((unsigned)(((size_t)a.M * a.N + 255) / 256)), dim3(256), 0, s, a);
        return hipGetLastError();
    }
    if (a.slab && !getenv("MV_GEMM_NO_SPLITK"))
        while (a.splitk < 8 && (size_t)a.splitk * 2 <= max_ranges && (size_t)nt * mt * a.splitk < 1024 && ksteps / (a.splitk * 2) >= 4) a.splitk *= 2;
    hipLaunchKernelGGL(gemm_bf16_kernel, dim3(nt, mt, a.splitk), dim3(256), 0, s, a);
    if (a.splitk > 1) hipLaunchKernelGGL(gemm_combine_kernel, dim3((unsigned)(((size_t)a.M * a.N + 255) / 256)), dim3(256), 0, s, a);
    return hipGetLastError();
}

// ---- row normalisation: y[m][:] = norm(x[m][:]) -> bf16 ----------------------------------------------------------------
// mode 0: first-stage RMSNorm on a bf16 stream: bf16(bf16(x * rstd) * w)              (fast_model.py:250-261)
// mode 1: second-stage RMSNorm under autocast: fp32 x * rstd * w, rounded once at the Linear input (layers.py:20-30)
// mode 2: second-stage LayerNorm (fp32, weight and optional bias), rounded once        (layers.py:8-17)
__global__ void __launch_bounds__(256) norm_rows_kernel(const NormArgs a) {
    __shared__ float red[2][4];
    const int m = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int C = a.C;
    float s = 0.f, s2 = 0.f;
    for (int c = tid; c < C; c += 256) {
        const float v = a.x_f32 ? a.x_f32[(size_t)m * C + c] : bf2f(a.x_bf16[(size_t)m * C + c]);
        s += v;
        s2 += v * v;
    }
    s = wave_sum(s);
    s2 = wave_sum(s2);
    if (lane == 0) { red[0][wave] = s; red[1][wave] = s2; }
    __syncthreads();
    const float sum = red[0][0] + red[0][1] + red[0][2] + red[0][3];
    const float sq = red[1][0] + red[1][1] + red[1][2] + red[1][3];
    const float mean = (a.mode == 2) ? sum / C : 0.f;
    const float var = (a.mode == 2) ? fmaxf(sq / C - mean * mean, 0.f) : sq / C;
    const float rstd = rsqrtf(var + a.eps);
    for (int c = tid; c < C; c += 256) {
        const float v = a.x_f32 ? a.x_f32[(size_t)m * C + c] : bf2f(a.x_bf16[(size_t)m * C + c]);
        const float w = a.w_f32 ? a.w_f32[c] : bf2f(a.w_bf16[c]);
        float y;
        if (a.mode == 0) y = round_bf(round_bf(v * rstd) * w);
        else if (a.mode == 1) y = round_bf(v * rstd * w);
        else y = round_bf((v - mean) * rstd * w + (a.b_f32 ? a.b_f32[c] : 0.f));
        a.out[(size_t)m * C + c] = f2bf(y);
    }
}
hipError_t launch_norm_rows(hipStream_t s, const NormArgs& a) {
    hipLaunchKernelGGL(norm_rows_kernel, dim3(a.M), dim3(256), 0, s, a);
    return hipGetLastError();
}

// ---- multi-query attention: one block per (query, head, batch row), keys streamed from L2 ------------------------------
// First-stage prefill: causal over the KV cache just written, `F.scaled_dot_product_attention(..., attn_mask=causal)`
// (fast_model.py:218-222).  Second stage: `is_causal=False` over all block_size positions (layers/attn.py:132-159).
// T <= 2048 and the K/V of one (row, head) is <= 512 KB: L2-resident, so a query re-streams it from L2; scores, softmax
// and P.V are fp32 (online softmax across 4-instruction chunks), output bf16.  HD/8 lanes cover one key row (16 B each).
template <int HD>
__global__ void __launch_bounds__(256) attn_rows_kernel(const AttnRowsArgs a) {
    constexpr int LPK = HD / 8;       // lanes per key
    constexpr int KPI = 64 / LPK;     // keys per wave instruction
    constexpr int UN = 4;
    __shared__ __attribute__((aligned(16))) float sm[4][LPK][12];
    const int t = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane / LPK, dl = lane % LPK;
    const int nkeys = a.causal ? t + 1 : a.T;
    const bf16_t* kb = a.k + (size_t)b * a.k_batch_stride + (size_t)h * a.k_head_stride + dl * 8;
    const bf16_t* vb = a.v + (size_t)b * a.k_batch_stride + (size_t)h * a.k_head_stride + dl * 8;
    const u32x4 qv = load16(a.q + ((size_t)b * a.T + t) * a.q_row_stride + h * HD + dl * 8);
    const float scale = rsqrtf((float)HD);

    float m = -INFINITY, l = 0.f, o[8];
#pragma unroll
    for (int d = 0; d < 8; ++d) o[d] = 0.f;
    for (int base = wave * KPI * UN; base < nkeys; base += 4 * KPI * UN) {
        u32x4 kv[UN], vv[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int kc = min(base + u * KPI + g, nkeys - 1);
            kv[u] = load16(kb + (size_t)kc * a.k_key_stride);
            vv[u] = load16(vb + (size_t)kc * a.k_key_stride);
        }
        float sc[UN], mc = -INFINITY;
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            float s = dot8_bf16(kv[u], qv, 0.f);
            s += dpp_f<0xB1>(s);
            s += dpp_f<0x4E>(s);
            s += dpp_f<0x141>(s);
            if (LPK == 16) s += dpp_f<0x140>(s);
            s = (base + u * KPI + g < nkeys) ? s * scale : -INFINITY;
            sc[u] = s;
            mc = fmaxf(mc, s);
        }
        if (mc == -INFINITY) continue;
        const float mn = fmaxf(m, mc);
        const float corr = (m == -INFINITY) ? 0.f : fast_exp(m - mn);
        l *= corr;
#pragma unroll
        for (int d = 0; d < 8; ++d) o[d] *= corr;
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const float p = (sc[u] == -INFINITY) ? 0.f : fast_exp(sc[u] - mn);
            l += p;
            o[0] = fmaf(p, bf_lo(vv[u].x), o[0]); o[1] = fmaf(p, bf_hi(vv[u].x), o[1]);
            o[2] = fmaf(p, bf_lo(vv[u].y), o[2]); o[3] = fmaf(p, bf_hi(vv[u].y), o[3]);
            o[4] = fmaf(p, bf_lo(vv[u].z), o[4]); o[5] = fmaf(p, bf_hi(vv[u].z), o[5]);
            o[6] = fmaf(p, bf_lo(vv[u].w), o[6]); o[7] = fmaf(p, bf_hi(vv[u].w), o[7]);
        }
        m = mn;
    }
    // fold the KPI lane groups of the wave (xor over the group index bits), then the 4 waves through LDS
#pragma unroll
    for (int off = LPK; off < 64; off <<= 1) {
        const float mo = fmaxf(m, __shfl_xor(m, off, 64));
        const float w = (m == -INFINITY) ? 0.f : fast_exp(m - mo);
        l *= w;
        l += __shfl_xor(l, off, 64);
#pragma unroll
        for (int d = 0; d < 8; ++d) {
            const float tt = o[d] * w;
            o[d] = tt + __shfl_xor(tt, off, 64);
        }
        m = mo;
    }
    if (g == 0) {
        sm[wave][dl][0] = m;
        sm[wave][dl][1] = l;
#pragma unroll
        for (int d = 0; d < 8; ++d) sm[wave][dl][2 + d] = o[d];
    }
    __syncthreads();
    if (threadIdx.x < LPK) {
        float mb = -INFINITY;
#pragma unroll
        for (int w = 0; w < 4; ++w) mb = fmaxf(mb, sm[w][dl][0]);
        float lb = 0.f, ob[8];
#pragma unroll
        for (int d = 0; d < 8; ++d) ob[d] = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const float mw = sm[w][dl][0];
            const float f = (mw == -INFINITY) ? 0.f : fast_exp(mw - mb);
            lb += sm[w][dl][1] * f;
#pragma unroll
            for (int d = 0; d < 8; ++d) ob[d] += sm[w][dl][2 + d] * f;
        }
        const float inv = 1.0f / lb;
        u32x4 y;
        y.x = pack_bf2(ob[0] * inv, ob[1] * inv);
        y.y = pack_bf2(ob[2] * inv, ob[3] * inv);
        y.z = pack_bf2(ob[4] * inv, ob[5] * inv);
        y.w = pack_bf2(ob[6] * inv, ob[7] * inv);
        *(u32x4*)(a.out + ((size_t)b * a.T + t) * a.out_row_stride + h * HD + dl * 8) = y;
    }
}
// ---- MFMA attention over whole rows (prefill of the first stage: causal; second stage: all positions) ------------------------
// One wave = 16 queries of one (batch row, head); no LDS, no barriers: every MFMA operand is a K-contiguous fragment loaded
// straight from global memory (the K / V of a (row, head) is at most 512 KB: L2-resident, re-read by the query blocks).
//   S^T = K Q^T  (the "swapped" product): v_mfma_f32_16x16x32_bf16 with A = 16 keys x head_dim (rows of the KV cache), B = the wave's
//   16 queries; the accumulator then holds, for query lane & 15, the scores of keys 4 (lane >> 4) + reg -- a lane owns ONE query's
//   scores, so the softmax statistics are per-lane running values merged once across the 4 lanes that share a query.
//   O = P V: A = P (the lane's own query row: its 8 probabilities of a 32-key block are exactly the 8 k-slots the MFMA asks this
//   lane for -- keys 4 g .. 4 g + 3 and 16 + 4 g .. + 3, g = lane >> 4; the order of keys inside a 32-key block is free as long as
//   V uses the same one), B = V^T rows (head dim x keys, key-contiguous): V is transposed once per call by vt_transpose_kernel into
//   the caller's scratch, zero-padded to a multiple of 64 keys.
// Two passes over the keys (pass 1: running max / sum, pass 2: P = exp(s - m) / l and P V), scores and softmax in fp32; P enters the
// MFMA as bf16 hi + lo (two MFMAs per tile: ~16 bits of P; attention is < 1 % of the prefill's FLOPs), output bf16.
// fast_model.py:218-222 (causal mask over the cache just written), fam/llm/layers/attn.py:132-159 (is_causal=False).
template <int HD>
__global__ void __launch_bounds__(256) vt_transpose_kernel(const AttnRowsArgs a, int Tpad) {
    __shared__ bf16_t tile[64][HD + 2];
    const int kb = blockIdx.x * 64, h = blockIdx.y, b = blockIdx.z;
    const bf16_t* vb = a.v + (size_t)b * a.k_batch_stride + (size_t)h * a.k_head_stride;
    for (int i = threadIdx.x; i < 64 * (HD / 8); i += 256) {
        const int key = i / (HD / 8), c = i % (HD / 8);
        u32x4 v = u32x4{0u, 0u, 0u, 0u};
        if (kb + key < a.T) v = load16(vb + (size_t)(kb + key) * a.k_key_stride + c * 8);
        bf16_t* row = &tile[key][c * 8];
        *(uint32_t*)(row + 0) = v.x; *(uint32_t*)(row + 2) = v.y; *(uint32_t*)(row + 4) = v.z; *(uint32_t*)(row + 6) = v.w;
    }
    __syncthreads();
    bf16_t* vt = a.vt_scratch + ((size_t)b * a.n_head + h) * HD * Tpad + kb;
    for (int i = threadIdx.x; i < HD * 8; i += 256) {
        const int d = i >> 3, k8 = (i & 7) * 8;
        u32x4 o;
        o.x = (uint32_t)tile[k8 + 0][d] | ((uint32_t)tile[k8 + 1][d] << 16);
        o.y = (uint32_t)tile[k8 + 2][d] | ((uint32_t)tile[k8 + 3][d] << 16);
        o.z = (uint32_t)tile[k8 + 4][d] | ((uint32_t)tile[k8 + 5][d] << 16);
        o.w = (uint32_t)tile[k8 + 6][d] | ((uint32_t)tile[k8 + 7][d] << 16);
        *(u32x4*)(vt + (size_t)d * Tpad + k8) = o;
    }
}

template <int HD>
__global__ void __launch_bounds__(64) attn_mfma_kernel(const AttnRowsArgs a, int Tpad) {
    typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
    typedef float f32x4_t __attribute__((ext_vector_type(4)));
    constexpr int KJ = HD / 32;   // MFMAs per 16 x 16 score tile
    constexpr int DT = HD / 16;   // output tiles along the head dimension
    const int qb = blockIdx.x * 16, h = blockIdx.y, b = blockIdx.z;
    const int lane = threadIdx.x, g = lane >> 4, c = lane & 15;
    const int T = a.T;
    const int query = min(qb + c, T - 1);                   // this lane's query (clamped: rows past T are computed, not stored)
    const int last_key = a.causal ? min(qb + 15, T - 1) : T - 1;   // keys any query of the block can see
    const int my_last = a.causal ? query : T - 1;
    const float scale = rsqrtf((float)HD);
    const bf16_t* kbase = a.k + (size_t)b * a.k_batch_stride + (size_t)h * a.k_head_stride + g * 8;
    const bf16_t* vt = a.vt_scratch + ((size_t)b * a.n_head + h) * HD * Tpad;

    bf16x8_t qf[KJ];
#pragma unroll
    for (int j = 0; j < KJ; ++j) qf[j] = __builtin_bit_cast(bf16x8_t, load16(a.q + ((size_t)b * T + query) * a.q_row_stride + h * HD + 32 * j + g * 8));

    // scores of keys kb + 4 g + reg for this lane's query; masked ones come back as -inf
    auto score_tile = [&](int kb) {
        const int key = min(kb + c, T - 1);
        f32x4_t sacc = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < KJ; ++j) {
            const bf16x8_t kf = __builtin_bit_cast(bf16x8_t, load16(kbase + (size_t)key * a.k_key_stride + 32 * j));
            sacc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[j], sacc, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) sacc[r] = (kb + 4 * g + r <= my_last) ? sacc[r] * scale : -INFINITY;
        return sacc;
    };

    // ---- pass 1: running max and sum over this lane's keys, merged across the 4 lanes of the query
    float m = -INFINITY, l = 0.f;
    for (int kb = 0; kb <= last_key; kb += 16) {
        const f32x4_t sv = score_tile(kb);
        const float mc = fmaxf(fmaxf(sv[0], sv[1]), fmaxf(sv[2], sv[3]));
        if (mc == -INFINITY) continue;
        const float mn = fmaxf(m, mc);
        l *= (m == -INFINITY) ? 0.f : fast_exp(m - mn);
#pragma unroll
        for (int r = 0; r < 4; ++r) l += (sv[r] == -INFINITY) ? 0.f : fast_exp(sv[r] - mn);
        m = mn;
    }
#pragma unroll
    for (int off = 16; off < 64; off <<= 1) {
        const float mo = __shfl_xor(m, off, 64), lo = __shfl_xor(l, off, 64);
        const float mn = fmaxf(m, mo);   // finite: key 0 is visible to every query
        l = l * ((m == -INFINITY) ? 0.f : fast_exp(m - mn)) + lo * ((mo == -INFINITY) ? 0.f : fast_exp(mo - mn));
        m = mn;
    }
    const float inv_l = 1.0f / l;

    // ---- pass 2: P = exp(s - m) / l in 32-key blocks, O += P V
    f32x4_t oacc[DT];
#pragma unroll
    for (int d = 0; d < DT; ++d) oacc[d] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    for (int kb = 0; kb <= last_key; kb += 32) {
        float p[8];
        {
            const f32x4_t s0 = score_tile(kb);
#pragma unroll
            for (int r = 0; r < 4; ++r) p[r] = (s0[r] == -INFINITY) ? 0.f : fast_exp(s0[r] - m) * inv_l;
            if (kb + 16 <= last_key) {
                const f32x4_t s1 = score_tile(kb + 16);
#pragma unroll
                for (int r = 0; r < 4; ++r) p[4 + r] = (s1[r] == -INFINITY) ? 0.f : fast_exp(s1[r] - m) * inv_l;
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) p[4 + r] = 0.f;
            }
        }
        u32x4 ph, pl;
        ph.x = pack_bf2(p[0], p[1]); ph.y = pack_bf2(p[2], p[3]); ph.z = pack_bf2(p[4], p[5]); ph.w = pack_bf2(p[6], p[7]);
        pl.x = pack_bf2(p[0] - bf_lo(ph.x), p[1] - bf_hi(ph.x)); pl.y = pack_bf2(p[2] - bf_lo(ph.y), p[3] - bf_hi(ph.y));
        pl.z = pack_bf2(p[4] - bf_lo(ph.z), p[5] - bf_hi(ph.z)); pl.w = pack_bf2(p[6] - bf_lo(ph.w), p[7] - bf_hi(ph.w));
        const bf16x8_t pfh = __builtin_bit_cast(bf16x8_t, ph), pfl = __builtin_bit_cast(bf16x8_t, pl);
#pragma unroll
        for (int d = 0; d < DT; ++d) {
            // V^T row d 16 + c, keys kb + 4 g .. + 3 and kb + 16 + 4 g .. + 3 (the k-slots of lane group g)
            const bf16_t* row = vt + (size_t)(d * 16 + c) * Tpad + kb + 4 * g;
            const uint2 v0 = *(const uint2*)row, v1 = *(const uint2*)(row + 16);
            const bf16x8_t vf = __builtin_bit_cast(bf16x8_t, u32x4{v0.x, v0.y, v1.x, v1.y});
            oacc[d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pfh, vf, oacc[d], 0, 0, 0);
            oacc[d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pfl, vf, oacc[d], 0, 0, 0);
        }
    }
    // oacc[d][reg] = O[query qb + 4 g + reg][d 16 + c]
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int qr = qb + 4 * g + r;
        if (qr >= T) continue;
        bf16_t* orow = a.out + ((size_t)b * T + qr) * a.out_row_stride + h * HD + c;
#pragma unroll
        for (int d = 0; d < DT; ++d) orow[d * 16] = f2bf(oacc[d][r]);
    }
}

template <int HD>
static hipError_t launch_attn_mfma(hipStream_t s, const AttnRowsArgs& a) {
    const int Tpad = (a.T + 63) / 64 * 64;
    hipLaunchKernelGGL(vt_transpose_kernel<HD>, dim3(Tpad / 64, a.n_head, a.batch), dim3(256), 0, s, a, Tpad);
    hipLaunchKernelGGL(attn_mfma_kernel<HD>, dim3((a.T + 15) / 16, a.n_head, a.batch), dim3(64), 0, s, a, Tpad);
    return hipGetLastError();
}

hipError_t launch_attn_rows(hipStream_t s, const AttnRowsArgs& a) {
    const bool force_valu = getenv("MV_ATTN_VALU") != nullptr;   // A/B switch, read per call (tests flip it inside one process): the one-query-per-block kernel everywhere
    if (a.vt_scratch != nullptr && !force_valu) {
        if (a.head_dim == 128) return launch_attn_mfma<128>(s, a);
        if (a.head_dim == 64) return launch_attn_mfma<64>(s, a);
        return hipErrorInvalidValue;
    }
    const dim3 grid(a.T, a.n_head, a.batch);
    if (a.head_dim == 128) hipLaunchKernelGGL(attn_rows_kernel<128>, grid, dim3(256), 0, s, a);
    else if (a.head_dim == 64) hipLaunchKernelGGL(attn_rows_kernel<64>, grid, dim3(256), 0, s, a);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}
