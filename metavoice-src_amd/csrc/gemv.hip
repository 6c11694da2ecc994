// Decode-step projections: out[r][n] = sum_k x[r][k] * W[n][k] for a handful of rows r (2 per utterance: the
// speaker-conditioned and the unconditioned CFG row, fam/llm/fast_model.py:132-134,152-157).
//
// Replaces, for S=1: RMSNorm + wqkv + KVCache.update (fast_model.py:179,209,104-113), wo + residual (:224-226,179),
// RMSNorm + silu(w1 x)*w3 x (:180,237), w2 + residual (:247,180), final norm + LM head (:161-162).
//
// Roofline: HBM.  Each bf16 weight is read exactly once per step and used for `rows` FMAs (2 FLOP/B at rows=2,
// ridge ~300 FLOP/B), so the kernels are organised around the weight stream, not around MFMA:
//   * W keeps the checkpoint's [N][K] layout; one wave owns whole output rows, lane l reads the 16 bytes at
//     columns 8l..8l+7 of each 512-column chunk -> every wave instruction is one contiguous 1 KiB segment,
//     `global_load_dwordx4 ... nt` (read-once, non-temporal).
//   * all weight loads of a wave (NMAT*NR*KCH x 16 B per lane) are issued BEFORE the activation prologue, so the
//     HBM latency overlaps the norm; no conditional loads (addresses are clamped instead: a branch per load
//     would serialise them behind vmcnt(0)).
//   * the tiny activation vector (rows x K bf16) is normalised once per block into LDS and read back with
//     conflict-free ds_read_b128 (lane-linear), reused by the NR weight rows of the wave.
//   * v_dot2c_f32_bf16 accumulates in fp32; one 64-lane butterfly per output; bf16 rounding points follow the
//     reference's bf16 tensors (DESIGN.md "Rounding points").
#include "gemv_core.h"
#include "kernels.h"

enum { EPI_PLAIN = GEMV_PLAIN, EPI_RESID = GEMV_RESID, EPI_SWIGLU = GEMV_SWIGLU, EPI_QKV = GEMV_QKV };

template <int R, int KCH, int NR, int EPI, bool NT>
__global__ void __launch_bounds__(1024) gemv_kernel(const GemvArgs a) {
    constexpr int NMAT = (EPI == EPI_SWIGLU) ? 2 : 1;
    constexpr int KP = KCH * 512;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    bf16_t* xs = (bf16_t*)smem;                 // [R][KP] bf16, zero beyond K
    float* red = (float*)(smem + R * KP * 2);   // [nwaves][R]

    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int nwaves = blockDim.x >> 6;
    const int K = a.K, N = a.N;
    const int n0 = (blockIdx.x * nwaves + wave) * NR;

    // ---- 1. put the whole weight working set of this wave in flight ------------------------------------
    u32x4 wv[NMAT][NR][KCH];
#pragma unroll
    for (int i = 0; i < NR; ++i)
#pragma unroll
        for (int j = 0; j < KCH; ++j) { wv[0][i][j] = u32x4{0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u}; wv[NMAT - 1][i][j] = wv[0][i][j]; }
    if (!(a.ablate & 1))
#pragma unroll
    for (int i = 0; i < NR; ++i) {
        const int n = min(n0 + i, N - 1);
#pragma unroll
        for (int j = 0; j < KCH; ++j) {
            const int c = min(j * 512 + lane * 8, K - 8);
            wv[0][i][j] = NT ? load_stream16(a.w + (size_t)n * K + c) : load16(a.w + (size_t)n * K + c);
            if (NMAT == 2) wv[NMAT - 1][i][j] = NT ? load_stream16(a.w2 + (size_t)n * K + c) : load16(a.w2 + (size_t)n * K + c);
        }
    }

    // The epilogue's own operands (this lane's residual element, the cache position) are fetched NOW: left in the
    // epilogue they are a dependent ~1 us round trip at the very end of every launch (profiles/r01e ablation).
    bf16_t res_pre = 0;
    int pos_pre = 0;
    {
        const int i = lane / R, r = lane % R, n = n0 + i;
        if (lane < NR * R && n < N) {
            if (EPI == EPI_RESID) res_pre = a.residual[(size_t)r * N + n];
            if (EPI == EPI_QKV) pos_pre = a.pos[r >> 1];
        }
    }

    // ---- 2. activation prologue -> LDS ------------------------------------------------------------------
    if (a.ablate & 4) {
    } else if (a.norm_w != nullptr) {
        // RMSNorm (fast_model.py:250-261): fp32 x * rsqrt(mean(x^2) + eps) -> bf16 -> * weight -> bf16.
        // Thread t owns columns 8t..8t+7 of every row (host guarantees blockDim.x * 8 >= KP).
        const int c = threadIdx.x * 8;
        const bool in_k = c < K;
        u32x4 xv[R];
        float ss[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            xv[r] = in_k ? load16(a.x + (size_t)r * K + c) : u32x4{0, 0, 0, 0};
            ss[r] = wave_sum(sumsq8(xv[r]));
        }
        if (lane == 0) {
#pragma unroll
            for (int r = 0; r < R; ++r) red[wave * R + r] = ss[r];
        }
        const u32x4 nw = in_k ? load16(a.norm_w + c) : u32x4{0, 0, 0, 0};
        __syncthreads();
        float tots[R];
#pragma unroll
        for (int r = 0; r < R; ++r) tots[r] = block_total(red + r, R, nwaves, lane);
        if (c < KP) {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                *(u32x4*)(xs + r * KP + c) = rms_apply8(xv[r], rms_rstd(tots[r], K, a.eps), nw);
            }
        }
    } else {
        for (int it = threadIdx.x; it < R * (KP / 8); it += blockDim.x) {
            const int r = it / (KP / 8), c = (it % (KP / 8)) * 8;
            *(u32x4*)(xs + r * KP + c) = (c < K) ? load16(a.x + (size_t)r * K + c) : u32x4{0, 0, 0, 0};
        }
    }
    __syncthreads();

    // ---- 3. dot products --------------------------------------------------------------------------------
    float acc[NMAT][NR][R];
#pragma unroll
    for (int m = 0; m < NMAT; ++m)
#pragma unroll
        for (int i = 0; i < NR; ++i)
#pragma unroll
            for (int r = 0; r < R; ++r) acc[m][i][r] = 0.f;
    if (!(a.ablate & 2))
#pragma unroll
    for (int j = 0; j < KCH; ++j) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const u32x4 xv = *(const u32x4*)(xs + r * KP + j * 512 + lane * 8);
#pragma unroll
            for (int m = 0; m < NMAT; ++m)
#pragma unroll
                for (int i = 0; i < NR; ++i) acc[m][i][r] = dot8_bf16(wv[m][i][j], xv, acc[m][i][r]);
        }
    }

    // ---- 4. 64-lane reductions; lane i*R+r keeps output (n0+i, r) ------------------------------------------
    float mine[NMAT];
#pragma unroll
    for (int m = 0; m < NMAT; ++m) mine[m] = 0.f;
    if (a.ablate & 8) {
        float t = 0.f;
#pragma unroll
        for (int m = 0; m < NMAT; ++m)
#pragma unroll
            for (int i = 0; i < NR; ++i)
#pragma unroll
                for (int r = 0; r < R; ++r) t += acc[m][i][r];
        if (t == 12345.678f) a.out[0] = 0;  // keep the accumulators alive
        return;
    }
#pragma unroll
    for (int m = 0; m < NMAT; ++m)
#pragma unroll
        for (int i = 0; i < NR; ++i)
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const float t = wave_sum(acc[m][i][r]);
                if (lane == i * R + r) mine[m] = t;
            }

    // ---- 5. epilogue ------------------------------------------------------------------------------------
    if (lane < NR * R) {
        const int i = lane / R, r = lane % R;
        const int n = n0 + i;
        if (n < N) {
            if (EPI == EPI_PLAIN) {
                a.out[(size_t)r * N + n] = f2bf(mine[0]);
            } else if (EPI == EPI_RESID) {
                a.out[(size_t)r * N + n] = epi_resid(res_pre, mine[0]);
            } else if (EPI == EPI_SWIGLU) {
                a.out[(size_t)r * N + n] = epi_swiglu(mine[0], mine[NMAT - 1]);
            } else {  // EPI_QKV: rows [0,dim) -> q, [dim,2dim) -> K cache, [2dim,3dim) -> V cache at input_pos
                const bf16_t v = f2bf(mine[0]);
                const int dim = a.dim;
                if (n < dim) {
                    a.out[(size_t)r * dim + n] = v;
                } else {
                    const int which = (n - dim) / dim;
                    const int hd = (n - dim) % dim;
                    const int h = hd / 128, d = hd % 128;
                    bf16_t* cache = which ? a.vcache : a.kcache;
                    cache[(((size_t)r * a.n_head + h) * a.cache_len + pos_pre) * 128 + d] = v;
                }
            }
        }
    }
}

MvTune g_tune;

template <int R, int KCH, int NR, int EPI>
static hipError_t launch_one(hipStream_t s, const GemvArgs& a, int threads) {
    const int nwaves = threads / 64;
    const int blocks = (a.N + nwaves * NR - 1) / (nwaves * NR);
    const size_t lds = (size_t)R * KCH * 512 * 2 + (size_t)nwaves * R * 4;
    if (g_tune.gemv_nt) hipLaunchKernelGGL((gemv_kernel<R, KCH, NR, EPI, true>), dim3(blocks), dim3(threads), lds, s, a);
    else hipLaunchKernelGGL((gemv_kernel<R, KCH, NR, EPI, false>), dim3(blocks), dim3(threads), lds, s, a);
    return hipGetLastError();
}

// Weight rows per wave.  Built-in: 1 (3 for the fused QKV); the alternative (2, or 1 for QKV) is kept for tools/sweep.py.
template <int R, int KCH, int EPI>
static hipError_t launch_nr(hipStream_t s, const GemvArgs& a, int threads, int nr) {
    constexpr int ALT = (EPI == EPI_QKV) ? 3 : 2;
    if (nr != 1 && KCH * ALT * (EPI == EPI_SWIGLU ? 2 : 1) <= 24 && R * ALT <= 64) return launch_one<R, KCH, ALT, EPI>(s, a, threads);
    return launch_one<R, KCH, 1, EPI>(s, a, threads);
}

template <int R, int KCH>
static hipError_t launch_epi(hipStream_t s, const GemvArgs& a, int epi, int threads) {
    static const int builtin_nr[4] = {1, 1, 1, 3};  // tools/sweep.py, profiles/r01b_attn_sampler
    const int nr = g_tune.gemv_nr[epi] > 0 ? g_tune.gemv_nr[epi] : builtin_nr[epi];
    switch (epi) {
        case EPI_PLAIN: return launch_nr<R, KCH, EPI_PLAIN>(s, a, threads, nr);
        case EPI_RESID: return launch_nr<R, KCH, EPI_RESID>(s, a, threads, nr);
        case EPI_SWIGLU: return launch_nr<R, KCH, EPI_SWIGLU>(s, a, threads, nr);
        case EPI_QKV: return launch_nr<R, KCH, EPI_QKV>(s, a, threads, nr);
    }
    return hipErrorInvalidValue;
}

template <int R>
static hipError_t launch_kch(hipStream_t s, const GemvArgs& a, int epi, int threads) {
    const int kch = (a.K + 511) / 512;
    switch (kch) {
        case 1: return launch_epi<R, 1>(s, a, epi, threads);
        case 2: return launch_epi<R, 2>(s, a, epi, threads);
        case 4: return launch_epi<R, 4>(s, a, epi, threads);
        case 11: return launch_epi<R, 11>(s, a, epi, threads);
    }
    return hipErrorInvalidValue;
}

// Threads per block: enough waves that (a) the norm prologue covers K with 8 columns per thread and (b) the grid
// lands on ~256 blocks (one per CU) for the MetaVoice-1B shapes: qkv 24 rows/block, wo/w2 8, w1|w3 22, head 12.
int gemv_default_threads(int epi, int N, int K) {
    const int kp8 = ((K + 511) / 512) * 64;  // threads needed by the norm prologue
    int t;
    if (g_tune.gemv_threads[epi] > 0) t = g_tune.gemv_threads[epi];
    else if (epi == EPI_SWIGLU) t = (N % 22 == 0) ? 704 : 512;
    else if (epi == EPI_PLAIN) t = (N > 2048 && N < 3072) ? 768 : 512;
    else t = 512;
    t = (t + 63) / 64 * 64;
    if (t > 1024) t = 1024;
    return t < kp8 ? kp8 : t;
}

bool gemv_supported(int rows, int K) {
    const int kch = (K + 511) / 512;
    return (rows == 2 || rows == 4 || rows == 8) && (kch == 1 || kch == 2 || kch == 4 || kch == 11) && K % 8 == 0 && K >= 8;
}

hipError_t launch_gemv(hipStream_t s, int rows, const GemvArgs& a0, int epi) {
    if (!gemv_supported(rows, a0.K)) return hipErrorInvalidValue;
    GemvArgs a = a0;
    a.ablate = g_tune.gemv_ablate;
    const int threads = gemv_default_threads(epi, a.N, a.K);
    switch (rows) {
        case 2: return launch_kch<2>(s, a, epi, threads);
        case 4: return launch_kch<4>(s, a, epi, threads);
        case 8: return launch_kch<8>(s, a, epi, threads);
    }
    return hipErrorInvalidValue;
}
