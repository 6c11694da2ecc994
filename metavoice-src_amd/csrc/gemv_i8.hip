// int8 (per output channel, symmetric) weight-only path: `quantisation_mode="int8"` of TTS / build_model
// (fam/llm/fast_inference.py:41-50, fam/llm/fast_inference_utils.py:283-296).
//
// Numerics are the reference's (fam/llm/fast_quantize.py):
//   quantiser  :38-67,162-178  on the fp32 copy of the weight: scale = clamp(max|w| / 127.5, eps) (fp32),
//                              q = clamp(rint(w / scale), -128, 127); the scale is STORED as bf16
//   matmul     :197-198        F.linear(x, W.to(bf16)) * scales  =  bf16( bf16(x . q^T) * scale_bf16 )
// EVERY nn.Linear is replaced, the 2562-row LM head and speaker_cond_pos included (:137-160).  The reference documents this
// mode as slower than bf16 (README.md:92); here it streams half the bytes through the same launch-bound step.
// Layout: [N][K] int8 row-major as quantised; a lane reads 16 B = 16 columns per load, 1024 columns per wave instruction.
#include "gemv_core.h"
#include "kernels.h"
#include "gemv_i8_core.h"

template <int R, int KCH, int NR, int EPI>
__global__ void __launch_bounds__(1024) gemv_i8_kernel(const GemvI8Args a) {
    constexpr int NMAT = (EPI == GEMV_SWIGLU) ? 2 : 1;
    constexpr int KP = KCH * 1024;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    bf16_t* xs = (bf16_t*)smem;                 // [R][KP] bf16, zero beyond K
    float* red = (float*)(smem + R * KP * 2);   // [nwaves][R]

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
    const int K = a.K, N = a.N;
    const int n0 = (blockIdx.x * nwaves + wave) * NR;

    u32x4 cw[NMAT][NR][KCH];
    float sc[NMAT][NR];
#pragma unroll
    for (int i = 0; i < NR; ++i) {
        const int n = min(n0 + i, N - 1);
#pragma unroll
        for (int j = 0; j < KCH; ++j) {
            const int c = min(j * 1024 + lane * 16, K - 16);
            cw[0][i][j] = load_stream16(a.q + (size_t)n * K + c);
            if (NMAT == 2) cw[NMAT - 1][i][j] = load_stream16(a.q2 + (size_t)n * K + c);
        }
        sc[0][i] = bf2f(a.scales[n]);
        if (NMAT == 2) sc[NMAT - 1][i] = bf2f(a.scales2[n]);
    }
    bf16_t res_pre = 0;
    int pos_pre = 0;
    {
        const int i = lane / R, r = lane % R, n = n0 + i;
        if (lane < NR * R && n < N) {
            if (EPI == GEMV_RESID) res_pre = a.residual[(size_t)r * N + n];
            if (EPI == GEMV_QKV) pos_pre = a.pos[r >> 1];
        }
    }
    // ---- 2. activation prologue -> LDS (identical to gemv.hip: thread t owns columns 8t..8t+7) ---------------------------
    if (a.norm_w != nullptr) {
        const int c = threadIdx.x * 8;
        const bool in_k = c < K;
        u32x4 xv[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            xv[r] = in_k ? load16(a.x + (size_t)r * K + c) : u32x4{0, 0, 0, 0};
            const float s = wave_sum(sumsq8(xv[r]));
            if (lane == 0) red[wave * R + r] = s;
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

    // ---- 3. convert each word once, dot it with every activation row (gemv_i8_core.h, shared with decode_mega.hip) ------------
    float acc[NMAT][NR][R];
#pragma unroll
    for (int m = 0; m < NMAT; ++m) i8_dots<R, KCH, NR>(cw[m], xs, KP, K, lane, acc[m]);

    // ---- 4. reductions; y = bf16(bf16(acc) * scale) is the Linear output, then the usual epilogues -----------------------
    float mine[NMAT];
#pragma unroll
    for (int m = 0; m < NMAT; ++m) mine[m] = 0.f;
#pragma unroll
    for (int m = 0; m < NMAT; ++m)
#pragma unroll
        for (int i = 0; i < NR; ++i)
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const float t = round_bf(round_bf(wave_sum(acc[m][i][r])) * sc[m][i]);
                if (lane == i * R + r) mine[m] = t;
            }
    if (lane < NR * R) {
        const int i = lane / R, r = lane % R, n = n0 + i;
        if (n < N) {
            if (EPI == GEMV_PLAIN) {
                a.out[(size_t)r * N + n] = f2bf(mine[0]);
            } else if (EPI == GEMV_RESID) {
                a.out[(size_t)r * N + n] = epi_resid(res_pre, mine[0]);
            } else if (EPI == GEMV_SWIGLU) {
                a.out[(size_t)r * N + n] = epi_swiglu(mine[0], mine[NMAT - 1]);
            } else {  // QKV
                const bf16_t v = f2bf(mine[0]);
                const int dim = a.dim;
                if (n < dim) {
                    a.out[(size_t)r * dim + n] = v;
                } else {
                    const int which = (n - dim) / dim, hd = (n - dim) % dim;
                    bf16_t* cache = which ? a.vcache : a.kcache;
                    cache[(((size_t)r * a.n_head + hd / 128) * a.cache_len + pos_pre) * 128 + hd % 128] = v;
                }
            }
        }
    }
}

template <int R, int KCH>
static hipError_t launch_i8_epi(hipStream_t s, const GemvI8Args& a, int epi) {
    const int kp8 = KCH * 128;  // threads the norm prologue needs (8 columns each)
    int threads = epi == GEMV_PLAIN ? 768 : 512;
    if (threads < kp8) threads = kp8;
    const int nwaves = threads / 64;
    const size_t lds = (size_t)R * KCH * 1024 * 2 + (size_t)nwaves * R * 4;
#define MV_I8_LAUNCH(NR, EPI)                                                                                           \
    hipLaunchKernelGGL((gemv_i8_kernel<R, KCH, NR, EPI>), dim3((a.N + nwaves * NR - 1) / (nwaves * NR)), dim3(threads), lds, s, a)
    switch (epi) {
        case GEMV_PLAIN: MV_I8_LAUNCH(1, GEMV_PLAIN); break;
        case GEMV_RESID: MV_I8_LAUNCH(1, GEMV_RESID); break;
        case GEMV_SWIGLU: MV_I8_LAUNCH(1, GEMV_SWIGLU); break;
        case GEMV_QKV: MV_I8_LAUNCH(3, GEMV_QKV); break;
        default: return hipErrorInvalidValue;
    }
#undef MV_I8_LAUNCH
    return hipGetLastError();
}

bool gemv_i8_supported(int rows, int K) {
    const int kch = (K + 1023) / 1024;
    return (rows == 2 || rows == 4) && K % 16 == 0 && (kch == 1 || kch == 2 || kch == 6);
}

hipError_t launch_gemv_i8(hipStream_t s, int rows, const GemvI8Args& a, int epi) {
    if (!gemv_i8_supported(rows, a.K)) return hipErrorInvalidValue;
    const int kch = (a.K + 1023) / 1024;
#define MV_I8_KCH(R) (kch == 1 ? launch_i8_epi<R, 1>(s, a, epi) : kch == 2 ? launch_i8_epi<R, 2>(s, a, epi) : launch_i8_epi<R, 6>(s, a, epi))
    return rows == 2 ? MV_I8_KCH(2) : MV_I8_KCH(4);
#undef MV_I8_KCH
}

// ---- quantiser (fast_quantize.py:38-67 on weight.float()): one block per row -----------------------------------------------
__global__ void __launch_bounds__(256) quantize_i8_kernel(const bf16_t* w, int N, int K, signed char* q, bf16_t* scales) {
    __shared__ float red[4];
    const int n = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float mx = 0.f;  // max(|min(min,0)|, max(max,0)) == max |w|
    for (int c = threadIdx.x; c < K; c += 256) mx = fmaxf(mx, fabsf(bf2f(w[(size_t)n * K + c])));
    mx = wave_max(mx);
    if (lane == 0) red[wave] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    const float scale = fmaxf(mx / 127.5f, 1.1920928955078125e-07f);  // torch.finfo(float32).eps
    for (int c = threadIdx.x; c < K; c += 256) {
        const float t = rintf(bf2f(w[(size_t)n * K + c]) / scale);
        q[(size_t)n * K + c] = (signed char)fminf(fmaxf(t, -128.f), 127.f);
    }
    if (threadIdx.x == 0) scales[n] = f2bf(scale);
}

hipError_t launch_quantize_i8(hipStream_t s, const bf16_t* w, int N, int K, signed char* q, bf16_t* scales) {
    hipLaunchKernelGGL(quantize_i8_kernel, dim3(N), dim3(256), 0, s, w, N, K, q, scales);
    return hipGetLastError();
}
