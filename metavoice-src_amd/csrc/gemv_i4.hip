// int4 (group-128, asymmetric) weight-only path of the decode step: `quantisation_mode="int4"` of TTS / build_model
// (fam/llm/fast_inference.py:41-50, fam/llm/fast_inference_utils.py:297-306).
//
// Numerics are the reference's (fam/llm/fast_quantize.py):
//   quantiser  :70-124   per 128-column group of a bf16 row, every step in bf16 arithmetic as torch does it:
//                        scale = bf16(max(bf16(max - min), bf16(1e-6)) / 15), zero = bf16(min + 8 scale),
//                        q = clamp(rint(bf16(bf16(w - bf16(zero - 8 scale)) / scale)), 0, 15)
//   matmul     :204-216  y = x @ dequant(W)^T with dequant = bf16((q - 8) * scale + zero), fp32 accumulate, bf16 out --
//                        the documented contract of aten::_weight_int4pack_mm (tinygemm dequantises to bf16 with one
//                        bf16 fma per weight, then runs the bf16 MMA).
// The PACKING is ours, not tinygemm's NVIDIA mma-shaped (N/8, K/128, 32, 4) int32 tiles: rows stay rows, a row is K/2
// bytes, so a wave reads one row's 2048 columns as one contiguous 1 KiB segment (16 B = 32 columns per lane, all in one
// 128-column group), plus 4 bytes of (scale, zero) per lane.  The quantiser writes the ABI layout (column 2j in the low and
// 2j+1 in the high nibble of byte j: what mv_op_quantize_int4 documents and the bit-exactness test reads); the GEMV kernels
// read every 32-bit word PAIR-MAJOR (gemv_i4_core.h), `repack_i4_pairs_kernel` converts once at mv_finalize.
//
// Roofline: HBM by bytes (0.53 B per weight instead of 2).  Round 5: the per-weight VALU work is ~2 lane-ops (OR the nibble
// into a bf16 128.0, dot2; the group's scale / zero applied once per lane and chunk -- gemv_i4_core.h) instead of ~5
// (convert, subtract, fma, bf16 round, dot): until then the int4 decode step was slower than the bf16 one.
#include "gemv_core.h"
#include "kernels.h"
#include "gemv_i4_core.h"

// One wave owns NR rows of W; per 2048-column chunk a lane holds 32 columns (16 B of codes) of each row.
template <int R, int KCH, int NR, int EPI>
__global__ void __launch_bounds__(1024) gemv_i4_kernel(const GemvI4Args a) {
    constexpr int NMAT = (EPI == GEMV_SWIGLU) ? 2 : 1;
    constexpr int KP = KCH * 2048;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    bf16_t* xs = (bf16_t*)smem;                 // [R][KP] bf16, zero beyond K
    float* red = (float*)(smem + R * KP * 2);   // [nwaves][R]

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
    const int K = a.K, N = a.N, G = K / 128;
    const int n0 = (blockIdx.x * nwaves + wave) * NR;

    // ---- 1. codes and (scale, zero) of the wave's rows in flight first -----------------------------------------------
    u32x4 cw[NMAT][NR][KCH];
    uint32_t sz[NMAT][NR][KCH];
#pragma unroll
    for (int i = 0; i < NR; ++i) {
        const int n = min(n0 + i, N - 1);
#pragma unroll
        for (int j = 0; j < KCH; ++j) {
            const int c = min(j * 2048 + lane * 32, K - 32);  // clamped; the x operand is zero beyond K
            cw[0][i][j] = load_stream16(a.codes + ((size_t)n * K + c) / 2);
            sz[0][i][j] = a.sz[(size_t)n * G + c / 128];
            if (NMAT == 2) {
                cw[NMAT - 1][i][j] = load_stream16(a.codes2 + ((size_t)n * K + c) / 2);
                sz[NMAT - 1][i][j] = a.sz2[(size_t)n * G + c / 128];
            }
        }
    }

    bf16_t res_pre = 0;   // epilogue operands fetched up front (see gemv.hip)
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

    // ---- 3. dequantise each word once, dot it with every activation row (gemv_i4_core.h, shared with decode_mega.hip) --------
    float acc[NMAT][NR][R];
#pragma unroll
    for (int m = 0; m < NMAT; ++m) i4_dots<R, KCH, NR>(cw[m], sz[m], xs, KP, K, lane, acc[m]);

    // ---- 4. reductions + epilogue (same as gemv.hip) ------------------------------------------------------------------
    float mine[NMAT];
#pragma unroll
    for (int m = 0; m < NMAT; ++m) mine[m] = 0.f;
#pragma unroll
    for (int m = 0; m < NMAT; ++m)
#pragma unroll
        for (int i = 0; i < NR; ++i)
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const float t = wave_sum(acc[m][i][r]);
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
static hipError_t launch_i4_epi(hipStream_t s, const GemvI4Args& a, int epi) {
    const int kp8 = KCH * 256;  // threads the norm prologue needs (8 columns each)
    int threads = 512;
    if (threads < kp8) threads = kp8;
    const int nwaves = threads / 64;
    const size_t lds = (size_t)R * KCH * 2048 * 2 + (size_t)nwaves * R * 4;
#define MV_I4_LAUNCH(NR, EPI)                                                                                           \
    hipLaunchKernelGGL((gemv_i4_kernel<R, KCH, NR, EPI>), dim3((a.N + nwaves * NR - 1) / (nwaves * NR)), dim3(threads), lds, s, a)
    switch (epi) {
        case GEMV_PLAIN: MV_I4_LAUNCH(1, GEMV_PLAIN); break;
        case GEMV_RESID: MV_I4_LAUNCH(1, GEMV_RESID); break;
        case GEMV_SWIGLU: MV_I4_LAUNCH(1, GEMV_SWIGLU); break;
        case GEMV_QKV: MV_I4_LAUNCH(3, GEMV_QKV); break;
        default: return hipErrorInvalidValue;
    }
#undef MV_I4_LAUNCH
    return hipGetLastError();
}

bool gemv_i4_supported(int rows, int K) {
    const int kch = (K + 2047) / 2048;
    return (rows == 2 || rows == 4) && K % 128 == 0 && (kch == 1 || kch == 3);
}

hipError_t launch_gemv_i4(hipStream_t s, int rows, const GemvI4Args& a, int epi) {
    if (!gemv_i4_supported(rows, a.K)) return hipErrorInvalidValue;
    const int kch = (a.K + 2047) / 2048;
    if (rows == 2) return kch == 1 ? launch_i4_epi<2, 1>(s, a, epi) : launch_i4_epi<2, 3>(s, a, epi);
    return kch == 1 ? launch_i4_epi<4, 1>(s, a, epi) : launch_i4_epi<4, 3>(s, a, epi);
}

// ---- quantiser (fast_quantize.py:70-124), one wave per (row, 2048-column chunk): 4 lanes share a 128-column group --------
__global__ void __launch_bounds__(256) quantize_i4_kernel(const bf16_t* w, int N, int K, unsigned char* codes, uint32_t* sz,
                                                         bf16_t* deq) {
    const int lane = threadIdx.x & 63;
    const int chunks = (K + 2047) / 2048;
    const int gw = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (gw >= N * chunks) return;
    const int n = gw / chunks, c0 = (gw % chunks) * 2048 + lane * 32;
    const bool live = c0 < K;
    const int c = live ? c0 : K - 32;
    float v[32];
    float mx = -INFINITY, mn = INFINITY;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const u32x4 t = load16(w + (size_t)n * K + c + q * 8);
        const uint32_t ws[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            v[q * 8 + 2 * k] = bf_lo(ws[k]);
            v[q * 8 + 2 * k + 1] = bf_hi(ws[k]);
        }
    }
#pragma unroll
    for (int k = 0; k < 32; ++k) { mx = fmaxf(mx, v[k]); mn = fminf(mn, v[k]); }
    mx = fmaxf(mx, dpp_f<0xB1>(mx)); mx = fmaxf(mx, dpp_f<0x4E>(mx));   // over the 4 lanes of the group
    mn = fminf(mn, dpp_f<0xB1>(mn)); mn = fminf(mn, dpp_f<0x4E>(mn));
    // get_group_qparams: every operation on bf16 tensors = fp32 op + round to bf16
    const float scale = round_bf(fmaxf(round_bf(mx - mn), round_bf(1e-6f)) / 15.0f);
    const float zero = round_bf(mn + scale * 8.0f);
    // group_quantize_tensor_from_qparams
    const float min_val = round_bf(zero - scale * 8.0f);
    uint32_t packed[4] = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int k = 0; k < 32; ++k) {
        const float t = rintf(round_bf(round_bf(v[k] - min_val) / scale));
        const uint32_t q = (uint32_t)fminf(fmaxf(t, 0.f), 15.f);
        packed[k >> 3] |= q << (4 * (k & 7));
        if (deq != nullptr && live) deq[(size_t)n * K + c + k] = f2bf(fmaf((float)q - 8.f, scale, zero));
    }
    if (live) {
        *(u32x4*)(codes + ((size_t)n * K + c) / 2) = u32x4{packed[0], packed[1], packed[2], packed[3]};
        if ((lane & 3) == 0) sz[(size_t)n * (K / 128) + c / 128] = (uint32_t)f2bf(scale) | ((uint32_t)f2bf(zero) << 16);
    }
}

__global__ void __launch_bounds__(256) repack_i4_pairs_kernel(const uint32_t* src, uint32_t* dst, size_t n_words) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n_words) dst[i] = i4_word_to_pairs(src[i]);
}
hipError_t launch_repack_i4_pairs(hipStream_t s, const unsigned char* src, unsigned char* dst, size_t n_bytes) {
    if (n_bytes % 4 != 0) return hipErrorInvalidValue;
    const size_t n_words = n_bytes / 4;
    if (n_words == 0) return hipSuccess;
    hipLaunchKernelGGL(repack_i4_pairs_kernel, dim3((unsigned)((n_words + 255) / 256)), dim3(256), 0, s, (const uint32_t*)src, (uint32_t*)dst, n_words);
    return hipGetLastError();
}

hipError_t launch_quantize_i4(hipStream_t s, const bf16_t* w, int N, int K, unsigned char* codes, uint32_t* sz, bf16_t* deq) {
    if (K % 128 != 0 || K < 128) return hipErrorInvalidValue;
    const int waves = N * ((K + 2047) / 2048);
    hipLaunchKernelGGL(quantize_i4_kernel, dim3((waves + 3) / 4), dim3(256), 0, s, w, N, K, codes, sz, deq);
    return hipGetLastError();
}
