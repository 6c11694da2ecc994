// EnCodec (24 kHz) token -> waveform decode: residual-VQ codebook lookup + the SEANet decoder, fp32.
//
// In the reference this arithmetic is third party: `fam/llm/decoders.py:13,84-85` calls audiocraft's
// `MultiBandDiffusion.tokens_to_wav`, which decodes the tokens with the EnCodec 24 kHz codec (`encodec==0.1.1`,
// `audiocraft==1.2.0`; neither is vendored under the reference nor installed here) -- as the conditioning latent of the
// diffusion bands and as the reference signal of its final EQ.  The architecture restated here is the published one
// (Defossez et al. 2022, "High Fidelity Neural Audio Compression"; identical module graph in
// transformers.models.encodec, which IS installed and is what the tests pin against):
//   RVQ decode      sum over the n_q codebooks of embed_q[token]                       -> (128, T)
//   Conv1d 128->512 k7 | LSTM x2 (512) + skip | 4 x { ELU, ConvTranspose1d C->C/2 (k = 2r, stride r, r = 8,5,4,2),
//   ResBlock(C/2): ELU, Conv k3 C/2->C/4, ELU, Conv k1 C/4->C/2, + Conv k1 shortcut } | ELU | Conv1d 32->1 k7
//   causal convolutions, reflect padding on the left, transposed convolutions trimmed on the right, weight-norm folded
//   into the weights on the host.
// The reference forces fp32 here (`torch.amp.autocast(dtype=float32)`, decoders.py:84): these kernels are fp32 VALU code.
//
// Layout: activations [C][T] (time contiguous: every load of a wave is one coalesced segment); weights re-laid out as
// [Cin][K][Cout] so that the CO_T output channels a block accumulates are contiguous and WAVE-UNIFORM: they come through
// the scalar cache (s_load) and feed v_fmac as SGPR operands -- no LDS staging, one vector load per (ci, k) for CO_T FMAs.
// ~15 GMAC for 12.8 s of audio; the two LSTM layers are a chain of 2 x T dependent steps (one small launch each).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "common.h"
#include "kernels.h"

int mv_fail(int code, const char* fmt, ...);  // engine.hip

#define CD_TRY(expr)                                                                                         \
    do {                                                                                                     \
        hipError_t _e = (expr);                                                                              \
        if (_e != hipSuccess) return mv_fail(MV_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), \
                                             __FILE__, __LINE__);                                            \
    } while (0)

namespace {

constexpr int CO_T = 8;

__device__ __forceinline__ float elu1(float x) { return x > 0.f ? x : expm1f(x); }

__global__ void elu_kernel(const float* __restrict__ in, float* __restrict__ out, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = elu1(in[i]);
}

// out[c][t] = sum_q embed[q][tok[q][t]][c]      (EncodecResidualVectorQuantizer.decode)
__global__ void rvq_decode_kernel(const int32_t* __restrict__ tok, const float* __restrict__ embed, float* __restrict__ out, int nq,
                                  int bins, int dim, int T) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x, c = blockIdx.y;
    if (t >= T) return;
    float s = 0.f;
    for (int q = 0; q < nq; ++q) s += embed[((size_t)q * bins + tok[(size_t)q * T + t]) * dim + c];
    out[(size_t)c * T + t] = s;
}

// Causal Conv1d, stride 1: out[co][t] = b[co] + sum_ci sum_k w[ci][k][co] * in[ci][reflect(t + k*dil - (K-1)*dil)] (+ out)
struct ConvArgs {
    const float* in;
    const float* w;     // [Cin][K][Cout_padded]
    const float* bias;  // [Cout_padded]
    float* out;
    int Cin, Cout, CoutP, K, dil, T, accumulate;
};
__global__ void __launch_bounds__(256) conv1d_kernel(const ConvArgs a) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    const int co0 = blockIdx.y * CO_T;
    const int tt = min(t, a.T - 1);
    const int P = (a.K - 1) * a.dil;
    float acc[CO_T];
#pragma unroll
    for (int j = 0; j < CO_T; ++j) acc[j] = a.bias[co0 + j];
    for (int ci = 0; ci < a.Cin; ++ci) {
        const float* row = a.in + (size_t)ci * a.T;
        const float* wr = a.w + ((size_t)ci * a.K) * a.CoutP + co0;
        for (int k = 0; k < a.K; ++k) {
            int i = tt + k * a.dil - P;
            i = i < 0 ? -i : i;  // reflect (no edge repeat); T > P is checked on the host
            const float x = row[i];
#pragma unroll
            for (int j = 0; j < CO_T; ++j) acc[j] = fmaf(wr[(size_t)k * a.CoutP + j], x, acc[j]);
        }
    }
    if (t < a.T) {
#pragma unroll
        for (int j = 0; j < CO_T; ++j) {
            if (co0 + j < a.Cout) {
                float* o = a.out + (size_t)(co0 + j) * a.T + t;
                *o = a.accumulate ? *o + acc[j] : acc[j];
            }
        }
    }
}

// Causal ConvTranspose1d, kernel 2s, stride s, right-trimmed by s: output j = i*s + p gets exactly two taps,
//   out[co][i*s+p] = b[co] + sum_ci ( w[ci][p][co] * in[ci][i] + w[ci][p+s][co] * in[ci][i-1] ),  in[.][-1] = 0
template <int S>
__global__ void __launch_bounds__(256) convt1d_kernel(const ConvArgs a) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int co0 = blockIdx.y * CO_T;
    const int ii = min(i, a.T - 1);
    float acc[S][CO_T];
#pragma unroll
    for (int p = 0; p < S; ++p)
#pragma unroll
        for (int j = 0; j < CO_T; ++j) acc[p][j] = a.bias[co0 + j];
    for (int ci = 0; ci < a.Cin; ++ci) {
        const float* row = a.in + (size_t)ci * a.T;
        const float x0 = row[ii], x1 = ii > 0 ? row[ii - 1] : 0.f;
        const float* wr = a.w + ((size_t)ci * 2 * S) * a.CoutP + co0;
#pragma unroll
        for (int p = 0; p < S; ++p)
#pragma unroll
            for (int j = 0; j < CO_T; ++j)
                acc[p][j] = fmaf(wr[(size_t)(p + S) * a.CoutP + j], x1, fmaf(wr[(size_t)p * a.CoutP + j], x0, acc[p][j]));
    }
    if (i < a.T) {
#pragma unroll
        for (int j = 0; j < CO_T; ++j)
            if (co0 + j < a.Cout) {
                float* o = a.out + (size_t)(co0 + j) * a.T * S + (size_t)i * S;
#pragma unroll
                for (int p = 0; p < S; ++p) o[p] = acc[p][j];
            }
    }
}

// ---- the same convolutions on the fp32 matrix cores (round 4) ---------------------------------------------------------------------------
// A Conv1d is the GEMM out[co][t] = sum_kk W[kk][co] X[kk][t] with kk = (ci, k) flattened and X[kk][t] = in[ci][reflect(t + k dil - P)]: on
// v_mfma_f32_32x32x2_f32 (exact fp32 products, fp32 accumulation, 64 FLOP per cycle and SIMD against the VALU's 8) lane l supplies
// A = W[kk + (l >> 5)][co0 + (l & 31)] -- the [Cin][K][CoutP] layout makes that one coalesced 256-byte load per pair of kk -- and
// B = X[kk + (l >> 5)][t0 + (l & 31)], a 128-byte run of one input row per half-wave.  A wave owns CT x TT tiles of 32 x 32 (CT TT = 4: 64
// accumulator registers); the loads of the next PF pairs are in flight under the current pair's four MFMAs.  The VALU kernels above ran the
// decoder's ~25 GFLOP at 9 TFLOP/s (3.2 ms per utterance, profiles/r03_final); MV_CODEC_VALU=1 keeps them (A/B, tests).
typedef float cd_f32x16_t __attribute__((ext_vector_type(16)));
template <int CT, int TT>
__global__ void __launch_bounds__(256) conv1d_mfma_kernel(const ConvArgs a) {
    constexpr int PF = 4;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, col = lane & 31, kh = lane >> 5;
    const int t0 = (blockIdx.x * 4 + wave) * (32 * TT), co0 = blockIdx.y * (32 * CT);
    const int P = (a.K - 1) * a.dil, npairs = a.Cin * a.K / 2;
    cd_f32x16_t acc[CT][TT];
#pragma unroll
    for (int c = 0; c < CT; ++c)
#pragma unroll
        for (int q = 0; q < TT; ++q)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[c][q][i] = 0.f;
    if (t0 >= a.T) return;
    float ra[PF][CT], rb[PF][TT];
    auto fetch = [&](int p, float (&fa)[CT], float (&fb)[TT]) {
        const int kk = min(2 * p + kh, a.Cin * a.K - 1), ci = kk / a.K, k = kk - ci * a.K;
        const float* wr = a.w + (size_t)kk * a.CoutP + co0 + col;
#pragma unroll
        for (int c = 0; c < CT; ++c) fa[c] = co0 + 32 * c < a.CoutP ? wr[32 * c] : 0.f;
        const float* row = a.in + (size_t)ci * a.T;
#pragma unroll
        for (int q = 0; q < TT; ++q) {
            int i = min(t0 + 32 * q + col, a.T - 1) + k * a.dil - P;
            i = i < 0 ? -i : i;   // reflect (no edge repeat); T > P is checked on the host
            fb[q] = row[i];
        }
    };
#pragma unroll
    for (int i = 0; i < PF; ++i) fetch(min(i, npairs - 1), ra[i], rb[i]);
    for (int p0 = 0; p0 < npairs; p0 += PF) {
#pragma unroll
        for (int i = 0; i < PF; ++i) {
            const int p = p0 + i;
            float fa[CT], fb[TT];
#pragma unroll
            for (int c = 0; c < CT; ++c) fa[c] = ra[i][c];
#pragma unroll
            for (int q = 0; q < TT; ++q) fb[q] = rb[i][q];
            fetch(min(p + PF, npairs - 1), ra[i], rb[i]);
            if (p < npairs) {
#pragma unroll
                for (int c = 0; c < CT; ++c)
#pragma unroll
                    for (int q = 0; q < TT; ++q) acc[c][q] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[c], fb[q], acc[c][q], 0, 0, 0);
            }
        }
    }
    // C layout of a 32 x 32 tile: lane -> column l & 31 (time); register i -> row 8 (i >> 2) + 4 (l >> 5) + (i & 3) (output channel)
#pragma unroll
    for (int c = 0; c < CT; ++c)
#pragma unroll
        for (int q = 0; q < TT; ++q) {
            const int t = t0 + 32 * q + col;
            if (t >= a.T) continue;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int co = co0 + 32 * c + 8 * (i >> 2) + 4 * kh + (i & 3);
                if (co < a.Cout) {
                    float* o = a.out + (size_t)co * a.T + t;
                    const float y = acc[c][q][i] + a.bias[co];
                    *o = a.accumulate ? *o + y : y;
                }
            }
        }
}
// ConvTranspose1d (kernel 2 S, stride S, right-trimmed): per output phase p a GEMM over kk = (ci, j), j = 0: tap p on in[ci][i], j = 1: tap p + S
// on in[ci][i - 1].  A wave owns one 32-channel tile x 32 input steps and ALL S phases (S accumulators): the B operand (lane l: in[ci][i0 +
// (l & 31) - (l >> 5)]) is shared by the S MFMAs of a channel pair; a lane ends up with the S consecutive output samples of its input step.
template <int S>
__global__ void __launch_bounds__(256) convt1d_mfma_kernel(const ConvArgs a) {
    constexpr int PF = 2;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, col = lane & 31, kh = lane >> 5;
    const int i0 = (blockIdx.x * 4 + wave) * 32, co0 = blockIdx.y * 32;
    cd_f32x16_t acc[S];
#pragma unroll
    for (int p = 0; p < S; ++p)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[p][i] = 0.f;
    if (i0 >= a.T) return;
    const int ii = min(i0 + col, a.T - 1) - kh;   // kh = 1: the previous input step (zero in front of the first one)
    float ra[PF][S], rb[PF];
    auto fetch = [&](int ci, float (&fa)[S], float& fb) {
        const float* wr = a.w + ((size_t)ci * 2 * S + (size_t)kh * S) * a.CoutP + co0 + col;
#pragma unroll
        for (int p = 0; p < S; ++p) fa[p] = wr[(size_t)p * a.CoutP];
        fb = ii >= 0 ? a.in[(size_t)ci * a.T + ii] : 0.f;
    };
#pragma unroll
    for (int i = 0; i < PF; ++i) fetch(min(i, a.Cin - 1), ra[i], rb[i]);
    for (int c0 = 0; c0 < a.Cin; c0 += PF) {
#pragma unroll
        for (int i = 0; i < PF; ++i) {
            const int ci = c0 + i;
            float fa[S];
#pragma unroll
            for (int p = 0; p < S; ++p) fa[p] = ra[i][p];
            const float fb = rb[i];
            fetch(min(ci + PF, a.Cin - 1), ra[i], rb[i]);
            if (ci < a.Cin) {
#pragma unroll
                for (int p = 0; p < S; ++p) acc[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[p], fb, acc[p], 0, 0, 0);
            }
        }
    }
    const int step = i0 + col;
    if (step >= a.T) return;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int co = co0 + 8 * (i >> 2) + 4 * kh + (i & 3);
        if (co >= a.Cout) continue;
        float* o = a.out + (size_t)co * a.T * S + (size_t)step * S;
        const float bco = a.bias[co];
#pragma unroll
        for (int p = 0; p < S; ++p) o[p] = acc[p][i] + bco;
    }
}

struct Conv {
    float *w = nullptr, *b = nullptr;  // device, re-laid out
    int cin = 0, cout = 0, coutp = 0, k = 0, dil = 1, stride = 1;
    bool transposed = false, loaded_w = false, loaded_b = false;
};

}  // namespace

struct mv_codec {
    mv_codec_config cfg{};
    hipStream_t stream = nullptr;
    std::vector<void*> allocs;
    unsigned long long* lstm_hx = nullptr;  // [2][dim0] tagged granules of the persistent LSTM kernel (lstm.hip)
    uint32_t* lstm_abort = nullptr;
    uint32_t lstm_serial = 0;
    float* embed = nullptr;  // [n_q][bins][dim]
    std::vector<bool> embed_loaded;
    Conv conv_in, conv_out;
    std::vector<Conv> up, res1, res2, shortcut;  // per upsampling stage
    // LSTM: per layer W_ih as a k=1 conv (bias = b_ih + b_hh), W_hh
    std::vector<Conv> lstm_ih;
    std::vector<float*> lstm_hh;
    std::vector<std::vector<float>> lstm_bias_host;  // b_ih + b_hh accumulated on the host
    std::vector<int> lstm_loaded;                    // bitmask: 1 w_ih, 2 w_hh, 4 b_ih, 8 b_hh
    bool finalized = false;
    int dim0 = 0;  // channels after conv_in (num_filters * 2^n_ratios)
    // per-call workspace, kept between calls (grown on demand): token ids, four activation buffers, the latent of the last call
    // [6]: the [T][H] granule sequence layer 1 of the fused two-layer LSTM launch leaves for layer 2
    void* ws[7] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    size_t ws_cap[7] = {0, 0, 0, 0, 0, 0, 0};
};

namespace {

int alloc_conv(mv_codec* c, Conv& cv, int cin, int cout, int k, int dil, bool transposed, int stride) {
    cv.cin = cin; cv.cout = cout; cv.k = k; cv.dil = dil; cv.transposed = transposed; cv.stride = stride;
    cv.coutp = (cout + 31) / 32 * 32;   // zero-padded to the 32-row tiles of the matrix-core kernels (a multiple of the VALU kernels' CO_T too)
    CD_TRY(hipMalloc((void**)&cv.w, (size_t)cin * k * cv.coutp * 4));
    CD_TRY(hipMalloc((void**)&cv.b, (size_t)cv.coutp * 4));
    c->allocs.push_back(cv.w);
    c->allocs.push_back(cv.b);
    CD_TRY(hipMemset(cv.w, 0, (size_t)cin * k * cv.coutp * 4));
    CD_TRY(hipMemset(cv.b, 0, (size_t)cv.coutp * 4));
    return MV_OK;
}

// host re-layout: Conv1d weight [Cout][Cin][K] / ConvTranspose1d weight [Cin][Cout][K] -> [Cin][K][CoutP]
int upload_conv_weight(Conv& cv, const float* w) {
    std::vector<float> t((size_t)cv.cin * cv.k * cv.coutp, 0.f);
    for (int co = 0; co < cv.cout; ++co)
        for (int ci = 0; ci < cv.cin; ++ci)
            for (int k = 0; k < cv.k; ++k) {
                const float v = cv.transposed ? w[((size_t)ci * cv.cout + co) * cv.k + k] : w[((size_t)co * cv.cin + ci) * cv.k + k];
                t[((size_t)ci * cv.k + k) * cv.coutp + co] = v;
            }
    CD_TRY(hipMemcpy(cv.w, t.data(), t.size() * 4, hipMemcpyHostToDevice));
    cv.loaded_w = true;
    return MV_OK;
}

hipError_t run_conv(hipStream_t s, const Conv& cv, const float* in, float* out, int T, bool accumulate) {
    ConvArgs a{in, cv.w, cv.b, out, cv.cin, cv.cout, cv.coutp, cv.k, cv.dil, T, accumulate ? 1 : 0};
    if ((cv.cin * cv.k) % 2 == 0) {   // fp32 matrix cores (an odd reduction length -- no such layer in EnCodec 24 kHz -- takes the VALU kernels below)
        if (!cv.transposed) {
            if (cv.coutp >= 64) hipLaunchKernelGGL((conv1d_mfma_kernel<2, 2>), dim3((T + 255) / 256, (cv.coutp + 63) / 64), dim3(256), 0, s, a);
            else hipLaunchKernelGGL((conv1d_mfma_kernel<1, 4>), dim3((T + 511) / 512, cv.coutp / 32), dim3(256), 0, s, a);
            return hipGetLastError();
        }
        const dim3 gt((T + 127) / 128, cv.coutp / 32);
        switch (cv.stride) {
            case 2: hipLaunchKernelGGL(convt1d_mfma_kernel<2>, gt, dim3(256), 0, s, a); return hipGetLastError();
            case 4: hipLaunchKernelGGL(convt1d_mfma_kernel<4>, gt, dim3(256), 0, s, a); return hipGetLastError();
            case 5: hipLaunchKernelGGL(convt1d_mfma_kernel<5>, gt, dim3(256), 0, s, a); return hipGetLastError();
            case 8: hipLaunchKernelGGL(convt1d_mfma_kernel<8>, gt, dim3(256), 0, s, a); return hipGetLastError();
            default: break;
        }
    }
    const dim3 grid((T + 255) / 256, cv.coutp / CO_T);
    if (!cv.transposed) {
        hipLaunchKernelGGL(conv1d_kernel, grid, dim3(256), 0, s, a);
    } else {
        switch (cv.stride) {
            case 2: hipLaunchKernelGGL(convt1d_kernel<2>, grid, dim3(256), 0, s, a); break;
            case 4: hipLaunchKernelGGL(convt1d_kernel<4>, grid, dim3(256), 0, s, a); break;
            case 5: hipLaunchKernelGGL(convt1d_kernel<5>, grid, dim3(256), 0, s, a); break;
            case 8: hipLaunchKernelGGL(convt1d_kernel<8>, grid, dim3(256), 0, s, a); break;
            default: return hipErrorInvalidValue;
        }
    }
    return hipGetLastError();
}

}  // namespace

extern "C" {

int mv_codec_create(const mv_codec_config* cfg, mv_codec** out) {
    if (!cfg || !out) return mv_fail(MV_ERR_INVALID, "null argument");
    const mv_codec_config& g = *cfg;
    if (g.n_ratios < 1 || g.n_ratios > 8 || g.n_q < 1 || g.n_q > 32 || g.num_filters < 1 || g.lstm_layers < 0 || g.lstm_layers > 4)
        return mv_fail(MV_ERR_INVALID, "bad codec configuration");
    for (int i = 0; i < g.n_ratios; ++i)
        if (g.ratios[i] != 2 && g.ratios[i] != 4 && g.ratios[i] != 5 && g.ratios[i] != 8)
            return mv_fail(MV_ERR_INVALID, "upsampling ratio %d not supported (2, 4, 5, 8)", g.ratios[i]);
    mv_codec* c = new mv_codec();
    c->cfg = g;
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { delete c; return mv_fail(MV_ERR_HIP, "stream creation failed"); }
    int ch = g.num_filters << g.n_ratios;
    c->dim0 = ch;
    if (ch % 4 != 0) { delete c; return mv_fail(MV_ERR_INVALID, "LSTM width %d must be a multiple of 4", ch); }
    int rc = alloc_conv(c, c->conv_in, g.hidden, ch, g.kernel_size, 1, false, 1);
    if (rc == MV_OK && hipMalloc((void**)&c->embed, (size_t)g.n_q * g.bins * g.hidden * 4) != hipSuccess) rc = mv_fail(MV_ERR_HIP, "hipMalloc failed");
    if (rc == MV_OK) c->allocs.push_back(c->embed);
    if (rc == MV_OK && g.lstm_layers > 0) {
        const int H0 = g.num_filters << g.n_ratios;
        if (H0 % 8 != 0 || lstm_seq_lds_bytes(1, H0) > 160 * 1024 - 256) rc = mv_fail(MV_ERR_INVALID, "LSTM width %d unsupported", H0);
        if (rc == MV_OK && (hipMalloc((void**)&c->lstm_hx, (size_t)4 * H0 * 8) != hipSuccess || hipMemset(c->lstm_hx, 0, (size_t)4 * H0 * 8) != hipSuccess ||
                            hipMalloc((void**)&c->lstm_abort, 4) != hipSuccess || hipMemset(c->lstm_abort, 0, 4) != hipSuccess))
            rc = mv_fail(MV_ERR_HIP, "hipMalloc failed");
        if (c->lstm_hx) c->allocs.push_back(c->lstm_hx);
        if (c->lstm_abort) c->allocs.push_back(c->lstm_abort);
    }
    c->embed_loaded.assign(g.n_q, false);
    c->lstm_ih.resize(g.lstm_layers);
    c->lstm_hh.assign(g.lstm_layers, nullptr);
    c->lstm_bias_host.assign(g.lstm_layers, std::vector<float>());
    c->lstm_loaded.assign(g.lstm_layers, 0);
    for (int l = 0; l < g.lstm_layers && rc == MV_OK; ++l) {
        rc = alloc_conv(c, c->lstm_ih[l], ch, 4 * ch, 1, 1, false, 1);
        if (rc == MV_OK && hipMalloc((void**)&c->lstm_hh[l], (size_t)4 * ch * ch * 4) != hipSuccess) rc = mv_fail(MV_ERR_HIP, "hipMalloc failed");
        if (rc == MV_OK) c->allocs.push_back(c->lstm_hh[l]);
        c->lstm_bias_host[l].assign((size_t)4 * ch, 0.f);
    }
    c->up.resize(g.n_ratios); c->res1.resize(g.n_ratios); c->res2.resize(g.n_ratios); c->shortcut.resize(g.n_ratios);
    for (int i = 0; i < g.n_ratios && rc == MV_OK; ++i) {
        const int r = g.ratios[i], half = ch / 2;
        rc = alloc_conv(c, c->up[i], ch, half, 2 * r, 1, true, r);
        if (rc == MV_OK) rc = alloc_conv(c, c->res1[i], half, half / g.compress, g.residual_kernel_size, 1, false, 1);
        if (rc == MV_OK) rc = alloc_conv(c, c->res2[i], half / g.compress, half, 1, 1, false, 1);
        if (rc == MV_OK) rc = alloc_conv(c, c->shortcut[i], half, half, 1, 1, false, 1);
        ch = half;
    }
    if (rc == MV_OK) rc = alloc_conv(c, c->conv_out, ch, g.audio_channels, g.last_kernel_size, 1, false, 1);
    if (rc != MV_OK) { mv_codec_destroy(c); return rc; }
    *out = c;
    return MV_OK;
}

int mv_codec_destroy(mv_codec* c) {
    if (!c) return MV_OK;
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    for (void* p : c->allocs) (void)hipFree(p);
    for (void* p : c->ws) if (p) (void)hipFree(p);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return MV_OK;
}

// Names follow transformers' EncodecModel state dict with weight-norm already folded ("....conv.weight"):
//   quantizer.layers.{q}.codebook.embed | decoder.layers.{i}.conv.{weight,bias} | decoder.layers.1.lstm.{weight,bias}_{ih,hh}_l{l}
//   decoder.layers.{i}.block.{1,3}.conv.{weight,bias} | decoder.layers.{i}.shortcut.conv.{weight,bias}
int mv_codec_load_weight(mv_codec* c, const char* name, const float* data, int64_t numel) {
    if (!c || !name || !data) return mv_fail(MV_ERR_INVALID, "null argument");
    const mv_codec_config& g = c->cfg;
    const std::string n = name;
    auto expect = [&](int64_t want) -> int { return numel == want ? MV_OK : mv_fail(MV_ERR_INVALID, "%s: %lld elements, expected %lld", name, (long long)numel, (long long)want); };
    auto conv_param = [&](Conv& cv, const std::string& leaf) -> int {
        if (leaf == "weight") {
            const int rc = expect((int64_t)cv.cin * cv.cout * cv.k);
            return rc != MV_OK ? rc : upload_conv_weight(cv, data);
        }
        const int rc = expect(cv.cout);
        if (rc != MV_OK) return rc;
        CD_TRY(hipMemcpy(cv.b, data, (size_t)cv.cout * 4, hipMemcpyHostToDevice));
        cv.loaded_b = true;
        return MV_OK;
    };
    int q = 0, i = 0, l = 0, blk = 0;
    char leaf[32], kind[8];
    if (sscanf(name, "quantizer.layers.%d.codebook.embed", &q) == 1 && n.find("embed") != std::string::npos && n.find("embed_avg") == std::string::npos) {
        if (q < 0 || q >= g.n_q) return MV_OK;  // more quantizers in the checkpoint than the 6 kbps decode uses
        const int rc = expect((int64_t)g.bins * g.hidden);
        if (rc != MV_OK) return rc;
        CD_TRY(hipMemcpy(c->embed + (size_t)q * g.bins * g.hidden, data, (size_t)numel * 4, hipMemcpyHostToDevice));
        c->embed_loaded[q] = true;
        return MV_OK;
    }
    if (sscanf(name, "decoder.layers.1.lstm.%7[a-z]_%2[a-z]_l%d", kind, leaf, &l) == 3) {
        if (l < 0 || l >= g.lstm_layers) return mv_fail(MV_ERR_INVALID, "%s: LSTM layer out of range", name);
        const int H = c->dim0;
        const bool ih = strcmp(leaf, "ih") == 0;
        if (strcmp(kind, "weight") == 0) {
            const int rc = expect((int64_t)4 * H * H);
            if (rc != MV_OK) return rc;
            if (ih) { const int r2 = upload_conv_weight(c->lstm_ih[l], data); if (r2 != MV_OK) return r2; }
            else CD_TRY(hipMemcpy(c->lstm_hh[l], data, (size_t)numel * 4, hipMemcpyHostToDevice));
            c->lstm_loaded[l] |= ih ? 1 : 2;
        } else {
            const int rc = expect((int64_t)4 * H);
            if (rc != MV_OK) return rc;
            for (int k = 0; k < 4 * H; ++k) c->lstm_bias_host[l][k] += data[k];
            c->lstm_loaded[l] |= ih ? 4 : 8;
        }
        return MV_OK;
    }
    if (sscanf(name, "decoder.layers.%d.block.%d.conv.%31s", &i, &blk, leaf) == 3 || sscanf(name, "decoder.layers.%d.shortcut.conv.%31s", &i, leaf) == 2 ||
        sscanf(name, "decoder.layers.%d.conv.%31s", &i, leaf) == 2) {
        const bool is_block = n.find(".block.") != std::string::npos, is_short = n.find(".shortcut.") != std::string::npos;
        // module indices of EncodecDecoder.layers: 0 conv_in, 1 lstm, then per stage [ELU, ConvT, ResBlock], then ELU, conv_out
        if (!is_block && !is_short) {
            if (i == 0) return conv_param(c->conv_in, leaf);
            if (i == 2 + 3 * g.n_ratios + 1) return conv_param(c->conv_out, leaf);
            const int st = (i - 3) / 3;
            if (i >= 3 && (i - 3) % 3 == 0 && st < g.n_ratios) return conv_param(c->up[st], leaf);
        } else {
            const int st = (i - 4) / 3;
            if (i >= 4 && (i - 4) % 3 == 0 && st < g.n_ratios) {
                if (is_short) return conv_param(c->shortcut[st], leaf);
                if (blk == 1) return conv_param(c->res1[st], leaf);
                if (blk == 3) return conv_param(c->res2[st], leaf);
            }
        }
    }
    return mv_fail(MV_ERR_INVALID, "unknown codec tensor '%s'", name);
}

int mv_codec_finalize(mv_codec* c) {
    if (!c) return mv_fail(MV_ERR_INVALID, "null handle");
    auto ok = [](const Conv& cv) { return cv.loaded_w && cv.loaded_b; };
    bool all = ok(c->conv_in) && ok(c->conv_out);
    for (int i = 0; i < c->cfg.n_ratios; ++i) all = all && ok(c->up[i]) && ok(c->res1[i]) && ok(c->res2[i]) && ok(c->shortcut[i]);
    for (int q = 0; q < c->cfg.n_q; ++q) all = all && c->embed_loaded[q];
    for (int l = 0; l < c->cfg.lstm_layers; ++l) all = all && c->lstm_loaded[l] == 15;
    if (!all) return mv_fail(MV_ERR_MISSING_WEIGHT, "codec: not every tensor was loaded");
    for (int l = 0; l < c->cfg.lstm_layers; ++l)
        CD_TRY(hipMemcpy(c->lstm_ih[l].b, c->lstm_bias_host[l].data(), c->lstm_bias_host[l].size() * 4, hipMemcpyHostToDevice));
    c->finalized = true;
    return MV_OK;
}

int64_t mv_codec_samples(const mv_codec* c, int32_t frames) {
    if (!c) return 0;
    int64_t n = frames;
    for (int i = 0; i < c->cfg.n_ratios; ++i) n *= c->cfg.ratios[i];
    return n;
}

}  // extern "C"

// Shared body of mv_codec_decode (results copied to host buffers) and mv_codec_decode_device (results left in the codec's own
// device workspace for the next stage).
static int codec_run(mv_codec* c, const int32_t* tokens, int32_t n_q, int32_t frames, float* wav_out, float* latent_out,
                     const float** d_wav, const float** d_latent) {
    if (!c || !c->finalized) return mv_fail(MV_ERR_STATE, "codec not finalized");
    const mv_codec_config& g = c->cfg;
    if (!tokens || (!wav_out && !d_wav) || n_q < 1 || n_q > g.n_q || frames < 8) return mv_fail(MV_ERR_INVALID, "bad tokens / n_q / frames (>= 8)");
    for (size_t i = 0; i < (size_t)n_q * frames; ++i)
        if (tokens[i] < 0 || tokens[i] >= g.bins) return mv_fail(MV_ERR_INVALID, "token %d outside the codebook", tokens[i]);
    hipStream_t s = c->stream;
    const int64_t n_out = mv_codec_samples(c, frames);
    // two ping-pong activation buffers sized for the widest tensor, plus scratch for the ELU / residual paths
    size_t widest = (size_t)4 * c->dim0 * frames;
    {
        int ch = c->dim0;
        int64_t T = frames;
        for (int i = 0; i < g.n_ratios; ++i) { T *= g.ratios[i]; ch /= 2; widest = std::max(widest, (size_t)ch * T); }
    }
    int32_t* d_tok = nullptr;
    float* buf[4] = {nullptr, nullptr, nullptr, nullptr};
    float* d_lat_keep = nullptr;
    auto cleanup = [&]() {};   // the workspace stays with the codec (hipMalloc / hipFree per call cost more than the small kernels)
#define CD_TRY2(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) { cleanup(); return mv_fail(MV_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(_e)); } } while (0)
    {
        const size_t need[7] = {(size_t)n_q * frames * 4, widest * 4, widest * 4, widest * 4, widest * 4, (size_t)g.hidden * frames * 4,
                                g.lstm_layers == 2 ? (size_t)frames * c->dim0 * 8 : 0};
        for (int i = 0; i < 7; ++i)
            if (c->ws_cap[i] < need[i]) {
                if (c->ws[i]) { CD_TRY2(hipStreamSynchronize(s)); (void)hipFree(c->ws[i]); c->ws[i] = nullptr; c->ws_cap[i] = 0; }
                CD_TRY2(hipMalloc(&c->ws[i], need[i]));
                c->ws_cap[i] = need[i];
                if (i == 6) CD_TRY2(hipMemsetAsync(c->ws[i], 0, need[i], s));   // granule tags start from a known state
            }
        d_tok = (int32_t*)c->ws[0];
        for (int i = 0; i < 4; ++i) buf[i] = (float*)c->ws[1 + i];
        d_lat_keep = (float*)c->ws[5];
    }
    CD_TRY2(hipMemcpyAsync(d_tok, tokens, (size_t)n_q * frames * 4, hipMemcpyHostToDevice, s));
    int T = frames;
    float *x = buf[0], *e = buf[1], *u = buf[2], *h = buf[3];
    hipLaunchKernelGGL(rvq_decode_kernel, dim3((T + 255) / 256, g.hidden), dim3(256), 0, s, d_tok, c->embed, e, n_q, g.bins, g.hidden, T);
    if (latent_out) CD_TRY2(hipMemcpyAsync(latent_out, e, (size_t)g.hidden * T * 4, hipMemcpyDeviceToHost, s));
    if (d_latent) CD_TRY2(hipMemcpyAsync(d_lat_keep, e, (size_t)g.hidden * T * 4, hipMemcpyDeviceToDevice, s));  // `e` is reused below
    CD_TRY2(run_conv(s, c->conv_in, e, x, T, false));
    // LSTM stack with a skip connection around it (EncodecLSTM.forward): y = lstm(x) + x
    {
        const int H = c->dim0;
        float* cur = x;   // input of the layer, [H][T]
        float* gin = u;   // [4H][T]
        if (g.lstm_layers == 2 && !lstm_stepwise_mode()) {
            // both layers in ONE persistent launch (lstm.hip: lstm_seq2_kernel): layer 2's step t right behind layer 1's step t
            CD_TRY2(run_conv(s, c->lstm_ih[0], cur, gin, T, false));
            if (++c->lstm_serial >= 65535u) {
                CD_TRY2(hipMemsetAsync(c->lstm_hx, 0, (size_t)4 * H * 8, s));
                CD_TRY2(hipMemsetAsync(c->ws[6], 0, c->ws_cap[6], s));
                c->lstm_serial = 1;
            }
            LstmSeq2Args a{};
            a.gin1 = gin; a.gs_t = 1; a.gs_r = T;
            a.whh1 = c->lstm_hh[0];
            a.wih2 = c->lstm_ih[1].w; a.wih2_ks = c->lstm_ih[1].coutp; a.bias2 = c->lstm_ih[1].b; a.whh2 = c->lstm_hh[1];
            a.y = e; a.ys_t = 1; a.ys_u = T; a.skip = x;   // y = lstm(x) + x lands in `e`
            a.hx1 = c->lstm_hx; a.hx2 = c->lstm_hx + (size_t)2 * H; a.seq1 = (unsigned long long*)c->ws[6];
            a.tag_base = c->lstm_serial << 16; a.abort_flag = c->lstm_abort; a.H = H; a.T = T;
            CD_TRY2(launch_lstm_seq2(s, a));
        } else
        for (int l = 0; l < g.lstm_layers; ++l) {
            CD_TRY2(run_conv(s, c->lstm_ih[l], cur, gin, T, false));
            float* y = (l == g.lstm_layers - 1) ? e : h;  // the last layer adds the skip and lands in `e`
            if (++c->lstm_serial >= 65535u) { CD_TRY2(hipMemsetAsync(c->lstm_hx, 0, (size_t)2 * H * 8, s)); c->lstm_serial = 1; }
            LstmSeqArgs a{};   // one persistent launch for all T steps of the layer (was T dependent launches)
            a.gin = gin; a.gs_b = 0; a.gs_t = 1; a.gs_r = T;
            a.whh = c->lstm_hh[l];
            a.y = y; a.ys_b = 0; a.ys_t = 1; a.ys_u = T;
            a.skip = (l == g.lstm_layers - 1) ? x : nullptr;
            a.hx = c->lstm_hx; a.tag_base = c->lstm_serial << 16; a.abort_flag = c->lstm_abort;
            a.B = 1; a.H = H; a.T = T;
            CD_TRY2(launch_lstm_seq(s, a));
            cur = y;
        }
        if (g.lstm_layers > 0) std::swap(x, e);
    }
    int ch = c->dim0;
    for (int i = 0; i < g.n_ratios; ++i) {
        const int r = g.ratios[i], half = ch / 2;
        hipLaunchKernelGGL(elu_kernel, dim3((unsigned)(((size_t)ch * T + 255) / 256)), dim3(256), 0, s, x, e, (size_t)ch * T);
        CD_TRY2(run_conv(s, c->up[i], e, u, T, false));   // [half][T*r]
        T *= r;
        hipLaunchKernelGGL(elu_kernel, dim3((unsigned)(((size_t)half * T + 255) / 256)), dim3(256), 0, s, u, e, (size_t)half * T);
        CD_TRY2(run_conv(s, c->res1[i], e, h, T, false));
        const size_t nh = (size_t)(half / g.compress) * T;
        hipLaunchKernelGGL(elu_kernel, dim3((unsigned)((nh + 255) / 256)), dim3(256), 0, s, h, e, nh);
        CD_TRY2(run_conv(s, c->res2[i], e, x, T, false));
        CD_TRY2(run_conv(s, c->shortcut[i], u, x, T, true));  // x = shortcut(u) + block(u)
        ch = half;
    }
    hipLaunchKernelGGL(elu_kernel, dim3((unsigned)(((size_t)ch * T + 255) / 256)), dim3(256), 0, s, x, e, (size_t)ch * T);
    CD_TRY2(run_conv(s, c->conv_out, e, u, T, false));
    if (wav_out) CD_TRY2(hipMemcpyAsync(wav_out, u, (size_t)g.audio_channels * n_out * 4, hipMemcpyDeviceToHost, s));
    if (d_wav) *d_wav = u;
    if (d_latent) *d_latent = d_lat_keep;
    uint32_t aborted = 0;
    if (c->lstm_abort) CD_TRY2(hipMemcpyAsync(&aborted, c->lstm_abort, 4, hipMemcpyDeviceToHost, s));
    CD_TRY2(hipStreamSynchronize(s));
#undef CD_TRY2
    cleanup();
    if (aborted) {
        (void)hipMemset(c->lstm_abort, 0, 4);
        if (!lstm_stepwise_mode()) {   // survive it: redo the decode with one LSTM launch per time step (and stay there)
            lstm_force_stepwise();
            return codec_run(c, tokens, n_q, frames, wav_out, latent_out, d_wav, d_latent);
        }
        return mv_fail(MV_ERR_HIP, "mv_codec_decode: the LSTM recurrence failed");
    }
    return MV_OK;
}

extern "C" {

int mv_codec_decode(mv_codec* c, const int32_t* tokens, int32_t n_q, int32_t frames, float* wav_out, float* latent_out) {
    if (!wav_out) return mv_fail(MV_ERR_INVALID, "mv_codec_decode: wav_out is null");
    return codec_run(c, tokens, n_q, frames, wav_out, latent_out, nullptr, nullptr);
}

int mv_codec_decode_device(mv_codec* c, const int32_t* tokens, int32_t n_q, int32_t frames, const float** d_wav, const float** d_latent) {
    if (!d_wav) return mv_fail(MV_ERR_INVALID, "mv_codec_decode_device: d_wav is null");
    return codec_run(c, tokens, n_q, frames, nullptr, nullptr, d_wav, d_latent);
}

}  // extern "C"
