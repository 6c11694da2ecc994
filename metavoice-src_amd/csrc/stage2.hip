// Second stage: the small non-causal GPT that expands 2 EnCodec codebooks into the remaining 6, one forward over all
// positions + one independent draw per (position, codebook).
//
// Replaces, for the `TTS.synthesise` path (fam/llm/fast_inference.py:141-155):
//   GPT.forward, non-causal branch            fam/llm/model.py:195-314
//   Block / SelfAttention / MLP / norms       fam/llm/layers/combined.py:7-52, attn.py:132-159, layers.py:8-72
//   NonCausalInferenceMixin._non_causal_sample fam/llm/mixins/non_causal.py:15-67
// under bf16 autocast (fam/llm/inference.py:80-82): Linear inputs/outputs, SDPA output and activation outputs are bf16;
// embeddings, norms and the residual stream are fp32.
//
// The model is ~10 M parameters (README.md:164) -- it lives in L2 / Infinity Cache, so nothing here is HBM-bound: the
// forward is ~10 launches per layer of MFMA GEMMs (gemm.hip) with fused bias / GELU / SwiGLU / fp32-residual epilogues,
// row norms and the L2-streaming multi-query attention.  Shape-generic: every size comes from the checkpoint's
// `model_args` (the real second_stage.pt is not available offline).
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "common.h"
#include "kernels.h"

int mv_fail(int code, const char* fmt, ...);  // engine.hip

#define S2_TRY(expr)                                                                                         \
    do {                                                                                                     \
        hipError_t _e = (expr);                                                                              \
        if (_e != hipSuccess) return mv_fail(MV_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), \
                                             __FILE__, __LINE__);                                            \
    } while (0)

namespace {

__global__ void cvt_f32_bf16_kernel(const float* in, bf16_t* out, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = f2bf(in[i]);
}
__global__ void cvt_bf16_f32_kernel(const bf16_t* in, float* out, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = bf2f(in[i]);
}

// x[b][t][:] = sum_i wte_i[idx[b][i][t]] + wpe[t] + spk[b]   (model.py:232-283), fp32
struct S2EmbedArgs {
    const int32_t* idx;       // [b][n_in][T]
    const float* wte[4];
    const float* wpe;
    const bf16_t* spk;        // [b][C] bf16 (the autocast Linear output) or null
    float* x;                 // [b*T][C]
    int32_t n_in, T, C;
};
__global__ void s2_embed_kernel(const S2EmbedArgs a) {
    const int t = blockIdx.x, b = blockIdx.y;
    for (int c = threadIdx.x; c < a.C; c += blockDim.x) {
        float v = 0.f;
        for (int i = 0; i < a.n_in; ++i) v += a.wte[i][(size_t)a.idx[((size_t)b * a.n_in + i) * a.T + t] * a.C + c];
        v += a.wpe[(size_t)t * a.C + c];
        if (a.spk) v += bf2f(a.spk[(size_t)b * a.C + c]);
        a.x[((size_t)b * a.T + t) * a.C + c] = v;
    }
}

// spk[b][n] = bf16(sum_k bf16(e[b][k]) * W[n][k] (+ bias))   -- `speaker_cond_pos` under autocast (model.py:121,236)
__global__ void s2_spk_kernel(int C, int spk_dim, const bf16_t* e, const bf16_t* w, const bf16_t* bias, bf16_t* out) {
    const int b = blockIdx.y;
    const int n = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (n >= C) return;
    float acc = 0.f;
    for (int k = lane; k < spk_dim; k += 64) acc += bf2f(e[(size_t)b * spk_dim + k]) * bf2f(w[(size_t)n * spk_dim + k]);
    acc = wave_sum(acc);
    if (lane == 0) out[(size_t)b * C + n] = f2bf(acc + (bias ? bf2f(bias[n]) : 0.f));
}

__device__ __forceinline__ uint32_t s2_philox_word0(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u;
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        const uint32_t hi0 = __umulhi(M0, c0), lo0 = M0 * c0;
        const uint32_t hi1 = __umulhi(M1, c2), lo1 = M1 * c2;
        const uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return c0;
}
__device__ __forceinline__ uint32_t f32_order_key(float f) {
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// One wave per (position, codebook): /T, top-k mask (`logits < kth value` -> -inf), softmax, one multinomial draw
// (non_causal.py:36-64).  The draw is the inverse CDF in id order at u ~ U(0,1) from Philox keyed by (seed; row, codebook).
struct S2SampleArgs {
    const bf16_t* logits[8];  // [M][V_i]
    int32_t vocab[8];
    int32_t* out;             // [b][n_out][T]
    int32_t T, n_out, top_k;
    float temperature;
    uint64_t seed;
};
#define S2_NPL 32
__global__ void __launch_bounds__(64) s2_sample_kernel(const S2SampleArgs a) {
    const int m = blockIdx.x, hd = blockIdx.y, lane = threadIdx.x;
    const int V = a.vocab[hd];
    const bf16_t* lg = a.logits[hd] + (size_t)m * V;
    const int npl = (V + 63) / 64, i0 = lane * npl;
    float v[S2_NPL], e[S2_NPL];
    uint32_t key[S2_NPL];
    bool ok[S2_NPL];
    float lmax = -INFINITY;
#pragma unroll
    for (int j = 0; j < S2_NPL; ++j) {
        const int i = i0 + j;
        ok[j] = (j < npl) && (i < V);
        v[j] = ok[j] ? bf2f(lg[i]) / a.temperature : -INFINITY;
        key[j] = ok[j] ? f32_order_key(v[j]) : 0u;
        lmax = fmaxf(lmax, v[j]);
    }
    const float vmax = wave_max(lmax);
    uint32_t X = 0u;  // keys below X are masked
    if (a.top_k > 0 && a.top_k < V) {
        const float kk = (float)a.top_k;
        for (int lvl = 15; lvl >= 0; --lvl) {  // largest X with #{key >= X} >= k  (= key of the k-th largest value)
            const uint32_t step = 1u << (2 * lvl);
            float c1 = 0.f, c2 = 0.f, c3 = 0.f;
#pragma unroll
            for (int j = 0; j < S2_NPL; ++j) {
                c1 += (ok[j] && key[j] >= X + step) ? 1.f : 0.f;
                c2 += (ok[j] && key[j] >= X + 2 * step) ? 1.f : 0.f;
                c3 += (ok[j] && key[j] >= X + 3 * step) ? 1.f : 0.f;
            }
            c1 = wave_sum(c1); c2 = wave_sum(c2); c3 = wave_sum(c3);
            uint32_t d = 0;
            if (c1 >= kk) d = 1;
            if (c2 >= kk) d = 2;
            if (c3 >= kk) d = 3;
            X += d * step;
        }
    }
    float ls = 0.f;
#pragma unroll
    for (int j = 0; j < S2_NPL; ++j) {
        e[j] = (ok[j] && key[j] >= X) ? __expf(v[j] - vmax) : 0.f;
        ls += e[j];
    }
    const float S = wave_sum(ls);
    const uint32_t w = s2_philox_word0((uint32_t)m, (uint32_t)hd, 1u, 0u, (uint32_t)a.seed, (uint32_t)(a.seed >> 32));
    const float target = (((float)(w >> 8) + 0.5f) * 5.9604644775390625e-08f) * S;
    float incl = ls;  // inclusive scan of the per-lane masses (lane order = id order)
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const float n = __shfl_up(incl, off, 64);
        if (lane >= off) incl += n;
    }
    float cum = incl - ls, cnt = 0.f;
#pragma unroll
    for (int j = 0; j < S2_NPL; ++j) {
        cum += e[j];
        cnt += (ok[j] && cum <= target) ? 1.f : 0.f;
    }
    const int choice = min((int)wave_sum(cnt), V - 1);
    if (lane == 0) {
        const int b = m / a.T, t = m % a.T;
        a.out[((size_t)b * a.n_out + hd) * a.T + t] = choice;
    }
}

struct Slot {
    std::string name;
    void** ptr;
    int64_t rows, cols;
    bool bf16;      // stored as bf16 (Linear weights/biases under autocast) or fp32 (embeddings, norms)
    bool optional;
    bool loaded = false;
};
struct S2Layer {
    float *ln1_w = nullptr, *ln1_b = nullptr, *ln2_w = nullptr, *ln2_b = nullptr;
    bf16_t *attn_w = nullptr, *attn_b = nullptr, *proj_w = nullptr, *proj_b = nullptr;
    bf16_t *w1 = nullptr, *b1 = nullptr, *w3 = nullptr, *b3 = nullptr, *fc_w = nullptr, *fc_b = nullptr;
    bf16_t *out_w = nullptr, *out_b = nullptr;
};

}  // namespace

struct mv_stage2 {
    mv_stage2_config cfg{};
    hipStream_t stream = nullptr;
    std::vector<Slot> slots;
    std::vector<void*> allocs;
    float* wte[4] = {nullptr, nullptr, nullptr, nullptr};
    float *wpe = nullptr, *lnf_w = nullptr, *lnf_b = nullptr;
    bf16_t *spk_w = nullptr, *spk_b = nullptr;
    bf16_t* heads[8] = {};
    std::vector<S2Layer> layers;
    bool finalized = false;
    // workspace
    int32_t* idx = nullptr;
    bf16_t *spk_in = nullptr, *spk = nullptr;
    float* x = nullptr;
    bf16_t *xn = nullptr, *qkv = nullptr, *y = nullptr, *h = nullptr, *vt = nullptr;
    bf16_t* logits[8] = {};
    int32_t* tokens = nullptr;
    int cur_batch = 0, cur_T = 0;
};

extern "C" {

int mv_stage2_create(const mv_stage2_config* cfg, mv_stage2** out) {
    if (!cfg || !out) return mv_fail(MV_ERR_INVALID, "null argument");
    const mv_stage2_config& c = *cfg;
    if (c.n_layer < 1 || c.n_head < 1 || c.n_embd % c.n_head != 0 || c.n_embd % 8 != 0)
        return mv_fail(MV_ERR_INVALID, "bad n_embd/n_head %d/%d", c.n_embd, c.n_head);
    const int hd = c.n_embd / c.n_head;
    if (hd != 64 && hd != 128) return mv_fail(MV_ERR_INVALID, "head_dim %d not supported (64 or 128)", hd);
    if (c.n_in < 1 || c.n_in > 4 || c.n_out < 1 || c.n_out > 8) return mv_fail(MV_ERR_INVALID, "n_in/n_out out of range");
    for (int i = 0; i < c.n_out; ++i)
        if (c.vocab_out[i] < 2 || c.vocab_out[i] > 64 * S2_NPL) return mv_fail(MV_ERR_INVALID, "target vocab %d > %d", c.vocab_out[i], 64 * S2_NPL);
    if (c.hidden % 8 != 0 || c.block_size < 1 || c.max_batch < 1) return mv_fail(MV_ERR_INVALID, "bad hidden/block_size/max_batch");
    mv_stage2* s = new mv_stage2();
    s->cfg = c;
    s->layers.resize(c.n_layer);
    if (hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking) != hipSuccess) { delete s; return mv_fail(MV_ERR_HIP, "stream creation failed"); }
    auto add = [&](const std::string& n, void** p, int64_t r, int64_t cc, bool bf, bool opt) { s->slots.push_back(Slot{n, p, r, cc, bf, opt}); };
    const int C = c.n_embd;
    for (int i = 0; i < c.n_in; ++i) add("transformer.wtes." + std::to_string(i) + ".weight", (void**)&s->wte[i], c.vocab_in[i], C, false, false);
    add("transformer.wpe.weight", (void**)&s->wpe, c.block_size, C, false, false);
    if (c.spk_dim > 0) {
        add("speaker_cond_pos.weight", (void**)&s->spk_w, C, c.spk_dim, true, false);
        add("speaker_cond_pos.bias", (void**)&s->spk_b, 1, C, true, true);
    }
    for (int l = 0; l < c.n_layer; ++l) {
        S2Layer& L = s->layers[l];
        const std::string p = "transformer.h." + std::to_string(l) + ".";
        const bool ln_bias = c.norm_type == 1 && c.bias;
        add(p + "ln_1.weight", (void**)&L.ln1_w, 1, C, false, false);
        add(p + "ln_1.bias", (void**)&L.ln1_b, 1, C, false, !ln_bias);
        add(p + "attn.c_attn.weight", (void**)&L.attn_w, 3 * C, C, true, false);
        add(p + "attn.c_attn.bias", (void**)&L.attn_b, 1, 3 * C, true, !c.bias);
        add(p + "attn.c_proj.weight", (void**)&L.proj_w, C, C, true, false);
        add(p + "attn.c_proj.bias", (void**)&L.proj_b, 1, C, true, !c.bias);
        add(p + "ln_2.weight", (void**)&L.ln2_w, 1, C, false, false);
        add(p + "ln_2.bias", (void**)&L.ln2_b, 1, C, false, !ln_bias);
        if (c.act == 0) {
            add(p + "mlp.swiglu.w1.weight", (void**)&L.w1, c.hidden, C, true, false);
            add(p + "mlp.swiglu.w1.bias", (void**)&L.b1, 1, c.hidden, true, !c.bias);
            add(p + "mlp.swiglu.w3.weight", (void**)&L.w3, c.hidden, C, true, false);
            add(p + "mlp.swiglu.w3.bias", (void**)&L.b3, 1, c.hidden, true, !c.bias);
        } else {
            add(p + "mlp.c_fc.weight", (void**)&L.fc_w, c.hidden, C, true, false);
            add(p + "mlp.c_fc.bias", (void**)&L.fc_b, 1, c.hidden, true, !c.bias);
        }
        add(p + "mlp.c_proj.weight", (void**)&L.out_w, C, c.hidden, true, false);
        add(p + "mlp.c_proj.bias", (void**)&L.out_b, 1, C, true, !c.bias);
    }
    add("transformer.ln_f.weight", (void**)&s->lnf_w, 1, C, false, false);
    add("transformer.ln_f.bias", (void**)&s->lnf_b, 1, C, false, !(c.norm_type == 1 && c.bias));
    for (int i = 0; i < c.n_out; ++i) add("lm_heads." + std::to_string(i) + ".weight", (void**)&s->heads[i], c.vocab_out[i], C, true, false);
    *out = s;
    return MV_OK;
}

int mv_stage2_destroy(mv_stage2* s) {
    if (!s) return MV_OK;
    if (s->stream) (void)hipStreamSynchronize(s->stream);
    for (void* p : s->allocs) (void)hipFree(p);
    if (s->stream) (void)hipStreamDestroy(s->stream);
    delete s;
    return MV_OK;
}

int mv_stage2_load_weight(mv_stage2* s, const char* name, const void* data, int32_t dtype, int64_t rows, int64_t cols) {
    if (!s || !name || !data) return mv_fail(MV_ERR_INVALID, "null argument");
    if (dtype != MV_BF16 && dtype != MV_F32) return mv_fail(MV_ERR_INVALID, "%s: dtype must be bf16 or f32", name);
    for (auto& sl : s->slots) {
        if (sl.name != name) continue;
        if (rows * cols != sl.rows * sl.cols || (sl.rows != 1 && (rows != sl.rows || cols != sl.cols)))
            return mv_fail(MV_ERR_INVALID, "%s: shape (%lld,%lld) != expected (%lld,%lld)", name, (long long)rows, (long long)cols,
                           (long long)sl.rows, (long long)sl.cols);
        const size_t n = (size_t)rows * cols, in_b = dtype == MV_BF16 ? 2 : 4, out_b = sl.bf16 ? 2 : 4;
        void* dst = nullptr;
        S2_TRY(hipMalloc(&dst, (n * out_b + 255) / 256 * 256));
        s->allocs.push_back(dst);
        if ((dtype == MV_BF16) == sl.bf16) {
            S2_TRY(hipMemcpy(dst, data, n * in_b, hipMemcpyDefault));
        } else {
            void* tmp = nullptr;
            S2_TRY(hipMalloc(&tmp, n * in_b));
            hipError_t ce = hipMemcpy(tmp, data, n * in_b, hipMemcpyDefault);
            if (ce == hipSuccess) {
                const unsigned blocks = (unsigned)((n + 255) / 256);
                if (sl.bf16) hipLaunchKernelGGL(cvt_f32_bf16_kernel, dim3(blocks), dim3(256), 0, s->stream, (const float*)tmp, (bf16_t*)dst, n);
                else hipLaunchKernelGGL(cvt_bf16_f32_kernel, dim3(blocks), dim3(256), 0, s->stream, (const bf16_t*)tmp, (float*)dst, n);
                ce = hipStreamSynchronize(s->stream);
            }
            (void)hipFree(tmp);
            if (ce != hipSuccess) return mv_fail(MV_ERR_HIP, "%s: upload failed: %s", name, hipGetErrorString(ce));
        }
        *sl.ptr = dst;
        sl.loaded = true;
        return MV_OK;
    }
    return mv_fail(MV_ERR_INVALID, "unknown second-stage tensor '%s'", name);
}

int mv_stage2_finalize(mv_stage2* s) {
    if (!s) return mv_fail(MV_ERR_INVALID, "null handle");
    if (s->finalized) return MV_OK;
    for (auto& sl : s->slots)
        if (!sl.loaded && !sl.optional) return mv_fail(MV_ERR_MISSING_WEIGHT, "tensor '%s' was never loaded", sl.name.c_str());
    const mv_stage2_config& c = s->cfg;
    const size_t M = (size_t)c.max_batch * c.block_size;
    auto alloc = [&](void** p, size_t bytes) -> hipError_t {
        hipError_t e = hipMalloc(p, bytes);
        if (e == hipSuccess) s->allocs.push_back(*p);
        return e;
    };
    S2_TRY(alloc((void**)&s->idx, M * c.n_in * 4));
    S2_TRY(alloc((void**)&s->spk_in, (size_t)c.max_batch * (c.spk_dim > 0 ? c.spk_dim : 1) * 2));
    S2_TRY(alloc((void**)&s->spk, (size_t)c.max_batch * c.n_embd * 2));
    S2_TRY(alloc((void**)&s->x, M * c.n_embd * 4));
    S2_TRY(alloc((void**)&s->xn, M * c.n_embd * 2));
    S2_TRY(alloc((void**)&s->qkv, M * 3 * c.n_embd * 2));
    S2_TRY(alloc((void**)&s->y, M * c.n_embd * 2));
    S2_TRY(alloc((void**)&s->h, M * c.hidden * 2));
    S2_TRY(alloc((void**)&s->vt, attn_rows_scratch_elems(c.max_batch, c.n_head, c.n_embd / c.n_head, c.block_size) * 2));  // V transposed (MFMA attention)
    for (int i = 0; i < c.n_out; ++i) S2_TRY(alloc((void**)&s->logits[i], M * c.vocab_out[i] * 2));
    S2_TRY(alloc((void**)&s->tokens, M * c.n_out * 4));
    s->finalized = true;
    return MV_OK;
}

int mv_stage2_forward(mv_stage2* s, int32_t batch, int32_t T, const int32_t* idx, const uint16_t* spk_emb) {
    if (!s || !s->finalized) return mv_fail(MV_ERR_STATE, "second stage not finalized");
    const mv_stage2_config& c = s->cfg;
    if (!idx || batch < 1 || batch > c.max_batch || T < 1 || T > c.block_size) return mv_fail(MV_ERR_INVALID, "bad batch/T %d/%d", batch, T);
    for (size_t i = 0; i < (size_t)batch * c.n_in * T; ++i) {
        const int which = (int)((i / T) % c.n_in);
        if (idx[i] < 0 || idx[i] >= c.vocab_in[which]) return mv_fail(MV_ERR_INVALID, "token %d out of hierarchy %d vocabulary", idx[i], which);
    }
    hipStream_t st = s->stream;
    const int C = c.n_embd, M = batch * T, hd = C / c.n_head;
    S2_TRY(hipMemcpyAsync(s->idx, idx, (size_t)batch * c.n_in * T * 4, hipMemcpyHostToDevice, st));
    const bool use_spk = spk_emb != nullptr && c.spk_dim > 0;
    if (use_spk) {
        S2_TRY(hipMemcpyAsync(s->spk_in, spk_emb, (size_t)batch * c.spk_dim * 2, hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(s2_spk_kernel, dim3((C + 3) / 4, batch), dim3(256), 0, st, C, c.spk_dim, s->spk_in, s->spk_w, s->spk_b, s->spk);
    }
    S2EmbedArgs ea;
    ea.idx = s->idx; ea.wpe = s->wpe; ea.spk = use_spk ? s->spk : nullptr; ea.x = s->x; ea.n_in = c.n_in; ea.T = T; ea.C = C;
    for (int i = 0; i < 4; ++i) ea.wte[i] = s->wte[i];
    hipLaunchKernelGGL(s2_embed_kernel, dim3(T, batch), dim3(128), 0, st, ea);
    auto norm = [&](const float* w, const float* b) -> hipError_t {
        NormArgs n;
        n.x_f32 = s->x; n.w_f32 = w; n.b_f32 = b; n.out = s->xn; n.M = M; n.C = C; n.mode = c.norm_type == 0 ? 1 : 2;
        n.eps = c.norm_type == 0 ? c.norm_eps : 1e-5f;
        return launch_norm_rows(st, n);
    };
    for (int l = 0; l < c.n_layer; ++l) {
        const S2Layer& L = s->layers[l];
        S2_TRY(norm(L.ln1_w, L.ln1_b));
        GemmArgs g;
        g.a = s->xn; g.w = L.attn_w; g.bias = L.attn_b; g.out_bf16 = s->qkv; g.M = M; g.N = 3 * C; g.K = C; g.lda = C; g.ldc = 3 * C;
        S2_TRY(launch_gemm(st, g));
        AttnRowsArgs at;
        at.q = s->qkv; at.k = s->qkv + C; at.v = s->qkv + 2 * C; at.out = s->y;
        at.k_batch_stride = (int64_t)T * 3 * C; at.k_head_stride = hd; at.k_key_stride = 3 * C; at.q_row_stride = 3 * C;
        at.out_row_stride = C; at.T = T; at.n_head = c.n_head; at.batch = batch; at.head_dim = hd; at.causal = c.causal;
        at.vt_scratch = (hd == 64 || hd == 128) ? s->vt : nullptr;
        S2_TRY(launch_attn_rows(st, at));
        g = GemmArgs();
        g.a = s->y; g.w = L.proj_w; g.bias = L.proj_b; g.res_f32 = s->x; g.out_f32 = s->x; g.M = M; g.N = C; g.K = C; g.lda = C; g.ldc = C;
        S2_TRY(launch_gemm(st, g));
        S2_TRY(norm(L.ln2_w, L.ln2_b));
        g = GemmArgs();
        g.a = s->xn; g.out_bf16 = s->h; g.M = M; g.N = c.hidden; g.K = C; g.lda = C; g.ldc = c.hidden;
        if (c.act == 0) { g.w = L.w1; g.w2 = L.w3; g.bias = L.b1; g.bias2 = L.b3; }
        else { g.w = L.fc_w; g.bias = L.fc_b; g.act = GEMM_ACT_GELU; }
        S2_TRY(launch_gemm(st, g));
        g = GemmArgs();
        g.a = s->h; g.w = L.out_w; g.bias = L.out_b; g.res_f32 = s->x; g.out_f32 = s->x; g.M = M; g.N = C; g.K = c.hidden; g.lda = c.hidden; g.ldc = C;
        S2_TRY(launch_gemm(st, g));
    }
    S2_TRY(norm(s->lnf_w, s->lnf_b));
    for (int i = 0; i < c.n_out; ++i) {
        GemmArgs g;
        g.a = s->xn; g.w = s->heads[i]; g.out_bf16 = s->logits[i]; g.M = M; g.N = c.vocab_out[i]; g.K = C; g.lda = C; g.ldc = c.vocab_out[i];
        S2_TRY(launch_gemm(st, g));
    }
    S2_TRY(hipStreamSynchronize(st));
    s->cur_batch = batch;
    s->cur_T = T;
    return MV_OK;
}

int mv_stage2_get_logits(mv_stage2* s, int32_t head, float* out) {
    if (!s || s->cur_batch == 0 || !out) return mv_fail(MV_ERR_STATE, "no forward has run");
    if (head < 0 || head >= s->cfg.n_out) return mv_fail(MV_ERR_INVALID, "head out of range");
    const size_t n = (size_t)s->cur_batch * s->cur_T * s->cfg.vocab_out[head];
    float* tmp = nullptr;
    S2_TRY(hipMalloc((void**)&tmp, n * 4));
    hipLaunchKernelGGL(cvt_bf16_f32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s->stream, s->logits[head], tmp, n);
    hipError_t ce = hipMemcpyAsync(out, tmp, n * 4, hipMemcpyDeviceToHost, s->stream);
    if (ce == hipSuccess) ce = hipStreamSynchronize(s->stream);
    (void)hipFree(tmp);
    if (ce != hipSuccess) return mv_fail(MV_ERR_HIP, "logits copy failed: %s", hipGetErrorString(ce));
    return MV_OK;
}

int mv_stage2_sample(mv_stage2* s, float temperature, int32_t top_k, uint64_t seed, int32_t* out) {
    if (!s || s->cur_batch == 0 || !out) return mv_fail(MV_ERR_STATE, "no forward has run");
    if (!(temperature > 0.f)) return mv_fail(MV_ERR_INVALID, "temperature must be > 0");
    const mv_stage2_config& c = s->cfg;
    S2SampleArgs a;
    for (int i = 0; i < 8; ++i) { a.logits[i] = s->logits[i]; a.vocab[i] = i < c.n_out ? c.vocab_out[i] : 0; }
    a.out = s->tokens; a.T = s->cur_T; a.n_out = c.n_out; a.top_k = top_k; a.temperature = temperature; a.seed = seed;
    hipLaunchKernelGGL(s2_sample_kernel, dim3(s->cur_batch * s->cur_T, c.n_out), dim3(64), 0, s->stream, a);
    S2_TRY(hipMemcpyAsync(out, s->tokens, (size_t)s->cur_batch * c.n_out * s->cur_T * 4, hipMemcpyDeviceToHost, s->stream));
    S2_TRY(hipStreamSynchronize(s->stream));
    return MV_OK;
}

}  // extern "C"
