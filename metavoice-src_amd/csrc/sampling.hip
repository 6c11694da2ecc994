// On-device sampler + next-token embedding: one workgroup per utterance, one launch per decode step.
//
// Replaces `sample()` / `logits_to_probs()` / `top_p_sample()` / `multinomial_sample_one_no_sync()`
// (fam/llm/fast_inference_utils.py:61-120), the per-token bookkeeping of `decode_n_tokens` (:148-174: append,
// `input_pos += 1`, end-of-audio test -- done on the device, so the per-token device->host sync of :161 is gone) and
// the input embedding of the NEXT step (fam/llm/fast_model.py:152-157: tok_emb + pos_emb + speaker projection * mask).
//
// Bytes are negligible (2 x 2562 logits); this kernel is latency-bound and sits on the critical path of every step.
//   1. CFG mix `g*cond + (1-g)*uncond` and the temperature divide, each rounded to bf16 exactly where the
//      reference's bf16 tensors are; everything after that is fp32 (DESIGN.md "Sampling").
//   2. top-k / top-p need the ascending order: 64-bit keys (order-preserving float bits << 32 | token id) are bitonic
//      sorted in LDS -- total order, ties by ascending id, same as the oracle's stable sort.
//   3. softmax over the sorted values, inclusive block scan for the ascending cumulative mass, removal of
//      `cum <= 1 - top_p` (largest always kept), renormalise.
//   4. Exp(1) race: argmax p_i / q_i with q from Philox4x32-10 keyed by the utterance's seed, counter
//      (token id, sequence index) -- independent of the slot the utterance occupies; ties to the lowest id.
#include "common.h"
#include "kernels.h"

#define SAMPLE_THREADS 1024
#define SAMPLE_MAXV 4096

__device__ __forceinline__ uint32_t philox4x32_10_word0(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                                        uint32_t k1) {
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

// Exp(1) variate for (token id, step, stream): mirrors oracle/rng.py::exp1_noise.
__device__ __forceinline__ float mv_exp1_noise(uint64_t seed, uint32_t step, uint32_t stream, uint32_t token) {
    const uint32_t w = philox4x32_10_word0(token, step, stream, 0u, (uint32_t)seed, (uint32_t)(seed >> 32));
    const float u = ((float)(w >> 8) + 0.5f) * 5.9604644775390625e-08f;  // 2^-24
    return -logf(u);
}

__device__ __forceinline__ uint32_t float_order_bits(float f) {
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float order_bits_float(uint32_t k) {
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k);
}

// Block-wide helpers (SAMPLE_THREADS threads, `red` has >= 2*16 floats of scratch).
__device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float t = 0.f;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += red[w];
    return t;
}
__device__ __forceinline__ float block_max(float v, float* red) {
    v = wave_max(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float t = -INFINITY;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t = fmaxf(t, red[w]);
    return t;
}

__global__ void __launch_bounds__(SAMPLE_THREADS) sample_kernel(const SampleArgs a) {
    __shared__ __attribute__((aligned(16))) unsigned long long keys[SAMPLE_MAXV];
    __shared__ float val[SAMPLE_MAXV];   // mixed logits (token order), later e_j (sorted order)
    __shared__ float red[64];
    __shared__ unsigned long long best[16];
    __shared__ int s_tok;

    const int u = blockIdx.x;
    const int tid = threadIdx.x;
    const int V = a.vocab;
    UttState st = a.utt[u];
    const bool standalone = (a.token_out != nullptr);
    if (!standalone && st.done) return;

    const int next_index = st.pos + 1;                  // sequence index of the token decided now
    const bool forced = next_index < st.prompt_len;      // teacher-forced prompt token (prefill by decode steps)
    const bf16_t* lc = a.logits + (size_t)(2 * u) * V;
    const bf16_t* lu = a.logits + (size_t)(2 * u + 1) * V;

    if (a.capture != nullptr && st.pos < a.capture_max) {
        float* cap = a.capture + ((size_t)st.pos * (2 * a.n_utt) + 2 * u) * V;
        for (int i = tid; i < V; i += blockDim.x) {
            cap[i] = bf2f(lc[i]);
            cap[V + i] = bf2f(lu[i]);
        }
    }

    int tok = 0;
    if (forced) {
        tok = a.seq[(size_t)u * a.block_size + next_index];
    } else {
        const mv_sampling sp = st.s;
        const float temp = fmaxf(sp.temperature, 1e-5f);
        const bool use_k = sp.top_k > 0;
        const bool use_p = sp.top_p > 0.f;
        int NP = 1;
        while (NP < V) NP <<= 1;

        // ---- 1. CFG mix + temperature (bf16 rounding points of fast_inference_utils.py:115-117,92)
        for (int i = tid; i < NP; i += blockDim.x) {
            if (i < V) {
                const float c = bf2f(lc[i]), un = bf2f(lu[i]);
                const float mixed = round_bf(round_bf(sp.guidance_scale * c) + round_bf(sp.one_minus_guidance * un));
                const float t = round_bf(mixed / temp);
                val[i] = t;
                keys[i] = ((unsigned long long)float_order_bits(t) << 32) | (unsigned)i;
            } else {
                keys[i] = ~0ull;
            }
        }
        __syncthreads();

        float vmax, S2;
        if (use_k || use_p) {
            // ---- 2. ascending bitonic sort of (value, id)
            for (int k = 2; k <= NP; k <<= 1) {
                for (int j = k >> 1; j > 0; j >>= 1) {
                    for (int t = tid; t < (NP >> 1); t += blockDim.x) {
                        const int i = 2 * t - (t & (j - 1));
                        const unsigned long long x = keys[i], y = keys[i + j];
                        const bool asc = ((i & k) == 0);
                        if ((x > y) == asc) { keys[i] = y; keys[i + j] = x; }
                    }
                    __syncthreads();
                }
            }
            vmax = order_bits_float((uint32_t)(keys[V - 1] >> 32));
            float pivot = -INFINITY;
            if (use_k) pivot = order_bits_float((uint32_t)(keys[V - min(sp.top_k, V)] >> 32));

            // ---- 3. e_j = exp(v_j - max) in sorted order (0 for top-k-masked), total mass, cumulative mass
            float loc[4];
            const int j0 = tid * 4;  // NP <= 4096 = 4 * SAMPLE_THREADS
            float lsum = 0.f;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int j = j0 + q;
                float e = 0.f;
                if (j < V) {
                    const float v = order_bits_float((uint32_t)(keys[j] >> 32));
                    e = (v < pivot) ? 0.f : __expf(v - vmax);
                }
                loc[q] = e;
                lsum += e;
            }
            const float S = block_sum(lsum, red);
            bool keep[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) keep[q] = (j0 + q < V) && (loc[q] > 0.f || !use_k);
            if (use_p) {
                // inclusive scan of per-thread sums: wave scan + wave offsets
                float incl = lsum;
#pragma unroll
                for (int off = 1; off < 64; off <<= 1) {
                    const float n = __shfl_up(incl, off, 64);
                    if ((tid & 63) >= off) incl += n;
                }
                __syncthreads();
                if ((tid & 63) == 63) red[32 + (tid >> 6)] = incl;
                __syncthreads();
                float woff = 0.f;
                for (int w = 0; w < (tid >> 6); ++w) woff += red[32 + w];
                float cum = woff + incl - lsum;  // exclusive prefix of this thread, in units of e
                const float thr = 1.0f - sp.top_p;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    cum += loc[q];
                    const int j = j0 + q;
                    if (j < V && j != V - 1 && (cum / S) <= thr) keep[q] = false;
                }
            }
            // ---- renormalise over the survivors
            float ksum = 0.f;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (!keep[q]) loc[q] = 0.f;
                ksum += loc[q];
            }
            S2 = block_sum(ksum, red);
            // ---- 4. race
            unsigned long long bst = 0ull;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int j = j0 + q;
                if (j < V) {
                    const unsigned id = (unsigned)(keys[j] & 0xFFFFFFFFu);
                    const float p = loc[q] / S2;
                    if (a.probs_out) a.probs_out[id] = p;
                    const float r = p / mv_exp1_noise(sp.seed, (uint32_t)next_index, 0u, id);
                    const unsigned long long cand = ((unsigned long long)__float_as_uint(r) << 32) | (0xFFFFFFFFu - id);
                    bst = cand > bst ? cand : bst;
                }
            }
            // wave max of 64-bit candidates
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                const unsigned long long o = __shfl_xor(bst, off, 64);
                bst = o > bst ? o : bst;
            }
            if ((tid & 63) == 0) best[tid >> 6] = bst;
            __syncthreads();
            if (tid == 0) {
                unsigned long long b = 0ull;
                for (int w = 0; w < (int)(blockDim.x >> 6); ++w) b = best[w] > b ? best[w] : b;
                s_tok = (int)(0xFFFFFFFFu - (unsigned)(b & 0xFFFFFFFFu));
            }
        } else {
            // no truncation: plain softmax over all tokens, no sort needed
            float lm = -INFINITY;
            for (int i = tid; i < V; i += blockDim.x) lm = fmaxf(lm, val[i]);
            vmax = block_max(lm, red);
            float ls = 0.f;
            for (int i = tid; i < V; i += blockDim.x) ls += __expf(val[i] - vmax);
            S2 = block_sum(ls, red);
            unsigned long long bst = 0ull;
            for (int i = tid; i < V; i += blockDim.x) {
                const float p = __expf(val[i] - vmax) / S2;
                if (a.probs_out) a.probs_out[i] = p;
                const float r = p / mv_exp1_noise(sp.seed, (uint32_t)next_index, 0u, (unsigned)i);
                const unsigned long long cand = ((unsigned long long)__float_as_uint(r) << 32) | (0xFFFFFFFFu - (unsigned)i);
                bst = cand > bst ? cand : bst;
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                const unsigned long long o = __shfl_xor(bst, off, 64);
                bst = o > bst ? o : bst;
            }
            if ((tid & 63) == 0) best[tid >> 6] = bst;
            __syncthreads();
            if (tid == 0) {
                unsigned long long b = 0ull;
                for (int w = 0; w < (int)(blockDim.x >> 6); ++w) b = best[w] > b ? best[w] : b;
                s_tok = (int)(0xFFFFFFFFu - (unsigned)(b & 0xFFFFFFFFu));
            }
        }
        __syncthreads();
        tok = s_tok;
    }

    if (standalone) {
        if (tid == 0) *a.token_out = tok;
        return;
    }

    // ---- bookkeeping of decode_n_tokens (fast_inference_utils.py:148-174), on the device -----------------------
    bool done = false;
    if (!forced && tok == st.eoa_token) done = true;   // the EOA token itself is part of the sequence (:161)
    if (next_index + 1 >= st.max_len) done = true;     // sequence reached prompt + max_new_tokens / block_size
    if (tid == 0) {
        if (!forced) a.seq[(size_t)u * a.block_size + next_index] = tok;
        UttState* g = a.utt + u;
        g->n_tokens = next_index + 1;
        g->pos = next_index;
        if (a.pos_arr) a.pos_arr[u] = next_index;
        if (done) {
            g->done = 1;
            atomicSub(a.n_active, 1);
        }
    }
    if (done) return;

    // ---- next step's input rows: bf16(bf16(tok_emb[tok] + pos_emb[pos]) + spk_proj * mask)  (fast_model.py:152-157)
    const int dim = a.dim;
    const bf16_t* te = a.tok_emb + (size_t)tok * dim;
    const bf16_t* pe = a.pos_emb + (size_t)next_index * dim;
    const bf16_t* sp = a.spk_proj + (size_t)u * dim;
    bf16_t* x0 = a.x_next + (size_t)(2 * u) * dim;
    bf16_t* x1 = x0 + dim;
    for (int i = tid; i < dim; i += blockDim.x) {
        const float base = round_bf(bf2f(te[i]) + bf2f(pe[i]));
        x0[i] = f2bf(base + bf2f(sp[i]));
        x1[i] = f2bf(base);  // + 0 * mask: the unconditioned row gets no speaker term
    }
}

// x for the very first position of each utterance (token prompt[0] at position 0).
__global__ void embed_first_kernel(const SampleArgs a) {
    const int u = blockIdx.x;
    const int tok = a.seq[(size_t)u * a.block_size];
    const int dim = a.dim;
    const bf16_t* te = a.tok_emb + (size_t)tok * dim;
    const bf16_t* pe = a.pos_emb;
    const bf16_t* sp = a.spk_proj + (size_t)u * dim;
    bf16_t* x0 = a.x_next + (size_t)(2 * u) * dim;
    bf16_t* x1 = x0 + dim;
    for (int i = threadIdx.x; i < dim; i += blockDim.x) {
        const float base = round_bf(bf2f(te[i]) + bf2f(pe[i]));
        x0[i] = f2bf(base + bf2f(sp[i]));
        x1[i] = f2bf(base);
    }
}

// spk_proj[u][n] = bf16(sum_k spk_emb[u][k] * W[n][k])  -- `speaker_cond_pos` (fast_model.py:123,156).  It is
// step-invariant, so it is hoisted out of the decode step (the reference recomputes it every token).
__global__ void spk_proj_kernel(int dim, int spk_dim, const bf16_t* spk_emb, const bf16_t* w, bf16_t* out) {
    const int u = blockIdx.y;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (wave >= dim) return;
    float acc = 0.f;
    for (int k = lane; k < spk_dim; k += 64) acc += bf2f(spk_emb[(size_t)u * spk_dim + k]) * bf2f(w[(size_t)wave * spk_dim + k]);
    acc = wave_sum(acc);
    if (lane == 0) out[(size_t)u * dim + wave] = f2bf(acc);
}

hipError_t launch_sample(hipStream_t s, const SampleArgs& a, bool standalone) {
    if (a.vocab > SAMPLE_MAXV || a.vocab < 2) return hipErrorInvalidValue;
    hipLaunchKernelGGL(sample_kernel, dim3(standalone ? 1 : a.n_utt), dim3(SAMPLE_THREADS), 0, s, a);
    return hipGetLastError();
}
hipError_t launch_embed_first(hipStream_t s, const SampleArgs& a) {
    hipLaunchKernelGGL(embed_first_kernel, dim3(a.n_utt), dim3(256), 0, s, a);
    return hipGetLastError();
}
hipError_t launch_spk_proj(hipStream_t s, int n_utt, int dim, int spk_dim, const bf16_t* spk_emb, const bf16_t* w,
                           bf16_t* out) {
    const int waves_per_block = 4;
    hipLaunchKernelGGL(spk_proj_kernel, dim3((dim + waves_per_block - 1) / waves_per_block, n_utt), dim3(64 * waves_per_block),
                       0, s, dim, spk_dim, spk_emb, w, out);
    return hipGetLastError();
}
