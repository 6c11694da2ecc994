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
//   2. top-k / top-p are defined on the ascending (value, id) order (stable sort, fast_inference_utils.py:68-82,95-97)
//      but never need the sorted array: after step 1 every value is a bf16, i.e. a 16-bit order-preserving key.
//      top-k: the k-th largest key by base-4 bisection on block-wide counts.  top-p: "remove while the ascending
//      cumulative mass <= 1 - top_p" == remove every token with key < X, where X is the largest key whose
//      strictly-below mass F(X) <= thr (base-4 bisection on block-wide sums, 8 rounds of 3 sums), plus the first m
//      (by id) of the tokens AT key X while F(X) + m*e_X <= thr; the largest (value, id) is always kept.  All sums
//      run in a fixed order (deterministic); ~13 block reductions instead of the 78 barrier-separated passes of a
//      4096-key bitonic sort (40 us -> 6 us per step, profiles/).
//   3. softmax over the survivors (exp against the global max, renormalise).
//   4. Exp(1) race: argmax p_i / q_i with q from Philox4x32-10 keyed by the utterance's seed, counter
//      (token id, sequence index) -- independent of the slot the utterance occupies; ties to the lowest id.
#include "common.h"
#include "kernels.h"

#define SAMPLE_THREADS 512
#define SAMPLE_WAVES (SAMPLE_THREADS / 64)
#define SAMPLE_MAXV 4096
#define SAMPLE_PER (SAMPLE_MAXV / SAMPLE_THREADS)

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

// bf16-valued float <-> 16-bit key with the same ordering
__device__ __forceinline__ uint32_t bf_order_key(float f) {
    const uint32_t u = __float_as_uint(f);
    return ((u & 0x80000000u) ? ~u : (u | 0x80000000u)) >> 16;
}
__device__ __forceinline__ float bf_key_value(uint32_t k) {
    const uint32_t k32 = k << 16;
    return (k32 & 0x80000000u) ? __uint_as_float(k32 & 0x7FFFFFFFu) : __uint_as_float(~(k32 | 0xFFFFu));
}

// Block-wide reductions over SAMPLE_THREADS threads.  `buf` alternates between two LDS regions from one call to the next
// (the caller flips `par`), so one barrier per reduction is enough: a region is rewritten only after every thread has
// passed the barrier that followed its last read.
struct SampleScratch {
    float f[2][3][SAMPLE_WAVES];
    unsigned long long q[2][SAMPLE_WAVES];
};
template <int NV>
__device__ __forceinline__ void block_sum_f(float (&v)[NV], SampleScratch& sc, int& par) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const float t = wave_sum(v[j]);
        if (lane == 0) sc.f[par][j][wave] = t;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < SAMPLE_WAVES; ++w) t += sc.f[par][j][w];
        v[j] = t;
    }
    par ^= 1;
}
__device__ __forceinline__ unsigned long long block_max_q(unsigned long long v, SampleScratch& sc, int& par) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const unsigned long long o = __shfl_xor(v, off, 64);
        v = o > v ? o : v;
    }
    if (lane == 0) sc.q[par][wave] = v;
    __syncthreads();
    unsigned long long t = 0ull;
#pragma unroll
    for (int w = 0; w < SAMPLE_WAVES; ++w) t = sc.q[par][w] > t ? sc.q[par][w] : t;
    par ^= 1;
    return t;
}

__global__ void __launch_bounds__(SAMPLE_THREADS) sample_kernel(const SampleArgs a) {
    __shared__ SampleScratch sc;
    int par = 0;

    const int u = blockIdx.x;
    const int tid = threadIdx.x;
    const int V = a.vocab;
    UttState st = a.utt[u];
    const bool standalone = (a.token_out != nullptr);
    // The hand-off epoch of decode_mega.hip advances once per LAUNCH, unconditionally: a replay after the utterance has ended
    // (the trailing launches of an mv_run chunk) must not leave the granules tagged with an epoch the next utterance's first
    // launch would read again.
    if (!standalone && a.step_counter != nullptr && u == 0 && tid == 0) a.step_counter[0] += 1u;
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
        const bool use_p = sp.top_p >= 0.f;  // None is encoded as a negative value; top_p = 0 keeps only the largest (reference: cum <= 1 - 0)
        const int npt = (V + SAMPLE_THREADS - 1) / SAMPLE_THREADS;  // thread t owns ids [t*npt, t*npt + npt)
        const int i0 = tid * npt;

        // ---- 1. CFG mix + temperature (bf16 rounding points of fast_inference_utils.py:115-117,92)
        float val[SAMPLE_PER], e[SAMPLE_PER];
        uint32_t key[SAMPLE_PER];
        bool valid[SAMPLE_PER];
        unsigned long long best = 0ull;
#pragma unroll
        for (int k = 0; k < SAMPLE_PER; ++k) {
            const int i = i0 + k;
            valid[k] = (k < npt) && (i < V);
            val[k] = 0.f;
            key[k] = 0u;
            if (valid[k]) {
                const float c = bf2f(lc[i]), un = bf2f(lu[i]);
                const float mixed = round_bf(round_bf(sp.guidance_scale * c) + round_bf(sp.one_minus_guidance * un));
                float t = round_bf(mixed / temp);
                t = (t == 0.f) ? 0.f : t;  // -0 sorts with +0 (stable sort compares values)
                val[k] = t;
                key[k] = bf_order_key(t);
                const unsigned long long cand = ((unsigned long long)key[k] << 32) | (unsigned)i;
                best = cand > best ? cand : best;
            }
        }
        best = block_max_q(best, sc, par);
        const uint32_t kmax = (uint32_t)(best >> 32);
        const int last_id = (int)(best & 0xFFFFFFFFu);  // last element of the ascending (value, id) order: always kept
        const float vmax = bf_key_value(kmax);

        // ---- 2. top-k pivot: the k-th largest key (tokens equal to the pivot all survive, `logits < pivot` is masked)
        uint32_t pivot = 0u;
        if (use_k) {
            const int kk = min(sp.top_k, V);
            if (kk == 1) {
                pivot = kmax;
            } else {
                for (int lvl = 7; lvl >= 0; --lvl) {  // largest X with #{key >= X} >= kk
                    const uint32_t step = 1u << (2 * lvl);
                    float cnt[3] = {0.f, 0.f, 0.f};  // counts <= 4096 are exact in fp32
#pragma unroll
                    for (int k = 0; k < SAMPLE_PER; ++k)
#pragma unroll
                        for (int j = 0; j < 3; ++j) cnt[j] += (valid[k] && key[k] >= pivot + (j + 1) * step) ? 1.f : 0.f;
                    block_sum_f(cnt, sc, par);
                    uint32_t d = 0;
                    if (cnt[0] >= (float)kk) d = 1;
                    if (cnt[1] >= (float)kk) d = 2;
                    if (cnt[2] >= (float)kk) d = 3;
                    pivot += d * step;
                }
            }
        }

        // ---- 3. e_i = exp(v_i - max) (0 when masked by top-k), total mass
        float S1[1] = {0.f};
#pragma unroll
        for (int k = 0; k < SAMPLE_PER; ++k) {
            e[k] = (valid[k] && key[k] >= pivot) ? __expf(val[k] - vmax) : 0.f;
            S1[0] += e[k];
        }
        block_sum_f(S1, sc, par);
        const float S = S1[0];
        bool keep[SAMPLE_PER];
#pragma unroll
        for (int k = 0; k < SAMPLE_PER; ++k) keep[k] = valid[k] && (e[k] > 0.f || !use_k);

        // ---- 4. top-p on the ascending order, without sorting
        if (use_p) {
            const float thr = 1.0f - sp.top_p;
            uint32_t X = 0u;
            float FX = 0.f;
            for (int lvl = 7; lvl >= 0; --lvl) {  // largest X with F(X) = mass{key < X} <= thr
                const uint32_t step = 1u << (2 * lvl);
                float f[3] = {0.f, 0.f, 0.f};
#pragma unroll
                for (int k = 0; k < SAMPLE_PER; ++k)
#pragma unroll
                    for (int j = 0; j < 3; ++j) f[j] += (key[k] < X + (j + 1) * step) ? e[k] : 0.f;
                block_sum_f(f, sc, par);
                uint32_t d = 0;
                if (f[0] / S <= thr) { d = 1; FX = f[0]; }
                if (f[1] / S <= thr) { d = 2; FX = f[1]; }
                if (f[2] / S <= thr) { d = 3; FX = f[2]; }
                X += d * step;
            }
            // tokens AT key X, in id order: the m-th (1-based) has cumulative mass F(X) + m * e_X
            float tcnt = 0.f;
#pragma unroll
            for (int k = 0; k < SAMPLE_PER; ++k) tcnt += (valid[k] && key[k] == X) ? 1.f : 0.f;
            float incl = tcnt;  // inclusive scan over threads (= id order: thread t owns a contiguous id range)
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const float n = __shfl_up(incl, off, 64);
                if ((tid & 63) >= off) incl += n;
            }
            if ((tid & 63) == 63) sc.f[par][0][tid >> 6] = incl;
            __syncthreads();
            float before = incl - tcnt;
            for (int w = 0; w < (tid >> 6); ++w) before += sc.f[par][0][w];
            par ^= 1;
#pragma unroll
            for (int k = 0; k < SAMPLE_PER; ++k) {
                if (!valid[k]) continue;
                if (key[k] < X) keep[k] = false;
                if (key[k] == X) {
                    before += 1.f;
                    if ((FX + before * e[k]) / S <= thr) keep[k] = false;
                }
                if (i0 + k == last_id) keep[k] = valid[k];  // "always keep the last" (fast_inference_utils.py:76)
            }
        }

        // ---- renormalise over the survivors
        float S2a[1] = {0.f};
#pragma unroll
        for (int k = 0; k < SAMPLE_PER; ++k) {
            if (!keep[k]) e[k] = 0.f;
            S2a[0] += e[k];
        }
        block_sum_f(S2a, sc, par);
        const float S2 = S2a[0];

        // ---- 5. Exp(1) race: argmax p_i / q_i, ties to the lowest id
        unsigned long long bst = 0ull;
#pragma unroll
        for (int k = 0; k < SAMPLE_PER; ++k) {
            if (!valid[k]) continue;
            const unsigned id = (unsigned)(i0 + k);
            const float p = e[k] / S2;
            if (a.probs_out) a.probs_out[id] = p;
            const float r = (p > 0.f) ? p / mv_exp1_noise(sp.seed, (uint32_t)next_index, 0u, id) : 0.f;
            const unsigned long long cand = ((unsigned long long)__float_as_uint(r) << 32) | (0xFFFFFFFFu - id);
            bst = cand > bst ? cand : bst;
        }
        bst = block_max_q(bst, sc, par);
        tok = (int)(0xFFFFFFFFu - (unsigned)(bst & 0xFFFFFFFFu));
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

// x rows of every live utterance at its CURRENT position (token seq[pos] at position pos): rebuilds the decode-step input after
// a rollback (mv_run: a persistent-kernel hand-off timed out and the chunk is redone with per-layer launches).
__global__ void embed_current_kernel(const SampleArgs a) {
    const int u = blockIdx.x;
    const UttState st = a.utt[u];
    if (st.done) return;
    const int tok = a.seq[(size_t)u * a.block_size + st.pos];
    const int dim = a.dim;
    const bf16_t* te = a.tok_emb + (size_t)tok * dim;
    const bf16_t* pe = a.pos_emb + (size_t)st.pos * dim;
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
__global__ void spk_proj_kernel(int dim, int spk_dim, const bf16_t* spk_emb, const bf16_t* w, const bf16_t* scales, bf16_t* out) {
    const int u = blockIdx.y;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (wave >= dim) return;
    float acc = 0.f;
    for (int k = lane; k < spk_dim; k += 64) acc += bf2f(spk_emb[(size_t)u * spk_dim + k]) * bf2f(w[(size_t)wave * spk_dim + k]);
    acc = wave_sum(acc);
    // int8 mode: w holds the int8 codes (exact in bf16) and the Linear is bf16(bf16(x.q^T) * scale)  (fast_quantize.py:197-198)
    if (lane == 0) out[(size_t)u * dim + wave] = scales ? f2bf(round_bf(acc) * bf2f(scales[wave])) : f2bf(acc);
}
__global__ void i8_to_bf16_kernel(const signed char* in, bf16_t* out, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = f2bf((float)in[i]);
}
hipError_t launch_i8_to_bf16(hipStream_t s, const signed char* in, bf16_t* out, size_t n) {
    hipLaunchKernelGGL(i8_to_bf16_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, in, out, n);
    return hipGetLastError();
}

// Prefill input rows: x[(r*T + t)] for the conditioned (r=0) and unconditioned (r=1) copy of the prompt (fast_model.py:152-157).
__global__ void embed_rows_kernel(const int32_t* seq, int T, int dim, const bf16_t* tok_emb, const bf16_t* pos_emb,
                                  const bf16_t* spk, bf16_t* out) {
    const int t = blockIdx.x;
    const bf16_t* te = tok_emb + (size_t)seq[t] * dim;
    const bf16_t* pe = pos_emb + (size_t)t * dim;
    bf16_t* x0 = out + (size_t)t * dim;
    bf16_t* x1 = out + ((size_t)T + t) * dim;
    for (int i = threadIdx.x; i < dim; i += blockDim.x) {
        const float base = round_bf(bf2f(te[i]) + bf2f(pe[i]));
        x0[i] = f2bf(base + bf2f(spk[i]));
        x1[i] = f2bf(base);
    }
}
hipError_t launch_embed_rows(hipStream_t s, const int32_t* seq, int T, int dim, const bf16_t* tok_emb, const bf16_t* pos_emb,
                             const bf16_t* spk, bf16_t* out) {
    hipLaunchKernelGGL(embed_rows_kernel, dim3(T), dim3(256), 0, s, seq, T, dim, tok_emb, pos_emb, spk, out);
    return hipGetLastError();
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
hipError_t launch_embed_current(hipStream_t s, const SampleArgs& a) {
    hipLaunchKernelGGL(embed_current_kernel, dim3(a.n_utt), dim3(256), 0, s, a);
    return hipGetLastError();
}
hipError_t launch_spk_proj(hipStream_t s, int n_utt, int dim, int spk_dim, const bf16_t* spk_emb, const bf16_t* w,
                           const bf16_t* scales, bf16_t* out) {
    const int waves_per_block = 4;
    hipLaunchKernelGGL(spk_proj_kernel, dim3((dim + waves_per_block - 1) / waves_per_block, n_utt), dim3(64 * waves_per_block),
                       0, s, dim, spk_dim, spk_emb, w, scales, out);
    return hipGetLastError();
}
