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
#include "sample_core.h"

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
        tok = sample_draw(st.s, V, next_index, a.probs_out, sc, par, tid, [&](int i, float& c, float& un) { c = bf2f(lc[i]); un = bf2f(lu[i]); });
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
