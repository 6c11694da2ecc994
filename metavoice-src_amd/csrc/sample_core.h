// The sampler's arithmetic, shared by sample_kernel (sampling.hip: one launch per decode step) and the persistent decode kernel
// (decode_mega.hip: the workgroup of utterance u draws the token at the end of every step, inside the launch).  One 512-thread
// workgroup per utterance; every reduction runs in a fixed order, so both callers draw the same token from the same logits.
// Test infrastructure reads oracle/sampling.py for the reference semantics (fam/llm/fast_inference_utils.py:61-120).
#pragma once
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
__device__ __forceinline__ void block_sum_f(float (&v)[NV], SampleScratch& sc, int& par, const int tid) {
    const int lane = tid & 63, wave = tid >> 6;
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
__device__ __forceinline__ unsigned long long block_max_q(unsigned long long v, SampleScratch& sc, int& par, const int tid) {
    const int lane = tid & 63, wave = tid >> 6;
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


// Steps 1-5 of sampling.hip's header: CFG mix + temperature -> top-k -> top-p -> softmax -> Exp(1) race.  `fetch(i, c, un)` returns
// the conditioned / unconditioned bf16 logit of token i as floats.  Every thread of the (SAMPLE_THREADS-wide) workgroup calls it
// and every thread returns the same token.  `tid` = threadIdx.x (a parameter so that a caller inside a long-running loop can hand in an
// id the optimiser cannot hoist address arithmetic on).
template <typename Fetch>
__device__ __forceinline__ int sample_draw(const mv_sampling sp, const int V, const int next_index, float* probs_out, SampleScratch& sc,
                                           int& par, const int tid, Fetch fetch) {
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
            float c, un;
            fetch(i, c, un);
            const float mixed = round_bf(round_bf(sp.guidance_scale * c) + round_bf(sp.one_minus_guidance * un));
            float t = round_bf(mixed / temp);
            t = (t == 0.f) ? 0.f : t;  // -0 sorts with +0 (stable sort compares values)
            val[k] = t;
            key[k] = bf_order_key(t);
            const unsigned long long cand = ((unsigned long long)key[k] << 32) | (unsigned)i;
            best = cand > best ? cand : best;
        }
    }
    best = block_max_q(best, sc, par, tid);
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
                block_sum_f(cnt, sc, par, tid);
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
    block_sum_f(S1, sc, par, tid);
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
            block_sum_f(f, sc, par, tid);
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
    block_sum_f(S2a, sc, par, tid);
    const float S2 = S2a[0];

    // ---- 5. Exp(1) race: argmax p_i / q_i, ties to the lowest id
    unsigned long long bst = 0ull;
#pragma unroll
    for (int k = 0; k < SAMPLE_PER; ++k) {
        if (!valid[k]) continue;
        const unsigned id = (unsigned)(i0 + k);
        const float p = e[k] / S2;
        if (probs_out) probs_out[id] = p;
        const float r = (p > 0.f) ? p / mv_exp1_noise(sp.seed, (uint32_t)next_index, 0u, id) : 0.f;
        const unsigned long long cand = ((unsigned long long)__float_as_uint(r) << 32) | (0xFFFFFFFFu - id);
        bst = cand > bst ? cand : bst;
    }
    bst = block_max_q(bst, sc, par, tid);
    return (int)(0xFFFFFFFFu - (unsigned)(bst & 0xFFFFFFFFu));
}
