// Single-query (decode) attention over the static KV cache, split-KV "flash decoding" with an in-launch combine.
//
// Replaces `F.scaled_dot_product_attention(q, k_cache, v_cache, attn_mask=causal_mask[input_pos])`
// (fam/llm/fast_model.py:151,218-222) for S=1.  The reference attends over all `block_size` cache slots and masks
// the unwritten ones (805 MB of cache per step); algorithmically only keys [0, pos] matter, so this kernel reads
// exactly those: 2 (K,V) * 256 B per key per (row, head).
//
// Roofline: HBM (pure streaming of K and V, ~1 FLOP/B).  Layout [row][head][position][128] bf16 makes each
// (row, head) a contiguous stream; one key row is 256 B = 16 lanes x 16 B, so a wave covers 4 keys per
// instruction with fully coalesced `global_load_dwordx4`, the query lives in registers, scores are reduced over
// 16-lane groups with DPP, and softmax/PV run as an online (running max / sum) recurrence in fp32.
// Work decomposition: (row, head) x up to ATTN_NSPLIT contiguous key ranges so that all 256 CUs stream at long
// contexts; partial (m, l, o[128]) go to a small fp32 scratch and the LAST block to arrive for a (row, head)
// combines them (agent-scope release / acquire, cdna_hip_programming.md Guideline 16 counter form) -- this
// saves a separate combine launch on the critical path of every layer.
#include "common.h"
#include "kernels.h"

#define ATTN_THREADS 256
#define ATTN_WAVES 4
#define ATTN_MIN_KEYS_PER_SPLIT 64

__global__ void __launch_bounds__(ATTN_THREADS) attn_decode_kernel(const AttnArgs a) {
    __shared__ __attribute__((aligned(16))) float sm[ATTN_WAVES][16][12];  // per wave, per dim-group: m, l, o[8] (+pad)
    __shared__ int s_last;

    const int split = blockIdx.x, h = blockIdx.y, r = blockIdx.z;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane >> 4, dl = lane & 15;

    const int P = a.pos_is_count ? a.pos[r >> 1] : a.pos[r >> 1] + 1;  // keys [0, P)
    const int nact = min(ATTN_NSPLIT, max(1, (P + ATTN_MIN_KEYS_PER_SPLIT - 1) / ATTN_MIN_KEYS_PER_SPLIT));
    if (split >= nact) return;
    const int chunk = (((P + nact - 1) / nact) + 15) & ~15;
    const int s0 = split * chunk;
    const int s1 = min(P, s0 + chunk);

    const size_t head_base = ((size_t)r * a.n_head + h) * (size_t)a.cache_len * 128;
    const bf16_t* kb = a.kcache + head_base + dl * 8;
    const bf16_t* vb = a.vcache + head_base + dl * 8;

    // q fragment for this lane's 8 dims (bf16 pairs kept packed for v_dot2c)
    const u32x4 qv = load16(a.q + (size_t)r * a.n_head * 128 + h * 128 + dl * 8);
    const float scale = 0.08838834764831845f;  // 1/sqrt(128)

    float m = -INFINITY, l = 0.f;
    float o[8];
#pragma unroll
    for (int d = 0; d < 8; ++d) o[d] = 0.f;

    // key index handled by this lane group in iteration `it`: s0 + it*16 + wave*4 + g
    constexpr int UNROLL = 4;
    for (int base = s0; base < s1; base += 16 * UNROLL) {
        u32x4 kv[UNROLL], vv[UNROLL];
        int key[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            key[u] = base + u * 16 + wave * 4 + g;
            const int kc = min(key[u], s1 - 1);  // clamped address, contribution masked below
            kv[u] = load_stream16(kb + (size_t)kc * 128);
            vv[u] = load_stream16(vb + (size_t)kc * 128);
        }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            float s = dot8_bf16(kv[u], qv, 0.f);
            s = sum_across16(s) * scale;
            if (key[u] < s1) {
                const float mn = fmaxf(m, s);
                const float corr = fast_exp(m - mn);
                const float p = fast_exp(s - mn);
                l = l * corr + p;
                o[0] = o[0] * corr + p * bf_lo(vv[u].x);
                o[1] = o[1] * corr + p * bf_hi(vv[u].x);
                o[2] = o[2] * corr + p * bf_lo(vv[u].y);
                o[3] = o[3] * corr + p * bf_hi(vv[u].y);
                o[4] = o[4] * corr + p * bf_lo(vv[u].z);
                o[5] = o[5] * corr + p * bf_hi(vv[u].z);
                o[6] = o[6] * corr + p * bf_lo(vv[u].w);
                o[7] = o[7] * corr + p * bf_hi(vv[u].w);
                m = mn;
            }
        }
    }

    // ---- merge the 4 lane groups of the wave (lanes dl, dl+16, dl+32, dl+48 hold the same dims) ---------
    {
        float mo = fmaxf(m, __shfl_xor(m, 16, 64));
        mo = fmaxf(mo, __shfl_xor(mo, 32, 64));
        const float w = (m == -INFINITY) ? 0.f : fast_exp(m - mo);
        l *= w;
        l += __shfl_xor(l, 16, 64);
        l += __shfl_xor(l, 32, 64);
#pragma unroll
        for (int d = 0; d < 8; ++d) {
            float t = o[d] * w;
            t += __shfl_xor(t, 16, 64);
            t += __shfl_xor(t, 32, 64);
            o[d] = t;
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

    // ---- merge the waves; lanes 0..15 of wave 0 own 8 dims each and write the block partial ---------------
    float* part = a.partial + (((size_t)r * a.n_head + h) * ATTN_NSPLIT + split) * ATTN_PART_STRIDE;
    if (threadIdx.x < 16) {
        float mb = -INFINITY;
#pragma unroll
        for (int w = 0; w < ATTN_WAVES; ++w) mb = fmaxf(mb, sm[w][dl][0]);
        float lb = 0.f, ob[8];
#pragma unroll
        for (int d = 0; d < 8; ++d) ob[d] = 0.f;
#pragma unroll
        for (int w = 0; w < ATTN_WAVES; ++w) {
            const float mw = sm[w][dl][0];
            const float f = (mw == -INFINITY) ? 0.f : fast_exp(mw - mb);
            lb += sm[w][dl][1] * f;
#pragma unroll
            for (int d = 0; d < 8; ++d) ob[d] += sm[w][dl][2 + d] * f;
        }
        if (nact == 1) {
            // single split: finish here, no hand-off needed
            const float inv = 1.0f / lb;
            u32x4 y;
            y.x = pack_bf2(ob[0] * inv, ob[1] * inv);
            y.y = pack_bf2(ob[2] * inv, ob[3] * inv);
            y.z = pack_bf2(ob[4] * inv, ob[5] * inv);
            y.w = pack_bf2(ob[6] * inv, ob[7] * inv);
            *(u32x4*)(a.out + (size_t)r * a.n_head * 128 + h * 128 + dl * 8) = y;
        } else {
            if (dl == 0) { part[0] = mb; part[1] = lb; }
            *(float4*)(part + 4 + dl * 8) = make_float4(ob[0], ob[1], ob[2], ob[3]);
            *(float4*)(part + 4 + dl * 8 + 4) = make_float4(ob[4], ob[5], ob[6], ob[7]);
        }
    }
    if (nact == 1) return;

    // ---- publish, then the last arriver combines (Guideline 16, counter form) -----------------------------
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned prev = __hip_atomic_fetch_add(a.counters + r * a.n_head + h, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = (prev == (unsigned)(nact - 1));
        if (s_last) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
    if (!s_last) return;

    if (threadIdx.x < 16) {
        const float* pbase = a.partial + ((size_t)r * a.n_head + h) * ATTN_NSPLIT * ATTN_PART_STRIDE;
        float mt = -INFINITY;
        for (int sidx = 0; sidx < nact; ++sidx) mt = fmaxf(mt, pbase[sidx * ATTN_PART_STRIDE]);
        float lt = 0.f, ot[8];
#pragma unroll
        for (int d = 0; d < 8; ++d) ot[d] = 0.f;
        for (int sidx = 0; sidx < nact; ++sidx) {
            const float* pp = pbase + sidx * ATTN_PART_STRIDE;
            const float f = fast_exp(pp[0] - mt);
            lt += pp[1] * f;
            const float4 a0 = *(const float4*)(pp + 4 + dl * 8);
            const float4 a1 = *(const float4*)(pp + 4 + dl * 8 + 4);
            ot[0] += a0.x * f; ot[1] += a0.y * f; ot[2] += a0.z * f; ot[3] += a0.w * f;
            ot[4] += a1.x * f; ot[5] += a1.y * f; ot[6] += a1.z * f; ot[7] += a1.w * f;
        }
        const float inv = 1.0f / lt;
        u32x4 y;
        y.x = pack_bf2(ot[0] * inv, ot[1] * inv);
        y.y = pack_bf2(ot[2] * inv, ot[3] * inv);
        y.z = pack_bf2(ot[4] * inv, ot[5] * inv);
        y.w = pack_bf2(ot[6] * inv, ot[7] * inv);
        *(u32x4*)(a.out + (size_t)r * a.n_head * 128 + h * 128 + dl * 8) = y;
        if (threadIdx.x == 0) {
            // re-arm for the next launch; nobody else touches this counter until then (stream order)
            __hip_atomic_store(a.counters + r * a.n_head + h, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

hipError_t launch_attention_decode(hipStream_t s, int rows, const AttnArgs& a) {
    hipLaunchKernelGGL(attn_decode_kernel, dim3(ATTN_NSPLIT, a.n_head, rows), dim3(ATTN_THREADS), 0, s, a);
    return hipGetLastError();
}
