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
// Work decomposition: (row, head) x ceil(P/128) blocks of 128 consecutive keys; a wave owns 32 keys and issues all of
// its 16 K/V loads before touching any (the whole live KV of a step is in flight in ONE round trip: 2048 keys x 32
// (row, head) = 2048 waves x 16 KB).  Partial (m, l, o[128]) go to a small fp32 scratch and the LAST block to arrive
// for a (row, head) combines them (8-byte agent-scope accesses on both sides, cdna_hip_programming.md Guideline 16)
// -- this saves a separate combine launch on the critical path of every layer.
#include "common.h"
#include "kernels.h"

#define ATTN_THREADS 256
#define ATTN_WAVES 4
#define ATTN_KEYS_PER_WAVE 32
#define ATTN_KEYS_PER_BLOCK (ATTN_WAVES * ATTN_KEYS_PER_WAVE)   // 128 keys = 64 KB of K+V per block, all in flight at once

typedef unsigned long long u64;
__device__ __forceinline__ void store_agent(float* p, float a, float b) {
    // 8-byte write-through (sc1) store: visible to every XCD without a release fence (Guideline 16, R1/R2 form)
    const u64 v = (u64)__float_as_uint(a) | ((u64)__float_as_uint(b) << 32);
    __hip_atomic_store((u64*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void load_agent(const float* p, float& a, float& b) {
    const u64 v = __hip_atomic_load((const u64*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    a = __uint_as_float((uint32_t)v);
    b = __uint_as_float((uint32_t)(v >> 32));
}

__global__ void __launch_bounds__(ATTN_THREADS) attn_decode_kernel(const AttnArgs a) {
    __shared__ __attribute__((aligned(16))) float sm[ATTN_WAVES][16][12];  // per wave, per dim-group: m, l, o[8] (+pad)

    const int split = blockIdx.x, h = blockIdx.y, r = blockIdx.z;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane >> 4, dl = lane & 15;

    const int P = a.pos_is_count ? a.pos[r >> 1] : a.pos[r >> 1] + 1;  // keys [0, P)
    const int nact = (P + ATTN_KEYS_PER_BLOCK - 1) / ATTN_KEYS_PER_BLOCK;
    if (split >= nact) return;
    const int s0 = split * ATTN_KEYS_PER_BLOCK + wave * ATTN_KEYS_PER_WAVE;   // first key of this wave
    const int s1 = min(P, split * ATTN_KEYS_PER_BLOCK + ATTN_KEYS_PER_BLOCK);  // end of this block's keys

    const size_t head_base = ((size_t)r * a.n_head + h) * (size_t)a.cache_len * 128;
    const bf16_t* kb = a.kcache + head_base + dl * 8;
    const bf16_t* vb = a.vcache + head_base + dl * 8;

    // ---- 1. the wave's whole K and V working set goes in flight at once (16 x 16 B per lane) -------------
    constexpr int U = ATTN_KEYS_PER_WAVE / 4;  // a wave instruction covers 4 keys (16 lanes x 16 B each)
    u32x4 kv[U], vv[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int kc = min(s0 + u * 4 + g, P - 1);  // clamped address, contribution masked below
        kv[u] = load_stream16(kb + (size_t)kc * 128);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int kc = min(s0 + u * 4 + g, P - 1);
        vv[u] = load_stream16(vb + (size_t)kc * 128);
    }
    // q fragment for this lane's 8 dims (bf16 pairs kept packed for v_dot2c)
    const u32x4 qv = load16(a.q + (size_t)r * a.n_head * 128 + h * 128 + dl * 8);
    const float scale = 0.08838834764831845f;  // 1/sqrt(128)

    // ---- 2. scores of the lane group's 8 keys, softmax numerators against the group max, P.V -------------
    float sc[U];
    float m = -INFINITY;
#pragma unroll
    for (int u = 0; u < U; ++u) {
        float s = sum_across16(dot8_bf16(kv[u], qv, 0.f)) * scale;
        s = (s0 + u * 4 + g < s1) ? s : -INFINITY;
        sc[u] = s;
        m = fmaxf(m, s);
    }
    float l = 0.f;
    float o[8];
#pragma unroll
    for (int d = 0; d < 8; ++d) o[d] = 0.f;
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const float p = (sc[u] == -INFINITY) ? 0.f : fast_exp(sc[u] - m);
        l += p;
        o[0] += p * bf_lo(vv[u].x);
        o[1] += p * bf_hi(vv[u].x);
        o[2] += p * bf_lo(vv[u].y);
        o[3] += p * bf_hi(vv[u].y);
        o[4] += p * bf_lo(vv[u].z);
        o[5] += p * bf_hi(vv[u].z);
        o[6] += p * bf_lo(vv[u].w);
        o[7] += p * bf_hi(vv[u].w);
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
    if (wave != 0) return;  // the rest is wave 0: lane (g, dl) merges wave g's slice, then the 4 groups are folded

    // ---- merge the waves: group g takes wave g's (m, l, o[8]) for dims 8dl..8dl+7, xor-16/32 butterflies fold them ----
    float mb, lb, ob[8];
    {
        const float mw = sm[g][dl][0];
        mb = fmaxf(mw, __shfl_xor(mw, 16, 64));
        mb = fmaxf(mb, __shfl_xor(mb, 32, 64));
        const float f = (mw == -INFINITY) ? 0.f : fast_exp(mw - mb);
        lb = sm[g][dl][1] * f;
        lb += __shfl_xor(lb, 16, 64);
        lb += __shfl_xor(lb, 32, 64);
#pragma unroll
        for (int d = 0; d < 8; ++d) {
            float t = sm[g][dl][2 + d] * f;
            t += __shfl_xor(t, 16, 64);
            t += __shfl_xor(t, 32, 64);
            ob[d] = t;
        }
    }
    bf16_t* outp = a.out + (size_t)r * a.n_head * 128 + h * 128 + dl * 8;
    if (nact == 1) {  // single split: finish here, no hand-off needed
        if (g == 0) {
            const float inv = 1.0f / lb;
            u32x4 y;
            y.x = pack_bf2(ob[0] * inv, ob[1] * inv);
            y.y = pack_bf2(ob[2] * inv, ob[3] * inv);
            y.z = pack_bf2(ob[4] * inv, ob[5] * inv);
            y.w = pack_bf2(ob[6] * inv, ob[7] * inv);
            *(u32x4*)outp = y;
        }
        return;
    }

    // ---- publish the block partial write-through, take a ticket; the last arriver combines -----------------
    // Guideline 16 in its "8-byte agent-scope accesses on both sides" form: no release / acquire fence (each costs
    // ~1.7 us and drops the CU's L1); correct for any placement of a (row, head)'s splits over XCDs.
    float* pbase = a.partial + ((size_t)r * a.n_head + h) * ATTN_NSPLIT * ATTN_PART_STRIDE;
    float* part = pbase + split * ATTN_PART_STRIDE;
    if (lane == 0) store_agent(part, mb, lb);
    {   // one 8-byte store per lane: group g publishes dims 2g, 2g+1 of each 8-dim slice
        const float lo = g == 0 ? ob[0] : g == 1 ? ob[2] : g == 2 ? ob[4] : ob[6];
        const float hi = g == 0 ? ob[1] : g == 1 ? ob[3] : g == 2 ? ob[5] : ob[7];
        store_agent(part + 4 + dl * 8 + 2 * g, lo, hi);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's stores have left before the ticket is drawn
    unsigned prev = 0;
    if (lane == 0)
        prev = __hip_atomic_fetch_add(a.counters + r * a.n_head + h, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    prev = __builtin_amdgcn_readfirstlane(prev);
    if (prev != (unsigned)(nact - 1)) return;

    // Last arriver: group g reads splits g, g+4, g+8, g+12 -- every load is issued before the first use (one L2 round
    // trip, not one per split), then the groups are folded like the waves above.
    float ms[4], ls[4], t[4][8];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int sidx = min(g + 4 * q, nact - 1);  // clamped: out-of-range slots are weighted 0 below
        const float* pp = pbase + sidx * ATTN_PART_STRIDE;
        load_agent(pp, ms[q], ls[q]);
        load_agent(pp + 4 + dl * 8 + 0, t[q][0], t[q][1]);
        load_agent(pp + 4 + dl * 8 + 2, t[q][2], t[q][3]);
        load_agent(pp + 4 + dl * 8 + 4, t[q][4], t[q][5]);
        load_agent(pp + 4 + dl * 8 + 6, t[q][6], t[q][7]);
    }
    float mt = -INFINITY;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        if (g + 4 * q >= nact) ms[q] = -INFINITY;
        mt = fmaxf(mt, ms[q]);
    }
    mt = fmaxf(mt, __shfl_xor(mt, 16, 64));
    mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
    float lt = 0.f, ot[8];
#pragma unroll
    for (int d = 0; d < 8; ++d) ot[d] = 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float f = (ms[q] == -INFINITY) ? 0.f : fast_exp(ms[q] - mt);
        lt += ls[q] * f;
#pragma unroll
        for (int d = 0; d < 8; ++d) ot[d] += t[q][d] * f;
    }
    lt += __shfl_xor(lt, 16, 64);
    lt += __shfl_xor(lt, 32, 64);
#pragma unroll
    for (int d = 0; d < 8; ++d) {
        ot[d] += __shfl_xor(ot[d], 16, 64);
        ot[d] += __shfl_xor(ot[d], 32, 64);
    }
    if (g == 0) {
        const float inv = 1.0f / lt;
        u32x4 y;
        y.x = pack_bf2(ot[0] * inv, ot[1] * inv);
        y.y = pack_bf2(ot[2] * inv, ot[3] * inv);
        y.z = pack_bf2(ot[4] * inv, ot[5] * inv);
        y.w = pack_bf2(ot[6] * inv, ot[7] * inv);
        *(u32x4*)outp = y;
    }
    if (lane == 0)  // re-arm for the next launch; nobody else touches this counter until then (stream order)
        __hip_atomic_store(a.counters + r * a.n_head + h, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

hipError_t launch_attention_decode(hipStream_t s, int rows, const AttnArgs& a) {
    // cache_len <= ATTN_NSPLIT * ATTN_KEYS_PER_BLOCK is checked by the callers
    const int nsplit = min(ATTN_NSPLIT, (a.cache_len + ATTN_KEYS_PER_BLOCK - 1) / ATTN_KEYS_PER_BLOCK);
    hipLaunchKernelGGL(attn_decode_kernel, dim3(nsplit, a.n_head, rows), dim3(ATTN_THREADS), 0, s, a);
    return hipGetLastError();
}
