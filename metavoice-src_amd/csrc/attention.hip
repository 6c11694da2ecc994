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
// Work decomposition: (row, head) x up to 8 even key ranges of <= 256 keys (attn_core.h); a block is 8 waves, a wave owns
// up to 32 keys and issues all of its 16 K/V loads before touching any (the whole live KV of a step is in flight in ONE
// round trip: 2048 keys x 32 (row, head) = 2048 waves x 16 KB).  Up to 256 keys there is no cross-block combine.  Partial (m, l, o[128]) go to a small fp32 scratch and the LAST block to arrive
// for a (row, head) combines them (8-byte agent-scope accesses on both sides, cdna_hip_programming.md Guideline 16)
// -- this saves a separate combine launch on the critical path of every layer.
#include "attn_core.h"

#define ATTN_THREADS (64 * ATTN_WAVES)

__global__ void __launch_bounds__(ATTN_THREADS) attn_decode_kernel(const AttnArgs a) {
    __shared__ __attribute__((aligned(16))) float sm[ATTN_WAVES][16][12];  // per wave, per dim-group: m, l, o[8] (+pad)

    const int split = blockIdx.x, h = blockIdx.y, r = blockIdx.z;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane >> 4, dl = lane & 15;

    const int P = a.pos_is_count ? a.pos[r >> 1] : a.pos[r >> 1] + 1;  // keys [0, P)
    const int cs = attn_keys_per_split(P);
    const int nact = (P + cs - 1) / cs;
    if (split >= nact) return;
    const int kw = attn_keys_per_wave(cs);
    const int s0 = split * cs + wave * kw;                   // first key of this wave
    const int s1 = min(min(P, split * cs + cs), s0 + kw);  // end of this wave's keys

    const size_t head_base = ((size_t)r * a.n_head + h) * (size_t)a.cache_len * 128;
    const bf16_t* kb = a.kcache + head_base + dl * 8;
    const bf16_t* vb = a.vcache + head_base + dl * 8;

    // the wave's whole K and V working set goes in flight at once (up to 16 x 16 B per lane); clamped addresses, masked later
    u32x4 kv[ATTN_U], vv[ATTN_U];
#pragma unroll
    for (int u = 0; u < ATTN_U; ++u) kv[u] = u * 4 < kw ? load_stream16(kb + (size_t)min(s0 + u * 4 + g, P - 1) * 128) : u32x4{0u, 0u, 0u, 0u};
#pragma unroll
    for (int u = 0; u < ATTN_U; ++u) vv[u] = u * 4 < kw ? load_stream16(vb + (size_t)min(s0 + u * 4 + g, P - 1) * 128) : u32x4{0u, 0u, 0u, 0u};
    const u32x4 qv = load16(a.q + (size_t)r * a.n_head * 128 + h * 128 + dl * 8);

    const AttnAcc w = attn_wave_partial(kv, vv, qv, s0, s1, g);
    if (g == 0) {
        sm[wave][dl][0] = w.m;
        sm[wave][dl][1] = w.l;
#pragma unroll
        for (int d = 0; d < 8; ++d) sm[wave][dl][2 + d] = w.o[d];
    }
    __syncthreads();
    if (wave != 0) return;

    AttnAcc b = attn_merge_waves(sm, g, dl);  // folds the 8 waves
    bf16_t* outp = a.out + (size_t)r * a.n_head * 128 + h * 128 + dl * 8;
    if (nact > 1) {
        float* pbase = a.partial + ((size_t)r * a.n_head + h) * ATTN_NSPLIT * ATTN_PART_STRIDE;
        if (!attn_publish_combine(b, pbase, a.counters + r * a.n_head + h, split, nact, lane)) return;
    }
    if (g == 0) *(u32x4*)outp = attn_normalise(b);
}

hipError_t launch_attention_decode(hipStream_t s, int rows, const AttnArgs& a) {
    hipLaunchKernelGGL(attn_decode_kernel, dim3(ATTN_NSPLIT, a.n_head, rows), dim3(ATTN_THREADS), 0, s, a);
    return hipGetLastError();
}
