// Decode-attention building blocks shared by the per-layer launch (attention.hip) and the persistent decode-step kernel
// (decode_mega.hip).  Both paths run EXACTLY this arithmetic in the same order, so they produce identical bits.  The
// library is built with -ffp-contract=off and every fused multiply-add is an explicit fmaf(): left to the compiler, the
// SAME source expression was fused in one kernel and not in the other (1-ulp fp32 differences, a bf16 flip every ~1e6
// outputs).
//
// Key partition of a (row, head): ATTN_NSPLIT = 16 contiguous ranges of `cs` keys, cs = ceil(P/16) rounded up to a
// multiple of 4 (a wave instruction covers 4 keys: 16 lanes x 16 B per 256-byte key row); only the first
// nact = ceil(P/cs) ranges are non-empty.  Inside a range 4 waves take 32 keys each (cs <= 128 for P <= 2048).
#pragma once
#include "common.h"
#include "kernels.h"

#define ATTN_WAVES 4
#define ATTN_KEYS_PER_WAVE 32
#define ATTN_U (ATTN_KEYS_PER_WAVE / 4)

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

__device__ __forceinline__ int attn_keys_per_split(int P) { return (((P + ATTN_NSPLIT - 1) / ATTN_NSPLIT) + 3) & ~3; }

struct AttnAcc {
    float m, l, o[8];
};

// Scores of the lane group's ATTN_U keys against q, softmax numerators against the group max, P.V; then the 4 lane
// groups of the wave (lanes dl, dl+16, dl+32, dl+48 hold the same 8 dims) are folded.  Key index of (u, g) = s0+4u+g,
// valid while < s1.
__device__ __forceinline__ AttnAcc attn_wave_partial(const u32x4 (&kv)[ATTN_U], const u32x4 (&vv)[ATTN_U], const u32x4& qv, int s0,
                                                     int s1, int g) {
    const float scale = 0.08838834764831845f;  // 1/sqrt(128)
    float sc[ATTN_U];
    AttnAcc a;
    a.m = -INFINITY;
#pragma unroll
    for (int u = 0; u < ATTN_U; ++u) {
        float s = sum_across16(dot8_bf16(kv[u], qv, 0.f)) * scale;
        s = (s0 + u * 4 + g < s1) ? s : -INFINITY;
        sc[u] = s;
        a.m = fmaxf(a.m, s);
    }
    a.l = 0.f;
#pragma unroll
    for (int d = 0; d < 8; ++d) a.o[d] = 0.f;
#pragma unroll
    for (int u = 0; u < ATTN_U; ++u) {
        const float p = (sc[u] == -INFINITY) ? 0.f : fast_exp(sc[u] - a.m);
        a.l += p;
        a.o[0] = fmaf(p, bf_lo(vv[u].x), a.o[0]);
        a.o[1] = fmaf(p, bf_hi(vv[u].x), a.o[1]);
        a.o[2] = fmaf(p, bf_lo(vv[u].y), a.o[2]);
        a.o[3] = fmaf(p, bf_hi(vv[u].y), a.o[3]);
        a.o[4] = fmaf(p, bf_lo(vv[u].z), a.o[4]);
        a.o[5] = fmaf(p, bf_hi(vv[u].z), a.o[5]);
        a.o[6] = fmaf(p, bf_lo(vv[u].w), a.o[6]);
        a.o[7] = fmaf(p, bf_hi(vv[u].w), a.o[7]);
    }
    float mo = fmaxf(a.m, __shfl_xor(a.m, 16, 64));
    mo = fmaxf(mo, __shfl_xor(mo, 32, 64));
    const float w = (a.m == -INFINITY) ? 0.f : fast_exp(a.m - mo);
    a.l *= w;
    a.l += __shfl_xor(a.l, 16, 64);
    a.l += __shfl_xor(a.l, 32, 64);
#pragma unroll
    for (int d = 0; d < 8; ++d) {
        float t = a.o[d] * w;
        t += __shfl_xor(t, 16, 64);
        t += __shfl_xor(t, 32, 64);
        a.o[d] = t;
    }
    a.m = mo;
    return a;
}

// `sm` = [4 waves][16 dim groups][12 floats] of per-wave (m, l, o[8]); executed by ONE whole wave: lane (g, dl) takes
// wave g's slice for dims 8dl..8dl+7 and the groups are folded with xor-16/32 butterflies.
__device__ __forceinline__ AttnAcc attn_merge_waves(const float (*sm)[16][12], int g, int dl) {
    AttnAcc b;
    const float mw = sm[g][dl][0];
    b.m = fmaxf(mw, __shfl_xor(mw, 16, 64));
    b.m = fmaxf(b.m, __shfl_xor(b.m, 32, 64));
    const float f = (mw == -INFINITY) ? 0.f : fast_exp(mw - b.m);
    b.l = sm[g][dl][1] * f;
    b.l += __shfl_xor(b.l, 16, 64);
    b.l += __shfl_xor(b.l, 32, 64);
#pragma unroll
    for (int d = 0; d < 8; ++d) {
        float t = sm[g][dl][2 + d] * f;
        t += __shfl_xor(t, 16, 64);
        t += __shfl_xor(t, 32, 64);
        b.o[d] = t;
    }
    return b;
}

__device__ __forceinline__ u32x4 attn_normalise(const AttnAcc& b) {
    const float inv = 1.0f / b.l;
    u32x4 y;
    y.x = pack_bf2(b.o[0] * inv, b.o[1] * inv);
    y.y = pack_bf2(b.o[2] * inv, b.o[3] * inv);
    y.z = pack_bf2(b.o[4] * inv, b.o[5] * inv);
    y.w = pack_bf2(b.o[6] * inv, b.o[7] * inv);
    return y;
}

// Executed by one whole wave holding the block result `b` (identical in the 4 lane groups).  Publishes the split
// partial write-through, draws a ticket; returns true in the LAST arriver of the (row, head), with `b` replaced by the
// combination of all `nact` partials (every partial load is issued before the first use: one L2 round trip).
// Guideline 16 in its "8-byte agent-scope accesses on both sides" form: no release / acquire fence; correct for any
// placement of the splits over XCDs; the counter re-arms itself.
__device__ __forceinline__ bool attn_publish_combine(AttnAcc& b, float* pbase, uint32_t* counter, int split, int nact, int lane) {
    const int g = lane >> 4, dl = lane & 15;
    float* part = pbase + split * ATTN_PART_STRIDE;
    if (lane == 0) store_agent(part, b.m, b.l);
    {   // one 8-byte store per lane: group g publishes dims 2g, 2g+1 of each 8-dim slice
        const float lo = g == 0 ? b.o[0] : g == 1 ? b.o[2] : g == 2 ? b.o[4] : b.o[6];
        const float hi = g == 0 ? b.o[1] : g == 1 ? b.o[3] : g == 2 ? b.o[5] : b.o[7];
        store_agent(part + 4 + dl * 8 + 2 * g, lo, hi);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's stores have left before the ticket is drawn
    unsigned prev = 0;
    if (lane == 0) prev = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    prev = __builtin_amdgcn_readfirstlane(prev);
    if (prev != (unsigned)(nact - 1)) return false;

    float ms[4], ls[4], t[4][8];
#pragma unroll
    for (int q = 0; q < 4; ++q) {  // group g reads splits g, g+4, g+8, g+12
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
        lt = fmaf(ls[q], f, lt);
#pragma unroll
        for (int d = 0; d < 8; ++d) ot[d] = fmaf(t[q][d], f, ot[d]);
    }
    lt += __shfl_xor(lt, 16, 64);
    lt += __shfl_xor(lt, 32, 64);
#pragma unroll
    for (int d = 0; d < 8; ++d) {
        ot[d] += __shfl_xor(ot[d], 16, 64);
        ot[d] += __shfl_xor(ot[d], 32, 64);
        b.o[d] = ot[d];
    }
    b.m = mt;
    b.l = lt;
    if (lane == 0)  // re-arm; nobody else touches this counter until the next step (stream order / data dependence)
        __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return true;
}
