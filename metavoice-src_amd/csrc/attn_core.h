// Decode-attention building blocks shared by the per-layer launch (attention.hip) and the persistent decode-step kernel
// (decode_mega.hip).  Both paths run EXACTLY this arithmetic in the same order, so they produce identical bits.  The
// library is built with -ffp-contract=off and every fused multiply-add is an explicit fmaf(): left to the compiler, the
// SAME source expression was fused in one kernel and not in the other (1-ulp fp32 differences, a bf16 flip every ~1e6
// outputs).
//
#pragma once
#include "common.h"
#include "kernels.h"

#define ATTN_WAVES 8
#define ATTN_KEYS_PER_WAVE 32
#define ATTN_U (ATTN_KEYS_PER_WAVE / 4)
#define ATTN_KEYS_PER_BLOCK (ATTN_WAVES * ATTN_KEYS_PER_WAVE)  // 256 keys = 128 KB of K+V, all in flight at once

typedef unsigned long long u64;

// 16-byte write-through (sc1) store / agent-scope (sc1) loads: what the split partials travel in.  An 8-byte sc1 store
// costs 2.7x a 16-byte one per byte (MI355X_MICROARCH.md "stores of each flavour"), and the publish sits on the critical
// path of every layer.  hipcc neither counts nor pads inline-asm memory operations (cdna_hip_programming.md 5.7): the
// store statement ends with s_nop 1, the loads carry their own s_waitcnt.
__device__ __forceinline__ void store16_agent(float* p, float a, float b, float c, float d) {
    const u32x4 v = {__float_as_uint(a), __float_as_uint(b), __float_as_uint(c), __float_as_uint(d)};
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}
struct AttnPart6 {
    u32x4 v[6];
};
__device__ __forceinline__ AttnPart6 load6x16_agent(const float* p0, const float* p1, const float* p2, const float* p3,
                                                    const float* p4, const float* p5) {
    AttnPart6 r;
    asm volatile(
        "global_load_dwordx4 %0, %6, off sc1\n\t"
        "global_load_dwordx4 %1, %7, off sc1\n\t"
        "global_load_dwordx4 %2, %8, off sc1\n\t"
        "global_load_dwordx4 %3, %9, off sc1\n\t"
        "global_load_dwordx4 %4, %10, off sc1\n\t"
        "global_load_dwordx4 %5, %11, off sc1\n\t"
        "s_waitcnt vmcnt(0)"
        : "=&v"(r.v[0]), "=&v"(r.v[1]), "=&v"(r.v[2]), "=&v"(r.v[3]), "=&v"(r.v[4]), "=&v"(r.v[5])
        : "v"(p0), "v"(p1), "v"(p2), "v"(p3), "v"(p4), "v"(p5)
        : "memory");
    return r;
}

// Key partition of a (row, head): up to 256 keys ONE block does the whole (row, head) and there is no cross-block combine;
// beyond that always ATTN_NSPLIT = 8 contiguous ranges of cs = ceil(P/8) keys (rounded up to a multiple of 4: a wave instruction
// covers 4 keys, 16 lanes x 16 B per 256-byte key row), so every CU streams 1/8 of a head at any context -- with
// ceil(P/256) ranges the K/V burst of a block was 64-128 KB at every context above 256, now it grows linearly to 128 KB at 2048.
// Inside a range the 8 waves take kw = ceil(cs/8) keys each (multiple of 4, <= 32) and request only those rows.
__device__ __forceinline__ int attn_keys_per_split(int P) {
    const int nsplit = P <= ATTN_KEYS_PER_BLOCK ? 1 : ATTN_NSPLIT;
    return (((P + nsplit - 1) / nsplit) + 3) & ~3;
}
__device__ __forceinline__ int attn_keys_per_wave(int cs) { return (((cs + ATTN_WAVES - 1) / ATTN_WAVES) + 3) & ~3; }

struct AttnAcc {
    float m, l, o[8];
};
// Scores of the lane group's ATTN_U keys against q, softmax numerators against the group max, P.V; then the 4 lane
// groups of the wave (lanes dl, dl+16, dl+32, dl+48 hold the same 8 dims) are folded.  Key index of (u, g) = s0+4u+g,
// valid while < s1.
template <bool PL = (MV_PERMLANE_SWAP != 0)>
__device__ __forceinline__ AttnAcc attn_wave_partial(const u32x4 (&kv)[ATTN_U], const u32x4 (&vv)[ATTN_U], const u32x4& qv, int s0,
                                                     int s1, int g) {
    const float scale = 0.08838834764831845f;  // 1/sqrt(128)
    float sc[ATTN_U];
    AttnAcc a;
    a.m = -INFINITY;
    // slots past the wave's key range (wave-uniform) are skipped: they are all masked (score -inf, weight 0, nothing added), so the bits
    // are the same -- at 1024 positions a wave holds 16 keys in 4 of its 8 slots
    const int nslot = __builtin_amdgcn_readfirstlane((s1 - s0 + 3) >> 2);
#pragma unroll
    for (int u = 0; u < ATTN_U; ++u) {
        sc[u] = -INFINITY;
        if (u < nslot) {
            float s = sum_across16(dot8_bf16(kv[u], qv, 0.f)) * scale;
            s = (s0 + u * 4 + g < s1) ? s : -INFINITY;
            sc[u] = s;
            a.m = fmaxf(a.m, s);
        }
    }
    a.l = 0.f;
#pragma unroll
    for (int d = 0; d < 8; ++d) a.o[d] = 0.f;
#pragma unroll
    for (int u = 0; u < ATTN_U; ++u) {
        if (!(u < nslot)) continue;
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
    const float mo = max_groups4<PL>(a.m);
    const float w = (a.m == -INFINITY) ? 0.f : fast_exp(a.m - mo);
    a.l = add_groups4<PL>(a.l * w);
#pragma unroll
    for (int d = 0; d < 8; ++d) a.o[d] = add_groups4<PL>(a.o[d] * w);
    a.m = mo;
    return a;
}

// `sm` = [8 waves][16 dim groups][12 floats] of per-wave (m, l, o[8]); executed by ONE whole wave: lane (g, dl) merges
// waves g and g+4 for dims 8dl..8dl+7, then the 4 groups are folded with xor-16/32 butterflies.
template <bool PL = (MV_PERMLANE_SWAP != 0)>
__device__ __forceinline__ AttnAcc attn_merge_waves(const float (*sm)[16][12], int g, int dl) {
    AttnAcc b;
    const float m0 = sm[g][dl][0], m1 = sm[g + 4][dl][0];
    const float mw = fmaxf(m0, m1);
    b.m = max_groups4<PL>(mw);
    const float f0 = (m0 == -INFINITY) ? 0.f : fast_exp(m0 - b.m);
    const float f1 = (m1 == -INFINITY) ? 0.f : fast_exp(m1 - b.m);
    b.l = add_groups4<PL>(fmaf(sm[g + 4][dl][1], f1, sm[g][dl][1] * f0));
#pragma unroll
    for (int d = 0; d < 8; ++d) b.o[d] = add_groups4<PL>(fmaf(sm[g + 4][dl][2 + d], f1, sm[g][dl][2 + d] * f0));
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

// Combination of up to 8 split partials by one whole wave: lane group g contributes splits g and g+4 (ms[q] = -inf marks a
// missing split), the 4 groups are folded with xor-16/32 butterflies.  Shared by the ticket form below and the tagged form of
// the persistent kernel: identical arithmetic, identical bits.
template <bool PL = (MV_PERMLANE_SWAP != 0)>
__device__ __forceinline__ void attn_combine_math(AttnAcc& b, const float (&ms)[2], const float (&ls)[2], const float (&t)[2][8]) {
    float mt = fmaxf(ms[0], ms[1]);
    mt = max_groups4<PL>(mt);
    float lt = 0.f, ot[8];
#pragma unroll
    for (int d = 0; d < 8; ++d) ot[d] = 0.f;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const float f = (ms[q] == -INFINITY) ? 0.f : fast_exp(ms[q] - mt);
        lt = fmaf(ls[q], f, lt);
#pragma unroll
        for (int d = 0; d < 8; ++d) ot[d] = fmaf(t[q][d], f, ot[d]);
    }
    lt = add_groups4<PL>(lt);
#pragma unroll
    for (int d = 0; d < 8; ++d) b.o[d] = add_groups4<PL>(ot[d]);
    b.m = mt;
    b.l = lt;
}

// Executed by one whole wave holding the block result `b` (identical in the 4 lane groups).  Publishes the split
// partial (33 x 16 B: [m, l, -, -] then o[128]) write-through and draws a ticket; returns true in the LAST arriver of
// the (row, head), with `b` replaced by the combination of all `nact` <= 8 partials -- lane group g reads splits g and
// g+4, all six 16-byte loads issued before the first use (one round trip).  Agent-scope write-through stores + ticket +
// agent-scope loads: no release / acquire fence (each costs ~1.7 us and drops the CU's L1); correct for any placement of
// the splits over XCDs; the counter re-arms itself.  A partial's address must be written at most once per launch (the
// one-shot loads cannot detect a line their L2 still holds from an earlier write): callers use one slab per layer.
__device__ __forceinline__ bool attn_publish_combine(AttnAcc& b, float* pbase, uint32_t* counter, int split, int nact, int lane) {
    const int g = lane >> 4, dl = lane & 15;
    float* part = pbase + split * ATTN_PART_STRIDE;
    if (lane == 0) store16_agent(part, b.m, b.l, 0.f, 0.f);
    if (g == 0) store16_agent(part + 4 + dl * 8, b.o[0], b.o[1], b.o[2], b.o[3]);
    if (g == 1) store16_agent(part + 4 + dl * 8 + 4, b.o[4], b.o[5], b.o[6], b.o[7]);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's stores have left before the ticket is drawn
    unsigned prev = 0;
    if (lane == 0) prev = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    prev = __builtin_amdgcn_readfirstlane(prev);
    if (prev != (unsigned)(nact - 1)) return false;

    const float* q0 = pbase + min(g, nact - 1) * ATTN_PART_STRIDE;      // clamped: out-of-range slots are weighted 0 below
    const float* q1 = pbase + min(g + 4, nact - 1) * ATTN_PART_STRIDE;
    const AttnPart6 r = load6x16_agent(q0, q0 + 4 + dl * 8, q0 + 8 + dl * 8, q1, q1 + 4 + dl * 8, q1 + 8 + dl * 8);
    float ms[2], ls[2], t[2][8];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        ms[q] = (g + 4 * q < nact) ? __uint_as_float(r.v[3 * q].x) : -INFINITY;
        ls[q] = __uint_as_float(r.v[3 * q].y);
        t[q][0] = __uint_as_float(r.v[3 * q + 1].x); t[q][1] = __uint_as_float(r.v[3 * q + 1].y);
        t[q][2] = __uint_as_float(r.v[3 * q + 1].z); t[q][3] = __uint_as_float(r.v[3 * q + 1].w);
        t[q][4] = __uint_as_float(r.v[3 * q + 2].x); t[q][5] = __uint_as_float(r.v[3 * q + 2].y);
        t[q][6] = __uint_as_float(r.v[3 * q + 2].z); t[q][7] = __uint_as_float(r.v[3 * q + 2].w);
    }
    attn_combine_math(b, ms, ls, t);
    if (lane == 0)  // re-arm; nobody else touches this counter until the next layer / step (data dependence / stream order)
        __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return true;
}

// ---- the same exchange through TAGGED granules (persistent kernel): no ticket, no one-shot loads ------------------------
// A partial is 130 floats sent as 130 granules {float, 32-bit tag}, two per 16-byte store: lane (g, dl) stores o[2g], o[2g+1]
// of its 8 dims (the 4 lane groups hold identical values), lane 0 adds {m, l}.  The workgroup of split 0 polls the partials of
// all `nact` splits (its own included) until every tag matches and combines them with attn_combine_math: one hand-off
// instead of store + vmcnt(0) + ticket round trip + load round trip.  Slots are reused across layers and steps: the tag makes
// a stale line harmless (the poll just retries).
__device__ __forceinline__ void attn_publish_tagged(const AttnAcc& b, u64* slot, uint32_t tag, int lane) {
    const int g = lane >> 4, dl = lane & 15;
    // The eight values are pinned in registers before lane group g picks its pair.  Written as a plain chain of selects over b.o[...]
    // hipcc turned "select of loads" into ONE load through a selected ADDRESS: the accumulator went to scratch memory (44 bytes per
    // lane in every instance of the persistent kernel), and the publish of every split partial waited for two scratch stores, two
    // dynamically addressed scratch loads and -- `s_waitcnt vmcnt(0)` -- every weight row the wave had in flight.
    float o0 = b.o[0], o1 = b.o[1], o2 = b.o[2], o3 = b.o[3], o4 = b.o[4], o5 = b.o[5], o6 = b.o[6], o7 = b.o[7];
    asm volatile("" : "+v"(o0), "+v"(o1), "+v"(o2), "+v"(o3), "+v"(o4), "+v"(o5), "+v"(o6), "+v"(o7));
    const float lo = g == 0 ? o0 : g == 1 ? o2 : g == 2 ? o4 : o6;
    const float hi = g == 0 ? o1 : g == 1 ? o3 : g == 2 ? o5 : o7;
    const u32x4 v = {__float_as_uint(lo), tag, __float_as_uint(hi), tag};
    asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(slot + dl * 8 + 2 * g), "v"(v) : "memory");
    if (lane == 0) {
        const u32x4 w = {__float_as_uint(b.m), tag, __float_as_uint(b.l), tag};
        asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(slot + 128), "v"(w) : "memory");
    }
}
// returns false on a timeout (caller raises the abort flag)
template <bool PL = (MV_PERMLANE_SWAP != 0)>
__device__ __forceinline__ bool attn_combine_tagged(AttnAcc& b, const u64* slots, uint32_t tag, int nact, int lane, int spin_limit) {
    const int g = lane >> 4, dl = lane & 15;
    for (int tries = 0;; ++tries) {
        u32x4 v[2][5];
        bool ok = true;
        // the ten loads and their wait in ONE statement (see load6x16_agent): an output register must not be copied or spilled by
        // the compiler between an inline-asm load and the s_waitcnt that covers it
        const u64* p0 = slots + (size_t)min(g, nact - 1) * ATTN_TPART_STRIDE;       // clamped: weighted 0 below
        const u64* p1 = slots + (size_t)min(g + 4, nact - 1) * ATTN_TPART_STRIDE;
        const u64 *p0d = p0 + dl * 8, *p1d = p1 + dl * 8;
        asm volatile(
            "global_load_dwordx4 %0, %10, off offset:1024 sc1\n\t"
            "global_load_dwordx4 %1, %11, off sc1\n\t"
            "global_load_dwordx4 %2, %11, off offset:16 sc1\n\t"
            "global_load_dwordx4 %3, %11, off offset:32 sc1\n\t"
            "global_load_dwordx4 %4, %11, off offset:48 sc1\n\t"
            "global_load_dwordx4 %5, %12, off offset:1024 sc1\n\t"
            "global_load_dwordx4 %6, %13, off sc1\n\t"
            "global_load_dwordx4 %7, %13, off offset:16 sc1\n\t"
            "global_load_dwordx4 %8, %13, off offset:32 sc1\n\t"
            "global_load_dwordx4 %9, %13, off offset:48 sc1\n\t"
            "s_waitcnt vmcnt(0)"
            : "=&v"(v[0][0]), "=&v"(v[0][1]), "=&v"(v[0][2]), "=&v"(v[0][3]), "=&v"(v[0][4]), "=&v"(v[1][0]), "=&v"(v[1][1]), "=&v"(v[1][2]),
              "=&v"(v[1][3]), "=&v"(v[1][4])
            : "v"(p0), "v"(p0d), "v"(p1), "v"(p1d)
            : "memory");
        __builtin_amdgcn_s_waitcnt(0x0F70);   // compiler-visible vmcnt(0) behind the asm's own: see mega_core.h, mega_loads_have_landed
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int k = 0; k < 5; ++k) ok = ok && v[q][k].y == tag && v[q][k].w == tag;
        if (__all(ok)) {
            float ms[2], ls[2], t[2][8];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                ms[q] = (g + 4 * q < nact) ? __uint_as_float(v[q][0].x) : -INFINITY;
                ls[q] = __uint_as_float(v[q][0].z);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    t[q][2 * k] = __uint_as_float(v[q][1 + k].x);
                    t[q][2 * k + 1] = __uint_as_float(v[q][1 + k].z);
                }
            }
            attn_combine_math<PL>(b, ms, ls, t);
            return true;
        }
        if (tries >= spin_limit) return false;
        __builtin_amdgcn_s_sleep(1);
    }
}
