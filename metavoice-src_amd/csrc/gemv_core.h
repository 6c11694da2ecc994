// Arithmetic shared by the per-launch GEMV kernels (gemv.hip) and the persistent decode-step kernel (decode_mega.hip):
// one source for every expression whose rounding matters, so the two paths produce identical bits.
#pragma once
#include "common.h"

// sum of squares of 8 bf16 (RMSNorm statistics, fam/llm/fast_model.py:256-261), fixed order
__device__ __forceinline__ float sumsq8(const u32x4& x) {
    float s = 0.f;
    s += bf_lo(x.x) * bf_lo(x.x) + bf_hi(x.x) * bf_hi(x.x);
    s += bf_lo(x.y) * bf_lo(x.y) + bf_hi(x.y) * bf_hi(x.y);
    s += bf_lo(x.z) * bf_lo(x.z) + bf_hi(x.z) * bf_hi(x.z);
    s += bf_lo(x.w) * bf_lo(x.w) + bf_hi(x.w) * bf_hi(x.w);
    return s;
}
__device__ __forceinline__ float rms_rstd(float total_sumsq, int K, float eps) { return rsqrtf(total_sumsq / (float)K + eps); }
// Sum of the per-wave partial sums `red[w * stride]`, w < nwaves <= 64, identical in every lane: lane w fetches partial w and
// the 64-lane DPP tree adds them in one fixed order (a serial `for w: tot += red[w]` loop is nwaves dependent LDS round
// trips, ~1 us of the prologue at 11 waves).  Waves beyond the K columns contribute exact zeros, so the result does not
// depend on the block size.
__device__ __forceinline__ float block_total(const float* red, int stride, int nwaves, int lane) {
    return wave_sum(lane < nwaves ? red[lane * stride] : 0.f);
}
// bf16(bf16(x * rstd) * w) for 8 elements.  Both elements of a pair take their first rounding in ONE v_cvt_pk_bf16_f32 (the same RNE
// conversion as two separate ones: identical bits); the empty asm keeps hipcc from splitting the packed conversion back into two
// single ones with a junk operand each plus a redundant mask (54 -> 45 VALU instructions per 8 elements; the RMSNorm apply is
// VALU-bound: 6.4 us of a 72 us layer at 8 utterances, profiles/r05u).
__device__ __forceinline__ uint32_t rms_apply2(uint32_t x, float rstd, uint32_t nw) {
    uint32_t p = pack_bf2(bf_lo(x) * rstd, bf_hi(x) * rstd);
    asm("" : "+v"(p));
    return pack_bf2(__uint_as_float(p << 16) * bf_lo(nw), __uint_as_float(p & 0xFFFF0000u) * bf_hi(nw));
}
// the form without the packed first rounding (one conversion per element): the same bits.  decode_mfma.hip keeps it -- with the packed form the
// 8-utterance instance's register allocation spilled two K/V slots on arrival (tests/test_isa_regressions_cpu.py)
__device__ __forceinline__ u32x4 rms_apply8_unpacked(const u32x4& x, float rstd, const u32x4& nw) {
    u32x4 y;
    y.x = pack_bf2(round_bf(bf_lo(x.x) * rstd) * bf_lo(nw.x), round_bf(bf_hi(x.x) * rstd) * bf_hi(nw.x));
    y.y = pack_bf2(round_bf(bf_lo(x.y) * rstd) * bf_lo(nw.y), round_bf(bf_hi(x.y) * rstd) * bf_hi(nw.y));
    y.z = pack_bf2(round_bf(bf_lo(x.z) * rstd) * bf_lo(nw.z), round_bf(bf_hi(x.z) * rstd) * bf_hi(nw.z));
    y.w = pack_bf2(round_bf(bf_lo(x.w) * rstd) * bf_lo(nw.w), round_bf(bf_hi(x.w) * rstd) * bf_hi(nw.w));
    return y;
}
__device__ __forceinline__ u32x4 rms_apply8(const u32x4& x, float rstd, const u32x4& nw) {
    u32x4 y;
    y.x = rms_apply2(x.x, rstd, nw.x);
    y.y = rms_apply2(x.y, rstd, nw.y);
    y.z = rms_apply2(x.z, rstd, nw.z);
    y.w = rms_apply2(x.w, rstd, nw.w);
    return y;
}
// y = bf16(W x); h = bf16(residual + y)                       (fast_model.py:179-180)
__device__ __forceinline__ bf16_t epi_resid(bf16_t residual, float acc) { return f2bf(bf2f(residual) + round_bf(acc)); }
// bf16(bf16(silu(bf16(w1 x))) * bf16(w3 x))                   (fast_model.py:237)
__device__ __forceinline__ bf16_t epi_swiglu(float acc1, float acc3) {
    const float g = round_bf(acc1);
    const float u = round_bf(acc3);
    const float s = round_bf(g / (1.0f + __expf(-g)));
    return f2bf(s * u);
}
