// Arithmetic of the int4 (group-128) weight-only GEMV, shared by gemv_i4.hip (per-layer launches) and decode_mega.hip
// (persistent kernel): both run EXACTLY this, in this order, so they produce identical bits.
//
// Round 5.  Until round 4 every weight was dequantised to bf16 the way tinygemm does it -- nibble -> v_cvt_f32_ubyte,
// subtract 8, fma with (scale, zero), round to bf16, pack: ~31 VALU instructions per 8 weights in front of the 8 dot2 of two activation
// rows -- and the int4 decode step was SLOWER than the bf16 one (673 vs 661 us: the dequantisation sat between every hand-off's poll and
// publish, profiles/r04_final).  Now:
//   * codes are stored PAIR-MAJOR inside each 32-bit word (`i4_pair_nibble`): columns 2m | 2m+1 of the word's 8 sit at bits [4m, 4m+3] |
//     [16+4m, 16+4m+3], so `(w >> 4m) & 0x000F000F | 0x43004300` IS the bf16 pair {128 + q_2m, 128 + q_2m+1} (0x4300 = 128.0 and a bf16 in
//     [128, 256) has an ulp of 1: OR-ing a nibble into the mantissa is exact) -- 2 instructions per pair, no convert;
//   * a lane's 32 columns of a chunk lie in ONE 128-column group, so with t = sum (128 + q_c) x_c (4 dot2 per word, fp32) and
//     Sx = sum x_c (16 dot2 against 1.0 per activation row and chunk, shared by all weight rows of the phase)
//         sum ((q_c - 8) s + z) x_c  =  s t + (z - 136 s) Sx                      -- two fma per (weight row, chunk, activation row).
// ~15 VALU instructions per 8 weights at two activation rows instead of ~39.
// What changes numerically: the reference's `aten::_weight_int4pack_mm` rounds every dequantised weight to bf16 before the bf16 MMA
// (fam/llm/fast_quantize.py:204-216); here the dequantised weight is never materialised, so that rounding (relative 2^-9 per weight,
// i.e. ~0.1 % of an output's standard deviation -- below the bf16 rounding of the output itself) does not happen.  The quantiser (codes,
// scales, zeros) stays bit-exact with the reference; tests/test_int4_gpu.py prints the distance to both evaluations, the 24-layer goldens
// hold the engine to 1.5 x the reference's own bf16 noise floor as before.  fp32 headroom: |128 + q| <= 143 against |q - 8| <= 8 costs
// ~5 of fp32's 24 mantissa bits in `s t - 136 s Sx` -- 19 bits left, against the 8 of a bf16 weight.
#pragma once
#include "common.h"

#ifndef MV_BF16PAIR_T
#define MV_BF16PAIR_T
typedef __bf16 bf16pair_t __attribute__((ext_vector_type(2)));
#endif

// nibble position (0..7) of column c (0..7) inside its 32-bit word, pair-major: even columns in the low half-word, odd ones 16 bits up
__host__ __device__ constexpr int i4_pair_nibble(int c) { return (c & 1) * 4 + (c >> 1); }
// a word of the ABI layout (nibble p = column p: column 2j in the low, 2j+1 in the high nibble of byte j) -> pair-major
__host__ __device__ inline uint32_t i4_word_to_pairs(uint32_t w) {
    uint32_t o = 0;
    for (int c = 0; c < 8; ++c) o |= ((w >> (4 * c)) & 15u) << (4 * i4_pair_nibble(c));
    return o;
}

__device__ __forceinline__ float dot2u(uint32_t w, uint32_t x, float acc) {
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16pair_t, w), __builtin_bit_cast(bf16pair_t, x), acc, false);
}
// t += sum over the word's 8 columns of (128 + q_c) x_c; xq = the 8 activations of those columns (4 bf16 pairs, column order).
// hipcc emits shift + v_and_b32 + v_or_b32 per pair (a literal each; gfx9's VOP3 v_and_or_b32 takes none).  The one-instruction form through
// inline asm (mask on the constant bus, bias in a VGPR) was tried three ways -- plain, tied to the activation operand, volatile -- and every
// one cost registers instead of saving time: the unpacked pairs are four times the size of the codes, and with an opaque instruction in the
// chain the scheduler unpacks whole rows the moment their codes have landed (221-248 -> 256 VGPRs, 2-128 spilled registers in the
// persistent kernel's int4 instances; a scratch reload waits behind every weight row in flight).
__device__ __forceinline__ float i4_word_dot(uint32_t w, const u32x4& xq, float t) {
    constexpr uint32_t MASK = 0x000F000Fu, BIAS = 0x43004300u;   // bf16 128.0 | 128.0
    t = dot2u((w & MASK) | BIAS, xq.x, t);
    t = dot2u(((w >> 4) & MASK) | BIAS, xq.y, t);
    t = dot2u(((w >> 8) & MASK) | BIAS, xq.z, t);
    t = dot2u(((w >> 12) & MASK) | BIAS, xq.w, t);
    return t;
}
// sum of 8 bf16 (4 pairs), fixed order: dot2 against {1.0, 1.0}
__device__ __forceinline__ float i4_sum8(const u32x4& xq, float s) {
    constexpr uint32_t ONES = 0x3F803F80u;
    s = dot2u(ONES, xq.x, s);
    s = dot2u(ONES, xq.y, s);
    s = dot2u(ONES, xq.z, s);
    s = dot2u(ONES, xq.w, s);
    return s;
}

// acc[i][r] = sum over this lane's columns of dequant(W row i) . x row r.  Per 2048-column chunk j a lane holds 32 columns
// (cw[i][j]: 16 B of pair-major codes, sz[i][j]: the (scale, zero) of their 128-column group) of each of the wave's NR rows; x is read
// from LDS, row stride `kstride` elements; columns at or beyond K (ragged last chunk, clamped duplicate fragment) contribute 0.
template <int R, int KCH, int NR>
__device__ __forceinline__ void i4_dots(const u32x4 (&cw)[NR][KCH], const uint32_t (&sz)[NR][KCH], const bf16_t* xs, int kstride, int K,
                                        int lane, float (&acc)[NR][R]) {
#pragma unroll
    for (int i = 0; i < NR; ++i)
#pragma unroll
        for (int r = 0; r < R; ++r) acc[i][r] = 0.f;
#pragma unroll
    for (int j = 0; j < KCH; ++j) {
        const bool live = j * 2048 + lane * 32 < K;
        // block q = the lane's columns 8 q .. 8 q + 7 of this chunk: one 16-byte LDS read per activation row serves the column sum and every
        // weight row's word q (8 live registers of x instead of 32: the engines of 2 / 4 utterances run at the 256-VGPR limit)
        float sx[R], t[NR][R];
#pragma unroll
        for (int r = 0; r < R; ++r) sx[r] = 0.f;
#pragma unroll
        for (int i = 0; i < NR; ++i)
#pragma unroll
            for (int r = 0; r < R; ++r) t[i][r] = 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            u32x4 xq[R];
#pragma unroll
            for (int r = 0; r < R; ++r) {
                xq[r] = live ? *(const u32x4*)(xs + r * kstride + j * 2048 + lane * 32 + q * 8) : u32x4{0, 0, 0, 0};
                sx[r] = i4_sum8(xq[r], sx[r]);
            }
#pragma unroll
            for (int i = 0; i < NR; ++i) {
                const uint32_t word = q == 0 ? cw[i][j].x : q == 1 ? cw[i][j].y : q == 2 ? cw[i][j].z : cw[i][j].w;
#pragma unroll
                for (int r = 0; r < R; ++r) t[i][r] = i4_word_dot(word, xq[r], t[i][r]);
            }
        }
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            const float s = bf_lo(sz[i][j]), z = bf_hi(sz[i][j]);
            const float zz = fmaf(-136.f, s, z);   // z - (128 + 8) s
#pragma unroll
            for (int r = 0; r < R; ++r) acc[i][r] = fmaf(s, t[i][r], fmaf(zz, sx[r], acc[i][r]));
        }
    }
}
