// Arithmetic of the int4 (group-128) weight-only GEMV, shared by gemv_i4.hip (per-layer launches) and decode_mega.hip
// (persistent kernel): both run EXACTLY this, in this order, so they produce identical bits.
#pragma once
#include "common.h"

#ifndef MV_BF16PAIR_T
#define MV_BF16PAIR_T
typedef __bf16 bf16pair_t __attribute__((ext_vector_type(2)));
#endif

// 8 nibbles of one word -> 4 bf16 pairs (columns 0|1, 2|3, 4|5, 6|7): bf16(fma(q - 8, s, z)), one rounding like tinygemm
__device__ __forceinline__ void deq8(uint32_t w, float s, float z, bf16pair_t (&out)[4]) {
    const uint32_t lo = w & 0x0F0F0F0Fu, hi = (w >> 4) & 0x0F0F0F0Fu;
    // (float)((x >> 8k) & 0xFF) selects v_cvt_f32_ubyte<k>
    const float a0 = (float)(lo & 0xFFu), a1 = (float)((lo >> 8) & 0xFFu), a2 = (float)((lo >> 16) & 0xFFu), a3 = (float)(lo >> 24);
    const float c0 = (float)(hi & 0xFFu), c1 = (float)((hi >> 8) & 0xFFu), c2 = (float)((hi >> 16) & 0xFFu), c3 = (float)(hi >> 24);
    out[0] = bf16pair_t{(__bf16)fmaf(a0 - 8.f, s, z), (__bf16)fmaf(c0 - 8.f, s, z)};
    out[1] = bf16pair_t{(__bf16)fmaf(a1 - 8.f, s, z), (__bf16)fmaf(c1 - 8.f, s, z)};
    out[2] = bf16pair_t{(__bf16)fmaf(a2 - 8.f, s, z), (__bf16)fmaf(c2 - 8.f, s, z)};
    out[3] = bf16pair_t{(__bf16)fmaf(a3 - 8.f, s, z), (__bf16)fmaf(c3 - 8.f, s, z)};
}
__device__ __forceinline__ float dot2p(bf16pair_t w, uint32_t x, float acc) {
    return __builtin_amdgcn_fdot2_f32_bf16(w, __builtin_bit_cast(bf16pair_t, x), acc, false);
}

// acc[i][r] = sum over this lane's columns of dequant(W row i) . x row r.  Per 2048-column chunk j a lane holds 32 columns
// (cw[i][j]: 16 B of codes, sz[i][j]: the (scale, zero) of their 128-column group) of each of the wave's NR rows; x is read
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
        u32x4 xv[R][4];
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                xv[r][q] = live ? *(const u32x4*)(xs + r * kstride + j * 2048 + lane * 32 + q * 8) : u32x4{0, 0, 0, 0};
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            const float s = bf_lo(sz[i][j]), z = bf_hi(sz[i][j]);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                bf16pair_t wp[4];
                const uint32_t word = q == 0 ? cw[i][j].x : q == 1 ? cw[i][j].y : q == 2 ? cw[i][j].z : cw[i][j].w;
                deq8(word, s, z, wp);
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    float t = acc[i][r];
                    t = dot2p(wp[0], xv[r][q].x, t);
                    t = dot2p(wp[1], xv[r][q].y, t);
                    t = dot2p(wp[2], xv[r][q].z, t);
                    t = dot2p(wp[3], xv[r][q].w, t);
                    acc[i][r] = t;
                }
            }
        }
    }
}
