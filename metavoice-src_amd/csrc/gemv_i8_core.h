// Arithmetic of the int8 (per output channel) weight-only GEMV, shared by gemv_i8.hip (per-layer launches) and decode_mega.hip
// (persistent kernel): both run EXACTLY this, in this order, so they produce identical bits.
#pragma once
#include "common.h"

#ifndef MV_BF16PAIR_T
#define MV_BF16PAIR_T
typedef __bf16 bf16pair_t __attribute__((ext_vector_type(2)));
#endif


__device__ __forceinline__ float dot2p8(bf16pair_t w, uint32_t x, float acc) {
    return __builtin_amdgcn_fdot2_f32_bf16(w, __builtin_bit_cast(bf16pair_t, x), acc, false);
}
// 4 int8 of one word -> 2 bf16 pairs (columns 0|1, 2|3); int8 values are exact in bf16
__device__ __forceinline__ void cvt4(uint32_t w, bf16pair_t (&out)[2]) {
    const int v = (int)w;
    out[0] = bf16pair_t{(__bf16)(float)((v << 24) >> 24), (__bf16)(float)((v << 16) >> 24)};
    out[1] = bf16pair_t{(__bf16)(float)((v << 8) >> 24), (__bf16)(float)(v >> 24)};
}

// acc[i][r] = sum over this lane's columns of (row i codes as bf16) . x row r.  Per 1024-column chunk j a lane holds 16 columns
// (cw[i][j]: 16 B of int8 codes) of each of the wave's NR rows; x is read from LDS, row stride `kstride` elements; columns at or
// beyond K (ragged last chunk, clamped duplicate fragment) contribute 0.
template <int R, int KCH, int NR>
__device__ __forceinline__ void i8_dots(const u32x4 (&cw)[NR][KCH], const bf16_t* xs, int kstride, int K, int lane, float (&acc)[NR][R]) {
#pragma unroll
    for (int i = 0; i < NR; ++i)
#pragma unroll
        for (int r = 0; r < R; ++r) acc[i][r] = 0.f;
#pragma unroll
    for (int j = 0; j < KCH; ++j) {
        const bool live = j * 1024 + lane * 16 < K;
        u32x4 xv[R][2];
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int q = 0; q < 2; ++q)
                xv[r][q] = live ? *(const u32x4*)(xs + r * kstride + j * 1024 + lane * 16 + q * 8) : u32x4{0, 0, 0, 0};
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            const uint32_t words[4] = {cw[i][j].x, cw[i][j].y, cw[i][j].z, cw[i][j].w};
#pragma unroll
            for (int q = 0; q < 4; ++q) {  // word q = columns 4q .. 4q+3 of the lane's 16
                bf16pair_t wp[2];
                cvt4(words[q], wp);
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const u32x4& xq = xv[r][q >> 1];
                    const uint32_t x0 = (q & 1) ? xq.z : xq.x, x1 = (q & 1) ? xq.w : xq.y;
                    acc[i][r] = dot2p8(wp[1], x1, dot2p8(wp[0], x0, acc[i][r]));
                }
            }
        }
    }
}
