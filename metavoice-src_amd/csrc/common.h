// Shared device helpers for the gfx950 (CDNA4, wave64) kernels of the MetaVoice engine.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef uint16_t bf16_t;  // raw bfloat16 bits
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));

#define MV_WAVE 64

__device__ __forceinline__ float bf2f(bf16_t b) { return __uint_as_float(((uint32_t)b) << 16); }
__device__ __forceinline__ float bf_lo(uint32_t p) { return __uint_as_float(p << 16); }
__device__ __forceinline__ float bf_hi(uint32_t p) { return __uint_as_float(p & 0xFFFF0000u); }

// fp32 -> bf16, round to nearest even (the rounding torch's .to(bfloat16) applies): gfx950 converts in hardware
// (v_cvt_pk_bf16_f32, two values per instruction) -- the integer add-and-shift form cost ~8 VALU ops per value, which made the
// RMSNorm apply and the epilogues of the decode kernels VALU-bound (1.6 us per norm in the persistent kernel).
typedef __bf16 mv_bf16pair_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack_bf2(float lo, float hi) {
    return __builtin_bit_cast(uint32_t, mv_bf16pair_t{(__bf16)lo, (__bf16)hi});
}
__device__ __forceinline__ bf16_t f2bf(float f) { return (bf16_t)pack_bf2(f, 0.f); }
__device__ __forceinline__ float round_bf(float f) { return __uint_as_float(pack_bf2(0.f, f) & 0xFFFF0000u); }

// acc += w.lo*x.lo + w.hi*x.hi  (v_dot2c_f32_bf16; products of two bf16 are exact in fp32)
__device__ __forceinline__ float dot2_bf16(uint32_t w, uint32_t x, float acc) {
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, w), __builtin_bit_cast(bf16x2_t, x), acc, false);
}
__device__ __forceinline__ float dot8_bf16(const u32x4& w, const u32x4& x, float acc) {
    acc = dot2_bf16(w.x, x.x, acc);
    acc = dot2_bf16(w.y, x.y, acc);
    acc = dot2_bf16(w.z, x.z, acc);
    acc = dot2_bf16(w.w, x.w, acc);
    return acc;
}

// Streaming (read-once) 16-byte load: global_load_dwordx4 ... nt  (MI355X_MICROARCH "nt-weights").
// The pointer is cast to the GLOBAL address space: through a generic pointer (weights reach the kernels inside argument structs)
// hipcc emits flat_load_dwordx4, and a FLAT load also counts in lgkmcnt -- so every LDS read issued while weight rows were in
// flight (`s_waitcnt lgkmcnt(0)`) waited for the whole weight stream.  global_load counts in vmcnt only.
typedef const __attribute__((address_space(1))) u32x4* mv_gptr16_t;
__device__ __forceinline__ u32x4 load_stream16(const void* p) { return __builtin_nontemporal_load((mv_gptr16_t)p); }
__device__ __forceinline__ u32x4 load16(const void* p) { return *(mv_gptr16_t)p; }
// scalars through global-address-space pointers for the same reason (scale / zero words of the quantised formats)
__device__ __forceinline__ uint32_t load_global_u32(const uint32_t* p) { return *(const __attribute__((address_space(1))) uint32_t*)p; }
__device__ __forceinline__ uint16_t load_global_u16(const uint16_t* p) { return *(const __attribute__((address_space(1))) uint16_t*)p; }

// ---- cross-lane sums over a 64-lane wave; every lane ends up with the total -------------------------------
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, false));
}
__device__ __forceinline__ float sum_across16(float v) {
    v += dpp_f<0xB1>(v);   // quad_perm [1,0,3,2]  (lane ^ 1)
    v += dpp_f<0x4E>(v);   // quad_perm [2,3,0,1]  (lane ^ 2)
    v += dpp_f<0x141>(v);  // row_half_mirror      (8-lane halves)
    v += dpp_f<0x140>(v);  // row_mirror           (16-lane rows)
    return v;
}
// Whole-wave sum, result wave-uniform.  After the 4 intra-row steps every lane holds its 16-lane row total; two DPP
// row broadcasts (row_bcast15 into rows 1,3; row_bcast31 into rows 2,3) leave the wave total in lane 63, read back
// through an SGPR.  6 DPP adds + 1 v_readlane (~60 cycles) instead of two ds_bpermute round trips through the LDS
// crossbar (~200 cycles): this sits on the serial tail of every decode kernel.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_bcast_f(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xF, false));
}
__device__ __forceinline__ float wave_sum(float v) {
    v = sum_across16(v);
    v += dpp_bcast_f<0x142, 0xA>(v);  // row_bcast15
    v += dpp_bcast_f<0x143, 0xC>(v);  // row_bcast31
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
// N independent wave sums, level by level.  Same tree per value as wave_sum (so the same bits) -- but written one value after the other
// hipcc ran every chain through the same two registers at the 240-VGPR kernels' register pressure: 17 dependent instructions with an
// `s_nop 1` behind every DPP add (a VALU write needs two wait states before a DPP read of it), ~100 cycles per sum, 22 sums per layer of the
// persistent decode kernel = 0.8 us per layer (profiles/r04t: 678 -> 658 us/step with the sums removed).  Level by level the N chains are
// independent neighbours, in place in the registers the accumulators already occupy: no wait states, ~4 cycles per instruction.
// The two row-broadcast levels as ONE instruction per value: `v_add_f32_dpp v, v, v row_bcast:15 row_mask:0xa` adds lane 15 of the row
// below into rows 1 and 3 and leaves rows 0 and 2 alone (a masked-off row keeps the destination = the tied input).  From C the compiler
// builds v_mov_b32 0 / v_mov_b32_dpp / v_add_f32 (its DPP combiner folds a masked mov only into integer ops), three instructions per
// value and level.  Lane 63 -- the only lane read -- gets the same bits (rows 1,3 then 2,3 are the enabled ones on its path).
// Inline assembly is invisible to the compiler's hazard recogniser, so each statement opens with its own `s_nop 1`: a VALU write
// needs two wait states before a DPP read of the register; within a statement the values are independent.  The last level's statements
// CLOSE with an `s_nop 0`: wave_sum_n reads the results with v_readlane right behind them, and a VALU write needs one wait state before
// a v_readlane of the register -- the recogniser does not know these statements write VGPRs.
#define MV_BC(k, C) "v_add_f32_dpp %" #k ", %" #k ", %" #k " " C "\n"
#define MV_BCAST15 "row_bcast:15 row_mask:0xa bank_mask:0xf"
#define MV_BCAST31 "row_bcast:31 row_mask:0xc bank_mask:0xf"
template <int N>
__device__ __forceinline__ void wave_bcast_levels(float (&v)[N]) {
    static_assert(N % 2 == 0, "pairs");
    if constexpr (N % 6 == 0) {
#pragma unroll
        for (int j = 0; j < N; j += 6)
            asm("s_nop 1\n" MV_BC(0, MV_BCAST15) MV_BC(1, MV_BCAST15) MV_BC(2, MV_BCAST15) MV_BC(3, MV_BCAST15) MV_BC(4, MV_BCAST15) MV_BC(5, MV_BCAST15)
                : "+v"(v[j]), "+v"(v[j + 1]), "+v"(v[j + 2]), "+v"(v[j + 3]), "+v"(v[j + 4]), "+v"(v[j + 5]));
#pragma unroll
        for (int j = 0; j < N; j += 6)
            asm("s_nop 1\n" MV_BC(0, MV_BCAST31) MV_BC(1, MV_BCAST31) MV_BC(2, MV_BCAST31) MV_BC(3, MV_BCAST31) MV_BC(4, MV_BCAST31) MV_BC(5, MV_BCAST31) "s_nop 0\n"
                : "+v"(v[j]), "+v"(v[j + 1]), "+v"(v[j + 2]), "+v"(v[j + 3]), "+v"(v[j + 4]), "+v"(v[j + 5]));
    } else {
#pragma unroll
        for (int j = 0; j < N; j += 2) asm("s_nop 1\n" MV_BC(0, MV_BCAST15) MV_BC(1, MV_BCAST15) : "+v"(v[j]), "+v"(v[j + 1]));
#pragma unroll
        for (int j = 0; j < N; j += 2) asm("s_nop 1\n" MV_BC(0, MV_BCAST31) MV_BC(1, MV_BCAST31) "s_nop 0\n" : "+v"(v[j]), "+v"(v[j + 1]));
    }
}
#ifndef MV_WAVE_SUM_ASM_BCAST
#define MV_WAVE_SUM_ASM_BCAST 1
#endif
template <int N>
__device__ __forceinline__ void wave_sum_n(float (&v)[N]) {
#pragma unroll
    for (int j = 0; j < N; ++j) v[j] += dpp_f<0xB1>(v[j]);   // quad_perm [1,0,3,2]
#pragma unroll
    for (int j = 0; j < N; ++j) v[j] += dpp_f<0x4E>(v[j]);   // quad_perm [2,3,0,1]
#pragma unroll
    for (int j = 0; j < N; ++j) v[j] += dpp_f<0x141>(v[j]);  // row_half_mirror
#pragma unroll
    for (int j = 0; j < N; ++j) v[j] += dpp_f<0x140>(v[j]);  // row_mirror
    if constexpr (MV_WAVE_SUM_ASM_BCAST && N % 2 == 0) wave_bcast_levels<N>(v);
    else {
#pragma unroll
        for (int j = 0; j < N; ++j) v[j] += dpp_bcast_f<0x142, 0xA>(v[j]);  // row_bcast15
#pragma unroll
        for (int j = 0; j < N; ++j) v[j] += dpp_bcast_f<0x143, 0xC>(v[j]);  // row_bcast31
    }
#pragma unroll
    for (int j = 0; j < N; ++j) v[j] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v[j]), 63));
}
// x (+ or max) the value of the lane 16 / 32 away -- `x op __shfl_xor(x, 16 | 32)` -- through gfx950's row / half swaps instead of
// ds_bpermute_b32 (address VGPR, a trip through the LDS crossbar and an lgkmcnt wait per level: ~120 cycles, three dependent pairs of
// them on the attention chain of every layer).  v_permlane16_swap exchanges the odd 16-lane rows of its first operand with the even rows
// of its second; with both = x the results are {rows 0,0,2,2} and {rows 1,1,3,3}: every lane holds its own value in one and its
// partner's in the other (which one depends on the row).  op is commutative, so the bits are those of the shuffle form.
// MEASURED (profiles/r04w, persistent decode kernel, same box, identical tokens): no faster -- bf16 661.0 -> 661-666 us/step, two
// utterances 904 -> 914, int8 634 -> 641, and the bf16 four-utterance instance spills 42 registers over the two copies each swap needs
// (1300 -> 1443).  The shuffles' LDS round trips overlap with the other waves' work; the swaps add a copy and two wait states per level.
// Kept as a switch, OFF.
#ifndef MV_PERMLANE_SWAP
#define MV_PERMLANE_SWAP 0
#endif
template <bool PL = (MV_PERMLANE_SWAP != 0)>
__device__ __forceinline__ float add_xor16(float x) {
    if constexpr (PL) {
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
    } else return x + __shfl_xor(x, 16, 64);
}
template <bool PL = (MV_PERMLANE_SWAP != 0)>
__device__ __forceinline__ float add_xor32(float x) {
    if constexpr (PL) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
    } else return x + __shfl_xor(x, 32, 64);
}
template <bool PL = (MV_PERMLANE_SWAP != 0)>
__device__ __forceinline__ float max_xor16(float x) {
    if constexpr (PL) {
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
    } else return fmaxf(x, __shfl_xor(x, 16, 64));
}
template <bool PL = (MV_PERMLANE_SWAP != 0)>
__device__ __forceinline__ float max_xor32(float x) {
    if constexpr (PL) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
    } else return fmaxf(x, __shfl_xor(x, 32, 64));
}
// the two levels that fold the wave's four 16-lane groups
template <bool PL = (MV_PERMLANE_SWAP != 0)>
__device__ __forceinline__ float add_groups4(float x) { return add_xor32<PL>(add_xor16<PL>(x)); }
template <bool PL = (MV_PERMLANE_SWAP != 0)>
__device__ __forceinline__ float max_groups4(float x) { return max_xor32<PL>(max_xor16<PL>(x)); }
__device__ __forceinline__ float max_across16(float v) {
    v = fmaxf(v, dpp_f<0xB1>(v));
    v = fmaxf(v, dpp_f<0x4E>(v));
    v = fmaxf(v, dpp_f<0x141>(v));
    v = fmaxf(v, dpp_f<0x140>(v));
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
    v = max_across16(v);
    // lanes outside the row mask receive `old`; -inf is the identity of max
    v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0xFF800000, __float_as_int(v), 0x142, 0xA, 0xF, false)));
    v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0xFF800000, __float_as_int(v), 0x143, 0xC, 0xF, false)));
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

__device__ __forceinline__ float fast_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.4426950408889634f); }
