// Persistent decode-step kernel for engines of 4 and 8 utterances (8 / 16 CFG rows) on the MATRIX CORES: all 24 layers + the LM head of one token
// in ONE launch, the same hand-off protocol, granule buffers and attention partition as decode_mega.hip -- but the five projections of a
// layer run on v_mfma_f32_16x16x32_bf16 instead of v_dot2 + 64-lane sums.
//
// Why.  decode_mega.hip gives a wave whole weight rows and applies them to every activation row with v_dot2c: 4 dot2 per 8 weights per
// activation row and one 64-lane DPP sum per (weight row, activation row).  At one utterance (2 rows) that is nothing next to the stream;
// at four (8 rows) it is 4 x the VALU issue on two waves per SIMD plus 22 x 4 rolled epilogue sums per layer: profiles/r04h -- of a 52.6 us
// layer, "qkv published" 1.9 us, "h published" 1.1, "act published" 3.6 behind their dots, norms 2.4 + 1.6.  A 16 x 16 x 32 MFMA multiplies 16
// weight rows by up to 16 activation rows in 16 cycles and its accumulator IS the output tile: no lane sums at all.
//
// Layout (the engine owns it: mv_finalize builds a second, TILED copy of the projection weights, tile_weights_kernel).  Workgroup b owns the
// same rows of every matrix as in decode_mega.hip (24 wqkv, 8 wo, 22 w1, 22 w3, 8 w2, 11 head rows); its slice of a matrix is stored
// k-step major, [K / 32][rows][32 columns], so the A fragment of row tile t and k-step s -- lane (rho = lane & 15, g = lane >> 4) holds
// columns 32 s + 8 g .. + 7 of row 16 t + rho -- is ONE 16-byte load per lane out of a contiguous rows x 64 B piece (fully coalesced; row
// slots beyond the slice re-read its last row: no extra HBM bytes, their outputs are never read).  The K dimension is split over the 8 waves
// of the workgroup (8 k-steps each at K = 2048, 22 at K = 5632), each wave accumulates its part of every row tile, the 8 partial tiles meet in
// LDS (one barrier) and thread (row, utterance) adds them in wave order, applies the epilogue of gemv_core.h and publishes the granule.
// The B fragment -- activation row lane & 15, the same 8 columns -- is a 16-byte LDS read from the operand rows xs[8][K + 8] (row stride =
// 4 banks mod 32: conflict-free for the instruction's 16-lane groups).
//
// Numerics.  Same rounding points as every other path (gemv_core.h epilogues, fp32 RMSNorm, attn_core.h attention -- the attention phase
// is decode_mega.hip's, function for function); what changes is the ORDER of the fp32 additions inside a projection (MFMA k-order per
// wave, then 8 wave partials) -- so this kernel is NOT bit-identical to the per-layer launches or to a single-utterance run; the tests
// hold it to the bf16 noise floor instead (tests/test_engine_gpu.py: logits within tolerance, margin-gated token agreement).
// `mv_tune("mfma", 0)` puts engines of 4 utterances back on decode_mega.hip's bit-identical instance.  Engines of 8 utterances (16 rows = one full
// B tile; the w2 phase in two passes over the column halves of the activation vector) exist only here.
#include <atomic>

#include "mega_core.h"

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));

// Compile switches that were measured and lost are kept as a patch, not in this file (tools/experiments/decode_mfma_r06_switches.patch; numbers on
// the round-6 transport, 4 | 8 utterances at 1024 steps, profiles/r06h): MF_KV_DB -- a second K/V register set, ranges in pairs -- 1385-1405 | 1801-1808
// us/step against 952-981 | 1320-1323 (370-430 spilled registers); MF_KV_LDS -- the next key range into LDS through global_load_lds while the current
// one is scored -- 983-992 | 1330-1334; MF_W1_EARLY -- w1 with wo under the q/k/v hand-off -- 1233 | 1586; MF_W3_EARLY -- w3 with w1 behind the
// attention -- 1098 | 1527 (profiles/r06e); the unpacked RMSNorm apply (one conversion per element) 986 | 1336-1357.
// Re-measured at -O2 (3 | 0 spilled registers, see the end of this file; profiles/r06sw, 1896 steps): MF_KV_LDS 967-968 | 1376-1380 against
// 954-955 | 1367-1368 -- it loses without scratch reloads too (the LDS -> register copy and its waits are on the range loop's path); MF_W1_EARLY
// 1019 | 1414-1416; MF_KV_DB still spills 96 | 104 registers at -O2 and was not timed.
// MF_NSPLIT: how many key ranges a (utterance, row, head) is cut into (see the geometry below): -1 = per instance -- 4 utterances: 1 | 4 | 8
// (two workgroups share a (utterance, row, head): ceil(P / 256) ranges split 3 + 2 between them and measured no gain, r05j: 1164-1167 vs
// 1169-1171 us/step), 8 utterances: ceil(P / 256) (one workgroup walks them all: 1859-1868 vs 1898 us/step)
#ifndef MF_NSPLIT
#define MF_NSPLIT -1
#endif
// MF_FOLD_NORM (round 6, default 1): the RMSNorm WEIGHT is folded into the tiled copy of the matrix that follows the norm (wqkv, w1 | w3, the LM
// head: W'[n][k] = bf16(W[n][k] g[k]), tile_weights_kernel) and the row's 1 / rms is applied to the reduced accumulator -- W (g . x rstd) =
// rstd (W diag(g)) x -- so the operand rows stay as they arrived and a norm is ONE pass of sums of squares instead of that plus a read-modify-
// write pass over 8 / 16 rows x 2048 with two roundings per element (3.4 + 2.6 us of a 68 us layer at 8 utterances, every workgroup repeating
// it: profiles/r06_final/stamps_8utt.txt).  Rounding: the reference rounds bf16(bf16(x rstd) g) per element and multiplies by bf16 weights;
// here the weights take one more bf16 rounding (relative 2^-9, once, at load) and the two per-element roundings of the normalised row go away
// -- a different point of the same bf16 noise; this kernel is held to the reference's 24-layer goldens at 1.5 x the reference's own
// bf16-vs-fp32 floor either way (tests/test_fullsize_mfma_golden_gpu.py).  0: the in-place norm on untouched weights (A/B builds).
#ifndef MF_FOLD_NORM
#define MF_FOLD_NORM 1
#endif
// which instance this translation unit holds (4: decode_mfma.hip, with the host-side helpers; 8: decode_mfma8.hip -- see the end of the file)
#ifndef MF_INSTANCE
#define MF_INSTANCE 4
#endif
// operand row stride in LDS (elements).  4 utterances: 8 rows of a whole 5632-column activation vector (11 280 B = 2820 words = 4 banks
// mod 32); 8 utterances: 16 rows of HALF of it (the w2 phase runs in two passes over the column halves: 16 whole rows would be 180 KB)
__host__ __device__ constexpr int mf_kp(int nu) { return (nu <= 4 ? MG_INTER : MG_INTER / 2) + 8; }
#define MF_RED_TILES 4            // partial tiles per wave in the exchange buffer (w1 | w3: two row tiles each)

// rows of each matrix one workgroup owns (decode_mega.hip's partition)
#define MF_R_QKV 24
#define MF_R_WO 8
#define MF_R_W13 22
#define MF_R_W2 8
#define MF_R_HEAD 11

#if MF_INSTANCE == 4
// ---- tiled weight copy ---------------------------------------------------------------------------------------------------------------
// dst[((b * (K / 32) + s) * rows + r) * 32 + c] = src[min(b * rows + r, N - 1)][32 s + c]   (b < 256 workgroups, r < rows per workgroup)
// col_scale (MF_FOLD_NORM): the norm weight g[k] of the RMSNorm in front of this matrix, multiplied into column k (fp32 product, one bf16 rounding)
__global__ void __launch_bounds__(256) tile_weights_kernel(const bf16_t* __restrict__ src, bf16_t* __restrict__ dst, int N, int K, int rows,
                                                           const bf16_t* __restrict__ col_scale) {
    const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 8;   // 8 consecutive columns = one 16-byte piece of a row's k-step
    const size_t total = (size_t)256 * rows * K;
    if (i >= total) return;
    const int c = (int)(i % 32);
    const size_t t = i / 32;
    const int r = (int)(t % rows);
    const size_t t2 = t / rows;
    const int s = (int)(t2 % (K / 32)), b = (int)(t2 / (K / 32));
    const int n = min(b * rows + r, N - 1);
    u32x4 w = *(const u32x4*)(src + (size_t)n * K + 32 * s + c);
    if (MF_FOLD_NORM && col_scale) {
        const u32x4 g = *(const u32x4*)(col_scale + 32 * s + c);
        w.x = pack_bf2(bf_lo(w.x) * bf_lo(g.x), bf_hi(w.x) * bf_hi(g.x)); w.y = pack_bf2(bf_lo(w.y) * bf_lo(g.y), bf_hi(w.y) * bf_hi(g.y));
        w.z = pack_bf2(bf_lo(w.z) * bf_lo(g.z), bf_hi(w.z) * bf_hi(g.z)); w.w = pack_bf2(bf_lo(w.w) * bf_lo(g.w), bf_hi(w.w) * bf_hi(g.w));
    }
    *(u32x4*)(dst + i) = w;
}
hipError_t launch_tile_weights(hipStream_t s, const bf16_t* src, bf16_t* dst, int N, int K, int rows_per_wg, const bf16_t* col_scale) {
    if (K % 32 != 0 || rows_per_wg < 1) return hipErrorInvalidValue;
    const size_t pieces = (size_t)256 * rows_per_wg * K / 8;
    hipLaunchKernelGGL(tile_weights_kernel, dim3((unsigned)((pieces + 255) / 256)), dim3(256), 0, s, src, dst, N, K, rows_per_wg, col_scale);
    return hipGetLastError();
}
size_t mfma_tiled_elems(int K, int rows_per_wg) { return (size_t)256 * rows_per_wg * K; }
#endif

namespace {

// A fragment: row tile t, k-step s of a workgroup's slice [K / 32][NROWS][32].  The address is (wave-uniform base of the k-step: SGPRs)
// + (the lane's 32-bit byte offset inside a k-step piece: ONE VGPR per row tile): global_load ... v_off, s[base] -- with 64-bit per-lane
// pointers hipcc formed all addresses of a request burst first (2 VGPRs per fragment in front of 4 per fragment of data) and spilled.
template <int NROWS>
__device__ __forceinline__ uint32_t mf_lane_off(int t, int lane) {   // byte offset of the lane's 16 bytes inside a k-step piece of NROWS x 64 B
    return (uint32_t)(min(16 * t + (lane & 15), NROWS - 1) * 64 + (lane >> 4) * 16);
}
// A PACKED half tile (a slice's last 8 or fewer rows, FIRST .. NROWS - 1): row slots 0..7 hold those rows at k-step s, row slots 8..15 hold
// the SAME rows at k-step s + SKIP -- one fragment, two MFMAs (against the activations of k-step s: accumulator rows 0..7 are the product,
// rows 8..15 garbage; against those of k-step s + SKIP: the other way round).  With the row slots beyond the slice merely duplicated an
// 8-row slice (wo, w2: one row per wave in decode_mega.hip) cost twice its bytes in registers -- 88 for w2 alone.
template <int NROWS, int FIRST, int SKIP>
__device__ __forceinline__ uint32_t mf_lane_off_half(int lane) {
    const int rho = lane & 15;
    return (uint32_t)(min(FIRST + (rho & 7), NROWS - 1) * 64 + (lane >> 4) * 16 + (rho >> 3) * SKIP * NROWS * 64);
}
// the two accumulators of a packed half tile as ONE tile: rows 0..7 from the MFMAs against k-step s (lane groups 0, 1 hold accumulator rows
// 0..7), rows 8..15 from those against k-step s + SKIP; the reducer adds row i and row i + 8
__device__ __forceinline__ f32x4_t mf_half_tile(const f32x4_t& lo, const f32x4_t& hi, int lane) { return (lane & 32) ? hi : lo; }
template <int NROWS>
__device__ __forceinline__ u32x4 mf_frag(const bf16_t* slice, int s, uint32_t lane_off) {   // slice, s: wave-uniform
    const char* piece = (const char*)slice + (size_t)s * NROWS * 64;
    return load_stream16(piece + lane_off);
}
// B fragment: activation row (lane & 15) mod R, columns 32 s + 8 (lane >> 4) .. + 7
template <int R>
__device__ __forceinline__ bf16x8_t mf_bfrag(const bf16_t* xs, int s, int lane) {
    return *(const bf16x8_t*)(xs + (size_t)(lane & (R - 1)) * mf_kp(R / 2) + 32 * s + (lane >> 4) * 8);
}
__device__ __forceinline__ f32x4_t mf_mma(const u32x4& a, const bf16x8_t& b, const f32x4_t& c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), b, c, 0, 0, 0);
}
// the wave's partial of row tile `tile` into the exchange buffer red[wave][tile][lane][4]
__device__ __forceinline__ void mf_put(float* red, int wave, int tile, int lane, const f32x4_t& acc) {
    *(f32x4_t*)(red + (((size_t)wave * MF_RED_TILES + tile) * 64 + lane) * 4) = acc;
}
// D[row slot i of tile][activation row j], summed over the 8 waves in wave order (accumulator layout: column = lane & 15 = activation row,
// row = 4 (lane >> 4) + register)
__device__ __forceinline__ float mf_get(const float* red, int tile, int i, int j) {
    const float* p = red + ((size_t)tile * 64 + (i >> 2) * 16 + j) * 4 + (i & 3);
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < MG_WAVES; ++w) s += p[(size_t)w * MF_RED_TILES * 256];
    return s;
}

// RMSNorm of the R operand rows in place (fp32 inside, bf16 rounding points of fast_model.py:256-261 through rms_apply8): wave w does rows
// w, w + 8, ...; lane l holds columns 512 q + 8 l .. + 7.  The norm weights come from LDS (`nws`, written by the whole workgroup a phase
// ahead) and the row is read twice (statistics, then apply), one 16-byte piece at a time: held in registers through the norm -- 16 for the
// weights, 16 for the row, ~60 of unrolled apply temporaries -- next to the 128 registers of w1 | w3 fragments in flight, the norm was what
// tipped this kernel over the 256-register budget.
// MF_FOLD_NORM: only the statistics -- rstd_s[row] = 1 / rms of the row (applied to the reduced accumulators), the rows stay untouched.
template <int R>
__device__ __forceinline__ void mf_norm(bf16_t* xs, const bf16_t* nws, float* rstd_s, float eps, int lane, int wave) {
#pragma unroll
    for (int rho = wave; rho < R; rho += MG_WAVES) {
        bf16_t* row = xs + (size_t)rho * mf_kp(R / 2) + lane * 8;
        float s = 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) s += sumsq8(*(const u32x4*)(row + q * 512));
        const float rstd = rms_rstd(wave_sum(s), MG_DIM, eps);
        if (MF_FOLD_NORM) {
            if (lane == 0) rstd_s[rho] = rstd;
            continue;
        }
#pragma unroll 1
        for (int q = 0; q < 4; ++q) *(u32x4*)(row + q * 512) = rms_apply8(*(const u32x4*)(row + q * 512), rstd, *(const u32x4*)(nws + q * 512 + lane * 8));
    }
}

// ---- hand-off of the activation vectors: TAG-FREE ROWS through LDS-DMA (round 6) ------------------------------------------------------
// decode_mega.hip exchanges every vector as 8-byte granules {2 x bf16, 32-bit tag}: right for 2 rows, where a hand-off is a latency.  At 8 /
// 16 rows it is a VOLUME: every workgroup polls every row of every vector -- 131 KB for each of x, h and the attention vector and 360 KB for
// the activations at 16 rows, ~190 MB per layer chip-wide against 103 MB of weights -- and a poll keeps at most 12 x 16 B per lane in flight
// (registers), so the activation vector came in four request -> land -> unpack rounds of ~3 us each (profiles/r05u: 12.4 of a layer's 72 us).
// Here the four vectors are plain bf16 rows X | ATT | H | ACT [2 NU][columns] -- 2 bytes per value instead of 4 -- and a consumer pulls whole
// rows straight into the operand rows `xs` with global_load_lds_dwordx4 (1 KB per wave instruction, no registers, no unpacking): all of a
// wave's 4-12 requests are in flight at once, ONE round.
//
// Validity without tags: a slot holds either a value or the SENTINEL 0xFFFFFFFF per 32-bit word (two bf16 NaNs with every bit set; a
// producer never stores that word: mf_word).  Producers publish with 4-byte (x, h, act) or 16-byte (attention output) agent-scope
// write-through stores, consumers load agent-scope (sc1) and look for the sentinel in what landed; a request that still shows one is
// repeated.  Every slot is put back to the sentinel by the thread that wrote it, once every reader is known to be done and before the
// next poll of that wave (whose `s_waitcnt vmcnt(0)` also completes the wave's stores) -- i.e. before anything this workgroup publishes
// later: a consumer that has seen those later publications can only find the sentinel or the NEW value (cdna_hip_programming.md Guideline
// 16, "sc1 payload -> vmcnt(0) -> sc1 flag"; the flag here is the next vector).  Who is done when:
//   H   (written P2 l, read P3 l)   every workgroup publishes act(l) behind its read     -> reset in front of the x poll of layer l + 1 (LM head: the last x poll)
//   ACT (written P3 l, read P4 l)   every workgroup publishes x(l) behind its read       -> reset in front of the attention-vector poll of layer l + 1 (layer 23: behind the LM head's poll)
//   X   (written P4 l, read P0 l+1) every workgroup publishes q/k/v(l+1) behind its read, and a complete attention vector needs every
//                                   workgroup's q/k/v rows                                -> reset in front of the h poll of layer l + 1 (layer 23's stays: reset in front of the next launch's first poll)
//   ATT (written P1 l, read P2 l)   every workgroup publishes h(l) behind its read       -> reset in front of the act poll of layer l
// There are TWO sets of the four vectors, used by even and odd layers: the thread that puts a slot back does so a whole layer (five hand-offs of
// its own) before it writes the slot again, which takes the order in which a reset and the next value become visible out of the discussion.
// mv_begin fills all of them with the sentinel (a fallback leaves them in any state).  q/k/v (3 KB per attention workgroup: a latency) and the
// split-attention partials stay tagged granules.
#define MF_SENTINEL 0xFFFFFFFFu
__device__ __forceinline__ uint32_t mf_word(bf16_t lo, bf16_t hi) {   // two adjacent columns of one row as they are published
    const uint32_t d = (uint32_t)lo | ((uint32_t)hi << 16);
    return d == MF_SENTINEL ? 0x7FFF7FFFu : d;   // two NaNs stay two NaNs
}
__device__ __forceinline__ void mf_publish(bf16_t* p, uint32_t word) {
    __hip_atomic_store((uint32_t*)p, word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void mf_publish16(bf16_t* p, u32x4 v) {
    v.x = v.x == MF_SENTINEL ? 0x7FFF7FFFu : v.x; v.y = v.y == MF_SENTINEL ? 0x7FFF7FFFu : v.y;
    v.z = v.z == MF_SENTINEL ? 0x7FFF7FFFu : v.z; v.w = v.w == MF_SENTINEL ? 0x7FFF7FFFu : v.w;
    // (s_nop 1: hipcc neither counts nor pads inline-asm memory operations -- a VALU write to the store's data registers needs wait states
    // behind a store of more than 64 bits, cdna_hip_programming.md 5.7; attn_core.h store16_agent does the same)
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void mf_reset16(bf16_t* p) {
    const u32x4 v = {MF_SENTINEL, MF_SENTINEL, MF_SENTINEL, MF_SENTINEL};
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}
// 64 lanes x 16 B of global memory -> 1 KB of LDS at `lds` (wave-uniform byte address), no registers: global_load_lds_dwordx4 (LDS address =
// M0 + 16 x lane), agent scope.  Completion is the wave's vmcnt like any load; the wave reads only what it requested itself.
__device__ __forceinline__ void mf_row_to_lds(uint32_t lds, const void* g) {
    uint32_t m0_save;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, off sc1\n\ts_mov_b32 m0, %0"
                 : "=&s"(m0_save) : "s"(lds), "v"(g) : "memory");
}

// NCOL columns (from `src`, row stride SRC_STRIDE elements) of the R rows of a vector into the operand rows: row r, columns 512 s .. 512 s + 511
// = one request of one wave (wave w takes requests w, w + 8, ...: PER each), landing at xs[r][512 s ..].  Returns when nothing that landed
// shows the sentinel (or the wait ran out: *s_abort set).
template <int R, int NCOL, int SRC_STRIDE, typename AF>
__device__ __forceinline__ int mf_fetch_rows(const bf16_t* src, bf16_t* xs, uint32_t xs_lds, uint32_t tag, AF abort_flag_of, int* s_abort, int lane, int wave_s) {
    constexpr int KP = mf_kp(R / 2);
    constexpr int NSEG = (NCOL + 511) / 512, NREQ = R * NSEG, PER = (NREQ + MG_WAVES - 1) / MG_WAVES;
    static_assert(NCOL % 8 == 0 && PER <= 16, "request geometry");
    uint32_t pending = (1u << PER) - 1u;   // wave-uniform
    auto issue = [&]() {
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int q = wave_s + MG_WAVES * i;
            if (q < NREQ && (pending >> i & 1u)) {
                const int r = q / NSEG, sg = q % NSEG;
                const int valid = NCOL - 512 * sg >= 512 ? 64 : (NCOL - 512 * sg) / 8;   // lanes of the row's last, partial request
                if (lane < valid) mf_row_to_lds(xs_lds + (uint32_t)(r * KP + 512 * sg) * 2u, src + (size_t)r * SRC_STRIDE + 512 * sg + lane * 8);
            }
        }
    };
    for (int tries = 0;; ++tries) {
        issue();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        mega_loads_have_landed();
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int q = wave_s + MG_WAVES * i;
            if (q < NREQ && (pending >> i & 1u)) {
                const int r = q / NSEG, sg = q % NSEG;
                const int valid = NCOL - 512 * sg >= 512 ? 64 : (NCOL - 512 * sg) / 8;
                bool bad = false;
                if (lane < valid) {
                    const u32x4 v = *(const u32x4*)(xs + (size_t)r * KP + 512 * sg + lane * 8);
                    bad = v.x == MF_SENTINEL || v.y == MF_SENTINEL || v.z == MF_SENTINEL || v.w == MF_SENTINEL;
                }
                if (!__any(bad)) pending &= ~(1u << i);
            }
        }
        pending = __builtin_amdgcn_readfirstlane(pending);
        if (pending == 0u) return tries;   // attempts that came back incomplete (the instrumented build records it)
        if ((tries & 1023) == 1023) {  // a peer is missing (grid not co-resident?) or already gave up
            uint32_t* abort_flag = mega_abort_ptr(abort_flag_of);
            const bool peer = __hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u;
            if (peer || tries >= MG_SPIN_LIMIT) {
                if (!peer) mega_report_timeout(abort_flag, tag);
                __hip_atomic_store(abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                *(volatile int*)s_abort = 1;
                return tries;
            }
        }
        __builtin_amdgcn_s_sleep(1);
    }
}

}  // namespace

// the exchange buffer of the MFMA phases (`red`) and the wave partials of the attention phase (`sm`: one slot per key range of the
// workgroup, up to 8) are never live together (barriers between the phases): ONE region
__host__ __device__ constexpr size_t mfma_red_bytes(int nu) {
    return (size_t)MG_WAVES * MF_RED_TILES * 256 * 4 > (size_t)(nu > 4 ? 8 : nu) * ATTN_WAVES * 16 * 12 * 4 ? (size_t)MG_WAVES * MF_RED_TILES * 256 * 4
                                                                                                                : (size_t)(nu > 4 ? 8 : nu) * ATTN_WAVES * 16 * 12 * 4;
}
__host__ __device__ constexpr size_t mfma_lds_bytes(int nu) {
    return (size_t)2 * nu * mf_kp(nu) * 2 + mfma_red_bytes(nu) + (size_t)3 * 2 * 128 * 2 + (size_t)2 * MG_DIM * 2;   // operand rows, red | sm, q/k/v, norm weights
}

template <int NU>
__global__ void __launch_bounds__(MG_THREADS, 1) decode_mfma_kernel(const MegaArgs a) {
    constexpr int R = 2 * NU;
    static_assert(NU == 4 || NU == 8, "instances: 4 or 8 utterances");
    constexpr int MF_KP = mf_kp(NU);
    extern __shared__ __attribute__((aligned(16))) unsigned char mf_lds[];
    bf16_t* xs = (bf16_t*)mf_lds;                                                        // [R][MF_KP] operand rows of the current phase
    float* red = (float*)(mf_lds + (size_t)R * MF_KP * 2);                               // [8 waves][4 tiles][64 lanes][4]
    float (*sm)[16][12] = (float(*)[16][12])red;                                         // [ranges of the workgroup][8 waves][16][12]: the same region
    bf16_t* qkv_s = (bf16_t*)(mf_lds + (size_t)R * MF_KP * 2 + mfma_red_bytes(NU));      // q | k_new | v_new of this workgroup's head, both rows
    bf16_t* nws = qkv_s + 3 * 2 * 128;                                                   // [2][2048]: attention norm | ffn norm weights of the layer (final norm: [0])
    __shared__ float rstd_s[R];   // MF_FOLD_NORM: 1 / rms of the operand rows of the current norm
    __shared__ int s_abort;
    __shared__ int geo[5][NU];   // P, cs, nact, kw, ranges per workgroup -- per utterance
    __shared__ int pos_s[NU];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, b = blockIdx.x;
    // a launch queued behind one that gave up returns at once (mv_run rolls the whole chunk back): its inputs are not what they should be
    if (tid == 0) s_abort = __hip_atomic_load(a.abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u;
    if (tid < NU) pos_s[tid] = a.pos[tid];
    __syncthreads();
    if (*(volatile int*)&s_abort != 0) return;
#define MF_LAYER(i) (*((const __attribute__((address_space(4))) MegaLayer*)(a.layers) + (i)))
#define MF_ABORTED() (*(volatile int*)&s_abort != 0)
#define MF_ABORT_RETURN() do { if (MF_ABORTED()) return; } while (0)
#define MF_GEO(k, u) __builtin_amdgcn_readfirstlane(geo[k][u])
    // optional phase stamps (instrumented library only: `make instr` compiles this file under -DMEGA_INSTR=1 too; workgroups 0 and 255, wave 0):
    // 0 layer top, 1 x ready, 2 norm done, 3 wqkv partial tiles stored, 4 q/k/v published, 5 q/k/v ready, 6 key ranges scored, 7 attention
    // partial / output published, 8 attention vector ready, 9 h published, 10 h ready, 11 norm done, 12 w1|w3 tiles stored, 13 act published,
    // 14 act ready (8 utterances: second half), 15 w2 tiles stored, 16 x published (tools/mega_stamps.py with MV_UTT=4 | 8)
#if defined(MEGA_INSTR) && MEGA_INSTR
    const bool stamping = a.stamps != nullptr && (b == 0 || b == 255) && wave == 0;
#define MF_STAMP(l, k) do { if (stamping && lane == 0) (a.stamps + (size_t)(b == 0 ? 0 : 1) * n_layer * MG_STAMP_SLOTS)[(l) * MG_STAMP_SLOTS + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#define MF_TRIES(l, k, expr) do { const int tries_ = (expr); if (stamping && lane == 0) (a.stamps + (size_t)(b == 0 ? 0 : 1) * n_layer * MG_STAMP_SLOTS)[(l) * MG_STAMP_SLOTS + (k)] = (unsigned long long)tries_; } while (0)
#else
#define MF_STAMP(l, k) do { } while (0)
#define MF_TRIES(l, k, expr) do { (void)(expr); } while (0)
#endif
    auto abort_flag_of = [&]() { return a.abort_flag; };
    const uint32_t epoch = a.step_counter[0] * 256u;
    const int n_layer = a.n_layer;

    // attention: this workgroup = (utterance, row, head, part): 8 / NU workgroups per (utterance, row, head) -- decode_mega.hip's MEGA_ATTN_V2
    // partition -- but with FEWER key ranges than attn_core.h's 8 above 256 keys.  A workgroup walks its ranges one after the other through
    // one K/V register set, every range a request -> land -> score round whose latency nothing hides; a range of 256 keys fills the set (8
    // waves x 8 slots x 4 keys), so fewer, fuller ranges mean fewer rounds (4 utterances: 4 ranges up to 1024 keys, r05f: 1147-1158 vs
    // 1174-1180 us/step; 8 utterances: ceil(P / 256) ranges, r05j).
    constexpr int BPH = 8 / NU, SPB = NU;
    const int at_u = b / (32 * BPH), at_r = (b / (16 * BPH)) & 1, at_h = (b / BPH) & 15;
#pragma unroll
    for (int u = NU - 1; u >= 0; --u) {
        const int Pu = pos_s[u] + 1;  // keys [0, P): the cache holds [0, P-1), key P-1 is produced by this step's QKV phase
        // MF_NSPLIT: 2 = as FEW ranges as the K/V register set allows (256 keys each: 8 waves x 8 slots x 4 keys), 1 = 1 | 4 | 8, 0 = attn_core.h's 1 | 8
        constexpr int NSM = MF_NSPLIT >= 0 ? MF_NSPLIT : (NU >= 8 ? 2 : 1);
        const int ns = Pu <= ATTN_KEYS_PER_BLOCK ? 1 : NSM == 2 ? (Pu + ATTN_KEYS_PER_BLOCK - 1) / ATTN_KEYS_PER_BLOCK
                                                     : (NSM == 1 && Pu <= 4 * ATTN_KEYS_PER_BLOCK ? 4 : ATTN_NSPLIT);
        const int csu = (((Pu + ns - 1) / ns) + 3) & ~3, nactu = (Pu + csu - 1) / csu;
        if (tid == 0) { geo[0][u] = Pu; geo[1][u] = csu; geo[2][u] = nactu; geo[3][u] = attn_keys_per_wave(csu); geo[4][u] = (ns + BPH - 1) / BPH; }
    }
    __syncthreads();
    const int rpw = geo[4][at_u];                     // key ranges per workgroup of this (utterance, row, head)
    const int at_s = (b % BPH) * rpw;                 // this workgroup's first range
    const bool any_att = at_s < geo[2][at_u];

    // the exchanged vectors sit in ONE slab at compile-time offsets (mfma_handoff_rows below; engine.hip arms the rows with the sentinel):
    // q/k/v as tagged granules {columns 2 j | 2 j + 1 of ONE row, tag}, [R][3 x 1024]; then the tag-free rows X | ATT | H | ACT
    gran_t* const g_qkv = a.g_x;
    // (two sets, by layer parity: a slot is put back to the sentinel a whole layer before it is written again)
    constexpr size_t ROWS_SET = (size_t)R * (3 * MG_DIM + MG_INTER);
    bf16_t* const rows0 = (bf16_t*)(a.g_x + (size_t)NU * 3 * MG_DIM);
    const uint32_t xs_lds = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void*)mf_lds);

    // reduce / publish thread t = (column pair j, row r): consecutive threads = the rows of one pair of weight rows
    uint32_t xres = 0u, hres = 0u;   // residual of columns 8 b + 2 pj, + 1 of row pr: threads < 4 R
    if (tid < MF_R_WO / 2 * R) {
        const int pr = tid % R, pj = tid / R;
        xres = *(const uint32_t*)(a.x_in + (size_t)pr * MG_DIM + 8 * b + 2 * pj);
    }

    u32x4 wq[12];      // wqkv fragments: rows 0..15 x 8 k-steps, rows 16..23 packed x 4 (head: 8 k-steps of one tile) -- requested at the end of a layer for the next
    u32x4 nwp = {0u, 0u, 0u, 0u};   // (MF_FOLD_NORM 0 only) this thread's 16 bytes of the next layer's norm weights (threads < 256: attention norm, the others: ffn norm), on their
                       // way to `nws` -- requested with the layer's wqkv fragments, stored to LDS once the x poll has seen them land

    // (lane, wave) below are the CALLER's ids, made opaque once per layer: every fragment address is a function of them and of the layer's
    // base pointer -- left alone hipcc hoists the ~90 loop-invariant offsets out of the layer loop, keeps them live and spills
    // (first build: 1418 spilled VGPRs); opaque, each address is one or two VALU ops next to its load (decode_mega.hip does the same)
    auto request_wqkv = [&](int l, int lane, int wave) {
        const bf16_t* slice = MF_LAYER(l).t[0] + (size_t)b * MF_R_QKV * MG_DIM;
        const int ws = __builtin_amdgcn_readfirstlane(wave);
        const uint32_t o0 = mf_lane_off<MF_R_QKV>(0, lane), o1 = mf_lane_off_half<MF_R_QKV, 16, 4>(lane);
#pragma unroll
        for (int i = 0; i < 8; ++i) wq[i] = mf_frag<MF_R_QKV>(slice, 8 * ws + i, o0);          // rows 0..15, k-step 8 w + i
#pragma unroll
        for (int i = 0; i < 4; ++i) wq[8 + i] = mf_frag<MF_R_QKV>(slice, 8 * ws + i, o1);      // rows 16..23, k-steps 8 w + i | 8 w + 4 + i
        const int t = wave * 64 + lane;
        if (!MF_FOLD_NORM) nwp = load16((t < 256 ? MF_LAYER(l).attn_norm : MF_LAYER(l).ffn_norm) + (t & 255) * 8);
    };
    request_wqkv(0, lane, wave);

    // this workgroup's K/V rows of key range `split` of utterance u (rows of EARLIER steps); part 0: K and V, 1: K only, 2: V only
    auto issue_kv = [&](u32x4 (&kd)[ATTN_U], u32x4 (&vd)[ATTN_U], int u, int layer, int split, int part, int lane, int wave) {
        if (part != 2) {
#pragma unroll
            for (int q = 0; q < ATTN_U; ++q) kd[q] = u32x4{0u, 0u, 0u, 0u};
        }
        if (part != 1) {
#pragma unroll
            for (int q = 0; q < ATTN_U; ++q) vd[q] = u32x4{0u, 0u, 0u, 0u};
        }
        const int Pu = MF_GEO(0, u), csu = MF_GEO(1, u), nactu = MF_GEO(2, u), kwu = MF_GEO(3, u);
        const int go = lane >> 4, dlo = lane & 15, s0o = split * csu + wave * kwu;
        const size_t layer_off = (size_t)layer * a.kv_layer_elems;
        const bf16_t* kc = a.kcache + layer_off;
        const bf16_t* vc = a.vcache + layer_off;
        const size_t head_base = ((size_t)(2 * u + at_r) * MG_HEADS + at_h) * (size_t)a.cache_len * 128 + dlo * 8;
        const int last_old = max(Pu - 2, 0);   // row P-1 is being written by this launch: never touched here (taken from the hand-off)
        if (Pu > 1 && split < nactu) {
            if (part != 2) {
#pragma unroll
                for (int q = 0; q < ATTN_U; ++q)
                    if (q * 4 < kwu) kd[q] = load_stream16(kc + head_base + (size_t)min(s0o + q * 4 + go, last_old) * 128);
            }
            if (part != 1) {
#pragma unroll
                for (int q = 0; q < ATTN_U; ++q)
                    if (q * 4 < kwu) vd[q] = load_stream16(vc + head_base + (size_t)min(s0o + q * 4 + go, last_old) * 128);
            }
        }
    };

    for (int l = 0; l < n_layer; ++l) {
        int lane_o = lane, wave_o = wave;
        asm volatile("" : "+v"(lane_o), "+v"(wave_o));
        const int tid_o = wave_o * 64 + lane_o;
        const int wave_s = __builtin_amdgcn_readfirstlane(wave_o);   // wave-uniform: the k-step part of a fragment address stays in SGPRs
        const int pr = tid_o % R, pj = tid_o / R;   // reduce / publish thread = (column pair pj, row pr): from the opaque id, see above
        // declared per layer: a register array that is assigned under a condition (the K/V rows of a workgroup that attends) and declared
        // outside the loop is live around the whole loop for the compiler
        u32x4 wo_r[4];     // wo: 8 rows packed, 4 fragments = 8 k-steps
        u32x4 w13[24];     // w1: rows 0..15 x 8 k-steps + rows 16..21 packed x 4; w3 the same behind it
        u32x4 w2_r[11];    // w2: 8 rows packed, 11 fragments = 22 k-steps
        u32x4 kv[ATTN_U], vv[ATTN_U];
        const __attribute__((address_space(4))) MegaLayer& L = MF_LAYER(l);
        const uint32_t tag0 = epoch + (uint32_t)l * 8u + 1u;  // + phase index
        // this layer's rows X | ATT | H | ACT, and the previous layer's (layer 0: the previous launch's last layer)
        bf16_t* const r_x = rows0 + (size_t)(l & 1) * ROWS_SET;
        bf16_t* const r_attn = r_x + (size_t)R * MG_DIM;
        bf16_t* const r_h = r_attn + (size_t)R * MG_DIM;
        bf16_t* const r_act = r_h + (size_t)R * MG_DIM;
        bf16_t* const p_x = rows0 + (size_t)((l > 0 ? l - 1 : n_layer - 1) & 1) * ROWS_SET;
        bf16_t* const p_h = p_x + (size_t)2 * R * MG_DIM;
        bf16_t* const p_act = p_h + (size_t)R * MG_DIM;

        // ================= P0: RMSNorm + wqkv (+ KV-cache write)        fast_model.py:179,209,218 ====================
        __syncthreads();
        MF_STAMP(l, 0);
        if (l == 0) {
            for (int it = tid_o; it < R * (MG_DIM / 8); it += MG_THREADS) {
                const int rho = it / (MG_DIM / 8), c = (it % (MG_DIM / 8)) * 8;
                *(u32x4*)(xs + (size_t)rho * MF_KP + c) = load16(a.x_in + (size_t)rho * MG_DIM + c);
            }
        } else {
            if (tid_o < MF_R_WO / 2 * R) mf_publish(p_h + (size_t)pr * MG_DIM + 8 * b + 2 * pj, MF_SENTINEL);   // h of layer l - 1: every reader is done (header)
            MF_TRIES(l, 17, (mf_fetch_rows<R, MG_DIM, MG_DIM>(p_x, xs, xs_lds, tag0 - 8u + 4u, abort_flag_of, &s_abort, lane_o, wave_s)));
        }
        if (!MF_FOLD_NORM) *(u32x4*)(nws + tid_o * 8) = nwp;   // this layer's norm weights (requested with its wqkv fragments; landed: the poll ended in vmcnt(0))
        __syncthreads();
        MF_ABORT_RETURN();
        MF_STAMP(l, 1);
        if (any_att) issue_kv(kv, vv, at_u, l, at_s, 1, lane_o, wave_o);   // K rows of the first range: they land under the norm and the MFMAs
        __builtin_amdgcn_sched_barrier(0);
        mf_norm<R>(xs, nws, rstd_s, a.eps, lane_o, wave_o);
        __syncthreads();
        MF_STAMP(l, 2);
        {
            f32x4_t acc0 = {0.f, 0.f, 0.f, 0.f}, lo = acc0, hi = acc0;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const bf16x8_t b0 = mf_bfrag<R>(xs, 8 * wave_o + i, lane_o), b1 = mf_bfrag<R>(xs, 8 * wave_o + 4 + i, lane_o);
                acc0 = mf_mma(wq[i], b0, acc0);
                acc0 = mf_mma(wq[4 + i], b1, acc0);
                lo = mf_mma(wq[8 + i], b0, lo);
                hi = mf_mma(wq[8 + i], b1, hi);
                if (i & 1) __builtin_amdgcn_sched_barrier(0);
            }
            mf_put(red, wave_o, 0, lane_o, acc0);
            mf_put(red, wave_o, 1, lane_o, mf_half_tile(lo, hi, lane_o));
        }
        __syncthreads();
        MF_STAMP(l, 3);
        if (tid_o < MF_R_QKV / 2 * R) {
            const int r = pr, n = MF_R_QKV * b + 2 * pj;   // weight rows n, n + 1 (an even pair: never across the q | k | v boundaries)
            // rows 0..15: tile 0; rows 16..23: the packed half tile, k-range halves in row slots i - 16 and i - 8
            const float rs = MF_FOLD_NORM ? rstd_s[r] : 1.0f;   // (folded norm: the row's 1 / rms on the reduced accumulator)
            auto out = [&](int i) { return f2bf(rs * (i < 16 ? mf_get(red, 0, i, r) : mf_get(red, 1, i - 16, r) + mf_get(red, 1, i - 8, r))); };
            const uint32_t y = (uint32_t)out(2 * pj) | ((uint32_t)out(2 * pj + 1) << 16);
            gran_store(g_qkv + (size_t)r * (3 * MG_DIM / 2) + n / 2, y, tag0 + 0u);
            if (n >= MG_DIM) {  // K / V rows also go to the cache for the FUTURE steps (this step reads them from the granules)
                const int which = (n - MG_DIM) / MG_DIM, hd = (n - MG_DIM) % MG_DIM;
                const size_t cache_len = (size_t)a.cache_len;
                bf16_t* cache = (which ? a.vcache : a.kcache) + (size_t)l * a.kv_layer_elems + (size_t)r * MG_HEADS * cache_len * 128;
                const size_t off = ((size_t)(hd / 128) * cache_len + (geo[0][r >> 1] - 1)) * 128 + hd % 128;   // the row differs per lane here: NOT MF_GEO (readfirstlane)
                *(uint32_t*)(cache + off) = y;
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        MF_STAMP(l, 4);
        if (any_att) issue_kv(kv, vv, at_u, l, at_s, 2, lane_o, wave_o);   // V rows behind the publish, in front of the q/k/v hand-off
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();   // every publish of the workgroup has left: bulk requests may enter the CU's memory pipeline
        {   // wo rows 8 b .. 8 b + 7 (one half tile), under the q/k/v hand-off
            const bf16_t* slice = L.t[1] + (size_t)b * MF_R_WO * MG_DIM;
            const uint32_t o0 = mf_lane_off_half<MF_R_WO, 0, 4>(lane_o);
#pragma unroll
            for (int i = 0; i < 4; ++i) wo_r[i] = mf_frag<MF_R_WO>(slice, 8 * wave_s + i, o0);   // k-steps 8 w + i | 8 w + 4 + i
        }
        __builtin_amdgcn_sched_barrier(0);
        // w1 behind the attention, w3 behind the h publish: all 32 fragments (128 registers) at once, next to the wo fragments and a poll's
        // 56, left no room for the norm (first builds: 250-1400 spilled registers -- a scratch reload waits behind every row in flight)
        auto request_w1 = [&]() {
            const bf16_t* s1 = L.t[2] + (size_t)b * MF_R_W13 * MG_DIM;
            const uint32_t o0 = mf_lane_off<MF_R_W13>(0, lane_o), o1 = mf_lane_off_half<MF_R_W13, 16, 4>(lane_o);
#pragma unroll
            for (int i = 0; i < 8; ++i) w13[i] = mf_frag<MF_R_W13>(s1, 8 * wave_s + i, o0);
#pragma unroll
            for (int i = 0; i < 4; ++i) w13[8 + i] = mf_frag<MF_R_W13>(s1, 8 * wave_s + i, o1);
        };

        // ================= P1: attention over keys [0, P)               fast_model.py:218-222 =========================
        if (!any_att) request_w1();
        if (any_att) {
            {   // q, k_new, v_new of head at_h of this workgroup's ROW: 3 x 64 column-paired granules
                const int sec = tid_o >> 6, c = tid_o & 63;
                wave_poll<1>(tag0 + 0u, abort_flag_of, &s_abort,
                             [&](int, u64& v) {
                                 if (tid_o >= 192) return false;
                                 v = gran_load(g_qkv + (size_t)(2 * at_u + at_r) * (3 * MG_DIM / 2) + sec * (MG_DIM / 2) + at_h * 64 + c);
                                 return true;
                             },
                             [&](int, uint32_t p) { *(uint32_t*)(qkv_s + (sec * 2 + at_r) * 128 + 2 * c) = p; });
            }
            __syncthreads();
            MF_ABORT_RETURN();
            MF_STAMP(l, 5);
            const int u = at_u;
            const int g = lane_o >> 4, dl = lane_o & 15, at_w = wave_o, lane = lane_o;
            const int Pu = MF_GEO(0, u), csu = MF_GEO(1, u), nactu = MF_GEO(2, u), kwu = MF_GEO(3, u);
            const int n_my = min(rpw, nactu - at_s);   // >= 1: this workgroup attends
            const u32x4 qv = *(const u32x4*)(qkv_s + (0 * 2 + at_r) * 128 + dl * 8);
            // ONE K/V register set, the workgroup's ranges back to back (rolled): decode_mega.hip's second set would live in registers this
            // kernel does not have.
#pragma unroll 1
            for (int j = 0; j < n_my; ++j) {
                const int split = at_s + j;
                const int s0 = split * csu + at_w * kwu, s1 = min(min(Pu, split * csu + csu), s0 + kwu);
#pragma unroll
                for (int q = 0; q < ATTN_U; ++q) {
                    if (s0 + q * 4 + g == Pu - 1) {  // the key produced by this step: not in the cache yet for this launch
                        kv[q] = *(const u32x4*)(qkv_s + (1 * 2 + at_r) * 128 + dl * 8);
                        vv[q] = *(const u32x4*)(qkv_s + (2 * 2 + at_r) * 128 + dl * 8);
                    }
                }
                const AttnAcc w = attn_wave_partial<false>(kv, vv, qv, s0, s1, g);
                if (j + 1 < n_my) issue_kv(kv, vv, u, l, split + 1, 0, lane_o, wave_o);   // the next range streams while this one is stored / merged
                if (g == 0) {
                    float (*smj)[16][12] = sm + j * ATTN_WAVES;
                    smj[at_w][dl][0] = w.m;
                    smj[at_w][dl][1] = w.l;
#pragma unroll
                    for (int d = 0; d < 8; ++d) smj[at_w][dl][2 + d] = w.o[d];
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            __syncthreads();
            MF_STAMP(l, 6);
            __builtin_amdgcn_sched_barrier(0);
            // wave j merges range j and publishes its partial; with a single range (<= 256 keys) the merged result is the output
            AttnAcc bb;
            if constexpr (BPH == 1) {
                // 8 utterances: ONE workgroup walks all ranges of its (utterance, row, head), so the merged partials meet in LDS (the operand rows
                // are idle between the wqkv MFMAs and the attention-vector hand-off) instead of travelling through memory to wave 0 of the same
                // workgroup -- a store, its visibility and a load round trip (~2-3 us of every layer) for a hand-off nobody else takes part in.
                // Same values, same lane assignment, same attn_combine_math as attn_combine_tagged: same bits.
                float (*pm)[16][12] = (float(*)[16][12])xs;   // [range][16 dim groups][m, l, o[8]]
                if (at_w < n_my) {
                    bb = attn_merge_waves<false>(sm + at_w * ATTN_WAVES, g, dl);
                    if (nactu > 1 && g == 0) {
                        pm[at_w][dl][0] = bb.m;
                        pm[at_w][dl][1] = bb.l;
#pragma unroll
                        for (int d = 0; d < 8; ++d) pm[at_w][dl][2 + d] = bb.o[d];
                    }
                }
                if (nactu > 1) __syncthreads();   // (uniform over the workgroup: the geometry of its utterance)
                if (at_w == 0) {
                    if (nactu > 1) {
                        float ms[2], ls[2], t[2][8];
#pragma unroll
                        for (int q = 0; q < 2; ++q) {
                            const int sp = min(g + 4 * q, nactu - 1);   // clamped: weighted 0 below
                            ms[q] = (g + 4 * q < nactu) ? pm[sp][dl][0] : -INFINITY;
                            ls[q] = pm[sp][dl][1];
#pragma unroll
                            for (int d = 0; d < 8; ++d) t[q][d] = pm[sp][dl][2 + d];
                        }
                        attn_combine_math<false>(bb, ms, ls, t);
                    }
                    if (g == 0)   // 8 dims per lane: one 16-byte store into the row
                        mf_publish16(r_attn + (size_t)(2 * u + at_r) * MG_DIM + at_h * 128 + dl * 8, attn_normalise(bb));
                }
            } else {
                u64* slots = (u64*)a.partial + ((size_t)(2 * u + at_r) * MG_HEADS + at_h) * ATTN_NSPLIT * ATTN_TPART_STRIDE;
                if (at_w < n_my) {
                    bb = attn_merge_waves<false>(sm + at_w * ATTN_WAVES, g, dl);
                    if (nactu > 1) attn_publish_tagged(bb, slots + (size_t)(at_s + at_w) * ATTN_TPART_STRIDE, tag0 + 5u, lane);
                }
                if (at_w == 0 && at_s == 0) {   // the workgroup of the first ranges combines (utterance, row, head) and publishes its output
                    bool last = true;
                    if (nactu > 1 && !attn_combine_tagged<false>(bb, slots, tag0 + 5u, nactu, lane, MG_SPIN_LIMIT)) {
                        uint32_t* af = a.abort_flag;
                        mega_report_timeout(af, tag0 + 5u);
                        __hip_atomic_store(af, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        *(volatile int*)&s_abort = 1;
                        last = false;
                    }
                    if (last && g == 0)   // 8 dims per lane: one 16-byte store into the row
                        mf_publish16(r_attn + (size_t)(2 * u + at_r) * MG_DIM + at_h * 128 + dl * 8, attn_normalise(bb));
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            MF_STAMP(l, 7);
            __syncthreads();
            request_w1();
        }
        __builtin_amdgcn_sched_barrier(0);

        // ================= P2: h = x + wo(attn)                         fast_model.py:224-226,179 =====================
        __syncthreads();
        if (l == 0) {   // x of the previous launch's last layer (the LM head read it; the launch boundary says every reader is done)
            if (tid_o < MF_R_W2 / 2 * R) mf_publish(p_x + (size_t)pr * MG_DIM + 8 * b + 2 * pj, MF_SENTINEL);
        } else if (tid_o < MF_R_W13 / 2 * R) mf_publish(p_act + (size_t)pr * MG_INTER + MF_R_W13 * b + 2 * pj, MF_SENTINEL);   // act of layer l - 1
        MF_TRIES(l, 18, (mf_fetch_rows<R, MG_DIM, MG_DIM>(r_attn, xs, xs_lds, tag0 + 1u, abort_flag_of, &s_abort, lane_o, wave_s)));
        __syncthreads();
        MF_ABORT_RETURN();
        MF_STAMP(l, 8);
        {
            f32x4_t lo = {0.f, 0.f, 0.f, 0.f}, hi = lo;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                lo = mf_mma(wo_r[i], mf_bfrag<R>(xs, 8 * wave_o + i, lane_o), lo);
                hi = mf_mma(wo_r[i], mf_bfrag<R>(xs, 8 * wave_o + 4 + i, lane_o), hi);
            }
            mf_put(red, wave_o, 0, lane_o, mf_half_tile(lo, hi, lane_o));
        }
        __syncthreads();
        if (tid_o < MF_R_WO / 2 * R) {
            const int r = pr, i = 2 * pj, n = MF_R_WO * b + i;
            const bf16_t y0 = epi_resid((bf16_t)xres, mf_get(red, 0, i, r) + mf_get(red, 0, i + 8, r));
            const bf16_t y1 = epi_resid((bf16_t)(xres >> 16), mf_get(red, 0, i + 1, r) + mf_get(red, 0, i + 9, r));
            hres = (uint32_t)y0 | ((uint32_t)y1 << 16);
            bool withhold = false;   // test hook (mv_tune "mega_fault"): see decode_mega.hip
            if (l == 1) { const int fault = a.fault; withhold = fault != 0 && fault < 1000 && b == ((fault - 1) & 255); }
            if (!withhold) mf_publish(r_h + (size_t)r * MG_DIM + n, mf_word(y0, y1));
        }
        __builtin_amdgcn_sched_barrier(0);
        MF_STAMP(l, 9);
        __syncthreads();   // published: the w3 fragments may go out (w1 is already in flight)
        {
            const bf16_t* s3 = L.t[3] + (size_t)b * MF_R_W13 * MG_DIM;
            const uint32_t o0 = mf_lane_off<MF_R_W13>(0, lane_o), o1 = mf_lane_off_half<MF_R_W13, 16, 4>(lane_o);
#pragma unroll
            for (int i = 0; i < 8; ++i) w13[12 + i] = mf_frag<MF_R_W13>(s3, 8 * wave_s + i, o0);
#pragma unroll
            for (int i = 0; i < 4; ++i) w13[20 + i] = mf_frag<MF_R_W13>(s3, 8 * wave_s + i, o1);
        }
        __builtin_amdgcn_sched_barrier(0);

        // ================= P3: act = silu(w1 n(h)) * w3 n(h)            fast_model.py:180,237 =========================
        if (l > 0 && tid_o < MF_R_W2 / 2 * R) mf_publish(p_x + (size_t)pr * MG_DIM + 8 * b + 2 * pj, MF_SENTINEL);   // x of layer l - 1 (layer 0: done in front of the attention-vector poll)
        MF_TRIES(l, 19, (mf_fetch_rows<R, MG_DIM, MG_DIM>(r_h, xs, xs_lds, tag0 + 2u, abort_flag_of, &s_abort, lane_o, wave_s)));
        __syncthreads();
        MF_ABORT_RETURN();
        MF_STAMP(l, 10);
        mf_norm<R>(xs, nws + MG_DIM, rstd_s, a.eps, lane_o, wave_o);
        __syncthreads();
        MF_STAMP(l, 11);
        {
            f32x4_t c0 = {0.f, 0.f, 0.f, 0.f}, l1 = c0, h1 = c0, c2 = c0, l3 = c0, h3 = c0;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const bf16x8_t b0 = mf_bfrag<R>(xs, 8 * wave_o + i, lane_o), b1 = mf_bfrag<R>(xs, 8 * wave_o + 4 + i, lane_o);
                c0 = mf_mma(w13[i], b0, c0);
                c0 = mf_mma(w13[4 + i], b1, c0);
                l1 = mf_mma(w13[8 + i], b0, l1);
                h1 = mf_mma(w13[8 + i], b1, h1);
                c2 = mf_mma(w13[12 + i], b0, c2);
                c2 = mf_mma(w13[16 + i], b1, c2);
                l3 = mf_mma(w13[20 + i], b0, l3);
                h3 = mf_mma(w13[20 + i], b1, h3);
                __builtin_amdgcn_sched_barrier(0);
            }
            mf_put(red, wave_o, 0, lane_o, c0);
            mf_put(red, wave_o, 1, lane_o, mf_half_tile(l1, h1, lane_o));
            mf_put(red, wave_o, 2, lane_o, c2);
            mf_put(red, wave_o, 3, lane_o, mf_half_tile(l3, h3, lane_o));
        }
        __syncthreads();
        MF_STAMP(l, 12);
        if (tid_o < MF_R_W13 / 2 * R) {
            const int r = pr, n = MF_R_W13 * b + 2 * pj;
            const float rs = MF_FOLD_NORM ? rstd_s[r] : 1.0f;
            auto out = [&](int base, int i) { return rs * (i < 16 ? mf_get(red, base, i, r) : mf_get(red, base + 1, i - 16, r) + mf_get(red, base + 1, i - 8, r)); };
            const bf16_t y0 = epi_swiglu(out(0, 2 * pj), out(2, 2 * pj));
            const bf16_t y1 = epi_swiglu(out(0, 2 * pj + 1), out(2, 2 * pj + 1));
            mf_publish(r_act + (size_t)r * MG_INTER + n, mf_word(y0, y1));
        }
        __builtin_amdgcn_sched_barrier(0);
        MF_STAMP(l, 13);
        __syncthreads();   // published: the w2 rows may go out
        {   // w2 rows 8 b .. 8 b + 7, K = 5632 = 176 k-steps: 11 of each column half per wave, packed
            const bf16_t* slice = L.t[4] + (size_t)b * MF_R_W2 * MG_INTER;
            const uint32_t o0 = mf_lane_off_half<MF_R_W2, 0, 88>(lane_o);
#pragma unroll
            for (int i = 0; i < 11; ++i) w2_r[i] = mf_frag<MF_R_W2>(slice, 11 * wave_s + i, o0);   // k-steps 11 w + i (columns 0..2815) | 88 + 11 w + i (2816..5631)
        }
        __builtin_amdgcn_sched_barrier(0);

        // ================= P4: x = h + w2(act)                          fast_model.py:247,180 =========================
        {
            f32x4_t lo = {0.f, 0.f, 0.f, 0.f}, hi = lo;
            // the attention vector of this layer: every reader is done (they all published h behind it)
            if (any_att && wave_o == 0 && at_s == 0 && lane_o < 16) mf_reset16(r_attn + (size_t)(2 * at_u + at_r) * MG_DIM + at_h * 128 + lane_o * 8);
            if constexpr (NU <= 4) {   // the whole activation vector in the operand rows: one pass
                MF_TRIES(l, 20, (mf_fetch_rows<R, MG_INTER, MG_INTER>(r_act, xs, xs_lds, tag0 + 3u, abort_flag_of, &s_abort, lane_o, wave_s)));
                __syncthreads();
                MF_ABORT_RETURN();
                MF_STAMP(l, 14);
#pragma unroll
                for (int i = 0; i < 11; ++i) {
                    lo = mf_mma(w2_r[i], mf_bfrag<R>(xs, 11 * wave_o + i, lane_o), lo);
                    hi = mf_mma(w2_r[i], mf_bfrag<R>(xs, 88 + 11 * wave_o + i, lane_o), hi);
                    if ((i & 3) == 3) __builtin_amdgcn_sched_barrier(0);
                }
            } else {                   // 16 rows: the operand rows hold one column half at a time
                MF_TRIES(l, 20, (mf_fetch_rows<R, MG_INTER / 2, MG_INTER>(r_act, xs, xs_lds, tag0 + 3u, abort_flag_of, &s_abort, lane_o, wave_s)));
                __syncthreads();
                MF_ABORT_RETURN();
#pragma unroll
                for (int i = 0; i < 11; ++i) lo = mf_mma(w2_r[i], mf_bfrag<R>(xs, 11 * wave_o + i, lane_o), lo);
                __builtin_amdgcn_sched_barrier(0);
                __syncthreads();       // every wave has read the first half
                MF_TRIES(l, 21, (mf_fetch_rows<R, MG_INTER / 2, MG_INTER>(r_act + MG_INTER / 2, xs, xs_lds, tag0 + 3u, abort_flag_of, &s_abort, lane_o, wave_s)));
                __syncthreads();
                MF_ABORT_RETURN();
                MF_STAMP(l, 14);
#pragma unroll
                for (int i = 0; i < 11; ++i) hi = mf_mma(w2_r[i], mf_bfrag<R>(xs, 11 * wave_o + i, lane_o), hi);
            }
            mf_put(red, wave_o, 0, lane_o, mf_half_tile(lo, hi, lane_o));
        }
        __syncthreads();
        MF_STAMP(l, 15);
        if (tid_o < MF_R_W2 / 2 * R) {
            const int r = pr, i = 2 * pj, n = MF_R_W2 * b + i;
            const bf16_t y0 = epi_resid((bf16_t)hres, mf_get(red, 0, i, r) + mf_get(red, 0, i + 8, r));
            const bf16_t y1 = epi_resid((bf16_t)(hres >> 16), mf_get(red, 0, i + 1, r) + mf_get(red, 0, i + 9, r));
            xres = (uint32_t)y0 | ((uint32_t)y1 << 16);
            mf_publish(r_x + (size_t)r * MG_DIM + n, mf_word(y0, y1));
        }
        __builtin_amdgcn_sched_barrier(0);
        MF_STAMP(l, 16);
        // test hook (mv_tune "mega_fault" 1000): SKEW -- the publishing wave of every workgroup writes back its XCD's L2 behind the last layer's x
        // publish, which scatters the arrival of the workgroups at the following hand-offs over microseconds (first looks come back incomplete,
        // the repeated looks run, a slot that is put back too early is caught: tests/test_engine_gpu.py::test_matrix_core_hand_offs_survive_skewed_
        // workgroups; this hook turned the 5 time-outs per 10^6 launches of the misplaced reset -- see the LM head below -- into 14 per 30 runs)
        // (out here, not inside the publish block above: there it cost both instances 8 more spilled registers)
        if (l + 1 == n_layer && wave_s == 0 && a.fault == 1000) asm volatile("s_waitcnt vmcnt(0)\n\tbuffer_wbl2 sc0 sc1\n\ts_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();   // published
        if (l + 1 < n_layer) request_wqkv(l + 1, lane_o, wave_o);
        else {   // LM-head rows 11 b .. 11 b + 10 (one row tile, 11 rows), and the final norm's weights
            const bf16_t* slice = a.head_t + (size_t)b * MF_R_HEAD * MG_DIM;
            const uint32_t o0 = mf_lane_off<MF_R_HEAD>(0, lane_o);
#pragma unroll
            for (int i = 0; i < 8; ++i) wq[i] = mf_frag<MF_R_HEAD>(slice, 8 * wave_s + i, o0);
#pragma unroll
            for (int i = 8; i < 12; ++i) wq[i] = u32x4{0u, 0u, 0u, 0u};   // defined on every path: else the wqkv fragments of P0 stay live through the layer
            if (!MF_FOLD_NORM) nwp = load16(a.final_norm + (tid_o & 255) * 8);
        }
        __builtin_amdgcn_sched_barrier(0);
    }

    // ================= LM head on the last position: logits = head(norm(x))     fast_model.py:161-162 =================
    __syncthreads();
    bf16_t* const r_x = rows0 + (size_t)((n_layer - 1) & 1) * ROWS_SET;
    bf16_t* const r_h = r_x + (size_t)2 * R * MG_DIM;
    bf16_t* const r_act = r_h + (size_t)R * MG_DIM;
    if (tid < MF_R_WO / 2 * R) mf_publish(r_h + (size_t)(tid % R) * MG_DIM + 8 * b + 2 * (tid / R), MF_SENTINEL);   // h of the last layer
    mf_fetch_rows<R, MG_DIM, MG_DIM>(r_x, xs, xs_lds, epoch + (uint32_t)(n_layer - 1) * 8u + 1u + 4u, abort_flag_of, &s_abort, lane, __builtin_amdgcn_readfirstlane(wave));
    if (!MF_FOLD_NORM) *(u32x4*)(nws + tid * 8) = nwp;   // the final norm's weights (both halves hold them)
    __syncthreads();
    MF_ABORT_RETURN();
    // act of the last layer.  BEHIND the barrier: "every reader is done" follows from the WHOLE x vector being there, and a wave that comes out of
    // mf_fetch_rows has seen only its own requests (wave 0: the columns of workgroups 0..63).  Issued in front of the barrier -- as round 6 first had
    // it -- a fast workgroup put its slots back while a slow one was still pulling the activation rows: that workgroup then looked at sentinels until
    // its wait ran out (5-7 launches per 10^6 of a 2-layer model, 14 runs of 30 under the skew hook; profiles/r06m).  (The launch's end completes the store.)
    if (tid < MF_R_W13 / 2 * R) mf_publish(r_act + (size_t)(tid % R) * MG_INTER + MF_R_W13 * b + 2 * (tid / R), MF_SENTINEL);
    mf_norm<R>(xs, nws, rstd_s, a.eps, lane, wave);
    __syncthreads();
    {
        f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 8; ++i) acc = mf_mma(wq[i], mf_bfrag<R>(xs, 8 * wave + i, lane), acc);
        mf_put(red, wave, 0, lane, acc);
    }
    __syncthreads();
    if (tid < MF_R_HEAD * NU) {
        const int i = tid / NU, u = tid % NU, n = MF_R_HEAD * b + i, vocab = a.vocab;
        if (n < vocab) {
            a.logits[(size_t)(2 * u) * vocab + n] = f2bf((MF_FOLD_NORM ? rstd_s[2 * u] : 1.0f) * mf_get(red, 0, i, 2 * u));
            a.logits[(size_t)(2 * u + 1) * vocab + n] = f2bf((MF_FOLD_NORM ? rstd_s[2 * u + 1] : 1.0f) * mf_get(red, 0, i, 2 * u + 1));
        }
    }
}

// The two instances are compiled as two translation units (round 6): decode_mfma.hip = the 4-utterance instance, decode_mfma8.hip (`#define
// MF_INSTANCE 8` + this file) = the 8-utterance one, each with its own flags (Makefile: MFMA4_FLAGS / MFMA8_FLAGS).
// * -O2 instead of -O3 for both: at -O3 hipcc's register allocation spills 40 | 59 VGPRs (112 | 140 B of scratch per lane), and a scratch reload
//   waits -- in order -- behind every weight request in flight; at -O2 3 | 7, and the 8-utterance instance with `-mllvm -disable-machine-licm`
//   none at all (for the 4-utterance one that flag also gives 0 but is 1 % slower than the 3 it removes).  Same instructions' results, same tokens:
//   4 utterances 1065 -> 958 us per step, 8 utterances 1545 -> 1370 (A B A B on one box, profiles/r06flags5; -O1, -Os and -O3 without the
//   vectorisers lie in between).
// * `-mllvm -amdgpu-sched-strategy=max-memory-clause` (hipcc's other scheduling strategy): at -O3 1.5 % faster for the 16-row instance and 1-2 %
//   slower for the 8-row one (profiles/r06flags2/ab.txt); at -O2 a tie for the 8-row one (957.8-958.4 against 958.7-959.8), so both files use it.
#if MF_INSTANCE == 4
// the tag-free rows X | ATT | H | ACT inside the engine's hand-off slab (behind the q/k/v granules): mv_begin fills them with the sentinel
void mfma_handoff_rows(int n_utt, size_t* offset_bytes, size_t* bytes) {
    *offset_bytes = (size_t)n_utt * 3 * MG_DIM * sizeof(gran_t);
    *bytes = (size_t)2 * 2 * n_utt * (3 * MG_DIM + MG_INTER) * sizeof(bf16_t);   // two sets (layer parity)
}

bool mfma_decode_supported(int weight_format, int n_utt) { return weight_format == MV_W_BF16 && (n_utt == 4 || n_utt == 8); }
#endif

template <int NU>
static bool mfma_instance_fits(int n_cus) {
    const size_t lds = mfma_lds_bytes(NU);
    if (hipFuncSetAttribute((const void*)decode_mfma_kernel<NU>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return false;
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)decode_mfma_kernel<NU>, MG_THREADS, lds) != hipSuccess) return false;
    return per_cu >= 1 && (long)per_cu * n_cus >= 256;
}
#if MF_INSTANCE == 8
bool mfma_decode_fits8(int n_cus) { return mfma_instance_fits<8>(n_cus); }
#else
bool mfma_decode_fits8(int n_cus);
bool mfma_decode_fits(int n_utt, int n_cus) { return n_utt == 4 ? mfma_instance_fits<4>(n_cus) : n_utt == 8 ? mfma_decode_fits8(n_cus) : false; }
#endif

template <int NU>
static hipError_t launch_mfma_instance(hipStream_t s, const MegaArgs& a) {
    static std::atomic<bool> raised_on[64];   // the attribute belongs to the device's copy of the function
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::atomic<bool>& raised = raised_on[dev & 63];
    const size_t lds = mfma_lds_bytes(NU);
    if (!raised) {
        hipError_t e = hipFuncSetAttribute((const void*)decode_mfma_kernel<NU>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        raised = true;
    }
    hipLaunchKernelGGL((decode_mfma_kernel<NU>), dim3(256), dim3(MG_THREADS), lds, s, a);
    return hipGetLastError();
}
#if MF_INSTANCE == 8
hipError_t launch_decode_mfma8(hipStream_t s, const MegaArgs& a) { return launch_mfma_instance<8>(s, a); }
#else
hipError_t launch_decode_mfma8(hipStream_t s, const MegaArgs& a);
hipError_t launch_decode_mfma(hipStream_t s, const MegaArgs& a) {
    if (a.head_t == nullptr) return hipErrorInvalidValue;
    return a.n_utt == 4 ? launch_mfma_instance<4>(s, a) : a.n_utt == 8 ? launch_decode_mfma8(s, a) : hipErrorInvalidValue;
}
#endif
