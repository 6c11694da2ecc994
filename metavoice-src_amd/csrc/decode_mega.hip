// Persistent decode-step kernel: all 24 layers + the LM head of one token in ONE launch (MetaVoice-1B shape; NU = 1, 2 or 4
// utterances = 2 NU CFG rows decoded together: the weight rows a wave holds in registers are applied to every utterance, so
// the weight stream and the six hand-offs of a layer are shared by all of them).  The FOLD instance (one utterance, bf16 / int4) also
// draws the token (sample_core.h), does the bookkeeping of decode_n_tokens, builds the next input rows and goes on with the next
// step: up to 512 steps per launch.  Same arithmetic, in the same order, as the per-layer launches (gemv.hip / attention.hip /
// sampling.hip share gemv_core.h / attn_core.h / sample_core.h with this file): the paths produce identical bits
// (tests/test_engine_gpu.py).
//
// Why (profiles/r01b_attn_sampler/NOTES.md): with one launch per op a decode step is 122 dependent launches of 1.5-9 us
// of streaming each; every launch pays ~2.8 us of boundary + ramp + tail during which HBM idles, ~34 % of the step.  The
// weights do not depend on the activations, so here 256 resident workgroups (one per CU) each own a fixed slice of every
// weight matrix and issue the NEXT phase's weight loads (into registers: the register file is the largest on-chip
// memory, 512 KB per CU) BEFORE waiting for the current phase's input: the weight stream never stops, and the
// all-to-all dependency between phases costs one hand-off instead of a kernel boundary plus a cold start.
// The hand-off protocol (tagged granules, bounded per-wave polls) lives in mega_core.h.
#include <atomic>

#include "mega_core.h"
#include "sample_core.h"
#include "gemv_i4_core.h"
#include "gemv_i8_core.h"

// MEGA_INSTR = 1 (libmetavoice_hip_instr.so, `make instr`): phase stamps (MV_MEGA_STAMPS), per-layer debug vectors (MV_DEBUG_VECTORS) and
// the no-weights ablation (mv_tune gemv_ablate) are compiled in.  The product library carries none of them: every one was a runtime
// branch (and a live pointer) at 20-odd places of the layer loop.
#ifndef MEGA_INSTR
#define MEGA_INSTR 0
#endif

// The launch arguments.  MegaArgs travels by value in the kernarg segment (~400 bytes); hipcc loads every field the kernel names at
// the top of the kernel and keeps it live: 80-odd SGPRs for the whole launch, i.e. (round 3) 175-241 SGPRs spilled into VGPR lanes and
// 319 v_readlane + 225 s_nop per layer iteration to get them back.  Now:
//  * HOT fields -- what the layer loop needs at its publishes and requests: the granule slab (one pointer: the six vectors sit at
//    compile-time offsets from it), the layer table, K/V cache, partial slab, a few scalars -- are read by value (`a.field`): they live
//    in SGPRs for the whole launch, there are 16 of them;
//  * COLD fields -- everything the step prologue, the token draw, the time-out path and the test hooks need -- are read through
//    `KC(i)`: the kernarg pointer plus `(i >> 30)` elements, i being the step or layer counter: the optimiser cannot prove the offset
//    zero, so it can neither hoist those s_loads out of the loop nor keep their results live across it.
//    `i` must stay below 2^30 (then `i >> 30` IS zero): today's indices are the workgroup id (< 256), the layer (< 32) and the step (<= 512).
// (Tried first: a VOLATILE view of the whole struct, every access an s_load at its use.  The publishes then waited for a scalar load
// each -- 683-690 vs 678 us/step -- and in the one-step instances hipcc moved the kernarg pointer into a VGPR pair and fetched every
// field with `global_load ... sc0 sc1`: 790 vs 691 us/step, profiles/r04a.  An empty asm with a "+s" operand on the pointer does not
// compile inside divergent regions: "illegal VGPR to SGPR copy".)
typedef const __attribute__((address_space(4))) MegaArgs* mega_kargs_t;
#define KC(i) ((mega_kargs_t)__builtin_amdgcn_kernarg_segment_ptr() + ((uint32_t)(i) >> 30))

#undef MG_STAMP
#if MEGA_INSTR
#define MG_STAMP(l, k) do { if (stamping && lane == 0) (a.stamps + (size_t)(b == 0 ? 0 : 1) * n_layer * MG_STAMP_SLOTS)[(l) * MG_STAMP_SLOTS + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#define MG_DBG_ON() (a.dbg != nullptr)
#define MG_ABLATE (a.ablate)
// how many attempts of a hand-off poll came back incomplete (wave 0 of the first / last workgroup), 12 bits per hop -- x, q/k/v, attention
// vector, h, activation -- packed into stamp slot 22 of the layer (tools/mega_stamps.py prints the means)
#define MG_TRIES_DECL() uint64_t tries_word = 0
#define MG_TRIES(k, expr) do { const int t_ = (expr); tries_word += (uint64_t)min(t_, 4095) << (12 * (k)); } while (0)
#define MG_TRIES_STORE(l) do { if (stamping && lane == 0) (a.stamps + (size_t)(b == 0 ? 0 : 1) * n_layer * MG_STAMP_SLOTS)[(l) * MG_STAMP_SLOTS + 22] = tries_word; } while (0)
#else
#define MG_ABLATE 0
#define MG_STAMP(l, k) do { } while (0)
#define MG_DBG_ON() false
#define MG_TRIES_DECL() do { } while (0)
#define MG_TRIES(k, expr) do { (void)(expr); } while (0)
#define MG_TRIES_STORE(l) do { } while (0)
#endif
bool mega_instrumented() { return MEGA_INSTR != 0; }

// RMSNorm of the NU row pairs xs[u][2][2048], IN PLACE, bit-identical to gemv.hip's prologue (threads 0..255 own 8 columns
// each, of both rows of every utterance; every thread holds its columns in registers across the two barriers).
// The norm weights `nw` (columns 8 tid .. 8 tid + 7) are requested by the caller a phase ahead: loaded here, their load would
// return in order behind the K/V and weight requests already in flight and the norm would wait ~2 us for data it does not need.
template <int N>
__device__ __forceinline__ void mega_wave_sums(float (&v)[N]) { wave_sum_n<N>(v); }   // the N 64-lane sums of an epilogue, their DPP chains interleaved
template <int NU>
__device__ __forceinline__ void mega_norm(bf16_t (*xs)[2 * MG_INTER], const u32x4& nw, float* red, float eps, const int tid) {
    const int lane = tid & 63, wave = tid >> 6;   // tid: the caller's (possibly opaque) thread id -- see the layer loop
    u32x4 xv_early = {0u, 0u, 0u, 0u};
    // statistics: lane l of "column wave" q sums the squares of columns 8 (64 q + l) .. + 7, the 64 lanes are added with wave_sum -- the
    // summation order of gemv.hip's prologue.  One utterance: waves 0..3 ARE the column waves, each does both rows.  Several utterances:
    // (utterance, row) pair p is done by wave p, which plays the four column waves one after the other with the four chains unrolled
    // -- all 8 waves busy and no serial walk over the utterances (rolled, each row waited for its own ds_read and wave_sum:
    // 2.5 us per norm at four utterances against 0.7 at one, profiles/r03u).
    if constexpr (NU == 1) {
        // the 8 columns a thread APPLIES are fetched in front of the barrier -- waves 0..3 apply row 0 of the very columns
        // they just summed (no second read at all), waves 4..7 read their row-1 columns while the others sum: behind the barrier hipcc had
        // the LDS read of the operand wait for the rsqrt of the total, two dependent LDS round trips
        if (tid < MG_DIM / 8) {
            float s[2];  // both rows' chains side by side (wave_sum_n: the same tree, interleaved)
            const u32x4 x0 = *(const u32x4*)(xs[0] + tid * 8), x1 = *(const u32x4*)(xs[0] + MG_DIM + tid * 8);
            s[0] = sumsq8(x0);
            s[1] = sumsq8(x1);
            xv_early = x0;
            wave_sum_n<2>(s);
            if (lane == 0) { red[wave * 2] = s[0]; red[wave * 2 + 1] = s[1]; }
        } else xv_early = *(const u32x4*)(xs[0] + MG_DIM + (tid & 255) * 8);
    } else {
        if (wave < 2 * NU) {
            const int u = wave >> 1, r = wave & 1;
            float s[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) s[q] = sumsq8(*(const u32x4*)(xs[u] + r * MG_DIM + (q * 64 + lane) * 8));
            wave_sum_n<4>(s);
            if (lane == 0) {
#pragma unroll
                for (int q = 0; q < 4; ++q) red[u * 2 * MG_WAVES + q * 2 + r] = s[q];
            }
        }
    }
    __syncthreads();
    // apply (element-wise, any partition gives the same bits): ALL 512 threads, waves 0..3 row 0, waves 4..7 row 1 -- the apply is
    // ~150 VALU ops per 8 columns and was the longer half of the norm when only half of the waves did it
    const int r = tid >> 8, c = (tid & 255) * 8;
    // all utterances' totals, then all reads, then the applies: the ds_reads and the wave_sums of different utterances overlap
    float rstd[NU];
    u32x4 xv[NU];
#pragma unroll
    for (int u = 0; u < NU; ++u) xv[u] = NU == 1 ? xv_early : *(const u32x4*)(xs[u] + r * MG_DIM + c);
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        // == block_total over the 4 column waves: wave_sum over lanes {v0, v1, v2, v3, 0, ...} is (v0 + v1) + (v2 + v3) plus exact zeros
        // (common.h: xor-1, xor-2, then mirrors and row broadcasts that only add zeros) -- the same bits without the 7-deep DPP chain
        const float* rd = red + u * 2 * MG_WAVES + r;
        const float tot = (rd[0] + rd[2]) + (rd[4] + rd[6]);
        rstd[u] = rms_rstd(tot, MG_DIM, eps);
    }
#pragma unroll
    for (int u = 0; u < NU; ++u) *(u32x4*)(xs[u] + r * MG_DIM + c) = rms_apply8(xv[u], rstd[u], nw);
    __syncthreads();
}

// acc[i][r] += W row i (registers) . x row r (LDS), lane l covering columns 512 j + 8 l .. + 7 of chunk j
template <int KCH, int NR, int KP>
__device__ __forceinline__ void mega_dots(const u32x4* wv, const bf16_t* xs, int lane, float (&acc)[NR][2]) {
#pragma unroll
    for (int i = 0; i < NR; ++i) { acc[i][0] = 0.f; acc[i][1] = 0.f; }
#pragma unroll
    for (int j = 0; j < KCH; ++j) {
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const u32x4 xv = *(const u32x4*)(xs + r * KP + j * 512 + lane * 8);
#pragma unroll
            for (int i = 0; i < NR; ++i) acc[i][r] = dot8_bf16(wv[i * KCH + j], xv, acc[i][r]);
        }
    }
}
template <int KCH>
__device__ __forceinline__ void mega_issue_row(u32x4* wv, const bf16_t* w, int n, int K, int lane, int ablate = 0) {
    const bf16_t* row = w + (size_t)n * K;
    if (ablate & 1) {  // profiling only (mv_tune gemv_ablate bit 0): no weight requests at all, results are garbage
#pragma unroll
        for (int j = 0; j < KCH; ++j) wv[j] = u32x4{0u, 0u, 0u, 0u};
        return;
    }
#pragma unroll
    for (int j = 0; j < KCH; ++j) wv[j] = load_stream16(row + min(j * 512 + lane * 8, K - 8));
}

// Early rows (bf16, one utterance).  A phase's weight rows are requested after the previous phase's publish, so the norm / dots /
// publish of a phase (1.1-1.8 us) run with nothing new in flight (profiles/r02b_fold/stamps_final.txt).  About half of the NEXT
// phase's rows is therefore requested into LDS (global_load_lds_dwordx4: the registers are full; ~130 KB of LDS are free at one
// utterance) right after a phase's input has arrived, i.e. BEFORE its compute: ~45 KB per CU drain in the 1.5 us the compute takes,
// so the publish still finds the CU's queue nearly empty (publish-before-prefetch in time instead of in program order).  The other
// half is requested where all of it used to be.  Round 2 measured W1 + W2 together: 704.9 vs 717-721 us/step, identical bits.
//   MEGA_EARLY_W2 : chunks 0..4 of the wave's w2 row, after "h ready", consumed from LDS in P4
//   MEGA_EARLY_W1 : chunks 0, 1 of each of the wave's 3 w1 rows, after "q/k/v ready", consumed from LDS in P3
// (The same for the next layer's wqkv rows, after "act ready", measured slower: tools/experiments/decode_mega_r04_switches.patch.)
// Visibility: a wave reads only LDS slots it requested itself, and between every request and its first read sits one of the
// wave's hand-off polls, whose asm statement ends in `s_waitcnt vmcnt(0)` (mega_core.h: poll16_loads) -- loads retire in order,
// so the rows have landed.  Every early `return` of the kernel drains the wave's requests first (MG_ABORT_RETURN): a row landing
// after the workgroup's LDS has been handed to the next workgroup would be written into somebody else's memory.
#ifndef MEGA_EARLY_W1
#define MEGA_EARLY_W1 1
#endif
#ifndef MEGA_EARLY_W2
#define MEGA_EARLY_W2 1
#endif
// how far the loops over the utterances of an engine of 2 / 4 are unrolled in the wqkv (P0) and w1|w3 (P3) phases: 1 = rolled (round 2:
// unrolled they interleaved the utterances' temporaries and spilled 60-190 VGPRs; with the launch arguments out of the way 2 fits)
// Round 4, with the epilogue sums interleaved (wave_sum_n): 2 gains a little (r04ak: two utterances 887 -> 881 us/step, four 1266 -> 1261) -- two
// utterances' 6 / 12 sums run side by side; the int4 instances stay rolled (5-6 spilled registers unrolled).
#ifndef MEGA_NU_UNROLL_P0
#define MEGA_NU_UNROLL_P0 2
#endif
#ifndef MEGA_NU_UNROLL_P3
#define MEGA_NU_UNROLL_P3 2
#endif
// Attention of the engines of 2 / 4 utterances (ATTN_V2 below).  Round 3 gave every workgroup one key range of (row, head) of EVERY utterance and walked
// the utterances one after the other: K/V slice requested, landed, scored, merged, published, the next slice requested -- 16.6 us of a
// 60 us layer at four utterances, with the memory system idle for most of it (profiles/r04a/stamps_4utt.txt).  Now a workgroup belongs
// to ONE (utterance, row, head) and owns NU consecutive key ranges of it (8 / NU workgroups per (utterance, row, head)): the ranges are
// the same 8 splits of cs keys with the same 8 x kw wave slices, scored, merged and combined by the same functions in the same order --
// the bits do not change -- but the NU slices of a workgroup are one contiguous stretch of the cache, streamed through TWO register
// sets (kv / vv, and the weight registers wv[0..7] / wv[16..23], idle between the wqkv dots and the w1 request): the next range's rows
// are in flight while the current one is scored, the wave partials of all ranges wait in LDS for ONE merge (wave j merges range j) and
// ONE publish, and each (utterance, row, head) is combined once, by the workgroup of its first ranges.
// One utterance only: for engines of 2 / 4 utterances the same early rows measured slower (1375 vs 1339, 956 vs 926 us/step, profiles/r04m).
#ifndef MEGA_EW2_CHUNKS
#define MEGA_EW2_CHUNKS 5   // of the 11 chunks of a w2 row (profiles/r04ae: 7 and 9 measured)
#endif
__host__ __device__ constexpr int mega_early_w2_chunks(int) { return MEGA_EW2_CHUNKS; }
__host__ __device__ constexpr bool mega_early_w1(int) { return MEGA_EARLY_W1 != 0; }
#define MEGA_EARLY_W1_BYTES (MG_WAVES * 6 * 1024)
__device__ __forceinline__ void mega_row_to_lds(uint32_t lds, const bf16_t* g, int ablate = 0) {  // 64 lanes x 16 B -> 1 KB of LDS at `lds`
    if (ablate & 1) return;  // profiling only (mv_tune gemv_ablate bit 0): no weight requests at all, results are garbage
    uint32_t m0_save;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, off nt\n\ts_mov_b32 m0, %0"
                 : "=&s"(m0_save) : "s"(lds), "v"(g) : "memory");
}
__host__ __device__ constexpr bool mega_early_rows(int wf, int nu) {
    return wf == MV_W_BF16 && nu == 1 && (MEGA_EARLY_W1 || MEGA_EARLY_W2);
}
__host__ __device__ constexpr size_t mega_early_lds_bytes(int wf, int nu) {   // the LDS slots of the early rows, behind mega_lds_bytes(nu)
    return !mega_early_rows(wf, nu) ? 0 : (size_t)(MEGA_EARLY_W2 ? MG_WAVES * mega_early_w2_chunks(nu) * 1024 : 0) + (mega_early_w1(nu) ? MEGA_EARLY_W1_BYTES : 0);
}

// int4 (group-128) weight rows: per 2048-column chunk a lane requests 16 B of codes (32 columns) and the (scale, zero) word of
// their 128-column group -- the same loads, and through i4_dots the same arithmetic, as gemv_i4.hip.
template <int KCH>
__device__ __forceinline__ void mega_issue_row_i4(u32x4* cq, uint32_t* cz, const unsigned char* codes, const uint32_t* sz, int n, int K,
                                                  int lane) {
#pragma unroll
    for (int j = 0; j < KCH; ++j) {
        const int c = min(j * 2048 + lane * 32, K - 32);
        cq[j] = load_stream16(codes + ((size_t)n * K + c) / 2);
        cz[j] = load_global_u32(sz + (size_t)n * (K / 128) + c / 128);
    }
}
template <int KCH, int NR, int KP>
__device__ __forceinline__ void mega_dots_i4(const u32x4* cq, const uint32_t* cz, const bf16_t* xs, int K, int lane, float (&acc)[NR][2]) {
    u32x4 cw[NR][KCH];
    uint32_t sw[NR][KCH];
#pragma unroll
    for (int i = 0; i < NR; ++i)
#pragma unroll
        for (int j = 0; j < KCH; ++j) { cw[i][j] = cq[i * KCH + j]; sw[i][j] = cz[i * KCH + j]; }
    i4_dots<2, KCH, NR>(cw, sw, xs, KP, K, lane, acc);
}

// int8 (per output channel) weight rows: per 1024-column chunk a lane requests 16 B = 16 codes; the bf16 row scale rides in a
// scalar.  Same loads and, through i8_dots, the same arithmetic as gemv_i8.hip; the Linear output is bf16(bf16(sum) * scale).
template <int KCH>
__device__ __forceinline__ void mega_issue_row_i8(u32x4* cq, const signed char* codes, int n, int K, int lane) {
#pragma unroll
    for (int j = 0; j < KCH; ++j) cq[j] = load_stream16(codes + (size_t)n * K + min(j * 1024 + lane * 16, K - 16));
}
template <int KCH, int NR, int KP>
__device__ __forceinline__ void mega_dots_i8(const u32x4* cq, const bf16_t* xs, int K, int lane, float (&acc)[NR][2]) {
    u32x4 cw[NR][KCH];
#pragma unroll
    for (int i = 0; i < NR; ++i)
#pragma unroll
        for (int j = 0; j < KCH; ++j) cw[i][j] = cq[i * KCH + j];
    i8_dots<2, KCH, NR>(cw, xs, KP, K, lane, acc);
}
__device__ __forceinline__ float i8_out(float wave_total, float scale) { return round_bf(round_bf(wave_total) * scale); }

// 16-byte polls of a row-paired 2048-column vector of every utterance ([NU][2048] granules): pair n = granules 2n, 2n+1 =
// columns 2n, 2n+1 of both rows.  ONE poll loop for all utterances: a loop per utterance would pay a round trip each.
template <int NU, typename AF>
__device__ __forceinline__ int mega_poll_rowpairs(const gran_t* g, uint32_t tag, bf16_t (*xs)[2 * MG_INTER], AF abort_flag, int* s_abort,
                                                  int tid) {
    return wave_poll16<2 * NU>(g, tag, abort_flag, s_abort, [&](int k) { return (k >> 1) * (MG_DIM / 2) + tid + (k & 1) * MG_THREADS; },
                        [&](int n, uint32_t p0, uint32_t p1) {
                            const int u = n / (MG_DIM / 2), c = n % (MG_DIM / 2);
                            *(uint32_t*)(xs[u] + 2 * c) = (p0 & 0xFFFFu) | (p1 << 16);
                            *(uint32_t*)(xs[u] + MG_DIM + 2 * c) = (p0 >> 16) | (p1 & 0xFFFF0000u);
                        });
}

__host__ __device__ constexpr size_t mega_lds_bytes(int n_utt) {  // dynamic LDS of the kernel below: xs, qkv_s, sm (one per key range of the workgroup), red
    return (size_t)n_utt * (2 * MG_INTER * 2 + 3 * 2 * 128 * 2) + (size_t)n_utt * ATTN_WAVES * 16 * 12 * 4 + (size_t)n_utt * MG_WAVES * 2 * 4;
}

// sample() + the bookkeeping of decode_n_tokens + (last step of a launch) the next launch's input rows, by the workgroup of
// utterance u: fast_inference_utils.py:61-120,148-174, fast_model.py:152-157 -- sample_core.h, the arithmetic of sample_kernel.
// Inlined on purpose.  Round 3 measured it OUT of line (`noinline`, so that sample_draw's ~60 live values stay out of the layer
// loop's register allocation and engines of 2 / 4 utterances and int8 engines could draw in the launch too): every instance got
// SLOWER by 100-220 us/step (bf16 866 vs 722, int8 792 vs 680, 2 / 4 utterances 1211 / 1863 vs 1092 / 1643; profiles/r03a) -- the
// call site saves and restores the kernel's live registers through scratch (368 B/lane) and the allocation around it degraded.
template <int NU>
__device__ __forceinline__ void mega_draw_and_publish(const mega_kargs_t a, gran_t* const g_ctl, const int u, const bf16_t* lg, UttState* st_p, SampleScratch* sc_p,
                                                                const int VP, const uint32_t tag_next, const bool last_step, const int tid_t) {
    UttState& st_s = *st_p;
    SampleScratch& smp_sc = *sc_p;
    const bool was_done = st_s.done != 0;
        bool done_now = was_done;
        if (!was_done) {
            const int pos = st_s.pos, next_index = pos + 1;     // sequence index of the token decided now
            const bool forced = next_index < st_s.prompt_len;   // teacher-forced prompt token (prefill by decode steps)
            const int V = a->vocab;
            if (a->capture != nullptr && pos < a->capture_max) {
                float* cap = a->capture + ((size_t)pos * (2 * NU) + 2 * u) * V;
                for (int i = tid_t; i < V; i += MG_THREADS) { cap[i] = bf2f(lg[i]); cap[V + i] = bf2f(lg[VP + i]); }
            }
            int tok;
            if (forced) tok = a->seq[(size_t)u * a->block_size + next_index];
            else {
                int par = 0;
                const mv_sampling sp = st_s.s;
                tok = sample_draw(sp, V, next_index, nullptr, smp_sc, par, tid_t, [&](int i, float& c, float& un) { c = bf2f(lg[i]); un = bf2f(lg[VP + i]); });
            }
            done_now = (!forced && tok == st_s.eoa_token) || (next_index + 1 >= st_s.max_len);
            // the control word first: it is what the other 255 workgroups are waiting for
            if (tid_t == 0 && !last_step) gran_store(g_ctl + u, (uint32_t)tok | ((uint32_t)next_index << 12) | ((uint32_t)done_now << 31), tag_next);
            __syncthreads();  // every thread has read st_s
            if (tid_t == 0) {
                if (!forced) a->seq[(size_t)u * a->block_size + next_index] = tok;
                UttState* gs = a->utt + u;
                gs->n_tokens = next_index + 1;
                gs->pos = next_index;
                a->pos_out[u] = next_index;
                st_s.pos = next_index;
                st_s.n_tokens = next_index + 1;
                if (done_now) {
                    gs->done = 1;
                    st_s.done = 1;
                    atomicSub(a->n_active, 1);
                }
            }
            if (!done_now && last_step) {
                // the next LAUNCH's (or the per-layer launches') input rows, as sample_kernel leaves them:
                // bf16(bf16(tok_emb[tok] + pos_emb[pos]) + spk_proj * mask).  Between the steps of one launch every workgroup
                // builds them itself from the control word (below).
                const bf16_t* te = a->tok_emb + (size_t)tok * MG_DIM;
                const bf16_t* pe = a->pos_emb + (size_t)next_index * MG_DIM;
                const bf16_t* sp = a->spk_proj + (size_t)u * MG_DIM;
                bf16_t* x0 = a->x_out + (size_t)(2 * u) * MG_DIM;
                for (int i = tid_t; i < MG_DIM; i += MG_THREADS) {
                    const float base = round_bf(bf2f(te[i]) + bf2f(pe[i]));
                    x0[i] = f2bf(base + bf2f(sp[i]));
                    x0[MG_DIM + i] = f2bf(base);  // + 0 * mask: the unconditioned row gets no speaker term
                }
            }
        } else if (tid_t == 0 && !last_step) {
            // a finished (or unused) utterance keeps its last input rows, as with sample_kernel: same token, same position
            gran_store(g_ctl + u, (uint32_t)a->seq[(size_t)u * a->block_size + st_s.pos] | ((uint32_t)st_s.pos << 12) | (1u << 31), tag_next);
        }
}

template <int WF, int NU, bool FOLD>  // WF: MV_W_BF16, MV_W_INT4_G128 (LM head bf16) or MV_W_INT8 (LM head int8 too); NU utterances;
                                      // FOLD: a.n_steps decode steps per launch, the token drawn inside (else one step, logits only)
__global__ void __launch_bounds__(MG_THREADS, (NU == 1 ? 2 : 1)) decode_mega_kernel(const MegaArgs a) {   // hot fields by name, cold ones through KC()
    extern __shared__ __attribute__((aligned(16))) unsigned char mg_lds[];
    // the GEMV operand of the current phase, per utterance; the residual stream is NOT kept here: a wave produces exactly the
    // columns (8 b + wave) whose residual it needs two phases later, so those ride in registers (xres / hres)
    bf16_t (*xs)[2 * MG_INTER] = (bf16_t(*)[2 * MG_INTER])mg_lds;
    bf16_t (*qkv_s)[3 * 2 * 128] = (bf16_t(*)[3 * 2 * 128])(mg_lds + (size_t)NU * 2 * MG_INTER * 2);  // q, k_new, v_new of this block's head
    float (*sm)[16][12] = (float(*)[16][12])(mg_lds + (size_t)NU * (2 * MG_INTER * 2 + 3 * 2 * 128 * 2));  // [ranges of this workgroup][ATTN_WAVES][16][12]
    float* red = (float*)(sm + NU * ATTN_WAVES);  // [NU][MG_WAVES * 2]
    bf16_t (*const xa)[2 * MG_INTER] = xs;   // x and h (the operand of P0 / P3 and of the head): the same buffer as the attention / activation vectors
    constexpr bool EARLY_ANY = mega_early_rows(WF, NU);
    constexpr bool EARLY_W2 = EARLY_ANY && MEGA_EARLY_W2, EARLY_W1 = EARLY_ANY && mega_early_w1(NU);
    constexpr int EW2 = mega_early_w2_chunks(NU);   // chunks of the w2 row that go to LDS early
    u32x4* w2_lds = (u32x4*)(mg_lds + mega_lds_bytes(NU));                                                        // [MG_WAVES][EW2 chunks][64 lanes]
    u32x4* w1_lds = (u32x4*)(mg_lds + mega_lds_bytes(NU) + (MEGA_EARLY_W2 ? MG_WAVES * EW2 * 1024 : 0));  // [MG_WAVES][3 rows x 2 chunks][64 lanes]
    bool w1_half_issued = false;
    auto lds_addr = [&](const u32x4* p) { return __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void*)p); };
    __shared__ int s_abort;
    // per-utterance attention geometry and the residual values of each wave's own column: in LDS so that the loops over the
    // utterances stay ROLLED (unrolled they interleave the utterances' temporaries: 60-190 spilled VGPRs, and a scratch
    // reload waits, in order, for every weight request in flight)
    __shared__ int geo[4][NU];                 // P, cs, nact, kw
    __shared__ uint32_t res_s[2][NU][MG_WAVES];  // x and h residuals of column 8 b + wave, rows packed

    // n_steps >= 1: that many decode steps in this launch, the token drawn by the workgroup of its utterance (block u) at the end
    // of every step; n_steps == 0: one step, logits only (sample_kernel follows; mv_probe_kernel, the captured decode graph)
    __shared__ int pos_s[NU], done_s[NU], tok_s[NU];
    __shared__ UttState st_s;      // block u < NU: the state of utterance u, carried across the steps of the launch
    __shared__ SampleScratch smp_sc;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, b = blockIdx.x;
    constexpr bool fold = FOLD;  // compile time: the one-step instance carries no sampler code (and shows up under its own name in rocprofv3)
    const int n_steps = fold ? KC(b)->n_steps : 1;
    if (tid == 0) s_abort = 0;
    if (tid < NU) { pos_s[tid] = KC(b)->pos[tid]; done_s[tid] = fold ? KC(b)->utt[tid].done : 0; }
    if (fold && b < NU && tid == 0) st_s = KC(b)->utt[b];
    __syncthreads();
    // NOTE: this read compiles to flat_load_dword (the address escapes into the poll helpers as a generic pointer), whose wait is
    // `vmcnt(0) lgkmcnt(0)`.  Reading it through an LDS-typed pointer (ds_read_b32, lgkmcnt only) measured SLOWER for bf16: 720.5 vs
    // 687.6 us/step for one utterance, 1054 vs 1027 for two, 1587 vs 1573 for four (profiles/r03q, same box, identical tokens) -- kept
    // The per-layer pointer table is read through a CONSTANT-address-space pointer: scalar loads (s_load, lgkmcnt).  Through the plain
    // pointer hipcc fetched the pointers with vector loads next to their uses (global_load_dwordx2 + v_readfirstlane) and waited for them
    // with `s_waitcnt vmcnt(0)` -- i.e. for every weight row in flight -- in front of the address arithmetic of the next requests.
#define MG_LAYER(i) (*((const __attribute__((address_space(4))) MegaLayer*)(a.layers) + (i)))
#define MG_ABORTED() (*(volatile int*)&s_abort != 0)
#define MG_ABORT_RETURN() do { if (MG_ABORTED()) { if (EARLY_ANY) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); return; } } while (0)
    if (fold) {  // a launch queued behind the one in which the last utterance ended: nothing to do
        bool all_done = true;
#pragma unroll
        for (int u = 0; u < NU; ++u) all_done = all_done && done_s[u] != 0;
        if (all_done) return;
    }
    const uint32_t counter0 = KC(b)->step_counter[0];
    const int VP = (a.vocab + 1) & ~1;  // logits travel as row-paired granules, polled two at a time

    // attention geometry: block (r, h, s) = one of the (up to) 8 even key ranges of (row r, head h) of EVERY utterance; 8 waves x 32 keys
    // ATTN_V2: block = (utterance, row, head, part): SPB = NU consecutive key ranges starting at at_s, 8 / NU blocks per (utterance, row, head)
    constexpr bool ATTN_V2 = NU > 1;
    constexpr int BPH = ATTN_V2 ? 8 / NU : 8, SPB = ATTN_V2 ? NU : 1;
    constexpr bool ATTN_PL = MV_PERMLANE_SWAP != 0;   // common.h: measured no faster (profiles/r04w), off
    constexpr bool ATTN_DB = ATTN_V2 && WF == MV_W_BF16;   // two register sets: the quantised instances have no idle weight registers to lend (one set, ranges back to back)
    const int at_u = ATTN_V2 ? b / (32 * BPH) : 0;
    const int at_r = (b / (16 * BPH)) & 1, at_h = (b / BPH) & 15, at_s = (b % BPH) * SPB, at_w = wave;
    const int g = lane >> 4, dl = lane & 15;
#define GEO(k, u) __builtin_amdgcn_readfirstlane(geo[k][u])

    // optional phase stamps (blocks 0 and 255, wave 0): poll-wait vs compute per phase, for profiles/
#if MEGA_INSTR
    const bool stamping = a.stamps != nullptr && (b == 0 || b == 255) && wave == 0;
#endif

    u32x4 wv[24];  // the weight working set of the current GEMV phase (next phase's rows are issued into it after the dots)
    // int4: codes / (scale, zero) words.  Slots: wqkv rows 0..2 | w1 rows 0..2, w3 rows 3..5, wo 6 | w2 (3 chunks) 6..8
    constexpr bool I4 = WF == MV_W_INT4_G128, I8 = WF == MV_W_INT8;
    u32x4 cq[I8 ? 14 : 9];  // int8 slots (2 chunks per 2048-column row, 6 per w2 row): wqkv 0..5 | w1 0..5, w3 6..11, wo 12..13 | w2 0..5 | head 0..3
    uint32_t cz[9];
    float sc_a[3] = {1.f, 1.f, 1.f}, sc_b[3] = {1.f, 1.f, 1.f}, sc_c = 1.f;  // int8 row scales: wqkv|w1|head rows, w3 rows, wo|w2 row
    // phase 0 of layer 0: rows 24 b + 3 wave + i of wqkv; the RMSNorm weights of the next norm ride along with the weight rows of
    // its phase (thread t: columns 8 (t & 255) ..).  Called at the top of the launch and, with several steps per launch, at the end
    // of every step -- before the wait for the next token.
    u32x4 nw_attn, nw_ffn = u32x4{0u, 0u, 0u, 0u};
    auto issue_layer0 = [&](int lane, int wave) {  // (lane, wave): the caller's ids, opaque to the optimiser inside the step loop
        const bf16_t* w = MG_LAYER(0).wqkv;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            if (I4) mega_issue_row_i4<1>(cq + i, cz + i, MG_LAYER(0).q[0], MG_LAYER(0).sz[0], 24 * b + 3 * wave + i, MG_DIM, lane);
            else if (I8) { mega_issue_row_i8<2>(cq + 2 * i, MG_LAYER(0).q8[0], 24 * b + 3 * wave + i, MG_DIM, lane); sc_a[i] = bf2f(load_global_u16(MG_LAYER(0).s8[0] + (24 * b + 3 * wave + i))); }
            else mega_issue_row<4>(wv + 4 * i, w, 24 * b + 3 * wave + i, MG_DIM, lane, MG_ABLATE);
        }
        nw_attn = load16(MG_LAYER(0).attn_norm + ((wave * 64 + lane) & 255) * 8);
    };
    issue_layer0(lane, wave);

    u32x4 kv[ATTN_U], vv[ATTN_U];
#pragma unroll 1
    for (int step = 0; step < n_steps; ++step) {
    const uint32_t epoch = (counter0 + (uint32_t)step) * 256u;
    // this step's attention geometry (positions: a.pos at the first step, the control words of the sampling workgroups after)
    bool any_att = false;
    int first_att = -1;
#pragma unroll
    for (int u = NU - 1; u >= 0; --u) {
        const int Pu = pos_s[u] + 1;  // keys [0, P): the cache holds [0, P-1), key P-1 is produced by this step's QKV phase
        const int csu = attn_keys_per_split(Pu), nactu = (Pu + csu - 1) / csu;
        if (tid == 0) { geo[0][u] = Pu; geo[1][u] = csu; geo[2][u] = nactu; geo[3][u] = attn_keys_per_wave(csu); }
        if (at_s < nactu && (!ATTN_V2 || u == at_u)) { any_att = true; first_att = u; }
    }
    // residual values of this wave's own column n = 8 b + wave, both rows packed (row 0 | row 1 << 16): x entering the layer
    // (layer 0: the sampler's / prefill's output -- in xs already after the first step; later: what this wave published in P4) and
    // h (published in P2)
#pragma unroll 1
    for (int u = 0; u < NU; ++u) {
        if (step == 0) {
            const bf16_t* xi = KC(step)->x_in + (size_t)u * 2 * MG_DIM + 8 * b + wave;
            if (lane == 0) res_s[0][u][wave] = (uint32_t)xi[0] | ((uint32_t)xi[MG_DIM] << 16);
        } else if (lane == 0) {
            res_s[0][u][wave] = (uint32_t)xa[u][8 * b + wave] | ((uint32_t)xa[u][MG_DIM + 8 * b + wave] << 16);
        }
    }
    __syncthreads();
    const int n_layer = a.n_layer;
    // the six exchanged vectors of every utterance sit in ONE slab at compile-time offsets (engine.hip: mega_args): one live pointer
    gran_t* const g_x = a.g_x;
    gran_t* const g_qkv = g_x + (size_t)NU * MG_DIM;
    gran_t* const g_attn = g_qkv + (size_t)NU * 3 * MG_DIM;
    gran_t* const g_h = g_attn + (size_t)NU * MG_DIM;
    gran_t* const g_act = g_h + (size_t)NU * MG_DIM;
    gran_t* const g_logits = g_act + (size_t)NU * MG_INTER;   // [NU][VP], then g_ctl [NU]
    for (int l = 0; l < n_layer; ++l) {
        // Address arithmetic below is a function of (lane, wave): left alone, the compiler hoists ~60 loop-invariant 64-bit
        // addresses out of the layer loop and spills them.  Making the two ids opaque once per layer keeps every address
        // computation next to its load (a handful of VALU ops) and the kernel spill-free.
        int lane_o = lane, wave_o = wave;
        asm volatile("" : "+v"(lane_o), "+v"(wave_o));
        const int tid_o = wave_o * 64 + lane_o;
        auto abort_flag_of = [&]() { return KC(l)->abort_flag; };   // read only when a wait runs out
        // engines of several utterances run at the 256-VGPR limit: there every LDS / granule address the optimiser hoists out of the layer
        // loop (functions of lane / wave) is a spilled register, reloaded behind `s_waitcnt vmcnt(0)` -- i.e. behind every weight row in
        // flight.  Their dots, norms and epilogues therefore use the opaque ids too; the one-utterance instances keep the plain ones.
        const int lane_p = NU > 1 ? lane_o : lane, wave_p = NU > 1 ? wave_o : wave, tid_p = NU > 1 ? tid_o : tid;
        const __attribute__((address_space(4))) MegaLayer& L = MG_LAYER(l);   // fields are fetched with scalar loads where they are used
        const uint32_t tag0 = epoch + (uint32_t)l * 8u + 1u;  // + phase index
        // This block's K/V slice of utterance u (rows of EARLIER steps: independent of this step), into the kv / vv registers.
        // Addresses are clamped to the newest row an EARLIER launch wrote (P-2): row P-1 is being written by this launch and
        // is taken from the hand-off -- it must not even be touched here (a line fetched before its producer's store could be
        // served stale to the next launch).
        // SET 0: kv / vv.  SET 1 (ATTN_V2): wv[0..7] / wv[16..23], idle between the wqkv dots and the w1 request.  `split`: the key range.
        auto issue_kv_set = [&](auto set_tag, int u, int kv_layer, int split, auto part_tag) {  // u, split: runtime, block-uniform
            constexpr int SET = decltype(set_tag)::value;
            constexpr int PART = decltype(part_tag)::value;   // 0: K and V, 1: K only, 2: V only
            u32x4* const kd = SET == 0 ? kv : wv;
            u32x4* const vd = SET == 0 ? vv : wv + 16;
            // unused slots are zero (a variant that always requested every slot -- no merge of a constant and a load result -- measured slower,
            // profiles/r04x)
            if (PART != 2) {
#pragma unroll
                for (int q = 0; q < ATTN_U; ++q) kd[q] = u32x4{0u, 0u, 0u, 0u};
            }
            if (PART != 1) {
#pragma unroll
                for (int q = 0; q < ATTN_U; ++q) vd[q] = u32x4{0u, 0u, 0u, 0u};
            }
            const int Pu = GEO(0, u), csu = GEO(1, u), nactu = GEO(2, u), kwu = GEO(3, u);
            const int go = lane_o >> 4, dlo = lane_o & 15, s0o = split * csu + wave_o * kwu;
            const size_t layer_off = (size_t)kv_layer * a.kv_layer_elems;
            const bf16_t* kc = a.kcache + layer_off;
            const bf16_t* vc = a.vcache + layer_off;
            const size_t head_base = ((size_t)(2 * u + at_r) * MG_HEADS + at_h) * (size_t)a.cache_len * 128 + dlo * 8;
            const int last_old = max(Pu - 2, 0);
            if (Pu > 1 && split < nactu) {
                if (PART != 2)
#pragma unroll
                for (int q = 0; q < ATTN_U; ++q)
                    if (q * 4 < kwu) kd[q] = load_stream16(kc + head_base + (size_t)min(s0o + q * 4 + go, last_old) * 128);
                if (PART != 1)
#pragma unroll
                for (int q = 0; q < ATTN_U; ++q)
                    if (q * 4 < kwu) vd[q] = load_stream16(vc + head_base + (size_t)min(s0o + q * 4 + go, last_old) * 128);
            }
        };
        auto issue_kv = [&](int u, int kv_layer) { issue_kv_set(std::integral_constant<int, 0>{}, u, kv_layer, at_s, std::integral_constant<int, 0>{}); };

        // ================= P0: RMSNorm + wqkv (+ KV-cache write)        fast_model.py:179,209,218 ====================
        __syncthreads();
        MG_STAMP(l, 0);
        MG_TRIES_DECL();
        if (l == 0) {
            if (step == 0) {
#pragma unroll
                for (int u = 0; u < NU; ++u)
                    *(u32x4*)(xa[u] + tid * 8) = load16(KC(step)->x_in + (size_t)u * 2 * MG_DIM + tid * 8);  // written by the sampler in the previous launch
            }  // later steps: built at the end of the previous step from the control word
        } else MG_TRIES(0, mega_poll_rowpairs<NU>(g_x, tag0 - 8u + 4u, xa, abort_flag_of, &s_abort, tid_o));
        __syncthreads();
        MG_ABORT_RETURN();
        MG_STAMP(l, 1);
        // The K rows of the first attention slice are requested here, before the norm and the wqkv dots: they land while those run.  (All of
        // the slice requested after the q/k/v publish sits in front of the q/k/v poll: 831 vs 794 us/step.)  The V rows follow behind the
        // q/k/v publish, in front of the q/k/v hand-off: the K/V burst of all 256 workgroups (8 MB at 1024 positions) drains at HBM speed and
        // the memory pipeline takes requests only as fast as it drains them -- with both halves here every wave spent ~2.4 us in front of its
        // norm issuing them (profiles/r04g, r04x).
#pragma unroll
        for (int q = 0; q < ATTN_U; ++q) kv[q] = u32x4{0u, 0u, 0u, 0u};
#pragma unroll
        for (int q = 0; q < ATTN_U; ++q) vv[q] = u32x4{0u, 0u, 0u, 0u};
        if (first_att >= 0) issue_kv_set(std::integral_constant<int, 0>{}, first_att, l, at_s, std::integral_constant<int, 1>{});
        __builtin_amdgcn_sched_barrier(0);
        MG_STAMP(l, 11);  // K/V requested
        mega_norm<NU>(xa, nw_attn, red, a.eps, tid_p);
        MG_STAMP(l, 12);  // norm done
#pragma unroll (I4 ? 1 : MEGA_NU_UNROLL_P0)
        for (int u = 0; u < NU; ++u) {
            float acc[3][2];
            if (I4) mega_dots_i4<1, 3, MG_DIM>(cq, cz, xa[u], MG_DIM, lane_p, acc);
            else if (I8) mega_dots_i8<2, 3, MG_DIM>(cq, xa[u], MG_DIM, lane_p, acc);
            else mega_dots<4, 3, MG_DIM>(wv, xa[u], lane_p, acc);
            __builtin_amdgcn_sched_barrier(0);
            if (u == 0) MG_STAMP(l, 13);  // wqkv dots done
            float v0 = 0.f, v1 = 0.f;
            float sums0[6] = {acc[0][0], acc[0][1], acc[1][0], acc[1][1], acc[2][0], acc[2][1]};
            mega_wave_sums<6>(sums0);
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                float t0 = sums0[2 * i], t1 = sums0[2 * i + 1];
                if (I8) { t0 = i8_out(t0, sc_a[i]); t1 = i8_out(t1, sc_a[i]); }
                if (lane_p == i) { v0 = t0; v1 = t1; }
            }
            if (lane_p < 3) {
                const int n = 24 * b + 3 * wave_p + lane_p;
                const bf16_t y0 = f2bf(v0), y1 = f2bf(v1);
                gran_store(g_qkv + (size_t)u * 3 * MG_DIM + n, (uint32_t)y0 | ((uint32_t)y1 << 16), tag0 + 0u);
                if (MG_DBG_ON() && u == 0 && n < MG_DIM) { bf16_t* d = a.dbg + ((size_t)l * 5 + 0) * 2 * MG_INTER; d[n] = y0; d[MG_INTER + n] = y1; }
                if (n >= MG_DIM) {  // K / V rows also go to the cache for the FUTURE steps (this step reads them from the granules)
                    const int which = (n - MG_DIM) / MG_DIM, hd = (n - MG_DIM) % MG_DIM;
                    const size_t cache_len = (size_t)a.cache_len;
                    bf16_t* cache = (which ? a.vcache : a.kcache) + (size_t)l * a.kv_layer_elems + (size_t)2 * u * MG_HEADS * cache_len * 128;
                    const size_t off = ((size_t)(hd / 128) * cache_len + (GEO(0, u) - 1)) * 128 + hd % 128;
                    // agent-scope (write-through) stores: with several steps per launch the row is read by other CUs -- other XCDs,
                    // other L2s -- in the next step, with no kernel boundary in between
                    __hip_atomic_store(cache + off, y0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(cache + (size_t)MG_HEADS * cache_len * 128 + off, y1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (first_att >= 0) {
            issue_kv_set(std::integral_constant<int, 0>{}, first_att, l, at_s, std::integral_constant<int, 2>{});
            __builtin_amdgcn_sched_barrier(0);
        }
        MG_STAMP(l, 2);
        // next: wo row 8 b + wave.  Publish-before-prefetch (profiles/r01g_handoff/NOTES.md): a wave that still has something to
        // publish does so BEFORE it queues its next weight loads -- the memory pipeline accepts loads only as fast as it
        // drains them, so a publish behind 100-200 KB of requests reaches its consumers microseconds late.
        // wo goes to wv[12..15]; the w1 rows of P3 ride along in wv[0..11] (the w3 rows follow after P2's dots into wv[12..23]):
        // w1|w3 is 45 % of a layer's bytes, asked for in one go it made P2 -> P3 the one stream-bound hand-off (8 us).
        const int rb13 = 22 * b + (wave_o < 6 ? 3 * wave_o : 18 + 2 * (wave_o - 6));
        auto request_wo = [&]() {
            if (I4) mega_issue_row_i4<1>(cq + 6, cz + 6, L.q[1], L.sz[1], 8 * b + wave_o, MG_DIM, lane_o);
            else if (I8) { mega_issue_row_i8<2>(cq + 12, L.q8[1], 8 * b + wave_o, MG_DIM, lane_o); sc_c = bf2f(load_global_u16(L.s8[1] + (8 * b + wave_o))); }
            else mega_issue_row<4>(wv + 12, L.wo, 8 * b + wave_o, MG_DIM, lane_o, MG_ABLATE);
        };
        auto request_w3 = [&]() {  // 22 rows per block: waves 0..5 take 3, waves 6, 7 take 2 (third slot clamped, masked)
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                if (I4) mega_issue_row_i4<1>(cq + 3 + i, cz + 3 + i, L.q[3], L.sz[3], min(rb13 + i, MG_INTER - 1), MG_DIM, lane_o);
                else if (I8) { mega_issue_row_i8<2>(cq + 6 + 2 * i, L.q8[3], min(rb13 + i, MG_INTER - 1), MG_DIM, lane_o); sc_b[i] = bf2f(load_global_u16(L.s8[3] + (min(rb13 + i, MG_INTER - 1)))); }
                else mega_issue_row<4>(wv + 12 + 4 * i, L.w3, min(rb13 + i, MG_INTER - 1), MG_DIM, lane_o, MG_ABLATE);
            }
        };
        auto request_w1 = [&]() {
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                if (I4) mega_issue_row_i4<1>(cq + i, cz + i, L.q[2], L.sz[2], min(rb13 + i, MG_INTER - 1), MG_DIM, lane_o);
                else if (I8) { mega_issue_row_i8<2>(cq + 2 * i, L.q8[2], min(rb13 + i, MG_INTER - 1), MG_DIM, lane_o); sc_a[i] = bf2f(load_global_u16(L.s8[2] + (min(rb13 + i, MG_INTER - 1)))); }
                else if (EARLY_W1) {
                    const bf16_t* row = L.w1 + (size_t)min(rb13 + i, MG_INTER - 1) * MG_DIM;
                    if (!w1_half_issued) {  // a workgroup that does not attend in this step (its key range is empty)
                        mega_row_to_lds(lds_addr(w1_lds + ((wave_o * 6 + i * 2 + 0) * 64)), row + lane_o * 8, MG_ABLATE);
                        mega_row_to_lds(lds_addr(w1_lds + ((wave_o * 6 + i * 2 + 1) * 64)), row + 512 + lane_o * 8, MG_ABLATE);
                    }
                    wv[4 * i + 2] = load_stream16(row + 2 * 512 + lane_o * 8);
                    wv[4 * i + 3] = load_stream16(row + 3 * 512 + lane_o * 8);
                }
                else mega_issue_row<4>(wv + 4 * i, L.w1, min(rb13 + i, MG_INTER - 1), MG_DIM, lane_o, MG_ABLATE);
            }
            w1_half_issued = false;  // next layer
            nw_ffn = load16(L.ffn_norm + (tid_o & 255) * 8);
        };

        // ================= P1: attention over keys [0, P)               fast_model.py:218-222 =========================
        // Publish-before-prefetch holds for the WORKGROUP here, not just per wave: wave 0 still has the block's partial (and, in
        // the workgroup of split 0, the combined output) to publish, and a store queued behind 112 KB of the other waves' bulk
        // requests in this CU's memory pipeline reaches the combiner -- on everybody's critical path -- microseconds late.  So the
        // other waves of an attending workgroup hold their requests (the next utterance's K/V, then wo | w1) until wave 0 is through.
        if (!any_att) { request_wo(); request_w1(); }
        if (any_att) {
            __syncthreads();
            // every wave of the block has published its q/k/v rows: the wo row (8 MB chip-wide, ~1 us of stream) is requested HERE, under
            // the q/k/v hand-off, which costs its ~1.7 us of latency whatever is queued behind it; w1 follows after the attention
            request_wo();
            {   // q, k_new, v_new of head at_h of every utterance this block attends for: 3 x 128 row-paired granules each
                const int sec = tid_o >> 7, c = tid_o & 127;
                const gran_t* gq = g_qkv;
                if constexpr (ATTN_V2)
                    MG_TRIES(1, wave_poll<1>(tag0 + 0u, abort_flag_of, &s_abort,
                                 [&](int, u64& v) {
                                     if (tid_o >= 384) return false;
                                     v = gran_load(gq + (size_t)at_u * 3 * MG_DIM + sec * MG_DIM + at_h * 128 + c);
                                     return true;
                                 },
                                 [&](int, uint32_t p) { qkv_s[0][(sec * 2 + 0) * 128 + c] = (bf16_t)p; qkv_s[0][(sec * 2 + 1) * 128 + c] = (bf16_t)(p >> 16); }));
                else
                MG_TRIES(1, wave_poll<NU>(tag0 + 0u, abort_flag_of, &s_abort,
                              [&](int k, u64& v) {
                                  if (tid_o >= 384 || !(at_s < GEO(2, k))) return false;
                                  v = gran_load(gq + (size_t)k * 3 * MG_DIM + sec * MG_DIM + at_h * 128 + c);
                                  return true;
                              },
                              [&](int k, uint32_t p) { qkv_s[k][(sec * 2 + 0) * 128 + c] = (bf16_t)p; qkv_s[k][(sec * 2 + 1) * 128 + c] = (bf16_t)(p >> 16); }));
            }
            __syncthreads();
            MG_ABORT_RETURN();
            MG_STAMP(l, 3);
            if constexpr (EARLY_W1) {  // half of w1 on its way while the partial is computed, merged and published
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        mega_row_to_lds(lds_addr(w1_lds + ((wave_o * 6 + i * 2 + j) * 64)),
                                        L.w1 + (size_t)min(rb13 + i, MG_INTER - 1) * MG_DIM + j * 512 + lane_o * 8, MG_ABLATE);
                w1_half_issued = true;
            }
            if constexpr (ATTN_V2) {
                // ---- this workgroup's key ranges at_s .. at_s + SPB - 1 of (at_u, at_r, at_h), two register sets in flight ----
                // The second range goes out HERE, behind the q / k / v poll: every poll ends in `s_waitcnt vmcnt(0)`, and loads retire in
                // order -- requested in front of the poll (r04g: "q/k/v ready" 4.7 us) the poll waited for the whole second range.
                if constexpr (ATTN_DB) {
                    if (at_s + 1 < GEO(2, at_u)) issue_kv_set(std::integral_constant<int, 1>{}, at_u, l, at_s + 1, std::integral_constant<int, 0>{});
                    __builtin_amdgcn_sched_barrier(0);
                }
                const int u = at_u;
                // opaque ids: the publish / poll addresses below are functions of the lane -- hoisted out of the layer loop they are spilled, and a
                // scratch reload in front of a publish waits for every K/V row in flight
                const int g = lane_o >> 4, dl = lane_o & 15, at_w = wave_o, lane = lane_o;
                const int Pu = GEO(0, u), csu = GEO(1, u), nactu = GEO(2, u), kwu = GEO(3, u);
                const int n_my = min(SPB, nactu - at_s);   // >= 1: this workgroup attends
                const u32x4 qv = *(const u32x4*)(qkv_s[0] + (0 * 2 + at_r) * 128 + dl * 8);
                auto& kvB = *reinterpret_cast<u32x4(*)[ATTN_U]>(&wv[0]);   // second register set (ATTN_DB): see issue_kv_set
                auto& vvB = *reinterpret_cast<u32x4(*)[ATTN_U]>(&wv[ATTN_DB ? 16 : 0]);
#pragma unroll
                for (int j = 0; j < SPB; ++j) {
                    if (j < n_my) {   // block-uniform
                        const int split = at_s + j;
                        const int s0 = split * csu + at_w * kwu, s1 = min(min(Pu, split * csu + csu), s0 + kwu);
                        u32x4 (&kx)[ATTN_U] = (ATTN_DB && (j & 1)) ? kvB : kv;
                        u32x4 (&vx)[ATTN_U] = (ATTN_DB && (j & 1)) ? vvB : vv;
#pragma unroll
                        for (int q = 0; q < ATTN_U; ++q) {
                            if (s0 + q * 4 + g == Pu - 1) {  // the key produced by this step: not in the cache yet for this launch
                                kx[q] = *(const u32x4*)(qkv_s[0] + (1 * 2 + at_r) * 128 + dl * 8);
                                vx[q] = *(const u32x4*)(qkv_s[0] + (2 * 2 + at_r) * 128 + dl * 8);
                            }
                        }
                        const AttnAcc w = attn_wave_partial<ATTN_PL>(kx, vx, qv, s0, s1, g);
                        if (j == 0) MG_STAMP(l, 14);  // first wave partial done
                        // this set's registers are free: the range after next is requested into them while the other set is scored
                        if constexpr (ATTN_DB) {
                            if (j + 2 < SPB && j + 2 < n_my) {
                                if (j & 1) issue_kv_set(std::integral_constant<int, 1>{}, u, l, split + 2, std::integral_constant<int, 0>{});
                                else issue_kv_set(std::integral_constant<int, 0>{}, u, l, split + 2, std::integral_constant<int, 0>{});
                            }
                        } else if (j + 1 < SPB && j + 1 < n_my) issue_kv_set(std::integral_constant<int, 0>{}, u, l, split + 1, std::integral_constant<int, 0>{});
                        if (g == 0) {
                            float (*smj)[16][12] = sm + j * ATTN_WAVES;
                            smj[at_w][dl][0] = w.m;
                            smj[at_w][dl][1] = w.l;
#pragma unroll
                            for (int d = 0; d < 8; ++d) smj[at_w][dl][2 + d] = w.o[d];
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                __syncthreads();
                __builtin_amdgcn_sched_barrier(0);
                // wave j merges range j and publishes its partial; with a single range (<= 256 keys) the merged result is the output
                u64* slots = (u64*)a.partial + ((size_t)(2 * u + at_r) * MG_HEADS + at_h) * ATTN_NSPLIT * ATTN_TPART_STRIDE;
                AttnAcc bb;
                if (at_w < n_my) {
                    bb = attn_merge_waves<ATTN_PL>(sm + at_w * ATTN_WAVES, g, dl);
                    if (nactu > 1) attn_publish_tagged(bb, slots + (size_t)(at_s + at_w) * ATTN_TPART_STRIDE, tag0 + 5u, lane);
                }
                if (at_w == 0 && at_s == 0) {   // the workgroup of the first ranges combines (utterance, row, head) and publishes its output
                    bool last = true;
                    if (nactu > 1 && !attn_combine_tagged<ATTN_PL>(bb, slots, tag0 + 5u, nactu, lane, MG_SPIN_LIMIT)) {
                        uint32_t* af = abort_flag_of();
                        mega_report_timeout(af, tag0 + 5u);
                        __hip_atomic_store(af, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        *(volatile int*)&s_abort = 1;
                        last = false;
                    }
                    if (last && g == 0) {  // 8 dims per lane = 4 column-paired granules
                        const u32x4 y = attn_normalise(bb);
                        gran_t* gp = g_attn + (size_t)u * MG_DIM + at_r * (MG_DIM / 2) + (at_h * 128 + dl * 8) / 2;
                        gran_store(gp + 0, y.x, tag0 + 1u);
                        gran_store(gp + 1, y.y, tag0 + 1u);
                        gran_store(gp + 2, y.z, tag0 + 1u);
                        gran_store(gp + 3, y.w, tag0 + 1u);
                        if (MG_DBG_ON() && u == 0) *(u32x4*)(a.dbg + ((size_t)l * 5 + 1) * 2 * MG_INTER + at_r * MG_INTER + at_h * 128 + dl * 8) = y;
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            } else
            // One K/V register set.  After an utterance's partial has been merged and published (workgroup-wide
            // publish-before-prefetch) EVERY wave requests its slice of the next attending utterance -- wave 0 too, before the
            // combine polls it may owe.  The split partials of utterance u are combined by the workgroup of split u % nact (not
            // all by split 0, which would then wait for a combine once per utterance).
#pragma unroll 1
            for (int u = 0; u < NU; ++u) {
                const int Pu = GEO(0, u), csu = GEO(1, u), nactu = GEO(2, u), kwu = GEO(3, u);
                if (!(at_s < nactu)) continue;  // block-uniform
                int next_att = -1;
#pragma unroll 1
                for (int v = NU - 1; v > u; --v)
                    if (at_s < GEO(2, v)) next_att = v;
                const int at_s0 = at_s * csu + at_w * kwu, at_s1 = min(min(Pu, at_s * csu + csu), at_s0 + kwu);
                const u32x4 qv = *(const u32x4*)(qkv_s[u] + (0 * 2 + at_r) * 128 + dl * 8);
#pragma unroll
                for (int q = 0; q < ATTN_U; ++q) {
                    if (at_s0 + q * 4 + g == Pu - 1) {  // the key produced by this step: not in the cache yet for this launch
                        kv[q] = *(const u32x4*)(qkv_s[u] + (1 * 2 + at_r) * 128 + dl * 8);
                        vv[q] = *(const u32x4*)(qkv_s[u] + (2 * 2 + at_r) * 128 + dl * 8);
                    }
                }
                const AttnAcc w = attn_wave_partial<ATTN_PL>(kv, vv, qv, at_s0, at_s1, g);
                if (u == 0) MG_STAMP(l, 14);  // wave partial done
                if (g == 0) {
                    sm[at_w][dl][0] = w.m;
                    sm[at_w][dl][1] = w.l;
#pragma unroll
                    for (int d = 0; d < 8; ++d) sm[at_w][dl][2 + d] = w.o[d];
                }
                __syncthreads();
                __builtin_amdgcn_sched_barrier(0);
                AttnAcc bb;
                u64* slots = (u64*)a.partial + ((size_t)(2 * u + at_r) * MG_HEADS + at_h) * ATTN_NSPLIT * ATTN_TPART_STRIDE;
                if (at_w == 0) {
                    bb = attn_merge_waves<ATTN_PL>(sm, g, dl);
                    // split partials as tagged granules: every split publishes, one workgroup polls all of them and
                    // combines (attn_core.h) -- one hand-off instead of store + vmcnt(0) + ticket round trip + load round trip
                    if (nactu > 1) attn_publish_tagged(bb, slots + (size_t)at_s * ATTN_TPART_STRIDE, tag0 + 5u, lane);
                }
                if (next_att >= 0) {
                    __syncthreads();  // the partial has left, `sm` has been consumed
                    issue_kv(next_att, l);
                }
                if (at_w == 0) {
                    bool last = true;
                    if (nactu > 1) {
                        last = at_s == (NU == 1 ? 0 : u % nactu);
                        if (last && !attn_combine_tagged<ATTN_PL>(bb, slots, tag0 + 5u, nactu, lane, MG_SPIN_LIMIT)) {
                            uint32_t* af = abort_flag_of();
                            mega_report_timeout(af, tag0 + 5u);
                            __hip_atomic_store(af, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            *(volatile int*)&s_abort = 1;
                            last = false;
                        }
                    }
                    if (last && g == 0) {  // 8 dims per lane = 4 column-paired granules
                        const u32x4 y = attn_normalise(bb);
                        gran_t* gp = g_attn + (size_t)u * MG_DIM + at_r * (MG_DIM / 2) + (at_h * 128 + dl * 8) / 2;
                        gran_store(gp + 0, y.x, tag0 + 1u);
                        gran_store(gp + 1, y.y, tag0 + 1u);
                        gran_store(gp + 2, y.z, tag0 + 1u);
                        gran_store(gp + 3, y.w, tag0 + 1u);
                        if (MG_DBG_ON() && u == 0) *(u32x4*)(a.dbg + ((size_t)l * 5 + 1) * 2 * MG_INTER + at_r * MG_INTER + at_h * 128 + dl * 8) = y;
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            MG_STAMP(l, 15);  // wave 0: attention output (or partial) published
            __syncthreads();
            request_w1();
        }
        __builtin_amdgcn_sched_barrier(0);

        // ================= P2: h = x + wo(attn)                         fast_model.py:224-226,179 =====================
        __syncthreads();
        MG_STAMP(l, 4);
        {
            // granule i of utterance u = elements 2i, 2i+1 of its [2][2048] attention output
            MG_TRIES(2, wave_poll16<2 * NU>(g_attn, tag0 + 1u, abort_flag_of, &s_abort, [&](int k) { return (k >> 1) * (MG_DIM / 2) * 2 / 2 + tid_o + (k & 1) * MG_THREADS; },
                                            [&](int n, uint32_t p0, uint32_t p1) { ((uint2*)xs[n / (MG_DIM / 2)])[n % (MG_DIM / 2)] = make_uint2(p0, p1); }));
            __syncthreads();
            MG_ABORT_RETURN();
            MG_STAMP(l, 5);
#pragma unroll 1
            for (int u = 0; u < NU; ++u) {
                float acc[1][2];
                if (I4) mega_dots_i4<1, 1, MG_DIM>(cq + 6, cz + 6, xs[u], MG_DIM, lane_p, acc);
                else if (I8) mega_dots_i8<2, 1, MG_DIM>(cq + 12, xs[u], MG_DIM, lane_p, acc);
                else mega_dots<4, 1, MG_DIM>(wv + 12, xs[u], lane_p, acc);
                __builtin_amdgcn_sched_barrier(0);
                if (u == 0) MG_STAMP(l, 16);  // wo dots done
                float sums2[2] = {acc[0][0], acc[0][1]};
                mega_wave_sums<2>(sums2);
                float t0 = sums2[0], t1 = sums2[1];
                if (I8) { t0 = i8_out(t0, sc_c); t1 = i8_out(t1, sc_c); }
                const uint32_t xr = res_s[0][u][wave_p];
                const bf16_t y0 = epi_resid((bf16_t)xr, t0), y1 = epi_resid((bf16_t)(xr >> 16), t1);
                const uint32_t hr = (uint32_t)y0 | ((uint32_t)y1 << 16);
                if (lane_p == 0) res_s[1][u][wave_p] = hr;
                // test hook (mv_tune "mega_fault"): one workgroup withholds this publish -> every peer's wait has to time out, raise the
                // abort flag and leave the kernel; mv_run then reports the error (tests/test_engine_gpu.py)
                bool withhold = false;
                if (l == 1) { const int fault = a.fault; withhold = fault != 0 && b == ((fault - 1) & 255); }
                if (lane_p == 0 && !withhold) {
                    const int n = 8 * b + wave_p;
                    gran_store(g_h + (size_t)u * MG_DIM + n, hr, tag0 + 2u);
                    if (MG_DBG_ON() && u == 0) { bf16_t* d = a.dbg + ((size_t)l * 5 + 2) * 2 * MG_INTER; d[n] = y0; d[MG_INTER + n] = y1; }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            MG_STAMP(l, 17);  // h published
            // next: the w3 rows (w1 is already in flight)
            request_w3();
            __builtin_amdgcn_sched_barrier(0);
        }

        // ================= P3: act = silu(w1 n(h)) * w3 n(h)            fast_model.py:180,237 =========================
        __syncthreads();
        MG_STAMP(l, 6);
        {
            MG_TRIES(3, mega_poll_rowpairs<NU>(g_h, tag0 + 2u, xa, abort_flag_of, &s_abort, tid_o));
            __syncthreads();
            MG_ABORT_RETURN();
            MG_STAMP(l, 7);
            auto request_w2_early = [&]() {  // chunks 0..EW2-1 of the w2 row on their way while the norm, the w1|w3 dots and the act publish run
#pragma unroll
                for (int j = 0; j < EW2; ++j)
                    mega_row_to_lds(lds_addr(w2_lds + ((wave_o * EW2 + j) * 64)), L.w2 + (size_t)(8 * b + wave_o) * MG_INTER + j * 512 + lane_o * 8, MG_ABLATE);
            };
            if constexpr (EARLY_W2) request_w2_early();
            mega_norm<NU>(xa, nw_ffn, red, a.eps, tid_p);
            MG_STAMP(l, 18);  // norm done
            const int nrows = wave_p < 6 ? 3 : 2;
            const int rb = 22 * b + (wave_p < 6 ? 3 * wave_p : 18 + 2 * (wave_p - 6));
#pragma unroll (I4 ? 1 : MEGA_NU_UNROLL_P3)
            for (int u = 0; u < NU; ++u) {
                float acc1[3][2], acc3[3][2];
                if (I4) {
                    // w1 rows (slots 0..2) and w3 rows (3..5) in ONE pass: the activation blocks and their column sums are read / formed once
                    // for all six rows; every row's own arithmetic is what two passes of three rows would do (same bits as gemv_i4.hip)
                    float acc6[6][2];
                    mega_dots_i4<1, 6, MG_DIM>(cq, cz, xa[u], MG_DIM, lane_p, acc6);
#pragma unroll
                    for (int i = 0; i < 3; ++i) { acc1[i][0] = acc6[i][0]; acc1[i][1] = acc6[i][1]; acc3[i][0] = acc6[3 + i][0]; acc3[i][1] = acc6[3 + i][1]; }
                } else if (I8) {
                    mega_dots_i8<2, 3, MG_DIM>(cq, xa[u], MG_DIM, lane_p, acc1);
                    mega_dots_i8<2, 3, MG_DIM>(cq + 6, xa[u], MG_DIM, lane_p, acc3);
                } else {
                    if constexpr (EARLY_W1) {  // same order as mega_dots: chunk j ascending, row r, weight row i
#pragma unroll
                        for (int i = 0; i < 3; ++i) acc1[i][0] = acc1[i][1] = 0.f;
#pragma unroll
                        for (int j = 0; j < 4; ++j)
#pragma unroll
                            for (int r = 0; r < 2; ++r) {
                                const u32x4 xv = *(const u32x4*)(xa[u] + r * MG_DIM + j * 512 + lane_p * 8);
#pragma unroll
                                for (int i = 0; i < 3; ++i) {
                                    const u32x4 w = j < 2 ? w1_lds[(wave_p * 6 + i * 2 + j) * 64 + lane_p] : wv[4 * i + j];
                                    acc1[i][r] = dot8_bf16(w, xv, acc1[i][r]);
                                }
                            }
                    } else
                    mega_dots<4, 3, MG_DIM>(wv, xa[u], lane_p, acc1);
                    mega_dots<4, 3, MG_DIM>(wv + 12, xa[u], lane_p, acc3);
                }
                __builtin_amdgcn_sched_barrier(0);
                if (u == 0) MG_STAMP(l, 19);  // w1|w3 dots done
                float g0 = 0.f, g1 = 0.f, u0 = 0.f, u1 = 0.f;
                float sums3[12] = {acc1[0][0], acc1[0][1], acc3[0][0], acc3[0][1], acc1[1][0], acc1[1][1], acc3[1][0], acc3[1][1],
                                   acc1[2][0], acc1[2][1], acc3[2][0], acc3[2][1]};
                mega_wave_sums<12>(sums3);
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    float a0 = sums3[4 * i], a1 = sums3[4 * i + 1], c0 = sums3[4 * i + 2], c1 = sums3[4 * i + 3];
                    if (I8) { a0 = i8_out(a0, sc_a[i]); a1 = i8_out(a1, sc_a[i]); c0 = i8_out(c0, sc_b[i]); c1 = i8_out(c1, sc_b[i]); }
                    if (lane_p == i) { g0 = a0; g1 = a1; u0 = c0; u1 = c1; }
                }
                if (lane_p < nrows) {
                    const bf16_t y0 = epi_swiglu(g0, u0), y1 = epi_swiglu(g1, u1);
                    gran_store(g_act + (size_t)u * MG_INTER + rb + lane_p, (uint32_t)y0 | ((uint32_t)y1 << 16), tag0 + 3u);
                    if (MG_DBG_ON() && u == 0) { bf16_t* d = a.dbg + ((size_t)l * 5 + 3) * 2 * MG_INTER; d[rb + lane_p] = y0; d[MG_INTER + rb + lane_p] = y1; }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            MG_STAMP(l, 20);  // act published
            // next: w2 row 8 b + wave (K = 5632: 11 chunks)
            if (I4) mega_issue_row_i4<3>(cq + 6, cz + 6, L.q[4], L.sz[4], 8 * b + wave_o, MG_INTER, lane_o);
            else if (I8) { mega_issue_row_i8<6>(cq, L.q8[4], 8 * b + wave_o, MG_INTER, lane_o); sc_c = bf2f(load_global_u16(L.s8[4] + (8 * b + wave_o))); }
            else if (EARLY_W2) {
#pragma unroll
                for (int j = EW2; j < 11; ++j) wv[j] = load_stream16(L.w2 + (size_t)(8 * b + wave_o) * MG_INTER + j * 512 + lane_o * 8);
            }
            else mega_issue_row<11>(wv, L.w2, 8 * b + wave_o, MG_INTER, lane_o, MG_ABLATE);
            __builtin_amdgcn_sched_barrier(0);
        }

        // ================= P4: x = h + w2(act)                          fast_model.py:247,180 =========================
        __syncthreads();
        MG_STAMP(l, 8);
        {
            // 16-byte polls carrying two granules (each half validated by its own tag): 5632 / 2 = 2816 pairs per utterance, 6 per
            // thread (was eleven 8-byte loads per thread); at most two utterances per poll loop (registers)
            constexpr int UPP = NU < 2 ? NU : 2;
            constexpr int PPT = (MG_INTER / 2 + MG_THREADS - 1) / MG_THREADS;  // pairs per thread and utterance
#pragma unroll
            for (int u0 = 0; u0 < NU; u0 += UPP) {
                MG_TRIES(4, wave_poll16<PPT * UPP>(g_act + (size_t)u0 * MG_INTER, tag0 + 3u, abort_flag_of, &s_abort,
                                       [&](int k) {
                                           const int n = tid_o + (k % PPT) * MG_THREADS;
                                           return n < MG_INTER / 2 ? (k / PPT) * (MG_INTER / 2) + n : -1;
                                       },
                                       [&](int n, uint32_t p0, uint32_t p1) {
                                           bf16_t* row = xs[u0 + n / (MG_INTER / 2)];
                                           const int c = n % (MG_INTER / 2);
                                           *(uint32_t*)(row + 2 * c) = (p0 & 0xFFFFu) | (p1 << 16);
                                           *(uint32_t*)(row + MG_INTER + 2 * c) = (p0 >> 16) | (p1 & 0xFFFF0000u);
                                       }));
            }
            MG_TRIES_STORE(l);
            __syncthreads();
            MG_ABORT_RETURN();
            MG_STAMP(l, 9);
#pragma unroll 1
            for (int u = 0; u < NU; ++u) {
                float acc[1][2];
                if (I4) mega_dots_i4<3, 1, MG_INTER>(cq + 6, cz + 6, xs[u], MG_INTER, lane_p, acc);
                else if (I8) mega_dots_i8<6, 1, MG_INTER>(cq, xs[u], MG_INTER, lane_p, acc);
                else if (EARLY_W2) {  // same order as mega_dots: chunk j ascending, row r
                    acc[0][0] = acc[0][1] = 0.f;
#pragma unroll
                    for (int j = 0; j < 11; ++j) {
                        const u32x4 w = j < EW2 ? w2_lds[(wave_p * EW2 + j) * 64 + lane_p] : wv[j];
#pragma unroll
                        for (int r = 0; r < 2; ++r) acc[0][r] = dot8_bf16(w, *(const u32x4*)(xs[u] + r * MG_INTER + j * 512 + lane_p * 8), acc[0][r]);
                    }
                }
                else mega_dots<11, 1, MG_INTER>(wv, xs[u], lane_p, acc);
                __builtin_amdgcn_sched_barrier(0);
                if (u == 0) MG_STAMP(l, 21);  // w2 dots done
                float sums2[2] = {acc[0][0], acc[0][1]};
                mega_wave_sums<2>(sums2);
                float t0 = sums2[0], t1 = sums2[1];
                if (I8) { t0 = i8_out(t0, sc_c); t1 = i8_out(t1, sc_c); }
                const uint32_t hr = res_s[1][u][wave_p];
                const bf16_t y0 = epi_resid((bf16_t)hr, t0), y1 = epi_resid((bf16_t)(hr >> 16), t1);
                const uint32_t xr = (uint32_t)y0 | ((uint32_t)y1 << 16);
                if (lane_p == 0) {
                    const int n = 8 * b + wave_p;
                    res_s[0][u][wave_p] = xr;
                    gran_store(g_x + (size_t)u * MG_DIM + n, xr, tag0 + 4u);
                    if (u == 0) MG_STAMP(l, 10);
                    if (MG_DBG_ON() && u == 0) { bf16_t* d = a.dbg + ((size_t)l * 5 + 4) * 2 * MG_INTER; d[n] = y0; d[MG_INTER + n] = y1; }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            // next: the following layer's wqkv rows, or the LM-head rows 11 b + wave + 8 i
            if (l + 1 < n_layer) {
                const bf16_t* w = MG_LAYER(l + 1).wqkv;
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    if (I4) mega_issue_row_i4<1>(cq + i, cz + i, MG_LAYER(l + 1).q[0], MG_LAYER(l + 1).sz[0], 24 * b + 3 * wave_o + i, MG_DIM, lane_o);
                    else if (I8) { mega_issue_row_i8<2>(cq + 2 * i, MG_LAYER(l + 1).q8[0], 24 * b + 3 * wave_o + i, MG_DIM, lane_o); sc_a[i] = bf2f(load_global_u16(MG_LAYER(l + 1).s8[0] + (24 * b + 3 * wave_o + i))); }
                    else mega_issue_row<4>(wv + 4 * i, w, 24 * b + 3 * wave_o + i, MG_DIM, lane_o, MG_ABLATE);
                }
            } else {
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const mega_kargs_t A = KC(l);
                    const int hn = min(11 * b + wave_o + 8 * i, a.vocab - 1);
                    if (I8) { mega_issue_row_i8<2>(cq + 2 * i, A->head_q8, hn, MG_DIM, lane_o); sc_a[i] = bf2f(load_global_u16(A->head_s8 + hn)); }
                    else mega_issue_row<4>(wv + 4 * i, A->head, hn, MG_DIM, lane_o, MG_ABLATE);
                }
            }
            nw_attn = load16((l + 1 < n_layer ? MG_LAYER(l + 1).attn_norm : KC(l)->final_norm) + (tid_o & 255) * 8);
            __builtin_amdgcn_sched_barrier(0);
        }
    }

    // ================= LM head on the last position: logits = head(norm(x))     fast_model.py:161-162 =================
    const uint32_t tag_head = epoch + (uint32_t)n_layer * 8u + 1u;
    __syncthreads();
    mega_poll_rowpairs<NU>(g_x, epoch + (uint32_t)(n_layer - 1) * 8u + 1u + 4u, xa, [&]() { return KC(step)->abort_flag; }, &s_abort, tid);
    __syncthreads();
    MG_ABORT_RETURN();
    mega_norm<NU>(xa, nw_attn, red, a.eps, tid);
#pragma unroll 1
    for (int u = 0; u < NU; ++u) {
        float acc[2][2];
        if (I8) mega_dots_i8<2, 2, MG_DIM>(cq, xa[u], MG_DIM, lane, acc);
        else mega_dots<4, 2, MG_DIM>(wv, xa[u], lane, acc);
        float v0 = 0.f, v1 = 0.f;
        float sumsh[4] = {acc[0][0], acc[0][1], acc[1][0], acc[1][1]};
        mega_wave_sums<4>(sumsh);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            float t0 = sumsh[2 * i], t1 = sumsh[2 * i + 1];
            if (I8) { t0 = i8_out(t0, sc_a[i]); t1 = i8_out(t1, sc_a[i]); }
            if (lane == i) { v0 = t0; v1 = t1; }
        }
        if (lane < 2) {
            const int slot = wave + 8 * lane, n = 11 * b + slot;
            const bf16_t y0 = f2bf(v0), y1 = f2bf(v1);
            const mega_kargs_t A = KC(step);
            const int vocab = a.vocab;
            if (fold && slot < 11 && n < VP) gran_store(g_logits + (size_t)u * VP + n, (uint32_t)y0 | ((uint32_t)y1 << 16), tag_head);
            if (slot < 11 && n < vocab) {
                bf16_t* lo = A->logits;
                lo[(size_t)(2 * u) * vocab + n] = y0;
                lo[(size_t)(2 * u + 1) * vocab + n] = y1;
            }
        }
    }
    if constexpr (fold) {

    // ================= the token: sample() + the bookkeeping of decode_n_tokens + the next input rows ===================
    // fast_inference_utils.py:61-120,148-174, fast_model.py:152-157 -- sample_core.h, the arithmetic of sample_kernel.  Block u
    // draws for utterance u; everybody else has already queued layer 0's rows for the next step and waits for the control word.
    // (address arithmetic on opaque ids, as in the layers: nothing of the tail is hoisted out of the step loop into live registers)
    int lane_t = lane, wave_t = wave;
    asm volatile("" : "+v"(lane_t), "+v"(wave_t));
    const int tid_t = wave_t * 64 + lane_t;
    const bool last_step = step + 1 == n_steps;
    const uint32_t tag_next = (counter0 + (uint32_t)step + 1u) * 256u;  // phase code 0 of the next epoch: no layer phase uses it
    __builtin_amdgcn_sched_barrier(0);
    if (b < NU) {
        const int u = b;
        bf16_t* lg = xs[0];  // [2][VP]: conditioned | unconditioned logits of this utterance
        __syncthreads();     // every wave is through with the head's operand in xs
        const bool was_done = st_s.done != 0;
        if (!was_done) {
            wave_poll16<3>(g_logits + (size_t)u * VP, tag_head, [&]() { return KC(step)->abort_flag; }, &s_abort,
                           [&](int k) { const int n = tid_t + k * MG_THREADS; return n < VP / 2 ? n : -1; },
                           [&](int n, uint32_t p0, uint32_t p1) {
                               *(uint32_t*)(lg + 2 * n) = (p0 & 0xFFFFu) | (p1 << 16);
                               *(uint32_t*)(lg + VP + 2 * n) = (p0 >> 16) | (p1 & 0xFFFF0000u);
                           });
        }
        // this workgroup's layer-0 rows: in flight while the token is drawn, landed before the publishes
        if (!last_step) issue_layer0(lane_t, wave_t);
        __syncthreads();
        MG_ABORT_RETURN();
        mega_draw_and_publish<NU>(KC(step), g_logits + (size_t)NU * VP, u, lg, &st_s, &smp_sc, VP, tag_next, last_step, tid_t);
        if (tid_t == 0 && u == 0) KC(step)->step_counter_out[0] = counter0 + (uint32_t)step + 1u;  // the next launch's first epoch
    } else if (!last_step) {
        issue_layer0(lane_t, wave_t);
    }
    if (last_step) return;
    __builtin_amdgcn_sched_barrier(0);
    // Every workgroup: wait for the control words {token | position << 12 | done << 31} -- ONE wave polls, one granule per utterance,
    // instead of 2048 waves spinning on a 16 KB vector of input rows for the ~10 us of a draw -- then build the next input rows from
    // the embedding tables (three 16-byte loads per thread; the token's row is shared by all 256 workgroups, i.e. L2 / MALL hits).
    if (wave_t == 0) {
        const gran_t* gc = g_logits + (size_t)NU * VP;   // g_ctl
        wave_poll<1>(tag_next, [&]() { return KC(step)->abort_flag; }, &s_abort,
                     [&](int, u64& v) { if (lane_t >= NU) return false; v = gran_load(gc + lane_t); return true; },
                     [&](int, uint32_t p) { tok_s[lane_t] = (int)(p & 0xFFFu); pos_s[lane_t] = (int)((p >> 12) & 0x7FFFFu); done_s[lane_t] = (int)(p >> 31); });
    }
    __syncthreads();  // also: block u has read `lg`, xs may be overwritten
    MG_ABORT_RETURN();
    {
        bool all_done = true;
#pragma unroll
        for (int u = 0; u < NU; ++u) all_done = all_done && done_s[u] != 0;
        if (all_done) return;
    }
    for (int t = tid_t; t < 256 * NU; t += MG_THREADS) {  // fast_model.py:152-157; item t: columns 8 (t & 255) .. of both rows of utterance t >> 8
        const int u = t >> 8, c = (t & 255) * 8;
        const mega_kargs_t A = KC(step);
        const u32x4 te = load16(A->tok_emb + (size_t)tok_s[u] * MG_DIM + c), pe = load16(A->pos_emb + (size_t)pos_s[u] * MG_DIM + c);
        const u32x4 sp = load16(A->spk_proj + (size_t)u * MG_DIM + c);
        u32x4 r0, r1;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t tw = te[j], pw = pe[j], sw = sp[j];
            const float b0 = round_bf(bf2f((bf16_t)tw) + bf2f((bf16_t)pw)), b1 = round_bf(bf2f((bf16_t)(tw >> 16)) + bf2f((bf16_t)(pw >> 16)));
            r0[j] = (uint32_t)f2bf(b0 + bf2f((bf16_t)sw)) | ((uint32_t)f2bf(b1 + bf2f((bf16_t)(sw >> 16))) << 16);
            r1[j] = (uint32_t)f2bf(b0) | ((uint32_t)f2bf(b1) << 16);
        }
        *(u32x4*)(xa[u] + c) = r0;
        *(u32x4*)(xa[u] + MG_DIM + c) = r1;
    }
    __syncthreads();
    }  // if constexpr (fold)
    }  // steps
}


bool mega_supported(const mv_config& c, int rows, int n_cus) {
    return (rows == 2 || rows == 4 || rows == 8 || rows == 16) && c.dim == MG_DIM && c.intermediate == MG_INTER && c.n_head == MG_HEADS && c.vocab <= 256 * 11 && n_cus >= 256 &&
           c.block_size <= ATTN_NSPLIT * 256 &&
           c.n_layer <= 31;  // hand-off tag = step * 256 + layer * 8 + phase + 1: layer * 8 + 6 must stay below 256
}

// One utterance per engine (the rolled utterance loops of the 2 / 4-utterance instances spill inside the layer loop once the sampler
// tail is compiled in).
// int8 engines draw their tokens inside the launch like bf16 / int4 ones: 647 vs 655 us/step (profiles/r03w; in round 2, with the
// accumulator of the split-partial publish in scratch, the instance was the slower one and int8 kept sample_kernel).
bool mega_folds(const mv_config& c) { return c.max_utterances == 1; }

__global__ void bump_counter_kernel(uint32_t* c) { c[0] += 1u; }
hipError_t launch_bump_counter(hipStream_t s, uint32_t* c) {
    hipLaunchKernelGGL(bump_counter_kernel, dim3(1), dim3(1), 0, s, c);
    return hipGetLastError();
}


template <int WF, int NU, bool FOLD>
static hipError_t launch_mega_instance(hipStream_t s, const MegaArgs& a) {
    // > 64 KB of LDS: dynamic, raised once per instance and DEVICE (the attribute belongs to the device's copy of the function)
    static std::atomic<bool> raised_on[64];
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::atomic<bool>& raised = raised_on[dev & 63];
    const size_t lds = mega_lds_bytes(NU) + mega_early_lds_bytes(WF, NU);
    if (!raised) {
        hipError_t e = hipFuncSetAttribute((const void*)decode_mega_kernel<WF, NU, FOLD>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        raised = true;
    }
    hipLaunchKernelGGL((decode_mega_kernel<WF, NU, FOLD>), dim3(256), dim3(MG_THREADS), lds, s, a);
    return hipGetLastError();
}
template <int WF>
static hipError_t launch_mega_format(hipStream_t s, const MegaArgs& a) {
    switch (a.n_utt) {
        case 1: return a.n_steps > 0 ? launch_mega_instance<WF, 1, true>(s, a) : launch_mega_instance<WF, 1, false>(s, a);
        case 2: return a.n_steps > 0 ? hipErrorInvalidValue : launch_mega_instance<WF, 2, false>(s, a);
        case 4: return a.n_steps > 0 ? hipErrorInvalidValue : launch_mega_instance<WF, 4, false>(s, a);
        default: return hipErrorInvalidValue;
    }
}
// Residency check in front of the first launch (instead of finding out by a time-out): the hand-offs need all 256 workgroups resident at once,
// i.e. >= 1 workgroup of this instance per CU by the occupancy query (registers, LDS) and >= 256 CUs.  What the query cannot see -- another
// process or stream holding CUs -- is still caught by the bounded waits (mv_run then redoes the launch with per-layer launches).
template <int WF, int NU, bool FOLD>
static bool mega_instance_fits(int n_cus) {
    const size_t lds = mega_lds_bytes(NU) + mega_early_lds_bytes(WF, NU);
    if (hipFuncSetAttribute((const void*)decode_mega_kernel<WF, NU, FOLD>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return false;
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)decode_mega_kernel<WF, NU, FOLD>, MG_THREADS, lds) != hipSuccess) return false;
    return per_cu >= 1 && (long)per_cu * n_cus >= 256;
}
template <int WF>
static bool mega_format_fits(int n_utt, int n_cus) {
    switch (n_utt) {
        case 1: return mega_instance_fits<WF, 1, false>(n_cus) && mega_instance_fits<WF, 1, true>(n_cus);   // the instances launch_mega_format launches
        case 2: return mega_instance_fits<WF, 2, false>(n_cus);
        case 4: return mega_instance_fits<WF, 4, false>(n_cus);
        default: return false;
    }
}
bool mega_fits(int weight_format, int n_utt, int n_cus) {
    if (weight_format == MV_W_INT4_G128) return mega_format_fits<MV_W_INT4_G128>(n_utt, n_cus);
    if (weight_format == MV_W_INT8) return mega_format_fits<MV_W_INT8>(n_utt, n_cus);
    return mega_format_fits<MV_W_BF16>(n_utt, n_cus);
}

hipError_t launch_decode_mega(hipStream_t s, const MegaArgs& a, int weight_format) {
    if (weight_format == MV_W_INT4_G128) return launch_mega_format<MV_W_INT4_G128>(s, a);
    if (weight_format == MV_W_INT8) return launch_mega_format<MV_W_INT8>(s, a);
    return launch_mega_format<MV_W_BF16>(s, a);
}
