// Experimental variant of the persistent decode-step kernel (mv_tune("mega", 2)): loader waves + an LDS weight ring.
// Bit-identical to decode_mega.hip and to the per-layer launches, slower than the former (profiles/r01h_loader_waves).
#include "mega_core.h"

// v3 of the same kernel: LOADER WAVES + an LDS weight ring.
//
// What limits v2 (profiles/r01g_handoff/NOTES.md): a wave's weight rows are requested only one phase ahead (registers), and the
// memory pipeline of a CU accepts loads only as fast as HBM drains them (~64 clk per 1 KB wave-load with every CU streaming): the
// requesting wave is held up for the stream time of its rows and its poll -- vector loads return in order -- completes only
// after they have LANDED.  Here 8 compute waves never touch the weights in global memory: 4 loader waves fetch them into
// their own registers a full phase pair ahead and copy them into a 128 KB LDS ring (1 KB "lines" = one 512-column chunk of
// one weight row) when the slots free up.  Loaders and compute waves run the SAME barrier sequence (a static software
// pipeline, no flags): requests are issued right after a barrier that follows a publish (the compute waves are about to
// wait for a hand-off), copies right after the barrier that frees the slots.
//
// Ring map (lines):  wqkv rows (24) -> [0,96), wo rows (8) -> [96,128); w1|w3 rows i=0,1 of every wave -> [0,128) (the first
// 96 as soon as wqkv is consumed, the last 32 when wo is); rows i=2 (waves 0..5) -> [0,48) once rows i=0,1 are consumed;
// w2 rows (8 x 11 lines): lines 0..79 -> [48,128), lines 80..87 -> [0,8).
// Arithmetic and its order are those of v2 / the per-layer launches: bit-identical results.
#define ML_THREADS 768
#define ML_LOADERS 4
#define ML_RING_LINES 128

__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
template <int N>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void line_load(const bf16_t* src, u32x4* ring, int line) {
    __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)src,
                                     (void __attribute__((address_space(3)))*)(ring + line * 64), 16, 0, 2 /* nt */);
}

// acc[i][r] = weight row i (KCH lines of the ring, line_of(i, j)) . x row r (LDS); same accumulation order as mega_dots
template <int KCH, int NR, int KP, typename LineOf>
__device__ __forceinline__ void lds_dots(const u32x4* ring, LineOf line_of, const bf16_t* xs, int lane, float (&acc)[NR][2]) {
#pragma unroll
    for (int i = 0; i < NR; ++i) { acc[i][0] = 0.f; acc[i][1] = 0.f; }
#pragma unroll
    for (int j = 0; j < KCH; ++j) {
        u32x4 wj[NR];
#pragma unroll
        for (int i = 0; i < NR; ++i) wj[i] = ring[line_of(i, j) * 64 + lane];
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const u32x4 xv = *(const u32x4*)(xs + r * KP + j * 512 + lane * 8);
#pragma unroll
            for (int i = 0; i < NR; ++i) acc[i][r] = dot8_bf16(wj[i], xv, acc[i][r]);
        }
    }
}

// RMSNorm of xs[2][2048] in place, in two halves around one barrier; bit-identical to mega_norm / gemv.hip's prologue
struct NormState { u32x4 xv[2]; u32x4 nw; };
__device__ __forceinline__ void norm_part1(const bf16_t* xs, const bf16_t* norm_w, float* red, NormState& st, int tid) {
    const int lane = tid & 63, wave = tid >> 6, c = tid * 8;
    const bool in_k = c < MG_DIM;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        st.xv[r] = in_k ? *(const u32x4*)(xs + r * MG_DIM + c) : u32x4{0, 0, 0, 0};
        const float s = wave_sum(sumsq8(st.xv[r]));
        if (lane == 0) red[wave * 2 + r] = s;
    }
    st.nw = in_k ? load16(norm_w + c) : u32x4{0, 0, 0, 0};
}
__device__ __forceinline__ void norm_part2(bf16_t* xs, const float* red, const NormState& st, float eps, int tid) {
    const int lane = tid & 63, c = tid * 8;
    float tots[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) tots[r] = block_total(red + r, 2, MG_WAVES, lane);
    if (c < MG_DIM) {
#pragma unroll
        for (int r = 0; r < 2; ++r) *(u32x4*)(xs + r * MG_DIM + c) = rms_apply8(st.xv[r], rms_rstd(tots[r], MG_DIM, eps), st.nw);
    }
}

// LDS of the v3 kernel (file scope: shared by the loader and the compute path, two functions inlined into one kernel)
__shared__ __attribute__((aligned(16))) u32x4 ring[ML_RING_LINES * 64];  // 128 KB of weight lines
__shared__ __attribute__((aligned(16))) bf16_t ml_xs[2 * MG_INTER];       // the GEMV operand of the current phase
__shared__ __attribute__((aligned(16))) bf16_t ml_qkv[3 * 2 * 128];
__shared__ __attribute__((aligned(16))) float ml_sm[ATTN_WAVES][16][12];
__shared__ float ml_red[MG_WAVES * 2];
__shared__ int ml_abort;
#define ML_ABORTED() (*(volatile int*)&ml_abort != 0)

// ---- v4 synchronisation: the compute waves never execute s_barrier (it would include the loader waves); they meet through
// one LDS progress word per wave (a barrier id, monotonic) -- ~0.15 us per meeting instead of ~0.05, and in exchange the loaders
// run free: they wait for ring slots by watching the same words and announce filled groups through one counter per loader.
__shared__ uint32_t ml_prog[MG_WAVES];
__shared__ uint32_t ml_ready[ML_LOADERS];
#define ML_SPIN_GUARD (1 << 24)
__device__ __forceinline__ void prog_wait(uint32_t id, int lane) {  // until every compute wave has reached barrier `id`
    for (int tries = 0; tries < ML_SPIN_GUARD; ++tries) {
        const uint32_t v = *(volatile uint32_t*)&ml_prog[lane & (MG_WAVES - 1)];
        if (__all((int)(v - id) >= 0)) return;
    }
    *(volatile int*)&ml_abort = 1;
}
__device__ __forceinline__ void cbar(uint32_t id, int wave, int lane) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this wave's LDS traffic is done
    if (lane == 0) *(volatile uint32_t*)&ml_prog[wave] = id;
    prog_wait(id, lane);
    asm volatile("" ::: "memory");
}
__device__ __forceinline__ void ready_wait(uint32_t groups, int lane) {  // until every loader has filled `groups` groups
    for (int tries = 0; tries < ML_SPIN_GUARD; ++tries) {
        const uint32_t v = *(volatile uint32_t*)&ml_ready[lane & (ML_LOADERS - 1)];
        if (__all((int)(v - groups) >= 0)) return;
    }
    *(volatile int*)&ml_abort = 1;
}

__device__ __forceinline__ void mega_lds_loader(const MegaArgs a) {
    const int tid0 = threadIdx.x, lane0 = tid0 & 63, wave0 = tid0 >> 6, b = blockIdx.x;
    // Loader wave q: chunk q of every 2048-column row, every 4th line of the w2 rows.  Lines go global -> registers -> LDS in
    // consumption order, in groups of <= 12 lines, two register buffers: group g+1 is requested before group g is waited for,
    // copied into the ring as soon as the compute waves' progress words say its slots are free, then announced (ml_ready).
    // Groups per layer (10, so that buffer parity repeats):  0,1 wqkv rows 0-11 / 12-23 -> ring [0,96)   free at 16 of layer l-1
    //   2 wo -> [96,128) (16')   3,4 w1|w3 rows i=0,1 lines 0-47 / 48-95 (barrier 4)   5 lines 96-127 (8)   6 rows i=2 -> [0,48) (12)
    //   7 w2 lines 0-39 -> [48,..) (12)   8 w2 lines 40-87 (the last 8 -> [0,8)) (14)   9 empty
    int q = __builtin_amdgcn_readfirstlane(wave0) - MG_WAVES, bo = b, lo = lane0;
    u32x4 R[2][12];
    auto rb13 = [&](int w) { return 22 * bo + (w < 6 ? 3 * w : 18 + 2 * (w - 6)); };
    auto src4 = [&](const bf16_t* w, int row) { return (const u32x4*)(w + (size_t)row * MG_DIM + q * 512); };
    auto ld = [&](const u32x4* base) { return __builtin_nontemporal_load(base + lo); };
    auto w13_row = [&](const MegaLayer& L, int idx) {  // idx = (i * 8 + w) * 2 + m
        const int i = idx >> 4, w = (idx >> 1) & 7, m = idx & 1;
        return src4(m ? L.w3 : L.w1, rb13(w) + i);
    };
    auto w2_src = [&](const MegaLayer& L, int t) {  // w2: line j = w * 11 + c, this loader takes j = 4 t + q
        const int j = 4 * t + q, w = j / 11, c = j - w * 11;
        return (const u32x4*)(L.w2 + (size_t)(8 * bo + w) * MG_INTER + c * 512);
    };
    auto put = [&](int line, const u32x4& v) { ring[line * 64 + lo] = v; };
    auto announce = [&](uint32_t groups) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (lo == 0) *(volatile uint32_t*)&ml_ready[q] = groups;
    };
    // request group g of layer L into buffer B; returns nothing (counts are compile-time: see ML_GROUP_N)
#define ML_REQ(B, g, L, is_head)                                                                                             \
    do {                                                                                                                   \
        __builtin_amdgcn_sched_barrier(0);                                                                                 \
        if ((g) == 0 || (g) == 1) {                                                                                        \
            _Pragma("unroll") for (int k = 0; k < 12; ++k) {                                                               \
                const int row = (is_head) ? min(11 * bo + min(12 * (g) + k, 10), a.vocab - 1) : 24 * bo + 12 * (g) + k;    \
                R[B][k] = ld(src4((is_head) ? a.head : (L).wqkv, row));                                                    \
            }                                                                                                              \
        } else if ((g) == 2) {                                                                                             \
            _Pragma("unroll") for (int w = 0; w < 8; ++w) R[B][w] = ld(src4((L).wo, 8 * bo + w));                          \
        } else if ((g) == 3 || (g) == 4) {                                                                                 \
            _Pragma("unroll") for (int k = 0; k < 12; ++k) R[B][k] = ld(w13_row((L), 12 * ((g) - 3) + k));                 \
        } else if ((g) == 5) {                                                                                             \
            _Pragma("unroll") for (int k = 0; k < 8; ++k) R[B][k] = ld(w13_row((L), 24 + k));                              \
        } else if ((g) == 6) {                                                                                             \
            _Pragma("unroll") for (int k = 0; k < 12; ++k) R[B][k] = ld(src4((k & 1) ? (L).w3 : (L).w1, rb13(k >> 1) + 2)); \
        } else if ((g) == 7) {                                                                                             \
            _Pragma("unroll") for (int t = 0; t < 10; ++t) R[B][t] = ld(w2_src((L), t));                                   \
        } else if ((g) == 8) {                                                                                             \
            _Pragma("unroll") for (int t = 10; t < 22; ++t) R[B][t - 10] = ld(w2_src((L), t));                             \
        }                                                                                                                  \
        __builtin_amdgcn_sched_barrier(0);                                                                                 \
    } while (0)
    // copy group g from buffer B into its ring slots
#define ML_PUT(B, g)                                                                                                       \
    do {                                                                                                                   \
        __builtin_amdgcn_sched_barrier(0);                                                                                 \
        if ((g) == 0 || (g) == 1) { _Pragma("unroll") for (int k = 0; k < 12; ++k) put((12 * (g) + k) * 4 + q, R[B][k]); } \
        else if ((g) == 2) { _Pragma("unroll") for (int w = 0; w < 8; ++w) put(96 + w * 4 + q, R[B][w]); }                 \
        else if ((g) == 3 || (g) == 4) { _Pragma("unroll") for (int k = 0; k < 12; ++k) put((12 * ((g) - 3) + k) * 4 + q, R[B][k]); } \
        else if ((g) == 5) { _Pragma("unroll") for (int k = 0; k < 8; ++k) put((24 + k) * 4 + q, R[B][k]); }               \
        else if ((g) == 6) { _Pragma("unroll") for (int k = 0; k < 12; ++k) put(k * 4 + q, R[B][k]); }                     \
        else if ((g) == 7) { _Pragma("unroll") for (int t = 0; t < 10; ++t) put(48 + 4 * t + q, R[B][t]); }                \
        else if ((g) == 8) {                                                                                               \
            _Pragma("unroll") for (int t = 10; t < 20; ++t) put(48 + 4 * t + q, R[B][t - 10]);                             \
            _Pragma("unroll") for (int t = 20; t < 22; ++t) put(4 * t + q - 80, R[B][t - 10]);                             \
        }                                                                                                                  \
        __builtin_amdgcn_sched_barrier(0);                                                                                 \
    } while (0)
    // one pipeline stage: request the NEXT group (NN = its request count), wait for THIS group's data, wait for its slots, copy, announce
#define ML_STAGE(g, NN, free_id, REQ_NEXT)                                                                                 \
    do {                                                                                                                   \
        REQ_NEXT;                                                                                                          \
        wait_vm<NN>();                                                                                                     \
        prog_wait((free_id), lo);                                                                                          \
        if (ML_ABORTED()) return;                                                                                          \
        ML_PUT((g) & 1, (g));                                                                                              \
        announce((uint32_t)l * 10u + (g) + 1u);                                                                            \
    } while (0)

    ML_REQ(0, 0, a.layers[0], false);
    int l = 0;
    for (; l < a.n_layer; ++l) {
        asm volatile("" : "+s"(q), "+s"(bo), "+v"(lo));
        const MegaLayer L = a.layers[l];
        const bool last_layer = l + 1 == a.n_layer;
        const MegaLayer Ln = a.layers[last_layer ? l : l + 1];
        const uint32_t id0 = (uint32_t)l * 16u;  // = barrier 16 of the previous layer
        ML_STAGE(0, 12, id0, ML_REQ(1, 1, L, false));
        ML_STAGE(1, 8, id0, ML_REQ(0, 2, L, false));
        ML_STAGE(2, 12, id0, ML_REQ(1, 3, L, false));
        ML_STAGE(3, 12, id0 + 4u, ML_REQ(0, 4, L, false));
        ML_STAGE(4, 8, id0 + 4u, ML_REQ(1, 5, L, false));
        ML_STAGE(5, 12, id0 + 8u, ML_REQ(0, 6, L, false));
        ML_STAGE(6, 10, id0 + 12u, ML_REQ(1, 7, L, false));
        ML_STAGE(7, 12, id0 + 12u, ML_REQ(0, 8, L, false));
        ML_STAGE(8, 0, id0 + 14u, (void)0);
        if (!last_layer) ML_STAGE(9, 12, id0, ML_REQ(0, 0, Ln, false));
        else ML_STAGE(9, 12, id0, ML_REQ(0, 0, Ln, true));
    }
    // LM head: rows 11 b + slot, slots 0..10 (the 12th line of the group is a duplicate, never read) -> ring [0,48)
    wait_vm<0>();
    prog_wait((uint32_t)a.n_layer * 16u, lo);
    if (ML_ABORTED()) return;
    ML_PUT(0, 0);
    announce((uint32_t)a.n_layer * 10u + 1u);
#undef ML_STAGE
#undef ML_PUT
#undef ML_REQ
}

__device__ __forceinline__ void mega_lds_compute(const MegaArgs a) {
    bf16_t* const xs = ml_xs;
    bf16_t* const qkv_s = ml_qkv;
    float (*const sm)[16][12] = ml_sm;
    float* const red = ml_red;
    const int tid0 = threadIdx.x, lane0 = tid0 & 63, wave0 = tid0 >> 6, b = blockIdx.x;
    const uint32_t epoch = a.step_counter[0] * 256u;
    const int P = a.pos[0] + 1;
    const int at_r = b >> 7, at_h = (b >> 3) & 15, at_s = b & 7;
    const int cs = attn_keys_per_split(P);
    const int nact = (P + cs - 1) / cs;
    const bool attending = at_s < nact;  // workgroup-uniform: loaders and compute waves take the same barriers

    // ============================================= compute waves ======================================================
    const bool stamping = a.stamps != nullptr && (b == 0 || b == 255) && wave0 == 0;
    unsigned long long* stamp_base = a.stamps ? a.stamps + (size_t)(b == 0 ? 0 : 1) * a.n_layer * MG_STAMP_SLOTS : nullptr;
    int lane_v = lane0, wave_v = wave0;

    for (int l = 0; l < a.n_layer; ++l) {
        // the thread's ids are re-made opaque every layer: every address below is a function of them, and left alone the
        // compiler hoists dozens of loop-invariant addresses out of the layer loop and spills them
        asm volatile("" : "+v"(lane_v), "+v"(wave_v));
        const int lane = lane_v, wave = wave_v, tid = wave * 64 + lane, lane_o = lane, wave_o = wave, tid_o = tid;
        const int at_w = wave, g = lane >> 4, dl = lane & 15;
        const int kw = attn_keys_per_wave(cs);
        const int at_s0 = at_s * cs + at_w * kw, at_s1 = min(min(P, at_s * cs + cs), at_s0 + kw);
        const int nrow = 8 * b + wave;  // the wo / w2 output row of this wave
        const MegaLayer L = a.layers[l];
        const uint32_t tag0 = epoch + (uint32_t)l * 8u + 1u;
        bf16_t* kc = a.kcache + (size_t)l * a.kv_layer_elems;
        bf16_t* vc = a.vcache + (size_t)l * a.kv_layer_elems;

        // ================= P0: RMSNorm + wqkv (+ KV-cache write)
        MG_STAMP(l, 0);
        if (l == 0) *(u32x4*)(xs + tid * 8) = load16(a.x_in + tid * 8);
        else wave_poll_rowpair<MG_DIM>(a.g_x, tag0 - 8u + 4u, xs, a.abort_flag, &ml_abort, tid_o);
        cbar((uint32_t)l * 16u + 1u, wave, lane);  // 1
        if (ML_ABORTED()) return;
        MG_STAMP(l, 1);
        u32x4 kv[ATTN_U], vv[ATTN_U];
        {   // K/V rows of EARLIER steps for this block's attention slice: independent of this step, requested first
            const int go = lane_o >> 4, dlo = lane_o & 15, s0o = at_s * cs + wave_o * kw;
            const size_t head_base = ((size_t)at_r * MG_HEADS + at_h) * (size_t)a.cache_len * 128 + dlo * 8;
            const int last_old = max(P - 2, 0);  // row P-1 is written by this launch: taken from the hand-off, never touched here
#pragma unroll
            for (int u = 0; u < ATTN_U; ++u) kv[u] = u32x4{0u, 0u, 0u, 0u};
#pragma unroll
            for (int u = 0; u < ATTN_U; ++u) vv[u] = u32x4{0u, 0u, 0u, 0u};
            if (P > 1 && attending) {
#pragma unroll
                for (int u = 0; u < ATTN_U; ++u)
                    if (u * 4 < kw) kv[u] = load_stream16(kc + head_base + (size_t)min(s0o + u * 4 + go, last_old) * 128);
#pragma unroll
                for (int u = 0; u < ATTN_U; ++u)
                    if (u * 4 < kw) vv[u] = load_stream16(vc + head_base + (size_t)min(s0o + u * 4 + go, last_old) * 128);
            }
        }
        const bf16_t rx0 = xs[nrow], rx1 = xs[MG_DIM + nrow];  // residual operands of P2, before the norm overwrites them
        NormState ns;
        norm_part1(xs, L.attn_norm, red, ns, tid);
        cbar((uint32_t)l * 16u + 2u, wave, lane);  // 2
        norm_part2(xs, red, ns, a.eps, tid);
        cbar((uint32_t)l * 16u + 3u, wave, lane);  // 3: normalised x complete, wqkv lines in the ring
        ready_wait((uint32_t)l * 10u + 2u, lane);
        {
            float acc[3][2];
            lds_dots<4, 3, MG_DIM>(ring, [&](int i, int j) { return (wave * 3 + i) * 4 + j; }, xs, lane, acc);
            float v0 = 0.f, v1 = 0.f;
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const float t0 = wave_sum(acc[i][0]), t1 = wave_sum(acc[i][1]);
                if (lane == i) { v0 = t0; v1 = t1; }
            }
            if (lane < 3) {
                const int n = 24 * b + 3 * wave + lane;
                const bf16_t y0 = f2bf(v0), y1 = f2bf(v1);
                gran_store(a.g_qkv + n, (uint32_t)y0 | ((uint32_t)y1 << 16), tag0 + 0u);
                if (a.dbg && n < MG_DIM) { bf16_t* d = a.dbg + ((size_t)l * 5 + 0) * 2 * MG_INTER; d[n] = y0; d[MG_INTER + n] = y1; }
                if (n >= MG_DIM) {
                    const int which = (n - MG_DIM) / MG_DIM, hd = (n - MG_DIM) % MG_DIM;
                    bf16_t* cache = which ? vc : kc;
                    const size_t off = ((size_t)(hd / 128) * a.cache_len + (P - 1)) * 128 + hd % 128;
                    cache[off] = y0;
                    cache[(size_t)MG_HEADS * a.cache_len * 128 + off] = y1;
                }
            }
        }
        MG_STAMP(l, 2);
        cbar((uint32_t)l * 16u + 4u, wave, lane);  // 4

        // ================= P1: attention over keys [0, P)
        if (attending) {
            {
                const bool mine = tid_o < 384;
                const int sec = tid_o >> 7, c = tid_o & 127;
                const gran_t* gp = a.g_qkv + sec * MG_DIM + at_h * 128 + c;
                wave_poll<1>(tag0 + 0u, a.abort_flag, &ml_abort, [&](int, u64& v) { if (!mine) return false; v = gran_load(gp); return true; },
                             [&](int, uint32_t p) { qkv_s[(sec * 2 + 0) * 128 + c] = (bf16_t)p; qkv_s[(sec * 2 + 1) * 128 + c] = (bf16_t)(p >> 16); });
            }
            cbar((uint32_t)l * 16u + 5u, wave, lane);  // 5
            if (ML_ABORTED()) return;
            MG_STAMP(l, 3);
            const u32x4 qv = *(const u32x4*)(qkv_s + (0 * 2 + at_r) * 128 + dl * 8);
#pragma unroll
            for (int u = 0; u < ATTN_U; ++u) {
                if (at_s0 + u * 4 + g == P - 1) {
                    kv[u] = *(const u32x4*)(qkv_s + (1 * 2 + at_r) * 128 + dl * 8);
                    vv[u] = *(const u32x4*)(qkv_s + (2 * 2 + at_r) * 128 + dl * 8);
                }
            }
            const AttnAcc w = attn_wave_partial(kv, vv, qv, at_s0, at_s1, g);
            if (g == 0) {
                sm[at_w][dl][0] = w.m;
                sm[at_w][dl][1] = w.l;
#pragma unroll
                for (int d = 0; d < 8; ++d) sm[at_w][dl][2 + d] = w.o[d];
            }
            cbar((uint32_t)l * 16u + 6u, wave, lane);  // 6
            if (at_w == 0) {
                AttnAcc bb = attn_merge_waves(sm, g, dl);
                bool last = true;
                if (nact > 1) {
                    float* pbase = a.partial + (((size_t)l * 2 + at_r) * MG_HEADS + at_h) * ATTN_NSPLIT * ATTN_PART_STRIDE;
                    last = attn_publish_combine(bb, pbase, a.counters + at_r * MG_HEADS + at_h, at_s, nact, lane);
                }
                if (last && g == 0) {
                    const u32x4 y = attn_normalise(bb);
                    gran_t* gp = a.g_attn + at_r * (MG_DIM / 2) + (at_h * 128 + dl * 8) / 2;
                    gran_store(gp + 0, y.x, tag0 + 1u);
                    gran_store(gp + 1, y.y, tag0 + 1u);
                    gran_store(gp + 2, y.z, tag0 + 1u);
                    gran_store(gp + 3, y.w, tag0 + 1u);
                    if (a.dbg) *(u32x4*)(a.dbg + ((size_t)l * 5 + 1) * 2 * MG_INTER + at_r * MG_INTER + at_h * 128 + dl * 8) = y;
                }
            }
        }

        // ================= P2: h = x + wo(attn)
        MG_STAMP(l, 4);
        wave_poll<4>(tag0 + 1u, a.abort_flag, &ml_abort, [&](int k, u64& v) { v = gran_load(a.g_attn + tid_o + k * MG_THREADS); return true; },
                     [&](int k, uint32_t p) { ((uint32_t*)xs)[tid_o + k * MG_THREADS] = p; });
        cbar((uint32_t)l * 16u + 7u, wave, lane);  // 7: attention vector complete, wo lines in the ring
        ready_wait((uint32_t)l * 10u + 3u, lane);
        if (ML_ABORTED()) return;
        MG_STAMP(l, 5);
        {
            float acc[1][2];
            lds_dots<4, 1, MG_DIM>(ring, [&](int, int j) { return 96 + wave * 4 + j; }, xs, lane, acc);
            const float t0 = wave_sum(acc[0][0]), t1 = wave_sum(acc[0][1]);
            if (lane == 0) {
                const bf16_t y0 = epi_resid(rx0, t0), y1 = epi_resid(rx1, t1);
                gran_store(a.g_h + nrow, (uint32_t)y0 | ((uint32_t)y1 << 16), tag0 + 2u);
                if (a.dbg) { bf16_t* d = a.dbg + ((size_t)l * 5 + 2) * 2 * MG_INTER; d[nrow] = y0; d[MG_INTER + nrow] = y1; }
            }
        }
        cbar((uint32_t)l * 16u + 8u, wave, lane);  // 8

        // ================= P3: act = silu(w1 n(h)) * w3 n(h)
        MG_STAMP(l, 6);
        wave_poll_rowpair<MG_DIM>(a.g_h, tag0 + 2u, xs, a.abort_flag, &ml_abort, tid_o);
        cbar((uint32_t)l * 16u + 9u, wave, lane);  // 9
        if (ML_ABORTED()) return;
        MG_STAMP(l, 7);
        const bf16_t rh0 = xs[nrow], rh1 = xs[MG_DIM + nrow];
        norm_part1(xs, L.ffn_norm, red, ns, tid);
        cbar((uint32_t)l * 16u + 10u, wave, lane);  // 10
        norm_part2(xs, red, ns, a.eps, tid);
        cbar((uint32_t)l * 16u + 11u, wave, lane);  // 11: normalised h complete, w1|w3 rows i=0,1 in the ring
        ready_wait((uint32_t)l * 10u + 6u, lane);
        {
            float acc01[4][2], acc2[2][2];  // (i, m) = (0,w1) (0,w3) (1,w1) (1,w3); then (2,w1) (2,w3)
            lds_dots<4, 4, MG_DIM>(ring, [&](int k, int j) { return (((k >> 1) * 8 + wave) * 2 + (k & 1)) * 4 + j; }, xs, lane, acc01);
            cbar((uint32_t)l * 16u + 12u, wave, lane);  // 12
            cbar((uint32_t)l * 16u + 13u, wave, lane);  // 13: rows i=2 in the ring
        ready_wait((uint32_t)l * 10u + 7u, lane);
            lds_dots<4, 2, MG_DIM>(ring, [&](int m, int j) { return (min(wave, 5) * 2 + m) * 4 + j; }, xs, lane, acc2);
            const int nrows = wave < 6 ? 3 : 2;
            const int rb = 22 * b + (wave < 6 ? 3 * wave : 18 + 2 * (wave - 6));
            float g0 = 0.f, g1 = 0.f, u0 = 0.f, u1 = 0.f;
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const float a0 = wave_sum(i < 2 ? acc01[2 * i][0] : acc2[0][0]), a1 = wave_sum(i < 2 ? acc01[2 * i][1] : acc2[0][1]);
                const float c0 = wave_sum(i < 2 ? acc01[2 * i + 1][0] : acc2[1][0]), c1 = wave_sum(i < 2 ? acc01[2 * i + 1][1] : acc2[1][1]);
                if (lane == i) { g0 = a0; g1 = a1; u0 = c0; u1 = c1; }
            }
            if (lane < nrows) {
                const bf16_t y0 = epi_swiglu(g0, u0), y1 = epi_swiglu(g1, u1);
                gran_store(a.g_act + rb + lane, (uint32_t)y0 | ((uint32_t)y1 << 16), tag0 + 3u);
                if (a.dbg) { bf16_t* d = a.dbg + ((size_t)l * 5 + 3) * 2 * MG_INTER; d[rb + lane] = y0; d[MG_INTER + rb + lane] = y1; }
            }
        }
        cbar((uint32_t)l * 16u + 14u, wave, lane);  // 14

        // ================= P4: x = h + w2(act)
        MG_STAMP(l, 8);
        wave_poll_rowpair<MG_INTER>(a.g_act, tag0 + 3u, xs, a.abort_flag, &ml_abort, tid_o);
        cbar((uint32_t)l * 16u + 15u, wave, lane);  // 15: act complete, w2 lines in the ring
        ready_wait((uint32_t)l * 10u + 9u, lane);
        if (ML_ABORTED()) return;
        MG_STAMP(l, 9);
        {
            float acc[1][2];
            lds_dots<11, 1, MG_INTER>(ring, [&](int, int j) { const int idx = wave * 11 + j; return idx < 80 ? 48 + idx : idx - 80; }, xs, lane, acc);
            const float t0 = wave_sum(acc[0][0]), t1 = wave_sum(acc[0][1]);
            if (lane == 0) {
                const bf16_t y0 = epi_resid(rh0, t0), y1 = epi_resid(rh1, t1);
                gran_store(a.g_x + nrow, (uint32_t)y0 | ((uint32_t)y1 << 16), tag0 + 4u);
                MG_STAMP(l, 10);
                if (a.dbg) { bf16_t* d = a.dbg + ((size_t)l * 5 + 4) * 2 * MG_INTER; d[nrow] = y0; d[MG_INTER + nrow] = y1; }
            }
        }
        cbar((uint32_t)l * 16u + 16u, wave, lane);  // 16
    }

    // ================= LM head on the last position
    const int lane = lane0, wave = wave0, tid = tid0;
    wave_poll_rowpair<MG_DIM>(a.g_x, epoch + (uint32_t)(a.n_layer - 1) * 8u + 1u + 4u, xs, a.abort_flag, &ml_abort, tid);
    cbar((uint32_t)a.n_layer * 16u + 1u, wave, lane);  // h1
    if (ML_ABORTED()) return;
    {
        NormState ns;
        norm_part1(xs, a.final_norm, red, ns, tid);
        cbar((uint32_t)a.n_layer * 16u + 2u, wave, lane);  // h2
        norm_part2(xs, red, ns, a.eps, tid);
    }
    cbar((uint32_t)a.n_layer * 16u + 3u, wave, lane);  // h3
    ready_wait((uint32_t)a.n_layer * 10u + 1u, lane);
    {
        float acc[2][2];
        lds_dots<4, 2, MG_DIM>(ring, [&](int i, int j) { return min(wave + 8 * i, 10) * 4 + j; }, xs, lane, acc);
        float v0 = 0.f, v1 = 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const float t0 = wave_sum(acc[i][0]), t1 = wave_sum(acc[i][1]);
            if (lane == i) { v0 = t0; v1 = t1; }
        }
        if (lane < 2) {
            const int slot = wave + 8 * lane, n = 11 * b + slot;
            if (slot < 11 && n < a.vocab) {
                a.logits[n] = f2bf(v0);
                a.logits[a.vocab + n] = f2bf(v1);
            }
        }
    }
}

__global__ void __launch_bounds__(ML_THREADS, 1) decode_mega_lds_kernel(const MegaArgs a) {
    if (threadIdx.x == 0) ml_abort = 0;
    if (threadIdx.x < MG_WAVES) ml_prog[threadIdx.x] = 0;
    if (threadIdx.x < ML_LOADERS) ml_ready[threadIdx.x] = 0;
    __syncthreads();
    if ((threadIdx.x >> 6) >= MG_WAVES) mega_lds_loader(a);
    else mega_lds_compute(a);
}

hipError_t launch_decode_mega_lds(hipStream_t s, const MegaArgs& a) {
    hipLaunchKernelGGL(decode_mega_lds_kernel, dim3(256), dim3(ML_THREADS), 0, s, a);
    return hipGetLastError();
}
