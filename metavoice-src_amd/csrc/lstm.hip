// LSTM recurrences of the synthesis path on gfx950, fp32: ONE persistent launch per layer for all T time steps.
//
//   * the two LSTM layers inside the EnCodec decoder (codec.hip; was 2 x T dependent launches of ~5 us each), and
//   * the speaker encoder of the reference, `fam/quantiser/audio/speaker_encoder/model.py:21-59`:
//       nn.LSTM(40 -> 256, 3 layers, batch_first) over N partial utterances x 160 mel frames, then
//       relu(Linear(256 -> 256)(h_T of the last layer)), L2-normalised per partial            (`forward`, :54-57)
//     exposed through the C ABI as `mv_spkenc_*`.  The mel front end is host pre-processing, as it is
//     in the reference (librosa on the CPU, `speaker_encoder/audio.py`).
//
// Kernel shape.  The recurrence h_t = f(W_hh h_{t-1} + gin_t) is a chain of T dependent mat-vecs; what bounds it is the
// latency of handing h_t to whoever holds W_hh, not bytes or flops.  So W_hh (4H x H fp32, 1-4 MB) is split over H/8 workgroups
// that each keep their 32 gate rows in LDS for the whole sequence (64 KB at H = 512), and h_t travels between them as 8-byte
// tagged granules {fp32 value, 32-bit tag = launch serial << 16 | t + 1} -- the hand-off of the persistent decode kernel
// (mega_core.h; cdna_hip_programming.md Guideline 16 recipe R2: one agent-scope store by the producer, agent-scope polling
// loads by the consumers, the data is the flag, no fence).  Two granule buffers alternate: a workgroup can only publish
// h_{t+2} after it has read all of h_{t+1}, which exists only once every workgroup has finished reading h_t.
// Every wait is bounded and raises an abort word (a plain launch is not guaranteed co-resident if the device is shared).
// gate order i, f, g, o (torch.nn.LSTM); the non-recurrent half W_ih x_t + b_ih + b_hh is a plain batched product computed
// before the launch (`lstm_inproj_kernel` here, a k = 1 convolution in codec.hip).
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "common.h"
#include "kernels.h"

int mv_fail(int code, const char* fmt, ...);  // engine.hip

namespace {

constexpr int LS_UPW = 8;          // hidden units per workgroup
constexpr int LS_THREADS = 256;    // 4 waves, 2 units each
constexpr int LS_SPIN_LIMIT = 1 << 20;

typedef unsigned long long gran_t;

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

// HK > 0 (H = 64 HK: 256 = the speaker encoder, 512): the wave's 8 gate rows in registers instead of LDS, as in lstm_seq2_kernel below
// (where the measurement is: a dot product whose every fma waits for its own two ds_reads was 3.0 us of a 4.7 us step at H = 512).
template <int HK>
__global__ void __launch_bounds__(LS_THREADS) lstm_seq_kernel(const LstmSeqArgs a) {
    extern __shared__ float lds[];
    const int H = a.H, B = a.B;
    float* w_s = lds;                       // HK = 0: [4 * LS_UPW][H]: row q * LS_UPW + u  <-  W_hh[q * H + u0 + u]
    float* h_s = w_s + (HK > 0 ? 0 : 4 * LS_UPW * H);      // [B][H]
    float* c_s = h_s + (size_t)B * H;       // [B][LS_UPW]
    __shared__ int s_abort;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int u0 = blockIdx.x * LS_UPW;
    if (tid == 0) s_abort = 0;
    constexpr int NR = HK > 0 ? 8 * HK : 1;
    float wh_r[NR];   // row j = uu * 4 + q of this wave, element i = column lane + 64 i
    if constexpr (HK > 0) {
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int i = 0; i < HK; ++i) wh_r[j * HK + i] = a.whh[(size_t)((j & 3) * H + u0 + wave * 2 + (j >> 2)) * H + lane + 64 * i];
    } else {
        wh_r[0] = 0.f;
        for (int i = tid; i < 4 * LS_UPW * H; i += LS_THREADS) {
            const int r = i / H, k = i - r * H, q = r / LS_UPW, u = r - q * LS_UPW;
            w_s[i] = a.whh[(size_t)(q * H + u0 + u) * H + k];
        }
    }
    for (int i = tid; i < B * H; i += LS_THREADS) h_s[i] = 0.f;
    for (int i = tid; i < B * LS_UPW; i += LS_THREADS) c_s[i] = 0.f;
    __syncthreads();
    for (int t = 0; t < a.T; ++t) {
        if (t > 0) {
            // h_{t-1} of every unit and batch row: granules tagged with this launch's serial and t
            const gran_t* src = a.hx + (size_t)(t & 1) * B * H;
            const uint32_t tag = a.tag_base + (uint32_t)t;
            // all of a thread's granules are requested before the first is examined (one round trip, not one per granule);
            // only the ones still carrying an old tag are re-read
            for (int n0 = tid; n0 < B * H; n0 += 4 * LS_THREADS) {
                unsigned long long v[4];
                bool need[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) { need[k] = n0 + k * LS_THREADS < B * H; v[k] = 0; }
                for (int tries = 0;; ++tries) {
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if (need[k]) v[k] = __hip_atomic_load(src + n0 + k * LS_THREADS, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    bool pending = false;
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if (need[k]) {
                            if ((uint32_t)(v[k] >> 32) == tag) { h_s[n0 + k * LS_THREADS] = __uint_as_float((uint32_t)v[k]); need[k] = false; }
                            else pending = true;
                        }
                    if (!pending) break;
                    if ((tries & 1023) == 1023) {
                        const bool peer = __hip_atomic_load(a.abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u;
                        if (peer || tries >= LS_SPIN_LIMIT) {
                            __hip_atomic_store(a.abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            *(volatile int*)&s_abort = 1;
                            break;
                        }
                    }
                    __builtin_amdgcn_s_sleep(1);
                }
            }
            __syncthreads();
            if (s_abort) return;
        }
        // wave w owns units 2w, 2w+1: all four gates of a unit stay in one wave, so the cell update needs no exchange
        gran_t* dst = a.hx + (size_t)((t + 1) & 1) * B * H;
        // Batch rows in groups of 8: the 8 gate-row totals of row b0 + bb land in lanes 8 bb .. 8 bb + 7, then ONE pass of the gate
        // non-linearities and ONE cell update serve 8 rows (64 live lanes instead of 8: with one pass per row the expf / tanhf bodies
        // were ~0.9 us per row and step, 33 of a 62 us step at the speaker encoder's 37 partials)
        for (int b0 = 0; b0 < B; b0 += 8) {
            const int nb = min(8, B - b0);
            float mine = 0.f;
            for (int bb = 0; bb < nb; ++bb) {
                const float* hb = h_s + (size_t)(b0 + bb) * H;
                // the wave's 8 gate rows (2 units x i, f, g, o): row totals are wave-uniform after wave_sum; lane 8 bb + j keeps the
                // total of (unit j >> 2, gate j & 3)
                if constexpr (HK > 0) {
                    float hv[HK];
#pragma unroll
                    for (int i = 0; i < HK; ++i) hv[i] = hb[lane + 64 * i];
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        float s = 0.f;
#pragma unroll
                        for (int i = 0; i < HK; ++i) s = fmaf(wh_r[j * HK + i], hv[i], s);
                        s = wave_sum(s);
                        if (lane == bb * 8 + j) mine = s;
                    }
                } else {
#pragma unroll
                    for (int uu = 0; uu < 2; ++uu)
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const float* w = w_s + (size_t)(q * LS_UPW + wave * 2 + uu) * H;
                            float s = 0.f;
                            for (int k = lane; k < H; k += 64) s = fmaf(w[k], hb[k], s);
                            s = wave_sum(s);
                            if (lane == bb * 8 + uu * 4 + q) mine = s;
                        }
                }
            }
            const int bl = lane >> 3, b = b0 + bl, u = wave * 2 + ((lane >> 2) & 1), q = lane & 3;
            float act = 0.f;
            if (bl < nb) {
                const float g = mine + a.gin[(size_t)b * a.gs_b + (size_t)t * a.gs_t + (size_t)(q * H + u0 + u) * a.gs_r];
                act = q == 2 ? tanhf(g) : sigmoidf_(g);
            }
            const int base = lane & ~3;  // lanes 8 bl and 8 bl + 4 gather their unit's i, f, g, o
            const float ig = __shfl(act, base + 0, 64), fg = __shfl(act, base + 1, 64), gg = __shfl(act, base + 2, 64), og = __shfl(act, base + 3, 64);
            if (q == 0 && bl < nb) {
                const float c = fmaf(fg, c_s[b * LS_UPW + u], ig * gg);
                const float h = og * tanhf(c);
                c_s[b * LS_UPW + u] = c;
                if (t + 1 < a.T)
                    __hip_atomic_store(dst + (size_t)b * H + u0 + u, ((unsigned long long)(a.tag_base + (uint32_t)t + 1u) << 32) | __float_as_uint(h),
                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (a.y) {
                    const size_t o = (size_t)b * a.ys_b + (size_t)t * a.ys_t + (size_t)(u0 + u) * a.ys_u;
                    a.y[o] = h + (a.skip ? a.skip[o] : 0.f);
                }
                if (a.h_last && t == a.T - 1) a.h_last[(size_t)b * H + u0 + u] = h;
            }
        }
        __syncthreads();  // h_s is overwritten by the next poll
    }
}

// Poll `n` tagged granules at `src` into `dst` (LDS): all of a thread's granules are requested before the first is examined; only the
// ones still carrying an old tag are re-read.  Returns false when the wait ran out (abort word raised).
__device__ __forceinline__ bool lstm_poll(const gran_t* src, uint32_t tag, float* dst, int n, uint32_t* abort_flag, int tid) {
    for (int n0 = tid; n0 < n; n0 += 4 * LS_THREADS) {
        unsigned long long v[4];
        bool need[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) { need[k] = n0 + k * LS_THREADS < n; v[k] = 0; }
        for (int tries = 0;; ++tries) {
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (need[k]) v[k] = __hip_atomic_load(src + n0 + k * LS_THREADS, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            bool pending = false;
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (need[k]) {
                    if ((uint32_t)(v[k] >> 32) == tag) { dst[n0 + k * LS_THREADS] = __uint_as_float((uint32_t)v[k]); need[k] = false; }
                    else pending = true;
                }
            if (!pending) break;
            if ((tries & 1023) == 1023) {
                const bool peer = __hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u;
                if (peer || tries >= LS_SPIN_LIMIT) {
                    __hip_atomic_store(abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    return false;
                }
            }
            __builtin_amdgcn_s_sleep(1);
        }
    }
    return true;
}

// Two vectors of n <= 2 * LS_THREADS granules each in ONE wait: all loads of both are in flight together, so a vector that is already
// there (layer 2's h1_t: layer 1 runs ahead) costs no load round trip of its own.
__device__ __forceinline__ bool lstm_poll_pair(const gran_t* src_a, uint32_t tag_a, float* dst_a, const gran_t* src_b, uint32_t tag_b, float* dst_b,
                                               int n, uint32_t* abort_flag, int tid) {
    unsigned long long v[4];
    bool need[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) { need[k] = tid + (k & 1) * LS_THREADS < n; v[k] = 0; }
    for (int tries = 0;; ++tries) {
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (need[k]) v[k] = __hip_atomic_load((k < 2 ? src_a : src_b) + tid + (k & 1) * LS_THREADS, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        bool pending = false;
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (need[k]) {
                if ((uint32_t)(v[k] >> 32) == (k < 2 ? tag_a : tag_b)) {
                    (k < 2 ? dst_a : dst_b)[tid + (k & 1) * LS_THREADS] = __uint_as_float((uint32_t)v[k]);
                    need[k] = false;
                } else pending = true;
            }
        if (!pending) return true;
        if ((tries & 1023) == 1023) {
            const bool peer = __hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u;
            if (peer || tries >= LS_SPIN_LIMIT) {
                __hip_atomic_store(abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                return false;
            }
        }
        __builtin_amdgcn_s_sleep(1);
    }
}

// Two stacked layers, batch 1 (kernels.h: LstmSeq2Args).  Workgroups [0, H/8) are layer 1 -- exactly lstm_seq_kernel, plus a copy of
// every h1_t into the [T][H] granule sequence --, workgroups [H/8, 2 H/8) are layer 2: they keep their 32 rows of W_ih2 AND of W_hh2
// in LDS, and at step t wait for h1_t (sequence buffer: never overwritten, so layer 2 may lag by any number of steps) and h2_{t-1}
// (their own two-slot ring).  Gate pre-activation of layer 2 = (W_hh2 h2_{t-1}) + ((W_ih2 h1_t) + b): both dot products per wave,
// the same reduction as the single-layer kernel.
//
// HK > 0 (H = 64 HK, the codec's 512): the wave's 8 gate rows live in REGISTERS (8 HK floats per lane and matrix), lane l holding
// columns l, l + 64, ...: a step reads its HK values of h from LDS once and runs 8 independent HK-long fma chains.  With the rows in
// LDS (HK = 0, any H) every fma waits for its own two ds_reads: 3.0 us of a 4.7 us step were these 8 x 512 dot products
// (profiles/r03n/lstm_stamps.txt).  Same products in the same order: the results do not change.
template <int HK>
__global__ void __launch_bounds__(LS_THREADS) lstm_seq2_kernel(const LstmSeq2Args a) {
    extern __shared__ float lds[];
    const int H = a.H, nwg = H / LS_UPW;
    const bool second = (int)blockIdx.x >= nwg;
    constexpr int NR = HK > 0 ? 8 * HK : 1;
    float* w_s = lds;                               // HK = 0: [4 * LS_UPW][H]: W_hh rows of this workgroup's units
    float* wi_s = w_s + (HK > 0 ? 0 : 4 * LS_UPW * H);             // HK = 0, layer 2 only: [4 * LS_UPW][H] rows of W_ih2
    float* h_s = wi_s + (HK == 0 && second ? 4 * LS_UPW * H : 0);   // [H]: h_{t-1} of this layer
    float* x_s = h_s + H;                           // layer 2 only: [H]: h1_t
    float* c_s = x_s + H;                           // [LS_UPW]
    __shared__ int s_abort;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int u0 = ((int)blockIdx.x - (second ? nwg : 0)) * LS_UPW;
    const float* whh = second ? a.whh2 : a.whh1;
    if (tid == 0) s_abort = 0;
    float wh_r[NR], wi_r[NR];   // row j = uu * 4 + q of this wave (unit wave * 2 + uu, gate q), element i = column lane + 64 i
    if constexpr (HK > 0) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const size_t grow = (size_t)((j & 3) * H + u0 + wave * 2 + (j >> 2));
#pragma unroll
            for (int i = 0; i < HK; ++i) {
                wh_r[j * HK + i] = whh[grow * H + lane + 64 * i];
                wi_r[j * HK + i] = second ? a.wih2[(size_t)(lane + 64 * i) * a.wih2_ks + grow] : 0.f;
            }
        }
    } else {
        wh_r[0] = wi_r[0] = 0.f;
        for (int i = tid; i < 4 * LS_UPW * H; i += LS_THREADS) {
            const int r = i / H, k = i - r * H, q = r / LS_UPW, u = r - q * LS_UPW;
            w_s[i] = whh[(size_t)(q * H + u0 + u) * H + k];
            if (second) wi_s[i] = a.wih2[(size_t)k * a.wih2_ks + (q * H + u0 + u)];
        }
    }
    for (int i = tid; i < H; i += LS_THREADS) { h_s[i] = 0.f; x_s[i] = 0.f; }
    if (tid < LS_UPW) c_s[tid] = 0.f;
    __syncthreads();
    gran_t* ring = second ? a.hx2 : a.hx1;
    // layer 2: the bias of the wave's 8 gate rows, once (read inside the step it was 8 global loads on every step's serial chain: the
    // stores of the loop keep the compiler from hoisting them)
    float bias_r[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) bias_r[j] = second ? a.bias2[(j & 3) * H + u0 + wave * 2 + (j >> 2)] : 0.f;
#ifdef LSTM_STAMPS
    unsigned long long ph[6] = {0, 0, 0, 0, 0, 0}, t_prev = __builtin_amdgcn_s_memtime(), t_start = t_prev;
#define LS_STAMP(i) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); ph[i] += now_ - t_prev; t_prev = now_; }
#else
#define LS_STAMP(i)
#endif
    for (int t = 0; t < a.T; ++t) {
        // layer 1: this step's input half of the gates, requested BEFORE the wait for h_{t-1} (it was a global load between the dot
        // products and the non-linearities of every step)
        float gin_v = 0.f;
        if (!second && lane < 8)
            gin_v = a.gin1[(size_t)t * a.gs_t + (size_t)((lane & 3) * H + u0 + wave * 2 + (lane >> 2)) * a.gs_r];
        // layer 2: likewise the skip value added to this step's output (both loads land under the wait that follows)
        float skip_v = 0.f;
        if (second && a.skip && (lane == 0 || lane == 4)) skip_v = a.skip[(size_t)t * a.ys_t + (size_t)(u0 + wave * 2 + (lane >> 2)) * a.ys_u];
        // layer 2, rows in registers: h1_t and h2_{t-1} in ONE wait -- h1_t is there early (layer 1 runs ahead), a wait of its own was a
        // pure load round trip (0.85 us of a 3.5 us step, profiles/r03n/lstm_stamps2.txt), and with the rows in registers the 8 extra
        // dot products behind the wait cost 0.1 us.  Rows in LDS: (W_ih2 h1_t) + b between two waits, overlapping h2_{t-1}'s flight.
        float pre[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) pre[j] = 0.f;
        bool ok = true;
        const bool one_wait = HK > 0 && HK <= 2 * LS_THREADS / 64 && second && t > 0;
        if (second) {
            if (one_wait)
                ok = lstm_poll_pair(a.seq1 + (size_t)t * H, a.tag_base + (uint32_t)t + 1u, x_s, ring + (size_t)(t & 1) * H, a.tag_base + (uint32_t)t, h_s,
                                    H, a.abort_flag, tid);
            else ok = lstm_poll(a.seq1 + (size_t)t * H, a.tag_base + (uint32_t)t + 1u, x_s, H, a.abort_flag, tid);
            if (!ok) *(volatile int*)&s_abort = 1;
            __syncthreads();
            if (s_abort) return;
            LS_STAMP(0);
            if constexpr (HK > 0) {
                float xv[HK];
#pragma unroll
                for (int i = 0; i < HK; ++i) xv[i] = x_s[lane + 64 * i];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    float sx = 0.f;
#pragma unroll
                    for (int i = 0; i < HK; ++i) sx = fmaf(wi_r[j * HK + i], xv[i], sx);
                    pre[j] = wave_sum(sx) + bias_r[j];
                }
            } else {
#pragma unroll
                for (int uu = 0; uu < 2; ++uu)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float* wi = wi_s + (size_t)(q * LS_UPW + wave * 2 + uu) * H;
                        float sx = 0.f;
                        for (int k = lane; k < H; k += 64) sx = fmaf(wi[k], x_s[k], sx);
                        pre[uu * 4 + q] = wave_sum(sx) + bias_r[uu * 4 + q];
                    }
            }
            LS_STAMP(1);
        }
        if (!one_wait) {
            if (t > 0) ok = lstm_poll(ring + (size_t)(t & 1) * H, a.tag_base + (uint32_t)t, h_s, H, a.abort_flag, tid);
            if (!ok) *(volatile int*)&s_abort = 1;
            __syncthreads();
            if (s_abort) return;
        }
        LS_STAMP(2);
        float mine = 0.f;
        if constexpr (HK > 0) {
            float hv[HK];
#pragma unroll
            for (int i = 0; i < HK; ++i) hv[i] = h_s[lane + 64 * i];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float sh = 0.f;
#pragma unroll
                for (int i = 0; i < HK; ++i) sh = fmaf(wh_r[j * HK + i], hv[i], sh);
                sh = wave_sum(sh);
                if (second) sh += pre[j];
                if (lane == j) mine = sh;
            }
        } else {
#pragma unroll
            for (int uu = 0; uu < 2; ++uu)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int row = q * LS_UPW + wave * 2 + uu;
                    const float* w = w_s + (size_t)row * H;
                    float sh = 0.f;
                    for (int k = lane; k < H; k += 64) sh = fmaf(w[k], h_s[k], sh);
                    sh = wave_sum(sh);
                    if (second) sh += pre[uu * 4 + q];
                    if (lane == uu * 4 + q) mine = sh;
                }
        }
        LS_STAMP(3);
        float act = 0.f;
        if (lane < 8) {
            const int q = lane & 3;
            const float g = second ? mine : mine + gin_v;
            act = q == 2 ? tanhf(g) : sigmoidf_(g);
        }
        const int base = lane & 4;
        const float ig = __shfl(act, base + 0, 64), fg = __shfl(act, base + 1, 64), gg = __shfl(act, base + 2, 64), og = __shfl(act, base + 3, 64);
        if (lane == 0 || lane == 4) {
            const int u = wave * 2 + (lane >> 2);
            const float c = fmaf(fg, c_s[u], ig * gg);
            const float h = og * tanhf(c);
            c_s[u] = c;
            const unsigned long long gr = ((unsigned long long)(a.tag_base + (uint32_t)t + 1u) << 32) | __float_as_uint(h);
            if (t + 1 < a.T) __hip_atomic_store(ring + (size_t)((t + 1) & 1) * H + u0 + u, gr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (!second) __hip_atomic_store(a.seq1 + (size_t)t * H + u0 + u, gr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else {
                const size_t o = (size_t)t * a.ys_t + (size_t)(u0 + u) * a.ys_u;
                a.y[o] = h + skip_v;
            }
        }
        LS_STAMP(4);
        __syncthreads();  // h_s / x_s are overwritten by the next polls
        LS_STAMP(5);
    }
#ifdef LSTM_STAMPS
    if (tid == 0 && (blockIdx.x == 0 || (int)blockIdx.x == nwg || (int)blockIdx.x == nwg - 1 || (int)blockIdx.x == 2 * nwg - 1))
        printf("lstm_seq2 stamps wg %d layer %d T %d: total %llu ticks; per step: x-wait %.0f ih-dots %.0f h-wait %.0f hh-dots %.0f act+publish %.0f end-barrier %.0f\n",
               (int)blockIdx.x, second ? 2 : 1, a.T, t_prev - t_start, (double)ph[0] / a.T, (double)ph[1] / a.T, (double)ph[2] / a.T, (double)ph[3] / a.T,
               (double)ph[4] / a.T, (double)ph[5] / a.T);
#endif
}

// gin[b][t][row] = bias[row] + sum_k w[row][k] * x[b][t][k]      (the non-recurrent half of every gate, all time steps at once)
constexpr int IP_TT = 8;
__global__ void __launch_bounds__(256) lstm_inproj_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                          float* __restrict__ gin, int BT, int I, int R) {
    extern __shared__ float xs[];  // [IP_TT][I]
    const int bt0 = blockIdx.x * IP_TT;
    for (int i = threadIdx.x; i < IP_TT * I; i += 256) {
        const int j = i / I, k = i - j * I;
        xs[i] = bt0 + j < BT ? x[(size_t)(bt0 + j) * I + k] : 0.f;
    }
    __syncthreads();
    for (int r = blockIdx.y * 256 + threadIdx.x; r < R; r += gridDim.y * 256) {
        float acc[IP_TT];
        const float b = bias[r];
#pragma unroll
        for (int j = 0; j < IP_TT; ++j) acc[j] = b;
        const float* wr = w + (size_t)r * I;
        for (int k = 0; k < I; ++k) {
            const float wv = wr[k];
#pragma unroll
            for (int j = 0; j < IP_TT; ++j) acc[j] = fmaf(wv, xs[j * I + k], acc[j]);
        }
#pragma unroll
        for (int j = 0; j < IP_TT; ++j)
            if (bt0 + j < BT) gin[(size_t)(bt0 + j) * R + r] = acc[j];
    }
}

// out[b] = v / ||v||_2 with v = relu(W h_last[b] + bias)        (speaker_encoder/model.py:55-57)
__global__ void __launch_bounds__(256) spk_head_kernel(const float* __restrict__ h, const float* __restrict__ w, const float* __restrict__ bias,
                                                       float* __restrict__ out, int H, int E) {
    __shared__ float red[4];
    __shared__ float hs[1024];
    const int b = blockIdx.x;
    for (int k = threadIdx.x; k < H; k += 256) hs[k] = h[(size_t)b * H + k];
    __syncthreads();
    float sq = 0.f;
    float v[4];  // E <= 1024
    int n = 0;
    for (int e = threadIdx.x; e < E; e += 256, ++n) {
        float s = bias[e];
        const float* wr = w + (size_t)e * H;
        for (int k = 0; k < H; ++k) s = fmaf(wr[k], hs[k], s);
        s = fmaxf(s, 0.f);
        v[n] = s;
        sq = fmaf(s, s, sq);
    }
    sq = wave_sum(sq);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = sq;
    __syncthreads();
    const float norm = sqrtf(red[0] + red[1] + red[2] + red[3]);
    n = 0;
    for (int e = threadIdx.x; e < E; e += 256, ++n) out[(size_t)b * E + e] = v[n] / norm;
}

// Fallback when the persistent kernels cannot run (their H / 8 workgroups were not co-resident: another process on the GPU): ONE
// launch per time step, h_{t-1} / h_t / c in plain global buffers, W_hh rows read through the caches.  Same unit -> wave -> lane
// layout and the same reductions as lstm_seq_kernel, so the results agree to fp32 rounding; ~5 us per step instead of ~4, and no
// residency requirement.
__global__ void __launch_bounds__(LS_THREADS) lstm_step_kernel(const LstmSeqArgs a, int t, const float* __restrict__ h_prev, float* __restrict__ h_next,
                                                               float* __restrict__ c_state) {
    const int H = a.H, lane = threadIdx.x & 63, wave = threadIdx.x >> 6, u0 = blockIdx.x * LS_UPW;
    for (int b = 0; b < a.B; ++b) {
        const float* hb = h_prev + (size_t)b * H;
        float mine = 0.f;
#pragma unroll
        for (int uu = 0; uu < 2; ++uu)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float* w = a.whh + (size_t)(q * H + u0 + wave * 2 + uu) * H;
                float s = 0.f;
                for (int k = lane; k < H; k += 64) s = fmaf(w[k], hb[k], s);
                s = wave_sum(s);
                if (lane == uu * 4 + q) mine = s;
            }
        float act = 0.f;
        if (lane < 8) {
            const int u = wave * 2 + (lane >> 2), q = lane & 3;
            const float g = mine + a.gin[(size_t)b * a.gs_b + (size_t)t * a.gs_t + (size_t)(q * H + u0 + u) * a.gs_r];
            act = q == 2 ? tanhf(g) : sigmoidf_(g);
        }
        const int base = lane & 4;
        const float ig = __shfl(act, base + 0, 64), fg = __shfl(act, base + 1, 64), gg = __shfl(act, base + 2, 64), og = __shfl(act, base + 3, 64);
        if (lane == 0 || lane == 4) {
            const int u = u0 + wave * 2 + (lane >> 2);
            const float c = fmaf(fg, c_state[(size_t)b * H + u], ig * gg);
            const float h = og * tanhf(c);
            c_state[(size_t)b * H + u] = c;
            h_next[(size_t)b * H + u] = h;
            if (a.y) {
                const size_t o = (size_t)b * a.ys_b + (size_t)t * a.ys_t + (size_t)u * a.ys_u;
                a.y[o] = h + (a.skip ? a.skip[o] : 0.f);
            }
            if (a.h_last && t == a.T - 1) a.h_last[(size_t)b * H + u] = h;
        }
    }
}

bool g_lstm_stepwise = false;   // process-wide and sticky, like the first-stage engine's fall-back to per-layer launches

}  // namespace

// test hooks: MV_LSTM_FORCE_STEPWISE=1 starts in the per-step mode; MV_LSTM_FAULT=1 raises the abort word behind the first persistent
// launch of the process, as a timed-out wait would (tests/test_codec_gpu.py runs both in a child process: the mode is sticky)
static bool lstm_fault_pending() {
    static int armed = getenv("MV_LSTM_FAULT") ? 1 : 0;
    if (armed == 1) { armed = 2; return true; }
    return false;
}
bool lstm_stepwise_mode() {
    static const bool forced = getenv("MV_LSTM_FORCE_STEPWISE") != nullptr;
    return g_lstm_stepwise || forced;
}
void lstm_force_stepwise() {
    if (!g_lstm_stepwise)
        fprintf(stderr, "metavoice_hip: a persistent LSTM launch timed out waiting for a peer workgroup (grid not co-resident: is another kernel "
                        "occupying the GPU?); the LSTM recurrences of this process continue with one launch per time step\n");
    g_lstm_stepwise = true;
}

size_t lstm_seq_lds_bytes(int B, int H) { return ((size_t)4 * LS_UPW * H + (size_t)B * H + (size_t)B * LS_UPW) * sizeof(float); }

hipError_t launch_lstm_seq(hipStream_t s, const LstmSeqArgs& a) {
    if (a.H % LS_UPW != 0 || a.T < 1 || a.T > 65535 || a.B < 1) return hipErrorInvalidValue;
    if (lstm_stepwise_mode()) {
        // the granule scratch doubles as plain fp32 state: h ping | h pong | c, 12 of its 16 bytes per (row, unit)
        float* h0 = (float*)a.hx;
        float* h1 = h0 + (size_t)a.B * a.H;
        float* cs = h1 + (size_t)a.B * a.H;
        hipError_t e = hipMemsetAsync(a.hx, 0, (size_t)2 * a.B * a.H * 8, s);
        if (e != hipSuccess) return e;
        for (int t = 0; t < a.T; ++t)
            hipLaunchKernelGGL(lstm_step_kernel, dim3(a.H / LS_UPW), dim3(LS_THREADS), 0, s, a, t, (t & 1) ? h1 : h0, (t & 1) ? h0 : h1, cs);
        e = hipMemsetAsync(a.hx, 0, (size_t)2 * a.B * a.H * 8, s);   // leave no float that could pass for a tagged granule
        return e != hipSuccess ? e : hipGetLastError();
    }
    static const bool rows_in_lds = getenv("MV_LSTM_ROWS_IN_LDS") != nullptr;   // A/B and tests: the general kernel everywhere
    const int hk = rows_in_lds ? 0 : a.H == 256 ? 4 : a.H == 512 ? 8 : 0;
    // LDS: [B][H] of h and [B][8] of c, plus the 32 gate rows when they are not in registers
    const size_t lds = lstm_seq_lds_bytes(a.B, a.H) - (hk ? (size_t)4 * LS_UPW * a.H * sizeof(float) : 0);
    if (lds > 160 * 1024 - 256) return hipErrorInvalidValue;  // the kernel also has a few static words
    const void* fn = hk == 4 ? (const void*)lstm_seq_kernel<4> : hk == 8 ? (const void*)lstm_seq_kernel<8> : (const void*)lstm_seq_kernel<0>;
    static std::atomic<size_t> raised_on[64][3];   // per device and instance: the attribute belongs to the device's copy of the function
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::atomic<size_t>& raised = raised_on[dev & 63][hk == 4 ? 1 : hk == 8 ? 2 : 0];
    if (lds > 64 * 1024 && lds > raised) {
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        raised = lds;
    }
    if (hk == 4) hipLaunchKernelGGL(lstm_seq_kernel<4>, dim3(a.H / LS_UPW), dim3(LS_THREADS), lds, s, a);
    else if (hk == 8) hipLaunchKernelGGL(lstm_seq_kernel<8>, dim3(a.H / LS_UPW), dim3(LS_THREADS), lds, s, a);
    else hipLaunchKernelGGL(lstm_seq_kernel<0>, dim3(a.H / LS_UPW), dim3(LS_THREADS), lds, s, a);
    if (lstm_fault_pending()) (void)hipMemsetAsync(a.abort_flag, 1, 1, s);
    return hipGetLastError();
}

hipError_t launch_lstm_seq2(hipStream_t s, const LstmSeq2Args& a) {
    if (a.H % LS_UPW != 0 || a.T < 1 || a.T > 65535) return hipErrorInvalidValue;
    static const bool rows_in_lds = getenv("MV_LSTM_ROWS_IN_LDS") != nullptr;   // A/B and tests: the general kernel at H = 512 too
    if (a.H == 512 && !rows_in_lds) {   // EnCodec's size: gate rows in registers, LDS holds h, x, c only
        const size_t lds = ((size_t)2 * a.H + LS_UPW) * sizeof(float);
        hipLaunchKernelGGL(lstm_seq2_kernel<8>, dim3(2 * a.H / LS_UPW), dim3(LS_THREADS), lds, s, a);
    } else {
        const size_t lds = ((size_t)8 * LS_UPW * a.H + (size_t)2 * a.H + LS_UPW) * sizeof(float);
        if (lds > 160 * 1024 - 256) return hipErrorInvalidValue;
        static std::atomic<size_t> raised_on[64];
        int dev = 0;
        (void)hipGetDevice(&dev);
        std::atomic<size_t>& raised = raised_on[dev & 63];
        if (lds > raised) {
            hipError_t e = hipFuncSetAttribute((const void*)lstm_seq2_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return e;
            raised = lds;
        }
        hipLaunchKernelGGL(lstm_seq2_kernel<0>, dim3(2 * a.H / LS_UPW), dim3(LS_THREADS), lds, s, a);
    }
    if (lstm_fault_pending()) (void)hipMemsetAsync(a.abort_flag, 1, 1, s);
    return hipGetLastError();
}

hipError_t launch_lstm_inproj(hipStream_t s, const float* x, const float* w, const float* bias, float* gin, int BT, int I, int R) {
    if ((size_t)IP_TT * I * 4 > 64 * 1024) return hipErrorInvalidValue;
    hipLaunchKernelGGL(lstm_inproj_kernel, dim3((BT + IP_TT - 1) / IP_TT, (R + 255) / 256 > 4 ? 4 : (R + 255) / 256), dim3(256), (size_t)IP_TT * I * 4, s,
                       x, w, bias, gin, BT, I, R);
    return hipGetLastError();
}

// ---- speaker encoder -------------------------------------------------------------------------------------------------

#define SE_TRY(expr)                                                                                          \
    do {                                                                                                      \
        hipError_t _e = (expr);                                                                               \
        if (_e != hipSuccess) return mv_fail(MV_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), \
                                             __FILE__, __LINE__);                                             \
    } while (0)

struct mv_spkenc {
    mv_spkenc_config cfg{};
    hipStream_t stream = nullptr;
    std::vector<float*> w_ih, w_hh, bias;   // device; bias = b_ih + b_hh
    std::vector<std::vector<float>> bias_host;
    std::vector<int> loaded;                // bitmask per layer: 1 w_ih, 2 w_hh, 4 b_ih, 8 b_hh
    float *lin_w = nullptr, *lin_b = nullptr;
    int lin_loaded = 0;
    unsigned long long* hx = nullptr;
    size_t hx_cap = 0;
    uint32_t* abort_flag = nullptr;
    uint32_t serial = 0;
    bool finalized = false;
};

extern "C" {

int mv_spkenc_create(const mv_spkenc_config* cfg, mv_spkenc** out) {
    if (!cfg || !out) return mv_fail(MV_ERR_INVALID, "mv_spkenc_create: null argument");
    const mv_spkenc_config& g = *cfg;
    if (g.n_mels < 1 || g.n_mels > 512 || g.hidden < LS_UPW || g.hidden > 1024 || g.hidden % LS_UPW || g.n_layers < 1 || g.n_layers > 8 ||
        g.emb < 1 || g.emb > 1024)
        return mv_fail(MV_ERR_INVALID, "mv_spkenc_create: unsupported sizes (n_mels %d hidden %d layers %d emb %d)", g.n_mels, g.hidden, g.n_layers, g.emb);
    mv_spkenc* e = new mv_spkenc();
    e->cfg = g;
    if (hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking) != hipSuccess) { delete e; return mv_fail(MV_ERR_HIP, "hipStreamCreate failed"); }
    const int H = g.hidden;
    e->w_ih.assign(g.n_layers, nullptr); e->w_hh.assign(g.n_layers, nullptr); e->bias.assign(g.n_layers, nullptr);
    e->bias_host.assign(g.n_layers, std::vector<float>((size_t)4 * H, 0.f));
    e->loaded.assign(g.n_layers, 0);
    bool ok = true;
    for (int l = 0; l < g.n_layers && ok; ++l) {
        const int I = l == 0 ? g.n_mels : H;
        ok = hipMalloc((void**)&e->w_ih[l], (size_t)4 * H * I * 4) == hipSuccess && hipMalloc((void**)&e->w_hh[l], (size_t)4 * H * H * 4) == hipSuccess &&
             hipMalloc((void**)&e->bias[l], (size_t)4 * H * 4) == hipSuccess;
    }
    ok = ok && hipMalloc((void**)&e->lin_w, (size_t)g.emb * H * 4) == hipSuccess && hipMalloc((void**)&e->lin_b, (size_t)g.emb * 4) == hipSuccess &&
         hipMalloc((void**)&e->abort_flag, 4) == hipSuccess && hipMemset(e->abort_flag, 0, 4) == hipSuccess;
    if (!ok) { mv_spkenc_destroy(e); return mv_fail(MV_ERR_HIP, "mv_spkenc_create: hipMalloc failed"); }
    *out = e;
    return MV_OK;
}

int mv_spkenc_destroy(mv_spkenc* e) {
    if (!e) return MV_OK;
    for (auto* p : e->w_ih) if (p) (void)hipFree(p);
    for (auto* p : e->w_hh) if (p) (void)hipFree(p);
    for (auto* p : e->bias) if (p) (void)hipFree(p);
    if (e->lin_w) (void)hipFree(e->lin_w);
    if (e->lin_b) (void)hipFree(e->lin_b);
    if (e->hx) (void)hipFree(e->hx);
    if (e->abort_flag) (void)hipFree(e->abort_flag);
    if (e->stream) (void)hipStreamDestroy(e->stream);
    delete e;
    return MV_OK;
}

// names of `checkpoint["model_state"]` (speaker_encoder/model.py:33-35,45): lstm.{weight,bias}_{ih,hh}_l{k}, linear.{weight,bias}
int mv_spkenc_load_weight(mv_spkenc* e, const char* name, const float* data, int64_t numel) {
    if (!e || !name || !data) return mv_fail(MV_ERR_INVALID, "mv_spkenc_load_weight: null argument");
    const int H = e->cfg.hidden;
    char kind[8], leaf[4];
    int l = -1;
    if (sscanf(name, "lstm.%7[a-z]_%2[a-z]_l%d", kind, leaf, &l) == 3) {
        if (l < 0 || l >= e->cfg.n_layers) return mv_fail(MV_ERR_INVALID, "%s: LSTM layer out of range", name);
        const bool ih = strcmp(leaf, "ih") == 0;
        const int I = l == 0 ? e->cfg.n_mels : H;
        if (strcmp(kind, "weight") == 0) {
            const int64_t want = (int64_t)4 * H * (ih ? I : H);
            if (numel != want) return mv_fail(MV_ERR_INVALID, "%s: expected %lld elements, got %lld", name, (long long)want, (long long)numel);
            SE_TRY(hipMemcpy(ih ? e->w_ih[l] : e->w_hh[l], data, (size_t)numel * 4, hipMemcpyHostToDevice));
            e->loaded[l] |= ih ? 1 : 2;
        } else if (strcmp(kind, "bias") == 0) {
            if (numel != 4 * H) return mv_fail(MV_ERR_INVALID, "%s: expected %d elements", name, 4 * H);
            if (e->loaded[l] & (ih ? 4 : 8)) return mv_fail(MV_ERR_STATE, "%s loaded twice", name);
            for (int k = 0; k < 4 * H; ++k) e->bias_host[l][k] += data[k];
            e->loaded[l] |= ih ? 4 : 8;
        } else return mv_fail(MV_ERR_INVALID, "unknown tensor %s", name);
        return MV_OK;
    }
    if (strcmp(name, "linear.weight") == 0) {
        if (numel != (int64_t)e->cfg.emb * H) return mv_fail(MV_ERR_INVALID, "linear.weight: wrong size");
        SE_TRY(hipMemcpy(e->lin_w, data, (size_t)numel * 4, hipMemcpyHostToDevice));
        e->lin_loaded |= 1;
        return MV_OK;
    }
    if (strcmp(name, "linear.bias") == 0) {
        if (numel != e->cfg.emb) return mv_fail(MV_ERR_INVALID, "linear.bias: wrong size");
        SE_TRY(hipMemcpy(e->lin_b, data, (size_t)numel * 4, hipMemcpyHostToDevice));
        e->lin_loaded |= 2;
        return MV_OK;
    }
    return mv_fail(MV_ERR_INVALID, "unknown speaker-encoder tensor %s", name);
}

int mv_spkenc_finalize(mv_spkenc* e) {
    if (!e) return mv_fail(MV_ERR_INVALID, "mv_spkenc_finalize: null engine");
    for (int l = 0; l < e->cfg.n_layers; ++l)
        if (e->loaded[l] != 15) return mv_fail(MV_ERR_MISSING_WEIGHT, "speaker encoder: LSTM layer %d incomplete (mask %d)", l, e->loaded[l]);
    if (e->lin_loaded != 3) return mv_fail(MV_ERR_MISSING_WEIGHT, "speaker encoder: linear.weight / linear.bias missing");
    for (int l = 0; l < e->cfg.n_layers; ++l)
        SE_TRY(hipMemcpy(e->bias[l], e->bias_host[l].data(), (size_t)4 * e->cfg.hidden * 4, hipMemcpyHostToDevice));
    e->finalized = true;
    return MV_OK;
}

// mels: host fp32 [B][T][n_mels] (the partial utterances of `embed_utterance`, model.py:88-96); out: host fp32 [B][emb],
// each row L2-normalised (`forward`, :54-57).  The mean over partials and its normalisation stay with the caller (:98-103).
int mv_spkenc_embed(mv_spkenc* e, const float* mels, int32_t B, int32_t T, float* out) {
    if (!e || !mels || !out) return mv_fail(MV_ERR_INVALID, "mv_spkenc_embed: null argument");
    if (!e->finalized) return mv_fail(MV_ERR_STATE, "mv_spkenc_embed before mv_spkenc_finalize");
    const int H = e->cfg.hidden, M = e->cfg.n_mels, E = e->cfg.emb;
    if (B < 1 || T < 1 || T > 65535) return mv_fail(MV_ERR_INVALID, "mv_spkenc_embed: B = %d, T = %d out of range", B, T);
    const int BMAX = (int)(((160 * 1024 - 256) / 4 - 4 * LS_UPW * H) / (H + LS_UPW));  // LDS budget of the recurrence kernel
    hipStream_t s = e->stream;
    float *x = nullptr, *gin = nullptr, *y0 = nullptr, *y1 = nullptr, *hl = nullptr, *d_out = nullptr;
    auto cleanup = [&]() { for (float* p : {x, gin, y0, y1, hl, d_out}) if (p) (void)hipFree(p); };
#define SE_TRY2(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) { cleanup(); return mv_fail(MV_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(_e)); } } while (0)
    SE_TRY2(hipMalloc((void**)&x, (size_t)B * T * M * 4));
    SE_TRY2(hipMalloc((void**)&gin, (size_t)B * T * 4 * H * 4));
    SE_TRY2(hipMalloc((void**)&y0, (size_t)B * T * H * 4));
    SE_TRY2(hipMalloc((void**)&y1, (size_t)B * T * H * 4));
    SE_TRY2(hipMalloc((void**)&hl, (size_t)B * H * 4));
    SE_TRY2(hipMalloc((void**)&d_out, (size_t)B * E * 4));
    if (e->hx_cap < (size_t)2 * BMAX * H) {
        if (e->hx) (void)hipFree(e->hx);
        e->hx = nullptr;
        SE_TRY2(hipMalloc((void**)&e->hx, (size_t)2 * BMAX * H * 8));
        SE_TRY2(hipMemset(e->hx, 0, (size_t)2 * BMAX * H * 8));
        e->hx_cap = (size_t)2 * BMAX * H;
        e->serial = 0;
    }
    SE_TRY2(hipMemcpyAsync(x, mels, (size_t)B * T * M * 4, hipMemcpyHostToDevice, s));
    const float* cur = x;
    int I = M;
    for (int l = 0; l < e->cfg.n_layers; ++l) {
        SE_TRY2(launch_lstm_inproj(s, cur, e->w_ih[l], e->bias[l], gin, B * T, I, 4 * H));
        float* y = (l & 1) ? y1 : y0;
        const bool last = l == e->cfg.n_layers - 1;
        for (int b0 = 0; b0 < B; b0 += BMAX) {   // batch tiles that fit the LDS budget; rows are independent sequences
            const int nb = B - b0 < BMAX ? B - b0 : BMAX;
            if (++e->serial >= 65535u) { SE_TRY2(hipMemsetAsync(e->hx, 0, e->hx_cap * 8, s)); e->serial = 1; }
            LstmSeqArgs a{};
            a.gin = gin + (size_t)b0 * T * 4 * H; a.gs_b = (int64_t)T * 4 * H; a.gs_t = 4 * H; a.gs_r = 1;
            a.whh = e->w_hh[l];
            a.y = last ? nullptr : y + (size_t)b0 * T * H; a.ys_b = (int64_t)T * H; a.ys_t = H; a.ys_u = 1;
            a.skip = nullptr;
            a.h_last = last ? hl + (size_t)b0 * H : nullptr;
            a.hx = e->hx; a.tag_base = e->serial << 16; a.abort_flag = e->abort_flag;
            a.B = nb; a.H = H; a.T = T;
            SE_TRY2(launch_lstm_seq(s, a));
        }
        cur = y;
        I = H;
    }
    hipLaunchKernelGGL(spk_head_kernel, dim3(B), dim3(256), 0, s, hl, e->lin_w, e->lin_b, d_out, H, E);
    SE_TRY2(hipGetLastError());
    uint32_t aborted = 0;
    SE_TRY2(hipMemcpyAsync(out, d_out, (size_t)B * E * 4, hipMemcpyDeviceToHost, s));
    SE_TRY2(hipMemcpyAsync(&aborted, e->abort_flag, 4, hipMemcpyDeviceToHost, s));
    SE_TRY2(hipStreamSynchronize(s));
    cleanup();
#undef SE_TRY2
    if (aborted) {
        (void)hipMemset(e->abort_flag, 0, 4);
        if (!lstm_stepwise_mode()) {   // survive it: redo the call with one launch per time step (and stay there)
            lstm_force_stepwise();
            return mv_spkenc_embed(e, mels, B, T, out);
        }
        return mv_fail(MV_ERR_HIP, "mv_spkenc_embed: the LSTM recurrence failed");
    }
    return MV_OK;
}

}  // extern "C"
