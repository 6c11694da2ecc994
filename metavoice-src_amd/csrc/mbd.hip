// Multi-band diffusion vocoder on gfx950: the waveform stage the reference reaches through `mbd.tokens_to_wav`
// (`fam/llm/decoders.py:13,84-85` -> audiocraft==1.2.0 `MultiBandDiffusion`, third party, not vendored, not installed here).
// What is restated (oracle/mbd.py cites the published module graph): per band a `DiffusionUnet` -- depth x [strided Conv1d ->
// GroupNorm -> act -> ResBlock] encoder, optional BiLSTM bottleneck, mirrored [ResBlock -> GroupNorm -> act -> ConvTranspose1d]
// decoder with skip connections, step embeddings on the encoder layers, codec latent added at the bottleneck -- evaluated 20
// times by the sub-sampled DDPM sampler; then the per-band affine of `MultiBandProcessor.return_sample`, the sum over the bands and
// the 32-band re-equalisation against the EnCodec decode (`re_eq`).
//
// Mapping to the hardware.  Activations are CHANNELS-LAST fp32 [T][C]: a Conv1d(k, stride s) output row m is then the dot
// product of the weight row with ONE CONTIGUOUS WINDOW of the input starting at row m s - pad (k rows x Cin channels), so every
// convolution is a plain GEMM  C[M][N] = A[M][K] W[N][K]^T  whose A rows OVERLAP (lda = s Cin < K = k Cin) -- no im2col buffer.
// A ConvTranspose1d with k = 2 s is the same with K = 2 Cin (rows q-1, q) and N = s Cout (the s output phases of step q).
// 75 % of the FLOPs sit in the 3072-channel layers (M = 1200, N = 3072, K = 9216 at 12.8 s of audio): MFMA-bound, so the GEMM is
// v_mfma_f32_16x16x32_bf16 on 128x128x64 tiles.  The reference asks for fp32 here (`autocast(dtype=float32)`, decoders.py:84);
// bf16 operands alone would leave 8 mantissa bits, so operands are split hi + lo and the product is evaluated as
// a_hi w_hi + a_hi w_lo + a_lo w_hi (three K-segments of one GEMM, fp32 accumulate): ~16 mantissa bits, past the TF32 that
// cuDNN uses for the reference's convolutions by default.  `precision = 1` keeps only the first segment (bf16, 3x faster).
// GroupNorm needs whole-layer statistics: a deterministic two-stage reduction, then one pass that normalises, applies the
// activation and emits the bf16 hi / lo operands of the next convolution (padded with zero rows: the conv padding).
// The thin ends (1 -> C first conv, C -> 1 last transposed conv), the DDPM update, the FIR band splits (julius.SplitBands:
// windowed-sinc low-pass bank, replicate padding) and the band statistics are HBM-bound elementwise / stencil kernels.
#include <hip/hip_runtime.h>

#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "common.h"
#include "kernels.h"

int mv_fail(int code, const char* fmt, ...);  // engine.hip

#define MB_TRY(expr)                                                                                          \
    do {                                                                                                      \
        hipError_t _e = (expr);                                                                               \
        if (_e != hipSuccess) return mv_fail(MV_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), \
                                             __FILE__, __LINE__);                                             \
    } while (0)

namespace {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));

constexpr int MB_MAX_SEG = 12;   // K-segments of one GEMM: (conv taps when dilated) x (precision terms)
constexpr int BN = 128, BK = 64;   // BM = 128 or 64 (template): the 3072-channel layers have M = 1200, 64-row tiles give 456 blocks

constexpr int MB_MAX_BANDS = 8;
struct PF { const float* p[MB_MAX_BANDS]; };     // one pointer per band model (weights differ per band; activations are band-strided)
struct PH { const bf16_t* p[MB_MAX_BANDS]; };

struct GemmSeg {
    const bf16_t* a;   // band 0: first row of the A window of output row 0 for this segment
    int w_lo;          // 0: the hi half of the weights, 1: the lo half
    int64_t w_off;     // element offset of the slice inside a weight row
    int K;             // multiple of 8
};
struct MbdGemmArgs {   // the band models run the same shapes: grid.z = band
    GemmSeg seg[MB_MAX_SEG];
    int n_seg = 0;
    int M = 0, N = 0;
    int64_t lda = 0;       // elements between the windows of consecutive output rows
    int64_t a_band = 0;    // elements between the operand buffers of consecutive bands
    int64_t ldw = 0;       // elements between weight rows
    PH w_hi, w_lo;         // [N][ldw] per band
    float* out = nullptr;  // band 0: [M][ldc]
    int64_t ldc = 0, out_band = 0;
    PF bias;               // [N] per band or null
    const float* residual = nullptr;  // band 0: [M][ldr] or null
    int64_t ldr = 0, res_band = 0;
    PF col_add;            // [N] per band or null (step embedding row)
    // GroupNorm statistics of the OUTPUT from the epilogue (256 x 256 and three-term 128 x 128 kernels; null = off): every tile writes
    // the sum and the sum of squares of its share of each of the gn_G channel groups (gn_cg = N / gn_G channels each, a multiple of 16)
    // to gn_part[band][tile][group][2] (double), gn_final_kernel adds the tiles in index order -- the output is not read again.
    double* gn_part = nullptr;
    int gn_cg = 0, gn_G = 0;
    int gn_period = 0;     // column n belongs to group (n % gn_period) / gn_cg: N for a convolution, Cout for a transposed one (N = stride x Cout)
    // Only the elements whose flat index m * ldc + n lies in [valid_lo, valid_hi) are stored (and counted in the statistics): the
    // transposed convolution writes the s output phases of step q as ONE row of s * Cout values, shifted by `pad` time steps against
    // the tensor it lands in -- the first `pad` steps of row 0 and whatever lies beyond the skip tensor's length are dropped here,
    // so `out` / `residual` may point `pad` steps in front of their buffers (round 6: z = up(x)[:, :, :T] + skip in the epilogue).
    int64_t valid_lo = 0, valid_hi = INT64_MAX;
    // Round 6: the output ALSO as the bf16 hi / lo operand of the next convolution (element f of `out` -> element f + op_shift of the operand
    // buffers: `pad` zero rows in front): the encoder's skip tensors fed a pass of their own that did nothing but this split.  Kernels with
    // the C^T epilogue only (launch_gemm reports through OpOut::written).
    bf16_t* op_hi = nullptr;
    bf16_t* op_lo = nullptr;
    int64_t op_band = 0, op_shift = 0;
};

// C = A W^T on BM x 128 x 128 tiles (BM = 128 or 64), 256 threads = 4 waves (2 x 2), each wave (BM / 2) x 64 = MI x 4 MFMA 16x16x32
// tiles.  Staging: 16-byte chunks, XOR-swizzled by row, global -> registers -> LDS with the NEXT K-tile's loads in flight under the
// current tile's MFMAs.  The K-tile is 128 deep: with one tile of prefetch distance a 64-deep tile left ~0.15 us of MFMA work to
// cover ~0.4 us of load latency (the first version ran at 19 % of the MFMA peak, one iteration per load round trip); two blocks
// per CU overlap one block's staging with the other's MFMAs.  Rows past M / N are clamped (duplicates) and dropped in the epilogue.
template <int BM>
__global__ void __launch_bounds__(256, 3) mbd_gemm_kernel(const MbdGemmArgs a) {
    constexpr int MI = BM / 32;   // MFMA row tiles per wave
    constexpr int CH = BK / 8;    // 16-byte chunks per row
    constexpr int RP = 256 / CH;  // rows staged per pass of the 256 threads
    constexpr int AR = BM / RP;   // A rows staged per thread
    constexpr int WR = BN / RP;   // W rows staged per thread
    __shared__ __attribute__((aligned(16))) u32x4 lds_a[BM * CH];  // [row][16 chunks of 16 B], chunk ^= row & 15
    __shared__ __attribute__((aligned(16))) u32x4 lds_w[BN * CH];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int tile_m = blockIdx.y * BM, tile_n = blockIdx.x * BN, band = blockIdx.z;
    const int M = a.M, N = a.N;
    const int sr = tid / CH, sc = tid % CH;  // thread stages chunk sc of rows sr + RP i

    f32x4_t acc[MI][4];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0, 0, 0, 0};
    const u32x4 zero = {0u, 0u, 0u, 0u};

    for (int s = 0; s < a.n_seg; ++s) {
        const GemmSeg sg = a.seg[s];
        // row offsets as 32-bit element indices off two base pointers (16 64-bit row pointers cost 32 VGPRs and spilled)
        const bf16_t* abase = sg.a + (size_t)band * a.a_band + sc * 8;
        const bf16_t* wbase = (sg.w_lo ? a.w_lo.p[band] : a.w_hi.p[band]) + sg.w_off + sc * 8;
        uint32_t oa[AR], ow[WR];
#pragma unroll
        for (int i = 0; i < AR; ++i) oa[i] = (uint32_t)((size_t)min(tile_m + sr + RP * i, M - 1) * a.lda);
#pragma unroll
        for (int i = 0; i < WR; ++i) ow[i] = (uint32_t)((size_t)min(tile_n + sr + RP * i, N - 1) * a.ldw);
        u32x4 ra[AR], rw[WR];
        const bool in0 = sc * 8 < sg.K;
#pragma unroll
        for (int i = 0; i < AR; ++i) ra[i] = in0 ? load16(abase + oa[i]) : zero;
#pragma unroll
        for (int i = 0; i < WR; ++i) rw[i] = in0 ? load16(wbase + ow[i]) : zero;
        for (int k0 = 0; k0 < sg.K; k0 += BK) {
            __syncthreads();  // everyone is done reading the previous tile
#pragma unroll
            for (int i = 0; i < AR; ++i) { const int r = sr + RP * i; lds_a[r * CH + (sc ^ (r & (CH - 1)))] = ra[i]; }
#pragma unroll
            for (int i = 0; i < WR; ++i) { const int r = sr + RP * i; lds_w[r * CH + (sc ^ (r & (CH - 1)))] = rw[i]; }
            __syncthreads();
            if (k0 + BK < sg.K) {
                const bool in = k0 + BK + sc * 8 < sg.K;
#pragma unroll
                for (int i = 0; i < AR; ++i) ra[i] = in ? load16(abase + oa[i] + k0 + BK) : zero;
#pragma unroll
                for (int i = 0; i < WR; ++i) rw[i] = in ? load16(wbase + ow[i] + k0 + BK) : zero;
            }
#pragma unroll 1
            for (int kk = 0; kk < BK / 32; ++kk) {
                bf16x8_t af[MI], bf[4];
#pragma unroll
                for (int i = 0; i < MI; ++i) {
                    const int r = wm * (BM / 2) + i * 16 + (lane & 15);
                    af[i] = __builtin_bit_cast(bf16x8_t, lds_a[r * CH + ((kk * 4 + (lane >> 4)) ^ (r & (CH - 1)))]);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int r = wn * 64 + j * 16 + (lane & 15);
                    bf[j] = __builtin_bit_cast(bf16x8_t, lds_w[r * CH + ((kk * 4 + (lane >> 4)) ^ (r & (CH - 1)))]);
                }
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
            }
        }
    }
    // epilogue: C fragment (i, j): column lane & 15, rows (lane >> 4) * 4 + reg
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = tile_n + wn * 64 + j * 16 + (lane & 15);
            if (n >= N) continue;
            const float add = (a.bias.p[band] ? a.bias.p[band][n] : 0.f) + (a.col_add.p[band] ? a.col_add.p[band][n] : 0.f);
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int m = tile_m + wm * (BM / 2) + i * 16 + (lane >> 4) * 4 + reg;
                if (m >= M) continue;
                const int64_t f = (int64_t)m * a.ldc + n;
                if (f < a.valid_lo || f >= a.valid_hi) continue;
                float y = acc[i][j][reg] + add;
                if (a.residual) y += a.residual[(size_t)band * a.res_band + (size_t)m * a.ldr + n];
                a.out[(size_t)band * a.out_band + f] = y;
            }
        }
}

// (mean, rstd) of the G groups of one band from its n_rec partial records [n_rec][G][2] (one 256-thread block; red: 256 doubles of LDS):
// thread t sums entry q = t % (2 G) of records t / (2 G), t / (2 G) + 256 / (2 G), ...; the 256 / (2 G) per-q partials are then added in
// index order -- a fixed order.  Round 6: eight records of a thread are in flight at once (added in the old order: same bits); with one load
// per trip the 450 records of a normalise pass cost 14 dependent L2 round trips, most of this kernel's 5.8 us (480 launches per utterance).
// Measured and dropped in round 6 (profiles/r06x): finishing the statistics inside the PRODUCING kernel -- records as write-through stores, a
// ticket counter per band, the last block to arrive reduces -- is bit-identical and removes this launch, but the 240-3600 same-address
// atomics of a launch serialise: the whole stage 124-125 ms against 121-123 with this kernel.
__device__ __forceinline__ void gn_finalize(const double* __restrict__ part, int n_rec, int G, double count, float eps, float* __restrict__ stats, double* red) {
    const int nq = 2 * G, lanes = 256 / nq, q = threadIdx.x % nq, j = threadIdx.x / nq;
    double a = 0.0;
    if (j < lanes) {
        int c = j;
        for (; c + 7 * lanes < n_rec; c += 8 * lanes) {
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = part[(size_t)(c + u * lanes) * nq + q];
#pragma unroll
            for (int u = 0; u < 8; ++u) a += v[u];
        }
        for (; c < n_rec; c += lanes) a += part[(size_t)c * nq + q];
    }
    red[threadIdx.x] = j < lanes ? a : 0.0;
    __syncthreads();
    if ((int)threadIdx.x < G) {
        double sum = 0.0, sq = 0.0;
        for (int k = 0; k < lanes; ++k) { sum += red[k * nq + 2 * threadIdx.x]; sq += red[k * nq + 2 * threadIdx.x + 1]; }
        const double mean = sum / count;
        stats[2 * threadIdx.x] = (float)mean;
        stats[2 * threadIdx.x + 1] = (float)(1.0 / sqrt(fmax(sq / count - mean * mean, 0.0) + (double)eps));
    }
}

// GroupNorm partial record of one 256-thread block (round 6: the statistics ride on whatever kernel PRODUCES the tensor -- the thin GEMM, the
// first convolution, the normalise / activation pass, the GEMM epilogues below -- so that no pass reads a tensor only to sum it).
// s[g] / s2[g]: this thread's fp32 sum and sum of squares per channel group (slots >= G stay zero; a thread has summed a few dozen
// elements).  The 64 lanes of a wave are added in fp32 by the fixed DPP tree of wave_sum_n (2 GM values side by side: ~100 cycles, against
// 24 GM ds_bpermute round trips for a double-precision shuffle tree: this runs at the end of every 4-16 KB block of an HBM-bound pass),
// the four waves in double in index order: the same bits on every run.  rec: [G][2] doubles.
template <int GM>
__device__ __forceinline__ void gn_block_record(const float (&s)[GM], const float (&s2)[GM], int G, double* __restrict__ rec, double (*red)[GM][2]) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float v[2 * GM];
#pragma unroll
    for (int g = 0; g < GM; ++g) { v[2 * g] = s[g]; v[2 * g + 1] = s2[g]; }
    wave_sum_n<2 * GM>(v);
    if (lane == 0) {
#pragma unroll
        for (int g = 0; g < GM; ++g) { red[wave][g][0] = (double)v[2 * g]; red[wave][g][1] = (double)v[2 * g + 1]; }
    }
    __syncthreads();
    if ((int)threadIdx.x < 2 * G) {
        const int g = threadIdx.x >> 1, w = threadIdx.x & 1;
        rec[threadIdx.x] = ((red[0][g][w] + red[1][g][w]) + red[2][g][w]) + red[3][g][w];
    }
}
// group of channel ch when the groups are cg channels wide (G <= GM <= 8: a compare chain instead of a division)
template <int GM>
__device__ __forceinline__ int gn_group_of(int ch, int cg) {
    int g = 0;
#pragma unroll
    for (int q = 1; q < GM; ++q) g += ch >= q * cg;
    return g;
}

// Epilogue of the kernels that accumulate C^T (W fragment as the first MFMA operand): a lane holds row m and the four consecutive columns
// n .. n + 3 of a 16 x 16 block -- bias + step-embedding row, residual, the [valid_lo, valid_hi) window and the store as 16-byte accesses (the
// C layout gave 4-byte ones: a quarter of the bytes per store instruction, and these epilogues write 59 MB per launch); s / s2: the lane's running
// GroupNorm sums of what it stored.  N, ldc, ldr, the band strides and the window bounds are multiples of 4.
__device__ __forceinline__ float4 epilogue_col_add(const MbdGemmArgs& a, int band, int n) {
    float4 add = make_float4(0.f, 0.f, 0.f, 0.f);
    if (a.bias.p[band]) add = *(const float4*)(a.bias.p[band] + n);
    if (a.col_add.p[band]) { const float4 c = *(const float4*)(a.col_add.p[band] + n); add.x += c.x; add.y += c.y; add.z += c.z; add.w += c.w; }
    return add;
}
__device__ __forceinline__ void epilogue_operand4(const MbdGemmArgs& a, int band, int64_t f, const float4& y) {
    if (!a.op_hi) return;
    const uint32_t h0 = pack_bf2(y.x, y.y), h1 = pack_bf2(y.z, y.w);
    const size_t o = (size_t)band * a.op_band + f + a.op_shift;
    *(uint2*)(a.op_hi + o) = make_uint2(h0, h1);
    if (a.op_lo) *(uint2*)(a.op_lo + o) = make_uint2(pack_bf2(y.x - bf_lo(h0), y.y - bf_hi(h0)), pack_bf2(y.z - bf_lo(h1), y.w - bf_hi(h1)));
}
__device__ __forceinline__ void epilogue_store4(const MbdGemmArgs& a, int band, int m, int n, const f32x4_t& acc, const float4& add, float& s, float& s2) {
    if (m >= a.M) return;
    const int64_t f = (int64_t)m * a.ldc + n;
    if (f < a.valid_lo || f >= a.valid_hi) return;
    float4 y = make_float4(acc[0] + add.x, acc[1] + add.y, acc[2] + add.z, acc[3] + add.w);
    if (a.residual) {
        const float4 r = *(const float4*)(a.residual + (size_t)band * a.res_band + (size_t)m * a.ldr + n);
        y.x += r.x; y.y += r.y; y.z += r.z; y.w += r.w;
    }
    *(float4*)(a.out + (size_t)band * a.out_band + f) = y;
    epilogue_operand4(a, band, f, y);
    s += (y.x + y.y) + (y.z + y.w);
    s2 = fmaf(y.x, y.x, fmaf(y.y, y.y, fmaf(y.z, y.z, fmaf(y.w, y.w, s2))));
}

// Thin layers (round 4): N <= 64 output channels, one contiguous K <= 160 window per term -- the 48-channel ResBlock convolutions of the
// outermost level (M = 76 800 rows per band, K = 3 x 48).  On the 128-column tiles above 62 % of every MFMA and of every staged W tile was
// padding (79 us per call, 80 calls per utterance: profiles/r03_final).  Here the WHOLE weight matrix of a band (hi and lo halves) lives in
// registers as B fragments -- NTN x 5 x 2 fragments of 4 VGPRs -- and a wave walks 16-row tiles of A, whose fragments come straight from
// global memory (16 B per lane; the windows of consecutive rows overlap by 2/3, so the 64-byte pieces hit in L1): no LDS, no barrier, no
// padding columns; three MFMAs (a_hi w_hi + a_hi w_lo + a_lo w_hi) per fragment pair as in the fused three-term kernels.  K is walked in
// steps of 32 with the tail zeroed on the A side (144 = 4 x 32 + 16).  Bound by its bytes: A hi | lo, the residual and the fp32 output.
// GM > 0 (round 6): the GroupNorm partials of the output from the epilogue, one record per block (a.gn_part[band][block][G][2]); a lane's
// column j * 16 + col has its own group (groups of 12 channels at 48 / 4: not aligned to the 16-column fragments).
// MT: 16-row tiles per wave (4, or -- round 6 -- 10: a wave fetches the band's whole W, 27.6 KB at 48 channels, for MT x 9 KB of activations it
// reads and writes; at 4 tiles the weights were as many bytes as the data (132 MB of L2 reads per launch), and 1200 blocks were 2.3 rounds of the
// 512 that are resident: launch_gemm picks the MT with fewer rounds x (MT + 1))
template <int NTN, int GM, int MT>
__global__ void __launch_bounds__(256, NTN <= 3 ? 2 : 1) mbd_gemm_thin_kernel(const MbdGemmArgs a) {
    constexpr int KS = 5;   // K-steps of 32 (K <= 160)
    __shared__ double gn_red[4][GM ? GM : 1][2];
    float gs[NTN], gs2[NTN];
#pragma unroll
    for (int j = 0; j < NTN; ++j) gs[j] = gs2[j] = 0.f;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, band = blockIdx.y;
    const int M = a.M, N = a.N, K = a.seg[0].K;
    const int col = lane & 15, kg = lane >> 4;
    const u32x4 zero = {0u, 0u, 0u, 0u};
    bf16x8_t wh[NTN][KS], wl[NTN][KS];
#pragma unroll
    for (int j = 0; j < NTN; ++j) {
        const size_t row = (size_t)min(j * 16 + col, N - 1) * a.ldw + a.seg[0].w_off;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int k = ks * 32 + kg * 8;
            wh[j][ks] = __builtin_bit_cast(bf16x8_t, k < K ? load16(a.w_hi.p[band] + row + k) : zero);
            wl[j][ks] = __builtin_bit_cast(bf16x8_t, k < K ? load16(a.w_lo.p[band] + row + k) : zero);
        }
    }
    const bf16_t* ah = a.seg[0].a + (size_t)band * a.a_band;   // segment 0: a_hi x w_hi, segment 2: a_lo x w_hi (launch_gemm's order)
    const bf16_t* al = a.seg[2].a + (size_t)band * a.a_band;
    const int tile0 = (blockIdx.x * 4 + wave) * (16 * MT);
    // Round 6: software-pipelined over the wave's MT tiles -- the A fragments of tile mt + 1 and the residual of tile mt are requested in
    // front of tile mt's MFMAs (two waves per SIMD at ~240 VGPRs hide nothing of a load round trip by themselves: with one tile at a time
    // the kernel ran at 2.8 TB/s of its own bytes).  Rows past M are clamped on the load side and dropped in the epilogue.
    constexpr int NBUF = NTN <= 3 ? 2 : 1;   // (64 channels: the second fragment set does not fit in 256 registers -- one tile at a time as before)
    bf16x8_t fh[NBUF][KS], fl[NBUF][KS];
    auto load_tile = [&](int buf, int m0) {
        const size_t arow = (size_t)min(m0 + col, M - 1) * a.lda;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int k = ks * 32 + kg * 8;
            fh[buf][ks] = __builtin_bit_cast(bf16x8_t, k < K ? load16(ah + arow + k) : zero);
            fl[buf][ks] = __builtin_bit_cast(bf16x8_t, k < K ? load16(al + arow + k) : zero);
        }
    };
    if (NBUF == 2) load_tile(0, tile0);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int m0 = tile0 + mt * 16, cur = NBUF == 2 ? mt & 1 : 0;
        if (NBUF == 1) load_tile(0, m0);
        else if (mt + 1 < MT) load_tile(cur ^ 1, m0 + 16);
        const int m = m0 + col;
        float4 res[NTN];
#pragma unroll
        for (int j = 0; j < NTN; ++j) {
            const int n = j * 16 + kg * 4;
            const int64_t f = (int64_t)min(m, M - 1) * a.ldc + n;   // (outside the window `residual` may point in front of its buffer)
            res[j] = a.residual && n < N && f >= a.valid_lo && f < a.valid_hi ? *(const float4*)(a.residual + (size_t)band * a.res_band + (size_t)min(m, M - 1) * a.ldr + n)
                                                                                : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        f32x4_t acc[NTN];
#pragma unroll
        for (int j = 0; j < NTN; ++j) acc[j] = f32x4_t{0, 0, 0, 0};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int j = 0; j < NTN; ++j) {
                acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[j][ks], fh[cur][ks], acc[j], 0, 0, 0);   // W first: C^T, see mbd_gemm_t256_kernel
                acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl[j][ks], fh[cur][ks], acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[j][ks], fl[cur][ks], acc[j], 0, 0, 0);
            }
        // C^T fragment j: row lane & 15, columns j * 16 + (lane >> 4) * 4 + reg
#pragma unroll
        for (int j = 0; j < NTN; ++j) {
            const int n = j * 16 + kg * 4;
            if (n >= N || m >= M) continue;
            const int64_t f = (int64_t)m * a.ldc + n;
            if (f < a.valid_lo || f >= a.valid_hi) continue;
            const float4 add = epilogue_col_add(a, band, n);
            const float4 y = make_float4(acc[j][0] + add.x + res[j].x, acc[j][1] + add.y + res[j].y, acc[j][2] + add.z + res[j].z, acc[j][3] + add.w + res[j].w);
            *(float4*)(a.out + (size_t)band * a.out_band + f) = y;
            epilogue_operand4(a, band, f, y);
            gs[j] += (y.x + y.y) + (y.z + y.w);
            gs2[j] = fmaf(y.x, y.x, fmaf(y.y, y.y, fmaf(y.z, y.z, fmaf(y.w, y.w, gs2[j]))));
        }
    }
    if constexpr (GM > 0) {
        float s[GM], s2[GM];
#pragma unroll
        for (int g = 0; g < GM; ++g) s[g] = s2[g] = 0.f;
#pragma unroll
        for (int j = 0; j < NTN; ++j) {
            const int gj = gn_group_of<GM>((j * 16 + kg * 4) % a.gn_period, a.gn_cg);   // (group widths are multiples of 4: launch_gemm checks)
#pragma unroll
            for (int g = 0; g < GM; ++g) { s[g] += g == gj ? gs[j] : 0.f; s2[g] += g == gj ? gs2[j] : 0.f; }
        }
        gn_block_record<GM>(s, s2, a.gn_G, a.gn_part + ((size_t)band * gridDim.x + blockIdx.x) * a.gn_G * 2, gn_red);
    }
}

// The same GEMM with DIRECT global -> LDS loads (`global_load_lds_dwordx4`, gfx950): no staging registers and no ds_write pass, so a
// block needs ~110 VGPRs and 32 KB of LDS and FOUR blocks share a CU; one block's load round trip is covered by the MFMAs of the
// other three (two barriers per K-step, no software pipeline inside a block).  A wave instruction writes 1 KB of LDS linearly
// (wave-uniform base + lane x 16 B) = 8 rows x 8 chunks, so the XOR swizzle of the LDS image is applied on the SOURCE side: lane l
// (row l >> 3, physical chunk l & 7) fetches logical chunk (l & 7) ^ (row & 7).  Used when every K-segment is a multiple of 64.
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;
__global__ void __launch_bounds__(256, 4) mbd_gemm_glds_kernel(const MbdGemmArgs a) {
    constexpr int BM = 128, CH = 8;   // BK = 64
    __shared__ __attribute__((aligned(1024))) u32x4 lds_a[BM * CH];
    __shared__ __attribute__((aligned(1024))) u32x4 lds_w[BN * CH];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int tile_m = blockIdx.y * BM, tile_n = blockIdx.x * BN, band = blockIdx.z;
    const int M = a.M, N = a.N;
    // this wave loads row groups (8 rows each) wave * 4 + i, i < 4, of both operands: lane -> (row lane >> 3, chunk (lane & 7) ^ (row & 7))
    const int lr = lane >> 3, lc = (lane & 7) ^ (lr & 7);   // rows of a group are 8-aligned: (row & 7) == lr

    f32x4_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0, 0, 0, 0};

    for (int s = 0; s < a.n_seg; ++s) {
        const GemmSeg sg = a.seg[s];
        const bf16_t* abase = sg.a + (size_t)band * a.a_band + lc * 8;
        const bf16_t* wbase = (sg.w_lo ? a.w_lo.p[band] : a.w_hi.p[band]) + sg.w_off + lc * 8;
        uint32_t oa[4], ow[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            oa[i] = (uint32_t)((size_t)min(tile_m + (wave * 4 + i) * 8 + lr, M - 1) * a.lda);
            ow[i] = (uint32_t)((size_t)min(tile_n + (wave * 4 + i) * 8 + lr, N - 1) * a.ldw);
        }
        for (int k0 = 0; k0 < sg.K; k0 += 64) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                __builtin_amdgcn_global_load_lds((glb_ptr_t)(abase + oa[i] + k0), (lds_ptr_t)(lds_a + (wave * 4 + i) * 64), 16, 0, 0);
                __builtin_amdgcn_global_load_lds((glb_ptr_t)(wbase + ow[i] + k0), (lds_ptr_t)(lds_w + (wave * 4 + i) * 64), 16, 0, 0);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();   // every wave's loads have landed
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                bf16x8_t af[4], bf[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int r = wm * 64 + i * 16 + (lane & 15);
                    af[i] = __builtin_bit_cast(bf16x8_t, lds_a[r * CH + ((kk * 4 + (lane >> 4)) ^ (r & 7))]);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int r = wn * 64 + j * 16 + (lane & 15);
                    bf[j] = __builtin_bit_cast(bf16x8_t, lds_w[r * CH + ((kk * 4 + (lane >> 4)) ^ (r & 7))]);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
            }
            __syncthreads();   // everyone is done reading before the next tile overwrites the buffers
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = tile_n + wn * 64 + j * 16 + (lane & 15);
            if (n >= N) continue;
            const float add = (a.bias.p[band] ? a.bias.p[band][n] : 0.f) + (a.col_add.p[band] ? a.col_add.p[band][n] : 0.f);
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int m = tile_m + wm * 64 + i * 16 + (lane >> 4) * 4 + reg;
                if (m >= M) continue;
                const int64_t f = (int64_t)m * a.ldc + n;
                if (f < a.valid_lo || f >= a.valid_hi) continue;
                float y = acc[i][j][reg] + add;
                if (a.residual) y += a.residual[(size_t)band * a.res_band + (size_t)m * a.ldr + n];
                a.out[(size_t)band * a.out_band + f] = y;
            }
        }
}

// GroupNorm partials of one output tile (MbdGemmArgs::gn_part).  gs / gs2[j]: this lane's sum and sum of squares (fp32, in the fixed order of
// the epilogue loops) over its rows of the four columns wave_col0 + j * 16 + (lane >> 4) * 4 .. + 3 of column block j of its wave; the four
// columns lie in one group (group widths are multiples of 4), so each lane files its sums under its own groups -- no condition on how the
// groups meet the 16-column blocks (round 6: 48 / 4 = 12 channels per group on the 64 x 192 tiles).  Lanes are added by the DPP tree, the
// waves in double in index order: the same bits on every run.  red: [NW][8][2] doubles of LDS.
template <int NW, int NJ = 4>   // NJ: 16-column blocks a wave holds (<= 4)
__device__ __forceinline__ void gn_tile_record(const MbdGemmArgs& a, const float (&gs)[4], const float (&gs2)[4], int wave_col0, int wave, int lane,
                                               int tid, double* dst, double (*red)[8][2]) {
    float v[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) v[q] = 0.f;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int gj = gn_group_of<8>((wave_col0 + j * 16 + (lane >> 4) * 4) % a.gn_period, a.gn_cg);   // (columns >= N carry zero sums)
#pragma unroll
        for (int g = 0; g < 8; ++g) { v[2 * g] += g == gj ? gs[j] : 0.f; v[2 * g + 1] += g == gj ? gs2[j] : 0.f; }
    }
    wave_sum_n<16>(v);
    if (lane == 0) {
#pragma unroll
        for (int g = 0; g < 8; ++g) { red[wave][g][0] = (double)v[2 * g]; red[wave][g][1] = (double)v[2 * g + 1]; }
    }
    __syncthreads();
    if (tid < 2 * a.gn_G) {
        const int g = tid >> 1, w = tid & 1;
        double t = 0.0;
        for (int wv = 0; wv < NW; ++wv) t += red[wv][g][w];
        dst[tid] = t;
    }
}

// Three-term products in ONE pass: per K-step the hi AND lo tiles of both operands are loaded (2 A + 2 W tiles instead of the
// 3 A + 3 W of three separate segments) and every fragment pair feeds three MFMAs (a_hi w_hi + a_hi w_lo + a_lo w_hi): a third
// fewer operand bytes and three times the MFMA work per barrier pair.  64 KB of LDS, two blocks per CU; segment 0 carries a_hi,
// segment 2 a_lo (launch_gemm's order), one tap.
// Tile shapes: <128, 128> (2 x 2 waves of 64 x 64) and, round 4, <64, 192> (1 x 4 waves of 64 x 48) for the 192-channel layers: on 128-column
// tiles their second column tile was half padding (25 % of the MFMAs and of the W bytes; 13 ms per utterance at 0.39 of the peak).
template <int BMR, int BNR>
__global__ void __launch_bounds__(256, 2) mbd_gemm_glds3_kernel(const MbdGemmArgs a) {
    constexpr int CH = 8;   // BK = 64
    constexpr int WM = BMR / 64, WN = 4 / WM, NI = BNR / WN / 16;   // waves along M / N; 16-column MFMA tiles per wave (4 row tiles always)
    constexpr int AI = BMR / 32, WI = BNR / 32;                     // 8-row wave loads per operand half and wave
    static_assert(NI <= 4 && NI * 16 * WN == BNR && WM * 64 == BMR, "tile shape");
    __shared__ __attribute__((aligned(1024))) u32x4 lds_ah[BMR * CH];
    __shared__ __attribute__((aligned(1024))) u32x4 lds_al[BMR * CH];
    __shared__ __attribute__((aligned(1024))) u32x4 lds_wh[BNR * CH];
    __shared__ __attribute__((aligned(1024))) u32x4 lds_wl[BNR * CH];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int tile_m = blockIdx.y * BMR, tile_n = blockIdx.x * BNR, band = blockIdx.z;
    const int M = a.M, N = a.N, K = a.seg[0].K;
    const int lr = lane >> 3, lc = (lane & 7) ^ (lr & 7);

    f32x4_t acc[4][NI];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = f32x4_t{0, 0, 0, 0};

    const bf16_t* ah = a.seg[0].a + (size_t)band * a.a_band + lc * 8;
    const bf16_t* al = a.seg[2].a + (size_t)band * a.a_band + lc * 8;
    const bf16_t* wh = a.w_hi.p[band] + a.seg[0].w_off + lc * 8;
    const bf16_t* wl = a.w_lo.p[band] + a.seg[0].w_off + lc * 8;
    uint32_t oa[AI], ow[WI];
#pragma unroll
    for (int i = 0; i < AI; ++i) oa[i] = (uint32_t)((size_t)min(tile_m + (wave * AI + i) * 8 + lr, M - 1) * a.lda);
#pragma unroll
    for (int i = 0; i < WI; ++i) ow[i] = (uint32_t)((size_t)min(tile_n + (wave * WI + i) * 8 + lr, N - 1) * a.ldw);
    for (int k0 = 0; k0 < K; k0 += 64) {
#pragma unroll
        for (int i = 0; i < AI; ++i) {
            const int slot = (wave * AI + i) * 64;
            __builtin_amdgcn_global_load_lds((glb_ptr_t)(ah + oa[i] + k0), (lds_ptr_t)(lds_ah + slot), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((glb_ptr_t)(al + oa[i] + k0), (lds_ptr_t)(lds_al + slot), 16, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < WI; ++i) {
            const int slot = (wave * WI + i) * 64;
            __builtin_amdgcn_global_load_lds((glb_ptr_t)(wh + ow[i] + k0), (lds_ptr_t)(lds_wh + slot), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((glb_ptr_t)(wl + ow[i] + k0), (lds_ptr_t)(lds_wl + slot), 16, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
#pragma unroll 1
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8_t afh[4], afl[4], bfh[NI], bfl[NI];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r = wm * 64 + i * 16 + (lane & 15), idx = r * CH + ((kk * 4 + (lane >> 4)) ^ (r & 7));
                afh[i] = __builtin_bit_cast(bf16x8_t, lds_ah[idx]);
                afl[i] = __builtin_bit_cast(bf16x8_t, lds_al[idx]);
            }
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                const int r = wn * (NI * 16) + j * 16 + (lane & 15), idx = r * CH + ((kk * 4 + (lane >> 4)) ^ (r & 7));
                bfh[j] = __builtin_bit_cast(bf16x8_t, lds_wh[idx]);
                bfl[j] = __builtin_bit_cast(bf16x8_t, lds_wl[idx]);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfh[j], afh[i], acc[i][j], 0, 0, 0);   // W first: C^T, see mbd_gemm_t256_kernel
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfl[j], afh[i], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfh[j], afl[i], acc[i][j], 0, 0, 0);
                }
        }
        __syncthreads();
    }
    float gs[4] = {0.f, 0.f, 0.f, 0.f}, gs2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < NI; ++j) {
        const int n = tile_n + wn * (NI * 16) + j * 16 + (lane >> 4) * 4;
        if (n >= N) continue;
        const float4 add = epilogue_col_add(a, band, n);
#pragma unroll
        for (int i = 0; i < 4; ++i) epilogue_store4(a, band, tile_m + wm * 64 + i * 16 + (lane & 15), n, acc[i][j], add, gs[j], gs2[j]);
    }
    if (a.gn_part) {   // the operand tiles are dead (the K loop ended with a barrier): their LDS carries the reduction
        const size_t tile = (size_t)blockIdx.y * gridDim.x + blockIdx.x, tiles = (size_t)gridDim.x * gridDim.y;
        gn_tile_record<4, NI>(a, gs, gs2, tile_n + wn * (NI * 16), wave, lane, tid, a.gn_part + ((size_t)band * tiles + tile) * a.gn_G * 2,
                              reinterpret_cast<double (*)[8][2]>(lds_wh));
    }
}

// Three-term products on 256 x 256 tiles (the 768- and 3072-channel layers: 94 % of an evaluation's FLOPs).
// Why: the 128 x 128 kernel above moves 64 KB of operands from L2 into LDS per 6.3 MFLOP (96 FLOP/B); at the MFMA peak that
// would be 26 TB/s of L2 -> LDS traffic, more than the L2s deliver, so it sits at 0.39 of the peak waiting for operands
// (profiles/r02b_fold).  A 256 x 256 tile halves the operand bytes per FLOP (192 FLOP/B) and lets ONE workgroup per CU
// pipeline its own loads: 8 waves (2 x 4) of 128 x 64 outputs each, K-step 32 = 64 KB of a_hi | a_lo | w_hi | w_lo per stage,
// TWO stages in LDS (128 KB): the loads of step k+1 (direct global -> LDS, no staging registers) are in flight under the 96
// MFMAs per wave of step k; one barrier per K-step.  Per step a wave reads 24 fragments (16 KB) for 96 MFMAs: 192 KB of LDS reads
// per CU against 3072 MFMA cycles per SIMD -- half of the LDS bandwidth.
// LDS image of one operand half: 16 sub-tiles of 16 rows x 64 B = 1 KB each, the unit one wave-wide `global_load_lds_dwordx4`
// writes (wave-uniform base + lane x 16 B).  `ds_read_b128` serves a wave in four fixed 16-lane groups
// ({0-3,12-15,20-27}, {4-11,16-19,28-31}, ... MI355X_MICROARCH.md, LDS): a fragment read (lane -> row lane & 15, chunk lane >> 4)
// is conflict-free when slot(row, chunk) = 4 row + (chunk ^ 2 (row >> 3)) -- each group then covers the 16 slots of a 256-byte
// bank row exactly once.  The permutation is applied on the SOURCE side: lane l of the loading wave fetches row l >> 2, chunk
// (l & 3) ^ 2 (l >> 5), which it writes to slot l.
// Workgroup -> tile: XCD-aware (workgroup L runs on XCD L mod 8): the m-tiles that share a W tile get consecutive ids on ONE XCD,
// so a weight tile crosses the fabric once per XCD instead of once per m-tile.
constexpr int T256_STAGE_BYTES = 4 * 16 * 1024;   // a_hi | a_lo | w_hi | w_lo, 16 sub-tiles of 1 KB each
#ifndef T256_SPREAD
#define T256_SPREAD 1   // r03c, same box: 140.0 ms (spread) vs 142.2 ms (all eight requests right after the barrier) for the whole stage
#endif
// T256_PREFETCH (default 0 = off; the code is kept as a measured negative result): an L2 prefetch T256_PREFETCH K-steps ahead --
// every wave touches the lines its loads of step k + T256_PREFETCH will fetch with two 4-byte-per-lane LDS-DMA requests into a
// scratch slot, and the step barrier becomes a raw s_barrier behind a COUNTED vmcnt that leaves those two requests in flight.
// The idea: two stages are all the LDS holds, so a stage's loads have one step to land, and an L2 miss would stretch the step.
// Measured on one box (profiles/r03c): the whole vocoder stage 142 ms without, 156 / 157 / 161 ms with distance 2 / 3 / 5 -- the
// step is not waiting for misses (the tile order already makes ~80 % of a stage's bytes L2 hits), the extra requests only add load.
#ifndef T256_PREFETCH
#define T256_PREFETCH 0
#endif
// profiling only (results are garbage): 1 = no global loads after the first stage, 2 = additionally no LDS fragment reads
#ifndef T256_ABLATE
#define T256_ABLATE 0
#endif
#ifndef T256_TERM_ORDER
#define T256_TERM_ORDER 0
#endif
constexpr int T256_LDS_BYTES = 2 * T256_STAGE_BYTES + 8 * 256;   // + a 256-byte scratch slot per wave for the prefetch requests
// Round 6, tile_rows = 240: all tiles of a launch run in ONE round (240 workgroups of the 3072-channel layers on 256 CUs), so a launch lasts as
// long as a full tile -- and M = 1200 rows are 4.7 tiles of 256, M = 4800 are 18.75: the last row tile computes mostly clamped duplicates while
// every other CU waits for nothing.  1200 = 5 x 240 and 4800 = 20 x 240: with 240-row tiles (15 blocks of 16 rows; the waves of the lower half
// skip their eighth block -- they share a SIMD with the upper half's waves, so the SIMD's MFMA count per K-step drops from 192 to 180) every
// tile is full and each is 1/16 shorter.  launch_gemm picks the height that minimises rounds x rows.
__global__ void __launch_bounds__(512, 2) mbd_gemm_t256_kernel(const MbdGemmArgs a, int nt_m, int nt_n, int n_blocks, int tile_rows) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char t256_lds[];   // [2 stages][4 halves][16 sub-tiles][1 KB]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 2, wn = wave & 3;
    // bijective XCD remap (cdna_hip_programming.md 5.5 T1): ids of one XCD become consecutive
    int id;
    {
        const int L = blockIdx.x, xcd = L & 7, q = n_blocks >> 3, r = n_blocks & 7;
        id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (L >> 3);
    }
    const int tm = id % nt_m, tn = (id / nt_m) % nt_n, band = id / (nt_m * nt_n);
    const int tile_m = tm * tile_rows, tile_n = tn * 256;
    const int ni = __builtin_amdgcn_readfirstlane(min(8, (tile_rows - wm * 128) >> 4));   // 16-row blocks of this wave that lie inside the tile (8, or 7 below row 240)
    const int M = a.M, N = a.N, K = a.seg[0].K;

    // loads: this wave fills sub-tiles 2 wave, 2 wave + 1 of each operand half; lane -> (row lane >> 2, chunk (lane & 3) ^ 2 (lane >> 5))
    const int lrow = lane >> 2, lchunk = (lane & 3) ^ ((lane >> 5) << 1);
    const bf16_t* ah = a.seg[0].a + (size_t)band * a.a_band + lchunk * 8;
    const bf16_t* al = a.seg[2].a + (size_t)band * a.a_band + lchunk * 8;
    const bf16_t* wh = a.w_hi.p[band] + a.seg[0].w_off + lchunk * 8;
    const bf16_t* wl = a.w_lo.p[band] + a.seg[0].w_off + lchunk * 8;
    uint32_t oa[2], ow[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        oa[i] = (uint32_t)((size_t)min(tile_m + (2 * wave + i) * 16 + lrow, M - 1) * a.lda);
        ow[i] = (uint32_t)((size_t)min(tile_n + (2 * wave + i) * 16 + lrow, N - 1) * a.ldw);
    }
    // piece p of a stage (8 per wave): sub-tile 2 wave + (p >> 2) of operand half p & 3
    const bool skip_a15 = __builtin_amdgcn_readfirstlane(wave == 7 && tile_rows < 256);   // sub-tile 15 of A = the next tile's first rows: not read
    auto stage_piece = [&](int stage, int k0, int p) {
        if (skip_a15 && (p >> 2) == 1 && (p & 3) < 2) return;
        unsigned char* dst = t256_lds + stage * T256_STAGE_BYTES + (p & 3) * 16384 + (2 * wave + (p >> 2)) * 1024;
        const bf16_t* src = (p & 3) == 0 ? ah + oa[p >> 2] : (p & 3) == 1 ? al + oa[p >> 2] : (p & 3) == 2 ? wh + ow[p >> 2] : wl + ow[p >> 2];
        __builtin_amdgcn_global_load_lds((glb_ptr_t)(src + k0), (lds_ptr_t)dst, 16, 0, 0);
    };
    auto stage_load = [&](int stage, int k0) {
#pragma unroll
        for (int p = 0; p < 8; ++p) stage_piece(stage, k0, p);
    };
    // prefetch: lane -> row segment (operand half lane >> 4, row lane & 15) of sub-tile 2 wave + q, q = 0, 1
    uint32_t pa[2];
    const bf16_t* pbase[2];
    {
        const int ph = lane >> 4, pr = lane & 15;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const uint32_t ra = (uint32_t)((size_t)min(tile_m + (2 * wave + q) * 16 + pr, M - 1) * a.lda);
            const uint32_t rw = (uint32_t)((size_t)min(tile_n + (2 * wave + q) * 16 + pr, N - 1) * a.ldw);
            pa[q] = ph < 2 ? ra : rw;
        }
        const bf16_t* b0 = a.seg[0].a + (size_t)band * a.a_band;
        const bf16_t* b1 = a.seg[2].a + (size_t)band * a.a_band;
        const bf16_t* b2 = a.w_hi.p[band] + a.seg[0].w_off;
        const bf16_t* b3 = a.w_lo.p[band] + a.seg[0].w_off;
        pbase[0] = pbase[1] = ph == 0 ? b0 : ph == 1 ? b1 : ph == 2 ? b2 : b3;
    }
    auto prefetch_step = [&](int k0) {
        unsigned char* slot = t256_lds + 2 * T256_STAGE_BYTES + wave * 256;
#pragma unroll
        for (int q = 0; q < 2; ++q) __builtin_amdgcn_global_load_lds((glb_ptr_t)(pbase[q] + pa[q] + k0), (lds_ptr_t)slot, 4, 0, 0);
    };
    // fragment reads: lane -> row lane & 15, chunk lane >> 4 of a sub-tile
    const int frow = lane & 15;
    const int fslot = (frow * 4 + ((lane >> 4) ^ ((frow >> 3) << 1))) * 16;   // byte offset inside the sub-tile

    f32x4_t acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0, 0, 0, 0};

    const int nk = K / 32;
    stage_load(0, 0);
    for (int kt = 0; kt < nk; ++kt) {
        // stage kt has landed for every wave; everybody is through with the other buffer (step kt - 1)
        if (T256_PREFETCH > 0) {
            // the two newest requests of this wave are the prefetches of step kt - 1 (if it issued any): they may stay in flight
            if (kt >= 1 && kt - 1 + T256_PREFETCH < nk) asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
        // The next stage's 8 load instructions per wave: all at once here (SPREAD 0), or two in front of each of the first four MFMA
        // groups (SPREAD 1): a wave issuing 8 LDS-DMA pieces back to back sits ~100 cycles on each (MI355X_MICROARCH.md, LDS-DMA
        // piece issue cost) while its SIMD partner does the same, i.e. the MFMA pipe idles
        const bool more = kt + 1 < nk && T256_ABLATE == 0;
        if (T256_SPREAD == 0 && more) stage_load((kt + 1) & 1, (kt + 1) * 32);
        if (T256_PREFETCH > 0 && T256_SPREAD == 0 && kt + T256_PREFETCH < nk) prefetch_step((kt + T256_PREFETCH) * 32);
        const unsigned char* st = t256_lds + (kt & 1) * T256_STAGE_BYTES;
        auto frag = [&](const unsigned char* p) {
            if (T256_ABLATE == 2) return __builtin_bit_cast(bf16x8_t, u32x4{(uint32_t)(uintptr_t)p, (uint32_t)kt, 0x3c003c00u, 0x3c003c00u});
            return __builtin_bit_cast(bf16x8_t, *(const u32x4*)p);
        };
        bf16x8_t bfh[4], bfl[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int off = (wn * 4 + j) * 1024 + fslot;
            bfh[j] = frag(st + 2 * 16384 + off);
            bfl[j] = frag(st + 3 * 16384 + off);
        }
#pragma unroll
        for (int half = 0; half < 2; ++half) {   // the A fragments four sub-tiles at a time: 32 registers instead of 64
            bf16x8_t afh[4], afl[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int off = (wm * 8 + half * 4 + i) * 1024 + fslot;
                afh[i] = frag(st + 0 * 16384 + off);
                afl[i] = frag(st + 1 * 16384 + off);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (T256_SPREAD == 1 && half == 0 && more) {
                    stage_piece((kt + 1) & 1, (kt + 1) * 32, 2 * i);
                    stage_piece((kt + 1) & 1, (kt + 1) * 32, 2 * i + 1);
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (T256_SPREAD == 2 && more) {   // one piece in front of each of the eight MFMA groups
                    stage_piece((kt + 1) & 1, (kt + 1) * 32, half * 4 + i);
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (half * 4 + i >= ni) continue;
#if T256_TERM_ORDER == 1
                // term-major: the four hi x hi products first (they need a_hi and w_hi only), then lo x hi, then hi x lo
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[half * 4 + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfh[j], afh[i], acc[half * 4 + i][j], 0, 0, 0);
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[half * 4 + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfh[j], afl[i], acc[half * 4 + i][j], 0, 0, 0);
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[half * 4 + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfl[j], afh[i], acc[half * 4 + i][j], 0, 0, 0);
#else
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    f32x4_t c = acc[half * 4 + i][j];
                    // W fragment FIRST (round 6): the accumulator then holds C^T -- a lane has row lane & 15 and FOUR CONSECUTIVE columns
                    // (lane >> 4) * 4 + reg, so the epilogue moves 16 bytes per lane and instruction instead of 4 (same products, same k order)
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfh[j], afh[i], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfl[j], afh[i], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfh[j], afl[i], c, 0, 0, 0);
                    acc[half * 4 + i][j] = c;
                }
#endif
            }
        }
    }
    float gs[4] = {0.f, 0.f, 0.f, 0.f}, gs2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int n = tile_n + wn * 64 + j * 16 + (lane >> 4) * 4;   // this lane's four columns (N is a multiple of 4)
        if (n >= N) continue;
        const float4 add = epilogue_col_add(a, band, n);
#pragma unroll
        for (int i = 0; i < 8; ++i)
            if (i < ni) epilogue_store4(a, band, tile_m + wm * 128 + i * 16 + (lane & 15), n, acc[i][j], add, gs[j], gs2[j]);
    }
    if (a.gn_part) {
        __shared__ double gn_red[8][8][2];
        const size_t tile = (size_t)tm * nt_n + tn, tiles = (size_t)nt_m * nt_n;
        gn_tile_record<8>(a, gs, gs2, tile_n + wn * 64, wave, lane, tid, a.gn_part + ((size_t)band * tiles + tile) * a.gn_G * 2, gn_red);
    }
}

// ---- GroupNorm statistics: deterministic two stages ------------------------------------------------------------------------
// stage 1: block (c, g) sums x and x^2 of group g over rows [c * rows_per, ...) -> part[c][g][2] (double).  A thread walks the
// (rows x C / G) sub-matrix in float4 steps (C / G is a multiple of 4 whenever C is a multiple of 16; otherwise scalar steps).
// Round 4: (i) four float4 loads of a thread are in flight at once -- the sums still run in the old order (one accumulator, ascending i), so
// the bits are unchanged; with one load per loop trip the kernel ran at 0.37 of the HBM rate (profiles/vocoder_hbm_kernels.json);
// (ii) optional second input and output: y = x + x2 is written to `sum_out` and the statistics are those of y -- the decoder's skip
// addition no longer is a pass of its own in front of this one.
__global__ void __launch_bounds__(256) gn_partial_kernel(const float* __restrict__ x, int64_t x_band, int T, int C, int G, int rows_per,
                                                         double* __restrict__ part, const float* __restrict__ x2 = nullptr, int64_t x2_band = 0,
                                                         float* __restrict__ sum_out = nullptr, int64_t out_band = 0) {
    __shared__ double red[4][2];
    const int c = blockIdx.x, g = blockIdx.y, cg = C / G;
    const int t0 = c * rows_per, t1 = min(T, t0 + rows_per);
    const size_t off0 = (size_t)t0 * C + (size_t)g * cg;
    const float* base = x + (size_t)blockIdx.z * x_band + off0;
    const float* base2 = x2 ? x2 + (size_t)blockIdx.z * x2_band + off0 : nullptr;
    float* obase = sum_out ? sum_out + (size_t)blockIdx.z * out_band + off0 : nullptr;
    part += (size_t)blockIdx.z * gridDim.x * G * 2;
    float s = 0.f, s2 = 0.f;
    if ((cg & 3) == 0) {
        const int q4 = cg >> 2, n = (t1 - t0) * q4;
        for (int i0 = threadIdx.x; i0 < n; i0 += 1024) {
            float4 v[4];
            size_t o[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = min(i0 + 256 * u, n - 1), r = i / q4, k = i - r * q4;
                o[u] = (size_t)r * C + 4 * k;
                v[u] = *(const float4*)(base + o[u]);
            }
            if (base2) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const float4 w = *(const float4*)(base2 + o[u]);
                    v[u].x += w.x; v[u].y += w.y; v[u].z += w.z; v[u].w += w.w;
                    if (i0 + 256 * u < n) *(float4*)(obase + o[u]) = v[u];
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (i0 + 256 * u < n) {
                    s += (v[u].x + v[u].y) + (v[u].z + v[u].w);
                    s2 = fmaf(v[u].x, v[u].x, fmaf(v[u].y, v[u].y, fmaf(v[u].z, v[u].z, fmaf(v[u].w, v[u].w, s2))));
                }
        }
    } else {
        const int n = (t1 - t0) * cg;
        for (int i = threadIdx.x; i < n; i += 256) {
            const int r = i / cg, k = i - r * cg;
            const float v = base[(size_t)r * C + k];
            s += v;
            s2 = fmaf(v, v, s2);
        }
    }
    double a = s, b = s2;
    for (int off = 32; off > 0; off >>= 1) { a += __shfl_down(a, off, 64); b += __shfl_down(b, off, 64); }
    if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6][0] = a; red[threadIdx.x >> 6][1] = b; }
    __syncthreads();
    if (threadIdx.x < 2) part[((size_t)c * G + g) * 2 + threadIdx.x] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}
// stage 2 (one block): stats[g] = (mean, rstd) with the biased variance torch.nn.GroupNorm uses.  Thread t sums entry q = t % (2 G)
// of chunks t / (2 G), t / (2 G) + 256 / (2 G), ...; the 256 / (2 G) per-q partials are then added in index order.
__global__ void __launch_bounds__(256) gn_final_kernel(const double* __restrict__ part, int n_chunks, int G, double count, float eps, float* __restrict__ stats) {
    __shared__ double red[256];
    gn_finalize(part + (size_t)blockIdx.x * n_chunks * G * 2, n_chunks, G, count, eps, stats + (size_t)blockIdx.x * 2 * G, red);   // one block per band
}

__device__ __forceinline__ float mb_act(float v, int act) {
    if (act == 0) return fmaxf(v, 0.f);                                            // relu
    if (act == 1) return 0.5f * v * (1.0f + erff(v * 0.70710678118654752f));      // gelu (exact)
    return v / (1.0f + expf(-v));                                                  // silu
}

// y = act(gn(x)) (stats == null: y = x), written as fp32 [T][C] (optional) and as the bf16 hi / lo operands of the next
// convolution: rows [0, front) and [front + T, front + T + back) of the operand buffers are the convolution's zero padding and are
// written here too (no separate memsets).
// Round 6: a thread moves EIGHT consecutive elements per step (two 16-byte loads; one 16-byte store per operand half instead of two
// 8-byte ones) with two steps in flight, a block walks `upb` such units, and -- GM > 0 -- the pass also leaves the GroupNorm partials of
// its OUTPUT (part[band][block][G][2]: the encoder's act(gn(conv(x))) is the first ResBlock's GroupNorm input; gn_partial_kernel read
// it again for that).  C is a multiple of 8, so a unit stays inside a row; its two halves may lie in different groups.
template <int GM>
__global__ void __launch_bounds__(256) norm_act_kernel(const float* __restrict__ x, int64_t x_band, int T, int C, int G, const float* __restrict__ stats,
                                                       const PF gammas, const PF betas, int act, float* __restrict__ y32, int64_t y_band,
                                                       bf16_t* __restrict__ hi, bf16_t* __restrict__ lo, int64_t op_band, int front, int back, int upb,
                                                       double* __restrict__ part, const double* __restrict__ fin_part, int fin_rec, double fin_count) {
    __shared__ double gn_red[4][GM ? GM : 1][2];
    __shared__ double fin_red[256];
    __shared__ float st[16];   // (mean, rstd) of the G <= 8 groups of this band
    const int band = blockIdx.y;
    // The statistics of x: finished by gn_final_kernel (stats), or -- fin_part: the producer's records, up to GN_DEFER_MAX_RECORDS of them --
    // finished HERE by every block in the same fixed order (4-32 KB of L2 reads and 1-2 us per block, all blocks of a CU at once) instead of a
    // 4.8 us launch and two kernel boundaries in front of this pass (23 of the 24 normalisations of an evaluation at the benched shape).
    if (fin_part) gn_finalize(fin_part + (size_t)band * fin_rec * G * 2, fin_rec, G, fin_count, 1e-5f, st, fin_red);
    else if (stats && (int)threadIdx.x < 2 * G) st[threadIdx.x] = stats[(size_t)band * 2 * G + threadIdx.x];
    __syncthreads();
    const bool normed = stats != nullptr || fin_part != nullptr;
    const uint32_t units = (uint32_t)(((size_t)(front + T + back) * C) >> 3);
    const uint32_t u0 = blockIdx.x * (uint32_t)upb, u1 = min(units, u0 + (uint32_t)upb);
    x += (size_t)band * x_band;
    if (y32) y32 += (size_t)band * y_band;
    if (hi) hi += (size_t)band * op_band;
    if (lo) lo += (size_t)band * op_band;
    const float* gamma = gammas.p[band];
    const float* beta = betas.p[band];
    const int cg = C / G;
    // channel of a unit's first element: (8 u) mod C (the margins are whole rows) -- one integer modulo per thread, then a running sum: the
    // thread's units are 256 apart (with this and the group look-ups as divisions, index arithmetic was half of the kernel's instructions)
    const int d256 = 2048 % C;
    int ch_run = (int)((8ull * (u0 + threadIdx.x)) % (uint32_t)C);
    const uint32_t body0 = (uint32_t)front * C, body1 = body0 + (uint32_t)T * C;
    float s[GM ? GM : 1], s2[GM ? GM : 1];
#pragma unroll
    for (int g = 0; g < (GM ? GM : 1); ++g) s[g] = s2[g] = 0.f;
    for (uint32_t ub = u0 + threadIdx.x; ub < u1; ub += 512) {
        uint32_t e[2];
        int chq[2];
        bool live[2], body[2];
        float4 xv[2][2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const uint32_t u = ub + 256 * q;
            chq[q] = ch_run;
            ch_run += d256; ch_run -= ch_run >= C ? C : 0;
            live[q] = u < u1;
            e[q] = u * 8;   // element index inside the operand buffer
            body[q] = live[q] && e[q] >= body0 && e[q] < body1;
            if (body[q]) {
                xv[q][0] = *(const float4*)(x + (e[q] - body0));
                xv[q][1] = *(const float4*)(x + (e[q] - body0) + 4);
            }
        }
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            if (!live[q]) continue;
            if (!body[q]) {
                if (hi) *(uint4*)(hi + e[q]) = make_uint4(0u, 0u, 0u, 0u);
                if (lo) *(uint4*)(lo + e[q]) = make_uint4(0u, 0u, 0u, 0u);
                continue;
            }
            const uint32_t j = e[q] - body0;   // element index inside x
            const int ch = chq[q];
            float v[8] = {xv[q][0].x, xv[q][0].y, xv[q][0].z, xv[q][0].w, xv[q][1].x, xv[q][1].y, xv[q][1].z, xv[q][1].w};
            if (normed) {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const float4 gm = *(const float4*)(gamma + ch + 4 * h), bt = *(const float4*)(beta + ch + 4 * h);
                    const float gmv[4] = {gm.x, gm.y, gm.z, gm.w}, btv[4] = {bt.x, bt.y, bt.z, bt.w};
                    if ((cg & 3) == 0) {  // the four channels share a group
                        const int g = gn_group_of<8>(ch + 4 * h, cg);
                        const float mean = st[2 * g], rstd = st[2 * g + 1];
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[4 * h + r] = mb_act((v[4 * h + r] - mean) * rstd * gmv[r] + btv[r], act);
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int g = (ch + 4 * h + r) / cg;
                            v[4 * h + r] = mb_act((v[4 * h + r] - st[2 * g]) * st[2 * g + 1] * gmv[r] + btv[r], act);
                        }
                    }
                }
            }
            if constexpr (GM > 0) {   // (the host asks for this only when the groups are float4-aligned)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int gh = gn_group_of<GM>(ch + 4 * h, cg);
                    const float ps = (v[4 * h] + v[4 * h + 1]) + (v[4 * h + 2] + v[4 * h + 3]);
                    const float ps2 = fmaf(v[4 * h], v[4 * h], fmaf(v[4 * h + 1], v[4 * h + 1], fmaf(v[4 * h + 2], v[4 * h + 2], v[4 * h + 3] * v[4 * h + 3])));
#pragma unroll
                    for (int g = 0; g < GM; ++g) { s[g] += g == gh ? ps : 0.f; s2[g] += g == gh ? ps2 : 0.f; }
                }
            }
            if (y32) {
                *(float4*)(y32 + j) = make_float4(v[0], v[1], v[2], v[3]);
                *(float4*)(y32 + j + 4) = make_float4(v[4], v[5], v[6], v[7]);
            }
            if (hi) {
                const uint32_t h0 = pack_bf2(v[0], v[1]), h1 = pack_bf2(v[2], v[3]), h2 = pack_bf2(v[4], v[5]), h3 = pack_bf2(v[6], v[7]);
                *(uint4*)(hi + e[q]) = make_uint4(h0, h1, h2, h3);
                if (lo)
                    *(uint4*)(lo + e[q]) = make_uint4(pack_bf2(v[0] - bf_lo(h0), v[1] - bf_hi(h0)), pack_bf2(v[2] - bf_lo(h1), v[3] - bf_hi(h1)),
                                                     pack_bf2(v[4] - bf_lo(h2), v[5] - bf_hi(h2)), pack_bf2(v[6] - bf_lo(h3), v[7] - bf_hi(h3)));
            }
        }
    }
    if constexpr (GM > 0) gn_block_record<GM>(s, s2, G, part + ((size_t)band * gridDim.x + blockIdx.x) * G * 2, gn_red);
}

// first encoder convolution, 1 -> C channels: out[m][c] = sum_k w[c][k] x[m s + k - pad]   (zero padding, no bias).  A thread
// produces four consecutive channels of one output step (one float4 store); the k input samples are shared by the C / 4 threads
// of a step, the C x k weights sit in the scalar / L1 caches.  32-bit indexing (M C < 2^31).
__global__ void __launch_bounds__(256) conv_in_kernel(const float* __restrict__ x, int T, const PF ws, int C, int k, int s, int pad,
                                                      float* __restrict__ out, int64_t out_band, int M) {
    // the C x k weights once per block into LDS, transposed to [j][c] (a thread reads its four channels of tap j as one float4):
    // read per output from global they were 32 loads per 16-byte store and the kernel ran at 0.4 TB/s of output (r03_final)
    extern __shared__ __attribute__((aligned(16))) float wl[];   // [k][C]
    for (int i = threadIdx.x; i < C * k; i += 256) wl[(i % k) * C + i / k] = ws.p[blockIdx.y][i];
    __syncthreads();
    const int q4 = C >> 2;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= M * q4) return;
    const int m = i / q4, c = (i - m * q4) * 4;
    x += (size_t)blockIdx.y * T;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int j = 0; j < k; ++j) {
        const int t = m * s + j - pad;
        const float xv = (t >= 0 && t < T) ? x[t] : 0.f;
        const float4 wv = *(const float4*)(wl + j * C + c);
        acc[0] = fmaf(wv.x, xv, acc[0]); acc[1] = fmaf(wv.y, xv, acc[1]); acc[2] = fmaf(wv.z, xv, acc[2]); acc[3] = fmaf(wv.w, xv, acc[3]);
    }
    *(float4*)(out + (size_t)blockIdx.y * out_band + (size_t)m * C + c) = make_float4(acc[0], acc[1], acc[2], acc[3]);
}
// Round 6: the same convolution for k = 8, s = 4, C <= 1024 with the block's input samples staged in LDS (one coalesced pass; a row's window
// is two aligned 16-byte LDS reads shared by the C / 4 threads of the row instead of eight global loads per thread), a thread's 4 x 8
// weights in registers (its channel quad is fixed: thread -> (row t / (C / 4), quad t % (C / 4)), rows advance by 256 / (C / 4) per pass) and
// -- GM > 0 -- the GroupNorm partials of the OUTPUT from the same pass (part[band][block][G][2]; the quad lies in one group).  Same fmaf
// order per output as conv_in_kernel: identical bits.
constexpr int CONV_IN_PASSES = 8;
template <int GM>
__global__ void __launch_bounds__(256) conv_in_rows_kernel(const float* __restrict__ x, int T, const PF ws, int C, int pad, float* __restrict__ out,
                                                           int64_t out_band, int M, int G, double* __restrict__ part) {
    constexpr int K = 8, S = 4;
    extern __shared__ __attribute__((aligned(16))) float xs[];   // [rows_per_block * S + K - S]
    __shared__ double gn_red[4][GM ? GM : 1][2];
    const int band = blockIdx.y, q4 = C >> 2, rp = 256 / q4, rows_per_block = rp * CONV_IN_PASSES;
    const int m0 = blockIdx.x * rows_per_block, n_x = rows_per_block * S + K - S;
    x += (size_t)band * T;
    for (int i = threadIdx.x; i < n_x; i += 256) {
        const int t = m0 * S - pad + i;
        xs[i] = (t >= 0 && t < T) ? x[t] : 0.f;
    }
    const int c4 = threadIdx.x % q4, r = threadIdx.x / q4, c = c4 * 4;
    const bool active = r < rp;
    float w[4][K];
    if (active) {
        const float4* wp = (const float4*)(ws.p[band] + (size_t)c * K);   // [C][k]: 32 contiguous floats for the quad
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 a = wp[2 * q], b = wp[2 * q + 1];
            w[q][0] = a.x; w[q][1] = a.y; w[q][2] = a.z; w[q][3] = a.w; w[q][4] = b.x; w[q][5] = b.y; w[q][6] = b.z; w[q][7] = b.w;
        }
    }
    __syncthreads();
    float ps = 0.f, ps2 = 0.f;
    if (active) {
        out += (size_t)band * out_band;
#pragma unroll 2
        for (int it = 0; it < CONV_IN_PASSES; ++it) {
            const int lr = r + it * rp, m = m0 + lr;
            if (m >= M) break;
            const float4 xa = *(const float4*)(xs + lr * S), xb = *(const float4*)(xs + lr * S + 4);
            const float xw[K] = {xa.x, xa.y, xa.z, xa.w, xb.x, xb.y, xb.z, xb.w};
            float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < K; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[q] = fmaf(w[q][j], xw[j], acc[q]);
            *(float4*)(out + (size_t)m * C + c) = make_float4(acc[0], acc[1], acc[2], acc[3]);
            if (GM) {
                ps += (acc[0] + acc[1]) + (acc[2] + acc[3]);
                ps2 = fmaf(acc[0], acc[0], fmaf(acc[1], acc[1], fmaf(acc[2], acc[2], fmaf(acc[3], acc[3], ps2))));
            }
        }
    }
    if constexpr (GM > 0) {
        float s[GM], s2[GM];
        const int gc = gn_group_of<GM>(c, C / G);
#pragma unroll
        for (int g = 0; g < GM; ++g) { s[g] = active && g == gc ? ps : 0.f; s2[g] = active && g == gc ? ps2 : 0.f; }
        gn_block_record<GM>(s, s2, G, part + ((size_t)band * gridDim.x + blockIdx.x) * G * 2, gn_red);
    }
}
// last decoder transposed convolution, C -> 1 channel: out[t'] = sum_c sum_{(t, j): t s + j - pad = t'} x[t][c] w[c][j].  One thread
// per output sample: it walks the <= k / s input rows that reach it (a row is C contiguous floats, shared through L1 by the s
// neighbouring samples); the weights are transposed into LDS as [j][c] once per block.
__global__ void __launch_bounds__(256) convtr_out_kernel(const float* __restrict__ x, int64_t x_band, int Tin, int C, const PF ws, int k, int s,
                                                         int pad, float* __restrict__ out, int64_t out_band, int Tout) {
    extern __shared__ float wt[];  // [k][C]
    const float* w = ws.p[blockIdx.y];
    for (int i = threadIdx.x; i < k * C; i += 256) wt[(i % k) * C + i / k] = w[i];   // w is [C][k]
    __syncthreads();
    const int tp = blockIdx.x * 256 + threadIdx.x;
    if (tp >= Tout) return;
    x += (size_t)blockIdx.y * x_band;
    float acc = 0.f;
    for (int j = (tp + pad) % s; j < k; j += s) {
        const int t = (tp + pad - j) / s;
        if (t < 0 || t >= Tin) continue;
        const float4* row = (const float4*)(x + (size_t)t * C);
        const float4* wj = (const float4*)(wt + j * C);
        for (int c = 0; c < (C >> 2); ++c) {
            const float4 xv = row[c], wv = wj[c];
            acc = fmaf(xv.x, wv.x, fmaf(xv.y, wv.y, fmaf(xv.z, wv.z, fmaf(xv.w, wv.w, acc))));
        }
    }
    out[(size_t)blockIdx.y * out_band + tp] = acc;
}

// Round 6: the same transposed convolution for k = 8, s = 4, C <= 64, by input ROWS -- and with the decoder's last GroupNorm + activation
// applied on the way in (stats != null), so that act(gn(x)) is never written to memory.  A block stages CTO_ROWS input rows (coalesced 16-byte
// loads, normalised, into LDS rows of C + 4 floats: a thread reading its own row in 16-byte steps is conflict-free at that stride), thread r
// forms the row's eight tap products p[r][j] = sum_c x[r][c] w[c][j] (weights: [C][8] in LDS, read at wave-uniform addresses), and output
// sample t s + j - pad = p[t][j] + p[t - 1][j + s] is put together from LDS: every input byte is read once (conv_out_kernel: one thread per
// output sample walking two rows from global memory, 0.33 of the HBM rate).  The first row of a block is the previous block's last (halo).
constexpr int CTO_ROWS = 64;   // rows = threads per block: 17 KB of LDS at C = 48, nine one-wave blocks per CU in different phases (256 rows: two blocks
                                // per CU marching in step, 40-43 us against the old kernel's 25 + a 24 us normalise pass)
__global__ void __launch_bounds__(CTO_ROWS) convtr_out_rows_kernel(const float* __restrict__ x, int64_t x_band, int Tin, int C, const PF ws, int pad, float* __restrict__ out,
                                                              int64_t out_band, int Tout, int G, const float* __restrict__ stats, const PF gammas, const PF betas, int act) {
    constexpr int K = 8, S = 4;
    extern __shared__ __attribute__((aligned(16))) float cto_lds[];   // x rows [CTO_ROWS][C + 4] | w [C][8] | p [CTO_ROWS][8]
    const int band = blockIdx.y, ldx = C + 4, q4 = C >> 2;
    float* xs = cto_lds;
    float* wl = xs + CTO_ROWS * ldx;
    float* pl = wl + C * K;
    const int t0 = blockIdx.x * (CTO_ROWS - 1);   // primary rows t0 .. t0 + CTO_ROWS - 2; local row i <-> input row t0 - 1 + i
    x += (size_t)band * x_band;
    // the weights as two 16-byte loads per thread, requested together with the rows below (C * K / 4 <= 2 * CTO_ROWS; a scalar loop here was six
    // dependent round trips per block: most of the 37-44 us the first versions took)
    float4 wv[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int i = threadIdx.x + CTO_ROWS * u;
        wv[u] = i < C * K / 4 ? *(const float4*)(ws.p[band] + 4 * i) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const float* gamma = gammas.p[band];
    const float* beta = betas.p[band];
    const int cg = C / G;
    // Staging: thread -> (channel quad tid % q4, row tid / q4), rows advancing by rp = CTO_ROWS / q4 per step: the quad -- and with it gamma, beta
    // and the group statistics -- is fixed per thread, so a step is four fused multiply-adds and the activation (a first version recomputed
    // (row, quad) from a running index: two integer divisions per 16 bytes made the kernel VALU-bound at 40-44 us whatever the block shape).
    // All <= 16 loads of a thread are in flight at once.
    {
        constexpr int NLD = 16;
        const int c = (threadIdx.x % q4) * 4, r0 = threadIdx.x / q4, rp = CTO_ROWS / q4;
        const bool active = r0 < rp;
        float mean[4] = {0.f, 0.f, 0.f, 0.f}, rstd[4] = {1.f, 1.f, 1.f, 1.f};
        float4 gm = make_float4(1.f, 1.f, 1.f, 1.f), bt = make_float4(0.f, 0.f, 0.f, 0.f);
        if (stats && active) {
            gm = *(const float4*)(gamma + c); bt = *(const float4*)(beta + c);
            const float* st = stats + (size_t)band * 2 * G;
#pragma unroll
            for (int e = 0; e < 4; ++e) { const int g = (c + e) / cg; mean[e] = st[2 * g]; rstd[e] = st[2 * g + 1]; }
        }
        float4 v[NLD];
        bool in[NLD];
#pragma unroll
        for (int u = 0; u < NLD; ++u) {
            const int r = r0 + rp * u, t = t0 - 1 + r;
            in[u] = active && r < CTO_ROWS && t >= 0 && t < Tin;
            v[u] = in[u] ? *(const float4*)(x + (size_t)t * C + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < NLD; ++u) {
            const int r = r0 + rp * u;
            if (!active || r >= CTO_ROWS) continue;
            if (in[u] && stats) {
                // (the expression and rounding order of norm_act_kernel)
                v[u].x = mb_act((v[u].x - mean[0]) * rstd[0] * gm.x + bt.x, act); v[u].y = mb_act((v[u].y - mean[1]) * rstd[1] * gm.y + bt.y, act);
                v[u].z = mb_act((v[u].z - mean[2]) * rstd[2] * gm.z + bt.z, act); v[u].w = mb_act((v[u].w - mean[3]) * rstd[3] * gm.w + bt.w, act);
            }
            *(float4*)(xs + r * ldx + c) = v[u];
        }
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int i = threadIdx.x + CTO_ROWS * u;
        if (i < C * K / 4) *(float4*)(wl + 4 * i) = wv[u];
    }
    __syncthreads();
    {
        float p[K];
#pragma unroll
        for (int j = 0; j < K; ++j) p[j] = 0.f;
        const float* xr = xs + threadIdx.x * ldx;
        for (int c = 0; c < C; c += 4) {
            const float4 xv = *(const float4*)(xr + c);
            const float xe[4] = {xv.x, xv.y, xv.z, xv.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float4 wa = *(const float4*)(wl + (c + e) * K), wb = *(const float4*)(wl + (c + e) * K + 4);
                p[0] = fmaf(xe[e], wa.x, p[0]); p[1] = fmaf(xe[e], wa.y, p[1]); p[2] = fmaf(xe[e], wa.z, p[2]); p[3] = fmaf(xe[e], wa.w, p[3]);
                p[4] = fmaf(xe[e], wb.x, p[4]); p[5] = fmaf(xe[e], wb.y, p[5]); p[6] = fmaf(xe[e], wb.z, p[6]); p[7] = fmaf(xe[e], wb.w, p[7]);
            }
        }
        *(float4*)(pl + threadIdx.x * K) = make_float4(p[0], p[1], p[2], p[3]);
        *(float4*)(pl + threadIdx.x * K + 4) = make_float4(p[4], p[5], p[6], p[7]);
    }
    __syncthreads();
    out += (size_t)band * out_band;
#pragma unroll
    for (int q = 0; q < S; ++q) {
        const int o = q * CTO_ROWS + threadIdx.x;          // output o of the block: primary local row 1 + o / S, phase o % S
        const int i = 1 + o / S, jj = o % S;
        if (i >= CTO_ROWS) continue;
        const int tp = (t0 - 1 + i) * S + jj - pad;   // sample index
        if (tp < 0 || tp >= Tout) continue;
        out[tp] = pl[i * K + jj] + pl[(i - 1) * K + jj + S];
    }
}

// z[t][c] = skip[t][c] (+ add[t][c])                      (decoder: z = z[:, :, :skip.T] + skip; the first z is zero or the LSTM's)
__global__ void add_kernel(const float* __restrict__ a, int64_t a_band, const float* __restrict__ b, int64_t b_band, float* __restrict__ out,
                           int64_t out_band, size_t n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x, band = blockIdx.y;
    if (i < n) out[band * out_band + i] = a[band * a_band + i] + (b ? b[band * b_band + i] : 0.f);
}
// bottleneck conditioning: z[t][c] += cond[nearest(t)][c], cond = conv_codec(latent) evaluated here (1x1 conv, codec_dim -> C):
// one block per bottleneck row
__global__ void __launch_bounds__(256) cond_add_kernel(float* __restrict__ z, int Tz, int C, const float* __restrict__ latent, int frames, int D,
                                                       const float* __restrict__ w, const float* __restrict__ bias) {
    extern __shared__ float lat[];
    const int t = blockIdx.x;
    const int f = min((int)floorf((float)t * ((float)frames / (float)Tz)), frames - 1);  // F.interpolate(mode="nearest")
    for (int d = threadIdx.x; d < D; d += 256) lat[d] = latent[(size_t)d * frames + f];
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) {
        float acc = bias[c];
        for (int d = 0; d < D; ++d) acc = fmaf(w[(size_t)c * D + d], lat[d], acc);
        z[(size_t)t * C + c] += acc;
    }
}

// DDPM step: cur = clamp((cur - est_c * estimate) * inv_sqrt_alpha + sigma * noise, -clip, clip) [* rescale on the last step]
__device__ __forceinline__ uint32_t mb_philox(uint32_t c0, uint32_t c1, uint32_t k0, uint32_t k1, uint32_t* second) {
    uint32_t x0 = c0, x1 = c1, x2 = 0x6d626421u, x3 = 0u;
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * x0, p1 = (uint64_t)0xCD9E8D57u * x2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ x1 ^ k0, n1 = (uint32_t)p1, n2 = (uint32_t)(p0 >> 32) ^ x3 ^ k1, n3 = (uint32_t)p0;
        x0 = n0; x1 = n1; x2 = n2; x3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    *second = x1;
    return x0;
}
__device__ __forceinline__ float mb_gauss(uint64_t seed, uint32_t stream, uint32_t i) {
    uint32_t b;
    const uint32_t a = mb_philox(i, stream, (uint32_t)seed, (uint32_t)(seed >> 32), &b);
    const float u1 = ((a >> 8) + 0.5f) * (1.0f / 16777216.0f), u2 = ((b >> 8) + 0.5f) * (1.0f / 16777216.0f);
    return sqrtf(-2.0f * logf(u1)) * cosf(6.283185307179586f * u2);
}
struct DdpmCoef { float est_c[MB_MAX_BANDS], inv_sqrt_alpha[MB_MAX_BANDS], sigma[MB_MAX_BANDS], noise_scale[MB_MAX_BANDS], clip[MB_MAX_BANDS], rescale[MB_MAX_BANDS]; };
__global__ void ddpm_kernel(float* __restrict__ cur, const float* __restrict__ est, int64_t est_band, const float* __restrict__ noise, int T,
                            const DdpmCoef c, uint64_t seed, uint32_t stream) {
    const int i = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
    if (i >= T) return;
    cur += (size_t)b * T;
    float v = (cur[i] - c.est_c[b] * (est[(size_t)b * est_band + i] * c.noise_scale[b])) * c.inv_sqrt_alpha[b];
    if (c.sigma[b] > 0.f) v += c.sigma[b] * (noise ? noise[(size_t)b * T + i] : mb_gauss(seed, stream + (uint32_t)b * 4096u, (uint32_t)i)) * c.noise_scale[b];
    if (c.clip[b] > 0.f) v = fminf(fmaxf(v, -c.clip[b]), c.clip[b]);
    cur[i] = v * c.rescale[b];
}
__global__ void init_noise_kernel(float* __restrict__ cur, const float* __restrict__ noise, int T, const DdpmCoef c, uint64_t seed) {
    const int i = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
    if (i < T) cur[(size_t)b * T + i] = (noise ? noise[(size_t)b * T + i] : mb_gauss(seed, (uint32_t)b * 4096u, (uint32_t)i)) * c.noise_scale[b];
}

// julius.LowPassFilters: low[f][t] = sum_j bank[f][j] x[clamp(t + j - half)]  (replicate padding).  Block = FIR_TILE outputs of
// one filter; the input tile with its halo and the taps sit in LDS.  A thread produces 4 consecutive outputs from a sliding
// register window: per tap one broadcast read of the coefficient and ONE new input sample for four MACs.
constexpr int FIR_TILE = 1024;
__global__ void __launch_bounds__(256) fir_bank_kernel(const float* __restrict__ x, int T, const float* __restrict__ bank, int taps,
                                                       float* __restrict__ low) {
    extern __shared__ __attribute__((aligned(16))) float sh[];  // [FIR_TILE + taps4 + 4] input, then [taps4] filter (zero padded)
    const int f = blockIdx.y, t0 = blockIdx.x * FIR_TILE, half = (taps - 1) / 2;
    const int taps4 = (taps + 3) & ~3, nx = FIR_TILE + taps4 + 4;
    float* xs = sh;
    float* fs = sh + nx;
    for (int i = threadIdx.x; i < nx; i += 256) xs[i] = x[min(max(t0 + i - half, 0), T - 1)];
    for (int i = threadIdx.x; i < taps4; i += 256) fs[i] = i < taps ? bank[(size_t)f * taps + i] : 0.f;
    __syncthreads();
    const int o = threadIdx.x * 4;   // outputs t0 + o .. + 3; 16-byte LDS reads at a 16-byte lane stride are conflict free
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    float4 cur = *reinterpret_cast<const float4*>(xs + o);
    for (int j = 0; j < taps4; j += 4) {
        const float4 nxt = *reinterpret_cast<const float4*>(xs + o + j + 4);
        const float4 c = *reinterpret_cast<const float4*>(fs + j);
        a0 = fmaf(c.x, cur.x, a0); a1 = fmaf(c.x, cur.y, a1); a2 = fmaf(c.x, cur.z, a2); a3 = fmaf(c.x, cur.w, a3);
        a0 = fmaf(c.y, cur.y, a0); a1 = fmaf(c.y, cur.z, a1); a2 = fmaf(c.y, cur.w, a2); a3 = fmaf(c.y, nxt.x, a3);
        a0 = fmaf(c.z, cur.z, a0); a1 = fmaf(c.z, cur.w, a1); a2 = fmaf(c.z, nxt.x, a2); a3 = fmaf(c.z, nxt.y, a3);
        a0 = fmaf(c.w, cur.w, a0); a1 = fmaf(c.w, nxt.x, a1); a2 = fmaf(c.w, nxt.y, a2); a3 = fmaf(c.w, nxt.z, a3);
        cur = nxt;
    }
    const int t = t0 + o;
    float* dst = low + (size_t)f * T + t;
    if (t + 3 < T) { dst[0] = a0; dst[1] = a1; dst[2] = a2; dst[3] = a3; }
    else { if (t < T) dst[0] = a0; if (t + 1 < T) dst[1] = a1; if (t + 2 < T) dst[2] = a2; }
}
// The same filter bank on the fp32 matrix cores (round 4): low[f][t] = sum_j bank[f][j] x[t + j - half] is a GEMM with a Toeplitz operand --
// A = the bank, 32 filter rows (zero rows beyond the real ones) x taps, B[j][t] = x[t + j - half] -- on v_mfma_f32_32x32x2_f32: exact
// fp32 products, fp32 accumulation, 64 FLOP per cycle and SIMD against the 8 of the VALU form above (which ran at 27 TFLOP/s: 4.4 ms per
// utterance for the 8-band and the two 32-band splits, 121 GFLOP).  A wave owns 32 filters x 128 outputs (four 32 x 32 accumulators): per
// pair of taps ONE coalesced 256-byte load of the transposed bank [taps][32] (L1 / L2 resident, eight pairs requested ahead) feeds four
// MFMAs whose B operands are single-dword LDS reads of the input tile (lane l: sample t + (l & 31) + (l >> 5): consecutive addresses,
// the second half-wave re-reads the first one's shifted by one: no bank conflicts).  Output columns are time: stores are 128-byte runs.
constexpr int FIRM_TILE = 512;   // outputs per workgroup: 4 waves x 128
typedef float f32x16_t __attribute__((ext_vector_type(16)));
__global__ void __launch_bounds__(256) fir_bank_mfma_kernel(const float* __restrict__ x, int T, const float* __restrict__ bank_t, int taps2, int half,
                                                            int n_filters, float* __restrict__ low) {
    extern __shared__ __attribute__((aligned(16))) float fsh[];   // [FIRM_TILE + taps2 + 2]: the input tile with its halo (replicate padding)
    const int t0 = blockIdx.x * FIRM_TILE, nx = FIRM_TILE + taps2 + 2;
    for (int i = threadIdx.x; i < nx; i += 256) fsh[i] = x[min(max(t0 + i - half, 0), T - 1)];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = lane & 31, kh = lane >> 5;
    f32x16_t acc[4];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[q][i] = 0.f;
    const float* ap = bank_t + lane;                 // pair p: bank_t[(2 p + kh) * 32 + col] = (2 p) * 32 + lane
    const float* bp = fsh + wave * 128 + col + kh;   // + 2 p (+ 32 q)
    constexpr int PF = 8;
    float areg[PF];
    const int npairs = taps2 / 2;
#pragma unroll
    for (int i = 0; i < PF; ++i) areg[i] = i < npairs ? ap[(size_t)i * 64] : 0.f;
    for (int p0 = 0; p0 < npairs; p0 += PF) {
#pragma unroll
        for (int i = 0; i < PF; ++i) {
            const int p = p0 + i;
            const float a = areg[i];
            areg[i] = p + PF < npairs ? ap[(size_t)(p + PF) * 64] : 0.f;   // the coefficient PF pairs ahead (zero past the end: its MFMAs add nothing)
            if (p < npairs) {
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bp[2 * p + 32 * q], acc[q], 0, 0, 0);
            }
        }
    }
    // C layout of the 32 x 32 tile: lane -> column l & 31; register i -> row 8 (i >> 2) + 4 (l >> 5) + (i & 3)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int t = t0 + wave * 128 + q * 32 + col;
        if (t >= T) continue;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int f = 8 * (i >> 2) + 4 * kh + (i & 3);
            if (f < n_filters) low[(size_t)f * T + t] = acc[q][i];
        }
    }
}
// bands from the low-passes in place: band[0] = low[0], band[i] = low[i] - low[i-1], band[n-1] = x - low[n-2]   (SplitBands.forward)
__global__ void bands_kernel(const float* __restrict__ x, const float* __restrict__ low, int T, int n_bands, float* __restrict__ bands) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= T) return;
    float prev = 0.f;
    for (int b = 0; b < n_bands - 1; ++b) {
        const float l = low[(size_t)b * T + t];
        bands[(size_t)b * T + t] = l - prev;
        prev = l;
    }
    bands[(size_t)(n_bands - 1) * T + t] = x[t] - prev;
}
// per-band sum and sum of squares (double), one block per band
__global__ void __launch_bounds__(1024) band_stats_kernel(const float* __restrict__ bands, int T, double* __restrict__ out) {
    __shared__ double red[2][16];
    const int b = blockIdx.x;
    double s = 0.0, s2 = 0.0;
    for (int t = threadIdx.x; t < T; t += 1024) {
        const float v = bands[(size_t)b * T + t];
        s += v;
        s2 += (double)v * v;
    }
    for (int off = 32; off > 0; off >>= 1) { s += __shfl_down(s, off, 64); s2 += __shfl_down(s2, off, 64); }
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = s; red[1][threadIdx.x >> 6] = s2; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double a = 0.0, c = 0.0;
        for (int w = 0; w < 16; ++w) { a += red[0][w]; c += red[1][w]; }
        out[2 * b] = a;
        out[2 * b + 1] = c;
    }
}
// out[t] (+)= sum_b bands[b][t] * scale[b] + shift[b]
__global__ void band_mix_kernel(const float* __restrict__ bands, int T, int n_bands, const float* __restrict__ scale, const float* __restrict__ shift,
                                float* __restrict__ out, int accumulate) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= T) return;
    float acc = accumulate ? out[t] : 0.f;
    for (int b = 0; b < n_bands; ++b) acc += bands[(size_t)b * T + t] * scale[b] + (shift ? shift[b] : 0.f);
    out[t] = acc;
}
// re_eq scales: scale[b] = std_ref[b] / std_wav[b] with the unbiased std torch.Tensor.std() uses
__global__ void eq_scale_kernel(const double* __restrict__ st_wav, const double* __restrict__ st_ref, int n_bands, double T, float* __restrict__ scale) {
    const int b = threadIdx.x;
    if (b >= n_bands) return;
    auto sd = [&](const double* st) { const double m = st[2 * b] / T; return sqrt(fmax((st[2 * b + 1] - T * m * m) / (T - 1.0), 0.0)); };
    scale[b] = (float)(sd(st_ref) / sd(st_wav));
}

struct ConvW {   // one convolution as a GEMM: bf16 hi / lo weights [N][Ktot] (tap-major K), fp32 bias
    bf16_t *hi = nullptr, *lo = nullptr;
    float* bias = nullptr;
    int N = 0, Ktot = 0;
    bool loaded_w = false, loaded_b = false, has_bias = false;
};
struct NormW { float *g = nullptr, *b = nullptr; bool lg = false, lb = false; };
struct ResW { NormW n1, n2; ConvW c1, c2; };
struct EncW { float* conv_in_w = nullptr; bool conv_in_loaded = false; ConvW conv; NormW norm; std::vector<ResW> res; float* emb = nullptr; bool emb_loaded = false; };
struct DecW { std::vector<ResW> res; NormW norm; ConvW convtr; float* convtr_out_w = nullptr; bool convtr_out_loaded = false; };
struct LstmW { std::vector<float*> w_ih, w_hh, bias; std::vector<std::vector<float>> bias_host; std::vector<int> loaded; float *lin_w = nullptr, *lin_b = nullptr; int lin_loaded = 0; };
struct BandW {
    std::vector<EncW> enc;
    std::vector<DecW> dec;
    float *codec_w = nullptr, *codec_b = nullptr;
    int codec_loaded = 0;
    LstmW lstm;
    // schedule (host): per evaluation step id, est coefficient, 1/sqrt(alpha), sigma
    std::vector<int> steps;
    std::vector<float> est_c, inv_sqrt_alpha, sigma;
    float clip = 5.f, rescale = 1.f, noise_scale = 1.f;
    // MultiBandProcessor.return_sample: band * scale + mean over proc_bands mel bands (scale empty: no processor)
    std::vector<float> proc_scale, proc_mean;
};

// device workspace of mv_mbd_generate: the same sequence of requests every call, so slot i is reused as long as it is large
// enough (about 25 hipMalloc + hipFree pairs per utterance otherwise: milliseconds, and hipFree synchronises the device)
struct Work {
    std::vector<void*> slots;
    std::vector<size_t> cap;
    size_t next = 0;
    hipStream_t stream = nullptr;
    void rewind(hipStream_t s) { next = 0; stream = s; }
    void release() { for (void* p : slots) if (p) (void)hipFree(p); slots.clear(); cap.clear(); next = 0; }
    template <typename T> hipError_t get(T** p, size_t count, bool zero) {
        const size_t bytes = count * sizeof(T) + 64;
        if (next == slots.size()) { slots.push_back(nullptr); cap.push_back(0); }
        if (cap[next] < bytes) {
            if (slots[next]) { (void)hipStreamSynchronize(stream); (void)hipFree(slots[next]); slots[next] = nullptr; cap[next] = 0; }
            hipError_t e = hipMalloc(&slots[next], bytes);
            if (e != hipSuccess) return e;
            cap[next] = bytes;
        }
        *p = (T*)slots[next++];
        return zero ? hipMemsetAsync(*p, 0, bytes, stream) : hipSuccess;
    }
};

}  // namespace

struct mv_mbd {
    mv_mbd_config cfg{};
    hipStream_t stream = nullptr;
    std::vector<void*> allocs;
    std::vector<BandW> band;
    std::vector<int> ch;                     // channels per encoder level
    float *proc_bank = nullptr, *eq_bank = nullptr;  // device FIR banks [n-1][taps]
    float *proc_bank_t = nullptr, *eq_bank_t = nullptr;  // the same banks transposed and zero padded [taps + 1][32]: the A operand of fir_bank_mfma_kernel
    int proc_taps = 0, eq_taps = 0;
    bool finalized = false;
    // LSTM scratch (bottleneck BiLSTM)
    unsigned long long* lstm_hx = nullptr;
    uint32_t* lstm_abort = nullptr;
    uint32_t lstm_serial = 0;
    Work work;   // device workspace of mv_mbd_generate, kept between calls
    int device = 0;   // the device the handle was created on: mv_mbd_generate may be called from a worker thread (whose current device is 0)
    // launches per GEMM kernel since mv_mbd_create (mv_mbd_kernel_counts: the parity tests assert WHICH kernels a shape ran on):
    // 0 t256, 1 glds3<64,192>, 2 glds3<128,128>, 3 thin, 4 glds, 5 mbd_gemm<128>, 6 mbd_gemm<64>
    mutable int64_t gemm_launches[8] = {0, 0, 0, 0, 0, 0, 0, 0};
};

namespace {

template <typename T>
int dev_alloc(mv_mbd* m, T** p, size_t count) {
    if (hipMalloc((void**)p, count * sizeof(T) + 64) != hipSuccess) return mv_fail(MV_ERR_HIP, "hipMalloc of %zu bytes failed", count * sizeof(T));
    m->allocs.push_back(*p);
    return MV_OK;
}

int upload_f32(mv_mbd* m, float** dst, const float* src, size_t n) {
    if (!*dst) { const int rc = dev_alloc(m, dst, n); if (rc != MV_OK) return rc; }
    MB_TRY(hipMemcpy(*dst, src, n * 4, hipMemcpyHostToDevice));
    return MV_OK;
}

inline bf16_t host_f2bf(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    if ((u & 0x7FFFFFFFu) > 0x7F800000u) return (bf16_t)((u >> 16) | 0x40u);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (bf16_t)(u >> 16);
}
inline float host_bf2f(bf16_t b) { uint32_t u = (uint32_t)b << 16; float f; memcpy(&f, &u, 4); return f; }

// host re-layout of a convolution weight into [N][Ktot] rows + hi / lo split, then upload
int upload_gemm_weight(mv_mbd* m, ConvW& cw, const std::vector<float>& rows, int N, int Ktot) {
    std::vector<bf16_t> hi(rows.size()), lo(rows.size());
    for (size_t i = 0; i < rows.size(); ++i) {
        hi[i] = host_f2bf(rows[i]);
        lo[i] = host_f2bf(rows[i] - host_bf2f(hi[i]));
    }
    cw.N = N; cw.Ktot = Ktot;
    if (!cw.hi) { int rc = dev_alloc(m, &cw.hi, rows.size()); if (rc != MV_OK) return rc; rc = dev_alloc(m, &cw.lo, rows.size()); if (rc != MV_OK) return rc; }
    MB_TRY(hipMemcpy(cw.hi, hi.data(), hi.size() * 2, hipMemcpyHostToDevice));
    MB_TRY(hipMemcpy(cw.lo, lo.data(), lo.size() * 2, hipMemcpyHostToDevice));
    cw.loaded_w = true;
    return MV_OK;
}

}  // namespace

extern "C" {

int mv_mbd_create(const mv_mbd_config* cfg, mv_mbd** out) {
    if (!cfg || !out) return mv_fail(MV_ERR_INVALID, "mv_mbd_create: null argument");
    const mv_mbd_config& g = *cfg;
    if (g.n_models < 1 || g.n_models > 8 || g.depth < 1 || g.depth > 8 || g.kernel != 2 * g.stride || g.stride < 2 || g.norm_groups < 1 ||
        g.norm_groups > 8 || g.res_blocks < 0 || g.res_blocks > 4 || g.chin != 1 || g.act < 0 || g.act > 2 || g.n_eval < 1 || g.n_eval > 1000 || g.precision < 1 || g.precision > 3)
        return mv_fail(MV_ERR_INVALID, "mv_mbd_create: unsupported configuration (kernel must be 2 x stride, chin 1)");
    for (int i = 0; i < g.depth; ++i)
        if (g.channels[i] < 8 || g.channels[i] % 8 || g.channels[i] % g.norm_groups)
            return mv_fail(MV_ERR_INVALID, "mv_mbd_create: channel count %d must be a multiple of 8 and of norm_groups", g.channels[i]);
    if (g.bilstm && (g.channels[g.depth - 1] % 8 || lstm_seq_lds_bytes(1, g.channels[g.depth - 1]) > 160 * 1024 - 256))
        return mv_fail(MV_ERR_INVALID, "mv_mbd_create: BiLSTM width %d not supported", g.channels[g.depth - 1]);
    mv_mbd* m = new mv_mbd();
    m->cfg = g;
    (void)hipGetDevice(&m->device);
    if (hipStreamCreateWithFlags(&m->stream, hipStreamNonBlocking) != hipSuccess) { delete m; return mv_fail(MV_ERR_HIP, "stream creation failed"); }
    m->ch.assign(g.channels, g.channels + g.depth);
    m->band.resize(g.n_models);
    for (auto& b : m->band) {
        b.enc.resize(g.depth);
        b.dec.resize(g.depth);
        for (auto& e : b.enc) e.res.resize(g.res_blocks);
        for (auto& d : b.dec) d.res.resize(g.res_blocks);
        if (g.bilstm) {
            b.lstm.w_ih.assign(4, nullptr); b.lstm.w_hh.assign(4, nullptr); b.lstm.bias.assign(4, nullptr);   // (layer, direction) = l * 2 + d
            b.lstm.bias_host.assign(4, std::vector<float>());
            b.lstm.loaded.assign(4, 0);
        }
    }
    *out = m;
    return MV_OK;
}

// A second handle on the SAME weights (device pointers shared, nothing copied) with its own stream, device workspace and LSTM scratch: the
// utterances of a batch are independent and differ in length (GroupNorm statistics span a whole utterance: no batch dimension), so a
// batching server runs them on several lanes at once -- the small-layer launches of one utterance fill the CUs another one's leave idle.
// The clone must be destroyed before its parent.
int mv_mbd_clone(mv_mbd* src, mv_mbd** out) {
    if (!src || !out) return mv_fail(MV_ERR_INVALID, "mv_mbd_clone: null argument");
    if (!src->finalized) return mv_fail(MV_ERR_STATE, "mv_mbd_clone before mv_mbd_finalize");
    mv_mbd* m = new mv_mbd();
    m->cfg = src->cfg;
    m->device = src->device;
    (void)hipSetDevice(m->device);
    m->band = src->band;   // device pointers: shared, owned by the parent (m->allocs stays empty)
    m->ch = src->ch;
    m->proc_bank = src->proc_bank; m->eq_bank = src->eq_bank; m->proc_bank_t = src->proc_bank_t; m->eq_bank_t = src->eq_bank_t;
    m->proc_taps = src->proc_taps; m->eq_taps = src->eq_taps;
    if (hipStreamCreateWithFlags(&m->stream, hipStreamNonBlocking) != hipSuccess) { delete m; return mv_fail(MV_ERR_HIP, "stream creation failed"); }
    if (m->cfg.bilstm) {   // the recurrence's hand-off granules and abort word are per handle
        const int H = m->ch[m->cfg.depth - 1];
        int rc = dev_alloc(m, &m->lstm_hx, (size_t)2 * H);
        if (rc == MV_OK) rc = dev_alloc(m, &m->lstm_abort, 1);
        if (rc != MV_OK || hipMemset(m->lstm_hx, 0, (size_t)2 * H * 8) != hipSuccess || hipMemset(m->lstm_abort, 0, 4) != hipSuccess) {
            mv_mbd_destroy(m);
            return mv_fail(MV_ERR_HIP, "mv_mbd_clone: LSTM scratch allocation failed");
        }
    }
    m->finalized = true;
    *out = m;
    return MV_OK;
}

int mv_mbd_destroy(mv_mbd* m) {
    if (!m) return MV_OK;
    if (m->stream) (void)hipStreamSynchronize(m->stream);
    m->work.release();
    for (void* p : m->allocs) (void)hipFree(p);
    if (m->stream) (void)hipStreamDestroy(m->stream);
    delete m;
    return MV_OK;
}

// Tensor names are those of audiocraft's DiffusionUnet state dict:
//   encoders.{i}.conv.weight [Cout][Cin][k] | encoders.{i}.norm.{weight,bias} | encoders.{i}.res_blocks.{r}.{norm1,norm2}.{weight,bias} |
//   encoders.{i}.res_blocks.{r}.{conv1,conv2}.{weight,bias} | decoders.{i}.res_blocks... | decoders.{i}.norm.* | decoders.{i}.convtr.weight
//   [Cin][Cout][k] | embedding.weight, embeddings.{i}.weight [num_steps][C] | conv_codec.{weight,bias} |
//   bilstm.lstm.{weight,bias}_{ih,hh}_l{0,1}[_reverse], bilstm.linear.{weight,bias}
int mv_mbd_load_weight(mv_mbd* m, int32_t band_idx, const char* name, const float* data, int64_t numel) {
    if (!m || !name || !data) return mv_fail(MV_ERR_INVALID, "mv_mbd_load_weight: null argument");
    if (band_idx < 0 || band_idx >= m->cfg.n_models) return mv_fail(MV_ERR_INVALID, "band %d out of range", band_idx);
    const mv_mbd_config& g = m->cfg;
    BandW& B = m->band[band_idx];
    const int k = g.kernel, s = g.stride;
    int i = -1, r = -1;
    char leaf[64];
    auto chan_in = [&](int lvl) { return lvl == 0 ? g.chin : m->ch[lvl - 1]; };
    auto want = [&](int64_t n) -> int { return numel == n ? MV_OK : mv_fail(MV_ERR_INVALID, "%s: expected %lld elements, got %lld", name, (long long)n, (long long)numel); };
    auto load_norm = [&](NormW& nw, bool is_w, int C) -> int {
        int rc = want(C);
        if (rc != MV_OK) return rc;
        rc = upload_f32(m, is_w ? &nw.g : &nw.b, data, C);
        if (rc == MV_OK) (is_w ? nw.lg : nw.lb) = true;
        return rc;
    };
    auto load_res = [&](ResW& rw, const char* lf, int C) -> int {
        if (!strcmp(lf, "norm1.weight")) return load_norm(rw.n1, true, C);
        if (!strcmp(lf, "norm1.bias")) return load_norm(rw.n1, false, C);
        if (!strcmp(lf, "norm2.weight")) return load_norm(rw.n2, true, C);
        if (!strcmp(lf, "norm2.bias")) return load_norm(rw.n2, false, C);
        const bool c1 = !strncmp(lf, "conv1.", 6);
        ConvW& cw = c1 ? rw.c1 : rw.c2;
        if (!strcmp(lf + 6, "weight")) {  // [C][C][3] -> [C][3 C] tap-major
            int rc = want((int64_t)C * C * 3);
            if (rc != MV_OK) return rc;
            std::vector<float> rows((size_t)C * 3 * C);
            for (int co = 0; co < C; ++co)
                for (int ci = 0; ci < C; ++ci)
                    for (int j = 0; j < 3; ++j) rows[(size_t)co * 3 * C + (size_t)j * C + ci] = data[((size_t)co * C + ci) * 3 + j];
            return upload_gemm_weight(m, cw, rows, C, 3 * C);
        }
        if (!strcmp(lf + 6, "bias")) {
            int rc = want(C);
            if (rc != MV_OK) return rc;
            rc = upload_f32(m, &cw.bias, data, C);
            if (rc == MV_OK) { cw.loaded_b = true; cw.has_bias = true; }
            return rc;
        }
        return mv_fail(MV_ERR_INVALID, "unknown tensor %s", name);
    };
    if (sscanf(name, "encoders.%d.res_blocks.%d.%63s", &i, &r, leaf) == 3) {
        if (i < 0 || i >= g.depth || r < 0 || r >= g.res_blocks) return mv_fail(MV_ERR_INVALID, "%s: index out of range", name);
        return load_res(B.enc[i].res[r], leaf, m->ch[i]);
    }
    if (sscanf(name, "decoders.%d.res_blocks.%d.%63s", &i, &r, leaf) == 3) {
        if (i < 0 || i >= g.depth || r < 0 || r >= g.res_blocks) return mv_fail(MV_ERR_INVALID, "%s: index out of range", name);
        return load_res(B.dec[i].res[r], leaf, m->ch[g.depth - 1 - i]);  // decoders run from the bottleneck outwards
    }
    if (sscanf(name, "encoders.%d.%63s", &i, leaf) == 2) {
        if (i < 0 || i >= g.depth) return mv_fail(MV_ERR_INVALID, "%s: index out of range", name);
        EncW& e = B.enc[i];
        const int Cin = chan_in(i), Cout = m->ch[i];
        if (!strcmp(leaf, "norm.weight")) return load_norm(e.norm, true, Cout);
        if (!strcmp(leaf, "norm.bias")) return load_norm(e.norm, false, Cout);
        if (!strcmp(leaf, "conv.weight")) {
            int rc = want((int64_t)Cout * Cin * k);
            if (rc != MV_OK) return rc;
            if (i == 0) { rc = upload_f32(m, &e.conv_in_w, data, (size_t)numel); if (rc == MV_OK) e.conv_in_loaded = true; return rc; }
            std::vector<float> rows((size_t)Cout * k * Cin);
            for (int co = 0; co < Cout; ++co)
                for (int ci = 0; ci < Cin; ++ci)
                    for (int j = 0; j < k; ++j) rows[(size_t)co * k * Cin + (size_t)j * Cin + ci] = data[((size_t)co * Cin + ci) * k + j];
            return upload_gemm_weight(m, e.conv, rows, Cout, k * Cin);
        }
        return mv_fail(MV_ERR_INVALID, "unknown tensor %s", name);
    }
    if (sscanf(name, "decoders.%d.%63s", &i, leaf) == 2) {
        if (i < 0 || i >= g.depth) return mv_fail(MV_ERR_INVALID, "%s: index out of range", name);
        DecW& d = B.dec[i];
        const int lvl = g.depth - 1 - i, Cin = m->ch[lvl], Cout = chan_in(lvl);
        if (!strcmp(leaf, "norm.weight")) return load_norm(d.norm, true, Cin);
        if (!strcmp(leaf, "norm.bias")) return load_norm(d.norm, false, Cin);
        if (!strcmp(leaf, "convtr.weight")) {  // [Cin][Cout][k]
            int rc = want((int64_t)Cin * Cout * k);
            if (rc != MV_OK) return rc;
            if (lvl == 0) { rc = upload_f32(m, &d.convtr_out_w, data, (size_t)numel); if (rc == MV_OK) d.convtr_out_loaded = true; return rc; }
            // GEMM row (p, co), K = [tap x[q-1] : kernel index p + s | tap x[q] : kernel index p]
            std::vector<float> rows((size_t)s * Cout * 2 * Cin);
            for (int p = 0; p < s; ++p)
                for (int co = 0; co < Cout; ++co)
                    for (int ci = 0; ci < Cin; ++ci) {
                        const size_t row = (size_t)p * Cout + co;
                        rows[row * 2 * Cin + ci] = data[((size_t)ci * Cout + co) * k + p + s];
                        rows[row * 2 * Cin + Cin + ci] = data[((size_t)ci * Cout + co) * k + p];
                    }
            return upload_gemm_weight(m, d.convtr, rows, s * Cout, 2 * Cin);
        }
        return mv_fail(MV_ERR_INVALID, "unknown tensor %s", name);
    }
    if (!strcmp(name, "embedding.weight") || sscanf(name, "embeddings.%d.weight", &i) == 1) {
        const int lvl = !strcmp(name, "embedding.weight") ? 0 : i + 1;
        if (lvl < 0 || lvl >= g.depth) return mv_fail(MV_ERR_INVALID, "%s: level out of range", name);
        int rc = want((int64_t)g.num_steps * m->ch[lvl]);
        if (rc != MV_OK) return rc;
        rc = upload_f32(m, &B.enc[lvl].emb, data, (size_t)numel);
        if (rc == MV_OK) B.enc[lvl].emb_loaded = true;
        return rc;
    }
    if (!strcmp(name, "conv_codec.weight") || !strcmp(name, "conv_codec.bias")) {
        if (g.codec_dim <= 0) return MV_OK;
        const bool w = name[11] == 'w';
        const int Cb = m->ch[g.depth - 1];
        int rc = want(w ? (int64_t)Cb * g.codec_dim : Cb);
        if (rc != MV_OK) return rc;
        rc = upload_f32(m, w ? &B.codec_w : &B.codec_b, data, (size_t)numel);
        if (rc == MV_OK) B.codec_loaded |= w ? 1 : 2;
        return rc;
    }
    if (!strncmp(name, "bilstm.", 7)) {
        if (!g.bilstm) return mv_fail(MV_ERR_INVALID, "%s: this configuration has no BiLSTM", name);
        const int H = m->ch[g.depth - 1];
        char kind[8], lf[4], rev[16] = "";
        int l = -1;
        if (!strcmp(name, "bilstm.linear.weight")) { int rc = want((int64_t)H * 2 * H); if (rc == MV_OK) rc = upload_f32(m, &B.lstm.lin_w, data, (size_t)numel); if (rc == MV_OK) B.lstm.lin_loaded |= 1; return rc; }
        if (!strcmp(name, "bilstm.linear.bias")) { int rc = want(H); if (rc == MV_OK) rc = upload_f32(m, &B.lstm.lin_b, data, (size_t)numel); if (rc == MV_OK) B.lstm.lin_loaded |= 2; return rc; }
        const int nf = sscanf(name, "bilstm.lstm.%7[a-z]_%2[a-z]_l%d%15s", kind, lf, &l, rev);
        if (nf < 3 || l < 0 || l > 1) return mv_fail(MV_ERR_INVALID, "unknown tensor %s", name);
        const int d = (nf == 4 && !strcmp(rev, "_reverse")) ? 1 : 0, slot = l * 2 + d;
        const bool ih = !strcmp(lf, "ih");
        const int I = l == 0 ? H : 2 * H;
        if (!strcmp(kind, "weight")) {
            int rc = want((int64_t)4 * H * (ih ? I : H));
            if (rc != MV_OK) return rc;
            rc = upload_f32(m, ih ? &B.lstm.w_ih[slot] : &B.lstm.w_hh[slot], data, (size_t)numel);
            if (rc == MV_OK) B.lstm.loaded[slot] |= ih ? 1 : 2;
            return rc;
        }
        int rc = want(4 * H);
        if (rc != MV_OK) return rc;
        if (B.lstm.bias_host[slot].empty()) B.lstm.bias_host[slot].assign((size_t)4 * H, 0.f);
        for (int q = 0; q < 4 * H; ++q) B.lstm.bias_host[slot][q] += data[q];
        B.lstm.loaded[slot] |= ih ? 4 : 8;
        return MV_OK;
    }
    return mv_fail(MV_ERR_INVALID, "unknown multi-band-diffusion tensor %s", name);
}

int mv_mbd_set_schedule(mv_mbd* m, int32_t band_idx, const int32_t* steps, const float* est_c, const float* inv_sqrt_alpha, const float* sigma,
                        float clip, float rescale, float noise_scale) {
    if (!m || band_idx < 0 || band_idx >= m->cfg.n_models || !steps || !est_c || !inv_sqrt_alpha || !sigma) return mv_fail(MV_ERR_INVALID, "mv_mbd_set_schedule: bad argument");
    BandW& B = m->band[band_idx];
    const int n = m->cfg.n_eval;
    B.steps.assign(steps, steps + n); B.est_c.assign(est_c, est_c + n); B.inv_sqrt_alpha.assign(inv_sqrt_alpha, inv_sqrt_alpha + n); B.sigma.assign(sigma, sigma + n);
    for (int i = 0; i < n; ++i)
        if (steps[i] < 0 || steps[i] >= m->cfg.num_steps) return mv_fail(MV_ERR_INVALID, "mv_mbd_set_schedule: step %d out of range", steps[i]);
    B.clip = clip; B.rescale = rescale; B.noise_scale = noise_scale;
    return MV_OK;
}

int mv_mbd_set_processor(mv_mbd* m, int32_t band_idx, int32_t n_bands, const float* scale, const float* mean) {
    if (!m || band_idx < 0 || band_idx >= m->cfg.n_models || n_bands != m->cfg.proc_bands || !scale || !mean) return mv_fail(MV_ERR_INVALID, "mv_mbd_set_processor: bad argument");
    m->band[band_idx].proc_scale.assign(scale, scale + n_bands);
    m->band[band_idx].proc_mean.assign(mean, mean + n_bands);
    return MV_OK;
}

int mv_mbd_set_filterbank(mv_mbd* m, int32_t which, int32_t n_filters, int32_t taps, const float* coeffs) {
    if (!m || !coeffs || taps < 1 || taps % 2 == 0 || taps > 7000)  /* (1024 + 2 taps) floats of LDS <= 64 KB */ return mv_fail(MV_ERR_INVALID, "mv_mbd_set_filterbank: bad argument");
    const int want = (which == 0 ? m->cfg.proc_bands : m->cfg.eq_bands) - 1;
    if (n_filters != want) return mv_fail(MV_ERR_INVALID, "mv_mbd_set_filterbank: expected %d filters, got %d", want, n_filters);
    float** dst = which == 0 ? &m->proc_bank : &m->eq_bank;
    *dst = nullptr;
    const int rc = upload_f32(m, dst, coeffs, (size_t)n_filters * taps);
    if (rc != MV_OK) return rc;
    (which == 0 ? m->proc_taps : m->eq_taps) = taps;
    if (n_filters <= 32) {   // [taps + 1][32], zero rows / columns beyond the real filters and taps
        std::vector<float> t((size_t)(taps + 1) * 32, 0.f);
        for (int f = 0; f < n_filters; ++f)
            for (int j = 0; j < taps; ++j) t[(size_t)j * 32 + f] = coeffs[(size_t)f * taps + j];
        float** dt = which == 0 ? &m->proc_bank_t : &m->eq_bank_t;
        *dt = nullptr;
        const int rt = upload_f32(m, dt, t.data(), t.size());
        if (rt != MV_OK) return rt;
    }
    return MV_OK;
}

int mv_mbd_finalize(mv_mbd* m) {
    if (!m) return mv_fail(MV_ERR_INVALID, "mv_mbd_finalize: null");
    const mv_mbd_config& g = m->cfg;
    for (int bi = 0; bi < g.n_models; ++bi) {
        BandW& B = m->band[bi];
        auto norm_ok = [](const NormW& n) { return n.lg && n.lb; };
        auto res_ok = [&](const ResW& r) { return norm_ok(r.n1) && norm_ok(r.n2) && r.c1.loaded_w && r.c1.loaded_b && r.c2.loaded_w && r.c2.loaded_b; };
        for (int i = 0; i < g.depth; ++i) {
            const EncW& e = B.enc[i];
            bool ok = (i == 0 ? e.conv_in_loaded : e.conv.loaded_w) && norm_ok(e.norm);
            for (const auto& r : e.res) ok = ok && res_ok(r);
            if (i == 0 || g.emb_all_layers) ok = ok && e.emb_loaded;
            const DecW& d = B.dec[i];
            const int lvl = g.depth - 1 - i;
            ok = ok && (lvl == 0 ? d.convtr_out_loaded : d.convtr.loaded_w) && norm_ok(d.norm);
            for (const auto& r : d.res) ok = ok && res_ok(r);
            if (!ok) return mv_fail(MV_ERR_MISSING_WEIGHT, "multi-band diffusion: band %d level %d has missing tensors", bi, i);
        }
        if (g.codec_dim > 0 && g.bilstm && B.codec_loaded != 3) return mv_fail(MV_ERR_MISSING_WEIGHT, "band %d: conv_codec missing", bi);
        if (g.bilstm) {
            const int H = m->ch[g.depth - 1];
            for (int sl = 0; sl < 4; ++sl) {
                if (B.lstm.loaded[sl] != 15) return mv_fail(MV_ERR_MISSING_WEIGHT, "band %d: BiLSTM tensors incomplete", bi);
                B.lstm.bias[sl] = nullptr;
                const int rc = upload_f32(m, &B.lstm.bias[sl], B.lstm.bias_host[sl].data(), (size_t)4 * H);
                if (rc != MV_OK) return rc;
            }
            if (B.lstm.lin_loaded != 3) return mv_fail(MV_ERR_MISSING_WEIGHT, "band %d: bilstm.linear missing", bi);
        }
        if ((int)B.steps.size() != g.n_eval) return mv_fail(MV_ERR_STATE, "band %d: mv_mbd_set_schedule not called", bi);
        if (g.proc_bands > 1 && (int)B.proc_scale.size() != g.proc_bands) return mv_fail(MV_ERR_STATE, "band %d: mv_mbd_set_processor not called", bi);
    }
    if (g.proc_bands > 1 && !m->proc_bank) return mv_fail(MV_ERR_STATE, "processor filter bank not set");
    if (g.eq_bands > 1 && !m->eq_bank) return mv_fail(MV_ERR_STATE, "re-equalisation filter bank not set");
    if (g.bilstm) {
        const int H = m->ch[g.depth - 1];
        int rc = dev_alloc(m, &m->lstm_hx, (size_t)2 * H);
        if (rc != MV_OK) return rc;
        MB_TRY(hipMemset(m->lstm_hx, 0, (size_t)2 * H * 8));
        rc = dev_alloc(m, &m->lstm_abort, 1);
        if (rc != MV_OK) return rc;
        MB_TRY(hipMemset(m->lstm_abort, 0, 4));
    }
    m->finalized = true;
    return MV_OK;
}

}  // extern "C"

namespace {

// everything the 20 evaluations need, allocated once per mv_mbd_generate call.  The band models have identical shapes and are
// independent, so every launch covers ALL of them (grid.z / grid.y = band): four times the blocks per launch -- the 3072-channel
// GEMMs alone had 456 -- and a quarter of the launches.  Buffers are [band][...] with the strides below.
struct UnetBuffers {
    int NB = 1;
    std::vector<int> T;                    // length at every encoder level (after the strided conv)
    std::vector<int> Tpad;                 // input length of level i padded to a multiple of the stride
    std::vector<float*> skip;              // encoder outputs [band][T_i][C_i]
    std::vector<size_t> skip_band;
    float *a32 = nullptr, *b32 = nullptr, *c32 = nullptr;   // fp32 scratch [band][widest]
    bf16_t *hi = nullptr, *lo = nullptr;   // bf16 conv operand with zero margins [band][op_band]
    // the strided convolution's operand of encoder level i >= 1, written by the epilogue that produces skip[i - 1]; zeroed once per call (only
    // body rows are ever written: the margins stay zero)
    std::vector<bf16_t*> hi_lvl, lo_lvl;
    double* part = nullptr;                // [band][part_band]
    double* part2 = nullptr;               // the records a normalise pass writes while it may still be finishing the ones in `part`
    float* stats = nullptr;                // [band][2 G]
    // statistics that no gn_final_kernel has finished: the next norm_act_kernel does it from these records (stats_of_gemm_output leaves them
    // when they are few); null: `stats` is current
    const double* pending_part = nullptr;
    int pending_rec = 0;
    double pending_count = 0.0;
    float* dec_out = nullptr;              // transposed-conv output [band][op_band]
    size_t widest = 0, op_band = 0;
};

template <typename F> PF pf_of(int nb, F f) { PF r{}; for (int b = 0; b < nb; ++b) r.p[b] = f(b); return r; }
template <typename F> PH ph_of(int nb, F f) { PH r{}; for (int b = 0; b < nb; ++b) r.p[b] = f(b); return r; }

// cw(b): the band's ConvW.  taps > 1 only when the taps are not contiguous in memory (dilation).
// GroupNorm statistics of a GEMM's output from its epilogue: `G` groups over the N output channels, partials into `part`; launch_gemm
// sets `tiles` to the number of partial records per band it will have written, or to 0 when the kernel it picked does not do it (the
// caller then runs the two-pass group_stats on the output).
struct GnFuse {
    int G = 0;
    double* part = nullptr;
    int tiles = 0;
    int period = 0;   // 0: the groups run over the GEMM's N columns; Cout for a transposed convolution (N = stride x Cout, MbdGemmArgs::gn_period)
};
// MV_MBD_NO_GN_FUSE=1: the round-2 scheme -- gn_partial_kernel reads every tensor again (tests compare both ways in one process: read per call)
inline bool gn_fuse_enabled() { return !getenv("MV_MBD_NO_GN_FUSE"); }
// the GEMM's output wanted as the next convolution's operand too (MbdGemmArgs::op_hi): `written` comes back false when the kernel that ran has
// no such epilogue (the caller then runs the split pass)
struct OpOut {
    bf16_t* hi = nullptr;
    bf16_t* lo = nullptr;
    int64_t band = 0, shift = 0;
    bool written = false;
};
template <typename CW>
hipError_t launch_gemm(hipStream_t s, const mv_mbd* m, int nb, MbdGemmArgs& g, const bf16_t* a_hi, const bf16_t* a_lo, CW cw, int taps, int64_t tap_stride,
                       int k_per_tap, GnFuse* gf = nullptr, OpOut* op = nullptr) {
    g.op_hi = g.op_lo = nullptr;
    if (op) op->written = false;
    auto arm_op = [&]() { if (op && op->hi) { g.op_hi = op->hi; g.op_lo = op->lo; g.op_band = op->band; g.op_shift = op->shift; op->written = true; } };
    const bool gn_fused = gn_fuse_enabled();
    const int gn_period = gf && gf->period ? gf->period : g.N;
    const bool gn_any = gf && gn_fused && gf->G > 0 && gf->G <= 8 && g.N % gn_period == 0 && gn_period % gf->G == 0 && (gn_period / gf->G) % 4 == 0;
    const bool gn_ok = gn_any;   // (until round 6 the tiled kernels needed group widths that are multiples of 16)
    g.gn_part = nullptr;
    if (gf) gf->tiles = 0;
    auto arm_gn = [&](int tiles) { g.gn_part = gf->part; g.gn_G = gf->G; g.gn_period = gn_period; g.gn_cg = gn_period / gf->G; gf->tiles = tiles; };
    g.n_seg = 0;
    g.ldw = cw(0).Ktot;
    g.w_hi = ph_of(nb, [&](int b) { return (const bf16_t*)cw(b).hi; });
    g.w_lo = ph_of(nb, [&](int b) { return (const bf16_t*)cw(b).lo; });
    const int prec = m->cfg.precision;
    for (int term = 0; term < prec; ++term) {      // a_hi w_hi, a_hi w_lo, a_lo w_hi
        const bf16_t* ab = term == 2 ? a_lo : a_hi;
        for (int t = 0; t < taps; ++t) {
            if (g.n_seg >= MB_MAX_SEG) return hipErrorInvalidValue;
            g.seg[g.n_seg++] = GemmSeg{ab + t * tap_stride, term == 1 ? 1 : 0, (int64_t)t * k_per_tap, k_per_tap};
        }
    }
    const int nt = (g.N + BN - 1) / BN;
    bool k64 = true;
    for (int i = 0; i < g.n_seg; ++i) k64 = k64 && g.seg[i].K % 64 == 0;
    static std::atomic<bool> t256_raised_on[64];   // per device
    int t256_dev = 0;
    (void)hipGetDevice(&t256_dev);
    std::atomic<bool>& t256_raised = t256_raised_on[t256_dev & 63];
    // 256 x 256 tiles once such tiles fill at least ~3/4 of the CUs (the 768- and 3072-channel layers with all bands in one launch)
    // tile height 256 or 240: whichever needs less (rounds of 256 workgroups) x rows
    const int nt_n256 = (g.N + 255) / 256;
    auto t256_cost = [&](int rows) { const size_t blocks = (size_t)((g.M + rows - 1) / rows) * nt_n256 * nb; return (blocks + 255) / 256 * (size_t)rows; };
    const int t256_rows = getenv("MV_MBD_T256_ROWS") ? atoi(getenv("MV_MBD_T256_ROWS")) : t256_cost(240) < t256_cost(256) ? 240 : 256;
    const int nt_m256 = (g.M + t256_rows - 1) / t256_rows;
    const char* t256_min_env = getenv("MV_MBD_T256_MIN_BLOCKS");   // tests lower it so that small shapes (ragged tiles) take this kernel
    const size_t t256_min = t256_min_env ? (size_t)atoi(t256_min_env) : 192;
    if (k64 && prec == 3 && taps == 1 && g.N >= 256 && (size_t)nt_m256 * nt_n256 * nb >= t256_min) {
        if (!t256_raised) {
            hipError_t e = hipFuncSetAttribute((const void*)mbd_gemm_t256_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, T256_LDS_BYTES);
            if (e != hipSuccess) return e;
            t256_raised = true;
        }
        const int n_blocks = nt_m256 * nt_n256 * nb;
        if (gn_ok) arm_gn(nt_m256 * nt_n256);
        arm_op();
        ++m->gemm_launches[0];
        hipLaunchKernelGGL(mbd_gemm_t256_kernel, dim3(n_blocks), dim3(512), T256_LDS_BYTES, s, g, nt_m256, nt_n256, n_blocks, t256_rows);
    } else if (k64 && prec == 3 && taps == 1 && g.N % 192 == 0 && g.N % 128 != 0 && !getenv("MV_MBD_NO_N192") &&
               (size_t)(g.N / 192) * ((g.M + 63) / 64) * nb >= (getenv("MV_MBD_N192_MIN_BLOCKS") ? (size_t)atoi(getenv("MV_MBD_N192_MIN_BLOCKS")) : 400)) {
        // 192-channel layers: 64 x 192 tiles, no padding columns (MV_MBD_N192_MIN_BLOCKS: tests lower the threshold for small shapes)
        const int nt192 = g.N / 192, mt64 = (g.M + 63) / 64;
        if (gn_ok) arm_gn(nt192 * mt64);
        arm_op();
        ++m->gemm_launches[1];
        hipLaunchKernelGGL((mbd_gemm_glds3_kernel<64, 192>), dim3(nt192, mt64, nb), dim3(256), 0, s, g);
    } else if (k64 && prec == 3 && taps == 1 && (size_t)nt * ((g.M + 127) / 128) * nb >= 400) {
        if (gn_ok) arm_gn(nt * ((g.M + 127) / 128));
        arm_op();
        ++m->gemm_launches[2];
        hipLaunchKernelGGL((mbd_gemm_glds3_kernel<128, 128>), dim3(nt, (g.M + 127) / 128, nb), dim3(256), 0, s, g);
    }
    else if (prec == 3 && taps == 1 && g.N <= 64 && g.N % 16 == 0 && g.seg[0].K <= 160 && g.seg[0].K % 8 == 0 && !getenv("MV_MBD_NO_THIN")) {
        // thin layers: the whole W in registers, A fragments straight from global memory (MV_MBD_NO_THIN=1: the 128-column tiles; tests, A/B)
        auto thin_cost = [&](int mt) { const size_t blocks = (size_t)((g.M + 64 * mt - 1) / (64 * mt)) * nb; return (blocks + 511) / 512 * (size_t)(mt + 1); };
        const int mt = getenv("MV_MBD_THIN_MT") ? atoi(getenv("MV_MBD_THIN_MT")) : thin_cost(10) < thin_cost(4) ? 10 : 4;
        const dim3 grid((unsigned)((g.M + 64 * mt - 1) / (64 * mt)), nb);
        ++m->gemm_launches[3];
        arm_op();
        if (gn_any) arm_gn((int)grid.x);   // one record per block; any group width (a lane's column has its own group)
        const int gm = !gn_any ? 0 : gf->G <= 4 ? 4 : 8;
#define MBD_THIN_MT(NTN, MT) do { if (gm == 0) hipLaunchKernelGGL((mbd_gemm_thin_kernel<NTN, 0, MT>), grid, dim3(256), 0, s, g); \
                                  else if (gm == 4) hipLaunchKernelGGL((mbd_gemm_thin_kernel<NTN, 4, MT>), grid, dim3(256), 0, s, g); \
                                  else hipLaunchKernelGGL((mbd_gemm_thin_kernel<NTN, 8, MT>), grid, dim3(256), 0, s, g); } while (0)
#define MBD_THIN(NTN) do { if (mt == 10) MBD_THIN_MT(NTN, 10); else MBD_THIN_MT(NTN, 4); } while (0)
        switch (g.N / 16) {
            case 1: MBD_THIN(1); break;
            case 2: MBD_THIN(2); break;
            case 3: MBD_THIN(3); break;
            default: MBD_THIN(4); break;
        }
#undef MBD_THIN
#undef MBD_THIN_MT
    }
    else if (k64 && (size_t)nt * ((g.M + 127) / 128) * nb >= 400) {
        ++m->gemm_launches[4];
        hipLaunchKernelGGL(mbd_gemm_glds_kernel, dim3(nt, (g.M + 127) / 128, nb), dim3(256), 0, s, g);
    } else if ((size_t)nt * ((g.M + 127) / 128) * nb >= 400) {
        ++m->gemm_launches[5];
        hipLaunchKernelGGL(mbd_gemm_kernel<128>, dim3(nt, (g.M + 127) / 128, nb), dim3(256), 0, s, g);
    } else {
        ++m->gemm_launches[6];
        hipLaunchKernelGGL(mbd_gemm_kernel<64>, dim3(nt, (g.M + 63) / 64, nb), dim3(256), 0, s, g);
    }
    return hipGetLastError();
}

// 8-element units per block of norm_act_kernel (8 K elements: two trips of 256 threads x 2 units).  Per launch under rocprofv3 on one box
// (profiles/r06z8/norm_upb_sweep.txt): 256: 28.7 us, 512: 26.0, 1024: 27.0, 2048: 28.5, 4096: 35.7; 1024 keeps the records of a pass with
// statistics at 450 per band, under GN_DEFER_MAX_RECORDS (at 512 units a gn_final_kernel launch comes back: the stage 120.4 against 119.8 ms)
constexpr int NORM_UPB = 1024;
// records up to which the consuming normalise pass finishes the statistics itself (a record is 2 G doubles: 32 KB of L2 reads per block at
// G = 4).  64 (the 256 x 256-tile GEMMs only) and 512 (every producer of the bench: 300 tiles of the 64 x 192 and thin GEMMs, 450 blocks of a
// normalise pass) measured 116.7 and 116.4 ms for the stage (profiles/r06z8): the pass gets 1.7 us longer, a 4.8 us launch goes away
constexpr int GN_DEFER_MAX_RECORDS = 512;
size_t gn_blocks(int T, int C, int G) {   // partial records x G of a [T][C] tensor: the two-pass kernels, or the tiles of a GEMM epilogue
    const int rows_per = std::max(1, (int)((size_t)16384 * G / C));
    const size_t two_pass = (size_t)((T + rows_per - 1) / rows_per), tiles = (size_t)((T + 127) / 128) * ((C + BN - 1) / BN),
                 tiles192 = (size_t)((T + 63) / 64) * ((C + 191) / 192);   // the 64 x 192 tiles of the 192-channel layers
    const size_t norm = ((size_t)T * C / 8 + NORM_UPB - 1) / NORM_UPB;   // norm_act_kernel with output statistics: one record per block
    return std::max(std::max(std::max(two_pass, tiles), tiles192), norm) * G;
}
hipError_t group_stats(hipStream_t s, const float* x, int64_t x_band, int T, int C, int G, UnetBuffers& ub) {
    // ~16 K elements per block whatever the [T][C] aspect ratio: rows_per rows of one group
    const int rows_per = std::max(1, (int)((size_t)16384 * G / C));
    const int chunks = (T + rows_per - 1) / rows_per;
    hipLaunchKernelGGL(gn_partial_kernel, dim3(chunks, G, ub.NB), dim3(256), 0, s, x, x_band, T, C, G, rows_per, ub.part);
    hipLaunchKernelGGL(gn_final_kernel, dim3(ub.NB), dim3(256), 0, s, ub.part, chunks, G, (double)T * (C / G), 1e-5f, ub.stats);
    ub.pending_part = nullptr;
    return hipGetLastError();
}

// z = a + b written to `out`, and the GroupNorm statistics of z, in ONE pass (the decoder's skip addition behind a BiLSTM bottleneck)
hipError_t add_and_group_stats(hipStream_t s, const float* a, int64_t a_band, const float* b, int64_t b_band, float* out, int64_t out_band, int T, int C, int G,
                               UnetBuffers& ub) {
    const int rows_per = std::max(1, (int)((size_t)16384 * G / C));
    const int chunks = (T + rows_per - 1) / rows_per;
    hipLaunchKernelGGL(gn_partial_kernel, dim3(chunks, G, ub.NB), dim3(256), 0, s, a, a_band, T, C, G, rows_per, ub.part, b, b_band, out, out_band);
    hipLaunchKernelGGL(gn_final_kernel, dim3(ub.NB), dim3(256), 0, s, ub.part, chunks, G, (double)T * (C / G), 1e-5f, ub.stats);
    ub.pending_part = nullptr;
    return hipGetLastError();
}

// the statistics of a tensor some kernel has produced: from the partial records it left (gf.tiles per band), or by the two passes over the tensor
hipError_t stats_of_gemm_output(hipStream_t s, const GnFuse& gf, const float* x, int64_t x_band, int T, int C, int G, UnetBuffers& ub) {
    if (gf.tiles == 0) return group_stats(s, x, x_band, T, C, G, ub);
    if (gf.tiles <= GN_DEFER_MAX_RECORDS && gn_fuse_enabled()) {   // few records: the normalise pass that needs them finishes them itself
        ub.pending_part = gf.part; ub.pending_rec = gf.tiles; ub.pending_count = (double)T * (C / G);
        return hipSuccess;
    }
    hipLaunchKernelGGL(gn_final_kernel, dim3(ub.NB), dim3(256), 0, s, gf.part, gf.tiles, G, (double)T * (C / G), 1e-5f, ub.stats);
    ub.pending_part = nullptr;
    return hipGetLastError();
}

// statistics left pending for a normalise pass (stats_of_gemm_output), wanted by something else: gn_final_kernel after all
hipError_t finish_pending_stats(hipStream_t s, UnetBuffers& ub, int G) {
    if (!ub.pending_part) return hipSuccess;
    hipLaunchKernelGGL(gn_final_kernel, dim3(ub.NB), dim3(256), 0, s, ub.pending_part, ub.pending_rec, G, ub.pending_count, 1e-5f, ub.stats);
    ub.pending_part = nullptr;
    return hipGetLastError();
}

// out_gf (optional): also leave the GroupNorm partials of the OUTPUT y (tiles = 0 when the groups are not float4-aligned or the fusion is
// switched off: the caller then runs the two-pass statistics on y32)
void launch_norm_act(hipStream_t s, UnetBuffers& ub, const float* x, int64_t x_band, int T, int C, int G, const float* stats, PF gamma, PF beta, int act,
                     float* y32, int64_t y_band, bf16_t* hi, bf16_t* lo, int front, int back, GnFuse* out_gf = nullptr) {
    const size_t units = (size_t)(front + T + back) * C / 8;
    const dim3 grid((unsigned)((units + NORM_UPB - 1) / NORM_UPB), ub.NB);
    const bool fuse = out_gf && gn_fuse_enabled() && ((C / G) & 3) == 0 && G <= 8 && front == 0 && back == 0;
    if (out_gf) out_gf->tiles = fuse ? (int)grid.x : 0;
    double* part = fuse ? out_gf->part : nullptr;
    // the statistics of x: `stats`, or the records stats_of_gemm_output left pending (consumed here).  A pass that writes its own records into
    // the buffer the pending ones live in cannot finish them itself (early blocks would overwrite what late ones still read): gn_final_kernel then
    if (stats && ub.pending_part && part == ub.pending_part) (void)finish_pending_stats(s, ub, G);
    const double* fin_part = stats ? ub.pending_part : nullptr;
    const int fin_rec = fin_part ? ub.pending_rec : 0;
    const double fin_count = fin_part ? ub.pending_count : 0.0;
    if (stats) ub.pending_part = nullptr;
    if (!fuse)
        hipLaunchKernelGGL(norm_act_kernel<0>, grid, dim3(256), 0, s, x, x_band, T, C, G, stats, gamma, beta, act, y32, y_band, hi, lo, (int64_t)ub.op_band, front, back,
                           NORM_UPB, part, fin_part, fin_rec, fin_count);
    else if (G <= 4)
        hipLaunchKernelGGL(norm_act_kernel<4>, grid, dim3(256), 0, s, x, x_band, T, C, G, stats, gamma, beta, act, y32, y_band, hi, lo, (int64_t)ub.op_band, front, back,
                           NORM_UPB, part, fin_part, fin_rec, fin_count);
    else
        hipLaunchKernelGGL(norm_act_kernel<8>, grid, dim3(256), 0, s, x, x_band, T, C, G, stats, gamma, beta, act, y32, y_band, hi, lo, (int64_t)ub.op_band, front, back,
                           NORM_UPB, part, fin_part, fin_rec, fin_count);
}

// x [band][T][C] fp32 -> ResBlock(x) into `out` (may alias x): x + conv2(act(gn2(conv1(act(gn1(x))))))  (+ col_add); rw(b) = band b's weights
// out_gf (optional): also leave the GroupNorm partials of the block's OUTPUT (stats_of_gemm_output turns them into statistics);
// in_gf (optional): the partials of the block's INPUT, left by the GEMM that produced it (the previous ResBlock's conv2, round 4: the
// two-pass statistics of that tensor -- one more read of 59 MB per ResBlock -- are gone)
template <typename RW>
hipError_t res_block(hipStream_t s, const mv_mbd* m, RW rw, const float* x, int64_t x_band, float* out, int64_t out_band, int T, int C, int dilation, PF col_add,
                     UnetBuffers& ub, GnFuse* out_gf = nullptr, const GnFuse* in_gf = nullptr, bool in_stats_ready = false, OpOut* out_op = nullptr) {
    const int G = m->cfg.norm_groups, act = m->cfg.act, nb = ub.NB;
    const int pad = dilation;  // kernel 3
    hipError_t e;
    GnFuse mid{G, ub.part, 0};   // conv1's output is only ever read through its GroupNorm: statistics from conv1's epilogue
    for (int half = 0; half < 2; ++half) {
        const float* in = half == 0 ? x : ub.b32;
        const int64_t in_band = half == 0 ? x_band : (int64_t)ub.widest;
        if ((e = half == 0 ? (in_stats_ready ? hipSuccess : in_gf ? stats_of_gemm_output(s, *in_gf, in, in_band, T, C, G, ub) : group_stats(s, in, in_band, T, C, G, ub))
                           : stats_of_gemm_output(s, mid, in, in_band, T, C, G, ub)) != hipSuccess) return e;
        launch_norm_act(s, ub, in, in_band, T, C, G, ub.stats, pf_of(nb, [&](int b) { return (const float*)(half == 0 ? rw(b).n1.g : rw(b).n2.g); }),
                        pf_of(nb, [&](int b) { return (const float*)(half == 0 ? rw(b).n1.b : rw(b).n2.b); }), act, nullptr, 0, ub.hi,
                        m->cfg.precision > 2 ? ub.lo : nullptr, pad, pad);
        MbdGemmArgs g;
        g.M = T; g.N = C; g.lda = C; g.a_band = (int64_t)ub.op_band; g.ldc = C;
        g.out = half == 0 ? ub.b32 : out; g.out_band = half == 0 ? (int64_t)ub.widest : out_band;
        g.bias = pf_of(nb, [&](int b) { return (const float*)(half == 0 ? rw(b).c1.bias : rw(b).c2.bias); });
        g.col_add = PF{};
        if (half == 1) { g.residual = x; g.ldr = C; g.res_band = x_band; g.col_add = col_add; }
        auto cw = [&](int b) -> const ConvW& { return half == 0 ? rw(b).c1 : rw(b).c2; };
        // window of output row t: rows t - pad + j * dilation, j = 0..2; contiguous (one segment per term) when dilation == 1
        GnFuse* gf = half == 0 ? &mid : out_gf;
        OpOut* op = half == 1 ? out_op : nullptr;
        if (dilation == 1) e = launch_gemm(s, m, nb, g, ub.hi, ub.lo, cw, 1, 0, 3 * C, gf, op);
        else e = launch_gemm(s, m, nb, g, ub.hi, ub.lo, cw, 3, (int64_t)dilation * C, C, gf, op);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

}  // namespace

extern "C" {

}  // extern "C"

// latent: fp32 [codec_dim][frames] (the codec's latent) or NULL when codec_dim == 0; wav_ref: fp32 [T], the EnCodec decode the
// result is re-equalised against (NULL: no re_eq) -- both on the host, or (inputs_on_device) in device memory, where they are read
// in place (mv_codec_decode_device's buffers: no PCIe leg between the codec and this stage); noise: host fp32
// [n_models][n_eval + 1][T] (initial noise, then one draw per evaluation; parity tests) or NULL = the engine's Philox stream keyed
// by `seed`; wav_out: host fp32 [T].
static int mbd_generate_impl(mv_mbd* m, const float* latent, int32_t frames, const float* wav_ref, int32_t T, const float* noise, uint64_t seed,
                             float* wav_out, bool inputs_on_device) {
    if (!m || !wav_out) return mv_fail(MV_ERR_INVALID, "mv_mbd_generate: null argument");
    if (!m->finalized) return mv_fail(MV_ERR_STATE, "mv_mbd_generate before mv_mbd_finalize");
    (void)hipSetDevice(m->device);   // lanes of a batch are driven from worker threads: a new thread's current device is 0
    const mv_mbd_config& g = m->cfg;
    if (T < g.kernel * 4 || (g.codec_dim > 0 && (!latent || frames < 1))) return mv_fail(MV_ERR_INVALID, "mv_mbd_generate: bad sizes (T = %d, frames = %d)", T, frames);
    hipStream_t s = m->stream;
    const int k = g.kernel, st = g.stride, pad = (k - st) / 2, D = g.depth, G = g.norm_groups, NB = g.n_models;
    Work& wk = m->work;
    wk.rewind(s);
    UnetBuffers ub;
    ub.NB = NB;
    // geometry
    ub.T.resize(D); ub.Tpad.resize(D);
    size_t max_c = 8, max_gn = 1;
    {
        int t = T;
        for (int i = 0; i < D; ++i) {
            ub.Tpad[i] = (t + st - 1) / st * st;
            ub.T[i] = (ub.Tpad[i] + 2 * pad - k) / st + 1;
            t = ub.T[i];
            ub.widest = std::max(ub.widest, (size_t)(ub.T[i] + 2 * st + 8) * m->ch[i]);
            max_c = std::max(max_c, (size_t)m->ch[i]);
            max_gn = std::max(max_gn, gn_blocks(ub.T[i], m->ch[i], G) + G);
        }
        if (ub.T[D - 1] < 2) return mv_fail(MV_ERR_INVALID, "mv_mbd_generate: %d samples are too few for depth %d", T, D);
        if (g.codec_dim > 0 && frames > 2 * ub.T[D - 1]) return mv_fail(MV_ERR_INVALID, "conditioning would be downsampled by more than 2");
    }
    ub.widest = (ub.widest + 7) & ~(size_t)7;
    ub.op_band = (ub.widest + (size_t)(2 * k + 2 * st + 8) * max_c + 7) & ~(size_t)7;   // operand rows + zero margins / transposed-conv overhang
#define MG_TRY(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) return mv_fail(MV_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); } while (0)
    ub.skip.resize(D); ub.skip_band.resize(D);
    for (int i = 0; i < D; ++i) {
        ub.skip_band[i] = ((size_t)ub.T[i] * m->ch[i] + 7) & ~(size_t)7;
        MG_TRY(wk.get(&ub.skip[i], ub.skip_band[i] * NB, false));
    }
    MG_TRY(wk.get(&ub.a32, ub.widest * NB, false));
    MG_TRY(wk.get(&ub.b32, ub.widest * NB, false));
    MG_TRY(wk.get(&ub.c32, ub.widest * NB, false));
    MG_TRY(wk.get(&ub.hi, ub.op_band * NB, true));
    MG_TRY(wk.get(&ub.lo, ub.op_band * NB, true));
    ub.hi_lvl.assign(D, nullptr); ub.lo_lvl.assign(D, nullptr);
    if (gn_fuse_enabled())
        for (int i = 1; i < D; ++i) {
            MG_TRY(wk.get(&ub.hi_lvl[i], ub.op_band * NB, true));
            if (g.precision > 2) MG_TRY(wk.get(&ub.lo_lvl[i], ub.op_band * NB, true));
        }
    MG_TRY(wk.get(&ub.dec_out, ub.op_band * NB, false));
    MG_TRY(wk.get(&ub.part, max_gn * 2 * NB, false));
    MG_TRY(wk.get(&ub.part2, max_gn * 2 * NB, false));
    MG_TRY(wk.get(&ub.stats, (size_t)2 * G * NB, false));
    const int64_t WB = (int64_t)ub.widest, OB = (int64_t)ub.op_band;
    float *cur = nullptr, *est = nullptr, *total = nullptr, *d_lat = nullptr, *d_noise = nullptr, *d_ref = nullptr;
    const int64_t est_band = (int64_t)T + 4 * st;
    MG_TRY(wk.get(&cur, (size_t)T * NB, false));
    MG_TRY(wk.get(&est, (size_t)est_band * NB, false));
    MG_TRY(wk.get(&total, (size_t)T, true));
    if (g.codec_dim > 0) {
        MG_TRY(wk.get(&d_lat, (size_t)g.codec_dim * frames, false));
        if (inputs_on_device) d_lat = const_cast<float*>(latent);   // read-only below
        else MG_TRY(hipMemcpyAsync(d_lat, latent, (size_t)g.codec_dim * frames * 4, hipMemcpyHostToDevice, s));
    }
    if (noise) MG_TRY(wk.get(&d_noise, (size_t)T * NB, false));
    const int max_bands = std::max(g.proc_bands, g.eq_bands);
    float *bands = nullptr, *bands2 = nullptr, *d_scale = nullptr, *d_shift = nullptr;
    double *st_a = nullptr, *st_b = nullptr;
    if (max_bands > 1) {
        MG_TRY(wk.get(&bands, (size_t)max_bands * T, false));
        MG_TRY(wk.get(&bands2, (size_t)max_bands * T, false));
        MG_TRY(wk.get(&d_scale, (size_t)max_bands, false));
        MG_TRY(wk.get(&d_shift, (size_t)max_bands, false));
        MG_TRY(wk.get(&st_a, (size_t)2 * max_bands, false));
        MG_TRY(wk.get(&st_b, (size_t)2 * max_bands, false));
    }
    float *lstm_gin = nullptr, *lstm_y0 = nullptr, *lstm_y1 = nullptr, *lstm_rev = nullptr;
    const int Cb = m->ch[D - 1], Tb = ub.T[D - 1];
    const int64_t lstm_band = (int64_t)Tb * Cb;
    if (g.bilstm) {
        MG_TRY(wk.get(&lstm_gin, (size_t)Tb * 4 * Cb, false));
        MG_TRY(wk.get(&lstm_y0, (size_t)Tb * 2 * Cb, false));
        MG_TRY(wk.get(&lstm_y1, (size_t)Tb * 2 * Cb, false));
        MG_TRY(wk.get(&lstm_rev, (size_t)lstm_band * NB, false));
    }
    auto split = [&](const float* x, int n_bands, const float* bank, int taps, float* lowbuf, float* out_bands) -> hipError_t {
        const float* bank_t = bank == m->proc_bank ? m->proc_bank_t : m->eq_bank_t;
        if (bank_t && !getenv("MV_MBD_FIR_VALU")) {   // the Toeplitz product on the fp32 matrix cores; MV_MBD_FIR_VALU=1: the VALU kernel (A/B, tests)
            const int taps2 = taps + 1;   // even: the transposed bank carries one zero row
            hipLaunchKernelGGL(fir_bank_mfma_kernel, dim3((T + FIRM_TILE - 1) / FIRM_TILE), dim3(256), (size_t)(FIRM_TILE + taps2 + 2) * 4, s, x, T, bank_t, taps2,
                               (taps - 1) / 2, n_bands - 1, lowbuf);
        } else
        hipLaunchKernelGGL(fir_bank_kernel, dim3((T + FIR_TILE - 1) / FIR_TILE, n_bands - 1), dim3(256), (size_t)(FIR_TILE + 2 * ((taps + 3) & ~3) + 4) * 4, s, x, T, bank, taps, lowbuf);
        hipLaunchKernelGGL(bands_kernel, dim3((T + 255) / 256), dim3(256), 0, s, x, lowbuf, T, n_bands, out_bands);
        return hipGetLastError();
    };
    auto coef = [&](int ev) {
        DdpmCoef c{};
        for (int b = 0; b < NB; ++b) {
            const BandW& B = m->band[b];
            c.noise_scale[b] = B.noise_scale; c.clip[b] = B.clip;
            if (ev >= 0) { c.est_c[b] = B.est_c[ev]; c.inv_sqrt_alpha[b] = B.inv_sqrt_alpha[ev]; c.sigma[b] = B.sigma[ev]; c.rescale[b] = B.steps[ev] == 0 ? B.rescale : 1.0f; }
        }
        return c;
    };
    auto upload_noise = [&](int slot) -> hipError_t {   // draw `slot` of every band -> d_noise [band][T]
        for (int b = 0; b < NB; ++b) {
            hipError_t e = hipMemcpyAsync(d_noise + (size_t)b * T, noise + ((size_t)b * (g.n_eval + 1) + slot) * T, (size_t)T * 4, hipMemcpyHostToDevice, s);
            if (e != hipSuccess) return e;
        }
        return hipSuccess;
    };

    if (noise) MG_TRY(upload_noise(0));
    hipLaunchKernelGGL(init_noise_kernel, dim3((T + 255) / 256, NB), dim3(256), 0, s, cur, noise ? d_noise : nullptr, T, coef(-1), seed);
    for (int ev = 0; ev < g.n_eval; ++ev) {
        // ---------------- encoder ---------------------------------------------------------------------------------------------
        const bool add_fuse = !getenv("MV_MBD_NO_ADD_FUSE");   // read per call: a test compares both ways in one process
        const bool skip_direct = add_fuse && !g.bilstm;         // z = 0: the first decoder ResBlock reads the innermost skip in place (no copy)
        GnFuse skip_gf{G, ub.part, 0};
        OpOut next_op;   // the operand the current level's last epilogue writes for the next level's strided convolution
        const float* zin = cur;   // [band][Tin][Cin]
        int64_t zin_band = T;
        int Tin = T;
        for (int i = 0; i < D; ++i) {
            const int Cin = i == 0 ? 1 : m->ch[i - 1], Cout = m->ch[i], To = ub.T[i];
            auto E = [&](int b) -> const EncW& { return m->band[b].enc[i]; };
            GnFuse enc_gf{G, ub.part, 0};   // the partials of the strided convolution's output, from the kernel that writes it
            if (i == 0) {
                const PF w_in = pf_of(NB, [&](int b) { return (const float*)E(b).conv_in_w; });
                if (k == 8 && st == 4 && Cout <= 1024 && gn_fuse_enabled()) {   // input samples through LDS, weights in registers, statistics from the same pass
                    const int rows_per_block = 256 / (Cout / 4) * CONV_IN_PASSES, blocks = (To + rows_per_block - 1) / rows_per_block;
                    const size_t lds = (size_t)(rows_per_block * st + k - st) * 4;
                    const bool fuse = ((Cout / G) & 3) == 0;
                    if (!fuse) hipLaunchKernelGGL(conv_in_rows_kernel<0>, dim3(blocks, NB), dim3(256), lds, s, zin, Tin, w_in, Cout, pad, ub.a32, WB, To, G, (double*)nullptr);
                    else if (G <= 4) hipLaunchKernelGGL(conv_in_rows_kernel<4>, dim3(blocks, NB), dim3(256), lds, s, zin, Tin, w_in, Cout, pad, ub.a32, WB, To, G, ub.part);
                    else hipLaunchKernelGGL(conv_in_rows_kernel<8>, dim3(blocks, NB), dim3(256), lds, s, zin, Tin, w_in, Cout, pad, ub.a32, WB, To, G, ub.part);
                    if (fuse) enc_gf.tiles = blocks;
                } else
                    hipLaunchKernelGGL(conv_in_kernel, dim3((unsigned)(((size_t)To * (Cout / 4) + 255) / 256), NB), dim3(256), (size_t)Cout * k * 4, s, zin, Tin, w_in,
                                       Cout, k, st, pad, ub.a32, WB, To);
            } else {
                // operand: zin with `pad` zero rows in front and (Tpad - Tin) + k behind
                const int back = ub.Tpad[i] - Tin + k;
                const bool have_op = next_op.written;   // (the previous level's last epilogue wrote hi_lvl / lo_lvl[i])
                if (!have_op) launch_norm_act(s, ub, zin, zin_band, Tin, Cin, G, nullptr, PF{}, PF{}, 0, nullptr, 0, ub.hi, g.precision > 2 ? ub.lo : nullptr, pad, back);
                MbdGemmArgs ga;
                ga.M = To; ga.N = Cout; ga.lda = (int64_t)st * Cin; ga.a_band = OB; ga.out = ub.a32; ga.ldc = Cout; ga.out_band = WB;
                ga.bias = PF{}; ga.col_add = PF{};
                MG_TRY(launch_gemm(s, m, NB, ga, have_op ? ub.hi_lvl[i] : ub.hi, have_op ? ub.lo_lvl[i] : ub.lo, [&](int b) -> const ConvW& { return E(b).conv; }, 1, 0,
                                   k * Cin, &enc_gf));
            }
            MG_TRY(stats_of_gemm_output(s, enc_gf, ub.a32, WB, To, Cout, G, ub));
            const bool has_emb = i == 0 || g.emb_all_layers;
            const PF emb = has_emb ? pf_of(NB, [&](int b) { return (const float*)(E(b).emb + (size_t)m->band[b].steps[ev] * Cout); }) : PF{};
            if (g.res_blocks == 0) return mv_fail(MV_ERR_INVALID, "res_blocks = 0 is not supported");
            // ResBlock r's input statistics: the partials its producer left -- this pass for the first one (round 6), ResBlock r - 1's conv2 after that
            GnFuse chain{G, ub.part2, 0};   // (its own records: this pass may still be finishing the strided convolution's in ub.part)
            launch_norm_act(s, ub, ub.a32, WB, To, Cout, G, ub.stats, pf_of(NB, [&](int b) { return (const float*)E(b).norm.g; }),
                            pf_of(NB, [&](int b) { return (const float*)E(b).norm.b; }), g.act, ub.c32, WB, nullptr, nullptr, 0, 0, &chain);
            for (int r = 0; r < g.res_blocks; ++r) {
                const bool lastr = r == g.res_blocks - 1;
                const GnFuse in_gf = chain;
                // the innermost skip is the decoder's first ResBlock input when there is no BiLSTM: its statistics come from this epilogue too
                GnFuse* out_gf = !lastr ? &chain : (i == D - 1 && skip_direct) ? &skip_gf : nullptr;
                next_op = OpOut{};
                if (lastr && i + 1 < D && ub.hi_lvl[i + 1]) next_op = OpOut{ub.hi_lvl[i + 1], ub.lo_lvl[i + 1], OB, (int64_t)pad * Cout, false};
                MG_TRY(res_block(s, m, [&](int b) -> const ResW& { return E(b).res[r]; }, ub.c32, WB, lastr ? ub.skip[i] : ub.c32,
                                 lastr ? (int64_t)ub.skip_band[i] : WB, To, Cout, 1 << r, lastr ? emb : PF{}, ub, out_gf, in_gf.tiles > 0 ? &in_gf : nullptr, false,
                                 lastr ? &next_op : nullptr));
            }
            zin = ub.skip[i];
            zin_band = (int64_t)ub.skip_band[i];
            Tin = To;
        }
        // ---------------- bottleneck ------------------------------------------------------------------------------------------
        // z = encoder output (+ codec conditioning); without a BiLSTM audiocraft's forward replaces z by zeros (unet.py), so the
        // decoder starts from the last skip alone
        const float* zb = nullptr;  // [band][Tb][Cb] added to the last skip, or null (zeros)
        if (g.bilstm) {
            for (int b = 0; b < NB; ++b) {   // the recurrences run band after band (latency-bound chains, 64 workgroups each)
                const BandW& B = m->band[b];
                MG_TRY(hipMemcpyAsync(ub.c32, ub.skip[D - 1] + (size_t)b * ub.skip_band[D - 1], (size_t)Tb * Cb * 4, hipMemcpyDeviceToDevice, s));
                if (g.codec_dim > 0)
                    hipLaunchKernelGGL(cond_add_kernel, dim3(Tb), dim3(256), (size_t)g.codec_dim * 4, s, ub.c32, Tb, Cb, d_lat, frames, g.codec_dim, B.codec_w, B.codec_b);
                // BLSTM: 2 layers, both directions (the reverse direction walks the sequence through negative strides), then Linear(2 C -> C)
                const float* in = ub.c32;
                int I = Cb;
                float* ys[2] = {lstm_y0, lstm_y1};
                for (int l = 0; l < 2; ++l) {
                    for (int d = 0; d < 2; ++d) {
                        const int sl = l * 2 + d;
                        MG_TRY(launch_lstm_inproj(s, in, B.lstm.w_ih[sl], B.lstm.bias[sl], lstm_gin, Tb, I, 4 * Cb));
                        if (++m->lstm_serial >= 65535u) { MG_TRY(hipMemsetAsync(m->lstm_hx, 0, (size_t)2 * Cb * 8, s)); m->lstm_serial = 1; }
                        LstmSeqArgs la{};
                        // direction 1 walks the sequence backwards: step t reads gin[Tb-1-t] and writes y[Tb-1-t]
                        la.gin = d == 0 ? lstm_gin : lstm_gin + (size_t)(Tb - 1) * 4 * Cb; la.gs_b = 0; la.gs_t = d == 0 ? 4 * Cb : -(int64_t)4 * Cb; la.gs_r = 1;
                        la.whh = B.lstm.w_hh[sl];
                        la.y = ys[l] + (d == 0 ? 0 : (size_t)(Tb - 1) * 2 * Cb) + (size_t)d * Cb; la.ys_b = 0; la.ys_t = d == 0 ? 2 * Cb : -(int64_t)2 * Cb; la.ys_u = 1;
                        la.hx = m->lstm_hx; la.tag_base = m->lstm_serial << 16; la.abort_flag = m->lstm_abort;
                        la.B = 1; la.H = Cb; la.T = Tb;
                        MG_TRY(launch_lstm_seq(s, la));
                    }
                    in = ys[l];
                    I = 2 * Cb;
                }
                MG_TRY(launch_lstm_inproj(s, lstm_y1, B.lstm.lin_w, B.lstm.lin_b, lstm_rev + (size_t)b * lstm_band, Tb, 2 * Cb, Cb));  // Linear(2C -> C)
            }
            zb = lstm_rev;
        }
        // ---------------- decoder ---------------------------------------------------------------------------------------------
        const float* z = zb;
        int64_t z_band = lstm_band;
        // Round 6: from the second level on z + skip is written by the up-convolution's own epilogue (MbdGemmArgs::valid_lo / residual),
        // together with the GroupNorm partials of the sum -- add_and_group_stats read both tensors again for that
        bool sum_ready = false;           // c32 already holds z + skip of this level
        GnFuse sum_gf{G, ub.part, 0};     // ... and these are its partials (tiles = 0: the kernel that ran leaves none)
        for (int i = 0; i < D; ++i) {
            const int lvl = D - 1 - i, C = m->ch[lvl], Cout = lvl == 0 ? 1 : m->ch[lvl - 1], Ts = ub.T[lvl];
            auto Dw = [&](int b) -> const DecW& { return m->band[b].dec[i]; };
            // z = z[:, :, :skip.T] + skip   (z rows beyond skip.T are dropped; z never is shorter)
            const size_t n = (size_t)Ts * C;
            const float* x0 = ub.c32;     // the first ResBlock's input
            int64_t x0_band = WB;
            const GnFuse* x0_gf = nullptr;
            bool x0_stats_ready = false;
            if (sum_ready) {
                if (sum_gf.tiles > 0) x0_gf = &sum_gf;
            } else if (z == nullptr && skip_direct) {
                x0 = ub.skip[lvl]; x0_band = (int64_t)ub.skip_band[lvl];
                if (skip_gf.tiles > 0) x0_gf = &skip_gf;
            } else if (z != nullptr && ((C / G) & 3) == 0 && add_fuse) {
                // the skip addition and the statistics of its result in one pass (behind the BiLSTM, whose output no GEMM epilogue of ours writes)
                MG_TRY(add_and_group_stats(s, ub.skip[lvl], (int64_t)ub.skip_band[lvl], z, z_band, ub.c32, WB, Ts, C, G, ub));
                x0_stats_ready = true;
            } else
                hipLaunchKernelGGL(add_kernel, dim3((unsigned)((n + 255) / 256), NB), dim3(256), 0, s, ub.skip[lvl], (int64_t)ub.skip_band[lvl], z, z_band, ub.c32, WB, n);
            GnFuse dec_gf{G, ub.part, 0};   // every ResBlock leaves its output partials: the next one's input statistics, then the decoder norm's
            for (int r = 0; r < g.res_blocks; ++r) {
                const GnFuse in_gf = dec_gf;
                MG_TRY(res_block(s, m, [&](int b) -> const ResW& { return Dw(b).res[r]; }, r == 0 ? x0 : ub.c32, r == 0 ? x0_band : WB, ub.c32, WB, Ts, C, 1 << r, PF{}, ub,
                                 &dec_gf, r == 0 ? x0_gf : in_gf.tiles > 0 ? &in_gf : nullptr, r == 0 && x0_stats_ready));
            }
            MG_TRY(stats_of_gemm_output(s, dec_gf, ub.c32, WB, Ts, C, G, ub));
            const PF ng = pf_of(NB, [&](int b) { return (const float*)Dw(b).norm.g; }), nbias = pf_of(NB, [&](int b) { return (const float*)Dw(b).norm.b; });
            if (lvl == 0) {
                const PF w_out = pf_of(NB, [&](int b) { return (const float*)Dw(b).convtr_out_w; });
                if (k == 8 && st == 4 && C <= 56 && gn_fuse_enabled()) {   // by rows, the decoder norm + activation applied on the way in: no act(gn(x)) tensor
                    MG_TRY(finish_pending_stats(s, ub, G));
                    const size_t lds = ((size_t)CTO_ROWS * (C + 4) + (size_t)C * k + (size_t)CTO_ROWS * k) * 4;
                    hipLaunchKernelGGL(convtr_out_rows_kernel, dim3((unsigned)((Ts + 1 + CTO_ROWS - 2) / (CTO_ROWS - 1)), NB), dim3(CTO_ROWS), lds, s, ub.c32, WB, Ts, C, w_out, pad,
                                       est, est_band, T, G, ub.stats, ng, nbias, g.act);
                } else {
                    launch_norm_act(s, ub, ub.c32, WB, Ts, C, G, ub.stats, ng, nbias, g.act, ub.a32, WB, nullptr, nullptr, 0, 0);
                    hipLaunchKernelGGL(convtr_out_kernel, dim3((T + 255) / 256, NB), dim3(256), (size_t)k * C * 4, s, ub.a32, WB, Ts, C, w_out, k, st, pad, est, est_band, T);
                }
            } else {
                // operand rows: [zero row | z rows 0..Ts-1 | zero row]; GEMM row q = (x[q-1], x[q]), q = 0..Ts
                launch_norm_act(s, ub, ub.c32, WB, Ts, C, G, ub.stats, ng, nbias, g.act, nullptr, 0, ub.hi, g.precision > 2 ? ub.lo : nullptr, 1, 1);
                MbdGemmArgs ga;
                ga.M = Ts + 1; ga.N = st * Cout; ga.lda = C; ga.a_band = OB; ga.ldc = (int64_t)st * Cout;
                ga.bias = PF{}; ga.col_add = PF{};
                // row q holds output times q st - pad + p: the valid signal starts `pad` rows into the row-major image
                if (add_fuse) {
                    // ... so with `out` and `residual` set `pad` time steps in front of c32 / the next skip, image element f is time step f / Cout - pad
                    // of both: the epilogue adds the skip, drops what lies outside [0, skip.T) and sums the rest for the next GroupNorm (c32 is free:
                    // the pass above was its last reader)
                    const int Tn = ub.T[lvl - 1];
                    ga.out = ub.c32 - (size_t)pad * Cout; ga.out_band = WB;
                    ga.residual = ub.skip[lvl - 1] - (size_t)pad * Cout; ga.ldr = ga.ldc; ga.res_band = (int64_t)ub.skip_band[lvl - 1];
                    ga.valid_lo = (int64_t)pad * Cout; ga.valid_hi = ga.valid_lo + (int64_t)Tn * Cout;
                    sum_gf = GnFuse{G, ub.part, 0, Cout};
                    MG_TRY(launch_gemm(s, m, NB, ga, ub.hi, ub.lo, [&](int b) -> const ConvW& { return Dw(b).convtr; }, 1, 0, 2 * C, &sum_gf));
                    sum_ready = true;
                    ++m->gemm_launches[7];
                } else {
                    ga.out = ub.dec_out; ga.out_band = OB;
                    MG_TRY(launch_gemm(s, m, NB, ga, ub.hi, ub.lo, [&](int b) -> const ConvW& { return Dw(b).convtr; }, 1, 0, 2 * C));
                    z = ub.dec_out + (size_t)pad * Cout;
                    z_band = OB;
                }
            }
        }
        MG_TRY(hipGetLastError());
        // ---------------- DDPM update -----------------------------------------------------------------------------------------
        bool any_sigma = false;
        for (int b = 0; b < NB; ++b) any_sigma = any_sigma || m->band[b].sigma[ev] > 0.f;
        if (noise && any_sigma) MG_TRY(upload_noise(ev + 1));
        hipLaunchKernelGGL(ddpm_kernel, dim3((T + 255) / 256, NB), dim3(256), 0, s, cur, est, est_band, noise ? d_noise : nullptr, T, coef(ev), seed, (uint32_t)(ev + 1));
    }
    // ---------------- MultiBandProcessor.return_sample of every band, summed in band order into the total ---------------------------
    for (int bi = 0; bi < NB; ++bi) {
        const BandW& B = m->band[bi];
        const float* x = cur + (size_t)bi * T;
        if (g.proc_bands > 1) {
            MG_TRY(split(x, g.proc_bands, m->proc_bank, m->proc_taps, bands2, bands));
            MG_TRY(hipMemcpyAsync(d_scale, B.proc_scale.data(), (size_t)g.proc_bands * 4, hipMemcpyHostToDevice, s));
            MG_TRY(hipMemcpyAsync(d_shift, B.proc_mean.data(), (size_t)g.proc_bands * 4, hipMemcpyHostToDevice, s));
            hipLaunchKernelGGL(band_mix_kernel, dim3((T + 255) / 256), dim3(256), 0, s, bands, T, g.proc_bands, d_scale, d_shift, total, 1);
            MG_TRY(hipStreamSynchronize(s));  // d_scale / d_shift are reused by the next band
        } else {
            hipLaunchKernelGGL(add_kernel, dim3((T + 255) / 256, 1), dim3(256), 0, s, total, (int64_t)0, x, (int64_t)0, total, (int64_t)0, (size_t)T);
        }
    }
    // ---------------- re_eq against the EnCodec decode ---------------------------------------------------------------------------------
    float* result = total;
    if (wav_ref && g.eq_bands > 1) {
        MG_TRY(wk.get(&d_ref, (size_t)T, false));
        if (inputs_on_device) d_ref = const_cast<float*>(wav_ref);
        else MG_TRY(hipMemcpyAsync(d_ref, wav_ref, (size_t)T * 4, hipMemcpyHostToDevice, s));
        float* low = nullptr;
        MG_TRY(wk.get(&low, (size_t)g.eq_bands * T, false));
        MG_TRY(split(d_ref, g.eq_bands, m->eq_bank, m->eq_taps, low, bands2));
        hipLaunchKernelGGL(band_stats_kernel, dim3(g.eq_bands), dim3(1024), 0, s, bands2, T, st_b);
        MG_TRY(split(total, g.eq_bands, m->eq_bank, m->eq_taps, low, bands));
        hipLaunchKernelGGL(band_stats_kernel, dim3(g.eq_bands), dim3(1024), 0, s, bands, T, st_a);
        hipLaunchKernelGGL(eq_scale_kernel, dim3(1), dim3(64), 0, s, st_a, st_b, g.eq_bands, (double)T, d_scale);
        hipLaunchKernelGGL(band_mix_kernel, dim3((T + 255) / 256), dim3(256), 0, s, bands, T, g.eq_bands, d_scale, (const float*)nullptr, cur, 0);
        result = cur;
    }
    MG_TRY(hipGetLastError());
    uint32_t aborted = 0;
    MG_TRY(hipMemcpyAsync(wav_out, result, (size_t)T * 4, hipMemcpyDeviceToHost, s));
    if (m->lstm_abort) MG_TRY(hipMemcpyAsync(&aborted, m->lstm_abort, 4, hipMemcpyDeviceToHost, s));
    MG_TRY(hipStreamSynchronize(s));
#undef MG_TRY
    if (aborted) {
        (void)hipMemset(m->lstm_abort, 0, 4);
        if (!lstm_stepwise_mode()) {   // survive it: redo the call with one LSTM launch per time step (and stay there)
            lstm_force_stepwise();
            return mbd_generate_impl(m, latent, frames, wav_ref, T, noise, seed, wav_out, inputs_on_device);
        }
        return mv_fail(MV_ERR_HIP, "mv_mbd_generate: the BiLSTM recurrence failed");
    }
    return MV_OK;
}

extern "C" {

int mv_mbd_generate(mv_mbd* m, const float* latent, int32_t frames, const float* wav_ref, int32_t T, const float* noise, uint64_t seed,
                    float* wav_out) {
    return mbd_generate_impl(m, latent, frames, wav_ref, T, noise, seed, wav_out, false);
}

// The same with the codec's outputs read in place from DEVICE memory (mv_codec_decode_device): d_latent fp32 [codec_dim][frames],
// d_wav_ref fp32 [T]; the engine's own noise stream.
int mv_mbd_generate_device(mv_mbd* m, const float* d_latent, int32_t frames, const float* d_wav_ref, int32_t T, uint64_t seed, float* wav_out) {
    return mbd_generate_impl(m, d_latent, frames, d_wav_ref, T, nullptr, seed, wav_out, true);
}

int mv_mbd_unet_forward(mv_mbd* m, int32_t band_idx, const float* x, int32_t T, int32_t step, const float* latent, int32_t frames, float* out) {
    if (!m || !x || !out || band_idx < 0 || band_idx >= m->cfg.n_models) return mv_fail(MV_ERR_INVALID, "mv_mbd_unet_forward: bad argument");
    // run the generator for ONE evaluation with a schedule that makes the update the identity on the estimate: cur' = -est * 1
    mv_mbd_config saved = m->cfg;
    BandW& B = m->band[band_idx];
    BandW keep = B;
    std::vector<BandW> all = m->band;
    // isolate the band: a one-model, one-evaluation view
    m->cfg.n_models = 1; m->cfg.n_eval = 1; m->cfg.proc_bands = 1; m->cfg.eq_bands = 1;
    m->band.assign(1, keep);
    BandW& V = m->band[0];
    V.steps = {step}; V.est_c = {1.0f}; V.inv_sqrt_alpha = {-1.0f}; V.sigma = {0.f}; V.clip = 0.f; V.rescale = 1.f; V.noise_scale = 1.f;
    // with cur = x: cur' = (x - est) * -1 = est - x  -> add x back on the host
    std::vector<float> nz((size_t)2 * T);
    memcpy(nz.data(), x, (size_t)T * 4);
    const int rc = mv_mbd_generate(m, latent, frames, nullptr, T, nz.data(), 0, out);
    m->cfg = saved;
    m->band = all;
    if (rc != MV_OK) return rc;
    for (int i = 0; i < T; ++i) out[i] += x[i];
    return MV_OK;
}

// One evaluation of EVERY band model in one pass -- the launch shapes of mv_mbd_generate (all bands per launch: that is what decides
// which GEMM kernel a layer runs on), which the one-band hook above cannot reproduce.  x: host fp32 [n_models][T], band b's input;
// out: host fp32 [T] = sum over the bands of unet_b(x_b) (the generator adds the bands up; the sum is what comes back).
int mv_mbd_unet_forward_sum(mv_mbd* m, const float* x, int32_t T, int32_t step, const float* latent, int32_t frames, float* out) {
    if (!m || !x || !out) return mv_fail(MV_ERR_INVALID, "mv_mbd_unet_forward_sum: null argument");
    const mv_mbd_config saved = m->cfg;
    const std::vector<BandW> all = m->band;
    const int NB = saved.n_models;
    // one evaluation, a schedule that makes the update cur' = est - x, no processor, no re_eq
    m->cfg.n_eval = 1; m->cfg.proc_bands = 1; m->cfg.eq_bands = 1;
    for (BandW& V : m->band) { V.steps = {step}; V.est_c = {1.0f}; V.inv_sqrt_alpha = {-1.0f}; V.sigma = {0.f}; V.clip = 0.f; V.rescale = 1.f; V.noise_scale = 1.f; }
    std::vector<float> nz((size_t)NB * 2 * T, 0.f);   // [band][initial | evaluation 0][T]
    for (int b = 0; b < NB; ++b) memcpy(nz.data() + (size_t)b * 2 * T, x + (size_t)b * T, (size_t)T * 4);
    const int rc = mv_mbd_generate(m, latent, frames, nullptr, T, nz.data(), 0, out);
    m->cfg = saved;
    m->band = all;
    if (rc != MV_OK) return rc;
    for (int b = 0; b < NB; ++b)
        for (int i = 0; i < T; ++i) out[i] += x[(size_t)b * T + i];
    return MV_OK;
}

// Launches per GEMM kernel since mv_mbd_create: out[0] mbd_gemm_t256_kernel, [1] mbd_gemm_glds3_kernel<64, 192>, [2] <128, 128>,
// [3] mbd_gemm_thin_kernel, [4] mbd_gemm_glds_kernel, [5] mbd_gemm_kernel<128>, [6] mbd_gemm_kernel<64>; [7] (not a kernel) the up-convolutions among
// them whose epilogue wrote z + skip (round 6).
int mv_mbd_kernel_counts(const mv_mbd* m, int64_t* out) {
    if (!m || !out) return mv_fail(MV_ERR_INVALID, "mv_mbd_kernel_counts: null argument");
    for (int i = 0; i < 8; ++i) out[i] = m->gemm_launches[i];
    return MV_OK;
}

}  // extern "C"
