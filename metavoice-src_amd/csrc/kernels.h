// Internal kernel-launch interface shared by the engine and the mv_op_* test entry points.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/metavoice_hip.h"

typedef uint16_t bf16_t;

// Engines of more utterances keep sample_kernel behind every step: with the token draw compiled in (inlined or out of line) those
// instances measured slower (r03a, r03x: tools/experiments/README.md).  One-utterance engines of every format draw inside the launch.

// ---- tuning knobs (mv_tune): defaults are the measured best for MetaVoice-1B on MI355X (DESIGN.md "Tuning") ------
struct MvTune {
    int gemv_threads[4] = {0, 0, 0, 0};  // per epilogue kind (GEMV_*): 0 = built-in choice
    int gemv_nr[4] = {0, 0, 0, 0};       // weight rows per wave: 0 = built-in choice
    int gemv_nt = 1;                     // 1: non-temporal weight loads, 0: default cache policy
    int use_graph = 1;
    int mega_fault = 0;                  // test hook: workgroup (mega_fault - 1) withholds one publish per step -> every wait must time out cleanly
    int gemv_ablate = 0;                 // profiling only (results are wrong): 1 no weight loads, 2 no dots, 4 no prologue, 8 no reductions
    int mega = 1;                        // 1: persistent decode-step kernel (decode_mega.hip) when the configuration supports it
                                         // (MetaVoice-1B shape, >= 256 CUs); 0: per-layer launches
    int mega_steps = 512;                // decode steps per launch of the folding persistent kernel
    int mega_fold = 1;                   // 1: the persistent kernel also draws the token and runs several steps per launch (engines
                                         // of one utterance); 0: one step per launch + sample_kernel
    int mfma = 1;                        // 1: engines of 4 utterances (bf16) run the persistent step on the matrix cores (decode_mfma.hip: tiled
                                         // weight copy, not bit-identical to the launches); 0: decode_mega.hip's bit-identical instance
};
extern MvTune g_tune;

// ---- gemv.hip ------------------------------------------------------------------------------------------------
enum { GEMV_PLAIN = 0, GEMV_RESID = 1, GEMV_SWIGLU = 2, GEMV_QKV = 3 };
struct GemvArgs {
    const bf16_t* x = nullptr;         // [rows][K]
    const bf16_t* w = nullptr;         // [N][K]
    const bf16_t* w2 = nullptr;        // [N][K], SWIGLU only (w3)
    const bf16_t* norm_w = nullptr;    // [K] or null: fused RMSNorm prologue
    const bf16_t* residual = nullptr;  // [rows][N], RESID only
    bf16_t* out = nullptr;             // [rows][N]  (QKV: q [rows][dim])
    bf16_t* kcache = nullptr;          // QKV only: [rows][n_head][cache_len][128]
    bf16_t* vcache = nullptr;
    const int32_t* pos = nullptr;      // QKV only: [rows/2] current input position per utterance
    int32_t cache_len = 0, n_head = 0, dim = 0;
    int32_t N = 0, K = 0;
    float eps = 1e-5f;
    int32_t ablate = 0;                // MvTune::gemv_ablate
};
hipError_t launch_gemv(hipStream_t s, int rows, const GemvArgs& a, int epi);
bool gemv_supported(int rows, int K);

// ---- gemv_i4.hip: int4 group-128 weight-only projections + the quantiser ---------------------------------------
struct GemvI4Args {
    const bf16_t* x = nullptr;            // [rows][K]
    const unsigned char* codes = nullptr; // [N][K/2], PAIR-MAJOR inside every 32-bit word (gemv_i4_core.h: i4_pair_nibble; launch_repack_i4_pairs)
    const uint32_t* sz = nullptr;         // [N][K/128]: bf16 scale | bf16 zero << 16
    const unsigned char* codes2 = nullptr;  // SWIGLU only (w3)
    const uint32_t* sz2 = nullptr;
    const bf16_t* norm_w = nullptr;
    const bf16_t* residual = nullptr;
    bf16_t* out = nullptr;
    bf16_t* kcache = nullptr;
    bf16_t* vcache = nullptr;
    const int32_t* pos = nullptr;
    int32_t cache_len = 0, n_head = 0, dim = 0;
    int32_t N = 0, K = 0;
    float eps = 1e-5f;
};
hipError_t launch_gemv_i4(hipStream_t s, int rows, const GemvI4Args& a, int epi);
bool gemv_i4_supported(int rows, int K);
// the reference's quantiser: codes [N][K/2] in the ABI layout (column 2j in the low, 2j+1 in the high nibble of byte j), sz as above;
// `deq` (may alias `w`) receives bf16((q - 8) * scale + zero) when non-null
hipError_t launch_quantize_i4(hipStream_t s, const bf16_t* w, int N, int K, unsigned char* codes, uint32_t* sz, bf16_t* deq);
// ABI layout -> the pair-major layout the GEMV kernels read (per 32-bit word; `dst` may alias `src`); n_bytes a multiple of 4
hipError_t launch_repack_i4_pairs(hipStream_t s, const unsigned char* src, unsigned char* dst, size_t n_bytes);

// ---- gemv_i8.hip: int8 per-channel weight-only projections + the quantiser ----------------------------------------
struct GemvI8Args {
    const bf16_t* x = nullptr;
    const signed char* q = nullptr;     // [N][K] int8
    const bf16_t* scales = nullptr;     // [N] bf16
    const signed char* q2 = nullptr;    // SWIGLU only (w3)
    const bf16_t* scales2 = nullptr;
    const bf16_t* norm_w = nullptr;
    const bf16_t* residual = nullptr;
    bf16_t* out = nullptr;
    bf16_t* kcache = nullptr;
    bf16_t* vcache = nullptr;
    const int32_t* pos = nullptr;
    int32_t cache_len = 0, n_head = 0, dim = 0;
    int32_t N = 0, K = 0;
    float eps = 1e-5f;
};
hipError_t launch_gemv_i8(hipStream_t s, int rows, const GemvI8Args& a, int epi);
bool gemv_i8_supported(int rows, int K);
hipError_t launch_quantize_i8(hipStream_t s, const bf16_t* w, int N, int K, signed char* q, bf16_t* scales);

// ---- attention.hip -------------------------------------------------------------------------------------------
constexpr int ATTN_NSPLIT = 8;          // max KV splits per (row, head): 8 x 256 keys = block_size 2048
constexpr int ATTN_PART_STRIDE = 132;   // floats per partial: m, l, pad, pad, o[128]
constexpr int ATTN_TPART_STRIDE = 264;  // persistent kernel: u64 tagged granules per partial slot (130 used; 2112 B, 64-byte aligned)
struct AttnArgs {
    const bf16_t* q = nullptr;        // [rows][n_head*128]
    const bf16_t* kcache = nullptr;   // [rows][n_head][cache_len][128]
    const bf16_t* vcache = nullptr;
    const int32_t* pos = nullptr;     // [rows/2]: keys [0, pos+1) are attended (pos = input position), or
    int32_t pos_is_count = 0;         //   when set, pos[] already holds the key count
    float* partial = nullptr;         // [rows][n_head][ATTN_NSPLIT][ATTN_PART_STRIDE]
    uint32_t* counters = nullptr;     // [rows][n_head], zero between launches
    bf16_t* out = nullptr;            // [rows][n_head*128]
    int32_t n_head = 0, cache_len = 0;
};
hipError_t launch_attention_decode(hipStream_t s, int rows, const AttnArgs& a);

// ---- sampling.hip --------------------------------------------------------------------------------------------
struct UttState {            // one per utterance, device resident
    int32_t pos;             // input position of the token currently in x (= index of the last token in seq)
    int32_t prompt_len;
    int32_t max_len;         // sequence length cap (prompt + new tokens, <= block_size)
    int32_t done;
    int32_t eoa_token;
    int32_t n_tokens;        // tokens in seq so far (prompt prefix consumed + generated)
    int32_t pad0, pad1;
    mv_sampling s;
};
struct SampleArgs {
    const bf16_t* logits = nullptr;       // [rows][vocab]
    UttState* utt = nullptr;              // [n_utt]
    int32_t* pos_arr = nullptr;           // [n_utt] mirror of utt[].pos read by the QKV / attention kernels
    int32_t* seq = nullptr;               // [n_utt][block_size]: prompt then generated tokens
    const bf16_t* tok_emb = nullptr;      // [vocab][dim]
    const bf16_t* pos_emb = nullptr;      // [block_size][dim]
    const bf16_t* spk_proj = nullptr;     // [n_utt][dim] (bf16 of speaker_cond_pos(spk_emb))
    bf16_t* x_next = nullptr;             // [rows][dim]
    int32_t* n_active = nullptr;          // device word: utterances still running
    float* capture = nullptr;             // optional [max_steps][rows][vocab] fp32
    int32_t* capture_step = nullptr;      // device step counter for capture
    int32_t capture_max = 0;
    float* probs_out = nullptr;           // optional [vocab] (mv_op_sample only)
    int32_t* token_out = nullptr;         // optional (mv_op_sample only)
    uint32_t fixed_step = 0;              // mv_op_sample only
    int32_t vocab = 0, dim = 0, block_size = 0, n_utt = 0;
    uint32_t* step_counter = nullptr;     // bumped once per launch: the epoch source of decode_mega.hip's hand-off tags
};
hipError_t launch_sample(hipStream_t s, const SampleArgs& a, bool standalone);
hipError_t launch_embed_first(hipStream_t s, const SampleArgs& a);
hipError_t launch_embed_current(hipStream_t s, const SampleArgs& a);  // x rows of every live utterance at utt[].pos (rollback)
hipError_t launch_embed_rows(hipStream_t s, const int32_t* seq, int T, int dim, const bf16_t* tok_emb, const bf16_t* pos_emb,
                             const bf16_t* spk, bf16_t* out);
hipError_t launch_spk_proj(hipStream_t s, int n_utt, int dim, int spk_dim, const bf16_t* spk_emb, const bf16_t* w,
                           const bf16_t* scales, bf16_t* out);
hipError_t launch_i8_to_bf16(hipStream_t s, const signed char* in, bf16_t* out, size_t n);


// ---- gemm.hip: many-row blocks shared by the first-stage prefill and the second-stage forward ---------------------
enum { GEMM_ACT_NONE = 0, GEMM_ACT_GELU = 1 };
struct GemmArgs {
    const bf16_t* a = nullptr;      // [M][lda] bf16 activations (K-contiguous)
    const bf16_t* w = nullptr;      // [N][K]
    const bf16_t* w2 = nullptr;     // [N][K]: gated (SwiGLU) second matrix, or null
    const bf16_t* bias = nullptr;   // [N] or null
    const bf16_t* bias2 = nullptr;
    const bf16_t* scale = nullptr;  // [N] or null: int8 weight-only Linear, out = bf16(bf16(x . q^T) * scale)  (fast_quantize.py:197-198)
    const bf16_t* scale2 = nullptr; // the same for w2 of a gated pair
    const bf16_t* res_bf16 = nullptr;  // [M][ldc] residual (bf16 stream) or null
    const float* res_f32 = nullptr;    // [M][ldc] residual (fp32 stream) or null
    bf16_t* out_bf16 = nullptr;     // [M][ldc]
    float* out_f32 = nullptr;       // [M][ldc]: when set the output (and residual) is fp32
    int32_t M = 0, N = 0, K = 0, lda = 0, ldc = 0, act = GEMM_ACT_NONE;
    // first-stage QKV epilogue (qkv_dim > 0): columns [0,dim) -> out_bf16 [M][dim]; K and V -> cache at position m % seq_len
    int32_t qkv_dim = 0, seq_len = 0, n_head = 0, cache_len = 0;
    bf16_t* kcache = nullptr;
    bf16_t* vcache = nullptr;
    // split-K (few rows, e.g. the M = 2T = 256 of a prefill: N / 64 x M / 64 tiles alone leave most CUs with one small block whose
    // every K-step is a dependent load round trip): `slab` = device scratch of gemm_slab_elems(M, N, gated) floats; launch_gemm then
    // cuts K into up to 8 ranges (blockIdx.z), the blocks write fp32 partial tiles and gemm_combine_kernel adds them in range order
    // and applies the epilogue.  Null: one range, epilogue in the GEMM kernel.
    float* slab = nullptr;
    size_t slab_elems = 0;          // capacity of `slab` in floats: launch_gemm never asks a kernel for more (ranges x M x N x (gated ? 2 : 1))
    int32_t splitk = 1;             // set by launch_gemm
};
hipError_t launch_gemm(hipStream_t s, const GemmArgs& a);
#define GEMM_TILE 64
inline size_t gemm_slab_elems(int M, int N, bool gated) { return (size_t)8 * M * N * (gated ? 2 : 1); }

struct NormArgs {
    const bf16_t* x_bf16 = nullptr;
    const float* x_f32 = nullptr;   // exactly one of x_bf16 / x_f32
    const bf16_t* w_bf16 = nullptr;
    const float* w_f32 = nullptr;
    const float* b_f32 = nullptr;   // LayerNorm bias or null
    bf16_t* out = nullptr;          // [M][C]
    int32_t M = 0, C = 0, mode = 0;
    float eps = 1e-5f;
};
hipError_t launch_norm_rows(hipStream_t s, const NormArgs& a);

struct AttnRowsArgs {
    const bf16_t* q = nullptr;      // query (b, t, h) at q + (b*T + t)*q_row_stride + h*head_dim
    const bf16_t* k = nullptr;      // key (b, h, j)   at k + b*k_batch_stride + h*k_head_stride + j*k_key_stride
    const bf16_t* v = nullptr;
    bf16_t* out = nullptr;          // (b, t, h) at out + (b*T + t)*out_row_stride + h*head_dim
    int64_t k_batch_stride = 0, k_head_stride = 0, k_key_stride = 0;
    int32_t q_row_stride = 0, out_row_stride = 0;
    int32_t T = 0, n_head = 0, batch = 0, head_dim = 128, causal = 0;
    bf16_t* vt_scratch = nullptr;   // device scratch of attn_rows_scratch_elems(batch, n_head, head_dim, T) bf16 for V transposed; with it the
                                    // MFMA kernel runs (attn_mfma_kernel), without it the one-query-per-block VALU kernel
};
hipError_t launch_attn_rows(hipStream_t s, const AttnRowsArgs& a);
inline size_t attn_rows_scratch_elems(int batch, int n_head, int head_dim, int T) { return (size_t)batch * n_head * head_dim * ((T + 63) / 64 * 64); }

// ---- lstm.hip: persistent LSTM recurrence (EnCodec decoder LSTMs, speaker encoder) -----------------------------------------
struct LstmSeqArgs {
    const float* gin = nullptr;      // W_ih x_t + b_ih + b_hh of (b, t, gate row r) at gin[b*gs_b + t*gs_t + r*gs_r], rows ordered i|f|g|o
    int64_t gs_b = 0, gs_t = 0, gs_r = 0;
    const float* whh = nullptr;      // [4H][H]
    float* y = nullptr;              // h_t[u] (+ skip) of (b, t, u) at y[b*ys_b + t*ys_t + u*ys_u], or null
    int64_t ys_b = 0, ys_t = 0, ys_u = 0;
    const float* skip = nullptr;     // same indexing as y, or null
    float* h_last = nullptr;         // [B][H]: h_T, or null
    unsigned long long* hx = nullptr;  // [2][B][H] tagged granules (device scratch, zeroed once)
    uint32_t tag_base = 0;           // launch serial << 16: tags of different launches never collide
    uint32_t* abort_flag = nullptr;  // device word, raised when a wait times out
    int32_t B = 1, H = 0, T = 0;
};
hipError_t launch_lstm_seq(hipStream_t s, const LstmSeqArgs& a);   // H % 8 == 0, T <= 65535, LDS (32 H + B (H + 8)) * 4 <= 160 KB
size_t lstm_seq_lds_bytes(int B, int H);
hipError_t launch_lstm_inproj(hipStream_t s, const float* x, const float* w, const float* bias, float* gin, int BT, int I, int R);
// Two stacked LSTM layers (batch 1, same H) in ONE persistent launch: layer 2's step t runs right behind layer 1's step t, so the
// chain is T + 1 dependent steps instead of 2 T (the EnCodec decoder's LSTM pair: 960 frames, 2 x 4.7 ms -> ~5 ms).
struct LstmSeq2Args {
    const float* gin1 = nullptr;     // layer 1: W_ih x_t + b of (t, gate row r) at gin1[t * gs_t + r * gs_r]
    int64_t gs_t = 0, gs_r = 0;
    const float* whh1 = nullptr;     // [4H][H]
    const float* wih2 = nullptr;     // layer 2 input weights, element (row r, column k) at wih2[k * wih2_ks + r]
    int64_t wih2_ks = 0;
    const float* bias2 = nullptr;    // [4H]: b_ih + b_hh of layer 2
    const float* whh2 = nullptr;     // [4H][H]
    float* y = nullptr;              // layer 2's h_t[u] (+ skip) at y[t * ys_t + u * ys_u]
    int64_t ys_t = 0, ys_u = 0;
    const float* skip = nullptr;     // same indexing as y, or null
    unsigned long long* hx1 = nullptr;   // [2][H] tagged granules, the recurrence of layer 1
    unsigned long long* hx2 = nullptr;   // [2][H] tagged granules, the recurrence of layer 2
    unsigned long long* seq1 = nullptr;  // [T][H] tagged granules: layer 1's outputs, read by layer 2 at its own pace
    uint32_t tag_base = 0;
    uint32_t* abort_flag = nullptr;
    int32_t H = 0, T = 0;
};
bool lstm_stepwise_mode();    // true once a persistent LSTM launch has timed out in this process: launch_lstm_seq then runs one launch per step
void lstm_force_stepwise();
hipError_t launch_lstm_seq2(hipStream_t s, const LstmSeq2Args& a);   // H % 8 == 0, (64 H + 2 H + 8) * 4 bytes of LDS <= 160 KB

// ---- decode_mega.hip: persistent decode-step kernel (MetaVoice-1B shape, one utterance) -----------------------------
struct MegaLayer {
    const bf16_t *attn_norm, *wqkv, *wo, *ffn_norm, *w1, *w3, *w2;
    // int4 (group-128) engines: packed codes and (scale, zero) words of wqkv, wo, w1, w3, w2 (gemv_i4.hip layout), else null
    const unsigned char* q[5];
    const uint32_t* sz[5];
    // int8 (per output channel) engines: codes [N][K] and bf16 row scales, same order (gemv_i8.hip layout), else null
    const signed char* q8[5];
    const bf16_t* s8[5];
    // engines of 4 utterances (bf16): the same five matrices TILED for the matrix cores (decode_mfma.hip: per workgroup [K / 32][rows][32]), else null
    const bf16_t* t[5];
};
struct MegaArgs {
    const bf16_t* x_in = nullptr;          // [2 n_utt][2048]: input rows of layer 0 (written by the sampler / prefill)
    const MegaLayer* layers = nullptr;     // device array [n_layer]
    int32_t n_layer = 0, vocab = 0, cache_len = 0;
    int32_t n_utt = 1;                     // utterances decoded together (1, 2 or 4): rows 2u, 2u+1 belong to utterance u
    const bf16_t* final_norm = nullptr;
    const bf16_t* head = nullptr;
    const bf16_t* head_t = nullptr;        // decode_mfma.hip: the LM head tiled like MegaLayer::t (11 rows per workgroup, rows clamped to the vocabulary)
    const signed char* head_q8 = nullptr;  // int8 engines: the LM head is quantised too
    const bf16_t* head_s8 = nullptr;
    bf16_t* logits = nullptr;              // [2 n_utt][vocab]
    bf16_t* kcache = nullptr;
    bf16_t* vcache = nullptr;
    size_t kv_layer_elems = 0;
    const int32_t* pos = nullptr;          // [n_utt]
    const uint32_t* step_counter = nullptr;
    // granule buffers, each [n_utt][columns]: x 2048, qkv 6144, attn 2048, h 2048, act 5632
    unsigned long long *g_x = nullptr, *g_qkv = nullptr, *g_attn = nullptr, *g_h = nullptr, *g_act = nullptr;
    float* partial = nullptr;
    uint32_t* counters = nullptr;
    uint32_t* abort_flag = nullptr;
    float eps = 1e-5f;
    bf16_t* dbg = nullptr;                 // optional [n_layer][5 vectors: q|attn|h|act|x][2][5632] (mv_debug_vectors)
    unsigned long long* stamps = nullptr;  // optional [2 blocks][n_layer][24] s_memtime stamps of wave 0 (MV_MEGA_STAMPS)
    int32_t ablate = 0;                    // MvTune::gemv_ablate bit 0: no weight requests (profiling only)
    int32_t fault = 0;                     // MvTune::mega_fault (test hook)
    // ---- sampler inside the launch (n_steps >= 1 decode steps per launch; 0 = one step, logits only, sample_kernel follows) ----
    // The workgroup of utterance u (block u) polls the step's logits, draws the token (sample_core.h), does the bookkeeping of
    // decode_n_tokens and publishes a control word (token | position | done) from which every workgroup builds the next input rows.
    int32_t n_steps = 0;
    int32_t block_size = 0, capture_max = 0;
    unsigned long long *g_logits = nullptr, *g_ctl = nullptr;  // [n_utt][vocab padded to even], [n_utt]: {token | position << 12 | done << 31}
    UttState* utt = nullptr;
    int32_t* pos_out = nullptr;            // = pos (mirror of utt[].pos for the per-layer launches)
    int32_t* seq = nullptr;
    const bf16_t *tok_emb = nullptr, *pos_emb = nullptr, *spk_proj = nullptr;
    bf16_t* x_out = nullptr;               // = x_in: the next launch's (or the per-layer launches') input rows
    int32_t* n_active = nullptr;
    float* capture = nullptr;
    uint32_t* step_counter_out = nullptr;  // = step_counter: advanced once per step by block 0
};
// ---- decode_mfma.hip: the persistent step of 4-utterance bf16 engines on v_mfma_f32_16x16x32_bf16 --------------------------------------
bool mfma_decode_supported(int weight_format, int n_utt);
bool mfma_decode_fits(int n_utt, int n_cus);                       // occupancy query, like mega_fits
size_t mfma_tiled_elems(int K, int rows_per_wg);                   // elements of a tiled matrix copy (256 workgroups x rows x K)
hipError_t launch_tile_weights(hipStream_t s, const bf16_t* src, bf16_t* dst, int N, int K, int rows_per_wg, const bf16_t* col_scale = nullptr);
void mfma_handoff_rows(int n_utt, size_t* offset_bytes, size_t* bytes);   // the tag-free hand-off rows inside the granule slab: 0xFF-filled by mv_begin
hipError_t launch_decode_mfma(hipStream_t s, const MegaArgs& a);   // one decode step (logits only; sample_kernel follows)
bool mega_supported(const mv_config& c, int rows, int n_cus);
bool mega_fits(int weight_format, int n_utt, int n_cus);  // occupancy query: one workgroup of the instance per CU, 256 resident (asked once, at mv_finalize)
bool mega_instrumented();  // decode_mega.hip was built with -DMEGA_INSTR=1 (phase stamps, debug vectors, the no-weights ablation)
bool mega_folds(const mv_config& c);  // the persistent kernel of this configuration draws its own tokens (several steps per launch)
hipError_t launch_decode_mega(hipStream_t s, const MegaArgs& a, int weight_format);  // every wave prefetches its weight rows into registers; bf16 or int4
hipError_t launch_bump_counter(hipStream_t s, uint32_t* counter);  // mv_probe_kernel: advances the hand-off epoch like the sampler does
