/* ser_hip.h -- C ABI of libserhip.so: hand-written gfx950 (MI355X / CDNA4) kernels
 * for the SSL embedding-extraction hot path of AI-Unicamp/interspeech_ser.
 *
 * The reference has no FFI of its own on this path: preprocessing/
 * preprocess_speech.py:49-50,66-67 calls  model(**inputs, output_hidden_states=True)
 * and preprocessing/preprocess_whisper.py:57,71 calls model.encoder(input_features,
 * output_hidden_states=True); every device op below replaces one implicit
 * cuDNN/cuBLAS/ATen launch inside those two calls (SURVEY.md 2.3 rows K1..K15,
 * cited per entry point).  INTEGRATION.md shows the ctypes binding a reference
 * maintainer would add.
 *
 * Conventions
 *   - plain C, no C++/torch types; every pointer is a DEVICE pointer unless named host_*;
 *   - the caller owns all memory; launchers never allocate, free or synchronise;
 *   - every launcher enqueues on `stream` (a hipStream_t passed as void*) and returns
 *     0 on success, <0 for an argument/shape error, >0 = hipError_t of the launch;
 *     ser_last_error() gives the thread-local message of the last non-zero return;
 *   - launchers are re-entrant (the reference drives its model from 4 Python threads,
 *     preprocess_speech.py:120-122).
 *
 * Data layout (DESIGN.md "HBM layout")
 *   - utterances are PACKED, never padded: a ragged batch is one [rows, C] matrix
 *     whose utterance b owns rows frame_offs[b] .. frame_offs[b+1]-1;
 *   - "act" tensors feed matrix-core GEMMs: bf16, row-major, 1 plane (SER_MODE_BF16) or
 *     2 planes hi/lo with x ~= hi + lo (SER_MODE_FP32X, the 3-product split that gives
 *     fp32-grade results on the bf16 MFMA pipe), 1 plane of fp16 (SER_MODE_FP16) or 2 planes of fp16 (SER_MODE_FP16X);
 *     plane p lives at base + p*plane_stride;
 *   - the residual stream / hidden states are fp32 row-major [rows, D].
 */
#ifndef SER_HIP_H
#define SER_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SER_ABI_VERSION 14

#define SER_MODE_BF16  1   /* act tensors have 1 plane; GEMMs do 1 bf16 MFMA product   */
#define SER_MODE_FP32X 2   /* act tensors have 2 planes; GEMMs do hi*hi + lo*hi + hi*lo */
#define SER_MODE_FP16  3   /* act tensors have 1 plane of IEEE fp16 (11 significand bits, saturated at +-65504);
                            * GEMMs do 1 f16 MFMA product -- same rate as bf16, 8x finer operand rounding.  The host
                            * runs the conv stem in FP32X and the encoder layers in FP16 ("f16" numerics mode) */
#define SER_MODE_FP16X 4   /* act tensors have 2 planes of IEEE fp16, x ~= hi + lo (22 significand bits; hi is exactly the FP16 copy, so
                            * a single-product FP16 launch may read plane 0 of an FP16X tensor); GEMMs and ser_attention do
                            * hi*hi + lo*hi + hi*lo on the f16 MFMA.  The host's "f16a" numerics mode runs the whole ATTENTION BLOCK of
                            * every encoder layer in it (packed projection, attention, output projection) and the feed-forward pair in
                            * FP16; "f16q" only the LOGIT path -- the q / k (+ WavLM gate) columns of the packed projection and, through
                            * SER_MODE_FP16Q, S = K Q^T: a softmax weight moves by (logit error) * ln 2, so the attention block is where
                            * single-product rounding is amplified (reference arithmetic is fp32: preprocess_speech.py:50,66) */
#define SER_MODE_FP16Q 5   /* ser_attention only: q and k are FP16X column blocks (3-product S = K Q^T), v is read from plane 0 and
                            * P V runs single fp16 products; the output is a single-plane FP16 tensor */

#define SER_MODE_FP16M 6   /* round 5, the "f16m" numerics mode: fp16 main product + block-scaled 8-bit cross terms on gfx950's
                            * v_mfma_scale_f32_16x16x128_f8f6f4 -- x W^T ~= x_hi w_hi + x_lo w_8 + x_8 w_lo, i.e. 1 + 1/2 + 1/2 = 2 fp16 product-
                            * equivalents per algorithmic FLOP instead of FP16X's 3 (the scaled e4m3 instruction runs at twice the fp16 rate).
                            * An FP16M tensor has
                            *   plane 0 (at base):                fp16 hi = fp16(x), [rows][ld], exactly the FP16 copy;
                            *   plane 1 (at base + plane_stride): the CROSS-TERM plane, same pitch in bytes: for every row and every 64-column tile t,
                            *       the 128 bytes at byte offset 2 * (row * ld + 64 t) hold [P: 64 e4m3 bytes, columns 64t .. 64t+63][Q: 64 e4m3 bytes];
                            *       activations: P = x - hi (the fp16 rounding residual), Q = x;   weights: P = w, Q = w - fp16(w);
                            *   scales (their own pointer): uint32 [ld / 64][scale_ld], word (t, row) = four E8M0 codes (value 2^(c - 127)), one per
                            *       32 consecutive elements: [P cols 0-31, P cols 32-63, Q cols 0-31, Q cols 32-63] of tile t -- the OCP MX block format,
                            *       the smallest power of two that brings the block's largest magnitude to <= 448.
                            * ser_gemm accumulates, per 64-deep K tile, two fp16 MFMA k-steps (hi x hi) and ONE scaled e4m3 MFMA whose 128-long K is
                            * [P | Q] of both operands: P_a P_w + Q_a Q_w = x_lo w_8 + x_8 w_lo.  The cross terms carry 2^-11 of the result, so their
                            * 4-bit significands leave an operand error of ~2^-15 (oracle/numerics_whatif_f16m.py).  Range: fp16's (saturation at
                            * +-65504, reported through range_flag).  Reference arithmetic is fp32 end to end (preprocess_speech.py:50,66). */

#define SER_ACT_NONE 0
#define SER_ACT_GELU 1     /* exact erf GELU (ACT2FN["gelu"]) */

int         ser_version(void);
const char* ser_last_error(void);

/* K1  zero_mean_unit_var_norm (HF feature_extraction_wav2vec2.py:77-97; call site
 * preprocess_speech.py:48).  wav/out: packed fp32 samples; sample_offs[B+1] (device). */
int ser_wave_norm(const float* wav, const int64_t* sample_offs, int B, float* out, void* stream);

/* K1 + framing for the matrix-core form of conv layer 0: normalise each utterance (as ser_wave_norm)
 * and write frame t as one act row [x_n[stride*t .. stride*t+k-1], 0, ...] of 64 elements, so that
 * Conv1d(1,C,k,stride)+LayerNorm+GELU (HF modeling_wavlm.py:696-720) is ser_gemm with K = 64 and the
 * LayerNorm epilogue.  out: act [total_rows, 64]; work: ser_workspace_bytes(SER_WS_WAVE_FRAMES, B, ...). */
int ser_wave_frames(const float* wav, const int64_t* sample_offs, const int32_t* frame_offs, int B,
                    int k, int stride, void* out, int64_t out_plane_stride, int mode, void* work,
                    int total_rows, void* stream);

/* K2  conv layer 0: Conv1d(1,C,k,stride) [+bias] -> LayerNorm(C) -> GELU
 * (HF modeling_wavlm.py:696-720).  One output row per frame.
 * w: [C,k] fp32; bias may be NULL; out: act [rows,C]. C multiple of 64, C <= 1024, k <= 16. */
int ser_conv0_ln_gelu(const float* wav_norm, const int64_t* sample_offs, const int32_t* frame_offs,
                      int B, const float* w, const float* bias, const float* ln_g, const float* ln_b,
                      void* out, int64_t out_plane_stride, int mode,
                      int C, int k, int stride, int total_rows, void* stream);

/* K3/K4/K5/K7/K10/K11/K12/K14  one matrix-core GEMM with an implicit-convolution
 * row map and a fused epilogue:   C[m,n] = sum_k A(m,k) * W[n,k]
 *   A(m,k)  = A[ a_row(m)*8 + (k / kc)*ldj + (k % kc) ]   (kc == 0: plain row-major, lda)
 *   a_row(m)= a_rowoff ? a_rowoff[m] : m*lda/8             (units of 8 elements = 16 bytes)
 * so that a strided Conv1d over a channels-last [frames, C] matrix (HF modeling_wavlm.py:
 * 696-720, modeling_whisper.py:618-619), the grouped positional conv (:48-90) and every
 * nn.Linear (:133-136, :288-294) are the same kernel.  Epilogue order:
 *   v = acc + bias[n];  v = act(v);  v += residual[(m % res_row_mod or m)*ldr + n];
 *   out_f32[m*ldo_f32 + n] = v;   out_act[out_row(m)*ldo_act + n] = split(v)
 * With ln_gamma != NULL the epilogue is  v = act(LayerNorm_row(acc + bias)) instead.
 * Requirements: K % 64 == 0, kc % 64 == 0, N % 8 == 0, all row starts 16-byte aligned. */
typedef struct ser_gemm_args {
    const void*    A;              /* act (bf16 planes) */
    int64_t        a_plane_stride; /* elements between hi and lo plane (FP32X) */
    const int32_t* a_rowoff;       /* [M] or NULL */
    int64_t        lda;            /* elements, used when a_rowoff == NULL */
    int32_t        kc;             /* K-chunk length of the conv map, 0 = none */
    int64_t        ldj;            /* element step between K-chunks (conv tap stride) */
    const void*    W;              /* bf16 [groups][N][K] (+ lo plane) */
    int64_t        w_plane_stride;
    int32_t        M, N, K;
    int32_t        groups;         /* blockIdx.y; 1 for dense */
    int64_t        a_group_stride; /* elements added to A per group (column offset) */
    int64_t        w_group_stride; /* elements per group in W */
    int32_t        c_group_stride; /* output columns per group */
    int32_t        mode;           /* SER_MODE_* */
    const float*   bias;           /* [groups*N] or NULL */
    int32_t        act;            /* SER_ACT_* */
    const float*   residual;       /* fp32 or NULL */
    int64_t        ldr;
    int32_t        res_row_mod;    /* 0: row m; >0: row m % res_row_mod (Whisper positions) */
    float*         out_f32;        /* may be NULL */
    int64_t        ldo_f32;
    void*          out_act;        /* may be NULL */
    int64_t        ldo_act;
    int64_t        out_plane_stride;
    const int32_t* out_rowmap;     /* [M] row index in out_act, or NULL (identity) */
    /* optional fused LayerNorm over the full output row (conv stack: Conv1d -> LayerNorm(C) -> GELU,
     * HF modeling_wavlm.py:712-719): v = LN(acc + bias) * gamma + beta, then act.  Needs N <= 512,
     * groups == 1, no residual. */
    const float*   ln_gamma;       /* [N] or NULL */
    const float*   ln_beta;        /* [N] */
    float          ln_eps;
    int32_t        tile_cfg;       /* 0 = auto; 1 = 128x128, 2 = 256x128, 3 = 256x256 block tile */
    /* DEFERRED LayerNorm of the A operand (encoder layers: LN -> Linear, HF modeling_wavlm.py:357-358,
     * 366): the LayerNorm kernel and its HBM round trip disappear.  With W' = W * gamma (folded at load),
     *   LN(x) W^T + b = rstd_m * (x W'^T - mu_m * colsum(W')_n) + (beta W^T + b)_n
     * A holds the raw (un-normalised) rows; mu/rstd come from per-row partial sums (sum, sum of squares
     * per 64-column group) that the PRODUCER of x wrote through stat_out.  bias must hold beta W^T + b. */
    const float*   ln_stats_in;    /* [M][ln_groups][2] or NULL */
    int32_t        ln_groups;      /* even */
    const float*   ln_colsum;      /* [N] */
    float*         stat_out;       /* [M][stat_groups][2] row partials of the values written, or NULL */
    int32_t        stat_groups;    /* >= N/64 */
    int32_t        f32_col_begin;  /* out_f32 receives only columns >= f32_col_begin (stored at n - f32_col_begin) */
    /* columns n < col_scale_end are multiplied by col_scale after bias (before act): the packed QKV projection
     * scales q by dh^-0.5 (what HF does before the product, modeling_whisper.py:309) times log2(e), so the
     * attention kernel's softmax runs in the exp2 domain with no per-score multiply. */
    float          col_scale;
    int32_t        col_scale_end;  /* multiple of 4; 0 = no scaling */
    /* SHIFTED operand copy for the deferred LayerNorm.  Real checkpoints carry offsets in the residual stream (rows
     * whose mean is many standard deviations); rounding such a row to bf16 and then forming acc - mu*colsum loses the
     * signal.  With shift_out != NULL a PRODUCER launch takes, per row,
     *   c[m] = shift_const + (shift_in ? shift_in[m] : 0)
     * -- shift_in = the absolute row mean of its RESIDUAL input, shift_const = a load-time constant (mean of the bias) --
     * stores it in shift_out[m] and writes out_act and stat_out for v - c[m] (out_f32 keeps v).  LayerNorm is shift
     * invariant, so the CONSUMER only has to report the absolute mean for the next producer down the residual stream:
     * mean_out[m] = mu_m + (ln_shift ? ln_shift[m] : 0), where mu_m is the mean it derives from ln_stats_in (relative
     * to the shift ln_shift its A rows were stored with). */
    const float*   shift_in;       /* [M] absolute row mean of the residual rows, or NULL (0) */
    float*         shift_out;      /* [M] or NULL (no shifting) */
    float          shift_const;
    int32_t        out_mode;       /* format of out_act: 0 = mode; SER_MODE_FP16 with mode == SER_MODE_FP32X converts (one plane of
                                    * fp16 written from a 3-product GEMM: the stem -> layers boundary of the host's "f16" mode);
                                    * SER_MODE_FP16X with mode == SER_MODE_FP16 writes hi + lo fp16 planes from a single-product
                                    * GEMM (FC2 -> the next layer's packed projection in the "f16q" / "f16a" modes); SER_MODE_FP16 with
                                    * mode == SER_MODE_FP16X writes one plane from a 3-product GEMM (output projection -> FC1, "f16a") */
    const float*   ln_shift;       /* [M] shift of the A rows / their partials (consumer), or NULL */
    float*         mean_out;       /* [M] absolute row mean of the A rows (consumer), or NULL */
    float*         lnstat_out;     /* [M][2] (row mean relative to ln_shift, 1/sqrt(var + eps)) the consumer derived from ln_stats_in, or
                                    * NULL: what ser_attention's in-kernel WavLM gate (ser_attention_args.gate_x) applies to the same rows */
    /* SER_MODE_FP16M operands (ABI 13): block scales of A (row index m; needs a_rowoff == NULL, kc == 0, groups == 1) and of W (row index n),
     * and -- with out_mode == SER_MODE_FP16M -- of the out_act copy (row index out_row(m), tile index (output column) / 64; the first output
     * column of the launch must be a multiple of 64).  *_scale_ld = words between consecutive 64-column tiles. */
    const uint32_t* a_scale;  int64_t a_scale_ld;
    const uint32_t* w_scale;  int64_t w_scale_ld;
    uint32_t*       out_scale; int64_t out_scale_ld;
    /* fp16 range guard (ABI 13): when not NULL, the launch ORs into *range_flag bit 0 if any value it rounds to an fp16 operand plane
     * (out_act in the FP16 / FP16X / FP16M formats) exceeds 65504 in magnitude (or is a NaN) BEFORE the saturating conversion, bit 1 if
     * one exceeds half that -- the host reads the word back with the batch's features and fails that batch's files instead of writing
     * clipped ones (preprocess_speech.py:46,72-73: a bad file is a printed failure, never silent garbage).  ser_layernorm_v,
     * ser_row_center_v, ser_wave_frames_v, ser_pack_act_v and ser_pack_f16m take the same word. */
    uint32_t*       range_flag;
} ser_gemm_args;
int ser_gemm(const ser_gemm_args* args, void* stream);

/* K6  LayerNorm over the last dim (+ optional GELU), fp32 in
 * (HF modeling_wavlm.py:357,366,513; conv-stack LN :716-718).  Either output may be NULL. */
int ser_layernorm(const float* x, int64_t ldx, const float* g, const float* b, float eps, int gelu,
                  float* out_f32, int64_t ldo_f32, void* out_act, int64_t ldo_act, int64_t out_plane_stride,
                  int mode, int rows, int D, void* stream);

/* Centred operand copy of hidden_states[0] for the deferred LayerNorm of encoder layer 0 (the LayerNorm itself is
 * HF modeling_wavlm.py:357): out_act = split(x - mean_row), stats[m][0] = (sum, sum of squares) of the centred row
 * (remaining slots zero), shift[m] = mean_row.  See ser_gemm_args.shift_out. */
int ser_row_center(const float* x, int64_t ldx, void* out_act, int64_t ldo_act, int64_t out_plane_stride,
                   float* stats /*[rows][stat_groups][2]*/, int stat_groups, float* shift /*[rows]*/,
                   int mode, int rows, int D, void* stream);

/* K8a WavLM relative-position bias as a [H, 2*T-1] table: column (key-query)+(T-1)
 * (compute_bias + _relative_positions_bucket, HF modeling_wavlm.py:243-271). */
int ser_wavlm_bias_table(const float* rel_attn_embed /*[num_buckets,H]*/, float* table /*[H,2T-1]*/,
                         int T, int H, int num_buckets, int max_distance, void* stream);

/* K8b WavLM GRU gate (HF modeling_wavlm.py:167-180): gate[row,h] from the layer-normed act x. */
int ser_wavlm_gate(const void* x_ln, int64_t ldx, int64_t plane_stride, int mode,
                   const float* w8 /*[8,dh]*/, const float* b8 /*[8]*/, const float* gru_const /*[H]*/,
                   float* gate /*[rows,H]*/, int rows, int H, int dh, void* stream);

/* K9  fused attention over a packed ragged batch:
 *   out[q,:] = softmax_k( q.k * scale + gate[q,h] * table[h, k-q+table_T-1] ) v
 * q/k/v are column blocks of one act matrix [rows, ld] (packed QKV projection output);
 * utterance b owns rows frame_offs[b] .. frame_offs[b+1]-1 and attends only to itself
 * (batch-of-one semantics of preprocess_speech.py:76-81, so no key-padding mask exists).
 * WavLM: HF modeling_wavlm.py:188-241; wav2vec2/HuBERT: modeling_wav2vec2.py:438-548;
 * Whisper: modeling_whisper.py:284-357 (pass scale = dh^-0.5, identical in exact arithmetic).
 * scale <= 0 means q is PRE-SCALED by dh^-0.5 * log2(e) (ser_gemm col_scale): scores are used as exp2 exponents.
 * dh in {64, 80, 96, 120, 128}; table/gate NULL for plain attention.
 * The WavLM gate is given either as gate[rows,H] (from ser_wavlm_gate) or, fused, as its two
 * pre-activations per head stored in columns gate_col + 2h, +1 of the qkv matrix (extra output
 * columns of the packed projection GEMM) together with gru_const[H]. */
int ser_attention(const void* qkv, int64_t ld, int64_t plane_stride, int q_col, int k_col, int v_col,
                  const int32_t* frame_offs /*[B+1] device*/, int B, int max_frames,
                  const float* table, int table_T, const float* gate,
                  void* out, int64_t ldo, int64_t out_plane_stride,
                  int H, int dh, float scale, int mode,
                  int gate_col, const float* gru_const /*[H]*/,
                  const int32_t* key_lens /*[B] or NULL: keys >= key_lens[b] are padding (RoBERTa attention_mask)*/,
                  const float* bias2d /*[B][H][max_frames][bias2d_ld] fp32 dense additive bias in the exp2 domain, or NULL
                                        (DeBERTa, from ser_deberta_bias): uniform-length batches, q pre-scaled, dh <= 64, modes BF16 / FP32X / FP16X; padded
                                        query rows (q >= key_lens[b]) then come out as the uniform average of all value rows*/,
                  int64_t bias2d_ld,
                  void* stream);

/* next row 8f-1 (text side): RoBERTa embeddings word[id] + position[cumsum(non-pad)] + token_type[0] -> LayerNorm
 * (HF modeling_roberta.py:56-155; call site preprocessing/preprocess_roberta.py:47-57).  ids: [B,T] int32.
 * mode (here and in ser_embed_ln_masked / ser_pack_rows / ser_zero_padded_rows): the format of the operand copy, BF16 / FP32X / FP16X. */
int ser_embed_ln(const int32_t* ids, const float* word_emb, const float* pos_emb, const float* type_emb,
                 const float* ln_g, const float* ln_b, float eps, float* out_f32, void* out_act,
                 int64_t out_plane_stride, int mode, int B, int T, int D, int pad_id, void* stream);

/* K13 Whisper log-mel front end (HF feature_extraction_whisper.py:135-169): packed fp32
 * samples -> [B, n_mels, 3000] fp32 (zero-pad/truncate to 480000, reflect pad, Hann,
 * 400-pt DFT power, mel, log10, per-utterance max-8 floor, (x+4)/4).
 * mel: [201, n_mels] fp32.  work: ser_workspace_bytes(SER_WS_LOGMEL, B, ...) bytes, prepared ONCE per buffer by
 * ser_logmel_init (the DFT twiddle table lives there; the per-utterance maxima are per-block partials reduced by the
 * finishing pass: no atomics, nothing to reset between calls). */
int ser_logmel_init(void* work, int B, void* stream);
int ser_logmel_whisper(const float* wav, const int64_t* sample_offs, int B, const float* mel, int n_mels,
                       float* out, void* work, void* stream);

/* fp32 [rows, C] -> act (bf16 / hi+lo planes), optional transpose of a [B, C, T] input
 * into channels-last rows with `halo` zero rows before and after each utterance
 * (feeds the Whisper stem convs, HF modeling_whisper.py:618-619). */
int ser_pack_act(const float* x, int B, int C, int T, int halo, void* out, int64_t ldo,
                 int64_t out_plane_stride, int mode, void* stream);

/* DeBERTa-v2/v3 variant of the text side (preprocessing/preprocess_deroberta.py:106-107 builds it with AutoModel).
 * ser_embed_ln_masked: LayerNorm(word_emb[id]) with rows t >= key_lens[b] zeroed (HF modeling_deberta_v2.py
 * DebertaV2Embeddings in the v3 configuration: no absolute positions, no token types, embeddings * mask).
 * ser_deberta_attention: softmax((Qc Kc^T + c2p + p2c) / sqrt(3 dh)) V per (sequence, head)
 * (DisentangledSelfAttention.forward / disentangled_attention_bias).  c2p, p2c: [B*T, ldp] fp32, head h in columns
 * h*Nr .. h*Nr+Nr-1 = content queries / keys times the position keys / queries (shared q/k projections of the
 * LayerNorm-ed relative embeddings) restricted to the Nr relative-position rows a T-token sequence reaches;
 * c2p_col / p2c_col: [2T-1] int32, column inside that window for signed distance d at index d + T - 1
 * (c2p reads c2p[q][c2p_col[q-k]], p2c reads p2c[k][p2c_col[k-q]]; the log-bucket map is built by the host).
 * A (query, key) pair counts only if both are real tokens (t < key_lens[b]); a padded query row comes out as the
 * uniform average of all T value rows, as HF's masked_fill(finfo.min) + softmax gives.  T <= 128, dh <= 64.
 * (ser_deberta_attention is the one-thread-per-query statement of the op, kept as the kernel-level reference; the encoder
 * runs ser_deberta_bias + ser_attention(bias2d), which has no 128-token limit.) */
int ser_embed_ln_masked(const int32_t* ids, const float* word_emb, const float* ln_g, const float* ln_b, float eps,
                        const int32_t* key_lens, float* out_f32, void* out_act, int64_t out_plane_stride,
                        int mode, int B, int T, int D, void* stream);
/* ConvLayer of deberta-v2-xlarge / xxlarge (HF modeling_deberta_v2.py ConvLayer; the checkpoint the reference's README runs
 * preprocess_deroberta.py with, README.md:66): Conv1d(D, D, 3) over the token axis of the embedding output, activation, + layer
 * 0's output, LayerNorm, padded rows zero.  The conv is ser_gemm's implicit-conv map over the halo'd copy ser_pack_rows makes
 * (row b*(T+2*halo) + halo + t of `out` = split(x[b*T + t]); the halo rows are the caller's zeros); ser_zero_padded_rows zeroes
 * rows t >= key_lens[b] of an fp32 matrix and / or its act copy. */
int ser_pack_rows(const float* x, int64_t ldx, int B, int T, int D, int halo, void* out, int64_t ldo, int64_t out_plane_stride,
                  int mode, void* stream);
int ser_zero_padded_rows(float* x, int64_t ldx, void* act, int64_t lda, int64_t plane_stride, int mode, const int32_t* key_lens,
                         int B, int T, int D, void* stream);

/* Dense disentangled-attention bias for ser_attention's bias2d argument (the matrix-core path the DeBERTa encoder uses):
 *   out[b][h][q][k] = c2p[q][c2p_col[q-k]] + p2c_scale * p2c[k][p2c_col[k-q]]   for q, k < key_lens[b], else 0;   row pitch ld.
 * c2p is expected from a q that already carries the score scale (ser_gemm col_scale), p2c from the plain k. */
int ser_deberta_bias(const float* c2p, const float* p2c, int64_t ldp, int Nr, const int32_t* c2p_col, const int32_t* p2c_col,
                     const int32_t* key_lens, float* out, int64_t ld, int B, int T, int H, float p2c_scale, void* stream);
int ser_deberta_attention(const void* qkv, int64_t ld, int64_t plane_stride, int q_col, int k_col, int v_col,
                          const float* c2p, const float* p2c, int64_t ldp, int Nr, const int32_t* c2p_col,
                          const int32_t* p2c_col, const int32_t* key_lens, void* out, int64_t ldo,
                          int64_t out_plane_stride, int B, int T, int H, int dh, int mode, void* stream);

/* K15 mean of 4 fp32 states (--use_average y; preprocess_speech.py:52-63). */
int ser_mean4(const float* s0, const float* s1, const float* s2, const float* s3, float* out,
              int64_t n, void* stream);

/* weights: fp32 -> bf16 hi (+ lo) planes, done once at load. */
int ser_split_bf16(const float* x, void* out, int64_t plane_stride, int mode, int64_t n, void* stream);

/* fp32 [rows][ldx] -> a SER_MODE_FP16M tensor (see the mode's comment): hi plane at out, cross-term plane at out + plane_stride (elements),
 * scale words at scales[t * scale_ld + row].  cols % 64 == 0, ldo % 64 == 0.  is_weight selects the plane roles (weights: P = w, Q = w - hi;
 * activations: P = x - hi, Q = x).  Used at load for the weights (HF modeling_wavlm.py:133-136,288-294 Linear weights) and by the kernel tests. */
int ser_pack_f16m(const float* x, int64_t ldx, int rows, int cols, void* out, int64_t ldo, int64_t plane_stride,
                  uint32_t* scales, int64_t scale_ld, int is_weight, uint32_t* range_flag, void* stream);

/* Row-index tables of a packed ragged batch, built on the device (no per-batch host tables to upload):
 *   out[m] = (base[b] + step * (m - row_offs[b])) * mult / div      for row_offs[b] <= m < row_offs[b+1]
 * row_offs: [B+1] int32 first output row of every utterance; base: [B] int64.  This is the implicit-conv row offset
 * table ser_gemm_args.a_rowoff of a strided Conv1d over packed utterances (base = first input row, step = stride,
 * mult = C_in, div = 8), the positional conv's halo map, and every other "utterance b starts here" table that
 * replaces the reference's padded [B, T] indexing (preprocess_speech.py:49 runs B = 1, so it never needs one). */
int ser_ragged_index(const int32_t* row_offs, const int64_t* base, int B, int64_t step, int64_t mult, int64_t div,
                     int32_t* out, int64_t total_rows, void* stream);

/* ---- command lists -----------------------------------------------------------------------------------------------
 * A forward over a packed ragged batch is ~150 launches whose POINTERS are fixed once the host has laid its buffers out
 * and whose only per-batch variables are a handful of row counts.  The host therefore records the launches once as an
 * array of ser_cmd, patches the row counts in place for every batch and replays the array with ONE call: the launching
 * thread leaves the interpreter once per batch instead of once per kernel (the reference pays one Python dispatch per
 * torch op, preprocess_speech.py:49).  Each ser_cmd carries the arguments of the launcher of the same name. */
typedef struct ser_attention_args {
    const void* qkv; int64_t ld; int64_t plane_stride; int32_t q_col, k_col, v_col, B;
    const int32_t* frame_offs; const float* table; const float* gate;
    int32_t max_frames, table_T;
    void* out; int64_t ldo; int64_t out_plane_stride;
    int32_t H, dh; float scale; int32_t mode; int32_t gate_col, out_mode;   /* out_mode: 0 = `mode`'s planes, SER_MODE_FP16M: see out_scale */
    const float* gru_const; const int32_t* key_lens;
    const float* bias2d; int64_t bias2d_ld;
    /* WavLM gate computed INSIDE the kernel (round 3; ser_attention_v only): the two pre-activations per (row, head) are linear in
     * LayerNorm1(x) restricted to the head's dh channels (HF modeling_wavlm.py:167-180), so instead of riding along as 2H extra output
     * columns of the packed projection (gate_col: a 13th 256-wide column tile for 32 columns at D = 1024) every query block multiplies
     * its rows' dh elements of the layer input's operand copy with the head's two folded weight rows on the matrix cores (one MFMA chain
     * shaped like S = K Q^T) and applies the deferred LayerNorm in closed form:
     *   pre_j = rstd * (sum_d x[d] * gate_w[h][j][d] - mean * gate_cb[h][j]) + gate_cb[h][2 + j],   j = 0, 1
     * gate_x = that copy (the A operand of the packed projection: element type and plane count of `mode`), gate_stat = the projection's
     * ser_gemm_args.lnstat_out, gate_w = gamma-folded summed weights as operand planes [planes][H][2][dh] in the same format (ser_split_bf16),
     * gate_cb = fp32 [H][4] (column sums of the fp32 fold | beta W^T + b).  Needs table and gru_const; gate and gate_col are ignored. */
    const void* gate_x; int64_t gate_x_ld; int64_t gate_x_plane_stride;
    const float* gate_stat; const void* gate_w; const float* gate_cb;
    int32_t gate_x_planes, reserved1;
    int64_t gate_w_plane_stride;
    /* ABI 14: context rows as SER_MODE_FP16M operands for an output projection in that format (out_mode = SER_MODE_FP16M; mode FP16X,
     * head dim 64 -- a head is one 64-column tile): plane 0 of `out` = the fp16 copy, plane 1 = [P | Q] e4m3 bytes, out_scale[D / 64][out_scale_ld]
     * = one word of four E8M0 codes per (head, row).  The reference has no counterpart (fp32 end to end, preprocess_speech.py:50). */
    uint32_t* out_scale; int64_t out_scale_ld;
} ser_attention_args;
/* ser_attention with its arguments in a struct (the form command lists carry); the only entry point that takes the gate_x fields. */
int ser_attention_v(const ser_attention_args* args, void* stream);

typedef struct ser_layernorm_args {
    const float* x; int64_t ldx; const float* g; const float* b; float eps; int32_t gelu;
    float* out_f32; int64_t ldo_f32; void* out_act; int64_t ldo_act; int64_t out_plane_stride;
    int32_t mode, rows, D, reserved0;
    uint32_t* range_flag;                           /* fp16 range guard, may be NULL (ABI 13) */
} ser_layernorm_args;
int ser_layernorm_v(const ser_layernorm_args* args, void* stream);

typedef struct ser_row_center_args {
    const float* x; int64_t ldx; void* out_act; int64_t ldo_act; int64_t out_plane_stride;
    float* stats; float* shift; int32_t stat_groups, mode, rows, D;
    uint32_t* out_scale; int64_t out_scale_ld;      /* mode == SER_MODE_FP16M: block scales of the copy (ABI 13) */
    uint32_t* range_flag;                           /* fp16 range guard, may be NULL (see ser_gemm_args.range_flag) */
} ser_row_center_args;
int ser_row_center_v(const ser_row_center_args* args, void* stream);

typedef struct ser_logmel_args {
    const float* wav; const int64_t* sample_offs; const float* mel; float* out; void* work; int32_t B, n_mels;
} ser_logmel_args;

typedef struct ser_pack_act_args {
    const float* x; void* out; int64_t ldo; int64_t out_plane_stride; int32_t B, C, T, halo, mode, reserved0;
    uint32_t* range_flag;                           /* fp16 range guard, may be NULL (ABI 13) */
} ser_pack_act_args;
int ser_pack_act_v(const ser_pack_act_args* args, void* stream);

typedef struct ser_wave_frames_args {
    const float* wav; const int64_t* sample_offs; const int32_t* frame_offs; int32_t B, k, stride, mode;
    void* out; int64_t out_plane_stride; void* work; int32_t total_rows, reserved0;
    uint32_t* range_flag;                           /* fp16 range guard, may be NULL (ABI 13) */
} ser_wave_frames_args;
int ser_wave_frames_v(const ser_wave_frames_args* args, void* stream);

#define SER_OP_GEMM 1
#define SER_OP_ATTENTION 2
#define SER_OP_LAYERNORM 3
#define SER_OP_WAVE_FRAMES 4
#define SER_OP_ROW_CENTER 5
#define SER_OP_LOGMEL 6
#define SER_OP_PACK_ACT 7
typedef struct ser_cmd {
    int32_t op, reserved0;
    union {
        ser_gemm_args        gemm;
        ser_attention_args   attention;
        ser_layernorm_args   layernorm;
        ser_wave_frames_args wave_frames;
        ser_row_center_args  row_center;
        ser_logmel_args      logmel;
        ser_pack_act_args    pack_act;
    } u;
} ser_cmd;

/* Enqueue cmds[0..n) in order on `stream`.  Stops at the first launcher that fails and returns its code
 * (ser_last_error() names it); *failed_at, if not NULL, receives the index. */
int ser_run(const ser_cmd* cmds, int32_t n, int32_t* failed_at, void* stream);

/* ---- host-side file I/O (HOST pointers, no stream; plain C, re-entrant, entered without the Python GIL) --------------
 * a5  ser_wav_read_f32: what librosa.load(path, sr=16000) -> soundfile yields for a RIFF/WAVE file
 *     (preprocessing/preprocess_speech.py:47) before any resampling: mono float32 in [-1, 1], integer PCM / 2^(bits-1),
 *     channels averaged.  Returns the number of frames, or < 0; with host_dst == NULL only the header is read.
 * a21 ser_pt_write_f32: the file torch.save(feats, "<name>.pt") leaves for the heads (preprocess_speech.py:69-71;
 *     consumer torch.load(path), bin/train_cat_bimodal_lazy_1head.py:227): a bare [rows, cols] float32 CPU tensor in
 *     torch's zip archive format. */
int64_t ser_wav_read_f32(const char* path, float* host_dst, int64_t capacity, int32_t* sample_rate, int32_t* channels);
int     ser_pt_write_f32(const char* path, const float* host_data, int64_t rows, int64_t cols);

#define SER_WS_LOGMEL 1
#define SER_WS_WAVE_FRAMES 2
size_t ser_workspace_bytes(int op, int B, int T, int D, int H, int mode);

#ifdef __cplusplus
}
#endif
#endif /* SER_HIP_H */
