// DeBERTa-v2/v3 text variant (next row 8f-1, preprocessing/preprocess_deroberta.py): the two pieces its encoder needs
// beyond the shared GEMM / LayerNorm kernels.  Sequences are 80 tokens in the reference (max_len), so these are
// small-problem kernels: clarity over throughput.
//
//   ser_embed_ln_masked   LayerNorm(word_embedding[id]) with padded rows zeroed
//                         (HF modeling_deberta_v2.py DebertaV2Embeddings: no absolute positions / token types in v3)
//   ser_deberta_attention disentangled attention (DisentangledSelfAttention.forward + disentangled_attention_bias):
//                         softmax((Qc Kc^T + c2p + p2c) / sqrt(3 dh)) V with the "both tokens real" mask
#include "ser_common.h"

// ---------------------------------------------------------------------------------------------- embeddings
template <int MODE>
__global__ __launch_bounds__(256) void embed_ln_masked_kernel(const int32_t* __restrict__ ids, const float* __restrict__ wemb,
                                                              const float* __restrict__ g, const float* __restrict__ b, float eps,
                                                              const int32_t* __restrict__ key_lens, float* __restrict__ of,
                                                              unsigned short* __restrict__ oa, int64_t plane, int T, int D,
                                                              int rows) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);                 // one wave per token
    if (row >= rows) return;
    const int seq = row / T, t = row - seq * T;
    const bool real = t < key_lens[seq];
    const float* w = wemb + (int64_t)ids[row] * D;
    f32x4 v[8];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int c = i * 256 + lane * 4;
        if (c < D) {
            v[i] = *(const f32x4*)(w + c);
            s += v[i][0] + v[i][1] + v[i][2] + v[i][3];
        } else v[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    const float mean = wave_sum(s) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int c = i * 256 + lane * 4;
        if (c < D) {
#pragma unroll
            for (int j = 0; j < 4; ++j) { const float d = v[i][j] - mean; q += d * d; }
        }
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)D + eps);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int c = i * 256 + lane * 4;
        if (c < D) {
            const f32x4 gg = *(const f32x4*)(g + c), bb = *(const f32x4*)(b + c);
            f32x4 y;
#pragma unroll
            for (int j = 0; j < 4; ++j) y[j] = real ? (v[i][j] - mean) * rstd * gg[j] + bb[j] : 0.f;   // embeddings * mask
            if (of) *(f32x4*)(of + (int64_t)row * D + c) = y;
            if (oa) store_act4<MODE>(oa + (int64_t)row * D + c, plane, y[0], y[1], y[2], y[3]);
        }
    }
}

extern "C" int ser_embed_ln_masked(const int32_t* ids, const float* word_emb, const float* ln_g, const float* ln_b, float eps,
                                   const int32_t* key_lens, float* out_f32, void* out_act, int64_t out_plane_stride,
                                   int mode, int B, int T, int D, void* stream) {
    if (!ids || !word_emb || !ln_g || !ln_b || !key_lens || (!out_f32 && !out_act))
        return ser_fail(-1, "ser_embed_ln_masked: null pointer");
    if (B <= 0 || T <= 0 || D % 4 || D > 2048) return ser_fail(-2, "ser_embed_ln_masked: bad B/T/D");
    if (mode != SER_MODE_BF16 && mode != SER_MODE_FP32X && mode != SER_MODE_FP16X) return ser_fail(-3, "ser_embed_ln_masked: bad mode");
    const int rows = B * T;
    dim3 grid((rows + 3) / 4), block(256);
    if (mode == SER_MODE_FP32X)
        hipLaunchKernelGGL(embed_ln_masked_kernel<SER_MODE_FP32X>, grid, block, 0, (hipStream_t)stream, ids, word_emb, ln_g, ln_b,
                           eps, key_lens, out_f32, (unsigned short*)out_act, out_plane_stride, T, D, rows);
    else if (mode == SER_MODE_FP16X)
        hipLaunchKernelGGL(embed_ln_masked_kernel<SER_MODE_FP16X>, grid, block, 0, (hipStream_t)stream, ids, word_emb, ln_g, ln_b,
                           eps, key_lens, out_f32, (unsigned short*)out_act, out_plane_stride, T, D, rows);
    else
        hipLaunchKernelGGL(embed_ln_masked_kernel<SER_MODE_BF16>, grid, block, 0, (hipStream_t)stream, ids, word_emb, ln_g, ln_b,
                           eps, key_lens, out_f32, (unsigned short*)out_act, out_plane_stride, T, D, rows);
    return ser_check_launch("ser_embed_ln_masked");
}

// ---------------------------------------------------------------------------------------------- attention
// One block per (sequence, head), one thread per query (T <= 128, dh <= 64).  K and V of the head sit in LDS as fp32,
// the score row of every query in LDS too; two passes (scores + max, then exp / sum / P V).
//   score(q,k) = (Qc_q . Kc_k + c2p[q][ci[q-k]] + p2c[k][pi[k-q]]) * scale        if q and k are real tokens
//              = lowest float                                                       otherwise
// so a padded query row is the uniform average of all T value rows, exactly as HF's masked_fill + softmax.
// c2p / p2c: [rows, H * Nr] fp32 = Qc / Kc times the (shared-projection) position keys / queries restricted to the
// Nr relative-position rows a T-token sequence can reach; ci / pi: [2T-1] column of that window per signed distance.
template <int MODE>
__global__ __launch_bounds__(128) void deberta_attention_kernel(
    const unsigned short* __restrict__ qkv, int64_t ld, int64_t plane, int q_col, int k_col, int v_col,
    const float* __restrict__ c2p, const float* __restrict__ p2c, int64_t ldp, int Nr,
    const int32_t* __restrict__ ci, const int32_t* __restrict__ pi, const int32_t* __restrict__ key_lens,
    unsigned short* __restrict__ out, int64_t ldo, int64_t out_plane, int T, int H, int dh, float scale) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* Ks = (float*)smem;                       // [T][64]
    float* Vs = Ks + T * 64;                        // [T][64]
    float* Ss = Vs + T * 64;                        // [T][T + 1]
    const int b = blockIdx.x / H, h = blockIdx.x - b * H;
    const int tid = threadIdx.x;
    const int64_t row0 = (int64_t)b * T;
    const int len = key_lens[b];
    auto act = [&](int64_t r, int col) -> float {   // one element of a bf16 (+ lo plane) operand
        const unsigned short* p = qkv + r * ld + col;
        float x = bf2f(p[0]);
        if (MODE == SER_MODE_FP32X) x += bf2f(p[plane]);
        return x;
    };
    for (int i = tid; i < T * 64; i += 128) {
        const int k = i >> 6, d = i & 63;
        Ks[i] = d < dh ? act(row0 + k, k_col + h * dh + d) : 0.f;
        Vs[i] = d < dh ? act(row0 + k, v_col + h * dh + d) : 0.f;
    }
    __syncthreads();
    const int q = tid;
    if (q >= T) return;
    // head dims below 64 are zero-padded in LDS / registers (dh % 4 == 0), so every inner loop is 16 aligned float4 steps
    f32x4 qv[16];
#pragma unroll
    for (int d4 = 0; d4 < 16; ++d4) {
        f32x4 t = {0.f, 0.f, 0.f, 0.f};
        if (4 * d4 < dh) {
            t[0] = act(row0 + q, q_col + h * dh + 4 * d4); t[1] = act(row0 + q, q_col + h * dh + 4 * d4 + 1);
            t[2] = act(row0 + q, q_col + h * dh + 4 * d4 + 2); t[3] = act(row0 + q, q_col + h * dh + 4 * d4 + 3);
        }
        qv[d4] = t;
    }
    const float* c2p_row = c2p + (row0 + q) * ldp + (int64_t)h * Nr;
    float* srow = Ss + q * (T + 1);
    const float lowest = -3.402823466e+38f;
    const bool qreal = q < len;
    float m = lowest;
    for (int k = 0; k < T; ++k) {
        float s = lowest;
        if (qreal && k < len) {
            const f32x4* kr = (const f32x4*)(Ks + k * 64);
            f32x4 acc4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int d4 = 0; d4 < 16; ++d4) acc4 = __builtin_elementwise_fma(qv[d4], kr[d4], acc4);
            const float dot = (acc4[0] + acc4[1]) + (acc4[2] + acc4[3]);
            const float bias = c2p_row[ci[q - k + T - 1]] + p2c[(row0 + k) * ldp + (int64_t)h * Nr + pi[k - q + T - 1]];
            s = (dot + bias) * scale;
        }
        srow[k] = s;
        m = fmaxf(m, s);
    }
    float l = 0.f;
    f32x4 o[16];
#pragma unroll
    for (int d4 = 0; d4 < 16; ++d4) o[d4] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < T; ++k) {
        const float e = __expf(srow[k] - m);                            // all-lowest row: exp(0) = 1 for every key -> uniform
        l += e;
        const f32x4* vr = (const f32x4*)(Vs + k * 64);
        const f32x4 e4 = {e, e, e, e};
#pragma unroll
        for (int d4 = 0; d4 < 16; ++d4) o[d4] = __builtin_elementwise_fma(e4, vr[d4], o[d4]);
    }
    const float inv = 1.0f / l;
    unsigned short* orow = out + (row0 + q) * ldo + h * dh;
#pragma unroll
    for (int d4 = 0; d4 < 16; ++d4)
        if (4 * d4 < dh)
            store_act4<MODE>(orow + 4 * d4, out_plane, o[d4][0] * inv, o[d4][1] * inv, o[d4][2] * inv, o[d4][3] * inv);
}

// --------------------------------------------------------------------- ConvLayer support (deberta-v2 xlarge / xxlarge)
// DebertaV2Encoder runs a token-axis Conv1d(D, D, 3) over the EMBEDDING output after encoder layer 0 (HF
// modeling_deberta_v2.py ConvLayer).  The conv itself is ser_gemm's implicit-conv map over a zero-halo'd copy of the rows
// (one zero row before and after every sequence), GELU and the residual ride in its epilogue, ser_layernorm follows; these two
// row kernels provide the halo'd operand copy and the final "padded rows are zero" of that layer.
template <int MODE>
__global__ __launch_bounds__(256) void pack_rows_kernel(const float* __restrict__ x, int64_t ldx, int T, int D, int halo,
                                                        unsigned short* __restrict__ o, int64_t ldo, int64_t plane, int64_t total) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;                    // one thread per 4 columns
    if (i >= total) return;
    const int c4 = (int)(i % (D / 4)) * 4;
    const int64_t row = i / (D / 4);
    const int64_t b = row / T;
    const int t = (int)(row - b * T);
    const f32x4 v = *(const f32x4*)(x + row * ldx + c4);
    store_act4<MODE>(o + (b * (T + 2 * halo) + halo + t) * ldo + c4, plane, v[0], v[1], v[2], v[3]);
}

extern "C" int ser_pack_rows(const float* x, int64_t ldx, int B, int T, int D, int halo, void* out, int64_t ldo,
                             int64_t out_plane_stride, int mode, void* stream) {
    if (!x || !out || B <= 0 || T <= 0 || D <= 0 || (D % 4) || halo < 0 || (ldx % 4) || (ldo % 4))
        return ser_fail(-1, "ser_pack_rows: bad arguments");
    const int64_t total = (int64_t)B * T * (D / 4);
    dim3 grid((unsigned)((total + 255) / 256)), block(256);
    if (mode == SER_MODE_FP32X)
        hipLaunchKernelGGL(pack_rows_kernel<SER_MODE_FP32X>, grid, block, 0, (hipStream_t)stream, x, ldx, T, D, halo,
                           (unsigned short*)out, ldo, out_plane_stride, total);
    else if (mode == SER_MODE_BF16)
        hipLaunchKernelGGL(pack_rows_kernel<SER_MODE_BF16>, grid, block, 0, (hipStream_t)stream, x, ldx, T, D, halo,
                           (unsigned short*)out, ldo, out_plane_stride, total);
    else if (mode == SER_MODE_FP16X)
        hipLaunchKernelGGL(pack_rows_kernel<SER_MODE_FP16X>, grid, block, 0, (hipStream_t)stream, x, ldx, T, D, halo,
                           (unsigned short*)out, ldo, out_plane_stride, total);
    else return ser_fail(-2, "ser_pack_rows: bad mode %d", mode);
    return ser_check_launch("ser_pack_rows");
}

template <int MODE>
__global__ __launch_bounds__(256) void zero_padded_rows_kernel(float* __restrict__ x, int64_t ldx, unsigned short* __restrict__ a,
                                                               int64_t lda, int64_t plane, const int32_t* __restrict__ key_lens,
                                                               int T, int D, int64_t total) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int c4 = (int)(i % (D / 4)) * 4;
    const int64_t row = i / (D / 4);
    const int64_t b = row / T;
    if ((int)(row - b * T) < key_lens[b]) return;
    if (x) *(f32x4*)(x + row * ldx + c4) = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (a) store_act4<MODE>(a + row * lda + c4, plane, 0.f, 0.f, 0.f, 0.f);
}

extern "C" int ser_zero_padded_rows(float* x, int64_t ldx, void* act, int64_t lda, int64_t plane_stride, int mode,
                                    const int32_t* key_lens, int B, int T, int D, void* stream) {
    if ((!x && !act) || !key_lens || B <= 0 || T <= 0 || D <= 0 || (D % 4) || (ldx % 4) || (lda % 4))
        return ser_fail(-1, "ser_zero_padded_rows: bad arguments");
    const int64_t total = (int64_t)B * T * (D / 4);
    dim3 grid((unsigned)((total + 255) / 256)), block(256);
    if (mode == SER_MODE_FP32X)
        hipLaunchKernelGGL(zero_padded_rows_kernel<SER_MODE_FP32X>, grid, block, 0, (hipStream_t)stream, x, ldx,
                           (unsigned short*)act, lda, plane_stride, key_lens, T, D, total);
    else if (mode == SER_MODE_BF16)
        hipLaunchKernelGGL(zero_padded_rows_kernel<SER_MODE_BF16>, grid, block, 0, (hipStream_t)stream, x, ldx,
                           (unsigned short*)act, lda, plane_stride, key_lens, T, D, total);
    else if (mode == SER_MODE_FP16X)
        hipLaunchKernelGGL(zero_padded_rows_kernel<SER_MODE_FP16X>, grid, block, 0, (hipStream_t)stream, x, ldx,
                           (unsigned short*)act, lda, plane_stride, key_lens, T, D, total);
    else return ser_fail(-2, "ser_zero_padded_rows: bad mode %d", mode);
    return ser_check_launch("ser_zero_padded_rows");
}

// ------------------------------------------------------------------------------------- dense bias
// bias[b][h][q][k] = c2p[q][ci[q-k]] + p2c_scale * p2c[k][pi[k-q]]  for real tokens q, k < key_lens[b], else 0
// (the attention kernel masks the keys a real query may not see and gives a padded query equal scores everywhere).
// c2p comes from the pre-scaled q, so only the position -> content term needs the score scale.  Thread = 4 consecutive keys.
__global__ __launch_bounds__(256) void deberta_bias_kernel(const float* __restrict__ c2p, const float* __restrict__ p2c,
                                                           int64_t ldp, int Nr, const int32_t* __restrict__ ci,
                                                           const int32_t* __restrict__ pi, const int32_t* __restrict__ key_lens,
                                                           float* __restrict__ out, int64_t ld, int T, int H, float p2c_scale,
                                                           int64_t total) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int k4 = (int)(i % (ld / 4)) * 4;
    const int64_t r = i / (ld / 4);
    const int q = (int)(r % T);
    const int64_t bh = r / T;
    const int h = (int)(bh % H), b = (int)(bh / H);
    const int len = key_lens[b];
    const int64_t row0 = (int64_t)b * T;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (q < len) {
        const float* c2p_row = c2p + (row0 + q) * ldp + (int64_t)h * Nr;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int k = k4 + j;
            if (k < len)
                v[j] = c2p_row[ci[q - k + T - 1]] + p2c_scale * p2c[(row0 + k) * ldp + (int64_t)h * Nr + pi[k - q + T - 1]];
        }
    }
    *(f32x4*)(out + r * ld + k4) = v;
}

extern "C" int ser_deberta_bias(const float* c2p, const float* p2c, int64_t ldp, int Nr, const int32_t* c2p_col,
                                const int32_t* p2c_col, const int32_t* key_lens, float* out, int64_t ld, int B, int T, int H,
                                float p2c_scale, void* stream) {
    if (!c2p || !p2c || !c2p_col || !p2c_col || !key_lens || !out) return ser_fail(-1, "ser_deberta_bias: null pointer");
    if (B <= 0 || H <= 0 || T <= 0 || Nr <= 0 || ldp < (int64_t)H * Nr) return ser_fail(-2, "ser_deberta_bias: bad shape");
    if (ld < T || (ld % 4)) return ser_fail(-3, "ser_deberta_bias: ld=%lld must be a multiple of 4 and >= T", (long long)ld);
    const int64_t total = (int64_t)B * H * T * (ld / 4);
    hipLaunchKernelGGL(deberta_bias_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, c2p, p2c, ldp,
                       Nr, c2p_col, p2c_col, key_lens, out, ld, T, H, p2c_scale, total);
    return ser_check_launch("ser_deberta_bias");
}

extern "C" int ser_deberta_attention(const void* qkv, int64_t ld, int64_t plane_stride, int q_col, int k_col, int v_col,
                                     const float* c2p, const float* p2c, int64_t ldp, int Nr, const int32_t* c2p_col,
                                     const int32_t* p2c_col, const int32_t* key_lens, void* out, int64_t ldo,
                                     int64_t out_plane_stride, int B, int T, int H, int dh, int mode, void* stream) {
    if (!qkv || !c2p || !p2c || !c2p_col || !p2c_col || !key_lens || !out) return ser_fail(-1, "ser_deberta_attention: null pointer");
    if (B <= 0 || H <= 0 || T <= 0 || T > 128) return ser_fail(-2, "ser_deberta_attention: T=%d must be in 1..128", T);
    if (dh % 4 || dh < 4 || dh > 64) return ser_fail(-3, "ser_deberta_attention: head dim %d unsupported (multiple of 4, <= 64)", dh);
    if (mode != SER_MODE_BF16 && mode != SER_MODE_FP32X) return ser_fail(-4, "ser_deberta_attention: bad mode %d", mode);
    if (Nr <= 0 || ldp < (int64_t)H * Nr || (ldo % 4)) return ser_fail(-5, "ser_deberta_attention: bad position-table pitch");
    const size_t lds = (size_t)(2 * T * 64 + T * (T + 1)) * 4;
    const float scale = 1.0f / sqrtf(3.0f * (float)dh);
    auto k = mode == SER_MODE_FP32X ? deberta_attention_kernel<SER_MODE_FP32X> : deberta_attention_kernel<SER_MODE_BF16>;
    if (lds > 65536) {
        hipError_t e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return ser_fail((int)e, "ser_deberta_attention: cannot raise dynamic LDS");
    }
    hipLaunchKernelGGL(k, dim3((unsigned)(B * H)), dim3(128), lds, (hipStream_t)stream, (const unsigned short*)qkv, ld, plane_stride,
                       q_col, k_col, v_col, c2p, p2c, ldp, Nr, c2p_col, p2c_col, key_lens, (unsigned short*)out, ldo,
                       out_plane_stride, T, H, dh, scale);
    return ser_check_launch("ser_deberta_attention");
}
