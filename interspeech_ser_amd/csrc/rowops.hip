// HBM-bound row kernels: waveform normalisation (K1), conv layer 0 + LayerNorm + GELU (K2),
// LayerNorm (K6), WavLM bias table / GRU gate (K8), state mean (K15), bf16 splitting.
// One 64-lane wave owns one row; reductions are DPP/shuffle butterflies, loads/stores 16 B/lane.
#include "ser_common.h"

// ------------------------------------------------------------------------------- K1
// One 1024-thread block per utterance, three sweeps (sum, squared deviation, write).
__global__ __launch_bounds__(1024) void wave_norm_kernel(const float* __restrict__ wav,
                                                         const int64_t* __restrict__ offs, float* __restrict__ out) {
    __shared__ double red[16];
    __shared__ float stat[2];
    const int b = blockIdx.x;
    const int64_t s0 = offs[b], n = offs[b + 1] - s0;
    const float* x = wav + s0;
    float* y = out + s0;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;

    double acc = 0.0;
    for (int64_t i = tid; i < n; i += 1024) acc += (double)x[i];
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if (lane == 0) red[wv] = acc;
    __syncthreads();
    if (tid == 0) {
        double t = 0.0;
        for (int i = 0; i < 16; ++i) t += red[i];
        stat[0] = (float)(t / (double)n);
    }
    __syncthreads();
    const float mean = stat[0];
    acc = 0.0;
    for (int64_t i = tid; i < n; i += 1024) { const float d = x[i] - mean; acc += (double)(d * d); }
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    __syncthreads();
    if (lane == 0) red[wv] = acc;
    __syncthreads();
    if (tid == 0) {
        double t = 0.0;
        for (int i = 0; i < 16; ++i) t += red[i];
        stat[1] = sqrtf((float)(t / (double)n) + 1e-7f);
    }
    __syncthreads();
    const float sd = stat[1];
    for (int64_t i = tid; i < n; i += 1024) y[i] = (x[i] - mean) / sd;
}

extern "C" int ser_wave_norm(const float* wav, const int64_t* sample_offs, int B, float* out, void* stream) {
    if (!wav || !sample_offs || !out || B <= 0) return ser_fail(-1, "ser_wave_norm: bad arguments");
    hipLaunchKernelGGL(wave_norm_kernel, dim3(B), dim3(1024), 0, (hipStream_t)stream, wav, sample_offs, out);
    return ser_check_launch("ser_wave_norm");
}

// ------------------------------------------------------------------------- K1 + im2col
// Waveform front door for the matrix-core conv0: per-utterance zero-mean/unit-variance (K1) fused
// with the framing of conv layer 0 -- every frame t becomes one 128-byte act row
// [x_n[s*t .. s*t+k-1], 0 ...] (K padded to 64), so Conv1d(1,C,k,s)+LayerNorm+GELU runs as the
// LN-epilogue GEMM and the normalised waveform never exists in HBM as fp32.
__global__ __launch_bounds__(256) void wave_stats_kernel(const float* __restrict__ wav, const int64_t* __restrict__ offs,
                                                         double* __restrict__ part /*[B][64][2]*/) {
    __shared__ double red[2][4];
    const int b = blockIdx.y, chunk = blockIdx.x;
    const int64_t s0 = offs[b], n = offs[b + 1] - s0;
    const int64_t per = (n + 63) / 64;
    const int64_t lo = chunk * per, hi = (lo + per < n) ? lo + per : n;
    double a1 = 0.0, a2 = 0.0;
    for (int64_t i = lo + threadIdx.x; i < hi; i += 256) { const double x = (double)wav[s0 + i]; a1 += x; a2 += x * x; }
    for (int o = 32; o > 0; o >>= 1) { a1 += __shfl_xor(a1, o, 64); a2 += __shfl_xor(a2, o, 64); }
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = a1; red[1][threadIdx.x >> 6] = a2; }
    __syncthreads();
    if (threadIdx.x == 0) {
        part[((int64_t)b * 64 + chunk) * 2] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
        part[((int64_t)b * 64 + chunk) * 2 + 1] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
    }
}

template <int MODE>
__global__ __launch_bounds__(256) void wave_frames_kernel(const float* __restrict__ wav, const int64_t* __restrict__ soffs,
                                                          const int32_t* __restrict__ foffs, const double* __restrict__ part,
                                                          unsigned short* __restrict__ out, int64_t plane, int k, int stride, uint32_t* __restrict__ rflag) {
    __shared__ float stat[2];
    float ramax = 0.f;
    const int b = blockIdx.y, tid = threadIdx.x;
    const int64_t s0 = soffs[b], n = soffs[b + 1] - s0;
    if (tid < 64) {
        double a1 = part[((int64_t)b * 64 + tid) * 2], a2 = part[((int64_t)b * 64 + tid) * 2 + 1];
        for (int o = 32; o > 0; o >>= 1) { a1 += __shfl_xor(a1, o, 64); a2 += __shfl_xor(a2, o, 64); }
        if (tid == 0) {
            const double mean = a1 / (double)n;
            const double var = a2 / (double)n - mean * mean;
            stat[0] = (float)mean;
            stat[1] = 1.0f / sqrtf((float)(var > 0.0 ? var : 0.0) + 1e-7f);
        }
    }
    __syncthreads();
    const float mean = stat[0], rstd = stat[1];
    const int row_begin = foffs[b], T = foffs[b + 1] - row_begin;
    const float* x = wav + s0;
    const int c = tid & 7;                                        // 16-byte chunk of the 128-byte row
    for (int t = blockIdx.x * 32 + (tid >> 3); t < T; t += gridDim.x * 32) {
        float v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int j = c * 8 + i;
            v[i] = (j < k) ? (x[(int64_t)t * stride + j] - mean) * rstd : 0.f;
        }
        unsigned short* dst = out + (int64_t)(row_begin + t) * 64 + c * 8;
        store_act4<MODE>(dst, plane, v[0], v[1], v[2], v[3]);
        store_act4<MODE>(dst + 4, plane, v[4], v[5], v[6], v[7]);
        if constexpr (mode_traits<MODE>::f16) {
#pragma unroll
            for (int i = 0; i < 8; ++i) ramax = fmaxf(ramax, fabsf(v[i]));
        }
    }
    if constexpr (mode_traits<MODE>::f16) range_report(rflag, ramax);
}

extern "C" int ser_wave_frames_v(const ser_wave_frames_args* a, void* stream) {
    if (!a) return ser_fail(-1, "ser_wave_frames: null pointer");
    const float* wav = a->wav; const int64_t* sample_offs = a->sample_offs; const int32_t* frame_offs = a->frame_offs;
    const int B = a->B, k = a->k, stride = a->stride, mode = a->mode, total_rows = a->total_rows;
    void* out = a->out; const int64_t out_plane_stride = a->out_plane_stride; void* work = a->work;
    if (!wav || !sample_offs || !frame_offs || !out || !work) return ser_fail(-1, "ser_wave_frames: null pointer");
    if (B <= 0 || k < 1 || k > 64 || stride < 1 || total_rows <= 0) return ser_fail(-2, "ser_wave_frames: bad B/k/stride/rows");
    if (mode != SER_MODE_BF16 && mode != SER_MODE_FP32X && mode != SER_MODE_FP16X) return ser_fail(-3, "ser_wave_frames: bad mode");
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(wave_stats_kernel, dim3(64, B), dim3(256), 0, s, wav, sample_offs, (double*)work);
    int blocks = ((total_rows + B - 1) / B + 31) / 32;
    if (blocks > 1024) blocks = 1024;
    if (blocks < 1) blocks = 1;
#define SER_WF(M_) hipLaunchKernelGGL(wave_frames_kernel<M_>, dim3(blocks, B), dim3(256), 0, s, wav, sample_offs, frame_offs, \
                                      (const double*)work, (unsigned short*)out, out_plane_stride, k, stride, a->range_flag)
    if (mode == SER_MODE_FP32X) SER_WF(SER_MODE_FP32X);
    else if (mode == SER_MODE_FP16X) SER_WF(SER_MODE_FP16X);
    else SER_WF(SER_MODE_BF16);
#undef SER_WF
    return ser_check_launch("ser_wave_frames");
}

extern "C" int ser_wave_frames(const float* wav, const int64_t* sample_offs, const int32_t* frame_offs, int B, int k,
                               int stride, void* out, int64_t out_plane_stride, int mode, void* work, int total_rows,
                               void* stream) {
    ser_wave_frames_args a = {};
    a.wav = wav; a.sample_offs = sample_offs; a.frame_offs = frame_offs; a.B = B; a.k = k; a.stride = stride; a.mode = mode;
    a.out = out; a.out_plane_stride = out_plane_stride; a.work = work; a.total_rows = total_rows;
    return ser_wave_frames_v(&a, stream);
}

// ------------------------------------------------------------------------------- K2
// Wave per GROUP of frames: consecutive frames overlap in the waveform (stride < kernel), so ONE
// coalesced 64-sample load serves `rpg` frames (8 for k=10, stride=5); taps are broadcast with
// v_readlane (wave-uniform index -> scalar operand of the FMA), the next group's samples are
// fetched before the current group is computed.  Lane owns CPL consecutive channels, weights in
// registers.  KT = compile-time bound on the kernel width.
template <int CPL, int KT, int MODE>
__global__ __launch_bounds__(256) void conv0_kernel(const float* __restrict__ x, const int64_t* __restrict__ soffs,
                                                    const int32_t* __restrict__ foffs,
                                                    const float* __restrict__ w, const float* __restrict__ bias,
                                                    const float* __restrict__ lg, const float* __restrict__ lb,
                                                    unsigned short* __restrict__ out, int64_t plane,
                                                    int k, int stride, int rpg) {
    constexpr int C = CPL * 64;
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.y;                                    // one grid row per utterance
    const int row_begin = foffs[b], T = foffs[b + 1] - row_begin;
    const int64_t s0 = soffs[b], nsamp = soffs[b + 1] - s0;
    const int wglobal = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int nwaves = gridDim.x * 4;
    const int ngroups = (T + rpg - 1) / rpg;
    const int c0 = lane * CPL;
    float wr[CPL][KT];
    float br[CPL], gr[CPL], be[CPL];
#pragma unroll
    for (int i = 0; i < CPL; ++i) {
#pragma unroll
        for (int j = 0; j < KT; ++j) wr[i][j] = j < k ? w[(c0 + i) * k + j] : 0.f;
        br[i] = bias ? bias[c0 + i] : 0.f;
        gr[i] = lg[c0 + i];
        be[i] = lb[c0 + i];
    }
    const float* xb = x + s0;
    auto fetch = [&](int g) -> float {
        const int64_t i = (int64_t)g * rpg * stride + lane;
        return (g < ngroups && i < nsamp) ? xb[i] : 0.f;
    };
    float xv = fetch(wglobal);
    for (int g = wglobal; g < ngroups; g += nwaves) {
        const float xn = fetch(g + nwaves);                       // in flight while this group computes
        const int t0 = g * rpg;
        const int nrow = (T - t0) < rpg ? (T - t0) : rpg;
        for (int j = 0; j < nrow; ++j) {
            float v[CPL];
#pragma unroll
            for (int i = 0; i < CPL; ++i) v[i] = br[i];
#pragma unroll
            for (int tap = 0; tap < KT; ++tap) {
                if (tap < k) {
                    const float xj = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(xv), j * stride + tap));
#pragma unroll
                    for (int i = 0; i < CPL; ++i) v[i] = fmaf(wr[i][tap], xj, v[i]);
                }
            }
            float s = 0.f;
#pragma unroll
            for (int i = 0; i < CPL; ++i) s += v[i];
            const float mean = wave_sum(s) * (1.0f / C);
            float q = 0.f;
#pragma unroll
            for (int i = 0; i < CPL; ++i) { const float d = v[i] - mean; q += d * d; }
            const float rstd = rsqrtf(wave_sum(q) * (1.0f / C) + 1e-5f);
#pragma unroll
            for (int i = 0; i < CPL; ++i) v[i] = gelu_erf((v[i] - mean) * rstd * gr[i] + be[i]);
            unsigned short* dst = out + (int64_t)(row_begin + t0 + j) * C + c0;
            if (CPL >= 4) {
#pragma unroll
                for (int i = 0; i < CPL; i += 4) store_act4<MODE>(dst + i, plane, v[i], v[i + 1], v[i + 2], v[i + 3]);
            } else {
#pragma unroll
                for (int i = 0; i < CPL; ++i) {
                    unsigned short h, l;
                    split_bf(v[i], h, l);
                    dst[i] = h;
                    if (MODE == SER_MODE_FP32X) dst[plane + i] = l;
                }
            }
        }
        xv = xn;
    }
}

extern "C" int ser_conv0_ln_gelu(const float* wav_norm, const int64_t* sample_offs, const int32_t* frame_offs,
                                 int B, const float* w, const float* bias, const float* ln_g, const float* ln_b,
                                 void* out, int64_t out_plane_stride, int mode, int C, int k, int stride,
                                 int total_rows, void* stream) {
    if (!wav_norm || !sample_offs || !frame_offs || !w || !ln_g || !ln_b || !out) return ser_fail(-1, "ser_conv0: null pointer");
    if (k < 1 || k > 16 || stride < 1 || total_rows <= 0 || B <= 0) return ser_fail(-2, "ser_conv0: bad k/stride/rows");
    if (mode != SER_MODE_BF16 && mode != SER_MODE_FP32X) return ser_fail(-3, "ser_conv0: bad mode");
    int rpg = (64 - k) / stride + 1;                             // frames served by one 64-sample load
    if (rpg > 8) rpg = 8;
    if (rpg < 1) rpg = 1;
    const int groups_per_utt = ((total_rows + B - 1) / B + rpg - 1) / rpg;
    int blocks = (groups_per_utt + 3) / 4;
    const int cap = (256 * 8 + B - 1) / B;
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    dim3 grid(blocks, B), block(256);
    hipStream_t s = (hipStream_t)stream;
    unsigned short* o = (unsigned short*)out;
#define LAUNCH2(CPL, KT)                                                                                        \
    do {                                                                                                        \
        if (mode == SER_MODE_BF16)                                                                              \
            hipLaunchKernelGGL((conv0_kernel<CPL, KT, SER_MODE_BF16>), grid, block, 0, s, wav_norm, sample_offs, \
                               frame_offs, w, bias, ln_g, ln_b, o, out_plane_stride, k, stride, rpg);          \
        else                                                                                                    \
            hipLaunchKernelGGL((conv0_kernel<CPL, KT, SER_MODE_FP32X>), grid, block, 0, s, wav_norm, sample_offs, \
                               frame_offs, w, bias, ln_g, ln_b, o, out_plane_stride, k, stride, rpg);          \
    } while (0)
#define LAUNCH(CPL)                                  \
    do {                                             \
        if (k <= 10) LAUNCH2(CPL, 10);               \
        else LAUNCH2(CPL, 16);                       \
    } while (0)
    switch (C) {
        case 64: LAUNCH(1); break;
        case 128: LAUNCH(2); break;
        case 256: LAUNCH(4); break;
        case 512: LAUNCH(8); break;
        default: return ser_fail(-4, "ser_conv0: C=%d unsupported (64/128/256/512)", C);
    }
#undef LAUNCH
#undef LAUNCH2
    return ser_check_launch("ser_conv0_ln_gelu");
}

// ------------------------------------------------------------------------------- K6
// Wave per row, row kept in registers (D <= 2048), two-pass mean / variance.
template <int MODE>
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, int64_t ldx,
                                                        const float* __restrict__ g, const float* __restrict__ b,
                                                        float eps, int gelu, float* __restrict__ of, int64_t ldof,
                                                        unsigned short* __restrict__ oa, int64_t ldoa, int64_t plane,
                                                        int rows, int D, uint32_t* __restrict__ rflag) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* xr = x + (int64_t)row * ldx;
    f32x4 v[8];
    float s = 0.f, ramax = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int c = i * 256 + lane * 4;
        if (c < D) { v[i] = *(const f32x4*)(xr + c); s += v[i][0] + v[i][1] + v[i][2] + v[i][3]; }
        else v[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
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
            for (int j = 0; j < 4; ++j) {
                float t = (v[i][j] - mean) * rstd * gg[j] + bb[j];
                y[j] = gelu ? gelu_erf(t) : t;
            }
            if (of) *(f32x4*)(of + (int64_t)row * ldof + c) = y;
            if (oa) store_act4<MODE>(oa + (int64_t)row * ldoa + c, plane, y[0], y[1], y[2], y[3]);
            if constexpr (mode_traits<MODE>::f16) { if (oa) ramax = fmaxf(ramax, fmaxf(fmaxf(fabsf(y[0]), fabsf(y[1])), fmaxf(fabsf(y[2]), fabsf(y[3])))); }
        }
    }
    if constexpr (mode_traits<MODE>::f16) range_report(rflag, ramax);
}

extern "C" int ser_layernorm_v(const ser_layernorm_args* a, void* stream) {
    if (!a) return ser_fail(-1, "ser_layernorm: null pointer");
    const float* x = a->x; const int64_t ldx = a->ldx, ldo_f32 = a->ldo_f32, ldo_act = a->ldo_act, out_plane_stride = a->out_plane_stride;
    const float* g = a->g; const float* b = a->b; const float eps = a->eps; const int gelu = a->gelu, mode = a->mode, rows = a->rows, D = a->D;
    float* out_f32 = a->out_f32; void* out_act = a->out_act;
    if (!x || !g || !b || (!out_f32 && !out_act)) return ser_fail(-1, "ser_layernorm: null pointer");
    if (D % 4 || D > 2048 || D <= 0 || rows <= 0) return ser_fail(-2, "ser_layernorm: D=%d rows=%d unsupported", D, rows);
    if ((ldx % 4) || (out_f32 && ldo_f32 % 4) || (out_act && ldo_act % 4)) return ser_fail(-3, "ser_layernorm: pitches must be multiples of 4");
    dim3 grid((rows + 3) / 4), block(256);
#define SER_LN(M_) hipLaunchKernelGGL(layernorm_kernel<M_>, grid, block, 0, (hipStream_t)stream, x, ldx, g, b, eps, gelu, out_f32, ldo_f32, \
                                      (unsigned short*)out_act, ldo_act, out_plane_stride, rows, D, a->range_flag)
    if (mode == SER_MODE_FP32X) SER_LN(SER_MODE_FP32X);
    else if (mode == SER_MODE_BF16) SER_LN(SER_MODE_BF16);
    else if (mode == SER_MODE_FP16) SER_LN(SER_MODE_FP16);
    else if (mode == SER_MODE_FP16X) SER_LN(SER_MODE_FP16X);
    else return ser_fail(-4, "ser_layernorm: bad mode %d", mode);
#undef SER_LN
    return ser_check_launch("ser_layernorm");
}

extern "C" int ser_layernorm(const float* x, int64_t ldx, const float* g, const float* b, float eps, int gelu,
                             float* out_f32, int64_t ldo_f32, void* out_act, int64_t ldo_act,
                             int64_t out_plane_stride, int mode, int rows, int D, void* stream) {
    ser_layernorm_args a = {};
    a.x = x; a.ldx = ldx; a.g = g; a.b = b; a.eps = eps; a.gelu = gelu; a.out_f32 = out_f32; a.ldo_f32 = ldo_f32; a.out_act = out_act;
    a.ldo_act = ldo_act; a.out_plane_stride = out_plane_stride; a.mode = mode; a.rows = rows; a.D = D;
    return ser_layernorm_v(&a, stream);
}

// ----------------------------------------------------------------- centred operand copy
// Wave per row: act copy of (x - mean_row), its row partials and the shift, in the layout the deferred-LayerNorm GEMMs
// consume (ser_gemm_args.ln_stats_in / shift_in).  Run ONCE per forward on hidden_states[0], whose row mean is produced
// by the launch that writes it (positional conv / Whisper stem) and therefore cannot be known to that launch's tiles;
// from then on every producer GEMM carries the shift forward from its residual input (ser_hip.h, "SHIFTED operand copy").
template <int MODE>
__global__ __launch_bounds__(256) void row_center_kernel(const float* __restrict__ x, int64_t ldx,
                                                         unsigned short* __restrict__ oa, int64_t ldoa, int64_t plane,
                                                         float* __restrict__ stats, int stat_groups,
                                                         float* __restrict__ shift, int rows, int D,
                                                         uint32_t* __restrict__ oscale, int64_t oscale_ld, uint32_t* __restrict__ rflag) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    float ramax = 0.f;
    const float* xr = x + (int64_t)row * ldx;
    f32x4 v[8];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int c = i * 256 + lane * 4;
        if (c < D) { v[i] = *(const f32x4*)(xr + c); s += v[i][0] + v[i][1] + v[i][2] + v[i][3]; }
        else v[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    const float mean = wave_sum(s) / (float)D;
    float q = 0.f, r1 = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int c = i * 256 + lane * 4;
        if (c < D) {
            f32x4 d;
#pragma unroll
            for (int j = 0; j < 4; ++j) { d[j] = v[i][j] - mean; q += d[j] * d[j]; r1 += d[j]; }
            if constexpr (MODE != SER_MODE_FP16M) store_act4<MODE>(oa + (int64_t)row * ldoa + c, plane, d[0], d[1], d[2], d[3]);
            if constexpr (mode_traits<MODE>::f16 && MODE != SER_MODE_FP16M) ramax = fmaxf(ramax, fmaxf(fmaxf(fabsf(d[0]), fabsf(d[1])), fmaxf(fabsf(d[2]), fabsf(d[3]))));
        }
        if constexpr (MODE == SER_MODE_FP16M) {                       // (D % 64 == 0: whole 32-column blocks; every lane joins the block shuffles)
            if (i * 256 < D) {
                const bool in = c < D;
                const float d4[4] = {in ? v[i][0] - mean : 0.f, in ? v[i][1] - mean : 0.f, in ? v[i][2] - mean : 0.f, in ? v[i][3] - mean : 0.f};
                ramax = fmaxf(ramax, mx_store_row<4>(d4, in, oa + (int64_t)row * ldoa, (unsigned char*)(oa + plane + (int64_t)row * ldoa), c,
                                                     oscale + row, oscale_ld, false));
            }
        }
    }
    if constexpr (mode_traits<MODE>::f16) range_report(rflag, ramax);
    q = wave_sum(q);
    r1 = wave_sum(r1);
    float* st = stats + (int64_t)row * stat_groups * 2;
    for (int g2 = lane; g2 < stat_groups * 2; g2 += 64) st[g2] = g2 == 0 ? r1 : (g2 == 1 ? q : 0.f);
    if (lane == 0) shift[row] = mean;
}

extern "C" int ser_row_center_v(const ser_row_center_args* a, void* stream) {
    if (!a) return ser_fail(-1, "ser_row_center: null pointer");
    const float* x = a->x; const int64_t ldx = a->ldx, ldo_act = a->ldo_act, out_plane_stride = a->out_plane_stride;
    void* out_act = a->out_act; float* stats = a->stats; float* shift = a->shift;
    const int stat_groups = a->stat_groups, mode = a->mode, rows = a->rows, D = a->D;
    if (!x || !out_act || !stats || !shift) return ser_fail(-1, "ser_row_center: null pointer");
    if (D % 4 || D > 2048 || D <= 0 || rows <= 0) return ser_fail(-2, "ser_row_center: D=%d rows=%d unsupported", D, rows);
    if ((ldx % 4) || (ldo_act % 4) || stat_groups < 2 || (stat_groups & 1))
        return ser_fail(-3, "ser_row_center: pitches must be multiples of 4, stat_groups even and >= 2");
    if (mode == SER_MODE_FP16M && (!a->out_scale || a->out_scale_ld < rows || (D % 64) || (ldo_act % 64)))
        return ser_fail(-5, "ser_row_center: FP16M needs out_scale (out_scale_ld >= rows), D %% 64 == 0, ldo_act %% 64 == 0");
    dim3 grid((rows + 3) / 4), block(256);
#define SER_RC(M_) hipLaunchKernelGGL(row_center_kernel<M_>, grid, block, 0, (hipStream_t)stream, x, ldx, (unsigned short*)out_act, ldo_act, \
                                      out_plane_stride, stats, stat_groups, shift, rows, D, a->out_scale, a->out_scale_ld, a->range_flag)
    if (mode == SER_MODE_FP32X) SER_RC(SER_MODE_FP32X);
    else if (mode == SER_MODE_BF16) SER_RC(SER_MODE_BF16);
    else if (mode == SER_MODE_FP16) SER_RC(SER_MODE_FP16);
    else if (mode == SER_MODE_FP16X) SER_RC(SER_MODE_FP16X);
    else if (mode == SER_MODE_FP16M) SER_RC(SER_MODE_FP16M);
    else return ser_fail(-4, "ser_row_center: bad mode %d", mode);
#undef SER_RC
    return ser_check_launch("ser_row_center");
}

extern "C" int ser_row_center(const float* x, int64_t ldx, void* out_act, int64_t ldo_act, int64_t out_plane_stride,
                              float* stats, int stat_groups, float* shift, int mode, int rows, int D, void* stream) {
    ser_row_center_args a = {};
    a.x = x; a.ldx = ldx; a.out_act = out_act; a.ldo_act = ldo_act; a.out_plane_stride = out_plane_stride; a.stats = stats; a.shift = shift;
    a.stat_groups = stat_groups; a.mode = mode; a.rows = rows; a.D = D;
    return ser_row_center_v(&a, stream);
}

// ------------------------------------------------------------------ text embeddings (8f-1)
// RoBERTa embeddings: word[id] + position[pos] + token_type[0] -> LayerNorm (HF modeling_roberta.py:56-120).
// Wave per token; position ids = cumsum(non-pad)*non-pad + pad_id, computed per sequence with a wave scan.
template <int MODE>
__global__ __launch_bounds__(256) void embed_ln_kernel(const int32_t* __restrict__ ids, const float* __restrict__ wemb,
                                                       const float* __restrict__ pemb, const float* __restrict__ temb,
                                                       const float* __restrict__ g, const float* __restrict__ b, float eps,
                                                       float* __restrict__ of, unsigned short* __restrict__ oa, int64_t plane,
                                                       int T, int D, int pad_id, int rows) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int seq = row / T, t = row - seq * T;
    // position id: number of non-pad tokens in ids[seq][0..t] (T <= 512), wave-parallel count
    int cnt = 0;
    for (int i = lane; i <= t; i += 64) cnt += (ids[seq * T + i] != pad_id);
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o, 64);
    const int id = ids[row];
    const int pos = (id != pad_id) ? cnt + pad_id : pad_id;
    const float* w = wemb + (int64_t)id * D;
    const float* pe = pemb + (int64_t)pos * D;
    f32x4 v[8];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int c = i * 256 + lane * 4;
        if (c < D) {
            const f32x4 a = *(const f32x4*)(w + c), p4 = *(const f32x4*)(pe + c), t4 = *(const f32x4*)(temb + c);
            v[i] = (a + p4) + t4;
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
            for (int j = 0; j < 4; ++j) y[j] = (v[i][j] - mean) * rstd * gg[j] + bb[j];
            if (of) *(f32x4*)(of + (int64_t)row * D + c) = y;
            if (oa) store_act4<MODE>(oa + (int64_t)row * D + c, plane, y[0], y[1], y[2], y[3]);
        }
    }
}

extern "C" int ser_embed_ln(const int32_t* ids, const float* word_emb, const float* pos_emb, const float* type_emb,
                            const float* ln_g, const float* ln_b, float eps, float* out_f32, void* out_act,
                            int64_t out_plane_stride, int mode, int B, int T, int D, int pad_id, void* stream) {
    if (!ids || !word_emb || !pos_emb || !type_emb || !ln_g || !ln_b || (!out_f32 && !out_act))
        return ser_fail(-1, "ser_embed_ln: null pointer");
    if (B <= 0 || T <= 0 || D % 4 || D > 2048) return ser_fail(-2, "ser_embed_ln: bad B/T/D");
    if (mode != SER_MODE_BF16 && mode != SER_MODE_FP32X && mode != SER_MODE_FP16X) return ser_fail(-3, "ser_embed_ln: bad mode");
    const int rows = B * T;
    dim3 grid((rows + 3) / 4), block(256);
    if (mode == SER_MODE_FP32X)
        hipLaunchKernelGGL(embed_ln_kernel<SER_MODE_FP32X>, grid, block, 0, (hipStream_t)stream, ids, word_emb, pos_emb, type_emb,
                           ln_g, ln_b, eps, out_f32, (unsigned short*)out_act, out_plane_stride, T, D, pad_id, rows);
    else if (mode == SER_MODE_FP16X)
        hipLaunchKernelGGL(embed_ln_kernel<SER_MODE_FP16X>, grid, block, 0, (hipStream_t)stream, ids, word_emb, pos_emb, type_emb,
                           ln_g, ln_b, eps, out_f32, (unsigned short*)out_act, out_plane_stride, T, D, pad_id, rows);
    else
        hipLaunchKernelGGL(embed_ln_kernel<SER_MODE_BF16>, grid, block, 0, (hipStream_t)stream, ids, word_emb, pos_emb, type_emb,
                           ln_g, ln_b, eps, out_f32, (unsigned short*)out_act, out_plane_stride, T, D, pad_id, rows);
    return ser_check_launch("ser_embed_ln");
}

// ------------------------------------------------------------------------------ K8a
__global__ void bias_table_kernel(const float* __restrict__ emb, float* __restrict__ table, int T, int H,
                                  int num_buckets, int max_distance) {
    const int W = 2 * T - 1;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= W) return;
    const int rel = idx - (T - 1);                               // key - query
    const int nb = num_buckets / 2, max_exact = nb / 2;
    const int n = rel < 0 ? -rel : rel;
    int bucket = rel > 0 ? nb : 0;
    if (n < max_exact) bucket += n;
    else {
        // float32 arithmetic in the order HF uses: log(n/max_exact) / log(max_distance/max_exact) * (nb-max_exact)
        float v = logf((float)n / (float)max_exact);
        v = v / (float)log((double)max_distance / (double)max_exact);
        v = v * (float)(nb - max_exact);
        int big = (int)((float)max_exact + v);
        bucket += big < nb - 1 ? big : nb - 1;
    }
    for (int h = 0; h < H; ++h) table[(int64_t)h * W + idx] = emb[(int64_t)bucket * H + h];
}

extern "C" int ser_wavlm_bias_table(const float* rel_attn_embed, float* table, int T, int H, int num_buckets,
                                    int max_distance, void* stream) {
    if (!rel_attn_embed || !table || T <= 0 || H <= 0) return ser_fail(-1, "ser_wavlm_bias_table: bad arguments");
    const int W = 2 * T - 1;
    hipLaunchKernelGGL(bias_table_kernel, dim3((W + 255) / 256), dim3(256), 0, (hipStream_t)stream, rel_attn_embed,
                       table, T, H, num_buckets, max_distance);
    return ser_check_launch("ser_wavlm_bias_table");
}

// ------------------------------------------------------------------------------ K8b
// gate[row,h] = a*(b*const_h - 1) + 2,  (a,b) = sigmoid(sum_{j<4} (w_j.x + b_j)), sigmoid(sum_{j>=4} ...)
// Wave per row; a lane loads 16-byte chunks (8 channels of ONE head, dh % 8 == 0), so a head is a
// group of dh/8 consecutive lanes and both dot products reduce with log2(dh/8) shuffles.
template <int MODE>
__global__ __launch_bounds__(256) void gate_kernel(const unsigned short* __restrict__ x, int64_t ldx, int64_t plane,
                                                   const float* __restrict__ w8, const float* __restrict__ b8,
                                                   const float* __restrict__ gconst, float* __restrict__ gate,
                                                   int rows, int H, int dh) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lpg = dh >> 3;                                     // lanes per head (power of two)
    const int d0 = (lane & (lpg - 1)) * 8;                       // this lane's offset inside its head
    float wa[8], wb[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        wa[j] = (w8[0 * dh + d0 + j] + w8[1 * dh + d0 + j]) + (w8[2 * dh + d0 + j] + w8[3 * dh + d0 + j]);
        wb[j] = (w8[4 * dh + d0 + j] + w8[5 * dh + d0 + j]) + (w8[6 * dh + d0 + j] + w8[7 * dh + d0 + j]);
    }
    const float ba = (b8[0] + b8[1]) + (b8[2] + b8[3]);
    const float bb = (b8[4] + b8[5]) + (b8[6] + b8[7]);
    const unsigned short* xr = x + (int64_t)row * ldx;
    const int D = H * dh;
    for (int c = lane * 8; c < D; c += 512) {
        float v[8];
        load_act8<MODE>(xr + c, plane, v);
        float pa = 0.f, pb = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) { pa = fmaf(v[j], wa[j], pa); pb = fmaf(v[j], wb[j], pb); }
        for (int o = lpg >> 1; o > 0; o >>= 1) { pa += __shfl_xor(pa, o, 64); pb += __shfl_xor(pb, o, 64); }
        if ((lane & (lpg - 1)) == 0) {
            const int h = c / dh;
            const float a = 1.f / (1.f + expf(-(pa + ba)));
            const float b = 1.f / (1.f + expf(-(pb + bb)));
            gate[(int64_t)row * H + h] = a * (b * gconst[h] - 1.f) + 2.f;
        }
    }
}

extern "C" int ser_wavlm_gate(const void* x_ln, int64_t ldx, int64_t plane_stride, int mode, const float* w8,
                              const float* b8, const float* gru_const, float* gate, int rows, int H, int dh,
                              void* stream) {
    if (!x_ln || !w8 || !b8 || !gru_const || !gate) return ser_fail(-1, "ser_wavlm_gate: null pointer");
    if ((dh != 8 && dh != 16 && dh != 32 && dh != 64 && dh != 128) || rows <= 0 || (ldx % 8))
        return ser_fail(-2, "ser_wavlm_gate: dh=%d unsupported (8..128, power of two)", dh);
    dim3 grid((rows + 3) / 4), block(256);
    if (mode == SER_MODE_FP32X)
        hipLaunchKernelGGL(gate_kernel<SER_MODE_FP32X>, grid, block, 0, (hipStream_t)stream, (const unsigned short*)x_ln,
                           ldx, plane_stride, w8, b8, gru_const, gate, rows, H, dh);
    else
        hipLaunchKernelGGL(gate_kernel<SER_MODE_BF16>, grid, block, 0, (hipStream_t)stream, (const unsigned short*)x_ln,
                           ldx, plane_stride, w8, b8, gru_const, gate, rows, H, dh);
    return ser_check_launch("ser_wavlm_gate");
}

// ------------------------------------------------------------------------------ K15
__global__ void mean4_kernel(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ c,
                             const float* __restrict__ d, float* __restrict__ o, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    // torch.mean(torch.stack(hs[-4:]), dim=0): sequential sum then divide
    if (i < n) o[i] = (((a[i] + b[i]) + c[i]) + d[i]) / 4.0f;
}

extern "C" int ser_mean4(const float* s0, const float* s1, const float* s2, const float* s3, float* out, int64_t n,
                         void* stream) {
    if (!s0 || !s1 || !s2 || !s3 || !out || n <= 0) return ser_fail(-1, "ser_mean4: bad arguments");
    hipLaunchKernelGGL(mean4_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, s0, s1, s2,
                       s3, out, n);
    return ser_check_launch("ser_mean4");
}

// ------------------------------------------------------------------- weights / packing
__global__ void split_kernel(const float* __restrict__ x, unsigned short* __restrict__ o, int64_t plane, int mode,
                             int64_t n) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t step = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += step) {
        if (mode == SER_MODE_FP16) {
            o[i] = (unsigned short)(pack_h2(x[i], 0.f) & 0xffffu);
            continue;
        }
        if (mode == SER_MODE_FP16X) {
            unsigned short h, l;
            split_h(x[i], h, l);
            o[i] = h;
            o[plane + i] = l;
            continue;
        }
        unsigned short h, l;
        split_bf(x[i], h, l);
        o[i] = h;
        if (mode == SER_MODE_FP32X) o[plane + i] = l;
    }
}

// out[m] = (base[b] + step * (m - row_offs[b])) * mult / div with b = utterance owning output row m
__global__ void ragged_index_kernel(const int32_t* __restrict__ row_offs, const int64_t* __restrict__ base, int B,
                                    int64_t step, int64_t mult, int64_t div, int32_t* __restrict__ out, int64_t total) {
    const int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= total) return;
    int lo = 0, hi = B;                              // largest b with row_offs[b] <= m (empty utterances skipped)
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if ((int64_t)row_offs[mid] <= m) lo = mid; else hi = mid;
    }
    out[m] = (int32_t)((base[lo] + step * (m - (int64_t)row_offs[lo])) * mult / div);
}

extern "C" int ser_ragged_index(const int32_t* row_offs, const int64_t* base, int B, int64_t step, int64_t mult, int64_t div,
                                int32_t* out, int64_t total_rows, void* stream) {
    if (!row_offs || !base || !out) return ser_fail(-1, "ser_ragged_index: null pointer");
    if (B <= 0 || total_rows < 0 || div <= 0) return ser_fail(-2, "ser_ragged_index: bad B=%d / rows / div", B);
    if (total_rows == 0) return 0;
    const int64_t blocks = (total_rows + 255) / 256;
    hipLaunchKernelGGL(ragged_index_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, row_offs, base, B,
                       step, mult, div, out, total_rows);
    return ser_check_launch("ser_ragged_index");
}

extern "C" int ser_split_bf16(const float* x, void* out, int64_t plane_stride, int mode, int64_t n, void* stream) {
    if (!x || !out || n <= 0) return ser_fail(-1, "ser_split_bf16: bad arguments");
    if (mode < SER_MODE_BF16 || mode > SER_MODE_FP16X) return ser_fail(-2, "ser_split_bf16: bad mode %d", mode);
    int64_t blocks = (n + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    hipLaunchKernelGGL(split_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x,
                       (unsigned short*)out, plane_stride, mode, n);
    return ser_check_launch("ser_split_bf16");
}

// [B, C, T] fp32 -> channels-last act rows with `halo` zero rows around each utterance:
// utterance b occupies rows b*(T+2*halo) .. ; rows [halo, halo+T) carry data.
template <int MODE>
__global__ void pack_act_kernel(const float* __restrict__ x, int B, int C, int T, int halo,
                                unsigned short* __restrict__ o, int64_t ldo, int64_t plane, uint32_t* __restrict__ rflag) {
    const int Tp = T + 2 * halo;
    const int64_t total = (int64_t)B * Tp * (C / 4);
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int tp = (int)(i % Tp);
    const int64_t r = i / Tp;
    const int cg = (int)(r % (C / 4));
    const int b = (int)(r / (C / 4));
    const int t = tp - halo;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if (t >= 0 && t < T) {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = x[((int64_t)b * C + cg * 4 + j) * T + t];
    }
    store_act4<MODE>(o + ((int64_t)b * Tp + tp) * ldo + cg * 4, plane, v[0], v[1], v[2], v[3]);
    if constexpr (mode_traits<MODE>::f16) range_report(rflag, fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
}

extern "C" int ser_pack_act_v(const ser_pack_act_args* a, void* stream) {
    if (!a) return ser_fail(-1, "ser_pack_act: bad arguments");
    const float* x = a->x; void* out = a->out; const int B = a->B, C = a->C, T = a->T, halo = a->halo, mode = a->mode;
    const int64_t ldo = a->ldo, out_plane_stride = a->out_plane_stride;
    if (!x || !out || B <= 0 || C % 4 || T <= 0 || halo < 0) return ser_fail(-1, "ser_pack_act: bad arguments");
    const int64_t total = (int64_t)B * (T + 2 * halo) * (C / 4);
    dim3 grid((unsigned)((total + 255) / 256)), block(256);
#define SER_PA(M_) hipLaunchKernelGGL(pack_act_kernel<M_>, grid, block, 0, (hipStream_t)stream, x, B, C, T, halo, (unsigned short*)out, ldo, \
                                      out_plane_stride, a->range_flag)
    if (mode == SER_MODE_FP32X) SER_PA(SER_MODE_FP32X);
    else if (mode == SER_MODE_FP16X) SER_PA(SER_MODE_FP16X);
    else SER_PA(SER_MODE_BF16);
#undef SER_PA
    return ser_check_launch("ser_pack_act");
}

extern "C" int ser_pack_act(const float* x, int B, int C, int T, int halo, void* out, int64_t ldo,
                            int64_t out_plane_stride, int mode, void* stream) {
    ser_pack_act_args a = {};
    a.x = x; a.out = out; a.ldo = ldo; a.out_plane_stride = out_plane_stride; a.B = B; a.C = C; a.T = T; a.halo = halo; a.mode = mode;
    return ser_pack_act_v(&a, stream);
}

// ------------------------------------------------------------------ SER_MODE_FP16M packing
// fp32 [rows][ldx] -> hi plane + e4m3 cross-term plane + block scales (ser_hip.h).  A lane owns 8 consecutive columns, 4 lanes a 32-column
// scale block, 8 lanes a 64-column tile; a wave walks 512 columns of one row at a time.  Weights at load, kernel tests.
__global__ __launch_bounds__(256) void pack_f16m_kernel(const float* __restrict__ x, int64_t ldx, int rows, int cols,
                                                        unsigned short* __restrict__ out, int64_t ldo, int64_t plane,
                                                        uint32_t* __restrict__ scales, int64_t sld, int weight, uint32_t* __restrict__ rflag) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    float ramax = 0.f;
    for (int c0 = 0; c0 < cols; c0 += 512) {
        const int c = c0 + lane * 8;
        float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (c < cols) {
            const f32x4 a = *(const f32x4*)(x + (int64_t)row * ldx + c), b = *(const f32x4*)(x + (int64_t)row * ldx + c + 4);
            v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3]; v[4] = b[0]; v[5] = b[1]; v[6] = b[2]; v[7] = b[3];
        }
        ramax = fmaxf(ramax, mx_store_row<8>(v, c < cols, out + (int64_t)row * ldo, (unsigned char*)(out + plane + (int64_t)row * ldo), c,
                                             scales + row, sld, weight != 0));
    }
    range_report(rflag, ramax);
}

extern "C" int ser_pack_f16m(const float* x, int64_t ldx, int rows, int cols, void* out, int64_t ldo, int64_t plane_stride,
                             uint32_t* scales, int64_t scale_ld, int is_weight, uint32_t* range_flag, void* stream) {
    if (!x || !out || !scales) return ser_fail(-1, "ser_pack_f16m: null pointer");
    if (rows <= 0 || cols <= 0 || (cols % 64) || (ldo % 64) || (ldx % 4) || scale_ld < rows || ldo < cols)
        return ser_fail(-2, "ser_pack_f16m: rows=%d cols=%d (cols %% 64 == 0, ldo %% 64 == 0, ldx %% 4 == 0, scale_ld >= rows)", rows, cols);
    hipLaunchKernelGGL(pack_f16m_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, ldx, rows, cols,
                       (unsigned short*)out, ldo, plane_stride, scales, scale_ld, is_weight, range_flag);
    return ser_check_launch("ser_pack_f16m");
}
