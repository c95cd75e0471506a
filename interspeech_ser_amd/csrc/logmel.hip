// Whisper log-mel front end on the GPU (ser_logmel_whisper).
// The reference computes this on the CPU inside WhisperFeatureExtractor
// (HF feature_extraction_whisper.py:135-169; call site preprocess_whisper.py:48).
//
//   zero-pad / truncate to 480000 -> reflect pad 200 -> frames of 400, hop 160 -> periodic Hann
//   -> 400-point real DFT (201 bins) -> |.|^2 -> drop frame 3000 -> mel^T (128x201) @ -> log10(max(.,1e-10))
//   -> max(., utterance_max - 8) -> (. + 4) / 4
//
// The DFT is a dense [frames x 400] x [400 x 402] product against a twiddle table, run on the matrix cores in
// fp64 (v_mfma_f64_16x16x4_f64; the whole front end is < 0.1 % of the encoder's FLOPs) so that bins 8 decades below
// the maximum keep the accuracy the fp32 pocketfft path of the reference has.
#include "ser_common.h"

#define LM_NFFT 400
#define LM_HOP 160
#define LM_BINS 201
#define LM_FRAMES 3000
#define LM_SAMPLES 480000
#define LM_FR 32              // frames per block: two 16-row MFMA tiles
#define LM_BB 13              // 16-bin column blocks (208 >= 201 bins)
#define LM_XP 404             // LDS row pitch of the windowed frames (floats): 404 = 20 (mod 64) -> the 16 rows x 4 taps
                              // of one MFMA A fragment fall on 64 distinct banks
#define LM_PP 212             // LDS row pitch of the power spectrum

#define LM_NBLK ((LM_FRAMES + LM_FR - 1) / LM_FR)    // frame blocks per utterance (94)
#define LM_BMAX 256                                   // partial-maximum slots per utterance (>= LM_NBLK)

// work layout: the fp64 twiddle table [400][208] of (cos, -sin) (built once per buffer by ser_logmel_init), then
// [B][LM_BMAX] floats (per-block maxima of one call)
#define LM_TW_BYTES ((size_t)LM_NFFT * LM_BB * 16 * sizeof(double2))
#define LM_HANN_BYTES ((size_t)LM_NFFT * sizeof(float))     // periodic Hann window, after the twiddles

typedef __attribute__((ext_vector_type(4))) double f64x4;

__global__ void logmel_init_kernel(double2* tw) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < LM_NFFT) ((float*)((char*)tw + LM_TW_BYTES))[i] = 0.5f - 0.5f * cospif(2.0f * (float)i / (float)LM_NFFT);
    if (i < LM_NFFT * LM_BB * 16) {
        const int n = i / (LM_BB * 16), k = i - n * (LM_BB * 16);
        double s = 0.0, c = 0.0;
        if (k < LM_BINS) {
            const int r = (n * k) % LM_NFFT;                 // exact argument reduction
            sincospi(2.0 * (double)r / (double)LM_NFFT, &s, &c);
        }
        tw[i] = make_double2(c, -s);                         // bins 201..207 are padding: zero columns
    }
}

// DFT on the matrix cores: [32 frames x 400 taps] x [400 x (201 bins x {re, im})] with v_mfma_f64_16x16x4_f64 --
// fp64 products and accumulation, so bins 8 decades below the maximum keep the accuracy the reference's fp32
// pocketfft path has (a plain fp32 sum of 400 terms does not), at the fp64 MFMA rate instead of an fp64 VALU loop on
// 201 of 256 lanes.  Wave w owns the 16-bin column blocks w, w+4, w+8(, w+12); one 16-byte load of (cos, -sin) feeds
// the re and the im MFMA of a block.  Operand lane maps (MI355X guide): A[i = lane & 15][k = lane >> 4],
// B[k = lane >> 4][j = lane & 15], D[row = (lane >> 4) + 4 r][col = lane & 15].
// three waves per SIMD (168 VGPRs, 16 spilled; 52 KB of LDS per block): measured 209 us per 8 x 30 s against 237 us at two
__global__ __launch_bounds__(256, 3) void logmel_kernel(const float* __restrict__ wav, const int64_t* __restrict__ offs,
                                                     const float* __restrict__ mel, int n_mels,
                                                     const double2* __restrict__ tw, float* __restrict__ out,
                                                     float* __restrict__ bmax) {
    __shared__ __attribute__((aligned(16))) float lds[LM_FR * LM_XP];       // windowed frames, later the power spectrum
    const int b = blockIdx.y, f0 = blockIdx.x * LM_FR, tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int64_t s0 = offs[b];
    const int64_t len = offs[b + 1] - s0;
    const float* hann = (const float*)((const char*)tw + LM_TW_BYTES);
    for (int i = tid; i < LM_FR * LM_NFFT; i += 256) {
        const int f = i / LM_NFFT, n = i - f * LM_NFFT;
        int s = (f0 + f) * LM_HOP - LM_NFFT / 2 + n;
        if (s < 0) s = -s;
        if (s >= LM_SAMPLES) s = 2 * (LM_SAMPLES - 1) - s;
        const float x = (s < len) ? wav[s0 + s] : 0.f;
        lds[f * LM_XP + n] = x * hann[n];
    }
    __syncthreads();

    constexpr int NB = 4;                                    // column blocks per wave (wave 0 uses all four, the others three)
    const int nb = wave == 0 ? 4 : 3;
    const int li = lane & 15, kk = lane >> 4;
    f64x4 re[2][NB], im[2][NB];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int q = 0; q < NB; ++q) { re[rt][q] = (f64x4){0.0, 0.0, 0.0, 0.0}; im[rt][q] = (f64x4){0.0, 0.0, 0.0, 0.0}; }
    const double2* twl = tw + (size_t)kk * (LM_BB * 16) + wave * 16 + li;       // + step * 4 rows, + q * 64 columns
    // one wave per SIMD (the accumulators take half the register file): the operands of step st+1 are requested before
    // the 16 MFMAs of step st issue, so their L2 / LDS latency sits under ~1000 cycles of matrix work
    double a0 = (double)lds[li * LM_XP + kk], a1 = (double)lds[(16 + li) * LM_XP + kk];
    double2 c[NB];
#pragma unroll
    for (int q = 0; q < NB; ++q) c[q] = (q < nb) ? twl[q * 64] : make_double2(0.0, 0.0);
    for (int st = 0; st < LM_NFFT / 4; ++st) {
        const int sn = st + 1 < LM_NFFT / 4 ? st + 1 : st;
        const double n0 = (double)lds[li * LM_XP + 4 * sn + kk];
        const double n1 = (double)lds[(16 + li) * LM_XP + 4 * sn + kk];
        double2 cn[NB];
#pragma unroll
        for (int q = 0; q < NB; ++q)
            cn[q] = (q < nb) ? twl[(size_t)sn * 4 * (LM_BB * 16) + q * 64] : make_double2(0.0, 0.0);
#pragma unroll
        for (int q = 0; q < NB; ++q) {
            if (q < nb) {                                    // wave-uniform
                re[0][q] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, c[q].x, re[0][q], 0, 0, 0);
                im[0][q] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, c[q].y, im[0][q], 0, 0, 0);
                re[1][q] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, c[q].x, re[1][q], 0, 0, 0);
                im[1][q] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, c[q].y, im[1][q], 0, 0, 0);
            }
        }
        a0 = n0; a1 = n1;
#pragma unroll
        for (int q = 0; q < NB; ++q) c[q] = cn[q];
    }
    __syncthreads();                                         // every wave is done with the frames: reuse the LDS
    float* pw = lds;                                         // [LM_FR][LM_PP]
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int q = 0; q < NB; ++q)
            if (q < nb) {
                const int bin = (wave + 4 * q) * 16 + li;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float fr = (float)re[rt][q][r], fi = (float)im[rt][q][r];
                    pw[(rt * 16 + kk + 4 * r) * LM_PP + bin] = fr * fr + fi * fi;
                }
            }
    __syncthreads();
    // mel projection + log10: thread = (mel, half of the frames); 16 consecutive frames of one mel row are 64 contiguous bytes
    float lmax = -INFINITY;
    for (int m = tid & 127; m < n_mels; m += 128) {
        const int fb = (tid >> 7) * (LM_FR / 2);
        float acc[LM_FR / 2];
#pragma unroll
        for (int f = 0; f < LM_FR / 2; ++f) acc[f] = 0.f;
        // 8 filter weights requested per pass (201 = 25 x 8 + 1): one L2 latency per 8 bins instead of one per bin
        for (int k0 = 0; k0 < LM_BINS; k0 += 8) {
            float w[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) w[u] = (k0 + u < LM_BINS) ? mel[(k0 + u) * n_mels + m] : 0.f;
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int f = 0; f < LM_FR / 2; ++f) acc[f] = fmaf(w[u], pw[(fb + f) * LM_PP + k0 + u], acc[f]);
        }
        float* orow = out + ((int64_t)b * n_mels + m) * LM_FRAMES + f0 + fb;
#pragma unroll
        for (int f4 = 0; f4 < LM_FR / 2; f4 += 4) {
            if (f0 + fb + f4 < LM_FRAMES) {                  // 3000 is a multiple of 4: a quad is all valid or all padding
                f32x4 v;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    v[j] = log10f(fmaxf(acc[f4 + j], 1e-10f));
                    lmax = fmaxf(lmax, v[j]);
                }
                *(f32x4*)(orow + f4) = v;
            }
        }
    }
    // block maximum -> its own slot: the finishing pass reduces the 94 slots of an utterance (no atomics, no reset)
    __shared__ float wmax[4];
    lmax = wave_max(lmax);
    if ((tid & 63) == 0) wmax[tid >> 6] = lmax;
    __syncthreads();
    if (tid == 0) bmax[b * LM_BMAX + blockIdx.x] = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));
}

__global__ __launch_bounds__(256) void logmel_finish_kernel(float* __restrict__ out, const float* __restrict__ bmax, int n_mels) {
    const int b = blockIdx.y;
    __shared__ float red[4];
    float m = threadIdx.x < LM_NBLK ? bmax[b * LM_BMAX + threadIdx.x] : -INFINITY;
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    const float mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    const int64_t per = (int64_t)n_mels * LM_FRAMES;                 // a multiple of 4 (3000 frames)
    const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i >= per) return;
    f32x4* o = (f32x4*)(out + (int64_t)b * per + i);
    f32x4 v = *o;
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = (fmaxf(v[j], mx - 8.0f) + 4.0f) / 4.0f;
    *o = v;
}

extern "C" int ser_logmel_init(void* work, int B, void* stream) {
    (void)B;
    if (!work) return ser_fail(-1, "ser_logmel_init: bad arguments");
    double2* tw = (double2*)work;
    hipLaunchKernelGGL(logmel_init_kernel, dim3((LM_NFFT * LM_BB * 16 + 255) / 256), dim3(256), 0, (hipStream_t)stream, tw);
    return ser_check_launch("ser_logmel_init");
}

extern "C" int ser_logmel_whisper(const float* wav, const int64_t* sample_offs, int B, const float* mel, int n_mels,
                                  float* out, void* work, void* stream) {
    if (!wav || !sample_offs || !mel || !out || !work || B <= 0 || n_mels <= 0)
        return ser_fail(-1, "ser_logmel_whisper: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    const double2* tw = (const double2*)work;                                            // ser_logmel_init
    float* bmax = (float*)((char*)work + LM_TW_BYTES + LM_HANN_BYTES);
    hipLaunchKernelGGL(logmel_kernel, dim3(LM_NBLK, B), dim3(256), 0, s, wav, sample_offs, mel, n_mels, tw, out, bmax);
    const int64_t per = (int64_t)n_mels * LM_FRAMES;
    hipLaunchKernelGGL(logmel_finish_kernel, dim3((unsigned)((per / 4 + 255) / 256), B), dim3(256), 0, s, out, bmax, n_mels);
    return ser_check_launch("ser_logmel_whisper");
}
