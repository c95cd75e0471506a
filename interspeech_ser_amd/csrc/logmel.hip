// Whisper log-mel front end on the GPU (ser_logmel_whisper).
// The reference computes this on the CPU inside WhisperFeatureExtractor
// (HF feature_extraction_whisper.py:135-169; call site preprocess_whisper.py:48).
//
//   zero-pad / truncate to 480000 -> reflect pad 200 -> frames of 400, hop 160 -> periodic Hann
//   -> 400-point real DFT (201 bins) -> |.|^2 -> drop frame 3000 -> mel^T (128x201) @ -> log10(max(.,1e-10))
//   -> max(., utterance_max - 8) -> (. + 4) / 4
//
// The DFT is a dense [frames x 400] x [400 x 402] product against a twiddle table; it is
// accumulated in fp64 (MI355X runs fp64 FMA at half the fp32 vector rate and the whole front
// end is < 0.1 % of the encoder's FLOPs) so that bins 8 decades below the maximum keep the
// accuracy the fp32 pocketfft path of the reference has.
#include "ser_common.h"

#define LM_NFFT 400
#define LM_HOP 160
#define LM_BINS 201
#define LM_FRAMES 3000
#define LM_SAMPLES 480000
#define LM_FR 16              // frames per block

#define LM_NBLK ((LM_FRAMES + LM_FR - 1) / LM_FR)    // frame blocks per utterance (188)
#define LM_BMAX 256                                   // partial-maximum slots per utterance (>= LM_NBLK)

// work layout: the fp64 twiddle table [400][201] (built once per buffer by ser_logmel_init), then [B][LM_BMAX] floats
// (per-block maxima of one call)
#define LM_TW_BYTES ((size_t)LM_NFFT * LM_BINS * sizeof(double2))
__global__ void logmel_init_kernel(double2* tw) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < LM_NFFT * LM_BINS) {
        const int n = i / LM_BINS, k = i - n * LM_BINS;
        const int r = (n * k) % LM_NFFT;                     // exact argument reduction
        double s, c;
        sincospi(2.0 * (double)r / (double)LM_NFFT, &s, &c);
        tw[i] = make_double2(c, -s);
    }
}

__global__ __launch_bounds__(256) void logmel_kernel(const float* __restrict__ wav, const int64_t* __restrict__ offs,
                                                     const float* __restrict__ mel, int n_mels,
                                                     const double2* __restrict__ tw, float* __restrict__ out,
                                                     float* __restrict__ bmax) {
    __shared__ float xw[LM_FR][LM_NFFT];
    __shared__ float pw[LM_FR][LM_BINS + 3];
    const int b = blockIdx.y, f0 = blockIdx.x * LM_FR, tid = threadIdx.x;
    const int64_t s0 = offs[b];
    const int64_t len = offs[b + 1] - s0;
    for (int i = tid; i < LM_FR * LM_NFFT; i += 256) {
        const int f = i / LM_NFFT, n = i - f * LM_NFFT;
        int s = (f0 + f) * LM_HOP - LM_NFFT / 2 + n;
        if (s < 0) s = -s;
        if (s >= LM_SAMPLES) s = 2 * (LM_SAMPLES - 1) - s;
        const float x = (s < len) ? wav[s0 + s] : 0.f;
        const float w = 0.5f - 0.5f * cospif(2.0f * (float)n / (float)LM_NFFT);
        xw[f][n] = x * w;
    }
    __syncthreads();
    if (tid < LM_BINS) {
        double re[LM_FR], im[LM_FR];
#pragma unroll
        for (int f = 0; f < LM_FR; ++f) { re[f] = 0.0; im[f] = 0.0; }
        for (int n = 0; n < LM_NFFT; ++n) {
            const double2 c = tw[n * LM_BINS + tid];
#pragma unroll
            for (int f = 0; f < LM_FR; ++f) {
                const double x = (double)xw[f][n];
                re[f] = fma(x, c.x, re[f]);
                im[f] = fma(x, c.y, im[f]);
            }
        }
#pragma unroll
        for (int f = 0; f < LM_FR; ++f) {
            const float fr = (float)re[f], fi = (float)im[f];
            pw[f][tid] = fr * fr + fi * fi;
        }
    }
    __syncthreads();
    float lmax = -INFINITY;
    for (int m = tid & 127; m < n_mels; m += 128) {
        const int fb = (tid >> 7) * (LM_FR / 2);
        float acc[LM_FR / 2];
#pragma unroll
        for (int f = 0; f < LM_FR / 2; ++f) acc[f] = 0.f;
        for (int k = 0; k < LM_BINS; ++k) {
            const float w = mel[k * n_mels + m];
#pragma unroll
            for (int f = 0; f < LM_FR / 2; ++f) acc[f] = fmaf(w, pw[fb + f][k], acc[f]);
        }
#pragma unroll
        for (int f = 0; f < LM_FR / 2; ++f) {
            const int fr = f0 + fb + f;
            if (fr < LM_FRAMES) {
                const float v = log10f(fmaxf(acc[f], 1e-10f));
                out[((int64_t)b * n_mels + m) * LM_FRAMES + fr] = v;
                lmax = fmaxf(lmax, v);
            }
        }
    }
    // block maximum -> its own slot: the finishing pass reduces the 188 slots of an utterance (no atomics, no reset)
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
    hipLaunchKernelGGL(logmel_init_kernel, dim3((LM_NFFT * LM_BINS + 255) / 256), dim3(256), 0, (hipStream_t)stream, tw);
    return ser_check_launch("ser_logmel_init");
}

extern "C" int ser_logmel_whisper(const float* wav, const int64_t* sample_offs, int B, const float* mel, int n_mels,
                                  float* out, void* work, void* stream) {
    if (!wav || !sample_offs || !mel || !out || !work || B <= 0 || n_mels <= 0)
        return ser_fail(-1, "ser_logmel_whisper: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    const double2* tw = (const double2*)work;                                            // ser_logmel_init
    float* bmax = (float*)((char*)work + LM_TW_BYTES);
    hipLaunchKernelGGL(logmel_kernel, dim3(LM_NBLK, B), dim3(256), 0, s, wav, sample_offs, mel, n_mels, tw, out, bmax);
    const int64_t per = (int64_t)n_mels * LM_FRAMES;
    hipLaunchKernelGGL(logmel_finish_kernel, dim3((unsigned)((per / 4 + 255) / 256), B), dim3(256), 0, s, out, bmax, n_mels);
    return ser_check_launch("ser_logmel_whisper");
}
