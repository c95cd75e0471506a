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

__device__ __forceinline__ int f2ord(float f) {
    const int i = __float_as_int(f);
    return i >= 0 ? i : i ^ 0x7fffffff;
}
__device__ __forceinline__ float ord2f(int i) {
    return __int_as_float(i >= 0 ? i : i ^ 0x7fffffff);
}

// work layout: [0, B*64) ints (one 256-byte granule per utterance max), then the twiddle table
__global__ void logmel_init_kernel(int* umax, int B, double2* tw) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < B) umax[i * 64] = f2ord(-INFINITY);
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
                                                     int* __restrict__ umax) {
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
    lmax = wave_max(lmax);
    if ((tid & 63) == 0) atomicMax(&umax[b * 64], f2ord(lmax));
}

__global__ void logmel_finish_kernel(float* __restrict__ out, const int* __restrict__ umax, int n_mels) {
    const int b = blockIdx.y;
    const int64_t per = (int64_t)n_mels * LM_FRAMES;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= per) return;
    const float mx = ord2f(umax[b * 64]);
    float* o = out + (int64_t)b * per;
    o[i] = (fmaxf(o[i], mx - 8.0f) + 4.0f) / 4.0f;
}

extern "C" int ser_logmel_whisper(const float* wav, const int64_t* sample_offs, int B, const float* mel, int n_mels,
                                  float* out, void* work, void* stream) {
    if (!wav || !sample_offs || !mel || !out || !work || B <= 0 || n_mels <= 0)
        return ser_fail(-1, "ser_logmel_whisper: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    int* umax = (int*)work;
    double2* tw = (double2*)((char*)work + (((size_t)B * 256 + 255) / 256) * 256);
    const int ninit = LM_NFFT * LM_BINS > B ? LM_NFFT * LM_BINS : B;
    hipLaunchKernelGGL(logmel_init_kernel, dim3((ninit + 255) / 256), dim3(256), 0, s, umax, B, tw);
    hipLaunchKernelGGL(logmel_kernel, dim3((LM_FRAMES + LM_FR - 1) / LM_FR, B), dim3(256), 0, s, wav, sample_offs, mel,
                       n_mels, tw, out, umax);
    const int64_t per = (int64_t)n_mels * LM_FRAMES;
    hipLaunchKernelGGL(logmel_finish_kernel, dim3((unsigned)((per + 255) / 256), B), dim3(256), 0, s, out, umax, n_mels);
    return ser_check_launch("ser_logmel_whisper");
}
