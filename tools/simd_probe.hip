// What two waves of one SIMD can overlap on gfx950 (MI355X), measured with s_memtime inside the kernel.
//   hipcc -O3 --offload-arch=gfx950 tools/simd_probe.hip -o tools/ab/simd_probe && tools/ab/simd_probe
// One block of 64 * W threads per CU with W = 4 (one wave per SIMD) or 8 (two per SIMD: waves w and w + 4 share a SIMD).
// role 0: a wave issues N v_mfma_f32_32x32x16_bf16 on `chains` independent accumulators;  role 1: N v_fma_f32;  role 2: N v_exp_f32;
// role 3: idle.  Waves 0-3 take roleA, waves 4-7 roleB.  Reported: cycles per instruction of the first wave of each half.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int CHAINS>
__device__ __forceinline__ float run_mfma(int n, float seed) {
    f32x16 acc[CHAINS];
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(seed + i); b[i] = (__bf16)(seed - i); }
    for (int c = 0; c < CHAINS; ++c) for (int r = 0; r < 16; ++r) acc[c][r] = seed;
    for (int i = 0; i < n; i += CHAINS)
#pragma unroll
        for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[c], 0, 0, 0);
    float s = 0.f;
    for (int c = 0; c < CHAINS; ++c) s += acc[c][0];
    return s;
}
__device__ __forceinline__ float run_fma(int n, float seed) {
    float x[8];
    for (int i = 0; i < 8; ++i) x[i] = seed + i;
    for (int i = 0; i < n; i += 64)
#pragma unroll
        for (int j = 0; j < 64; ++j) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(x[j & 7]) : "v"(seed));
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += x[i];
    return s;
}
__device__ __forceinline__ float run_exp(int n, float seed) {
    float x[8];
    for (int i = 0; i < 8; ++i) x[i] = seed * 0.01f + i * 0.001f;
    for (int i = 0; i < n; i += 64)
#pragma unroll
        for (int j = 0; j < 64; ++j) asm volatile("v_exp_f32 %0, %0" : "+v"(x[j & 7]));
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += x[i];
    return s;
}

template <int CHAINS>
__global__ __launch_bounds__(512, 1) void probe(int roleA, int roleB, int n, float seed, unsigned long long* out, float* sink, int prioA, int prioB) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int role = wave < 4 ? roleA : roleB;
    const int prio = wave < 4 ? prioA : prioB;
    if (prio == 1) __builtin_amdgcn_s_setprio(1);
    if (prio == 2) __builtin_amdgcn_s_setprio(2);
    if (prio == 3) __builtin_amdgcn_s_setprio(3);
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    float r = 0.f;
    if (role == 0) r = run_mfma<CHAINS>(n, seed);
    else if (role == 1) r = run_fma(n, seed);
    else if (role == 2) r = run_exp(n, seed);
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 7) out[wave] = t1 - t0;
    if (r == 12345.678f) sink[threadIdx.x] = r;
}

int main() {
    unsigned long long* out; float* sink;
    hipMalloc(&out, 64); hipMalloc(&sink, 4096);
    const char* names[] = {"mfma", "fma", "exp", "idle"};
    const int N = 4096;
    auto run = [&](int chains, int threads, int ra, int rb, int pa = 0, int pb = 0) {
        hipMemset(out, 0, 64);
        for (int rep = 0; rep < 3; ++rep) {
            if (chains == 1) hipLaunchKernelGGL(probe<1>, dim3(256), dim3(threads), 0, 0, ra, rb, N, 1.0f, out, sink, pa, pb);
            else if (chains == 2) hipLaunchKernelGGL(probe<2>, dim3(256), dim3(threads), 0, 0, ra, rb, N, 1.0f, out, sink, pa, pb);
            else hipLaunchKernelGGL(probe<4>, dim3(256), dim3(threads), 0, 0, ra, rb, N, 1.0f, out, sink, pa, pb);
        }
        hipDeviceSynchronize();
        unsigned long long h[8];
        hipMemcpy(h, out, 64, hipMemcpyDeviceToHost);
        printf("chains %d, %d waves/SIMD: waves 0-3 %-4s %6.1f cyc/instr", chains, threads / 256, names[ra], (double)h[0] / N);
        if (threads == 512) printf("   | waves 4-7 %-4s %6.1f cyc/instr   (prio %d / %d)", names[rb], (double)h[4] / N, pa, pb);
        printf("\n");
    };
    for (int c : {1, 2, 4}) run(c, 256, 0, 3);
    run(4, 256, 1, 3); run(4, 256, 2, 3);
    printf("-- two waves per SIMD, same role\n");
    for (int c : {1, 2, 4}) run(c, 512, 0, 0);
    run(4, 512, 1, 1); run(4, 512, 2, 2);
    printf("-- two waves per SIMD, different roles (overlap if each keeps its solo rate)\n");
    run(4, 512, 0, 1); run(4, 512, 0, 2); run(1, 512, 0, 1); run(4, 512, 1, 2);
    printf("-- roles swapped: the VALU wave is the older one\n");
    run(4, 512, 1, 0); run(4, 512, 2, 0);
    printf("-- s_setprio: VALU wave raised above the MFMA wave\n");
    run(4, 512, 0, 1, 0, 1); run(4, 512, 0, 1, 0, 3); run(4, 512, 0, 2, 0, 3); run(4, 512, 1, 0, 3, 0);
    printf("-- s_setprio: MFMA wave raised\n");
    run(4, 512, 0, 1, 3, 0); run(4, 512, 1, 0, 0, 3);
    return 0;
}
