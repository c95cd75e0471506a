// What the matrix pipe sustains at the board's power cap, by instruction shape and wave tile (tools/, not product code; round 5).
// The timed bf16 step runs at the 1 400 W cap (profiles/r05_power_sample_bf16_step.txt), so throughput follows energy per FLOP.  This probe
// runs GEMM-like inner loops -- fragments from LDS (random bf16 data, a different 1-KiB-aligned window every k-step) into MFMAs, all 256 CUs,
// two waves per SIMD, no global traffic -- for ~2 s each and prints the sustained TFLOP/s; a shell loop samples rocm-smi beside it
// (tools/energy_probe.sh).  Variants (wave tile, instruction):
//   m16_64x64    16 x v_mfma_f32_16x16x32_bf16 per k32 on 4 + 4 fragments     (the 256x128 tile's wave)
//   m32_64x64     8 x v_mfma_f32_32x32x16_bf16 per k32 on the same 8 fragments (half the instructions and VGPR operand reads per FLOP)
//   m16_128x64   32 x 16x16x32 per k32 on 8 + 4 fragments                      (the 256x256 tile's wave)
//   m32_128x64   16 x 32x32x16 per k32 on the same 12 fragments
//   m16_noread / m32_noread   the same instruction streams on register operands that never change (matrix pipe alone)
//   reads_only   the 64x64 loop's 8 ds_read_b128 per k32 without the MFMAs
//   dma_l2 / dma_mall / dma_hbm   m16_128x64 plus the 256x256 tile's L2 -> LDS traffic (4 global_load_lds_dwordx4 per wave and k32 = 64 KiB per
//                block and 64-deep K tile) from a source every block shares (2 MiB: L2 hits), from 0.5 MiB per block (128 MiB: L2 misses,
//                Infinity Cache hits) and from 6 MiB per block (1.5 GiB: HBM)
//   barrier      m16_128x64 plus one s_barrier per k32
// build: hipcc --offload-arch=gfx950 -O3 tools/energy_probe.hip -o tools/ab/energy_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <time.h>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int LDSB = 64 * 1024;            // per block (8 waves); two such blocks would not fit -> one block of 8 waves per CU = 2 waves per SIMD

// VAR: 0 m16_64x64, 1 m32_64x64, 2 m16_128x64, 3 m32_128x64, 4 m16_noread (64x64), 5 m32_noread (64x64), 6 reads_only (64x64)
__device__ __forceinline__ void dma16(const void* ubase, unsigned voff, unsigned lds_addr) {
    asm volatile("s_mov_b32 m0, %2\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(voff), "s"(ubase), "s"(lds_addr) : "memory");
}

template <int VAR>
__global__ __launch_bounds__(512, 2) void probe(const u32x4* __restrict__ seed, float* __restrict__ out, int iters, const char* __restrict__ src, unsigned span) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < LDSB / 16; i += 512) ((u32x4*)lds)[i] = seed[(blockIdx.x * 131 + i) & 65535];
    __syncthreads();
    constexpr bool WIDE = (VAR == 2 || VAR == 3 || (VAR >= 7 && VAR <= 10));
    constexpr bool DMA = (VAR == 7 || VAR == 8 || VAR == 9 || VAR >= 11);
    constexpr bool DMAONLY = (VAR >= 11);
    constexpr int NQ = (VAR == 12 || VAR == 13) ? 8 : 4;
    // DMA variants: the block's LDS image is 64 KiB of fragments + a 64 KiB landing zone behind it (never read: the data the MFMAs see stays the seed's)
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)lds);
    const size_t sb_ = (size_t)src + (size_t)((VAR == 7 || VAR == 11 || VAR == 12) ? 0 : blockIdx.x) * span;
    const unsigned sb_lo = (unsigned)__builtin_amdgcn_readfirstlane((unsigned)sb_), sb_hi = (unsigned)__builtin_amdgcn_readfirstlane((unsigned)(sb_ >> 32));
    const char* sbase = (const char*)(((size_t)sb_hi << 32) | (size_t)sb_lo);
    unsigned doff = (unsigned)(wave * 4096 + lane * 16);
    constexpr bool M32 = (VAR == 1 || VAR == 3 || VAR == 5);
    constexpr int NA = WIDE ? 8 : 4, NB = 4;                       // 16-row fragments per k32 (A rows / B columns of the wave tile)
    f32x4 acc16[M32 ? 1 : NA * NB];
    f32x16 acc32[M32 ? (NA / 2) * (NB / 2) : 1];
#pragma unroll
    for (int i = 0; i < (M32 ? 1 : NA * NB); ++i) acc16[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < (M32 ? (NA / 2) * (NB / 2) : 1); ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc32[i][r] = 0.f;
    bf16x8 fa[NA], fb[NB];
    // fixed operands for the no-read variants (and the initial value of the others)
#pragma unroll
    for (int x = 0; x < NA; ++x) fa[x] = *(const bf16x8*)(lds + ((wave * 4096 + x * 1024 + lane * 16) & (LDSB - 1)));
#pragma unroll
    for (int x = 0; x < NB; ++x) fb[x] = *(const bf16x8*)(lds + ((wave * 4096 + 32768 + x * 1024 + lane * 16) & (LDSB - 1)));
    float sink = 0.f;
    for (int it = 0; it < iters; ++it) {
        if constexpr (VAR != 4 && VAR != 5 && !DMAONLY) {
            const int base = ((it * 7 + wave * 5) & 15) * 4096;    // a different window of the block's LDS image every k-step
#pragma unroll
            for (int x = 0; x < NA; ++x) fa[x] = *(const bf16x8*)(lds + ((base + x * 1024 + lane * 16) & (LDSB - 1)));
#pragma unroll
            for (int x = 0; x < NB; ++x) fb[x] = *(const bf16x8*)(lds + ((base + 8192 + x * 1024 + lane * 16) & (LDSB - 1)));
        }
        if constexpr (DMA) {
            if constexpr (NQ == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");  // the previous k-step's pieces may still fly
#pragma unroll
            for (int q = 0; q < NQ; ++q) dma16(sbase, doff + q * 1024 + (q >= 4 ? 32768 - 4096 : 0), lds0 + LDSB + wave * 8192 + (NQ == 8 ? q : (it & 1) * 4 + q) * 1024);
            doff += NQ * 8192;                                     // the block walks its span 32 (64) KiB per k-step (8 waves x 4 (8) KiB)
            if (doff >= span) doff -= span;
        }
        if constexpr (VAR == 10) __builtin_amdgcn_s_barrier();
        if constexpr (DMAONLY) {
        } else if constexpr (VAR == 6) {
#pragma unroll
            for (int x = 0; x < NA; ++x) sink += (float)fa[x][0];
#pragma unroll
            for (int x = 0; x < NB; ++x) sink += (float)fb[x][0];
        } else if constexpr (!M32) {
#pragma unroll
            for (int b = 0; b < NB; ++b)
#pragma unroll
                for (int a = 0; a < NA; ++a) acc16[b * NA + a] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[b], fa[a], acc16[b * NA + a], 0, 0, 0);
        } else {
            // 32x32x16: a k32 step = two k16 instructions per 32x32 output block; operands = pairs of the 16-row fragments' registers
            // (which 8 elements of which row a lane feeds does not matter for the rate or the energy: the data is random either way)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int b = 0; b < NB / 2; ++b)
#pragma unroll
                    for (int a = 0; a < NA / 2; ++a)
                        acc32[b * (NA / 2) + a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[2 * b + kk], fa[2 * a + kk], acc32[b * (NA / 2) + a], 0, 0, 0);
        }
    }
#pragma unroll
    for (int i = 0; i < (M32 ? 1 : NA * NB); ++i) sink += acc16[i][0] + acc16[i][3];
#pragma unroll
    for (int i = 0; i < (M32 ? (NA / 2) * (NB / 2) : 1); ++i) sink += acc32[i][0] + acc32[i][15];
    if constexpr (DMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (sink == 123.456f) out[blockIdx.x * 512 + tid] = sink;      // keeps everything live, never true on random data in practice
}

static double now() { timespec t; clock_gettime(CLOCK_REALTIME, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }

template <int VAR>
static void run(const char* name, double flop_per_iter_per_wave, const u32x4* seed, float* out, double seconds, const char* src = nullptr, unsigned span = 0) {
    const int LDSB = (VAR == 7 || VAR == 8 || VAR == 9 || VAR >= 11) ? 2 * ::LDSB : ::LDSB;
    const int iters = 20000;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipFuncSetAttribute((const void*)probe<VAR>, hipFuncAttributeMaxDynamicSharedMemorySize, LDSB));
    hipLaunchKernelGGL(probe<VAR>, dim3(256), dim3(512), LDSB, 0, seed, out, iters, src, span);      // warm-up, and the duration of one launch
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(probe<VAR>, dim3(256), dim3(512), LDSB, 0, seed, out, iters, src, span);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms1 = 0; CK(hipEventElapsedTime(&ms1, e0, e1));
    const int launches = (int)(seconds * 1e3 / ms1) + 1;
    const double t0 = now();
    CK(hipEventRecord(e0));
    for (int i = 0; i < launches; ++i) hipLaunchKernelGGL(probe<VAR>, dim3(256), dim3(512), LDSB, 0, seed, out, iters, src, span);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    const double t1 = now();
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    // the last quarter alone (the clock has settled at the cap by then)
    const int tail = launches / 4 + 1;
    CK(hipEventRecord(e0));
    for (int i = 0; i < tail; ++i) hipLaunchKernelGGL(probe<VAR>, dim3(256), dim3(512), LDSB, 0, seed, out, iters, src, span);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    const double t2 = now();
    float mst = 0; CK(hipEventElapsedTime(&mst, e0, e1));
    const double flops = flop_per_iter_per_wave * iters * 8 * 256;
    printf("%-12s first launch %7.3f ms = %7.1f TF/s | %4d launches %7.1f TF/s | settled (last %3d) %7.1f TF/s   wall %.3f .. %.3f .. %.3f\n", name, ms1,
           flops / ms1 / 1e9, launches, flops * launches / ms / 1e9, tail, flops * tail / mst / 1e9, t0, t1, t2);
    fflush(stdout);
}

int main(int argc, char** argv) {
    const double seconds = argc > 1 ? atof(argv[1]) : 2.0;
    std::vector<unsigned> h(65536 * 4);
    unsigned s = 12345u;
    for (auto& v : h) {                                           // random bf16 pairs with exponents around 1 (no NaN / inf / denormals)
        s = s * 1664525u + 1013904223u;
        const unsigned lo = (s >> 8) & 0x807fu, hi = (s >> 20) & 0x807fu;
        v = (lo | 0x3f00u) | ((hi | 0x3f00u) << 16);
    }
    u32x4* seed; float* out;
    CK(hipMalloc(&seed, h.size() * 4)); CK(hipMalloc(&out, 256 * 512 * 4));
    CK(hipMemcpy(seed, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    const double f64 = 2.0 * 64 * 64 * 32, f128 = 2.0 * 128 * 64 * 32;
    run<0>("m16_64x64", f64, seed, out, seconds);
    run<1>("m32_64x64", f64, seed, out, seconds);
    run<2>("m16_128x64", f128, seed, out, seconds);
    run<3>("m32_128x64", f128, seed, out, seconds);
    run<4>("m16_noread", f64, seed, out, seconds);
    run<5>("m32_noread", f64, seed, out, seconds);
    run<6>("reads_only", f64, seed, out, seconds);
    run<10>("barrier", f128, seed, out, seconds);
    // DMA sources: random bf16 like the seed (1.5 GiB = 256 blocks x 6 MiB), filled by copies of the 1 MiB seed
    char* src; const size_t total = (size_t)256 * 6 * 1024 * 1024;
    CK(hipMalloc(&src, total + (1u << 20)));                      // + slack: a block reads up to 3 KiB past its span
    for (size_t o = 0; o < total + (1u << 20); o += h.size() * 4) CK(hipMemcpyAsync(src + o, seed, h.size() * 4, hipMemcpyDeviceToDevice, 0));
    CK(hipDeviceSynchronize());
    run<7>("dma_l2", f128, seed, out, seconds, src, 2u << 20);
    run<8>("dma_mall", f128, seed, out, seconds, src, 512u << 10);
    run<9>("dma_hbm", f128, seed, out, seconds, src, 6u << 20);
    // the fill alone (no MFMAs, no fragment reads): "TF/s" here = what a 256x256 bf16 tile (128 FLOP per byte filled) could run at this fill rate;
    // bytes per second = TF/s / 128 (dmaonly_4) -- dmaonly_8 moves twice the bytes per iteration: TF/s / 64
    run<11>("dmaonly_4", f128, seed, out, seconds, src, 2u << 20);
    run<12>("dmaonly_8", f128, seed, out, seconds, src, 2u << 20);
    run<13>("dmaonly_8m", f128, seed, out, seconds, src, 512u << 10);
    return 0;
}
