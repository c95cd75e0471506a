// Probe of the gfx950 block-scaled matrix instruction the "f16m" numerics mode is built on (tools/, not product code):
//   (1) which (row, k) of A and (k, col) of B each byte of a lane's 32-byte operand of v_mfma_scale_f32_16x16x128_f8f6f4 is,
//   (2) which lane's scale byte applies to which (row / col, 32-deep k block), and that E8M0 code c means 2^(c - 127),
//   (3) what v_cvt_pk_fp8_f32 does on gfx950 (OCP e4m3, rounding, saturation),
//   (4) the issue rate of the f16 + scaled-e4m3 / e2m3 mix against three f16 products.
// build: hipcc --offload-arch=gfx950 -O3 tools/mxprobe.hip -o gpurun_out/mxprobe ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <vector>
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;

__global__ void mx_once(const i32x8* a, const i32x8* b, const int* sa, const int* sb, f32x4* c) {
    const int l = threadIdx.x;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a[l], b[l], acc, 0, 0, 0, sa[l], 0, sb[l]);
    c[l] = acc;
}
__global__ void mx_once_fp6(const i32x8* a, const i32x8* b, const int* sa, const int* sb, f32x4* c) {
    const int l = threadIdx.x;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a[l], b[l], acc, 2, 2, 0, sa[l], 0, sb[l]);
    c[l] = acc;
}
__global__ void cvt_probe(const float* f, int* o, int n) {
    const int i = threadIdx.x;
    if (i < n) o[i] = __builtin_amdgcn_cvt_pk_fp8_f32(f[i], 0.f, 0, false) & 0xff;
}
// rate: one wave per SIMD (256 threads), REP iterations of a fixed MFMA mix on 4 independent accumulators
template <int KIND>
__global__ __launch_bounds__(256) void rate(float* out, int rep, unsigned long long* cyc) {
    const int l = threadIdx.x;
    i32x8 a8, b8;
    f16x8 ah, bh;
    for (int i = 0; i < 8; ++i) { a8[i] = 0x38383838 + l * 0x01010101 * (i & 1); b8[i] = 0x38383838 ^ (l << 3); ah[i] = (_Float16)(0.01f * l + i); bh[i] = (_Float16)(0.5f - 0.02f * i); }
    f32x4 acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int s = 127;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < rep; ++r) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (KIND == 0) {            // three fp16 products per 32 k  x 2 k-steps = 6 MFMAs per 64 k
#pragma unroll
                for (int u = 0; u < 6; ++u) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, acc[i], 0, 0, 0);
            } else if (KIND == 1) {     // fp16 main product (2 k-steps) + one scaled e4m3 MFMA per 64 k
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, acc[i], 0, 0, 0);
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, acc[i], 0, 0, 0);
                acc[i] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a8, b8, acc[i], 0, 0, 0, s, 0, s);
            } else if (KIND == 2) {     // the same with e2m3 planes
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, acc[i], 0, 0, 0);
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, acc[i], 0, 0, 0);
                acc[i] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a8, b8, acc[i], 2, 2, 0, s, 0, s);
            } else if (KIND == 3) {     // scaled e4m3 alone
                acc[i] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a8, b8, acc[i], 0, 0, 0, s, 0, s);
            } else if (KIND == 4) {     // scaled e2m3 alone
                acc[i] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a8, b8, acc[i], 2, 2, 0, s, 0, s);
            } else {                    // one fp16 MFMA alone
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, acc[i], 0, 0, 0);
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s4 = 0.f;
    for (int i = 0; i < 4; ++i) s4 += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + l] = s4;
    if (l == 0) cyc[blockIdx.x] = t1 - t0;
}

// the GEMM loop's shape: 16 accumulators (a 64x64 wave tile); per 64-deep K tile 32 fp16 MFMAs (2 k-steps) then 16 scaled ones
template <int FMT>
__global__ __launch_bounds__(256, 2) void rate16(float* out, int rep, unsigned long long* cyc) {
    const int l = threadIdx.x;
    i32x8 a8, b8;
    f16x8 ah, bh;
    for (int i = 0; i < 8; ++i) { a8[i] = 0x38383838 + l * 0x01010101 * (i & 1); b8[i] = 0x38383838 ^ (l << 3); ah[i] = (_Float16)(0.01f * l + i); bh[i] = (_Float16)(0.5f - 0.02f * i); }
    f32x4 acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int s = 127;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < rep; ++r) {
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, acc[i], 0, 0, 0);
        if (FMT >= 0) {
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a8, b8, acc[i], FMT, FMT, 0, s, 0, s);
        }
        if (FMT == -2) {                 // f16x: two more fp16 products per k-step
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh, ah, acc[i], 0, 0, 0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s4 = 0.f;
    for (int i = 0; i < 16; ++i) s4 += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + l] = s4;
    if (l == 0) cyc[blockIdx.x] = t1 - t0;
}

static float e4m3_to_f(unsigned v) {
    const int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
    float x = e ? ldexpf(1.0f + m / 8.0f, e - 7) : ldexpf(m / 8.0f, -6);
    if (e == 15 && m == 7) x = NAN;
    return s ? -x : x;
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

int main() {
    i32x8 *da, *db; int *dsa, *dsb; f32x4* dc;
    CK(hipMalloc(&da, 64 * 32)); CK(hipMalloc(&db, 64 * 32)); CK(hipMalloc(&dsa, 256)); CK(hipMalloc(&dsb, 256)); CK(hipMalloc(&dc, 64 * 16));
    std::vector<unsigned char> ha(2048), hb(2048);
    std::vector<int> hsa(64, 127), hsb(64, 127);
    std::vector<float> hc(256);
    auto run = [&](bool fp6) {
        hipMemcpy(da, ha.data(), 2048, hipMemcpyHostToDevice); hipMemcpy(db, hb.data(), 2048, hipMemcpyHostToDevice);
        hipMemcpy(dsa, hsa.data(), 256, hipMemcpyHostToDevice); hipMemcpy(dsb, hsb.data(), 256, hipMemcpyHostToDevice);
        if (fp6) mx_once_fp6<<<1, 64>>>(da, db, dsa, dsb, dc); else mx_once<<<1, 64>>>(da, db, dsa, dsb, dc);
        hipMemcpy(hc.data(), dc, 1024, hipMemcpyDeviceToHost);
    };
    // C/D layout (guide): lane l, reg r -> row (l>>4)*4 + r, col l & 15
    auto C = [&](int row, int col) { return hc[((row >> 2) * 16 + col) * 4 + (row & 3)]; };
    // ---- (1) operand byte maps.  B = 1.0 everywhere except that each k-slot carries a distinct value is impossible in e4m3 (128 ks),
    // so: A one-hot 1.0 at (lane, byte); B[lane', byte'] = 1.0 only where the HYPOTHESIS says the same k lives: hypothesis
    // k = 32 * (lane >> 4) + byte for both operands, row = lane & 15 (A), col = lane & 15 (B).
    int bad = 0;
    for (int lane = 0; lane < 64; ++lane)
        for (int byte = 0; byte < 32; byte += 5) {
            std::fill(ha.begin(), ha.end(), 0); std::fill(hb.begin(), hb.end(), 0);
            ha[lane * 32 + byte] = 0x38;                                   // 1.0
            // B: every column, only the hypothesised k: lanes with the same (lane >> 4), the same byte -> value 2^(col & 3) + distinct per col
            for (int col = 0; col < 16; ++col) hb[((lane >> 4) * 16 + col) * 32 + byte] = 0x38 + 8 * (col & 7);   // 2^(col & 7)
            run(false);
            for (int row = 0; row < 16; ++row)
                for (int col = 0; col < 16; ++col) {
                    const float want = (row == (lane & 15)) ? ldexpf(1.0f, col & 7) : 0.f;
                    if (C(row, col) != want) { if (bad < 8) printf("map mismatch lane %d byte %d row %d col %d got %g want %g\n", lane, byte, row, col, C(row, col), want); ++bad; }
                }
        }
    printf("(1) operand map k = 32*(lane>>4) + byte, row/col = lane&15: %s (%d mismatches)\n", bad ? "WRONG" : "confirmed", bad);
    // negative control: a different k on the B side must give zero
    {
        std::fill(ha.begin(), ha.end(), 0); std::fill(hb.begin(), hb.end(), 0);
        ha[5 * 32 + 3] = 0x38;
        for (int col = 0; col < 16; ++col) hb[(0 * 16 + col) * 32 + 4] = 0x38;
        run(false);
        float s = 0; for (float v : hc) s += fabsf(v);
        printf("    control (k differs): sum |C| = %g (want 0)\n", s);
    }
    // ---- (2) scales: all ones in A and B; scale code of lane l: A 127 + ((l * 7) % 5) - 2, B 127 + ((l * 3) % 4) - 1
    {
        std::fill(ha.begin(), ha.end(), 0x38); std::fill(hb.begin(), hb.end(), 0x38);
        for (int l = 0; l < 64; ++l) { hsa[l] = 127 + ((l * 7) % 5) - 2; hsb[l] = 127 + ((l * 3) % 4) - 1; }
        run(false);
        int badS = 0;
        for (int row = 0; row < 16; ++row)
            for (int col = 0; col < 16; ++col) {
                float want = 0.f;
                for (int g = 0; g < 4; ++g) want += 32.0f * ldexpf(1.0f, hsa[g * 16 + row] - 127) * ldexpf(1.0f, hsb[g * 16 + col] - 127);
                if (C(row, col) != want) { if (badS < 8) printf("scale mismatch row %d col %d got %g want %g\n", row, col, C(row, col), want); ++badS; }
            }
        printf("(2) scale of lane l applies to (row/col l&15, k block l>>4), code c = 2^(c-127): %s (%d mismatches)\n", badS ? "WRONG" : "confirmed", badS);
        // upper bytes of the scale register must be ignored with op_sel 0
        for (int l = 0; l < 64; ++l) { hsa[l] |= 0x55aa1100 & 0xffffff00; }
        run(false);
        int badU = 0;
        for (int row = 0; row < 16; ++row)
            for (int col = 0; col < 16; ++col) {
                float want = 0.f;
                for (int g = 0; g < 4; ++g) want += 32.0f * ldexpf(1.0f, (hsa[g * 16 + row] & 0xff) - 127) * ldexpf(1.0f, hsb[g * 16 + col] - 127);
                if (C(row, col) != want) ++badU;
            }
        printf("    upper scale-register bytes ignored with op_sel 0: %s\n", badU ? "NO" : "yes");
        for (int l = 0; l < 64; ++l) { hsa[l] = 127; hsb[l] = 127; }
    }
    // ---- (2b) which lane's scale applies to byte j of lane group g: A one-hot 1.0 at (row 0, group g, byte j), B the same, A's scale of lane group b = 2^b
    {
        printf("(2b) scale owner (lane group whose scale register applies) of byte j, per lane group g of the A operand:\n");
        for (int g = 0; g < 4; ++g) {
            printf("     g=%d:", g);
            for (int j = 0; j < 32; ++j) {
                std::fill(ha.begin(), ha.end(), 0); std::fill(hb.begin(), hb.end(), 0);
                ha[(g * 16 + 0) * 32 + j] = 0x38; hb[(g * 16 + 0) * 32 + j] = 0x38;
                for (int l = 0; l < 64; ++l) { hsa[l] = 127 + (l >> 4); hsb[l] = 127; }
                run(false);
                printf(" %d", (int)lrintf(log2f(C(0, 0))));
            }
            printf("\n");
        }
        printf("     and of the B operand:\n");
        for (int g = 0; g < 4; ++g) {
            printf("     g=%d:", g);
            for (int j = 0; j < 32; ++j) {
                std::fill(ha.begin(), ha.end(), 0); std::fill(hb.begin(), hb.end(), 0);
                ha[(g * 16 + 0) * 32 + j] = 0x38; hb[(g * 16 + 0) * 32 + j] = 0x38;
                for (int l = 0; l < 64; ++l) { hsb[l] = 127 + (l >> 4); hsa[l] = 127; }
                run(false);
                printf(" %d", (int)lrintf(log2f(C(0, 0))));
            }
            printf("\n");
        }
        // does the ROW of the scale lane matter?  scale of lane (row r, group b) = 2^r, element in row 5
        std::fill(ha.begin(), ha.end(), 0); std::fill(hb.begin(), hb.end(), 0);
        ha[(1 * 16 + 5) * 32 + 7] = 0x38; hb[(1 * 16 + 3) * 32 + 7] = 0x38;
        for (int l = 0; l < 64; ++l) { hsa[l] = 127 + (l & 15); hsb[l] = 127; }
        run(false);
        printf("     scale row check: element in A row 5 x B col 3 -> C(5,3) = 2^%d (want 5)\n", (int)lrintf(log2f(C(5, 3))));
        for (int l = 0; l < 64; ++l) { hsa[l] = 127; hsb[l] = 127; }
    }
    // ---- random e4m3 data by class, fp64 check
    for (int cls = 0; cls < 6; ++cls) {
        srand(7 + cls);
        auto gen = [&](unsigned char& v) {
            unsigned e = 1 + rand() % 14, m = rand() & 7, sgn = 0;
            if (cls >= 1) sgn = rand() & 1;
            if (cls >= 2) e = 1 + rand() % 15;
            if (cls >= 3) e = rand() % 16;
            v = (unsigned char)((sgn << 7) | (e << 3) | m);
            if ((v & 0x7f) == 0x7f) v = 0;
        };
        for (auto& v : ha) gen(v);
        for (auto& v : hb) gen(v);
        for (int l = 0; l < 64; ++l) { hsa[l] = 127; hsb[l] = 127; }
        if (cls >= 4) for (int l = 0; l < 64; ++l) { hsa[l] = 120 + rand() % 12; hsb[l] = 122 + rand() % 9; }
        if (cls >= 5) for (int l = 0; l < 64; ++l) { hsa[l] = 100 + rand() % 12; hsb[l] = 90 + rand() % 9; }
        run(false);
        double worst = 0, big = 0; int nan = 0;
        for (int row = 0; row < 16; ++row)
            for (int col = 0; col < 16; ++col) {
                double want = 0;
                for (int g = 0; g < 4; ++g)
                    for (int j = 0; j < 32; ++j)
                        want += (double)e4m3_to_f(ha[(g * 16 + row) * 32 + j]) * e4m3_to_f(hb[(g * 16 + col) * 32 + j]) * ldexp(1.0, hsa[g * 16 + row] - 127) * ldexp(1.0, hsb[g * 16 + col] - 127);
                if (C(row, col) != C(row, col)) ++nan;
                if (row == 3 && col == 5) printf("    class %d sample: got %.9g want %.9g\n", cls, C(row, col), want);
                worst = fmax(worst, fabs(want - C(row, col))); big = fmax(big, fabs(want));
            }
        printf("    class %d (0 pos normals e<15, 1 +signs, 2 +e=15, 3 +subnormals, 4 +scales, 5 small scales) vs fp64: max |err| %.3g of max |C| %.3g, NaNs %d\n", cls, worst, big, nan);
        for (int l = 0; l < 64; ++l) { hsa[l] = 127; hsb[l] = 127; }
    }
    // ---- fp6 (e2m3) operand map: 32 x 6 bits packed little-endian in the first 24 bytes?  value 1.0 = 0b001000 = 8
    {
        int bad6 = 0;
        for (int lane = 0; lane < 64; lane += 7)
            for (int e = 0; e < 32; e += 3) {
                std::fill(ha.begin(), ha.end(), 0); std::fill(hb.begin(), hb.end(), 0);
                auto put = [](unsigned char* p, int idx, unsigned v) { const int bit = idx * 6; for (int b = 0; b < 6; ++b) if (v >> b & 1) p[(bit + b) >> 3] |= 1u << ((bit + b) & 7); };
                put(&ha[lane * 32], e, 8);
                for (int col = 0; col < 16; ++col) put(&hb[((lane >> 4) * 16 + col) * 32], e, 8 + 2 * (col & 3));     // 1.0, 1.25, 1.5, 1.75
                run(true);
                for (int row = 0; row < 16; ++row)
                    for (int col = 0; col < 16; ++col) {
                        const float want = (row == (lane & 15)) ? 1.0f + 0.25f * (col & 3) : 0.f;
                        if (C(row, col) != want) { if (bad6 < 6) printf("fp6 mismatch lane %d elem %d row %d col %d got %g want %g\n", lane, e, row, col, C(row, col), want); ++bad6; }
                    }
            }
        printf("(1b) fp6: element e of a lane = bits [6e, 6e+6) of its first 24 bytes, k = 32*(lane>>4) + e: %s (%d)\n", bad6 ? "WRONG" : "confirmed", bad6);
    }
    // ---- (3) v_cvt_pk_fp8_f32
    {
        const float xs[] = {0.f, 1.f, 1.0625f, 1.1875f, 1.0624f, 1.0626f, 440.f, 448.f, 449.f, 464.f, 480.f, 500.f, 1e9f, -1e9f, INFINITY, 0.015625f, 0.0078125f,
                            0.001953125f, 0.0009765625f, 0.00097f, 0.0029296875f, 0.0048828125f, -3.3f, 17.f, 18.f, 19.f, 25.f, 27.f};
        const int n = sizeof(xs) / sizeof(xs[0]);
        float* df; int* dout; hipMalloc(&df, 4 * 64); hipMalloc(&dout, 4 * 64);
        hipMemcpy(df, xs, 4 * n, hipMemcpyHostToDevice);
        cvt_probe<<<1, 64>>>(df, dout, n);
        int ho[64]; hipMemcpy(ho, dout, 4 * 64, hipMemcpyDeviceToHost);
        printf("(3) v_cvt_pk_fp8_f32:");
        for (int i = 0; i < n; ++i) printf(" %g->0x%02x(%g)", xs[i], ho[i], e4m3_to_f(ho[i]));
        printf("\n");
    }
    // ---- (4) rates
    {
        float* dout; unsigned long long* dcyc; hipMalloc(&dout, 256 * 256 * 4); hipMalloc(&dcyc, 256 * 8);
        const int rep = 2000;
        const char* names[] = {"6 x f16 16x16x32 (f16x per 64 k)", "2 x f16 + 1 x scaled e4m3 (f16m8 per 64 k)", "2 x f16 + 1 x scaled e2m3 (f16m6 per 64 k)", "scaled e4m3 alone", "scaled e2m3 alone", "f16 16x16x32 alone"};
        const int per[] = {24, 12, 12, 4, 4, 4};
        for (int kind = 0; kind < 6; ++kind) {
            for (int it = 0; it < 2; ++it) {
                switch (kind) {
                    case 0: rate<0><<<256, 256>>>(dout, rep, dcyc); break;
                    case 1: rate<1><<<256, 256>>>(dout, rep, dcyc); break;
                    case 2: rate<2><<<256, 256>>>(dout, rep, dcyc); break;
                    case 3: rate<3><<<256, 256>>>(dout, rep, dcyc); break;
                    case 4: rate<4><<<256, 256>>>(dout, rep, dcyc); break;
                    default: rate<5><<<256, 256>>>(dout, rep, dcyc); break;
                }
                hipDeviceSynchronize();
            }
            unsigned long long hcyc[256]; hipMemcpy(hcyc, dcyc, 256 * 8, hipMemcpyDeviceToHost);
            double s = 0; for (int i = 0; i < 256; ++i) s += (double)hcyc[i];
            s /= 256;
            printf("(4) %-46s %.1f cycles per iteration (4 accumulators x the mix), %.1f per MFMA\n", names[kind], s / rep, s / rep / per[kind]);
        }
    }
    {
        float* dout; unsigned long long* dcyc; hipMalloc(&dout, 256 * 256 * 4); hipMalloc(&dcyc, 256 * 8);
        const int rep = 1000;
        const char* names[] = {"32 f16 + 16 scaled e4m3 (f16m8)", "32 f16 + 16 scaled e2m3 (f16m6)", "32 f16 only (single product)", "96 f16 (f16x)"};
        for (int kind = 0; kind < 4; ++kind) {
            for (int it = 0; it < 2; ++it) {
                switch (kind) {
                    case 0: rate16<0><<<256, 256>>>(dout, rep, dcyc); break;
                    case 1: rate16<2><<<256, 256>>>(dout, rep, dcyc); break;
                    case 2: rate16<-1><<<256, 256>>>(dout, rep, dcyc); break;
                    default: rate16<-2><<<256, 256>>>(dout, rep, dcyc); break;
                }
                hipDeviceSynchronize();
            }
            unsigned long long hcyc[256]; hipMemcpy(hcyc, dcyc, 256 * 8, hipMemcpyDeviceToHost);
            double s = 0; for (int i = 0; i < 256; ++i) s += (double)hcyc[i];
            s /= 256;
            printf("(5) 64x64 wave tile, per 64-deep K tile: %-36s %.1f cycles\n", names[kind], s / rep);
        }
    }
    return 0;
}
