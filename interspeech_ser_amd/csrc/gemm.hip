// Matrix-core GEMM with implicit-convolution row map and fused epilogue (ser_gemm).
//
//   C[m,n] = sum_k A(m,k) * W[n,k]     A: act bf16 (1 or 2 planes), W: bf16 [N][K] (1 or 2 planes)
//
// gfx950 design
//   * 128x128x64 block tile, 256 threads = 4 waves (2x2), 64x64 per wave,
//     v_mfma_f32_16x16x32_bf16 with the WEIGHT tile as the MFMA A operand and the
//     activation tile as the B operand: the accumulator then holds 4 consecutive output
//     columns per register quad, and by permuting which weight row sits in which LDS row
//     each lane ends up owning 16 CONSECUTIVE output columns of one row -> 16-byte
//     epilogue loads/stores for bias, residual, fp32 and bf16 outputs.
//   * global -> LDS by global_load_lds_dwordx4 (16 B/lane, no VGPR round trip).  LDS image is
//     [row][64 bf16] (128-B rows) with 16-B chunk index XOR (row & 7): the DMA destination
//     stays lane-linear and the swizzle is applied to the per-lane SOURCE address and to
//     the ds_read_b128 address (both sides or neither).  Verified conflict-free for the
//     four 16-lane groups of ds_read_b128.
//   * 2-stage LDS ring (64 KiB -> 2 blocks / CU): loads of tile t+1 are issued before the
//     MFMAs of tile t, one vmcnt(0)+barrier per K tile.
//   * FP32X mode runs the same loop over 3 K-segments (hi*hi, lo*hi, hi*lo) into one
//     accumulator: fp32-grade products on the bf16 pipe.
//   * blockIdx.x -> tile through a bijective XCD swizzle so the 8 L2s each see a
//     contiguous run of tiles that share activation panels.
#include "ser_common.h"

#define GBM 128
#define GBN 128
#define GBK 64
#define GSTAGE 32768

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

template <int MODE>
__global__ __launch_bounds__(256, 2) void ser_gemm_kernel(const ser_gemm_args p) {
    __shared__ __attribute__((aligned(16))) char lds[2 * GSTAGE];
    constexpr int NSEG = (MODE == SER_MODE_FP32X) ? 3 : 1;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;

    const int ntn = (p.N + GBN - 1) / GBN;
    const int ntm = (p.M + GBM - 1) / GBM;
    int bid = blockIdx.x;
    {   // bijective XCD remap: blocks with equal (bid & 7) share an L2
        const int nwg = ntm * ntn;
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
        const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
        bid = base + (bid >> 3);
    }
    const int mt = bid / ntn, nt = bid - mt * ntn;
    const int g = blockIdx.y;
    const int m0 = mt * GBM, n0 = nt * GBN;

    const unsigned short* Abase = (const unsigned short*)p.A + (int64_t)g * p.a_group_stride;
    const unsigned short* Wbase = (const unsigned short*)p.W + (int64_t)g * p.w_group_stride;

    // ---- per-lane DMA source rows (4 A rows + 4 W rows per k-tile) -------------------
    const unsigned short* aptr[4];
    const unsigned short* wptr[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int R = wave * 32 + q * 8 + (lane >> 3);           // LDS row this lane fills
        const int c = (lane & 7) ^ (R & 7);                       // logical 16-B chunk it must fetch
        int m = m0 + R;
        m = m < p.M ? m : p.M - 1;
        const int64_t arow = p.a_rowoff ? (int64_t)p.a_rowoff[m] * 8 : (int64_t)m * p.lda;
        aptr[q] = Abase + arow + c * 8;
        // weight row permutation: LDS row (ni*16 + i) of a wave column holds n = (i>>2)*16 + ni*4 + (i&3)
        const int i = R & 15, ni = (R >> 4) & 3;
        int n = n0 + (R & 64) + (i >> 2) * 16 + ni * 4 + (i & 3);
        n = n < p.N ? n : p.N - 1;
        wptr[q] = Wbase + (int64_t)n * p.K + c * 8;
    }

    const int nk = p.K / GBK;
    const int total = nk * NSEG;
    const int tpc = p.kc ? p.kc / GBK : 0x7fffffff;              // k-tiles per conv chunk

    // running scalar state of the *issue* side
    int i_kk = 0, i_cc = 0, i_cj = 0, i_seg = 0;
    auto issue = [&](int stage) {
        const int64_t koffA = (int64_t)i_cj * p.ldj + (int64_t)i_cc * GBK
                            + ((NSEG == 3 && i_seg == 1) ? p.a_plane_stride : 0);
        const int64_t koffW = (int64_t)i_kk * GBK + ((NSEG == 3 && i_seg == 2) ? p.w_plane_stride : 0);
        char* dstA = lds + stage * GSTAGE + wave * 4096;
        char* dstW = dstA + 16384;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            __builtin_amdgcn_global_load_lds((gptr_t)(aptr[q] + koffA), (lptr_t)(dstA + q * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((gptr_t)(wptr[q] + koffW), (lptr_t)(dstW + q * 1024), 16, 0, 0);
        }
        ++i_kk; ++i_cc;
        if (i_cc == tpc) { i_cc = 0; ++i_cj; }
        if (i_kk == nk) { i_kk = 0; i_cc = 0; i_cj = 0; ++i_seg; }
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // per-lane fragment read offsets (row & 7 == lane & 7 for every fragment row)
    const int frow = lane & 15, fq = lane >> 4;
    int offA[2], offW[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const int phys = ((s * 4 + fq) ^ (lane & 7)) << 4;
        offA[s] = (wm * 64 + frow) * 128 + phys;
        offW[s] = 16384 + (wn * 64 + frow) * 128 + phys;
    }

    issue(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    for (int kt = 0; kt < total; ++kt) {
        const int stage = kt & 1;
        if (kt + 1 < total) issue(stage ^ 1);
        const char* sb = lds + stage * GSTAGE;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            bf16x8 af[4], wf[4];
#pragma unroll
            for (int x = 0; x < 4; ++x) {
                af[x] = *(const bf16x8*)(sb + offA[s] + x * 2048);
                wf[x] = *(const bf16x8*)(sb + offW[s] + x * 2048);
            }
#pragma unroll
            for (int ni = 0; ni < 4; ++ni)
#pragma unroll
                for (int mi = 0; mi < 4; ++mi)
                    acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ni], af[mi], acc[ni][mi], 0, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }

    // ---- epilogue: lane owns row m, 16 consecutive columns ---------------------------
    const int ncol0 = n0 + wn * 64 + fq * 16;                     // within the group
    const int64_t gcol = (int64_t)g * p.c_group_stride + ncol0;   // in the output matrices
    float bias[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) bias[j] = 0.f;
    if (p.bias) {
#pragma unroll
        for (int j4 = 0; j4 < 4; ++j4)
            if (ncol0 + j4 * 4 < p.N) {
                const f32x4 b = *(const f32x4*)(p.bias + (int64_t)g * p.N + ncol0 + j4 * 4);
                bias[j4 * 4 + 0] = b[0]; bias[j4 * 4 + 1] = b[1]; bias[j4 * 4 + 2] = b[2]; bias[j4 * 4 + 3] = b[3];
            }
    }
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
        const int m = m0 + wm * 64 + mi * 16 + frow;
        if (m >= p.M) continue;
        const int rrow = p.res_row_mod ? (m % p.res_row_mod) : m;
        const int64_t orow = p.out_rowmap ? (int64_t)p.out_rowmap[m] : (int64_t)m;
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
            if (ncol0 + ni * 4 >= p.N) continue;
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float x = acc[ni][mi][r] + bias[ni * 4 + r];
                if (p.act == SER_ACT_GELU) x = gelu_erf(x);
                v[r] = x;
            }
            if (p.residual) {
                const f32x4 rr = *(const f32x4*)(p.residual + (int64_t)rrow * p.ldr + gcol + ni * 4);
                v[0] += rr[0]; v[1] += rr[1]; v[2] += rr[2]; v[3] += rr[3];
            }
            if (p.out_f32) {
                f32x4 o = {v[0], v[1], v[2], v[3]};
                *(f32x4*)(p.out_f32 + (int64_t)m * p.ldo_f32 + gcol + ni * 4) = o;
            }
            if (p.out_act) {
                store_act4<MODE>((unsigned short*)p.out_act + orow * p.ldo_act + gcol + ni * 4,
                                 p.out_plane_stride, v[0], v[1], v[2], v[3]);
            }
        }
    }
}

extern "C" int ser_gemm(const ser_gemm_args* a, void* stream) {
    if (!a || !a->A || !a->W) return ser_fail(-1, "ser_gemm: null operand");
    if (a->M <= 0 || a->N <= 0 || a->K <= 0) return ser_fail(-2, "ser_gemm: bad shape M=%d N=%d K=%d", a->M, a->N, a->K);
    if (a->K % GBK) return ser_fail(-3, "ser_gemm: K=%d must be a multiple of %d", a->K, GBK);
    if (a->kc && (a->kc % GBK || a->K % a->kc)) return ser_fail(-4, "ser_gemm: kc=%d must divide K and be a multiple of %d", a->kc, GBK);
    if (a->N % 8) return ser_fail(-5, "ser_gemm: N=%d must be a multiple of 8", a->N);
    if (a->mode != SER_MODE_BF16 && a->mode != SER_MODE_FP32X) return ser_fail(-6, "ser_gemm: bad mode %d", a->mode);
    if (!a->a_rowoff && (a->lda % 8)) return ser_fail(-7, "ser_gemm: lda must be a multiple of 8");
    if (a->groups < 1) return ser_fail(-8, "ser_gemm: groups=%d", a->groups);
    if (!a->out_f32 && !a->out_act) return ser_fail(-9, "ser_gemm: no output");
    if ((a->ldo_f32 % 4) || (a->ldo_act % 4) || (a->ldr % 4) || (a->c_group_stride % 4))
        return ser_fail(-10, "ser_gemm: output/residual pitches must be multiples of 4");
    const int ntm = (a->M + GBM - 1) / GBM, ntn = (a->N + GBN - 1) / GBN;
    dim3 grid((unsigned)(ntm * ntn), (unsigned)a->groups, 1), block(256, 1, 1);
    if (a->mode == SER_MODE_BF16)
        hipLaunchKernelGGL(ser_gemm_kernel<SER_MODE_BF16>, grid, block, 0, (hipStream_t)stream, *a);
    else
        hipLaunchKernelGGL(ser_gemm_kernel<SER_MODE_FP32X>, grid, block, 0, (hipStream_t)stream, *a);
    return ser_check_launch("ser_gemm");
}
