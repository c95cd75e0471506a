// Matrix-core GEMM with implicit-convolution row map and fused epilogues (ser_gemm).
//
//   C[m,n] = sum_k A(m,k) * W[n,k]     A: act bf16 (1 or 2 planes), W: bf16 [N][K] (1 or 2 planes)
//
// gfx950 design
//   * One kernel template, several tile configurations (host picks per shape):
//       128x128x64, 4 waves, 2-stage ring, 2 blocks/CU   -- small grids (N = 1024 projections)
//       256x128x64, 8 waves, 3-stage ring                -- large grids
//       256x256x64, 8 waves, 2-stage ring                -- largest grids (half the L2->LDS bytes per FLOP)
//       128x512x32, 8 waves, 3-stage ring, LayerNorm+GELU epilogue over the full 512-wide row
//   * v_mfma_f32_16x16x32_bf16 with the WEIGHT tile as the MFMA A operand and the activation
//     tile as the B operand: the accumulator then holds 4 consecutive output columns per register
//     quad, and by permuting which weight row sits in which LDS row each lane ends up owning
//     TN*4 CONSECUTIVE output columns of one row -> 16-byte epilogue loads/stores.
//   * global -> LDS by global_load_lds_dwordx4 (16 B/lane, no VGPR round trip).  The LDS image is
//     [row][BK bf16] with the 16-B chunk index XOR f(row): the DMA destination stays lane-linear,
//     the swizzle is applied to the per-lane SOURCE address and to the ds_read_b128 address (both
//     sides or neither).  f = row & 7 (BK=64) / (row >> 1) & 3 (BK=32): conflict-free for all four
//     16-lane groups of ds_read_b128 (checked by enumeration).
//   * S-stage LDS ring with COUNTED s_waitcnt vmcnt(N) and a raw s_barrier: loads of the next S-1
//     K tiles stay in flight across the barrier (a __syncthreads() would drain them), one barrier
//     per K tile, s_setprio around the MFMA cluster.
//   * FP32X mode runs the same loop over 3 K-segments (hi*hi, lo*hi, hi*lo) into one accumulator.
//   * FP16M mode (round 5): the ring carries TWO units per 64-deep K tile -- the fp16 hi planes (two f16 MFMA k-steps) and the e4m3 cross-term
//     planes + their block scales (ONE v_mfma_scale_f32_16x16x128_f8f6f4 per fragment pair): 2 product-equivalents instead of FP16X's 3.
//   * blockIdx.x -> tile through a bijective XCD swizzle so each of the 8 L2s sees a contiguous run
//     of tiles that share activation panels.
#include "ser_common.h"
#ifndef SER_GEMM_PP
#define SER_GEMM_PP 27        // ping-pong schedule, bit mask: 1 = 256x256, 2 = 256x128 / 64x512 (64x64 wave tiles), 4 = 128x512 LayerNorm tile, 8 = the FP32X 128x128 tile, 16 = the FP32X 128x512 LayerNorm tile; 0 = plain ring (A/B builds)
#endif
#ifndef SER_GEMM_PSW
#define SER_GEMM_PSW 0        // the hand-placed software pipeline instead of the ping-pong phases on the 8-wave single-plane tiles: 1 = 256x256, 2 = 256x128 (two-stage ring)
#endif
#ifndef SER_M16_READS_FIRST
#define SER_M16_READS_FIRST 1  // FP16M ping-pong loop: fragment reads of a unit are issued before its DMAs (0: DMAs first, like the other tiles)
#endif
#ifndef SER_M16_PRIO
#define SER_M16_PRIO 1        // FP16M ping-pong loop: 1 = s_setprio 1 around the MFMA phase (as the other tiles), 0 = none, 2 = around the read / DMA-issue phase
#endif
#ifndef SER_WHATIF
#define SER_WHATIF 0          // diagnostic builds only (tools/ab): 1 = no MFMAs, 2 = no DMA, 4 = no fragment reads after the first K tile of the FP16M loop
#endif
#ifndef SER_GEMM_SUPER
#define SER_GEMM_SUPER 0      // 1 = 4 x 8 super-tile order of the blocks inside an XCD's run (see the kernel): FC1 alone 2 - 4 % faster, the step unchanged (profiles/r05_gemm_super_tile_ab.txt)
#endif
#include <stdlib.h>
#include <stdio.h>
#include <atomic>
#include <type_traits>
#ifdef SER_GEMM_DBG
// diagnostic build only (tools/gemm_clock.py): wave 0 of every block stamps s_memtime / s_memrealtime around its K loop
extern "C" { void* ser_gemm_dbg_ptr = nullptr; }
__device__ unsigned long long* ser_gemm_dbg_dev = nullptr;
#endif

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// One LDS-DMA wave instruction in its scalar-base form: 64 lanes x BYTES from (uniform base + the lane's 32-bit offset) to LDS bytes
// [lds_addr, lds_addr + 64 * BYTES), lane-linear.  M0 carries the LDS address; nothing else in this library uses M0.
// a pointer the compiler may hold in VGPRs although every lane has the same value -> SGPR pair (the "s" asm constraint is not enforced
// for 64-bit operands: without this the -DSER_GEMM_DBG build handed the instruction a VGPR pair)
__device__ __forceinline__ const char* uniform_ptr(const void* p) {
    const uint64_t v = (uint64_t)p;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return (const char*)(((uint64_t)hi << 32) | lo);
}
template <int BYTES>
__device__ __forceinline__ void dma_lds(const void* ubase, uint32_t voff, uint32_t lds_addr) {
    static_assert(BYTES == 16 || BYTES == 4, "dwordx4 or dword");
    if constexpr (BYTES == 16)
        asm volatile("s_mov_b32 m0, %2\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(voff), "s"(ubase), "s"(lds_addr) : "memory");
    else
        asm volatile("s_mov_b32 m0, %2\n\tglobal_load_lds_dword %0, %1" :: "v"(voff), "s"(ubase), "s"(lds_addr) : "memory");
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if constexpr (N == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
    else if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else if constexpr (N == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if constexpr (N == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
    else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else if constexpr (N == 7) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
    else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if constexpr (N == 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
    else if constexpr (N == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else if constexpr (N == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else static_assert(N < 0, "add a case");
}

// acc += (block-scaled e4m3 A) x (block-scaled e4m3 B), 16 x 16 x 128.  Issued from inline asm with the accumulator TIED: hipcc (ROCm 7.2) gives the
// builtin's result a fresh register quad (its destination is early-clobber, never coalesced with the C input), so the 128 accumulators of the
// 256x256 tile rotate through the file and 50 - 120 registers spill, some inside the K loop.  Hazards the compiler no longer sees: every call site
// sits behind an s_barrier + LDS waits after the last writer of its operands, consecutive calls target different accumulators, and the K loop ends
// in mfma_scale_drain() before anything else reads the accumulators.
#ifndef SER_MX_ASM
#define SER_MX_ASM 1
#endif
#ifndef SER_MX_FP6_WHATIF
#define SER_MX_FP6_WHATIF 0       // 1: the cross-term MFMAs read their operands as e2m3 (timing what-if, WRONG results: profiles/r05_f16m_e2m3_whatif.txt)
#endif
typedef __attribute__((ext_vector_type(6))) int i32x6;
__device__ __forceinline__ void mfma_scale8(f32x4& acc, const i32x8& a, const i32x8& b, int sa, int sb) {
#if SER_MX_FP6_WHATIF
    const i32x6 a6 = {a[0], a[1], a[2], a[3], a[4], a[5]}, b6 = {b[0], b[1], b[2], b[3], b[4], b[5]};
    asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %4 op_sel_hi:[0,0,0] cbsz:2 blgp:2" : "+v"(acc) : "v"(a6), "v"(b6), "v"(sa), "v"(sb));
#elif SER_MX_ASM
    asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %4 op_sel_hi:[0,0,0]" : "+v"(acc) : "v"(a), "v"(b), "v"(sa), "v"(sb));
#else
    acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, acc, 0, 0, 0, sa, 0, sb);
#endif
}
__device__ __forceinline__ void mfma_scale_drain() {
#if SER_MX_ASM
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 7" ::: "memory");          // > the 8 passes of the last scaled MFMA
#endif
}

// WM x WN waves, each TM x TN MFMA tiles of 16x16; BK = K tile; ST = ring stages.
// OM: format of the out_act copy (== MODE unless the launch converts, e.g. an FP32X stem GEMM feeding FP16 layers).
// (Round 3's persistent form -- only the resident blocks are launched and each walks several tiles -- measured -8 % on the step and was
// removed from the kernel in round 4; DESIGN.md section 10 keeps the record.)
template <int WM, int WN, int TM, int TN, int BK, int ST, int MODE, bool LNEPI, int OM = MODE>
__global__ __launch_bounds__(64 * WM * WN, (WM * WN == 4 && TM * TN == 64) ? 1 : 2)
void ser_gemm_kernel(const ser_gemm_args p) {
    constexpr int NW = WM * WN, NT = 64 * NW;
    constexpr int BM = WM * TM * 16, BN = WN * TN * 16;
    constexpr int CH = BK / 8;                    // 16-byte chunks per LDS row
    constexpr int ROWB = BK * 2;                  // bytes per LDS row
    constexpr int RPP = 1024 / ROWB;              // rows per 1-KiB DMA piece (one wave instruction)
    // FP32X: both planes (hi, lo) of the A and W tiles of a K tile sit in one stage, fetched ONCE, and every
    // fragment pair feeds 3 MFMAs (hi*hi, lo*hi, hi*lo): 2x the L2->LDS bytes of bf16 for 3x the products,
    // instead of three full passes over K
    // FP16M: a stage holds ONE unit of a K tile -- its fp16 hi planes (even units) or its e4m3 cross-term planes (odd units), both
    // [rows][128 bytes] images with the same swizzle -- so the ring is the single-plane one walked twice per K tile
    constexpr bool M16 = (MODE == SER_MODE_FP16M);
    constexpr int NPL = M16 ? 1 : mode_traits<MODE>::planes;     // FP32X (bf16 hi/lo) and FP16X (fp16 hi/lo)
    constexpr int A_BYTES = BM * ROWB, W_BYTES = BN * ROWB, STAGE = NPL * (A_BYTES + W_BYTES);
    static_assert(!M16 || (BK == 64 && !LNEPI && (BM + BN) % 64 == 0), "FP16M: 64-deep K tiles, dense epilogue");
    constexpr int SCW = BM + BN;                  // FP16M: scale words per unit (one per A row and W row of the tile)
    constexpr int LA = BM / RPP / NW, LW = BN / RPP / NW;       // DMA pieces per wave per K tile
    constexpr int LPT = NPL * (LA + LW);
    constexpr int KS = BK / 32;                   // MFMA k-steps per K tile
    static_assert(BM % (RPP * NW) == 0 && BN % (RPP * NW) == 0, "tile/wave mismatch");
    extern __shared__ __attribute__((aligned(16))) char lds[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
#ifdef SER_GEMM_DBG
    const unsigned long long dbg_e0 = __builtin_amdgcn_s_memtime();
#endif

    const int ntn = (p.N + BN - 1) / BN;
    const int ntm = (p.M + BM - 1) / BM;
    int bid = blockIdx.x;
    {   // bijective XCD remap: blocks with equal (bid & 7) share an L2
        const int nwg = ntm * ntn;
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
        const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
        bid = base + (bid >> 3);
    }
    // Tile order inside an XCD's run (round 5).  The blocks of one XCD start in id order on its 32 CUs, so 32 consecutive ids are resident
    // together and share that XCD's 4 MiB L2.  Row-major ids make them ONE row of tiles when the grid is >= 32 tiles wide (FC1: one A panel,
    // 32 weight panels: 1 + 32 x 1/2 = 17 panel fetches per K step for 32 tiles); as a 4 x 8 super-tile they fetch 4 + 8 x 1/2 = 8 -- the L2 -> LDS
    // fill, which bounds these K loops, is served from L2 (~70 GB/s per CU) instead of the Infinity Cache (~33) for that much more of it.
    // Needs ntn % 8 == 0 (otherwise the row-major order stays); the last, partial super-row keeps 8-wide groups of its own height.
    int mt, nt;
    if (SER_GEMM_SUPER && (ntn & 7) == 0 && ntn > 8) {
        const int per_sr = 4 * ntn;                                   // tiles per full super-row
        const int sr = bid / per_sr;
        const int tt = bid - sr * per_sr;
        const int h = (sr * 4 + 4 <= ntm) ? 4 : ntm - sr * 4;          // height of this super-row
        const int sc = tt / (h * 8), j = tt - sc * (h * 8);
        mt = sr * 4 + j / 8;
        nt = sc * 8 + (j & 7);
    } else {
        mt = bid / ntn;
        nt = bid - mt * ntn;
    }
    const int g = blockIdx.y;
    const int m0 = mt * BM, n0 = nt * BN;

    const unsigned short* Abase = (const unsigned short*)p.A + (int64_t)g * p.a_group_stride;
    const unsigned short* Wbase = (const unsigned short*)p.W + (int64_t)g * p.w_group_stride;

    // ---- per-lane DMA sources: a 32-bit byte offset from a UNIFORM tile base ---------------------
    // (round 5) global_load_lds is issued in its scalar-base form -- global_load_lds_dwordx4 v_offset, s[base:base+1] -- from inline asm:
    // the K offset of a tile, the plane and the group go into the 64-bit SGPR base with scalar adds, the lane contributes the 32-bit
    // offset of its row / chunk inside the block's tile and there is NO vector instruction in front of a DMA.  With the builtin hipcc
    // keeps 64-bit VGPR pointers and emits two v_lshl_add_u64 per DMA (it never selects the scalar-base form); those vector instructions
    // compete for the SIMD's issue slots with the partner wave's MFMAs -- the DMA-issuing phase of one wave and the MFMA phase of the other
    // were not overlapping (FP16M FC2 at M = 7 984: MFMAs alone 82 us, DMA + reads alone 79 us, together 127 us; tools/ab what-if builds).
    // Offsets are relative to the tile's first row (a_rowoff maps must be non-decreasing in m -- every map the hosts build is), so they
    // stay far below 2^32 whatever the tensor size.
    const int prow = lane / CH, ppos = lane % CH;
    uint32_t aoff[LA], woff[LW];
    const int m0c = m0 < p.M ? m0 : p.M - 1;
    const int64_t arow0 = p.a_rowoff ? (int64_t)p.a_rowoff[m0c] * 8 : (int64_t)m0c * p.lda;      // uniform
    const char* const Atile = uniform_ptr(Abase + arow0);
    const char* const Wtile = uniform_ptr(Wbase + (int64_t)n0 * p.K);
#pragma unroll
    for (int q = 0; q < LA; ++q) {
        const int R = (q * NW + wave) * RPP + prow;                               // LDS row this lane fills
        const int f = (BK == 64) ? (R & 7) : ((R >> 1) & 3);
        const int c = ppos ^ f;                                                   // logical chunk to fetch
        int m = m0 + R;
        m = m < p.M ? m : p.M - 1;
        const int64_t arow = p.a_rowoff ? (int64_t)p.a_rowoff[m] * 8 : (int64_t)m * p.lda;
        aoff[q] = (uint32_t)((arow - arow0 + c * 8) * 2);
    }
#pragma unroll
    for (int q = 0; q < LW; ++q) {
        const int R = (q * NW + wave) * RPP + prow;
        const int f = (BK == 64) ? (R & 7) : ((R >> 1) & 3);
        const int c = ppos ^ f;
        int n = n0 + R;
        n = n < p.N ? n : p.N - 1;
        woff[q] = (uint32_t)(((int64_t)(n - n0) * p.K + c * 8) * 2);
    }
    const uint32_t lds0 = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)lds);

    const int nk = p.K / BK;
    const int total = M16 ? 2 * nk : nk;          // FP16M: two ring units per K tile
    const int tpc = p.kc ? p.kc / BK : 0x7fffffff;                                // K tiles per conv chunk

    // FP16M: block scales of the tile's rows, one word per (row, K tile), fetched with the cross-term unit: wave w brings 64 of the
    // BM + BN words (A rows first), 4 bytes per lane; with more waves than 64-row chunks the surplus waves repeat a chunk (same bytes,
    // same place), so every wave issues the same number of DMAs and the counted waits stay uniform
    unsigned char* const scl = (unsigned char*)(lds + ST * STAGE + (LNEPI ? 0 : BM * 8));     // [ST][SCW] words, behind the row statistics
    const int sc_chunk = M16 ? (wave % (SCW / 64)) : 0;
    const bool sc_is_a = sc_chunk * 64 < BM;                                     // wave-uniform: this wave's 64 words are A rows / W rows
    const char* const sc_base = uniform_ptr(sc_is_a ? p.a_scale : p.w_scale);
    const int64_t sc_step = (sc_is_a ? p.a_scale_ld : p.w_scale_ld) * 4;       // bytes between K tiles
    uint32_t sc_off = 0;
    if constexpr (M16) {
        const int r = sc_chunk * 64 + lane;
        int i = sc_is_a ? m0 + r : n0 + r - BM;
        const int lim = sc_is_a ? p.M : p.N;
        i = i < lim ? i : lim - 1;
        sc_off = (uint32_t)i * 4u;
    }
    int i_kk = 0, i_cc = 0, i_cj = 0, i_stage = 0;                               // issue-side scalar state
    // one ring unit = LPT DMA pieces per wave.  issue() sends them back to back; the software-pipelined 4-wave tile sends them one at a time
    // between its MFMAs (issue_begin / issue_piece(j) / issue_end).
    const char* is_ua = nullptr; const char* is_uw = nullptr;
    uint32_t is_dstA = 0, is_dstW = 0;
    auto issue_begin = [&]() {
        int64_t koffA = (int64_t)i_cj * p.ldj + (int64_t)i_cc * BK;
        int64_t koffW = (int64_t)i_kk * BK;
        if constexpr (M16) {                                     // unit i_kk: K tile i_kk / 2, plane i_kk & 1 (same byte offsets in both planes)
            koffA = (int64_t)(i_kk >> 1) * BK + (i_kk & 1) * p.a_plane_stride;
            koffW = (int64_t)(i_kk >> 1) * BK + (i_kk & 1) * p.w_plane_stride;
            if (i_kk & 1) dma_lds<4>(sc_base + (int64_t)(i_kk >> 1) * sc_step, sc_off, lds0 + (uint32_t)(ST * STAGE + (LNEPI ? 0 : BM * 8) + (i_stage * SCW + sc_chunk * 64) * 4));
        }
        is_dstA = lds0 + i_stage * STAGE + wave * 1024;          // stage = [A hi][A lo][W hi][W lo]
        is_dstW = is_dstA + NPL * A_BYTES;
        is_ua = Atile + koffA * 2;
        is_uw = Wtile + koffW * 2;
    };
    auto issue_piece = [&](int j) {                              // j in [0, LPT): plane-major, A pieces then W pieces (compile-time j)
        const int pl = j / (LA + LW), q = j % (LA + LW);
        if (q < LA) dma_lds<16>(is_ua + (int64_t)pl * p.a_plane_stride * 2, aoff[q < LA ? q : 0], is_dstA + pl * A_BYTES + q * NW * 1024);
        else dma_lds<16>(is_uw + (int64_t)pl * p.w_plane_stride * 2, woff[q >= LA ? q - LA : 0], is_dstW + pl * W_BYTES + (q - LA) * NW * 1024);
    };
    auto issue_end = [&]() {
        ++i_kk; ++i_cc;
        if (i_cc == tpc) { i_cc = 0; ++i_cj; }
        i_stage = (i_stage + 1 == ST) ? 0 : i_stage + 1;
    };
    auto issue = [&]() {
        issue_begin();
#pragma unroll
        for (int j = 0; j < LPT; ++j) issue_piece(j);
        issue_end();
    };

    f32x4 acc[TN][TM];
#pragma unroll
    for (int a = 0; a < TN; ++a)
#pragma unroll
        for (int b = 0; b < TM; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // per-lane fragment read offsets; f(row) depends only on the lane because tile bases are multiples of 16
    const int frow = lane & 15, fq = lane >> 4;
    const int fsw = (BK == 64) ? (lane & 7) : ((frow >> 1) & 3);
    int offA[KS], offW[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        const int phys = ((s * 4 + fq) ^ fsw) << 4;
        offA[s] = (wm * TM * 16 + frow) * ROWB + phys;
        offW[s] = NPL * A_BYTES + (wn * TN * 16 + frow) * ROWB + phys;
    }

    // Natural MFMA accumulator layout: acc[ni][mi][r] is row (mi*16 + frow), column (ni*16 + fq*4 + r) of the
    // wave tile, so the four lanes of a row cover 16 consecutive columns: fp32 stores / residual loads are 64
    // contiguous bytes per row, and the bf16 stores reach the same after one cross-row register swap.
    constexpr int CPL = TN * 4;
    const int ncol0 = n0 + wn * (TN * 16) + fq * 4;                   // column of acc[0][.][0] within the group
    const int64_t gcol = (int64_t)g * p.c_group_stride + ncol0;       // in the output matrices
    // per-column epilogue constants (bias, deferred-LayerNorm column sums): 2 * TN * 4 registers.  The FP16M tiles request them AFTER the
    // K loop -- beside the 128 accumulators and the 64 + 8 registers of a cross-term phase they would spill (86 - 116 registers, some
    // inside the loop, on the 256x256 tile); everybody else keeps them across the loop (the loads fly beside the first DMA tiles)
    float bias[CPL], csum[CPL];
    auto load_cols = [&]() {
#pragma unroll
        for (int j = 0; j < CPL; ++j) { bias[j] = 0.f; csum[j] = 0.f; }
        if (p.bias) {
#pragma unroll
            for (int j4 = 0; j4 < TN; ++j4)
                if (ncol0 + j4 * 16 < p.N) {
                    const f32x4 b = *(const f32x4*)(p.bias + (int64_t)g * p.N + ncol0 + j4 * 16);
                    bias[j4 * 4 + 0] = b[0]; bias[j4 * 4 + 1] = b[1]; bias[j4 * 4 + 2] = b[2]; bias[j4 * 4 + 3] = b[3];
                }
        }
        // deferred LayerNorm (see ser_hip.h): per-column sum of the gamma-folded weights
        if (p.ln_colsum) {
#pragma unroll
            for (int j4 = 0; j4 < TN; ++j4)
                if (ncol0 + j4 * 16 < p.N) {
                    const f32x4 b = *(const f32x4*)(p.ln_colsum + ncol0 + j4 * 16);
                    csum[j4 * 4 + 0] = b[0]; csum[j4 * 4 + 1] = b[1]; csum[j4 * 4 + 2] = b[2]; csum[j4 * 4 + 3] = b[3];
                }
        }
    };
    constexpr bool LATE_COLS = M16 || (WM * WN == 4 && TM * TN == 64) || (SER_GEMM_PSW != 0 && WM * WN == 8 && BK == 64 && ST == 2 && !LNEPI && mode_traits<MODE>::planes == 1);      // ... and the 4-wave 256x256 tile: its 256 non-accumulator registers
    if constexpr (!LATE_COLS) load_cols();                                  // hold two fragment buffers (128) already

#pragma unroll
    for (int t = 0; t < ST - 1; ++t)
        if (t < total) issue();

    // deferred LayerNorm: row mean / rstd of the A rows from the producer's partial sums, into the
    // LDS words behind the ring (one row per thread; the loads fly beside the first DMA tiles.  Requesting them BEFORE
    // the tiles -- so that they land first -- was measured and changes nothing: the reduction is short, what the block
    // waits for is one memory round trip either way)
    float* lnst = (float*)(lds + ST * STAGE);                         // [BM][2]
    if (p.ln_stats_in) {
        for (int r = tid; r < BM; r += NT) {
            int m = m0 + r;
            m = m < p.M ? m : p.M - 1;
            const float* ps = p.ln_stats_in + (int64_t)m * p.ln_groups * 2;
            // all partials are requested before any is consumed (up to 32 groups = 16 loads in flight)
            f32x4 pv[16];
#pragma unroll
            for (int u = 0; u < 16; ++u)
                pv[u] = (2 * u < p.ln_groups) ? *(const f32x4*)(ps + 4 * u) : (f32x4){0.f, 0.f, 0.f, 0.f};
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int u = 0; u < 16; ++u) { s1 += pv[u][0] + pv[u][2]; s2 += pv[u][1] + pv[u][3]; }
            const float invK = 1.0f / (float)p.K;
            const float mu = s1 * invK;
            const float var = fmaxf(s2 * invK - mu * mu, 0.f);
            lnst[2 * r] = mu;
            lnst[2 * r + 1] = rsqrtf(var + p.ln_eps);
            // absolute row mean of the A rows (their partials are relative to ln_shift): the next producer's shift
            if (p.mean_out && nt == 0 && g == 0 && m0 + r < p.M) p.mean_out[m] = mu + (p.ln_shift ? p.ln_shift[m] : 0.f);
            // (relative mean, rstd) of the rows for ser_attention's in-kernel WavLM gate, which normalises the same operand copy
            if (p.lnstat_out && nt == 0 && g == 0 && m0 + r < p.M) *(f32x2*)(p.lnstat_out + 2 * (int64_t)m) = (f32x2){mu, lnst[2 * r + 1]};
        }
        __syncthreads();
    }

    // Ping-pong schedule of the 8-wave tiles (two waves per SIMD: wave w and w + 4).  Left alone, both waves of a SIMD leave
    // the per-tile barrier together, read their fragments together and then compete for the matrix pipe together.  Here
    // waves 4..7 run ONE PHASE behind waves 0..3: a phase is either a batch of LDS reads (the fragments of one k-step for
    // the 128x64 / 64x128 wave tiles, of the whole K tile for 64x64 ones, plus this wave's share of the next DMA) or the 32
    // MFMAs that consume them; every phase ends in a barrier, so while one wave of a SIMD owns the matrix pipe the other one
    // owns the LDS port.  With PH read phases per K tile, the early half reads batch (kt, s) in global phase 2 (PH kt + s) and
    // multiplies in the next one; the late half does the same one phase later.  The stage of tile kt - 1 is last read in
    // phase 2 PH kt - 1 (late half), so both halves may refill it from their first read phase of tile kt on, and both make
    // sure tile kt + 1 has landed at the end of phase 2 PH (kt + 1) - 1, before the early half reads it.  Barrier counts match:
    // the late half has one extra up front and skips the one after its last MFMA phase -- which overlaps the early half's
    // epilogue.  Same arithmetic, same accumulation order: results are bit-identical to the plain ring.
#ifdef SER_GEMM_DBG
    const unsigned long long dbg_t0 = __builtin_amdgcn_s_memtime(), dbg_r0 = __builtin_amdgcn_s_memrealtime();
#endif
    constexpr int PP_BIT = NPL == 2 ? (LNEPI ? 16 : 8) : (LNEPI ? 4 : (TM * TN >= 32 ? 1 : 2));
    constexpr bool PSW8 = NPL == 1 && !M16 && KS == 2 && ST == 2 && !LNEPI && NW == 8 && ((TM * TN == 32 && (SER_GEMM_PSW & 1)) || (TM * TN == 16 && (SER_GEMM_PSW & 2)));
    constexpr bool PP = !M16 && !PSW8 && (SER_GEMM_PP & PP_BIT) && NW == 8 && (KS == 2 || NPL == 2);
    constexpr int PH = (NPL == 1 && TM * TN >= 32) ? KS : 1;          // read phases per K tile (FP32X: 12 fragments + 24 MFMAs per k-step, one phase)
    constexpr int SPP = KS / PH;                                      // k-steps per phase
    // FP32X tiles with 64x128 wave tiles (the LayerNorm tile: one k-step per 32-deep K tile, 24 fragments, 96 MFMAs): the two read
    // phases of a K tile are the two halves of the weight fragments (the activation fragments stay in registers across them), so
    // 64 fragment registers are live beside the 128 accumulators instead of 96
    constexpr bool PPW = PP && NPL == 2 && KS == 1 && TN >= 8;
    if constexpr (PPW) {
        constexpr int TNH = TN / 2;
        const int late = wave >> 2;
        if (total >= ST - 1) wait_vmcnt<LPT * (ST - 2)>(); else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        if (late) __builtin_amdgcn_s_barrier();
        int stage = 0;
        for (int kt = 0; kt < total; ++kt) {
            const bool more = kt + ST - 1 < total;
            if (more) issue();
            const char* sb = lds + stage * STAGE;
            stage = (stage + 1 == ST) ? 0 : stage + 1;
            const bool last = kt + 1 == total;
            bf16x8 ah[TM], al[TM];
#pragma unroll
            for (int ph = 0; ph < 2; ++ph) {
                bf16x8 wh[TNH], wl[TNH];
                if (ph == 0) {
#pragma unroll
                    for (int x = 0; x < TM; ++x) {
                        ah[x] = *(const bf16x8*)(sb + offA[0] + x * 16 * ROWB);
                        al[x] = *(const bf16x8*)(sb + offA[0] + A_BYTES + x * 16 * ROWB);
                    }
                }
#pragma unroll
                for (int x = 0; x < TNH; ++x) {
                    wh[x] = *(const bf16x8*)(sb + offW[0] + (ph * TNH + x) * 16 * ROWB);
                    wl[x] = *(const bf16x8*)(sb + offW[0] + W_BYTES + (ph * TNH + x) * 16 * ROWB);
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (ph == 1 && late) { if (more) wait_vmcnt<LPT * (ST - 2)>(); else wait_vmcnt<0>(); }
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int nj = 0; nj < TNH; ++nj)
#pragma unroll
                    for (int mi = 0; mi < TM; ++mi) {
                        const int ni = ph * TNH + nj;
                        acc[ni][mi] = mfma16<MODE>(wl[nj], ah[mi], acc[ni][mi]);
                        acc[ni][mi] = mfma16<MODE>(wh[nj], al[mi], acc[ni][mi]);
                        acc[ni][mi] = mfma16<MODE>(wh[nj], ah[mi], acc[ni][mi]);
                    }
                __builtin_amdgcn_s_setprio(0);
                if (ph == 1 && !late) { if (more) wait_vmcnt<LPT * (ST - 2)>(); else wait_vmcnt<0>(); }
                __builtin_amdgcn_sched_barrier(0);
                if (!(ph == 1 && last && late)) __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    if constexpr (PP && !PPW) {
        const int late = wave >> 2;
        if (total >= ST - 1) wait_vmcnt<LPT * (ST - 2)>(); else wait_vmcnt<0>();    // fewer tiles in flight when K is short
        __builtin_amdgcn_s_barrier();                                 // tile 0 is visible to every wave
        __builtin_amdgcn_sched_barrier(0);
        if (late) __builtin_amdgcn_s_barrier();
        int stage = 0;
        for (int kt = 0; kt < total; ++kt) {
            const bool more = kt + ST - 1 < total;
            if (more) issue();
            const char* sb = lds + stage * STAGE;
            stage = (stage + 1 == ST) ? 0 : stage + 1;
            const bool last = kt + 1 == total;
#pragma unroll
            for (int ph = 0; ph < PH; ++ph) {
                bf16x8 af[SPP][TM], wf[SPP][TN];
                bf16x8 al[NPL == 2 ? SPP : 1][NPL == 2 ? TM : 1], wl[NPL == 2 ? SPP : 1][NPL == 2 ? TN : 1];   // FP32X: the lo planes
#pragma unroll
                for (int u = 0; u < SPP; ++u) {
#pragma unroll
                    for (int x = 0; x < TM; ++x) {
                        af[u][x] = *(const bf16x8*)(sb + offA[ph * SPP + u] + x * 16 * ROWB);
                        if constexpr (NPL == 2) al[u][x] = *(const bf16x8*)(sb + offA[ph * SPP + u] + A_BYTES + x * 16 * ROWB);
                    }
#pragma unroll
                    for (int x = 0; x < TN; ++x) {
                        wf[u][x] = *(const bf16x8*)(sb + offW[ph * SPP + u] + x * 16 * ROWB);
                        if constexpr (NPL == 2) wl[u][x] = *(const bf16x8*)(sb + offW[ph * SPP + u] + W_BYTES + x * 16 * ROWB);
                    }
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (ph == PH - 1 && late) { if (more) wait_vmcnt<LPT * (ST - 2)>(); else wait_vmcnt<0>(); }
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int u = 0; u < SPP; ++u)
#pragma unroll
                    for (int ni = 0; ni < TN; ++ni)
#pragma unroll
                        for (int mi = 0; mi < TM; ++mi) {
                            if constexpr (NPL == 2) {                     // same product order as the plain ring: lo*hi, hi*lo, hi*hi
                                acc[ni][mi] = mfma16<MODE>(wl[u][ni], af[u][mi], acc[ni][mi]);
                                acc[ni][mi] = mfma16<MODE>(wf[u][ni], al[u][mi], acc[ni][mi]);
                                acc[ni][mi] = mfma16<MODE>(wf[u][ni], af[u][mi], acc[ni][mi]);
                            } else {
                                acc[ni][mi] = mfma16<MODE>(wf[u][ni], af[u][mi], acc[ni][mi]);
                            }
                        }
                __builtin_amdgcn_s_setprio(0);
                if (ph == PH - 1 && !late) { if (more) wait_vmcnt<LPT * (ST - 2)>(); else wait_vmcnt<0>(); }
                __builtin_amdgcn_sched_barrier(0);
                if (!(ph == PH - 1 && last && late)) __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }

    // ---- FP16M: per 64-deep K tile one H unit (fp16 hi planes: the two f16 k-steps of the single-plane loop) and one E unit (e4m3
    // cross-term planes: each lane's 32 operand bytes are chunks fq and fq + 4 of its row -- the addresses of the two k-steps -- and
    // its scale byte is byte fq of the row's scale word; the instruction pairs chunk c of the weights with chunk c of the activations
    // and applies the scale of lane group b to chunks 2b, 2b + 1: [P cols 0-31, P cols 32-63, Q cols 0-31, Q cols 32-63], tools/mxprobe.hip).
    // Same ring, same ping-pong phases as the single-plane tiles; an E phase of the 256x128 / 128x128 tiles is TM x TN = 16 scaled MFMAs = the
    // matrix-pipe time of the 32 f16 MFMAs of its H phase.
    if constexpr (M16) {
        constexpr int LPTH = LPT, LPTE = LPT + 1;
        constexpr bool PPM = (NW == 8);
        constexpr int PHM = (TM * TN >= 32) ? 2 : 1;                  // phases per unit (256x256: two, like its bf16 form)
        constexpr int SPM = KS / PHM;                                 // H unit: k-steps per phase
        // E unit: phases and activation row fragments per phase
        constexpr int PHE = PHM;
        constexpr int TMP = TM / PHE;
        // unit kt has landed once at most the youngest unit (kt + ST - 1 ... only ST = 2, 3 exist) is still in flight
        auto wait_landed = [&](bool more, int young) {
            if (!more || ST == 2) wait_vmcnt<0>();
            else if (young & 1) wait_vmcnt<LPTE * (ST - 2)>();
            else wait_vmcnt<LPTH * (ST - 2)>();
        };
        auto h_phase_reads = [&](const char* sb, int ph, auto& af, auto& wf) {
#pragma unroll
            for (int u = 0; u < SPM; ++u) {
#pragma unroll
                for (int x = 0; x < TM; ++x) af[u][x] = *(const bf16x8*)(sb + offA[ph * SPM + u] + x * 16 * ROWB);
#pragma unroll
                for (int x = 0; x < TN; ++x) wf[u][x] = *(const bf16x8*)(sb + offW[ph * SPM + u] + x * 16 * ROWB);
            }
        };
        auto h_phase_mfma = [&](auto& af, auto& wf) {
#pragma unroll
            for (int u = 0; u < SPM; ++u)
#pragma unroll
                for (int ni = 0; ni < TN; ++ni)
#pragma unroll
                    for (int mi = 0; mi < TM; ++mi) acc[ni][mi] = mfma16<SER_MODE_FP16>(wf[u][ni], af[u][mi], acc[ni][mi]);
        };
        auto e_phase_reads = [&](const char* sb, const unsigned char* scb, int ph, auto& a8, auto& w8, auto& sa, auto& sw) {
#pragma unroll
            for (int x = 0; x < TMP; ++x) {
                const int mi = ph * TMP + x;
                const u32x4 c0 = *(const u32x4*)(sb + offA[0] + mi * 16 * ROWB), c1 = *(const u32x4*)(sb + offA[1] + mi * 16 * ROWB);
                a8[x] = (i32x8){(int)c0[0], (int)c0[1], (int)c0[2], (int)c0[3], (int)c1[0], (int)c1[1], (int)c1[2], (int)c1[3]};
                sa[x] = scb[(wm * TM * 16 + mi * 16 + frow) * 4 + fq];
            }
#pragma unroll
            for (int x = 0; x < TN; ++x) {
                const u32x4 c0 = *(const u32x4*)(sb + offW[0] + x * 16 * ROWB), c1 = *(const u32x4*)(sb + offW[1] + x * 16 * ROWB);
                w8[x] = (i32x8){(int)c0[0], (int)c0[1], (int)c0[2], (int)c0[3], (int)c1[0], (int)c1[1], (int)c1[2], (int)c1[3]};
                sw[x] = scb[(BM + wn * TN * 16 + x * 16 + frow) * 4 + fq];
            }
        };
        auto e_phase_mfma = [&](int ph, auto& a8, auto& w8, auto& sa, auto& sw) {
#pragma unroll
            for (int ni = 0; ni < TN; ++ni)
#pragma unroll
                for (int x = 0; x < TMP; ++x)
                    mfma_scale8(acc[ni][ph * TMP + x], w8[ni], a8[x], sw[ni], sa[x]);
        };
        // (the two kinds of unit are separate instantiations of one lambda, called alternately: with a run-time parity branch around the
        // phase bodies hipcc kept both fragment sets live and spilled 160 - 540 registers)
        if constexpr (PPM) {
            const int late = wave >> 2;
            wait_landed(total >= ST - 1, ST - 2);                         // units 0 .. ST-2 are in flight; unit 0 has landed
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            if (late) __builtin_amdgcn_s_barrier();
            int stage = 0;
            auto unit = [&](int kt, auto etag) {
                constexpr bool EU = decltype(etag)::value;
                const bool more = kt + ST - 1 < total;
#ifdef SER_GEMM_DBG
                unsigned long long stp[8] = {0, 0, 0, 0, 0, 0, 0, 0};     // phase stamps of units 8 .. 15 of one block (tools/gemm_m16_phases.py)
                const bool stamp = ser_gemm_dbg_dev && blockIdx.x == 40 && lane == 0 && kt >= 8 && kt < 16;
#define SER_STAMP(i) do { __builtin_amdgcn_sched_barrier(0); if (stamp) stp[i] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } while (0)
                SER_STAMP(0);
#else
#define SER_STAMP(i) do { } while (0)
#endif
#if !SER_M16_READS_FIRST
#if SER_WHATIF & 2
                if (more && kt < 2) issue();                       // what-if: the ring is filled once, then no more DMA (wrong results)
#else
                if (more) issue();
#endif
#endif
                SER_STAMP(1);                                      // DMAs of unit kt + ST - 1 issued
                const char* sb = lds + stage * STAGE;
                const unsigned char* scb = scl + stage * SCW * 4;
                stage = (stage + 1 == ST) ? 0 : stage + 1;
                const bool last = kt + 1 == total;
                constexpr int NPH = EU ? PHE : PHM;
#pragma unroll
                for (int ph = 0; ph < NPH; ++ph) {
                    bf16x8 af[EU ? 1 : SPM][EU ? 1 : TM], wf[EU ? 1 : SPM][EU ? 1 : TN];
                    i32x8 a8[EU ? TMP : 1], w8[EU ? TN : 1];
                    int sa[EU ? TMP : 1], sw[EU ? TN : 1];
#if SER_WHATIF & 4
                    if (kt < 2) {                                  // what-if: fragments read for the first K tile only (wrong results)
#endif
                    if constexpr (!EU) h_phase_reads(sb, ph, af, wf);
                    else e_phase_reads(sb, scb, ph, a8, w8, sa, sw);
#if SER_WHATIF & 4
                    }
#endif
#if SER_M16_READS_FIRST
                    // the DMAs of unit kt + ST - 1 go out BEHIND the fragment reads of the unit's first phase: the LDS latency of those reads
                    // passes under the ~350 cycles the DMA issue holds the wave (the stage they refill was last read one unit ago)
                    if (ph == 0 && more) { __builtin_amdgcn_sched_barrier(0); issue(); }
#endif
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    if (ph == 0) SER_STAMP(2);                      // fragments in registers
                    if (ph == NPH - 1 && late) wait_landed(more, kt + ST - 1);
                    if (ph == 0) SER_STAMP(3);
                    __builtin_amdgcn_sched_barrier(0);
                    __builtin_amdgcn_s_barrier();
                    __builtin_amdgcn_sched_barrier(0);
                    if (ph == 0) SER_STAMP(4);                      // past the barrier in front of the MFMA phase
#if SER_M16_PRIO == 1
                    __builtin_amdgcn_s_setprio(1);
#elif SER_M16_PRIO == 2
                    __builtin_amdgcn_s_setprio(0);
#endif
#if SER_WHATIF & 1
                    if (kt < 2) {                                  // what-if: matrix instructions for the first K tile only (wrong results)
#endif
                    if constexpr (!EU) h_phase_mfma(af, wf);
                    else e_phase_mfma(ph, a8, w8, sa, sw);
#if SER_WHATIF & 1
                    }
#endif
#if SER_M16_PRIO == 1
                    __builtin_amdgcn_s_setprio(0);
#elif SER_M16_PRIO == 2
                    __builtin_amdgcn_s_setprio(1);                 // the read / DMA-issue phase that follows runs at priority
#endif
                    if (ph == 0) SER_STAMP(5);                      // MFMAs issued
                    if (ph == NPH - 1 && !late) wait_landed(more, kt + ST - 1);
                    if (ph == 0) SER_STAMP(6);
                    __builtin_amdgcn_sched_barrier(0);
                    if (!(ph == NPH - 1 && last && late)) __builtin_amdgcn_s_barrier();
                    __builtin_amdgcn_sched_barrier(0);
                    if (ph == 0) SER_STAMP(7);
                }
#ifdef SER_GEMM_DBG
                if (stamp) for (int i = 0; i < 8; ++i) ser_gemm_dbg_dev[262144 + ((wave * 8 + (kt - 8)) * 8 + i)] = stp[i];
#endif
            };
            for (int kt = 0; kt < total; kt += 2) {
                unit(kt, std::false_type{});
                unit(kt + 1, std::true_type{});
            }
        } else {
            int stage = 0;
            auto unit = [&](int kt, auto etag) {
                constexpr bool EU = decltype(etag)::value;
                wait_landed(kt + (ST - 2) < total, kt + ST - 2);
                __builtin_amdgcn_s_barrier();
                if (kt + ST - 1 < total) issue();
                const char* sb = lds + stage * STAGE;
                const unsigned char* scb = scl + stage * SCW * 4;
                stage = (stage + 1 == ST) ? 0 : stage + 1;
#pragma unroll
                for (int ph = 0; ph < (EU ? PHE : PHM); ++ph) {
                    if constexpr (!EU) {
                        bf16x8 af[SPM][TM], wf[SPM][TN];
                        h_phase_reads(sb, ph, af, wf);
                        __builtin_amdgcn_s_setprio(1);
                        h_phase_mfma(af, wf);
                        __builtin_amdgcn_s_setprio(0);
                    } else {
                        i32x8 a8[TMP], w8[TN];
                        int sa[TMP], sw[TN];
                        e_phase_reads(sb, scb, ph, a8, w8, sa, sw);
                        __builtin_amdgcn_s_setprio(1);
                        e_phase_mfma(ph, a8, w8, sa, sw);
                        __builtin_amdgcn_s_setprio(0);
                    }
                }
            };
            for (int kt = 0; kt < total; kt += 2) {
                unit(kt, std::false_type{});
                unit(kt + 1, std::true_type{});
            }
        }
    }
    // ---- 256x256 on FOUR waves, one per SIMD, hand-placed software pipeline (round 5; the round-4 verdict's "one bounded attempt").
    // A wave owns 128x128 of the tile (64 MFMA tiles, 256 accumulator registers of its 512) and nothing else runs on its SIMD, so every
    // latency has to pass under its own MFMAs: the 64 MFMAs of a k-step are issued in groups of four with ONE other instruction behind each
    // group -- the 16 fragment reads of the NEXT k-step (second fragment buffer) and, in the second k-step of a K tile, the 16 DMA pieces of
    // the tile two ahead -- and a sched_barrier after every group keeps hipcc from regrouping them (left alone it clusters the reads in
    // front: the compiler-scheduled form of this tile lost 30 % in round 2).  One s_barrier per K tile, at the start of its second k-step:
    // behind it every wave's pieces of tile kt + 1 have landed (they were issued a whole K tile earlier) and every wave has read the last
    // fragments of tile kt - 1's stage... which the DMAs that follow refill.
    // The same pipeline on the 8-wave tiles (SER_GEMM_PSW bit 1: 256x256, bit 2: 256x128 on a TWO-stage ring): two waves per SIMD each run it on
    // their own 128x64 / 64x64 wave tile and share the matrix pipe instruction by instruction instead of phase by phase.
    constexpr bool P4 = NPL == 1 && !M16 && KS == 2 && ST == 2 && !LNEPI &&
                        ((NW == 4 && TM * TN == 64) || (NW == 8 && TM * TN == 32 && (SER_GEMM_PSW & 1)) || (NW == 8 && TM * TN == 16 && (SER_GEMM_PSW & 2)));
    if constexpr (P4) {
        constexpr int NM = TM * TN, NF = TM + TN;                     // MFMAs and fragment reads per k-step
        constexpr int RS = (NM * 3 / 4) / NF < 1 ? 1 : (NM * 3 / 4) / NF;   // a fragment read behind every RS-th MFMA: the last quarter of the MFMAs covers the last read's latency
        static_assert(LPT <= NF, "DMA pieces per wave fit the read slots");
        bf16x8 fa[2][TM], fw[2][TN];
        auto frag_read = [&](const char* sb, int step, int j, int buf) {         // j in [0, NF): A fragments first (the first MFMAs need all of them)
            if (j < TM) fa[buf][j < TM ? j : 0] = *(const bf16x8*)(sb + offA[step] + j * 16 * ROWB);
            else fw[buf][j >= TM ? j - TM : 0] = *(const bf16x8*)(sb + offW[step] + (j - TM) * 16 * ROWB);
        };
        if (1 < total) issue();                                       // tiles 0 and 1 in flight
        if (1 < total) wait_vmcnt<LPT>(); else wait_vmcnt<0>();       // tile 0 has landed
        __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int j = 0; j < NF; ++j) frag_read(lds, 0, j, 0);
        // the side instructions sit behind MFMAs 2, 5, 8, ... 47 (reads) and 0, 3, ... 45 (DMA pieces): the last 16 MFMAs of a k-step cover
        // the latency of its last fragment read.  The MFMAs are issued from inline asm with the accumulator TIED and in the accumulation half
        // of the register file ("+a"): with the builtin hipcc moved the 256 accumulators between VGPRs, AGPRs and scratch across the loop's
        // branches (341 spilled registers, 144 v_accvgpr moves per k-step).  Consecutive MFMAs target different accumulators; the fragment
        // registers come from ds_read (the compiler waits lgkmcnt for asm operands like for any other use); mfma_drain() before the epilogue.
        auto mma = [&](f32x4& c, const bf16x8& w, const bf16x8& x) {
            if constexpr (mode_traits<MODE>::f16) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(c) : "v"(w), "v"(x));
            else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(w), "v"(x));
        };
        auto step0 = [&](const char* sb) {                            // MFMAs on buffer 0; the fragments of k-step 1 go to buffer 1
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < NM; ++i) {
                mma(acc[i / TM][i % TM], fw[0][i / TM], fa[0][i % TM]);
                if (i % RS == RS - 1 && i / RS < NF) { frag_read(sb, 1, i / RS, 1); __builtin_amdgcn_sched_barrier(0); }
            }
        };
        auto step1 = [&](const char* sbn, auto more_tag, auto next_tag) {
            constexpr bool MORE = decltype(more_tag)::value, NEXT = decltype(next_tag)::value;
            // behind the barrier tile kt + 1 is visible to every wave and tile kt's stage has been read out by every wave (its last fragment
            // reads were issued 16 MFMAs ago): the DMAs of tile kt + 2 may refill it
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if constexpr (NEXT) {
                wait_vmcnt<0>();                                      // the pieces of tile kt + 1 (issued a K tile ago): nothing younger is in flight
                __builtin_amdgcn_s_barrier();
            }
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (MORE) issue_begin();
#pragma unroll
            for (int i = 0; i < NM; ++i) {
                mma(acc[i / TM][i % TM], fw[1][i / TM], fa[1][i % TM]);
                if constexpr (MORE) { if (i % RS == 0 && i / RS < LPT) { issue_piece(i / RS); __builtin_amdgcn_sched_barrier(0); } }
                if constexpr (NEXT) { if (i % RS == RS - 1 && i / RS < NF) { frag_read(sbn, 0, i / RS, 0); __builtin_amdgcn_sched_barrier(0); } }
            }
            if constexpr (MORE) issue_end();
        };
        int kt = 0;
        for (; kt + 2 < total; ++kt) {                                // steady state: no branch inside a K tile
            step0(lds + (kt & 1) * STAGE);
            step1(lds + ((kt + 1) & 1) * STAGE, std::true_type{}, std::true_type{});
        }
        if (kt + 1 < total) {                                         // second to last tile: nothing left to fetch
            step0(lds + (kt & 1) * STAGE);
            step1(lds + ((kt + 1) & 1) * STAGE, std::false_type{}, std::true_type{});
            ++kt;
        }
        step0(lds + (kt & 1) * STAGE);                                // last tile
        step1(lds, std::false_type{}, std::false_type{});
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");           // the last MFMAs have written their accumulators
    }
    if constexpr (M16) mfma_scale_drain();
    int c_stage = 0;
    for (int kt = 0; kt < ((PP || M16 || P4) ? 0 : total); ++kt) {
        // tile kt has landed once at most (ST-2) younger tiles are still in flight
        if (kt + (ST - 2) < total) wait_vmcnt<LPT * (ST - 2)>();
        else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        if (kt + ST - 1 < total) issue();           // refills the stage every wave finished reading before the barrier
        const char* sb = lds + c_stage * STAGE;
        c_stage = (c_stage + 1 == ST) ? 0 : c_stage + 1;
        if constexpr (NPL == 1) {
            // all k-steps' fragments are requested up front: the LDS reads of step s+1 complete under the
            // MFMAs of step s (the compiler places counted lgkmcnt waits between the clusters)
            // (PIPE; the 128x64-per-wave configuration has no registers to spare and loads per step)
            constexpr bool PIPE = (TM * TN * 4 + KS * (TM + TN) * 4) <= 208;
            bf16x8 af[KS][TM], wf[KS][TN];
            if constexpr (PIPE) {
#pragma unroll
                for (int s = 0; s < KS; ++s) {
#pragma unroll
                    for (int x = 0; x < TM; ++x) af[s][x] = *(const bf16x8*)(sb + offA[s] + x * 16 * ROWB);
#pragma unroll
                    for (int x = 0; x < TN; ++x) wf[s][x] = *(const bf16x8*)(sb + offW[s] + x * 16 * ROWB);
                }
            }
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                if constexpr (!PIPE) {
#pragma unroll
                    for (int x = 0; x < TM; ++x) af[s][x] = *(const bf16x8*)(sb + offA[s] + x * 16 * ROWB);
#pragma unroll
                    for (int x = 0; x < TN; ++x) wf[s][x] = *(const bf16x8*)(sb + offW[s] + x * 16 * ROWB);
                }
                __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int ni = 0; ni < TN; ++ni)
#pragma unroll
                    for (int mi = 0; mi < TM; ++mi)
                        acc[ni][mi] = mfma16<MODE>(wf[s][ni], af[s][mi], acc[ni][mi]);
                __builtin_amdgcn_s_setprio(0);
            }
        } else {
            // FP32X: 4 fragment sets per k-step, 3 products per (weight, activation) sub-tile pair
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                bf16x8 ah[TM], al[TM], wh[TN], wl[TN];
#pragma unroll
                for (int x = 0; x < TM; ++x) {
                    ah[x] = *(const bf16x8*)(sb + offA[s] + x * 16 * ROWB);
                    al[x] = *(const bf16x8*)(sb + offA[s] + A_BYTES + x * 16 * ROWB);
                }
#pragma unroll
                for (int x = 0; x < TN; ++x) {
                    wh[x] = *(const bf16x8*)(sb + offW[s] + x * 16 * ROWB);
                    wl[x] = *(const bf16x8*)(sb + offW[s] + W_BYTES + x * 16 * ROWB);
                }
                __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int ni = 0; ni < TN; ++ni)
#pragma unroll
                    for (int mi = 0; mi < TM; ++mi) {
                        acc[ni][mi] = mfma16<MODE>(wl[ni], ah[mi], acc[ni][mi]);
                        acc[ni][mi] = mfma16<MODE>(wh[ni], al[mi], acc[ni][mi]);
                        acc[ni][mi] = mfma16<MODE>(wh[ni], ah[mi], acc[ni][mi]);
                    }
                __builtin_amdgcn_s_setprio(0);
            }
        }
    }

#ifdef SER_GEMM_DBG
    {
        __builtin_amdgcn_sched_barrier(0);
        const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
        unsigned long long* d = ser_gemm_dbg_dev;
        if (d && wave == 0 && lane == 0 && blockIdx.y == 0) {
            d[blockIdx.x * 2] = t1 - dbg_t0; d[blockIdx.x * 2 + 1] = r1 - dbg_r0;
            d[131072 + blockIdx.x * 4] = dbg_t0 - dbg_e0; d[131072 + blockIdx.x * 4 + 1] = t1;
        }
    }
#endif
    if constexpr (LATE_COLS) load_cols();
    // ---- epilogue: lane owns row m (per mi) and TN*4 consecutive columns ----------------------
    float ramax = 0.f;                            // largest |value| this lane rounded to an fp16 operand plane (range guard)
    if constexpr (LNEPI) {
        // LayerNorm over the full row (N <= BN, one N tile) + GELU, two-pass statistics.
        // Row partials cross the WN waves of a block row through LDS (the ring is idle now).
        __syncthreads();
        float* red = (float*)lds;                                     // [BM][WN][2]
        const float invN = 1.0f / (float)p.N;
        float mean[TM], rstd[TM], nmr[TM];
        // one exchange: per-row (sum, sum of squares) partials of the WN waves of a block row.
        // (fp32 accumulators, |mean| << std for conv outputs: E[x^2]-mean^2 is safe here)
#pragma unroll
        for (int mi = 0; mi < TM; ++mi) {
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int ni = 0; ni < TN; ++ni)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float x = acc[ni][mi][r] + bias[ni * 4 + r];
                    acc[ni][mi][r] = x;
                    if (ncol0 + ni * 16 < p.N) { s1 += x; s2 += x * x; }
                }
            s1 += __shfl_xor(s1, 16, 64); s2 += __shfl_xor(s2, 16, 64);
            s1 += __shfl_xor(s1, 32, 64); s2 += __shfl_xor(s2, 32, 64);
            if (fq == 0) {
                red[((wm * TM * 16 + mi * 16 + frow) * WN + wn) * 2] = s1;
                red[((wm * TM * 16 + mi * 16 + frow) * WN + wn) * 2 + 1] = s2;
            }
        }
        __syncthreads();
#pragma unroll
        for (int mi = 0; mi < TM; ++mi) {
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int w = 0; w < WN; ++w) {
                s1 += red[((wm * TM * 16 + mi * 16 + frow) * WN + w) * 2];
                s2 += red[((wm * TM * 16 + mi * 16 + frow) * WN + w) * 2 + 1];
            }
            mean[mi] = s1 * invN;
            rstd[mi] = rsqrtf(fmaxf(s2 * invN - mean[mi] * mean[mi], 0.f) + p.ln_eps);
            nmr[mi] = -mean[mi] * rstd[mi];                            // (x - mean) * rstd = fma(x, rstd, nmr)
        }
        float lg[CPL], lb[CPL];
#pragma unroll
        for (int j4 = 0; j4 < TN; ++j4) {
            f32x4 a = {0.f, 0.f, 0.f, 0.f}, b = {0.f, 0.f, 0.f, 0.f};
            if (ncol0 + j4 * 16 < p.N) {
                a = *(const f32x4*)(p.ln_gamma + ncol0 + j4 * 16);
                b = *(const f32x4*)(p.ln_beta + ncol0 + j4 * 16);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) { lg[j4 * 4 + r] = a[r]; lb[j4 * 4 + r] = b[r]; }
        }
#pragma unroll
        for (int mi = 0; mi < TM; ++mi) {
            const int m = m0 + wm * TM * 16 + mi * 16 + frow;
            if (m >= p.M) continue;
            const int64_t orow = p.out_rowmap ? (int64_t)p.out_rowmap[m] : (int64_t)m;
#pragma unroll
            for (int ni = 0; ni < TN; ni += 2) {                      // fragment column blocks ni, ni+1: 4 + 4 columns
                const bool ok0 = ncol0 + ni * 16 < p.N, ok1 = ncol0 + ni * 16 + 16 < p.N;
                float v[8];
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    const int nj = ni + (r >> 2), rr = r & 3;
                    v[r] = fmaf(fmaf(acc[nj][mi][rr], rstd[mi], nmr[mi]), lg[nj * 4 + rr], lb[nj * 4 + rr]);   // 2 FMAs
                }
                if (p.act == SER_ACT_GELU) {
#pragma unroll
                    for (int r = 0; r < 8; r += 2) {
                        const f32x2 y = gelu2<MODE == SER_MODE_BF16>((f32x2){v[r], v[r + 1]});
                        v[r] = y[0]; v[r + 1] = y[1];
                    }
                }
                if (p.out_f32) {
                    f32x4 o0 = {v[0], v[1], v[2], v[3]}, o1 = {v[4], v[5], v[6], v[7]};
                    if (ok0) *(f32x4*)(p.out_f32 + (int64_t)m * p.ldo_f32 + gcol + ni * 16) = o0;
                    if (ok1) *(f32x4*)(p.out_f32 + (int64_t)m * p.ldo_f32 + gcol + ni * 16 + 16) = o1;
                }
                if (p.out_act) {
                    const int c8 = ni * 16 + ((fq & 1) ? 12 : 0);     // after the swap this lane holds columns c8..c8+7
                    store_act8_swap<MODE>((unsigned short*)p.out_act + orow * p.ldo_act + gcol + c8, p.out_plane_stride,
                                          ncol0 + c8 < p.N, v);
                    if constexpr (mode_traits<MODE>::f16) {           // fp16 range guard (ser_hip.h range_flag)
                        if (p.range_flag) {
#pragma unroll
                            for (int r = 0; r < 8; r += 2) ramax = fmaxf(ramax, fmaxf(fabsf(v[r]), fabsf(v[r + 1])));
                        }
                    }
                }
            }
        }
    } else {
        // wave-uniform feature flags: the column scale and the row partial sums cost 2 VALU each per element, and
        // most launches use neither (the epilogue is VALU-bound where it carries a GELU)
        const bool do_scale = p.col_scale_end > 0;
        const bool do_stat = p.stat_out != nullptr;
        // SHIFTED operand copy (ser_hip.h): c[m] = absolute row mean of the residual input + a load-time constant.  The
        // act copy and the row partials written below hold v - c[m], so rows whose mean is many standard deviations
        // (offsets living in the residual stream) still round to bf16 relative to their spread, and the consumer's
        // E[x^2] - mean^2 does not cancel.  LayerNorm is shift invariant: the consumer needs no change.
        float cs[TM];
#pragma unroll
        for (int mi = 0; mi < TM; ++mi) {
            cs[mi] = 0.f;
            if (p.shift_out) {
                int m = m0 + wm * TM * 16 + mi * 16 + frow;
                m = m < p.M ? m : p.M - 1;
                cs[mi] = p.shift_const + (p.shift_in ? p.shift_in[m] : 0.f);
            }
        }
        // The residual tile is requested a batch of 4 fragment rows at a time, BEFORE that batch's first store: the
        // residual may be the output buffer itself (next state, written in place), so the compiler cannot move a row's
        // loads above the previous row's stores, and the epilogue used to pay the load latency (HBM / Infinity Cache:
        // 1-2 us) once per fragment row -- four to eight times per block.  The fragment registers of the K loop are dead
        // here, so the 64 registers of a batch cost no occupancy.  (A lane reads and writes only its own elements: moving
        // its loads ahead of its stores to OTHER rows cannot change what it reads.)
        constexpr int RB = (TM * TN > 16) ? 2 : (TM < 4 ? TM : 4);          // 256x256 tile (128 accumulators): 32 registers per batch
        f32x4 rres[RB][TN];
#pragma unroll
        for (int mi = 0; mi < TM; ++mi) {
            if (mi % RB == 0 && p.residual) {
#pragma unroll
                for (int u = 0; u < RB; ++u) {
                    int mr = m0 + wm * TM * 16 + (mi + u) * 16 + frow;
                    mr = mr < p.M ? mr : p.M - 1;
                    const int rr_ = p.res_row_mod ? (mr % p.res_row_mod) : mr;
#pragma unroll
                    for (int ni = 0; ni < TN; ++ni) {
                        f32x4 r = {0.f, 0.f, 0.f, 0.f};
                        if (ncol0 + ni * 16 < p.N) r = *(const f32x4*)(p.residual + (int64_t)rr_ * p.ldr + gcol + ni * 16);
                        rres[u][ni] = r;
                    }
                }
            }
            const int m = m0 + wm * TM * 16 + mi * 16 + frow;
            if (m >= p.M) continue;
            const int64_t orow = p.out_rowmap ? (int64_t)p.out_rowmap[m] : (int64_t)m;
            float mu = 0.f, rs = 1.f;
            if (p.ln_stats_in) {
                mu = lnst[2 * (wm * TM * 16 + mi * 16 + frow)];
                rs = lnst[2 * (wm * TM * 16 + mi * 16 + frow) + 1];
            }
            const float cshift = cs[mi];
            if (p.shift_out && nt == 0 && g == 0 && wn == 0 && fq == 0) p.shift_out[m] = cshift;
            float st1 = 0.f, st2 = 0.f;
            unsigned mcode[2] = {0u, 0u};                             // FP16M out copy: (P, Q) codes of the two blocks of a 64-column tile
#pragma unroll
            for (int ni = 0; ni < TN; ni += 2) {                      // fragment column blocks ni, ni+1: 4 + 4 columns
                const bool ok0 = ncol0 + ni * 16 < p.N, ok1 = ncol0 + ni * 16 + 16 < p.N;
                float v[8];
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    const int nj = ni + (r >> 2), rr = r & 3;
                    // LN(x) W^T = rstd * (x W'^T - mu * colsum(W')) + (beta W^T + b)
                    float x = fmaf(rs, acc[nj][mi][rr] - mu * csum[nj * 4 + rr], bias[nj * 4 + rr]);
                    if (do_scale && ncol0 + nj * 16 < p.col_scale_end) x *= p.col_scale;   // e.g. q *= dh^-0.5 * log2(e)
                    v[r] = x;
                }
                if (p.act == SER_ACT_GELU) {
#pragma unroll
                    for (int r = 0; r < 8; r += 2) {
                        const f32x2 y = gelu2<MODE == SER_MODE_BF16>((f32x2){v[r], v[r + 1]});
                        v[r] = y[0]; v[r + 1] = y[1];
                    }
                }
                if (p.residual) {
                    const f32x4 r0 = rres[mi % RB][ni], r1 = rres[mi % RB][ni + 1];
                    v[0] += r0[0]; v[1] += r0[1]; v[2] += r0[2]; v[3] += r0[3];
                    v[4] += r1[0]; v[5] += r1[1]; v[6] += r1[2]; v[7] += r1[3];
                }
                if (p.out_f32) {
                    float* op = p.out_f32 + (int64_t)m * p.ldo_f32 + gcol + ni * 16 - p.f32_col_begin;
                    f32x4 o0 = {v[0], v[1], v[2], v[3]}, o1 = {v[4], v[5], v[6], v[7]};
                    if (ok0 && ncol0 + ni * 16 >= p.f32_col_begin) *(f32x4*)op = o0;
                    if (ok1 && ncol0 + ni * 16 + 16 >= p.f32_col_begin) *(f32x4*)(op + 16) = o1;
                }
                if (p.shift_out) {                                    // wave-uniform; the fp32 output above stays unshifted
#pragma unroll
                    for (int r = 0; r < 8; ++r) v[r] -= cshift;
                }
                if (do_stat && ok0) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) { st1 += v[r]; st2 += v[r] * v[r]; }
                }
                if (do_stat && ok1) {
#pragma unroll
                    for (int r = 4; r < 8; ++r) { st1 += v[r]; st2 += v[r] * v[r]; }
                }
                if (p.out_act) {
                    const int c8 = ni * 16 + ((fq & 1) ? 12 : 0);     // after the swap this lane holds columns c8..c8+7
                    if constexpr (OM == SER_MODE_FP16M) {
                        // hi plane = the fp16 copy; cross-term plane: P = v - hi, Q = v as e4m3 with one power-of-two scale per 32 columns
                        // (this fragment pair = one block: the four lanes of the row hold 8 of its columns each)
                        store_act8_swap<SER_MODE_FP16>((unsigned short*)p.out_act + orow * p.ldo_act + gcol + c8, 0, ncol0 + c8 < p.N, v);
                        float lo[8], ax = 0.f, al = 0.f;
#pragma unroll
                        for (int r = 0; r < 8; r += 2) {
                            const unsigned h2 = pack_h2(v[r], v[r + 1]);
                            lo[r] = v[r] - h2f((unsigned short)(h2 & 0xffffu));
                            lo[r + 1] = v[r + 1] - h2f((unsigned short)(h2 >> 16));
                            ax = fmaxf(ax, fmaxf(fabsf(v[r]), fabsf(v[r + 1])));
                            al = fmaxf(al, fmaxf(fabsf(lo[r]), fabsf(lo[r + 1])));
                        }
                        ramax = fmaxf(ramax, ax);
                        ax = max_rowquad(ax); al = max_rowquad(al);
                        const unsigned cx = mx_code(ax), cl = mx_code(al);
                        const float ix = mx_inv(cx), il = mx_inv(cl);
                        unsigned q4[4] = {mx_pack4(lo[0] * il, lo[1] * il, lo[2] * il, lo[3] * il), mx_pack4(lo[4] * il, lo[5] * il, lo[6] * il, lo[7] * il),
                                          mx_pack4(v[0] * ix, v[1] * ix, v[2] * ix, v[3] * ix), mx_pack4(v[4] * ix, v[5] * ix, v[6] * ix, v[7] * ix)};
                        transpose_rowquad(q4);        // lane group fq now holds 16 consecutive bytes: P cols 0-15 | P 16-31 | Q 0-15 | Q 16-31 of the block
                        const int64_t bcol = gcol - fq * 4 + ni * 16;                    // the block's first output column
                        unsigned char* seg = (unsigned char*)((unsigned short*)p.out_act + p.out_plane_stride) + (orow * p.ldo_act + (bcol & ~(int64_t)63)) * 2;
                        if (ok0) *(u32x4*)(seg + (fq >> 1) * 64 + (bcol & 63) + (fq & 1) * 16) = (u32x4){q4[0], q4[1], q4[2], q4[3]};
                        mcode[(ni >> 1) & 1] = cl | (cx << 16);
                        if (((ni >> 1) & 1) && fq == 0 && ok0)                            // second block of a 64-column tile: its scale word
                            p.out_scale[(bcol >> 6) * p.out_scale_ld + orow] = (mcode[0] & 0xffu) | ((mcode[1] & 0xffu) << 8) | (mcode[0] & 0xff0000u) | ((mcode[1] & 0xff0000u) << 8);
                    } else {
                        store_act8_swap<OM>((unsigned short*)p.out_act + orow * p.ldo_act + gcol + c8, p.out_plane_stride,
                                            ncol0 + c8 < p.N, v);
                        if constexpr (mode_traits<OM>::f16) {             // fp16 range guard (ser_hip.h range_flag)
                            if (p.range_flag) {
#pragma unroll
                                for (int r = 0; r < 8; r += 2) ramax = fmaxf(ramax, fmaxf(fabsf(v[r]), fabsf(v[r + 1])));
                            }
                        }
                    }
                }
            }
            if (p.stat_out) {
                // row partials over this wave's 64 columns (deterministic: one slot per 64-column group)
                st1 += __shfl_xor(st1, 16, 64); st2 += __shfl_xor(st2, 16, 64);
                st1 += __shfl_xor(st1, 32, 64); st2 += __shfl_xor(st2, 32, 64);
                const int cstart = n0 + wn * (TN * 16);
                const int grp = g * ((p.N + 63) >> 6) + (cstart >> 6);
                if (fq == 0 && cstart < p.N && grp < p.stat_groups) {
                    float* d = p.stat_out + ((int64_t)m * p.stat_groups + grp) * 2;
                    d[0] = st1; d[1] = st2;
                }
            }
        }
    }
    if constexpr (mode_traits<OM>::f16) range_report(p.range_flag, ramax);
#ifdef SER_GEMM_DBG
    {
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long t2 = __builtin_amdgcn_s_memtime();
        unsigned long long* d = ser_gemm_dbg_dev;
        if (d && wave == 0 && lane == 0 && blockIdx.y == 0) d[131072 + blockIdx.x * 4 + 2] = t2;
    }
#endif
}

// Tile-selection thresholds are constants in the product build; `make EXPERIMENTS=1` turns them back into the environment knobs the
// A/B scripts under tools/ set (SER_GEMM_T256_MIN, SER_GEMM_FORCE, ...).
#ifdef SER_EXPERIMENTS
#define SER_KNOB(name, dflt) ([] { const char* e_ = getenv(name); return e_ ? atol(e_) : (long)(dflt); }())
#else
#define SER_KNOB(name, dflt) ((long)(dflt))
#endif

// ------------------------------------------------------------------------------------------------
enum { CFG_128x128 = 0, CFG_256x128 = 1, CFG_256x256 = 2, CFG_LN512 = 3, CFG_LN512_M64 = 4, CFG_LN512_M32 = 5, CFG_128x64 = 6, CFG_256x256_W4 = 7 };

template <int WM, int WN, int TM, int TN, int BK, int ST, int MODE, bool LNEPI, int OM = MODE>
static hipError_t launch_mode(const ser_gemm_args* a, dim3 grid, dim3 block, int LDS, hipStream_t s) {
    auto k = ser_gemm_kernel<WM, WN, TM, TN, BK, ST, MODE, LNEPI, OM>;
    // per instantiation; the drivers launch from several host threads: an atomic flag (two threads may both make the
    // idempotent call, neither reads a half-written flag)
    static std::atomic<bool> ready{false};
    if (LDS > 65536 && !ready.load(std::memory_order_acquire)) {
        hipError_t e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        if (e != hipSuccess) return e;
        ready.store(true, std::memory_order_release);
    }
    hipLaunchKernelGGL(k, grid, block, LDS, s, *a);
    return hipSuccess;
}

// X32: the configuration serves the two-plane modes FP32X / FP16X (both planes per stage); otherwise the single-plane modes BF16 / FP16.
template <int WM, int WN, int TM, int TN, int BK, int ST, bool LNEPI, bool X32 = false>
static int launch_cfg(const ser_gemm_args* a, hipStream_t s) {
    constexpr int BM = WM * TM * 16, BN = WN * TN * 16;
    const int npl = (a->mode == SER_MODE_FP32X || a->mode == SER_MODE_FP16X) ? 2 : 1;
    const int LDS = npl * ST * (BM + BN) * BK * 2 + (LNEPI ? 0 : BM * 8)    // ring (+ [BM][2] row statistics)
                  + (a->mode == SER_MODE_FP16M ? ST * (BM + BN) * 4 : 0);   // FP16M: + the block-scale words of every stage
    const int ntm = (a->M + BM - 1) / BM, ntn = (a->N + BN - 1) / BN;
    dim3 grid((unsigned)(ntm * ntn), (unsigned)a->groups, 1), block(64 * WM * WN, 1, 1);
    hipError_t e = hipSuccess;
    if constexpr (X32) {
        if constexpr (!LNEPI) {
            if (a->mode == SER_MODE_FP16X && a->out_mode == SER_MODE_FP16M) {   // output projection of "f16m": 3 products on the attention kernel's hi + lo
                if constexpr (BN >= 128)                                         // context rows, FP16M copy for FC1
                    e = launch_mode<WM, WN, TM, TN, BK, ST, SER_MODE_FP16X, LNEPI, SER_MODE_FP16M>(a, grid, block, LDS, s);
                else return ser_fail(-22, "ser_gemm: FP16X -> FP16M output needs a dense tile (N > 64)");
            } else
            if (a->mode == SER_MODE_FP16X && a->out_mode == SER_MODE_FP16)      // output projection of "f16a": 3 products, one-plane copy for FC1
                e = launch_mode<WM, WN, TM, TN, BK, ST, SER_MODE_FP16X, LNEPI, SER_MODE_FP16>(a, grid, block, LDS, s);
            else if (a->mode == SER_MODE_FP16X)        // attention block ("f16a") / logit path ("f16q"): 3 products on the f16 MFMA
                e = launch_mode<WM, WN, TM, TN, BK, ST, SER_MODE_FP16X, LNEPI>(a, grid, block, LDS, s);
            else if (a->out_mode == SER_MODE_FP16)     // stem -> layers boundary of the "f16" numerics mode
                e = launch_mode<WM, WN, TM, TN, BK, ST, SER_MODE_FP32X, LNEPI, SER_MODE_FP16>(a, grid, block, LDS, s);
            else
                e = launch_mode<WM, WN, TM, TN, BK, ST, SER_MODE_FP32X, LNEPI>(a, grid, block, LDS, s);
        } else {
            if (a->mode == SER_MODE_FP16X)             // conv stack of the f16 / f16q / f16a modes: fp16 hi + lo planes (22-bit operands)
                e = launch_mode<WM, WN, TM, TN, BK, ST, SER_MODE_FP16X, LNEPI>(a, grid, block, LDS, s);
            else
                e = launch_mode<WM, WN, TM, TN, BK, ST, SER_MODE_FP32X, LNEPI>(a, grid, block, LDS, s);
        }
    } else {
        if (a->mode == SER_MODE_FP16M) {
            if constexpr (!LNEPI && BN >= 128 && BK == 64) {
                if (a->out_mode == SER_MODE_FP16X)     // packed projection of "f16m": q, k, v leave as fp16 hi + lo planes for ser_attention
                    e = launch_mode<WM, WN, TM, TN, BK, ST, SER_MODE_FP16M, LNEPI, SER_MODE_FP16X>(a, grid, block, LDS, s);
                else
                    e = launch_mode<WM, WN, TM, TN, BK, ST, SER_MODE_FP16M, LNEPI>(a, grid, block, LDS, s);
            } else return ser_fail(-22, "ser_gemm: FP16M needs a dense tile (N > 64, no LayerNorm epilogue)");
        } else
        if (a->mode == SER_MODE_FP16) {
            if constexpr (!LNEPI && BN >= 128) {
                if (a->out_mode == SER_MODE_FP16X)     // FC2 of the "f16q" mode: the next layer's q / k projection reads hi + lo planes
                    e = launch_mode<WM, WN, TM, TN, BK, ST, SER_MODE_FP16, LNEPI, SER_MODE_FP16X>(a, grid, block, LDS, s);
                else
                    e = launch_mode<WM, WN, TM, TN, BK, ST, SER_MODE_FP16, LNEPI>(a, grid, block, LDS, s);
            } else {
                if (a->out_mode == SER_MODE_FP16X) return ser_fail(-22, "ser_gemm: FP16 -> FP16X output needs a dense tile (N > 64, no LayerNorm epilogue)");
                e = launch_mode<WM, WN, TM, TN, BK, ST, SER_MODE_FP16, LNEPI>(a, grid, block, LDS, s);
            }
        }
        else e = launch_mode<WM, WN, TM, TN, BK, ST, SER_MODE_BF16, LNEPI>(a, grid, block, LDS, s);
    }
    if (e != hipSuccess) return ser_fail((int)e, "ser_gemm: cannot raise dynamic LDS to %d", LDS);
    return ser_check_launch("ser_gemm");
}

static int pick_cfg(const ser_gemm_args* a) {
    if (a->ln_gamma) {
        // Row-complete LayerNorm tiles are BM x 512.  The last conv layers have few rows (M = 16k / 8k / 4k for eight
        // 10 s utterances): 128-row tiles would leave half to 7/8 of the CUs idle, so the tile gets shorter until
        // the grid covers the chip (measured per layer with tools/gemm_by_layer.py).
        static const int force_bm = (int)SER_KNOB("SER_GEMM_LN_BM", 0);
        const int bm = force_bm ? force_bm : (a->M >= 200 * 128 ? 128 : (a->M >= 200 * 64 ? 64 : 32));
        return bm == 128 ? CFG_LN512 : (bm == 64 ? CFG_LN512_M64 : CFG_LN512_M32);
    }
#ifdef SER_EXPERIMENTS
    if (a->tile_cfg == 4) return CFG_256x256_W4;
#endif
    if (a->tile_cfg > 0) return a->tile_cfg - 1;
    // Measured on MI355X (tools/gemm_sweep.py, M = 7984): the simple ring keeps the 128x128 tile
    // (2 blocks/CU) ahead of 256x128 on every N <= 3072 shape; 256x256 wins once it has >= ~2 full
    // rounds of blocks (N = 4096, conv layers) because it halves the L2->LDS bytes per FLOP.
    // In the real step (two utterance groups in flight) 256x256 already pays from ~200 tiles (QKV and FC1
    // at M = 3992): 9.6 -> 9.1 ms per step, A/B on one device.
#ifdef SER_EXPERIMENTS
    {   // experiments: SER_GEMM_FORCE="N:K:cfg[,N:K:cfg...]" pins the tile config of matching launches (tools/)
        struct Rule { int n, k, cfg; };
        static Rule rules[8];
        static const int nrules = [] {
            const char* e = getenv("SER_GEMM_FORCE");
            int n = 0;
            while (e && *e && n < 8) {
                int a0, a1, a2, used = 0;
                if (sscanf(e, "%d:%d:%d%n", &a0, &a1, &a2, &used) != 3) break;
                rules[n++] = {a0, a1, a2};
                e += used;
                if (*e == ',') ++e;
            }
            return n;
        }();
        for (int i = 0; i < nrules; ++i)
            if (rules[i].n == a->N && rules[i].k == a->K && rules[i].cfg >= 0 && rules[i].cfg <= CFG_256x256) return rules[i].cfg;
    }
#endif
    const long t256x256 = (long)((a->M + 255) / 256) * ((a->N + 255) / 256) * a->groups;
    static const long t256_min = SER_KNOB("SER_GEMM_T256_MIN", 150);      // 200 before the ping-pong schedule; XLS-R-2B's QKV (184 tiles at 4 x 10 s): +3.2 % on its step
    if (a->N >= 256 && t256x256 >= t256_min) return CFG_256x256;
    // grouped positional conv: 64 output channels per group -> a 128x64 tile wastes no MFMA columns
    static const int n64 = (int)SER_KNOB("SER_GEMM_N64", 1);
    if (n64 && a->N <= 64) return CFG_128x64;
    // Deep-K, narrow-N GEMMs (FC2: N = D, K = 4D) that are too small for the 256x256 tile: 256x128 tiles are SLOWER
    // in isolation (52 -> 64 us at M = 3992: only 128 blocks) but +1.5 % on the real step in five A/B pairs -- the
    // launch then occupies half the CUs for its whole (long) K loop and the other utterance group's kernels own the
    // other half, instead of both time-slicing every CU.  The shallow out-projection (K = D) loses with it.
    static const int deepk = (int)SER_KNOB("SER_GEMM_DEEPK_256x128", 1);
    if (deepk && a->groups == 1 && a->K >= 2048 && a->N >= 128 && a->M >= 512) return CFG_256x128;
    return CFG_128x128;
}

extern "C" int ser_gemm(const ser_gemm_args* a, void* stream) {
    if (!a || !a->A || !a->W) return ser_fail(-1, "ser_gemm: null operand");
    if (a->M <= 0 || a->N <= 0 || a->K <= 0) return ser_fail(-2, "ser_gemm: bad shape M=%d N=%d K=%d", a->M, a->N, a->K);
    if (a->K % 64) return ser_fail(-3, "ser_gemm: K=%d must be a multiple of 64", a->K);
    if (a->kc && (a->kc % 64 || a->K % a->kc)) return ser_fail(-4, "ser_gemm: kc=%d must divide K and be a multiple of 64", a->kc);
    if (a->N % 8) return ser_fail(-5, "ser_gemm: N=%d must be a multiple of 8", a->N);
    if (a->mode != SER_MODE_BF16 && a->mode != SER_MODE_FP32X && a->mode != SER_MODE_FP16 && a->mode != SER_MODE_FP16X && a->mode != SER_MODE_FP16M)
        return ser_fail(-6, "ser_gemm: bad mode %d", a->mode);
    if (a->mode == SER_MODE_FP16M) {
        if (!a->a_scale || !a->w_scale || a->a_scale_ld < a->M || a->w_scale_ld < a->N)
            return ser_fail(-23, "ser_gemm: FP16M needs a_scale / w_scale with *_scale_ld >= the row count");
        if (a->a_rowoff || a->kc || a->groups != 1 || a->ln_gamma || (a->lda % 64))
            return ser_fail(-23, "ser_gemm: FP16M takes plain row-major operands (no row map / conv chunks / groups / LayerNorm epilogue), lda %% 64 == 0");
    }
    if (a->out_mode == SER_MODE_FP16M || (a->mode == SER_MODE_FP16M && !a->out_mode && a->out_act)) {
        if (!a->out_scale || (a->N % 64) || (a->ldo_act % 64) || a->groups != 1 || a->ln_gamma)
            return ser_fail(-24, "ser_gemm: an FP16M out_act needs out_scale, N %% 64 == 0, ldo_act %% 64 == 0, groups == 1, no LayerNorm epilogue");
    }
    if (!a->a_rowoff && (a->lda % 8)) return ser_fail(-7, "ser_gemm: lda must be a multiple of 8");
    if (a->groups < 1) return ser_fail(-8, "ser_gemm: groups=%d", a->groups);
    if (!a->out_f32 && !a->out_act) return ser_fail(-9, "ser_gemm: no output");
    if ((a->ldo_f32 % 4) || (a->ldo_act % 4) || (a->ldr % 4) || (a->c_group_stride % 4))
        return ser_fail(-10, "ser_gemm: output/residual pitches must be multiples of 4");
    if (a->ln_gamma) {
        if (!a->ln_beta) return ser_fail(-11, "ser_gemm: ln_gamma without ln_beta");
        if (a->N > 512 || a->groups != 1 || a->residual)
            return ser_fail(-12, "ser_gemm: the LayerNorm epilogue needs N <= 512, groups == 1, no residual");
    }
    if (a->ln_stats_in) {
        if (!a->ln_colsum || a->ln_groups < 2 || a->ln_groups > 32 || (a->ln_groups & 1) || a->ln_gamma)
            return ser_fail(-14, "ser_gemm: deferred LayerNorm needs ln_colsum, an even ln_groups in [2,32] and no fused-LN epilogue");
    }
    if (a->stat_out && (a->ln_gamma || a->stat_groups < a->groups * ((a->N + 63) / 64)))
        return ser_fail(-15, "ser_gemm: stat_out needs stat_groups >= groups*ceil(N/64) and no fused-LN epilogue");
    if (a->shift_out && (!a->stat_out || a->ln_gamma)) return ser_fail(-19, "ser_gemm: shift_out needs stat_out and no fused-LN epilogue");
    if (a->mean_out && !a->ln_stats_in) return ser_fail(-20, "ser_gemm: mean_out needs ln_stats_in");
    if (a->lnstat_out && !a->ln_stats_in) return ser_fail(-20, "ser_gemm: lnstat_out needs ln_stats_in");
    if (a->out_mode && a->out_mode != a->mode && !a->ln_gamma &&
        !((a->mode == SER_MODE_FP32X && a->out_mode == SER_MODE_FP16) || (a->mode == SER_MODE_FP16 && a->out_mode == SER_MODE_FP16X) ||
          (a->mode == SER_MODE_FP16X && a->out_mode == SER_MODE_FP16) || (a->mode == SER_MODE_FP16X && a->out_mode == SER_MODE_FP16M) ||
          (a->mode == SER_MODE_FP16M && a->out_mode == SER_MODE_FP16X)))
        return ser_fail(-21, "ser_gemm: out_mode %d with mode %d (FP32X -> FP16, FP16 <-> FP16X, FP16X <-> FP16M convert)", a->out_mode, a->mode);
    if (a->out_mode && a->out_mode != a->mode && a->ln_gamma)
        return ser_fail(-21, "ser_gemm: out_mode %d with the LayerNorm epilogue", a->out_mode);
    if (a->col_scale_end % 4) return ser_fail(-17, "ser_gemm: col_scale_end must be a multiple of 4");
    if ((a->ldo_act % 8) || (a->c_group_stride % 8)) return ser_fail(-18, "ser_gemm: act pitch / group stride must be multiples of 8");
    if (a->f32_col_begin < 0 || (a->f32_col_begin % 8)) return ser_fail(-16, "ser_gemm: f32_col_begin must be a non-negative multiple of 8");
#ifdef SER_EXPERIMENTS
    if (a->tile_cfg < 0 || a->tile_cfg > 4) return ser_fail(-13, "ser_gemm: tile_cfg=%d (0 auto, 1..4)", a->tile_cfg);
#else
    if (a->tile_cfg < 0 || a->tile_cfg > 3) return ser_fail(-13, "ser_gemm: tile_cfg=%d (0 auto, 1..3)", a->tile_cfg);
#endif
    hipStream_t s = (hipStream_t)stream;
#ifdef SER_GEMM_DBG
    {
        static void* planted = nullptr;
        if (planted != ser_gemm_dbg_ptr) {
            planted = ser_gemm_dbg_ptr;
            (void)hipMemcpyToSymbol(HIP_SYMBOL(ser_gemm_dbg_dev), &planted, sizeof(void*));
        }
    }
#endif
    if (a->mode == SER_MODE_FP32X || a->mode == SER_MODE_FP16X) {
        // both planes share a stage: the ring doubles, so FP32X uses the two configurations that still fit 160 KiB
        if (a->ln_gamma) return launch_cfg<2, 4, 4, 8, 32, 2, true, true>(a, s);
        // grouped positional conv (<= 64 output channels per group): 128x64 tiles of 32x64 wave tiles, like the bf16 path's -- the
        // 128x128 tile below computes 64 dead columns per group (fp32x pos-conv: 360 us against 100 us in bf16)
        static const int x32_n64 = (int)SER_KNOB("SER_GEMM_N64", 1);
        if (x32_n64 && a->N <= 64) return launch_cfg<4, 1, 2, 4, 32, 2, false, true>(a, s);
        // Large grids: 256x128 tiles of 64x64 wave tiles on a 32-deep, 3-stage ring (144 KiB): 16 fragments feed 48 MFMAs per
        // k-step (0.33 LDS fragment reads per MFMA against 0.5 for the 32x64 wave tile below), one ping-pong phase per K tile
        static const long x32_256_min = SER_KNOB("SER_GEMM_X32_256_MIN", 100);   // 100: M = 3992 out-proj / FC2 (128 tiles) gain, M = 1996 ones (64 tiles) lose
        const long t256x128 = (long)((a->M + 255) / 256) * ((a->N + 127) / 128) * a->groups;
        // Largest grids (round 3): 256x256 tiles of 64x128 wave tiles, both planes of a 32-deep K tile per stage, 2 stages (128 KiB):
        // 24 fragments feed 96 MFMAs per K tile (0.25 LDS fragment reads per MFMA, half the L2 -> LDS bytes per product of the
        // 256x128 tile), the weight fragments taken in two halves like the LayerNorm tile's (PPW).  From SER_GEMM_X32_SQ_MIN tiles.
        static const long x32_sq_min = SER_KNOB("SER_GEMM_X32_SQ_MIN", 150);
        const long t256sq = (long)((a->M + 255) / 256) * ((a->N + 255) / 256) * a->groups;
        // Measured on the step (two A/B pairs, one box): the packed QKV projection on it f16a 1 110 / 1 114 -> 1 134 / 1 130 utt/s; FC1 too
        // (its GELU epilogue on 128 accumulators + 64 spilled bias / column-sum registers) gives the gain back: fp32x 848 -> 846.
        // Hence only launches without an activation take it.
        // ... and only grids whose last round of 256 blocks is mostly full (round 4): WavLM-large's packed projection has 384 such tiles = 1.5 rounds
        // (efficiency 0.75) and is 0.5 % faster on the f16x step as 768 tiles of 256x128 = 3 full rounds (XLS-R-2B at 8 x 10 s, 368 tiles: + 0.2 %), while
        // HuBERT-xlarge's 480 (0.94) and Whisper's 705 (0.92) lose 2.7 % / 4.7 % of their steps without the square tile (same-box pairs, experiments build).
        const long sq_rounds = (t256sq + 255) / 256;
        const bool sq_full = t256sq * 100 >= sq_rounds * 256 * 85;
        if (x32_sq_min > 0 && a->N >= 256 && t256sq >= x32_sq_min && sq_full && a->act == SER_ACT_NONE) return launch_cfg<4, 2, 4, 8, 32, 2, false, true>(a, s);
        if (x32_256_min > 0 && a->N >= 128 && t256x128 >= x32_256_min) return launch_cfg<4, 2, 4, 4, 32, 3, false, true>(a, s);
        return launch_cfg<4, 2, 2, 4, 64, 2, false, true>(a, s);      // 128x128 tile on 8 waves (32x64 each): 2 waves/SIMD hide the LDS reads
    }
    if (a->mode == SER_MODE_FP16M) {
        // the single-plane tiles walked twice per K tile (H unit, E unit): 256x256 from 100 tiles up, 256x128 on a three-stage ring, 128x128 on four
        // waves for small M.  Alone, one launch at a time, the 256x256 form is the SLOWER one at M = 7 984 (packed projection 139 us against 122 on
        // 256x128, FC2 -- 128 tiles, half the chip -- 206 against 130: profiles/r05_gemm_f16m_bench.txt); on the step it is the FASTER one: 928 ->
        // 973 utt/s with the packed projection and FC1 on it, 990 with FC2 and the output projection too (profiles/r05_f16m_tile_choice_step_ab.txt).
        // The step runs at the board's power cap (profiles/r05_power_sample_bf16_step.txt): what counts is energy per FLOP, and the square tile
        // moves 2/3 of the L2 -> LDS bytes per FLOP.  (The same threshold on the bf16 tiles LOSES 1.3 %: T256_MIN stays 150.)
        const long t256sq = (long)((a->M + 255) / 256) * ((a->N + 255) / 256);
        const long t256x128 = (long)((a->M + 255) / 256) * ((a->N + 127) / 128);
        static const long m_sq_min = SER_KNOB("SER_GEMM_M16_SQ_MIN", 100), m_256_min = SER_KNOB("SER_GEMM_M16_256_MIN", 100);
        if (a->tile_cfg == 3 || (!a->tile_cfg && a->N >= 256 && t256sq >= m_sq_min)) return launch_cfg<2, 4, 8, 4, 64, 2, false>(a, s);
        if (a->tile_cfg == 2 || (!a->tile_cfg && a->N >= 128 && t256x128 >= m_256_min)) return launch_cfg<4, 2, 4, 4, 64, 3, false>(a, s);
        return launch_cfg<2, 2, 4, 4, 64, 2, false>(a, s);
    }
    switch (pick_cfg(a)) {
        case CFG_LN512:   return launch_cfg<2, 4, 4, 8, 64, 2, true>(a, s);     // 160 KiB ring, one barrier per 64-deep K tile
        case CFG_LN512_M64: return launch_cfg<1, 8, 4, 4, 64, 2, true>(a, s);   // 64 x 512 tile, 8 waves of 64x64
        case CFG_LN512_M32: return launch_cfg<1, 4, 2, 8, 64, 2, true>(a, s);   // 32 x 512 tile, 4 waves of 32x128
        case CFG_128x64:  return launch_cfg<4, 1, 2, 4, 64, 2, false>(a, s);    // 4 waves of 32x64: a wave owns a whole 64-column stat group
        case CFG_256x256: return launch_cfg<2, 4, 8, 4, 64, 2, false>(a, s);
#ifdef SER_EXPERIMENTS
        // four waves of 128x128, one per SIMD, hand-placed software pipeline (round 5): its K loop runs at 89 % of the matrix pipe (FC2 shape: 1.16 us per
        // 64-deep K tile against ~1.8) but the epilogue of a lone wave per SIMD costs more than the loop saves (QKV 75 against 70 us, FC1 84 / 80);
        // the same pipeline on the 8-wave tiles (-DSER_GEMM_PSW) is 8 % / 3.5 % faster on QKV / FC1 alone and moves the step by nothing: the step
        // runs AT THE BOARD'S 1 400 W POWER CAP (profiles/r05_power_sample_bf16_step.txt), cycle savings come back as clock.  make EXPERIMENTS=1 only.
        case CFG_256x256_W4: return launch_cfg<2, 2, 8, 8, 64, 2, false>(a, s);
#endif
        case CFG_256x128: return launch_cfg<4, 2, 4, 4, 64, (SER_GEMM_PSW & 2) ? 2 : 3, false>(a, s);
        default:          return launch_cfg<2, 2, 4, 4, 64, 2, false>(a, s);
    }
}
