// ser_attention, resident form: K and V of one (utterance, head) live in LDS for the whole block (round 4).
//
// The tiled kernel (attention.hip) gives every 128 queries of an (utterance, head) pair their own block: at T = 499 that is four
// blocks per pair, each of which requests, stages and synchronises on all eight 64-key K/V tiles again, and each of which pays the
// prologue latency chain (frame offsets -> Q / K / V / bias row: 3.2 of a block's 16 us, profiles/r03_attn_phases_bf16.txt).  Up to
// T = 512 frames (10.2 s of audio; the SER corpora's utterances and BASELINE configs[1]'s 10 s clips) K and V of one head are
// 2 x T x 128 B <= 128 KiB and the head's 2T - 1 bias distances x 4 shifted copies <= 17 KiB: both fit the CU's 160 KiB of LDS.
//
//   * ONE block of 4 waves per (utterance, head) -- one wave per SIMD, the whole 512-register file each: 256 blocks for a batch of
//     16 x 16 heads = one per CU, one round.
//   * K/V arrive once, by LDS-DMA (global_load_lds_dwordx4: no register round trip, 32 pieces of 1 KiB per wave issued up front in
//     tile order); the XOR swizzles of the K rows / V units are applied on the per-lane SOURCE address (the LDS destination of a
//     DMA piece is lane-linear).
//   * a wave owns 64 consecutive queries at a time = TWO 32-query blocks that it walks through the key tiles TOGETHER: the K / V
//     fragments of a tile are read from LDS once for both, and the two blocks are independent dependency chains in one instruction
//     stream, so one block's MFMAs issue beside the other's softmax arithmetic.  (Why not two waves per SIMD: stamped with
//     tools/attn_res_phases.py, two co-resident waves of this loop -- in lockstep, half a tile apart behind barriers, or free-running --
//     take 3 200 ticks for a tile each against 1 700 for a wave that has its SIMD alone: no overlap between waves to speak of; the
//     overlap has to be inside the wave.  DESIGN.md section 10 keeps those stamps.)
//   * the first 64 queries of a wave are computed while the tiles land -- per tile one COUNTED s_waitcnt vmcnt(4 x tiles still in
//     flight) + s_barrier --, the later ones run on resident tiles with no barrier, no staging and no waits.
//   * per (query, key tile) the arithmetic is the tiled kernel's (S^T = K Q^T with the key index in the accumulator registers, online
//     softmax with a stale running maximum, O^T = V^T P^T from the accumulator registers, ds_read_b64_tr_b16 for V^T); checked
//     against the same fp64 reference (tests/test_gpu_kernels.py::test_attention_resident_form ...).
//   * Q fragments of every query block of the wave, and the WavLM gate of its queries, are requested in the prologue beside the DMA.
//
// Single-plane modes (bf16, fp16), head dim 64, pre-scaled q, with or without the WavLM bias table.  Everything else -- longer
// utterances, other head dims, the two-plane modes, key padding, dense biases -- stays on the tiled kernel.
#include "attn_common.h"
#include <atomic>
#include <type_traits>

namespace {

constexpr int RW = 4;                       // waves per block
constexpr int RNT = 64 * RW;
constexpr int RS = 128;                     // LDS row bytes (64 x 16-bit)
constexpr int TILE = ABKV * RS;             // one 64-key K (or V) tile
constexpr int U = 2;                        // 32-query blocks a wave walks through the key tiles together
constexpr int MAXJ = 2;                     // rounds of U blocks per wave: T <= 512 -> 16 blocks of 32 = 4 waves x 2 rounds x 2
constexpr int MAXT = 512;
constexpr int NTILES = MAXT / ABKV;

#ifdef SER_ATTN_DBG
// one s_memtime stamp per (wave, slot) of block 100: tools/attn_res_phases.py (debug build only)
#define RDBG(slot) do { __builtin_amdgcn_sched_barrier(0); if (p.dbg && blockIdx.x == 100 && lane == 0) p.dbg[wave * 64 + (slot)] = __builtin_amdgcn_s_memtime(); \
                        __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define RDBG(slot) do {} while (0)
#endif

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }

// TBL: WavLM bias table + gate; GX: the gate is computed here from the layer input's operand copy (ser_attention_args.gate_x), else read from gate[]
template <int MODE, bool TBL, bool GX>
__global__ __launch_bounds__(RNT, 2)          // (2: at most 256 registers, all addressable by the vector ALU -- with 1 hipcc puts the MFMA accumulators in the AGPR half and copies)
void attention_resident_kernel(const AttnParams p) {
    constexpr int KS = 4, DSUB = 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    RDBG(0);
    const int bh = blockIdx.x / p.nq, split = blockIdx.x - bh * p.nq;         // nq = blocks per (utterance, head): 64-query groups dealt round-robin
    const int h = bh % p.H, b = bh / p.H;
    const int row0 = p.frame_offs[b];
    const int T = p.frame_offs[b + 1] - row0;
    const int nkt = (T + ABKV - 1) / ABKV;
    const int nktm = p.nitems;                                                // tiles the LDS images were sized for
    char* ldsK = smem;
    char* ldsV = smem + nktm * TILE;
    float* ldsB = (float*)(smem + 2 * nktm * TILE);
    const int hh = lane >> 5, l31 = lane & 31;
    if (T <= 0) return;
    RDBG(1);

    // ---- bias window of the whole utterance, requested FIRST (vector-memory data returns in issue order): copy c [j] = row[j + c + jmin].
    // Unconditional loads from clamped indices, zeroed by select: a guarded load is a branch with a vmcnt(0) behind it in hipcc's
    // output, and the first build of this kernel spent 3 - 7 us of its prologue in seven of them per thread.  Two passes of the
    // 256 threads cover the <= 1 100 floats of a copy.
    const int jmin = -((3 - (T - 1)) & 3);                                    // T-1-jmin == 3 (mod 4): a quad of queries reads one aligned offset
    float bw[2][8];
    if (TBL) {
        const float* slice = p.table + (int64_t)h * (2 * p.table_T - 1) + (p.table_T - T);      // the 2T-1 distances of this utterance length
#pragma unroll
        for (int ps = 0; ps < 2; ++ps) {
            const int j4 = 4 * (tid + ps * RNT);
#pragma unroll
            for (int e = 0; e < 7; ++e) {
                const int idx = j4 + e + jmin;                                  // index into the slice; may be -3..-1 or >= 2T-1
                const int ic = idx < 0 ? 0 : (idx > 2 * T - 2 ? 2 * T - 2 : idx);
                const float v = slice[ic];
                bw[ps][e] = (idx == ic) ? v : 0.f;
            }
            bw[ps][7] = 0.f;
        }
    }

    // ---- Q fragments (and the gate inputs) of this wave's query blocks: unconditional loads from clamped rows
    int qbase[MAXJ][U];
    bool act[MAXJ];                                                           // the round has at least one real query (wave-uniform)
    bf16x8 qf[MAXJ][U][KS];
    float gq2[MAXJ][U];
    float g_in0[MAXJ][U];
    bf16x8 gxf[MAXJ][U][KS], gwf[KS];
    f32x2 gst[MAXJ][U];
    float g_c = 0.f;
    f32x4 gcb = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < MAXJ; ++j) {
        act[j] = ((j * RW + wave) * p.nq + split) * (32 * U) < T;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            qbase[j][u] = ((j * RW + wave) * p.nq + split) * (32 * U) + 32 * u;
            gq2[j][u] = 0.f;
            g_in0[j][u] = 0.f;
            const int q = qbase[j][u] + l31, qc = q < T ? q : T - 1;
            const unsigned short* qrow = p.qkv + (int64_t)(row0 + qc) * p.ld + p.q_col + h * 64;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
                qf[j][u][ks] = __builtin_bit_cast(bf16x8, *(const u32x4*)(qrow + ks * 16 + hh * 8));
            if (TBL && GX) {
                const unsigned short* xr = p.gx + (int64_t)(row0 + qc) * p.gx_ld + h * 64;
#pragma unroll
                for (int ks = 0; ks < KS; ++ks)
                    gxf[j][u][ks] = __builtin_bit_cast(bf16x8, *(const u32x4*)(xr + ks * 16 + hh * 8));
                gst[j][u] = *(const f32x2*)(p.gstat + 2 * (int64_t)(row0 + qc));
            } else if (TBL) {
                g_in0[j][u] = p.gate[(int64_t)(row0 + qc) * p.H + h];
            }
        }
    }
    if (TBL && GX) {
        const unsigned short* wr = p.gw + ((int64_t)h * 2 + (l31 & 1)) * 64;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) gwf[ks] = __builtin_bit_cast(bf16x8, *(const u32x4*)(wr + ks * 16 + hh * 8));
        g_c = p.gru_const[h];
        gcb = *(const f32x4*)(p.gcb + 4 * h);
    }

    // ---- K / V by LDS-DMA, in tile order: wave w fills rows 16w .. 16w+15 of every tile (two 1 KiB pieces each for K and V).
    // The pieces are issued from inline asm, so hipcc does not know them: ROCm 7.2's waitcnt pass treats a global_load_lds as a pending
    // FLAT access and answers every later register dependency -- the Q fragments' first MFMA, every ds_read's consumer -- with
    // s_waitcnt vmcnt(0) lgkmcnt(0), i.e. "all 128 KiB have landed" before tile 0 may start (measured in the first build of this
    // kernel).  What hipcc does know, the register loads above, is drained by the one modelled wait below, which also covers tile 0's
    // four pieces; after it the only outstanding vector-memory operations are the 28 pieces of tiles 1 .. 7, ALWAYS 28 whatever the
    // utterance's tile count (tiles past its last re-request that last tile -- same bytes, an L2 hit -- into a spare slot, or into
    // its own slot when the images have no spare one): tile kt has landed when all but the 4 x (7 - kt) youngest are done.
    // Per-lane source offsets do not depend on the tile (64 rows further is the same swizzle phase): one pointer pair per piece, advanced
    // by a constant; only the utterance's last tile clamps its rows.
    const unsigned short* gsrc[2][2];                                          // [piece][K, V] of tile 0
    const int64_t tile_step = (int64_t)ABKV * p.ld;
    const unsigned short* glast[2][2];                                         // the same for the utterance's last tile (rows clamped to T - 1)
#pragma unroll
    for (int pc = 0; pc < 2; ++pc) {
        const int kr = wave * 16 + pc * 8 + (lane >> 3), cp = lane & 7;        // row within the tile, destination chunk
        const int ck = cp ^ ((kr >> 1) & 7);                                   // k_swz is an involution: the chunk that belongs at slot cp
        const int cv = (((cp >> 2) ^ ((kr >> 1) & 1)) << 2) | (cp & 3);        // v_unit_swz on the 64-byte unit, 16-byte sub-chunk kept
        const int k0 = kr < T ? kr : T - 1;
        const int kl = (nkt - 1) * ABKV + kr < T ? (nkt - 1) * ABKV + kr : T - 1;
        const unsigned short* r0 = p.qkv + (int64_t)(row0 + k0) * p.ld + h * 64;
        const unsigned short* rl = p.qkv + (int64_t)(row0 + kl) * p.ld + h * 64;
        gsrc[pc][0] = r0 + p.k_col + ck * 8;  gsrc[pc][1] = r0 + p.v_col + cv * 8;
        glast[pc][0] = rl + p.k_col + ck * 8; glast[pc][1] = rl + p.v_col + cv * 8;
    }
    auto dma_tile = [&](int t) {
        const bool inner = t < nkt - 1;                                        // wave-uniform
        const int td = t < nktm ? t : nktm - 1;
#pragma unroll
        for (int pc = 0; pc < 2; ++pc) {
            const unsigned short* gk = inner ? gsrc[pc][0] + t * tile_step : glast[pc][0];
            const unsigned short* gv = inner ? gsrc[pc][1] + t * tile_step : glast[pc][1];
            const unsigned dk = (unsigned)(uintptr_t)(ldsK + td * TILE + (wave * 2 + pc) * 1024);   // wave-uniform LDS byte address (the hardware adds lane x 16)
            const unsigned dv = (unsigned)(uintptr_t)(ldsV + td * TILE + (wave * 2 + pc) * 1024);
            unsigned keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\t"
                         "s_mov_b32 m0, %4\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, off\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(gk), "v"(gv), "s"(__builtin_amdgcn_readfirstlane(dk)), "s"(__builtin_amdgcn_readfirstlane(dv)) : "memory");
        }
    };
    __builtin_amdgcn_sched_barrier(0);                                         // every register load above is REQUESTED before the first piece
    RDBG(2);
    dma_tile(0);
    __builtin_amdgcn_s_waitcnt(0x0F70);                                        // vmcnt(0) (expcnt / lgkmcnt untouched): registers + tile 0
    __builtin_amdgcn_sched_barrier(0);
    RDBG(3);
#pragma unroll
    for (int t = 1; t < NTILES; ++t) dma_tile(t);
    __builtin_amdgcn_sched_barrier(0);
    RDBG(4);

    // ---- the four shifted bias copies
    if (TBL) {
#pragma unroll
        for (int ps = 0; ps < 2; ++ps) {
            const int j4 = 4 * (tid + ps * RNT);
            if (j4 < p.bias_stride) {
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    *(f32x4*)(ldsB + c * p.bias_stride + j4) = (f32x4){bw[ps][c], bw[ps][c + 1], bw[ps][c + 2], bw[ps][c + 3]};
            }
        }
    }

    // ---- gates: gq2 = (a (b const - 1) + 2) log2(e) per query (HF modeling_wavlm.py:167-180), as in the tiled kernel: one MFMA chain per
    // query block with the head's two LayerNorm-folded weight rows in the place of the keys, the deferred LayerNorm in closed form
    if (TBL) {
#pragma unroll
        for (int j = 0; j < MAXJ; ++j)
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (GX) {
                    f32x16 ga;
#pragma unroll
                    for (int r = 0; r < 16; ++r) ga[r] = 0.f;
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks) ga = mfma32<MODE>(gwf[ks], gxf[j][u][ks], ga);
                    const float pa = fmaf(gst[j][u][1], ga[0] - gst[j][u][0] * gcb[0], gcb[2]);
                    const float pb = fmaf(gst[j][u][1], ga[1] - gst[j][u][0] * gcb[1], gcb[3]);
                    const float ga_ = __builtin_amdgcn_rcpf(1.f + __expf(-pa)), gb_ = __builtin_amdgcn_rcpf(1.f + __expf(-pb));
                    gq2[j][u] = (ga_ * (gb_ * g_c - 1.f) + 2.f) * LOG2E;
                } else {
                    gq2[j][u] = g_in0[j][u] * LOG2E;
                }
            }
    }

    const int vg = lane >> 4, vqq = (lane >> 2) & 3, vpp = lane & 3;
    constexpr bool LAZY = TBL;                                                 // stale running maximum inside the accumulators' start value
    constexpr float LAZY_T = 8.0f;

    auto wait_tile = [&](int kt) {            // this wave's four pieces of tile kt have landed (the prologue's wait covered tile 0)
        switch (kt) {
            case 0: break;
            case 1: wait_vm<24>(); break;
            case 2: wait_vm<20>(); break;
            case 3: wait_vm<16>(); break;
            case 4: wait_vm<12>(); break;
            case 5: wait_vm<8>(); break;
            case 6: wait_vm<4>(); break;
            default: wait_vm<0>(); break;
        }
    };

    RDBG(5);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                         // the bias copies written above
    // rounds of U query blocks: every wave of the block walks the first round's barriers, active or not
#pragma unroll
    for (int j = 0; j < MAXJ; ++j) {                                           // (unrolled: j indexes register arrays)
        if (j > 0 && !act[j]) break;
        const bool on = act[j];
        int qq[U], qc[U];
        const float* bcopy[U];
        f32x16 ot[U][DSUB];
        float m_run[U], l_run[U], gq[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            qq[u] = qbase[j][u] + l31; qc[u] = qq[u] < T ? qq[u] : T - 1;
            const int bsh = (T - 1 - qc[u] - jmin) & 3;
            bcopy[u] = ldsB + bsh * p.bias_stride + ((T - 1 - qc[u] - jmin) - bsh);
#pragma unroll
            for (int i = 0; i < DSUB; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) ot[u][i][r] = 0.f;
            m_run[u] = LAZY ? 0.f : -1e30f; l_run[u] = 0.f;
            gq[u] = gq2[j][u];
        }

        // One HALF key tile (32 keys) for the wave's U query blocks: the S accumulators of both blocks are 32 registers, so the whole
        // state (O 64, S 32, Q 32, K / V fragments 32) stays inside the 256 registers the vector ALU can address -- with 64-key steps hipcc
        // parked part of it in the accumulator file and moved 192 registers per tile back and forth (v_accvgpr_read / write).
        // Plain scalar fp32 arithmetic on purpose (and the file is built with -fno-slp-vectorize): a v_pk_add_f32 / v_pk_fma_f32 beside
        // MFMAs costs ~20 cycles more than the two plain instructions it replaces (guide, cycle-constants table, "packed f32 VALU").
        auto half = [&](int kh, auto ragged_tag) {                             // kh = half-tile index: keys 32 kh .. 32 kh + 31
            constexpr bool RAGGED = decltype(ragged_tag)::value;
            const char* tK = ldsK + kh * (TILE / 2);                           // (a 64-key tile is two consecutive 32-key halves)
            const char* tV = ldsV + kh * (TILE / 2);
            f32x16 st[U];
            bf16x8 kf[KS];
#define HDBG(slot) do { if (j == 1 && kh == 4) RDBG(slot); } while (0)
            HDBG(40);
            {
                const int key = l31;                                           // swizzle phase: ((key >> 1) & 7) is the same in both halves of a tile
#pragma unroll
                for (int ks = 0; ks < KS; ++ks)
                    kf[ks] = *(const bf16x8*)(tK + key * RS + (k_swz<64>(key, ks * 2 + hh) << 4));
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (TBL) {
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4) {
                        const f32x4 bv = *(const f32x4*)(bcopy[u] + kh * 32 + 8 * g4 + 4 * hh);
#pragma unroll
                        for (int r = 0; r < 4; ++r) st[u][4 * g4 + r] = fmaf(gq[u], bv[r], -m_run[u]);
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) st[u][r] = 0.f;
                }
            }
            HDBG(41);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                for (int u = 0; u < U; ++u)
                    st[u] = mfma32<MODE>(kf[ks], qf[j][u][ks], st[u]);
            HDBG(42);

            typedef __attribute__((ext_vector_type(8))) short s16x8;
            auto v_frag = [&](int s2, int ds) -> bf16x8 {
                const int kb = s2 * 16 + 4 * (vg >> 1);
                const int key0 = kb + vqq, key1 = kb + 8 + vqq;
                const int o0 = key0 * RS + (v_unit_swz<64>(key0, ds) << 6) + ((vg & 1) << 5) + (vpp << 3);
                const int o1 = key1 * RS + (v_unit_swz<64>(key1, ds) << 6) + ((vg & 1) << 5) + (vpp << 3);
                const s16x4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(tV + o0));
                const s16x4 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(tV + o1));
                const s16x8 av = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
                return __builtin_bit_cast(bf16x8, av);
            };
            bf16x8 vfr[2][DSUB];
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                for (int ds = 0; ds < DSUB; ++ds) vfr[s2][ds] = v_frag(s2, ds);

            HDBG(43);
            float mloc[U], alpha[U], m_new[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                float m = -1e30f;
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const int kb = kh * 32 + 8 * g4 + 4 * hh;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float v = st[u][4 * g4 + r];
                        if (RAGGED) v = (kb + r < T) ? v : -INFINITY;
                        st[u][4 * g4 + r] = v;
                        m = fmaxf(m, v);
                    }
                }
                const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(m), __float_as_uint(m), false, false);
                mloc[u] = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
                alpha[u] = 1.0f; m_new[u] = 0.f;
            }
            HDBG(44);
            if constexpr (LAZY) {
                // ONE rarely-taken branch for both blocks (after the first keys a score seldom exceeds the stale maximum by 2^8): the
                // common path stays two long basic blocks in which the scheduler can interleave the two chains
                if (kh == 0 || !__all(fmaxf(mloc[0], mloc[1]) <= LAZY_T)) {
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const float d = kh == 0 ? (mloc[u] > -1e29f ? mloc[u] : 0.f) : fmaxf(mloc[u], 0.f);
                        m_run[u] += d;
                        if (kh != 0) {
                            alpha[u] = __builtin_amdgcn_exp2f(-d);
#pragma unroll
                            for (int i = 0; i < DSUB; ++i)
#pragma unroll
                                for (int r = 0; r < 16; ++r) ot[u][i][r] *= alpha[u];
                        }
#pragma unroll
                        for (int r = 0; r < 16; ++r) st[u][r] -= d;
                    }
                }
            } else {
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    m_new[u] = fmaxf(m_run[u], mloc[u]);
                    alpha[u] = __builtin_amdgcn_exp2f(m_run[u] - m_new[u]);
                    m_run[u] = m_new[u];
                }
                if (!__all(alpha[0] == 1.0f && alpha[1] == 1.0f)) {
#pragma unroll
                    for (int u = 0; u < U; ++u)
#pragma unroll
                        for (int i = 0; i < DSUB; ++i)
#pragma unroll
                            for (int r = 0; r < 16; ++r) ot[u][i][r] *= alpha[u];
                }
            }

            HDBG(45);
            float ls0[U], ls1[U];
#pragma unroll
            for (int u = 0; u < U; ++u) { ls0[u] = 0.f; ls1[u] = 0.f; }
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    bf16x8 ph;
                    float e[8];
#pragma unroll
                    for (int jj = 0; jj < 8; jj += 2) {
                        e[jj] = __builtin_amdgcn_exp2f(LAZY ? st[u][8 * s2 + jj] : st[u][8 * s2 + jj] - m_new[u]);
                        e[jj + 1] = __builtin_amdgcn_exp2f(LAZY ? st[u][8 * s2 + jj + 1] : st[u][8 * s2 + jj + 1] - m_new[u]);
                        ls0[u] += e[jj];
                        ls1[u] += e[jj + 1];
                    }
                    if constexpr (mode_traits<MODE>::f16) {
                        f16x8 p16;
#pragma unroll
                        for (int jj = 0; jj < 8; ++jj) p16[jj] = (_Float16)e[jj];
                        ph = __builtin_bit_cast(bf16x8, p16);
                    } else {
#pragma unroll
                        for (int jj = 0; jj < 8; ++jj) ph[jj] = (__bf16)e[jj];
                    }
#pragma unroll
                    for (int ds = 0; ds < DSUB; ++ds) ot[u][ds] = mfma32<MODE>(vfr[s2][ds], ph, ot[u][ds]);
                }
            HDBG(46);
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const float lh = ls0[u] + ls1[u];
                const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(lh), __float_as_uint(lh), false, false);
                l_run[u] = l_run[u] * alpha[u] + (__uint_as_float(sw[0]) + __uint_as_float(sw[1]));
            }
            HDBG(47);
#undef HDBG
        };

        // FIRST round: the tiles are still landing -- tile kt is complete once every wave's own pieces are (counted wait) and the
        // barrier has made the other waves' pieces visible.  Later rounds: everything is resident, nothing to wait for.
        const int nh = (T + 31) >> 5;                                          // half tiles with at least one real key
        const int nhfull = (T & 31) ? nh - 1 : nh;
        if (on) {
            for (int kh = 0; kh < nhfull; ++kh) {
                if (!(kh & 1)) RDBG(8 + j * 8 + (kh >> 1));
                if (j == 0 && !(kh & 1)) { wait_tile(kh >> 1); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); }
                half(kh, std::false_type{});
            }
            if (nhfull < nh) {
                if (j == 0 && !(nhfull & 1)) { wait_tile(nhfull >> 1); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); }
                half(nhfull, std::true_type{});
            }
        } else {                                                               // (j == 0 only) a wave past the utterance's last query: waits and barriers
            for (int kt = 0; kt < nkt; ++kt) { wait_tile(kt); __builtin_amdgcn_s_barrier(); }
        }
        RDBG(32 + 2 * j);
        if (on) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (qq[u] < T) {
                    const float inv = 1.0f / l_run[u];
                    unsigned short* orow = p.out + (int64_t)(row0 + qq[u]) * p.ldo + h * 64;
#pragma unroll
                    for (int ds = 0; ds < DSUB; ++ds)
#pragma unroll
                        for (int r4 = 0; r4 < 4; ++r4) {
                            const int d = ds * 32 + 8 * r4 + 4 * hh;
                            store_act4<MODE>(orow + d, p.out_plane, ot[u][ds][4 * r4] * inv, ot[u][ds][4 * r4 + 1] * inv,
                                             ot[u][ds][4 * r4 + 2] * inv, ot[u][ds][4 * r4 + 3] * inv);
                        }
                }
            }
        }
        RDBG(33 + 2 * j);
    }
#ifdef SER_ATTN_DBG
    if (p.dbg && blockIdx.x == 100 && lane == 0) p.dbg[wave * 64 + 63] = __builtin_amdgcn_s_memrealtime();
#endif
}

template <int MODE, bool TBL, bool GX>
int launch_resident(const AttnParams& p, dim3 grid, size_t lds, hipStream_t s) {
    auto k = attention_resident_kernel<MODE, TBL, GX>;
    static std::atomic<bool> ready{false};
    if (!ready.load(std::memory_order_acquire)) {
        hipError_t e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return ser_fail((int)e, "ser_attention: cannot raise dynamic LDS");
        ready.store(true, std::memory_order_release);
    }
    hipLaunchKernelGGL(k, grid, dim3(RNT), lds, s, p);
    return ser_check_launch("ser_attention");
}

}  // namespace

int ser_attention_resident(const AttnParams& p0, int mode, int max_frames, hipStream_t s) {
    static const int knob = [] { const char* e = getenv("SER_ATTN_RESIDENT"); return e ? atoi(e) : 0; }();      // experiments build only; 1 = take eligible launches
    if (!knob || (mode != SER_MODE_BF16 && mode != SER_MODE_FP16) || p0.dh != 64 || p0.scale > 0.f || max_frames > MAXT ||
        p0.key_lens || p0.bias2d || (p0.table && !p0.gate && !p0.gx))           // (gate columns in the packed projection: the round-3 form, tiled kernel)
        return 1;
    AttnParams p = p0;
    const int nktm = (max_frames + ABKV - 1) / ABKV;
    int bias_stride = 0;
    if (p.table) {
        bias_stride = ((2 * max_frames - 1 + 3 + 4 + 3) / 4) * 4;                // all 2T-1 distances (+ the 0..3 alignment slots + the widest shift)
        bias_stride += (16 - (bias_stride & 63) + 64) & 63;                       // == 16 (mod 64) floats: the 4 copies x 4 query phases on distinct bank quads
    }
    const size_t lds = (size_t)2 * nktm * TILE + (size_t)4 * bias_stride * 4;
    if (lds > 160 * 1024 || bias_stride > 2 * 4 * RNT) return 1;
    // 64-query groups of one (utterance, head) over nq blocks when the batch alone would leave CUs idle
    const int bh = p.B * p.H;
    int nq = 1;
    while (nq < 2 && bh * nq <= 128 && max_frames > 32 * U * RW * nq) nq *= 2;
    p.nq = nq;
    p.nitems = nktm;
    p.bias_stride = bias_stride;
    dim3 grid((unsigned)(bh * nq), 1, 1);
#define SER_RES(M_) (!p.table ? launch_resident<M_, false, false>(p, grid, lds, s) \
                              : (p.gate ? launch_resident<M_, true, false>(p, grid, lds, s) : launch_resident<M_, true, true>(p, grid, lds, s)))
    return mode == SER_MODE_BF16 ? SER_RES(SER_MODE_BF16) : SER_RES(SER_MODE_FP16);
#undef SER_RES
}
