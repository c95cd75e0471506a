// Fused self-attention over a packed ragged batch (ser_attention).
//
//   out[q,:] = softmax_k( (q.k) * scale + gate[q,h] * table[h, k-q+T-1] ) v      per utterance, per head
//
// gfx950 design (flash-style, scores never leave registers)
//   * block = 4 waves, wave = 32 query rows, K/V tiles of 64 keys staged in LDS.
//   * S^T = K Q^T with v_mfma_f32_32x32x16_bf16: the key index lands in the accumulator
//     REGISTERS and the query on the LANE, so the row max / row sum of the online softmax
//     are 32 lane-local ops + one cross-half shuffle (no LDS, no butterflies).
//   * O^T = V^T P^T reuses the S^T accumulator as the MFMA B operand with no data
//     movement (registers 8s..8s+7 -> k-step s); V^T is the A operand and is produced
//     from the row-major V tile by ds_read_b64_tr_b16 (hardware transpose read).
//     O^T has the query on the lane as well, so the rescale by 2^(m_old-m_new) is lane-local
//     and is skipped (wave-uniform branch) when no row maximum moved.
//   * K/V tiles are double-buffered in LDS with an issue-early / write-late register stage:
//     the global loads of tile t+1 are issued before the MFMAs of tile t and written to the
//     other LDS buffer after them -> one barrier per tile, HBM/L2 latency under compute.
//   * softmax in the exp2 domain: score*log2(e) folded into the one FMA that applies scale and
//     bias, v_exp_f32 directly; the key-padding select only exists in the last (ragged) tile.
//   * WavLM's gated relative bias needs only the 2T-1 distinct distances: the head's table row is
//     staged once in LDS as 4 copies shifted by 0..3 elements, so the 4 consecutive keys a
//     register quad holds are ONE aligned ds_read_b128 whatever (key - query) mod 4 is.
//   * LDS images: K rows XOR-swizzled for conflict-free ds_read_b128 row reads, V 64-byte
//     units XOR-swizzled so the 4 keys of a transposed read hit 4 different bank quarters.
//   * FP32X mode: every product is the 3-term bf16 split (hi*hi + lo*hi + hi*lo).
//   * round 4: a 96-wide form for head dims 72 .. 96, two waves per SIMD for every two-plane form that fits 256 registers, and a
//     high-occupancy arm (OCC: one K/V buffer, fragments read just in time, 128 registers, four blocks per CU) that the launcher picks
//     for single-plane 64-wide launches of 513 .. 1 024 blocks -- see the template's comment and DESIGN.md section 5.
#include "attn_common.h"
#include <stdlib.h>
#include <type_traits>
#include <atomic>

#ifdef SER_ATTN_DBG
extern "C" { void* ser_attn_dbg_ptr = nullptr; }
#endif
// TBL: a relative-position bias table is present (WavLM) -- compile-time, so the plain path carries no bias code
// and the bias path does not zero accumulators it is about to overwrite.
// NWV waves (x 32 queries) per block.  8 waves halve the K/V global->LDS traffic and the bias-row copies per query:
// used for bf16 head dims <= 64 once an utterance has more than one 128-query tile.
#ifndef SER_ATTN_LAZY
#define SER_ATTN_LAZY 1          // pre-scaled launches keep a STALE row maximum inside the accumulators' start value (see the tile loop)
#endif
#ifndef SER_ATTN_MINW
#define SER_ATTN_MINW 2          // waves per SIMD the register allocation must leave room for (A/B knob at build time)
#endif
// B2D: dense additive bias from global memory instead of the relative-position table (needs PRE, excludes TBL).
// GB: the relative-position table is read from GLOBAL memory (L2: 2T-1 floats per head) instead of an LDS window -- the form for
// utterances whose window (T + 192 distances x 4 shifted copies) does not fit the 160 KiB of LDS (beyond ~2 min of audio): the
// reference has no length limit (preprocess_speech.py:47-50 runs whatever librosa.load returns).  Needs PRE and TBL, head dim <= 64.
// (Round 3's persistent form -- resident blocks walking the items -- and its 8-wave form measured +- 0 / -1.2 % on the step and were
// removed from the build in round 4: DESIGN.md section 10 keeps the record.)
// OCC (round 4; single-plane 64-wide forms only): the HIGH-OCCUPANCY form -- ONE K/V buffer and fragments read just in time keep the
// kernel at 128 registers and 28 KiB of LDS (499 frames), i.e. FOUR waves per SIMD / four blocks per CU, so that the 1 024 blocks of a
// 16-utterance x 16-head x 499-frame launch are resident in ONE round instead of two (40.5 -> 37.0 us; ragged 64..499 frames 34.3 -> 30.5).
// With at most two blocks per CU to place (8-utterance launches: 512 blocks) the low-occupancy form -- double buffer, all LDS reads of a phase
// up front, 198 registers -- is the faster one (23.0 against 24.4 us): the launcher picks by grid size.
template <int DHP, int MODE, bool PRE, bool TBL, int NWV = 4, bool B2D = false, bool GB = false, bool OCC = false>
// waves per SIMD the registers leave room for: two everywhere (round 4: the 96-wide two-plane forms and the 128-wide ones WITHOUT a bias table
// fit 256 registers unspilled -- head dim 80 in the 3-product modes 98.9 -> 49.9 us per 8 x 499 frames) except the 128-wide two-plane form
// with a bias table (208 spilled registers at two; no encoder uses it: WavLM's head dim is 64)
__global__ __launch_bounds__(64 * NWV, (DHP == 128 && mode_traits<MODE>::planes == 2 && TBL) ? 1 : (OCC ? 4 : ((DHP == 64 && mode_traits<MODE>::planes == 1) ? SER_ATTN_MINW : 2)))
void attention_kernel(const AttnParams p) {
    constexpr int NT = 64 * NWV;                // threads per block
    // NP: planes of Q and K (the logit path S = K Q^T: 3 products when 2), NPV: planes of V and P.  FP16Q (the "f16q" numerics
    // mode) keeps the logits fp32-grade -- q, k arrive as fp16 hi + lo planes -- and runs P V on single fp16 products; its
    // output is the single-plane FP16 operand of the output projection.  FP16X ("f16a") splits everything, like FP32X.
    constexpr int NP = mode_traits<MODE>::planes;
    constexpr int NPV = (MODE == SER_MODE_FP32X || MODE == SER_MODE_FP16X) ? 2 : 1;
    constexpr int OUTM = (MODE == SER_MODE_FP16Q) ? SER_MODE_FP16 : MODE;
    constexpr int RS = DHP * 2;                 // LDS row bytes
    constexpr int KS = DHP / 16;                // QK^T k-steps
    constexpr int DSUB = DHP / 32;              // 32-wide output column blocks
    constexpr int CPR = DHP / 8;                // 16-byte chunks per row
    constexpr int TILE = ABKV * RS;             // bytes of one K or V plane tile
    constexpr int NCH = ABKV * CPR / NT;        // staged 16-B chunks per thread per plane per operand
    // double-buffer when the register stage is <= 32 VGPRs (the high-occupancy form: one buffer)
    constexpr bool DB = !OCC && (NCH * (NP + NPV)) <= 8;
    constexpr int NBUF = DB ? 2 : 1;
    constexpr int BUF = (NP + NPV) * TILE;      // one K+V buffer
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* ldsB = (float*)(smem + NBUF * BUF);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef SER_ATTN_DBG
    const unsigned long long dbg_entry = __builtin_amdgcn_s_memtime();
    const unsigned long long dbg_entry_rt = __builtin_amdgcn_s_memrealtime();
    unsigned long long dbg_p[6] = {0, 0, 0, 0, 0, 0};
#endif
    // XCD-aware decode of the linear block id L: the q-tiles of one (utterance, head) sit at
    // L, L+8, L+16, ... -> same XCD (blocks are dealt round-robin over the 8 XCDs), close in time,
    // so K/V are fetched from HBM once and re-read from that XCD's L2 by the other q-tiles.
    const int L = blockIdx.x;
    const int bh = (L / (8 * p.nq)) * 8 + (L & 7), qt = (L >> 3) % p.nq;
    if (bh >= p.H * p.B) return;
    const int h = bh % p.H, b = bh / p.H;
    const int row0 = p.frame_offs[b];
    const int T = p.frame_offs[b + 1] - row0;
    const int q0 = qt * (32 * NWV);
    if (q0 >= T) return;
    DBG_P(0);
    const int dh = p.dh;
    const int hh = lane >> 5, l31 = lane & 31;
    const int TK = p.key_lens ? p.key_lens[b] : T;               // attendable keys (== T for speech)
    const int TS = B2D ? T : TK;                                 // keys staged / visited (padded queries of B2D see all T)
    const int nkt = (TS + ABKV - 1) / ABKV;
    // first relative-position slot any query of this block reads, moved down by 0..3 (possibly below 0: zero-filled) so
    // that T-1-jmin == 3 (mod 4).  Queries 4m..4m+3 of a quad then read the SAME aligned offset of the 4 shifted copies
    // (one 16-lane group of a ds_read_b128 = 4 quads x 4 copies = 16 distinct 4-bank slots); with any other residue a
    // quad straddles two offsets and two of its lanes collide with the neighbouring quad (the q-tile holding the
    // utterance's last query used to run like that: 498 = 2 mod 4 at T = 499).
    const int jmin0 = max(0, (T - 1) - (q0 + 32 * NWV - 1));
    const int jmin = jmin0 - ((3 - (T - 1 - jmin0)) & 3);

    // ---- staging helpers: thread owns chunks c = tid + i*256 of the [64 keys][CPR] tile --------
    u32x4 stg[NP][2][NCH];
    const bool padded = (dh != DHP);
    // per-thread source of its chunks in key tile 0; a full tile kt is that plus kt * 64 rows (one 64-bit add per chunk
    // instead of the clamp + 64-bit multiply the ragged last tile needs: ~25 VALU per tile off the loop)
    const unsigned short* src0[NCH];
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int c = tid + i * NT;
        const int key = c / CPR, ch = c - key * CPR;
        src0[i] = p.qkv + (int64_t)(row0 + key) * p.ld + h * dh + (ch * 8 < dh ? ch * 8 : 0);
    }
    const int64_t tile_step = (int64_t)ABKV * p.ld;
    auto stage_load = [&](int kt, bool masked) {
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int c = tid + i * NT;
            const int key = c / CPR, ch = c - key * CPR;
            const int kg = kt * ABKV + key;
            // branch-free: always load from a valid address, zero by select (keys >= T, pad columns >= dh)
            const bool ok = (kg < TS) && (ch * 8 < dh);
            const unsigned short* src = (kg < T || !masked) ? src0[i] + kt * tile_step
                                                            : src0[i] + (int64_t)(T - 1 - key) * p.ld;
            const unsigned int keep = ok ? 0xffffffffu : 0u;
#pragma unroll
            for (int pl = 0; pl < NP; ++pl) {
                u32x4 kv = *(const u32x4*)(src + p.k_col + pl * p.plane);
                u32x4 vv = {0u, 0u, 0u, 0u};
                if (pl < NPV) vv = *(const u32x4*)(src + p.v_col + pl * p.plane);
                if (masked) {                                    // wave-uniform: only ragged tiles / padded head dims
#pragma unroll
                    for (int e = 0; e < 4; ++e) { kv[e] &= keep; vv[e] &= keep; }
                }
                stg[pl][0][i] = kv;
                if (pl < NPV) stg[pl][1][i] = vv;
            }
        }
    };
    auto stage_write = [&](int buf) {
        char* base = smem + buf * BUF;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int c = tid + i * NT;
            const int key = c / CPR, ch = c - key * CPR;
#pragma unroll
            for (int pl = 0; pl < NP; ++pl) {
                *(u32x4*)(base + pl * TILE + key * RS + (k_swz<DHP>(key, ch) << 4)) = stg[pl][0][i];
                if (pl < NPV)
                    *(u32x4*)(base + NP * TILE + pl * TILE + key * RS + (v_unit_swz<DHP>(key, ch >> 2) << 6) + ((ch & 3) << 4)) = stg[pl][1][i];
            }
        }
    };

    // tiles without key padding; with a dense bias every tile takes the per-element key check (real queries stop at TK,
    // padded ones at T; 80-token problems: two tiles)
    const int nfull = B2D ? 0 : ((TK & (ABKV - 1)) ? nkt - 1 : nkt);

    // ---- this head's bias window, requested FIRST: vector-memory data returns in issue order, so these 8 words land before
    // the K/V/Q data and the four shifted LDS copies are written while tile 0 is still in flight (they used to be requested
    // last and copied, 20 guarded ds_write_b32 per thread, after everything had landed: 1.7 us of a 16 us block at T = 499).
    // Thread i owns the aligned quad j = 4i..4i+3 of every copy: copy c [j] = window[j + c], so it needs window[4i .. 4i+6].
    const float* trow = nullptr;
    int bn = 0;
    float bw[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    auto bias_load = [&](int j4) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int idx = j4 + e;                                  // window index; jmin may be -3..-1, the tail runs past 2T-1
            bw[e] = (e < 7 && idx < bn && idx + jmin >= 0 && j4 < p.bias_stride) ? trow[idx] : 0.f;
        }
    };
    auto bias_write = [&](int j4) {
        if (j4 < p.bias_stride) {
#pragma unroll
            for (int c = 0; c < 4; ++c)
                *(f32x4*)(ldsB + c * p.bias_stride + j4) = (f32x4){bw[c], bw[c + 1], bw[c + 2], bw[c + 3]};
        }
    };
    if (TBL && !GB) {
        // only the distances this block's queries can see: key - query + T-1 in [jmin, jmin + 32*NWV + T + 63]
        trow = p.table + (int64_t)h * (2 * p.table_T - 1) + (p.table_T - T) + jmin;
        bn = 2 * T - 1 - jmin;
        bias_load(4 * tid);
    }
    stage_load(0, padded || nfull == 0);
    DBG_P(1);

    // ---- Q fragments: lane holds Q[q][16*ks + 8*hh + j].  Requested here, with the gate inputs, so that their
    // latency overlaps the first K/V tile's and the bias row's instead of following them.
    const int q = q0 + wave * 32 + l31;
    const int qc = q < T ? q : T - 1;
    bf16x8 qf[NP][KS];
    {
        const unsigned short* qrow = p.qkv + (int64_t)(row0 + qc) * p.ld + p.q_col + h * dh;
#pragma unroll
        for (int pl = 0; pl < NP; ++pl)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const int d = ks * 16 + hh * 8;
                u32x4 v = {0u, 0u, 0u, 0u};
                if (d < dh && !(B2D && q >= TK)) v = *(const u32x4*)(qrow + pl * p.plane + d);      // padded query: all scores equal
                qf[pl][ks] = __builtin_bit_cast(bf16x8, v);
            }
    }
    float g_in0 = 0.f, g_in1 = 0.f, g_c = 0.f;                     // raw gate inputs (consumed after the bias copy)
    // in-kernel gate: one MFMA chain per query block, shaped like S = K Q^T with the two folded weight rows of the head in the place of the
    // keys (row r of the "key" operand = weight row r & 1, so EVERY lane finds pre-activation a in accumulator 0 and b in accumulator 1) and
    // the query rows' head slice of the layer input in the place of Q.  Fragments are requested here, multiplied after the bias copy.
    // (First form: 128 FMAs per lane on fp32 weights -- 21 vector loads per lane instead of 9, +3.3 us on the 16-utterance launch: the
    // prologue's vector-memory INSTRUCTIONS are what it costs, 16 cycles of the texture-address path each.)
    bf16x8 gxf[NP][KS], gwf[NP][KS];
    f32x2 gst = {0.f, 1.f};
    if (TBL) {
        if (p.gate) {
            g_in0 = p.gate[(int64_t)(row0 + qc) * p.H + h];
        } else if (p.gx) {
            const unsigned short* xr = p.gx + (int64_t)(row0 + qc) * p.gx_ld + h * dh;
            const unsigned short* wr = p.gw + ((int64_t)h * 2 + (l31 & 1)) * dh;
#pragma unroll
            for (int pl = 0; pl < NP; ++pl)
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {                    // unconditional loads from valid addresses (a guarded load costs hipcc a
                    const int d = ks * 16 + hh * 8;                  // branch and a vmcnt(0) each); chunks past dh re-read chunk 0 and the
                    const int dd = d < dh ? d : 0;                   // activation side is zeroed
                    u32x4 xv = *(const u32x4*)(xr + pl * p.gx_plane + dd);
                    if (d >= dh) xv = (u32x4){0u, 0u, 0u, 0u};
                    gxf[pl][ks] = __builtin_bit_cast(bf16x8, xv);
                    gwf[pl][ks] = __builtin_bit_cast(bf16x8, *(const u32x4*)(wr + pl * p.gw_plane + dd));
                }
            gst = *(const f32x2*)(p.gstat + 2 * (int64_t)(row0 + qc));
            g_c = p.gru_const[h];
        } else {
            // gate pre-activations ride along as two extra columns per head of the packed projection
            const unsigned short* gp = p.qkv + (int64_t)(row0 + qc) * p.ld + p.gate_col + 2 * h;
            g_in0 = elem2f<MODE>(gp[0]); g_in1 = elem2f<MODE>(gp[1]);
            if (NP == 2) { g_in0 += elem2f<MODE>(gp[p.plane]); g_in1 += elem2f<MODE>(gp[p.plane + 1]); }
            g_c = p.gru_const[h];
        }
    }

    DBG_P(2);
    // ---- this head's bias row, 4 shifted copies: copy c [j] = table[h][(table_T-T) + jmin + j + c] -------
    if (TBL && !GB) {
        bias_write(4 * tid);
        for (int j4 = 4 * (tid + NT); j4 < p.bias_stride; j4 += 4 * NT) {   // utterances beyond ~8 s: further passes
            bias_load(j4);
            bias_write(j4);
        }
    }

    DBG_P(3);
    const int klim = (B2D && q >= TK) ? T : TK;                   // keys this lane's query may attend to
    const float* brow = B2D ? p.bias2d + ((int64_t)bh * p.b2d_T + qc) * p.b2d_ld : nullptr;
    const float c1 = p.scale * LOG2E;
    float gq2 = 0.f;
    if (TBL) {
        if (!p.gate && p.gx) {
            f32x16 ga;
#pragma unroll
            for (int r = 0; r < 16; ++r) ga[r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                if (NP == 2) {                                       // same product order as S: lo * hi, hi * lo, hi * hi
                    ga = mfma32<MODE>(gwf[NP - 1][ks], gxf[0][ks], ga);
                    ga = mfma32<MODE>(gwf[0][ks], gxf[NP - 1][ks], ga);
                }
                ga = mfma32<MODE>(gwf[0][ks], gxf[0][ks], ga);
            }
            const float da = ga[0], db = ga[1];
            const f32x4 cb = *(const f32x4*)(p.gcb + 4 * h);
            g_in0 = fmaf(gst[1], da - gst[0] * cb[0], cb[2]);           // deferred LayerNorm 1 in closed form (ser_hip.h)
            g_in1 = fmaf(gst[1], db - gst[0] * cb[1], cb[3]);
        }
        if (p.gate) {
            gq2 = g_in0 * LOG2E;
        } else {
            const float ga = __builtin_amdgcn_rcpf(1.f + __expf(-g_in0)), gb = __builtin_amdgcn_rcpf(1.f + __expf(-g_in1));
            gq2 = (ga * (gb * g_c - 1.f) + 2.f) * LOG2E;
        }
    }
    // aligned bias window: index of key kb (multiple of 4) is kb - qc + T-1 - jmin = a + sh with a % 4 == 0
    const int bsh = (T - 1 - qc - jmin) & 3;
    const float* bcopy = ldsB + bsh * p.bias_stride + ((T - 1 - qc - jmin) - bsh);
    // GB: this query's row of distances in the global table: gbias[k] = table[h][k - q + T-1 (+ table_T - T)], k = key index
    const float* gbias = GB ? p.table + (int64_t)h * (2 * p.table_T - 1) + (p.table_T - T) + (T - 1 - qc) : nullptr;
    auto bias_quad = [&](int kb, bool ragged) -> f32x4 {          // gate-less bias of keys kb..kb+3 for this lane's query
        if constexpr (GB) {
            if (!ragged) {
                const f32x4_u u = *(const f32x4_u*)(gbias + kb);
                return (f32x4){u.v[0], u.v[1], u.v[2], u.v[3]};
            }
            f32x4 r;
#pragma unroll
            for (int e = 0; e < 4; ++e) r[e] = (kb + e < T) ? gbias[kb + e] : 0.f;      // never past the utterance's own distances
            return r;
        } else {
            return *(const f32x4*)(bcopy + kb);
        }
    };

    f32x16 ot[DSUB];
#pragma unroll
    for (int i = 0; i < DSUB; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) ot[i][r] = 0.f;
    // LAZY (pre-scaled q + WavLM bias table): the S accumulators start at gate * bias - m_run instead of gate * bias -- the same one FMA per
    // score -- so the MFMA chain delivers scores already relative to the running maximum and the exponentials need no subtraction; m_run is
    // only raised when a tile's scores exceed it by more than LAZY_T (P <= 2^LAZY_T: far inside fp32 / fp16 range), which after the first
    // tiles is rare, and O and l are rescaled in that branch only.  O / l is unchanged: every P of a row carries the same factor
    // 2^(m_true - m_run).  Measured (tools/attn_lazy_ab.sh, two A/B pairs): WavLM-large step 1 968 / 1 974 -> 1 988 / 1 985 utt/s.  Without a
    // bias table the start value needs its own 16-register block that hipcc copies per tile: Whisper 380.4 -> 377.2, so those keep the exact maximum.
    constexpr bool LAZY = SER_ATTN_LAZY && PRE && TBL;
    constexpr float LAZY_T = 8.0f;
    float m_run = LAZY ? 0.f : -1e30f, l_run = 0.f;
#ifdef SER_ATTN_DBG
    const bool dbg_on = p.dbg && (int)blockIdx.x == ser_attn_dbg_block_dev;
    if (dbg_on && lane == 0) p.dbg[(wave * 64 + 63) * 6] = __builtin_amdgcn_s_memtime();     // end of the prologue loads' issue
#endif

    stage_write(0);
    DBG_P(4);
    __syncthreads();
    DBG_P(5);
#ifdef SER_ATTN_DBG
    if (dbg_on && lane == 0) for (int i = 0; i < 6; ++i) p.dbg[(wave * 64 + 61) * 6 + i] = dbg_p[i] - dbg_entry;
#endif

    auto tile = [&](int kt, auto ragged_tag) {
        constexpr bool RAGGED = decltype(ragged_tag)::value;
        // WIDE (bf16, head dim <= 64): every LDS read of a phase is issued before its first consumer, so a phase
        // pays the LDS latency once; left alone, hipcc re-uses one fragment register and emits
        // read -> wait(0) -> MFMA eight times in a row (measured: 1400 of a tile's 3600 cycles).
        constexpr bool WIDE = !OCC && (NP == 1 && DHP == 64);
        const int cur = DB ? (kt & 1) : 0;
        const char* ldsK = smem + cur * BUF;
        const char* ldsV = ldsK + NP * TILE;
#ifdef SER_ATTN_DBG
        unsigned long long dbg_t[6] = {0, 0, 0, 0, 0, 0};
#endif
        DBG_T(0);
        if (DB && kt + 1 < nkt) stage_load(kt + 1, padded || kt + 1 >= nfull);   // issue early: lands under the MFMAs below

        // ---- S^T = K Q^T  (rows = keys in registers, col = query on the lane)
        // PRE (q pre-scaled by dh^-0.5*log2e in the projection epilogue): the accumulators START at the
        // gated bias, so the MFMA chain delivers finished exp2-domain scores (no per-score multiply/add).
        f32x16 st[2];
        if constexpr (WIDE) {
            f32x4 bvv[2][4];
            bf16x8 kf[2][KS];
            if (PRE && TBL) {
#pragma unroll
                for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4)
                        bvv[sub][g4] = bias_quad(kt * ABKV + sub * 32 + 8 * g4 + 4 * hh, RAGGED);
            }
            if (PRE && B2D) {                                        // rows are b2d_ld = a multiple of 64 floats long, zero padded
#pragma unroll
                for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4)
                        bvv[sub][g4] = *(const f32x4*)(brow + kt * ABKV + sub * 32 + 8 * g4 + 4 * hh);
            }
#pragma unroll
            for (int sub = 0; sub < 2; ++sub) {
                const int key = sub * 32 + l31;
#pragma unroll
                for (int ks = 0; ks < KS; ++ks)
                    kf[sub][ks] = *(const bf16x8*)(ldsK + key * RS + (k_swz<DHP>(key, ks * 2 + hh) << 4));
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        st[sub][4 * g4 + r] = (PRE && TBL) ? (LAZY ? fmaf(gq2, bvv[sub][g4][r], -m_run) : gq2 * bvv[sub][g4][r])
                                                           : ((PRE && B2D) ? bvv[sub][g4][r] : 0.f);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)                          // the two key halves are independent chains
#pragma unroll
                for (int sub = 0; sub < 2; ++sub)
                    st[sub] = mfma32<MODE>(kf[sub][ks], qf[0][ks], st[sub]);
        } else {
#pragma unroll
            for (int sub = 0; sub < 2; ++sub) {
                if (PRE && TBL) {
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4) {
                        const int kb = kt * ABKV + sub * 32 + 8 * g4 + 4 * hh;
                        const f32x4 bv = bias_quad(kb, RAGGED);
#pragma unroll
                        for (int r = 0; r < 4; ++r) st[sub][4 * g4 + r] = LAZY ? fmaf(gq2, bv[r], -m_run) : gq2 * bv[r];
                    }
                } else if (PRE && B2D) {
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4) {
                        const f32x4 bv = *(const f32x4*)(brow + kt * ABKV + sub * 32 + 8 * g4 + 4 * hh);
#pragma unroll
                        for (int r = 0; r < 4; ++r) st[sub][4 * g4 + r] = bv[r];
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) st[sub][r] = 0.f;
                }
                const int key = sub * 32 + l31;
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const int off = key * RS + (k_swz<DHP>(key, ks * 2 + hh) << 4);
                    const bf16x8 kh = *(const bf16x8*)(ldsK + off);
                    st[sub] = mfma32<MODE>(kh, qf[0][ks], st[sub]);
                    if (NP == 2) {
                        const bf16x8 kl = *(const bf16x8*)(ldsK + TILE + off);
                        st[sub] = mfma32<MODE>(kl, qf[0][ks], st[sub]);
                        st[sub] = mfma32<MODE>(kh, qf[NP - 1][ks], st[sub]);
                    }
                }
            }
        }

        DBG_T(1);
        // transposed V reads: lane (qq,pp) of its 16-lane group addresses key kb+qq, 8 bytes at pp
        typedef __attribute__((ext_vector_type(8))) short s16x8;
        const int vg = lane >> 4, vqq = (lane >> 2) & 3, vpp = lane & 3;
        auto v_frag = [&](const char* plane, int sub, int s2, int ds) -> bf16x8 {
            const int kb = sub * 32 + s2 * 16 + 4 * (vg >> 1);
            const int key0 = kb + vqq, key1 = kb + 8 + vqq;
            const int o0 = key0 * RS + (v_unit_swz<DHP>(key0, ds) << 6) + ((vg & 1) << 5) + (vpp << 3);
            const int o1 = key1 * RS + (v_unit_swz<DHP>(key1, ds) << 6) + ((vg & 1) << 5) + (vpp << 3);
            const s16x4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(plane + o0));
            const s16x4 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(plane + o1));
            const s16x8 av = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
            return __builtin_bit_cast(bf16x8, av);
        };
        bf16x8 vfr[WIDE ? 2 : 1][WIDE ? 2 : 1][WIDE ? DSUB : 1];
        if constexpr (WIDE) {                                        // requested now: they land under the QK MFMAs / the max
#pragma unroll
            for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                    for (int ds = 0; ds < DSUB; ++ds) vfr[sub][s2][ds] = v_frag(ldsV, sub, s2, ds);
            __builtin_amdgcn_sched_barrier(0);
        }

        // ---- scores (log2 domain) and running max
        float mloc = -1e30f;
#pragma unroll
        for (int sub = 0; sub < 2; ++sub)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int kb = kt * ABKV + sub * 32 + 8 * g4 + 4 * hh;
                f32x4 bv = {0.f, 0.f, 0.f, 0.f};
                if (!PRE && TBL) bv = *(const f32x4*)(bcopy + kb);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float v = PRE ? st[sub][4 * g4 + r] : fmaf(st[sub][4 * g4 + r], c1, gq2 * bv[r]);
                    if (RAGGED) v = (kb + r < klim) ? v : -INFINITY;
                    st[sub][4 * g4 + r] = v;
                    mloc = fmaxf(mloc, v);
                }
            }
        {   // other half-wave's maximum for the same query: one v_permlane32_swap instead of an LDS bpermute
            const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mloc), __float_as_uint(mloc), false, false);
            mloc = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
        }
        DBG_T(2);
        float m_new = 0.f, alpha = 1.0f;                             // LAZY: the scores are already relative to m_run
        if constexpr (LAZY) {
            if (kt == 0 || !__all(mloc <= LAZY_T)) {                 // wave-uniform; kt == 0: nothing accumulated yet, take the tile's own maximum
                const float d = kt == 0 ? (mloc > -1e29f ? mloc : 0.f) : fmaxf(mloc, 0.f);   // (a query whose first tile is all padding: 0)
                m_run += d;
                if (kt != 0) {
                    alpha = __builtin_amdgcn_exp2f(-d);
#pragma unroll
                    for (int i = 0; i < DSUB; ++i)
#pragma unroll
                        for (int r = 0; r < 16; ++r) ot[i][r] *= alpha;
                }
                const f32x2 d2 = {d, d};
#pragma unroll
                for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                    for (int r = 0; r < 16; r += 2) {
                        const f32x2 v = (f32x2){st[sub][r], st[sub][r + 1]} - d2;
                        st[sub][r] = v[0]; st[sub][r + 1] = v[1];
                    }
            }
        } else {
            m_new = fmaxf(m_run, mloc);
            alpha = __builtin_amdgcn_exp2f(m_run - m_new);
            m_run = m_new;
            if (!__all(alpha == 1.0f)) {                             // wave-uniform: most tiles after the first few skip it
#pragma unroll
                for (int i = 0; i < DSUB; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) ot[i][r] *= alpha;
            }
        }

        // ---- P = exp2(S - m) and O^T += V^T P^T, 16 keys at a time: accumulator registers 8s..8s+7 of a key half
        // are the B fragment of k-step s, so the exponentials of the next 16 keys issue while these MFMAs run
        f32x2 lsum2 = {0.f, 0.f};                                    // LAZY: the row sum through v_pk_add_f32 (two scores per VALU issue; round 4: plain adds
                                                                     // under -fno-slp-vectorize measured no different, profiles/r04_attn_experiments.txt)
#pragma unroll
        for (int sub = 0; sub < 2; ++sub)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                bf16x8 ph, plo;
                float e[8];
                if constexpr (LAZY) {
#pragma unroll
                    for (int j = 0; j < 8; j += 2) {
                        const f32x2 ee = {__builtin_amdgcn_exp2f(st[sub][8 * s2 + j]), __builtin_amdgcn_exp2f(st[sub][8 * s2 + j + 1])};
                        lsum2 += ee;
                        e[j] = ee[0]; e[j + 1] = ee[1];
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        e[j] = __builtin_amdgcn_exp2f(st[sub][8 * s2 + j] - m_new);
                        lsum2[0] += e[j];
                    }
                }
                if constexpr (mode_traits<MODE>::f16) {
                    f16x8 p16, p16lo;                                // P in [0, 1] (LAZY: [0, 2^LAZY_T]): no saturation needed
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        p16[j] = (_Float16)e[j];
                        if (NPV == 2) p16lo[j] = (_Float16)(e[j] - (float)p16[j]);
                    }
                    ph = __builtin_bit_cast(bf16x8, p16);
                    if (NPV == 2) plo = __builtin_bit_cast(bf16x8, p16lo);
                } else {
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const __bf16 hi = (__bf16)e[j];
                        ph[j] = hi;
                        if (NPV == 2) plo[j] = (__bf16)(e[j] - (float)hi);
                    }
                }
#pragma unroll
                for (int ds = 0; ds < DSUB; ++ds) {
                    bf16x8 vh;
                    if constexpr (WIDE) vh = vfr[sub][s2][ds];
                    else vh = v_frag(ldsV, sub, s2, ds);
                    ot[ds] = mfma32<MODE>(vh, ph, ot[ds]);
                    if (NPV == 2) {
                        const bf16x8 vl = v_frag(ldsV + TILE, sub, s2, ds);
                        ot[ds] = mfma32<MODE>(vl, ph, ot[ds]);
                        ot[ds] = mfma32<MODE>(vh, plo, ot[ds]);
                    }
                }
            }
        float lsum;
        {   // the other half-wave summed the other 32 keys of this query
            const float lh = lsum2[0] + lsum2[1];
            const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(lh), __float_as_uint(lh), false, false);
            lsum = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
        }
        l_run = l_run * alpha + lsum;
        DBG_T(3);

        // ---- publish the next tile
        if (kt + 1 < nkt) {
            if (DB) {
                stage_write(cur ^ 1);                                // write late: other buffer, nobody reads it now
                DBG_T(4);
                __syncthreads();
            } else {
                __syncthreads();                                     // everyone done with the single buffer
                stage_load(kt + 1, padded || kt + 1 >= nfull);
                stage_write(0);
                __syncthreads();
            }
        }
        DBG_T(5);
#ifdef SER_ATTN_DBG
        if (dbg_on && lane == 0)
            for (int i = 0; i < 6; ++i) p.dbg[(wave * 64 + kt) * 6 + i] = dbg_t[i];
#endif
    };
    // the key-padding select exists only in the last tile of a ragged utterance
    if constexpr (B2D) {
        for (int kt = 0; kt < nkt; ++kt) tile(kt, std::true_type{});      // every tile checks keys per element (see nfull)
    } else {
        for (int kt = 0; kt < nfull; ++kt) tile(kt, std::false_type{});
        if (nfull < nkt) tile(nkt - 1, std::true_type{});
    }

#ifdef SER_ATTN_DBG
    if (dbg_on && lane == 0) { p.dbg[(wave * 64 + 62) * 6] = dbg_entry; p.dbg[(wave * 64 + 62) * 6 + 1] = __builtin_amdgcn_s_memtime();
                               p.dbg[(wave * 64 + 62) * 6 + 2] = __builtin_amdgcn_s_memrealtime() - dbg_entry_rt; }
#endif
    // ---- epilogue: O[q][d] = O^T[d][q] / l ; lane owns query q, 4 consecutive d per register quad
    // Context rows as SER_MODE_FP16M operands (round 5, head dim 64: a head is one 64-column tile of the output projection's A operand):
    // plane 0 = the fp16 copy, plane 1 = per row and head 128 bytes [P = v - hi: 64 e4m3 | Q = v: 64 e4m3] with one E8M0 scale per 32
    // columns, packed as the word out_scale[head][row] = [P 0-31, P 32-63, Q 0-31, Q 32-63] (include/ser_hip.h SER_MODE_FP16M).  A query's 32
    // columns of a block sit in lanes l and l ^ 32 (16 each): one cross-lane max per block and plane.  Every lane takes part in the shuffles.
    if constexpr (MODE == SER_MODE_FP16X && DHP == 64) {
        if (p.out_scale) {
            const float inv = (q < T) ? 1.0f / l_run : 0.f;
            const int64_t row = (int64_t)row0 + (q < T ? q : 0);
            unsigned short* orow = p.out + row * p.ldo + h * dh;
            unsigned char* seg = (unsigned char*)(p.out + p.out_plane) + (row * p.ldo + h * dh) * 2;
            unsigned codes = 0;
#pragma unroll
            for (int ds = 0; ds < DSUB; ++ds) {
                float v[16], lo[16], ax = 0.f, al = 0.f;
#pragma unroll
                for (int i = 0; i < 16; i += 2) {
                    v[i] = ot[ds][i] * inv; v[i + 1] = ot[ds][i + 1] * inv;
                    const unsigned h2 = pack_h2(v[i], v[i + 1]);
                    lo[i] = v[i] - h2f((unsigned short)(h2 & 0xffffu));
                    lo[i + 1] = v[i + 1] - h2f((unsigned short)(h2 >> 16));
                    ax = fmaxf(ax, fmaxf(fabsf(v[i]), fabsf(v[i + 1])));
                    al = fmaxf(al, fmaxf(fabsf(lo[i]), fabsf(lo[i + 1])));
                }
                ax = fmaxf(ax, __shfl_xor(ax, 32, 64));
                al = fmaxf(al, __shfl_xor(al, 32, 64));
                const unsigned cx = mx_code(ax), cl = mx_code(al);
                const float ix = mx_inv(cx), il = mx_inv(cl);
                codes |= (cl << (8 * ds)) | (cx << (16 + 8 * ds));
                // lanes l (hh = 0) and l + 32 (hh = 1) hold the interleaved 4-column groups of the block: v_permlane32_swap trades two of each
                // lane's four groups so that hh = 0 ends up with columns 0-15 and hh = 1 with 16-31 -- 16-byte stores (2 per plane and block)
                unsigned ha[4], hb[4], pw[4], qw[4];
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                    ha[r4] = pack_h2(v[4 * r4], v[4 * r4 + 1]);
                    hb[r4] = pack_h2(v[4 * r4 + 2], v[4 * r4 + 3]);
                    pw[r4] = mx_pack4(lo[4 * r4] * il, lo[4 * r4 + 1] * il, lo[4 * r4 + 2] * il, lo[4 * r4 + 3] * il);
                    qw[r4] = mx_pack4(v[4 * r4] * ix, v[4 * r4 + 1] * ix, v[4 * r4 + 2] * ix, v[4 * r4 + 3] * ix);
                }
                const auto a0 = __builtin_amdgcn_permlane32_swap(ha[0], ha[2], false, false), b0 = __builtin_amdgcn_permlane32_swap(hb[0], hb[2], false, false);
                const auto a1 = __builtin_amdgcn_permlane32_swap(ha[1], ha[3], false, false), b1 = __builtin_amdgcn_permlane32_swap(hb[1], hb[3], false, false);
                const auto p0 = __builtin_amdgcn_permlane32_swap(pw[0], pw[2], false, false), p1 = __builtin_amdgcn_permlane32_swap(pw[1], pw[3], false, false);
                const auto q0_ = __builtin_amdgcn_permlane32_swap(qw[0], qw[2], false, false), q1_ = __builtin_amdgcn_permlane32_swap(qw[1], qw[3], false, false);
                if (q < T) {
                    const int c0 = ds * 32 + 16 * hh;                     // this lane's 16 columns of the block
                    *(u32x4*)(orow + c0) = (u32x4){a0[0], b0[0], a0[1], b0[1]};
                    *(u32x4*)(orow + c0 + 8) = (u32x4){a1[0], b1[0], a1[1], b1[1]};
                    *(u32x4*)(seg + c0) = (u32x4){p0[0], p0[1], p1[0], p1[1]};
                    *(u32x4*)(seg + 64 + c0) = (u32x4){q0_[0], q0_[1], q1_[0], q1_[1]};
                }
            }
            if (q < T && hh == 0) p.out_scale[(int64_t)h * p.out_scale_ld + row] = codes;
            return;
        }
    }
    if (q < T) {
        const float inv = 1.0f / l_run;
        unsigned short* orow = p.out + (int64_t)(row0 + q) * p.ldo + h * dh;
#pragma unroll
        for (int ds = 0; ds < DSUB; ++ds)
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                const int d = ds * 32 + 8 * r4 + 4 * hh;
                if (d < dh)
                    store_act4<OUTM>(orow + d, p.out_plane, ot[ds][4 * r4] * inv, ot[ds][4 * r4 + 1] * inv,
                                     ot[ds][4 * r4 + 2] * inv, ot[ds][4 * r4 + 3] * inv);
            }
    }
}

template <int DHP, int MODE, bool PRE, bool TBL, int NWV = 4, bool B2D = false, bool GB = false, bool OCC = false>
static int launch_attention(const AttnParams& p, dim3 grid, size_t lds, hipStream_t s) {
    auto k = attention_kernel<DHP, MODE, PRE, TBL, NWV, B2D, GB, OCC>;
    static std::atomic<bool> ready{false};          // several host threads launch (see gemm.hip launch_mode)
    if (lds > 65536 && !ready.load(std::memory_order_acquire)) {
        hipError_t e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return ser_fail((int)e, "ser_attention: cannot raise dynamic LDS");
        ready.store(true, std::memory_order_release);
    }
    hipLaunchKernelGGL(k, grid, dim3(64 * NWV), lds, s, p);
    return ser_check_launch("ser_attention");
}

extern "C" int ser_attention(const void* qkv, int64_t ld, int64_t plane_stride, int q_col, int k_col, int v_col,
                             const int32_t* frame_offs, int B, int max_frames, const float* table, int table_T,
                             const float* gate, void* out, int64_t ldo, int64_t out_plane_stride, int H, int dh,
                             float scale, int mode, int gate_col, const float* gru_const, const int32_t* key_lens,
                             const float* bias2d, int64_t bias2d_ld, void* stream) {
    ser_attention_args a = {};
    a.qkv = qkv; a.ld = ld; a.plane_stride = plane_stride; a.q_col = q_col; a.k_col = k_col; a.v_col = v_col; a.B = B;
    a.frame_offs = frame_offs; a.table = table; a.gate = gate; a.max_frames = max_frames; a.table_T = table_T;
    a.out = out; a.ldo = ldo; a.out_plane_stride = out_plane_stride; a.H = H; a.dh = dh; a.scale = scale; a.mode = mode;
    a.gate_col = gate_col; a.gru_const = gru_const; a.key_lens = key_lens; a.bias2d = bias2d; a.bias2d_ld = bias2d_ld;
    return ser_attention_v(&a, stream);
}

extern "C" int ser_attention_v(const ser_attention_args* args, void* stream) {
    if (!args) return ser_fail(-1, "ser_attention: null pointer");
    const void* qkv = args->qkv; const int64_t ld = args->ld, plane_stride = args->plane_stride;
    const int q_col = args->q_col, k_col = args->k_col, v_col = args->v_col, B = args->B, max_frames = args->max_frames;
    const int32_t* frame_offs = args->frame_offs; const float* table = args->table; const int table_T = args->table_T;
    const float* gate = args->gate; void* out = args->out; const int64_t ldo = args->ldo, out_plane_stride = args->out_plane_stride;
    const int H = args->H, dh = args->dh, mode = args->mode, gate_col = args->gate_col; const float scale = args->scale;
    const float* gru_const = args->gru_const; const int32_t* key_lens = args->key_lens;
    const float* bias2d = args->bias2d; const int64_t bias2d_ld = args->bias2d_ld;
    if (!qkv || !frame_offs || !out) return ser_fail(-1, "ser_attention: null pointer");
    if (args->gate_x) {
        if (!table || !gru_const || gate || !args->gate_stat || !args->gate_w || !args->gate_cb)
            return ser_fail(-13, "ser_attention: gate_x needs table, gru_const, gate_stat, gate_w, gate_cb and no gate[]");
        const int np_ = (args->mode == SER_MODE_FP32X || args->mode == SER_MODE_FP16X || args->mode == SER_MODE_FP16Q) ? 2 : 1;
        if ((args->gate_x_ld % 8) || (args->gate_x_plane_stride % 8) || (args->gate_w_plane_stride % 8) || args->gate_x_planes != np_ || (args->dh % 8))
            return ser_fail(-13, "ser_attention: gate_x / gate_w pitch, or gate_x_planes != the planes of mode %d", args->mode);
    }
    if (args->out_mode && args->out_mode != SER_MODE_FP16M) return ser_fail(-14, "ser_attention: out_mode %d (0 or SER_MODE_FP16M)", args->out_mode);
    if (args->out_mode == SER_MODE_FP16M &&
        (args->mode != SER_MODE_FP16X || args->dh != 64 || !args->out_scale || (args->ldo % 64) || args->bias2d || args->out_scale_ld <= 0))
        return ser_fail(-14, "ser_attention: SER_MODE_FP16M context rows need mode FP16X, head dim 64, out_scale, ldo %% 64 == 0 and no bias2d");
    if (B <= 0 || H <= 0 || max_frames <= 0) return ser_fail(-2, "ser_attention: bad B/H/max_frames");
    if (dh % 8 || dh < 8 || dh > 128) return ser_fail(-3, "ser_attention: head dim %d unsupported (multiple of 8, <= 128)", dh);
    if ((ld % 8) || (ldo % 4) || (q_col % 8) || (k_col % 8) || (v_col % 8)) return ser_fail(-4, "ser_attention: misaligned pitches/columns");
    if (mode < SER_MODE_BF16 || mode > SER_MODE_FP16Q) return ser_fail(-5, "ser_attention: bad mode %d", mode);
    if ((table != nullptr) != (gate != nullptr || gru_const != nullptr))
        return ser_fail(-6, "ser_attention: the bias table needs a gate (gate[] or gate_col + gru_const) and vice versa");
    if (gate && gru_const) return ser_fail(-9, "ser_attention: give gate[] or gru_const, not both");
    if (gru_const && !args->gate_x && (gate_col < 0 || (gate_col % 2))) return ser_fail(-10, "ser_attention: bad gate_col %d", gate_col);
    if (table && table_T < max_frames) return ser_fail(-7, "ser_attention: bias table built for T=%d < max_frames=%d", table_T, max_frames);
    if (bias2d) {
        if (table || !key_lens || scale > 0.f || dh > 64 || mode == SER_MODE_FP16 || mode == SER_MODE_FP16Q)
            return ser_fail(-11, "ser_attention: bias2d needs key_lens, a pre-scaled q (scale <= 0), dh <= 64, no table, bf16 / fp32x / fp16x");
        if (bias2d_ld < max_frames || (bias2d_ld % ABKV))
            return ser_fail(-12, "ser_attention: bias2d_ld=%lld must be a multiple of %d and >= max_frames", (long long)bias2d_ld, ABKV);
    }
    // padded head dim of the LDS images and MFMA loops: 64, 96 (head dims 72 .. 96: HuBERT-xlarge's 80 runs 6 + 6 k-steps and 3 output column
    // blocks instead of the 8 + 8 and 4 of the 128-wide form it used through round 3) or 128
    const int dhp = dh <= 64 ? 64 : (dh <= 96 ? 96 : 128);
    const int np = (mode == SER_MODE_FP32X || mode == SER_MODE_FP16X || mode == SER_MODE_FP16Q) ? 2 : 1;      // planes of q / k
    const int npv = (mode == SER_MODE_FP32X || mode == SER_MODE_FP16X) ? 2 : 1;                                 // planes of v
    const int nwv = 4;
    const int nch = ABKV * (dhp / 8) / (64 * nwv);
    const int nq_tiles = (max_frames + 32 * nwv - 1) / (32 * nwv);
    const unsigned nblocks = (unsigned)(((H * B + 7) / 8) * 8 * nq_tiles);
    // high-occupancy form (see the kernel): single-plane 64-wide launches with more than two blocks per CU to place that fit ONE round at
    // four per CU (measured: 16 x 16 heads x 300 / 499 frames and the ragged mix gain 8 - 11 %; Whisper's 1 920-block launches run several
    // rounds either way and LOSE 1 - 3 % on their step with it: left on the low-occupancy form)
    bool occ = dhp == 64 && np == 1 && !bias2d && nblocks > 2u * 256u && nblocks <= 4u * 256u;
    int nbuf = (!occ && nch * (np + npv) <= 8) ? 2 : 1;
    // copy stride == 16 (mod 64) floats: the 4 shifted copies x the 4 query phases of a ds_read_b128
    // lane group then land on 16 distinct 4-bank slots (a multiple of 64 made them 2-way conflicts)
    int bias_stride = 0;
    if (table) {
        bias_stride = ((max_frames + 32 * nwv + 2 * ABKV + 3 + 4) / 4) * 4;   // window of one query block (+ the 0..3 alignment slots), not all 2T-1 distances
        bias_stride += (16 - (bias_stride & 63) + 64) & 63;
    }
    size_t lds = (size_t)nbuf * (np + npv) * ABKV * dhp * 2 + (size_t)4 * bias_stride * 4;
    // ... and only while four blocks' LDS fit a CU: a long utterance's bias window (B = 1, ~8 000 frames: 1 008 blocks, 133 KiB of window) would
    // otherwise take the single-buffer 128-register form at ONE block per CU, the slow combination
    if (occ && lds > 40 * 1024) {
        occ = false;
        nbuf = (nch * (np + npv) <= 8) ? 2 : 1;
        lds = (size_t)nbuf * (np + npv) * ABKV * dhp * 2 + (size_t)4 * bias_stride * 4;
    }
    // a bias window that does not fit (utterances beyond ~2 min; ~1.5 min in the two-plane modes): the table is read from global
    // memory instead (GB forms: pre-scaled q, head dim <= 64 -- what the WavLM encoders use)
    const bool gbias = table && lds > 160 * 1024 && scale <= 0.f && dhp == 64 && nwv == 4;
    if (gbias) {                                                  // (the GB forms have no high-occupancy instantiation: double buffer)
        occ = false;
        nbuf = (nch * (np + npv) <= 8) ? 2 : 1;
        bias_stride = 0;
        lds = (size_t)nbuf * (np + npv) * ABKV * dhp * 2;
    }
    if (lds > 160 * 1024) return ser_fail(-8, "ser_attention: LDS need %zu > 160 KiB (max_frames=%d)", lds, max_frames);
    AttnParams p;
    p.qkv = (const unsigned short*)qkv; p.ld = ld; p.plane = plane_stride;
    p.q_col = q_col; p.k_col = k_col; p.v_col = v_col;
    p.frame_offs = frame_offs; p.key_lens = key_lens; p.table = table; p.table_T = table_T; p.gate = gate;
    p.gru_const = gru_const; p.gate_col = gate_col;
    p.gx = (const unsigned short*)args->gate_x; p.gx_ld = args->gate_x_ld; p.gx_plane = args->gate_x_plane_stride;
    p.gx_planes = args->gate_x_planes; p.gstat = args->gate_stat; p.gw = (const unsigned short*)args->gate_w; p.gw_plane = args->gate_w_plane_stride;
    p.gcb = args->gate_cb;
    p.out = (unsigned short*)out; p.ldo = ldo; p.out_plane = out_plane_stride;
    p.H = H; p.dh = dh; p.bias_stride = bias_stride; p.scale = scale;
    p.bias2d = bias2d; p.b2d_ld = bias2d_ld; p.b2d_T = max_frames;
    p.out_scale = args->out_mode == SER_MODE_FP16M ? (unsigned*)args->out_scale : nullptr; p.out_scale_ld = args->out_scale_ld;
#ifdef SER_ATTN_DBG
    p.dbg = (unsigned long long*)ser_attn_dbg_ptr;
#endif
    p.B = B; p.nq = (max_frames + 32 * nwv - 1) / (32 * nwv);
    dim3 grid((unsigned)(((H * B + 7) / 8) * 8 * p.nq), 1, 1);
    p.nitems = (int)grid.x;
    hipStream_t s = (hipStream_t)stream;
    const bool pre = scale <= 0.f;
#ifdef SER_EXPERIMENTS
    if (!gbias) {                                                 // round-4 experiment (make EXPERIMENTS=1; SER_ATTN_RESIDENT=1): K / V of a whole
        const int r = ser_attention_resident(p, mode, max_frames, s);      // (utterance, head) resident in LDS, attention_res.hip -- not faster, see its header
        if (r <= 0) return r;
    }
#endif
#define SER_ATTN_O(D_, M_, O_) (pre ? (table ? launch_attention<D_, M_, true, true, 4, false, false, O_>(p, grid, lds, s) : launch_attention<D_, M_, true, false, 4, false, false, O_>(p, grid, lds, s)) \
                                    : (table ? launch_attention<D_, M_, false, true, 4, false, false, O_>(p, grid, lds, s) : launch_attention<D_, M_, false, false, 4, false, false, O_>(p, grid, lds, s)))
#define SER_ATTN(D_, M_) SER_ATTN_O(D_, M_, false)
    if (gbias) {
        switch (mode) {
            case SER_MODE_BF16:  return launch_attention<64, SER_MODE_BF16, true, true, 4, false, true>(p, grid, lds, s);
            case SER_MODE_FP16:  return launch_attention<64, SER_MODE_FP16, true, true, 4, false, true>(p, grid, lds, s);
            case SER_MODE_FP32X: return launch_attention<64, SER_MODE_FP32X, true, true, 4, false, true>(p, grid, lds, s);
            case SER_MODE_FP16X: return launch_attention<64, SER_MODE_FP16X, true, true, 4, false, true>(p, grid, lds, s);
            default:             return launch_attention<64, SER_MODE_FP16Q, true, true, 4, false, true>(p, grid, lds, s);
        }
    }
    if (bias2d)
        return mode == SER_MODE_FP32X ? launch_attention<64, SER_MODE_FP32X, true, false, 4, true>(p, grid, lds, s)
             : mode == SER_MODE_FP16X ? launch_attention<64, SER_MODE_FP16X, true, false, 4, true>(p, grid, lds, s)
                                      : launch_attention<64, SER_MODE_BF16, true, false, 4, true>(p, grid, lds, s);
#define SER_ATTN_D(M_) (dhp == 64 ? SER_ATTN(64, M_) : (dhp == 96 ? SER_ATTN(96, M_) : SER_ATTN(128, M_)))
    if (occ) return mode == SER_MODE_FP16 ? SER_ATTN_O(64, SER_MODE_FP16, true) : SER_ATTN_O(64, SER_MODE_BF16, true);
    if (mode == SER_MODE_FP16) return SER_ATTN_D(SER_MODE_FP16);
    if (mode == SER_MODE_FP16X || mode == SER_MODE_FP16Q) {      // "f16a" / "f16q": the host always pre-scales q; only the PRE forms are built
        if (!pre) return ser_fail(-13, "ser_attention: FP16X / FP16Q need a pre-scaled q (scale <= 0)");
#define SER_ATTN_X(D_, M_) (table ? launch_attention<D_, M_, true, true>(p, grid, lds, s) : launch_attention<D_, M_, true, false>(p, grid, lds, s))
#define SER_ATTN_XD(M_) (dhp == 64 ? SER_ATTN_X(64, M_) : (dhp == 96 ? SER_ATTN_X(96, M_) : SER_ATTN_X(128, M_)))
        if (mode == SER_MODE_FP16X) return SER_ATTN_XD(SER_MODE_FP16X);
        return SER_ATTN_XD(SER_MODE_FP16Q);
#undef SER_ATTN_XD
#undef SER_ATTN_X
    }
    if (np == 1) return SER_ATTN_D(SER_MODE_BF16);
    return SER_ATTN_D(SER_MODE_FP32X);
#undef SER_ATTN_D
#undef SER_ATTN
#undef SER_ATTN_O
}
