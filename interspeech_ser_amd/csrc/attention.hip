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
//     O^T has the query on the lane as well, so the rescale by exp(m_old-m_new) is lane-local.
//   * WavLM's gated relative bias needs only the 2T-1 distinct distances: the head's
//     table row is staged once in LDS and indexed by (key - query).
//   * LDS images: K rows XOR-swizzled for conflict-free ds_read_b128 row reads, V 64-byte
//     units XOR-swizzled so the 4 keys of a transposed read hit 4 different bank quarters.
//   * FP32X mode: every product is the 3-term bf16 split (hi*hi + lo*hi + hi*lo).
#include "ser_common.h"

#define ABQ 128      // query rows per block
#define ABKV 64      // keys per tile

struct AttnParams {
    const unsigned short* qkv;
    int64_t ld, plane;
    int q_col, k_col, v_col;
    const int32_t* frame_offs;
    const float* table;
    int table_T;
    const float* gate;
    unsigned short* out;
    int64_t ldo, out_plane;
    int H, dh;
    float scale;
};

template <int DHP>
__device__ __forceinline__ int k_swz(int key, int chunk) {
    return DHP == 64 ? (chunk ^ ((key >> 1) & 7)) : (chunk ^ (key & 15));
}
template <int DHP>
__device__ __forceinline__ int v_unit_swz(int key, int unit) {
    return DHP == 64 ? (unit ^ ((key >> 1) & 1)) : (unit ^ (key & 3));
}

template <int DHP, int MODE>
__global__ __launch_bounds__(256) void attention_kernel(const AttnParams p) {
    constexpr int NP = (MODE == SER_MODE_FP32X) ? 2 : 1;
    constexpr int RS = DHP * 2;                 // LDS row bytes
    constexpr int KS = DHP / 16;                // QK^T k-steps
    constexpr int DSUB = DHP / 32;              // 32-wide output column blocks
    constexpr int CPR = DHP / 8;                // 16-byte chunks per row
    constexpr int TILE = ABKV * RS;             // bytes of one K or V plane tile
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* ldsK = smem;                          // [NP][64][RS]
    char* ldsV = smem + NP * TILE;              // [NP][64][RS]
    float* ldsB = (float*)(smem + 2 * NP * TILE);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = blockIdx.y, b = blockIdx.z;
    const int row0 = p.frame_offs[b];
    const int T = p.frame_offs[b + 1] - row0;
    const int q0 = blockIdx.x * ABQ;
    if (q0 >= T) return;
    const int dh = p.dh;
    const int hh = lane >> 5, l31 = lane & 31;

    // ---- stage this head's bias row: ldsB[i] = table[h][(table_T-1) - (T-1) + i], i < 2T-1
    if (p.table) {
        const float* trow = p.table + (int64_t)h * (2 * p.table_T - 1) + (p.table_T - T);
        for (int i = tid; i < 2 * T - 1; i += 256) ldsB[i] = trow[i];
    }

    // ---- Q fragments: lane holds Q[q][16*ks + 8*hh + j]
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
                if (d < dh) v = *(const u32x4*)(qrow + pl * p.plane + d);
                qf[pl][ks] = __builtin_bit_cast(bf16x8, v);
            }
    }
    const float gq = p.gate ? p.gate[(int64_t)(row0 + qc) * p.H + h] : 0.f;

    f32x16 ot[DSUB];
#pragma unroll
    for (int i = 0; i < DSUB; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) ot[i][r] = 0.f;
    float m_run = -1e30f, l_run = 0.f;

    const int nkt = (T + ABKV - 1) / ABKV;
    for (int kt = 0; kt < nkt; ++kt) {
        __syncthreads();                                            // previous tile fully consumed
        // ---- stage K and V tiles (zero-filled beyond T and beyond dh)
        for (int c = tid; c < ABKV * CPR; c += 256) {
            const int key = c / CPR, ch = c - key * CPR;
            const int kg = kt * ABKV + key;
            const bool ok = (kg < T) && (ch * 8 < dh);
            const unsigned short* src = p.qkv + (int64_t)(row0 + (kg < T ? kg : T - 1)) * p.ld + h * dh + ch * 8;
#pragma unroll
            for (int pl = 0; pl < NP; ++pl) {
                u32x4 kv = {0u, 0u, 0u, 0u}, vv = {0u, 0u, 0u, 0u};
                if (ok) {
                    kv = *(const u32x4*)(src + p.k_col + pl * p.plane);
                    vv = *(const u32x4*)(src + p.v_col + pl * p.plane);
                }
                *(u32x4*)(ldsK + pl * TILE + key * RS + (k_swz<DHP>(key, ch) << 4)) = kv;
                const int unit = ch >> 2;
                *(u32x4*)(ldsV + pl * TILE + key * RS + (v_unit_swz<DHP>(key, unit) << 6) + ((ch & 3) << 4)) = vv;
            }
        }
        __syncthreads();

        // ---- S^T = K Q^T  (rows = keys in registers, col = query on the lane)
        f32x16 st[2];
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
#pragma unroll
            for (int r = 0; r < 16; ++r) st[sub][r] = 0.f;
            const int key = sub * 32 + l31;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const int off = key * RS + (k_swz<DHP>(key, ks * 2 + hh) << 4);
                const bf16x8 kh = *(const bf16x8*)(ldsK + off);
                st[sub] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kh, qf[0][ks], st[sub], 0, 0, 0);
                if (NP == 2) {
                    const bf16x8 kl = *(const bf16x8*)(ldsK + TILE + off);
                    st[sub] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kl, qf[0][ks], st[sub], 0, 0, 0);
                    st[sub] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kh, qf[NP - 1][ks], st[sub], 0, 0, 0);
                }
            }
        }

        // ---- scores, online softmax (lane-local rows)
        float mloc = -1e30f;
#pragma unroll
        for (int sub = 0; sub < 2; ++sub)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = kt * ABKV + sub * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
                float v = st[sub][r] * p.scale;
                if (p.table) v += gq * ldsB[key < T ? key - qc + (T - 1) : 0];
                v = key < T ? v : -INFINITY;
                st[sub][r] = v;
                mloc = fmaxf(mloc, v);
            }
        mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
        const float m_new = fmaxf(m_run, mloc);
        const float alpha = __expf(m_run - m_new);
        float lsum = 0.f;
#pragma unroll
        for (int sub = 0; sub < 2; ++sub)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float e = __expf(st[sub][r] - m_new);
                st[sub][r] = e;
                lsum += e;
            }
        lsum += __shfl_xor(lsum, 32, 64);
        l_run = l_run * alpha + lsum;
        m_run = m_new;
#pragma unroll
        for (int i = 0; i < DSUB; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) ot[i][r] *= alpha;

        // ---- O^T += V^T P^T : accumulator registers 8s..8s+7 are the B fragment of k-step s
#pragma unroll
        for (int sub = 0; sub < 2; ++sub)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                bf16x8 ph, plo;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float e = st[sub][8 * s + j];
                    const __bf16 hi = (__bf16)e;
                    ph[j] = hi;
                    if (NP == 2) plo[j] = (__bf16)(e - (float)hi);
                }
                // transposed V reads: lane (qq,pp) of its 16-lane group addresses key kb+qq, 8 bytes at pp
                const int g = lane >> 4, qq = (lane >> 2) & 3, pp = lane & 3;
                const int kb = sub * 32 + s * 16 + 4 * (g >> 1);
#pragma unroll
                for (int ds = 0; ds < DSUB; ++ds) {
                    const int key0 = kb + qq, key1 = kb + 8 + qq;
                    const int o0 = key0 * RS + (v_unit_swz<DHP>(key0, ds) << 6) + ((g & 1) << 5) + (pp << 3);
                    const int o1 = key1 * RS + (v_unit_swz<DHP>(key1, ds) << 6) + ((g & 1) << 5) + (pp << 3);
                    const s16x4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(ldsV + o0));
                    const s16x4 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(ldsV + o1));
                    typedef __attribute__((ext_vector_type(8))) short s16x8;
                    const s16x8 av = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
                    const bf16x8 vh = __builtin_bit_cast(bf16x8, av);
                    ot[ds] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vh, ph, ot[ds], 0, 0, 0);
                    if (NP == 2) {
                        const s16x4 b0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(ldsV + TILE + o0));
                        const s16x4 b1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(ldsV + TILE + o1));
                        const s16x8 bv = {b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
                        const bf16x8 vl = __builtin_bit_cast(bf16x8, bv);
                        ot[ds] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vl, ph, ot[ds], 0, 0, 0);
                        ot[ds] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vh, plo, ot[ds], 0, 0, 0);
                    }
                }
            }
    }

    // ---- epilogue: O[q][d] = O^T[d][q] / l ; lane owns query q, 4 consecutive d per register quad
    if (q < T) {
        const float inv = 1.0f / l_run;
        unsigned short* orow = p.out + (int64_t)(row0 + q) * p.ldo + h * dh;
#pragma unroll
        for (int ds = 0; ds < DSUB; ++ds)
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                const int d = ds * 32 + 8 * r4 + 4 * hh;
                if (d < dh)
                    store_act4<MODE>(orow + d, p.out_plane, ot[ds][4 * r4] * inv, ot[ds][4 * r4 + 1] * inv,
                                     ot[ds][4 * r4 + 2] * inv, ot[ds][4 * r4 + 3] * inv);
            }
    }
}

extern "C" int ser_attention(const void* qkv, int64_t ld, int64_t plane_stride, int q_col, int k_col, int v_col,
                             const int32_t* frame_offs, int B, int max_frames, const float* table, int table_T,
                             const float* gate, void* out, int64_t ldo, int64_t out_plane_stride, int H, int dh,
                             float scale, int mode, void* stream) {
    if (!qkv || !frame_offs || !out) return ser_fail(-1, "ser_attention: null pointer");
    if (B <= 0 || H <= 0 || max_frames <= 0) return ser_fail(-2, "ser_attention: bad B/H/max_frames");
    if (dh % 8 || dh < 8 || dh > 128) return ser_fail(-3, "ser_attention: head dim %d unsupported (multiple of 8, <= 128)", dh);
    if ((ld % 8) || (ldo % 4) || (q_col % 8) || (k_col % 8) || (v_col % 8)) return ser_fail(-4, "ser_attention: misaligned pitches/columns");
    if (mode != SER_MODE_BF16 && mode != SER_MODE_FP32X) return ser_fail(-5, "ser_attention: bad mode %d", mode);
    if ((table != nullptr) != (gate != nullptr)) return ser_fail(-6, "ser_attention: table and gate must be given together");
    if (table && table_T < max_frames) return ser_fail(-7, "ser_attention: bias table built for T=%d < max_frames=%d", table_T, max_frames);
    const int dhp = dh <= 64 ? 64 : 128;
    const int np = mode == SER_MODE_FP32X ? 2 : 1;
    const size_t lds = (size_t)2 * np * ABKV * dhp * 2 + (table ? (size_t)(2 * max_frames) * 4 : 0);
    if (lds > 65536) return ser_fail(-8, "ser_attention: LDS need %zu > 64 KiB (max_frames=%d)", lds, max_frames);
    AttnParams p;
    p.qkv = (const unsigned short*)qkv; p.ld = ld; p.plane = plane_stride;
    p.q_col = q_col; p.k_col = k_col; p.v_col = v_col;
    p.frame_offs = frame_offs; p.table = table; p.table_T = table_T; p.gate = gate;
    p.out = (unsigned short*)out; p.ldo = ldo; p.out_plane = out_plane_stride;
    p.H = H; p.dh = dh; p.scale = scale;
    dim3 grid((max_frames + ABQ - 1) / ABQ, H, B), block(256);
    hipStream_t s = (hipStream_t)stream;
    if (dhp == 64 && np == 1) hipLaunchKernelGGL((attention_kernel<64, SER_MODE_BF16>), grid, block, lds, s, p);
    else if (dhp == 64) hipLaunchKernelGGL((attention_kernel<64, SER_MODE_FP32X>), grid, block, lds, s, p);
    else if (np == 1) hipLaunchKernelGGL((attention_kernel<128, SER_MODE_BF16>), grid, block, lds, s, p);
    else hipLaunchKernelGGL((attention_kernel<128, SER_MODE_FP32X>), grid, block, lds, s, p);
    return ser_check_launch("ser_attention");
}
