// Shared by the two forms of ser_attention: the tiled kernel (attention.hip) and the resident-K/V kernel (attention_res.hip).
#pragma once
#include "ser_common.h"

#define ABQ 128      // query rows per 4-wave block (an 8-wave block takes 256)
#define ABKV 64      // keys per tile
#define LOG2E 1.4426950408889634f

struct AttnParams {
    const unsigned short* qkv;
    int64_t ld, plane;
    int q_col, k_col, v_col;
    const int32_t* frame_offs;
    const int32_t* key_lens;   // optional [B]: keys >= key_lens[b] are padding (text encoders); queries keep all rows
    const float* table;
    int table_T;
    const float* gate;
    const float* gru_const;
    int gate_col;
    // gate pre-activations computed here from the layer input's operand copy (ser_attention_args.gate_x)
    const unsigned short* gx;
    int64_t gx_ld, gx_plane;
    int gx_planes;
    const float* gstat;        // [rows][2] relative mean, rstd (ser_gemm lnstat_out)
    const unsigned short* gw;  // [planes][H][2][dh] folded weights in the operand format of `mode`
    int64_t gw_plane;
    const float* gcb;          // [H][4]
    unsigned short* out;
    int64_t ldo, out_plane;
    int H, dh, B, nq;
    int nitems;           // grid size of the one-block-per-item form (PERSIST blocks walk items up to it)
    int bias_stride;      // floats per shifted bias copy in LDS
    float scale;
    // dense additive bias (DeBERTa's disentangled-attention terms, built by ser_deberta_bias): [B][H][T][b2d_ld] fp32 in the
    // exp2 domain, zero where a pair is masked.  With it, padded QUERY rows (q >= key_lens[b]) follow HF's masked_fill(min) +
    // softmax: every score equal -> the uniform average of all T value rows.
    const float* bias2d;
    int64_t b2d_ld;
    int b2d_T;            // rows per (utterance, head) block of bias2d (= max_frames: uniform-length batches)
    // context rows as SER_MODE_FP16M operands (ser_attention_args.out_mode): block-scale words [D / 64][out_scale_ld], or null
    unsigned* out_scale;
    int64_t out_scale_ld;
#ifdef SER_ATTN_DBG
    unsigned long long* dbg;   // phase timestamps of one wave (tools/attn_phases.py; never in the product build)
#endif
};
#ifdef SER_ATTN_DBG
extern "C" { extern void* ser_attn_dbg_ptr; }
#define DBG_P(i) do { __builtin_amdgcn_sched_barrier(0); dbg_p[i] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } while (0)
#define DBG_T(i) do { __builtin_amdgcn_sched_barrier(0); if (dbg_on) dbg_t[i] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define DBG_T(i) do {} while (0)
#define DBG_P(i) do {} while (0)
#endif

#ifdef SER_ATTN_DBG
static __constant__ int ser_attn_dbg_block_dev = 100;
#endif
// LDS images of a K / V tile, [key][DHP x 16 bit] rows.  DHP = 64 / 128 (128- / 256-byte rows): XOR swizzles of the 16-byte chunk (K, read
// by rows with ds_read_b128) and of the 64-byte unit (V, read transposed with ds_read_b64_tr_b16).  DHP = 96 (192-byte rows = 48 banks,
// head dims 72 .. 96: HuBERT-xlarge's 80): 12 chunks per row are not a power of two, so the K chunk is ROTATED by (key >> 2) & 3 --
// the 16 rows of every ds_read_b128 lane group then start on 16 distinct 4-bank slots (12 r + c' mod 16, enumerated for all groups and
// chunks) -- and V needs no swizzle at all: four consecutive keys' units already sit on four different bank quarters ((3 r + u) mod 4).
template <int DHP>
__device__ __forceinline__ int k_swz(int key, int chunk) {
    if (DHP == 96) {
        const int c = chunk + ((key >> 2) & 3);
        return c >= 12 ? c - 12 : c;
    }
    return DHP == 64 ? (chunk ^ ((key >> 1) & 7)) : (chunk ^ (key & 15));
}
template <int DHP>
__device__ __forceinline__ int v_unit_swz(int key, int unit) {
    if (DHP == 96) return unit;
    return DHP == 64 ? (unit ^ ((key >> 1) & 1)) : (unit ^ (key & 3));
}

struct __attribute__((packed, aligned(4))) f32x4_u { float v[4]; };       // 16-byte load from a 4-byte aligned address

// attention_res.hip: K and V of one (utterance, head) resident in LDS.  Returns 0 when launched, < 0 on error, 1 when the launch
// does not fit this form (the tiled kernel takes it).
int ser_attention_resident(const AttnParams& p, int mode, int max_frames, hipStream_t s);
