// Shared device helpers for libserhip (gfx950 only: wave64, MFMA, 160 KiB LDS).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/ser_hip.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
typedef __attribute__((ext_vector_type(4))) short s16x4;

#define SER_WAVE 64

// host-side error plumbing (capi.hip)
int ser_fail(int code, const char* fmt, ...);
int ser_check_launch(const char* what);

__device__ __forceinline__ unsigned short f2bf(float x) {
    return __builtin_bit_cast(unsigned short, (__bf16)x);      // v_cvt_pk_bf16_f32: RNE, NaN-safe
}
__device__ __forceinline__ float bf2f(unsigned short h) {
    return __uint_as_float(((unsigned int)h) << 16);
}
__device__ __forceinline__ unsigned int pack_bf2(float a, float b) {
    return (unsigned int)f2bf(a) | ((unsigned int)f2bf(b) << 16);
}
// fp16 operands (SER_MODE_FP16): 11 significand bits instead of bf16's 8 at the same MFMA rate.  The range is the
// price: values are saturated at +-65504 on the way in (a raw residual row of a real checkpoint stays far below).
__device__ __forceinline__ unsigned int pack_h2(float a, float b) {
    const float lim = 65504.0f;
    typedef __attribute__((ext_vector_type(2))) float f2;
    const f2 v = {__builtin_amdgcn_fmed3f(a, -lim, lim), __builtin_amdgcn_fmed3f(b, -lim, lim)};
    return __builtin_bit_cast(unsigned int, __builtin_convertvector(v, f16x2));   // v_cvt_pk_f16_f32, RNE
}
__device__ __forceinline__ float h2f(unsigned short h) {
    return (float)__builtin_bit_cast(_Float16, h);
}
// element format / plane count of a numerics mode's operand tensors
template <int MODE> struct mode_traits {
    static constexpr bool f16 = (MODE == SER_MODE_FP16 || MODE == SER_MODE_FP16X || MODE == SER_MODE_FP16Q || MODE == SER_MODE_FP16M);      // IEEE fp16 elements (else bf16)
    // FP16Q (ser_attention): of q, k.  FP16M: plane 1 is the 8-bit cross-term plane (ser_hip.h), not a 16-bit lo plane -- its users take the mx_* helpers below
    static constexpr int planes = (MODE == SER_MODE_FP32X || MODE == SER_MODE_FP16X || MODE == SER_MODE_FP16Q || MODE == SER_MODE_FP16M) ? 2 : 1;
};
// one 16-bit operand element -> fp32, in the format of MODE's planes
template <int MODE>
__device__ __forceinline__ float elem2f(unsigned short h) {
    if (mode_traits<MODE>::f16) return h2f(h);
    return bf2f(h);
}
// two fp32 -> one packed pair of MODE's 16-bit operand format (single-plane modes)
template <int MODE>
__device__ __forceinline__ unsigned int pack2(float a, float b) {
    if (mode_traits<MODE>::f16) return pack_h2(a, b);
    return pack_bf2(a, b);
}
// x ~= hi + lo with both halves bf16: 16 significand bits survive.
__device__ __forceinline__ void split_bf(float x, unsigned short& hi, unsigned short& lo) {
    hi = f2bf(x);
    lo = f2bf(x - bf2f(hi));
}
// x ~= hi + lo with both halves fp16 (SER_MODE_FP16X): 22 significand bits survive while lo stays a normal fp16 number
// (|x| >= 2^-3); below that lo is an fp16 subnormal and the pair keeps an ABSOLUTE error of 2^-25 -- the MFMA keeps fp16
// subnormal operands (tools/f16_denorm_probe.py).  hi is the plain fp16 copy, so single-product FP16 launches read it as is.
__device__ __forceinline__ void split_h(float x, unsigned short& hi, unsigned short& lo) {
    hi = (unsigned short)(pack_h2(x, 0.f) & 0xffffu);
    lo = (unsigned short)(pack_h2(x - h2f(hi), 0.f) & 0xffffu);
}
// two-plane split in MODE's element format
template <int MODE>
__device__ __forceinline__ void split2(float x, unsigned short& hi, unsigned short& lo) {
    if (mode_traits<MODE>::f16) split_h(x, hi, lo);
    else split_bf(x, hi, lo);
}
// erf by a clamped odd rational minimax x*P(x^2)/Q(x^2) on [-4, 4] (|err| < 5e-7, branch-free,
// 11 FMA + v_rcp_f32): the GEMM epilogues apply GELU to 30M+ elements per launch, where the
// library erff (two ranges, ~2x the instructions) showed up as 20 % of the FFN1 kernel.
__device__ __forceinline__ float erf_fast(float a) {
    const float x = fminf(fmaxf(a, -4.0f), 4.0f);
    const float x2 = x * x;
    float p = -2.72614225801306e-10f;
    p = fmaf(p, x2, 2.77068142495902e-08f);
    p = fmaf(p, x2, -2.10102402082508e-06f);
    p = fmaf(p, x2, -5.69250639462346e-05f);
    p = fmaf(p, x2, -7.34990630326855e-04f);
    p = fmaf(p, x2, -2.95459980854025e-03f);
    p = fmaf(p, x2, -1.60960333262415e-02f);
    float q = -1.45660718464996e-05f;
    q = fmaf(q, x2, -2.13374055278905e-04f);
    q = fmaf(q, x2, -1.68282697438203e-03f);
    q = fmaf(q, x2, -7.37332916720468e-03f);
    q = fmaf(q, x2, -1.42647390514189e-02f);
    return x * p * __builtin_amdgcn_rcpf(q);
}
__device__ __forceinline__ float gelu_erf(float x) {
    return 0.5f * x * (1.0f + erf_fast(x * 0.70710678118654752440f));
}

// Two GELUs at once on packed fp32 math (v_pk_mul_f32 / v_pk_fma_f32): the Horner chains of the
// rational erf are the bulk of the FC1 / conv epilogues, and packed FMAs halve their issue slots.
typedef __attribute__((ext_vector_type(2))) float f32x2;
__device__ __forceinline__ f32x2 gelu_erf2(f32x2 v) {
    const f32x2 a = v * 0.70710678118654752440f;
    f32x2 x;
    x[0] = fminf(fmaxf(a[0], -4.0f), 4.0f);
    x[1] = fminf(fmaxf(a[1], -4.0f), 4.0f);
    const f32x2 x2 = x * x;
    f32x2 p = {-2.72614225801306e-10f, -2.72614225801306e-10f};
    p = __builtin_elementwise_fma(p, x2, (f32x2){2.77068142495902e-08f, 2.77068142495902e-08f});
    p = __builtin_elementwise_fma(p, x2, (f32x2){-2.10102402082508e-06f, -2.10102402082508e-06f});
    p = __builtin_elementwise_fma(p, x2, (f32x2){-5.69250639462346e-05f, -5.69250639462346e-05f});
    p = __builtin_elementwise_fma(p, x2, (f32x2){-7.34990630326855e-04f, -7.34990630326855e-04f});
    p = __builtin_elementwise_fma(p, x2, (f32x2){-2.95459980854025e-03f, -2.95459980854025e-03f});
    p = __builtin_elementwise_fma(p, x2, (f32x2){-1.60960333262415e-02f, -1.60960333262415e-02f});
    f32x2 q = {-1.45660718464996e-05f, -1.45660718464996e-05f};
    q = __builtin_elementwise_fma(q, x2, (f32x2){-2.13374055278905e-04f, -2.13374055278905e-04f});
    q = __builtin_elementwise_fma(q, x2, (f32x2){-1.68282697438203e-03f, -1.68282697438203e-03f});
    q = __builtin_elementwise_fma(q, x2, (f32x2){-7.37332916720468e-03f, -7.37332916720468e-03f});
    q = __builtin_elementwise_fma(q, x2, (f32x2){-1.42647390514189e-02f, -1.42647390514189e-02f});
    f32x2 r;
    r[0] = __builtin_amdgcn_rcpf(q[0]);
    r[1] = __builtin_amdgcn_rcpf(q[1]);
    const f32x2 e = x * p * r;                                  // erf
    const f32x2 h = v * 0.5f;
    return __builtin_elementwise_fma(h, e, h);                  // 0.5 v (1 + erf)
}

// bf16-mode GELU: gelu(v) = v * Phi(v) with Phi(v) ~= sigmoid(v (c1 + c3 v^2 + c5 v^4)) (a Page-style logistic fit of the
// normal CDF, coefficients minimax-fitted against scipy erf on [-9, 9]): |gelu error| <= 2.6e-5 absolute in fp32
// arithmetic, below the bf16 rounding of the result it feeds except for |gelu| < 0.01.  7 VALU + v_exp_f32 + v_rcp_f32
// per element against 17 for a [3/3] rational erf (|error| 8e-6, the previous choice) and 24 for the fp32x-mode erf:
// the conv-0 LayerNorm+GELU epilogue is VALU-bound (135 -> 118 us per 8-utterance group) and so is part of FC1's.
// The constants are -c_k * log2(e) so that v_exp_f32 (2^x) evaluates exp(-y); v is clamped to +-8 inside the polynomial
// only (beyond it the quintic's sign would flip near |v| = 10.7), the result still scales the unclamped v.
__device__ __forceinline__ float gelu_fast(float v) {
    const float x = __builtin_amdgcn_fmed3f(v, -8.0f, 8.0f);
    const float u = x * x;
    float p = 1.01881229e-03f;
    p = fmaf(p, u, -1.06803738e-01f);
    p = fmaf(p, u, -2.30109051e+00f);
    const float e = __builtin_amdgcn_exp2f(x * p);
    return v * __builtin_amdgcn_rcpf(1.0f + e);
}
// the same arithmetic on two values through v_pk_mul / v_pk_fma / v_pk_add_f32 (6 packed + 2 clamps + 4 transcendental issues per pair
// instead of 14 + 4): identical results, element for element.  Step 2 067 / 2 064 / 2 066 -> 2 072 / 2 075 / 2 072 utt/s (tools/lib_ab.sh).
__device__ __forceinline__ f32x2 gelu_fast2(f32x2 v) {
    f32x2 x;
    x[0] = __builtin_amdgcn_fmed3f(v[0], -8.0f, 8.0f);
    x[1] = __builtin_amdgcn_fmed3f(v[1], -8.0f, 8.0f);
    const f32x2 u = x * x;
    f32x2 p = {1.01881229e-03f, 1.01881229e-03f};
    p = __builtin_elementwise_fma(p, u, (f32x2){-1.06803738e-01f, -1.06803738e-01f});
    p = __builtin_elementwise_fma(p, u, (f32x2){-2.30109051e+00f, -2.30109051e+00f});
    const f32x2 y = x * p;
    f32x2 e;
    e[0] = __builtin_amdgcn_exp2f(y[0]);
    e[1] = __builtin_amdgcn_exp2f(y[1]);
    e = e + (f32x2){1.0f, 1.0f};
    f32x2 r;
    r[0] = __builtin_amdgcn_rcpf(e[0]);
    r[1] = __builtin_amdgcn_rcpf(e[1]);
    return v * r;
}
#ifndef SER_GELU_PK
#define SER_GELU_PK 1
#endif
template <bool FAST>
__device__ __forceinline__ f32x2 gelu2(f32x2 v) {
    if constexpr (FAST) {
        if constexpr (SER_GELU_PK) return gelu_fast2(v);
        else return (f32x2){gelu_fast(v[0]), gelu_fast(v[1])};
    } else return gelu_erf2(v);
}

// One MFMA step on 8 + 8 operand elements per lane in MODE's 16-bit format (same lane maps for bf16 and f16).
template <int MODE>
__device__ __forceinline__ f32x4 mfma16(bf16x8 a, bf16x8 b, f32x4 c) {
    if (mode_traits<MODE>::f16)
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
template <int MODE>
__device__ __forceinline__ f32x16 mfma32(bf16x8 a, bf16x8 b, f32x16 c) {
    if (mode_traits<MODE>::f16)
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// Store n (multiple of 4, <= 16) consecutive fp32 values as act (bf16 hi [+ lo]) at element
// pointer `dst` (8-byte aligned).  `plane` = element distance to the lo plane.
template <int MODE>
__device__ __forceinline__ void store_act4(unsigned short* dst, int64_t plane, float a, float b, float c, float d) {
    if (mode_traits<MODE>::planes == 1) {
        u32x2 v = {pack2<MODE>(a, b), pack2<MODE>(c, d)};
        *(u32x2*)dst = v;
    } else {
        unsigned short h0, h1, h2, h3, l0, l1, l2, l3;
        split2<MODE>(a, h0, l0); split2<MODE>(b, h1, l1); split2<MODE>(c, h2, l2); split2<MODE>(d, h3, l3);
        u32x2 vh = {(unsigned)h0 | ((unsigned)h1 << 16), (unsigned)h2 | ((unsigned)h3 << 16)};
        u32x2 vl = {(unsigned)l0 | ((unsigned)l1 << 16), (unsigned)l2 | ((unsigned)l3 << 16)};
        *(u32x2*)dst = vh;
        *(u32x2*)(dst + plane) = vl;
    }
}

// 8 consecutive values -> one 16-byte store per plane (dst 16-byte aligned): half the store
// instructions of two store_act4 -- epilogue store tails are issue-bound, not bandwidth-bound.
template <int MODE>
__device__ __forceinline__ void store_act8(unsigned short* dst, int64_t plane, const float (&v)[8]) {
    if (mode_traits<MODE>::planes == 1) {
        u32x4 o = {pack2<MODE>(v[0], v[1]), pack2<MODE>(v[2], v[3]), pack2<MODE>(v[4], v[5]), pack2<MODE>(v[6], v[7])};
        *(u32x4*)dst = o;
    } else {
        unsigned short h[8], l[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) split2<MODE>(v[i], h[i], l[i]);
        u32x4 oh, ol;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            oh[i] = (unsigned)h[2 * i] | ((unsigned)h[2 * i + 1] << 16);
            ol[i] = (unsigned)l[2 * i] | ((unsigned)l[2 * i + 1] << 16);
        }
        *(u32x4*)dst = oh;
        *(u32x4*)(dst + plane) = ol;
    }
}

// Epilogue store for the natural MFMA accumulator layout.  v[0..3] are 4 consecutive columns c..c+3 of one
// fragment column block, v[4..7] the same lane columns of the next block (16 columns further).  Lanes l and l^16
// (same row, lane-column groups 2j and 2j+1) trade one packed half with v_permlane16_swap_b32 so that each ends
// up with 8 CONSECUTIVE columns: even groups keep block 0 and get c+4..c+7 from the neighbour, odd groups keep
// block 1 and get its c-4..c-1.  16-byte stores, and the four lanes of a row write 64 contiguous bytes.
// dst already points at this lane's 8-column run; every lane of the pair must call (ok only gates the store).
template <int MODE>
__device__ __forceinline__ void store_act8_swap(unsigned short* dst, int64_t plane, bool ok, const float (&v)[8]) {
    if (mode_traits<MODE>::planes == 1) {
        const auto s0 = __builtin_amdgcn_permlane16_swap(pack2<MODE>(v[0], v[1]), pack2<MODE>(v[4], v[5]), false, false);
        const auto s1 = __builtin_amdgcn_permlane16_swap(pack2<MODE>(v[2], v[3]), pack2<MODE>(v[6], v[7]), false, false);
        if (ok) *(u32x4*)dst = (u32x4){s0[0], s1[0], s0[1], s1[1]};
    } else {
        unsigned short h[8], l[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) split2<MODE>(v[i], h[i], l[i]);
        unsigned ph[4], pl[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            ph[i] = (unsigned)h[2 * i] | ((unsigned)h[2 * i + 1] << 16);
            pl[i] = (unsigned)l[2 * i] | ((unsigned)l[2 * i + 1] << 16);
        }
        const auto h0 = __builtin_amdgcn_permlane16_swap(ph[0], ph[2], false, false);
        const auto h1 = __builtin_amdgcn_permlane16_swap(ph[1], ph[3], false, false);
        const auto l0 = __builtin_amdgcn_permlane16_swap(pl[0], pl[2], false, false);
        const auto l1 = __builtin_amdgcn_permlane16_swap(pl[1], pl[3], false, false);
        if (ok) {
            *(u32x4*)dst = (u32x4){h0[0], h1[0], h0[1], h1[1]};
            *(u32x4*)(dst + plane) = (u32x4){l0[0], l1[0], l0[1], l1[1]};
        }
    }
}

// Load 8 consecutive act elements as fp32 (hi [+ lo]).
template <int MODE>
__device__ __forceinline__ void load_act8(const unsigned short* src, int64_t plane, float (&v)[8]) {
    u32x4 h = *(const u32x4*)src;
    if (mode_traits<MODE>::f16) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            v[2 * i] = h2f((unsigned short)(h[i] & 0xffffu));
            v[2 * i + 1] = h2f((unsigned short)(h[i] >> 16));
        }
        if (MODE == SER_MODE_FP16X) {
            const u32x4 l = *(const u32x4*)(src + plane);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                v[2 * i] += h2f((unsigned short)(l[i] & 0xffffu));
                v[2 * i + 1] += h2f((unsigned short)(l[i] >> 16));
            }
        }
        return;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        v[2 * i] = __uint_as_float(h[i] << 16);
        v[2 * i + 1] = __uint_as_float(h[i] & 0xffff0000u);
    }
    if (MODE == SER_MODE_FP32X) {
        u32x4 l = *(const u32x4*)(src + plane);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            v[2 * i] += __uint_as_float(l[i] << 16);
            v[2 * i + 1] += __uint_as_float(l[i] & 0xffff0000u);
        }
    }
}

// ---- SER_MODE_FP16M (include/ser_hip.h): fp16 hi plane + block-scaled e4m3 cross-term plane ---------------------------------------------
typedef __attribute__((ext_vector_type(8))) int i32x8;
#define SER_F16_MAX 65504.0f
// E8M0 code of the smallest power of two s with amax / s <= 448 (e4m3's largest value): ceil(log2(amax / 448)) + 127 in [1, 254].
// (amax / 448 is rounded once: a block whose quotient lands a hair above 448 still converts to 448 -- v_cvt_pk_fp8_f32 rounds up to 464 down.)
__device__ __forceinline__ unsigned mx_code(float amax) {
    const unsigned c = (__float_as_uint(amax * (1.0f / 448.0f)) + 0x7fffffu) >> 23;
    return c < 1u ? 1u : (c > 254u ? 254u : c);
}
__device__ __forceinline__ float mx_inv(unsigned code) { return __uint_as_float((254u - code) << 23); }     // 2^(127 - code)
// four fp32 (already divided by their block scale) -> four e4m3 bytes, element 0 in the low byte (v_cvt_pk_fp8_f32: OCP e4m3fn, RNE)
__device__ __forceinline__ unsigned mx_pack4(float a, float b, float c, float d) {
    int r = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);
    r = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, r, true);
    return (unsigned)r;
}
// max over the four lanes l, l ^ 16, l ^ 32, l ^ 48 (the lanes of one accumulator row in the 16x16 MFMA layouts): two swaps, no LDS
__device__ __forceinline__ float max_rowquad(float v) {
    const auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
    const auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}
// 4 x 4 transpose of one dword per (lane row fq = lane >> 4, register j): afterwards register j of row fq holds what register fq of row j held.
// v_permlane32_swap exchanges the off-diagonal 2 x 2 blocks, v_permlane16_swap transposes inside every 2 x 2 block.
__device__ __forceinline__ void transpose_rowquad(unsigned (&r)[4]) {
    const auto a0 = __builtin_amdgcn_permlane32_swap(r[0], r[2], false, false);
    const auto a1 = __builtin_amdgcn_permlane32_swap(r[1], r[3], false, false);
    const auto b0 = __builtin_amdgcn_permlane16_swap(a0[0], a1[0], false, false);
    const auto b1 = __builtin_amdgcn_permlane16_swap(a0[1], a1[1], false, false);
    r[0] = b0[0]; r[1] = b0[1]; r[2] = b1[0]; r[3] = b1[1];
}
// Row kernels: VPL (4 or 8) consecutive values of one row per lane, 32 / VPL consecutive lanes per scale block, 64 / VPL per 64-column tile.
//   hi_row: the row in plane 0; x8_row: the same row in plane 1 (as bytes); col: the lane's first column (multiple of VPL);
//   srow: scale words of this row (tile t at srow[t * sld]); weight: plane roles (ser_hip.h); every lane of the wave must call (shuffles) --
//   `ok` gates the stores of lanes past the end of the row.  Returns max |value| of the lane (for the fp16 range guard).
template <int VPL>
__device__ __forceinline__ float mx_store_row(const float (&v)[VPL], bool ok, unsigned short* hi_row, unsigned char* x8_row, int col,
                                              uint32_t* srow, int64_t sld, bool weight) {
    static_assert(VPL == 4 || VPL == 8, "4 or 8 values per lane");
    constexpr int LPB = 32 / VPL;
    float lo[VPL], ax = 0.f, al = 0.f;
    unsigned hp[VPL / 2];
#pragma unroll
    for (int i = 0; i < VPL; i += 2) {
        hp[i / 2] = pack_h2(v[i], v[i + 1]);
        lo[i] = v[i] - h2f((unsigned short)(hp[i / 2] & 0xffffu));
        lo[i + 1] = v[i + 1] - h2f((unsigned short)(hp[i / 2] >> 16));
        ax = fmaxf(ax, fmaxf(fabsf(v[i]), fabsf(v[i + 1])));
        al = fmaxf(al, fmaxf(fabsf(lo[i]), fabsf(lo[i + 1])));
    }
    const float amax_lane = ax;
#pragma unroll
    for (int o = 1; o < LPB; o <<= 1) { ax = fmaxf(ax, __shfl_xor(ax, o, 64)); al = fmaxf(al, __shfl_xor(al, o, 64)); }
    const unsigned cx = mx_code(ax), cl = mx_code(al);
    const float ix = mx_inv(cx), il = mx_inv(cl);
    unsigned bx[VPL / 4], bl[VPL / 4];
#pragma unroll
    for (int i = 0; i < VPL; i += 4) {
        bx[i / 4] = mx_pack4(v[i] * ix, v[i + 1] * ix, v[i + 2] * ix, v[i + 3] * ix);
        bl[i / 4] = mx_pack4(lo[i] * il, lo[i + 1] * il, lo[i + 2] * il, lo[i + 3] * il);
    }
    // scale word of the lane's 64-column tile: codes of its two blocks, [P0, P1, Q0, Q1]
    const unsigned cp = weight ? cx : cl, cq = weight ? cl : cx;
    const unsigned cp1 = __shfl_xor(cp, LPB, 64), cq1 = __shfl_xor(cq, LPB, 64);      // the tile's other block
    if (ok) {
        if constexpr (VPL == 8) *(u32x4*)(hi_row + col) = (u32x4){hp[0], hp[1], hp[2], hp[3]};
        else *(u32x2*)(hi_row + col) = (u32x2){hp[0], hp[1]};
        unsigned char* seg = x8_row + (col & ~63) * 2;                               // 128 bytes per 64-column tile
        unsigned char* pp = seg + (col & 63), *qq = pp + 64;
        if constexpr (VPL == 8) {
            *(u32x2*)pp = weight ? (u32x2){bx[0], bx[1]} : (u32x2){bl[0], bl[1]};
            *(u32x2*)qq = weight ? (u32x2){bl[0], bl[1]} : (u32x2){bx[0], bx[1]};
        } else {
            *(unsigned*)pp = weight ? bx[0] : bl[0];
            *(unsigned*)qq = weight ? bl[0] : bx[0];
        }
        if ((col & 63) == 0) srow[(int64_t)(col >> 6) * sld] = cp | (cp1 << 8) | (cq << 16) | (cq1 << 24);
    }
    return amax_lane;
}
// fp16 range guard (ser_hip.h range_flag): bit 0 = a value beyond +-65504 (or a NaN) was rounded to an fp16 operand plane (it saturated),
// bit 1 = a value beyond half that range was.  One rare atomic per lane that saw such a value; nothing otherwise.
__device__ __forceinline__ void range_report(uint32_t* flag, float amax) {
    if (flag && !(amax <= 0.5f * SER_F16_MAX)) atomicOr(flag, !(amax <= SER_F16_MAX) ? 3u : 2u);
}
