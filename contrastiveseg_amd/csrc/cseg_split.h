// Split-operand arithmetic of the matrix-core convolutions: an fp32 operand is written as a short sum of 16-bit pieces,
// the product of two operands is accumulated in fp32 from the piece products that matter. Two forms:
//
//   SplitBF16x6 ("bf16x6", round 2): three bf16 pieces (8 + 8 + 8 mantissa bits, exact split), six products
//       a0b0 + (a0b1 + a1b0) + (a0b2 + a1b1 + a2b0); dropped terms <= 3 * 2^-24 of the product. Roof 2500 / 6 = 417 TFLOP/s.
//   SplitF16x3 ("f16x3", round 3): two fp16 pieces of the operand scaled by a power of two (11 + 11 mantissa bits:
//       hi = rn16(s x), lo = rn16(s x - hi), |s x - hi - lo| <= 2^-22 |s x|), three products a0b0 + (a0b1 + a1b0); the
//       dropped a1b1 is <= 2^-22 of the product. Roof 2500 / 3 = 833 TFLOP/s of fp32-equivalent work.
//       The scale s = 2^k is chosen per TENSOR from max|x| so that max|s x| lies in [2^14, 2^15) (fp16 overflows at
//       2^16): a power of two changes no mantissa bit, and it is undone exactly in the epilogue (acc * 2^-(ka + kb)).
//       fp16 keeps subnormals (checked on the MI355X: tools/probes/mfma_f16_probe.hip), so an element that is 2^17 times
//       smaller than the tensor's maximum still has a lo piece with an absolute error of 2^-25 (scaled units), i.e. a
//       relative error <= 2^-22 of that element; smaller elements lose relative precision gradually, but their absolute
//       error stays <= 2^-40 of the tensor's maximum.
//   Accuracy of the whole network in either arithmetic, fp64 as truth (tools/split_bf16_probe.py, reference HRNet-W48,
//   max |logit error| at |logit| <= 1.97): fp32 4.3e-5, bf16x6 2.1e-5, f16x3 5.1e-5 (rms 9.6e-6 vs fp32's 9.1e-6),
//   bf16x3 1.4e-3, TF32 9.6e-2. Both forms are dominated by the fp32 accumulation, like fp32 itself.
//
// A kernel is written once against the traits below: AR::NP pieces per operand, AR::NTERMS MFMAs per (A tile, B tile,
// K-step), term t multiplies A piece AR::ta(t) with B piece AR::tb(t) (smallest terms first).
#pragma once
#include "cseg_common.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));


// max|x| of a tensor is handed around as a RECORD of CSEG_AMAX_SLOTS words, CSEG_AMAX_STRIDE words apart (include/cseg_hip.h),
// each the BIT PATTERN of a non-negative float (monotone as an unsigned integer, so a plain atomicMax accumulates it; the slots
// spread the atomics of thousands of producer blocks over 32 cache lines). Scale of the f16x3 split: 2^k with
// k = 14 - floor(log2 max|x|). Must be called by EVERY thread of the block (before any early exit): lane l reads slot l mod 32
// and the 32-lane groups reduce with shuffles.
__device__ __forceinline__ unsigned split_amax_exp(const unsigned* __restrict__ amax_rec) {
    unsigned v = amax_rec ? amax_rec[(threadIdx.x & (CSEG_AMAX_SLOTS - 1)) * CSEG_AMAX_STRIDE] : 0x3f800000u;
#pragma unroll
    for (int o = CSEG_AMAX_SLOTS / 2; o > 0; o >>= 1) v = max(v, (unsigned)__shfl_xor((int)v, o, 64));
    const unsigned e = v >> 23 & 0xffu;                                          // biased exponent of max|x|
    return e < 15u ? 15u : (e > 253u ? 253u : e);                                // all-zero / tiny / non-finite tensors: any finite scale
}
// a block's maximum into its slot of the record: at most one atomic per block, none when the slot already holds more (a stale
// read only costs a redundant atomic: the words never decrease)
__device__ __forceinline__ void amax_publish_block(unsigned block_max, unsigned* __restrict__ amax_rec) {
    unsigned* slot = amax_rec + (blockIdx.x & (CSEG_AMAX_SLOTS - 1)) * CSEG_AMAX_STRIDE;
    if (block_max > *reinterpret_cast<volatile unsigned*>(slot)) atomicMax(slot, block_max);
}
// The maximum of a 256-thread block whose waves each hold a candidate (bit pattern of a non-negative float, already reduced over the wave):
// combined through LDS, ONE look at the slot and at most one atomic per block. Measured on the producers of round 6 (MI355X, kernel
// time inside the step): with one look + atomic per WAVE the 32 words of a record become a hot spot -- 184 000 requests to 32 cache lines in
// the upsample + concat kernel, +24 us; the fused sum of the exchange units (12 000 blocks of 1 024 pixels) took 34.7 us per call instead of
// 24.7, more than the separate cseg_amax_f32 pass it was meant to replace (8 us), whether the look came first, last or not at all.
__device__ __forceinline__ void amax_publish_waves(unsigned wave_max, unsigned* __restrict__ amax_rec) {
    __shared__ unsigned cseg_amax_red[4];
    if ((threadIdx.x & 63) == 0) cseg_amax_red[(threadIdx.x >> 6) & 3] = wave_max;
    __syncthreads();
    if (threadIdx.x == 0)
        amax_publish_block(max(max(cseg_amax_red[0], cseg_amax_red[1]), max(cseg_amax_red[2], cseg_amax_red[3])), amax_rec);
}
__device__ __forceinline__ float split_scale_of(unsigned e) { return __builtin_bit_cast(float, (268u - e) << 23); }     // 2^(141 - e)
__device__ __forceinline__ float split_unscale_of(unsigned e) { return __builtin_bit_cast(float, (e - 14u) << 23); }    // 2^(e - 141)

struct SplitBF16x6 {
    static constexpr int ID = CSEG_ARITH_BF16X6;
    static constexpr int NP = 3, NTERMS = 6;
    static constexpr bool SCALED = false;
    typedef bf16x8 frag_t;
    __host__ __device__ static constexpr int ta(int t) { return t == 0 ? 2 : t == 1 ? 0 : t == 2 ? 1 : t == 3 ? 1 : 0; }
    __host__ __device__ static constexpr int tb(int t) { return t == 0 ? 0 : t == 1 ? 2 : t == 2 ? 1 : t == 3 ? 0 : t == 4 ? 1 : 0; }
    __device__ static __forceinline__ f32x4 mfma(frag_t a, frag_t b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    }
    __device__ static __forceinline__ void split(float v, float /*scale*/, unsigned short (&p)[3]) {
        const __bf16 bh = (__bf16)v;
        const float r1 = v - (float)bh;            // exact
        const __bf16 bm = (__bf16)r1;
        const float r2 = r1 - (float)bm;           // exact
        const __bf16 bl = (__bf16)r2;
        p[0] = __builtin_bit_cast(unsigned short, bh);
        p[1] = __builtin_bit_cast(unsigned short, bm);
        p[2] = __builtin_bit_cast(unsigned short, bl);
    }
    // two values -> one packed dword per piece (a in the low half)
    __device__ static __forceinline__ void split2(float a, float b, float scale, unsigned (&d)[3]) {
        unsigned short pa[3], pb[3];
        split(a, scale, pa);
        split(b, scale, pb);
#pragma unroll
        for (int p = 0; p < 3; ++p) d[p] = pa[p] | ((unsigned)pb[p] << 16);
    }
};

struct SplitF16x3 {
    static constexpr int ID = CSEG_ARITH_F16X3;
    static constexpr int NP = 2, NTERMS = 3;
    static constexpr bool SCALED = true;
    typedef f16x8 frag_t;
    __host__ __device__ static constexpr int ta(int t) { return t == 0 ? 1 : 0; }
    __host__ __device__ static constexpr int tb(int t) { return t == 1 ? 1 : 0; }
    __device__ static __forceinline__ f32x4 mfma(frag_t a, frag_t b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    }
    __device__ static __forceinline__ void split(float v, float scale, unsigned short (&p)[2]) {
        const float t = v * scale;                 // exact (power of two)
        const _Float16 h = (_Float16)t;
        const _Float16 l = (_Float16)(t - (float)h);
        p[0] = __builtin_bit_cast(unsigned short, h);
        p[1] = __builtin_bit_cast(unsigned short, l);
    }
    // two values -> one packed dword per piece. Written on half2 vectors so that the compiler uses gfx950's packed instructions
    // (v_pk_mul_f32, v_cvt_pk_f16_f32, v_pk_fma_f32): 12 VALU instructions per four values instead of 24 with scalar converts
    // and shift / or packing -- the same arithmetic, bit for bit. It matters because the staging waves share their SIMDs with the
    // waves that issue MFMAs, and VALU and MFMA issue do not overlap there.
    __device__ static __forceinline__ void split2(float a, float b, float scale, unsigned (&d)[2]) {
        typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
        const float ta = a * scale, tb = b * scale;            // exact (power of two)
        const _Float16 ha = (_Float16)ta, hb = (_Float16)tb;
        const _Float16 la = (_Float16)(ta - (float)ha), lb = (_Float16)(tb - (float)hb);
        d[0] = __builtin_bit_cast(unsigned, h2_t{ha, hb});
        d[1] = __builtin_bit_cast(unsigned, h2_t{la, lb});
    }
};

__device__ __forceinline__ void split_cells8_f16_lean(const float (&v)[8], float scale, uint4 (&out)[2]);

// eight values -> AR::NP cells of 16 bytes (element j of piece p in half-word j of out[p])
template <class AR>
__device__ __forceinline__ void split_cells8(const float (&v)[8], float scale, uint4 (&out)[AR::NP]) {
#ifndef CSEG_SPLIT_CLASSIC                                  // (A/B builds of the library only: the compiler's own instruction choice)
    if constexpr (AR::ID == CSEG_ARITH_F16X3) {            // round 4: the 16-instruction form below, the same pieces bit for bit
        split_cells8_f16_lean(v, scale, out);
        return;
    }
#endif
    unsigned d[4][AR::NP];
#pragma unroll
    for (int j = 0; j < 4; ++j) AR::split2(v[2 * j], v[2 * j + 1], scale, d[j]);
#pragma unroll
    for (int p = 0; p < AR::NP; ++p) out[p] = make_uint4(d[0][p], d[1][p], d[2][p], d[3][p]);
}

// Round 4: the f16x3 split of eight values in 16 VALU instructions instead of the
// 24 - 36 the compiler makes of split_cells8 (it converts the hi piece back to fp32 and re-packs, or computes it twice). Per pair:
// v_pk_mul_f32 (t = v * s), v_cvt_pk_f16_f32 (hi), and the lo piece straight from the operands with the mixed-precision FMA --
// lo = rn16(v * s - hi), where v * s - hi is exact in fp32, so nothing is rounded twice: bit-identical to SplitF16x3::split2.
// VALU instructions of ANY wave are paid on top of the MFMA time of its SIMD, so the count matters.
__device__ __forceinline__ unsigned split_lo_pair_f16(float a, float b, float s, unsigned hi_pk) {
#if defined(__AMDGCN__)
    unsigned lo;
    asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(lo) : "v"(a), "v"(s), "v"(hi_pk));
    asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(lo) : "v"(b), "v"(s), "v"(hi_pk));
    return lo;
#else   // host pass / CPU emulation of the execution model (tests/emu): the same arithmetic in plain C++
    typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
    const h2_t h = __builtin_bit_cast(h2_t, hi_pk);
    const float ra = __builtin_fmaf(a, s, -(float)h[0]), rb = __builtin_fmaf(b, s, -(float)h[1]);
    return __builtin_bit_cast(unsigned, h2_t{(_Float16)ra, (_Float16)rb});
#endif
}
// eight values -> the hi and lo cells of 16 bytes; `scale` may be 0 (a pixel outside the image: every piece is +-0, finite inputs)
__device__ __forceinline__ void split_cells8_f16_lean(const float (&v)[8], float scale, uint4 (&out)[2]) {
    typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
    typedef float f2_t __attribute__((ext_vector_type(2)));
    unsigned hi[4], lo[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const f2_t t = (f2_t){v[2 * j], v[2 * j + 1]} * scale;                 // exact (power of two, or zero)
        hi[j] = __builtin_bit_cast(unsigned, __builtin_convertvector(t, h2_t));
        lo[j] = split_lo_pair_f16(v[2 * j], v[2 * j + 1], scale, hi[j]);
    }
    out[0] = make_uint4(hi[0], hi[1], hi[2], hi[3]);
    out[1] = make_uint4(lo[0], lo[1], lo[2], lo[3]);
}

// eight values of ONE pixel that may lie outside the image (`ok` false: all pieces zero). f16x3 folds the mask into the scale
// (v * 0 = +-0 for the finite values a clamped address fetched -- one select instead of eight); the unscaled bf16x6 masks the values.
template <class AR>
__device__ __forceinline__ void split_cells8_masked(const float (&v)[8], bool ok, float scale, uint4 (&out)[AR::NP]) {
#ifndef CSEG_SPLIT_CLASSIC
    if constexpr (AR::ID == CSEG_ARITH_F16X3) {
        split_cells8_f16_lean(v, ok ? scale : 0.f, out);
    } else
#endif
    {
        float m[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) m[j] = ok ? v[j] : 0.f;
        split_cells8<AR>(m, scale, out);
    }
}

// four values -> AR::NP cells of 8 bytes
template <class AR>
__device__ __forceinline__ void split_cells4(const float4& v, float scale, uint2 (&out)[AR::NP]) {
    // (the fma_mix form of split_cells8 was tried here too and made the weight gradients 6-7 % SLOWER -- their loader waves stage
    // underneath the consumers' MFMAs, and there the mixed-precision FMAs cost more than the packed sequence the compiler picks:
    // 50.3 vs 46.9 us at 48 channels, 6.49 vs 6.13 ms at 720, A/B/A/B on one box, profiles/r04_split_ab.jsonl)
    unsigned d[2][AR::NP];
    AR::split2(v.x, v.y, scale, d[0]);
    AR::split2(v.z, v.w, scale, d[1]);
#pragma unroll
    for (int p = 0; p < AR::NP; ++p) out[p] = make_uint2(d[0][p], d[1][p]);
}
