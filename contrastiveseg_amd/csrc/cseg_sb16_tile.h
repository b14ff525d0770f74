// Tile pieces shared by the 16-channel-chunk 3x3 kernels (conv3x3_sb16.hip: one-tile, persistent, 8-row, dilated) and the grouped
// launch over several convolutions (conv3x3_group.hip): geometry of a 4 x 64-pixel tile and its patch image in LDS, one K-step of a
// wave, the epilogue store. See conv3x3_sb16.hip for the layout these describe.
#pragma once
#include "cseg_pack.h"
#include "cseg_stats.h"

namespace cseg_sb16t {

constexpr int TR = 4;                 // output rows per block (one per wave)
constexpr int TC = 64;                // output columns per block
constexpr int XROWS = TR + 2;
constexpr int XCOLS = TC + 2;         // cell 0 = column x0 - 1
constexpr int CELLS = XROWS * XCOLS;  // 396 pixels per (piece, octet)
constexpr int PLANE = (CELLS + 15) / 16 * 16;     // LDS stride of a (piece, octet) plane: 0 mod 256 bytes, see conv3x3_sb.hip
constexpr int NOCT = 2;               // channel octets per chunk
constexpr int A_ITEMS = NOCT * CELLS; // (octet, pixel) staging items of a 16-channel chunk
constexpr int AU = (A_ITEMS + 511) / 512;     // staging items per thread (512 threads): 2
constexpr int STEPS = 5;              // K-steps per chunk: taps (0,1) (2,3) (4,5) (6,7) (8,-)

__host__ __device__ constexpr int steps16(int Cin) { return pack_steps_c3_16(Cin); }

// One K-step of a wave: 4 pixel tiles x NTW channel tiles x 6 piece products (see conv3x3_sb.hip:sb_kstep)
template <class AR, int NTW, int NTMAX>
__device__ __forceinline__ void sb16_kstep(const uint4* __restrict__ ap, const uint4* __restrict__ bp,
                                           f32x4 (&acc)[4][NTMAX]) {
    typedef typename AR::frag_t frag_t;
    frag_t a[4][AR::NP];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int p = 0; p < AR::NP; ++p) a[mt][p] = __builtin_bit_cast(frag_t, ap[p * NOCT * PLANE + 16 * mt]);
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) {
        frag_t b[AR::NP];
#pragma unroll
        for (int p = 0; p < AR::NP; ++p) b[p] = __builtin_bit_cast(frag_t, bp[(nt * AR::NP + p) * 64]);
#pragma unroll
        for (int t = 0; t < AR::NTERMS; ++t)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) acc[mt][nt] = AR::mfma(a[mt][AR::ta(t)], b[AR::tb(t)], acc[mt][nt]);
    }
}

// accumulator layout: D[m = 4*g + r][n]: pixel column x0 + 16*mt + 4*g + r, channel co0 + 16*nt + n
template <int NTW, int NTMAX>
__device__ __forceinline__ void sb16_store(const f32x4 (&acc)[4][NTMAX], float* __restrict__ ybc,
                                           const float* __restrict__ bias, const float* __restrict__ abc, int co0, size_t plane, int yy,
                                           int x0, int W, int g, int n, float unscale) {
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) {
        const size_t roff = (size_t)(co0 + nt * 16 + n) * plane + (size_t)yy * W;
        float* orow = ybc + roff;
        const float bv = bias ? bias[co0 + nt * 16 + n] : 0.f;
        const bool vec = (W & 3) == 0;              // rows 16-byte aligned (else element by element: cseg_store_row4)
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            const int xx = x0 + 16 * mt + 4 * g;
            f32x4 v = acc[mt][nt] * unscale;
            v += bv;
            cseg_store_row4(orow, abc ? abc + roff : nullptr, xx, W, vec, v);       // epilogue addend: see conv3x3_sb.hip:sb_store
        }
    }
}

}  // namespace cseg_sb16t
