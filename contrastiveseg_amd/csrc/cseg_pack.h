// Weight packing of the split-operand convolutions, one element function per packed format (shared by the per-layer pack kernels
// of conv3x3_sb.hip / conv3x3_sb16.hip / conv1x1_sb.hip and by the batched kernel of split_batch.hip, which packs the weights of
// every layer of the network in ONE launch per optimizer step).
// Every format is Wp[co_tile][kstep][nt][piece][lane] of uint4 (8 half-words, element j), lane = 16*g + n; what differs is which
// (input channel, tap) a (kstep, g, j) names. One thread per (co_tile, kstep, nt, lane).
#pragma once
#include "cseg_split.h"

#define CSEG_PACK_C3 0        // conv3x3_sb.hip: 32-channel chunks x 9 taps, 16-channel tail pairs two taps per K-step
#define CSEG_PACK_C3_16 1     // conv3x3_sb16.hip: 16-channel chunks, every K-step pairs two taps
#define CSEG_PACK_C1 2        // conv1x1_sb.hip: 32 input channels per K-step
#define CSEG_PACK_C3_S2T 3    // conv3x3_s2.hip: backward-data operator of the stride-2 convolution (taps grouped by output parity)

__host__ __device__ constexpr int pack_steps_c3(int Cin) { return (Cin / 32) * 9 + ((Cin & 31) ? 5 : 0); }
__host__ __device__ constexpr int pack_steps_c3_16(int Cin) { return (Cin / 16) * 5; }
__host__ __device__ constexpr int pack_steps_c1(int Cin) { return (Cin + 31) / 32; }

template <class AR>
__device__ __forceinline__ void pack_store(const float (&v)[8], float wscale, uint4* __restrict__ wp, size_t slot, int lane) {
    uint4 cells[AR::NP];
    split_cells8<AR>(v, wscale, cells);
    uint4* dst = wp + slot * AR::NP * 64 + lane;
#pragma unroll
    for (int p = 0; p < AR::NP; ++p) dst[64 * p] = cells[p];
}

// w = the forward's [Cout, Cin, 3, 3]; transpose_flip packs the backward-data operator (maps Cout -> Cin channels, mirrored taps)
template <class AR>
__device__ __forceinline__ void pack_elem_c3(const float* __restrict__ w, int Cout, int Cin, int transpose_flip, int NT, float wscale,
                                             uint4* __restrict__ wp, int e) {
    const int conv_in = transpose_flip ? Cout : Cin;
    const int n_full = conv_in / 32, n_steps = pack_steps_c3(conv_in);
    int r = e;
    const int lane = r & 63; r >>= 6;
    const int nt = r % NT; r /= NT;
    const int ks = r % n_steps;
    const int co_tile = r / n_steps;
    const int g = lane >> 4, n = lane & 15;
    const int oc = (co_tile * NT + nt) * 16 + n;           // output channel of THIS convolution
    int tap, ic0;
    if (ks < n_full * 9) { tap = ks % 9; ic0 = (ks / 9) * 32 + 8 * g; }
    else { tap = 2 * (ks - n_full * 9) + (g >> 1); ic0 = n_full * 32 + 8 * (g & 1); }
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int ic = ic0 + j;                            // input channel of THIS convolution
        float t = 0.f;
        if (tap <= 8) {
            if (!transpose_flip) t = w[((size_t)oc * Cin + ic) * 9 + tap];            // w[co][ci][ky][kx]
            else t = w[((size_t)ic * Cin + oc) * 9 + (8 - tap)];                      // w[co=ic][ci=oc][2-ky][2-kx]
        }
        v[j] = t;
    }
    pack_store<AR>(v, wscale, wp, (size_t)(co_tile * n_steps + ks) * NT + nt, lane);
}

// K-step ks = 5*chunk + q: value(co, ci = 16*chunk + 8*(g&1) + j, tap = 2q + (g>>1))   (zero when tap > 8)
template <class AR>
__device__ __forceinline__ void pack_elem_c3_16(const float* __restrict__ w, int Cout, int Cin, int transpose_flip, int NT,
                                                float wscale, uint4* __restrict__ wp, int e) {
    const int conv_in = transpose_flip ? Cout : Cin;
    const int n_steps = pack_steps_c3_16(conv_in);
    int r = e;
    const int lane = r & 63; r >>= 6;
    const int nt = r % NT; r /= NT;
    const int ks = r % n_steps;
    const int co_tile = r / n_steps;
    const int g = lane >> 4, n = lane & 15;
    const int oc = (co_tile * NT + nt) * 16 + n;
    const int chunk = ks / 5, q = ks - chunk * 5;
    const int tap = 2 * q + (g >> 1), ic0 = 16 * chunk + 8 * (g & 1);
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int ic = ic0 + j;
        float t = 0.f;
        if (tap <= 8) {
            if (!transpose_flip) t = w[((size_t)oc * Cin + ic) * 9 + tap];
            else t = w[((size_t)ic * Cin + oc) * 9 + (8 - tap)];
        }
        v[j] = t;
    }
    pack_store<AR>(v, wscale, wp, (size_t)(co_tile * n_steps + ks) * NT + nt, lane);
}

// w = the forward's [Cout, Cin]; transpose = 1 packs the backward-data operator: value(co, ci = 32*kstep + 8g + j), zero beyond
template <class AR>
__device__ __forceinline__ void pack_elem_c1(const float* __restrict__ w, int Cout, int Cin, int transpose, int NT, float wscale,
                                             uint4* __restrict__ wp, int e) {
    const int conv_in = transpose ? Cout : Cin;
    const int n_steps = pack_steps_c1(conv_in);
    int r = e;
    const int lane = r & 63; r >>= 6;
    const int nt = r % NT; r /= NT;
    const int ks = r % n_steps;
    const int co_tile = r / n_steps;
    const int g = lane >> 4, n = lane & 15;
    const int oc = (co_tile * NT + nt) * 16 + n;           // output channel of THIS operator
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int ic = ks * 32 + 8 * g + j;                // input channel of THIS operator
        float t = 0.f;
        if (ic < conv_in) t = transpose ? w[(size_t)ic * Cin + oc] : w[(size_t)oc * Cin + ic];
        v[j] = t;
    }
    pack_store<AR>(v, wscale, wp, (size_t)(co_tile * n_steps + ks) * NT + nt, lane);
}

// Backward-data operator of the 3x3 / stride 2 / pad 1 convolution (conv3x3_s2.hip): dx[ci][2 qy + py][2 qx + px] sums, over the
// output channels co and over the taps whose parity matches -- ky = 1 for py = 0, ky in {0, 2} for py = 1, same in x --
// dy[co][qy + dyy][qx + dxx] * w[co][ci][ky][kx] with dyy = 1 for ky = 0 and 0 otherwise (likewise dxx). 16-channel chunks of co
// like CSEG_PACK_C3_16, five K-steps per chunk, K-step q pairs two taps (lane groups g >> 1 = 0 / 1) of ONE parity class:
//   q = 0: class (py, px) = (0, 0): tap (1, 1) | nothing (zeros)
//   q = 1: class (0, 1): taps (1, 0) | (1, 2)
//   q = 2: class (1, 0): taps (0, 1) | (2, 1)
//   q = 3: class (1, 1): taps (0, 0) | (0, 2)
//   q = 4: class (1, 1): taps (2, 0) | (2, 2)
// value(oc = ci of the convolution, ic = co = 16*chunk + 8*(g&1) + j). w = the forward's [Cout, Cin, 3, 3].
__host__ __device__ constexpr int pack_s2t_tap(int q, int second) {       // tap index ky*3 + kx, -1 = none
    return q == 0 ? (second ? -1 : 4) : q == 1 ? (second ? 5 : 3) : q == 2 ? (second ? 7 : 1) : q == 3 ? (second ? 2 : 0) : (second ? 8 : 6);
}
template <class AR>
__device__ __forceinline__ void pack_elem_c3_s2t(const float* __restrict__ w, int Cout, int Cin, int NT, float wscale,
                                                 uint4* __restrict__ wp, int e) {
    const int n_steps = pack_steps_c3_16(Cout);            // the operator's input channels are the convolution's output channels
    int r = e;
    const int lane = r & 63; r >>= 6;
    const int nt = r % NT; r /= NT;
    const int ks = r % n_steps;
    const int co_tile = r / n_steps;
    const int g = lane >> 4, n = lane & 15;
    const int oc = (co_tile * NT + nt) * 16 + n;           // = ci of the convolution
    const int chunk = ks / 5, q = ks - chunk * 5;
    const int tap = pack_s2t_tap(q, g >> 1), ic0 = 16 * chunk + 8 * (g & 1);
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = tap >= 0 ? w[((size_t)(ic0 + j) * Cin + oc) * 9 + tap] : 0.f;
    pack_store<AR>(v, wscale, wp, (size_t)(co_tile * n_steps + ks) * NT + nt, lane);
}
