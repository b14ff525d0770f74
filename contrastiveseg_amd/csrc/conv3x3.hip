// 3x3 / stride 1 / pad 1 convolution for the narrow HRNet branches (48 and 96 channels at 1/4 and 1/8 resolution),
// forward and backward-data, NCHW fp32, as an implicit GEMM on the fp32 MFMA (v_mfma_f32_16x16x4_f32).
//
// Why a hand-written kernel on a path that otherwise leaves convolutions to MIOpen: the 64 basic-block convolutions
// of the 48-channel branch are the worst-performing shape of the benched step. MIOpen's best solver for them (Winograd
// F(2,3), NCHW) runs at 0.22 / 0.24 ms per forward / backward-data call = 45-50 TFLOP/s of the 157 TFLOP/s fp32 MFMA
// peak (profiles/r02_conv_layout_probe_nchw_vs_channels_last.jsonl) -- the Winograd transforms are HBM-bound at 48
// channels -- while the arithmetic intensity of the direct form (108 flop/B) is 4x past the ridge.
//
// Mapping (one block = 4 waves = 4 image rows x 64 columns x 48 output channels):
//   GEMM M = pixels, N = output channels, K = input channels x 9 taps.
//   A operand (pixels x K): read from an LDS tile of the input [8 channels][6 rows][72 floats] (halo included, row
//       stride 72 so that the four 16-lane groups of a wave -- which walk 4 consecutive input channels -- fall on
//       disjoint bank halves: conflict-free ds_read_b32 with compile-time offsets for tap and channel pair);
//   B operand (K x 48 channels): pre-packed by cseg_conv3x3_pack_weights into the exact per-lane order, so a k-step's
//       three B registers are three coalesced dword loads (L1/L2 resident: every wave reads the same 83 KB);
//   C: 4 pixel tiles x 3 channel tiles of 16x16 per wave = 12 accumulators; lane holds 4 consecutive pixels of one
//       channel per accumulator -> 16-byte stores.
//   The input channels are walked in chunks of 8, double-buffered in LDS: global loads of chunk c+1 are issued before
//   the 216 MFMAs of chunk c and written to the other buffer afterwards; one barrier per chunk.
// Backward-data is the same kernel on weights packed transposed and flipped (dx = conv(dy, W^T flipped)).
// Numerics: v_mfma_f32_16x16x4_f32 is an exact fp32 FMA chain (MI355X guide, section 3); accumulation order is
// input-channel-major, taps inner.
#include "cseg_common.h"
#include <stdlib.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int CK = 8;            // input channels per LDS chunk
constexpr int TR = 4;            // output rows per block (one per wave)
constexpr int TC = 64;           // output columns per block
constexpr int LDW = 72;          // LDS row stride (floats): [3] = left halo, [4..67] = columns, [68] = right halo
constexpr int XROWS = TR + 2;
constexpr int CHUNK_FLOATS = CK * XROWS * LDW;       // 3456
constexpr int CO_T = 48;         // output channels per block (3 MFMA column tiles)
constexpr int KSTEPS = 18;       // per chunk: 9 taps x 2 channel quads

// Packed weights: Wp[co_tile][chunk][kstep][nt][lane]; kstep = tap*2 + cq, lane = 16*g + n:
//   value = Wsrc(co = co_tile*48 + nt*16 + n, ci = chunk*8 + cq*4 + g, tap)
__global__ __launch_bounds__(256) void pack_weights_kernel(const float* __restrict__ w, int Cout, int Cin, int transpose_flip,
                                                           float* __restrict__ wp, int total) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= total) return;
    int r = e;
    const int lane = r & 63; r >>= 6;
    const int nt = r % 3; r /= 3;
    const int ks = r % KSTEPS; r /= KSTEPS;
    const int n_chunks = (transpose_flip ? Cout : Cin) / CK;
    const int chunk = r % n_chunks;
    const int co_tile = r / n_chunks;
    const int tap = ks >> 1, cq = ks & 1, g = lane >> 4, n = lane & 15;
    const int oc = co_tile * CO_T + nt * 16 + n;     // output channel of THIS convolution
    const int ic = chunk * CK + cq * 4 + g;          // input channel of THIS convolution
    float v;
    if (!transpose_flip) {
        v = w[((size_t)oc * Cin + ic) * 9 + tap];                 // w[co][ci][ky][kx]
    } else {
        // backward-data: "output" channels are the forward's input channels; taps are mirrored
        v = w[((size_t)ic * Cin + oc) * 9 + (8 - tap)];           // w[co=ic][ci=oc][2-ky][2-kx]
    }
    wp[e] = v;
}

// Persistent over output tiles: block b walks tiles b, b + gridDim.x, ...; the (tile, chunk) pairs form one software
// pipeline, so the first chunk of the next tile is already in flight while the last chunk of the current one runs and
// the output stores of a tile overlap the MFMAs of the next (prologue / epilogue are paid once per block, not per tile).
__global__ __launch_bounds__(256, 3) void conv3x3_kernel(const float* __restrict__ x, const float* __restrict__ wp,
                                                         int Cin, int Cout, int H, int W, int tiles_x, int tiles_y,
                                                         int n_tiles, float* __restrict__ y) {
    __shared__ __attribute__((aligned(16))) float xs[2][CHUNK_FLOATS];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int g = lane >> 4, n = lane & 15;
    const int n_cot = Cout / CO_T;
    const int n_chunks = Cin / CK;
    const size_t plane = (size_t)H * W;

    // tile index -> (image b, channel tile cot, tile row ty, tile column tx); x is fastest so that neighbouring blocks
    // share halo rows in L2
    auto decode = [&](int t, int& b, int& cot, int& y0, int& x0) {
        const int tx = t % tiles_x; t /= tiles_x;
        const int ty = t % tiles_y; t /= tiles_y;
        cot = t % n_cot;
        b = t / n_cot;
        x0 = tx * TC; y0 = ty * TR;
    };

    // ---- staging assignment: 48 (channel, row) pairs per chunk; 16 float4 per pair + 2 halo scalars
    float4 pv[3];
    float ps = 0.f;
    auto issue_loads = [&](int t, int chunk) {
        int b, cot, y0, x0;
        decode(t, b, cot, y0, x0);
        const float* xc = x + ((size_t)b * Cin + (size_t)chunk * CK) * plane;
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            const int v = tid + 256 * u;
            const int pair = v >> 4, q = v & 15;
            const int ci = pair / XROWS, r = pair - ci * XROWS;
            const int yy = y0 + r - 1, xx = x0 + 4 * q;
            float4 tt = make_float4(0.f, 0.f, 0.f, 0.f);
            if (yy >= 0 && yy < H) {
                const float* p = xc + (size_t)ci * plane + (size_t)yy * W + xx;
                if (xx + 3 < W) tt = *reinterpret_cast<const float4*>(p);
                else {
                    if (xx < W) tt.x = p[0];
                    if (xx + 1 < W) tt.y = p[1];
                    if (xx + 2 < W) tt.z = p[2];
                }
            }
            pv[u] = tt;
        }
        ps = 0.f;
        if (tid < 2 * CK * XROWS) {
            const int pair = tid >> 1, side = tid & 1;
            const int ci = pair / XROWS, r = pair - ci * XROWS;
            const int yy = y0 + r - 1, xx = side ? x0 + TC : x0 - 1;
            if (yy >= 0 && yy < H && xx >= 0 && xx < W) ps = xc[(size_t)ci * plane + (size_t)yy * W + xx];
        }
    };
    auto store_lds = [&](int buf) {
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            const int v = tid + 256 * u;
            const int pair = v >> 4, q = v & 15;
            *reinterpret_cast<float4*>(&xs[buf][pair * LDW + 4 + 4 * q]) = pv[u];
        }
        if (tid < 2 * CK * XROWS) {
            const int pair = tid >> 1, side = tid & 1;
            xs[buf][pair * LDW + (side ? 4 + TC : 3)] = ps;
        }
    };

    f32x4 acc[4][3];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < 3; ++nt) acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

    int t = blockIdx.x;
    if (t >= n_tiles) return;
    issue_loads(t, 0);
    store_lds(0);
    __syncthreads();

    // this lane's A-operand base: channel g of a quad, row = wave (+ky), column = 3 + n (+kx + 16*mt)
    const int a_base = (g * XROWS + wave) * LDW + 3 + n;
    int buf = 0;
    while (t < n_tiles) {
        int b, cot, y0, x0;
        decode(t, b, cot, y0, x0);
        const float* wbase = wp + ((size_t)cot * n_chunks) * (KSTEPS * 3 * 64) + lane;
        const int t_next = t + gridDim.x;
        for (int c = 0; c < n_chunks; ++c) {
            const bool last = c + 1 == n_chunks;
            const bool more = !last || t_next < n_tiles;
            if (more) issue_loads(last ? t_next : t, last ? 0 : c + 1);
            const float* wc = wbase + (size_t)c * (KSTEPS * 3 * 64);
            const float* xa = &xs[buf][a_base];
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int ky = tap / 3, kx = tap - 3 * ky;
#pragma unroll
                for (int cq = 0; cq < 2; ++cq) {
                    const int ks = tap * 2 + cq;
                    const float b0 = wc[(ks * 3 + 0) * 64], b1 = wc[(ks * 3 + 1) * 64], b2 = wc[(ks * 3 + 2) * 64];
                    const int off = (cq * 4 * XROWS + ky) * LDW + kx;
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt) {
                        const float a = xa[off + 16 * mt];
                        acc[mt][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b0, acc[mt][0], 0, 0, 0);
                        acc[mt][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b1, acc[mt][1], 0, 0, 0);
                        acc[mt][2] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b2, acc[mt][2], 0, 0, 0);
                    }
                }
            }
            if (last) {
                // accumulator layout: D[m = 4*g + r][n]: pixel column x0 + 16*mt + 4*g + r, channel cot*48 + 16*nt + n
                const int yy = y0 + wave;
                if (yy < H) {
#pragma unroll
                    for (int nt = 0; nt < 3; ++nt) {
                        const int co = cot * CO_T + nt * 16 + n;
                        float* orow = y + (((size_t)b * Cout + co) * H + yy) * W;
#pragma unroll
                        for (int mt = 0; mt < 4; ++mt) {
                            const int xx = x0 + 16 * mt + 4 * g;
                            const f32x4 v = acc[mt][nt];
                            if (xx + 3 < W) *reinterpret_cast<float4*>(orow + xx) = make_float4(v[0], v[1], v[2], v[3]);
                            else {
                                if (xx < W) orow[xx] = v[0];
                                if (xx + 1 < W) orow[xx + 1] = v[1];
                                if (xx + 2 < W) orow[xx + 2] = v[2];
                            }
                        }
                    }
                }
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                    for (int nt = 0; nt < 3; ++nt) acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
            if (more) store_lds(buf ^ 1);
            __syncthreads();
            buf ^= 1;
        }
        t = t_next;
    }
}

// ---------------------------------------------------------------------------------------------------------
// Weight gradient: dW[co][ci][ky][kx] = sum_{b,y,x} dy[b][co][y][x] * x[b][ci][y+ky-1][x+kx-1].
// GEMM per tap: M = co, N = ci, K = pixels (4 per v_mfma_f32_16x16x4_f32). One block = 6 waves owns a 48 x 48 channel
// block and a slice of the spatial tiles (2 rows x 64 columns each); wave w = (co tile w % 3, half w / 3 of the 27
// (ci tile, tap) pairs) keeps 14 (13) accumulators over all its tiles. LDS per tile: x [4 rows][48 ci][pitch 66] (slot
// 64 = right halo, slot 65 = left halo of the NEXT row, so that column -1 of a row is the element before it) and dy
// [2 rows][48 co][pitch 66]; pitch 66 = 2 mod 32 makes both operand reads (16 channels x 2 pixel groups per half-wave)
// conflict-free, and the tap / channel-tile offsets are ds_read immediates. Partials [split][tap][co][ci] are summed in
// a fixed order by wrw_reduce_kernel: deterministic, no atomics. MIOpen's solver for these shapes is an NHWC
// implicit-GEMM kernel wrapped in three layout transposes (x, dy in; dW out).
// ---------------------------------------------------------------------------------------------------------
constexpr int WR = 2;                        // tile rows
constexpr int WP = 66;                       // LDS pitch per (row, channel)
constexpr int WX_FLOATS = (WR + 2) * 48 * WP;   // 12672
constexpr int WD_FLOATS = WR * 48 * WP;         // 6336
constexpr int WGUARD = 2;                    // x tile starts 2 floats in: index -1 of the first row stays in bounds

template <int HALF>
__device__ __forceinline__ void wrw_tile_mfma(const float* __restrict__ xs, const float* __restrict__ ds, int cot, int g,
                                              int n, f32x4 (&acc)[14]) {
    constexpr int P0 = HALF * 14;
    constexpr int NP = HALF == 0 ? 14 : 13;
#pragma unroll 1
    for (int r = 0; r < WR; ++r) {
        const float* da = ds + (r * 48 + cot * 16 + n) * WP + g;          // A: dy[co = n][pixel 4*ks + g]
        const float* xb = xs + (r * 48 + n) * WP + g - 1;                 // B: x[ci = n][pixel 4*ks + g + kx - 1]
#pragma unroll 4
        for (int ks = 0; ks < 16; ++ks) {
            const float a = da[4 * ks];
#pragma unroll
            for (int q = 0; q < NP; ++q) {
                const int pair = P0 + q, cit = pair / 9, tap = pair - 9 * cit;
                const int ky = tap / 3, kx = tap - 3 * ky;
                const float b = xb[4 * ks + (ky * 48 + cit * 16) * WP + kx];
                acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[q], 0, 0, 0);
            }
        }
    }
}

template <int HALF>
__device__ __forceinline__ void wrw_store(const f32x4 (&acc)[14], float* __restrict__ pbase, int Cout, int Cin) {
    constexpr int NP = HALF == 0 ? 14 : 13;
#pragma unroll
    for (int q = 0; q < NP; ++q) {
        const int pair = HALF * 14 + q, cit = pair / 9, tap = pair - 9 * cit;
        float* dst = pbase + (size_t)tap * Cout * Cin + cit * 16;
#pragma unroll
        for (int r = 0; r < 4; ++r) dst[(size_t)r * Cin] = acc[q][r];
    }
}

__global__ __launch_bounds__(384, 3) void conv3x3_wrw_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                             int B, int Cin, int Cout, int H, int W, int tiles_x,
                                                             int tiles_y, int n_split, float* __restrict__ partial) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* xs = smem + WGUARD;
    float* ds = smem + WGUARD + WX_FLOATS + 2;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int g = lane >> 4, n = lane & 15;
    const int cot_w = wave % 3, half = wave / 3;
    int blk = blockIdx.x;
    const int split = blk % n_split; blk /= n_split;
    const int n_cit = Cin / 48;
    const int cib = blk % n_cit;          // 48-wide input-channel block
    const int cob = blk / n_cit;          // 48-wide output-channel block
    const size_t plane = (size_t)H * W;
    const int n_tiles = B * tiles_y * tiles_x;

    f32x4 acc[14];
#pragma unroll
    for (int q = 0; q < 14; ++q) acc[q] = (f32x4){0.f, 0.f, 0.f, 0.f};

    for (int t = split; t < n_tiles; t += n_split) {
        int tt = t;
        const int tx = tt % tiles_x; tt /= tiles_x;
        const int ty = tt % tiles_y;
        const int b = tt / tiles_y;
        const int x0 = tx * 64, y0 = ty * WR;
        const float* xg = x + ((size_t)b * Cin + cib * 48) * plane;
        const float* dg = dy + ((size_t)b * Cout + cob * 48) * plane;
        __syncthreads();                                   // previous tile's operand reads are done
        // x tile: (WR+2) rows x 48 channels x 16 float4 = 3072 float4, 8 per thread
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int v = tid + 384 * u;
            const int q = v & 15, rc = v >> 4;             // rc = r * 48 + ci
            const int r = rc / 48, ci = rc - r * 48;
            const int yy = y0 + r - 1, xx = x0 + 4 * q;
            float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
            if (yy >= 0 && yy < H && xx < W) val = *reinterpret_cast<const float4*>(xg + (size_t)ci * plane + (size_t)yy * W + xx);
            float2* dst = reinterpret_cast<float2*>(xs + rc * WP + 4 * q);
            dst[0] = make_float2(val.x, val.y);
            dst[1] = make_float2(val.z, val.w);
        }
        {   // halos: slot 64 of row rc = x[x0 + 64], slot 65 of row rc - 1 (= index -1 of row rc) = x[x0 - 1]
            const int rc = tid >> 1, side = tid & 1;       // 192 rows x 2 sides = 384 threads
            const int r = rc / 48, ci = rc - r * 48;
            const int yy = y0 + r - 1, xx = side ? x0 + 64 : x0 - 1;
            float v = 0.f;
            if (yy >= 0 && yy < H && xx >= 0 && xx < W) v = xg[(size_t)ci * plane + (size_t)yy * W + xx];
            xs[rc * WP + (side ? 64 : -1)] = v;
        }
        // dy tile: WR rows x 48 channels x 16 float4 = 1536 float4, 4 per thread
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int v = tid + 384 * u;
            const int q = v & 15, rc = v >> 4;
            const int r = rc / 48, co = rc - r * 48;
            const int yy = y0 + r, xx = x0 + 4 * q;
            float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
            if (yy < H && xx < W) val = *reinterpret_cast<const float4*>(dg + (size_t)co * plane + (size_t)yy * W + xx);
            float2* dst = reinterpret_cast<float2*>(ds + rc * WP + 4 * q);
            dst[0] = make_float2(val.x, val.y);
            dst[1] = make_float2(val.z, val.w);
        }
        __syncthreads();
        if (half == 0) wrw_tile_mfma<0>(xs, ds, cot_w, g, n, acc);
        else wrw_tile_mfma<1>(xs, ds, cot_w, g, n, acc);
    }
    // accumulator q of this wave: pair = half*14 + q -> (ci tile, tap); D[m = co][n = ci], lane: ci = n, co = 4*g + r
    float* pbase = partial + ((size_t)split * 9 * Cout + cob * 48 + cot_w * 16 + 4 * g) * Cin + cib * 48 + n;
    if (half == 0) wrw_store<0>(acc, pbase, Cout, Cin);
    else wrw_store<1>(acc, pbase, Cout, Cin);
}

// dW[co][ci][tap] = sum over splits of partial[split][tap][co][ci], fixed order. Block = 64 consecutive outputs x 4
// waves; wave w sums the splits = w (mod 4) with four loads in flight, the four partial sums are added in wave order.
__global__ __launch_bounds__(256) void wrw_reduce_kernel(const float* __restrict__ partial, int n_split, int Cout, int Cin,
                                                         float* __restrict__ dw) {
    __shared__ float red[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int e = blockIdx.x * 64 + lane;                  // e = (tap * Cout + co) * Cin + ci
    const int total = 9 * Cout * Cin;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (e < total) {
        int sp = wave;
        for (; sp + 12 < n_split; sp += 16) {
            s0 += partial[(size_t)sp * total + e];
            s1 += partial[(size_t)(sp + 4) * total + e];
            s2 += partial[(size_t)(sp + 8) * total + e];
            s3 += partial[(size_t)(sp + 12) * total + e];
        }
        for (; sp < n_split; sp += 4) s0 += partial[(size_t)sp * total + e];
    }
    red[wave][lane] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (wave == 0 && e < total) {
        const float v = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
        const int ci = e % Cin, rest = e / Cin;
        const int co = rest % Cout, tap = rest / Cout;
        dw[((size_t)co * Cin + ci) * 9 + tap] = v;
    }
}

int wrw_splits(int B, int Cin, int Cout, int H, int W) {
    const int tiles = B * ((H + WR - 1) / WR) * ((W + 63) / 64);
    const int blocks_per_split = (Cin / 48) * (Cout / 48);
    int n = 512 / blocks_per_split;              // ~2 blocks per CU in total
    if (n < 1) n = 1;
    if (n > tiles) n = tiles;
    return n;
}

}  // namespace

extern "C" size_t cseg_conv3x3_packed_floats(int Cin, int Cout) {
    if (Cin <= 0 || Cout <= 0 || Cin % CK || Cout % CO_T) return 0;
    return (size_t)(Cout / CO_T) * (Cin / CK) * KSTEPS * 3 * 64;
}

extern "C" int cseg_conv3x3_pack_weights(const float* w, int Cout, int Cin, int transpose_flip, float* wp,
                                         cseg_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    // transpose_flip: w is still the forward's [Cout, Cin, 3, 3]; the packed operator maps Cout -> Cin channels
    const int conv_in = transpose_flip ? Cout : Cin, conv_out = transpose_flip ? Cin : Cout;
    CSEG_REQUIRE(w && wp, "conv3x3_pack_weights: null pointer");
    CSEG_REQUIRE(conv_in % CK == 0 && conv_out % CO_T == 0,
                 "conv3x3: needs input channels %% 8 == 0 and output channels %% 48 == 0 (got %d -> %d)", conv_in, conv_out);
    const int total = (int)cseg_conv3x3_packed_floats(conv_in, conv_out);
    hipLaunchKernelGGL(pack_weights_kernel, dim3((total + 255) / 256), dim3(256), 0, stream, w, Cout, Cin, transpose_flip,
                       wp, total);
    CSEG_CHECK_LAUNCH("conv3x3_pack_weights");
    return 1;
}

extern "C" int cseg_conv3x3_fwd(const float* x, const float* wp, int B, int Cin, int Cout, int H, int W, float* y,
                                cseg_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    CSEG_REQUIRE(x && wp && y, "conv3x3: null pointer");
    CSEG_REQUIRE(B > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0 && Cin % CK == 0 && Cout % CO_T == 0,
                 "conv3x3: unsupported shape B=%d Cin=%d Cout=%d %dx%d", B, Cin, Cout, H, W);
    CSEG_REQUIRE((reinterpret_cast<uintptr_t>(x) & 15) == 0 && (reinterpret_cast<uintptr_t>(y) & 15) == 0 && W % 4 == 0,
                 "conv3x3: tensors must be 16-byte aligned and W a multiple of 4");
    const int tiles_x = (W + TC - 1) / TC, tiles_y = (H + TR - 1) / TR;
    const long n_tiles = (long)B * (Cout / CO_T) * tiles_y * tiles_x;
    CSEG_REQUIRE(n_tiles < 2147483647L, "conv3x3: grid too large");
    // persistent blocks: at most two tiles each, never fewer than 512 blocks
    constexpr int tpb = 2;
    long blocks = (n_tiles + tpb - 1) / tpb;
    if (blocks < 512) blocks = n_tiles < 512 ? n_tiles : 512;
    hipLaunchKernelGGL(conv3x3_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, x, wp, Cin, Cout, H, W, tiles_x,
                       tiles_y, (int)n_tiles, y);
    CSEG_CHECK_LAUNCH("conv3x3_kernel");
    return 1;
}

extern "C" size_t cseg_conv3x3_wrw_ws_floats(int B, int Cin, int Cout, int H, int W) {
    if (B <= 0 || Cin <= 0 || Cout <= 0 || H <= 0 || W <= 0 || Cin % 48 || Cout % 48) return 0;
    return (size_t)wrw_splits(B, Cin, Cout, H, W) * 9 * Cin * Cout;
}

extern "C" int cseg_conv3x3_wrw(const float* x, const float* dy, int B, int Cin, int Cout, int H, int W, float* ws,
                                float* dw, cseg_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    CSEG_REQUIRE(x && dy && ws && dw, "conv3x3_wrw: null pointer");
    CSEG_REQUIRE(B > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0 && Cin % 48 == 0 && Cout % 48 == 0,
                 "conv3x3_wrw: unsupported shape B=%d Cin=%d Cout=%d %dx%d", B, Cin, Cout, H, W);
    CSEG_REQUIRE((reinterpret_cast<uintptr_t>(x) & 15) == 0 && (reinterpret_cast<uintptr_t>(dy) & 15) == 0 && W % 4 == 0,
                 "conv3x3_wrw: tensors must be 16-byte aligned and W a multiple of 4");
    const int tiles_x = (W + 63) / 64, tiles_y = (H + WR - 1) / WR;
    const int n_split = wrw_splits(B, Cin, Cout, H, W);
    const int blocks = n_split * (Cin / 48) * (Cout / 48);
    const size_t lds = sizeof(float) * (WGUARD + WX_FLOATS + 2 + WD_FLOATS + 2);
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void*)conv3x3_wrw_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) !=
            hipSuccess) {
            cseg_set_error("conv3x3_wrw: cannot raise dynamic LDS to %zu bytes", lds);
            return 0;
        }
        attr_set = true;
    }
    hipLaunchKernelGGL(conv3x3_wrw_kernel, dim3(blocks), dim3(384), lds, stream, x, dy, B, Cin, Cout, H, W, tiles_x, tiles_y,
                       n_split, ws);
    CSEG_CHECK_LAUNCH("conv3x3_wrw_kernel");
    const int total = 9 * Cin * Cout;
    hipLaunchKernelGGL(wrw_reduce_kernel, dim3((total + 63) / 64), dim3(256), 0, stream, ws, n_split, Cout, Cin, dw);
    CSEG_CHECK_LAUNCH("wrw_reduce_kernel");
    return 1;
}
