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

template <int VARIANT>
__global__ __launch_bounds__(256, 3) void conv3x3_kernel(const float* __restrict__ x, const float* __restrict__ wp,
                                                         int Cin, int Cout, int H, int W, int tiles_x, int tiles_y,
                                                         float* __restrict__ y) {
    __shared__ __attribute__((aligned(16))) float xs[2][CHUNK_FLOATS];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int g = lane >> 4, n = lane & 15;
    int blk = blockIdx.x;
    const int tx = blk % tiles_x; blk /= tiles_x;
    const int ty = blk % tiles_y; blk /= tiles_y;
    const int n_cot = Cout / CO_T;
    const int cot = blk % n_cot;
    const int b = blk / n_cot;
    const int x0 = tx * TC, y0 = ty * TR;
    const int n_chunks = Cin / CK;
    const size_t plane = (size_t)H * W;
    const float* xb = x + (size_t)b * Cin * plane;

    // ---- staging assignment: 48 (channel, row) pairs per chunk; 16 float4 per pair + 2 halo scalars
    // vector part: v = tid + 256*u, u < 3  ->  pair = v >> 4, quad = v & 15
    // scalar part: tid < 96 -> pair = tid >> 1, side = tid & 1
    float4 pv[3];
    float ps = 0.f;
    auto issue_loads = [&](int chunk) {
        const float* xc = xb + (size_t)chunk * CK * plane;
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            const int v = tid + 256 * u;
            const int pair = v >> 4, q = v & 15;
            const int ci = pair / XROWS, r = pair - ci * XROWS;
            const int yy = y0 + r - 1, xx = x0 + 4 * q;
            float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
            if (yy >= 0 && yy < H) {
                const float* p = xc + (size_t)ci * plane + (size_t)yy * W + xx;
                if (xx + 3 < W) t = *reinterpret_cast<const float4*>(p);
                else {
                    if (xx < W) t.x = p[0];
                    if (xx + 1 < W) t.y = p[1];
                    if (xx + 2 < W) t.z = p[2];
                }
            }
            pv[u] = t;
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

    issue_loads(0);
    store_lds(0);
    __syncthreads();

    // this lane's A-operand base: channel g of a quad, row = wave (+ky), column = 3 + n (+kx + 16*mt)
    const int a_base = (g * XROWS + wave) * LDW + 3 + n;
    const float* wbase = wp + ((size_t)cot * n_chunks) * (KSTEPS * 3 * 64) + lane;

    for (int c = 0; c < n_chunks; ++c) {
        const int buf = c & 1;
        if (c + 1 < n_chunks) issue_loads(c + 1);
        const float* wc = wbase + (size_t)c * (KSTEPS * 3 * 64);
        const float* xa = &xs[buf][a_base];
        float wr[VARIANT == 1 ? KSTEPS * 3 : 1];
        if (VARIANT == 1) {               // all B operands of the chunk in flight before the first MFMA
#pragma unroll
            for (int i = 0; i < KSTEPS * 3; ++i) wr[i] = wc[i * 64];
        }
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int ky = tap / 3, kx = tap - 3 * ky;
#pragma unroll
            for (int cq = 0; cq < 2; ++cq) {
                const int ks = tap * 2 + cq;
                const float b0 = VARIANT == 1 ? wr[ks * 3 + 0] : wc[(ks * 3 + 0) * 64];
                const float b1 = VARIANT == 1 ? wr[ks * 3 + 1] : wc[(ks * 3 + 1) * 64];
                const float b2 = VARIANT == 1 ? wr[ks * 3 + 2] : wc[(ks * 3 + 2) * 64];
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
        if (c + 1 < n_chunks) store_lds(buf ^ 1);
        __syncthreads();
    }

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
    const long blocks = (long)B * (Cout / CO_T) * tiles_y * tiles_x;
    CSEG_REQUIRE(blocks < 2147483647L, "conv3x3: grid too large");
    static const int variant = getenv("CSEG_CONV3X3_VARIANT") ? atoi(getenv("CSEG_CONV3X3_VARIANT")) : 0;
    if (variant == 1)
        hipLaunchKernelGGL(conv3x3_kernel<1>, dim3((unsigned)blocks), dim3(256), 0, stream, x, wp, Cin, Cout, H, W, tiles_x,
                           tiles_y, y);
    else
        hipLaunchKernelGGL(conv3x3_kernel<0>, dim3((unsigned)blocks), dim3(256), 0, stream, x, wp, Cin, Cout, H, W, tiles_x,
                           tiles_y, y);
    CSEG_CHECK_LAUNCH("conv3x3_kernel");
    return 1;
}
