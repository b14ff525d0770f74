// The FIRST convolution of the stem: nn.Conv2d(3, 64, 3, 2, 1, bias=False) on the image (reference
// lib/models/backbones/hrnet/hrnet_backbone.py:516-517; the ResNet stems of lib/models/backbones/resnet/resnet_models.py start the same
// way at other widths). Three input channels: K = 27, nothing for the matrix cores -- both directions are STREAMS (forward: 50 MB in,
// 268 MB out at 8 x 512 x 1024; weight gradient: 268 + 50 MB in, 6.9 KB out) with 1.8 G fp32 multiply-adds beside them, so they are plain
// fp32 FMA kernels in NCHW: no layout change, no split arithmetic, a fixed summation order (bit-reproducible run to run).
// Round 6: until now these two were the last MIOpen convolutions of the HRNet step (an NHWC implicit GEMM with layout transposes of the
// 268 MB output / output gradient); an aten convolution call also costs the host ~0.4 ms where the backward pass is host-bound
// (DESIGN.md section 13.11). There is no backward-data operator: the image needs no gradient.
#include "cseg_common.h"

namespace {

constexpr int ST_CO = 64;                           // output channels (the only count built)
constexpr int ST_TR = 4, ST_TC = 64;                // output tile of a block: 4 rows x 64 columns
constexpr int ST_PR = 2 * ST_TR + 1;                // input rows of the tile's patch: 9
constexpr int ST_PC = 2 * ST_TC + 1;                // input columns: 129
constexpr int ST_PP = 132;                          // pitch of a patch row (floats)
constexpr int ST_PATCH = 3 * ST_PR * ST_PP;         // floats of one patch
constexpr int ST_DP = ST_TR * ST_TC + 1;            // pitch of a dy channel row in LDS: 257 (odd: the 64 lanes = 64 channels hit 64 banks)

// patch[ci][r][c] = x[b][ci][2 y0 - 1 + r][2 x0 - 1 + c], zero outside the image (the convolution's padding)
__device__ __forceinline__ void stem_load_patch(const float* __restrict__ xb, int H, int W, int y0, int x0, float* __restrict__ patch, int tid,
                                                int nthreads) {
    for (int i = tid; i < 3 * ST_PR * ST_PC; i += nthreads) {
        const int ci = i / (ST_PR * ST_PC), rc = i - ci * (ST_PR * ST_PC);
        const int r = rc / ST_PC, c = rc - r * ST_PC;
        const int iy = 2 * y0 - 1 + r, ix = 2 * x0 - 1 + c;
        const bool in = iy >= 0 && iy < H && ix >= 0 && ix < W;
        patch[(ci * ST_PR + r) * ST_PP + c] = in ? xb[((size_t)ci * H + iy) * W + ix] : 0.f;
    }
}

// y[b][co][oy][ox] = sum_k x-patch[k] * w[co][k], k = (ci, ky, kx) ascending. One thread = one output pixel x all 64 channels:
// 27 patch values from LDS (each used 64 times from a register), the weights as broadcast 16-byte LDS reads.
__global__ __launch_bounds__(256) void stem_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w, int H, int W, int Ho, int Wo,
                                                       int tiles_x, int tiles_y, float* __restrict__ y) {
    __shared__ __attribute__((aligned(16))) float patch[ST_PATCH];
    __shared__ __attribute__((aligned(16))) float wl[27 * ST_CO];           // [k][co]
    const int tid = threadIdx.x, tx = tid & 63, ty = tid >> 6;
    int t = blockIdx.x;
    const int txb = t % tiles_x; t /= tiles_x;
    const int tyb = t % tiles_y;
    const int b = t / tiles_y;
    const int x0 = txb * ST_TC, y0 = tyb * ST_TR;
    for (int i = tid; i < 27 * ST_CO; i += 256) {
        const int co = i / 27, k = i - co * 27;                             // w is [co][ci][ky][kx] = [co][k]
        wl[k * ST_CO + co] = w[i];
    }
    stem_load_patch(x + (size_t)b * 3 * H * W, H, W, y0, x0, patch, tid, 256);
    __syncthreads();
    float acc[ST_CO];
#pragma unroll
    for (int co = 0; co < ST_CO; ++co) acc[co] = 0.f;
#pragma unroll
    for (int k = 0; k < 27; ++k) {
        const int ci = k / 9, ky = (k % 9) / 3, kx = k % 3;
        const float xv = patch[(ci * ST_PR + 2 * ty + ky) * ST_PP + 2 * tx + kx];
#pragma unroll
        for (int c4 = 0; c4 < ST_CO / 4; ++c4) {
            const float4 wv = *reinterpret_cast<const float4*>(wl + k * ST_CO + 4 * c4);
            acc[4 * c4 + 0] = __builtin_fmaf(xv, wv.x, acc[4 * c4 + 0]);
            acc[4 * c4 + 1] = __builtin_fmaf(xv, wv.y, acc[4 * c4 + 1]);
            acc[4 * c4 + 2] = __builtin_fmaf(xv, wv.z, acc[4 * c4 + 2]);
            acc[4 * c4 + 3] = __builtin_fmaf(xv, wv.w, acc[4 * c4 + 3]);
        }
    }
    const int oy = y0 + ty, ox = x0 + tx;
    if (oy < Ho && ox < Wo) {
        float* yp = y + (((size_t)b * ST_CO) * Ho + oy) * Wo + ox;          // lanes = 64 consecutive columns of one row: 256-byte stores
        const size_t plane = (size_t)Ho * Wo;
#pragma unroll
        for (int co = 0; co < ST_CO; ++co) yp[(size_t)co * plane] = acc[co];
    }
}

// dw[co][k] = sum over (image, output pixel) of dy[b][co][oy][ox] * x-patch[k]. Persistent blocks walk the 4 x 64 tiles; lane = output
// channel, wave = (row of the tile, half of its columns); the 27 patch values of a pixel are the same for all 64 lanes (broadcast LDS
// reads), dy comes through LDS so that the global reads run along rows and the per-lane reads along channels. Every lane keeps 27 sums;
// the eight waves of a block are added in a fixed order, the blocks by the reduction kernel below: deterministic.
__global__ __launch_bounds__(512) void stem_wrw_kernel(const float* __restrict__ x, const float* __restrict__ dy, int H, int W, int Ho, int Wo,
                                                       int tiles_x, int tiles_y, int n_tiles, float* __restrict__ partial) {
    extern __shared__ __attribute__((aligned(16))) float smem_stem[];
    float* patch = smem_stem;                          // ST_PATCH
    float* dyl = smem_stem + ST_PATCH;                 // [co 64][ST_DP]; after the tiles: the eight waves' sums [8][27][64]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int row = wave >> 1, c0 = (wave & 1) * 32;
    float acc[27];
#pragma unroll
    for (int k = 0; k < 27; ++k) acc[k] = 0.f;
    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        int t = tile;
        const int txb = t % tiles_x; t /= tiles_x;
        const int tyb = t % tiles_y;
        const int b = t / tiles_y;
        const int x0 = txb * ST_TC, y0 = tyb * ST_TR;
        __syncthreads();                               // the previous tile has been read
        stem_load_patch(x + (size_t)b * 3 * H * W, H, W, y0, x0, patch, tid, 512);
        const float* dyb = dy + (size_t)b * ST_CO * Ho * Wo;
        if ((Wo & 3) == 0 && x0 + ST_TC <= Wo && y0 + ST_TR <= Ho) {
            // a tile inside the image whose rows are 16-byte aligned: 16-byte loads (8 per thread instead of 32)
            for (int i = tid; i < ST_CO * ST_TR * ST_TC / 4; i += 512) {
                const int co = i >> 6, p = (i & 63) * 4;
                const float4 v = *reinterpret_cast<const float4*>(dyb + ((size_t)co * Ho + y0 + (p >> 6)) * Wo + x0 + (p & 63));
                float* d = dyl + co * ST_DP + p;           // (pitch 257: not 16-byte aligned)
                d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
            }
        } else {
            for (int i = tid; i < ST_CO * ST_TR * ST_TC; i += 512) {
                const int co = i >> 8, p = i & 255;        // p = r * 64 + c
                const int oy = y0 + (p >> 6), ox = x0 + (p & 63);
                dyl[co * ST_DP + p] = (oy < Ho && ox < Wo) ? dyb[((size_t)co * Ho + oy) * Wo + ox] : 0.f;
            }
        }
        __syncthreads();
        // two pixels per turn: their patch columns 2c .. 2c + 4 of a (channel, filter row) are one aligned 16-byte read + one more
        // float (all lanes read the same address: broadcast) -- 20 LDS reads per pixel pair instead of 56
#pragma unroll 2
        for (int c = c0; c < c0 + 32; c += 2) {
            const float d0 = dyl[lane * ST_DP + row * ST_TC + c], d1 = dyl[lane * ST_DP + row * ST_TC + c + 1];
#pragma unroll
            for (int cr = 0; cr < 9; ++cr) {               // cr = ci * 3 + ky
                const int ci = cr / 3, ky = cr % 3;
                const float* pr = patch + (ci * ST_PR + 2 * row + ky) * ST_PP + 2 * c;
                const float4 v = *reinterpret_cast<const float4*>(pr);
                const float v4 = pr[4];
                acc[3 * cr + 0] = __builtin_fmaf(d0, v.x, acc[3 * cr + 0]);
                acc[3 * cr + 1] = __builtin_fmaf(d0, v.y, acc[3 * cr + 1]);
                acc[3 * cr + 2] = __builtin_fmaf(d0, v.z, acc[3 * cr + 2]);
                acc[3 * cr + 0] = __builtin_fmaf(d1, v.z, acc[3 * cr + 0]);
                acc[3 * cr + 1] = __builtin_fmaf(d1, v.w, acc[3 * cr + 1]);
                acc[3 * cr + 2] = __builtin_fmaf(d1, v4, acc[3 * cr + 2]);
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 27; ++k) dyl[(wave * 27 + k) * ST_CO + lane] = acc[k];
    __syncthreads();
    for (int i = tid; i < 27 * ST_CO; i += 512) {       // i = k * 64 + co
        float s = 0.f;
#pragma unroll
        for (int q = 0; q < 8; ++q) s += dyl[q * 27 * ST_CO + i];
        partial[(size_t)blockIdx.x * 27 * ST_CO + i] = s;
    }
}

// dw[co][k] = sum over blocks of partial[block][k][co], fixed order (four waves x four chains per 64 elements, see sb_wrw1_reduce_kernel)
__global__ __launch_bounds__(256) void stem_wrw_reduce_kernel(const float* __restrict__ partial, int n_blocks, float* __restrict__ dw) {
    __shared__ float red[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int total = 27 * ST_CO;
    const int e = blockIdx.x * 64 + lane;               // e = k * 64 + co
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (e < total) {
        int sp = wave;
        for (; sp + 12 < n_blocks; sp += 16) {
            s0 += partial[(size_t)sp * total + e];
            s1 += partial[(size_t)(sp + 4) * total + e];
            s2 += partial[(size_t)(sp + 8) * total + e];
            s3 += partial[(size_t)(sp + 12) * total + e];
        }
        for (; sp < n_blocks; sp += 4) s0 += partial[(size_t)sp * total + e];
    }
    red[wave][lane] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (wave == 0 && e < total) {
        const int k = e / ST_CO, co = e - k * ST_CO;
        dw[co * 27 + k] = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
    }
}

int stem_wrw_blocks(long n_tiles) { return (int)(n_tiles < 512 ? n_tiles : 512); }      // two 512-thread blocks per CU (2 x 78 KB of LDS): one loads while the other multiplies

bool stem_shape_ok(int B, int Cout, int H, int W) {
    return B > 0 && Cout == ST_CO && H >= 2 && W >= 2 && H % 2 == 0 && W % 2 == 0 && (long)B * 3 * H * W < 2147483647L &&
           (long)B * ST_CO * (H / 2) * (W / 2) < 2147483647L;
}

}  // namespace

// y [B, 64, H/2, W/2] = conv2d(x [B, 3, H, W], w [64, 3, 3, 3], stride 2, padding 1): fp32, plain NCHW tensors, no packed weights
extern "C" int cseg_conv3x3_s2_rgb_fwd(const float* x, const float* w, int B, int Cout, int H, int W, float* y, cseg_stream_t stream_) {
    CSEG_REQUIRE(x && w && y, "conv3x3_s2_rgb_fwd: null pointer");
    CSEG_REQUIRE(stem_shape_ok(B, Cout, H, W), "conv3x3_s2_rgb_fwd: unsupported shape B=%d Cout=%d %dx%d (needs Cout = 64, even H and W)", B, Cout, H, W);
    const int Ho = H / 2, Wo = W / 2;
    const int tiles_x = (Wo + ST_TC - 1) / ST_TC, tiles_y = (Ho + ST_TR - 1) / ST_TR;
    const long n_tiles = (long)B * tiles_x * tiles_y;
    CSEG_REQUIRE(n_tiles < 2147483647L, "conv3x3_s2_rgb_fwd: grid too large");
    hipLaunchKernelGGL(stem_fwd_kernel, dim3((unsigned)n_tiles), dim3(256), 0, (hipStream_t)stream_, x, w, H, W, Ho, Wo, tiles_x, tiles_y, y);
    CSEG_CHECK_LAUNCH("stem_fwd_kernel");
    return 1;
}

// floats of the workspace of cseg_conv3x3_s2_rgb_wrw (0 = unsupported shape)
extern "C" size_t cseg_conv3x3_s2_rgb_wrw_ws_floats(int B, int Cout, int H, int W) {
    if (!stem_shape_ok(B, Cout, H, W)) return 0;
    const long n_tiles = (long)B * ((W / 2 + ST_TC - 1) / ST_TC) * ((H / 2 + ST_TR - 1) / ST_TR);
    return (size_t)stem_wrw_blocks(n_tiles) * 27 * ST_CO;
}

// dw [64, 3, 3, 3] of that convolution for the output gradient dy [B, 64, H/2, W/2]. Deterministic (fixed-order sums).
extern "C" int cseg_conv3x3_s2_rgb_wrw(const float* x, const float* dy, int B, int Cout, int H, int W, float* ws, float* dw,
                                       cseg_stream_t stream_) {
    CSEG_REQUIRE(x && dy && ws && dw, "conv3x3_s2_rgb_wrw: null pointer");
    CSEG_REQUIRE(stem_shape_ok(B, Cout, H, W), "conv3x3_s2_rgb_wrw: unsupported shape B=%d Cout=%d %dx%d (needs Cout = 64, even H and W)", B, Cout, H, W);
    const int Ho = H / 2, Wo = W / 2;
    const int tiles_x = (Wo + ST_TC - 1) / ST_TC, tiles_y = (Ho + ST_TR - 1) / ST_TR;
    const long n_tiles = (long)B * tiles_x * tiles_y;
    CSEG_REQUIRE(n_tiles < 2147483647L, "conv3x3_s2_rgb_wrw: too many tiles");
    const int n_blocks = stem_wrw_blocks(n_tiles);
    const size_t lds = sizeof(float) * (ST_PATCH + ST_CO * ST_DP);
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void*)stem_wrw_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
            cseg_set_error("conv3x3_s2_rgb_wrw: cannot raise dynamic LDS to %zu bytes", lds);
            return 0;
        }
        attr_set = true;
    }
    hipStream_t stream = (hipStream_t)stream_;
    hipLaunchKernelGGL(stem_wrw_kernel, dim3((unsigned)n_blocks), dim3(512), lds, stream, x, dy, H, W, Ho, Wo, tiles_x, tiles_y, (int)n_tiles, ws);
    CSEG_CHECK_LAUNCH("stem_wrw_kernel");
    hipLaunchKernelGGL(stem_wrw_reduce_kernel, dim3((27 * ST_CO + 63) / 64), dim3(256), 0, stream, ws, n_blocks, dw);
    CSEG_CHECK_LAUNCH("stem_wrw_reduce_kernel");
    return 1;
}
