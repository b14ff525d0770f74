// The classifier convolution: a 1x1 convolution onto a HANDFUL of output channels (num_classes: 19 on Cityscapes) from a wide
// activation (720 channels at 1/4 resolution: 755 MB at the benched batch). Reference: the last layer of `cls_head`,
// lib/models/nets/hrnet.py:73-80 (nn.Conv2d(720, num_classes, 1) behind BNReLU + Dropout2d(0.10)) and the OCR / DeepLab classifiers.
// With K <= 32 outputs all three operators are STREAMS over the wide tensor -- 2 K flops per 4 bytes -- so they are written as plain
// fp32 FMA kernels bounded by HBM, not as GEMMs: in the round-6 step the same work ran as rocBLAS `Cijk_*_MT128x128x32` (0.40 ms
// forward at 1.9 TB/s, a 128-wide tile for 19 columns), `Cijk_*_MT64x128x16` (0.26 ms backward-data) and MIOpen's NHWC implicit-GEMM
// weight gradient with two layout transposes of the 755 MB activation (0.20 + 0.34 ms) -- the last library kernels of the HRNet step.
//
// Weights are PER IMAGE: wt [B][C][KP] (KP = K padded to 20 or 32, pad columns zero), because that is how the channel dropout in front
// of the classifier is folded in: Dropout2d multiplies channel c of image b by m[b][c] in {0, 1/(1-p)}, and
//   sum_c w[k][c] (m[b][c] x[b][c][p]) = sum_c (w[k][c] m[b][c]) x[b][c][p],
// so the host multiplies the 19 x 720 matrix by the mask (a few KB) instead of the 755 MB tensor (one read + one write forward, the
// same again backward). Without dropout the B copies are equal.
//   forward        y[b][k][p]  = bias[k] + sum_c wt[b][c][k] x[b][c][p]
//   backward-data  dx[b][c][p] = sum_k wt[b][c][k] dy[b][k][p]
//   weight grad.   dwt[b][c][k] = sum_p x[b][c][p] dy[b][k][p]      (the host folds the mask and the batch sum: tiny tensors)
// All sums in a fixed order (no atomics): deterministic.
#include "cseg_common.h"
#include "cseg_hip.h"

namespace {

constexpr int PX = 128;                // pixels per block of the forward kernel (one per lane, two channel halves)

// ---- forward: thread = (pixel, channel half); the two halves of a pixel are summed through LDS -----------------------------------------
template <int KP>
__global__ __launch_bounds__(256) void cls1x1_fwd_kernel(const float* __restrict__ x, const float* __restrict__ wt,
                                                         const float* __restrict__ bias, int C, int K, long P, int tiles,
                                                         float* __restrict__ y) {
    __shared__ float red[PX][KP + 1];
    const int b = blockIdx.x / tiles, tile = blockIdx.x - b * tiles;
    const int half = __builtin_amdgcn_readfirstlane((int)threadIdx.x / PX), lp = threadIdx.x - half * PX;      // (waves 0, 1 / 2, 3)
    const long p = (long)tile * PX + lp;
    const bool live = p < P;
    const long pc = live ? p : P - 1;
    const int c_mid = (C + 1) / 2;
    const int c0 = half ? c_mid : 0, c1 = half ? C : c_mid;
    const float* xp = x + ((size_t)b * C + c0) * P + pc;
    const float* wp = wt + ((size_t)b * C + c0) * KP;          // wave-uniform: scalar loads
    float acc[KP];
#pragma unroll
    for (int k = 0; k < KP; ++k) acc[k] = 0.f;
    int c = c0;
    for (; c + 8 <= c1; c += 8) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = xp[(size_t)j * P];
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int k = 0; k < KP; ++k) acc[k] = __builtin_fmaf(v[j], wp[j * KP + k], acc[k]);
        xp += (size_t)8 * P;
        wp += 8 * KP;
    }
    for (; c < c1; ++c) {
        const float v = xp[0];
#pragma unroll
        for (int k = 0; k < KP; ++k) acc[k] = __builtin_fmaf(v, wp[k], acc[k]);
        xp += P;
        wp += KP;
    }
    if (half) {
#pragma unroll
        for (int k = 0; k < KP; ++k) red[lp][k] = acc[k];
    }
    __syncthreads();
    if (!half && live) {
        float* yp = y + (size_t)b * K * P + p;
#pragma unroll
        for (int k = 0; k < KP; ++k)
            if (k < K) yp[(size_t)k * P] = (acc[k] + red[lp][k]) + (bias ? bias[k] : 0.f);      // first half + second half, then the bias
    }
}

// ---- backward-data: thread = four consecutive pixels (16-byte loads of dy and stores of dx when P % 4 == 0), blockIdx.y = a part of the channels
template <int KP, bool VEC>
__global__ __launch_bounds__(256) void cls1x1_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ wt, int C, int K,
                                                         long P, int tiles, float* __restrict__ dx) {
    constexpr int V = VEC ? 4 : 1;
    const int b = blockIdx.x / tiles, tile = blockIdx.x - b * tiles;
    const long p = ((long)tile * 256 + threadIdx.x) * V;
    const bool live = p < P;
    const long pc = live ? p : 0;
    float d[KP][V];
#pragma unroll
    for (int k = 0; k < KP; ++k) {
        if (k < K) {
            const float* src = dy + ((size_t)b * K + k) * P + pc;
            if (VEC) {
                const float4 t4 = *reinterpret_cast<const float4*>(src);
                d[k][0] = t4.x; d[k][1 % V] = t4.y; d[k][2 % V] = t4.z; d[k][3 % V] = t4.w;
            } else {
                d[k][0] = *src;
            }
        } else {
#pragma unroll
            for (int t = 0; t < V; ++t) d[k][t] = 0.f;
        }
    }
    const int per = (C + (int)gridDim.y - 1) / (int)gridDim.y;
    const int c0 = blockIdx.y * per, c1 = min(C, c0 + per);
    const float* wp = wt + ((size_t)b * C + c0) * KP;          // wave-uniform
    float* op = dx + ((size_t)b * C + c0) * P + pc;
    for (int c = c0; c < c1; ++c) {
        float s[V];
#pragma unroll
        for (int t = 0; t < V; ++t) s[t] = 0.f;
#pragma unroll
        for (int k = 0; k < KP; ++k)
#pragma unroll
            for (int t = 0; t < V; ++t) s[t] = __builtin_fmaf(wp[k], d[k][t], s[t]);
        if (live) {
            if (VEC) *reinterpret_cast<float4*>(op) = make_float4(s[0], s[1 % V], s[2 % V], s[3 % V]);
            else *op = s[0];
        }
        wp += KP;
        op += P;
    }
}

// ---- weight gradient: block = (image, tile of 64 channels, pixel split); lane = channel, the four waves share the 64 pixels of a stage ----
constexpr int WC = 64, WP = 64;        // channels / pixels per stage
constexpr int A_PITCH = WP + 1;        // lane = channel reads column p of its row: banks (c + p) mod 64, conflict-free

template <int KP, bool VEC>
__global__ __launch_bounds__(256) void cls1x1_wrw_kernel(const float* __restrict__ x, const float* __restrict__ dy, int C, int K, long P,
                                                         int c_tiles, int n_split, float* __restrict__ partial) {
    __shared__ __attribute__((aligned(16))) float As[WC * A_PITCH];            // [channel][pixel]
    __shared__ __attribute__((aligned(16))) float Ds[WP * KP];                 // [pixel][class]: one b128 read = four classes, broadcast
    __shared__ float red[3][WC][KP + 1];
    int blk = blockIdx.x;
    const int split = blk % n_split; blk /= n_split;
    const int ct = blk % c_tiles;
    const int b = blk / c_tiles;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const long stages = (P + WP - 1) / WP;
    const long s0 = stages * split / n_split, s1 = stages * (split + 1) / n_split;
    const int cb = ct * WC;
    float acc[KP];
#pragma unroll
    for (int k = 0; k < KP; ++k) acc[k] = 0.f;
    // loader roles: A: thread -> (channel tid / 4, sixteen pixels (tid % 4) * 16 ..); D: thread -> (class tid / 16, four pixels (tid % 16) * 4 ..)
    const int a_ch = tid >> 2, a_px = (tid & 3) * 16;
    const bool a_ok = cb + a_ch < C;
    const float* a_src = x + ((size_t)b * C + min(cb + a_ch, C - 1)) * P;
    const int d_k = tid >> 4, d_px = (tid & 15) * 4;
    for (long st = s0; st < s1; ++st) {
        const long p0 = st * WP;
        float av[16], dv0[4], dv1[4];
        if (VEC && p0 + WP <= P) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 t4 = *reinterpret_cast<const float4*>(a_src + p0 + a_px + 4 * q);
                av[4 * q] = t4.x; av[4 * q + 1] = t4.y; av[4 * q + 2] = t4.z; av[4 * q + 3] = t4.w;
            }
        } else {
#pragma unroll
            for (int j = 0; j < 16; ++j) av[j] = p0 + a_px + j < P ? a_src[p0 + a_px + j] : 0.f;
        }
        // dy rows: classes d_k and d_k + 16 (KP <= 32)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const long p = p0 + d_px + j;
            dv0[j] = (d_k < K && p < P) ? dy[((size_t)b * K + d_k) * P + p] : 0.f;
            dv1[j] = (d_k + 16 < K && p < P) ? dy[((size_t)b * K + d_k + 16) * P + p] : 0.f;
        }
        __syncthreads();                                       // the previous stage has been consumed
#pragma unroll
        for (int j = 0; j < 16; ++j) As[a_ch * A_PITCH + a_px + j] = a_ok ? av[j] : 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            Ds[(d_px + j) * KP + d_k] = dv0[j];
            if (d_k + 16 < KP) Ds[(d_px + j) * KP + d_k + 16] = dv1[j];
        }
        __syncthreads();
#pragma unroll 4
        for (int j = 0; j < WP / 4; ++j) {
            const int p = wave * (WP / 4) + j;
            const float a = As[lane * A_PITCH + p];
#pragma unroll
            for (int k4 = 0; k4 < KP / 4; ++k4) {
                const float4 d4 = *reinterpret_cast<const float4*>(Ds + p * KP + 4 * k4);
                acc[4 * k4] = __builtin_fmaf(a, d4.x, acc[4 * k4]);
                acc[4 * k4 + 1] = __builtin_fmaf(a, d4.y, acc[4 * k4 + 1]);
                acc[4 * k4 + 2] = __builtin_fmaf(a, d4.z, acc[4 * k4 + 2]);
                acc[4 * k4 + 3] = __builtin_fmaf(a, d4.w, acc[4 * k4 + 3]);
            }
        }
    }
    // the four waves hold the sums over their quarter of every stage's pixels: wave 0 adds them in the order 0, 1, 2, 3
    if (wave > 0) {
#pragma unroll
        for (int k = 0; k < KP; ++k) red[wave - 1][lane][k] = acc[k];
    }
    __syncthreads();
    if (wave == 0 && cb + lane < C) {
        const int B = gridDim.x / (n_split * c_tiles);
        float* out = partial + (((size_t)split * B + b) * C + (cb + lane)) * KP;          // [split][image][channel][class]
#pragma unroll
        for (int k = 0; k < KP; ++k) out[k] = ((acc[k] + red[0][lane][k]) + red[1][lane][k]) + red[2][lane][k];
    }
}

// dwt[e] = sum over the splits, in order
__global__ __launch_bounds__(256) void cls1x1_wrw_reduce_kernel(const float* __restrict__ partial, int n_split, long total,
                                                                float* __restrict__ dwt) {
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e >= total) return;
    float s = 0.f;
    for (int sp = 0; sp < n_split; ++sp) s += partial[(size_t)sp * total + e];
    dwt[e] = s;
}

int wrw_splits(int B, int C, long P) {
    const long stages = (P + WP - 1) / WP;
    const long groups = (long)B * ((C + WC - 1) / WC);
    long n = (1536 + groups - 1) / groups;                     // ~6 blocks per CU in flight (25 KB of LDS each)
    if (n > stages) n = stages;
    if (n > 64) n = 64;
    return (int)(n < 1 ? 1 : n);
}

bool shape_ok(int B, int C, int K, int KP, long P) {
    return B > 0 && C > 0 && K > 0 && K <= KP && (KP == 20 || KP == 32) && P > 0 && (long)B * C * P < (1L << 40) &&
           (long)B * ((P + PX - 1) / PX) < 2147483647L / 16;
}

}  // namespace

extern "C" int cseg_cls1x1_fwd(const float* x, const float* wt, const float* bias, int B, int C, int K, int KP, long P, float* y,
                               cseg_stream_t stream_) {
    CSEG_REQUIRE(x && wt && y, "cls1x1_fwd: null pointer");
    CSEG_REQUIRE(shape_ok(B, C, K, KP, P), "cls1x1_fwd: unsupported shape B=%d C=%d K=%d KP=%d P=%ld (K <= KP, KP 20 or 32)", B, C, K, KP, P);
    hipStream_t stream = (hipStream_t)stream_;
    const int tiles = (int)((P + PX - 1) / PX);
    if (KP == 20)
        hipLaunchKernelGGL(cls1x1_fwd_kernel<20>, dim3((unsigned)(B * tiles)), dim3(256), 0, stream, x, wt, bias, C, K, P, tiles, y);
    else
        hipLaunchKernelGGL(cls1x1_fwd_kernel<32>, dim3((unsigned)(B * tiles)), dim3(256), 0, stream, x, wt, bias, C, K, P, tiles, y);
    CSEG_CHECK_LAUNCH("cls1x1_fwd_kernel");
    return 1;
}

extern "C" int cseg_cls1x1_bwd(const float* dy, const float* wt, int B, int C, int K, int KP, long P, float* dx, cseg_stream_t stream_) {
    CSEG_REQUIRE(dy && wt && dx, "cls1x1_bwd: null pointer");
    CSEG_REQUIRE(shape_ok(B, C, K, KP, P), "cls1x1_bwd: unsupported shape B=%d C=%d K=%d KP=%d P=%ld (K <= KP, KP 20 or 32)", B, C, K, KP, P);
    hipStream_t stream = (hipStream_t)stream_;
    const bool vec = P % 4 == 0 && ((reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(dx)) & 15) == 0;
    const long per_block = vec ? 1024 : 256;
    const int tiles = (int)((P + per_block - 1) / per_block);
    int parts = 1;                                             // channel parts: enough blocks for the chip (each re-reads the K-channel dy)
    while (parts < 16 && (long)B * tiles * parts < 2048 && C / (parts * 2) >= 32) parts *= 2;
#define CLS_BWD(KPV, V)                                                                                                                \
    hipLaunchKernelGGL((cls1x1_bwd_kernel<KPV, V>), dim3((unsigned)(B * tiles), parts), dim3(256), 0, stream, dy, wt, C, K, P, tiles, dx)
    if (KP == 20) { if (vec) CLS_BWD(20, true); else CLS_BWD(20, false); }
    else          { if (vec) CLS_BWD(32, true); else CLS_BWD(32, false); }
#undef CLS_BWD
    CSEG_CHECK_LAUNCH("cls1x1_bwd_kernel");
    return 1;
}

extern "C" size_t cseg_cls1x1_wrw_ws_floats(int B, int C, int KP, long P) {
    if (B <= 0 || C <= 0 || P <= 0 || (KP != 20 && KP != 32)) return 0;
    return (size_t)wrw_splits(B, C, P) * B * C * KP;
}

extern "C" int cseg_cls1x1_wrw(const float* x, const float* dy, int B, int C, int K, int KP, long P, float* ws, float* dwt,
                               cseg_stream_t stream_) {
    CSEG_REQUIRE(x && dy && ws && dwt, "cls1x1_wrw: null pointer");
    CSEG_REQUIRE(shape_ok(B, C, K, KP, P), "cls1x1_wrw: unsupported shape B=%d C=%d K=%d KP=%d P=%ld (K <= KP, KP 20 or 32)", B, C, K, KP, P);
    hipStream_t stream = (hipStream_t)stream_;
    const int n_split = wrw_splits(B, C, P), c_tiles = (C + WC - 1) / WC;
    const long blocks = (long)B * c_tiles * n_split;
    CSEG_REQUIRE(blocks < 2147483647L, "cls1x1_wrw: grid too large");
    const bool vec = P % 4 == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0;
#define CLS_WRW(KPV, V)                                                                                                            \
    hipLaunchKernelGGL((cls1x1_wrw_kernel<KPV, V>), dim3((unsigned)blocks), dim3(256), 0, stream, x, dy, C, K, P, c_tiles, n_split, ws)
    if (KP == 20) { if (vec) CLS_WRW(20, true); else CLS_WRW(20, false); }
    else          { if (vec) CLS_WRW(32, true); else CLS_WRW(32, false); }
#undef CLS_WRW
    CSEG_CHECK_LAUNCH("cls1x1_wrw_kernel");
    const long total = (long)B * C * KP;
    hipLaunchKernelGGL(cls1x1_wrw_reduce_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, ws, n_split, total, dwt);
    CSEG_CHECK_LAUNCH("cls1x1_wrw_reduce_kernel");
    return 1;
}
