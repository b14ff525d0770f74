// HRNet exchange unit: out = relu( sum of same-resolution terms + sum of bilinear(align_corners=True) upsampled
// coarse terms ), one pass. Reference: HighResolutionModule.forward, lib/models/backbones/hrnet/hrnet_backbone.py:
// 271-286 -- there every `y = y + F.interpolate(...)` is an interpolate kernel (full-size temporary) plus an add
// kernel, followed by a separate ReLU: up to 14 full-tensor passes per output branch; here 1 read per same-res term
// and 1 write. HBM-bound; no MFMA. Backward: masked gradient g = d_out * (out > 0) is the gradient of every
// same-resolution term; coarse terms get the exact adjoint gather (cseg_bilinear.h) with the mask fused in.
#include "cseg_common.h"
#include "cseg_bilinear.h"
#include "cseg_split.h"

namespace {

struct FuseArgs {
    const float* same[4];
    const float* low[3];
    int lh[3], lw[3];
    int n_same, n_low;
    int C, h, w;
};

// `amax` (may be null): max|out| record for the split-operand convolutions of the next unit (cseg_amax_f32's format), accumulated while
// the values are stored.
template <bool VEC>
__global__ __launch_bounds__(256) void fuse_sum_kernel(FuseArgs a, int relu, float* __restrict__ out, unsigned* __restrict__ amax) {
    constexpr int V = VEC ? 4 : 1;
    const int c = blockIdx.y, b = blockIdx.z;
    const int wv = a.w / V;
    const int e = blockIdx.x * 256 + threadIdx.x;
    const bool live = e < a.h * wv;
    float vmax = 0.f;
    if (live) {
        const int y = e / wv, x = (e - y * wv) * V;
        const size_t off = (((size_t)b * a.C + c) * a.h + y) * a.w + x;
        float acc[V];
#pragma unroll
        for (int t = 0; t < V; ++t) acc[t] = 0.f;
        for (int s = 0; s < a.n_same; ++s) {
            if (VEC) {
                const float4 v = *reinterpret_cast<const float4*>(a.same[s] + off);
                acc[0] += v.x; acc[1 % V] += v.y; acc[2 % V] += v.z; acc[3 % V] += v.w;
            } else {
                acc[0] += a.same[s][off];
            }
        }
        for (int l = 0; l < a.n_low; ++l) {
            const int hs = a.lh[l], ws = a.lw[l];
            const float sy = ac_scale(hs, a.h), sx = ac_scale(ws, a.w);
            const float fy = sy * (float)y;
            const int y0 = (int)fy;
            const int y1 = y0 + (y0 < hs - 1 ? 1 : 0);
            const float ly1 = fy - (float)y0, ly0 = 1.f - ly1;
            const float* r0 = a.low[l] + (((size_t)b * a.C + c) * hs + y0) * ws;
            const float* r1 = a.low[l] + (((size_t)b * a.C + c) * hs + y1) * ws;
#pragma unroll
            for (int t = 0; t < V; ++t) {
                const float fx = sx * (float)(x + t);
                const int x0 = (int)fx;
                const int x1 = x0 + (x0 < ws - 1 ? 1 : 0);
                const float lx1 = fx - (float)x0, lx0 = 1.f - lx1;
                acc[t] += ly0 * (lx0 * r0[x0] + lx1 * r0[x1]) + ly1 * (lx0 * r1[x0] + lx1 * r1[x1]);
            }
        }
        if (relu) {
#pragma unroll
            for (int t = 0; t < V; ++t) acc[t] = fmaxf(acc[t], 0.f);
        }
        if (VEC) *reinterpret_cast<float4*>(out + off) = make_float4(acc[0], acc[1 % V], acc[2 % V], acc[3 % V]);
        else out[off] = acc[0];
#pragma unroll
        for (int t = 0; t < V; ++t) vmax = fmaxf(vmax, fabsf(acc[t]));
    }
    if (amax) {                                    // (block-uniform) one candidate per wave
        unsigned bits = __builtin_bit_cast(unsigned, vmax);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) bits = max(bits, (unsigned)__shfl_xor((int)bits, o, 64));
        amax_publish_waves(bits, amax);
    }
}

__global__ __launch_bounds__(256) void relu_mask_kernel(const float* __restrict__ d_out, const float* __restrict__ act,
                                                        size_t n4, float* __restrict__ g) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const float4 d = reinterpret_cast<const float4*>(d_out)[i];
    const float4 a = reinterpret_cast<const float4*>(act)[i];
    reinterpret_cast<float4*>(g)[i] = make_float4(a.x > 0.f ? d.x : 0.f, a.y > 0.f ? d.y : 0.f,
                                                   a.z > 0.f ? d.z : 0.f, a.w > 0.f ? d.w : 0.f);
}

__global__ __launch_bounds__(256) void relu_mask_tail_kernel(const float* __restrict__ d_out,
                                                             const float* __restrict__ act, size_t start, size_t n,
                                                             float* __restrict__ g) {
    const size_t i = start + (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) g[i] = act[i] > 0.f ? d_out[i] : 0.f;
}

}  // namespace

static int fuse_sum_fwd_impl(const float* const* same, int n_same, const float* const* low, const int* low_h, const int* low_w, int n_low,
                             int B, int C, int h, int w, int relu, float* out, unsigned* amax, cseg_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    CSEG_REQUIRE(n_same >= 0 && n_same <= 4 && n_low >= 0 && n_low <= 3 && n_same + n_low >= 1,
                 "fuse_sum: n_same=%d n_low=%d out of range", n_same, n_low);
    CSEG_REQUIRE(B > 0 && C > 0 && h > 0 && w > 0 && C <= 65535 && B <= 65535, "fuse_sum: bad shape");
    FuseArgs a;
    a.n_same = n_same; a.n_low = n_low; a.C = C; a.h = h; a.w = w;
    for (int i = 0; i < 4; ++i) a.same[i] = i < n_same ? same[i] : nullptr;
    for (int i = 0; i < 3; ++i) {
        a.low[i] = i < n_low ? low[i] : nullptr;
        a.lh[i] = i < n_low ? low_h[i] : 1;
        a.lw[i] = i < n_low ? low_w[i] : 1;
        if (i < n_low) CSEG_REQUIRE(low_h[i] > 0 && low_w[i] > 0 && low_h[i] <= h && low_w[i] <= w,
                                    "fuse_sum: coarse term %d is %dx%d for a %dx%d output", i, low_h[i], low_w[i], h, w);
    }
    if (w % 4 == 0) {
        dim3 grid((h * (w / 4) + 255) / 256, C, B);
        hipLaunchKernelGGL(fuse_sum_kernel<true>, grid, dim3(256), 0, stream, a, relu, out, amax);
    } else {
        dim3 grid((h * w + 255) / 256, C, B);
        hipLaunchKernelGGL(fuse_sum_kernel<false>, grid, dim3(256), 0, stream, a, relu, out, amax);
    }
    CSEG_CHECK_LAUNCH("fuse_sum_kernel");
    return 1;
}

extern "C" int cseg_fuse_sum_fwd(const float* const* same, int n_same, const float* const* low, const int* low_h,
                                 const int* low_w, int n_low, int B, int C, int h, int w, int relu, float* out,
                                 cseg_stream_t stream) {
    return fuse_sum_fwd_impl(same, n_same, low, low_h, low_w, n_low, B, C, h, w, relu, out, nullptr, stream);
}

// the same with max|out| accumulated into `amax` (a zeroed record of CSEG_AMAX_WORDS words, as cseg_amax_f32 fills it)
extern "C" int cseg_fuse_sum_fwd_amax(const float* const* same, int n_same, const float* const* low, const int* low_h,
                                      const int* low_w, int n_low, int B, int C, int h, int w, int relu, float* out,
                                      unsigned* amax, cseg_stream_t stream) {
    CSEG_REQUIRE(amax, "fuse_sum_fwd_amax: null record");
    return fuse_sum_fwd_impl(same, n_same, low, low_h, low_w, n_low, B, C, h, w, relu, out, amax, stream);
}

extern "C" int cseg_fuse_sum_bwd(const float* d_out, const float* out_act, const int* low_h, const int* low_w, int n_low,
                                 int B, int C, int h, int w, float* g_same, float* const* d_low,
                                 cseg_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    CSEG_REQUIRE(n_low >= 0 && n_low <= 3 && B > 0 && C > 0 && h > 0 && w > 0, "fuse_sum_bwd: bad arguments");
    const size_t n = (size_t)B * C * h * w;
    if (g_same) {
        CSEG_REQUIRE(out_act, "fuse_sum_bwd: g_same needs the activation (without ReLU the gradient is d_out itself)");
        const size_t n4 = n / 4;
        if (n4) hipLaunchKernelGGL(relu_mask_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, stream, d_out,
                                   out_act, n4, g_same);
        if (n4 * 4 < n) hipLaunchKernelGGL(relu_mask_tail_kernel, dim3(1), dim3(256), 0, stream, d_out, out_act, n4 * 4,
                                           n, g_same);
        CSEG_CHECK_LAUNCH("relu_mask_kernel");
    }
    for (int i = 0; i < n_low; ++i) {
        if (!d_low[i]) continue;
        if (g_same)   // masked gradient already materialised for the same-resolution terms: read it instead of 2 tensors
            launch_bilinear_adjoint<false>(g_same, C, 0, C, low_h[i], low_w[i], h, w, B, nullptr, d_low[i], stream);
        else if (out_act)
            launch_bilinear_adjoint<true>(d_out, C, 0, C, low_h[i], low_w[i], h, w, B, out_act, d_low[i], stream);
        else
            launch_bilinear_adjoint<false>(d_out, C, 0, C, low_h[i], low_w[i], h, w, B, nullptr, d_low[i], stream);
        CSEG_CHECK_LAUNCH("bilinear_adjoint (fuse)");
    }
    return 1;
}

// ---- out[b][c][p] = a[c] + b[c] * u[b][c][p]: the dense part of the BatchNorm input gradient when the output gradient lives on N pixels
// (lib/models/modules/projection.py, row-sparse tail of the projection head: du = A_c + B_c u everywhere, plus N corrected rows), with
// max|out| accumulated for the split-operand convolution that reads it (torch.addcmul + a cseg_amax_f32 pass over 755 MB before).
namespace {
__global__ __launch_bounds__(256) void affine_channels_kernel(const float* __restrict__ u, const float* __restrict__ a,
                                                              const float* __restrict__ bcoef, int C, long P, int chunks,
                                                              float* __restrict__ out, unsigned* __restrict__ amax) {
    const long row = blockIdx.x / chunks;                      // (image, channel)
    const int chunk = blockIdx.x - (int)(row * chunks);
    const int c = (int)(row % C);
    const float av = a[c], bv = bcoef[c];
    const float* up = u + (size_t)row * P;
    float* op = out + (size_t)row * P;
    float vmax = 0.f;
    const long p4 = P / 4;
    for (long i = (long)chunk * 256 + threadIdx.x; i < p4; i += (long)chunks * 256) {
        const float4 v = reinterpret_cast<const float4*>(up)[i];
        const float4 o = make_float4(__builtin_fmaf(bv, v.x, av), __builtin_fmaf(bv, v.y, av), __builtin_fmaf(bv, v.z, av),
                                     __builtin_fmaf(bv, v.w, av));
        reinterpret_cast<float4*>(op)[i] = o;
        vmax = fmaxf(fmaxf(vmax, fmaxf(fabsf(o.x), fabsf(o.y))), fmaxf(fabsf(o.z), fabsf(o.w)));
    }
    if (chunk == 0) {
        for (long p = p4 * 4 + threadIdx.x; p < P; p += 256) {
            const float o = __builtin_fmaf(bv, up[p], av);
            op[p] = o;
            vmax = fmaxf(vmax, fabsf(o));
        }
    }
    if (amax) {
        unsigned bits = __builtin_bit_cast(unsigned, vmax);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) bits = max(bits, (unsigned)__shfl_xor((int)bits, o, 64));
        amax_publish_waves(bits, amax);
    }
}
}  // namespace

extern "C" int cseg_affine_channels(const float* u, const float* a, const float* b, int B, int C, long P, float* out, unsigned* amax,
                                    cseg_stream_t stream_) {
    CSEG_REQUIRE(u && a && b && out && B > 0 && C > 0 && P > 0, "affine_channels: bad arguments");
    CSEG_REQUIRE((P % 4 == 0 && ((reinterpret_cast<uintptr_t>(u) | reinterpret_cast<uintptr_t>(out)) & 15) == 0),
                 "affine_channels: rows must be 16-byte aligned (P %% 4 == 0, aligned tensors); got P=%ld", P);
    const long rows = (long)B * C;
    int chunks = (int)((P / 4 + 2047) / 2048);                 // <= 8 float4 per thread
    if (chunks < 1) chunks = 1;
    CSEG_REQUIRE(rows * chunks < 2147483647L, "affine_channels: grid too large");
    hipLaunchKernelGGL(affine_channels_kernel, dim3((unsigned)(rows * chunks)), dim3(256), 0, (hipStream_t)stream_, u, a, b, C, P, chunks, out,
                       amax);
    CSEG_CHECK_LAUNCH("affine_channels_kernel");
    return 1;
}
