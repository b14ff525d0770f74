// Fused (Sync)BatchNorm + residual add + ReLU for NCHW fp32 activations -- SURVEY.md section 8 (f2).
// Reference: ModuleHelper.BNReLU / BatchNorm2d (lib/models/tools/module_helper.py:29-68 -> nn.SyncBatchNorm), the
// `out = relu(bn(conv(x)))` / `out = relu(bn(conv(x)) + residual)` chains of the HRNet / ResNet blocks
// (lib/models/backbones/hrnet/hrnet_backbone.py:49-105, resnet/resnet_models.py) and the BNReLU heads.
// torch runs that as BN (statistics pass + normalise pass), an add kernel and a clamp kernel, each a full-tensor
// round trip through HBM, and in backward a threshold kernel + BN backward (two more statistics reads + dx pass).
// Here:
//   forward   bn_stats (1 read of x)                      -> per-channel (sum, sum of squares) in fp64
//             [one all-reduce of the packed [C,2] fp64 moments across ranks -- done by the host, SyncBN only]
//             bn_finalize (C threads)                     -> mean, 1/sqrt(var+eps), running statistics, batch counter
//             bn_apply (1 read of x [+1 of residual], 1 write): y = relu((x-mean)*invstd*gamma + beta [+ residual])
//   backward  bn_bwd_reduce (reads dy, x [, out]): sums of dy' and dy'*(x-mean) with the ReLU mask recomputed from x
//             (or taken from `out` when a residual was added; then dy' is also written once, it IS d_residual)
//             [one all-reduce of [C,2] fp64 sums -- SyncBN only; d_gamma / d_beta stay rank-local like torch's SyncBN]
//             bn_bwd_apply (reads dy, x; writes dx)
// All kernels are HBM-bound streaming passes: 16-byte loads/stores, one (image, channel) plane chunk of 4096 floats
// per 256-thread block iteration, >= 1024 blocks. Reductions are fixed-order (block partials -> fp64 per-channel sum):
// run-to-run deterministic. Statistics are accumulated as shifted sums (shift = first element of the channel) so
// that the fp32 partials do not cancel when |mean| >> std.
#include "cseg_split.h"

namespace {

constexpr int CHUNK = 4096;      // floats per block iteration (256 threads x 4 x float4)
constexpr int MAX_SPLITS = 64;   // partials per channel, reduced by one wave

struct BnDims {
    int B, C, HW;
    int n_ck;      // chunks per (image, channel) plane
    int S;         // splits (blocks) per channel in the reduction kernels
};

__host__ BnDims bn_dims(int B, int C, int HW) {
    BnDims d;
    d.B = B; d.C = C; d.HW = HW;
    d.n_ck = (HW + CHUNK - 1) / CHUNK;
    const long total = (long)B * d.n_ck;
    long target = (2048 + C - 1) / C;                    // aim at >= 2048 blocks
    if (target < 1) target = 1;
    if (target > MAX_SPLITS) target = MAX_SPLITS;
    if (target > total) target = total;
    const long per = (total + target - 1) / target;      // chunks per block
    d.S = (int)((total + per - 1) / per);
    return d;
}

__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// block-wide sum of two floats; result valid in thread 0
__device__ __forceinline__ void block_sum2(float& a, float& b, float (*red)[4]) {
    a = wave_sum(a);
    b = wave_sum(b);
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = a; red[1][threadIdx.x >> 6] = b; }
    __syncthreads();
    if (threadIdx.x == 0) {
        a = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
        b = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
    }
}

// ---------------------------------------------------------------------------------------------------------
// forward statistics
// ---------------------------------------------------------------------------------------------------------
template <bool VEC>
__global__ __launch_bounds__(256) void bn_stats_kernel(const float* __restrict__ x, BnDims d, float* __restrict__ partial) {
    __shared__ float red[2][4];
    const int c = blockIdx.x % d.C, s = blockIdx.x / d.C;
    const float shift = x[(size_t)c * d.HW];
    float s1 = 0.f, s2 = 0.f;
    const int total = d.B * d.n_ck;
    for (int q = s; q < total; q += d.S) {
        const int b = q / d.n_ck, ck = q - b * d.n_ck;
        const float* p = x + ((size_t)b * d.C + c) * d.HW + (size_t)ck * CHUNK;
        const int len = min(CHUNK, d.HW - ck * CHUNK);
        if (VEC) {
            if (len == CHUNK) {
                float4 v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const float4*>(p + (u * 256 + threadIdx.x) * 4);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const float a0 = v[u].x - shift, a1 = v[u].y - shift, a2 = v[u].z - shift, a3 = v[u].w - shift;
                    s1 += (a0 + a1) + (a2 + a3);
                    s2 += (a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3);
                }
            } else {
                for (int i = threadIdx.x * 4; i < len; i += 1024) {
                    const float4 v = *reinterpret_cast<const float4*>(p + i);
                    const float a0 = v.x - shift, a1 = v.y - shift, a2 = v.z - shift, a3 = v.w - shift;
                    s1 += (a0 + a1) + (a2 + a3);
                    s2 += (a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3);
                }
            }
        } else {
            for (int i = threadIdx.x; i < len; i += 256) {
                const float a = p[i] - shift;
                s1 += a;
                s2 += a * a;
            }
        }
    }
    block_sum2(s1, s2, red);
    if (threadIdx.x == 0) {
        partial[((size_t)s * d.C + c) * 2 + 0] = s1;
        partial[((size_t)s * d.C + c) * 2 + 1] = s2;
    }
}

// one wave per channel: partials -> fp64 raw moments (sum x, sum x^2) of this rank's n = B*HW values
__device__ __forceinline__ void channel_moments(const float* __restrict__ x, const float* __restrict__ partial, BnDims d,
                                                int c, int lane, double& m0, double& m1) {
    double s1 = 0.0, s2 = 0.0;
    for (int s = lane; s < d.S; s += 64) {
        s1 += (double)partial[((size_t)s * d.C + c) * 2 + 0];
        s2 += (double)partial[((size_t)s * d.C + c) * 2 + 1];
    }
    s1 = wave_sum_d(s1);
    s2 = wave_sum_d(s2);
    const double k = (double)x[(size_t)c * d.HW];
    const double n = (double)d.B * (double)d.HW;
    m0 = n * k + s1;
    m1 = s2 + 2.0 * k * s1 + n * k * k;
}

__global__ __launch_bounds__(256) void bn_moments_kernel(const float* __restrict__ x, const float* __restrict__ partial,
                                                         BnDims d, double* __restrict__ moments) {
    const int c = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (blockIdx.x == 0 && threadIdx.x == 0) {           // row C: this rank's element count per channel (summed by the exchange)
        moments[2 * d.C] = (double)d.B * (double)d.HW;
        moments[2 * d.C + 1] = 0.0;
    }
    if (c >= d.C) return;
    double m0, m1;
    channel_moments(x, partial, d, c, lane, m0, m1);
    if (lane == 0) { moments[2 * c] = m0; moments[2 * c + 1] = m1; }
}

__device__ __forceinline__ void finalize_channel(double m0, double m1, double count, float eps, float momentum,
                                                 float* running_mean, float* running_var, int c,
                                                 float* __restrict__ mean_invstd) {
    const double mean = m0 / count;
    double var = m1 / count - mean * mean;          // fp64: the cancellation is harmless at 1e-16
    if (var < 0.0) var = 0.0;
    mean_invstd[2 * c] = (float)mean;
    mean_invstd[2 * c + 1] = (float)(1.0 / sqrt(var + (double)eps));
    if (running_mean) {
        const double unbiased = count > 1.0 ? var * (count / (count - 1.0)) : var;
        running_mean[c] = (float)((1.0 - (double)momentum) * (double)running_mean[c] + (double)momentum * mean);
        running_var[c] = (float)((1.0 - (double)momentum) * (double)running_var[c] + (double)momentum * unbiased);
    }
}

// single-rank path: partials -> mean / invstd / running statistics in one launch
__global__ __launch_bounds__(256) void bn_moments_finalize_kernel(const float* __restrict__ x,
                                                                  const float* __restrict__ partial, BnDims d, float eps,
                                                                  float momentum, float* running_mean, float* running_var,
                                                                  int64_t* num_batches_tracked,
                                                                  float* __restrict__ mean_invstd) {
    const int c = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (blockIdx.x == 0 && threadIdx.x == 0 && num_batches_tracked) *num_batches_tracked += 1;
    if (c >= d.C) return;
    double m0, m1;
    channel_moments(x, partial, d, c, lane, m0, m1);
    if (lane == 0)
        finalize_channel(m0, m1, (double)d.B * (double)d.HW, eps, momentum, running_mean, running_var, c, mean_invstd);
}

// multi-rank path: globally summed moments -> mean / invstd / running statistics
__global__ __launch_bounds__(256) void bn_finalize_kernel(const double* __restrict__ moments, int C, double count, float eps,
                                                          float momentum, float* running_mean, float* running_var,
                                                          int64_t* num_batches_tracked, float* __restrict__ mean_invstd) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c == 0 && num_batches_tracked) *num_batches_tracked += 1;
    if (c >= C) return;
    if (count <= 0.0) count = moments[2 * C];            // the exchanged element count (row C), no host round trip
    finalize_channel(moments[2 * c], moments[2 * c + 1], count, eps, momentum, running_mean, running_var, c, mean_invstd);
}

// max|value written| of the block -> the block's slot of the max|.| record amax_out (csrc/cseg_split.h: what the f16x3
// convolutions scale their operands with).
__device__ __forceinline__ unsigned abs_bits(float v) { return __builtin_bit_cast(unsigned, v) & 0x7fffffffu; }
__device__ __forceinline__ void publish_amax(unsigned m, unsigned* __restrict__ amax_out) {
    __shared__ unsigned amax_red[4];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o, 64));
    if ((threadIdx.x & 63) == 0) amax_red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) amax_publish_block(max(max(amax_red[0], amax_red[1]), max(amax_red[2], amax_red[3])), amax_out);
}

// ---------------------------------------------------------------------------------------------------------
// forward apply: y = act((x - mean) * (gamma * invstd) + beta [+ residual])
// ---------------------------------------------------------------------------------------------------------
// (`bid`: the block's index inside THIS layer's grid -- blockIdx.x for the one-layer launch, blockIdx.x minus the member's first block
// in a grouped launch, bn_group_* below)
template <bool VEC, bool RES>
__device__ __forceinline__ void bn_apply_body(const float* __restrict__ x, const float* __restrict__ res,
                                              const float* __restrict__ mean_invstd,
                                              const float* __restrict__ weight, const float* __restrict__ bias,
                                              const BnDims& d, int relu, float* __restrict__ y, unsigned* __restrict__ amax_out, int bid) {
    const int plane = bid / d.n_ck, ck = bid - plane * d.n_ck;
    const int c = plane % d.C;
    const float mean = mean_invstd[2 * c];
    const float a = (weight ? weight[c] : 1.f) * mean_invstd[2 * c + 1];
    const float beta = bias ? bias[c] : 0.f;
    const size_t base = (size_t)plane * d.HW + (size_t)ck * CHUNK;
    const int len = min(CHUNK, d.HW - ck * CHUNK);
    const bool rl = relu != 0;      // NaN-propagating ReLU like torch's clamp: (v < 0) ? 0 : v
    unsigned am = 0;
    if (VEC) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = (u * 256 + threadIdx.x) * 4;
            if (i < len) {
                const float4 v = *reinterpret_cast<const float4*>(x + base + i);
                float4 o;
                o.x = fmaf(v.x - mean, a, beta); o.y = fmaf(v.y - mean, a, beta);
                o.z = fmaf(v.z - mean, a, beta); o.w = fmaf(v.w - mean, a, beta);
                if (RES) {
                    const float4 r = *reinterpret_cast<const float4*>(res + base + i);
                    o.x += r.x; o.y += r.y; o.z += r.z; o.w += r.w;
                }
                o.x = (rl && o.x < 0.f) ? 0.f : o.x; o.y = (rl && o.y < 0.f) ? 0.f : o.y;
                o.z = (rl && o.z < 0.f) ? 0.f : o.z; o.w = (rl && o.w < 0.f) ? 0.f : o.w;
                *reinterpret_cast<float4*>(y + base + i) = o;
                am = max(max(am, abs_bits(o.x)), max(abs_bits(o.y), max(abs_bits(o.z), abs_bits(o.w))));
            }
        }
    } else {
        for (int i = threadIdx.x; i < len; i += 256) {
            float o = fmaf(x[base + i] - mean, a, beta);
            if (RES) o += res[base + i];
            o = (rl && o < 0.f) ? 0.f : o;
            y[base + i] = o;
            am = max(am, abs_bits(o));
        }
    }
    if (amax_out) publish_amax(am, amax_out);
}
template <bool VEC, bool RES>
__global__ __launch_bounds__(256) void bn_apply_kernel(const float* __restrict__ x, const float* __restrict__ res,
                                                       const float* __restrict__ mean_invstd,
                                                       const float* __restrict__ weight, const float* __restrict__ bias,
                                                       BnDims d, int relu, float* __restrict__ y, unsigned* __restrict__ amax_out) {
    bn_apply_body<VEC, RES>(x, res, mean_invstd, weight, bias, d, relu, y, amax_out, (int)blockIdx.x);
}

// ---------------------------------------------------------------------------------------------------------
// backward reduction: sums of dy' and dy' * (x - mean); dy' = dy masked by the ReLU
//   MODE 0: no activation          MODE 1: mask recomputed from x ((x-mean)*a+beta > 0)
//   MODE 2: mask from `out` (residual case); dy' is written to g_out (it is also the gradient of the residual)
// ---------------------------------------------------------------------------------------------------------
template <bool VEC, int MODE>
__device__ __forceinline__ void bn_bwd_reduce_body(const float* __restrict__ dy, const float* __restrict__ x,
                                                   const float* __restrict__ out,
                                                   const float* __restrict__ mean_invstd,
                                                   const float* __restrict__ weight,
                                                   const float* __restrict__ bias, const BnDims& d,
                                                   float* __restrict__ g_out, float* __restrict__ partial, int bid) {
    __shared__ float red[2][4];
    const int c = bid % d.C, s = bid / d.C;
    const float mean = mean_invstd[2 * c];
    const float a = (weight ? weight[c] : 1.f) * mean_invstd[2 * c + 1];
    const float beta = bias ? bias[c] : 0.f;
    float s0 = 0.f, s1 = 0.f;
    const int total = d.B * d.n_ck;
    for (int q = s; q < total; q += d.S) {
        const int b = q / d.n_ck, ck = q - b * d.n_ck;
        const size_t base = ((size_t)b * d.C + c) * d.HW + (size_t)ck * CHUNK;
        const int len = min(CHUNK, d.HW - ck * CHUNK);
        if (VEC) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = (u * 256 + threadIdx.x) * 4;
                if (i < len) {
                    float4 g = *reinterpret_cast<const float4*>(dy + base + i);
                    const float4 v = *reinterpret_cast<const float4*>(x + base + i);
                    const float x0 = v.x - mean, x1 = v.y - mean, x2 = v.z - mean, x3 = v.w - mean;
                    if (MODE == 1) {
                        g.x = fmaf(x0, a, beta) > 0.f ? g.x : 0.f; g.y = fmaf(x1, a, beta) > 0.f ? g.y : 0.f;
                        g.z = fmaf(x2, a, beta) > 0.f ? g.z : 0.f; g.w = fmaf(x3, a, beta) > 0.f ? g.w : 0.f;
                    } else if (MODE == 2) {
                        const float4 o = *reinterpret_cast<const float4*>(out + base + i);
                        g.x = o.x > 0.f ? g.x : 0.f; g.y = o.y > 0.f ? g.y : 0.f;
                        g.z = o.z > 0.f ? g.z : 0.f; g.w = o.w > 0.f ? g.w : 0.f;
                        *reinterpret_cast<float4*>(g_out + base + i) = g;
                    }
                    s0 += (g.x + g.y) + (g.z + g.w);
                    s1 += (g.x * x0 + g.y * x1) + (g.z * x2 + g.w * x3);
                }
            }
        } else {
            for (int i = threadIdx.x; i < len; i += 256) {
                float g = dy[base + i];
                const float xm = x[base + i] - mean;
                if (MODE == 1) g = fmaf(xm, a, beta) > 0.f ? g : 0.f;
                if (MODE == 2) { g = out[base + i] > 0.f ? g : 0.f; g_out[base + i] = g; }
                s0 += g;
                s1 += g * xm;
            }
        }
    }
    block_sum2(s0, s1, red);
    if (threadIdx.x == 0) {
        partial[((size_t)s * d.C + c) * 2 + 0] = s0;
        partial[((size_t)s * d.C + c) * 2 + 1] = s1;
    }
}
template <bool VEC, int MODE>
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                            const float* __restrict__ out,
                                                            const float* __restrict__ mean_invstd,
                                                            const float* __restrict__ weight,
                                                            const float* __restrict__ bias, BnDims d,
                                                            float* __restrict__ g_out, float* __restrict__ partial) {
    bn_bwd_reduce_body<VEC, MODE>(dy, x, out, mean_invstd, weight, bias, d, g_out, partial, (int)blockIdx.x);
}

// one wave per channel: partials -> fp64 sums [C,2]; d_gamma = s1 * invstd, d_beta = s0 (rank-local, like torch's SyncBN)
__global__ __launch_bounds__(256) void bn_bwd_sums_kernel(const float* __restrict__ partial, BnDims d,
                                                          const float* __restrict__ mean_invstd, double* __restrict__ sums,
                                                          float* __restrict__ d_weight, float* __restrict__ d_bias) {
    const int c = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (blockIdx.x == 0 && threadIdx.x == 0) {           // row C: this rank's element count per channel
        sums[2 * d.C] = (double)d.B * (double)d.HW;
        sums[2 * d.C + 1] = 0.0;
    }
    if (c >= d.C) return;
    double s0 = 0.0, s1 = 0.0;
    for (int s = lane; s < d.S; s += 64) {
        s0 += (double)partial[((size_t)s * d.C + c) * 2 + 0];
        s1 += (double)partial[((size_t)s * d.C + c) * 2 + 1];
    }
    s0 = wave_sum_d(s0);
    s1 = wave_sum_d(s1);
    if (lane == 0) {
        sums[2 * c] = s0;
        sums[2 * c + 1] = s1;
        if (d_weight) d_weight[c] = (float)(s1 * (double)mean_invstd[2 * c + 1]);
        if (d_bias) d_bias[c] = (float)s0;
    }
}

// dx = a * (dy' - sum(dy')/N - (x-mean) * invstd^2 * sum(dy'*(x-mean))/N);  with frozen statistics (eval) dx = a * dy'
template <bool VEC, bool MASK>
__device__ __forceinline__ void bn_bwd_apply_body(const float* __restrict__ dy, const float* __restrict__ x,
                                                  const float* __restrict__ mean_invstd,
                                                  const float* __restrict__ weight,
                                                  const float* __restrict__ bias, const double* __restrict__ sums,
                                                  double inv_count, const BnDims& d, float* __restrict__ dx, unsigned* __restrict__ amax_out, int bid) {
    const int plane = bid / d.n_ck, ck = bid - plane * d.n_ck;
    const int c = plane % d.C;
    const float mean = mean_invstd[2 * c], invstd = mean_invstd[2 * c + 1];
    const float a = (weight ? weight[c] : 1.f) * invstd;
    const float beta = bias ? bias[c] : 0.f;
    // k0 = mean(dy') is subtracted from EVERY element: a float-rounded k0 would shift all of them by the same ~6e-8*|k0|,
    // an error that adds up coherently in the weight gradient of the convolution in front (sum_p dx_p * input_p, where
    // sum_p dx_p = 0 in exact arithmetic). Carry it as hi + lo.
    if (sums && inv_count <= 0.0) inv_count = 1.0 / sums[2 * d.C];      // the exchanged element count (row C)
    const double k0d = sums ? sums[2 * c] * inv_count : 0.0;
    const float k0 = (float)k0d, k0l = (float)(k0d - (double)k0);
    const float k1 = sums ? (float)(sums[2 * c + 1] * inv_count * (double)invstd * (double)invstd) : 0.f;
    const size_t base = (size_t)plane * d.HW + (size_t)ck * CHUNK;
    const int len = min(CHUNK, d.HW - ck * CHUNK);
    unsigned am = 0;
    if (VEC) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = (u * 256 + threadIdx.x) * 4;
            if (i < len) {
                float4 g = *reinterpret_cast<const float4*>(dy + base + i);
                const float4 v = *reinterpret_cast<const float4*>(x + base + i);
                const float x0 = v.x - mean, x1 = v.y - mean, x2 = v.z - mean, x3 = v.w - mean;
                if (MASK) {
                    g.x = fmaf(x0, a, beta) > 0.f ? g.x : 0.f; g.y = fmaf(x1, a, beta) > 0.f ? g.y : 0.f;
                    g.z = fmaf(x2, a, beta) > 0.f ? g.z : 0.f; g.w = fmaf(x3, a, beta) > 0.f ? g.w : 0.f;
                }
                float4 o;
                o.x = a * ((g.x - k0) - k0l - x0 * k1); o.y = a * ((g.y - k0) - k0l - x1 * k1);
                o.z = a * ((g.z - k0) - k0l - x2 * k1); o.w = a * ((g.w - k0) - k0l - x3 * k1);
                *reinterpret_cast<float4*>(dx + base + i) = o;
                am = max(max(am, abs_bits(o.x)), max(abs_bits(o.y), max(abs_bits(o.z), abs_bits(o.w))));
            }
        }
    } else {
        for (int i = threadIdx.x; i < len; i += 256) {
            float g = dy[base + i];
            const float xm = x[base + i] - mean;
            if (MASK) g = fmaf(xm, a, beta) > 0.f ? g : 0.f;
            const float o = a * ((g - k0) - k0l - xm * k1);
            dx[base + i] = o;
            am = max(am, abs_bits(o));
        }
    }
    if (amax_out) publish_amax(am, amax_out);
}
template <bool VEC, bool MASK>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                           const float* __restrict__ mean_invstd,
                                                           const float* __restrict__ weight,
                                                           const float* __restrict__ bias, const double* __restrict__ sums,
                                                           double inv_count, BnDims d, float* __restrict__ dx, unsigned* __restrict__ amax_out) {
    bn_bwd_apply_body<VEC, MASK>(dy, x, mean_invstd, weight, bias, sums, inv_count, d, dx, amax_out, (int)blockIdx.x);
}


// ---------------------------------------------------------------------------------------------------------
// single-rank fast path: the per-channel finalisation (partials -> mean / invstd, running statistics) is done by
// wave 0 of every apply block instead of a separate 5-us launch per layer (307 layers per step); the block of image 0,
// chunk 0 of each channel publishes mean_invstd for the backward and updates the running statistics.
// ---------------------------------------------------------------------------------------------------------
struct BnFinalize {
    const float* partial;
    float eps, momentum;
    float* running_mean;
    float* running_var;
    int64_t* num_batches_tracked;
    float* mean_invstd;           // out [C,2]
};

template <bool VEC, bool RES>
__global__ __launch_bounds__(256) void bn_apply_fused_kernel(const float* __restrict__ x, const float* __restrict__ res,
                                                             BnFinalize f, const float* __restrict__ weight,
                                                             const float* __restrict__ bias, BnDims d, int relu,
                                                             float* __restrict__ y, unsigned* __restrict__ amax_out) {
    __shared__ float mi[2];
    const int plane = blockIdx.x / d.n_ck, ck = blockIdx.x - plane * d.n_ck;
    const int c = plane % d.C;
    if (threadIdx.x < 64) {
        double m0, m1;
        channel_moments(x, f.partial, d, c, threadIdx.x, m0, m1);
        if (threadIdx.x == 0) {
            const double count = (double)d.B * (double)d.HW;
            const double mean = m0 / count;
            double var = m1 / count - mean * mean;
            if (var < 0.0) var = 0.0;
            mi[0] = (float)mean;
            mi[1] = (float)(1.0 / sqrt(var + (double)f.eps));
            if (plane == c && ck == 0) {              // image 0, first chunk: the channel's publisher
                f.mean_invstd[2 * c] = mi[0];
                f.mean_invstd[2 * c + 1] = mi[1];
                if (f.running_mean) {
                    const double unbiased = count > 1.0 ? var * (count / (count - 1.0)) : var;
                    f.running_mean[c] = (float)((1.0 - (double)f.momentum) * (double)f.running_mean[c] +
                                                (double)f.momentum * mean);
                    f.running_var[c] = (float)((1.0 - (double)f.momentum) * (double)f.running_var[c] +
                                               (double)f.momentum * unbiased);
                }
                if (c == 0 && f.num_batches_tracked) *f.num_batches_tracked += 1;
            }
        }
    }
    __syncthreads();
    const float mean = mi[0];
    const float a = (weight ? weight[c] : 1.f) * mi[1];
    const float beta = bias ? bias[c] : 0.f;
    const size_t base = (size_t)plane * d.HW + (size_t)ck * CHUNK;
    const int len = min(CHUNK, d.HW - ck * CHUNK);
    const bool rl = relu != 0;
    unsigned am = 0;
    if (VEC) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = (u * 256 + threadIdx.x) * 4;
            if (i < len) {
                const float4 v = *reinterpret_cast<const float4*>(x + base + i);
                float4 o;
                o.x = fmaf(v.x - mean, a, beta); o.y = fmaf(v.y - mean, a, beta);
                o.z = fmaf(v.z - mean, a, beta); o.w = fmaf(v.w - mean, a, beta);
                if (RES) {
                    const float4 r = *reinterpret_cast<const float4*>(res + base + i);
                    o.x += r.x; o.y += r.y; o.z += r.z; o.w += r.w;
                }
                o.x = (rl && o.x < 0.f) ? 0.f : o.x; o.y = (rl && o.y < 0.f) ? 0.f : o.y;
                o.z = (rl && o.z < 0.f) ? 0.f : o.z; o.w = (rl && o.w < 0.f) ? 0.f : o.w;
                *reinterpret_cast<float4*>(y + base + i) = o;
                am = max(max(am, abs_bits(o.x)), max(abs_bits(o.y), max(abs_bits(o.z), abs_bits(o.w))));
            }
        }
    } else {
        for (int i = threadIdx.x; i < len; i += 256) {
            float o = fmaf(x[base + i] - mean, a, beta);
            if (RES) o += res[base + i];
            o = (rl && o < 0.f) ? 0.f : o;
            y[base + i] = o;
            am = max(am, abs_bits(o));
        }
    }
    if (amax_out) publish_amax(am, amax_out);
}

// backward twin: every bwd-apply block reduces the channel's partial sums itself; the publisher block writes
// d_weight / d_bias (and the fp64 sums, kept for inspection)
template <bool VEC, bool MASK>
__device__ __forceinline__ void bn_bwd_apply_fused_body(const float* __restrict__ dy, const float* __restrict__ x,
                                                        const float* __restrict__ mean_invstd,
                                                        const float* __restrict__ weight,
                                                        const float* __restrict__ bias,
                                                        const float* __restrict__ partial, int training,
                                                        const BnDims& d, float* __restrict__ d_weight,
                                                        float* __restrict__ d_bias, float* __restrict__ dx, unsigned* __restrict__ amax_out, int bid) {
    __shared__ float kk[3];            // k0 (hi), k1, k0 (lo): see bn_bwd_apply_kernel
    const int plane = bid / d.n_ck, ck = bid - plane * d.n_ck;
    const int c = plane % d.C;
    const float mean = mean_invstd[2 * c], invstd = mean_invstd[2 * c + 1];
    if (threadIdx.x < 64) {
        double s0 = 0.0, s1 = 0.0;
        for (int s = threadIdx.x; s < d.S; s += 64) {
            s0 += (double)partial[((size_t)s * d.C + c) * 2 + 0];
            s1 += (double)partial[((size_t)s * d.C + c) * 2 + 1];
        }
        s0 = wave_sum_d(s0);
        s1 = wave_sum_d(s1);
        if (threadIdx.x == 0) {
            const double inv = 1.0 / ((double)d.B * (double)d.HW);
            const double k0d = training ? s0 * inv : 0.0;
            kk[0] = (float)k0d;
            kk[2] = (float)(k0d - (double)kk[0]);
            kk[1] = training ? (float)(s1 * inv * (double)invstd * (double)invstd) : 0.f;
            if (plane == c && ck == 0) {
                if (d_weight) d_weight[c] = (float)(s1 * (double)invstd);
                if (d_bias) d_bias[c] = (float)s0;
            }
        }
    }
    __syncthreads();
    const float a = (weight ? weight[c] : 1.f) * invstd;
    const float beta = bias ? bias[c] : 0.f;
    const float k0 = kk[0], k1 = kk[1], k0l = kk[2];
    const size_t base = (size_t)plane * d.HW + (size_t)ck * CHUNK;
    const int len = min(CHUNK, d.HW - ck * CHUNK);
    unsigned am = 0;
    if (VEC) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = (u * 256 + threadIdx.x) * 4;
            if (i < len) {
                float4 g = *reinterpret_cast<const float4*>(dy + base + i);
                const float4 v = *reinterpret_cast<const float4*>(x + base + i);
                const float x0 = v.x - mean, x1 = v.y - mean, x2 = v.z - mean, x3 = v.w - mean;
                if (MASK) {
                    g.x = fmaf(x0, a, beta) > 0.f ? g.x : 0.f; g.y = fmaf(x1, a, beta) > 0.f ? g.y : 0.f;
                    g.z = fmaf(x2, a, beta) > 0.f ? g.z : 0.f; g.w = fmaf(x3, a, beta) > 0.f ? g.w : 0.f;
                }
                float4 o;
                o.x = a * ((g.x - k0) - k0l - x0 * k1); o.y = a * ((g.y - k0) - k0l - x1 * k1);
                o.z = a * ((g.z - k0) - k0l - x2 * k1); o.w = a * ((g.w - k0) - k0l - x3 * k1);
                *reinterpret_cast<float4*>(dx + base + i) = o;
                am = max(max(am, abs_bits(o.x)), max(abs_bits(o.y), max(abs_bits(o.z), abs_bits(o.w))));
            }
        }
    } else {
        for (int i = threadIdx.x; i < len; i += 256) {
            float g = dy[base + i];
            const float xm = x[base + i] - mean;
            if (MASK) g = fmaf(xm, a, beta) > 0.f ? g : 0.f;
            const float o = a * ((g - k0) - k0l - xm * k1);
            dx[base + i] = o;
            am = max(am, abs_bits(o));
        }
    }
    if (amax_out) publish_amax(am, amax_out);
}
template <bool VEC, bool MASK>
__global__ __launch_bounds__(256) void bn_bwd_apply_fused_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                                 const float* __restrict__ mean_invstd,
                                                                 const float* __restrict__ weight,
                                                                 const float* __restrict__ bias,
                                                                 const float* __restrict__ partial, int training,
                                                                 BnDims d, float* __restrict__ d_weight,
                                                                 float* __restrict__ d_bias, float* __restrict__ dx, unsigned* __restrict__ amax_out) {
    bn_bwd_apply_fused_body<VEC, MASK>(dy, x, mean_invstd, weight, bias, partial, training, d, d_weight, d_bias, dx, amax_out, (int)blockIdx.x);
}

// ---------------------------------------------------------------------------------------------------------
// Statistics that came out of the producing convolution's epilogue (cseg_stats.h): per channel T float4 = (count, mean, M2) of
// 64-pixel segments. One BLOCK per channel turns them into raw fp64 moments in a single pass -- n, sum n_t mean_t,
// sum (M2_t + n_t mean_t^2) -- and var = m1 / n - mean^2 is then taken in fp64, where the cancellation is harmless (1e-16; the
// fp32 segment records themselves are mean-centred, so nothing was lost before). Fixed order: deterministic.
// (First version, GPU call r04j6: one WAVE per channel, two passes of T / 64 dependent 16-byte loads -- 100 us per layer at
// T = 4096, ten times the statistics pass it replaces; the step got 3 ms slower. Hence a block per channel, loads four deep.)
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void tiles_combine(const float4* __restrict__ st, long T, int c, double& count, double& m0, double& m1) {
    __shared__ double red[3][4];
    const float4* p = st + (size_t)c * T;
    double n = 0.0, s = 0.0, q = 0.0;
    long t = threadIdx.x;
    for (; t + 768 < T; t += 1024) {
        const float4 v0 = p[t], v1 = p[t + 256], v2 = p[t + 512], v3 = p[t + 768];
        const double a0 = (double)v0.x * (double)v0.y, a1 = (double)v1.x * (double)v1.y, a2 = (double)v2.x * (double)v2.y,
                     a3 = (double)v3.x * (double)v3.y;
        n += ((double)v0.x + (double)v1.x) + ((double)v2.x + (double)v3.x);
        s += (a0 + a1) + (a2 + a3);
        q += (((double)v0.z + a0 * (double)v0.y) + ((double)v1.z + a1 * (double)v1.y)) +
             (((double)v2.z + a2 * (double)v2.y) + ((double)v3.z + a3 * (double)v3.y));
    }
    for (; t < T; t += 256) {
        const float4 v = p[t];
        const double a = (double)v.x * (double)v.y;
        n += (double)v.x;
        s += a;
        q += (double)v.z + a * (double)v.y;
    }
    n = wave_sum_d(n);
    s = wave_sum_d(s);
    q = wave_sum_d(q);
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = n; red[1][threadIdx.x >> 6] = s; red[2][threadIdx.x >> 6] = q; }
    __syncthreads();
    count = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
    m0 = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
    m1 = (red[2][0] + red[2][1]) + (red[2][2] + red[2][3]);
}

__device__ __forceinline__ void bn_tiles_finalize_body(const float4* __restrict__ st, long T, float eps, float momentum,
                                                       float* running_mean, float* running_var, int64_t* num_batches_tracked,
                                                       float* __restrict__ mean_invstd, int c) {
    if (c == 0 && threadIdx.x == 0 && num_batches_tracked) *num_batches_tracked += 1;
    double count, m0, m1;
    tiles_combine(st, T, c, count, m0, m1);
    if (threadIdx.x == 0) finalize_channel(m0, m1, count > 0.0 ? count : 1.0, eps, momentum, running_mean, running_var, c, mean_invstd);
}
__global__ __launch_bounds__(256) void bn_tiles_finalize_kernel(const float4* __restrict__ st, long T, int C, float eps, float momentum,
                                                                float* running_mean, float* running_var, int64_t* num_batches_tracked,
                                                                float* __restrict__ mean_invstd) {
    bn_tiles_finalize_body(st, T, eps, momentum, running_mean, running_var, num_batches_tracked, mean_invstd, (int)blockIdx.x);
}

// SyncBN form: the raw fp64 moments [C+1, 2] the exchange all-reduces (row C = this rank's element count)
__global__ __launch_bounds__(256) void bn_tiles_moments_kernel(const float4* __restrict__ st, long T, int C, double* __restrict__ moments) {
    const int c = blockIdx.x;
    double count, m0, m1;
    tiles_combine(st, T, c, count, m0, m1);
    if (threadIdx.x == 0) {
        moments[2 * c] = m0;
        moments[2 * c + 1] = m1;
        if (c == 0) { moments[2 * C] = count; moments[2 * C + 1] = 0.0; }
    }
}

bool vec_ok(int HW, const void* p0, const void* p1 = nullptr, const void* p2 = nullptr, const void* p3 = nullptr) {
    auto al = [](const void* p) { return p == nullptr || (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    return HW % 4 == 0 && al(p0) && al(p1) && al(p2) && al(p3);
}

int check_dims(const char* who, int B, int C, int HW) {
    CSEG_REQUIRE(B > 0 && C > 0 && HW > 0 && (long)B * C * ((HW + CHUNK - 1) / CHUNK) < 2147483647L,
                 "%s: bad shape B=%d C=%d HW=%d", who, B, C, HW);
    return 1;
}

}  // namespace

extern "C" size_t cseg_bn_ws_floats(int B, int C, int HW) {
    if (B <= 0 || C <= 0 || HW <= 0) return 0;
    const BnDims d = bn_dims(B, C, HW);
    return (size_t)d.S * C * 2 + (size_t)C * 4 + 8;      // partials + room for [C+1,2] fp64 sums (cseg_bn_bwd without dx)
}

extern "C" int cseg_bn_stats(const float* x, int B, int C, int HW, float* ws, double* moments, cseg_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!check_dims("bn_stats", B, C, HW)) return 0;
    CSEG_REQUIRE(x && ws && moments, "bn_stats: null pointer");
    const BnDims d = bn_dims(B, C, HW);
    if (vec_ok(HW, x)) hipLaunchKernelGGL(bn_stats_kernel<true>, dim3(d.S * C), dim3(256), 0, stream, x, d, ws);
    else hipLaunchKernelGGL(bn_stats_kernel<false>, dim3(d.S * C), dim3(256), 0, stream, x, d, ws);
    hipLaunchKernelGGL(bn_moments_kernel, dim3((C + 3) / 4), dim3(256), 0, stream, x, ws, d, moments);
    CSEG_CHECK_LAUNCH("bn_stats");
    return 1;
}

extern "C" int cseg_bn_finalize(const double* moments, int C, double count, float eps, float momentum, float* running_mean,
                                float* running_var, int64_t* num_batches_tracked, float* mean_invstd,
                                cseg_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    CSEG_REQUIRE(moments && mean_invstd && C > 0 && count >= 0.0, "bn_finalize: bad arguments");
    CSEG_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "bn_finalize: running_mean/var must come together");
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 255) / 256), dim3(256), 0, stream, moments, C, count, eps, momentum,
                       running_mean, running_var, num_batches_tracked, mean_invstd);
    CSEG_CHECK_LAUNCH("bn_finalize");
    return 1;
}

extern "C" int cseg_bn_tiles_finalize(const float* stats, int C, long T, float eps, float momentum, float* running_mean,
                                      float* running_var, int64_t* num_batches_tracked, float* mean_invstd, cseg_stream_t stream_) {
    CSEG_REQUIRE(stats && mean_invstd && C > 0 && T > 0, "bn_tiles_finalize: bad arguments");
    CSEG_REQUIRE((reinterpret_cast<uintptr_t>(stats) & 15) == 0, "bn_tiles_finalize: the statistics buffer must be 16-byte aligned");
    CSEG_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "bn_tiles_finalize: running_mean/var must come together");
    hipLaunchKernelGGL(bn_tiles_finalize_kernel, dim3(C), dim3(256), 0, (hipStream_t)stream_,
                       reinterpret_cast<const float4*>(stats), T, C, eps, momentum, running_mean, running_var, num_batches_tracked,
                       mean_invstd);
    CSEG_CHECK_LAUNCH("bn_tiles_finalize");
    return 1;
}

extern "C" int cseg_bn_tiles_moments(const float* stats, int C, long T, double* moments, cseg_stream_t stream_) {
    CSEG_REQUIRE(stats && moments && C > 0 && T > 0, "bn_tiles_moments: bad arguments");
    CSEG_REQUIRE((reinterpret_cast<uintptr_t>(stats) & 15) == 0, "bn_tiles_moments: the statistics buffer must be 16-byte aligned");
    hipLaunchKernelGGL(bn_tiles_moments_kernel, dim3(C), dim3(256), 0, (hipStream_t)stream_,
                       reinterpret_cast<const float4*>(stats), T, C, moments);
    CSEG_CHECK_LAUNCH("bn_tiles_moments");
    return 1;
}

extern "C" int cseg_bn_stats_finalize(const float* x, int B, int C, int HW, float* ws, float eps, float momentum,
                                      float* running_mean, float* running_var, int64_t* num_batches_tracked,
                                      float* mean_invstd, cseg_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!check_dims("bn_stats_finalize", B, C, HW)) return 0;
    CSEG_REQUIRE(x && ws && mean_invstd, "bn_stats_finalize: null pointer");
    CSEG_REQUIRE((running_mean == nullptr) == (running_var == nullptr),
                 "bn_stats_finalize: running_mean/var must come together");
    const BnDims d = bn_dims(B, C, HW);
    if (vec_ok(HW, x)) hipLaunchKernelGGL(bn_stats_kernel<true>, dim3(d.S * C), dim3(256), 0, stream, x, d, ws);
    else hipLaunchKernelGGL(bn_stats_kernel<false>, dim3(d.S * C), dim3(256), 0, stream, x, d, ws);
    hipLaunchKernelGGL(bn_moments_finalize_kernel, dim3((C + 3) / 4), dim3(256), 0, stream, x, ws, d, eps, momentum,
                       running_mean, running_var, num_batches_tracked, mean_invstd);
    CSEG_CHECK_LAUNCH("bn_stats_finalize");
    return 1;
}

static int bn_apply_impl(const float* x, const float* residual, const float* mean_invstd, const float* weight,
                         const float* bias, int relu, int B, int C, int HW, float* y, unsigned* amax_out, cseg_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!check_dims("bn_apply", B, C, HW)) return 0;
    CSEG_REQUIRE(x && mean_invstd && y, "bn_apply: null pointer");
    const BnDims d = bn_dims(B, C, HW);
    const dim3 grid((unsigned)((long)B * C * d.n_ck));
    const bool v = vec_ok(HW, x, residual, y);
#define LAUNCH(V, R) hipLaunchKernelGGL((bn_apply_kernel<V, R>), grid, dim3(256), 0, stream, x, residual, mean_invstd, \
                                        weight, bias, d, relu, y, amax_out)
    if (v) { if (residual) LAUNCH(true, true); else LAUNCH(true, false); }
    else { if (residual) LAUNCH(false, true); else LAUNCH(false, false); }
#undef LAUNCH
    CSEG_CHECK_LAUNCH("bn_apply");
    return 1;
}

extern "C" int cseg_bn_bwd_reduce(const float* dy, const float* x, const float* out, const float* mean_invstd,
                                  const float* weight, const float* bias, int mode, int B, int C, int HW, float* ws,
                                  float* g_masked, double* sums, float* d_weight, float* d_bias, cseg_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!check_dims("bn_bwd_reduce", B, C, HW)) return 0;
    CSEG_REQUIRE(dy && x && mean_invstd && ws && sums, "bn_bwd_reduce: null pointer");
    CSEG_REQUIRE(mode >= 0 && mode <= 2, "bn_bwd_reduce: mode %d", mode);
    CSEG_REQUIRE(mode != 2 || (out && g_masked), "bn_bwd_reduce: mode 2 needs `out` and `g_masked`");
    const BnDims d = bn_dims(B, C, HW);
    const bool v = vec_ok(HW, dy, x, out, g_masked);
    const dim3 grid(d.S * C);
#define LAUNCH(V, M) hipLaunchKernelGGL((bn_bwd_reduce_kernel<V, M>), grid, dim3(256), 0, stream, dy, x, out, mean_invstd, \
                                        weight, bias, d, g_masked, ws)
    if (v) { if (mode == 0) LAUNCH(true, 0); else if (mode == 1) LAUNCH(true, 1); else LAUNCH(true, 2); }
    else { if (mode == 0) LAUNCH(false, 0); else if (mode == 1) LAUNCH(false, 1); else LAUNCH(false, 2); }
#undef LAUNCH
    hipLaunchKernelGGL(bn_bwd_sums_kernel, dim3((C + 3) / 4), dim3(256), 0, stream, ws, d, mean_invstd, sums, d_weight,
                       d_bias);
    CSEG_CHECK_LAUNCH("bn_bwd_reduce");
    return 1;
}

static int bn_bwd_apply_impl(const float* dy, const float* x, const float* mean_invstd, const float* weight,
                             const float* bias, const double* sums, double count, int mask_from_x, int B, int C,
                             int HW, float* dx, unsigned* amax_out, cseg_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!check_dims("bn_bwd_apply", B, C, HW)) return 0;
    CSEG_REQUIRE(dy && x && mean_invstd && dx, "bn_bwd_apply: null pointer");
    CSEG_REQUIRE(sums == nullptr || count >= 0.0, "bn_bwd_apply: count must not be negative");
    const BnDims d = bn_dims(B, C, HW);
    const dim3 grid((unsigned)((long)B * C * d.n_ck));
    const bool v = vec_ok(HW, dy, x, dx);
    const double inv = (sums && count > 0.0) ? 1.0 / count : 0.0;      // 0 = the kernel reads the count from row C of `sums`
#define LAUNCH(V, M) hipLaunchKernelGGL((bn_bwd_apply_kernel<V, M>), grid, dim3(256), 0, stream, dy, x, mean_invstd, weight, \
                                        bias, sums, inv, d, dx, amax_out)
    if (v) { if (mask_from_x) LAUNCH(true, true); else LAUNCH(true, false); }
    else { if (mask_from_x) LAUNCH(false, true); else LAUNCH(false, false); }
#undef LAUNCH
    CSEG_CHECK_LAUNCH("bn_bwd_apply");
    return 1;
}

// Single-rank training forward in two launches: statistics partials, then apply with the finalisation folded in.
static int bn_fwd_impl(const float* x, const float* residual, const float* weight, const float* bias, int relu, int B,
                       int C, int HW, float* ws, float eps, float momentum, float* running_mean, float* running_var,
                       int64_t* num_batches_tracked, float* mean_invstd, float* y, unsigned* amax_out, cseg_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!check_dims("bn_fwd", B, C, HW)) return 0;
    CSEG_REQUIRE(x && ws && mean_invstd && y, "bn_fwd: null pointer");
    CSEG_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "bn_fwd: running_mean/var must come together");
    const BnDims d = bn_dims(B, C, HW);
    if (vec_ok(HW, x)) hipLaunchKernelGGL(bn_stats_kernel<true>, dim3(d.S * C), dim3(256), 0, stream, x, d, ws);
    else hipLaunchKernelGGL(bn_stats_kernel<false>, dim3(d.S * C), dim3(256), 0, stream, x, d, ws);
    BnFinalize f;
    f.partial = ws; f.eps = eps; f.momentum = momentum; f.running_mean = running_mean; f.running_var = running_var;
    f.num_batches_tracked = num_batches_tracked; f.mean_invstd = mean_invstd;
    const dim3 grid((unsigned)((long)B * C * d.n_ck));
    const bool v = vec_ok(HW, x, residual, y);
#define LAUNCH(V, R) hipLaunchKernelGGL((bn_apply_fused_kernel<V, R>), grid, dim3(256), 0, stream, x, residual, f, weight, \
                                        bias, d, relu, y, amax_out)
    if (v) { if (residual) LAUNCH(true, true); else LAUNCH(true, false); }
    else { if (residual) LAUNCH(false, true); else LAUNCH(false, false); }
#undef LAUNCH
    CSEG_CHECK_LAUNCH("bn_fwd");
    return 1;
}

// Single-rank backward in two launches (mode as in cseg_bn_bwd_reduce; training = 0: frozen statistics).
static int bn_bwd_impl(const float* dy, const float* x, const float* out, const float* mean_invstd,
                       const float* weight, const float* bias, int mode, int training, int B, int C, int HW, float* ws,
                       float* g_masked, float* d_weight, float* d_bias, float* dx, unsigned* amax_out, cseg_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!check_dims("bn_bwd", B, C, HW)) return 0;
    CSEG_REQUIRE(dy && x && mean_invstd && ws, "bn_bwd: null pointer");
    CSEG_REQUIRE(mode >= 0 && mode <= 2, "bn_bwd: mode %d", mode);
    CSEG_REQUIRE(mode != 2 || (out && g_masked), "bn_bwd: mode 2 needs `out` and `g_masked`");
    const BnDims d = bn_dims(B, C, HW);
    {
        const bool v = vec_ok(HW, dy, x, out, g_masked);
        const dim3 grid(d.S * C);
#define LAUNCH(V, M) hipLaunchKernelGGL((bn_bwd_reduce_kernel<V, M>), grid, dim3(256), 0, stream, dy, x, out, mean_invstd, \
                                        weight, bias, d, g_masked, ws)
        if (v) { if (mode == 0) LAUNCH(true, 0); else if (mode == 1) LAUNCH(true, 1); else LAUNCH(true, 2); }
        else { if (mode == 0) LAUNCH(false, 0); else if (mode == 1) LAUNCH(false, 1); else LAUNCH(false, 2); }
#undef LAUNCH
    }
    if (dx) {
        const float* g = mode == 2 ? g_masked : dy;
        const dim3 grid((unsigned)((long)B * C * d.n_ck));
        const bool v = vec_ok(HW, g, x, dx);
#define LAUNCH(V, M) hipLaunchKernelGGL((bn_bwd_apply_fused_kernel<V, M>), grid, dim3(256), 0, stream, g, x, mean_invstd, \
                                        weight, bias, ws, training, d, d_weight, d_bias, dx, amax_out)
        if (v) { if (mode == 1) LAUNCH(true, true); else LAUNCH(true, false); }
        else { if (mode == 1) LAUNCH(false, true); else LAUNCH(false, false); }
#undef LAUNCH
    } else {
        // no input gradient wanted: only the parameter gradients
        hipLaunchKernelGGL(bn_bwd_sums_kernel, dim3((C + 3) / 4), dim3(256), 0, stream, ws, d, mean_invstd,
                           reinterpret_cast<double*>(ws + (size_t)d.S * C * 2), d_weight, d_bias);
    }
    CSEG_CHECK_LAUNCH("bn_bwd");
    return 1;
}

// ---- the four apply-type entry points, plain and with max|output| accumulated into *amax_out (round 3: the word the f16x3
// convolution that consumes the tensor scales it with; the caller zeroes it, cseg_amax_f32 semantics) ----------------------
extern "C" int cseg_bn_apply(const float* x, const float* residual, const float* mean_invstd, const float* weight,
                             const float* bias, int relu, int B, int C, int HW, float* y, cseg_stream_t stream_) {
    return bn_apply_impl(x, residual, mean_invstd, weight, bias, relu, B, C, HW, y, nullptr, stream_);
}
extern "C" int cseg_bn_apply_amax(const float* x, const float* residual, const float* mean_invstd, const float* weight,
                                  const float* bias, int relu, int B, int C, int HW, float* y, unsigned* amax_out,
                                  cseg_stream_t stream_) {
    return bn_apply_impl(x, residual, mean_invstd, weight, bias, relu, B, C, HW, y, amax_out, stream_);
}
extern "C" int cseg_bn_bwd_apply(const float* dy, const float* x, const float* mean_invstd, const float* weight,
                                 const float* bias, const double* sums, double count, int mask_from_x, int B, int C,
                                 int HW, float* dx, cseg_stream_t stream_) {
    return bn_bwd_apply_impl(dy, x, mean_invstd, weight, bias, sums, count, mask_from_x, B, C, HW, dx, nullptr, stream_);
}
extern "C" int cseg_bn_bwd_apply_amax(const float* dy, const float* x, const float* mean_invstd, const float* weight,
                                      const float* bias, const double* sums, double count, int mask_from_x, int B, int C,
                                      int HW, float* dx, unsigned* amax_out, cseg_stream_t stream_) {
    return bn_bwd_apply_impl(dy, x, mean_invstd, weight, bias, sums, count, mask_from_x, B, C, HW, dx, amax_out, stream_);
}
extern "C" int cseg_bn_fwd(const float* x, const float* residual, const float* weight, const float* bias, int relu, int B,
                           int C, int HW, float* ws, float eps, float momentum, float* running_mean, float* running_var,
                           int64_t* num_batches_tracked, float* mean_invstd, float* y, cseg_stream_t stream_) {
    return bn_fwd_impl(x, residual, weight, bias, relu, B, C, HW, ws, eps, momentum, running_mean, running_var,
                       num_batches_tracked, mean_invstd, y, nullptr, stream_);
}
extern "C" int cseg_bn_fwd_amax(const float* x, const float* residual, const float* weight, const float* bias, int relu, int B,
                                int C, int HW, float* ws, float eps, float momentum, float* running_mean, float* running_var,
                                int64_t* num_batches_tracked, float* mean_invstd, float* y, unsigned* amax_out,
                                cseg_stream_t stream_) {
    return bn_fwd_impl(x, residual, weight, bias, relu, B, C, HW, ws, eps, momentum, running_mean, running_var,
                       num_batches_tracked, mean_invstd, y, amax_out, stream_);
}
extern "C" int cseg_bn_bwd(const float* dy, const float* x, const float* out, const float* mean_invstd,
                           const float* weight, const float* bias, int mode, int training, int B, int C, int HW, float* ws,
                           float* g_masked, float* d_weight, float* d_bias, float* dx, cseg_stream_t stream_) {
    return bn_bwd_impl(dy, x, out, mean_invstd, weight, bias, mode, training, B, C, HW, ws, g_masked, d_weight, d_bias, dx,
                       nullptr, stream_);
}
extern "C" int cseg_bn_bwd_amax(const float* dy, const float* x, const float* out, const float* mean_invstd,
                                const float* weight, const float* bias, int mode, int training, int B, int C, int HW, float* ws,
                                float* g_masked, float* d_weight, float* d_bias, float* dx, unsigned* amax_out,
                                cseg_stream_t stream_) {
    return bn_bwd_impl(dy, x, out, mean_invstd, weight, bias, mode, training, B, C, HW, ws, g_masked, d_weight, d_bias, dx,
                       amax_out, stream_);
}

// ---------------------------------------------------------------------------------------------------------
// Round 6: GROUPED launches -- the BatchNorm passes of several independent layers (the sites of one depth of HRNet's parallel
// branches: reference lib/models/backbones/hrnet/hrnet_backbone.py:262-288 loops over the branches, :49-65 is the block) in ONE
// launch per pass. The grid is the concatenation of the members' one-layer grids (member i owns blocks [block0_i, block0_{i+1})),
// every block runs the one-layer kernel's body on its member with its local block index: bit-identical results, a quarter of the
// launches, and the 5-us kernels of the coarse branches ride along with the fine branch's instead of paying a launch each.
// ---------------------------------------------------------------------------------------------------------
namespace {

struct BnGM {
    const float* x; const float* res; float* y; const float4* st; float* mi; const float* w; const float* b;
    float* rm; float* rv; int64_t* nbt; unsigned* amax;
    const float* dy; const float* out; float* g; float* dw; float* db; float* dx; float* ws;
    BnDims d;
    long T;
    float eps, momentum;
    int block0;
};
struct BnGArgs { BnGM m[CSEG_GROUP_MAX]; int n; };

__device__ __forceinline__ int bn_group_member(const BnGArgs& a, int bid) {
    int mi = 0;
    for (int i = 1; i < a.n; ++i) mi = bid >= a.m[i].block0 ? i : mi;
    return mi;
}

__global__ __launch_bounds__(256) void bn_group_tiles_finalize_kernel(const BnGArgs a) {
    const BnGM& M = a.m[bn_group_member(a, (int)blockIdx.x)];
    bn_tiles_finalize_body(M.st, M.T, M.eps, M.momentum, M.rm, M.rv, M.nbt, M.mi, (int)blockIdx.x - M.block0);
}
template <bool RES>
__global__ __launch_bounds__(256) void bn_group_apply_kernel(const BnGArgs a, int relu) {
    const BnGM& M = a.m[bn_group_member(a, (int)blockIdx.x)];
    bn_apply_body<true, RES>(M.x, M.res, M.mi, M.w, M.b, M.d, relu, M.y, M.amax, (int)blockIdx.x - M.block0);
}
template <int MODE>
__global__ __launch_bounds__(256) void bn_group_bwd_reduce_kernel(const BnGArgs a) {
    const BnGM& M = a.m[bn_group_member(a, (int)blockIdx.x)];
    bn_bwd_reduce_body<true, MODE>(M.dy, M.x, M.out, M.mi, M.w, M.b, M.d, M.g, M.ws, (int)blockIdx.x - M.block0);
}
template <bool MASK, bool FROM_G>
__global__ __launch_bounds__(256) void bn_group_bwd_apply_kernel(const BnGArgs a, int training) {
    const BnGM& M = a.m[bn_group_member(a, (int)blockIdx.x)];
    bn_bwd_apply_fused_body<true, MASK>(FROM_G ? M.g : M.dy, M.x, M.mi, M.w, M.b, M.ws, training, M.d, M.dw, M.db, M.dx, M.amax,
                                        (int)blockIdx.x - M.block0);
}

// SyncBN forms (the statistics cross the ranks between two launches): the members' [C_i + 1, 2] fp64 moments / gradient sums live
// back to back in ONE buffer (member i at row `row0`), which is what the host all-reduces -- one collective per depth and direction,
// no concatenation kernel.
struct BnSyncRows { int row0[CSEG_GROUP_MAX]; };

__global__ __launch_bounds__(256) void bn_group_tiles_moments_kernel(const BnGArgs a, const BnSyncRows r, double* __restrict__ packed) {
    const int mi = bn_group_member(a, (int)blockIdx.x);
    const BnGM& M = a.m[mi];
    const int c = (int)blockIdx.x - M.block0;
    double count, m0, m1;
    tiles_combine(M.st, M.T, c, count, m0, m1);
    if (threadIdx.x == 0) {
        double* mom = packed + 2 * (size_t)r.row0[mi];
        mom[2 * c] = m0;
        mom[2 * c + 1] = m1;
        if (c == 0) { mom[2 * M.d.C] = count; mom[2 * M.d.C + 1] = 0.0; }
    }
}
// (globally summed moments -> mean / invstd / running statistics; the body of bn_finalize_kernel with the exchanged count of row C)
__global__ __launch_bounds__(256) void bn_group_finalize_kernel(const BnGArgs a, const BnSyncRows r, const double* __restrict__ packed) {
    const int mi = bn_group_member(a, (int)blockIdx.x);
    const BnGM& M = a.m[mi];
    const int c = ((int)blockIdx.x - M.block0) * 256 + (int)threadIdx.x;
    if (c == 0 && M.nbt) *M.nbt += 1;
    if (c >= M.d.C) return;
    const double* mom = packed + 2 * (size_t)r.row0[mi];
    finalize_channel(mom[2 * c], mom[2 * c + 1], mom[2 * M.d.C], M.eps, M.momentum, M.rm, M.rv, c, M.mi);
}
// (block partials of bn_group_bwd_reduce_kernel -> fp64 sums + this rank's element count, d_weight / d_bias: bn_bwd_sums_kernel)
__global__ __launch_bounds__(256) void bn_group_bwd_sums_kernel(const BnGArgs a, const BnSyncRows r, double* __restrict__ packed) {
    const int mi = bn_group_member(a, (int)blockIdx.x);
    const BnGM& M = a.m[mi];
    const int lb = (int)blockIdx.x - M.block0;
    const int c = lb * 4 + ((int)threadIdx.x >> 6), lane = threadIdx.x & 63;
    double* sums = packed + 2 * (size_t)r.row0[mi];
    if (lb == 0 && threadIdx.x == 0) {
        sums[2 * M.d.C] = (double)M.d.B * (double)M.d.HW;
        sums[2 * M.d.C + 1] = 0.0;
    }
    if (c >= M.d.C) return;
    double s0 = 0.0, s1 = 0.0;
    for (int s = lane; s < M.d.S; s += 64) {
        s0 += (double)M.ws[((size_t)s * M.d.C + c) * 2 + 0];
        s1 += (double)M.ws[((size_t)s * M.d.C + c) * 2 + 1];
    }
    s0 = wave_sum_d(s0);
    s1 = wave_sum_d(s1);
    if (lane == 0) {
        sums[2 * c] = s0;
        sums[2 * c + 1] = s1;
        if (M.dw) M.dw[c] = (float)(s1 * (double)M.mi[2 * c + 1]);
        if (M.db) M.db[c] = (float)s0;
    }
}

template <bool MASK, bool FROM_G>
__global__ __launch_bounds__(256) void bn_group_bwd_apply_sync_kernel(const BnGArgs a, const BnSyncRows r, const double* __restrict__ packed) {
    const int mi = bn_group_member(a, (int)blockIdx.x);
    const BnGM& M = a.m[mi];
    // inv_count 0: the exchanged element count of row C (bn_bwd_apply_body)
    bn_bwd_apply_body<true, MASK>(FROM_G ? M.g : M.dy, M.x, M.mi, M.w, M.b, packed + 2 * (size_t)r.row0[mi], 0.0, M.d, M.dx, M.amax,
                                  (int)blockIdx.x - M.block0);
}

int bn_group_fill(const char* who, const cseg_bn_group_member* mem, int n, int grid_kind, BnGArgs& a, long& total, bool& all_vec) {
    CSEG_REQUIRE(mem && n >= 1 && n <= CSEG_GROUP_MAX, "%s: needs 1 .. %d members", who, CSEG_GROUP_MAX);
    total = 0;
    all_vec = true;
    for (int i = 0; i < n; ++i) {
        const cseg_bn_group_member& s = mem[i];
        BnGM& m = a.m[i];
        if (grid_kind != 0 && !check_dims(who, s.B, s.C, s.HW)) return 0;
        CSEG_REQUIRE(s.C > 0, "%s: member %d has no channels", who, i);
        m.x = s.x; m.res = s.residual; m.y = s.y; m.st = reinterpret_cast<const float4*>(s.stats); m.mi = s.mean_invstd; m.w = s.weight; m.b = s.bias;
        m.rm = s.running_mean; m.rv = s.running_var; m.nbt = s.num_batches_tracked; m.amax = s.amax_out;
        m.dy = s.dy; m.out = s.out; m.g = s.g_masked; m.dw = s.d_weight; m.db = s.d_bias; m.dx = s.dx; m.ws = s.ws;
        m.d = grid_kind != 0 ? bn_dims(s.B, s.C, s.HW) : BnDims{s.B, s.C, s.HW, 0, 0};
        m.T = s.T; m.eps = s.eps; m.momentum = s.momentum;
        m.block0 = (int)total;
        // 0: one block per channel; 1: one per (image, channel, chunk); 2: splits x channels (the reduction kernels)
        total += grid_kind == 0 ? s.C : grid_kind == 1 ? (long)s.B * s.C * m.d.n_ck : (long)m.d.S * s.C;
        CSEG_REQUIRE(total < 2147483647L, "%s: grid too large", who);
    }
    for (int i = n; i < CSEG_GROUP_MAX; ++i) a.m[i] = a.m[0];
    a.n = n;
    return 1;
}

}  // namespace

// == cseg_bn_tiles_finalize per member (stats, T, eps, momentum, running statistics, num_batches_tracked -> mean_invstd)
extern "C" int cseg_bn_group_tiles_finalize(const cseg_bn_group_member* mem, int n, cseg_stream_t stream_) {
    BnGArgs a;
    long total;
    bool vec;
    if (!bn_group_fill("bn_group_tiles_finalize", mem, n, 0, a, total, vec)) return 0;
    for (int i = 0; i < n; ++i) {
        CSEG_REQUIRE(mem[i].stats && mem[i].mean_invstd && mem[i].T > 0, "bn_group_tiles_finalize: member %d: bad arguments", i);
        CSEG_REQUIRE((reinterpret_cast<uintptr_t>(mem[i].stats) & 15) == 0, "bn_group_tiles_finalize: member %d: the statistics buffer must be 16-byte aligned", i);
        CSEG_REQUIRE((mem[i].running_mean == nullptr) == (mem[i].running_var == nullptr), "bn_group_tiles_finalize: running_mean/var must come together");
    }
    hipLaunchKernelGGL(bn_group_tiles_finalize_kernel, dim3((unsigned)total), dim3(256), 0, (hipStream_t)stream_, a);
    CSEG_CHECK_LAUNCH("bn_group_tiles_finalize");
    return 1;
}

// == cseg_bn_apply_amax per member (x, residual, mean_invstd, weight, bias -> y, amax_out). Members that cannot take the 16-byte path
// (H*W not a multiple of 4, unaligned pointers), or a group that mixes members with and without a residual, run one launch per member.
extern "C" int cseg_bn_group_apply(const cseg_bn_group_member* mem, int n, int relu, cseg_stream_t stream_) {
    BnGArgs a;
    long total;
    bool vec;
    if (!bn_group_fill("bn_group_apply", mem, n, 1, a, total, vec)) return 0;
    bool grouped = true;
    for (int i = 0; i < n; ++i) {
        CSEG_REQUIRE(mem[i].x && mem[i].mean_invstd && mem[i].y, "bn_group_apply: member %d: null pointer", i);
        grouped = grouped && vec_ok(mem[i].HW, mem[i].x, mem[i].residual, mem[i].y) && ((mem[i].residual != nullptr) == (mem[0].residual != nullptr));
    }
    if (!grouped) {
        for (int i = 0; i < n; ++i)
            if (!bn_apply_impl(mem[i].x, mem[i].residual, mem[i].mean_invstd, mem[i].weight, mem[i].bias, relu, mem[i].B, mem[i].C, mem[i].HW,
                               mem[i].y, mem[i].amax_out, stream_))
                return 0;
        return 1;
    }
    if (mem[0].residual) hipLaunchKernelGGL(bn_group_apply_kernel<true>, dim3((unsigned)total), dim3(256), 0, (hipStream_t)stream_, a, relu);
    else hipLaunchKernelGGL(bn_group_apply_kernel<false>, dim3((unsigned)total), dim3(256), 0, (hipStream_t)stream_, a, relu);
    CSEG_CHECK_LAUNCH("bn_group_apply");
    return 1;
}

// == cseg_bn_bwd_amax per member (dy, x, out, mean_invstd, weight, bias, ws -> g_masked, d_weight, d_bias, dx, amax_out), dx wanted
// for every member: two launches for the whole group (reduction, apply).
extern "C" int cseg_bn_group_bwd(const cseg_bn_group_member* mem, int n, int mode, int training, cseg_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    CSEG_REQUIRE(mode >= 0 && mode <= 2, "bn_group_bwd: mode %d", mode);
    BnGArgs a;
    long total;
    bool vec;
    if (!bn_group_fill("bn_group_bwd", mem, n, 2, a, total, vec)) return 0;
    bool grouped = true;
    for (int i = 0; i < n; ++i) {
        CSEG_REQUIRE(mem[i].dy && mem[i].x && mem[i].mean_invstd && mem[i].ws, "bn_group_bwd: member %d: null pointer", i);
        CSEG_REQUIRE(mode != 2 || (mem[i].out && mem[i].g_masked), "bn_group_bwd: mode 2 needs `out` and `g_masked`");
        grouped = grouped && mem[i].dx && vec_ok(mem[i].HW, mem[i].dy, mem[i].x, mem[i].out, mem[i].g_masked) && vec_ok(mem[i].HW, mem[i].dx);
    }
    if (!grouped) {
        for (int i = 0; i < n; ++i)
            if (!bn_bwd_impl(mem[i].dy, mem[i].x, mem[i].out, mem[i].mean_invstd, mem[i].weight, mem[i].bias, mode, training, mem[i].B, mem[i].C,
                             mem[i].HW, mem[i].ws, mem[i].g_masked, mem[i].d_weight, mem[i].d_bias, mem[i].dx, mem[i].amax_out, stream_))
                return 0;
        return 1;
    }
    if (mode == 0) hipLaunchKernelGGL(bn_group_bwd_reduce_kernel<0>, dim3((unsigned)total), dim3(256), 0, stream, a);
    else if (mode == 1) hipLaunchKernelGGL(bn_group_bwd_reduce_kernel<1>, dim3((unsigned)total), dim3(256), 0, stream, a);
    else hipLaunchKernelGGL(bn_group_bwd_reduce_kernel<2>, dim3((unsigned)total), dim3(256), 0, stream, a);
    if (!bn_group_fill("bn_group_bwd", mem, n, 1, a, total, vec)) return 0;
    if (mode == 1) hipLaunchKernelGGL((bn_group_bwd_apply_kernel<true, false>), dim3((unsigned)total), dim3(256), 0, stream, a, training);
    else if (mode == 2) hipLaunchKernelGGL((bn_group_bwd_apply_kernel<false, true>), dim3((unsigned)total), dim3(256), 0, stream, a, training);
    else hipLaunchKernelGGL((bn_group_bwd_apply_kernel<false, false>), dim3((unsigned)total), dim3(256), 0, stream, a, training);
    CSEG_CHECK_LAUNCH("bn_group_bwd");
    return 1;
}

// ---- SyncBN forms of the grouped passes (cseg_hip.h): the members' fp64 [C_i + 1, 2] rows back to back in `packed`
namespace {
int bn_sync_rows(const cseg_bn_group_member* mem, int n, BnSyncRows& r) {
    int row = 0;
    for (int i = 0; i < CSEG_GROUP_MAX; ++i) {
        r.row0[i] = row;
        if (i < n) row += mem[i].C + 1;
    }
    return row;
}
}  // namespace

// == cseg_bn_tiles_moments per member, written at row sum_{j<i} (C_j + 1) of packed [sum (C_i + 1), 2] f64
extern "C" int cseg_bn_group_tiles_moments(const cseg_bn_group_member* mem, int n, double* packed, cseg_stream_t stream_) {
    BnGArgs a;
    long total;
    bool vec;
    if (!bn_group_fill("bn_group_tiles_moments", mem, n, 0, a, total, vec)) return 0;
    CSEG_REQUIRE(packed, "bn_group_tiles_moments: null output");
    for (int i = 0; i < n; ++i)
        CSEG_REQUIRE(mem[i].stats && mem[i].T > 0 && (reinterpret_cast<uintptr_t>(mem[i].stats) & 15) == 0, "bn_group_tiles_moments: member %d: bad statistics buffer", i);
    BnSyncRows r;
    bn_sync_rows(mem, n, r);
    hipLaunchKernelGGL(bn_group_tiles_moments_kernel, dim3((unsigned)total), dim3(256), 0, (hipStream_t)stream_, a, r, packed);
    CSEG_CHECK_LAUNCH("bn_group_tiles_moments");
    return 1;
}

// == cseg_bn_finalize(count 0) per member from its rows of the (all-reduced) packed moments
extern "C" int cseg_bn_group_finalize(const cseg_bn_group_member* mem, int n, const double* packed, cseg_stream_t stream_) {
    CSEG_REQUIRE(mem && packed && n >= 1 && n <= CSEG_GROUP_MAX, "bn_group_finalize: bad arguments");
    BnGArgs a;
    long total = 0;
    for (int i = 0; i < n; ++i) {
        const cseg_bn_group_member& s = mem[i];
        CSEG_REQUIRE(s.C > 0 && s.mean_invstd, "bn_group_finalize: member %d: bad arguments", i);
        CSEG_REQUIRE((s.running_mean == nullptr) == (s.running_var == nullptr), "bn_group_finalize: running_mean/var must come together");
        BnGM& m = a.m[i];
        m = BnGM();
        m.mi = s.mean_invstd; m.rm = s.running_mean; m.rv = s.running_var; m.nbt = s.num_batches_tracked;
        m.d = BnDims{s.B, s.C, s.HW, 0, 0};
        m.eps = s.eps; m.momentum = s.momentum;
        m.block0 = (int)total;
        total += (s.C + 255) / 256;
    }
    for (int i = n; i < CSEG_GROUP_MAX; ++i) a.m[i] = a.m[0];
    a.n = n;
    BnSyncRows r;
    bn_sync_rows(mem, n, r);
    hipLaunchKernelGGL(bn_group_finalize_kernel, dim3((unsigned)total), dim3(256), 0, (hipStream_t)stream_, a, r, packed);
    CSEG_CHECK_LAUNCH("bn_group_finalize");
    return 1;
}

// == cseg_bn_bwd_reduce per member: g_masked (mode 2), d_weight, d_bias, and the fp64 sums [C_i + 1, 2] into the member's rows of packed
extern "C" int cseg_bn_group_bwd_reduce(const cseg_bn_group_member* mem, int n, int mode, double* packed, cseg_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    CSEG_REQUIRE(mode >= 0 && mode <= 2 && packed, "bn_group_bwd_reduce: mode %d / null output", mode);
    BnGArgs a;
    long total;
    bool vec;
    if (!bn_group_fill("bn_group_bwd_reduce", mem, n, 2, a, total, vec)) return 0;
    BnSyncRows r;
    bn_sync_rows(mem, n, r);
    bool grouped = true;
    for (int i = 0; i < n; ++i) {
        CSEG_REQUIRE(mem[i].dy && mem[i].x && mem[i].mean_invstd && mem[i].ws, "bn_group_bwd_reduce: member %d: null pointer", i);
        CSEG_REQUIRE(mode != 2 || (mem[i].out && mem[i].g_masked), "bn_group_bwd_reduce: mode 2 needs `out` and `g_masked`");
        grouped = grouped && vec_ok(mem[i].HW, mem[i].dy, mem[i].x, mem[i].out, mem[i].g_masked);
    }
    if (!grouped) {
        for (int i = 0; i < n; ++i)
            if (!cseg_bn_bwd_reduce(mem[i].dy, mem[i].x, mem[i].out, mem[i].mean_invstd, mem[i].weight, mem[i].bias, mode, mem[i].B, mem[i].C, mem[i].HW,
                                    mem[i].ws, mem[i].g_masked, packed + 2 * (size_t)r.row0[i], mem[i].d_weight, mem[i].d_bias, stream_))
                return 0;
        return 1;
    }
    if (mode == 0) hipLaunchKernelGGL(bn_group_bwd_reduce_kernel<0>, dim3((unsigned)total), dim3(256), 0, stream, a);
    else if (mode == 1) hipLaunchKernelGGL(bn_group_bwd_reduce_kernel<1>, dim3((unsigned)total), dim3(256), 0, stream, a);
    else hipLaunchKernelGGL(bn_group_bwd_reduce_kernel<2>, dim3((unsigned)total), dim3(256), 0, stream, a);
    long blocks = 0;
    for (int i = 0; i < n; ++i) { a.m[i].block0 = (int)blocks; blocks += (mem[i].C + 3) / 4; }
    hipLaunchKernelGGL(bn_group_bwd_sums_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, a, r, packed);
    CSEG_CHECK_LAUNCH("bn_group_bwd_reduce");
    return 1;
}

// == cseg_bn_bwd_apply_amax(count 0) per member from its rows of the (all-reduced) packed sums: dx, max|dx|
extern "C" int cseg_bn_group_bwd_apply(const cseg_bn_group_member* mem, int n, int mode, const double* packed, cseg_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    CSEG_REQUIRE(mode >= 0 && mode <= 2 && packed, "bn_group_bwd_apply: mode %d / null sums", mode);
    BnGArgs a;
    long total;
    bool vec;
    if (!bn_group_fill("bn_group_bwd_apply", mem, n, 1, a, total, vec)) return 0;
    BnSyncRows r;
    bn_sync_rows(mem, n, r);
    bool grouped = true;
    for (int i = 0; i < n; ++i) {
        CSEG_REQUIRE(mem[i].x && mem[i].mean_invstd && mem[i].dx && (mode == 2 ? mem[i].g_masked != nullptr : mem[i].dy != nullptr),
                     "bn_group_bwd_apply: member %d: null pointer", i);
        grouped = grouped && vec_ok(mem[i].HW, mode == 2 ? mem[i].g_masked : mem[i].dy, mem[i].x, mem[i].dx);
    }
    if (!grouped) {
        for (int i = 0; i < n; ++i)
            if (!bn_bwd_apply_impl(mode == 2 ? mem[i].g_masked : mem[i].dy, mem[i].x, mem[i].mean_invstd, mem[i].weight, mem[i].bias,
                                   packed + 2 * (size_t)r.row0[i], 0.0, mode == 1, mem[i].B, mem[i].C, mem[i].HW, mem[i].dx, mem[i].amax_out, stream_))
                return 0;
        return 1;
    }
    if (mode == 1) hipLaunchKernelGGL((bn_group_bwd_apply_sync_kernel<true, false>), dim3((unsigned)total), dim3(256), 0, stream, a, r, packed);
    else if (mode == 2) hipLaunchKernelGGL((bn_group_bwd_apply_sync_kernel<false, true>), dim3((unsigned)total), dim3(256), 0, stream, a, r, packed);
    else hipLaunchKernelGGL((bn_group_bwd_apply_sync_kernel<false, false>), dim3((unsigned)total), dim3(256), 0, stream, a, r, packed);
    CSEG_CHECK_LAUNCH("bn_group_bwd_apply");
    return 1;
}
