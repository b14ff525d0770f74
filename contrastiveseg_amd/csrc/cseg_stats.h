// BatchNorm statistics from the epilogue of the convolution that produces the tensor (first half).
// Reference shape of the work: conv -> bn -> relu chains of lib/models/backbones/hrnet/hrnet_backbone.py:49-65 and
// lib/models/tools/module_helper.py:35-39; torch (and round 3 here) re-reads the whole convolution output once more just to sum it.
//
// Every split-operand forward kernel ends with the same accumulator layout: lane (g = lane / 16, n = lane % 16) holds, per
// 16-channel tile nt, the 16 output values of channel co0 + 16 nt + n at pixels first + 16 mt + 4 g + r (mt, r = 0..3): one
// wave = one 64-pixel SEGMENT of the output (a row piece of a 3x3 tile, a run of 64 flat pixels of a 1x1 convolution) x its
// channel tiles. cseg_stats_emit() reduces each channel over the wave's segment -- in registers and four shuffles, while the
// values are still there -- and lane g == 0 writes one float4 (count, mean, M2 = sum (v - mean)^2) per (channel, segment) to
//        stats[(channel * T + segment)]              T = segments per channel (cseg_conv_stat_segments)
// Mean-centred per segment (<= 64 values), so nothing cancels whatever |mean| / std is; the per-channel combination (Chan et al.)
// runs in fp64 in bn.hip (cseg_bn_tiles_finalize / cseg_bn_tiles_moments). Fixed order everywhere: run-to-run deterministic.
#pragma once
#include "cseg_split.h"

// FAST (round 4, default): a segment that lies fully inside the row -- all but the ragged last one -- skips the per-value masks
// (16 selects + 16 counted adds per pass): the same sums in the same order (up to the multiply-adds the compiler may fuse in the
// unmasked form). `store` = false: timing experiments.
template <int NTW, int NTMAX, bool FAST = true>
__device__ __forceinline__ void cseg_stats_emit(const f32x4 (&acc)[4][NTMAX], const float* __restrict__ bias, int co0, float unscale,
                                                long first, long limit, int g, int n, float4* __restrict__ st, size_t T,
                                                bool store = true) {
    if (FAST && first + 64 <= limit) {
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) {
            const float bv = bias ? bias[co0 + nt * 16 + n] : 0.f;
            f32x4 a[4];
            float s = 0.f;
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                a[mt] = acc[mt][nt] * unscale + bv;
#pragma unroll
                for (int r = 0; r < 4; ++r) s += a[mt][r];
            }
            s += __shfl_xor(s, 16, 64);
            s += __shfl_xor(s, 32, 64);
            const float mean = s / 64.f;
            float m2 = 0.f;
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float d = a[mt][r] - mean;
                    m2 += d * d;
                }
            m2 += __shfl_xor(m2, 16, 64);
            m2 += __shfl_xor(m2, 32, 64);
            if (g == 0 && store) st[(size_t)(nt * 16 + n) * T] = make_float4(64.f, mean, m2, 0.f);
        }
        return;
    }
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) {
        const float bv = bias ? bias[co0 + nt * 16 + n] : 0.f;
        float v[16];
        float s = 0.f, cnt = 0.f;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            const f32x4 a = acc[mt][nt] * unscale + bv;           // exactly the value the store wrote
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const bool ok = first + 16 * mt + 4 * g + r < limit;
                v[4 * mt + r] = a[r];
                s += ok ? a[r] : 0.f;
                cnt += ok ? 1.f : 0.f;
            }
        }
        s += __shfl_xor(s, 16, 64);
        s += __shfl_xor(s, 32, 64);
        cnt += __shfl_xor(cnt, 16, 64);
        cnt += __shfl_xor(cnt, 32, 64);
        const float mean = cnt > 0.f ? s / cnt : 0.f;
        float m2 = 0.f;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const bool ok = first + 16 * mt + 4 * g + r < limit;
                const float d = v[4 * mt + r] - mean;
                m2 += ok ? d * d : 0.f;
            }
        m2 += __shfl_xor(m2, 16, 64);
        m2 += __shfl_xor(m2, 32, 64);
        if (g == 0 && store) st[(size_t)(nt * 16 + n) * T] = make_float4(cnt, mean, m2, 0.f);
    }
}
