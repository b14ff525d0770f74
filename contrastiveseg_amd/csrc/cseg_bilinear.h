// Exact adjoint of bilinear(align_corners=True) upsampling, written as a gather (no atomics => deterministic):
// one thread per coarse element sums its footprint of the fine gradient, re-deriving taps with the forward's own
// fp32 index arithmetic. Shared by the HRNet head (upcat.hip) and the HRNet exchange unit (fuse.hip).
#pragma once
#include "cseg_common.h"

// grid = (ceil(hs*ws/256), Cm, B). d_out has Ctot channels, this map's channels start at coff.
// MASKED: multiply the fine gradient by (act > 0) (ReLU backward fused into the gather), act has d_out's layout.
template <bool MASKED>
__global__ __launch_bounds__(256) void bilinear_adjoint_gather_kernel(const float* __restrict__ d_out, int Ctot, int coff,
                                                                      int Cm, int hs, int ws, int h0, int w0,
                                                                      const float* __restrict__ act,
                                                                      float* __restrict__ dx) {
    const int c = blockIdx.y, b = blockIdx.z;
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= hs * ws) return;
    const int ys = e / ws, xs = e - ys * ws;
    const float sy = ac_scale(hs, h0), sx = ac_scale(ws, w0);
    int y_lo = 0, y_hi = h0 - 1, x_lo = 0, x_hi = w0 - 1;
    if (sy > 0.f) {
        y_lo = max(0, (int)ceilf((float)(ys - 1) / sy) - 1);
        y_hi = min(h0 - 1, (int)floorf((float)(ys + 1) / sy) + 1);
    }
    if (sx > 0.f) {
        x_lo = max(0, (int)ceilf((float)(xs - 1) / sx) - 1);
        x_hi = min(w0 - 1, (int)floorf((float)(xs + 1) / sx) + 1);
    }
    const size_t plane = ((size_t)b * Ctot + coff + c) * h0 * w0;
    const float* g = d_out + plane;
    const float* a = MASKED ? act + plane : nullptr;
    float acc = 0.f;
    for (int y = y_lo; y <= y_hi; ++y) {
        const float fy = sy * (float)y;
        const int y0 = (int)fy;
        const int y1 = y0 + (y0 < hs - 1 ? 1 : 0);
        const float ly1 = fy - (float)y0;
        float wy = 0.f;
        if (y0 == ys) wy += 1.f - ly1;
        if (y1 == ys) wy += ly1;
        if (wy == 0.f) continue;
        float racc = 0.f;
        for (int x = x_lo; x <= x_hi; ++x) {
            const float fx = sx * (float)x;
            const int x0 = (int)fx;
            const int x1 = x0 + (x0 < ws - 1 ? 1 : 0);
            const float lx1 = fx - (float)x0;
            float wx = 0.f;
            if (x0 == xs) wx += 1.f - lx1;
            if (x1 == xs) wx += lx1;
            if (wx != 0.f) {
                float gv = g[(size_t)y * w0 + x];
                if (MASKED) gv = a[(size_t)y * w0 + x] > 0.f ? gv : 0.f;
                racc += wx * gv;
            }
        }
        acc += wy * racc;
    }
    dx[(((size_t)b * Cm + c) * hs + ys) * ws + xs] = acc;
}
