// Exact adjoint of bilinear(align_corners=True) upsampling, shared by the HRNet head (upcat.hip) and the HRNet
// exchange unit (fuse.hip). Deterministic (no atomics); taps are re-derived with the forward's own fp32 index
// arithmetic so the result is the exact transpose of the forward operator.
//
// Band kernel (default): a block owns TY coarse rows of one (image, channel) plane. It streams the band of fine rows
// that touch those coarse rows into LDS once (coalesced, optionally multiplied by the ReLU mask), applies the
// horizontal adjoint (fine columns -> coarse columns) out of LDS, then the vertical adjoint. Every fine gradient
// element is read from HBM once per band (halo rows of neighbouring bands are the only re-reads).
// Gather kernel (fallback when a band does not fit in LDS): one thread per coarse element walks its footprint.
#pragma once
#include "cseg_common.h"

__device__ __forceinline__ void bl_tap(float s, int n_in, int o, int& i0, int& i1, float& l1) {
    const float f = s * (float)o;
    i0 = (int)f;
    i1 = i0 + (i0 < n_in - 1 ? 1 : 0);
    l1 = f - (float)i0;
}

// first / last fine index whose lower tap i0 lies in [c_lo - 1, c_hi]  (c_lo..c_hi coarse indices, inclusive)
__device__ __forceinline__ void bl_fine_range(float s, int n_fine, int c_lo, int c_hi, int& lo, int& hi) {
    lo = 0; hi = n_fine - 1;
    if (s > 0.f) {
        lo = max(0, (int)ceilf((float)(c_lo - 1) / s) - 1);
        while (lo < n_fine - 1 && (int)(s * (float)lo) < c_lo - 1) ++lo;
        hi = min(n_fine - 1, (int)floorf((float)(c_hi + 1) / s) + 1);
        while (hi > 0 && (int)(s * (float)hi) > c_hi) --hi;
    }
}

constexpr int BL_MAX_TAPS = 40;   // fine columns touching one coarse column (2/scale + 1) -- up to 16x upsampling
constexpr int BL_TY = 4;          // coarse rows per block (larger bands measured slower: fewer, LDS-heavier blocks)

// grid = (ceil(hs / TY), Cm, B), TY coarse rows per block; dynamic LDS = (nY_max * w0 + nY_max * ws) floats
template <bool MASKED, int TAPS>
__global__ __launch_bounds__(256) void bilinear_adjoint_band_kernel(const float* __restrict__ d_out, int Ctot, int coff,
                                                                    int Cm, int hs, int ws, int h0, int w0, int nY_max,
                                                                    int TY, const float* __restrict__ act,
                                                                    float* __restrict__ dx) {
    extern __shared__ __attribute__((aligned(16))) float bl_smem[];
    float* band = bl_smem;                 // [nY][w0]
    float* hbuf = bl_smem + nY_max * w0;   // [nY][ws]
    __shared__ int rng[2];
    const int c = blockIdx.y, b = blockIdx.z, tid = threadIdx.x;
    const int ys0 = blockIdx.x * TY;
    const int ys1 = min(hs - 1, ys0 + TY - 1);
    const float sy = ac_scale(hs, h0), sx = ac_scale(ws, w0);
    if (tid == 0) {
        int lo, hi;
        bl_fine_range(sy, h0, ys0, ys1, lo, hi);
        rng[0] = lo; rng[1] = min(nY_max, max(0, hi - lo + 1));
    }
    __syncthreads();
    const int Y_lo = rng[0], nY = rng[1];
    const size_t plane = ((size_t)b * Ctot + coff + c) * h0 * w0;
    const float* g = d_out + plane + (size_t)Y_lo * w0;
    const float* a = MASKED ? act + plane + (size_t)Y_lo * w0 : nullptr;
    const int n = nY * w0;
    if ((w0 & 3) == 0) {
        for (int e = tid * 4; e < n; e += 1024) {
            float4 v = *reinterpret_cast<const float4*>(g + e);
            if (MASKED) {
                const float4 m = *reinterpret_cast<const float4*>(a + e);
                v.x = m.x > 0.f ? v.x : 0.f; v.y = m.y > 0.f ? v.y : 0.f;
                v.z = m.z > 0.f ? v.z : 0.f; v.w = m.w > 0.f ? v.w : 0.f;
            }
            *reinterpret_cast<float4*>(band + e) = v;
        }
    } else {
        for (int e = tid; e < n; e += 256) {
            float v = g[e];
            if (MASKED) v = a[e] > 0.f ? v : 0.f;
            band[e] = v;
        }
    }
    __syncthreads();
    // horizontal adjoint: thread = (coarse column xs, row group); taps of xs cached in registers
    const int groups = max(1, 256 / ws);
    const int xs = tid % ws, grp = tid / ws;
    if (grp < groups) {
        int x_lo, x_hi;
        bl_fine_range(sx, w0, xs, xs, x_lo, x_hi);
        float wx[TAPS];
        const int nt = min(TAPS, x_hi - x_lo + 1);
#pragma unroll
        for (int t = 0; t < TAPS; ++t) {
            float wv = 0.f;
            if (t < nt) {
                int x0, x1; float l1;
                bl_tap(sx, ws, x_lo + t, x0, x1, l1);
                if (x0 == xs) wv += 1.f - l1;
                if (x1 == xs) wv += l1;
            }
            wx[t] = wv;
        }
        for (int fy = grp; fy < nY; fy += groups) {
            const float* row = band + fy * w0 + x_lo;
            float acc = 0.f;
#pragma unroll
            for (int t = 0; t < TAPS; ++t)
                if (t < nt) acc += wx[t] * row[t];
            hbuf[fy * ws + xs] = acc;
        }
    }
    __syncthreads();
    // vertical adjoint
    for (int e = tid; e < (ys1 - ys0 + 1) * ws; e += 256) {
        const int yl = e / ws, x = e - yl * ws;
        const int ys = ys0 + yl;
        float acc = 0.f;
        for (int fy = 0; fy < nY; ++fy) {
            int y0, y1; float l1;
            bl_tap(sy, hs, Y_lo + fy, y0, y1, l1);
            float wy = 0.f;
            if (y0 == ys) wy += 1.f - l1;
            if (y1 == ys) wy += l1;
            if (wy != 0.f) acc += wy * hbuf[fy * ws + x];
        }
        dx[(((size_t)b * Cm + c) * hs + ys) * ws + x] = acc;
    }
}

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
    int y_lo, y_hi, x_lo, x_hi;
    bl_fine_range(sy, h0, ys, ys, y_lo, y_hi);
    bl_fine_range(sx, w0, xs, xs, x_lo, x_hi);
    const size_t plane = ((size_t)b * Ctot + coff + c) * h0 * w0;
    const float* g = d_out + plane;
    const float* a = MASKED ? act + plane : nullptr;
    float acc = 0.f;
    for (int y = y_lo; y <= y_hi; ++y) {
        int y0, y1; float ly1;
        bl_tap(sy, hs, y, y0, y1, ly1);
        float wy = 0.f;
        if (y0 == ys) wy += 1.f - ly1;
        if (y1 == ys) wy += ly1;
        if (wy == 0.f) continue;
        float racc = 0.f;
        for (int x = x_lo; x <= x_hi; ++x) {
            int x0, x1; float lx1;
            bl_tap(sx, ws, x, x0, x1, lx1);
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

// Host-side dispatch: band kernel (tap count templated by the upsampling factor), gather when a band cannot fit.
template <bool MASKED, int TAPS>
static inline void launch_band(const float* d_out, int Ctot, int coff, int Cm, int hs, int ws, int h0, int w0, int B,
                               int nY_max, size_t lds, const float* act, float* dx, hipStream_t stream) {
    dim3 grid((hs + BL_TY - 1) / BL_TY, Cm, B);
    hipLaunchKernelGGL((bilinear_adjoint_band_kernel<MASKED, TAPS>), grid, dim3(256), lds, stream, d_out, Ctot, coff, Cm,
                       hs, ws, h0, w0, nY_max, BL_TY, act, dx);
}

template <bool MASKED>
static inline int launch_bilinear_adjoint(const float* d_out, int Ctot, int coff, int Cm, int hs, int ws, int h0, int w0,
                                          int B, const float* act, float* dx, hipStream_t stream) {
    const float sy = ac_scale(hs, h0), sx = ac_scale(ws, w0);
    const int taps = sx > 0.f ? (int)(2.0f / sx) + 3 : w0;
    const int nY_max = sy > 0.f ? (int)((float)(BL_TY + 1) / sy) + 3 : h0;
    const size_t lds = sizeof(float) * ((size_t)nY_max * w0 + (size_t)nY_max * ws);
    if (ws <= 256 && taps <= BL_MAX_TAPS && lds <= 64 * 1024) {
        if (taps <= 8) launch_band<MASKED, 8>(d_out, Ctot, coff, Cm, hs, ws, h0, w0, B, nY_max, lds, act, dx, stream);
        else if (taps <= 12) launch_band<MASKED, 12>(d_out, Ctot, coff, Cm, hs, ws, h0, w0, B, nY_max, lds, act, dx, stream);
        else if (taps <= 24) launch_band<MASKED, 24>(d_out, Ctot, coff, Cm, hs, ws, h0, w0, B, nY_max, lds, act, dx, stream);
        else launch_band<MASKED, BL_MAX_TAPS>(d_out, Ctot, coff, Cm, hs, ws, h0, w0, B, nY_max, lds, act, dx, stream);
    } else {
        dim3 grid((hs * ws + 255) / 256, Cm, B);
        hipLaunchKernelGGL((bilinear_adjoint_gather_kernel<MASKED>), grid, dim3(256), 0, stream, d_out, Ctot, coff, Cm,
                           hs, ws, h0, w0, act, dx);
    }
    return 1;
}
