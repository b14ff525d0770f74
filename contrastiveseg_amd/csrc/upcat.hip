// HRNet head input: bilinear(align_corners=True) upsample of the coarse branches fused with the channel concat.
// Reference: lib/models/nets/hrnet.py:86-91 (3x F.interpolate + torch.cat -> 705 MB of temporaries at bs8).
// HBM-bound: forward writes each output element once with 16-byte stores, coarse maps are re-read from L2;
// backward is the exact adjoint written as a gather (no atomics => deterministic).
#include "cseg_common.h"
#include "cseg_bilinear.h"
#include "cseg_split.h"

namespace {

struct UpcatMaps {
    const float* x[4];
    int C[4], h[4], w[4], coff[5];
    int n;
};

// grid = (column tiles x row tiles, Ctot, B): the source map is block-uniform. A block covers 16 rows x 256 columns:
// thread = (column quad, row group) and walks 4 consecutive rows, so the horizontal taps (float->int conversions, edge
// clamps, weights) are computed once per thread and reused for every row; the output is written with 16-byte
// non-temporal stores (755 MB at bs8: it does not fit any cache and is read back only by the head convolution).
constexpr int UP_ROWS = 4;
typedef float v4f __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void store_nt4(float* p, float a, float b, float c, float d) {
    __builtin_nontemporal_store((v4f){a, b, c, d}, reinterpret_cast<v4f*>(p));
}

// `amax` (may be null): max|out| record for the split-operand convolutions that read the result (cseg_amax_f32's format) -- accumulated
// while the values are in registers instead of a second pass over the 755 MB tensor.
__global__ __launch_bounds__(256) void upcat_fwd_kernel(UpcatMaps m, int Ctot, int tiles_x, float* __restrict__ out,
                                                        unsigned* __restrict__ amax) {
    const int c = blockIdx.y, b = blockIdx.z;
    const int h0 = m.h[0], w0 = m.w[0];
    const int tx = blockIdx.x % tiles_x, ty = blockIdx.x / tiles_x;
    const int x = (tx * 64 + (threadIdx.x & 63)) * 4;
    const int yb = ty * (4 * UP_ROWS) + (threadIdx.x >> 6) * UP_ROWS;
    const bool live = x < w0 && yb < h0;
    float vmax = 0.f;
    if (live) {
        int mi = 0;
#pragma unroll
        for (int k = 1; k < 4; ++k) if (k < m.n && c >= m.coff[k]) mi = k;
        const int cm = c - m.coff[mi];
        float* orow = out + (((size_t)b * Ctot + c) * h0 + yb) * w0 + x;
        if (mi == 0) {
            const float* irow = m.x[0] + (((size_t)b * m.C[0] + cm) * h0 + yb) * w0 + x;
#pragma unroll
            for (int r = 0; r < UP_ROWS; ++r) {
                if (yb + r < h0) {
                    const float4 v = *reinterpret_cast<const float4*>(irow + (size_t)r * w0);
                    store_nt4(orow + (size_t)r * w0, v.x, v.y, v.z, v.w);
                    vmax = fmaxf(fmaxf(vmax, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
                }
            }
        } else {
            const int hs = m.h[mi], ws = m.w[mi];
            const float sy = ac_scale(hs, h0), sx = ac_scale(ws, w0);
            int x0[4], x1[4];
            float lx1[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const float fx = sx * (float)(x + t);
                x0[t] = (int)fx;
                x1[t] = x0[t] + (x0[t] < ws - 1 ? 1 : 0);
                lx1[t] = fx - (float)x0[t];
            }
            const float* plane = m.x[mi] + ((size_t)b * m.C[mi] + cm) * hs * ws;
#pragma unroll
            for (int r = 0; r < UP_ROWS; ++r) {
                const int y = yb + r;
                if (y < h0) {
                    const float fy = sy * (float)y;
                    const int y0 = (int)fy;
                    const int y1 = y0 + (y0 < hs - 1 ? 1 : 0);
                    const float ly1 = fy - (float)y0, ly0 = 1.f - ly1;
                    const float* r0 = plane + (size_t)y0 * ws;
                    const float* r1 = plane + (size_t)y1 * ws;
                    float o[4];
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const float lx0 = 1.f - lx1[t];
                        o[t] = ly0 * (lx0 * r0[x0[t]] + lx1[t] * r0[x1[t]]) + ly1 * (lx0 * r1[x0[t]] + lx1[t] * r1[x1[t]]);
                    }
                    store_nt4(orow + (size_t)r * w0, o[0], o[1], o[2], o[3]);
                    vmax = fmaxf(fmaxf(vmax, fmaxf(fabsf(o[0]), fabsf(o[1]))), fmaxf(fabsf(o[2]), fabsf(o[3])));
                }
            }
        }
    }
    if (amax) {                                    // (block-uniform) one candidate per wave; |v| >= 0, so the bit patterns order like the values
        unsigned bits = __builtin_bit_cast(unsigned, vmax);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) bits = max(bits, (unsigned)__shfl_xor((int)bits, o, 64));
        amax_publish_waves(bits, amax);
    }
}

// scalar fallback when w0 % 4 != 0
__global__ __launch_bounds__(256) void upcat_fwd_scalar_kernel(UpcatMaps m, int Ctot, float* __restrict__ out) {
    const int c = blockIdx.y, b = blockIdx.z;
    const int h0 = m.h[0], w0 = m.w[0];
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= h0 * w0) return;
    const int y = e / w0, x = e - y * w0;
    int mi = 0;
#pragma unroll
    for (int k = 1; k < 4; ++k) if (k < m.n && c >= m.coff[k]) mi = k;
    const int cm = c - m.coff[mi];
    float v;
    if (mi == 0) {
        v = m.x[0][(((size_t)b * m.C[0] + cm) * h0 + y) * w0 + x];
    } else {
        const int hs = m.h[mi], ws = m.w[mi];
        const float fy = ac_scale(hs, h0) * (float)y, fx = ac_scale(ws, w0) * (float)x;
        const int y0 = (int)fy, x0 = (int)fx;
        const int y1 = y0 + (y0 < hs - 1 ? 1 : 0), x1 = x0 + (x0 < ws - 1 ? 1 : 0);
        const float ly1 = fy - (float)y0, ly0 = 1.f - ly1, lx1 = fx - (float)x0, lx0 = 1.f - lx1;
        const float* p = m.x[mi] + ((size_t)b * m.C[mi] + cm) * hs * ws;
        v = ly0 * (lx0 * p[y0 * ws + x0] + lx1 * p[y0 * ws + x1]) + ly1 * (lx0 * p[y1 * ws + x0] + lx1 * p[y1 * ws + x1]);
    }
    out[(((size_t)b * Ctot + c) * h0 + y) * w0 + x] = v;
}

// backward for the pass-through map: strided channel-slice copy
__global__ __launch_bounds__(256) void upcat_bwd_copy_kernel(const float* __restrict__ d_out, int Ctot, int C0, int hw,
                                                             float* __restrict__ dx0) {
    const int c = blockIdx.y, b = blockIdx.z;
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e < hw) dx0[((size_t)b * C0 + c) * hw + e] = d_out[((size_t)b * Ctot + c) * hw + e];
}

int fill_maps(UpcatMaps* m, const int* C, const int* hs, const int* ws, int n_maps) {
    CSEG_REQUIRE(n_maps >= 1 && n_maps <= 4, "upcat: n_maps=%d not in [1,4]", n_maps);
    m->n = n_maps;
    m->coff[0] = 0;
    for (int i = 0; i < 4; ++i) {
        m->x[i] = nullptr;
        m->C[i] = i < n_maps ? C[i] : 0;
        m->h[i] = i < n_maps ? hs[i] : 1;
        m->w[i] = i < n_maps ? ws[i] : 1;
        m->coff[i + 1] = m->coff[i] + m->C[i];
        if (i < n_maps) CSEG_REQUIRE(C[i] > 0 && hs[i] > 0 && ws[i] > 0, "upcat: empty map %d", i);
    }
    return 1;
}

}  // namespace

static int upcat_fwd_impl(const float* const* xs, const int* C, const int* hs, const int* ws, int n_maps, int B, float* out, unsigned* amax,
                          cseg_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    UpcatMaps m;
    if (!fill_maps(&m, C, hs, ws, n_maps)) return 0;
    for (int i = 0; i < n_maps; ++i) m.x[i] = xs[i];
    const int Ctot = m.coff[n_maps], h0 = hs[0], w0 = ws[0];
    CSEG_REQUIRE(Ctot <= 65535 && B <= 65535, "upcat: grid too large");
    if (w0 % 4 == 0) {
        const int tiles_x = (w0 / 4 + 63) / 64, tiles_y = (h0 + 4 * UP_ROWS - 1) / (4 * UP_ROWS);
        dim3 grid(tiles_x * tiles_y, Ctot, B);
        hipLaunchKernelGGL(upcat_fwd_kernel, grid, dim3(256), 0, stream, m, Ctot, tiles_x, out, amax);
    } else {
        CSEG_REQUIRE(!amax, "upcat_fwd: the max|out| record needs a width that is a multiple of 4 (got %d)", w0);
        dim3 grid((h0 * w0 + 255) / 256, Ctot, B);
        hipLaunchKernelGGL(upcat_fwd_scalar_kernel, grid, dim3(256), 0, stream, m, Ctot, out);
    }
    CSEG_CHECK_LAUNCH("upcat_fwd_kernel");
    return 1;
}

extern "C" int cseg_upcat_fwd(const float* const* xs, const int* C, const int* hs, const int* ws, int n_maps, int B, float* out,
                              cseg_stream_t stream) {
    return upcat_fwd_impl(xs, C, hs, ws, n_maps, B, out, nullptr, stream);
}

// the same with max|out| accumulated into `amax` (a zeroed record of CSEG_AMAX_WORDS words, as cseg_amax_f32 fills it); width % 4 == 0
extern "C" int cseg_upcat_fwd_amax(const float* const* xs, const int* C, const int* hs, const int* ws, int n_maps, int B, float* out,
                                   unsigned* amax, cseg_stream_t stream) {
    CSEG_REQUIRE(amax, "upcat_fwd_amax: null record");
    return upcat_fwd_impl(xs, C, hs, ws, n_maps, B, out, amax, stream);
}

extern "C" int cseg_upcat_bwd(const float* d_out, const int* C, const int* hs, const int* ws, int n_maps, int B,
                              float* const* d_xs, cseg_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    UpcatMaps m;
    if (!fill_maps(&m, C, hs, ws, n_maps)) return 0;
    const int Ctot = m.coff[n_maps], h0 = hs[0], w0 = ws[0];
    CSEG_REQUIRE(Ctot <= 65535 && B <= 65535, "upcat: grid too large");
    if (d_xs[0]) {
        dim3 grid((h0 * w0 + 255) / 256, C[0], B);
        hipLaunchKernelGGL(upcat_bwd_copy_kernel, grid, dim3(256), 0, stream, d_out, Ctot, C[0], h0 * w0, d_xs[0]);
        CSEG_CHECK_LAUNCH("upcat_bwd_copy_kernel");
    }
    for (int i = 1; i < n_maps; ++i) {
        if (!d_xs[i]) continue;
        launch_bilinear_adjoint<false>(d_out, Ctot, m.coff[i], C[i], hs[i], ws[i], h0, w0, B, nullptr, d_xs[i], stream);
        CSEG_CHECK_LAUNCH("bilinear_adjoint (upcat)");
    }
    return 1;
}
