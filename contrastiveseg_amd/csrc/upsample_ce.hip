// Segmentation term: bilinear(align_corners=True) upsample of the logits to label resolution fused with the
// weighted cross entropy (ignore index, mean over the sum of weights).
// Reference: lib/loss/loss_contrast.py:180-181 + lib/loss/loss_helper.py:169-206. The reference materialises the
// [B,K,H,W] logits (319 MB at bs8) and makes ~3 passes over them forward plus the same again backward; here the
// coarse logits tile is staged in LDS once per block and the upsampled tensor never exists.
// HBM-bound (algorithmic bytes: seg + target forward; seg + target + d_seg backward); no MFMA.
// Both directions are atomics-free, so the loss and the gradient are run-to-run deterministic.
#include "cseg_common.h"

namespace {

constexpr int FT_H = 8, FT_W = 32;  // forward: hi-res tile per block (one pixel per thread)
constexpr int BT = 8;               // backward: low-res tile edge per block

struct CeDims {
    int B, K, h, w, H, W;
    float sy, sx;
    int ignore_label;
};

__device__ __forceinline__ void tap(float s, int n_in, int o, int& i0, int& i1, float& l1) {
    const float f = s * (float)o;
    i0 = (int)f;
    i1 = i0 + (i0 < n_in - 1 ? 1 : 0);
    l1 = f - (float)i0;
}


// log-sum-exp over the K bilinearly interpolated logits of one fine pixel (taps o00..o11 into the staged tile).
// K <= 32: the values are kept in registers, one max pass + one exp per class (no divergent rescale);
// larger K: online (running max) form. *vt receives the logit of class t (pass t = -1 to skip).
__device__ __forceinline__ float pixel_lse(const float* __restrict__ tile, int plane, int K, int o00, int o01, int o10,
                                           int o11, float ly0, float ly1, float lx0, float lx1, int t, float* vt) {
    if (K <= 32) {
        float v[32];
        float m = -INFINITY;
#pragma unroll
        for (int k = 0; k < 32; ++k) {
            float x = -INFINITY;
            if (k < K) {
                const float* pl = tile + k * plane;
                x = ly0 * (lx0 * pl[o00] + lx1 * pl[o01]) + ly1 * (lx0 * pl[o10] + lx1 * pl[o11]);
                if (k == t) *vt = x;
            }
            v[k] = x;
            m = fmaxf(m, x);
        }
        float se = 0.f;
#pragma unroll
        for (int k = 0; k < 32; ++k) se += (k < K) ? expf(v[k] - m) : 0.f;
        return m + logf(se);
    }
    float m = -INFINITY, se = 0.f;
    for (int k = 0; k < K; ++k) {
        const float* pl = tile + k * plane;
        const float x = ly0 * (lx0 * pl[o00] + lx1 * pl[o01]) + ly1 * (lx0 * pl[o10] + lx1 * pl[o11]);
        if (k == t) *vt = x;
        if (x > m) { se = se * expf(m - x) + 1.f; m = x; }
        else se += expf(x - m);
    }
    return m + logf(se);
}

// ---------------------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ce_fwd_kernel(const float* __restrict__ seg, const int64_t* __restrict__ target,
                                                     const float* __restrict__ weight, CeDims d, int rows_max,
                                                     int cols_max, float* __restrict__ partial,
                                                     int32_t* __restrict__ status) {
    extern __shared__ __attribute__((aligned(16))) float smem[];  // [K][rows_max][cols_max]
    __shared__ float red[2][4];
    const int tiles_x = (d.W + FT_W - 1) / FT_W, tiles_y = (d.H + FT_H - 1) / FT_H;
    int blk = blockIdx.x;
    const int tx = blk % tiles_x; blk /= tiles_x;
    const int ty = blk % tiles_y;
    const int b = blk / tiles_y;
    const int Y0 = ty * FT_H, X0 = tx * FT_W;
    const int Yl = min(d.H - 1, Y0 + FT_H - 1), Xl = min(d.W - 1, X0 + FT_W - 1);
    const int ry0 = (int)(d.sy * (float)Y0), rx0 = (int)(d.sx * (float)X0);
    const int ry1 = min(d.h - 1, (int)(d.sy * (float)Yl) + 1), rx1 = min(d.w - 1, (int)(d.sx * (float)Xl) + 1);
    const int nr = ry1 - ry0 + 1, nc = rx1 - rx0 + 1;
    const int plane = rows_max * cols_max;
    for (int e = threadIdx.x; e < d.K * nr * nc; e += 256) {
        const int k = e / (nr * nc), rem = e - k * nr * nc;
        const int r = rem / nc, c = rem - r * nc;
        smem[k * plane + r * cols_max + c] = seg[(((size_t)b * d.K + k) * d.h + ry0 + r) * d.w + rx0 + c];
    }
    __syncthreads();
    const int Y = Y0 + (threadIdx.x >> 5), X = X0 + (threadIdx.x & 31);
    float wnll = 0.f, wsum = 0.f;
    if (Y < d.H && X < d.W) {
        const int64_t t64 = target[((size_t)b * d.H + Y) * d.W + X];
        if (t64 != (int64_t)d.ignore_label) {
            if (t64 < 0 || t64 >= d.K) {
                atomicAdd(&status[1], 1);
            } else {
                const int t = (int)t64;
                int y0, y1, x0, x1;
                float ly1, lx1;
                tap(d.sy, d.h, Y, y0, y1, ly1);
                tap(d.sx, d.w, X, x0, x1, lx1);
                const float ly0 = 1.f - ly1, lx0 = 1.f - lx1;
                const int o00 = (y0 - ry0) * cols_max + (x0 - rx0), o01 = (y0 - ry0) * cols_max + (x1 - rx0);
                const int o10 = (y1 - ry0) * cols_max + (x0 - rx0), o11 = (y1 - ry0) * cols_max + (x1 - rx0);
                float vt = 0.f;
                const float lse = pixel_lse(smem, plane, d.K, o00, o01, o10, o11, ly0, ly1, lx0, lx1, t, &vt);
                const float wt = weight ? weight[t] : 1.f;
                wnll = wt * (lse - vt);
                wsum = wt;
            }
        }
    }
    wnll = wave_sum(wnll);
    wsum = wave_sum(wsum);
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = wnll; red[1][threadIdx.x >> 6] = wsum; }
    __syncthreads();
    if (threadIdx.x == 0) {
        partial[2 * (size_t)blockIdx.x + 0] = red[0][0] + red[0][1] + red[0][2] + red[0][3];
        partial[2 * (size_t)blockIdx.x + 1] = red[1][0] + red[1][1] + red[1][2] + red[1][3];
    }
}

__global__ __launch_bounds__(1024) void ce_finish_kernel(const float* __restrict__ partial, int n_blocks,
                                                         float* __restrict__ out) {
    __shared__ double red[2][16];
    double a = 0.0, w = 0.0;
    for (int i = threadIdx.x; i < n_blocks; i += 1024) { a += partial[2 * (size_t)i]; w += partial[2 * (size_t)i + 1]; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o, 64); w += __shfl_xor(w, o, 64); }
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = a; red[1][threadIdx.x >> 6] = w; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double sa = 0.0, sw = 0.0;
        for (int i = 0; i < 16; ++i) { sa += red[0][i]; sw += red[1][i]; }
        out[0] = (float)(sa / sw);
        out[1] = (float)sw;
    }
}

// ---------------------------------------------------------------------------------------------------------
// backward: block owns a BT x BT low-res tile of d_seg for one image. It rebuilds the softmax statistics of every
// hi-res pixel in the tile's footprint once (phase 1), then per class k: (A) d_k = coef * (softmax_k - onehot_k)
// for every footprint pixel, (B) horizontal adjoint of the bilinear taps, (C) vertical adjoint -> d_seg[k] tile.
// The separable form keeps all 64 lanes busy and evaluates every exp once; no atomics.
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ce_bwd_kernel(const float* __restrict__ seg, const int64_t* __restrict__ target,
                                                     const float* __restrict__ weight, CeDims d, int nY_max,
                                                     int nX_max, const float* __restrict__ out,
                                                     const float* __restrict__ d_loss, float* __restrict__ d_seg) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int SR = BT + 2;  // staged seg rows/cols (tile + 1 halo each side)
    const int plane = SR * SR;
    const int nF = nY_max * nX_max;
    float* seg_t = smem;                          // [K][SR][SR]
    float* f_lse = seg_t + d.K * plane;           // [nF]
    float* f_coef = f_lse + nF;                   // [nF]
    int* f_tgt = (int*)(f_coef + nF);             // [nF]
    float* dk = (float*)(f_tgt + nF);             // [nF]   d_k of the current class
    float* hrow = dk + nF;                        // [nY_max][BT] horizontally reduced
    int* yi0 = (int*)(hrow + nY_max * BT);        // [nY_max] y0 of each footprint row
    float* yl1 = (float*)(yi0 + nY_max);
    int* xi0 = (int*)(yl1 + nY_max);
    float* xl1 = (float*)(xi0 + nX_max);
    __shared__ int rng[4];         // Y_lo, nY, X_lo, nX
    __shared__ int lo_hi[4][BT];   // per tile row: fy_lo, fy_hi ; per tile col: fx_lo, fx_hi (inclusive)

    const int tiles_x = (d.w + BT - 1) / BT, tiles_y = (d.h + BT - 1) / BT;
    int blk = blockIdx.x;
    const int tx = blk % tiles_x; blk /= tiles_x;
    const int ty = blk % tiles_y;
    const int b = blk / tiles_y;
    const int ys0 = ty * BT, xs0 = tx * BT;
    const int sr0 = ys0 - 1, sc0 = xs0 - 1;  // staged region origin (may be -1)
    const int tid = threadIdx.x;

    if (tid == 0) {
        // hi-res rows whose y0 lies in [ys0-1, ys0+BT-1] (their y0 or y1 tap can hit the tile)
        int lo = 0, hi = d.H - 1;
        if (d.sy > 0.f) {
            lo = max(0, (int)ceilf((float)(ys0 - 1) / d.sy) - 1);
            while (lo < d.H - 1 && (int)(d.sy * (float)lo) < ys0 - 1) ++lo;
            hi = min(d.H - 1, (int)floorf((float)(ys0 + BT) / d.sy) + 1);
            while (hi > 0 && (int)(d.sy * (float)hi) > ys0 + BT - 1) --hi;
        }
        rng[0] = lo; rng[1] = max(0, hi - lo + 1);
        lo = 0; hi = d.W - 1;
        if (d.sx > 0.f) {
            lo = max(0, (int)ceilf((float)(xs0 - 1) / d.sx) - 1);
            while (lo < d.W - 1 && (int)(d.sx * (float)lo) < xs0 - 1) ++lo;
            hi = min(d.W - 1, (int)floorf((float)(xs0 + BT) / d.sx) + 1);
            while (hi > 0 && (int)(d.sx * (float)hi) > xs0 + BT - 1) --hi;
        }
        rng[2] = lo; rng[3] = max(0, hi - lo + 1);
    }
    for (int e = tid; e < d.K * plane; e += 256) {
        const int k = e / plane, rem = e - k * plane;
        const int r = sr0 + rem / SR, c = sc0 + rem % SR;
        float v = 0.f;
        if (r >= 0 && r < d.h && c >= 0 && c < d.w) v = seg[(((size_t)b * d.K + k) * d.h + r) * d.w + c];
        seg_t[e] = v;
    }
    __syncthreads();
    const int Y_lo = rng[0], nY = min(rng[1], nY_max), X_lo = rng[2], nX = min(rng[3], nX_max);
    for (int e = tid; e < nY; e += 256) {
        int i0, i1; float l1;
        tap(d.sy, d.h, Y_lo + e, i0, i1, l1);
        yi0[e] = i0; yl1[e] = l1;
    }
    for (int e = tid; e < nX; e += 256) {
        int i0, i1; float l1;
        tap(d.sx, d.w, X_lo + e, i0, i1, l1);
        xi0[e] = i0; xl1[e] = l1;
    }
    __syncthreads();
    // footprint index ranges per tile row / column (y0 in {s-1, s})
    if (tid < 2 * BT) {
        const bool is_x = tid >= BT;
        const int s = (is_x ? xs0 : ys0) + (tid & (BT - 1));
        const int n = is_x ? nX : nY;
        const int* i0 = is_x ? xi0 : yi0;
        int lo = n, hi = -1;
        for (int f = 0; f < n; ++f) {
            if (i0[f] >= s - 1 && i0[f] <= s) { lo = min(lo, f); hi = f; }
        }
        lo_hi[is_x ? 2 : 0][tid & (BT - 1)] = lo;
        lo_hi[is_x ? 3 : 1][tid & (BT - 1)] = hi;
    }
    const float gscale = d_loss[0] / out[1];
    // phase 1: softmax statistics of every footprint pixel
    for (int e = tid; e < nY * nX; e += 256) {
        const int fy = e / nX, fx = e - fy * nX;
        const int Y = Y_lo + fy, X = X_lo + fx;
        const int64_t t64 = target[((size_t)b * d.H + Y) * d.W + X];
        float lse = 0.f, coef = 0.f;
        int t = -1;
        if (t64 != (int64_t)d.ignore_label && t64 >= 0 && t64 < d.K) {
            t = (int)t64;
            const int y0 = yi0[fy], x0 = xi0[fx];
            const int y1 = y0 + (y0 < d.h - 1 ? 1 : 0), x1 = x0 + (x0 < d.w - 1 ? 1 : 0);
            const float ly1 = yl1[fy], lx1 = xl1[fx], ly0 = 1.f - ly1, lx0 = 1.f - lx1;
            const int o00 = (y0 - sr0) * SR + (x0 - sc0), o01 = (y0 - sr0) * SR + (x1 - sc0);
            const int o10 = (y1 - sr0) * SR + (x0 - sc0), o11 = (y1 - sr0) * SR + (x1 - sc0);
            float unused;
            lse = pixel_lse(seg_t, plane, d.K, o00, o01, o10, o11, ly0, ly1, lx0, lx1, -1, &unused);
            coef = (weight ? weight[t] : 1.f) * gscale;
        }
        f_lse[e] = lse; f_coef[e] = coef; f_tgt[e] = t;
    }
    __syncthreads();
    for (int k = 0; k < d.K; ++k) {
        const float* pl = seg_t + k * plane;
        // (A) d_k at every footprint pixel
        for (int e = tid; e < nY * nX; e += 256) {
            const float coef = f_coef[e];
            float dv = 0.f;
            if (coef != 0.f) {
                const int fy = e / nX, fx = e - fy * nX;
                const int y0 = yi0[fy], x0 = xi0[fx];
                const int y1 = y0 + (y0 < d.h - 1 ? 1 : 0), x1 = x0 + (x0 < d.w - 1 ? 1 : 0);
                const float ly1 = yl1[fy], lx1 = xl1[fx], ly0 = 1.f - ly1, lx0 = 1.f - lx1;
                const int o00 = (y0 - sr0) * SR + (x0 - sc0), o01 = (y0 - sr0) * SR + (x1 - sc0);
                const int o10 = (y1 - sr0) * SR + (x0 - sc0), o11 = (y1 - sr0) * SR + (x1 - sc0);
                const float v = ly0 * (lx0 * pl[o00] + lx1 * pl[o01]) + ly1 * (lx0 * pl[o10] + lx1 * pl[o11]);
                dv = coef * (expf(v - f_lse[e]) - (f_tgt[e] == k ? 1.f : 0.f));
            }
            dk[e] = dv;
        }
        __syncthreads();
        // (B) horizontal adjoint: hrow[fy][xl] = sum_fx wx(fx, xs) * d_k[fy][fx]
        for (int e = tid; e < nY * BT; e += 256) {
            const int fy = e / BT, xl = e - fy * BT;
            const int xs = xs0 + xl;
            float acc = 0.f;
            for (int fx = lo_hi[2][xl]; fx <= lo_hi[3][xl]; ++fx) {
                const int x0 = xi0[fx];
                const int x1 = x0 + (x0 < d.w - 1 ? 1 : 0);
                const float lx1 = xl1[fx];
                float wx = 0.f;
                if (x0 == xs) wx += 1.f - lx1;
                if (x1 == xs) wx += lx1;
                acc += wx * dk[fy * nX + fx];
            }
            hrow[e] = acc;
        }
        __syncthreads();
        // (C) vertical adjoint -> output tile of class k
        if (tid < BT * BT) {
            const int yl = tid / BT, xl = tid - yl * BT;
            const int ys = ys0 + yl, xs = xs0 + xl;
            if (ys < d.h && xs < d.w) {
                float acc = 0.f;
                for (int fy = lo_hi[0][yl]; fy <= lo_hi[1][yl]; ++fy) {
                    const int y0 = yi0[fy];
                    const int y1 = y0 + (y0 < d.h - 1 ? 1 : 0);
                    const float ly1 = yl1[fy];
                    float wy = 0.f;
                    if (y0 == ys) wy += 1.f - ly1;
                    if (y1 == ys) wy += ly1;
                    acc += wy * hrow[fy * BT + xl];
                }
                d_seg[(((size_t)b * d.K + k) * d.h + ys) * d.w + xs] = acc;
            }
        }
        // hrow is rewritten only after the next (A)+barrier; dk only after this (B)'s barrier: no extra sync needed
    }
}

int make_dims(CeDims* d, int B, int K, int h, int w, int H, int W, int ignore_label) {
    CSEG_REQUIRE(B > 0 && K > 0 && h > 0 && w > 0 && H > 0 && W > 0, "upsample_ce: empty shape");
    CSEG_REQUIRE(H >= h && W >= w, "upsample_ce: only upsampling is supported (%dx%d -> %dx%d)", h, w, H, W);
    d->B = B; d->K = K; d->h = h; d->w = w; d->H = H; d->W = W;
    d->sy = ac_scale(h, H); d->sx = ac_scale(w, W);
    d->ignore_label = ignore_label;
    return 1;
}

}  // namespace

extern "C" int cseg_upsample_ce_blocks(int B, int H, int W) {
    return B * ((H + FT_H - 1) / FT_H) * ((W + FT_W - 1) / FT_W);
}

extern "C" int cseg_upsample_ce_fwd(const float* seg, const int64_t* target, const float* weight, int ignore_label,
                                    int B, int K, int h, int w, int H, int W, float* partial, float* out,
                                    int32_t* status, cseg_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    CeDims d;
    if (!make_dims(&d, B, K, h, w, H, W, ignore_label)) return 0;
    const int rows_max = (int)(d.sy * (float)(FT_H - 1)) + 3, cols_max = (int)(d.sx * (float)(FT_W - 1)) + 3;
    const size_t lds = sizeof(float) * (size_t)K * rows_max * cols_max;
    CSEG_REQUIRE(lds <= 64 * 1024, "upsample_ce_fwd: K=%d needs %zu B of LDS per block", K, lds);
    const int n_blocks = cseg_upsample_ce_blocks(B, H, W);
    hipLaunchKernelGGL(ce_fwd_kernel, dim3(n_blocks), dim3(256), lds, stream, seg, target, weight, d, rows_max,
                       cols_max, partial, status);
    CSEG_CHECK_LAUNCH("ce_fwd_kernel");
    hipLaunchKernelGGL(ce_finish_kernel, dim3(1), dim3(1024), 0, stream, partial, n_blocks, out);
    CSEG_CHECK_LAUNCH("ce_finish_kernel");
    return 1;
}

extern "C" int cseg_upsample_ce_bwd(const float* seg, const int64_t* target, const float* weight, int ignore_label,
                                    int B, int K, int h, int w, int H, int W, const float* out, const float* d_loss,
                                    float* d_seg, cseg_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    CeDims d;
    if (!make_dims(&d, B, K, h, w, H, W, ignore_label)) return 0;
    // footprint of BT low-res rows: hi-res rows with y0 in [ys0-1, ys0+BT-1]  ->  at most (BT+1)/sy + 2 rows
    const int nY_max = d.sy > 0.f ? (int)((float)(BT + 1) / d.sy) + 3 : H;
    const int nX_max = d.sx > 0.f ? (int)((float)(BT + 1) / d.sx) + 3 : W;
    const size_t lds = sizeof(float) * ((size_t)K * (BT + 2) * (BT + 2) + 4 * (size_t)nY_max * nX_max +
                                        (size_t)nY_max * BT + 2 * (size_t)nY_max + 2 * (size_t)nX_max);
    CSEG_REQUIRE(lds <= 150 * 1024, "upsample_ce_bwd: K=%d scale %dx needs %zu B of LDS per block", K, H / h, lds);
    if (lds > 64 * 1024) {
        if (hipFuncSetAttribute((const void*)ce_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) !=
            hipSuccess) {
            cseg_set_error("upsample_ce_bwd: cannot raise dynamic LDS to %zu", lds);
            return 0;
        }
    }
    const int n_blocks = B * ((h + BT - 1) / BT) * ((w + BT - 1) / BT);
    hipLaunchKernelGGL(ce_bwd_kernel, dim3(n_blocks), dim3(256), lds, stream, seg, target, weight, d, nY_max, nX_max,
                       out, d_loss, d_seg);
    CSEG_CHECK_LAUNCH("ce_bwd_kernel");
    return 1;
}
