// Segmentation term: bilinear(align_corners=True) upsample of the logits to label resolution fused with the
// weighted cross entropy (ignore index, mean over the sum of weights).
// Reference: lib/loss/loss_contrast.py:180-181 + lib/loss/loss_helper.py:169-206. The reference materialises the
// [B,K,H,W] logits (319 MB at bs8) and makes ~3 passes over them forward plus the same again backward; here the
// upsampled tensor never exists.
//
// Work decomposition (round 2; replaces the LDS-tile kernels whose backward ran a serial loop over the K classes with
// three barriers each): a CELL = the fine pixels of one label row that share the left coarse tap x0 = c (4-5 pixels at
// the 4.02x upsampling of the HRNet head, 1 at label resolution, <= 17). One lane owns one coarse column c, so the
// four coarse logits a cell needs per class are loaded with unit stride across the wave straight from L1/L2 -- no LDS
// tile, no bank conflicts -- the vertical blend is done once per class and every fine pixel costs two FMAs.
//   forward   one thread per (label row, cell): max pass + sum-exp pass over the classes, log-sum-exp written to a
//             [B,H,W] buffer for the backward, fixed-order block partials of (weighted nll, weight).
//   backward  a block = (image, band of R coarse rows, group of KG classes); lane = coarse column. The block marches
//             down the label rows of its band; per row each lane evaluates g_k = coef * (softmax_k - onehot_k) for
//             its cell, splits it into the part that lands on column c and the part for column c+1 (handed to the
//             right neighbour with one DPP shift; one LDS word per wave boundary), and accumulates the two coarse rows
//             the label row touches in registers (a sliding pair: the march is monotone). Every exp is evaluated once
//             per (pixel, class), all lanes work on all classes of the group, no atomics: run-to-run deterministic.
// HBM-bound (algorithmic bytes: seg + target [+ lse] forward; seg + target + lse + d_seg backward); no MFMA.
#include "cseg_common.h"

namespace {

constexpr int BAND = 4;        // coarse rows per backward block
constexpr int MAX_PX = 17;     // fine pixels per cell (upsampling factors up to 16x)

struct CeDims {
    int B, K, h, w, H, W;
    float sy, sx;
    int ignore_label;
};

__host__ __device__ __forceinline__ int tap0(float s, int o) { return (int)(s * (float)o); }

__device__ __forceinline__ void tap(float s, int n_in, int o, int& i0, int& i1, float& l1) {
    const float f = s * (float)o;
    i0 = (int)f;
    i1 = i0 + (i0 < n_in - 1 ? 1 : 0);
    l1 = f - (float)i0;
}

// smallest o in [0, n_out] with tap0(s, o) >= c (n_out when there is none); tap0 is monotone in o
__host__ __device__ __forceinline__ int first_with_tap(float s, int n_out, int c) {
    if (c <= 0) return 0;
    if (!(s > 0.f)) return n_out;
    int o = (int)((float)c / s);
    if (o > n_out) o = n_out;
    if (o < 0) o = 0;
    while (o > 0 && tap0(s, o - 1) >= c) --o;
    while (o < n_out && tap0(s, o) < c) ++o;
    return o;
}

// ---------------------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------------------
template <int PX>
__global__ __launch_bounds__(256) void ce_fwd_kernel(const float* __restrict__ seg, const int64_t* __restrict__ target,
                                                     const float* __restrict__ weight, CeDims d, float* __restrict__ lse_out,
                                                     float* __restrict__ partial, int32_t* __restrict__ status) {
    __shared__ float red[2][4];
    const long g = (long)blockIdx.x * 256 + threadIdx.x;
    const long n_cells = (long)d.B * d.H * d.w;
    float wnll = 0.f, wsum = 0.f;
    if (g < n_cells) {
        const int c = (int)(g % d.w);
        const long row = g / d.w;
        const int Y = (int)(row % d.H), b = (int)(row / d.H);
        const int Xs = first_with_tap(d.sx, d.W, c), Xe = first_with_tap(d.sx, d.W, c + 1);
        const int n = min(Xe - Xs, PX);
        if (n > 0) {
            int y0, y1;
            float ly1;
            tap(d.sy, d.h, Y, y0, y1, ly1);
            const float ly0 = 1.f - ly1;
            const int c1 = c + (c < d.w - 1 ? 1 : 0);
            float lx1[PX], m[PX], se[PX], vt[PX];
            int t[PX];
            const int64_t* trow = target + ((size_t)b * d.H + Y) * d.W + Xs;
#pragma unroll
            for (int p = 0; p < PX; ++p) {
                lx1[p] = 0.f; m[p] = -INFINITY; se[p] = 0.f; vt[p] = 0.f; t[p] = -1;
                if (p < n) {
                    lx1[p] = d.sx * (float)(Xs + p) - (float)c;
                    const int64_t t64 = trow[p];
                    if (t64 != (int64_t)d.ignore_label) {
                        if (t64 < 0 || t64 >= d.K) atomicAdd(&status[1], 1);
                        else t[p] = (int)t64;
                    }
                }
            }
            const float* s0 = seg + (((size_t)b * d.K) * d.h + y0) * d.w;
            const float* s1 = seg + (((size_t)b * d.K) * d.h + y1) * d.w;
            const size_t plane = (size_t)d.h * d.w;
            for (int k = 0; k < d.K; ++k) {                       // pass 1: running max
                const float r0 = ly0 * s0[k * plane + c] + ly1 * s1[k * plane + c];
                const float r1 = ly0 * s0[k * plane + c1] + ly1 * s1[k * plane + c1];
#pragma unroll
                for (int p = 0; p < PX; ++p) m[p] = fmaxf(m[p], fmaf(lx1[p], r1 - r0, r0));
            }
            for (int k = 0; k < d.K; ++k) {                       // pass 2: sum of exponentials, target logit
                const float r0 = ly0 * s0[k * plane + c] + ly1 * s1[k * plane + c];
                const float r1 = ly0 * s0[k * plane + c1] + ly1 * s1[k * plane + c1];
#pragma unroll
                for (int p = 0; p < PX; ++p) {
                    const float v = fmaf(lx1[p], r1 - r0, r0);
                    se[p] += __expf(v - m[p]);
                    vt[p] = (k == t[p]) ? v : vt[p];
                }
            }
            float* lrow = lse_out ? lse_out + ((size_t)b * d.H + Y) * d.W + Xs : nullptr;
#pragma unroll
            for (int p = 0; p < PX; ++p) {
                if (p < n) {
                    const float lse = m[p] + __logf(se[p]);
                    if (lrow) lrow[p] = lse;
                    if (t[p] >= 0) {
                        const float wt = weight ? weight[t[p]] : 1.f;
                        wnll += wt * (lse - vt[p]);
                        wsum += wt;
                    }
                }
            }
        }
    }
    wnll = wave_sum(wnll);
    wsum = wave_sum(wsum);
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = wnll; red[1][threadIdx.x >> 6] = wsum; }
    __syncthreads();
    if (threadIdx.x == 0) {
        partial[2 * (size_t)blockIdx.x + 0] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
        partial[2 * (size_t)blockIdx.x + 1] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
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
// backward
// ---------------------------------------------------------------------------------------------------------
template <int PX, int KG>
__global__ __launch_bounds__(256) void ce_bwd_kernel(const float* __restrict__ seg, const int64_t* __restrict__ target,
                                                     const float* __restrict__ weight, const float* __restrict__ lse,
                                                     CeDims d, int n_groups, int n_bands, int halo,
                                                     const float* __restrict__ out, const float* __restrict__ d_loss,
                                                     float* __restrict__ d_seg) {
    __shared__ float xchg[2][4][KG];          // [row parity][wave][class]: hB of the wave's last lane
    int blk = blockIdx.x;
    const int grp = blk % n_groups; blk /= n_groups;
    const int band = blk % n_bands; blk /= n_bands;
    const int n_cb = (d.w + (256 - halo) - 1) / (256 - halo);
    const int cb = blk % n_cb;
    const int b = blk / n_cb;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = cb * (256 - halo) + tid - halo;                 // this lane's cell / output column
    const bool cell_ok = c >= 0 && c < d.w;
    const bool writes = cell_ok && tid >= halo;                   // halo lane only feeds its right neighbour
    const int k0 = grp * KG;
    const int ys0 = band * BAND;
    const int ys_end = min(ys0 + BAND, d.h);                      // rows [ys0, ys_end) are written by this block
    const float gscale = d_loss[0] / out[1];

    const int Xs = cell_ok ? first_with_tap(d.sx, d.W, c) : 0;
    const int n = cell_ok ? min(first_with_tap(d.sx, d.W, c + 1) - Xs, PX) : 0;
    const int c1 = c + (c < d.w - 1 ? 1 : 0);
    float lx1[PX];
#pragma unroll
    for (int p = 0; p < PX; ++p) lx1[p] = p < n ? d.sx * (float)(Xs + p) - (float)c : 0.f;

    // label rows whose upper tap lies in [ys0 - 1, ys_end - 1]
    const int Y_lo = first_with_tap(d.sy, d.H, ys0 - 1), Y_hi = first_with_tap(d.sy, d.H, ys_end);
    float acc_lo[KG], acc_hi[KG];            // coarse rows y_cur and y_cur + 1
#pragma unroll
    for (int j = 0; j < KG; ++j) { acc_lo[j] = 0.f; acc_hi[j] = 0.f; }
    int y_cur = max(ys0 - 1, 0);
    if (Y_lo < Y_hi) y_cur = tap0(d.sy, Y_lo);

    auto flush_lo = [&](int row) {
        if (writes && row >= ys0 && row < ys_end) {
#pragma unroll
            for (int j = 0; j < KG; ++j)
                if (k0 + j < d.K) d_seg[(((size_t)b * d.K + k0 + j) * d.h + row) * d.w + c] = acc_lo[j];
        }
    };

    for (int Y = Y_lo; Y < Y_hi; ++Y) {
        int y0, y1;
        float ly1;
        tap(d.sy, d.h, Y, y0, y1, ly1);
        const float ly0 = 1.f - ly1;
        while (y_cur < y0) {                 // the march is monotone: retire the finished coarse row
            flush_lo(y_cur);
#pragma unroll
            for (int j = 0; j < KG; ++j) { acc_lo[j] = acc_hi[j]; acc_hi[j] = 0.f; }
            ++y_cur;
        }
        float coef[PX], ls[PX];
        int t[PX];
#pragma unroll
        for (int p = 0; p < PX; ++p) {
            coef[p] = 0.f; ls[p] = INFINITY; t[p] = -1;      // exp(v - inf) = 0: inactive pixels contribute exactly 0
            if (p < n) {
                const size_t o = ((size_t)b * d.H + Y) * d.W + Xs + p;
                const int64_t t64 = target[o];
                if (t64 != (int64_t)d.ignore_label && t64 >= 0 && t64 < d.K) {
                    t[p] = (int)t64;
                    coef[p] = (weight ? weight[t[p]] : 1.f) * gscale;
                    ls[p] = lse[o];
                }
            }
        }
        float hA[KG], hB[KG];
#pragma unroll
        for (int j = 0; j < KG; ++j) {
            hA[j] = 0.f; hB[j] = 0.f;
            const int k = k0 + j;
            if (k < d.K && n > 0) {
                const float* s0 = seg + (((size_t)b * d.K + k) * d.h + y0) * d.w;
                const float* s1 = seg + (((size_t)b * d.K + k) * d.h + y1) * d.w;
                const float r0 = ly0 * s0[c] + ly1 * s1[c];
                const float r1 = ly0 * s0[c1] + ly1 * s1[c1];
#pragma unroll
                for (int p = 0; p < PX; ++p) {
                    const float v = fmaf(lx1[p], r1 - r0, r0);
                    const float gk = coef[p] * (__expf(v - ls[p]) - (t[p] == k ? 1.f : 0.f));   // coef = 0: nothing
                    const float gb = lx1[p] * gk;
                    hB[j] += gb;                   // lands on column c + 1 (x1 tap)
                    hA[j] += gk - gb;              // lands on column c     (x0 tap, weight 1 - lx1)
                }
            }
        }
        // at the right image edge x1 == x0: both taps land on column c
        if (cell_ok && c == d.w - 1) {
#pragma unroll
            for (int j = 0; j < KG; ++j) { hA[j] += hB[j]; hB[j] = 0.f; }
        }
        const int par = Y & 1;
        if (lane == 63) {
#pragma unroll
            for (int j = 0; j < KG; ++j) xchg[par][wave][j] = hB[j];
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < KG; ++j) {
            float left = __shfl_up(hB[j], 1, 64);
            if (lane == 0) left = wave > 0 ? xchg[par][wave - 1][j] : 0.f;
            const float H = hA[j] + left;
            if (y1 == y0) acc_lo[j] += H;          // bottom edge: both vertical taps on the same coarse row
            else { acc_lo[j] += ly0 * H; acc_hi[j] += ly1 * H; }
        }
    }
    flush_lo(y_cur);
#pragma unroll
    for (int j = 0; j < KG; ++j) acc_lo[j] = acc_hi[j];
    flush_lo(y_cur + 1);
    // coarse rows of the band that no label row touches (cannot happen when upsampling; kept for safety)
    for (int row = max(y_cur + 2, ys0); row < ys_end; ++row) {
#pragma unroll
        for (int j = 0; j < KG; ++j) acc_lo[j] = 0.f;
        flush_lo(row);
    }
    if (Y_lo >= Y_hi) {
        for (int row = ys0; row < ys_end; ++row) {
#pragma unroll
            for (int j = 0; j < KG; ++j) acc_lo[j] = 0.f;
            flush_lo(row);
        }
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

// largest number of fine pixels that share one left tap (host side, exact: same float arithmetic as the device)
int max_cell_px(const CeDims& d) {
    int best = 0, run = 0, prev = -1;
    for (int X = 0; X < d.W; ++X) {
        const int x0 = tap0(d.sx, X);
        run = (x0 == prev) ? run + 1 : 1;
        prev = x0;
        if (run > best) best = run;
    }
    return best;
}

int pick_px(int need) {
    const int opts[] = {1, 2, 3, 5, 9, 17};
    for (int o : opts)
        if (need <= o) return o;
    return 0;
}

int pick_kg(int K) {
    const int opts[] = {8, 6, 5, 4};
    int best = 4, waste = 1 << 30;
    for (int o : opts) {
        const int wst = (K + o - 1) / o * o - K;
        if (wst < waste) { waste = wst; best = o; }
    }
    return best;
}

}  // namespace

extern "C" int cseg_upsample_ce_blocks(int B, int H, int W) {
    // upper bound of the forward grid for any coarse width w <= W (one thread per (label row, coarse column))
    return (int)(((long)B * H * W + 255) / 256);
}

extern "C" int cseg_upsample_ce_fwd(const float* seg, const int64_t* target, const float* weight, int ignore_label,
                                    int B, int K, int h, int w, int H, int W, float* partial, float* out,
                                    int32_t* status, float* lse, cseg_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    CeDims d;
    if (!make_dims(&d, B, K, h, w, H, W, ignore_label)) return 0;
    const int px = pick_px(max_cell_px(d));
    CSEG_REQUIRE(px > 0, "upsample_ce: %d label pixels share one coarse tap (%d -> %d); at most %d are supported", max_cell_px(d), w, W, MAX_PX);
    const long n_cells = (long)B * H * w;
    const int n_blocks = (int)((n_cells + 255) / 256);
#define LAUNCH(P) hipLaunchKernelGGL(ce_fwd_kernel<P>, dim3(n_blocks), dim3(256), 0, stream, seg, target, weight, d, lse, \
                                     partial, status)
    switch (px) {
        case 1: LAUNCH(1); break;
        case 2: LAUNCH(2); break;
        case 3: LAUNCH(3); break;
        case 5: LAUNCH(5); break;
        case 9: LAUNCH(9); break;
        default: LAUNCH(17); break;
    }
#undef LAUNCH
    CSEG_CHECK_LAUNCH("ce_fwd_kernel");
    hipLaunchKernelGGL(ce_finish_kernel, dim3(1), dim3(1024), 0, stream, partial, n_blocks, out);
    CSEG_CHECK_LAUNCH("ce_finish_kernel");
    return 1;
}

extern "C" int cseg_upsample_ce_bwd(const float* seg, const int64_t* target, const float* weight, int ignore_label,
                                    int B, int K, int h, int w, int H, int W, const float* out, const float* d_loss,
                                    const float* lse, float* d_seg, cseg_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    CeDims d;
    if (!make_dims(&d, B, K, h, w, H, W, ignore_label)) return 0;
    CSEG_REQUIRE(lse, "upsample_ce_bwd: needs the log-sum-exp buffer written by cseg_upsample_ce_fwd");
    const int px = pick_px(max_cell_px(d));
    CSEG_REQUIRE(px > 0, "upsample_ce: %d label pixels share one coarse tap (%d -> %d); at most %d are supported", max_cell_px(d), w, W, MAX_PX);
    const int kg = pick_kg(K);
    const int n_groups = (K + kg - 1) / kg, n_bands = (h + BAND - 1) / BAND;
    const int halo = w > 256 ? 1 : 0;
    const int n_cb = (w + (256 - halo) - 1) / (256 - halo);
    const int threads = n_cb > 1 ? 256 : ((w + 63) / 64) * 64;
    const long n_blocks = (long)B * n_cb * n_bands * n_groups;
    CSEG_REQUIRE(n_blocks < 2147483647L, "upsample_ce_bwd: grid too large");
#define LAUNCH(P, G) hipLaunchKernelGGL((ce_bwd_kernel<P, G>), dim3((unsigned)n_blocks), dim3(threads), 0, stream, seg, target, \
                                        weight, lse, d, n_groups, n_bands, halo, out, d_loss, d_seg)
#define BY_KG(P)                                   \
    switch (kg) {                                  \
        case 8: LAUNCH(P, 8); break;               \
        case 6: LAUNCH(P, 6); break;               \
        case 5: LAUNCH(P, 5); break;               \
        default: LAUNCH(P, 4); break;              \
    }
    switch (px) {
        case 1: BY_KG(1); break;
        case 2: BY_KG(2); break;
        case 3: BY_KG(3); break;
        case 5: BY_KG(5); break;
        case 9: BY_KG(9); break;
        default: BY_KG(17); break;
    }
#undef BY_KG
#undef LAUNCH
    CSEG_CHECK_LAUNCH("ce_bwd_kernel");
    return 1;
}
