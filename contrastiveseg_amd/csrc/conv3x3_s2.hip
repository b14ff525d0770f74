// 3x3 / stride 2 / pad 1 convolution, forward and backward-data, split-operand f16x3 arithmetic (cseg_split.h). Round 3.
// Reference: the downsampling convolutions of HRNet's fuse layers (lib/models/backbones/hrnet/hrnet_backbone.py:230-250) and
// transition layers (:652-660), nn.Conv2d(cin, cout, 3, 2, 1, bias=False): 52 of the 309 convolutions of HRNet-W48. On MIOpen they
// ran as miopenSp3AsmConv_*_stride2 (forward, 32 launches, 2.95 ms per step of the benched configuration) and
// miopenSp3AsmConv_*_dilation2 (backward-data, 33 launches, 3.3 ms) -- profiles/r03_step_steady_kernel_stats.csv.
//
// Both kernels are the 16-channel-chunk kernel of conv3x3_sb16.hip (block = 8 waves = 4 rows x 64 columns of output x NT channel
// tiles, wave = (row, half of the channel tiles), K-step = two taps x 16 channels, weights streamed by LDS-DMA into a double buffer,
// fp32 patch fetched with buffer loads, split into two scaled fp16 pieces and stored as [piece][octet][cell] of 16-byte cells) with
// a different patch geometry:
//   forward: the patch of a 4 x 64 output tile is 9 x 129 input pixels. Its columns are stored DE-INTERLEAVED -- cells 0..64 of a
//     patch row hold the even patch columns (kx = 0 and, one cell further, kx = 2), cells 65..128 the odd ones (kx = 1) -- so that
//     the fragment of 16 consecutive output pixels is 16 consecutive cells for every tap (conflict-free ds_read_b128), as at
//     stride 1. Same packed weights as conv3x3_sb16.hip (CSEG_PACK_C3_16).
//   backward-data: dx[ci][2 qy + py][2 qx + px] is, per parity class (py, px), a stride-1 correlation of dy with 1 / 2 / 2 / 4 of
//     the nine taps (cseg_pack.h: CSEG_PACK_C3_S2T). A block owns 4 x 64 "quads" (8 x 128 pixels of dx) and ONE row parity py: two
//     accumulator sets (px = 0, 1), 2 (py = 0) or 3 (py = 1) K-steps per 16-channel chunk, a dy patch of 5 x 65 pixels, and writes
//     whole rows of dx (the two column parities interleaved in registers: 32 contiguous bytes per lane).
#include "cseg_pack.h"
#include "cseg_stats.h"
#include <stdlib.h>

namespace {

constexpr int TR = 4;                 // output rows (forward) / quad rows (backward-data) per block, one per wave
constexpr int TC = 64;                // output columns / quad columns per block
constexpr int NOCT = 2;               // channel octets per chunk
constexpr int STEPS = 5;              // packed K-steps per 16-channel chunk

// ---- forward geometry
constexpr int F_XROWS = 2 * TR + 1;   // 9 patch rows: input rows 2 y0 - 1 .. 2 y0 + 7
constexpr int F_XP = 130;             // cells per patch row: 65 even columns, 64 odd columns, 1 pad
constexpr int F_ODD = 65;             // first odd-column cell
constexpr int F_CELLS = F_XROWS * F_XP;                 // 1170
constexpr int F_PLANE = (F_CELLS + 15) / 16 * 16;       // 1184: 0 mod 256 bytes
constexpr int F_AU = (NOCT * F_CELLS + 511) / 512;      // 5 staging items per thread

// ---- backward-data geometry
constexpr int D_XROWS = TR + 1, D_XP = TC + 1;          // dy patch: quad rows y0 .. y0 + 4, columns x0 .. x0 + 64
constexpr int D_CELLS = D_XROWS * D_XP;                 // 325
constexpr int D_PLANE = (D_CELLS + 15) / 16 * 16;       // 336
constexpr int D_AU = (NOCT * D_CELLS + 511) / 512;      // 2

template <class AR, int NTW, int NTMAX, int PLANE>
__device__ __forceinline__ void s2_kstep(const uint4* __restrict__ ap, const uint4* __restrict__ bp, f32x4 (&acc)[4][NTMAX]) {
    typedef typename AR::frag_t frag_t;
    frag_t a[4][AR::NP];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int p = 0; p < AR::NP; ++p) a[mt][p] = __builtin_bit_cast(frag_t, ap[p * NOCT * PLANE + 16 * mt]);
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) {
        frag_t b[AR::NP];
#pragma unroll
        for (int p = 0; p < AR::NP; ++p) b[p] = __builtin_bit_cast(frag_t, bp[(nt * AR::NP + p) * 64]);
#pragma unroll
        for (int t = 0; t < AR::NTERMS; ++t)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) acc[mt][nt] = AR::mfma(a[mt][AR::ta(t)], b[AR::tb(t)], acc[mt][nt]);
    }
}

// ---------------------------------------------------------------------------------------------------------
// forward: x [B, Cin, 2 Ho, 2 Wo] -> y [B, Cout, Ho, Wo]
// ---------------------------------------------------------------------------------------------------------
template <int NTW, int NTMAX>
__device__ __forceinline__ void s2_store_fwd(const f32x4 (&acc)[4][NTMAX], float* __restrict__ ybc, int co0, size_t oplane, int yy,
                                             int x0, int Wo, int g, int n, float unscale) {
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) {
        float* orow = ybc + (size_t)(co0 + nt * 16 + n) * oplane + (size_t)yy * Wo;
        const bool vec = (Wo & 3) == 0;             // output rows 16-byte aligned (else element by element: cseg_store_row4)
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            const int xx = x0 + 16 * mt + 4 * g;
            const f32x4 v = acc[mt][nt] * unscale;
            cseg_store_row4(orow, nullptr, xx, Wo, vec, v);
        }
    }
}

template <class AR, int NT>
__global__ __launch_bounds__(512, 2) void conv3x3_s2_fwd_kernel(const float* __restrict__ x, const uint4* __restrict__ wp, int Cin,
                                                                int Cout, int Ho, int Wo, int tiles_x, int tiles_y,
                                                                const unsigned* __restrict__ amax_x,
                                                                const unsigned* __restrict__ amax_w, float* __restrict__ y,
                                                                float4* __restrict__ stats, int n_seg) {
    extern __shared__ __attribute__((aligned(16))) uint4 smem_s2[];
    constexpr int NP = AR::NP;
    constexpr int A_CELLS = NP * NOCT * F_PLANE;
    uint4* As = smem_s2;                           // [piece][octet 2][F_PLANE]
    uint4* Bs = smem_s2 + A_CELLS;                 // [2][NT*NP*64]
    constexpr int BSTEP = NT * NP * 64;
    const unsigned ex = split_amax_exp(amax_x), ew = split_amax_exp(amax_w);
    const float xscale = split_scale_of(ex);
    constexpr int NT0 = (NT + 1) / 2, NT1 = NT - NT0;

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int row = wave & 3, half = wave >> 2;
    const int g = lane >> 4, n = lane & 15;
    const int n_cot = Cout / (NT * 16);
    const int H = 2 * Ho, W = 2 * Wo;
    const size_t plane = (size_t)H * W, oplane = (size_t)Ho * Wo;
    int t = blockIdx.x;
    const int tx = t % tiles_x; t /= tiles_x;
    const int ty = t % tiles_y; t /= tiles_y;
    const int cot = t % n_cot;
    const int b = t / n_cot;
    const int x0 = tx * TC, y0 = ty * TR;

    const int n_chunks = Cin / 16;
    const int n_steps = n_chunks * STEPS;
    const uint4* wbase = wp + (size_t)cot * n_steps * BSTEP;

    auto b_glds = [&](int ks, int buf) {
#pragma unroll
        for (int i = 0; i < (NT * NP + 7) / 8; ++i) {
            const int r = wave + 8 * i;
            if (r < NT * NP)
                __builtin_amdgcn_global_load_lds(
                    (const __attribute__((address_space(1))) void*)(wbase + (size_t)ks * BSTEP + r * 64 + lane),
                    (__attribute__((address_space(3))) void*)(Bs + buf * BSTEP + r * 64), 16, 0, 0);
        }
    };

    // staging item = (octet, patch cell); cell c of patch row r: c < 65 -> patch column 2 c, else 2 (c - 65) + 1; input pixel
    // (2 y0 - 1 + r, 2 x0 - 1 + column)
    float apre[F_AU][8];
    auto a_item = [&](int u, int& oct, int& rc, int& yy, int& xx, bool& ok) {
        const int item = tid + 512 * u;
        oct = item / F_CELLS; rc = item - oct * F_CELLS;
        const int r = rc / F_XP, c = rc - r * F_XP;
        const int col = c < F_ODD ? 2 * c : 2 * (c - F_ODD) + 1;
        yy = 2 * y0 - 1 + r; xx = 2 * x0 - 1 + col;
        ok = oct < NOCT && yy >= 0 && yy < H && xx >= 0 && xx < W;
    };
    auto a_issue = [&](int chunk) {
        const float* xc = x + ((size_t)b * Cin + (size_t)chunk * 16) * plane;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)xc, 0, (int)(16 * plane * sizeof(float)),
                                                                            0x00020000);
#pragma unroll
        for (int u = 0; u < F_AU; ++u) {
            int oct, rc, yy, xx;
            bool ok;
            a_item(u, oct, rc, yy, xx, ok);
            const int octc = min(oct, NOCT - 1), yc = min(max(yy, 0), H - 1), xcl = min(max(xx, 0), W - 1);
            const int off = (octc * 8 * (int)plane + yc * W + xcl) * (int)sizeof(float);
#pragma unroll
            for (int j = 0; j < 8; ++j)
                apre[u][j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                                           rs, off, j * (int)plane * (int)sizeof(float), 0));
        }
    };
    auto a_store = [&]() {
#pragma unroll
        for (int u = 0; u < F_AU; ++u) {
            int oct, rc, yy, xx;
            bool ok;
            a_item(u, oct, rc, yy, xx, ok);
            if (oct < NOCT) {
                uint4 cells[NP];
                split_cells8_masked<AR>(apre[u], ok, xscale, cells);            // zero padding / outside the tensor
                const int item = oct * F_PLANE + rc;
#pragma unroll
                for (int p = 0; p < NP; ++p) As[p * NOCT * F_PLANE + item] = cells[p];
            }
        }
    };

    f32x4 acc[4][NT0];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT0; ++nt) acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

    a_issue(0);
    b_glds(0, 0);
    a_store();
    __syncthreads();

    const uint4* a_lane = As + 2 * row * F_XP + n;                     // patch row 2 * row + ky
    const uint4* b_lane = Bs + (half ? NT0 * NP * 64 : 0) + lane;
    int ks = 0, buf = 0;
    for (int c = 0; c < n_chunks; ++c) {
#pragma unroll 1
        for (int s = 0; s < STEPS; ++s) {
            if (ks + 1 < n_steps) b_glds(ks + 1, buf ^ 1);
            if (s == STEPS - 3 && c + 1 < n_chunks) a_issue(c + 1);
            const int tap = min(2 * s + (g >> 1), 8);               // the tenth tap does not exist: zero weights
            const int ky = tap / 3, kx = tap - 3 * ky;
            const int a_off = (g & 1) * F_PLANE + ky * F_XP + (kx == 1 ? F_ODD : kx >> 1);
            if (half == 0) s2_kstep<AR, NT0, NT0, F_PLANE>(a_lane + a_off, b_lane + buf * BSTEP, acc);
            else if (NT1 > 0) s2_kstep<AR, NT1, NT0, F_PLANE>(a_lane + a_off, b_lane + buf * BSTEP, acc);
            if (s == STEPS - 1 && c + 1 < n_chunks) {
                __syncthreads();
                a_store();
            }
            __syncthreads();
            buf ^= 1;
            ++ks;
        }
    }

    const int yy = y0 + row;
    if (yy < Ho) {
        float* ybc = y + (size_t)b * Cout * oplane;
        const int co0 = cot * NT * 16;
        const float unscale = split_unscale_of(ex) * split_unscale_of(ew);
        if (half == 0) s2_store_fwd<NT0, NT0>(acc, ybc, co0, oplane, yy, x0, Wo, g, n, unscale);
        else if (NT1 > 0) s2_store_fwd<NT1, NT0>(acc, ybc, co0 + NT0 * 16, oplane, yy, x0, Wo, g, n, unscale);
        if (stats) {                                // BatchNorm statistics of what was just stored (cseg_stats.h)
            const size_t seg = ((size_t)b * Ho + yy) * tiles_x + tx;
            if (half == 0) cseg_stats_emit<NT0, NT0>(acc, nullptr, co0, unscale, x0, Wo, g, n, stats + (size_t)co0 * n_seg + seg, n_seg);
            else if (NT1 > 0)
                cseg_stats_emit<NT1, NT0>(acc, nullptr, co0 + NT0 * 16, unscale, x0, Wo, g, n, stats + (size_t)(co0 + NT0 * 16) * n_seg + seg,
                                          n_seg);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// backward-data: dy [B, Cout, Ho, Wo] -> dx [B, Cin, 2 Ho, 2 Wo]; block = (quad tile, channel tile group of Cin, row parity py)
// ---------------------------------------------------------------------------------------------------------
template <int NTW, int NTMAX>
__device__ __forceinline__ void s2_store_bwd(const f32x4 (&acc0)[4][NTMAX], const f32x4 (&acc1)[4][NTMAX], float* __restrict__ dxb,
                                             int ci0, size_t plane, int iy, int x0, int Wo, int g, int n, float unscale) {
    const int W = 2 * Wo;
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) {
        float* orow = dxb + (size_t)(ci0 + nt * 16 + n) * plane + (size_t)iy * W;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            const int qx = x0 + 16 * mt + 4 * g;                        // quads qx .. qx + 3 = pixels 2 qx .. 2 qx + 7
            const f32x4 e = acc0[mt][nt] * unscale, o = acc1[mt][nt] * unscale;
            if (qx + 3 < Wo) {
                *reinterpret_cast<float4*>(orow + 2 * qx) = make_float4(e[0], o[0], e[1], o[1]);
                *reinterpret_cast<float4*>(orow + 2 * qx + 4) = make_float4(e[2], o[2], e[3], o[3]);
            } else {
#pragma unroll
                for (int r = 0; r < 3; ++r)
                    if (qx + r < Wo) { orow[2 * (qx + r)] = e[r]; orow[2 * (qx + r) + 1] = o[r]; }
            }
        }
    }
}

template <class AR, int NT>
__global__ __launch_bounds__(512, 2) void conv3x3_s2_bwd_kernel(const float* __restrict__ dy, const uint4* __restrict__ wp, int Cin,
                                                                int Cout, int Ho, int Wo, int tiles_x, int tiles_y,
                                                                const unsigned* __restrict__ amax_dy,
                                                                const unsigned* __restrict__ amax_w, float* __restrict__ dx) {
    extern __shared__ __attribute__((aligned(16))) uint4 smem_s2[];
    constexpr int NP = AR::NP;
    constexpr int A_CELLS = NP * NOCT * D_PLANE;
    uint4* As = smem_s2;
    uint4* Bs = smem_s2 + A_CELLS;
    constexpr int BSTEP = NT * NP * 64;
    const unsigned ed = split_amax_exp(amax_dy), ew = split_amax_exp(amax_w);
    const float dscale = split_scale_of(ed);
    constexpr int NT0 = (NT + 1) / 2, NT1 = NT - NT0;

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int row = wave & 3, half = wave >> 2;
    const int g = lane >> 4, n = lane & 15;
    const int n_cot = Cin / (NT * 16);                                  // channel tile groups of the OUTPUT (dx) channels
    const size_t oplane = (size_t)Ho * Wo, plane = 4 * oplane;
    int t = blockIdx.x;
    const int py = t & 1; t >>= 1;
    const int tx = t % tiles_x; t /= tiles_x;
    const int ty = t % tiles_y; t /= tiles_y;
    const int cot = t % n_cot;
    const int b = t / n_cot;
    const int x0 = tx * TC, y0 = ty * TR;

    const int n_chunks = Cout / 16;
    const uint4* wbase = wp + (size_t)cot * n_chunks * STEPS * BSTEP;
    // this block's K-steps of a chunk: packed steps q0 .. q0 + nq - 1 (cseg_pack.h: q = 0, 1 belong to py = 0; 2, 3, 4 to py = 1)
    const int q0 = py ? 2 : 0, nq = py ? 3 : 2;
    const int n_steps = n_chunks * nq;

    auto b_glds = [&](int ls, int buf) {            // ls = local step index: chunk = ls / nq, q = q0 + ls % nq
        const int chunk = ls / nq, q = q0 + ls - chunk * nq;
        const uint4* src = wbase + (size_t)(chunk * STEPS + q) * BSTEP;
#pragma unroll
        for (int i = 0; i < (NT * NP + 7) / 8; ++i) {
            const int r = wave + 8 * i;
            if (r < NT * NP)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + r * 64 + lane),
                                                 (__attribute__((address_space(3))) void*)(Bs + buf * BSTEP + r * 64), 16, 0, 0);
        }
    };

    float apre[D_AU][8];
    auto a_item = [&](int u, int& oct, int& rc, int& yy, int& xx, bool& ok) {
        const int item = tid + 512 * u;
        oct = item / D_CELLS; rc = item - oct * D_CELLS;
        const int r = rc / D_XP, c = rc - r * D_XP;
        yy = y0 + r; xx = x0 + c;
        ok = oct < NOCT && yy < Ho && xx < Wo;
    };
    auto a_issue = [&](int chunk) {
        const float* dc = dy + ((size_t)b * Cout + (size_t)chunk * 16) * oplane;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)dc, 0, (int)(16 * oplane * sizeof(float)),
                                                                            0x00020000);
#pragma unroll
        for (int u = 0; u < D_AU; ++u) {
            int oct, rc, yy, xx;
            bool ok;
            a_item(u, oct, rc, yy, xx, ok);
            const int octc = min(oct, NOCT - 1), yc = min(yy, Ho - 1), xcl = min(xx, Wo - 1);
            const int off = (octc * 8 * (int)oplane + yc * Wo + xcl) * (int)sizeof(float);
#pragma unroll
            for (int j = 0; j < 8; ++j)
                apre[u][j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                                           rs, off, j * (int)oplane * (int)sizeof(float), 0));
        }
    };
    auto a_store = [&]() {
#pragma unroll
        for (int u = 0; u < D_AU; ++u) {
            int oct, rc, yy, xx;
            bool ok;
            a_item(u, oct, rc, yy, xx, ok);
            if (oct < NOCT) {
                uint4 cells[NP];
                split_cells8_masked<AR>(apre[u], ok, dscale, cells);
                const int item = oct * D_PLANE + rc;
#pragma unroll
                for (int p = 0; p < NP; ++p) As[p * NOCT * D_PLANE + item] = cells[p];
            }
        }
    };

    f32x4 acc0[4][NT0], acc1[4][NT0];               // column parity px = 0 / 1
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT0; ++nt) {
            acc0[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
            acc1[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }

    a_issue(0);
    b_glds(0, 0);
    a_store();
    __syncthreads();

    const uint4* a_lane = As + row * D_XP + n;
    const uint4* b_lane = Bs + (half ? NT0 * NP * 64 : 0) + lane;
    const int second = g >> 1;
    int ls = 0, buf = 0;
    for (int c = 0; c < n_chunks; ++c) {
#pragma unroll 1
        for (int s = 0; s < nq; ++s) {
            if (ls + 1 < n_steps) b_glds(ls + 1, buf ^ 1);
            if (s == 0 && c + 1 < n_chunks) a_issue(c + 1);
            // (dyy, dxx) of this lane group's tap: the tap (ky, kx) reads dy[qy + (ky == 0)][qx + (kx == 0)]
            const int q = q0 + s;
            const int tap = max(pack_s2t_tap(q, second), 0);            // "none" (zero weights): any cell of the patch
            const int ky = tap / 3, kx = tap - 3 * ky;
            const int a_off = (g & 1) * D_PLANE + (ky == 0 ? D_XP : 0) + (kx == 0 ? 1 : 0);
            const bool px1 = q == 1 || q >= 3;                          // which accumulator set (uniform over the block)
            if (half == 0) {
                if (px1) s2_kstep<AR, NT0, NT0, D_PLANE>(a_lane + a_off, b_lane + buf * BSTEP, acc1);
                else s2_kstep<AR, NT0, NT0, D_PLANE>(a_lane + a_off, b_lane + buf * BSTEP, acc0);
            } else if (NT1 > 0) {
                if (px1) s2_kstep<AR, NT1, NT0, D_PLANE>(a_lane + a_off, b_lane + buf * BSTEP, acc1);
                else s2_kstep<AR, NT1, NT0, D_PLANE>(a_lane + a_off, b_lane + buf * BSTEP, acc0);
            }
            if (s == nq - 1 && c + 1 < n_chunks) {
                __syncthreads();
                a_store();
            }
            __syncthreads();
            buf ^= 1;
            ++ls;
        }
    }

    const int qy = y0 + row;
    if (qy < Ho) {
        float* dxb = dx + (size_t)b * Cin * plane;
        const int ci0 = cot * NT * 16;
        const float unscale = split_unscale_of(ed) * split_unscale_of(ew);
        if (half == 0) s2_store_bwd<NT0, NT0>(acc0, acc1, dxb, ci0, plane, 2 * qy + py, x0, Wo, g, n, unscale);
        else if (NT1 > 0) s2_store_bwd<NT1, NT0>(acc0, acc1, dxb, ci0 + NT0 * 16, plane, 2 * qy + py, x0, Wo, g, n, unscale);
    }
}

template <int NT>
__global__ __launch_bounds__(256) void pack_s2_kernel(const float* __restrict__ w, int Cout, int Cin, int transposed,
                                                      const unsigned* __restrict__ amax_w, uint4* __restrict__ wp, int total) {
    const float wscale = split_scale_of(split_amax_exp(amax_w));         // every thread (shuffles inside)
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= total) return;
    if (transposed) pack_elem_c3_s2t<SplitF16x3>(w, Cout, Cin, NT, wscale, wp, e);
    else pack_elem_c3_16<SplitF16x3>(w, Cout, Cin, 0, NT, wscale, wp, e);
}

bool s2_nt_ok(int nt, int conv_out) { return (nt == 3 || nt == 4 || nt == 6) && conv_out % (nt * 16) == 0; }      // 4: 256 channels

template <class K>
bool s2_set_lds(K kernel, size_t lds, bool& done) {
    if (done) return true;
    if (hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
        cseg_set_error("conv3x3_s2: cannot raise dynamic LDS to %zu bytes", lds);
        return false;
    }
    done = true;
    return true;
}

}  // namespace

// bytes of the packed operator (either direction): conv_in = its input channels (% 16), conv_out = its output channels
extern "C" size_t cseg_conv3x3_s2_split_packed_bytes(int conv_in, int conv_out) {
    if (conv_in <= 0 || conv_out <= 0 || conv_in % 16 || (conv_out % 48 && conv_out % 64)) return 0;
    return (size_t)(conv_out / 16) * pack_steps_c3_16(conv_in) * 2 * 64 * sizeof(uint4);
}

// plan of one pack call for the batched packer (cseg_split_pack_batch): kind = CSEG_PACK_C3_16 (forward) / CSEG_PACK_C3_S2T
extern "C" int cseg_conv3x3_s2_split_plan(int conv_in, int conv_out, int transposed, int nt, int* kind, long* threads) {
    if (!kind || !threads || conv_in <= 0 || conv_out <= 0 || conv_in % 16 || !s2_nt_ok(nt, conv_out)) return 0;
    *kind = transposed ? CSEG_PACK_C3_S2T : CSEG_PACK_C3_16;
    *threads = (long)(conv_out / 16) * pack_steps_c3_16(conv_in) * 64;
    return 1;
}

// w = the convolution's [Cout, Cin, 3, 3]; transposed = 0: forward operator (Cin -> Cout), 1: backward-data operator (Cout -> Cin)
extern "C" int cseg_conv3x3_s2_split_pack(const float* w, int Cout, int Cin, int transposed, int nt, const unsigned* amax_w, void* wp,
                                          cseg_stream_t stream_) {
    const int conv_in = transposed ? Cout : Cin, conv_out = transposed ? Cin : Cout;
    CSEG_REQUIRE(w && wp && amax_w, "conv3x3_s2 pack: null pointer");
    CSEG_REQUIRE(conv_in % 16 == 0 && s2_nt_ok(nt, conv_out), "conv3x3_s2 pack: unsupported channels %d -> %d with %d tiles per block",
                 conv_in, conv_out, nt);
    CSEG_REQUIRE((reinterpret_cast<uintptr_t>(wp) & 15) == 0, "conv3x3_s2 pack: packed buffer must be 16-byte aligned");
    const long total = (long)(conv_out / 16) * pack_steps_c3_16(conv_in) * 64;
    CSEG_REQUIRE(total < 2147483647L, "conv3x3_s2 pack: too large");
    hipStream_t stream = (hipStream_t)stream_;
    if (nt == 3)
        hipLaunchKernelGGL(pack_s2_kernel<3>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, w, Cout, Cin, transposed, amax_w,
                           (uint4*)wp, (int)total);
    else if (nt == 4)
        hipLaunchKernelGGL(pack_s2_kernel<4>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, w, Cout, Cin, transposed, amax_w,
                           (uint4*)wp, (int)total);
    else
        hipLaunchKernelGGL(pack_s2_kernel<6>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, w, Cout, Cin, transposed, amax_w,
                           (uint4*)wp, (int)total);
    CSEG_CHECK_LAUNCH("conv3x3_s2 pack");
    return 1;
}

static int s2_fwd_impl(const float* x, const void* wp, int B, int Cin, int Cout, int Ho, int Wo, int nt, const unsigned* amax_x,
                       const unsigned* amax_w, float* y, float4* stats, cseg_stream_t stream_) {
    CSEG_REQUIRE(x && wp && y && amax_x && amax_w, "conv3x3_s2_fwd: null pointer");
    CSEG_REQUIRE((reinterpret_cast<uintptr_t>(stats) & 15) == 0, "conv3x3_s2_fwd: the statistics buffer must be 16-byte aligned");
    CSEG_REQUIRE(B > 0 && Ho > 0 && Wo > 0 && Cin > 0 && Cin % 16 == 0 && s2_nt_ok(nt, Cout) &&
                     (long)Ho * Wo * 4 * 16 * 4 < 2147483647L,
                 "conv3x3_s2_fwd: unsupported shape Cin=%d Cout=%d out %dx%d with %d channel tiles per block", Cin, Cout, Ho, Wo, nt);
    CSEG_REQUIRE((reinterpret_cast<uintptr_t>(wp) & 15) == 0 && (reinterpret_cast<uintptr_t>(y) & 15) == 0,
                 "conv3x3_s2_fwd: buffers must be 16-byte aligned");
    const int tiles_x = (Wo + TC - 1) / TC, tiles_y = (Ho + TR - 1) / TR;
    const long n_tiles = (long)B * (Cout / (nt * 16)) * tiles_y * tiles_x;
    CSEG_REQUIRE(n_tiles < 2147483647L, "conv3x3_s2_fwd: grid too large");
    hipStream_t stream = (hipStream_t)stream_;
    const size_t lds = sizeof(uint4) * (2 * NOCT * F_PLANE + 2 * nt * 2 * 64);
    static bool set3 = false, set4 = false, set6 = false;
    if (nt == 3) {
        if (!s2_set_lds(conv3x3_s2_fwd_kernel<SplitF16x3, 3>, lds, set3)) return 0;
        hipLaunchKernelGGL((conv3x3_s2_fwd_kernel<SplitF16x3, 3>), dim3((unsigned)n_tiles), dim3(512), lds, stream, x, (const uint4*)wp, Cin,
                           Cout, Ho, Wo, tiles_x, tiles_y, amax_x, amax_w, y, stats, B * Ho * tiles_x);
    } else if (nt == 4) {
        if (!s2_set_lds(conv3x3_s2_fwd_kernel<SplitF16x3, 4>, lds, set4)) return 0;
        hipLaunchKernelGGL((conv3x3_s2_fwd_kernel<SplitF16x3, 4>), dim3((unsigned)n_tiles), dim3(512), lds, stream, x, (const uint4*)wp, Cin,
                           Cout, Ho, Wo, tiles_x, tiles_y, amax_x, amax_w, y, stats, B * Ho * tiles_x);
    } else {
        if (!s2_set_lds(conv3x3_s2_fwd_kernel<SplitF16x3, 6>, lds, set6)) return 0;
        hipLaunchKernelGGL((conv3x3_s2_fwd_kernel<SplitF16x3, 6>), dim3((unsigned)n_tiles), dim3(512), lds, stream, x, (const uint4*)wp, Cin,
                           Cout, Ho, Wo, tiles_x, tiles_y, amax_x, amax_w, y, stats, B * Ho * tiles_x);
    }
    CSEG_CHECK_LAUNCH("conv3x3_s2_fwd_kernel");
    return 1;
}

extern "C" int cseg_conv3x3_s2_split_fwd(const float* x, const void* wp, int B, int Cin, int Cout, int Ho, int Wo, int nt,
                                         const unsigned* amax_x, const unsigned* amax_w, float* y, cseg_stream_t stream_) {
    return s2_fwd_impl(x, wp, B, Cin, Cout, Ho, Wo, nt, amax_x, amax_w, y, nullptr, stream_);
}

// with the BatchNorm statistics of the output from the epilogue: stats [Cout][cseg_conv_stat_segments(0, B, Ho, Wo)] float4
extern "C" int cseg_conv3x3_s2_split_fwd_st(const float* x, const void* wp, int B, int Cin, int Cout, int Ho, int Wo, int nt,
                                            const unsigned* amax_x, const unsigned* amax_w, float* y, float* stats,
                                            cseg_stream_t stream_) {
    CSEG_REQUIRE(stats, "conv3x3_s2_split_fwd_st: null statistics buffer");
    return s2_fwd_impl(x, wp, B, Cin, Cout, Ho, Wo, nt, amax_x, amax_w, y, reinterpret_cast<float4*>(stats), stream_);
}

// nt tiles the channels of dx (= Cin of the convolution)
extern "C" int cseg_conv3x3_s2_split_bwd(const float* dy, const void* wp, int B, int Cin, int Cout, int Ho, int Wo, int nt,
                                         const unsigned* amax_dy, const unsigned* amax_w, float* dx, cseg_stream_t stream_) {
    CSEG_REQUIRE(dy && wp && dx && amax_dy && amax_w, "conv3x3_s2_bwd: null pointer");
    CSEG_REQUIRE(B > 0 && Ho > 0 && Wo > 0 && Cout > 0 && Cout % 16 == 0 && s2_nt_ok(nt, Cin) && Wo % 2 == 0 &&
                     (long)Ho * Wo * 16 * 4 < 2147483647L,
                 "conv3x3_s2_bwd: unsupported shape Cin=%d Cout=%d out %dx%d with %d channel tiles per block", Cin, Cout, Ho, Wo, nt);
    CSEG_REQUIRE((reinterpret_cast<uintptr_t>(wp) & 15) == 0 && (reinterpret_cast<uintptr_t>(dx) & 15) == 0,
                 "conv3x3_s2_bwd: buffers must be 16-byte aligned");
    const int tiles_x = (Wo + TC - 1) / TC, tiles_y = (Ho + TR - 1) / TR;
    const long n_blocks = (long)B * (Cin / (nt * 16)) * tiles_y * tiles_x * 2;
    CSEG_REQUIRE(n_blocks < 2147483647L, "conv3x3_s2_bwd: grid too large");
    hipStream_t stream = (hipStream_t)stream_;
    const size_t lds = sizeof(uint4) * (2 * NOCT * D_PLANE + 2 * nt * 2 * 64);
    static bool set3 = false, set4 = false, set6 = false;
    if (nt == 3) {
        if (!s2_set_lds(conv3x3_s2_bwd_kernel<SplitF16x3, 3>, lds, set3)) return 0;
        hipLaunchKernelGGL((conv3x3_s2_bwd_kernel<SplitF16x3, 3>), dim3((unsigned)n_blocks), dim3(512), lds, stream, dy, (const uint4*)wp,
                           Cin, Cout, Ho, Wo, tiles_x, tiles_y, amax_dy, amax_w, dx);
    } else if (nt == 4) {
        if (!s2_set_lds(conv3x3_s2_bwd_kernel<SplitF16x3, 4>, lds, set4)) return 0;
        hipLaunchKernelGGL((conv3x3_s2_bwd_kernel<SplitF16x3, 4>), dim3((unsigned)n_blocks), dim3(512), lds, stream, dy, (const uint4*)wp,
                           Cin, Cout, Ho, Wo, tiles_x, tiles_y, amax_dy, amax_w, dx);
    } else {
        if (!s2_set_lds(conv3x3_s2_bwd_kernel<SplitF16x3, 6>, lds, set6)) return 0;
        hipLaunchKernelGGL((conv3x3_s2_bwd_kernel<SplitF16x3, 6>), dim3((unsigned)n_blocks), dim3(512), lds, stream, dy, (const uint4*)wp,
                           Cin, Cout, Ho, Wo, tiles_x, tiles_y, amax_dy, amax_w, dx);
    }
    CSEG_CHECK_LAUNCH("conv3x3_s2_bwd_kernel");
    return 1;
}
