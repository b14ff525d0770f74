// 3x3 / stride 1 / pad 1 split-bf16 convolution, variant for SMALL channel counts (CSEG_CONV3X3_SB_VAR=2, opt-in, first
// hardware run pending): the kernel of conv3x3_sb.hip with 16-channel chunks instead of 32-channel ones.
// Why: at 48 / 96 / 192 channels conv3x3_sb_kernel<3|6> runs at 0.26-0.37 of its roof (95 / 62 / 80 us per launch at the
// benched shapes, 29 ms per step in total). A block there owns 76 KB of LDS for the split patch of a 32-channel chunk, so
// ONE block fits a CU, and its phases -- wait for the patch loads, split + store, 9 + 5 barrier-separated K-steps of 24-48
// MFMAs per wave, store -- have nothing to overlap with. With 16-channel chunks the patch image is 38 KB, a block needs
// 56 KB (3 channel tiles) / 75 KB (6), the patch loads are buffer loads (one 32-bit offset per staging item: the kernel
// fits 128 VGPRs), and TWO blocks share a CU: one block's barriers and staging overlap the other's MFMAs.
// Every chunk is processed like the 16-channel tail of the original: a K-step = two taps x 16 channels (lane groups 0,1 =
// first tap, 2,3 = second; the ninth tap pairs with zero weights), 5 K-steps per chunk, (Cin / 16) * 5 in total (15 instead
// of 14 at 48 channels). Packing, fragment layout, B streaming (LDS-DMA, double-buffered) and the epilogue are the same.
// Entry points: the cseg_conv3x3_sb_* family dispatches here when CSEG_CONV3X3_SB_VAR=2 (packing and forward must run under
// the same setting; kernels.conv3x3_sb_run does both back to back).
// Round 3: default for 48 / 192 output channels; written against the arithmetic traits of cseg_split.h (bf16x6 and f16x3).
#include "cseg_sb16_tile.h"
#include <stdlib.h>

namespace {

using namespace cseg_sb16t;

// Packed weights: Wp[co_tile][kstep][nt][piece][lane] of uint4 (8 bf16, element j), lane = 16*g + n; K-step ks = 5*chunk + q:
//   value(co = (co_tile*NT + nt)*16 + n, ci = 16*chunk + 8*(g&1) + j, tap = 2q + (g>>1))   (zero when tap > 8)
template <class AR>
__global__ __launch_bounds__(256) void pack_weights_sb16_kernel(const float* __restrict__ w, int Cout, int Cin,
                                                                int transpose_flip, int NT, const unsigned* __restrict__ amax_w,
                                                                uint4* __restrict__ wp, int total) {
    const float wscale = AR::SCALED ? split_scale_of(split_amax_exp(amax_w)) : 1.f;      // every thread (shuffles inside)
    const int e = blockIdx.x * 256 + threadIdx.x;          // one thread per (co_tile, kstep, nt, lane)
    if (e >= total) return;
    pack_elem_c3_16<AR>(w, Cout, Cin, transpose_flip, NT, wscale, wp, e);
}

// 8 waves: wave = (row = wave & 3, half = wave >> 2); the halves split the NT channel tiles. Two blocks per CU = 4 waves per
// SIMD (the second launch-bounds argument of HIP is waves per execution unit): 128 VGPRs.
template <class AR, int NT>
__global__ __launch_bounds__(512, NT == 3 ? 4 : 2) void conv3x3_sb16_kernel(const float* __restrict__ x, const uint4* __restrict__ wp,
                                                              const float* __restrict__ bias, const float* __restrict__ addend, int Cin, int Cout, int H,
                                                              int W, int tiles_x, int tiles_y,
                                                              const unsigned* __restrict__ amax_x,
                                                              const unsigned* __restrict__ amax_w, float* __restrict__ y,
                                                              float4* __restrict__ stats, int n_seg, int xmap) {
    extern __shared__ __attribute__((aligned(16))) uint4 smem_s16[];
    constexpr int NP = AR::NP;
    constexpr int A_CELLS = NP * NOCT * PLANE;
    uint4* As = smem_s16;                          // [piece NP][octet 2][CELLS]
    uint4* Bs = smem_s16 + A_CELLS;                // [2][NT*NP*64]
    constexpr int BSTEP = NT * NP * 64;            // uint4 per K-step
    const unsigned ex = AR::SCALED ? split_amax_exp(amax_x) : 141u, ew = AR::SCALED ? split_amax_exp(amax_w) : 141u;
    const float xscale = split_scale_of(ex);       // 1 for the unscaled arithmetic
    constexpr int NT0 = (NT + 1) / 2, NT1 = NT - NT0;

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int row = wave & 3, half = wave >> 2;
    const int g = lane >> 4, n = lane & 15;
    const int n_cot = Cout / (NT * 16);
    const size_t plane = (size_t)H * W;
    // plain order: column tile fastest, then row tile, channel tile group, image. XCD-aware order (opt-in): every XCD gets a
    // contiguous run of logical blocks with the CHANNEL TILE GROUP fastest -- the n_cot blocks that read one patch run side by side
    // on one XCD, followed by the neighbouring tiles that share its halo
    int t = cseg_xcd_block(blockIdx.x, gridDim.x, xmap);
    const bool cot_first = xmap && (gridDim.x & 7) == 0;
    int cot = 0;
    if (cot_first) { cot = t % n_cot; t /= n_cot; }
    const int tx = t % tiles_x; t /= tiles_x;
    const int ty = t % tiles_y; t /= tiles_y;
    if (!cot_first) { cot = t % n_cot; t /= n_cot; }
    const int b = t;
    const int x0 = tx * TC, y0 = ty * TR;

    const int n_chunks = Cin / 16;
    const int n_steps = n_chunks * STEPS;
    const uint4* wbase = wp + (size_t)cot * n_steps * BSTEP;

    auto b_glds = [&](int ks, int buf) {
#pragma unroll
        for (int i = 0; i < (NT * NP + 7) / 8; ++i) {
            const int r = wave + 8 * i;                  // one 1 KB row (channel tile, piece) per wave instruction
            if (r < NT * NP)
                __builtin_amdgcn_global_load_lds(
                    (const __attribute__((address_space(1))) void*)(wbase + (size_t)ks * BSTEP + r * 64 + lane),
                    (__attribute__((address_space(3))) void*)(Bs + buf * BSTEP + r * 64), 16, 0, 0);
        }
    };

    // A staging: item = (octet, patch pixel); 8 channel loads each (coalesced along the row), branch-free: the address is
    // clamped into the tensor and the value masked when it is stored
    float apre[AU][8];
    auto a_item = [&](int u, int& oct, int& rc, bool& ok) {
        const int item = tid + 512 * u;
        oct = item / CELLS; rc = item - oct * CELLS;
        const int r = rc / XCOLS, col = rc - r * XCOLS;
        const int yy = y0 + r - 1, xx = x0 + col - 1;
        ok = oct < NOCT && yy >= 0 && yy < H && xx >= 0 && xx < W;
    };
    auto a_issue = [&](int chunk) {
        const float* xc = x + ((size_t)b * Cin + (size_t)chunk * 16) * plane;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)xc, 0, (int)(16 * plane * sizeof(float)),
                                                                            0x00020000);
#pragma unroll
        for (int u = 0; u < AU; ++u) {
            int oct, rc;
            bool ok;
            a_item(u, oct, rc, ok);
            const int r = rc / XCOLS, col = rc - r * XCOLS;
            const int octc = min(oct, NOCT - 1), yc = min(max(y0 + r - 1, 0), H - 1), xcl = min(max(x0 + col - 1, 0), W - 1);
            const int off = (octc * 8 * (int)plane + yc * W + xcl) * (int)sizeof(float);
#pragma unroll
            for (int j = 0; j < 8; ++j)
                apre[u][j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                                           rs, off, j * (int)plane * (int)sizeof(float), 0));
        }
    };
    auto a_store = [&]() {
#pragma unroll
        for (int u = 0; u < AU; ++u) {
            int oct, rc;
            bool ok;
            a_item(u, oct, rc, ok);
            if (oct < NOCT) {
                uint4 cells[NP];
                split_cells8_masked<AR>(apre[u], ok, xscale, cells);            // zero padding / outside the tensor
                const int item = oct * PLANE + rc;
#pragma unroll
                for (int p = 0; p < NP; ++p) As[p * NOCT * PLANE + item] = cells[p];
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

    const uint4* a_lane = As + row * XCOLS + n;                        // + octet / tap offset per K-step
    const uint4* b_lane = Bs + (half ? NT0 * NP * 64 : 0) + lane;      // + buffer offset per K-step
    int ks = 0, buf = 0;
    for (int c = 0; c < n_chunks; ++c) {
#pragma unroll 1
        for (int s = 0; s < STEPS; ++s) {
            const bool more = ks + 1 < n_steps;
            if (more) b_glds(ks + 1, buf ^ 1);          // that buffer was last read in step ks - 1 (barrier since)
            if (s == STEPS - 3 && c + 1 < n_chunks) a_issue(c + 1);
            // the ninth tap is paired with a tenth that does not exist: its packed weights are zero, so whatever (finite)
            // patch values those lanes read contribute nothing
            const int tap = min(2 * s + (g >> 1), 8);
            const int ky = tap / 3, kx = tap - 3 * ky;
            const int a_off = (g & 1) * PLANE + ky * XCOLS + kx;
            if (half == 0) sb16_kstep<AR, NT0, NT0>(a_lane + a_off, b_lane + buf * BSTEP, acc);
            else if (NT1 > 0) sb16_kstep<AR, NT1, NT0>(a_lane + a_off, b_lane + buf * BSTEP, acc);
            if (s == STEPS - 1 && c + 1 < n_chunks) {
                __syncthreads();                    // every wave is done with this chunk's patch
                a_store();
            }
            __syncthreads();
            buf ^= 1;
            ++ks;
        }
    }

    const int yy = y0 + row;
    if (yy < H) {
        float* ybc = y + (size_t)b * Cout * plane;
        const int co0 = cot * NT * 16;
        const float unscale = split_unscale_of(ex) * split_unscale_of(ew);
        const float* abc = addend ? addend + (size_t)b * Cout * plane : nullptr;
        if (half == 0) sb16_store<NT0, NT0>(acc, ybc, bias, abc, co0, plane, yy, x0, W, g, n, unscale);
        else if (NT1 > 0) sb16_store<NT1, NT0>(acc, ybc, bias, abc, co0 + NT0 * 16, plane, yy, x0, W, g, n, unscale);
        if (stats) {                                // BatchNorm statistics of what was just stored (cseg_stats.h)
            const size_t seg = ((size_t)b * H + yy) * tiles_x + tx;
            if (half == 0) cseg_stats_emit<NT0, NT0>(acc, bias, co0, unscale, x0, W, g, n, stats + (size_t)co0 * n_seg + seg, n_seg);
            else if (NT1 > 0)
                cseg_stats_emit<NT1, NT0>(acc, bias, co0 + NT0 * 16, unscale, x0, W, g, n, stats + (size_t)(co0 + NT0 * 16) * n_seg + seg, n_seg);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// Round 5: the one-tile kernel above with a DILATION (2 or 4, padding = dilation): the 3x3 convolutions of the dilated ResNet stages
// of DeepLab-V3 (reference lib/models/backbones/resnet/resnet_backbone.py:88-101 `_nostride_dilate`: layer3 at rate 2, layer4 at
// rate 4; 23 + 1 of the 3x3 layers of R-101-d8, 256 -> 256 and 512 -> 512 on 65 x 129 maps). Everything is the kernel above --
// same packed weights (CSEG_PACK_C3_16), K-steps, fragment layout, epilogue -- except the patch: (TR + 2 d) x (TC + 2 d) pixels
// around the 4 x 64 tile, tap (ky, kx) at cell offset (ky d, kx d). Backward-data = the same kernel on the transposed packing.
// ASPP's rates (12 / 24 / 36) would need patches of 28 .. 76 rows for 4 output rows and stay on MIOpen.
// ---------------------------------------------------------------------------------------------------------
template <int DIL>
struct DilGeom {
    static constexpr int XROWS_D = TR + 2 * DIL, XCOLS_D = TC + 2 * DIL;
    static constexpr int CELLS_D = XROWS_D * XCOLS_D;
    static constexpr int PLANE_D = (CELLS_D + 15) / 16 * 16;
    static constexpr int A_ITEMS_D = NOCT * CELLS_D;
    static constexpr int AU_D = (A_ITEMS_D + 511) / 512;
};

template <class AR, int NT, int DIL>
__global__ __launch_bounds__(512, 2) void conv3x3_sb16d_kernel(const float* __restrict__ x, const uint4* __restrict__ wp,
                                                               const float* __restrict__ bias, const float* __restrict__ addend, int Cin, int Cout, int H,
                                                               int W, int tiles_x, int tiles_y, const unsigned* __restrict__ amax_x,
                                                               const unsigned* __restrict__ amax_w, float* __restrict__ y,
                                                               float4* __restrict__ stats, int n_seg, int xmap) {
    extern __shared__ __attribute__((aligned(16))) uint4 smem_s16d[];
    typedef DilGeom<DIL> G;
    constexpr int NP = AR::NP;
    constexpr int A_CELLS = NP * NOCT * G::PLANE_D;
    uint4* As = smem_s16d;                         // [piece NP][octet 2][PLANE_D]
    uint4* Bs = smem_s16d + A_CELLS;               // [2][NT*NP*64]
    constexpr int BSTEP = NT * NP * 64;
    const unsigned ex = AR::SCALED ? split_amax_exp(amax_x) : 141u, ew = AR::SCALED ? split_amax_exp(amax_w) : 141u;
    const float xscale = split_scale_of(ex);
    constexpr int NT0 = (NT + 1) / 2, NT1 = NT - NT0;

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int row = wave & 3, half = wave >> 2;
    const int g = lane >> 4, n = lane & 15;
    const int n_cot = Cout / (NT * 16);
    const size_t plane = (size_t)H * W;
    int t = cseg_xcd_block(blockIdx.x, gridDim.x, xmap);
    const bool cot_first = xmap && (gridDim.x & 7) == 0;
    int cot = 0;
    if (cot_first) { cot = t % n_cot; t /= n_cot; }
    const int tx = t % tiles_x; t /= tiles_x;
    const int ty = t % tiles_y; t /= tiles_y;
    if (!cot_first) { cot = t % n_cot; t /= n_cot; }
    const int b = t;
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
    float apre[G::AU_D][8];
    auto a_item = [&](int u, int& oct, int& rc, bool& ok) {
        const int item = tid + 512 * u;
        oct = item / G::CELLS_D; rc = item - oct * G::CELLS_D;
        const int r = rc / G::XCOLS_D, col = rc - r * G::XCOLS_D;
        const int yy = y0 + r - DIL, xx = x0 + col - DIL;
        ok = oct < NOCT && yy >= 0 && yy < H && xx >= 0 && xx < W;
    };
    auto a_issue = [&](int chunk) {
        const float* xc = x + ((size_t)b * Cin + (size_t)chunk * 16) * plane;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)xc, 0, (int)(16 * plane * sizeof(float)),
                                                                            0x00020000);
#pragma unroll
        for (int u = 0; u < G::AU_D; ++u) {
            int oct, rc;
            bool ok;
            a_item(u, oct, rc, ok);
            const int r = rc / G::XCOLS_D, col = rc - r * G::XCOLS_D;
            const int octc = min(oct, NOCT - 1), yc = min(max(y0 + r - DIL, 0), H - 1), xcl = min(max(x0 + col - DIL, 0), W - 1);
            const int off = (octc * 8 * (int)plane + yc * W + xcl) * (int)sizeof(float);
#pragma unroll
            for (int j = 0; j < 8; ++j)
                apre[u][j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                                           rs, off, j * (int)plane * (int)sizeof(float), 0));
        }
    };
    auto a_store = [&]() {
#pragma unroll
        for (int u = 0; u < G::AU_D; ++u) {
            int oct, rc;
            bool ok;
            a_item(u, oct, rc, ok);
            if (oct < NOCT) {
                uint4 cells[NP];
                split_cells8_masked<AR>(apre[u], ok, xscale, cells);            // zero padding / outside the tensor
                const int item = oct * G::PLANE_D + rc;
#pragma unroll
                for (int p = 0; p < NP; ++p) As[p * NOCT * G::PLANE_D + item] = cells[p];
            }
        }
    };
    // one K-step of a wave on the dilated image (sb16_kstep with this geometry's plane pitch)
    auto kstep = [&](const uint4* ap, const uint4* bp, f32x4 (&acc)[4][NT0], int ntw) {
        typedef typename AR::frag_t frag_t;
        frag_t a[4][NP];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int p = 0; p < NP; ++p) a[mt][p] = __builtin_bit_cast(frag_t, ap[p * NOCT * G::PLANE_D + 16 * mt]);
#pragma unroll
        for (int nt = 0; nt < NT0; ++nt) {
            if (nt < ntw) {
                frag_t bfr[NP];
#pragma unroll
                for (int p = 0; p < NP; ++p) bfr[p] = __builtin_bit_cast(frag_t, bp[(nt * NP + p) * 64]);
#pragma unroll
                for (int tt = 0; tt < AR::NTERMS; ++tt)
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt) acc[mt][nt] = AR::mfma(a[mt][AR::ta(tt)], bfr[AR::tb(tt)], acc[mt][nt]);
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

    const uint4* a_lane = As + row * G::XCOLS_D + n;
    const uint4* b_lane = Bs + (half ? NT0 * NP * 64 : 0) + lane;
    int ks = 0, buf = 0;
    for (int c = 0; c < n_chunks; ++c) {
#pragma unroll 1
        for (int s_ = 0; s_ < STEPS; ++s_) {
            const bool more = ks + 1 < n_steps;
            if (more) b_glds(ks + 1, buf ^ 1);
            if (s_ == STEPS - 3 && c + 1 < n_chunks) a_issue(c + 1);
            const int tap = min(2 * s_ + (g >> 1), 8);                 // the tenth tap slot multiplies zero weights
            const int ky = tap / 3, kx = tap - 3 * ky;
            const int a_off = (g & 1) * G::PLANE_D + ky * DIL * G::XCOLS_D + kx * DIL;
            kstep(a_lane + a_off, b_lane + buf * BSTEP, acc, half == 0 ? NT0 : NT1);
            if (s_ == STEPS - 1 && c + 1 < n_chunks) {
                __syncthreads();                    // every wave is done with this chunk's patch
                a_store();
            }
            __syncthreads();
            buf ^= 1;
            ++ks;
        }
    }

    const int yy = y0 + row;
    if (yy < H) {
        float* ybc = y + (size_t)b * Cout * plane;
        const int co0 = cot * NT * 16;
        const float unscale = split_unscale_of(ex) * split_unscale_of(ew);
        const float* abc = addend ? addend + (size_t)b * Cout * plane : nullptr;
        if (half == 0) sb16_store<NT0, NT0>(acc, ybc, bias, abc, co0, plane, yy, x0, W, g, n, unscale);
        else if (NT1 > 0) sb16_store<NT1, NT0>(acc, ybc, bias, abc, co0 + NT0 * 16, plane, yy, x0, W, g, n, unscale);
        if (stats) {
            const size_t seg = ((size_t)b * H + yy) * tiles_x + tx;
            if (half == 0) cseg_stats_emit<NT0, NT0>(acc, bias, co0, unscale, x0, W, g, n, stats + (size_t)co0 * n_seg + seg, n_seg);
            else if (NT1 > 0)
                cseg_stats_emit<NT1, NT0>(acc, bias, co0 + NT0 * 16, unscale, x0, W, g, n, stats + (size_t)(co0 + NT0 * 16) * n_seg + seg, n_seg);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// Round 3: persistent form for the small-channel convolutions. The kernel above is LATENCY-bound on the branches (48 channels at
// 8x128x256: 52 us against 9 us of MFMA time and 20 us of HBM time): every one of its 15 K-steps ends in a barrier that waits for
// the LDS-DMA of the next step's weights, issued only one short step (~0.25 us of MFMAs) earlier, and a block lives for one
// 4x64 tile, so prologue (first patch from HBM) and epilogue overlap nothing when one block fits a CU.
// Here a block keeps ONE channel tile group and walks several spatial tiles:
//   * weights: the whole packed operator of the channel tile group stays RESIDENT in LDS when it fits (48 -> 48: 15 K-steps x
//     6 KB = 90 KB, loaded once per block by LDS-DMA), otherwise it is streamed one 16-channel chunk (5 K-steps) ahead into a
//     double buffer -- a whole chunk of MFMAs (~1.2 us) lies between the issue of a DMA and the barrier that waits for it;
//   * patch: double-buffered; the fp32 loads of the NEXT (tile, chunk) item are issued before the MFMAs of the current one and
//     split + stored into the other buffer after them -- also across tile boundaries, so a new tile starts without a prologue;
//   * ONE barrier per 16-channel chunk (5 K-steps) instead of one per K-step.
// Same packed-weight format (CSEG_PACK_C3_16), same fragment layout and epilogue as above. f16x3 only (with three pieces the
// buffers do not fit). CSEG_CONV3X3_SB16_P=0 switches back to the one-tile kernel.
// ---------------------------------------------------------------------------------------------------------
template <class AR, int NT, bool RES>
__global__ __launch_bounds__(512, 1) void conv3x3_sb16p_kernel(const float* __restrict__ x, const uint4* __restrict__ wp,
                                                               const float* __restrict__ bias, const float* __restrict__ addend, int Cin, int Cout, int H, int W,
                                                               int tiles_x, int tiles_y, int n_spatial, int groups,
                                                               const unsigned* __restrict__ amax_x,
                                                               const unsigned* __restrict__ amax_w, float* __restrict__ y,
                                                               float4* __restrict__ stats, int n_seg, int xmap) {
    extern __shared__ __attribute__((aligned(16))) uint4 smem_s16p[];
    constexpr int NP = AR::NP;
    constexpr int A_CELLS = NP * NOCT * PLANE;
    constexpr int BSTEP = NT * NP * 64;            // uint4 per K-step
    constexpr int BCHUNK = STEPS * BSTEP;          // uint4 per 16-channel chunk
    constexpr int NT0 = (NT + 1) / 2, NT1 = NT - NT0;
    uint4* As = smem_s16p;                         // [2][piece NP][octet 2][CELLS]
    uint4* Bs = smem_s16p + 2 * A_CELLS;           // RES: [n_chunks][BCHUNK]; else [2][BCHUNK]
    const unsigned ex = AR::SCALED ? split_amax_exp(amax_x) : 141u, ew = AR::SCALED ? split_amax_exp(amax_w) : 141u;
    const float xscale = split_scale_of(ex);
    const float unscale = split_unscale_of(ex) * split_unscale_of(ew);

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int row = wave & 3, half = wave >> 2;
    const int g = lane >> 4, n = lane & 15;
    const int n_cot = Cout / (NT * 16);
    const size_t plane = (size_t)H * W;
    // this block: channel tile group `cot`, spatial tiles t_first, t_first + t_step, ... < t_end. Plain order: tiles grp, grp + groups,
    // ... XCD-aware order (xmap, round 4; needs groups % 8 == 0): block b runs on XCD b % 8 (observed on gfx950; only speed depends
    // on it) -- the tile list is cut into 8 contiguous ranges, one per XCD, and the n_cot blocks of a tile sit on the SAME XCD: the
    // patch of a tile is fetched through the fabric once instead of n_cot times, halos of neighbouring tiles and the streamed
    // weights of a channel tile group meet in that XCD's L2.
    int cot = blockIdx.x % n_cot, grp = blockIdx.x / n_cot;
    int t_first = grp, t_step = groups, t_end = n_spatial;
    if (xmap) {
        const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3, per = (n_spatial + 7) / 8;
        cot = idx % n_cot;
        t_first = xcd * per + idx / n_cot;
        t_step = groups >> 3;
        t_end = min((xcd + 1) * per, n_spatial);
    }
    const int n_chunks = Cin / 16;
    const uint4* wbase = wp + (size_t)cot * n_chunks * BCHUNK;
    const int my_tiles = t_first < t_end ? (t_end - t_first + t_step - 1) / t_step : 0;
    const int n_items = my_tiles * n_chunks;
    if (n_items == 0) return;

    auto b_dma = [&](int chunk, uint4* dst) {      // one 16-channel chunk of packed weights: STEPS * NT * NP rows of 1 KB
        constexpr int ROWS = STEPS * NT * NP;
#pragma unroll
        for (int i = 0; i < (ROWS + 7) / 8; ++i) {
            const int r = wave + 8 * i;
            if (r < ROWS)
                __builtin_amdgcn_global_load_lds(
                    (const __attribute__((address_space(1))) void*)(wbase + (size_t)chunk * BCHUNK + r * 64 + lane),
                    (__attribute__((address_space(3))) void*)(dst + r * 64), 16, 0, 0);
        }
    };
    auto tile_of = [&](int tile, int& b, int& y0, int& x0) {      // tile = index of one of this block's spatial tiles
        int t = t_first + tile * t_step;
        const int tx = t % tiles_x; t /= tiles_x;
        const int ty = t % tiles_y;
        b = t / tiles_y;
        x0 = tx * TC; y0 = ty * TR;
    };
    // Staging items. What does not depend on the tile -- which (octet, patch row, patch column) a thread stages, the LDS cell it
    // writes -- is computed ONCE per block (round 4: the per-item divisions by CELLS / XCOLS and the three tile_of() calls per item
    // were a quarter of the kernel's VALU instructions, 2.9 per MFMA in the round-3 counters; VALU issue of any wave of a SIMD comes
    // on top of its MFMA time). Tile coordinates are tracked for the tile being computed (cb, cy0, cx0) and the one being staged
    // (sb, sy0, sx0), and recomputed only when an item starts a new tile.
    float apre[AU][8];
    int it_r[AU], it_col[AU], it_cell[AU], it_plane[AU];
    bool it_in[AU];
#pragma unroll
    for (int u = 0; u < AU; ++u) {
        const int item = tid + 512 * u;
        const int oct = item / CELLS, rc = item - oct * CELLS;
        it_r[u] = rc / XCOLS;
        it_col[u] = rc - it_r[u] * XCOLS;
        it_in[u] = oct < NOCT;
        it_cell[u] = oct * PLANE + rc;
        it_plane[u] = min(oct, NOCT - 1) * 8 * (int)plane;
    }
    int sb, sy0, sx0;                              // tile of the item being staged
    auto a_issue = [&](int it) {
        const int chunk = it % n_chunks;
        if (chunk == 0) tile_of(it / n_chunks, sb, sy0, sx0);
        const float* xc = x + ((size_t)sb * Cin + (size_t)chunk * 16) * plane;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)xc, 0, (int)(16 * plane * sizeof(float)),
                                                                            0x00020000);
#pragma unroll
        for (int u = 0; u < AU; ++u) {
            const int yc = min(max(sy0 + it_r[u] - 1, 0), H - 1), xcl = min(max(sx0 + it_col[u] - 1, 0), W - 1);
            const int off = (it_plane[u] + yc * W + xcl) * (int)sizeof(float);
#pragma unroll
            for (int j = 0; j < 8; ++j)
                apre[u][j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                                           rs, off, j * (int)plane * (int)sizeof(float), 0));
        }
    };
    auto a_store = [&](uint4* dst) {               // the item a_issue() fetched last (same tile coordinates)
#pragma unroll
        for (int u = 0; u < AU; ++u) {
            if (it_in[u]) {
                const int yy = sy0 + it_r[u] - 1, xx = sx0 + it_col[u] - 1;
                const bool ok = yy >= 0 && yy < H && xx >= 0 && xx < W;
                uint4 cells[NP];
                split_cells8_masked<AR>(apre[u], ok, xscale, cells);            // zero padding / outside the tensor
#pragma unroll
                for (int p = 0; p < NP; ++p) dst[p * NOCT * PLANE + it_cell[u]] = cells[p];
            }
        }
    };

    f32x4 acc[4][NT0];
    // ---- prologue: weights (all of them, or chunk 0), patch of item 0
    a_issue(0);
    if (RES) {
        for (int c = 0; c < n_chunks; ++c) b_dma(c, Bs + (size_t)c * BCHUNK);
    } else {
        b_dma(0, Bs);
    }
    a_store(As);
    __syncthreads();
    int cb = sb, cy0 = sy0, cx0 = sx0;             // tile of the item being computed

    const int a_lane_off = row * XCOLS + n;
    const int b_lane_off = (half ? NT0 * NP * 64 : 0) + lane;
#pragma unroll 1
    for (int it = 0; it < n_items; ++it) {
        const int chunk = it % n_chunks;
        const bool more = it + 1 < n_items;
        if (chunk == 0) {
            cb = sb; cy0 = sy0; cx0 = sx0;         // (the staged tile is still this item's: a_issue(it + 1) comes below)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT0; ++nt) acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        if (more) {
            a_issue(it + 1);                                             // fp32 loads of the next item fly under the MFMAs below
            if (!RES) b_dma((it + 1) % n_chunks, Bs + (size_t)((it + 1) & 1) * BCHUNK);   // that buffer was last read in item it - 1
        }
        const uint4* a_lane = As + (size_t)(it & 1) * A_CELLS + a_lane_off;
        const uint4* b_base = (RES ? Bs + (size_t)chunk * BCHUNK : Bs + (size_t)(it & 1) * BCHUNK) + b_lane_off;
#pragma unroll
        for (int s = 0; s < STEPS; ++s) {
            // the ninth tap is paired with a tenth that does not exist: its packed weights are zero
            const int tap = min(2 * s + (g >> 1), 8);
            const int ky = tap / 3, kx = tap - 3 * ky;
            const int a_off = (g & 1) * PLANE + ky * XCOLS + kx;
            if (half == 0) sb16_kstep<AR, NT0, NT0>(a_lane + a_off, b_base + s * BSTEP, acc);
            else if (NT1 > 0) sb16_kstep<AR, NT1, NT0>(a_lane + a_off, b_base + s * BSTEP, acc);
        }
        if (chunk == n_chunks - 1) {
            const int b = cb, y0 = cy0, x0 = cx0;
            const int yy = y0 + row;
            if (yy < H) {
                float* ybc = y + (size_t)b * Cout * plane;
                const float* abc = addend ? addend + (size_t)b * Cout * plane : nullptr;
                const int co0 = cot * NT * 16;
                if (half == 0) sb16_store<NT0, NT0>(acc, ybc, bias, abc, co0, plane, yy, x0, W, g, n, unscale);
                else if (NT1 > 0) sb16_store<NT1, NT0>(acc, ybc, bias, abc, co0 + NT0 * 16, plane, yy, x0, W, g, n, unscale);
                if (stats) {
                    const size_t seg = ((size_t)b * H + yy) * tiles_x + x0 / TC;
                    if (half == 0) cseg_stats_emit<NT0, NT0>(acc, bias, co0, unscale, x0, W, g, n, stats + (size_t)co0 * n_seg + seg, n_seg);
                    else if (NT1 > 0)
                        cseg_stats_emit<NT1, NT0>(acc, bias, co0 + NT0 * 16, unscale, x0, W, g, n, stats + (size_t)(co0 + NT0 * 16) * n_seg + seg,
                                                  n_seg);
                }
            }
        }
        if (more) a_store(As + (size_t)((it + 1) & 1) * A_CELLS);          // the other patch buffer: last read in item it - 1
        __syncthreads();
    }
}

// CUs a persistent launch spreads its blocks over (one block per CU). (Round 4 tried two branch kernels side by side on half the chip
// each: not faster than two full-chip launches back to back, DESIGN.md section 11.8 -- the switch is gone.)
constexpr long persist_cus() { return 256; }

// XCD-aware tile order of the persistent kernels: CSEG_SB16_XCD (default 1), possible when the tile groups divide over the 8 XCDs
int xcd_order(long groups) {
    const char* e = getenv("CSEG_SB16_XCD");
    return (!e || atoi(e) != 0) && groups % 8 == 0 ? 1 : 0;
}

template <class AR, int NT, bool RES>
int launch_sb16p(const float* x, const uint4* wp, const float* bias, const float* addend, int B, int Cin, int Cout, int H, int W,
                 const unsigned* amax_x, const unsigned* amax_w, float* y, float4* stats, size_t lds, hipStream_t stream) {
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void*)(conv3x3_sb16p_kernel<AR, NT, RES>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds) != hipSuccess) {
            cseg_set_error("conv3x3_sb16p: cannot raise dynamic LDS to %zu bytes", lds);
            return 0;
        }
        attr_set = true;
    }
    const int tiles_x = (W + TC - 1) / TC, tiles_y = (H + TR - 1) / TR;
    const int n_cot = Cout / (NT * 16);
    const long n_spatial = (long)B * tiles_y * tiles_x;
    CSEG_REQUIRE(n_spatial < 2147483647L, "conv3x3_sb16p: too many tiles");
    long groups = persist_cus() / n_cot;                       // one block per CU: 256 CUs shared by the channel tile groups
    if (groups < 1) groups = 1;
    if (groups > n_spatial) groups = n_spatial;
    hipLaunchKernelGGL((conv3x3_sb16p_kernel<AR, NT, RES>), dim3((unsigned)(groups * n_cot)), dim3(512), lds, stream, x, wp, bias,
                       addend, Cin, Cout, H, W, tiles_x, tiles_y, (int)n_spatial, (int)groups, amax_x, amax_w, y, stats, B * H * tiles_x,
                       xcd_order(groups));
    CSEG_CHECK_LAUNCH("conv3x3_sb16p_kernel");
    return 1;
}

// the persistent kernel when its buffers fit the 160 KB of LDS (f16x3); `lds` and `res` say how
bool sb16p_plan(int arith, int Cin, int NT, size_t& lds, bool& res) {
    const char* e = getenv("CSEG_CONV3X3_SB16_P");
    if (arith != CSEG_ARITH_F16X3 || (e && atoi(e) == 0)) return false;
    const size_t a = 2 * (size_t)2 * NOCT * PLANE * sizeof(uint4);               // two patch buffers, two pieces
    const size_t chunk = (size_t)STEPS * NT * 2 * 64 * sizeof(uint4);
    const size_t all = (size_t)(Cin / 16) * chunk;
    const size_t cap = 160 * 1024;
    if (a + all <= cap) { lds = a + all; res = true; return true; }
    // streamed weights: measured a win at 3 channel tiles per block (192 ch 47.4 vs 52.2 us, 384 ch 79.8 vs 95.7) and a loss at
    // 4 (64 ch: 83.2 vs 72.9 us for the one-tile kernel, which keeps two blocks per CU) -- tools/branch_conv_probe.py
    if (NT == 3 && a + 2 * chunk <= cap) { lds = a + 2 * chunk; res = false; return true; }
    return false;
}

// ---------------------------------------------------------------------------------------------------------
// Round 4, default where 8-row tiles still give every CU a block (CSEG_SB16_ROWS8=0 switches it off): the persistent kernel on
// 8 x 64-pixel tiles, ONE WAVE PER OUTPUT ROW x all three channel tiles. Measured (tools/probes/conv_probe, profiles/
// r04_rows8_probe.jsonl): 8 x 48 x 128 x 256 forward with statistics 52.0 -> 46.9 us, plain 48.8 -> 43.8, output bit-identical on
// the MI355X; at 192 / 384 channels (128 blocks of 8-row tiles) it would lose (60 vs 45, 106 vs 78 us) and is not taken. Why: on 4 x 64 tiles the two halves of a block split the CHANNEL
// tiles and both read the row's pixel fragments -- 22 ds_read_b128 per 36 MFMAs, 704 LDS cycles per K-step against 576 of MFMA
// issue; the K-steps of the 48-channel layers take 17-19 us where the MFMAs need 13. Here a wave reads its row's four pixel tiles
// and the three channel tiles once per K-step: 14 reads per 36 MFMAs (448 cycles), and the halo of a tile is 1.29 x instead of
// 1.55 x its pixels (fewer loads, fewer cells to split). LDS: the patch image of a 16-channel chunk is 43 KB, so
//   * weights RESIDENT (input channels <= 48: 3 x 30 KB) leave room for ONE patch buffer: two barriers per chunk, the split of the
//     next chunk is not hidden behind MFMAs -- it was not hidden before either (section 11.8: the phases add up);
//   * weights STREAMED (more input channels): two patch buffers + two weight chunks = 146 KB, one barrier per chunk as before.
// Same packed weights (CSEG_PACK_C3_16, three channel tiles per block), same accumulation order per output element as the 4-row
// kernels: bit-identical results. Statistics segments are rows x 64 columns as everywhere (cseg_stats.h).
// ---------------------------------------------------------------------------------------------------------
namespace r8 {
constexpr int TR8 = 8;
constexpr int XROWS8 = TR8 + 2;
constexpr int CELLS8 = XROWS8 * XCOLS;             // 660 pixels per (piece, octet)
constexpr int PLANE8 = 672;                        // LDS stride of a (piece, octet) plane: 0 mod 256 bytes
constexpr int A_ITEMS8 = NOCT * CELLS8;            // 1320 staging items per chunk
constexpr int AU8 = (A_ITEMS8 + 511) / 512;        // 3 per thread
constexpr int NT8 = 3;
}  // namespace r8

template <class AR, bool RES>
__global__ __launch_bounds__(512, 1) void conv3x3_sb16r_kernel(const float* __restrict__ x, const uint4* __restrict__ wp,
                                                               const float* __restrict__ bias, const float* __restrict__ addend, int Cin, int Cout, int H, int W,
                                                               int tiles_x, int tiles_y, int n_spatial, int groups,
                                                               const unsigned* __restrict__ amax_x,
                                                               const unsigned* __restrict__ amax_w, float* __restrict__ y,
                                                               float4* __restrict__ stats, int n_seg, int xmap) {
    using namespace r8;
    extern __shared__ __attribute__((aligned(16))) uint4 smem_s16r[];
    constexpr int NP = AR::NP;
    constexpr int A_CELLS = NP * NOCT * PLANE8;
    constexpr int BSTEP = NT8 * NP * 64;
    constexpr int BCHUNK = STEPS * BSTEP;
    constexpr int NBUF = RES ? 1 : 2;              // patch buffers
    uint4* As = smem_s16r;                         // [NBUF][piece NP][octet 2][PLANE8]
    uint4* Bs = smem_s16r + NBUF * A_CELLS;        // RES: [n_chunks][BCHUNK]; else [2][BCHUNK]
    const unsigned ex = AR::SCALED ? split_amax_exp(amax_x) : 141u, ew = AR::SCALED ? split_amax_exp(amax_w) : 141u;
    const float xscale = split_scale_of(ex);
    const float unscale = split_unscale_of(ex) * split_unscale_of(ew);

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int row = wave;                          // one output row of the tile per wave
    const int g = lane >> 4, n = lane & 15;
    const int n_cot = Cout / (NT8 * 16);
    const size_t plane = (size_t)H * W;
    const int cot = blockIdx.x % n_cot, grp = blockIdx.x / n_cot;
    const int n_chunks = Cin / 16;
    const uint4* wbase = wp + (size_t)cot * n_chunks * BCHUNK;
    // Which tiles a block walks. Plain order: tiles grp, grp + groups, ... -- the blocks running at any moment cover consecutive
    // tiles, but block b runs on XCD b % 8 (observed on gfx950; only speed depends on it), so neither the horizontal nor the
    // vertical neighbour of a tile (4 tiles further at 256 columns) shares its XCD's L2, and every halo row is fetched through the
    // fabric again. xmap (default; one channel tile group, groups % 8 == 0): the tile list is cut into 8 contiguous ranges, one per XCD, and
    // the 32 blocks of an XCD walk their range side by side -- halos of neighbouring tiles meet in that XCD's L2.
    int t_first = grp, t_step = groups, t_end = n_spatial;
    if (xmap) {                                    // (as in conv3x3_sb16p_kernel; this kernel runs with one channel tile group)
        const int xcd = grp & 7, idx = grp >> 3, per = (n_spatial + 7) / 8;
        t_first = xcd * per + idx;
        t_step = groups >> 3;
        t_end = min((xcd + 1) * per, n_spatial);
    }
    const int my_tiles = t_first < t_end ? (t_end - t_first + t_step - 1) / t_step : 0;
    const int n_items = my_tiles * n_chunks;
    if (n_items == 0) return;

    auto b_dma = [&](int chunk, uint4* dst) {
        constexpr int ROWS = STEPS * NT8 * NP;
#pragma unroll
        for (int i = 0; i < (ROWS + 7) / 8; ++i) {
            const int r = wave + 8 * i;
            if (r < ROWS)
                __builtin_amdgcn_global_load_lds(
                    (const __attribute__((address_space(1))) void*)(wbase + (size_t)chunk * BCHUNK + r * 64 + lane),
                    (__attribute__((address_space(3))) void*)(dst + r * 64), 16, 0, 0);
        }
    };
    auto tile_of = [&](int tile, int& b, int& y0, int& x0) {
        int t = t_first + tile * t_step;
        const int tx = t % tiles_x; t /= tiles_x;
        const int ty = t % tiles_y;
        b = t / tiles_y;
        x0 = tx * TC; y0 = ty * TR8;
    };
    float apre[AU8][8];
    int it_r[AU8], it_col[AU8], it_cell[AU8], it_plane[AU8];
    bool it_in[AU8];
#pragma unroll
    for (int u = 0; u < AU8; ++u) {
        const int item = min(tid + 512 * u, A_ITEMS8 - 1);
        const int oct = item / CELLS8, rc = item - oct * CELLS8;
        it_r[u] = rc / XCOLS;
        it_col[u] = rc - it_r[u] * XCOLS;
        it_in[u] = tid + 512 * u < A_ITEMS8;
        it_cell[u] = oct * PLANE8 + rc;
        it_plane[u] = oct * 8 * (int)plane;
    }
    int sb, sy0, sx0;                              // tile of the item being staged
    auto a_issue = [&](int it) {
        const int chunk = it % n_chunks;
        if (chunk == 0) tile_of(it / n_chunks, sb, sy0, sx0);
        const float* xc = x + ((size_t)sb * Cin + (size_t)chunk * 16) * plane;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)xc, 0, (int)(16 * plane * sizeof(float)),
                                                                            0x00020000);
#pragma unroll
        for (int u = 0; u < AU8; ++u) {
            const int yc = min(max(sy0 + it_r[u] - 1, 0), H - 1), xcl = min(max(sx0 + it_col[u] - 1, 0), W - 1);
            const int off = (it_plane[u] + yc * W + xcl) * (int)sizeof(float);
#pragma unroll
            for (int j = 0; j < 8; ++j)
                apre[u][j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                                           rs, off, j * (int)plane * (int)sizeof(float), 0));
        }
    };
    auto a_store = [&](uint4* dst) {               // the item a_issue() fetched last (same tile coordinates)
#pragma unroll
        for (int u = 0; u < AU8; ++u) {
            if (it_in[u]) {
                const int yy = sy0 + it_r[u] - 1, xx = sx0 + it_col[u] - 1;
                const bool ok = yy >= 0 && yy < H && xx >= 0 && xx < W;
                uint4 cells[NP];
                split_cells8_masked<AR>(apre[u], ok, xscale, cells);            // zero padding / outside the tensor
#pragma unroll
                for (int p = 0; p < NP; ++p) dst[p * NOCT * PLANE8 + it_cell[u]] = cells[p];
            }
        }
    };
    // one K-step of a wave: its row's four pixel tiles x three channel tiles x the piece products (sb16_kstep on the 8-row image)
    auto kstep = [&](const uint4* ap, const uint4* bp, f32x4 (&acc)[4][NT8]) {
        typedef typename AR::frag_t frag_t;
        frag_t a[4][NP];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int p = 0; p < NP; ++p) a[mt][p] = __builtin_bit_cast(frag_t, ap[p * NOCT * PLANE8 + 16 * mt]);
#pragma unroll
        for (int nt = 0; nt < NT8; ++nt) {
            frag_t b[NP];
#pragma unroll
            for (int p = 0; p < NP; ++p) b[p] = __builtin_bit_cast(frag_t, bp[(nt * NP + p) * 64]);
#pragma unroll
            for (int t = 0; t < AR::NTERMS; ++t)
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) acc[mt][nt] = AR::mfma(a[mt][AR::ta(t)], b[AR::tb(t)], acc[mt][nt]);
        }
    };

    f32x4 acc[4][NT8];
    a_issue(0);
    if (RES) {
        for (int c = 0; c < n_chunks; ++c) b_dma(c, Bs + (size_t)c * BCHUNK);
    } else {
        b_dma(0, Bs);
    }
    a_store(As);
    __syncthreads();
    int cb = sb, cy0 = sy0, cx0 = sx0;             // tile of the item being computed

    const int a_lane_off = row * XCOLS + n;
#pragma unroll 1
    for (int it = 0; it < n_items; ++it) {
        const int chunk = it % n_chunks;
        const bool more = it + 1 < n_items;
        if (chunk == 0) {
            cb = sb; cy0 = sy0; cx0 = sx0;         // (the staged tile is still this item's: a_issue(it + 1) comes below)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT8; ++nt) acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        if (more) {
            a_issue(it + 1);                                             // fp32 loads of the next item fly under the MFMAs below
            if (!RES) b_dma((it + 1) % n_chunks, Bs + (size_t)((it + 1) & 1) * BCHUNK);   // that buffer was last read in item it - 1
        }
        const uint4* a_lane = As + (RES ? 0 : (size_t)(it & 1) * A_CELLS) + a_lane_off;
        const uint4* b_base = (RES ? Bs + (size_t)chunk * BCHUNK : Bs + (size_t)(it & 1) * BCHUNK) + lane;
#pragma unroll
        for (int s = 0; s < STEPS; ++s) {
            const int tap = min(2 * s + (g >> 1), 8);                    // the tenth tap slot multiplies zero weights
            const int ky = tap / 3, kx = tap - 3 * ky;
            kstep(a_lane + (g & 1) * PLANE8 + ky * XCOLS + kx, b_base + s * BSTEP, acc);
        }
        if (chunk == n_chunks - 1) {
            const int yy = cy0 + row;
            if (yy < H) {
                float* ybc = y + (size_t)cb * Cout * plane;
                const float* abc = addend ? addend + (size_t)cb * Cout * plane : nullptr;
                const int co0 = cot * NT8 * 16;
                sb16_store<NT8, NT8>(acc, ybc, bias, abc, co0, plane, yy, cx0, W, g, n, unscale);
                if (stats) {
                    const size_t seg = ((size_t)cb * H + yy) * tiles_x + cx0 / TC;
                    cseg_stats_emit<NT8, NT8>(acc, bias, co0, unscale, cx0, W, g, n, stats + (size_t)co0 * n_seg + seg, n_seg);
                }
            }
        }
        if (RES) {
            __syncthreads();                       // every wave is done with the one patch image
            if (more) a_store(As);
        } else if (more) {
            a_store(As + (size_t)((it + 1) & 1) * A_CELLS);              // the other patch buffer: last read in item it - 1
        }
        __syncthreads();
    }
}

// (Round 5 tried a two-team form of this kernel -- four waves run the K-steps of their tile while the other four store outputs, split
// the next patch and issue loads, one barrier per phase, weights shared -- bit-identical output, NOT faster: 41.7 / 48.0 us against
// 40.9 / 44.6 at 8 x 48 x 128 x 256 (profiles/r05_two_team_probe.jsonl). The launch is the memory side's 29-32 us at ~4 TB/s plus
// ~10 us of launch / prologue / tail; the MFMAs were already hidden. The kernel left the library again: git history)
template <class AR, bool RES>
int launch_sb16r(const float* x, const uint4* wp, const float* bias, const float* addend, int B, int Cin, int Cout, int H, int W,
                 const unsigned* amax_x, const unsigned* amax_w, float* y, float4* stats, size_t lds, hipStream_t stream) {
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void*)(conv3x3_sb16r_kernel<AR, RES>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) !=
            hipSuccess) {
            cseg_set_error("conv3x3_sb16r: cannot raise dynamic LDS to %zu bytes", lds);
            return 0;
        }
        attr_set = true;
    }
    const int tiles_x = (W + TC - 1) / TC, tiles_y = (H + r8::TR8 - 1) / r8::TR8;
    const int n_cot = Cout / (r8::NT8 * 16);
    const long n_spatial = (long)B * tiles_y * tiles_x;
    CSEG_REQUIRE(n_spatial < 2147483647L, "conv3x3_sb16r: too many tiles");
    long groups = persist_cus() / n_cot;
    if (groups < 1) groups = 1;
    if (groups > n_spatial) groups = n_spatial;
    const int xmap = n_cot == 1 ? xcd_order(groups) : 0;
    hipLaunchKernelGGL((conv3x3_sb16r_kernel<AR, RES>), dim3((unsigned)(groups * n_cot)), dim3(512), lds, stream, x, wp, bias, addend,
                       Cin, Cout, H, W, tiles_x, tiles_y, (int)n_spatial, (int)groups, amax_x, amax_w, y, stats, B * H * tiles_x, xmap);
    CSEG_CHECK_LAUNCH("conv3x3_sb16r_kernel");
    return 1;
}

// the 8-row form: f16x3, three channel tiles per block, and only where it still gives every CU a block
bool sb16r_plan(int arith, int B, int Cin, int Cout, int H, int W, int NT, size_t& lds, bool& res) {
    const char* e = getenv("CSEG_SB16_ROWS8");
    const int mode = e ? atoi(e) : 1;                                    // 0: off, 1 (default): where it fills the chip, 2: wherever it fits
    if (mode == 0 || arith != CSEG_ARITH_F16X3 || NT != 3) return false;
    const long blocks = (long)B * ((H + 7) / 8) * ((W + TC - 1) / TC) * (Cout / 48);
    if (blocks < 256 && mode < 2) return false;
    const size_t a = (size_t)2 * NOCT * r8::PLANE8 * sizeof(uint4);             // one patch image, two pieces
    const size_t chunk = (size_t)STEPS * 3 * 2 * 64 * sizeof(uint4);
    const size_t all = (size_t)(Cin / 16) * chunk;
    const size_t cap = 160 * 1024;
    if (a + all <= cap) { lds = a + all; res = true; return true; }
    if (2 * a + 2 * chunk <= cap) { lds = 2 * a + 2 * chunk; res = false; return true; }
    return false;
}

template <class AR, int NT>
int launch_sb16(const float* x, const uint4* wp, const float* bias, const float* addend, int B, int Cin, int Cout, int H, int W,
                const unsigned* amax_x, const unsigned* amax_w, float* y, float4* stats, hipStream_t stream) {
    const size_t lds = sizeof(uint4) * (AR::NP * NOCT * PLANE + 2 * NT * AR::NP * 64);
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void*)conv3x3_sb16_kernel<AR, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) !=
            hipSuccess) {
            cseg_set_error("conv3x3_sb16: cannot raise dynamic LDS to %zu bytes", lds);
            return 0;
        }
        attr_set = true;
    }
    const int tiles_x = (W + TC - 1) / TC, tiles_y = (H + TR - 1) / TR;
    const long n_tiles = (long)B * (Cout / (NT * 16)) * tiles_y * tiles_x;
    CSEG_REQUIRE(n_tiles < 2147483647L, "conv3x3_sb16: grid too large");
    hipLaunchKernelGGL((conv3x3_sb16_kernel<AR, NT>), dim3((unsigned)n_tiles), dim3(512), lds, stream, x, wp, bias, addend, Cin, Cout, H, W,
                       tiles_x, tiles_y, amax_x, amax_w, y, stats, B * H * tiles_x, cseg_xcd_remap());
    CSEG_CHECK_LAUNCH("conv3x3_sb16_kernel");
    return 1;
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------
// 3x3 / stride 1 / pad 1 split f16x3 convolution, variant for the HEAD (720 -> 720 at 8 x 128 x 256; 44 % of the forward FLOPs of
// HRNet-W48-contrast, lib/models/nets/hrnet.py:72-77 of the reference): a block owns 8 rows x 64 columns x 144 output channels.
// Why: conv3x3_sb_kernel<9> moves 24 KB per K-step from L2 for a 4 x 64-pixel tile -- 18.4 KB of
// packed weights and 5.6 KB of patch for 1 824 MFMA cycles per SIMD = 13 B per cycle and CU, above the ~10 B/cycle/CU a CU
// sustains; with every global access removed it runs in 4.1 instead of 5.5 ms. Twice the pixels per block halve the weight bytes
// per MFMA (7 B/cycle/CU). What that costs and how it is paid:
//   * 144 accumulator registers per wave (one row = 4 pixel tiles x 9 channel tiles): little else may live in registers across
//     the MFMAs, so the fp32 patch does NOT travel through registers. It is brought in by LDS-DMA (global_load_lds, one dword per
//     lane, one patch row per instruction with a scalar base, the image border clamped) into a raw staging area while the MFMAs of
//     the previous chunk run,
//     and a short split phase between two chunks turns it into the [piece][octet][cell] image of 16-byte cells the fragments are
//     read from (8 strided ds_read_b32 -> packed split -> 2 ds_write_b128 per item; ~130 VALU instructions per wave and chunk
//     against 600 MFMAs);
//   * LDS: 16-channel chunks (pieces 43 KB + raw 42.2 KB + 2 weight stages 36.9 KB = 122 KB); the K-step pairs two taps x 16
//     channels like conv3x3_sb16.hip (same packed weights, CSEG_PACK_C3_16 with 9 channel tiles; the tenth tap slot is zero: 11 %
//     more MFMA slots than taps).
// Wave = one output row of the tile; per K-step a wave keeps its 4 pixel tiles' fragments x 2 pieces in registers and streams the
// nine channel tiles' weight fragments: 26 ds_read_b128 for 108 MFMAs. Weights: LDS-DMA, double-buffered, one K-step ahead,
// one barrier per K-step (1 920 MFMA cycles). f16x3 only.
// Selected with nt = CSEG_NT_SB8 through the cseg_conv3x3_split_* entry points (pack and forward must use the same nt).
namespace sb8 {

constexpr int R8 = 8, C8 = 64;                  // output rows / columns per block
constexpr int XR = R8 + 2, XC = C8 + 2;         // patch: 10 x 66 pixels, cell (0, 0) = pixel (y0 - 1, x0 - 1)
constexpr int CELLS8 = XR * XC;                 // 660
constexpr int PLANE8 = 672;                     // LDS stride of a (piece, octet) plane: 0 mod 256 bytes
constexpr int RAW_LINES = 16 * XR;              // 160 (channel, patch row) lines per chunk
constexpr int RAW_MAIN = RAW_LINES * 64;        // floats: patch columns 0 .. 63 of every line, one DMA instruction per line
constexpr int RAW_EXTRA = RAW_LINES * 2;        // patch columns 64, 65 of every line: 5 more instructions
constexpr int RAW_FLOATS = RAW_MAIN + RAW_EXTRA;
constexpr int S_ITEMS = 2 * CELLS8;             // (octet, cell) items of the split phase: 1 320
constexpr int S_U = (S_ITEMS + 511) / 512;      // 3 per thread
constexpr int NT = 9;
constexpr int STEPS = 5;

// One K-step of a wave: its row's four pixel tiles (fragments resident: 8 registers x 4) x all nine channel tiles (weight
// fragments streamed, two at a time) x three piece products; term-major, four independent accumulators between dependent MFMAs.
template <class AR>
__device__ __forceinline__ void kstep8(const uint4* __restrict__ ap, const uint4* __restrict__ bp, f32x4 (&acc)[4][NT]) {
    typedef typename AR::frag_t frag_t;
    frag_t a[4][AR::NP];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int p = 0; p < AR::NP; ++p) a[mt][p] = __builtin_bit_cast(frag_t, ap[p * 2 * PLANE8 + 16 * mt]);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        frag_t b[AR::NP];
#pragma unroll
        for (int p = 0; p < AR::NP; ++p) b[p] = __builtin_bit_cast(frag_t, bp[(nt * AR::NP + p) * 64]);
#pragma unroll
        for (int t = 0; t < AR::NTERMS; ++t)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) acc[mt][nt] = AR::mfma(a[mt][AR::ta(t)], b[AR::tb(t)], acc[mt][nt]);
    }
}

__device__ __forceinline__ void store8(const f32x4 (&acc)[4][NT], float* __restrict__ ybc, const float* __restrict__ bias, int co0,
                                       size_t plane, int yy, int x0, int W, int g, int n, float unscale) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const float bv = bias ? bias[co0 + nt * 16 + n] : 0.f;
        float* orow = ybc + (size_t)(co0 + nt * 16 + n) * plane + (size_t)yy * W;
        const bool vec = (W & 3) == 0;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            const int xx = x0 + 16 * mt + 4 * g;
            f32x4 v = acc[mt][nt] * unscale;
            v += bv;
            cseg_store_row4(orow, nullptr, xx, W, vec, v);
        }
    }
}

template <class AR>
__global__ __launch_bounds__(512, 1) void conv3x3_sb8_kernel(const float* __restrict__ x, const uint4* __restrict__ wp,
                                                             const float* __restrict__ bias, int Cin, int Cout, int H, int W,
                                                             int tiles_x, int tiles_y, const unsigned* __restrict__ amax_x,
                                                             const unsigned* __restrict__ amax_w, float* __restrict__ y,
                                                             float4* __restrict__ stats, int n_seg, int xmap) {
    extern __shared__ __attribute__((aligned(16))) uint4 smem_s8[];
    constexpr int NP = AR::NP;
    constexpr int A_CELLS = NP * 2 * PLANE8;
    constexpr int BSTEP = NT * NP * 64;
    uint4* As = smem_s8;                                       // [piece][octet 2][PLANE8]
    uint4* Bs = smem_s8 + A_CELLS;                             // [2][BSTEP]
    float* Raw = reinterpret_cast<float*>(Bs + 2 * BSTEP);     // [16 ch x 10 rows][64] + [16 ch x 10 rows][2]
    const unsigned ex = split_amax_exp(amax_x), ew = split_amax_exp(amax_w);
    const float xscale = split_scale_of(ex);

    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int row = wave;                          // one output row of the tile per wave
    const int g = lane >> 4, n = lane & 15;
    const int n_cot = Cout / (NT * 16);
    const size_t plane = (size_t)H * W;
    // plain order: column tile fastest, then row tile, channel tile group, image. XCD-aware order (opt-in): every XCD gets a
    // contiguous run of logical blocks with the CHANNEL TILE GROUP fastest -- the n_cot blocks that read one patch run side by side
    // on one XCD, followed by the neighbouring tiles that share its halo
    int t = cseg_xcd_block(blockIdx.x, gridDim.x, xmap);
    const bool cot_first = xmap && (gridDim.x & 7) == 0;
    int cot = 0;
    if (cot_first) { cot = t % n_cot; t /= n_cot; }
    const int tx = t % tiles_x; t /= tiles_x;
    const int ty = t % tiles_y; t /= tiles_y;
    if (!cot_first) { cot = t % n_cot; t /= n_cot; }
    const int b = t;
    const int x0 = tx * C8, y0 = ty * R8;
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

    // raw patch of one 16-channel chunk by LDS-DMA. Main part: instruction `line` (= channel * 10 + patch row, 160 per chunk)
    // fetches patch columns 0 .. 63 of that line, lane = column: scalar base (chunk, channel, row) + one per-lane column offset
    // that never changes -> no vector address arithmetic. Columns 64, 65 of all lines: five more instructions with a fixed
    // per-lane offset. Wave w issues lines w, w + 8, ... and extra instruction w (w < 5); the issue is spread over the five
    // K-steps of the previous chunk (`part` = 0 .. 4; part < 0: everything) so that no barrier waits for a whole chunk at once.
    // Border pixels are fetched from the nearest pixel inside the image (any finite value: the split phase writes zeros there).
    const int col_off = min(max(x0 - 1 + lane, 0), W - 1);
    int extra_off;                                 // element offset inside a chunk of this lane's extra pixel
    {
        const int line = min(32 * wave + (lane >> 1), RAW_LINES - 1), e = lane & 1;
        const int ch = line / XR, r = line - ch * XR;
        extra_off = ch * (int)plane + min(max(y0 - 1 + r, 0), H - 1) * W + min(max(x0 - 1 + 64 + e, 0), W - 1);
    }
    auto raw_dma = [&](int chunk, int part) {
        const float* xc = x + ((size_t)b * Cin + (size_t)chunk * 16) * plane;
#pragma unroll 1
        for (int i = (part < 0 ? 0 : part); i < RAW_LINES / 8; i += (part < 0 ? 1 : STEPS)) {
            const int line = wave + 8 * i;                              // uniform
            const int ch = line / XR, r = line - ch * XR;
            const float* rowp = xc + (size_t)ch * plane + (size_t)min(max(y0 - 1 + r, 0), H - 1) * W;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(rowp + col_off),
                                             (__attribute__((address_space(3))) void*)(Raw + 64 * line), 4, 0, 0);
        }
        if (wave < RAW_EXTRA / 64 && part <= 0)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(xc + extra_off),
                                             (__attribute__((address_space(3))) void*)(Raw + RAW_MAIN + 64 * wave), 4, 0, 0);
    };

    // split phase: item = (octet, cell); everything about an item is fixed for the block
    int s_raw[S_U], s_dst[S_U], s_chs[S_U];
    bool s_do[S_U], s_in[S_U];
#pragma unroll
    for (int u = 0; u < S_U; ++u) {
        const int item = tid + 512 * u, itc = min(item, S_ITEMS - 1);
        const int oct = itc / CELLS8, cell = itc - oct * CELLS8;
        const int r = cell / XC, c = cell - r * XC;
        // raw position of channel 0 of the octet and the stride to the next channel (main part / extra columns)
        s_raw[u] = c < 64 ? (oct * 8 * XR + r) * 64 + c : RAW_MAIN + (oct * 8 * XR + r) * 2 + (c - 64);
        s_chs[u] = c < 64 ? XR * 64 : XR * 2;
        s_dst[u] = oct * PLANE8 + cell;
        s_do[u] = item < S_ITEMS;
        const int yy = y0 - 1 + r, xx = x0 - 1 + c;
        s_in[u] = yy >= 0 && yy < H && xx >= 0 && xx < W;
    }
    auto split_phase = [&]() {
#pragma unroll
        for (int u = 0; u < S_U; ++u) {
            if (s_do[u]) {
                float v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = Raw[s_raw[u] + j * s_chs[u]];
                uint4 cells[NP];
                split_cells8_masked<AR>(v, s_in[u], xscale, cells);             // outside the image: zero pieces
#pragma unroll
                for (int p = 0; p < NP; ++p) As[p * 2 * PLANE8 + s_dst[u]] = cells[p];
            }
        }
    };

    f32x4 acc[4][NT];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

    raw_dma(0, -1);
    b_glds(0, 0);
    __syncthreads();                               // (the fence of the barrier waits for this wave's DMA; the barrier for everyone's)
    split_phase();
    __syncthreads();

    const uint4* a_lane = As + row * XC + n;
    const uint4* b_lane = Bs + lane;
    int ks = 0, buf = 0;
#pragma unroll 1
    for (int c = 0; c < n_chunks; ++c) {
#pragma unroll 1
        for (int s = 0; s < STEPS; ++s) {
            if (ks + 1 < n_steps) b_glds(ks + 1, buf ^ 1);              // that stage was last read in step ks - 1 (barrier since)
            if (c + 1 < n_chunks) raw_dma(c + 1, s);                    // a fifth of the next chunk's raw patch (Raw is free: split
                                                                        // phase of this chunk + barrier are behind us)
            const int tap = min(2 * s + (g >> 1), 8);                   // the tenth tap slot multiplies zero weights
            const int ky = tap / 3, kx = tap - 3 * ky;
            const int a_off = (g & 1) * PLANE8 + ky * XC + kx;
            kstep8<AR>(a_lane + a_off, b_lane + buf * BSTEP, acc);
            __syncthreads();
            buf ^= 1;
            ++ks;
        }
        if (c + 1 < n_chunks) {
            split_phase();                         // raw chunk c + 1 landed before the last barrier (its fence waited for the DMA)
            __syncthreads();
        }
    }

    float* ybc = y + (size_t)b * Cout * plane;
    const int co0 = cot * NT * 16;
    const float unscale = split_unscale_of(ex) * split_unscale_of(ew);
    if (y0 + row < H) {
        store8(acc, ybc, bias, co0, plane, y0 + row, x0, W, g, n, unscale);
        if (stats)
            cseg_stats_emit<NT, NT>(acc, bias, co0, unscale, x0, W, g, n,
                                    stats + (size_t)co0 * n_seg + ((size_t)b * H + y0 + row) * tiles_x + tx, n_seg);
    }
}

constexpr size_t lds_bytes() { return sizeof(uint4) * (2 * 2 * PLANE8 + 2 * NT * 2 * 64) + sizeof(float) * RAW_FLOATS; }      // 122 112

// ---- round 6, version 2 of the head kernel: the same tile, packed weights, LDS images and arithmetic (bit-identical results), with
// the three things taken out of the K-step that kept the matrix pipe idle at every one of its 225 barriers (both waves of a SIMD run
// the same code in the same phase, so whatever one of them does between two MFMA groups the other does too):
//   (1) the fragments of the NEXT K-step are in registers before the barrier: the weight stages form a ring of three (the stage of
//       K-step ks + 1 landed one barrier earlier, so its first fragment pair may be read during K-step ks) and the patch image does not
//       change inside a chunk -- one patch fragment of the next tap rides with each MFMA group, the next weight pair with the last.
//       Before: ten ds_read_b128 per wave right after every barrier = 80 KB through the CU's LDS port with nothing to multiply;
//   (2) inside a K-step the weight pair of group nt + 1 is requested before the MFMAs of group nt (two register pairs);
//   (3) the raw-patch DMA addresses: the 20 (channel, row) line offsets of a wave are the same for every chunk -- computed once (SGPRs)
//       instead of ~50 scalar instructions per line (64-bit divisions by 10) at the top of every K-step; the five K-steps of a chunk are
//       unrolled so that line, tap and register-set indices are compile-time constants.
// LDS: pieces 43 008 + 3 weight stages 55 296 + raw 42 240 + item geometry 6 144 = 146 688 bytes. CSEG_SB8_V=1 selects the version above.
template <class AR>
__global__ __launch_bounds__(512, 1) void conv3x3_sb8p_kernel(const float* __restrict__ x, const uint4* __restrict__ wp,
                                                              const float* __restrict__ bias, int Cin, int Cout, int H, int W,
                                                              int tiles_x, int tiles_y, const unsigned* __restrict__ amax_x,
                                                              const unsigned* __restrict__ amax_w, float* __restrict__ y,
                                                              float4* __restrict__ stats, int n_seg, int xmap) {
    extern __shared__ __attribute__((aligned(16))) uint4 smem_s8p[];
    typedef typename AR::frag_t frag_t;
    constexpr int NP = AR::NP;
    constexpr int A_CELLS = NP * 2 * PLANE8;
    constexpr int BSTEP = NT * NP * 64;
    constexpr int LPW = RAW_LINES / 8;                         // raw lines per wave and chunk: 20
    uint4* As = smem_s8p;                                      // [piece][octet 2][PLANE8]
    uint4* Bs = smem_s8p + A_CELLS;                            // [3][BSTEP]
    float* Raw = reinterpret_cast<float*>(Bs + 3 * BSTEP);     // [16 ch x 10 rows][64] + [16 ch x 10 rows][2]
    const unsigned ex = split_amax_exp(amax_x), ew = split_amax_exp(amax_w);
    const float xscale = split_scale_of(ex);

    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int row = wave;
    const int g = lane >> 4, n = lane & 15;
    const int n_cot = Cout / (NT * 16);
    const size_t plane = (size_t)H * W;
    int t = cseg_xcd_block(blockIdx.x, gridDim.x, xmap);       // (block order: see conv3x3_sb8_kernel)
    const bool cot_first = xmap && (gridDim.x & 7) == 0;
    int cot = 0;
    if (cot_first) { cot = t % n_cot; t /= n_cot; }
    const int tx = t % tiles_x; t /= tiles_x;
    const int ty = t % tiles_y; t /= tiles_y;
    if (!cot_first) { cot = t % n_cot; t /= n_cot; }
    const int b = t;
    const int x0 = tx * C8, y0 = ty * R8;
    const int n_chunks = Cin / 16;
    const int n_steps = n_chunks * STEPS;
    const uint4* wbase = wp + (size_t)cot * n_steps * BSTEP;

    auto b_glds = [&](int ks, int stage) {
#pragma unroll
        for (int i = 0; i < (NT * NP + 7) / 8; ++i) {
            const int r = wave + 8 * i;
            if (r < NT * NP)
                __builtin_amdgcn_global_load_lds(
                    (const __attribute__((address_space(1))) void*)(wbase + (size_t)ks * BSTEP + r * 64 + lane),
                    (__attribute__((address_space(3))) void*)(Bs + stage * BSTEP + r * 64), 16, 0, 0);
        }
    };

    // raw patch by LDS-DMA, as in version 1; the element offset of this wave's line i inside a chunk is wave-uniform and constant
    const int col_off = min(max(x0 - 1 + lane, 0), W - 1);
    int raw_off[LPW];
#pragma unroll
    for (int i = 0; i < LPW; ++i) {
        const int line = wave + 8 * i;
        const int ch = line / XR, r = line - ch * XR;
        raw_off[i] = __builtin_amdgcn_readfirstlane(ch * (int)plane + min(max(y0 - 1 + r, 0), H - 1) * W);
    }
    int extra_off;
    {
        const int line = min(32 * wave + (lane >> 1), RAW_LINES - 1), e = lane & 1;
        const int ch = line / XR, r = line - ch * XR;
        extra_off = ch * (int)plane + min(max(y0 - 1 + r, 0), H - 1) * W + min(max(x0 - 1 + 64 + e, 0), W - 1);
    }
    auto raw_line = [&](const float* xc, int i) {
        const float* rp = xc + raw_off[i];         // wave-uniform; pinned to scalar registers so that the 20 per-lane sums raw_off[i] + col_off are
#if defined(__AMDGCN__)                           // not hoisted out of the loop as 40 vector registers (they were, and spilled to scratch)
        asm volatile("" : "+s"(rp));
#endif
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(rp + col_off),
                                         (__attribute__((address_space(3))) void*)(Raw + 64 * (wave + 8 * i)), 4, 0, 0);
    };
    auto raw_extra = [&](const float* xc) {
        if (wave < RAW_EXTRA / 64)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(xc + extra_off),
                                             (__attribute__((address_space(3))) void*)(Raw + RAW_MAIN + 64 * wave), 4, 0, 0);
    };

    // split phase: as in version 1, but an item's geometry (raw position, piece cell, flags) is a dword in LDS instead of five registers
    // held across the MFMAs: bits 0..13 raw position of channel 0, 14..24 destination cell, 25 main part (channel stride 640, else 20),
    // 26 the pixel lies inside the image. Items past S_ITEMS are not stored (the thread's third item may not exist).
    unsigned* Geo = reinterpret_cast<unsigned*>(Raw + RAW_FLOATS);          // [S_U][512]
#pragma unroll
    for (int u = 0; u < S_U; ++u) {
        const int item = tid + 512 * u, itc = min(item, S_ITEMS - 1);
        const int oct = itc / CELLS8, cell = itc - oct * CELLS8;
        const int r = cell / XC, c = cell - r * XC;
        const int raw = c < 64 ? (oct * 8 * XR + r) * 64 + c : RAW_MAIN + (oct * 8 * XR + r) * 2 + (c - 64);
        const int yy = y0 - 1 + r, xx = x0 - 1 + c;
        const bool in = yy >= 0 && yy < H && xx >= 0 && xx < W;
        Geo[u * 512 + tid] = (unsigned)raw | ((unsigned)(oct * PLANE8 + cell) << 14) | (c < 64 ? 1u << 25 : 0u) | (in ? 1u << 26 : 0u);
    }
    auto split_phase = [&]() {
#pragma unroll
        for (int u = 0; u < S_U; ++u) {
            if (tid + 512 * u < S_ITEMS) {
                const unsigned geo = Geo[u * 512 + tid];
                const int raw = geo & 0x3fff, dst = (geo >> 14) & 0x7ff, chs = (geo >> 25) & 1 ? XR * 64 : XR * 2;
                float v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = Raw[raw + j * chs];
                uint4 cells[NP];
                split_cells8_masked<AR>(v, (geo >> 26) & 1, xscale, cells);
#pragma unroll
                for (int p = 0; p < NP; ++p) As[p * 2 * PLANE8 + dst] = cells[p];
            }
        }
    };

    f32x4 acc[4][NT];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // per-lane cell offset of K-step s of a chunk: lanes g = 0, 1 read tap 2 s, lanes g = 2, 3 tap 2 s + 1 (the tenth slot: zero weights).
    // Two compile-time constants per K-step and one select (not five registers held across the MFMAs).
    const bool tap_hi = (g >> 1) != 0;
    auto a_off = [&](int s) {
        const int t0 = 2 * s, t1 = min(2 * s + 1, 8);
        return tap_hi ? (t1 / 3) * XC + (t1 % 3) : (t0 / 3) * XC + (t0 % 3);
    };
    const uint4* a_lane = As + (g & 1) * PLANE8 + row * XC + n;
    const uint4* b_lane = Bs + lane;

    frag_t a[2][4][NP];            // patch fragments of this K-step (set s & 1) and of the next
    frag_t bf[2][NP];              // weight fragments of this group and of the next
    int st = 0;                    // ring stage of the K-step about to run

    auto fresh = [&]() {           // first K-step of a chunk: nothing could be read ahead (the split phase has just rewritten the pieces)
        const uint4* ap = a_lane + a_off(0);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int p = 0; p < NP; ++p) a[0][mt][p] = __builtin_bit_cast(frag_t, ap[p * 2 * PLANE8 + 16 * mt]);
        const uint4* bp = b_lane + st * BSTEP;
#pragma unroll
        for (int p = 0; p < NP; ++p) bf[0][p] = __builtin_bit_cast(frag_t, bp[p * 64]);
    };

    const float* xc = x + (size_t)b * Cin * plane;             // chunk 0 of this image
#pragma unroll
    for (int i = 0; i < LPW; ++i) raw_line(xc, i);
    raw_extra(xc);
    b_glds(0, 0);
    b_glds(1, 1);                                              // (n_steps >= 5)
    __syncthreads();                                           // (the fence of the barrier waits for this wave's DMA; the barrier for everyone's)
    split_phase();
    __syncthreads();
    fresh();

    int ks = 0;
#pragma unroll 1
    for (int c = 0; c < n_chunks; ++c) {
        const bool more = c + 1 < n_chunks;
        const float* xn = xc + 16 * plane;                     // next chunk's raw patch
        auto kstep = [&](auto S_) {
            constexpr int S = decltype(S_)::value;
            constexpr int P0 = S & 1;
            constexpr bool PF = S + 1 < STEPS;
            const int st1 = st == 2 ? 0 : st + 1, st2 = st1 == 2 ? 0 : st1 + 1;
            const uint4* bp = b_lane + st * BSTEP;
            const uint4* bpn = b_lane + st1 * BSTEP;
            const uint4* apn = a_lane + a_off(PF ? S + 1 : S);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int cur = (P0 + nt) & 1;
                if (nt + 1 < NT) {
#pragma unroll
                    for (int p = 0; p < NP; ++p) bf[cur ^ 1][p] = __builtin_bit_cast(frag_t, bp[((nt + 1) * NP + p) * 64]);
                } else if (PF) {
#pragma unroll
                    for (int p = 0; p < NP; ++p) bf[cur ^ 1][p] = __builtin_bit_cast(frag_t, bpn[p * 64]);
                }
                if (PF && nt < 4 * NP)
                    a[P0 ^ 1][nt / NP][nt % NP] = __builtin_bit_cast(frag_t, apn[(nt % NP) * 2 * PLANE8 + 16 * (nt / NP)]);
                if (nt == 1) {                                 // DMA issue under the first MFMA groups: stage ks + 2, a fifth of the next raw patch
                    if (ks + 2 < n_steps) b_glds(ks + 2, st2);
                    if (more) {
#pragma unroll
                        for (int j = 0; j < LPW / STEPS; ++j) raw_line(xn, S + STEPS * j);
                        if (S == 0) raw_extra(xn);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int tt = 0; tt < AR::NTERMS; ++tt)
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt) acc[mt][nt] = AR::mfma(a[P0][mt][AR::ta(tt)], bf[cur][AR::tb(tt)], acc[mt][nt]);
                __builtin_amdgcn_sched_barrier(0);
            }
            __syncthreads();
            st = st1;
            ++ks;
        };
        kstep(std::integral_constant<int, 0>());
        kstep(std::integral_constant<int, 1>());
        kstep(std::integral_constant<int, 2>());
        kstep(std::integral_constant<int, 3>());
        kstep(std::integral_constant<int, 4>());
        if (more) {
            split_phase();                         // raw chunk c + 1 landed before the last barrier (its fence waited for the DMA)
            __syncthreads();
            fresh();
        }
        xc = xn;
    }

    float* ybc = y + (size_t)b * Cout * plane;
    const int co0 = cot * NT * 16;
    const float unscale = split_unscale_of(ex) * split_unscale_of(ew);
    if (y0 + row < H) {
        store8(acc, ybc, bias, co0, plane, y0 + row, x0, W, g, n, unscale);
        if (stats)
            cseg_stats_emit<NT, NT>(acc, bias, co0, unscale, x0, W, g, n,
                                    stats + (size_t)co0 * n_seg + ((size_t)b * H + y0 + row) * tiles_x + tx, n_seg);
    }
}

constexpr size_t lds_bytes_p() { return sizeof(uint4) * (2 * 2 * PLANE8 + 3 * NT * 2 * 64) + sizeof(float) * RAW_FLOATS + sizeof(unsigned) * S_U * 512; }    // 146 688

}  // namespace sb8

// Reached from the cseg_conv3x3_sb_* / cseg_conv3x3_split_* entry points of conv3x3_sb.hip (NT = 3, 4, 6; 9 = the 8-row head kernel).
namespace cseg_sb16 {

size_t packed_bytes(int arith, int Cin, int Cout) {
    return (size_t)(Cout / 16) * steps16(Cin) * (arith == CSEG_ARITH_F16X3 ? 2 : 3) * 64 * sizeof(uint4);
}

int pack(const float* w, int Cout, int Cin, int transpose_flip, int NT, int arith, const unsigned* amax_w, void* wp,
         hipStream_t stream) {
    const int conv_in = transpose_flip ? Cout : Cin, conv_out = transpose_flip ? Cin : Cout;
    CSEG_REQUIRE((NT == 3 || NT == 4 || NT == 6 || NT == 9) && conv_out % (NT * 16) == 0 && conv_in % 16 == 0,
                 "conv3x3_sb16: needs 3, 4, 6 or 9 channel tiles per block (got %d) and input channels %% 16 == 0", NT);
    const long total = (long)(conv_out / 16) * steps16(conv_in) * 64;
    CSEG_REQUIRE(total < 2147483647L, "conv3x3_sb16 pack: too large");
    if (arith == CSEG_ARITH_F16X3)
        hipLaunchKernelGGL(pack_weights_sb16_kernel<SplitF16x3>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, w, Cout,
                           Cin, transpose_flip, NT, amax_w, (uint4*)wp, (int)total);
    else
        hipLaunchKernelGGL(pack_weights_sb16_kernel<SplitBF16x6>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, w, Cout,
                           Cin, transpose_flip, NT, amax_w, (uint4*)wp, (int)total);
    CSEG_CHECK_LAUNCH("conv3x3_sb16 pack");
    return 1;
}

int fwd(const float* x, const void* wp, const float* bias, const float* addend, int B, int Cin, int Cout, int H, int W, int NT, int arith,
        const unsigned* amax_x, const unsigned* amax_w, float* y, float4* stats, hipStream_t stream) {
    CSEG_REQUIRE((NT == 3 || NT == 4 || NT == 6) && Cout % (NT * 16) == 0 && Cin % 16 == 0 && (long)H * W * 16 * 4 < 2147483647L,
                 "conv3x3_sb16: unsupported shape Cin=%d Cout=%d %dx%d with %d channel tiles per block", Cin, Cout, H, W, NT);
    const uint4* wq = (const uint4*)wp;
    size_t lds = 0;
    bool res = false;
    if (sb16r_plan(arith, B, Cin, Cout, H, W, NT, lds, res))
        return res ? launch_sb16r<SplitF16x3, true>(x, wq, bias, addend, B, Cin, Cout, H, W, amax_x, amax_w, y, stats, lds, stream)
                   : launch_sb16r<SplitF16x3, false>(x, wq, bias, addend, B, Cin, Cout, H, W, amax_x, amax_w, y, stats, lds, stream);
    if (sb16p_plan(arith, Cin, NT, lds, res)) {
#define SB16P(N)                                                                                                              \
    return res ? launch_sb16p<SplitF16x3, N, true>(x, wq, bias, addend, B, Cin, Cout, H, W, amax_x, amax_w, y, stats, lds, stream)          \
               : launch_sb16p<SplitF16x3, N, false>(x, wq, bias, addend, B, Cin, Cout, H, W, amax_x, amax_w, y, stats, lds, stream);
        if (NT == 3) { SB16P(3) }
        if (NT == 4) { SB16P(4) }
        if (NT == 6) { SB16P(6) }
#undef SB16P
    }
    if (arith == CSEG_ARITH_F16X3) {
        if (NT == 6) return launch_sb16<SplitF16x3, 6>(x, wq, bias, addend, B, Cin, Cout, H, W, amax_x, amax_w, y, stats, stream);
        if (NT == 4) return launch_sb16<SplitF16x3, 4>(x, wq, bias, addend, B, Cin, Cout, H, W, amax_x, amax_w, y, stats, stream);
        return launch_sb16<SplitF16x3, 3>(x, wq, bias, addend, B, Cin, Cout, H, W, amax_x, amax_w, y, stats, stream);
    }
    if (NT == 6) return launch_sb16<SplitBF16x6, 6>(x, wq, bias, addend, B, Cin, Cout, H, W, amax_x, amax_w, y, stats, stream);
    if (NT == 4) return launch_sb16<SplitBF16x6, 4>(x, wq, bias, addend, B, Cin, Cout, H, W, amax_x, amax_w, y, stats, stream);
    return launch_sb16<SplitBF16x6, 3>(x, wq, bias, addend, B, Cin, Cout, H, W, amax_x, amax_w, y, stats, stream);
}

// dilated form (conv3x3_sb16d_kernel): f16x3, dilation 2 or 4
template <int NT, int DIL>
int launch_sb16d(const float* x, const uint4* wp, const float* bias, const float* addend, int B, int Cin, int Cout, int H, int W,
                 const unsigned* amax_x, const unsigned* amax_w, float* y, float4* stats, hipStream_t stream) {
    typedef DilGeom<DIL> G;
    const size_t lds = sizeof(uint4) * (2 * NOCT * G::PLANE_D + 2 * NT * 2 * 64);
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void*)conv3x3_sb16d_kernel<SplitF16x3, NT, DIL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) !=
            hipSuccess) {
            cseg_set_error("conv3x3_sb16d: cannot raise dynamic LDS to %zu bytes", lds);
            return 0;
        }
        attr_set = true;
    }
    const int tiles_x = (W + TC - 1) / TC, tiles_y = (H + TR - 1) / TR;
    const long n_tiles = (long)B * (Cout / (NT * 16)) * tiles_y * tiles_x;
    CSEG_REQUIRE(n_tiles < 2147483647L, "conv3x3_sb16d: grid too large");
    hipLaunchKernelGGL((conv3x3_sb16d_kernel<SplitF16x3, NT, DIL>), dim3((unsigned)n_tiles), dim3(512), lds, stream, x, wp, bias, addend, Cin,
                       Cout, H, W, tiles_x, tiles_y, amax_x, amax_w, y, stats, B * H * tiles_x, cseg_xcd_remap());
    CSEG_CHECK_LAUNCH("conv3x3_sb16d_kernel");
    return 1;
}

int fwd_dil(const float* x, const void* wp, const float* bias, const float* addend, int B, int Cin, int Cout, int H, int W, int NT, int dil,
            int arith, const unsigned* amax_x, const unsigned* amax_w, float* y, float4* stats, hipStream_t stream) {
    CSEG_REQUIRE(arith == CSEG_ARITH_F16X3 && amax_x && amax_w, "conv3x3 dilated: f16x3 only (needs max|x| and max|w|)");
    CSEG_REQUIRE((dil == 2 || dil == 4) && (NT == 3 || NT == 4 || NT == 6) && Cout % (NT * 16) == 0 && Cin % 16 == 0 &&
                     (long)H * W * 16 * 4 < 2147483647L,
                 "conv3x3 dilated: unsupported dilation %d / shape Cin=%d Cout=%d %dx%d with %d channel tiles per block", dil, Cin, Cout, H, W, NT);
    const uint4* wq = (const uint4*)wp;
#define SB16D(N)                                                                                                                     \
    return dil == 2 ? launch_sb16d<N, 2>(x, wq, bias, addend, B, Cin, Cout, H, W, amax_x, amax_w, y, stats, stream)                  \
                    : launch_sb16d<N, 4>(x, wq, bias, addend, B, Cin, Cout, H, W, amax_x, amax_w, y, stats, stream);
    if (NT == 6) { SB16D(6) }
    if (NT == 4) { SB16D(4) }
    SB16D(3)
#undef SB16D
}

// the 8-row head kernel (namespace sb8 above): f16x3, 9 channel tiles per block, weights packed by pack(..., NT = 9, ...)
int fwd8(const float* x, const void* wp, const float* bias, int B, int Cin, int Cout, int H, int W, int arith, const unsigned* amax_x,
         const unsigned* amax_w, float* y, float4* stats, hipStream_t stream) {
    CSEG_REQUIRE(arith == CSEG_ARITH_F16X3 && amax_x && amax_w, "conv3x3_sb8: f16x3 only (needs max|x| and max|w|)");
    CSEG_REQUIRE(Cout % 144 == 0 && Cin % 16 == 0 && (long)H * W * 16 * 4 < 2147483647L,
                 "conv3x3_sb8: unsupported shape Cin=%d Cout=%d %dx%d (needs Cout %% 144, Cin %% 16)", Cin, Cout, H, W);
    const char* ver = getenv("CSEG_SB8_V");          // (read per call: tests switch it inside one process)
    const int version = ver ? atoi(ver) : 2;
    const int tiles_x = (W + sb8::C8 - 1) / sb8::C8, tiles_y = (H + sb8::R8 - 1) / sb8::R8;
    const long n_tiles = (long)B * (Cout / 144) * tiles_y * tiles_x;
    CSEG_REQUIRE(n_tiles < 2147483647L, "conv3x3_sb8: grid too large");
    if (version != 1) {
        const size_t lds = sb8::lds_bytes_p();
        static bool attr_set_p = false;
        if (!attr_set_p) {
            if (hipFuncSetAttribute((const void*)sb8::conv3x3_sb8p_kernel<SplitF16x3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) !=
                hipSuccess) {
                cseg_set_error("conv3x3_sb8p: cannot raise dynamic LDS to %zu bytes", lds);
                return 0;
            }
            attr_set_p = true;
        }
        hipLaunchKernelGGL(sb8::conv3x3_sb8p_kernel<SplitF16x3>, dim3((unsigned)n_tiles), dim3(512), lds, stream, x, (const uint4*)wp, bias, Cin,
                           Cout, H, W, tiles_x, tiles_y, amax_x, amax_w, y, stats, B * H * tiles_x, cseg_xcd_remap());
        CSEG_CHECK_LAUNCH("conv3x3_sb8p_kernel");
        return 1;
    }
    const size_t lds = sb8::lds_bytes();
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void*)sb8::conv3x3_sb8_kernel<SplitF16x3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) !=
            hipSuccess) {
            cseg_set_error("conv3x3_sb8: cannot raise dynamic LDS to %zu bytes", lds);
            return 0;
        }
        attr_set = true;
    }
    hipLaunchKernelGGL(sb8::conv3x3_sb8_kernel<SplitF16x3>, dim3((unsigned)n_tiles), dim3(512), lds, stream, x, (const uint4*)wp, bias, Cin,
                       Cout, H, W, tiles_x, tiles_y, amax_x, amax_w, y, stats, B * H * tiles_x, cseg_xcd_remap());
    CSEG_CHECK_LAUNCH("conv3x3_sb8_kernel");
    return 1;
}

}  // namespace cseg_sb16
