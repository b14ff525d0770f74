// 1x1 / stride 1 convolution, NCHW fp32 in and out, on the BF16 matrix cores with split operands ("bf16x6": see
// conv3x3_sb.hip for the arithmetic and the evidence that it stays in the fp32 rounding class).
// Reference sites: the projection head (lib/models/modules/projection.py:8-24: 720 -> 720 -> 256) and the 1x1 layers of
// the encoder; today they run on rocBLAS / MIOpen fp32 kernels (720 -> 720 at 8x128x256: 2.4 ms forward, 2.4 ms
// backward-data, profiles/r02_conv_layout_probe_nchw_vs_channels_last.jsonl).
//
// GEMM per image: M = pixels (a flat run of 256 pixels of the H*W plane per block), N = output channels (NT*16 per
// block), K = input channels, v_mfma_f32_16x16x32_bf16. Same skeleton as conv3x3_sb_kernel without the spatial part:
//   8 waves = 4 pixel quarters (64 pixels = 4 MFMA row tiles) x 2 halves of the NT channel tiles;
//   K-step = 32 input channels (a trailing 16-channel chunk runs as a K-step whose upper half multiplies zero weights);
//   A: the fp32 [32 ch][256 px] slab is split while it is staged into an LDS image [piece][channel octet][pixel] of
//      16-byte cells, DOUBLE-buffered (48 KB each): the loads of slab k+1 are issued before the MFMAs of K-step k and
//      split + stored after them, one barrier per K-step;
//   B: pre-split, pre-packed weights streamed one K-step ahead by LDS-DMA into a double-buffered stage.
// Backward-data = the same kernel on the transposed packing.
// Status: index-checked against a numpy lane model; first hardware run pending -> the host side keeps it opt-in
// (kernels.CONV1X1_SPLIT_BF16).
// Round 3: default on the GPU (720 -> 720: 1.5-1.65 vs 2.2 ms on rocBLAS in the round-2 driver pass); written against the
// arithmetic traits of cseg_split.h (bf16x6 and f16x3: two scaled fp16 pieces, three MFMAs per product).
#include "cseg_pack.h"
#include "cseg_stats.h"

namespace {

constexpr int MT_PX = 256;                  // pixels per block

__host__ __device__ constexpr int steps1(int Cin) { return pack_steps_c1(Cin); }

// Packed weights: Wp[co_tile][kstep][nt][piece][lane] of uint4 (8 bf16, element j), lane = 16*g + n:
//   value(co = (co_tile*NT + nt)*16 + n, ci = 32*kstep + 8g + j), zero beyond the channel count.
// w is the forward's [Cout, Cin]; transpose = 1 packs the backward-data operator (maps Cout -> Cin channels).
template <class AR>
__global__ __launch_bounds__(256) void pack_weights_1x1_kernel(const float* __restrict__ w, int Cout, int Cin, int transpose,
                                                               int NT, const unsigned* __restrict__ amax_w,
                                                               uint4* __restrict__ wp, int total) {
    const float wscale = AR::SCALED ? split_scale_of(split_amax_exp(amax_w)) : 1.f;      // every thread (shuffles inside)
    const int e = blockIdx.x * 256 + threadIdx.x;          // one thread per (co_tile, kstep, nt, lane)
    if (e >= total) return;
    pack_elem_c1<AR>(w, Cout, Cin, transpose, NT, wscale, wp, e);
}

template <class AR, int NTW, int NTMAX>
__device__ __forceinline__ void o_kstep(const uint4* __restrict__ ap, const uint4* __restrict__ bp, f32x4 (&acc)[4][NTMAX]) {
    typedef typename AR::frag_t frag_t;
    frag_t a[4][AR::NP];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int p = 0; p < AR::NP; ++p) a[mt][p] = __builtin_bit_cast(frag_t, ap[p * 4 * MT_PX + 16 * mt]);
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

// accumulator layout: D[m = 4*g + r][n]: pixel px0 + 16*mt + 4*g + r, channel co0 + 16*nt + n
template <int NTW, int NTMAX>
__device__ __forceinline__ void o_store(const f32x4 (&acc)[4][NTMAX], float* __restrict__ ybc, const float* __restrict__ bias,
                                        int co0, size_t plane, int px0, int g, int n, float unscale,
                                        const float* __restrict__ abc = nullptr) {
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) {
        float* orow = ybc + (size_t)(co0 + nt * 16 + n) * plane;
        const float* arow = abc ? abc + (size_t)(co0 + nt * 16 + n) * plane : nullptr;      // y = conv + bias + addend (same layout as y)
        const float bv = bias ? bias[co0 + nt * 16 + n] : 0.f;
        const bool vec = (plane & 3) == 0;          // channel planes 16-byte aligned (else element by element: cseg_store_row4)
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            const long px = (long)px0 + 16 * mt + 4 * g;
            f32x4 v = acc[mt][nt] * unscale;
            v += bv;
            cseg_store_row4(orow, arow, px, (long)plane, vec, v);
        }
    }
}

template <class AR, int NT>
__global__ __launch_bounds__(512, 1) void conv1x1_sb_kernel(const float* __restrict__ x, const uint4* __restrict__ wp,
                                                            const float* __restrict__ bias, int Cin, int Cout, int plane_i,
                                                            int tiles_p, const unsigned* __restrict__ amax_x,
                                                            const unsigned* __restrict__ amax_w, float* __restrict__ y,
                                                            float4* __restrict__ stats, int n_seg, int xmap,
                                                            const float* __restrict__ addend) {
    extern __shared__ __attribute__((aligned(16))) uint4 smem_o[];
    constexpr int NP = AR::NP;
    constexpr int A1_CELLS = NP * 4 * MT_PX;       // one A buffer: [piece][octet][pixel]
    uint4* As = smem_o;                            // [2][piece NP][octet 4][MT_PX]
    uint4* Bs = smem_o + 2 * A1_CELLS;             // [2][NT*NP*64]
    constexpr int BSTEP = NT * NP * 64;
    const unsigned ex = AR::SCALED ? split_amax_exp(amax_x) : 141u, ew = AR::SCALED ? split_amax_exp(amax_w) : 141u;
    const float xscale = split_scale_of(ex);       // 1 for the unscaled arithmetic
    constexpr int NT0 = (NT + 1) / 2, NT1 = NT - NT0;

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int quarter = wave & 3, half = wave >> 2;
    const int g = lane >> 4, n = lane & 15;
    const int n_cot = Cout / (NT * 16);
    const size_t plane = (size_t)plane_i;
    // (orders as in conv3x3_sb.hip: plain = pixel tile fastest; XCD-aware, opt-in = contiguous runs per XCD, channel tile group fastest:
    // a 1x1 convolution has no halo, what its blocks share is the input tile of the n_cot groups)
    int t = cseg_xcd_block(blockIdx.x, gridDim.x, xmap);
    const bool cot_first = xmap && (gridDim.x & 7) == 0;
    int cot = 0;
    if (cot_first) { cot = t % n_cot; t /= n_cot; }
    const int tp = t % tiles_p; t /= tiles_p;
    if (!cot_first) { cot = t % n_cot; t /= n_cot; }
    const int b = t;
    const int px0 = tp * MT_PX;
    const int n_steps = steps1(Cin);
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

    // A staging: item = (octet, pixel): 4 x 256 = 1024 items, two per thread; 8 channel loads each, coalesced along pixels
    float apre[2][8];
    auto a_issue = [&](int ks) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int item = tid + 512 * u;
            const int oct = item >> 8, p = item & 255;
            const int c0 = ks * 32 + oct * 8;
            const size_t px = min((size_t)px0 + p, plane - 1);
            const float* src = x + ((size_t)b * Cin) * plane + px;
#pragma unroll
            for (int j = 0; j < 8; ++j) apre[u][j] = src[(size_t)min(c0 + j, Cin - 1) * plane];     // raw; masked below
        }
    };
    auto a_store = [&](int ks, int buf) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int item = tid + 512 * u;
            const int oct = item >> 8, p = item & 255;
            const int c0 = ks * 32 + oct * 8;
            const bool px_ok = (size_t)px0 + p < plane;
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = (px_ok && c0 + j < Cin) ? apre[u][j] : 0.f;
            uint4 cells[NP];
            split_cells8<AR>(v, xscale, cells);
            uint4* dst = As + buf * A1_CELLS + oct * MT_PX + p;
#pragma unroll
            for (int q = 0; q < NP; ++q) dst[q * 4 * MT_PX] = cells[q];
        }
    };

    f32x4 acc[4][NT0];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT0; ++nt) acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

    a_issue(0);
    b_glds(0, 0);
    a_store(0, 0);
    __syncthreads();

    const uint4* a_lane = As + g * MT_PX + quarter * 64 + n;            // + buffer offset per K-step
    const uint4* b_lane = Bs + (half ? NT0 * NP * 64 : 0) + lane;
#pragma unroll 1
    for (int ks = 0; ks < n_steps; ++ks) {
        const int buf = ks & 1;
        const bool more = ks + 1 < n_steps;
        if (more) {
            b_glds(ks + 1, buf ^ 1);               // both stages of the other buffer were last read in step ks - 1
            a_issue(ks + 1);
        }
        if (half == 0) o_kstep<AR, NT0, NT0>(a_lane + buf * A1_CELLS, b_lane + buf * BSTEP, acc);
        else if (NT1 > 0) o_kstep<AR, NT1, NT0>(a_lane + buf * A1_CELLS, b_lane + buf * BSTEP, acc);
        if (more) a_store(ks + 1, buf ^ 1);
        __syncthreads();
    }

    float* ybc = y + (size_t)b * Cout * plane;
    const int co0 = cot * NT * 16;
    const float unscale = split_unscale_of(ex) * split_unscale_of(ew);
    const float* abc = addend ? addend + (size_t)b * Cout * plane : nullptr;
    if (half == 0) o_store<NT0, NT0>(acc, ybc, bias, co0, plane, px0 + quarter * 64, g, n, unscale, abc);
    else if (NT1 > 0) o_store<NT1, NT0>(acc, ybc, bias, co0 + NT0 * 16, plane, px0 + quarter * 64, g, n, unscale, abc);
    if (stats && (size_t)(px0 + quarter * 64) < plane) {      // BatchNorm statistics of what was just stored (cseg_stats.h)
        const size_t seg = (size_t)b * ((plane + 63) / 64) + (size_t)(px0 + quarter * 64) / 64;
        if (half == 0)
            cseg_stats_emit<NT0, NT0>(acc, bias, co0, unscale, px0 + quarter * 64, (long)plane, g, n, stats + (size_t)co0 * n_seg + seg, n_seg);
        else if (NT1 > 0)
            cseg_stats_emit<NT1, NT0>(acc, bias, co0 + NT0 * 16, unscale, px0 + quarter * 64, (long)plane, g, n,
                                      stats + (size_t)(co0 + NT0 * 16) * n_seg + seg, n_seg);
    }
}

// channel tiles per block: the largest of {9, 8, 6, 4, 3} x 16 that divides Cout. Round 5, f16x3 only (two pieces: the LDS holds
// 2 x 32 KB of patch + 2 x 32 KB of weights): 16 tiles for multiples of 256 from 512 on, 15 for multiples of 240 (the 720-channel
// projection head) -- a block of 144 / 128 output channels re-reads every 32-channel slab of the input for 64 flop per byte, below
// what a CU can stream (the kernel ran at 0.19-0.29 of its roof, load-bound); 240 / 256 channels per block nearly double that.
// Not for exactly 256 output channels: one channel tile group leaves grids of ~264 blocks (DeepLab's 65 x 129 maps at batch 8)
// with a nearly empty second round. Measured (profiles/r05_conv1x1_wide_tiling.txt): 720 -> 720 at 8 x 128 x 256 1014 -> 869 us,
// 512 -> 512 at 8 x 65 x 129 228 -> 216 us. CSEG_CONV1X1_WIDE=0 restores the narrow tiling; read ONCE per process, because the
// packed weights carry the tiling and a pack made under one setting must never meet a forward made under the other.
int pick_nt1(int Cout, int arith) {
    static const bool wide = [] { const char* e = getenv("CSEG_CONV1X1_WIDE"); return !(e && atoi(e) == 0); }();
    if (arith == CSEG_ARITH_F16X3 && wide) {
        if (Cout % 256 == 0 && Cout >= 512) return 16;
        if (Cout % 240 == 0) return 15;
    }
    const int opts[] = {9, 8, 6, 4, 3};
    for (int nt : opts)
        if (Cout % (nt * 16) == 0) return nt;
    return 0;
}

template <class AR, int NT>
int launch_1x1(const float* x, const uint4* wp, const float* bias, int B, int Cin, int Cout, int plane, const unsigned* amax_x,
               const unsigned* amax_w, float* y, float4* stats, hipStream_t stream, const float* addend) {
    const size_t lds = sizeof(uint4) * (2 * AR::NP * 4 * MT_PX + 2 * NT * AR::NP * 64);
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void*)conv1x1_sb_kernel<AR, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) !=
            hipSuccess) {
            cseg_set_error("conv1x1_sb: cannot raise dynamic LDS to %zu bytes", lds);
            return 0;
        }
        attr_set = true;
    }
    const int tiles_p = (plane + MT_PX - 1) / MT_PX;
    const long n_tiles = (long)B * (Cout / (NT * 16)) * tiles_p;
    CSEG_REQUIRE(n_tiles < 2147483647L, "conv1x1_sb: grid too large");
    hipLaunchKernelGGL((conv1x1_sb_kernel<AR, NT>), dim3((unsigned)n_tiles), dim3(512), lds, stream, x, wp, bias, Cin, Cout, plane,
                       tiles_p, amax_x, amax_w, y, stats, B * ((plane + 63) / 64), cseg_xcd_remap(), addend);
    CSEG_CHECK_LAUNCH("conv1x1_sb_kernel");
    return 1;
}

template <class AR>
int fwd_1x1(const float* x, const uint4* wq, const float* bias, int B, int Cin, int Cout, int HW, int NT, const unsigned* amax_x,
            const unsigned* amax_w, float* y, float4* stats, hipStream_t stream, const float* addend) {
    switch (NT) {
        case 16: if constexpr (AR::NP == 2) return launch_1x1<AR, 16>(x, wq, bias, B, Cin, Cout, HW, amax_x, amax_w, y, stats, stream, addend);
        case 15: if constexpr (AR::NP == 2) return launch_1x1<AR, 15>(x, wq, bias, B, Cin, Cout, HW, amax_x, amax_w, y, stats, stream, addend);
        case 9: return launch_1x1<AR, 9>(x, wq, bias, B, Cin, Cout, HW, amax_x, amax_w, y, stats, stream, addend);
        case 8: return launch_1x1<AR, 8>(x, wq, bias, B, Cin, Cout, HW, amax_x, amax_w, y, stats, stream, addend);
        case 6: return launch_1x1<AR, 6>(x, wq, bias, B, Cin, Cout, HW, amax_x, amax_w, y, stats, stream, addend);
        case 4: return launch_1x1<AR, 4>(x, wq, bias, B, Cin, Cout, HW, amax_x, amax_w, y, stats, stream, addend);
        default: return launch_1x1<AR, 3>(x, wq, bias, B, Cin, Cout, HW, amax_x, amax_w, y, stats, stream, addend);
    }
}

int pack_1x1(const float* w, int Cout, int Cin, int transpose, int arith, const unsigned* amax_w, void* wp, hipStream_t stream) {
    const int conv_in = transpose ? Cout : Cin, conv_out = transpose ? Cin : Cout;
    CSEG_REQUIRE(w && wp, "conv1x1_sb_pack_weights: null pointer");
    CSEG_REQUIRE(arith == CSEG_ARITH_BF16X6 || (arith == CSEG_ARITH_F16X3 && amax_w), "conv1x1 split pack: arithmetic %d needs max|w|",
                 arith);
    const int NT = pick_nt1(conv_out, arith);
    CSEG_REQUIRE(conv_in % 16 == 0 && NT > 0,
                 "conv1x1_sb: needs input channels %% 16 == 0 and output channels %% 48 == 0 or %% 64 == 0 (got %d -> %d)", conv_in,
                 conv_out);
    CSEG_REQUIRE((reinterpret_cast<uintptr_t>(wp) & 15) == 0, "conv1x1_sb_pack_weights: packed buffer must be 16-byte aligned");
    const long total = (long)(conv_out / 16) * steps1(conv_in) * 64;
    CSEG_REQUIRE(total < 2147483647L, "conv1x1_sb_pack_weights: too large");
    if (arith == CSEG_ARITH_F16X3)
        hipLaunchKernelGGL(pack_weights_1x1_kernel<SplitF16x3>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, w, Cout,
                           Cin, transpose, NT, amax_w, (uint4*)wp, (int)total);
    else
        hipLaunchKernelGGL(pack_weights_1x1_kernel<SplitBF16x6>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, w, Cout,
                           Cin, transpose, NT, amax_w, (uint4*)wp, (int)total);
    CSEG_CHECK_LAUNCH("conv1x1_sb_pack_weights");
    return 1;
}

int run_1x1(const float* x, const void* wp, const float* bias, int B, int Cin, int Cout, int HW, int arith, const unsigned* amax_x,
            const unsigned* amax_w, float* y, hipStream_t stream, float4* stats = nullptr, const float* addend = nullptr) {
    CSEG_REQUIRE((reinterpret_cast<uintptr_t>(addend) & 15) == 0, "conv1x1 split: the addend must be 16-byte aligned like y");
    CSEG_REQUIRE((reinterpret_cast<uintptr_t>(stats) & 15) == 0, "conv1x1 split: the statistics buffer must be 16-byte aligned");
    CSEG_REQUIRE(x && wp && y, "conv1x1_sb: null pointer");
    CSEG_REQUIRE(arith == CSEG_ARITH_BF16X6 || (arith == CSEG_ARITH_F16X3 && amax_x && amax_w),
                 "conv1x1 split: arithmetic %d needs max|x| and max|w|", arith);
    const int NT = pick_nt1(Cout, arith);
    CSEG_REQUIRE(B > 0 && HW > 0 && Cin > 0 && Cin % 16 == 0 && NT > 0, "conv1x1_sb: unsupported shape B=%d Cin=%d Cout=%d HW=%d",
                 B, Cin, Cout, HW);
    CSEG_REQUIRE((reinterpret_cast<uintptr_t>(wp) & 15) == 0 && (reinterpret_cast<uintptr_t>(y) & 15) == 0,
                 "conv1x1_sb: packed weights / output must be 16-byte aligned and H*W a multiple of 4");
    const uint4* wq = (const uint4*)wp;
    if (arith == CSEG_ARITH_F16X3) return fwd_1x1<SplitF16x3>(x, wq, bias, B, Cin, Cout, HW, NT, amax_x, amax_w, y, stats, stream, addend);
    return fwd_1x1<SplitBF16x6>(x, wq, bias, B, Cin, Cout, HW, NT, amax_x, amax_w, y, stats, stream, addend);
}

}  // namespace

// channel tiles per block and packing threads of a (conv_in -> conv_out) operator in the given arithmetic (the tiling depends on it
// since round 5: pick_nt1). cseg_conv1x1_split_plan = the bf16x6 answer (round-3 signature, kept).
extern "C" int cseg_conv1x1_split_plan_arith(int arith, int conv_in, int conv_out, int* nt, long* threads) {
    if (!nt || !threads || conv_in <= 0 || conv_out <= 0 || conv_in % 16 || pick_nt1(conv_out, arith) == 0) return 0;
    *nt = pick_nt1(conv_out, arith);
    *threads = (long)(conv_out / 16) * steps1(conv_in) * 64;
    return 1;
}
extern "C" int cseg_conv1x1_split_plan(int conv_in, int conv_out, int* nt, long* threads) {
    return cseg_conv1x1_split_plan_arith(CSEG_ARITH_BF16X6, conv_in, conv_out, nt, threads);
}

extern "C" size_t cseg_conv1x1_split_packed_bytes(int arith, int Cin, int Cout) {
    if ((arith != CSEG_ARITH_BF16X6 && arith != CSEG_ARITH_F16X3) || Cin <= 0 || Cout <= 0 || Cin % 16 || pick_nt1(Cout, arith) == 0) return 0;
    return (size_t)(Cout / 16) * steps1(Cin) * (arith == CSEG_ARITH_F16X3 ? 2 : 3) * 64 * sizeof(uint4);
}

extern "C" size_t cseg_conv1x1_sb_packed_bytes(int Cin, int Cout) { return cseg_conv1x1_split_packed_bytes(CSEG_ARITH_BF16X6, Cin, Cout); }

extern "C" int cseg_conv1x1_sb_pack_weights(const float* w, int Cout, int Cin, int transpose, void* wp, cseg_stream_t stream_) {
    return pack_1x1(w, Cout, Cin, transpose, CSEG_ARITH_BF16X6, nullptr, wp, (hipStream_t)stream_);
}

extern "C" int cseg_conv1x1_split_pack(const float* w, int Cout, int Cin, int transpose, int arith, const unsigned* amax_w, void* wp,
                                       cseg_stream_t stream_) {
    return pack_1x1(w, Cout, Cin, transpose, arith, amax_w, wp, (hipStream_t)stream_);
}

extern "C" int cseg_conv1x1_sb_fwd(const float* x, const void* wp, const float* bias, int B, int Cin, int Cout, int HW,
                                   float* y, cseg_stream_t stream_) {
    return run_1x1(x, wp, bias, B, Cin, Cout, HW, CSEG_ARITH_BF16X6, nullptr, nullptr, y, (hipStream_t)stream_);
}

extern "C" int cseg_conv1x1_split_fwd(const float* x, const void* wp, const float* bias, int B, int Cin, int Cout, int HW, int arith,
                                      const unsigned* amax_x, const unsigned* amax_w, float* y, cseg_stream_t stream_) {
    return run_1x1(x, wp, bias, B, Cin, Cout, HW, arith, amax_x, amax_w, y, (hipStream_t)stream_);
}

// The same convolution with the BatchNorm statistics of its output from the epilogue: stats [Cout][cseg_conv_stat_segments(1, B, HW, 1)]
// float4 = (count, mean, M2) per run of 64 flat pixels (cseg_stats.h; see cseg_conv3x3_split_fwd_st).
extern "C" int cseg_conv1x1_split_fwd_st(const float* x, const void* wp, const float* bias, int B, int Cin, int Cout, int HW, int arith,
                                         const unsigned* amax_x, const unsigned* amax_w, float* y, float* stats, cseg_stream_t stream_) {
    CSEG_REQUIRE(stats, "conv1x1_split_fwd_st: null statistics buffer");
    return run_1x1(x, wp, bias, B, Cin, Cout, HW, arith, amax_x, amax_w, y, (hipStream_t)stream_, reinterpret_cast<float4*>(stats));
}

// y = conv1x1(x) + bias + addend (addend [B, Cout, H*W] like y, 16-byte aligned): the input gradient of a residual block's first 1x1
// convolution plus the gradient that arrives over the skip connection, in the epilogue instead of a separate add over both tensors.
extern "C" int cseg_conv1x1_split_fwd_add(const float* x, const void* wp, const float* bias, const float* addend, int B, int Cin, int Cout,
                                          int HW, int arith, const unsigned* amax_x, const unsigned* amax_w, float* y,
                                          cseg_stream_t stream_) {
    CSEG_REQUIRE(addend, "conv1x1_split_fwd_add: null addend");
    return run_1x1(x, wp, bias, B, Cin, Cout, HW, arith, amax_x, amax_w, y, (hipStream_t)stream_, nullptr, addend);
}
