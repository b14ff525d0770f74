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
#include "cseg_common.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int MT_PX = 256;                  // pixels per block
constexpr int A1_CELLS = 3 * 4 * MT_PX;     // one A buffer: [piece][octet][pixel]

__host__ __device__ constexpr int steps1(int Cin) { return (Cin + 31) / 32; }

__device__ __forceinline__ void split3o(float v, unsigned short& h, unsigned short& m, unsigned short& l) {
    const __bf16 bh = (__bf16)v;
    const float r1 = v - (float)bh;
    const __bf16 bm = (__bf16)r1;
    const float r2 = r1 - (float)bm;
    const __bf16 bl = (__bf16)r2;
    h = __builtin_bit_cast(unsigned short, bh);
    m = __builtin_bit_cast(unsigned short, bm);
    l = __builtin_bit_cast(unsigned short, bl);
}

__device__ __forceinline__ void split8o(const float (&v)[8], uint4& h, uint4& m, uint4& l) {
    unsigned short hs[8], ms[8], ls[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) split3o(v[j], hs[j], ms[j], ls[j]);
    h = make_uint4(hs[0] | ((unsigned)hs[1] << 16), hs[2] | ((unsigned)hs[3] << 16), hs[4] | ((unsigned)hs[5] << 16),
                   hs[6] | ((unsigned)hs[7] << 16));
    m = make_uint4(ms[0] | ((unsigned)ms[1] << 16), ms[2] | ((unsigned)ms[3] << 16), ms[4] | ((unsigned)ms[5] << 16),
                   ms[6] | ((unsigned)ms[7] << 16));
    l = make_uint4(ls[0] | ((unsigned)ls[1] << 16), ls[2] | ((unsigned)ls[3] << 16), ls[4] | ((unsigned)ls[5] << 16),
                   ls[6] | ((unsigned)ls[7] << 16));
}

// Packed weights: Wp[co_tile][kstep][nt][piece][lane] of uint4 (8 bf16, element j), lane = 16*g + n:
//   value(co = (co_tile*NT + nt)*16 + n, ci = 32*kstep + 8g + j), zero beyond the channel count.
// w is the forward's [Cout, Cin]; transpose = 1 packs the backward-data operator (maps Cout -> Cin channels).
__global__ __launch_bounds__(256) void pack_weights_1x1_kernel(const float* __restrict__ w, int Cout, int Cin, int transpose,
                                                               int NT, uint4* __restrict__ wp, int total) {
    const int e = blockIdx.x * 256 + threadIdx.x;          // one thread per (co_tile, kstep, nt, lane)
    if (e >= total) return;
    const int conv_in = transpose ? Cout : Cin;
    const int n_steps = steps1(conv_in);
    int r = e;
    const int lane = r & 63; r >>= 6;
    const int nt = r % NT; r /= NT;
    const int ks = r % n_steps;
    const int co_tile = r / n_steps;
    const int g = lane >> 4, n = lane & 15;
    const int oc = (co_tile * NT + nt) * 16 + n;           // output channel of THIS operator
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int ic = ks * 32 + 8 * g + j;                // input channel of THIS operator
        float t = 0.f;
        if (ic < conv_in) t = transpose ? w[(size_t)ic * Cin + oc] : w[(size_t)oc * Cin + ic];
        v[j] = t;
    }
    uint4 h, m, l;
    split8o(v, h, m, l);
    uint4* dst = wp + (((size_t)(co_tile * n_steps + ks) * NT + nt) * 3) * 64 + lane;
    dst[0] = h; dst[64] = m; dst[128] = l;
}

template <int NTW, int NTMAX>
__device__ __forceinline__ void o_kstep(const uint4* __restrict__ ap, const uint4* __restrict__ bp, f32x4 (&acc)[4][NTMAX]) {
    bf16x8 a[4][3];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int p = 0; p < 3; ++p) a[mt][p] = __builtin_bit_cast(bf16x8, ap[p * 4 * MT_PX + 16 * mt]);
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) {
        const bf16x8 b0 = __builtin_bit_cast(bf16x8, bp[(nt * 3 + 0) * 64]);
        const bf16x8 b1 = __builtin_bit_cast(bf16x8, bp[(nt * 3 + 1) * 64]);
        const bf16x8 b2 = __builtin_bit_cast(bf16x8, bp[(nt * 3 + 2) * 64]);
#define O_TERM(P, Q)                                                                                      \
    _Pragma("unroll") for (int mt = 0; mt < 4; ++mt)                                                      \
        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[mt][P], Q, acc[mt][nt], 0, 0, 0);
        O_TERM(2, b0)
        O_TERM(0, b2)
        O_TERM(1, b1)
        O_TERM(1, b0)
        O_TERM(0, b1)
        O_TERM(0, b0)
#undef O_TERM
    }
}

// accumulator layout: D[m = 4*g + r][n]: pixel px0 + 16*mt + 4*g + r, channel co0 + 16*nt + n
template <int NTW, int NTMAX>
__device__ __forceinline__ void o_store(const f32x4 (&acc)[4][NTMAX], float* __restrict__ ybc, const float* __restrict__ bias,
                                        int co0, size_t plane, int px0, int g, int n) {
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) {
        float* orow = ybc + (size_t)(co0 + nt * 16 + n) * plane;
        const float bv = bias ? bias[co0 + nt * 16 + n] : 0.f;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            const size_t px = (size_t)px0 + 16 * mt + 4 * g;
            f32x4 v = acc[mt][nt];
            v += bv;
            if (px + 3 < plane) *reinterpret_cast<float4*>(orow + px) = make_float4(v[0], v[1], v[2], v[3]);
            else {
                if (px < plane) orow[px] = v[0];
                if (px + 1 < plane) orow[px + 1] = v[1];
                if (px + 2 < plane) orow[px + 2] = v[2];
            }
        }
    }
}

template <int NT>
__global__ __launch_bounds__(512, 1) void conv1x1_sb_kernel(const float* __restrict__ x, const uint4* __restrict__ wp,
                                                            const float* __restrict__ bias, int Cin, int Cout, int plane_i,
                                                            int tiles_p, float* __restrict__ y) {
    extern __shared__ __attribute__((aligned(16))) uint4 smem_o[];
    uint4* As = smem_o;                            // [2][piece 3][octet 4][MT_PX]
    uint4* Bs = smem_o + 2 * A1_CELLS;             // [2][NT*3*64]
    constexpr int BSTEP = NT * 3 * 64;
    constexpr int NT0 = (NT + 1) / 2, NT1 = NT - NT0;

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int quarter = wave & 3, half = wave >> 2;
    const int g = lane >> 4, n = lane & 15;
    const int n_cot = Cout / (NT * 16);
    const size_t plane = (size_t)plane_i;
    int t = blockIdx.x;
    const int tp = t % tiles_p; t /= tiles_p;
    const int cot = t % n_cot;
    const int b = t / n_cot;
    const int px0 = tp * MT_PX;
    const int n_steps = steps1(Cin);
    const uint4* wbase = wp + (size_t)cot * n_steps * BSTEP;

    auto b_glds = [&](int ks, int buf) {
#pragma unroll
        for (int i = 0; i < (NT * 3 + 7) / 8; ++i) {
            const int r = wave + 8 * i;
            if (r < NT * 3)
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
            uint4 h, m, l;
            split8o(v, h, m, l);
            uint4* dst = As + buf * A1_CELLS + oct * MT_PX + p;
            dst[0] = h;
            dst[4 * MT_PX] = m;
            dst[8 * MT_PX] = l;
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
    const uint4* b_lane = Bs + (half ? NT0 * 3 * 64 : 0) + lane;
#pragma unroll 1
    for (int ks = 0; ks < n_steps; ++ks) {
        const int buf = ks & 1;
        const bool more = ks + 1 < n_steps;
        if (more) {
            b_glds(ks + 1, buf ^ 1);               // both stages of the other buffer were last read in step ks - 1
            a_issue(ks + 1);
        }
        if (half == 0) o_kstep<NT0, NT0>(a_lane + buf * A1_CELLS, b_lane + buf * BSTEP, acc);
        else if (NT1 > 0) o_kstep<NT1, NT0>(a_lane + buf * A1_CELLS, b_lane + buf * BSTEP, acc);
        if (more) a_store(ks + 1, buf ^ 1);
        __syncthreads();
    }

    float* ybc = y + (size_t)b * Cout * plane;
    const int co0 = cot * NT * 16;
    if (half == 0) o_store<NT0, NT0>(acc, ybc, bias, co0, plane, px0 + quarter * 64, g, n);
    else if (NT1 > 0) o_store<NT1, NT0>(acc, ybc, bias, co0 + NT0 * 16, plane, px0 + quarter * 64, g, n);
}

// channel tiles per block: the largest of {9, 8, 6, 4, 3} x 16 that divides Cout
int pick_nt1(int Cout) {
    const int opts[] = {9, 8, 6, 4, 3};
    for (int nt : opts)
        if (Cout % (nt * 16) == 0) return nt;
    return 0;
}

template <int NT>
int launch_1x1(const float* x, const uint4* wp, const float* bias, int B, int Cin, int Cout, int plane, float* y,
               hipStream_t stream) {
    const size_t lds = sizeof(uint4) * (2 * A1_CELLS + 2 * NT * 3 * 64);
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void*)conv1x1_sb_kernel<NT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) !=
            hipSuccess) {
            cseg_set_error("conv1x1_sb: cannot raise dynamic LDS to %zu bytes", lds);
            return 0;
        }
        attr_set = true;
    }
    const int tiles_p = (plane + MT_PX - 1) / MT_PX;
    const long n_tiles = (long)B * (Cout / (NT * 16)) * tiles_p;
    CSEG_REQUIRE(n_tiles < 2147483647L, "conv1x1_sb: grid too large");
    hipLaunchKernelGGL(conv1x1_sb_kernel<NT>, dim3((unsigned)n_tiles), dim3(512), lds, stream, x, wp, bias, Cin, Cout, plane,
                       tiles_p, y);
    CSEG_CHECK_LAUNCH("conv1x1_sb_kernel");
    return 1;
}

}  // namespace

extern "C" size_t cseg_conv1x1_sb_packed_bytes(int Cin, int Cout) {
    if (Cin <= 0 || Cout <= 0 || Cin % 16 || pick_nt1(Cout) == 0) return 0;
    return (size_t)(Cout / 16) * steps1(Cin) * 3 * 64 * sizeof(uint4);
}

extern "C" int cseg_conv1x1_sb_pack_weights(const float* w, int Cout, int Cin, int transpose, void* wp, cseg_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    const int conv_in = transpose ? Cout : Cin, conv_out = transpose ? Cin : Cout;
    CSEG_REQUIRE(w && wp, "conv1x1_sb_pack_weights: null pointer");
    const int NT = pick_nt1(conv_out);
    CSEG_REQUIRE(conv_in % 16 == 0 && NT > 0,
                 "conv1x1_sb: needs input channels %% 16 == 0 and output channels %% 48 == 0 or %% 64 == 0 (got %d -> %d)", conv_in,
                 conv_out);
    CSEG_REQUIRE((reinterpret_cast<uintptr_t>(wp) & 15) == 0, "conv1x1_sb_pack_weights: packed buffer must be 16-byte aligned");
    const long total = (long)(conv_out / 16) * steps1(conv_in) * 64;
    CSEG_REQUIRE(total < 2147483647L, "conv1x1_sb_pack_weights: too large");
    hipLaunchKernelGGL(pack_weights_1x1_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, w, Cout, Cin,
                       transpose, NT, (uint4*)wp, (int)total);
    CSEG_CHECK_LAUNCH("conv1x1_sb_pack_weights");
    return 1;
}

extern "C" int cseg_conv1x1_sb_fwd(const float* x, const void* wp, const float* bias, int B, int Cin, int Cout, int HW,
                                   float* y, cseg_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    CSEG_REQUIRE(x && wp && y, "conv1x1_sb: null pointer");
    const int NT = pick_nt1(Cout);
    CSEG_REQUIRE(B > 0 && HW > 0 && Cin > 0 && Cin % 16 == 0 && NT > 0, "conv1x1_sb: unsupported shape B=%d Cin=%d Cout=%d HW=%d",
                 B, Cin, Cout, HW);
    CSEG_REQUIRE((reinterpret_cast<uintptr_t>(wp) & 15) == 0 && (reinterpret_cast<uintptr_t>(y) & 15) == 0 && HW % 4 == 0,
                 "conv1x1_sb: packed weights / output must be 16-byte aligned and H*W a multiple of 4");
    const uint4* wq = (const uint4*)wp;
    switch (NT) {
        case 9: return launch_1x1<9>(x, wq, bias, B, Cin, Cout, HW, y, stream);
        case 8: return launch_1x1<8>(x, wq, bias, B, Cin, Cout, HW, y, stream);
        case 6: return launch_1x1<6>(x, wq, bias, B, Cin, Cout, HW, y, stream);
        case 4: return launch_1x1<4>(x, wq, bias, B, Cin, Cout, HW, y, stream);
        default: return launch_1x1<3>(x, wq, bias, B, Cin, Cout, HW, y, stream);
    }
}
