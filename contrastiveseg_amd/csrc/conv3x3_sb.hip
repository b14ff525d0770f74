// 3x3 / stride 1 / pad 1 convolution, NCHW fp32 in and out, evaluated on the BF16 matrix cores with split operands
// ("bf16x6"): every fp32 operand is written as hi + mid + lo with three bf16 pieces (8 + 8 + 8 mantissa bits: the
// split is exact for normal numbers), and the product a*b is accumulated in fp32 from the six piece products whose
// weight is >= 2^-16 of the leading one:
//        a*b ~= a0*b0 + (a0*b1 + a1*b0) + (a0*b2 + a1*b1 + a2*b0)            (dropped: 2^-24 and below)
// Products of bf16 pieces are exact in fp32 and the MFMA accumulates in fp32, so the result is in the fp32 rounding
// class: on the reference HRNet-W48 the logits deviate from the fp64 evaluation by 2.1e-5 (plain fp32: 4.3e-5; three
// terms only: 1.4e-3; the reference's own TF32 default on Ampere-class GPUs: 9.6e-2) -- tools/split_bf16_probe.py.
// Why: gfx950's fp32 MFMA runs at 1/16 of the bf16 rate (157 vs 2500 TFLOP/s). Six bf16 MFMAs per fp32-equivalent
// product put the bound at 2500/6 = 417 TFLOP/s of fp32-equivalent work, 2.65x the fp32 MFMA roofline that bounds
// MIOpen's kernels (the 720->720 head convolution runs there at 0.78-0.83 of that roof already).
//
// Mapping (one block = 8 waves = 4 image rows x 64 columns x NT*16 output channels; two waves per row split the
// channel tiles):
//   GEMM M = pixels, N = output channels, K = input channels x 9 taps, v_mfma_f32_16x16x32_bf16.
//   K-step = 32 k-values = one tap x 32 consecutive input channels (lane group g = lane/16 owns channels 8g..8g+7);
//       a trailing 16-channel chunk (720 = 22*32 + 16, 48 = 32 + 16) pairs two taps per K-step instead (groups 0,1 =
//       first tap, groups 2,3 = second tap; the ninth tap is paired with zero weights).
//   A operand: the fp32 input patch of a 32-channel chunk (6 rows x 66 columns incl. halo) is split while it is staged:
//       LDS image [piece][channel octet][row][column] of 16-byte cells (8 bf16 of one pixel) -- a lane's fragment is one
//       ds_read_b128, consecutive lanes read consecutive cells (conflict-free), and the tap offset is a whole number of
//       cells, so shifted reads stay aligned.
//   B operand: weights pre-split and pre-packed by cseg_conv3x3_sb_pack_weights in lane order per (channel tile, K-step,
//       column tile, piece); the block streams one K-step (NT*3 KB) ahead through a double-buffered LDS stage.
//   C: 4 pixel tiles x NT channel tiles of 16x16 per wave; six MFMAs per (pixel tile, channel tile, K-step), smallest
//       terms first. Lane holds 4 consecutive pixels of one channel per accumulator -> 16-byte stores.
// Backward-data is the same kernel on weights packed transposed and mirrored.
// Measured on MI355X (tools/conv3x3_sb_probe.py, profiles/r02_conv3x3_split_bf16_probe.jsonl): 720->720 at 8x128x256
// 10.97 ms forward / 10.93 ms backward-data incl. weight packing = 223 TFLOP/s fp32-equivalent (0.53 of 417) vs MIOpen's
// fp32 kernel 19.81 ms (123.5 TFLOP/s); 96 ch 70 vs 101 us; 48 ch 99 vs 106 us (fp32-MFMA kernel) / 141 us (MIOpen).
// Switch: kernels.CONV3X3_SPLIT_BF16 (env CSEG_CONV3X3_SPLIT_BF16, default 1; 0 = the fp32 MFMA / MIOpen path).
// Round 3: the kernel is written against the arithmetic traits of cseg_split.h and instantiated for both forms -- bf16x6 (above)
// and f16x3 (two scaled fp16 pieces, three MFMAs per product, roof 2500 / 3 = 833 TFLOP/s of fp32-equivalent work). The f16x3
// form needs max|x| and max|w| (device pointers, bit patterns; cseg_amax_f32): the patch is scaled while it is split, the weights
// while they are packed, and the epilogue multiplies the accumulators by the inverse power of two.
#include "cseg_pack.h"
#include "cseg_stats.h"
#include <stdlib.h>

// conv3x3_sb16.hip: the small-channel variant (16-channel chunks, two blocks per CU), reached under CSEG_CONV3X3_SB_VAR=2
namespace cseg_sb16 {
size_t packed_bytes(int arith, int Cin, int Cout);
int pack(const float* w, int Cout, int Cin, int transpose_flip, int NT, int arith, const unsigned* amax_w, void* wp,
         hipStream_t stream);
int fwd8(const float* x, const void* wp, const float* bias, int B, int Cin, int Cout, int H, int W, int arith, const unsigned* amax_x,
         const unsigned* amax_w, float* y, float4* stats, hipStream_t stream);
int fwd(const float* x, const void* wp, const float* bias, const float* addend, int B, int Cin, int Cout, int H, int W, int NT, int arith,
        const unsigned* amax_x, const unsigned* amax_w, float* y, float4* stats, hipStream_t stream);
int fwd_dil(const float* x, const void* wp, const float* bias, const float* addend, int B, int Cin, int Cout, int H, int W, int NT, int dil,
            int arith, const unsigned* amax_x, const unsigned* amax_w, float* y, float4* stats, hipStream_t stream);
}  // namespace cseg_sb16

namespace {

// CSEG_CONV3X3_SB_VAR: 0 = the kernel as verified and timed on the MI355X; 1 = buffer-load addressing of the patch (template
// comment below); 2 = conv3x3_sb16.hip for convolutions with at most 192 output channels (3 channel tiles per block unless
// the caller asks for 6), everything else as 1. Read per call: tests and probes switch it inside one process.
// Unset = the per-shape choice measured on the MI355X in the round-2 driver pass (GPUTEST_r02.json, fwd_us): the 16-channel-chunk
// kernel at 48 / 192 output channels (67-78 vs 95-105 us), buffer-load addressing everywhere else (head 9.8 vs 11.0 ms, 96
// channels 60 vs 64 us).
int sb_variant() {
    const char* e = getenv("CSEG_CONV3X3_SB_VAR");
    return e ? atoi(e) : -1;
}
// Output channel counts that are multiples of 64 but not of 48 (4 channel tiles per block: the 3x3 convolutions of the layer-1
// bottlenecks, the backward-data operator of the 256 -> 48 transition) exist only in the 16-channel-chunk kernel.
// CSEG_CONV3X3_SB16_CH = comma-separated output channel counts that go to conv3x3_sb16.hip (tuning; overrides the default list).
bool use_sb16(int conv_out) {
    const int v = sb_variant();
    if (conv_out % 48 != 0) return conv_out % 64 == 0;     // 64, 128, 256, ...: four channel tiles per block, 16-channel-chunk kernel only
    const char* list = getenv("CSEG_CONV3X3_SB16_CH");
    if (list) {
        for (const char* p = list; *p;) {
            if (atoi(p) == conv_out) return conv_out % 48 == 0;
            while (*p && *p != ',') ++p;
            if (*p == ',') ++p;
        }
        return false;
    }
    if (v == 2) return conv_out <= 192 && conv_out % 48 == 0;
    // measured on the MI355X, f16x3, benched shapes (tools/branch_conv_probe.py, profiles/r03_branch_conv_probe.jsonl): 48 ch 51.6 vs
    // 72.4 us on the 32-channel-chunk kernel, 192 ch 47.4 vs 48.1, 384 ch 79.8 vs 87.1; 96 ch stays (41-45 vs 48)
    return v < 0 && (conv_out == 48 || conv_out == 192 || conv_out == 384);
}
int sb16_nt(int conv_out, int NT) { return conv_out % 48 != 0 ? 4 : (NT == 6 ? 6 : 3); }

constexpr int TR = 4;                 // output rows per block (one per wave)
constexpr int TC = 64;                // output columns per block
constexpr int XROWS = TR + 2;
constexpr int XCOLS = TC + 2;         // cell 0 = column x0 - 1
constexpr int CELLS = XROWS * XCOLS;  // 396 pixels per (piece, octet)
// LDS stride of one (piece, octet) plane: a multiple of 16 cells (256 bytes = all 64 banks). ds_read_b128 serves the lane groups
// {0-3,12-15,20-27}, {4-11,16-19,28-31}, ... (MI355X_MICROARCH.md), i.e. lanes of TWO channel octets (g, g+1) per group: with
// the planes 396 cells apart their bank offsets differ by 192 bytes and every fragment read was 2-way conflicted (round-3
// counters: SQ_LDS_BANK_CONFLICT = 30 % of SQ_LDS_IDX_ACTIVE on the head kernel); a stride of 0 mod 256 bytes is conflict-free.
constexpr int PLANE = (CELLS + 15) / 16 * 16;
constexpr int A_ITEMS = 4 * CELLS;    // (octet, pixel) staging items of a 32-channel chunk
constexpr int AU = (A_ITEMS + 511) / 512;     // staging items per thread (512 threads)

__host__ __device__ constexpr int steps_of(int Cin) { return pack_steps_c3(Cin); }

// Packed weights: Wp[co_tile][kstep][nt][piece][lane] of uint4 (8 bf16, element j), lane = 16*g + n:
//   full K-step (chunk c, tap t):  value(co = (co_tile*NT + nt)*16 + n, ci = 32c + 8g + j, tap t)
//   tail K-step q (last 16 channels): ci = 32*n_full + 8*(g&1) + j, tap = 2q + (g>>1)   (zero when tap > 8)
template <class AR>
__global__ __launch_bounds__(256) void pack_weights_sb_kernel(const float* __restrict__ w, int Cout, int Cin,
                                                              int transpose_flip, int NT, const unsigned* __restrict__ amax_w,
                                                              uint4* __restrict__ wp, int total) {
    const float wscale = AR::SCALED ? split_scale_of(split_amax_exp(amax_w)) : 1.f;      // every thread (shuffles inside)
    const int e = blockIdx.x * 256 + threadIdx.x;          // one thread per (co_tile, kstep, nt, lane)
    if (e >= total) return;
    pack_elem_c3<AR>(w, Cout, Cin, transpose_flip, NT, wscale, wp, e);
}

// One K-step of a wave: 4 pixel tiles x NTW channel tiles x 6 piece products. `ap` = this lane's cell in the hi-piece
// image (octet, row, tap and column already applied), `bp` = this lane's slot in the staged B step.
template <class AR, int NTW, int NTMAX, int ABL = 0>
__device__ __forceinline__ void sb_kstep(const uint4* __restrict__ ap, const uint4* __restrict__ bp,
                                         f32x4 (&acc)[4][NTMAX]) {
    typedef typename AR::frag_t frag_t;
    frag_t a[4][AR::NP];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int p = 0; p < AR::NP; ++p) a[mt][p] = __builtin_bit_cast(frag_t, ap[p * 4 * PLANE + 16 * mt]);
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) {
        frag_t b[AR::NP];
#pragma unroll
        for (int p = 0; p < AR::NP; ++p) b[p] = __builtin_bit_cast(frag_t, bp[(nt * AR::NP + p) * 64]);
        // term-major, smallest terms first: the four pixel tiles between two MFMAs on the same accumulator hide the
        // dependent-accumulator latency
#pragma unroll
        for (int t = 0; t < AR::NTERMS; ++t)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                if (ABL & 1) acc[mt][nt][0] += (float)a[mt][AR::ta(t)][0] + (float)b[AR::tb(t)][0];
                else acc[mt][nt] = AR::mfma(a[mt][AR::ta(t)], b[AR::tb(t)], acc[mt][nt]);
            }
    }
}

// accumulator layout: D[m = 4*g + r][n]: pixel column x0 + 16*mt + 4*g + r, channel co0 + 16*nt + n
// addend (nullable): a tensor of the output's shape added in the epilogue -- the residual gradient that meets the backward-data result of
// a BasicBlock's first convolution (hrnet_backbone.py: BasicBlock.forward); saves the separate elementwise add autograd would launch
template <int NTW, int NTMAX>
__device__ __forceinline__ void sb_store(const f32x4 (&acc)[4][NTMAX], float* __restrict__ ybc,
                                         const float* __restrict__ bias, const float* __restrict__ abc, int co0, size_t plane, int yy,
                                         int x0, int W, int g, int n, float unscale) {
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) {
        const size_t roff = (size_t)(co0 + nt * 16 + n) * plane + (size_t)yy * W;
        float* orow = ybc + roff;
        const float bv = bias ? bias[co0 + nt * 16 + n] : 0.f;
        const bool vec = (W & 3) == 0;              // rows 16-byte aligned (else element by element: cseg_store_row4)
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            const int xx = x0 + 16 * mt + 4 * g;
            f32x4 v = acc[mt][nt] * unscale;
            v += bv;
            cseg_store_row4(orow, abc ? abc + roff : nullptr, xx, W, vec, v);
        }
    }
}

template <class AR, int NT, bool GLDS, int VAR = 0, int ABL = 0, int SPS = 1>
__global__ __launch_bounds__(512, 1) void conv3x3_sb_kernel(const float* __restrict__ x, const uint4* __restrict__ wp,
                                                            const float* __restrict__ bias, const float* __restrict__ addend, int Cin, int Cout, int H,
                                                            int W, int tiles_x, int tiles_y,
                                                            const unsigned* __restrict__ amax_x,
                                                            const unsigned* __restrict__ amax_w, float* __restrict__ y,
                                                            float4* __restrict__ stats, int n_seg, int xmap) {
    extern __shared__ __attribute__((aligned(16))) uint4 smem_sb[];
    constexpr int NP = AR::NP;
    constexpr int A_CELLS = NP * 4 * PLANE;
    uint4* As = smem_sb;                           // [piece NP][octet 4][CELLS]
    uint4* Bs = smem_sb + A_CELLS;                 // [2][SPS][NT*NP*64]
    constexpr int BSTEP = NT * NP * 64;            // uint4 per K-step
    constexpr int BSTAGE = SPS * BSTEP;
    const unsigned ex = AR::SCALED ? split_amax_exp(amax_x) : 141u, ew = AR::SCALED ? split_amax_exp(amax_w) : 141u;
    const float xscale = split_scale_of(ex);       // 1 for the unscaled arithmetic
    constexpr int BU = (BSTEP + 511) / 512;
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

    const int n_full = Cin / 32;
    const int n_chunks = n_full + ((Cin & 31) ? 1 : 0);
    const int n_steps = steps_of(Cin);
    const uint4* wbase = wp + (size_t)cot * n_steps * BSTEP;

    uint4 bpre[BU];
    auto b_issue = [&](int ks) {
#pragma unroll
        for (int u = 0; u < BU; ++u) {
            const int idx = tid + 512 * u;
            bpre[u] = wbase[(size_t)ks * BSTEP + (idx < BSTEP ? idx : BSTEP - 1)];      // clamped: always defined
        }
    };
    auto b_store = [&](int buf) {
#pragma unroll
        for (int u = 0; u < BU; ++u) {
            const int idx = tid + 512 * u;
            if (idx < BSTEP) Bs[buf * BSTEP + idx] = bpre[u];
        }
    };

    // A staging: item = (octet, patch pixel); 8 channel loads each (coalesced along the row), branch-free: the address
    // is clamped into the tensor and the value masked
    auto b_glds = [&](int ks, int buf) {
        if (ABL & 8) return;
#pragma unroll
        for (int i = 0; i < (NT * NP + 7) / 8; ++i) {
            const int r = wave + 8 * i;                  // one 1 KB row (channel tile, piece) per wave instruction
            if (r < NT * NP)
                __builtin_amdgcn_global_load_lds(
                    (const __attribute__((address_space(1))) void*)(wbase + (size_t)ks * BSTEP + r * 64 + lane),
                    (__attribute__((address_space(3))) void*)(Bs + buf * BSTEP + r * 64), 16, 0, 0);
        }
    };

    float apre[AU][8];
    auto a_item = [&](int u, int chunk, int& oct, int& rc, bool& ok) {
        const int n_oct = chunk < n_full ? 4 : 2;
        const int item = tid + 512 * u;
        oct = item / CELLS; rc = item - oct * CELLS;
        const int r = rc / XCOLS, col = rc - r * XCOLS;
        const int yy = y0 + r - 1, xx = x0 + col - 1;
        ok = oct < n_oct && yy >= 0 && yy < H && xx >= 0 && xx < W;
    };
    auto a_issue = [&](int chunk) {
        if (ABL & 4) {
#pragma unroll
            for (int u = 0; u < AU; ++u)
#pragma unroll
                for (int j = 0; j < 8; ++j) apre[u][j] = 1.f + j;
            return;
        }
        const int n_oct = chunk < n_full ? 4 : 2;
        const float* xc = x + ((size_t)b * Cin + (size_t)chunk * 32) * plane;
#pragma unroll
        for (int u = 0; u < AU; ++u) {
            int oct, rc;
            bool ok;
            a_item(u, chunk, oct, rc, ok);
            const int r = rc / XCOLS, col = rc - r * XCOLS;
            const int octc = min(oct, n_oct - 1), yc = min(max(y0 + r - 1, 0), H - 1), xcl = min(max(x0 + col - 1, 0), W - 1);
            if constexpr (VAR & 1) {
                // buffer_load_dword v, v_off, s[rsrc], s_soff offen: the chunk's 32 channel planes as one buffer resource
                // (SGPRs), the channel plane as scalar offset, ONE per-lane byte offset per item
                const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
                    (void*)xc, 0, (int)((size_t)n_oct * 8 * plane * sizeof(float)), 0x00020000);
                const int off = (octc * 8 * (int)plane + yc * W + xcl) * (int)sizeof(float);
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    apre[u][j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                                               rs, off, j * (int)plane * (int)sizeof(float), 0));
            } else {
                const float* p = xc + (size_t)(octc * 8) * plane + (size_t)yc * W + xcl;
#pragma unroll
                for (int j = 0; j < 8; ++j) apre[u][j] = p[(size_t)j * plane];      // raw; masked when it is stored
            }
        }
    };
    auto a_store = [&](int chunk) {
        const int n_oct = chunk < n_full ? 4 : 2;
#pragma unroll
        for (int u = 0; u < AU; ++u) {
            int oct, rc;
            bool ok;
            a_item(u, chunk, oct, rc, ok);
            if (oct < n_oct) {
                float v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = (ABL & 2) ? (ok ? apre[u][j] : 0.f) : apre[u][j];
                uint4 cells[NP];
                if (ABL & 2) {
#pragma unroll
                    for (int p = 0; p < NP; ++p)
                        cells[p] = make_uint4(__builtin_bit_cast(unsigned, v[0]) >> 16 | (__builtin_bit_cast(unsigned, v[1]) & 0xffff0000u),
                                              __builtin_bit_cast(unsigned, v[2]) >> 16 | (__builtin_bit_cast(unsigned, v[3]) & 0xffff0000u),
                                              __builtin_bit_cast(unsigned, v[4]) >> 16 | (__builtin_bit_cast(unsigned, v[5]) & 0xffff0000u),
                                              __builtin_bit_cast(unsigned, v[6]) >> 16 | (__builtin_bit_cast(unsigned, v[7]) & 0xffff0000u));
                } else split_cells8_masked<AR>(v, ok, xscale, cells);           // zero padding / outside the tensor
                const int item = oct * PLANE + rc;
#pragma unroll
                for (int p = 0; p < NP; ++p) As[p * 4 * PLANE + item] = cells[p];
            }
        }
    };

    f32x4 acc[4][NT0];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT0; ++nt) acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

    if constexpr (SPS > 1) {
        static_assert(GLDS, "staged weights need the LDS-DMA path");
        auto b_stage = [&](int ks0, int nsteps, int buf) {           // nsteps consecutive K-steps -> stage `buf`
            if (ABL & 8) return;
            const int rows = nsteps * NT * NP;
#pragma unroll
            for (int i = 0; i < (SPS * NT * NP + 7) / 8; ++i) {
                const int r = wave + 8 * i;                          // one 1 KB row per wave instruction
                if (r < rows)
                    __builtin_amdgcn_global_load_lds(
                        (const __attribute__((address_space(1))) void*)(wbase + (size_t)ks0 * BSTEP + r * 64 + lane),
                        (__attribute__((address_space(3))) void*)(Bs + buf * BSTAGE + r * 64), 16, 0, 0);
            }
        };
        auto steps_of_chunk = [&](int c) { return c < n_full ? 9 : 5; };
        a_issue(0);
        b_stage(0, min(SPS, steps_of_chunk(0)), 0);
        a_store(0);
        __syncthreads();
        const uint4* a_lane = As + row * XCOLS + n;
        const uint4* b_lane = Bs + (half ? NT0 * NP * 64 : 0) + lane;
        int ks = 0, buf = 0;
        for (int c = 0; c < n_chunks; ++c) {
            const bool full = c < n_full;
            const int steps = steps_of_chunk(c);
#pragma unroll 1
            for (int s0 = 0; s0 < steps; s0 += SPS) {
                const int nst = min(SPS, steps - s0);
                const bool last = s0 + nst >= steps;                 // last stage of this chunk
                if (ks + nst < n_steps)                              // next stage: the rest of this chunk or the head of the next
                    b_stage(ks + nst, last ? min(SPS, steps_of_chunk(c + 1)) : min(SPS, steps - s0 - nst), buf ^ 1);
                if (last && c + 1 < n_chunks) a_issue(c + 1);
#pragma unroll
                for (int j = 0; j < SPS; ++j) {
                    if (j < nst) {
                        const int st = s0 + j;
                        int a_off;
                        if (full) {
                            const int ky = st / 3, kx = st - 3 * ky;
                            a_off = g * PLANE + ky * XCOLS + kx;
                        } else {
                            const int tap = min(2 * st + (g >> 1), 8);   // tail chunk: two taps x 16 channels per K-step
                            const int ky = tap / 3, kx = tap - 3 * ky;
                            a_off = (g & 1) * PLANE + ky * XCOLS + kx;
                        }
                        if (half == 0) sb_kstep<AR, NT0, NT0, ABL>(a_lane + a_off, b_lane + buf * BSTAGE + j * BSTEP, acc);
                        else if (NT1 > 0) sb_kstep<AR, NT1, NT0, ABL>(a_lane + a_off, b_lane + buf * BSTAGE + j * BSTEP, acc);
                    }
                }
                if (last && c + 1 < n_chunks) {
                    __syncthreads();                // every wave is done with this chunk's patch
                    a_store(c + 1);
                }
                __syncthreads();
                buf ^= 1;
                ks += nst;
            }
        }
    } else {
        a_issue(0);
        if (GLDS) b_glds(0, 0);
        else b_issue(0);
        a_store(0);
        if (!GLDS) b_store(0);
        __syncthreads();

        const uint4* a_lane = As + row * XCOLS + n;                        // + octet / tap offset per K-step
        const uint4* b_lane = Bs + (half ? NT0 * NP * 64 : 0) + lane;      // + buffer offset per K-step
        int ks = 0, buf = 0;
        for (int c = 0; c < n_chunks; ++c) {
            const bool full = c < n_full;
            const int steps = full ? 9 : 5;
    #pragma unroll 1
            for (int s = 0; s < steps; ++s) {
                const bool more = ks + 1 < n_steps;
                if (more) {
                    if (GLDS) b_glds(ks + 1, buf ^ 1);      // that buffer was last read in step ks - 1 (barrier since)
                    else b_issue(ks + 1);
                }
                if (s == steps - 3 && c + 1 < n_chunks) a_issue(c + 1);
                int a_off;
                if (full) {
                    const int ky = s / 3, kx = s - 3 * ky;
                    a_off = g * PLANE + ky * XCOLS + kx;
                } else {
                    // the ninth tap is paired with a tenth that does not exist: its packed weights are zero, so whatever
                    // (finite) patch values those lanes read contribute nothing
                    const int tap = min(2 * s + (g >> 1), 8);
                    const int ky = tap / 3, kx = tap - 3 * ky;
                    a_off = (g & 1) * PLANE + ky * XCOLS + kx;
                }
                if (half == 0) sb_kstep<AR, NT0, NT0, ABL>(a_lane + a_off, b_lane + buf * BSTEP, acc);
                else if (NT1 > 0) sb_kstep<AR, NT1, NT0, ABL>(a_lane + a_off, b_lane + buf * BSTEP, acc);
                if (more && !GLDS) b_store(buf ^ 1);
                if (s == steps - 1 && c + 1 < n_chunks) {
                    __syncthreads();                    // every wave is done with this chunk's patch
                    a_store(c + 1);
                }
                __syncthreads();
                buf ^= 1;
                ++ks;
            }
        }
    }

    const int yy = y0 + row;
    if (yy < H) {
        float* ybc = y + (size_t)b * Cout * plane;
        const int co0 = cot * NT * 16;
        const float unscale = split_unscale_of(ex) * split_unscale_of(ew);
        const float* abc = addend ? addend + (size_t)b * Cout * plane : nullptr;
        if (half == 0) sb_store<NT0, NT0>(acc, ybc, bias, abc, co0, plane, yy, x0, W, g, n, unscale);
        else if (NT1 > 0) sb_store<NT1, NT0>(acc, ybc, bias, abc, co0 + NT0 * 16, plane, yy, x0, W, g, n, unscale);
        if (stats) {                                // BatchNorm statistics of what was just stored (cseg_stats.h)
            const size_t seg = ((size_t)b * H + yy) * tiles_x + tx;
            if (half == 0) cseg_stats_emit<NT0, NT0>(acc, bias, co0, unscale, x0, W, g, n, stats + (size_t)co0 * n_seg + seg, n_seg);
            else if (NT1 > 0)
                cseg_stats_emit<NT1, NT0>(acc, bias, co0 + NT0 * 16, unscale, x0, W, g, n, stats + (size_t)(co0 + NT0 * 16) * n_seg + seg, n_seg);
        }
    }
}

// channel tiles per block: the largest of {9, 6, 3} x 16 that divides Cout
int pick_nt(int Cout) {
    if (Cout % 144 == 0) return 9;
    if (Cout % 96 == 0) return 6;
    if (Cout % 48 == 0) return 3;
    if (Cout % 64 == 0) return 4;                          // conv3x3_sb16.hip only (64: layer-1 bottlenecks; 256: input side of transition 1)
    return 0;
}

template <class AR, int NT, bool GLDS, int VAR = 0, int ABL = 0, int SPS = 1>
int launch_sb(const float* x, const uint4* wp, const float* bias, const float* addend, int B, int Cin, int Cout, int H, int W,
              const unsigned* amax_x, const unsigned* amax_w, float* y, float4* stats, hipStream_t stream) {
    const size_t lds = sizeof(uint4) * (AR::NP * 4 * PLANE + 2 * SPS * NT * AR::NP * 64);
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void*)conv3x3_sb_kernel<AR, NT, GLDS, VAR, ABL, SPS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) !=
            hipSuccess) {
            cseg_set_error("conv3x3_sb: cannot raise dynamic LDS to %zu bytes", lds);
            return 0;
        }
        attr_set = true;
    }
    const int tiles_x = (W + TC - 1) / TC, tiles_y = (H + TR - 1) / TR;
    const long n_tiles = (long)B * (Cout / (NT * 16)) * tiles_y * tiles_x;
    CSEG_REQUIRE(n_tiles < 2147483647L, "conv3x3_sb: grid too large");
    hipLaunchKernelGGL((conv3x3_sb_kernel<AR, NT, GLDS, VAR, ABL, SPS>), dim3((unsigned)n_tiles), dim3(512), lds, stream, x, wp, bias, addend, Cin, Cout, H, W, tiles_x,
                       tiles_y, amax_x, amax_w, y, stats, B * H * tiles_x, cseg_xcd_remap());
    CSEG_CHECK_LAUNCH("conv3x3_sb_kernel");
    return 1;
}

}  // namespace

namespace {
bool nt_ok(int nt, int Cout) { return (nt == 3 || nt == 6 || nt == 9) && Cout % (nt * 16) == 0; }
bool arith_ok(int arith) { return arith == CSEG_ARITH_BF16X6 || arith == CSEG_ARITH_F16X3; }
int np_of(int arith) { return arith == CSEG_ARITH_F16X3 ? 2 : 3; }
}  // namespace

extern "C" size_t cseg_conv3x3_split_packed_bytes(int arith, int Cin, int Cout) {
    if (!arith_ok(arith) || Cin <= 0 || Cout <= 0 || Cin % 16 || pick_nt(Cout) == 0) return 0;
    if (use_sb16(Cout)) return cseg_sb16::packed_bytes(arith, Cin, Cout);
    const size_t own = (size_t)(Cout / 16) * steps_of(Cin) * np_of(arith) * 64 * sizeof(uint4);
    if (arith == CSEG_ARITH_F16X3 && Cout % 48 == 0) {           // nt = CSEG_NT_SB8 / CSEG_NT_GROUP pack the 16-channel-chunk format: room for either
        const size_t alt = cseg_sb16::packed_bytes(arith, Cin, Cout);
        return alt > own ? alt : own;
    }
    return own;
}

extern "C" int cseg_conv3x3_split_plan(int conv_in, int conv_out, int nt_request, int* kind, int* nt, long* threads) {
    if (!kind || !nt || !threads || conv_in <= 0 || conv_out <= 0 || conv_in % 16 || pick_nt(conv_out) == 0) return 0;
    if (nt_request == CSEG_NT_SB8) {                             // the 8-row head kernel: 16-channel-chunk format, nine tiles
        if (conv_out % 144) return 0;
        *kind = CSEG_PACK_C3_16;
        *nt = 9;
        *threads = (long)(conv_out / 16) * pack_steps_c3_16(conv_in) * 64;
        return 1;
    }
    if (nt_request == CSEG_NT_GROUP) {                           // member of a grouped launch: 16-channel-chunk format, three tiles
        if (conv_out % 48) return 0;
        *kind = CSEG_PACK_C3_16;
        *nt = 3;
        *threads = (long)(conv_out / 16) * pack_steps_c3_16(conv_in) * 64;
        return 1;
    }
    if (use_sb16(conv_out)) {
        *kind = CSEG_PACK_C3_16;
        *nt = sb16_nt(conv_out, nt_request);
        if (conv_out % (*nt * 16)) return 0;
        *threads = (long)(conv_out / 16) * pack_steps_c3_16(conv_in) * 64;
        return 1;
    }
    *kind = CSEG_PACK_C3;
    *nt = nt_request ? nt_request : pick_nt(conv_out);
    if (!nt_ok(*nt, conv_out)) return 0;
    *threads = (long)(conv_out / 16) * pack_steps_c3(conv_in) * 64;
    return 1;
}

extern "C" size_t cseg_conv3x3_sb_packed_bytes(int Cin, int Cout) {
    return cseg_conv3x3_split_packed_bytes(CSEG_ARITH_BF16X6, Cin, Cout);
}

static int pack_impl(const float* w, int Cout, int Cin, int transpose_flip, int NT, int arith, const unsigned* amax_w, void* wp,
                     hipStream_t stream) {
    // transpose_flip: w is still the forward's [Cout, Cin, 3, 3]; the packed operator maps Cout -> Cin channels
    const int conv_in = transpose_flip ? Cout : Cin, conv_out = transpose_flip ? Cin : Cout;
    CSEG_REQUIRE(w && wp, "conv3x3_sb_pack_weights: null pointer");
    CSEG_REQUIRE(arith_ok(arith) && (arith == CSEG_ARITH_BF16X6 || amax_w), "conv3x3 split pack: arithmetic %d needs max|w|", arith);
    if (NT == CSEG_NT_SB8) {
        CSEG_REQUIRE(arith == CSEG_ARITH_F16X3 && conv_out % 144 == 0, "conv3x3 split pack: nt = CSEG_NT_SB8 needs f16x3 and output channels %% 144");
        return cseg_sb16::pack(w, Cout, Cin, transpose_flip, 9, arith, amax_w, wp, stream);
    }
    if (NT == CSEG_NT_GROUP) {
        CSEG_REQUIRE(conv_out % 48 == 0, "conv3x3 split pack: nt = CSEG_NT_GROUP needs output channels %% 48");
        return cseg_sb16::pack(w, Cout, Cin, transpose_flip, 3, arith, amax_w, wp, stream);
    }
    if (use_sb16(conv_out)) return cseg_sb16::pack(w, Cout, Cin, transpose_flip, sb16_nt(conv_out, NT), arith, amax_w, wp, stream);
    if (NT == 0) NT = pick_nt(conv_out);
    CSEG_REQUIRE(conv_in % 16 == 0 && NT > 0 && nt_ok(NT, conv_out),
                 "conv3x3_sb: needs input channels %% 16 == 0 and output channels %% 48 == 0 (got %d -> %d)", conv_in, conv_out);
    CSEG_REQUIRE((reinterpret_cast<uintptr_t>(wp) & 15) == 0, "conv3x3_sb_pack_weights: packed buffer must be 16-byte aligned");
    const long total = (long)(conv_out / 16) * steps_of(conv_in) * 64;
    CSEG_REQUIRE(total < 2147483647L, "conv3x3_sb_pack_weights: too large");
    if (arith == CSEG_ARITH_F16X3)
        hipLaunchKernelGGL(pack_weights_sb_kernel<SplitF16x3>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, w, Cout, Cin,
                           transpose_flip, NT, amax_w, (uint4*)wp, (int)total);
    else
        hipLaunchKernelGGL(pack_weights_sb_kernel<SplitBF16x6>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, w, Cout, Cin,
                           transpose_flip, NT, amax_w, (uint4*)wp, (int)total);
    CSEG_CHECK_LAUNCH("conv3x3_sb_pack_weights");
    return 1;
}

extern "C" int cseg_conv3x3_sb_pack_weights(const float* w, int Cout, int Cin, int transpose_flip, void* wp,
                                            cseg_stream_t stream_) {
    return pack_impl(w, Cout, Cin, transpose_flip, 0, CSEG_ARITH_BF16X6, nullptr, wp, (hipStream_t)stream_);
}

// explicit channel tiles per block (3, 6 or 9; must divide conv_out / 16): lets the caller trade the block's reuse of the
// staged patch against the number of blocks (192 channels at 8x32x64: NT = 3 gives 256 blocks, 81 vs 113 us at NT = 6)
extern "C" int cseg_conv3x3_sb_pack_weights_nt(const float* w, int Cout, int Cin, int transpose_flip, int nt, void* wp,
                                               cseg_stream_t stream_) {
    CSEG_REQUIRE(nt == 3 || nt == 6 || nt == 9, "conv3x3_sb_pack_weights_nt: nt must be 3, 6 or 9 (got %d)", nt);
    return pack_impl(w, Cout, Cin, transpose_flip, nt, CSEG_ARITH_BF16X6, nullptr, wp, (hipStream_t)stream_);
}

// nt = 0: the library's channel tiling. arith: CSEG_ARITH_BF16X6 (amax_w may be null) | CSEG_ARITH_F16X3 (amax_w = max|w| bits)
extern "C" int cseg_conv3x3_split_pack(const float* w, int Cout, int Cin, int transpose_flip, int nt, int arith,
                                       const unsigned* amax_w, void* wp, cseg_stream_t stream_) {
    CSEG_REQUIRE(nt == 0 || nt == 3 || nt == 6 || nt == 9 || nt == CSEG_NT_SB8 || nt == CSEG_NT_GROUP, "conv3x3_split_pack: nt must be 0, 3, 6, 9, CSEG_NT_SB8 or CSEG_NT_GROUP (got %d)", nt);
    return pack_impl(w, Cout, Cin, transpose_flip, nt, arith, amax_w, wp, (hipStream_t)stream_);
}

static int fwd_impl(const float* x, const void* wp, const float* bias, const float* addend, int B, int Cin, int Cout, int H, int W, int NT, int arith,
                    const unsigned* amax_x, const unsigned* amax_w, float* y, hipStream_t stream, float4* stats = nullptr) {
    CSEG_REQUIRE(!stats || (!addend && (reinterpret_cast<uintptr_t>(stats) & 15) == 0),
                 "conv3x3 split: the statistics epilogue takes no addend and needs a 16-byte aligned buffer");
    CSEG_REQUIRE(x && wp && y, "conv3x3_sb: null pointer");
    CSEG_REQUIRE(arith_ok(arith) && (arith == CSEG_ARITH_BF16X6 || (amax_x && amax_w)),
                 "conv3x3 split: arithmetic %d needs max|x| and max|w|", arith);
    if (NT == CSEG_NT_SB8) {
        CSEG_REQUIRE((reinterpret_cast<uintptr_t>(wp) & 15) == 0 && (reinterpret_cast<uintptr_t>(y) & 15) == 0,
                     "conv3x3_sb: packed weights / output must be 16-byte aligned and W a multiple of 4");
        CSEG_REQUIRE(!addend, "conv3x3_sb: the 8-row kernel takes no addend");
        return cseg_sb16::fwd8(x, wp, bias, B, Cin, Cout, H, W, arith, amax_x, amax_w, y, stats, stream);
    }
    if (NT == CSEG_NT_GROUP || use_sb16(Cout)) {
        CSEG_REQUIRE((reinterpret_cast<uintptr_t>(wp) & 15) == 0 && (reinterpret_cast<uintptr_t>(y) & 15) == 0,
                     "conv3x3_sb: packed weights / output must be 16-byte aligned and W a multiple of 4");
        CSEG_REQUIRE(NT != CSEG_NT_GROUP || Cout % 48 == 0, "conv3x3_sb: nt = CSEG_NT_GROUP needs output channels %% 48");
        return cseg_sb16::fwd(x, wp, bias, addend, B, Cin, Cout, H, W, NT == CSEG_NT_GROUP ? 3 : sb16_nt(Cout, NT), arith, amax_x, amax_w, y,
                              stats, stream);
    }
    if (NT == 0) NT = pick_nt(Cout);
    CSEG_REQUIRE(B > 0 && H > 0 && W > 0 && Cin > 0 && Cin % 16 == 0 && NT > 0 && nt_ok(NT, Cout),
                 "conv3x3_sb: unsupported shape B=%d Cin=%d Cout=%d %dx%d", B, Cin, Cout, H, W);
    CSEG_REQUIRE((reinterpret_cast<uintptr_t>(wp) & 15) == 0 && (reinterpret_cast<uintptr_t>(y) & 15) == 0,
                 "conv3x3_sb: packed weights / output must be 16-byte aligned and W a multiple of 4");
    const uint4* wq = (const uint4*)wp;
    // CSEG_CONV3X3_SB_GLDS=0: stage B through registers instead of LDS-DMA
    const char* glds_env = getenv("CSEG_CONV3X3_SB_GLDS");       // read per call: tests switch it inside one process
    const bool glds = !(glds_env && atoi(glds_env) == 0);
    int var = sb_variant();                                      // tuning variants of the kernel (see the template comment)
    if (var < 0) var = (long)H * W * 32 * 4 < 2147483647L ? 1 : 0;     // default: buffer-load addressing when the offsets fit
    CSEG_REQUIRE(var == 0 || ((var == 1 || var == 2) && (long)H * W * 32 * 4 < 2147483647L),
                 "conv3x3_sb: unsupported CSEG_CONV3X3_SB_VAR=%d", var);
#define SB_LAUNCH(AR, G, V)                                                                                            \
    switch (NT) {                                                                                                      \
        case 9: return launch_sb<AR, 9, G, V>(x, wq, bias, addend, B, Cin, Cout, H, W, amax_x, amax_w, y, stats, stream);             \
        case 6: return launch_sb<AR, 6, G, V>(x, wq, bias, addend, B, Cin, Cout, H, W, amax_x, amax_w, y, stats, stream);             \
        default: return launch_sb<AR, 3, G, V>(x, wq, bias, addend, B, Cin, Cout, H, W, amax_x, amax_w, y, stats, stream);            \
    }
    if (arith == CSEG_ARITH_F16X3) {
        // LDS-DMA for the weights, buffer-load addressing of the patch when the offsets fit 32 bits; weights staged a filter row
        // (3 K-steps) at a time unless CSEG_CONV3X3_SB_SPS=1
        const char* sps_env = getenv("CSEG_CONV3X3_SB_SPS");
        if (var >= 1 && !(sps_env && atoi(sps_env) == 1)) {
            switch (NT) {
                case 9: return launch_sb<SplitF16x3, 9, true, 1, 0, 3>(x, wq, bias, addend, B, Cin, Cout, H, W, amax_x, amax_w, y, stats, stream);
                case 6: return launch_sb<SplitF16x3, 6, true, 1, 0, 3>(x, wq, bias, addend, B, Cin, Cout, H, W, amax_x, amax_w, y, stats, stream);
                default: return launch_sb<SplitF16x3, 3, true, 1, 0, 3>(x, wq, bias, addend, B, Cin, Cout, H, W, amax_x, amax_w, y, stats, stream);
            }
        }
        if (var >= 1) { SB_LAUNCH(SplitF16x3, true, 1) }
        SB_LAUNCH(SplitF16x3, true, 0)
    }
    if (glds && var >= 1) { SB_LAUNCH(SplitBF16x6, true, 1) }
    if (glds) { SB_LAUNCH(SplitBF16x6, true, 0) }
    SB_LAUNCH(SplitBF16x6, false, 0)
#undef SB_LAUNCH
}

extern "C" int cseg_conv3x3_sb_fwd(const float* x, const void* wp, const float* bias, int B, int Cin, int Cout, int H,
                                   int W, float* y, cseg_stream_t stream_) {
    return fwd_impl(x, wp, bias, nullptr, B, Cin, Cout, H, W, 0, CSEG_ARITH_BF16X6, nullptr, nullptr, y, (hipStream_t)stream_);
}

extern "C" int cseg_conv3x3_sb_fwd_nt(const float* x, const void* wp, const float* bias, int B, int Cin, int Cout, int H,
                                      int W, int nt, float* y, cseg_stream_t stream_) {
    CSEG_REQUIRE(nt == 3 || nt == 6 || nt == 9, "conv3x3_sb_fwd_nt: nt must be 3, 6 or 9 (got %d)", nt);
    return fwd_impl(x, wp, bias, nullptr, B, Cin, Cout, H, W, nt, CSEG_ARITH_BF16X6, nullptr, nullptr, y, (hipStream_t)stream_);
}

extern "C" int cseg_conv3x3_split_fwd(const float* x, const void* wp, const float* bias, int B, int Cin, int Cout, int H, int W,
                                      int nt, int arith, const unsigned* amax_x, const unsigned* amax_w, float* y,
                                      cseg_stream_t stream_) {
    CSEG_REQUIRE(nt == 0 || nt == 3 || nt == 6 || nt == 9 || nt == CSEG_NT_SB8 || nt == CSEG_NT_GROUP, "conv3x3_split_fwd: nt must be 0, 3, 6, 9, CSEG_NT_SB8 or CSEG_NT_GROUP (got %d)", nt);
    return fwd_impl(x, wp, bias, nullptr, B, Cin, Cout, H, W, nt, arith, amax_x, amax_w, y, (hipStream_t)stream_);
}

// The same convolution with the BatchNorm statistics of its OUTPUT produced by the epilogue (cseg_stats.h): stats receives
// [Cout][cseg_conv_stat_segments(0, B, H, W)] float4 = (count, mean, M2) per 64-pixel row segment; cseg_bn_tiles_finalize /
// cseg_bn_tiles_moments turn them into what the BatchNorm that follows needs, so that nobody re-reads y to sum it.
extern "C" int cseg_conv3x3_split_fwd_st(const float* x, const void* wp, const float* bias, int B, int Cin, int Cout, int H, int W,
                                         int nt, int arith, const unsigned* amax_x, const unsigned* amax_w, float* y, float* stats,
                                         cseg_stream_t stream_) {
    CSEG_REQUIRE(nt == 0 || nt == 3 || nt == 6 || nt == 9 || nt == CSEG_NT_SB8 || nt == CSEG_NT_GROUP, "conv3x3_split_fwd_st: nt must be 0, 3, 6, 9, CSEG_NT_SB8 or CSEG_NT_GROUP (got %d)", nt);
    CSEG_REQUIRE(stats, "conv3x3_split_fwd_st: null statistics buffer");
    return fwd_impl(x, wp, bias, nullptr, B, Cin, Cout, H, W, nt, arith, amax_x, amax_w, y, (hipStream_t)stream_,
                    reinterpret_cast<float4*>(stats));
}

// Round 5: y = conv2d(x, w, bias, stride 1, padding = dilation, dilation) (+ addend, + statistics epilogue), dilation 2 or 4 -- the
// 3x3 convolutions of the dilated ResNet stages of DeepLab-V3 (reference lib/models/backbones/resnet/resnet_backbone.py:88-101). f16x3;
// output channel counts whose packed form is the 16-channel-chunk one (cseg_conv3x3_split_plan -> kind CSEG_PACK_C3_16): multiples of
// 64 that are not multiples of 48 (256, 512, ...), or 48 / 192. The SAME packed weights as the undilated operator (pack the
// transposed operator for backward-data). addend / stats nullable (not both).
extern "C" int cseg_conv3x3_split_dil_fwd(const float* x, const void* wp, const float* bias, const float* addend, int B, int Cin, int Cout,
                                          int H, int W, int dil, int arith, const unsigned* amax_x, const unsigned* amax_w, float* y,
                                          float* stats, cseg_stream_t stream_) {
    CSEG_REQUIRE(x && wp && y, "conv3x3 dilated: null pointer");
    CSEG_REQUIRE(!stats || (!addend && (reinterpret_cast<uintptr_t>(stats) & 15) == 0),
                 "conv3x3 dilated: the statistics epilogue takes no addend and needs a 16-byte aligned buffer");
    CSEG_REQUIRE(use_sb16(Cout), "conv3x3 dilated: %d output channels are not packed in the 16-channel-chunk form", Cout);
    CSEG_REQUIRE((reinterpret_cast<uintptr_t>(wp) & 15) == 0 && (reinterpret_cast<uintptr_t>(y) & 15) == 0,
                 "conv3x3 dilated: packed weights / output must be 16-byte aligned");
    return cseg_sb16::fwd_dil(x, wp, bias, addend, B, Cin, Cout, H, W, sb16_nt(Cout, 0), dil, arith, amax_x, amax_w, y,
                              reinterpret_cast<float4*>(stats), (hipStream_t)stream_);
}

// segments per channel of a statistics buffer: kind 0 = 3x3 kernels (stride 1: H, W of the tensor; stride 2: of the OUTPUT),
// kind 1 = 1x1 kernels (H * W flat)
extern "C" size_t cseg_conv_stat_segments(int kind, int B, int H, int W) {
    if (B <= 0 || H <= 0 || W <= 0) return 0;
    if (kind == 0) return (size_t)B * H * ((W + 63) / 64);
    if (kind == 1) return (size_t)B * (((size_t)H * W + 63) / 64);
    return 0;
}

// ---- max|x| of a tensor, accumulated into the record amax_bits[CSEG_AMAX_WORDS] (include/cseg_hip.h; the caller zeroes it) ----
namespace {
__global__ __launch_bounds__(256) void amax_kernel(const float* __restrict__ x, long n, unsigned* __restrict__ amax_bits) {
    __shared__ unsigned red[4];
    const long n4 = n >> 2;
    unsigned m = 0;
    const uint4* x4 = reinterpret_cast<const uint4*>(x);
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const uint4 v = x4[i];
        m = max(max(m, v.x & 0x7fffffffu), max(v.y & 0x7fffffffu, max(v.z & 0x7fffffffu, v.w & 0x7fffffffu)));
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) m = max(m, reinterpret_cast<const unsigned*>(x)[(n4 << 2) + threadIdx.x] & 0x7fffffffu);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) amax_publish_block(max(max(red[0], red[1]), max(red[2], red[3])), amax_bits);
}
}  // namespace

extern "C" int cseg_amax_f32(const float* x, long n, unsigned* amax_bits, cseg_stream_t stream_) {
    CSEG_REQUIRE(x && amax_bits && n > 0, "amax: null pointer / empty tensor");
    CSEG_REQUIRE((reinterpret_cast<uintptr_t>(x) & 15) == 0, "amax: tensor must be 16-byte aligned");
    long blocks = ((n >> 2) + 256 * 8 - 1) / (256 * 8);
    blocks = blocks < 1 ? 1 : (blocks > 2048 ? 2048 : blocks);
    hipLaunchKernelGGL(amax_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream_, x, n, amax_bits);
    CSEG_CHECK_LAUNCH("amax_kernel");
    return 1;
}

// The same convolution with a tensor of the output's shape added in the epilogue (y = conv(x) + bias + addend; addend may be null):
// one pass less than a separate elementwise add. 16-byte aligned like y. Not with nt = CSEG_NT_SB8.
extern "C" int cseg_conv3x3_split_fwd_add(const float* x, const void* wp, const float* bias, const float* addend, int B, int Cin,
                                          int Cout, int H, int W, int nt, int arith, const unsigned* amax_x, const unsigned* amax_w,
                                          float* y, cseg_stream_t stream_) {
    CSEG_REQUIRE(nt == 0 || nt == 3 || nt == 6 || nt == 9 || nt == CSEG_NT_GROUP, "conv3x3_split_fwd_add: nt must be 0, 3, 6, 9 or CSEG_NT_GROUP (got %d)", nt);
    CSEG_REQUIRE((reinterpret_cast<uintptr_t>(addend) & 15) == 0, "conv3x3_split_fwd_add: addend must be 16-byte aligned");
    return fwd_impl(x, wp, bias, addend, B, Cin, Cout, H, W, nt, arith, amax_x, amax_w, y, (hipStream_t)stream_);
}
