// Weight gradient of the 3x3 / stride 1 / pad 1 convolution on the BF16 matrix cores with split operands ("bf16x6",
// see conv3x3_sb.hip for the arithmetic):  dW[co][ci][ky][kx] = sum_{b,y,x} dy[b][co][y][x] * x[b][ci][y+ky-1][x+kx-1].
// GEMM per tap: M = co, N = ci, K = pixels -- the contraction index is the contiguous dimension of both NCHW operands,
// so a lane's fragment (8 consecutive k) is 8 consecutive pixels of one channel: one aligned ds_read_b128 from an LDS
// image [piece][channel][pixel] of bf16.
//
// Block = 8 waves = a 48 (co) x 64 (ci) channel block x all 9 taps, fed with row segments of 64 pixels:
//   wave = (ci tile = wave & 3, K-slice = wave >> 2); a K-slice is 32 pixels = one K-step of v_mfma_f32_16x16x32_bf16.
//   Per row-step a wave loads 3 co tiles x 3 pieces of dy (9 reads) and, per filter row ky, 3 pieces of its x row
//   (one b128 + one b32 each): the three horizontal taps are the same 10 pixels shifted by 0 / 1 / 2 elements, built in
//   registers (v_alignbit for the odd shift, a register rename for the even one) instead of three shifted LDS copies.
//   27 accumulators (9 taps x 3 co tiles), 162 MFMAs per wave and row-step, term-major so that 9 independent
//   accumulators separate two MFMAs on the same one.
//   x rows live in a 3-slot ring per K-slice (row r in slot (r + 3) % 3): walking down a run of rows stages ONE new x row
//   and one dy row per step; both are fetched into registers before the MFMAs of the current row and split + written
//   to LDS after them (two barriers per row-step; the image is single-buffered: 113 KB).
//   At the end the two K-slices of a channel tile are summed through LDS and the block writes its partial
//   [split][tap][co][ci]; splits are summed in a fixed order by a second kernel: deterministic, no atomics.
// Work split: unit = (image, 64-pixel column segment, run of ROWS_PER_UNIT rows); split s takes units s, s + n_split, ...
// Measured on MI355X (tools/conv3x3_sb_wrw_probe.py, profiles/r02_conv3x3_split_bf16_wrw_probe.jsonl), kernel + reduction:
// 48 ch @8x128x256 108 us (fp32-MFMA kernel 164, MIOpen 201); 96 ch @8x64x128 112 us (152 / 148); 720 ch @8x128x256
// 17.6 ms vs MIOpen 19.5 ms -- at 720 channels the 48x64 channel block re-reads both operands 12-15 times and the kernel
// is bandwidth-bound; a wider block / XCD-local ordering of the blocks that share pixels is the next step.
// Parity vs fp64 and determinism: tests/test_gpu_conv3x3_sb.py. Host side: opt-in (kernels.CONV3X3_SB_WRW) until the
// one-SGD-step goldens have run on it.
// Round 3: version 2 (below) is the default and is written against the arithmetic traits of cseg_split.h -- bf16x6 and f16x3
// (two scaled fp16 pieces, three MFMAs per product; x is scaled by 2^kx, dy by 2^kd while they are split, the partial sums are
// multiplied by 2^-(kx + kd) when they are written). Version 1 stays bf16x6.
#include "cseg_split.h"
#include <stdlib.h>

namespace {

constexpr int CO_B = 48, CI_B = 64, SEG = 64;
constexpr int XP = 40;                 // LDS pitch of one x row (bf16 elements): entry i = pixel x0s - 1 + i, i < 34
constexpr int DP = 72;                 // LDS pitch of one dy row segment (64 pixels + pad)
constexpr int XS_ELEMS = 2 * 3 * CI_B * 3 * XP;      // [slice][piece][ci][ring slot][XP]
constexpr int DS_ELEMS = 3 * CO_B * DP;              // [piece][co][DP]
constexpr int ROWS_PER_UNIT = 8;

__device__ __forceinline__ int xs_idx(int s, int p, int ci, int slot, int i) { return (((s * 3 + p) * CI_B + ci) * 3 + slot) * XP + i; }
__device__ __forceinline__ int ds_idx(int p, int co, int i) { return (p * CO_B + co) * DP + i; }

__device__ __forceinline__ void split8w(const float (&v)[8], uint4& h, uint4& m, uint4& l) {
    uint4 c[3];
    split_cells8<SplitBF16x6>(v, 1.f, c);
    h = c[0]; m = c[1]; l = c[2];
}

__global__ __launch_bounds__(512, 1) void conv3x3_sb_wrw_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                                int B, int Cin, int Cout, int H, int W, int n_split,
                                                                float* __restrict__ partial) {
    extern __shared__ __attribute__((aligned(16))) unsigned short smem_w[];
    unsigned short* xs = smem_w;
    unsigned short* ds = smem_w + XS_ELEMS;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int sl = wave & 3, ksl = wave >> 2;          // ci tile of the block, K-slice
    const int g = lane >> 4, n = lane & 15;
    int blk = blockIdx.x;
    const int split = blk % n_split; blk /= n_split;
    const int n_cib = (Cin + CI_B - 1) / CI_B;
    const int cib = blk % n_cib;
    const int cob = blk / n_cib;
    const size_t plane = (size_t)H * W;
    const int segs = W / SEG;
    const int runs = (H + ROWS_PER_UNIT - 1) / ROWS_PER_UNIT;
    const int n_units = B * segs * runs;
    const bool tile_ok = cib * CI_B + sl * 16 < Cin;   // ragged last channel block: this wave has no tile

    f32x4 acc[9][3];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int c = 0; c < 3; ++c) acc[t][c] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // ---- staging: x row items (slice, ci, octet of 8 entries; 5 octets cover the 34 entries), dy row items (co, octet)
    float xpre[2][8], dpre[8];
    auto x_item = [&](int u, int& s, int& ci, int& q) {
        const int item = tid + 512 * u;                // 640 items
        q = item % 5;
        ci = (item / 5) % CI_B;
        s = item / (5 * CI_B);                         // 0, 1 (>= 2: no item)
    };
    auto x_issue = [&](int b, int x0, int row) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            int s, ci, q;
            x_item(u, s, ci, q);
            const int cic = min(cib * CI_B + ci, Cin - 1), rowc = min(max(row, 0), H - 1), sc = min(s, 1);
            const float* p = x + ((size_t)b * Cin + cic) * plane + (size_t)rowc * W;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int px = x0 + 32 * sc - 1 + 8 * q + j;
                xpre[u][j] = p[min(max(px, 0), W - 1)];                 // raw; masked when it is stored
            }
        }
    };
    auto x_store = [&](int x0, int row) {
        const int slot = (row + 3) % 3;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            int s, ci, q;
            x_item(u, s, ci, q);
            if (s < 2) {
                const bool ch_ok = cib * CI_B + ci < Cin && row >= 0 && row < H;
                float v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int i = 8 * q + j, px = x0 + 32 * s - 1 + i;
                    v[j] = (ch_ok && i < 34 && px >= 0 && px < W) ? xpre[u][j] : 0.f;
                }
                uint4 h, m, l;
                split8w(v, h, m, l);
                *reinterpret_cast<uint4*>(xs + xs_idx(s, 0, ci, slot, 8 * q)) = h;
                *reinterpret_cast<uint4*>(xs + xs_idx(s, 1, ci, slot, 8 * q)) = m;
                *reinterpret_cast<uint4*>(xs + xs_idx(s, 2, ci, slot, 8 * q)) = l;
            }
        }
    };
    auto d_issue = [&](int b, int x0, int row) {
        const int item = min(tid, CO_B * 8 - 1);       // 384 items: (co, octet)
        const int co = item >> 3, q = item & 7;
        const float* p = dy + (((size_t)b * Cout + cob * CO_B + co) * H + min(row, H - 1)) * W + x0 + 8 * q;
#pragma unroll
        for (int j = 0; j < 8; ++j) dpre[j] = p[j];
    };
    auto d_store = [&](int row) {
        if (tid < CO_B * 8) {
            const int co = tid >> 3, q = tid & 7;
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = row < H ? dpre[j] : 0.f;
            uint4 h, m, l;
            split8w(v, h, m, l);
            *reinterpret_cast<uint4*>(ds + ds_idx(0, co, 8 * q)) = h;
            *reinterpret_cast<uint4*>(ds + ds_idx(1, co, 8 * q)) = m;
            *reinterpret_cast<uint4*>(ds + ds_idx(2, co, 8 * q)) = l;
        }
    };

    // ---- one row-step of this wave: output row `row` of the segment, K-slice ksl
    auto compute = [&](int row) {
        bf16x8 a[3][3];
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int p = 0; p < 3; ++p)
                a[c][p] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(ds + ds_idx(p, c * 16 + n, 32 * ksl + 8 * g)));
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int slot = (row + ky - 1 + 3) % 3;
            bf16x8 bfr[3][3];                          // [kx][piece]
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                const unsigned short* src = xs + xs_idx(ksl, p, sl * 16 + n, slot, 8 * g);
                const uint4 c4 = *reinterpret_cast<const uint4*>(src);              // entries 8g .. 8g+7   (kx = 0)
                const unsigned nx = *reinterpret_cast<const unsigned*>(src + 8);    // entries 8g+8, 8g+9
                const uint4 s1 = make_uint4(__builtin_amdgcn_alignbit(c4.y, c4.x, 16), __builtin_amdgcn_alignbit(c4.z, c4.y, 16),
                                            __builtin_amdgcn_alignbit(c4.w, c4.z, 16), __builtin_amdgcn_alignbit(nx, c4.w, 16));
                const uint4 s2 = make_uint4(c4.y, c4.z, c4.w, nx);
                bfr[0][p] = __builtin_bit_cast(bf16x8, c4);
                bfr[1][p] = __builtin_bit_cast(bf16x8, s1);
                bfr[2][p] = __builtin_bit_cast(bf16x8, s2);
            }
#define SBW_TERM(P, Q)                                                                                          \
    _Pragma("unroll") for (int kx = 0; kx < 3; ++kx) _Pragma("unroll") for (int c = 0; c < 3; ++c)             \
        acc[ky * 3 + kx][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[c][P], bfr[kx][Q], acc[ky * 3 + kx][c], 0, 0, 0);
            SBW_TERM(2, 0)
            SBW_TERM(0, 2)
            SBW_TERM(1, 1)
            SBW_TERM(1, 0)
            SBW_TERM(0, 1)
            SBW_TERM(0, 0)
#undef SBW_TERM
        }
    };

    for (int unit = split; unit < n_units; unit += n_split) {
        int t = unit;
        const int run = t % runs; t /= runs;
        const int seg = t % segs;
        const int b = t / segs;
        const int x0 = seg * SEG, ya = run * ROWS_PER_UNIT, yb = min(ya + ROWS_PER_UNIT, H);
        // prologue: x rows ya-1, ya, ya+1 and dy row ya (the previous unit's last barrier has released the image)
#pragma unroll 1
        for (int r = ya - 1; r <= ya + 1; ++r) {
            x_issue(b, x0, r);
            x_store(x0, r);
        }
        d_issue(b, x0, ya);
        d_store(ya);
        __syncthreads();
#pragma unroll 1
        for (int row = ya; row < yb; ++row) {
            const bool more = row + 1 < yb;
            if (more) {
                x_issue(b, x0, row + 2);
                d_issue(b, x0, row + 1);
            }
            if (tile_ok) compute(row);
            __syncthreads();                           // all reads of this row-step are done
            if (more) {
                x_store(x0, row + 2);                  // slot of row - 1
                d_store(row + 1);
            }
            __syncthreads();
        }
    }

    // ---- sum the two K-slices of every channel tile through LDS (the operand image is dead), then write the partial
    float* red = reinterpret_cast<float*>(smem_w);     // [ci tile 4][27][64 lanes][4]
    if (ksl == 1 && tile_ok) {
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int c = 0; c < 3; ++c)
                *reinterpret_cast<f32x4*>(red + (((sl * 27) + t * 3 + c) * 64 + lane) * 4) = acc[t][c];
    }
    __syncthreads();
    if (ksl == 0 && tile_ok) {
        const int ci = cib * CI_B + sl * 16 + n;
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const f32x4 o = *reinterpret_cast<const f32x4*>(red + (((sl * 27) + t * 3 + c) * 64 + lane) * 4);
                const f32x4 v = acc[t][c] + o;
                // D[m = 4g + r][n]: co = cob*48 + 16c + 4g + r, ci = this lane's column
                float* dst = partial + (((size_t)split * 9 + t) * Cout + cob * CO_B + c * 16 + 4 * g) * Cin + ci;
#pragma unroll
                for (int r = 0; r < 4; ++r) dst[(size_t)r * Cin] = v[r];
            }
    }
}

// ---------------------------------------------------------------------------------------------------------
// Version 2: producer / consumer waves. Version 1 above spends a third of a row-step with every wave splitting and
// storing operands between two barriers (no MFMA can issue then) and fetches them with 24 scalar loads per thread.
// Here waves 0-3 only compute -- one ci tile each, the full 64-pixel segment = 2 K-steps, 324 MFMAs per row-step, one
// compute wave per SIMD -- and waves 4-7 only stage: aligned float4 loads (the x row is stored from pixel x0 - 4, so
// that every chunk is 16-byte aligned and lies fully inside or outside the image), split, 8-byte LDS writes into the
// free slot of a 4-slot x ring / the other dy buffer, one tick ahead of the consumers. One barrier per row-step; the
// staging VALU work runs on the same SIMDs underneath the MFMAs. Horizontal taps: entries 8g + kx + 3 .. + 10 of the row
// image = two aligned cells shifted by 3 / 4 / 5 elements (five v_alignbit per piece).
// LDS: x [piece][ci 64][slot 4][72] with a per-channel stride of 148 dwords (= 4 * odd: the 16 lanes of a b128 read
// fall on 16 distinct bank quads), dy [2][piece][co 48][72]: 155 KB.
// Status: index-checked against the numpy lane model; first hardware run pending (CSEG_CONV3X3_SB_WRW_V=2 selects it).
// ---------------------------------------------------------------------------------------------------------
// Round 3: SEGW = 64 (two K-steps per row-step) or 32 (one: the 384-channel maps are 32 columns wide). Per (piece, ci) the x ring
// holds 4 slots of SEGW + 8 entries + 8 pad = 296 / 168 half-words = 148 / 84 dwords (4 x odd either way); a dy row has
// SEGW + 8 half-words (36 / 20 dwords = 4 x odd).
// Pitches (half-words) chosen against the REAL ds_read_b128 lane groups ({0-3,12-15,20-27}, ...: lanes of two K-groups g, g+1
// share a group, MI355X_MICROARCH.md): with address = n * pitch + 16 g bytes a pitch of 6 or 10 (mod 16) x 16 bytes is
// conflict-free; round 2's 592 / 144 bytes (5 / 9 mod 16) cost 12 / 28 extra cycles per 64 lanes (round-3 counters:
// SQ_LDS_BANK_CONFLICT = 55 % of SQ_LDS_IDX_ACTIVE). Three pieces (bf16x6) keep the old pitches: the new ones would not fit 160 KB.
// Two pieces (f16x3), since the end of round 3: the row image starts 8 pixels left of the segment (entry i = pixel x0 - 8 + i, SEGW + 16
// entries per slot), so that the CENTRE tap's fragment (pixels x0 + 8g ..) is one aligned cell with no register work, and the left /
// right taps are that cell shifted by one half-word against the last dword of the cell in front / the first dword of the cell behind
// (4 + 4 v_perm, two ds_read_b32). Before, all three taps were cut out of two cells (5 v_perm + 7 v_mov per piece and filter row:
// the assembled centre fragment had to be copied into a contiguous register tuple) -- 144 -> ~100 VALU instructions per row-step on
// the SIMD that issues the row-step's 162 MFMAs. Three pieces (bf16x6) keep the 4-pixel lead: the wider image would not fit.
__host__ __device__ constexpr int x2_lead(int np) { return np == 2 ? 8 : 4; }
__host__ __device__ constexpr int x2_slot(int np, int segw) { return segw + 2 * x2_lead(np); }        // entries per ring slot
__host__ __device__ constexpr int x2_ch(int np, int segw) { return np == 2 ? (segw == 64 ? 336 : 208) : 4 * (segw + 8) + 8; }
__host__ __device__ constexpr int d2_pitch(int np, int segw) { return np == 2 ? (segw == 64 ? 80 : 48) : segw + 8; }
__host__ __device__ constexpr int x2_elems(int np, int segw) { return np * CI_B * x2_ch(np, segw); }
__host__ __device__ constexpr int d2_elems(int np, int segw) { return 2 * np * CO_B * d2_pitch(np, segw); }

template <int NP, int SEGW>
__device__ __forceinline__ int x2_idx(int p, int ci, int slot, int i) { return (p * CI_B + ci) * x2_ch(NP, SEGW) + slot * x2_slot(NP, SEGW) + i; }
template <int NP, int SEGW>
__device__ __forceinline__ int d2_idx(int buf, int p, int co, int i) { return ((buf * NP + p) * CO_B + co) * d2_pitch(NP, SEGW) + i; }

// ABL != 0: ABLATION builds for timing experiments only (results are wrong): bit 0 = the consumers skip their MFMAs, bit 1 = the
// loaders store truncated bits instead of splitting (no conversion arithmetic), bit 2 = the loaders skip the global loads, bit 3 =
// the loaders do not stage anything after a unit's prologue (consumer-only time).
// (The ablation switches are a template parameter that the library no longer instantiates: timing experiments only, round 4.)
// RAGGED (round 5): widths that are not multiples of the 32 / 64-pixel row segment (65 x 129 of DeepLab-R101-d8; 130 / 65 / 33 / 17
// of HRNet at 520 x 520). The last segment of a row is partly outside the image and rows are not 16-byte aligned any more: the
// loaders fetch the four pixels of a chunk one by one from columns clamped into the row and zero what lies outside (as they already
// do for the halo columns); the consumers see full zero-padded segments and do not change.
// (`bid`: the block's index inside THIS layer's grid -- blockIdx.x for the one-layer launch, blockIdx.x minus the member's first block,
// a multiple of 8, in the grouped launch conv3x3_wrw2_group_kernel below; `smem_w`: the block's dynamic LDS)
template <class AR, int SEGW, int ABL = 0, bool RAGGED = false>
__device__ __forceinline__ void conv3x3_sb_wrw2_body(const float* __restrict__ x, const float* __restrict__ dy,
                                                     int B, int Cin, int Cout, int H, int W, int n_split, int rpu,
                                                     int SC, int SI, const unsigned* __restrict__ amax_x,
                                                     const unsigned* __restrict__ amax_dy,
                                                     float* __restrict__ partial, int bid, unsigned short* smem_w) {
    constexpr int NP = AR::NP;
    typedef typename AR::frag_t frag_t;
    unsigned short* xs = smem_w;
    unsigned short* ds = smem_w + x2_elems(NP, SEGW);
    constexpr int XCH = x2_slot(NP, SEGW) / 4, XU = (CI_B * XCH + 255) / 256;  // float4 chunks per x row, per loader thread
    constexpr int DCH = SEGW / 4, DU = (CO_B * DCH + 255) / 256;             // float4 chunks per dy row, per loader thread
    const unsigned ex = AR::SCALED ? split_amax_exp(amax_x) : 141u, ed = AR::SCALED ? split_amax_exp(amax_dy) : 141u;
    const float xscale = split_scale_of(ex), dscale = split_scale_of(ed);      // 1 for the unscaled arithmetic
    // readfirstlane: the role split below must be a SCALAR branch (the wave index is uniform, which the compiler cannot see)
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const bool loader = wave >= 4;
    const int lt = tid - 256;                          // loader thread index
    const int g = lane >> 4, n = lane & 15;
    // XCD-aware block order. Block b runs on XCD b % 8 (observed on gfx950; only speed depends on it). The blocks of one XCD
    // walk the (pixel split, channel block) grid group by group: a group = SC x SI channel blocks of the SAME pixel range, which
    // run side by side on that XCD's CUs and share their dy rows (SI times) and x rows (SC times) through its L2. At 720
    // channels (15 x 12 channel blocks, groups of 5 x 6) a block's share of the HBM traffic drops from 112 to 21 channels.
    const int n_cib = (Cin + CI_B - 1) / CI_B;
    int split, cib, cob;
    {
        const int n_cob = (Cout + CO_B - 1) / CO_B;    // round 6: the last block may hold 16 or 32 channels (Cout % 16 == 0: 64, 128, 256 ...)
        const int n_si = n_cib / SI, gsz = SC * SI, n_groups = n_split * (n_cob / SC) * n_si;
        const int xcd = bid & 7, l = bid >> 3;
        const int grp = (l / gsz) * 8 + xcd, j = l % gsz;
        if (grp >= n_groups) return;                   // the grid is rounded up to 8 x groups-per-XCD x group size
        split = grp % n_split;
        const int st = grp / n_split;
        cib = (st % n_si) * SI + j % SI;
        cob = (st / n_si) * SC + j / SI;
    }
    const size_t plane = (size_t)H * W;
    const int segs = RAGGED ? (W + SEGW - 1) / SEGW : W / SEGW;
    const int runs = (H + rpu - 1) / rpu;
    const int n_units = B * segs * runs;
    const bool tile_ok = !loader && cib * CI_B + wave * 16 < Cin;

    // ---- loader side. x row: 64 ci x 18 chunks of 4 entries (entry i = pixel x0 - 4 + i); dy row: 48 co x 16 chunks.
    // Round 3: everything about a staging item that does not change from row to row is computed ONCE (per kernel: which channel
    // and chunk, its LDS offset; per unit: its byte offset inside the image and whether its columns exist), and the fetch is a
    // global load with a SCALAR base (image + row) and that 32-bit offset -- the per-row cost of an item is the load, a select and
    // the split. (The first
    // version recomputed item / 18, three clamps and a 64-bit address per item and row: ~400 VALU instructions per row-step on
    // the SIMDs that issue the consumers' MFMAs; with all loads removed the kernel ran 22 % faster, tools/ablate_probe.py.)
    int xi_lds[XU], xi_px[XU], xi_ch[XU];          // LDS half-word offset (piece 0, slot 0); pixel offset from x0; channel (clamped)
    bool xi_ok[XU];
#pragma unroll
    for (int u = 0; u < XU; ++u) {
        const int item = lt + 256 * u, itc = min(max(item, 0), CI_B * XCH - 1);
        const int ci = itc / XCH, c = itc - ci * XCH;
        xi_lds[u] = x2_idx<NP, SEGW>(0, ci, 0, 4 * c);
        xi_px[u] = 4 * c - x2_lead(NP);
        xi_ch[u] = min(cib * CI_B + ci, Cin - 1);
        xi_ok[u] = loader && item < CI_B * XCH && cib * CI_B + ci < Cin;
    }
    int di_lds[DU], di_off[DU], di_c4[DU];         // LDS offset (buffer 0, piece 0); element offset of (co, chunk) inside the image; 4 * chunk
    bool di_ok[DU];
#pragma unroll
    for (int u = 0; u < DU; ++u) {
        const int item = lt + 256 * u, itc = min(max(item, 0), CO_B * DCH - 1);
        const int co = itc / DCH, c = itc - co * DCH;
        di_lds[u] = d2_idx<NP, SEGW>(0, 0, co, 4 * c);
        di_c4[u] = 4 * c;
        di_off[u] = min(cob * CO_B + co, Cout - 1) * (int)plane + 4 * c;
        // (channel rows behind Cout in the last block are never staged: whatever their LDS rows hold reaches only accumulator rows
        // that the epilogue does not store -- an output row depends on its own dy channel alone)
        di_ok[u] = loader && item < CO_B * DCH && cob * CO_B + co < Cout;
    }
    // per unit (set by unit_setup): base pointers of the image, byte offsets of the items inside it, column validity
    const float* x_img = x;
    const float* d_img = dy;
    unsigned xu_off[XU], du_off[DU];
    bool xu_ok[XU];
    int xu_px[XU], du_px[DU];                      // RAGGED: column of element 0 of the item's chunk
    auto unit_setup = [&](int b, int x0) {
        x_img = x + (size_t)b * Cin * plane;
        d_img = dy + (size_t)b * Cout * plane;
#pragma unroll
        for (int u = 0; u < XU; ++u) {
            const int px = x0 + xi_px[u];
            if (RAGGED) {
                xu_off[u] = (unsigned)(xi_ch[u] * (int)plane) * (unsigned)sizeof(float);      // the channel row; columns per element
                xu_px[u] = px;
                xu_ok[u] = xi_ok[u];
            } else {
                xu_off[u] = (unsigned)(xi_ch[u] * (int)plane + min(max(px, 0), W - 4)) * (unsigned)sizeof(float);
                xu_ok[u] = xi_ok[u] && px >= 0 && px < W;
            }
        }
#pragma unroll
        for (int u = 0; u < DU; ++u) {
            if (RAGGED) {
                du_off[u] = (unsigned)(di_off[u] - di_c4[u]) * (unsigned)sizeof(float);           // the channel row; columns per element
                du_px[u] = x0 + di_c4[u];
            } else {
                du_off[u] = (unsigned)(di_off[u] + x0) * (unsigned)sizeof(float);
            }
        }
    };
    auto x_load = [&](int row, float4 (&v)[XU]) __attribute__((always_inline)) {
        const float* rowp = x_img + (size_t)min(max(row, 0), H - 1) * W;          // uniform
#pragma unroll
        for (int u = 0; u < XU; ++u) {
            if (ABL & 4) v[u] = make_float4(1.f, 2.f, 3.f, 4.f);
            else if (RAGGED) {
                const char* chrow = reinterpret_cast<const char*>(rowp) + xu_off[u];
                float e[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) e[k] = *reinterpret_cast<const float*>(chrow + (long)min(max(xu_px[u] + k, 0), W - 1) * 4);
                v[u] = make_float4(e[0], e[1], e[2], e[3]);
            } else v[u] = cseg_load_f4(rowp, xu_off[u]);
        }
    };
    auto x_put = [&](int row, int slot, const float4 (&v)[XU]) __attribute__((always_inline)) {
        const bool row_ok = row >= 0 && row < H;
#pragma unroll
        for (int u = 0; u < XU; ++u) {
            if (xi_ok[u]) {
                float4 t = (xu_ok[u] && row_ok) ? v[u] : make_float4(0.f, 0.f, 0.f, 0.f);
                if (RAGGED) {                          // columns left of the image (halo) or right of it (halo / the ragged last segment)
                    const int px = xu_px[u];
                    t.x = (px >= 0 && px < W) ? t.x : 0.f;         t.y = (px + 1 >= 0 && px + 1 < W) ? t.y : 0.f;
                    t.z = (px + 2 >= 0 && px + 2 < W) ? t.z : 0.f; t.w = (px + 3 >= 0 && px + 3 < W) ? t.w : 0.f;
                }
                uint2 cells[NP];
                if (ABL & 2) {
#pragma unroll
                    for (int p = 0; p < NP; ++p)
                        cells[p] = make_uint2(__builtin_bit_cast(unsigned, t.x) >> 16 | (__builtin_bit_cast(unsigned, t.y) & 0xffff0000u),
                                              __builtin_bit_cast(unsigned, t.z) >> 16 | (__builtin_bit_cast(unsigned, t.w) & 0xffff0000u));
                } else split_cells4<AR>(t, xscale, cells);
#pragma unroll
                for (int p = 0; p < NP; ++p)
                    *reinterpret_cast<uint2*>(xs + xi_lds[u] + p * CI_B * x2_ch(NP, SEGW) + slot * x2_slot(NP, SEGW)) = cells[p];
            }
        }
    };
    auto d_load = [&](int row, float4 (&v)[DU]) __attribute__((always_inline)) {
        const float* rowp = d_img + (size_t)min(row, H - 1) * W;
#pragma unroll
        for (int u = 0; u < DU; ++u) {
            if (ABL & 4) v[u] = make_float4(1.f, 2.f, 3.f, 4.f);
            else if (RAGGED) {
                const char* chrow = reinterpret_cast<const char*>(rowp) + du_off[u];
                float e[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) e[k] = *reinterpret_cast<const float*>(chrow + (long)min(du_px[u] + k, W - 1) * 4);
                v[u] = make_float4(e[0], e[1], e[2], e[3]);
            } else v[u] = cseg_load_f4(rowp, du_off[u]);
        }
    };
    auto d_put = [&](int buf, const float4 (&v)[DU]) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < DU; ++u) {
            if (di_ok[u]) {
                uint2 cells[NP];
                float4 t = v[u];
                if (RAGGED) {                          // the ragged last segment: columns right of the image contribute nothing
                    const int px = du_px[u];
                    t.x = px < W ? t.x : 0.f;     t.y = px + 1 < W ? t.y : 0.f;
                    t.z = px + 2 < W ? t.z : 0.f; t.w = px + 3 < W ? t.w : 0.f;
                }
                if (ABL & 2) {
#pragma unroll
                    for (int p = 0; p < NP; ++p)
                        cells[p] = make_uint2(__builtin_bit_cast(unsigned, v[u].x) >> 16 | (__builtin_bit_cast(unsigned, v[u].y) & 0xffff0000u),
                                              __builtin_bit_cast(unsigned, v[u].z) >> 16 | (__builtin_bit_cast(unsigned, v[u].w) & 0xffff0000u));
                } else split_cells4<AR>(t, dscale, cells);
#pragma unroll
                for (int p = 0; p < NP; ++p)
                    *reinterpret_cast<uint2*>(ds + di_lds[u] + (buf * NP + p) * CO_B * d2_pitch(NP, SEGW)) = cells[p];
            }
        }
    };

    // ---- consumer side: output row with x rows in slots s0 (row - 1), s0 + 1, s0 + 2 (mod 4), dy in buffer `buf`.
    // A row-step is (SEGW / 32) x 3 groups (K-step, filter row) of 27 MFMAs x AR::NTERMS. Round 3: software-pipelined by hand --
    // the LDS reads of group i + 1 are issued before the MFMAs of group i (the compiler's own order waited on freshly issued reads
    // ~20 times per row-step, and with ONE consumer wave per SIMD nothing else can issue MFMAs meanwhile).
    struct XRaw { uint4 c0[NP], c1[NP]; unsigned left[NP], right[NP]; };
    auto load_a = [&](int ks, int buf, frag_t (&a)[3][NP]) {
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int p = 0; p < NP; ++p)
                a[c][p] = __builtin_bit_cast(frag_t, *reinterpret_cast<const uint4*>(ds + d2_idx<NP, SEGW>(buf, p, c * 16 + n, 32 * ks + 8 * g)));
    };
    auto load_x = [&](int ks, int slot, XRaw& r) {
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const unsigned short* src = xs + x2_idx<NP, SEGW>(p, wave * 16 + n, slot, 32 * ks + 8 * g);
            if constexpr (NP == 2) {
                r.c1[p] = *reinterpret_cast<const uint4*>(src + 8);           // entries e+8 .. e+15 = pixels x0 + 32 ks + 8 g .. + 7
                r.left[p] = *reinterpret_cast<const unsigned*>(src + 6);      // entries e+6, e+7: the pixel in front in the high half
                r.right[p] = *reinterpret_cast<const unsigned*>(src + 16);    // entries e+16, e+17: the pixel behind in the low half
            } else {
                r.c0[p] = *reinterpret_cast<const uint4*>(src);               // entries e .. e+7    (d0..d3)
                r.c1[p] = *reinterpret_cast<const uint4*>(src + 8);           // entries e+8 .. e+15 (d4..d7)
            }
        }
    };
    auto compute = [&](int s0, int buf, f32x4 (&acc)[9][3]) __attribute__((always_inline)) {
        constexpr int NG = (SEGW / 32) * 3;
        frag_t a[2][3][NP];
        XRaw xr[2];
        load_a(0, buf, a[0]);
        load_x(0, s0 & 3, xr[0]);
#pragma unroll
        for (int grp = 0; grp < NG; ++grp) {
            const int ks = grp / 3, ky = grp % 3;
            if (grp + 1 < NG) {
                const int nks = (grp + 1) / 3, nky = (grp + 1) % 3;
                if (nky == 0) load_a(nks, buf, a[nks & 1]);
                load_x(nks, (s0 + nky) & 3, xr[(grp + 1) & 1]);
                __builtin_amdgcn_sched_barrier(0);         // keep the prefetch in front of this group's MFMAs
            }
            frag_t bfr[3][NP];                             // [kx][piece]
#pragma unroll
            for (int p = 0; p < NP; ++p) {
                if constexpr (NP == 2) {
                    const uint4 c = xr[grp & 1].c1[p];
                    const unsigned lf = xr[grp & 1].left[p], rt = xr[grp & 1].right[p];
                    bfr[0][p] = __builtin_bit_cast(frag_t, make_uint4(__builtin_amdgcn_alignbit(c.x, lf, 16), __builtin_amdgcn_alignbit(c.y, c.x, 16),
                                                                      __builtin_amdgcn_alignbit(c.z, c.y, 16), __builtin_amdgcn_alignbit(c.w, c.z, 16)));
                    bfr[1][p] = __builtin_bit_cast(frag_t, c);
                    bfr[2][p] = __builtin_bit_cast(frag_t, make_uint4(__builtin_amdgcn_alignbit(c.y, c.x, 16), __builtin_amdgcn_alignbit(c.z, c.y, 16),
                                                                      __builtin_amdgcn_alignbit(c.w, c.z, 16), __builtin_amdgcn_alignbit(rt, c.w, 16)));
                } else {
                    const uint4 c0 = xr[grp & 1].c0[p], c1 = xr[grp & 1].c1[p];
                    const unsigned a21 = __builtin_amdgcn_alignbit(c0.z, c0.y, 16), a32 = __builtin_amdgcn_alignbit(c0.w, c0.z, 16),
                                   a43 = __builtin_amdgcn_alignbit(c1.x, c0.w, 16), a54 = __builtin_amdgcn_alignbit(c1.y, c1.x, 16),
                                   a65 = __builtin_amdgcn_alignbit(c1.z, c1.y, 16);
                    bfr[0][p] = __builtin_bit_cast(frag_t, make_uint4(a21, a32, a43, a54));     // entries e+3 .. e+10 (kx = 0)
                    bfr[1][p] = __builtin_bit_cast(frag_t, make_uint4(c0.z, c0.w, c1.x, c1.y)); // entries e+4 .. e+11
                    bfr[2][p] = __builtin_bit_cast(frag_t, make_uint4(a32, a43, a54, a65));     // entries e+5 .. e+12
                }
            }
            // term-major, smallest terms first; nine independent accumulators between two MFMAs on the same one
#pragma unroll
            for (int t = 0; t < AR::NTERMS; ++t)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                    for (int c = 0; c < 3; ++c)
                        if (ABL & 1) acc[ky * 3 + kx][c][0] += (float)a[ks & 1][c][AR::ta(t)][0] + (float)bfr[kx][AR::tb(t)][0];
                        else acc[ky * 3 + kx][c] = AR::mfma(a[ks & 1][c][AR::ta(t)], bfr[kx][AR::tb(t)], acc[ky * 3 + kx][c]);
        }
    };

    auto unit_dims = [&](int unit, int& b, int& x0, int& ya, int& yb) {
        int t = unit;
        const int run = t % runs; t /= runs;
        const int seg = t % segs;
        b = t / segs;
        x0 = seg * SEGW; ya = run * rpu; yb = min(ya + rpu, H);
    };

    // The two roles run separate loops (separate register budgets: staging registers on one side, 27 accumulators on the
    // other) that execute the SAME sequence of barriers: one after the prologue of a unit, one per row. Both walk the rows of a
    // unit four at a time, so that ring slots and dy buffers are compile-time constants (LDS offsets become instruction immediates).
    if (loader) {
        for (int unit = split; unit < n_units; unit += n_split) {
            int b, x0, ya, yb;
            unit_dims(unit, b, x0, ya, yb);
            unit_setup(b, x0);
            float4 xv[XU], dv[DU];
            {   // prologue tick (the previous unit's last barrier has released the image): x rows ya-1, ya, ya+1 -> slots
                // 0, 1, 2; dy row ya -> buffer 0; then the loads of the first steady tick are put in flight
                float4 x0v[XU], x1v[XU];
                x_load(ya - 1, x0v);
                x_load(ya, x1v);
                x_load(ya + 1, xv);
                d_load(ya, dv);
                x_put(ya - 1, 0, x0v);
                x_put(ya, 1, x1v);
                x_put(ya + 1, 2, xv);
                d_put(0, dv);
            }
            if (ya + 1 < yb) {
                x_load(ya + 2, xv);
                d_load(ya + 1, dv);
            }
            __syncthreads();
#pragma unroll 1
            for (int r0 = ya; r0 < yb; r0 += 4) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {          // row r0 + j = step k of the unit with k & 3 == j
                    const int row = r0 + j;
                    if (row < yb) {
                        if (row + 1 < yb) {
                            if (!(ABL & 8)) {
                                x_put(row + 2, (j + 3) & 3, xv);       // the slot that held row - 2
                                d_put((j + 1) & 1, dv);
                            }
                            if (row + 2 < yb) {
                                x_load(row + 3, xv);
                                d_load(row + 2, dv);
                            }
                        }
                        __syncthreads();
                    }
                }
            }
        }
    } else {
        f32x4 acc[9][3];
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int c = 0; c < 3; ++c) acc[t][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int unit = split; unit < n_units; unit += n_split) {
            int b, x0, ya, yb;
            unit_dims(unit, b, x0, ya, yb);
            __syncthreads();
#pragma unroll 1
            for (int r0 = ya; r0 < yb; r0 += 4) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (r0 + j < yb) {
                        if (tile_ok) compute(j, j & 1, acc);
                        __syncthreads();
                    }
                }
            }
        }
        if (tile_ok) {
            const int ci = cib * CI_B + wave * 16 + n;
            const float unscale = split_unscale_of(ex) * split_unscale_of(ed);
#pragma unroll
            for (int t = 0; t < 9; ++t)
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    // D[m = 4g + r][n]: co = cob*48 + 16c + 4g + r, ci = this lane's column
                    if (cob * CO_B + c * 16 >= Cout) continue;             // the last block of a channel count that is not a multiple of 48
                    float* dst = partial + (((size_t)split * 9 + t) * Cout + cob * CO_B + c * 16 + 4 * g) * Cin + ci;
#pragma unroll
                    for (int r = 0; r < 4; ++r) dst[(size_t)r * Cin] = acc[t][c][r] * unscale;
                }
        }
    }
}

template <class AR, int SEGW, int ABL = 0, bool RAGGED = false>
__global__ __launch_bounds__(512, 1) void conv3x3_sb_wrw2_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                                 int B, int Cin, int Cout, int H, int W, int n_split, int rpu,
                                                                 int SC, int SI, const unsigned* __restrict__ amax_x,
                                                                 const unsigned* __restrict__ amax_dy,
                                                                 float* __restrict__ partial) {
    extern __shared__ __attribute__((aligned(16))) unsigned short smem_w2[];
    conv3x3_sb_wrw2_body<AR, SEGW, ABL, RAGGED>(x, dy, B, Cin, Cout, H, W, n_split, rpu, SC, SI, amax_x, amax_dy, partial, (int)blockIdx.x, smem_w2);
}

// dW[co][ci][tap] = sum over splits of partial[split][tap][co][ci], fixed order (same scheme as conv3x3.hip)
__device__ __forceinline__ void sb_wrw_reduce_body(const float* __restrict__ partial, int n_split, int Cout, int Cin,
                                                   float* __restrict__ dw, int bid) {
    __shared__ float red[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int e = bid * 64 + lane;                         // e = (tap * Cout + co) * Cin + ci
    const int total = 9 * Cout * Cin;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (e < total) {
        int sp = wave;
        for (; sp + 12 < n_split; sp += 16) {
            s0 += partial[(size_t)sp * total + e];
            s1 += partial[(size_t)(sp + 4) * total + e];
            s2 += partial[(size_t)(sp + 8) * total + e];
            s3 += partial[(size_t)(sp + 12) * total + e];
        }
        for (; sp < n_split; sp += 4) s0 += partial[(size_t)sp * total + e];
    }
    red[wave][lane] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (wave == 0 && e < total) {
        const float v = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
        const int ci = e % Cin, rest = e / Cin;
        const int co = rest % Cout, tap = rest / Cout;
        dw[((size_t)co * Cin + ci) * 9 + tap] = v;
    }
}
__global__ __launch_bounds__(256) void sb_wrw_reduce_kernel(const float* __restrict__ partial, int n_split, int Cout, int Cin,
                                                            float* __restrict__ dw) {
    sb_wrw_reduce_body(partial, n_split, Cout, Cin, dw, (int)blockIdx.x);
}

int sb_wrw_version() {
    const char* e = getenv("CSEG_CONV3X3_SB_WRW_V");
    return e && atoi(e) == 1 ? 1 : 2;      // default: the producer / consumer version (13.2 vs 17.6 ms at 720 channels, 76-82 vs 108-114 us on the branches)
}

// version 2: 64-pixel row segments, or 32-pixel ones when the width is 32 mod 64 (the 384-channel maps of HRNet-W48: 16 x 32);
// runs of 16 rows (8 on maps lower than 32 rows, so that the 16 x 32 maps still give every split a unit)
int wrw2_seg(int W) { return W % 64 == 0 ? 64 : (W % 32 == 0 ? 32 : (W > 48 ? 64 : 32)); }      // ragged widths: the segment that wastes less
bool wrw2_ragged(int W) { return W % wrw2_seg(W) != 0; }
int wrw2_rpu(int H) { return H >= 32 ? 16 : 8; }      // 8-row runs at 32 rows cost the 192-channel maps 52 -> 62 us (more splits, more partials)

void sb_wrw_group(int n_cob, int n_cib, int& SC, int& SI);

// Number of pixel splits. Version 1: ~3 blocks per CU in total. Version 2 (round 3): the blocks of a split run group by group on the
// XCDs (conv3x3_sb_wrw2_kernel), block b on XCD b % 8, so what counts is how many ROUNDS of 32 CUs the busiest XCD needs times the
// units a block walks per round, plus the cost of writing and re-reading one more set of partials. At 720 channels (30-block
// groups, 6 per split, 256 units): 5 splits = 30 groups = 4 rounds x 52 units, 4 splits = 24 groups = 3 rounds x 64 units (-8 %).
int sb_wrw_splits(int B, int Cin, int Cout, int H, int W, int arith) {
    const bool v2 = arith == CSEG_ARITH_F16X3 || sb_wrw_version() == 2;
    const int rpu = v2 ? wrw2_rpu(H) : ROWS_PER_UNIT;
    const int seg = v2 ? wrw2_seg(W) : SEG;
    const int units = B * ((W + seg - 1) / seg) * ((H + rpu - 1) / rpu);
    const int n_cib = (Cin + CI_B - 1) / CI_B, n_cob = (Cout + CO_B - 1) / CO_B;
    const int pairs = n_cib * n_cob;
    if (!v2) {
        int n = (768 + pairs - 1) / pairs;
        if (n > 256) n = 256;                        // bounds the partial buffer (256 x 9 x Cout x Cin floats)
        if (n > units) n = units;
        return n < 1 ? 1 : n;
    }
    int SC, SI;
    sb_wrw_group(n_cob, n_cib, SC, SI);
    const int gsz = SC * SI, groups_per_split = (n_cob / SC) * (n_cib / SI);
    const double unit_us = rpu * (seg / 32) * 1.15;                                  // ~1.15 us per K-step row (measured: 2.3 us per 64-pixel row-step)
    const double split_us = 2.0 * 9.0 * Cin * Cout * sizeof(float) / 4.0e6;          // partials written + read back at ~4 TB/s
    const char* force = getenv("CSEG_SB_WRW_SPLITS");                                // tuning runs (tools/wrw_split_probe.py)
    if (force && atoi(force) > 0) return atoi(force) < units ? atoi(force) : units;
    auto cost_of = [&](int n) {
        const long per_xcd = ((long)n * groups_per_split + 7) / 8 * gsz;             // blocks of the busiest XCD
        const long rounds = (per_xcd + 31) / 32;
        return rounds * ((double)((units + n - 1) / n) * unit_us + 5.0) + n * split_us;   // 5 us per block: prologue + partial write
    };
    const int n_max = units < 256 ? units : 256;
    double best_cost = 1e30;
    for (int n = 1; n <= n_max; ++n) best_cost = cost_of(n) < best_cost ? cost_of(n) : best_cost;
    for (int n = 1; n <= n_max; ++n)
        if (cost_of(n) <= 1.03 * best_cost) return n;                                // near-ties: the fewest partials
    return n_max;
}

}  // namespace

// the larger of the two arithmetics' needs (they may split differently: version 1 has no 32-pixel segments)
extern "C" size_t cseg_conv3x3_sb_wrw_ws_floats(int B, int Cin, int Cout, int H, int W) {
    if (B <= 0 || Cin <= 0 || Cout <= 0 || H <= 0 || W <= 0 || Cin % 16 || Cout % 16) return 0;
    // output channel counts that are not multiples of 48 (64, 128, 256 ...: a partly filled last channel block): f16x3, version 2 only
    if (Cout % CO_B) return (size_t)sb_wrw_splits(B, Cin, Cout, H, W, CSEG_ARITH_F16X3) * 9 * Cin * Cout;
    // ragged widths: f16x3 only, and f16x3 always runs version 2 whatever CSEG_CONV3X3_SB_WRW_V says (wrw_impl) -- the size of that
    // launch (with the version switch at 1 this returned 0 and the autograd path raised instead of running)
    if (W % 32) return (size_t)sb_wrw_splits(B, Cin, Cout, H, W, CSEG_ARITH_F16X3) * 9 * Cin * Cout;
    if (W % SEG && sb_wrw_version() != 2) return (size_t)sb_wrw_splits(B, Cin, Cout, H, W, CSEG_ARITH_F16X3) * 9 * Cin * Cout;
    const int a = sb_wrw_splits(B, Cin, Cout, H, W, CSEG_ARITH_BF16X6), b = sb_wrw_splits(B, Cin, Cout, H, W, CSEG_ARITH_F16X3);
    return (size_t)(a > b ? a : b) * 9 * Cin * Cout;
}

namespace {

// group shape of the XCD-aware block order: SC | n_cob, SI | n_cib, SC * SI <= 32 (the CUs of one XCD), fewest distinct
// channels per block
void sb_wrw_group(int n_cob, int n_cib, int& SC, int& SI) {
    SC = SI = 1;
    double best = CO_B + CI_B;
    for (int sc = 1; sc <= n_cob; ++sc) {
        if (n_cob % sc) continue;
        for (int si = 1; si <= n_cib && sc * si <= 32; ++si) {
            if (n_cib % si) continue;
            const double per_block = (double)(sc * CO_B + si * CI_B) / (sc * si);
            if (per_block < best - 1e-9) { best = per_block; SC = sc; SI = si; }
        }
    }
}

template <class AR, int SEGW, int ABL = 0, bool RAGGED = false>
int launch_wrw2(const float* x, const float* dy, int B, int Cin, int Cout, int H, int W, int n_split, const unsigned* amax_x,
                const unsigned* amax_dy, float* ws, hipStream_t stream) {
    const size_t lds2 = sizeof(unsigned short) * (x2_elems(AR::NP, SEGW) + d2_elems(AR::NP, SEGW));
    static bool attr2_set = false;
    if (!attr2_set) {
        if (hipFuncSetAttribute((const void*)(conv3x3_sb_wrw2_kernel<AR, SEGW, ABL, RAGGED>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2) !=
            hipSuccess) {
            cseg_set_error("conv3x3_sb_wrw: cannot raise dynamic LDS to %zu bytes", lds2);
            return 0;
        }
        attr2_set = true;
    }
    const int n_cob = (Cout + CO_B - 1) / CO_B, n_cib = (Cin + CI_B - 1) / CI_B;
    int SC, SI;
    sb_wrw_group(n_cob, n_cib, SC, SI);
    const long n_groups = (long)n_split * (n_cob / SC) * (n_cib / SI);
    const long blocks = ((n_groups + 7) / 8) * 8 * SC * SI;
    CSEG_REQUIRE(blocks < 2147483647L, "conv3x3_sb_wrw: grid too large");
    hipLaunchKernelGGL((conv3x3_sb_wrw2_kernel<AR, SEGW, ABL, RAGGED>), dim3((unsigned)blocks), dim3(512), lds2, stream, x, dy, B, Cin, Cout, H, W,
                       n_split, wrw2_rpu(H), SC, SI, amax_x, amax_dy, ws);
    CSEG_CHECK_LAUNCH("conv3x3_sb_wrw2_kernel");
    return 1;
}

int wrw_impl(const float* x, const float* dy, int B, int Cin, int Cout, int H, int W, int arith, const unsigned* amax_x,
             const unsigned* amax_dy, float* ws, float* dw, hipStream_t stream) {
    CSEG_REQUIRE(x && dy && ws && dw, "conv3x3_sb_wrw: null pointer");
    const bool v2 = arith == CSEG_ARITH_F16X3 || sb_wrw_version() == 2;
    const bool ragged = v2 && arith == CSEG_ARITH_F16X3 && wrw2_ragged(W);          // round 5: any width (f16x3, version 2)
    CSEG_REQUIRE(B > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0 && Cin % 16 == 0 && Cout % 16 == 0 &&
                     (Cout % CO_B == 0 || (v2 && arith == CSEG_ARITH_F16X3)) && (ragged || W % (v2 ? 32 : SEG) == 0),
                 "conv3x3_sb_wrw: unsupported shape B=%d Cin=%d Cout=%d %dx%d (needs Cin %% 16, Cout %% 48 -- f16x3: Cout %% 16; bf16x6: W %% 32; version 1: W %% 64)",
                 B, Cin, Cout, H, W);
    CSEG_REQUIRE(arith == CSEG_ARITH_BF16X6 || (arith == CSEG_ARITH_F16X3 && amax_x && amax_dy),
                 "conv3x3 split wrw: arithmetic %d needs max|x| and max|dy|", arith);
    const int n_split = sb_wrw_splits(B, Cin, Cout, H, W, arith);
    const long blocks = (long)n_split * ((Cin + CI_B - 1) / CI_B) * ((Cout + CO_B - 1) / CO_B);
    CSEG_REQUIRE(blocks < 2147483647L && (long)9 * Cin * Cout < 2147483647L, "conv3x3_sb_wrw: grid too large");
    if (v2) {
        CSEG_REQUIRE(ragged || ((reinterpret_cast<uintptr_t>(x) & 15) == 0 && (reinterpret_cast<uintptr_t>(dy) & 15) == 0),
                     "conv3x3_sb_wrw: tensors must be 16-byte aligned");
        CSEG_REQUIRE((long)Cin * H * W * 4 < 2147483647L && (long)Cout * H * W * 4 < 2147483647L,
                     "conv3x3_sb_wrw: one image of x / dy must stay below 2 GiB (32-bit offsets)");
        const bool wide = wrw2_seg(W) == 64;
        const int ok = ragged
                           ? (wide ? launch_wrw2<SplitF16x3, 64, 0, true>(x, dy, B, Cin, Cout, H, W, n_split, amax_x, amax_dy, ws, stream)
                                   : launch_wrw2<SplitF16x3, 32, 0, true>(x, dy, B, Cin, Cout, H, W, n_split, amax_x, amax_dy, ws, stream))
                       : arith == CSEG_ARITH_F16X3
                           ? (wide ? launch_wrw2<SplitF16x3, 64>(x, dy, B, Cin, Cout, H, W, n_split, amax_x, amax_dy, ws, stream)
                                   : launch_wrw2<SplitF16x3, 32>(x, dy, B, Cin, Cout, H, W, n_split, amax_x, amax_dy, ws, stream))
                           : (wide ? launch_wrw2<SplitBF16x6, 64>(x, dy, B, Cin, Cout, H, W, n_split, amax_x, amax_dy, ws, stream)
                                   : launch_wrw2<SplitBF16x6, 32>(x, dy, B, Cin, Cout, H, W, n_split, amax_x, amax_dy, ws, stream));
        if (!ok) return 0;
    } else {
        const size_t lds = sizeof(unsigned short) * (XS_ELEMS + DS_ELEMS);
        static bool attr_set = false;
        if (!attr_set) {
            if (hipFuncSetAttribute((const void*)conv3x3_sb_wrw_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)lds) != hipSuccess) {
                cseg_set_error("conv3x3_sb_wrw: cannot raise dynamic LDS to %zu bytes", lds);
                return 0;
            }
            attr_set = true;
        }
        hipLaunchKernelGGL(conv3x3_sb_wrw_kernel, dim3((unsigned)blocks), dim3(512), lds, stream, x, dy, B, Cin, Cout, H, W,
                           n_split, ws);
        CSEG_CHECK_LAUNCH("conv3x3_sb_wrw_kernel");
    }
    const int total = 9 * Cin * Cout;
    hipLaunchKernelGGL(sb_wrw_reduce_kernel, dim3((total + 63) / 64), dim3(256), 0, stream, ws, n_split, Cout, Cin, dw);
    CSEG_CHECK_LAUNCH("sb_wrw_reduce_kernel");
    return 1;
}


// ---- Round 6: the weight gradients of several independent convolutions in ONE launch (+ one for the fixed-order reductions):
// the layers of one depth of HRNet's parallel branches (reference lib/models/backbones/hrnet/hrnet_backbone.py:262-288). The grid is
// the concatenation of the members' one-layer grids (each a multiple of 8 blocks, so the XCD-aware order of a member is kept), every
// block runs conv3x3_sb_wrw2_body on its member with the split count the one-layer launch would use: bit-identical gradients.
struct WrwGM {
    const float* x; const float* dy; const unsigned* amax_x; const unsigned* amax_dy; float* ws; float* dw;
    int B, Cin, Cout, H, W, n_split, rpu, SC, SI, kind, block0, rblock0;      // kind: bit 0 = 32-pixel segments, bit 1 = ragged width
};
struct WrwGArgs { WrwGM m[CSEG_GROUP_MAX]; int n; };

template <int SEGW>
__global__ __launch_bounds__(512, 1) void conv3x3_wrw2_group_kernel(const WrwGArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned short smem_wg[];
    int mi = 0;
    for (int i = 1; i < a.n; ++i) mi = (int)blockIdx.x >= a.m[i].block0 ? i : mi;
    const WrwGM& M = a.m[mi];
    conv3x3_sb_wrw2_body<SplitF16x3, SEGW, 0, false>(M.x, M.dy, M.B, M.Cin, M.Cout, M.H, M.W, M.n_split, M.rpu, M.SC, M.SI, M.amax_x, M.amax_dy, M.ws,
                                                     (int)blockIdx.x - M.block0, smem_wg);
}
__global__ __launch_bounds__(256) void sb_wrw_reduce_group_kernel(const WrwGArgs a) {
    int mi = 0;
    for (int i = 1; i < a.n; ++i) mi = (int)blockIdx.x >= a.m[i].rblock0 ? i : mi;
    const WrwGM& M = a.m[mi];
    sb_wrw_reduce_body(M.ws, M.n_split, M.Cout, M.Cin, M.dw, (int)blockIdx.x - M.rblock0);
}

template <int SEGW>
int launch_wrw2_group(const WrwGArgs& a, long blocks, hipStream_t stream) {
    const size_t lds = sizeof(unsigned short) * (x2_elems(2, SEGW) + d2_elems(2, SEGW));
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void*)conv3x3_wrw2_group_kernel<SEGW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
            cseg_set_error("conv3x3 group wrw: cannot raise dynamic LDS to %zu bytes", lds);
            return 0;
        }
        attr_set = true;
    }
    hipLaunchKernelGGL(conv3x3_wrw2_group_kernel<SEGW>, dim3((unsigned)blocks), dim3(512), lds, stream, a);
    CSEG_CHECK_LAUNCH("conv3x3_wrw2_group_kernel");
    return 1;
}

int wrw_group_impl(const cseg_wrw_group_member* mem, int n, hipStream_t stream) {
    bool any_ragged = false;
    for (int i = 0; i < n; ++i) {
        const cseg_wrw_group_member& s = mem[i];
        CSEG_REQUIRE(s.x && s.dy && s.ws && s.dw && s.amax_x && s.amax_dy, "conv3x3 group wrw: member %d has a null pointer", i);
        CSEG_REQUIRE(s.B > 0 && s.H > 0 && s.W > 0 && s.Cin > 0 && s.Cout > 0 && s.Cin % 16 == 0 && s.Cout % CO_B == 0,
                     "conv3x3 group wrw: member %d: unsupported shape B=%d Cin=%d Cout=%d %dx%d (needs Cin %% 16, Cout %% 48)", i, s.B, s.Cin, s.Cout,
                     s.H, s.W);
        any_ragged = any_ragged || wrw2_ragged(s.W);
    }
    if (any_ragged) {                               // widths that are not multiples of the row segment: one launch per member (same results)
        for (int i = 0; i < n; ++i)
            if (!wrw_impl(mem[i].x, mem[i].dy, mem[i].B, mem[i].Cin, mem[i].Cout, mem[i].H, mem[i].W, CSEG_ARITH_F16X3, mem[i].amax_x, mem[i].amax_dy,
                          mem[i].ws, mem[i].dw, stream))
                return 0;
        return 1;
    }
    WrwGArgs all;                                   // every member (the reduction launch), and the members of one segment width each
    long rblocks = 0;
    for (int i = 0; i < n; ++i) {
        const cseg_wrw_group_member& s = mem[i];
        CSEG_REQUIRE((reinterpret_cast<uintptr_t>(s.x) & 15) == 0 && (reinterpret_cast<uintptr_t>(s.dy) & 15) == 0,
                     "conv3x3 group wrw: member %d: tensors must be 16-byte aligned", i);
        CSEG_REQUIRE((long)s.Cin * s.H * s.W * 4 < 2147483647L && (long)s.Cout * s.H * s.W * 4 < 2147483647L && (long)9 * s.Cin * s.Cout < 2147483647L,
                     "conv3x3 group wrw: member %d: one image of x / dy must stay below 2 GiB (32-bit offsets)", i);
        WrwGM& m = all.m[i];
        m.x = s.x; m.dy = s.dy; m.amax_x = s.amax_x; m.amax_dy = s.amax_dy; m.ws = s.ws; m.dw = s.dw;
        m.B = s.B; m.Cin = s.Cin; m.Cout = s.Cout; m.H = s.H; m.W = s.W;
        m.n_split = sb_wrw_splits(s.B, s.Cin, s.Cout, s.H, s.W, CSEG_ARITH_F16X3);
        m.rpu = wrw2_rpu(s.H);
        const int n_cob = s.Cout / CO_B, n_cib = (s.Cin + CI_B - 1) / CI_B;
        sb_wrw_group(n_cob, n_cib, m.SC, m.SI);
        m.kind = wrw2_seg(s.W) == 64 ? 0 : 1;
        m.block0 = 0;
        m.rblock0 = (int)rblocks;
        rblocks += (9L * s.Cin * s.Cout + 63) / 64;
        CSEG_REQUIRE(rblocks < 2147483647L, "conv3x3 group wrw: grid too large");
    }
    for (int i = n; i < CSEG_GROUP_MAX; ++i) all.m[i] = all.m[0];
    all.n = n;
    for (int kind = 0; kind < 2; ++kind) {
        WrwGArgs a;
        long blocks = 0;
        int k = 0;
        for (int i = 0; i < n; ++i) {
            if (all.m[i].kind != kind) continue;
            a.m[k] = all.m[i];
            const WrwGM& m = a.m[k];
            const long n_groups = (long)m.n_split * ((m.Cout / CO_B) / m.SC) * (((m.Cin + CI_B - 1) / CI_B) / m.SI);
            a.m[k].block0 = (int)blocks;
            blocks += ((n_groups + 7) / 8) * 8 * m.SC * m.SI;
            CSEG_REQUIRE(blocks < 2147483647L, "conv3x3 group wrw: grid too large");
            ++k;
        }
        if (k == 0) continue;
        for (int i = k; i < CSEG_GROUP_MAX; ++i) a.m[i] = a.m[0];
        a.n = k;
        if (!(kind == 0 ? launch_wrw2_group<64>(a, blocks, stream) : launch_wrw2_group<32>(a, blocks, stream))) return 0;
    }
    hipLaunchKernelGGL(sb_wrw_reduce_group_kernel, dim3((unsigned)rblocks), dim3(256), 0, stream, all);
    CSEG_CHECK_LAUNCH("sb_wrw_reduce_group_kernel");
    return 1;
}

}  // namespace

// == cseg_conv3x3_split_wrw (f16x3) per member: dw_i [Cout, Cin, 3, 3] of conv2d(x_i, w_i, stride 1, padding 1) for the output gradient
// dy_i; ws_i = cseg_conv3x3_sb_wrw_ws_floats(B, Cin, Cout, H, W) floats. Two launches for the whole group.
extern "C" int cseg_conv3x3_split_group_wrw(const cseg_wrw_group_member* mem, int n, int arith, cseg_stream_t stream_) {
    CSEG_REQUIRE(mem && n >= 1 && n <= CSEG_GROUP_MAX, "conv3x3 group wrw: needs 1 .. %d members", CSEG_GROUP_MAX);
    CSEG_REQUIRE(arith == CSEG_ARITH_F16X3, "conv3x3 group wrw: f16x3 arithmetic only (got %d)", arith);
    return wrw_group_impl(mem, n, (hipStream_t)stream_);
}

// ---------------------------------------------------------------------------------------------------------
// Stride 2 (round 3): dW[co][ci][ky][kx] = sum_{b,oy,ox} dy[b][co][oy][ox] * x[b][ci][2 oy + ky - 1][2 ox + kx - 1]   (pad 1, x is
// 2 Ho x 2 Wo). The 52 downsampling convolutions of HRNet-W48's fuse / transition layers ran on MIOpen's NHWC implicit-GEMM
// weight gradient, which brings two layout transposes per call: ~9 ms of the 108 ms step (profiles/r03_step_steady_kernel_stats.csv).
// Same producer / consumer structure, block shape (48 co x 64 ci x 9 taps), partial buffer, XCD-aware block order and reduction
// as version 2 above; what changes is the x image in LDS:
//   * an output row oy needs the EVEN input row 2 oy (ky = 1) and the ODD rows 2 oy - 1, 2 oy + 1 (ky = 0, 2); the odd row
//     2 oy + 1 is used again by oy + 1. Rings: two slots for even rows, three for odd rows; per row-step the loaders stage one of
//     each (twice the x traffic of stride 1 per output row, as the operator demands) and one dy row;
//   * within a row the columns are stored DE-INTERLEAVED: plane 0 holds the even input columns 2 ox (kx = 1), plane 1 the odd
//     ones 2 ox + 1 (kx = 2; kx = 0 is the same plane one entry to the left), entry i of a plane <-> ox = i - 8, so that the
//     fragment of a K-group (8 consecutive ox) is ONE aligned ds_read_b128 for kx = 1 and kx = 2 and that cell shifted by one
//     half-word (4 v_alignbit + the last dword of the cell in front) for kx = 0. The loaders fetch aligned float4 (4 consecutive
//     input columns = 2 even + 2 odd) and write one dword per plane and piece.
// Segments of 32 output columns = one K-step per row-step (with 64 the five-slot image does not fit: 174 KB). f16x3 only.
// LDS: x [piece 2][ci 64][slot 5][plane 2][40] half-words, channel pitch 432 (= 54 x 16 B, 6 mod 16: conflict-free b128 reads),
// dy [2][piece][co 48][48]: 110.6 + 18.4 = 129 KB.
// ---------------------------------------------------------------------------------------------------------
namespace {

constexpr int S2_SEG = 32;
constexpr int S2_PL = S2_SEG + 8;                  // entries of one column-parity plane
constexpr int S2_ROWIMG = 2 * S2_PL;               // half-words of one staged x row
constexpr int S2_XCH = 432;                        // per (piece, ci) pitch: 5 slots x 80 + 32 pad
constexpr int S2_DP = 48;                          // dy row pitch
constexpr int S2_KCH = S2_SEG / 2 + 2;             // float4 chunks of a staged x row: input columns 2 x0 - 8 .. 2 x0 + 63
constexpr int s2_x_elems(int np) { return np * CI_B * S2_XCH; }
constexpr int s2_d_elems(int np) { return 2 * np * CO_B * S2_DP; }

template <class AR>
__global__ __launch_bounds__(512, 1) void conv3x3_sb_wrw_s2_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                                   int B, int Cin, int Cout, int Ho, int Wo, int n_split, int rpu,
                                                                   int SC, int SI, const unsigned* __restrict__ amax_x,
                                                                   const unsigned* __restrict__ amax_dy,
                                                                   float* __restrict__ partial) {
    extern __shared__ __attribute__((aligned(16))) unsigned short smem_w[];
    constexpr int NP = AR::NP;
    typedef typename AR::frag_t frag_t;
    unsigned short* xs = smem_w;
    unsigned short* ds = smem_w + s2_x_elems(NP);
    constexpr int XU = (CI_B * S2_KCH + 255) / 256;      // 5 float4 chunks per loader thread and x row
    constexpr int DCH = S2_SEG / 4, DU = (CO_B * DCH + 255) / 256;
    const unsigned ex = split_amax_exp(amax_x), ed = split_amax_exp(amax_dy);
    const float xscale = split_scale_of(ex), dscale = split_scale_of(ed);
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const bool loader = wave >= 4;
    const int lt = tid - 256;
    const int g = lane >> 4, n = lane & 15;
    const int n_cib = (Cin + CI_B - 1) / CI_B;
    int split, cib, cob;
    {   // XCD-aware block order, see conv3x3_sb_wrw2_kernel
        const int n_cob = (Cout + CO_B - 1) / CO_B;    // (round 6: a partly filled last block, as in the stride-1 kernel)
        const int n_si = n_cib / SI, gsz = SC * SI, n_groups = n_split * (n_cob / SC) * n_si;
        const int xcd = blockIdx.x & 7, l = blockIdx.x >> 3;
        const int grp = (l / gsz) * 8 + xcd, j = l % gsz;
        if (grp >= n_groups) return;
        split = grp % n_split;
        const int st = grp / n_split;
        cib = (st % n_si) * SI + j % SI;
        cob = (st / n_si) * SC + j / SI;
    }
    const int H = 2 * Ho, W = 2 * Wo;
    const size_t plane = (size_t)H * W, oplane = (size_t)Ho * Wo;
    const int segs = Wo / S2_SEG;
    const int runs = (Ho + rpu - 1) / rpu;
    const int n_units = B * segs * runs;
    const bool tile_ok = !loader && cib * CI_B + wave * 16 < Cin;

    auto x_at = [&](int p, int ci, int slot, int pl, int i) { return xs + (p * CI_B + ci) * S2_XCH + slot * S2_ROWIMG + pl * S2_PL + i; };
    auto d_at = [&](int buf, int p, int co, int i) { return ds + ((buf * NP + p) * CO_B + co) * S2_DP + i; };

    // ---- loader side. Chunk k (2 .. 19) of a staged x row = input columns 2 x0 - 16 + 4k .. + 3 -> entries 2k, 2k + 1 of both planes.
    // Item descriptors are computed once, offsets once per unit, loads use a scalar base (see conv3x3_sb_wrw2_kernel).
    int xi_lds[XU], xi_px[XU], xi_ch[XU];
    bool xi_ok[XU];
#pragma unroll
    for (int u = 0; u < XU; ++u) {
        const int item = lt + 256 * u, itc = min(max(item, 0), CI_B * S2_KCH - 1);
        const int ci = itc / S2_KCH, k = itc - ci * S2_KCH + 2;
        xi_lds[u] = (int)(x_at(0, ci, 0, 0, 2 * k) - xs);
        xi_px[u] = 4 * k - 16;
        xi_ch[u] = min(cib * CI_B + ci, Cin - 1);
        xi_ok[u] = loader && item < CI_B * S2_KCH && cib * CI_B + ci < Cin;
    }
    int di_lds[DU], di_off[DU];
    bool di_ok[DU];
#pragma unroll
    for (int u = 0; u < DU; ++u) {
        const int item = lt + 256 * u, itc = min(max(item, 0), CO_B * DCH - 1);
        const int co = itc / DCH, c = itc - co * DCH;
        di_lds[u] = (int)(d_at(0, 0, co, 4 * c) - ds);
        di_off[u] = min(cob * CO_B + co, Cout - 1) * (int)oplane + 4 * c;
        di_ok[u] = loader && item < CO_B * DCH && cob * CO_B + co < Cout;
    }
    const float* x_img = x;
    const float* d_img = dy;
    unsigned xu_off[XU], du_off[DU];
    bool xu_ok[XU];
    auto unit_setup = [&](int b, int x0) {
        x_img = x + (size_t)b * Cin * plane;
        d_img = dy + (size_t)b * Cout * oplane;
#pragma unroll
        for (int u = 0; u < XU; ++u) {
            const int px = 2 * x0 + xi_px[u];
            xu_off[u] = (unsigned)(xi_ch[u] * (int)plane + min(max(px, 0), W - 4)) * (unsigned)sizeof(float);
            xu_ok[u] = xi_ok[u] && px >= 0 && px < W;
        }
#pragma unroll
        for (int u = 0; u < DU; ++u) du_off[u] = (unsigned)(di_off[u] + x0) * (unsigned)sizeof(float);
    };
    auto x_load = [&](int row, float4 (&v)[XU]) __attribute__((always_inline)) {
        const float* rowp = x_img + (size_t)min(max(row, 0), H - 1) * W;
#pragma unroll
        for (int u = 0; u < XU; ++u) v[u] = cseg_load_f4(rowp, xu_off[u]);
    };
    auto x_put = [&](int row, int slot, const float4 (&v)[XU]) __attribute__((always_inline)) {
        const bool row_ok = row >= 0 && row < H;
#pragma unroll
        for (int u = 0; u < XU; ++u) {
            if (xi_ok[u]) {
                const float4 t = (xu_ok[u] && row_ok) ? v[u] : make_float4(0.f, 0.f, 0.f, 0.f);
                uint2 cells[NP];
                split_cells4<AR>(t, xscale, cells);
#pragma unroll
                for (int p = 0; p < NP; ++p) {
                    // half-words (v0, v1 | v2, v3): the even columns v0, v2 and the odd columns v1, v3
                    unsigned short* dst = xs + xi_lds[u] + p * CI_B * S2_XCH + slot * S2_ROWIMG;
                    *reinterpret_cast<unsigned*>(dst) = (cells[p].x & 0xffffu) | (cells[p].y << 16);
                    *reinterpret_cast<unsigned*>(dst + S2_PL) = (cells[p].x >> 16) | (cells[p].y & 0xffff0000u);
                }
            }
        }
    };
    auto d_load = [&](int row, float4 (&v)[DU]) __attribute__((always_inline)) {
        const float* rowp = d_img + (size_t)min(row, Ho - 1) * Wo;
#pragma unroll
        for (int u = 0; u < DU; ++u) v[u] = cseg_load_f4(rowp, du_off[u]);
    };
    auto d_put = [&](int buf, const float4 (&v)[DU]) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < DU; ++u) {
            if (di_ok[u]) {
                uint2 cells[NP];
                split_cells4<AR>(v[u], dscale, cells);
#pragma unroll
                for (int p = 0; p < NP; ++p) *reinterpret_cast<uint2*>(ds + di_lds[u] + (buf * NP + p) * CO_B * S2_DP) = cells[p];
            }
        }
    };

    // ---- consumer side. Step k of a unit (output row ya + k): the even input row sits in slot k & 1, the odd rows 2 oy - 1 and
    // 2 oy + 1 in slots 2 + k % 3 and 2 + (k + 1) % 3. Hand-pipelined like version 2: the reads of filter row ky + 1 are issued
    // before the MFMAs of ky.
    struct XRaw { uint4 e[NP], o1[NP]; unsigned o0[NP]; };
    auto load_x = [&](int slot, XRaw& r) {
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const unsigned short* src = x_at(p, wave * 16 + n, slot, 0, 8 * g);
            r.e[p] = *reinterpret_cast<const uint4*>(src + 8);                    // even columns, ox = 8g .. 8g + 7
            r.o1[p] = *reinterpret_cast<const uint4*>(src + S2_PL + 8);           // odd columns 2 ox + 1
            r.o0[p] = *reinterpret_cast<const unsigned*>(src + S2_PL + 6);        // its left neighbour 2 (8g - 1) + 1 in the high half
        }
    };
    auto compute = [&](int k, int m3, f32x4 (&acc)[9][3]) {
        frag_t a[3][NP];
        XRaw xr[2];
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int p = 0; p < NP; ++p) a[c][p] = __builtin_bit_cast(frag_t, *reinterpret_cast<const uint4*>(d_at(k & 1, p, c * 16 + n, 8 * g)));
        const int s_odd0 = 2 + m3, s_odd1 = 2 + (m3 == 2 ? 0 : m3 + 1), s_even = k & 1;
        load_x(s_odd0, xr[0]);
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            if (ky < 2) {
                load_x(ky == 0 ? s_even : s_odd1, xr[(ky + 1) & 1]);
                __builtin_amdgcn_sched_barrier(0);
            }
            frag_t bfr[3][NP];
#pragma unroll
            for (int p = 0; p < NP; ++p) {
                const uint4 o1 = xr[ky & 1].o1[p];
                bfr[0][p] = __builtin_bit_cast(frag_t, make_uint4(__builtin_amdgcn_alignbit(o1.x, xr[ky & 1].o0[p], 16),
                                                                  __builtin_amdgcn_alignbit(o1.y, o1.x, 16),
                                                                  __builtin_amdgcn_alignbit(o1.z, o1.y, 16),
                                                                  __builtin_amdgcn_alignbit(o1.w, o1.z, 16)));      // 2 ox - 1
                bfr[1][p] = __builtin_bit_cast(frag_t, xr[ky & 1].e[p]);                                           // 2 ox
                bfr[2][p] = __builtin_bit_cast(frag_t, o1);                                                        // 2 ox + 1
            }
#pragma unroll
            for (int t = 0; t < AR::NTERMS; ++t)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                    for (int c = 0; c < 3; ++c)
                        acc[ky * 3 + kx][c] = AR::mfma(a[c][AR::ta(t)], bfr[kx][AR::tb(t)], acc[ky * 3 + kx][c]);
        }
    };

    auto unit_dims = [&](int unit, int& b, int& x0, int& ya, int& yb) {
        int t = unit;
        const int run = t % runs; t /= runs;
        const int seg = t % segs;
        b = t / segs;
        x0 = seg * S2_SEG; ya = run * rpu; yb = min(ya + rpu, Ho);
    };

    if (loader) {
        for (int unit = split; unit < n_units; unit += n_split) {
            int b, x0, ya, yb;
            unit_dims(unit, b, x0, ya, yb);
            unit_setup(b, x0);
            float4 ev[XU], ov[XU], dv[DU];
            {   // prologue tick: even row 2 ya -> slot 0, odd rows 2 ya - 1, 2 ya + 1 -> slots 2, 3, dy row ya -> buffer 0
                float4 o0v[XU];
                x_load(2 * ya - 1, o0v);
                x_load(2 * ya, ev);
                x_load(2 * ya + 1, ov);
                d_load(ya, dv);
                x_put(2 * ya - 1, 2, o0v);
                x_put(2 * ya, 0, ev);
                x_put(2 * ya + 1, 3, ov);
                d_put(0, dv);
            }
            if (ya + 1 < yb) {
                x_load(2 * ya + 2, ev);
                x_load(2 * ya + 3, ov);
                d_load(ya + 1, dv);
            }
            __syncthreads();
            int m3 = 0;                                // k % 3
#pragma unroll 1
            for (int row = ya; row < yb; ++row) {
                const int k = row - ya;
                if (row + 1 < yb) {
                    x_put(2 * row + 2, (k + 1) & 1, ev);                       // even row of the next step
                    x_put(2 * row + 3, 2 + (m3 == 0 ? 2 : m3 - 1), ov);        // odd row 2 (oy + 1) + 1 -> slot 2 + (k + 2) % 3
                    d_put((k + 1) & 1, dv);
                    if (row + 2 < yb) {
                        x_load(2 * row + 4, ev);
                        x_load(2 * row + 5, ov);
                        d_load(row + 2, dv);
                    }
                }
                __syncthreads();
                m3 = m3 == 2 ? 0 : m3 + 1;
            }
        }
    } else {
        f32x4 acc[9][3];
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int c = 0; c < 3; ++c) acc[t][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int unit = split; unit < n_units; unit += n_split) {
            int b, x0, ya, yb;
            unit_dims(unit, b, x0, ya, yb);
            __syncthreads();
            int m3 = 0;
#pragma unroll 1
            for (int row = ya; row < yb; ++row) {
                if (tile_ok) compute(row - ya, m3, acc);
                __syncthreads();
                m3 = m3 == 2 ? 0 : m3 + 1;
            }
        }
        if (tile_ok) {
            const int ci = cib * CI_B + wave * 16 + n;
            const float unscale = split_unscale_of(ex) * split_unscale_of(ed);
#pragma unroll
            for (int t = 0; t < 9; ++t)
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    if (cob * CO_B + c * 16 >= Cout) continue;
                    float* dst = partial + (((size_t)split * 9 + t) * Cout + cob * CO_B + c * 16 + 4 * g) * Cin + ci;
#pragma unroll
                    for (int r = 0; r < 4; ++r) dst[(size_t)r * Cin] = acc[t][c][r] * unscale;
                }
        }
    }
}

// rows per unit: the largest of 16 / 8 / 4 that still gives >= 256 blocks; number of pixel splits like sb_wrw_splits
void s2_plan(int B, int Cin, int Cout, int Ho, int Wo, int& rpu, int& n_split) {
    const int pairs = ((Cin + CI_B - 1) / CI_B) * ((Cout + CO_B - 1) / CO_B);
    int want = (768 + pairs - 1) / pairs;
    if (want > 256) want = 256;
    rpu = 4;
    const char* force = getenv("CSEG_S2_WRW_RPU");            // tests: run lengths the small emulated shapes would not reach
    const int forced = force ? atoi(force) : 0;
    for (int r = 16; r >= 4; r >>= 1) {
        if (forced > 0) { rpu = forced; break; }
        const long units = (long)B * (Wo / S2_SEG) * ((Ho + r - 1) / r);
        if (units * pairs >= 256 || r == 4) { rpu = r; break; }
    }
    const long units = (long)B * (Wo / S2_SEG) * ((Ho + rpu - 1) / rpu);
    n_split = (int)(want < units ? want : units);
    if (n_split < 1) n_split = 1;
}

bool s2_shape_ok(int B, int Cin, int Cout, int Ho, int Wo) {
    return B > 0 && Cin > 0 && Cout > 0 && Ho > 0 && Wo > 0 && Cin % 16 == 0 && Cout % 16 == 0 && Wo % S2_SEG == 0;
}

}  // namespace

// workspace (floats) of cseg_conv3x3_s2_split_wrw; 0 = unsupported shape (needs Cin % 16, Cout % 48, output width % 32)
extern "C" size_t cseg_conv3x3_s2_wrw_ws_floats(int B, int Cin, int Cout, int Ho, int Wo) {
    if (!s2_shape_ok(B, Cin, Cout, Ho, Wo)) return 0;
    int rpu, n_split;
    s2_plan(B, Cin, Cout, Ho, Wo, rpu, n_split);
    return (size_t)n_split * 9 * Cin * Cout;
}

// Weight gradient of conv2d(x, w, stride 2, padding 1), 3x3: x [B, Cin, 2 Ho, 2 Wo], dy [B, Cout, Ho, Wo] -> dw [Cout, Cin, 3, 3].
// arith must be CSEG_ARITH_F16X3 (amax_x / amax_dy: max|x| / max|dy| records). Deterministic (fixed-order reduction of the partials).
extern "C" int cseg_conv3x3_s2_split_wrw(const float* x, const float* dy, int B, int Cin, int Cout, int Ho, int Wo, int arith,
                                         const unsigned* amax_x, const unsigned* amax_dy, float* ws, float* dw,
                                         cseg_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    CSEG_REQUIRE(x && dy && ws && dw, "conv3x3_s2_wrw: null pointer");
    CSEG_REQUIRE(s2_shape_ok(B, Cin, Cout, Ho, Wo), "conv3x3_s2_wrw: unsupported shape B=%d Cin=%d Cout=%d out %dx%d (needs Cin %% 16, Cout %% 16, Wo %% 32)",
                 B, Cin, Cout, Ho, Wo);
    CSEG_REQUIRE(arith == CSEG_ARITH_F16X3 && amax_x && amax_dy, "conv3x3_s2_wrw: f16x3 only (needs max|x| and max|dy|)");
    CSEG_REQUIRE((reinterpret_cast<uintptr_t>(x) & 15) == 0 && (reinterpret_cast<uintptr_t>(dy) & 15) == 0,
                 "conv3x3_s2_wrw: tensors must be 16-byte aligned");
    CSEG_REQUIRE((long)9 * Cin * Cout < 2147483647L, "conv3x3_s2_wrw: too large");
    CSEG_REQUIRE((long)Cin * Ho * Wo * 16 < 2147483647L && (long)Cout * Ho * Wo * 4 < 2147483647L,
                 "conv3x3_s2_wrw: one image of x / dy must stay below 2 GiB (32-bit offsets)");
    int rpu, n_split;
    s2_plan(B, Cin, Cout, Ho, Wo, rpu, n_split);
    const size_t lds = sizeof(unsigned short) * (s2_x_elems(2) + s2_d_elems(2));
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void*)(conv3x3_sb_wrw_s2_kernel<SplitF16x3>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) !=
            hipSuccess) {
            cseg_set_error("conv3x3_s2_wrw: cannot raise dynamic LDS to %zu bytes", lds);
            return 0;
        }
        attr_set = true;
    }
    const int n_cob = (Cout + CO_B - 1) / CO_B, n_cib = (Cin + CI_B - 1) / CI_B;
    int SC, SI;
    sb_wrw_group(n_cob, n_cib, SC, SI);
    const long n_groups = (long)n_split * (n_cob / SC) * (n_cib / SI);
    const long blocks = ((n_groups + 7) / 8) * 8 * SC * SI;
    CSEG_REQUIRE(blocks < 2147483647L, "conv3x3_s2_wrw: grid too large");
    hipLaunchKernelGGL((conv3x3_sb_wrw_s2_kernel<SplitF16x3>), dim3((unsigned)blocks), dim3(512), lds, stream, x, dy, B, Cin, Cout, Ho, Wo,
                       n_split, rpu, SC, SI, amax_x, amax_dy, ws);
    CSEG_CHECK_LAUNCH("conv3x3_sb_wrw_s2_kernel");
    const int total = 9 * Cin * Cout;
    hipLaunchKernelGGL(sb_wrw_reduce_kernel, dim3((total + 63) / 64), dim3(256), 0, stream, ws, n_split, Cout, Cin, dw);
    CSEG_CHECK_LAUNCH("sb_wrw_reduce_kernel");
    return 1;
}

extern "C" int cseg_conv3x3_sb_wrw(const float* x, const float* dy, int B, int Cin, int Cout, int H, int W, float* ws,
                                   float* dw, cseg_stream_t stream_) {
    return wrw_impl(x, dy, B, Cin, Cout, H, W, CSEG_ARITH_BF16X6, nullptr, nullptr, ws, dw, (hipStream_t)stream_);
}

// arith: CSEG_ARITH_BF16X6 | CSEG_ARITH_F16X3 (then amax_x / amax_dy = max|x| / max|dy| bit patterns, cseg_amax_f32); workspace:
// cseg_conv3x3_sb_wrw_ws_floats
extern "C" int cseg_conv3x3_split_wrw(const float* x, const float* dy, int B, int Cin, int Cout, int H, int W, int arith,
                                      const unsigned* amax_x, const unsigned* amax_dy, float* ws, float* dw,
                                      cseg_stream_t stream_) {
    return wrw_impl(x, dy, B, Cin, Cout, H, W, arith, amax_x, amax_dy, ws, dw, (hipStream_t)stream_);
}
