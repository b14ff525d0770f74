// Grouped launch: several independent 3x3 / stride 1 / pad 1 split-operand convolutions (f16x3) as ONE kernel.
// Reference shape of the work: HighResolutionModule runs its parallel branches one after the other in a Python loop
// (lib/models/backbones/hrnet/hrnet_backbone.py:262-288: `for i in range(self.num_branches): x[i] = self.branches[i](x[i])`), i.e.
// at every depth of the residual chains 2-4 convolutions of equal flops and very different shape (HRNet-W48 at batch 8: 48 ch on
// 128 x 256, 96 on 64 x 128, 192 on 32 x 64, 384 on 16 x 32) that do not depend on each other. One launch per branch gives the
// coarse branches too few tiles for 256 CUs (384 ch: 0.17 of the split roof, 192 ch: 0.29) and the step ~2 800 dispatches.
// Here the members of a group share one persistent launch:
//   * work unit = (member, 512-pixel tile, group of 48 output channels); a unit costs Cin / 16 chunk iterations (16 input channels
//     x 9 taps = 5 K-steps), so units of the wide members are 8x heavier than those of the narrow ones;
//   * the tile is 8 rows x 64 columns, 16 x 32 on maps at most 32 wide, 32 x 16 on maps at most 16 wide: the 16 x 32 maps of the
//     384-channel branch are covered exactly (the 4 x 64 tiles of the one-layer kernels leave half of every tile outside the image);
//   * ONE WAVE = 64 pixels x all three channel tiles (the 8-row form of conv3x3_sb16r_kernel): 14 ds_read_b128 per 36 MFMAs, and
//     per chunk iteration 42 KB of patch + 30 KB of packed weights for twice the MFMAs of a 4 x 64 tile (measured on the MI355X,
//     first version of this kernel on 4 x 64 tiles: a chunk iteration costs ~3.3 us whatever the member -- 1.2 us of MFMA issue
//     plus the exposed latency of its loads, split and barrier -- so the work per iteration is the lever);
//   * every XCD owns a contiguous range of each member's tiles (halos and the channel groups that read one patch meet in that
//     XCD's L2) and a queue over its units ordered HEAVY FIRST; the 32 blocks of an XCD pull units with one atomic each (longest
//     processing time first: the heavy units start together, the light ones fill the tail), so MFMA-bound and HBM-bound members
//     share the chip and nobody waits for a launch boundary;
//   * patch of the NEXT chunk iteration fetched under the MFMAs of the current one -- across unit and member boundaries --,
//     weights by LDS-DMA one chunk ahead into a double buffer;
//   * per output element the K order is that of the one-launch kernels (chunk by chunk, five K-steps, term-major): outputs are
//     BIT-IDENTICAL to cseg_conv3x3_split_fwd[_st|_add] with nt = CSEG_NT_GROUP, whichever block computes a tile and whatever the
//     tile geometry. (Statistics records: the same (count, mean, M2) per row segment, summed in this kernel's own fixed order --
//     equal to the one-layer kernels' to rounding.)
// Epilogue per member: bias, addend (the residual gradient of a BasicBlock's first convolution), BatchNorm statistics records.
// The queue counters live in `sched` (CSEG_GROUP_SCHED_INTS ints, zero before the first launch); the last block to finish puts
// them back to zero, so the same buffer serves every launch on a stream.
#include "cseg_sb16_tile.h"
#include <type_traits>

namespace {

using cseg_sb16t::NOCT;
using cseg_sb16t::STEPS;

constexpr int GNT = 3;                  // 16-channel tiles per unit (48 output channels)
constexpr int SCHED_STRIDE = 32;        // ints between two counters (128 bytes: one cache line each)
constexpr int GPLANE = 672;             // LDS stride of a (piece, octet) plane in cells: >= 660 (10 x 66) / 612 (18 x 34, 34 x 18), 0 mod 256 bytes
constexpr int GAU = 3;                  // staging items per thread: 2 octets x <= 660 cells over 512 threads

struct GMember {
    const float* x;
    const uint4* wp;
    const float* bias;
    const float* addend;
    float* y;
    float4* stats;
    const unsigned* amax_x;
    const unsigned* amax_w;
    int Cin, Cout, H, W;
    int tiles_x, tiles_y, n_spatial, n_seg;
    int n_cot, n_chunks, per_xcd, unit0;          // per_xcd: tiles per XCD range; unit0: first unit of this member in an XCD's queue
    int geo, pad0, pad1, pad2;                    // tile geometry: (8 << geo) rows x (64 >> geo) columns
};

struct GArgs {
    GMember m[CSEG_GROUP_MAX];
    int n_members, units_per_xcd;
    int* sched;
    int ablate, pad;                    // timing experiments only (CSEG_GROUP_ABLATE, producer / consumer kernel; results are wrong when set):
                                        // 1 no MFMA K-steps, 2 no patch loads, 4 no split / LDS store, 8 no output stores, 16 no weight DMA
};

// BatchNorm statistics of a wave's 64 outputs per channel (cseg_stats.h): one record per (channel, image row, 64-column tile). With
// GEO = 0 the wave holds exactly one such segment; GEO = 1 / 2: 2 / 4 rows of 32 / 16 pixels, one record each.
template <int GEO>
__device__ __forceinline__ void group_stats_emit(const f32x4 (&acc)[4][GNT], const float* __restrict__ bias, int co0, float unscale, int b,
                                                 int y0w, int x0, int H, int W, int tiles_x64, int g, int n, float4* __restrict__ st, size_t T) {
    constexpr int ROWS = 1 << GEO, PER = 4 >> GEO;         // rows of the wave, 16-pixel tiles per row
#pragma unroll
    for (int nt = 0; nt < GNT; ++nt) {
        const float bv = bias ? bias[co0 + nt * 16 + n] : 0.f;
#pragma unroll
        for (int rr = 0; rr < ROWS; ++rr) {
            const int yy = y0w + rr;
            float v[PER * 4];
            float s = 0.f, cnt = 0.f;
#pragma unroll
            for (int k = 0; k < PER; ++k) {
                const f32x4 a = acc[rr * PER + k][nt] * unscale + bv;            // exactly the value the store wrote
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const bool ok = x0 + 16 * k + 4 * g + r < W;
                    v[4 * k + r] = a[r];
                    s += ok ? a[r] : 0.f;
                    cnt += ok ? 1.f : 0.f;
                }
            }
            s += __shfl_xor(s, 16, 64);
            s += __shfl_xor(s, 32, 64);
            cnt += __shfl_xor(cnt, 16, 64);
            cnt += __shfl_xor(cnt, 32, 64);
            const float mean = cnt > 0.f ? s / cnt : 0.f;
            float m2 = 0.f;
#pragma unroll
            for (int k = 0; k < PER; ++k)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const bool ok = x0 + 16 * k + 4 * g + r < W;
                    const float d = v[4 * k + r] - mean;
                    m2 += ok ? d * d : 0.f;
                }
            m2 += __shfl_xor(m2, 16, 64);
            m2 += __shfl_xor(m2, 32, 64);
            if (g == 0 && yy < H) st[(size_t)(nt * 16 + n) * T + ((size_t)b * H + yy) * tiles_x64 + x0 / 64] = make_float4(cnt, mean, m2, 0.f);
        }
    }
}

template <class AR>
__global__ __launch_bounds__(512, 1) void conv3x3_group_kernel(const GArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint4 smem_grp[];
    __shared__ int next_unit;
    __shared__ float scales[CSEG_GROUP_MAX][2];
    constexpr int NP = AR::NP;
    constexpr int A_CELLS = NP * NOCT * GPLANE;
    constexpr int BSTEP = GNT * NP * 64;            // uint4 per K-step
    constexpr int BCHUNK = STEPS * BSTEP;           // uint4 per 16-channel chunk
    uint4* As = smem_grp;                           // [2][piece NP][octet 2][GPLANE]
    uint4* Bs = smem_grp + 2 * A_CELLS;             // [2][BCHUNK]

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int g = lane >> 4, n = lane & 15;
    const int q = blockIdx.x & 7;                   // block b runs on XCD b % 8 (observed on gfx950; only speed depends on it)
    int* head = a.sched + q * SCHED_STRIDE;

    // per-member scales of the f16x3 split (max|x|, max|w| records -> powers of two), once per block
    for (int i = 0; i < a.n_members; ++i) {
        const unsigned ex = split_amax_exp(a.m[i].amax_x), ew = split_amax_exp(a.m[i].amax_w);      // every thread (shuffles inside)
        if (tid == 0) {
            scales[i][0] = split_scale_of(ex);
            scales[i][1] = split_unscale_of(ex) * split_unscale_of(ew);
        }
    }

    // thread 0: the next unit of this XCD's queue that names a tile inside the image batch, or -1
    auto grab = [&]() -> int {
        for (;;) {
            const int u = cseg_counter_add(head, 1);
            if (u >= a.units_per_xcd) return -1;
            int mi = 0;
            for (int i = 1; i < a.n_members; ++i) mi = u >= a.m[i].unit0 ? i : mi;
            const int tl = (u - a.m[mi].unit0) / a.m[mi].n_cot;
            if (q * a.m[mi].per_xcd + tl < a.m[mi].n_spatial) return u;
        }
    };

    struct Unit {                                   // everything a unit needs; uniform over the block
        const float* x;
        const uint4* wbase;
        float* y;
        const float* bias;
        const float* addend;
        float4* stats;
        int Cin, Cout, H, W, n_seg, n_chunks, geo, cot, b, y0, x0;
        float xscale, unscale;
    };
    auto decode = [&](int u, Unit& U) {
        int mi = 0;
        for (int i = 1; i < a.n_members; ++i) mi = u >= a.m[i].unit0 ? i : mi;
        const GMember& M = a.m[mi];
        const int local = u - M.unit0;
        U.cot = local % M.n_cot;
        int t = q * M.per_xcd + local / M.n_cot;
        const int tx = t % M.tiles_x; t /= M.tiles_x;
        const int ty = t % M.tiles_y;
        U.b = t / M.tiles_y;
        U.geo = M.geo;
        U.x0 = tx * (64 >> M.geo); U.y0 = ty * (8 << M.geo);
        U.x = M.x; U.y = M.y; U.bias = M.bias; U.addend = M.addend; U.stats = M.stats;
        U.Cin = M.Cin; U.Cout = M.Cout; U.H = M.H; U.W = M.W; U.n_seg = M.n_seg; U.n_chunks = M.n_chunks;
        U.wbase = M.wp + (size_t)U.cot * M.n_chunks * BCHUNK;
        U.xscale = scales[mi][0]; U.unscale = scales[mi][1];
    };

    if (tid == 0) next_unit = grab();
    __syncthreads();                                // (also publishes `scales`)
    const int u_first = __builtin_amdgcn_readfirstlane(next_unit);

    Unit su;                                        // unit being staged (patch loads, weight DMA)
    // Staging items of a thread: (octet, patch row, patch column), cell 0 = pixel (y0 - 1, x0 - 1); they depend on the tile geometry
    // only and are recomputed when a unit of another geometry is staged.
    float apre[GAU][8];
    int it_r[GAU], it_col[GAU], it_cell[GAU], it_oct8[GAU];
    bool it_in[GAU];
    int s_geo = -1;
    auto stage_geometry = [&](int geo) {
        const int xcols = (64 >> geo) + 2, cells = ((8 << geo) + 2) * xcols;
#pragma unroll
        for (int u = 0; u < GAU; ++u) {
            const int item = tid + 512 * u;
            const int oct = item >= cells ? 1 : 0, rc = min(item - oct * cells, cells - 1);
            it_r[u] = rc / xcols;
            it_col[u] = rc - it_r[u] * xcols;
            it_in[u] = item < 2 * cells;
            it_cell[u] = oct * GPLANE + rc;
            it_oct8[u] = oct * 8;
        }
        s_geo = geo;
    };
    auto a_issue = [&](int chunk) {                 // fp32 patch of (su, chunk) into registers
        const int plane = su.H * su.W;
        const float* xc = su.x + ((size_t)su.b * su.Cin + (size_t)chunk * 16) * plane;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)xc, 0, (int)(16 * plane * sizeof(float)), 0x00020000);
#pragma unroll
        for (int u = 0; u < GAU; ++u) {
            const int yc = min(max(su.y0 + it_r[u] - 1, 0), su.H - 1), xcl = min(max(su.x0 + it_col[u] - 1, 0), su.W - 1);
            const int off = (it_oct8[u] * plane + yc * su.W + xcl) * (int)sizeof(float);
#pragma unroll
            for (int j = 0; j < 8; ++j)
                apre[u][j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, off, j * plane * (int)sizeof(float), 0));
        }
    };
    auto a_store = [&](uint4* dst) {                // ... split and stored (same unit as the a_issue before it)
#pragma unroll
        for (int u = 0; u < GAU; ++u) {
            if (it_in[u]) {
                const int yy = su.y0 + it_r[u] - 1, xx = su.x0 + it_col[u] - 1;
                const bool ok = yy >= 0 && yy < su.H && xx >= 0 && xx < su.W;
                uint4 cells[NP];
                split_cells8_masked<AR>(apre[u], ok, su.xscale, cells);         // zero padding / outside the tensor
#pragma unroll
                for (int p = 0; p < NP; ++p) dst[p * NOCT * GPLANE + it_cell[u]] = cells[p];
            }
        }
    };
    auto b_dma = [&](const uint4* wbase, int chunk, int slot) {   // one 16-channel chunk of packed weights: STEPS * GNT * NP rows of 1 KB
        constexpr int ROWS = STEPS * GNT * NP;
        uint4* dst = Bs + (size_t)slot * BCHUNK;
#pragma unroll
        for (int i = 0; i < (ROWS + 7) / 8; ++i) {
            const int r = wave + 8 * i;
            if (r < ROWS)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wbase + (size_t)chunk * BCHUNK + r * 64 + lane),
                                                 (__attribute__((address_space(3))) void*)(dst + r * 64), 16, 0, 0);
        }
    };

    if (u_first >= 0) {
        decode(u_first, su);
        stage_geometry(su.geo);
        // ---- prologue: patch and weights of the first item
        a_issue(0);
        b_dma(su.wbase, 0, 0);
        a_store(As);
        __syncthreads();

        Unit cu = su;                               // unit being computed
        f32x4 acc[4][GNT];
        // this wave's four 16-pixel tiles inside the patch image of the unit being computed (row, column of the tile's first pixel)
        int c_row[4], c_col[4], c_off[4], c_xcols = 0;
        auto compute_geometry = [&](int geo) {
            c_xcols = (64 >> geo) + 2;
            const int per = 4 >> geo;               // 16-pixel tiles per image row of the tile
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                c_row[mt] = (wave << geo) + mt / per;
                c_col[mt] = 16 * (mt % per);
                c_off[mt] = c_row[mt] * c_xcols + c_col[mt] + n;
            }
        };
        compute_geometry(cu.geo);
        int chunk = 0, buf = 0;
#pragma unroll 1
        for (;;) {
            if (chunk == 0) {
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                    for (int nt = 0; nt < GNT; ++nt) acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (tid == 0) next_unit = grab();                    // read at this unit's last chunk: at least one barrier later (n_chunks >= 2)
            }
            const bool last = chunk == cu.n_chunks - 1;
            bool more = true;
            int nchunk = chunk + 1;
            if (last) {
                const int nu = __builtin_amdgcn_readfirstlane(next_unit);
                more = nu >= 0;
                nchunk = 0;
                if (more) {
                    decode(nu, su);
                    if (su.geo != s_geo) stage_geometry(su.geo);
                }
            }
            if (more) {
                a_issue(nchunk);                                     // fp32 loads of the next item fly under the MFMAs below
                b_dma(su.wbase, nchunk, buf ^ 1);                    // that slot was last read in the previous item (barrier since)
            }
            const uint4* a_base = As + (size_t)buf * A_CELLS;
            const uint4* b_base = Bs + (size_t)buf * BCHUNK + lane;
#pragma unroll
            for (int s = 0; s < STEPS; ++s) {
                // the ninth tap is paired with a tenth that does not exist: its packed weights are zero
                const int tap = min(2 * s + (g >> 1), 8);
                const int ky = tap / 3, kx = tap - 3 * ky;
                const uint4* ap = a_base + (g & 1) * GPLANE + ky * c_xcols + kx;
                typedef typename AR::frag_t frag_t;
                frag_t af[4][NP];
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                    for (int p = 0; p < NP; ++p) af[mt][p] = __builtin_bit_cast(frag_t, ap[p * NOCT * GPLANE + c_off[mt]]);
#pragma unroll
                for (int nt = 0; nt < GNT; ++nt) {
                    frag_t bf[NP];
#pragma unroll
                    for (int p = 0; p < NP; ++p) bf[p] = __builtin_bit_cast(frag_t, b_base[s * BSTEP + (nt * NP + p) * 64]);
#pragma unroll
                    for (int t = 0; t < AR::NTERMS; ++t)
#pragma unroll
                        for (int mt = 0; mt < 4; ++mt) acc[mt][nt] = AR::mfma(af[mt][AR::ta(t)], bf[AR::tb(t)], acc[mt][nt]);
                }
            }
            if (last) {
                const size_t plane = (size_t)cu.H * cu.W;
                float* ybc = cu.y + (size_t)cu.b * cu.Cout * plane;
                const float* abc = cu.addend ? cu.addend + (size_t)cu.b * cu.Cout * plane : nullptr;
                const int co0 = cu.cot * GNT * 16;
                const bool vec = (cu.W & 3) == 0;
#pragma unroll
                for (int nt = 0; nt < GNT; ++nt) {
                    const float bv = cu.bias ? cu.bias[co0 + nt * 16 + n] : 0.f;
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt) {
                        const int yy = cu.y0 + c_row[mt];
                        if (yy < cu.H) {
                            const size_t roff = (size_t)(co0 + nt * 16 + n) * plane + (size_t)yy * cu.W;
                            f32x4 v = acc[mt][nt] * cu.unscale;
                            v += bv;
                            cseg_store_row4(ybc + roff, abc ? abc + roff : nullptr, cu.x0 + c_col[mt] + 4 * g, cu.W, vec, v);
                        }
                    }
                }
                if (cu.stats) {                     // BatchNorm statistics of what was just stored
                    float4* st = cu.stats + (size_t)co0 * cu.n_seg;
                    const int tx64 = (cu.W + 63) / 64, y0w = cu.y0 + (wave << cu.geo);
                    if (cu.geo == 0) group_stats_emit<0>(acc, cu.bias, co0, cu.unscale, cu.b, y0w, cu.x0, cu.H, cu.W, tx64, g, n, st, cu.n_seg);
                    else if (cu.geo == 1) group_stats_emit<1>(acc, cu.bias, co0, cu.unscale, cu.b, y0w, cu.x0, cu.H, cu.W, tx64, g, n, st, cu.n_seg);
                    else group_stats_emit<2>(acc, cu.bias, co0, cu.unscale, cu.b, y0w, cu.x0, cu.H, cu.W, tx64, g, n, st, cu.n_seg);
                }
            }
            if (more) a_store(As + (size_t)(buf ^ 1) * A_CELLS);       // the other patch buffer: last read in the previous item
            __syncthreads();
            if (!more) break;
            if (last) {
                if (su.geo != cu.geo) compute_geometry(su.geo);
                cu = su;
                chunk = 0;
            } else {
                ++chunk;
            }
            buf ^= 1;
        }
    }
    // ---- the last block to get here puts the counters back to zero (every block has made its last, failing, grab by then)
    if (tid == 0) {
        int* done = a.sched + 8 * SCHED_STRIDE;
        if (cseg_counter_add(done, 1) == (int)gridDim.x - 1) {
            for (int i = 0; i < 8; ++i) cseg_counter_store(a.sched + i * SCHED_STRIDE, 0);
            cseg_counter_store(done, 0);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// The 512-pixel form with PRODUCER / CONSUMER waves (CSEG_GROUP_PC, see the launcher). In the kernel above all eight waves walk the
// same phases -- issue the next patch's loads, K-steps, wait for the loads, split + store them, barrier -- and the phases ADD UP
// (measured: ~6.1 us per chunk iteration against 2.4 us of MFMA issue; DESIGN.md section 11.8 found the same sum for the one-layer
// kernels). Here waves 0-3 only compute: one per SIMD, two 64-pixel row segments x three channel tiles each (24 accumulators, per
// K-step 22 ds_read_b128 for 72 MFMAs), and waves 4-7 only stage: the fp32 patch of the NEXT chunk iteration (loads, f16x3 split,
// LDS image) and its weight chunk (LDS-DMA), on the same SIMDs underneath the consumers' MFMAs (separate pipes: a matrix-only and a
// VALU / memory-only wave of one SIMD run concurrently, MI355X_MICROARCH.md "Wave scheduling"). One barrier per chunk iteration,
// the same unit queue, the same arithmetic per output element (bit-identical results).
// ---------------------------------------------------------------------------------------------------------
constexpr int PAU = 6;                  // staging items per PRODUCER thread: 2 octets x <= 660 cells over 256 threads

template <class AR>
__global__ __launch_bounds__(512, 1) void conv3x3_group_pc_kernel(const GArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint4 smem_gpc[];
    __shared__ int next_unit;
    __shared__ float scales[CSEG_GROUP_MAX][2];
    constexpr int NP = AR::NP;
    constexpr int A_CELLS = NP * NOCT * GPLANE;
    constexpr int BSTEP = GNT * NP * 64;            // uint4 per K-step
    constexpr int BCHUNK = STEPS * BSTEP;           // uint4 per 16-channel chunk
    uint4* As = smem_gpc;                           // [2][piece NP][octet 2][GPLANE]
    uint4* Bs = smem_gpc + 2 * A_CELLS;             // [2][BCHUNK]

    // readfirstlane: the role split below must be a SCALAR branch (the wave index is uniform, which the compiler cannot see)
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const bool producer = wave >= 4;
    const int pt = tid - 256;                       // producer thread index
    const int g = lane >> 4, n = lane & 15;
    const int q = blockIdx.x & 7;
    int* head = a.sched + q * SCHED_STRIDE;

    // the first unit's atomic and all 2 n max|.| records are requested before anything waits for one of them (one round trip, not 2 n + 1)
    int pending = 0;
    if (tid == 0) pending = cseg_counter_add(head, 1);
    {
        unsigned rec[2 * CSEG_GROUP_MAX];
#pragma unroll
        for (int i = 0; i < CSEG_GROUP_MAX; ++i) {          // (members beyond n_members are copies of member 0: valid pointers)
            rec[2 * i] = a.m[i].amax_x[(tid & (CSEG_AMAX_SLOTS - 1)) * CSEG_AMAX_STRIDE];
            rec[2 * i + 1] = a.m[i].amax_w[(tid & (CSEG_AMAX_SLOTS - 1)) * CSEG_AMAX_STRIDE];
        }
#pragma unroll
        for (int i = 0; i < CSEG_GROUP_MAX; ++i) {
            unsigned e[2];
#pragma unroll
            for (int k = 0; k < 2; ++k) {                   // split_amax_exp (cseg_split.h) on the value already loaded
                unsigned v = rec[2 * i + k];
#pragma unroll
                for (int o = CSEG_AMAX_SLOTS / 2; o > 0; o >>= 1) v = max(v, (unsigned)__shfl_xor((int)v, o, 64));
                const unsigned ee = v >> 23 & 0xffu;
                e[k] = ee < 15u ? 15u : (ee > 253u ? 253u : ee);
            }
            if (tid == 0) {
                scales[i][0] = split_scale_of(e[0]);
                scales[i][1] = split_unscale_of(e[0]) * split_unscale_of(e[1]);
            }
        }
    }
    // thread 0: `u` = a value taken from this XCD's queue -> the next unit that names a tile inside the image batch, or -1
    auto settle = [&](int u) -> int {
        for (;;) {
            if (u >= a.units_per_xcd) return -1;
            int mi = 0;
            for (int i = 1; i < a.n_members; ++i) mi = u >= a.m[i].unit0 ? i : mi;
            const int tl = (u - a.m[mi].unit0) / a.m[mi].n_cot;
            if (q * a.m[mi].per_xcd + tl < a.m[mi].n_spatial) return u;
            u = cseg_counter_add(head, 1);
        }
    };
    struct Unit {
        const float* x;
        const uint4* wbase;
        float* y;
        const float* bias;
        const float* addend;
        float4* stats;
        int Cin, Cout, H, W, n_seg, n_chunks, geo, cot, b, y0, x0;
        float xscale, unscale;
    };
    auto decode = [&](int u, Unit& U) {
        int mi = 0;
        for (int i = 1; i < a.n_members; ++i) mi = u >= a.m[i].unit0 ? i : mi;
        const GMember& M = a.m[mi];
        const int local = u - M.unit0;
        U.cot = local % M.n_cot;
        int t = q * M.per_xcd + local / M.n_cot;
        const int tx = t % M.tiles_x; t /= M.tiles_x;
        const int ty = t % M.tiles_y;
        U.b = t / M.tiles_y;
        U.geo = M.geo;
        U.x0 = tx * (64 >> M.geo); U.y0 = ty * (8 << M.geo);
        U.x = M.x; U.y = M.y; U.bias = M.bias; U.addend = M.addend; U.stats = M.stats;
        U.Cin = M.Cin; U.Cout = M.Cout; U.H = M.H; U.W = M.W; U.n_seg = M.n_seg; U.n_chunks = M.n_chunks;
        U.wbase = M.wp + (size_t)U.cot * M.n_chunks * BCHUNK;
        U.xscale = scales[mi][0]; U.unscale = scales[mi][1];
    };

    if (tid == 0) next_unit = settle(pending);
    __syncthreads();
    const int u_first = __builtin_amdgcn_readfirstlane(next_unit);
    if (u_first >= 0) {
        Unit su;                                    // unit of the item being staged (producers) / the next one (consumers track it for `cu`)
        decode(u_first, su);
        if (producer) {
            // ---------------- producers: the patches of items k + 1 and k + 2 are in flight while the consumers compute item k.
            // The loads of an item are issued TWO chunk iterations before the consumers need it: a fetch sees the L2 (61 % hits in
            // the first version, rocprofv3 counters in profiles/r06_group_pmc.csv), the fabric and HBM under the load of 256 CUs,
            // and with one iteration of lead every chunk iteration exposed a round trip (MFMA pipe busy 40 % of the kernel).
            // Two register sets (A: items of even distance, B: odd) hold the raw fp32 values; the loop body is written twice so that
            // the sets are compile-time. Loads are issued UNCONDITIONALLY (past the end of the queue: the last item's addresses
            // again), so that the compiler counts outstanding loads exactly and the wait for one set leaves the other in flight
            // (a conditional issue makes s_waitcnt drain both: DESIGN.md section 11.8). Producers issue no LDS-DMA for the same
            // reason (the consumers bring the weights in).
            float apre[2][PAU][8];
            int set_cell[2][PAU];                   // LDS cell of every staged item (piece 0), -1 = no such item
            unsigned set_ok[2];                     // bit u: item u lies inside the image (else zero padding)
            float set_scale[2];
            bool set_valid[2] = {false, false};
            int it_r[PAU], it_col[PAU], it_cell[PAU], it_oct8[PAU];
            bool it_in[PAU];
            int s_geo = -1;
            auto stage_geometry = [&](int geo) {
                const int xcols = (64 >> geo) + 2, cells = ((8 << geo) + 2) * xcols;
#pragma unroll
                for (int u = 0; u < PAU; ++u) {
                    const int item = pt + 256 * u;
                    const int oct = item >= cells ? 1 : 0, rc = min(item - oct * cells, cells - 1);
                    it_r[u] = rc / xcols;
                    it_col[u] = rc - it_r[u] * xcols;
                    it_in[u] = item < 2 * cells;
                    it_cell[u] = oct * GPLANE + rc;
                    it_oct8[u] = oct * 8;
                }
                s_geo = geo;
            };
            // issue cursor: the next item to fetch = (su, i_chunk); i_chunk == su.n_chunks: the first chunk of the unit after su
            int i_chunk = 0;
            bool i_valid = true;
            auto issue = [&](auto SET) {
                constexpr int S = decltype(SET)::value;
                if (i_valid && i_chunk == su.n_chunks) {             // (the consumers are at least one barrier past the grab of that unit)
                    const int nu = __builtin_amdgcn_readfirstlane(next_unit);
                    i_valid = nu >= 0;
                    if (i_valid) {
                        decode(nu, su);
                        if (su.geo != s_geo) stage_geometry(su.geo);
                        i_chunk = 0;
                    }
                }
                const int chunk = i_valid ? i_chunk : su.n_chunks - 1;
                if (a.ablate & 64) {                         // (timing experiments: no address arithmetic either)
                    set_valid[S] = i_valid;
#pragma unroll
                    for (int u = 0; u < PAU; ++u) set_cell[S][u] = -1;
                    if (i_valid) ++i_chunk;
                    return;
                }
                const int plane = su.H * su.W;
                const float* xc = su.x + ((size_t)su.b * su.Cin + (size_t)chunk * 16) * plane;
                const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)xc, 0, (int)(16 * plane * sizeof(float)), 0x00020000);
                unsigned okm = 0;
#pragma unroll
                for (int u = 0; u < PAU; ++u) {
                    const int yy = su.y0 + it_r[u] - 1, xx = su.x0 + it_col[u] - 1;
                    okm |= (yy >= 0 && yy < su.H && xx >= 0 && xx < su.W) ? 1u << u : 0u;
                    const int yc = min(max(yy, 0), su.H - 1), xcl = min(max(xx, 0), su.W - 1);
                    const int off = (it_oct8[u] * plane + yc * su.W + xcl) * (int)sizeof(float);
#pragma unroll
                    for (int j = 0; j < 8; ++j)
                        apre[S][u][j] = (a.ablate & 2) ? 1.f : __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, off, j * plane * (int)sizeof(float), 0));
                    set_cell[S][u] = (it_in[u] && !(a.ablate & 4)) ? it_cell[u] : -1;
                }
                set_ok[S] = okm;
                set_scale[S] = su.xscale;
                set_valid[S] = i_valid;
                if (i_valid) ++i_chunk;
            };
            auto store = [&](auto SET, uint4* dst) {
                constexpr int S = decltype(SET)::value;
#pragma unroll
                for (int u = 0; u < PAU; ++u) {
                    if (set_cell[S][u] >= 0) {
                        uint4 cells[NP];
                        split_cells8_masked<AR>(apre[S][u], (set_ok[S] >> u & 1u) != 0, set_scale[S], cells);
#pragma unroll
                        for (int p = 0; p < NP; ++p) dst[p * NOCT * GPLANE + set_cell[S][u]] = cells[p];
                    }
                }
            };
            typedef std::integral_constant<int, 0> SetA;
            typedef std::integral_constant<int, 1> SetB;
            stage_geometry(su.geo);
            issue(SetA());                          // item 0
            store(SetA(), As);
            issue(SetB());                          // item 1
            __syncthreads();                        // (P) item 0 is staged
            // iteration k (the consumers compute item k from buffer k & 1): fetch item k + 2 into the set that held item k, then
            // split + store item k + 1 (fetched an iteration ago) into the other buffer
#pragma unroll 1
            for (;;) {
                {   // even k: item k + 1 is in set B
                    const bool have_next = set_valid[1];
                    issue(SetA());
                    if (have_next) store(SetB(), As + A_CELLS);
                    if (!(a.ablate & 32)) __syncthreads();                // (I)
                    if (!have_next) break;
                }
                {   // odd k: item k + 1 is in set A
                    const bool have_next = set_valid[0];
                    issue(SetB());
                    if (have_next) store(SetA(), As);
                    if (!(a.ablate & 32)) __syncthreads();                // (I)
                    if (!have_next) break;
                }
            }
        } else {
            // ---------------- consumers: wave w computes the pixels of the 8-wave form's waves 2 w and 2 w + 1
            Unit cu = su;
            f32x4 acc[8][GNT];
            int c_row[8], c_col[8], c_off[8], c_xcols = 0;
            auto compute_geometry = [&](int geo) {
                c_xcols = (64 >> geo) + 2;
                const int per = 4 >> geo;
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt) {
                        c_row[4 * h + mt] = ((2 * wave + h) << geo) + mt / per;
                        c_col[4 * h + mt] = 16 * (mt % per);
                        c_off[4 * h + mt] = c_row[4 * h + mt] * c_xcols + c_col[4 * h + mt] + n;
                    }
            };
            compute_geometry(cu.geo);
            auto b_dma = [&](const uint4* wbase, int chunk, int slot) {      // STEPS * GNT * NP = 30 rows of 1 KB over the four consumer waves
                constexpr int ROWS = STEPS * GNT * NP;
                uint4* dst = Bs + (size_t)slot * BCHUNK;
#pragma unroll
                for (int i = 0; i < (ROWS + 3) / 4; ++i) {
                    const int r = wave + 4 * i;
                    if (r < ROWS)
                        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wbase + (size_t)chunk * BCHUNK + r * 64 + lane),
                                                         (__attribute__((address_space(3))) void*)(dst + r * 64), 16, 0, 0);
                }
            };
            b_dma(cu.wbase, 0, 0);                  // weights of item 0
            __syncthreads();                        // (P)
            int chunk = 0, buf = 0;
#pragma unroll 1
            for (;;) {
                if (chunk == 0) {
#pragma unroll
                    for (int mt = 0; mt < 8; ++mt)
#pragma unroll
                        for (int nt = 0; nt < GNT; ++nt) acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
                    // the atomic that names the unit after this one is ISSUED here and its result looked at only after the K-steps
                    // below: an L2 round trip (~2 us with 32 blocks on one counter) no wave waits for
                    if (tid == 0) pending = cseg_counter_add(head, 1);
                }
                const bool last = chunk == cu.n_chunks - 1;
                bool more = true;
                if (last) {
                    const int nu = __builtin_amdgcn_readfirstlane(next_unit);
                    more = nu >= 0;
                    if (more) decode(nu, su);
                }
                if (more && !(a.ablate & 16)) b_dma(last ? su.wbase : cu.wbase, last ? 0 : chunk + 1, buf ^ 1);      // weights of item k + 1: that slot was last read in item k - 1
                const uint4* a_base = As + (size_t)buf * A_CELLS;
                const uint4* b_base = Bs + (size_t)buf * BCHUNK + lane;
                typedef typename AR::frag_t frag_t;
                if (!(a.ablate & 1))
#pragma unroll
                for (int s = 0; s < STEPS; ++s) {
                    const int tap = min(2 * s + (g >> 1), 8);        // the tenth tap slot multiplies zero weights
                    const int ky = tap / 3, kx = tap - 3 * ky;
                    const uint4* ap = a_base + (g & 1) * GPLANE + ky * c_xcols + kx;
                    frag_t bf[GNT][NP];
#pragma unroll
                    for (int nt = 0; nt < GNT; ++nt)
#pragma unroll
                        for (int p = 0; p < NP; ++p) bf[nt][p] = __builtin_bit_cast(frag_t, b_base[s * BSTEP + (nt * NP + p) * 64]);
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        frag_t af[4][NP];
#pragma unroll
                        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                            for (int p = 0; p < NP; ++p) af[mt][p] = __builtin_bit_cast(frag_t, ap[p * NOCT * GPLANE + c_off[4 * h + mt]]);
#pragma unroll
                        for (int nt = 0; nt < GNT; ++nt)
#pragma unroll
                            for (int t = 0; t < AR::NTERMS; ++t)
#pragma unroll
                                for (int mt = 0; mt < 4; ++mt)
                                    acc[4 * h + mt][nt] = AR::mfma(af[mt][AR::ta(t)], bf[nt][AR::tb(t)], acc[4 * h + mt][nt]);
                    }
                }
                if (chunk == 0 && tid == 0) next_unit = settle(pending);           // read by everybody at this unit's last chunk (n_chunks >= 2)
                if (last) {
                    const size_t plane = (size_t)cu.H * cu.W;
                    float* ybc = cu.y + (size_t)cu.b * cu.Cout * plane;
                    const float* abc = cu.addend ? cu.addend + (size_t)cu.b * cu.Cout * plane : nullptr;
                    const int co0 = cu.cot * GNT * 16;
                    const bool vec = (cu.W & 3) == 0;
#pragma unroll
                    for (int nt = 0; nt < GNT; ++nt) {
                        const float bv = cu.bias ? cu.bias[co0 + nt * 16 + n] : 0.f;
#pragma unroll
                        for (int mt = 0; mt < 8; ++mt) {
                            const int yy = cu.y0 + c_row[mt];
                            if (yy < cu.H && !(a.ablate & 8)) {
                                const size_t roff = (size_t)(co0 + nt * 16 + n) * plane + (size_t)yy * cu.W;
                                f32x4 v = acc[mt][nt] * cu.unscale;
                                v += bv;
                                cseg_store_row4(ybc + roff, abc ? abc + roff : nullptr, cu.x0 + c_col[mt] + 4 * g, cu.W, vec, v);
                            }
                        }
                    }
                    if (cu.stats) {
                        float4* st = cu.stats + (size_t)co0 * cu.n_seg;
                        const int tx64 = (cu.W + 63) / 64;
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            const f32x4(&ah)[4][GNT] = reinterpret_cast<const f32x4(&)[4][GNT]>(acc[4 * h]);
                            const int y0w = cu.y0 + ((2 * wave + h) << cu.geo);
                            if (cu.geo == 0) group_stats_emit<0>(ah, cu.bias, co0, cu.unscale, cu.b, y0w, cu.x0, cu.H, cu.W, tx64, g, n, st, cu.n_seg);
                            else if (cu.geo == 1) group_stats_emit<1>(ah, cu.bias, co0, cu.unscale, cu.b, y0w, cu.x0, cu.H, cu.W, tx64, g, n, st, cu.n_seg);
                            else group_stats_emit<2>(ah, cu.bias, co0, cu.unscale, cu.b, y0w, cu.x0, cu.H, cu.W, tx64, g, n, st, cu.n_seg);
                        }
                    }
                }
                if (!(a.ablate & 32)) __syncthreads();                    // (I)
                if (!more) break;
                if (last) {
                    if (su.geo != cu.geo) compute_geometry(su.geo);
                    cu = su;
                    chunk = 0;
                } else {
                    ++chunk;
                }
                buf ^= 1;
            }
        }
    }
    if (tid == 0) {
        int* done = a.sched + 8 * SCHED_STRIDE;
        if (cseg_counter_add(done, 1) == (int)gridDim.x - 1) {
            for (int i = 0; i < 8; ++i) cseg_counter_store(a.sched + i * SCHED_STRIDE, 0);
            cseg_counter_store(done, 0);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// TWELVE waves: the eight computing waves of the first 512-pixel form (two per SIMD: one's LDS reads and waits hide under the
// other's MFMAs -- measured: with ONE computing wave per SIMD the K-steps alone run at half the matrix rate, 120 us for the four
// branches at batch 8 against 71 us at the sustained fp16 rate) PLUS four staging waves (one per SIMD) that fetch, split and store
// the next chunk iteration's patch underneath them. Three waves per SIMD: 168 VGPRs each. Same queue, same arithmetic per output
// element. CSEG_GROUP_PC selects: 2 (default) this kernel, 1 the 4 + 4 form above, 0 the form without staging waves.
// ---------------------------------------------------------------------------------------------------------
template <class AR>
__global__ __launch_bounds__(768) void conv3x3_group_pc12_kernel(const GArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint4 smem_g12[];
    __shared__ int next_unit;
    __shared__ float scales[CSEG_GROUP_MAX][2];
    constexpr int NP = AR::NP;
    constexpr int A_CELLS = NP * NOCT * GPLANE;
    constexpr int BSTEP = GNT * NP * 64;
    constexpr int BCHUNK = STEPS * BSTEP;
    uint4* As = smem_g12;                           // [2][piece NP][octet 2][GPLANE]
    uint4* Bs = smem_g12 + 2 * A_CELLS;             // [2][BCHUNK]

    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const bool producer = wave >= 8;
    const int pt = tid - 512;                       // producer thread index
    const int g = lane >> 4, n = lane & 15;
    const int q = blockIdx.x & 7;
    int* head = a.sched + q * SCHED_STRIDE;

    // CYCLE ACCOUNT (measurement only: built with -DCSEG_GROUP_TIMERS, switched on by CSEG_GROUP_ABLATE bit 128): where wave 0 and
    // wave 4 (the two computing waves of one SIMD) and wave 8 (staging) of every block spend their shader cycles, summed over the blocks
    // into the spare counter line of `sched` in units of 64 cycles (tools/probes/group_probe prints them, profiles/r06_group_cycles.txt):
    // [0] blocks, wave 0: [1] until the first item is staged, [2] K-steps, [3] epilogue, [4] at the barrier, [7] draining loads / stores
    // before it; wave 8: [5] loads issued -> split and stored, [6] at the barrier; wave 4: [8..12] as [1..4], [7]
#ifdef CSEG_GROUP_TIMERS
    const bool timed = (a.ablate & 128) != 0;
#else
    constexpr bool timed = false;
#endif
    long long tmr[5] = {0, 0, 0, 0, 0}, tlast = timed ? (long long)__builtin_readcyclecounter() : 0;
    auto lap = [&](int k) {
        if (timed) {
            const long long now = (long long)__builtin_readcyclecounter();
            tmr[k] += now - tlast;
            tlast = now;
        }
    };
    int pending = 0;
    if (tid == 0) pending = cseg_counter_add(head, 1);
    {
        unsigned rec[2 * CSEG_GROUP_MAX];
#pragma unroll
        for (int i = 0; i < CSEG_GROUP_MAX; ++i) {          // (members beyond n_members are copies of member 0: valid pointers)
            rec[2 * i] = a.m[i].amax_x[(tid & (CSEG_AMAX_SLOTS - 1)) * CSEG_AMAX_STRIDE];
            rec[2 * i + 1] = a.m[i].amax_w[(tid & (CSEG_AMAX_SLOTS - 1)) * CSEG_AMAX_STRIDE];
        }
#pragma unroll
        for (int i = 0; i < CSEG_GROUP_MAX; ++i) {
            unsigned e[2];
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                unsigned v = rec[2 * i + k];
#pragma unroll
                for (int o = CSEG_AMAX_SLOTS / 2; o > 0; o >>= 1) v = max(v, (unsigned)__shfl_xor((int)v, o, 64));
                const unsigned ee = v >> 23 & 0xffu;
                e[k] = ee < 15u ? 15u : (ee > 253u ? 253u : ee);
            }
            if (tid == 0) {
                scales[i][0] = split_scale_of(e[0]);
                scales[i][1] = split_unscale_of(e[0]) * split_unscale_of(e[1]);
            }
        }
    }
    auto settle = [&](int u) -> int {
        for (;;) {
            if (u >= a.units_per_xcd) return -1;
            int mi = 0;
            for (int i = 1; i < a.n_members; ++i) mi = u >= a.m[i].unit0 ? i : mi;
            const int tl = (u - a.m[mi].unit0) / a.m[mi].n_cot;
            if (q * a.m[mi].per_xcd + tl < a.m[mi].n_spatial) return u;
            u = cseg_counter_add(head, 1);
        }
    };
    struct Unit {
        const float* x;
        const uint4* wbase;
        float* y;
        const float* bias;
        const float* addend;
        float4* stats;
        int Cin, Cout, H, W, n_seg, n_chunks, geo, cot, b, y0, x0;
        float xscale, unscale;
    };
    auto decode = [&](int u, Unit& U) {
        int mi = 0;
        for (int i = 1; i < a.n_members; ++i) mi = u >= a.m[i].unit0 ? i : mi;
        const GMember& M = a.m[mi];
        const int local = u - M.unit0;
        U.cot = local % M.n_cot;
        int t = q * M.per_xcd + local / M.n_cot;
        const int tx = t % M.tiles_x; t /= M.tiles_x;
        const int ty = t % M.tiles_y;
        U.b = t / M.tiles_y;
        U.geo = M.geo;
        U.x0 = tx * (64 >> M.geo); U.y0 = ty * (8 << M.geo);
        U.x = M.x; U.y = M.y; U.bias = M.bias; U.addend = M.addend; U.stats = M.stats;
        U.Cin = M.Cin; U.Cout = M.Cout; U.H = M.H; U.W = M.W; U.n_seg = M.n_seg; U.n_chunks = M.n_chunks;
        U.wbase = M.wp + (size_t)U.cot * M.n_chunks * BCHUNK;
        U.xscale = scales[mi][0]; U.unscale = scales[mi][1];
    };

    if (tid == 0) next_unit = settle(pending);
    __syncthreads();
    const int u_first = __builtin_amdgcn_readfirstlane(next_unit);
    if (u_first >= 0) {
        Unit su;
        decode(u_first, su);
        if (producer) {
            // ---------------- staging waves: the patch of item k + 1 while the computing waves are on item k
            float apre[PAU][8];
            int it_r[PAU], it_col[PAU], it_cell[PAU], it_oct8[PAU];
            bool it_in[PAU];
            int s_geo = -1;
            auto stage_geometry = [&](int geo) {
                const int xcols = (64 >> geo) + 2, cells = ((8 << geo) + 2) * xcols;
#pragma unroll
                for (int u = 0; u < PAU; ++u) {
                    const int item = pt + 256 * u;
                    const int oct = item >= cells ? 1 : 0, rc = min(item - oct * cells, cells - 1);
                    it_r[u] = rc / xcols;
                    it_col[u] = rc - it_r[u] * xcols;
                    it_in[u] = item < 2 * cells;
                    it_cell[u] = oct * GPLANE + rc;
                    it_oct8[u] = oct * 8;
                }
                s_geo = geo;
            };
            auto a_issue = [&](int chunk) {
                const int plane = su.H * su.W;
                const float* xc = su.x + ((size_t)su.b * su.Cin + (size_t)chunk * 16) * plane;
                const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)xc, 0, (int)(16 * plane * sizeof(float)), 0x00020000);
#pragma unroll
                for (int u = 0; u < PAU; ++u) {
                    const int yc = min(max(su.y0 + it_r[u] - 1, 0), su.H - 1), xcl = min(max(su.x0 + it_col[u] - 1, 0), su.W - 1);
                    const int off = (it_oct8[u] * plane + yc * su.W + xcl) * (int)sizeof(float);
#pragma unroll
                    for (int j = 0; j < 8; ++j)
                        apre[u][j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, off, j * plane * (int)sizeof(float), 0));
                }
            };
            auto a_store = [&](uint4* dst) {
#pragma unroll
                for (int u = 0; u < PAU; ++u) {
                    if (it_in[u]) {
                        const int yy = su.y0 + it_r[u] - 1, xx = su.x0 + it_col[u] - 1;
                        const bool ok = yy >= 0 && yy < su.H && xx >= 0 && xx < su.W;
                        uint4 cells[NP];
                        split_cells8_masked<AR>(apre[u], ok, su.xscale, cells);
#pragma unroll
                        for (int p = 0; p < NP; ++p) dst[p * NOCT * GPLANE + it_cell[u]] = cells[p];
                    }
                }
            };
            stage_geometry(su.geo);
            a_issue(0);
            a_store(As);
            __syncthreads();                        // (P) item 0 is staged
            lap(0);
            int chunk = 0, buf = 0, n_chunks = su.n_chunks;      // the item the computing waves are on
#pragma unroll 1
            for (;;) {
                const bool last = chunk == n_chunks - 1;
                bool more = true;
                int nchunk = chunk + 1;
                if (last) {
                    const int nu = __builtin_amdgcn_readfirstlane(next_unit);
                    more = nu >= 0;
                    nchunk = 0;
                    if (more) {
                        decode(nu, su);
                        if (su.geo != s_geo) stage_geometry(su.geo);
                    }
                }
                if (more) {
                    a_issue(nchunk);
                    a_store(As + (size_t)(buf ^ 1) * A_CELLS);
                }
                lap(1);
                __syncthreads();                    // (I)
                lap(2);
                if (!more) break;
                if (last) { chunk = 0; n_chunks = su.n_chunks; } else ++chunk;
                buf ^= 1;
            }
            if (timed && tid == 512) {
                atomicAdd(a.sched + 9 * SCHED_STRIDE + 5, (int)(tmr[1] >> 6));
                atomicAdd(a.sched + 9 * SCHED_STRIDE + 6, (int)(tmr[2] >> 6));
            }
        } else {
            // ---------------- computing waves: wave w = 64 pixels x all three channel tiles
            Unit cu = su;
            f32x4 acc[4][GNT];
            int c_row[4], c_col[4], c_off[4], c_xcols = 0;
            auto compute_geometry = [&](int geo) {
                c_xcols = (64 >> geo) + 2;
                const int per = 4 >> geo;
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) {
                    c_row[mt] = (wave << geo) + mt / per;
                    c_col[mt] = 16 * (mt % per);
                    c_off[mt] = c_row[mt] * c_xcols + c_col[mt] + n;
                }
            };
            compute_geometry(cu.geo);
            auto b_dma = [&](const uint4* wbase, int chunk, int slot) {      // 30 rows of 1 KB over the eight computing waves
                constexpr int ROWS = STEPS * GNT * NP;
                uint4* dst = Bs + (size_t)slot * BCHUNK;
#pragma unroll
                for (int i = 0; i < (ROWS + 7) / 8; ++i) {
                    const int r = wave + 8 * i;
                    if (r < ROWS)
                        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wbase + (size_t)chunk * BCHUNK + r * 64 + lane),
                                                         (__attribute__((address_space(3))) void*)(dst + r * 64), 16, 0, 0);
                }
            };
            b_dma(cu.wbase, 0, 0);
            __syncthreads();                        // (P)
            lap(0);
            int chunk = 0, buf = 0;
#pragma unroll 1
            for (;;) {
                if (chunk == 0) {
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                        for (int nt = 0; nt < GNT; ++nt) acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
                    if (tid == 0) pending = cseg_counter_add(head, 1);      // looked at after the K-steps: nobody waits for the round trip
                }
                const bool last = chunk == cu.n_chunks - 1;
                bool more = true;
                if (last) {
                    const int nu = __builtin_amdgcn_readfirstlane(next_unit);
                    more = nu >= 0;
                    if (more) decode(nu, su);
                }
                if (more) b_dma(last ? su.wbase : cu.wbase, last ? 0 : chunk + 1, buf ^ 1);
                const uint4* a_base = As + (size_t)buf * A_CELLS;
                const uint4* b_base = Bs + (size_t)buf * BCHUNK + lane;
                typedef typename AR::frag_t frag_t;
                // Tried on this loop and measured equal or worse (round 6, profiles/r06_group_cycles.txt; not
                // kept): fragments of group i + 1 requested before group i multiplies (two register sets, reads
                // pinned with sched_barrier: 132.8 / 135.6 us against 133.7 / 134.8); two items in flight in the staging waves (132.4);
                // s_setprio 1 / 2 on the staging waves (144 / 145 us: their VALU work then displaces MFMAs one for one). The cycle
                // account says why: wave 0 issues its 180 MFMAs per iteration in 5 740 cycles = the SIMD's matrix rate shared with wave
                // 4, and every other instruction on the SIMD (staging ~260 VALU, epilogue, address arithmetic) ADDS to that.
#pragma unroll
                for (int s = 0; s < STEPS; ++s) {
                    const int tap = min(2 * s + (g >> 1), 8);        // the tenth tap slot multiplies zero weights
                    const int ky = tap / 3, kx = tap - 3 * ky;
                    const uint4* ap = a_base + (g & 1) * GPLANE + ky * c_xcols + kx;
                    frag_t af[4][NP];
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                        for (int p = 0; p < NP; ++p) af[mt][p] = __builtin_bit_cast(frag_t, ap[p * NOCT * GPLANE + c_off[mt]]);
#pragma unroll
                    for (int nt = 0; nt < GNT; ++nt) {
                        frag_t bf[NP];
#pragma unroll
                        for (int p = 0; p < NP; ++p) bf[p] = __builtin_bit_cast(frag_t, b_base[s * BSTEP + (nt * NP + p) * 64]);
#pragma unroll
                        for (int t = 0; t < AR::NTERMS; ++t)
#pragma unroll
                            for (int mt = 0; mt < 4; ++mt) acc[mt][nt] = AR::mfma(af[mt][AR::ta(t)], bf[AR::tb(t)], acc[mt][nt]);
                    }
                }
                if (chunk == 0 && tid == 0) next_unit = settle(pending);           // read by everybody at this unit's last chunk (n_chunks >= 2)
                lap(1);
                if (last) {
                    const size_t plane = (size_t)cu.H * cu.W;
                    float* ybc = cu.y + (size_t)cu.b * cu.Cout * plane;
                    const float* abc = cu.addend ? cu.addend + (size_t)cu.b * cu.Cout * plane : nullptr;
                    const int co0 = cu.cot * GNT * 16;
                    const bool vec = (cu.W & 3) == 0;
#pragma unroll
                    for (int nt = 0; nt < GNT; ++nt) {
                        const float bv = cu.bias ? cu.bias[co0 + nt * 16 + n] : 0.f;
#pragma unroll
                        for (int mt = 0; mt < 4; ++mt) {
                            const int yy = cu.y0 + c_row[mt];
                            if (yy < cu.H) {
                                const size_t roff = (size_t)(co0 + nt * 16 + n) * plane + (size_t)yy * cu.W;
                                f32x4 v = acc[mt][nt] * cu.unscale;
                                v += bv;
                                cseg_store_row4(ybc + roff, abc ? abc + roff : nullptr, cu.x0 + c_col[mt] + 4 * g, cu.W, vec, v);
                            }
                        }
                    }
                    if (cu.stats) {
                        float4* st = cu.stats + (size_t)co0 * cu.n_seg;
                        const int tx64 = (cu.W + 63) / 64, y0w = cu.y0 + (wave << cu.geo);
                        if (cu.geo == 0) group_stats_emit<0>(acc, cu.bias, co0, cu.unscale, cu.b, y0w, cu.x0, cu.H, cu.W, tx64, g, n, st, cu.n_seg);
                        else if (cu.geo == 1) group_stats_emit<1>(acc, cu.bias, co0, cu.unscale, cu.b, y0w, cu.x0, cu.H, cu.W, tx64, g, n, st, cu.n_seg);
                        else group_stats_emit<2>(acc, cu.bias, co0, cu.unscale, cu.b, y0w, cu.x0, cu.H, cu.W, tx64, g, n, st, cu.n_seg);
                    }
                }
                lap(2);
                if (timed) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); lap(4); }
                __syncthreads();                    // (I)
                lap(3);
                if (!more) break;
                if (last) {
                    if (su.geo != cu.geo) compute_geometry(su.geo);
                    cu = su;
                    chunk = 0;
                } else {
                    ++chunk;
                }
                buf ^= 1;
            }
            if (timed && tid == 0) {
                atomicAdd(a.sched + 9 * SCHED_STRIDE, 1);
                for (int k = 0; k < 4; ++k) atomicAdd(a.sched + 9 * SCHED_STRIDE + 1 + k, (int)(tmr[k] >> 6));
                atomicAdd(a.sched + 9 * SCHED_STRIDE + 7, (int)(tmr[4] >> 6));
            }
            if (timed && tid == 256)
                for (int k = 0; k < 5; ++k) atomicAdd(a.sched + 9 * SCHED_STRIDE + 8 + k, (int)(tmr[k] >> 6));
        }
    }
    if (tid == 0) {
        int* done = a.sched + 8 * SCHED_STRIDE;
        if (cseg_counter_add(done, 1) == (int)gridDim.x - 1) {
            for (int i = 0; i < 8; ++i) cseg_counter_store(a.sched + i * SCHED_STRIDE, 0);
            cseg_counter_store(done, 0);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// The same launch on 4 x 64-pixel tiles (256 pixels per unit: two waves per image row split the three channel tiles, the tile body
// of conv3x3_sb16p_kernel), for SMALL groups: when the heaviest unit of the 512-pixel form would be longer than a CU's share of the
// whole group (per-GPU batches of 1 - 4 images: the 384-channel units are 24 chunk iterations, the group 12 per CU), units of half
// the size balance better -- measured on the MI355X, 4 branches of HRNet-W48: batch 4 108.6 vs 133.1 us, batch 1 78.9 vs 113.9,
// batch 8 195.6 vs 146.7 (profiles/r06_group_probe.jsonl). Weights of a member whose whole operator fits three chunk slots
// (Cin <= 48) stay RESIDENT in LDS while the block stays on that member; otherwise slots 1 / 2 are the stream's double buffer.
// ---------------------------------------------------------------------------------------------------------
namespace g4 {
using namespace cseg_sb16t;
constexpr int WSLOTS = 3;               // weight chunk slots in LDS: a resident operator of <= 3 chunks, or slots 1 / 2 as the stream's double buffer

template <class AR>
__global__ __launch_bounds__(512, 1) void conv3x3_group4_kernel(const GArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint4 smem_grp4[];
    __shared__ int next_unit;
    __shared__ float scales[CSEG_GROUP_MAX][2];
    constexpr int NP = AR::NP;
    constexpr int A_CELLS = NP * NOCT * PLANE;
    constexpr int BSTEP = GNT * NP * 64;            // uint4 per K-step
    constexpr int BCHUNK = STEPS * BSTEP;           // uint4 per 16-channel chunk
    constexpr int NT0 = (GNT + 1) / 2, NT1 = GNT - NT0;
    uint4* As = smem_grp4;                           // [2][piece NP][octet 2][PLANE]
    uint4* Bs = smem_grp4 + 2 * A_CELLS;             // [WSLOTS][BCHUNK]

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int row = wave & 3, half = wave >> 2;
    const int g = lane >> 4, n = lane & 15;
    const int q = blockIdx.x & 7;                   // block b runs on XCD b % 8 (observed on gfx950; only speed depends on it)
    int* head = a.sched + q * SCHED_STRIDE;

    // per-member scales of the f16x3 split (max|x|, max|w| records -> powers of two), once per block
    for (int i = 0; i < a.n_members; ++i) {
        const unsigned ex = split_amax_exp(a.m[i].amax_x), ew = split_amax_exp(a.m[i].amax_w);      // every thread (shuffles inside)
        if (tid == 0) {
            scales[i][0] = split_scale_of(ex);
            scales[i][1] = split_unscale_of(ex) * split_unscale_of(ew);
        }
    }

    // thread 0: the next unit of this XCD's queue that names a tile inside the image batch, or -1
    auto grab = [&]() -> int {
        for (;;) {
            const int u = cseg_counter_add(head, 1);
            if (u >= a.units_per_xcd) return -1;
            int mi = 0;
            for (int i = 1; i < a.n_members; ++i) mi = u >= a.m[i].unit0 ? i : mi;
            const int tl = (u - a.m[mi].unit0) / a.m[mi].n_cot;
            if (q * a.m[mi].per_xcd + tl < a.m[mi].n_spatial) return u;
        }
    };

    struct Unit {                                   // everything a unit needs; uniform over the block
        const float* x;
        const uint4* wbase;
        float* y;
        const float* bias;
        const float* addend;
        float4* stats;
        int Cin, Cout, H, W, tiles_x, n_seg, n_chunks, key, cot, b, y0, x0;
        float xscale, unscale;
    };
    auto decode = [&](int u, Unit& U) {
        int mi = 0;
        for (int i = 1; i < a.n_members; ++i) mi = u >= a.m[i].unit0 ? i : mi;
        const GMember& M = a.m[mi];
        const int local = u - M.unit0;
        U.cot = local % M.n_cot;
        int t = q * M.per_xcd + local / M.n_cot;
        const int tx = t % M.tiles_x; t /= M.tiles_x;
        const int ty = t % M.tiles_y;
        U.b = t / M.tiles_y;
        U.x0 = tx * TC; U.y0 = ty * TR;
        U.x = M.x; U.y = M.y; U.bias = M.bias; U.addend = M.addend; U.stats = M.stats;
        U.Cin = M.Cin; U.Cout = M.Cout; U.H = M.H; U.W = M.W; U.tiles_x = M.tiles_x; U.n_seg = M.n_seg; U.n_chunks = M.n_chunks;
        U.wbase = M.wp + (size_t)U.cot * M.n_chunks * BCHUNK;
        U.key = mi * 4096 + U.cot;
        U.xscale = scales[mi][0]; U.unscale = scales[mi][1];
    };

    if (tid == 0) next_unit = grab();
    __syncthreads();                                // (also publishes `scales`)
    int u_first = __builtin_amdgcn_readfirstlane(next_unit);

    Unit su;                                        // unit being staged (patch loads, weight DMA)
    // staging items of a thread: (octet, patch row, patch column) do not depend on the unit
    float apre[AU][8];
    int it_r[AU], it_col[AU], it_cell[AU], it_oct8[AU];
    bool it_in[AU];
#pragma unroll
    for (int u = 0; u < AU; ++u) {
        const int item = tid + 512 * u;
        const int oct = item / CELLS, rc = item - oct * CELLS;
        it_r[u] = rc / XCOLS;
        it_col[u] = rc - it_r[u] * XCOLS;
        it_in[u] = oct < NOCT;
        it_cell[u] = oct * PLANE + rc;
        it_oct8[u] = min(oct, NOCT - 1) * 8;
    }
    auto a_issue = [&](int chunk) {                 // fp32 patch of (su, chunk) into registers
        const int plane = su.H * su.W;
        const float* xc = su.x + ((size_t)su.b * su.Cin + (size_t)chunk * 16) * plane;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)xc, 0, (int)(16 * plane * sizeof(float)), 0x00020000);
#pragma unroll
        for (int u = 0; u < AU; ++u) {
            const int yc = min(max(su.y0 + it_r[u] - 1, 0), su.H - 1), xcl = min(max(su.x0 + it_col[u] - 1, 0), su.W - 1);
            const int off = (it_oct8[u] * plane + yc * su.W + xcl) * (int)sizeof(float);
#pragma unroll
            for (int j = 0; j < 8; ++j)
                apre[u][j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, off, j * plane * (int)sizeof(float), 0));
        }
    };
    auto a_store = [&](uint4* dst) {                // ... split and stored (same unit as the a_issue before it)
#pragma unroll
        for (int u = 0; u < AU; ++u) {
            if (it_in[u]) {
                const int yy = su.y0 + it_r[u] - 1, xx = su.x0 + it_col[u] - 1;
                const bool ok = yy >= 0 && yy < su.H && xx >= 0 && xx < su.W;
                uint4 cells[NP];
                split_cells8_masked<AR>(apre[u], ok, su.xscale, cells);         // zero padding / outside the tensor
#pragma unroll
                for (int p = 0; p < NP; ++p) dst[p * NOCT * PLANE + it_cell[u]] = cells[p];
            }
        }
    };
    auto b_dma = [&](const uint4* wbase, int chunk, int slot) {   // one 16-channel chunk of packed weights: STEPS * GNT * NP rows of 1 KB
        constexpr int ROWS = STEPS * GNT * NP;
        uint4* dst = Bs + (size_t)slot * BCHUNK;
#pragma unroll
        for (int i = 0; i < (ROWS + 7) / 8; ++i) {
            const int r = wave + 8 * i;
            if (r < ROWS)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wbase + (size_t)chunk * BCHUNK + r * 64 + lane),
                                                 (__attribute__((address_space(3))) void*)(dst + r * 64), 16, 0, 0);
        }
    };

    if (u_first >= 0) {
        decode(u_first, su);
        int w_key = -1;                             // (member, channel group) whose whole operator sits in slots 0 .. n_chunks - 1
        int ws_cur;                                 // weight slot the current item reads
        int late = -1;                              // chunk of a resident operator that could not be loaded while its slot was being read
        // ---- prologue: patch of the first item, its weights
        a_issue(0);
        if (su.n_chunks <= WSLOTS) {
            for (int c = 0; c < su.n_chunks; ++c) b_dma(su.wbase, c, c);
            w_key = su.key;
            ws_cur = 0;
        } else {
            b_dma(su.wbase, 0, 1);
            ws_cur = 1;
        }
        a_store(As);
        __syncthreads();

        Unit cu = su;                               // unit being computed
        f32x4 acc[4][NT0];
        const int a_lane_off = row * XCOLS + n;
        const int b_lane_off = (half ? NT0 * NP * 64 : 0) + lane;
        int chunk = 0, buf = 0;
#pragma unroll 1
        for (;;) {
            if (chunk == 0) {
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NT0; ++nt) acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (tid == 0) next_unit = grab();                    // read at this unit's last chunk: at least one barrier later (n_chunks >= 2)
            }
            if (late >= 0) {                        // the slot the previous unit's last item was reading: free since the barrier
                b_dma(cu.wbase, late, late);
                late = -1;
            }
            const bool last = chunk == cu.n_chunks - 1;
            bool more = true;
            int nchunk = chunk + 1;
            if (last) {
                const int nu = __builtin_amdgcn_readfirstlane(next_unit);
                more = nu >= 0;
                nchunk = 0;
                if (more) decode(nu, su);
            }
            int ws_next = ws_cur;
            if (more) {
                a_issue(nchunk);                                     // fp32 loads of the next item fly under the MFMAs below
                if (su.n_chunks <= WSLOTS) {                         // resident operator
                    if (nchunk == 0 && su.key != w_key) {
                        for (int c = 0; c < su.n_chunks; ++c) {
                            if (c != ws_cur) b_dma(su.wbase, c, c);
                            else late = c;                           // (never chunk 0: a streamed item reads slot 1 or 2, a resident one its last chunk)
                        }
                        w_key = su.key;
                    }
                    ws_next = nchunk;
                } else {                                             // streamed: slots 1 and 2 alternate
                    w_key = -1;
                    ws_next = ws_cur == 1 ? 2 : 1;
                    b_dma(su.wbase, nchunk, ws_next);
                }
            }
            const uint4* a_lane = As + (size_t)buf * A_CELLS + a_lane_off;
            const uint4* b_base = Bs + (size_t)ws_cur * BCHUNK + b_lane_off;
#pragma unroll
            for (int s = 0; s < STEPS; ++s) {
                // the ninth tap is paired with a tenth that does not exist: its packed weights are zero
                const int tap = min(2 * s + (g >> 1), 8);
                const int ky = tap / 3, kx = tap - 3 * ky;
                const int a_off = (g & 1) * PLANE + ky * XCOLS + kx;
                if (half == 0) sb16_kstep<AR, NT0, NT0>(a_lane + a_off, b_base + s * BSTEP, acc);
                else if (NT1 > 0) sb16_kstep<AR, NT1, NT0>(a_lane + a_off, b_base + s * BSTEP, acc);
            }
            if (last) {
                const int yy = cu.y0 + row;
                if (yy < cu.H) {
                    const size_t plane = (size_t)cu.H * cu.W;
                    float* ybc = cu.y + (size_t)cu.b * cu.Cout * plane;
                    const float* abc = cu.addend ? cu.addend + (size_t)cu.b * cu.Cout * plane : nullptr;
                    const int co0 = cu.cot * GNT * 16;
                    if (half == 0) sb16_store<NT0, NT0>(acc, ybc, cu.bias, abc, co0, plane, yy, cu.x0, cu.W, g, n, cu.unscale);
                    else if (NT1 > 0) sb16_store<NT1, NT0>(acc, ybc, cu.bias, abc, co0 + NT0 * 16, plane, yy, cu.x0, cu.W, g, n, cu.unscale);
                    if (cu.stats) {
                        const size_t seg = ((size_t)cu.b * cu.H + yy) * cu.tiles_x + cu.x0 / TC;
                        if (half == 0)
                            cseg_stats_emit<NT0, NT0>(acc, cu.bias, co0, cu.unscale, cu.x0, cu.W, g, n, cu.stats + (size_t)co0 * cu.n_seg + seg, cu.n_seg);
                        else if (NT1 > 0)
                            cseg_stats_emit<NT1, NT0>(acc, cu.bias, co0 + NT0 * 16, cu.unscale, cu.x0, cu.W, g, n,
                                                      cu.stats + (size_t)(co0 + NT0 * 16) * cu.n_seg + seg, cu.n_seg);
                    }
                }
            }
            if (more) a_store(As + (size_t)(buf ^ 1) * A_CELLS);       // the other patch buffer: last read in the previous item
            __syncthreads();
            if (!more) break;
            if (last) { cu = su; chunk = 0; } else ++chunk;
            buf ^= 1;
            ws_cur = ws_next;
        }
    }
    // ---- the last block to get here puts the counters back to zero (every block has made its last, failing, grab by then)
    if (tid == 0) {
        int* done = a.sched + 8 * SCHED_STRIDE;
        if (cseg_counter_add(done, 1) == (int)gridDim.x - 1) {
            for (int i = 0; i < 8; ++i) cseg_counter_store(a.sched + i * SCHED_STRIDE, 0);
            cseg_counter_store(done, 0);
        }
    }
}


constexpr size_t lds_bytes() { return sizeof(uint4) * (2 * 2 * NOCT * PLANE + WSLOTS * STEPS * GNT * 2 * 64); }      // 143 360
}  // namespace g4

constexpr size_t group_lds_bytes() { return sizeof(uint4) * (2 * 2 * NOCT * GPLANE + 2 * STEPS * GNT * 2 * 64); }      // 147 456

}  // namespace

// One launch for n <= CSEG_GROUP_MAX independent convolutions y_i = conv2d(x_i, w_i, bias_i, stride 1, padding 1) [+ addend_i]
// [+ statistics records of y_i] -- see include/cseg_hip.h. f16x3 only.
extern "C" int cseg_conv3x3_split_group_fwd(const cseg_conv_group_member* mem, int n, int arith, int* sched, cseg_stream_t stream_) {
    CSEG_REQUIRE(mem && sched && n >= 1 && n <= CSEG_GROUP_MAX, "conv3x3 group: needs 1 .. %d members and a scheduling record", CSEG_GROUP_MAX);
    CSEG_REQUIRE(arith == CSEG_ARITH_F16X3, "conv3x3 group: f16x3 arithmetic only (got %d)", arith);
    CSEG_REQUIRE((reinterpret_cast<uintptr_t>(sched) & 3) == 0, "conv3x3 group: misaligned scheduling record");
    // heavy first: members in descending order of the chunk iterations per unit (stable: equal members keep the caller's order)
    int order[CSEG_GROUP_MAX];
    for (int i = 0; i < n; ++i) order[i] = i;
    for (int i = 1; i < n; ++i)
        for (int j = i; j > 0 && mem[order[j]].Cin > mem[order[j - 1]].Cin; --j) { const int t = order[j]; order[j] = order[j - 1]; order[j - 1] = t; }
    // Tile form. 512-pixel tiles do twice the MFMAs per chunk iteration for 1.3x its loads; but a unit is then as long as
    // Cin / 16 of those iterations, and a group whose heaviest unit is longer than a CU's share of all the work (small per-GPU
    // batches) finishes when that unit does -- there the 256-pixel form (half-size units) wins. CSEG_GROUP_TILE = 4 | 8 forces one.
    long work8 = 0;
    int longest = 0;
    for (int i = 0; i < n; ++i) {
        const cseg_conv_group_member& s = mem[i];
        if (s.W <= 0 || s.H <= 0 || s.B <= 0 || s.Cin <= 0 || s.Cout <= 0) continue;          // (rejected below)
        const int geo = s.W > 32 ? 0 : (s.W > 16 ? 1 : 2);
        const long tiles = (long)s.B * ((s.H + (8 << geo) - 1) / (8 << geo)) * ((s.W + (64 >> geo) - 1) / (64 >> geo));
        work8 += tiles * (s.Cout / 48) * (s.Cin / 16);
        longest = s.Cin / 16 > longest ? s.Cin / 16 : longest;
    }
    bool big = work8 >= 256L * longest;
    if (const char* e = getenv("CSEG_GROUP_TILE")) big = atoi(e) == 8 ? true : (atoi(e) == 4 ? false : big);
    GArgs a;
    int unit0 = 0;
    for (int k = 0; k < n; ++k) {
        const cseg_conv_group_member& s = mem[order[k]];
        CSEG_REQUIRE(s.x && s.wp && s.y && s.amax_x && s.amax_w, "conv3x3 group: member %d has a null pointer", order[k]);
        CSEG_REQUIRE(s.B > 0 && s.H > 0 && s.W > 0 && s.Cin >= 32 && s.Cin % 16 == 0 && s.Cout > 0 && s.Cout % 48 == 0 &&
                         (long)s.H * s.W * 16 * 4 < 2147483647L,
                     "conv3x3 group: member %d: unsupported shape B=%d Cin=%d Cout=%d %dx%d (needs Cin %% 16, Cin >= 32, Cout %% 48)", order[k],
                     s.B, s.Cin, s.Cout, s.H, s.W);
        CSEG_REQUIRE((reinterpret_cast<uintptr_t>(s.wp) & 15) == 0 && (reinterpret_cast<uintptr_t>(s.y) & 15) == 0 &&
                         (reinterpret_cast<uintptr_t>(s.addend) & 15) == 0 && (reinterpret_cast<uintptr_t>(s.stats) & 15) == 0,
                     "conv3x3 group: member %d: packed weights / output / addend / statistics must be 16-byte aligned", order[k]);
        CSEG_REQUIRE(!(s.stats && s.addend), "conv3x3 group: member %d: the statistics epilogue takes no addend", order[k]);
        GMember& m = a.m[k];
        m.x = s.x; m.wp = (const uint4*)s.wp; m.bias = s.bias; m.addend = s.addend; m.y = s.y; m.stats = (float4*)s.stats;
        m.amax_x = s.amax_x; m.amax_w = s.amax_w;
        m.Cin = s.Cin; m.Cout = s.Cout; m.H = s.H; m.W = s.W;
        m.geo = big ? (s.W > 32 ? 0 : (s.W > 16 ? 1 : 2)) : 0;
        m.pad0 = m.pad1 = m.pad2 = 0;
        const int tc = big ? 64 >> m.geo : cseg_sb16t::TC, tr = big ? 8 << m.geo : cseg_sb16t::TR;
        m.tiles_x = (s.W + tc - 1) / tc; m.tiles_y = (s.H + tr - 1) / tr;
        const long n_spatial = (long)s.B * m.tiles_y * m.tiles_x;
        CSEG_REQUIRE(n_spatial < (1L << 28) && s.Cout / 48 < 4096, "conv3x3 group: member %d: too many tiles", order[k]);
        m.n_spatial = (int)n_spatial;
        m.n_seg = s.B * s.H * ((s.W + 63) / 64);
        m.n_cot = s.Cout / (GNT * 16);
        m.n_chunks = s.Cin / 16;
        m.per_xcd = (m.n_spatial + 7) / 8;
        m.unit0 = unit0;
        const long units = (long)m.per_xcd * m.n_cot;
        CSEG_REQUIRE(unit0 + units < 2147483647L, "conv3x3 group: too many units");
        unit0 += (int)units;
    }
    for (int k = n; k < CSEG_GROUP_MAX; ++k) a.m[k] = a.m[0];
    a.n_members = n;
    a.units_per_xcd = unit0;
    a.sched = sched;
    const char* abl = getenv("CSEG_GROUP_ABLATE");
    a.ablate = abl ? atoi(abl) : 0;
    a.pad = 0;
    const size_t lds = big ? group_lds_bytes() : g4::lds_bytes();
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void*)conv3x3_group_kernel<SplitF16x3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)group_lds_bytes()) != hipSuccess ||
            hipFuncSetAttribute((const void*)conv3x3_group_pc_kernel<SplitF16x3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)group_lds_bytes()) != hipSuccess ||
            hipFuncSetAttribute((const void*)conv3x3_group_pc12_kernel<SplitF16x3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)group_lds_bytes()) != hipSuccess ||
            hipFuncSetAttribute((const void*)g4::conv3x3_group4_kernel<SplitF16x3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)g4::lds_bytes()) != hipSuccess) {
            cseg_set_error("conv3x3 group: cannot raise dynamic LDS to %zu bytes", lds);
            return 0;
        }
        attr_set = true;
    }
    const int per_xcd_blocks = unit0 < 32 ? unit0 : 32;       // one block per CU: 32 CUs per XCD
    // CSEG_GROUP_PC=0: the 512-pixel form with all eight waves in the same phases (first version of round 6) instead of producer / consumer waves
    const char* pce = getenv("CSEG_GROUP_PC");
    int pc = pce ? atoi(pce) : 2;
    if (pc == 1)
        for (int i = 0; i < n; ++i) pc = mem[i].Cin >= 48 ? pc : 2;      // (the 4 + 4 form's producers run two chunk iterations ahead: units of >= 3 chunks)
    if (big && pc == 2) hipLaunchKernelGGL(conv3x3_group_pc12_kernel<SplitF16x3>, dim3((unsigned)(8 * per_xcd_blocks)), dim3(768), lds, (hipStream_t)stream_, a);
    else if (big && pc == 1) hipLaunchKernelGGL(conv3x3_group_pc_kernel<SplitF16x3>, dim3((unsigned)(8 * per_xcd_blocks)), dim3(512), lds, (hipStream_t)stream_, a);
    else if (big) hipLaunchKernelGGL(conv3x3_group_kernel<SplitF16x3>, dim3((unsigned)(8 * per_xcd_blocks)), dim3(512), lds, (hipStream_t)stream_, a);
    else hipLaunchKernelGGL(g4::conv3x3_group4_kernel<SplitF16x3>, dim3((unsigned)(8 * per_xcd_blocks)), dim3(512), lds, (hipStream_t)stream_, a);
    CSEG_CHECK_LAUNCH("conv3x3_group_kernel");
    return 1;
}
