// Weight gradient of the 1x1 convolution on the BF16 matrix cores with split operands (bf16x6, see conv3x3_sb.hip):
//   dW[co][ci] = sum_{b,p} dy[b][co][p] * x[b][ci][p]        (p = pixel of the H*W plane)
// A GEMM whose contraction index (pixels) is the contiguous dimension of both NCHW operands: fragments are 8 consecutive
// pixels of one channel, read with one ds_read_b128 from [piece][channel][pixel] bf16 images -- no transposition.
// Reference site: nn.Conv2d(720, 720, 1) / (720, 256, 1) of the projection head (lib/models/modules/projection.py:8-24);
// MIOpen's fp32 weight gradient for 720 -> 720 at 8x128x256 takes 5.7 ms = 48 TFLOP/s
// (profiles/r02_conv_layout_probe_nchw_vs_channels_last.jsonl).
//
// Block = 8 waves on a 144 (co) x 128 (ci) channel block: waves 0-3 compute (2 co halves of 5 + 4 tiles x 2 ci halves of
// 4 tiles: 20 accumulators, 120 MFMAs per 32-pixel stage), waves 4-7 stage: aligned float4 loads of the next 32 pixels of
// all 272 channels, split into three bf16 pieces, 8-byte LDS writes into the other buffer; one barrier per stage
// (producer / consumer roles in separate loops with the same barrier sequence, as in conv3x3_sb_wrw2_kernel).
// Split-K over (image, 32-pixel stage) units, contiguous range per split; partial [split][co][ci] summed in a fixed
// order by a second kernel: deterministic.
// Status: index-checked against a numpy lane model; first hardware run pending -> opt-in (kernels.CONV1X1_SB_WRW).
// Round 3: default for channel counts >= 256 (720 x 720: 2.98 vs 5.75 ms on MIOpen in the round-2 driver pass); written against
// the arithmetic traits of cseg_split.h (bf16x6 and f16x3).
#include "cseg_split.h"

namespace {

constexpr int CO_T = 144, CI_T = 128, STG = 32;      // channel block, pixels per stage
constexpr int PITCH = 48;                            // half-words per (piece, channel) row: 32 + 16 pad = 96 bytes (6 x 16: conflict-free
                                                     // under the real ds_read_b128 lane groups, see conv3x3_sb_wrw.hip; 80 bytes was not)
__host__ __device__ constexpr int dy_elems(int np) { return np * CO_T * PITCH; }      // one dy buffer
__host__ __device__ constexpr int x_elems(int np) { return np * CI_T * PITCH; }       // one x buffer
constexpr int CH_ALL = CO_T + CI_T;                  // 272 channel rows per stage

constexpr int LD_ITEMS = CH_ALL * 8;                 // float4 chunks per stage: 272 rows x 8 chunks of 4 pixels
constexpr int LD_U = (LD_ITEMS + 255) / 256;         // per loader thread (9)

// RAGGED (round 5): planes whose size is not a multiple of 32 pixels (DeepLab-R101-d8: 65 x 129 = 8 385; HRNet-OCR at 520 x 520:
// 130 x 130 = 16 900). The last stage of an image is partly outside the plane, and channel rows are no longer 16-byte aligned: the
// loaders fetch the four pixels of a chunk one by one from addresses clamped into the plane and zero what lies behind it; the
// consumers see full zero-padded stages and do not change.
template <class AR, bool RAGGED = false>
__global__ __launch_bounds__(512, 1) void conv1x1_sb_wrw_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                                int B, int Cin, int Cout, int plane_i, int n_split,
                                                                const unsigned* __restrict__ amax_x,
                                                                const unsigned* __restrict__ amax_dy,
                                                                float* __restrict__ partial) {
    extern __shared__ __attribute__((aligned(16))) unsigned short smem_q[];
    constexpr int NP = AR::NP;
    constexpr int DY_ELEMS = dy_elems(NP), X_ELEMS = x_elems(NP);
    typedef typename AR::frag_t frag_t;
    unsigned short* ds = smem_q;                       // [2][piece][co 144][PITCH]
    unsigned short* xs = smem_q + 2 * DY_ELEMS;        // [2][piece][ci 128][PITCH]
    const unsigned ex = AR::SCALED ? split_amax_exp(amax_x) : 141u, ed = AR::SCALED ? split_amax_exp(amax_dy) : 141u;
    const float xscale = split_scale_of(ex), dscale = split_scale_of(ed);      // 1 for the unscaled arithmetic
    // readfirstlane: the role split below must be a SCALAR branch (the wave index is uniform, which the compiler cannot see)
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const bool loader = wave >= 4;
    const int lt = tid - 256;
    const int g = lane >> 4, n = lane & 15;
    int blk = blockIdx.x;
    const int split = blk % n_split; blk /= n_split;
    const int n_cib = (Cin + CI_T - 1) / CI_T;
    const int cib = blk % n_cib;
    const int cob = blk / n_cib;
    const size_t plane = (size_t)plane_i;
    const int stages_per_img = RAGGED ? (plane_i + STG - 1) / STG : plane_i / STG;
    const long n_units = (long)B * stages_per_img;
    const long u_lo = n_units * split / n_split, u_hi = n_units * (split + 1) / n_split;     // this split's stages

    if (loader) {
        // item = (channel row r of the 272, chunk c of 8): rows 0..143 = dy channels cob*144 + r, rows 144..271 = x channels.
        // Round 3: what an item is (which tensor, channel, chunk, LDS offset) is fixed for the whole kernel, so it is computed once;
        // a stage then costs, per item, one load with a SCALAR base (tensor + image + pixel offset of the stage) and a constant
        // 32-bit lane offset, a select and the split. Whether an item belongs to dy or x is uniform per wave (1 152 dy items =
        // 4.5 x 256), which the compiler cannot see: it is derived from the scalar wave index. (Before: a 64-bit division and a
        // 64-bit address per item and stage, ~360 VALU instructions per stage against the consumers' 60 MFMAs.)
        unsigned off[LD_U];
        int lds_off[LD_U];
        int room[4] = {0, 0, 0, 0};                    // RAGGED: pixels from the start of the stage held by register set s to the end of the plane
        int pc[LD_U];                                  // first pixel of the item's chunk inside a stage
        bool ok[LD_U], is_dy[LD_U];
#pragma unroll
        for (int u = 0; u < LD_U; ++u) {
            const int item = lt + 256 * u, itc = min(item, LD_ITEMS - 1);
            const int r = itc >> 3, c = itc & 7;
            is_dy[u] = 256 * u + 64 * (wave - 4) < CO_T * 8;                     // scalar
            const int ch = is_dy[u] ? min(cob * CO_T + r, Cout - 1) : min(cib * CI_T + r - CO_T, Cin - 1);
            off[u] = (unsigned)(ch * plane_i + 4 * c) * (unsigned)sizeof(float);
            ok[u] = item < LD_ITEMS && (is_dy[u] ? cob * CO_T + r < Cout : cib * CI_T + r - CO_T < Cin);
            lds_off[u] = is_dy[u] ? r * PITCH + 4 * c : (r - CO_T) * PITCH + 4 * c;
            pc[u] = 4 * c;
        }
        int img = (int)(u_lo / stages_per_img), stage = (int)(u_lo % stages_per_img);       // of the next stage to LOAD
        long next_unit = u_lo;                                                                // its index; past u_hi - 1 the last stage is fetched again
        auto load = [&](float4 (&v)[LD_U], int& room_of_set) __attribute__((always_inline)) {
            const float* dyp = dy + ((size_t)img * Cout * plane + (size_t)stage * STG);     // uniform
            const float* xp = x + ((size_t)img * Cin * plane + (size_t)stage * STG);
#pragma unroll
            for (int u = 0; u < LD_U; ++u) {
                if (!RAGGED) {
                    v[u] = cseg_load_f4(is_dy[u] ? dyp : xp, off[u]);
                } else {
                    // off[u] = (channel * plane + 4 * chunk) * 4 bytes from the start of the stage: element k sits `k` floats further,
                    // unless that is behind the plane -- then the last pixel of the plane is read (finite) and zeroed in put()
                    const int left = plane_i - stage * STG - pc[u];          // pixels from this chunk to the end of the plane
                    const char* base = reinterpret_cast<const char*>(is_dy[u] ? dyp : xp) + off[u];
                    float e[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) e[k] = *reinterpret_cast<const float*>(base + (long)min(k, left - 1) * 4);
                    v[u] = make_float4(e[0], e[1], e[2], e[3]);
                }
            }
            room_of_set = plane_i - stage * STG;
            // (issued UNCONDITIONALLY -- a stage behind this split's range re-reads the last one -- so that the compiler counts the
            // outstanding loads exactly and the wait for one register set leaves the younger ones in flight)
            if (++next_unit < u_hi) {
                if (++stage == stages_per_img) { stage = 0; ++img; }
            }
        };
        auto put = [&](int buf, const float4 (&v)[LD_U], int room_of_set) __attribute__((always_inline)) {
#pragma unroll
            for (int u = 0; u < LD_U; ++u) {
                if (ok[u] || lt + 256 * u < LD_ITEMS) {                          // rows beyond the channel count are zero-filled
                    float4 t = ok[u] ? v[u] : make_float4(0.f, 0.f, 0.f, 0.f);
                    if (RAGGED) {
                        const int nv = room_of_set - pc[u];              // pixels of this chunk inside the plane (<= 0: none)
                        t.x = nv > 0 ? t.x : 0.f; t.y = nv > 1 ? t.y : 0.f;
                        t.z = nv > 2 ? t.z : 0.f; t.w = nv > 3 ? t.w : 0.f;
                    }
                    uint2 cells[NP];
                    split_cells4<AR>(t, is_dy[u] ? dscale : xscale, cells);
                    unsigned short* base = is_dy[u] ? ds + buf * DY_ELEMS + lds_off[u] : xs + buf * X_ELEMS + lds_off[u];
                    const int pstride = is_dy[u] ? CO_T * PITCH : CI_T * PITCH;
#pragma unroll
                    for (int q = 0; q < NP; ++q) *reinterpret_cast<uint2*>(base + q * pstride) = cells[q];
                }
            }
        };
        // FOUR stages in flight (round 6). A stage is 32 pixels of 272 channel rows = 35 KB and 0.45 us of MFMAs; with ONE stage in
        // flight (rounds 2-5) every stage waited out an HBM round trip of ~2 us -- the kernel ran at 0.23 of the split roof on the
        // 720 x 720 head and at 1.7 TB/s on the 64 <-> 256 bottleneck layers, whose weight gradient is a stream of 335 MB
        // (profiles/r06_bench_detail_default.json). Register set s & 3 holds stage s from its load until its put().
        constexpr int PF = 4;
        float4 v[PF][LD_U];
        if (u_lo < u_hi) {
            load(v[0], room[0]);
            put(0, v[0], room[0]);
#pragma unroll
            for (int s = 1; s < PF; ++s) load(v[s], room[s]);           // stages 1 .. 3
            load(v[0], room[0]);                                        // stage 4
        }
        __syncthreads();
        long unit = u_lo;
        // main part: four stages that all have a successor -- no branch between the loads, so s_waitcnt leaves three sets in flight
#pragma unroll 1
        for (; unit + PF < u_hi; unit += PF) {
#pragma unroll
            for (int j = 0; j < PF; ++j) {             // stage k = unit - u_lo + j: LDS buffer k & 1 = j & 1, register set k & 3 = j
                put((j + 1) & 1, v[(j + 1) & 3], room[(j + 1) & 3]);       // that buffer was last read in stage k - 1 (barrier since)
                load(v[(j + 1) & 3], room[(j + 1) & 3]);                   // stage k + 5
                __syncthreads();
            }
        }
        // the last one to four stages
#pragma unroll
        for (int j = 0; j < PF; ++j) {
            if (unit + j < u_hi) {
                if (unit + j + 1 < u_hi) put((j + 1) & 1, v[(j + 1) & 3], room[(j + 1) & 3]);
                __syncthreads();
            }
        }
    } else {
        const int mh = wave & 1, nh = wave >> 1;       // co half (tiles 0-4 / 5-8), ci half (tiles 0-3 / 4-7)
        const int cot0 = mh ? 5 : 0, cit0 = nh * 4;
        f32x4 acc[5][4];
#pragma unroll
        for (int a = 0; a < 5; ++a)
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[a][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
        __syncthreads();
        auto stage_mfmas = [&](int buf) __attribute__((always_inline)) {
            frag_t bf[4][NP];
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int p = 0; p < NP; ++p)
                    bf[c][p] = __builtin_bit_cast(frag_t, *reinterpret_cast<const uint4*>(
                        xs + buf * X_ELEMS + (p * CI_T + (cit0 + c) * 16 + n) * PITCH + 8 * g));
#pragma unroll
            for (int a = 0; a < 5; ++a) {
                if (a == 4 && mh) break;               // the upper co half has four tiles
                frag_t af[NP];
#pragma unroll
                for (int p = 0; p < NP; ++p)
                    af[p] = __builtin_bit_cast(frag_t, *reinterpret_cast<const uint4*>(
                        ds + buf * DY_ELEMS + (p * CO_T + (cot0 + a) * 16 + n) * PITCH + 8 * g));
#pragma unroll
                for (int t = 0; t < AR::NTERMS; ++t)
#pragma unroll
                    for (int c = 0; c < 4; ++c) acc[a][c] = AR::mfma(af[AR::ta(t)], bf[c][AR::tb(t)], acc[a][c]);
            }
        };
#pragma unroll 1
        for (long unit = u_lo; unit < u_hi; unit += 2) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {              // buffer index = stage parity = j: LDS offsets are immediates
                if (unit + j < u_hi) {
                    stage_mfmas(j);
                    __syncthreads();
                }
            }
        }
        // D[m = 4g + r][n]: co = cob*144 + (cot0 + a)*16 + 4g + r, ci = cib*128 + (cit0 + c)*16 + n
        const float unscale = split_unscale_of(ex) * split_unscale_of(ed);
#pragma unroll
        for (int a = 0; a < 5; ++a) {
            if (a == 4 && mh) break;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int ci = cib * CI_T + (cit0 + c) * 16 + n;
                const int co = cob * CO_T + (cot0 + a) * 16 + 4 * g;
                if (ci < Cin) {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (co + r < Cout) partial[((size_t)split * Cout + co + r) * Cin + ci] = acc[a][c][r] * unscale;
                }
            }
        }
    }
}

// dW[co][ci] = sum over splits of partial[split][co][ci], fixed order. A block = 64 elements x four waves: wave w adds the splits w, w + 4,
// ... in four independent chains, the four waves' sums meet in LDS in a fixed order (the form of sb_wrw_reduce_body in conv3x3_sb_wrw.hip).
// Round 6: with up to 256 splits the one-thread-per-element loop of rounds 2-5 (two chains of 128 dependent adds over loads 18 KB apart,
// 18 blocks for a 96 x 48 gradient) took longer than the gradient kernel itself.
__global__ __launch_bounds__(256) void sb_wrw1_reduce_kernel(const float* __restrict__ partial, int n_split, int total,
                                                             float* __restrict__ dw) {
    __shared__ float red[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int e = blockIdx.x * 64 + lane;
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
    if (wave == 0 && e < total) dw[e] = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
}

// Pixel splits: one block per CU at a time (104 KB of LDS), so the launch runs in ROUNDS of 256 blocks; the count that minimises
// rounds x stages per block (+ the partials written and re-read per split) is taken. 720 x 720 at 8 x 128 x 256: 30 channel blocks;
// the old rule (768 / 30 = 26 splits = 780 blocks) paid a fourth round for 12 blocks.
int wrw1_splits(int B, int Cin, int Cout, int plane) {
    const long units = (long)B * ((plane + STG - 1) / STG);
    const int pairs = ((Cin + CI_T - 1) / CI_T) * ((Cout + CO_T - 1) / CO_T);
    const double stage_us = 0.75, split_us = 2.0 * Cin * Cout * sizeof(float) / 4.0e6;
    // up to 256 splits (round 6; 64 before): with ONE or two channel-block pairs -- the 64 <-> 256 bottleneck layers, the 1x1 convolutions of
    // the exchange units -- 64 splits were 64 / 128 blocks on 256 CUs for a gradient that is a stream of up to 335 MB
    const long n_max = units < 256 ? units : 256;
    auto cost_of = [&](long n) {
        const long rounds = (n * pairs + 255) / 256;
        return rounds * ((double)((units + n - 1) / n) * stage_us + 5.0) + n * split_us;
    };
    double best = 1e30;
    for (long n = 1; n <= n_max; ++n) best = cost_of(n) < best ? cost_of(n) : best;
    for (long n = 1; n <= n_max; ++n)
        if (cost_of(n) <= 1.03 * best) return (int)n;
    return (int)n_max;
}

}  // namespace

extern "C" size_t cseg_conv1x1_sb_wrw_ws_floats(int B, int Cin, int Cout, int HW) {
    if (B <= 0 || Cin <= 0 || Cout <= 0 || HW <= 0 || Cin % 16 || Cout % 16) return 0;
    return (size_t)wrw1_splits(B, Cin, Cout, HW) * Cin * Cout;
}

namespace {
template <class AR, bool RAGGED>
int launch_wrw1(const float* x, const float* dy, int B, int Cin, int Cout, int HW, int n_split, long blocks, const unsigned* amax_x,
                const unsigned* amax_dy, float* ws, hipStream_t stream) {
    const size_t lds = sizeof(unsigned short) * 2 * (dy_elems(AR::NP) + x_elems(AR::NP));
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void*)conv1x1_sb_wrw_kernel<AR, RAGGED>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) !=
            hipSuccess) {
            cseg_set_error("conv1x1_sb_wrw: cannot raise dynamic LDS to %zu bytes", lds);
            return 0;
        }
        attr_set = true;
    }
    hipLaunchKernelGGL((conv1x1_sb_wrw_kernel<AR, RAGGED>), dim3((unsigned)blocks), dim3(512), lds, stream, x, dy, B, Cin, Cout, HW, n_split,
                       amax_x, amax_dy, ws);
    CSEG_CHECK_LAUNCH("conv1x1_sb_wrw_kernel");
    return 1;
}

int wrw1_impl(const float* x, const float* dy, int B, int Cin, int Cout, int HW, int arith, const unsigned* amax_x,
              const unsigned* amax_dy, float* ws, float* dw, hipStream_t stream) {
    CSEG_REQUIRE(x && dy && ws && dw, "conv1x1_sb_wrw: null pointer");
    CSEG_REQUIRE(B > 0 && HW > 0 && Cin > 0 && Cout > 0 && Cin % 16 == 0 && Cout % 16 == 0,
                 "conv1x1_sb_wrw: unsupported shape B=%d Cin=%d Cout=%d HW=%d (needs channels %% 16)", B, Cin, Cout, HW);
    const bool ragged = HW % STG != 0;                 // (the aligned form reads 16 bytes at a time from 16-byte aligned channel rows)
    CSEG_REQUIRE(ragged || ((reinterpret_cast<uintptr_t>(x) & 15) == 0 && (reinterpret_cast<uintptr_t>(dy) & 15) == 0),
                 "conv1x1_sb_wrw: tensors must be 16-byte aligned");
    CSEG_REQUIRE(arith == CSEG_ARITH_BF16X6 || (arith == CSEG_ARITH_F16X3 && amax_x && amax_dy),
                 "conv1x1 split wrw: arithmetic %d needs max|x| and max|dy|", arith);
    CSEG_REQUIRE((long)Cin * HW * 4 < 2147483647L && (long)Cout * HW * 4 < 2147483647L,
                 "conv1x1_sb_wrw: one image of x / dy must stay below 2 GiB (32-bit offsets)");
    const int n_split = wrw1_splits(B, Cin, Cout, HW);
    const long blocks = (long)n_split * ((Cin + CI_T - 1) / CI_T) * ((Cout + CO_T - 1) / CO_T);
    CSEG_REQUIRE(blocks < 2147483647L && (long)Cin * Cout < 2147483647L, "conv1x1_sb_wrw: grid too large");
    int ok;
    if (arith == CSEG_ARITH_F16X3)
        ok = ragged ? launch_wrw1<SplitF16x3, true>(x, dy, B, Cin, Cout, HW, n_split, blocks, amax_x, amax_dy, ws, stream)
                    : launch_wrw1<SplitF16x3, false>(x, dy, B, Cin, Cout, HW, n_split, blocks, amax_x, amax_dy, ws, stream);
    else
        ok = ragged ? launch_wrw1<SplitBF16x6, true>(x, dy, B, Cin, Cout, HW, n_split, blocks, amax_x, amax_dy, ws, stream)
                    : launch_wrw1<SplitBF16x6, false>(x, dy, B, Cin, Cout, HW, n_split, blocks, amax_x, amax_dy, ws, stream);
    if (!ok) return 0;
    const int total = Cin * Cout;
    hipLaunchKernelGGL(sb_wrw1_reduce_kernel, dim3((total + 63) / 64), dim3(256), 0, stream, ws, n_split, total, dw);
    CSEG_CHECK_LAUNCH("sb_wrw1_reduce_kernel");
    return 1;
}
}  // namespace

extern "C" int cseg_conv1x1_sb_wrw(const float* x, const float* dy, int B, int Cin, int Cout, int HW, float* ws, float* dw,
                                   cseg_stream_t stream_) {
    return wrw1_impl(x, dy, B, Cin, Cout, HW, CSEG_ARITH_BF16X6, nullptr, nullptr, ws, dw, (hipStream_t)stream_);
}

extern "C" int cseg_conv1x1_split_wrw(const float* x, const float* dy, int B, int Cin, int Cout, int HW, int arith,
                                      const unsigned* amax_x, const unsigned* amax_dy, float* ws, float* dw, cseg_stream_t stream_) {
    return wrw1_impl(x, dy, B, Cin, Cout, HW, arith, amax_x, amax_dy, ws, dw, (hipStream_t)stream_);
}
