// Contrastive term of PixelContrastLoss on gfx950.
//
// Reference: lib/loss/loss_contrast.py:91-128 (self), lib/loss/loss_contrast_mem.py:91-152 (memory bank).
// Restated for parity in oracle/cseg_oracle.py:_contrast_core.
//
// Kernels
//   s_gemm_kernel    S = A . C^T / tau           fp32 MFMA (v_mfma_f32_32x32x2_f32), 64x64 tile per block, operand
//                                                tiles streamed with coalesced 16-B loads and staged in LDS
//   row_pass_kernel  per anchor row: max, sum of negatives, positive count, mean log-prob of positives
//   mean_kernel      loss = mean_i row_loss[i]  (fixed-order reduction: results are run-to-run deterministic)
//   bwd_kernel       dA = dloss/tau . H . C,  H = G (+ G^T in self mode) rebuilt on the fly from S and the row
//                    statistics directly in the MFMA A-operand register layout (no LDS round trip), C rows staged
//                    through wave-private LDS with coalesced 16-byte loads as the B operand; partial sums over
//                    column splits go to d_anchor_parts.
//
// MFMA 32x32x2 f32 layouts (MI355X guide section 3): A: lane l holds A[i=l&31][k=l>>5]; B: B[k=l>>5][j=l&31];
// C/D: col = l&31, row = (r&3) + 8*(r>>2) + 4*(l>>5) for accumulator register r in [0,16).
// The reduction index order is free as long as A and B agree, so lane half h=l>>5 walks the contiguous
// feature range [h*D/2, (h+1)*D/2): every operand row is read with 16-byte loads.
#include "cseg_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

struct ColSrc {
    int mode;  // 0 self, 1 plain, 2 bank
    int M, D;
    const float* rows;      // self: anchors, plain: contrast
    const int32_t* labs;    // self: a_lab, plain: c_lab
    const float* segq;
    const float* pixq;
    int ms;                 // bank slots per queue
    int packed;             // (K-1) * 2 * ms : columns that carry real bank rows

    // row pointer of contrast column j, nullptr for zero rows (bank tail) and out-of-range columns
    __device__ __forceinline__ const float* row(int j) const {
        if (j >= M) return nullptr;
        if (mode != 2) return rows + (size_t)j * D;
        if (j >= packed) return nullptr;
        const int two = 2 * ms;
        const int c = 1 + j / two;
        const int s = j - (c - 1) * two;
        return s < ms ? segq + ((size_t)c * ms + s) * D : pixq + ((size_t)c * ms + (s - ms)) * D;
    }
    __device__ __forceinline__ int label(int j) const {
        if (mode != 2) return j < M ? labs[j] : -0x7fffffff;
        return j < packed ? 1 + j / (2 * ms) : 0;
    }
    // row pointers and labels of 4 consecutive columns j..j+3 with ONE integer division in bank mode
    __device__ __forceinline__ void decode4(int j, const float* (&rp)[4], int (&lb)[4]) const {
        if (mode != 2) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const bool ok = j + t < M;
                rp[t] = ok ? rows + (size_t)(j + t) * D : nullptr;
                lb[t] = ok ? labs[j + t] : -0x7fffffff;
            }
            return;
        }
        const int two = 2 * ms;
        int c = j / two;              // class index - 1 of column j
        int sl = j - c * two;         // slot inside the class block (segment queue first, then pixel queue)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            if (j + t < packed) {
                const int cc = c + 1;
                rp[t] = sl < ms ? segq + ((size_t)cc * ms + sl) * D : pixq + ((size_t)cc * ms + (sl - ms)) * D;
                lb[t] = cc;
            } else {
                rp[t] = nullptr;      // zero tail of the bank (label 0) or beyond M
                lb[t] = j + t < M ? 0 : -0x7fffffff;
            }
            if (++sl == two) { sl = 0; ++c; }
        }
    }
};

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }

// ---------------------------------------------------------------------------------------------------------
// S = A . C^T / tau.  Block = 4 waves = one 64 x 64 tile of S (2 x 2 waves of 32 x 32). The feature axis is walked
// in chunks of 64 floats: both operand tiles are streamed from HBM/L2 with fully coalesced 16-byte loads (16 lanes
// per 256-byte row segment), staged in LDS with a 68-float row stride (conflict-free ds_read_b128 for the
// row-per-lane fragment reads), and the next chunk is prefetched into registers while the MFMAs of the current one
// run. Inside a chunk lane half h = lane >> 5 covers features [32h, 32h+32): the reduction order is free as long as
// A and B agree. ldS = M rounded up to 32.
// ---------------------------------------------------------------------------------------------------------
constexpr int SG_T = 64;      // tile edge (rows and columns)
constexpr int SG_KC = 64;     // features per chunk
constexpr int SG_LD = 68;     // LDS row stride in floats

// this wave's 32 x 32 part (wi = wave >> 1, wj = wave & 1) of the 64 x 64 tile (I0, J0) of A . C^T, unscaled; As / Cs: the block's
// staging arrays. Starts with a block barrier (the previous tile's fragment reads), so calls can follow each other directly.
__device__ __forceinline__ f32x16 s_tile_64(const float* __restrict__ A, int N, const ColSrc& col, int I0, int J0, float* __restrict__ As,
                                            float* __restrict__ Cs) {
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int wi = wave >> 1, wj = wave & 1;
    const int h = lane >> 5, r32 = lane & 31;
    const int D = col.D;
    // staging assignment: float4 index f = tid + 256*u -> row f>>4, 16-byte column f&15 (16 lanes per row segment)
    const float* arow[4];
    const float* crow[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int r = (tid + 256 * u) >> 4;
        arow[u] = (I0 + r) < N ? A + (size_t)(I0 + r) * D : nullptr;
        crow[u] = col.row(J0 + r);
    }
    const int c4 = (tid & 15) * 4;
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 pa[4], pc[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const bool ok = c4 < D;
        pa[u] = (ok && arow[u]) ? ld4(arow[u] + c4) : z4;
        pc[u] = (ok && crow[u]) ? ld4(crow[u] + c4) : z4;
    }
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;

    for (int kc = 0; kc < D; kc += SG_KC) {
        __syncthreads();                                   // previous chunk's fragment reads are done
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int r = (tid + 256 * u) >> 4;
            *reinterpret_cast<float4*>(As + r * SG_LD + c4) = pa[u];
            *reinterpret_cast<float4*>(Cs + r * SG_LD + c4) = pc[u];
        }
        __syncthreads();
        const int kn = kc + SG_KC;
        if (kn < D) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const bool ok = kn + c4 < D;
                pa[u] = (ok && arow[u]) ? ld4(arow[u] + kn + c4) : z4;
                pc[u] = (ok && crow[u]) ? ld4(crow[u] + kn + c4) : z4;
            }
        }
        const float* ap = As + (wi * 32 + r32) * SG_LD + h * 32;
        const float* cp = Cs + (wj * 32 + r32) * SG_LD + h * 32;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const float4 a4 = *reinterpret_cast<const float4*>(ap + 4 * q);
            const float4 b4 = *reinterpret_cast<const float4*>(cp + 4 * q);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, b4.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, b4.y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, b4.z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, b4.w, acc, 0, 0, 0);
        }
    }
    return acc;
}

__global__ __launch_bounds__(256) void s_gemm_kernel(const float* __restrict__ A, int N, ColSrc col, float inv_tau,
                                                     float* __restrict__ S, int ldS, int nJb) {
    __shared__ __attribute__((aligned(16))) float As[SG_T * SG_LD];
    __shared__ __attribute__((aligned(16))) float Cs[SG_T * SG_LD];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int I0 = (blockIdx.x / nJb) * SG_T, J0 = (blockIdx.x % nJb) * SG_T;
    const int wi = wave >> 1, wj = wave & 1;
    const int h = lane >> 5, r32 = lane & 31;
    const f32x16 acc = s_tile_64(A, N, col, I0, J0, As, Cs);
    // C layout: this lane holds column j, rows (r&3) + 8*(r>>2) + 4*h of the wave's 32 x 32 tile
    const int j = J0 + wj * 32 + r32;
    if (j < ldS) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ii = I0 + wi * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (ii < N) S[(size_t)ii * ldS + j] = acc[r] * inv_tau;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// Row statistics + per-row loss. One block (256 threads) per anchor row.
// row_stats[i] = { m_i, Neg_i, w_i = coef / (N * P_i), R_i = sum_pos 1 / (e^{L_ij} + Neg_i) }
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float block_sum(float v, float* red, int tid) {
    v = wave_sum(v);
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}
__device__ __forceinline__ float block_max(float v, float* red, int tid) {
    v = wave_max(v);
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    return fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

__global__ __launch_bounds__(256) void row_pass_kernel(const float* __restrict__ S, int ldS, int N, ColSrc col,
                                                       const int32_t* __restrict__ a_lab, float coef,
                                                       float* __restrict__ row_stats, float* __restrict__ row_loss) {
    __shared__ float red[4];
    const int i = blockIdx.x, tid = threadIdx.x;
    const int M = col.M;
    const float* row = S + (size_t)i * ldS;
    const int yi = a_lab[i];
    float m = -INFINITY;
    for (int j = tid; j < M; j += 256) {
        const float s = row[j];
        m = (s > m || s != s) ? s : m;  // NaN propagates like torch.max
    }
    m = block_max(m, red, tid);
    float neg = 0.f, cnt = 0.f;
    for (int j = tid; j < M; j += 256) {
        const int yj = col.label(j);
        if (yj != yi) neg += expf(row[j] - m);
        else if (j != i) cnt += 1.f;
    }
    neg = block_sum(neg, red, tid);
    cnt = block_sum(cnt, red, tid);
    float slp = 0.f, rs = 0.f;
    for (int j = tid; j < M; j += 256) {
        if (j != i && col.label(j) == yi) {
            const float L = row[j] - m;
            const float den = expf(L) + neg;
            slp += L - logf(den);
            rs += 1.f / den;
        }
    }
    slp = block_sum(slp, red, tid);
    rs = block_sum(rs, red, tid);
    if (tid == 0) {
        row_loss[i] = -coef * (slp / cnt);             // 0/0 -> NaN exactly like the reference (no positives)
        float4 st = make_float4(m, neg, coef / ((float)N * cnt), rs);
        *reinterpret_cast<float4*>(row_stats + 4 * (size_t)i) = st;
    }
}

__global__ __launch_bounds__(256) void mean_kernel(const float* __restrict__ row_loss, int N, float* __restrict__ loss) {
    __shared__ float red[4];
    float v = 0.f;
    for (int i = threadIdx.x; i < N; i += 256) v += row_loss[i];
    v = block_sum(v, red, threadIdx.x);
    if (threadIdx.x == 0) loss[0] = v / (float)N;
}

// ---------------------------------------------------------------------------------------------------------
// Round 5: the forward as ONE launch (north_star: "the contrastive loss as a fused CDNA4 kernel").
// The three kernels above write S [N x M] to HBM, read it back three times in row_pass_kernel and once more in the backward. Here
// a block owns a STRIP of 64 anchor rows and every nsplit-th 64-column tile of it:
//   phase 1  S tiles on the fp32 MFMA (s_tile_64); every lane keeps an ONLINE (max, sum of negatives, positive count) for its
//            16 rows over the columns it sees -- the rescaling form of the log-sum-exp, no second pass over a row; the first
//            `t_cache` tiles of the block stay in LDS in the accumulator layout. Wavefront reductions (xor butterflies inside the
//            32-lane halves) + one LDS step give the block's partial per row -> part1[split][row].
//   hand-off per strip: release + counter; every block of the strip waits for the strip's nsplit partials (cseg_wait_counter: the
//            grid is sized to be resident, MI355X_MICROARCH.md "Residency") and combines them in split order.
//            Why a wait and not "the last block does the rest": log(exp(L_ij) + Neg_i) of every POSITIVE needs the complete
//            negative sum of its row (loss_contrast.py:119-126), i.e. a second sweep over the positives of the whole strip --
//            by one block that is 32 x fewer CUs than the problem has (110 us); with the wait every
//            block sweeps its own tiles again, from LDS (or recomputed on the matrix cores beyond t_cache tiles).
//   phase 2  sum over positives of L - log(e^L + Neg) and of 1 / (e^L + Neg) -> part2[split][row]; the LAST block of a strip to
//            arrive combines them in split order and writes row_stats / row_loss; the last strip's finisher takes the mean in
//            mean_kernel's order and resets the counters for the next launch.
// Deterministic: every sum has a fixed order (tile order per lane, butterfly, wave pair, split index), whichever block arrives
// last. Not bit-identical to the three-kernel path (online rescaling instead of max-then-sum): equal to ~1e-7 relative.
// S_out (optional): the tiles are also stored once, for the backward that reads S (cseg_contrast_bwd); nullptr = no N x M array
// exists at all (backward: cseg_contrast_bwd_recompute).
// ---------------------------------------------------------------------------------------------------------
constexpr int FU_T = 64;                    // strip height and column tile width
constexpr int FU_TILE_FLOATS = 4 * 16 * 64; // one tile in the accumulator layout: [wave][register][lane]

__device__ __forceinline__ float max_nan(float a, float b) { return (b > a || b != b) ? b : a; }     // NaN wins, like torch.max
// (m1, n1) + (m2, n2): sums of exponentials relative to their maxima -> relative to the common maximum
__device__ __forceinline__ void lse_merge(float& m, float& n, float m2, float n2) {
    const float mm = max_nan(m, m2);
    const float e1 = (m == mm) ? 1.f : expf(m - mm), e2 = (m2 == mm) ? 1.f : expf(m2 - mm);      // (-inf, 0) merges as (x, 0 * 1)
    n = n * e1 + n2 * e2;
    m = mm;
}

__global__ __launch_bounds__(256) void contrast_fused_fwd_kernel(const float* __restrict__ A, int N, ColSrc col,
                                                                 const int32_t* __restrict__ a_lab, float inv_tau, float coef,
                                                                 int nsplit, int n_strips, int t_cache, float* __restrict__ part1,
                                                                 float* __restrict__ part2, int* __restrict__ counters,
                                                                 float* __restrict__ S_out, int ldS, float* __restrict__ row_stats,
                                                                 float* __restrict__ row_loss, float* __restrict__ loss) {
    __shared__ __attribute__((aligned(16))) float As[SG_T * SG_LD];
    __shared__ __attribute__((aligned(16))) float Cs[SG_T * SG_LD];
    __shared__ float p1[2][FU_T][4];        // per column half (wj): m, neg, cnt | slp, rs
    __shared__ float fin[FU_T][4];          // per row of the strip: m, Neg, cnt
    __shared__ float red[4];
    __shared__ int flag;
    extern __shared__ __attribute__((aligned(16))) float smem_fu[];          // [t_cache][FU_TILE_FLOATS]
    float* stash = smem_fu;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int wi = wave >> 1, wj = wave & 1, h = lane >> 5, r32 = lane & 31;
    const int strip = blockIdx.x / nsplit, split = blockIdx.x - strip * nsplit;
    const int I0 = strip * FU_T, M = col.M;
    const int nJ = (M + FU_T - 1) / FU_T;
    const int Npad = n_strips * FU_T;
    int* cnt1 = counters;
    int* cnt2 = counters + n_strips;
    int* done = counters + 2 * n_strips;

    float m[16], ng[16], ct[16];
    int yi[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int ii = I0 + wi * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        yi[r] = ii < N ? a_lab[ii] : -0x7ffffffe;
        m[r] = -INFINITY; ng[r] = 0.f; ct[r] = 0.f;
    }
    // ---- phase 1
    int t = 0;
    for (int jt = split; jt < nJ; jt += nsplit, ++t) {
        const int J0 = jt * FU_T;
        f32x16 acc = s_tile_64(A, N, col, I0, J0, As, Cs);
        const int j = J0 + wj * 32 + r32;
        const bool j_ok = j < M;
        const int yj = col.label(j);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float sv = acc[r] * inv_tau;
            acc[r] = sv;
            if (j_ok) {
                const int ii = I0 + wi * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                const float mm = max_nan(m[r], sv);
                const float e_old = (m[r] == mm) ? 1.f : expf(m[r] - mm);
                const bool same = yj == yi[r];
                ng[r] = ng[r] * e_old + (same ? 0.f : expf(sv - mm));
                m[r] = mm;
                ct[r] += (same && j != ii) ? 1.f : 0.f;
            }
        }
        if (t < t_cache) {
            float* st = stash + (size_t)t * FU_TILE_FLOATS + wave * (16 * 64) + lane;
#pragma unroll
            for (int r = 0; r < 16; ++r) st[r * 64] = acc[r];
        }
        if (S_out != nullptr && j < ldS) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ii = I0 + wi * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (ii < N) S_out[(size_t)ii * ldS + j] = acc[r];
            }
        }
    }
    // the 32 lanes of a half hold different columns of the same 16 rows
#pragma unroll
    for (int r = 0; r < 16; ++r) {
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) {
            const float m2 = __shfl_xor(m[r], o, 64), n2 = __shfl_xor(ng[r], o, 64), c2 = __shfl_xor(ct[r], o, 64);
            lse_merge(m[r], ng[r], m2, n2);
            ct[r] += c2;
        }
    }
    if (r32 == 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = wi * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            p1[wj][row][0] = m[r]; p1[wj][row][1] = ng[r]; p1[wj][row][2] = ct[r];
        }
    }
    __syncthreads();
    if (tid < FU_T) {
        float mm = p1[0][tid][0], nn = p1[0][tid][1];
        lse_merge(mm, nn, p1[1][tid][0], p1[1][tid][1]);
        float* o = part1 + ((size_t)split * Npad + I0 + tid) * 4;
        o[0] = mm; o[1] = nn; o[2] = p1[0][tid][2] + p1[1][tid][2];
    }
    // ---- the strip's partials: publish, wait for all of them, combine in split order
    __syncthreads();
    if (tid == 0) {
        cseg_release_agent();
        cseg_counter_add(cnt1 + strip, 1);
        cseg_wait_counter(cnt1 + strip, nsplit);
        cseg_acquire_agent();
    }
    __syncthreads();
    if (tid < FU_T) {
        float mm = -INFINITY, nn = 0.f, cc = 0.f;
        for (int s_ = 0; s_ < nsplit; ++s_) {
            const float* o = part1 + ((size_t)s_ * Npad + I0 + tid) * 4;
            lse_merge(mm, nn, o[0], o[1]);
            cc += o[2];
        }
        fin[tid][0] = mm; fin[tid][1] = nn; fin[tid][2] = cc;
    }
    __syncthreads();
    // ---- phase 2: the positives, with the complete row statistics
    float slp[16], rs[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = wi * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        m[r] = fin[row][0]; ng[r] = fin[row][1];
        slp[r] = 0.f; rs[r] = 0.f;
    }
    t = 0;
    for (int jt = split; jt < nJ; jt += nsplit, ++t) {
        const int J0 = jt * FU_T;
        f32x16 acc;
        if (t < t_cache) {
            const float* st = stash + (size_t)t * FU_TILE_FLOATS + wave * (16 * 64) + lane;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = st[r * 64];
        } else {
            acc = s_tile_64(A, N, col, I0, J0, As, Cs);
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] *= inv_tau;
        }
        const int j = J0 + wj * 32 + r32;
        const bool j_ok = j < M;
        const int yj = col.label(j);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ii = I0 + wi * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (j_ok && yj == yi[r] && j != ii) {
                const float L = acc[r] - m[r];
                const float den = expf(L) + ng[r];
                slp[r] += L - logf(den);
                rs[r] += 1.f / den;
            }
        }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) {
            slp[r] += __shfl_xor(slp[r], o, 64);
            rs[r] += __shfl_xor(rs[r], o, 64);
        }
    }
    __syncthreads();                               // p1 is reused
    if (r32 == 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = wi * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            p1[wj][row][0] = slp[r]; p1[wj][row][1] = rs[r];
        }
    }
    __syncthreads();
    if (tid < FU_T) {
        float* o = part2 + ((size_t)split * Npad + I0 + tid) * 2;
        o[0] = p1[0][tid][0] + p1[1][tid][0];
        o[1] = p1[0][tid][1] + p1[1][tid][1];
    }
    __syncthreads();
    if (tid == 0) {
        cseg_release_agent();
        const int old = cseg_counter_add(cnt2 + strip, 1);
        flag = old == nsplit - 1;
        if (flag) cseg_acquire_agent();
    }
    __syncthreads();
    if (!flag) return;
    // ---- the last block of the strip: row results
    if (tid < FU_T && I0 + tid < N) {
        float sl = 0.f, rr = 0.f;
        for (int s_ = 0; s_ < nsplit; ++s_) {
            const float* o = part2 + ((size_t)s_ * Npad + I0 + tid) * 2;
            sl += o[0];
            rr += o[1];
        }
        const int i = I0 + tid;
        const float cnt = fin[tid][2];
        row_loss[i] = -coef * (sl / cnt);                 // 0/0 -> NaN exactly like the reference (no positives)
        *reinterpret_cast<float4*>(row_stats + 4 * (size_t)i) = make_float4(fin[tid][0], fin[tid][1], coef / ((float)N * cnt), rr);
    }
    __syncthreads();
    if (tid == 0) {
        cseg_release_agent();
        const int old = cseg_counter_add(done, 1);
        flag = old == n_strips - 1;
        if (flag) cseg_acquire_agent();
    }
    __syncthreads();
    if (!flag) return;
    // ---- the last strip: loss = mean_i row_loss[i] in mean_kernel's order; counters back to zero for the next launch
    float v = 0.f;
    for (int i = tid; i < N; i += 256) v += row_loss[i];
    v = block_sum(v, red, tid);
    if (tid == 0) loss[0] = v / (float)N;
    for (int i = tid; i < 2 * n_strips + 1; i += 256) cseg_counter_store(counters + i, 0);
}

// ---------------------------------------------------------------------------------------------------------
// Backward. grid = (nI * nDt, nsplit); block = 4 waves. Block (I, dt, split) owns one 32 x 32 tile of dA and the column
// tiles [split*per, (split+1)*per), which its waves interleave; the waves' partial tiles are summed through LDS.
// H = G (+ G^T) is rebuilt per (row tile, column tile) directly in the MFMA A-operand layout (swapped-operand trick,
// no LDS round trip). A variant that builds H once per column tile and keeps all 8 feature-tile accumulators in one
// wave (128 AGPRs, 1 wave/SIMD) measured 1.5-1.8x SLOWER on MI355X (63 vs 35 us at N=912, 262 vs 170 us at
// 1024x4104): at these sizes the kernel is latency-bound and the many small waves of this layout hide it better.
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float grad_elem(float s, float4 st, bool pos, bool neg) {
    // st = {m, Neg, w, R}
    const float E = expf(s - st.x);
    return pos ? -st.z * st.y / (E + st.y) : (neg ? E * st.z * st.w : 0.f);
}

constexpr int BW_DT = 4;      // 32-wide feature tiles per wave: H is rebuilt once per BW_DT tiles (2x fewer exp/label
                              // evaluations than one tile per wave at ~140 VGPRs, still 3 waves per SIMD)

template <bool SELF>
__global__ __launch_bounds__(256) void bwd_kernel(const float* __restrict__ S, int ldS, int N, ColSrc col,
                                                  const int32_t* __restrict__ a_lab,
                                                  const float* __restrict__ row_stats,
                                                  const float* __restrict__ d_loss, float inv_tau, int nDg, int nJ,
                                                  int per_split, float* __restrict__ parts) {
    __shared__ float red[32][BW_DT * 32 + 4];      // waves add their tiles one after the other (fixed order)
    __shared__ __attribute__((aligned(16))) float cstage[4][32 * 32];   // per wave: 32 contrast rows x 32 features
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int h = lane >> 5, r32 = lane & 31;
    const int It = blockIdx.x / nDg, dg = blockIdx.x % nDg;
    const int split = blockIdx.y;
    const int I0 = It * 32, d0 = dg * (BW_DT * 32);
    const int M = col.M, D = col.D;
    const int i = I0 + r32;
    const bool i_ok = i < N;
    const float4 sti = i_ok ? *reinterpret_cast<const float4*>(row_stats + 4 * (size_t)i)
                            : make_float4(0.f, 1.f, 0.f, 0.f);
    const int yi = i_ok ? a_lab[i] : -0x7ffffffe;

    f32x16 acc[BW_DT];
#pragma unroll
    for (int t = 0; t < BW_DT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    const int jt_lo = split * per_split, jt_hi = min(nJ, jt_lo + per_split);
    for (int jt = jt_lo + wave; jt < jt_hi; jt += 4) {
        const int J0 = jt * 32;
        float hv[16];
        // B operand: the 32 contrast rows of this column tile are staged feature tile by feature tile in a wave-private
        // LDS block with coalesced 16-byte loads (8 lanes per 128-byte row segment) instead of 16 strided scalar loads
        // per feature tile; missing rows (padding columns, zero tail of the bank) are staged as zeros.
        const float* srow[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) srow[u] = col.row(J0 + (lane >> 3) + 8 * u);
        float* cw = cstage[wave];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int jb = J0 + 8 * q + 4 * h;  // this lane's 4 consecutive columns for registers 4q..4q+3
            const float4 s4 = i_ok ? ld4(S + (size_t)i * ldS + jb) : make_float4(0.f, 0.f, 0.f, 0.f);
            const float sv[4] = {s4.x, s4.y, s4.z, s4.w};
            const float* rp[4];
            int lb[4];
            col.decode4(jb, rp, lb);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int j = jb + t;
                const bool j_ok = j < M;
                const int yj = lb[t];
                const bool same = (yj == yi);
                const bool pos = same && (j != i), neg = !same;
                float g = grad_elem(sv[t], sti, pos, neg);
                if (SELF) {
                    const float4 stj = j_ok ? *reinterpret_cast<const float4*>(row_stats + 4 * (size_t)j)
                                            : make_float4(0.f, 1.f, 0.f, 0.f);
                    g += grad_elem(sv[t], stj, pos, neg);
                }
                // a missing row (padding column or the zero tail of the bank) contributes 0 whatever H is
                hv[4 * q + t] = (i_ok && rp[t]) ? g : 0.f;
            }
        }
#pragma unroll
        for (int t = 0; t < BW_DT; ++t) {
            const int dbase = d0 + t * 32 + (lane & 7) * 4;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float4 v = (srow[u] && dbase < D) ? ld4(srow[u] + dbase) : make_float4(0.f, 0.f, 0.f, 0.f);
                *reinterpret_cast<float4*>(cw + ((lane >> 3) + 8 * u) * 32 + (lane & 7) * 4) = v;
            }
            CSEG_WAVE_LOCKSTEP();
            float bv[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) bv[r] = cw[(8 * (r >> 2) + 4 * h + (r & 3)) * 32 + r32];
            CSEG_WAVE_LOCKSTEP();                      // the next feature tile's stores must not overtake these reads
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(hv[r], bv[r], acc[t], 0, 0, 0);
        }
    }
    // acc[t] layout: col = d0 + t*32 + (lane&31), rows I0 + (r&3) + 8*(r>>2) + 4*h
    for (int w = 0; w < 4; ++w) {
        if (wave == w) {
#pragma unroll
            for (int t = 0; t < BW_DT; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float* p = &red[(r & 3) + 8 * (r >> 2) + 4 * h][t * 32 + r32];
                    *p = (w == 0) ? acc[t][r] : *p + acc[t][r];
                }
        }
        __syncthreads();
    }
    const float scale = d_loss[0] * inv_tau;
    for (int e = threadIdx.x; e < 32 * BW_DT * 32; e += 256) {
        const int rr = e / (BW_DT * 32), cc = e - rr * (BW_DT * 32);
        const int ii = I0 + rr, dd = d0 + cc;
        if (ii < N && dd < D) parts[((size_t)split * N + ii) * D + dd] = red[rr][cc] * scale;
    }
}

int make_col(const cseg_contrast_desc* d, ColSrc* c) {
    c->mode = d->mode;
    c->D = d->D;
    c->M = d->M;
    c->rows = nullptr; c->labs = nullptr; c->segq = nullptr; c->pixq = nullptr; c->ms = 0; c->packed = 0;
    if (d->mode == 0) {
        CSEG_REQUIRE(d->M == d->N, "contrast: self mode needs M == N (got %d, %d)", d->M, d->N);
        c->rows = d->anchors; c->labs = d->a_lab;
    } else if (d->mode == 1) {
        CSEG_REQUIRE(d->contrast && d->c_lab, "contrast: plain mode needs contrast and c_lab");
        c->rows = d->contrast; c->labs = d->c_lab;
    } else if (d->mode == 2) {
        CSEG_REQUIRE(d->segment_queue && d->pixel_queue && d->bank_classes > 0 && d->bank_size > 0,
                     "contrast: bank mode needs both queues");
        CSEG_REQUIRE(d->M == d->bank_classes * 2 * d->bank_size, "contrast: bank mode needs M == K*2*ms");
        c->segq = d->segment_queue; c->pixq = d->pixel_queue; c->ms = d->bank_size;
        c->packed = (d->bank_classes - 1) * 2 * d->bank_size;
    } else {
        cseg_set_error("contrast: unknown mode %d", d->mode);
        return 0;
    }
    CSEG_REQUIRE(d->N > 0 && d->M > 0 && d->D > 0, "contrast: empty problem");
    CSEG_REQUIRE(d->D % 8 == 0, "contrast: D=%d must be a multiple of 8", d->D);
    // the reference's scatter_(1, arange(N)) self mask raises when N > M (loss_contrast_mem.py:134-138)
    CSEG_REQUIRE(d->N <= d->M, "contrast: N=%d anchors > M=%d contrast columns", d->N, d->M);
    CSEG_REQUIRE(d->temperature > 0.f && d->base_temperature > 0.f, "contrast: temperatures must be > 0");
    return 1;
}

inline int round32(int x) { return (x + 31) / 32 * 32; }

inline int bwd_splits(int N, int M, int D) {
    const int nI = (N + 31) / 32, nDt = (D + BW_DT * 32 - 1) / (BW_DT * 32), nJ = (M + 31) / 32;
    int want = (768 + nI * nDt - 1) / (nI * nDt);      // ~3 blocks per CU
    int max_split = (nJ + 3) / 4;                       // at least one tile per wave
    int s = want < 1 ? 1 : want;
    if (s > max_split) s = max_split;
    if (s < 1) s = 1;
    if (s > 64) s = 64;
    return s;
}

}  // namespace

extern "C" size_t cseg_contrast_ws_bytes(int N, int M) { return (size_t)round32(N) * round32(M) * sizeof(float); }

extern "C" int cseg_contrast_bwd_parts(int N, int M, int D) { return bwd_splits(N, M, D); }

extern "C" int cseg_contrast_fwd(const cseg_contrast_desc* d, float* S_ws, float* row_stats, float* row_loss,
                                 float* loss, cseg_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    ColSrc col;
    if (!make_col(d, &col)) return 0;
    const int ldS = round32(d->M);
    const int nIb = (d->N + SG_T - 1) / SG_T, nJb = (d->M + SG_T - 1) / SG_T;
    hipLaunchKernelGGL(s_gemm_kernel, dim3(nIb * nJb), dim3(256), 0, stream, d->anchors, d->N, col,
                       1.0f / d->temperature, S_ws, ldS, nJb);
    CSEG_CHECK_LAUNCH("s_gemm_kernel");
    const float coef = d->temperature / d->base_temperature;
    hipLaunchKernelGGL(row_pass_kernel, dim3(d->N), dim3(256), 0, stream, S_ws, ldS, d->N, col, d->a_lab, coef,
                       row_stats, row_loss);
    CSEG_CHECK_LAUNCH("row_pass_kernel");
    hipLaunchKernelGGL(mean_kernel, dim3(1), dim3(256), 0, stream, row_loss, d->N, loss);
    CSEG_CHECK_LAUNCH("mean_kernel");
    return 1;
}

// Grid of the fused forward: n_strips x nsplit blocks, all of which must be resident at once (the blocks of a strip wait for each
// other). At most FUSED_GRID_MAX blocks of <= 80 KB of LDS: two per CU fit, so even two processes that share a GPU (the 2-rank gloo
// tests) cannot starve each other of the slots they wait for. CSEG_CONTRAST_FUSED_GRID overrides (tuning / tests).
constexpr int FUSED_GRID_MAX = 248;
constexpr size_t FUSED_LDS_CAP = 80 * 1024;
constexpr size_t FUSED_LDS_STATIC = sizeof(float) * (2 * SG_T * SG_LD + 2 * FU_T * 4 + FU_T * 4 + 4) + 16;

struct FusedPlan { int n_strips, nJ, nsplit, t_cache; size_t lds; bool resident; };

// Blocks of the fused forward that are resident at once ON THE CURRENT DEVICE (the grid was sized for a full MI355X; on a
// partitioned one -- CPX mode, ~32 CUs -- or a part with less LDS the blocks of a strip would wait for blocks that are never
// scheduled, until the 10 s trap). Asked once per device: CUs x what the occupancy calculator admits per CU at the kernel's LDS cap,
// minus a margin (MI355X_MICROARCH.md: near a register-file edge the calculator can be one block per CU high), capped at
// FUSED_GRID_MAX. 0 = the dynamic-LDS attribute cannot be raised here: no fused forward on this device.
constexpr int FUSED_MAX_DEVICES = 64;
int fused_resident_blocks() {
    static int cap[FUSED_MAX_DEVICES];
    static bool known[FUSED_MAX_DEVICES];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= FUSED_MAX_DEVICES) return 0;
    if (!known[dev]) {
        int cus = 0, per_cu = 0, c = 0;
        if (hipFuncSetAttribute((const void*)contrast_fused_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)(FUSED_LDS_CAP - FUSED_LDS_STATIC)) == hipSuccess &&
            hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess &&
            hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)contrast_fused_fwd_kernel, 256, FUSED_LDS_CAP - FUSED_LDS_STATIC) == hipSuccess)
            c = cus * (per_cu > 2 ? 2 : per_cu) - cus / 32;      // at most two per CU counted; 8 of 512 slots kept free on a full MI355X
        cap[dev] = c < 0 ? 0 : (c > FUSED_GRID_MAX ? FUSED_GRID_MAX : c);
        known[dev] = true;
    }
    return cap[dev];
}

inline FusedPlan fused_plan(int N, int M) {
    FusedPlan p;
    p.n_strips = (N + FU_T - 1) / FU_T;
    p.nJ = (M + FU_T - 1) / FU_T;
    const char* e = getenv("CSEG_CONTRAST_FUSED_GRID");
    int grid_max = e ? atoi(e) : fused_resident_blocks();
    p.resident = grid_max >= p.n_strips;            // every strip needs at least one resident block
    if (grid_max < p.n_strips) grid_max = p.n_strips;
    p.nsplit = grid_max / p.n_strips;
    if (p.nsplit > p.nJ) p.nsplit = p.nJ;
    if (p.nsplit < 1) p.nsplit = 1;
    const int per_block = (p.nJ + p.nsplit - 1) / p.nsplit;
    const int fit = (int)((FUSED_LDS_CAP - FUSED_LDS_STATIC) / (FU_TILE_FLOATS * sizeof(float)));
    p.t_cache = per_block < fit ? per_block : fit;
    p.lds = (size_t)p.t_cache * FU_TILE_FLOATS * sizeof(float);
    return p;
}

// floats of scratch: part1 [nsplit][Npad][4], part2 [nsplit][Npad][2], then 2 * n_strips + 1 int counters (ZERO at the first launch
// that uses the buffer; the kernel leaves them zero)
extern "C" size_t cseg_contrast_fused_ws_bytes(int N, int M) {
    const FusedPlan p = fused_plan(N, M);
    return ((size_t)p.nsplit * p.n_strips * FU_T * 6 + 2 * p.n_strips + 1) * sizeof(float);
}
extern "C" size_t cseg_contrast_fused_counter_offset(int N, int M) {
    const FusedPlan p = fused_plan(N, M);
    return (size_t)p.nsplit * p.n_strips * FU_T * 6 * sizeof(float);
}

extern "C" int cseg_contrast_fwd_fused(const cseg_contrast_desc* d, float* fused_ws, float* S_out, float* row_stats, float* row_loss,
                                       float* loss, cseg_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    ColSrc col;
    if (!make_col(d, &col)) return 0;
    CSEG_REQUIRE(fused_ws != nullptr, "contrast (fused forward): no scratch buffer");
    const FusedPlan p = fused_plan(d->N, d->M);
    if (!p.resident) {
        // this device cannot keep one block per strip resident (a partitioned GPU, a part with less LDS, an attribute call that
        // failed): the same outputs from the three launches -- they need the similarity array, which S_out is when the caller keeps it
        CSEG_REQUIRE(S_out != nullptr, "contrast (fused forward): %d strips do not fit this device's %d resident blocks and no S_out was given "
                     "for the three-launch form", p.n_strips, fused_resident_blocks());
        return cseg_contrast_fwd(d, S_out, row_stats, row_loss, loss, stream_);
    }
    const size_t npart = (size_t)p.nsplit * p.n_strips * FU_T;
    float* part1 = fused_ws;
    float* part2 = fused_ws + npart * 4;
    int* counters = reinterpret_cast<int*>(fused_ws + npart * 6);
    const float coef = d->temperature / d->base_temperature;
    CSEG_GRID_RESIDENT_LAUNCH();
    hipLaunchKernelGGL(contrast_fused_fwd_kernel, dim3(p.n_strips * p.nsplit), dim3(256), p.lds, stream, d->anchors, d->N, col, d->a_lab,
                       1.0f / d->temperature, coef, p.nsplit, p.n_strips, p.t_cache, part1, part2, counters, S_out, round32(d->M),
                       row_stats, row_loss, loss);
    CSEG_CHECK_LAUNCH("contrast_fused_fwd_kernel");
    return 1;
}

extern "C" int cseg_contrast_bwd(const cseg_contrast_desc* d, const float* S_ws, const float* row_stats,
                                 const float* d_loss, float* d_anchor_parts, cseg_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    ColSrc col;
    if (!make_col(d, &col)) return 0;
    const int nI = (d->N + 31) / 32, nJ = (d->M + 31) / 32, nDt = (d->D + BW_DT * 32 - 1) / (BW_DT * 32);
    const int ldS = round32(d->M);
    const int nsplit = bwd_splits(d->N, d->M, d->D);
    const int per_split = (nJ + nsplit - 1) / nsplit;
    dim3 grid(nI * nDt, nsplit);
    if (d->mode == 0)
        hipLaunchKernelGGL(bwd_kernel<true>, grid, dim3(256), 0, stream, S_ws, ldS, d->N, col, d->a_lab, row_stats,
                           d_loss, 1.0f / d->temperature, nDt, nJ, per_split, d_anchor_parts);
    else
        hipLaunchKernelGGL(bwd_kernel<false>, grid, dim3(256), 0, stream, S_ws, ldS, d->N, col, d->a_lab, row_stats,
                           d_loss, 1.0f / d->temperature, nDt, nJ, per_split, d_anchor_parts);
    CSEG_CHECK_LAUNCH("bwd_kernel");
    return 1;
}
