// Contrastive term of PixelContrastLoss on gfx950.
//
// Reference: lib/loss/loss_contrast.py:91-128 (self), lib/loss/loss_contrast_mem.py:91-152 (memory bank).
// Restated for parity in oracle/cseg_oracle.py:_contrast_core.
//
// Kernels
//   s_gemm_kernel    S = A . C^T / tau           fp32 MFMA (v_mfma_f32_32x32x2_f32), one wave per 32x32 tile,
//                                                operands streamed row-contiguously (16 B per lane) from L2
//   row_pass_kernel  per anchor row: max, sum of negatives, positive count, mean log-prob of positives
//   mean_kernel      loss = mean_i row_loss[i]  (fixed-order reduction: results are run-to-run deterministic)
//   bwd_kernel       dA = dloss/tau . H . C,  H = G (+ G^T in self mode) rebuilt on the fly from S and the row
//                    statistics directly in the MFMA A-operand register layout (no LDS round trip), C rows
//                    streamed from L2 as the B operand; partial sums over column splits go to d_anchor_parts.
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
};

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }

// ---------------------------------------------------------------------------------------------------------
// S = A . C^T / tau.  Block = 4 waves, each wave one 32x32 tile. ldS = M rounded up to 32.
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void s_gemm_kernel(const float* __restrict__ A, int N, ColSrc col, float inv_tau,
                                                     float* __restrict__ S, int ldS, int nJ, int n_tiles) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int t = blockIdx.x * 4 + wave;
    if (t >= n_tiles) return;
    const int I0 = (t / nJ) * 32, J0 = (t % nJ) * 32;
    const int h = lane >> 5, r32 = lane & 31;
    const int D = col.D, Dh = D >> 1;
    const int i = I0 + r32, j = J0 + r32;
    const float* ap = i < N ? A + (size_t)i * D + h * Dh : nullptr;
    const float* cp = col.row(j);
    if (cp) cp += h * Dh;

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;

    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 a_cur[8], c_cur[8], a_nxt[8], c_nxt[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const bool ok = 4 * q < Dh;
        a_cur[q] = (ok && ap) ? ld4(ap + 4 * q) : z4;
        c_cur[q] = (ok && cp) ? ld4(cp + 4 * q) : z4;
    }
    for (int kc = 0; kc < Dh; kc += 32) {
        const int kn = kc + 32;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const bool ok = kn + 4 * q < Dh;
            a_nxt[q] = (ok && ap) ? ld4(ap + kn + 4 * q) : z4;
            c_nxt[q] = (ok && cp) ? ld4(cp + kn + 4 * q) : z4;
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[q].x, c_cur[q].x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[q].y, c_cur[q].y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[q].z, c_cur[q].z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[q].w, c_cur[q].w, acc, 0, 0, 0);
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) { a_cur[q] = a_nxt[q]; c_cur[q] = c_nxt[q]; }
    }
    // C layout: this lane holds column j, rows I0 + (r&3) + 8*(r>>2) + 4*h
    if (j < ldS) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ii = I0 + (r & 3) + 8 * (r >> 2) + 4 * h;
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
// Backward. grid = (nI * nDt, nsplit); block = 4 waves. Block (I, dt, split) owns column tiles
// [split*per, (split+1)*per) and its waves interleave them; partial 32x32 tiles are summed through LDS.
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float grad_elem(float s, float4 st, bool pos, bool neg) {
    // st = {m, Neg, w, R}
    const float E = expf(s - st.x);
    return pos ? -st.z * st.y / (E + st.y) : (neg ? E * st.z * st.w : 0.f);
}

template <bool SELF>
__global__ __launch_bounds__(256) void bwd_kernel(const float* __restrict__ S, int ldS, int N, ColSrc col,
                                                  const int32_t* __restrict__ a_lab,
                                                  const float* __restrict__ row_stats,
                                                  const float* __restrict__ d_loss, float inv_tau, int nDt, int nJ,
                                                  int per_split, float* __restrict__ parts) {
    __shared__ float red[4][32][33];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int h = lane >> 5, r32 = lane & 31;
    const int It = blockIdx.x / nDt, dt = blockIdx.x % nDt;
    const int split = blockIdx.y;
    const int I0 = It * 32, d0 = dt * 32;
    const int M = col.M, D = col.D;
    const int i = I0 + r32;
    const bool i_ok = i < N;
    const float4 sti = i_ok ? *reinterpret_cast<const float4*>(row_stats + 4 * (size_t)i)
                            : make_float4(0.f, 1.f, 0.f, 0.f);
    const int yi = i_ok ? a_lab[i] : -0x7ffffffe;
    const int d = d0 + r32;
    const bool d_ok = d < D;

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;

    const int jt_lo = split * per_split, jt_hi = min(nJ, jt_lo + per_split);
    for (int jt = jt_lo + wave; jt < jt_hi; jt += 4) {
        const int J0 = jt * 32;
        float hv[16];
        float bv[16];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int jb = J0 + 8 * q + 4 * h;  // this lane's 4 consecutive columns for registers 4q..4q+3
            const float4 s4 = i_ok ? ld4(S + (size_t)i * ldS + jb) : make_float4(0.f, 0.f, 0.f, 0.f);
            const float sv[4] = {s4.x, s4.y, s4.z, s4.w};
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int j = jb + t;
                const bool j_ok = j < M;
                const int yj = col.label(j);
                const bool same = (yj == yi);
                const bool pos = same && (j != i), neg = !same;
                float g = grad_elem(sv[t], sti, pos, neg);
                if (SELF) {
                    const float4 stj = j_ok ? *reinterpret_cast<const float4*>(row_stats + 4 * (size_t)j)
                                            : make_float4(0.f, 1.f, 0.f, 0.f);
                    g += grad_elem(sv[t], stj, pos, neg);
                }
                hv[4 * q + t] = (i_ok && j_ok) ? g : 0.f;
                const float* cr = col.row(j);
                bv[4 * q + t] = (cr && d_ok) ? cr[d] : 0.f;
            }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(hv[r], bv[r], acc, 0, 0, 0);
    }
    // acc layout: col = d0 + (lane&31), rows I0 + (r&3) + 8*(r>>2) + 4*h
#pragma unroll
    for (int r = 0; r < 16; ++r) red[wave][(r & 3) + 8 * (r >> 2) + 4 * h][r32] = acc[r];
    __syncthreads();
    const float scale = d_loss[0] * inv_tau;
    for (int e = threadIdx.x; e < 32 * 32; e += 256) {
        const int rr = e >> 5, cc = e & 31;
        const int ii = I0 + rr, dd = d0 + cc;
        if (ii < N && dd < D) {
            const float v = red[0][rr][cc] + red[1][rr][cc] + red[2][rr][cc] + red[3][rr][cc];
            parts[((size_t)split * N + ii) * D + dd] = v * scale;
        }
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
    const int nI = (N + 31) / 32, nDt = (D + 31) / 32, nJ = (M + 31) / 32;
    int want = (512 + nI * nDt - 1) / (nI * nDt);      // ~2 blocks per CU
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
    const int nI = (d->N + 31) / 32, nJ = (d->M + 31) / 32, ldS = round32(d->M);
    const int n_tiles = nI * nJ;
    hipLaunchKernelGGL(s_gemm_kernel, dim3((n_tiles + 3) / 4), dim3(256), 0, stream, d->anchors, d->N, col,
                       1.0f / d->temperature, S_ws, ldS, nJ, n_tiles);
    CSEG_CHECK_LAUNCH("s_gemm_kernel");
    const float coef = d->temperature / d->base_temperature;
    hipLaunchKernelGGL(row_pass_kernel, dim3(d->N), dim3(256), 0, stream, S_ws, ldS, d->N, col, d->a_lab, coef,
                       row_stats, row_loss);
    CSEG_CHECK_LAUNCH("row_pass_kernel");
    hipLaunchKernelGGL(mean_kernel, dim3(1), dim3(256), 0, stream, row_loss, d->N, loss);
    CSEG_CHECK_LAUNCH("mean_kernel");
    return 1;
}

extern "C" int cseg_contrast_bwd(const cseg_contrast_desc* d, const float* S_ws, const float* row_stats,
                                 const float* d_loss, float* d_anchor_parts, cseg_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    ColSrc col;
    if (!make_col(d, &col)) return 0;
    const int nI = (d->N + 31) / 32, nJ = (d->M + 31) / 32, nDt = (d->D + 31) / 32, ldS = round32(d->M);
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
