// Anchor mining kernels: label nearest-downsample + argmax + hard/easy classification, per-class counts and the
// stable (ascending pixel index) partition that torch's .nonzero() order implies; gather of mined pixels from
// NCHW embeddings and the backward scatter.
//
// Reference: lib/loss/loss_contrast.py:30-89, 130-147 (see include/cseg_hip.h for the line-by-line mapping).
// All of this is integer/HBM-bound work: coalesced loads along the pixel axis, LDS histograms, wave ballots
// and prefix scans; no MFMA.
#include "cseg_common.h"

namespace {

constexpr int CLS_THREADS = 256;

// One thread per low-resolution pixel. seg reads are coalesced along p for every class plane.
__global__ __launch_bounds__(CLS_THREADS) void classify_kernel(
    const float* __restrict__ seg, const int64_t* __restrict__ pred_in, const int64_t* __restrict__ target, int K,
    int h, int w, int H, int W, float scale_y, float scale_x, int ignore_label, int32_t* __restrict__ lab_out, int32_t* __restrict__ pred_out,
    int16_t* __restrict__ key_out, int32_t* __restrict__ counts, int32_t* __restrict__ status) {
    extern __shared__ int hist[];  // [2K]
    const int P = h * w;
    const int b = blockIdx.y;
    const int p = blockIdx.x * CLS_THREADS + threadIdx.x;
    for (int i = threadIdx.x; i < 2 * K; i += CLS_THREADS) hist[i] = 0;
    __syncthreads();
    int bad = 0;
    if (p < P) {
        const int y = p / w, x = p - y * w;
        // legacy nearest: src = min(floor(dst * float(in)/float(out)), in - 1)
        int sy = (int)floorf((float)y * scale_y);
        int sx = (int)floorf((float)x * scale_x);
        sy = sy < H - 1 ? sy : H - 1;
        sx = sx < W - 1 ? sx : W - 1;
        // the reference round-trips the label through float32 before .long()
        const int lab = (int)(float)target[((size_t)b * H + sy) * W + sx];
        int arg = 0;
        if (seg) {
            const float* s = seg + (size_t)b * K * P + p;
            float best = s[0];
            for (int k = 1; k < K; ++k) {
                const float v = s[(size_t)k * P];
                // torch.max(dim): first maximal value; NaN wins over numbers
                if (v > best || (v != v && best == best)) { best = v; arg = k; }
            }
        } else {
            arg = (int)pred_in[(size_t)b * P + p];
        }
        int key = -1;
        if (lab != ignore_label) {
            if (lab >= 0 && lab < K) key = 2 * lab + (arg == lab ? 1 : 0);
            else bad = 1;
        }
        if (lab_out) lab_out[(size_t)b * P + p] = lab;
        if (pred_out) pred_out[(size_t)b * P + p] = arg;
        key_out[(size_t)b * P + p] = (int16_t)key;
        if (key >= 0) atomicAdd(&hist[key], 1);
    }
    if (bad) atomicAdd(&status[0], 1);
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * K; i += CLS_THREADS) {
        const int v = hist[i];
        if (v) atomicAdd(&counts[(size_t)b * 2 * K + i], v);
    }
}

// 8 consecutive int16 keys of one lane: one 16-byte load when aligned and fully in range
__device__ __forceinline__ void load_keys8(const int16_t* __restrict__ kb, int p0, int p_hi, bool vec_ok, int16_t (&ks)[8]) {
    if (vec_ok && p0 + 8 <= p_hi) {
        const int4 v = *reinterpret_cast<const int4*>(kb + p0);
        const int w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            ks[2 * t] = (int16_t)(w[t] & 0xffff);
            ks[2 * t + 1] = (int16_t)((unsigned)w[t] >> 16);
        }
    } else {
#pragma unroll
        for (int t = 0; t < 8; ++t) ks[t] = (p0 + t < p_hi) ? kb[p0 + t] : (int16_t)-1;
    }
}

constexpr int PART_THREADS = 1024;
constexpr int PART_WAVES = PART_THREADS / 64;

// One block per (image, class). Every wave owns a contiguous chunk of the image's pixels, each lane 8
// consecutive pixels of it per step, so ranks inside a (class, hard|easy) list follow ascending pixel index.
__global__ __launch_bounds__(PART_THREADS) void partition_kernel(
    const int16_t* __restrict__ key, const int32_t* __restrict__ counts, int K, int P,
    int32_t* __restrict__ seg_off, int32_t* __restrict__ part_idx) {
    __shared__ int wave_cnt[2][PART_WAVES];
    __shared__ int base_off[2];
    const int b = blockIdx.y, c = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int32_t* cnt = counts + (size_t)b * 2 * K;
    if (wv == 0) {
        // exclusive offsets of key 2c and 2c+1 inside image b: sum of counts of smaller keys
        int acc = 0;
        for (int i = lane; i < 2 * c; i += 64) acc += cnt[i];
        acc = wave_sum_i(acc);
        if (lane == 0) {
            base_off[0] = acc;
            base_off[1] = acc + cnt[2 * c];
            seg_off[((size_t)b * K + c) * 2 + 0] = acc;
            seg_off[((size_t)b * K + c) * 2 + 1] = acc + cnt[2 * c];
        }
    }
    if (cnt[2 * c] + cnt[2 * c + 1] == 0) return;  // block-uniform
    const int chunk = ((P + PART_WAVES * 8 - 1) / (PART_WAVES * 8)) * 8;  // multiple of 8 pixels per wave
    const int p_lo = wv * chunk;
    const int p_hi = min(P, p_lo + chunk);
    const int16_t* kb = key + (size_t)b * P;
    const bool vec_ok = ((P & 7) == 0);      // every lane's 8-key group is then 16-byte aligned
    const int16_t k_hard = (int16_t)(2 * c), k_easy = (int16_t)(2 * c + 1);
    // pass 1: per-wave totals
    int n_h = 0, n_e = 0;
    for (int p0 = p_lo + lane * 8; p0 < p_hi; p0 += 64 * 8) {
        int16_t ks[8];
        load_keys8(kb, p0, p_hi, vec_ok, ks);
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            n_h += (ks[t] == k_hard);
            n_e += (ks[t] == k_easy);
        }
    }
    n_h = wave_sum_i(n_h);
    n_e = wave_sum_i(n_e);
    if (lane == 0) { wave_cnt[0][wv] = n_h; wave_cnt[1][wv] = n_e; }
    __syncthreads();
    int off_h = base_off[0], off_e = base_off[1];
    for (int i = 0; i < wv; ++i) { off_h += wave_cnt[0][i]; off_e += wave_cnt[1][i]; }
    // pass 2: ranks + writes
    int32_t* out = part_idx + (size_t)b * P;
    for (int p0 = p_lo + lane * 8; p0 - lane * 8 < p_hi; p0 += 64 * 8) {  // wave-uniform trip count
        int16_t ks[8];
        int c_h = 0, c_e = 0;
        load_keys8(kb, p0, p_hi, vec_ok, ks);
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            c_h += (ks[t] == k_hard);
            c_e += (ks[t] == k_easy);
        }
        const int inc_h = wave_incl_scan_i(c_h, lane), inc_e = wave_incl_scan_i(c_e, lane);
        int r_h = off_h + inc_h - c_h, r_e = off_e + inc_e - c_e;
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            if (ks[t] == k_hard) out[r_h++] = p0 + t;
            else if (ks[t] == k_easy) out[r_e++] = p0 + t;
        }
        off_h += __shfl(inc_h, 63, 64);
        off_e += __shfl(inc_e, 63, 64);
    }
}

// One wave per anchor row: lanes stride over the D channel planes of the NCHW embedding.
__global__ __launch_bounds__(256) void gather_kernel(const float* __restrict__ embed, int D, int P,
                                                     const int32_t* __restrict__ part_idx,
                                                     const int32_t* __restrict__ sel_pos, int N,
                                                     float* __restrict__ anchors, int32_t* __restrict__ sel_pix) {
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (r >= N) return;
    const int pos = sel_pos[r];
    const int b = pos / P;
    const int pix = part_idx[pos];
    if (lane == 0) sel_pix[r] = b * P + pix;
    const float* src = embed + (size_t)b * D * P + pix;
    for (int d = lane; d < D; d += 64) anchors[(size_t)r * D + d] = src[(size_t)d * P];
}

__global__ __launch_bounds__(256) void scatter_kernel(const float* __restrict__ parts, int n_parts,
                                                      const int32_t* __restrict__ sel_pix, int N, int D, int P,
                                                      float scale, float* __restrict__ d_embed) {
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (r >= N) return;
    const int bp = sel_pix[r];
    const int b = bp / P, pix = bp - b * P;
    float* dst = d_embed + (size_t)b * D * P + pix;
    for (int d = lane; d < D; d += 64) {
        float acc = 0.f;
        for (int s = 0; s < n_parts; ++s) acc += parts[((size_t)s * N + r) * D + d];
        dst[(size_t)d * P] = acc * scale;
    }
}

}  // namespace

extern "C" int cseg_classify_partition(const float* seg, const int64_t* pred_in, const int64_t* target, int B, int K, int h, int w, int H,
                                       int W, int ignore_label, int32_t* lab, int32_t* pred, int16_t* key,
                                       int32_t* counts, int32_t* seg_off, int32_t* part_idx, int32_t* status,
                                       cseg_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    CSEG_REQUIRE(B > 0 && K > 0 && h > 0 && w > 0 && H > 0 && W > 0, "classify_partition: empty shape");
    CSEG_REQUIRE(seg || pred_in, "classify_partition: need seg or pred_in");
    CSEG_REQUIRE(K <= 16383, "classify_partition: K=%d does not fit the int16 key", K);
    CSEG_REQUIRE((size_t)B * h * w < ((size_t)1 << 31), "classify_partition: B*h*w overflows int32 indices");
    const int P = h * w;
    if (hipMemsetAsync(counts, 0, sizeof(int32_t) * (size_t)B * K * 2, stream) != hipSuccess ||
        hipMemsetAsync(status, 0, sizeof(int32_t) * 4, stream) != hipSuccess) {
        cseg_set_error("classify_partition: memset failed");
        return 0;
    }
    const float scale_y = (float)H / (float)h, scale_x = (float)W / (float)w;
    dim3 g1((P + CLS_THREADS - 1) / CLS_THREADS, B);
    hipLaunchKernelGGL(classify_kernel, g1, dim3(CLS_THREADS), sizeof(int) * 2 * K, stream, seg, pred_in, target, K, h, w,
                       H, W, scale_y, scale_x, ignore_label, lab, pred, key, counts, status);
    CSEG_CHECK_LAUNCH("classify_kernel");
    dim3 g2(K, B);
    hipLaunchKernelGGL(partition_kernel, g2, dim3(PART_THREADS), 0, stream, key, counts, K, P, seg_off, part_idx);
    CSEG_CHECK_LAUNCH("partition_kernel");
    return 1;
}

extern "C" int cseg_gather_anchors(const float* embed, int B, int D, int P, const int32_t* part_idx,
                                   const int32_t* sel_pos, int N, float* anchors, int32_t* sel_pix,
                                   cseg_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    (void)B;
    if (N <= 0) return 1;
    hipLaunchKernelGGL(gather_kernel, dim3((N + 3) / 4), dim3(256), 0, stream, embed, D, P, part_idx, sel_pos, N,
                       anchors, sel_pix);
    CSEG_CHECK_LAUNCH("gather_kernel");
    return 1;
}

extern "C" int cseg_scatter_anchor_grad(const float* d_anchor_parts, int n_parts, const int32_t* sel_pix, int N,
                                        int D, int P, float scale, float* d_embed, cseg_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (N <= 0) return 1;
    CSEG_REQUIRE(n_parts >= 1, "scatter_anchor_grad: n_parts=%d", n_parts);
    hipLaunchKernelGGL(scatter_kernel, dim3((N + 3) / 4), dim3(256), 0, stream, d_anchor_parts, n_parts, sel_pix, N,
                       D, P, scale, d_embed);
    CSEG_CHECK_LAUNCH("scatter_kernel");
    return 1;
}
