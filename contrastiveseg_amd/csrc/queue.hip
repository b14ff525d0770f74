// Memory-bank (segment / pixel queue) update kernels.
// Reference: segmentor/trainer_contrastive.py:102-138 (_dequeue_and_enqueue). The Python loop over
// (image, class) with unique / nonzero / mean / normalize per pair becomes: one histogram, one pass over the key
// map that accumulates every class at once, and two small row writers. Pointer arithmetic and the CPU randperm
// stay on the host (contrastiveseg_amd/segmentor/trainer_contrastive.py). HBM-bound; no MFMA.
#include "cseg_common.h"

namespace {

__global__ __launch_bounds__(256) void queue_count_kernel(const int64_t* __restrict__ labels, int H, int W, int stride,
                                                          int Hs, int Ws, int K, int32_t* __restrict__ counts) {
    extern __shared__ int hist[];
    const int b = blockIdx.y;
    for (int i = threadIdx.x; i < K; i += 256) hist[i] = 0;
    __syncthreads();
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q < Hs * Ws) {
        const int ys = q / Ws, xs = q - ys * Ws;
        const int64_t l = labels[((size_t)b * H + (size_t)ys * stride) * W + (size_t)xs * stride];
        if (l >= 0 && l < K) atomicAdd(&hist[(int)l], 1);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < K; i += 256)
        if (hist[i]) atomicAdd(&counts[(size_t)b * K + i], hist[i]);
}

// One wave per (image, channel): lanes stride over the Q positions (coalesced), every lane keeps one private
// accumulator per class of the current chunk of 32 classes; fixed-order wave reduction => deterministic sums.
__global__ __launch_bounds__(256) void queue_class_sums_kernel(const float* __restrict__ keys,
                                                               const int64_t* __restrict__ labels, int D, int Pk,
                                                               int H, int W, int stride, int Hs, int Ws, int K,
                                                               float* __restrict__ sums) {
    const int b = blockIdx.y;
    const int dch = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (dch >= D) return;
    const int Q = Hs * Ws;
    const float* kp = keys + ((size_t)b * D + dch) * Pk;
    for (int c0 = 0; c0 < K; c0 += 32) {
        float acc[32];
#pragma unroll
        for (int c = 0; c < 32; ++c) acc[c] = 0.f;
        for (int q = lane; q < Q; q += 64) {
            const int ys = q / Ws, xs = q - ys * Ws;
            const int l = (int)labels[((size_t)b * H + (size_t)ys * stride) * W + (size_t)xs * stride] - c0;
            const float v = kp[q];
#pragma unroll
            for (int c = 0; c < 32; ++c) acc[c] += (l == c) ? v : 0.f;
        }
#pragma unroll
        for (int c = 0; c < 32; ++c) {
            const float s = wave_sum(acc[c]);
            if (lane == 0 && c0 + c < K) sums[((size_t)b * K + c0 + c) * D + dch] = s;
        }
    }
}

// block per job: mean, L2 normalise (F.normalize eps 1e-12), write one bank row
__global__ __launch_bounds__(256) void queue_write_segments_kernel(const float* __restrict__ sums,
                                                                   const int32_t* __restrict__ counts,
                                                                   const int32_t* __restrict__ job_img,
                                                                   const int32_t* __restrict__ job_cls,
                                                                   const int32_t* __restrict__ job_dst, int K, int D,
                                                                   float* __restrict__ segq, int ms) {
    __shared__ float red[4];
    const int j = blockIdx.x;
    const int b = job_img[j], c = job_cls[j], row = job_dst[j];
    const float inv = 1.f / (float)counts[(size_t)b * K + c];
    const float* src = sums + ((size_t)b * K + c) * D;
    float ss = 0.f;
    for (int d = threadIdx.x; d < D; d += 256) { const float v = src[d] * inv; ss += v * v; }
    ss = wave_sum(ss);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = ss;
    __syncthreads();
    const float nrm = fmaxf(sqrtf(red[0] + red[1] + red[2] + red[3]), 1e-12f);
    float* dst = segq + ((size_t)c * ms + row) * D;
    for (int d = threadIdx.x; d < D; d += 256) dst[d] = src[d] * inv / nrm;
}

// wave per row: gather one pixel's channel vector from the NCHW key map, normalise, write
__global__ __launch_bounds__(256) void queue_write_pixels_kernel(const float* __restrict__ keys, int D, int Pk,
                                                                 const int32_t* __restrict__ src_img,
                                                                 const int32_t* __restrict__ src_pos,
                                                                 const int32_t* __restrict__ dst_cls,
                                                                 const int32_t* __restrict__ dst_row, int n_rows,
                                                                 float* __restrict__ pixq, int ms) {
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (r >= n_rows) return;
    const float* src = keys + (size_t)src_img[r] * D * Pk + src_pos[r];
    float ss = 0.f;
    for (int d = lane; d < D; d += 64) { const float v = src[(size_t)d * Pk]; ss += v * v; }
    const float nrm = fmaxf(sqrtf(wave_sum(ss)), 1e-12f);
    float* dst = pixq + ((size_t)dst_cls[r] * ms + dst_row[r]) * D;
    for (int d = lane; d < D; d += 64) dst[d] = src[(size_t)d * Pk] / nrm;
}

}  // namespace

extern "C" int cseg_queue_count(const int64_t* labels, int B, int H, int W, int stride, int K, int32_t* counts,
                                cseg_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    CSEG_REQUIRE(stride >= 1 && B > 0 && K > 0, "queue_count: bad arguments");
    const int Hs = (H + stride - 1) / stride, Ws = (W + stride - 1) / stride;
    if (hipMemsetAsync(counts, 0, sizeof(int32_t) * (size_t)B * K, stream) != hipSuccess) {
        cseg_set_error("queue_count: memset failed");
        return 0;
    }
    dim3 grid((Hs * Ws + 255) / 256, B);
    hipLaunchKernelGGL(queue_count_kernel, grid, dim3(256), sizeof(int) * K, stream, labels, H, W, stride, Hs, Ws, K,
                       counts);
    CSEG_CHECK_LAUNCH("queue_count_kernel");
    return 1;
}

extern "C" int cseg_queue_class_sums(const float* keys, const int64_t* labels, int B, int D, int Pk, int H, int W,
                                     int stride, int K, float* sums, cseg_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    CSEG_REQUIRE(stride >= 1 && B > 0 && K > 0 && D > 0, "queue_class_sums: bad arguments");
    const int Hs = (H + stride - 1) / stride, Ws = (W + stride - 1) / stride;
    // the reference indexes keys[b].view(D,-1) with positions of the strided label map (:111-120)
    CSEG_REQUIRE(Hs * Ws <= Pk, "queue_class_sums: strided label map (%d) larger than key map (%d)", Hs * Ws, Pk);
    dim3 grid((D + 3) / 4, B);
    hipLaunchKernelGGL(queue_class_sums_kernel, grid, dim3(256), 0, stream, keys, labels, D, Pk, H, W, stride, Hs, Ws,
                       K, sums);
    CSEG_CHECK_LAUNCH("queue_class_sums_kernel");
    return 1;
}

extern "C" int cseg_queue_write_segments(const float* sums, const int32_t* counts, const int32_t* job_img,
                                         const int32_t* job_cls, const int32_t* job_dst_row, int n_jobs, int K, int D,
                                         float* segment_queue, int ms, cseg_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (n_jobs <= 0) return 1;
    hipLaunchKernelGGL(queue_write_segments_kernel, dim3(n_jobs), dim3(256), 0, stream, sums, counts, job_img, job_cls,
                       job_dst_row, K, D, segment_queue, ms);
    CSEG_CHECK_LAUNCH("queue_write_segments_kernel");
    return 1;
}

extern "C" int cseg_queue_write_pixels(const float* keys, int B, int D, int Pk, const int32_t* src_img,
                                       const int32_t* src_pos, const int32_t* dst_cls, const int32_t* dst_row,
                                       int n_rows, float* pixel_queue, int ms, cseg_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    (void)B;
    if (n_rows <= 0) return 1;
    hipLaunchKernelGGL(queue_write_pixels_kernel, dim3((n_rows + 3) / 4), dim3(256), 0, stream, keys, D, Pk, src_img,
                       src_pos, dst_cls, dst_row, n_rows, pixel_queue, ms);
    CSEG_CHECK_LAUNCH("queue_write_pixels_kernel");
    return 1;
}
