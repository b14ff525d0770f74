// One launch per optimizer step instead of three per layer: max|w| of every split-operand convolution weight
// (cseg_amax_batch) and the split + packed forms of all of them, forward and backward-data operators (cseg_split_pack_batch).
// Round-3 trace of the benched step: 213 weight maxima + 426 pack launches of ~5 us each = 3.9 ms of GPU time per step and
// as many host-side launches, for 140 MB of weights. The jobs live in a device-side table of cseg_split_job records
// (include/cseg_hip.h) that the host builds once; blocks find their job by binary search over the jobs' first block index.
#include "cseg_pack.h"

namespace {

__device__ __forceinline__ int find_job(const cseg_split_job* __restrict__ jobs, int n_jobs, int block) {
    int lo = 0, hi = n_jobs - 1;                   // last job with block0 <= block
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (jobs[mid].block0 <= block) lo = mid; else hi = mid - 1;
    }
    return lo;
}

template <class AR>
__global__ __launch_bounds__(256) void pack_batch_kernel(const cseg_split_job* __restrict__ jobs, int n_jobs) {
    const cseg_split_job jb = jobs[find_job(jobs, n_jobs, blockIdx.x)];
    const float wscale = AR::SCALED ? split_scale_of(split_amax_exp(jb.amax)) : 1.f;      // every thread (shuffles inside)
    const int e = (blockIdx.x - jb.block0) * 256 + threadIdx.x;
    if (e >= jb.total) return;
    const float* w = static_cast<const float*>(jb.src);
    uint4* wp = static_cast<uint4*>(jb.dst);
    if (jb.kind == CSEG_PACK_C3) pack_elem_c3<AR>(w, jb.cout, jb.cin, jb.flag, jb.nt, wscale, wp, e);
    else if (jb.kind == CSEG_PACK_C3_16) pack_elem_c3_16<AR>(w, jb.cout, jb.cin, jb.flag, jb.nt, wscale, wp, e);
    else if (jb.kind == CSEG_PACK_C3_S2T) pack_elem_c3_s2t<AR>(w, jb.cout, jb.cin, jb.nt, wscale, wp, e);
    else pack_elem_c1<AR>(w, jb.cout, jb.cin, jb.flag, jb.nt, wscale, wp, e);
}

// job.total = number of floats, job.dst unused; the job's blocks stride over its tensor
__global__ __launch_bounds__(256) void amax_batch_kernel(const cseg_split_job* __restrict__ jobs, int n_jobs, int total_blocks) {
    __shared__ unsigned red[4];
    const int j = find_job(jobs, n_jobs, blockIdx.x);
    const cseg_split_job jb = jobs[j];
    const int nb = (j + 1 < n_jobs ? jobs[j + 1].block0 : total_blocks) - jb.block0;
    const unsigned* x = static_cast<const unsigned*>(jb.src);
    unsigned m = 0;
    for (long i = (long)(blockIdx.x - jb.block0) * 256 + threadIdx.x; i < jb.total; i += (long)nb * 256) m = max(m, x[i] & 0x7fffffffu);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) amax_publish_block(max(max(red[0], red[1]), max(red[2], red[3])), jb.amax);
}

}  // namespace

extern "C" int cseg_amax_batch(const cseg_split_job* jobs_dev, int n_jobs, int total_blocks, cseg_stream_t stream_) {
    CSEG_REQUIRE(jobs_dev && n_jobs > 0 && total_blocks >= n_jobs, "amax_batch: bad job table");
    hipLaunchKernelGGL(amax_batch_kernel, dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)stream_, jobs_dev, n_jobs,
                       total_blocks);
    CSEG_CHECK_LAUNCH("amax_batch_kernel");
    return 1;
}

extern "C" int cseg_split_pack_batch(const cseg_split_job* jobs_dev, int n_jobs, int total_blocks, int arith, cseg_stream_t stream_) {
    CSEG_REQUIRE(jobs_dev && n_jobs > 0 && total_blocks >= n_jobs, "split_pack_batch: bad job table");
    CSEG_REQUIRE(arith == CSEG_ARITH_BF16X6 || arith == CSEG_ARITH_F16X3, "split_pack_batch: unknown arithmetic %d", arith);
    if (arith == CSEG_ARITH_F16X3)
        hipLaunchKernelGGL(pack_batch_kernel<SplitF16x3>, dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)stream_, jobs_dev, n_jobs);
    else
        hipLaunchKernelGGL(pack_batch_kernel<SplitBF16x6>, dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)stream_, jobs_dev, n_jobs);
    CSEG_CHECK_LAUNCH("pack_batch_kernel");
    return 1;
}
