// Hardware facts the split-fp16 ("f16x3") kernels rely on, measured rather than assumed (gfx950):
//   1. does v_mfma_f32_16x16x32_f16 keep fp16 SUBNORMAL inputs (the lo pieces of small operands) or flush them?
//   2. f16 vs bf16 MFMA issue rate (16x16x32 and 32x32x16), one and two waves per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_f16_probe.hip -o tools/probes/mfma_f16_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ void denorm_kernel(float a_val, float b_val, float* out) {
    f16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (_Float16)a_val; b[j] = (_Float16)b_val; }
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc, 0, 0, 0);
    if (threadIdx.x == 0) out[0] = acc[0];
}

template <int KIND>
__global__ __launch_bounds__(512) void rate_kernel(int iters, float* out) {
    f32x4 acc4[8];
    f32x16 acc16[4];
    for (int i = 0; i < 8; ++i) acc4[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) acc16[i][j] = 0.f;
    f16x8 ah, bh;
    bf16x8 ab, bb;
    typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
    f16x4 a4, b4;
    for (int j = 0; j < 4; ++j) { a4[j] = (_Float16)(0.001f * (threadIdx.x + j)); b4[j] = (_Float16)(0.002f * (threadIdx.x ^ j)); }
    for (int j = 0; j < 8; ++j) {
        ah[j] = (_Float16)(0.001f * (threadIdx.x + j)); bh[j] = (_Float16)(0.002f * (threadIdx.x ^ j));
        ab[j] = (__bf16)(0.001f * (threadIdx.x + j)); bb[j] = (__bf16)(0.002f * (threadIdx.x ^ j));
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (KIND == 0) acc4[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, acc4[i], 0, 0, 0);
            if (KIND == 1) acc4[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab, bb, acc4[i], 0, 0, 0);
            if (KIND == 4) acc4[i] = __builtin_amdgcn_mfma_f32_16x16x16f16(a4, b4, acc4[i], 0, 0, 0);      // the K = 16 form (round 6: is it half the time?)
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (KIND == 2) acc16[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc16[i], 0, 0, 0);
            if (KIND == 3) acc16[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, acc16[i], 0, 0, 0);
        }
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += acc4[i][0];
    for (int i = 0; i < 4; ++i) s += acc16[i][0];
    if (s == 12345.f) out[0] = s;
}

template <int KIND>
void rate(const char* name, int threads, float* d) {
    const int iters = 20000, blocks = 256 * 4;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    rate_kernel<KIND><<<blocks, threads>>>(100, d);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    rate_kernel<KIND><<<blocks, threads>>>(iters, d);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    const double per = KIND == 4 ? 8.0 * 16 * 16 * 16 * 2 : KIND < 2 ? 8.0 * 16 * 16 * 32 * 2 : 4.0 * 32 * 32 * 16 * 2;
    const double flop = per * iters * (threads / 64) * blocks;
    printf("{\"probe\": \"mfma_rate\", \"inst\": \"%s\", \"threads_per_block\": %d, \"ms\": %.3f, \"TFLOPs\": %.1f}\n", name, threads, ms,
           flop / ms * 1e-9);
}

int main() {
    float* d;
    hipMalloc(&d, 64);
    struct { float a, b; const char* what; } cases[] = {
        {5.9604645e-8f, 1024.f, "a = 2^-24 (smallest fp16 subnormal), b = 2^10: kept -> 32 * 2^-14 = 0.001953125"},
        {3.0517578e-5f, 1.f, "a = 2^-15 (subnormal), b = 1: kept -> 32 * 2^-15 = 0.0009765625"},
        {6.1035156e-5f, 1.f, "a = 2^-14 (smallest normal), b = 1: 0.001953125"},
    };
    for (auto& c : cases) {
        float h = -1.f;
        denorm_kernel<<<1, 64>>>(c.a, c.b, d);
        hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
        printf("{\"probe\": \"mfma_f16_denorm\", \"case\": \"%s\", \"result\": %.10g}\n", c.what, h);
    }
    rate<0>("v_mfma_f32_16x16x32_f16", 256, d);
    rate<1>("v_mfma_f32_16x16x32_bf16", 256, d);
    rate<2>("v_mfma_f32_32x32x16_f16", 256, d);
    rate<3>("v_mfma_f32_32x32x16_bf16", 256, d);
    rate<4>("v_mfma_f32_16x16x16_f16", 256, d);
    rate<4>("v_mfma_f32_16x16x16_f16", 512, d);
    rate<0>("v_mfma_f32_16x16x32_f16", 512, d);
    rate<1>("v_mfma_f32_16x16x32_bf16", 512, d);
    return 0;
}
