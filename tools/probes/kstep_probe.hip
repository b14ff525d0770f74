// What bounds the K-loop of the split-fp16 3x3 kernels? (round 6). The grouped kernel's computing waves (conv3x3_group.hip) run
// 4 pixel tiles x 3 channel tiles x 3 piece products = 36 v_mfma_f32_16x16x32_f16 per K-step from 14 ds_read_b128, five K-steps per
// barrier, beside staging waves (global loads, fp32 -> two fp16 pieces, ds_write_b128) and the weight DMA (global_load_lds). This probe
// rebuilds that instruction mix piece by piece on LDS filled with constants and reports shader cycles per MFMA and SIMD for each mix:
//   reads   0 fragments loaded once | 1 the kernel's 14 reads per K-step | 2 the same reads, software-pipelined (next group's first)
//   barrier 0 / 1 one __syncthreads per five K-steps
//   stage   0 no staging waves | 1 staging waves: VALU split + ds_write_b128 | 2 + their global loads
//   dma     0 / 1 thirty 1 KB global_load_lds rows per chunk from the computing waves
//   cw      computing waves per block (4 = one per SIMD, 8 = two per SIMD); staging waves: 4 when stage > 0
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/probes/kstep_probe.hip -o tools/probes/kstep_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int XC = 66, GPLANE = 672, NP = 2, NOCT = 2, GNT = 3, STEPS = 5;
constexpr int A_CELLS = NP * NOCT * GPLANE;          // uint4 per patch buffer
constexpr int BSTEP = GNT * NP * 64, BCHUNK = STEPS * BSTEP;
constexpr int LDS_CELLS = 2 * A_CELLS + 2 * BCHUNK;

template <int READS, int BARRIER, int STAGE, int DMA, int CW, int RT>
__global__ __launch_bounds__((CW + (STAGE ? 4 : 0)) * 64) void kstep_kernel(int chunks, const float* __restrict__ x, const uint4* __restrict__ w,
                                                                              float* out, long long* cyc) {
    extern __shared__ __attribute__((aligned(16))) uint4 lds[];
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int g = lane >> 4, n = lane & 15;
    for (int i = tid; i < LDS_CELLS; i += blockDim.x) lds[i] = make_uint4(0x3c003c00u, 0x3c003c00u, 0x38003800u, 0x38003800u);
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
    uint4* As = lds;
    uint4* Bs = lds + 2 * A_CELLS;
    if (wave < CW) {
        constexpr int MT = 4 * RT;                   // pixel tiles per wave (RT = 1: 4, RT = 2: 8)
        f32x4 acc[MT][GNT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < GNT; ++nt) acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        int c_off[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) c_off[mt] = ((wave * RT + mt / 4) & 7) * XC + 16 * (mt & 3) + n;
        int buf = 0;
        f16x8 af[2][MT][NP], bf[2][NP];
        if (READS == 0) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int p = 0; p < NP; ++p) af[0][mt][p] = __builtin_bit_cast(f16x8, As[p * NOCT * GPLANE + c_off[mt]]);
#pragma unroll
            for (int p = 0; p < NP; ++p) bf[0][p] = __builtin_bit_cast(f16x8, Bs[p * 64 + lane]);
        }
#pragma unroll 1
        for (int c = 0; c < chunks; ++c) {
            const uint4* a_base = As + (size_t)buf * A_CELLS;
            const uint4* b_base = Bs + (size_t)buf * BCHUNK + lane;
            if (DMA) {
#pragma unroll
                for (int i = 0; i < (STEPS * GNT * NP + CW - 1) / CW; ++i) {
                    const int r = wave + CW * i;
                    if (r < STEPS * GNT * NP)
                        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(w + (size_t)(c & 7) * BCHUNK + r * 64 + lane),
                                                         (__attribute__((address_space(3))) void*)(Bs + (size_t)(buf ^ 1) * BCHUNK + r * 64), 16, 0, 0);
                }
            }
            auto load_a = [&](int s, f16x8 (&dst)[MT][NP]) {
                const int tap = min(2 * s + (g >> 1), 8);
                const int ky = tap / 3, kx = tap - 3 * ky;
                const uint4* ap = a_base + (g & 1) * GPLANE + ky * XC + kx;
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int p = 0; p < NP; ++p) dst[mt][p] = __builtin_bit_cast(f16x8, ap[p * NOCT * GPLANE + c_off[mt]]);
            };
            auto load_b = [&](int i, f16x8 (&dst)[NP]) {
                const int s = i / GNT, nt = i - s * GNT;
#pragma unroll
                for (int p = 0; p < NP; ++p) dst[p] = __builtin_bit_cast(f16x8, b_base[s * BSTEP + (nt * NP + p) * 64]);
            };
            if (READS == 2) { load_a(0, af[0]); load_b(0, bf[0]); }
#pragma unroll
            for (int i = 0; i < STEPS * GNT; ++i) {
                const int s = i / GNT, nt = i - s * GNT;
                int sa = 0, sb = 0;
                if (READS == 1) {
                    if (nt == 0) load_a(s, af[0]);
                    load_b(i, bf[0]);
                } else if (READS == 2) {
                    if (i + 1 < STEPS * GNT) load_b(i + 1, bf[(i + 1) & 1]);
                    if (nt == 0 && s + 1 < STEPS) load_a(s + 1, af[(s + 1) & 1]);
                    __builtin_amdgcn_sched_barrier(0);
                    sa = s & 1; sb = i & 1;
                }
#pragma unroll
                for (int t = 0; t < 3; ++t)
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[sa][mt][t == 2 ? 1 : 0], bf[sb][t == 1 ? 1 : 0], acc[mt][nt], 0, 0, 0);
            }
            if (BARRIER) __syncthreads();
            buf ^= 1;
        }
        float s = 0.f;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < GNT; ++nt) s += acc[mt][nt][0] + acc[mt][nt][3];
        if (s == 12345.f) out[0] = s;
    } else {
        // staging waves: 256 threads, (8 + 2) x 66 x 2 octets = 1320 items of 8 channels -> PAU = 6 items per thread
        constexpr int PAU = 6;
        const int pt = tid - CW * 64;
        int buf = 0;
        const float xs = 0.37f;
#pragma unroll 1
        for (int c = 0; c < chunks; ++c) {
            float v[PAU][8];
#pragma unroll
            for (int u = 0; u < PAU; ++u)
#pragma unroll
                for (int j = 0; j < 8; ++j) v[u][j] = STAGE == 2 ? x[(size_t)((c & 15) * 16 + (u & 1) * 8 + j) * 32768 + (blockIdx.x & 31) * 1024 + ((pt + 256 * (u >> 1)) & 1023)] : (float)(pt + u + j + c);
            uint4* dst = As + (size_t)(buf ^ 1) * A_CELLS;
#pragma unroll
            for (int u = 0; u < PAU; ++u) {
                const int item = pt + 256 * u;
                if (item < 1320) {
                    unsigned hi[4], lo[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float a0 = v[u][2 * j] * xs, a1 = v[u][2 * j + 1] * xs;
                        const _Float16 h0 = (_Float16)a0, h1 = (_Float16)a1;
                        const _Float16 l0 = (_Float16)(a0 - (float)h0), l1 = (_Float16)(a1 - (float)h1);
                        hi[j] = (unsigned)__builtin_bit_cast(unsigned short, h0) | (unsigned)__builtin_bit_cast(unsigned short, h1) << 16;
                        lo[j] = (unsigned)__builtin_bit_cast(unsigned short, l0) | (unsigned)__builtin_bit_cast(unsigned short, l1) << 16;
                    }
                    const int cell = item >= 660 ? GPLANE + item - 660 : item;
                    dst[cell] = make_uint4(hi[0], hi[1], hi[2], hi[3]);
                    dst[NOCT * GPLANE + cell] = make_uint4(lo[0], lo[1], lo[2], lo[3]);
                }
            }
            if (BARRIER) __syncthreads();
            buf ^= 1;
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}

struct Case { const char* name; void (*fn)(int, const float*, const uint4*, float*, long long*); int threads; int cw; int rt; };

int main(int argc, char** argv) {
    const int chunks = argc > 1 ? atoi(argv[1]) : 2000;
    float *x, *out;
    uint4* w;
    long long* cyc;
    hipMalloc(&x, (size_t)256 * 32768 * 4); hipMemset(x, 0x3c, (size_t)256 * 32768 * 4);
    hipMalloc(&w, (size_t)8 * BCHUNK * 16); hipMemset(w, 0x3c, (size_t)8 * BCHUNK * 16);
    hipMalloc(&out, 64); hipMalloc(&cyc, 256 * 8);
    const size_t lds = (size_t)LDS_CELLS * 16;                  // 144 KB: one block per CU
#define CASE(R, B, S, D, CW, RT) {"reads" #R " barrier" #B " stage" #S " dma" #D " cw" #CW " rt" #RT, kstep_kernel<R, B, S, D, CW, RT>, (CW + (S ? 4 : 0)) * 64, CW, RT}
    Case cases[] = {
        CASE(0, 0, 0, 0, 4, 1), CASE(0, 0, 0, 0, 8, 1), CASE(0, 0, 0, 0, 4, 2),
        CASE(1, 0, 0, 0, 4, 1), CASE(1, 0, 0, 0, 8, 1), CASE(1, 0, 0, 0, 4, 2),
        CASE(2, 0, 0, 0, 4, 1), CASE(2, 0, 0, 0, 8, 1),
        CASE(1, 1, 0, 0, 8, 1), CASE(2, 1, 0, 0, 8, 1), CASE(1, 1, 0, 0, 4, 2),
        CASE(1, 1, 0, 1, 8, 1), CASE(2, 1, 0, 1, 8, 1),
        CASE(1, 1, 1, 0, 8, 1), CASE(1, 1, 2, 0, 8, 1), CASE(1, 1, 2, 1, 8, 1), CASE(2, 1, 2, 1, 8, 1),
        CASE(1, 1, 1, 0, 4, 2), CASE(1, 1, 2, 1, 4, 2),
    };
    for (const Case& cs : cases) {
        hipFuncSetAttribute((const void*)cs.fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(cs.fn, dim3(256), dim3(cs.threads), lds, 0, 50, x, w, out, cyc);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(cs.fn, dim3(256), dim3(cs.threads), lds, 0, chunks, x, w, out, cyc);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        { hipError_t er = hipGetLastError(); if (er != hipSuccess) { printf("%s: launch failed: %s\n", cs.name, hipGetErrorString(er)); continue; } }
        float ms = 0.f;
        hipEventElapsedTime(&ms, e0, e1);
        long long hc[256];
        hipMemcpy(hc, cyc, sizeof hc, hipMemcpyDeviceToHost);
        double mean = 0;
        for (int i = 0; i < 256; ++i) mean += (double)hc[i] / 256;
        const double mfma_per_simd = (double)chunks * STEPS * GNT * 12 * cs.rt * cs.cw / 4;
        const double tf = 256.0 * 4 * mfma_per_simd * 16384 / (ms * 1e-3) * 1e-12;
        printf("{\"case\": \"%s\", \"us_per_chunk\": %.3f, \"cycles_per_mfma_simd\": %.2f, \"f16_tflops\": %.0f, \"ghz\": %.2f}\n", cs.name,
               ms * 1e3 / chunks, mean / mfma_per_simd, tf, mean / (ms * 1e6));
        fflush(stdout);
    }
    return 0;
}
