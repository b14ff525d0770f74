// TEST INFRASTRUCTURE -- CPU emulation of the execution model the split-bf16 kernels are written against, so that the
// kernel SOURCES under contrastiveseg_amd/csrc can be run functionally in `-m "not gpu"` tests on a machine without a GPU.
// This header stands in for <hip/hip_runtime.h> when tests/emu/build_emu.py compiles a .hip file for the host
// (x86-64 clang, -I tests/emu first on the include path). Nothing here is part of the product.
//
// Model: one kernel launch = blocks executed one after the other; the threads of a block are fibers (ucontext) run by a
// cooperative scheduler, wave by wave (64 consecutive threads). A thread runs until it reaches
//   __syncthreads()                      -> block barrier: released when every live thread of the block has arrived
//   a wave-level operation (MFMA, shfl)  -> wave rendezvous: the 64 lanes exchange registers, then continue
// Threads that return drop out of the barriers, as exited waves do on the hardware. A pass of the scheduler in which no
// thread makes progress is a deadlock (divergent barrier) and aborts with a message.
// The order in which waves are scheduled between two block barriers is a launch parameter (ascending, descending or a
// seeded shuffle: CSEG_EMU_WAVE_ORDER): a result that depends on it exposes a missing barrier between a producer and a
// consumer wave.
// Emulated device operations (semantics from the CDNA3/4 ISA guides; the MFMA operand layout is additionally pinned by
// kernels that have passed parity on an MI355X and must keep passing here):
//   v_mfma_f32_16x16x32_bf16   A[i][k]: lane i + 16*(k/8), element k%8;  B[k][j]: lane j + 16*(k/8), element k%8;
//                              D[i][j]: lane j + 16*(i/4), register i%4; fp32 accumulation in k order
//   global_load_lds_dwordx4    LDS[base + 16*lane .. +15] = *global pointer of the lane (base is wave-uniform)
//   v_alignbit_b32, __shfl_xor / __shfl_up (width 64)
// Dynamic LDS is one 160 KB buffer, filled with 0xFF (bf16 / fp32 NaN patterns) before every block so that a read of a
// cell no thread has written shows up in the result; a launch asking for more than 160 KB fails like the hardware does.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>

#define CSEG_WAVE_LOCKSTEP() emu::wave_sync()
#define CSEG_GRID_RESIDENT_LAUNCH() emu::request_resident()
#define CSEG_SPIN_PAUSE() emu::spin_pause()
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __shared__ static thread_local   // block-scope arrays (one block runs per OS thread); `extern __shared__` lines are
                                         // rewritten by build_emu.py to point at emu::dyn_lds()

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct emu_uint3 { unsigned x, y, z; };

struct alignas(16) uint4 { unsigned x, y, z, w; };
struct alignas(8) uint2 { unsigned x, y; };
struct alignas(16) int4 { int x, y, z, w; };
struct alignas(8) int2 { int x, y; };
static inline int4 make_int4(int x, int y, int z, int w) { return int4{x, y, z, w}; }
static inline int2 make_int2(int x, int y) { return int2{x, y}; }
struct alignas(16) float4 { float x, y, z, w; };
struct alignas(8) float2 { float x, y; };
static inline uint4 make_uint4(unsigned x, unsigned y, unsigned z, unsigned w) { return uint4{x, y, z, w}; }
static inline uint2 make_uint2(unsigned x, unsigned y) { return uint2{x, y}; }
static inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }
static inline float2 make_float2(float x, float y) { return float2{x, y}; }
using std::max;
using std::min;

typedef void* hipStream_t;
typedef int hipError_t;
enum { hipSuccess = 0, hipErrorLaunchFailure = 719 };
enum { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };

namespace emu {
constexpr int WAVE = 64;
constexpr int LDS_BYTES = 160 * 1024;
extern thread_local emu_uint3 t_threadIdx, t_blockIdx;
extern thread_local dim3 t_blockDim, t_gridDim;
unsigned char* dyn_lds();
int lane_id();
void block_barrier();
constexpr int XSTRIDE = 64;
// wave rendezvous: every live lane of the calling thread's wave deposits `bytes` (<= XSTRIDE) from `mine` into slot `lane`
// of the wave's exchange area ([64][XSTRIDE] bytes) and receives a pointer to the area once all lanes have arrived;
// wave_release() is the second rendezvous of the operation: no lane starts its next exchange before all have read.
const unsigned char* wave_exchange(const void* mine, int bytes);
void wave_release();
inline void wave_sync() { wave_exchange(nullptr, 0); }      // CSEG_WAVE_LOCKSTEP(): lockstep hand-over through LDS
template <class T>
inline const T& slot(const unsigned char* area, int lane) { return *reinterpret_cast<const T*>(area + (size_t)lane * XSTRIDE); }
hipError_t launch(dim3 grid, dim3 block, size_t lds_bytes, const std::function<void()>& body);
// Kernels whose blocks wait for each other (a counter another block of the SAME launch increments): the launch site announces it
// (CSEG_GRID_RESIDENT_LAUNCH) and the next launch of this host thread gets one OS thread per block, so that every block is "resident";
// a waiting lane calls spin_pause() between polls (CSEG_SPIN_PAUSE: s_sleep on the GPU).
void request_resident();
void spin_pause();
hipError_t last_error();
}  // namespace emu

#define threadIdx (emu::t_threadIdx)
#define blockIdx (emu::t_blockIdx)
#define blockDim (emu::t_blockDim)
#define gridDim (emu::t_gridDim)

static inline void __syncthreads() { emu::block_barrier(); }
static inline hipError_t hipGetLastError() { return emu::last_error(); }
static inline const char* hipGetErrorString(hipError_t e) { return e == hipSuccess ? "no error" : "emulated launch failure"; }
template <class F>
static inline hipError_t hipFuncSetAttribute(F, int, int bytes) { return bytes <= emu::LDS_BYTES ? hipSuccess : hipErrorLaunchFailure; }
// the emulated device: a full MI355X (256 CUs; as many blocks per CU as 160 KB of LDS hold)
enum { hipDeviceAttributeMultiprocessorCount = 63 };
static inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
static inline hipError_t hipDeviceGetAttribute(int* v, int, int) { *v = 256; return hipSuccess; }
template <class F>
static inline hipError_t hipOccupancyMaxActiveBlocksPerMultiprocessor(int* n, F, int, size_t lds) { *n = lds ? (int)(emu::LDS_BYTES / lds) : 8; return hipSuccess; }
// CSEG_EMU_TRACE=<file>: one line per launch with the kernel expression as written at the launch site (which kernel a routing
// switch really took -- tests assert on it)
static inline void emu_trace_launch(const char* what) {
    if (const char* path = getenv("CSEG_EMU_TRACE")) {
        if (FILE* f = fopen(path, "a")) {
            fprintf(f, "%s\n", what);
            fclose(f);
        }
    }
}
#define hipLaunchKernelGGL(kernel, grid, block, lds, stream, ...) \
    (emu_trace_launch(#kernel), emu::launch((grid), (block), (lds), [=]() { kernel(__VA_ARGS__); }))

// ---- device operations ---------------------------------------------------------------------------------------------------
typedef __bf16 emu_bf16x8 __attribute__((ext_vector_type(8)));
typedef float emu_f32x4 __attribute__((ext_vector_type(4)));

static inline emu_f32x4 emu_mfma_f32_16x16x32_bf16(emu_bf16x8 a, emu_bf16x8 b, emu_f32x4 c, int, int, int) {
    struct Ops { emu_bf16x8 a, b; } mine{a, b};
    const unsigned char* all = emu::wave_exchange(&mine, sizeof(Ops));
    const int lane = emu::lane_id(), j = lane & 15, g = lane >> 4;
    emu_f32x4 d = c;
    for (int r = 0; r < 4; ++r) {
        const int i = 4 * g + r;
        float acc = c[r];
        for (int k = 0; k < 32; ++k)
            acc += (float)emu::slot<Ops>(all, i + 16 * (k >> 3)).a[k & 7] * (float)emu::slot<Ops>(all, j + 16 * (k >> 3)).b[k & 7];
        d[r] = acc;
    }
    emu::wave_release();
    return d;
}
#define __builtin_amdgcn_mfma_f32_16x16x32_bf16 emu_mfma_f32_16x16x32_bf16

// v_mfma_f32_16x16x32_f16: same operand layout; fp16 products are exact in fp32 (11 x 11 bits), fp32 accumulation in k order
typedef _Float16 emu_f16x8 __attribute__((ext_vector_type(8)));
static inline emu_f32x4 emu_mfma_f32_16x16x32_f16(emu_f16x8 a, emu_f16x8 b, emu_f32x4 c, int, int, int) {
    struct Ops { emu_f16x8 a, b; } mine{a, b};
    const unsigned char* all = emu::wave_exchange(&mine, sizeof(Ops));
    const int lane = emu::lane_id(), j = lane & 15, g = lane >> 4;
    emu_f32x4 d = c;
    for (int r = 0; r < 4; ++r) {
        const int i = 4 * g + r;
        float acc = c[r];
        for (int k = 0; k < 32; ++k)
            acc += (float)emu::slot<Ops>(all, i + 16 * (k >> 3)).a[k & 7] * (float)emu::slot<Ops>(all, j + 16 * (k >> 3)).b[k & 7];
        d[r] = acc;
    }
    emu::wave_release();
    return d;
}
#define __builtin_amdgcn_mfma_f32_16x16x32_f16 emu_mfma_f32_16x16x32_f16

static inline unsigned emu_alignbit(unsigned hi, unsigned lo, unsigned shift) {
    return (unsigned)((((uint64_t)hi << 32) | lo) >> (shift & 31));
}
#define __builtin_amdgcn_alignbit emu_alignbit
#define CSEG_KEEP_DWORD(v) ((void)(v))
#define __builtin_amdgcn_sched_barrier(mask) ((void)0)      // instruction-scheduling fence: nothing to emulate

// global_load_lds: the LDS operand is the wave-uniform base, every lane lands at base + lane * size
static inline void emu_global_load_lds(const void* gptr, void* lds_base, unsigned size, int, int) {
    memcpy(static_cast<unsigned char*>(lds_base) + (size_t)emu::lane_id() * size, gptr, size);
}
#define __builtin_amdgcn_global_load_lds(g, l, size, off, aux) \
    emu_global_load_lds((const void*)(g), (void*)(l), (size), (off), (aux))

// v_mfma_f32_32x32x2_f32: A[i][k] lane i + 32k, B[k][j] lane j + 32k, D[i][j] lane j + 32*((i/4)%2), register i%4 + 4*(i/8)
template <class V16>
static inline V16 emu_mfma_f32_32x32x2f32(float a, float b, V16 c, int, int, int) {
    struct Ops { float a, b; } mine{a, b};
    const unsigned char* all = emu::wave_exchange(&mine, sizeof(Ops));
    const int lane = emu::lane_id(), j = lane & 31, h = lane >> 5;
    V16 d = c;
    for (int v = 0; v < 16; ++v) {
        const int i = (v & 3) + 8 * (v >> 2) + 4 * h;
        float acc = c[v];
        for (int k = 0; k < 2; ++k) acc += emu::slot<Ops>(all, i + 32 * k).a * emu::slot<Ops>(all, j + 32 * k).b;
        d[v] = acc;
    }
    emu::wave_release();
    return d;
}
#define __builtin_amdgcn_mfma_f32_32x32x2f32 emu_mfma_f32_32x32x2f32

// v_mfma_f32_16x16x4_f32: A[i][k] lane i + 16k, B[k][j] lane j + 16k, D[i][j] lane j + 16*(i/4), register i%4
template <class V4>
static inline V4 emu_mfma_f32_16x16x4f32(float a, float b, V4 c, int, int, int) {
    struct Ops { float a, b; } mine{a, b};
    const unsigned char* all = emu::wave_exchange(&mine, sizeof(Ops));
    const int lane = emu::lane_id(), j = lane & 15, g = lane >> 4;
    V4 d = c;
    for (int r = 0; r < 4; ++r) {
        const int i = 4 * g + r;
        float acc = c[r];
        for (int k = 0; k < 4; ++k) acc += emu::slot<Ops>(all, i + 16 * k).a * emu::slot<Ops>(all, j + 16 * k).b;
        d[r] = acc;
    }
    emu::wave_release();
    return d;
}
#define __builtin_amdgcn_mfma_f32_16x16x4f32 emu_mfma_f32_16x16x4f32

// atomics: blocks run concurrently on several OS threads, the threads of one block never do
static inline int atomicAdd(int* p, int v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
static inline unsigned atomicAdd(unsigned* p, unsigned v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
static inline unsigned atomicMax(unsigned* p, unsigned v) {
    unsigned old = __atomic_load_n(p, __ATOMIC_RELAXED);
    while (old < v && !__atomic_compare_exchange_n(p, &old, v, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
    return old;
}
static inline unsigned long long atomicAdd(unsigned long long* p, unsigned long long v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
template <class F, class I>
static inline F emu_atomic_add_fp(F* p, F v) {
    I* ip = reinterpret_cast<I*>(p);
    I old = __atomic_load_n(ip, __ATOMIC_RELAXED);
    for (;;) {
        F f;
        memcpy(&f, &old, sizeof f);
        f += v;
        I want;
        memcpy(&want, &f, sizeof f);
        if (__atomic_compare_exchange_n(ip, &old, want, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {
            memcpy(&f, &old, sizeof f);
            return f;
        }
    }
}
static inline float atomicAdd(float* p, float v) { return emu_atomic_add_fp<float, uint32_t>(p, v); }
static inline double atomicAdd(double* p, double v) { return emu_atomic_add_fp<double, uint64_t>(p, v); }
static inline hipError_t hipMemsetAsync(void* p, int value, size_t bytes, hipStream_t) { memset(p, value, bytes); return hipSuccess; }
// individually rounded float operations (the device intrinsics of the same names forbid FMA contraction)
static inline float __fmul_rn(float a, float b) { volatile float r = a * b; return r; }
static inline float __fadd_rn(float a, float b) { volatile float r = a + b; return r; }
static inline float __fsub_rn(float a, float b) { volatile float r = a - b; return r; }
static inline float emu_fast_expf(float x) { return expf(x); }
static inline float emu_fast_logf(float x) { return logf(x); }
#define __expf emu_fast_expf          // (glibc's math.h declares functions with these names)
#define __logf emu_fast_logf

// raw buffer loads: resource = (base, bytes); an offset past the end returns 0 like the hardware's range check
struct emu_buffer_rsrc { const unsigned char* base; unsigned num_records; };
typedef emu_buffer_rsrc __amdgpu_buffer_rsrc_t;
static inline emu_buffer_rsrc emu_make_buffer_rsrc(void* p, short, int num_records, int) {
    return emu_buffer_rsrc{static_cast<const unsigned char*>(p), (unsigned)num_records};
}
static inline int emu_raw_buffer_load_b32(emu_buffer_rsrc r, int voffset, int soffset, int) {
    const unsigned o = (unsigned)voffset + (unsigned)soffset;
    int v = 0;
    if ((uint64_t)o + 4 <= r.num_records) memcpy(&v, r.base + o, 4);
    return v;
}
#define __builtin_amdgcn_make_buffer_rsrc emu_make_buffer_rsrc
#define __builtin_amdgcn_raw_buffer_load_b32 emu_raw_buffer_load_b32

template <class T>
static inline T emu_shfl_from(T v, int src_lane_or_neg) {
    const unsigned char* all = emu::wave_exchange(&v, sizeof(T));
    const T out = (src_lane_or_neg >= 0 && src_lane_or_neg < emu::WAVE) ? emu::slot<T>(all, src_lane_or_neg) : v;
    emu::wave_release();
    return out;
}
template <class T>
static inline T __shfl_xor(T v, int mask, int = 64) { return emu_shfl_from(v, emu::lane_id() ^ mask); }
template <class T>
static inline T __shfl_up(T v, int delta, int = 64) { return emu_shfl_from(v, emu::lane_id() - delta); }
template <class T>
static inline T __shfl(T v, int src, int = 64) { return emu_shfl_from(v, src & 63); }
static inline int emu_readfirstlane(int v) { return emu_shfl_from(v, 0); }       // all lanes are active in these kernels
#define __builtin_amdgcn_readfirstlane emu_readfirstlane
