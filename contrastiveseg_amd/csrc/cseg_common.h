// Shared helpers for libcseg_hip.so (gfx950 only: 64-wide wavefronts hard-coded).
#pragma once
#include <stdlib.h>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "cseg_hip.h"

#define CSEG_WAVE 64

// Marks a point where the lanes of ONE wave hand data to each other through LDS and rely on the wave executing in
// lockstep (the LDS writes of all 64 lanes have been issued, in order, before any lane's read). On the hardware no
// instruction is needed for that, but the COMPILER must not move LDS accesses across the point: a wavefront-scope release
// fence plus a wave barrier (both are scheduling constraints only -- no code is emitted for them on gfx950). The CPU emulation
// of the execution model used by the tests (tests/emu) runs lanes one after the other between rendezvous points and compiles
// this into a wave rendezvous.
#ifndef CSEG_WAVE_LOCKSTEP
#define CSEG_WAVE_LOCKSTEP()                                  \
    do {                                                      \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); \
        __builtin_amdgcn_wave_barrier();                      \
    } while (0)
#endif

// "This 32-bit value is used here": keeps the compiler from narrowing a 16-byte LDS read whose outer dwords the arithmetic does
// not need into ds_read2_b64 / ds_read2_b32 pairs -- those have a different bank pattern than the ds_read_b128 the LDS pitches
// were chosen for (weight gradient: half the banks idle, SQ_LDS_BANK_CONFLICT = 55 % of SQ_LDS_IDX_ACTIVE). No instruction is
// emitted. (tests/emu defines it as a no-op.)
#ifndef CSEG_KEEP_DWORD
#define CSEG_KEEP_DWORD(v) asm volatile("" ::"v"(v))
#endif

// 16-byte load from a UNIFORM base pointer plus a per-lane 32-bit byte offset: the form the compiler turns into
// `global_load_dwordx4 v, v_off, s[base]` (scalar base, no 64-bit vector address arithmetic). Not a buffer load: ROCm 7.2's
// __builtin_amdgcn_raw_buffer_load_b128 lowers to buffer_load_dword (ONE dword; found by the fp64 parity test on the MI355X,
// /tmp-size reproducer in DESIGN.md section 4), so 16-byte buffer loads are not available from HIP source in this toolchain.
__device__ __forceinline__ float4 cseg_load_f4(const void* uniform_base, unsigned byte_offset) {
    return *reinterpret_cast<const float4*>(static_cast<const char*>(uniform_base) + byte_offset);
}

// ---- hand-off between workgroups of ONE launch (MI355X_MICROARCH.md "Workgroup dispatch, XCD placement & inter-workgroup
// visibility"): producer = plain stores -> __syncthreads() -> ONE lane: cseg_release_agent() -> cseg_counter_add(); consumer = ONE lane
// polls cseg_counter_load() (relaxed, with CSEG_SPIN_PAUSE between polls) -> cseg_acquire_agent() -> __syncthreads() -> plain loads.
// (Per-XCD L2s are not coherent and a CU's L1 is never refreshed by another CU's stores; workgroup-scope fences are not enough. The
// explicit vmcnt(0) after the release fence is the guide's fix for a compiler pass that may drop it.) The CPU emulation of the
// execution model (tests/emu) compiles the #else branches; a launch whose blocks wait for each other announces itself with
// CSEG_GRID_RESIDENT_LAUNCH() so that the emulator runs all its blocks at once.
__device__ __forceinline__ void cseg_release_agent() {
#if defined(__AMDGCN__)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#else
    __atomic_thread_fence(__ATOMIC_SEQ_CST);
#endif
}
__device__ __forceinline__ void cseg_acquire_agent() {
#if defined(__AMDGCN__)
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
#else
    __atomic_thread_fence(__ATOMIC_SEQ_CST);
#endif
}
__device__ __forceinline__ int cseg_counter_add(int* p, int v) {       // -> the value before
#if defined(__AMDGCN__)
    return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
    return __atomic_fetch_add(p, v, __ATOMIC_SEQ_CST);
#endif
}
__device__ __forceinline__ int cseg_counter_load(const int* p) {
#if defined(__AMDGCN__)
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
    return __atomic_load_n(p, __ATOMIC_SEQ_CST);
#endif
}
__device__ __forceinline__ void cseg_counter_store(int* p, int v) {
#if defined(__AMDGCN__)
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
    __atomic_store_n(p, v, __ATOMIC_SEQ_CST);
#endif
}
#ifndef CSEG_SPIN_PAUSE
#if defined(__AMDGCN__)
#define CSEG_SPIN_PAUSE() __builtin_amdgcn_s_sleep(16)
#else
#define CSEG_SPIN_PAUSE() ((void)0)
#endif
#endif
#ifndef CSEG_GRID_RESIDENT_LAUNCH
#define CSEG_GRID_RESIDENT_LAUNCH() ((void)0)
#endif
// one lane waits until *counter reaches `want`; a wait that cannot end (a block of the launch that never became resident: a grid larger
// than the chip holds, which the launchers rule out) ends in a trap after ~10 s instead of hanging the device
__device__ __forceinline__ void cseg_wait_counter(const int* counter, int want) {
    for (long spins = 0; cseg_counter_load(counter) < want; ++spins) {
        CSEG_SPIN_PAUSE();
        if (spins > (1L << 25)) __builtin_trap();
    }
}

// Four consecutive outputs v[0..3] of one NCHW row at columns xx .. xx + 3 (xx % 4 == 0), optional epilogue addend of the same layout.
// Rows of a tensor whose width (or, for the flat-plane kernels, H*W) is a multiple of 4 floats start 16-byte aligned: one dwordx4 store
// (and load). Round 5: other widths (the 65 x 129 maps of DeepLab-R101-d8, the 130 / 65 / 33 / 17-wide maps of HRNet at 520 x 520) go
// element by element -- a uniform branch, so the aligned shapes pay nothing.
typedef float cseg_f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void cseg_store_row4(float* __restrict__ orow, const float* __restrict__ arow, long xx, long W, bool vec,
                                                cseg_f32x4 v) {
    if (vec && xx + 3 < W) {
        if (arow) {
            const float4 ad = *reinterpret_cast<const float4*>(arow + xx);
            v[0] += ad.x; v[1] += ad.y; v[2] += ad.z; v[3] += ad.w;
        }
        *reinterpret_cast<float4*>(orow + xx) = make_float4(v[0], v[1], v[2], v[3]);
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (xx + k < W) orow[xx + k] = v[k] + (arow ? arow[xx + k] : 0.f);
    }
}

void cseg_set_error(const char* fmt, ...);

// reference convention: 1 = ok, 0 = error (lib/extensions/cc_attention/src/ca.cu:199-204)
// XCD-aware block order for grids of independent tiles (round 4; default since round 5, CSEG_XCD_REMAP=0 switches it off). Workgroup b
// of a grid runs on XCD b % 8 (observed on gfx950; only speed depends on it), so consecutive tiles -- neighbours that share halo rows,
// the channel tile groups that read one patch -- land in 8 different L2 caches. The remap gives every XCD a contiguous run of n / 8
// logical blocks, walked in order (the kernels then decode the logical index with the channel tile group fastest); it needs
// n % 8 == 0 and is the identity otherwise. Measured on the MI355X, A/B/A/B on one box, outputs bit-identical
// (profiles/r05_xcd_remap_one_tile.txt): 96 channels at 8 x 64 x 128 42.5 / 42.0 -> 40.1 / 39.9 us with statistics, 64 channels at
// 8 x 128 x 256 76.5 / 77.7 -> 73.6 / 74.7, 192 channels 1-2 %, the 720-channel head 5.46 / 5.48 -> 5.36 / 5.38 ms. (The persistent
// kernels of conv3x3_sb16.hip have their own form of it: -7 % at 48 channels)
__device__ __forceinline__ int cseg_xcd_block(int b, int n, int on) { return (on && (n & 7) == 0) ? (b & 7) * (n >> 3) + (b >> 3) : b; }
static inline int cseg_xcd_remap() {
    const char* e = getenv("CSEG_XCD_REMAP");
    return e && atoi(e) == 0 ? 0 : 1;
}

#define CSEG_CHECK_LAUNCH(name)                                                     \
    do {                                                                            \
        hipError_t e__ = hipGetLastError();                                         \
        if (e__ != hipSuccess) {                                                    \
            cseg_set_error("%s: %s", name, hipGetErrorString(e__));                 \
            return 0;                                                               \
        }                                                                           \
    } while (0)

#define CSEG_REQUIRE(cond, ...)                                                     \
    do {                                                                            \
        if (!(cond)) {                                                              \
            cseg_set_error(__VA_ARGS__);                                            \
            return 0;                                                               \
        }                                                                           \
    } while (0)

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ int wave_sum_i(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// inclusive prefix sum over the 64 lanes
__device__ __forceinline__ int wave_incl_scan_i(int v, int lane) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        int n = __shfl_up(v, o, 64);
        if (lane >= o) v += n;
    }
    return v;
}

// torch's area_pixel_compute_source_index for align_corners=True, fp32 (ATen UpSample.h): src = scale * dst
__host__ __device__ __forceinline__ float ac_scale(int in_size, int out_size) {
    return out_size > 1 ? (float)(in_size - 1) / (float)(out_size - 1) : 0.0f;
}
