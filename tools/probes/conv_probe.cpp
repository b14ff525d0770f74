// Torch-free timing probe of the split-operand 3x3 convolutions through the C-ABI (include/cseg_hip.h): seconds on the GPU box
// instead of the minutes a Python process needs to import torch, so that kernel variants can be compared within a small budget.
//   g++ -O2 -std=c++17 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude tools/probes/conv_probe.cpp -o tools/probes/conv_probe \
//       -L/opt/rocm/lib -lamdhip64 -ldl
//   tools/probes/conv_probe [--shape B,C,H,W]... [--variant 'name:KEY=VAL;KEY=VAL']... [--wrw] [--c1] [--pair] [--iters N] [--nt N]
// For every shape: x (non-negative, like an activation) and w are generated on the host, max|.| records and packed weights are
// made by the library, then every variant (a set of environment switches the library reads per call) is timed with HIP events:
// forward with the BatchNorm statistics epilogue (`fwd_st`), plain forward (`fwd`), optionally the weight gradient. The output of
// each variant is compared with the first one's. One JSON line per (shape, variant).
#include <dlfcn.h>
#include <hip/hip_runtime_api.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "cseg_hip.h"

#define HIPCHECK(e)                                                                       \
    do {                                                                                  \
        hipError_t err_ = (e);                                                            \
        if (err_ != hipSuccess) {                                                         \
            fprintf(stderr, "%s:%d: %s\n", __FILE__, __LINE__, hipGetErrorString(err_));  \
            exit(2);                                                                      \
        }                                                                                 \
    } while (0)

static void* g_lib;
template <class F>
static F sym(const char* name) {
    void* p = dlsym(g_lib, name);
    if (!p) { fprintf(stderr, "missing symbol %s\n", name); exit(2); }
    return reinterpret_cast<F>(p);
}

struct Variant { std::string name; std::vector<std::pair<std::string, std::string>> env; };
struct Shape { int B, C, H, W; };

static uint32_t g_seed = 12345u;
static float urand() { g_seed = g_seed * 1664525u + 1013904223u; return (float)(g_seed >> 8) * (1.0f / 16777216.0f); }
static float nrand() { float s = 0.f; for (int i = 0; i < 4; ++i) s += urand(); return (s - 2.f) * 1.7320508f; }

template <class Fn>
static double time_us(Fn&& fn, int iters, hipStream_t st) {
    hipEvent_t e0, e1;
    HIPCHECK(hipEventCreate(&e0));
    HIPCHECK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) fn();
    HIPCHECK(hipStreamSynchronize(st));
    double best = 1e30;
    for (int r = 0; r < 3; ++r) {
        HIPCHECK(hipEventRecord(e0, st));
        for (int i = 0; i < iters; ++i) fn();
        HIPCHECK(hipEventRecord(e1, st));
        HIPCHECK(hipEventSynchronize(e1));
        float ms = 0.f;
        HIPCHECK(hipEventElapsedTime(&ms, e0, e1));
        best = std::min(best, (double)ms * 1e3 / iters);
    }
    HIPCHECK(hipEventDestroy(e0));
    HIPCHECK(hipEventDestroy(e1));
    return best;
}

int main(int argc, char** argv) {
    std::vector<Shape> shapes;
    std::vector<Variant> variants;
    bool wrw = false, pair = false, c1 = false;
    int iters = 20, nt = 0;
    std::vector<std::string> keys;
    for (int i = 1; i < argc; ++i) {
        std::string a = argv[i];
        if (a == "--shape" && i + 1 < argc) {
            Shape s;
            if (sscanf(argv[++i], "%d,%d,%d,%d", &s.B, &s.C, &s.H, &s.W) != 4) { fprintf(stderr, "bad shape\n"); return 2; }
            shapes.push_back(s);
        } else if (a == "--variant" && i + 1 < argc) {
            std::string v = argv[++i];
            Variant var;
            size_t c = v.find(':');
            var.name = v.substr(0, c);
            std::string rest = c == std::string::npos ? "" : v.substr(c + 1);
            while (!rest.empty()) {
                size_t comma = rest.find(';');
                std::string kv = rest.substr(0, comma);
                rest = comma == std::string::npos ? "" : rest.substr(comma + 1);
                size_t eq = kv.find('=');
                if (eq == std::string::npos) continue;
                var.env.push_back({kv.substr(0, eq), kv.substr(eq + 1)});
                keys.push_back(kv.substr(0, eq));
            }
            variants.push_back(var);
        } else if (a == "--wrw") wrw = true;
        else if (a == "--iters" && i + 1 < argc) iters = atoi(argv[++i]);
        else if (a == "--c1") c1 = true;              // also: the 1x1 convolution C -> C on the same tensor (`c1_us`)
        else if (a == "--pair") pair = true;          // also: two launches side by side on two streams (own outputs), wall clock per pair
        else if (a == "--nt" && i + 1 < argc) nt = atoi(argv[++i]);          // channel tiling of pack + forward (265 = CSEG_NT_SB8: the head kernel)
    }
    if (shapes.empty()) shapes = {{8, 48, 128, 256}, {8, 96, 64, 128}, {8, 192, 32, 64}, {8, 384, 16, 32}};
    if (variants.empty()) variants.push_back({"default", {}});

    const char* libpath = getenv("CSEG_LIB") ? getenv("CSEG_LIB") : "contrastiveseg_amd/libcseg_hip.so";
    g_lib = dlopen(libpath, RTLD_NOW | RTLD_GLOBAL);
    if (!g_lib) { fprintf(stderr, "dlopen %s: %s\n", libpath, dlerror()); return 2; }
    auto p_amax = sym<decltype(&cseg_amax_f32)>("cseg_amax_f32");
    auto p_bytes = sym<decltype(&cseg_conv3x3_split_packed_bytes)>("cseg_conv3x3_split_packed_bytes");
    auto p_pack = sym<decltype(&cseg_conv3x3_split_pack)>("cseg_conv3x3_split_pack");
    auto p_fwd = sym<decltype(&cseg_conv3x3_split_fwd)>("cseg_conv3x3_split_fwd");
    auto p_fwd_st = sym<decltype(&cseg_conv3x3_split_fwd_st)>("cseg_conv3x3_split_fwd_st");
    auto p_seg = sym<decltype(&cseg_conv_stat_segments)>("cseg_conv_stat_segments");
    auto p_wrw_ws = sym<decltype(&cseg_conv3x3_sb_wrw_ws_floats)>("cseg_conv3x3_sb_wrw_ws_floats");
    auto p_wrw = sym<decltype(&cseg_conv3x3_split_wrw)>("cseg_conv3x3_split_wrw");
    auto p_err = sym<decltype(&cseg_last_error)>("cseg_last_error");
    auto p_c1_bytes = sym<decltype(&cseg_conv1x1_split_packed_bytes)>("cseg_conv1x1_split_packed_bytes");
    auto p_c1_pack = sym<decltype(&cseg_conv1x1_split_pack)>("cseg_conv1x1_split_pack");
    auto p_c1_fwd = sym<decltype(&cseg_conv1x1_split_fwd)>("cseg_conv1x1_split_fwd");

    HIPCHECK(hipSetDevice(0));
    hipStream_t st;
    HIPCHECK(hipStreamCreate(&st));

    {   // launch floor: a near-empty kernel back to back on the stream (max|.| of 64 floats)
        float* t;
        unsigned* r;
        HIPCHECK(hipMalloc(&t, 4096));
        HIPCHECK(hipMalloc(&r, CSEG_AMAX_WORDS * 4));
        HIPCHECK(hipMemset(t, 0, 4096));
        HIPCHECK(hipMemset(r, 0, CSEG_AMAX_WORDS * 4));
        bool ok = true;
        const double us = time_us([&]() { ok = ok && p_amax(t, 64, r, st); }, 200, st);
        printf("{\"launch_floor_us\": %.2f, \"ok\": %d}\n", us, (int)ok);
        HIPCHECK(hipFree(t));
        HIPCHECK(hipFree(r));
    }
    for (const Shape& s : shapes) {
        const size_t n = (size_t)s.B * s.C * s.H * s.W, nw = (size_t)s.C * s.C * 9;
        std::vector<float> hx(n), hw(nw), hy(n), hy0, hdw(nw), hdw0;
        for (auto& v : hx) v = std::max(nrand(), 0.f);
        const float ws = 1.f / (3.f * std::sqrt((float)s.C));
        for (auto& v : hw) v = nrand() * ws;
        float *x, *w, *y, *dy, *stats, *wsb = nullptr, *dw = nullptr;
        unsigned* rec;
        void* wp;
        const size_t T = p_seg(0, s.B, s.H, s.W);
        HIPCHECK(hipMalloc(&x, n * 4));
        HIPCHECK(hipMalloc(&y, n * 4));
        HIPCHECK(hipMalloc(&dy, n * 4));
        HIPCHECK(hipMalloc(&w, nw * 4));
        HIPCHECK(hipMalloc(&stats, (size_t)s.C * T * 16));
        HIPCHECK(hipMalloc(&rec, 3 * CSEG_AMAX_WORDS * 4));
        HIPCHECK(hipMalloc(&wp, p_bytes(CSEG_ARITH_F16X3, s.C, s.C)));
        HIPCHECK(hipMemcpy(x, hx.data(), n * 4, hipMemcpyHostToDevice));
        HIPCHECK(hipMemcpy(w, hw.data(), nw * 4, hipMemcpyHostToDevice));
        for (auto& v : hx) v = nrand() * 1e-3f;                   // dy: a gradient-sized tensor
        HIPCHECK(hipMemcpy(dy, hx.data(), n * 4, hipMemcpyHostToDevice));
        HIPCHECK(hipMemsetAsync(rec, 0, 3 * CSEG_AMAX_WORDS * 4, st));
        unsigned *ax = rec, *aw = rec + CSEG_AMAX_WORDS, *ady = rec + 2 * CSEG_AMAX_WORDS;
        if (!p_amax(x, (long)n, ax, st) || !p_amax(w, (long)nw, aw, st) || !p_amax(dy, (long)n, ady, st)) {
            fprintf(stderr, "amax: %s\n", p_err());
            return 2;
        }
        if (wrw) {
            HIPCHECK(hipMalloc(&wsb, p_wrw_ws(s.B, s.C, s.C, s.H, s.W) * 4));
            HIPCHECK(hipMalloc(&dw, nw * 4));
        }
        bool first = true;
        for (const Variant& v : variants) {
            for (const auto& k : keys) unsetenv(k.c_str());
            for (const auto& kv : v.env) setenv(kv.first.c_str(), kv.second.c_str(), 1);
            // pack under the variant's switches too (pack and forward of one operator must agree on the tiling)
            if (!p_pack(w, s.C, s.C, 0, nt, CSEG_ARITH_F16X3, aw, wp, st)) { fprintf(stderr, "pack: %s\n", p_err()); return 2; }
            bool ok = true;
            auto f_st = [&]() { ok = ok && p_fwd_st(x, wp, nullptr, s.B, s.C, s.C, s.H, s.W, nt, CSEG_ARITH_F16X3, ax, aw, y, stats, st); };
            auto f_pl = [&]() { ok = ok && p_fwd(x, wp, nullptr, s.B, s.C, s.C, s.H, s.W, nt, CSEG_ARITH_F16X3, ax, aw, y, st); };
            HIPCHECK(hipMemsetAsync(y, 0, n * 4, st));
            const double us_st = time_us(f_st, iters, st);
            const double us_pl = time_us(f_pl, iters, st);
            if (!ok) { fprintf(stderr, "forward (%s): %s\n", v.name.c_str(), p_err()); return 2; }
            HIPCHECK(hipMemcpy(hy.data(), y, n * 4, hipMemcpyDeviceToHost));
            double maxdiff = 0.0, maxabs = 0.0;
            if (first) hy0 = hy;
            for (size_t i = 0; i < n; ++i) {
                maxdiff = std::max(maxdiff, (double)std::fabs(hy[i] - hy0[i]));
                maxabs = std::max(maxabs, (double)std::fabs(hy0[i]));
            }
            double us_pair = -1.0;
            if (pair) {
                static hipStream_t st2 = nullptr;
                static float* y2 = nullptr;
                static size_t y2n = 0;
                if (!st2) HIPCHECK(hipStreamCreate(&st2));
                if (y2n < n) { if (y2) HIPCHECK(hipFree(y2)); HIPCHECK(hipMalloc(&y2, n * 4)); y2n = n; }
                auto both = [&]() {
                    ok = ok && p_fwd(x, wp, nullptr, s.B, s.C, s.C, s.H, s.W, nt, CSEG_ARITH_F16X3, ax, aw, y, st);
                    ok = ok && p_fwd(x, wp, nullptr, s.B, s.C, s.C, s.H, s.W, nt, CSEG_ARITH_F16X3, ax, aw, y2, st2);
                };
                for (int i = 0; i < 3; ++i) both();
                HIPCHECK(hipDeviceSynchronize());
                us_pair = 1e30;
                for (int r = 0; r < 3; ++r) {
                    hipEvent_t e0, e1, e2;
                    HIPCHECK(hipEventCreate(&e0)); HIPCHECK(hipEventCreate(&e1)); HIPCHECK(hipEventCreate(&e2));
                    HIPCHECK(hipEventRecord(e0, st));
                    HIPCHECK(hipStreamWaitEvent(st2, e0, 0));            // both streams start together
                    for (int i = 0; i < iters; ++i) both();
                    HIPCHECK(hipEventRecord(e2, st2));
                    HIPCHECK(hipStreamWaitEvent(st, e2, 0));             // ... and the clock stops when both are done
                    HIPCHECK(hipEventRecord(e1, st));
                    HIPCHECK(hipEventSynchronize(e1));
                    float ms = 0.f;
                    HIPCHECK(hipEventElapsedTime(&ms, e0, e1));
                    us_pair = std::min(us_pair, (double)ms * 1e3 / iters);
                    HIPCHECK(hipEventDestroy(e0)); HIPCHECK(hipEventDestroy(e1)); HIPCHECK(hipEventDestroy(e2));
                }
                if (!ok) { fprintf(stderr, "pair (%s): %s\n", v.name.c_str(), p_err()); return 2; }
            }
            double us_c1 = -1.0, c1diff = 0.0;
            if (c1) {                      // the first C x C weights of `w` as a 1x1 operator
                static std::vector<float> hc1, hc10;
                void* wp1;
                HIPCHECK(hipMalloc(&wp1, p_c1_bytes(CSEG_ARITH_F16X3, s.C, s.C)));
                if (!p_c1_pack(w, s.C, s.C, 0, CSEG_ARITH_F16X3, aw, wp1, st)) { fprintf(stderr, "c1 pack: %s\n", p_err()); return 2; }
                auto f1 = [&]() { ok = ok && p_c1_fwd(x, wp1, nullptr, s.B, s.C, s.C, s.H * s.W, CSEG_ARITH_F16X3, ax, aw, y, st); };
                us_c1 = time_us(f1, iters, st);
                if (!ok) { fprintf(stderr, "c1 (%s): %s\n", v.name.c_str(), p_err()); return 2; }
                hc1.resize(n);
                HIPCHECK(hipMemcpy(hc1.data(), y, n * 4, hipMemcpyDeviceToHost));
                if (first) hc10 = hc1;
                for (size_t i = 0; i < n; ++i) c1diff = std::max(c1diff, (double)std::fabs(hc1[i] - hc10[i]));
                HIPCHECK(hipFree(wp1));
            }
            double us_wrw = -1.0, wdiff = 0.0;
            if (wrw) {
                auto f_w = [&]() { ok = ok && p_wrw(x, dy, s.B, s.C, s.C, s.H, s.W, CSEG_ARITH_F16X3, ax, ady, wsb, dw, st); };
                us_wrw = time_us(f_w, iters, st);
                if (!ok) { fprintf(stderr, "wrw (%s): %s\n", v.name.c_str(), p_err()); return 2; }
                HIPCHECK(hipMemcpy(hdw.data(), dw, nw * 4, hipMemcpyDeviceToHost));
                if (first) hdw0 = hdw;
                for (size_t i = 0; i < nw; ++i) wdiff = std::max(wdiff, (double)std::fabs(hdw[i] - hdw0[i]));
            }
            const double gf = 2.0 * s.B * s.H * s.W * (double)s.C * s.C * 9 * 1e-9;
            printf("{\"shape\": [%d, %d, %d, %d], \"variant\": \"%s\", \"fwd_st_us\": %.1f, \"fwd_us\": %.1f, \"tflops_fwd_st\": %.1f, "
                   "\"max_abs_diff_vs_first\": %.3g, \"max_abs_out\": %.3g",
                   s.B, s.C, s.H, s.W, v.name.c_str(), us_st, us_pl, gf / us_st * 1e-3, maxdiff, maxabs);
            if (wrw) printf(", \"wrw_us\": %.1f, \"wrw_max_abs_diff_vs_first\": %.3g", us_wrw, wdiff);
            if (pair) printf(", \"pair_us\": %.1f", us_pair);
            if (c1) printf(", \"c1_us\": %.1f, \"c1_max_abs_diff_vs_first\": %.3g", us_c1, c1diff);
            printf("}\n");
            fflush(stdout);
            first = false;
        }
        HIPCHECK(hipFree(x)); HIPCHECK(hipFree(y)); HIPCHECK(hipFree(dy)); HIPCHECK(hipFree(w)); HIPCHECK(hipFree(stats));
        HIPCHECK(hipFree(rec)); HIPCHECK(hipFree(wp));
        if (wsb) HIPCHECK(hipFree(wsb));
        if (dw) HIPCHECK(hipFree(dw));
    }
    return 0;
}
