// Torch-free timing probe of the GROUPED launches (include/cseg_hip.h, round 6) against the one-layer launches they replace:
//   g++ -O2 -std=c++17 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude tools/probes/group_probe.cpp -o tools/probes/group_probe \
//       -L/opt/rocm/lib -lamdhip64 -ldl
//   tools/probes/group_probe [--batch B] [--branches N] [--iters N] [--variant 'name:KEY=VAL;KEY=VAL']...
// Members = the first N branches of HRNet-W48 at 1024 x 512 input (48 ch on 128 x 256, 96 on 64 x 128, 192 on 32 x 64, 384 on
// 16 x 32), batch B. Timed with HIP events on one stream, best of three runs of `iters` repetitions:
//   seq_default_us  one cseg_conv3x3_split_fwd_st launch per member, library tiling (what the step ran before round 6)
//   seq_group_nt_us the same with nt = CSEG_NT_GROUP (the tile body of the grouped kernel, one launch per member)
//   group_us        ONE cseg_conv3x3_split_group_fwd launch
// and the grouped outputs / statistics records are compared bit for bit with the nt = CSEG_NT_GROUP launches.
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

struct Variant { std::string name; std::vector<std::pair<std::string, std::string>> env; };

struct Member {
    int B, C, H, W;
    size_t n, nw, T;
    float *x, *w, *y, *y_ref, *stats, *stats_ref;
    unsigned* rec;
    void *wp_def, *wp_grp;
};

int main(int argc, char** argv) {
    int batch = 8, branches = 4, iters = 20;
    std::vector<Variant> variants;
    std::vector<std::string> keys;
    for (int i = 1; i < argc; ++i) {
        std::string a = argv[i];
        if (a == "--batch" && i + 1 < argc) batch = atoi(argv[++i]);
        else if (a == "--branches" && i + 1 < argc) branches = atoi(argv[++i]);
        else if (a == "--iters" && i + 1 < argc) iters = atoi(argv[++i]);
        else if (a == "--variant" && i + 1 < argc) {
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
        }
    }
    if (variants.empty()) variants.push_back({"default", {}});
    const char* libpath = getenv("CSEG_LIB") ? getenv("CSEG_LIB") : "contrastiveseg_amd/libcseg_hip.so";
    g_lib = dlopen(libpath, RTLD_NOW | RTLD_GLOBAL);
    if (!g_lib) { fprintf(stderr, "dlopen %s: %s\n", libpath, dlerror()); return 2; }
    auto p_amax = sym<decltype(&cseg_amax_f32)>("cseg_amax_f32");
    auto p_bytes = sym<decltype(&cseg_conv3x3_split_packed_bytes)>("cseg_conv3x3_split_packed_bytes");
    auto p_pack = sym<decltype(&cseg_conv3x3_split_pack)>("cseg_conv3x3_split_pack");
    auto p_fwd_st = sym<decltype(&cseg_conv3x3_split_fwd_st)>("cseg_conv3x3_split_fwd_st");
    auto p_seg = sym<decltype(&cseg_conv_stat_segments)>("cseg_conv_stat_segments");
    auto p_group = sym<decltype(&cseg_conv3x3_split_group_fwd)>("cseg_conv3x3_split_group_fwd");
    auto p_err = sym<decltype(&cseg_last_error)>("cseg_last_error");

    HIPCHECK(hipSetDevice(0));
    hipStream_t st;
    HIPCHECK(hipStreamCreate(&st));
    int* sched;
    HIPCHECK(hipMalloc(&sched, CSEG_GROUP_SCHED_INTS * 4));
    HIPCHECK(hipMemset(sched, 0, CSEG_GROUP_SCHED_INTS * 4));

    std::vector<Member> ms;
    for (int k = 0; k < branches; ++k) {
        Member m;
        m.B = batch; m.C = 48 << k; m.H = 128 >> k; m.W = 256 >> k;
        m.n = (size_t)m.B * m.C * m.H * m.W; m.nw = (size_t)m.C * m.C * 9;
        m.T = p_seg(0, m.B, m.H, m.W);
        std::vector<float> hx(m.n), hw(m.nw);
        for (auto& v : hx) v = std::max(nrand(), 0.f);
        const float ws = 1.f / (3.f * std::sqrt((float)m.C));
        for (auto& v : hw) v = nrand() * ws;
        HIPCHECK(hipMalloc(&m.x, m.n * 4)); HIPCHECK(hipMalloc(&m.y, m.n * 4)); HIPCHECK(hipMalloc(&m.y_ref, m.n * 4));
        HIPCHECK(hipMalloc(&m.w, m.nw * 4));
        HIPCHECK(hipMalloc(&m.stats, (size_t)m.C * m.T * 16)); HIPCHECK(hipMalloc(&m.stats_ref, (size_t)m.C * m.T * 16));
        HIPCHECK(hipMalloc(&m.rec, 2 * CSEG_AMAX_WORDS * 4));
        HIPCHECK(hipMalloc(&m.wp_def, p_bytes(CSEG_ARITH_F16X3, m.C, m.C))); HIPCHECK(hipMalloc(&m.wp_grp, p_bytes(CSEG_ARITH_F16X3, m.C, m.C)));
        HIPCHECK(hipMemcpy(m.x, hx.data(), m.n * 4, hipMemcpyHostToDevice));
        HIPCHECK(hipMemcpy(m.w, hw.data(), m.nw * 4, hipMemcpyHostToDevice));
        HIPCHECK(hipMemsetAsync(m.rec, 0, 2 * CSEG_AMAX_WORDS * 4, st));
        if (!p_amax(m.x, (long)m.n, m.rec, st) || !p_amax(m.w, (long)m.nw, m.rec + CSEG_AMAX_WORDS, st)) { fprintf(stderr, "amax: %s\n", p_err()); return 2; }
        ms.push_back(m);
    }
    const int nt_def = 0;
    for (const Variant& v : variants) {
        for (const auto& k : keys) unsetenv(k.c_str());
        for (const auto& kv : v.env) setenv(kv.first.c_str(), kv.second.c_str(), 1);
        bool ok = true;
        for (Member& m : ms) {
            // 192 / 384 channels: the step asks for three tiles per block explicitly (kernels.conv3x3_sb_pick_nt)
            const int ntd = (m.C == 192 || m.C == 384) ? 3 : nt_def;
            ok = ok && p_pack(m.w, m.C, m.C, 0, ntd, CSEG_ARITH_F16X3, m.rec + CSEG_AMAX_WORDS, m.wp_def, st);
            ok = ok && p_pack(m.w, m.C, m.C, 0, CSEG_NT_GROUP, CSEG_ARITH_F16X3, m.rec + CSEG_AMAX_WORDS, m.wp_grp, st);
        }
        if (!ok) { fprintf(stderr, "pack: %s\n", p_err()); return 2; }
        std::vector<double> per_def, per_grp;
        for (Member& m : ms) {
            const int ntd = (m.C == 192 || m.C == 384) ? 3 : nt_def;
            per_def.push_back(time_us([&]() { ok = ok && p_fwd_st(m.x, m.wp_def, nullptr, m.B, m.C, m.C, m.H, m.W, ntd, CSEG_ARITH_F16X3, m.rec, m.rec + CSEG_AMAX_WORDS, m.y_ref, m.stats_ref, st); }, iters, st));
            per_grp.push_back(time_us([&]() { ok = ok && p_fwd_st(m.x, m.wp_grp, nullptr, m.B, m.C, m.C, m.H, m.W, CSEG_NT_GROUP, CSEG_ARITH_F16X3, m.rec, m.rec + CSEG_AMAX_WORDS, m.y_ref, m.stats_ref, st); }, iters, st));
        }
        auto seq = [&](bool grp) {
            for (Member& m : ms) {
                const int ntd = grp ? CSEG_NT_GROUP : ((m.C == 192 || m.C == 384) ? 3 : nt_def);
                ok = ok && p_fwd_st(m.x, grp ? m.wp_grp : m.wp_def, nullptr, m.B, m.C, m.C, m.H, m.W, ntd, CSEG_ARITH_F16X3, m.rec, m.rec + CSEG_AMAX_WORDS,
                                    m.y_ref, m.stats_ref, st);
            }
        };
        const double us_seq_def = time_us([&]() { seq(false); }, iters, st);
        const double us_seq_grp = time_us([&]() { seq(true); }, iters, st);       // leaves the reference outputs of the group tiling
        if (!ok) { fprintf(stderr, "one-layer launches (%s): %s\n", v.name.c_str(), p_err()); return 2; }
        std::vector<cseg_conv_group_member> gm(ms.size());
        for (size_t i = 0; i < ms.size(); ++i) {
            Member& m = ms[i];
            memset(&gm[i], 0, sizeof gm[i]);
            gm[i].x = m.x; gm[i].wp = m.wp_grp; gm[i].y = m.y; gm[i].stats = m.stats; gm[i].amax_x = m.rec; gm[i].amax_w = m.rec + CSEG_AMAX_WORDS;
            gm[i].B = m.B; gm[i].Cin = m.C; gm[i].Cout = m.C; gm[i].H = m.H; gm[i].W = m.W;
            HIPCHECK(hipMemsetAsync(m.y, 0xFF, m.n * 4, st));
            HIPCHECK(hipMemsetAsync(m.stats, 0xFF, (size_t)m.C * m.T * 16, st));
        }
        const double us_group = time_us([&]() { ok = ok && p_group(gm.data(), (int)gm.size(), CSEG_ARITH_F16X3, sched, st); }, iters, st);
        if (!ok) { fprintf(stderr, "group (%s): %s\n", v.name.c_str(), p_err()); return 2; }
        HIPCHECK(hipStreamSynchronize(st));
        size_t bad_y = 0, bad_st = 0;
        for (Member& m : ms) {
            std::vector<uint32_t> a(m.n), b(m.n);
            HIPCHECK(hipMemcpy(a.data(), m.y, m.n * 4, hipMemcpyDeviceToHost));
            HIPCHECK(hipMemcpy(b.data(), m.y_ref, m.n * 4, hipMemcpyDeviceToHost));
            for (size_t i = 0; i < m.n; ++i) bad_y += a[i] != b[i];
            const size_t ns = (size_t)m.C * m.T * 4;
            std::vector<float> sa(ns), sb(ns);                 // (count, mean, M2, -): equal to rounding (each kernel sums a segment in its own fixed order)
            HIPCHECK(hipMemcpy(sa.data(), m.stats, ns * 4, hipMemcpyDeviceToHost));
            HIPCHECK(hipMemcpy(sb.data(), m.stats_ref, ns * 4, hipMemcpyDeviceToHost));
            for (size_t i = 0; i < ns; i += 4)
                bad_st += sa[i] != sb[i] || std::fabs(sa[i + 1] - sb[i + 1]) > 2e-6f * (1.f + std::fabs(sb[i + 1])) ||
                          std::fabs(sa[i + 2] - sb[i + 2]) > 1e-4f * (1e-6f + std::fabs(sb[i + 2]));
        }
        std::vector<int> sc(CSEG_GROUP_SCHED_INTS);
        HIPCHECK(hipMemcpy(sc.data(), sched, sc.size() * 4, hipMemcpyDeviceToHost));
        int sched_dirty = 0;
        for (size_t i = 0; i < 9 * 32; ++i) sched_dirty += sc[i] != 0;
        double gf = 0.0;
        for (Member& m : ms) gf += 2.0 * m.B * m.H * m.W * (double)m.C * m.C * 9 * 1e-9;
        printf("{\"batch\": %d, \"branches\": %d, \"variant\": \"%s\", \"seq_default_us\": %.1f, \"seq_group_nt_us\": %.1f, \"group_us\": %.1f, "
               "\"group_tflops\": %.1f, \"mismatched_outputs\": %zu, \"mismatched_stats\": %zu, \"sched_dirty\": %d, \"per_member_default_us\": [",
               batch, branches, v.name.c_str(), us_seq_def, us_seq_grp, us_group, gf / us_group * 1e-3, bad_y, bad_st, sched_dirty);
        for (size_t i = 0; i < per_def.size(); ++i) printf("%s%.1f", i ? ", " : "", per_def[i]);
        printf("], \"timers_x64cyc\": [");              // CSEG_GROUP_ABLATE bit 128: the kernel's cycle account (conv3x3_group.hip)
        for (int i = 0; i < 13; ++i) printf("%s%d", i ? ", " : "", sc[9 * 32 + i]);
        HIPCHECK(hipMemset(sched + 9 * 32, 0, 32 * 4));
        printf("], \"per_member_group_nt_us\": [");
        for (size_t i = 0; i < per_grp.size(); ++i) printf("%s%.1f", i ? ", " : "", per_grp[i]);
        printf("]}\n");
        fflush(stdout);
    }
    return 0;
}
