// Native executor of a residual block (round 4, opt-in: CSEG_NATIVE_BLOCK=1).
// Reference shape of the work: BasicBlock.forward of lib/models/backbones/hrnet/hrnet_backbone.py:49-65 (conv3x3 -> bn -> relu ->
// conv3x3 -> bn -> += x -> relu) and its backward; 104 such blocks per step of HRNet-W48.
// Why: the step is within 10 % of being bound by the ONE host thread that feeds four HIP queues (79 ms of enqueue time against 86 ms
// of GPU time at batch 8, tools/host_profile.py), and tools/host_null_bench.py (stub kernels) puts ~200 us of Python per block forward
// and ~300 us per block backward on this path -- six resp. ten launches each, every one wrapped in allocation, argument checking
// and ctypes conversion. Here one call from Python runs the same C-ABI entry points of libcseg_hip.so (include/cseg_hip.h) in the
// same order with the same arguments as kernels.BasicBlockSplit: bit-identical results, one autograd node as before (the node
// itself stays in Python: kernels.BasicBlockNative).
// The entry points are handed over as ADDRESSES (bind()), taken from the ctypes handle the package already holds -- no second
// copy of the library, and the tests point the executor at the emulated library the same way (tests/emu/inject.py).
// Built by csrc_host/build.py into contrastiveseg_amd/_cseg_native.so (g++ against libtorch + pybind11).
#include <torch/extension.h>

#include <cstdint>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>

namespace {

typedef void* P;
// signatures of include/cseg_hip.h (the subset this file calls)
typedef int (*fwd_st_t)(const float*, const void*, const float*, int, int, int, int, int, int, int, const unsigned*, const unsigned*,
                        float*, float*, P);
typedef int (*fwd_t)(const float*, const void*, const float*, int, int, int, int, int, int, int, const unsigned*, const unsigned*,
                     float*, P);
typedef int (*fwd_add_t)(const float*, const void*, const float*, const float*, int, int, int, int, int, int, int, const unsigned*,
                         const unsigned*, float*, P);
typedef int (*wrw_t)(const float*, const float*, int, int, int, int, int, int, const unsigned*, const unsigned*, float*, float*, P);
typedef size_t (*wrw_ws_t)(int, int, int, int, int);
typedef size_t (*seg_t)(int, int, int, int);
typedef int (*tiles_fin_t)(const float*, int, long, float, float, float*, float*, long*, float*, P);
typedef int (*bn_apply_t)(const float*, const float*, const float*, const float*, const float*, int, int, int, int, float*, unsigned*,
                          P);
typedef int (*bn_bwd_t)(const float*, const float*, const float*, const float*, const float*, const float*, int, int, int, int, int,
                        float*, float*, float*, float*, float*, unsigned*, P);
typedef const char* (*err_t)(void);

struct Api {
    fwd_st_t fwd_st = nullptr;
    fwd_t fwd = nullptr;
    fwd_add_t fwd_add = nullptr;
    wrw_t wrw = nullptr;
    wrw_ws_t wrw_ws = nullptr;
    seg_t seg = nullptr;
    tiles_fin_t tiles_fin = nullptr;
    bn_apply_t bn_apply = nullptr;
    bn_bwd_t bn_bwd = nullptr;
    err_t err = nullptr;
} api;

void bind(const std::unordered_map<std::string, uint64_t>& t) {
    auto get = [&](const char* name) -> void* {
        auto it = t.find(name);
        if (it == t.end() || it->second == 0) throw std::runtime_error(std::string("block executor: no address for ") + name);
        return reinterpret_cast<void*>(it->second);
    };
    api.fwd_st = reinterpret_cast<fwd_st_t>(get("cseg_conv3x3_split_fwd_st"));
    api.fwd = reinterpret_cast<fwd_t>(get("cseg_conv3x3_split_fwd"));
    api.fwd_add = reinterpret_cast<fwd_add_t>(get("cseg_conv3x3_split_fwd_add"));
    api.wrw = reinterpret_cast<wrw_t>(get("cseg_conv3x3_split_wrw"));
    api.wrw_ws = reinterpret_cast<wrw_ws_t>(get("cseg_conv3x3_sb_wrw_ws_floats"));
    api.seg = reinterpret_cast<seg_t>(get("cseg_conv_stat_segments"));
    api.tiles_fin = reinterpret_cast<tiles_fin_t>(get("cseg_bn_tiles_finalize"));
    api.bn_apply = reinterpret_cast<bn_apply_t>(get("cseg_bn_apply_amax"));
    api.bn_bwd = reinterpret_cast<bn_bwd_t>(get("cseg_bn_bwd_amax"));
    api.err = reinterpret_cast<err_t>(get("cseg_last_error"));
}

void check(int ok, const char* what) {
    if (ok != 1) throw std::runtime_error(std::string(what) + ": " + (api.err ? api.err() : "failed"));
}

const float* fp(const at::Tensor& t) { return t.data_ptr<float>(); }
float* fpm(at::Tensor& t) { return t.data_ptr<float>(); }
const float* fpo(const c10::optional<at::Tensor>& t) { return t.has_value() ? t->data_ptr<float>() : nullptr; }
float* fpmo(const c10::optional<at::Tensor>& t) { return t.has_value() ? t->data_ptr<float>() : nullptr; }
long* lpo(const c10::optional<at::Tensor>& t) { return t.has_value() ? reinterpret_cast<long*>(t->data_ptr<int64_t>()) : nullptr; }
const unsigned* up(uint64_t a) { return reinterpret_cast<const unsigned*>(a); }
unsigned* upm(uint64_t a) { return reinterpret_cast<unsigned*>(a); }

void require_f32c(const at::Tensor& t, const char* what) {
    if (t.scalar_type() != at::kFloat || !t.is_contiguous())
        throw std::runtime_error(std::string("block executor: ") + what + " must be a contiguous fp32 tensor");
}

// One BatchNorm of the block: its parameters, running statistics and constants
struct Bn {
    c10::optional<at::Tensor> weight, bias, running_mean, running_var, num_batches_tracked;
    double eps, momentum;
};

// forward: conv1 (+ statistics) -> finalize -> bn1 + relu (+ max|a1| into am1) -> conv2 (+ statistics) -> finalize -> bn2 + x + relu
// (+ max|out| into am2). wp* / aw*: packed weights and max|w| records (addresses, kept alive by kernels.SPLIT_WEIGHTS); ax: max|x|.
std::vector<at::Tensor> block_forward(const at::Tensor& x, uint64_t wp1, uint64_t aw1, uint64_t wp2, uint64_t aw2, uint64_t ax,
                                      const Bn& bn1, const Bn& bn2, uint64_t am1, uint64_t am2, int nt, int arith, uint64_t stream) {
    require_f32c(x, "x");
    const int B = (int)x.size(0), C = (int)x.size(1), H = (int)x.size(2), W = (int)x.size(3), HW = H * W;
    const long T = (long)api.seg(0, B, H, W);
    P st = reinterpret_cast<P>(stream);
    const auto opt = x.options();
    at::Tensor c1 = at::empty_like(x), s1 = at::empty({C, T, 4}, opt), mi1 = at::empty({C, 2}, opt), a1 = at::empty_like(x);
    check(api.fwd_st(fp(x), reinterpret_cast<const void*>(wp1), nullptr, B, C, C, H, W, nt, arith, up(ax), up(aw1), fpm(c1), fpm(s1), st),
          "cseg_conv3x3_split_fwd_st");
    check(api.tiles_fin(fp(s1), C, T, (float)bn1.eps, (float)bn1.momentum, fpmo(bn1.running_mean), fpmo(bn1.running_var),
                        lpo(bn1.num_batches_tracked), fpm(mi1), st),
          "cseg_bn_tiles_finalize");
    check(api.bn_apply(fp(c1), nullptr, fp(mi1), fpo(bn1.weight), fpo(bn1.bias), 1, B, C, HW, fpm(a1), upm(am1), st), "cseg_bn_apply_amax");
    at::Tensor c2 = at::empty_like(x), s2 = at::empty({C, T, 4}, opt), mi2 = at::empty({C, 2}, opt), out = at::empty_like(x);
    check(api.fwd_st(fp(a1), reinterpret_cast<const void*>(wp2), nullptr, B, C, C, H, W, nt, arith, up(am1), up(aw2), fpm(c2), fpm(s2), st),
          "cseg_conv3x3_split_fwd_st");
    check(api.tiles_fin(fp(s2), C, T, (float)bn2.eps, (float)bn2.momentum, fpmo(bn2.running_mean), fpmo(bn2.running_var),
                        lpo(bn2.num_batches_tracked), fpm(mi2), st),
          "cseg_bn_tiles_finalize");
    check(api.bn_apply(fp(c2), fp(x), fp(mi2), fpo(bn2.weight), fpo(bn2.bias), 1, B, C, HW, fpm(out), upm(am2), st), "cseg_bn_apply_amax");
    return {out, c1, a1, c2, mi1, mi2};
}

// backward, in the order of kernels.BasicBlockSplit.backward. wp*t / aw*: the backward-data (transposed, flipped) packs.
// ax / am1: the forward's max|x| / max|a1| records; am / amb: fresh zeroed records for max|dc2| / max|dc1|; ws: the BN reduction
// scratch of this (device, stream). -> {dx, dw1, dg1, db1, dw2, dg2, db2} (undefined tensors where not needed).
std::vector<at::Tensor> block_backward(const at::Tensor& dy, const at::Tensor& x, const at::Tensor& c1, const at::Tensor& a1,
                                       const at::Tensor& c2, const at::Tensor& out, const at::Tensor& mi1, const at::Tensor& mi2,
                                       uint64_t wp1t, uint64_t aw1, uint64_t wp2t, uint64_t aw2, uint64_t ax, uint64_t am1, uint64_t am,
                                       uint64_t amb, const Bn& bn1, const Bn& bn2, at::Tensor ws, int nt, int arith, bool need_dx,
                                       bool need_dw1, bool need_dw2, uint64_t stream) {
    require_f32c(dy, "dy");
    const int B = (int)x.size(0), C = (int)x.size(1), H = (int)x.size(2), W = (int)x.size(3), HW = H * W;
    P st = reinterpret_cast<P>(stream);
    const auto opt = x.options();
    // bn2 + add + ReLU: mask from `out`; the masked gradient g is also the identity path's gradient
    at::Tensor dwb2 = at::empty({2, C}, opt), g = at::empty_like(x), dc2 = at::empty_like(x);
    check(api.bn_bwd(fp(dy), fp(c2), fp(out), fp(mi2), fpo(bn2.weight), fpo(bn2.bias), 2, 1, B, C, HW, fpm(ws), fpm(g), fpm(dwb2),
                     fpm(dwb2) + C, fpm(dc2), upm(am), st),
          "cseg_bn_bwd_amax");
    at::Tensor da1 = at::empty_like(x);
    check(api.fwd(fp(dc2), reinterpret_cast<const void*>(wp2t), nullptr, B, C, C, H, W, nt, arith, up(am), up(aw2), fpm(da1), st),
          "cseg_conv3x3_split_fwd");
    at::Tensor dw2, dw1;
    const size_t n_ws = (need_dw1 || need_dw2) ? api.wrw_ws(B, C, C, H, W) : 0;
    if ((need_dw1 || need_dw2) && n_ws == 0) throw std::runtime_error("block executor: the weight gradient does not take this shape");
    if (need_dw2) {
        at::Tensor w2s = at::empty({(long)n_ws}, opt);
        dw2 = at::empty({C, C, 3, 3}, opt);
        check(api.wrw(fp(a1), fp(dc2), B, C, C, H, W, arith, up(am1), up(am), fpm(w2s), fpm(dw2), st), "cseg_conv3x3_split_wrw");
    }
    // bn1 + ReLU: mask recomputed from c1
    at::Tensor dwb1 = at::empty({2, C}, opt), dc1 = at::empty_like(x);
    check(api.bn_bwd(fp(da1), fp(c1), nullptr, fp(mi1), fpo(bn1.weight), fpo(bn1.bias), 1, 1, B, C, HW, fpm(ws), nullptr, fpm(dwb1),
                     fpm(dwb1) + C, fpm(dc1), upm(amb), st),
          "cseg_bn_bwd_amax");
    at::Tensor dx;
    if (need_dx) {      // conv1: backward-data with the identity path's gradient added in the epilogue
        dx = at::empty_like(x);
        check(api.fwd_add(fp(dc1), reinterpret_cast<const void*>(wp1t), nullptr, fp(g), B, C, C, H, W, nt, arith, up(amb), up(aw1), fpm(dx),
                          st),
              "cseg_conv3x3_split_fwd_add");
    }
    if (need_dw1) {
        at::Tensor w1s = at::empty({(long)n_ws}, opt);
        dw1 = at::empty({C, C, 3, 3}, opt);
        check(api.wrw(fp(x), fp(dc1), B, C, C, H, W, arith, up(ax), up(amb), fpm(w1s), fpm(dw1), st), "cseg_conv3x3_split_wrw");
    }
    return {dx, dw1, dwb1[0], dwb1[1], dw2, dwb2[0], dwb2[1]};
}

}  // namespace

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
    m.doc() = "native executor of a residual block over the C-ABI of libcseg_hip.so";
    py::class_<Bn>(m, "Bn")
        .def(py::init([](c10::optional<at::Tensor> w, c10::optional<at::Tensor> b, c10::optional<at::Tensor> rm, c10::optional<at::Tensor> rv,
                         c10::optional<at::Tensor> nbt, double eps, double momentum) {
            return Bn{std::move(w), std::move(b), std::move(rm), std::move(rv), std::move(nbt), eps, momentum};
        }));
    m.def("bind", &bind, "addresses of the C-ABI entry points by name");
    m.def("block_forward", &block_forward);
    m.def("block_backward", &block_backward);
}
