// Host-side native helper: performs, in ONE call, the sequence of `torch.randperm(n)` draws that the reference makes
// one Python call at a time while mining anchors (lib/loss/loss_contrast.py:79-82) and filling the pixel queue
// (segmentor/trainer_contrastive.py:127), on PyTorch's own default CPU generator -- so the mt19937 stream, and with it
// every mined index, stays bit-identical to `torch.manual_seed(s); torch.randperm(n) ...`, without ~300 Python->ATen
// round trips per step. Same algorithm as ATen's randperm_cpu (aten/src/ATen/native/TensorFactories.cpp): forward
// Fisher-Yates, z = generator->random() % (n - i), n-1 draws (none for n <= 1).
// C-ABI (ctypes), 1 = ok / 0 = error like the device library.
#include <ATen/CPUGeneratorImpl.h>
#include <ATen/core/Generator.h>

#include <cstdint>
#include <mutex>
#include <vector>

extern "C" int cseg_host_randperm_prefixes(const int64_t* n_list, const int64_t* keep, int64_t n_calls, int64_t* out) {
    try {
        auto gen = at::get_generator_or_default<at::CPUGeneratorImpl>(c10::nullopt, at::detail::getDefaultCPUGenerator());
        std::lock_guard<std::mutex> lock(gen->mutex_);
        std::vector<int64_t> r;
        int64_t o = 0;
        for (int64_t c = 0; c < n_calls; ++c) {
            const int64_t n = n_list[c], k = keep[c];
            if (n < 0 || k < 0 || k > n) return 0;
            r.resize((size_t)n);
            for (int64_t i = 0; i < n; ++i) r[(size_t)i] = i;
            for (int64_t i = 0; i < n - 1; ++i) {
                const int64_t z = (int64_t)(gen->random() % (uint64_t)(n - i));
                const int64_t t = r[(size_t)i];
                r[(size_t)i] = r[(size_t)(z + i)];
                r[(size_t)(z + i)] = t;
            }
            for (int64_t i = 0; i < k; ++i) out[o++] = r[(size_t)i];
        }
        return 1;
    } catch (...) {
        return 0;
    }
}

extern "C" int cseg_host_abi_version(void) { return 1; }
