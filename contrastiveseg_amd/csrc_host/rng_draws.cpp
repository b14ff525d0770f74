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
    // torch.randperm(n) on the CPU is a Fisher-Yates pass: step i (0 <= i < n - 1) draws z = random() % (n - i) and swaps r[i] with
    // r[i + z]; position i is final after step i. Only the first k entries are wanted, so only the first k steps are carried out --
    // on a sparse image of r (positions never touched still hold their index) -- and the other n - 1 - k draws are taken from the
    // generator and dropped: the stream advances exactly as torch's does (round 6: the 64-bit modulo and the two random accesses
    // per draw were 2/3 of this call's ~1 ms per step, and the GPU waits for it between the mined counts and the loss kernels).
    try {
        auto gen = at::get_generator_or_default<at::CPUGeneratorImpl>(c10::nullopt, at::detail::getDefaultCPUGenerator());
        std::lock_guard<std::mutex> lock(gen->mutex_);
        std::vector<int64_t> pos, val;                // touched positions of r and what they hold (a handful: k <= max_views)
        int64_t o = 0;
        for (int64_t c = 0; c < n_calls; ++c) {
            const int64_t n = n_list[c], k = keep[c];
            if (n < 0 || k < 0 || k > n) return 0;
            const int64_t steps = n > 0 ? n - 1 : 0, kk = k < steps ? k : steps;
            pos.clear();
            val.clear();
            auto get = [&](int64_t p) -> int64_t {
                for (size_t q = 0; q < pos.size(); ++q)
                    if (pos[q] == p) return val[q];
                return p;
            };
            auto set = [&](int64_t p, int64_t v) {
                for (size_t q = 0; q < pos.size(); ++q)
                    if (pos[q] == p) { val[q] = v; return; }
                pos.push_back(p);
                val.push_back(v);
            };
            for (int64_t i = 0; i < kk; ++i) {
                const int64_t z = (int64_t)(gen->random() % (uint64_t)(n - i));
                const int64_t vi = get(i), vj = get(z + i);
                out[o++] = vj;                        // r[i] after the swap
                set(z + i, vi);
            }
            if (k > kk) out[o++] = get(n - 1);        // k == n: the last entry is what is left
            for (int64_t i = kk; i < steps; ++i) (void)gen->random();
        }
        return 1;
    } catch (...) {
        return 0;
    }
}

extern "C" int cseg_host_abi_version(void) { return 1; }
