// TEST INFRASTRUCTURE -- fiber scheduler behind tests/emu/hip/hip_runtime.h (see the model described there).
#include <hip/hip_runtime.h>
#if !defined(__x86_64__)
#include <ucontext.h>
#endif

#include <atomic>
#include <memory>
#include <mutex>
#include <string>
#include <chrono>
#include <thread>
#include <vector>

namespace emu {

thread_local emu_uint3 t_threadIdx, t_blockIdx;
thread_local dim3 t_blockDim, t_gridDim;

namespace {
constexpr size_t STACK_BYTES = 192 * 1024;

// Context switch between fibers. x86-64: save / restore the callee-saved registers by hand (tens of nanoseconds; every
// emulated MFMA costs 128 switches per wave, and swapcontext() spends most of its time in a signal-mask system call);
// elsewhere: ucontext.
#if defined(__x86_64__)
extern "C" void emu_switch(void** save_sp, void* load_sp);
asm(R"(
    .text
    .globl emu_switch
    .type emu_switch,@function
emu_switch:
    pushq %rbp
    pushq %rbx
    pushq %r12
    pushq %r13
    pushq %r14
    pushq %r15
    movq %rsp, (%rdi)
    movq %rsi, %rsp
    popq %r15
    popq %r14
    popq %r13
    popq %r12
    popq %rbx
    popq %rbp
    ret
    .size emu_switch, .-emu_switch
)");
struct Context { void* sp = nullptr; };
inline void ctx_switch(Context& from, Context& to) { emu_switch(&from.sp, to.sp); }
inline void ctx_make(Context& c, char* stack, size_t bytes, void (*entry)()) {
    uintptr_t top = (reinterpret_cast<uintptr_t>(stack) + bytes) & ~uintptr_t(15);
    void** sp = reinterpret_cast<void**>(top - 8);          // as if `entry` had been called: rsp = 8 (mod 16) at its first instruction
    *--sp = reinterpret_cast<void*>(entry);                 // popped by `ret`
    for (int i = 0; i < 6; ++i) *--sp = nullptr;            // rbp rbx r12 r13 r14 r15
    c.sp = sp;
}
#else
struct Context { ucontext_t uc; };
inline void ctx_switch(Context& from, Context& to) { swapcontext(&from.uc, &to.uc); }
inline void ctx_make(Context& c, char* stack, size_t bytes, void (*entry)()) {
    getcontext(&c.uc);
    c.uc.uc_stack.ss_sp = stack;
    c.uc.uc_stack.ss_size = bytes;
    c.uc.uc_link = nullptr;
    makecontext(&c.uc, entry, 0);
}
#endif

struct Fiber {
    Context ctx;
    bool started = false, done = false;
    bool at_bar = false;                        // waiting at the block barrier
};
struct WaveState {
    alignas(16) unsigned char area[WAVE * XSTRIDE];
    int live = 0, arrived = 0, rel_arrived = 0;
    unsigned gen = 0, rel_gen = 0;
};
struct BlockState {
    std::vector<Fiber> fibers;
    std::vector<WaveState> waves;
    std::unique_ptr<char[]> stacks;             // not zero-filled: pages are touched only as deep as the fibers go
    Context sched;
    const std::function<void()>* body = nullptr;
    int n_threads = 0, cur = -1, live = 0, bar_arrived = 0;
    unsigned bar_gen = 0;
    unsigned long events = 0;
    dim3 block;
    emu_uint3 bidx;
    std::vector<unsigned char> lds;
    std::string error;
};
thread_local BlockState* B = nullptr;
std::atomic<int> g_last_error{hipSuccess};

void yield_to_scheduler() {
    Fiber& f = B->fibers[B->cur];
    ctx_switch(f.ctx, B->sched);
}

void release_block_barrier() {
    B->bar_arrived = 0;
    ++B->bar_gen;
    for (Fiber& f : B->fibers) f.at_bar = false;
    ++B->events;
}

void fiber_main() {
    BlockState* b = B;
    (*b->body)();
    Fiber& f = b->fibers[b->cur];
    f.done = true;
    --b->live;
    WaveState& w = b->waves[b->cur / WAVE];
    --w.live;
    ++b->events;
    // an exited thread no longer takes part in barriers (as on the hardware): complete what was waiting for it
    if (b->live > 0 && b->bar_arrived == b->live) release_block_barrier();
    if (w.live > 0 && w.arrived == w.live) { w.arrived = 0; ++w.gen; }
    if (w.live > 0 && w.rel_arrived == w.live) { w.rel_arrived = 0; ++w.rel_gen; }
    ctx_switch(f.ctx, b->sched);
    abort();                                    // a finished fiber is never resumed
}

void resume(int t) {
    Fiber& f = B->fibers[t];
    B->cur = t;
    const dim3& bd = B->block;
    t_threadIdx = emu_uint3{(unsigned)t % bd.x, ((unsigned)t / bd.x) % bd.y, (unsigned)t / (bd.x * bd.y)};
    t_blockIdx = B->bidx;
    if (!f.started) {
        f.started = true;
        ++B->events;
        ctx_make(f.ctx, B->stacks.get() + (size_t)t * STACK_BYTES, STACK_BYTES, fiber_main);
    }
    ctx_switch(B->sched, f.ctx);
    B->cur = -1;
}

std::vector<int> wave_order(int n_waves, unsigned long pass) {
    std::vector<int> o(n_waves);
    for (int i = 0; i < n_waves; ++i) o[i] = i;
    const char* e = getenv("CSEG_EMU_WAVE_ORDER");
    if (!e || !strncmp(e, "asc", 3)) return o;
    if (!strncmp(e, "desc", 4)) {
        std::reverse(o.begin(), o.end());
        return o;
    }
    unsigned long s = strtoul(strchr(e, ':') ? strchr(e, ':') + 1 : "1", nullptr, 10) * 2654435761UL + pass * 40503UL + 12345UL;
    for (int i = n_waves - 1; i > 0; --i) {                 // seeded Fisher-Yates, a new permutation every pass
        s = s * 6364136223846793005UL + 1442695040888963407UL;
        std::swap(o[i], o[(s >> 33) % (unsigned long)(i + 1)]);
    }
    return o;
}

// runs one block to completion on the calling OS thread
bool run_block(BlockState& bs) {
    B = &bs;
    unsigned long pass = 0;
    while (bs.live > 0) {
        const unsigned long before_pass = bs.events;
        for (int w : wave_order((int)bs.waves.size(), pass++)) {
            for (;;) {                                       // run this wave until all its lanes sit at the block barrier
                const unsigned long before = bs.events;
                bool ran = false;
                for (int l = 0; l < WAVE; ++l) {
                    const int t = w * WAVE + l;
                    if (t >= bs.n_threads) break;
                    Fiber& f = bs.fibers[t];
                    if (f.done || f.at_bar) continue;
                    resume(t);
                    ran = true;
                }
                if (!ran || bs.events == before) break;
            }
        }
        if (bs.events == before_pass) {
            int at_bar = 0, waiting = 0;
            for (Fiber& f : bs.fibers) { at_bar += f.at_bar && !f.done; waiting += !f.done; }
            char msg[256];
            snprintf(msg, sizeof msg, "emu: deadlock in block (%u,%u,%u): %d live threads, %d at the block barrier -- a barrier or a "
                     "wave-level operation is reached by only part of its threads", bs.bidx.x, bs.bidx.y, bs.bidx.z, waiting, at_bar);
            bs.error = msg;
            B = nullptr;
            return false;
        }
    }
    B = nullptr;
    return true;
}

}  // namespace

unsigned char* dyn_lds() { return B->lds.data(); }
int lane_id() { return B->cur % WAVE; }

void block_barrier() {
    Fiber& f = B->fibers[B->cur];
    ++B->bar_arrived;
    if (B->bar_arrived == B->live) {
        release_block_barrier();
        return;
    }
    const unsigned my = B->bar_gen;
    f.at_bar = true;
    while (B->bar_gen == my) yield_to_scheduler();
}

const unsigned char* wave_exchange(const void* mine, int bytes) {
    WaveState& w = B->waves[B->cur / WAVE];
    if (bytes > XSTRIDE) { fprintf(stderr, "emu: exchange payload too large\n"); abort(); }
    memcpy(w.area + (size_t)lane_id() * XSTRIDE, mine, bytes);
    ++w.arrived;
    if (w.arrived == w.live) {
        w.arrived = 0;
        ++w.gen;
        ++B->events;
    } else {
        const unsigned my = w.gen;
        while (w.gen == my) yield_to_scheduler();
    }
    return w.area;
}

void wave_release() {
    WaveState& w = B->waves[B->cur / WAVE];
    ++w.rel_arrived;
    if (w.rel_arrived == w.live) {
        w.rel_arrived = 0;
        ++w.rel_gen;
        ++B->events;
        return;
    }
    const unsigned my = w.rel_gen;
    while (w.rel_gen == my) yield_to_scheduler();
}

hipError_t last_error() { return g_last_error.exchange(hipSuccess); }

thread_local bool t_resident_next = false;
void request_resident() { t_resident_next = true; }
void spin_pause() { std::this_thread::sleep_for(std::chrono::microseconds(50)); }

hipError_t launch(dim3 grid, dim3 block, size_t lds_bytes, const std::function<void()>& body) {
    const int n_threads = (int)(block.x * block.y * block.z);
    const long n_blocks = (long)grid.x * grid.y * grid.z;
    if (lds_bytes > (size_t)LDS_BYTES || n_threads <= 0 || n_threads > 1024 || n_blocks <= 0) {
        g_last_error = hipErrorLaunchFailure;
        return hipErrorLaunchFailure;
    }
    int n_os = (int)std::min<long>(n_blocks, std::max(1u, std::thread::hardware_concurrency()));
    if (const char* e = getenv("CSEG_EMU_THREADS")) n_os = std::max(1, std::min(n_os, atoi(e)));
    if (t_resident_next) {                      // blocks of this launch wait for each other: all of them must be running
        t_resident_next = false;
        if (n_blocks > 1024) {
            fprintf(stderr, "emu: a grid-resident launch of %ld blocks (the hardware holds a few hundred)\n", n_blocks);
            g_last_error = hipErrorLaunchFailure;
            return hipErrorLaunchFailure;
        }
        n_os = (int)n_blocks;
    }
    std::atomic<long> next{0};
    std::atomic<bool> failed{false};
    std::string first_error;
    std::mutex mu;
    auto worker = [&]() {
        BlockState bs;
        bs.stacks.reset(new char[(size_t)n_threads * STACK_BYTES]);
        bs.lds.resize(LDS_BYTES + 64);
        for (;;) {
            const long b = next.fetch_add(1);
            if (b >= n_blocks || failed.load()) break;
            bs.fibers.assign(n_threads, Fiber());
            bs.waves.assign((n_threads + WAVE - 1) / WAVE, WaveState());
            for (int t = 0; t < n_threads; ++t) ++bs.waves[t / WAVE].live;
            bs.n_threads = bs.live = n_threads;
            bs.bar_arrived = 0;
            bs.body = &body;
            bs.block = block;
            bs.bidx = emu_uint3{(unsigned)(b % grid.x), (unsigned)((b / grid.x) % grid.y), (unsigned)(b / ((long)grid.x * grid.y))};
            t_blockDim = block;
            t_gridDim = grid;
            memset(bs.lds.data(), 0xFF, bs.lds.size());           // unwritten LDS reads back as NaN patterns
            if (!run_block(bs)) {
                std::lock_guard<std::mutex> g(mu);
                if (!failed.exchange(true)) first_error = bs.error;
            }
        }
    };
    std::vector<std::thread> pool;
    for (int i = 0; i < n_os; ++i) pool.emplace_back(worker);
    for (auto& t : pool) t.join();
    if (failed.load()) {
        fprintf(stderr, "%s\n", first_error.c_str());
        g_last_error = hipErrorLaunchFailure;
        return hipErrorLaunchFailure;
    }
    return hipSuccess;
}

}  // namespace emu
