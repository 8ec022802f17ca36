// tests/emul/wavesim.cpp — the execution model behind tests/emul/hip/hip_runtime.h (TEST INFRASTRUCTURE; see there).
#include <hip/hip_runtime.h>

#include <execinfo.h>
#include <signal.h>
#include <sys/mman.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <climits>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

thread_local emu_idx3 threadIdx, blockIdx;
thread_local dim3 blockDim, gridDim;

// ---- a minimal x86-64 context switch: callee-saved registers + stack pointer ----
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
.size emu_switch,.-emu_switch
)");

namespace emu {

namespace {

constexpr size_t STACK_BYTES = 256 * 1024;
constexpr unsigned MAX_THREADS = 1024;

enum State : int { RUNNABLE = 0, AT_WAVE = 1, AT_BARRIER = 2, DONE = 3 };
std::atomic<uint64_t> g_split_rounds{0};  // rounds in which the lanes of a wavefront were parked at more than one collective

struct Fiber {
    void* sp;
    int state;
    int op, site, arg, width;
    uint64_t value, result;
    float a, b, c[4], d[4];
    emu_idx3 tid;
    unsigned lin;
};

struct Worker {
    char* stacks = nullptr;  // MAX_THREADS stacks, mapped once per host thread
    Fiber fib[MAX_THREADS];
    uint16_t order[MAX_THREADS];
    void* sched_sp = nullptr;
    Fiber* cur = nullptr;
    body_fn fn = nullptr;
    void* ctx = nullptr;
    std::vector<unsigned char> dyn;
    char* strict_map = nullptr;  // RP_EMUL_LDS_STRICT
    size_t strict_len = 0;
    char* strict_dyn = nullptr;
    const char* kernel = "";
    ~Worker() {
        if (stacks) munmap(stacks, STACK_BYTES * MAX_THREADS);
    }
};
thread_local Worker* tl_worker = nullptr;

Worker& worker() {
    if (!tl_worker) {
        tl_worker = new Worker();
        void* p = mmap(nullptr, STACK_BYTES * MAX_THREADS, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
        if (p == MAP_FAILED) {
            fprintf(stderr, "emu: cannot map fiber stacks\n");
            abort();
        }
        tl_worker->stacks = static_cast<char*>(p);
    }
    return *tl_worker;
}

[[noreturn]] void die(const char* what) {
    Worker& w = worker();
    fprintf(stderr, "emu: %s in kernel %s, workgroup (%u,%u,%u)\n", what, w.kernel, blockIdx.x, blockIdx.y, blockIdx.z);
    abort();
}

void fiber_entry() {
    Worker& w = *tl_worker;
    w.fn(w.ctx);
    w.cur->state = DONE;
    emu_switch(&w.cur->sp, w.sched_sp);
    abort();  // a finished fiber is never resumed
}

void park(Worker& w) {  // back to the scheduler; returns when the scheduler resumes this fiber
    Fiber* f = w.cur;
    emu_switch(&f->sp, w.sched_sp);
}

void resolve_group(Worker& w, unsigned base, uint64_t members) {
    // members: lanes of the wave (bit i = fiber base + i) parked at the same (site, op)
    Fiber* f0 = &w.fib[base + (unsigned)__builtin_ctzll(members)];
    const int op = f0->op;
    if (op == OP_BALLOT) {
        uint64_t m = 0;
        for (uint64_t r = members; r; r &= r - 1) {
            const unsigned l = (unsigned)__builtin_ctzll(r);
            if (w.fib[base + l].value) m |= 1ull << l;
        }
        for (uint64_t r = members; r; r &= r - 1) w.fib[base + (unsigned)__builtin_ctzll(r)].result = m;
    } else if (op == OP_FIRST) {
        for (uint64_t r = members; r; r &= r - 1) w.fib[base + (unsigned)__builtin_ctzll(r)].result = f0->value;
    } else if (op == OP_WAVE_BARRIER) {
        // nothing to exchange
    } else if (op == OP_MFMA16) {
        if (members != ~0ull) die("mfma with inactive lanes");
        float A[16][4], B[4][16];
        for (unsigned l = 0; l < 64; ++l) {
            A[l & 15][l >> 4] = w.fib[base + l].a;
            B[l >> 4][l & 15] = w.fib[base + l].b;
        }
        for (unsigned l = 0; l < 64; ++l) {
            Fiber& f = w.fib[base + l];
            const unsigned j = l & 15;
            for (unsigned r = 0; r < 4; ++r) {
                const unsigned i = 4 * (l >> 4) + r;
                float acc = f.c[r];
                for (unsigned k = 0; k < 4; ++k) acc = fmaf(A[i][k], B[k][j], acc);
                f.d[r] = acc;
            }
        }
    } else {  // shuffles
        for (uint64_t r = members; r; r &= r - 1) {
            const unsigned l = (unsigned)__builtin_ctzll(r);
            Fiber& f = w.fib[base + l];
            const int width = f.width > 0 && f.width <= 64 ? f.width : 64;
            const int seg = (int)l & ~(width - 1), pos = (int)l & (width - 1);
            int src;
            if (op == OP_SHFL) src = seg + (f.arg & (width - 1));
            else if (op == OP_SHFL_UP) src = pos - f.arg >= 0 ? (int)l - f.arg : (int)l;
            else if (op == OP_SHFL_DOWN) src = pos + f.arg < width ? (int)l + f.arg : (int)l;
            else src = (pos ^ f.arg) < width ? seg + (pos ^ f.arg) : (int)l;
            // a source lane outside EXEC: ds_bpermute returns 0 for it
            f.result = (src >= 0 && src < 64 && ((members >> src) & 1u)) ? w.fib[base + (unsigned)src].value : 0;
        }
    }
    for (uint64_t r = members; r; r &= r - 1) w.fib[base + (unsigned)__builtin_ctzll(r)].state = RUNNABLE;
}

void run_block(Worker& w, unsigned nthreads) {
    for (unsigned t = 0; t < nthreads; ++t) {
        Fiber& f = w.fib[t];
        f.state = RUNNABLE;
        f.lin = t;
        f.tid.x = t % blockDim.x;
        f.tid.y = (t / blockDim.x) % blockDim.y;
        f.tid.z = t / (blockDim.x * blockDim.y);
        char* top = w.stacks + (size_t)(t + 1) * STACK_BYTES;
        void** sp = reinterpret_cast<void**>(top);
        *--sp = nullptr;                                     // the entry's (never used) return address slot
        *--sp = reinterpret_cast<void*>(&fiber_entry);       // popped by emu_switch's ret
        for (int r = 0; r < 6; ++r) *--sp = nullptr;         // rbp rbx r12 r13 r14 r15
        f.sp = sp;
    }
    const unsigned nwaves = (nthreads + 63) / 64;
    // The order in which parked work-items are resumed.  The hardware interleaves the wavefronts of a workgroup freely; a fixed
    // 0..n-1 order would hide a missing barrier (the producer always ran first).  RP_EMUL_ORDER=reverse | random (RP_EMUL_SEED)
    {
        static const int mode = [] {
            const char* e = getenv("RP_EMUL_ORDER");
            return !e ? 0 : (!strcmp(e, "reverse") ? 1 : (!strcmp(e, "random") ? 2 : 0));
        }();
        for (unsigned t = 0; t < nthreads; ++t) w.order[t] = (uint16_t)(mode == 1 ? nthreads - 1 - t : t);
        if (mode == 2) {
            static const uint64_t seed = getenv("RP_EMUL_SEED") ? strtoull(getenv("RP_EMUL_SEED"), nullptr, 10) : 1;
            uint64_t x = seed * 0x9e3779b97f4a7c15ull + blockIdx.x * 0xbf58476d1ce4e5b9ull + blockIdx.y * 0x94d049bb133111ebull + 1;
            for (unsigned t = nthreads; t > 1; --t) {
                x ^= x << 13;
                x ^= x >> 7;
                x ^= x << 17;
                std::swap(w.order[t - 1], w.order[x % t]);
            }
        }
    }
    for (;;) {
        bool ran = false;
        for (unsigned o = 0; o < nthreads; ++o) {
            Fiber& f = w.fib[w.order[o]];
            if (f.state != RUNNABLE) continue;
            ran = true;
            w.cur = &f;
            threadIdx = f.tid;
            emu_switch(&w.sched_sp, f.sp);
        }
        // every fiber is parked or done: settle the wavefront collectives
        bool resolved = false;
        for (unsigned wv = 0; wv < nwaves; ++wv) {
            const unsigned base = wv * 64, n = std::min(64u, nthreads - base);
            uint64_t waiting = 0;
            for (unsigned l = 0; l < n; ++l)
                if (w.fib[base + l].state == AT_WAVE) waiting |= 1ull << l;
            if (waiting) {
                // Lanes parked at DIFFERENT collectives: the hardware runs a divergent region to its end before the code after
                // it, so the lanes at the later collective wait for the others to arrive there (a lane that skipped a branch
                // waits for those inside it; a lane that left a loop waits for those still iterating).  "Later" is read off the
                // source line; only the earliest collective is settled in this round.
                int first = INT32_MAX;
                for (uint64_t r = waiting; r; r &= r - 1) first = std::min(first, w.fib[base + (unsigned)__builtin_ctzll(r)].site);
                uint64_t members = 0;
                int op = 0;
                for (uint64_t r = waiting; r; r &= r - 1) {
                    const unsigned l = (unsigned)__builtin_ctzll(r);
                    if (w.fib[base + l].site != first) continue;
                    if (!members) op = w.fib[base + l].op;
                    if (w.fib[base + l].op == op) members |= 1ull << l;
                }
                if (members != waiting) g_split_rounds.fetch_add(1, std::memory_order_relaxed);
                resolve_group(w, base, members);
                resolved = true;
            }
        }
        if (resolved) continue;
        unsigned at_barrier = 0, done = 0;
        for (unsigned t = 0; t < nthreads; ++t) {
            at_barrier += w.fib[t].state == AT_BARRIER;
            done += w.fib[t].state == DONE;
        }
        if (done == nthreads) return;
        if (at_barrier && at_barrier + done == nthreads) {
            for (unsigned t = 0; t < nthreads; ++t)
                if (w.fib[t].state == AT_BARRIER) w.fib[t].state = RUNNABLE;
            continue;
        }
        if (!ran) die("deadlock (no runnable work-item)");
    }
}

void on_fault(int sig, siginfo_t* si, void*) {
    Worker* w = tl_worker;
    char buf[512];
    int n = snprintf(buf, sizeof buf, "emu: signal %d at address %p in kernel %s, workgroup (%u,%u,%u), work-item %u\n", sig, si->si_addr,
                     w ? w->kernel : "?", blockIdx.x, blockIdx.y, blockIdx.z, w && w->cur ? w->cur->lin : 0u);
    if (write(2, buf, (size_t)n) < 0) {}
    void* bt[48];
    backtrace_symbols_fd(bt, backtrace(bt, 48), 2);
    _exit(139);
}
void trap_faults() {
    static std::once_flag once;
    std::call_once(once, [] {
        if (!getenv("RP_EMUL_TRAP")) return;
        static char alt[1 << 16];
        stack_t ss{};
        ss.ss_sp = alt;
        ss.ss_size = sizeof alt;
        sigaltstack(&ss, nullptr);
        struct sigaction sa {};
        sa.sa_sigaction = on_fault;
        sa.sa_flags = SA_SIGINFO | SA_ONSTACK;
        sigaction(SIGSEGV, &sa, nullptr);
        sigaction(SIGBUS, &sa, nullptr);
    });
}

struct Job {
    dim3 grid, block;
    size_t shmem;
    const char* name;
    body_fn fn;
    void* ctx;
    uint64_t nblocks;
    unsigned nthreads;
    std::atomic<uint64_t> next{0};
    Job(dim3 g, dim3 b, size_t s, const char* n, body_fn f, void* c, uint64_t nb, unsigned nt)
        : grid(g), block(b), shmem(s), name(n), fn(f), ctx(c), nblocks(nb), nthreads(nt) {}
    void run() {  // on any host thread: take workgroups until none is left
        Worker& w = worker();
        w.fn = fn;
        w.ctx = ctx;
        w.kernel = name;
        // the LDS aperture: a read past the workgroup's allocation returns junk on the device, it does not fault (the
        // traversal's software pipeline reads a few slots past its arrays by design) — keep a wide margin mapped
        if (w.dyn.size() < ((size_t)1 << 20) + shmem) w.dyn.assign(((size_t)1 << 20) + shmem, 0xA5);
        // RP_EMUL_LDS_STRICT=1: the dynamic LDS ends at an inaccessible page instead (a kernel that indexes past what the host asked
        // for faults; k_traverse_lds' look-ahead reads do so by design — leave it out of such a run)
        static const bool strict = getenv("RP_EMUL_LDS_STRICT") != nullptr;
        if (strict) {
            const size_t page = 4096, body = (shmem + 15) & ~(size_t)15, span = ((body + page - 1) & ~(page - 1)) + page;
            if (w.strict_len < span + page) {
                if (w.strict_map) munmap(w.strict_map, w.strict_len);
                w.strict_len = span + page;
                w.strict_map = static_cast<char*>(mmap(nullptr, w.strict_len, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0));
            } else {
                mprotect(w.strict_map, w.strict_len, PROT_READ | PROT_WRITE);
            }
            mprotect(w.strict_map + span - page, w.strict_len - (span - page), PROT_NONE);
            w.strict_dyn = w.strict_map + (span - page) - body;
        } else {
            w.strict_dyn = nullptr;
        }
        blockDim = block;
        gridDim = grid;
        for (;;) {
            const uint64_t b = next.fetch_add(1);
            if (b >= nblocks) break;
            blockIdx.x = (uint32_t)(b % grid.x);
            blockIdx.y = (uint32_t)((b / grid.x) % grid.y);
            blockIdx.z = (uint32_t)(b / ((uint64_t)grid.x * grid.y));
            run_block(w, nthreads);
        }
    }
};

// host threads that stay for the life of the process (a launch is synchronous: the caller works too and waits)
class Pool {
  public:
    void run(Job* job, unsigned helpers) {
        {
            std::unique_lock<std::mutex> lk(mu_);
            while (threads_.size() < helpers) threads_.emplace_back([this, id = (unsigned)threads_.size()] { loop(id); });
            job_ = job;
            want_ = helpers;
            busy_ = helpers;
            ++gen_;
        }
        cv_.notify_all();
        job->run();
        std::unique_lock<std::mutex> lk(mu_);
        done_.wait(lk, [this] { return busy_ == 0; });
        job_ = nullptr;
    }

  private:
    void loop(unsigned id) {
        uint64_t seen = 0;
        for (;;) {
            Job* j;
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_.wait(lk, [&] { return gen_ != seen; });
                seen = gen_;
                if (id >= want_) continue;
                j = job_;
            }
            j->run();
            std::unique_lock<std::mutex> lk(mu_);
            if (--busy_ == 0) done_.notify_all();
        }
    }
    std::mutex mu_;
    std::condition_variable cv_, done_;
    std::vector<std::thread> threads_;
    Job* job_ = nullptr;
    unsigned want_ = 0, busy_ = 0;
    uint64_t gen_ = 0;
};
Pool& pool() {
    static Pool* p = new Pool();  // never destroyed: its threads wait forever and die with the process
    return *p;
}

struct LaunchStats {
    std::mutex mu;
    uint64_t launches = 0, blocks = 0;
} g_stats;

unsigned host_threads() {
    static unsigned n = [] {
        const char* e = getenv("RP_EMUL_THREADS");
        unsigned v = e ? (unsigned)atoi(e) : std::min(8u, std::max(1u, std::thread::hardware_concurrency()));
        return std::max(1u, v);
    }();
    return n;
}

}  // namespace

uint64_t collective(int op, int site, uint64_t value, int arg, int width) {
    Worker& w = *tl_worker;
    Fiber* f = w.cur;
    f->op = op;
    f->site = site;
    f->value = value;
    f->arg = arg;
    f->width = width;
    f->state = AT_WAVE;
    park(w);
    return f->result;
}

void mfma16x16x4(int site, float a, float b, const float* c, float* d) {
    Worker& w = *tl_worker;
    Fiber* f = w.cur;
    f->op = OP_MFMA16;
    f->site = site;
    f->a = a;
    f->b = b;
    memcpy(f->c, c, 16);
    f->state = AT_WAVE;
    park(w);
    memcpy(d, f->d, 16);
}

void syncthreads() {
    Worker& w = *tl_worker;
    w.cur->state = AT_BARRIER;
    park(w);
}

unsigned lane() { return tl_worker->cur->lin & 63u; }

void* dyn_smem() { return tl_worker->strict_dyn ? static_cast<void*>(tl_worker->strict_dyn) : static_cast<void*>(tl_worker->dyn.data()); }

void launch(dim3 grid, dim3 block, size_t shmem, const char* name, body_fn fn, void* ctx) {
    const uint64_t nblocks = (uint64_t)grid.x * grid.y * grid.z;
    const unsigned nthreads = block.x * block.y * block.z;
    if (nthreads == 0 || nthreads > MAX_THREADS) {
        fprintf(stderr, "emu: kernel %s launched with %u work-items per workgroup\n", name, nthreads);
        abort();
    }
    if (shmem > 160 * 1024) {
        fprintf(stderr, "emu: kernel %s asks for %zu bytes of dynamic LDS (160 KB per CU)\n", name, shmem);
        abort();
    }
    if (nblocks == 0) return;
    trap_faults();
    {
        std::lock_guard<std::mutex> lk(g_stats.mu);
        g_stats.launches += 1;
        g_stats.blocks += nblocks;
    }
    Job job{grid, block, shmem, name, fn, ctx, nblocks, nthreads};
    const unsigned nt = (unsigned)std::min<uint64_t>(host_threads(), nblocks);
    if (nt <= 1) {
        job.run();
    } else {
        static std::mutex one_launch;  // launches from several host threads take turns
        std::lock_guard<std::mutex> lk(one_launch);
        pool().run(&job, nt - 1);
    }
}

}  // namespace emu

// ---- runtime API ----
struct ihipEvent_t {
    std::chrono::steady_clock::time_point t;
};
struct ihipStream_t {
    int id;
};

extern "C" {
// RP_EMUL_GUARD=1: every device allocation ends at an inaccessible page (and starts after one), so that a kernel reading or
// writing past its buffer faults at the access — on the device such an access usually lands in a neighbouring allocation unseen
namespace {
struct Mapping {
    void* base;
    size_t len;
};
std::mutex g_map_mu;
std::vector<std::pair<void*, Mapping>> g_maps;
int fill_byte() {  // RP_EMUL_FILL=0: fresh allocations read as zero (to tell a forgotten initialisation from another fault)
    static const int v = getenv("RP_EMUL_FILL") ? atoi(getenv("RP_EMUL_FILL")) : 0xA5;
    return v;
}
bool guard_mode() {
    static const bool on = getenv("RP_EMUL_GUARD") != nullptr;
    return on;
}
}  // namespace
hipError_t emu_hipMalloc(void** p, size_t bytes) {
    if (guard_mode()) {
        const size_t page = 4096, body = (std::max<size_t>(bytes, 1) + 15) & ~(size_t)15, span = (body + page - 1) & ~(page - 1);
        char* m = static_cast<char*>(mmap(nullptr, span + 2 * page, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0));
        if (m == MAP_FAILED) return hipErrorOutOfMemory;
        mprotect(m, page, PROT_NONE);
        mprotect(m + page + span, page, PROT_NONE);
        char* q = m + page + (span - body);
        if (body <= (size_t)1 << 30) memset(q, fill_byte(), body);
        std::lock_guard<std::mutex> lk(g_map_mu);
        g_maps.push_back({q, Mapping{m, span + 2 * page}});
        *p = q;
        return hipSuccess;
    }
    void* q = nullptr;
    const size_t n = (std::max<size_t>(bytes, 1) + 255) & ~(size_t)255;
    if (posix_memalign(&q, 256, n) != 0) return hipErrorOutOfMemory;
    if (n <= (size_t)1 << 30) memset(q, fill_byte(), n);  // fresh device memory is not zero: make forgotten initialisation visible
    *p = q;
    return hipSuccess;
}
hipError_t hipFree(void* p) {
    if (!p) return hipSuccess;
    if (guard_mode()) {
        std::lock_guard<std::mutex> lk(g_map_mu);
        for (size_t i = 0; i < g_maps.size(); ++i)
            if (g_maps[i].first == p) {
                munmap(g_maps[i].second.base, g_maps[i].second.len);
                g_maps[i] = g_maps.back();
                g_maps.pop_back();
                return hipSuccess;
            }
        return hipErrorInvalidValue;
    }
    free(p);
    return hipSuccess;
}
hipError_t hipMemcpy(void* dst, const void* src, size_t n, hipMemcpyKind) {
    if (n) memmove(dst, src, n);
    return hipSuccess;
}
hipError_t hipMemcpyAsync(void* dst, const void* src, size_t n, hipMemcpyKind, hipStream_t) {
    if (n) memmove(dst, src, n);
    return hipSuccess;
}
hipError_t hipMemset(void* dst, int v, size_t n) {
    if (n) memset(dst, v, n);
    return hipSuccess;
}
hipError_t hipMemsetAsync(void* dst, int v, size_t n, hipStream_t) {
    if (n) memset(dst, v, n);
    return hipSuccess;
}
hipError_t hipSetDevice(int d) { return d >= 0 && d < 8 ? hipSuccess : hipErrorInvalidValue; }  // eight names for the same host memory
hipError_t hipGetDevice(int* d) {
    *d = 0;
    return hipSuccess;
}
hipError_t hipGetDeviceCount(int* n) {
    *n = 8;
    return hipSuccess;
}
hipError_t hipDeviceSynchronize(void) { return hipSuccess; }
hipError_t hipGetLastError(void) { return hipSuccess; }
hipError_t hipPeekAtLastError(void) { return hipSuccess; }
const char* hipGetErrorString(hipError_t e) { return e == hipSuccess ? "hipSuccess (emulated)" : "hip error (emulated)"; }
hipError_t hipStreamCreate(hipStream_t* s) {
    *s = new ihipStream_t{1};
    return hipSuccess;
}
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { return hipStreamCreate(s); }
hipError_t hipStreamDestroy(hipStream_t s) {
    delete s;
    return hipSuccess;
}
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
hipError_t hipEventCreate(hipEvent_t* e) {
    *e = new ihipEvent_t{std::chrono::steady_clock::now()};
    return hipSuccess;
}
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { return hipEventCreate(e); }
hipError_t hipEventDestroy(hipEvent_t e) {
    delete e;
    return hipSuccess;
}
hipError_t hipEventRecord(hipEvent_t e, hipStream_t) {
    e->t = std::chrono::steady_clock::now();
    return hipSuccess;
}
hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) {
    *ms = std::chrono::duration<float, std::milli>(b->t - a->t).count();
    return hipSuccess;
}
hipError_t hipDeviceGetAttribute(int* v, hipDeviceAttribute_t, int) {
    *v = 256;
    return hipSuccess;
}
hipError_t hipMemGetInfo(size_t* free_b, size_t* total_b) {
    *free_b = (size_t)8 << 30;
    *total_b = (size_t)8 << 30;
    return hipSuccess;
}
__attribute__((visibility("default"))) void emu_stats(uint64_t* launches, uint64_t* blocks) {
    std::lock_guard<std::mutex> lk(emu::g_stats.mu);
    *launches = emu::g_stats.launches;
    *blocks = emu::g_stats.blocks;
}
__attribute__((visibility("default"))) uint64_t emu_split_rounds(void) { return emu::g_split_rounds.load(); }
__attribute__((visibility("default"))) unsigned emu_host_threads(void) { return emu::host_threads(); }
}
