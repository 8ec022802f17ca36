// tests/emul/fake_rccl.cpp — TEST INFRASTRUCTURE (tests/emul).
//
// The six RCCL entry points csrc/comm.cpp binds with dlsym, for ranks that are PROCESSES ON ONE HOST exchanging host memory
// through a POSIX shared-memory segment named after the communicator's unique id.  Built as tests/emul/_build/rccl/
// librp_emul_rccl.so; the emulated build of csrc/comm.cpp opens it in place of librccl.so.1 (tests/emul/build.py; the product
// sources are not touched), and with it the native sharded entry points (rp_mccfr_step_comm, rp_nlhe_step_comm, rp_kmeans_step_comm)
// run with a world of two on a machine without a GPU.  Collectives are synchronous (the emulated streams are too).
#include <fcntl.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <ctime>

namespace {
constexpr int MAX_WORLD = 16;
constexpr size_t SLOT_BYTES = (size_t)64 << 20;  // per rank, sparse: only what a collective touches is ever committed (a larger one fails)

struct Header {
    std::atomic<uint32_t> arrived;
    std::atomic<uint32_t> generation;
    std::atomic<uint32_t> attached;
    uint32_t world;
};
struct Comm {
    Header* hdr;
    char* slots;
    int rank, world;
    char name[64];
    size_t bytes;
};

void barrier(Comm* c) {
    Header* h = c->hdr;
    const uint32_t gen = h->generation.load(std::memory_order_acquire);
    if (h->arrived.fetch_add(1, std::memory_order_acq_rel) + 1 == (uint32_t)c->world) {
        h->arrived.store(0, std::memory_order_relaxed);
        h->generation.fetch_add(1, std::memory_order_release);
    } else {
        unsigned spins = 0;
        while (h->generation.load(std::memory_order_acquire) == gen) {
            if (++spins > 200) {
                timespec ts{0, 50000};
                nanosleep(&ts, nullptr);
            } else {
                sched_yield();
            }
        }
    }
}
size_t type_bytes(int t) {
    switch (t) {
        case 0: case 1: return 1;   // int8 / uint8
        case 2: case 3: return 4;   // int32 / uint32
        case 4: case 5: return 8;   // int64 / uint64
        default: return 0;
    }
}
}  // namespace

extern "C" {
typedef struct { char internal[128]; } ncclUniqueId;

__attribute__((visibility("default"))) int ncclGetUniqueId(ncclUniqueId* id) {
    memset(id, 0, sizeof *id);
    const int fd = open("/dev/urandom", O_RDONLY);
    if (fd < 0 || read(fd, id->internal, 16) != 16) return 1;
    close(fd);
    return 0;
}

__attribute__((visibility("default"))) int ncclCommInitRank(void** out, int world, ncclUniqueId id, int rank) {
    if (world < 1 || world > MAX_WORLD || rank < 0 || rank >= world) return 4;
    Comm* c = new Comm();
    c->rank = rank;
    c->world = world;
    uint64_t a, b;
    memcpy(&a, id.internal, 8);
    memcpy(&b, id.internal + 8, 8);
    snprintf(c->name, sizeof c->name, "/rp_emul_rccl_%016llx%016llx", (unsigned long long)a, (unsigned long long)b);
    c->bytes = 4096 + SLOT_BYTES * (size_t)world;
    const int fd = shm_open(c->name, O_CREAT | O_RDWR, 0600);
    if (fd < 0 || ftruncate(fd, (off_t)c->bytes) != 0) return 2;
    void* m = mmap(nullptr, c->bytes, PROT_READ | PROT_WRITE, MAP_SHARED | MAP_NORESERVE, fd, 0);
    close(fd);
    if (m == MAP_FAILED) return 2;
    c->hdr = static_cast<Header*>(m);  // a fresh segment is zero: counters start at 0
    c->slots = static_cast<char*>(m) + 4096;
    c->hdr->world = (uint32_t)world;
    c->hdr->attached.fetch_add(1);
    while (c->hdr->attached.load() < (uint32_t)world) sched_yield();  // every rank is mapped before anyone proceeds
    barrier(c);
    if (rank == 0) shm_unlink(c->name);  // the mappings keep it alive; nothing is left behind when the ranks exit
    *out = c;
    return 0;
}

__attribute__((visibility("default"))) int ncclCommDestroy(void* comm) {
    Comm* c = static_cast<Comm*>(comm);
    if (!c) return 0;
    munmap(c->hdr, c->bytes);
    delete c;
    return 0;
}

__attribute__((visibility("default"))) int ncclAllGather(const void* send, void* recv, size_t count, int type, void* comm, void*) {
    Comm* c = static_cast<Comm*>(comm);
    const size_t n = count * type_bytes(type);
    if (!type_bytes(type) || n > SLOT_BYTES) return 4;
    barrier(c);  // the previous collective's readers are done with the slots
    memcpy(c->slots + SLOT_BYTES * (size_t)c->rank, send, n);
    barrier(c);
    for (int r = 0; r < c->world; ++r) memcpy(static_cast<char*>(recv) + n * (size_t)r, c->slots + SLOT_BYTES * (size_t)r, n);
    return 0;
}

__attribute__((visibility("default"))) int ncclAllReduce(const void* send, void* recv, size_t count, int type, int op, void* comm, void*) {
    Comm* c = static_cast<Comm*>(comm);
    const size_t tb = type_bytes(type), n = count * tb;
    if (op != 0 || (tb != 4 && tb != 8) || n > SLOT_BYTES) return 4;  // integer sums only: all this library asks for
    barrier(c);
    memcpy(c->slots + SLOT_BYTES * (size_t)c->rank, send, n);
    barrier(c);
    for (size_t i = 0; i < count; ++i) {
        if (tb == 4) {
            uint32_t s = 0;
            for (int r = 0; r < c->world; ++r) s += reinterpret_cast<const uint32_t*>(c->slots + SLOT_BYTES * (size_t)r)[i];
            static_cast<uint32_t*>(recv)[i] = s;
        } else {
            uint64_t s = 0;
            for (int r = 0; r < c->world; ++r) s += reinterpret_cast<const uint64_t*>(c->slots + SLOT_BYTES * (size_t)r)[i];
            static_cast<uint64_t*>(recv)[i] = s;
        }
    }
    return 0;
}

__attribute__((visibility("default"))) const char* ncclGetErrorString(int code) {
    return code == 0 ? "success (emulated rccl)" : "emulated rccl error";
}
}
