// shm_rccl.cpp — TEST INFRASTRUCTURE, not product code: a stand-in for the five RCCL entry points comm.hip binds at run time
// (ncclGetUniqueId, ncclCommInitRank, ncclCommDestroy, ncclAllGather, ncclAllReduce, ncclGetErrorString), so that the library's
// multi-rank path — comet_comm_create, comet_index_search_sharded_async / _wait, merge_topk_kernel on the exchange stream, the
// slot ring, the all-reduce barrier — runs at world size 2 / 4 / 8 as PROCESSES SHARING ONE GPU (the gpurun boxes have one GPU,
// and real RCCL refuses several ranks on one device). comm.hip loads it instead of librccl when COMET_RCCL_LIB names it; nothing
// else in the library knows.
//
// Semantics kept from NCCL: collectives are enqueued on the caller's stream and are asynchronous to the host; every rank issues
// the same sequence of collectives on a communicator. Transport: a POSIX shared-memory segment named by the unique id. One
// collective = [host function: wait until the slot's previous use was read by every rank] -> device-to-host copy of this rank's
// block into the slot -> [host function: arrive, wait for all ranks] -> host-to-device copy of the gathered blocks (all-reduce:
// reduced on the host first) -> [host function: depart]. NSLOT slots are used round-robin by sequence number, so up to NSLOT
// collectives can be in flight.
#include <hip/hip_runtime.h>

#include <atomic>
#include <cerrno>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

namespace {
constexpr int NSLOT = 8, MAX_RANKS = 16;
constexpr size_t SLOT_BYTES = (size_t)4 << 20;        // per slot: world x block (/dev/shm of a container is small)
struct Header {
    std::atomic<int> ready;                 // ranks attached
    std::atomic<long> arrive[NSLOT], depart[NSLOT];
};
struct Comm {
    int rank = 0, world = 1;
    char name[64] = {0};
    Header* hdr = nullptr; unsigned char* data = nullptr; size_t map_bytes = 0;
    long seq = 0;
    void* stage = nullptr;                  // pinned staging for all-reduce results
};
struct Op { Comm* c; int slot; long use; int phase; size_t bytes; int dtype, op; size_t count; };   // phase 0: wait depart, 1: arrive + wait, 2: depart, 3: arrive + wait + reduce

void spin_until(std::atomic<long>& a, long target) {
    const time_t t0 = time(nullptr);
    while (a.load(std::memory_order_acquire) < target) {
        usleep(20);
        if (time(nullptr) - t0 > 120) { fprintf(stderr, "[shm_rccl] a rank did not arrive within 120 s — aborting\n"); abort(); }
    }
}
size_t dtype_size(int dt) { switch (dt) { case 0: case 1: return 1; case 2: case 3: case 7: return 4; case 4: case 5: case 8: return 8; case 6: return 2; default: return 4; } }

void host_fn(void* p) {
    Op* o = static_cast<Op*>(p);
    Header* h = o->c->hdr;
    const int w = o->c->world;
    if (o->phase == 0) spin_until(h->depart[o->slot], (long)w * o->use);
    else if (o->phase == 1 || o->phase == 3) {
        h->arrive[o->slot].fetch_add(1, std::memory_order_acq_rel);
        spin_until(h->arrive[o->slot], (long)w * (o->use + 1));
        if (o->phase == 3) {                 // reduce the world blocks of the slot into the pinned stage
            unsigned char* base = o->c->data + (size_t)o->slot * SLOT_BYTES;
            for (size_t i = 0; i < o->count; i++) {
                if (o->dtype == 8) {         // ncclFloat64
                    double acc = reinterpret_cast<double*>(base)[i];
                    for (int r = 1; r < w; r++) { const double v = reinterpret_cast<double*>(base + (size_t)r * o->bytes)[i]; acc = o->op == 0 ? acc + v : (o->op == 2 ? (v > acc ? v : acc) : (v < acc ? v : acc)); }
                    static_cast<double*>(o->c->stage)[i] = acc;
                } else if (o->dtype == 7) {  // ncclFloat32
                    float acc = reinterpret_cast<float*>(base)[i];
                    for (int r = 1; r < w; r++) { const float v = reinterpret_cast<float*>(base + (size_t)r * o->bytes)[i]; acc = o->op == 0 ? acc + v : (o->op == 2 ? (v > acc ? v : acc) : (v < acc ? v : acc)); }
                    static_cast<float*>(o->c->stage)[i] = acc;
                } else {                     // ncclInt32 / ncclUint32
                    int acc = reinterpret_cast<int*>(base)[i];
                    for (int r = 1; r < w; r++) { const int v = reinterpret_cast<int*>(base + (size_t)r * o->bytes)[i]; acc = o->op == 0 ? acc + v : (o->op == 2 ? (v > acc ? v : acc) : (v < acc ? v : acc)); }
                    static_cast<int*>(o->c->stage)[i] = acc;
                }
            }
        }
    } else h->depart[o->slot].fetch_add(1, std::memory_order_acq_rel);
    delete o;
}
int enqueue(Comm* c, const void* send, void* recv, size_t bytes, bool reduce, int dtype, int op, size_t count, hipStream_t s) {
    if (bytes * c->world > SLOT_BYTES) { fprintf(stderr, "[shm_rccl] collective of %zu bytes x %d ranks exceeds the slot\n", bytes, c->world); return 2; }
    const long q = c->seq++;
    const int slot = (int)(q % NSLOT); const long use = q / NSLOT;
    unsigned char* base = c->data + (size_t)slot * SLOT_BYTES;
    if (hipLaunchHostFunc(s, host_fn, new Op{c, slot, use, 0, bytes, dtype, op, count}) != hipSuccess) return 1;
    if (hipMemcpyAsync(base + (size_t)c->rank * bytes, send, bytes, hipMemcpyDeviceToHost, s) != hipSuccess) return 1;
    if (hipLaunchHostFunc(s, host_fn, new Op{c, slot, use, reduce ? 3 : 1, bytes, dtype, op, count}) != hipSuccess) return 1;
    if (reduce) { if (hipMemcpyAsync(recv, c->stage, bytes, hipMemcpyHostToDevice, s) != hipSuccess) return 1; }
    else if (hipMemcpyAsync(recv, base, bytes * c->world, hipMemcpyHostToDevice, s) != hipSuccess) return 1;
    if (hipLaunchHostFunc(s, host_fn, new Op{c, slot, use, 2, bytes, dtype, op, count}) != hipSuccess) return 1;
    return 0;
}
}  // namespace

extern "C" {

struct ncclUniqueId128 { char internal[128]; };

__attribute__((visibility("default"))) int ncclGetUniqueId(void* out) {
    char* o = static_cast<char*>(out);
    memset(o, 0, 128);
    snprintf(o, 64, "/comet_shm_rccl_%d_%ld", (int)getpid(), (long)time(nullptr));
    return 0;
}
__attribute__((visibility("default"))) int ncclCommInitRank(void** comm, int world, ncclUniqueId128 id, int rank) {
    if (world < 1 || world > MAX_RANKS || rank < 0 || rank >= world) return 4;
    Comm* c = new Comm();
    c->rank = rank; c->world = world;
    memcpy(c->name, id.internal, 63);
    static_assert(sizeof(Header) <= 4096, "header page");
    c->map_bytes = 4096 + (size_t)NSLOT * SLOT_BYTES;
    int fd = -1;
    if (rank == 0) {
        fd = shm_open(c->name, O_CREAT | O_RDWR | O_EXCL, 0600);
        if (fd < 0 || ftruncate(fd, (off_t)c->map_bytes) != 0) { perror("[shm_rccl] shm_open/ftruncate"); delete c; return 2; }
    } else {
        for (int tries = 0; tries < 60000 && fd < 0; tries++) {      // until rank 0 created and sized it
            fd = shm_open(c->name, O_RDWR, 0600);
            struct stat st;
            if (fd >= 0 && (fstat(fd, &st) != 0 || (size_t)st.st_size < c->map_bytes)) { close(fd); fd = -1; }
            if (fd < 0) usleep(1000);
        }
        if (fd < 0) { fprintf(stderr, "[shm_rccl] rank %d: segment %s never appeared\n", rank, c->name); delete c; return 2; }
    }
    void* m = mmap(nullptr, c->map_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (m == MAP_FAILED) { perror("[shm_rccl] mmap"); delete c; return 2; }
    c->hdr = static_cast<Header*>(m);                     // a fresh segment is zero-filled: every counter starts at 0
    c->data = static_cast<unsigned char*>(m) + 4096;
    (void)hipHostRegister(m, c->map_bytes, hipHostRegisterDefault);       // async copies where the runtime can pin the mapping; pageable staging otherwise
    if (hipHostMalloc(&c->stage, 1 << 20, hipHostMallocDefault) != hipSuccess) { delete c; return 1; }
    c->hdr->ready.fetch_add(1);
    const time_t t0 = time(nullptr);
    while (c->hdr->ready.load() < world) { usleep(100); if (time(nullptr) - t0 > 120) { fprintf(stderr, "[shm_rccl] rank %d: not all ranks attached\n", rank); return 2; } }
    if (rank == 0) shm_unlink(c->name);                   // everyone is attached: the name can go, the memory lives while mapped
    *comm = c;
    return 0;
}
__attribute__((visibility("default"))) int ncclCommDestroy(void* comm) {
    Comm* c = static_cast<Comm*>(comm);
    if (!c) return 0;
    if (c->stage) (void)hipHostFree(c->stage);
    if (c->hdr) munmap(c->hdr, c->map_bytes);
    delete c;
    return 0;
}
__attribute__((visibility("default"))) int ncclAllGather(const void* send, void* recv, size_t count, int dtype, void* comm, hipStream_t s) {
    return enqueue(static_cast<Comm*>(comm), send, recv, count * dtype_size(dtype), false, dtype, 0, count, s);
}
// ncclRedOp_t: 0 sum, 1 prod, 2 max, 3 min
__attribute__((visibility("default"))) int ncclAllReduce(const void* send, void* recv, size_t count, int dtype, int op, void* comm, hipStream_t s) {
    if (count * dtype_size(dtype) > (1 << 20) || op == 1) return 4;
    return enqueue(static_cast<Comm*>(comm), send, recv, count * dtype_size(dtype), true, dtype, op, count, s);
}
__attribute__((visibility("default"))) const char* ncclGetErrorString(int r) {
    switch (r) { case 0: return "success"; case 1: return "HIP call failed"; case 2: return "shared-memory transport failed"; case 4: return "invalid argument"; default: return "error"; }
}

}  // extern "C"
