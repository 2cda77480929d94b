// shm_rccl.cpp — TEST INFRASTRUCTURE, not product code: a stand-in for the five RCCL entry points comm.hip binds at run time
// (ncclGetUniqueId, ncclCommInitRank, ncclCommDestroy, ncclAllGather, ncclAllReduce, ncclGetErrorString), so that the library's
// multi-rank path — comet_comm_create, comet_index_search_sharded_async / _wait, merge_topk_kernel on the exchange stream, the
// slot ring, the all-reduce barrier — runs at world size 2 / 4 / 8 as PROCESSES SHARING ONE GPU (the gpurun boxes have one GPU,
// and real RCCL refuses several ranks on one device). comm.hip loads it instead of librccl when COMET_RCCL_LIB names it; nothing
// else in the library knows.
//
// Semantics: every rank issues the same sequence of collectives on a communicator, as with NCCL; unlike NCCL a collective here
// BLOCKS THE CALLING HOST THREAD until it is complete (stream drained -> this rank's block copied into the shared segment -> all
// ranks arrived -> gathered / reduced blocks copied back). Blocking inside stream callbacks instead (the asynchronous form) deadlocks:
// HIP runs the host functions of all streams of a process on one thread, the library issues collectives on two streams (bounds on the
// search stream, result blocks on the exchange stream), and two ranks can then each sit in the callback the other one needs next.
// What still runs asynchronously is everything the library enqueues around the collectives (searches, the merge on the exchange stream).
// Transport: a POSIX shared-memory segment named by the unique id, NSLOT slots used round-robin by sequence number.
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
void spin_until(std::atomic<long>& a, long target) {
    const time_t t0 = time(nullptr);
    while (a.load(std::memory_order_acquire) < target) {
        usleep(10);
        if (time(nullptr) - t0 > 120) { fprintf(stderr, "[shm_rccl] a rank did not arrive within 120 s — aborting\n"); abort(); }
    }
}
size_t dtype_size(int dt) { switch (dt) { case 0: case 1: return 1; case 2: case 3: case 7: return 4; case 4: case 5: case 8: return 8; case 6: return 2; default: return 4; } }

template <class T> void reduce_into(T* out, const unsigned char* base, size_t block_bytes, size_t count, int world, int op) {
    for (size_t i = 0; i < count; i++) {
        T acc = reinterpret_cast<const T*>(base)[i];
        for (int r = 1; r < world; r++) { const T v = reinterpret_cast<const T*>(base + (size_t)r * block_bytes)[i]; acc = op == 0 ? acc + v : (op == 2 ? (v > acc ? v : acc) : (v < acc ? v : acc)); }
        out[i] = acc;
    }
}
int collective(Comm* c, const void* send, void* recv, size_t bytes, bool reduce, int dtype, int op, size_t count, hipStream_t s) {
    if (bytes * c->world > SLOT_BYTES) { fprintf(stderr, "[shm_rccl] collective of %zu bytes x %d ranks exceeds the slot\n", bytes, c->world); return 2; }
    const long q = c->seq++;
    const int slot = (int)(q % NSLOT); const long use = q / NSLOT;
    unsigned char* base = c->data + (size_t)slot * SLOT_BYTES;
    Header* h = c->hdr;
    spin_until(h->depart[slot], (long)c->world * use);                 // the slot's previous use has been read by every rank
    if (hipMemcpyAsync(base + (size_t)c->rank * bytes, send, bytes, hipMemcpyDeviceToHost, s) != hipSuccess) return 1;
    if (hipStreamSynchronize(s) != hipSuccess) return 1;               // everything enqueued before the collective + the copy
    h->arrive[slot].fetch_add(1, std::memory_order_acq_rel);
    spin_until(h->arrive[slot], (long)c->world * (use + 1));
    if (reduce) {
        if (dtype == 8) reduce_into(static_cast<double*>(c->stage), base, bytes, count, c->world, op);
        else if (dtype == 7) reduce_into(static_cast<float*>(c->stage), base, bytes, count, c->world, op);
        else reduce_into(static_cast<int*>(c->stage), base, bytes, count, c->world, op);
        if (hipMemcpyAsync(recv, c->stage, bytes, hipMemcpyHostToDevice, s) != hipSuccess) return 1;
    } else if (hipMemcpyAsync(recv, base, bytes * c->world, hipMemcpyHostToDevice, s) != hipSuccess) return 1;
    if (hipStreamSynchronize(s) != hipSuccess) return 1;               // the slot (and the stage) may be reused after this
    h->depart[slot].fetch_add(1, std::memory_order_acq_rel);
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
    return collective(static_cast<Comm*>(comm), send, recv, count * dtype_size(dtype), false, dtype, 0, count, s);
}
// ncclRedOp_t: 0 sum, 1 prod, 2 max, 3 min
__attribute__((visibility("default"))) int ncclAllReduce(const void* send, void* recv, size_t count, int dtype, int op, void* comm, hipStream_t s) {
    if (count * dtype_size(dtype) > (1 << 20) || op == 1) return 4;
    return collective(static_cast<Comm*>(comm), send, recv, count * dtype_size(dtype), true, dtype, op, count, s);
}
__attribute__((visibility("default"))) const char* ncclGetErrorString(int r) {
    switch (r) { case 0: return "success"; case 1: return "HIP call failed"; case 2: return "shared-memory transport failed"; case 4: return "invalid argument"; default: return "error"; }
}

}  // extern "C"
