// seq_rccl.cpp — TEST / MEASUREMENT INFRASTRUCTURE, not product code: a stand-in for the RCCL entry points comm.hip binds at run time
// (ncclGetUniqueId, ncclCommInitRank, ncclCommDestroy, ncclAllGather, ncclAllReduce, ncclGetErrorString) that lets ONE process on ONE GPU play
// every rank of an N-rank job, one rank after the other, and still hand each rank what its peers would have sent:
//
//   pass 1 (RECORD): every virtual rank runs its sequence of searches. A collective stores the rank's contribution in a per-(rank, call
//                    number) log and answers with the rank's own data only (all-reduce: the input; all-gather: the own block, zeros elsewhere).
//   pass 2 (REPLAY): the same sequences again. Call number i of any rank is answered from the logs of ALL ranks' call number i — the true
//                    reduction / the true gathered blocks — by small device kernels on the caller's stream (no host synchronisation).
//
// That is exact for collectives whose CONTRIBUTIONS do not depend on earlier collective results — which holds for the library: a rank's stage-1
// bounds of the sharded IVFPQ search and its per-shard top-K block are functions of its shard and the queries only (bounds change how much a
// rank scans, never what it returns). tools/shard_probe.py uses it to time ONE rank of an N-GPU job with the bounds and blocks of all N
// (the one-GPU scaling model); what it cannot time is the collective's own transfer, which the model prices from the block size.
// Loaded by comm.hip when COMET_RCCL_LIB names it. Control: seq_rccl_set_mode(0 record / 1 replay), seq_rccl_reset() (forget the logs).
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <vector>

namespace {
constexpr int MAX_RANKS = 64;
struct Comm { int rank = 0, world = 1; long seq = 0; };
struct Entry { void* dev = nullptr; size_t bytes = 0; };
std::mutex g_mu;
int g_mode = 0;                                    // 0 record, 1 replay
std::map<long, Entry> g_log[MAX_RANKS];            // [rank][call number] -> the rank's contribution (device copy)
std::vector<Comm*> g_comms;
size_t dtype_size(int dt) { switch (dt) { case 0: case 1: return 1; case 2: case 3: case 7: return 4; case 4: case 5: case 8: return 8; case 6: return 2; default: return 4; } }

struct Ptrs { const void* p[MAX_RANKS]; };
template <class T> __global__ void reduce_kernel(Ptrs src, int world, T* out, size_t count, int op) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    T acc = reinterpret_cast<const T*>(src.p[0])[i];
    for (int r = 1; r < world; r++) { const T v = reinterpret_cast<const T*>(src.p[r])[i]; acc = op == 0 ? acc + v : (op == 2 ? (v > acc ? v : acc) : (v < acc ? v : acc)); }
    out[i] = acc;
}
bool log_store(int rank, long seq, const void* send, size_t bytes, hipStream_t s) {
    Entry& e = g_log[rank][seq];
    if (e.bytes != bytes) { if (e.dev) (void)hipFree(e.dev); e.dev = nullptr; if (hipMalloc(&e.dev, bytes ? bytes : 4) != hipSuccess) return false; e.bytes = bytes; }
    return hipMemcpyAsync(e.dev, send, bytes, hipMemcpyDeviceToDevice, s) == hipSuccess;
}
}  // namespace

extern "C" {
__attribute__((visibility("default"))) void seq_rccl_set_mode(int mode) {
    std::lock_guard<std::mutex> lk(g_mu);
    g_mode = mode;
    for (Comm* c : g_comms) c->seq = 0;            // both passes number their calls from zero
}
__attribute__((visibility("default"))) void seq_rccl_reset(void) {
    std::lock_guard<std::mutex> lk(g_mu);
    (void)hipDeviceSynchronize();
    for (auto& m : g_log) { for (auto& kv : m) if (kv.second.dev) (void)hipFree(kv.second.dev); m.clear(); }
    for (Comm* c : g_comms) c->seq = 0;
}
__attribute__((visibility("default"))) int ncclGetUniqueId(void* id) { std::memset(id, 0, 128); std::memcpy(id, "seq_rccl", 8); return 0; }
struct UniqueId { char internal[128]; };
__attribute__((visibility("default"))) int ncclCommInitRank(void** comm, int world, UniqueId, int rank) {
    if (world > MAX_RANKS || rank < 0 || rank >= world) return 4;
    std::lock_guard<std::mutex> lk(g_mu);
    Comm* c = new Comm(); c->rank = rank; c->world = world;
    g_comms.push_back(c);
    *comm = c;
    return 0;
}
__attribute__((visibility("default"))) int ncclCommDestroy(void* comm) {
    std::lock_guard<std::mutex> lk(g_mu);
    Comm* c = static_cast<Comm*>(comm);
    for (size_t i = 0; i < g_comms.size(); i++) if (g_comms[i] == c) { g_comms.erase(g_comms.begin() + i); break; }
    delete c;
    return 0;
}
__attribute__((visibility("default"))) const char* ncclGetErrorString(int r) { return r == 0 ? "ok" : (r == 5 ? "replay without a recorded contribution of every rank" : "seq_rccl error"); }

__attribute__((visibility("default"))) int ncclAllReduce(const void* send, void* recv, size_t count, int dtype, int op, void* comm, hipStream_t s) {
    std::lock_guard<std::mutex> lk(g_mu);
    Comm* c = static_cast<Comm*>(comm);
    const size_t bytes = count * dtype_size(dtype);
    const long seq = c->seq++;
    if (g_mode == 0) {
        if (!log_store(c->rank, seq, send, bytes, s)) return 1;
        if (recv != send && hipMemcpyAsync(recv, send, bytes, hipMemcpyDeviceToDevice, s) != hipSuccess) return 1;
        return 0;
    }
    Ptrs p{};
    for (int r = 0; r < c->world; r++) {
        auto it = g_log[r].find(seq);
        if (it == g_log[r].end() || it->second.bytes != bytes) return 5;
        p.p[r] = it->second.dev;
    }
    const unsigned grid = (unsigned)((count + 255) / 256);
    if (dtype == 7) hipLaunchKernelGGL(reduce_kernel<float>, dim3(grid), dim3(256), 0, s, p, c->world, (float*)recv, count, op);
    else if (dtype == 8) hipLaunchKernelGGL(reduce_kernel<double>, dim3(grid), dim3(256), 0, s, p, c->world, (double*)recv, count, op);
    else if (dtype == 2) hipLaunchKernelGGL(reduce_kernel<int>, dim3(grid), dim3(256), 0, s, p, c->world, (int*)recv, count, op);
    else return 3;
    return hipGetLastError() == hipSuccess ? 0 : 1;
}
__attribute__((visibility("default"))) int ncclAllGather(const void* send, void* recv, size_t count, int dtype, void* comm, hipStream_t s) {
    std::lock_guard<std::mutex> lk(g_mu);
    Comm* c = static_cast<Comm*>(comm);
    const size_t bytes = count * dtype_size(dtype);
    const long seq = c->seq++;
    if (g_mode == 0) {
        if (!log_store(c->rank, seq, send, bytes, s)) return 1;
        if (hipMemsetAsync(recv, 0, bytes * c->world, s) != hipSuccess) return 1;
        if (hipMemcpyAsync((char*)recv + (size_t)c->rank * bytes, send, bytes, hipMemcpyDeviceToDevice, s) != hipSuccess) return 1;
        return 0;
    }
    for (int r = 0; r < c->world; r++) {
        auto it = g_log[r].find(seq);
        if (it == g_log[r].end() || it->second.bytes != bytes) return 5;
        if (hipMemcpyAsync((char*)recv + (size_t)r * bytes, it->second.dev, bytes, hipMemcpyDeviceToDevice, s) != hipSuccess) return 1;
    }
    return 0;
}
}  // extern "C"
