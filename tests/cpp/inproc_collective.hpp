// inproc_collective.hpp -- TEST INFRASTRUCTURE: an all-gather / all-reduce between the host THREADS of one process, each
// thread playing one rank of a row-partitioned Ell on the SAME GPU (RCCL refuses two ranks per device, and the build /
// test boxes have one GPU).  Used twice: by tests/cpp/fake_rccl.cpp (a stand-in librccl.so that libellhip.so opens
// through ELLHIP_RCCL_PATH, so the library's own RCCL call path runs with > 1 rank) and by
// tests/cpp/sharded_ranks_runner.cpp as the host-supplied collective of ellhip_sharded_create_custom.
// Semantics follow the real collectives as include/ellhip_sharded.h uses them: in place, on the caller's stream (here:
// the stream is drained, the exchange is done with blocking copies, and the result is in place when the call returns).
// Every wait is bounded (a rank that never arrives fails the call instead of hanging the test).
#pragma once

#include <hip/hip_runtime_api.h>

#include <chrono>
#include <condition_variable>
#include <mutex>
#include <vector>

namespace inproc {

struct Group {
    int nranks = 0;
    std::mutex m;
    std::condition_variable cv;
    int arrived = 0;
    long gen = 0;
    std::vector<const double*> send;
    std::vector<double*> recv;
    long ncalls = 0;  // collectives completed (counted by rank 0)
    explicit Group(int p) : nranks(p), send((size_t)p, nullptr), recv((size_t)p, nullptr) {}

    bool barrier() {
        std::unique_lock<std::mutex> lk(m);
        const long my = gen;
        if (++arrived == nranks) {
            arrived = 0;
            ++gen;
            cv.notify_all();
            return true;
        }
        return cv.wait_for(lk, std::chrono::seconds(120), [&] { return gen != my; });
    }
};

// in place: `send` = recv + rank * count
inline int allgather(Group& g, int rank, const double* send, double* recv, size_t count, hipStream_t st) {
    if (hipStreamSynchronize(st) != hipSuccess) return 1;  // this rank's part is complete
    g.send[(size_t)rank] = send;
    g.recv[(size_t)rank] = recv;
    if (!g.barrier()) return 2;
    // (on the caller's stream and waited for: a device-to-device hipMemcpy goes to the null stream and may return before it has run,
    // while the caller's stream is non-blocking -- the kernels that read `recv` next, or a peer rewriting its part behind the barrier
    // below, could overtake the copy)
    for (int r = 0; r < g.nranks; ++r) {
        if (r == rank) continue;
        if (hipMemcpyAsync(recv + (size_t)r * count, g.send[(size_t)r], count * sizeof(double), hipMemcpyDeviceToDevice, st) != hipSuccess) return 3;
    }
    if (hipStreamSynchronize(st) != hipSuccess) return 3;
    if (!g.barrier()) return 2;  // nobody rewrites its part while a peer still reads it
    if (rank == 0) ++g.ncalls;
    return 0;
}

// element-wise sum in RANK ORDER on the host: the same bits on every rank
inline int allreduce(Group& g, int rank, const double* send, double* recv, size_t count, hipStream_t st) {
    if (hipStreamSynchronize(st) != hipSuccess) return 1;
    g.send[(size_t)rank] = send;
    g.recv[(size_t)rank] = recv;
    if (!g.barrier()) return 2;
    std::vector<double> acc(count), tmp(count);
    for (int r = 0; r < g.nranks; ++r) {
        if (hipMemcpy(r == 0 ? acc.data() : tmp.data(), g.send[(size_t)r], count * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) return 3;
        if (r > 0)
            for (size_t i = 0; i < count; ++i) acc[i] += tmp[i];
    }
    if (!g.barrier()) return 2;  // every rank has read every send buffer (recv aliases send)
    if (hipMemcpyAsync(recv, acc.data(), count * sizeof(double), hipMemcpyHostToDevice, st) != hipSuccess) return 3;
    if (hipStreamSynchronize(st) != hipSuccess) return 3;  // (`acc` leaves scope; and the caller's next kernel reads `recv`)
    if (rank == 0) ++g.ncalls;
    return 0;
}

}  // namespace inproc
