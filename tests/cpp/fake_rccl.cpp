// fake_rccl.cpp -- TEST DOUBLE for librccl: the six entry points libellhip.so opens (csrc/sharded_capi.inc.hpp), with
// the ranks of a communicator being host THREADS of one process on one GPU (tests/cpp/inproc_collective.hpp).  Built as
// tests/cpp/_build/libfakerccl.so and handed to the library through ELLHIP_RCCL_PATH, so that the library's RCCL call
// path -- ncclUniqueId by value, in-place ncclAllGather at offset row0, ncclAllReduce of the symmetric shards, the
// communicator's lifetime -- executes with nranks > 1 on a one-GPU box.  Signatures as in <rccl/rccl.h>.
#include <cstdint>
#include <cstring>
#include <map>
#include <memory>
#include <string>

#include "inproc_collective.hpp"

namespace {
struct Id {
    char internal[128];
};
struct Comm {
    std::shared_ptr<inproc::Group> group;
    int rank;
};
std::mutex g_m;
std::map<std::string, std::weak_ptr<inproc::Group>> g_groups;
uint64_t g_next_id = 1;
constexpr int kDouble = 8, kSum = 0;
}  // namespace

extern "C" {

int ncclGetUniqueId(Id* id) {
    if (!id) return 4;  // ncclInvalidArgument
    std::lock_guard<std::mutex> lk(g_m);
    std::memset(id->internal, 0, sizeof id->internal);
    const uint64_t v = g_next_id++;
    std::memcpy(id->internal, "FAKERCCL", 8);
    std::memcpy(id->internal + 8, &v, sizeof v);
    return 0;
}

int ncclCommInitRank(void** comm, int nranks, Id id, int rank) {
    if (!comm || nranks < 1 || rank < 0 || rank >= nranks || std::memcmp(id.internal, "FAKERCCL", 8) != 0) return 4;
    std::lock_guard<std::mutex> lk(g_m);
    const std::string key(id.internal, sizeof id.internal);
    std::shared_ptr<inproc::Group> g = g_groups[key].lock();
    if (!g) {
        g = std::make_shared<inproc::Group>(nranks);
        g_groups[key] = g;
    }
    if (g->nranks != nranks) return 4;
    *comm = new Comm{g, rank};
    return 0;
}

int ncclCommDestroy(void* comm) {
    delete static_cast<Comm*>(comm);
    return 0;
}

int ncclAllGather(const void* sendbuff, void* recvbuff, size_t sendcount, int datatype, void* comm, hipStream_t stream) {
    Comm* c = static_cast<Comm*>(comm);
    if (!c || datatype != kDouble) return 4;
    // the library calls it in place (send = recv + rank * count); anything else is not what this double models
    if (static_cast<const double*>(sendbuff) != static_cast<double*>(recvbuff) + (size_t)c->rank * sendcount) return 4;
    return inproc::allgather(*c->group, c->rank, static_cast<const double*>(sendbuff), static_cast<double*>(recvbuff), sendcount, stream) ? 1 : 0;
}

int ncclAllReduce(const void* sendbuff, void* recvbuff, size_t count, int datatype, int op, void* comm, hipStream_t stream) {
    Comm* c = static_cast<Comm*>(comm);
    if (!c || datatype != kDouble || op != kSum) return 4;
    return inproc::allreduce(*c->group, c->rank, static_cast<const double*>(sendbuff), static_cast<double*>(recvbuff), count, stream) ? 1 : 0;
}

const char* ncclGetErrorString(int code) { return code == 0 ? "no error" : (code == 4 ? "invalid argument (fake rccl)" : "unhandled error (fake rccl)"); }

}  // extern "C"
