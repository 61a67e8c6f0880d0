// th_comm.hip - the RCCL transport of a context (th_kernels.hpp: Transport): the path's collectives issued by the library
// itself, on the context's own stream, so that ANY host (the Node host through th_napi.cc, the Python host through ctypes)
// runs row-band shards with one process per GPU and no other transport than the 128-byte id it hands from rank 0 to the others.
//
// What the integrator exchanges is the statistics block (particles shard by row band, the flow texture is
// replicated, "only a small RCCL all-reduce over xGMI for global spawn/stat counters"): th_counters = 5 x u64 and one f64
// added up, one f64 maximised - three ncclAllReduce calls inside one group (one launch), in place on the device block
// th_stats_async fills.  The same communicator carries the row-band exchanges of the spawners (all-gather of a ring
// buffer, th_api.hip: th_comm_allgather_state).
//
// librccl is bound at run time: the copy the process has already mapped (torch's, when the Python host imported torch -
// one HIP runtime per process, tendrils_amd/_capi.py) or the system's; nothing here is linked against it, so a
// single-GPU host never loads it.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>

#include "th_kernels.hpp"

namespace th {
namespace {

struct Rccl {
    void *lib = nullptr;
    std::string where, error;
    decltype(&ncclGetVersion) GetVersion = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclCommCount) CommCount = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclBroadcast) Broadcast = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
};

Rccl &rccl()
{
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        const char *env = getenv("TH_RCCL_LIB");
        const char *names[] = {env, "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
        for (const char *n : names) {
            if (!n || !*n) continue;
            // RTLD_NOLOAD first: a copy the process already holds (by SONAME) wins over a second one from another path
            void *h = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
            if (!h) h = dlopen(n, RTLD_NOW | RTLD_LOCAL);
            if (h) { r.lib = h; r.where = n; break; }
            const char *e = dlerror();
            r.error += std::string(r.error.empty() ? "" : "; ") + n + ": " + (e ? e : "?");
        }
        if (!r.lib) return;
        bool all = true;
#define TH_SYM(f) do { r.f = reinterpret_cast<decltype(r.f)>(dlsym(r.lib, "nccl" #f)); if (!r.f) { all = false; r.error += " missing nccl" #f; } } while (0)
        TH_SYM(GetVersion); TH_SYM(GetUniqueId); TH_SYM(CommInitRank); TH_SYM(CommDestroy); TH_SYM(CommCount);
        TH_SYM(GetErrorString); TH_SYM(AllReduce); TH_SYM(AllGather); TH_SYM(Broadcast); TH_SYM(Send); TH_SYM(Recv);
        TH_SYM(GroupStart); TH_SYM(GroupEnd);
#undef TH_SYM
        if (!all) { dlclose(r.lib); r.lib = nullptr; }
    });
    return r;
}

thread_local std::string g_comm_error;

#define TH_NCCL(expr)                                                                                     \
    do {                                                                                                  \
        const ncclResult_t r_ = (expr);                                                                   \
        if (r_ != ncclSuccess) return comm_fail(std::string(#expr " failed: ") + R.GetErrorString(r_));   \
    } while (0)

// ncclGroupStart ... ncclGroupEnd around a scope: a call that fails inside still closes the group (an open group would
// swallow every later collective of the thread, the communicator's destruction included)
struct Group {
    Rccl &R;
    bool open = false;
    explicit Group(Rccl &r) : R(r) {}
    ncclResult_t start() { const ncclResult_t r = R.GroupStart(); open = r == ncclSuccess; return r; }
    ncclResult_t end() { open = false; return R.GroupEnd(); }
    ~Group() { if (open) (void)R.GroupEnd(); }
};

int rccl_destroy(void *comm)
{
    Rccl &R = rccl();
    if (!R.lib || !comm) return 0;
    TH_NCCL(R.CommDestroy(static_cast<ncclComm_t>(comm)));
    return 0;
}

int rccl_allreduce_counters(void *comm, void *counters_dev, hipStream_t stream)
{
    Rccl &R = rccl();
    ncclComm_t c = static_cast<ncclComm_t>(comm);
    char *b = static_cast<char *>(counters_dev);
    Group g(R);
    TH_NCCL(g.start());
    TH_NCCL(R.AllReduce(b, b, 5, ncclUint64, ncclSum, c, stream));
    TH_NCCL(R.AllReduce(b + 40, b + 40, 1, ncclFloat64, ncclSum, c, stream));
    TH_NCCL(R.AllReduce(b + 48, b + 48, 1, ncclFloat64, ncclMax, c, stream));
    TH_NCCL(g.end());
    return 0;
}

// (bands may differ by a row, the last owner's texel range may be short: broadcasts in one group rather than ncclAllGather,
// which wants equal parts)
int rccl_allgather_bytes(void *comm, const void *send, void *recv, const size_t *bytes, const size_t *offset, int rank, int world, hipStream_t stream)
{
    Rccl &R = rccl();
    ncclComm_t c = static_cast<ncclComm_t>(comm);
    bool equal = true;
    for (int r = 0; r < world; ++r) equal = equal && bytes[r] == bytes[0] && offset[r] == (size_t)r * bytes[0];
    if (equal) {
        TH_NCCL(R.AllGather(send, recv, bytes[0], ncclChar, c, stream));
        return 0;
    }
    Group g(R);
    TH_NCCL(g.start());
    for (int r = 0; r < world; ++r) {
        char *at = static_cast<char *>(recv) + offset[r];
        if (bytes[r]) TH_NCCL(R.Broadcast(r == rank ? send : at, at, bytes[r], ncclChar, r, c, stream));
    }
    TH_NCCL(g.end());
    return 0;
}

int rccl_alltoallv(void *comm, const void *send, const size_t *send_counts, const size_t *send_off, void *recv, const size_t *recv_counts,
                   const size_t *recv_off, size_t elem, int world, hipStream_t stream)
{
    Rccl &R = rccl();
    ncclComm_t c = static_cast<ncclComm_t>(comm);
    Group g(R);
    TH_NCCL(g.start());
    for (int r = 0; r < world; ++r) {
        if (send_counts[r]) TH_NCCL(R.Send(static_cast<const char *>(send) + send_off[r] * elem, send_counts[r] * elem, ncclChar, r, c, stream));
        if (recv_counts[r]) TH_NCCL(R.Recv(static_cast<char *>(recv) + recv_off[r] * elem, recv_counts[r] * elem, ncclChar, r, c, stream));
    }
    TH_NCCL(g.end());
    return 0;
}

int rccl_allreduce_max_u32(void *comm, uint32_t *word_dev, hipStream_t stream)
{
    Rccl &R = rccl();
    TH_NCCL(R.AllReduce(word_dev, word_dev, 1, ncclUint32, ncclMax, static_cast<ncclComm_t>(comm), stream));
    return 0;
}

const Transport kRccl = {"rccl", rccl_destroy, rccl_allreduce_counters, rccl_allgather_bytes, rccl_alltoallv, rccl_allreduce_max_u32};

}  // namespace

const char *comm_error() { return g_comm_error.c_str(); }
int comm_fail(const std::string &m) { g_comm_error = m; return 1; }

// 0 = ok; otherwise comm_error() says why
int comm_available(int *version)
{
    Rccl &R = rccl();
    if (!R.lib) return comm_fail("librccl could not be loaded (" + R.error + ")");
    int v = 0;
    TH_NCCL(R.GetVersion(&v));
    if (version) *version = v;
    return 0;
}

int comm_unique_id(void *out, size_t bytes)
{
    Rccl &R = rccl();
    if (!R.lib) return comm_fail("librccl could not be loaded (" + R.error + ")");
    if (bytes != sizeof(ncclUniqueId)) return comm_fail("a communicator id is " + std::to_string(sizeof(ncclUniqueId)) + " bytes");
    ncclUniqueId id;
    TH_NCCL(R.GetUniqueId(&id));
    memcpy(out, &id, sizeof id);
    return 0;
}

// (the device of the calling context is current)
int comm_init(void **comm, const Transport **transport, const void *id_bytes, size_t bytes, int rank, int world)
{
#ifdef TH_TESTING
    if (loopback_id(id_bytes)) return loopback_init(comm, transport, id_bytes, bytes, rank, world);      // (th_loopback.hip: test builds)
#endif
    Rccl &R = rccl();
    if (!R.lib) return comm_fail("librccl could not be loaded (" + R.error + ")");
    if (bytes != sizeof(ncclUniqueId)) return comm_fail("a communicator id is " + std::to_string(sizeof(ncclUniqueId)) + " bytes");
    ncclUniqueId id;
    memcpy(&id, id_bytes, sizeof id);
    ncclComm_t c = nullptr;
    TH_NCCL(R.CommInitRank(&c, world, id, rank));
    int n = 0;
    TH_NCCL(R.CommCount(c, &n));
    if (n != world) { R.CommDestroy(c); return comm_fail("the communicator holds " + std::to_string(n) + " ranks, not " + std::to_string(world)); }
    *comm = c;
    *transport = &kRccl;
    return 0;
}

}  // namespace th
