// th_loopback.hip - the in-process transport of a context (th_kernels.hpp: Transport): the ranks of a job are contexts of
// ONE process - on one device or several - each driven by its own host thread.  A collective is: every rank finishes
// its stream, posts its (pointers, counts, offsets) on a board in host memory, all meet at a barrier, every rank copies
// what is addressed to it device to device on its own stream and finishes it, all meet again.  Nothing of the path's
// arithmetic lives here (th_shard.hip computes who gets what); what this file adds is the possibility to RUN that
// arithmetic with 2, 3, 4 ... ranks on a box with one GPU (tests/test_gpu_loopback.py) - RCCL refuses two ranks on one
// device.  Not a product transport: a job of one process per GPU uses RCCL (th_comm.hip).
//
// An id (th_comm_loopback_id) is 128 bytes like RCCL's: a magic word and a key into the process's table of worlds.
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "th_kernels.hpp"

namespace th {
namespace {

constexpr char kMagic[16] = "TH-LOOPBACK-ID.";

struct Post {                       // what a rank shows the others during one collective
    const void *send = nullptr;
    const size_t *counts = nullptr, *offsets = nullptr;     // (alltoallv: per destination; allgather: unused)
    size_t elem = 0, bytes = 0;
    uint32_t word = 0;
    unsigned long long counters[5] = {0, 0, 0, 0, 0};
    double sum = 0.0, max = 0.0;
};

struct World {
    std::mutex m;
    std::condition_variable cv;
    int size = 0, arrived = 0, joined = 0;
    unsigned long long generation = 0;
    bool broken = false;            // a rank gave up waiting: every later collective fails at once
    std::vector<Post> posts;
    std::vector<bool> taken;
};

struct Member {
    std::shared_ptr<World> w;
    int rank = 0;
};

std::mutex g_table_mutex;
std::map<unsigned long long, std::weak_ptr<World>> g_table;
std::atomic<unsigned long long> g_next{1};

long timeout_ms()
{
    const char *e = getenv("TH_LOOPBACK_TIMEOUT_MS");
    const long v = e ? atol(e) : 0;
    return v > 0 ? v : 120000;
}

// every rank of the world has arrived (0) - or one of them did not within the timeout (1: the world is broken from then on)
int meet(World &w)
{
    std::unique_lock<std::mutex> lock(w.m);
    if (w.broken) return comm_fail("loopback: an earlier collective of this world timed out");
    const unsigned long long gen = w.generation;
    if (++w.arrived == w.size) {
        w.arrived = 0;
        ++w.generation;
        w.cv.notify_all();
        return 0;
    }
    if (!w.cv.wait_for(lock, std::chrono::milliseconds(timeout_ms()), [&] { return w.generation != gen || w.broken; })) {
        w.broken = true;
        w.cv.notify_all();
        return comm_fail("loopback: the other ranks did not arrive at a collective (every rank of an in-process world needs a host thread of its own)");
    }
    return w.broken ? comm_fail("loopback: a rank gave up waiting at a collective") : 0;
}

#define TH_LOOP_HIP(expr)                                                                              \
    do {                                                                                               \
        const hipError_t e_ = (expr);                                                                  \
        if (e_ != hipSuccess) bad = comm_fail(std::string("loopback: " #expr " failed: ") + hipGetErrorString(e_)); \
    } while (0)

// The frame of every collective: finish the stream, post, meet, `copy` (reads the others' posts, enqueues on `stream`),
// finish the stream, meet.  A rank whose own part fails still goes to both meetings, so that nobody is left waiting.
template <typename Copy>
int collective(Member &me, const Post &mine, hipStream_t stream, Copy copy)
{
    World &w = *me.w;
    int bad = 0;
    TH_LOOP_HIP(hipStreamSynchronize(stream));
    { std::lock_guard<std::mutex> lock(w.m); w.posts[(size_t)me.rank] = mine; }
    if (meet(w)) return 1;
    if (!bad) bad = copy(w.posts);
    if (!bad) TH_LOOP_HIP(hipStreamSynchronize(stream));
    const std::string why = bad ? std::string(comm_error()) : std::string();
    if (meet(w)) return 1;
    return bad ? comm_fail(why) : 0;
}

int loop_destroy(void *comm)
{
    Member *me = static_cast<Member *>(comm);
    if (!me) return 0;
    {
        std::lock_guard<std::mutex> lock(me->w->m);
        me->w->taken[(size_t)me->rank] = false;
        --me->w->joined;
    }
    delete me;
    return 0;
}

int loop_allreduce_counters(void *comm, void *counters_dev, hipStream_t stream)
{
    Member &me = *static_cast<Member *>(comm);
    Post mine;
    int bad = 0;
    char host[56];
    TH_LOOP_HIP(hipMemcpyAsync(host, counters_dev, sizeof host, hipMemcpyDeviceToHost, stream));
    TH_LOOP_HIP(hipStreamSynchronize(stream));
    memcpy(mine.counters, host, 40); memcpy(&mine.sum, host + 40, 8); memcpy(&mine.max, host + 48, 8);
    const int rc = collective(me, mine, stream, [&](const std::vector<Post> &posts) {
        unsigned long long c[5] = {0, 0, 0, 0, 0};
        double sum = 0.0, mx = posts[0].max;
        for (const Post &p : posts) {               // (rank order on every rank: the same double on all of them)
            for (int k = 0; k < 5; ++k) c[k] += p.counters[k];
            sum += p.sum;
            mx = p.max > mx ? p.max : mx;
        }
        memcpy(host, c, 40); memcpy(host + 40, &sum, 8); memcpy(host + 48, &mx, 8);
        int bad = 0;
        TH_LOOP_HIP(hipMemcpyAsync(counters_dev, host, sizeof host, hipMemcpyHostToDevice, stream));
        return bad;
    });
    return (bad || rc) ? 1 : 0;
}

int loop_allgather_bytes(void *comm, const void *send, void *recv, const size_t *bytes, const size_t *offset, int rank, int world, hipStream_t stream)
{
    Member &me = *static_cast<Member *>(comm);
    if (rank != me.rank || world != me.w->size) return comm_fail("loopback: rank / world do not match the communicator");
    Post mine;
    mine.send = send; mine.bytes = bytes[rank];
    return collective(me, mine, stream, [&](const std::vector<Post> &posts) {
        int bad = 0;
        for (int r = 0; r < world && !bad; ++r) {
            if (posts[(size_t)r].bytes != bytes[r]) return comm_fail("loopback: rank " + std::to_string(r) + " gathers " + std::to_string(posts[(size_t)r].bytes) + " bytes where rank " + std::to_string(rank) + " expects " + std::to_string(bytes[r]));
            char *at = static_cast<char *>(recv) + offset[r];
            if (bytes[r] && posts[(size_t)r].send != at) TH_LOOP_HIP(hipMemcpyAsync(at, posts[(size_t)r].send, bytes[r], hipMemcpyDefault, stream));
        }
        return bad;
    });
}

int loop_alltoallv(void *comm, const void *send, const size_t *send_counts, const size_t *send_off, void *recv, const size_t *recv_counts,
                   const size_t *recv_off, size_t elem, int world, hipStream_t stream)
{
    Member &me = *static_cast<Member *>(comm);
    if (world != me.w->size) return comm_fail("loopback: world does not match the communicator");
    Post mine;
    mine.send = send; mine.counts = send_counts; mine.offsets = send_off; mine.elem = elem;
    return collective(me, mine, stream, [&](const std::vector<Post> &posts) {
        int bad = 0;
        for (int r = 0; r < world && !bad; ++r) {
            const Post &p = posts[(size_t)r];
            if (p.elem != elem || p.counts[me.rank] != recv_counts[r])
                return comm_fail("loopback: rank " + std::to_string(r) + " sends " + std::to_string(p.counts[me.rank]) + " x " + std::to_string(p.elem) + " bytes where rank " + std::to_string(me.rank) + " receives " + std::to_string(recv_counts[r]) + " x " + std::to_string(elem));
            if (recv_counts[r])
                TH_LOOP_HIP(hipMemcpyAsync(static_cast<char *>(recv) + recv_off[r] * elem, static_cast<const char *>(p.send) + p.offsets[me.rank] * elem,
                                           recv_counts[r] * elem, hipMemcpyDefault, stream));
        }
        return bad;
    });
}

int loop_allreduce_max_u32(void *comm, uint32_t *word_dev, hipStream_t stream)
{
    Member &me = *static_cast<Member *>(comm);
    Post mine;
    int bad = 0;
    TH_LOOP_HIP(hipMemcpyAsync(&mine.word, word_dev, sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
    TH_LOOP_HIP(hipStreamSynchronize(stream));
    uint32_t worst = 0;
    const int rc = collective(me, mine, stream, [&](const std::vector<Post> &posts) {
        for (const Post &p : posts) worst = p.word > worst ? p.word : worst;
        int bad = 0;
        TH_LOOP_HIP(hipMemcpyAsync(word_dev, &worst, sizeof worst, hipMemcpyHostToDevice, stream));
        return bad;
    });
    return (bad || rc) ? 1 : 0;
}

const Transport kLoopback = {"loopback", loop_destroy, loop_allreduce_counters, loop_allgather_bytes, loop_alltoallv, loop_allreduce_max_u32};

}  // namespace

int loopback_unique_id(void *out, size_t bytes)
{
    if (bytes < sizeof kMagic + sizeof(unsigned long long)) return comm_fail("a communicator id is at least 24 bytes");
    memset(out, 0, bytes);
    memcpy(out, kMagic, sizeof kMagic);
    const unsigned long long key = g_next.fetch_add(1);
    memcpy(static_cast<char *>(out) + sizeof kMagic, &key, sizeof key);
    return 0;
}

bool loopback_id(const void *id_bytes) { return id_bytes && memcmp(id_bytes, kMagic, sizeof kMagic) == 0; }

int loopback_init(void **comm, const Transport **transport, const void *id_bytes, size_t bytes, int rank, int world)
{
    if (bytes < sizeof kMagic + sizeof(unsigned long long) || !loopback_id(id_bytes)) return comm_fail("not an in-process communicator id");
    unsigned long long key = 0;
    memcpy(&key, static_cast<const char *>(id_bytes) + sizeof kMagic, sizeof key);
    std::shared_ptr<World> w;
    {
        std::lock_guard<std::mutex> lock(g_table_mutex);
        w = g_table[key].lock();
        if (!w) {
            w = std::make_shared<World>();
            w->size = world;
            w->posts.resize((size_t)world);
            w->taken.assign((size_t)world, false);
            g_table[key] = w;
        }
    }
    std::lock_guard<std::mutex> lock(w->m);
    if (w->size != world) return comm_fail("loopback: this world has " + std::to_string(w->size) + " ranks, not " + std::to_string(world));
    if (w->taken[(size_t)rank]) return comm_fail("loopback: rank " + std::to_string(rank) + " of this world is taken");
    w->taken[(size_t)rank] = true;
    ++w->joined;
    Member *me = new Member;
    me->w = w; me->rank = rank;
    *comm = me;
    *transport = &kLoopback;
    return 0;
}

}  // namespace th
