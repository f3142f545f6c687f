// RCCL slab-neighbour communication behind include/yalla_hip.h (ya_comm_*).
//
// One process per GPU.  RCCL is bound at run time (dlopen) so that libyalla_hip.so has
// no link-time dependency on it: single-GPU programs never load it.  The copy that is bound is
// the one that belongs to the HIP runtime this library itself runs on (the librccl next to the
// loaded libamdhip64): a process may carry a second ROCm stack (PyTorch ships its own HIP, HSA
// and RCCL), whichever was loaded first serves the `libamdhip64.so.7` soname for everyone, and
// an RCCL of the other stack on that runtime fails in ncclCommInitRank ("unhandled cuda error":
// seen when this library was loaded before `import torch`).
#include <hip/hip_runtime.h>

#include <arpa/inet.h>
#include <dlfcn.h>
#include <errno.h>
#include <netdb.h>
#include <netinet/in.h>
#include <poll.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/socket.h>
#include <unistd.h>

#include <chrono>
#include <condition_variable>
#include <mutex>
#include <string>
#include <vector>

#include "yalla_hip.h"

namespace {

// the part of rccl.h this file needs (ABI-stable NCCL 2 declarations)
typedef struct ncclComm* ncclComm_t;
typedef struct {
    char internal[YA_COMM_ID_BYTES];
} ncclUniqueId;
typedef int ncclResult_t;
enum { NCCL_INT8 = 0, NCCL_FLOAT32 = 7, NCCL_FLOAT64 = 8 };
enum { NCCL_SUM = 0, NCCL_MAX = 2 };

struct Rccl {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*Send)(const void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*CommUserRank)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*CommCuDevice)(const ncclComm_t, int*) = nullptr;
};

Rccl* load_rccl();
Rccl* rccl()
{
    // loaded once, by whichever thread asks first (function-local static: thread-safe)
    static Rccl* const loaded = load_rccl();
    return loaded;
}

Rccl* load_rccl()
{
    static Rccl r;
    // first choice: beside the HIP runtime in use (by path, so that an RCCL of another stack
    // that is already loaded under the same soname is not picked up instead)
    Dl_info hip{};
    if (dladdr((void*)&hipGetDeviceCount, &hip) && hip.dli_fname) {
        const char* slash = strrchr(hip.dli_fname, '/');
        if (slash) {
            const std::string dir(hip.dli_fname, (size_t)(slash - hip.dli_fname + 1));
            for (const char* name : {"librccl.so.1", "librccl.so"}) {
                r.lib = dlopen((dir + name).c_str(), RTLD_NOW | RTLD_LOCAL);
                if (r.lib) break;
            }
        }
    }
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* name : names) {
        if (r.lib) break;
        r.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
    }
    if (!r.lib) {
        fprintf(stderr, "yalla-hip: cannot load RCCL (librccl.so.1): %s\n", dlerror());
        return nullptr;
    }
#define YA_SYM(field, name)                                                      \
    *(void**)(&r.field) = dlsym(r.lib, name);                                    \
    if (!r.field) {                                                              \
        fprintf(stderr, "yalla-hip: RCCL lacks %s\n", name);                     \
        r.lib = nullptr;                                                         \
        return nullptr;                                                          \
    }
    YA_SYM(GetUniqueId, "ncclGetUniqueId")
    YA_SYM(CommInitRank, "ncclCommInitRank")
    YA_SYM(CommDestroy, "ncclCommDestroy")
    YA_SYM(Send, "ncclSend")
    YA_SYM(Recv, "ncclRecv")
    YA_SYM(AllReduce, "ncclAllReduce")
    YA_SYM(GroupStart, "ncclGroupStart")
    YA_SYM(GroupEnd, "ncclGroupEnd")
    YA_SYM(GetErrorString, "ncclGetErrorString")
    YA_SYM(CommCount, "ncclCommCount")
    YA_SYM(CommUserRank, "ncclCommUserRank")
    YA_SYM(CommCuDevice, "ncclCommCuDevice")
#undef YA_SYM
    return &r;
}

int rccl_error(const char* what, ncclResult_t code)
{
    Rccl* r = rccl();
    fprintf(stderr, "yalla-hip: %s failed: %s\n", what, r ? r->GetErrorString(code) : "RCCL not loaded");
    return 1000 + (int)code;
}

#define YA_RCCL(call, what)                       \
    do {                                          \
        const ncclResult_t rc_ = (call);          \
        if (rc_ != 0) return rccl_error(what, rc_); \
    } while (0)
// inside ncclGroupStart / ncclGroupEnd: a failed call must still close the group, or the
// communicator stays unusable
#define YA_RCCL_IN_GROUP(call, what)              \
    do {                                          \
        const ncclResult_t rc_ = (call);          \
        if (rc_ != 0) {                           \
            (void)r->GroupEnd();                  \
            return rccl_error(what, rc_);         \
        }                                         \
    } while (0)

int send_all(int fd, const void* data, size_t bytes)
{
    const char* p = (const char*)data;
    while (bytes > 0) {
        const ssize_t k = send(fd, p, bytes, 0);
        if (k <= 0) return -1;
        p += k;
        bytes -= (size_t)k;
    }
    return 0;
}

int recv_all(int fd, void* data, size_t bytes)
{
    char* p = (char*)data;
    while (bytes > 0) {
        const ssize_t k = recv(fd, p, bytes, 0);
        if (k <= 0) return -1;
        p += k;
        bytes -= (size_t)k;
    }
    return 0;
}

}  // namespace

// ---- loopback communicators: the slabs of ONE process, on one GPU -----------------------------
// ya_comm_create_loopback makes `world` communicators whose messages are device-to-device copies and
// whose all-reduce is a kernel -- all of it STREAM-ORDERED like RCCL's: a call returns once its
// work is queued, a receive lands when the sender's stream has got to its send, a send completes
// (in stream order) when the receiver has taken the data.  Every communicator is driven by its own
// host thread, as a rank's process would drive RCCL; the threads only rendezvous to hand each other
// the events to wait on.  What it is for: the decomposed step's asynchronous choreography
// (Slab_grid_solver::stage_exchange: event -> exchange on the communication stream beside the
// interior launch -> event -> join) needs a peer to run against, RCCL refuses two ranks on one GPU,
// and the boxes of this build have one GPU.  A test transport; nothing in it is specific to slabs.
struct Loop_group {
    static constexpr int RING = 8;
    struct Post {  // what a rank hands its peers in one phase of one operation
        const void* send_lo = nullptr;
        const void* send_hi = nullptr;
        size_t send_lo_bytes = 0, send_hi_bytes = 0;
        hipEvent_t event = nullptr;  // recorded in the poster's stream at that point
        double host_values[64];
    };
    struct Mail {
        long tag = -1;  // the highest phase posted so far (phases are numbered alike on every rank)
        Post ring[RING];
        hipEvent_t events[RING] = {};
        // all-reduce: "I have read every slot of this parity" (of my latest all-reduce of that parity)
        hipEvent_t read_event[2] = {};
        bool read_valid[2] = {false, false};
    };
    std::mutex m;
    std::condition_variable cv;
    int world = 0, alive = 0;
    std::vector<Mail> mail;
    float* d_slots = nullptr;  // [2][world][SLOT_FLOATS]: all-reduce contributions, double-buffered
    static constexpr int SLOT_FLOATS = 64;

    // rank posts `post` under the next tag and waits until every peer in [peer_lo, peer_hi] has posted
    // that tag too; returns the tag, or -1 if a peer has not shown up within ten minutes (its thread died:
    // a test must fail, not hang)
    long post_and_wait(int rank, long tag, const Post& post, int peer_lo, int peer_hi)
    {
        std::unique_lock<std::mutex> lock(m);
        Mail& mine = mail[rank];
        const hipEvent_t keep = mine.ring[tag % RING].event;
        mine.ring[tag % RING] = post;
        if (!post.event) mine.ring[tag % RING].event = keep;
        mine.tag = tag;
        cv.notify_all();
        const bool met = cv.wait_for(lock, std::chrono::minutes(10), [&] {
            for (int p = peer_lo; p <= peer_hi; p++)
                if (p != rank && p >= 0 && p < world && mail[p].tag < tag) return false;
            return true;
        });
        return met ? tag : -1;
    }
    Post peek(int peer, long tag)
    {
        std::lock_guard<std::mutex> lock(m);
        return mail[peer].ring[tag % RING];
    }
    hipEvent_t event_for(int rank, long tag)  // this rank's reusable event of that slot of the ring
    {
        std::lock_guard<std::mutex> lock(m);
        hipEvent_t& e = mail[rank].events[tag % RING];
        if (!e) (void)hipEventCreateWithFlags(&e, hipEventDisableTiming);
        return e;
    }
};

__global__ void k_loop_allreduce(float* __restrict__ buf, const float* __restrict__ slots, int world, int count,
    int slot_floats)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= count) return;
    float sum = slots[k];  // rank 0, then the others in rank order: the same bits on every rank
    for (int r = 1; r < world; r++) sum = sum + slots[(size_t)r * slot_floats + k];
    buf[k] = sum;
}

struct ya_comm {
    int rank = 0, world = 1;
    ncclComm_t comm = nullptr;
    double* d_bounce = nullptr;  // 64 doubles for ya_comm_allreduce_host
    Loop_group* loop = nullptr;  // a loopback communicator (ya_comm_create_loopback)
    long phase = 0;              // loopback: phases posted so far
    long allreduces = 0;         // loopback: all-reduces begun so far
};

namespace {
int loop_exchange(ya_comm* c, const void* send_lo, size_t send_lo_bytes, void* recv_lo, size_t recv_lo_bytes,
    const void* send_hi, size_t send_hi_bytes, void* recv_hi, size_t recv_hi_bytes, hipStream_t st)
{
    Loop_group& g = *c->loop;
    const int lo = c->rank - 1, hi = c->rank + 1;
    // phase A: "my send buffers are ready once my stream has got here"
    Loop_group::Post mine;
    mine.send_lo = send_lo;
    mine.send_hi = send_hi;
    mine.send_lo_bytes = send_lo_bytes;
    mine.send_hi_bytes = send_hi_bytes;
    const long tag_a = c->phase++;
    mine.event = g.event_for(c->rank, tag_a);
    if (hipEventRecord(mine.event, st) != hipSuccess) return 996;
    if (g.post_and_wait(c->rank, tag_a, mine, lo, hi) < 0) return 994;
    // the receives: behind the senders' streams, into my buffers, on my stream
    if (lo >= 0 && recv_lo_bytes) {
        const Loop_group::Post from = g.peek(lo, tag_a);
        if (from.send_hi_bytes != recv_lo_bytes) return 995;  // the two ends disagree about a message's size
        if (hipStreamWaitEvent(st, from.event, 0) != hipSuccess ||
            hipMemcpyAsync(recv_lo, from.send_hi, recv_lo_bytes, hipMemcpyDeviceToDevice, st) != hipSuccess)
            return 996;
    }
    if (hi < c->world && recv_hi_bytes) {
        const Loop_group::Post from = g.peek(hi, tag_a);
        if (from.send_lo_bytes != recv_hi_bytes) return 995;
        if (hipStreamWaitEvent(st, from.event, 0) != hipSuccess ||
            hipMemcpyAsync(recv_hi, from.send_lo, recv_hi_bytes, hipMemcpyDeviceToDevice, st) != hipSuccess)
            return 996;
    }
    // phase B: "I have taken what you sent once my stream has got here" -- a send completes, in the
    // sender's stream order, when its receiver has the data (the sender may then reuse the buffer)
    Loop_group::Post taken;
    const long tag_b = c->phase++;
    taken.event = g.event_for(c->rank, tag_b);
    if (hipEventRecord(taken.event, st) != hipSuccess) return 996;
    if (g.post_and_wait(c->rank, tag_b, taken, lo, hi) < 0) return 994;
    if (lo >= 0 && send_lo_bytes && hipStreamWaitEvent(st, g.peek(lo, tag_b).event, 0) != hipSuccess) return 996;
    if (hi < c->world && send_hi_bytes && hipStreamWaitEvent(st, g.peek(hi, tag_b).event, 0) != hipSuccess) return 996;
    return 0;
}

int loop_allreduce(ya_comm* c, float* d_buf, int count, hipStream_t st)
{
    Loop_group& g = *c->loop;
    if (count > Loop_group::SLOT_FLOATS) return (int)hipErrorInvalidValue;
    const int parity = (int)(c->allreduces++ & 1);
    float* slots = g.d_slots + (size_t)parity * g.world * Loop_group::SLOT_FLOATS;
    // My slot of this parity was read by everybody in the all-reduce before last: wait for those reads.
    // (Every rank has queued them by now: it passed the last all-reduce's rendezvous after that.)
    for (int p = 0; p < g.world; p++) {
        hipEvent_t e = nullptr;
        {
            std::lock_guard<std::mutex> lock(g.m);
            if (g.mail[p].read_valid[parity]) e = g.mail[p].read_event[parity];
        }
        if (p != c->rank && e && hipStreamWaitEvent(st, e, 0) != hipSuccess) return 996;
    }
    if (hipMemcpyAsync(slots + (size_t)c->rank * Loop_group::SLOT_FLOATS, d_buf, (size_t)count * sizeof(float),
            hipMemcpyDeviceToDevice, st) != hipSuccess)
        return 996;
    Loop_group::Post mine;
    const long tag = c->phase++;
    mine.event = g.event_for(c->rank, tag);
    if (hipEventRecord(mine.event, st) != hipSuccess) return 996;
    if (g.post_and_wait(c->rank, tag, mine, 0, g.world - 1) < 0) return 994;
    for (int p = 0; p < g.world; p++)
        if (p != c->rank && hipStreamWaitEvent(st, g.peek(p, tag).event, 0) != hipSuccess) return 996;
    k_loop_allreduce<<<1, 64, 0, st>>>(d_buf, slots, g.world, count, Loop_group::SLOT_FLOATS);
    {
        std::lock_guard<std::mutex> lock(g.m);
        Loop_group::Mail& me = g.mail[c->rank];
        if (!me.read_event[parity]) (void)hipEventCreateWithFlags(&me.read_event[parity], hipEventDisableTiming);
        if (hipEventRecord(me.read_event[parity], st) != hipSuccess) return 996;
        me.read_valid[parity] = true;
    }
    return 0;
}
}  // namespace

extern "C" {

int ya_comm_unique_id(void* id_out)
{
    Rccl* r = rccl();
    if (!r || !id_out) return 999;
    ncclUniqueId id;
    YA_RCCL(r->GetUniqueId(&id), "ncclGetUniqueId");
    memcpy(id_out, &id, sizeof(id));
    return 0;
}

int ya_comm_create(const void* id_bytes, int rank, int world, ya_comm** out)
{
    if (!out || world < 1 || rank < 0 || rank >= world) return (int)hipErrorInvalidValue;
    ya_comm* c = new ya_comm;
    c->rank = rank;
    c->world = world;
    // world 1 with an id: a real one-rank RCCL communicator (tests: the binding exercised on a
    // one-GPU box); world 1 without: no RCCL at all
    if (world > 1 || id_bytes) {
        Rccl* r = rccl();
        if (!r || !id_bytes) {
            delete c;
            return 999;
        }
        ncclUniqueId id;
        memcpy(&id, id_bytes, sizeof(id));
        const ncclResult_t rc = r->CommInitRank(&c->comm, world, id, rank);
        if (rc != 0) {
            delete c;
            return rccl_error("ncclCommInitRank", rc);
        }
        if (hipMalloc(&c->d_bounce, 64 * sizeof(double)) != hipSuccess) {
            (void)r->CommDestroy(c->comm);
            delete c;
            return (int)hipErrorOutOfMemory;
        }
    }
    *out = c;
    return 0;
}

int ya_comm_create_from_env(int port_offset, ya_comm** out)
{
    const char* s_rank = getenv("RANK");
    const char* s_world = getenv("WORLD_SIZE");
    const int rank = s_rank ? atoi(s_rank) : 0;
    const int world = s_world ? atoi(s_world) : 1;
    if (world <= 1) return ya_comm_create(nullptr, 0, 1, out);
    // One process per GPU: unless the program has already chosen its device (YALLA_KEEP_DEVICE=1,
    // or it called hipSetDevice to something other than 0), rank r of a node takes GPU
    // LOCAL_RANK (torchrun, bench.py's launcher) -- else every rank would sit on device 0 and
    // ncclCommInitRank refuses two ranks on one GPU.
    {
        int current = 0, count = 0;
        (void)hipGetDevice(&current);
        (void)hipGetDeviceCount(&count);
        const char* s_local = getenv("LOCAL_RANK");
        const char* keep = getenv("YALLA_KEEP_DEVICE");
        if (!(keep && atoi(keep)) && current == 0 && count > 1) {
            const int local = s_local ? atoi(s_local) : rank;
            if (hipSetDevice(local % count) != hipSuccess) {
                fprintf(stderr, "yalla-hip: rank %d cannot select GPU %d of %d\n", rank, local % count, count);
                return 997;
            }
        } else if (count == 1 && !(keep && atoi(keep))) {
            fprintf(stderr, "yalla-hip: rank %d of %d sees one GPU only: RCCL needs one GPU per rank "
                            "(set YALLA_KEEP_DEVICE=1 to try anyway)\n", rank, world);
        }
    }
    const char* addr = getenv("MASTER_ADDR") ? getenv("MASTER_ADDR") : "127.0.0.1";
    const int port = (getenv("MASTER_PORT") ? atoi(getenv("MASTER_PORT")) : 29500) + port_offset;
    char id[YA_COMM_ID_BYTES];
    if (rank == 0) {
        const int rc = ya_comm_unique_id(id);
        if (rc) return rc;
        const int srv = socket(AF_INET, SOCK_STREAM, 0);
        int yes = 1;
        setsockopt(srv, SOL_SOCKET, SO_REUSEADDR, &yes, sizeof(yes));
        sockaddr_in sa{};
        sa.sin_family = AF_INET;
        sa.sin_addr.s_addr = htonl(INADDR_ANY);
        sa.sin_port = htons((unsigned short)port);
        if (srv < 0 || bind(srv, (sockaddr*)&sa, sizeof(sa)) != 0 || listen(srv, world) != 0) {
            fprintf(stderr, "yalla-hip: rank 0 cannot listen on port %d: %s\n", port, strerror(errno));
            if (srv >= 0) close(srv);
            return 998;
        }
        for (int k = 1; k < world; k++) {
            // a rank that died before connecting must not hang rank 0 for ever: ten minutes
            // (the first import of a big runtime on a fresh box can take two), then give up
            pollfd waiting{srv, POLLIN, 0};
            const int ready = poll(&waiting, 1, 600 * 1000);
            if (ready <= 0) {
                fprintf(stderr, "yalla-hip: rank 0 waited ten minutes for rank connection %d of %d on port %d\n",
                    k, world - 1, port);
                close(srv);
                return 998;
            }
            const int fd = accept(srv, nullptr, nullptr);
            if (fd < 0 || send_all(fd, id, sizeof(id)) != 0) {
                fprintf(stderr, "yalla-hip: rank 0 could not hand the id to a rank: %s\n", strerror(errno));
                if (fd >= 0) close(fd);
                close(srv);
                return 998;
            }
            close(fd);
        }
        close(srv);
    } else {
        addrinfo hints{}, *res = nullptr;
        hints.ai_family = AF_INET;
        hints.ai_socktype = SOCK_STREAM;
        char port_s[16];
        snprintf(port_s, sizeof(port_s), "%d", port);
        if (getaddrinfo(addr, port_s, &hints, &res) != 0 || !res) {
            fprintf(stderr, "yalla-hip: cannot resolve MASTER_ADDR %s\n", addr);
            return 998;
        }
        int fd = -1;
        for (int attempt = 0; attempt < 6000; attempt++) {  // rank 0 may not listen yet (first import of a big runtime): up to ten minutes
            fd = socket(AF_INET, SOCK_STREAM, 0);
            if (fd >= 0 && connect(fd, res->ai_addr, res->ai_addrlen) == 0) break;
            if (fd >= 0) close(fd);
            fd = -1;
            usleep(100000);
        }
        freeaddrinfo(res);
        if (fd < 0 || recv_all(fd, id, sizeof(id)) != 0) {
            fprintf(stderr, "yalla-hip: rank %d got no id from %s:%d\n", rank, addr, port);
            if (fd >= 0) close(fd);
            return 998;
        }
        close(fd);
    }
    return ya_comm_create(id, rank, world, out);
}

int ya_comm_create_loopback(int world, ya_comm** out)
{
    if (!out || world < 1 || world > 64) return (int)hipErrorInvalidValue;
    Loop_group* g = new Loop_group;
    g->world = g->alive = world;
    g->mail.resize(world);
    if (hipMalloc(&g->d_slots, 2 * (size_t)world * Loop_group::SLOT_FLOATS * sizeof(float)) != hipSuccess) {
        delete g;
        return (int)hipErrorOutOfMemory;
    }
    for (int r = 0; r < world; r++) {
        ya_comm* c = new ya_comm;
        c->rank = r;
        c->world = world;
        c->loop = g;
        out[r] = c;
    }
    return 0;
}

int ya_comm_destroy(ya_comm* c)
{
    if (!c) return 0;
    if (c->loop) {
        Loop_group* g = c->loop;
        bool last;
        {
            std::lock_guard<std::mutex> lock(g->m);
            last = --g->alive == 0;
        }
        if (last) {
            (void)hipDeviceSynchronize();
            for (auto& mail : g->mail) {
                for (hipEvent_t e : mail.events)
                    if (e) (void)hipEventDestroy(e);
                for (hipEvent_t e : mail.read_event)
                    if (e) (void)hipEventDestroy(e);
            }
            (void)hipFree(g->d_slots);
            delete g;
        }
        delete c;
        return 0;
    }
    if (c->comm) {
        Rccl* r = rccl();
        if (r) (void)r->CommDestroy(c->comm);
    }
    if (c->d_bounce) (void)hipFree(c->d_bounce);
    delete c;
    return 0;
}

int ya_comm_info(const ya_comm* c, int info[8], char pci_bus_id[32])
{
    if (!c || !info) return (int)hipErrorInvalidValue;
    for (int k = 0; k < 8; k++) info[k] = 0;
    int current = 0;
    if (hipGetDevice(&current) != hipSuccess) return (int)hipGetLastError();
    info[0] = c->world, info[1] = c->rank, info[2] = current, info[3] = current;
    info[4] = c->loop ? 2 : (c->comm ? 1 : 0);
    if (c->comm) {  // RCCL's own account
        Rccl* r = rccl();
        if (!r) return 999;
        YA_RCCL(r->CommCount(c->comm, &info[0]), "ncclCommCount");
        YA_RCCL(r->CommUserRank(c->comm, &info[1]), "ncclCommUserRank");
        YA_RCCL(r->CommCuDevice(c->comm, &info[2]), "ncclCommCuDevice");
    }
    char bus[32] = {0};
    if (hipDeviceGetPCIBusId(bus, (int)sizeof(bus), info[2]) != hipSuccess) return (int)hipGetLastError();
    unsigned domain = 0, b = 0, d = 0, f = 0;
    if (sscanf(bus, "%x:%x:%x.%x", &domain, &b, &d, &f) == 4) info[5] = (int)((domain << 16) | (b << 8) | (d << 3) | f);
    if (pci_bus_id) {
        strncpy(pci_bus_id, bus, 31);
        pci_bus_id[31] = 0;
    }
    return 0;
}

int ya_comm_rank(const ya_comm* c) { return c ? c->rank : 0; }
int ya_comm_world(const ya_comm* c) { return c ? c->world : 1; }

int ya_comm_exchange(ya_comm* c, const void* d_send_lo, void* d_recv_lo, const void* d_send_hi,
    void* d_recv_hi, size_t bytes, void* stream)
{
    if (!c) return (int)hipErrorInvalidValue;
    if (c->world == 1 || bytes == 0) return 0;
    if (c->loop) {
        const bool has_lo = c->rank > 0, has_hi = c->rank + 1 < c->world;
        return loop_exchange(c, d_send_lo, has_lo ? bytes : 0, d_recv_lo, has_lo ? bytes : 0, d_send_hi, has_hi ? bytes : 0,
            d_recv_hi, has_hi ? bytes : 0, (hipStream_t)stream);
    }
    Rccl* r = rccl();
    if (!r) return 999;
    hipStream_t st = (hipStream_t)stream;
    const bool lo = c->rank > 0, hi = c->rank + 1 < c->world;
    if ((lo && (!d_send_lo || !d_recv_lo)) || (hi && (!d_send_hi || !d_recv_hi)))
        return (int)hipErrorInvalidValue;
    YA_RCCL(r->GroupStart(), "ncclGroupStart");
    if (lo) {
        YA_RCCL_IN_GROUP(r->Send(d_send_lo, bytes, NCCL_INT8, c->rank - 1, c->comm, st), "ncclSend (lower slab)");
        YA_RCCL_IN_GROUP(r->Recv(d_recv_lo, bytes, NCCL_INT8, c->rank - 1, c->comm, st), "ncclRecv (lower slab)");
    }
    if (hi) {
        YA_RCCL_IN_GROUP(r->Send(d_send_hi, bytes, NCCL_INT8, c->rank + 1, c->comm, st), "ncclSend (upper slab)");
        YA_RCCL_IN_GROUP(r->Recv(d_recv_hi, bytes, NCCL_INT8, c->rank + 1, c->comm, st), "ncclRecv (upper slab)");
    }
    YA_RCCL(r->GroupEnd(), "ncclGroupEnd");
    return 0;
}

int ya_comm_exchange_v(ya_comm* c, const void* d_send_lo, size_t send_lo_bytes, void* d_recv_lo,
    size_t recv_lo_bytes, const void* d_send_hi, size_t send_hi_bytes, void* d_recv_hi, size_t recv_hi_bytes,
    void* stream)
{
    if (!c) return (int)hipErrorInvalidValue;
    if (c->world == 1) return 0;
    if (c->loop)
        return loop_exchange(c, d_send_lo, c->rank > 0 ? send_lo_bytes : 0, d_recv_lo, c->rank > 0 ? recv_lo_bytes : 0,
            d_send_hi, c->rank + 1 < c->world ? send_hi_bytes : 0, d_recv_hi, c->rank + 1 < c->world ? recv_hi_bytes : 0,
            (hipStream_t)stream);
    Rccl* r = rccl();
    if (!r) return 999;
    hipStream_t st = (hipStream_t)stream;
    const bool lo = c->rank > 0, hi = c->rank + 1 < c->world;
    if ((lo && ((send_lo_bytes && !d_send_lo) || (recv_lo_bytes && !d_recv_lo))) ||
        (hi && ((send_hi_bytes && !d_send_hi) || (recv_hi_bytes && !d_recv_hi))))
        return (int)hipErrorInvalidValue;
    YA_RCCL(r->GroupStart(), "ncclGroupStart");
    if (lo && send_lo_bytes)
        YA_RCCL_IN_GROUP(r->Send(d_send_lo, send_lo_bytes, NCCL_INT8, c->rank - 1, c->comm, st), "ncclSend (lower slab)");
    if (lo && recv_lo_bytes)
        YA_RCCL_IN_GROUP(r->Recv(d_recv_lo, recv_lo_bytes, NCCL_INT8, c->rank - 1, c->comm, st), "ncclRecv (lower slab)");
    if (hi && send_hi_bytes)
        YA_RCCL_IN_GROUP(r->Send(d_send_hi, send_hi_bytes, NCCL_INT8, c->rank + 1, c->comm, st), "ncclSend (upper slab)");
    if (hi && recv_hi_bytes)
        YA_RCCL_IN_GROUP(r->Recv(d_recv_hi, recv_hi_bytes, NCCL_INT8, c->rank + 1, c->comm, st), "ncclRecv (upper slab)");
    YA_RCCL(r->GroupEnd(), "ncclGroupEnd");
    return 0;
}

int ya_comm_self_exchange(ya_comm* c, const void* d_send, void* d_recv, size_t bytes, void* stream)
{
    if (!c || !c->comm || !d_send || !d_recv) return (int)hipErrorInvalidValue;
    Rccl* r = rccl();
    if (!r) return 999;
    hipStream_t st = (hipStream_t)stream;
    YA_RCCL(r->GroupStart(), "ncclGroupStart");
    YA_RCCL_IN_GROUP(r->Send(d_send, bytes, NCCL_INT8, c->rank, c->comm, st), "ncclSend (self)");
    YA_RCCL_IN_GROUP(r->Recv(d_recv, bytes, NCCL_INT8, c->rank, c->comm, st), "ncclRecv (self)");
    YA_RCCL(r->GroupEnd(), "ncclGroupEnd");
    return 0;
}

int ya_comm_allreduce_sum(ya_comm* c, float* d_buf, int count, void* stream)
{
    if (!c || !d_buf || count < 0) return (int)hipErrorInvalidValue;
    if (c->loop) return c->world == 1 || count == 0 ? 0 : loop_allreduce(c, d_buf, count, (hipStream_t)stream);
    if (!c->comm || count == 0) return 0;
    Rccl* r = rccl();
    if (!r) return 999;
    YA_RCCL(r->AllReduce(d_buf, d_buf, (size_t)count, NCCL_FLOAT32, NCCL_SUM, c->comm, (hipStream_t)stream),
        "ncclAllReduce");
    return 0;
}

int ya_comm_allreduce_host(ya_comm* c, double* values, int count, int take_max)
{
    if (!c || !values || count < 0 || count > 64) return (int)hipErrorInvalidValue;
    if (c->loop) {  // host values: through the rendezvous itself
        if (c->world == 1 || count == 0) return 0;
        Loop_group& g = *c->loop;
        Loop_group::Post mine;
        for (int k = 0; k < count; k++) mine.host_values[k] = values[k];
        const long tag = c->phase++;
        if (g.post_and_wait(c->rank, tag, mine, 0, g.world - 1) < 0) return 994;
        for (int k = 0; k < count; k++) {
            double acc = g.peek(0, tag).host_values[k];
            for (int p = 1; p < g.world; p++) {
                const double v = g.peek(p, tag).host_values[k];
                acc = take_max ? (v > acc ? v : acc) : acc + v;
            }
            values[k] = acc;
        }
        // (nobody's post of this tag is overwritten before everybody has read it: the ring holds 8 phases
        // and the next operation's rendezvous comes first)
        return g.post_and_wait(c->rank, c->phase++, Loop_group::Post{}, 0, g.world - 1) < 0 ? 994 : 0;
    }
    if (!c->comm || count == 0) return 0;
    Rccl* r = rccl();
    if (!r) return 999;
    if (hipMemcpy(c->d_bounce, values, (size_t)count * sizeof(double), hipMemcpyHostToDevice) != hipSuccess)
        return (int)hipGetLastError();
    YA_RCCL(r->AllReduce(c->d_bounce, c->d_bounce, (size_t)count, NCCL_FLOAT64, take_max ? NCCL_MAX : NCCL_SUM,
                c->comm, nullptr),
        "ncclAllReduce (host values)");
    if (hipMemcpy(values, c->d_bounce, (size_t)count * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess)
        return (int)hipGetLastError();
    return 0;
}

}  // extern "C"
