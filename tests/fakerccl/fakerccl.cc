// libfakerccl.so -- TEST INFRASTRUCTURE ONLY: a stand-in for the eight librccl entry points libcerebro_hip.so binds
// (cerebro_amd/csrc/chip_multi.hip, struct Rccl), selected through the library's existing dlopen override
// (CHIP_RCCL_LIBRARY=<this file>).  Real RCCL refuses two ranks on one device, so on a 1-GPU box the one-process-per-GPU exchange
// (chip_comm_init_rank + ncclAllGather / ncclBroadcast sequencing at world 2 / 4, failure marks, agreement rounds, owner fetches)
// could never execute; with this stub N real PROCESSES share device 0 and run exactly that code.  It is a correct (and slow)
// implementation of the collectives' semantics, not a model of their performance:
//   * transport: one POSIX shared-memory segment per communicator, named inside the 128-byte unique id;
//   * a collective is executed synchronously at call time: wait for the caller's stream (its producers), copy the send buffer to
//     the segment, wait until every rank has arrived at the same operation number, copy the gathered data into the receive buffer
//     -- so work enqueued on the stream afterwards sees the result, as with an in-stream collective;
//   * every wait has a deadline (FAKERCCL_TIMEOUT_MS, default 60 s): a rank that never arrives makes its peers return
//     ncclSystemError instead of hanging the test box;
//   * payloads larger than a slot go through in slot-sized pieces, each its own operation;
//   * FAKERCCL_ASYNC=1 (round 6): small collectives (<= 16 KiB per rank: the per-tick lists, the agreement word) are ENQUEUED like the
//     real thing -- D2H copy, a host function on the caller's stream that does the exchange (and blocks that stream, like a collective
//     kernel waiting for its peers, while the rank's other streams run on), H2D copy -- and the call returns at once: the overlap of
//     all-gather(i) with scan(i+1) that the library's tick pipeline is built for is then really exercised.  Operations keep their issue
//     order across streams by an event chain (as RCCL orders them); a peer timeout surfaces as ncclSystemError of the NEXT call;
//   * ncclCommInitAll (round 6): n communicators of ONE process over one segment (the one-process / G-devices layout on one device).
// Nothing under cerebro_amd/ references this file; the product path loads librccl.so.1.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <thread>
#include <unistd.h>

namespace {

constexpr int kMaxRanks = 16;
constexpr size_t kSlotBytes = 256u << 10;        // per rank, per buffer half (segment: 2 x 16 x 256 KiB = 8 MiB of /dev/shm)

struct Shared {                                   // lives in the segment; zero-filled by ftruncate
    std::atomic<uint64_t> arrive[kMaxRanks];      // operations this rank has contributed to
    std::atomic<uint64_t> depart[kMaxRanks];      // operations this rank has finished reading
    std::atomic<int> attached;                    // ranks that have mapped the segment
    std::atomic<int> detached;
    char data[2][kMaxRanks][kSlotBytes];          // double-buffered: operation n uses half n & 1
};

constexpr int kAsyncRing = 64;                    // operations in flight per communicator (the library rings 64 list buffers)
constexpr size_t kAsyncSlot = 16u << 10;          // per rank: payloads up to this size are enqueued, larger ones take the synchronous path

struct Comm;
struct AsyncOp {                                  // one enqueued collective
    Comm *c = nullptr;
    uint64_t n = 0;
    size_t bytes = 0;
    int root = -1;                                // >= 0: broadcast from root
    char *h = nullptr;                            // pinned: this rank's piece on the way out, every rank's pieces on the way in
    hipEvent_t done = nullptr;                    // recorded behind the H2D copy of the slot's previous use
    bool used = false;
};

struct Comm {
    Shared *sh = nullptr;
    int n = 0, rank = 0;
    uint64_t op = 0;                              // operations issued by this rank so far
    char name[64] = {0};
    void *host = nullptr;                         // pinned bounce buffer, kMaxRanks * kSlotBytes
    bool async = false;
    AsyncOp ring[kAsyncRing];
    void *ring_host = nullptr;                    // pinned, kAsyncRing * kMaxRanks * kAsyncSlot
    hipEvent_t last = nullptr;                    // behind the most recent enqueued operation (issue order across streams)
    bool have_last = false;
    std::atomic<int> async_error{0};
    bool shared_mapping = false;                  // ncclCommInitAll: the process's n communicators share ONE mapping (the last one out unmaps it)
    uint64_t n_async = 0;
};

int timeout_ms()
{
    const char *t = std::getenv("FAKERCCL_TIMEOUT_MS");
    const int v = t ? std::atoi(t) : 0;
    return v > 0 ? v : 60000;
}

template <class F> bool wait_until(F &&cond)
{
    const auto t0 = std::chrono::steady_clock::now();
    int spins = 0;
    while (!cond()) {
        if (++spins > 2000) { std::this_thread::sleep_for(std::chrono::microseconds(50)); }
        if ((spins & 1023) == 0 &&
            std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t0).count() > timeout_ms())
            return false;
    }
    return true;
}

// one exchange step: every rank contributes `bytes` (<= kSlotBytes) from host memory; on return `all` (n * bytes) holds every
// rank's contribution in rank order.  contribute == false: this rank sends nothing (broadcast from another root).
ncclResult_t step_n(Comm *c, uint64_t n, const void *mine, size_t bytes, bool contribute, char *all, int only_root)
{
    Shared *sh = c->sh;
    const int half = (int)(n & 1);
    if (n >= 2) {   // the half is reused: every rank must have finished reading operation n - 2
        if (!wait_until([&] { for (int r = 0; r < c->n; r++) if (sh->depart[r].load(std::memory_order_acquire) < n - 1) return false; return true; })) {
            std::fprintf(stderr, "[fakerccl] rank %d: timeout waiting for peers to leave operation %llu\n", c->rank, (unsigned long long)(n - 2));
            return ncclSystemError;
        }
    }
    if (contribute) std::memcpy(sh->data[half][c->rank], mine, bytes);
    sh->arrive[c->rank].store(n + 1, std::memory_order_release);
    if (!wait_until([&] { for (int r = 0; r < c->n; r++) if (sh->arrive[r].load(std::memory_order_acquire) < n + 1) return false; return true; })) {
        std::fprintf(stderr, "[fakerccl] rank %d: timeout waiting for peers to arrive at operation %llu\n", c->rank, (unsigned long long)n);
        return ncclSystemError;
    }
    if (only_root >= 0) std::memcpy(all, sh->data[half][only_root], bytes);
    else for (int r = 0; r < c->n; r++) std::memcpy(all + (size_t)r * bytes, sh->data[half][r], bytes);
    sh->depart[c->rank].store(n + 1, std::memory_order_release);
    return ncclSuccess;
}

ncclResult_t step(Comm *c, const void *mine, size_t bytes, bool contribute, char *all, int only_root)
{
    return step_n(c, c->op++, mine, bytes, contribute, all, only_root);
}

void async_host_fn(void *p)                       // runs on the stream's callback thread: no HIP calls in here
{
    AsyncOp *o = static_cast<AsyncOp *>(p);
    Comm *c = o->c;
    const ncclResult_t r = step_n(c, o->n, o->h, o->bytes, o->root < 0 || c->rank == o->root, o->h, o->root);
    if (r != ncclSuccess) c->async_error.store((int)r);
}

// enqueue one small collective on `stream`.  send == nullptr: this rank contributes nothing (broadcast from another root).
ncclResult_t enqueue(Comm *c, const void *send, void *recv, size_t bytes, int root, hipStream_t stream)
{
    if (const int e = c->async_error.load()) return (ncclResult_t)e;
    const uint64_t n = c->op++;
    AsyncOp &o = c->ring[n % kAsyncRing];
    if (o.used && hipEventSynchronize(o.done) != hipSuccess) return ncclUnhandledCudaError;   // the slot's previous use has landed
    o.c = c; o.n = n; o.bytes = bytes; o.root = root; o.used = true;
    if (c->have_last && hipStreamWaitEvent(stream, c->last, 0) != hipSuccess) return ncclUnhandledCudaError;   // issue order, whatever the stream
    if (send && bytes && hipMemcpyAsync(o.h, send, bytes, hipMemcpyDeviceToHost, stream) != hipSuccess) return ncclUnhandledCudaError;
    if (hipLaunchHostFunc(stream, async_host_fn, &o) != hipSuccess) return ncclUnhandledCudaError;
    if (bytes) {
        if (root >= 0) {
            if (hipMemcpyAsync(recv, o.h, bytes, hipMemcpyHostToDevice, stream) != hipSuccess) return ncclUnhandledCudaError;
        } else if (hipMemcpyAsync(recv, o.h, bytes * (size_t)c->n, hipMemcpyHostToDevice, stream) != hipSuccess) return ncclUnhandledCudaError;
    }
    if (hipEventRecord(o.done, stream) != hipSuccess || hipEventRecord(c->last, stream) != hipSuccess) return ncclUnhandledCudaError;
    c->have_last = true;
    c->n_async++;
    return ncclSuccess;
}

// a synchronous operation behind enqueued ones: they come first (on whatever stream they sit)
ncclResult_t drain(Comm *c)
{
    if (c->have_last && hipEventSynchronize(c->last) != hipSuccess) return ncclUnhandledCudaError;
    if (const int e = c->async_error.load()) return (ncclResult_t)e;
    return ncclSuccess;
}

bool setup_async(Comm *c)
{
    const char *a = std::getenv("FAKERCCL_ASYNC");
    // (communicators of ONE process stay synchronous: their host functions would wait for each other on the runtime's callback thread)
    c->async = a && a[0] == '1' && !c->shared_mapping;
    if (!c->async) return true;
    if (hipHostMalloc(&c->ring_host, (size_t)kAsyncRing * kMaxRanks * kAsyncSlot, hipHostMallocDefault) != hipSuccess) return false;
    for (int i = 0; i < kAsyncRing; i++) {
        c->ring[i].h = static_cast<char *>(c->ring_host) + (size_t)i * kMaxRanks * kAsyncSlot;
        if (hipEventCreateWithFlags(&c->ring[i].done, hipEventDisableTiming) != hipSuccess) return false;
    }
    return hipEventCreateWithFlags(&c->last, hipEventDisableTiming) == hipSuccess;
}

size_t type_bytes(ncclDataType_t t)
{
    switch (t) {
    case ncclInt8: case ncclUint8: return 1;
    case ncclFloat16: case ncclBfloat16: return 2;
    case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
    default: return 8;
    }
}

}  // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId *id)
{
    std::memset(id, 0, sizeof *id);
    uint64_t r = (uint64_t)std::chrono::steady_clock::now().time_since_epoch().count() ^ ((uint64_t)getpid() << 32);
    std::snprintf(id->internal, sizeof id->internal, "/fakerccl_%d_%llx", (int)getpid(), (unsigned long long)r);
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t *comm, int nranks, ncclUniqueId id, int rank)
{
    if (!comm || nranks < 1 || nranks > kMaxRanks || rank < 0 || rank >= nranks) return ncclInvalidArgument;
    Comm *c = new Comm();
    c->n = nranks; c->rank = rank;
    std::memcpy(c->name, id.internal, sizeof c->name - 1);
    const int fd = shm_open(c->name, O_CREAT | O_RDWR, 0600);
    if (fd < 0 || ftruncate(fd, (off_t)sizeof(Shared)) != 0) { if (fd >= 0) close(fd); delete c; return ncclSystemError; }
    void *p = mmap(nullptr, sizeof(Shared), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) { delete c; return ncclSystemError; }
    c->sh = static_cast<Shared *>(p);
    if (hipHostMalloc(&c->host, (size_t)kMaxRanks * kSlotBytes, hipHostMallocDefault) != hipSuccess) { munmap(p, sizeof(Shared)); delete c; return ncclUnhandledCudaError; }
    c->sh->attached.fetch_add(1);
    if (!wait_until([&] { return c->sh->attached.load() >= nranks; })) {     // the rendezvous: every rank has mapped the segment
        std::fprintf(stderr, "[fakerccl] rank %d: rendezvous timeout (%d of %d ranks)\n", rank, c->sh->attached.load(), nranks);
        (void)hipHostFree(c->host); munmap(p, sizeof(Shared)); delete c;
        return ncclSystemError;
    }
    if (!setup_async(c)) return ncclUnhandledCudaError;
    std::fprintf(stderr, "[fakerccl] TEST STUB: communicator of %d ranks over shared memory %s (rank %d%s)\n", nranks, c->name, rank, c->async ? ", collectives enqueued" : "");
    *comm = reinterpret_cast<ncclComm_t>(c);
    return ncclSuccess;
}

// one process, n devices (here: n sub-contexts of one device): n communicators over one segment; each is driven by its own thread
ncclResult_t ncclCommInitAll(ncclComm_t *comms, int ndev, const int *devlist)
{
    if (!comms || ndev < 1 || ndev > kMaxRanks) return ncclInvalidArgument;
    ncclUniqueId id;
    ncclGetUniqueId(&id);
    const int fd = shm_open(id.internal, O_CREAT | O_RDWR, 0600);
    if (fd < 0 || ftruncate(fd, (off_t)sizeof(Shared)) != 0) { if (fd >= 0) close(fd); return ncclSystemError; }
    void *p = mmap(nullptr, sizeof(Shared), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) return ncclSystemError;
    int dev0 = 0;
    (void)hipGetDevice(&dev0);
    for (int r = 0; r < ndev; r++) {
        Comm *c = new Comm();
        c->n = ndev; c->rank = r; c->shared_mapping = true;
        std::memcpy(c->name, id.internal, sizeof c->name - 1);
        c->sh = static_cast<Shared *>(p);
        if (devlist) (void)hipSetDevice(devlist[r]);
        if (hipHostMalloc(&c->host, (size_t)kMaxRanks * kSlotBytes, hipHostMallocDefault) != hipSuccess || !setup_async(c)) return ncclUnhandledCudaError;
        c->sh->attached.fetch_add(1);
        comms[r] = reinterpret_cast<ncclComm_t>(c);
    }
    (void)hipSetDevice(dev0);
    std::fprintf(stderr, "[fakerccl] TEST STUB: %d communicators of one process over shared memory %s\n", ndev, id.internal);
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm)
{
    Comm *c = reinterpret_cast<Comm *>(comm);
    if (!c) return ncclSuccess;
    if (c->have_last) (void)hipEventSynchronize(c->last);
    const bool last_out = c->sh->detached.fetch_add(1) + 1 >= c->n;
    if (last_out) shm_unlink(c->name);                                      // the last rank out removes the name
    if (!c->shared_mapping || last_out) munmap(c->sh, sizeof(Shared));
    (void)hipHostFree(c->host);
    if (c->ring_host) (void)hipHostFree(c->ring_host);
    for (AsyncOp &o : c->ring) if (o.done) (void)hipEventDestroy(o.done);
    if (c->last) (void)hipEventDestroy(c->last);
    delete c;
    return ncclSuccess;
}

ncclResult_t ncclCommCount(const ncclComm_t comm, int *count)
{
    if (!comm || !count) return ncclInvalidArgument;
    *count = reinterpret_cast<const Comm *>(comm)->n;
    return ncclSuccess;
}

ncclResult_t ncclAllGather(const void *send, void *recv, size_t count, ncclDataType_t type, ncclComm_t comm, hipStream_t stream)
{
    Comm *c = reinterpret_cast<Comm *>(comm);
    const size_t bytes = count * type_bytes(type);
    if (c->async && bytes <= kAsyncSlot) return enqueue(c, send, recv, bytes, -1, stream);
    if (c->async) { const ncclResult_t d = drain(c); if (d != ncclSuccess) return d; }
    if (hipStreamSynchronize(stream) != hipSuccess) return ncclUnhandledCudaError;       // the send buffer's producers
    char *h = static_cast<char *>(c->host);       // this rank's piece on the way out, then every rank's pieces on the way in
    size_t off = 0;
    do {
        const size_t piece = bytes - off < kSlotBytes ? bytes - off : kSlotBytes;
        if (piece && hipMemcpy(h, static_cast<const char *>(send) + off, piece, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
        const ncclResult_t r = step(c, h, piece, true, h, -1);
        if (r != ncclSuccess) return r;
        for (int rk = 0; rk < c->n && piece; rk++)   // rank rk's piece belongs at recv + rk * bytes + off
            if (hipMemcpy(static_cast<char *>(recv) + (size_t)rk * bytes + off, h + (size_t)rk * piece, piece, hipMemcpyHostToDevice) != hipSuccess)
                return ncclUnhandledCudaError;
        off += piece;
    } while (off < bytes);
    return ncclSuccess;
}

ncclResult_t ncclBroadcast(const void *send, void *recv, size_t count, ncclDataType_t type, int root, ncclComm_t comm, hipStream_t stream)
{
    Comm *c = reinterpret_cast<Comm *>(comm);
    if (root < 0 || root >= c->n) return ncclInvalidArgument;
    const size_t bytes = count * type_bytes(type);
    if (c->async && bytes <= kAsyncSlot) return enqueue(c, c->rank == root ? send : nullptr, recv, bytes, root, stream);
    if (c->async) { const ncclResult_t d = drain(c); if (d != ncclSuccess) return d; }
    if (hipStreamSynchronize(stream) != hipSuccess) return ncclUnhandledCudaError;
    char *h = static_cast<char *>(c->host);
    size_t off = 0;
    do {
        const size_t piece = bytes - off < kSlotBytes ? bytes - off : kSlotBytes;
        if (c->rank == root && piece && hipMemcpy(h, static_cast<const char *>(send) + off, piece, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
        const ncclResult_t r = step(c, h, piece, c->rank == root, h, root);
        if (r != ncclSuccess) return r;
        if (piece && hipMemcpy(static_cast<char *>(recv) + off, h, piece, hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
        off += piece;
    } while (off < bytes);
    return ncclSuccess;
}

const char *ncclGetErrorString(ncclResult_t r)
{
    switch (r) {
    case ncclSuccess: return "no error";
    case ncclUnhandledCudaError: return "fakerccl: HIP call failed";
    case ncclSystemError: return "fakerccl: system error (shared memory / rendezvous / peer timeout)";
    case ncclInvalidArgument: return "fakerccl: invalid argument";
    case ncclInvalidUsage: return "fakerccl: invalid usage";
    default: return "fakerccl: error";
    }
}

}  // extern "C"
