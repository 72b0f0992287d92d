// rccl_comm.hpp -- the two collectives pantax_hip_profile asks its host for, over RCCL (one process per GPU, xGMI between
// them).  Used by the pantax-hip CLI with --ranks N --rank r; a Rust host would do the same through the rccl crate or FFI
// (INTEGRATION.md section 3b'').
//
// Bootstrap without MPI: rank 0 creates the ncclUniqueId and publishes it through a file in the work directory (written
// under a temporary name, then renamed) together with a per-launch NONCE (--comm-nonce; default: $TORCHELASTIC_RUN_ID + restart
// count when a launcher sets them, else $MASTER_PORT, else "0").  The other ranks accept a file only if it carries THEIR nonce
// AND is not older than their own start minus two seconds -- both, always: a launcher may hand the same nonce to consecutive
// launches (torchrun's default port never changes), and a file a killed run left behind with that nonce would send a rank that
// starts before rank 0's unlink into ncclCommInitRank with a dead id (round-3 advisor finding).
// Round 6: the bootstrap has an IN-PROCESS deadline (--comm-timeout, default 300 s for the whole init).  ncclCommInitRank runs on a helper thread and
// the calling thread waits for it with a deadline: a peer that never starts, or dies inside the bootstrap, no longer leaves this rank waiting
// for a launcher-side timeout -- init() fails, and the process leaves with _exit (the helper is still inside RCCL and cannot be cancelled;
// nothing is re-exec'ed: a process that touched the GPU only ever exits).
#pragma once
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <sys/stat.h>
#include <unistd.h>
#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>

struct RcclComm {
    ncclComm_t comm = nullptr;
    hipStream_t stream = nullptr;
    double *d_buf = nullptr;
    size_t d_cap = 0;
    int rank = 0, world = 1;
    std::string err;

    bool fail(const std::string &what) { err = what; return false; }

    bool init(int rank_, int world_, const std::string &id_file, uint64_t nonce = 0, double timeout_s = 300.0) {
        rank = rank_; world = world_;
        const auto t_start = std::chrono::system_clock::now();
        ncclUniqueId id;
        struct Rec { uint64_t nonce; ncclUniqueId id; } rec;
        if (rank == 0) {
            ::unlink(id_file.c_str());   // a file left behind by an earlier run
            if (ncclGetUniqueId(&id) != ncclSuccess) return fail("ncclGetUniqueId failed");
            rec.nonce = nonce; rec.id = id;
            const std::string tmp = id_file + ".tmp";
            FILE *f = std::fopen(tmp.c_str(), "wb");
            if (!f || std::fwrite(&rec, sizeof(rec), 1, f) != 1) { if (f) std::fclose(f); return fail("cannot write " + tmp); }
            std::fclose(f);
            if (std::rename(tmp.c_str(), id_file.c_str()) != 0) return fail("cannot publish " + id_file);
        } else {
            const auto deadline = std::chrono::steady_clock::now() + std::chrono::duration<double>(timeout_s);
            for (;;) {
                struct stat st;
                if (::stat(id_file.c_str(), &st) == 0 && (size_t)st.st_size == sizeof(rec)) {
                    const auto mt = std::chrono::system_clock::from_time_t(st.st_mtime);
                    FILE *f = std::fopen(id_file.c_str(), "rb");
                    const bool ok = f && std::fread(&rec, sizeof(rec), 1, f) == 1;
                    if (f) std::fclose(f);
                    // this launch's file: its nonce AND not older than this rank's start (st_mtime has one-second resolution: two seconds of slack)
                    if (ok && rec.nonce == nonce && mt + std::chrono::seconds(2) >= t_start) { id = rec.id; break; }
                }
                if (std::chrono::steady_clock::now() > deadline) return fail("rank 0 did not publish " + id_file + " in time");
                std::this_thread::sleep_for(std::chrono::milliseconds(20));
            }
        }
        if (hipStreamCreateWithFlags(&stream, hipStreamNonBlocking) != hipSuccess) return fail("hipStreamCreate failed");
        // The bootstrap under what is left of the deadline: ncclCommInitRank on a helper thread, this thread waits on a condition variable.  (A
        // non-blocking communicator -- config.blocking = 0 + ncclCommGetAsyncError -- was tried first: in this RCCL the call itself sits in the
        // bootstrap's rendezvous until every rank has connected, so it never returned to be polled.)  At the deadline the helper is left behind,
        // `abandoned` is set and the caller must leave the process with _exit: the stuck thread cannot be cancelled, and nothing may wait for it.
        const double left_s = std::max(1.0, timeout_s - std::chrono::duration<double>(std::chrono::system_clock::now() - t_start).count());
        struct Boot { std::mutex mu; std::condition_variable cv; bool done = false; ncclResult_t rc = ncclSuccess; ncclComm_t comm = nullptr; };
        auto boot = std::make_shared<Boot>();
        int dev = 0;
        (void)hipGetDevice(&dev);
        std::thread([boot, id, dev, w = world, r = rank] {
            (void)hipSetDevice(dev);
            ncclComm_t c = nullptr;
            const ncclResult_t rc = ncclCommInitRank(&c, w, id, r);
            std::lock_guard<std::mutex> g(boot->mu);
            boot->rc = rc; boot->comm = c; boot->done = true;
            boot->cv.notify_all();
        }).detach();
        {
            std::unique_lock<std::mutex> lk(boot->mu);
            if (!boot->cv.wait_for(lk, std::chrono::duration<double>(left_s), [&] { return boot->done; })) {
                abandoned = true;
                return fail("RCCL bootstrap did not complete within the deadline (a rank of this launch never started, or died); giving up");
            }
            if (boot->rc != ncclSuccess) return fail(std::string("ncclCommInitRank failed: ") + ncclGetErrorString(boot->rc));
            comm = boot->comm;
        }
        return true;
    }
    bool abandoned = false;   // init() gave up at its deadline with the bootstrap thread still inside RCCL: leave the process with _exit
    // a collective enqueued on a non-blocking communicator may return ncclInProgress: wait for it to be accepted (same deadline rule as the bootstrap)
    bool settle(ncclResult_t rc, double timeout_s = 300.0) {
        if (rc == ncclSuccess) return true;
        if (rc != ncclInProgress) return false;
        const auto deadline = std::chrono::steady_clock::now() + std::chrono::duration<double>(timeout_s);
        for (;;) {
            ncclResult_t state = ncclSuccess;
            if (ncclCommGetAsyncError(comm, &state) != ncclSuccess) return false;
            if (state == ncclSuccess) return true;
            if (state != ncclInProgress || std::chrono::steady_clock::now() > deadline) return false;
            std::this_thread::sleep_for(std::chrono::microseconds(50));
        }
    }
    void destroy(const std::string &id_file) {
        if (comm) (void)ncclCommDestroy(comm);
        if (stream) (void)hipStreamDestroy(stream);
        if (d_buf) (void)hipFree(d_buf);
        if (rank == 0) ::unlink(id_file.c_str());
        comm = nullptr; stream = nullptr; d_buf = nullptr;
    }

    // int (*allreduce_sum)(void *user, double *buf, uint64_t n): host buffer, summed in place over the ranks
    static int allreduce_sum(void *user, double *buf, uint64_t n) {
        RcclComm *c = static_cast<RcclComm *>(user);
        if (n == 0) return 0;
        if (n > c->d_cap) {
            if (c->d_buf) (void)hipFree(c->d_buf);
            c->d_buf = nullptr; c->d_cap = 0;
            if (hipMalloc(reinterpret_cast<void **>(&c->d_buf), n * sizeof(double)) != hipSuccess) return 1;
            c->d_cap = n;
        }
        if (hipMemcpyAsync(c->d_buf, buf, n * sizeof(double), hipMemcpyHostToDevice, c->stream) != hipSuccess) return 2;
        if (!c->settle(ncclAllReduce(c->d_buf, c->d_buf, n, ncclDouble, ncclSum, c->comm, c->stream))) return 3;
        if (hipMemcpyAsync(buf, c->d_buf, n * sizeof(double), hipMemcpyDeviceToHost, c->stream) != hipSuccess) return 4;
        return hipStreamSynchronize(c->stream) == hipSuccess ? 0 : 5;
    }
    // int (*alltoallv)(void *user, const void *send, const uint64_t *send_off, void *recv, const uint64_t *recv_off):
    // DEVICE buffers (comm_device_buffers = 1), byte offsets [world + 1]
    static int alltoallv(void *user, const void *send, const uint64_t *send_off, void *recv, const uint64_t *recv_off) {
        RcclComm *c = static_cast<RcclComm *>(user);
        const char *s = static_cast<const char *>(send);
        char *r = static_cast<char *>(recv);
        if (ncclGroupStart() != ncclSuccess) return 1;
        int bad = 0;
        auto ok = [](ncclResult_t rc) { return rc == ncclSuccess || rc == ncclInProgress; };   // (inside a group of a non-blocking communicator the calls only queue)
        for (int peer = 0; peer < c->world && !bad; ++peer) {
            const uint64_t ns = send_off[peer + 1] - send_off[peer], nr = recv_off[peer + 1] - recv_off[peer];
            if (ns && !ok(ncclSend(s + send_off[peer], ns, ncclUint8, peer, c->comm, c->stream))) bad = 2;
            if (!bad && nr && !ok(ncclRecv(r + recv_off[peer], nr, ncclUint8, peer, c->comm, c->stream))) bad = 3;
        }
        if (!c->settle(ncclGroupEnd()) && !bad) bad = 4;   // the group is closed on every path
        if (bad) return bad;
        return hipStreamSynchronize(c->stream) == hipSuccess ? 0 : 5;
    }
};
