// shm_exchange.cpp -- an all-gather between the processes of ONE host through POSIX shared memory: the transport of
// pnr_trace_replay_sharded's per-poll exchange when all ranks share a node (8 GPUs of one MI355X box).  The records that are
// exchanged are written by the GPU into pinned HOST memory and consumed by the HOST replay, so a host-side transport saves the
// host -> device -> xGMI -> device -> host round trip of a device collective: an exchange costs a few microseconds instead of
// ~100 us.  (Across hosts, or if the processes cannot share memory, the exchange callback is RCCL: pnr_amd/multigpu.py.)
// The same segment serves the few other collectives of the sharded path in the C++ host (advantra_cli --ranks N).
//
// Layout: header | 2 x world x capacity bytes (double-buffered).  One sense-reversing barrier per exchange: a rank writes its block
// of parity p, waits for everybody, reads all blocks of parity p; it can only reach the exchange after the next one -- which
// reuses parity p -- after every rank has passed the next barrier, i.e. has finished reading.
#include "../../include/pnr_hip.h"
#include <atomic>
#include <cerrno>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <fcntl.h>
#include <string>
#include <sys/mman.h>
#include <sys/stat.h>
#include <thread>
#include <time.h>
#include <signal.h>
#include <unistd.h>

namespace pnr { void set_error(const char *fmt, ...); }

namespace {
struct ShmHeader {
    std::atomic<uint32_t> magic;   // set last by the creator; SHM_DEAD once a later job's rank 0 has found the segment stale
    uint32_t world;
    uint64_t capacity;             // bytes per rank and buffer
    std::atomic<uint32_t> arrived; // barrier: ranks that have arrived in this phase
    std::atomic<uint32_t> phase;   // barrier: generation
    std::atomic<uint32_t> attached, failed; // attached: ranks that mapped the segment (reported when the attach barrier times out)
    uint64_t stamp;                // CLOCK_REALTIME ns at creation (diagnostics)
    uint32_t owner_pid;            // the creator's process id: a later job's rank 0 only declares a segment stale when that process is gone
};
constexpr uint32_t SHM_MAGIC = 0x504e5258u; // "PNRX"
constexpr uint32_t SHM_DEAD = 0x44454144u;  // "DEAD"
constexpr size_t HDR = 256;

inline void cpu_relax()
{
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#else
    std::this_thread::yield();
#endif
}
} // namespace

struct pnr_shm_exchange {
    std::string name;
    int rank = 0, world = 1, fd = -1;
    size_t bytes = 0;
    unsigned char *base = nullptr;
    ShmHeader *h = nullptr;
    uint64_t round = 0;
    double timeout_s = 300.0;
    bool owner = false;
};

static bool shm_barrier(pnr_shm_exchange *x)
{
    ShmHeader *h = x->h;
    const uint32_t gen = h->phase.load(std::memory_order_acquire);
    if (h->arrived.fetch_add(1, std::memory_order_acq_rel) + 1 == (uint32_t)x->world) {
        h->arrived.store(0, std::memory_order_relaxed);
        h->phase.store(gen + 1, std::memory_order_release);
        return true;
    }
    // Spin for about a quarter of a millisecond (ranks in step meet within microseconds, and an exchange happens every millisecond
    // or two: a sleeping rank's wake-up latency would be paid by everybody at the NEXT barrier -- measured: naps that grew to 0.5 ms
    // doubled the tracing time of 8 emulated ranks), then sleep in short slices: a rank that waits for a much slower one must not
    // burn the core its own host threads were counted on (host_threads = CPUs / local_ranks).
    const auto t0 = std::chrono::steady_clock::now();
    for (uint64_t spins = 0; h->phase.load(std::memory_order_acquire) == gen; spins++) {
        if (h->failed.load(std::memory_order_relaxed)) return false;
        if ((spins & 255) != 255) { cpu_relax(); continue; }
        const double waited = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (waited < 250e-6) continue;
        std::this_thread::sleep_for(std::chrono::microseconds(waited < 5e-3 ? 20 : 200));
        if (waited > x->timeout_s) {
            h->failed.store(1, std::memory_order_relaxed);
            return false;
        }
    }
    return true;
}

extern "C" {

int pnr_shm_exchange_open(const char *name, int rank, int world, int64_t capacity_bytes, pnr_shm_exchange **out)
{
    if (!name || !out || world < 1 || rank < 0 || rank >= world || capacity_bytes < 16) {
        pnr::set_error("pnr_shm_exchange_open: bad argument");
        return PNR_E_ARG;
    }
    *out = nullptr;
    pnr_shm_exchange *x = new pnr_shm_exchange();
    x->name = name[0] == '/' ? name : std::string("/") + name;
    x->rank = rank; x->world = world;
    const size_t cap = ((size_t)capacity_bytes + 63) / 64 * 64;
    x->bytes = HDR + 2 * (size_t)world * cap;
    const auto t0 = std::chrono::steady_clock::now();
    if (rank == 0) {
        // A stale segment of a crashed run under the same name: a rank of THIS job may have opened it already (it was there before
        // we were) -- mark it dead first, so that whoever waits on it gives up and opens the name again, then remove the name.  Only
        // a segment whose creator is gone is stale: one whose creator still runs belongs to a live job that is attaching right now
        // (the name is removed once everybody is attached) -- its name is taken over, as ever, but its ranks are left alone.
        {
            const int ofd = shm_open(x->name.c_str(), O_RDWR, 0600);
            struct stat sb;
            if (ofd >= 0 && fstat(ofd, &sb) == 0 && (size_t)sb.st_size >= HDR) {
                void *m = mmap(nullptr, HDR, PROT_READ | PROT_WRITE, MAP_SHARED, ofd, 0);
                if (m != MAP_FAILED) {
                    ShmHeader *old = (ShmHeader *)m;
                    const uint32_t opid = old->owner_pid;
                    const bool initialised = old->magic.load(std::memory_order_acquire) == SHM_MAGIC;
                    const bool owner_gone = !initialised || opid == 0 || (kill((pid_t)opid, 0) != 0 && errno == ESRCH);
                    if (owner_gone) {
                        old->magic.store(SHM_DEAD, std::memory_order_release);
                        old->failed.store(1, std::memory_order_release);
                    }
                    munmap(m, HDR);
                }
            }
            if (ofd >= 0) close(ofd);
        }
        shm_unlink(x->name.c_str());
        x->fd = shm_open(x->name.c_str(), O_CREAT | O_EXCL | O_RDWR, 0600);
        if (x->fd < 0 || ftruncate(x->fd, (off_t)x->bytes) != 0) {
            pnr::set_error("shm_open / ftruncate of %s (%zu B) failed: %s", x->name.c_str(), x->bytes, strerror(errno));
            if (x->fd >= 0) { close(x->fd); shm_unlink(x->name.c_str()); }
            delete x;
            return PNR_E_NOMEM;
        }
        x->owner = true;
    }
    auto unmap = [&]() {
        if (x->base && x->base != (unsigned char *)MAP_FAILED) munmap(x->base, x->bytes);
        if (x->fd >= 0) close(x->fd);
        x->base = nullptr; x->h = nullptr; x->fd = -1;
    };
    auto late = [&](double limit) { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > limit; };
    for (;;) { // (a non-owner comes back here when the segment it found turns out to be a dead one)
        if (rank != 0) {
            for (;;) { // the creator may not be there yet
                x->fd = shm_open(x->name.c_str(), O_RDWR, 0600);
                struct stat sb;
                if (x->fd >= 0 && fstat(x->fd, &sb) == 0 && (size_t)sb.st_size >= x->bytes) break;
                if (x->fd >= 0) { close(x->fd); x->fd = -1; }
                if (late(120.0)) {
                    pnr::set_error("shared segment %s did not appear", x->name.c_str());
                    delete x;
                    return PNR_E_STATE;
                }
                std::this_thread::sleep_for(std::chrono::milliseconds(2));
            }
        }
        x->base = (unsigned char *)mmap(nullptr, x->bytes, PROT_READ | PROT_WRITE, MAP_SHARED, x->fd, 0);
        if (x->base == (unsigned char *)MAP_FAILED) {
            pnr::set_error("mmap of %s failed: %s", x->name.c_str(), strerror(errno));
            x->base = nullptr;
            unmap();
            if (x->owner) shm_unlink(x->name.c_str());
            delete x;
            return PNR_E_NOMEM;
        }
        x->h = (ShmHeader *)x->base;
        if (rank == 0) {
            struct timespec now;
            clock_gettime(CLOCK_REALTIME, &now);
            x->h->world = (uint32_t)world;
            x->h->capacity = cap;
            x->h->stamp = (uint64_t)now.tv_sec * 1000000000ull + (uint64_t)now.tv_nsec;
            x->h->owner_pid = (uint32_t)getpid();
            x->h->arrived.store(0); x->h->phase.store(0); x->h->attached.store(0); x->h->failed.store(0);
            x->h->magic.store(SHM_MAGIC, std::memory_order_release);
        } else {
            uint32_t mg;
            while ((mg = x->h->magic.load(std::memory_order_acquire)) != SHM_MAGIC && mg != SHM_DEAD) {
                if (late(120.0)) {
                    pnr::set_error("shared segment %s was never initialised", x->name.c_str());
                    unmap();
                    delete x;
                    return PNR_E_STATE;
                }
                std::this_thread::sleep_for(std::chrono::milliseconds(1));
            }
            if (mg == SHM_DEAD) { unmap(); std::this_thread::sleep_for(std::chrono::milliseconds(1)); continue; } // a crashed job's segment: open the name again
            if (x->h->world != (uint32_t)world || x->h->capacity != cap) {
                pnr::set_error("shared segment %s belongs to a different job (world %u, capacity %llu)", x->name.c_str(), x->h->world, (unsigned long long)x->h->capacity);
                unmap();
                delete x;
                return PNR_E_STATE;
            }
        }
        x->h->attached.fetch_add(1);
        if (!shm_barrier(x)) { // everybody is attached: the name can go (the mappings stay)
            // it was a stale one after all?  Rank 0 of this job may be a moment away from saying so (a crashed job's segment can carry
            // `failed` already): look again for a while -- at the magic word, and at whether the name leads to another segment by now
            bool stale = false;
            for (int tries = 0; rank != 0 && !stale && tries < 400 && !late(120.0); tries++) {
                if (x->h->magic.load(std::memory_order_acquire) == SHM_DEAD) { stale = true; break; }
                struct stat mine, named;
                const int nfd = shm_open(x->name.c_str(), O_RDWR, 0600);
                if (nfd >= 0) {
                    if (fstat(x->fd, &mine) == 0 && fstat(nfd, &named) == 0 && (mine.st_ino != named.st_ino || mine.st_dev != named.st_dev)) stale = true;
                    close(nfd);
                }
                if (!stale) std::this_thread::sleep_for(std::chrono::milliseconds(5));
            }
            if (stale) { unmap(); continue; }
            pnr::set_error("ranks did not all attach to %s (%u of %d did)", x->name.c_str(), x->h->attached.load(), world);
            unmap();
            if (x->owner) shm_unlink(x->name.c_str());
            delete x;
            return PNR_E_STATE;
        }
        if (rank != 0 && x->h->magic.load(std::memory_order_acquire) != SHM_MAGIC) { unmap(); continue; } // (a dead segment whose old barrier count let us through)
        break;
    }
    if (x->owner) shm_unlink(x->name.c_str());
    *out = x;
    return PNR_OK;
}

// pnr_allgather_fn: user = the pnr_shm_exchange
int pnr_shm_allgather(void *user, const void *send, void *recv, int64_t bytes)
{
    pnr_shm_exchange *x = (pnr_shm_exchange *)user;
    if (!x || !send || !recv || bytes < 0 || (uint64_t)bytes > x->h->capacity) return PNR_E_ARG;
    const size_t cap = (size_t)x->h->capacity;
    unsigned char *buf = x->base + HDR + (size_t)(x->round & 1) * (size_t)x->world * cap;
    std::memcpy(buf + (size_t)x->rank * cap, send, (size_t)bytes);
    if (!shm_barrier(x)) return PNR_E_STATE;
    for (int r = 0; r < x->world; r++) std::memcpy((unsigned char *)recv + (size_t)r * (size_t)bytes, buf + (size_t)r * cap, (size_t)bytes);
    x->round++;
    return PNR_OK;
}

void pnr_shm_exchange_close(pnr_shm_exchange *x)
{
    if (!x) return;
    if (x->base) munmap(x->base, x->bytes);
    if (x->fd >= 0) close(x->fd);
    delete x;
}

} // extern "C"
