// stream_sched.h -- the streaming trace scheduler: the production form of the trace loop of reconstruction_func
// (Advantra_plugin.cpp:2658-2710) for one GPU or for the sorted seeds sharded over several ranks.  Host-only logic; the
// particle filters behind it are a StreamEngine (the phased HIP kernels in smc_phased.hip; tests put a host engine that plays
// recorded map-free traces in their place to exercise this file and the exchange without a GPU).
//
// A window of trace slots is kept full.  Every `poll` SMC steps the host collects the traces that have stopped, replays --
// strictly in seed order, as far as the finished traces reach -- the bookkeeping of Tracker::trackPos (replay.h), pushes the
// voxels that replay filled into the engine's density map, and hands the free slots to the next seeds (a seed on a voxel the
// replayed map already saturates is never traced, :2669-2670).  The engine ends a trace at the first iteration whose centroid
// voxel is saturated in its map (DENSITY stop, tracker.cpp:855).  That map only ever holds replayed -- final -- nodes of
// lower-ranked seeds, so it can only under-count what the sequential reference would see: no trace is cut earlier than the
// reference cuts it, and the replay, which applies the true map, yields exactly the node graph of tracing everything to its
// map-free end.
//
// Tentative replay (option "tentative", default on).  Speculation is what the window costs: a trace beyond the frontier runs without
// the nodes of the unreplayed seeds in front of it, and nearly half of the SMC iterations run that way are cut away by the replay
// later.  After every poll the scheduler therefore replays -- on an overlay over the final map, never into it -- ALL records it
// knows of the unreplayed seeds, in rank order: the finished traces' and what the running traces have written so far.  A cut trace
// takes its later nodes out of the picture, exactly as in the final replay, so the overlay is what the final map would be if
// nothing more were recorded.  A running trace that this replay cuts inside its recorded iterations is PAUSED (no longer stepped;
// it keeps its slot and its state) and resumed if a later pass no longer cuts it (one pause in twenty); as long as every seed in
// front is complete the pass is not tentative at all -- it IS the final replay's verdict -- and a cut trace is ended on the spot,
// its record delivered with the iterations it has.  None of this can change the result: the final replay alone builds the
// graph, from the same records in the same order; a paused trace only stops producing iterations the replay would cut away.
// Measured on the recorded traces of the bench workload (scripts/sim_tentative.py): 177 k -> 119 k SMC iterations for the same
// number of steps, nineteen pauses in twenty never resumed.
//
// Sharded (world > 1): rank r owns the sorted seeds r, r + world, ...  After every poll the ranks all-gather the records of the
// traces that finished on them (one fixed-size block per rank and poll; what does not fit is carried to the next poll) and
// EVERY rank replays the same records in global seed order, so every rank holds the same replayed map -- still only final nodes
// of lower-ranked seeds, so the argument above carries over -- and ends with the same node graph.  The replay is not sharded:
// it is order-dependent by definition.  The reference has no counterpart.
#pragma once
#include "replay.h"
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

namespace pnr {

struct SchedOptions {
    int window = 1536; // trace slots kept busy on this rank
    int look0 = 0;     // seeds admitted at most max(look0, frontier * look_pct / 100) ranks beyond the replay frontier;
    int look_pct = -1; // look0 = 0 / look_pct < 0: automatic -- max(512, 200 %) on one GPU with the tentative replay (there `target` is what
                       // binds), max(128, 64 * world) / min(400, 50 * world) otherwise (scripts/sim_sharded.py)
    int target = -1;   // > 0: seeds are admitted only while fewer than this many traces are RUNNING on this rank (all groups; paused ones do
                       // not count): late seeds live a few iterations, early ones dozens, so a constant number of running traces wastes
                       // fewer iterations per step than a rank window (scripts/sim_tentative.py).  0: off.  -1: automatic -- 120 on one GPU
                       // with the tentative replay and poll = 3 (round 3 measured 160 / 200 / 240 / 320: 1130 / 1131 / 1131 / 1168 ms and took 200;
                       // with ph_predict / ph_update at a third of their time smaller launches cost less -- round 4, bench step: target
                       // 100 / 110 / 120 / 130 / 140 / 160 / 200 at poll 3: 917 / 898 / 890 / 899 / 905 / 918 / 941 ms), 96 per rank sharded
                       // (emulated ranks: 2 / 4 / 8 ranks 654 / 408 / 291 ms at 128, 652 / 401 / 277 ms at 96), off without the tentative replay
    int overfill = 1;  // ... counted as the mean over a poll (see the admission)
    int lag = -1;      // steps of a poll that run while the host works on the state in front of them (see StreamEngine::launch): hides the host's
                       // share of a poll (tentative pass, exchange) where no other trace group covers it, at the price of `lag` more
                       // iterations for every trace about to be paused or ended.  -1: automatic -- half a poll when the group is the only
                       // one running (one trace group, or `concentrate` at work), one step otherwise (measured with two groups on one GPU,
                       // lag 0 / 1 / 2 / 3: tracing 988 / 929 / 945 / 968 ms; 8 emulated ranks, one group each: 391 / 316 / 308 / 304 ms)
    int concentrate = 1; // one GPU, several groups: admit into group 0 only while few traces survive a poll (see the admission)
    int poll = 0;      // SMC steps between two polls (0: automatic, see run_stream)
    int groups = 2;    // trace groups stepping independently (engine permitting)
    int tentative = 1; // pause traces that a tentative replay of everything recorded so far cuts (see above)
    int timing = 0;    // one line of statistics on stderr
};

struct ShardSpec {
    int rank = 0, world = 1;
    pnr_allgather_fn exchange = nullptr;
    void *user = nullptr;
    int64_t block_bytes = 0; // bytes every rank contributes per exchange (0: default)
};

struct SchedStats {
    int64_t steps = 0, polls = 0, iters = 0, exchanges = 0, carried = 0, launched = 0, skipped = 0;
    int64_t paused = 0, resumed = 0, ended = 0, tent_nodes = 0, tent_passes = 0; // tentative replay
    double tent_ms = 0, wait_ms = 0;                                               // host time in it / blocked in E.wait()
};

class StreamEngine {
public:
    virtual ~StreamEngine() {}
    virtual int slots() const = 0;      // trace slots (even)
    virtual int max_groups() const = 0; // independent trace groups the engine can step
    // queue: hand `m` slots to new traces of group g (s6 = m x (x, y, z, vx, vy, vz))
    virtual int admit(int g, const int *slots, const float *s6, int m) = 0;
    // queue: `poll` SMC steps over the active traces of group g (at most `active`), then the read-back of the slot states
    // `poll` steps of group g; the state wait() hands back is the one after the first poll - lag of them (lag > 0: the host works on it
    // while the last `lag` steps still run, and what it then asks for -- pauses, ends, admissions -- takes effect behind them)
    virtual int launch(int g, int active, int poll, int lag) = 0;
    // block until the last launch of group g has landed; *active = traces of the group still running
    virtual int wait(int g, int *active) = 0;
    // after wait(g): has the trace in `slot` stopped?  *T = its successful iterations; rows() = its min(T + 1, ni) estimates
    virtual bool finished(int g, int slot, int *T) const = 0;
    virtual const pnr_xest *rows(int slot) const = 0;
    // after wait(g), for a trace of group g that is still running: the iterations whose estimates can be read in rows(slot)
    virtual void idle(int g) { (void)g; } // group g has nothing to step this turn
    virtual int progress(int g, int slot) const = 0;
    // queue, in front of group g's next steps and admissions: take the running traces in pause[0..np) off the group's list (they
    // keep slot and state) and put the paused traces in resume[0..nr) back on it
    virtual int control(int g, const int *pause, int np, const int *resume, int nr) = 0;
    // push the density of the voxels in r.touched (final values r.den_at) to the engine's map, in front of group g's next steps
    // (g's last steps have been waited for; the other groups may see a voxel before or after the update: both are under-counts)
    virtual int density_update(const Replayer &r, int g) = 0;
    // block until everything queued for group g (control, admissions) has executed; only called for a group with nothing in flight
    virtual int settle(int g) { (void)g; return 0; }
    // the end of group g's turn (after its control / density_update / admit and, if it had anything to step, its launch): an engine
    // that gathers those requests into one dispatch sends whatever no launch has carried
    virtual int end_turn(int g) { (void)g; return 0; }
    virtual void drain() = 0;
    virtual const char *error() const { return ""; }
};

namespace sched_detail {
enum { HDR_WORDS = 4 }; // payload words, busy, abort, reserved
enum { STALL_TURNS = 12 }; // synchronised turns without progress before every rank gives up -- a constant, so that ranks whose
                           // window allows fewer trace groups (G is local: the stash budget differs per GPU) still stop in the same turn
}

// bytes every rank contributes to one exchange: a function of the options, the world and ni only -- the same on every rank
inline int64_t exchange_block_bytes(int64_t block_opt, int world, int ni)
{
    using namespace sched_detail;
    const size_t rec_max_words = 4 + (size_t)ni * 8;
    int64_t block = block_opt > 0 ? block_opt : std::max<int64_t>(32768, 262144 / std::max(1, world)); // a rank's share of the finished traces shrinks with the world
    block = std::max<int64_t>(block, (int64_t)(HDR_WORDS + rec_max_words) * 4);
    return (block + 15) / 16 * 16;
}

// A rank that fails BEFORE its scheduler runs (engine set-up: out of memory, bad state) must still take part in the first
// exchange the other ranks are about to enter, or they wait for it for ever: one block whose abort word is set.
inline void abort_exchange(const ShardSpec &sh, int ni)
{
    using namespace sched_detail;
    if (sh.world <= 1 || !sh.exchange) return;
    const int64_t block = exchange_block_bytes(sh.block_bytes, sh.world, ni);
    std::vector<int32_t> send((size_t)block / 4, 0), recv((size_t)block / 4 * (size_t)sh.world, 0);
    send[2] = 1;
    (void)sh.exchange(sh.user, send.data(), recv.data(), block);
}

namespace sched_detail {
// the tentative replay's density overlay: voxel -> nodes added by the pass (open addressing, emptied by a generation stamp)
struct Overlay {
    std::vector<int64_t> key;
    std::vector<uint32_t> gen;
    std::vector<uint8_t> cnt;
    uint32_t cur = 0;
    size_t used = 0;
    Overlay() { resize(1 << 14); }
    void resize(size_t nsz) { key.assign(nsz, 0); gen.assign(nsz, 0); cnt.assign(nsz, 0); }
    void begin()
    {
        used = 0;
        if (++cur == 0) { std::fill(gen.begin(), gen.end(), 0u); cur = 1; }
    }
    static size_t hash(int64_t v) { return (size_t)(((uint64_t)v * 0x9E3779B97F4A7C15ull) >> 20); }
    void prefetch(int64_t v) const { __builtin_prefetch(&gen[hash(v) & (key.size() - 1)]); }
    int get(int64_t v) const
    {
        const size_t mask = key.size() - 1;
        for (size_t h = hash(v) & mask;; h = (h + 1) & mask) {
            if (gen[h] != cur) return 0;
            if (key[h] == v) return cnt[h];
        }
    }
    uint8_t &slot(int64_t v) // the voxel's counter, entered with 0 if the pass has not met it (one probe for "read, test, count")
    {
        if (2 * (used + 1) > key.size()) grow();
        const size_t mask = key.size() - 1;
        for (size_t h = hash(v) & mask;; h = (h + 1) & mask) {
            if (gen[h] != cur) { gen[h] = cur; key[h] = v; cnt[h] = 0; used++; return cnt[h]; }
            if (key[h] == v) return cnt[h];
        }
    }
    void add(int64_t v)
    {
        if (2 * (used + 1) > key.size()) grow();
        const size_t mask = key.size() - 1;
        for (size_t h = hash(v) & mask;; h = (h + 1) & mask) {
            if (gen[h] != cur) { gen[h] = cur; key[h] = v; cnt[h] = 1; used++; return; }
            if (key[h] == v) { if (cnt[h] < 255) cnt[h]++; return; }
        }
    }
    void grow()
    {
        std::vector<int64_t> k0; std::vector<uint32_t> g0; std::vector<uint8_t> c0;
        k0.swap(key); g0.swap(gen); c0.swap(cnt);
        resize(k0.size() * 2);
        used = 0;
        const size_t mask = key.size() - 1;
        for (size_t i = 0; i < k0.size(); i++)
            if (g0[i] == cur) {
                size_t h = hash(k0[i]) & mask;
                while (gen[h] == cur) h = (h + 1) & mask;
                gen[h] = cur; key[h] = k0[i]; cnt[h] = c0[i]; used++;
            }
    }
};
} // namespace sched_detail

// Returns 0 or a PNR_E_* code (message in `err`).  `r` ends up holding the node graph; *stats the local counters.
inline int run_stream(StreamEngine &E, const pnr_seed *seeds, int64_t n, int ni, SchedOptions o, const ShardSpec &sh, Replayer &r,
                      SchedStats *stats, std::string &err)
{
    using namespace sched_detail;
    SchedStats st;
    if (stats) *stats = st;
    if (n == 0) return PNR_OK;
    const int world = std::max(1, sh.world), rank = sh.rank;
    if (world > 1 && !sh.exchange) { err = "sharded tracing needs an exchange callback"; return PNR_E_ARG; }
    if (rank < 0 || rank >= world) { err = "rank out of range"; return PNR_E_ARG; }
    if (o.target < 0) o.target = o.tentative ? (world == 1 ? 120 : 96) : 0; // (200 until ph_predict / ph_update took a third of their time: smaller launches cost less now)
    if (o.look0 <= 0) o.look0 = (world == 1 && o.tentative) ? (o.target > 0 ? 512 : 256) : std::max(128, 64 * world);
    if (o.look_pct < 0) o.look_pct = (world == 1 && o.tentative) ? (o.target > 0 ? 200 : 100) : std::min(400, 50 * world);
    // 0: automatic.  On one GPU a poll is one dispatch and one state copy in the stream (round 4): two steps between polls beat three
    // and four on both bench workloads; a sharded poll also exchanges and replays every rank's records, and from four ranks on four
    // steps are better (emulated ranks, step in ms at poll 2 / 3 / 4: 2 ranks 647 / 665 / 673, 4 ranks 430 / 436 / 425, 8 ranks 309 / 311 / 297)
    if (o.poll <= 0) o.poll = world <= 2 ? 2 : 4;
    o.poll = std::max(1, o.poll);
    const int NT = E.slots() - (E.slots() & 1);
    if (NT < 2) { err = "no trace slots"; abort_exchange(sh, ni); return PNR_E_STATE; }
    int G = std::min(std::max(1, o.groups), E.max_groups());
    if (NT < 4 * G) G = 1;
    const int64_t look0 = o.look0, look_pct = o.look_pct;

    struct SeedRec {
        uint8_t have = 0;      // directions whose record has arrived
        uint8_t got = 0;       // ... which ones (bit = direction)
        uint8_t part = 0;      // directions whose trace is PAUSED on its rank and has published what it recorded (prow rows in xc)
        int32_t prow[2] = {0, 0};
        bool skipped = false;  // sits on a saturated voxel: never traced (:2669-2670)
        int32_t T[2] = {0, 0};
        std::vector<pnr_xest> xc; // [2][ni], allocated when the first record arrives, dropped after the replay
        std::vector<int64_t> vox; // [2][ni] voxels of the first nvox[dir] recorded estimates: the tentative replay visits them at every poll
        int32_t nvox[2] = {0, 0};
    };
    std::vector<SeedRec> rec((size_t)n);
    struct Grp {
        int active = 0; bool inflight = false;
        std::vector<int> busy;    // slots of this group's traces that have not delivered their record (running or paused)
        // Slots of traces the HOST ended on a final verdict.  The device may still be stepping such a trace (the last `lag` steps of the
        // poll in flight) and the control() that takes it off the group's list is only queued behind them: the group itself may reuse the
        // slot at once (its admission follows in stream order), another group -- another stream -- only after this group's next wait()
        // or settle() has shown that the control() has run.
        std::vector<int> held;
        int npaused = 0;
        int lag = 0;          // steps of the poll in flight that follow the state wait() will hand back
        int last_start = 0;   // traces on the device's list when the last poll's steps were launched
        double keep = 1.0;    // smoothed share of them still running when the poll was collected
    };
    std::vector<Grp> grp((size_t)G);
    std::vector<int> free_slots;
    for (int k = NT - 1; k >= 0; k--) free_slots.push_back(k);
    std::vector<int64_t> slot_seed((size_t)NT, -1);
    std::vector<int> slot_dir((size_t)NT, 0), slot_group((size_t)NT, -1);
    std::vector<uint8_t> slot_paused((size_t)NT, 0);
    std::vector<uint8_t> slot_fresh((size_t)NT, 0); // admitted since its group was last waited for: the engine's view of the slot is still the previous occupant's
    std::vector<int32_t> seed_slot((size_t)(2 * n), -1); // slot of trace (seed, direction) while it runs on this rank
    int64_t max_known = -1;                               // highest seed any record or admission has touched
    Overlay ov;
    std::vector<int> pause_list, resume_list;
    // Sharded, a rank only knows its OWN running traces; every rank therefore PUBLISHES what its unfinished traces have recorded since
    // the last exchange (record codes 2 / 3: a few KB per rank and exchange), so that every rank's tentative replay sees all traces
    // and all of them end a trace in the same turn as its owner once the verdict is final.  (Without that a paused trace waits for
    // the frontier to reach it, one exchange per seed: measured on 8 emulated ranks, 944 against 468 ms.)
    const bool tentative = o.tentative != 0;
    std::vector<int> new_slots;
    std::vector<float> new_s6;
    int64_t next = rank, frontier = 0;

    // ---- records: [seed, code, T, rows] + rows x 8 floats, as 32-bit words.  code: 0 / 1 the final record of that direction, -1 the
    // seed was skipped, 2 / 3 PROGRESS of the unfinished trace of direction code - 2: `rows` further estimates from iteration T on
    // (sharded: every rank publishes what its running traces have recorded since the last exchange, so that every rank's tentative
    // replay sees all traces, and all ranks end a trace in the same turn as its owner once the verdict is final), 4 / 5 reserved
    std::vector<int32_t> outbox; // finished on this rank, not yet applied / sent
    size_t out_head = 0;
    std::vector<int32_t> slot_pub((size_t)NT, 0); // iterations of the slot's trace the other ranks have been told of
    // sharded: tell the other ranks what the trace in `slot` has recorded since they were last told (no-op on one GPU)
    auto publish_progress = [&](int slot, const pnr_xest *X, int nr) {
        if (world <= 1) return;
        const int from = slot_pub[(size_t)slot];
        if (nr <= from) return;
        const size_t at = outbox.size();
        outbox.resize(at + 4 + (size_t)(nr - from) * 8);
        outbox[at] = (int32_t)slot_seed[(size_t)slot]; outbox[at + 1] = 2 + slot_dir[(size_t)slot]; outbox[at + 2] = from; outbox[at + 3] = nr - from;
        std::memcpy(&outbox[at + 4], X + from, (size_t)(nr - from) * sizeof(pnr_xest));
        slot_pub[(size_t)slot] = nr;
    };

    const int64_t block = exchange_block_bytes(sh.block_bytes, world, ni);
    const size_t block_words = (size_t)block / 4;
    std::vector<int32_t> sendbuf, recvbuf;
    if (world > 1) { sendbuf.assign(block_words, 0); recvbuf.assign(block_words * (size_t)world, 0); }

    auto apply = [&](const int32_t *w, size_t nw) -> bool { // one rank's block of whole records
        size_t k = 0;
        while (k < nw) {
            if (k + 4 > nw) return false;
            const int64_t s = w[k];
            const int code = w[k + 1], Tn = w[k + 2], rows = w[k + 3];
            if (s < 0 || s >= n || rows < 0 || rows > ni || k + 4 + (size_t)rows * 8 > nw || code < -1 || code > 5) return false;
            SeedRec &sr = rec[(size_t)s];
            const int dir = code & 1;
            if (code < 0) {
                sr.skipped = true;
            } else if (sr.got >> dir & 1) {
                // this rank has already concluded the trace from its published rows (the tentative replay's verdict was final): what
                // its owner sends afterwards says the same
            } else if (code <= 1) {
                if (sr.have >= 2) return false;
                if (sr.xc.empty()) sr.xc.resize((size_t)2 * ni);
                sr.T[dir] = Tn;
                if (rows > 0) std::memcpy(sr.xc.data() + (size_t)dir * ni, w + k + 4, (size_t)rows * sizeof(pnr_xest));
                sr.have++;
                sr.got |= (uint8_t)(1 << dir);
                sr.part &= (uint8_t)~(1 << dir);
            } else if (code <= 3) {
                if (Tn < 0 || Tn + rows > ni) return false;
                if (sr.xc.empty()) sr.xc.resize((size_t)2 * ni);
                if (rows > 0) std::memcpy(sr.xc.data() + (size_t)dir * ni + Tn, w + k + 4, (size_t)rows * sizeof(pnr_xest));
                sr.prow[dir] = Tn + rows;
                sr.part |= (uint8_t)(1 << dir);
            }
            if (s > max_known) max_known = s;
            k += 4 + (size_t)rows * 8;
        }
        return true;
    };
    auto out_record_words = [&](size_t at) { return 4 + (size_t)outbox[at + 3] * 8; };

    long long dbg_end[4] = {0, 0, 0, 0}, dbg_act = 0, dbg_turns = 0, dbg_paused = 0, dbg_free = 0;
    long long dbg_size[5] = {0, 0, 0, 0, 0}, dbg_drain = 0, dbg_drain_act = 0; // launches by size (< 16, < 32, < 64, < 128, more); launches after the last admission
    bool single = false; // admissions go to group 0 only (see the admission)
    int rc = PNR_OK;
    int idle_turns = 0;
    bool aborted = false;
    // one exchange: what fits of the outbox + this rank's busy / abort flags; *busy_all = ranks with traces in flight or records
    // still to send (the same number on every rank)
    bool all_done = false; // every rank said, in the same exchange, that its replay has reached the end
    auto exchange = [&](int busy, int abort_flag, int *busy_all, int *abort_rank) -> int {
        size_t nw = 0;
        if (out_head < outbox.size()) busy = 1;
        if (!abort_flag)
            while (out_head + nw < outbox.size()) {
                const size_t rw = out_record_words(out_head + nw);
                if (HDR_WORDS + nw + rw > block_words) break;
                nw += rw;
            }
        // word 3: this rank's replay is complete.  The ranks leave together, after an exchange in which ALL of them said so: with the
        // tentative replay a rank can draw a final verdict a turn before the others, so "my frontier is at the end" is no longer
        // something every rank reaches in the same turn
        sendbuf[0] = (int32_t)nw; sendbuf[1] = busy; sendbuf[2] = abort_flag; sendbuf[3] = (frontier >= n || r.stopped) ? 1 : 0;
        if (nw) std::memcpy(sendbuf.data() + HDR_WORDS, outbox.data() + out_head, nw * 4);
        out_head += nw;
        if (out_head < outbox.size()) st.carried++;
        else { outbox.clear(); out_head = 0; }
        const int xrc = sh.exchange(sh.user, sendbuf.data(), recvbuf.data(), block);
        st.exchanges++;
        if (xrc) { err = "exchange callback failed (" + std::to_string(xrc) + ")"; return PNR_E_STATE; }
        *busy_all = 0; *abort_rank = -1;
        all_done = true;
        for (int q = 0; q < world; q++) {
            const int32_t *b = recvbuf.data() + (size_t)q * block_words;
            if (b[2]) { *abort_rank = q; all_done = false; continue; }
            *busy_all += b[1] ? 1 : 0;
            all_done = all_done && b[3] != 0;
            if (b[0] < 0 || (size_t)b[0] > block_words - HDR_WORDS || !apply(b + HDR_WORDS, (size_t)b[0])) {
                err = "malformed trace records from rank " + std::to_string(q);
                return PNR_E_STATE;
            }
        }
        return PNR_OK;
    };
    // a failing rank tells the others in one last exchange, so that nobody is left waiting for it -- unless the run is over for
    // everybody (the last exchange has brought the frontier to the end, or MAX_TRACE_COUNT): then nobody exchanges again
    auto fail = [&](int code) -> int {
        E.drain();
        if (world > 1 && !aborted && !all_done) { int b = 0, a = -1; std::string keep = err; (void)exchange(0, 1, &b, &a); err = keep; }
        return code;
    };

    for (int g = 0;; g = (g + 1) % G) {
        Grp &q = grp[(size_t)g];
        // ---- the group's last poll: which of its traces have stopped?
        if (q.inflight) {
            const auto tw0 = std::chrono::steady_clock::now();
            rc = E.wait(g, &q.active);
            st.wait_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tw0).count();
            if (rc) { err = E.error(); return fail(rc); }
            q.inflight = false;
            st.polls++;
            free_slots.insert(free_slots.end(), q.held.begin(), q.held.end()); // (the control() that removed them ran before the state copy)
            q.held.clear();
            for (int slot : q.busy) slot_fresh[(size_t)slot] = 0;
            size_t keep = 0;
            for (size_t b = 0; b < q.busy.size(); b++) {
                const int slot = q.busy[b];
                int Tn = 0;
                if (!E.finished(g, slot, &Tn)) { q.busy[keep++] = slot; continue; }
                if (slot_paused[(size_t)slot]) { slot_paused[(size_t)slot] = 0; q.npaused--; } // (stopped in the steps that were queued in front of its pause: lag)
                seed_slot[(size_t)(2 * slot_seed[(size_t)slot] + slot_dir[(size_t)slot])] = -1;
                const int rows = std::min(std::max(Tn, 0) + 1, ni); // + the iteration that failed (its corr is what the reference prints)
                const size_t at = outbox.size();
                outbox.resize(at + 4 + (size_t)rows * 8);
                outbox[at] = (int32_t)slot_seed[(size_t)slot]; outbox[at + 1] = slot_dir[(size_t)slot]; outbox[at + 2] = Tn; outbox[at + 3] = rows;
                if (rows > 0) std::memcpy(&outbox[at + 4], E.rows(slot), (size_t)rows * sizeof(pnr_xest));
                st.iters += std::min(Tn + 1, ni);
                slot_seed[(size_t)slot] = -1;
                free_slots.push_back(slot);
            }
            q.busy.resize(keep);
        } else if (!q.held.empty()) { // nothing in flight to wait for: make sure the queued control() has run before other groups may take the slots
            rc = E.settle(g);
            if (rc) { err = E.error(); return fail(rc); }
            free_slots.insert(free_slots.end(), q.held.begin(), q.held.end());
            q.held.clear();
        }
        // ---- the finished records reach the replay: directly, or through the all-gather of every rank's block -- once per rotation
        // of the trace groups (every rank runs the same number of groups, so all of them exchange in the same turns)
        int busy_all = 0;
        const bool sync_turn = world == 1 || g == G - 1;
        const int64_t frontier_was = frontier;
        if (sync_turn) {
            if (world == 1) {
                if (!apply(outbox.data(), outbox.size())) { err = "malformed trace record"; return fail(PNR_E_STATE); }
                outbox.clear();
            } else {
                if (tentative)
                    for (int k = 0; k < G; k++)
                        for (int slot : grp[(size_t)k].busy)
                            if (!slot_fresh[(size_t)slot] && !slot_paused[(size_t)slot])
                                publish_progress(slot, E.rows(slot), std::min(std::max(E.progress(k, slot), 0), ni));
                int busy = 0, abort_rank = -1;
                for (int k = 0; k < G; k++) busy |= (grp[(size_t)k].inflight || grp[(size_t)k].active > 0 || grp[(size_t)k].npaused > 0) ? 1 : 0;
                rc = exchange(busy, 0, &busy_all, &abort_rank);
                if (rc) { aborted = true; return fail(rc); }
                if (abort_rank >= 0) { aborted = true; err = "rank " + std::to_string(abort_rank) + " aborted the sharded trace"; return fail(PNR_E_STATE); }
                if (all_done) break;
            }
            // ---- replay in seed order as far as the finished traces reach, push the new density to the engine
            r.touched.clear();
            auto final_replay = [&]() {
                while (frontier < n && !r.stopped) {
                    SeedRec &sr = rec[(size_t)frontier];
                    if (!sr.skipped && sr.have < 2) break;
                    if (!sr.skipped) { // (a skipped seed sits on a saturated voxel: the replay would skip it as well)
                        r.log_base = frontier;
                        r.add(&seeds[frontier], 1, sr.T, sr.xc.data());
                        std::vector<pnr_xest>().swap(sr.xc);
                        std::vector<int64_t>().swap(sr.vox);
                    }
                    frontier++;
                }
            };
            final_replay();
            // ---- tentative replay of everything known beyond the frontier (see the head of this file): pause / resume / end the
            // traces of THIS group (its staging buffers are free: its last steps have just been waited for)
            if (tentative && !r.stopped && frontier < n) {
                const auto tt0 = std::chrono::steady_clock::now();
                bool ended_any = false;
                pause_list.clear(); resume_list.clear(); // what this turn asks of the engine, sent once after the last round
                auto want_pause = [&](int slot) { // (a resume asked for earlier in this turn is simply taken back: the trace never left the device's list)
                    auto f = std::find(resume_list.begin(), resume_list.end(), slot);
                    if (f != resume_list.end()) resume_list.erase(f); else pause_list.push_back(slot);
                };
                auto want_resume = [&](int slot) {
                    auto f = std::find(pause_list.begin(), pause_list.end(), slot);
                    if (f != pause_list.end()) pause_list.erase(f); else resume_list.push_back(slot);
                };
                for (int round = 0; round < 2; round++) { // (records the pass itself produced are applied at once on one rank)
                    ov.begin();
                    st.tent_passes++;
                    bool exact = true; // every seed in front is complete: the overlay is what the final replay will see
                    const int64_t scan_hi = std::min<int64_t>(n, std::max<int64_t>(max_known + 1, frontier));
                    ended_any = false;
                    for (int64_t s = frontier; s < scan_hi; s++) {
                        SeedRec &sr = rec[(size_t)s];
                        if (sr.skipped) continue;
                        const pnr_seed &sd = seeds[s];
                        const int64_t sv = r.voxel(sd.x, sd.y, sd.z);
                        const bool seed_sat = r.den_at(sv) + ov.get(sv) >= r.prm.nodepervol; // the replay would not look at its traces
                        for (int dir = 0; dir < 2; dir++) {
                            const int slot = seed_slot[(size_t)(2 * s + dir)];
                            const bool done = (sr.got >> dir & 1) != 0;
                            const pnr_xest *X = nullptr;
                            int nr = 0;
                            const bool remote_paused = !done && slot < 0 && (sr.part >> dir & 1); // unfinished on its rank: the rows it has published
                            if (done) { X = sr.xc.data() + (size_t)dir * ni; nr = std::min(sr.T[dir], ni); }
                            else if (slot >= 0) { X = E.rows(slot); nr = slot_fresh[(size_t)slot] ? 0 : std::min(std::max(E.progress(slot_group[(size_t)slot], slot), 0), ni); }
                            else if (remote_paused) { X = sr.xc.data() + (size_t)dir * ni; nr = std::min(sr.prow[dir], ni); }
                            else { exact = false; continue; } // running elsewhere, or its record is on its way: nothing known
                            int cut = seed_sat ? 0 : -1;
                            if (cut < 0 && nr > 0) { // (a trace's recorded estimates never change: their voxels are computed once)
                                if (sr.vox.empty()) sr.vox.resize((size_t)2 * ni);
                                int64_t *vc = sr.vox.data() + (size_t)dir * ni;
                                for (int32_t &k = sr.nvox[dir]; k < nr; k++) vc[k] = r.voxel(X[k].x, X[k].y, X[k].z);
                            }
                            const int64_t *vc = sr.vox.empty() ? nullptr : sr.vox.data() + (size_t)dir * ni;
                            for (int i = 0; cut < 0 && i < nr; i++) {
                                const int64_t crd = vc[i];
                                if (i + 6 < nr) { r.cells.prefetch(vc[i + 6]); ov.prefetch(vc[i + 6]); }
                                if (r.soma_at(crd)) { cut = i; break; }
                                uint8_t &oc = ov.slot(crd);
                                if (r.den_at(crd) + oc >= r.prm.nodepervol) { cut = i; break; }
                                if (oc < 255) oc++;
                                st.tent_nodes++;
                                if (r.prm.vol > 1) {
                                    int64_t nb[26];
                                    const int cnt = density_neighbours(crd, r.W, r.H, r.L, r.prm.vol, nb);
                                    for (int q2 = 0; q2 < cnt; q2++) ov.add(nb[q2]);
                                }
                            }
                            if (done) continue;
                            if (remote_paused) {
                                // every seed in front complete and the published rows cut: the verdict is final for every rank alike -- the
                                // trace is complete with the rows it has (its owner ends it in its own pass and says so; that record is ignored)
                                if (exact && cut >= 0) { sr.T[dir] = nr; sr.got |= (uint8_t)(1 << dir); sr.part &= (uint8_t)~(1 << dir); sr.have++; ended_any = true; }
                                else exact = false;
                                continue;
                            }
                            const bool here = slot_group[(size_t)slot] == g; // (another group's staging may still be in use: it acts in its own turn)
                            if (cut >= 0) {
                                if (exact && here) { // not tentative: this IS the final replay's verdict -- end the trace, deliver its record
                                    const size_t at = outbox.size();
                                    outbox.resize(at + 4 + (size_t)nr * 8);
                                    outbox[at] = (int32_t)s; outbox[at + 1] = dir; outbox[at + 2] = nr; outbox[at + 3] = nr;
                                    if (nr > 0) std::memcpy(&outbox[at + 4], X, (size_t)nr * sizeof(pnr_xest));
                                    st.iters += nr; // (the iterations whose records were collected; with a lag the device steps a running trace up to `lag` times more before the pause lands)
                                    st.ended++;
                                    if (!slot_paused[(size_t)slot]) want_pause(slot); else q.npaused--;
                                    slot_paused[(size_t)slot] = 0;
                                    seed_slot[(size_t)(2 * s + dir)] = -1;
                                    slot_seed[(size_t)slot] = -1;
                                    q.busy.erase(std::find(q.busy.begin(), q.busy.end(), slot));
                                    q.held.push_back(slot); // (control() takes it off the device's list before this turn's admissions of THIS group)
                                    ended_any = true;
                                } else {
                                    if (here && !slot_paused[(size_t)slot]) {
                                        want_pause(slot); slot_paused[(size_t)slot] = 1; q.npaused++; st.paused++;
                                        publish_progress(slot, X, nr);
                                    }
                                    exact = false;
                                }
                            } else {
                                if (here && slot_paused[(size_t)slot]) { want_resume(slot); slot_paused[(size_t)slot] = 0; q.npaused--; st.resumed++; }
                                exact = false; // still running: what it will add is not known
                            }
                        }
                    }
                    if (!ended_any) break;
                    if (world == 1) { // (sharded, what this rank ended itself comes back with the next exchange)
                        if (!apply(outbox.data(), outbox.size())) { err = "malformed trace record"; return fail(PNR_E_STATE); }
                        outbox.clear();
                    }
                    final_replay();
                    if (r.stopped || frontier >= n) break;
                }
                if (!pause_list.empty() || !resume_list.empty()) {
                    rc = E.control(g, pause_list.data(), (int)pause_list.size(), resume_list.data(), (int)resume_list.size());
                    if (rc) { err = E.error(); return fail(rc); }
                    q.active += (int)resume_list.size() - (int)pause_list.size(); // (what control() leaves on the device's list)
                    if (q.active < 0) q.active = 0;
                    idle_turns = 0;
                }
                st.tent_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tt0).count();
            }
            if (!r.touched.empty()) {
                rc = E.density_update(r, g);
                if (rc) { err = E.error(); return fail(rc); }
            }
            // MAX_TRACE_COUNT (:2702) or the last seed replayed: whatever is still running is never looked at.  (Sharded, this rank
            // says so in its next exchange and leaves with the others.)
            if (world == 1 && (r.stopped || frontier >= n)) break;
        }
        if (r.stopped || frontier >= n) { // (sharded, waiting for the others: nothing more to admit or to step)
            if (world > 1 && !sync_turn) continue;
            if (world > 1) { idle_turns = busy_all > 0 ? 0 : idle_turns + 1; if (idle_turns > STALL_TURNS) { E.drain(); err = "ranks did not finish together"; return PNR_E_STATE; } }
            continue;
        }
        // ---- admission into this group: this rank's seeds inside the lookahead
        const int64_t lim = frontier + std::max<int64_t>(look0, frontier * look_pct / 100);
        int m = 0, m_max = 2 * NT;
        if (G > 1) { // keep the groups the same size: this one is filled up to its share of what the window will hold
            const int64_t ahead = next < lim ? (std::min<int64_t>(lim, n) - next + world - 1) / world : 0;
            const int64_t room = std::min<int64_t>((int64_t)(free_slots.size() + q.held.size()) / 2, ahead);
            int64_t total = 2 * std::max<int64_t>(room, 0);
            int least = q.active;
            for (int k = 0; k < G; k++) { total += grp[(size_t)k].active; least = std::min(least, grp[(size_t)k].active); }
            m_max = (int)std::max<int64_t>(0, (total + G - 1) / G - q.active);
            if (q.active <= least) m_max = std::max(m_max, 2); // the smallest group can always take a seed
        }
        if (o.target > 0) { // ... and only up to this group's share of the running traces asked for (at least one seed when it has none)
            // The target is the MEAN over the steps of a poll: where most traces end or are paused within a poll (late seeds live a few
            // iterations), the poll starts with up to twice the share.
            if (q.last_start > 0) q.keep = 0.5 * q.keep + 0.5 * std::min(1.0, (double)q.active / q.last_start);
            // Late seeds live a few iterations: launches shrink to a few dozen traces, where two overlapping groups only share the
            // per-launch floors.  Then everything new goes to group 0 (the others run out), until most traces survive a poll again.
            if (G > 1 && world == 1 && o.concentrate && g == 0) {
                const double enter = (o.concentrate == 1 ? 60 : o.concentrate) / 100.0; // (values above 1: the threshold in percent)
                if (!single && q.keep < enter) single = true;
                else if (single && q.keep > std::min(0.97, enter + 0.25)) single = false;
            }
            const int share = single ? (g == 0 ? o.target : 0) : std::max(2, (o.target + G - 1) / G);
            const int goal = o.overfill ? (int)std::min(2.0 * share, share * 2.0 / (1.0 + q.keep) + 0.5) : share;
            if (single && g == 0) m_max = 2 * NT; // (the groups are not kept the same size)
            m_max = std::min(m_max, std::max(0, goal - q.active));
        }
        new_slots.clear(); new_s6.clear();
        while (m + 2 <= m_max && next < n && free_slots.size() + q.held.size() >= 2 && next < lim) {
            const pnr_seed &sd = seeds[next];
            if (r.seed_saturated(sd)) {
                outbox.insert(outbox.end(), {(int32_t)next, -1, 0, 0});
                st.skipped++;
                next += world;
                continue;
            }
            for (int dir = 0; dir < 2; dir++) {
                int slot; // (slots this group's own control() frees first: nobody else may have them yet)
                if (!q.held.empty()) { slot = q.held.back(); q.held.pop_back(); }
                else { slot = free_slots.back(); free_slots.pop_back(); }
                slot_seed[(size_t)slot] = next; slot_dir[(size_t)slot] = dir; slot_group[(size_t)slot] = g; slot_paused[(size_t)slot] = 0; slot_fresh[(size_t)slot] = 1; slot_pub[(size_t)slot] = 0;
                seed_slot[(size_t)(2 * next + dir)] = slot;
                q.busy.push_back(slot);
                new_slots.push_back(slot);
                const float sg = dir ? -1.f : 1.f; // trackNeg starts from the negated seed direction (tracker.cpp:819-823)
                const float a[6] = {sd.x, sd.y, sd.z, sg * sd.vx, sg * sd.vy, sg * sd.vz};
                new_s6.insert(new_s6.end(), a, a + 6);
                m++;
            }
            st.launched++;
            if (next > max_known) max_known = next;
            next += world;
        }
        if (o.timing) { // why the admission of this turn ended
            if (next >= n) dbg_end[0]++; else if (next >= lim) dbg_end[1]++; else if (free_slots.size() < 2) dbg_end[2]++; else dbg_end[3]++;
            dbg_act += q.active + m; dbg_turns++; dbg_paused += q.npaused; dbg_free += (long long)free_slots.size();
        }
        if (m > 0) {
            rc = E.admit(g, new_slots.data(), new_s6.data(), m);
            if (rc) { err = E.error(); return fail(rc); }
            q.active += m;
        }
        if (q.active > 0) {
            q.last_start = q.active;
            // the host's share of the next poll is covered by another group's steps, or by the last steps of this one
            bool alone = true;
            for (int k = 0; k < G; k++) alone = alone && (k == g || (grp[(size_t)k].active == 0 && !grp[(size_t)k].inflight));
            q.lag = o.lag >= 0 ? std::min(o.lag, o.poll - 1) : (alone ? o.poll / 2 : std::min(1, o.poll - 1));
            rc = E.launch(g, q.active, o.poll, q.lag);
            if (rc) { err = E.error(); return fail(rc); }
            if (o.timing) {
                dbg_size[q.active < 16 ? 0 : q.active < 32 ? 1 : q.active < 64 ? 2 : q.active < 128 ? 3 : 4]++;
                if (next >= n) { dbg_drain++; dbg_drain_act += q.active; }
            }
            st.steps += o.poll;
            q.inflight = true;
        } else {
            E.idle(g); // (nothing of this group runs until its next launch: the other groups have the GPU to themselves)
        }
        rc = E.end_turn(g);
        if (rc) { err = E.error(); return fail(rc); }
        // ---- nothing running anywhere and nothing admitted: the frontier cannot move any more
        if (world > 1 && !sync_turn) continue; // the idle count only moves in the turns every rank synchronises in
        bool any = frontier > frontier_was;
        if (world > 1) {
            any = any || busy_all > 0; // only what every rank knows: all ranks count the same idle turns
        } else {
            any = any || m > 0 || !outbox.empty();
            for (int k = 0; k < G; k++) any = any || grp[(size_t)k].inflight;
            // (paused traces are no progress by themselves: the pass resumes or ends the ones at the frontier within a rotation)
        }
        idle_turns = any ? 0 : idle_turns + 1;
        if (idle_turns > STALL_TURNS) {
            E.drain();
            err = "trace scheduler stalled at seed " + std::to_string((long long)frontier) + " of " + std::to_string((long long)n);
            return PNR_E_STATE; // every rank sees the same counters and stops in the same turn
        }
    }
    E.drain(); // what is still running is never looked at again, but it writes into buffers that outlive this call
    if (o.timing)
        fprintf(stderr, "[pnr trace] rank %d/%d: %lld seeds, window %d slots, lookahead max(%d, %d%%), target %d running, %lld steps, %lld polls, %lld iterations here, "
                        "%lld exchanges (%lld carried), %zu nodes; tentative replay: %lld passes, %lld nodes, %lld pauses, %lld resumed, %lld ended by the host, %.1f ms of host time; %.1f ms blocked waiting for the GPU\n",
                rank, world, (long long)n, NT, o.look0, o.look_pct, o.target, (long long)st.steps,
                (long long)st.polls, (long long)st.iters, (long long)st.exchanges, (long long)st.carried, r.nodes.size(), (long long)st.tent_passes,
                (long long)st.tent_nodes, (long long)st.paused, (long long)st.resumed, (long long)st.ended, st.tent_ms, st.wait_ms);
    if (o.timing)
        fprintf(stderr, "[pnr trace] launches of < 16 / < 32 / < 64 / < 128 / more traces: %lld / %lld / %lld / %lld / %lld; after the last admission: %lld launches, %.1f traces each\n",
                dbg_size[0], dbg_size[1], dbg_size[2], dbg_size[3], dbg_size[4], dbg_drain, dbg_drain ? (double)dbg_drain_act / dbg_drain : 0.0);
    if (o.timing && dbg_turns)
        fprintf(stderr, "[pnr trace] admissions ended by: no seeds left %lld, lookahead %lld, no free slot %lld, target / group share %lld; per turn: %.1f launched, %.1f paused in the group, %.1f free slots\n",
                dbg_end[0], dbg_end[1], dbg_end[2], dbg_end[3], (double)dbg_act / dbg_turns, (double)dbg_paused / dbg_turns, (double)dbg_free / dbg_turns);
    if (stats) *stats = st;
    return PNR_OK;
}

} // namespace pnr
