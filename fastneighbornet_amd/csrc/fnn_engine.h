// fnn_engine.h -- host side of the engine, shared by the HIP backend
// (fnn_hip.hip -> libfastnn_hip.so, the product) and by the CPU emulation backend
// used only by tests (tests/emu/fnn_emu.cpp).  The template parameter B supplies
// memory management and the per-event kernel sequence; everything here is
// integer / control logic: handle lifecycle, matrix upload, the event loop,
// and expandNodes (NetMakerOriginal.java:246-325), which is Theta(n) pointer
// relinking on the host.
#ifndef FNN_ENGINE_H
#define FNN_ENGINE_H

#include <chrono>
#include <climits>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/fastnn.h"
#include "fnn_core.h"

namespace fnn {

static_assert(sizeof(Event) == sizeof(fnn_event), "event layout");

inline thread_local std::string g_last_error;
inline int32_t fail(int32_t code, const std::string& msg) {
    g_last_error = msg;
    return code;
}

inline double now_s() {
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

inline int64_t round_up(int64_t v, int64_t a) { return (v + a - 1) / a * a; }
// workgroups that may write a pair of partial sums into Dev::upart (k_update's grid)
inline size_t upart_capacity(int32_t n) { size_t g = (size_t)(n > 0 ? n : 1) / 256 + 2; return g < 64 ? 64 : g; }

template <class B>
class Engine {
  public:
    B be;
    Dev dev{};
    int32_t n = 0;
    int64_t ld = 0, ldh = 0, nrows = 0;
    fnn_opts opts{};
    bool have_matrix = false, begun = false, ended = false;
    State hst{};           // host copy of the device state
    int32_t m_bound = 0;   // upper bound of num_active known to the host
    fnn_stats stats{};
    std::vector<fnn_event> events;  // trajectory of the last run
    std::vector<Agg3Rec> agglog;
    int32_t last3[3] = {0, 0, 0};
    int32_t batch = 128;   // events enqueued between host round trips in run() (round 4: 128 instead of 64, -0.7 % at 32768 taxa)
    int64_t ev_counter = 0;  // events enqueued since begin(): drives the schedule of the lookahead windows' base scans
    int64_t sched_at = 0;    // event count at which the host expects the open window to have served its K events
    bool state_seen = false; // the host has read the device state since begin() (nonneg / screen_ok are the prep kernel's verdict)
    // several GPUs: 0 = single, 1 = RCCL all-gather on the stream, 2 = host callback (tests)
    int32_t comm_mode = 0, world = 1, rank = 0;
    fnn_allgather_fn host_fn = nullptr;
    void* host_ctx = nullptr;
    int32_t* d_status = nullptr;  // several ranks over RCCL: every rank's State::error, gathered before each host round trip
    int fault_error_rank = -1, fault_error_event = -1;  // test hook FNN_FAULT_ERROR="rank:event": an error code on one rank only

    bool relaxed() const { return opts.mode == FNN_MODE_RELAXED; }

    int32_t create(int32_t n_, const fnn_opts* o) {
        if (n_ < 0) return fail(FNN_EINVAL, "fnn_create: n < 0");
        n = n_;
        if (o) opts = *o;
        if (opts.mode != FNN_MODE_CANONICAL && opts.mode != FNN_MODE_RELAXED) return fail(FNN_EINVAL, "fnn_create: unknown mode");
        int32_t rc = be.open(opts.device);
        if (rc != FNN_OK) return rc;
        be.set_problem_size(n);
        // padded geometry: rows to a multiple of the scan tile height, row stride to a
        // multiple of the tile width plus 32 doubles so that column sweeps (stride ld)
        // do not hammer one HBM channel
        nrows = round_up(n > 0 ? n : 1, B::kRowPad);
        ld = round_up(n > 0 ? n : 1, B::kColPad) + 32;
        // the bf16 copy has its own row stride (FNN_LDH_PAD: shifts between 64 B and 4.5 KB per row were measured in the
        // screening pass at n = 32768 - all within 4.5-4.85 TB/s; a 512-B shift reached 5.1 TB/s only together with a
        // 2-KB shift of the fp64 rows, which slows the event chain by 7 %: no robust gain, so both use the same padding)
        ldh = ld;
        if (const char* e = std::getenv("FNN_LDH_PAD")) { int v = std::atoi(e); if (v >= 32 && v % 8 == 0 && v <= 65536) ldh = round_up(n > 0 ? n : 1, B::kColPad) + v; }
        dev.n = n;
        dev.fault_event = -1;
        dev.ld = ld;
        dev.ldh = ldh;
        dev.cstride = round_up(n > 0 ? n : 1, CH_SC);
        dev.rank = 0;
        dev.world = 1;
        dev.gather = 0;
        size_t nn = (size_t)(n > 0 ? n : 1);
        if (!(dev.D = (double*)be.alloc(sizeof(double) * (size_t)nrows * (size_t)ld)) ||
            !(dev.Sx = (double*)be.alloc(sizeof(double) * (nn + 8))) ||
            !(dev.sid = (int32_t*)be.alloc(sizeof(int32_t) * (nn + 8))) ||
            !(dev.spos = (int32_t*)be.alloc(sizeof(int32_t) * (nn + 8))) ||
            !(dev.pslot = (int32_t*)be.alloc(sizeof(int32_t) * (nn + 8))) ||
            !(dev.chain = (double*)be.alloc(sizeof(double) * CHAIN_BUFS * (size_t)dev.cstride)) ||
            !(dev.plan = (uint64_t*)be.alloc(sizeof(uint64_t) * PLAN_WORDS)) ||
            !(dev.recs = (Cand*)be.alloc(sizeof(Cand) * be.max_records(n))) ||
            !(dev.rchk = (uint64_t*)be.alloc(sizeof(uint64_t) * (2048 + 8))) ||
            !(dev.T = (double*)be.alloc(sizeof(double) * (nn + 8))) ||
            !(dev.srec = (float*)be.alloc(sizeof(float) * (2 * (size_t)screen_unit_count(n > 0 ? n : 1) + 16))) ||
            !(dev.stile = (float*)be.alloc(sizeof(float) * (2 * (size_t)screen_unit_count(n > 0 ? n : 1) / 4 + 16))) ||
            !(dev.shit = (uint64_t*)be.alloc(sizeof(uint64_t) * ((size_t)screen_unit_count(n > 0 ? n : 1) + 16))) ||
            !(dev.clist = (int32_t*)be.alloc(sizeof(int32_t) * SCR_CAP)) ||
            !(dev.islot = (int32_t*)be.alloc(sizeof(int32_t) * (3 * nn + 8))) ||
            !(dev.cstamp = (int32_t*)be.alloc(sizeof(int32_t) * (3 * nn + 8))) ||
            !(dev.tpairs = (int32_t*)be.alloc(sizeof(int32_t) * LA_REC_INTS * LA_PCAP)) ||
            !(dev.fresh = (int32_t*)be.alloc(sizeof(int32_t) * 3 * LA_KMAX)) ||
            !(dev.ticket = (uint32_t*)be.alloc(sizeof(uint32_t) * 32 * 72)) ||
            !(dev.lacnt = (int32_t*)be.alloc(256)) ||
            !(dev.ticks = (int64_t*)be.alloc(sizeof(int64_t) * TICK_WORDS)) ||
            !(dev.lalog = (double*)be.alloc(sizeof(double) * 5 * LA_LOGCAP)) ||
            // k_update (deferred close) leaves {sum, sum of magnitudes} per workgroup: ceil(m / 256) + 1 workgroups;
            // -> sized from the update grid like rxpart
            !(dev.upart = (double*)be.alloc(sizeof(double) * 4 * upart_capacity(n))) ||
            !(dev.gsend = (Cand*)be.alloc(sizeof(Cand) * GATHER_RECS)) ||
            (relaxed() && (!(dev.rperm = (int32_t*)be.alloc(sizeof(int32_t) * (nn + 8))) ||
                           !(dev.rl_stamp = (int32_t*)be.alloc(sizeof(int32_t) * (nn + 8))) ||
                           !(dev.rl_cnt = (int32_t*)be.alloc(sizeof(int32_t) * (nn + 8))) ||
                           !(dev.rl_list = (int32_t*)be.alloc(sizeof(int32_t) * RL_TIES * (nn + 8))) ||
                           !(dev.rl_val = (double*)be.alloc(sizeof(double) * (nn + 8))) ||
                           !(dev.rl_mail = (uint64_t*)be.alloc(sizeof(uint64_t) * RL_MAIL_WORDS)))) ||
            !(dev.grecv = (Cand*)be.alloc(sizeof(Cand) * GATHER_RECS * 64)) ||
            !(dev.st = (State*)be.alloc(sizeof(State))) ||
            !(dev.evlog = (Event*)be.alloc(sizeof(Event) * (nn + 8))) ||
            !(dev.agglog = (Agg3Rec*)be.alloc(sizeof(Agg3Rec) * (nn + 8))))
            return fail(FNN_ENOMEM, "fnn_create: device allocation failed (" + be.err() + ")");
        dev.H = nullptr;
        if (!opts.disable_screen && !relaxed() && n >= be.screen_min_n()) {  // (Relaxed mode: no windows, plain scans at the end)
            if (!(dev.H = (uint16_t*)be.alloc(sizeof(uint16_t) * (size_t)nrows * (size_t)ldh)))
                return fail(FNN_ENOMEM, "fnn_create: device allocation of the bf16 copy failed (" + be.err() + ")");
        }
        // zero the padding once so that stray loads never see signalling patterns
        if (be.memset(dev.D, 0, sizeof(double) * (size_t)nrows * (size_t)ld) != FNN_OK)
            return fail(FNN_EHIP, "fnn_create: memset failed (" + be.err() + ")");
        return FNN_OK;
    }

    void destroy() {
        be.free(dev.D); be.free(dev.Sx); be.free(dev.sid); be.free(dev.spos); be.free(dev.pslot);
        be.free(dev.chain); be.free(dev.plan); be.free(dev.recs); be.free(dev.rchk); be.free(dev.T); be.free(dev.gsend); be.free(dev.grecv); be.free(dev.wsend); be.free(dev.wrecv); be.free(d_status); d_status = nullptr; be.free(dev.H); be.free(dev.srec); be.free(dev.stile); be.free(dev.clist); be.free(dev.shit); be.free(dev.islot); be.free(dev.cstamp); be.free(dev.tpairs); be.free(dev.fresh); be.free(dev.ticket); be.free(dev.lacnt); be.free(dev.rperm); be.free(dev.rl_stamp); be.free(dev.rl_cnt); be.free(dev.rl_list); be.free(dev.rl_val); be.free(dev.rl_mail); be.free(dev.ticks); be.free(dev.lalog); be.free(dev.upart); be.free(dev.st); be.free(dev.evlog); be.free(dev.agglog);
        dev = Dev{};
        be.close();
    }

    int32_t set_rows(int32_t row0, int32_t cnt, const double* rows, int64_t ld_in) {
        if (!rows || row0 < 0 || cnt < 0 || (int64_t)row0 + cnt > n || ld_in < n)
            return fail(FNN_EINVAL, "fnn_set_rows: bad arguments");
        if (cnt == 0) return FNN_OK;
        if (be.h2d_2d(dev.D + (int64_t)row0 * ld, ld, rows, ld_in, n, cnt) != FNN_OK)
            return fail(FNN_EHIP, "fnn_set_rows: copy failed (" + be.err() + ")");
        if ((int64_t)row0 + cnt == n) have_matrix = true;  // rows are expected in order
        begun = ended = false;
        return FNN_OK;
    }

    // DistancesAndNames' packed strict upper triangle -> both triangles on the device
    // (FastNN.java:307-312 does this expansion on the host)
    int32_t set_packed_upper(const double* packed) {
        if (!packed && n > 1) return fail(FNN_EINVAL, "fnn_set_packed_upper: bad arguments");
        const int64_t cap = (int64_t)16 << 20;  // entries per staged chunk (128 MiB); a row has < 2^15.. entries
        for (int64_t row0 = 0; row0 + 1 < n;) {
            int64_t cnt = 0, entries = 0;
            while (row0 + cnt + 1 < n && (cnt == 0 || entries + (n - 1 - (row0 + cnt)) <= cap)) {
                entries += n - 1 - (row0 + cnt);
                cnt++;
            }
            const int64_t p0 = packed_row_base(n, row0) + row0 + 1;
            if (be.unpack_rows(dev, packed + p0, p0, entries, (int32_t)row0, (int32_t)cnt) != FNN_OK)
                return fail(FNN_EHIP, "fnn_set_packed_upper: copy failed (" + be.err() + ")");
            row0 += cnt;
        }
        if (n > 0 && be.launch_mirror(dev) != FNN_OK)
            return fail(FNN_EHIP, "fnn_set_packed_upper: launch failed (" + be.err() + ")");
        have_matrix = true;
        begun = ended = false;
        return FNN_OK;
    }

    int32_t set_matrix_device(const double* dmat, int64_t ld_in) {
        if (!dmat || ld_in < n) return fail(FNN_EINVAL, "fnn_set_matrix_device: bad arguments");
        if (n > 0 && be.d2d_2d(dev.D, ld, dmat, ld_in, n, n) != FNN_OK)
            return fail(FNN_EHIP, "fnn_set_matrix_device: copy failed (" + be.err() + ")");
        have_matrix = true;
        begun = ended = false;
        return FNN_OK;
    }

    int32_t get_matrix(double* out, int64_t ld_out) {
        if (!out || ld_out < n) return fail(FNN_EINVAL, "fnn_get_matrix: bad arguments");
        if (!have_matrix) return fail(FNN_ESTATE, "fnn_get_matrix: no matrix resident (upload or fnn_synth first; a run consumes it)");
        if (n > 0 && (be.sync() != FNN_OK || be.d2h_2d(out, ld_out, dev.D, ld, n, n) != FNN_OK))
            return fail(FNN_EHIP, "fnn_get_matrix: download failed (" + be.err() + ")");
        return FNN_OK;
    }

    int32_t synth(uint64_t seed, int32_t dist) {
        if (dist != 0 && dist != 1) return fail(FNN_EINVAL, "fnn_synth: dist must be 0 or 1");
        if (n > 0 && be.launch_synth(dev, seed, dist) != FNN_OK)
            return fail(FNN_EHIP, "fnn_synth: launch failed (" + be.err() + ")");
        have_matrix = true;
        begun = ended = false;
        return FNN_OK;
    }

    int32_t pull_state() {
        if (be.d2h(&hst, dev.st, sizeof(State)) != FNN_OK)
            return fail(FNN_EHIP, "state download failed (" + be.err() + ")");
        if (hst.error) return fail(FNN_ESTATE, "engine reached an unreachable branch, code " + std::to_string(hst.error));
        return FNN_OK;
    }
    // Several ranks: an error in ONE rank's device state (codes 12 / 13: an exchange block of another event; any internal-consistency
    // code) must stop EVERY rank at the same host round trip - a rank that stopped alone would leave the others waiting in their next
    // all-gather for ever.  So the ranks exchange their error word whenever the host looks at the state: 4 bytes per
    // rank, once per batch of events.  enqueue_status_exchange() goes on the stream BEFORE the synchronisation (RCCL transport);
    // pull_state_ranks() after it.
    int32_t enqueue_status_exchange() {
        if (fault_error_rank == rank && ev_counter > fault_error_event && fault_error_event >= 0) {
            const int32_t code = 99;
            fault_error_event = -1;
            if (be.sync() != FNN_OK || be.h2d((uint8_t*)dev.st + offsetof(State, error), &code, sizeof(code)) != FNN_OK)
                return fail(FNN_EHIP, "fault injection failed (" + be.err() + ")");
        }
        if (comm_mode != 1) return FNN_OK;
        if (be.allgather_bytes_on_stream((const uint8_t*)dev.st + offsetof(State, error), d_status, sizeof(int32_t)) != FNN_OK)
            return fail(FNN_ERCCL, "all-gather of the ranks' status failed (" + be.err() + ")");
        return FNN_OK;
    }
    int32_t pull_state_ranks() {
        if (comm_mode == 0) return pull_state();
        if (be.d2h(&hst, dev.st, sizeof(State)) != FNN_OK) return fail(FNN_EHIP, "state download failed (" + be.err() + ")");
        std::vector<int32_t> codes((size_t)world, 0);
        if (comm_mode == 1) {
            if (be.d2h(codes.data(), d_status, sizeof(int32_t) * (size_t)world) != FNN_OK)
                return fail(FNN_EHIP, "status download failed (" + be.err() + ")");
        } else {
            const int32_t mine = hst.error;
            if (!host_fn || host_fn(host_ctx, &mine, codes.data(), (int32_t)sizeof(int32_t)) != 0)
                return fail(FNN_ERCCL, "host all-gather callback failed");
        }
        for (int32_t r = 0; r < world; r++)
            if (codes[(size_t)r])
                return fail(FNN_ESTATE, "engine reached an unreachable branch, code " + std::to_string(codes[(size_t)r]) + " on rank " + std::to_string(r) +
                                            " (every rank stops here; this is rank " + std::to_string(rank) + ")");
        if (hst.error) return fail(FNN_ESTATE, "engine reached an unreachable branch, code " + std::to_string(hst.error));
        return FNN_OK;
    }

    // runNeighborNet up to and including initialize() (NetMakerOriginal.java:141-159)
    int32_t begin() {
        if (!have_matrix) return fail(FNN_ESTATE, "fnn_begin: no matrix uploaded");
        stats = fnn_stats{};
        events.clear();
        agglog.clear();
        double t0 = now_s();
        if (n > 0 && opts.validate) {
            int32_t bad = 0;
            if (be.launch_validate(dev, &bad) != FNN_OK)
                return fail(FNN_EHIP, "fnn_begin: validate failed (" + be.err() + ")");
            if (bad) return fail(FNN_EINVAL, "fnn_begin: matrix must be finite, symmetric, with zero diagonal");
        }
        std::memset(&hst, 0, sizeof(hst));
        hst.n = n; hst.m = n; hst.c = n; hst.P = 0; hst.num_nodes = n;
        hst.done = (n <= 3) ? 1 : 0;  // :133-140
        hst.nonneg = 1;                // cleared by the prep kernel if a negative entry exists
        hst.record_events = opts.record_events ? 1 : 0;
        hst.ev_timed = 1;
        hst.force_exact_rx = opts.force_exact_rx ? 1 : 0;
        if (relaxed()) {  // NeighborNetLocal's constructor (:25-32)
            hst.rl_on = 1;
            hst.rl_min = opts.relaxed_min_active > 0 ? opts.relaxed_min_active : 1024;
            hst.rl_first = 0;      // (:177-183 done here: identity permutation, top = ntax - 1)
            hst.rl_top = n - 1;
            {
                std::vector<int32_t> iota((size_t)(n > 0 ? n : 1));
                for (size_t i = 0; i < iota.size(); i++) iota[i] = (int32_t)i;
                if (be.h2d(dev.rperm, iota.data(), sizeof(int32_t) * iota.size()) != FNN_OK)
                    return fail(FNN_EHIP, "fnn_begin: upload failed (" + be.err() + ")");
            }
            hst.rl_rng = JavaRandom::scramble(((uint64_t)opts.relaxed_seed_hi << 32) | (uint64_t)opts.relaxed_seed_lo);
            if (be.memset(dev.rl_stamp, 0, sizeof(int32_t) * ((size_t)(n > 0 ? n : 1) + 8)) != FNN_OK ||
                be.memset(dev.rl_mail, 0, sizeof(uint64_t) * RL_MAIL_WORDS) != FNN_OK)
                return fail(FNN_EHIP, "fnn_begin: memset failed (" + be.err() + ")");
        }
        // lookahead windows (fnn_core.h "Lookahead"): single rank with a screening copy
        {
            // default window length: 16 + n / 1024 events (48 at n = 32768; shorter windows for smaller problems, whose
            // minima move faster relative to the spread of Q - measured optimum at 8192 / 16384 / 32768 taxa)
            int32_t Kdef = 16 + n / 1024;
            if (Kdef > 64) Kdef = 64;
            hst.la_Kcur = 0;
            int32_t K = opts.lookahead < 0 ? 0 : (opts.lookahead > 0 ? opts.lookahead : Kdef);
            // (round 4, after k_emit got cheaper: 49152 wanted pairs instead of 32768 - a fifth of the windows that could not certify -
            //  and window length 16 + m / 512 instead of 16 + m / 1024 below the cap: 1.372 s instead of 1.397 s at 32768 taxa)
            int32_t target = opts.lookahead_pairs > 0 ? opts.lookahead_pairs : 49152;
            hst.la_kbase = 16;
            hst.la_kdiv = 512;
            if (const char* e = std::getenv("FNN_LA_KBASE")) { int v = std::atoi(e); if (v >= 1 && v <= 64) hst.la_kbase = v; }
            if (const char* e = std::getenv("FNN_LA_KDIV")) { int v = std::atoi(e); if (v >= 64) hst.la_kdiv = v; }
            if (const char* e = std::getenv("FNN_LA_K")) K = std::atoi(e);
            if (const char* e = std::getenv("FNN_LA_TARGET")) target = std::atoi(e);
            if (K > LA_KMAX) K = LA_KMAX;
            if (target < 1) target = 1;
            if (target > LA_PCAP) target = LA_PCAP;
            // (several ranks: windows stay on - only the base scans are sharded and exchanged, see enqueue_event)
            hst.la_on = (dev.H && K > 0) ? 1 : 0;
            hst.la_K = K > 0 ? K : 0;
            hst.la_target = target;
            hst.la_min_m = be.screen_min_m;
            be.set_relaxed(hst.rl_on ? hst.rl_min : 0);
            dev.la = hst.la_on;
            dev.wx = (comm_mode != 0 && hst.la_on) ? 1 : 0;
            dev.strict = dev.wx;
            dev.fault_event = -1;
            dev.plan_ticks = std::getenv("FNN_TICKS") ? 1 : 0;
            if (const char* e = std::getenv("FNN_FAULT_GIVEUP")) {  // test hook: "rank:event"
                int fr = -1, fe = -1;
                if (std::sscanf(e, "%d:%d", &fr, &fe) == 2 && fr == rank) dev.fault_event = fe;
            }
            ev_counter = 0;
            sched_at = 0;
            hst.la_pcap = LA_PCAP;
            if (const char* e = std::getenv("FNN_LA_PCAP")) { int v = std::atoi(e); if (v >= 1 && v <= LA_PCAP) hst.la_pcap = v; }
        }
        if (be.h2d(dev.st, &hst, sizeof(State)) != FNN_OK)
            return fail(FNN_EHIP, "fnn_begin: state upload failed (" + be.err() + ")");
        if (be.memset(dev.islot, 0xFF, sizeof(int32_t) * (3 * (size_t)(n > 0 ? n : 1) + 8)) != FNN_OK ||
            be.memset(dev.cstamp, 0, sizeof(int32_t) * (3 * (size_t)(n > 0 ? n : 1) + 8)) != FNN_OK ||
            be.memset(dev.ticket, 0, sizeof(uint32_t) * 32 * 72) != FNN_OK || be.memset(dev.plan, 0, sizeof(uint64_t) * PLAN_WORDS) != FNN_OK || be.memset(dev.lacnt, 0, 256) != FNN_OK || be.memset(dev.ticks, 0, sizeof(int64_t) * TICK_WORDS) != FNN_OK)
            return fail(FNN_EHIP, "fnn_begin: memset failed (" + be.err() + ")");
        if (n > 3) {
            // max |D| (error bounds of the screening pass and of the certified 4-candidate choice), the bf16 copy if wanted
            if (be.launch_prep_screen(dev, nrows) != FNN_OK)
                return fail(FNN_EHIP, "fnn_begin: bf16 copy failed (" + be.err() + ")");
            if (be.launch_init(dev) != FNN_OK || be.sync() != FNN_OK)
                return fail(FNN_EHIP, "fnn_begin: init failed (" + be.err() + ")");
            // the prep kernel's verdict on the matrix (nonneg, screen_ok, max |D|): the schedule of the scans depends on it
            const int32_t rcp = pull_state();
            if (rcp != FNN_OK) return rcp;
            state_seen = true;
            be.screen_off = false;
            // A matrix with a negative entry (or one the bf16 bound does not cover) can open no window, and its mixed-sign
            // brackets admit so many candidate units that the screened scan costs more than the plain one (measured, round 3:
            // 3.9 ms per event at 32768 taxa against 0.4 ms): such a run takes the plain fp64 scan for every event.
            // (FNN_SCREEN_MIN_N / FNN_SCREEN_MIN_M - the tests' way to force the screening pass on - keep it.)
            if (dev.H && (!hst.nonneg || !hst.screen_ok) && !be.keep_generic_screen() && !std::getenv("FNN_SCREEN_MIN_N") && !std::getenv("FNN_SCREEN_MIN_M")) {
                hst.la_on = 0;
                dev.la = 0; dev.wx = 0; dev.strict = 0;
                be.screen_off = true;
                if (be.h2d(dev.st, &hst, sizeof(State)) != FNN_OK) return fail(FNN_EHIP, "fnn_begin: state upload failed (" + be.err() + ")");
            }
        }
        m_bound = n;
        have_matrix = false;  // consumed
        begun = true;
        ended = (n <= 3);
        stats.t_init_s = now_s() - t0;
        return FNN_OK;
    }

    int32_t comm_set(int32_t mode, int32_t world_, int32_t rank_) {
        if (world_ < 1 || world_ > 64 || rank_ < 0 || rank_ >= world_)
            return fail(FNN_EINVAL, "fnn_comm_init: need 1 <= world <= 64 and 0 <= rank < world");
        if (relaxed() && world_ > 1) return fail(FNN_EINVAL, "fnn_comm_init: the Relaxed mode runs on one GPU (its search is a chain of row minima)");
        // world == 1 normally needs no exchange; FNN_COMM_FORCE=1 keeps the exchange path on
        // (lets a one-GPU box exercise the RCCL plumbing end to end)
        comm_mode = (world_ > 1 || std::getenv("FNN_COMM_FORCE")) ? mode : 0;
        world = world_;
        rank = rank_;
        dev.world = world_;
        dev.rank = rank_;
        dev.gather = comm_mode != 0 ? 1 : 0;
        be.free(dev.wsend); be.free(dev.wrecv); be.free(d_status);
        dev.wsend = dev.wrecv = nullptr;
        d_status = nullptr;
        if (comm_mode != 0) {
            const size_t bb = (size_t)wx_block_bytes(world_);
            if (!(dev.wsend = (uint8_t*)be.alloc(bb)) || !(dev.wrecv = (uint8_t*)be.alloc(bb * (size_t)world_)) ||
                !(d_status = (int32_t*)be.alloc(sizeof(int32_t) * 64)))
                return fail(FNN_ENOMEM, "fnn_comm_init: device allocation failed (" + be.err() + ")");
        }
        fault_error_rank = fault_error_event = -1;
        if (const char* e = std::getenv("FNN_FAULT_ERROR")) {  // test hook: "rank:event"
            if (std::sscanf(e, "%d:%d", &fault_error_rank, &fault_error_event) != 2) fault_error_rank = fault_error_event = -1;
        }
        return FNN_OK;
    }

    // one event: scan (+ exchange of the per-rank candidates) + the rest of the sequence.
    // after a state download: when will the open window have served its K events?
    void resync_schedule() {
        if (!dev.la) return;
        sched_at = (hst.la_valid && !hst.stall) ? ev_counter + (hst.la_Kcur + 1 > hst.la_k ? hst.la_Kcur + 1 - hst.la_k : 0) : ev_counter;
    }
    // A rank contributes nper candidate records (1 after a local reduction, or the scan's
    // GATHER_RECS per-workgroup records as they are).
    // force_sched: -1 = follow the schedule, 0 / 1 = the caller knows whether this event scans
    int32_t enqueue_event(int force_sched = -1) {
        // lookahead windows: a new window is opened (base scan) at the first two events (the second
        // one knows the previous minimum) and then every la_K events; in between a scan only runs
        // if the window fails, which the device finds out by itself
        const int64_t cnt = ev_counter++;
        // (a matrix with a negative entry, or one the bf16 bound does not cover, can never open a window - la_prepare_base -: every
        //  event scans, and the host must SCHEDULE that scan; left to the window schedule such a run stalled 63 of 64 launch
        //  sequences: 310 s instead of 3 s at 16384 taxa, found with the round-3 input classes)
        const bool no_windows = dev.la && state_seen && (!hst.nonneg || !hst.screen_ok);
        bool sched = !dev.la || no_windows || cnt <= 1 || cnt >= sched_at;
        if (force_sched >= 0) sched = force_sched != 0;
        if (sched) sched_at = cnt + (hst.la_Kcur > 0 ? hst.la_Kcur : hst.la_K) + 1;
        if (comm_mode == 0) return be.launch_event(dev, m_bound, sched) == FNN_OK ? FNN_OK : fail(FNN_EHIP, "launch failed (" + be.err() + ")");
        if (dev.wx) {
            // Several ranks with lookahead windows.  Every rank holds the whole matrix and runs the whole event chain
            // itself; the ranks stay in step because every decision is a deterministic function of identical state.
            // Only a BASE SCAN of the screened regime is shared out (tiles by index mod world) and followed by ONE
            // exchange: candidate records of the exact rescans + the pairs each rank emitted for the new window.
            Dev solo = dev;
            solo.world = 1; solo.rank = 0; solo.gather = 0; solo.wx = 0;  // (solo.strict stays on)
            if (!sched || !be.use_screen(dev, m_bound))  // a window event, or the end game's small plain scans: no exchange
                return be.launch_event(solo, m_bound, sched) == FNN_OK ? FNN_OK : fail(FNN_EHIP, "launch failed (" + be.err() + ")");
            if (be.launch_wx_scan(dev, m_bound) != FNN_OK) return fail(FNN_EHIP, "launch failed (" + be.err() + ")");
            const size_t bb = (size_t)wx_block_bytes(world);
            if (comm_mode == 1) {
                if (be.allgather_wx_on_stream(dev, bb) != FNN_OK) return fail(FNN_ERCCL, "all-gather failed (" + be.err() + ")");
            } else {
                std::vector<uint8_t> mine(bb), all(bb * (size_t)world);
                if (be.sync() != FNN_OK || be.d2h(mine.data(), dev.wsend, bb) != FNN_OK)
                    return fail(FNN_EHIP, "exchange block download failed (" + be.err() + ")");
                if (!host_fn || host_fn(host_ctx, mine.data(), all.data(), (int32_t)bb) != 0)
                    return fail(FNN_ERCCL, "host all-gather callback failed");
                if (be.h2d(dev.wrecv, all.data(), all.size()) != FNN_OK)
                    return fail(FNN_EHIP, "exchange block upload failed (" + be.err() + ")");
            }
            if (be.launch_wx_rest(dev, m_bound) != FNN_OK) return fail(FNN_EHIP, "launch failed (" + be.err() + ")");
            return FNN_OK;
        }
        // Several ranks without windows (no screening copy: small problems, or lookahead off): every event scans
        // 1/world of the tiles on each rank and exchanges the candidate records
        int32_t nper = 1;
        if (be.launch_event_scan(dev, m_bound, &nper) != FNN_OK) return fail(FNN_EHIP, "launch failed (" + be.err() + ")");
        if (comm_mode == 1) {
            if (be.allgather_on_stream(dev, nper) != FNN_OK) return fail(FNN_ERCCL, "all-gather failed (" + be.err() + ")");
        } else {
            std::vector<Cand> mine((size_t)nper), all((size_t)nper * (size_t)world);
            if (be.sync() != FNN_OK || be.d2h(mine.data(), dev.gsend, sizeof(Cand) * (size_t)nper) != FNN_OK)
                return fail(FNN_EHIP, "candidate download failed (" + be.err() + ")");
            if (!host_fn || host_fn(host_ctx, mine.data(), all.data(), (int32_t)(sizeof(Cand) * (size_t)nper)) != 0)
                return fail(FNN_ERCCL, "host all-gather callback failed");
            if (be.h2d(dev.grecv, all.data(), sizeof(Cand) * all.size()) != FNN_OK)
                return fail(FNN_EHIP, "candidate upload failed (" + be.err() + ")");
        }
        if (be.launch_event_rest(dev, m_bound, nper * world) != FNN_OK) return fail(FNN_EHIP, "launch failed (" + be.err() + ")");
        return FNN_OK;
    }

    // one event with a host round trip (tests / diagnostics)
    int32_t step(fnn_event* ev) {
        if (!begun) return fail(FNN_ESTATE, "fnn_step: call fnn_begin first");
        if (ended) return 0;
        int32_t rc;
        {
            rc = enqueue_event();
            if (rc != FNN_OK) return rc;
            if (be.sync() != FNN_OK) return fail(FNN_EHIP, "fnn_step: sync failed (" + be.err() + ")");
            rc = pull_state();
            if (rc != FNN_OK) return rc;
            if (hst.stall) {  // the event had no scan kernels and its window could not serve it: once more, with a scan
                rc = enqueue_event(1);
                if (rc != FNN_OK) return rc;
                if (be.sync() != FNN_OK) return fail(FNN_EHIP, "fnn_step: sync failed (" + be.err() + ")");
                rc = pull_state();
                if (rc != FNN_OK) return rc;
            }
        }
        m_bound = hst.m;
        resync_schedule();
        if (std::getenv("FNN_DEBUG")) {
            std::fprintf(stderr, "[fnn] m=%d c=%d P=%d nn=%d done=%d active=%d finish=%d need_rx=%d sa=%d sap=%d sb=%d sbp=%d xs=%d ys=%d U=%d nops=%d m_old=%d P_old=%d err=%d\n",
                         hst.m, hst.c, hst.P, hst.num_nodes, hst.done, hst.ev_active, hst.ev_finish, hst.need_rx, hst.sa,
                         hst.sap, hst.sb, hst.sbp, hst.xs, hst.ys, hst.U, hst.nops, hst.m_old, hst.P_old, hst.error);
            for (int i = 0; i < hst.nops; i++)
                std::fprintf(stderr, "[fnn]   op%d kind=%d a=%d b=%d c=%d d=%d e=%d mcur=%d flag=%d\n", i, hst.ops[i].kind,
                             hst.ops[i].a, hst.ops[i].b, hst.ops[i].c, hst.ops[i].d, hst.ops[i].e, hst.ops[i].mcur, hst.ops[i].flag);
        }
        if (!hst.ev_active) { ended = true; return 0; }
        if (ev) std::memcpy(ev, &hst.cur, sizeof(fnn_event));
        if (hst.done) ended = true;
        return 1;
    }

    // the whole agglomNodes loop without per-event round trips
    int32_t agglomerate() {
        if (!begun) return fail(FNN_ESTATE, "agglomerate: call fnn_begin first");
        double t0 = now_s();
        if (const char* e = std::getenv("FNN_BATCH")) { int v = std::atoi(e); if (v >= 1 && v <= 4096) batch = v; }
        while (!ended) {
            // with lookahead windows on, k_update closes the events and the exact row sum of the new
            // cluster is computed inside the next event's k_track (flushed before the host looks)
            be.defer_chain = dev.la != 0 && !std::getenv("FNN_NO_DEFER");
            const bool graph = be.graph_batches && comm_mode == 0;
            if (graph && be.capture_begin() != FNN_OK) return fail(FNN_EHIP, "stream capture failed (" + be.err() + ")");
            for (int i = 0; i < batch; i++) {
                int32_t rce = enqueue_event();
                if (rce != FNN_OK) return rce;
            }
            if (be.defer_chain && be.launch_chain_flush(dev) != FNN_OK) return fail(FNN_EHIP, "launch failed (" + be.err() + ")");
            if (graph && be.capture_end_launch() != FNN_OK) return fail(FNN_EHIP, "graph launch failed (" + be.err() + ")");
            be.defer_chain = false;
            int32_t rc = enqueue_status_exchange();
            if (rc != FNN_OK) return rc;
            if (be.sync() != FNN_OK) return fail(FNN_EHIP, "fnn_run: sync failed (" + be.err() + ")");
            rc = pull_state_ranks();
            if (rc != FNN_OK) return rc;
            m_bound = hst.m;
            resync_schedule();
            be.note_rx_exact(hst.n_rx_exact);
            if (hst.done) ended = true;
        }
        stats.t_agglom_s = now_s() - t0;
        return FNN_OK;
    }

    // expandNodes (NetMakerOriginal.java:246-325) from the merge log
    int32_t finish(int32_t* order_out) {
        if (!order_out) return fail(FNN_EINVAL, "fnn_finish: order_out is NULL");
        if (!begun) return fail(FNN_ESTATE, "fnn_finish: call fnn_begin first");
        if (n <= 3) {  // :133-140
            for (int32_t i = 0; i <= n; i++) order_out[i] = i;
            return FNN_OK;
        }
        if (!ended) return fail(FNN_ESTATE, "fnn_finish: agglomeration has not ended");
        double t0 = now_s();
        int32_t rc = pull_state();
        if (rc != FNN_OK) return rc;
        agglog.resize((size_t)hst.n_agg3);
        if (hst.n_agg3 > 0 && be.d2h(agglog.data(), dev.agglog, sizeof(Agg3Rec) * (size_t)hst.n_agg3) != FNN_OK)
            return fail(FNN_EHIP, "fnn_finish: log download failed (" + be.err() + ")");
        if (hst.record_events) {
            events.resize((size_t)hst.n_events);
            if (hst.n_events > 0 && be.d2h(events.data(), dev.evlog, sizeof(Event) * (size_t)hst.n_events) != FNN_OK)
                return fail(FNN_EHIP, "fnn_finish: event download failed (" + be.err() + ")");
        }
        // netNodes[0..2]: ids of the nodes at reference positions 0, 1, 2
        std::vector<int32_t> pslot((size_t)n), sid((size_t)n);
        if (be.d2h(pslot.data(), dev.pslot, sizeof(int32_t) * (size_t)n) != FNN_OK ||
            be.d2h(sid.data(), dev.sid, sizeof(int32_t) * (size_t)n) != FNN_OK)
            return fail(FNN_EHIP, "fnn_finish: layout download failed (" + be.err() + ")");
        for (int k = 0; k < 3; k++) {
            int32_t s = pslot[(size_t)k];
            if (s < 0 || s >= n) return fail(FNN_ESTATE, "fnn_finish: corrupt layout");
            last3[k] = sid[(size_t)s];
        }
        rc = expand(order_out);
        stats.t_expand_s = now_s() - t0;
        stats.n_events = hst.n_events;
        stats.sum_entries = hst.sum_entries;
        stats.scan_bytes = hst.bytes_timed;
        stats.bytes_total = hst.bytes_streamed;
        stats.plain_bytes = hst.bytes_plain;
        stats.n_rx_certified = hst.n_rx_certified;
        stats.n_rx_exact = hst.n_rx_exact;
        stats.n_screen_events = hst.n_screen_events;
        stats.n_rescan_units = hst.n_rescan_units;
        stats.n_base_scans = hst.n_base_scans;
        stats.n_window_hits = hst.n_la_hits;
        stats.n_window_fails = hst.n_la_fail;
        stats.window_pairs = hst.la_pairs_sum;
        stats.n_handover_retries = hst.n_ev_persistent;
        stats.n_sweeps_exact = hst.n_su_exact + hst.n_sweep_waits;
        stats.n_stalled_events = hst.n_stalled;
        stats.n_relaxed_events = hst.n_rl_events;
        be.collect_timing(stats);
        return rc;
    }

    int32_t expand(int32_t* ordering) {
        // ids run up to num_nodes; index 0 unused
        size_t N = (size_t)hst.num_nodes + 1;
        std::vector<int32_t> next(N, 0), prev(N, 0), ch1(N, 0), ch2(N, 0), nbr(N, 0);
        for (const Agg3Rec& r : agglog) {
            int32_t u = r.u_id, v = r.u_id + 1;
            if (u <= 0 || (size_t)v >= N) return fail(FNN_ESTATE, "fnn_finish: corrupt merge log");
            ch1[(size_t)u] = r.x_id; ch2[(size_t)u] = r.y_id;  // :610-611
            ch1[(size_t)v] = r.y_id; ch2[(size_t)v] = r.z_id;  // :615-616
            nbr[(size_t)u] = v; nbr[(size_t)v] = u;            // :647-648
        }
        int32_t x = last3[0], y = last3[1], z = last3[2];
        next[(size_t)x] = y; next[(size_t)y] = z; next[(size_t)z] = x;
        prev[(size_t)x] = z; prev[(size_t)y] = x; prev[(size_t)z] = y;
        for (size_t k = agglog.size(); k-- > 0;) {  // while (!amalgs.empty()) pop
            int32_t u = agglog[k].u_id;
            int32_t v = nbr[(size_t)u];
            x = ch1[(size_t)u]; y = ch2[(size_t)u]; z = ch2[(size_t)v];
            if (v != next[(size_t)u]) {
                int32_t t = u; u = v; v = t;
                t = x; x = z; z = t;
            }
            prev[(size_t)x] = prev[(size_t)u];
            next[(size_t)prev[(size_t)x]] = x;
            next[(size_t)x] = y;
            prev[(size_t)y] = x;
            next[(size_t)y] = z;
            prev[(size_t)z] = y;
            next[(size_t)z] = next[(size_t)v];
            prev[(size_t)next[(size_t)z]] = z;
        }
        int64_t guard = 0;
        while (x != 1) {
            x = next[(size_t)x];
            if (++guard > (int64_t)N) return fail(FNN_ESTATE, "fnn_finish: ring does not contain taxon 1");
        }
        int32_t a = x, t = 0;
        ordering[0] = 0;
        do {
            if (t >= n) return fail(FNN_ESTATE, "fnn_finish: ring longer than ntax");
            ordering[++t] = a;
            a = next[(size_t)a];
        } while (a != x);
        if (t != n) return fail(FNN_ESTATE, "fnn_finish: ring shorter than ntax");
        return FNN_OK;
    }

    int32_t run(int32_t* order_out, fnn_stats* out) {
        double t0 = now_s();
        int32_t rc = begin();
        if (rc != FNN_OK) return rc;
        if (n > 3) {
            rc = agglomerate();
            if (rc != FNN_OK) return rc;
        }
        rc = finish(order_out);
        if (rc != FNN_OK) return rc;
        stats.t_total_s = now_s() - t0;
        if (out) *out = stats;
        return FNN_OK;
    }

    int32_t get_nodes(int32_t* id, int32_t* nbr_id, double* Sx) {
        if (!begun) return fail(FNN_ESTATE, "fnn_get_nodes: call fnn_begin first");
        int32_t rc = pull_state();
        if (rc != FNN_OK) return rc;
        if (n <= 3) return FNN_OK;
        std::vector<int32_t> pslot((size_t)n), sid((size_t)n);
        std::vector<double> sx((size_t)n);
        if (be.d2h(pslot.data(), dev.pslot, sizeof(int32_t) * (size_t)n) != FNN_OK ||
            be.d2h(sid.data(), dev.sid, sizeof(int32_t) * (size_t)n) != FNN_OK ||
            be.d2h(sx.data(), dev.Sx, sizeof(double) * (size_t)n) != FNN_OK)
            return fail(FNN_EHIP, "fnn_get_nodes: download failed (" + be.err() + ")");
        for (int32_t i = 0; i < hst.m; i++) {
            int32_t s = pslot[(size_t)i];
            if (s < 0) {  // only after the special finish (netNodes[3] == null)
                if (id) id[i] = 0;
                if (nbr_id) nbr_id[i] = 0;
                if (Sx) Sx[i] = 0.0;
                continue;
            }
            if (id) id[i] = sid[(size_t)s];
            if (nbr_id) nbr_id[i] = s < 2 * hst.P ? sid[(size_t)(s ^ 1)] : 0;
            if (Sx) Sx[i] = sx[(size_t)s];
        }
        return FNN_OK;
    }

    int32_t get_live_matrix(double* out) {
        if (!begun) return fail(FNN_ESTATE, "fnn_get_live_matrix: call fnn_begin first");
        int32_t rc = pull_state();
        if (rc != FNN_OK) return rc;
        int32_t m = hst.m;
        if (n <= 3 || m <= 0) return FNN_OK;
        std::vector<int32_t> pslot((size_t)n);
        if (be.d2h(pslot.data(), dev.pslot, sizeof(int32_t) * (size_t)n) != FNN_OK)
            return fail(FNN_EHIP, "fnn_get_live_matrix: download failed (" + be.err() + ")");
        std::vector<double> rows((size_t)m * (size_t)m);
        if (be.d2h_2d(rows.data(), m, dev.D, ld, m, m) != FNN_OK)
            return fail(FNN_EHIP, "fnn_get_live_matrix: download failed (" + be.err() + ")");
        for (int32_t i = 0; i < m; i++)
            for (int32_t j = 0; j < m; j++) {
                int32_t si = pslot[(size_t)i], sj = pslot[(size_t)j];
                out[(size_t)i * m + j] = (si < 0 || sj < 0) ? 0.0 : rows[(size_t)si * m + sj];
            }
        return FNN_OK;
    }

    int64_t get_events(fnn_event* out, int64_t maxn) {
        int64_t k = (int64_t)events.size();
        if (out && maxn > 0) std::memcpy(out, events.data(), sizeof(fnn_event) * (size_t)(k < maxn ? k : maxn));
        return k;
    }
};

}  // namespace fnn
#endif
