// fnn_emu.cpp -- TEST INFRASTRUCTURE ONLY.
//
// CPU emulation backend for fnn::Engine: runs the per-thread bodies of
// fastneighbornet_amd/csrc/fnn_core.h in loops (in forward, reverse or shuffled
// "thread" order, to expose any dependence on execution order that would be a
// race on the GPU).  It lets the `-m "not gpu"` tests check the slot layout
// bookkeeping and the host logic (event loop, expandNodes) against the oracle in
// a container without a GPU.  It is NOT part of the product and libfastnn_hip.so
// never falls back to it.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <string>
#include <type_traits>
#include <vector>

#include "../../fastneighbornet_amd/csrc/fnn_engine.h"
#include "../../fastneighbornet_amd/csrc/fnn_chain.h"

static void build_targets_lanes(fnn::State& st);

namespace {

int g_order_mode = 0;  // 0 forward, 1 reverse, 2 shuffled
int g_update_mode = 0; // interleaving of k_update's bulk threads and special phases
int g_force_rescan_all = 0;       // screening: pretend the candidate list overflowed
int g_cand_cap = fnn::SCR_CAP;    // screening: capacity of the candidate list
uint64_t g_shuffle_state = 0x12345678ULL;

std::vector<int32_t> thread_order(int32_t count) {
    std::vector<int32_t> v((size_t)count);
    std::iota(v.begin(), v.end(), 0);
    if (g_order_mode == 1) std::reverse(v.begin(), v.end());
    else if (g_order_mode == 2) {
        for (int32_t i = count - 1; i > 0; i--) {
            g_shuffle_state = g_shuffle_state * 6364136223846793005ULL + 1442695040888963407ULL;
            int32_t j = (int32_t)((g_shuffle_state >> 33) % (uint64_t)(i + 1));
            std::swap(v[(size_t)i], v[(size_t)j]);
        }
    }
    return v;
}

struct EmuBackend {
    static constexpr int64_t kRowPad = 32;
    static constexpr int64_t kColPad = 2048;
    int32_t screen_min_n() const { return 8; }   // tiny on purpose: the CPU tests exercise screening
    int32_t screen_min_m = 8;
    bool screen_off = false;                          // set by the engine for a matrix with negative entries when FNN_EMU_PLAIN_NEG is set (the product's behaviour)
    void note_rx_exact(int64_t) {}                    // (the helper workgroups of the exact ComputeRx sums exist on the GPU only)
    // the emulation keeps screening matrices with negative entries: it is the CPU suite's coverage of the mixed-sign brackets,
    // which the product no longer uses by default; FNN_EMU_PLAIN_NEG=1 gives it the product's semantics (plain fp64 scan at
    // every event, no windows; with several ranks every event's scan is sharded and exchanged)
    bool keep_generic_screen() const { return std::getenv("FNN_EMU_PLAIN_NEG") == nullptr; }
    void set_problem_size(int32_t) {}
    void set_relaxed(int32_t) {}
    bool defer_chain = false; // (so does the deferred chain sum)
    int32_t launch_chain_flush(const fnn::Dev&) { return FNN_OK; }
    std::string err() const { return "emu"; }
    int32_t open(int32_t) { return FNN_OK; }
    void close() {}
    void* alloc(size_t b) { return std::malloc(b ? b : 1); }
    void free(void* p) { std::free(p); }
    size_t max_records(int32_t) { return 1; }
    int32_t memset(void* p, int v, size_t b) { std::memset(p, v, b); return FNN_OK; }
    int32_t h2d(void* d, const void* s, size_t b) { std::memcpy(d, s, b); return FNN_OK; }
    int32_t d2h(void* d, const void* s, size_t b) { std::memcpy(d, s, b); return FNN_OK; }
    int32_t h2d_2d(double* d, int64_t ldd, const double* s, int64_t lds, int64_t w, int64_t h) {
        for (int64_t r = 0; r < h; r++) std::memcpy(d + r * ldd, s + r * lds, sizeof(double) * (size_t)w);
        return FNN_OK;
    }
    int32_t d2d_2d(double* d, int64_t ldd, const double* s, int64_t lds, int64_t w, int64_t h) {
        return h2d_2d(d, ldd, s, lds, w, h);
    }
    int32_t d2h_2d(double* d, int64_t ldd, const double* s, int64_t lds, int64_t w, int64_t h) {
        return h2d_2d(d, ldd, s, lds, w, h);
    }
    int32_t unpack_rows(const fnn::Dev& d, const double* src, int64_t p0, int64_t, int32_t row0, int32_t cnt) {
        for (int64_t r = row0; r < (int64_t)row0 + cnt; r++)
            for (int64_t c = r + 1; c < d.n; c++) d.D[r * d.ld + c] = src[fnn::packed_row_base(d.n, r) + c - p0];
        return FNN_OK;
    }
    int32_t launch_mirror(const fnn::Dev& d) {
        for (int64_t r = 0; r < d.n; r++) {
            d.D[r * d.ld + r] = 0.0;
            for (int64_t c = 0; c < r; c++) d.D[r * d.ld + c] = d.D[c * d.ld + r];
        }
        return FNN_OK;
    }
    int32_t sync() { return FNN_OK; }
    bool graph_batches = false;
    int32_t capture_begin() { return FNN_OK; }
    int32_t capture_end_launch() { return FNN_OK; }
    void collect_timing(fnn_stats&) {}

    int32_t launch_synth(const fnn::Dev& d, uint64_t seed, int32_t dist) {
        for (int64_t i = 0; i < d.n; i++) {
            d.D[i * d.ld + i] = 0.0;
            for (int64_t j = i + 1; j < d.n; j++) {
                double v = fnn::synth_entry(d.n, i, j, seed, dist);
                d.D[i * d.ld + j] = v;
                d.D[j * d.ld + i] = v;
            }
        }
        return FNN_OK;
    }

    int32_t launch_validate(const fnn::Dev& d, int32_t* bad) {
        *bad = 0;
        for (int64_t i = 0; i < d.n; i++)
            for (int64_t j = 0; j <= i; j++) {
                double a = d.D[i * d.ld + j], b = d.D[j * d.ld + i];
                uint64_t ua, ub;
                std::memcpy(&ua, &a, 8); std::memcpy(&ub, &b, 8);
                bool fin = ((ua >> 52) & 0x7FF) != 0x7FF;
                if (ua != ub || !fin || (i == j && ua != 0)) *bad = 1;
            }
        return FNN_OK;
    }

    int32_t launch_prep_screen(const fnn::Dev& d, int64_t nrows) {
        uint64_t mx = 0;
        for (int64_t r = 0; r < nrows; r++)
            for (int64_t c = 0; c < d.ld; c++) {
                double v = d.D[r * d.ld + c];
                if (d.H) d.H[r * d.ldh + c] = fnn::bf16_from_double(v);
                uint64_t b;
                std::memcpy(&b, &v, 8);
                b &= 0x7FFFFFFFFFFFFFFFULL;
                if (r < d.n && c < d.n && b > mx) mx = b;
            }
        d.st->dmax_bits = mx;
        for (int64_t r = 0; r < d.n; r++)
            for (int64_t c = 0; c < d.n; c++)
                if (d.D[r * d.ld + c] < 0.0) d.st->nonneg = 0;
        return FNN_OK;
    }

    int32_t launch_init(const fnn::Dev& d) {
        for (int32_t k : thread_order(d.n)) fnn::init_thread(d, k);
        return FNN_OK;
    }

    // exact scan of one 32 x 512 unit (k_resolve)
    void rescan_unit(const fnn::Dev& d, int32_t u, fnn::Cand& best) {
        fnn::State& st = *d.st;
        const int32_t m = st.m, twoP = 2 * st.P;
        const double cm2 = (double)st.c - 2.0;
        int32_t rt, ct;
        fnn::tri_tile_decode(u / 4, fnn::SCR_TW / fnn::SCR_TH, rt, ct);
        const int32_t rb = rt * fnn::SCR_TH, cb = ct * fnn::SCR_TW + (u % 4) * fnn::SCR_UW;
        for (int32_t r0 = rb; r0 < rb + fnn::SCR_TH && r0 < m; r0 += 2)
            for (int32_t c0 = cb; c0 < cb + fnn::SCR_UW && c0 <= r0; c0 += 2) {
                const double* R0 = d.D + (int64_t)r0 * d.ld;
                const double* R1 = d.D + (int64_t)(r0 + 1) * d.ld;
                bool r1 = r0 + 1 < m, c1 = c0 + 1 < m;
                fnn::scan_micro(r0, c0, m, twoP, cm2, R0[c0], R0[c0 + 1], R1[c0], R1[c0 + 1],
                                d.Sx[r0], r1 ? d.Sx[r0 + 1] : 0.0, d.spos[r0], r1 ? d.spos[r0 + 1] : 0,
                                d.Sx[c0], c1 ? d.Sx[c0 + 1] : 0.0, d.spos[c0], c1 ? d.spos[c0 + 1] : 0, best);
            }
    }
    // k_screen + k_resolve: bf16 screening with per-pair brackets, candidate units, exact rescan;
    // on a base scan of a lookahead window the pass also emits the pairs to track
    fnn::Cand scan_screened(const fnn::Dev& d) {
        fnn::State& st = *d.st;
        fnn::Cand best;
        best = fnn::cand_none();
        if (st.done) return best;
        const int32_t m = st.m, twoP = 2 * st.P;
        const float cm2 = (float)((double)st.c - 2.0), cm2k = fnn::screen_cm2k(st);
        const float k1 = fnn::screen_k1(st), k2 = fnn::screen_k2(st);
        const bool nn = st.nonneg != 0;
        const int32_t nunits = fnn::screen_unit_count(m);
        const float finf = (float)fnn::inf_f64();
        float* lbrec = d.srec;
        float* ubrec = d.srec + nunits;
        // (several ranks with windows: a rank's pairs go to its exchange block, capped at its share of the list)
        auto emit = [&](int32_t rs, int32_t cs, float lb) {
            if (!(st.la_emit && lb <= st.la_theta_pred)) return;
            if (!d.wx) { fnn::la_append(d, rs, cs, twoP); return; }
            const int32_t i = (*d.lacnt)++;
            if (i < fnn::wx_pair_cap(d.world)) fnn::la_record(d, reinterpret_cast<int32_t*>(d.wsend + fnn::wx_pairs_off()) + fnn::LA_REC_INTS * (int64_t)i, rs, cs, twoP);
        };
        for (int32_t u : thread_order(nunits)) {
            fnn::Brk b{finf, finf};
            if ((u / 4) % d.world == d.rank) {
                int32_t rt, ct;
                fnn::tri_tile_decode(u / 4, fnn::SCR_TW / fnn::SCR_TH, rt, ct);
                const int32_t rb = rt * fnn::SCR_TH, cb = ct * fnn::SCR_TW + (u % 4) * fnn::SCR_UW;
                for (int32_t r0 = rb; r0 < rb + fnn::SCR_TH && r0 < m; r0 += 2)
                    for (int32_t c0 = cb; c0 < cb + fnn::SCR_UW && c0 <= r0; c0 += 2) {
                        const uint16_t* R0 = d.H + (int64_t)r0 * d.ldh;
                        const uint16_t* R1 = d.H + (int64_t)(r0 + 1) * d.ldh;
                        bool r1 = r0 + 1 < m, c1 = c0 + 1 < m;
                        const float e00 = fnn::bf16_to_float(R0[c0]), e01 = fnn::bf16_to_float(R0[c0 + 1]);
                        const float e10 = fnn::bf16_to_float(R1[c0]), e11 = fnn::bf16_to_float(R1[c0 + 1]);
                        const float s0 = (float)d.Sx[r0], s1 = r1 ? (float)d.Sx[r0 + 1] : 0.f;
                        const float q0 = (float)d.Sx[c0], q1 = c1 ? (float)d.Sx[c0 + 1] : 0.f;
                        if (nn) fnn::screen_micro_nn(r0, c0, m, twoP, k1, k2, e00, e01, e10, e11, s0, s1, q0, q1, b, emit);
                        else fnn::screen_micro(r0, c0, m, twoP, cm2, cm2k, e00, e01, e10, e11, s0, s1, q0, q1, b);
                    }
            }
            lbrec[u] = b.lb;
            ubrec[u] = b.ub;
        }
    // k_resolve: smallest upper bound, then the units whose lower bound does not exceed it
        float ubg = finf;
        for (int32_t u = 0; u < nunits; u++) ubg = fnn::fminf_(ubg, ubrec[u]);  // (units of other ranks' tiles hold +inf)
        const float thr = ubg + 2.0f * fnn::screen_delta(st);
        st.ncand = 0;
        st.rescan_all = (!st.screen_ok || !(thr == thr) || g_force_rescan_all) ? 1 : 0;
        if (!st.rescan_all)
            for (int32_t u : thread_order(nunits))
                if (lbrec[u] <= thr) {
                    if (st.ncand >= g_cand_cap) { st.rescan_all = 1; break; }
                    d.clist[st.ncand++] = u;
                }
        int64_t rescanned = 0;
        if (st.rescan_all) {
            for (int32_t u : thread_order(nunits))
                if ((u / 4) % d.world == d.rank) { rescan_unit(d, u, best); rescanned++; }
        } else {
            for (int32_t i : thread_order(st.ncand)) { rescan_unit(d, d.clist[i], best); rescanned++; }
        }
        if (d.wx) {  // the base scan is closed after the exchange (wx_merge)
            int32_t* h = reinterpret_cast<int32_t*>(d.wsend);
            h[0] = st.la_emit ? *d.lacnt : 0; h[1] = st.rescan_all; h[2] = (int32_t)rescanned; h[3] = st.n_events;
            fnn::Cand* recs = reinterpret_cast<fnn::Cand*>(d.wsend + fnn::wx_recs_off());
            for (int j = 0; j < fnn::GATHER_RECS; j++) recs[j] = fnn::cand_none();
            recs[(d.rank * 7 + 3) % fnn::GATHER_RECS] = best;  // (any of the records may carry it)
            return best;
        }
        st.n_screen_events++;
        st.ev_screened = 1;
        st.n_rescan_units += rescanned;
        fnn::la_close_base(st, d.lalog, d.lacnt);
        return best;
    }

    // k_scan over this rank's share of the micro-tiles
    fnn::Cand scan_local(const fnn::Dev& d, bool sched = true) {
        fnn::State& st = *d.st;
        st.ev_timed = sched ? 1 : 0;
        // k_track: serve the event from the open lookahead window if it can certify the minimum
        if (!sched && fnn::la_active(st)) {
            fnn::Cand tb;
            tb = fnn::cand_none();
            const fnn::TrackArgs ta = fnn::track_args(st);
            for (int32_t it : thread_order((int32_t)fnn::track_item_count(ta))) fnn::track_item(d, it, ta, tb);
            fnn::la_track_done(d, tb, ta);
            if (st.la_hit) return d.recs[0];
        } else {
            if (st.la_valid) st.la_prev_end = 0;
            fnn::la_prepare_base(st, d.lacnt);
        }
        if (d.H && !screen_off && st.m >= screen_min_m) return scan_screened(d);
        fnn::Cand best;
        best = fnn::cand_none();
        st.rl_active = 0;
        if (!st.done && st.m > 3 && !(st.m == 4 && st.c == 2) && st.rl_on && st.m > st.rl_min) {  // k_relaxed
            fnn::RlSerialEnv env;
            return fnn::relaxed_find(d, env);
        }
        if (!st.done) {
            int32_t m = st.m, twoP = 2 * st.P;
            double cm2 = (double)st.c - 2.0;
            std::vector<std::pair<int32_t, int32_t>> tiles;
            int64_t idx = 0;
            for (int32_t r0 = 0; r0 < m; r0 += 2)
                for (int32_t c0 = 0; c0 <= r0; c0 += 2, idx++)
                    if (idx % d.world == d.rank) tiles.emplace_back(r0, c0);
            for (int32_t t : thread_order((int32_t)tiles.size())) {
                int32_t r0 = tiles[(size_t)t].first, c0 = tiles[(size_t)t].second;
                const double* R0 = d.D + (int64_t)r0 * d.ld;
                const double* R1 = d.D + (int64_t)(r0 + 1) * d.ld;
                bool r1 = r0 + 1 < m, c1 = c0 + 1 < m;
                fnn::scan_micro(r0, c0, m, twoP, cm2, R0[c0], R0[c0 + 1], R1[c0], R1[c0 + 1],
                                d.Sx[r0], r1 ? d.Sx[r0 + 1] : 0.0, d.spos[r0], r1 ? d.spos[r0 + 1] : 0,
                                d.Sx[c0], c1 ? d.Sx[c0 + 1] : 0.0, d.spos[c0], c1 ? d.spos[c0 + 1] : 0, best);
            }
        }
        return best;
    }
    // Same launch-sequence semantics as the GPU backend: an unscheduled event of the screened regime
    // carries no scan kernels; if its window cannot serve it, it stalls (this and the following such
    // sequences do nothing) until the host launches an event with a scan.
    int32_t launch_event(const fnn::Dev& d, int32_t m_bound, bool sched) {
        fnn::State& st = *d.st;
        const bool screen = d.H && !screen_off && m_bound >= screen_min_m;
        const bool has_scan = sched || !screen;
        if (has_scan) {
            if (!st.done) st.stall = 0;
            return event_rest(d, m_bound, scan_local(d, sched));
        }
        if (st.done) return FNN_OK;
        if (st.stall) { st.n_stalled++; return FNN_OK; }
        if (fnn::la_active(st)) {
            fnn::Cand tb;
            tb = fnn::cand_none();
            const fnn::TrackArgs ta = fnn::track_args(st);
            for (int32_t it : thread_order((int32_t)fnn::track_item_count(ta))) fnn::track_item(d, it, ta, tb);
            fnn::la_track_done(d, tb, ta);
            if (st.la_hit) return event_rest(d, m_bound, d.recs[0]);
        } else {
            if (st.la_valid) st.la_prev_end = 0;
            fnn::la_prepare_base(st, d.lacnt);
        }
        st.stall = 1;
        st.n_stalled++;
        return FNN_OK;
    }
    int32_t launch_event_scan(const fnn::Dev& d, int32_t, int32_t* nper) {
        // contribute 3 records (the real one plus two "none") to exercise the multi-record exchange
        fnn::Cand none;
        none = fnn::cand_none();
        d.gsend[0] = none; d.gsend[1] = scan_local(d); d.gsend[2] = none;
        *nper = 3;
        return FNN_OK;
    }
    int32_t allgather_on_stream(const fnn::Dev&, int32_t) { return FNN_ERCCL; }  // no RCCL in the emulation
    int32_t allgather_wx_on_stream(const fnn::Dev&, size_t) { return FNN_ERCCL; }
    int32_t allgather_bytes_on_stream(const void*, void*, size_t) { return FNN_ERCCL; }
    bool use_screen(const fnn::Dev& d, int32_t m_bound) const { return d.H != nullptr && !screen_off && m_bound >= screen_min_m; }
    // several ranks with lookahead windows: the sharded part of a base scan ... (exchange) ... merge + the rest
    int32_t launch_wx_scan(const fnn::Dev& d, int32_t) {
        fnn::State& st = *d.st;
        if (!st.done) st.stall = 0;
        st.ev_timed = 1;
        if (st.la_valid) st.la_prev_end = 0;
        fnn::la_prepare_base(st, d.lacnt);
        (void)scan_screened(d);
        return FNN_OK;
    }
    int32_t launch_wx_rest(const fnn::Dev& d, int32_t m_bound) {
        fnn::wx_merge(d);
        fnn::Cand best = fnn::cand_none();
        for (int32_t r = 0; r < d.world * fnn::GATHER_RECS; r++)
            if (fnn::cand_better(d.grecv[r], best)) best = d.grecv[r];
        return event_rest(d, m_bound, best);
    }
    int32_t launch_event_rest(const fnn::Dev& d, int32_t m_bound, int32_t ntotal) {
        fnn::Cand best;
        best = fnn::cand_none();
        for (int32_t r = 0; r < ntotal; r++)
            if (fnn::cand_better(d.grecv[r], best)) best = d.grecv[r];
        return event_rest(d, m_bound, best);
    }

    // weighted row sum of the node in slot z, recomputed from the matrix (check of Dev::T, small problems only)
    static double t_recompute(const fnn::Dev& d, int32_t z) {
        const fnn::State& st = *d.st;
        long double acc = 0.0L;
        for (int32_t p = 0; p < st.m; p++)
            if (p != z) acc += (long double)(p < 2 * st.P ? 0.5 : 1.0) * (long double)d.D[(int64_t)z * d.ld + p];
        return (double)acc;
    }

    int32_t event_rest(const fnn::Dev& d, int32_t m_bound, fnn::Cand best) {
        fnn::State& st = *d.st;
        // ---- the decide step (tail of k_track for window events, k_decide otherwise)
        fnn::t_finalize(d);
        if (d.n <= 300 && !st.done && !st.error) {  // Dev::T against the matrix, within the certification's own bound
            const double bound = fnn::rx_bound(st);
            for (int32_t z = 0; z < st.m; z++) {
                const double want = t_recompute(d, z);
                if (!(std::fabs(d.T[z] - want) <= 0.25 * bound)) { st.error = 16; break; }
            }
        }
        int32_t a = 0, b = 0, ida = 0, idb = 0;
        if (fnn::pick_needs_candidate(st)) {
            a = d.pslot[(int32_t)(best.key >> 32)]; b = d.pslot[(int32_t)(best.key & 0xFFFFFFFFu)];
            ida = d.sid[a]; idb = d.sid[b];
        }
        fnn::pick(d, best, a, b, ida, idb);
        if (!st.ev_active) return FNN_OK;
        if (!st.ev_finish) {
            fnn::Quad qd;
            fnn::quad_load(d, qd);
            double rx[4] = {0.0, 0.0, 0.0, 0.0};
            if (st.need_rx) {
                fnn::rx_from_T(st, qd, rx);
                if (fnn::rx_certify(st, qd, rx)) st.n_rx_certified++;
                else {  // the exact sequential sums (ComputeRx, NetMakerOriginal.java:549-561)
                    const int32_t z[4] = {st.sa, st.sap, st.sb, st.sbp};
                    for (int32_t t : thread_order(st.m_old)) fnn::rx_fill_thread(d, t, st.m_old, 2 * st.P_old, z);
                    for (int k = 0; k < 4; k++) rx[k] = z[k] >= 0 ? fnn::chain_sum(d.chain + (int64_t)(k + 1) * d.cstride, st.m_old) : 0.0;
                    st.n_rx_exact++;
                }
            }
            // the plan on the slot tables in memory, or (as the GPU does) on a preloaded copy of the
            // entries an event can touch; a miss of that copy is a finding
            if (g_order_mode == 0) { fnn::GlobalTab T{d}; fnn::decide_plan(d, T, qd, rx); }
            else {
                int32_t key[fnn::TAB_NK], vsid[fnn::TAB_NK], vspos[fnn::TAB_NK], pkey[fnn::TAB_NP], vpslot[fnn::TAB_NP], misses = 0;
                fnn::CachedTab T{d, key, vsid, vspos, pkey, vpslot, &misses};
                fnn::tab_preload(T, st.sa, st.sb, st.P, st.m);
                fnn::decide_plan(d, T, qd, rx);
                if (misses && !st.error) st.error = 15;
            }
        }
        {
            // the GPU replays the micro-ops lane-parallel (fnn_hip.hip: build_targets_wave); the same
            // algorithm on arrays of 64 "lanes" must give what the generic build_targets gives
            fnn::State lanes = st;
            build_targets_lanes(lanes);
            fnn::build_targets(d);
            bool same = lanes.nS == st.nS && lanes.ntgt == st.ntgt && lanes.tU == st.tU && lanes.tV == st.tV;
            for (int i = 0; same && i < st.nS; i++) same = lanes.S[i] == st.S[i];
            for (int i = 0; same && i < st.ntgt; i++)
                same = std::memcmp(&lanes.tgt[i], &st.tgt[i], sizeof(fnn::Tgt)) == 0;
            if (!same && !st.error) st.error = 12;
        }
        // k_update: bulk columns and the special phases run concurrently on the GPU; emulate
        // different interleavings (g_update_mode) to expose any conflict between them.
        // The GPU runs the special phases on a copy of the S x S block of the involved slots (LDS):
        // the same phases on such a copy (BlockAcc) must leave exactly what the phases on the matrix leave.
        {
            const int32_t nph = fnn::update_special_phases(st);
            double blk[fnn::MAX_S * fnn::MAX_S] = {0}, sxl[fnn::MAX_S] = {0}, tl[fnn::MAX_S] = {0};
            int32_t berr = 0;
            double tu = 0.0, tv = 0.0, tub = 0.0, tvb = 0.0, tuv[2];
            for (int32_t e : thread_order(fnn::MAX_S * fnn::MAX_S)) fnn::special_block_load(d, blk, sxl, tl, e);
            {
                int32_t Sl[fnn::MAX_S];
                for (int i = 0; i < fnn::MAX_S; i++) Sl[i] = st.S[i];
                const fnn::BlockAcc A{blk, sxl, tl, Sl, st.nS, &berr};
                for (int32_t ph = 0; ph < nph; ph++)
                    for (int32_t i : thread_order(fnn::MAX_S)) { fnn::update_special_acc(A, d, ph, i, tuv); tub += tuv[0]; tvb += tuv[1]; }
            }
            const fnn::PlanView pv = fnn::plan_view(st, fnn::UniId{});
            auto bulk1 = [&](int32_t k) { fnn::update_bulk_mem(d, pv, k, tuv); tu += tuv[0]; tv += tuv[1]; };
            auto bulk = [&]() { for (int32_t k : thread_order(m_bound)) bulk1(k); };
            auto special = [&](int32_t ph) { for (int32_t i : thread_order(fnn::MAX_S)) { fnn::update_special(d, ph, i, tuv); tu += tuv[0]; tv += tuv[1]; } };
            if (g_update_mode == 0) { bulk(); for (int32_t ph = 0; ph < nph; ph++) special(ph); }
            else if (g_update_mode == 1) { for (int32_t ph = 0; ph < nph; ph++) special(ph); bulk(); }
            else {  // bulk split in two halves around the middle phase
                std::vector<int32_t> ord = thread_order(m_bound);
                size_t half = ord.size() / 2;
                for (int32_t ph = 0; ph < nph; ph++) {
                    if (ph == nph / 2) for (size_t q = 0; q < half; q++) bulk1(ord[q]);
                    special(ph);
                }
                for (size_t q = half; q < ord.size(); q++) bulk1(ord[q]);
            }
            // the partial sums of T of the new cluster's two nodes (one "workgroup" here); summed by the next decide step
            d.upart[2] = tu; d.upart[3] = tv;
            st.tp_n = st.ev_finish ? 0 : 1;
            st.tp_U = st.U;
            (void)tub; (void)tvb;
            bool same = berr == 0 || st.ev_finish;
            if (st.ev_finish) same = true;
            for (int i = 0; same && !st.ev_finish && i < st.nS; i++) {
                same = std::memcmp(&sxl[i], &d.Sx[st.S[i]], 8) == 0 && std::memcmp(&tl[i], &d.T[st.S[i]], 8) == 0;
                for (int j = 0; same && j < st.nS; j++)
                    same = std::memcmp(&blk[i * fnn::MAX_S + j], &d.D[(int64_t)st.S[i] * d.ld + st.S[j]], 8) == 0;
            }
            if (!same && !st.error) {
                st.error = berr ? 13 : 14;
                if (std::getenv("FNN_DEBUG")) {
                    std::fprintf(stderr, "[emu] block mismatch: nS=%d kind=%d nops=%d finish=%d S=", st.nS, st.cur.kind, st.nops, st.ev_finish);
                    for (int i = 0; i < st.nS; i++) std::fprintf(stderr, "%d ", st.S[i]);
                    std::fprintf(stderr, "\n");
                    for (int i = 0; i < st.nS; i++) {
                        std::fprintf(stderr, "  sx %g vs %g |", sxl[i], d.Sx[st.S[i]]);
                        for (int j = 0; j < st.nS; j++) std::fprintf(stderr, " %g/%g", blk[i * fnn::MAX_S + j], d.D[(int64_t)st.S[i] * d.ld + st.S[j]]);
                        std::fprintf(stderr, "\n");
                    }
                }
            }
        }
        double usx = 0.0;
        if (!st.ev_finish) usx = fnn::chain_sum(d.chain, st.m);
        fnn::finalize(d, usx);
        return FNN_OK;
    }
};

using EmuEngine = fnn::Engine<EmuBackend>;

}  // namespace

// fnn_hip.hip: build_targets_wave on arrays of 64 lanes (ballot = loop over the lanes, readlane = index)
static void build_targets_lanes(fnn::State& st) {
    using namespace fnn;
    const int nops = st.nops, U = st.U, m = st.m, ev_finish = st.ev_finish;
    int Sl[64], sk[64], sa[64], sb[64], sc[64], sd[64];
    for (int l = 0; l < 64; l++) Sl[l] = -1;
    int nS = 0, err = 0;
    auto ballot = [&](auto pred) { unsigned long long mk = 0; for (int l = 0; l < 64; l++) if (pred(l)) mk |= 1ULL << l; return mk; };
    auto add_slot = [&](int sl) {
        if (ballot([&](int l) { return l < nS && Sl[l] == sl; })) return;
        if (nS >= MAX_S) { err = 5; return; }
        Sl[nS] = sl;
        nS++;
    };
    add_slot(U);
    add_slot(U + 1);
    for (int i = 0; i < nops; i++) {
        const Op& o = st.ops[i];
        add_slot(o.a); add_slot(o.b);
        if (o.kind == OP_AGG3) { add_slot(o.c); add_slot(o.d); add_slot(o.e); }
    }
    for (int l = 0; l < 64; l++) { sk[l] = T_COPY; sa[l] = Sl[l]; sb[l] = sc[l] = sd[l] = -1; }
    struct Sym { int kind, a, b, c, d; };
    auto idx = [&](int sl) { const unsigned long long hit = ballot([&](int l) { return l < nS && Sl[l] == sl; }); return hit ? __builtin_ctzll(hit) : 0; };
    auto get = [&](int i) { return Sym{sk[i], sa[i], sb[i], sc[i], sd[i]}; };
    auto put = [&](int i, const Sym& t) { sk[i] = t.kind; sa[i] = t.a; sb[i] = t.b; sc[i] = t.c; sd[i] = t.d; };
    auto comb = [&](const Sym& A, const Sym& B) {
        Sym r{T_COPY, -1, -1, -1, -1};
        if (A.kind == T_COPY && B.kind == T_COPY) { r.kind = T_L1; r.a = A.a; r.b = B.a; }
        else if (A.kind == T_L1 && B.kind == T_L1 && A.b == B.b) { r.kind = T_L2U; r.a = A.a; r.b = A.b; r.c = B.a; }
        else if (A.kind == T_COPY && B.kind == T_L1) { r.kind = T_L2V; r.d = A.a; r.c = B.a; r.b = B.b; }
        else err = 6;
        return r;
    };
    for (int i = 0; i < nops; i++) {
        const Op& o = st.ops[i];
        if (o.kind == OP_SWAP) { const int ia = idx(o.a), ib = idx(o.b); const Sym ta = get(ia), tb = get(ib); put(ia, tb); put(ib, ta); }
        else if (o.kind == OP_MOVE) { const Sym t = get(idx(o.a)); put(idx(o.b), t); }
        else if (o.kind == OP_AGG3) {
            const Sym sx = get(idx(o.a)), sy = get(idx(o.b)), sz = get(idx(o.c));
            const Sym nu = comb(sx, sy), nv = comb(sz, sy);
            put(idx(o.d), nu);
            put(idx(o.e), nv);
        }
    }
    auto keep = [&](int l) {
        const bool isUV = (Sl[l] == U || Sl[l] == U + 1);
        return l < nS && !(Sl[l] >= m && !ev_finish) && !(!isUV && sk[l] == T_COPY && sa[l] == Sl[l]);
    };
    const unsigned long long km = ballot(keep);
    const int ntgt = __builtin_popcountll(km);
    st.tU = -1; st.tV = -1;
    for (int l = 0; l < 64; l++) {
        if (!keep(l)) continue;
        const int pos = __builtin_popcountll(km & ((1ULL << l) - 1ULL));
        if (pos < MAX_TGT) { Tgt t; t.dst = Sl[l]; t.kind = sk[l]; t.a = sa[l]; t.b = sb[l]; t.c = sc[l]; t.d = sd[l]; st.tgt[pos] = t; }
        if (Sl[l] == U) st.tU = pos;
        if (Sl[l] == U + 1) st.tV = pos;
    }
    for (int l = 0; l < nS; l++) st.S[l] = Sl[l];
    st.nS = nS;
    st.ntgt = ntgt < MAX_TGT ? ntgt : MAX_TGT;
    if (ntgt > MAX_TGT) st.error = 7;
    else if (err) st.error = err;
}

// CPU model of the block-parallel exact chain sum (fnn_hip.hip: block_chain_sum): same
// primitives (fnn_chain.h), same unit structure (thread chunks of `ept` addends, waves of
// 64 threads, runs of equal predicted binade, walker with the two fallback levels).  The
// predicted prefix is computed in long double so that it differs from the true sequential
// partial sums the way the GPU's tree-ordered prefix does.
static double chain_model(const double* buf, int m, int ept, int guard_bits, fnn::ChainStats* cs) {
    using namespace fnn;
    const int T = 1024;
    double s = 0.0;
    ChainStats st{0, 0, 0, 0};
    for (int base = 0; base < m; base += T * ept) {
        std::vector<int> pure(T), E(T);
        std::vector<Mono> own(T);
        long double pre = 0.0L;
        for (int t = 0; t < T; t++) {
            // (same per-thread logic as block_chain_sum in fnn_hip.hip)
            double A0 = (double)((long double)s + pre);
            long double pre_end = pre;
            for (int i = 0; i < ept; i++) {
                int idx = base + t * ept + i;
                pre_end += (long double)(idx < m ? buf[idx] : 0.0);
            }
            double A1 = (double)((long double)s + pre_end);
            pre = pre_end;
            int32_t Et = -1;
            bool ok = chain_predict(A0, A1, guard_bits != 0, Et);
            Mono mt = mono_identity();
            if (ok) {
                const double invu = inv_ulp(Et);
                auto chunk = [&](auto tag) {  // the GPU's chunk: addends handled independently, increments summed as a tree
                    constexpr int N = decltype(tag)::value;
                    double a[N];
                    for (int i = 0; i < N; i++) a[i] = (base + t * ept + i < m) ? buf[base + t * ept + i] : 0.0;
                    return chain_chunk<N>(a, invu, mt);
                };
                if (ept == 32) ok = chunk(std::integral_constant<int, 32>());
                else if (ept == 16) ok = chunk(std::integral_constant<int, 16>());
                else if (ept == 8) ok = chunk(std::integral_constant<int, 8>());
                else
                    for (int i = 0; i < ept; i++) {
                        int idx = base + t * ept + i;
                        double a = idx < m ? buf[idx] : 0.0;
                        if (!chain_accumulate(a, invu, mt)) { ok = false; break; }
                    }
            }
            pure[t] = ok;
            E[t] = Et;
            own[t] = mt;
        }
        auto serial = [&](int t) {
            for (int i = 0; i < ept; i++) {
                int idx = base + t * ept + i;
                if (idx < m) s += buf[idx];
            }
        };
        for (int w = 0; w < T / 64; w++) {
            int t = w * 64;
            while (t < (w + 1) * 64) {
                if (!pure[t]) { serial(t); st.mixed++; t++; continue; }
                int e = t;
                Mono run = own[t];
                while (e + 1 < (w + 1) * 64 && pure[e + 1] && E[e + 1] == E[t]) { e++; run = mono_compose(run, own[e]); }
                // (the GPU walker applies the automata on the bit pattern; both forms must agree)
                double s2 = s;
                const bool okb = mono_apply_bits(s2, E[t], mono_inc_bits(run.i0), mono_inc_bits(run.i1));
                const bool okf = mono_apply(s, E[t], run);
                if (okb != okf || (okf && f2u(s2) != f2u(s))) st.thread_fail += 1000000;  // would show in the tests
                if (okf) st.runs++;
                else {
                    st.run_fail++;
                    for (int j = t; j <= e; j++) {
                        double sj = s;
                        const bool ob = mono_apply_bits(sj, E[j], mono_inc_bits(own[j].i0), mono_inc_bits(own[j].i1));
                        const bool of = mono_apply(s, E[j], own[j]);
                        if (ob != of || (of && f2u(sj) != f2u(s))) st.thread_fail += 1000000;
                        if (!of) { st.thread_fail++; serial(j); }
                    }
                }
                t = e + 1;
            }
        }
    }
    if (cs) *cs = st;
    return s;
}

// CPU model of the record form of the block-parallel sum (fnn_chain.h "records"; fnn_hip.hip: block_chain_sum2): the
// same per-thread records (chain_thread_record) from a prefix that differs from the true partial sums the way the
// GPU's tree-ordered prefix does, the same evaluation of the records, the same fallback to the first form.
static double chain_model2(const double* buf, int m, int guard_bits, fnn::ChainStats* cs) {
    using namespace fnn;
    const int T = 1024, EPT = 32;
    double s = 0.0;
    ChainStats st{0, 0, 0, 0};
    for (int base = 0; base < m; base += T * EPT) {
        std::vector<ChRec> rec(T);
        long double pre = 0.0L;
        for (int t = 0; t < T; t++) {
            double a[EPT];
            int cnt = 0;
            double loc = 0.0;
            for (int i = 0; i < EPT; i++) {
                const int idx = base + t * EPT + i;
                a[i] = idx < m ? buf[idx] : 0.0;
                if (idx < m) cnt++;
                loc += a[i];
            }
            const double A0 = (double)((long double)s + pre);
            pre += (long double)loc;
            rec[t] = chain_thread_record<EPT>(a, cnt, A0, loc, guard_bits != 0);
            if (rec[t].kind == CHR_SERIAL) st.mixed++;
        }
        int32_t applied = 0;
        double s2 = s;
        const bool ok = chain_walk_records(rec.data(), T, s2, &applied, [&](int t, double& acc) {
            for (int i = 0; i < EPT; i++) { const int idx = base + t * EPT + i; if (idx < m) acc += buf[idx]; }
        });
        if (!ok) {  // a verification failed: the whole sum through the first form
            ChainStats c1{0, 0, 0, 0};
            const double r = chain_model(buf, m, EPT, guard_bits, &c1);
            st.runs += c1.runs; st.mixed += c1.mixed; st.run_fail += 1 + c1.run_fail; st.thread_fail += c1.thread_fail;
            if (cs) *cs = st;
            return r;
        }
        st.runs += applied;
        s = s2;
    }
    if (cs) *cs = st;
    return s;
}

extern "C" {

double emu_chain_model(const double* buf, int32_t m, int32_t ept, int32_t guard_bits, int32_t* stats4) {
    fnn::ChainStats cs;
    double r = chain_model(buf, m, ept, guard_bits, &cs);
    if (stats4) { stats4[0] = cs.runs; stats4[1] = cs.mixed; stats4[2] = cs.run_fail; stats4[3] = cs.thread_fail; }
    return r;
}
double emu_chain_model2(const double* buf, int32_t m, int32_t guard_bits, int32_t* stats4) {
    fnn::ChainStats cs;
    double r = chain_model2(buf, m, guard_bits, &cs);
    if (stats4) { stats4[0] = cs.runs; stats4[1] = cs.mixed; stats4[2] = cs.run_fail; stats4[3] = cs.thread_fail; }
    return r;
}
double emu_chain_serial(const double* buf, int32_t m) {
    double s = 0.0;
    for (int32_t i = 0; i < m; i++) s += buf[i];
    return s;
}

void emu_set_order_mode(int32_t mode) { g_order_mode = mode % 3; g_update_mode = (mode / 3) % 3; }
void emu_set_screen_debug(int32_t force_rescan_all, int32_t cand_cap) {
    g_force_rescan_all = force_rescan_all;
    g_cand_cap = cand_cap > 0 ? cand_cap : fnn::SCR_CAP;
}
const char* emu_last_error(void) { return fnn::g_last_error.c_str(); }

int32_t emu_create(int32_t n, const fnn_opts* opts, void** out) {
    auto* e = new EmuEngine();
    int32_t rc = e->create(n, opts);
    if (rc != FNN_OK) { e->destroy(); delete e; return rc; }
    *out = e;
    return FNN_OK;
}
int32_t emu_destroy(void* h) { auto* e = (EmuEngine*)h; e->destroy(); delete e; return FNN_OK; }
int32_t emu_set_rows(void* h, int32_t row0, int32_t nrows, const double* rows, int64_t ld) {
    return ((EmuEngine*)h)->set_rows(row0, nrows, rows, ld);
}
int32_t emu_set_packed_upper(void* h, const double* packed) { return ((EmuEngine*)h)->set_packed_upper(packed); }
int32_t emu_synth(void* h, uint64_t seed, int32_t dist) { return ((EmuEngine*)h)->synth(seed, dist); }
int32_t emu_run(void* h, int32_t* order, fnn_stats* st) { return ((EmuEngine*)h)->run(order, st); }
int32_t emu_begin(void* h) { return ((EmuEngine*)h)->begin(); }
int32_t emu_step(void* h, fnn_event* ev) { return ((EmuEngine*)h)->step(ev); }
int32_t emu_finish(void* h, int32_t* order) { return ((EmuEngine*)h)->finish(order); }
int64_t emu_get_events(void* h, fnn_event* out, int64_t maxn) { return ((EmuEngine*)h)->get_events(out, maxn); }
int32_t emu_get_counts(void* h, int32_t* m, int32_t* c, int32_t* nn) {
    auto* e = (EmuEngine*)h;
    int32_t rc = e->pull_state();
    if (rc != FNN_OK) return rc;
    if (m) *m = e->hst.m;
    if (c) *c = e->hst.c;
    if (nn) *nn = e->hst.num_nodes;
    return FNN_OK;
}
int32_t emu_get_nodes(void* h, int32_t* id, int32_t* nbr, double* sx) { return ((EmuEngine*)h)->get_nodes(id, nbr, sx); }
int32_t emu_get_live_matrix(void* h, double* out) { return ((EmuEngine*)h)->get_live_matrix(out); }
int32_t emu_get_matrix(void* h, double* out, int64_t ld_out) { return ((EmuEngine*)h)->get_matrix(out, ld_out); }
int32_t emu_comm_init_host(void* h, int32_t world, int32_t rank, fnn_allgather_fn fn, void* ctx) {
    auto* e = (EmuEngine*)h;
    int32_t rc = e->comm_set(2, world, rank);
    if (rc != FNN_OK) return rc;
    e->host_fn = fn;
    e->host_ctx = ctx;
    return FNN_OK;
}

}  // extern "C"
