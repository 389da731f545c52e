// fnn_core.h -- per-thread bodies of the Canonical Neighbor-Net engine.
//
// Everything here is a __host__ __device__ function over plain pointers: the HIP
// kernels in fnn_hip.hip are thin wrappers that map (block, thread) to the
// arguments of these bodies, and tests/emu/fnn_emu.cpp drives the SAME bodies in
// loops on the CPU so that the slot bookkeeping can be checked against the oracle
// without a GPU.  The emulation driver is test infrastructure; the product path
// is the HIP build only.
//
// ---------------------------------------------------------------------------
// Layout ("slots").  The reference keeps nodes in a packed array netNodes[0..m)
// (NetMakerOriginal.java:141-150) and addresses the matrix through node.distID.
// We keep the n x n fp64 matrix in SLOT order instead: live nodes occupy slots
// [0, m); slots [0, 2P) hold the P two-node clusters, the cluster's smaller id
// (its representative, NeighborNetCanonical.java:153) in the even slot and its
// partner (NetNode.nbr) in the odd slot; slots [2P, m) hold singletons.  A
// cluster pair therefore owns a dense 2x2 / 1x2 / 1x1 block of the matrix and the
// scan streams the lower triangle of the leading m x m block with no indirection.
// The reference's own position of a node (NetNode.positionID) decides tie-breaks
// and the order of the three sequential sums, so it is carried per slot (spos)
// together with its inverse (pslot).
//
// All fp64 expressions are written in the reference's evaluation order and the
// translation units are compiled with -ffp-contract=off.
// ---------------------------------------------------------------------------
#ifndef FNN_CORE_H
#define FNN_CORE_H

#include <stdint.h>

#if defined(__HIPCC__)
#define FNN_HD __host__ __device__ __forceinline__
#else
#define FNN_HD inline
#endif
// post-increment of a counter shared by the threads of a launch (a plain ++ in the CPU emulation)
// FNN_COUNTER_READ: the value of such a counter as seen by a workgroup that has observed the arrival of every
// workgroup that adds to it: a read-modify-write of zero at agent scope, i.e. performed where the other workgroups'
// atomic adds were performed (a plain or even an L1-bypassing load may be served from a stale line of this XCD's L2).
#if defined(__HIP_DEVICE_COMPILE__)
#define FNN_ATOMIC_INC(p) atomicAdd((p), 1)
#define FNN_COUNTER_READ(p) __hip_atomic_fetch_add((p), 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#else
#define FNN_ATOMIC_INC(p) ((*(p))++)
#define FNN_COUNTER_READ(p) (*(p))
#endif

namespace fnn {

constexpr int OP_NONE = 0;
constexpr int OP_SWAP = 1;  // exchange slots a <-> b
constexpr int OP_MOVE = 2;  // move slot a into dead slot b
constexpr int OP_AGG3 = 3;  // agg3way: rows X=a, Y=b, Z=c -> new rows U=d, V=e
constexpr int MAX_OPS = 6;

constexpr int KIND_2WAY = 2, KIND_3WAY = 3, KIND_4WAY = 4, KIND_FINISH = 5;
constexpr int GATHER_RECS = 64;  // multi-GPU: candidate records a rank may contribute per event

// The net effect of an event's micro-ops on one matrix row, as a recipe over the rows of the
// matrix BEFORE the event ("o[r]" = old row r), used by the fused update kernel for the
// columns that are not themselves involved in the event:
//   T_COPY(a)      o[a]                                   (slot swap / move)
//   T_L1(a,b)      (2/3)*o[a] + o[b]/3                    (agg3way, NetMakerOriginal.java:655-656)
//   T_L2U(a,b,c)   (2/3)*L1(a,b) + L1(c,b)/3              (agg4way: u of the second agg3way)
//   T_L2V(d,c,b)   (2/3)*o[d]    + L1(c,b)/3              (agg4way: v of the second agg3way)
// every intermediate is rounded to fp64 exactly as when the reference stores it in D.
constexpr int T_COPY = 0, T_L1 = 1, T_L2U = 2, T_L2V = 3;
constexpr int MAX_TGT = 8;
constexpr int MAX_S = 8;
struct Tgt {
    int32_t dst, kind, a, b, c, d;
};

struct Op {
    int32_t kind, a, b, c, d, e;
    int32_t mcur;  // slots [0, mcur) are swept
    int32_t flag;  // AGG3: 1 if pos(u) < pos(v) (aliasing rule, see agg3_special)
};

// scan candidate: q = +inf means "none"; key = (i << 32) | j, i > j reference positions; si, sj: the slots of the
// nodes at positions i, j (so that the decide step need not look them up: one dependent round trip less)
struct Cand {
    double q;
    uint64_t key;
    int32_t si, sj;
};
FNN_HD Cand cand_none() {
    Cand c;
    c.q = __builtin_bit_cast(double, (uint64_t)0x7FF0000000000000ULL);
    c.key = ~0ULL;
    c.si = c.sj = -1;
    return c;
}

struct Event {  // == fnn_event / nno_event
    int32_t m_before, c_before, cx_id, cy_id, x_id, y_id, kind, u_id;
    double best;
    int64_t entries;
};

struct Agg3Rec {  // one agg3way call: u = (x,y), v = (y,z); v.id = u_id + 1
    int32_t u_id, x_id, y_id, z_id;
};

struct State {
    int32_t n, m, c, P, num_nodes, done, n_agg3, record_events;
    int64_t n_events, sum_entries;
    int32_t error;  // non-zero if an "unreachable" branch was taken
    // ---- fp32 screening (see screen_delta) ----
    int32_t screen_ok;     // the bound is valid for this matrix (finite, |D| < 1e37)
    int32_t rescan_all;    // candidate list overflowed: rescan every unit
    int32_t ncand;         // units in clist
    int32_t nonneg;        // no negative entry in the input matrix (then none ever appears)
    uint64_t dmax_bits;    // bit pattern of max |D| over the input matrix
    int64_t n_rescan_units, n_screen_events;  // statistics
    int64_t bytes_streamed;  // matrix bytes the scans had to stream: 2 (bf16 pass) or 8 per entry + rescans
    int32_t ev_screened, pad_scr2;
    // ---- lookahead window (see "Lookahead" below) ----
    int32_t la_on;          // configured: windows enabled (single rank, screening copy present)
    int32_t la_K;           // events one window may serve (upper limit)
    int32_t la_Kcur;        // ... the open / opening window: min(la_K, la_kbase + m / la_kdiv) at its base scan
    int32_t la_kbase, la_kdiv;
    int32_t la_target;      // wanted number of tracked pairs per window
    int32_t la_min_m;       // windows only while m >= la_min_m (the screened regime)
    int32_t la_valid;       // a window is open
    int32_t la_k;           // events completed since the window's base scan (the base event included)
    int32_t la_np;          // tracked pairs
    int32_t la_nf;          // clusters created since the base scan ("fresh")
    int32_t la_nf_done;     // ... of which the first la_nf_done have had their rows swept (pairs inserted)
    int32_t la_prev_end;    // how the previous window ended: 0 schedule, 1 could not certify, 2 list overflow
    int32_t la_hit;         // this event's minimum came from the window (k_track)
    int32_t la_emit;        // this event's screening pass emits the pairs of a new window
    int32_t la_count_unused; // (the append counter of the tracked list lives outside the control block: Dev.lacnt)
    int32_t la_pcap;        // capacity of the tracked-pair list (<= LA_PCAP)
    int32_t la_skip, la_backoff;  // after an overflow: base scans that do not try to open a window
    int32_t la_base_stamp;  // n_events at the base scan: clusters stamped later are fresh
    int32_t la_have_mprev;
    float la_theta_pred;    // emission threshold of the open / opening window (fp32)
    double la_theta_eff;    // acceptance threshold: theta_pred minus the error slack
    double la_W;            // window width above the previous event's minimum
    double la_coef;         // c - 2 - K of the base scan: the coefficient of the window's lower bounds
    double la_mprev;        // previous event's scan minimum
    int64_t n_base_scans, n_la_hits, n_la_fail, n_la_overflow, la_pairs_sum, la_items_sum;
    int32_t ev_timed, la_k_prev;  // the host brackets this event's scan launch with HIP events
    int64_t bytes_timed;      // the part of bytes_streamed that belongs to timed screening launches (k_screen + rescans)
    int64_t bytes_plain;      // ... and to the plain fp64 scans (k_scan; always timed)
    int64_t n_ev_persistent;  // k_track's fan-in: events whose records had to be read a second time (check word mismatch; expected 0)
    int64_t n_su_exact;       // ... of which the sweep had to wait for the exact row sum of the new cluster
    // deferred row sum of the newest cluster: k_update closes the event, the exact sequential sum is
    // computed by a workgroup of the NEXT event's k_track (or by k_chain_flush before the host looks)
    int32_t chain_pending, chain_m, chain_U, upart_n;
    // approximate weighted row sums (see "The 4-candidate choice without an O(m) pass"): the two nodes of the newest
    // cluster (slots tp_U, tp_U + 1) still await the sum of the tp_n per-workgroup partials the update left (0: none)
    int32_t tp_n, tp_U;
    // an event launched WITHOUT scan kernels found that its window cannot serve it: this and the
    // following such events do nothing until the host (which resyncs every batch) launches one with a scan
    int32_t stall;
    int32_t chain_buf;        // which chain buffer (0 or CHAIN_ALT) holds the addends of the pending row sum: the fused event kernel fills
                              // the other one while the chain workgroup of the same launch still reads this one
    // reference positions that this event's plan changed for slots that need NOT be involved slots: the node at the last position
    // takes y's place (agg3_plan; NetMakerOriginal.java:641-643).  The fused event kernel hands them to its column threads, which run
    // in the same launch as the plan and cannot rely on seeing its plain stores to d.spos
    int32_t pov_n, pov_slot[2], pov_pos[2], pad_pov;
    int64_t n_stalled;        // events skipped that way
    int64_t n_sweep_waits;    // k_track: sweeps that had to wait for the exact row sum
    int64_t ev_ticks[8];      // (unused; the phase-split diagnostics live in Dev::ticks)
    // ---- Relaxed mode (NeighborNetLocal.java; see "Relaxed mode" below) ----
    int32_t rl_on, rl_min;      // findNodes is the relaxed search while m > rl_min (1024, NetMakerOriginal.java:361)
    int32_t rl_top, rl_first;   // NeighborNetLocal.top / firstTime (:17-18)
    int32_t rl_active, pad_rl;  // this event's pair came from the relaxed search (one record, d.recs[0])
    uint64_t rl_rng;            // java.util.Random state (48 bits)
    int64_t rl_evals;           // Q values evaluated by this event's search (recorded as the event's `entries`)
    int64_t n_rl_rows, n_rl_events;  // statistics: row minima computed / events served by the relaxed search
    // ---- current event ----
    int32_t ev_active, ev_finish, need_rx;
    int32_t sa, sap, sb, sbp;  // slots of Cx, Cx.nbr, Cy, Cy.nbr (-1: none)
    int32_t m_old, P_old, c_old;
    int32_t xs, ys;            // slots (old layout) of the chosen x, y
    int32_t U;                 // slot (new layout) of the node u returned by the merge
    int32_t nops;
    Op ops[MAX_OPS];
    Event cur;
    double rx[4];  // exact ComputeRx results for Cx, Cx.nbr, Cy, Cy.nbr (only when not certified)
    int32_t pad_rx;
    int32_t force_exact_rx;     // diagnostic: never certify (tests the exact path)
    int64_t n_rx_certified, n_rx_exact;  // statistics
    // fused update (k_update): slots involved in the event, and the recipes of the rows that change
    int32_t nS, S[MAX_S];
    int32_t ntgt, tU, tV;  // tU / tV: index in tgt[] of the rows of u and u.nbr
    Tgt tgt[MAX_TGT];
};

// Fused event kernel (k_track with fuse): what the deciding workgroup tells the column workgroups of the SAME launch - the
// part of the control block that plan_view reads (same field names) and the values those threads cannot read from memory
// because this very launch produces them.  Travels as PLAN_WORDS words of (int | launch tag << 32): a reader that finds the
// launch's tag in every word has the whole message, in ONE round trip.
struct PlanMsg {
    int32_t kind;             // 0: no update in this launch (the window could not serve the event, the run has ended ...), 1: update
    int32_t last_wg;          // the deciding workgroup: its block of columns goes to the spare workgroup
    int32_t pU, cU;           // previous event's cluster: T[pU], T[pU+1] are tfin; Sx[cU], Sx[cU+1] come from the chain workgroup
    int32_t chain_wait;       // ... of this launch (wait for its flag == evtag)
    int32_t evtag;
    int32_t chain_dst;        // chain buffer that takes this event's row-sum addends
    int32_t nbulk;            // blocks of 1024 columns
    int32_t pov_n, pov_slot[2], pov_pos[2], pad0;
    int32_t tfin[4];          // two doubles
    int32_t m_old, P_old, ev_finish, xs, ys, pad1;
    int32_t nS, S[MAX_S];
    int32_t ntgt, tU, tV;
    Tgt tgt[MAX_TGT];
    int32_t fill[44];
};
static_assert(sizeof(PlanMsg) == 4 * 128, "PlanMsg is PLAN_WORDS ints");

struct Dev {
    double* D;       // slot-ordered matrix, row stride ld
    int64_t ld;
    int32_t n;
    double* Sx;      // per slot: NetNode.Sx
    double* T;       // per slot: approximate weighted row sum of the node (certified 4-candidate choice only)
    int32_t* sid;    // per slot: NetNode.id
    int32_t* spos;   // per slot: NetNode.positionID
    int32_t* pslot;  // reference position -> slot (-1 if empty)
    double* chain;   // 6 buffers of cstride doubles, addressed through chain_addr(position): [0] / [CHAIN_ALT] the addends of the new cluster's
                     // row sum (State.chain_buf says which; consumed by the NEXT event's k_track while that event may already fill [1..4], the
                     // <= 4 ComputeRx sums of its own decision, and - fused event kernel - the other row-sum buffer)
    uint64_t* plan;  // fused event kernel (k_track with fuse): the plan message of the deciding workgroup to the column workgroups of the
                     // same launch, PLAN_WORDS words of (payload | launch tag << 32)
    int64_t cstride; // n rounded up to a whole super-chunk
    Cand* recs;      // per-block scan records
    uint64_t* rchk;  // k_track's fan-in: a check word per record (2 x 1024), see rec_publish; [2048 ..): results of the ComputeRx helper workgroups
    uint16_t* H;     // bf16 screening copy of D: H[r * ldh + c] == bf16(D[r][c]) (its own row stride, see fnn_create)
    int64_t ldh;
    float* srec;     // screening: per unit [tile][4] lower bounds, then [tile][4] upper bounds
    float* stile;    // screening: per tile lower bound, then per tile upper bound
    uint64_t* shit;  // screening: per unit, the lanes (8 columns each) that hold a pair under the window's threshold
    int32_t* clist;  // screening: units that may hold the true minimum
    int32_t* islot;  // lookahead: node id -> slot (-1: dead); 3n + 8 entries
    int32_t* cstamp; // lookahead: node id -> n_events + 1 when its current cluster was formed (0: initial)
    int32_t* tpairs; // lookahead: tracked pairs, LA_PCAP records of LA_REC_INTS ints (la_record: ids, slot hints, the pair's 2 x 2 entries)
    int32_t* fresh;  // lookahead: per fresh cluster {representative id, stamp, slot} (LA_KMAX entries)
    uint32_t* ticket; // lookahead: arrival counter of k_track's workgroups
    int64_t* ticks;  // diagnostics (FNN_TICKS=1), 100 MHz ticks summed over events: [0..7] k_track's last workgroup, [8..15] k_update
    int32_t* lacnt;  // lookahead: append counter of the tracked list (its own word: the control block may be
                     // cached in LDS by a workgroup while every thread appends)
    double* upart;   // per workgroup of k_update, 4 doubles: {sum, sum of magnitudes} of the new cluster's exact row-sum
                     // addends (for the sweep beside the exact chain), and the partial sums of T of its two nodes
    double* lalog;   // lookahead diagnostics: per base scan {event, m, W, pairs, events served by the previous window} (LA_LOGCAP records)
    Cand* gsend;     // multi-GPU: this rank's candidate record(s) of the event (<= GATHER_RECS)
    uint8_t* wsend;  // multi-GPU with lookahead windows: this rank's exchange block of a base scan (wx_block_bytes)
    uint8_t* wrecv;  // ... and all ranks' blocks after the all-gather
    Cand* grecv;     // multi-GPU: all ranks' candidate records
    int32_t la;      // lookahead windows enabled for this run (k_track is part of the launch sequence)
    int32_t rank, world;  // scan sharding: rank scans the tiles with index = rank (mod world)
    int32_t gather;       // non-zero: candidate records are exchanged between ranks (go to gsend)
    int32_t wx;           // non-zero: several ranks WITH lookahead windows - only the base scans are sharded and exchanged
    int32_t strict;       // non-zero (several ranks): a window event that gives up for a reason other ranks cannot see - a record of
                          // k_track's fan-in that could not be read back - is an ERROR: the ranks must take identical decisions
    int32_t fault_event;  // test hook (FNN_FAULT_GIVEUP=rank:event, -1 = none): this rank's window gives that event up
    int32_t plan_ticks;   // diagnostic (FNN_TICKS=1): decide_plan stamps its stages into ticks[20 .. 23]
    // Relaxed mode: NeighborNetLocal.rowPermutation (positions), and the per-call HashMap foundRowMinimums as per-slot
    // entries {stamp == n_events + 1 of the call, value, number of tied rows, their slots in position order}
    int32_t* rperm;
    int32_t* rl_stamp;
    int32_t* rl_cnt;
    int32_t* rl_list;  // RL_TIES slots per slot
    double* rl_val;
    uint64_t* rl_mail; // several workgroups in k_relaxed: command word, arrival counter, per-workgroup records (RL_MAIL_WORDS)
    State* st;
    Event* evlog;    // n records (if record_events)
    Agg3Rec* agglog; // n records
};

// Chain buffers (addends of the sequential sums, indexed by reference position e) are
// stored chunk-interleaved: the chain kernel gives thread t of a 1024-thread workgroup the
// CH_EPT consecutive addends [t*CH_EPT, (t+1)*CH_EPT) of a 32768-addend super-chunk; with the
// j-th PAIR of every chunk stored contiguously over t, each of the thread's 16-byte loads is
// perfectly coalesced across the wave.  The producers scatter by position anyway.
constexpr int CHAIN_ALT = 5;             // the second row-sum buffer of Dev.chain (see State.chain_buf)
constexpr int CHAIN_BUFS = 6;
constexpr int PLAN_WORDS = 128;
constexpr int TICK_WG = 256;                      // diagnostics per workgroup of k_update (FNN_TICKS=1): sums of its start and end stamps
constexpr int TICK_WORDS = 32 + 3 * TICK_WG;      // Dev::ticks
constexpr int CH_T = 1024;               // threads of the chain workgroup
constexpr int CH_EPT = 32;               // addends per thread
constexpr int CH_SC = CH_T * CH_EPT;     // addends per super-chunk
constexpr int CH_NSLOT = 48;             // parked ("mixed") chunks per super-chunk
FNN_HD int64_t chain_addr(int32_t e) {
    const int32_t sc = e / CH_SC, r = e % CH_SC;
    const int32_t t = r / CH_EPT, j = r % CH_EPT;
    return (int64_t)sc * CH_SC + (int64_t)(j >> 1) * (2 * CH_T) + 2 * t + (j & 1);
}

FNN_HD double inf_f64() {
    union { uint64_t u; double d; } v;
    v.u = 0x7FF0000000000000ULL;
    return v.d;
}

// x / 3.0, correctly rounded - bit for bit the IEEE quotient the reference's `D[y][p] / 3.0` produces - in three
// operations instead of the ~35-instruction division expansion (a column thread of the update divides up to 48 times
// and runs alone on its SIMD: nothing hides that latency).  Markstein's correction step: with z = RN(1/3) the product
// q = RN(x z) is within one ulp of x / 3, the residual r = x - 3 q is exact in an FMA, and RN(q + r z) is the correctly
// rounded quotient (3's significand is not all ones; a quotient is never a rounding midpoint).  Outside a generous
// exponent range (where r could underflow or 3 q overflow) the plain division is used.  Checked against `/ 3.0` on
// 5e9 random and structured operands (tests/tools/div3_check.c) and, through the oracle, by every parity test.
FNN_HD double div3(double x) {
    const double ax = __builtin_fabs(x);
    if (!(ax >= 0x1p-900 && ax <= 0x1p900)) return x / 3.0;  // zero, tiny, huge, inf, NaN
    const double z = 0x1.5555555555555p-2;
    const double q = x * z;
    const double r = __builtin_fma(-3.0, q, x);
    return __builtin_fma(r, z, q);
}

FNN_HD bool cand_better(const Cand& a, const Cand& b) {
    // total order (Q, i, j): the reference keeps the FIRST strict minimum of its
    // (i asc, j asc) scan (NeighborNetCanonical.java:173), i.e. the argmin on this order
    return a.q < b.q || (a.q == b.q && a.key < b.key);
}

FNN_HD void consider(double q, int32_t pa, int32_t pb, int32_t sa, int32_t sb, Cand& best) {
    uint32_t i = (uint32_t)(pa > pb ? pa : pb), j = (uint32_t)(pa > pb ? pb : pa);
    Cand c;
    c.q = q;
    c.key = ((uint64_t)i << 32) | (uint64_t)j;
    c.si = pa > pb ? sa : sb;
    c.sj = pa > pb ? sb : sa;
    if (cand_better(c, best)) best = c;
}

// Qpq = ((double)num_clusters - 2.0) * Dpq - p.Sx - q.Sx with p the node at the
// LARGER reference position (NeighborNetCanonical.java:152,156,170)
FNN_HD double qval(double cm2, double dpq, double sxa, int32_t pa, double sxb, int32_t pb) {
    double sp = pa > pb ? sxa : sxb;
    double sq = pa > pb ? sxb : sxa;
    return (cm2 * dpq - sp) - sq;
}

// One 2x2 micro-tile of the scan: rows r0, r0+1 (r0 even), columns c0, c0+1
// (c0 even, c0 <= r0).  e[r][c] are the four matrix entries.  Sx / pos of the two
// rows and two columns are passed in.  Follows NeighborNetCanonical.java:151-178.
// Every cluster pair of the tile goes to sink(q, dpq, sx row node, sx column node, pos row node,
// pos column node, row slot, column slot); scan_micro's sink keeps the best (Q, i, j).
template <class Sink>
FNN_HD void scan_micro_t(int32_t r0, int32_t c0, int32_t m, int32_t twoP, double cm2,
                         double e00, double e01, double e10, double e11,
                         double sxr0, double sxr1, int32_t pr0, int32_t pr1,
                         double sxc0, double sxc1, int32_t pc0, int32_t pc1, Sink&& sink) {
    if (r0 >= m || c0 >= m) return;
    if (r0 < twoP) {
        // rows are one two-node cluster; c0 <= r0 < 2P so the columns are one too
        if (r0 == c0) return;  // same cluster (q.nbr == p, :160)
        // p = representative at the larger reference position; 4-term mean in the
        // order D[p][q] + D[p][q.nbr] + D[p.nbr][q] + D[p.nbr][q.nbr]  (:169)
        double dpq;
        if (pr0 > pc0) dpq = (((e00 + e01) + e10) + e11) / 4.0;
        else dpq = (((e00 + e10) + e01) + e11) / 4.0;
        sink(qval(cm2, dpq, sxr0, pr0, sxc0, pc0), dpq, sxr0, sxc0, pr0, pc0, r0, c0);
    } else if (c0 < twoP) {
        // rows are singletons, columns one two-node cluster: 2-term mean (:165/:167,
        // the two forms add the same two entries in the same order)
        double d0 = (e00 + e01) / 2.0;
        sink(qval(cm2, d0, sxr0, pr0, sxc0, pc0), d0, sxr0, sxc0, pr0, pc0, r0, c0);
        if (r0 + 1 < m) {
            double d1 = (e10 + e11) / 2.0;
            sink(qval(cm2, d1, sxr1, pr1, sxc0, pc0), d1, sxr1, sxc0, pr1, pc0, r0 + 1, c0);
        }
    } else {
        // singletons x singletons: Dpq = D[p][q] (:163); only row > column
        bool c1ok = (c0 + 1 < m), r1ok = (r0 + 1 < m);
        if (r0 > c0) {
            sink(qval(cm2, e00, sxr0, pr0, sxc0, pc0), e00, sxr0, sxc0, pr0, pc0, r0, c0);
            if (c1ok) sink(qval(cm2, e01, sxr0, pr0, sxc1, pc1), e01, sxr0, sxc1, pr0, pc1, r0, c0 + 1);
            if (r1ok) {
                sink(qval(cm2, e10, sxr1, pr1, sxc0, pc0), e10, sxr1, sxc0, pr1, pc0, r0 + 1, c0);
                if (c1ok) sink(qval(cm2, e11, sxr1, pr1, sxc1, pc1), e11, sxr1, sxc1, pr1, pc1, r0 + 1, c0 + 1);
            }
        } else {  // diagonal micro-tile: only (r0+1, c0)
            if (r1ok) sink(qval(cm2, e10, sxr1, pr1, sxc0, pc0), e10, sxr1, sxc0, pr1, pc0, r0 + 1, c0);
        }
    }
}
struct BestSink {
    Cand& best;
    FNN_HD void operator()(double q, double, double, double, int32_t pa, int32_t pb, int32_t rs, int32_t cs) const { consider(q, pa, pb, rs, cs, best); }
};
FNN_HD void scan_micro(int32_t r0, int32_t c0, int32_t m, int32_t twoP, double cm2,
                       double e00, double e01, double e10, double e11,
                       double sxr0, double sxr1, int32_t pr0, int32_t pr1,
                       double sxc0, double sxc1, int32_t pc0, int32_t pc1, Cand& best) {
    scan_micro_t(r0, c0, m, twoP, cm2, e00, e01, e10, e11, sxr0, sxr1, pr0, pr1, sxc0, sxc1, pc0, pc1, BestSink{best});
}

// ---------------------------------------------------------------------------
// Screening of the scan with a 2-byte copy (the fp64 master decides, a narrow copy
// may screen).  H holds every matrix entry rounded to bf16 (8 significant bits); the screening
// pass streams H (2 B per entry instead of 8) and brackets every cluster pair's Q:
//     Qh = fma(cm2, mean(h), -Sp~) - Sq~          (fp32; Sx rounded to float)
//     e  = cm2k * mean(|h|)                       (cm2k >= (c-2) * kappa)
//     LB = Qh - e - dlt <= Q <= Qh + e + dlt = UB
// kappa bounds the RELATIVE entry error |d - h| <= kappa |h| of double -> float -> bf16 (round to
// nearest twice: (2^-8 + 1.01 * 2^-24) / (1 - 2^-8 - 1.01 * 2^-24) < 0.003945), so pairs with small
// distances - the ones that can win - get tight brackets.  dlt (screen_delta) is the global slack
// for everything done in fp32: with u = 2^-24, Dmax = max |D| of the INPUT matrix (later entries
// are convex combinations of earlier ones), |Sx| <= n Dmax, M = (c-2) Dmax + 2 n Dmax:
//   <=3 adds of the mean, amplified by (c-2)    3 u (c-2) Dmax
//   the product / fma                            u (c-2) Dmax
//   rounding Sp, Sq to float                     2 u n Dmax
//   the subtractions                             2 u M
//   the same again for e, plus subnormal slack   (generously) the same
//   total <= u Dmax (7 (c-2) + 6 n), doubled by screen_delta.
// Per unit the pass records min LB and min UB.  Let UBg be the smallest UB (of this rank).  The
// pair with the true minimum has LB <= Q* <= UBg, and every pair with LB > UBg has a true Q
// STRICTLY above the true minimum, so rescanning (fp64, exact tie-break) only the units whose
// min LB <= UBg finds the same pair as the full scan.
// ---------------------------------------------------------------------------
constexpr int SCR_TH = 32;    // rows per screening tile
constexpr int SCR_TW = 2048;  // columns per screening tile (256 threads x 8)
constexpr int SCR_UW = 512;   // columns per unit (one wave of a tile)
constexpr int SCR_CAP = 16384;  // capacity of the candidate-unit list (emulation)
constexpr float SCR_KAPPA = 0.003945f;

FNN_HD uint16_t bf16_from_double(double v) {
    const float f = (float)v;
    uint32_t u = __builtin_bit_cast(uint32_t, f);
    if ((u & 0x7F800000u) == 0x7F800000u) return (uint16_t)((u >> 16) | ((u & 0xFFFFu) ? 0x40u : 0u));  // inf / NaN
    u += 0x7FFFu + ((u >> 16) & 1u);  // round to nearest even
    return (uint16_t)(u >> 16);
}
FNN_HD float bf16_to_float(uint16_t h) { return __builtin_bit_cast(float, (uint32_t)h << 16); }

FNN_HD float screen_delta(const State& st) {
    const double dmax = __builtin_bit_cast(double, st.dmax_bits);
    const double u = 5.9604644775390625e-08;  // 2^-24
    const double dlt = 2.0 * u * dmax * (7.0 * ((double)st.c - 2.0) + 6.0 * (double)st.n) + 1e-30;
    return (float)(dlt * 1.0000002);  // round up when narrowing to float
}
// (c - 2) * kappa, rounded up
FNN_HD float screen_cm2k(const State& st) { return (float)(((double)st.c - 2.0) * (double)SCR_KAPPA * 1.000001); }

// For matrices without negative entries mean|h| == mean h, so LB / UB are affine in the mean:
// LB = k1 * mean - Sp - Sq with k1 <= (c-2)(1 - kappa) and UB = k2 * mean - Sp - Sq with
// k2 >= (c-2)(1 + kappa); the coefficients are rounded in the safe direction.  With lookahead
// windows on, the lower bound uses (c - 2 - K) instead of (c - 2): it then also bounds the pair's
// Q in each of the next K events (see "Lookahead" below); it is merely a little looser now.
FNN_HD float screen_k1(const State& st) {
    double cc = (double)st.c - 2.0 - (st.la_on ? (double)st.la_Kcur : 0.0);
    if (cc < 0.0) cc = 0.0;
    return (float)((cc * (1.0 - (double)SCR_KAPPA * 1.000001)) * (1.0 - 2e-7));
}
FNN_HD float screen_k2(const State& st) { return (float)((((double)st.c - 2.0) * (1.0 + (double)SCR_KAPPA * 1.000001)) * (1.0 + 2e-7)); }

FNN_HD float fminf_(float a, float b) { return __builtin_fminf(a, b); }  // v_min_f32 (no NaN arises when screen_ok)
FNN_HD float fabsf_(float a) { return __builtin_fabsf(a); }             // a source modifier on the GPU

struct Brk { float lb, ub; };  // running minima of the lower / upper bounds

FNN_HD void brk_take(float dmean, float amean, float cm2, float cm2k, float sp, float sq, Brk& b) {
    const float q = __builtin_fmaf(cm2, dmean, -sp) - sq;
    const float e = cm2k * amean;
    b.lb = fminf_(b.lb, q - e);
    b.ub = fminf_(b.ub, q + e);
}

// one 2x2 block of the screening pass: same case analysis as scan_micro, on bf16-decoded floats
FNN_HD void screen_micro(int32_t r0, int32_t c0, int32_t m, int32_t twoP, float cm2, float cm2k,
                         float e00, float e01, float e10, float e11,
                         float sxr0, float sxr1, float sxc0, float sxc1, Brk& b) {
    if (r0 >= m || c0 >= m || c0 > r0) return;
    if (r0 < twoP) {
        if (r0 == c0) return;
        brk_take((((e00 + e01) + e10) + e11) * 0.25f,
                 (((fabsf_(e00) + fabsf_(e01)) + fabsf_(e10)) + fabsf_(e11)) * 0.25f, cm2, cm2k, sxr0, sxc0, b);
    } else if (c0 < twoP) {
        brk_take((e00 + e01) * 0.5f, (fabsf_(e00) + fabsf_(e01)) * 0.5f, cm2, cm2k, sxr0, sxc0, b);
        if (r0 + 1 < m) brk_take((e10 + e11) * 0.5f, (fabsf_(e10) + fabsf_(e11)) * 0.5f, cm2, cm2k, sxr1, sxc0, b);
    } else {
        const bool c1ok = (c0 + 1 < m), r1ok = (r0 + 1 < m);
        if (r0 > c0) {
            brk_take(e00, fabsf_(e00), cm2, cm2k, sxr0, sxc0, b);
            if (c1ok) brk_take(e01, fabsf_(e01), cm2, cm2k, sxr0, sxc1, b);
            if (r1ok) {
                brk_take(e10, fabsf_(e10), cm2, cm2k, sxr1, sxc0, b);
                if (c1ok) brk_take(e11, fabsf_(e11), cm2, cm2k, sxr1, sxc1, b);
            }
        } else if (r1ok) brk_take(e10, fabsf_(e10), cm2, cm2k, sxr1, sxc0, b);
    }
}

// The same block for matrices without negative entries: affine brackets (screen_k1 / screen_k2).
// `emit(row slot, column slot, lb)` is called for every pair of the block with its lower bound
// (the lookahead window's emission pass; a no-op functor for plain bracketing).
template <class Emit>
FNN_HD void screen_micro_nn(int32_t r0, int32_t c0, int32_t m, int32_t twoP, float k1, float k2,
                            float e00, float e01, float e10, float e11,
                            float sxr0, float sxr1, float sxc0, float sxc1, Brk& b, Emit&& emit) {
    if (r0 >= m || c0 >= m || c0 > r0) return;
    auto take = [&](float mean, float sp, float sq, int32_t rs, int32_t cs) {
        const float lb = __builtin_fmaf(k1, mean, -sp) - sq;
        const float ub = __builtin_fmaf(k2, mean, -sp) - sq;
        b.lb = fminf_(b.lb, lb);
        b.ub = fminf_(b.ub, ub);
        emit(rs, cs, lb);
    };
    if (r0 < twoP) {
        if (r0 == c0) return;
        take((((e00 + e01) + e10) + e11) * 0.25f, sxr0, sxc0, r0, c0);
    } else if (c0 < twoP) {
        take((e00 + e01) * 0.5f, sxr0, sxc0, r0, c0);
        if (r0 + 1 < m) take((e10 + e11) * 0.5f, sxr1, sxc0, r0 + 1, c0);
    } else {
        const bool c1ok = (c0 + 1 < m), r1ok = (r0 + 1 < m);
        if (r0 > c0) {
            take(e00, sxr0, sxc0, r0, c0);
            if (c1ok) take(e01, sxr0, sxc1, r0, c0 + 1);
            if (r1ok) {
                take(e10, sxr1, sxc0, r0 + 1, c0);
                if (c1ok) take(e11, sxr1, sxc1, r0 + 1, c0 + 1);
            }
        } else if (r1ok) take(e10, sxr1, sxc0, r0 + 1, c0);
    }
}
struct NoEmit { FNN_HD void operator()(int32_t, int32_t, float) const {} };

// triangular tile bookkeeping, shared by the exact scan (R = 16) and the screening pass (R = 32):
// row tiles of TH rows come in bands of R; a row tile of band g owns g + 1 column tiles
FNN_HD int32_t tri_tile_count(int32_t m, int32_t TH, int32_t R) {
    const int32_t nrt = (m + TH - 1) / TH;
    const int32_t G = nrt / R, rr = nrt % R;
    return (R / 2) * G * (G + 1) + rr * (G + 1);
}
FNN_HD void tri_tile_decode(int32_t t, int32_t R, int32_t& rt, int32_t& ct) {
    // largest g with (R/2) g (g+1) <= t; the float estimate is corrected by the two loops
    int32_t g = (int32_t)((__builtin_sqrtf(1.0f + (float)t * (8.0f / (float)R)) - 1.0f) * 0.5f);
    if (g < 0) g = 0;
    while ((R / 2) * (g + 1) * (g + 2) <= t) g++;
    while (g > 0 && (R / 2) * g * (g + 1) > t) g--;
    const int32_t r = t - (R / 2) * g * (g + 1);
    rt = R * g + r / (g + 1);
    ct = r % (g + 1);
}
// number of screening units for m live slots: 4 per screening tile
FNN_HD int32_t screen_unit_count(int32_t m) { return 4 * tri_tile_count(m, SCR_TH, SCR_TW / SCR_TH); }

// every store into the matrix keeps the bf16 copy in step
// (only the lower triangle of the copy, diagonal included, is ever read: the screening pass and the
//  emission pass look at 2 x 2 blocks at or below the diagonal and use no entry above it)
FNN_HD void store_d(const Dev& d, int64_t r, int64_t c, double v) {
    const int64_t idx = r * d.ld + c;
    d.D[idx] = v;
    if (d.H && c <= r) d.H[r * d.ldh + c] = bf16_from_double(v);
}

// ---------------------------------------------------------------------------
// Lookahead windows: one screening pass serves up to K events.
//
// For a matrix without negative entries and two clusters P, Q that take part in none of the
// events t .. t+k-1:
//   * D(P,Q) does not change (only rows / columns of merged nodes are rewritten);
//   * the coefficient c - 2 drops by one per event, so (c_t - 2 - k) D(P,Q) >= (c_t - 2 - K) D(P,Q);
//   * S_P never grows: an event replaces the distances to the two merged clusters X, Y by the
//     distance to the new cluster, and D(P,new) <= D(P,X) + D(P,Y) for all three merge shapes
//     (2-way: (x+y)/2;  3-way: (x+y+z)/3 against x + (y+z)/2;  4-way: (2(x2+x+y)/3 + y2)/3
//     against (x2+x)/2 + (y+y2)/2; all entries >= 0), up to a few ulps of rounding.
// Hence  Q_{t+k}(P,Q) >= (c_t - 2 - K) D(P,Q) - S_P(t) - S_Q(t)  for every k <= K, and the
// screening pass of event t, run with the coefficient (c_t - 2 - K)(1 - kappa) (screen_k1),
// yields a lower bound LB_K of the pair's Q in all of the next K events.
//
// A base scan therefore also EMITS every pair with LB_K <= theta (the "tracked" pairs, stored
// as the two representatives' node ids) where theta = previous minimum + W.  In the following
// events k_track evaluates exactly (fp64, scan_micro: the scan's own body, exact tie-break)
//   (a) the tracked pairs whose two clusters still exist unchanged, and
//   (b) every pair that involves a cluster created since the base scan ("fresh"): the event after
//       its creation sweeps the new cluster's two rows in full, and every pair whose bound
//       (c_base - 2 - K) D - S_P - S_Q (exact fp64 this time) lies under theta joins the tracked
//       list; the others are untracked pairs like any other from then on.
// Every other live pair is an old, untracked pair with Q >= LB_K - slack > theta - slack =
// theta_eff.  So if the minimum M over (a) + (b) satisfies M <= theta_eff, it is the global
// minimum with the reference's tie-break (all ties of M are inside (a) + (b) too), and the event
// needs no scan at all.  If M > theta_eff, or after K events, the event runs a new base scan.
// The result never depends on W, K or the capacity: they only decide how often a base scan runs.
// ---------------------------------------------------------------------------
constexpr int LA_KMAX = 512;      // fresh clusters per window (>= K)
constexpr int LA_PCAP = 65536;    // tracked pairs per window
constexpr int LA_REC_INTS = 12;   // a tracked-pair record: 48 bytes (la_record)
constexpr int LA_LOGCAP = 8192;   // diagnostic records (one per base scan)

// Several ranks with lookahead windows: the matrix and the whole event chain are replicated; only a BASE SCAN is
// sharded (tile index mod world: screening pass, emission, exact rescans) and followed by ONE exchange.  A rank's
// exchange block: header {pairs emitted, rescan_all, candidate units, -}, its GATHER_RECS candidate records of the
// exact rescans, and the pairs it emitted (tracked-pair records, fnn_core.h: la_record); every rank then builds
// the same tracked list (rank 0's pairs, rank 1's, ...) and reduces the same candidate records.
constexpr int WX_HDR_WORDS = 4;
FNN_HD int32_t wx_pair_cap(int32_t world) { return LA_PCAP / (world > 0 ? world : 1); }
FNN_HD int64_t wx_recs_off() { return (int64_t)sizeof(int32_t) * WX_HDR_WORDS; }
FNN_HD int64_t wx_pairs_off() { return wx_recs_off() + (int64_t)sizeof(Cand) * GATHER_RECS; }
FNN_HD int64_t wx_block_bytes(int32_t world) { return wx_pairs_off() + (int64_t)sizeof(int32_t) * LA_REC_INTS * (int64_t)wx_pair_cap(world); }

// slack between the fp32 lower bound and the true fp64 Q over the window: the screening slack
// (screen_delta, applied twice for good measure) plus the rounding drift of the row sums
// (3 roundings of a value <= n Dmax per event, each <= 2^-53 relative)
FNN_HD double la_delta(const State& st) {
    const double dmax = __builtin_bit_cast(double, st.dmax_bits);
    return 2.0 * (double)screen_delta(st) + 1e-15 * (double)(st.la_K + 1) * ((double)st.n + 4.0) * dmax + 1e-30;
}

// does k_track serve this event from the open window?
FNN_HD bool la_active(const State& st) {
    return st.la_on && st.la_valid && !st.done && st.la_k <= st.la_Kcur && st.la_nf <= LA_KMAX && st.m >= st.la_min_m &&
           st.m > 4;
}

// this event runs a scan: close the window; decide whether the screening pass opens a new one
FNN_HD void la_prepare_base(State& st, int32_t* lacnt) {
    st.la_hit = 0;
    if (st.la_valid) st.la_k_prev = st.la_k;
    st.la_valid = 0;
    st.la_emit = 0;
    {   // shorter windows as the problem shrinks
        const int32_t kc = st.la_kbase + st.m / st.la_kdiv;
        st.la_Kcur = kc < st.la_K ? kc : st.la_K;
    }
    *lacnt = 0;
    if (st.la_on && st.nonneg && st.screen_ok && !st.done && st.m >= st.la_min_m) {
        if (st.la_skip > 0) st.la_skip--;
        else if (st.la_have_mprev) {
            const double th = st.la_mprev + st.la_W;
            st.la_theta_pred = (float)th;
            if (st.la_theta_pred == st.la_theta_pred && st.la_theta_pred < 3e38f && st.la_theta_pred > -3e38f) st.la_emit = 1;
        }
    }
}

// after the scan of a base event (its exact minimum is already known): open the window
FNN_HD void la_close_base(State& st, double* lalog, const int32_t* lacnt) {
    if (lalog && st.n_base_scans < LA_LOGCAP) {
        double* r = lalog + 5 * st.n_base_scans;
        r[0] = (double)st.n_events; r[1] = (double)st.m; r[2] = st.la_emit ? st.la_W : -1.0;
        r[3] = st.la_emit ? (double)*lacnt : -1.0; r[4] = (double)st.la_k_prev;
    }
    st.n_base_scans++;
    if (!st.la_emit) return;
    st.la_emit = 0;
    const int32_t cnt = *lacnt;
    const double dmax = __builtin_bit_cast(double, st.dmax_bits);
    const int32_t prev_end = st.la_prev_end;
    st.la_prev_end = 0;
    if (cnt > st.la_pcap) {  // too many pairs under the threshold: narrower next time, and back off
        st.n_la_overflow++;
        st.la_W *= 0.85;
        st.la_backoff = st.la_backoff < 1 ? 1 : (st.la_backoff >= 512 ? 1024 : 2 * st.la_backoff);
        st.la_skip = st.la_backoff;
        return;
    }
    st.la_backoff = 0;
    st.la_valid = 1;
    st.la_np = cnt;
    st.la_k = 0;
    st.la_nf = 0;
    st.la_nf_done = 0;
    st.la_base_stamp = (int32_t)st.n_events;
    st.la_theta_eff = (double)st.la_theta_pred - la_delta(st);
    st.la_coef = (double)st.c - 2.0 - (double)st.la_Kcur;
    if (st.la_coef < 0.0) st.la_coef = 0.0;
    st.la_pairs_sum += cnt;
    // width for the next window: wide enough to last its K events, narrow enough for the list.  The
    // number of pairs under the threshold grows very steeply with the width (x 30 for + 25 %), so
    // the steps are small
    if (prev_end == 2) st.la_W *= 0.85;
    else if (prev_end == 1 && cnt < st.la_pcap / 2) st.la_W *= 1.1;
    else if (cnt > st.la_target) st.la_W *= 0.95;
    const double wmin = 1e-6 * dmax + 1e-300, wmax = 1e12 * dmax + 1e-300;
    if (!(st.la_W >= wmin)) st.la_W = wmin;
    if (st.la_W > wmax) st.la_W = wmax;
}

// A tracked pair (48 bytes): the two representatives' node ids with their "paired" status in the top bit, the slots
// they sat in when the pair was recorded (hints: the tracking pass checks that the ids still sit there; only a moved node
// costs a look-up in islot), and the pair's 2 x 2 block of matrix entries D[a][b], D[a][b'], D[a'][b], D[a'][b']
// (a', b' the partners; unused entries 0).  The window's premise is that the distance of two clusters that take part in
// no event does not change - slot moves copy entries bit for bit - so the entries are fetched ONCE, when the pair is
// recorded, and the tracking pass of the following events reads no matrix entry at all: a record, two row sums, two
// positions (all from small, cache-resident arrays) instead of two random 16-byte reads in an 8 GiB matrix.
// A live node's cluster changes in one way only - a singleton becomes paired - so "same id, same paired
// status" means "same cluster" (a paired node keeps its partner until both disappear in a merge).
constexpr int32_t LA_PAIRED_BIT = (int32_t)0x80000000u;
FNN_HD void la_record(const Dev& d, int32_t* t, int32_t rs, int32_t cs, int32_t twoP) {
    const bool pa = rs < twoP, pb = cs < twoP;
    t[0] = d.sid[rs] | (pa ? LA_PAIRED_BIT : 0);
    t[1] = d.sid[cs] | (pb ? LA_PAIRED_BIT : 0);
    t[2] = rs;
    t[3] = cs;
    const double* R0 = d.D + (int64_t)rs * d.ld;
    const double* R1 = R0 + d.ld;  // (rows are padded: rs + 1 is in bounds)
    double e[4];
    e[0] = R0[cs];
    e[1] = pb ? R0[cs + 1] : 0.0;
    e[2] = pa ? R1[cs] : 0.0;
    e[3] = (pa && pb) ? R1[cs + 1] : 0.0;
    double* te = reinterpret_cast<double*>(t + 4);
    te[0] = e[0]; te[1] = e[1]; te[2] = e[2]; te[3] = e[3];
}
// append the pair of the representatives in slots rs, cs to the window's tracked list
FNN_HD void la_append(const Dev& d, int32_t rs, int32_t cs, int32_t twoP) {
    State& st = *d.st;
    const int32_t i = FNN_ATOMIC_INC(d.lacnt);
    if (i < st.la_pcap) la_record(d, d.tpairs + LA_REC_INTS * (int64_t)i, rs, cs, twoP);
}

// After the exchange of a sharded base scan: the tracked list from all ranks' emitted pairs, all ranks' candidate
// records into d.grecv, and the close of the base scan (serial form: CPU emulation; the GPU's k_merge is the same,
// spread over a workgroup).  A rank that found more pairs than its share of the list holds, or a total beyond the
// list's capacity, counts as an overflow: the window is not opened (la_close_base), nothing else changes.
FNN_HD void wx_merge(const Dev& d) {
    State& st = *d.st;
    if (st.la_hit) return;
    const int32_t capr = wx_pair_cap(d.world);
    const int64_t bb = wx_block_bytes(d.world);
    bool overflow = false;
    int64_t total = 0, units = 0;
    for (int32_t r = 0; r < d.world; r++) {
        const int32_t* h = reinterpret_cast<const int32_t*>(d.wrecv + r * bb);
        if (!st.done && h[3] != st.n_events) st.error = 12;  // a block of ANOTHER event: the ranks have taken different decisions
        if (h[0] > capr) overflow = true;
        total += h[0] < capr ? h[0] : capr;
        units += h[2];
    }
    if (total > st.la_pcap) overflow = true;
    if (st.la_emit && !overflow) {
        int64_t off = 0;
        for (int32_t r = 0; r < d.world; r++) {
            const int32_t* h = reinterpret_cast<const int32_t*>(d.wrecv + r * bb);
            const int32_t* src = reinterpret_cast<const int32_t*>(d.wrecv + r * bb + wx_pairs_off());
            for (int64_t i = 0; i < LA_REC_INTS * (int64_t)h[0]; i++) d.tpairs[LA_REC_INTS * off + i] = src[i];
            off += h[0];
        }
    }
    *d.lacnt = overflow ? st.la_pcap + 1 : (int32_t)total;
    for (int32_t r = 0; r < d.world; r++) {
        const Cand* src = reinterpret_cast<const Cand*>(d.wrecv + r * bb + wx_recs_off());
        for (int j = 0; j < GATHER_RECS; j++) d.grecv[r * GATHER_RECS + j] = src[j];
    }
    const int32_t* mine = reinterpret_cast<const int32_t*>(d.wrecv + d.rank * bb);
    st.rescan_all = mine[1];
    st.ncand = mine[1] ? 0 : mine[2];
    st.n_screen_events += 1;
    st.n_rescan_units += units;
    st.ev_screened = 1;
    la_close_base(st, d.lalog, d.lacnt);
}

// a new two-node cluster with representative id `rep` exists from the next event on
FNN_HD void la_note_cluster(const Dev& d, State& st, int32_t rep, int32_t partner) {
    if (!d.cstamp) return;
    const int32_t stamp = (int32_t)st.n_events + 1;
    d.cstamp[rep] = stamp;
    d.cstamp[partner] = stamp;
    if (st.la_valid) {
        if (st.la_nf < LA_KMAX) { d.fresh[3 * st.la_nf] = rep; d.fresh[3 * st.la_nf + 1] = stamp; d.fresh[3 * st.la_nf + 2] = st.U; }
        st.la_nf++;  // beyond LA_KMAX the window is no longer served (la_active)
    }
}

// all pairs between the fresh two-node cluster in slots (f0, f0 + 1) and the node(s) in slots
// (s, s + 1), s even: the four entries are read from ROWS f0, f0 + 1 (contiguous in s; the matrix
// is bit-symmetric) and handed to the scan's body in the orientation it would have met them.
// Besides competing for this event's minimum, a pair whose lower bound for the rest of the window
// lies under the threshold joins the tracked list.
struct SweepSink {
    const Dev& d;
    Cand& best;
    double coef, th;
    int32_t twoP;
    FNN_HD void operator()(double q, double dpq, double sxa, double sxb, int32_t pa, int32_t pb, int32_t rs, int32_t cs) const {
        consider(q, pa, pb, rs, cs, best);
        const double lbf = (coef * dpq - sxa) - sxb;
        if (lbf <= th) la_append(d, rs, cs, twoP);
    }
};
FNN_HD void fresh_eval(const Dev& d, int32_t f0, int32_t s, int32_t m, int32_t twoP, double cm2, double coef, double th,
                       double sxf0, double sxf1, Cand& best) {
    if (s >= m || s == f0) return;
    const double* F0 = d.D + (int64_t)f0 * d.ld + s;
    const double* F1 = F0 + d.ld;
    const double a0 = F0[0], a1 = F0[1], b0 = F1[0], b1 = F1[1];
    const SweepSink sink{d, best, coef, th, twoP};
    if (f0 > s)
        scan_micro_t(f0, s, m, twoP, cm2, a0, a1, b0, b1, sxf0, sxf1, d.spos[f0], d.spos[f0 + 1],
                     d.Sx[s], d.Sx[s + 1], d.spos[s], d.spos[s + 1], sink);
    else
        scan_micro_t(s, f0, m, twoP, cm2, a0, b0, a1, b1, d.Sx[s], d.Sx[s + 1], d.spos[s], d.spos[s + 1],
                     sxf0, sxf1, d.spos[f0], d.spos[f0 + 1], sink);
}

// one work item of k_track: item < np: tracked pair; otherwise (unswept fresh cluster, column pair)
struct TrackArgs {
    int32_t np, nf0, nf, m, twoP;
    double cm2, coef, th;
    // the swept cluster's exact row sum may still be on its way (k_track's chain workgroup); the sweep
    // then runs with the tree-ordered sum `sxu` and its pairs compete in a separate record
    int32_t approx;
    int32_t usl;  // slot of the cluster whose sum sxu is (-2: not checked)
    double sxu;
};
FNN_HD TrackArgs track_args(const State& st) {
    TrackArgs a;
    a.np = st.la_np;
    a.nf = st.la_nf < LA_KMAX ? st.la_nf : LA_KMAX;
    a.nf0 = st.la_nf_done < a.nf ? st.la_nf_done : a.nf;
    a.m = st.m;
    a.twoP = 2 * st.P;
    a.cm2 = (double)st.c - 2.0;
    a.coef = st.la_coef;
    a.th = (double)st.la_theta_pred;
    a.approx = 0;
    a.usl = -2;
    a.sxu = 0.0;
    return a;
}
FNN_HD int64_t track_item_count(const TrackArgs& a) { return (int64_t)a.np + (int64_t)(a.nf - a.nf0) * ((a.m + 1) / 2); }
// a tracked pair: evaluated exactly if both clusters still exist unchanged - the scan's own expressions
// (scan_micro_t: the 1-, 2- or 4-term cluster distance in the reference's term order, Q with the row sums taken by
// reference position, the (Q, i, j) tie-break) on the entries the record carries
struct PairRec { int32_t wa, wb, sa, sb; double e[4]; };
FNN_HD PairRec track_pair_load(const Dev& d, int64_t item) {
    const int32_t* t = d.tpairs + LA_REC_INTS * item;
    PairRec r;
    r.wa = t[0]; r.wb = t[1]; r.sa = t[2]; r.sb = t[3];
    const double* te = reinterpret_cast<const double*>(t + 4);
    r.e[0] = te[0]; r.e[1] = te[1]; r.e[2] = te[2]; r.e[3] = te[3];
    return r;
}
FNN_HD void track_pair_rec(const Dev& d, const PairRec& r, const TrackArgs& a, Cand& best) {
    const int32_t wa = r.wa, wb = r.wb;
    const int32_t ia = wa & ~LA_PAIRED_BIT, ib = wb & ~LA_PAIRED_BIT;
    int32_t sa = r.sa, sb = r.sb;
    // (a slot beyond the live range may still carry the id of a node that was moved out of it)
    if (!(sa >= 0 && sb >= 0 && sa < a.m && sb < a.m) || d.sid[sa] != ia || d.sid[sb] != ib) {
        sa = d.islot[ia]; sb = d.islot[ib];
        if (sa < 0 || sb < 0) return;  // a cluster is gone
    }
    const bool pa = wa < 0, pb = wb < 0;
    if ((sa < a.twoP) != pa || (sb < a.twoP) != pb) return;  // a singleton has become half of a new cluster
    const double sxa = d.Sx[sa], sxb = d.Sx[sb];
    const int32_t posa = d.spos[sa], posb = d.spos[sb];
    // e[.] = D[a][b], D[a][b'], D[a'][b], D[a'][b'].  The scan meets the pair in the micro-tile whose ROWS are the
    // cluster in the larger slots; entry (row node i, column node j) of that tile:
    const bool arow = sa > sb;
    const double e00 = r.e[0], e01 = arow ? r.e[1] : r.e[2], e10 = arow ? r.e[2] : r.e[1], e11 = r.e[3];
    const double sxr = arow ? sxa : sxb, sxc = arow ? sxb : sxa;
    const int32_t pr = arow ? posa : posb, pc = arow ? posb : posa;
    const int32_t srow = arow ? sa : sb, scol = arow ? sb : sa;
    double dpq;
    if (pa && pb) dpq = pr > pc ? (((e00 + e01) + e10) + e11) / 4.0 : (((e00 + e10) + e01) + e11) / 4.0;  // (:169)
    else if (pa || pb) dpq = (e00 + (pa == arow ? e10 : e01)) / 2.0;  // singleton x pair: the singleton's two entries (:165/:167)
    else dpq = e00;                                                    // (:163)
    consider(qval(a.cm2, dpq, sxr, pr, sxc, pc), pr, pc, srow, scol, best);
}
FNN_HD void track_pair_item(const Dev& d, int64_t item, const TrackArgs& a, Cand& best) {
    track_pair_rec(d, track_pair_load(d, item), a, best);
}
// the slot of the fi-th fresh cluster's representative, -1 if the cluster has been consumed by a later event
FNN_HD int32_t fresh_slot(const Dev& d, int32_t fi, int32_t m) {
    const int32_t id = d.fresh[3 * fi], stamp = d.fresh[3 * fi + 1];
    int32_t f0 = d.fresh[3 * fi + 2];
    if (f0 >= m || d.sid[f0] != id) {
        f0 = d.islot[id];
        if (f0 < 0 || d.cstamp[id] != stamp) return -1;
    }
    return f0;
}
// item r of the sweep of the new clusters' rows.  `bestu` receives the swept pairs when the sweep runs
// on an approximate row sum (a.approx); false: the swept cluster is not the one whose sum a.sxu is.
FNN_HD bool track_sweep_item(const Dev& d, int64_t r, const TrackArgs& a, Cand& best, Cand& bestu) {
    const int32_t half = (a.m + 1) / 2;
    const int32_t fi = a.nf0 + (int32_t)(r / half), cp = (int32_t)(r % half);
    if (fi >= a.nf) return true;
    const int32_t f0 = fresh_slot(d, fi, a.m);
    if (f0 < 0) return true;
    // (insertion with the approximate sum: its error, ~1e-16 m n Dmax, is far inside the gap la_delta ~ 1e-6 n Dmax
    //  between the insertion threshold theta_pred and the acceptance threshold theta_eff)
    if (a.approx) {
        if (a.usl != -2 && f0 != a.usl) return false;
        fresh_eval(d, f0, 2 * cp, a.m, a.twoP, a.cm2, a.coef, a.th, a.sxu, a.sxu, bestu);
    } else fresh_eval(d, f0, 2 * cp, a.m, a.twoP, a.cm2, a.coef, a.th, d.Sx[f0], d.Sx[f0 + 1], best);
    return true;
}
FNN_HD void track_item(const Dev& d, int64_t item, const TrackArgs& a, Cand& best, Cand& bestu) {
    if (item < a.np) track_pair_item(d, item, a, best);
    else (void)track_sweep_item(d, item - a.np, a, best, bestu);
}
FNN_HD void track_item(const Dev& d, int64_t item, const TrackArgs& a, Cand& best) { track_item(d, item, a, best, best); }

// the window's verdict on the minimum over all items (one thread, after the reduction)
FNN_HD void la_track_hit(const Dev& d, const TrackArgs& a);
FNN_HD void la_track_done(const Dev& d, Cand best, const TrackArgs& a) {
    State& st = *d.st;
    if (best.q <= st.la_theta_eff && st.n_events != d.fault_event) {
        d.recs[0] = best;
        la_track_hit(d, a);
    } else {
        st.n_la_fail++;
        st.la_prev_end = 1;
        la_prepare_base(st, d.lacnt);
    }
}
FNN_HD void la_track_hit(const Dev& d, const TrackArgs& a) {
    State& st = *d.st;
    {
        st.la_hit = 1;
        st.n_la_hits++;
        const int64_t items = track_item_count(a);
        st.la_items_sum += items;
        st.bytes_streamed += 32 * items;
        st.la_nf_done = a.nf;
        // (the sweep's workgroups of THIS launch added to the counter: read it where they added, FNN_COUNTER_READ; every
        //  append's add had returned before its workgroup's arrival was counted, so the value is complete)
        const int32_t cnt = FNN_COUNTER_READ(d.lacnt);
        if (cnt < st.la_np) st.error = 11;  // (the list only grows within a window: a smaller count would be a stale read)
        if (cnt > st.la_pcap) {  // the sweep found more pairs than the list can take: this
            st.la_valid = 0;             // event is served, the next one opens a new window
            st.la_k_prev = st.la_k;
            st.la_prev_end = 2;
            st.n_la_overflow++;
        } else st.la_np = cnt;
    }
}

// ---------------------------------------------------------------------------
// plan building (single thread)
//
// The integer side of an event reads and writes the slot tables (sid, spos, pslot) a few dozen
// times, each access depending on the previous one.  On the GPU every such access to memory is a
// round trip of ~1 us for the single planning thread, so the tables are addressed through an
// accessor: GlobalTab goes to memory; CachedTab holds the entries an event can touch - the slots of
// Cx, Cy and their partners, the two slots at the pair / singleton boundary on either side, the last
// two live slots, and pslot of the last two positions (see tab_keys) - preloaded with ONE batch of
// independent loads, writes through to memory, and falls back to memory on a miss.
// ---------------------------------------------------------------------------
struct GlobalTab {
    const Dev& d;
    FNN_HD int32_t sid(int32_t s) const { return d.sid[s]; }
    FNN_HD int32_t spos(int32_t s) const { return d.spos[s]; }
    FNN_HD int32_t pslot(int32_t p) const { return d.pslot[p]; }
    FNN_HD void set_sid(int32_t s, int32_t v) { d.sid[s] = v; }
    FNN_HD void set_spos(int32_t s, int32_t v) { d.spos[s] = v; }
    FNN_HD void set_pslot(int32_t p, int32_t v) { d.pslot[p] = v; }
};
constexpr int TAB_NK = 10, TAB_NP = 2;
// the slots (keys) and positions an event with candidates in slots a, b can read; -1 = none
FNN_HD void tab_keys(int32_t a, int32_t b, int32_t P, int32_t m, int32_t key[TAB_NK], int32_t pkey[TAB_NP]) {
    key[0] = a; key[1] = a ^ 1; key[2] = b; key[3] = b ^ 1;
    key[4] = 2 * P; key[5] = 2 * P + 1; key[6] = 2 * P - 2; key[7] = 2 * P - 1;
    key[8] = m - 1; key[9] = m - 2;
    pkey[0] = m - 1; pkey[1] = m - 2;
}
struct CachedTab {
    const Dev& d;
    int32_t *key, *vsid, *vspos;  // [TAB_NK]   (the GPU keeps these arrays in LDS)
    int32_t *pkey, *vpslot;       // [TAB_NP]
    int32_t* misses;
    FNN_HD int32_t sid(int32_t s) {
        int32_t r = 0; bool hit = false;
        for (int i = 0; i < TAB_NK; i++) if (key[i] == s) { r = vsid[i]; hit = true; }
        if (!hit) { r = d.sid[s]; ++*misses; }
        return r;
    }
    FNN_HD int32_t spos(int32_t s) {
        int32_t r = 0; bool hit = false;
        for (int i = 0; i < TAB_NK; i++) if (key[i] == s) { r = vspos[i]; hit = true; }
        if (!hit) { r = d.spos[s]; ++*misses; }
        return r;
    }
    FNN_HD int32_t pslot(int32_t p) {
        int32_t r = 0; bool hit = false;
        for (int i = 0; i < TAB_NP; i++) if (pkey[i] == p) { r = vpslot[i]; hit = true; }
        if (!hit) { r = d.pslot[p]; ++*misses; }
        return r;
    }
    FNN_HD void set_sid(int32_t s, int32_t v) {
        for (int i = 0; i < TAB_NK; i++) if (key[i] == s) vsid[i] = v;
        d.sid[s] = v;
    }
    FNN_HD void set_spos(int32_t s, int32_t v) {
        for (int i = 0; i < TAB_NK; i++) if (key[i] == s) vspos[i] = v;
        d.spos[s] = v;
    }
    FNN_HD void set_pslot(int32_t p, int32_t v) {
        for (int i = 0; i < TAB_NP; i++) if (pkey[i] == p) vpslot[i] = v;
        d.pslot[p] = v;
    }
};
// fill a CachedTab straight from memory (CPU emulation; the GPU loads the same entries lane-parallel)
FNN_HD void tab_preload(CachedTab& T, int32_t a, int32_t b, int32_t P, int32_t m) {
    tab_keys(a, b, P, m, T.key, T.pkey);
    *T.misses = 0;
    for (int i = 0; i < TAB_NK; i++) {
        const bool ok = T.key[i] >= 0 && T.key[i] < T.d.n;
        if (!ok) T.key[i] = -1;
        T.vsid[i] = ok ? T.d.sid[T.key[i]] : 0;
        T.vspos[i] = ok ? T.d.spos[T.key[i]] : 0;
    }
    for (int i = 0; i < TAB_NP; i++) {
        const bool ok = T.pkey[i] >= 0 && T.pkey[i] < T.d.n;
        if (!ok) T.pkey[i] = -1;
        T.vpslot[i] = ok ? T.d.pslot[T.pkey[i]] : 0;
    }
}

FNN_HD void emit(State& st, int32_t kind, int32_t a, int32_t b, int32_t c, int32_t d, int32_t e,
                 int32_t flag) {
    if (st.nops >= MAX_OPS) { st.error = 2; return; }
    Op& o = st.ops[st.nops++];
    o.kind = kind; o.a = a; o.b = b; o.c = c; o.d = d; o.e = e;
    o.mcur = st.m_old;
    o.flag = flag;
}

// (values that were just stored are kept in locals, not read back)
template <class Tab>
FNN_HD void swap_slots(const Dev& d, Tab& T, int32_t s1, int32_t s2) {
    const int32_t id1 = T.sid(s1), id2 = T.sid(s2), p1 = T.spos(s1), p2 = T.spos(s2);
    T.set_sid(s1, id2); T.set_sid(s2, id1);
    T.set_spos(s1, p2); T.set_spos(s2, p1);
    T.set_pslot(p2, s1);
    T.set_pslot(p1, s2);
    if (d.islot) { d.islot[id2] = s1; d.islot[id1] = s2; }
    emit(*d.st, OP_SWAP, s1, s2, 0, 0, 0, 0);
}

template <class Tab>
FNN_HD void move_slot(const Dev& d, Tab& T, int32_t src, int32_t dst) {
    const int32_t id = T.sid(src), ps = T.spos(src);
    T.set_sid(dst, id);
    T.set_spos(dst, ps);
    T.set_pslot(ps, dst);
    if (d.islot) d.islot[id] = dst;
    emit(*d.st, OP_MOVE, src, dst, 0, 0, 0, 0);
}

// Integer side of agg3way(x, y, z, ..., num_nodes = nn, num_active = mc)
// (NetMakerOriginal.java:608-648): nodes x, y, z sit in slots X, Y, Z; the new
// nodes u (id nn+1) and v (id nn+2) are put in slots U and V, both in {X,Y,Z}.
template <class Tab>
FNN_HD void agg3_plan(const Dev& d, Tab& T, int32_t X, int32_t Y, int32_t Z, int32_t U, int32_t V,
                      int32_t nn, int32_t mc) {
    State& st = *d.st;
    int32_t px = T.spos(X), py = T.spos(Y), pz = T.spos(Z);
    Agg3Rec& r = d.agglog[st.n_agg3++];
    const int32_t idx = T.sid(X), idy = T.sid(Y), idz = T.sid(Z);
    r.u_id = nn + 1; r.x_id = idx; r.y_id = idy; r.z_id = idz;
    if (d.islot) { d.islot[idx] = -1; d.islot[idy] = -1; d.islot[idz] = -1; d.islot[nn + 1] = U; d.islot[nn + 2] = V; }
    T.set_sid(U, nn + 1);  // u replaces x in the list (:623-625)
    T.set_sid(V, nn + 2);  // v replaces z (:630-632)
    T.set_spos(U, px);
    T.set_spos(V, pz);
    T.set_pslot(px, U);
    T.set_pslot(pz, V);
    // remove y: netNodes[y.pos] = netNodes[mc-1] (:641-643)
    int32_t last = T.pslot(mc - 1);
    int32_t pu = px, pv = pz;  // = spos[U], spos[V] as they stand in memory
    if (py != mc - 1) {
        T.set_spos(last, py);
        T.set_pslot(py, last);
        if (last == U) pu = py;
        if (last == V) pv = py;
        if (st.pov_n < 2) { st.pov_slot[st.pov_n] = last; st.pov_pos[st.pov_n] = py; st.pov_n++; }
        else st.error = 15;  // (an event has at most two agg3way calls)
    }
    T.set_pslot(mc - 1, -1);
    emit(st, OP_AGG3, X, Y, Z, U, V, pu < pv ? 1 : 0);
}

// Symbolic replay of the event's micro-ops: which old rows does every involved slot hold
// afterwards?  Fills st.S (involved slots) and st.tgt (rows that change and stay live).
FNN_HD void build_targets(const Dev& d) {
    State& st = *d.st;
    st.nS = 0; st.ntgt = 0; st.tU = -1; st.tV = -1;
    auto add_slot = [&](int32_t sl) {
        for (int i = 0; i < st.nS; i++) if (st.S[i] == sl) return;
        if (st.nS >= MAX_S) { st.error = 5; return; }
        st.S[st.nS++] = sl;
    };
    add_slot(st.U); add_slot(st.U + 1);
    for (int i = 0; i < st.nops; i++) {
        const Op& o = st.ops[i];
        add_slot(o.a); add_slot(o.b);
        if (o.kind == OP_AGG3) { add_slot(o.c); add_slot(o.d); add_slot(o.e); }
    }
    Tgt sym[MAX_S];
    for (int i = 0; i < st.nS; i++) { sym[i].dst = st.S[i]; sym[i].kind = T_COPY; sym[i].a = st.S[i]; sym[i].b = sym[i].c = sym[i].d = -1; }
    auto idx = [&](int32_t sl) { for (int i = 0; i < st.nS; i++) if (st.S[i] == sl) return i; return 0; };
    // value (2/3)*A + B/3
    auto comb = [&](const Tgt& A, const Tgt& B) {
        Tgt r; r.dst = -1; r.kind = T_COPY; r.a = r.b = r.c = r.d = -1;
        if (A.kind == T_COPY && B.kind == T_COPY) { r.kind = T_L1; r.a = A.a; r.b = B.a; }
        else if (A.kind == T_L1 && B.kind == T_L1 && A.b == B.b) { r.kind = T_L2U; r.a = A.a; r.b = A.b; r.c = B.a; }
        else if (A.kind == T_COPY && B.kind == T_L1) { r.kind = T_L2V; r.d = A.a; r.c = B.a; r.b = B.b; }
        else st.error = 6;
        return r;
    };
    for (int i = 0; i < st.nops; i++) {
        const Op& o = st.ops[i];
        if (o.kind == OP_SWAP) { int ia = idx(o.a), ib = idx(o.b); Tgt t = sym[ia]; sym[ia] = sym[ib]; sym[ib] = t; }
        else if (o.kind == OP_MOVE) { sym[idx(o.b)] = sym[idx(o.a)]; }
        else if (o.kind == OP_AGG3) {
            Tgt sx = sym[idx(o.a)], sy = sym[idx(o.b)], sz = sym[idx(o.c)];
            Tgt nu = comb(sx, sy), nv = comb(sz, sy);
            sym[idx(o.d)] = nu; sym[idx(o.e)] = nv;
        }
    }
    for (int i = 0; i < st.nS; i++) {
        const int32_t sl = st.S[i];
        const bool isUV = (sl == st.U || sl == st.U + 1);
        if (sl >= st.m && !st.ev_finish) continue;  // not live after the event
        if (!isUV && sym[i].kind == T_COPY && sym[i].a == sl) continue;  // unchanged
        if (st.ntgt >= MAX_TGT) { st.error = 7; return; }
        Tgt t = sym[i];
        t.dst = sl;
        if (sl == st.U) st.tU = st.ntgt;
        if (sl == st.U + 1) st.tV = st.ntgt;
        st.tgt[st.ntgt++] = t;
    }
}

// Special finish, NetMakerOriginal.java:343-360 (num_active == 4, num_clusters == 2); rare: tables in memory
FNN_HD void finish_plan(const Dev& d) {
    State& st = *d.st;
    const double* D = d.D; const int64_t ld = d.ld;
    GlobalTab T{d};
    st.ev_finish = 1;
    st.need_rx = 0;
    int32_t ps = d.pslot[0];
    int32_t qs = d.pslot[1];
    if (qs == (ps ^ 1)) qs = d.pslot[2];
    int32_t pn = ps ^ 1, qn = qs ^ 1;
    int32_t X = ps, Y, Z;
    if (D[ps * ld + qs] + D[pn * ld + qn] < D[ps * ld + qn] + D[pn * ld + qs]) { Y = qs; Z = qn; }
    else { Y = qn; Z = qs; }
    st.cur.kind = KIND_FINISH;
    st.cur.x_id = d.sid[ps];
    st.cur.y_id = d.sid[Y];
    st.cur.u_id = st.num_nodes + 1;
    int32_t k = qs >> 1;
    agg3_plan(d, T, X, Y, Z, 2 * k, 2 * k + 1, st.num_nodes, 4);
    st.num_nodes += 2;
    st.U = 2 * k;
    // (the caller replays the micro-op: build_targets / build_targets_wave)
}

// After the scan: turn the best candidate into Cx, Cy (NetMakerOriginal.java:376-380).
// a, b: the slots pslot[i], pslot[j] of the candidate's positions; ida, idb: their node ids.
FNN_HD void pick(const Dev& d, Cand best, int32_t a, int32_t b, int32_t ida, int32_t idb) {
    State& st = *d.st;
    st.chain_pending = 0;  // (the previous event's u.Sx is delivered by k_track's chain workgroup before the next kernel)
    st.ev_active = 0;
    if (st.done) return;
    if (st.m <= 3) { st.done = 1; return; }
    st.ev_active = 1;
    st.ev_finish = 0;
    st.nops = 0;
    st.pov_n = 0;
    st.m_old = st.m; st.P_old = st.P; st.c_old = st.c;
    Event& cur = st.cur;
    cur.m_before = st.m; cur.c_before = st.c;
    cur.cx_id = cur.cy_id = cur.x_id = cur.y_id = cur.kind = cur.u_id = 0;
    cur.best = 0.0; cur.entries = 0;
    if (st.m == 4 && st.c == 2) { finish_plan(d); return; }
    cur.entries = (int64_t)st.m * (st.m - 1) / 2 - (st.m - st.c);
    if (st.rl_active) { cur.entries = st.rl_evals; st.n_rl_events++; }  // (Relaxed mode: the search's own count)
    cur.best = best.q;
    if (!st.la_hit) {  // (a window hit has already added the bytes k_track read)
        const int64_t bytes = (st.ev_screened ? 2 : 8) * cur.entries +
                              (st.ev_screened ? (int64_t)(st.rescan_all ? 0 : st.ncand) * SCR_TH * SCR_UW * 8 : 0);
        st.bytes_streamed += bytes;
        if (!st.ev_screened) st.bytes_plain += bytes;
        else if (st.ev_timed) st.bytes_timed += bytes;
    }
    st.ev_screened = 0;
    if (ida > idb) { int32_t t = a; a = b; b = t; t = ida; ida = idb; idb = t; }  // Cx.id < Cy.id
    int32_t twoP = 2 * st.P;
    st.sa = a; st.sb = b;
    st.sap = a < twoP ? (a ^ 1) : -1;
    st.sbp = b < twoP ? (b ^ 1) : -1;
    st.need_rx = (st.sap >= 0 || st.sbp >= 0) ? 1 : 0;
    cur.cx_id = ida;
    cur.cy_id = idb;
}
// does the event that `pick` would open need a candidate at all? (loop ended / special finish: no)
FNN_HD bool pick_needs_candidate(const State& st) { return !st.done && st.m > 3 && !(st.m == 4 && st.c == 2); }

// ComputeRx term for slot s (NetMakerOriginal.java:555-558), written to the chain
// buffer at the node's reference position (the exact sequential sums read them there)
FNN_HD void rx_fill_thread(const Dev& d, int32_t s, int32_t m, int32_t twoP, const int32_t z[4]) {
    if (s >= m) return;
    bool full = (s == z[0] || s == z[1] || s == z[2] || s == z[3] || s >= twoP);
    int32_t pos = d.spos[s];
    for (int k = 0; k < 4; k++) {
        if (z[k] < 0) continue;
        double v = d.D[(int64_t)z[k] * d.ld + s];
        d.chain[(int64_t)(k + 1) * d.cstride + chain_addr(pos)] = full ? v : v / 2.0;
    }
}

// the same for ONE of the four nodes (buffer k + 1): a helper workgroup of k_track gathers its own row
FNN_HD void rx_fill_one(const Dev& d, int32_t s, int32_t m, int32_t twoP, const int32_t z[4], int k, int32_t zk) {
    if (s >= m || zk < 0) return;
    const bool full = (s == z[0] || s == z[1] || s == z[2] || s == z[3] || s >= twoP);
    const double v = d.D[(int64_t)zk * d.ld + s];
    d.chain[(int64_t)(k + 1) * d.cstride + chain_addr(d.spos[s])] = full ? v : v / 2.0;
}

// Rx = 0.0; for i in position order: Rx += term  (sequential, :551-560)
FNN_HD double chain_sum(const double* buf, int32_t m) {
    double s = 0.0;
    for (int32_t i = 0; i < m; i++) s += buf[chain_addr(i)];
    return s;
}

// ---------------------------------------------------------------------------
// The 4-candidate choice without an O(m) pass.
//
// ComputeRx(z) (:549-561) = sum over the live nodes p of w'(p) D[z][p], w' = 1 for singletons and
// for the (up to four) nodes of the two candidate clusters, 1/2 for every other paired node.  The
// engine maintains, per node z, the APPROXIMATE weighted row sum
//     T(z) = sum over live p of w(p) D[z][p],   w = 1 for singletons, 1/2 for paired nodes
// incrementally in the fused update (every bystander's T changes by the old / new entries towards
// the merged nodes, which that kernel has in registers anyway; the two new nodes get a fresh tree
// sum).  Then  Rx(z) = T(z) + 1/2 sum over the PAIRED nodes k among {Cx, Cx.nbr, Cy, Cy.nbr} of
// D[z][k]: six matrix entries instead of four row sweeps.  T is only ever used to CERTIFY the
// comparisons of :428-452 (rx_certify): |T_computed - T_real| is bounded by the initial sequential
// sum (n eps n Dmax), a fresh tree sum (m eps m Dmax) and the drift of <= eps (n + 64) Dmax per
// event; the reference's own sequential sum is within m eps m Dmax of the real one.  If any two
// candidates are closer than the sum of their bounds the exact sequential sums decide (rare).
// ---------------------------------------------------------------------------
struct Quad {        // what the decision reads besides the control block
    double Tz[4];    // T of Cx, Cx.nbr, Cy, Cy.nbr (0 where absent)
    double Dab, Dapb, Dabp, Dapbp;  // D[Cx][Cy], D[Cx.nbr][Cy], D[Cx][Cy.nbr], D[Cx.nbr][Cy.nbr]
    double Daap, Dbbp;              // D[Cx][Cx.nbr], D[Cy][Cy.nbr]
};
FNN_HD void quad_load(const Dev& d, Quad& q) {  // (CPU emulation / rare paths; the GPU loads these lane-parallel)
    const State& st = *d.st;
    const double* D = d.D; const int64_t ld = d.ld;
    const int32_t a = st.sa, ap = st.sap, b = st.sb, bp = st.sbp;
    q.Tz[0] = d.T[a]; q.Tz[1] = ap >= 0 ? d.T[ap] : 0.0; q.Tz[2] = d.T[b]; q.Tz[3] = bp >= 0 ? d.T[bp] : 0.0;
    q.Dab = D[a * ld + b];
    q.Dapb = ap >= 0 ? D[ap * ld + b] : 0.0;
    q.Dabp = bp >= 0 ? D[a * ld + bp] : 0.0;
    q.Dapbp = (ap >= 0 && bp >= 0) ? D[ap * ld + bp] : 0.0;
    q.Daap = ap >= 0 ? D[a * ld + ap] : 0.0;
    q.Dbbp = bp >= 0 ? D[b * ld + bp] : 0.0;
}
FNN_HD void rx_from_T(const State& st, const Quad& q, double rx[4]) {
    const bool pa = st.sap >= 0, pb = st.sbp >= 0;
    rx[0] = q.Tz[0] + 0.5 * ((pa ? q.Daap : 0.0) + (pb ? q.Dab + q.Dabp : 0.0));
    rx[1] = pa ? q.Tz[1] + 0.5 * (q.Daap + (pb ? q.Dapb + q.Dapbp : 0.0)) : 0.0;
    rx[2] = q.Tz[2] + 0.5 * ((pb ? q.Dbbp : 0.0) + (pa ? q.Dab + q.Dapb : 0.0));
    rx[3] = pb ? q.Tz[3] + 0.5 * (q.Dbbp + (pa ? q.Dabp + q.Dapbp : 0.0)) : 0.0;
}

// The <=4 candidate values of NetMakerOriginal.java:428-452 in the reference's order
// (Cx,Cy), (Cx.nbr,Cy), (Cx,Cy.nbr), (Cx.nbr,Cy.nbr); ok[i] = candidate exists.
FNN_HD void candidate_q(const State& st, const Quad& qd, const double rx[4], double q[4], bool ok[4], double fd[4]) {
    const int32_t ap = st.sap, bp = st.sbp;
    int32_t mm = st.c;
    if (ap >= 0) mm++;
    if (bp >= 0) mm++;
    const double f = (double)mm - 2.0;
    ok[0] = true; ok[1] = ap >= 0; ok[2] = bp >= 0; ok[3] = ap >= 0 && bp >= 0;
    q[0] = q[1] = q[2] = q[3] = 0.0;
    fd[0] = fd[1] = fd[2] = fd[3] = 0.0;
    fd[0] = f * qd.Dab;
    q[0] = fd[0] - rx[0] - rx[2];
    if (ok[1]) { fd[1] = f * qd.Dapb; q[1] = fd[1] - rx[1] - rx[2]; }
    if (ok[2]) { fd[2] = f * qd.Dabp; q[2] = fd[2] - rx[0] - rx[3]; }
    if (ok[3]) { fd[3] = f * qd.Dapbp; q[3] = fd[3] - rx[1] - rx[3]; }
}

// bound on |rxa(z) - ComputeRx(z) as the reference's sequential loop rounds it| (see above; factor 2 of slack)
FNN_HD double rx_bound(const State& st) {
    const double eps = 1.1102230246251565e-16;  // 2^-53
    const double dmax = __builtin_bit_cast(double, st.dmax_bits);
    const double n = (double)st.n;
    return 2.0 * eps * dmax * (n + 64.0) * (2.0 * n + (double)st.n_events + 8.0);
}
// Can the 4-candidate choice be made from the approximate sums rxa?  If every pair of existing
// candidates is further apart than the sum of their bounds, the strict comparisons of :431-451
// come out the same with the exact sums, hence the same (x, y).
FNN_HD bool rx_certify(const State& st, const Quad& qd, const double rxa[4]) {
    if (st.force_exact_rx) return false;
    double q[4], fd[4], bnd[4];
    bool ok[4];
    candidate_q(st, qd, rxa, q, ok, fd);
    const double eps = 1.1102230246251565e-16;
    const double e1 = rx_bound(st);
    const int ia[4] = {0, 1, 0, 1}, ib[4] = {2, 2, 3, 3};
    for (int i = 0; i < 4; i++) {
        const double ra = rxa[ia[i]] < 0.0 ? -rxa[ia[i]] : rxa[ia[i]], rb = rxa[ib[i]] < 0.0 ? -rxa[ib[i]] : rxa[ib[i]];
        const double afd = fd[i] < 0.0 ? -fd[i] : fd[i];
        bnd[i] = 2.0 * e1 + 16.0 * eps * (afd + ra + rb);
        if (!(bnd[i] == bnd[i]) || !(q[i] == q[i]) || !(bnd[i] < 1.7e308)) return false;  // NaN / overflow anywhere: exact path
    }
    for (int i = 0; i < 4; i++)
        for (int j = i + 1; j < 4; j++) {
            if (!ok[i] || !ok[j]) continue;
            const double diff = q[i] < q[j] ? q[j] - q[i] : q[i] - q[j];
            if (!(diff > bnd[i] + bnd[j])) return false;
        }
    return true;
}

// T of the newest cluster's two nodes from the per-workgroup partial sums the update left (tree order)
FNN_HD void t_finalize(const Dev& d) {
    State& st = *d.st;
    if (st.tp_n <= 0) return;
    double tu = 0.0, tv = 0.0;
    for (int32_t g = 0; g < st.tp_n; g++) { tu += d.upart[4 * g + 2]; tv += d.upart[4 * g + 3]; }
    d.T[st.tp_U] = tu;
    d.T[st.tp_U + 1] = tv;
    st.tp_n = 0;
}

// ---------------------------------------------------------------------------
// Relaxed mode: NeighborNetLocal.java:88-264 with additive == false (FastNN.java:329-338), serial branch.
//
// findNodes walks a random permutation of the positions; for the cluster p it lands on it computes the row minimum of
// Q(p, .) over ALL rows (ties kept, in position order), then the row minima of those rows, and stops at the first p
// that is a row minimum of one of its own row minima; one of the mutual pairs is drawn at random.  While
// num_active <= 1024 the base class's full scan is used instead (NetMakerOriginal.java:361-365).
//
// Randomness: the reference draws from ThreadLocalRandom (:30), which cannot be seeded - no two runs of the reference
// agree.  The engine draws from java.util.Random (the generator of the line it replaced, :27) with an explicit seed,
// so that a run is reproducible and equals what the reference computes with `myRandom = new Random(seed)`.
//
// The body below is the control flow, written once: `Env` supplies single-reader loads / single-writer stores (GPU:
// lane 0 of the control wave + a broadcast; the control values are wave-uniform) and the row minimum itself (GPU: the
// whole workgroup; CPU emulation: a loop).  The HashMap foundRowMinimums of one call is the per-slot entry
// {rl_stamp == n_events + 1, rl_val, rl_cnt, rl_list}; a list longer than RL_TIES, or more than RL_MINS mutual pairs,
// is reported as an error (degenerate inputs only: every list holds the two nodes of a cluster at most otherwise).
// ---------------------------------------------------------------------------
constexpr int RL_TIES = 16;
constexpr int RL_MINS = 64;
constexpr int RL_GMAX = 16;                       // workgroups the row pass of one minimum may be spread over
constexpr int RL_REC_WORDS = 2 + RL_TIES;         // a workgroup's record: value, (count | tag), RL_TIES (position | slot)
constexpr int RL_MAIL_WORDS = 32 + RL_GMAX * 32;  // command word (line 0), arrival counter (line 1), records (256 B apart)

struct JavaRandom {  // java.util.Random: next(bits), nextInt(bound)
    uint64_t s;
    FNN_HD static uint64_t scramble(uint64_t seed) { return (seed ^ 0x5DEECE66DULL) & ((1ULL << 48) - 1ULL); }
    FNN_HD int32_t next(int bits) {
        s = (s * 0x5DEECE66DULL + 0xBULL) & ((1ULL << 48) - 1ULL);
        return (int32_t)(s >> (48 - bits));
    }
    FNN_HD int32_t next_int(int32_t bound) {
        int32_t r = next(31);
        const int32_t m = bound - 1;
        if ((bound & m) == 0) r = (int32_t)(((int64_t)bound * (int64_t)r) >> 31);
        else {
            for (int32_t u = r; (int32_t)((uint32_t)u - (uint32_t)(r = u % bound) + (uint32_t)m) < 0; u = next(31)) {}
        }
        return r;
    }
};

// Q(p, q) of NeighborNetLocal.java:104-114 for the nodes in slots ps (partner pp or -1) and qs (partner qp or -1)
FNN_HD double rl_q(const Dev& d, int32_t ps, int32_t pp, int32_t qs, int32_t qp, double cm2, double sxp) {
    const double* Rp = d.D + (int64_t)ps * d.ld;
    double Dpq;
    if (pp < 0 && qp < 0) Dpq = Rp[qs];
    else if (pp >= 0 && qp < 0) Dpq = (Rp[qs] + d.D[(int64_t)pp * d.ld + qs]) / 2.0;
    else if (pp < 0 && qp >= 0) Dpq = (Rp[qs] + Rp[qp]) / 2.0;
    else {
        const double* Rn = d.D + (int64_t)pp * d.ld;
        Dpq = (((Rp[qs] + Rp[qp]) + Rn[qs]) + Rn[qp]) / 4.0;
    }
    return (cm2 * Dpq - sxp) - d.Sx[qs];
}

struct RlRow { int32_t me, cnt; double value; int32_t l0, l1; };  // (l0, l1: the first two rows of the list)

// The control wave pays one global round trip per dependent load, so independent loads are issued together: `Env`
// has load3 / load4 (several words, one round trip) and row(key) (count, value and the first two rows of a cached
// list); a row minimum that was just computed comes back from the workgroup without touching memory.

// findRowMin (:88-157): the cached list of p, else of p.nbr, else computed (and cached under p).  sp / sn: the stamps
// of p's and p.nbr's cache entries (sn = 0 without a partner).
template <class Env>
FNN_HD RlRow rl_find_row_min(const Dev& d, Env& env, int32_t ps, int32_t pp, int32_t sp, int32_t sn, int32_t stamp, int64_t& evals) {
    if (sp == stamp) return env.row(d, ps);
    if (pp >= 0 && sn == stamp) return env.row(d, pp);
    evals += (int64_t)d.st->m - 1 - (pp >= 0 ? 1 : 0);
    return env.rowmin(d, ps, pp, stamp);
}
template <class Env>
FNN_HD int32_t rl_list_at(const Dev& d, Env& env, const RlRow& r, int32_t k) {
    return k == 0 ? r.l0 : (k == 1 ? r.l1 : env.load(&d.rl_list[(int64_t)r.me * RL_TIES + k]));
}

// findNodes (:170-264).  Returns the pair as a candidate record {value, slots of Cx = combineMe.me and Cy = combineMe.row};
// the state of the search (generator, permutation, top) is stored by the caller's leader.
template <class Env>
FNN_HD Cand relaxed_find(const Dev& d, Env& env) {
    State& st = *d.st;
    const int32_t m = st.m, twoP = 2 * st.P;
    JavaRandom rng;
    rng.s = st.rl_rng;
    int32_t top = st.rl_top;
    const int32_t stamp = (int32_t)st.n_events + 1;
    int64_t evals = 0, rows0 = 0;
    int32_t err = 0;
    // (:177-183, the identity permutation and top = ntax - 1 of the first call, are set up with the state: fnn_begin)
    Cand out = cand_none();
    int32_t nmin = 0;
    bool found = false;
    for (int32_t i = top + 1; i > 0; i--) {
        const int32_t swapCell = rng.next_int(i);
        int32_t vsc, vi, vt;  // rowPermutation[swapCell], [i-1], [top]  (0 <= i-1 <= top inside the loop)
        env.load3(&d.rperm[swapCell], &d.rperm[i - 1], &d.rperm[top], vsc, vi, vt);
        if (vsc >= m) {  // a position that is no longer active: drop it from the permutation (:188-198)
            env.store(&d.rperm[swapCell], vt);
            env.store(&d.rperm[top], vsc);
            if (i == top + 1) i--;
            else i++;
            top--;
            continue;
        }
        env.store(&d.rperm[i - 1], vsc);  // swap(rowPermutation, i-1, swapCell) (:199)
        env.store(&d.rperm[swapCell], vi);
        const int32_t ps = env.load(&d.pslot[vsc]);   // p = netNodes[rowPermutation[i-1]]
        const int32_t pp = ps < twoP ? (ps ^ 1) : -1;
        int32_t idp, idn, sp, sn;  // ids and cache stamps of p and p.nbr
        env.load4(&d.sid[ps], &d.sid[pp >= 0 ? pp : ps], &d.rl_stamp[ps], &d.rl_stamp[pp >= 0 ? pp : ps], idp, idn, sp, sn);
        if (pp >= 0 && idn < idp) continue;  // one node per cluster (:201-203)
        const RlRow r1 = rl_find_row_min(d, env, ps, pp, sp, pp >= 0 ? sn : 0, stamp, evals);
        rows0++;
        for (int32_t a = 0; a < r1.cnt; a++) {
            const int32_t other = rl_list_at(d, env, r1, a);
            const int32_t op = other < twoP ? (other ^ 1) : -1;
            int32_t so, son, u0, u1;
            env.load4(&d.rl_stamp[other], &d.rl_stamp[op >= 0 ? op : other], &d.rl_stamp[other], &d.rl_stamp[other], so, son, u0, u1);
            (void)u0; (void)u1;
            const RlRow r2 = rl_find_row_min(d, env, other, op, so, op >= 0 ? son : 0, stamp, evals);
            for (int32_t b = 0; b < r2.cnt; b++) {
                const int32_t row = rl_list_at(d, env, r2, b);
                const int32_t rowp = row < twoP ? (row ^ 1) : -1;
                // testRM.row is p or p.nbr (the four clauses of :210-212)
                if (row == ps || (rowp >= 0 && rowp == ps) || (rowp >= 0 && pp >= 0 && rowp == pp) || (pp >= 0 && row == pp)) {
                    if (nmin < RL_MINS) env.keep(nmin, r2.me, row, r2.value);
                    nmin++;
                    break;
                }
            }
        }
        if (nmin > 0) {
            if (nmin > RL_MINS) { err = 22; break; }
            const int32_t choice = rng.next_int(nmin);
            out = env.kept(d, choice);
            found = true;
            break;  // break outerloop (:258)
        }
    }
    if (!found && !err) err = 21;  // (the reference would reuse the previous event's Cx / Cy)
    if (env.lead()) {
        st.rl_rng = rng.s;
        st.rl_top = top;
        st.rl_evals = evals;
        st.rl_active = 1;
        st.n_rl_rows += rows0;
        if (err && !st.error) st.error = err;
        if (env.err() && !st.error) st.error = env.err();
    }
    return out;
}

// Env of relaxed_find for one thread (CPU emulation): plain loads and stores, the row minimum as the reference's loop
struct RlSerialEnv {
    int32_t e = 0;
    int32_t kme[RL_MINS], krow[RL_MINS];
    double kval[RL_MINS];
    FNN_HD bool lead() const { return true; }
    FNN_HD int32_t err() const { return e; }
    FNN_HD int32_t load(const int32_t* p) const { return *p; }
    FNN_HD double loadd(const double* p) const { return *p; }
    FNN_HD void store(int32_t* p, int32_t v) const { *p = v; }
    FNN_HD void keep(int32_t i, int32_t me, int32_t row, double v) { kme[i] = me; krow[i] = row; kval[i] = v; }
    FNN_HD Cand kept(const Dev& d, int32_t i) const {  // (key: the two positions, as in the scans' records)
        Cand c;
        c.q = kval[i]; c.si = kme[i]; c.sj = krow[i];
        c.key = ((uint64_t)(uint32_t)d.spos[kme[i]] << 32) | (uint64_t)(uint32_t)d.spos[krow[i]];
        return c;
    }
    FNN_HD void load3(const int32_t* pa, const int32_t* pb, const int32_t* pc, int32_t& a, int32_t& b, int32_t& c) const {
        a = *pa; b = *pb; c = *pc;
    }
    FNN_HD void load4(const int32_t* pa, const int32_t* pb, const int32_t* pc, const int32_t* pd, int32_t& a, int32_t& b,
                      int32_t& c, int32_t& dd) const {
        a = *pa; b = *pb; c = *pc; dd = *pd;
    }
    FNN_HD RlRow row(const Dev& d, int32_t key) const {
        RlRow r;
        r.me = key; r.cnt = d.rl_cnt[key]; r.value = d.rl_val[key];
        r.l0 = d.rl_list[(int64_t)key * RL_TIES]; r.l1 = d.rl_list[(int64_t)key * RL_TIES + 1];
        return r;
    }
    FNN_HD RlRow rowmin(const Dev& d, int32_t ps, int32_t pp, int32_t stamp) {  // NeighborNetLocal.java:96-125
        const State& st = *d.st;
        const int32_t m = st.m, twoP = 2 * st.P;
        const double cm2 = (double)st.c - 2.0, sxp = d.Sx[ps];
        double myMin = 1.7976931348623157e308;  // Double.MAX_VALUE
        int32_t cnt = 0;
        for (int32_t row = 0; row < m; row++) {
            const int32_t qs = d.pslot[row];
            if (qs == ps || qs == pp) continue;
            const double q = rl_q(d, ps, pp, qs, qs < twoP ? (qs ^ 1) : -1, cm2, sxp);
            if (q < myMin) { myMin = q; cnt = 0; }
            if (q == myMin) {
                if (cnt < RL_TIES) d.rl_list[(int64_t)ps * RL_TIES + cnt] = qs;
                cnt++;
            }
        }
        if (cnt > RL_TIES) { e = 20; cnt = RL_TIES; }
        d.rl_stamp[ps] = stamp;
        d.rl_cnt[ps] = cnt;
        d.rl_val[ps] = myMin;
        return row(d, ps);
    }
};

// handleAgglomerationEvent: candidate choice (:422-452), bookkeeping of the merge
// (:462-488) and the micro-op plan for the matrix.  rx = {Rx(Cx), Rx(Cx.nbr),
// Rx(Cy), Rx(Cy.nbr)}, 0.0 where the reference leaves the 0.0 initialiser.
// (diagnostic, FNN_TICKS=1 on the GPU: 100 MHz stamps of the plan's stages into Dev::ticks[20 .. 23]; nothing elsewhere)
#if defined(__HIP_DEVICE_COMPILE__)
#define FNN_PLAN_TICK(k) do { if (d.plan_ticks && (threadIdx.x & 63) == 0) { const long long now_ = (long long)wall_clock64(); d.ticks[20 + (k)] += now_ - ptk_; ptk_ = now_; } } while (0)
#define FNN_PLAN_TICK0() long long ptk_ = d.plan_ticks ? (long long)wall_clock64() : 0
#else
#define FNN_PLAN_TICK(k) do { } while (0)
#define FNN_PLAN_TICK0() do { } while (0)
#endif
template <class Tab>
FNN_HD void decide_plan(const Dev& d, Tab& T, const Quad& qd, const double rx[4]) {
    State& st = *d.st;
    Event& cur = st.cur;
    FNN_PLAN_TICK0();
    int32_t a = st.sa, ap = st.sap, b = st.sb, bp = st.sbp;
    double q[4], fd[4];
    bool ok[4];
    candidate_q(st, qd, rx, q, ok, fd);
    int32_t x = a, y = b;
    double best = q[0];
    if (ok[1] && q[1] < best) { x = ap; y = b; best = q[1]; }
    if (ok[2] && q[2] < best) { x = a; y = bp; best = q[2]; }
    if (ok[3] && q[3] < best) { x = ap; y = bp; best = q[3]; }
    int32_t twoP = 2 * st.P;
    int32_t xn = x < twoP ? (x ^ 1) : -1;
    int32_t yn = y < twoP ? (y ^ 1) : -1;
    st.xs = x; st.ys = y;
    FNN_PLAN_TICK(0);  // the candidates' values and the choice
    const int32_t idx = T.sid(x), idy = T.sid(y);
    cur.x_id = idx;
    cur.y_id = idy;
    int32_t m = st.m, P = st.P, nn = st.num_nodes;
    FNN_PLAN_TICK(1);  // the chosen nodes' ids, the counters

    if (xn < 0 && yn < 0) {
        // agg2way (:570-577): both isolated. New two-node cluster goes to slots 2P, 2P+1,
        // smaller id in the even slot.
        cur.kind = KIND_2WAY;
        int32_t lo = idx < idy ? x : y;
        int32_t hi = (lo == x) ? y : x;
        cur.u_id = idx;  // agg2way returns x
        int32_t t0 = 2 * P, t1 = 2 * P + 1;
        if (lo != t0) { swap_slots(d, T, lo, t0); if (hi == t0) hi = lo; }
        if (hi != t1) swap_slots(d, T, hi, t1);
        st.P = P + 1;
        st.c -= 1;
        st.U = t0;  // u = x, and x always has the smaller id (Cx.id < Cy.id, :376-380)
        // slot t0 now holds the smaller id, t1 the larger one
        if ((idx < idy ? idx : idy) != cur.u_id) st.error = 3;
        FNN_PLAN_TICK(2);  // the slot operations
        la_note_cluster(d, st, idx < idy ? idx : idy, idx < idy ? idy : idx);
        FNN_PLAN_TICK(3);
    } else if (xn < 0 || yn < 0) {
        // agg3way(x, y, y.nbr) (:466) or agg3way(y, x, x.nbr) (:476)
        cur.kind = KIND_3WAY;
        int32_t X, Y, Z;
        if (xn < 0) { X = x; Y = y; Z = yn; }
        else { X = y; Y = x; Z = xn; }
        int32_t k = Y >> 1;
        cur.u_id = nn + 1;
        agg3_plan(d, T, X, Y, Z, 2 * k, 2 * k + 1, nn, m);
        if (X != m - 1) move_slot(d, T, m - 1, X);  // close the hole in the singleton region
        st.num_nodes = nn + 2;
        st.m = m - 1;
        st.c -= 1;
        st.U = 2 * k;
        FNN_PLAN_TICK(2);
        la_note_cluster(d, st, nn + 1, nn + 2);
        FNN_PLAN_TICK(3);
    } else {
        if (m == 4) st.error = 4;  // (:474) unreachable: m == 4 with two pairs is the special finish
        // agg4way(x.nbr, x, y, y.nbr) (:484, :707-726): two agg3way calls
        cur.kind = KIND_4WAY;
        int32_t kx = x >> 1, ky = y >> 1;
        int32_t U = 2 * kx, V = 2 * kx + 1;
        agg3_plan(d, T, xn, x, y, U, V, nn, m);          // u1 = (x2,x), v1 = (x,y)
        agg3_plan(d, T, U, V, yn, U, V, nn + 2, m - 1);  // u2 = (u1,v1), v2 = (v1,y2)
        cur.u_id = nn + 3;
        // slots 2ky, 2ky+1 are now empty: refill from the last pair, then shrink the pair
        // region by one and refill its last two slots from the end of the singleton region
        int32_t lastp = P - 1;
        if (ky != lastp) {
            move_slot(d, T, 2 * lastp, 2 * ky);
            move_slot(d, T, 2 * lastp + 1, 2 * ky + 1);
            if (kx == lastp) U = 2 * ky;
        }
        int32_t h0 = 2 * lastp, h1 = 2 * lastp + 1;
        int32_t S = m - 2 * P;
        if (S >= 1) move_slot(d, T, m - 1, h0);
        if (S >= 2) move_slot(d, T, m - 2, h1);
        st.P = P - 1;
        st.num_nodes = nn + 4;
        st.m = m - 2;
        st.c -= 1;
        st.U = U;
        FNN_PLAN_TICK(2);
        la_note_cluster(d, st, nn + 3, nn + 4);
        FNN_PLAN_TICK(3);
    }
}

// ---------------------------------------------------------------------------
// The <= MAX_S slots involved in an event go through the reference's own per-node bodies
// (subtractClusterDistance, the micro-ops, updateClusterDistances) one after the other.  Every
// entry they read or write has BOTH indices in the involved set S (checked case by case in
// DESIGN.md "k_update"), so they work on the S x S block of the matrix.  The bodies are written
// against an accessor: GlobalAcc addresses the matrix in memory; BlockAcc a copy of the block
// (the GPU keeps it in LDS: ~10 dependent phases at LDS instead of global-memory latency).
// ---------------------------------------------------------------------------
struct GlobalAcc {
    const Dev& d;
    FNN_HD double get(int32_t r, int32_t c) const { return d.D[(int64_t)r * d.ld + c]; }
    FNN_HD void put(int32_t r, int32_t c, double v) const { store_d(d, r, c, v); }
    FNN_HD double sx(int32_t s) const { return d.Sx[s]; }
    FNN_HD void set_sx(int32_t s, double v) const { d.Sx[s] = v; }
    FNN_HD double t(int32_t s) const { return d.T[s]; }
    FNN_HD void set_t(int32_t s, double v) const { d.T[s] = v; }
};
struct BlockAcc {
    double* blk;       // [MAX_S][MAX_S], entry (i, j) = D[S[i]][S[j]]
    double* sxl;       // [MAX_S] row sums (NetNode.Sx)
    double* tl;        // [MAX_S] approximate weighted row sums (Dev::T)
    const int32_t* S;  // the involved slots
    int32_t nS;
    int32_t* err;      // set to 13 if a slot outside S is addressed (never expected)
    FNN_HD int ix(int32_t s) const {
#pragma unroll
        for (int i = 0; i < MAX_S; i++)
            if (i < nS && S[i] == s) return i;
        *err = 13;
        return 0;
    }
    FNN_HD double get(int32_t r, int32_t c) const { return blk[ix(r) * MAX_S + ix(c)]; }
    FNN_HD void put(int32_t r, int32_t c, double v) const { blk[ix(r) * MAX_S + ix(c)] = v; }
    FNN_HD double sx(int32_t s) const { return sxl[ix(s)]; }
    FNN_HD void set_sx(int32_t s, double v) const { sxl[ix(s)] = v; }
    FNN_HD double t(int32_t s) const { return tl[ix(s)]; }
    FNN_HD void set_t(int32_t s, double v) const { tl[ix(s)] = v; }
};

// subtractClusterDistance(p, x); subtractClusterDistance(p, y) for p = node in slot s
// (NetMakerOriginal.java:455-461, 681-696).  Old layout.
template <class Acc>
FNN_HD void subtract_thread(const Acc& A, const State& st, int32_t s) {
    if (s >= st.m_old) return;
    if (s == st.xs || s == st.ys) return;  // i != x.positionID && i != y.positionID
    int32_t twoP = 2 * st.P_old;
    bool sp = s < twoP;
    if (sp && (s & 1)) return;  // not the representative
    double sx = A.sx(s);
    double sxn = sp ? A.sx(s ^ 1) : 0.0;
    double told0 = 0.0, told1 = 0.0;  // what the merging nodes contributed to T of node s (and of its partner)
    bool bystander = true;
    for (int k = 0; k < 2; k++) {
        int32_t t = k == 0 ? st.xs : st.ys;
        int32_t tn = t < twoP ? (t ^ 1) : -1;
        if (s == t || s == tn) { bystander = false; continue; }
        double v;
        if (!sp && tn < 0) { const double e = A.get(t, s); v = e; told0 += e; }
        else if (sp && tn < 0) { const double e0 = A.get(t, s), e1 = A.get(t, s ^ 1); v = (e0 + e1) / 2.0; told0 += e0; told1 += e1; }
        else if (!sp && tn >= 0) { const double e0 = A.get(t, s), f0 = A.get(tn, s); v = (e0 + f0) / 2.0; told0 += 0.5 * (e0 + f0); }
        else {
            const double e0 = A.get(t, s), f0 = A.get(tn, s), e1 = A.get(t, s ^ 1), f1 = A.get(tn, s ^ 1);
            v = (((e0 + f0) + e1) + f1) / 4.0;
            told0 += 0.5 * (e0 + f0); told1 += 0.5 * (e1 + f1);
        }
        sx -= v;   // p.Sx -= Dpx
        sxn -= v;  // p.nbr.Sx -= Dpx
    }
    A.set_sx(s, sx);
    if (sp) A.set_sx(s ^ 1, sxn);
    if (bystander) {  // (a node of a merging cluster either disappears or gets a fresh T)
        A.set_t(s, A.t(s) - told0);
        if (sp) A.set_t(s ^ 1, A.t(s ^ 1) - told1);
    }
}

// Bulk + special parts of one micro-op for slot k
template <class Acc>
FNN_HD void op_thread(const Acc& A, const Op& op, int32_t k) {
    if (k >= op.mcur) return;
    if (op.kind == OP_SWAP) {
        int32_t a = op.a, b = op.b;
        if (k == a) {
            double t = A.sx(a); A.set_sx(a, A.sx(b)); A.set_sx(b, t);
            t = A.t(a); A.set_t(a, A.t(b)); A.set_t(b, t);
        } else if (k != b) {
            double ta = A.get(a, k), tb = A.get(b, k);
            A.put(a, k, tb); A.put(k, a, tb);
            A.put(b, k, ta); A.put(k, b, ta);
        }
    } else if (op.kind == OP_MOVE) {
        int32_t src = op.a, dst = op.b;
        if (k == src) {
            A.put(dst, dst, 0.0);
            A.set_sx(dst, A.sx(src));
            A.set_t(dst, A.t(src));
        } else if (k != dst) {
            double t = A.get(src, k);
            A.put(dst, k, t); A.put(k, dst, t);
        }
    } else if (op.kind == OP_AGG3) {
        int32_t X = op.a, Y = op.b, Z = op.c, U = op.d, V = op.e;
        if (k == X) {
            // the aliased entry D[u][v] and the diagonal (NetMakerOriginal.java:653-656, 670):
            // the in-place loop writes D[u][v] twice, once at p = u and once at p = v, the
            // second write reading the first
            double dxz = A.get(X, Z), dyx = A.get(Y, X), dyz = A.get(Y, Z);
            double uv;
            if (op.flag) uv = (2.0 / 3.0) * ((2.0 / 3.0) * dxz + div3(dyx)) + div3(dyz);
            else uv = (2.0 / 3.0) * ((2.0 / 3.0) * dxz + div3(dyz)) + div3(dyx);
            A.put(U, U, 0.0); A.put(V, V, 0.0);
            A.put(U, V, uv); A.put(V, U, uv);
        } else if (k != Y && k != Z) {
            double dx = A.get(X, k), dy = A.get(Y, k), dz = A.get(Z, k);
            const double dy3 = div3(dy);
            double nu = (2.0 / 3.0) * dx + dy3;
            double nv = (2.0 / 3.0) * dz + dy3;
            A.put(U, k, nu); A.put(k, U, nu);
            A.put(V, k, nv); A.put(k, V, nv);
        }
    }
}

// updateClusterDistances(u), per-node part (NetMakerOriginal.java:520-533). New layout.
// (the addend of the new cluster's sequential row sum goes to the chain buffer in memory either way)
// tuv[0..1] receive this node's (and its partner's) terms of T of the new cluster's two nodes u, v.
template <class Acc>
FNN_HD double add_thread(const Acc& A, const Dev& d, const State& st, int32_t s, double tuv[2]) {
    tuv[0] = tuv[1] = 0.0;
    if (s >= st.m) return 0.0;
    int32_t twoP = 2 * st.P;
    int32_t U = st.U, V = st.U + 1;
    double val = 0.0;
    bool sp = s < twoP;
    bool rep = !sp || !(s & 1);
    if (rep && s != U) {
        double dpu;
        const double u0 = A.get(U, s), v0 = A.get(V, s);
        if (!sp) {
            dpu = (u0 + v0) / 2.0;
            A.set_t(s, A.t(s) + 0.5 * (u0 + v0));
            tuv[0] = u0; tuv[1] = v0;
        } else {
            const double u1 = A.get(U, s + 1), v1 = A.get(V, s + 1);
            dpu = (((u0 + v0) + u1) + v1) / 4.0;
            A.set_t(s, A.t(s) + 0.5 * (u0 + v0));
            A.set_t(s + 1, A.t(s + 1) + 0.5 * (u1 + v1));
            tuv[0] = 0.5 * (u0 + u1); tuv[1] = 0.5 * (v0 + v1);
        }
        A.set_sx(s, A.sx(s) + dpu);
        if (sp) A.set_sx(s + 1, A.sx(s + 1) + dpu);
        val = dpu;
    } else if (s == U) {  // the partner's weight 1/2 in each other's sum
        const double uv = A.get(U, V);
        tuv[0] = 0.5 * uv; tuv[1] = 0.5 * uv;
    }
    d.chain[chain_addr(d.spos[s])] = val;  // adding +0.0 to a running sum that starts at +0.0 changes no bit
    return val;  // (this node's addend of the new cluster's row sum)
}

// The part of the control block the column threads of the update read, BY VALUE: on the GPU every field
// is wave-uniform and sits in scalar registers (fetched once; a `const State&` in memory costs one
// dependent round trip of ~1.7 us each time the compiler reaches a field it has not loaded yet).
struct PlanView {
    int32_t m_old, P_old, ev_finish, nS, ntgt, tU, tV;
    int32_t S[MAX_S];
    // the rows a column thread reads are rows of involved slots: everything below is an INDEX into S
    int32_t ix, ixn, iy, iyn;  // x, x.nbr, y, y.nbr (the merging nodes; -1: no partner)
    int32_t tdst[MAX_TGT], tkind[MAX_TGT], ta[MAX_TGT], tb[MAX_TGT], tc[MAX_TGT], td[MAX_TGT];  // the recipes (Tgt), sources as indices
};
template <class Src, class Uni>
FNN_HD PlanView plan_view(const Src& st, Uni uni) {  // uni(x): x as a wave-uniform value; Src: State, or the plan message of the fused event kernel
    PlanView v;
    v.m_old = uni(st.m_old); v.P_old = uni(st.P_old); v.ev_finish = uni(st.ev_finish);
    v.nS = uni(st.nS); v.ntgt = uni(st.ntgt); v.tU = uni(st.tU); v.tV = uni(st.tV);
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int i = 0; i < MAX_S; i++) v.S[i] = uni(st.S[i]);
    auto idx = [&](int32_t slot) {
        int32_t r = -1;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
        for (int i = 0; i < MAX_S; i++) if (i < v.nS && v.S[i] == slot && slot >= 0) r = i;
        return r;
    };
    const int32_t xs = uni(st.xs), ys = uni(st.ys), twoP = 2 * v.P_old;
    v.ix = idx(xs); v.ixn = xs < twoP ? idx(xs ^ 1) : -1;
    v.iy = idx(ys); v.iyn = ys < twoP ? idx(ys ^ 1) : -1;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int i = 0; i < MAX_TGT; i++) {
        v.tdst[i] = uni(st.tgt[i].dst); v.tkind[i] = uni(st.tgt[i].kind);
        v.ta[i] = idx(uni(st.tgt[i].a)); v.tb[i] = idx(uni(st.tgt[i].b));
        v.tc[i] = idx(uni(st.tgt[i].c)); v.td[i] = idx(uni(st.tgt[i].d));
    }
    return v;
}
struct UniId { FNN_HD int32_t operator()(int32_t x) const { return x; } };

// the value of recipe t at the column whose involved-row entries R(i) returns (i: index into S)
template <class RowVal>
FNN_HD double tgt_eval(const PlanView& st, int t, RowVal&& R) {
    const int32_t kind = st.tkind[t];
    if (kind == T_COPY) return R(st.ta[t]);
    if (kind == T_L1) return (2.0 / 3.0) * R(st.ta[t]) + div3(R(st.tb[t]));
    const double b3 = div3(R(st.tb[t]));
    const double v1 = (2.0 / 3.0) * R(st.tc[t]) + b3;
    if (kind == T_L2U) {
        const double u1 = (2.0 / 3.0) * R(st.ta[t]) + b3;
        return (2.0 / 3.0) * u1 + div3(v1);
    }
    return (2.0 / 3.0) * R(st.td[t]) + div3(v1);  // T_L2V
}

// Fused per-event update for a cluster whose slot(s) k (and k+1) are NOT involved in the
// event: subtract_thread + every op_thread + add_thread for these columns in one pass.
// Reads only rows of involved slots at its own column(s) and writes only entries with
// exactly one index equal to its own column(s), so it cannot conflict with any other thread;
// the involved slots themselves are handled by update_special_*.
// R(i, c): the matrix entry D[S[i]][k + c] BEFORE the event (c = 0, 1).  The CPU emulation reads memory;
// the GPU fetches the <= 8 (x 2) entries of its column up front in ONE round trip and serves R from LDS.
// tuv[0..1] receive this cluster's terms of T of the new cluster's two nodes.
template <class RowVal>
FNN_HD double update_bulk(const Dev& d, const PlanView& st, int32_t k, bool paired, double sx0, double sx1, double t0_old,
                          double t1_old, int32_t pos0, int32_t pos1, double tuv[2], RowVal&& R) {
    double told0 = 0.0, told1 = 0.0;  // what the merging nodes contributed to T of node k (k + 1)
    if (!st.ev_finish) {
        // subtractClusterDistance(p, x); subtractClusterDistance(p, y) (:455-461, 681-696)
        for (int q = 0; q < 2; q++) {
            const int32_t it = q == 0 ? st.ix : st.iy;
            const int32_t itn = q == 0 ? st.ixn : st.iyn;
            double v;
            if (!paired && itn < 0) { const double e = R(it, 0); v = e; told0 += e; }
            else if (paired && itn < 0) { const double e0 = R(it, 0), e1 = R(it, 1); v = (e0 + e1) / 2.0; told0 += e0; told1 += e1; }
            else if (!paired && itn >= 0) { const double e0 = R(it, 0), f0 = R(itn, 0); v = (e0 + f0) / 2.0; told0 += 0.5 * (e0 + f0); }
            else {
                const double e0 = R(it, 0), f0 = R(itn, 0), e1 = R(it, 1), f1 = R(itn, 1);
                v = (((e0 + f0) + e1) + f1) / 4.0;
                told0 += 0.5 * (e0 + f0); told1 += 0.5 * (e1 + f1);
            }
            sx0 -= v;
            sx1 -= v;
        }
    }
    // all new values first (a changed row may be the source of another), then the stores
    double tv[MAX_TGT][2];
    double u0 = 0.0, u1 = 0.0, v0 = 0.0, v1 = 0.0;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int t = 0; t < MAX_TGT; t++) {
        tv[t][0] = 0.0; tv[t][1] = 0.0;
        if (t < st.ntgt) {
            tv[t][0] = tgt_eval(st, t, [&](int32_t i) { return R(i, 0); });
            if (paired) tv[t][1] = tgt_eval(st, t, [&](int32_t i) { return R(i, 1); });
            if (t == st.tU) { u0 = tv[t][0]; u1 = tv[t][1]; }
            if (t == st.tV) { v0 = tv[t][0]; v1 = tv[t][1]; }
        }
    }
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int t = 0; t < MAX_TGT; t++) {
        if (t < st.ntgt) {
            // the entry and its mirror; of the two only the one at or below the diagonal has a bf16 copy (k is not
            // an involved slot, so k != dst): one conversion, one 2-byte store
            const int32_t dst = st.tdst[t];
            const int64_t rb = (int64_t)dst * d.ld, rk = (int64_t)k * d.ld;
            d.D[rb + k] = tv[t][0];
            d.D[rk + dst] = tv[t][0];
            if (d.H) d.H[k < dst ? (int64_t)dst * d.ldh + k : (int64_t)k * d.ldh + dst] = bf16_from_double(tv[t][0]);
            if (paired) {
                d.D[rb + k + 1] = tv[t][1];
                d.D[rk + d.ld + dst] = tv[t][1];
                if (d.H) d.H[k + 1 < dst ? (int64_t)dst * d.ldh + k + 1 : (int64_t)(k + 1) * d.ldh + dst] = bf16_from_double(tv[t][1]);
            }
        }
    }
    tuv[0] = tuv[1] = 0.0;
    if (!st.ev_finish) {
        // updateClusterDistances, per-node part (:520-531)
        double dpu;
        if (!paired) dpu = (u0 + v0) / 2.0;
        else dpu = (((u0 + v0) + u1) + v1) / 4.0;
        d.Sx[k] = sx0 + dpu;
        d.chain[chain_addr(pos0)] = dpu;
        // approximate weighted row sums: the merged nodes go, the new cluster's two nodes come (weight 1/2 each)
        d.T[k] = (t0_old - told0) + 0.5 * (u0 + v0);
        if (paired) {
            d.Sx[k + 1] = sx1 + dpu;
            d.chain[chain_addr(pos1)] = 0.0;
            d.T[k + 1] = (t1_old - told1) + 0.5 * (u1 + v1);
            tuv[0] = 0.5 * (u0 + u1); tuv[1] = 0.5 * (v0 + v1);
        } else { tuv[0] = u0; tuv[1] = v0; }
        return dpu;  // (this cluster's addend of the new cluster's row sum)
    }
    return 0.0;
}
// which column thread works: returns false for threads without a cluster of their own; `paired`: two columns
FNN_HD bool bulk_active(const PlanView& st, int32_t k, bool& paired) {
    paired = false;
    if (k >= st.m_old) return false;
    paired = k < 2 * st.P_old;
    if (paired && (k & 1)) return false;  // the even thread of a two-node cluster does both columns
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int i = 0; i < MAX_S; i++) if (i < st.nS && st.S[i] == k) return false;
    return true;
}
// the whole column thread with everything read straight from memory (CPU emulation)
FNN_HD double update_bulk_mem(const Dev& d, const PlanView& st, int32_t k, double tuv[2]) {
    tuv[0] = tuv[1] = 0.0;
    bool paired;
    if (!bulk_active(st, k, paired)) return 0.0;
    const double* D = d.D; const int64_t ld = d.ld;
    const int32_t* S = st.S;
    return update_bulk(d, st, k, paired, d.Sx[k], paired ? d.Sx[k + 1] : 0.0, d.T[k], paired ? d.T[k + 1] : 0.0, d.spos[k],
                       paired ? d.spos[k + 1] : 0, tuv, [&](int32_t i, int c) { return D[(int64_t)S[i] * ld + k + c]; });
}

// The involved slots (<= MAX_S columns) go through the per-column bodies above in phases:
// phase 0 subtract, phases 1..nops one micro-op each, last phase add.  Within a phase the
// columns are independent; phases are separated by a workgroup barrier on the GPU.
FNN_HD int32_t update_special_phases(const State& st) { return st.nops + 2; }
template <class Acc>
FNN_HD double update_special_acc(const Acc& A, const Dev& d, int32_t phase, int32_t i, double tuv[2]) {
    const State& st = *d.st;
    tuv[0] = tuv[1] = 0.0;
    if (i >= st.nS) return 0.0;
    const int32_t k = st.S[i];
    if (phase == 0) { if (!st.ev_finish) subtract_thread(A, st, k); }
    else if (phase <= st.nops) { Op op = st.ops[phase - 1]; op_thread(A, op, k); }
    else { if (!st.ev_finish) return add_thread(A, d, st, k, tuv); }
    return 0.0;
}
FNN_HD double update_special(const Dev& d, int32_t phase, int32_t i, double tuv[2]) { return update_special_acc(GlobalAcc{d}, d, phase, i, tuv); }
// the block form: copy the S x S block and the row sums of S out of memory ... (phases on the copy) ... and back
FNN_HD void special_block_load(const Dev& d, double* blk, double* sxl, double* tl, int32_t e) {  // e in [0, MAX_S * MAX_S)
    const State& st = *d.st;
    const int32_t i = e / MAX_S, j = e % MAX_S;
    if (i < st.nS && j < st.nS) blk[e] = d.D[(int64_t)st.S[i] * d.ld + st.S[j]];
    if (j == 0 && i < st.nS) sxl[i] = d.Sx[st.S[i]];
    if (j == 1 && i < st.nS) tl[i] = d.T[st.S[i]];
}
FNN_HD void special_block_store(const Dev& d, const double* blk, const double* sxl, const double* tl, int32_t e) {
    const State& st = *d.st;
    const int32_t i = e / MAX_S, j = e % MAX_S;
    if (i < st.nS && j < st.nS) store_d(d, st.S[i], st.S[j], blk[e]);
    if (j == 0 && i < st.nS) d.Sx[st.S[i]] = sxl[i];
    if (j == 1 && i < st.nS) d.T[st.S[i]] = tl[i];
}

// u.Sx = sequential sum; u.nbr.Sx = u.Sx (:518-519, 532, 535); close the event
// everything that closes an event except the new cluster's exact row sum (computed beside the next
// event's tracking, see k_track)
FNN_HD void close_event(const Dev& d);
FNN_HD void finalize(const Dev& d, double usx) {
    State& st = *d.st;
    if (!st.ev_finish) {
        d.Sx[st.U] = usx;
        d.Sx[st.U + 1] = usx;
    } else {  // special finish: u, v are fresh nodes whose Sx keeps its default (NetNode.java:15)
        d.Sx[st.U] = 0.0;
        d.Sx[st.U + 1] = 0.0;
    }
    close_event(d);
}
FNN_HD void close_event(const Dev& d) {
    State& st = *d.st;
    if (st.record_events) d.evlog[st.n_events] = st.cur;
    if (st.la_valid) st.la_k += 1;
    if (!st.ev_finish) { st.la_mprev = st.cur.best; st.la_have_mprev = 1; }
    st.n_events += 1;
    st.sum_entries += st.cur.entries;
    if (st.ev_finish || st.m <= 3) st.done = 1;
}

// NetMakerOriginal.initialize (:164-191) for the all-singleton start: node k receives
// D[0][k], ..., D[k-1][k] (as q of the outer nodes p < k) and then D[k][k+1..n-1] (as p),
// i.e. one sequential sum over j != k in ascending j.  Also sets up the identity layout.
FNN_HD void init_thread(const Dev& d, int32_t k) {
    if (k >= d.n) return;
    const double* D = d.D; const int64_t ld = d.ld;
    double s = 0.0;
    // == D[k][j] (symmetric input); the column walk coalesces across k.  Eight loads in flight, added in
    // the reference's order (the sum is sequential, the loads are not)
    for (int32_t j0 = 0; j0 < d.n; j0 += 8) {
        double v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) v[u] = (j0 + u < d.n) ? D[(int64_t)(j0 + u) * ld + k] : 0.0;
#pragma unroll
        for (int u = 0; u < 8; u++)
            if (j0 + u < d.n && j0 + u != k) s += v[u];
    }
    d.Sx[k] = s;
    d.T[k] = s;  // all singletons: the weighted row sum is the row sum
    d.sid[k] = k + 1;
    d.spos[k] = k;
    d.pslot[k] = k;
    if (d.islot) d.islot[k + 1] = k;
    if (k == 0) {
        // the screening bound needs a finite, float-representable bound on |D| (prep kernel)
        const double dmax = __builtin_bit_cast(double, d.st->dmax_bits);
        d.st->screen_ok = (d.H != nullptr && dmax == dmax && dmax < 1e37) ? 1 : 0;
        d.st->la_W = 16.0 * dmax;  // first lookahead window width; adapted at every base scan (la_close_base)
    }
}

// SplitMix64, k-th output for a given seed
FNN_HD uint64_t splitmix64_at(uint64_t seed, uint64_t k) {
    uint64_t z = seed + (k + 1) * 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

// packed strict upper triangle (DistancesAndNames.java:24-38): entry (a, b), a < b, sits at
// packed_row_base(n, a) + b
FNN_HD int64_t packed_row_base(int64_t n, int64_t a) { return a * (n - 1) - a * (a - 1) / 2 - (a + 1); }

// synthetic entry (i < j) of the generator; index k = i*n - i(i+1)/2 + (j-i-1)
FNN_HD double synth_entry(int64_t n, int64_t i, int64_t j, uint64_t seed, int32_t dist) {
    uint64_t k = (uint64_t)(i * n - i * (i + 1) / 2 + (j - i - 1));
    double u = (double)(splitmix64_at(seed, k) >> 11) * 0x1.0p-53;
    if (dist == 1) return (double)((int64_t)(u * 1e4) + 1) / 1e4;
    return u + 0x1.0p-10;
}

}  // namespace fnn
#endif
