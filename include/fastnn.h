/*
 * fastnn.h -- C ABI of libfastnn_hip.so, the MI355X-native Canonical Neighbor-Net
 * agglomeration engine.
 *
 * The reference (JacobPorter/FastNeighborNet, Java) has no FFI layer; its seam for
 * this path is the abstract class NetMakerOriginal:
 *     ctor  (double[][] d, int numTaxa, int numThreads, ExecutorService pool)
 *                                                 NetMakerOriginal.java:51-56
 *     public int[] runNeighborNet()               NetMakerOriginal.java:129-162
 * as instantiated by FastNN.main for -mode Canonical (FastNN.java:324-328) and
 * called once (FastNN.java:378 / :391).  The entry points below are exactly what a
 * JNI binding of that seam needs (see INTEGRATION.md for the Java stub); plain
 * pointers and sizes only.
 *
 * All arithmetic on the path is IEEE binary64, one rounding per source-level
 * operation, no FMA contraction, in the reference's evaluation order, so the
 * circular order is bit-identical to the Java `-threads 1` result.
 *
 * Thread-safety: a handle is single-caller; different handles may be used from
 * different threads.  Functions return FNN_OK (0) or a negative fnn_status;
 * fnn_last_error() gives a thread-local message.  The library never aborts the
 * process and has NO CPU fallback: without a usable HIP device every call that
 * needs one fails with FNN_EHIP.
 */
#ifndef FASTNN_H
#define FASTNN_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FASTNN_ABI_VERSION 2  /* 2: fnn_sw_stats grew (route, certificate, give-up reason), FNN_ECAPACITY / FNN_EINEXACT */

typedef enum fnn_status {
    FNN_OK      = 0,
    FNN_EINVAL  = -1,  /* bad argument / matrix not symmetric, not finite, non-zero diagonal */
    FNN_ENOMEM  = -2,  /* host or device allocation failed */
    FNN_EHIP    = -3,  /* HIP runtime error or no device */
    FNN_ERCCL   = -4,  /* collective error (several GPUs: RCCL or the host callback) */
    FNN_ESTATE  = -5,  /* call sequence error (e.g. run before the matrix is set) */
    FNN_ECAPACITY = -6,/* split weights: the optimum has more positive splits than the dense factor of the block method
                          holds (nearly circular distances: O(n^2) splits) and the fall-back - CircularSplitWeights.java's
                          conjugate-gradient route - is not affordable at this size; nothing is returned
                          (FNN_SW_ALLOW_REFERENCE_ROUTE=1 takes that route anyway) */
    FNN_EINEXACT = -7  /* split weights: weights_out IS filled, but the solver's own Kuhn-Tucker check of them exceeds
                          1e-9 of max|A^T d| (fnn_sw_stats.kkt_violation says by how much): not the certified optimum */
} fnn_status;

/* Kind of agglomeration event (NetMakerOriginal.java:462-488; special finish :343-360). */
enum { FNN_KIND_2WAY = 2, FNN_KIND_3WAY = 3, FNN_KIND_4WAY = 4, FNN_KIND_FINISH = 5 };

/* Options.  Zero-initialise, then set what you need. */
typedef struct fnn_opts {
    int32_t device;        /* HIP device ordinal (default 0) */
    int32_t validate;      /* 1: check symmetry / zero diagonal / finiteness on device before running */
    int32_t record_events; /* 1: keep the per-event trajectory (fnn_get_events) */
    int32_t force_exact_rx;/* diagnostic: 1 = always evaluate the ComputeRx sums with the exact
                              sequential-sum kernel instead of certifying the 4-candidate choice
                              from tree sums (same result; exercises the rare path) */
    int32_t disable_screen;/* 1 = always scan the fp64 matrix in full; default (0): from 4096 taxa on, events
                              with >= min(2048, n / 4) live nodes (not below 512) first stream a bf16 copy of the
                              matrix (2 bytes per entry, a quarter of the fp64 bytes) to find, within a rigorous
                              error bound, the few 32 x 512 tile units that can hold the minimum; only those are
                              rescanned in fp64 (same result) */
    int32_t lookahead;     /* events one screening pass may serve ("lookahead window"): 0 = default
                              (min(16 + n / 1024, 64) at most, and min(that, 16 + m / 512) for a window opened
                              with m live nodes: 48 at n = 32768), < 0 = off (every event scans), > 0 = that
                              many (capped at 512); same result either way */
    int32_t lookahead_pairs;/* wanted number of tracked pairs per window (0 = default 49152; the list holds 65536) */
    int32_t mode;          /* FNN_MODE_CANONICAL (0, default) or FNN_MODE_RELAXED: `-mode Relaxed` without `-additive`
                              (FastNN.java:329-338, NeighborNetLocal.java:170-264) - while more than 1024 nodes are
                              active the pair to merge is found by the randomised search for mutual row minima instead
                              of the full scan; everything else (4-candidate choice, merges, expansion) is the same */
    uint32_t relaxed_seed_lo, relaxed_seed_hi; /* seed of the Relaxed mode's generator: java.util.Random(seed) (the
                              reference draws from ThreadLocalRandom, NeighborNetLocal.java:30, which cannot be seeded;
                              with java.util.Random - the generator of the line it replaced, :27 - a run is reproducible) */
    int32_t relaxed_min_active; /* tests: the relaxed search runs while num_active > this (0 = the reference's 1024,
                              NetMakerOriginal.java:361) */
    int32_t reserved[5];
} fnn_opts;

enum { FNN_MODE_CANONICAL = 0, FNN_MODE_RELAXED = 1 };

/* One agglomeration event == one iteration of the loop of
 * NetMakerOriginal.agglomNodes (:339-393).  Same fields as the test oracle's
 * nno_event so trajectories can be compared field by field. */
typedef struct fnn_event {
    int32_t m_before;      /* num_active at loop entry */
    int32_t c_before;      /* num_clusters at loop entry */
    int32_t cx_id, cy_id;  /* Cx.id, Cy.id after the id swap (:376-380); 0 for FINISH */
    int32_t x_id, y_id;    /* nodes chosen among the <=4 candidates (:428-452) */
    int32_t kind;          /* FNN_KIND_* */
    int32_t u_id;          /* id of the node returned by agg2way/agg3way/agg4way */
    double  best;          /* scan minimum Qpq (NeighborNetCanonical.java:170-177) */
    int64_t entries;       /* E_t = m(m-1)/2 - (m-c): matrix entries the scan must read */
} fnn_event;

typedef struct fnn_stats {
    int64_t n_events;        /* agglomeration events executed */
    int64_t sum_entries;     /* sum_t E_t  (algorithmic entries; x8 = algorithmic bytes) */
    double  t_init_s;        /* initial row sums (NetMakerOriginal.initialize :164-191) */
    double  t_agglom_s;      /* agglomNodes loop, device time incl. launches */
    double  t_expand_s;      /* expandNodes on the host (:246-325) */
    double  t_total_s;       /* matrix resident on device -> order on host */
    double  t_scan_s;        /* sum of the durations of the timed k_screen launches (HIP events; 0 unless timing
                                is enabled): all of them without lookahead windows, the scheduled base scans with */
    int64_t scan_launches;   /* number of those launches */
    int64_t scan_bytes;      /* matrix bytes those launches had to stream: E_t entries at 2 B (bf16 screening
                                copy) per launch, plus the fp64 rescans of the candidate units */
    int64_t n_rx_certified;  /* events whose 4-candidate choice was certified from approximate row sums */
    int64_t n_rx_exact;      /* events that needed the exact sequential ComputeRx sums */
    int64_t n_screen_events; /* events whose scan went through the bf16 screening pass */
    int64_t n_rescan_units;  /* 32 x 512 units rescanned in fp64 over those events */
    int64_t n_base_scans;    /* events that ran a scan (all of them without lookahead windows) */
    int64_t n_window_hits;   /* events whose minimum came from an open lookahead window (no scan) */
    int64_t n_window_fails;  /* events whose window could not certify the minimum (they rescanned) */
    int64_t window_pairs;    /* tracked pairs summed over all windows */
    int64_t bytes_total;     /* matrix bytes read by ALL scan work of the run (timed or not, window items too) */
    int64_t n_handover_retries; /* window events whose per-workgroup records failed their check word on the first read
                                (the hand-over inside k_track is checked: fence + second read, then the event scans);
                                expected 0 */
    int64_t n_sweeps_exact;  /* ... whose sweep of the newest cluster's rows had to wait for its exact row sum */
    double  t_plain_s;       /* sum of the durations of the plain fp64 scan launches (k_scan; m below the screening threshold) */
    int64_t plain_launches;  /* number of those launches */
    int64_t plain_bytes;     /* 8 * E_t summed over them */
    int64_t n_stalled_events;/* launch sequences without scan kernels that found their window gone (they do nothing; the
                                host relaunches with a scan at its next look at the state) */
    int64_t n_relaxed_events;/* Relaxed mode: events whose pair came from the search for mutual row minima (the others scanned) */
} fnn_stats;

typedef struct fnn_handle fnn_handle;

/* Library / device probing. */
int32_t     fnn_abi_version(void);
const char* fnn_last_error(void);
int32_t     fnn_device_count(void);                 /* <0 on error */

/* Handle lifecycle.  n = number of taxa (ntax of NetMakerOriginal.java:52). */
int32_t fnn_create(int32_t n, const fnn_opts* opts, fnn_handle** out);
int32_t fnn_destroy(fnn_handle* h);

/* Matrix upload: rows [row0, row0+nrows) of the symmetric n x n fp64 matrix, host
 * memory, row stride ld_in doubles.  Plays the role of the `double[][] d`
 * constructor argument; the caller's buffer is never written (the reference
 * destroys its argument in place, NetMakerOriginal.java:653-656). */
int32_t fnn_set_rows(fnn_handle* h, int32_t row0, int32_t nrows, const double* rows, int64_t ld_in);
/* The same matrix from the reference's own container: the packed strict upper triangle of
 * DistancesAndNames (`double[] distances`, n(n-1)/2 entries, row-major: entry (a, b), a < b, at
 * a(n-1) - a(a-1)/2 + b - (a+1), DistancesAndNames.java:24-38).  Replaces the dense expansion of
 * FastNN.java:307-312: half the bytes cross the bus, the device mirrors the triangle itself. */
int32_t fnn_set_packed_upper(fnn_handle* h, const double* packed);
/* Same from DEVICE memory (whole matrix, row stride ld_in doubles).  The copy runs on the engine's own (non-blocking)
 * stream, which is NOT ordered against the stream that produced d_matrix: synchronise the producer (stream or device)
 * before this call.  The call returns after the copy has completed. */
int32_t fnn_set_matrix_device(fnn_handle* h, const double* d_matrix, int64_t ld_in);
/* Fill the device matrix with the synthetic generator of SURVEY.md 8(d)
 * (SplitMix64; dist 0 = uniform53, 1 = dec4), bit-identical to the host generator. */
int32_t fnn_synth(fnn_handle* h, uint64_t seed, int32_t dist);

/* The matrix as it is resident on the device (after an upload or fnn_synth, before the run consumes it), to host memory with
 * row stride ld_out doubles: what a caller needs who generated the distances on the device and wants them for the split
 * weights or the Nexus document afterwards (the reference re-parses its file for that, FastNN.java:438). */
int32_t fnn_get_matrix(fnn_handle* h, double* out, int64_t ld_out);

/* runNeighborNet (NetMakerOriginal.java:129-162): order_out has n+1 entries,
 * order_out[0] = 0, order_out[1] = 1, 1-based taxon ids in circular order; n <= 3
 * gives the identity (:133-140).  Consumes the device matrix (a new upload or
 * fnn_synth is needed before another run).  stats may be NULL. */
int32_t fnn_run(fnn_handle* h, int32_t* order_out, fnn_stats* stats);

/* Test/diagnostic stepping: fnn_begin computes the initial row sums; fnn_step
 * executes exactly one event and returns 1 (ev filled), or 0 when the loop has
 * ended; fnn_finish expands the merge stack into the order.  fnn_run ==
 * begin + step* + finish without the per-event host round trip. */
int32_t fnn_begin(fnn_handle* h);
int32_t fnn_step(fnn_handle* h, fnn_event* ev);
int32_t fnn_finish(fnn_handle* h, int32_t* order_out);

/* Trajectory of the last run (needs opts.record_events). Copies up to max_events
 * records; returns the number of events of the run (may exceed max_events). */
int64_t fnn_get_events(fnn_handle* h, fnn_event* out, int64_t max_events);

/* State inspection for parity tests.  For reference position i in
 * [0, num_active): node id, partner id (0 if none), Sx.  Arrays of length >= n. */
int32_t fnn_get_counts(fnn_handle* h, int32_t* num_active, int32_t* num_clusters, int32_t* num_nodes);
int32_t fnn_get_nodes(fnn_handle* h, int32_t* id, int32_t* nbr_id, double* Sx);
/* Sub-matrix of live nodes in reference position order: out[i*m + j] =
 * D[N[i].distID][N[j].distID], m = num_active. */
int32_t fnn_get_live_matrix(fnn_handle* h, double* out);

/* Diagnostic: one record of 5 doubles per scanned event of the last run {events done, live nodes,
 * window width W (-1: no window opened), pairs emitted, events the previous window served}; returns
 * the number of records copied (at most 8192 are kept). */
int64_t fnn_debug_window_log(fnn_handle* h, double* out, int64_t max_records);

/* Diagnostic: 100 MHz ticks (s_memrealtime) that thread 0 of the LAST-arriving tracking workgroup of k_track
 * spent, summed over the window events of the last run, in {prologue (control block, partial sums), tracked
 * pairs, sweep of the newest cluster's rows, workgroup reduction, arrival tickets, reading all workgroups'
 * records, the window's verdict, the decision tail (Cx/Cy, 4-candidate choice, merge plan)}. */
int32_t fnn_debug_event_ticks(fnn_handle* h, int64_t* out8);
/* Diagnostic (FNN_TICKS=1): the same for k_update, summed over all events: thread 0 of the workgroup of the involved
 * slots {control block, block load, phases, tail} and of the first bulk workgroup {control block, -, columns, tail}. */
int32_t fnn_debug_update_ticks(fnn_handle* h, int64_t* out8);
/* ... and the decide step inside k_track's tail: {the one round trip of loads, Cx/Cy + certified choice, merge plan,
 * symbolic replay of the micro-ops}. */
int32_t fnn_debug_decide_ticks(fnn_handle* h, int64_t* out4);
/* ... and inside the merge plan (wave 0; all decide steps of the run, k_decide's included): {the <= 4 candidates' values and the choice,
 * the chosen nodes' ids, the slot operations of the merge (swaps / agg3way plans / moves), the window's bookkeeping for the new cluster}. */
int32_t fnn_debug_plan_ticks(fnn_handle* h, int64_t* out4);
/* ... and k_update per workgroup: out768[w] = sum over the events of workgroup w's start stamp, out768[256 + w] = of its end stamp,
 * out768[512 + w] = the number of events it took part in (w = 255: the workgroup of the involved slots; bulk workgroups >= 254
 * share slot 254): which workgroup ends last, and by how much. */
int32_t fnn_debug_update_wg_ticks(fnn_handle* h, int64_t* out768);
/* Relaxed mode, FNN_TICKS=1: 100 MHz ticks summed over the run in {the workgroup's row-minimum passes, their finish by
 * the control lane, the whole search kernels}, and the number of row minima computed. */
int32_t fnn_debug_relaxed_ticks(fnn_handle* h, int64_t* out4);

/* Per-launch HIP-event timing on the engine's stream (two event records per timed launch): 1 = the streaming scan
 * kernels only (bench.py's roofline figure: fnn_stats.t_scan_s / scan_launches), 2 = every kernel of the launch
 * sequences (perturbs the run a little: for an extra, untimed run), 0 = off. */
int32_t fnn_set_scan_timing(fnn_handle* h, int32_t enable);
/* With timing on: milliseconds and launches of the last run by kernel class {0 k_scan, 1 k_screen, 2 k_track,
 * 3 k_decide, 4 k_update, 5 k_emit, 6 k_resolve, 7 k_finalize}. */
int32_t fnn_get_kernel_times(fnn_handle* h, double* ms8, int64_t* launches8);
/* Several ranks, timing 2: milliseconds and calls of {0 the all-gather of a sharded base scan on the stream, 1 k_merge}. */
int32_t fnn_get_exchange_times(fnn_handle* h, double* ms2, int64_t* launches2);

/* One-call convenience == create + set_rows + run + destroy
 * (FastNN.java:326 + :378). */
int32_t fnn_canonical_order_f64(const double* D, int32_t n, int64_t ld, const fnn_opts* opts,
                                int32_t* order_out, fnn_stats* stats);

/* ---- several GPUs of one node (one process per GPU): ONE problem on all ranks -----------------
 * Every rank holds the whole matrix (8 GiB at n = 32768, 3 % of an MI355X) and runs the whole event chain
 * itself (the ranks stay in step because every decision is a deterministic function of identical state), so no
 * matrix row ever crosses xGMI.  All ranks must upload the same matrix and make the same calls; every rank
 * returns the same order.  What is shared out:
 *   - with lookahead windows (a screening copy exists: n >= 4096, lookahead not off): only the BASE SCANS of the
 *     windows, by screening-tile index mod world; each is followed by ONE all-gather of a fixed block per rank -
 *     16 B header + 64 candidate records of 24 B + 65536 / world tracked-pair records of 48 B - after which every
 *     rank builds the same tracked list and reduces the same candidate records;
 *   - without windows: the scan of every event, by tile index mod world, with one all-gather of <= 64 candidate
 *     records (24 B each) per rank and event.
 *
 * fnn_comm_unique_id: rank 0 creates the 128-byte RCCL id and the caller ships it to the other
 * ranks (bench.py uses torch.distributed for that).  fnn_comm_init_rccl: collective over all
 * ranks; the all-gathers then run as ncclAllGather on the engine's stream (no host round
 * trip).  rccl_path may name the librccl.so to dlopen (NULL: default search).
 * fnn_comm_init_host: test transport - the engine synchronises once per exchange and calls
 * `fn(ctx, send, recv, bytes_per_rank)` on the host, which must fill recv with every rank's
 * `send` in rank order and return 0. */
#define FNN_COMM_ID_BYTES 128
typedef int32_t (*fnn_allgather_fn)(void* ctx, const void* send, void* recv, int32_t bytes_per_rank);
int32_t fnn_comm_unique_id(uint8_t* id_out, const char* rccl_path);
/* Can this process load librccl (dlopen + the four symbols the engine uses)?  No call is made into the library: no
 * bootstrap listener is started, nothing needs tearing down.  FNN_OK or FNN_ERCCL with fnn_last_error(). */
int32_t fnn_comm_probe(const char* rccl_path);
int32_t fnn_comm_init_rccl(fnn_handle* h, int32_t world, int32_t rank, const uint8_t* id, const char* rccl_path);
int32_t fnn_comm_init_host(fnn_handle* h, int32_t world, int32_t rank, fnn_allgather_fn fn, void* ctx);

/* ---- circular split weights ------------------------------------------------
 * Non-negative least-squares weights of the n(n-1)/2 circular splits of `ordering` (the array
 * fnn_run returns: n + 1 entries, [1..n] = 1-based taxon ids in circular order) for the symmetric
 * n x n distances D (host memory, row stride ld): the unique optimum that the reference's live path
 * computes with a dense design matrix (FastNN.java:401-454), here on the implicit operators A b, A^T y
 * of CircularSplitWeights.java (2-D prefix sums) with the re-ordering of the distances restored.
 * Three routes to the same optimum: the Chepoi-Fichet closed form when it is feasible; "from below"
 * (a BLOCK active-set method: whole blocks of splits - local maxima of the multiplier - enter the free set
 * per step and every weight that is not positive leaves; the free set's normal equations, whose entries have
 * a closed form, are held as an inverse Cholesky factor that is appended to, never updated; right for
 * distances that are far from circular, where only ~2.4 n ... 3.8 n splits end up positive; 32768 taxa in
 * about a minute); CircularSplitWeights.java's own active-set / conjugate-gradient
 * method (from above) for inputs whose free set outgrows the dense factor (nearly circular metrics) - and
 * only where that route is affordable: see fnn_sw_stats.status.  weights_out has n(n-1)/2
 * entries in the index order of the reference's live path (FastNN.java:405-419): k runs over
 * (i, j), 0 <= i < j <= n-1, row-major, split k = taxa ordering[i+1 .. j] against the rest.  The
 * reference keeps the splits with weight > 1e-6 (FastNN.java:455). */
typedef struct fnn_sw_stats {
    int64_t outer_iterations; /* from below: steps (a BLOCK of splits enters the free set per step); reference method: passes of its active-set loop */
    int64_t cg_calls;         /* conjugate-gradient solves (reference method only) */
    int64_t cg_iterations;    /* ... and their iterations (each applies A and A^T once) */
    int64_t nsplits;          /* weights above 1e-6 */
    double  t_solve_s;        /* device time from the re-ordered distances to the weights */
    int64_t reserved[3];      /* [0] 1 = solved from below (block active-set method), [1] rebuilds of the factor, [2] sub-problems solved */
    /* ---- ABI version 2 ---- */
    int32_t route;            /* FNN_SW_ROUTE_*: which of the three routes produced weights_out */
    int32_t certified;        /* 1 = the weights passed the solver's own Kuhn-Tucker check (kkt_violation <= 1e-9) */
    double  kkt_violation;    /* max(-min x, max |g| over x > 0, max -g over x = 0) / max|A^T d| with g = A^T (A x - d), evaluated
                                 on the device from the RETURNED weights with the implicit operators (independent of the factor) */
    double  final_threshold_rel; /* from below: the candidates' final multiplier threshold / max|A^T d|: 1e-12 unless the
                                 noise-floor rule raised it */
    int64_t n_set_aside;      /* from below: splits still set aside at termination (numerically dependent on the factor, or no
                                 measurable descent); their multipliers are covered by kkt_violation */
    int64_t capacity;         /* from below: splits the dense factor holds on this device for this n */
    int64_t free_set_peak;    /* from below: the largest number of splits the factor held */
    int32_t giveup_reason;    /* from below gave up (FNN_SW_GIVEUP_*; 0 = it did not): the reference route ran, or FNN_ECAPACITY */
    int32_t pad_;
    int64_t entered, screened_out, departed; /* from below: splits appended / dropped by the Schur-complement screening / left */
    double  t_alloc_s;        /* seconds in hipMalloc for the factor's buffers (0.1-2.7 s from box to box at 32768 taxa) */
    int64_t reserved2[4];
} fnn_sw_stats;
enum { FNN_SW_ROUTE_CLOSED_FORM = 0, FNN_SW_ROUTE_FROM_BELOW = 1, FNN_SW_ROUTE_REFERENCE = 2 };
enum { FNN_SW_GIVEUP_NONE = 0, FNN_SW_GIVEUP_CAPACITY = 1,   /* the free set outgrew the dense factor */
       FNN_SW_GIVEUP_STEPS = 2,                              /* step limit (40 n + 1000) */
       FNN_SW_GIVEUP_NUMERIC = 3,                            /* BLAS / Cholesky failure */
       FNN_SW_GIVEUP_DEPARTED = 4,                           /* the departed splits outgrew their buffers */
       FNN_SW_GIVEUP_SETUP = 5 };                            /* allocation / library set-up failed */
/* Returns FNN_OK with the certified optimum; FNN_EINEXACT with weights that failed the check (from below only: the
 * reference route stops by its own rule, CG_EPSILON = 1e-8, ~1e-5 short of the optimum, and reports certified = 0 with
 * FNN_OK); FNN_ECAPACITY when the block method's factor cannot hold the free set and n is above the size up to which the
 * reference route is taken automatically (FNN_SW_REFERENCE_MAX_N, default 1024: measured in DESIGN.md section 7).  Before
 * that the block method retries with a factor of four times the capacity, up to what device memory holds (~150 000 splits). */
int32_t fnn_split_weights_f64(const double* D, int32_t n, int64_t ld, const int32_t* ordering, int32_t device,
                              double* weights_out, fnn_sw_stats* stats);
/* The same solve, returning only the weights above `threshold` (the reference's list keeps x[k] > 1e-6, FastNN.java:455-466): pairs
 * (index_out[q], weight_out[q]) in ascending live index k - the order of the reference's `splits` list -, *count_out = how many
 * there are (if it exceeds `capacity`, only the first `capacity` found are returned: call again with more room).  At 32768 taxa:
 * 77 000 pairs instead of 5.4e8 doubles (4.3 GB) over the bus.  FNN_EINEXACT / FNN_ECAPACITY as above. */
int32_t fnn_split_weights_sparse_f64(const double* D, int32_t n, int64_t ld, const int32_t* ordering, int32_t device, double threshold,
                                     int64_t* index_out, double* weight_out, int64_t capacity, int64_t* count_out, fnn_sw_stats* stats);
/* The solver keeps its large device buffers (>= 64 MiB each, ~220 GB at 32768 taxa) in a per-process pool between calls - hipMalloc of
 * them costs 0.5-4.8 s - and hands them out again to a later call of the same size; this call gives them back to the driver.  (The
 * pool is also emptied when a device allocation of the solver fails; the order engine's own 10 GiB at 32768 taxa fit beside it.) */
int32_t fnn_split_weights_release_cache(void);

/* Diagnostic: the exact block-parallel evaluation of the sequential fp64 sum
 * (((0 + b[0]) + b[1]) + ...) used for ComputeRx / u.Sx (NetMakerOriginal.java:551-560,
 * :532), run on an arbitrary host buffer.  ept = addends per thread (32, fixed by the
 * buffer layout); guard_bits != 0 in production, 0 to provoke the fallback paths.  stats4 (may be NULL)
 * = {runs applied, chunks added one by one, rejected runs, rejected chunks}. */
int32_t fnn_test_chain_sum(int32_t device, const double* host_buf, int32_t m, int32_t guard_bits,
                           int32_t ept, double* out, int32_t* stats4);

/* Device-side read-only streaming probe: reads `bytes` bytes `reps` times and
 * returns the achieved GB/s (the "measured stream" line next to the nominal
 * 8 TB/s peak). */
int32_t fnn_stream_probe(int32_t device, int64_t bytes, int32_t reps, double* gbps_out);

#ifdef __cplusplus
}
#endif
#endif /* FASTNN_H */
