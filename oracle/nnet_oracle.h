/*
 * nnet_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C, IEEE fp64, no FMA contraction) of the Canonical
 * Neighbor-Net circular-ordering path of JacobPorter/FastNeighborNet:
 *   NetNode.java:5-15, NetMakerOriginal.java:129-726,
 *   NeighborNetCanonical.java:151-179 (serial branch).
 *
 * PARITY UNPINNED: the reference ships no tests, fixtures or golden vectors and
 * there is no JVM in the build container, so this restatement could not be
 * checked against outputs of the reference itself.  It is pinned only by a
 * hand-derived known-answer trace (tests/golden/kat5.json), by an independent
 * second restatement (oracle/nnet_ref.py) and by structural invariants.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * this library.  The product (libfastnn_hip.so) never links or calls it.
 */
#ifndef NNET_ORACLE_H
#define NNET_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* kind of agglomeration event (NetMakerOriginal.java:462-488, 343-360) */
enum { NNO_KIND_2WAY = 2, NNO_KIND_3WAY = 3, NNO_KIND_4WAY = 4, NNO_KIND_FINISH = 5 };

typedef struct nno_event {
    int32_t m_before;      /* num_active when the event started */
    int32_t c_before;      /* num_clusters when the event started */
    int32_t cx_id, cy_id;  /* Cx.id, Cy.id after the id swap (:376-380); 0 for FINISH */
    int32_t x_id, y_id;    /* nodes picked among the <=4 candidates (:428-452); for FINISH: p.id, q.id */
    int32_t kind;          /* NNO_KIND_* */
    int32_t u_id;          /* id of the node returned by agg2way/agg3way/agg4way */
    double  best;          /* scan minimum Qpq (value of field `best` after findNodes); 0 for FINISH */
    int64_t entries;       /* E_t = m(m-1)/2 - (m-c): matrix entries the scan had to read */
} nno_event;

typedef struct nno_handle nno_handle;

/* D: n*n row-major fp64, symmetric, zero diagonal.  The oracle COPIES it (the
 * reference mutates the caller's array in place, NetMakerOriginal.java:653-656;
 * the copy plays that role).  threads: 1 = the exact serial scan; >1 = an
 * OpenMP scan reduced on the total order (Q, i, j), which by construction
 * returns the same pair as the serial first-strict-minimum scan. */
nno_handle* nno_create(const double* D, int32_t n, int32_t threads);
void        nno_destroy(nno_handle* h);

/* Run one agglomeration event.  Returns 1 if an event was executed (ev filled),
 * 0 if the agglomeration loop has ended (num_active <= 3 or the special finish
 * has run), negative on error. */
int32_t nno_step(nno_handle* h, nno_event* ev);

/* State inspection (any time between steps). */
int32_t nno_num_active(const nno_handle* h);
int32_t nno_num_clusters(const nno_handle* h);
int32_t nno_num_nodes(const nno_handle* h);
/* For position i in [0, num_active): id, distID, nbr id (0 if none), Sx. Arrays of length >= n. */
void    nno_get_nodes(const nno_handle* h, int32_t* id, int32_t* distID, int32_t* nbr_id, double* Sx);
/* Pointer to the (mutated) n*n matrix, indexed by distID. */
const double* nno_matrix(const nno_handle* h);

/* expandNodes (NetMakerOriginal.java:246-325).  Call after nno_step returned 0.
 * order_out has n+1 entries. Returns 0 on success. */
int32_t nno_expand(nno_handle* h, int32_t* order_out);

/* Whole run: runNeighborNet (NetMakerOriginal.java:129-162). events_out may be
 * NULL; otherwise up to max_events records are stored and *n_events receives
 * the number of events executed.  sum_entries (may be NULL) receives sum E_t. */
int32_t nno_run(const double* D, int32_t n, int32_t threads, int32_t* order_out,
                nno_event* events_out, int64_t max_events, int64_t* n_events,
                int64_t* sum_entries);

/* Relaxed mode (NeighborNetLocal.java, additive == false; FastNN.java:329-338): call after nno_create and before the
 * first nno_step.  The reference's ThreadLocalRandom cannot be seeded; the draws come from java.util.Random(seed)
 * instead (the generator of the line it replaced, NeighborNetLocal.java:27).  min_active: the relaxed search runs
 * while num_active > min_active (0 = the reference's 1024, NetMakerOriginal.java:361; smaller values for tests).
 * Events of the relaxed search record best = the chosen RowMinimum's value and entries = Q values evaluated.
 * nno_step returns -3 if the search runs out without a pair of mutual row minima. */
void    nno_set_relaxed(nno_handle* h, uint64_t seed, int32_t min_active);
int32_t nno_run_relaxed(const double* D, int32_t n, uint64_t seed, int32_t min_active, int32_t* order_out,
                        nno_event* events_out, int64_t max_events, int64_t* n_events);
/* java.util.Random.nextInt(bound) on an explicit state (known-answer tests of the generator). */
int32_t nno_java_random_next_int(uint64_t* state, int32_t bound, int32_t init_with_seed);

/* SplitMix64 synthetic generator shared by tests and bench:
 * strict upper triangle row-major, k-th value from the k-th SplitMix64 output.
 * dist 0 = uniform53: (next>>11)*2^-53 + 2^-10 ; dist 1 = dec4:
 * floor(u*1e4 + 1)/1e4 with u = (next>>11)*2^-53 (4-decimal, tie-rich).
 * Writes a full symmetric n*n matrix with zero diagonal. */
void nno_synth(double* D, int32_t n, uint64_t seed, int32_t dist);

#ifdef __cplusplus
}
#endif
#endif
