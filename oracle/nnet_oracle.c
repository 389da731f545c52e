/*
 * nnet_oracle.c -- TEST INFRASTRUCTURE ONLY (see nnet_oracle.h; PARITY UNPINNED).
 *
 * Statement-by-statement CPU restatement of the reference's Canonical path.
 * Every function cites the reference lines it follows.  Node objects are kept
 * as structs with pointers, as in NetNode.java:5-15, so that pointer identity
 * tests (p == Cx, q.nbr == p ...) read exactly like the Java.
 *
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math (see oracle/Makefile).
 * Every fp64 expression is written in the reference's source order; Java
 * evaluates a+b+c+d as ((a+b)+c)+d and never contracts to FMA.
 */
#include "nnet_oracle.h"

#include <float.h>
#include <stdlib.h>
#include <string.h>

typedef struct NetNode {
    int id;          /* NetNode.java:6 */
    int distID;      /* :7 */
    int positionID;  /* :8 */
    struct NetNode* nbr;  /* :9 */
    struct NetNode* ch1;  /* :10 */
    struct NetNode* ch2;  /* :11 */
    struct NetNode* next; /* :12 */
    struct NetNode* prev; /* :13 */
    double Sx;       /* :15 */
    /* Relaxed mode only: this node's entry of the per-call HashMap foundRowMinimums (NeighborNetLocal.java:171) */
    int64_t rm_stamp;     /* == handle's rm_call when the entry is present */
    int64_t rm_off;       /* first RowMinimum of its list in the handle's rm_pool */
    int32_t rm_cnt;
} NetNode;

/* NetMakerOriginal.java:39-48 */
typedef struct RowMinimum {
    NetNode* me;
    NetNode* row;
    double value;
} RowMinimum;

struct nno_handle {
    int ntax;
    int threads;
    double* D;            /* ntax*ntax, by distID (NetMakerOriginal.java:30) */
    NetNode** netNodes;   /* packed active array (:141) */
    NetNode* pool;        /* node storage: ntax + 2 per agg3way (<= 3*ntax) */
    int pool_used;
    NetNode** amalgs;     /* Stack<NetNode> (:154) */
    int amalgs_top;
    int num_nodes, num_active, num_clusters;
    int finished;         /* agglomNodes loop left */
    /* fields of NetMakerOriginal (:33-35) */
    double best;
    NetNode* Cx;
    NetNode* Cy;
    /* Relaxed mode (NeighborNetLocal.java:14-32) */
    int relaxed;          /* 0 = Canonical */
    int relaxed_min;      /* findNodes is the relaxed search while num_active > relaxed_min (1024, NetMakerOriginal.java:361) */
    uint64_t rng;         /* java.util.Random state (48 bits) */
    int* rowPermutation;  /* :16 */
    int firstTime;        /* :17 */
    int top;              /* :18 */
    int64_t rm_call;      /* findNodes call number (a fresh HashMap per call, :171) */
    RowMinimum* rm_pool; int64_t rm_used, rm_cap;
    int64_t q_evals;      /* Q values evaluated by the current call (the commented-out counter, :76/:119) */
    double chosen_value;  /* combineMe.value of the current call */
};

#define DD(h, a, b) ((h)->D[(size_t)(a) * (size_t)(h)->ntax + (size_t)(b)])

static NetNode* new_node(nno_handle* h) {
    NetNode* p = &h->pool[h->pool_used++];
    memset(p, 0, sizeof(*p)); /* Java field defaults: 0 / null (NetNode.java:6-15) */
    return p;
}

/* NetMakerOriginal.java:164-191 initialize */
static void initialize(nno_handle* h, int num_nodes) {
    NetNode** netNodes = h->netNodes;
    for (int pi = 0; pi < h->ntax; pi++) {
        NetNode* p = netNodes[pi];
        if (p->nbr == NULL || p->nbr->id > p->id) {
            for (int j = p->positionID + 1; j < num_nodes; j++) {
                NetNode* q = netNodes[j];
                if (q->nbr == NULL || ((q->nbr->id > q->id) && (q->nbr != p))) {
                    double Dpq = 0.0;
                    if ((p->nbr == NULL) && (q->nbr == NULL))
                        Dpq = DD(h, p->distID, q->distID);
                    else if ((p->nbr != NULL) && (q->nbr == NULL))
                        Dpq = (DD(h, p->distID, q->distID) + DD(h, p->nbr->distID, q->distID)) / 2.0;
                    else if ((p->nbr == NULL) && (q->nbr != NULL))
                        Dpq = (DD(h, p->distID, q->distID) + DD(h, p->distID, q->nbr->distID)) / 2.0;
                    else
                        Dpq = (DD(h, p->distID, q->distID) + DD(h, p->distID, q->nbr->distID) +
                               DD(h, p->nbr->distID, q->distID) + DD(h, p->nbr->distID, q->nbr->distID)) / 4.0;
                    p->Sx += Dpq;
                    if (p->nbr != NULL) p->nbr->Sx += Dpq;
                    q->Sx += Dpq;
                    if (q->nbr != NULL) q->nbr->Sx += Dpq;
                }
            }
        }
    }
}

/* One row of the scan: NeighborNetCanonical.java:152-178 for a fixed i,
 * identical to NetMakerOriginal.java:209-234.  Updates (*Cx,*Cy,*best) with the
 * reference's first-strict-minimum rule. */
static inline void scan_row(const nno_handle* h, int i, int num_clusters,
                            NetNode** Cx, NetNode** Cy, double* best) {
    NetNode** netNodes = h->netNodes;
    double Dpq, Qpq;
    NetNode* p = netNodes[i];
    if ((p->nbr != NULL) && (p->nbr->id < p->id)) return; /* one node per cluster */
    for (int j = 0; j != i; j++) {
        NetNode* q = netNodes[j];
        if ((q->nbr != NULL) && (q->nbr->id < q->id)) continue;
        if (q->nbr == p) continue;
        if ((p->nbr == NULL) && (q->nbr == NULL))
            Dpq = DD(h, p->distID, q->distID);
        else if ((p->nbr != NULL) && (q->nbr == NULL))
            Dpq = (DD(h, p->distID, q->distID) + DD(h, p->nbr->distID, q->distID)) / 2.0;
        else if ((p->nbr == NULL) && (q->nbr != NULL))
            Dpq = (DD(h, p->distID, q->distID) + DD(h, p->distID, q->nbr->distID)) / 2.0;
        else
            Dpq = (DD(h, p->distID, q->distID) + DD(h, p->distID, q->nbr->distID) +
                   DD(h, p->nbr->distID, q->distID) + DD(h, p->nbr->distID, q->nbr->distID)) / 4.0;
        Qpq = ((double)num_clusters - 2.0) * Dpq - p->Sx - q->Sx;
        if ((*Cx == NULL || (Qpq < *best)) && (p->nbr != q)) {
            *Cx = p;
            *Cy = q;
            *best = Qpq;
        }
    }
}

/* NeighborNetCanonical.java:139-179 findNodes (numThreads == 1 branch)
 * == NetMakerOriginal.java:197-236 findNodesDefault. */
static void findNodes_serial(nno_handle* h, int num_active, int num_clusters) {
    h->Cx = h->Cy = NULL;
    h->best = DBL_MAX;
    NetNode* Cx = NULL; NetNode* Cy = NULL; double best = DBL_MAX;
    for (int i = 0; i < num_active; i++) scan_row(h, i, num_clusters, &Cx, &Cy, &best);
    h->Cx = Cx; h->Cy = Cy; h->best = best;
}

/* CPU-baseline variant (not a restatement of the reference's racy pool branch,
 * NeighborNetCanonical.java:180-206, which SURVEY.md F7 shows is not an oracle):
 * rows are scanned in parallel; per-thread results are combined on the total
 * order (Q, i, j), which selects the same pair as the serial scan because the
 * serial rule "first strict minimum in (i asc, j asc)" is exactly argmin on that
 * order.  A thread's local result over an ascending subset of rows obeys the
 * same rule, so only the cross-thread merge needs the explicit key compare. */
static void findNodes_omp(nno_handle* h, int num_active, int num_clusters) {
    NetNode* gCx = NULL; NetNode* gCy = NULL; double gbest = DBL_MAX;
#pragma omp parallel num_threads(h->threads)
    {
        NetNode* Cx = NULL; NetNode* Cy = NULL; double best = DBL_MAX;
#pragma omp for schedule(dynamic, 16) nowait
        for (int i = 0; i < num_active; i++) scan_row(h, i, num_clusters, &Cx, &Cy, &best);
#pragma omp critical
        {
            if (Cx != NULL) {
                int take = 0;
                if (gCx == NULL) take = 1;
                else if (best < gbest) take = 1;
                else if (best == gbest) {
                    if (Cx->positionID < gCx->positionID) take = 1;
                    else if (Cx->positionID == gCx->positionID && Cy->positionID < gCy->positionID) take = 1;
                }
                if (take) { gCx = Cx; gCy = Cy; gbest = best; }
            }
        }
    }
    h->Cx = gCx; h->Cy = gCy; h->best = gbest;
}

/* ---------------------------------------------------------------------------------------------
 * Relaxed mode: NeighborNetLocal.java (serial branch, additive == false).
 *
 * Randomness: the reference draws from ThreadLocalRandom.current() (:30), which cannot be seeded, so no two runs of the
 * reference agree; the line it replaced, `new Random(System.currentTimeMillis())` (:27), is java.util.Random.  The
 * oracle (and the engine) use java.util.Random's documented generator with an explicit seed: the run is then exactly
 * what the reference computes with `myRandom = new Random(seed)`.
 * ------------------------------------------------------------------------------------------- */
static void jr_set_seed(nno_handle* h, uint64_t seed) { h->rng = (seed ^ 0x5DEECE66DULL) & ((1ULL << 48) - 1); }
static int32_t jr_next(nno_handle* h, int bits) {  /* java.util.Random.next */
    h->rng = (h->rng * 0x5DEECE66DULL + 0xBULL) & ((1ULL << 48) - 1);
    return (int32_t)(h->rng >> (48 - bits));
}
static int32_t jr_next_int(nno_handle* h, int32_t bound) {  /* java.util.Random.nextInt(int) */
    int32_t r = jr_next(h, 31);
    const int32_t m = bound - 1;
    if ((bound & m) == 0) r = (int32_t)(((int64_t)bound * (int64_t)r) >> 31);
    else {
        for (int32_t u = r; (int32_t)((uint32_t)u - (uint32_t)(r = u % bound) + (uint32_t)m) < 0; u = jr_next(h, 31)) {}
    }
    return r;
}

static RowMinimum* rm_push(nno_handle* h, NetNode* me, NetNode* row, double value) {
    if (h->rm_used == h->rm_cap) {
        h->rm_cap = h->rm_cap ? 2 * h->rm_cap : 1024;
        h->rm_pool = (RowMinimum*)realloc(h->rm_pool, sizeof(RowMinimum) * (size_t)h->rm_cap);
    }
    RowMinimum* r = &h->rm_pool[h->rm_used++];
    r->me = me; r->row = row; r->value = value;
    return r;
}

/* NeighborNetLocal.java:88-157 findRowMin, numThreads == 1 branch (:96-125).  Returns the list through (*off, *cnt). */
static void findRowMin(nno_handle* h, NetNode* p, int num_active, int num_clusters, int64_t* off, int32_t* cnt) {
    if (p->rm_stamp == h->rm_call) { *off = p->rm_off; *cnt = p->rm_cnt; return; }                         /* :89-91 */
    if (p->nbr != NULL && p->nbr->rm_stamp == h->rm_call) { *off = p->nbr->rm_off; *cnt = p->nbr->rm_cnt; return; } /* :92-94 */
    NetNode** netNodes = h->netNodes;
    double myMin = DBL_MAX;
    int64_t start = h->rm_used;
    double Dpq, Qpq;
    for (int row = 0; row < num_active; row++) {
        NetNode* q = netNodes[row];
        if ((p == q) || ((p->nbr != NULL) && (p->nbr == q))) continue;
        if ((p->nbr == NULL) && (q->nbr == NULL))
            Dpq = DD(h, p->distID, q->distID);
        else if ((p->nbr != NULL) && (q->nbr == NULL))
            Dpq = (DD(h, p->distID, q->distID) + DD(h, p->nbr->distID, q->distID)) / 2.0;
        else if ((p->nbr == NULL) && (q->nbr != NULL))
            Dpq = (DD(h, p->distID, q->distID) + DD(h, p->distID, q->nbr->distID)) / 2.0;
        else
            Dpq = (DD(h, p->distID, q->distID) + DD(h, p->distID, q->nbr->distID) +
                   DD(h, p->nbr->distID, q->distID) + DD(h, p->nbr->distID, q->nbr->distID)) / 4.0;
        Qpq = ((double)num_clusters - 2.0) * Dpq - p->Sx - q->Sx;
        h->q_evals++;
        if (Qpq < myMin) {
            myMin = Qpq;
            h->rm_used = start;            /* myMinimums.clear() */
            rm_push(h, p, q, Qpq);
        } else if (Qpq == myMin) {
            rm_push(h, p, q, Qpq);
        }
    }
    p->rm_stamp = h->rm_call; p->rm_off = start; p->rm_cnt = (int32_t)(h->rm_used - start);   /* :155 */
    *off = p->rm_off; *cnt = p->rm_cnt;
}

static void swap_int(int* x, int a, int b) { int t = x[a]; x[a] = x[b]; x[b] = t; }   /* :159-163 */

/* NeighborNetLocal.java:170-264 findNodes with additive == false.  Returns 0 if the loop ran out without a pair of
 * mutual row minima (the reference would then reuse the previous event's Cx / Cy: not restated, reported as an error). */
static int findNodes_relaxed(nno_handle* h, int num_active, int num_clusters) {
    NetNode** netNodes = h->netNodes;
    int* rowPermutation = h->rowPermutation;
    h->rm_call++;                 /* new HashMap (:171) */
    h->rm_used = 0;
    h->q_evals = 0;
    /* myMinimums (:172): entries are copies of RowMinimum records */
    RowMinimum myMinimums[64];
    int nmin = 0;
    if (h->firstTime) {
        for (int i = 0; i < h->ntax; i++) rowPermutation[i] = i;
        h->firstTime = 0;
        h->top = h->ntax - 1;
    }
    for (int i = h->top + 1; i > 0; i--) {
        int swapCell = jr_next_int(h, i);
        if (rowPermutation[swapCell] >= num_active) {
            swap_int(rowPermutation, swapCell, h->top);
            if (i == h->top + 1) i--;
            else i++;
            h->top--;
            continue;
        }
        swap_int(rowPermutation, i - 1, swapCell);
        NetNode* p = netNodes[rowPermutation[i - 1]];
        if ((p->nbr != NULL) && (p->nbr->id < p->id)) continue;   /* one node per cluster (:201-203) */
        int64_t off; int32_t cnt;
        findRowMin(h, p, num_active, num_clusters, &off, &cnt);
        for (int32_t a = 0; a < cnt; a++) {
            NetNode* other = h->rm_pool[off + a].row;
            int64_t off2; int32_t cnt2;
            findRowMin(h, other, num_active, num_clusters, &off2, &cnt2);
            for (int32_t b = 0; b < cnt2; b++) {
                const RowMinimum testRM = h->rm_pool[off2 + b];
                if ((testRM.row == p) || ((testRM.row->nbr != NULL) && (testRM.row->nbr == p)) ||
                    ((testRM.row->nbr != NULL) && (p->nbr != NULL) && (testRM.row->nbr == p->nbr)) ||
                    ((p->nbr != NULL) && (testRM.row == p->nbr))) {
                    if (nmin < 64) myMinimums[nmin] = testRM;
                    nmin++;
                    break;
                }
            }
        }
        if (nmin > 0) {
            if (nmin > 64) return -2;   /* (more tied mutual minima than this restatement keeps) */
            int choice = jr_next_int(h, nmin);
            RowMinimum combineMe = myMinimums[choice];
            h->Cx = combineMe.me;
            h->Cy = combineMe.row;
            h->chosen_value = combineMe.value;
            return 1;                 /* break outerloop (:258) */
        }
    }
    return 0;
}

/* NetMakerOriginal.java:549-561 ComputeRx */
static double ComputeRx(nno_handle* h, NetNode* z, NetNode* Cx, NetNode* Cy, int num_active) {
    double Rx = 0.0;
    for (int i = 0; i < num_active; i++) {
        NetNode* p = h->netNodes[i];
        if (p == Cx || p == Cx->nbr || p == Cy || p == Cy->nbr || p->nbr == NULL)
            Rx += DD(h, z->distID, p->distID);
        else
            Rx += DD(h, z->distID, p->distID) / 2.0;
    }
    return Rx;
}

/* NetMakerOriginal.java:570-577 agg2way */
static NetNode* agg2way(NetNode* x, NetNode* y) {
    x->nbr = y;
    y->nbr = x;
    return x;
}

/* NetMakerOriginal.java:589-674 agg3way */
static NetNode* agg3way(nno_handle* h, NetNode* x, NetNode* y, NetNode* z, int num_nodes, int num_active) {
    NetNode** netNodes = h->netNodes;
    NetNode* u = new_node(h);
    u->id = num_nodes + 1;
    u->ch1 = x;
    u->ch2 = y;

    NetNode* v = new_node(h);
    v->id = num_nodes + 2;
    v->ch1 = y;
    v->ch2 = z;

    /* Replace x by u (:623-625) */
    netNodes[x->positionID] = u;
    u->positionID = x->positionID;
    u->distID = x->distID;

    /* Replace z by v (:630-632) */
    netNodes[z->positionID] = v;
    v->positionID = z->positionID;
    v->distID = z->distID;

    /* Remove y (:641-643) */
    netNodes[y->positionID] = netNodes[num_active - 1];
    netNodes[y->positionID]->positionID = y->positionID;
    netNodes[num_active - 1] = NULL;

    u->nbr = v;
    v->nbr = u;

    /* Update distance matrix (:653-656); sequential, in place, reads see earlier writes */
    for (int i = 0; i < num_active - 1; i++) {
        NetNode* p = netNodes[i];
        double t1 = (2.0 / 3.0) * DD(h, x->distID, p->distID) + DD(h, y->distID, p->distID) / 3.0;
        DD(h, p->distID, u->distID) = t1;
        DD(h, u->distID, p->distID) = t1;
        double t2 = (2.0 / 3.0) * DD(h, z->distID, p->distID) + DD(h, y->distID, p->distID) / 3.0;
        DD(h, p->distID, v->distID) = t2;
        DD(h, v->distID, p->distID) = t2;
    }
    DD(h, v->distID, v->distID) = 0.0; /* :670 */
    DD(h, u->distID, u->distID) = 0.0;
    h->amalgs[h->amalgs_top++] = u;    /* :672 */
    return u;
}

/* NetMakerOriginal.java:681-696 subtractClusterDistance */
static void subtractClusterDistance(nno_handle* h, NetNode* p, NetNode* x) {
    if (p != x && p != x->nbr && (p->nbr == NULL || (p->nbr->id > p->id))) {
        double Dpx = 0.0;
        if ((p->nbr == NULL) && (x->nbr == NULL))
            Dpx = DD(h, p->distID, x->distID);
        else if ((p->nbr != NULL) && (x->nbr == NULL))
            Dpx = (DD(h, p->distID, x->distID) + DD(h, p->nbr->distID, x->distID)) / 2.0;
        else if ((p->nbr == NULL) && (x->nbr != NULL))
            Dpx = (DD(h, p->distID, x->distID) + DD(h, p->distID, x->nbr->distID)) / 2.0;
        else
            Dpx = (DD(h, p->distID, x->distID) + DD(h, p->distID, x->nbr->distID) +
                   DD(h, p->nbr->distID, x->distID) + DD(h, p->nbr->distID, x->nbr->distID)) / 4.0;
        p->Sx -= Dpx;
        if (p->nbr != NULL) p->nbr->Sx -= Dpx;
    }
}

/* NetMakerOriginal.java:707-726 agg4way */
static NetNode* agg4way(nno_handle* h, NetNode* x2, NetNode* x, NetNode* y, NetNode* y2,
                        int num_nodes, int num_active) {
    NetNode *u, *v;
    u = agg3way(h, x2, x, y, num_nodes, num_active);
    num_nodes += 2;
    v = agg3way(h, u, u->nbr, y2, num_nodes, num_active - 1);
    num_nodes += 2;
    x2->positionID = -1;
    x->positionID = -1;
    y->positionID = -1;
    y2->positionID = -1;
    u->positionID = -1;
    u->nbr->positionID = -1;
    return v;
}

/* NetMakerOriginal.java:517-536 updateClusterDistances */
static void updateClusterDistances(nno_handle* h, NetNode* u, int num_active) {
    u->Sx = 0;
    u->nbr->Sx = 0;
    for (int i = 0; i < num_active; i++) {
        NetNode* p = h->netNodes[i];
        if ((p->nbr == NULL || p->nbr->id > p->id) && (u->nbr != p) && (u != p)) {
            double Dpu = 0.0;
            if (p->nbr == NULL) {
                Dpu = (DD(h, p->distID, u->distID) + DD(h, p->distID, u->nbr->distID)) / 2.0;
            } else {
                Dpu = (DD(h, p->distID, u->distID) + DD(h, p->distID, u->nbr->distID) +
                       DD(h, p->nbr->distID, u->distID) + DD(h, p->nbr->distID, u->nbr->distID)) / 4.0;
            }
            p->Sx += Dpu;
            if (p->nbr != NULL) p->nbr->Sx += Dpu;
            u->Sx += Dpu;
        }
    }
    u->nbr->Sx = u->Sx;
}

/* NetMakerOriginal.java:397-515 handleAgglomerationEvent */
static void handleAgglomerationEvent(nno_handle* h, NetNode* Cx, NetNode* Cy, nno_event* ev) {
    int num_nodes = h->num_nodes, num_active = h->num_active, num_clusters = h->num_clusters;
    NetNode* x = Cx;
    NetNode* y = Cy;
    int m;
    double Qpq;
    double Cx_Rx = 0.0, Cx_nbr_Rx = 0.0, Cy_Rx = 0.0, Cy_nbr_Rx = 0.0;
    if (Cx->nbr != NULL || Cy->nbr != NULL) {
        Cx_Rx = ComputeRx(h, Cx, Cx, Cy, num_active);
        if (Cx->nbr != NULL) Cx_nbr_Rx = ComputeRx(h, Cx->nbr, Cx, Cy, num_active);
        Cy_Rx = ComputeRx(h, Cy, Cx, Cy, num_active);
        if (Cy->nbr != NULL) Cy_nbr_Rx = ComputeRx(h, Cy->nbr, Cx, Cy, num_active);
    }

    m = num_clusters;
    if (Cx->nbr != NULL) m++;
    if (Cy->nbr != NULL) m++;

    double best = ((double)m - 2.0) * DD(h, Cx->distID, Cy->distID) - Cx_Rx - Cy_Rx;
    if (Cx->nbr != NULL) {
        Qpq = ((double)m - 2.0) * DD(h, Cx->nbr->distID, Cy->distID) - Cx_nbr_Rx - Cy_Rx;
        if (Qpq < best) { x = Cx->nbr; y = Cy; best = Qpq; }
    }
    if (Cy->nbr != NULL) {
        Qpq = ((double)m - 2.0) * DD(h, Cx->distID, Cy->nbr->distID) - Cx_Rx - Cy_nbr_Rx;
        if (Qpq < best) { x = Cx; y = Cy->nbr; best = Qpq; }
    }
    if ((Cx->nbr != NULL) && (Cy->nbr != NULL)) {
        Qpq = ((double)m - 2.0) * DD(h, Cx->nbr->distID, Cy->nbr->distID) - Cx_nbr_Rx - Cy_nbr_Rx;
        if (Qpq < best) { x = Cx->nbr; y = Cy->nbr; best = Qpq; }
    }
    h->best = best; /* the Java stores into the field `best` here (:428) */
    ev->x_id = x->id;
    ev->y_id = y->id;

    /* Subtract old cluster distances (:455-461) */
    NetNode* u;
    for (int i = 0; i < num_active; i++) {
        NetNode* p = h->netNodes[i];
        if (i != x->positionID && i != y->positionID) {
            subtractClusterDistance(h, p, x);
            subtractClusterDistance(h, p, y);
        }
    }
    if ((NULL == x->nbr) && (NULL == y->nbr)) {
        u = agg2way(x, y);
        num_clusters--;
        ev->kind = NNO_KIND_2WAY;
    } else if (NULL == x->nbr) {
        u = agg3way(h, x, y, y->nbr, num_nodes, num_active);
        num_nodes += 2;
        num_active--;
        num_clusters--;
        x->positionID = -1;
        y->positionID = -1;
        y->nbr->positionID = -1;
        ev->kind = NNO_KIND_3WAY;
    } else if ((NULL == y->nbr) || (num_active == 4)) {
        u = agg3way(h, y, x, x->nbr, num_nodes, num_active);
        num_nodes += 2;
        num_active--;
        num_clusters--;
        x->positionID = -1;
        y->positionID = -1;
        x->nbr->positionID = -1;
        ev->kind = NNO_KIND_3WAY;
    } else {
        u = agg4way(h, x->nbr, x, y, y->nbr, num_nodes, num_active);
        num_nodes += 4;
        num_active -= 2;
        num_clusters--;
        ev->kind = NNO_KIND_4WAY;
    }
    /* Add new cluster distances (:490) */
    updateClusterDistances(h, u, num_active);
    ev->u_id = u->id;
    h->num_nodes = num_nodes;
    h->num_active = num_active;
    h->num_clusters = num_clusters;
}

nno_handle* nno_create(const double* D, int32_t n, int32_t threads) {
    if (n < 0 || D == NULL) return NULL;
    nno_handle* h = (nno_handle*)calloc(1, sizeof(*h));
    if (!h) return NULL;
    h->ntax = n;
    h->threads = threads < 1 ? 1 : threads;
    size_t nn = (size_t)n * (size_t)n;
    h->D = (double*)malloc(sizeof(double) * (nn ? nn : 1));
    h->netNodes = (NetNode**)calloc((size_t)n + 1, sizeof(NetNode*));
    h->pool = (NetNode*)malloc(sizeof(NetNode) * (3 * (size_t)n + 8));
    h->amalgs = (NetNode**)calloc((size_t)n + 8, sizeof(NetNode*));
    if (!h->D || !h->netNodes || !h->pool || !h->amalgs) { nno_destroy(h); return NULL; }
    memcpy(h->D, D, sizeof(double) * nn);

    /* runNeighborNet (:141-160): node creation */
    for (int i = n; i >= 1; i--) {
        NetNode* taxNode = new_node(h);
        taxNode->id = i;
        taxNode->positionID = i - 1;
        h->netNodes[i - 1] = taxNode;
        taxNode->distID = i - 1;
    }
    h->num_nodes = n;
    h->num_active = n;   /* agglomNodes :334-335 */
    h->num_clusters = n;
    h->finished = 0;
    h->relaxed = 0; h->relaxed_min = 1024; h->firstTime = 1;
    h->rowPermutation = (int*)calloc((size_t)n + 1, sizeof(int));
    if (n > 3) initialize(h, n); /* :159 (for n <= 3 runNeighborNet returns before, :133-140) */
    else h->finished = 1;
    return h;
}

void nno_destroy(nno_handle* h) {
    if (!h) return;
    free(h->D); free(h->netNodes); free(h->pool); free(h->amalgs); free(h->rowPermutation); free(h->rm_pool);
    free(h);
}

/* One iteration of the while loop of agglomNodes (NetMakerOriginal.java:339-393) */
int32_t nno_step(nno_handle* h, nno_event* ev) {
    nno_event local;
    if (!ev) ev = &local;
    if (h->finished) return 0;
    if (!(h->num_active > 3)) { h->finished = 1; return 0; }
    memset(ev, 0, sizeof(*ev));
    int num_active = h->num_active, num_clusters = h->num_clusters;
    ev->m_before = num_active;
    ev->c_before = num_clusters;

    if (num_active == 4 && num_clusters == 2) { /* :343-360 */
        NetNode* q;
        NetNode* p = h->netNodes[0];
        if (p->nbr != h->netNodes[1]) q = h->netNodes[1];
        else q = h->netNodes[2];
        ev->kind = NNO_KIND_FINISH;
        ev->x_id = p->id;
        NetNode* u;
        if (DD(h, p->distID, q->distID) + DD(h, p->nbr->distID, q->nbr->distID) <
            DD(h, p->distID, q->nbr->distID) + DD(h, p->nbr->distID, q->distID)) {
            ev->y_id = q->id;
            u = agg3way(h, p, q, q->nbr, h->num_nodes, num_active);
            h->num_nodes += 2;
        } else {
            ev->y_id = q->nbr->id;
            u = agg3way(h, p, q->nbr, q, h->num_nodes, num_active);
            h->num_nodes += 2;
        }
        ev->u_id = u->id;
        h->finished = 1; /* break */
        return 1;
    }
    ev->entries = (int64_t)num_active * (num_active - 1) / 2 - (num_active - num_clusters);
    /* :361-366: both branches are the same serial scan for numThreads == 1 (Canonical); in Relaxed mode findNodes
     * (num_active > 1024) is NeighborNetLocal's search, findNodesDefault the scan */
    if (h->relaxed && num_active > h->relaxed_min) {
        int rc = findNodes_relaxed(h, num_active, num_clusters);
        if (rc != 1) return rc == 0 ? -3 : rc;
        h->best = h->chosen_value;       /* (the reference leaves `best` alone; recorded for the trajectory) */
        ev->entries = h->q_evals;        /* Q values evaluated instead of E_t */
    } else if (h->threads > 1) findNodes_omp(h, num_active, num_clusters);
    else findNodes_serial(h, num_active, num_clusters);
    ev->best = h->best;
    if (h->Cx->id > h->Cy->id) { /* :376-380 */
        NetNode* temp = h->Cx;
        h->Cx = h->Cy;
        h->Cy = temp;
    }
    ev->cx_id = h->Cx->id;
    ev->cy_id = h->Cy->id;
    handleAgglomerationEvent(h, h->Cx, h->Cy, ev);
    return 1;
}

void nno_set_relaxed(nno_handle* h, uint64_t seed, int32_t min_active) {
    h->relaxed = 1;
    h->relaxed_min = min_active > 0 ? min_active : 1024;
    jr_set_seed(h, seed);
}
int32_t nno_java_random_next_int(uint64_t* state, int32_t bound, int32_t init_with_seed) {  /* (known-answer tests) */
    nno_handle tmp;
    if (init_with_seed) { jr_set_seed(&tmp, *state); } else tmp.rng = *state;
    int32_t r = jr_next_int(&tmp, bound);
    *state = tmp.rng;
    return r;
}

int32_t nno_num_active(const nno_handle* h) { return h->num_active; }
int32_t nno_num_clusters(const nno_handle* h) { return h->num_clusters; }
int32_t nno_num_nodes(const nno_handle* h) { return h->num_nodes; }
const double* nno_matrix(const nno_handle* h) { return h->D; }

void nno_get_nodes(const nno_handle* h, int32_t* id, int32_t* distID, int32_t* nbr_id, double* Sx) {
    for (int i = 0; i < h->num_active && i < h->ntax; i++) {
        const NetNode* p = h->netNodes[i];
        if (!p) { /* only after the special finish: netNodes[3] is null (:643) */
            if (id) id[i] = 0;
            if (distID) distID[i] = -1;
            if (nbr_id) nbr_id[i] = 0;
            if (Sx) Sx[i] = 0.0;
            continue;
        }
        if (id) id[i] = p->id;
        if (distID) distID[i] = p->distID;
        if (nbr_id) nbr_id[i] = p->nbr ? p->nbr->id : 0;
        if (Sx) Sx[i] = p->Sx;
    }
}

/* NetMakerOriginal.java:246-325 expandNodes; n <= 3 identity (:133-140) */
int32_t nno_expand(nno_handle* h, int32_t* ordering) {
    int ntax = h->ntax;
    if (ntax <= 3) {
        for (int i = 0; i <= ntax; i++) ordering[i] = i;
        return 0;
    }
    NetNode *x, *y, *z, *u, *v, *a;
    x = h->netNodes[0];
    y = h->netNodes[1];
    z = h->netNodes[2];
    x->next = y; y->next = z; z->next = x;
    x->prev = z; y->prev = x; z->prev = y;
    while (h->amalgs_top > 0) {
        u = h->amalgs[--h->amalgs_top];
        v = u->nbr;
        x = u->ch1;
        y = u->ch2;
        z = v->ch2;
        if (v != u->next) {
            NetNode* tmp = u; u = v; v = tmp;
            tmp = x; x = z; z = tmp;
        }
        x->prev = u->prev;
        x->prev->next = x;
        x->next = y;
        y->prev = x;
        y->next = z;
        z->prev = y;
        z->next = v->next;
        z->next->prev = z;
    }
    while (x->id != 1) x = x->next;
    a = x;
    int t = 0;
    ordering[0] = 0; /* Java int[] default */
    do {
        ordering[++t] = a->id;
        a = a->next;
    } while (a != x);
    return 0;
}

int32_t nno_run(const double* D, int32_t n, int32_t threads, int32_t* order_out,
                nno_event* events_out, int64_t max_events, int64_t* n_events,
                int64_t* sum_entries) {
    nno_handle* h = nno_create(D, n, threads);
    if (!h) return -1;
    int64_t k = 0, se = 0;
    nno_event ev;
    while (nno_step(h, &ev) == 1) {
        if (events_out && k < max_events) events_out[k] = ev;
        se += ev.entries;
        k++;
    }
    if (n_events) *n_events = k;
    if (sum_entries) *sum_entries = se;
    int32_t rc = nno_expand(h, order_out);
    nno_destroy(h);
    return rc;
}

/* Whole Relaxed run (FastNN.java:329-338 with additive == false). */
int32_t nno_run_relaxed(const double* D, int32_t n, uint64_t seed, int32_t min_active, int32_t* order_out,
                        nno_event* events_out, int64_t max_events, int64_t* n_events) {
    nno_handle* h = nno_create(D, n, 1);
    if (!h) return -1;
    nno_set_relaxed(h, seed, min_active);
    int64_t k = 0;
    nno_event ev;
    int32_t r;
    while ((r = nno_step(h, &ev)) == 1) {
        if (events_out && k < max_events) events_out[k] = ev;
        k++;
    }
    if (n_events) *n_events = k;
    if (r < 0) { nno_destroy(h); return r; }
    int32_t rc = nno_expand(h, order_out);
    nno_destroy(h);
    return rc;
}

static inline uint64_t splitmix64_at(uint64_t seed, uint64_t k) {
    /* k-th output (k = 0,1,...) of SplitMix64 seeded with `seed` */
    uint64_t z = seed + (k + 1) * 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

void nno_synth(double* D, int32_t n, uint64_t seed, int32_t dist) {
    /* upper triangle, row-major generator order (rows are independent: the k-th value
     * is a pure function of k) ... */
#pragma omp parallel for schedule(dynamic, 64)
    for (int i = 0; i < n; i++) {
        uint64_t k = (uint64_t)i * (uint64_t)n - (uint64_t)i * ((uint64_t)i + 1) / 2;
        D[(size_t)i * n + i] = 0.0;
        for (int j = i + 1; j < n; j++, k++) {
            double u = (double)(splitmix64_at(seed, k) >> 11) * 0x1.0p-53;
            double d;
            if (dist == 1) d = (double)((int64_t)(u * 1e4) + 1) / 1e4;
            else d = u + 0x1.0p-10;
            D[(size_t)i * n + j] = d;
        }
    }
    /* ... then mirrored into the lower triangle in 64 x 64 blocks */
#pragma omp parallel for schedule(dynamic, 1)
    for (int bi = 0; bi < n; bi += 64)
        for (int bj = 0; bj <= bi; bj += 64)
            for (int i = bi; i < bi + 64 && i < n; i++)
                for (int j = bj; j < bj + 64 && j < i; j++)
                    D[(size_t)i * n + j] = D[(size_t)j * n + i];
}
