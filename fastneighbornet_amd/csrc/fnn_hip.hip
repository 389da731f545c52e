// fnn_hip.hip -- gfx950 kernels and the C ABI of libfastnn_hip.so.
//
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -shared
// (-ffp-contract=off on host AND device: every fp64 operation of the path is
// rounded once, in the reference's source order; no FMA anywhere).
//
// Kernel inventory (one agglomeration event = a fixed launch sequence of HipBackend::launch_event;
// all control state lives in device memory so the host never has to wait for a decision):
//   k_track     lookahead windows (fnn_core.h "Lookahead"): the event's minimum from the tracked pairs
//               + one sweep of the newest cluster's rows, exact fp64, when the open window can certify
//               it (no scan then); workgroup 0 computes the previous event's exact u.Sx beside it; the
//               workgroup that arrives last runs the decide step (below) for a window event
//   k_scan      all-pairs Q-criterion argmin over the lower triangle of the live
//               m x m block (NeighborNetCanonical.java:151-178).  HBM-bound: reads
//               each live matrix entry once, 16 B per lane, 1 KiB per wave-load.
//   k_screen    events with >= 2048 live nodes that scan: the same scan as a bracketing pass over the
//               bf16 copy of the matrix (2 B per entry); k_emit: the pairs a new window tracks;
//               k_resolve: exact fp64 rescan of the few 32 x 512 units that can hold the minimum
//   k_decide    events that scanned: reduces the per-workgroup (or all ranks') candidate records, then the
//               decide step: Cx / Cy (NetMakerOriginal.java:376-380), the 4-candidate choice (:413-452)
//               certified from the maintained row sums T, else from the <= 4 exact sequential ComputeRx
//               sums (:549-561); the merge plan (:462-488) and its symbolic replay
//   k_update    fused: subtractClusterDistance x2 per node (:455-461, 681-696), the net effect of
//               the plan's micro-ops (agg3way row/column rewrite :653-656, slot swaps / moves)
//               and updateClusterDistances' per-node part (:520-531); closes the event when the
//               exact u.Sx sum is deferred to the next k_track
//   k_finalize  exact sequential u.Sx sum (:532), event log, loop condition (:339) - stepping API and
//               configurations without windows; k_chain_flush: a deferred sum before the host looks
//   k_relaxed   -mode Relaxed: the randomised search for mutual row minima (NeighborNetLocal.java:88-264)
//   k_init, k_synth, k_unpack / k_mirror, k_validate, k_prep_screen, k_stream: set-up, ingest, probes
// Several GPUs: every rank screens 1/world of the tiles of a base scan; the candidate records and the new
// window's pairs are all-gathered (k_reduce_local + ncclAllGather, then k_merge) before k_decide.
#include <hip/hip_runtime.h>

#include <dlfcn.h>

#include <cmath>
#include <new>

#include "fnn_chain.h"
#include "fnn_engine.h"

namespace fnn {

struct fnn_nccl_id { char internal[FNN_COMM_ID_BYTES]; };  // == ncclUniqueId (rccl.h:43)

constexpr int SCAN_TW = 512;  // columns per scan tile (256 threads x 2)
constexpr int SCAN_TH = 32;   // rows per scan tile
constexpr int SCAN_THREADS = 256;

// ------------------------------------------------------------------ reductions
// minimum of the wave on the total order (Q, i, j); valid in every lane.  Only (q, key) are reduced; the slots of the
// winning pair are read from a lane that contributed it (a key names one pair).  The reduction runs on DPP moves
// (row_shr 1, 2, 4, 8, then row_bcast 15 / 31: the result lands in lane 63) instead of __shfl_down, which hipcc lowers to
// ds_bpermute: ~100 cycles of LDS-crossbar latency per dependent step, and k_track does four such reductions on its
// critical path.
// A workgroup's minimum handed to the workgroup that arrives last: three 8-byte write-through stores (agent scope:
// they leave the XCD's L2) by ONE lane, which then waits for their acknowledgement and counts its arrival; the reader
// loads them past its L1 after its own arrival returned last.  No cache write-back or invalidate on the event chain.
static_assert(sizeof(Cand) == 24, "rec_publish / rec_fetch move a Cand as three 8-byte words");
// The hand-over is checked: a fourth word, a function of the record's three words and of the event number, goes out
// with them; the reader recomputes it.  A record that is stale or torn (not observed; the write-through form is measured
// behaviour of this part, not an architectural promise) makes the reader fence and read again, and if that does not
// help the window gives the event up (it scans) - slower, never wrong.
__device__ inline uint64_t rec_check_word(const Cand& c, uint32_t tag) {
    return __builtin_bit_cast(uint64_t, c.q) ^ (c.key * 0x9E3779B97F4A7C15ULL) ^
           ((uint64_t)(uint32_t)c.si | ((uint64_t)(uint32_t)c.sj << 32)) ^ (((uint64_t)tag + 1ULL) * 0xBF58476D1CE4E5B9ULL);
}
__device__ inline void rec_publish(Cand* p, const Cand& c, uint64_t* chk, uint32_t tag) {
    __hip_atomic_store(chk, rec_check_word(c, tag), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    uint64_t* w = (uint64_t*)p;
    __hip_atomic_store(w + 0, __builtin_bit_cast(uint64_t, c.q), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(w + 1, c.key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(w + 2, (uint64_t)(uint32_t)c.si | ((uint64_t)(uint32_t)c.sj << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ inline Cand rec_fetch(const Cand* p) {
    uint64_t* w = (uint64_t*)p;
    const uint64_t a = __hip_atomic_load(w + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const uint64_t b = __hip_atomic_load(w + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const uint64_t x = __hip_atomic_load(w + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    Cand c;
    c.q = __builtin_bit_cast(double, a);
    c.key = b;
    c.si = (int)(uint32_t)x;
    c.sj = (int)(uint32_t)(x >> 32);
    return c;
}

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ void dpp_pull(double q, uint64_t key, double& oq, uint64_t& ok) {
    const uint64_t qb = __builtin_bit_cast(uint64_t, q);
    const int ql = (int)(uint32_t)qb, qh = (int)(uint32_t)(qb >> 32), kl = (int)(uint32_t)key, kh = (int)(uint32_t)(key >> 32);
    // (lanes without a source - out of the row, or rows not selected by the mask - keep their own value: no change under min)
    const uint32_t oql = (uint32_t)__builtin_amdgcn_update_dpp(ql, ql, CTRL, ROW_MASK, 0xf, false);
    const uint32_t oqh = (uint32_t)__builtin_amdgcn_update_dpp(qh, qh, CTRL, ROW_MASK, 0xf, false);
    const uint32_t okl = (uint32_t)__builtin_amdgcn_update_dpp(kl, kl, CTRL, ROW_MASK, 0xf, false);
    const uint32_t okh = (uint32_t)__builtin_amdgcn_update_dpp(kh, kh, CTRL, ROW_MASK, 0xf, false);
    oq = __builtin_bit_cast(double, ((uint64_t)oqh << 32) | oql);
    ok = ((uint64_t)okh << 32) | okl;
}
__device__ __forceinline__ Cand wave_reduce(Cand c) {
    double q = c.q;
    uint64_t key = c.key;
#define FNN_DPP_STEP(CTRL, MASK) do { double oq; uint64_t ok; dpp_pull<CTRL, MASK>(q, key, oq, ok); \
        if (oq < q || (oq == q && ok < key)) { q = oq; key = ok; } } while (0)
    FNN_DPP_STEP(0x111, 0xf);  // row_shr:1
    FNN_DPP_STEP(0x112, 0xf);  // row_shr:2
    FNN_DPP_STEP(0x114, 0xf);  // row_shr:4
    FNN_DPP_STEP(0x118, 0xf);  // row_shr:8   -> lane 15 of every row of 16 holds the row's minimum
    FNN_DPP_STEP(0x142, 0xa);  // row_bcast:15 into rows 1 and 3
    FNN_DPP_STEP(0x143, 0xc);  // row_bcast:31 into rows 2 and 3 -> lane 63 holds the wave's minimum
#undef FNN_DPP_STEP
    Cand r;
    const uint64_t qb = __builtin_bit_cast(uint64_t, q);
    const uint32_t rql = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)qb, 63), rqh = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(qb >> 32), 63);
    const uint32_t rkl = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)key, 63), rkh = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(key >> 32), 63);
    r.q = __builtin_bit_cast(double, ((uint64_t)rqh << 32) | rql);
    r.key = ((uint64_t)rkh << 32) | rkl;
    const unsigned long long src = __ballot(c.key == r.key);
    const int l = src ? (int)__builtin_ctzll(src) : 0;
    r.si = __builtin_amdgcn_readlane(c.si, l);
    r.sj = __builtin_amdgcn_readlane(c.sj, l);
    return r;
}

template <int NWAVES>
__device__ __forceinline__ Cand block_reduce(Cand c, Cand* sh) {
    c = wave_reduce(c);
    int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) sh[w] = c;
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int k = 1; k < NWAVES; k++)
            if (cand_better(sh[k], c)) c = sh[k];
    }
    return c;  // valid in thread 0
}

// ------------------------------------------------------------------ k_scan
// Tiles of SCAN_TH rows x SCAN_TW columns cover the lower triangle of the live m x m block.
// Row tiles come in bands of R = SCAN_TW / SCAN_TH; every row tile of band g owns g + 1 column
// tiles (fnn_core.h: tri_tile_count / tri_tile_decode).  A fixed-size grid strides over the
// linear tile index (no empty workgroups, at most gridDim.x records for k_decide).
constexpr int SCAN_R = SCAN_TW / SCAN_TH;

__host__ __device__ inline int scan_tile_count(int m) { return tri_tile_count(m, SCAN_TH, SCAN_R); }

typedef double fnn_v2f64 __attribute__((ext_vector_type(2)));
typedef float fnn_v4f32 __attribute__((ext_vector_type(4)));
template <bool NT>
__device__ __forceinline__ double2 ld16(const double* p) {
    if (NT) {
        const fnn_v2f64 v = __builtin_nontemporal_load(reinterpret_cast<const fnn_v2f64*>(p));
        return make_double2(v.x, v.y);
    }
    return *reinterpret_cast<const double2*>(p);
}
template <bool NT>
__device__ __forceinline__ float4 ld16f(const float* p) {
    if (NT) {
        const fnn_v4f32 v = __builtin_nontemporal_load(reinterpret_cast<const fnn_v4f32*>(p));
        return make_float4(v.x, v.y, v.z, v.w);
    }
    return *reinterpret_cast<const float4*>(p);
}

// exact scan of rows [rbase, rbase + 32) at the two columns c0, c0 + 1 of this thread
template <bool NT>
__device__ __forceinline__ void scan_rows_exact(const Dev& d, int rbase, int c0, int m, int twoP, double cm2, Cand& best) {
    if (!(c0 < m && c0 <= rbase + SCAN_TH - 2)) return;
    const double2 sxc = *reinterpret_cast<const double2*>(d.Sx + c0);
    const int2 pc = *reinterpret_cast<const int2*>(d.spos + c0);
    const double* colbase = d.D + c0;
#pragma unroll 1
    for (int half = 0; half < SCAN_TH / 16; half++) {
        const int rb = rbase + 16 * half;
        if (rb >= m) break;
        double2 a[8], b[8];
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const int r0 = rb + 2 * k;  // r0 + 1 < nrows (padded), c0 + 1 < ld: always in bounds
            a[k] = ld16<NT>(colbase + (int64_t)r0 * d.ld);
            b[k] = ld16<NT>(colbase + (int64_t)(r0 + 1) * d.ld);
        }
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const int r0 = rb + 2 * k;
            if (c0 <= r0) {
                const double2 sxr = *reinterpret_cast<const double2*>(d.Sx + r0);
                const int2 pr = *reinterpret_cast<const int2*>(d.spos + r0);
                scan_micro(r0, c0, m, twoP, cm2, a[k].x, a[k].y, b[k].x, b[k].y,
                           sxr.x, sxr.y, pr.x, pr.y, sxc.x, sxc.y, pc.x, pc.y, best);
            }
        }
    }
}

template <bool NT>
__global__ __launch_bounds__(SCAN_THREADS) void k_scan(Dev d) {
    __shared__ Cand sh[SCAN_THREADS / 64];
    const State* st = d.st;
    Cand best;
    best = cand_none();
    if (st->la_hit) return;  // the lookahead window already holds this event's minimum (recs[0])
    if (st->rl_active) return;  // Relaxed mode: k_relaxed has found this event's pair (recs[0])
    if (!st->done) {
        const int m = st->m;
        const int twoP = 2 * st->P;
        const double cm2 = (double)st->c - 2.0;
        const int ntiles = scan_tile_count(m);
        // several GPUs: rank r of `world` takes the tiles with index = r (mod world)
        for (int t = blockIdx.x * d.world + d.rank; t < ntiles; t += gridDim.x * d.world) {
            int rt, ct;
            tri_tile_decode(t, SCAN_R, rt, ct);
            scan_rows_exact<NT>(d, rt * SCAN_TH, ct * SCAN_TW + 2 * (int)threadIdx.x, m, twoP, cm2, best);
        }
    }
    best = block_reduce<SCAN_THREADS / 64>(best, sh);
    if (threadIdx.x == 0) d.recs[blockIdx.x] = best;
}

// ------------------------------------------------------------------ bf16 screening (fnn_core.h: screen_micro)
// k_screen streams the bf16 copy (2 B per entry): tiles of 32 rows x 2048 columns, a thread owns
// 8 adjacent columns (one 16-byte load per row), a wave owns one 32 x 512 "unit" and records the
// minima of the lower and upper bounds of Q over it.  k_resolve turns the records into the list
// of units that can hold the true minimum and rescans exactly those in fp64 with the exact body.
constexpr int SCR_R = SCR_TW / SCR_TH;

typedef unsigned int fnn_v4u32 __attribute__((ext_vector_type(4)));
template <bool NT>
__device__ __forceinline__ uint4 ld16h(const uint16_t* p) {
    if (NT) {
        const fnn_v4u32 v = __builtin_nontemporal_load(reinterpret_cast<const fnn_v4u32*>(p));
        return make_uint4(v.x, v.y, v.z, v.w);
    }
    return *reinterpret_cast<const uint4*>(p);
}
__device__ __forceinline__ float bf_lo(unsigned w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf_hi(unsigned w) { return __uint_as_float(w & 0xFFFF0000u); }

// bracket of one cluster pair from the mean dm of its entries and the mean am of their magnitudes
__device__ __forceinline__ void brk_fast(float dm, float am, float cm2, float cm2k, float sp, float sq, float& lb, float& ub) {
    const float q = __builtin_fmaf(cm2, dm, -sp) - sq;
    const float e = cm2k * am;
    lb = __builtin_fminf(lb, q - e);
    ub = __builtin_fminf(ub, q + e);
}

// Tile classes with a branch-free body: strictly below the diagonal, entirely inside the live
// block, and with all rows / all columns of one kind (two-node clusters or singletons).
enum { SCR_GENERIC = 0, SCR_PP = 1, SCR_SP = 2, SCR_SS = 3 };

constexpr int SCR_BATCH = 4;  // row pairs (2 x 16-byte loads each) a thread keeps in flight

template <bool NT, int CLS>
__device__ __forceinline__ void screen_tile_fast(const Dev& d, int rbase, int c0, float cm2, float cm2k, float (&lb)[4], float (&ub)[4]) {
    float sxc[8];
#pragma unroll
    for (int k = 0; k < 8; k += 2) {
        const double2 sv = *reinterpret_cast<const double2*>(d.Sx + c0 + k);
        sxc[k] = (float)sv.x;
        sxc[k + 1] = (float)sv.y;
    }
    const uint16_t* colbase = d.H + c0;
#pragma unroll 1
    for (int part = 0; part < SCR_TH / (2 * SCR_BATCH); part++) {
        const int rb = rbase + 2 * SCR_BATCH * part;
        uint4 a[SCR_BATCH], b[SCR_BATCH];
#pragma unroll
        for (int k = 0; k < SCR_BATCH; k++) {
            a[k] = ld16h<NT>(colbase + (int64_t)(rb + 2 * k) * d.ldh);
            b[k] = ld16h<NT>(colbase + (int64_t)(rb + 2 * k + 1) * d.ldh);
        }
#pragma unroll
        for (int k = 0; k < SCR_BATCH; k++) {
            const double2 sxr = *reinterpret_cast<const double2*>(d.Sx + rb + 2 * k);
            const float s0 = (float)sxr.x, s1 = (float)sxr.y;
            const unsigned aw[4] = {a[k].x, a[k].y, a[k].z, a[k].w};
            const unsigned bw[4] = {b[k].x, b[k].y, b[k].z, b[k].w};
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const float e00 = bf_lo(aw[j]), e01 = bf_hi(aw[j]), e10 = bf_lo(bw[j]), e11 = bf_hi(bw[j]);
                if (CLS == SCR_PP) {
                    brk_fast((((e00 + e01) + e10) + e11) * 0.25f,
                             (((__builtin_fabsf(e00) + __builtin_fabsf(e01)) + __builtin_fabsf(e10)) + __builtin_fabsf(e11)) * 0.25f,
                             cm2, cm2k, s0, sxc[2 * j], lb[j], ub[j]);
                } else if (CLS == SCR_SP) {
                    brk_fast((e00 + e01) * 0.5f, (__builtin_fabsf(e00) + __builtin_fabsf(e01)) * 0.5f, cm2, cm2k, s0, sxc[2 * j], lb[j], ub[j]);
                    brk_fast((e10 + e11) * 0.5f, (__builtin_fabsf(e10) + __builtin_fabsf(e11)) * 0.5f, cm2, cm2k, s1, sxc[2 * j], lb[j], ub[j]);
                } else {
                    brk_fast(e00, __builtin_fabsf(e00), cm2, cm2k, s0, sxc[2 * j], lb[j], ub[j]);
                    brk_fast(e01, __builtin_fabsf(e01), cm2, cm2k, s0, sxc[2 * j + 1], lb[j], ub[j]);
                    brk_fast(e10, __builtin_fabsf(e10), cm2, cm2k, s1, sxc[2 * j], lb[j], ub[j]);
                    brk_fast(e11, __builtin_fabsf(e11), cm2, cm2k, s1, sxc[2 * j + 1], lb[j], ub[j]);
                }
            }
        }
    }
}

// The same for matrices without negative entries (the usual case: distances): mean|h| == mean h,
// the brackets are affine in the sum of the entries (coefficients screen_k1 / screen_k2, with
// the 1/2, 1/4 of the mean folded in), and the per-column term -Sq is applied once per tile
// (x -> fl(x - Sq) is monotone, so it commutes with the running minimum).  ~5 instead of ~9
// VALU instructions per entry; the pass is VALU-bound (SQ_ACTIVE_INST_VALU ~ 80 % of a SIMD).
template <bool NT, int CLS>
__device__ __forceinline__ void screen_tile_nonneg(const Dev& d, int rbase, int c0, float k1, float k2, float (&lb)[4], float (&ub)[4]) {
    constexpr int NA = (CLS == SCR_SS) ? 8 : 4;
    float al[NA], au[NA];
#pragma unroll
    for (int i = 0; i < NA; i++) { al[i] = __builtin_inff(); au[i] = __builtin_inff(); }
    const float f = (CLS == SCR_PP) ? 0.25f : (CLS == SCR_SP) ? 0.5f : 1.0f;
    const float a1 = k1 * f, a2 = k2 * f;  // exact scalings
    const uint16_t* colbase = d.H + c0;
#pragma unroll 1
    for (int part = 0; part < SCR_TH / (2 * SCR_BATCH); part++) {
        const int rb = rbase + 2 * SCR_BATCH * part;
        uint4 a[SCR_BATCH], b[SCR_BATCH];
#pragma unroll
        for (int k = 0; k < SCR_BATCH; k++) {
            a[k] = ld16h<NT>(colbase + (int64_t)(rb + 2 * k) * d.ldh);
            b[k] = ld16h<NT>(colbase + (int64_t)(rb + 2 * k + 1) * d.ldh);
        }
#pragma unroll
        for (int k = 0; k < SCR_BATCH; k++) {
            const double2 sxr = *reinterpret_cast<const double2*>(d.Sx + rb + 2 * k);
            const float s0 = (float)sxr.x, s1 = (float)sxr.y;
            const unsigned aw[4] = {a[k].x, a[k].y, a[k].z, a[k].w};
            const unsigned bw[4] = {b[k].x, b[k].y, b[k].z, b[k].w};
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const float e00 = bf_lo(aw[j]), e01 = bf_hi(aw[j]), e10 = bf_lo(bw[j]), e11 = bf_hi(bw[j]);
                if (CLS == SCR_PP) {
                    const float sum = ((e00 + e01) + e10) + e11;
                    al[j] = __builtin_fminf(al[j], __builtin_fmaf(a1, sum, -s0));
                    au[j] = __builtin_fminf(au[j], __builtin_fmaf(a2, sum, -s0));
                } else if (CLS == SCR_SP) {
                    const float t0 = e00 + e01, t1 = e10 + e11;
                    al[j] = __builtin_fminf(al[j], __builtin_fminf(__builtin_fmaf(a1, t0, -s0), __builtin_fmaf(a1, t1, -s1)));
                    au[j] = __builtin_fminf(au[j], __builtin_fminf(__builtin_fmaf(a2, t0, -s0), __builtin_fmaf(a2, t1, -s1)));
                } else {
                    al[2 * j] = __builtin_fminf(al[2 * j], __builtin_fminf(__builtin_fmaf(a1, e00, -s0), __builtin_fmaf(a1, e10, -s1)));
                    au[2 * j] = __builtin_fminf(au[2 * j], __builtin_fminf(__builtin_fmaf(a2, e00, -s0), __builtin_fmaf(a2, e10, -s1)));
                    al[2 * j + 1] = __builtin_fminf(al[2 * j + 1], __builtin_fminf(__builtin_fmaf(a1, e01, -s0), __builtin_fmaf(a1, e11, -s1)));
                    au[2 * j + 1] = __builtin_fminf(au[2 * j + 1], __builtin_fminf(__builtin_fmaf(a2, e01, -s0), __builtin_fmaf(a2, e11, -s1)));
                }
            }
        }
    }
    // the column terms, once per tile
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const double2 sv = *reinterpret_cast<const double2*>(d.Sx + c0 + 2 * j);
        const float q0 = (float)sv.x, q1 = (float)sv.y;
        if (CLS == SCR_SS) {
            lb[j] = __builtin_fminf(al[2 * j] - q0, al[2 * j + 1] - q1);
            ub[j] = __builtin_fminf(au[2 * j] - q0, au[2 * j + 1] - q1);
        } else {
            lb[j] = al[j] - q0;
            ub[j] = au[j] - q0;
        }
    }
}

// Emission pass of a lookahead window (fnn_core.h "Lookahead"), k_emit: k_screen has marked, per
// unit (32 rows x 512 columns, one wave of a screening tile), the lanes whose 8 columns hold a
// pair with a lower bound under the window's threshold.  Every such 32 x 8 block is walked again
// and the pairs under the threshold are appended, as the two representatives' node ids, to the
// tracked list.
constexpr int EMIT_LDS = 2048;  // pairs a workgroup of k_emit collects before it reserves their place in the list
__global__ __launch_bounds__(256) void k_emit(Dev d) {
    // (one device-scope atomic on a single address costs ~50 ns on this multi-XCD part: the pairs
    //  are collected per workgroup in LDS and get their place in the list with ONE atomic add)
    __shared__ int lcount, lbase;
    __shared__ int2 lbuf[EMIT_LDS];
    State* st = d.st;
    if (st->done || st->la_hit || !st->la_emit) return;
    // (several ranks: a rank walks the units of its own tiles and collects its pairs in its exchange block)
    const int m = st->m, twoP = 2 * st->P, pcap = d.wx ? wx_pair_cap(d.world) : st->la_pcap;
    int32_t* const outp = d.wx ? reinterpret_cast<int32_t*>(d.wsend + wx_pairs_off()) : d.tpairs;
    const float k1 = screen_k1(*st), k2 = screen_k2(*st), tp = st->la_theta_pred;
    const int nunits = 4 * tri_tile_count(m, SCR_TH, SCR_R);
    const int lane = threadIdx.x & 63;
    const int wave = (int)blockIdx.x * 4 + ((int)threadIdx.x >> 6), nwaves = (int)gridDim.x * 4;
    if (threadIdx.x == 0) lcount = 0;
    __syncthreads();
    auto emit = [&](int32_t rs, int32_t cs, float lb) {
        if (!(lb <= tp)) return;
        const int i = atomicAdd(&lcount, 1);
        if (i < EMIT_LDS) lbuf[i] = make_int2(rs, cs);
    };
    // `per` units per wave and step (a power of two, 4..64: few enough that all waves of the grid get
    // some when the live matrix is small): lane l < per fetches one unit's mask; the marked 32 x 8
    // blocks of these units are then dealt out evenly, one block per lane and round (a node with an
    // extreme row sum marks a whole row of blocks in ONE unit)
    // The units of a step are dealt to the waves ROUND-ROBIN (unit (ustep + lane) nwaves + wave): the units of one row of tiles are
    // consecutive, and a node with an extreme row sum marks all 64 blocks of every unit of its row - given to one wave as a
    // contiguous chunk (round 3) that wave walked 4096 blocks in 64 rounds while the others idled: 98 us at 32768 live nodes.
    int per = 4;
    while (per < 64 && per * nwaves < nunits) per <<= 1;
    for (int ustep = 0; ustep * nwaves < nunits; ustep += per) {
        const int myu = (ustep + lane) * nwaves + wave;
        unsigned long long mymask = 0;
        if (lane < per && myu < nunits && (myu >> 2) % d.world == d.rank && d.srec[myu] <= tp) mymask = d.shit[myu];
        int incl = __builtin_popcountll(mymask);
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int o = __shfl_up(incl, off, 64);
            if (lane >= off) incl += o;
        }
        // (an item is a QUARTER of a marked block - 8 of its 32 rows, one batch of loads: four lanes share a block, so that a wave's
        //  ~14 blocks keep 56 lanes busy for one round trip instead of 14 lanes for four)
        const int T = __shfl(incl, 63, 64);
        for (int t0 = 0; t0 < 4 * T; t0 += 64) {
            const int t = (t0 + lane) >> 2, part = (t0 + lane) & 3;
            // owner = first lane whose inclusive prefix exceeds t (binary search over the lanes)
            int lo = 0, hi = 63;
#pragma unroll
            for (int it = 0; it < 6; it++) {
                const int mid = (lo + hi) >> 1;
                const int pm = __shfl(incl, mid, 64);
                if (pm > t) hi = mid; else lo = mid + 1;
            }
            const int owner = lo > 63 ? 63 : lo;
            const int oincl = __shfl(incl, owner, 64);
            unsigned long long omask = (unsigned long long)__shfl((long long)mymask, owner, 64);
            if (t >= T) continue;
            int k = t - (oincl - __builtin_popcountll(omask));  // index of the block among the owner's marked ones
            while (k-- > 0) omask &= omask - 1;
            const int l = __builtin_ctzll(omask);
            const int u = (ustep + owner) * nwaves + wave;
            int rt, ct;
            tri_tile_decode(u >> 2, SCR_R, rt, ct);
            const int rbase = rt * SCR_TH, c0 = ct * SCR_TW + (u & 3) * SCR_UW + 8 * l;
            if (c0 >= m) continue;
            float sxc[8];
#pragma unroll
            for (int q = 0; q < 8; q += 2) {
                const double2 sv = *reinterpret_cast<const double2*>(d.Sx + c0 + q);
                sxc[q] = (float)sv.x;
                sxc[q + 1] = (float)sv.y;
            }
            const uint16_t* colbase = d.H + c0;
            static_assert(SCR_TH / 8 == 4, "an item is a quarter of a block");
            {  // 4 row pairs (8 x 16-byte loads) in flight
                const int rb = rbase + 8 * part;
                if (rb >= m) continue;
                uint4 a[4], b[4];
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    a[q] = ld16h<false>(colbase + (int64_t)(rb + 2 * q) * d.ldh);
                    b[q] = ld16h<false>(colbase + (int64_t)(rb + 2 * q + 1) * d.ldh);
                }
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const int r0 = rb + 2 * q;
                    const double2 sxr = *reinterpret_cast<const double2*>(d.Sx + r0);
                    const float s0 = (float)sxr.x, s1 = (float)sxr.y;
                    const unsigned aw[4] = {a[q].x, a[q].y, a[q].z, a[q].w};
                    const unsigned bw[4] = {b[q].x, b[q].y, b[q].z, b[q].w};
                    Brk dummy;
                    dummy.lb = __builtin_inff();
                    dummy.ub = __builtin_inff();
#pragma unroll
                    for (int j = 0; j < 4; j++)
                        screen_micro_nn(r0, c0 + 2 * j, m, twoP, k1, k2, bf_lo(aw[j]), bf_hi(aw[j]), bf_lo(bw[j]), bf_hi(bw[j]),
                                        s0, s1, sxc[2 * j], sxc[2 * j + 1], dummy, emit);
                }
            }
        }
    }
    __syncthreads();
    const int total = lcount;
    if (total == 0) return;
    // a workgroup that found more than it can hold reports a count beyond the capacity, so that the
    // window is not opened (la_close_base treats it as an overflow)
    if (threadIdx.x == 0) lbase = atomicAdd(d.lacnt, total > EMIT_LDS ? pcap + 1 : total);
    __syncthreads();
    if (total > EMIT_LDS) return;
    const int base = lbase;
    for (int i = threadIdx.x; i < total; i += 256) {
        const int idx = base + i;
        if (idx < pcap) {
            const int2 pr = lbuf[i];
            la_record(d, outp + LA_REC_INTS * (int64_t)idx, pr.x, pr.y, twoP);
        }
    }
}

// SCHED only separates, by kernel name, the host-scheduled base scans of the lookahead windows
// (always real work) from the launches that run only when a window failed (mostly no-ops).
template <bool NT, bool SCHED>
__global__ __launch_bounds__(256) void k_screen(Dev d) {
    __shared__ float shl[4], shu[4];
    const State* st = d.st;
    if (st->done || st->la_hit) return;
    const int m = st->m;
    const int twoP = 2 * st->P;
    const float cm2 = (float)((double)st->c - 2.0);
    const float cm2k = screen_cm2k(*st);
    const float k1 = screen_k1(*st), k2 = screen_k2(*st);
    const bool nonneg = st->nonneg != 0;
    const float tp = st->la_emit ? st->la_theta_pred : -__builtin_inff();
    const int ntiles = tri_tile_count(m, SCR_TH, SCR_R);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    float* lbrec = d.srec;
    float* ubrec = d.srec + 4 * (size_t)ntiles;
    float* lbt = d.stile;
    float* ubt = d.stile + ntiles;
    for (int t = blockIdx.x * d.world + d.rank; t < ntiles; t += gridDim.x * d.world) {
        int rt, ct;
        tri_tile_decode(t, SCR_R, rt, ct);
        const int rbase = rt * SCR_TH, cbase = ct * SCR_TW;
        const int c0 = cbase + 8 * (int)threadIdx.x;
        float lbv[4], ubv[4];  // independent running minima per column pair
#pragma unroll
        for (int j = 0; j < 4; j++) { lbv[j] = __builtin_inff(); ubv[j] = __builtin_inff(); }
        int cls = SCR_GENERIC;
        if (cbase + SCR_TW <= rbase && rbase + SCR_TH <= m) {  // below the diagonal, inside the live block
            if (rbase + SCR_TH <= twoP) cls = SCR_PP;
            else if (cbase >= twoP) cls = SCR_SS;
            else if (rbase >= twoP && cbase + SCR_TW <= twoP) cls = SCR_SP;
        }
        if (cls != SCR_GENERIC && nonneg) {
            if (cls == SCR_PP) screen_tile_nonneg<NT, SCR_PP>(d, rbase, c0, k1, k2, lbv, ubv);
            else if (cls == SCR_SS) screen_tile_nonneg<NT, SCR_SS>(d, rbase, c0, k1, k2, lbv, ubv);
            else screen_tile_nonneg<NT, SCR_SP>(d, rbase, c0, k1, k2, lbv, ubv);
        }
        else if (cls == SCR_PP) screen_tile_fast<NT, SCR_PP>(d, rbase, c0, cm2, cm2k, lbv, ubv);
        else if (cls == SCR_SS) screen_tile_fast<NT, SCR_SS>(d, rbase, c0, cm2, cm2k, lbv, ubv);
        else if (cls == SCR_SP) screen_tile_fast<NT, SCR_SP>(d, rbase, c0, cm2, cm2k, lbv, ubv);
        else if (c0 < m && c0 <= rbase + SCR_TH - 2) {
            Brk bk[4];
#pragma unroll
            for (int j = 0; j < 4; j++) { bk[j].lb = __builtin_inff(); bk[j].ub = __builtin_inff(); }
            float sxc[8];
#pragma unroll
            for (int k = 0; k < 8; k += 2) {
                const double2 sv = *reinterpret_cast<const double2*>(d.Sx + c0 + k);
                sxc[k] = (float)sv.x;
                sxc[k + 1] = (float)sv.y;
            }
            const uint16_t* colbase = d.H + c0;
#pragma unroll 1
            for (int part = 0; part < SCR_TH / (2 * SCR_BATCH); part++) {
                const int rb = rbase + 2 * SCR_BATCH * part;
                if (rb >= m) break;
                uint4 a[SCR_BATCH], b[SCR_BATCH];
#pragma unroll
                for (int k = 0; k < SCR_BATCH; k++) {
                    const int r0 = rb + 2 * k;  // rows < nrows (padded), c0 + 7 < ld (ld padded to 2048)
                    a[k] = ld16h<NT>(colbase + (int64_t)r0 * d.ldh);
                    b[k] = ld16h<NT>(colbase + (int64_t)(r0 + 1) * d.ldh);
                }
#pragma unroll
                for (int k = 0; k < SCR_BATCH; k++) {
                    const int r0 = rb + 2 * k;
                    const double2 sxr = *reinterpret_cast<const double2*>(d.Sx + r0);
                    const float sxr0 = (float)sxr.x, sxr1 = (float)sxr.y;
                    const unsigned aw[4] = {a[k].x, a[k].y, a[k].z, a[k].w};
                    const unsigned bw[4] = {b[k].x, b[k].y, b[k].z, b[k].w};
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        if (nonneg)
                            screen_micro_nn(r0, c0 + 2 * j, m, twoP, k1, k2, bf_lo(aw[j]), bf_hi(aw[j]), bf_lo(bw[j]), bf_hi(bw[j]),
                                            sxr0, sxr1, sxc[2 * j], sxc[2 * j + 1], bk[j], NoEmit());
                        else
                            screen_micro(r0, c0 + 2 * j, m, twoP, cm2, cm2k, bf_lo(aw[j]), bf_hi(aw[j]), bf_lo(bw[j]), bf_hi(bw[j]),
                                         sxr0, sxr1, sxc[2 * j], sxc[2 * j + 1], bk[j]);
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < 4; j++) { lbv[j] = bk[j].lb; ubv[j] = bk[j].ub; }
        }
        float lb = __builtin_fminf(__builtin_fminf(lbv[0], lbv[1]), __builtin_fminf(lbv[2], lbv[3]));
        float ub = __builtin_fminf(__builtin_fminf(ubv[0], ubv[1]), __builtin_fminf(ubv[2], ubv[3]));
        const unsigned long long hitmask = __ballot(lb <= tp);  // lanes k_emit has to look at
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            lb = __builtin_fminf(lb, __shfl_down(lb, off, 64));
            ub = __builtin_fminf(ub, __shfl_down(ub, off, 64));
        }
        if (lane == 0) { lbrec[4 * t + w] = lb; ubrec[4 * t + w] = ub; shl[w] = lb; shu[w] = ub; d.shit[4 * t + w] = hitmask; }
        __syncthreads();
        if (threadIdx.x == 0) {
            lbt[t] = __builtin_fminf(__builtin_fminf(shl[0], shl[1]), __builtin_fminf(shl[2], shl[3]));
            ubt[t] = __builtin_fminf(__builtin_fminf(shu[0], shu[1]), __builtin_fminf(shu[2], shu[3]));
        }
        __syncthreads();
    }
}

// k_resolve: from the per-unit brackets to the exact pair, in one launch.  Every workgroup derives
// the same candidate list (units whose lower bound does not exceed this rank's smallest upper
// bound).  Few candidates (the usual case): workgroup 0 rescans them exactly and forms Cx/Cy
// (single GPU) or this rank's candidate for the all-gather (several GPUs).  Many: all workgroups
// share the rescans and the last one to arrive finishes.
constexpr int RES_BLOCKS = GATHER_RECS;  // its per-workgroup records go straight into the all-gather
constexpr int RES_LIST = 4096;

// exact scan of one 32 x 512 unit by a whole 1024-thread workgroup: every thread owns one column
// pair and four of the 16 row pairs x 2 column halves, and issues all its loads before computing
__device__ __forceinline__ void scan_unit_block(const Dev& d, int rbase, int cb, int m, int twoP, double cm2, Cand& best) {
    const int cp = (int)threadIdx.x & 127, rg = (int)threadIdx.x >> 7;  // rg in 0..7
    double2 a[4], b[4];
    int rr[4], cc[4];
    bool ok[4];
#pragma unroll
    for (int q = 0; q < 4; q++) {
        cc[q] = cb + (q >> 1) * 256 + 2 * cp;
        rr[q] = rbase + 2 * rg + (q & 1) * 16;
        ok[q] = cc[q] < m && rr[q] < m && cc[q] <= rr[q];
        a[q] = make_double2(0.0, 0.0);
        b[q] = a[q];
        if (ok[q]) {
            a[q] = *reinterpret_cast<const double2*>(d.D + (int64_t)rr[q] * d.ld + cc[q]);
            b[q] = *reinterpret_cast<const double2*>(d.D + (int64_t)(rr[q] + 1) * d.ld + cc[q]);
        }
    }
#pragma unroll
    for (int q = 0; q < 4; q++) {
        if (!ok[q]) continue;
        const double2 sxc = *reinterpret_cast<const double2*>(d.Sx + cc[q]);
        const int2 pc = *reinterpret_cast<const int2*>(d.spos + cc[q]);
        const double2 sxr = *reinterpret_cast<const double2*>(d.Sx + rr[q]);
        const int2 pr = *reinterpret_cast<const int2*>(d.spos + rr[q]);
        scan_micro(rr[q], cc[q], m, twoP, cm2, a[q].x, a[q].y, b[q].x, b[q].y, sxr.x, sxr.y, pr.x, pr.y, sxc.x, sxc.y, pc.x, pc.y, best);
    }
}

__global__ __launch_bounds__(1024) void k_resolve(Dev d) {
    __shared__ float shmin[16];
    __shared__ int cnt;
    __shared__ int list[RES_LIST];
    __shared__ Cand shc[16];
    State* st = d.st;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    Cand best;
    best = cand_none();
    if (st->la_hit) return;  // the lookahead window already holds this event's minimum (recs[0])
    if (st->done) {  // nothing to scan: leave "no candidate" records
        if (tid == 0) {
            (d.wx ? reinterpret_cast<Cand*>(d.wsend + wx_recs_off()) : d.gather ? d.gsend : d.recs)[blockIdx.x] = best;
            if (d.wx && blockIdx.x == 0) { int32_t* h = reinterpret_cast<int32_t*>(d.wsend); h[0] = 0; h[1] = 0; h[2] = 0; h[3] = 0; }
        }
        return;
    }
    const int m = st->m;
    const int ntiles = tri_tile_count(m, SCR_TH, SCR_R);
    const float finf = __builtin_inff();
    // 1. this rank's smallest upper bound, from the per-tile records
    const float* lbrec = d.srec;
    const float* lbt = d.stile;
    const float* ubt = d.stile + ntiles;
    float mn = finf;
    for (int t = tid; t < ntiles; t += 1024)
        if (t % d.world == d.rank) mn = fminf_(mn, ubt[t]);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) mn = fminf_(mn, __shfl_down(mn, off, 64));
    if (lane == 0) shmin[w] = mn;
    if (tid == 0) cnt = 0;
    __syncthreads();
    float g = shmin[0];
#pragma unroll
    for (int k = 1; k < 16; k++) g = fminf_(g, shmin[k]);
    // (a rank's smallest upper bound is >= the global one, so it only admits more units)
    const float thr = g + 2.0f * screen_delta(*st);
    bool all = !st->screen_ok || !(thr == thr);
    // 2. candidate units: the units, within the few tiles that pass, whose lower bound passes
    if (!all) {
        for (int t = tid; t < ntiles; t += 1024)
            if (t % d.world == d.rank && lbt[t] <= thr) {
                const float4 x = *reinterpret_cast<const float4*>(lbrec + 4 * (size_t)t);
                if (x.x <= thr) { const int i = atomicAdd(&cnt, 1); if (i < RES_LIST) list[i] = 4 * t; }
                if (x.y <= thr) { const int i = atomicAdd(&cnt, 1); if (i < RES_LIST) list[i] = 4 * t + 1; }
                if (x.z <= thr) { const int i = atomicAdd(&cnt, 1); if (i < RES_LIST) list[i] = 4 * t + 2; }
                if (x.w <= thr) { const int i = atomicAdd(&cnt, 1); if (i < RES_LIST) list[i] = 4 * t + 3; }
            }
    }
    __syncthreads();
    const int count = cnt;
    if (count > RES_LIST) all = true;
    const int total = all ? 4 * ntiles : count;
    // 3. exact rescans shared by all workgroups (k_decide reduces the per-workgroup results)
    {
        const int twoP = 2 * st->P;
        const double cm2 = (double)st->c - 2.0;
        // every workgroup holds the same SET of candidates, but in its own (atomic) order, so the
        // work is split by unit id, not by list position
        for (int i = 0; i < total; i++) {
            const int u = all ? i : list[i];
            if (u % (int)gridDim.x != (int)blockIdx.x) continue;
            if (all && (u >> 2) % d.world != d.rank) continue;
            int rt, ct;
            tri_tile_decode(u >> 2, SCR_R, rt, ct);
            scan_unit_block(d, rt * SCR_TH, ct * SCR_TW + (u & 3) * SCR_UW, m, twoP, cm2, best);
        }
    }
    best = block_reduce<16>(best, shc);
    if (tid == 0) {
        (d.wx ? reinterpret_cast<Cand*>(d.wsend + wx_recs_off()) : d.gather ? d.gsend : d.recs)[blockIdx.x] = best;  // several GPUs: these records are all-gathered
        if (blockIdx.x == 0 && d.wx) {
            // several ranks with windows: the base scan is closed after the exchange (k_merge); the header tells the
            // others how many pairs this rank emitted (k_emit has run) and how many units it rescanned
            int32_t* h = reinterpret_cast<int32_t*>(d.wsend);
            h[0] = st->la_emit ? *d.lacnt : 0;
            h[1] = all ? 1 : 0;
            h[2] = all ? (4 * ntiles) / d.world : count;
            h[3] = st->n_events;  // the event this block belongs to: k_merge refuses blocks of different events
        } else if (blockIdx.x == 0) {
            st->rescan_all = all ? 1 : 0;
            st->ncand = all ? 0 : count;
            st->n_screen_events += 1;
            st->n_rescan_units += all ? (int64_t)(4 * ntiles) / d.world : (int64_t)count;
            st->ev_screened = 1;
            la_close_base(*st, d.lalog, d.lacnt);  // (the screening pass may have emitted the pairs of a new lookahead window)
        }
    }
}

// After the exchange of a sharded base scan (several ranks with lookahead windows; fnn_core.h: wx_merge is the serial
// form): every rank builds the same tracked list from all ranks' emitted pairs, copies all ranks' candidate records
// into d.grecv for k_decide, and closes the base scan.
__global__ __launch_bounds__(1024) void k_merge(Dev d) {
    __shared__ int32_t cnt[64], off[65];
    __shared__ int32_t ovf;
    State* st = d.st;
    if (st->la_hit) return;
    const int tid = threadIdx.x;
    const int32_t capr = wx_pair_cap(d.world);
    const int64_t bb = wx_block_bytes(d.world);
    if (tid < d.world) {
        cnt[tid] = reinterpret_cast<const int32_t*>(d.wrecv + tid * bb)[0];
        // a block of ANOTHER event: the ranks have taken different decisions (must not happen: every decision is a
        // deterministic function of identical state) - an error on every rank that sees it, never a silent divergence
        if (!st->done && reinterpret_cast<const int32_t*>(d.wrecv + tid * bb)[3] != st->n_events) st->error = 12;
    }
    __syncthreads();
    if (tid == 0) {
        int64_t total = 0;
        int o = 0;
        for (int r = 0; r < d.world; r++) {
            if (cnt[r] > capr) o = 1;
            off[r] = (int32_t)total;
            total += cnt[r] < capr ? cnt[r] : capr;
        }
        off[d.world] = (int32_t)total;
        if (total > st->la_pcap) o = 1;
        ovf = o;
    }
    __syncthreads();
    if (st->la_emit && !ovf) {
        for (int r = 0; r < d.world; r++) {
            const int4* src = reinterpret_cast<const int4*>(d.wrecv + r * bb + wx_pairs_off());
            int4* dst = reinterpret_cast<int4*>(d.tpairs) + (int64_t)(LA_REC_INTS / 4) * off[r];
            for (int i = tid; i < (LA_REC_INTS / 4) * cnt[r]; i += 1024) dst[i] = src[i];
        }
    }
    for (int i = tid; i < d.world * GATHER_RECS; i += 1024) {
        const int r = i / GATHER_RECS, j = i % GATHER_RECS;
        d.grecv[i] = reinterpret_cast<const Cand*>(d.wrecv + r * bb + wx_recs_off())[j];
    }
    __syncthreads();
    if (tid == 0) {
        int64_t units = 0;
        for (int r = 0; r < d.world; r++) units += reinterpret_cast<const int32_t*>(d.wrecv + r * bb)[2];
        const int32_t* mine = reinterpret_cast<const int32_t*>(d.wrecv + d.rank * bb);
        *d.lacnt = ovf ? st->la_pcap + 1 : off[d.world];
        st->rescan_all = mine[1];
        st->ncand = mine[1] ? 0 : mine[2];
        st->n_screen_events += 1;
        st->n_rescan_units += units;
        st->ev_screened = 1;
        la_close_base(*st, d.lalog, d.lacnt);
    }
}

// D -> H (bf16) for the whole padded matrix, and max |D| over the n x n input
__global__ __launch_bounds__(256) void k_prep_screen(Dev d, int64_t nrows) {
    __shared__ unsigned long long shmax[4];
    unsigned long long b = 0;
    int neg = 0;
    const int64_t total = nrows * d.ld;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const double v = d.D[i];
        const int64_t r = i / d.ld, c = i - r * d.ld;
        if (d.H) d.H[r * d.ldh + c] = bf16_from_double(v);
        if (r < d.n && c < d.n) {
            const unsigned long long x = f2u(v) & 0x7FFFFFFFFFFFFFFFULL;
            b = x > b ? x : b;
            if (v < 0.0) neg = 1;
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const unsigned long long o = __shfl_down(b, off, 64);
        b = o > b ? o : b;
    }
    if ((threadIdx.x & 63) == 0) shmax[threadIdx.x >> 6] = b;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long x = shmax[0];
        for (int k = 1; k < 4; k++) x = shmax[k] > x ? shmax[k] : x;
        if (x) atomicMax(reinterpret_cast<unsigned long long*>(&d.st->dmax_bits), x);
    }
    if (__any(neg) && (threadIdx.x & 63) == 0) d.st->nonneg = 0;  // (set to 1 before the launch)
}

// ------------------------------------------------------------------ record reduction
__device__ __forceinline__ Cand reduce_records(const Dev& d, const Cand* src, int nrecs, Cand* sh) {
    Cand best;
    best = cand_none();
    if (!d.st->done) {
        for (int i = threadIdx.x; i < nrecs; i += 1024) {
            Cand c = src[i];
            if (cand_better(c, best)) best = c;
        }
    }
    return block_reduce<16>(best, sh);
}

// several GPUs: this rank's records -> one candidate for the all-gather
__global__ __launch_bounds__(1024) void k_reduce_local(Dev d, int nrecs) {
    __shared__ Cand sh[16];
    Cand best = reduce_records(d, d.recs, nrecs, sh);
    if (threadIdx.x == 0) d.gsend[0] = best;
}

// ------------------------------------------------------------------ exact block-parallel chain sum
// Sequential fp64 sum of buf[0..m) in index order, bit-identical to a scalar loop, evaluated by
// one 1024-thread workgroup (algorithm and proof sketch: fnn_chain.h; CPU model: tests/emu).
//   1. every thread loads EPT consecutive addends; a block-wide prefix sum of the thread totals
//      gives each thread a PREDICTED partial sum in front of its chunk;
//   2. the thread turns its addends into parity automata relative to the predicted binade and
//      composes them; a chunk whose addends do not all sit in one binade is "mixed" and its
//      addends are parked in LDS;
//   3. a segmented scan over the lanes of each wave composes runs of chunks of equal binade;
//   4. wave 0 walks over the runs: one exact O(1) update per run, ordinary additions for mixed
//      chunks, and retries at finer granularity where the binade assumption is refuted.

template <int EPT>
struct ChainLds {
    double wtot[CH_T / 64];
    double s;
    int slot_count;
    int32_t E[CH_T];
    int32_t flags[CH_T];  // bit0 pure, bit1 last chunk of its run
    int32_t slot[CH_T];
    uint64_t own[CH_T], sc[CH_T];  // increments of a chunk / of the run up to the chunk, as integers (mono_inc_bits);
                                   // chunks with a tie are "mixed", so an automaton here is one constant
    double vals[CH_NSLOT][EPT];
    // record form (block_chain_sum2): special addend per thread; per-wave totals of the segmented scan
    double spv[CH_T];
    ChRec prec[CH_NSLOT];      // records of the parked chunks, computed lane-parallel (chain_record_wave)
    double pend_A0[CH_NSLOT];
    int32_t pend_cnt[CH_NSLOT];
    uint64_t wk[CH_T / 64];
    int32_t wr[CH_T / 64];
    int32_t fail;
};

__device__ __forceinline__ uint64_t shfl_up_u64(uint64_t v, int d) {
    return (uint64_t)__shfl_up((unsigned long long)v, d, 64);
}
__device__ __forceinline__ int readlane_i32(int v, int lane) { return __builtin_amdgcn_readlane(v, lane); }
__device__ __forceinline__ uint64_t readlane_u64(uint64_t v, int lane) {
    uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, lane);
    uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), lane);
    return ((uint64_t)hi << 32) | lo;
}

__device__ __forceinline__ uint64_t readfirstlane_u64(uint64_t v) {
    uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
    uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32));
    return ((uint64_t)hi << 32) | lo;
}

__device__ __forceinline__ void chain_serial_global(double& s, const double* buf, int start, int cnt, int m) {
    for (int i = 0; i < cnt; i++)
        if (start + i < m) s += buf[chain_addr(start + i)];
}

__device__ int g_chain_form = 2;        // diagnostic (FNN_CHAIN_FORM=1): the first form in fnn_test_chain_sum
__device__ int g_chain_stop_after = 0;  // diagnostic: 1 loads+prefix, 2 +automata, 3 +segmented scan (0 = full)

// MODE 0: the whole sum.  MODE 1: steps 1-3 only (the records stay in L; m <= CH_T * EPT).  MODE 2:
// step 4 only, on records that are in L already (m <= CH_T * EPT).
template <int EPT, int MODE = 0>
__device__ __forceinline__ double block_chain_sum(const double* __restrict__ buf, int m, int guard_bits, ChainLds<EPT>& L,
                                  ChainStats* stats) {
    const int stop_after = stats ? g_chain_stop_after : 0;
    static_assert(EPT % 2 == 0 && EPT <= 64, "EPT");
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    ChainStats cs{0, 0, 0, 0};
    if (tid == 0) L.s = 0.0;
    for (int base = 0; base < m; base += CH_T * EPT) {
        double s_in = 0.0;
        if (MODE != 2) {
        if (tid == 0) L.slot_count = 0;
        // 1. addends of this thread: chunk-interleaved buffer (fnn_core.h chain_addr), so the j-th
        //    16-byte load of all lanes of a wave is one contiguous 1 KiB segment
        static_assert(EPT == CH_EPT, "the buffer layout is fixed to CH_EPT addends per thread");
        double a[EPT];
        const int idx0 = base + tid * EPT;
        const double* src = buf + base + 2 * tid;
#pragma unroll
        for (int i = 0; i < EPT; i += 2) {
            double2 v = make_double2(0.0, 0.0);
            if (idx0 + i < m) v = *reinterpret_cast<const double2*>(src + (i >> 1) * (2 * CH_T));
            a[i] = v.x;
            a[i + 1] = (idx0 + i + 1 < m) ? v.y : 0.0;
        }
        double loc = 0.0;
#pragma unroll
        for (int i = 0; i < EPT; i++) loc += a[i];
        double inc = loc;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const double t = __shfl_up(inc, d, 64);
            if (lane >= d) inc += t;
        }
        if (lane == 63) L.wtot[w] = inc;
        __syncthreads();
        double wpre = 0.0;
        for (int k = 0; k < w; k++) wpre += L.wtot[k];
        s_in = L.s;
        double A = s_in + (wpre + (inc - loc));  // predicted partial sum in front of this chunk
        if (stop_after == 1) { if (tid == 0) L.s = A; __syncthreads(); continue; }
        // 2. automaton of the chunk: the binade is predicted once per chunk (partial sums are
        //    monotone over non-negative addends; a negative addend makes the chunk "mixed")
        int32_t E = -1;
        bool pure = chain_predict(A, A + loc, guard_bits != 0, E);
        // a chunk that is not predicted to stay in one binade is "mixed": its addends are parked now,
        // while they are still in registers (nothing of a[] stays live across the automata)
        int slot = -1;
        if (!pure && idx0 < m) {
            slot = atomicAdd(&L.slot_count, 1);
            if (slot < CH_NSLOT) {
#pragma unroll
                for (int i = 0; i < EPT; i++) L.vals[slot][i] = a[i];
            } else slot = -1;
        }
        Mono mt = mono_identity();
        if (pure) {
            pure = chain_chunk<EPT>(a, inv_ulp(E), mt);
            if (!pure) {  // rare (negative / non-finite addend, tie): parked from memory
                slot = atomicAdd(&L.slot_count, 1);
                if (slot < CH_NSLOT) {
#pragma unroll 1
                    for (int i = 0; i < EPT; i++) L.vals[slot][i] = (idx0 + i < m) ? buf[chain_addr(idx0 + i)] : 0.0;
                } else slot = -1;
            }
        }
        if (stop_after == 2) { if (tid == 0) L.s = mt.i0 + (double)E + (pure ? 1.0 : 0.0); __syncthreads(); continue; }
        // 3. runs inside the wave: segmented inclusive scan of the chunk automata
        const int pure_prev = __shfl_up((int)pure, 1, 64);
        const int E_prev = __shfl_up(E, 1, 64);
        const bool head = (lane == 0) || !pure || !pure_prev || (E != E_prev);
        double sc = mt.i0;  // (== mt.i1: constant automata compose by addition)
        int f = head ? 1 : 0;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const double o = __shfl_up(sc, d, 64);
            const int fo = __shfl_up(f, d, 64);
            if (lane >= d && !f) { sc = o + sc; f = fo; }
        }
        const int head_next = __shfl_down((int)head, 1, 64);
        const bool endf = (lane == 63) || head_next || (idx0 + EPT >= m);
        L.E[tid] = E;
        L.flags[tid] = (pure ? 1 : 0) | (endf ? 2 : 0);
        L.slot[tid] = slot;
        L.own[tid] = mono_inc_bits(mt.i0);
        L.sc[tid] = mono_inc_bits(sc);
        __syncthreads();
        if (stop_after == 3) { if (tid == 0) L.s = sc; __syncthreads(); continue; }
        if (MODE == 1) return 0.0;
        } else {
            __syncthreads();
            s_in = L.s;
        }
        // 4. the walk (every lane of wave 0 carries the same s)
        if (w == 0) {
            // the running sum lives as its bit pattern in scalar registers: a run is applied by a few
            // scalar integer operations (mono_apply_pattern) instead of a chain of dependent fp64 ones
            uint64_t sb = readfirstlane_u64(f2u(s_in));
            // (the records of the next 64 chunks are fetched from LDS while the current ones are walked)
            int nE = L.E[lane], nf = L.flags[lane], nslot = L.slot[lane];
            uint64_t no = L.own[lane], ns = L.sc[lane];
            for (int ww = 0; ww < CH_T / 64; ww++) {
                const int first = base + ww * 64 * EPT;
                if (first >= m) break;
                const int nvalid = min(64, (m - first + EPT - 1) / EPT);
                const int rE = nE, rf = nf, rslot = nslot;
                const uint64_t ro = no, rs = ns;
                if (ww + 1 < CH_T / 64) {
                    const int id = (ww + 1) * 64 + lane;
                    nE = L.E[id]; nf = L.flags[id]; nslot = L.slot[id];
                    no = L.own[id]; ns = L.sc[id];
                }
                uint64_t endmask = __ballot((rf & 2) != 0);
                if (nvalid < 64) endmask &= (1ULL << nvalid) - 1;
                int prev_end = -1;
                while (endmask) {
                    const int e = __builtin_ctzll(endmask);
                    endmask &= endmask - 1;
                    const int fe = readlane_i32(rf, e);
                    if (stop_after == 4) { sb += (uint64_t)fe; prev_end = e; continue; }  // diagnostic: loop structure only
                    if (fe & 1) {
                        const uint64_t m0 = readlane_u64(rs, e);
                        if (stop_after == 5) { sb += m0; prev_end = e; continue; }  // diagnostic: no apply, no mixed adds
                        if (mono_apply_pattern(sb, readlane_i32(rE, e), m0, m0)) cs.runs++;
                        else {
                            cs.run_fail++;
                            for (int j = prev_end + 1; j <= e; j++) {
                                const uint64_t mj = readlane_u64(ro, j);
                                if (!mono_apply_pattern(sb, readlane_i32(rE, j), mj, mj)) {
                                    cs.thread_fail++;
                                    double sv = u2f(sb);
                                    chain_serial_global(sv, buf, first + j * EPT, EPT, m);
                                    sb = readfirstlane_u64(f2u(sv));
                                }
                            }
                        }
                    } else {
                        cs.mixed++;
                        if (stop_after == 5) { prev_end = e; continue; }
                        const int sl = readlane_i32(rslot, e);
                        double sv = u2f(sb);
                        if (sl >= 0) {
                            // (every lane adds the same parked addends in order: LDS broadcast reads, which
                            //  pipeline under the dependent additions - no cross-lane traffic)
                            const double* pv = L.vals[sl];
#pragma unroll
                            for (int j = 0; j < EPT; j++) sv += pv[j];
                        } else chain_serial_global(sv, buf, first + e * EPT, EPT, m);
                        sb = readfirstlane_u64(f2u(sv));
                    }
                    prev_end = e;
                }
            }
            if (lane == 0) L.s = u2f(sb);
        }
        __syncthreads();
    }
    __syncthreads();
    if (stats && tid == 0) *stats = cs;
    return L.s;
}


// ------------------------------------------------------------------ the record form of the chain sum
// fnn_chain.h "records": a thread describes its 32 addends as constants around at most ONE addend that needs an
// ordinary addition (a binade crossing or a tie); neighbouring constant pieces of equal binade are merged by a
// segmented scan over the 1024 threads, and wave 0 only walks over the ~25 places where something has to happen
// (a verified integer update of the sum's bit pattern, then one addition) instead of adding ~20 chunks of 32
// addends one by one.  Any failed verification -> the first form computes the whole sum.
// The record (fnn_chain.h: ChRec) of ONE chunk of <= 32 addends by one wave, lane l < 32 holding addend l: what
// chain_thread_record does with a 32-step loop per thread (and every wave has some thread that needs it) takes a
// prefix scan, two ballots and two masked reductions here.  Same contract: constants are only claimed for pieces whose
// addends are all non-negative, finite, tie-free multiples-after-rounding in ONE predicted binade; the walker verifies
// every piece when it applies it.
__device__ __forceinline__ ChRec chain_record_wave(double al, int cnt, double A0, bool guard) {
    const int lane = threadIdx.x & 63;
    ChRec r;
    r.kind = CHR_SERIAL; r.E0 = r.E1 = -1; r.c0 = r.c1 = 0; r.sp = 0.0;
    const bool in = lane < cnt;
    // predicted partial sums in front of / behind every addend
    double inc = in ? al : 0.0;
#pragma unroll
    for (int d = 1; d < 32; d <<= 1) {
        const double t = __shfl_up(inc, d, 64);
        if ((lane & 31) >= d) inc += t;
    }
    const double pb = A0 + (inc - (in ? al : 0.0)), pa = A0 + inc;
    int32_t El = -1;
    const bool okl = in && chain_predict(pb, pa, guard, El);
    const int32_t E_first = __builtin_amdgcn_readlane(El, 0);
    const bool ok0 = __builtin_amdgcn_readlane((int)okl, 0) != 0;
    const unsigned long long inm = __ballot(in);
    const unsigned long long bad = __ballot(in && (!okl || El != E_first)) | (ok0 ? 0ULL : 1ULL);
    // per-lane constant relative to a binade, with the tie / sign / range checks
    auto konst = [&](int32_t E, bool& tie, bool& inval) {
        const double t = al * inv_ulp(E);
        const double ci = __builtin_rint(t);
        tie = __builtin_fabs(t - ci) == 0.5;
        inval = !(t >= 0.0 && t < CH_TWO53) || (hi32(t) >> 31);
        return ci;
    };
    auto msum = [&](double v, unsigned long long mask) {  // sum of v over the lanes of mask (integer-valued: any order)
        double x = ((mask >> lane) & 1ULL) ? v : 0.0;
#pragma unroll
        for (int off = 16; off >= 1; off >>= 1) x += __shfl_xor(x, off, 64);
        return __builtin_bit_cast(double, readfirstlane_u64(f2u(x)));
    };
    if (!bad) {
        // one binade for the whole chunk: no tie -> CONST; one tie -> SPLIT around it
        bool tie, inval;
        const double ci = konst(E_first, tie, inval);
        if (__ballot(in && inval)) return r;
        const unsigned long long tm = __ballot(in && tie);
        if (__builtin_popcountll(tm) > 1) return r;
        if (!tm) { r.kind = CHR_CONST; r.E0 = E_first; r.c0 = mono_inc_bits(msum(ci, inm)); return r; }
        const int jt = __builtin_ctzll(tm);
        const unsigned long long lo = inm & ((1ULL << jt) - 1ULL), hi = inm & ~((2ULL << jt) - 1ULL);
        r.kind = CHR_SPLIT; r.E0 = E_first; r.E1 = E_first;
        r.c0 = mono_inc_bits(msum(ci, lo)); r.c1 = mono_inc_bits(msum(ci, hi));
        r.sp = __builtin_bit_cast(double, readlane_u64(f2u(al), jt));
        return r;
    }
    const int js = __builtin_ctzll(bad);
    const unsigned long long lo = inm & ((1ULL << js) - 1ULL), hi = inm & ~((2ULL << js) - 1ULL);
    int32_t E1 = -1;
    if (hi) {
        const int jn = __builtin_ctzll(hi);
        E1 = __builtin_amdgcn_readlane(El, jn);
        if (__ballot(((hi >> lane) & 1ULL) && (!okl || El != E1))) return r;  // the rest leaves its binade again
    }
    bool tie0 = false, inv0 = false, tie1 = false, inv1 = false;
    const double c0l = lo ? konst(E_first, tie0, inv0) : 0.0;
    const double c1l = hi ? konst(E1, tie1, inv1) : 0.0;
    if (__ballot((((lo >> lane) & 1ULL) && (tie0 || inv0)) || (((hi >> lane) & 1ULL) && (tie1 || inv1)))) return r;
    r.kind = CHR_SPLIT;
    r.E0 = lo ? E_first : -1; r.c0 = lo ? mono_inc_bits(msum(c0l, lo)) : 0;
    r.E1 = hi ? E1 : -1; r.c1 = hi ? mono_inc_bits(msum(c1l, hi)) : 0;
    r.sp = __builtin_bit_cast(double, readlane_u64(f2u(al), js));
    return r;
}

// The rare paths (a mispredicted binade in the record form; an uncertified 4-candidate choice) call the first form
// out of line: inlined, its 32 addends per thread are part of the register allocation of k_track's tail.
template <int EPT>
__device__ __attribute__((noinline)) double block_chain_sum_rare(const double* __restrict__ buf, int m, int guard_bits, ChainLds<EPT>* L,
                                                               ChainStats* stats) {
    return block_chain_sum<EPT>(buf, m, guard_bits, *L, stats);
}

template <int EPT>
__device__ __forceinline__ double block_chain_sum2(const double* __restrict__ buf, int m, int guard_bits, ChainLds<EPT>& L,
                                                   ChainStats* stats) {
    static_assert(EPT == CH_EPT, "the buffer layout is fixed to CH_EPT addends per thread");
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    ChainStats cs{0, 0, 0, 0};
    if (tid == 0) { L.s = 0.0; L.fail = 0; }
    for (int base = 0; base < m; base += CH_T * EPT) {
        if (tid == 0) L.slot_count = 0;
        // 1. this thread's addends (chunk-interleaved buffer) and the predicted partial sum in front of them
        double a[EPT];
        const int idx0 = base + tid * EPT;
        const double* src = buf + base + 2 * tid;
#pragma unroll
        for (int i = 0; i < EPT; i += 2) {
            double2 v = make_double2(0.0, 0.0);
            if (idx0 + i < m) v = *reinterpret_cast<const double2*>(src + (i >> 1) * (2 * CH_T));
            a[i] = v.x;
            a[i + 1] = (idx0 + i + 1 < m) ? v.y : 0.0;
        }
        double l4[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int i = 0; i < EPT; i++) l4[i & 3] += a[i];
        const double loc = (l4[0] + l4[1]) + (l4[2] + l4[3]);  // (prediction only: any order)
        double inc = loc;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const double t = __shfl_up(inc, d, 64);
            if (lane >= d) inc += t;
        }
        if (lane == 63) L.wtot[w] = inc;
        __syncthreads();
        double wpre = 0.0;
        for (int k = 0; k < w; k++) wpre += L.wtot[k];
        const double s_in = L.s;
        const double A0 = s_in + (wpre + (inc - loc));
        // 2. the record of the chunk.  The common case - every addend a tie-free constant of ONE predicted binade - is
        //    decided by the thread itself; any other chunk is parked in LDS and its record is worked out by a wave,
        //    lane-parallel (chain_record_wave), so that no wave runs through a 32-step search for one of its threads
        const int cnt = idx0 >= m ? 0 : (m - idx0 < EPT ? m - idx0 : EPT);
        ChRec r;
        r.kind = CHR_EMPTY; r.E0 = r.E1 = -1; r.c0 = r.c1 = 0; r.sp = 0.0;
        int slot = -1;
        if (cnt > 0) {
            int32_t E = -1;
            int nt = 0;
            uint64_t c = 0;
            if (chain_predict(A0, A0 + loc, guard_bits != 0, E) && chain_consts(a, 0, cnt, E, c, &nt, nullptr) && nt == 0) {
                r.kind = CHR_CONST; r.E0 = E; r.c0 = c;
            } else {
                r.kind = CHR_SERIAL;
                slot = atomicAdd(&L.slot_count, 1);
                if (slot < CH_NSLOT) {
#pragma unroll
                    for (int i = 0; i < EPT; i++) L.vals[slot][i] = a[i];
                    L.pend_A0[slot] = A0;
                    L.pend_cnt[slot] = cnt;
                } else slot = -1;  // (more odd chunks than slots: added one by one from memory)
            }
        }
        __syncthreads();
        {
            const int ns = L.slot_count < CH_NSLOT ? L.slot_count : CH_NSLOT;
            for (int sl = w; sl < ns; sl += CH_T / 64) {
                const double al = lane < EPT ? L.vals[sl][lane < EPT ? lane : 0] : 0.0;
                const ChRec pr = chain_record_wave(al, L.pend_cnt[sl], L.pend_A0[sl], guard_bits != 0);
                if (lane == 0) L.prec[sl] = pr;
            }
        }
        __syncthreads();
        if (slot >= 0) r = L.prec[slot];
        // 3. segmented scan of the constant pieces: element (reset, value); tailE = binade of the piece a thread leaves open
        const int32_t tailE = r.kind == CHR_CONST ? r.E0 : (r.kind == CHR_SPLIT ? r.E1 : -2);
        int32_t prevTail = __shfl_up(tailE, 1, 64);
        L.E[tid] = tailE;
        __syncthreads();
        if (lane == 0) prevTail = tid > 0 ? L.E[tid - 1] : -2;
        const bool ebreak = r.kind == CHR_CONST && prevTail != r.E0;
        int rr = (r.kind == CHR_SPLIT || r.kind == CHR_SERIAL || ebreak) ? 1 : 0;
        uint64_t kk = r.kind == CHR_CONST ? r.c0 : (r.kind == CHR_SPLIT ? (r.E1 >= 0 ? r.c1 : 0) : 0);
        const int r_own = rr;
        const uint64_t k_own = kk;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int pr = __shfl_up(rr, d, 64);
            const uint64_t pk = shfl_up_u64(kk, d);
            if (lane >= d) {
                if (!rr) { const uint64_t t = pk + kk; kk = t > (1ULL << 60) ? (1ULL << 60) : t; }
                rr |= pr;
            }
        }
        if (lane == 63) { L.wr[w] = rr; L.wk[w] = kk; }
        __syncthreads();
        uint64_t X = 0;
        for (int q = 0; q < w; q++) { const uint64_t t = X + L.wk[q]; X = L.wr[q] ? L.wk[q] : (t > (1ULL << 60) ? (1ULL << 60) : t); }
        uint64_t after;
        { const uint64_t t = X + kk; after = rr ? kk : (t > (1ULL << 60) ? (1ULL << 60) : t); }
        uint64_t xin = shfl_up_u64(after, 1);
        if (lane == 0) xin = X;
        (void)r_own; (void)k_own;
        // what the walker needs of an event thread: kind, binade and value of what is open in front of it, its own pieces
        const bool evt = r.kind == CHR_SPLIT || r.kind == CHR_SERIAL || (ebreak && xin != 0);
        __syncthreads();  // (L.E is re-used below)
        L.E[tid] = prevTail;
        L.flags[tid] = r.kind | (evt ? 4 : 0);
        L.slot[tid] = r.kind == CHR_SERIAL ? slot : r.E0;
        L.own[tid] = r.c0;
        L.sc[tid] = xin;
        L.spv[tid] = r.sp;
        const int lastT = ((m - base < CH_T * EPT ? m - base : CH_T * EPT) - 1) / EPT;
        if (tid == lastT) { L.wk[0] = after; L.wr[0] = tailE; }  // (the piece left open at the end of this super-chunk)
        __syncthreads();
        if (r.kind == CHR_SERIAL && tid <= lastT) cs.mixed++;
        // 4. the walk: only the event threads, in order
        if (w == 0) {
            uint64_t sb = readfirstlane_u64(f2u(s_in));
            bool ok = true;
            auto apply = [&](int E, uint64_t c) {
                if (c == 0) return;
                if (mono_apply_pattern(sb, E, c, c)) cs.runs++;
                else ok = false;
            };
            for (int ww = 0; ww < CH_T / 64 && ok; ww++) {
                if (ww * 64 > lastT) break;
                const int id = ww * 64 + lane;
                const int rf = L.flags[id], rEin = L.E[id], rs = L.slot[id];
                const uint64_t rc0 = L.own[id], rx = L.sc[id];
                const double rsp = L.spv[id];
                unsigned long long em = __ballot((rf & 4) != 0 && id <= lastT);
                while (em && ok) {
                    const int e = __builtin_ctzll(em);
                    em &= em - 1;
                    const int kind = readlane_i32(rf, e) & 3, Ein = readlane_i32(rEin, e), sl = readlane_i32(rs, e);
                    const uint64_t x = readlane_u64(rx, e), c0 = readlane_u64(rc0, e);
                    if (kind == CHR_CONST) apply(Ein, x);  // (a change of binade without a special addend in between)
                    else if (kind == CHR_SPLIT) {
                        const int E0 = sl;
                        if (E0 >= 0 && E0 == Ein) { const uint64_t t = x + c0; apply(E0, t > (1ULL << 60) ? (1ULL << 60) : t); }
                        else { apply(Ein, x); if (E0 >= 0) apply(E0, c0); }
                        if (!ok) break;
                        const double sv = u2f(sb) + u2f(readlane_u64(f2u(rsp), e));
                        sb = readfirstlane_u64(f2u(sv));
                    } else {
                        apply(Ein, x);
                        if (!ok) break;
                        double sv = u2f(sb);
                        if (sl >= 0) {
                            const double* pv = L.vals[sl];
#pragma unroll
                            for (int j = 0; j < EPT; j++) sv += pv[j];
                        } else chain_serial_global(sv, buf, base + (ww * 64 + e) * EPT, EPT, m);
                        sb = readfirstlane_u64(f2u(sv));
                    }
                }
            }
            if (ok) apply((int)L.wr[0], L.wk[0]);
            if (lane == 0) { L.s = u2f(sb); if (!ok) L.fail = 1; }
        }
        __syncthreads();
        if (L.fail) break;
    }
    __syncthreads();
    if (L.fail) {  // (rare: a mispredicted binade; the first form redoes the whole sum with its own fallbacks)
        ChainStats c1{0, 0, 0, 0};
        const double r1 = block_chain_sum_rare<EPT>(buf, m, guard_bits, &L, stats ? &c1 : nullptr);
        if (stats && tid == 0) { stats->runs = cs.runs + c1.runs; stats->mixed = cs.mixed + c1.mixed; stats->run_fail = 1 + c1.run_fail; stats->thread_fail = c1.thread_fail; }
        return r1;
    }
    if (stats) {
        // (the serial chunks were counted by their own threads)
        __shared__ int mixed_total;
        if (tid == 0) mixed_total = 0;
        __syncthreads();
        if (cs.mixed) atomicAdd(&mixed_total, cs.mixed);
        __syncthreads();
        if (tid == 0) { cs.mixed = mixed_total; *stats = cs; }
    }
    return L.s;
}

// ------------------------------------------------------------------ the decide step
// Cx / Cy from the event's best candidate (NetMakerOriginal.java:376-380), the 4-candidate choice
// (:413-452) and the merge plan (:462-488), by ONE workgroup: the last tracking workgroup of k_track
// for events served by a lookahead window, k_decide for events that scanned.
//   * the candidate's slots, the slot-table entries the plan can touch (fnn_core.h: tab_keys), the
//     4 x 4 block of the matrix over the two clusters' nodes and their approximate weighted row sums
//     T are fetched lane-parallel in two round trips; wave 1 sums the partial sums of T of the
//     previous event's new cluster beside the first one;
//   * the choice among the <= 4 candidates is certified from T (fnn_core.h: rx_from_T, rx_certify);
//     only if two candidates are closer than the error bound (exact ties, ...) the whole workgroup
//     evaluates the <= 4 ComputeRx sums exactly (block_chain_sum) first;
//   * one thread replays the integer side of the merge on the preloaded table entries (CachedTab:
//     registers; stores write through), one wave replays the micro-ops symbolically
//     (build_targets_wave: lane i holds involved slot S[i] and its symbolic row; searches are
//     ballots, the micro-ops run with wave-uniform control through readlane and lane-conditional
//     moves).  All of it on an LDS copy of the control block that the workgroup copies in and out.
__device__ __forceinline__ void build_targets_wave(State& st) {
    const int lane = threadIdx.x & 63;
    const int nops = __builtin_amdgcn_readfirstlane(st.nops);
    const int U = __builtin_amdgcn_readfirstlane(st.U), m = __builtin_amdgcn_readfirstlane(st.m);
    const int ev_finish = __builtin_amdgcn_readfirstlane(st.ev_finish);
    static_assert(sizeof(Op) == 32, "Op is read as 8 dwords");
    const int opw = lane < 8 * MAX_OPS ? reinterpret_cast<const int32_t*>(st.ops)[lane] : 0;
    int err = 0;
    int Sl = -1, nS = 0;
    auto add_slot = [&](int sl) {
        if (__ballot(lane < nS && Sl == sl)) return;
        if (nS >= MAX_S) { err = 5; return; }
        if (lane == nS) Sl = sl;
        nS++;
    };
    add_slot(U);
    add_slot(U + 1);
    for (int i = 0; i < nops; i++) {
        const int kind = __builtin_amdgcn_readlane(opw, 8 * i);
        add_slot(__builtin_amdgcn_readlane(opw, 8 * i + 1));
        add_slot(__builtin_amdgcn_readlane(opw, 8 * i + 2));
        if (kind == OP_AGG3) {
            add_slot(__builtin_amdgcn_readlane(opw, 8 * i + 3));
            add_slot(__builtin_amdgcn_readlane(opw, 8 * i + 4));
            add_slot(__builtin_amdgcn_readlane(opw, 8 * i + 5));
        }
    }
    struct Sym { int kind, a, b, c, d; };
    int sk = T_COPY, sa = Sl, sb = -1, sc = -1, sd = -1;  // this lane's symbolic row
    auto idx = [&](int sl) {
        const unsigned long long hit = __ballot(lane < nS && Sl == sl);
        return hit ? (int)__builtin_ctzll(hit) : 0;
    };
    auto get = [&](int i) {
        Sym t;
        t.kind = __builtin_amdgcn_readlane(sk, i); t.a = __builtin_amdgcn_readlane(sa, i);
        t.b = __builtin_amdgcn_readlane(sb, i); t.c = __builtin_amdgcn_readlane(sc, i);
        t.d = __builtin_amdgcn_readlane(sd, i);
        return t;
    };
    auto put = [&](int i, const Sym& t) {
        if (lane == i) { sk = t.kind; sa = t.a; sb = t.b; sc = t.c; sd = t.d; }
    };
    auto comb = [&](const Sym& A, const Sym& B) {  // value (2/3)*A + B/3
        Sym r;
        r.kind = T_COPY; r.a = r.b = r.c = r.d = -1;
        if (A.kind == T_COPY && B.kind == T_COPY) { r.kind = T_L1; r.a = A.a; r.b = B.a; }
        else if (A.kind == T_L1 && B.kind == T_L1 && A.b == B.b) { r.kind = T_L2U; r.a = A.a; r.b = A.b; r.c = B.a; }
        else if (A.kind == T_COPY && B.kind == T_L1) { r.kind = T_L2V; r.d = A.a; r.c = B.a; r.b = B.b; }
        else err = 6;
        return r;
    };
    for (int i = 0; i < nops; i++) {
        const int kind = __builtin_amdgcn_readlane(opw, 8 * i);
        const int oa = __builtin_amdgcn_readlane(opw, 8 * i + 1), ob = __builtin_amdgcn_readlane(opw, 8 * i + 2);
        if (kind == OP_SWAP) {
            const int ia = idx(oa), ib = idx(ob);
            const Sym ta = get(ia), tb = get(ib);
            put(ia, tb);
            put(ib, ta);
        } else if (kind == OP_MOVE) {
            const Sym t = get(idx(oa));
            put(idx(ob), t);
        } else if (kind == OP_AGG3) {
            const int oc = __builtin_amdgcn_readlane(opw, 8 * i + 3), od = __builtin_amdgcn_readlane(opw, 8 * i + 4),
                      oe = __builtin_amdgcn_readlane(opw, 8 * i + 5);
            const Sym sx = get(idx(oa)), sy = get(idx(ob)), sz = get(idx(oc));
            const Sym nu = comb(sx, sy), nv = comb(sz, sy);
            put(idx(od), nu);
            put(idx(oe), nv);
        }
    }
    const bool isUV = (Sl == U || Sl == U + 1);
    const bool keep = lane < nS && !(Sl >= m && !ev_finish) && !(!isUV && sk == T_COPY && sa == Sl);
    const unsigned long long km = __ballot(keep);
    const int pos = __builtin_popcountll(km & ((1ULL << lane) - 1ULL));
    const int ntgt = __builtin_popcountll(km);
    if (keep && pos < MAX_TGT) {
        Tgt t;
        t.dst = Sl; t.kind = sk; t.a = sa; t.b = sb; t.c = sc; t.d = sd;
        st.tgt[pos] = t;
    }
    if (lane < nS) st.S[lane] = Sl;
    const unsigned long long um = __ballot(keep && Sl == U), vm = __ballot(keep && Sl == U + 1);
    if (lane == 0) {
        st.nS = nS;
        st.ntgt = ntgt < MAX_TGT ? ntgt : MAX_TGT;
        st.tU = um ? __builtin_popcountll(km & ((1ULL << __builtin_ctzll(um)) - 1ULL)) : -1;
        st.tV = vm ? __builtin_popcountll(km & ((1ULL << __builtin_ctzll(vm)) - 1ULL)) : -1;
        if (ntgt > MAX_TGT) st.error = 7;
        else if (err) st.error = err;
    }
}


// The cached slot-table entries in the lanes of one wave (fnn_core.h: CachedTab is the same thing on arrays):
// lane i < TAB_NK holds key[i] with its sid / spos, lane i < TAB_NP pkey[i] with its pslot.  The whole wave runs the
// plan with wave-uniform control; a lookup is a ballot and a readlane instead of a search through memory.
struct WaveTab {
    const Dev& d;
    int lane;
    int32_t key, vsid, vspos, pkey, vpslot;
    int32_t misses;
    // (a miss - an entry outside tab_keys' set - is not expected: the hit path is the straight-line one.  The table's writes go
    //  out from every lane: the value is wave-uniform, and one store instruction without the exec-mask detour of "lane 0 only")
    __device__ __forceinline__ int32_t sid(int32_t s) {
        const unsigned long long hit = __ballot(key == s);
        if (__builtin_expect(hit != 0ULL, 1)) return __builtin_amdgcn_readlane(vsid, (int)__builtin_ctzll(hit));
        misses++;
        return d.sid[s];
    }
    __device__ __forceinline__ int32_t spos(int32_t s) {
        const unsigned long long hit = __ballot(key == s);
        if (__builtin_expect(hit != 0ULL, 1)) return __builtin_amdgcn_readlane(vspos, (int)__builtin_ctzll(hit));
        misses++;
        return d.spos[s];
    }
    __device__ __forceinline__ int32_t pslot(int32_t p) {
        const unsigned long long hit = __ballot(pkey == p);
        if (__builtin_expect(hit != 0ULL, 1)) return __builtin_amdgcn_readlane(vpslot, (int)__builtin_ctzll(hit));
        misses++;
        return d.pslot[p];
    }
    __device__ __forceinline__ void set_sid(int32_t s, int32_t v) { if (key == s) vsid = v; d.sid[s] = v; }
    __device__ __forceinline__ void set_spos(int32_t s, int32_t v) { if (key == s) vspos = v; d.spos[s] = v; }
    __device__ __forceinline__ void set_pslot(int32_t p, int32_t v) { if (pkey == p) vpslot = v; d.pslot[p] = v; }
};

struct DecideLds {
    State lst;                 // the control block while the workgroup works on it
    int32_t key[TAB_NK], vsid[TAB_NK], vspos[TAB_NK];
    int32_t pkey[TAB_NP], vpslot[TAB_NP];
    int32_t ab[2];             // slots of the candidate's two nodes
    int32_t need, cert, exact, misses;
    long long tk[10];          // k_track: timestamps of the phase split (diagnostic)
    long long dk[6];           // decide step: timestamps {entry, loads done, Cx/Cy + choice, plan, symbolic replay}
    int32_t tkon;
    double quad[16];           // D over {a, a^1, b, b^1} x {a, a^1, b, b^1}
    double tz[4];              // T of a, a^1, b, b^1
    double tfin[2];            // T of the previous event's new cluster, just summed
    double rx[4];
    Quad qd;
    // fused event kernel (k_track with fuse): the plan message (sent by the deciding workgroup, received by the column workgroups),
    // the slot of the previous event's cluster whose T the decide step has just summed (tfin; -2: none), and the deciding
    // workgroup's copy of the involved slots' block
    PlanMsg msg;
    int32_t msg_pU, berr;
    int32_t fz[6];             // {chain_pending, chain_buf, chain_U, event tag, column blocks} at the start of the launch, the workgroup's index
    double blk[MAX_S * MAX_S], sxl[MAX_S], tl[MAX_S];
};

constexpr int ST_NW = (int)(sizeof(State) / 4);
static_assert(sizeof(State) % 4 == 0, "State must be a whole number of dwords");
__device__ __forceinline__ void state_in(State& lst, const State* gst) {
    const uint32_t* src = reinterpret_cast<const uint32_t*>(gst);
    uint32_t* dst = reinterpret_cast<uint32_t*>(&lst);
    for (int i = threadIdx.x; i < ST_NW; i += blockDim.x) dst[i] = src[i];
}
__device__ __forceinline__ void state_out(State* gst, const State& lst) {
    const uint32_t* src = reinterpret_cast<const uint32_t*>(&lst);
    uint32_t* dst = reinterpret_cast<uint32_t*>(gst);
    for (int i = threadIdx.x; i < ST_NW; i += blockDim.x) dst[i] = src[i];
}

__device__ __forceinline__ Cand best_bcast(Cand b) {  // lane 0's candidate to the whole wave
    Cand r;
    r.q = __shfl(b.q, 0, 64);
    r.key = (uint64_t)__shfl((unsigned long long)b.key, 0, 64);
    r.si = __shfl(b.si, 0, 64);
    r.sj = __shfl(b.sj, 0, 64);
    return r;
}

// All threads of a 1024-thread workgroup call this; S.lst holds the control block (visible to the whole
// workgroup); `best` is valid in thread 0.  d.st is NOT used: dl.st points at the LDS copy.
// Helper workgroups for the exact ComputeRx sums (k_track only; nhelp = 0 elsewhere).  On tie-rich inputs (additive trees,
// integer-valued distances) the 4-candidate choice can rarely be certified, and the <= 4 exact sequential sums - one after the
// other in the deciding workgroup, ~30 us each - made an event 157 us long (profiles/r03/r03_input_classes.md: 4.8 s instead of
// 1.4 s at 32768 taxa).  k_track is launched with four more workgroups that wait for a JOB word: the deciding workgroup posts
// {launch tag | which sums | job flag} once per launch - "no job" as soon as the choice is certified (or the window cannot
// serve the event, or the launch returns early), so the helpers leave long before the kernel ends - or "job" behind its six
// parameter words (the four nodes' slots, m, 2P: write-through, drained): helper b gathers the ComputeRx terms of node b into its
// own buffer, computes their exact sequential sum with the fast form, stores the result write-through and counts its arrival;
// the deciding workgroup then reads the results past its L1.  Every wait has a deadline (error 14, never a hang).  The helpers are launched only while
// the run needs them (HipBackend::note_rx_exact): a run on distances without ties never sees them.
constexpr int TRK_JOB = 32 * 68, TRK_JOBDONE = 32 * 69, TRK_JOBM = 32 * 70;  // words of d.ticket (32 * 72 words; 65, 67: TRK_FLAG, TRK_BAD); JOBM: 6 parameter words
constexpr int TRK_NHELP = 4;
constexpr int RX_RES = 2048;  // d.rchk[RX_RES + b]: bit pattern of helper b's sum (behind the 2 x 1024 record check words)
constexpr long long TRK_WAIT_TICKS = 200000000LL;  // 2 s of the 100 MHz wall clock
__device__ __forceinline__ void job_post(const Dev& d, unsigned tag, unsigned low5) {
    __hip_atomic_store(d.ticket + TRK_JOB, (tag << 5) | (low5 & 31u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void rx_helper_workgroup(const Dev& d, ChainLds<CH_EPT>& L, int b, unsigned tag, unsigned* hword) {
    if (threadIdx.x == 0) {
        const long long t0 = (long long)wall_clock64();
        unsigned w = 0;
        for (;;) {
            w = __hip_atomic_load(d.ticket + TRK_JOB, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if ((w >> 5) == tag) break;
            if ((long long)wall_clock64() - t0 > TRK_WAIT_TICKS) { w = 0; break; }  // (the deciding workgroup reports it)
            __builtin_amdgcn_s_sleep(2);
        }
        *hword = w;
    }
    __syncthreads();
    const unsigned w = *hword;
    if (!(w & 1u) || !((w >> (1 + b)) & 1u)) return;
    // the job's parameters (write-through stores of the deciding workgroup, drained before the JOB word went out)
    int32_t z[4];
#pragma unroll
    for (int k = 0; k < 4; k++) z[k] = (int32_t)__hip_atomic_load(d.ticket + TRK_JOBM + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int32_t m_old = (int32_t)__hip_atomic_load(d.ticket + TRK_JOBM + 4, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int32_t twoP_old = (int32_t)__hip_atomic_load(d.ticket + TRK_JOBM + 5, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // this helper's row: the ComputeRx terms of node z[b] in reference position order (rx_fill_thread, one buffer), then their
    // exact sequential sum.  The matrix and the slot tables do not change during this launch (k_update writes them).
    const int32_t only[4] = {b == 0 ? z[0] : -1, b == 1 ? z[1] : -1, b == 2 ? z[2] : -1, b == 3 ? z[3] : -1};
    for (int32_t sl = threadIdx.x; sl < m_old; sl += blockDim.x) rx_fill_one(d, sl, m_old, twoP_old, z, b, only[b]);
    __threadfence_block();
    __syncthreads();
    const double r = block_chain_sum2<CH_EPT>(d.chain + (size_t)(b + 1) * d.cstride, m_old, CH_GUARD_BITS, L, nullptr);
    if (threadIdx.x == 0) {
        __hip_atomic_store(d.rchk + RX_RES + b, __builtin_bit_cast(uint64_t, r), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        atomicAdd(d.ticket + TRK_JOBDONE, 1u);
    }
}

// HELP: the kernel was launched with the helper workgroups (its own instantiation: the variant without them carries none of
// that code - inlined into k_track it cost the hot path ten spilled registers and 2 % of the run on inputs without ties)
template <bool HELP>
__device__ __forceinline__ void decide_step(const Dev& d, DecideLds& S, ChainLds<CH_EPT>& L, Cand best, unsigned jobtag) {
    State& lst = S.lst;
    Dev dl = d;
    dl.st = &lst;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
#define DEC_TICK(slot) do { if (tid == 0 && S.tkon) S.dk[slot] = (long long)wall_clock64(); } while (0)
    DEC_TICK(0);
    if (tid == 0) {
        S.need = pick_needs_candidate(lst) ? 1 : 0;
        S.ab[0] = best.si;   // the slots of the candidate's two nodes travel with the candidate
        S.ab[1] = best.sj;
        S.cert = 1;
        S.exact = 0;
    }
    __syncthreads();
    const bool need = S.need != 0;
    const int32_t m = lst.m, P = lst.P;
    const int32_t tpU = lst.tp_n > 0 ? lst.tp_U : -2;
    if (tid == 0) S.msg_pU = tpU;
    // ONE round trip: wave 0 fetches the table entries the plan can touch, the 4 x 4 block of the matrix over the two
    // clusters' nodes and their T; wave 1 sums the partial sums of T of the previous event's new cluster
    if (wv == 0) {
        if (lane >= 40 && lane < 40 + TAB_NP) {
            const int32_t pk = m - 1 - (lane - 40);
            S.pkey[lane - 40] = pk >= 0 ? pk : -1;
            S.vpslot[lane - 40] = pk >= 0 ? d.pslot[pk] : 0;
        }
        if (need) {
            const int32_t a = S.ab[0], b = S.ab[1];
            int32_t key[TAB_NK], pkey[TAB_NP];
            tab_keys(a, b, P, m, key, pkey);
            const int32_t s4[4] = {a, a ^ 1, b, b ^ 1};
            if (lane < TAB_NK) {
                int32_t k = -1;
#pragma unroll
                for (int q = 0; q < TAB_NK; q++) if (q == lane) k = key[q];
                if (k < 0 || k >= d.n) k = -1;
                S.key[lane] = k;
                S.vsid[lane] = k >= 0 ? d.sid[k] : 0;
                S.vspos[lane] = k >= 0 ? d.spos[k] : 0;
            } else if (lane >= 16 && lane < 32) {
                const int r = (lane - 16) >> 2, c = (lane - 16) & 3;
                int32_t sr = 0, sc = 0;
#pragma unroll
                for (int q = 0; q < 4; q++) { if (q == r) sr = s4[q]; if (q == c) sc = s4[q]; }
                S.quad[lane - 16] = d.D[(int64_t)sr * d.ld + sc];
            } else if (lane >= 32 && lane < 36) {
                int32_t sl = 0;
#pragma unroll
                for (int q = 0; q < 4; q++) if (q == lane - 32) sl = s4[q];
                S.tz[lane - 32] = (sl == tpU || sl == tpU + 1) ? 0.0 : d.T[sl];  // (the newest cluster's T: from wave 1, below)
            }
        }
    } else if (wv == 1) {
        const int np = lst.tp_n;
        if (np > 0) {
            double tu = 0.0, tv = 0.0;
            for (int g = lane; g < np; g += 64) {
                const double2 v = *reinterpret_cast<const double2*>(d.upart + 4 * g + 2);
                tu += v.x; tv += v.y;
            }
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) { tu += __shfl_down(tu, off, 64); tv += __shfl_down(tv, off, 64); }
            if (lane == 0) {
                S.tfin[0] = tu; S.tfin[1] = tv;
                // write-through: in the fused event kernel a column thread of the same launch (another compute unit, maybe another L2
                // slice) stores to the same address later; two dirty copies would be written back in an undefined order
                __hip_atomic_store(reinterpret_cast<uint64_t*>(d.T + lst.tp_U), __builtin_bit_cast(uint64_t, tu), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(reinterpret_cast<uint64_t*>(d.T + lst.tp_U + 1), __builtin_bit_cast(uint64_t, tv), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
    __syncthreads();
    if (tid == 64) lst.tp_n = 0;
    if (tid < 4 && need && tpU >= 0) {  // T of the newest cluster's nodes, where they are among the four
        const int32_t sl = (tid & 2 ? S.ab[1] : S.ab[0]) ^ (tid & 1);
        if (sl == tpU) S.tz[tid] = S.tfin[0];
        else if (sl == tpU + 1) S.tz[tid] = S.tfin[1];
    }
    __syncthreads();
    DEC_TICK(1);
    // Cx / Cy, then the certified choice: by wave 0, every lane computing the same (wave-uniform control)
    Quad& qd = S.qd;
    if (wv == 0) {
        const int32_t a = need ? S.ab[0] : 0, b = need ? S.ab[1] : 0;
        pick(dl, best_bcast(best), a, b, need ? S.vsid[0] : 0, need ? S.vsid[2] : 0);
        if (lst.ev_active && !lst.ev_finish) {
            // (pick may have exchanged the two nodes: Cx is the one with the smaller id)
            const int oa = lst.sa == a ? 0 : 2, ob = 2 - oa;
            auto q4 = [&](int r, int c) { return S.quad[r * 4 + c]; };
            qd.Tz[0] = S.tz[oa]; qd.Tz[1] = lst.sap >= 0 ? S.tz[oa + 1] : 0.0;
            qd.Tz[2] = S.tz[ob]; qd.Tz[3] = lst.sbp >= 0 ? S.tz[ob + 1] : 0.0;
            qd.Dab = q4(oa, ob);
            qd.Dapb = lst.sap >= 0 ? q4(oa + 1, ob) : 0.0;
            qd.Dabp = lst.sbp >= 0 ? q4(oa, ob + 1) : 0.0;
            qd.Dapbp = (lst.sap >= 0 && lst.sbp >= 0) ? q4(oa + 1, ob + 1) : 0.0;
            qd.Daap = lst.sap >= 0 ? q4(oa, oa + 1) : 0.0;
            qd.Dbbp = lst.sbp >= 0 ? q4(ob, ob + 1) : 0.0;
            double rx[4] = {0.0, 0.0, 0.0, 0.0};
            if (lst.need_rx) {
                rx_from_T(lst, qd, rx);
                if (rx_certify(lst, qd, rx)) lst.n_rx_certified++;
                else { S.cert = 0; lst.n_rx_exact++; }
            }
            S.rx[0] = rx[0]; S.rx[1] = rx[1]; S.rx[2] = rx[2]; S.rx[3] = rx[3];
        }
    }
    __syncthreads();
    DEC_TICK(2);
    const bool want_exact = lst.ev_active && !lst.ev_finish && !S.cert;
    if (HELP && !want_exact && tid == 0) job_post(d, jobtag, 0u);  // the helpers may leave
    if (!lst.ev_active) return;  // the loop has ended
    if (lst.ev_finish) {         // the special finish: planned inside pick; only the symbolic replay is left
        if (tid < 64) build_targets_wave(lst);
        __syncthreads();
        return;
    }
    if (!S.cert) {
        // rare: the <= 4 sums exactly, one after the other in this workgroup; their addends go to the chain buffers first
        const int32_t z[4] = {lst.sa, lst.sap, lst.sb, lst.sbp};
        const int32_t m_old = lst.m_old, twoP_old = 2 * lst.P_old;
        if (HELP) {
            // the four sums by the four helper workgroups, side by side: each gathers its own row and sums it
            const unsigned mask = (z[0] >= 0 ? 1u : 0u) | (z[1] >= 0 ? 2u : 0u) | (z[2] >= 0 ? 4u : 0u) | (z[3] >= 0 ? 8u : 0u);
            if (tid == 0) {
                const int32_t par[6] = {z[0], z[1], z[2], z[3], m_old, twoP_old};
#pragma unroll
                for (int k = 0; k < 6; k++) __hip_atomic_store(d.ticket + TRK_JOBM + k, (unsigned)par[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the parameters are on their way out before the JOB word)
                job_post(d, jobtag, 1u | (mask << 1));
            }
            if (tid == 0) {
                const unsigned want = (unsigned)__builtin_popcount(mask);
                const long long t0 = (long long)wall_clock64();
                while (__hip_atomic_load(d.ticket + TRK_JOBDONE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != want) {
                    if ((long long)wall_clock64() - t0 > TRK_WAIT_TICKS) { lst.error = 14; break; }
                    __builtin_amdgcn_s_sleep(2);
                }
                __hip_atomic_store(d.ticket + TRK_JOBDONE, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                for (int b = 0; b < 4; b++)
                    S.rx[b] = ((mask >> b) & 1u) ? __builtin_bit_cast(double, __hip_atomic_load(d.rchk + RX_RES + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) : 0.0;
            }
            __syncthreads();
        } else {
        for (int32_t sl = tid; sl < m_old; sl += blockDim.x) rx_fill_thread(d, sl, m_old, twoP_old, z);
        __threadfence_block();
        __syncthreads();
#pragma unroll
        for (int b = 0; b < 4; b++) {
            double r = 0.0;
            if (z[b] >= 0) r = block_chain_sum_rare<CH_EPT>(d.chain + (size_t)(b + 1) * d.cstride, m_old, CH_GUARD_BITS, &L, nullptr);  // (buffer 0 belongs to the pending row sum)
            if (tid == 0) S.rx[b] = r;
            __syncthreads();
        }
        }
    }
    if (wv == 0) {
        WaveTab T{d, lane, lane < TAB_NK ? S.key[lane] : -1, lane < TAB_NK ? S.vsid[lane] : 0, lane < TAB_NK ? S.vspos[lane] : 0,
                  lane < TAB_NP ? S.pkey[lane] : -1, lane < TAB_NP ? S.vpslot[lane] : 0, 0};
        const double rx[4] = {S.rx[0], S.rx[1], S.rx[2], S.rx[3]};
        decide_plan(dl, T, qd, rx);
        if (lane == 0) S.misses = T.misses;
    }
    __syncthreads();
    DEC_TICK(3);
    if (tid < 64) build_targets_wave(lst);
    __syncthreads();
    DEC_TICK(4);
    if (tid == 0 && S.tkon) {
        for (int q = 0; q < 4; q++) d.ticks[16 + q] += S.dk[q + 1] - S.dk[q];
    }
#undef DEC_TICK
}

// ------------------------------------------------------------------ k_relaxed
// Relaxed mode (fnn_core.h "Relaxed mode"): the search for a pair of mutual row minima, one workgroup.  Wave 0 runs
// relaxed_find with wave-uniform control (memory is read by lane 0 and broadcast, written by lane 0); a row minimum
// is a command to the whole workgroup: every thread evaluates Q(p, .) at its positions, the minimum is reduced, a
// second pass collects the positions that attain it (usually the two nodes of one cluster), lane 0 sorts them into
// position order and files the list under p.  The other waves loop on the command word.
constexpr int RL_T = 1024;
struct RlLds {
    int32_t cmd;               // 1: row minimum of slot ps (partner pp); 2: done
    int32_t ps, pp, err;
    int32_t tcnt;
    int32_t tpos[RL_TIES], tslot[RL_TIES];
    double wmin[RL_T / 64];
    double gmin;
    int32_t kme[RL_MINS], krow[RL_MINS];
    double kval[RL_MINS];
    int32_t tkon;
    long long tk[4];           // diagnostics (FNN_TICKS=1): ticks in {row-minimum pass, its finish, the rest}, row minima
    int32_t wg, nwg;           // this workgroup's share of the row pass: pair chunks wg, wg + nwg, ... (nwg = 1: all)
    uint32_t seq, base;        // control workgroup: commands published so far, arrival counter at the start
};

// rl_q (fnn_core.h) without branches: the four entries are loaded whatever the kinds of p and q (the partner column
// q ^ 1 and the partner row exist in the padded matrix; unused sums are discarded), so that the loads of several
// slots are in flight together.  Same operations on the same operands for the case that applies.
__device__ __forceinline__ double rl_q_flat(const double* Rp, const double* Rn, const double* Sx, bool ppair, int32_t twoP,
                                            int32_t qs, double cm2, double sxp) {
    const double a = Rp[qs], b = Rp[qs ^ 1], c = Rn[qs], e = Rn[qs ^ 1], sq = Sx[qs];
    const bool qpair = qs < twoP;
    const double t01 = (a + b) / 2.0, t10 = (a + c) / 2.0, t11 = (((a + b) + c) + e) / 4.0;
    const double Dpq = ppair ? (qpair ? t11 : t10) : (qpair ? t01 : a);
    return (cm2 * Dpq - sxp) - sq;
}

__device__ __forceinline__ void rl_rowmin_block(const Dev& d, RlLds& L) {
    const State* st = d.st;
    const int32_t m = st->m, twoP = 2 * st->P, ps = L.ps, pp = L.pp;
    const double cm2 = (double)st->c - 2.0, sxp = d.Sx[ps];
    const int tid = threadIdx.x;
    // ONE pass, over the live slots [0, m) instead of the positions (a minimum does not depend on the order; the row
    // of p is then read contiguously): every thread keeps its minimum, the first slot that attains it and how many do
    // (the second pass below, for a thread with several, is rare: the two nodes of a cluster - the usual tie - are
    //  one thread's pair, so that thread goes over its slots again)
    double mine = 1.7976931348623157e308;  // Double.MAX_VALUE (:95)
    int32_t mslot = -1, mslot2 = -1, mcnt = 0;  // (the first two slots that attain it: the two nodes of a cluster tie)
    const double* Rp = d.D + (int64_t)ps * d.ld;
    const double* Rn = d.D + (int64_t)(pp >= 0 ? pp : ps) * d.ld;
    const bool ppair = pp >= 0;
    // A thread takes slot PAIRS (2j, 2j + 1): the pair's entries in p's row, in the partner's row and its two row sums
    // are three 16-byte loads; eight pairs per round, all 24 loads issued before the first value is used (left to
    // itself the compiler sinks the partner row's loads into the branch that needs them: two dependent round trips per
    // slot).  One round covers 16 384 slots; a round costs about one HBM latency (~4 us) whatever it loads.
    // (several workgroups: chunks of RL_T pairs are dealt out round-robin, workgroup wg takes chunks wg, wg + nwg, ...)
    constexpr int RL_PAIRS = 8;
    const int32_t npairs = (m + 1) >> 1;
    const int32_t cstep = L.nwg * RL_T, cfirst = L.wg * RL_T;
    for (int32_t j0 = cfirst + tid; j0 - tid < npairs; j0 += RL_PAIRS * cstep) {
        double2 A[RL_PAIRS], Cc[RL_PAIRS], S[RL_PAIRS];
        const int32_t nu = (npairs - (j0 - tid) + cstep - 1) / cstep;  // (uniform: the chunks this round still has)
#pragma unroll
        for (int u = 0; u < RL_PAIRS; u++) {
            A[u] = Cc[u] = S[u] = make_double2(0.0, 0.0);
            if (u < nu) {
                const int32_t j = j0 + u * cstep, jj = j < npairs ? j : npairs - 1;
                A[u] = *reinterpret_cast<const double2*>(Rp + 2 * jj);
                Cc[u] = *reinterpret_cast<const double2*>(Rn + 2 * jj);
                S[u] = *reinterpret_cast<const double2*>(d.Sx + 2 * jj);
            }
        }
        asm volatile("" ::: "memory");
#pragma unroll
        for (int u = 0; u < RL_PAIRS; u++) {
            const int32_t j = j0 + u * cstep;
            if (j >= npairs) continue;
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const int32_t qs = 2 * j + h;
                if (qs >= m || qs == ps || qs == pp) continue;
                const bool qpair = qs < twoP;
                const double a = h ? A[u].y : A[u].x, b = h ? A[u].x : A[u].y;      // D[p][q], D[p][q.nbr]
                const double c = h ? Cc[u].y : Cc[u].x, e = h ? Cc[u].x : Cc[u].y;  // D[p.nbr][q], D[p.nbr][q.nbr]
                const double t01 = (a + b) / 2.0, t10 = (a + c) / 2.0, t11 = (((a + b) + c) + e) / 4.0;
                const double Dpq = ppair ? (qpair ? t11 : t10) : (qpair ? t01 : a);
                const double q = (cm2 * Dpq - sxp) - (h ? S[u].y : S[u].x);
                if (q < mine) { mine = q; mslot = qs; mcnt = 1; }
                else if (q == mine) { if (mcnt == 0) mslot = qs; else if (mcnt == 1) mslot2 = qs; mcnt++; }
            }
        }
    }
    double wm = mine;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const double o = __shfl_xor(wm, off, 64);
        if (o < wm) wm = o;
    }
    if ((tid & 63) == 0) L.wmin[tid >> 6] = wm;
    if (tid == 0) L.tcnt = 0;
    __syncthreads();
    double g = L.wmin[0];
#pragma unroll
    for (int w = 1; w < RL_T / 64; w++) { const double o = L.wmin[w]; if (o < g) g = o; }
    // the rows that attain it, with their positions (the list is in position order, :98); a thread with several
    // (usually the two nodes of one cluster, which are one thread's pair) goes over its slots again
    if (mcnt > 0 && mine == g) {
        if (mcnt <= 2) {
            const int k = atomicAdd(&L.tcnt, mcnt);
            if (k < RL_TIES) { L.tpos[k] = d.spos[mslot]; L.tslot[k] = mslot; }
            if (mcnt == 2 && k + 1 < RL_TIES) { L.tpos[k + 1] = d.spos[mslot2]; L.tslot[k + 1] = mslot2; }
        } else {
            for (int32_t j = cfirst + tid; j < npairs; j += cstep)  // this thread's own pairs again (same expression, same bits)
                for (int h = 0; h < 2; h++) {
                    const int32_t qs = 2 * j + h;
                    if (qs >= m || qs == ps || qs == pp) continue;
                    const double q = rl_q_flat(Rp, Rn, d.Sx, ppair, twoP, qs, cm2, sxp);
                    if (q == g) {
                        const int k = atomicAdd(&L.tcnt, 1);
                        if (k < RL_TIES) { L.tpos[k] = d.spos[qs]; L.tslot[k] = qs; }
                    }
                }
        }
    }
    if (tid == 0) L.gmin = g;
    __syncthreads();
}

// Several workgroups on one row minimum (the row pass is bound by what ONE CU can stream, ~40 GB/s): the control
// workgroup publishes the command as ONE 8-byte write-through store {event stamp | sequence number | slot of p}; a worker
// workgroup polls that word past its L1, takes its share of the pair chunks, writes its record {minimum, count | sequence
// number, (position | slot) of the rows that attain it} write-through, waits for the acknowledgement and counts its
// arrival; the control wave polls the counter and reads the records past its L1.  (MI355X_MICROARCH.md "valid forms":
// an 8-byte granule needs no ordering; every handed-off byte stored sc1 and drained before the counter, every load of
// it sc1, by the wave whose poll matched.)  Every wait is bounded: a worker that never sees its command leaves, the
// control wave then reports error 23 instead of hanging.
constexpr uint32_t RL_EXIT = 0xFFFFFu;
constexpr long long RL_WAIT_TICKS = 200000000LL;  // deadline of every wait, in ticks of the 100 MHz wall clock (2 s): a count of polls would
                                                  // turn a late-scheduled workgroup on a shared or profiled GPU into a spurious error 23
__device__ __forceinline__ uint64_t rl_cmd_word(uint32_t stamp, uint32_t seq, uint32_t ps) {
    return ((uint64_t)(stamp & 0xFFFFFFu) << 40) | ((uint64_t)(seq & 0xFFFFFu) << 20) | (uint64_t)(ps & 0xFFFFFu);
}
__device__ __forceinline__ uint64_t ld_sc1(const uint64_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_sc1(uint64_t* p, uint64_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ uint64_t* rl_rec(const Dev& d, int w) { return d.rl_mail + 32 + 32 * w; }

// a worker workgroup's record of the row minimum it has just taken part in (thread 0, after rl_rowmin_block)
// 16-bit digests that travel in a record's head word, so that a value or tie list that does not belong to this command
// (not observed; see rec_publish) is reported instead of used
__device__ __forceinline__ uint32_t rl_h16(uint64_t x) {
    x *= 0x9E3779B97F4A7C15ULL;
    return (uint32_t)(x >> 48);
}
__device__ __forceinline__ uint64_t rl_head(int32_t cnt, uint32_t seq, uint64_t vbits, uint64_t txor) {
    // (both digests mix the command's sequence number in: a stale value or tie list identical to the previous command's does not pass)
    return ((uint64_t)(uint32_t)(cnt & 0xFF) << 56) | ((uint64_t)(seq & 0xFFFFFFu) << 32) | ((uint64_t)rl_h16((txor + 1ULL) ^ ((uint64_t)seq << 1)) << 16) |
           (uint64_t)rl_h16(vbits ^ ((uint64_t)seq << 1));
}
__device__ __forceinline__ void rl_publish_part(const Dev& d, RlLds& L, uint32_t seq) {
    uint64_t* r = rl_rec(d, L.wg);
    int32_t cnt = L.tcnt;
    if (cnt > RL_TIES) cnt = RL_TIES + 1;  // (more ties than a list holds: the control wave reports it)
    const uint64_t vbits = __builtin_bit_cast(uint64_t, L.gmin);
    st_sc1(r + 0, vbits);
    uint64_t txor = 0;
    for (int i = 0; i < cnt && i < RL_TIES; i++) {
        const uint64_t e = ((uint64_t)(uint32_t)L.tpos[i] << 32) | (uint64_t)(uint32_t)L.tslot[i];
        txor ^= e * (uint64_t)(2 * i + 3);
        st_sc1(r + 2 + i, e);
    }
    st_sc1(r + 1, rl_head(cnt, seq, vbits, txor));
    __builtin_amdgcn_s_waitcnt(0);
    atomicAdd(reinterpret_cast<uint32_t*>(d.rl_mail + 16), 1u);
}

struct RlBlockEnv {
    RlLds& L;
    int lane;
    __device__ __forceinline__ bool lead() const { return lane == 0; }
    __device__ __forceinline__ int32_t err() const { return L.err; }
    __device__ __forceinline__ int32_t load(const int32_t* p) const {
        int32_t v = 0;
        if (lane == 0) v = *p;
        return __builtin_amdgcn_readfirstlane(v);
    }
    __device__ __forceinline__ double loadd(const double* p) const {
        uint64_t v = 0;
        if (lane == 0) v = __builtin_bit_cast(uint64_t, *p);
        const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
        const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32));
        return __builtin_bit_cast(double, ((uint64_t)hi << 32) | (uint64_t)lo);
    }
    __device__ __forceinline__ void store(int32_t* p, int32_t v) const { if (lane == 0) *p = v; }
    __device__ __forceinline__ void keep(int32_t i, int32_t me, int32_t row, double v) {
        if (lane == 0) { L.kme[i] = me; L.krow[i] = row; L.kval[i] = v; }
    }
    __device__ __forceinline__ Cand kept(const Dev& d, int32_t i) const {
        Cand c;
        c.q = L.kval[i]; c.si = L.kme[i]; c.sj = L.krow[i];
        c.key = ((uint64_t)(uint32_t)load(&d.spos[c.si]) << 32) | (uint64_t)(uint32_t)load(&d.spos[c.sj]);
        return c;
    }
    __device__ __forceinline__ void load3(const int32_t* pa, const int32_t* pb, const int32_t* pc, int32_t& a, int32_t& b, int32_t& c) const {
        int32_t x = 0, y = 0, z = 0;
        if (lane == 0) { x = *pa; y = *pb; z = *pc; }  // (issued together: one round trip)
        a = __builtin_amdgcn_readfirstlane(x); b = __builtin_amdgcn_readfirstlane(y); c = __builtin_amdgcn_readfirstlane(z);
    }
    __device__ __forceinline__ void load4(const int32_t* pa, const int32_t* pb, const int32_t* pc, const int32_t* pd, int32_t& a,
                                          int32_t& b, int32_t& c, int32_t& dd) const {
        int32_t x = 0, y = 0, z = 0, w = 0;
        if (lane == 0) { x = *pa; y = *pb; z = *pc; w = *pd; }
        a = __builtin_amdgcn_readfirstlane(x); b = __builtin_amdgcn_readfirstlane(y);
        c = __builtin_amdgcn_readfirstlane(z); dd = __builtin_amdgcn_readfirstlane(w);
    }
    __device__ __forceinline__ RlRow row(const Dev& d, int32_t key) const {  // a cached list: one round trip
        int32_t cnt = 0, l0 = 0, l1 = 0;
        uint64_t v = 0;
        if (lane == 0) {
            cnt = d.rl_cnt[key];
            v = __builtin_bit_cast(uint64_t, d.rl_val[key]);
            const int2 l = *reinterpret_cast<const int2*>(d.rl_list + (int64_t)key * RL_TIES);
            l0 = l.x; l1 = l.y;
        }
        RlRow r;
        r.me = key;
        r.cnt = __builtin_amdgcn_readfirstlane(cnt);
        r.l0 = __builtin_amdgcn_readfirstlane(l0);
        r.l1 = __builtin_amdgcn_readfirstlane(l1);
        const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
        const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32));
        r.value = __builtin_bit_cast(double, ((uint64_t)hi << 32) | (uint64_t)lo);
        return r;
    }
    __device__ __forceinline__ RlRow rowmin(const Dev& d, int32_t ps, int32_t pp, int32_t stamp) {
        long long t0 = 0, t1 = 0;
        if (L.tkon) t0 = (long long)wall_clock64();
        const int nwg = L.nwg;
        const uint32_t seq = L.seq + 1u;
        if (lane == 0) {
            L.cmd = 1; L.ps = ps; L.pp = pp; L.seq = seq;
            if (nwg > 1) st_sc1(d.rl_mail, rl_cmd_word((uint32_t)stamp, seq, (uint32_t)ps));  // the other workgroups' command
        }
        __syncthreads();          // the other waves pick the command up
        rl_rowmin_block(d, L);
        if (nwg > 1) {
            // the other workgroups' records: wait for their arrivals, then lane w reads the head of workgroup w's record
            int ok = 1;
            if (lane == 0) {
                const uint32_t want = L.base + (uint32_t)(nwg - 1) * seq;
                const long long t_wait = (long long)wall_clock64();
                while ((uint32_t)ld_sc1(d.rl_mail + 16) != want) {
                    if ((long long)wall_clock64() - t_wait > RL_WAIT_TICKS) { ok = 0; break; }
                    __builtin_amdgcn_s_sleep(1);
                }
            }
            ok = __builtin_amdgcn_readfirstlane(ok);
            double v = 1.7976931348623157e308;
            int32_t c = 0;
            uint32_t th = 0;  // digest of the record's tie list
            if (lane == 0) { v = L.gmin; c = L.tcnt > RL_TIES ? RL_TIES + 1 : L.tcnt; }
            if (ok && lane >= 1 && lane < nwg) {
                const uint64_t* r = rl_rec(d, lane);
                const uint64_t head = ld_sc1(r + 1);
                const uint64_t vbits = ld_sc1(r + 0);
                v = __builtin_bit_cast(double, vbits);
                c = (int32_t)(head >> 56);
                th = (uint32_t)(head >> 16) & 0xFFFFu;
                if ((uint32_t)((head >> 32) & 0xFFFFFFu) != (seq & 0xFFFFFFu) || ((uint32_t)head & 0xFFFFu) != rl_h16(vbits ^ ((uint64_t)seq << 1))) ok = 0;
            }
            ok = __ballot(!ok) ? 0 : 1;
            double g = v;
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) {
                const double o = __shfl_xor(g, off, 64);
                if (o < g) g = o;
            }
            // the merged list: this workgroup's own rows if they attain g, then the other workgroups' (any order: sorted below)
            const unsigned long long has = __ballot(lane < nwg && c > 0 && v == g);
            int32_t total = (has & 1ULL) ? __builtin_amdgcn_readfirstlane(c) : 0;
            if (total > RL_TIES) total = RL_TIES + 1;
            for (unsigned long long rest = has & ~1ULL; rest; rest &= rest - 1) {
                const int w = (int)__builtin_ctzll(rest);
                const int32_t cw = __builtin_amdgcn_readlane(c, w);
                if (cw > RL_TIES || total + cw > RL_TIES) { total = RL_TIES + 1; break; }
                uint64_t e = 0;
                if (lane < cw) {
                    e = ld_sc1(rl_rec(d, w) + 2 + lane);
                    L.tpos[total + lane] = (int32_t)(e >> 32);
                    L.tslot[total + lane] = (int32_t)(uint32_t)e;
                    e *= (uint64_t)(2 * lane + 3);
                }
#pragma unroll
                for (int off = 32; off >= 1; off >>= 1) e ^= (uint64_t)__shfl_xor((unsigned long long)e, off, 64);
                if (rl_h16((e + 1ULL) ^ ((uint64_t)seq << 1)) != (uint32_t)__builtin_amdgcn_readlane((int)th, w)) ok = 0;
                total += cw;
            }
            if (lane == 0) {
                L.tcnt = total;
                L.gmin = g;
                if (!ok && !L.err) L.err = 23;
            }
        }
        if (L.tkon) t1 = (long long)wall_clock64();
        if (lane == 0) {
            int32_t cnt = L.tcnt;
            if (cnt > RL_TIES) { L.err = 20; cnt = RL_TIES; L.tcnt = cnt; }
            for (int i = 1; i < cnt; i++) {  // position order (:98 walks the rows in order)
                const int32_t kp = L.tpos[i], ks = L.tslot[i];
                int j = i - 1;
                while (j >= 0 && L.tpos[j] > kp) { L.tpos[j + 1] = L.tpos[j]; L.tslot[j + 1] = L.tslot[j]; j--; }
                L.tpos[j + 1] = kp; L.tslot[j + 1] = ks;
            }
            for (int i = 0; i < cnt; i++) d.rl_list[(int64_t)ps * RL_TIES + i] = L.tslot[i];
            d.rl_stamp[ps] = stamp;
            d.rl_cnt[ps] = cnt;
            d.rl_val[ps] = L.gmin;
        }
        if (L.tkon && lane == 0) { L.tk[0] += t1 - t0; L.tk[1] += (long long)wall_clock64() - t1; L.tk[3]++; }
        // the result straight from LDS (the same wave wrote it: LDS operations of a wave complete in order)
        RlRow r;
        r.me = ps;
        r.cnt = L.tcnt;
        r.value = L.gmin;
        r.l0 = L.tslot[0];
        r.l1 = L.tslot[1];
        return r;
    }
};

__global__ __launch_bounds__(RL_T) void k_relaxed(Dev d, int ticks) {
    __shared__ RlLds L;
    State* st = d.st;
    const long long tstart = ticks ? (long long)wall_clock64() : 0;
    // (uniform over the whole grid: every thread reads the same words; the control workgroup writes to the control block
    //  only after every other workgroup has arrived for the last row minimum, i.e. after they have all read it)
    const bool run = !st->done && st->rl_on && st->m > st->rl_min && st->m > 3 && !(st->m == 4 && st->c == 2);
    const uint32_t stamp = (uint32_t)st->n_events + 1u;
    if (!run) {
        if (blockIdx.x == 0 && threadIdx.x == 0) st->rl_active = 0;
        return;
    }
    if (threadIdx.x == 0) {
        L.err = 0; L.cmd = 0; L.tkon = blockIdx.x == 0 ? ticks : 0; L.tk[0] = L.tk[1] = L.tk[2] = L.tk[3] = 0;
        L.wg = (int32_t)blockIdx.x; L.nwg = (int32_t)gridDim.x; L.seq = 0;
        L.base = gridDim.x > 1 && blockIdx.x == 0 ? (uint32_t)ld_sc1(d.rl_mail + 16) : 0u;  // (nobody arrives before the first command)
    }
    __syncthreads();
    if (blockIdx.x != 0) {
        // a worker workgroup: commands until the exit command
        for (uint32_t seq = 1;; seq++) {
            if (threadIdx.x == 0) {
                const long long t_wait = (long long)wall_clock64();
                uint64_t w;
                for (;;) {
                    w = ld_sc1(d.rl_mail);
                    if ((uint32_t)(w >> 40) == (stamp & 0xFFFFFFu) && (uint32_t)((w >> 20) & 0xFFFFFu) == (seq & 0xFFFFFu)) break;
                    if ((long long)wall_clock64() - t_wait > 4 * RL_WAIT_TICKS) { w = RL_EXIT; break; }  // (never seen: leave; the control wave reports it)
                    __builtin_amdgcn_s_sleep(1);
                }
                const uint32_t ps = (uint32_t)(w & 0xFFFFFu);
                L.cmd = ps == RL_EXIT ? 2 : 1;
                L.ps = (int32_t)ps;
                L.pp = (int32_t)ps < 2 * st->P ? (int32_t)(ps ^ 1u) : -1;
            }
            __syncthreads();
            if (L.cmd == 2) break;
            rl_rowmin_block(d, L);
            if (threadIdx.x == 0) rl_publish_part(d, L, seq);
        }
        return;
    }
    if (threadIdx.x < 64) {
        RlBlockEnv env{L, (int)threadIdx.x};
        const Cand out = relaxed_find(d, env);
        if (threadIdx.x == 0) {
            d.recs[0] = out;
            L.cmd = 2;
            if (gridDim.x > 1) st_sc1(d.rl_mail, rl_cmd_word(stamp, L.seq + 1u, RL_EXIT));
        }
        if (ticks && threadIdx.x == 0) {
            d.ticks[24] += L.tk[0]; d.ticks[25] += L.tk[1]; d.ticks[26] += (long long)wall_clock64() - tstart; d.ticks[27] += L.tk[3];
        }
        __syncthreads();
    } else {
        for (;;) {
            __syncthreads();
            if (L.cmd == 2) break;
            rl_rowmin_block(d, L);
        }
    }
}

// ------------------------------------------------------------------ k_decide
// the decide step of an event that scanned: the scan's (or all ranks') candidate records are reduced first
template <bool HELP>
__global__ __launch_bounds__(CH_T) void k_decide(Dev d, const Cand* src, int nrecs, unsigned jobtag) {
    __shared__ ChainLds<CH_EPT> L;
    __shared__ DecideLds S;
    __shared__ Cand shc[CH_T / 64];
    __shared__ unsigned hword;
    State* st = d.st;
    if (HELP && blockIdx.x > 0) {  // a helper workgroup of the exact ComputeRx sums (see decide_step)
        rx_helper_workgroup(d, L, (int)blockIdx.x - 1, jobtag, &hword);
        return;
    }
    if (st->stall || st->la_hit) {  // (nothing to decide / the tail of k_track has decided this event already)
        if (HELP && threadIdx.x == 0) job_post(d, jobtag, 0u);
        return;
    }
    state_in(S.lst, st);
    if (threadIdx.x == 0) S.tkon = 0;
    Cand best = reduce_records(d, src, st->rl_active ? 1 : nrecs, shc);  // (Relaxed mode: the search's one record)
    __syncthreads();
    decide_step<HELP>(d, S, L, best, jobtag);
    __syncthreads();
    if (threadIdx.x == 0) S.lst.rl_active = 0;
    __syncthreads();
    state_out(st, S.lst);
}

// ------------------------------------------------------------------ k_track
// k_track: serve the event from the open lookahead window (fnn_core.h "Lookahead").  The work
// items (tracked pairs, then the sweep of the newest cluster's rows) are spread over the track
// workgroups; the last one to arrive reduces the per-workgroup minima and decides whether the
// window certifies the minimum (la_hit: the scan kernels of this event return at once) or the
// event has to scan.  force_base: the host's schedule asks for a new window at this event.
//
// Workgroup 0 is the CHAIN workgroup: when the previous event's k_update left the new cluster's
// exact sequential row sum to be computed (chain_pending; ~20 us of one workgroup), it is computed
// here, BESIDE the tracking.  Only the sweep of that cluster's own rows needs the sum: it runs on
// the tree-ordered sum of k_update's partials, and its pairs compete in a separate record.  Both
// summation orders are within eps of the exact sum, so if the best swept pair lies further than
// the margin above the best other pair, the winner - an exactly evaluated pair - is certain.
// Otherwise the last workgroup waits for the chain workgroup and sweeps again with the exact sum.
constexpr int TRK_THREADS = 1024;
constexpr int TRK_GROUP = 16;   // arrival tickets in two levels: same-address device-scope atomics cost ~50 ns each
constexpr int TRK_FLAG = 32 * 65;  // word of d.ticket that carries "chain of event # done"
constexpr int TRK_BAD = 32 * 67;   // word of d.ticket: a sweep item did not find the cluster it expected
constexpr int TRK_ERR = 32 * 66;   // word of d.ticket: a wait inside the fused event kernel ran into its deadline (reported by the next launch)

__device__ __forceinline__ void chain_workgroup(const Dev& d, ChainLds<CH_EPT>& L) {
    State* st = d.st;
    if (!st->chain_pending) return;
    // (everything this workgroup needs from the control block is read NOW: in the fused event kernel the deciding workgroup
    //  closes the event and writes the control block back while this sum may still be running)
    const int32_t cU = st->chain_U, cm = st->chain_m, cb = st->chain_buf;
    const unsigned evtag = (unsigned)st->n_events;
    const double usx = block_chain_sum2<CH_EPT>(d.chain + (size_t)cb * d.cstride, cm, CH_GUARD_BITS, L, nullptr);
    if (threadIdx.x == 0) {
        // u.Sx and u.nbr.Sx (NetMakerOriginal.java:532, 535): write-through, drained, then the flag - column threads of the
        // SAME launch may be waiting for them (fused event kernel); every later kernel sees them anyway
        __hip_atomic_store(reinterpret_cast<uint64_t*>(d.Sx + cU), __builtin_bit_cast(uint64_t, usx), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(reinterpret_cast<uint64_t*>(d.Sx + cU + 1), __builtin_bit_cast(uint64_t, usx), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __threadfence();
        __hip_atomic_store(d.ticket + TRK_FLAG, evtag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// exact evaluation of the sweep items only, without inserting pairs (the approximate sweep did that)
__device__ __forceinline__ void sweep_exact_item(const Dev& d, int64_t r, const TrackArgs& ta, Cand& bx) {
    const int32_t half = (ta.m + 1) / 2;
    const int32_t fi = ta.nf0 + (int32_t)(r / half), cp = (int32_t)(r % half);
    if (fi >= ta.nf) return;
    const int32_t f0 = fresh_slot(d, fi, ta.m);
    if (f0 < 0) return;
    const int32_t s2 = 2 * cp;
    if (s2 >= ta.m || s2 == f0) return;
    const double* F0 = d.D + (int64_t)f0 * d.ld + s2;
    const double* F1 = F0 + d.ld;
    const double a0 = F0[0], a1 = F0[1], b0 = F1[0], b1 = F1[1];
    if (f0 > s2)
        scan_micro(f0, s2, ta.m, ta.twoP, ta.cm2, a0, a1, b0, b1, d.Sx[f0], d.Sx[f0 + 1], d.spos[f0], d.spos[f0 + 1],
                   d.Sx[s2], d.Sx[s2 + 1], d.spos[s2], d.spos[s2 + 1], bx);
    else
        scan_micro(s2, f0, ta.m, ta.twoP, ta.cm2, a0, b0, a1, b1, d.Sx[s2], d.Sx[s2 + 1], d.spos[s2], d.spos[s2 + 1],
                   d.Sx[f0], d.Sx[f0 + 1], d.spos[f0], d.spos[f0 + 1], bx);
}

constexpr int TRK_REC_U = 1024;  // offset of the swept-pair records in d.recs

struct SpecialOut { double val, tu, tv; };
// WGB: the phases are separated by workgroup barriers (k_update: the other waves of the workgroup keep them company); without
// it the calling wave is alone with the block - its LDS accesses execute in program order, a fence keeps the compiler from
// moving them - and no other wave of the workgroup has to take part (the fused event kernel)
template <bool WGB>
__device__ __forceinline__ void special_sync() {
    if (WGB) __syncthreads();
    else {  // (LDS instructions of one wave execute in order: only the compiler has to be kept from moving them - no s_waitcnt on the
            //  global stores and loads that are in flight, which a workgroup-scope fence would insert)
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}
template <bool WGB>
__device__ __forceinline__ SpecialOut special_wave(const Dev& d, const State& lst, double* blk, double* sxl, double* tl, int32_t* berr, double* chain_dst) {
    const int lane = threadIdx.x & 63;
    const int nS = __builtin_amdgcn_readfirstlane(lst.nS);
    const int32_t Sl = lane < nS ? lst.S[lane < MAX_S ? lane : 0] : -1;
    const int m_old = __builtin_amdgcn_readfirstlane(lst.m_old), twoP_old = 2 * __builtin_amdgcn_readfirstlane(lst.P_old);
    const int m = __builtin_amdgcn_readfirstlane(lst.m), twoP = 2 * __builtin_amdgcn_readfirstlane(lst.P);
    const int xs = __builtin_amdgcn_readfirstlane(lst.xs), ys = __builtin_amdgcn_readfirstlane(lst.ys);
    const int U = __builtin_amdgcn_readfirstlane(lst.U), V = U + 1;
    const int nops = __builtin_amdgcn_readfirstlane(lst.nops), ev_finish = __builtin_amdgcn_readfirstlane(lst.ev_finish);
    const int32_t myspos = (Sl >= 0 && Sl < m) ? d.spos[Sl] : 0;  // (new layout; for the chain addend of the add phase)
    auto ixu = [&](int slot) -> int {  // local index of a slot that is the same for all lanes (-1: not involved)
        const unsigned long long b = __ballot(Sl == slot && Sl >= 0);
        return b ? (int)__builtin_ctzll(b) : -1;
    };
    auto ixl = [&](int32_t slot) -> int {  // local index of a per-lane slot
        int r = -1;
#pragma unroll
        for (int j = 0; j < MAX_S; j++) if (__builtin_amdgcn_readlane(Sl, j) == slot && slot >= 0) r = j;
        return r;
    };
    const int me = lane;
    const int pold = (Sl >= 0 && Sl < twoP_old) ? ixl(Sl ^ 1) : -1;         // partner's column, old layout
    const int pnew = (Sl >= 0 && Sl < twoP && !(Sl & 1)) ? ixl(Sl + 1) : -1;  // ... of a representative, new layout
    auto B = [&](int r, int c) -> double& { return blk[r * MAX_S + c]; };
    auto bad = [&]() { *berr = 13; };
    SpecialOut out{0.0, 0.0, 0.0};
    // ---- subtract (old layout), NetMakerOriginal.java:455-461, 681-696
    if (!ev_finish) {
        const bool sp = Sl >= 0 && Sl < twoP_old;
        const bool act = Sl >= 0 && Sl < m_old && Sl != xs && Sl != ys && !(sp && (Sl & 1));
        double sx = 0.0, sxn = 0.0, told0 = 0.0, told1 = 0.0;
        bool bystander = true;
        if (act) { sx = sxl[me]; if (sp) { if (pold < 0) bad(); else sxn = sxl[pold]; } }
#pragma unroll
        for (int q = 0; q < 2; q++) {
            const int t = q == 0 ? xs : ys;
            const int tn = t < twoP_old ? (t ^ 1) : -1;
            const int it = ixu(t), itn = tn >= 0 ? ixu(tn) : -1;
            if (it < 0 || (tn >= 0 && itn < 0)) { bad(); continue; }
            if (!act) continue;
            if (Sl == t || Sl == tn) { bystander = false; continue; }
            double v;
            if (!sp && tn < 0) { const double e = B(it, me); v = e; told0 += e; }
            else if (sp && tn < 0) { const double e0 = B(it, me), e1 = B(it, pold); v = (e0 + e1) / 2.0; told0 += e0; told1 += e1; }
            else if (!sp && tn >= 0) { const double e0 = B(it, me), f0 = B(itn, me); v = (e0 + f0) / 2.0; told0 += 0.5 * (e0 + f0); }
            else {
                const double e0 = B(it, me), f0 = B(itn, me), e1 = B(it, pold), f1 = B(itn, pold);
                v = (((e0 + f0) + e1) + f1) / 4.0;
                told0 += 0.5 * (e0 + f0); told1 += 0.5 * (e1 + f1);
            }
            sx -= v;
            sxn -= v;
        }
        if (act) {
            sxl[me] = sx;
            if (sp && pold >= 0) sxl[pold] = sxn;
            if (bystander) {
                tl[me] = tl[me] - told0;
                if (sp && pold >= 0) tl[pold] = tl[pold] - told1;
            }
        }
    }
    special_sync<WGB>();
    // ---- the micro-ops, one phase each (fnn_core.h: op_thread)
    for (int ph = 0; ph < nops; ph++) {
        const int kind = __builtin_amdgcn_readfirstlane(lst.ops[ph].kind), mcur = __builtin_amdgcn_readfirstlane(lst.ops[ph].mcur);
        const int oa = __builtin_amdgcn_readfirstlane(lst.ops[ph].a), ob = __builtin_amdgcn_readfirstlane(lst.ops[ph].b);
        const bool in = Sl >= 0 && Sl < mcur;
        if (kind == OP_SWAP) {
            const int ia = ixu(oa), ib = ixu(ob);
            if (ia < 0 || ib < 0) bad();
            else if (in) {
                if (me == ia) {
                    double t = sxl[ia]; sxl[ia] = sxl[ib]; sxl[ib] = t;
                    t = tl[ia]; tl[ia] = tl[ib]; tl[ib] = t;
                } else if (me != ib) {
                    const double ta = B(ia, me), tb = B(ib, me);
                    B(ia, me) = tb; B(me, ia) = tb;
                    B(ib, me) = ta; B(me, ib) = ta;
                }
            }
        } else if (kind == OP_MOVE) {
            const int is = ixu(oa), id = ixu(ob);
            if (is < 0 || id < 0) bad();
            else if (in) {
                if (me == is) { B(id, id) = 0.0; sxl[id] = sxl[is]; tl[id] = tl[is]; }
                else if (me != id) { const double t = B(is, me); B(id, me) = t; B(me, id) = t; }
            }
        } else if (kind == OP_AGG3) {
            const int oc = __builtin_amdgcn_readfirstlane(lst.ops[ph].c), od = __builtin_amdgcn_readfirstlane(lst.ops[ph].d),
                      oe = __builtin_amdgcn_readfirstlane(lst.ops[ph].e), flag = __builtin_amdgcn_readfirstlane(lst.ops[ph].flag);
            const int iX = ixu(oa), iY = ixu(ob), iZ = ixu(oc), iU = ixu(od), iV = ixu(oe);
            if (iX < 0 || iY < 0 || iZ < 0 || iU < 0 || iV < 0) bad();
            else if (in) {
                if (me == iX) {
                    // the aliased entry D[u][v] and the diagonal (NetMakerOriginal.java:653-656, 670)
                    const double dxz = B(iX, iZ), dyx = B(iY, iX), dyz = B(iY, iZ);
                    double uv;
                    if (flag) uv = (2.0 / 3.0) * ((2.0 / 3.0) * dxz + div3(dyx)) + div3(dyz);
                    else uv = (2.0 / 3.0) * ((2.0 / 3.0) * dxz + div3(dyz)) + div3(dyx);
                    B(iU, iU) = 0.0; B(iV, iV) = 0.0;
                    B(iU, iV) = uv; B(iV, iU) = uv;
                } else if (me != iY && me != iZ) {
                    const double dx = B(iX, me), dy = B(iY, me), dz = B(iZ, me);
                    const double dy3 = div3(dy);
                    const double nu = (2.0 / 3.0) * dx + dy3;
                    const double nv = (2.0 / 3.0) * dz + dy3;
                    B(iU, me) = nu; B(me, iU) = nu;
                    B(iV, me) = nv; B(me, iV) = nv;
                }
            }
        }
        special_sync<WGB>();
    }
    // ---- add (new layout), NetMakerOriginal.java:520-533
    if (!ev_finish) {
        const int iU = ixu(U), iV = ixu(V);
        if (iU < 0 || iV < 0) bad();
        else if (Sl >= 0 && Sl < m) {
            const bool sp = Sl < twoP;
            const bool rep = !sp || !(Sl & 1);
            double val = 0.0;
            if (rep && Sl != U) {
                const double u0 = B(iU, me), v0 = B(iV, me);
                double dpu;
                if (!sp) {
                    dpu = (u0 + v0) / 2.0;
                    tl[me] = tl[me] + 0.5 * (u0 + v0);
                    out.tu = u0; out.tv = v0;
                } else if (pnew < 0) { bad(); dpu = 0.0; }
                else {
                    const double u1 = B(iU, pnew), v1 = B(iV, pnew);
                    dpu = (((u0 + v0) + u1) + v1) / 4.0;
                    tl[me] = tl[me] + 0.5 * (u0 + v0);
                    tl[pnew] = tl[pnew] + 0.5 * (u1 + v1);
                    out.tu = 0.5 * (u0 + u1); out.tv = 0.5 * (v0 + v1);
                    sxl[pnew] = sxl[pnew] + dpu;
                }
                sxl[me] = sxl[me] + dpu;
                val = dpu;
            } else if (Sl == U) {
                const double uv = B(iU, iV);
                out.tu = 0.5 * uv; out.tv = 0.5 * uv;
            }
            chain_dst[chain_addr(myspos)] = val;
            out.val = val;
        }
    }
    special_sync<WGB>();
    return out;
}

struct UniLane { __device__ __forceinline__ int32_t operator()(int32_t x) const { return __builtin_amdgcn_readfirstlane(x); } };

// ------------------------------------------------------------------ the column thread of the update
// ONE thread per column (node) - the two columns of a paired cluster sit in adjacent lanes and exchange the two
// or four values the cluster distances need by shuffles.  (fnn_core.h: update_bulk is the same computation with
// one thread per cluster; the CPU emulation runs that one against the oracle.  A wave runs alone on its SIMD
// here and issues one instruction every ~4 cycles, so the length of a thread's instruction stream IS the
// duration: splitting the pair halves it.)  All 64 lanes of a wave must call it (shuffles).
// FUSED (k_track with fuse): the column threads run in the SAME launch as the decide step and the chain workgroup; what
// those produce for a column comes in through `fx` instead of plain loads.
struct BulkFix {
    int32_t cU = -2, chain_wait = 0; unsigned evtag = 0;   // Sx[cU], Sx[cU + 1]: delivered by the chain workgroup of this launch
    int32_t pU = -2; double tfin0 = 0.0, tfin1 = 0.0;      // T[pU], T[pU + 1]: summed by this launch's decide step
    int32_t pov_n = 0, pov_slot[2] = {-1, -1}, pov_pos[2] = {0, 0};  // reference positions the plan changed (State.pov_*)
};
template <bool FUSED>
__device__ __forceinline__ void bulk_column(const Dev& d, const PlanView& pv, const int32_t k, double* __restrict__ chain_dst, const BulkFix& fx,
                                            double& dsum, double& dabs, double& tu, double& tv) {
    const bool paired = k < 2 * pv.P_old;
    bool act = k < pv.m_old;
#pragma unroll
    for (int i = 0; i < MAX_S; i++) if (i < pv.nS && pv.S[i] == k) act = false;
    // everything a column reads, in one batch: its entries in the rows of all involved slots, Sx, T, position
    double e[MAX_S];
    const double* colp = d.D + k;
#pragma unroll
    for (int i = 0; i < MAX_S; i++) e[i] = (act && i < pv.nS) ? colp[(int64_t)pv.S[i] * d.ld] : 0.0;
    double sx = act ? d.Sx[k] : 0.0;
    double t_old = act ? d.T[k] : 0.0;
    int32_t pos = act ? d.spos[k] : 0;
    if (FUSED && act) {
        // what this launch itself produced cannot be read with plain loads (other compute units, other L2 slices):
        if (k == fx.pU) t_old = fx.tfin0;          // T of the previous event's cluster: summed by this launch's decide step
        if (k == fx.pU + 1) t_old = fx.tfin1;
#pragma unroll
        for (int q = 0; q < 2; q++) if (q < fx.pov_n && k == fx.pov_slot[q]) pos = fx.pov_pos[q];  // a position the plan changed
        if (fx.chain_wait && (k == fx.cU || k == fx.cU + 1)) {  // its exact row sum: from the chain workgroup of this launch
            const long long t0 = (long long)wall_clock64();
            while (__hip_atomic_load(d.ticket + TRK_FLAG, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != fx.evtag) {
                if ((long long)wall_clock64() - t0 > TRK_WAIT_TICKS) { atomicOr(d.ticket + TRK_ERR, 1u); break; }
                __builtin_amdgcn_s_sleep(2);
            }
            sx = __builtin_bit_cast(double, __hip_atomic_load(reinterpret_cast<const uint64_t*>(d.Sx + k), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        }
    }
    const bool odd = (k & 1) != 0;
    double told = 0.0;
    if (!pv.ev_finish) {
        // subtractClusterDistance(p, x); subtractClusterDistance(p, y) (:455-461, 681-696)
#pragma unroll
        for (int q = 0; q < 2; q++) {
            const int32_t it = q == 0 ? pv.ix : pv.iy, itn = q == 0 ? pv.ixn : pv.iyn;
            const double me0 = it >= 0 ? e[it] : 0.0, mf0 = itn >= 0 ? e[itn] : 0.0;   // this column's entries towards t, t.nbr
            const double oe = __shfl_xor(me0, 1, 64), of = __shfl_xor(mf0, 1, 64);     // ... and the partner column's
            const double e0 = odd ? oe : me0, f0 = odd ? of : mf0, e1 = odd ? me0 : oe, f1 = odd ? mf0 : of;
            double v;
            if (!paired && itn < 0) { v = me0; told += me0; }
            else if (paired && itn < 0) { v = (e0 + e1) / 2.0; told += me0; }
            else if (!paired && itn >= 0) { v = (me0 + mf0) / 2.0; told += 0.5 * (me0 + mf0); }
            else { v = (((e0 + f0) + e1) + f1) / 4.0; told += 0.5 * (me0 + mf0); }
            sx -= v;
        }
    }
    // the rows that change, at this column (all values first: a changed row may be the source of another)
    double nv[MAX_TGT];
    double um = 0.0, vm = 0.0;
#pragma unroll
    for (int t = 0; t < MAX_TGT; t++) {
        nv[t] = 0.0;
        if (t < pv.ntgt) {
            nv[t] = tgt_eval(pv, t, [&](int32_t i) { return e[i]; });
            if (t == pv.tU) um = nv[t];
            if (t == pv.tV) vm = nv[t];
        }
    }
    if (act) {
        const int64_t rk = (int64_t)k * d.ld;
#pragma unroll
        for (int t = 0; t < MAX_TGT; t++) {
            if (t < pv.ntgt) {
                // the entry and its mirror; of the two only the one at or below the diagonal has a bf16 copy.  (Tried, round 4: the two
                // mirror entries of the new cluster's adjacent rows U, U + 1 as ONE 16-byte store and their bf16 copies as one 4-byte
                // store - half the scattered stores, but the extra selects lengthen every thread's instruction stream: 1.54 s instead
                // of 1.41 s at 32768 taxa.)
                const int32_t dst = pv.tdst[t];
                const int64_t rb = (int64_t)dst * d.ld;
                d.D[rb + k] = nv[t];
                d.D[rk + dst] = nv[t];
                if (d.H) d.H[k < dst ? (int64_t)dst * d.ldh + k : (int64_t)k * d.ldh + dst] = bf16_from_double(nv[t]);
            }
        }
    }
    if (!pv.ev_finish) {
        // updateClusterDistances, per-node part (:520-531)
        const double uo = __shfl_xor(um, 1, 64), vo = __shfl_xor(vm, 1, 64);
        const double u0 = odd ? uo : um, v0 = odd ? vo : vm, u1 = odd ? um : uo, v1 = odd ? vm : vo;
        const double dpu = paired ? (((u0 + v0) + u1) + v1) / 4.0 : (um + vm) / 2.0;
        if (act) {
            const bool rep = !paired || !odd;
            d.Sx[k] = sx + dpu;
            chain_dst[chain_addr(pos)] = rep ? dpu : 0.0;
            d.T[k] = (t_old - told) + 0.5 * (um + vm);
            dsum = rep ? dpu : 0.0;
            dabs = dsum < 0.0 ? -dsum : dsum;
            tu = paired ? 0.5 * um : um;
            tv = paired ? 0.5 * vm : vm;
        }
    }
}


// ---- the plan message of the fused event kernel (fnn_core.h: PlanMsg) ----
// post: wave 0 of the deciding workgroup, from its LDS copy; every word carries the launch's tag, so no flag and no drain
__device__ __forceinline__ void plan_post(const Dev& d, const PlanMsg& m, unsigned tag) {
    const int lane = threadIdx.x & 63;
    const uint32_t* w = reinterpret_cast<const uint32_t*>(&m);
    __hip_atomic_store(d.plan + lane, ((uint64_t)tag << 32) | w[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(d.plan + 64 + lane, ((uint64_t)tag << 32) | w[64 + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// "no update in this launch" (kind 0): the waiting workgroups leave.  One wave.
__device__ __forceinline__ void plan_post_none(const Dev& d, unsigned tag) {
    const int lane = threadIdx.x & 63;
    __hip_atomic_store(d.plan + lane, (uint64_t)tag << 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(d.plan + 64 + lane, (uint64_t)tag << 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// wait (wave 0 polls, the whole workgroup joins at the barrier); a deadline instead of a hang
__device__ __forceinline__ void plan_wait(const Dev& d, PlanMsg& m, unsigned tag) {
    if (threadIdx.x < 64) {
        const int lane = threadIdx.x;
        uint32_t* w = reinterpret_cast<uint32_t*>(&m);
        const long long t0 = (long long)wall_clock64();
        for (;;) {
            const uint64_t a = __hip_atomic_load(d.plan + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const uint64_t b = __hip_atomic_load(d.plan + 64 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const bool ok = (uint32_t)(a >> 32) == tag && (uint32_t)(b >> 32) == tag;
            if (__ballot(ok) == ~0ULL) { w[lane] = (uint32_t)a; w[64 + lane] = (uint32_t)b; break; }
            if ((long long)wall_clock64() - t0 > TRK_WAIT_TICKS) {
                w[lane] = 0u; w[64 + lane] = 0u;  // (kind = 0)
                if (lane == 0) atomicOr(d.ticket + TRK_ERR, 2u);
                break;
            }
            __builtin_amdgcn_s_sleep(6);
        }
    }
    __syncthreads();
}

// fuse != 0: a WINDOW event in ONE launch.  The launch sequence has no scan kernels (has_scan == 0) and no k_update: when
// the window serves the event, the deciding workgroup (the tracking workgroup that arrives last) posts the plan as a message
// (plan_post), runs the involved slots' phases itself and closes the event; tracking workgroup b < ceil(m / 1024) has waited for
// the message and now updates the columns [1024 b, 1024 b + 1024) (bulk_column<true>); the SPARE workgroup (the launch's last)
// takes the block of the deciding workgroup.  Saves a kernel boundary per event (~4.5 us: dispatch, the release at the end of
// one kernel and the acquire at the start of the next) at the price of one message round trip (~1 us), and the chain
// workgroup's exact row sum (14-23 us) now runs beside tracking AND update.  What the launch itself produces for a column
// thread - T of the previous event's cluster (decide step), its exact row sum (chain workgroup), reference positions the plan
// moved - travels in the message or is read write-through (BulkFix); the row-sum addends alternate between two chain buffers
// (State.chain_buf) because the chain workgroup of this launch is still reading the other one.
// ---- a COLUMN workgroup of the fused event kernel: block `block` of BULK_COLS columns by its first BULK_COLS threads (one wave per
// SIMD, as in k_update: the column thread is a chain of latencies and scattered stores - a 1024-column block on one compute unit
// took 7 us, 256 columns take 4); the other waves only keep the barriers company.
constexpr int BULK_COLS = 256;
__device__ __forceinline__ void column_workgroup(const Dev& d, PlanMsg& M, int block, double (*shp)[4], unsigned tag, int prof2) {
    long long tk0 = prof2 ? (long long)wall_clock64() : 0;
    // the plan cannot be there before the tracking, the fan-in and the decide step of this launch have run (>= 12 us): sleep through
    // most of that instead of polling (the polls of all column workgroups go to the same few cache lines)
    if (threadIdx.x < 64) {
        const long long t0 = (long long)wall_clock64();
        while ((long long)wall_clock64() - t0 < 1400) __builtin_amdgcn_s_sleep(32);  // 14 us of the 100 MHz clock
    }
    plan_wait(d, M, tag);
    if (prof2) { const long long now_ = (long long)wall_clock64(); d.ticks[12] += now_ - tk0; tk0 = now_; }
    if (M.kind != 1 || block >= M.nbulk) return;
    double dsum = 0.0, dabs = 0.0, tu = 0.0, tv = 0.0;
    if (threadIdx.x < BULK_COLS) {
        const PlanView pv = plan_view(M, UniLane{});
        BulkFix fx;
        fx.cU = __builtin_amdgcn_readfirstlane(M.cU); fx.chain_wait = __builtin_amdgcn_readfirstlane(M.chain_wait);
        fx.evtag = (unsigned)__builtin_amdgcn_readfirstlane(M.evtag);
        fx.pU = __builtin_amdgcn_readfirstlane(M.pU);
        fx.tfin0 = __builtin_bit_cast(double, ((uint64_t)(uint32_t)M.tfin[1] << 32) | (uint32_t)M.tfin[0]);
        fx.tfin1 = __builtin_bit_cast(double, ((uint64_t)(uint32_t)M.tfin[3] << 32) | (uint32_t)M.tfin[2]);
        fx.pov_n = __builtin_amdgcn_readfirstlane(M.pov_n);
        fx.pov_slot[0] = __builtin_amdgcn_readfirstlane(M.pov_slot[0]); fx.pov_slot[1] = __builtin_amdgcn_readfirstlane(M.pov_slot[1]);
        fx.pov_pos[0] = __builtin_amdgcn_readfirstlane(M.pov_pos[0]); fx.pov_pos[1] = __builtin_amdgcn_readfirstlane(M.pov_pos[1]);
        const int chain_dst = __builtin_amdgcn_readfirstlane(M.chain_dst);
        if (prof2) { const long long now_ = (long long)wall_clock64(); d.ticks[13] += now_ - tk0; tk0 = now_; }
        bulk_column<true>(d, pv, (int32_t)(block * BULK_COLS + (int)threadIdx.x), d.chain + (size_t)chain_dst * d.cstride, fx, dsum, dabs, tu, tv);
        if (prof2) { const long long now_ = (long long)wall_clock64(); d.ticks[14] += now_ - tk0; tk0 = now_; }
        // per-workgroup partial sums (tree order), as k_update's
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            dsum += __shfl_down(dsum, off, 64); dabs += __shfl_down(dabs, off, 64);
            tu += __shfl_down(tu, off, 64); tv += __shfl_down(tv, off, 64);
        }
        if ((threadIdx.x & 63) == 0) { double* r = shp[threadIdx.x >> 6]; r[0] = dsum; r[1] = dabs; r[2] = tu; r[3] = tv; }
    }
    __syncthreads();
    if (threadIdx.x < 4) {
        const int k = threadIdx.x;
        d.upart[4 * (size_t)block + k] = ((shp[0][k] + shp[1][k]) + shp[2][k]) + shp[3][k];
    }
    if (prof2) d.ticks[15] += (long long)wall_clock64() - tk0;
}

// FUSE: a WINDOW event in ONE launch.  The launch sequence has no scan kernels (has_scan == 0) and no k_update: the grid carries
// `ncol` COLUMN workgroups behind the tracking (and helper) workgroups.  When the window serves the event, the deciding workgroup
// (the tracking workgroup that arrives last) posts the plan as a message (plan_post), runs the involved slots' phases itself - wave 0
// alone, no workgroup barrier - and closes the event; column workgroup b has waited for the message and updates the columns
// [256 b, 256 b + 256) (bulk_column<true>).  Saves a kernel boundary per event (~4.5 us: dispatch, the release at the end of one
// kernel and the acquire at the start of the next) at the price of one message round trip (~1 us), and the chain workgroup's exact
// row sum (14-23 us) runs beside tracking AND update.  What the launch itself produces for a column thread - T of the previous
// event's cluster (decide step), its exact row sum (chain workgroup), reference positions the plan moved - travels in the message
// or is read write-through (BulkFix); the row-sum addends alternate between two chain buffers (State.chain_buf) because the chain
// workgroup of this launch is still reading the other one.  No address is written twice in one launch by different workgroups
// with plain stores (their L2 slices would write back in an undefined order): T of the previous cluster goes out write-through.
template <bool HELP, bool FUSE>
__global__ __launch_bounds__(TRK_THREADS) void k_track(Dev d, int force_base, int timed, int has_scan, int tgroup, int ticks, unsigned jobtag, int ncol) {
    __shared__ ChainLds<CH_EPT> L;
    __shared__ DecideLds S;
    __shared__ Cand sh[TRK_THREADS / 64], shu[TRK_THREADS / 64];
    __shared__ int lastflag;
    __shared__ double shs[2];
    __shared__ unsigned hword;
    __shared__ double shp[BULK_COLS / 64][4];
    State* st = d.st;
    if (blockIdx.x == 0) {  // the chain workgroup
        chain_workgroup(d, L);
        return;
    }
    const int wg = (int)blockIdx.x - 1, G = (int)gridDim.x - 1 - (HELP ? TRK_NHELP : 0) - (FUSE ? ncol : 0);
    if (FUSE && wg >= G + (HELP ? TRK_NHELP : 0)) {  // a column workgroup
        const int block = wg - G - (HELP ? TRK_NHELP : 0);
        column_workgroup(d, S.msg, block, shp, jobtag, (ticks != 0 && threadIdx.x == 0 && block == 0) ? 1 : 0);
        return;
    }
    if (HELP && wg >= G) {  // a helper workgroup of the exact ComputeRx sums (see decide_step)
        rx_helper_workgroup(d, L, wg - G, jobtag, &hword);
        return;
    }
    // phase split of the last-arriving workgroup (diagnostic, FNN_TICKS=1): thread 0 stamps the 100 MHz clock
    const bool prof = ticks != 0 && threadIdx.x == 0;
#define TRK_TICK(slot) do { if (prof) S.tk[(slot) + 1] = (long long)wall_clock64(); } while (0)
    if (threadIdx.x == 0) S.tkon = ticks;
    if (prof) S.tk[0] = (long long)wall_clock64();
    // (this thread's first tracked pair is fetched beside the control block: its address depends on nothing)
    const int64_t start = (int64_t)wg * TRK_THREADS + threadIdx.x;
    PairRec rec0;
    rec0.wa = rec0.wb = 0; rec0.sa = rec0.sb = -1;
    rec0.e[0] = rec0.e[1] = rec0.e[2] = rec0.e[3] = 0.0;
    if (start < LA_PCAP) rec0 = track_pair_load(d, start);
    if (st->done) {
        if (wg == 0 && threadIdx.x == 0) {
            st->ev_active = 0;  // (a launch sequence without a decide kernel must not replay the last event)
            if (HELP) job_post(d, jobtag, 0u);
        }
        if (FUSE && wg == 0 && threadIdx.x < 64) plan_post_none(d, jobtag);
        return;
    }
    // every tracking workgroup fetches the control block now (nothing writes to it while they track): the one that
    // arrives last continues on this LDS copy and writes it back at the end
    state_in(S.lst, st);
    if (!has_scan && st->stall) {  // (the launch sequence has no scan kernels and the window is gone: nothing to do)
        if (wg == 0 && threadIdx.x == 0) {
            st->n_stalled++;
            if (HELP) job_post(d, jobtag, 0u);
        }
        if (FUSE && wg == 0 && threadIdx.x < 64) plan_post_none(d, jobtag);
        return;
    }
    if (force_base || !la_active(*st)) {
        if (wg == 0 && threadIdx.x == 0) {
            if (HELP) job_post(d, jobtag, 0u);
            st->ev_timed = timed;
            if (st->la_valid) st->la_prev_end = 0;  // the window ends on schedule
            la_prepare_base(*st, d.lacnt);
            st->stall = has_scan ? 0 : 1;
            if (!has_scan) st->n_stalled++;
        }
        if (FUSE && wg == 0 && threadIdx.x < 64) plan_post_none(d, jobtag);
        return;
    }
    // (fused event kernel) the pending row sum as the control block describes it at the START of the launch
    const int32_t chain_pending0 = st->chain_pending, chain_buf0 = st->chain_buf, chain_U0 = st->chain_U, m0 = st->m;
    const unsigned evtag0 = (unsigned)st->n_events;
    if (FUSE && threadIdx.x == 0) {  // (parked in LDS for the tail: no scalar register stays live across the decide step for them)
        S.fz[0] = chain_pending0; S.fz[1] = chain_buf0; S.fz[2] = chain_U0; S.fz[3] = (int32_t)evtag0;
        S.fz[4] = (m0 + BULK_COLS - 1) / BULK_COLS; S.fz[5] = 0;
    }
    TrackArgs ta = track_args(*st);
    const int64_t items = track_item_count(ta);
    // The swept cluster's exact row sum is being computed by workgroup 0: the sweep runs on the tree-ordered
    // sum of k_update's partials, which are fetched now and summed after the tracked pairs (their loads
    // overlap).  (More than one unswept cluster is not expected: the window ends.)
    double eps_u = 0.0;
    bool giveup = false;  // the window cannot serve this event: it ends here (as after a failed certification)
    const bool pending = chain_pending0 != 0 && ta.nf > ta.nf0;
    if (pending && ta.nf - ta.nf0 != 1) giveup = true;
    const bool approx = pending && !giveup;
    double2 up0 = make_double2(0.0, 0.0), up1 = up0, up2 = up0;
    const int npart = approx ? st->upart_n : 0;
    if (threadIdx.x < 64) {
        const double2* up = reinterpret_cast<const double2*>(d.upart);  // [workgroup] {sum, sum of magnitudes | T partials}
        if ((int)threadIdx.x < npart) up0 = up[2 * threadIdx.x];
        if ((int)threadIdx.x + 64 < npart) up1 = up[2 * (threadIdx.x + 64)];
        if ((int)threadIdx.x + 128 < npart) up2 = up[2 * (threadIdx.x + 128)];
    }
    Cand best, bestu;
    best = cand_none();
    bestu = best;
    const int64_t stride = (int64_t)G * TRK_THREADS;
    TRK_TICK(0);
    if (start < ta.np) track_pair_rec(d, rec0, ta, best);
    for (int64_t it = start + stride; it < ta.np; it += stride) track_pair_item(d, it, ta, best);
    TRK_TICK(1);
    if (approx) {
        if (threadIdx.x < 64) {
            double su = (up0.x + up1.x) + up2.x, sa = (up0.y + up1.y) + up2.y;
            for (int b = threadIdx.x + 192; b < npart; b += 64) { su += d.upart[4 * b]; sa += d.upart[4 * b + 1]; }
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) { su += __shfl_down(su, off, 64); sa += __shfl_down(sa, off, 64); }
            if (threadIdx.x == 0) { shs[0] = su; shs[1] = sa; }
        }
        __syncthreads();
        ta.approx = 1;
        ta.usl = chain_U0;
        ta.sxu = shs[0];
        // both the sequential and the tree order are within gamma_m sum|terms| of the exact sum
        eps_u = 4.0 * ((double)ta.m + 8.0) * 1.1102230246251565e-16 * shs[1];
    }
    bool swept_ok = true;
    // (the sweep items are dealt out from the far end of the grid: tracked pairs and sweep run side by side in
    //  different workgroups as long as there are fewer items than threads)
    if (!giveup)
        for (int64_t r = stride - 1 - start; r < items - ta.np; r += stride) swept_ok = track_sweep_item(d, r, ta, best, bestu) && swept_ok;
    if (!swept_ok) atomicOr(d.ticket + TRK_BAD, 1u);  // (not expected: the swept cluster is not the chain's)
    TRK_TICK(2);
    // both minima in one pass: wave reduction, one barrier, thread 0 folds the waves
    {
        best = wave_reduce(best);
        if (ta.approx) bestu = wave_reduce(bestu);
        const int lane_ = threadIdx.x & 63, w_ = threadIdx.x >> 6;
        if (lane_ == 0) { sh[w_] = best; shu[w_] = bestu; }
        __syncthreads();
        if (w_ == 0) {  // the waves' minima: one more wave reduction (no serial walk through LDS)
            Cand b2, bu2;
            b2 = cand_none(); bu2 = b2;
            if (lane_ < TRK_THREADS / 64) { b2 = sh[lane_]; bu2 = shu[lane_]; }
            best = wave_reduce(b2);
            if (ta.approx) bestu = wave_reduce(bu2);
            if (lane_ == 0) {
                rec_publish(&d.recs[wg], best, d.rchk + wg, evtag0);
                if (ta.approx) rec_publish(&d.recs[TRK_REC_U + wg], bestu, d.rchk + TRK_REC_U + wg, evtag0);
            }
        }
    }
    TRK_TICK(3);
    if (threadIdx.x == 0) {
        // the record went out write-through (rec_publish) from this very thread: once its stores are acknowledged
        // the arrival may be counted - no release fence (a cache write-back) on the event chain
        __builtin_amdgcn_s_waitcnt(0);
        // last of its group of TRK_GROUP workgroups -> last of the groups (counters 128 bytes apart)
        const unsigned g = (unsigned)wg / (unsigned)tgroup, ngroups = ((unsigned)G + tgroup - 1) / (unsigned)tgroup;
        const unsigned gsize = g + 1 < ngroups ? (unsigned)tgroup : (unsigned)G - g * tgroup;
        int last = 0;
        if (atomicAdd(d.ticket + 32 * (g + 1), 1u) == gsize - 1) {
            d.ticket[32 * (g + 1)] = 0u;  // (read again by the next launch only)
            last = atomicAdd(d.ticket, 1u) == ngroups - 1 ? 1 : 0;
        }
        lastflag = last;
    }
    __syncthreads();
    if (!lastflag) return;
    // (no acquire fence: the records are read past the caches, rec_fetch, after the barrier above)
    TRK_TICK(4);
    // from here on this workgroup works on its LDS copy of the control block (no other workgroup writes to it
    // during this launch); the copy goes back at the end
    State& lst = S.lst;
    Dev dl = d;
    dl.st = &lst;
    // the records of all workgroups, both sets at once, by the first wave
    if (threadIdx.x < 64) {
        Cand b, bu;
        b = cand_none();
        bu = b;
        const unsigned badword = __hip_atomic_load(d.ticket + TRK_BAD, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned errword = FUSE ? __hip_atomic_load(d.ticket + TRK_ERR, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
        const uint32_t tag = evtag0;
        bool stale = false, reread = false;
        for (int pass = 0; pass < 2; pass++) {
            b = cand_none();
            bu = b;
            stale = false;
            for (int i = threadIdx.x; i < G; i += 64) {
                Cand c = rec_fetch(&d.recs[i]);
                if (rec_check_word(c, tag) != __hip_atomic_load(d.rchk + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) stale = true;
                if (cand_better(c, b)) b = c;
                if (ta.approx) {
                    c = rec_fetch(&d.recs[TRK_REC_U + i]);
                    if (rec_check_word(c, tag) != __hip_atomic_load(d.rchk + TRK_REC_U + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) stale = true;
                    if (cand_better(c, bu)) bu = c;
                }
            }
            stale = __ballot(stale) != 0ULL;
            if (!stale) break;
            reread = true;
            __threadfence();  // (not expected: see rec_publish)
        }
        b = wave_reduce(b);
        if (ta.approx) bu = wave_reduce(bu);
        if (threadIdx.x == 0) {
            sh[0] = b;
            shu[0] = bu;
            if (badword != 0u) {
                d.ticket[TRK_BAD] = 0u;
                lastflag = 2;
            }
            if (errword != 0u) S.lst.error = 16;  // a wait of an earlier fused launch ran into its deadline
            if (stale) lastflag = 2;  // (a record that could not be read back intact: the window gives this event up)
            if (reread) S.lst.n_ev_persistent++;
        }
    }
    __syncthreads();
    best = sh[0];
    bestu = shu[0];
    if (lastflag == 2) {
        giveup = true;
        if (d.strict && threadIdx.x == 0) S.lst.error = 13;  // several ranks: a give-up the other ranks cannot see would desynchronise them
    }
    __syncthreads();
    TRK_TICK(5);
    if (ta.approx && !giveup) {
        // |Q~ - Q| <= eps_u + the roundings of (c-2) D - Sp - Sq, each <= 2^-53 of a term <= (c + 2n) Dmax
        const double dmax = __builtin_bit_cast(double, st->dmax_bits);
        const double margin = 2.0 * eps_u + 192.0 * 1.1102230246251565e-16 * ((double)st->n + 4.0) * dmax + 1e-300;
        const bool certain = (bestu.q - margin > best.q) || (bestu.q == inf_f64());
        if (!certain) {
            // a swept pair may be the minimum: wait for the chain workgroup, sweep again with the exact sum
            if (threadIdx.x == 0) {
                const unsigned want = evtag0;
                const long long t_wait = (long long)wall_clock64();
                while (__hip_atomic_load(d.ticket + TRK_FLAG, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != want) {
                    if ((long long)wall_clock64() - t_wait > TRK_WAIT_TICKS) { lst.error = 10; break; }
                    __builtin_amdgcn_s_sleep(2);
                }
                __threadfence();
                lst.n_sweep_waits++;
            }
            __syncthreads();
            Cand bx;
            bx = cand_none();
            for (int64_t r = threadIdx.x; r < items - ta.np; r += TRK_THREADS) sweep_exact_item(d, r, ta, bx);
            bx = block_reduce<TRK_THREADS / 64>(bx, sh);
            if (threadIdx.x == 0) sh[0] = bx;
            __syncthreads();
            bx = sh[0];
            if (cand_better(bx, best)) best = bx;
        }
        // certain: every swept pair's exact Q lies strictly above best.q, so best wins as it stands
    }
    if (giveup) { best = cand_none(); }
    if (threadIdx.x == 0) {
        *d.ticket = 0u;
        lst.ev_timed = timed;
        if (giveup) lst.n_sweep_waits++;
        la_track_done(dl, best, ta);
        lst.stall = (!lst.la_hit && !has_scan) ? 1 : 0;
        if (lst.stall) lst.n_stalled++;
    }
    __syncthreads();
    TRK_TICK(6);
    // the window has certified the minimum: Cx / Cy, the 4-candidate choice and the merge plan follow at once
    // (the launch sequence of a window event has no decide kernel); otherwise the event scans (or stalls)
    if (lst.la_hit) decide_step<HELP>(d, S, L, best, jobtag);
    else if (HELP && threadIdx.x == 0) job_post(d, jobtag, 0u);  // (the event scans or stalls: nothing for the helpers)
    TRK_TICK(7);
    if (prof) {
        for (int q = 0; q < 8; q++) d.ticks[q] += S.tk[q + 1] - S.tk[q];
    }
    __syncthreads();
    if (FUSE) {
        // The rest of the event by WAVE 0 alone (the other waves wait at the barrier in front of the write-back): no workgroup
        // barrier on this stretch - one wave's LDS accesses execute in program order (special_sync<false>).
        if (threadIdx.x < 64) {
            const int lane = (int)threadIdx.x;
            const int32_t chain_pending0 = S.fz[0], chain_buf0 = S.fz[1], chain_U0 = S.fz[2], nbulk = S.fz[4];
            const unsigned evtag0 = (unsigned)S.fz[3];
            long long tk0 = prof ? (long long)wall_clock64() : 0;
#define FUS_TICK(slot) do { if (prof) { const long long now_ = (long long)wall_clock64(); d.ticks[8 + (slot)] += now_ - tk0; tk0 = now_; } } while (0)
            const bool upd = lst.la_hit && lst.ev_active;
            if (!upd) plan_post_none(d, jobtag);
            else {
                // ---- the plan message: the tail of the control block (nS .. tgt) word for word, the scalars by three lanes
                PlanMsg& M = S.msg;
                const int chain_wait = chain_pending0 ? 1 : 0;
                const int chain_dst = (chain_pending0 && chain_buf0 == 0) ? CHAIN_ALT : 0;
                static_assert(offsetof(PlanMsg, fill) - offsetof(PlanMsg, nS) == 60 * 4 && sizeof(State) - offsetof(State, nS) == 60 * 4, "plan tail layout");
                if (lane < 60) (&M.nS)[lane] = (&lst.nS)[lane];
                if (lane < 44) M.fill[lane] = 0;
                if (lane == 61) {
                    M.kind = 1; M.last_wg = wg; M.pU = S.msg_pU; M.cU = chain_pending0 ? chain_U0 : -2; M.chain_wait = chain_wait;
                    M.evtag = (int32_t)evtag0; M.chain_dst = chain_dst; M.nbulk = nbulk;
                }
                if (lane == 62) {
                    M.pov_n = lst.pov_n; M.pov_slot[0] = lst.pov_slot[0]; M.pov_slot[1] = lst.pov_slot[1];
                    M.pov_pos[0] = lst.pov_pos[0]; M.pov_pos[1] = lst.pov_pos[1]; M.pad0 = 0;
                    M.m_old = lst.m_old; M.P_old = lst.P_old; M.ev_finish = lst.ev_finish; M.xs = lst.xs; M.ys = lst.ys; M.pad1 = 0;
                }
                if (lane == 63) {
                    const uint64_t t0b = __builtin_bit_cast(uint64_t, S.tfin[0]), t1b = __builtin_bit_cast(uint64_t, S.tfin[1]);
                    M.tfin[0] = (int32_t)(uint32_t)t0b; M.tfin[1] = (int32_t)(uint32_t)(t0b >> 32);
                    M.tfin[2] = (int32_t)(uint32_t)t1b; M.tfin[3] = (int32_t)(uint32_t)(t1b >> 32);
                }
                special_sync<false>();
                plan_post(d, M, jobtag);
                FUS_TICK(0);
                // ---- the involved slots' phases on this LDS copy (k_update's special workgroup, here without fetching the control block)
                double* blk = S.blk; double* sxl = S.sxl; double* tl = S.tl;
                if (lane == 0) S.berr = 0;
                {
                    bool mine = lane < lst.nS && chain_wait && (lst.S[lane < MAX_S ? lane : 0] == chain_U0 || lst.S[lane < MAX_S ? lane : 0] == chain_U0 + 1);
                    if (__ballot(mine) != 0ULL) {  // an involved slot's exact row sum is still on its way from the chain workgroup of this launch
                        const long long t_wait = (long long)wall_clock64();
                        while (__hip_atomic_load(d.ticket + TRK_FLAG, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != evtag0) {
                            if ((long long)wall_clock64() - t_wait > TRK_WAIT_TICKS) { if (lane == 0) lst.error = 10; break; }
                            __builtin_amdgcn_s_sleep(2);
                        }
                    }
                }
                {
                    const int32_t e = (int32_t)lane, i = e / MAX_S, j = e % MAX_S;
                    special_block_load(dl, blk, sxl, tl, e);
                    if (i < lst.nS && j == 0 && chain_wait && (lst.S[i] == chain_U0 || lst.S[i] == chain_U0 + 1))
                        sxl[i] = __builtin_bit_cast(double, __hip_atomic_load(reinterpret_cast<const uint64_t*>(d.Sx + lst.S[i]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                    if (i < lst.nS && j == 1 && S.msg_pU >= 0 && (lst.S[i] == S.msg_pU || lst.S[i] == S.msg_pU + 1)) tl[i] = S.tfin[lst.S[i] - S.msg_pU];
                }
                special_sync<false>();
                FUS_TICK(1);
                const SpecialOut o = special_wave<false>(d, lst, blk, sxl, tl, &S.berr, d.chain + (size_t)chain_dst * d.cstride);
                double dsum = o.val, dabs = o.val < 0.0 ? -o.val : o.val, tu = o.tu, tv = o.tv;
#pragma unroll
                for (int off = 32; off >= 1; off >>= 1) {
                    dsum += __shfl_down(dsum, off, 64); dabs += __shfl_down(dabs, off, 64);
                    tu += __shfl_down(tu, off, 64); tv += __shfl_down(tv, off, 64);
                }
                if (lane == 0) {  // this workgroup's partial sums: record nbulk (behind the column blocks')
                    double* r = d.upart + 4 * (size_t)nbulk;
                    r[0] = dsum; r[1] = dabs; r[2] = tu; r[3] = tv;
                }
                FUS_TICK(2);
                special_block_store(dl, blk, sxl, tl, (int32_t)lane);
                // the close of the event (nothing in the update reads what it writes)
                if (lane == 0) {
                    lst.tp_n = lst.ev_finish ? 0 : nbulk + 1;  // partial sums of T of the new cluster's nodes: the next decide step adds them up
                    lst.tp_U = lst.U;
                    if (S.berr) lst.error = S.berr;
                    lst.upart_n = nbulk + 1;
                    lst.chain_m = lst.m;
                    lst.chain_U = lst.U;
                    lst.chain_buf = chain_dst;
                    lst.chain_pending = lst.ev_finish ? 0 : 1;
                    close_event(dl);
                    if (lst.ev_finish) {  // special finish: u, v keep the default Sx (NetNode.java:15); nothing to sum
                        d.Sx[lst.U] = 0.0;
                        d.Sx[lst.U + 1] = 0.0;
                    }
                }
                FUS_TICK(3);
            }
#undef FUS_TICK
        }
        __syncthreads();
    }
    state_out(st, lst);
#undef TRK_TICK
}


// ------------------------------------------------------------------ k_update
// subtract + every micro-op + add of one event in ONE launch (fnn_core.h: update_bulk /
// update_special).  The last workgroup handles the <= 8 involved slots in phases.
// The involved slots' phases (fnn_core.h: subtract_thread / op_thread / add_thread on the S x S block) by ONE wave:
// lane i < nS owns involved slot S[i] and column i of the block in LDS.  The generic bodies address the block through
// slot numbers (a search per access: ~1000 instructions per phase for a lone wave); here every operand of a phase is
// resolved once - slots that are the same for all lanes by a ballot, a lane's own partner by eight readlanes up front -
// so a phase is a few dozen instructions.  Same operations, same order, same roundings as the generic bodies (which
// the CPU emulation runs against the oracle); the GPU parity tests compare Sx, T's consumers and the live matrix
// after every event.  Returns this lane's addends {row-sum addend, T terms of u, v} of the add phase.
__global__ __launch_bounds__(256) void k_update(Dev d, int defer, int ticks) {
    __shared__ double shp[4][4];
    __shared__ State lst;  // every workgroup fetches the control block ONCE, with one coalesced load
    State* st = d.st;
    // diagnostic (FNN_TICKS=1): 100 MHz stamps of thread 0 of the special workgroup (slots 0-3) and of bulk workgroup 0 (4-7)
    long long tk0 = 0;
    const bool prof = ticks != 0 && threadIdx.x == 0 && (blockIdx.x == gridDim.x - 1 || blockIdx.x == 0);
    const int tkb = blockIdx.x == gridDim.x - 1 ? 0 : 4;
#define UPD_TICK(slot) do { if (prof) { const long long now_ = (long long)wall_clock64(); d.ticks[8 + tkb + (slot)] += now_ - tk0; tk0 = now_; } } while (0)
    if (prof) tk0 = (long long)wall_clock64();
    // (... and, per workgroup, the sums of its start and end stamps over the events: which workgroup ends last, and how much later)
    const long long wg_t0 = (ticks != 0 && threadIdx.x == 0) ? (long long)wall_clock64() : 0;
    state_in(lst, st);
    __syncthreads();
    if (!lst.ev_active || lst.stall) return;
    UPD_TICK(0);
    double dsum = 0.0, dabs = 0.0, tu = 0.0, tv = 0.0;
    const bool special = blockIdx.x == gridDim.x - 1;
    if (special) {
        // the <= 8 involved slots: their S x S block of the matrix, their row sums and their T are copied to LDS,
        // the reference's per-node bodies run on the copy in phases (subtract, one per micro-op, add: up to 8
        // dependent steps at LDS latency), and the block goes back in one sweep.  The control block is worked on
        // in LDS too (the phases read the plan from it; the last wave closes the event on it) and goes back whole:
        // the bulk workgroups only read fields that this workgroup does not change.
        __shared__ double blk[MAX_S * MAX_S], sxl[MAX_S], tl[MAX_S];
        __shared__ int32_t berr;
        Dev dl = d;
        dl.st = &lst;
        if (threadIdx.x == 0) berr = 0;
        if (threadIdx.x < MAX_S * MAX_S) special_block_load(dl, blk, sxl, tl, (int32_t)threadIdx.x);
        __syncthreads();
        UPD_TICK(1);
        if (threadIdx.x < 64) {  // (the other waves only keep the barriers company)
            const SpecialOut o = special_wave<true>(d, lst, blk, sxl, tl, &berr, d.chain);
            dsum = o.val;
            dabs = o.val < 0.0 ? -o.val : o.val;
            tu = o.tu;
            tv = o.tv;
        } else {
            const int nb = 2 + lst.nops;
            for (int q = 0; q < nb; q++) __syncthreads();
        }
        UPD_TICK(2);
        if (threadIdx.x < MAX_S * MAX_S) special_block_store(dl, blk, sxl, tl, (int32_t)threadIdx.x);
        // the close of the event (nothing in the update reads what it writes)
        if (threadIdx.x == 255) {
            lst.tp_n = lst.ev_finish ? 0 : (int)gridDim.x;  // partial sums of T of the new cluster's nodes: the next decide step adds them up
            lst.tp_U = lst.U;
            if (berr) lst.error = berr;
            if (defer) {
                lst.upart_n = (int)gridDim.x;
                lst.chain_m = lst.m;
                lst.chain_U = lst.U;
                lst.chain_buf = 0;
                lst.chain_pending = lst.ev_finish ? 0 : 1;
                close_event(dl);
                if (lst.ev_finish) {  // special finish: u, v keep the default Sx (NetNode.java:15); nothing to sum
                    d.Sx[lst.U] = 0.0;
                    d.Sx[lst.U + 1] = 0.0;
                }
            }
        }
        __syncthreads();
        state_out(st, lst);
    } else {
        // ONE thread per column (node): bulk_column, shared with the fused event kernel (k_track with fuse)
        const PlanView pv = plan_view(lst, UniLane{});
        UPD_TICK(1);
        const BulkFix nofix{};
        bulk_column<false>(d, pv, (int32_t)(blockIdx.x * 256 + threadIdx.x), d.chain, nofix, dsum, dabs, tu, tv);
        UPD_TICK(2);
    }
    // per-workgroup partial sums (tree order): of the new cluster's row-sum addends and their magnitudes (for
    // the next event's sweep while the exact sequential sum is on its way) and of T of its two nodes
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        dsum += __shfl_down(dsum, off, 64);
        dabs += __shfl_down(dabs, off, 64);
        tu += __shfl_down(tu, off, 64);
        tv += __shfl_down(tv, off, 64);
    }
    if ((threadIdx.x & 63) == 0) { double* r = shp[threadIdx.x >> 6]; r[0] = dsum; r[1] = dabs; r[2] = tu; r[3] = tv; }
    __syncthreads();
    if (threadIdx.x < 4) {
        const int k = threadIdx.x;
        d.upart[4 * blockIdx.x + k] = ((shp[0][k] + shp[1][k]) + shp[2][k]) + shp[3][k];
    }
    UPD_TICK(3);
    if (ticks != 0 && threadIdx.x == 0) {
        // (slot TICK_WG - 1 is the workgroup of the involved slots, whatever the grid; bulk workgroups beyond TICK_WG - 2 share a slot)
        const int slot = blockIdx.x == gridDim.x - 1 ? TICK_WG - 1 : ((int)blockIdx.x < TICK_WG - 2 ? (int)blockIdx.x : TICK_WG - 2);
        d.ticks[32 + slot] += wg_t0;
        d.ticks[32 + TICK_WG + slot] += (long long)wall_clock64();
        d.ticks[32 + 2 * TICK_WG + slot] += 1;
    }
#undef UPD_TICK
}

// the pending exact row sum, on its own (before the host looks at the state)
__global__ __launch_bounds__(CH_T) void k_chain_flush(Dev d) {
    __shared__ ChainLds<CH_EPT> L;
    chain_workgroup(d, L);
    __syncthreads();
    if (threadIdx.x == 0) d.st->chain_pending = 0;
}

// ------------------------------------------------------------------ k_finalize
// The sequential sum as the reference writes it, for matrices with negative entries: the parallel forms (fnn_chain.h) rest on
// non-negative addends (a running sum that only grows) and fall back to ordinary additions piece by piece otherwise - 260 us
// per sum at 8192 taxa.  Here the addends are staged through LDS in position order, 8192 at a time, and ONE lane adds them
// (the loads do not depend on the sum, the additions do: ~8 cycles each).  Valid in thread 0.
constexpr int SER_CHUNK = 8192;
__device__ __forceinline__ double serial_chain_sum(const double* buf, int m, double* ser) {
    double s = 0.0;
    for (int base = 0; base < m; base += SER_CHUNK) {
        const int cnt = m - base < SER_CHUNK ? m - base : SER_CHUNK;
        for (int i = threadIdx.x; i < cnt; i += CH_T) ser[i] = buf[chain_addr(base + i)];
        __syncthreads();
        if (threadIdx.x == 0) {
#pragma unroll 8
            for (int i = 0; i < cnt; i++) s += ser[i];
        }
        __syncthreads();
    }
    return s;
}

__global__ __launch_bounds__(CH_T) void k_finalize(Dev d) {
    __shared__ ChainLds<CH_EPT> L;
    __shared__ double ser[SER_CHUNK];
    State* st = d.st;
    if (!st->ev_active || st->stall) return;
    double usx = 0.0;
    if (!st->ev_finish) usx = st->nonneg ? block_chain_sum2<CH_EPT>(d.chain, st->m, CH_GUARD_BITS, L, nullptr) : serial_chain_sum(d.chain, st->m, ser);
    if (threadIdx.x == 0) finalize(d, usx);
}

// diagnostic entry: the block chain sum on an arbitrary buffer (tests)
template <int EPT>
__global__ __launch_bounds__(CH_T) void k_test_chain(const double* buf, int m, int guard_bits, double* out, ChainStats* stats) {
    __shared__ ChainLds<EPT> L;
    double r = g_chain_form == 1 ? block_chain_sum<EPT>(buf, m, guard_bits, L, stats) : block_chain_sum2<EPT>(buf, m, guard_bits, L, stats);
    if (threadIdx.x == 0) *out = r;
}

// ------------------------------------------------------------------ setup kernels
__global__ __launch_bounds__(256) void k_init(Dev d) { init_thread(d, blockIdx.x * 256 + threadIdx.x); }

__global__ __launch_bounds__(256) void k_synth(Dev d, uint64_t seed, int dist) {
    int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
    int64_t r = blockIdx.y;
    if (c >= d.n) return;
    double v = 0.0;
    if (r < c) v = synth_entry(d.n, r, c, seed, dist);
    else if (c < r) v = synth_entry(d.n, c, r, seed, dist);
    d.D[r * d.ld + c] = v;
}

// rows [row0, row0 + gridDim.y) of the packed strict upper triangle (staged chunk starting at packed
// index p0) -> the upper triangle of D
__global__ __launch_bounds__(256) void k_unpack(Dev d, const double* __restrict__ stage, int64_t p0, int32_t row0) {
    const int64_t r = row0 + (int64_t)blockIdx.y;
    const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (c <= r || c >= d.n) return;
    d.D[r * d.ld + c] = __builtin_nontemporal_load(stage + (packed_row_base(d.n, r) + c - p0));
}

// lower triangle := transpose of the upper one, diagonal := 0 (32x32 tiles through LDS)
__global__ __launch_bounds__(256) void k_mirror(Dev d) {
    __shared__ double tile[32][33];
    const int bx = blockIdx.x, by = blockIdx.y;  // tile row by, tile column bx
    if (bx < by) return;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    for (int k = ty; k < 32; k += 8) {
        const int64_t r = (int64_t)by * 32 + k, c = (int64_t)bx * 32 + tx;
        tile[k][tx] = (r < d.n && c < d.n && r < c) ? d.D[r * d.ld + c] : 0.0;
    }
    __syncthreads();
    for (int k = ty; k < 32; k += 8) {
        const int64_t r = (int64_t)bx * 32 + k, c = (int64_t)by * 32 + tx;  // the transposed tile
        if (r < d.n && c < d.n && c <= r) d.D[r * d.ld + c] = (c == r) ? 0.0 : tile[tx][k];
    }
}

// symmetric (bitwise), finite, zero diagonal: 32x32 tiles, transposed partner through LDS
__global__ __launch_bounds__(256) void k_validate(Dev d, int* bad) {
    __shared__ uint64_t tile[32][33];
    const int bx = blockIdx.x, by = blockIdx.y;
    if (bx > by) return;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    const uint64_t* D = reinterpret_cast<const uint64_t*>(d.D);
    for (int k = ty; k < 32; k += 8) {
        int r = bx * 32 + k, c = by * 32 + tx;  // the transposed tile
        tile[k][tx] = (r < d.n && c < d.n) ? D[(int64_t)r * d.ld + c] : 0;
    }
    __syncthreads();
    int flag = 0;
    for (int k = ty; k < 32; k += 8) {
        int r = by * 32 + k, c = bx * 32 + tx;
        if (r < d.n && c < d.n) {
            uint64_t v = D[(int64_t)r * d.ld + c];
            uint64_t t = tile[tx][k];
            if (v != t) flag = 1;
            if (((v >> 52) & 0x7FF) == 0x7FF) flag = 1;
            if (r == c && v != 0) flag = 1;
        }
    }
    if (flag) atomicOr(bad, 1);
}

__global__ __launch_bounds__(256) void k_stream(const double2* p, int64_t n16, double* sink) {
    double acc = 0.0;
    int64_t stride = (int64_t)gridDim.x * 256 * 4;
    for (int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x); i + 3 * (int64_t)gridDim.x * 256 < n16; i += stride) {
        double2 a = p[i], b = p[i + (int64_t)gridDim.x * 256], c = p[i + 2 * (int64_t)gridDim.x * 256],
                e = p[i + 3 * (int64_t)gridDim.x * 256];
        acc += (a.x + a.y) + (b.x + b.y) + (c.x + c.y) + (e.x + e.y);
    }
    if (acc == 123.456) *sink = acc;  // keep the loads alive
}

// ------------------------------------------------------------------ backend
#define HIPOK(x) ((last = (x)) == hipSuccess)

struct HipBackend {
    static constexpr int64_t kRowPad = SCAN_TH;
    static constexpr int64_t kColPad = SCR_TW;   // whole screening tiles (and scan tiles) stay in bounds
    // below this many taxa the fp32 copy is not even allocated (FNN_SCREEN_MIN_N, tests)
    int32_t screen_min_n() const { if (const char* e = std::getenv("FNN_SCREEN_MIN_N")) { int v = std::atoi(e); if (v >= 8) return v; } return 4096; }
    // events with fewer live nodes use the plain fp64 scan.  Windows in the end game pay while the matrix is not yet
    // averaged out: measured, n = 4096 gains 6 % with a floor of 512-1024 (17 events per window there), n = 32768 LOSES
    // 4 % with 1024 (after 30 000 merges the criterion values lie so close together that a window holds ~1 event).
    // Hence min(2048, n / 4), not below 512; FNN_SCREEN_MIN_M overrides.
    int screen_min_m = 2048;
    bool screen_min_m_fixed = false;
    int relaxed_grid = 0;  // workgroups of k_relaxed (FNN_RELAXED_GRID; 0 = by the number of live nodes)
    int relaxed_min = 0;   // > 0: Relaxed mode - events with more live nodes than this search (k_relaxed) instead of scanning
    void set_relaxed(int32_t min_active) { relaxed_min = min_active; }
    void set_problem_size(int32_t n) {
        if (screen_min_m_fixed) return;
        const int v = n / 4;
        screen_min_m = v < 512 ? 512 : (v > 2048 ? 2048 : v);
    }
    hipError_t last = hipSuccess;
    hipStream_t stream = nullptr;
    int device = 0;
    bool opened = false;
    // HIP-event timing of launches: 0 off, 1 the streaming scan kernels (bench.py's roofline figure), 2 every kernel of the
    // launch sequences (bench.py's `chain` object, an extra untimed run)
    int timing = 0;
    double class_ms[16] = {0};   // by kernel class: TC_*
    int64_t class_n[16] = {0};
    std::vector<hipEvent_t> ev_pool;
    std::vector<char> ev_kind;   // per event pair: 1 = screening pass (k_screen), 0 = plain fp64 scan (k_scan)
    size_t ev_used = 0;
    double scan_ms = 0.0, plain_ms = 0.0;       // k_screen launches / k_scan launches
    int64_t scan_launches = 0, plain_launches = 0;
    int* d_bad = nullptr;
    double* d_stage = nullptr;  // staging for fnn_set_packed_upper
    int64_t stage_cap = 0;
    // RCCL (dlopen'ed: the process may already hold PyTorch's copy of the library)
    void* rccl_lib = nullptr;
    void* rccl_comm = nullptr;
    int (*p_ncclCommInitRank)(void**, int, fnn_nccl_id, int) = nullptr;
    int (*p_ncclAllGather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
    int (*p_ncclCommDestroy)(void*) = nullptr;
    const char* (*p_ncclGetErrorString)(int) = nullptr;
    std::string comm_err;
    int scan_grid = 8192;   // workgroups of the scan (FNN_SCAN_GRID)
    int unsched_grid = 2048; // workgroups of the screening launches that only run when a lookahead window fails (FNN_UNSCHED_GRID)
    int emit_grid = 256;    // workgroups of k_emit (FNN_EMIT_GRID)
    int track_grid = 64;    // track workgroups of k_track (FNN_TRACK_GRID); one more computes the pending chain
    int rx_helpers_cfg = TRK_NHELP;  // helper workgroups of k_track for the exact ComputeRx sums (FNN_RX_HELPERS=0: the deciding workgroup does all four)
    int rx_helpers = 0;              // ... launched only while the run needs them: on after a batch of events with exact sums, off after one without
    int64_t rx_exact_seen = 0;
    void note_rx_exact(int64_t n_rx_exact) {  // (the host's look at the state after every batch)
        rx_helpers = n_rx_exact > rx_exact_seen ? rx_helpers_cfg : 0;
        rx_exact_seen = n_rx_exact;
    }
    unsigned track_tag = 0;      // tag of a k_track launch in the helpers' JOB word
    bool skip_unsched_scans = true; // FNN_UNSCHED_SCANS=1: keep the (mostly idle) scan kernels in unscheduled events
    int track_group = TRK_GROUP;  // k_track: workgroups per first-level arrival counter (FNN_TRACK_GROUP)
    bool defer_chain = false; // set by the engine: k_update closes the event, the exact u.Sx sum runs inside the next k_track
    bool fuse_events = false; // window events as ONE launch (k_track<.., true>; FNN_FUSE=1).  Off: measured 1.456 s against 1.408 s at 32768 taxa
    bool fused_last = false;  // the launch sequence being enqueued is such a fused one
    bool scan_nt = true;    // non-temporal matrix loads in the scan (FNN_SCAN_NT)
    bool ticks = false;     // FNN_TICKS=1: k_track records the phase split of its last workgroup (fnn_debug_event_ticks)

    std::string err() const { return comm_err.empty() ? std::string(hipGetErrorString(last)) : comm_err; }

    static void* open_rccl(const char* path, std::string& why) {
        // an explicit path is binding (a caller that names a library must not silently get another one)
        const char* cands[4] = {path, path && *path ? nullptr : std::getenv("FNN_RCCL_PATH"), path && *path ? nullptr : "librccl.so.1",
                                path && *path ? nullptr : "librccl.so"};
        for (const char* c : cands) {
            if (!c || !*c) continue;
            if (void* h = dlopen(c, RTLD_NOW | RTLD_LOCAL)) return h;
            why = dlerror();
        }
        return nullptr;
    }
    int32_t comm_init_rccl(int world, int rank, const uint8_t* id, const char* path) {
        comm_err.clear();
        if (!rccl_lib && !(rccl_lib = open_rccl(path, comm_err))) { comm_err = "cannot load librccl: " + comm_err; return FNN_ERCCL; }
        p_ncclCommInitRank = (decltype(p_ncclCommInitRank))dlsym(rccl_lib, "ncclCommInitRank");
        p_ncclAllGather = (decltype(p_ncclAllGather))dlsym(rccl_lib, "ncclAllGather");
        p_ncclCommDestroy = (decltype(p_ncclCommDestroy))dlsym(rccl_lib, "ncclCommDestroy");
        p_ncclGetErrorString = (decltype(p_ncclGetErrorString))dlsym(rccl_lib, "ncclGetErrorString");
        if (!p_ncclCommInitRank || !p_ncclAllGather || !p_ncclCommDestroy) { comm_err = "librccl lacks the expected symbols"; return FNN_ERCCL; }
        fnn_nccl_id uid;
        std::memcpy(uid.internal, id, sizeof(uid.internal));
        (void)hipSetDevice(device);
        int rc = p_ncclCommInitRank(&rccl_comm, world, uid, rank);
        if (rc != 0) {
            comm_err = std::string("ncclCommInitRank: ") + (p_ncclGetErrorString ? p_ncclGetErrorString(rc) : "error");
            rccl_comm = nullptr;
            return FNN_ERCCL;
        }
        return FNN_OK;
    }
    // (the ranks' status words before a host round trip, fnn_engine.h: enqueue_status_exchange)
    int32_t allgather_bytes_on_stream(const void* send, void* recv, size_t bytes_per_rank) {
        if (!rccl_comm) { comm_err = "RCCL communicator not initialised"; return FNN_ERCCL; }
        int rc = p_ncclAllGather(send, recv, bytes_per_rank, /*ncclInt8*/ 0, rccl_comm, stream);
        if (rc != 0) { comm_err = std::string("ncclAllGather: ") + (p_ncclGetErrorString ? p_ncclGetErrorString(rc) : "error"); return FNN_ERCCL; }
        return FNN_OK;
    }
    int32_t allgather_on_stream(const Dev& d, int nper) {
        if (!rccl_comm) { comm_err = "RCCL communicator not initialised"; return FNN_ERCCL; }
        int rc = p_ncclAllGather(d.gsend, d.grecv, sizeof(Cand) * (size_t)nper, /*ncclInt8*/ 0, rccl_comm, stream);
        if (rc != 0) { comm_err = std::string("ncclAllGather: ") + (p_ncclGetErrorString ? p_ncclGetErrorString(rc) : "error"); return FNN_ERCCL; }
        return FNN_OK;
    }

    int32_t open(int32_t dev) {
        int cnt = 0;
        if (!HIPOK(hipGetDeviceCount(&cnt)) || cnt <= 0)
            return fail(FNN_EHIP, "no HIP device available (" + err() + ")");
        if (dev < 0 || dev >= cnt) return fail(FNN_EINVAL, "device ordinal out of range");
        device = dev;
        if (!HIPOK(hipSetDevice(device))) return fail(FNN_EHIP, "hipSetDevice failed (" + err() + ")");
        if (!HIPOK(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking)))
            return fail(FNN_EHIP, "hipStreamCreate failed (" + err() + ")");
        if (const char* e = std::getenv("FNN_SCAN_GRID")) { int v = std::atoi(e); if (v >= 1 && v <= 65535) scan_grid = v; }
        if (const char* e = std::getenv("FNN_SCAN_NT")) scan_nt = std::atoi(e) != 0;
        if (const char* e = std::getenv("FNN_TICKS")) ticks = std::atoi(e) != 0;
        if (const char* e = std::getenv("FNN_UNSCHED_GRID")) { int v = std::atoi(e); if (v >= 1 && v <= 65535) unsched_grid = v; }
        if (const char* e = std::getenv("FNN_UNSCHED_SCANS")) skip_unsched_scans = std::atoi(e) == 0;
        if (const char* e = std::getenv("FNN_EMIT_GRID")) { int v = std::atoi(e); if (v >= 1 && v <= 65535) emit_grid = v; }
        if (const char* e = std::getenv("FNN_RX_HELPERS")) rx_helpers_cfg = std::atoi(e) != 0 ? TRK_NHELP : 0;
        if (const char* e = std::getenv("FNN_TRACK_GROUP")) { int v = std::atoi(e); if (v >= 2 && v <= 1024) track_group = v; }
        if (const char* e = std::getenv("FNN_FUSE")) fuse_events = std::atoi(e) != 0;
        if (const char* e = std::getenv("FNN_GRAPH")) graph_batches = std::atoi(e) != 0;
        if (const char* e = std::getenv("FNN_TRACK_GRID")) { int v = std::atoi(e); if (v >= 1 && v <= 1024) track_grid = v; }
        // the first-level arrival counters sit at d.ticket + 32 (g + 1), g < ceil(grid / group); word 32 * 65 is TRK_FLAG, 32 * 67
        // TRK_BAD and the array holds 32 * 72 words: at most 64 groups (the two switches are development aids, but an
        // inconsistent pair must not run the atomics into the flag words or out of bounds)
        static_assert(TRK_FLAG == 32 * 65 && TRK_BAD == 32 * 67, "ticket layout: groups 1..64, then the flag words");
        while ((track_grid + track_group - 1) / track_group > 64) track_group *= 2;
        if (const char* e = std::getenv("FNN_RELAXED_GRID")) { int v = std::atoi(e); if (v >= 1 && v <= RL_GMAX) relaxed_grid = v; }
        if (const char* e = std::getenv("FNN_SCREEN_MIN_M")) { int v = std::atoi(e); if (v >= 8) { screen_min_m = v; screen_min_m_fixed = true; } }
        opened = true;
        return FNN_OK;
    }
    void close() {
        if (!opened) return;
        (void)hipSetDevice(device);
        if (rccl_comm && p_ncclCommDestroy) (void)p_ncclCommDestroy(rccl_comm);
        rccl_comm = nullptr;
        for (hipEvent_t e : ev_pool) (void)hipEventDestroy(e);
        ev_pool.clear();
        if (d_bad) (void)hipFree(d_bad);
        d_bad = nullptr;
        if (d_stage) (void)hipFree(d_stage);
        d_stage = nullptr;
        stage_cap = 0;
        if (stream) (void)hipStreamDestroy(stream);
        stream = nullptr;
        opened = false;
    }
    void* alloc(size_t b) {
        void* p = nullptr;
        (void)hipSetDevice(device);
        if (!HIPOK(hipMalloc(&p, b ? b : 8))) {
            // the split-weight solver keeps its large buffers in a pool between calls (fnn_splits.hip): give them back and try again
            (void)hipGetLastError();
            (void)fnn_split_weights_release_cache();
            if (!HIPOK(hipMalloc(&p, b ? b : 8))) return nullptr;
        }
        return p;
    }
    void free(void* p) {
        if (p) (void)hipFree(p);
    }
    size_t max_records(int32_t) { return 65536; }
    int32_t memset(void* p, int v, size_t b) {
        return HIPOK(hipMemsetAsync(p, v, b, stream)) && HIPOK(hipStreamSynchronize(stream)) ? FNN_OK : FNN_EHIP;
    }
    int32_t h2d(void* d, const void* s, size_t b) {
        return HIPOK(hipMemcpyAsync(d, s, b, hipMemcpyHostToDevice, stream)) && HIPOK(hipStreamSynchronize(stream)) ? FNN_OK : FNN_EHIP;
    }
    int32_t d2h(void* d, const void* s, size_t b) {
        return HIPOK(hipMemcpyAsync(d, s, b, hipMemcpyDeviceToHost, stream)) && HIPOK(hipStreamSynchronize(stream)) ? FNN_OK : FNN_EHIP;
    }
    int32_t copy2d(double* d, int64_t ldd, const double* s, int64_t lds, int64_t w, int64_t h, hipMemcpyKind kind) {
        return HIPOK(hipMemcpy2DAsync(d, (size_t)ldd * 8, s, (size_t)lds * 8, (size_t)w * 8, (size_t)h, kind, stream)) &&
                       HIPOK(hipStreamSynchronize(stream))
                   ? FNN_OK
                   : FNN_EHIP;
    }
    int32_t h2d_2d(double* d, int64_t ldd, const double* s, int64_t lds, int64_t w, int64_t h) {
        return copy2d(d, ldd, s, lds, w, h, hipMemcpyHostToDevice);
    }
    int32_t d2d_2d(double* d, int64_t ldd, const double* s, int64_t lds, int64_t w, int64_t h) {
        return copy2d(d, ldd, s, lds, w, h, hipMemcpyDeviceToDevice);
    }
    int32_t d2h_2d(double* d, int64_t ldd, const double* s, int64_t lds, int64_t w, int64_t h) {
        return copy2d(d, ldd, s, lds, w, h, hipMemcpyDeviceToHost);
    }

    enum { TC_SCAN = 0, TC_SCREEN = 1, TC_TRACK = 2, TC_DECIDE = 3, TC_UPDATE = 4, TC_EMIT = 5, TC_RESOLVE = 6, TC_OTHER = 7, TC_GATHER = 8, TC_MERGE = 9 };
    void drain_timing() {
        for (size_t i = 0; i + 1 < ev_used && i / 2 < ev_kind.size(); i += 2) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, ev_pool[i], ev_pool[i + 1]) == hipSuccess) {
                const int c = ev_kind[i / 2] & 15;
                class_ms[c] += ms;
                class_n[c]++;
                if (c == TC_SCREEN) scan_ms += ms;
                else if (c == TC_SCAN) plain_ms += ms;
            }
        }
        ev_used = 0;
        ev_kind.clear();
    }
    // a launch between two event records (only when both events could be had: a failed hipEventCreate skips the timing
    // of that launch altogether, so durations and classes cannot get out of step)
    template <class F>
    void timed(int cls, bool on, F&& launch) {
        hipEvent_t e0 = on && !capturing ? next_event() : nullptr, e1 = e0 ? next_event() : nullptr;
        if (e0 && !e1) ev_used--;
        if (e0 && e1) {
            ev_kind.push_back((char)cls);
            (void)hipEventRecord(e0, stream);
            launch();
            (void)hipEventRecord(e1, stream);
        } else launch();
    }
    // Experiment (FNN_GRAPH=1, off by default): a batch of events captured into a hipGraph and launched as one - to see what the
    // runtime's per-dispatch handling contributes to the kernels' traced durations (the launch parameters differ from batch to batch,
    // so every batch is captured and instantiated anew: this measures the device side, it is not a way to run faster as it stands).
    double graph_exec_s = 0.0;
    int graph_count = 0;
    bool graph_batches = false, capturing = false;  // (no event records inside a capture: the scans' live timing is off in this mode)
    int32_t capture_begin() {
        capturing = HIPOK(hipStreamBeginCapture(stream, hipStreamCaptureModeThreadLocal));
        return capturing ? FNN_OK : FNN_EHIP;
    }
    int32_t capture_end_launch() {
        hipGraph_t g = nullptr;
        hipGraphExec_t ge = nullptr;
        capturing = false;
        if (!HIPOK(hipStreamEndCapture(stream, &g))) return FNN_EHIP;
        bool fine = HIPOK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        const auto t0 = std::chrono::steady_clock::now();
        fine = fine && HIPOK(hipGraphLaunch(ge, stream)) && HIPOK(hipStreamSynchronize(stream));
        graph_exec_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (++graph_count % 64 == 0) std::fprintf(stderr, "[fnn] graphs: %d launched, %.4f s between launch and completion\n", graph_count, graph_exec_s);
        if (ge) (void)hipGraphExecDestroy(ge);
        (void)hipGraphDestroy(g);
        return fine ? FNN_OK : FNN_EHIP;
    }
    int32_t sync() {
        if (!HIPOK(hipStreamSynchronize(stream))) return FNN_EHIP;
        if (timing) drain_timing();
        return FNN_OK;
    }
    void collect_timing(fnn_stats& s) {
        s.t_scan_s = scan_ms * 1e-3;
        s.scan_launches = scan_launches;
        s.t_plain_s = plain_ms * 1e-3;
        s.plain_launches = plain_launches;
    }
    void reset_timing() {
        scan_ms = plain_ms = 0.0; scan_launches = plain_launches = 0; ev_used = 0; ev_kind.clear();
        for (int c = 0; c < 16; c++) { class_ms[c] = 0.0; class_n[c] = 0; }
    }

    hipEvent_t next_event() {
        if (ev_used == ev_pool.size()) {
            hipEvent_t e;
            if (hipEventCreate(&e) != hipSuccess) return nullptr;
            ev_pool.push_back(e);
        }
        return ev_pool[ev_used++];
    }

    static dim3 grid1(int32_t cnt) { return dim3((unsigned)((cnt + 255) / 256 > 0 ? (cnt + 255) / 256 : 1)); }

    int32_t launch_synth(const Dev& d, uint64_t seed, int32_t dist) {
        dim3 g((unsigned)((d.n + 255) / 256), (unsigned)d.n);
        hipLaunchKernelGGL(k_synth, g, dim3(256), 0, stream, d, seed, dist);
        return HIPOK(hipGetLastError()) && HIPOK(hipStreamSynchronize(stream)) ? FNN_OK : FNN_EHIP;
    }
    int32_t unpack_rows(const Dev& d, const double* src, int64_t p0, int64_t entries, int32_t row0, int32_t cnt) {
        if (entries > stage_cap) {
            if (d_stage) (void)hipFree(d_stage);
            d_stage = nullptr;
            stage_cap = 0;
            if (!HIPOK(hipMalloc((void**)&d_stage, sizeof(double) * (size_t)entries))) return FNN_EHIP;
            stage_cap = entries;
        }
        if (!HIPOK(hipMemcpyAsync(d_stage, src, sizeof(double) * (size_t)entries, hipMemcpyHostToDevice, stream))) return FNN_EHIP;
        dim3 g((unsigned)((d.n + 255) / 256), (unsigned)cnt);
        hipLaunchKernelGGL(k_unpack, g, dim3(256), 0, stream, d, (const double*)d_stage, p0, row0);
        return HIPOK(hipGetLastError()) && HIPOK(hipStreamSynchronize(stream)) ? FNN_OK : FNN_EHIP;
    }
    int32_t launch_mirror(const Dev& d) {
        if (d_stage) (void)hipFree(d_stage);
        d_stage = nullptr;
        stage_cap = 0;
        unsigned t = (unsigned)((d.n + 31) / 32);
        hipLaunchKernelGGL(k_mirror, dim3(t, t), dim3(256), 0, stream, d);
        return HIPOK(hipGetLastError()) && HIPOK(hipStreamSynchronize(stream)) ? FNN_OK : FNN_EHIP;
    }
    int32_t launch_validate(const Dev& d, int32_t* bad) {
        if (!d_bad && !HIPOK(hipMalloc((void**)&d_bad, sizeof(int)))) return FNN_EHIP;
        if (!HIPOK(hipMemsetAsync(d_bad, 0, sizeof(int), stream))) return FNN_EHIP;
        unsigned t = (unsigned)((d.n + 31) / 32);
        hipLaunchKernelGGL(k_validate, dim3(t, t), dim3(256), 0, stream, d, d_bad);
        if (!HIPOK(hipGetLastError())) return FNN_EHIP;
        int hb = 0;
        if (!HIPOK(hipMemcpyAsync(&hb, d_bad, sizeof(int), hipMemcpyDeviceToHost, stream)) ||
            !HIPOK(hipStreamSynchronize(stream)))
            return FNN_EHIP;
        *bad = hb;
        return FNN_OK;
    }
    int32_t launch_prep_screen(const Dev& d, int64_t nrows) {
        const int64_t want = (nrows * d.ld + 255) / 256;
        hipLaunchKernelGGL(k_prep_screen, dim3((unsigned)(want < 8192 ? (want > 0 ? want : 1) : 8192)), dim3(256), 0, stream, d, nrows);
        return HIPOK(hipGetLastError()) ? FNN_OK : FNN_EHIP;
    }
    int32_t launch_init(const Dev& d) {
        reset_timing();
        rx_exact_seen = 0; rx_helpers = 0;  // (a new run: the device's counter starts again)
        hipLaunchKernelGGL(k_init, grid1(d.n), dim3(256), 0, stream, d);
        return HIPOK(hipGetLastError()) ? FNN_OK : FNN_EHIP;
    }

    dim3 scan_dims(const Dev& d, int32_t m_bound) const {
        int nt = (scan_tile_count(m_bound) + d.world - 1) / d.world;  // tiles of this rank
        return dim3((unsigned)(nt < scan_grid ? (nt > 0 ? nt : 1) : scan_grid));
    }
    bool screen_off = false;  // set by the engine for a matrix that can open no window (negative entries): plain fp64 scans throughout
    bool keep_generic_screen() const { return false; }  // (the CPU emulation of the tests keeps the mixed-sign screening pass alive)
    bool use_screen(const Dev& d, int32_t m_bound) const { return d.H != nullptr && !screen_off && m_bound >= screen_min_m; }
    // the scan of one event; returns the number of per-workgroup records it leaves in d.recs.
    // With lookahead windows (d.la) k_track goes first: it either serves the event from the open
    // window (the scan kernels then return at once) or lets the scan run.  `sched`: the host's
    // schedule opens a new window at this event, so the scan is certain to run: only those
    // launches (and the plain fp64 scans) are timed for the roofline figure.
    int enqueue_scan(const Dev& d, int32_t m_bound, bool sched) {
        const bool screen = use_screen(d, m_bound);
        const bool tscan = timing != 0 && (sched || !screen), tall = timing == 2;
        // (an unscheduled event is launched without scan kernels: if its window cannot serve it, the
        //  device stalls - this and the following such events do nothing - until the host, which
        //  looks at the state every batch, launches an event with a scan)
        const bool has_scan = sched || !screen || !skip_unsched_scans;
        // a WINDOW event in one launch (k_track with fuse: tracking, decide step and the update; no k_update follows): only launch
        // sequences without scan kernels, with the row sum deferred, and while every block of 1024 columns finds a tracking workgroup
        fused_last = fuse_events && d.la && screen && !has_scan && defer_chain;
        const int fz = fused_last ? (m_bound + BULK_COLS - 1) / BULK_COLS : 0;  // column workgroups of a fused launch
        if (d.la) timed(TC_TRACK, tall, [&]() {
            track_tag = (track_tag % 0x7FFFFFEu) + 1u;  // (never 0: the JOB word and the plan message start out as 0)
            const int fb = (sched || !screen) ? 1 : 0, hs = has_scan ? 1 : 0, tk = ticks ? 1 : 0;
            if (rx_helpers > 0 && fz) hipLaunchKernelGGL((k_track<true, true>), dim3(track_grid + 1 + TRK_NHELP + fz), dim3(TRK_THREADS), 0, stream, d, fb, fb, hs, track_group, tk, track_tag, fz);
            else if (rx_helpers > 0) hipLaunchKernelGGL((k_track<true, false>), dim3(track_grid + 1 + TRK_NHELP), dim3(TRK_THREADS), 0, stream, d, fb, fb, hs, track_group, tk, track_tag, 0);
            else if (fz) hipLaunchKernelGGL((k_track<false, true>), dim3(track_grid + 1 + fz), dim3(TRK_THREADS), 0, stream, d, fb, fb, hs, track_group, tk, track_tag, fz);
            else hipLaunchKernelGGL((k_track<false, false>), dim3(track_grid + 1), dim3(TRK_THREADS), 0, stream, d, fb, fb, hs, track_group, tk, track_tag, 0);
        });
        int nrecs;
        if (screen && !has_scan) nrecs = 0;  // (a window event: the tail of k_track decides; no decide kernel follows)
        else if (screen) {
            int nt = (tri_tile_count(m_bound, SCR_TH, SCR_R) + d.world - 1) / d.world;
            const int want = sched ? scan_grid : unsched_grid;
            dim3 gs((unsigned)(nt < want ? (nt > 0 ? nt : 1) : want));
            timed(TC_SCREEN, tscan, [&]() {
                if (sched) {
                    if (scan_nt) hipLaunchKernelGGL((k_screen<true, true>), gs, dim3(256), 0, stream, d);
                    else hipLaunchKernelGGL((k_screen<false, true>), gs, dim3(256), 0, stream, d);
                } else {
                    if (scan_nt) hipLaunchKernelGGL((k_screen<true, false>), gs, dim3(256), 0, stream, d);
                    else hipLaunchKernelGGL((k_screen<false, false>), gs, dim3(256), 0, stream, d);
                }
            });
            if (d.la) timed(TC_EMIT, tall, [&]() { hipLaunchKernelGGL(k_emit, dim3(emit_grid), dim3(256), 0, stream, d); });
            timed(TC_RESOLVE, tall, [&]() { hipLaunchKernelGGL(k_resolve, dim3(RES_BLOCKS), dim3(1024), 0, stream, d); });
            nrecs = RES_BLOCKS;
            if (tscan) scan_launches++;
        } else {
            dim3 gs = scan_dims(d, m_bound);
            // Relaxed mode: the search first; the scan returns at once if it found the pair (it runs when fewer
            // nodes are live than the bound the host knows, i.e. at the switch to the full scans)
            if (relaxed_min > 0 && m_bound > relaxed_min)
                timed(TC_OTHER, tall, [&]() {
                    // one workgroup streams ~40 GB/s: from ~12 000 live nodes on the row pass is spread over several
                    // (a hand-over between workgroups costs ~3 us per row minimum)
                    int g = relaxed_grid > 0 ? relaxed_grid : (m_bound >= 12288 ? m_bound / 4096 : 1);
                    if (g > RL_GMAX) g = RL_GMAX;
                    hipLaunchKernelGGL(k_relaxed, dim3(g), dim3(RL_T), 0, stream, d, ticks ? 1 : 0);
                });
            timed(TC_SCAN, tscan, [&]() {
                if (scan_nt) hipLaunchKernelGGL(k_scan<true>, gs, dim3(SCAN_THREADS), 0, stream, d);
                else hipLaunchKernelGGL(k_scan<false>, gs, dim3(SCAN_THREADS), 0, stream, d);
            });
            nrecs = (int)gs.x;
            if (tscan) plain_launches++;
        }
        return nrecs;
    }
    // everything after the scan; `src` holds the nrecs candidate records to reduce (0: a window event, already
    // decided by the tail of k_track - or stalled)
    void enqueue_rest(const Dev& d, int32_t m_bound, const Cand* src, int nrecs) {
        dim3 g1 = grid1(m_bound);
        const bool tall = timing == 2;
        if (nrecs > 0) timed(TC_DECIDE, tall, [&]() {
            track_tag = (track_tag % 0x7FFFFFEu) + 1u;
            if (rx_helpers > 0) hipLaunchKernelGGL(k_decide<true>, dim3(1 + TRK_NHELP), dim3(CH_T), 0, stream, d, src, nrecs, track_tag);
            else hipLaunchKernelGGL(k_decide<false>, dim3(1), dim3(CH_T), 0, stream, d, src, nrecs, track_tag);
        });
        if (nrecs == 0 && fused_last) return;  // (the window event's k_track has done the update itself)
        timed(TC_UPDATE, tall, [&]() { hipLaunchKernelGGL(k_update, dim3(g1.x + 1), dim3(256), 0, stream, d, defer_chain ? 1 : 0, ticks ? 1 : 0); });
        if (!defer_chain) timed(TC_OTHER, tall, [&]() { hipLaunchKernelGGL(k_finalize, dim3(1), dim3(CH_T), 0, stream, d); });
    }
    // single GPU: the whole event
    int32_t launch_event(const Dev& d, int32_t m_bound, bool sched) {
        if (m_bound < 1) m_bound = 1;
        int nrecs = enqueue_scan(d, m_bound, sched);
        enqueue_rest(d, m_bound, (const Cand*)d.recs, nrecs);
        return HIPOK(hipGetLastError()) ? FNN_OK : FNN_EHIP;
    }
    // a pending deferred row sum, before the host reads the state
    int32_t launch_chain_flush(const Dev& d) {
        hipLaunchKernelGGL(k_chain_flush, dim3(1), dim3(CH_T), 0, stream, d);
        return HIPOK(hipGetLastError()) ? FNN_OK : FNN_EHIP;
    }
    // several GPUs: scan of this rank's tiles ... (all-gather of the candidate records) ... the rest
    int32_t launch_event_scan(const Dev& d, int32_t m_bound, int32_t* nper) {
        if (m_bound < 1) m_bound = 1;
        int nrecs = enqueue_scan(d, m_bound, true);
        if (nrecs == RES_BLOCKS && use_screen(d, m_bound)) {
            *nper = nrecs;  // k_resolve has written its per-workgroup records straight into d.gsend
        } else {
            hipLaunchKernelGGL(k_reduce_local, dim3(1), dim3(1024), 0, stream, d, nrecs);
            *nper = 1;
        }
        return HIPOK(hipGetLastError()) ? FNN_OK : FNN_EHIP;
    }
    // several ranks with lookahead windows: a base scan's sharded part ... (exchange) ... merge + the rest
    int32_t launch_wx_scan(const Dev& d, int32_t m_bound) {
        if (m_bound < 1) m_bound = 1;
        (void)enqueue_scan(d, m_bound, true);
        return HIPOK(hipGetLastError()) ? FNN_OK : FNN_EHIP;
    }
    int32_t allgather_wx_on_stream(const Dev& d, size_t bytes) {
        if (!rccl_comm) { comm_err = "RCCL communicator not initialised"; return FNN_ERCCL; }
        int rc = 0;
        timed(TC_GATHER, timing == 2, [&] { rc = p_ncclAllGather(d.wsend, d.wrecv, bytes, /*ncclInt8*/ 0, rccl_comm, stream); });
        if (rc != 0) { comm_err = std::string("ncclAllGather: ") + (p_ncclGetErrorString ? p_ncclGetErrorString(rc) : "error"); return FNN_ERCCL; }
        return FNN_OK;
    }
    int32_t launch_wx_rest(const Dev& d, int32_t m_bound) {
        if (m_bound < 1) m_bound = 1;
        timed(TC_MERGE, timing == 2, [&] { hipLaunchKernelGGL(k_merge, dim3(1), dim3(1024), 0, stream, d); });
        enqueue_rest(d, m_bound, (const Cand*)d.grecv, d.world * GATHER_RECS);
        return HIPOK(hipGetLastError()) ? FNN_OK : FNN_EHIP;
    }
    int32_t launch_event_rest(const Dev& d, int32_t m_bound, int32_t ntotal) {
        if (m_bound < 1) m_bound = 1;
        enqueue_rest(d, m_bound, (const Cand*)d.grecv, ntotal);
        return HIPOK(hipGetLastError()) ? FNN_OK : FNN_EHIP;
    }
};

using HipEngine = Engine<HipBackend>;

}  // namespace fnn

// ---------------------------------------------------------------------- C ABI
struct fnn_handle {
    fnn::HipEngine eng;
};

#define FNN_TRY(body)                                                     \
    try {                                                                 \
        body                                                              \
    } catch (const std::bad_alloc&) {                                     \
        return fnn::fail(FNN_ENOMEM, "out of host memory");               \
    } catch (const std::exception& e) {                                   \
        return fnn::fail(FNN_ESTATE, std::string("exception: ") + e.what()); \
    }

extern "C" {

int32_t fnn_abi_version(void) { return FASTNN_ABI_VERSION; }
const char* fnn_last_error(void) { return fnn::g_last_error.c_str(); }

int32_t fnn_device_count(void) {
    int cnt = 0;
    hipError_t e = hipGetDeviceCount(&cnt);
    if (e != hipSuccess) return fnn::fail(FNN_EHIP, std::string("hipGetDeviceCount: ") + hipGetErrorString(e));
    return cnt;
}

int32_t fnn_create(int32_t n, const fnn_opts* opts, fnn_handle** out) {
    if (!out) return fnn::fail(FNN_EINVAL, "fnn_create: out is NULL");
    *out = nullptr;
    FNN_TRY(
        fnn_handle* h = new fnn_handle();
        int32_t rc = h->eng.create(n, opts);
        if (rc != FNN_OK) { h->eng.destroy(); delete h; return rc; }
        *out = h;
        return FNN_OK;
    )
}

int32_t fnn_destroy(fnn_handle* h) {
    if (!h) return FNN_OK;
    h->eng.destroy();
    delete h;
    return FNN_OK;
}

#define FNN_NEED(h) if (!(h)) return fnn::fail(FNN_EINVAL, "NULL handle")

int32_t fnn_set_rows(fnn_handle* h, int32_t row0, int32_t nrows, const double* rows, int64_t ld_in) {
    FNN_NEED(h);
    FNN_TRY(return h->eng.set_rows(row0, nrows, rows, ld_in);)
}
int32_t fnn_set_packed_upper(fnn_handle* h, const double* packed) {
    if (!h) return fnn::fail(FNN_EINVAL, "fnn_set_packed_upper: null handle");
    FNN_TRY(return h->eng.set_packed_upper(packed);)
}
int32_t fnn_set_matrix_device(fnn_handle* h, const double* d_matrix, int64_t ld_in) {
    FNN_NEED(h);
    FNN_TRY(return h->eng.set_matrix_device(d_matrix, ld_in);)
}
int32_t fnn_synth(fnn_handle* h, uint64_t seed, int32_t dist) {
    FNN_NEED(h);
    FNN_TRY(return h->eng.synth(seed, dist);)
}
int32_t fnn_run(fnn_handle* h, int32_t* order_out, fnn_stats* stats) {
    FNN_NEED(h);
    if (!order_out) return fnn::fail(FNN_EINVAL, "fnn_run: order_out is NULL");
    FNN_TRY(return h->eng.run(order_out, stats);)
}
int32_t fnn_begin(fnn_handle* h) {
    FNN_NEED(h);
    FNN_TRY(return h->eng.begin();)
}
int32_t fnn_step(fnn_handle* h, fnn_event* ev) {
    FNN_NEED(h);
    FNN_TRY(return h->eng.step(ev);)
}
int32_t fnn_finish(fnn_handle* h, int32_t* order_out) {
    FNN_NEED(h);
    FNN_TRY(return h->eng.finish(order_out);)
}
int64_t fnn_get_events(fnn_handle* h, fnn_event* out, int64_t max_events) {
    if (!h) return fnn::fail(FNN_EINVAL, "NULL handle");
    return h->eng.get_events(out, max_events);
}
int32_t fnn_get_counts(fnn_handle* h, int32_t* num_active, int32_t* num_clusters, int32_t* num_nodes) {
    FNN_NEED(h);
    FNN_TRY(
        if (!h->eng.begun) return fnn::fail(FNN_ESTATE, "fnn_get_counts: call fnn_begin first");
        int32_t rc = h->eng.pull_state();
        if (rc != FNN_OK) return rc;
        if (num_active) *num_active = h->eng.hst.m;
        if (num_clusters) *num_clusters = h->eng.hst.c;
        if (num_nodes) *num_nodes = h->eng.hst.num_nodes;
        return FNN_OK;
    )
}
int32_t fnn_get_nodes(fnn_handle* h, int32_t* id, int32_t* nbr_id, double* Sx) {
    FNN_NEED(h);
    FNN_TRY(return h->eng.get_nodes(id, nbr_id, Sx);)
}
int32_t fnn_get_matrix(fnn_handle* h, double* out, int64_t ld_out) {
    FNN_NEED(h);
    FNN_TRY(return h->eng.get_matrix(out, ld_out);)
}
int32_t fnn_get_live_matrix(fnn_handle* h, double* out) {
    FNN_NEED(h);
    if (!out) return fnn::fail(FNN_EINVAL, "fnn_get_live_matrix: out is NULL");
    FNN_TRY(return h->eng.get_live_matrix(out);)
}
int64_t fnn_debug_window_log(fnn_handle* h, double* out, int64_t max_records) {
    if (!h) return fnn::fail(FNN_EINVAL, "NULL handle");
    if (h->eng.pull_state() != FNN_OK) return FNN_EHIP;
    int64_t k = h->eng.hst.n_base_scans < fnn::LA_LOGCAP ? h->eng.hst.n_base_scans : fnn::LA_LOGCAP;
    if (k > max_records) k = max_records;
    if (out && k > 0 && h->eng.be.d2h(out, h->eng.dev.lalog, sizeof(double) * 5 * (size_t)k) != FNN_OK) return FNN_EHIP;
    return k;
}
int32_t fnn_debug_event_ticks(fnn_handle* h, int64_t* out8) {
    FNN_NEED(h);
    if (!out8) return fnn::fail(FNN_EINVAL, "fnn_debug_event_ticks: out8 is NULL");
    if (h->eng.be.d2h(out8, h->eng.dev.ticks, sizeof(int64_t) * 8) != FNN_OK) return FNN_EHIP;
    return FNN_OK;
}
int32_t fnn_debug_update_ticks(fnn_handle* h, int64_t* out8) {
    FNN_NEED(h);
    if (!out8) return fnn::fail(FNN_EINVAL, "fnn_debug_update_ticks: out8 is NULL");
    if (h->eng.be.d2h(out8, h->eng.dev.ticks + 8, sizeof(int64_t) * 8) != FNN_OK) return FNN_EHIP;
    return FNN_OK;
}
int32_t fnn_debug_relaxed_ticks(fnn_handle* h, int64_t* out4) {
    FNN_NEED(h);
    if (!out4) return fnn::fail(FNN_EINVAL, "fnn_debug_relaxed_ticks: out4 is NULL");
    if (h->eng.be.d2h(out4, h->eng.dev.ticks + 24, sizeof(int64_t) * 4) != FNN_OK) return FNN_EHIP;
    return FNN_OK;
}
int32_t fnn_debug_decide_ticks(fnn_handle* h, int64_t* out4) {
    FNN_NEED(h);
    if (!out4) return fnn::fail(FNN_EINVAL, "fnn_debug_decide_ticks: out4 is NULL");
    if (h->eng.be.d2h(out4, h->eng.dev.ticks + 16, sizeof(int64_t) * 4) != FNN_OK) return FNN_EHIP;
    return FNN_OK;
}
int32_t fnn_debug_update_wg_ticks(fnn_handle* h, int64_t* out768) {
    FNN_NEED(h);
    if (!out768) return fnn::fail(FNN_EINVAL, "fnn_debug_update_wg_ticks: out768 is NULL");
    if (h->eng.be.d2h(out768, h->eng.dev.ticks + 32, sizeof(int64_t) * 3 * fnn::TICK_WG) != FNN_OK) return FNN_EHIP;
    return FNN_OK;
}
int32_t fnn_debug_plan_ticks(fnn_handle* h, int64_t* out4) {
    FNN_NEED(h);
    if (!out4) return fnn::fail(FNN_EINVAL, "fnn_debug_plan_ticks: out4 is NULL");
    if (h->eng.be.d2h(out4, h->eng.dev.ticks + 20, sizeof(int64_t) * 4) != FNN_OK) return FNN_EHIP;
    return FNN_OK;
}
int32_t fnn_set_scan_timing(fnn_handle* h, int32_t enable) {
    FNN_NEED(h);
    h->eng.be.timing = enable < 0 ? 0 : (enable > 2 ? 2 : enable);
    return FNN_OK;
}
int32_t fnn_get_kernel_times(fnn_handle* h, double* ms8, int64_t* launches8) {
    FNN_NEED(h);
    if (!ms8 || !launches8) return fnn::fail(FNN_EINVAL, "fnn_get_kernel_times: NULL output");
    for (int c = 0; c < 8; c++) { ms8[c] = h->eng.be.class_ms[c]; launches8[c] = h->eng.be.class_n[c]; }
    return FNN_OK;
}

int32_t fnn_get_exchange_times(fnn_handle* h, double* ms2, int64_t* launches2) {
    FNN_NEED(h);
    if (!ms2 || !launches2) return fnn::fail(FNN_EINVAL, "fnn_get_exchange_times: NULL output");
    for (int c = 0; c < 2; c++) { ms2[c] = h->eng.be.class_ms[8 + c]; launches2[c] = h->eng.be.class_n[8 + c]; }
    return FNN_OK;
}

int32_t fnn_canonical_order_f64(const double* D, int32_t n, int64_t ld, const fnn_opts* opts,
                                int32_t* order_out, fnn_stats* stats) {
    if (n < 0 || !order_out || (n > 0 && !D)) return fnn::fail(FNN_EINVAL, "fnn_canonical_order_f64: bad arguments");
    if (n <= 3) {  // NetMakerOriginal.java:133-140, no device needed
        for (int32_t i = 0; i <= n; i++) order_out[i] = i;
        if (stats) *stats = fnn_stats{};
        return FNN_OK;
    }
    fnn_handle* h = nullptr;
    int32_t rc = fnn_create(n, opts, &h);
    if (rc != FNN_OK) return rc;
    rc = fnn_set_rows(h, 0, n, D, ld);
    if (rc == FNN_OK) rc = fnn_run(h, order_out, stats);
    std::string keep = fnn::g_last_error;
    fnn_destroy(h);
    fnn::g_last_error = keep;
    return rc;
}

int32_t fnn_comm_unique_id(uint8_t* id_out, const char* rccl_path) {
    if (!id_out) return fnn::fail(FNN_EINVAL, "fnn_comm_unique_id: id_out is NULL");
    std::string why;
    void* lib = fnn::HipBackend::open_rccl(rccl_path, why);
    if (!lib) return fnn::fail(FNN_ERCCL, "cannot load librccl: " + why);
    auto get = (int (*)(fnn::fnn_nccl_id*))dlsym(lib, "ncclGetUniqueId");
    if (!get) return fnn::fail(FNN_ERCCL, "librccl lacks ncclGetUniqueId");
    fnn::fnn_nccl_id uid;
    int rc = get(&uid);
    if (rc != 0) return fnn::fail(FNN_ERCCL, "ncclGetUniqueId failed");
    std::memcpy(id_out, uid.internal, FNN_COMM_ID_BYTES);
    return FNN_OK;
}
int32_t fnn_comm_probe(const char* rccl_path) {
    std::string why;
    void* lib = fnn::HipBackend::open_rccl(rccl_path, why);
    if (!lib) return fnn::fail(FNN_ERCCL, "cannot load librccl: " + why);
    for (const char* sym : {"ncclGetUniqueId", "ncclCommInitRank", "ncclAllGather", "ncclCommDestroy"})
        if (!dlsym(lib, sym)) return fnn::fail(FNN_ERCCL, std::string("librccl lacks ") + sym);
    return FNN_OK;
}
int32_t fnn_comm_init_rccl(fnn_handle* h, int32_t world, int32_t rank, const uint8_t* id, const char* rccl_path) {
    FNN_NEED(h);
    if (!id) return fnn::fail(FNN_EINVAL, "fnn_comm_init_rccl: id is NULL");
    FNN_TRY(
        int32_t rc = h->eng.comm_set(1, world, rank);
        if (rc != FNN_OK) return rc;
        if (h->eng.comm_mode == 0) return FNN_OK;
        rc = h->eng.be.comm_init_rccl(world, rank, id, rccl_path);
        if (rc != FNN_OK) { h->eng.comm_set(0, 1, 0); return fnn::fail(rc, h->eng.be.err()); }
        return FNN_OK;
    )
}
int32_t fnn_comm_init_host(fnn_handle* h, int32_t world, int32_t rank, fnn_allgather_fn fn, void* ctx) {
    FNN_NEED(h);
    if (world > 1 && !fn) return fnn::fail(FNN_EINVAL, "fnn_comm_init_host: callback is NULL");
    FNN_TRY(
        int32_t rc = h->eng.comm_set(2, world, rank);
        if (rc != FNN_OK) return rc;
        h->eng.host_fn = fn;
        h->eng.host_ctx = ctx;
        return FNN_OK;
    )
}

int32_t fnn_test_chain_sum(int32_t device, const double* host_buf, int32_t m, int32_t guard_bits, int32_t ept,
                           double* out, int32_t* stats4) {
    if (!host_buf || m < 0 || !out || ept != fnn::CH_EPT)
        return fnn::fail(FNN_EINVAL, "fnn_test_chain_sum: bad arguments (ept must be 32)");
    hipError_t e;
    if ((e = hipSetDevice(device)) != hipSuccess) return fnn::fail(FNN_EHIP, hipGetErrorString(e));
    // the kernel reads the chunk-interleaved layout the engine's producers write
    const size_t cap = (size_t)fnn::round_up(m > 0 ? m : 1, fnn::CH_SC);
    std::vector<double> perm(cap, 0.0);
    for (int32_t i = 0; i < m; i++) perm[(size_t)fnn::chain_addr(i)] = host_buf[i];
    double* dbuf = nullptr; double* dout = nullptr; fnn::ChainStats* dst = nullptr;
    if (hipMalloc((void**)&dbuf, sizeof(double) * cap) != hipSuccess || hipMalloc((void**)&dout, 8) != hipSuccess ||
        hipMalloc((void**)&dst, sizeof(fnn::ChainStats)) != hipSuccess) {
        (void)hipFree(dbuf); (void)hipFree(dout); (void)hipFree(dst);
        return fnn::fail(FNN_ENOMEM, "fnn_test_chain_sum: hipMalloc failed");
    }
    (void)hipMemcpy(dbuf, perm.data(), sizeof(double) * cap, hipMemcpyHostToDevice);
    {
        int stop = 0;
        if (const char* ev = std::getenv("FNN_CHAIN_STOP")) stop = std::atoi(ev);
        (void)hipMemcpyToSymbol(HIP_SYMBOL(fnn::g_chain_stop_after), &stop, sizeof(int));
        int form = stop ? 1 : 2;
        if (const char* ev = std::getenv("FNN_CHAIN_FORM")) form = std::atoi(ev);
        (void)hipMemcpyToSymbol(HIP_SYMBOL(fnn::g_chain_form), &form, sizeof(int));
    }
    hipLaunchKernelGGL(fnn::k_test_chain<fnn::CH_EPT>, dim3(1), dim3(fnn::CH_T), 0, 0, dbuf, m, guard_bits, dout, dst);
    e = hipDeviceSynchronize();
    if (std::getenv("FNN_CHAIN_TIME")) {  // development aid: average duration of the kernel
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        (void)hipEventRecord(e0, 0);
        for (int i = 0; i < 50; i++)
            hipLaunchKernelGGL(fnn::k_test_chain<fnn::CH_EPT>, dim3(1), dim3(fnn::CH_T), 0, 0, dbuf, m, guard_bits, dout, dst);
        (void)hipEventRecord(e1, 0);
        (void)hipEventSynchronize(e1);
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, e0, e1);
        std::fprintf(stderr, "[fnn] k_test_chain m=%d: %.2f us per launch\n", m, ms * 1e3 / 50);
        (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    }
    fnn::ChainStats hs{0, 0, 0, 0};
    if (e == hipSuccess) e = hipMemcpy(out, dout, 8, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(&hs, dst, sizeof(hs), hipMemcpyDeviceToHost);
    (void)hipFree(dbuf); (void)hipFree(dout); (void)hipFree(dst);
    if (e != hipSuccess) return fnn::fail(FNN_EHIP, std::string("fnn_test_chain_sum: ") + hipGetErrorString(e));
    if (stats4) { stats4[0] = hs.runs; stats4[1] = hs.mixed; stats4[2] = hs.run_fail; stats4[3] = hs.thread_fail; }
    return FNN_OK;
}

int32_t fnn_stream_probe(int32_t device, int64_t bytes, int32_t reps, double* gbps_out) {
    if (bytes < (1 << 20) || reps < 1 || !gbps_out) return fnn::fail(FNN_EINVAL, "fnn_stream_probe: bad arguments");
    hipError_t e;
    if ((e = hipSetDevice(device)) != hipSuccess) return fnn::fail(FNN_EHIP, hipGetErrorString(e));
    void* p = nullptr;
    double* sink = nullptr;
    if ((e = hipMalloc(&p, (size_t)bytes)) != hipSuccess) return fnn::fail(FNN_ENOMEM, hipGetErrorString(e));
    if ((e = hipMalloc((void**)&sink, 8)) != hipSuccess) { (void)hipFree(p); return fnn::fail(FNN_ENOMEM, hipGetErrorString(e)); }
    (void)hipMemset(p, 0, (size_t)bytes);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    int64_t n16 = bytes / 16;
    dim3 g(256 * 8);
    hipLaunchKernelGGL(fnn::k_stream, g, dim3(256), 0, 0, (const double2*)p, n16, sink);  // warm-up
    (void)hipEventRecord(e0, 0);
    for (int i = 0; i < reps; i++) hipLaunchKernelGGL(fnn::k_stream, g, dim3(256), 0, 0, (const double2*)p, n16, sink);
    (void)hipEventRecord(e1, 0);
    e = hipEventSynchronize(e1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    (void)hipFree(p);
    (void)hipFree(sink);
    if (e != hipSuccess || ms <= 0.f) return fnn::fail(FNN_EHIP, "fnn_stream_probe: timing failed");
    *gbps_out = (double)bytes * reps / (ms * 1e-3) / 1e9;
    return FNN_OK;
}

}  // extern "C"
