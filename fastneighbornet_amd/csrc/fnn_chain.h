// fnn_chain.h -- exact parallel evaluation of a SEQUENTIAL fp64 sum.
//
// The reference accumulates three quantities with a scalar loop in node-position
// order: the initial row sums (NetMakerOriginal.java:181-186), ComputeRx (:551-560)
// and the new cluster's Sx (:532).  fp64 addition is not associative, so a tree
// reduction gives different last bits and, through the Q criterion, eventually a
// different circular order.  A literal one-lane loop over m = 32768 terms costs
// ~0.2 ms per sum on the GPU (profiles/r01/v1_*: 36 % of the whole run).
//
// Observation.  While the running sum s stays inside one binade [2^E, 2^(E+1)) and the
// addends are non-negative, s is an integer multiple S of u = ulp(s) and
//     fl(s + a) = (S + q + r) * u,   a = q*u + f, 0 <= f < u,
//     r = 0 if f < u/2, 1 if f > u/2, and on a tie (f == u/2) r makes S + q + r even.
// So each addend acts on (S mod 2) -> increment as a tiny automaton that only looks at
// the parity of S, and automata compose ASSOCIATIVELY:
//     (A then B)(b) = A(b) + B((b + A(b)) mod 2).
// A run of addends can therefore be reduced in any tree order to a pair of integers
// (increment if S enters even, increment if S enters odd) and applied to the exact s in
// O(1) -- provided the binade assumption holds, which is checked exactly when the pair is
// applied: s must lie in binade E at the start of the run and S + increment must stay
// below 2^53 (monotonicity then covers every intermediate partial sum).  Where the check
// fails, and for every addend that is negative, non-finite or crosses a binade, the
// ordinary fp64 addition is executed instead.  The binade each addend will see is only
// PREDICTED (from an approximate parallel prefix sum); exactness never depends on the
// prediction, only speed does.
//
// Representation.  All integers of the scheme are < 2^53 when the run is valid, so they are
// carried as doubles in units of u: scaling by the power of two 1/u is exact, floor / fract /
// sums of such integers are exact, and a sum that leaves the exact range can only come out
// >= 2^53 (fp addition of non-negative terms is monotone), which the validity check rejects.
// Every operation below is a single correctly rounded IEEE operation whose result is exact
// in the cases that are accepted; nothing depends on FMA contraction being on or off.
#ifndef FNN_CHAIN_H
#define FNN_CHAIN_H

#include <stdint.h>

#include "fnn_core.h"

namespace fnn {

struct Mono {
    double i0, i1;  // increment of S (in ulps of the run's binade) for S entering even / odd
};

constexpr int CH_E_MIN = 128, CH_E_MAX = 1900;  // biased binades served by the integer path
constexpr int CH_GUARD_BITS = 1;  // non-zero: keep the 2^-20 guard band around powers of two
constexpr double CH_TWO52 = 4503599627370496.0;
constexpr double CH_TWO53 = 9007199254740992.0;

FNN_HD uint64_t f2u(double x) { return __builtin_bit_cast(uint64_t, x); }
FNN_HD double u2f(uint64_t x) { return __builtin_bit_cast(double, x); }
FNN_HD uint32_t hi32(double x) { return (uint32_t)(f2u(x) >> 32); }

// exact power of two 2^k, k in [-1022, 1023]
FNN_HD double pow2i(int k) { return u2f((uint64_t)(k + 1023) << 52); }
// 1 / ulp of binade E (biased): ulp = 2^(E - 1075)
FNN_HD double inv_ulp(int32_t E) { return pow2i(1075 - E); }
FNN_HD double ulp_of(int32_t E) { return pow2i(E - 1075); }

FNN_HD double ch_floor(double x) { return __builtin_floor(x); }
// x is a non-negative integer < 2^53 held in a double: is it odd?
FNN_HD bool ch_odd(double x) {
    const double h = x * 0.5;
    return h != ch_floor(h);
}

FNN_HD Mono mono_identity() {
    Mono m;
    m.i0 = 0.0;
    m.i1 = 0.0;
    return m;
}

// A then B
FNN_HD Mono mono_compose(Mono A, Mono B) {
    Mono r;
    r.i0 = A.i0 + (ch_odd(A.i0) ? B.i1 : B.i0);
    r.i1 = A.i1 + (ch_odd(A.i1) ? B.i0 : B.i1);  // entering odd: parity after A is odd + A.i1
    return r;
}

// Binade prediction from the approximate partial sums in front of (A0) and behind (A1) an
// addend: true and E when both lie in the same served binade and neither is within 2^-20
// (relative) of a power of two.  Uses the high words only.
FNN_HD bool chain_predict(double A0, double A1, bool guard, int32_t& E) {
    const uint32_t h0 = hi32(A0), h1 = hi32(A1);
    const uint32_t e0 = h0 >> 20, e1 = h1 >> 20;  // sign bit included: negative -> e >= 2048
    if (e0 != e1 || e0 < (uint32_t)CH_E_MIN || e0 > (uint32_t)CH_E_MAX) return false;
    if (guard) {
        const uint32_t f0 = h0 & 0xFFFFFu, f1 = h1 & 0xFFFFFu;
        if (f0 == 0u || f0 == 0xFFFFFu || f1 == 0u || f1 == 0xFFFFFu) return false;
    }
    E = (int32_t)e0;
    return true;
}

// Automaton of addend a relative to binade E (invu = inv_ulp(E)).  False for addends that
// must go through the ordinary addition (negative, -0.0, NaN, too large for the binade).
FNN_HD bool chain_automaton(double a, double invu, Mono& mo) {
    if (!(a >= 0.0) || (f2u(a) >> 63)) return false;  // NaN, negative, -0.0
    const double t = a * invu;                         // a in ulps; exact (see header)
    if (!(t < CH_TWO53)) return false;                 // a >= 2^(E+1) (or inf): crosses the binade
    const double q = ch_floor(t);
    const double f = t - q;                            // exact
    double up0, up1;
    if (f < 0.5) { up0 = 0.0; up1 = 0.0; }
    else if (f > 0.5) { up0 = 1.0; up1 = 1.0; }
    else {  // tie: S + q + r must be even
        const bool qodd = ch_odd(q);
        up0 = qodd ? 1.0 : 0.0;
        up1 = qodd ? 0.0 : 1.0;
    }
    mo.i0 = q + up0;
    mo.i1 = q + up1;
    return true;
}

// mt := mt then automaton(a), with the common case (no tie: the addend's automaton is the
// constant q + r) reduced to two additions.  False when a must go through ordinary addition.
FNN_HD bool chain_accumulate(double a, double invu, Mono& mt) {
    const double t = a * invu;                   // a in ulps; exact (see header); -0.0 counts as 0
    if (!(t >= 0.0 && t < CH_TWO53)) return false;  // NaN, negative, or a >= 2^(E+1)
    const double q = ch_floor(t);
    const double f = t - q;                      // exact
    if (f == 0.5) {                              // tie: S + q + r must be even
        Mono mo;
        const bool qodd = ch_odd(q);
        mo.i0 = q + (qodd ? 1.0 : 0.0);
        mo.i1 = q + (qodd ? 0.0 : 1.0);
        mt = mono_compose(mt, mo);
    } else {
        const double c = q + (f > 0.5 ? 1.0 : 0.0);
        mt.i0 += c;
        mt.i1 += c;
    }
    return true;
}

// Apply a composed automaton to the exact running sum.  False (s untouched) when the
// binade assumption does not hold.
FNN_HD bool mono_apply(double& s, int32_t E, Mono mo) {
    if (E < CH_E_MIN || E > CH_E_MAX) return false;
    const double S = s * inv_ulp(E);  // exact scaling; in [2^52, 2^53) iff s is in binade E
    if (!(S >= CH_TWO52 && S < CH_TWO53)) return false;
    const double S2 = S + (ch_odd(S) ? mo.i1 : mo.i0);
    if (!(S2 < CH_TWO53)) return false;
    s = S2 * ulp_of(E);               // exact
    return true;
}

// The automaton of a whole chunk of N addends relative to one binade, with the addends handled
// independently of each other (no dependency from one addend to the next: the per-addend work
// pipelines).  Without a tie every addend's automaton is the constant q + r, and constants compose
// by addition; the increments are integer-valued doubles, so their sum is exact in ANY order while
// it stays below 2^53, and a sum that leaves that range can only come out >= 2^53 (each partial sum
// is either exact or the rounding of an exact value >= 2^53), which mono_apply rejects.  False: some
// addend needs the ordinary addition, or there is a tie (an addend ending exactly half an ulp above a
// multiple of the ulp; with k bits of the addend below the ulp of the sum that has probability 2^-k,
// so it happens in the first chunks of a sum, which cross binades anyway, and ~1/j-th of the time in
// chunk j): the chunk's addends are then added one by one.
template <int N>
FNN_HD bool chain_chunk(const double (&a)[N], double invu, Mono& mt) {
    double acc[4] = {0.0, 0.0, 0.0, 0.0};  // four independent accumulators
    double tmin = 0.0;
    bool tie = false;
#pragma unroll
    for (int i = 0; i < N; i++) {
        const double t = a[i] * invu;                 // a in ulps; exact (see header); -0.0 counts as 0
        const double c = __builtin_rint(t);           // q + r unless there is a tie (round to nearest even)
        const double e = t - c;                       // exact, |e| <= 0.5
        tie = tie || (__builtin_fabs(e) == 0.5);
        acc[i & 3] += c;
        tmin = __builtin_fmin(tmin, t);               // (fmin drops a NaN: caught by the sum below)
    }
    const double c = (acc[0] + acc[1]) + (acc[2] + acc[3]);
    // NaN addend -> c is NaN; negative addend -> tmin < 0; an addend >= 2^(E+1) (or inf) makes the sum
    // >= 2^53, which no run accepts
    if (!(c < CH_TWO53) || tmin < 0.0 || tie) return false;
    mt.i0 = c;
    mt.i1 = c;
    return true;
}

// Increment of an automaton as an integer for mono_apply_bits; anything outside the exact range
// becomes 2^53, which mono_apply_bits always rejects.
FNN_HD uint64_t mono_inc_bits(double inc) {
    return (inc >= 0.0 && inc < CH_TWO53) ? (uint64_t)inc : (1ULL << 53);
}

// mono_apply on the bit pattern of s: with s in binade E, S = s / ulp is 2^52 + mantissa, so its
// parity is bit 0 of the pattern, S + increment is an integer addition on the pattern, and the sum
// stays below 2^53 exactly when no carry reaches the exponent field.  Same accept / reject decisions
// and the same result as mono_apply; a handful of integer operations instead of a chain of dependent
// fp64 ones.
FNN_HD bool mono_apply_pattern(uint64_t& sb, int32_t E, uint64_t i0, uint64_t i1) {
    if (E < CH_E_MIN || E > CH_E_MAX) return false;
    if ((int64_t)(sb >> 52) != (int64_t)E) return false;       // positive and in binade E
    const uint64_t inc = (sb & 1ULL) ? i1 : i0;
    if (inc > (1ULL << 52)) return false;
    const uint64_t nb = sb + inc;
    if ((int64_t)(nb >> 52) != (int64_t)E) return false;       // S + increment >= 2^53
    sb = nb;
    return true;
}
FNN_HD bool mono_apply_bits(double& s, int32_t E, uint64_t i0, uint64_t i1) {
    uint64_t sb = f2u(s);
    if (!mono_apply_pattern(sb, E, i0, i1)) return false;
    s = u2f(sb);
    return true;
}

// ---------------------------------------------------------------------------
// Second form of the block-parallel sum ("records"): almost every chunk that the first form adds one by one
// holds exactly ONE addend that needs an ordinary addition - the one that carries the running sum into the next
// binade, or one whose fraction is exactly half an ulp (a tie) - and constants on either side of it.  A thread
// therefore describes its chunk as
//     CONST  (E, c)                         every addend is a constant increment c_i = rint(a_i / ulp_E)
//     SPLIT  (E0, c0) a* (E1, c1)           constants before and after ONE special addend a*
//     SERIAL                                anything else (several specials, negative / non-finite addends, ...)
// and the sequence of all threads' pieces is evaluated as: merge neighbouring constant pieces of equal binade
// (integer sums, any order), apply each merged piece to the bit pattern of the running sum (mono_apply_pattern: the
// SAME verified step as a run of the first form - binade of the sum before, no carry into the exponent after), add
// a* with an ordinary addition, add SERIAL chunks one by one.  Every constant piece is verified when applied, so
// exactness never depends on the predictions (where a special addend sits, which binade follows it); a failed
// verification makes the caller fall back to the first form for this sum.
// ---------------------------------------------------------------------------
constexpr int CHR_CONST = 0, CHR_SPLIT = 1, CHR_SERIAL = 2, CHR_EMPTY = 3;
struct ChRec {
    int32_t kind, E0, E1;   // E0 < 0 on a SPLIT: no constant piece in front of the special addend
    uint64_t c0, c1;
    double sp;
};

// constants of a[lo, hi) relative to binade E: their sum; false on a tie, a negative / non-finite addend or an
// addend too large for the binade.  `tie_at` (may be null) receives the index of the first tie and the scan goes on.
template <int N>
FNN_HD bool chain_consts(const double (&a)[N], int lo, int hi, int32_t E, uint64_t& c, int* ntie, int* tie_at) {
    const double invu = inv_ulp(E);
    double acc0 = 0.0, acc1 = 0.0;
    uint32_t sgn = 0;  // sign bits of the scaled addends, OR-ed (a negative addend - or -0.0, harmless - sets it)
    int ties = 0, first = -1;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int i = 0; i < N; i++) {
        if (i < lo || i >= hi) continue;
        const double t = a[i] * invu;                 // a in ulps; exact (see header)
        const double ci = __builtin_rint(t);          // q + r unless there is a tie (round to nearest even)
        const double e = t - ci;                      // exact, |e| <= 0.5
        if (__builtin_fabs(e) == 0.5) { if (first < 0) first = i; ties++; }
        else if (i & 1) acc1 += ci;
        else acc0 += ci;
        sgn |= hi32(t);
    }
    const double cs = acc0 + acc1;
    const double tmin = (sgn >> 31) ? -1.0 : 0.0;
    if (ntie) *ntie = ties;
    if (tie_at) *tie_at = first;
    c = mono_inc_bits(cs);
    return cs < CH_TWO53 && !(tmin < 0.0) && cs == cs;
}

// a[j] for a run-time j without indexing the register array (unrolled selects)
template <int N>
FNN_HD double chain_pick(const double (&a)[N], int j) {
    double v = 0.0;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int i = 0; i < N; i++) if (i == j) v = a[i];
    return v;
}

// the record of one thread's chunk: a[0, cnt) are its addends, A0 the PREDICTED partial sum in front of the chunk
template <int N>
FNN_HD ChRec chain_thread_record(const double (&a)[N], int cnt, double A0, double loc, bool guard) {
    ChRec r;
    r.kind = CHR_SERIAL; r.E0 = r.E1 = -1; r.c0 = r.c1 = 0; r.sp = 0.0;
    if (cnt <= 0) { r.kind = CHR_EMPTY; return r; }
    int32_t E = -1;
    if (chain_predict(A0, A0 + loc, guard, E)) {
        int nt = 0, jt = -1;
        uint64_t c = 0;
        if (!chain_consts(a, 0, cnt, E, c, &nt, &jt)) return r;  // (negative / non-finite / too large: one by one)
        if (nt == 0) { r.kind = CHR_CONST; r.E0 = E; r.c0 = c; return r; }
        if (nt > 1) return r;
        // exactly one tie: constants before and after it, in the same binade
        uint64_t c0 = 0, c1 = 0;
        if (!chain_consts(a, 0, jt, E, c0, nullptr, nullptr) || !chain_consts(a, jt + 1, cnt, E, c1, nullptr, nullptr)) return r;
        r.kind = CHR_SPLIT; r.E0 = E; r.c0 = c0; r.sp = chain_pick(a, jt); r.E1 = E; r.c1 = c1;
        return r;
    }
    // the predicted partial sums of the chunk do not stay in one binade: find the addend that leaves it
    double p = A0, pafter = A0, asp = 0.0;
    int js = -1;
    int32_t E0 = -1;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int i = 0; i < N; i++) {  // (unrolled with a predicate: a run-time index would put the addends in scratch memory)
        if (js < 0 && i < cnt) {
            int32_t Ei = -1;
            const double pn = p + a[i];
            if (!chain_predict(p, pn, guard, Ei) || (i > 0 && Ei != E0)) { js = i; pafter = pn; asp = a[i]; }
            else { E0 = Ei; p = pn; }
        }
    }
    if (js < 0) return r;  // (cannot happen: the whole-chunk prediction failed)
    int32_t E1 = -1;
    if (js + 1 < cnt && !chain_predict(pafter, A0 + loc, guard, E1)) return r;  // the rest leaves its binade again
    uint64_t c0 = 0, c1 = 0;
    int nt = 0;
    if (js > 0) { if (!chain_consts(a, 0, js, E0, c0, &nt, nullptr) || nt) return r; }
    if (js + 1 < cnt) { if (!chain_consts(a, js + 1, cnt, E1, c1, &nt, nullptr) || nt) return r; }
    r.kind = CHR_SPLIT; r.E0 = js > 0 ? E0 : -1; r.c0 = c0; r.sp = asp; r.E1 = js + 1 < cnt ? E1 : -1; r.c1 = c1;
    return r;
}

// serial evaluation of a sequence of records (CPU model; the GPU walker does the same with a segmented scan in front).
// `serial(t, s)` adds thread t's addends one by one.  False: a verification failed (the caller falls back).
template <class Serial>
FNN_HD bool chain_walk_records(const ChRec* rec, int nthreads, double& s, int32_t* applied, Serial&& serial) {
    bool have = false;
    int32_t E = -1;
    uint64_t c = 0;
    auto flush = [&]() {
        if (have && c != 0) {
            if (!mono_apply_bits(s, E, c, c)) return false;
            if (applied) ++*applied;
        }
        have = false; c = 0;
        return true;
    };
    auto piece = [&](int32_t Ep, uint64_t cp) {
        if (have && E == Ep) { c = c + cp > (1ULL << 60) ? (1ULL << 60) : c + cp; return true; }  // (anything beyond 2^52 is rejected when applied)
        if (!flush()) return false;
        have = true; E = Ep; c = cp;
        return true;
    };
    for (int t = 0; t < nthreads; t++) {
        const ChRec& r = rec[t];
        if (r.kind == CHR_EMPTY) continue;
        if (r.kind == CHR_CONST) { if (!piece(r.E0, r.c0)) return false; }
        else if (r.kind == CHR_SPLIT) {
            if (r.E0 >= 0 && !piece(r.E0, r.c0)) return false;
            if (!flush()) return false;
            s += r.sp;
            if (r.E1 >= 0 && !piece(r.E1, r.c1)) return false;
        } else {
            if (!flush()) return false;
            serial(t, s);
        }
    }
    return flush();
}

// Counters of the walker (diagnostics / tests)
struct ChainStats {
    int32_t runs;         // composed runs applied
    int32_t mixed;        // thread chunks added one by one (binade crossings etc.)
    int32_t run_fail;     // composed run rejected by mono_apply -> per-thread retry
    int32_t thread_fail;  // single-thread automaton rejected -> its addends added one by one
};

}  // namespace fnn
#endif
