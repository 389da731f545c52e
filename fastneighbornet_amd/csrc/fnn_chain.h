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
// applied: s must have exponent E at the start of the run and S + increment must stay
// below 2^53 (monotonicity then covers every intermediate partial sum).  Where the check
// fails, and for every addend that is negative, non-finite or crosses a binade, the
// ordinary fp64 addition is executed instead.  The binade each addend will see is only
// PREDICTED (from an approximate parallel prefix sum); exactness never depends on the
// prediction, only speed does.
#ifndef FNN_CHAIN_H
#define FNN_CHAIN_H

#include <stdint.h>

#include "fnn_core.h"

namespace fnn {

struct Mono {
    uint64_t i0, i1;  // increment of S (in ulps of the run's binade) for S entering even / odd
};

constexpr uint64_t CH_MANT = (1ULL << 52) - 1;
constexpr uint64_t CH_IMPL = 1ULL << 52;
constexpr uint64_t CH_SAT = 1ULL << 62;  // saturation keeps mispredicted runs from wrapping
constexpr int CH_GUARD_BITS = 22;        // predicted prefix must be 2^-30 (relative) away from a power of two

FNN_HD uint64_t f2u(double x) { return __builtin_bit_cast(uint64_t, x); }
FNN_HD double u2f(uint64_t x) { return __builtin_bit_cast(double, x); }

FNN_HD uint64_t sat_add(uint64_t a, uint64_t b) {  // a, b <= CH_SAT
    uint64_t c = a + b;
    return c > CH_SAT ? CH_SAT : c;
}

FNN_HD Mono mono_identity() {
    Mono m;
    m.i0 = 0;
    m.i1 = 0;
    return m;
}

// branch-free select (also keeps hipcc from forming a scalar select on a vector compare)
FNN_HD uint64_t sel_by_parity(uint64_t parity_src, uint64_t even_v, uint64_t odd_v) {
    uint64_t mask = 0 - (parity_src & 1ULL);
    return (even_v & ~mask) | (odd_v & mask);
}

// A then B
FNN_HD Mono mono_compose(Mono A, Mono B) {
    Mono r;
    r.i0 = sat_add(A.i0, sel_by_parity(A.i0, B.i0, B.i1));
    r.i1 = sat_add(A.i1, sel_by_parity(A.i1 + 1, B.i0, B.i1));
    return r;
}

// Addend a, with the PREDICTED partial sums before (A0) and after (A1) it.  Returns true
// and the addend's automaton relative to the binade E (biased exponent of A0) when the
// addend can be treated on the integer path.
FNN_HD bool chain_classify(double a, double A0, double A1, int guard_bits, int32_t& E, Mono& mo) {
    const uint64_t ua = f2u(a), u0 = f2u(A0), u1 = f2u(A1);
    if ((ua | u0 | u1) >> 63) return false;  // negative (or -0.0) anywhere: ordinary addition
    const uint32_t eab = (uint32_t)(ua >> 52) & 0x7FF;
    const uint32_t e0 = (uint32_t)(u0 >> 52) & 0x7FF, e1 = (uint32_t)(u1 >> 52) & 0x7FF;
    if (eab == 0x7FF || e0 == 0 || e0 == 0x7FF || e1 != e0) return false;
    if (guard_bits > 0) {
        const uint64_t g = 1ULL << guard_bits;
        const uint64_t f0 = u0 & CH_MANT, f1 = u1 & CH_MANT;
        if (f0 < g || f0 > CH_MANT - g || f1 < g || f1 > CH_MANT - g) return false;
    }
    const uint32_t ea = eab ? eab : 1;                      // subnormals share the exponent of DBL_MIN
    const uint64_t Ma = (ua & CH_MANT) | (eab ? CH_IMPL : 0);  // a = Ma * 2^(ea - 1075)
    const int shift = (int)e0 - (int)ea;                    // a / u = Ma / 2^shift
    if (shift < 0) return false;
    uint64_t q, up0, up1;
    if (shift == 0) {
        q = Ma; up0 = up1 = 0;
    } else if (shift >= 64) {
        q = 0; up0 = up1 = 0;  // a < u / 2
    } else {
        q = Ma >> shift;
        const uint64_t rem = Ma & ((1ULL << shift) - 1), half = 1ULL << (shift - 1);
        if (rem < half) up0 = up1 = 0;
        else if (rem > half) up0 = up1 = 1;
        else { up0 = q & 1; up1 = (q & 1) ^ 1; }  // tie: S + q + r must be even
    }
    E = (int32_t)e0;
    mo.i0 = q + up0;
    mo.i1 = q + up1;
    return true;
}

// Apply a composed automaton to the exact running sum.  False (s untouched) when the
// binade assumption does not hold.
FNN_HD bool mono_apply(double& s, int32_t E, Mono mo) {
    const uint64_t us = f2u(s);
    if ((us >> 52) != (uint64_t)E) return false;  // sign bit clear and biased exponent == E
    const uint64_t S = (us & CH_MANT) | CH_IMPL;
    const uint64_t S2 = S + sel_by_parity(S, mo.i0, mo.i1);
    if (S2 >= (1ULL << 53)) return false;
    s = u2f(((uint64_t)E << 52) | (S2 & CH_MANT));
    return true;
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
