// fastnn_host.hpp -- C++ host side above the C ABI (include/fastnn.h), mirroring the
// reference's own interface for the Canonical path so that a FastNN user finds the same
// names and argument meanings:
//
//   nnet::DistancesAndNames      Phylip reader          DistancesAndNames.java:12-153
//   nnet::NetMakerOriginal       abstract engine seam   NetMakerOriginal.java:17-162
//   nnet::NeighborNetCanonical   -mode Canonical        NeighborNetCanonical.java:29-36
//
// The Java toolchain is not available in the build image (no JDK), so the host side is
// C++; INTEGRATION.md shows the JNI stub a maintainer of the Java code base would add.
#ifndef FASTNN_HOST_HPP
#define FASTNN_HOST_HPP

#include <cstdint>
#include <cstdio>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/fastnn.h"

namespace nnet {

// DistancesAndNames.java:12-153 -- packed strict upper triangle + taxon names
class DistancesAndNames {
  public:
    std::vector<double> distances;  // (n*(n-1))/2, row-major upper triangle
    int nTaxa = 0;
    std::vector<std::string> names;

    // upperIndex (DistancesAndNames.java:24-38); 64-bit where the Java int would overflow
    int64_t upperIndex(int i, int j) const {
        if (i == j) return -1;
        int64_t a = i < j ? i : j, b = i < j ? j : i;
        return a * (nTaxa - 1) - a * (a - 1) / 2 + b - (a + 1);
    }
    // get(i, j) (:142-149): symmetric accessor, zero diagonal
    double get(int i, int j) const {
        int64_t u = upperIndex(i, j);
        return u < 0 ? 0.0 : distances[(size_t)u];
    }
    DistancesAndNames() = default;
    // ctor (:43-132): throws std::runtime_error where the Java would throw
    DistancesAndNames(const std::string& fileString, int numTaxa);
    // the dense matrix FastNN.main builds (FastNN.java:307-312)
    std::vector<double> toMatrix() const;
};

// header line: all whitespace removed, Integer.parseInt (FastNN.java:269-274)
int readTaxaCount(const std::string& fileName);

// Arrays.toString(int[]) as printed by `-order` (FastNN.java:394-397)
std::string orderingToString(const std::vector<int32_t>& ordering);

// NetMakerOriginal.java:17-162: the seam.  `d` is the dense row-major n x n matrix
// (`double[][] d`); numThreads / pool are accepted for signature compatibility.
class NetMakerOriginal {
  public:
    enum class NMMode { CANONICAL, RELAXED, RANDOM_N, RANDOM_NLOGN, RANDOM_LOGN, ORIGINAL };  // :19-21
    NetMakerOriginal(const double* d, int numTaxa, int numThreads, void* pool)
        : D(d), ntax(numTaxa), numThreads(numThreads), pool(pool) {}
    virtual ~NetMakerOriginal() = default;
    virtual std::vector<int32_t> runNeighborNet() = 0;  // :129
    const std::vector<int32_t>& getOrdering() const { return ordering; }  // :72-74

  protected:
    const double* D;
    const int ntax;
    const int numThreads;
    void* pool;
    std::vector<int32_t> ordering;
};

// -mode Canonical on the GPU engine.  Unlike the reference (NetMakerOriginal.java:653-656)
// the caller's matrix is not modified.
class NeighborNetCanonical : public NetMakerOriginal {
  public:
    NeighborNetCanonical(const double* d, int numTaxa, int numThreads = 1, void* pool = nullptr, int device = 0)
        : NetMakerOriginal(d, numTaxa, numThreads, pool), device(device) {}
    // from the reader's own container: the packed triangle goes to the device as it is
    // (fnn_set_packed_upper) instead of through the dense copy of FastNN.java:307-312
    explicit NeighborNetCanonical(const DistancesAndNames& dan, int numThreads = 1, void* pool = nullptr, int device = 0)
        : NetMakerOriginal(nullptr, dan.nTaxa, numThreads, pool), device(device), packed(&dan.distances) {}
    std::vector<int32_t> runNeighborNet() override {
        ordering.assign((size_t)ntax + 1, 0);
        fnn_opts o{};
        o.device = device;
        o.validate = 1;
        configure(o);
        int32_t rc;
        if (!packed) {
            rc = fnn_canonical_order_f64(D, ntax, ntax, &o, ordering.data(), &stats);
        } else if (ntax <= 3) {  // NetMakerOriginal.java:133-140
            for (int i = 0; i <= ntax; i++) ordering[(size_t)i] = i;
            rc = FNN_OK;
        } else {
            fnn_handle* h = nullptr;
            rc = fnn_create(ntax, &o, &h);
            if (rc == FNN_OK) rc = fnn_set_packed_upper(h, packed->data());
            if (rc == FNN_OK) rc = fnn_run(h, ordering.data(), &stats);
            const std::string why = rc == FNN_OK ? "" : fnn_last_error();
            if (h) fnn_destroy(h);
            if (rc != FNN_OK) throw std::runtime_error("fastnn: " + why);
        }
        if (rc != FNN_OK) throw std::runtime_error(std::string("fastnn: ") + fnn_last_error());
        return ordering;
    }
    fnn_stats stats{};

  protected:
    virtual void configure(fnn_opts&) const {}

  private:
    int device;
    const std::vector<double>* packed = nullptr;
};

// -mode Relaxed on the GPU engine: NeighborNetLocal.java (constructor :25-32) without the additivity check.  The
// reference draws from ThreadLocalRandom, which cannot be seeded; here the draws are java.util.Random(seed)'s, so a
// run can be repeated (seed: the caller's, e.g. the clock as in the line the reference commented out, :27).
class NeighborNetLocal : public NeighborNetCanonical {
  public:
    NeighborNetLocal(const double* d, int numTaxa, int numThreads, bool additive, void* pool, uint64_t seed, int device = 0)
        : NeighborNetCanonical(d, numTaxa, numThreads, pool, device), seed(seed) { check(additive); }
    NeighborNetLocal(const DistancesAndNames& dan, int numThreads, bool additive, void* pool, uint64_t seed, int device = 0)
        : NeighborNetCanonical(dan, numThreads, pool, device), seed(seed) { check(additive); }

  protected:
    void configure(fnn_opts& o) const override {
        o.mode = FNN_MODE_RELAXED;
        o.relaxed_seed_lo = (uint32_t)(seed & 0xFFFFFFFFu);
        o.relaxed_seed_hi = (uint32_t)(seed >> 32);
    }

  private:
    static void check(bool additive) {
        if (additive) throw std::invalid_argument("fastnn: the additivity check of the relaxed search (-additive) is not provided");
    }
    uint64_t seed;
};

// SplitAndWeight (CircularSplitWeights.java:47-50): the BitSet as the ascending list of its set bits
// (1-based taxon ids)
struct SplitAndWeight {
    std::vector<int32_t> split;
    double weight = 0.0;
};

// Double.toString semantics (shortest digits that round-trip; plain decimals for 1e-3 <= |d| < 1e7,
// otherwise d.dddE[-]x), as the reference's string concatenations print distances and weights
std::string javaDoubleToString(double d);

// the live path's split list (FastNN.java:405-419, :455-466) from the weights in live index order:
// split k = (i, j), 0 <= i < j <= n-1 = taxa ordering[i+1 .. j]; kept if weight > 1e-6
std::vector<SplitAndWeight> splitsFromWeights(const std::vector<int32_t>& ordering, const double* weights, int nTaxa);

// OutputPrinter.NexusWithSplitsAndDistances (OutputPrinter.java:8-96), written to `out`
void printNexusWithSplitsAndDistances(std::FILE* out, const std::vector<int32_t>& order, const DistancesAndNames& dan,
                                      const std::vector<SplitAndWeight>& splits);
// the same document from the weights in live index order (splitsFromWeights' rule), formatted by all host threads without
// materialising the splits' member lists; returns the number of splits
size_t printNexusFromWeights(std::FILE* out, const std::vector<int32_t>& order, const DistancesAndNames& dan, const double* weights);

}  // namespace nnet
#endif
