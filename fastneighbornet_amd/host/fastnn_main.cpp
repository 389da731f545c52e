// fastnn -- command line front end keeping FastNN's surface for the Canonical path
// (FastNN.java:110-397): -distFile <file> -mode <string> -threads <int> -mult <int>
// -order -additive -time -help, banner and progress lines on stderr, the circular order as
// Arrays.toString on stdout with -order.
//
// A run without -order continues as FastNN.java:401-540 does: non-negative least-squares weights of
// the circular splits (on the GPU, fnn_split_weights_f64: CircularSplitWeights.java's method) and
// the Nexus document on stdout (OutputPrinter.java).  -mode Relaxed (NeighborNetLocal.java) runs on the engine too, with
// -seed <long> for its generator (default: the clock, as in NeighborNetLocal.java:27); its -additive variant and the
// Random_* / Filter modes are not provided (a message and exit status 2 instead of silently doing something else).
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <iostream>
#include <memory>
#include <string>

#include "fastnn_host.hpp"

static void help() {
    // layout of commons-cli HelpFormatter.printHelp("FastNN", options) (FastNN.java:104-108)
    std::printf(
        "usage: FastNN\n"
        " -additive                  Performs an additivity check for the relaxed\n"
        "                            search strategy.\n"
        " -distFile <file_location>  The distance file in Phyllip format\n"
        " -help                      print this message\n"
        " -mode <string>             Determines the algorithm mode to run.  The\n"
        "                            options are: Canonical, Relaxed, Filter,\n"
        "                            Random_N, Random_NLOGN, Random_LOGN.  Default:\n"
        "                            Canonical\n"
        " -mult <integer>            For the random mode, this gives the constant\n"
        "                            multiplier that multiplies the search amount.\n"
        "                            Default: 5\n"
        " -order                     Outputs the circular order only.\n"
        " -threads <integer>         The number of threads to use.  Default: 1\n"
        " -time                      Show timing results.\n"
        " -gpu <integer>             (extension) HIP device ordinal.  Default: 0\n");
}

int main(int argc, char** argv) {
    std::fprintf(stderr, "FastNN Version: 0.3.5\n");
    std::fprintf(stderr, "Engine: fastnn-mi355x (HIP gfx950, C ABI %d)\n", fnn_abi_version());
    std::string fileName, modeStr = "CANONICAL";
    bool haveFile = false, order = false, timeMe = false, wantHelp = false, additive = false, haveSeed = false;
    unsigned long long seed = 0;
    int nThreads = 1, device = 0;
    for (int i = 1; i < argc; i++) {
        std::string a = argv[i];
        auto needArg = [&](const char* name) -> const char* {
            if (i + 1 >= argc) {
                std::fprintf(stderr, "Parsing failed.  Reason: Missing argument for option: %s\n", name);
                help();
                std::exit(0);
            }
            return argv[++i];
        };
        if (a == "-help" || a == "--help") wantHelp = true;
        else if (a == "-distFile" || a == "--distFile") { fileName = needArg("distFile"); haveFile = true; }
        else if (a == "-threads" || a == "--threads") nThreads = std::atoi(needArg("threads"));
        else if (a == "-mode" || a == "--mode") modeStr = needArg("mode");
        else if (a == "-mult" || a == "--mult") (void)needArg("mult");
        else if (a == "-gpu" || a == "--gpu") device = std::atoi(needArg("gpu"));
        else if (a == "-order" || a == "--order") order = true;
        else if (a == "-additive" || a == "--additive") additive = true;
        else if (a == "-seed" || a == "--seed") { seed = std::strtoull(needArg("seed"), nullptr, 10); haveSeed = true; }
        else if (a == "-time" || a == "--time") timeMe = true;
        else {
            std::fprintf(stderr, "Parsing failed.  Reason: Unrecognized option: %s\n", a.c_str());
            help();
            return 0;
        }
    }
    if (wantHelp) { help(); return 0; }
    if (!haveFile) {  // FastNN.java:235-238
        std::fprintf(stderr, "The program needs a distance file!!\n");
        help();
        return 0;
    }
    std::transform(modeStr.begin(), modeStr.end(), modeStr.begin(), [](unsigned char c) { return (char)std::toupper(c); });
    const char* known[] = {"CANONICAL", "RELAXED", "RANDOM_N", "RANDOM_NLOGN", "RANDOM_LOGN", "ORIGINAL"};
    if (std::find_if(std::begin(known), std::end(known), [&](const char* k) { return modeStr == k; }) == std::end(known)) {
        // NMMode.valueOf throws IllegalArgumentException (FastNN.java:263-265)
        std::fprintf(stderr, "Exception in thread \"main\" java.lang.IllegalArgumentException: No enum constant nnet.NetMakerOriginal.NMMode.%s\n",
                     modeStr.c_str());
        return 1;
    }
    int nTaxa = 0;
    try {
        nTaxa = nnet::readTaxaCount(fileName);
    } catch (const std::exception& e) {
        std::string msg = e.what();
        if (msg.rfind("FileNotFound", 0) == 0) {  // FastNN.java:280-285
            std::fprintf(stderr, "%s\n", msg.c_str());
            help();
            return 0;
        }
        std::fprintf(stderr, "Exception in thread \"main\" %s\n", msg.c_str());
        return 1;
    }
    std::fprintf(stderr, "Calculating a tree for %d taxa using %d thread(s).\n", nTaxa, nThreads);
    std::fprintf(stderr, "Getting distances from the file: %s\n", fileName.c_str());
    const bool relaxed = modeStr == "RELAXED";
    if ((modeStr != "CANONICAL" && modeStr != "ORIGINAL" && !relaxed) || (relaxed && additive)) {
        std::fprintf(stderr, "fastnn-mi355x: -mode %s%s is not provided by this engine (Canonical and Relaxed are).\n", modeStr.c_str(),
                     relaxed ? " -additive" : "");
        return 2;
    }
    try {
        nnet::DistancesAndNames danOrg(fileName, nTaxa);
        std::unique_ptr<nnet::NeighborNetCanonical> myNMO;
        if (relaxed) {  // FastNN.java:329-338
            if (!haveSeed) seed = (unsigned long long)std::chrono::duration_cast<std::chrono::milliseconds>(
                                      std::chrono::system_clock::now().time_since_epoch()).count();
            myNMO.reset(new nnet::NeighborNetLocal(danOrg, nThreads, false, nullptr, seed, device));
            std::fprintf(stderr, "Using the relaxed version without additivity checking.\n");
            std::fprintf(stderr, "Relaxed search seed (java.util.Random): %llu\n", seed);
        } else {
            myNMO.reset(new nnet::NeighborNetCanonical(danOrg, nThreads, nullptr, device));
            std::fprintf(stderr, "Using the canonical implementation.\n");
        }
        auto t0 = std::chrono::steady_clock::now();
        std::vector<int32_t> ordering = myNMO->runNeighborNet();
        auto t1 = std::chrono::steady_clock::now();
        if (timeMe) std::fprintf(stderr, "Got the order in (s): %.9g\n", std::chrono::duration<double>(t1 - t0).count());
        if (order) {
            std::printf("%s\n", nnet::orderingToString(ordering).c_str());
            return 0;
        }
        // FastNN.java:398-491: split weights, then the Nexus document (the reference parses the file a
        // second time for the names and distances, FastNN.java:438)
        t0 = std::chrono::steady_clock::now();
        std::vector<double> weights((size_t)nTaxa * (size_t)(nTaxa - 1) / 2);
        fnn_sw_stats sw{};
        std::vector<double> D = danOrg.toMatrix();
        if (nTaxa >= 2) {
            int32_t rc = fnn_split_weights_f64(D.data(), nTaxa, nTaxa, ordering.data(), device, weights.data(), &sw);
            if (rc == FNN_EINEXACT)  // the weights are there; say what they are worth and go on (the reference has no such check)
                std::fprintf(stderr, "fastnn: warning: %s\n", fnn_last_error());
            else if (rc != FNN_OK) throw std::runtime_error(std::string("fastnn: ") + fnn_last_error());
        }
        const nnet::DistancesAndNames dan(fileName, nTaxa);
        t1 = std::chrono::steady_clock::now();
        if (timeMe) std::fprintf(stderr, "Got the splits and weights in (s): %.9g\n", std::chrono::duration<double>(t1 - t0).count());
        t0 = std::chrono::steady_clock::now();
        (void)nnet::printNexusFromWeights(stdout, ordering, dan, weights.data());  // (splitsFromWeights + printNexusWithSplitsAndDistances, in one parallel pass)
        t1 = std::chrono::steady_clock::now();
        if (timeMe) std::fprintf(stderr, "Wrote the output in (s): %.9g\n", std::chrono::duration<double>(t1 - t0).count());
        return 0;
    } catch (const std::exception& e) {
        std::fprintf(stderr, "Exception in thread \"main\" %s\n", e.what());
        return 1;
    }
}
