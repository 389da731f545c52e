// fastnn_host.cpp -- Phylip reader and output formatting (no GPU code).
// Follows DistancesAndNames.java:43-132 including its quirks: the first line is skipped,
// lines are split on single spaces, the first token is the name, remaining non-blank tokens
// are further split on tabs, only the first `row` values of a line are used (so square and
// lower-triangular files both work), the token buffer is NOT cleared between lines, reading
// stops at EOF or at the first line beyond numTaxa rows.
#include "fastnn_host.hpp"

#include <charconv>
#include <algorithm>
#include <vector>
#include <cstring>
#include <string>
#include <thread>
#include <cmath>
#include <cstdlib>
#include <limits>

#include <cctype>
#include <cerrno>
#include <cstdlib>
#include <fstream>
#include <sstream>

namespace nnet {

static std::vector<std::string> javaSplit(const std::string& s, char sep) {
    // String.split(String) for a one-character separator: trailing empty strings are dropped
    std::vector<std::string> out;
    size_t start = 0;
    while (true) {
        size_t p = s.find(sep, start);
        if (p == std::string::npos) { out.push_back(s.substr(start)); break; }
        out.push_back(s.substr(start, p - start));
        start = p + 1;
    }
    while (!out.empty() && out.back().empty()) out.pop_back();
    if (out.empty() && s.empty()) out.push_back("");  // "".split(" ") -> [""]
    return out;
}

static bool blankAfterTrim(const std::string& s) {
    for (unsigned char c : s)
        if (c > ' ') return false;  // String.trim() strips code points <= U+0020
    return true;
}

static double javaDoubleValueOf(const std::string& tok) {
    // Double.valueOf: surrounding whitespace is trimmed; the rest must be a complete literal
    size_t b = 0, e = tok.size();
    while (b < e && (unsigned char)tok[b] <= ' ') b++;
    while (e > b && (unsigned char)tok[e - 1] <= ' ') e--;
    std::string t = tok.substr(b, e - b);
    if (t.empty()) throw std::runtime_error("NumberFormatException: empty String");
    if (!t.empty() && (t.back() == 'd' || t.back() == 'D' || t.back() == 'f' || t.back() == 'F') &&
        t.find("0x") == std::string::npos && t.find("0X") == std::string::npos && t != "Infinity" &&
        t != "+Infinity" && t != "-Infinity")
        t.pop_back();
    if (t == "NaN" || t == "+NaN" || t == "-NaN") return std::strtod("nan", nullptr);
    if (t == "Infinity" || t == "+Infinity") return std::strtod("inf", nullptr);
    if (t == "-Infinity") return -std::strtod("inf", nullptr);
    for (char c : t)  // strtod would accept "inf"/"nan"/"infinity" spellings Java rejects
        if (std::isalpha((unsigned char)c) && c != 'e' && c != 'E' && c != 'x' && c != 'X' && c != 'p' && c != 'P' &&
            !std::isxdigit((unsigned char)c))
            throw std::runtime_error("NumberFormatException: For input string: \"" + tok + "\"");
    char* end = nullptr;
    errno = 0;
    double v = std::strtod(t.c_str(), &end);
    if (end == t.c_str() || *end != '\0') throw std::runtime_error("NumberFormatException: For input string: \"" + tok + "\"");
    return v;
}

DistancesAndNames::DistancesAndNames(const std::string& fileString, int numTaxa) : nTaxa(numTaxa) {
    const int64_t npairs = ((int64_t)numTaxa * (numTaxa - 1)) / 2;
    distances.assign((size_t)(npairs > 0 ? npairs : 0), 0.0);
    names.assign((size_t)(numTaxa > 0 ? numTaxa : 0), std::string());
    std::ifstream in(fileString);
    if (!in) {  // the Java prints "IOException: ..." and carries on with zeros (:112-114)
        std::fprintf(stderr, "IOException: %s\n", fileString.c_str());
        return;
    }
    std::string line;
    std::getline(in, line);  // header line, parsed by the caller
    int row = 0;
    std::vector<std::string> copy((size_t)(numTaxa > 0 ? numTaxa : 0));
    std::vector<char> filled(copy.size(), 0);
    while (std::getline(in, line)) {
        if (!line.empty() && line.back() == '\r') line.pop_back();  // BufferedReader.readLine strips \r\n
        std::vector<std::string> ss = javaSplit(line, ' ');
        if (row >= numTaxa) break;  // names[row] -> ArrayIndexOutOfBounds -> break (:67-71)
        names[(size_t)row] = ss.empty() ? std::string() : ss[0];
        size_t innerCount = 0;
        for (size_t i = 1; i < ss.size(); i++) {
            if (!blankAfterTrim(ss[i])) {
                std::vector<std::string> temp = javaSplit(ss[i], '\t');
                for (const std::string& t : temp) {
                    if (innerCount >= copy.size())
                        throw std::runtime_error("ArrayIndexOutOfBoundsException: more than " + std::to_string(numTaxa) +
                                                 " values on line " + std::to_string(row + 2));
                    copy[innerCount] = t;
                    filled[innerCount] = 1;
                    innerCount++;
                }
            }
        }
        for (int column = 0; column < row; column++) {
            if (!filled[(size_t)column])
                throw std::runtime_error("NullPointerException: line " + std::to_string(row + 2) + " has fewer than " +
                                         std::to_string(row) + " values");
            distances[(size_t)upperIndex(row, column)] = javaDoubleValueOf(copy[(size_t)column]);
        }
        row++;
    }
}

std::vector<double> DistancesAndNames::toMatrix() const {
    std::vector<double> D((size_t)nTaxa * (size_t)nTaxa, 0.0);
    for (int i = 0; i < nTaxa; i++)
        for (int j = 0; j < nTaxa; j++) D[(size_t)i * nTaxa + j] = get(i, j);
    return D;
}

int readTaxaCount(const std::string& fileName) {
    std::ifstream in(fileName);
    if (!in) throw std::runtime_error("FileNotFound: " + fileName + " (No such file or directory)");
    std::string data;
    std::getline(in, data);
    std::string t;
    for (unsigned char c : data)
        if (!std::isspace(c)) t.push_back((char)c);  // replaceAll("\\s", "")
    if (t.empty()) throw std::runtime_error("NumberFormatException: For input string: \"\"");
    size_t i = (t[0] == '-' || t[0] == '+') ? 1 : 0;
    if (i >= t.size()) throw std::runtime_error("NumberFormatException: For input string: \"" + t + "\"");
    for (size_t k = i; k < t.size(); k++)
        if (!std::isdigit((unsigned char)t[k])) throw std::runtime_error("NumberFormatException: For input string: \"" + t + "\"");
    long v = std::strtol(t.c_str(), nullptr, 10);
    if (v > 2147483647L || v < -2147483648L) throw std::runtime_error("NumberFormatException: For input string: \"" + t + "\"");
    return (int)v;
}

std::string orderingToString(const std::vector<int32_t>& o) {
    std::ostringstream ss;
    ss << "[";
    for (size_t i = 0; i < o.size(); i++) {
        if (i) ss << ", ";
        ss << o[i];
    }
    ss << "]";
    return ss.str();
}

}  // namespace nnet

// small C surface so the Python tests can exercise the reader without a GPU
namespace nnet {

// Double.toString (the shortest decimal that reads back as the same double - java.lang.Double's contract, JDK 19+'s
// algorithm; plain notation for 1e-3 <= |d| < 1e7, else d.dddE<x>) into out (>= 32 bytes); returns the length.
// std::to_chars yields the shortest round-trip digits in one pass (the first version searched the precision with
// snprintf + strtod: 2.8 us per number, 50 minutes for the distance block of a 32768-taxon document).
int javaDoubleToChars(double d, char* out) {
    if (d != d) { std::memcpy(out, "NaN", 3); return 3; }
    if (d == std::numeric_limits<double>::infinity()) { std::memcpy(out, "Infinity", 8); return 8; }
    if (d == -std::numeric_limits<double>::infinity()) { std::memcpy(out, "-Infinity", 9); return 9; }
    if (d == 0.0) { const char* z = std::signbit(d) ? "-0.0" : "0.0"; const int l = (int)std::strlen(z); std::memcpy(out, z, (size_t)l); return l; }
    char buf[40];
    const auto res = std::to_chars(buf, buf + sizeof(buf), d, std::chars_format::scientific);
    // [-]d[.ddd]e[+-]xx
    const char* p = buf;
    char* o = out;
    if (*p == '-') { *o++ = '-'; p++; }
    char digits[24];
    int nd = 0;
    for (; p < res.ptr && *p != 'e'; p++)
        if (*p >= '0' && *p <= '9') digits[nd++] = *p;
    int x = 0;
    if (p < res.ptr) {  // exponent
        p++;
        bool xneg = false;
        if (*p == '-') { xneg = true; p++; } else if (*p == '+') p++;
        for (; p < res.ptr; p++) x = x * 10 + (*p - '0');
        if (xneg) x = -x;
    }
    while (nd > 1 && digits[nd - 1] == '0') nd--;
    const double a = std::fabs(d);
    if (a >= 1e-3 && a < 1e7) {
        if (x >= 0) {
            for (int i = 0; i <= x; i++) *o++ = i < nd ? digits[i] : '0';
            *o++ = '.';
            if (nd > x + 1) for (int i = x + 1; i < nd; i++) *o++ = digits[i];
            else *o++ = '0';
        } else {
            *o++ = '0'; *o++ = '.';
            for (int i = 0; i < -x - 1; i++) *o++ = '0';
            for (int i = 0; i < nd; i++) *o++ = digits[i];
        }
    } else {
        *o++ = digits[0]; *o++ = '.';
        if (nd > 1) for (int i = 1; i < nd; i++) *o++ = digits[i];
        else *o++ = '0';
        *o++ = 'E';
        o += std::snprintf(o, 8, "%d", x);
    }
    return (int)(o - out);
}
std::string javaDoubleToString(double d) {
    char buf[40];
    return std::string(buf, (size_t)javaDoubleToChars(d, buf));
}

std::vector<SplitAndWeight> splitsFromWeights(const std::vector<int32_t>& ordering, const double* weights, int nTaxa) {
    std::vector<SplitAndWeight> splits;
    const double optionThreshold = 0.000001;  // FastNN.java:455
    int64_t index = 0;
    for (int i = 0; i < nTaxa; i++) {
        std::vector<char> member((size_t)nTaxa + 1, 0);
        for (int j = i + 1; j < nTaxa; j++) {
            member[(size_t)ordering[(size_t)j]] = 1;  // split.set(ordering[j]) (FastNN.java:415)
            if (weights[index] > optionThreshold) {
                SplitAndWeight saw;
                saw.weight = weights[index];
                for (int t = 0; t <= nTaxa; t++)
                    if (member[(size_t)t]) saw.split.push_back(t);
                splits.push_back(std::move(saw));
            }
            index++;
        }
    }
    return splits;
}

namespace {

int hostThreads() { return (int)std::max(1u, std::min(16u, std::thread::hardware_concurrency())); }

// runs fn(t) for t = 0 .. count-1 on threads of its own and joins them
template <class F>
void onThreads(int count, F fn) {
    std::vector<std::thread> th;
    for (int t = 1; t < count; t++) th.emplace_back(fn, t);
    if (count > 0) fn(0);
    for (auto& x : th) x.join();
}

void printTaxa(std::FILE* out, const DistancesAndNames& dan) {  // OutputPrinter.java:21-32
    std::fprintf(out, "#nexus\n\n");
    std::fprintf(out, "BEGIN Taxa;\nDIMENSIONS ntax=%d;\nTAXLABELS\n", dan.nTaxa);
    for (int i = 0; i < dan.nTaxa; i++) std::fprintf(out, "[%d] '%s'\n", i + 1, dan.names[(size_t)i].c_str());
    std::fprintf(out, ";\nEND; [Taxa]\n\n");
}

// PrintDistances (:34-47): n^2 numbers, a billion at 32768 taxa.  Bands of rows are formatted by all host threads into
// character buffers of their own and written in order.  A band is first gathered from the packed triangle into a dense
// tile: its entries left of the diagonal are COLUMN accesses of the triangle (stride ~ n doubles), which the gather
// walks column by column, so that the band's rows share the cache lines (88 -> 52 ns per number).
void printDistances(std::FILE* out, const DistancesAndNames& dan) {
    const int ntax = dan.nTaxa;
    std::fprintf(out, "BEGIN Distances;\nDIMENSIONS ntax=%d;\nFORMAT labels=no diagonal triangle=both;\nMATRIX\n", ntax);
    const int nth = hostThreads();
    const int rows_per = ntax >= 2048 ? 32 : std::max(1, 4096 / std::max(ntax, 1));
    std::vector<std::vector<char>> text((size_t)nth);
    std::vector<std::vector<double>> tile((size_t)nth);
    std::vector<size_t> len((size_t)nth);
    for (int base = 0; base < ntax; base += nth * rows_per) {
        onThreads(nth, [&](int t) {
            const int r0 = std::min(ntax, base + t * rows_per), r1 = std::min(ntax, r0 + rows_per), R = r1 - r0;
            len[(size_t)t] = 0;
            if (R <= 0) return;
            std::vector<double>& tl = tile[(size_t)t];
            tl.resize((size_t)rows_per * (size_t)ntax);
            for (int j = 0; j < r0; j++) {  // left of the band's diagonal block: entries (j, r0 .. r1-1) are adjacent
                const double* src = dan.distances.data() + dan.upperIndex(j, r0);
                for (int k = 0; k < R; k++) tl[(size_t)k * (size_t)ntax + (size_t)j] = src[k];
            }
            for (int i = r0; i < r1; i++) {
                double* row = tl.data() + (size_t)(i - r0) * (size_t)ntax;
                for (int j = r0; j < r1; j++) row[j] = dan.get(i, j);
                if (r1 < ntax) std::memcpy(row + r1, dan.distances.data() + dan.upperIndex(i, r1), sizeof(double) * (size_t)(ntax - r1));
            }
            std::vector<char>& b = text[(size_t)t];
            b.resize((size_t)rows_per * ((size_t)ntax * 27 + 1));  // " " + at most 26 characters per number
            char* o = b.data();
            for (int k = 0; k < R; k++) {
                const double* row = tl.data() + (size_t)k * (size_t)ntax;
                for (int j = 0; j < ntax; j++) { *o++ = ' '; o += javaDoubleToChars(row[j], o); }
                *o++ = '\n';
            }
            len[(size_t)t] = (size_t)(o - b.data());
        });
        for (int t = 0; t < nth; t++)
            if (len[(size_t)t]) std::fwrite(text[(size_t)t].data(), 1, len[(size_t)t], out);
    }
    std::fprintf(out, ";\nEND; [Distances]\n\n");
}

void printSplitsHead(std::FILE* out, int ntax, size_t nsplits, const std::vector<int32_t>& order) {  // :49-62
    std::fprintf(out, "BEGIN Splits;\nDIMENSIONS ntax=%d nsplits=%zu;\n", ntax, nsplits);
    std::fprintf(out, "FORMAT labels=no weights=yes confidences=no intervals=no;\nPROPERTIES fit=-1.0 cyclic;\nCYCLE");
    for (size_t i = 1; i < order.size(); i++) std::fprintf(out, " %d", order[i]);
    std::fprintf(out, ";\nMATRIX\n");
}

void printAssumptions(std::FILE* out, int ntax) {  // :87-96
    std::fprintf(out, ";\nEND; [Splits]\n\n");
    std::fprintf(out, "BEGIN st_Assumptions;\nuptodate;\ndisttransform=NeighborNet;\nsplitstransform=EqualAngle;\n");
    std::fprintf(out, "SplitsPostProcess filter=dimension value=%d;\n exclude  no missing;\nautolayoutnodelabels;\nEND; [st_Assumptions]\n\n", ntax);
}

}  // namespace

void printNexusWithSplitsAndDistances(std::FILE* out, const std::vector<int32_t>& order, const DistancesAndNames& dan,
                                      const std::vector<SplitAndWeight>& splits) {
    const int ntax = dan.nTaxa;
    printTaxa(out, dan);
    printDistances(out, dan);
    // PrintSplits (:49-69), PrintSplit (:71-85)
    printSplitsHead(out, ntax, splits.size(), order);
    int counter = 1;
    for (const SplitAndWeight& saw : splits) {
        int size = (int)saw.split.size();
        if (ntax - size < size) size = ntax - size;
        std::fprintf(out, "[%d, size=%d] \t %s \t ", counter, size, javaDoubleToString(saw.weight).c_str());
        for (int32_t t : saw.split) std::fprintf(out, " %d", t);
        std::fprintf(out, ",\n");
        counter++;
    }
    printAssumptions(out, ntax);
}

// The same document straight from the weights in live index order, without the member lists (at 32768 taxa the 77 000
// splits list 0.8 billion taxon ids: 3 GB as vectors, a minute of fprintf calls).  The positive splits are found and their
// lines formatted by all host threads: split (i, j) is the ascending list of ordering[i+1 .. j], read off a bit set that a
// thread extends from one split to the next of the same i; the ids' texts come out of a table.
size_t printNexusFromWeights(std::FILE* out, const std::vector<int32_t>& order, const DistancesAndNames& dan, const double* weights) {
    const int ntax = dan.nTaxa;
    const double optionThreshold = 0.000001;  // FastNN.java:455
    printTaxa(out, dan);
    printDistances(out, dan);
    const int nth = hostThreads();
    struct Pos { int32_t i, j; double w; };
    std::vector<Pos> pos;
    {   // the positive splits in index order: thread t looks at the rows i = t, t + nth, ... (balanced: row i has n - 1 - i entries)
        std::vector<std::vector<Pos>> part((size_t)nth);
        onThreads(nth, [&](int t) {
            for (int i = t; i < ntax; i += nth) {
                const int64_t at = (int64_t)i * (ntax - 1) - (int64_t)i * (i - 1) / 2 - (i + 1);  // + j: the live index of (i, j)
                for (int j = i + 1; j < ntax; j++)
                    if (weights[at + j] > optionThreshold) part[(size_t)t].push_back(Pos{i, j, weights[at + j]});
            }
        });
        std::vector<size_t> at((size_t)nth, 0);
        size_t total = 0;
        for (const auto& p : part) total += p.size();
        pos.reserve(total);
        for (int i = 0; i < ntax; i++) {  // rows back in order
            const std::vector<Pos>& p = part[(size_t)(i % nth)];
            size_t& a = at[(size_t)(i % nth)];
            while (a < p.size() && p[a].i == i) pos.push_back(p[a++]);
        }
    }
    printSplitsHead(out, ntax, pos.size(), order);
    // " <id>" for the ids 0 .. ntax
    std::vector<char> idtext;
    std::vector<uint32_t> idoff((size_t)ntax + 2);
    for (int t = 0; t <= ntax; t++) {
        char tmp[16];
        const int l = std::snprintf(tmp, sizeof(tmp), " %d", t);
        idoff[(size_t)t] = (uint32_t)idtext.size();
        idtext.insert(idtext.end(), tmp, tmp + l);
    }
    idoff[(size_t)ntax + 1] = (uint32_t)idtext.size();
    const size_t idmax = (size_t)(idoff[(size_t)ntax + 1] - idoff[(size_t)ntax]);
    const size_t words = ((size_t)ntax + 64) / 64;
    const size_t budget = (size_t)24 << 20;  // characters per thread and round (more only for one very long line)
    std::vector<std::vector<char>> text((size_t)nth);
    std::vector<std::vector<uint64_t>> bits((size_t)nth, std::vector<uint64_t>(words));
    std::vector<size_t> len((size_t)nth), from((size_t)nth + 1);
    for (size_t next = 0; next < pos.size();) {
        for (int t = 0; t < nth; t++) {  // deal the next splits out by their size
            from[(size_t)t] = next;
            size_t bound = 0;
            while (next < pos.size()) {
                const size_t line = 96 + (size_t)(pos[next].j - pos[next].i) * idmax;
                if (bound > 0 && bound + line > budget) break;
                bound += line;
                next++;
            }
            len[(size_t)t] = bound;
        }
        from[(size_t)nth] = next;
        onThreads(nth, [&](int t) {
            const size_t s0 = from[(size_t)t], s1 = from[(size_t)t + 1];
            std::vector<char>& b = text[(size_t)t];
            if (b.size() < len[(size_t)t]) b.resize(len[(size_t)t]);
            char* o = b.data();
            std::vector<uint64_t>& bs = bits[(size_t)t];
            int ci = -1, cj = -1;  // the bit set holds ordering[ci+1 .. cj]
            for (size_t s = s0; s < s1; s++) {
                const Pos& p = pos[s];
                if (p.i != ci) { std::fill(bs.begin(), bs.end(), 0); ci = p.i; cj = p.i; }
                for (; cj < p.j; cj++) { const int32_t id = order[(size_t)cj + 1]; bs[(size_t)id >> 6] |= (uint64_t)1 << (id & 63); }
                int size = p.j - p.i;
                if (ntax - size < size) size = ntax - size;
                o += std::snprintf(o, 40, "[%zu, size=%d] \t ", s + 1, size);
                o += javaDoubleToChars(p.w, o);
                std::memcpy(o, " \t ", 3); o += 3;
                for (size_t wd = 0; wd < words; wd++)
                    for (uint64_t m = bs[wd]; m; m &= m - 1) {
                        const size_t id = wd * 64 + (size_t)__builtin_ctzll(m);
                        const uint32_t a = idoff[id], l = idoff[id + 1] - a;
                        std::memcpy(o, idtext.data() + a, l);
                        o += l;
                    }
                *o++ = ','; *o++ = '\n';
            }
            len[(size_t)t] = (size_t)(o - b.data());
        });
        for (int t = 0; t < nth; t++)
            if (len[(size_t)t]) std::fwrite(text[(size_t)t].data(), 1, len[(size_t)t], out);
    }
    printAssumptions(out, ntax);
    return pos.size();
}

}  // namespace nnet

extern "C" {
// Double.toString of d into buf (>= 32 bytes); test hook
void fnnh_java_double(double d, char* buf) { std::snprintf(buf, 32, "%s", nnet::javaDoubleToString(d).c_str()); }
// Nexus document for (names, D, order, live-order weights) into the file `path`; test hook. Returns the number of splits or -1.
int32_t fnnh_write_nexus(const char* path, int32_t n, const double* D, const char* names256, const int32_t* order, const double* weights) {
    try {
        nnet::DistancesAndNames dan;
        dan.nTaxa = n;
        dan.distances.resize((size_t)n * (size_t)(n - 1) / 2);
        for (int i = 0; i < n; i++) dan.names.emplace_back(names256 + (size_t)i * 256);
        const int nth = nnet::hostThreads();
        nnet::onThreads(nth, [&](int t) {
            for (int i = t; i < n - 1; i += nth)
                std::memcpy(dan.distances.data() + dan.upperIndex(i, i + 1), D + (size_t)i * (size_t)n + (size_t)i + 1, sizeof(double) * (size_t)(n - 1 - i));
        });
        std::vector<int32_t> ord(order, order + n + 1);
        std::FILE* f = std::fopen(path, "w");
        if (!f) return -1;
        const size_t ns = nnet::printNexusFromWeights(f, ord, dan, weights);
        std::fclose(f);
        return (int32_t)ns;
    } catch (...) { return -1; }
}
// the same through the member lists (splitsFromWeights + printNexusWithSplitsAndDistances); test hook
int32_t fnnh_write_nexus_lists(const char* path, int32_t n, const double* D, const char* names256, const int32_t* order, const double* weights) {
    try {
        nnet::DistancesAndNames dan;
        dan.nTaxa = n;
        dan.distances.resize((size_t)n * (size_t)(n - 1) / 2);
        for (int i = 0; i < n; i++) {
            dan.names.emplace_back(names256 + (size_t)i * 256);
            for (int j = i + 1; j < n; j++) dan.distances[(size_t)dan.upperIndex(i, j)] = D[(size_t)i * (size_t)n + (size_t)j];
        }
        std::vector<int32_t> ord(order, order + n + 1);
        auto splits = nnet::splitsFromWeights(ord, weights, n);
        std::FILE* f = std::fopen(path, "w");
        if (!f) return -1;
        nnet::printNexusWithSplitsAndDistances(f, ord, dan, splits);
        std::fclose(f);
        return (int32_t)splits.size();
    } catch (...) { return -1; }
}
int32_t fnnh_read_taxa_count(const char* path) {
    try { return nnet::readTaxaCount(path); } catch (...) { return -1; }
}
// out: n*n dense matrix; names_out (may be NULL): n buffers of 256 bytes. Returns 0 or -1.
int32_t fnnh_read_phylip(const char* path, int32_t n, double* out, char* names_out) {
    try {
        nnet::DistancesAndNames dan(path, n);
        std::vector<double> D = dan.toMatrix();
        for (size_t i = 0; i < D.size(); i++) out[i] = D[i];
        if (names_out)
            for (int i = 0; i < n; i++) {
                std::string s = dan.names[(size_t)i].substr(0, 255);
                std::snprintf(names_out + (size_t)i * 256, 256, "%s", s.c_str());
            }
        return 0;
    } catch (...) { return -1; }
}
}
