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

void printNexusWithSplitsAndDistances(std::FILE* out, const std::vector<int32_t>& order, const DistancesAndNames& dan,
                                      const std::vector<SplitAndWeight>& splits) {
    const int ntax = dan.nTaxa;
    std::fprintf(out, "#nexus\n\n");
    // PrintTaxa (OutputPrinter.java:21-32)
    std::fprintf(out, "BEGIN Taxa;\nDIMENSIONS ntax=%d;\nTAXLABELS\n", ntax);
    for (int i = 0; i < ntax; i++) std::fprintf(out, "[%d] '%s'\n", i + 1, dan.names[(size_t)i].c_str());
    std::fprintf(out, ";\nEND; [Taxa]\n\n");
    // PrintDistances (:34-47)
    std::fprintf(out, "BEGIN Distances;\nDIMENSIONS ntax=%d;\nFORMAT labels=no diagonal triangle=both;\nMATRIX\n", ntax);
    {   // n^2 numbers (a billion at 32768 taxa): rows are formatted by all host threads into buffers, written in order
        const int nth = (int)std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
        const int rows_per = ntax >= 4096 ? 16 : std::max(1, 4096 / std::max(ntax, 1));
        std::vector<std::string> bufs((size_t)nth);
        for (int base = 0; base < ntax; base += nth * rows_per) {
            std::vector<std::thread> th;
            for (int t = 0; t < nth; t++) {
                const int r0 = base + t * rows_per, r1 = std::min(ntax, r0 + rows_per);
                bufs[(size_t)t].clear();
                if (r0 >= r1) continue;
                th.emplace_back([&, t, r0, r1]() {
                    std::string& b = bufs[(size_t)t];
                    b.reserve((size_t)(r1 - r0) * (size_t)ntax * 22);
                    char tmp[40];
                    for (int i = r0; i < r1; i++) {
                        for (int j = 0; j < ntax; j++) { b.push_back(' '); b.append(tmp, (size_t)javaDoubleToChars(dan.get(i, j), tmp)); }
                        b.push_back('\n');
                    }
                });
            }
            for (auto& x : th) x.join();
            for (int t = 0; t < nth; t++) std::fwrite(bufs[(size_t)t].data(), 1, bufs[(size_t)t].size(), out);
        }
    }
    std::fprintf(out, ";\nEND; [Distances]\n\n");
    // PrintSplits (:49-69), PrintSplit (:71-85)
    std::fprintf(out, "BEGIN Splits;\nDIMENSIONS ntax=%d nsplits=%zu;\n", ntax, splits.size());
    std::fprintf(out, "FORMAT labels=no weights=yes confidences=no intervals=no;\nPROPERTIES fit=-1.0 cyclic;\nCYCLE");
    for (size_t i = 1; i < order.size(); i++) std::fprintf(out, " %d", order[i]);
    std::fprintf(out, ";\nMATRIX\n");
    int counter = 1;
    for (const SplitAndWeight& saw : splits) {
        int size = (int)saw.split.size();
        if (ntax - size < size) size = ntax - size;
        std::fprintf(out, "[%d, size=%d] \t %s \t ", counter, size, javaDoubleToString(saw.weight).c_str());
        for (int32_t t : saw.split) std::fprintf(out, " %d", t);
        std::fprintf(out, ",\n");
        counter++;
    }
    std::fprintf(out, ";\nEND; [Splits]\n\n");
    // PrintAssumptions (:87-96)
    std::fprintf(out, "BEGIN st_Assumptions;\nuptodate;\ndisttransform=NeighborNet;\nsplitstransform=EqualAngle;\n");
    std::fprintf(out, "SplitsPostProcess filter=dimension value=%d;\n exclude  no missing;\nautolayoutnodelabels;\nEND; [st_Assumptions]\n\n", ntax);
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
