// fastnn_host.cpp -- Phylip reader and output formatting (no GPU code).
// Follows DistancesAndNames.java:43-132 including its quirks: the first line is skipped,
// lines are split on single spaces, the first token is the name, remaining non-blank tokens
// are further split on tabs, only the first `row` values of a line are used (so square and
// lower-triangular files both work), the token buffer is NOT cleared between lines, reading
// stops at EOF or at the first line beyond numTaxa rows.
#include "fastnn_host.hpp"

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
extern "C" {
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
