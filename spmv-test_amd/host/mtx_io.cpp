// mtx_io.cpp -- see mtx_io.hpp.
#include "mtx_io.hpp"
#include <algorithm>
#include <cctype>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <numeric>
#include <sstream>

namespace {
std::string lower(std::string s)
{
    for (auto &c : s) c = (char)std::tolower((unsigned char)c);
    return s;
}
struct Entry {
    int32_t r, c;
    float v;
};
}  // namespace

std::string read_matrix_market(const std::string &path, HostCsr &out)
{
    std::ifstream f(path);
    if (!f) return "cannot open " + path;
    std::string line;
    if (!std::getline(f, line)) return "empty file";
    std::istringstream hs(line);
    std::string banner, object, format, field, symmetry;
    hs >> banner >> object >> format >> field >> symmetry;
    if (lower(banner) != "%%matrixmarket" || lower(object) != "matrix") return "not a Matrix Market matrix file";
    if (lower(format) != "coordinate") return "only the coordinate format is supported";
    field = lower(field);
    symmetry = lower(symmetry);
    const bool pattern = field == "pattern";
    if (!pattern && field != "real" && field != "integer") return "unsupported field '" + field + "'";
    const bool sym = symmetry == "symmetric", skew = symmetry == "skew-symmetric";
    if (!sym && !skew && symmetry != "general") return "unsupported symmetry '" + symmetry + "'";
    while (std::getline(f, line))
        if (!line.empty() && line[0] != '%') break;
    long long R = 0, C = 0, NZ = 0;
    if (std::sscanf(line.c_str(), "%lld %lld %lld", &R, &C, &NZ) != 3) return "bad size line";
    if (R < 0 || C < 0 || NZ < 0 || R >= (1LL << 31) || C >= (1LL << 31)) return "dimensions out of range";
    std::vector<Entry> e;
    e.reserve((size_t)(sym || skew ? 2 * NZ : NZ));
    for (long long k = 0; k < NZ; ++k) {
        long long r, c;
        double v = 1.0;
        if (!(f >> r >> c)) return "truncated entry list";
        if (!pattern && !(f >> v)) return "truncated entry list";
        if (r < 1 || r > R || c < 1 || c > C) return "entry index out of range";
        e.push_back({(int32_t)(r - 1), (int32_t)(c - 1), (float)v});
        if ((sym || skew) && r != c) e.push_back({(int32_t)(c - 1), (int32_t)(r - 1), (float)(skew ? -v : v)});
    }
    std::stable_sort(e.begin(), e.end(), [](const Entry &a, const Entry &b) { return a.r != b.r ? a.r < b.r : a.c < b.c; });
    out = HostCsr();
    out.rows = R;
    out.cols = C;
    out.row_ptr.assign((size_t)R + 1, 0);
    for (size_t i = 0; i < e.size();) {  // duplicates are summed, in file order
        size_t j = i;
        float v = 0.0f;
        while (j < e.size() && e[j].r == e[i].r && e[j].c == e[i].c) v += e[j++].v;
        out.col_idx.push_back(e[i].c);
        out.vals.push_back(v);
        out.row_ptr[(size_t)e[i].r + 1]++;
        i = j;
    }
    if (out.vals.size() >= (1ull << 31)) return "more than 2^31 nonzeros: split into row blocks";
    std::partial_sum(out.row_ptr.begin(), out.row_ptr.end(), out.row_ptr.begin());
    return "";
}

static const char kMagic[8] = {'S', 'P', 'M', 'V', 'C', 'S', 'R', '1'};

std::string write_csr_binary(const std::string &path, const HostCsr &m)
{
    std::ofstream f(path, std::ios::binary);
    if (!f) return "cannot create " + path;
    const int64_t hdr[3] = {m.rows, m.cols, m.nnz()};
    f.write(kMagic, 8);
    f.write((const char *)hdr, sizeof hdr);
    f.write((const char *)m.row_ptr.data(), (std::streamsize)(sizeof(int32_t) * m.row_ptr.size()));
    f.write((const char *)m.col_idx.data(), (std::streamsize)(sizeof(int32_t) * m.col_idx.size()));
    f.write((const char *)m.vals.data(), (std::streamsize)(sizeof(float) * m.vals.size()));
    return f ? "" : "write failed";
}

std::string read_csr_binary(const std::string &path, HostCsr &out)
{
    std::ifstream f(path, std::ios::binary);
    if (!f) return "cannot open " + path;
    char magic[8];
    int64_t hdr[3];
    f.read(magic, 8);
    f.read((char *)hdr, sizeof hdr);
    if (!f || std::memcmp(magic, kMagic, 8) != 0) return "not a .csrbin file";
    if (hdr[0] < 0 || hdr[1] < 0 || hdr[2] < 0 || hdr[0] >= (1LL << 31) || hdr[1] >= (1LL << 31) || hdr[2] >= (1LL << 31))
        return "header out of range";
    out = HostCsr();
    out.rows = hdr[0];
    out.cols = hdr[1];
    out.row_ptr.resize((size_t)hdr[0] + 1);
    out.col_idx.resize((size_t)hdr[2]);
    out.vals.resize((size_t)hdr[2]);
    f.read((char *)out.row_ptr.data(), (std::streamsize)(sizeof(int32_t) * out.row_ptr.size()));
    f.read((char *)out.col_idx.data(), (std::streamsize)(sizeof(int32_t) * out.col_idx.size()));
    f.read((char *)out.vals.data(), (std::streamsize)(sizeof(float) * out.vals.size()));
    if (!f) return "truncated file";
    if (out.row_ptr.front() != 0 || out.row_ptr.back() != hdr[2]) return "inconsistent row_ptr";
    return "";
}

std::string write_vector_text(const std::string &path, const std::vector<float> &y)
{
    std::FILE *f = std::fopen(path.c_str(), "w");
    if (!f) return "cannot create " + path;
    for (float v : y) std::fprintf(f, "%.9g\n", v);
    return std::fclose(f) == 0 ? "" : "write failed";
}
