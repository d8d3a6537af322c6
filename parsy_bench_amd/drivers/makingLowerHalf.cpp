// makingLowerHalf <full.mtx> > lower.mtx
//
// The input converter of the reference's workflow (examples/MakingLowerHalf.cpp): turns a
// "%%MatrixMarket matrix coordinate real general|symmetric" file holding a symmetric matrix into the
// form the drivers read (common/Util.h:77): lower triangle only, entries sorted by column, header
// "... real symmetric".  As the reference, the diagonal is moved away from zero by tol = 0.1
// (value >= 0 ? value + tol : value - tol).  Differences, on purpose: the input need not be sorted
// (entries are ordered here: column, then row), a symmetric-format input (lower or upper triangle
// stored) is accepted as well, duplicates are summed, a missing diagonal entry becomes tol, the entry
// count of the header is exact, and values are printed with 17 significant digits instead of 6.
#include <algorithm>
#include <cctype>
#include <cstdio>
#include <fstream>
#include <iostream>
#include <sstream>
#include <string>
#include <tuple>
#include <vector>

int main(int argc, char* argv[]) {
    if (argc < 2) {
        std::fprintf(stderr, "usage: %s <matrix.mtx>  (lower half goes to stdout)\n", argv[0]);
        return -1;
    }
    std::ifstream in(argv[1]);
    if (!in) {
        std::cerr << "cannot open " << argv[1] << "\n";
        return -1;
    }
    std::string line;
    std::getline(in, line);
    std::string low = line;
    std::transform(low.begin(), low.end(), low.begin(), [](unsigned char c) { return std::tolower(c); });
    std::istringstream hs(low);
    std::string banner, mtx, crd, arith, sym;
    if (!(hs >> banner >> mtx >> crd >> arith >> sym) || banner != "%%matrixmarket" || mtx != "matrix" ||
        crd != "coordinate" || arith != "real") {
        std::cerr << argv[1] << ": not a 'matrix coordinate real' MatrixMarket file\n";
        return -1;
    }
    do {
        if (!std::getline(in, line)) return -1;
    } while (line.empty() || line[0] == '%');
    long long nr = 0, nc = 0, nnz = 0;
    {
        std::istringstream ss(line);
        if (!(ss >> nr >> nc >> nnz) || nr != nc || nr <= 0 || nnz <= 0) {
            std::cerr << argv[1] << ": bad size line\n";
            return -1;
        }
    }
    const double tol = 0.1;
    std::vector<std::tuple<long long, long long, double>> ent;  // (col, row, value), row >= col, 1-based
    ent.reserve((size_t)nnz);
    for (long long k = 0; k < nnz; ++k) {
        long long r, c;
        double v;
        if (!(in >> r >> c >> v)) {
            std::cerr << argv[1] << ": truncated at entry " << k << "\n";
            return -1;
        }
        if (r < 1 || c < 1 || r > nr || c > nr) {
            std::cerr << argv[1] << ": entry " << k << " out of range\n";
            return -1;
        }
        if (sym == "general") {
            if (r >= c) ent.emplace_back(c, r, v);  // the upper half of a general file is the mirror image: dropped
        } else {
            ent.emplace_back(std::min(r, c), std::max(r, c), v);  // symmetric file: either triangle may be stored
        }
    }
    std::sort(ent.begin(), ent.end(), [](const auto& a, const auto& b) {
        return std::get<0>(a) != std::get<0>(b) ? std::get<0>(a) < std::get<0>(b) : std::get<1>(a) < std::get<1>(b);
    });
    // sum duplicates, make sure every diagonal entry exists
    std::vector<std::tuple<long long, long long, double>> out;
    out.reserve(ent.size() + (size_t)nr);
    size_t q = 0;
    for (long long c = 1; c <= nr; ++c) {
        bool have_diag = false;
        while (q < ent.size() && std::get<0>(ent[q]) == c) {
            long long r = std::get<1>(ent[q]);
            double v = 0.0;
            while (q < ent.size() && std::get<0>(ent[q]) == c && std::get<1>(ent[q]) == r) v += std::get<2>(ent[q++]);
            if (r == c) {
                have_diag = true;
                v = v >= 0 ? v + tol : v - tol;
            }
            out.emplace_back(c, r, v);
        }
        if (!have_diag) {
            // keep the column sorted: the diagonal is its first entry
            size_t pos = out.size();
            while (pos > 0 && std::get<0>(out[pos - 1]) == c) --pos;
            out.insert(out.begin() + (long)pos, std::make_tuple(c, c, tol));
        }
    }
    std::printf("%%%%MatrixMarket matrix coordinate real symmetric\n%lld %lld %zu\n", nr, nr, out.size());
    for (const auto& e : out) std::printf("%lld %lld %.17g\n", std::get<1>(e), std::get<0>(e), std::get<2>(e));
    return 0;
}
