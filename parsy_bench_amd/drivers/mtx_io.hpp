// MatrixMarket / ordering-file readers of the drivers.
//
// Same input contract as the reference's readers (common/Util.h:77-179 `readMatrix`,
// :187-223 `readOrdering`): "%%MatrixMarket matrix coordinate real ..." holding the
// LOWER triangle only, entries sorted by column; an ordering file is the dimension
// followed by n zero-based permutation entries.  Unlike the reference this reader
// checks what it reads (sortedness, bounds, entry count).
#pragma once
#include <algorithm>
#include <cctype>
#include <fstream>
#include <iostream>
#include <sstream>
#include <string>
#include <vector>

namespace parsy_io {

inline bool read_lower_mtx(const std::string& path, int& n, std::vector<int>& Ap, std::vector<int>& Ai,
                           std::vector<double>& Ax) {
    std::ifstream in(path);
    if (!in) {
        std::cerr << "cannot open " << path << "\n";
        return false;
    }
    std::string line;
    std::getline(in, line);
    std::string low = line;
    std::transform(low.begin(), low.end(), low.begin(), [](unsigned char c) { return std::tolower(c); });
    std::istringstream hs(low);
    std::string banner, mtx, crd, arith, sym;
    if (!(hs >> banner >> mtx >> crd >> arith >> sym) || banner != "%%matrixmarket" || mtx != "matrix" ||
        crd != "coordinate" || arith != "real") {
        std::cerr << path << ": not a 'matrix coordinate real' MatrixMarket file\n";
        return false;
    }
    do {
        if (!std::getline(in, line)) return false;
    } while (line.empty() || line[0] == '%');
    long long nr = 0, nc = 0, nnz = 0;
    {
        std::istringstream ss(line);
        if (!(ss >> nr >> nc >> nnz) || nr != nc || nr <= 0 || nnz <= 0) {
            std::cerr << path << ": bad size line\n";
            return false;
        }
    }
    n = (int)nr;
    Ap.assign(n + 1, 0);
    Ai.resize((size_t)nnz);
    Ax.resize((size_t)nnz);
    int prev_col = 0, prev_row = -1;
    for (long long k = 0; k < nnz; ++k) {
        long long r, c;
        double v;
        if (!(in >> r >> c >> v)) {
            std::cerr << path << ": truncated at entry " << k << "\n";
            return false;
        }
        --r;
        --c;
        if (r < 0 || r >= n || c < 0 || c >= n || r < c) {
            std::cerr << path << ": entry " << k << " is outside the lower triangle\n";
            return false;
        }
        if (c < prev_col || (c == prev_col && r <= prev_row)) {
            std::cerr << path << ": entries must be sorted by column, then row (reference README.md:31)\n";
            return false;
        }
        if (c != prev_col) prev_row = -1;
        prev_col = (int)c;
        prev_row = (int)r;
        Ai[k] = (int)r;
        Ax[k] = v;
        Ap[c + 1]++;
    }
    for (int j = 0; j < n; ++j) Ap[j + 1] += Ap[j];
    for (int j = 0; j < n; ++j)
        if (Ap[j + 1] == Ap[j] || Ai[Ap[j]] != j) {
            std::cerr << path << ": column " << j << " has no diagonal entry\n";
            return false;
        }
    return true;
}

inline bool read_ordering(const std::string& path, int n, std::vector<int>& perm) {
    std::ifstream in(path);
    if (!in) {
        std::cerr << "cannot open " << path << "\n";
        return false;
    }
    std::string line;
    long long nn = -1;
    while (std::getline(in, line)) {
        if (line.empty() || line[0] == '%') continue;
        std::istringstream ss(line);
        if (ss >> nn) break;
    }
    if (nn != n) {
        std::cerr << path << ": ordering is for n = " << nn << ", matrix has n = " << n << "\n";
        return false;
    }
    perm.resize(n);
    std::vector<char> seen(n, 0);
    for (int k = 0; k < n; ++k) {
        long long v;
        if (!(in >> v) || v < 0 || v >= n || seen[v]) {
            std::cerr << path << ": not a permutation\n";
            return false;
        }
        seen[v] = 1;
        perm[k] = (int)v;
    }
    return true;
}

}  // namespace parsy_io
