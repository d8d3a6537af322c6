// Host inspector: see inspector.hpp for the contract and reference citations.
#include "inspector.hpp"

#include <algorithm>
#include <cassert>
#include <climits>
#include <numeric>
#include <stdexcept>

namespace parsy {

// ---------------------------------------------------------------------------
// CSC helpers
// ---------------------------------------------------------------------------
static void transpose_pattern(const CscPattern& A, CscPattern& T) {
    const int n = A.n;
    const size_t nnz = A.i.size();
    T.n = n;
    T.p.assign(n + 1, 0);
    T.i.resize(nnz);
    const bool hasx = !A.x.empty(), hassrc = !A.src.empty();
    if (hasx) T.x.resize(nnz); else T.x.clear();
    if (hassrc) T.src.resize(nnz); else T.src.clear();
    for (size_t k = 0; k < nnz; ++k) T.p[A.i[k] + 1]++;
    for (int j = 0; j < n; ++j) T.p[j + 1] += T.p[j];
    std::vector<int> next(T.p.begin(), T.p.end() - 1);
    for (int j = 0; j < n; ++j) {
        for (int k = A.p[j]; k < A.p[j + 1]; ++k) {
            int q = next[A.i[k]]++;
            T.i[q] = j;
            if (hasx) T.x[q] = A.x[k];
            if (hassrc) T.src[q] = A.src[k];
        }
    }
}

void permute_sym(int n, const int* Ap, const int* Ai, const double* Ax, const int* perm,
                 CscPattern& up, CscPattern& lo) {
    std::vector<int> inv(n);
    for (int k = 0; k < n; ++k) inv[perm ? perm[k] : k] = k;
    const int nnz = Ap[n];
    CscPattern raw;  // upper triangle, columns unsorted
    raw.n = n;
    raw.p.assign(n + 1, 0);
    raw.i.resize(nnz);
    raw.src.resize(nnz);
    if (Ax) raw.x.resize(nnz);
    for (int j = 0; j < n; ++j)
        for (int k = Ap[j]; k < Ap[j + 1]; ++k) {
            int a = inv[Ai[k]], b = inv[j];
            raw.p[std::max(a, b) + 1]++;
        }
    for (int j = 0; j < n; ++j) raw.p[j + 1] += raw.p[j];
    std::vector<int> next(raw.p.begin(), raw.p.end() - 1);
    for (int j = 0; j < n; ++j)
        for (int k = Ap[j]; k < Ap[j + 1]; ++k) {
            int a = inv[Ai[k]], b = inv[j];
            int q = next[std::max(a, b)]++;
            raw.i[q] = std::min(a, b);
            raw.src[q] = k;
            if (Ax) raw.x[q] = Ax[k];
        }
    transpose_pattern(raw, lo);  // lower triangle, sorted
    transpose_pattern(lo, up);   // upper triangle, sorted
}

// ---------------------------------------------------------------------------
// Elimination tree (Liu's algorithm with path compression) of the upper
// triangle: parent[j] = min{ i > j : L(i,j) != 0 }.
// ---------------------------------------------------------------------------
void etree_upper(const CscPattern& up, std::vector<int>& parent) {
    const int n = up.n;
    parent.assign(n, -1);
    std::vector<int> anc(n, -1);
    for (int k = 0; k < n; ++k) {
        for (int q = up.p[k]; q < up.p[k + 1]; ++q) {
            int i = up.i[q];
            while (i != -1 && i < k) {
                int nxt = anc[i];
                anc[i] = k;
                if (nxt == -1) parent[i] = k;
                i = nxt;
            }
        }
    }
}

// Postorder of a forest. Children are visited lightest-first when `weight` is
// given (ties by node index), by ascending index otherwise; roots by index.
// (reference semantics: common/PostOrder.h:10-160)
void postorder(const std::vector<int>& parent, const int* weight, std::vector<int>& post) {
    const int n = (int)parent.size();
    std::vector<int> cptr(n + 1, 0), child(n);
    for (int j = 0; j < n; ++j)
        if (parent[j] >= 0) cptr[parent[j] + 1]++;
    for (int j = 0; j < n; ++j) cptr[j + 1] += cptr[j];
    {
        std::vector<int> fill(cptr.begin(), cptr.end() - 1);
        if (!weight) {
            for (int j = 0; j < n; ++j)
                if (parent[j] >= 0) child[fill[parent[j]]++] = j;
        } else {
            // counting sort of nodes by clamped weight, stable in the index
            std::vector<int> wcnt(n + 1, 0), byw(n);
            auto cw = [&](int j) { return std::min(std::max(weight[j], 0), n - 1); };
            for (int j = 0; j < n; ++j) wcnt[cw(j) + 1]++;
            for (int w = 0; w < n; ++w) wcnt[w + 1] += wcnt[w];
            for (int j = 0; j < n; ++j) byw[wcnt[cw(j)]++] = j;
            for (int t = 0; t < n; ++t) {
                int j = byw[t];
                if (parent[j] >= 0) child[fill[parent[j]]++] = j;
            }
        }
    }
    post.clear();
    post.reserve(n);
    std::vector<int> stack, pos(n, 0);
    for (int r = 0; r < n; ++r) {
        if (parent[r] != -1) continue;
        stack.push_back(r);
        while (!stack.empty()) {
            int v = stack.back();
            if (pos[v] < cptr[v + 1] - cptr[v]) {
                stack.push_back(child[cptr[v] + pos[v]++]);
            } else {
                post.push_back(v);
                stack.pop_back();
            }
        }
    }
}

// Column counts of L (Gilbert, Ng & Peyton skeleton-matrix algorithm, in the
// formulation of Davis, "Direct Methods for Sparse Linear Systems", ch. 4).
void col_counts(const CscPattern& lo, const std::vector<int>& parent,
                const std::vector<int>& post, std::vector<int>& cc) {
    const int n = lo.n;
    std::vector<int> first(n, -1), maxfirst(n, -1), prevleaf(n, -1), anc(n);
    cc.assign(n, 0);
    for (int k = 0; k < n; ++k) {
        int j = post[k];
        cc[j] = (first[j] == -1) ? 1 : 0;
        for (; j != -1 && first[j] == -1; j = parent[j]) first[j] = k;
    }
    std::iota(anc.begin(), anc.end(), 0);
    for (int k = 0; k < n; ++k) {
        const int j = post[k];
        if (parent[j] != -1) cc[parent[j]]--;
        for (int q = lo.p[j]; q < lo.p[j + 1]; ++q) {
            const int i = lo.i[q];
            if (i <= j || first[j] <= maxfirst[i]) continue;
            maxfirst[i] = first[j];
            const int jprev = prevleaf[i];
            prevleaf[i] = j;
            cc[j]++;
            if (jprev != -1) {
                int r = jprev;
                while (r != anc[r]) r = anc[r];
                for (int t = jprev; t != r;) {
                    int nx = anc[t];
                    anc[t] = r;
                    t = nx;
                }
                cc[r]--;
            }
        }
        if (parent[j] != -1) anc[j] = parent[j];
    }
    for (int k = 0; k < n; ++k) {
        int j = post[k];
        if (parent[j] != -1) cc[parent[j]] += cc[j];
    }
}

// Level sets by height above the leaves: a node enters the level after all of
// its children's levels; nodes ascending within a level
// (reference semantics: common/TreeUtils.h:119-169 `getLevelSet`).
void level_sets(const std::vector<int>& tree, std::vector<int>& levelPtr,
                std::vector<int>& levelSet) {
    const int n = (int)tree.size();
    std::vector<int> h(n, 0);
    int maxh = -1;
    // children may have larger ids than parents in a general tree: iterate to a
    // fixed point via topological processing (supernodal etrees are postordered,
    // so one ascending pass suffices; keep the general form for safety).
    std::vector<int> nchild(n, 0), queue;
    for (int k = 0; k < n; ++k)
        if (tree[k] >= 0) nchild[tree[k]]++;
    for (int k = 0; k < n; ++k)
        if (nchild[k] == 0) queue.push_back(k);
    for (size_t q = 0; q < queue.size(); ++q) {
        int v = queue[q], par = tree[v];
        maxh = std::max(maxh, h[v]);
        if (par >= 0) {
            h[par] = std::max(h[par], h[v] + 1);
            if (--nchild[par] == 0) queue.push_back(par);
        }
    }
    const int nl = maxh + 1;
    levelPtr.assign(nl + 1, 0);
    for (int k = 0; k < n; ++k) levelPtr[h[k] + 1]++;
    for (int l = 0; l < nl; ++l) levelPtr[l + 1] += levelPtr[l];
    levelSet.resize(n);
    std::vector<int> fill(levelPtr.begin(), levelPtr.end() - 1);
    for (int k = 0; k < n; ++k) levelSet[fill[h[k]]++] = k;
}

// ---------------------------------------------------------------------------
// Supernodes: fundamental supernodes, relaxed amalgamation, row patterns
// (reference semantics: cholesky/Inspection_BlockC.h:116-760)
// ---------------------------------------------------------------------------
static void supernodes(Symbolic& S, const int nrelax[3], const double zrelax[3]) {
    const int n = S.n;
    const std::vector<int>& parent = S.parent;
    const std::vector<int>& cc = S.colcount;

    std::vector<int> nchild(n, 0);
    for (int j = 0; j < n; ++j)
        if (parent[j] != -1) nchild[parent[j]]++;

    std::vector<int> fsuper;  // first column of each fundamental supernode
    if (n > 0) fsuper.push_back(0);
    for (int j = 1; j < n; ++j)
        if (parent[j - 1] != j || cc[j - 1] != cc[j] + 1 || nchild[j] > 1) fsuper.push_back(j);
    const int nf = (int)fsuper.size();
    fsuper.push_back(n);

    std::vector<int> smap(n);
    for (int s = 0; s < nf; ++s)
        for (int k = fsuper[s]; k < fsuper[s + 1]; ++k) smap[k] = s;
    std::vector<int> fpar(nf);
    for (int s = 0; s < nf; ++s) {
        int pj = parent[fsuper[s + 1] - 1];
        fpar[s] = pj == -1 ? -1 : smap[pj];
    }

    // relaxed amalgamation: a supernode may absorb its parent only when the
    // parent is the next supernode in the list.
    std::vector<int> merged(nf, -1), ncol(nf), nrow(nf), zeros(nf, 0);
    for (int s = 0; s < nf; ++s) {
        ncol[s] = fsuper[s + 1] - fsuper[s];
        nrow[s] = cc[fsuper[s]];
    }
    for (int s = nf - 2; s >= 0; --s) {
        int ss = fpar[s];
        if (ss == -1) continue;
        while (merged[ss] != -1) ss = merged[ss];
        const int cur = ss;
        for (ss = fpar[s]; merged[ss] != -1;) {
            int nx = merged[ss];
            merged[ss] = cur;
            ss = nx;
        }
        if (cur != s + 1) continue;
        const int c0 = ncol[s], c1 = ncol[s + 1], ns = c0 + c1;
        int totzeros = zeros[s + 1];
        const double l1 = (double)nrow[s + 1];
        bool merge;
        if (ns <= nrelax[0]) {
            merge = true;
        } else {
            const double l0 = (double)nrow[s];
            const double xnew = c0 * (l1 + c0 - l0);
            const int newzeros = c0 * (nrow[s + 1] + c0 - nrow[s]);
            if (xnew == 0) {
                merge = true;
            } else {
                const double xtot = (double)totzeros + xnew;
                const double xns = (double)ns;
                const double xsize = (xns * (xns + 1) / 2) + xns * (l1 - c1);
                const double z = xtot / xsize;
                totzeros += newzeros;
                merge = ((ns <= nrelax[1] && z < zrelax[0]) || (ns <= nrelax[2] && z < zrelax[1]) ||
                         (z < zrelax[2])) &&
                        (xsize < (double)(INT_MAX / sizeof(double)));
            }
        }
        if (merge) {
            zeros[s] = totzeros;
            merged[s + 1] = s;
            nrow[s] = c0 + nrow[s + 1];
            ncol[s] += ncol[s + 1];
        }
    }

    S.super.clear();
    std::vector<int> snz;
    for (int s = 0; s < nf; ++s)
        if (merged[s] == -1) {
            S.super.push_back(fsuper[s]);
            snz.push_back(nrow[s]);
        }
    const int ns = (int)S.super.size();
    S.super.push_back(n);
    S.nsuper = ns;

    S.col2sup.resize(n);
    for (int s = 0; s < ns; ++s)
        for (int k = S.super[s]; k < S.super[s + 1]; ++k) S.col2sup[k] = s;
    S.sparent.resize(ns);
    for (int s = 0; s < ns; ++s) {
        int pj = parent[S.super[s + 1] - 1];
        S.sparent[s] = pj == -1 ? -1 : S.col2sup[pj];
    }

    S.pi.assign(ns + 1, 0);
    S.ssize = 0;
    S.xsize = 0;
    S.nnzL = 0;
    S.flops_stored = 0;
    S.maxSupWid = 0;
    S.maxCol = 0;
    for (int s = 0; s < ns; ++s) {
        const int64_t w = S.super[s + 1] - S.super[s], r = snz[s];
        S.pi[s + 1] = S.pi[s] + (size_t)r;
        S.ssize += r;
        S.xsize += w * r;
        S.nnzL += w * r - w * (w - 1) / 2;
        for (int64_t t = 0; t < w; ++t) S.flops_stored += (double)(r - t) * (double)(r - t);
        S.maxSupWid = std::max<int>(S.maxSupWid, (int)w);
        S.maxCol = std::max<int>(S.maxCol, (int)r);
    }

    // row patterns: row k belongs to every supernode on the supernodal-etree
    // paths from the supernodes of A(0:k1-1, k) up to (excluding) k's own.
    S.s.assign((size_t)S.ssize, 0);
    std::vector<size_t> fill(S.pi.begin(), S.pi.end() - 1);
    std::vector<int> flag(ns, -1);
    int mark = -1;
    for (int s = 0; s < ns; ++s) {
        const int k1 = S.super[s], k2 = S.super[s + 1];
        for (int k = k1; k < k2; ++k) S.s[fill[s]++] = k;
        for (int k = k1; k < k2; ++k) {
            ++mark;
            flag[s] = mark;
            for (int q = S.A1.p[k]; q < S.A1.p[k + 1]; ++q) {
                const int i = S.A1.i[q];
                if (i >= k1) break;  // sorted: the rest is inside the supernode
                for (int t = S.col2sup[i]; flag[t] != mark; t = S.sparent[t]) {
                    if (fill[t] >= S.pi[t + 1])
                        throw std::runtime_error("inspector: supernode row count overflow");
                    S.s[fill[t]++] = k;
                    flag[t] = mark;
                }
            }
        }
    }
    for (int s = 0; s < ns; ++s)
        if (fill[s] != S.pi[s + 1]) throw std::runtime_error("inspector: supernode row count mismatch");

    // value / row pointers per column (reference: cholesky/LSparsity.h:767-785)
    S.p.assign(n + 1, 0);
    S.i_ptr.assign(n + 1, 0);
    size_t px = 0;
    for (int s = 0; s < ns; ++s) {
        const size_t r = S.pi[s + 1] - S.pi[s];
        for (int k = S.super[s]; k < S.super[s + 1]; ++k) {
            S.p[k] = px;
            S.i_ptr[k] = S.pi[s];
            px += r;
        }
    }
    S.p[n] = px;
    S.i_ptr[n] = S.pi[ns];
}

int ereach_supernodal(const PatternRef& S, int target, std::vector<int>& out,
                      std::vector<char>& mark, std::vector<int>& seq) {
    // paths are discovered in column/entry order; the reference pushes each new
    // path *in front of* the ones found before it (common/Reach.h:122-137).
    seq.clear();
    static thread_local std::vector<int> pstart;
    pstart.clear();
    mark[target] = 1;
    for (int k = S.super[target]; k < S.super[target + 1]; ++k) {
        for (int q = S.A1p[k]; q < S.A1p[k + 1]; ++q) {
            const int row = S.A1i[q];
            if (row > k) continue;
            int i = S.col2sup[row];
            if (mark[i]) continue;
            pstart.push_back((int)seq.size());
            for (; !mark[i]; i = S.sparent[i]) {
                seq.push_back(i);
                mark[i] = 1;
                if (S.sparent[i] < 0) throw std::runtime_error("ereach: walked past a root (invalid etree)");
            }
        }
    }
    pstart.push_back((int)seq.size());
    out.clear();
    for (int pth = (int)pstart.size() - 2; pth >= 0; --pth)
        for (int t = pstart[pth]; t < pstart[pth + 1]; ++t) out.push_back(seq[t]);
    for (int v : seq) mark[v] = 0;
    mark[target] = 0;
    return (int)out.size();
}

void build_update_lists(const PatternRef& S, std::vector<int64_t>& ptr, std::vector<int>& sn,
                        std::vector<int>& lbv, std::vector<int>& ubv) {
    const int ns = S.nsuper;
    ptr.assign(ns + 1, 0);
    sn.clear();
    lbv.clear();
    ubv.clear();
    std::vector<char> mark(ns, 0);
    std::vector<int> tmp, lst;
    for (int t = 0; t < ns; ++t) {
        if (S.prunePtr) {
            if (S.prunePtr[t] > S.prunePtr[t + 1]) throw std::runtime_error("inspector: prunePtr is not monotone");
            lst.assign(S.pruneSet + S.prunePtr[t], S.pruneSet + S.prunePtr[t + 1]);
            for (int d : lst)
                if (d < 0 || d >= t) throw std::runtime_error("inspector: prune set entry is not a descendant");
        } else {
            ereach_supernodal(S, t, lst, mark, tmp);
        }
        const int c0 = S.super[t], c1 = S.super[t + 1];
        for (int d : lst) {
            const size_t b = S.i_ptr[S.super[d]], e = S.i_ptr[S.super[d + 1]];
            const int* rows = S.s + b;
            const int nr = (int)(e - b);
            const int lb = (int)(std::lower_bound(rows, rows + nr, c0) - rows);
            const int ub = (int)(std::lower_bound(rows, rows + nr, c1) - rows) - 1;
            if (ub < lb) throw std::runtime_error("inspector: descendant without overlap rows");
            sn.push_back(d);
            lbv.push_back(lb);
            ubv.push_back(ub);
        }
        ptr[t + 1] = (int64_t)sn.size();
    }
}

PatternRef pattern_ref(const Symbolic& S) {
    PatternRef R;
    R.n = S.n;
    R.nsuper = S.nsuper;
    R.super = S.super.data();
    R.col2sup = S.col2sup.data();
    R.sparent = S.sparent.data();
    R.i_ptr = S.i_ptr.data();
    R.s = S.s.data();
    R.A1p = S.A1.p.data();
    R.A1i = S.A1.i.data();
    return R;
}

// The reference builds A1 with a permuting transpose of the lower triangle
// (examples/choleskyTest01.cpp:190 -> cholesky/Transpose.h:554), which leaves the
// rows of a column in traversal order, not sorted: first the rows r whose
// original index precedes the column's (ascending r), then the diagonal, then the
// remaining rows by ascending original index.  ereach_sn discovers its paths in
// that order, so the update order of the numeric phase depends on it.
static void reference_entry_order(CscPattern& up, const std::vector<int>& perm) {
    const bool hasx = !up.x.empty(), hassrc = !up.src.empty();
    std::vector<int> idx;
    std::vector<int> ti, ts;
    std::vector<double> tx;
    for (int c = 0; c < up.n; ++c) {
        const int b = up.p[c], e = up.p[c + 1];
        idx.clear();
        for (int q = b; q < e; ++q)
            if (up.i[q] < c && perm[up.i[q]] < perm[c]) idx.push_back(q);
        for (int q = b; q < e; ++q)
            if (up.i[q] == c) idx.push_back(q);
        const size_t g2 = idx.size();
        for (int q = b; q < e; ++q)
            if (up.i[q] < c && perm[up.i[q]] > perm[c]) idx.push_back(q);
        std::sort(idx.begin() + g2, idx.end(),
                  [&](int a, int bq) { return perm[up.i[a]] < perm[up.i[bq]]; });
        ti.clear(); ts.clear(); tx.clear();
        for (int q : idx) {
            ti.push_back(up.i[q]);
            if (hasx) tx.push_back(up.x[q]);
            if (hassrc) ts.push_back(up.src[q]);
        }
        for (size_t t = 0; t < idx.size(); ++t) {
            up.i[b + t] = ti[t];
            if (hasx) up.x[b + t] = tx[t];
            if (hassrc) up.src[b + t] = ts[t];
        }
    }
}

void analyze(int n, const int* Ap, const int* Ai, const double* Ax, const int* perm,
             const int nrelax[3], const double zrelax[3], Symbolic& S) {
    S = Symbolic();
    S.n = n;
    // pass 1 (reference: analyze_ordering, cholesky/LSparsity.h:167-247)
    CscPattern up, lo;
    permute_sym(n, Ap, Ai, nullptr, perm, up, lo);
    std::vector<int> parent0, post0, cc0, post1;
    etree_upper(up, parent0);
    postorder(parent0, nullptr, post0);
    col_counts(lo, parent0, post0, cc0);
    // weighted postorder folded into the permutation (cholesky/LSparsity.h:675-723)
    postorder(parent0, cc0.data(), post1);
    if ((int)post1.size() != n) throw std::runtime_error("inspector: postorder incomplete");
    S.perm.resize(n);
    S.colcount.resize(n);
    S.parent.resize(n);
    std::vector<int> invpost(n);
    for (int k = 0; k < n; ++k) {
        S.perm[k] = perm ? perm[post1[k]] : post1[k];
        S.colcount[k] = cc0[post1[k]];
        invpost[post1[k]] = k;
    }
    for (int k = 0; k < n; ++k) {
        int op = parent0[post1[k]];
        S.parent[k] = op == -1 ? -1 : invpost[op];
    }
    S.flops_colcount = 0;
    for (int k = 0; k < n; ++k) S.flops_colcount += (double)S.colcount[k] * (double)S.colcount[k];
    // pass 2: permuted matrices (examples/choleskyTest01.cpp:190-191)
    permute_sym(n, Ap, Ai, Ax, S.perm.data(), S.A1, S.A2);
    supernodes(S, nrelax, zrelax);
    reference_entry_order(S.A1, S.perm);
    level_sets(S.sparent, S.levelPtr, S.levelSet);
    build_update_lists(pattern_ref(S), S.upd_ptr, S.upd_sn, S.upd_lb, S.upd_ub);
}

}  // namespace parsy
