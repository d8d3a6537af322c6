// Graph nested dissection (see ordering.hpp).  Host only.
#include "ordering.hpp"

#include <algorithm>
#include <cstdint>
#include <stdexcept>

namespace parsy {

namespace {

struct Graph {
    std::vector<int64_t> xadj;
    std::vector<int> adj;
};

// breadth-first level structure of the vertices with tag[v] == id, from `root`; returns the levels as
// ranges of `order` (level l = order[lptr[l] .. lptr[l+1])); touches only reachable vertices
void bfs_levels(const Graph& G, const std::vector<int>& tag, int id, int root, std::vector<int>& seen, int stamp,
                std::vector<int>& order, std::vector<int>& lptr) {
    order.clear();
    lptr.clear();
    order.push_back(root);
    seen[root] = stamp;
    lptr.push_back(0);
    size_t head = 0;
    while (head < order.size()) {
        const size_t end = order.size();
        lptr.push_back((int)end);
        for (; head < end; ++head) {
            const int v = order[head];
            for (int64_t q = G.xadj[v]; q < G.xadj[v + 1]; ++q) {
                const int u = G.adj[q];
                if (tag[u] == id && seen[u] != stamp) {
                    seen[u] = stamp;
                    order.push_back(u);
                }
            }
        }
    }
    // lptr has one entry too many when the last sweep added nothing
    while (lptr.size() >= 2 && lptr.back() == lptr[lptr.size() - 2]) lptr.pop_back();
    if (lptr.back() != (int)order.size()) lptr.push_back((int)order.size());
}

}  // namespace

void order_nested_dissection(int n, const int* Ap, const int* Ai, int leaf, std::vector<int>& perm) {
    if (n < 0 || !Ap || !Ai) throw std::invalid_argument("order_nested_dissection: bad arguments");
    if (leaf < 1) leaf = 64;
    perm.assign((size_t)n, -1);
    // symmetric adjacency without the diagonal
    Graph G;
    G.xadj.assign((size_t)n + 1, 0);
    for (int j = 0; j < n; ++j)
        for (int q = Ap[j]; q < Ap[j + 1]; ++q) {
            const int i = Ai[q];
            if (i < 0 || i >= n) throw std::invalid_argument("order_nested_dissection: row index out of range");
            if (i != j) {
                G.xadj[i + 1]++;
                G.xadj[j + 1]++;
            }
        }
    for (int v = 0; v < n; ++v) G.xadj[v + 1] += G.xadj[v];
    G.adj.resize((size_t)G.xadj[n]);
    {
        std::vector<int64_t> pos(G.xadj.begin(), G.xadj.end() - 1);
        for (int j = 0; j < n; ++j)
            for (int q = Ap[j]; q < Ap[j + 1]; ++q) {
                const int i = Ai[q];
                if (i != j) {
                    G.adj[(size_t)pos[i]++] = j;
                    G.adj[(size_t)pos[j]++] = i;
                }
            }
    }
    std::vector<int> tag((size_t)n, 0), seen((size_t)n, 0);
    int next_id = 1, stamp = 0;
    struct Piece { std::vector<int> verts; int64_t lo; int id; };  // ordered into positions [lo, lo + |verts|)
    std::vector<Piece> stack;
    {
        Piece all;
        all.verts.resize((size_t)n);
        for (int v = 0; v < n; ++v) all.verts[v] = v;
        all.lo = 0;
        all.id = 0;
        stack.push_back(std::move(all));
    }
    std::vector<int> order, lptr, order2, lptr2;
    while (!stack.empty()) {
        Piece P = std::move(stack.back());
        stack.pop_back();
        if (P.verts.empty()) continue;
        // --- split into connected components (each handled as its own piece, smallest positions first)
        ++stamp;
        bfs_levels(G, tag, P.id, P.verts[0], seen, stamp, order, lptr);
        if (order.size() < P.verts.size()) {
            Piece comp, rest;
            comp.verts = order;
            comp.id = next_id++;
            for (int v : comp.verts) tag[v] = comp.id;
            rest.id = P.id;
            rest.verts.reserve(P.verts.size() - order.size());
            for (int v : P.verts)
                if (tag[v] == P.id) rest.verts.push_back(v);
            comp.lo = P.lo;
            rest.lo = P.lo + (int64_t)comp.verts.size();
            stack.push_back(std::move(rest));
            stack.push_back(std::move(comp));
            continue;
        }
        // --- pseudo-peripheral root: restart from a minimum-degree vertex of the last level while the
        // structure gets deeper
        int root = P.verts[0];
        for (int sweep = 0; sweep < 4; ++sweep) {
            const int nl = (int)lptr.size() - 1;
            int best = order[lptr[nl - 1]];
            int64_t bestdeg = INT64_MAX;
            for (int q = lptr[nl - 1]; q < lptr[nl]; ++q) {
                const int64_t deg = G.xadj[order[q] + 1] - G.xadj[order[q]];
                if (deg < bestdeg) bestdeg = deg, best = order[q];
            }
            ++stamp;
            bfs_levels(G, tag, P.id, best, seen, stamp, order2, lptr2);
            if ((int)lptr2.size() <= (int)lptr.size() && sweep > 0) break;
            const bool deeper = lptr2.size() > lptr.size();
            order.swap(order2);
            lptr.swap(lptr2);
            root = best;
            if (!deeper) break;
        }
        (void)root;
        const int nl = (int)lptr.size() - 1;
        const int64_t np = (int64_t)P.verts.size();
        if (np <= leaf || nl < 3) {
            // --- leaf: reverse Cuthill-McKee = the level structure backwards
            for (int64_t q = 0; q < np; ++q) perm[(size_t)(P.lo + q)] = order[(size_t)(np - 1 - q)];
            continue;
        }
        // --- separator: the level in the middle third (by vertex count) with the fewest vertices
        int lsep = -1;
        int64_t best_size = INT64_MAX;
        for (int l = 1; l + 1 < nl; ++l) {
            const int64_t below = lptr[l], above = np - lptr[l + 1];
            if (below * 3 < np || above * 3 < np) {
                if (below * 5 < np || above * 5 < np) continue;  // too lopsided
            }
            const int64_t size = lptr[l + 1] - lptr[l];
            const int64_t imbalance = std::abs(below - above);
            const int64_t score = size * 4 + imbalance / 8;  // small separators first, balance as a tie-breaker
            if (score < best_size) best_size = score, lsep = l;
        }
        if (lsep < 0) lsep = nl / 2;
        // thin it: a vertex of the level with no neighbour in the next level belongs to the near side
        const int idA = next_id++, idB = next_id++;
        Piece A, B;
        A.id = idA;
        B.id = idB;
        std::vector<int> sep;
        for (int q = 0; q < lptr[lsep]; ++q) A.verts.push_back(order[q]);
        for (int q = lptr[lsep + 1]; q < (int)np; ++q) {
            B.verts.push_back(order[q]);
            tag[order[q]] = idB;
        }
        for (int q = lptr[lsep]; q < lptr[lsep + 1]; ++q) {
            const int v = order[q];
            bool touches = false;
            for (int64_t e = G.xadj[v]; e < G.xadj[v + 1] && !touches; ++e) touches = tag[G.adj[e]] == idB;
            if (touches) sep.push_back(v);
            else A.verts.push_back(v);
        }
        for (int v : A.verts) tag[v] = idA;
        for (int v : sep) tag[v] = -1;  // ordered: out of every later piece
        A.lo = P.lo;
        B.lo = P.lo + (int64_t)A.verts.size();
        const int64_t slo = B.lo + (int64_t)B.verts.size();
        for (size_t q = 0; q < sep.size(); ++q) perm[(size_t)slo + q] = sep[q];
        stack.push_back(std::move(B));
        stack.push_back(std::move(A));
    }
    for (int v = 0; v < n; ++v)
        if (perm[v] < 0) throw std::runtime_error("order_nested_dissection: internal error (position left empty)");
}

}  // namespace parsy
