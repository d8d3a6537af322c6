// Distribution of ONE factorization over the devices of a node (host logic; no HIP here).
//
// Below a cut, whole etree subtrees go to one rank each: disjoint subtrees are independent in left-looking
// Cholesky -- a target only reads its descendants (reference common/Reach.h:122-135), the independence the
// reference's w-partitions rest on (cholesky/InspectionLevel_06.h:196-217); they are packed onto the ranks
// heaviest first (the reference's bin-packing idea, common/TreeUtils.h:217-255).  Above the cut the PIECES of
// the Cholesky view (supernodes, the very wide separators cut into column ranges) are dealt over the ranks
// level by level: the owner of a piece applies every update INTO it (its BIG tasks, its wave streams, its
// chain), so all ranks work on the top separators at once -- each on its own target pieces -- and what travels
// is a finished piece: after a level is complete, every piece of that level is sent to the ranks that own a
// target it updates (fan-out; only the rows those targets read).  The factor each rank computes for its
// pieces is bitwise the single-device one: every target receives the same updates in the same order from the
// same kernels.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "schedule.hpp"

namespace parsy {

// Default `block`: two consecutive pieces of a split supernode per rank -- every other hand-off on the chain of the top
// separator's pieces stays on one device (model on measured launch times, Flan-class, N = 8: 85 ms against 111 ms
// with block 1 at the same balance; tools/mg_model.py, profiles/r03_mg_model_flan*.txt).
constexpr int kDistBlock = 2;

struct DistMessage {               // what rank `src` sends to rank `dst` after level `level`
    int32_t level = 0, src = 0, dst = 0;
    std::vector<int64_t> off;      // segments of lValues (offset, length): rows [first needed, end) of one
    std::vector<int32_t> len;      // panel column each
    std::vector<int64_t> packed;   // their offsets in the packed buffer (prefix sums of len)
    int64_t total = 0;             // elements of the packed buffer
};

struct Dist {
    int nranks = 1, nlevels = 0, npieces = 0;
    std::vector<int32_t> owner;        // per piece of the Cholesky view
    std::vector<double> cost;          // per piece: flops of the updates into it + its own factorization
    std::vector<double> rank_cost;     // per rank
    std::vector<double> level_cost;    // nlevels * nranks: cost of every rank's pieces on a level
    std::vector<uint8_t> in_subtree;   // per piece: 1 = part of a subtree that went to one rank as a whole
    int n_subtrees = 0, n_root_pieces = 0;
    double total_cost = 0, root_cost = 0;
    int64_t exchange_elements = 0;     // sum over all messages
    std::vector<DistMessage> msgs;     // sorted by (level, src, dst)
    std::vector<int64_t> level_msg0;   // nlevels + 1: first message of every level
};

// Build the distribution of schedule S (any active set: ownership covers every piece) over nranks ranks.
// block: consecutive pieces of one split supernode that stay on one rank (>= 1).
void build_dist(const Schedule& S, int nranks, int block, Dist& D);
// Consistency: every update whose source and target are on different ranks finds the source rows it reads in a
// message delivered right after the source's level; subtrees walked by one workgroup are not split.
int64_t check_dist(const Schedule& S, const Dist& D, std::string& what);

}  // namespace parsy
