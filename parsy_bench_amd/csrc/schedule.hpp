// Device schedule of the supernodal Cholesky and the BCSC forward solve.
//
// Built once per pattern on the host from the reference-shaped symbolic arrays
// (the argument lists of cholesky/parallel_PB_Cholesky_05.h:27-39 and
// triangularSolve/Triangular_BCSC.h:115) and uploaded to HBM; the numeric kernels
// only ever read flat descriptor arrays.
//
// Cholesky, one etree level (wavefront) at a time:
//   SMALL   one workgroup per supernode whose panel (rows x width doubles) fits
//           the LDS budget: assemble, apply every update, POTRF+TRSM, store.
//   TILES   every other supernode is cut into 64x64 tiles of its panel (lower trapezoid
//           only).  The updates from descendants at least two levels below the target are
//           applied EARLY, by a launch on a side stream that runs concurrently with the
//           previous level's chain (one workgroup per tile, FP64 MFMA, one wave per 32x32
//           sub-tile accumulating in LDS).
//   CHAIN   one launch per level finishes those supernodes as a dataflow over their tiles:
//           a workgroup takes the next tile (I,J) in block-column order from a ticket
//           counter, applies the remaining external updates, then -- left-looking inside the
//           supernode -- the updates by block columns k < J as soon as tiles (I,k) and (J,k)
//           are published, then POTRF (I == J) or TRSM against the published diagonal tile,
//           writes the tile and publishes it (one flag per tile, agent-scope release/acquire).
//           Tickets are handed out in start order, so every tile a workgroup waits for belongs
//           to a workgroup that has already started: no residency assumption, no deadlock.
// The solve mirrors it (SOLVE_SMALL: width <= 64; wide supernodes: one chain launch per level,
// or SOLVE_PANEL per block column + one SOLVE_FIXUP when the chain would not be resident).
//
// Subtree launches (the reference's w-partitions: one thread walks the supernodes of a partition in
// order, cholesky/parallel_PB_Cholesky_05.h:66-84, triangularSolve/Triangular_BCSC.h:171-232): the
// bottom of the etree -- whole subtrees made of supernodes that a single workgroup handles (SMALL /
// width <= 64) -- is cut into subtrees of bounded cost, and ONE workgroup walks each of them supernode by
// supernode in index order (descendants first; the backward solve in reverse).  Everything a supernode
// of such a subtree depends on lies in the same subtree, so there is no hand-off and no level barrier
// inside it: one launch replaces the SMALL launches of the narrow levels, in the factorization and in
// both solves.
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

#include "inspector.hpp"

namespace parsy {

constexpr int kTile = 64;             // tile edge of the tiled path / block-column width
constexpr int kSub = 32;              // per-wave sub-tile edge
constexpr int kSmallMaxEntries = 6144; // panel entries the SMALL kernel keeps in LDS (48 KiB)
constexpr int kSmallMaxWidth = 64;
constexpr int kBelowRows = 512;       // backward solve: rows of a k_bsolve_below chunk
constexpr int kBelowMaxGroups = 512;  // ... used by the chain launches of at most this many workgroups (the top of the tree: few,
                                      // tall panels -- a block column's rows below are otherwise streamed by its own four waves only)
constexpr int kBackGroup = 2;         // backward chain, one right-hand side: block columns per workgroup
constexpr int kTinyWidth = 16;        // solves: supernodes this narrow are solved by one wave each (and walked in
                                      // subtrees); kTinyWidth2: second width class of the one-wave kernels
constexpr int kTinyWidth2 = 32;
constexpr int kPanelRows = 128;       // TRSM row chunk per workgroup (staged in LDS)
constexpr int kSolveRows = 256;       // solve row chunk per workgroup
constexpr int kSolveRowsMrhs = 128;   // ... of the many-right-hand-side chain launches (k_solve_blocks_mrhs: 32 rows per wave)

struct SnDesc {       // one per supernode
    int64_t px;       // offset of the panel in lValues
    int64_t pi;       // offset of the row ids in lR
    int64_t upd0;     // first update descriptor
    int32_t c0, w, r; // first column, width, rows (incl. the w diagonal rows)
    int32_t nupd;     // number of update descriptors
    int32_t a0, a1;   // entries [a0,a1) of A2 belong to this supernode's columns
    int32_t dslot;    // first 64x64 slot of the per-block-column scratch (solve: inverse blocks; -1: SMALL)
    int32_t tflag0;   // first tile flag of a tiled supernode: tile (I,J) has flag tflag0 + I*ceil(w/64) + J
    int32_t ld;       // leading dimension of the panel: r, except for a piece of a split supernode
                      // (rows of the whole supernode; the piece is a window of its panel)
    int32_t rbias;    // piece of a split supernode: offset of its first column inside the supernode
                      // (relpos values count from the supernode's first row; panel row = relpos - rbias)
};

struct UpdDesc {      // one per (target, descendant) pair, in the reference's update order
    int64_t src;      // offset in lValues of row lb of the descendant's panel
    int64_t rel;      // offset into relpos: position in the target's row list of
                      // descendant rows lb, lb+1, ...  (the reference's map[lR[..]]);
                      // -1: identity (row lb + k of the source is row k of the target's panel:
                      // the earlier pieces of a split supernode)
    int32_t ld;       // rows of the descendant (leading dimension)
    int32_t K;        // width of the descendant
    int32_t m;        // nSupRs: descendant rows from lb to the end
    int32_t n1;       // ndrow1: of those, rows inside the target's columns
};

struct WaveEntry {    // one (descendant, 32x32 sub-tile) pair of the tile kernel's update stream
    int64_t src;      // offset in lValues of row lb of the descendant's panel (UpdDesc::src)
    int32_t rel;      // offset into relpos of row lb (UpdDesc::rel)
    int32_t ld, K;    // rows and width of the descendant
    int32_t ia, ja;   // first descendant row (counted from lb) inside the sub-tile's row / column window
    int32_t mn;       // rows in the row window | rows in the column window << 8  (1..32 each)
                      // BIG entries (windows of up to 128): | identity row map << 16 | rows of the ROW STRIP behind the
                      // block << 17 | columns of the COLUMN STRIP beside it << 22 (0..16 each: big_strip_rows / _cols)
};
// A dense BIG entry (a full 128 x 128 block of a source's rows) may carry the remainder of its source's row run -- up to
// kStripMax rows right behind the block, times the block's column window -- and / or of its column run: k_chol_dense
// multiplies them with the operands it has staged for the block, plus a strip of 16 rows per chunk.
constexpr int kStripMax = 16;
inline bool big_ident(const WaveEntry& E) { return (E.mn >> 16) & 1; }
inline int big_strip_rows(const WaveEntry& E) { return (E.mn >> 17) & 31; }
inline int big_strip_cols(const WaveEntry& E) { return (E.mn >> 22) & 31; }
constexpr int kDenseStripTask = 1 << 30;   // TileDesc::part of a k_chol_dense task: an entry of it carries a strip

struct TileDesc {     // one workgroup of the TILES / CHAIN kernels
    int32_t sn;
    int32_t row0, col0;   // tile origin inside the panel (multiples of 64, row0 >= col0)
    int32_t part;         // tiles whose early update stream is split over several workgroups:
                          // TILES: part index | number of parts << 8; CHAIN: number of parts (0 / 1: not split)
    int64_t wp;           // wave_ptr[wp + q .. wp + q + 1] = WaveEntry range of wave q (sub-tile rows
                          // 32*(q>>1).., columns 32*(q&1)..): early list (TILES) / late list (CHAIN);
                          // a split TILES task: split_ranges[wp + 2q], [wp + 2q + 1] = its share of wave q's list
    int64_t sp;           // split tiles: offset (doubles) in the tile scratch of the partial tiles of parts 1..
};

struct PanelDesc {    // one workgroup of the PANEL / SOLVE_PANEL kernels
    int32_t sn;
    int32_t jb;       // block column
    int32_t row0;     // first panel row of this workgroup's chunk (>= 64*(jb+1)), or -1:
                      // the workgroup that parks the factored diagonal block / solved x block
    int32_t pad;
};

// Subtree launches of the solves with many right-hand sides (k_solve_sub_mrhs / k_bsolve_sub_mrhs): ONE WAVE walks a
// subtree for 16 right-hand sides and keeps everything the subtree's supernodes hand to each other in LDS -- one
// slot (16 doubles) per column of a path to the subtree's root and per row outside it that a member touches.  Forward:
// the slots accumulate `L21 * y` (reference Triangular_BCSC.h:139-157: x[Li[l]] -= tmp[k]); a member's x block is its
// right-hand side minus its slots, and only the outside slots leave the wave, once, as atomics.  Backward: the slots
// hold x itself (outside rows gathered once).
// A member is a block of at most 16 columns: a supernode, or -- wider supernodes -- one of its 16-column blocks (a
// window of its panel: ld = rows of the supernode), the later blocks standing to the earlier ones as a parent to its
// child.  TIERS: tier 0 = the subtrees of narrow supernodes the bottom of the etree is cut into (the subtree launch of
// the few-right-hand-side kernels), tier k > 0 = a band of levels above, every supernode of it in the tree of its
// highest ancestor inside the band: one launch per tier instead of one or two per level (sub_tiers).
struct SubMember {    // in the order of the walk (index order inside a subtree)
    int64_t px;       // offset in lValues of the block's first diagonal entry
    int32_t c0, w, r; // first column, width (<= kTinyWidth), rows from the diagonal entry down
    int32_t slot0;    // slot of its first column
    int32_t so;       // first 16-row chunk of its rows below in sub_slots
    int32_t ld;       // leading dimension of the panel
};
struct SubTree {      // one per workgroup of a tier's launch
    int32_t m0, m1;   // members [m0, m1) of sub_members
    int32_t ncols;    // slots [0, ncols): the members' columns (a stack)
    int32_t nout;     // slots [ncols, ncols + nout): rows outside, ids sub_out_rows[out0 ..); slot ncols + nout: padding rows
    int32_t out0;
    int32_t pad[3];
};
struct SubTier {
    int32_t tree0, ntrees;   // its trees in sub_trees
    int32_t max_slots;       // ncols + nout + 1 of the largest
    int32_t top_level;       // the band of etree levels it ends with (tier 0: -1 -- not a band)
};
constexpr int kSubMaxSlots = 320;   // subtrees of the solves are cut so that columns + outside rows of the root fit
                                    // (x 17 doubles of LDS per wave)
constexpr int kSubTierMaxWidth = 64;     // bands end below the first supernode wider than this ...
constexpr int kSubTierMinTrees = 512;    // ... and where fewer trees than this would be left (PARSY_SUB_TIER_MIN_TREES)
constexpr int kSubTierMaxSlots = 448;    // ... or a tree would need more slots
constexpr int kSubTierMaxDensity = 700;  // bands only for factors of fewer stored entries per row than this (gate sweep, round 5)

enum LaunchKind : int32_t {
    kLaunchSmall = 0, kLaunchTiles = 1, kLaunchChain = 2, kLaunchBig = 3,
    kLaunchBackBelow = 4,   // backward solve, one right-hand side: the part of a tall wide supernode's sums that comes from
                            // the rows below its own columns, in 512-row chunks over the whole device (k_bsolve_below),
                            // right before the level's chain launch
    kLaunchSolveSmall = 5, kLaunchSolvePanel = 6, kLaunchSolveFixup = 7, kLaunchBackBlock = 8,
    kLaunchDense = 9,   // the dense entries of a BIG launch's tasks (k_chol_dense), right before that launch's ragged rest
};
inline bool is_chol_launch(int kind) { return (kind >= 0 && kind <= kLaunchBig) || kind == kLaunchDense; }

struct Launch {
    int32_t kind;
    int32_t first, count;  // range in the kind's descriptor array
    int32_t level;         // etree level of the targets; side launches: level whose main-stream launches wait for it
    int32_t jb;            // SMALL: stage size; CHAIN: index of its ticket counter; SOLVE_PANEL / BACK: block column
    int32_t lds_bytes;     // dynamic LDS (SMALL); BACK chain launch: first entry of its workgroups in bsolve_pairs;
                           // SOLVE_PANEL chain launch: first entry of its tasks in solve_mtasks
    int32_t fused;         // SOLVE_PANEL / BACK: 1 = chain launch of the whole level; SMALL, SOLVE_SMALL, BACK:
                           // 2 = subtree launch (first / count: (begin, end) pairs in the kind's range array);
                           // CHAIN: 1 = the level's second launch (tiles below the diagonal squares only)
    int32_t side;          // 1: runs on the plan's side stream (TILES), 0: main stream
    int32_t wait_level;    // side launches: wait until this etree level is complete (-1: init only);
                           // BACK chain launch: number of its entries in bsolve_pairs; SOLVE_PANEL chain launch: in solve_mtasks
    int32_t early;         // TILES: always 1 (kept for the launch dumps)
};

constexpr int kSplitChunks = 192;        // early wave streams longer than this (16-wide k chunks) are split ...
constexpr int kSplitTarget = 128;        // ... into parts of about this length (at most kSplitMaxParts)
constexpr int kSplitMaxParts = 4;
constexpr int kSplitFewTiles = 128;      // levels with at most this many tiles: streams longer than kSplitFewChunks are cut into
constexpr int kSplitFewChunks = 32;      // ... parts of about kSplitFewTarget, at most kSplitFewMaxParts (ex15-class 0.563 -> 0.485 ms,
constexpr int kSplitFewTarget = 24;      // mid3d-class 1.00 -> 0.92 ms; on every level the nd24k-class input lost 5 %:
constexpr int kSplitFewMaxParts = 8;     // profiles/r04_small_jobs.txt)
constexpr int kWalkerBatch = 64;          // CHAIN: supernodes whose tiles are interleaved block column by block column
                                         // (upper bound; Schedule::walker_batch follows the device's CU count)
constexpr int kBigTile = 128;             // tile edge of the BIG (LDS-staged GEMM) update kernel
constexpr int kBigMinK = 128;             // updates from descendants at least this wide go through it (PARSY_BIG_MINK)
constexpr int kPieceWidth = 512;          // supernodes wider than 1.5 x this are factored as a chain of pieces
                                          // of this many columns (PARSY_PIECE_WIDTH; 0: never split)
constexpr double kBigAutoFlops = 1e11;    // update flops of a pattern from which the BIG launches are used ...
constexpr double kPieceAutoFlops = 2e12;  // ... and from which the very wide supernodes are cut into pieces
constexpr double kChainSplitAutoFlops = 1e11;   // jobs from this many update flops on: two chain launches per level (square / rows below)
constexpr int kBigSuperMinTasks = 12288;    // BIG launches with at least this many single-tile tasks ...
constexpr double kBigSuperMaxFill = 0.85;   // ... whose windows hold less than this share of their 8 x 8 fragments take 2 x 2 super-tiles
constexpr int kBigGroup = 8;               // BIG launches: edge of the super-tiles whose tasks share an XCD (PARSY_BIG_GROUP)
constexpr int kBigGroupsPerXcd = 2;        // ... used only where every XCD gets at least this many of them
constexpr int kBigTailGroups = 16;         // ... the lightest groups are dealt again at half the edge
constexpr int kDenseChunk = 8;             // k extent of k_chol_dense's chunks (TileDesc::part counts them)
constexpr double kDenseMinShare = 0.25;    // a BIG launch uses k_chol_dense when at least this share of its products is dense
                                           // (PARSY_BIG_DENSE=0: never, 2: wherever there is a dense entry)
constexpr double kDenseMinFill = 0.70;     // in a split launch an entry goes to k_chol_dense when its window holds at least this share of 128 x 128
constexpr double kDenseAllShare = 0.06;    // ... and takes the launch's ragged entries too when they are at most this share of its products
constexpr int kPushGroup = 1;             // pieces whose updates of the pieces further right are merged (PARSY_PUSH_GROUP)
constexpr int kOneMaxSupernodes = 8192;         // plans of at most this many supernodes (outside the subtree launches) -- or twice as many when the supernodes
constexpr int64_t kOneLargeEntries = 4096;      // hold at least this many entries of L on average (a 3-D problem: 64^3 grid, 13 583
                                                // supernodes, 1.12 -> 0.97 ms; a 500 x 500 grid, 26 092 supernodes of 600 entries, loses:
                                                // 0.35 -> 0.46 ms, as does the parabolic_fem-class input) -- and at most
constexpr int64_t kOneMaxEntries = 1 << 28;     // ... this many stored entries solve in ONE launch per direction (Flan-class, forced:
                                                // 4.6 -> 7.1 ms forward, 5.8 -> 6.2 backward) when
constexpr int kOneMaxSupernodesBig = 32768;     // above a subtree launch, one right-hand side at a time: up to this many supernodes outside it
constexpr int kOneSmallBlocks = 1024;     // launches of at most this many blocks take up to kOneMaxRhs right-hand sides, larger ones 4
constexpr int kOneMidBlocks = 4096;       // ... (both: plans of at most this many blocks)
constexpr int kOneRhs8Density = 300;      // forward: up to kOneMaxRhs right-hand sides too where the factor stores fewer entries per row than this
                                          // (400 until the level launches of at most 16 right-hand sides shared their rows: 30^3 7-point, 316 per row,
                                          // 8 right-hand sides 0.407 with the ONE launch, 0.350 without; 96 x 96 x 12 27-point, 351: 0.689 / 0.646)
constexpr int kOneMaxRhs = 8;             // ... the block has at most this many right-hand sides (PARSY_SOLVE_ONE=0: never, 2: always)
constexpr int kCholSubtreesPerCu = 64;    // ... the factorization's: this many
constexpr int kSubtreesPerCu = 16;        // subtree launches: aim at this many subtrees per compute unit ...
constexpr double kSubtreeMinCost = 2e5;   // ... but never cut below this cost (flop equivalents; solves: 1/16 of it)
constexpr int kSubtreeMinPerSlot = 2;     // ... and only where there are this many eligible supernodes per subtree
constexpr int kSolveSubtreeMinMembers = 2048;   // ... the solves: from this many eligible supernodes on

struct Schedule {
    int n = 0, nsuper = 0, nlevels = 0;
    bool solve_only = false;       // built without A / update lists: only the solve launches exist
    int64_t nnzA = 0, ssize = 0, xsize = 0, nnzL = 0;
    int max_width = 0, max_rows = 0, n_small = 0, n_big = 0;
    int n_solve_wide = 0;          // supernodes wider than a tile (solve: block-column chain)
    int64_t n_dslots = 0;          // block columns of the tiled supernodes (64*64 doubles of scratch each)
    int64_t n_tflags = 0;          // tiles of the tiled supernodes (one publication flag each)
    int n_chain_launches = 0;      // CHAIN launches (one ticket counter each)
    int n_solve_chain_launches = 0;  // chain launches of the forward and backward solve (one ticket counter each)
    int walker_batch = kWalkerBatch;  // walkers interleaved per batch: at most a quarter of the resident workgroups
                                      // (2 per CU), so that a small partition of the GPU cannot fill up with walkers
    double flops_stored = 0, update_flops = 0, reread_bytes = 0;
    double tile_update_flops = 0;  // external-update flops of the tiled supernodes (TILES launches)
    double inner_flops = 0;        // in-supernode SYRK/GEMM flops of the tiled path (CHAIN launches)

    std::vector<SnDesc> sn;          // the supernodes of the pattern (solve launches)
    std::vector<int> sparent;        // supernodal etree (-1: root), postordered
    // Cholesky view: the same supernodes, the very wide ones cut into pieces (column ranges) that are
    // factored one after the other like a chain of supernodes -- a piece's panel is a window of the
    // supernode's panel (ld, rbias), the pieces to its left update it like descendants (identity row
    // map).  Levels, update lists, tiles and launches of the factorization are built on this view.
    std::vector<SnDesc> csn;
    std::vector<int32_t> piece0;     // per supernode: its first piece in csn (nsuper + 1 entries)
    std::vector<int32_t> csn_real;   // per piece: its supernode
    std::vector<int> clevelPtr, clevelSet;  // level sets of the chain-extended etree
    int cnlevels = 0;
    int big_min_k = kBigMinK, piece_width = kPieceWidth, push_group = kPushGroup;
    int big_super_r = 0, big_super_c = 0;   // forced super-tile size (PARSY_BIG_SUPER), 0: per launch
    std::vector<UpdDesc> upd;        // update descriptors of the Cholesky view (per piece)
    std::vector<int32_t> upd_src;    // ... and the piece that completes each one's source (its last piece)
    std::vector<int32_t> relpos;
    std::vector<int64_t> a_dst;     // destination in lValues of every A2 entry
    std::vector<int32_t> rows;      // lR
    std::vector<WaveEntry> wave_entries;  // update streams of the tile kernel, one list per (tile, phase, wave)
    std::vector<int64_t> wave_ptr;        // ... in the order [supernode][J][I][phase][wave], one closing entry each

    // Subtrees walked by one workgroup each (PARSY_SUBTREES=0: none): per supernode its subtree or -1, and
    // the cost estimate the subtrees were cut by (launches start the expensive ones first)
    // (forward solve: subtrees of tiny supernodes, one wave each; backward solve: of all supernodes of one block
    // column, one workgroup each)
    std::vector<int32_t> chol_subtree, solve_subtree, bsolve_subtree;
    std::vector<int32_t> solve_subtree_all, bsolve_subtree_all;   // the subtrees of the whole pattern (build_launches keeps,
                                                                   // under a mask, those the mask contains whole)
    std::vector<double> chol_cost, solve_cost;
    int n_chol_subtrees = 0, n_solve_subtrees = 0, n_bsolve_subtrees = 0;
    // (begin, end) pairs into small_list / solve_small_list / bsolve_blocks, one per workgroup of a subtree launch
    std::vector<int32_t> small_ranges, solve_small_ranges, bsolve_ranges;
    // the solves' subtree launch for many right-hand sides (SubMember / SubTree above); sub_slots: per 16-row chunk of a
    // member's rows below 16 slot numbers, entry 4 kq + v = slot of row 16 chunk + 4 v + kq (the lane order of the
    // matrix cores' result registers)
    std::vector<SubMember> sub_members;
    std::vector<SubTree> sub_trees;
    std::vector<uint16_t> sub_slots;
    std::vector<int32_t> sub_out_rows;
    std::vector<SubTier> sub_tiers;   // (empty: that form of the launches does not exist)
    int sub_cover_level = -1;         // every active supernode of the levels <= this is in a tier: a solve with many right-hand
                                      // sides skips the level launches up to it
    int sub_max_slots = 0;            // slots of the largest tree of any tier

    // Cholesky launch data
    std::vector<int32_t> small_list;
    std::vector<TileDesc> tiles;       // TILES and CHAIN descriptors
    // BIG: updates from wide descendants (K >= big_min_k), one workgroup per 128x128 tile of the target
    // and launch; launches go by the level of the SOURCE: NEXT(s) = targets one level up (main stream,
    // before their chain), PUSH(s) = targets further up (side stream, as soon as level s is complete).
    std::vector<WaveEntry> big_entries;   // per task: (source, row window, column window), sources in update order
    // A task's entries [e0, e1): first its DENSE ones [e0, em) -- full 128 x 128 blocks of a source's rows entirely on
    // or below the target's diagonal, for k_chol_dense (em = e0 where the launch does not use that kernel) -- then the
    // ragged rest [em, e1) for k_chol_big; each part in update order.  dchunks: 8-wide k chunks of the dense part.
    struct BigTask { int32_t sn, row0, col0, weight; int64_t e0, e1; int32_t src_level, next;
                     int32_t sr, sc;      // sr x sc: the super-tile's edge in 128 x 128 tiles
                     int64_t em; int32_t dweight, dchunks;
                     int32_t strips; };   // dense entries of it that carry a strip (WaveEntry::mn)
    std::vector<BigTask> big_all;         // every task, grouped by (src_level, next)
    std::vector<TileDesc> big_tasks;      // tasks of the launches (active targets): k_chol_big: wp, sp = the ragged part [em, e1) of a
                                          // task; k_chol_dense: its dense part [e0, em), part = its 8-wide k chunks
    double big_flops = 0;                 // flops of the BIG launches (dense + ragged entries)
    double dense_flops = 0;               // ... of which through k_chol_dense
    int64_t n_dense_entries = 0;
    int64_t n_strip_entries = 0;          // dense entries that carry a row / column strip
    std::vector<Launch> chol;

    // solve launch data
    std::vector<int32_t> solve_small_list;
    std::vector<PanelDesc> solve_panels;
    std::vector<PanelDesc> solve_mtasks;  // chain launches, many right-hand sides: per wide supernode its block columns
                                          // (row0 = -1), then the 256-row chunks of the rows below its columns (row0 >= w);
                                          // Launch::lds_bytes / wait_level = first entry / number of entries of a launch
    std::vector<int32_t> solve_fix_list;  // wide supernodes solved by per-block-column launches
    std::vector<int32_t> solve_wide_list; // (supernode, block column) pairs of the wide supernodes solved by
                                          // SOLVE_CHAIN: the diagonal blocks whose inverses a solve needs
    std::vector<Launch> solve;

    // backward solve L' x = y: levels from the root down, wide supernodes block column by block column
    std::vector<PanelDesc> bsolve_pairs;   // chain launches, one right-hand side: (supernode, highest block column,
                                           // blocks = 1 .. kBackGroup) per workgroup, from a supernode's last block
                                           // column up
    std::vector<PanelDesc> bsolve_blocks;
    std::vector<PanelDesc> bsolve_below;   // k_bsolve_below tasks: (supernode, block column, first panel row of the chunk,
                                           // pad = slot of its 64 partial sums); the chain's groups name the supernode's
                                           // first slot + 1 in pad (0: the chain streams those rows itself)
    int64_t n_bpart_slots = 0;             // 64 doubles each
    std::vector<Launch> bsolve;

    // ONE-launch solves (k_solve_one, k_bsolve_one): level launches of a few microseconds of work each are a job of
    // launch latencies; instead one workgroup per block column, taken by ticket in level order, and every value handed
    // over as the data itself (a buffer armed with a NaN pattern: the data is the flag).  Forward: block p (sn[p]: <= 64
    // columns of a supernode, a window of its panel) writes what it subtracts from the x of row k below its columns to
    // slot slot0[p] + k - w (one slot per such row, written once), and the block that owns the row gathers its slots:
    // [pull_ptr[p], pull_ptr[p + 1]) of (slot, column of the block).  Backward: the same blocks in reverse order; x itself
    // is handed over (n values per right-hand side).
    // Which supernodes (member): all of them, or -- plans with subtree launches (one_subtrees) -- the ones outside the
    // direction's subtree launch: forward = that launch (one wave per subtree of tiny supernodes: thousands of them are
    // too light for a workgroup each), then the ONE launch for everything above; backward the other way round.  A member's
    // ancestors are members, so every row below a member's columns is owned by a member.
    struct OneLists {
        std::vector<SnDesc> sn;           // the block columns in ticket order (level by level, left to right)
        std::vector<int64_t> slot0;
        std::vector<int32_t> wleft;       // per block: columns of its supernode from the block's first column on (backward:
                                          // the first wleft - w rows below the block are the supernode's later columns)
        int64_t nslots = 0;
        std::vector<int32_t> pull_ptr, pull_slot, pull_pos;
        std::vector<uint8_t> member;      // per supernode
        void clear() { *this = OneLists(); }
    };
    bool solve_one = false, solve_one_back = false;   // forward / backward solve
    bool one_subtrees = false;           // the subtree launches stay: first launch of `solve`, last of `bsolve`
    bool one_big = false;                // taken by the rule for much larger plans: one right-hand side at a time
    bool one_forced = false;             // PARSY_SOLVE_ONE=2: whatever the size, and for every block of <= kOneMaxRhs right-hand sides
    OneLists one_f, one_b;               // (one_b.sn empty: the backward solve uses one_f -- the same supernodes)
    const OneLists& one_back() const { return one_b.sn.empty() ? one_f : one_b; }

    std::vector<uint8_t> active;       // per supernode, 1 = processed by the launches (solves)
    std::vector<uint8_t> active_piece; // per piece of the Cholesky view, 1 = factored by the launches
    std::vector<size_t> chol_level_begin;  // cnlevels + 1: first launch of every level's step in `chol`
    std::vector<int> levelPtr, levelSet;  // etree level sets the launches follow
    std::vector<int64_t> split_ranges;    // split tiles: per part 4 x (begin, end) into wave_entries
    std::vector<int64_t> tile_split;      // per (tiled supernode, J, I) as tile_w / 2: index into split_desc or -1
    struct SplitDesc { int64_t ranges; int64_t sp; int32_t nparts; int32_t pad; };
    std::vector<SplitDesc> split_desc;    // ranges: first index in split_ranges (8 per part); sp: scratch offset
    int64_t n_split_doubles = 0;          // tile scratch: (nparts - 1) * 64 * 64 doubles per split tile
    std::vector<int64_t> sn_wp0;       // per supernode: first index into wave_ptr (-1: SMALL)
    std::vector<int64_t> sn_tw0;       // per supernode: first index into tile_w (-1: SMALL)
    std::vector<int32_t> tile_w;       // per (tiled supernode, J, I, phase): 16-wide k chunks of the longest of its
                                       // four wave streams (phase 0 = early: descendants two or more levels below)
    std::vector<int32_t> level_of;     // etree level of every supernode
};

// Build descriptors + launch lists. `active` (nsuper bytes or null = all) restricts
// the LAUNCHES to a subset of supernodes (multi-GPU shards); descriptors always
// cover the whole pattern.
void build_schedule(const PatternRef& P, const size_t* lC, const int* A2p, const int* A2i,
                    const uint8_t* active, Schedule& out, int compute_units = 0);
// Recompute only the launch lists for a new active set.
void build_launches(Schedule& S, const uint8_t* active, const uint8_t* active_pieces = nullptr);
// Dry run of the CHAIN launches' hand-off protocol with `slots` resident workgroups (tickets in start
// order, a workgroup leaves when its tile is published, a walker when its supernode is done):
// returns the number of tiles that are never finished (0 = the schedule cannot deadlock at that
// residency).  Host only; used by the tests and available to callers that want to check a plan.
int64_t simulate_chain(const Schedule& S, int slots);
// Host-side consistency check of the Cholesky view: levels of sources and targets, windows of the wave and BIG
// entries, exact cover of every update by its entries (flop identity), order of the launch sequence -- and of the
// launches of both solves (every active supernode / chunk / block column exactly once, subtree runs in order,
// width classes, the backward chain's groups).  Returns the number of violations (0 = consistent) and describes
// the first one in `what`.
int64_t check_schedule(const Schedule& S, std::string& what);
constexpr int kSmallMaxUpdates = 8;   // supernodes with more updates go the tiled way (MFMA streams, parallel tiles)
inline bool is_small(const SnDesc& d) {
    return d.w <= kSmallMaxWidth && (int64_t)d.w * d.r <= kSmallMaxEntries && d.nupd <= kSmallMaxUpdates;
}

}  // namespace parsy
