// Shared between lsfm_solve.hip (Schur assembly, SpMV, back-substitution) and lsfm_pcg.hip (preconditioned CG).
#pragma once
#include "lsfm_internal.hpp"

// features per tile of K9's panel kernel (lsfm_schur_panel.hip); the per-feature fallback k_schur_w keeps tiles of 128 and looks its
// flag up in the panel kernel's tiling
#ifndef LSFM_PM_TILE
#define LSFM_PM_TILE 128
#endif

namespace lsfm {

// What K9's panel kernel works out per tile of 128 features from index arrays alone (which poses see the tile = its slots,
// the slot of every W block, which blocks repeat a (pose, feature) pair): recorded by the first run of a resident tree,
// read by the later ones instead of hashing the tile's poses again.  ns[tile] = -1: no panel variant takes the tile.
struct K9Cache {
	int* ns = nullptr;            // [tiles]
	int* pose = nullptr;          // [tiles * 64]
	unsigned char* eslot = nullptr; // [NW]
	int* wlist = nullptr;         // [3 * tiles] tiles of the 32- / 48- / 64-slot variants, wcnt[v] of them each
	int* wcnt = nullptr;          // [8]: [0..2] tiles in the lists (structure, kept by a plan), [4..6] the cursors of a launch
	int record = 0; // (unused since the slots have a kernel of their own)
};

struct SchurSystem {
	int M = 0, nnzb = 0;
	int *rowptr = nullptr, *colidx = nullptr;      // upper block CSR (diagonal block first in every row)
	const unsigned long long* upper_keys = nullptr; // [nnzb] sorted (row << 32 | col) of the upper pattern
	double* S = nullptr;                           // [nnzb*36]
	double* E = nullptr;                           // [M*6]
	double* IV = nullptr;                          // [NF*9]
	double* LY = nullptr;                          // [NF*9] Cholesky factor of V^-1 (6) and L^T eb (3) per feature, for K9's panel kernel
	int* longrows = nullptr;        // rows with more than SP_LONG blocks (hub poses), *d_nlong of them
	const int* d_nlong = nullptr;
	// cache-resident matrices: both orientations of every block sorted by the row they contribute to (k_spmv_gather);
	// 2 nnzb slots, the holes of the diagonal blocks (~0) at the end; null for a matrix that streams from HBM (k_spmv)
	const unsigned long long* gent = nullptr;
	const int* goth = nullptr;
	const unsigned long long* tab = nullptr; // pose pair -> block of S (open addressing), values in hval
	const int* hval = nullptr;
	unsigned long long mask = 0;
	// K9's sums in fixed point (lsfm_device.hpp "order-independent sums"): acc = one poison word, then [nnzb * 36] accumulators of
	// W V^-1 W^T, then [6 M] high and [6 M] low limbs of the right-hand side's -W V^-1 eb; sexp = [6 M] binary exponents of the
	// pose scalars' scales (2^sexp > 2 sqrt(U_ii)) and, at [6 M], the exponent that bounds |L^T eb| over the level
	long long* acc = nullptr;
	int* sexp = nullptr;
	double* uu = nullptr;    // [NF * 6] fused right-hand side: L^-1 x_f for the End / Cur source of every feature (K9Out::uu)
	double* ymax = nullptr; // [work-groups of k_vinv] largest |L^T eb|^2 of each
	K9Cache k9;      // per-tile structure of K9 (null: every run works it out)
	int k9_tiles = 0, k9_NW = 0;
	double k9_flops = 0; // algorithmic flops of the numeric Schur complement of this system (structure only)
};

// The pattern of a Mono level's camera system from the one below (the Stereo levels have had this since round 3, through the early
// pattern): every pose pair inside one source map is in the level below's pattern -- a joint feature is seen by everything its
// sources were seen by --, the transform's hub links are blocks of the joint U, and what is new are the pairs ACROSS the two sources of
// a matched feature.  Poses are renumbered by the join (Cur's copies of the shared reference / scale pose become End's: pnew), and the
// blocks of the gauge-fixed reference pose are dropped (Imp.cpp:7482-7547, 7619-7700): `dropped`.
struct PatternSeed {
	const unsigned long long* prev_keys = nullptr; // sorted upper pattern of the level below, in the INPUT batch's pose numbering
	int prev_nnzb = 0;
	const int* pnew = nullptr;               // [input poses] -> joint pose
	const unsigned char* dropped = nullptr;  // [input poses] 1: a pose whose U / W blocks the join drops
	int NFY = 0;                             // joint features
	const int *srcE = nullptr, *srcC = nullptr; // [NFY] source features in the input batch (-1: none)
	const int *fptr_in = nullptr, *photo_in = nullptr; // W runs of the input batch
};
void build_schur_pattern(lsfm_context* ctx, const SolveIO& io, SchurSystem& sy);
bool schur_pattern_early_finish(lsfm_context* ctx, const SolveIO& io, SchurSystem& sy);
void schur_pattern_early_extras(lsfm_context* ctx, const SolveIO& io, SchurSystem& sy);
bool schur_pattern_prefetch(lsfm_context* ctx, const DevBatch& Y, const int* d_tref, const unsigned long long* prev_keys, int prev_nnzb, SchurSystem& sy,
                            std::vector<int>* counts = nullptr, bool want_pattern = true, LevelIndex* keep = nullptr);
void build_schur_values(lsfm_context* ctx, const SolveIO& io, SchurSystem& sy);
void schur_vinv(lsfm_context* ctx, const SolveIO& io, SchurSystem& sy);
void build_spmv_index(lsfm_context* ctx, SchurSystem& sy, const unsigned long long* sorted_upper, int* d_flags, int nmir = -1);
void launch_spmv(lsfm_context* ctx, const SchurSystem& sy, const double* x, double* y, const unsigned char* fixed, const double* dotw,
                 const int* pose_seg, double* dot, int dot_stride);
double spmv_bytes(const SchurSystem& sy);
void launch_schur_slots(lsfm_context* ctx, int NF, const int* fptr, const int* photo, unsigned char* fallback, K9Cache kc);
// where K9 adds: the fixed-point accumulators of a SchurSystem and the scales they are in
struct K9Out {
	long long* S = nullptr;      // [nnzb * 36] += W V^-1 W^T in units of 2^(sexp_r + sexp_c - 60)
	long long *Ehi = nullptr, *Elo = nullptr; // [6 M] += -W V^-1 eb, two limbs: units of 2^(sexp + ey - 62) and 2^-40 of that
	long long* poison = nullptr; // != 0: an addend was not finite or outside its bound (the information matrices are not positive semi-definite)
	const int* sexp = nullptr;   // [6 M]
	const int* ey = nullptr;     // -> sexp[6 M]: 2^ey bounds |L^T eb| over the level's features
	// The W part of the join's right-hand sides taken by K9 itself (a pass over W of its own until round 5: k_join_rhs_w): non-null =
	// eb holds the V part only (k_join_features), E starts as U's part of the pose side.  With P = W L of a pass:
	//   y = L^T (eb + W^T x_p) = L^T eb + P^T x_p           (x_p: the estimates of the poses, xpose)
	//   E -= P (y - u),  u = L^-1 x_f                        (x_f: the estimate of the feature in the map the block came from:
	//                                                         uu = [u of its End source | u of its Cur source] per feature, k_vinv;
	//                                                         pside[pose] & 1 says which map a pose -- and so its blocks -- came from)
	// (lsfm_schur_panel.hip has how the panel kernel gets both out of its matrix products); the back-substitution needs no y of its
	// own: x_f = L (L^T eb - L^T sum W^T (x_p - x^_p)), k_backsub.   Imp.cpp:2770-2786, 2822-2838, 2891-2906
	const double* xpose = nullptr;
	const int* pside = nullptr;
	const double* uu = nullptr;
};
// what the fused right-hand side needs beside the joint map (lsfm_join.hip): per joint feature its sources in the level's input,
// their estimates, the poses' estimates and maps
struct RhsFused {
	const int *srcE = nullptr, *srcC = nullptr; // [NF] source feature in the input batch, -1: none
	const double* feat_src = nullptr;           // [input features * 3]
	const double* pose_src = nullptr;           // [M * 6]
	const int* pose_map_src = nullptr;          // [M] map of the input batch a pose belongs to (odd: the second map of its pair)
};
void launch_schur_panel(lsfm_context* ctx, int NF, const int* fptr, const int* photo, const double* W, const double* LY,
                        const unsigned long long* tab, const int* val, unsigned long long mask, K9Out out, unsigned char* fallback,
                        int max_poses_per_system, K9Cache kc, bool fresh_lists = false);
void launch_backsub(lsfm_context* ctx, const SolveIO& io, const SchurSystem& sy, const double* x);

} // namespace lsfm
