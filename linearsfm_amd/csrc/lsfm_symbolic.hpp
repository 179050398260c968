// Host side of K10: ordering + symbolic factorisation of the camera system S of one tree level (all independent systems of
// the level in one block matrix).  Stands where the reference calls cholmod_amd / cholmod_analyze_p in every join
// (pba_solveCholmod{LM,GN}, Imp.cpp:2380-2449 / 7043-7121).  Plain C++ (no HIP): compiled by g++, callable without a
// device -- lsfm_symbolic_analyse (include/lsfm.h) is the test / measurement entry.
#pragma once
#include <vector>

namespace lsfm {

/* most block columns of a supernode group (lsfm_pcg.hip: k_sn_panel keeps 6 * CHOL_GS dense scalar rows in LDS).  8 since round 4 (16
 * before): the panel kernel is left-looking inside a group -- the dot products of a column grow with the columns before it in the group
 * -- while the update between groups runs on the matrix pipes of the whole chip; measured on the NC3500-like set (factorisation +
 * refinement per tree): 16 columns 10.8 ms, 12 10.2, 8 9.4, 6 9.5, 4 9.9; RS468-like 5.76 -> 5.05, synth-16k 141 -> 131.
 * LSFM_GS=<n <= CHOL_GS> narrows it further (measurements). */
#define CHOL_GS 8

struct CholSymbolic {
	int M = 0, nnzL = 0, nlevels = 0, tail_begin = 0;
	// block CSC of L (row indices ascending inside a column, diagonal first), ordering
	std::vector<int> colptr, rowidx, perm, pinv, order, level_ptr;
	// tasks: connected pieces of the elimination tree that one work-group walks (leaf sub-trees; chains above them)
	std::vector<int> task_cols, task_ptr, col_task, col_lpos, col_nin;
	std::vector<int> tlevel_ptr, tlevel_maxsize, tlevel_col0, tlevel_nsmall, tlevel_small_lds, tlevel_outer;
	// supernode groups over the columns above the leaf tasks, ordered by group level
	int ngroups = 0;
	std::vector<int> grp_c0, grp_s, grp_nr, glevel_ptr, glevel_maxnr, glevel_maxs;
	// for the debug census
	std::vector<int> parent, ccount;
	int task_x = 0;
	// Distributed factorisation (block_maps > 0): the poses whose separator level lies inside one block of block_maps consecutive
	// local maps are INTERIOR to that block -- their columns, and everything those columns update below the inter-block
	// separators, belong to the rank that owns the block; the columns of the inter-block separators (the last columns of the
	// ordering, from first_shared on) are shared.  col_owner[j] (new numbering): block of column j, -1: shared.  Empty when off.
	std::vector<int> col_owner;
	int first_shared = 0;
	std::vector<char> glevel_owned, glevel_shared; // per group level: holds groups of interior / of shared columns
	// block products of the numeric factorisation (sum over the columns of c (c + 1) / 2, c = blocks of the column): all of them / those of
	// the shared columns, which every rank repeats -- the replicated share of a distributed factorisation
	double work_total = 0, work_shared = 0;
};

// keys[nnzb]: sorted (row << 32 | col) of the upper block pattern of S (every diagonal block present); origin[M]: index of
// the local map that brought each pose (drives the nested dissection along the join tree).
// block_maps > 0: also the ownership of a factorisation distributed over the ranks that joined blocks of block_maps local maps each.
void chol_symbolic(const unsigned long long* keys, int nnzb, const int* origin, int M, CholSymbolic& out, int block_maps = 0);

} // namespace lsfm
