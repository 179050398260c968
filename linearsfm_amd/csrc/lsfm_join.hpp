// Shared by the Stereo and the Mono join (lsfm_join.hip, lsfm_join_mono.hip).
#pragma once
#include "lsfm_internal.hpp"

namespace lsfm {

struct JGroup {
	int F0E, nE, F0C, nC; // feature ranges of End / Cur in the input batch (nC = 0: carried map)
	int FY0;              // first joint feature
	int rC0;              // rank offset of Cur's unmatched features
};

// state of a Stereo join between its two steps (lsfm_join.hip)
struct JoinState {
	size_t smark = 0;
	Arena* ar = nullptr; // where the joint map lives
	int *newf = nullptr, *lenE = nullptr, *srcf = nullptr, *wbase = nullptr;
	double *eP = nullptr, *eF = nullptr;
	std::vector<unsigned char> seg_active, act_padded;
	unsigned char* d_act = nullptr; // device copy of seg_active
	std::vector<int> seg_rows;
	bool fuse_rhs = false;                    // the W part of the right-hand sides goes with the Schur assembly (RhsFused)
	int *srcE = nullptr, *srcC = nullptr;     // per joint feature its sources in the input batch (-1: none)
};
void join_stereo_prepare(lsfm_context* ctx, Arena& ar, const DevBatch& in, DevBatch& out, JoinState& st);
void join_stereo_finish(lsfm_context* ctx, const DevBatch& in, DevBatch& out, JoinState& st, double* eP_out, double* eF_out);

// match[f] = feature of the pair's first map with the same label (-1 none), unm[f] = 1 for unmatched features of the
// second map (unm[NF] = 0)
void join_match_features(lsfm_context* ctx, const DevBatch& in, int* match, int* unm);
__global__ void k_gather_at(const int* __restrict__ src, const int* __restrict__ idx, int n, int* __restrict__ out);
__global__ void k_join_features(int NF, const int* __restrict__ feat_map, const int* __restrict__ feat_id, const double* __restrict__ feat,
                                const double* __restrict__ V, const int* __restrict__ fptr, const int* __restrict__ match,
                                const int* __restrict__ R, const JGroup* __restrict__ grp, int* __restrict__ newf, int* __restrict__ lenE,
                                int* __restrict__ lenC, double* __restrict__ Vy, double* __restrict__ eF, int* __restrict__ fid_y,
                                double* __restrict__ feat_y, int* __restrict__ srcE, int* __restrict__ srcC, int side);
__global__ void k_add_lens(int n, const int* __restrict__ a, const int* __restrict__ b, int* __restrict__ out);

} // namespace lsfm
