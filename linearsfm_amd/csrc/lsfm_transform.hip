// Batched coordinate transform of maps: new state x' = f(x) and new information I' = J^T I J.
// Replaces lmj_Transform_PF3DStereo (Imp.cpp:349-1924) and lmj_Transform_PF3DMono (Imp.cpp:3173-6509) for all maps
// of a tree level in one set of launches.
//
// Formulation (DESIGN.md "Transform"): the Jacobian of the OLD state w.r.t. the NEW state is
//     J = blkdiag(D) + sum_s C_s e_{h_s}^T          D = "J1", C_1 = "J2" (column of the old reference pose h_1),
//                                                   Mono: C_2 = "J3" (column of the old scale pose h_2)
// so  I' = D^T I D  +  [D^T G_s] e_{h_s}^T + e_{h_s} [D^T G_s]^T  +  e_{h_s} (C_s^T G_t) e_{h_t}^T ,  G_s = I C_s .
// The reference accumulates the same sums block by block ("Algorithm Line 3/4/5/6").  Here:
//   k_tr_feat_pre / k_tr_entries / k_tr_feat_post   V' = D_f^T V D_f, W' = D_k^T W D_f, the feature rows of G, the
//                  pose rows of G (LDS table per work-group), C_f^T G_f  -- one lane per W block for the bulk
//   k_tr_ublocks   one lane per U block: U' = D_a^T U D_b and the U part of the pose rows of G
//   k_tr_poseslots one lane per pose: the new (k,h_s) blocks D_k^T G_s,k and C_k^T G_k
//   k_tr_diag      adds sum_a C_a^T G_a to the (h,h) blocks
// Output slot layout is the reference's: per map first the m blocks (k,h_1) [Mono: then m blocks (k,h_2)], then the
// untouched blocks in their old order; per feature first its block(s) to h_1 [h_2], then its old blocks.
#include <algorithm>
#include <climits>

#include "lsfm_device.hpp"
#include "lsfm_internal.hpp"

namespace lsfm {

struct TMap {
	int active, nh;
	int P0, m, F0, n;
	int U0, U0n, W0, W0n, kU0, kW0;
	int hub[2];        // global pose index of the old reference pose / old scale pose (= the dense columns)
	int nref, nscap;   // Mono: global index of the NEW reference / scale pose
	int newFix, oldFix, newLabel;
	int c2fix, c3zero; // Mono gauge special cases, Imp.cpp:3703-3710
	int tref, tscap, oldref, oldscap;
	// phase 1: rotation / translation / scale that map old coordinates to new ones
	double R1[9], t1[3], scale1;
	int sign1;
	// phase 2: quantities of the Jacobian, evaluated at the new state of the old reference pose
	double R[9], dRA[9], dRB[9], dRG[9], t[3], dA[3], dB[3], dG[3];
	double Scale, Scale2, dSdA[3], dSdB[3], dSdG[3], dSdt[9], dSdtt[9];
};

__global__ void k_tr_find(const int* __restrict__ pose_id, const int* __restrict__ pose_map, int M, TMap* tm)
{
	int k = blockIdx.x * blockDim.x + threadIdx.x;
	if (k >= M) return;
	TMap& t = tm[pose_map[k]];
	if (!t.active) return;
	int id = pose_id[k];
	if (id == t.tref) t.nref = k;
	if (t.nh == 2)
	{
		if (id == t.tscap) t.nscap = k;
		if (id == t.oldref) t.hub[0] = k;
		if (id == t.oldscap) t.hub[1] = k;
	}
	else if (id == t.tref) t.hub[0] = k; // Stereo: the slot of the new reference pose holds the old one afterwards
}

// Imp.cpp:389-400 / 3216-3244
__global__ void k_tr_params1(const double* __restrict__ pose, TMap* tm, int B, int* err)
{
	int b = blockIdx.x * blockDim.x + threadIdx.x;
	if (b >= B) return;
	TMap& t = tm[b];
	if (!t.active) return;
	if (t.nref < 0 || t.hub[0] < 0 || (t.nh == 2 && (t.nscap < 0 || t.hub[1] < 0))) { atomicExch(err, 1 + b); t.active = -1; return; }
	const double* p = pose + (size_t)t.nref * 6;
	t.t1[0] = p[0]; t.t1[1] = p[1]; t.t1[2] = p[2];
	rmat_ypr(t.R1, p[3], p[4], p[5]);
	t.scale1 = 1.0; t.sign1 = 1;
	if (t.nh == 2)
	{
		const double* q = pose + (size_t)t.nscap * 6;
		double d[3] = { q[0] - p[0], q[1] - p[1], q[2] - p[2] }, ts[3];
		mv3(t.R1, d, ts);
		t.scale1 = fabs(ts[t.newFix]);
		t.sign1 = ts[t.newFix] >= 0 ? 1 : -1;
		t.c2fix = (t.hub[0] == t.nscap);
		t.c3zero = (t.hub[1] == t.nref);
	}
}

// new pose values, Imp.cpp:421-446 / 3268-3297; also copies ids (Stereo relabels the hub slot, Imp.cpp:416-417)
__global__ void k_tr_new_poses(const double* __restrict__ pose, const int* __restrict__ pose_id, const int* __restrict__ pose_map,
                               int M, const TMap* __restrict__ tm, double* __restrict__ npose, int* __restrict__ npose_id)
{
	int k = blockIdx.x * blockDim.x + threadIdx.x;
	if (k >= M) return;
	const TMap& t = tm[pose_map[k]];
	const double* p = pose + (size_t)k * 6;
	double* o = npose + (size_t)k * 6;
	if (t.active <= 0)
	{
		for (int i = 0; i < 6; i++) o[i] = p[i];
		npose_id[k] = pose_id[k];
		return;
	}
	double R2[9], R3[9], a, b, g;
	if (t.nh == 1 && k == t.hub[0])
	{
		o[0] = -(t.R1[0] * t.t1[0] + t.R1[1] * t.t1[1] + t.R1[2] * t.t1[2]);
		o[1] = -(t.R1[3] * t.t1[0] + t.R1[4] * t.t1[1] + t.R1[5] * t.t1[2]);
		o[2] = -(t.R1[6] * t.t1[0] + t.R1[7] * t.t1[1] + t.R1[8] * t.t1[2]);
		inv_rmat_ypr_T(t.R1, a, b, g);
		o[3] = a; o[4] = b; o[5] = g;
		npose_id[k] = t.newLabel;
		return;
	}
	double d[3] = { p[0] - t.t1[0], p[1] - t.t1[1], p[2] - t.t1[2] }, v[3];
	mv3(t.R1, d, v);
	if (t.nh == 2) { v[0] = v[0] / t.scale1; v[1] = v[1] / t.scale1; v[2] = v[2] / t.scale1; }
	rmat_ypr(R2, p[3], p[4], p[5]);
	times_rrt(R3, R2, t.R1);
	inv_rmat_ypr(R3, a, b, g);
	o[0] = v[0]; o[1] = v[1]; o[2] = v[2]; o[3] = a; o[4] = b; o[5] = g;
	if (t.nh == 2)
	{
		if (k == t.nref) for (int i = 0; i < 6; i++) o[i] = 0.0;
		if (k == t.nscap) o[t.newFix] = (double)t.sign1;
	}
	npose_id[k] = pose_id[k];
}

// Imp.cpp:459-471 / 3311-3365
// hub_out (optional): hub pose of every map for the early pattern of S (lsfm_solve.hip): the dense column a transformed map gets,
// -1: passed through
__global__ void k_tr_params2(const double* __restrict__ npose, TMap* tm, int B, int* __restrict__ hub_out)
{
	int b = blockIdx.x * blockDim.x + threadIdx.x;
	if (b >= B) return;
	TMap& t = tm[b];
	if (hub_out) hub_out[b] = t.active > 0 ? t.hub[0] : -1;
	if (t.active <= 0) return;
	const double* p = npose + (size_t)t.hub[0] * 6;
	t.t[0] = p[0]; t.t[1] = p[1]; t.t[2] = p[2];
	r_derivation(p[3], p[4], p[5], t.R, t.dRA, t.dRB, t.dRG);
	ypr_rates<true>(t.dA, t.dRA, t.R); ypr_rates<true>(t.dB, t.dRB, t.R); ypr_rates<true>(t.dG, t.dRG, t.R);
	t.Scale = 1.0; t.Scale2 = 1.0;
	if (t.nh == 2)
	{
		const double* q = npose + (size_t)t.hub[1] * 6;
		double d[3] = { q[0] - p[0], q[1] - p[1], q[2] - p[2] }, ts[3];
		mv3(t.R, d, ts);
		t.Scale = fabs(ts[t.oldFix]);
		t.Scale2 = t.Scale * t.Scale;
		double Sign = ts[t.oldFix] >= 0 ? 1.0 : -1.0;
		for (int i = 0; i < 9; i++) { t.dSdt[i] = -t.R[i] * Sign; t.dSdtt[i] = t.R[i] * Sign; }
		mv3(t.dRA, d, t.dSdA); mv3(t.dRB, d, t.dSdB); mv3(t.dRG, d, t.dSdG);
		for (int i = 0; i < 3; i++) { t.dSdA[i] *= Sign; t.dSdB[i] *= Sign; t.dSdG[i] *= Sign; }
	}
}

// translation part of the Mono Jacobians for an element at new position xn (Imp.cpp:3420-3469 / 3591-3640):
// dt2dt22 = R/Scale (-> D), [dt2dt | tmp1 tmp2 tmp3] (-> C_1), dt2dtt (-> C_2)
__device__ __forceinline__ void mono_trans_jac(const TMap& t, const double* xn, double* dt2dt22, double* dt2dt, double* tmpc, double* dt2dtt)
{
	double t222[3] = { xn[0] - t.t[0], xn[1] - t.t[1], xn[2] - t.t[2] }, t22[3], v[3];
	const int mFix = t.oldFix;
	mv3(t.R, t222, t22);
	mv3(t.dRA, t222, v);
#pragma unroll
	for (int r = 0; r < 3; r++) tmpc[3 * r + 0] = (v[r] * t.Scale - t22[r] * t.dSdA[mFix]) / t.Scale2;
	mv3(t.dRB, t222, v);
#pragma unroll
	for (int r = 0; r < 3; r++) tmpc[3 * r + 1] = (v[r] * t.Scale - t22[r] * t.dSdB[mFix]) / t.Scale2;
	mv3(t.dRG, t222, v);
#pragma unroll
	for (int r = 0; r < 3; r++) tmpc[3 * r + 2] = (v[r] * t.Scale - t22[r] * t.dSdG[mFix]) / t.Scale2;
#pragma unroll
	for (int r = 0; r < 3; r++)
#pragma unroll
		for (int c = 0; c < 3; c++)
		{
			dt2dt22[3 * r + c] = t.R[3 * r + c] / t.Scale;
			dt2dt[3 * r + c] = (-t.R[3 * r + c] * t.Scale - t22[r] * t.dSdt[3 * mFix + c]) / t.Scale2;
			dt2dtt[3 * r + c] = (-t22[r] * t.dSdtt[3 * mFix + c]) / t.Scale2;
		}
}

// per pose: D (6x6) and C_s (6x6), Imp.cpp:485-635 / 3383-3584 and the gauge zeroing 3691-3710
template <int NH>
__global__ void __launch_bounds__(128, 1)
k_tr_pose_jac(const double* __restrict__ npose, const int* __restrict__ pose_map, int M, const TMap* __restrict__ tm,
                              double* __restrict__ Dp, double* __restrict__ Cp)
{
	int k = blockIdx.x * blockDim.x + threadIdx.x;
	if (k >= M) return;
	const TMap& t = tm[pose_map[k]];
	if (t.active <= 0 || t.nh != NH) return;
	double D[36], C1[36], C2[36];
	zero<36>(D); zero<36>(C1); zero<36>(C2);
	const double* p = npose + (size_t)k * 6;
	if (NH == 1 && k == t.hub[0])
	{
		double tmp1[3], tmp2[3], tmp3[3];
		mv3(t.dRA, t.t, tmp1); mv3(t.dRB, t.t, tmp2); mv3(t.dRG, t.t, tmp3);
#pragma unroll
		for (int r = 0; r < 3; r++)
		{
			D[6 * r + 0] = -t.R[3 * r]; D[6 * r + 1] = -t.R[3 * r + 1]; D[6 * r + 2] = -t.R[3 * r + 2];
			D[6 * r + 3] = -tmp1[r]; D[6 * r + 4] = -tmp2[r]; D[6 * r + 5] = -tmp3[r];
			D[6 * (3 + r) + 3] = t.dA[r]; D[6 * (3 + r) + 4] = t.dB[r]; D[6 * (3 + r) + 5] = t.dG[r];
		}
	}
	else
	{
		double R2[9], dRA2[9], dRB2[9], dRG2[9], Ri[9], dRi[9], ddA2[3], ddB2[3], ddG2[3], ddA[3], ddB[3], ddG[3];
		r_derivation(p[3], p[4], p[5], R2, dRA2, dRB2, dRG2);
		times_rrt(Ri, R2, t.R);
		times_rrt(dRi, dRA2, t.R); ypr_rates<false>(ddA2, dRi, Ri);
		times_rrt(dRi, dRB2, t.R); ypr_rates<false>(ddB2, dRi, Ri);
		times_rrt(dRi, dRG2, t.R); ypr_rates<false>(ddG2, dRi, Ri);
		times_rrt(dRi, R2, t.dRA); ypr_rates<false>(ddA, dRi, Ri);
		times_rrt(dRi, R2, t.dRB); ypr_rates<false>(ddB, dRi, Ri);
		times_rrt(dRi, R2, t.dRG); ypr_rates<false>(ddG, dRi, Ri);
		if (NH == 1)
		{
			double d[3] = { p[0] - t.t[0], p[1] - t.t[1], p[2] - t.t[2] }, tmp1[3], tmp2[3], tmp3[3];
			mv3(t.dRA, d, tmp1); mv3(t.dRB, d, tmp2); mv3(t.dRG, d, tmp3);
#pragma unroll
			for (int r = 0; r < 3; r++)
			{
				C1[6 * r + 0] = -t.R[3 * r]; C1[6 * r + 1] = -t.R[3 * r + 1]; C1[6 * r + 2] = -t.R[3 * r + 2];
				C1[6 * r + 3] = tmp1[r]; C1[6 * r + 4] = tmp2[r]; C1[6 * r + 5] = tmp3[r];
				C1[6 * (3 + r) + 3] = ddA[r]; C1[6 * (3 + r) + 4] = ddB[r]; C1[6 * (3 + r) + 5] = ddG[r];
				D[6 * r + 0] = t.R[3 * r]; D[6 * r + 1] = t.R[3 * r + 1]; D[6 * r + 2] = t.R[3 * r + 2];
				D[6 * (3 + r) + 3] = ddA2[r]; D[6 * (3 + r) + 4] = ddB2[r]; D[6 * (3 + r) + 5] = ddG2[r];
			}
		}
		else
		{
			double a22[9], adt[9], atmp[9], adtt[9];
			mono_trans_jac(t, p, a22, adt, atmp, adtt);
			// the old reference / scale pose collects its C term in D (Imp.cpp:3495-3556 / 3558-3581).  Written with selects on
			// the values, not on the destination: a pointer chosen at run time puts all three blocks into scratch memory
			const bool h0 = k == t.hub[0], h1 = k == t.hub[1];
			double X1[36], X2[9];
			zero<36>(X1);
#pragma unroll
			for (int r = 0; r < 3; r++)
			{
#pragma unroll
				for (int c = 0; c < 3; c++) { D[6 * r + c] += a22[3 * r + c]; X1[6 * r + c] = adt[3 * r + c]; X2[3 * r + c] = adtt[3 * r + c]; }
				D[6 * (3 + r) + 3] += ddA2[r]; D[6 * (3 + r) + 4] += ddB2[r]; D[6 * (3 + r) + 5] += ddG2[r];
				X1[6 * (3 + r) + 3] = ddA[r]; X1[6 * (3 + r) + 4] = ddB[r]; X1[6 * (3 + r) + 5] = ddG[r];
				X1[6 * r + 3] = atmp[3 * r + 0]; X1[6 * r + 4] = atmp[3 * r + 1]; X1[6 * r + 5] = atmp[3 * r + 2];
			}
#pragma unroll
			for (int q = 0; q < 36; q++) { D[q] += h0 ? X1[q] : 0.0; C1[q] += h0 ? 0.0 : X1[q]; }
#pragma unroll
			for (int r = 0; r < 3; r++)
#pragma unroll
				for (int c = 0; c < 3; c++) { D[6 * r + c] += h1 ? X2[3 * r + c] : 0.0; C2[6 * r + c] += h1 ? 0.0 : X2[3 * r + c]; }
			if (k == t.nref) zero<36>(D);
			const bool zD = k == t.nscap, zC = t.c2fix != 0;
#pragma unroll
			for (int r = 0; r < 6; r++)
#pragma unroll
				for (int c = 0; c < 6; c++)
					if (c == t.newFix) { if (zD) D[6 * r + c] = 0.0; if (zC) C1[6 * r + c] = 0.0; }
			if (t.c3zero) zero<36>(C2);
		}
	}
	st<36>(Dp + (size_t)k * 36, D);
	st<36>(Cp + (size_t)k * 36, C1);
	if (NH == 2) st<36>(Cp + (size_t)M * 36 + (size_t)k * 36, C2);
	if (NH == 1)
	{
		// Stereo, every pose but the hub: D_k = [R 0; 0 DD2_k], C_k = [-R T_k; 0 DD_k] with R the map's rotation (which a W
		// block's feature record holds too) -- 27 pose-dependent numbers instead of 72.  k_tr_entries gathers these: the
		// two 288-byte gathers per W block were what its time went with (doubling them doubled it).
		double* pj = Cp + (size_t)M * 36 + (size_t)k * 27;
#pragma unroll
		for (int r = 0; r < 3; r++)
#pragma unroll
			for (int c = 0; c < 3; c++)
			{
				pj[3 * r + c] = D[6 * (3 + r) + 3 + c];
				pj[9 + 3 * r + c] = C1[6 * r + 3 + c];
				pj[18 + 3 * r + c] = C1[6 * (3 + r) + 3 + c];
			}
	}
	else
	{
		// Mono: every one of D_k, C1_k, C2_k is [tl tr; 0 br] (the hub poses' D collects the C terms and keeps that shape; C2 is
		// its top-left block alone): 63 numbers instead of 108
		double* pj = Cp + (size_t)M * 72 + (size_t)k * 63;
#pragma unroll
		for (int r = 0; r < 3; r++)
#pragma unroll
			for (int c = 0; c < 3; c++)
			{
				pj[3 * r + c] = D[6 * r + c]; pj[9 + 3 * r + c] = D[6 * r + 3 + c]; pj[18 + 3 * r + c] = D[6 * (3 + r) + 3 + c];
				pj[27 + 3 * r + c] = C1[6 * r + c]; pj[36 + 3 * r + c] = C1[6 * r + 3 + c]; pj[45 + 3 * r + c] = C1[6 * (3 + r) + 3 + c];
				pj[54 + 3 * r + c] = C2[6 * r + c];
			}
	}
}

__global__ void k_tr_flags(const int* __restrict__ Ui, const int* __restrict__ Uj, int NU, const int* __restrict__ photo, int NW,
                           const int* __restrict__ pose_map, const TMap* __restrict__ tm, int* __restrict__ keepU, int* __restrict__ keepW)
{
	int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i < NU)
	{
		const TMap& t = tm[pose_map[Ui[i]]];
		int a = Ui[i], b = Uj[i], keep = 1;
		if (t.active > 0) keep = (a != t.hub[0] && b != t.hub[0] && (t.nh == 1 || (a != t.hub[1] && b != t.hub[1])));
		keepU[i] = keep;
	}
	if (i < NW)
	{
		int k = photo[i];
		const TMap& t = tm[pose_map[k]];
		int keep = 1;
		if (t.active > 0) keep = (k != t.hub[0] && (t.nh == 1 || k != t.hub[1]));
		keepW[i] = keep;
	}
	if (i == 0) { keepU[NU] = 0; keepW[NW] = 0; }
}

// (+ the target-not-found flag and, Mono, the sign of every map's new scale: everything the host reads of a transform that
// analyses, in one copy)
__global__ void k_tr_gather_counts(const int* __restrict__ KU, const int* __restrict__ KW, const int* __restrict__ uoff, const int* __restrict__ woff,
                                   int B, int* __restrict__ out, const TMap* __restrict__ tm, const int* __restrict__ err)
{
	int b = blockIdx.x * blockDim.x + threadIdx.x;
	if (b > B) return;
	out[b] = KU[uoff[b]];
	out[B + 1 + b] = KW[woff[b]];
	if (b < B) out[2 * (B + 1) + b] = tm[b].active > 0 ? tm[b].sign1 : 0;
	else out[2 * (B + 1) + B] = *err;
}

// ---------------------------------------------------------------------------------------------------------
// the feature stage (K3/K4 of SURVEY 2a): Imp.cpp:1300-1915 / 5017-6501, in three launches
//   k_tr_feat_pre   one lane per feature : x', D_f, C_s,f, V' = D_f^T V D_f, new run pointer
//   k_tr_entries    one lane per W block : W' = D_k^T W D_f at its new place, the block's share of the feature rows
//                   of G (W^T C_s,k, summed per feature through LDS) and of the pose rows of G (W C_s,f, LDS table)
//   k_tr_feat_post  one lane per feature : G_s,f complete, the new block(s) to the hub pose(s), C_s^T G_t
// The W blocks -- 90 % of the bytes -- are read and written by consecutive lanes (one 144-byte block per lane,
// contiguous over the wave); a lane per feature walking its run strided the wave's accesses by the run length and
// needed 340 registers (one wave per SIMD).
// ---------------------------------------------------------------------------------------------------------
#define TRE_TILE 128 /* features per work-group: the LDS pose table is flushed once per tile */
#define TRE_ROUND 256 /* W blocks per round = threads */

template <int NH>
__global__ void __launch_bounds__(256)
k_tr_feat_pre(int NF, const TMap* __restrict__ tm, const int* __restrict__ feat_map, const double* __restrict__ feat, const int* __restrict__ fptr,
              const double* __restrict__ Vold, const int* __restrict__ KW, double* __restrict__ nfeat, int* __restrict__ nfptr,
              double* __restrict__ Vn, double* __restrict__ FD, int4* __restrict__ finfo, int* __restrict__ hubJ)
{
	const int f = blockIdx.x * blockDim.x + threadIdx.x;
	if (f >= NF) return;
#pragma unroll
	for (int sidx = 0; sidx < NH; sidx++) hubJ[(size_t)f * NH + sidx] = -1; // (no old block to a hub pose seen yet: k_tr_entries records them)
	const TMap* t = &tm[feat_map[f]];
	const int j0 = fptr[f];
	const double* x = feat + (size_t)f * 3;
	if (!(t->active > 0 && t->nh == NH))
	{
		// pass-through map: plain copy at the new offsets
		nfeat[3 * (size_t)f] = x[0]; nfeat[3 * (size_t)f + 1] = x[1]; nfeat[3 * (size_t)f + 2] = x[2];
		for (int i = 0; i < 9; i++) Vn[(size_t)f * 9 + i] = Vold[(size_t)f * 9 + i];
		nfptr[f] = t->W0n + (j0 - t->W0);
		finfo[f] = make_int4(0, -1, -1, -j0); // new place of block j: (start of the feature's new run) + w + j
		return;
	}
	nfptr[f] = t->W0n + NH * (f - t->F0) + (KW[j0] - t->kW0);
	// per feature record for the block kernel: active, hub pose(s), new place of a kept block j = (start of the
	// feature's new run) + w + KW[j]
	finfo[f] = make_int4(1, t->hub[0], NH == 2 ? t->hub[1] : -1, NH - KW[j0]);
	double Df[9], Cf[NH][18], xn[3];
	// new feature value, Imp.cpp:449-451 / 3300-3302
	double d[3] = { x[0] - t->t1[0], x[1] - t->t1[1], x[2] - t->t1[2] };
	mv3(t->R1, d, xn);
	if (NH == 2) { xn[0] = xn[0] / t->scale1; xn[1] = xn[1] / t->scale1; xn[2] = xn[2] / t->scale1; }
	nfeat[3 * (size_t)f] = xn[0]; nfeat[3 * (size_t)f + 1] = xn[1]; nfeat[3 * (size_t)f + 2] = xn[2];
	// D_f, C_f: Imp.cpp:638-680 / 3587-3684
	if (NH == 1)
	{
		double dd[3] = { xn[0] - t->t[0], xn[1] - t->t[1], xn[2] - t->t[2] }, tmp1[3], tmp2[3], tmp3[3];
		mv3(t->dRA, dd, tmp1); mv3(t->dRB, dd, tmp2); mv3(t->dRG, dd, tmp3);
#pragma unroll
		for (int r = 0; r < 3; r++)
		{
			Df[3 * r] = t->R[3 * r]; Df[3 * r + 1] = t->R[3 * r + 1]; Df[3 * r + 2] = t->R[3 * r + 2];
			Cf[0][6 * r] = -t->R[3 * r]; Cf[0][6 * r + 1] = -t->R[3 * r + 1]; Cf[0][6 * r + 2] = -t->R[3 * r + 2];
			Cf[0][6 * r + 3] = tmp1[r]; Cf[0][6 * r + 4] = tmp2[r]; Cf[0][6 * r + 5] = tmp3[r];
		}
	}
	else
	{
		double adt[9], atmp[9], adtt[9];
		mono_trans_jac(*t, xn, Df, adt, atmp, adtt);
#pragma unroll
		for (int r = 0; r < 3; r++)
		{
#pragma unroll
			for (int c = 0; c < 3; c++) { Cf[0][6 * r + c] = adt[3 * r + c]; Cf[NH - 1][6 * r + c] = adtt[3 * r + c]; Cf[NH - 1][6 * r + 3 + c] = 0.0; }
			Cf[0][6 * r + 3] = atmp[3 * r]; Cf[0][6 * r + 4] = atmp[3 * r + 1]; Cf[0][6 * r + 5] = atmp[3 * r + 2];
		}
		if (t->c2fix)
#pragma unroll
			for (int r = 0; r < 3; r++)
#pragma unroll
				for (int c = 0; c < 6; c++)
					if (c == t->newFix) Cf[0][6 * r + c] = 0.0; // (a run-time index would put Cf into scratch memory)
		if (t->c3zero) zero<18>(Cf[NH - 1]);
	}
	// V' = D_f^T V D_f
	double V[9], T[9], Vnew[9];
	ld<9>(V, Vold + (size_t)f * 9);
	mtm<3, 3, 3, false>(Df, V, T);
	mm<3, 3, 3, false>(T, Df, Vnew);
	st<9>(Vn + (size_t)f * 9, Vnew);
	double* fd = FD + (size_t)f * (9 + 18 * NH);
	st<9>(fd, Df);
#pragma unroll
	for (int s = 0; s < NH; s++) st<18>(fd + 9 + 18 * s, Cf[s]);
}

// Profiling aid (make K9_TIMING=1; tools/tr_phase_times.py): lane 0 of every work-group adds up the shader clocks it spends
// in each phase of a round.  Compiled out otherwise.
#ifdef LSFM_K9_TIMING
__device__ unsigned long long g_tr_t[16];
#define TRT_DECL unsigned long long tracc[12] = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 }; unsigned long long trprev = __builtin_readcyclecounter()
#define TRT(i) do { const unsigned long long n_ = __builtin_readcyclecounter(); tracc[i] += n_ - trprev; trprev = n_; } while (0)
#define TRT_FLUSH() do { if (threadIdx.x == 0) { for (int i_ = 0; i_ < 12; i_++) atomicAdd(&g_tr_t[i_], tracc[i_]); atomicAdd(&g_tr_t[12], 1ull); } } while (0)
extern "C" void lsfm_debug_tr(unsigned long long* out, int reset)
{
	(void)hipDeviceSynchronize();
	if (out) (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_tr_t), sizeof(unsigned long long) * 16);
	if (reset) { unsigned long long z[16] = { 0 }; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_tr_t), z, sizeof(z)); }
}
#else
#define TRT_DECL do { } while (0)
#define TRT(i) do { } while (0)
#define TRT_FLUSH() do { } while (0)
#endif

template <int NH>
__global__ void __launch_bounds__(TRE_ROUND, NH == 1 ? 3 : 2)
k_tr_entries(int NF, int M, const int4* __restrict__ finfo, const int* __restrict__ fptr,
             const double* __restrict__ Wold, const int* __restrict__ photo, const int* __restrict__ KW, const double* __restrict__ Dp,
             const double* __restrict__ Cp, const double* __restrict__ FD, double* __restrict__ Wn_, int* __restrict__ nphoto,
             int* __restrict__ nfeature, double* __restrict__ Gsum, double* __restrict__ Gpose, int* __restrict__ hubJ, int alias_passthrough,
             const int* __restrict__ wbase, const int* __restrict__ newf, int* __restrict__ srcf)
{
	// LDS: pose table + 36 KB block rows.  Stereo: 32 poses, 9 KB -> three work-groups per CU.  Mono: two hub columns, and far up
	// a deep tree a tile of 128 features is seen by 33-64 poses (two hub blocks per feature and level) -- with a 32-entry table
	// most of a top level's blocks fell through to 72 global atomics each on rows the neighbouring tiles also add to (a synth-16k
	// root transform: 35 ms); 64 entries, 37 KB: still the two work-groups per CU its registers allow
	constexpr int GCAP = NH == 1 ? 32 : 64, TW = 18 * NH;
	__shared__ int gkeys[GCAP];
	__shared__ double gvals[NH * GCAP * 36];
	__shared__ double sT[TRE_ROUND * 18];
	__shared__ int sFp[TRE_TILE + 1];
	__shared__ unsigned char sAct[TRE_TILE]; // finfo.x of the tile's features (the feature sums asked memory for it every round)
	const int tid = threadIdx.x;
	TRT_DECL;
	const int f0 = blockIdx.x * TRE_TILE, f1 = min(f0 + TRE_TILE, NF), nft = f1 - f0;
	for (int i = tid; i < GCAP; i += TRE_ROUND) gkeys[i] = -1;
	for (int i = tid; i < NH * GCAP * 36; i += TRE_ROUND) gvals[i] = 0.0;
	for (int i = tid; i <= nft; i += TRE_ROUND) sFp[i] = fptr[f0 + i];
	for (int i = tid; i < nft; i += TRE_ROUND) sAct[i] = finfo[f0 + i].x != 0;
	__syncthreads();
	int la = 0; // first feature (tile-local) of the round
	int pf_j = -1, pf_f = 0, pf_k = 0, pf_kw = 0, pf_wb = 0, pf_lab = 0;
	int4 pf_fi = make_int4(0, 0, 0, 0);
	TRT(0);
	while (la < nft)
	{
		// a round = whole features from la on with at most TRE_ROUND blocks; a longer feature is walked in chunks
		const int e0 = sFp[la];
		int lb = la + 1;
		{
			int lo = la + 1, hi = nft; // largest lb with sFp[lb] - e0 <= TRE_ROUND (at least la + 1)
			while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (sFp[mid] - e0 <= TRE_ROUND) lo = mid; else hi = mid - 1; }
			lb = lo;
		}
		const int e1 = sFp[lb];
		for (int ce0 = e0; ce0 < e1 || ce0 == e0; ce0 += TRE_ROUND)
		{
			const int ce1 = min(ce0 + TRE_ROUND, e1);
			const int j = ce0 + tid;
			const bool have = j < ce1;
			TRT(1); // round set-up (search of the round's end)
#ifdef LSFM_K9_TIMING
			if (tid == 0) tracc[11]++;
#endif
			bool act = false, is_hub = false;
			int f = 0, k = 0, sl = -1;
			double W[18];
			const double* fd = nullptr;
			if (have)
			{
				int4 fi; // x: active, y/z: hub poses, w: placement
				int wb, lab, kw;
				if (pf_j == j) { f = pf_f; fi = pf_fi; k = pf_k; wb = pf_wb; lab = pf_lab; kw = pf_kw; } // fetched during the last round
				else
				{
					int lo = la, hi = lb - 1; // feature of block j: last fl with sFp[fl] <= j
					while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (sFp[mid] <= j) lo = mid; else hi = mid - 1; }
					f = f0 + lo;
					fi = finfo[f];
					k = photo[j];
					wb = wbase[f]; lab = newf ? newf[f] : f; kw = KW[j];
				}
				act = fi.x != 0;
				if (!act)
				{
					// pass-through map: indices only when the consumer (a join) reads the block from the input (W_alias)
					const int pos = wb + fi.w + j;
					if (!alias_passthrough)
					{
						ld<18>(W, Wold + (size_t)j * 18);
						st<18>(Wn_ + (size_t)pos * 18, W);
					}
					nphoto[pos] = k; nfeature[pos] = lab;
					if (srcf) srcf[pos] = f;
				}
				else
				{
					ld<18>(W, Wold + (size_t)j * 18);
					fd = FD + (size_t)f * (9 + TW);
					const bool hub = (k == fi.y) || (NH == 2 && k == fi.z);
					is_hub = hub;
					if (!hub)
					{
						// W' = D_k^T W D_f at its new place: after the feature's hub block(s), old order kept
						double T1[18], Df[9], Wn[18];
						ld<9>(Df, fd);
						if constexpr (NH == 1)
						{
							// D_k = [R 0; 0 DD2_k], R = D_f (Stereo): the same sums as the 6x6 product, without its zero terms
							double DD2[9];
							ld<9>(DD2, Cp + (size_t)M * 36 + (size_t)k * 27);
#pragma unroll
							for (int i = 0; i < 3; i++)
#pragma unroll
								for (int j = 0; j < 3; j++)
								{
									double s0 = 0.0, s1 = 0.0;
#pragma unroll
									for (int q = 0; q < 3; q++) { s0 = fma(Df[3 * q + i], W[3 * q + j], s0); s1 = fma(DD2[3 * q + i], W[3 * (3 + q) + j], s1); }
									T1[3 * i + j] = s0; T1[3 * (3 + i) + j] = s1;
								}
						}
						else
						{
							// D_k = [tl tr; 0 br]: the sums of the 6x6 product in its order, without the zero terms
							double Dt[27];
							ld<27>(Dt, Cp + (size_t)M * 72 + (size_t)k * 63);
#pragma unroll
							for (int i = 0; i < 3; i++)
#pragma unroll
								for (int j = 0; j < 3; j++)
								{
									double s0 = 0.0, s1 = 0.0;
#pragma unroll
									for (int q = 0; q < 3; q++) { s0 = fma(Dt[3 * q + i], W[3 * q + j], s0); s1 = fma(Dt[9 + 3 * q + i], W[3 * q + j], s1); }
#pragma unroll
									for (int q = 0; q < 3; q++) s1 = fma(Dt[18 + 3 * q + i], W[3 * (3 + q) + j], s1);
									T1[3 * i + j] = s0; T1[3 * (3 + i) + j] = s1;
								}
						}
						mm<6, 3, 3, false>(T1, Df, Wn);
						const int pos = wb + fi.w + kw;
						st<18>(Wn_ + (size_t)pos * 18, Wn);
						nphoto[pos] = k; nfeature[pos] = lab;
						if (srcf) srcf[pos] = f;
					}
					else
					{
						// an old block to a hub pose: the epilogue folds it into the new hub block; remember where it is
						// (-1 none, >= 0 the block, -2 several -- duplicates add up -- and the epilogue walks the run)
						int* h = hubJ + (size_t)f * NH + ((k == fi.y) ? 0 : NH - 1);
						if (atomicCAS(h, -1, j) != -1) atomicExch(h, -2);
					}
					sl = lds_slot(gkeys, GCAP, k);
				}
			}
			TRT(2); // own block: index, W load, W' = D^T W D, store
			// the index data of this lane's block in the NEXT round (it starts where this one ends): the loads fly during the
			// reductions below instead of heading the next round's dependent chain
			{
				const int jn = ce1 + tid;
				pf_j = -1;
				if (jn < sFp[nft])
				{
					int lo = 0, hi = nft - 1;
					while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (sFp[mid] <= jn) lo = mid; else hi = mid - 1; }
					pf_j = jn; pf_f = f0 + lo;
					pf_fi = finfo[pf_f]; pf_k = photo[jn]; pf_kw = KW[jn]; pf_wb = wbase[pf_f]; pf_lab = newf ? newf[pf_f] : pf_f;
				}
			}
			TRT(3); // prefetch issue
#pragma unroll
			for (int s = 0; s < NH; s++)
			{
				if (act)
				{
					double Cf[18], R[9];
					if constexpr (NH == 1) ld<9>(R, fd);
					if constexpr (NH == 1)
					{
						// C_f = [-R | T_f] (k_tr_feat_pre): R is in registers already, only the feature's own 9 numbers are fetched
#pragma unroll
						for (int m = 0; m < 3; m++)
#pragma unroll
							for (int c = 0; c < 3; c++) { Cf[6 * m + c] = -R[3 * m + c]; Cf[6 * m + 3 + c] = fd[9 + 6 * m + 3 + c]; }
					}
					else ld<18>(Cf, fd + 9 + 18 * s);
					// pose row of G_s: W C_s,f  [6x6] into the tile's LDS table.  The lanes of a wave that share a pose (a tile's 128
					// features are seen by a dozen poses: ~4 lanes per pose and wave) used to add element (r, c) of their products to
					// the SAME address in the same instruction, 36 times -- an LDS double-precision atomic serialises the lanes of an
					// address at ~11 clocks each, CU-wide (tools/microbench/lds_ops.hip: 8 clocks per conflict-free instruction, 92
					// with 8 lanes per address): 20 % of this kernel.  Now a lane parks half a product (18 numbers) in its own row
					// of sT and adds them in an order rotated by its lane number: lanes of one pose hit different elements.
					double* gl = sl >= 0 ? &gvals[(s * GCAP + sl) * 36] : nullptr;
					double* gg = Gpose + (size_t)s * M * 36 + (size_t)k * 36;
					double* my = &sT[tid * 18];
					const int rot = (tid & 63) % 18;
#pragma unroll
					for (int half = 0; half < 2; half++)
					{
#pragma unroll
						for (int i = 0; i < 18; i++)
						{
							const int r = 3 * half + i / 6, c = i % 6;
							my[i] = W[3 * r] * Cf[c] + W[3 * r + 1] * Cf[6 + c] + W[3 * r + 2] * Cf[12 + c];
							if (i % 6 == 5) __builtin_amdgcn_sched_barrier(0); // (a row at a time: 18 products formed before the first store do not fit the registers)
						}
#pragma unroll 1
						for (int e = 0; e < 18; e++)
						{
							int i = e + rot;
							if (i >= 18) i -= 18;
							const double v = my[i];
							if (gl) lds_add_f64(gl + 18 * half + i, v); else atomic_add_f64(gg + 18 * half + i, v);
						}
					}
					// (W^T C_s,k only now: its 18 numbers are not alive while the pose rows are formed -- the kernel is at its register budget)
					double T[18];
					zero<18>(T);
					if constexpr (NH == 1)
					{
						// C_k = [-R T_k; 0 DD_k] (zero for the hub pose): W^T C_k from the 18 packed numbers and R = D_f
						if (!is_hub)
						{
							double TK[9], DDk[9];
							ld<9>(TK, Cp + (size_t)M * 36 + (size_t)k * 27 + 9);
							ld<9>(DDk, Cp + (size_t)M * 36 + (size_t)k * 27 + 18);
#pragma unroll
							for (int i = 0; i < 3; i++)
#pragma unroll
								for (int j = 0; j < 3; j++)
								{
									double s0 = 0.0, s1 = 0.0;
#pragma unroll
									for (int q = 0; q < 3; q++) { s0 = fma(W[3 * q + i], -R[3 * q + j], s0); s1 = fma(W[3 * q + i], TK[3 * q + j], s1); }
#pragma unroll
									for (int q = 0; q < 3; q++) s1 = fma(W[3 * (3 + q) + i], DDk[3 * q + j], s1);
									T[6 * i + j] = s0; T[6 * i + 3 + j] = s1;
								}
						}
					}
					else
					{
						// share of G_s,f: W^T C_s,k [3x6] with C1_k = [tl tr; 0 br], C2_k = [tl 0; 0 0]
						const double* pc = Cp + (size_t)M * 72 + (size_t)k * 63 + (s == 0 ? 27 : 54);
						double Ct[27];
						ld<9>(Ct, pc);
						if (s == 0) ld<18>(Ct + 9, pc + 9);
#pragma unroll
						for (int i = 0; i < 3; i++)
#pragma unroll
							for (int j = 0; j < 3; j++)
							{
								double s0 = 0.0;
#pragma unroll
								for (int q = 0; q < 3; q++) s0 = fma(W[3 * q + i], Ct[3 * q + j], s0);
								T[6 * i + j] = s0;
								if (s == 0)
								{
									double s1 = 0.0;
#pragma unroll
									for (int q = 0; q < 3; q++) s1 = fma(W[3 * q + i], Ct[9 + 3 * q + j], s1);
#pragma unroll
									for (int q = 0; q < 3; q++) s1 = fma(W[3 * (3 + q) + i], Ct[18 + 3 * q + j], s1);
									T[6 * i + 3 + j] = s1;
								}
							}
					}
					TRT(5); // C_k load, W^T C_k
					st<18>(&sT[tid * 18], T); // (the lane's row of W^T C_s,k for the feature sums below, over the parking place)
				}
				TRT(8); // pose rows: 36 LDS atomic adds
				__syncthreads();
				TRT(6);
				// per feature sums of the W^T C_s,k rows of this chunk (the rows of inactive maps are never read)
				for (int idx = tid; idx < (lb - la) * 18; idx += TRE_ROUND)
				{
					const int fl = la + idx / 18, q = idx % 18;
					const int r0 = max(sFp[fl], ce0) - ce0, r1 = min(sFp[fl + 1], ce1) - ce0;
					if (!sAct[fl]) continue;
					double sum = 0.0;
					for (int r = r0; r < r1; r++) sum += sT[r * 18 + q];
					double* g = Gsum + (size_t)(f0 + fl) * TW + 18 * s + q;
					if (sFp[fl] >= ce0) *g = sum; else if (r1 > r0) *g += sum; // first chunk of the feature stores
				}
				TRT(7); // feature sums
				__syncthreads();
				TRT(6);
			}
			if (ce1 >= e1) break;
		}
		la = lb;
	}
#pragma unroll
	for (int s = 0; s < NH; s++) tile_flush<36>(gkeys, gvals + s * GCAP * 36, GCAP, Gpose + (size_t)s * M * 36);
	TRT(9);
	TRT_FLUSH();
}

template <int NH>
__global__ void __launch_bounds__(256, NH == 1 ? 2 : 1)
k_tr_feat_post(int NF, int M, const TMap* __restrict__ tm, const int* __restrict__ feat_map, const int* __restrict__ fptr,
               const double* __restrict__ Vold, const double* __restrict__ Wold, const int* __restrict__ photo, const double* __restrict__ Dp,
               const double* __restrict__ FD, const double* __restrict__ Gsum, const int* __restrict__ hubJ, const int* __restrict__ wbase,
               const int* __restrict__ newf, int* __restrict__ srcf, double* __restrict__ Wn_,
               int* __restrict__ nphoto, int* __restrict__ nfeature, double* __restrict__ PP)
{
	constexpr int TW = 18 * NH;
	const int f = blockIdx.x * blockDim.x + threadIdx.x;
	const bool inb = f < NF;
	const TMap* t = inb ? &tm[feat_map[f]] : nullptr;
	const bool act = inb && t->active > 0 && t->nh == NH;
	double Df[9], Cf[NH][18], G[NH][18];
	if (act)
	{
		const double* fd = FD + (size_t)f * (9 + TW);
		ld<9>(Df, fd);
		const int base = wbase[f];
		const int hubs[2] = { t->hub[0], NH == 2 ? t->hub[1] : -1 };
		double hubW[NH][18];
#pragma unroll
		for (int s = 0; s < NH; s++) zero<18>(hubW[s]);
		// old blocks of the feature to the hub pose(s): their D_k^T W D_f joins the new hub block.  The entry kernel
		// left the block's index; only duplicates make the lane walk its run
		bool walk = false;
#pragma unroll
		for (int s = 0; s < NH; s++)
		{
			const int j = hubJ[(size_t)f * NH + s];
			if (j == -2) walk = true;
			if (j < 0) continue;
			double W[18], Dk[36], T1[18], Wn[18];
			ld<18>(W, Wold + (size_t)j * 18);
			ld<36>(Dk, Dp + (size_t)hubs[s] * 36);
			mtm<6, 6, 3, false>(Dk, W, T1);
			mm<6, 3, 3, false>(T1, Df, Wn);
			for (int i = 0; i < 18; i++) hubW[s][i] += Wn[i];
		}
		if (walk)
			for (int j = fptr[f]; j < fptr[f + 1]; j++)
			{
				const int k = photo[j];
				const int s = (k == hubs[0]) ? 0 : ((NH == 2 && k == hubs[1]) ? 1 : -1);
				if (s < 0 || hubJ[(size_t)f * NH + (s == 0 ? 0 : NH - 1)] != -2) continue;
				double W[18], Dk[36], T1[18], Wn[18];
				ld<18>(W, Wold + (size_t)j * 18);
				ld<36>(Dk, Dp + (size_t)k * 36);
				mtm<6, 6, 3, false>(Dk, W, T1);
				mm<6, 3, 3, false>(T1, Df, Wn);
				for (int i = 0; i < 18; i++) hubW[s == 0 ? 0 : NH - 1][i] += Wn[i];
			}
		// G_s,f = V C_s,f + sum_k W_kf^T C_s,k  (after the hub blocks above: C_f, V and the sums are not alive while a 6x6 pose
		// Jacobian and a block are -- the Stereo kernel spilled 16 registers with everything loaded up front)
		{
			double V[9];
			ld<9>(V, Vold + (size_t)f * 9);
#pragma unroll
			for (int s = 0; s < NH; s++)
			{
				ld<18>(Cf[s], fd + 9 + 18 * s);
				double Gs[18];
				ld<18>(Gs, Gsum + (size_t)f * TW + 18 * s);
				mm<3, 3, 6, false>(V, Cf[s], G[s]);
				for (int i = 0; i < 18; i++) G[s][i] += Gs[i];
			}
		}
		// leading hub block(s) of the feature: W'(h_s, f) = hubW_s + G_s^T D_f
#pragma unroll
		for (int s = 0; s < NH; s++)
		{
			double Wh[18];
			ld<18>(Wh, hubW[s]);
			mtm<3, 6, 3, true>(G[s], Df, Wh);
			st<18>(Wn_ + (size_t)(base + s) * 18, Wh);
			nphoto[base + s] = hubs[s]; nfeature[base + s] = newf ? newf[f] : f;
			if (srcf) srcf[base + s] = f;
		}
	}
	// C_s^T G_t for the (h,h) blocks, summed per map
#pragma unroll
	for (int s = 0; s < NH; s++)
#pragma unroll
		for (int s2 = s; s2 < NH; s2++)
		{
			double P[36];
			if (act) mtm<3, 6, 6, false>(Cf[s], G[s2], P);
			const int idx = (s == 0 ? s2 : 2);
			const int mp = act ? feat_map[f] : 0;
			wave_scatter_add<36>(PP + ((size_t)mp * 3 + idx) * 36, P, act);
		}
}

template <int NH>
__global__ void __launch_bounds__(128, 1)
k_tr_ublocks(int NU, int M, const TMap* __restrict__ tm, const int* __restrict__ pose_map, const double* __restrict__ Uold,
             const int* __restrict__ Ui, const int* __restrict__ Uj, const int* __restrict__ KU, const double* __restrict__ Dp,
             const double* __restrict__ Cp, double* __restrict__ Un, int* __restrict__ nUi, int* __restrict__ nUj,
             double* __restrict__ Gpose)
{
	// pose rows of G: summed per work-group in an LDS table first (the hub rows collect a block from every pose: 108
	// scattered 8-byte atomics per lane ran at the 0.08 TB/s of 64-rows-per-instruction atomics)
	constexpr int GCAP = 64;
	__shared__ int gkeys[GCAP];
	__shared__ double gvals[NH * GCAP * 36];
	__shared__ double sStage[128 * 37]; // a lane's 36 numbers on their way to the table (tile_scatter_add_rot); odd stride
	for (int q = threadIdx.x; q < GCAP; q += blockDim.x) gkeys[q] = -1;
	for (int q = threadIdx.x; q < NH * GCAP * 36; q += blockDim.x) gvals[q] = 0.0;
	__syncthreads();
	const int i0 = blockIdx.x * blockDim.x + threadIdx.x;
	const bool inb = i0 < NU;
	const int i = inb ? i0 : 0;
	const int a = Ui[i], b = Uj[i];
	const TMap& t = tm[pose_map[a]];
	const bool act = inb && t.active > 0 && t.nh == NH;
	double U[36];
	ld<36>(U, Uold + (size_t)i * 36);
	if (inb && t.active <= 0)
	{
		const int pos = t.U0n + (i - t.U0);
		st<36>(Un + (size_t)pos * 36, U);
		nUi[pos] = a; nUj[pos] = b;
	}
	if (act)
	{
		double T1[36], T[36];
		{
			double Da[36];
			ld<36>(Da, Dp + (size_t)a * 36);
			mtm<6, 6, 6, false>(Da, U, T1);
		}
		{
			double Db[36];
			ld<36>(Db, Dp + (size_t)b * 36);
			mm<6, 6, 6, false>(T1, Db, T);        // D_a^T U D_b, block (a,b) with a<=b
		}
		int slot = -1;                            // Imp.cpp:1079-1094 / 4249-4271
		if (a == t.hub[0]) slot = t.U0n + (b - t.P0);
		else if (b == t.hub[0]) slot = t.U0n + (a - t.P0);
		else if (NH == 2 && a == t.hub[1]) slot = t.U0n + t.m + (b - t.P0);
		else if (NH == 2 && b == t.hub[1]) slot = t.U0n + t.m + (a - t.P0);
		if (slot >= 0)
		{
			for (int q = 0; q < 36; q++) atomic_add_f64(Un + (size_t)slot * 36 + q, T[q]);
		}
		else
		{
			const int pos = t.U0n + NH * t.m + (KU[i] - t.kU0);
			st<36>(Un + (size_t)pos * 36, T);
			nUi[pos] = a; nUj[pos] = b;
		}
	}
	// pose rows of G_s = I C_s : G_a += U C_b ; a != b: G_b += U^T C_a   (wave collectives: every lane takes part)
#pragma unroll
	for (int s = 0; s < NH; s++)
	{
		double C[36], Y[36];
		if (act)
		{
			ld<36>(C, Cp + (size_t)s * M * 36 + (size_t)b * 36);
			mm<6, 6, 6, false>(U, C, Y);
		}
		tile_scatter_add_rot<36>(gkeys, gvals + s * GCAP * 36, GCAP, a, Gpose + (size_t)s * M * 36 + (size_t)a * 36, Y, act, &sStage[threadIdx.x * 37]);
		const bool off = act && a != b;
		if (off)
		{
			ld<36>(C, Cp + (size_t)s * M * 36 + (size_t)a * 36);
			mtm<6, 6, 6, false>(U, C, Y);
		}
		tile_scatter_add_rot<36>(gkeys, gvals + s * GCAP * 36, GCAP, b, Gpose + (size_t)s * M * 36 + (size_t)b * 36, Y, off, &sStage[threadIdx.x * 37]);
	}
	__syncthreads();
#pragma unroll
	for (int s = 0; s < NH; s++) tile_flush<36>(gkeys, gvals + s * GCAP * 36, GCAP, Gpose + (size_t)s * M * 36);
}

__device__ __forceinline__ void add_oriented(double* dst, const double* X, int r, int c)
{
	if (r < c) { for (int q = 0; q < 36; q++) dst[q] += X[q]; }
	else if (r > c) { for (int i = 0; i < 6; i++) for (int j = 0; j < 6; j++) dst[j * 6 + i] += X[i * 6 + j]; }
	else { for (int i = 0; i < 6; i++) for (int j = 0; j < 6; j++) dst[i * 6 + j] += X[i * 6 + j] + X[j * 6 + i]; }
}

// one lane per pose: new blocks (k,h_s) = D_k^T G_s,k (+ what k_tr_ublocks already put there) and C_k^T G_k
template <int NH>
__global__ void __launch_bounds__(128, 1) // a lane holds several 6x6 blocks: let it have the registers (no spills)
k_tr_poseslots(int M, const TMap* __restrict__ tm, const int* __restrict__ pose_map, const double* __restrict__ Dp,
                               const double* __restrict__ Cp, const double* __restrict__ Gpose, double* __restrict__ Un, int* __restrict__ nUi,
                               int* __restrict__ nUj, double* __restrict__ PP)
{
	const int k0 = blockIdx.x * blockDim.x + threadIdx.x;
	const bool inb = k0 < M;
	const int k = inb ? k0 : 0;
	const int mp = pose_map[k];
	const TMap& t = tm[mp];
	const bool act = inb && t.active > 0 && t.nh == NH; // no early return: the per-map sums below are wave collectives
	double D[36];
	ld<36>(D, Dp + (size_t)k * 36);
	const int h0 = t.hub[0], h1 = NH == 2 ? t.hub[1] : -1;
	const int slot1 = t.U0n + (k - t.P0), slot2 = t.U0n + t.m + (k - t.P0);
	// labels, Imp.cpp:711-723 / 3739-3765 (the second set compares against posID as well -- reference quirk)
	if (act)
	{
		if (k <= h0) { nUi[slot1] = k; nUj[slot1] = h0; } else { nUi[slot1] = h0; nUj[slot1] = k; }
		if (NH == 2) { if (k <= h0) { nUi[slot2] = k; nUj[slot2] = h1; } else { nUi[slot2] = h1; nUj[slot2] = k; } }
	}
#pragma unroll
	for (int s = 0; s < NH; s++)
	{
		double G[36], X[36];
		ld<36>(G, Gpose + (size_t)s * M * 36 + (size_t)k * 36);
		mtm<6, 6, 6, false>(D, G, X); // contribution to I'(k, h_s)
		const int hs = s == 0 ? h0 : h1;
		if (!act) {}
		else if (NH == 2 && s == 1 && k == h0)
		{
			// pair (h_1,h_2): everything goes to the first-set slot of h_2 (the reference splits it over two slots with
			// the same coordinates; only their sum is defined)
			double* dst = Un + (size_t)(t.U0n + (h1 - t.P0)) * 36;
			if (h0 < h1) { for (int q = 0; q < 36; q++) atomic_add_f64(dst + q, X[q]); }
			else { for (int i = 0; i < 6; i++) for (int j = 0; j < 6; j++) atomic_add_f64(dst + j * 6 + i, X[i * 6 + j]); }
		}
		else if (NH == 2 && s == 0 && k == h1)
		{
			double* dst = Un + (size_t)slot1 * 36; // block (min(h0,h1), max): X is I'(h1,h0)
			if (h1 < h0) { for (int q = 0; q < 36; q++) atomic_add_f64(dst + q, X[q]); }
			else { for (int i = 0; i < 6; i++) for (int j = 0; j < 6; j++) atomic_add_f64(dst + j * 6 + i, X[i * 6 + j]); }
		}
		else
		{
			// only this lane touches the block in this kernel except the two cross cases above, which use atomics on
			// slot1(h1); keep everything atomic on that one slot
			double* dst = Un + (size_t)(s == 0 ? slot1 : slot2) * 36;
			add_oriented(dst, X, k, hs);
		}
		// sum_a C_a^T G_a
#pragma unroll
		for (int s0 = 0; s0 <= s; s0++)
		{
			double C[36], P[36];
			ld<36>(C, Cp + (size_t)s0 * M * 36 + (size_t)k * 36);
			mtm<6, 6, 6, false>(C, G, P);
			const int idx = (s0 == 0 ? s : 2);
			wave_scatter_add<36>(PP + ((size_t)mp * 3 + idx) * 36, P, act);
		}
	}
}

template <int NH>
__global__ void k_tr_diag(int B, const TMap* __restrict__ tm, const double* __restrict__ PP, double* __restrict__ Un)
{
	int b = blockIdx.x * blockDim.x + threadIdx.x;
	if (b >= B) return;
	const TMap& t = tm[b];
	if (t.active <= 0 || t.nh != NH) return;
	const double* P = PP + (size_t)b * 3 * 36;
	double* d0 = Un + (size_t)(t.U0n + (t.hub[0] - t.P0)) * 36;
	for (int q = 0; q < 36; q++) d0[q] += P[q];
	if (NH == 2)
	{
		double* d2 = Un + (size_t)(t.U0n + t.m + (t.hub[1] - t.P0)) * 36;
		for (int q = 0; q < 36; q++) d2[q] += P[2 * 36 + q];
		double* d1 = Un + (size_t)(t.U0n + (t.hub[1] - t.P0)) * 36; // block (h_1,h_2), upper orientation
		const double* P01 = P + 36;                                  // C_1^T I C_2 = contribution to I'(h_1,h_2)
		if (t.hub[0] < t.hub[1]) { for (int q = 0; q < 36; q++) d1[q] += P01[q]; }
		else { for (int i = 0; i < 6; i++) for (int j = 0; j < 6; j++) d1[j * 6 + i] += P01[i * 6 + j]; }
	}
}

template <int NH>
static void launch_stage(lsfm_context* ctx, const DevBatch& in, DevBatch& out, const TMap* d_tm, const int* KU, const int* KW,
                         double* Dp, double* Cp, double* Gpose, double* PP, double nw_act_in, double nw_act_out, double nf_act,
                         const std::function<TrRedirect(DevBatch&)>* hook)
{
	hipStream_t s = ctx->stream;
	const int M = in.M;
	bool ev_entries = false, u_early = false;
	static const bool side = !getenv("LSFM_NO_SIDE_STREAM");
	static const bool early_u = !getenv("LSFM_U_STAGE_LATE");
	if (!in.NF && hook) (void)(*hook)(out); // nothing to redirect, but the consumer still lays out its container
	if (in.NF)
	{
		// per feature: D_f and C_s,f (written by the prologue, read per W block), sums of W^T C_s,k (entries -> epilogue)
		double* FD = ctx->scratch.alloc<double>((size_t)in.NF * (9 + 18 * NH));
		double* Gsum = ctx->scratch.alloc<double>((size_t)in.NF * 18 * NH);
		int* hubJ = ctx->scratch.alloc<int>((size_t)in.NF * NH);
		int4* finfo = ctx->scratch.alloc<int4>(in.NF);
		hipLaunchKernelGGL(k_tr_feat_pre<NH>, dim3((in.NF + 255) / 256), dim3(256), 0, s, in.NF, d_tm, in.feat_map, in.feat, in.fptr, in.V, KW, out.feat,
		                   out.fptr, out.V, FD, finfo, hubJ);
		// the W blocks go to this container -- or straight into the next one when the consumer has laid it out already
		TrRedirect rd;
		if (hook) rd = (*hook)(out);
		if (!rd.W) { rd = TrRedirect(); rd.wbase = out.fptr; rd.W = out.W; rd.photo = out.photo; rd.feature = out.feature; }
		// The U blocks' kernel reads nothing the feature kernels write (U, the poses' Jacobians, the kept-block ranks) and ADDS to the pose
		// rows of G like they do: it goes to the side stream HERE, beside the block kernel of the features, not behind it -- one lane per
		// 6x6 block with a few hundred dependent multiply-adds each, it takes 160-190 us whatever the level's size, and behind the
		// block kernel the main stream waited 60 us per level for it (k_tr_diag needs the U stage).  Only the poses' kernel, which
		// reads the finished rows, still waits for the block kernel.  LSFM_U_STAGE_LATE=1: as until round 6.
		if (side && early_u && !ctx->comm && in.NU)
		{
			LSFM_CHECK_HIP(hipEventRecord(ctx->evU, s));
			LSFM_CHECK_HIP(hipStreamWaitEvent(ctx->stream2, ctx->evU, 0));
			hipLaunchKernelGGL(k_tr_ublocks<NH>, dim3((in.NU + 127) / 128), dim3(128), 0, ctx->stream2, in.NU, M, d_tm, in.pose_map, in.U, in.Ui, in.Uj, KU,
			                   Dp, Cp, out.U, out.Ui, out.Uj, Gpose);
			u_early = true;
		}
		hipEvent_t e0 = nullptr, e1 = nullptr; // the events bracket k_tr_entries alone; read at the end of the run
		if (ctx->stats) { e0 = ctx->pool_event(); e1 = ctx->pool_event(); LSFM_REC_T(e0, s); }
		hipLaunchKernelGGL(k_tr_entries<NH>, dim3((in.NF + TRE_TILE - 1) / TRE_TILE), dim3(TRE_ROUND), 0, s, in.NF, M, finfo, in.fptr, in.W,
		                   in.photo, KW, Dp, Cp, FD, rd.W, rd.photo, rd.feature, Gsum, Gpose, hubJ, out.W_alias ? 1 : 0, rd.wbase, rd.newf, rd.srcf);
		if (ctx->stats) LSFM_REC_T(e1, s);
		LSFM_CHECK_HIP(hipEventRecord(ctx->evA, s)); // pose rows of G complete: the U stage may start on the side stream
		ev_entries = true;
		hipLaunchKernelGGL(k_tr_feat_post<NH>, dim3((in.NF + 255) / 256), dim3(256), 0, s, in.NF, M, d_tm, in.feat_map, in.fptr, in.V, in.W, in.photo,
		                   Dp, FD, Gsum, hubJ, rd.wbase, rd.newf, rd.srcf, rd.W, rd.photo, rd.feature, PP);
		if (ctx->stats)
		{
			ctx->defer_time(e0, e1, &ctx->stats->trf_ms);
			ctx->stats->trf_launches++;
			// k_tr_entries, every input and output once.  Blocks of transformed maps: W block + photo + kept-rank in, W' block
			// + photo' + feature' out (the blocks to the hub poses are not written here); per feature of a transformed map
			// D_f/C_f + record + run pointer in, the W^T C sums out.  Blocks of pass-through maps: photo in, photo' + feature'
			// out, plus the W block itself both ways when it is copied (not aliased)
			const double nw_pass = (double)in.NW - nw_act_in;
			ctx->stats->trf_bytes += nw_act_in * (144 + 4 + 4) + nf_act * ((9 + 18 * NH) * 8 + 16 + 4) + (nw_act_out - NH * nf_act) * (144 + 8) +
			                         nf_act * 18 * NH * 8 + nw_pass * (4 + 8 + (out.W_alias ? 0 : 288)) + (double)(in.NF - nf_act) * (16 + 4);
		}
	}
	// U stage: its second kernel needs the pose rows of G (complete after k_tr_entries and k_tr_ublocks) but nothing of the feature
	// epilogue, which runs on the main stream meanwhile; k_tr_diag needs both
	// feature-sharded run: what the features of this rank's slice added to the pose rows of G (k_tr_entries) and to the hub-hub
	// blocks (k_tr_feat_post) becomes the sum over all slices before the pose kernels read and extend it
	if (ctx->comm) ctx->comm->allreduce(s, Gpose, (size_t)M * 36 * NH + (size_t)in.B * 3 * 36, LSFM_DTYPE_F64);
	hipStream_t su = (side && ev_entries && !ctx->comm) ? ctx->stream2 : s;
	if (su != s) LSFM_CHECK_HIP(hipStreamWaitEvent(su, ctx->evA, 0));
	if (in.NU && !u_early)
		hipLaunchKernelGGL(k_tr_ublocks<NH>, dim3((in.NU + 127) / 128), dim3(128), 0, su, in.NU, M, d_tm, in.pose_map, in.U, in.Ui, in.Uj, KU,
		                   Dp, Cp, out.U, out.Ui, out.Uj, Gpose);
	if (M)
		hipLaunchKernelGGL(k_tr_poseslots<NH>, dim3((M + 127) / 128), dim3(128), 0, su, M, d_tm, in.pose_map, Dp, Cp, Gpose, out.U, out.Ui,
		                   out.Uj, PP);
	if (su != s)
	{
		LSFM_CHECK_HIP(hipEventRecord(ctx->evB, su));
		LSFM_CHECK_HIP(hipStreamWaitEvent(s, ctx->evB, 0));
	}
	hipLaunchKernelGGL(k_tr_diag<NH>, dim3((in.B + 127) / 128), dim3(128), 0, s, in.B, d_tm, PP, out.U);
}

// the output offsets of every map (known to the host once the kept-block counts are) into the device copy of the map records,
// which keeps what the device computed (hub indices, parameters); also the end of the new run pointers, and -- a level that does
// not stop for it -- the target-not-found flag into the run's record (three one-thread launches of their own until round 5)
__global__ void k_tr_patch(TMap* tm, const int4* __restrict__ offs, int B, int* __restrict__ fptr_last, int nw_total, const int* __restrict__ err, RunStatsDev* run)
{
	int b = blockIdx.x * blockDim.x + threadIdx.x;
	if (b == 0)
	{
		*fptr_last = nw_total;
		if (run && *err && !run->tr_err) run->tr_err = *err;
	}
	if (b >= B) return;
	const int4 o = offs[b];
	tm[b].kU0 = o.x; tm[b].kW0 = o.y; tm[b].U0n = o.z; tm[b].W0n = o.w;
}

// a planned Mono level took the signs of the new scales from the plan (the host mirrors of the map records need them
// before the device has computed anything): they are values, not structure -- check them against what k_tr_params1 found
__global__ void k_tr_sign_check(const TMap* __restrict__ tm, const int* __restrict__ planned, int B, RunStatsDev* run)
{
	int b = blockIdx.x * blockDim.x + threadIdx.x;
	if (b < B && tm[b].active > 0 && tm[b].sign1 != planned[b]) atomicExch(&run->plan_stale, 1);
}

void transform_batch(lsfm_context* ctx, Arena& ar, const DevBatch& in, const std::vector<int>& target_ref,
                     const std::vector<int>& target_scap, const std::vector<int>& target_fix, bool mono, DevBatch& out,
                     bool alias_passthrough, const std::function<TrRedirect(DevBatch&)>* hook)
{
	hipStream_t s = ctx->stream;
	const int B = in.B, nh = mono ? 2 : 1;
	size_t smark = ctx->scratch.mark();
	std::vector<TMap> tm(B);
	bool any = false;
	for (int b = 0; b < B; b++)
	{
		TMap& t = tm[b];
		memset(&t, 0, sizeof t);
		t.nh = nh;
		t.P0 = in.pose_off[b]; t.m = in.pose_off[b + 1] - t.P0;
		t.F0 = in.feat_off[b]; t.n = in.feat_off[b + 1] - t.F0;
		t.U0 = in.u_off[b]; t.W0 = in.w_off[b];
		t.hub[0] = t.hub[1] = t.nref = t.nscap = -1;
		bool act = target_ref[b] >= 0 && !(in.Ref[b] == target_ref[b] && (!mono || in.ScaP[b] == target_scap[b])); // Imp.cpp:352 / 3176
		t.active = act ? 1 : 0;
		any |= act;
		t.tref = target_ref[b]; t.tscap = mono ? target_scap[b] : 0; t.newFix = mono ? target_fix[b] : 0;
		t.oldref = in.Ref[b]; t.oldscap = mono ? in.ScaP[b] : 0; t.oldFix = mono ? in.Fix[b] : 0;
		t.newLabel = in.Ref[b];
	}
	TMap* d_tm = ctx->scratch.alloc<TMap>(B);
	// the accumulators of the stage, zeroed by ONE fill ahead of its first kernel: the target-not-found flag, the pose rows of I C
	// (Gpose) and the hub-hub blocks (PP).  Feature-sharded run: the two the features add to live back to back in the caller's
	// buffer and are summed over the ranks between the feature kernels and the pose kernels (launch_stage)
	int* d_err;
	double *Gpose, *PP;
	{
		const size_t ngp = (size_t)in.M * 36 * nh, npp = (size_t)B * 3 * 36;
		ZeroSpan zs(ctx->scratch);
		d_err = ctx->scratch.alloc<int>(1);
		if (ctx->comm)
		{
			ctx->comm->restart();
			Gpose = ctx->comm->alloc<double>(ngp + npp);
			PP = Gpose + ngp;
			fill_async(s, Gpose, 0, (ngp + npp) * sizeof(double));
		}
		else
		{
			Gpose = ctx->scratch.alloc<double>(ngp);
			PP = ctx->scratch.alloc<double>(npp);
		}
		zs.zero(s);
	}
	CopyBatch cb(ctx); // the small copies of the prologue: map records, offsets, label arrays -- one transfer, one kernel
	cb.h2d(d_tm, tm.data(), sizeof(TMap) * B);

	out = DevBatch();
	out.B = B; out.M = in.M; out.NF = in.NF;
	if (mono) { out.s_keys = in.s_keys; out.s_nnzb = in.s_nnzb; } // (pose indices are unchanged: the pattern below travels with the maps, lsfm_join_mono.hip)
	out.pose_off = in.pose_off; out.feat_off = in.feat_off;
	out.Ref = in.Ref; out.FRef = in.FRef; out.ScaP = in.ScaP; out.Fix = in.Fix; out.Sign = in.Sign; out.FScaP = in.FScaP; out.FFix = in.FFix;
	// nothing of `out` may alias `in`: the two live in different arenas with different lifetimes
	batch_set_offsets(ctx, ar, out, &cb);
	out.feat_id = ar.alloc<int>(in.NF);
	cb.d2d(out.feat_id, in.feat_id, (size_t)in.NF * sizeof(int));
	out.pose = ar.alloc<double>((size_t)in.M * 6);
	out.pose_id = ar.alloc<int>(in.M);
	out.pose_origin = ar.alloc<int>(in.M);
	cb.d2d(out.pose_origin, in.pose_origin, (size_t)in.M * sizeof(int));
	out.feat = ar.alloc<double>((size_t)in.NF * 3);
	out.V = ar.alloc<double>((size_t)in.NF * 9);
	out.fptr = ar.alloc<int>(in.NF + 1);
	int* d_uoff = ctx->scratch.alloc<int>(B + 1);
	int* d_woff = ctx->scratch.alloc<int>(B + 1);
	cb.h2d(d_uoff, in.u_off.data(), (B + 1) * sizeof(int));
	cb.h2d(d_woff, in.w_off.data(), (B + 1) * sizeof(int));
	cb.flush();
	batch_fill_maps(ctx, out);

	const int M = in.M;
	if (M)
	{
		hipLaunchKernelGGL(k_tr_find, dim3((M + 255) / 256), dim3(256), 0, s, in.pose_id, in.pose_map, M, d_tm);
		hipLaunchKernelGGL(k_tr_params1, dim3((B + 127) / 128), dim3(128), 0, s, in.pose, d_tm, B, d_err);
		hipLaunchKernelGGL(k_tr_new_poses, dim3((M + 255) / 256), dim3(256), 0, s, in.pose, in.pose_id, in.pose_map, M, d_tm, out.pose, out.pose_id);
		// (for the consumer laid out inside the hook -- a Stereo join that analyses --: what the early pattern of S is made from)
		int* d_hub = (hook && !mono && B) ? ctx->scratch.alloc<int>(B) : nullptr;
		hipLaunchKernelGGL(k_tr_params2, dim3((B + 127) / 128), dim3(128), 0, s, out.pose, d_tm, B, d_hub);
		ctx->tr_in = nullptr; ctx->tr_hub = nullptr;
		if (d_hub) { ctx->tr_in = &in; ctx->tr_hub = d_hub; }
	}
	else { ctx->tr_in = nullptr; ctx->tr_hub = nullptr; }
	double* Dp = ctx->scratch.alloc<double>((size_t)M * 36);
	// Stereo: + the 27 pose-dependent entries of (D_k, C_k) packed per pose, for k_tr_entries (after the one C section)
	double* Cp = ctx->scratch.alloc<double>((size_t)M * 36 * nh + (nh == 1 ? (size_t)M * 27 : (size_t)M * 63));
	if (M)
	{
		if (mono) hipLaunchKernelGGL(k_tr_pose_jac<2>, dim3((M + 127) / 128), dim3(128), 0, s, out.pose, in.pose_map, M, d_tm, Dp, Cp);
		else hipLaunchKernelGGL(k_tr_pose_jac<1>, dim3((M + 127) / 128), dim3(128), 0, s, out.pose, in.pose_map, M, d_tm, Dp, Cp);
	}
	// structure of the output: which blocks survive as they are, prefix sums, per-map offsets -- from the level's plan when it holds
	// them (LevelIndex: left by the preparation one level ahead, or kept by a resident tree), worked out here otherwise
	LevelPlan* plan = ctx->plan;
	const bool warm = ctx->warm();
	static const bool reuse_index = !getenv("LSFM_NO_INDEX_REUSE");
	const int *KU = nullptr, *KW = nullptr;
	if (reuse_index && warm && plan->idx.KW && plan->idx.KU && plan->idx.NU == in.NU && plan->idx.NW == in.NW) { KU = plan->idx.KU; KW = plan->idx.KW; }
	else
	{
		int* keepU = ctx->scratch.alloc<int>(in.NU + 1);
		int* keepW = ctx->scratch.alloc<int>(in.NW + 1);
		int* kU = ctx->scratch.alloc<int>(in.NU + 2);
		int* kW = ctx->scratch.alloc<int>(in.NW + 2);
		int nmax = std::max(std::max(in.NU, in.NW), 1);
		hipLaunchKernelGGL(k_tr_flags, dim3((nmax + 255) / 256), dim3(256), 0, s, in.Ui, in.Uj, in.NU, in.photo, in.NW, in.pose_map, d_tm, keepU, keepW);
		dev_exclusive_scan(ctx, keepU, kU, in.NU);
		dev_exclusive_scan(ctx, keepW, kW, in.NW);
		KU = kU; KW = kW;
		if (reuse_index && plan && !warm && plan != &ctx->pre_plan && ctx->in_tree_run)
		{
			// a resident tree records the level: its later runs skip the three launches above
			plan->idx.KU = level_index_keep(ctx, plan->idx, kU, (size_t)in.NU + 2);
			plan->idx.KW = level_index_keep(ctx, plan->idx, kW, (size_t)in.NW + 2);
			plan->idx.NU = in.NU; plan->idx.NW = in.NW;
		}
	}
	int* d_cnt = ctx->scratch.alloc<int>(3 * (B + 1));
	std::vector<int> cnt(2 * (B + 1)), dev_sign;
	if (warm) cnt = plan->tr_cnt; // the structure of this level is known from an earlier run of the same tree: no round trip
	else
	{
		hipLaunchKernelGGL(k_tr_gather_counts, dim3((B + 1 + 127) / 128), dim3(128), 0, s, KU, KW, d_uoff, d_woff, B, d_cnt, d_tm, d_err);
		ctx->mark("tr_enq");
		std::vector<int> all(3 * (size_t)(B + 1));
		d2h_ints(ctx, d_cnt, all.data(), all.size());
		std::copy(all.begin(), all.begin() + 2 * (B + 1), cnt.begin());
		dev_sign.assign(all.begin() + 2 * (B + 1), all.begin() + 2 * (B + 1) + B);
		const int err = all[3 * (size_t)(B + 1) - 1];
		ctx->mark("tr_cnt");
		if (err) LSFM_FAIL(LSFM_ERR_ARG, "transform: target pose id not found in map " + std::to_string(err - 1));
		if (plan) plan->tr_cnt = cnt;
	}
	out.u_off.assign(B + 1, 0); out.w_off.assign(B + 1, 0);
	for (int b = 0; b < B; b++)
	{
		TMap& t = tm[b];
		t.kU0 = cnt[b]; t.kW0 = cnt[B + 1 + b];
		int keptU = cnt[b + 1] - cnt[b], keptW = cnt[B + 1 + b + 1] - cnt[B + 1 + b];
		t.U0n = out.u_off[b]; t.W0n = out.w_off[b];
		out.u_off[b + 1] = t.U0n + (t.active ? nh * t.m : 0) + keptU;
		out.w_off[b + 1] = t.W0n + (t.active ? nh * t.n : 0) + keptW;
		if (t.active)
		{
			// Imp.cpp:409-411 / 3253-3259
			out.Ref[b] = t.tref;
			if (mono) { out.ScaP[b] = t.tscap; out.Fix[b] = t.newFix; }
		}
	}
	out.NU = out.u_off[B]; out.NW = out.w_off[B];
	if (alias_passthrough && !in.W_alias && !hook)
	{
		// pass-through maps keep their W blocks in `in` (the caller keeps `in` alive until the join has consumed `out`)
		std::vector<int> delta(B);
		for (int b = 0; b < B; b++) delta[b] = tm[b].active ? INT_MIN : in.w_off[b] - out.w_off[b];
		int* d_alias = ar.alloc<int>(B);
		h2d(ctx, d_alias, delta.data(), B * sizeof(int));
		out.d_alias = d_alias;
		out.W_alias = in.W;
	}
	// the offsets go into the device copy of the map records (a kernel: the copy keeps what the device computed)
	{
		std::vector<int4> offs(B);
		for (int b = 0; b < B; b++) offs[b] = make_int4(tm[b].kU0, tm[b].kW0, tm[b].U0n, tm[b].W0n);
		int4* d_offs = ctx->scratch.alloc<int4>(B);
		h2d(ctx, d_offs, offs.data(), sizeof(int4) * B);
		hipLaunchKernelGGL(k_tr_patch, dim3(std::max(1, (B + 127) / 128)), dim3(128), 0, s, d_tm, d_offs, B, out.fptr + in.NF, out.NW, d_err, (warm && ctx->d_run) ? ctx->d_run : (RunStatsDev*)nullptr);
	}
	if (mono)
	{
		// Imp.cpp:3241-3244: the sign of the new scale is part of the map record the caller keeps
		if (warm)
		{
			for (int b = 0; b < B; b++) if (tm[b].active) out.Sign[b] = plan->tr_sign[b];
			if (ctx->d_run)
			{
				int* d_sign = ctx->scratch.alloc<int>(B);
				h2d(ctx, d_sign, plan->tr_sign.data(), sizeof(int) * B);
				hipLaunchKernelGGL(k_tr_sign_check, dim3((B + 127) / 128), dim3(128), 0, s, d_tm, d_sign, B, ctx->d_run);
			}
		}
		else
		{
			for (int b = 0; b < B; b++) if (tm[b].active && dev_sign[b]) out.Sign[b] = dev_sign[b]; // (read with the counts)
			if (plan) plan->tr_sign = out.Sign;
		}
	}
	out.U = ar.alloc<double>((size_t)out.NU * 36); out.Ui = ar.alloc<int>(out.NU); out.Uj = ar.alloc<int>(out.NU);
	if (!hook) { out.W = ar.alloc<double>((size_t)out.NW * 18); out.photo = ar.alloc<int>(out.NW); out.feature = ar.alloc<int>(out.NW); }
	dev_zero(ctx, out.U, (size_t)out.NU * 36 * sizeof(double)); // the (k,h) slots are accumulated into
	double nw_act_in = 0, nw_act_out = 0, nf_act = 0;
	for (int b = 0; b < B; b++)
		if (tm[b].active)
		{
			nw_act_in += in.w_off[b + 1] - in.w_off[b]; nw_act_out += out.w_off[b + 1] - out.w_off[b];
			nf_act += in.feat_off[b + 1] - in.feat_off[b];
		}
	if (mono) launch_stage<2>(ctx, in, out, d_tm, KU, KW, Dp, Cp, Gpose, PP, nw_act_in, nw_act_out, nf_act, hook);
	else launch_stage<1>(ctx, in, out, d_tm, KU, KW, Dp, Cp, Gpose, PP, nw_act_in, nw_act_out, nf_act, hook);
	ctx->mark("tr_done");
	LSFM_CHECK_HIP(hipGetLastError());
	(void)any;
	ctx->tr_in = nullptr; ctx->tr_hub = nullptr;
	// Scratch is released for the caller's next stage: everything that touches it is ordered on the main stream (the side
	// stream's part rejoins it through evB above).  A first run also stops here so that a failure surfaces at its stage --
	// unless its consumer was laid out inside the hook: that one goes on enqueuing (its solve stops for the device soon
	// enough), so that the host's part of the solve's analysis runs beside these kernels, not after them.
	if (!warm && !hook && !ctx->in_tree_run) LSFM_CHECK_HIP(hipStreamSynchronize(s)); // (a level of a tree run is only enqueued: errors are read at the end of the run)
	if (!hook) ctx->scratch.release(smark); // with a hook its allocations outlive this call: the caller releases
}

} // namespace lsfm
