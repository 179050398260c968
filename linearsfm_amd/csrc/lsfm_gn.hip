// Gauss-Newton polish of the map-joining objective (named in BASELINE.json's north_star).
//
// NO COUNTERPART IN THE REFERENCE -- parity unpinned: the reference joins once, hierarchically, and has no iterative step.
// What is minimised is the objective its joins linearise, over the GLOBAL state x and all N local maps at once,
//
//     F(x) = sum_k || x^_k - f_k(x) ||^2_{I_k}
//
// with f_k the reference's own change of frame (lmj_Transform_PF3DStereo Imp.cpp:421-455, Mono 3268-3306: origin at the map's
// reference pose, Mono: unit = component Fix_k of its scale pose) and its Jacobian as the reference forms it (J1 / J2 / J3,
// Imp.cpp:485-683 / 3383-3688: J = blkdiag(D) + sum_s C_s e_{h_s}^T, "old state with respect to new state at the new state" -- here
// old = frame k, new = the global frame, h_s = the global poses Ref_k [, ScaP_k]).  One step (the tests hold every step against the CPU checker's statement of it):
//
//     H = sum_k J_k^T I_k J_k,   b = sum_k J_k^T I_k r_k,   r_k = x^_k - f_k(x) (angles wrapped),   H d = b,   x += a d
//
// a = 1, halved while F does not fall.  H is laid out as ONE joint map in the library's own layout (U / W sorted by feature / V) whose
// structure -- which local block lands where -- is worked out once per call on the host from the labels; every step then is
//   k_gn_hubs      one lane per map        : R, dR, t, scale of the map's frame at the current state
//   k_gn_poses     one lane per local pose : r_a, D_a, C_s,a
//   k_gn_ublocks   one lane per local U    : U' = D_a^T U D_b into its place, U [C_s r] into the pose rows of G = I [C_s r]
//   k_gn_features  one lane per local feature (consecutive lanes, consecutive features; Stereo: consecutive W blocks): r_f, D_f, C_s,f,
//                  its run of W: W' = D_a^T W D_f and the hub blocks G_s,f^T D_f into their places in the joint W, V' and the
//                  right-hand side per instance, the pose rows of G and the hub-hub sums by wave-level segmented sums + one atomic per
//                  (wave, key, number)
//   k_gn_pose_post / k_gn_hubhub / k_gn_gather_*   the (a, h_s) and (h_s, h_t) blocks, V and the right-hand sides summed per global
//                  variable over its instances in a fixed order
// and the system goes through the same solve_batch as a tree level (pattern of S, K9 panels, supernodal factorisation, refinement,
// back-substitution).  HBM-bound: a step streams the local maps' W once (144 B in + 144 B out per block, + 144 B per hub block) and
// the joint map through the solver.
#include <algorithm>
#include <chrono>

#include "lsfm_device.hpp"
#include "lsfm_internal.hpp"

namespace lsfm {

struct GnMap {
	// structure (uploaded once)
	int nh;        // hub columns: 0 = the map's frame is the global frame (Stereo, Ref_k == the global Ref), 1 Stereo, 2 Mono
	int hub[2];    // global pose of Ref_k [, ScaP_k]
	int fix;       // Mono: Fix_k
	int p0, m, f0, n, u0, nu; // its poses / features / U blocks in the uploaded batch
	int ubase;     // its first block in the joint U: nh * m blocks (a, h_s), nh (nh + 1) / 2 blocks (h_s, h_t), nu own blocks
	// the frame at the current state (k_gn_hubs)
	double R[9], dRA[9], dRB[9], dRG[9], t[3];
	double Scale, Scale2, dSdt[3], dSdtt[3], dSdA, dSdB, dSdG; // (row Fix_k of the reference's dSdt / dSdtt, component Fix_k of dSdA..)
};

__global__ void k_gn_hubs(int N, GnMap* gm, const double* __restrict__ xp)
{
	const int k = blockIdx.x * blockDim.x + threadIdx.x;
	if (k >= N) return;
	GnMap& t = gm[k];
	double R[9] = { 1, 0, 0, 0, 1, 0, 0, 0, 1 }, dRA[9], dRB[9], dRG[9];
	zero<9>(dRA); zero<9>(dRB); zero<9>(dRG);
	t.t[0] = t.t[1] = t.t[2] = 0.0;
	t.Scale = t.Scale2 = 1.0;
	t.dSdA = t.dSdB = t.dSdG = 0.0;
	for (int i = 0; i < 3; i++) { t.dSdt[i] = 0.0; t.dSdtt[i] = 0.0; }
	if (t.nh > 0)
	{
		const double* p = xp + (size_t)t.hub[0] * 6;
		t.t[0] = p[0]; t.t[1] = p[1]; t.t[2] = p[2];
		r_derivation(p[3], p[4], p[5], R, dRA, dRB, dRG);
		if (t.nh == 2)
		{
			// Imp.cpp:3311-3365
			const double* q = xp + (size_t)t.hub[1] * 6;
			double d[3] = { q[0] - p[0], q[1] - p[1], q[2] - p[2] }, ts[3], v[3];
			mv3(R, d, ts);
			const double Sign = ts[t.fix] >= 0 ? 1.0 : -1.0;
			t.Scale = fabs(ts[t.fix]); t.Scale2 = t.Scale * t.Scale;
			for (int c = 0; c < 3; c++) { t.dSdt[c] = -R[3 * t.fix + c] * Sign; t.dSdtt[c] = R[3 * t.fix + c] * Sign; }
			mv3(dRA, d, v); t.dSdA = v[t.fix] * Sign;
			mv3(dRB, d, v); t.dSdB = v[t.fix] * Sign;
			mv3(dRG, d, v); t.dSdG = v[t.fix] * Sign;
		}
	}
	for (int i = 0; i < 9; i++) { t.R[i] = R[i]; t.dRA[i] = dRA[i]; t.dRB[i] = dRB[i]; t.dRG[i] = dRG[i]; }
}

// a position xn of the global state in the map's frame, and the Jacobian of that with respect to xn (a22), the hub's translation and
// angles (adt | tmpc: columns alpha, beta, gamma) and the scale pose's translation (adtt): Imp.cpp:3420-3469 / 3591-3640; Stereo is the
// case Scale = 1, dS = 0 (Imp.cpp:638-680)
__device__ __forceinline__ void gn_trans(const GnMap& t, const double* xn, double* f, double* a22, double* adt, double* tmpc, double* adtt)
{
	double t222[3] = { xn[0] - t.t[0], xn[1] - t.t[1], xn[2] - t.t[2] }, t22[3], v[3];
	mv3(t.R, t222, t22);
#pragma unroll
	for (int r = 0; r < 3; r++) f[r] = t22[r] / t.Scale;
	mv3(t.dRA, t222, v);
#pragma unroll
	for (int r = 0; r < 3; r++) tmpc[3 * r + 0] = (v[r] * t.Scale - t22[r] * t.dSdA) / t.Scale2;
	mv3(t.dRB, t222, v);
#pragma unroll
	for (int r = 0; r < 3; r++) tmpc[3 * r + 1] = (v[r] * t.Scale - t22[r] * t.dSdB) / t.Scale2;
	mv3(t.dRG, t222, v);
#pragma unroll
	for (int r = 0; r < 3; r++) tmpc[3 * r + 2] = (v[r] * t.Scale - t22[r] * t.dSdG) / t.Scale2;
#pragma unroll
	for (int r = 0; r < 3; r++)
#pragma unroll
		for (int c = 0; c < 3; c++)
		{
			a22[3 * r + c] = t.R[3 * r + c] / t.Scale;
			adt[3 * r + c] = (-t.R[3 * r + c] * t.Scale - t22[r] * t.dSdt[c]) / t.Scale2;
			adtt[3 * r + c] = (-t22[r] * t.dSdtt[c]) / t.Scale2;
		}
}

__device__ __forceinline__ double gn_wrap(double a)
{
	const double pi = 3.14159265358979323846;
	while (a > pi) a -= 2 * pi;
	while (a < -pi) a += 2 * pi;
	return a;
}

// per local pose: residual, D (6x6), C_1, C_2 (6x6): Imp.cpp:485-635 / 3383-3584; a local pose that IS a hub collects the hub's
// column in its own (Imp.cpp:3495-3581)
__global__ void __launch_bounds__(128)
k_gn_poses(int P, const int* __restrict__ pose_map, const GnMap* __restrict__ gm, const int* __restrict__ gp, const double* __restrict__ xp,
           const double* __restrict__ xhat, double* __restrict__ Dp, double* __restrict__ Cp, double* __restrict__ rp)
{
	const int a = blockIdx.x * blockDim.x + threadIdx.x;
	if (a >= P) return;
	const GnMap& t = gm[pose_map[a]];
	const double* xn = xp + (size_t)gp[a] * 6;
	const double* xh = xhat + (size_t)a * 6;
	double f[3], a22[9], adt[9], tmpc[9], adtt[9], D[36], C1[36], C2[36];
	gn_trans(t, xn, f, a22, adt, tmpc, adtt);
	double R2[9], dRA2[9], dRB2[9], dRG2[9], Ri[9], dRi[9], dd2[3][3], dd[3][3], al, be, ga;
	r_derivation(xn[3], xn[4], xn[5], R2, dRA2, dRB2, dRG2);
	times_rrt(Ri, R2, t.R);
	inv_rmat_ypr(Ri, al, be, ga);
	double* r = rp + (size_t)a * 6;
	r[0] = xh[0] - f[0]; r[1] = xh[1] - f[1]; r[2] = xh[2] - f[2];
	r[3] = gn_wrap(xh[3] - al); r[4] = gn_wrap(xh[4] - be); r[5] = gn_wrap(xh[5] - ga);
	times_rrt(dRi, dRA2, t.R); ypr_rates<false>(dd2[0], dRi, Ri);
	times_rrt(dRi, dRB2, t.R); ypr_rates<false>(dd2[1], dRi, Ri);
	times_rrt(dRi, dRG2, t.R); ypr_rates<false>(dd2[2], dRi, Ri);
	times_rrt(dRi, R2, t.dRA); ypr_rates<false>(dd[0], dRi, Ri);
	times_rrt(dRi, R2, t.dRB); ypr_rates<false>(dd[1], dRi, Ri);
	times_rrt(dRi, R2, t.dRG); ypr_rates<false>(dd[2], dRi, Ri);
	zero<36>(D); zero<36>(C1); zero<36>(C2);
	const bool hubs = t.nh > 0;
#pragma unroll
	for (int rr = 0; rr < 3; rr++)
#pragma unroll
		for (int c = 0; c < 3; c++)
		{
			D[6 * rr + c] = a22[3 * rr + c];
			D[6 * (3 + rr) + 3 + c] = dd2[c][rr];
			C1[6 * rr + c] = hubs ? adt[3 * rr + c] : 0.0;
			C1[6 * rr + 3 + c] = hubs ? tmpc[3 * rr + c] : 0.0;
			C1[6 * (3 + rr) + 3 + c] = hubs ? dd[c][rr] : 0.0;
			C2[6 * rr + c] = t.nh == 2 ? adtt[3 * rr + c] : 0.0;
		}
	const bool h0 = hubs && gp[a] == t.hub[0], h1 = t.nh == 2 && gp[a] == t.hub[1];
#pragma unroll
	for (int q = 0; q < 36; q++)
	{
		D[q] += (h0 ? C1[q] : 0.0) + (h1 ? C2[q] : 0.0);
		C1[q] = h0 ? 0.0 : C1[q];
		C2[q] = h1 ? 0.0 : C2[q];
	}
	st<36>(Dp + (size_t)a * 36, D);
	st<36>(Cp + (size_t)a * 36, C1);
	st<36>(Cp + (size_t)P * 36 + (size_t)a * 36, C2);
}

// v[0..N) of the lanes that share `key` summed over the wave, one atomic per (key, number) into base[key * stride + i].  All 64 lanes.
template <int N>
__device__ __forceinline__ void wave_key_add(int key, bool valid, const double* v, double* base, int stride)
{
	unsigned long long todo = __ballot(valid);
	const int lane = threadIdx.x & (LSFM_WAVE - 1);
	while (todo)
	{
		const int leader = __ffsll((long long)todo) - 1;
		const int k = __shfl(key, leader, LSFM_WAVE);
		const bool mine = valid && key == k;
#pragma unroll
		for (int i = 0; i < N; i++)
		{
			const double s = wave_sum(mine ? v[i] : 0.0);
			if (lane == leader) atomic_add_f64(base + (size_t)k * stride + i, s);
		}
		todo &= ~__ballot(mine);
	}
}

#define GN_GW 78 /* pose row of G = I [C_1 C_2 r]: 36 + 36 + 6 */
#define GN_HW 121 /* per map: C_1^T G_1, C_1^T G_2, C_2^T G_2 (36 each), C_1^T g, C_2^T g (6 each), r^T g */

// one lane per local U block (a, b): its part of the pose rows of G, and U' = D_a^T U D_b at its place in the joint U
__global__ void __launch_bounds__(128)
k_gn_ublocks(int NU, int P, const int* __restrict__ Ui, const int* __restrict__ Uj, const double* __restrict__ U, const int* __restrict__ pose_map,
             const GnMap* __restrict__ gm, const int* __restrict__ gp, const double* __restrict__ Dp, const double* __restrict__ Cp,
             const double* __restrict__ rp, double* __restrict__ Gacc, double* __restrict__ UJ)
{
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= NU) return;
	const int a = Ui[i], b = Uj[i];
	const GnMap& t = gm[pose_map[a]];
	double Ub[36], X[36], Y[36];
	ld<36>(Ub, U + (size_t)i * 36);
	for (int s = 0; s < 2; s++)
	{
		ld<36>(X, Cp + (size_t)s * P * 36 + (size_t)b * 36);
		mm<6, 6, 6, false>(Ub, X, Y);
		for (int q = 0; q < 36; q++) if (Y[q] != 0.0) atomic_add_f64(Gacc + (size_t)a * GN_GW + 36 * s + q, Y[q]);
		if (a != b)
		{
			ld<36>(X, Cp + (size_t)s * P * 36 + (size_t)a * 36);
			mtm<6, 6, 6, false>(Ub, X, Y);
			for (int q = 0; q < 36; q++) if (Y[q] != 0.0) atomic_add_f64(Gacc + (size_t)b * GN_GW + 36 * s + q, Y[q]);
		}
	}
	{
		double ra[6], rb[6], y[6];
		ld<6>(rb, rp + (size_t)b * 6);
		mm<6, 6, 1, false>(Ub, rb, y);
		for (int q = 0; q < 6; q++) atomic_add_f64(Gacc + (size_t)a * GN_GW + 72 + q, y[q]);
		if (a != b)
		{
			ld<6>(ra, rp + (size_t)a * 6);
			mtm<6, 6, 1, false>(Ub, ra, y);
			for (int q = 0; q < 6; q++) atomic_add_f64(Gacc + (size_t)b * GN_GW + 72 + q, y[q]);
		}
	}
	ld<36>(X, Dp + (size_t)a * 36);
	mtm<6, 6, 6, false>(X, Ub, Y);
	ld<36>(X, Dp + (size_t)b * 36);
	mm<6, 6, 6, false>(Y, X, Ub);
	double* d = UJ + (size_t)(t.ubase + t.nh * t.m + t.nh * (t.nh + 1) / 2 + (i - t.u0)) * 36;
	if (gp[a] <= gp[b]) st<36>(d, Ub);
	else { transpose<6, 6>(Ub, X); st<36>(d, X); }
}

// one lane per local feature (see the header).  NH: 1 Stereo, 2 Mono (a Stereo map in the global frame has t.nh = 0: its C are zero
// and it owns no hub blocks)
template <int NH>
__global__ void __launch_bounds__(256)
k_gn_features(int NF, int P, const int* __restrict__ feat_map, const GnMap* __restrict__ gm, const int* __restrict__ gf, const double* __restrict__ xf,
              const double* __restrict__ fhat, const int* __restrict__ fptr, const int* __restrict__ photo, const double* __restrict__ W,
              const double* __restrict__ V, const double* __restrict__ Dp, const double* __restrict__ Cp, const double* __restrict__ rp,
              const int* __restrict__ wdst, double* __restrict__ WJ, double* __restrict__ Vinst, double* __restrict__ eFinst,
              double* __restrict__ Gacc, double* __restrict__ Hacc)
{
	const int f = blockIdx.x * blockDim.x + threadIdx.x;
	const bool live = f < NF;
	const int fc = live ? f : 0;
	const int k = feat_map[fc];
	const GnMap& t = gm[k];
	const bool hubs = t.nh > 0;
	double fv[3], Df[9], adt[9], tmpc[9], adtt[9], Cf[NH][18], rf[3], Vb[9];
	gn_trans(t, xf + (size_t)gf[fc] * 3, fv, Df, adt, tmpc, adtt);
#pragma unroll
	for (int r = 0; r < 3; r++)
	{
		rf[r] = fhat[(size_t)fc * 3 + r] - fv[r];
#pragma unroll
		for (int c = 0; c < 3; c++)
		{
			Cf[0][6 * r + c] = hubs ? adt[3 * r + c] : 0.0;
			Cf[0][6 * r + 3 + c] = hubs ? tmpc[3 * r + c] : 0.0;
			if (NH == 2) { Cf[NH - 1][6 * r + c] = adtt[3 * r + c]; Cf[NH - 1][6 * r + 3 + c] = 0.0; }
		}
	}
	ld<9>(Vb, V + (size_t)fc * 9);
	double gfv[3], Gf[NH][18];
	mm<3, 3, 1, false>(Vb, rf, gfv);
#pragma unroll
	for (int s = 0; s < NH; s++) mm<3, 3, 6, false>(Vb, Cf[s], Gf[s]);
	const int j0 = fptr[fc], len = live ? fptr[fc + 1] - j0 : 0;
	int maxlen = len;
#pragma unroll
	for (int off = 32; off > 0; off >>= 1) maxlen = max(maxlen, __shfl_xor(maxlen, off, LSFM_WAVE));
	const int dst0 = wdst[fc] + t.nh;
	for (int jj = 0; jj < maxlen; jj++)
	{
		const bool on = jj < len;
		const int j = on ? j0 + jj : j0;
		const int a = photo[j];
		double Wb[18], pc[GN_GW];
		ld<18>(Wb, W + (size_t)j * 18);
		{
			double ra[6], T[18], Da[36];
			ld<6>(ra, rp + (size_t)a * 6);
			if (on) mtm<6, 3, 1, true>(Wb, ra, gfv);
#pragma unroll
			for (int s = 0; s < NH; s++)
			{
				ld<36>(Da, Cp + (size_t)s * P * 36 + (size_t)a * 36);
				if (on) mtm<6, 3, 6, true>(Wb, Da, Gf[s]);
			}
			ld<36>(Da, Dp + (size_t)a * 36);
			mtm<6, 6, 3, false>(Da, Wb, T);
			double Wn[18];
			mm<6, 3, 3, false>(T, Df, Wn);
			if (on) st<18>(WJ + (size_t)(dst0 + jj) * 18, Wn);
		}
		// this block's share of its pose's row of G
		mm<6, 3, 6, false>(Wb, Cf[0], pc);
		if (NH == 2) mm<6, 3, 6, false>(Wb, Cf[NH - 1], pc + 36);
		else zero<36>(pc + 36);
		mm<6, 3, 1, false>(Wb, rf, pc + 72);
		if (NH == 2) wave_key_add<GN_GW>(a, on, pc, Gacc, GN_GW);
		else
		{
			// (Stereo: the second hub column does not exist)
			wave_key_add<36>(a, on, pc, Gacc, GN_GW);
			wave_key_add<6>(a, on, pc + 72, Gacc + 72, GN_GW);
		}
	}
	if (live)
	{
		// the feature's hub blocks G_s,f^T D_f, its V' and right-hand side
#pragma unroll
		for (int s = 0; s < NH; s++)
			if (s < t.nh)
			{
				double Y[18];
				mtm<3, 6, 3, false>(Gf[s], Df, Y);
				st<18>(WJ + (size_t)(wdst[f] + s) * 18, Y);
			}
		double T[9], Vn[9], e[3];
		mtm<3, 3, 3, false>(Df, Vb, T);
		mm<3, 3, 3, false>(T, Df, Vn);
		st<9>(Vinst + (size_t)f * 9, Vn);
		mtm<3, 3, 1, false>(Df, gfv, e);
		st<3>(eFinst + (size_t)f * 3, e);
	}
	// the map's hub-hub sums
	{
		double h[GN_HW];
		mtm<3, 6, 6, false>(Cf[0], Gf[0], h);
		if (NH == 2) { mtm<3, 6, 6, false>(Cf[0], Gf[NH - 1], h + 36); mtm<3, 6, 6, false>(Cf[NH - 1], Gf[NH - 1], h + 72); }
		mtm<3, 6, 1, false>(Cf[0], gfv, h + 108);
		if (NH == 2) mtm<3, 6, 1, false>(Cf[NH - 1], gfv, h + 114);
		h[120] = rf[0] * gfv[0] + rf[1] * gfv[1] + rf[2] * gfv[2];
		if (NH == 2) wave_key_add<GN_HW>(k, live, h, Hacc, GN_HW);
		else
		{
			wave_key_add<36>(k, live, h, Hacc, GN_HW);
			wave_key_add<6>(k, live, h + 108, Hacc + 108, GN_HW);
			wave_key_add<1>(k, live, h + 120, Hacc + 120, GN_HW);
		}
	}
}

// one lane per local pose, after the rows of G are complete: the (a, h_s) blocks D_a^T G_s,a, the pose's right-hand side, its share of
// the hub-hub sums
__global__ void __launch_bounds__(128)
k_gn_pose_post(int P, const int* __restrict__ pose_map, const GnMap* __restrict__ gm, const int* __restrict__ gp, const double* __restrict__ Dp,
               const double* __restrict__ Cp, const double* __restrict__ rp, const double* __restrict__ Gacc, double* __restrict__ UJ,
               double* __restrict__ ePinst, double* __restrict__ Hacc)
{
	const int a = blockIdx.x * blockDim.x + threadIdx.x;
	if (a >= P) return;
	const int k = pose_map[a];
	const GnMap& t = gm[k];
	double D[36], G[2][36], g[6], X[36], e[6];
	ld<36>(D, Dp + (size_t)a * 36);
	ld<36>(G[0], Gacc + (size_t)a * GN_GW); ld<36>(G[1], Gacc + (size_t)a * GN_GW + 36); ld<6>(g, Gacc + (size_t)a * GN_GW + 72);
	for (int s = 0; s < t.nh; s++)
	{
		mtm<6, 6, 6, false>(D, G[s], X);
		double* d = UJ + (size_t)(t.ubase + s * t.m + (a - t.p0)) * 36;
		const int ga = gp[a], h = t.hub[s];
		for (int r = 0; r < 6; r++)
			for (int c = 0; c < 6; c++)
				d[6 * r + c] = ga == h ? X[6 * r + c] + X[6 * c + r] : (ga < h ? X[6 * r + c] : X[6 * c + r]); // a diagonal block in full; row <= column
	}
	mtm<6, 6, 1, false>(D, g, e);
	st<6>(ePinst + (size_t)a * 6, e);
	double* H = Hacc + (size_t)k * GN_HW;
	double C[2][36];
	ld<36>(C[0], Cp + (size_t)a * 36); ld<36>(C[1], Cp + (size_t)P * 36 + (size_t)a * 36);
	if (t.nh > 0)
	{
		mtm<6, 6, 6, false>(C[0], G[0], X);
		for (int q = 0; q < 36; q++) if (X[q] != 0.0) atomic_add_f64(H + q, X[q]);
		mtm<6, 6, 1, false>(C[0], g, e);
		for (int q = 0; q < 6; q++) atomic_add_f64(H + 108 + q, e[q]);
	}
	if (t.nh == 2)
	{
		mtm<6, 6, 6, false>(C[0], G[1], X);
		for (int q = 0; q < 36; q++) if (X[q] != 0.0) atomic_add_f64(H + 36 + q, X[q]);
		mtm<6, 6, 6, false>(C[1], G[1], X);
		for (int q = 0; q < 36; q++) if (X[q] != 0.0) atomic_add_f64(H + 72 + q, X[q]);
		mtm<6, 6, 1, false>(C[1], g, e);
		for (int q = 0; q < 6; q++) atomic_add_f64(H + 114 + q, e[q]);
	}
	double rr[6], dot = 0.0;
	ld<6>(rr, rp + (size_t)a * 6);
	for (int q = 0; q < 6; q++) dot += rr[q] * g[q];
	atomic_add_f64(H + 120, dot);
}

// one lane per map: the hub-hub blocks into the joint U, the map's part of F
__global__ void k_gn_hubhub(int N, const GnMap* __restrict__ gm, const double* __restrict__ Hacc, double* __restrict__ UJ, double* __restrict__ Fsum)
{
	const int k = blockIdx.x * blockDim.x + threadIdx.x;
	if (k >= N) return;
	const GnMap& t = gm[k];
	const double* H = Hacc + (size_t)k * GN_HW;
	atomic_add_f64(Fsum, H[120]);
	if (t.nh == 0) return;
	double* d = UJ + (size_t)(t.ubase + t.nh * t.m) * 36;
	for (int q = 0; q < 36; q++) d[q] = H[q]; // (h_1, h_1)
	if (t.nh == 2)
	{
		const bool up = t.hub[0] < t.hub[1];
		for (int r = 0; r < 6; r++)
			for (int c = 0; c < 6; c++) d[36 + 6 * r + c] = up ? H[36 + 6 * r + c] : H[36 + 6 * c + r]; // (h_1, h_2), row <= column
		for (int q = 0; q < 36; q++) d[72 + q] = H[72 + q]; // (h_2, h_2)
	}
}

// V and the features' right-hand side: every global feature sums its instances in their order
__global__ void k_gn_gather_feat(int NFG, const int* __restrict__ sptr, const int* __restrict__ sidx, const double* __restrict__ Vinst,
                                 const double* __restrict__ eFinst, double* __restrict__ VJ, double* __restrict__ eb)
{
	const int g = blockIdx.x * blockDim.x + threadIdx.x;
	if (g >= NFG) return;
	double v[9], e[3];
	zero<9>(v); zero<3>(e);
	for (int q = sptr[g]; q < sptr[g + 1]; q++)
	{
		const int f = sidx[q];
		for (int i = 0; i < 9; i++) v[i] += Vinst[(size_t)f * 9 + i];
		for (int i = 0; i < 3; i++) e[i] += eFinst[(size_t)f * 3 + i];
	}
	st<9>(VJ + (size_t)g * 9, v);
	st<3>(eb + (size_t)g * 3, e);
}
// the poses' right-hand side: instances (>= 0) and hub roles (-1 - (2 map + s))
__global__ void k_gn_gather_pose(int M, const int* __restrict__ sptr, const int* __restrict__ sidx, const double* __restrict__ ePinst,
                                 const double* __restrict__ Hacc, double* __restrict__ ea)
{
	const int g = blockIdx.x * blockDim.x + threadIdx.x;
	if (g >= M) return;
	double e[6];
	zero<6>(e);
	for (int q = sptr[g]; q < sptr[g + 1]; q++)
	{
		const int sdx = sidx[q];
		const double* src = sdx >= 0 ? ePinst + (size_t)sdx * 6 : Hacc + (size_t)((-1 - sdx) >> 1) * GN_HW + 108 + 6 * ((-1 - sdx) & 1);
		for (int i = 0; i < 6; i++) e[i] += src[i];
	}
	st<6>(ea + (size_t)g * 6, e);
}

// largest |v| over the scalars that are not fixed, as the bits of a non-negative double (they order like the numbers)
__global__ void k_gn_maxabs(size_t n, const double* __restrict__ v, const unsigned char* __restrict__ fixed, unsigned long long* out)
{
	double m = 0.0;
	for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
		if (!(fixed && fixed[i])) { const double a = fabs(v[i]); m = a > m || a != a ? a : m; }
#pragma unroll
	for (int off = 32; off > 0; off >>= 1) { const double o = __shfl_xor(m, off, LSFM_WAVE); m = o > m || o != o ? o : m; }
	if ((threadIdx.x & 63) == 0) atomicMax(out, (unsigned long long)__double_as_longlong(m));
}

__global__ void k_gn_step(size_t n, const double* __restrict__ x0, const double* __restrict__ d, double alpha, const unsigned char* __restrict__ fixed, double* __restrict__ x)
{
	const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
	if (i < n) x[i] = (fixed && fixed[i]) ? x0[i] : x0[i] + alpha * d[i];
}

namespace {
struct Lab { int id, idx; };
int lab_find(const std::vector<Lab>& t, int id)
{
	auto it = std::lower_bound(t.begin(), t.end(), id, [](const Lab& a, int v) { return a.id < v; });
	return (it != t.end() && it->id == id) ? it->idx : -1;
}
inline dim3 grid_for(size_t n, int b) { return dim3((unsigned)std::max<size_t>(1, (n + b - 1) / b)); }
} // namespace

// x: the global state (stno / stVal / m / n, Ref; Mono: ScaP, Fix; optional pose_origin); stVal is updated in place.
// obj / gnorm: [iters + 1]; halvings: [iters] or null.  Returns LSFM_OK or LSFM_NOT_CONVERGED (a step's camera system was left above its
// residual bound); throws Error for invalid input.
int gn_polish(lsfm_context* ctx, const lsfm_map* maps, int N, bool mono, lsfm_map* x, int iters, double* obj, double* gnorm, int* halvings)
{
	hipStream_t s = ctx->stream;
	const int M = x->m, NFG = x->n;
	// LSFM_GN_TIMING=1: wall clock of the call's parts on stderr (every part ends with a synchronisation of its own)
	static const bool timing = getenv("LSFM_GN_TIMING") != nullptr;
	auto wall = []() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
	const double tw0 = wall();
	double t_asm = 0.0, t_solve = 0.0;
	int n_asm = 0, n_solve = 0;
	if (M <= 0 || NFG < 0 || !x->stno || !x->stVal) LSFM_FAIL(LSFM_ERR_ARG, "gn polish: the global state is empty");
	// ---- structure, once per call (host): which global variable every local one is, where every local block lands ----
	std::vector<Lab> pt(M), ft(NFG);
	for (int i = 0; i < M; i++) { if (x->stno[6 * i] > 0) LSFM_FAIL(LSFM_ERR_ARG, "gn polish: state label of a pose must be <= 0"); pt[i] = Lab{ -x->stno[6 * i], i }; }
	for (int i = 0; i < NFG; i++) { if (x->stno[6 * M + 3 * i] <= 0) LSFM_FAIL(LSFM_ERR_ARG, "gn polish: state label of a feature must be > 0"); ft[i] = Lab{ x->stno[6 * M + 3 * i], i }; }
	auto by_id = [](const Lab& a, const Lab& b) { return a.id < b.id; };
	std::sort(pt.begin(), pt.end(), by_id); std::sort(ft.begin(), ft.end(), by_id);
	size_t need = (size_t)128 << 20;
	{
		size_t P = 0, FI = 0, nW = 0, nU = 0;
		for (int k = 0; k < N; k++) { P += maps[k].m; FI += maps[k].n; nW += maps[k].nW; nU += maps[k].nU; }
		need += ((nW + 2 * FI) * 400 + (nU + 3 * P + 3 * (size_t)N) * 800 + (FI + NFG) * 500 + (P + M) * 6000) * 2;
	}
	ctx->ensure_arenas(need);
	ctx->arena[0].reset(); ctx->arena[1].reset(); ctx->scratch.reset();
	Arena& ar = ctx->arena[0];
	DevBatch X;
	batch_upload(ctx, ar, maps, N, mono, X);
	const int P = X.M, FI = X.NF;
	std::vector<GnMap> gm(N);
	std::vector<int> gp(P), gfi(FI), wdst(FI), origin(M, INT32_MAX);
	std::vector<int> fcnt(NFG + 1, 0), pcnt(M + 1, 0);
	int NUJ = 0;
	for (int k = 0; k < N; k++)
	{
		const lsfm_map& L = maps[k];
		GnMap& t = gm[k];
		memset(&t, 0, sizeof t);
		t.p0 = X.pose_off[k]; t.m = L.m; t.f0 = X.feat_off[k]; t.n = L.n; t.u0 = X.u_off[k]; t.nu = L.nU;
		t.hub[0] = t.hub[1] = -1;
		if (!mono && L.Ref == x->Ref) t.nh = 0;
		else
		{
			t.nh = mono ? 2 : 1;
			if ((t.hub[0] = lab_find(pt, L.Ref)) < 0) LSFM_FAIL(LSFM_ERR_ARG, "gn polish: the reference pose of local map " + std::to_string(k + 1) + " is not in the global state");
			if (mono)
			{
				if ((t.hub[1] = lab_find(pt, L.ScaP)) < 0) LSFM_FAIL(LSFM_ERR_ARG, "gn polish: the scale pose of local map " + std::to_string(k + 1) + " is not in the global state");
				if (L.Fix < 0 || L.Fix > 2) LSFM_FAIL(LSFM_ERR_ARG, "gn polish: Fix must be 0, 1 or 2");
				t.fix = L.Fix;
			}
		}
		t.ubase = NUJ;
		NUJ += t.nh * L.m + t.nh * (t.nh + 1) / 2 + L.nU;
		for (int i = 0; i < L.m; i++)
		{
			const int g = lab_find(pt, -L.stno[6 * i]);
			if (g < 0) LSFM_FAIL(LSFM_ERR_ARG, "gn polish: a pose of local map " + std::to_string(k + 1) + " is not in the global state (Stereo: the global reference pose cannot be a variable of a map)");
			gp[t.p0 + i] = g; pcnt[g + 1]++;
			origin[g] = std::min(origin[g], k);
		}
		for (int sdx = 0; sdx < t.nh; sdx++) { pcnt[t.hub[sdx] + 1]++; origin[t.hub[sdx]] = std::min(origin[t.hub[sdx]], k); }
		for (int i = 0; i < L.n; i++)
		{
			const int g = lab_find(ft, L.stno[6 * L.m + 3 * i]);
			if (g < 0) LSFM_FAIL(LSFM_ERR_ARG, "gn polish: a feature of local map " + std::to_string(k + 1) + " is not in the global state");
			gfi[t.f0 + i] = g;
		}
	}
	// joint W: per global feature its instances in map order, an instance's hub block(s) ahead of its own run
	std::vector<int> fptr_loc(FI + 1);
	for (int k = 0; k < N; k++)
	{
		int j = 0; // (W sorted by feature, every feature at least one block: batch_upload has checked)
		for (int f = 0; f < maps[k].n; f++) { fptr_loc[X.feat_off[k] + f] = X.w_off[k] + j; while (j < maps[k].nW && maps[k].feature[j] == f) j++; }
	}
	fptr_loc[FI] = X.NW;
	std::vector<int> fptrJ(NFG + 1, 0);
	for (int k = 0; k < N; k++)
		for (int f = X.feat_off[k]; f < X.feat_off[k + 1]; f++) { fptrJ[gfi[f] + 1] += gm[k].nh + (fptr_loc[f + 1] - fptr_loc[f]); fcnt[gfi[f] + 1]++; }
	for (int g = 0; g < NFG; g++)
	{
		if (!fcnt[g + 1]) LSFM_FAIL(LSFM_ERR_ARG, "gn polish: a feature of the global state is in no local map");
		fptrJ[g + 1] += fptrJ[g]; fcnt[g + 1] += fcnt[g];
	}
	for (int g = 0; g < M; g++)
	{
		if (!pcnt[g + 1]) LSFM_FAIL(LSFM_ERR_ARG, "gn polish: a pose of the global state is in no local map");
		pcnt[g + 1] += pcnt[g];
	}
	const int NWJ = fptrJ[NFG];
	std::vector<int> photoJ(NWJ), fsrc(FI), psrc(pcnt[M]);
	{
		std::vector<int> wcur(fptrJ.begin(), fptrJ.end() - 1), fcur(fcnt.begin(), fcnt.end() - 1), pcur(pcnt.begin(), pcnt.end() - 1);
		for (int k = 0; k < N; k++)
		{
			const GnMap& t = gm[k];
			for (int f = t.f0; f < t.f0 + t.n; f++)
			{
				const int g = gfi[f], len = fptr_loc[f + 1] - fptr_loc[f];
				wdst[f] = wcur[g];
				for (int sdx = 0; sdx < t.nh; sdx++) photoJ[wcur[g]++] = t.hub[sdx];
				for (int j = 0; j < len; j++) photoJ[wcur[g]++] = gp[t.p0 + maps[k].photo[fptr_loc[f] - X.w_off[k] + j]];
				fsrc[fcur[g]++] = f;
			}
			for (int a = t.p0; a < t.p0 + t.m; a++) psrc[pcur[gp[a]]++] = a;
			for (int sdx = 0; sdx < t.nh; sdx++) psrc[pcur[t.hub[sdx]]++] = -1 - (2 * k + sdx);
		}
	}
	// joint U coordinates (row <= column)
	std::vector<int> UiJ(NUJ), UjJ(NUJ);
	for (int k = 0; k < N; k++)
	{
		const GnMap& t = gm[k];
		int q = t.ubase;
		for (int sdx = 0; sdx < t.nh; sdx++)
			for (int a = 0; a < t.m; a++, q++) { const int ga = gp[t.p0 + a], h = t.hub[sdx]; UiJ[q] = std::min(ga, h); UjJ[q] = std::max(ga, h); }
		if (t.nh >= 1) { UiJ[q] = UjJ[q] = t.hub[0]; q++; }
		if (t.nh == 2) { UiJ[q] = std::min(t.hub[0], t.hub[1]); UjJ[q] = std::max(t.hub[0], t.hub[1]); q++; UiJ[q] = UjJ[q] = t.hub[1]; q++; }
		for (int i = 0; i < t.nu; i++, q++)
		{
			const int ga = gp[t.p0 + maps[k].Ui[i]], gb = gp[t.p0 + maps[k].Uj[i]];
			UiJ[q] = std::min(ga, gb); UjJ[q] = std::max(ga, gb);
		}
	}
	if (x->pose_origin) for (int g = 0; g < M; g++) origin[g] = x->pose_origin[g];
	std::vector<unsigned char> fixed;
	if (mono)
	{
		const int pr = lab_find(pt, x->Ref), ps = lab_find(pt, x->ScaP);
		if (pr < 0 || ps < 0 || x->Fix < 0 || x->Fix > 2) LSFM_FAIL(LSFM_ERR_ARG, "gn polish: the global state's reference / scale pose (Ref, ScaP, Fix) is not in it");
		fixed.assign((size_t)M * 6 + (size_t)NFG * 3, 0);
		for (int i = 0; i < 6; i++) fixed[(size_t)pr * 6 + i] = 1;
		fixed[(size_t)ps * 6 + x->Fix] = 1;
	}
	// ---- device arrays ----
	const size_t RS = (size_t)M * 6 + (size_t)NFG * 3;
	GnMap* d_gm = ar.alloc<GnMap>(N);
	int *d_gp = ar.alloc<int>(P), *d_gf = ar.alloc<int>(FI), *d_wdst = ar.alloc<int>(FI), *d_fptrJ = ar.alloc<int>(NFG + 1), *d_photoJ = ar.alloc<int>(NWJ);
	int *d_fsp = ar.alloc<int>(NFG + 1), *d_fsi = ar.alloc<int>(FI), *d_psp = ar.alloc<int>(M + 1), *d_psi = ar.alloc<int>(psrc.size());
	int *d_UiJ = ar.alloc<int>(NUJ), *d_UjJ = ar.alloc<int>(NUJ), *d_org = ar.alloc<int>(M), *d_seg = ar.alloc<int>(M + NFG + 1);
	unsigned char* d_fixed = mono ? ar.alloc<unsigned char>(RS) : nullptr;
	double *d_x = ar.alloc<double>(RS), *d_x0 = ar.alloc<double>(RS), *d_dl = ar.alloc<double>(RS);
	double *Dp = ar.alloc<double>((size_t)P * 36), *Cp = ar.alloc<double>((size_t)P * 72), *rp = ar.alloc<double>((size_t)P * 6);
	double *Gacc = ar.alloc<double>((size_t)P * GN_GW + (size_t)N * GN_HW + 2), *Hacc = Gacc + (size_t)P * GN_GW, *Fsum = Hacc + (size_t)N * GN_HW;
	double *ePinst = ar.alloc<double>((size_t)P * 6), *Vinst = ar.alloc<double>((size_t)FI * 9), *eFinst = ar.alloc<double>((size_t)FI * 3);
	double *UJ = ar.alloc<double>((size_t)NUJ * 36), *WJ = ar.alloc<double>((size_t)NWJ * 18), *VJ = ar.alloc<double>((size_t)NFG * 9);
	double *ea = ar.alloc<double>((size_t)M * 6), *eb = ar.alloc<double>((size_t)NFG * 3);
	unsigned long long* d_max = ar.alloc<unsigned long long>(2);
	h2d(ctx, d_gm, gm.data(), gm.size() * sizeof(GnMap));
	h2d(ctx, d_gp, gp.data(), gp.size() * sizeof(int)); h2d(ctx, d_gf, gfi.data(), gfi.size() * sizeof(int)); h2d(ctx, d_wdst, wdst.data(), wdst.size() * sizeof(int));
	h2d(ctx, d_fptrJ, fptrJ.data(), fptrJ.size() * sizeof(int)); h2d(ctx, d_photoJ, photoJ.data(), photoJ.size() * sizeof(int));
	h2d(ctx, d_fsp, fcnt.data(), fcnt.size() * sizeof(int)); h2d(ctx, d_fsi, fsrc.data(), fsrc.size() * sizeof(int));
	h2d(ctx, d_psp, pcnt.data(), pcnt.size() * sizeof(int)); h2d(ctx, d_psi, psrc.data(), psrc.size() * sizeof(int));
	h2d(ctx, d_UiJ, UiJ.data(), UiJ.size() * sizeof(int)); h2d(ctx, d_UjJ, UjJ.data(), UjJ.size() * sizeof(int));
	h2d(ctx, d_org, origin.data(), origin.size() * sizeof(int));
	if (mono) h2d(ctx, d_fixed, fixed.data(), fixed.size());
	h2d(ctx, d_x, x->stVal, RS * sizeof(double));
	dev_zero(ctx, d_seg, (size_t)(M + NFG + 1) * sizeof(int));

	// F and the step's system at the state in d_x
	auto assemble = [&]() -> double {
		const double ta = wall();
		dev_zero(ctx, Gacc, ((size_t)P * GN_GW + (size_t)N * GN_HW + 2) * sizeof(double));
		hipLaunchKernelGGL(k_gn_hubs, grid_for(N, 128), dim3(128), 0, s, N, d_gm, d_x);
		hipLaunchKernelGGL(k_gn_poses, grid_for(P, 128), dim3(128), 0, s, P, X.pose_map, d_gm, d_gp, d_x, X.pose, Dp, Cp, rp);
		if (X.NU) hipLaunchKernelGGL(k_gn_ublocks, grid_for(X.NU, 128), dim3(128), 0, s, X.NU, P, X.Ui, X.Uj, X.U, X.pose_map, d_gm, d_gp, Dp, Cp, rp, Gacc, UJ);
		if (FI)
		{
			if (mono) hipLaunchKernelGGL((k_gn_features<2>), grid_for(FI, 256), dim3(256), 0, s, FI, P, X.feat_map, d_gm, d_gf, d_x + (size_t)M * 6, X.feat, X.fptr, X.photo, X.W, X.V, Dp, Cp, rp, d_wdst, WJ, Vinst, eFinst, Gacc, Hacc);
			else hipLaunchKernelGGL((k_gn_features<1>), grid_for(FI, 256), dim3(256), 0, s, FI, P, X.feat_map, d_gm, d_gf, d_x + (size_t)M * 6, X.feat, X.fptr, X.photo, X.W, X.V, Dp, Cp, rp, d_wdst, WJ, Vinst, eFinst, Gacc, Hacc);
		}
		hipLaunchKernelGGL(k_gn_pose_post, grid_for(P, 128), dim3(128), 0, s, P, X.pose_map, d_gm, d_gp, Dp, Cp, rp, Gacc, UJ, ePinst, Hacc);
		hipLaunchKernelGGL(k_gn_hubhub, grid_for(N, 128), dim3(128), 0, s, N, d_gm, Hacc, UJ, Fsum);
		if (NFG) hipLaunchKernelGGL(k_gn_gather_feat, grid_for(NFG, 256), dim3(256), 0, s, NFG, d_fsp, d_fsi, Vinst, eFinst, VJ, eb);
		hipLaunchKernelGGL(k_gn_gather_pose, grid_for(M, 128), dim3(128), 0, s, M, d_psp, d_psi, ePinst, Hacc, ea);
		double F = 0.0;
		d2h(ctx, &F, Fsum, sizeof(double));
		t_asm += wall() - ta; n_asm++;
		return F;
	};
	auto grad_norm = [&]() -> double {
		dev_zero(ctx, d_max, 2 * sizeof(unsigned long long));
		hipLaunchKernelGGL(k_gn_maxabs, dim3(256), dim3(256), 0, s, (size_t)M * 6, ea, d_fixed, d_max);
		if (NFG) hipLaunchKernelGGL(k_gn_maxabs, dim3(256), dim3(256), 0, s, (size_t)NFG * 3, eb, (const unsigned char*)nullptr, d_max);
		double v = 0.0;
		d2h(ctx, &v, d_max, sizeof(double));
		return v;
	};
	SolveIO io;
	io.M = M; io.NF = NFG; io.NU = NUJ; io.NW = NWJ; io.nseg = 1;
	io.d_pose_seg = d_seg; io.d_feat_seg = d_seg + M;
	io.U = UJ; io.Ui = d_UiJ; io.Uj = d_UjJ; io.W = WJ; io.photo = d_photoJ; io.fptr = d_fptrJ; io.V = VJ;
	io.ea = ea; io.eb = eb; io.x_pose = d_dl; io.x_feat = d_dl + (size_t)M * 6;
	io.d_fixed = d_fixed; io.d_pose_origin = d_org;
	io.seg_rows.assign(1, M);

	int ret = LSFM_OK;
	bool stopped = false;
	LSFM_CHECK_HIP(hipStreamSynchronize(s));
	const double tw1 = wall();
	double F = assemble();
	for (int it = 0; it <= iters; it++)
	{
		obj[it] = F; gnorm[it] = grad_norm();
		if (it == iters) break;
		if (halvings) halvings[it] = 0;
		if (stopped) continue;
		if (!(F == F)) LSFM_FAIL(LSFM_ERR_INTERNAL, "gn polish: the objective is not a number");
		const size_t smark = ctx->scratch.mark();
		const double ts = wall();
		const int rc = solve_batch(ctx, io);
		LSFM_CHECK_HIP(hipStreamSynchronize(s));
		t_solve += wall() - ts; n_solve++;
		ctx->scratch.release(smark);
		if (rc) ret = LSFM_NOT_CONVERGED;
		LSFM_CHECK_HIP(hipMemcpyAsync(d_x0, d_x, RS * sizeof(double), hipMemcpyDeviceToDevice, s));
		double alpha = 1.0, F1 = 0.0;
		int h = 0;
		for (; h <= 8; h++, alpha *= 0.5)
		{
			hipLaunchKernelGGL(k_gn_step, grid_for(RS, 256), dim3(256), 0, s, RS, d_x0, d_dl, alpha, d_fixed, d_x);
			F1 = assemble(); // (the system at the new state: the next step's, when this one is taken)
			if (F1 <= F + 1e-12 * fabs(F)) break; // (at the minimiser two evaluations differ by their rounding)
		}
		if (h > 8)
		{
			// no decrease along the step: the state stays where it was, the run ends
			LSFM_CHECK_HIP(hipMemcpyAsync(d_x, d_x0, RS * sizeof(double), hipMemcpyDeviceToDevice, s));
			F = assemble();
			stopped = true;
		}
		else F = F1;
		if (halvings) halvings[it] = h > 8 ? 9 : h;
	}
	d2h(ctx, x->stVal, d_x, RS * sizeof(double));
	if (timing)
		fprintf(stderr, "lsfm_gn: %d maps, %d poses, %d features, joint system %d U / %d W blocks; structure + upload %.2f ms, %d assemblies %.2f ms each, "
		                "%d solves %.2f ms each, call %.2f ms\n", N, M, NFG, NUJ, NWJ, tw1 - tw0, n_asm, n_asm ? t_asm / n_asm : 0.0, n_solve,
		        n_solve ? t_solve / n_solve : 0.0, wall() - tw0);
	return ret;
}

} // namespace lsfm
