// The bottom of the tree: levels whose camera systems have at most 16 poses (Stereo: the four lowest levels, 2-16 poses a join --
// thousands of independent joins a level) solved by ONE launch, one work-group per join, the whole camera system in LDS.
//
// What it replaces for such a level (lmj_solveLinearSFM{Stereo,Mono}, Imp.cpp:2119-2378 / 6756-7041, as the level pipeline runs it,
// lsfm_pcg.hip solve_batch): k_vinv, k_schur_u, k_schur_scale, k_schur_slots / lists, k_schur_panel, k_schur_w, k_schur_finish, the
// scatter into the factor's storage, the leaf factorisation, two triangular sweeps, four products with S, the CG bookkeeping
// kernels, k_backsub and ~20 fills / copies -- some 55 launches of which 40 are 5 us of dispatch latency each, on systems of 12-96
// scalars: 0.8 of the 2.1 ms such a level took (profiles/r04 kernel trace).  Here a work-group
//   1. puts the U blocks of its join into a dense S in LDS, the pose part of the right-hand side into E;
//   2. streams the W blocks of its features ONCE, 16 features a pass: V^-1 = L L^T and y = L^T eb per feature in registers, the
//      panel P = [W L] staged in LDS (rows = the join's poses: no slot search, a pose's rows are where its index says), P P^T on
//      v_mfma_f64_16x16x4_f64 with the accumulators kept in registers ACROSS all passes -- every entry of W V^-1 W^T leaves the matrix
//      pipes once per join, not once per 128-feature tile -- and E -= P y;
//   3. factors S = L L^T in place (dense right-looking Cholesky, the original kept in the upper triangle), solves, takes one
//      refinement step against the original S, checks the residual;
//   4. streams the W blocks a second time (they are in L2 / Infinity Cache) for the features: x_f = V^-1 (eb - sum W^T x_p)
//      (pba_solveFeatures, Imp.cpp:2980-3020).
// No sum of it crosses work-groups: the result is the same bits in every run.  Levels with larger systems keep the sparse path.
#include "lsfm_device.hpp"
#include "lsfm_internal.hpp"
#include "lsfm_solve.hpp"

namespace lsfm {

#define SM_PASS 16            /* features per pass */
#define SM_K (3 * SM_PASS)
#define SM_KS (SM_K + 1)      /* odd row stride of the panel */
#define SM_THREADS 256
#define SM_CHUNK 256           /* features whose run pointers are held in LDS */

typedef double sm_v4d __attribute__((ext_vector_type(4)));

struct SmallArgs {
	int nseg;
	const int *pose_off, *feat_off, *u_off; // [nseg + 1] ranges of every join in the batch's global numbering
	const unsigned char* seg_active;        // [nseg] or null; 0: a carried map -- its state is copied through
	const double* U; const int *Ui, *Uj;
	const double* W; const int *photo, *fptr;
	const double* V;
	const double *ea, *eb, *x0;
	const unsigned char* fixed;              // [M * 6] or null: scalars removed from the system (Mono gauge)
	double *x_pose, *x_feat;
	RunStatsDev* run;                        // outcome of the level (may be null)
	int* status;                             // [2]: += systems left above the residual bound, 1 + first system with a non-positive pivot
	double* max_rel;                         // largest relative residual of the level (bit pattern, atomicMax)
};

// NTR 16-row strips: the panel has 16 NTR >= 6 m rows.  LDS: dense S (R x (R + 1)), panel, vectors.
template <int NTR>
struct SmallShared {
	static constexpr int R = 16 * NTR;
	double S[R * (R + 1)];     // row-major, stride R + 1; after the factorisation: L in the lower triangle, the original above it
	double P[R * SM_KS];
	double d0[R];              // the original diagonal
	double E[R], x[R], r[R], v[R];
	double ly[SM_PASS * 9];
	int fpc[SM_CHUNK + 1];     // run pointers of the 256 features the passes are working through (a pass must not wait for them)
	unsigned char fx[R];
	int bad;
	int strips; // bit i: a row of 16-row strip i of the panel was staged in this pass
};

template <int NTR>
__global__ void __launch_bounds__(SM_THREADS, NTR >= 6 ? 1 : (NTR >= 3 ? 2 : 4))
k_small_solve(SmallArgs a)
{
	constexpr int R = 16 * NTR, LD = R + 1;
	constexpr int NTILE = NTR * (NTR + 1) / 2, NW = SM_THREADS / 64, T = (NTILE + NW - 1) / NW;
	extern __shared__ double sm_raw[];
	SmallShared<NTR>& sh = *reinterpret_cast<SmallShared<NTR>*>(sm_raw);
	const int g = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
	const int p0 = a.pose_off[g], m = a.pose_off[g + 1] - p0, n = 6 * m;
	const int f0 = a.feat_off[g], f1 = a.feat_off[g + 1];
	if (a.seg_active && !a.seg_active[g])
	{
		// a carried map: its poses keep their values (its features were written by the join)
		for (int i = tid; i < n; i += SM_THREADS) a.x_pose[(size_t)p0 * 6 + i] = a.x0 ? a.x0[(size_t)p0 * 6 + i] : 0.0;
		return;
	}
	// ---- 1. S = U, E = ea ----------------------------------------------------------------------------------------------------
	for (int i = tid; i < R * LD; i += SM_THREADS) sh.S[i] = 0.0;
	for (int i = tid; i < R; i += SM_THREADS)
	{
		sh.E[i] = i < n ? a.ea[(size_t)p0 * 6 + i] : 0.0;
		sh.fx[i] = (i >= n || (a.fixed && a.fixed[(size_t)p0 * 6 + i])) ? 1 : 0;
	}
	if (tid == 0) sh.bad = 0;
	__syncthreads();
	{
		const int u0 = a.u_off[g], u1 = a.u_off[g + 1];
		// (a block per 36 lanes; duplicates of a pair add up -- two addends commute, more are not met on this path)
		for (int q = tid; q < (u1 - u0) * 36; q += SM_THREADS)
		{
			const int i = u0 + q / 36, e = q % 36, r = e / 6, c = e % 6;
			const int pa = a.Ui[i] - p0, pb = a.Uj[i] - p0;
			const double v = a.U[(size_t)i * 36 + e];
			// stored (a <= b): block (a, b); the matrix is symmetric: the mirrored entry too (a diagonal block is stored full)
			lds_add_f64(&sh.S[(6 * pa + r) * LD + 6 * pb + c], v);
			if (pa != pb) lds_add_f64(&sh.S[(6 * pb + c) * LD + 6 * pa + r], v);
		}
	}
	// ---- 2. S -= W V^-1 W^T, E -= W V^-1 eb: the panel on the matrix cores ------------------------------------------------------------
	int ti[T], tj[T];
	sm_v4d acc[T];
#pragma unroll
	for (int t = 0; t < T; t++)
	{
		const int q = wave + NW * t; // q = j (j + 1) / 2 + i, i <= j
		int i = 0, j = 0;
		if (q < NTILE)
		{
			while ((j + 1) * (j + 2) / 2 <= q) j++;
			i = q - j * (j + 1) / 2;
		}
		ti[t] = q < NTILE ? i : -1; tj[t] = j;
		acc[t] = (sm_v4d){ 0.0, 0.0, 0.0, 0.0 };
	}
	constexpr int NE = (NTR + NW - 1) / NW;
	double eacc[NE];
#pragma unroll
	for (int i = 0; i < NE; i++) eacc[i] = 0.0;
	const int lbase = (lane & 15) * SM_KS + (lane >> 4);
	// what a pass reads from memory is fetched one pass ahead: the W rows of a pass are ONE contiguous range (3 doubles a row), its
	// features' V and eb 12 doubles each
	constexpr int PF = 3; // rows per lane held in registers (768 rows = 8 blocks per feature); longer passes read the rest late
	double pw[PF][3];
	int pk[PF];
	double pv[9], pe[3];
	int qb0 = 0, RW = 0, c0 = f0;
	auto prefetch = [&](int q0) {
		const int nf = min(SM_PASS, f1 - q0);
		if (q0 - c0 >= SM_CHUNK || q0 == f0)
		{
			// (every 16th pass: the staging of the pass before is done with the old pointers)
			__syncthreads();
			c0 = q0;
			for (int i = tid; i <= min(SM_CHUNK, f1 - c0); i += SM_THREADS) sh.fpc[i] = a.fptr[c0 + i];
			__syncthreads();
		}
		qb0 = sh.fpc[q0 - c0];
		RW = (sh.fpc[q0 - c0 + nf] - qb0) * 6;
		if (tid < nf)
		{
			ld<9>(pv, a.V + (size_t)(q0 + tid) * 9);
			ld<3>(pe, a.eb + (size_t)(q0 + tid) * 3);
		}
		const double* wb = a.W + (size_t)qb0 * 18;
#pragma unroll
		for (int i = 0; i < PF; i++)
		{
			const int w = tid + SM_THREADS * i;
			if (w < RW) { pw[i][0] = wb[3 * (size_t)w]; pw[i][1] = wb[3 * (size_t)w + 1]; pw[i][2] = wb[3 * (size_t)w + 2]; pk[i] = a.photo[qb0 + w / 6]; }
		}
	};
	if (f0 < f1) prefetch(f0);
	for (int q0 = f0; q0 < f1; q0 += SM_PASS)
	{
		const int nf = min(SM_PASS, f1 - q0);
		__syncthreads(); // the pass before is consumed
		for (int q = tid; q < R * SM_KS; q += SM_THREADS) sh.P[q] = 0.0;
		if (tid == SM_THREADS - 1) sh.strips = 0;
		if (tid < SM_PASS)
		{
			// V^-1 = L L^T, y = L^T eb (k_vinv's arithmetic)
			double l[9];
#pragma unroll
			for (int i = 0; i < 9; i++) l[i] = 0.0;
			if (tid < nf)
			{
				double o[9];
				inv3_sym(pv, o);
				const double d0 = o[0];
				l[0] = sqrt(d0); l[1] = o[3] / l[0]; l[3] = o[6] / l[0];
				const double d1 = o[4] - l[1] * l[1];
				l[2] = sqrt(d1); l[4] = (o[7] - l[3] * l[1]) / l[2];
				const double d2 = o[8] - l[3] * l[3] - l[4] * l[4];
				l[5] = sqrt(d2);
				l[6] = l[0] * pe[0] + l[1] * pe[1] + l[3] * pe[2];
				l[7] = l[2] * pe[1] + l[4] * pe[2];
				l[8] = l[5] * pe[2];
				if (!(d0 > 0.0) || !(d1 > 0.0) || !(d2 > 0.0)) sh.bad = 1; // V of a feature is not positive definite
			}
			st<9>(&sh.ly[tid * 9], l);
		}
		__syncthreads();
		{
			// stage P = W L; a (pose, feature) pair may hold two blocks (the joins keep both when a feature was seen from the hub
			// pose on either side, Imp.cpp:1277): they add up
			const double* wb = a.W + (size_t)qb0 * 18;
			auto stage = [&](int w, double w0, double w1, double w2, int k) {
				const int e = w / 6, r = w - 6 * e, j = qb0 + e;
				int lo = 0, hi = nf - 1; // feature of block j: the last run of the pass that starts at or before it
				while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (sh.fpc[q0 - c0 + mid] <= j) lo = mid; else hi = mid - 1; }
				const double* l = &sh.ly[lo * 9];
				const int row = 6 * (k - p0) + r;
				double* d = &sh.P[row * SM_KS + 3 * lo];
				if (r == 0 || (row & 15) == 0) atomicOr(&sh.strips, 1 << (row >> 4)); // (a block's six rows lie in one strip or two)
				lds_add_f64(d + 0, w0 * l[0] + w1 * l[1] + w2 * l[3]);
				lds_add_f64(d + 1, w1 * l[2] + w2 * l[4]);
				lds_add_f64(d + 2, w2 * l[5]);
			};
#pragma unroll
			for (int i = 0; i < PF; i++)
			{
				const int w = tid + SM_THREADS * i;
				if (w < RW) stage(w, pw[i][0], pw[i][1], pw[i][2], pk[i]);
			}
			for (int w = tid + SM_THREADS * PF; w < RW; w += SM_THREADS) stage(w, wb[3 * (size_t)w], wb[3 * (size_t)w + 1], wb[3 * (size_t)w + 2], a.photo[qb0 + w / 6]);
		}
		__syncthreads();
		if (q0 + SM_PASS < f1) prefetch(q0 + SM_PASS);
		// The panel has a row for every scalar of the join's poses, and a pass of 16 features is seen by a few of them: only the
		// 16-row strips a block of this pass was staged into are worked on (with all strips, level 3 of an NC3500-like tree -- 16
		// poses, 1.7 blocks a feature -- spent 2.2 us a pass in 72 matrix instructions a wave, nearly all of them on zeros)
		const int smask = sh.strips;
		// E -= P y (Imp.cpp:2321-2328): 16-row strips over the waves, a lane = (row of the strip, quarter of the 48 columns)
		{
			const int kq = lane >> 4;
#pragma unroll
			for (int si = 0; si < NE; si++)
			{
				const int strip = wave + NW * si;
				if (strip < NTR && ((smask >> strip) & 1))
				{
					const double* pr = &sh.P[(16 * strip + (lane & 15)) * SM_KS + 12 * kq];
					const double* yq = &sh.ly[kq * 36 + 6];
					double s0 = 0.0, s1 = 0.0;
#pragma unroll
					for (int k = 0; k < 12; k += 2)
					{
						s0 = fma(pr[k], yq[(k / 3) * 9 + k % 3], s0);
						s1 = fma(pr[k + 1], yq[((k + 1) / 3) * 9 + (k + 1) % 3], s1);
					}
					double sum = s0 + s1;
					sum += __shfl_xor(sum, 16, 64);
					sum += __shfl_xor(sum, 32, 64);
					eacc[si] -= sum;
				}
			}
		}
		// P P^T: lane l feeds A[row l & 15][k = l >> 4] and B[k = l >> 4][col l & 15]
#pragma unroll 2
		for (int ks = 0; ks < SM_K / 4; ks++)
		{
#pragma unroll
			for (int t = 0; t < T; t++)
			{
				if (ti[t] < 0 || !((smask >> ti[t]) & (smask >> tj[t]) & 1)) continue; // (wave-uniform)
				const double av = sh.P[16 * ti[t] * SM_KS + lbase + 4 * ks], bv = sh.P[16 * tj[t] * SM_KS + lbase + 4 * ks];
				acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc[t], 0, 0, 0);
			}
		}
	}
	__syncthreads();
	// the accumulators into S: C/D of v_mfma_f64_16x16x4_f64: col = lane & 15, row = (lane >> 4) + 4 * reg.  Every entry of the lower
	// AND the upper triangle has exactly one writer (tile (i, j), i <= j, holds rows of strip i against columns of strip j)
#pragma unroll
	for (int t = 0; t < T; t++)
	{
		if (ti[t] < 0) continue;
		const int C = 16 * tj[t] + (lane & 15);
#pragma unroll
		for (int e = 0; e < 4; e++)
		{
			const int Rr = 16 * ti[t] + (lane >> 4) + 4 * e;
			const double v = acc[t][e];
			sh.S[Rr * LD + C] -= v;
			if (ti[t] != tj[t]) sh.S[C * LD + Rr] -= v;
		}
	}
	if (lane < 16)
#pragma unroll
		for (int si = 0; si < NE; si++)
		{
			const int strip = wave + NW * si;
			if (strip < NTR) sh.E[16 * strip + lane] += eacc[si];
		}
	__syncthreads();
	// scalars that are not part of the system (the Mono gauge; the padding beyond 6 m): identity rows, zero right-hand side
	for (int q = tid; q < R * R; q += SM_THREADS)
	{
		const int i = q / R, j = q - i * R;
		if (sh.fx[i] || sh.fx[j]) sh.S[i * LD + j] = i == j ? 1.0 : 0.0;
	}
	for (int i = tid; i < R; i += SM_THREADS) if (sh.fx[i]) sh.E[i] = 0.0;
	__syncthreads();
	for (int i = tid; i < R; i += SM_THREADS) sh.d0[i] = sh.S[i * LD + i];
	__syncthreads();
	// ---- 3. S = L L^T in place: right-looking, column by column; L ends up in the lower triangle (with its diagonal), the upper
	// triangle and d0 keep the original ------------------------------------------------------------------------------------------
	int pivot_bad = -1;
	for (int j = 0; j < n; j++)
	{
		const double d = sh.S[j * LD + j];
		if (!(d > 0.0) || !(d < 1e300)) { pivot_bad = j; break; } // (uniform: every thread reads the same entry)
		const double piv = sqrt(d), ip = 1.0 / piv;
		__syncthreads();
		for (int i = j + tid; i < n; i += SM_THREADS) sh.S[i * LD + j] = i == j ? piv : sh.S[i * LD + j] * ip;
		__syncthreads();
		// trailing update of the lower triangle: (i, k), j < k <= i
		const int nt = n - j - 1;
		for (int q = tid; q < nt * nt; q += SM_THREADS)
		{
			const int ii = q / nt, kk = q - ii * nt;
			if (kk <= ii)
			{
				const int i = j + 1 + ii, k = j + 1 + kk;
				sh.S[i * LD + k] = fma(-sh.S[i * LD + j], sh.S[k * LD + j], sh.S[i * LD + k]);
			}
		}
		__syncthreads();
	}
	if (pivot_bad >= 0)
	{
		if (tid == 0) { atomicCAS(&a.status[1], 0, 1 + g); if (a.run && !a.run->chol_err) a.run->chol_err = 1 + p0 + pivot_bad / 6; }
		return;
	}
	// x = S^-1 E, then one refinement step against the original S (upper triangle + d0).  One wave does the triangular sweeps: row i
	// of the solution is final before row i + 1 needs it -- a chain, whatever the number of lanes
	auto solve = [&](const double* rhs, double* out) {
		// forward: L v = rhs ; backward: L^T out = v
		for (int i = tid; i < R; i += SM_THREADS) sh.v[i] = i < n ? rhs[i] : 0.0;
		__syncthreads();
		if (wave == 0)
		{
			for (int j = 0; j < n; j++)
			{
				const double vj = sh.v[j] / sh.S[j * LD + j];
				__builtin_amdgcn_wave_barrier();
				if (lane == 0) sh.v[j] = vj;
				for (int i = j + 1 + lane; i < n; i += 64) sh.v[i] = fma(-sh.S[i * LD + j], vj, sh.v[i]);
				__builtin_amdgcn_wave_barrier();
			}
			for (int j = n - 1; j >= 0; j--)
			{
				const double vj = sh.v[j] / sh.S[j * LD + j];
				__builtin_amdgcn_wave_barrier();
				if (lane == 0) sh.v[j] = vj;
				for (int i = lane; i < j; i += 64) sh.v[i] = fma(-sh.S[j * LD + i], vj, sh.v[i]);
				__builtin_amdgcn_wave_barrier();
			}
		}
		__syncthreads();
		for (int i = tid; i < R; i += SM_THREADS) out[i] = sh.v[i];
		__syncthreads();
	};
	auto residual = [&](const double* xx) {
		// r = E - S x with the ORIGINAL S: entry (i, j) = upper triangle for i < j, d0 on the diagonal, mirrored below
		for (int i = tid; i < n; i += SM_THREADS)
		{
			double s = sh.E[i] - sh.d0[i] * xx[i];
			for (int j = 0; j < n; j++)
				if (j != i) s = fma(-(i < j ? sh.S[i * LD + j] : sh.S[j * LD + i]), xx[j], s);
			sh.r[i] = s;
		}
		__syncthreads();
	};
	solve(sh.E, sh.x);
	residual(sh.x);
	solve(sh.r, sh.v);      // (out aliases the work vector: harmless, see the copies in solve)
	for (int i = tid; i < n; i += SM_THREADS) sh.x[i] += sh.v[i];
	__syncthreads();
	residual(sh.x);
	if (tid == 0)
	{
		double rr = 0.0, ee = 0.0;
		for (int i = 0; i < n; i++) { rr += sh.r[i] * sh.r[i]; ee += sh.E[i] * sh.E[i]; }
		const double rel = ee > 0.0 ? sqrt(rr / ee) : 0.0;
		if (!(rel < 1e-8) || sh.bad)
		{
			atomicAdd(&a.status[0], 1);
			if (a.run) atomicAdd(&a.run->not_converged, 1);
		}
		const unsigned long long bits = (unsigned long long)__double_as_longlong(rel == rel ? rel : 1e300);
		atomicMax(reinterpret_cast<unsigned long long*>(a.max_rel), bits);
		if (a.run) atomicMax(reinterpret_cast<unsigned long long*>(&a.run->max_rel_residual), bits);
	}
	for (int i = tid; i < n; i += SM_THREADS) a.x_pose[(size_t)p0 * 6 + i] = sh.fx[i] ? 0.0 : sh.x[i];
	// ---- 4. the features: x_f = V^-1 (eb - sum_p W_pf^T x_p), one lane per feature (its run is a handful of blocks) ---------------------
	for (int f = f0 + tid; f < f1; f += SM_THREADS)
	{
		double V[9], iv[9];
		ld<9>(V, a.V + (size_t)f * 9);
		inv3_sym(V, iv);
		double d[3] = { a.eb[(size_t)f * 3], a.eb[(size_t)f * 3 + 1], a.eb[(size_t)f * 3 + 2] };
		double s3[3] = { 0.0, 0.0, 0.0 };
		for (int j = a.fptr[f]; j < a.fptr[f + 1]; j++)
		{
			double w[18];
			ld<18>(w, a.W + (size_t)j * 18);
			const double* xp = &sh.x[6 * (a.photo[j] - p0)];
#pragma unroll
			for (int c = 0; c < 3; c++)
			{
				double sacc = 0.0;
#pragma unroll
				for (int r = 0; r < 6; r++) sacc = fma(w[3 * r + c], sh.fx[6 * (a.photo[j] - p0) + r] ? 0.0 : xp[r], sacc);
				s3[c] += sacc;
			}
		}
		d[0] -= s3[0]; d[1] -= s3[1]; d[2] -= s3[2];
		for (int r = 0; r < 3; r++) a.x_feat[(size_t)f * 3 + r] = iv[3 * r] * d[0] + iv[3 * r + 1] * d[1] + iv[3 * r + 2] * d[2];
	}
}

// strips of 16 rows that hold 6 * most scalars; 0: too large for this path
int small_solve_strips(int most_poses)
{
	if (most_poses <= 2) return 1;
	if (most_poses <= 5) return 2;
	if (most_poses <= 8) return 3;
	if (most_poses <= 16) return 6;
	return 0;
}

template <int NTR>
static void launch_small(hipStream_t s, const SmallArgs& a)
{
	const size_t lds = sizeof(SmallShared<NTR>);
	static const bool set = []() {
		(void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_small_solve<NTR>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(SmallShared<NTR>));
		return true;
	}();
	(void)set;
	hipLaunchKernelGGL(k_small_solve<NTR>, dim3(a.nseg), dim3(SM_THREADS), lds, s, a);
}

// status: device int[2] + double[1] (zeroed by the caller).  Enqueues only.
void small_solve_launch(lsfm_context* ctx, const SolveIO& io, int strips, int* status, double* max_rel)
{
	SmallArgs a;
	a.nseg = io.nseg; a.pose_off = io.d_pose_off; a.feat_off = io.d_feat_off; a.u_off = io.d_u_off; a.seg_active = io.d_seg_active;
	a.U = io.U; a.Ui = io.Ui; a.Uj = io.Uj; a.W = io.W; a.photo = io.photo; a.fptr = io.fptr; a.V = io.V;
	a.ea = io.ea; a.eb = io.eb; a.x0 = io.x0; a.fixed = io.d_fixed; a.x_pose = io.x_pose; a.x_feat = io.x_feat;
	a.run = (ctx->in_tree_run && ctx->d_run) ? ctx->d_run : nullptr;
	a.status = status; a.max_rel = max_rel;
	if (!io.nseg) return;
	switch (strips)
	{
	case 1: launch_small<1>(ctx->stream, a); break;
	case 2: launch_small<2>(ctx->stream, a); break;
	case 3: launch_small<3>(ctx->stream, a); break;
	default: launch_small<6>(ctx->stream, a); break;
	}
	LSFM_CHECK_HIP(hipGetLastError());
}

} // namespace lsfm
