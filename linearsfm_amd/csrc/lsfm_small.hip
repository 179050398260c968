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
#define SM_SUPER 64            /* features per super-pass (4 passes) */
#define SM_WROWS 1024          /* W rows of a super-pass held in LDS: 170 blocks = 2.7 a feature (the low levels have 1.2-1.8); the rest is read from memory */

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
	int dbg;                                 // LSFM_SMALL_DEBUG (timing probes): bits switch phases OFF -- results are then garbage
};

// NTR 16-row strips: the panel has 16 NTR >= 6 m rows.  LDS: dense S (R x (R + 1)), panel, vectors.
template <int NTR>
struct SmallShared {
	static constexpr int R = 16 * NTR;
	double S[R * (R + 1)];     // row-major, stride R + 1; after the factorisation: L in the lower triangle, the original above it
	double P[R * SM_KS];
	double d0[R];              // the original diagonal
	double E[R], x[R], r[R], v[R], dinv[R]; // dinv: 1 / diagonal of the factor
	double ly[SM_SUPER * 9];   // per feature of the super-pass: l00 l10 l11 l20 l21 l22 of V^-1 = L L^T, then y = L^T eb; zero past the last one
	double wst[SM_WROWS * 3];  // the W rows of the super-pass (row w of block e: 6 e + r)
	int sph[SM_WROWS / 6 + 1]; // pose (join-local) of every block of the super-pass
	int fpc[2][SM_CHUNK + 1];  // run pointers of the 256 features the passes are working through (a pass must not wait for them); two chunks:
	                           // the last super-pass of one is still at work when the first of the next is prefetched
	unsigned char fx[R];
	int bad;
	int strips[4]; // per pass of the super-pass: bit i = a block lies in 16-row strip i of the panel
};

template <int NTR>
#ifndef LSFM_SMALL_OCC
#define LSFM_SMALL_OCC 2 /* work-groups per CU of the 2- and 5-pose instances (levels 0 and 1 of a Stereo tree): 3 = 168 registers, 76-104 bytes spilled */
#endif
__global__ void __launch_bounds__(SM_THREADS, NTR >= 6 ? 1 : (NTR <= 2 ? LSFM_SMALL_OCC : 2)) // (the prefetched super-pass is ~45 registers: three waves a SIMD would spill 160)
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
	// The features are worked through in SUPER-PASSES of SM_SUPER = 64 (four passes of 16 columns of the panel).  What a super-pass
	// reads from memory -- its W rows, ONE contiguous range of 3 doubles a row, the photo of every block, V and eb of its features --
	// is fetched into registers a whole super-pass ahead and parked in LDS when its turn comes: one exposed memory latency per 64
	// features, not per 16 (the first version prefetched pass by pass and spent a third of its time waiting).
	constexpr int PF = SM_WROWS / SM_THREADS; // rows per lane held in registers; a super-pass with more rows reads the rest from memory
	constexpr int MCAP = R / 6;               // poses the panel has rows for
	double pw[PF][3];
	int pk[PF];
	double pv[9], pe[3];
	int qbn = 0, c0 = f0, cb = 1; // first block of the prefetched super-pass; first feature / buffer of the run-pointer chunk it lies in
	auto prefetch = [&](int q0) {
		const int nf = min(SM_SUPER, f1 - q0);
		if (q0 - c0 >= SM_CHUNK || q0 == f0)
		{
			// (every fourth super-pass, into the OTHER buffer: the super-pass at work still reads the chunk before)
			c0 = q0; cb ^= 1;
			for (int i = tid; i <= min(SM_CHUNK, f1 - c0); i += SM_THREADS) sh.fpc[cb][i] = a.fptr[c0 + i];
			__syncthreads();
		}
		qbn = sh.fpc[cb][q0 - c0];
		const int rw = (sh.fpc[cb][q0 - c0 + nf] - qbn) * 6;
		if (tid < nf)
		{
			ld<9>(pv, a.V + (size_t)(q0 + tid) * 9);
			ld<3>(pe, a.eb + (size_t)(q0 + tid) * 3);
		}
		const double* wb = a.W + (size_t)qbn * 18;
#pragma unroll
		for (int i = 0; i < PF; i++)
		{
			const int w = tid + SM_THREADS * i;
			if (w < rw) { pw[i][0] = wb[3 * (size_t)w]; pw[i][1] = wb[3 * (size_t)w + 1]; pw[i][2] = wb[3 * (size_t)w + 2]; pk[i] = a.photo[qbn + w / 6]; }
		}
	};
	for (int i = tid; i < R * SM_KS; i += SM_THREADS) sh.P[i] = 0.0; // (rows beyond the last pose's are never written: they stay zero)
	if (f0 < f1) prefetch(f0);
	for (int s0 = f0; s0 < f1; s0 += SM_SUPER)
	{
		const int nfs = min(SM_SUPER, f1 - s0);
		const int qbs = qbn, cs = c0, bs = cb; // this super-pass: first block, chunk (start, buffer) its run pointers are in
		const int rws = (sh.fpc[bs][s0 - cs + nfs] - qbs) * 6;
		__syncthreads(); // the super-pass before is consumed
		// ---- park the rows, the photos and (V^-1 = L L^T, y = L^T eb) of the 64 features in LDS ----
#pragma unroll
		for (int i = 0; i < PF; i++)
		{
			const int w = tid + SM_THREADS * i;
			if (w < rws)
			{
				sh.wst[3 * w] = pw[i][0]; sh.wst[3 * w + 1] = pw[i][1]; sh.wst[3 * w + 2] = pw[i][2];
				if (w % 6 == 0) sh.sph[w / 6] = pk[i] - p0;
			}
		}
		if (tid < 4) sh.strips[tid] = 0;
		if (tid < SM_SUPER)
		{
			// (k_vinv's arithmetic; 64 lanes at once)
			double l[9];
#pragma unroll
			for (int i = 0; i < 9; i++) l[i] = 0.0;
			if (tid < nfs && !(a.dbg & 1))
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
		if (s0 + SM_SUPER < f1) prefetch(s0 + SM_SUPER); // (in flight during the four passes below)
		for (int sub = 0; sub * SM_PASS < nfs; sub++)
		{
			const int q0 = s0 + sub * SM_PASS, nf = min(SM_PASS, f1 - q0);
			// ---- the panel P = [W L] of the pass: one lane per (pose, feature) cell.  It walks the feature's run for the blocks
			// of its pose -- the joins keep both blocks when a feature was seen from the hub pose on either side (Imp.cpp:1277):
			// they add up, in run order -- and stores its 6 x 3 cell, zeros included: no fill of the panel, no LDS atomics (an LDS
			// add of a double is serialised lane by lane on this chip: staging by row with atomics was 1.4 of the first version's
			// 3.7 ms per tree) ----
			{
				const int lp = tid >> 4, fl = tid & 15;
				bool found = false;
				if (lp < MCAP && !(a.dbg & 2))
				{
					double c18[18];
#pragma unroll
					for (int i = 0; i < 18; i++) c18[i] = 0.0;
					if (fl < nf && lp < m)
					{
						const int j0 = sh.fpc[bs][q0 - cs + fl], j1 = sh.fpc[bs][q0 - cs + fl + 1];
						const double* l = &sh.ly[(sub * SM_PASS + fl) * 9];
						const double l0 = l[0], l1 = l[1], l2 = l[2], l3 = l[3], l4 = l[4], l5 = l[5];
						for (int j = j0; j < j1; j++)
						{
							const int e = j - qbs;
							const int ph = 6 * e < SM_WROWS ? sh.sph[e] : a.photo[j] - p0;
							if (ph != lp) continue;
							found = true;
#pragma unroll
							for (int r = 0; r < 6; r++)
							{
								double w0, w1, w2;
								if (6 * e + 5 < SM_WROWS) { w0 = sh.wst[3 * (6 * e + r)]; w1 = sh.wst[3 * (6 * e + r) + 1]; w2 = sh.wst[3 * (6 * e + r) + 2]; }
								else { const double* wg = a.W + (size_t)j * 18 + 3 * r; w0 = wg[0]; w1 = wg[1]; w2 = wg[2]; }
								c18[3 * r] += w0 * l0 + w1 * l1 + w2 * l3;
								c18[3 * r + 1] += w1 * l2 + w2 * l4;
								c18[3 * r + 2] += w2 * l5;
							}
						}
					}
#pragma unroll
					for (int r = 0; r < 6; r++)
					{
						double* d = &sh.P[(6 * lp + r) * SM_KS + 3 * fl];
						d[0] = c18[3 * r]; d[1] = c18[3 * r + 1]; d[2] = c18[3 * r + 2];
					}
				}
				// which 16-row strips of the panel hold anything in this pass: one LDS OR per wave and strip
				const int b0 = (6 * lp) >> 4, b1 = (6 * lp + 5) >> 4;
#pragma unroll
				for (int sidx = 0; sidx < NTR; sidx++)
					if (__ballot(found && (b0 == sidx || b1 == sidx)) != 0ull && lane == 0) atomicOr(&sh.strips[sub], 1 << sidx);
			}
			__syncthreads();
			// The panel has a row for every scalar of the join's poses, and a pass of 16 features is seen by a few of them: only the
			// 16-row strips that hold a block of this pass are worked on (with all strips, level 3 of an NC3500-like tree -- 16
			// poses, 1.7 blocks a feature -- spent 2.2 us a pass in 72 matrix instructions a wave, nearly all of them on zeros)
			const int smask = sh.strips[sub];
			// E -= P y (Imp.cpp:2321-2328): 16-row strips over the waves, a lane = (row of the strip, quarter of the 48 columns)
			{
				const int kq = lane >> 4;
#pragma unroll
				for (int si = 0; si < NE; si++)
				{
					const int strip = wave + NW * si;
					if (strip < NTR && ((smask >> strip) & 1) && !(a.dbg & 4))
					{
						const double* pr = &sh.P[(16 * strip + (lane & 15)) * SM_KS + 12 * kq];
						const double* yq = &sh.ly[sub * SM_PASS * 9 + kq * 36 + 6];
						double e0 = 0.0, e1 = 0.0;
#pragma unroll
						for (int k = 0; k < 12; k += 2)
						{
							e0 = fma(pr[k], yq[(k / 3) * 9 + k % 3], e0);
							e1 = fma(pr[k + 1], yq[((k + 1) / 3) * 9 + (k + 1) % 3], e1);
						}
						double sum = e0 + e1;
						sum += __shfl_xor(sum, 16, 64);
						sum += __shfl_xor(sum, 32, 64);
						eacc[si] -= sum;
					}
				}
			}
			// P P^T: lane l feeds A[row l & 15][k = l >> 4] and B[k = l >> 4][col l & 15]
			bool on[T];
#pragma unroll
			for (int t = 0; t < T; t++) on[t] = ti[t] >= 0 && (((smask >> ti[t]) & (smask >> tj[t]) & 1) != 0) && !(a.dbg & 8);
#pragma unroll 2
			for (int ks = 0; ks < SM_K / 4; ks++)
			{
#pragma unroll
				for (int t = 0; t < T; t++)
				{
					if (!on[t]) continue; // (wave-uniform)
					const double av = sh.P[16 * ti[t] * SM_KS + lbase + 4 * ks], bv = sh.P[16 * tj[t] * SM_KS + lbase + 4 * ks];
					acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc[t], 0, 0, 0);
				}
			}
			__syncthreads(); // (the next pass overwrites the panel)
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
	for (int j = 0; j < ((a.dbg & 16) ? 0 : n); j++)
	{
		const double d = sh.S[j * LD + j];
		if (!(d > 0.0) || !(d < 1e300)) { pivot_bad = j; break; } // (uniform: every thread reads the same entry)
		const double piv = sqrt(d), ip = 1.0 / piv;
		__syncthreads();
		for (int i = j + tid; i < n; i += SM_THREADS) sh.S[i * LD + j] = i == j ? piv : sh.S[i * LD + j] * ip;
		__syncthreads();
		// trailing update of the lower triangle: (i, k), j < k <= i -- the work-group as a 16 x 16 grid over it (no division in the loop)
		{
			const int ty = tid >> 4, tx = tid & 15;
			for (int i = j + 1 + ty; i < n; i += 16)
			{
				const double lij = sh.S[i * LD + j];
				for (int k = j + 1 + tx; k <= i; k += 16) sh.S[i * LD + k] = fma(-lij, sh.S[k * LD + j], sh.S[i * LD + k]);
			}
		}
		__syncthreads();
	}
	if (pivot_bad >= 0)
	{
		if (tid == 0) { atomicCAS(&a.status[1], 0, 1 + g); if (a.run && !a.run->chol_err) a.run->chol_err = 1 + p0 + pivot_bad / 6; }
		return;
	}
	for (int i = tid; i < n; i += SM_THREADS) sh.dinv[i] = 1.0 / sh.S[i * LD + i];
	__syncthreads();
	// x = S^-1 E, then one refinement step against the original S (upper triangle + d0).  One wave does the triangular sweeps: row i
	// of the solution is final before row i + 1 needs it -- a chain, whatever the number of lanes
	auto solve = [&](const double* rhs, double* out) {
		// forward: L v = rhs ; backward: L^T out = v
		for (int i = tid; i < R; i += SM_THREADS) sh.v[i] = i < n ? rhs[i] : 0.0;
		__syncthreads();
		if (wave == 0)
		{
			// (d0 is free by now -- the residual reads it, so the inverse diagonal of L lives in sh.r's neighbour: sh.dinv)
			for (int j = 0; j < n; j++)
			{
				const double vj = sh.v[j] * sh.dinv[j];
				__builtin_amdgcn_wave_barrier();
				if (lane == 0) sh.v[j] = vj;
				for (int i = j + 1 + lane; i < n; i += 64) sh.v[i] = fma(-sh.S[i * LD + j], vj, sh.v[i]);
				__builtin_amdgcn_wave_barrier();
			}
			for (int j = n - 1; j >= 0; j--)
			{
				const double vj = sh.v[j] * sh.dinv[j];
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
	if (!(a.dbg & 16))
	{
	solve(sh.E, sh.x);
	residual(sh.x);
	solve(sh.r, sh.v);      // (out aliases the work vector: harmless, see the copies in solve)
	for (int i = tid; i < n; i += SM_THREADS) sh.x[i] += sh.v[i];
	__syncthreads();
	residual(sh.x);
	}
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
	for (int f = f0 + tid; f < ((a.dbg & 32) ? f0 : f1); f += SM_THREADS)
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
int small_solve_strips(int most_poses, int cap)
{
	if (most_poses > cap) return 0;
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
	a.dbg = getenv("LSFM_SMALL_DEBUG") ? atoi(getenv("LSFM_SMALL_DEBUG")) : 0;
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
