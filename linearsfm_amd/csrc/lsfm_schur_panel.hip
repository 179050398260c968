// K9, panel formulation on the matrix cores: S(p,q) -= sum_f W_pf V_f^-1 W_qf^T and E_p -= sum_f W_pf V_f^-1 eb_f
// (Imp.cpp:2244-2332).
//
// A tile of PM_TILE consecutive features is observed by a small set of poses (the hub poses of the levels below plus the
// few frames that see it): its "slots".  With V_f^-1 = L_f L_f^T the tile's contribution is a dense symmetric rank-k
// update  P P^T,  P = [ W_sf L_f ]  (rows = 6 * slot + r, columns = 3 * feature + c, absent blocks zero), i.e. a real
// contraction over the 3 * PM_TILE feature columns: the panel is staged PM_PASS features at a time in LDS and the
// 16x16 tiles of the upper block triangle of P P^T are accumulated with v_mfma_f64_16x16x4_f64, the tiles dealt round
// robin to the waves of the work-group.  No lane idles on an absent pose pair and every touched block of S leaves the
// work-group once.  The right-hand side rides on the same products: E -= P (y - u) with y = L^T eb + P^T x_p (K9Out) is
//   E_r -= (P z_side(r))_r + sum_r' (P P^T)_rr' x_p,r',    z_side = L^T eb - u_side,
// so the two vectors z_End, z_Cur are two more ROWS of the panel behind the poses' (the products of a row with them come out of
// the MFMAs as two more columns of P P^T -- in the spare rows of the last 16-row strip unless the poses fill it), and the second
// term is the finished tile of P P^T times the poses' estimates, once per tile.  (Round 5 first had both per PASS on the vector
// units -- column sums of the panel, two more barriers, a strip-wise P z: a fifth of the kernel.)
//
// Variants by the number of slots of a tile (launch_schur_panel): 8 / 16 / 32 slots with 256 threads (4 / 3 / 2
// work-groups per CU), 48 and 64 slots with 1024 threads (one work-group per CU; 64: the output tiles in two sweeps over
// the tile's passes).  A variant flags the tiles that exceed it for the next one; what exceeds 64 poses, or holds a V^-1
// without a Cholesky factor, goes to the per-feature kernel k_schur_w.
//
// A pass works on LDS only between its barriers: the W rows of a pass (ONE contiguous range), and per feature L and y
// (k_vinv leaves them), are fetched into registers during the MFMA phase of the pass before; first blocks of a
// (pose, feature) pair are staged with plain stores, repeats -- the joins keep both blocks of a feature seen from the hub
// pose on either side -- with LDS atomics after a barrier that passes without repeats skip.  What a tile works out from
// index arrays alone (its poses, the slot of every block, the repeats) is recorded by the first run of a resident tree
// (K9Cache) and read by the later ones.  DESIGN.md section 3 "Inside a tile" has the measurements behind each of these.
#include "lsfm_device.hpp"
#include "lsfm_internal.hpp"
#include "lsfm_solve.hpp"

namespace lsfm {

#define PM_TILE LSFM_PM_TILE
#ifndef LSFM_K9_PASS16
#define LSFM_K9_PASS16 16 /* features per pass of the 8- and 16-slot variants: 16, or 8 (half the panel in LDS, twice the barriers per tile) */
#endif
#ifndef LSFM_K9_PASS32
#define LSFM_K9_PASS32 16 /* ... of the 32-, 48- and 64-slot variants */
#endif
#define PM_SMAX 32      /* slots of the common variant: 256 threads, two work-groups per CU */
#define PM_SMAX_BIG 48  /* slots of the variant for the tiles that exceed it: 1024 threads (16 waves share the 171 output tiles) */
#define PM_SMAX_MAX 64  /* the widest panel: the same 1024 threads take its 300 output tiles in two sweeps over the tile's passes */
#define PM_HASH 64
#define PM_THREADS 256
#define PM_WIDE 512     /* threads of the 48 / 64-slot variants */
#ifndef LSFM_K9_T16
#define LSFM_K9_T16 256 /* threads of the 16-slot variant (256 | 512) */
#endif
#ifndef LSFM_K9_T32
#define LSFM_K9_T32 512 /* threads of the 32-slot variant (512 | 1024) */
#endif
#ifndef LSFM_K9_OCC16
#define LSFM_K9_OCC16 3 /* work-groups per CU of the 16-slot variant: 3 = 170 registers a wave (22 spilled), 2 = 256 (none) */
#endif
#ifndef LSFM_K9_OCC16W
#define LSFM_K9_OCC16W 2 /* work-groups per CU of a 512-thread 16-slot variant: 2 = 128 registers a wave, 3 = 80 */
#endif
#define PM_MAXE (PM_TILE == 128 ? 3584 : 5120) /* W blocks of the tile whose slot is kept in LDS (one byte each: 28 / 20 per feature on average); later ones are added after the first blocks */
#define PM_BF 352   /* >= rows held in registers per pass / 6 (4 x 256 or 3 x 512) */
#define PM_DUP 0x80  /* eslot: a block whose (pose, feature) an earlier block of the tile already holds */

typedef double v4d __attribute__((ext_vector_type(4)));

// Profiling aid (make K9_TIMING=1; tools/k9_phase_times.py): lane 0 of every work-group adds up the shader clocks it
// spends in each phase of a tile and leaves the sums in g_k9_t.  Compiled out otherwise.
#ifdef LSFM_K9_TIMING
__device__ unsigned long long g_k9_t[64]; // 16 per variant (8 / 16 / 32 / 48 slots)
#define K9T_DECL unsigned long long k9acc[16] = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 }; unsigned long long k9t_prev = __builtin_readcyclecounter()
#define K9T(i) do { const unsigned long long n_ = __builtin_readcyclecounter(); k9acc[i] += n_ - k9t_prev; k9t_prev = n_; } while (0)
#define K9T_BASE (SMAX <= 8 ? 0 : (SMAX <= 16 ? 16 : (SMAX <= 32 ? 32 : 48)))
#define K9T_FLUSH(a, b) do { if (threadIdx.x == 0) for (int i_ = (a); i_ < (b); i_++) atomicAdd(&g_k9_t[K9T_BASE + i_], k9acc[i_]); } while (0)
#define K9T_COUNT(ns, T) do { if (threadIdx.x == 0) { atomicAdd(&g_k9_t[K9T_BASE + 8], 1ull); atomicAdd(&g_k9_t[K9T_BASE + 9], (unsigned long long)(ns)); atomicAdd(&g_k9_t[K9T_BASE + 10], (unsigned long long)(T)); } } while (0)
#else
#define K9T_DECL do { } while (0)
#define K9T(i) do { } while (0)
#define K9T_FLUSH(a, b) do { } while (0)
#define K9T_COUNT(ns, T) do { } while (0)
#endif

__device__ __forceinline__ unsigned long long pn_mix64(unsigned long long x)
{
	x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33;
	return x;
}
__device__ __forceinline__ int pn_hash_find(const unsigned long long* __restrict__ tab, const int* __restrict__ val, unsigned long long mask,
                                            int p, int q)
{
	const unsigned long long key = p <= q ? (((unsigned long long)(unsigned)p << 32) | (unsigned)q) : (((unsigned long long)(unsigned)q << 32) | (unsigned)p);
	unsigned long long h = pn_mix64(key) & mask;
	for (int probe = 0; probe < 4096; probe++)
	{
		const unsigned long long cur = tab[h];
		if (cur == key) return val[h];
		if (cur == ~0ull) return -1;
		h = (h + 1) & mask;
	}
	return -1;
}

template <int SMAX>
struct PmShared {
	// (the widest panel fills a CU's LDS: its list of block slots is shorter -- later blocks read the level's copy -- and nothing here is wider than it must be)
	// features per pass, panel columns, row stride (odd: the 16 rows x 2 k of a half-wave fall into distinct LDS banks)
	static constexpr int PASS = SMAX <= 16 ? LSFM_K9_PASS16 : LSFM_K9_PASS32, K = 3 * PASS, KS = K + 1;
	static constexpr int MAXE = SMAX > PM_SMAX_BIG ? 2560 : PM_MAXE;
	// rows of the panel: six per slot and two more for the right-hand side (below); whole 16-row strips.  The widest variant has no
	// room for a strip more: it takes tiles of up to 63 poses
	static constexpr int PROWS = SMAX >= PM_SMAX_MAX ? 6 * SMAX : 6 * SMAX + 16;
	static constexpr int CAP = (PROWS - 2) / 6 < SMAX ? (PROWS - 2) / 6 : SMAX; // poses of a tile this variant takes
	int pose_of[SMAX];
	int nslots, bad;
	unsigned char bf[PM_BF]; // block of the pass -> its feature (the first PM_BF blocks; later ones search the run pointers)
	int fpt[PM_TILE + 1]; // run pointers of the tile's features: the prefetch of a pass must not wait for them first
	short sexp[6 * SMAX];   // binary exponent of the scale of every panel row (K9Out::sexp of the slot's pose)
	// right-hand side (K9Out): estimate of every panel row's pose scalar (zero: not fused, and past the tile's rows), the map side
	// of every slot's pose, per pass L^-1 x_f of the features' two sources
	double xs[PROWS];
	unsigned char side[SMAX];
	double uu[PASS * 6];
	double ly[PASS * 9]; // per feature of the pass: l00 l10 l11 l20 l21 l22 of V^-1 = L L^T, then y = L^T eb; zero past the last one
	alignas(16) double P[PROWS * KS];
	unsigned char eslot[MAXE]; // slot of the tile's W blocks (| PM_DUP), filled once: the passes do not touch photo[] again
	// block of S of every slot pair si <= sj (at sj (sj + 1) / 2 + si), looked up while the first pass's rows are on their way instead of
	// behind the last pass (the widest variants have no room: they look them up at the tile's end, into the free panel)
	static constexpr bool PLANNED = SMAX <= PM_SMAX;
	int pslot[PLANNED ? SMAX * (SMAX + 1) / 2 : 1];
};

// The end of a tile, behind the barrier that ends its passes: every touched block of S leaves the work-group once (fixed point), the
// right-hand side rows are summed per wave and then over the waves in their order.  `scratch`: the (now free) panel; wslot: the row
// set of this wave's output tiles (0 .. NWV - 1).
template <int T, int NWV, int THREADS, class SH>
__device__ __forceinline__ void k9_tile_end(SH& sh, double* scratch, int ns, const v4d (&acc)[T], const int (&ti)[T], const int (&tj)[T], int wslot, bool fused, int ey,
                                            const K9Out& o, const unsigned long long* __restrict__ tab, const int* __restrict__ val, unsigned long long mask,
                                            unsigned char* __restrict__ fallback, int tile)
{
	const int tid = threadIdx.x, lane = tid & 63, rows = 6 * ns;
	if (sh.bad)
	{
		if (tid == 0) fallback[tile] = 1;
		return;
	}
	// ---- every touched block leaves the work-group once.  Slot of S for every slot pair: planned at the tile's start, or now, in the
	// (now free) panel ----
	int* pslot = reinterpret_cast<int*>(scratch);
	if constexpr (!SH::PLANNED)
	{
		for (int q = tid; q < ns * ns; q += THREADS)
		{
			const int si = q / ns, sj = q - si * ns;
			pslot[q] = si <= sj ? pn_hash_find(tab, val, mask, sh.pose_of[si], sh.pose_of[sj]) : -1;
		}
	}
	// ... and behind them the tile's right-hand side rows, one set per wave: a wave adds its output tiles' shares in the order it holds
	// them, the waves' sets are added in their order below -- the same bits every run, no atomics, one conversion to the fixed point per row
	double* wsum = scratch + ((ns * ns + 1) >> 1);
	for (int q = tid; q < NWV * rows; q += THREADS) wsum[q] = 0.0;
	__syncthreads();
	bool bad = false;
	// E_R -= sum_C (P P^T)_RC m_C: m_C = the estimate of pose scalar C; for the two columns behind the poses' (P z_End, P z_Cur)
	// 1 on the rows of that side's poses.  Off the diagonal a tile stands for its mirror image too: E_C -= sum_R (P P^T)_RC x_R.
	double* mine = wsum + (wslot < 0 ? 0 : wslot) * rows; // (a wave without output tiles -- wslot < 0 -- adds nothing)
#pragma unroll
	for (int t = 0; t < T; t++)
	{
		if (ti[t] < 0) continue; // (uniform)
		const int C = 16 * tj[t] + (lane & 15), Rb = 16 * ti[t] + (lane >> 4);
		if (!fused && 16 * tj[t] + 15 < rows) continue; // (uniform: without estimates only the tiles of the last strip count)
		const double xc = C < rows ? sh.xs[C] : 0.0;
		const int zc = C - rows;
		const bool mirror = ti[t] != tj[t] && fused;
		double rs[4], cs = 0.0;
#pragma unroll
		for (int e = 0; e < 4; e++)
		{
			const int R = Rb + 4 * e;
			const double v = acc[t][e];
			double term = v * xc;
			if (zc >= 0) term = (zc < 2 && R < rows && sh.side[R / 6] == zc) ? v : 0.0;
			rs[e] = term;
			if (mirror && C < rows) cs = fma(v, sh.xs[R], cs); // (xs is zero past the poses' rows)
		}
		// the four row sums over the 16 lanes of a row group in five exchanges: halves of the group trade the pair of sums they do not
		// keep, quarters the one, then two plain steps -- lane bits (3, 2) of the group say whose sum a lane ends up with
		{
			const bool b3 = (lane & 8) != 0, b2 = (lane & 4) != 0;
			const double s0 = b3 ? rs[0] : rs[2], s1 = b3 ? rs[1] : rs[3];
			const double a0 = (b3 ? rs[2] : rs[0]) + __shfl_xor(s0, 8, 64), a1 = (b3 ? rs[3] : rs[1]) + __shfl_xor(s1, 8, 64);
			double r = (b2 ? a1 : a0) + __shfl_xor(b2 ? a0 : a1, 4, 64);
			r += __shfl_xor(r, 2, 64);
			r += __shfl_xor(r, 1, 64);
			const int R = Rb + 4 * ((b3 ? 2 : 0) + (b2 ? 1 : 0));
			if ((lane & 3) == 0 && R < rows) mine[R] -= r;
		}
		if (mirror)
		{
			cs += __shfl_xor(cs, 16, 64);
			cs += __shfl_xor(cs, 32, 64);
			if (lane < 16 && C < rows) mine[C] -= cs; // (rows of another strip than the ones above: ti != tj)
		}
	}
#pragma unroll
	for (int t = 0; t < T; t++)
	{
		if (ti[t] < 0) continue;
		const int C = 16 * tj[t] + (lane & 15);
		if (C >= rows) continue;
		const int sj = C / 6, c = C - 6 * sj;
#pragma unroll
		for (int e = 0; e < 4; e++)
		{
			// C/D of v_mfma_f64_16x16x4_f64: col = lane & 15, row = (lane >> 4) + 4 * reg
			const int R = 16 * ti[t] + (lane >> 4) + 4 * e;
			const double v = acc[t][e];
			if (R >= rows || !(v != 0.0)) continue; // exact zero: this pose pair shares no feature of the tile
			const int si = R / 6, r = R - 6 * si;
			if (si > sj) continue; // diagonal tile: the mirrored element covers it
			int slot;
			if constexpr (SH::PLANNED) slot = sh.pslot[sj * (sj + 1) / 2 + si];
			else slot = pslot[si * ns + sj];
			if (slot < 0) continue;
			// order-independent: the entry in fixed point, units of 2^(sexp_R + sexp_C - 60) -- every partial sum of an entry of
			// W V^-1 W^T is below sqrt(U_RR U_CC) < 2^(sexp_R + sexp_C - 2) (the joint information matrix is positive semi-definite)
			long long* d = o.S + (size_t)slot * 36;
			const long long q = to_fixed(v, 60 - sh.sexp[R] - sh.sexp[C], bad);
			if (si == sj)
			{
				// a pose with itself is stored full; across a tile boundary only this half was computed
				atomic_add_i64(d + r * 6 + c, q);
				if (ti[t] != tj[t]) atomic_add_i64(d + c * 6 + r, q);
			}
			else
			{
				// stored orientation: rows = smaller pose index
				const bool up = sh.pose_of[si] <= sh.pose_of[sj];
				atomic_add_i64(d + (up ? r * 6 + c : c * 6 + r), q);
			}
		}
	}
	__syncthreads();
	{
		for (int row = tid; row < rows; row += THREADS)
		{
			double e = 0.0;
#pragma unroll
			for (int w = 0; w < NWV; w++) e += wsum[w * rows + row];
			if (e != 0.0)
			{
				// |any partial sum of (W V^-1 eb)_row| <= sqrt(U_row,row) |L^T eb| < 2^(sexp_row + ey - 1): two limbs below that
				long long hi, lo;
				to_fixed2(e, 62 - sh.sexp[row] - ey, hi, lo, bad);
				const size_t at = (size_t)sh.pose_of[row / 6] * 6 + row % 6;
				atomic_add_i64(o.Ehi + at, hi);
				if (lo) atomic_add_i64(o.Elo + at, lo);
			}
		}
	}
	if (bad) atomic_add_i64(o.poison, 1);
}

// T = 16x16 tiles per wave (the work-group's upper-triangle tiles are dealt q = wave + NW t over its NW waves; slots
// past the last tile recompute tile (0,0) and are dropped)
template <int T, int SMAX, int THREADS, class Fill>
__device__ __forceinline__ void pm_body(PmShared<SMAX>& sh, int ns, int f0, int f1, int jb, const int* __restrict__ fptr, const int* __restrict__ photo,
                                        const double* __restrict__ W, const double* __restrict__ LY,
                                        const unsigned long long* __restrict__ tab, const int* __restrict__ val, unsigned long long mask,
                                        const K9Out& o, unsigned char* __restrict__ fallback,
                                        const unsigned char* __restrict__ ces, int tile, Fill&& fill, int q0 = 0)
{
	const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6); // uniform: tile coordinates live in SGPRs
	constexpr int PM_PASS = PmShared<SMAX>::PASS, PM_K = PmShared<SMAX>::K, PM_KS = PmShared<SMAX>::KS;
	K9T_DECL;
	// (rows: the poses' rows; behind them rows and rows + 1 hold z of the End / the Cur side)
	const int rows = 6 * ns, NT = (rows + 2 + 15) >> 4, ntile = NT * (NT + 1) / 2;
	int ti[T], tj[T], offA[T], offB[T]; // wave-uniform
	const int lbase = (lane & 15) * PM_KS + (lane >> 4);
	v4d acc[T];
#pragma unroll
	for (int t = 0; t < T; t++)
	{
		const int q = q0 + wave + (THREADS / 64) * t; // (a sweep of the widest variant starts at output tile q0)
		int i = 0, j = 0;
		if (q < ntile)
		{
			// q = j (j + 1) / 2 + i, i <= j
			j = (int)((sqrtf(8.0f * q + 1.0f) - 1.0f) * 0.5f);
			while (j * (j + 1) / 2 > q) j--;
			while ((j + 1) * (j + 2) / 2 <= q) j++;
			i = q - j * (j + 1) / 2;
		}
		ti[t] = q < ntile ? i : -1;
		tj[t] = j;
		offA[t] = 16 * i * PM_KS;
		offB[t] = 16 * j * PM_KS;
		acc[t] = (v4d){ 0.0, 0.0, 0.0, 0.0 };
	}
	// Everything a pass reads from memory is fetched into registers before the MFMA loop of the pass before it, and none of
	// it behind a dependent load: the run pointers of the tile sit in LDS, the W rows of a pass are ONE contiguous range
	// (row w of the pass = 3 doubles at W[18 qb0 + 3 w]: consecutive lanes, consecutive 24 bytes, whatever the lengths of
	// the 16 runs are), (L, y) of its features 144 contiguous doubles (k_vinv leaves them per feature: no square roots or
	// divisions here).  Between the barriers a pass then works on LDS only: the feature of a row is found in the pass's run
	// pointers, its L read from the staged copy.  (Measured before: staging with the global loads inside 4.5 of K9's 9.5 ms;
	// the per-pass Cholesky of V^-1 behind a dependent load a quarter of a tile; 16 lanes per feature -- a feature seen by
	// more than 10 poses went back to memory inside the staging -- a third of it.)
#ifndef LSFM_K9_PF6
#define LSFM_K9_PF6 3
#endif
	// rows per lane in flight: 4 x 256 rows = 10.6 poses per feature on average (LSFM_K9_PF6 = 3: the 16-slot variant fits its 170
	// registers without spilling, with 8 poses per feature prefetched)
#ifndef LSFM_K9_PFW
#define LSFM_K9_PFW 1
#endif
	// (rows past the prefetched ones are loaded INSIDE the staging, a memory round trip in the open: the instances that hold fewer output
	// tiles spend the registers on a deeper prefetch -- T = 4, the 11-13-pose tiles, five rows a lane = 213 blocks a pass)
	// (sixteen waves: 2 x 1024 rows = 21 poses per feature; eight waves on a 16-slot panel: 3 x 512 rows = every block it can hold)
	constexpr int PF = THREADS == 1024 ? 2
	                 : THREADS != 256 ? (SMAX <= 16 ? (LSFM_K9_OCC16W >= 3 ? 2 : 3) : (LSFM_K9_PFW ? (T <= 8 ? 4 : 3) : (T <= 6 ? 3 : 2)))
	                                  : (T <= 3 ? 4 : (T <= 4 ? (LSFM_K9_PFW ? 5 : LSFM_K9_PF6) : (T <= 6 ? LSFM_K9_PF6 : (T <= 14 ? 3 : 2))));
	double pw[PF][3];
	double lyv = 0.0, uuv = 0.0;
	const bool fused = o.xpose != nullptr; // (uniform)
	int qb0 = 0, R = 0;
	auto prefetch = [&](int p0n) {
		const int nfn = min(PM_PASS, f1 - p0n);
		qb0 = sh.fpt[p0n - f0];
		R = (sh.fpt[p0n - f0 + nfn] - qb0) * 6;
		lyv = 0.0; uuv = 0.0;
		if (tid < nfn * 9) lyv = LY[(size_t)p0n * 9 + tid];
		if (fused && tid < nfn * 6) uuv = o.uu[(size_t)p0n * 6 + tid];
		const double* wb = W + (size_t)qb0 * 18;
#pragma unroll
		for (int i = 0; i < PF; i++)
		{
			const int w = tid + THREADS * i;
			if (w < R) { pw[i][0] = wb[3 * (size_t)w]; pw[i][1] = wb[3 * (size_t)w + 1]; pw[i][2] = wb[3 * (size_t)w + 2]; }
		}
	};
	if (f0 < f1) prefetch(f0);
	// what the tile keeps in LDS beside the run pointers (slots of its blocks, scales and estimates of its poses, the blocks of S it
	// adds to): loaded behind the first pass's rows, not in front of them -- one memory round trip of a tile's three less in the open;
	// visible to the passes through the barrier at the top of the first one
	if (q0 == 0) fill();
	K9T(0); // (with the header loads of the kernel: "poses -> slots")
	const int ey = *o.ey;
	for (int p0 = f0; p0 < f1; p0 += PM_PASS)
	{
		const int nf = min(PM_PASS, f1 - p0), pb = p0 - f0;
		K9T(5);
		__syncthreads(); // the previous pass is fully consumed
		K9T(1);
		{
			// (two doubles a store: the panel starts 16-byte aligned and NT * 16 * PM_KS is even)
			typedef double d2 __attribute__((ext_vector_type(2)));
			d2* P2 = reinterpret_cast<d2*>(sh.P);
			for (int q = tid; q < NT * 8 * PM_KS; q += THREADS) P2[q] = (d2){ 0.0, 0.0 };
		}
		if (tid < PM_PASS * 9)
		{
			sh.ly[tid] = lyv; // zero for the features past the end of the tile
			if (tid < PM_PASS * 6) sh.uu[tid] = uuv;
			if (tid < nf * 9 && tid % 9 == 0 && !(lyv == lyv)) sh.bad = 1; // k_vinv: V^-1 of this feature has no Cholesky factor
		}
		{
			// block -> feature of the pass: 16 lanes per feature walk its run
			const int ff = tid >> 4;
			if (ff < nf)
			{
				const int a = sh.fpt[pb + ff] - qb0, b = min(sh.fpt[pb + ff + 1] - qb0, PM_BF);
				for (int e = a + (tid & 15); e < b; e += 16) sh.bf[e] = (unsigned char)ff;
			}
		}
		__syncthreads();
		K9T(2);
#ifdef LSFM_K9_TIMING
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // what is left of the prefetch's latency, apart from the staging itself
		K9T(7);
#endif
		// stage P = W L: first blocks of their (pose, feature) pair with plain stores; the repeats (and the blocks past the
		// LDS slot list) wait for the barrier and are added
		bool later = false;
		const double* wb = W + (size_t)qb0 * 18;
		if (tid < PM_K)
		{
			// the right-hand side's two rows: z = L^T eb - u of either source side (u = 0: not fused; zero past the last feature)
			const int fl = tid / 3, c = tid - 3 * fl;
			const double yv = sh.ly[fl * 9 + 6 + c];
			sh.P[rows * PM_KS + tid] = yv - sh.uu[fl * 6 + c];
			sh.P[(rows + 1) * PM_KS + tid] = yv - sh.uu[fl * 6 + 3 + c];
		}
		auto stage = [&](int w, double w0, double w1, double w2, bool second) {
			const int e = w / 6, r = w - 6 * e, j = qb0 + e;
			int sl, dup = 1;
			if (j - jb < PmShared<SMAX>::MAXE) { sl = sh.eslot[j - jb]; dup = sl & PM_DUP; sl &= PM_DUP - 1; }
			else
			{
				if (!second) { later = true; return; }
				sl = ces[j] & (PM_DUP - 1); // the level's copy (k_schur_slots, or the plan)
			}
			if (!second && dup) { later = true; return; }
			if (second && !dup) return;
			int lo = 0;
			if (e < PM_BF) lo = sh.bf[e];
			else
			{
				int hi = nf - 1; // feature of block j: the last run of the pass that starts at or before it
				while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (sh.fpt[pb + mid] <= j) lo = mid; else hi = mid - 1; }
			}
			const double* l = &sh.ly[lo * 9];
			double* d = &sh.P[(6 * sl + r) * PM_KS + 3 * lo];
			const double v0 = w0 * l[0] + w1 * l[1] + w2 * l[3], v1 = w1 * l[2] + w2 * l[4], v2 = w2 * l[5];
			if (!second)
			{
				d[0] = v0; d[1] = v1; d[2] = v2;
			}
			else
			{
				__hip_atomic_fetch_add(d + 0, v0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
				__hip_atomic_fetch_add(d + 1, v1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
				__hip_atomic_fetch_add(d + 2, v2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
			}
		};
#pragma unroll
		for (int i = 0; i < PF; i++)
		{
			const int w = tid + THREADS * i;
			if (w < R) stage(w, pw[i][0], pw[i][1], pw[i][2], false);
		}
		for (int w = tid + THREADS * PF; w < R; w += THREADS) stage(w, wb[3 * (size_t)w], wb[3 * (size_t)w + 1], wb[3 * (size_t)w + 2], false);
		K9T(3);
		if (__syncthreads_or(later))
		{
#pragma unroll
			for (int i = 0; i < PF; i++)
			{
				const int w = tid + THREADS * i;
				if (w < R) stage(w, pw[i][0], pw[i][1], pw[i][2], true);
			}
			for (int w = tid + THREADS * PF; w < R; w += THREADS) stage(w, wb[3 * (size_t)w], wb[3 * (size_t)w + 1], wb[3 * (size_t)w + 2], true);
			__syncthreads();
		}
		K9T(4);
		if (p0 + PM_PASS < f1) prefetch(p0 + PM_PASS);
		else { R = 0; lyv = 0.0; uuv = 0.0; }
		K9T(12);
		// P P^T, four feature columns per MFMA: lane l feeds A[row l & 15][k = l >> 4] and B[k = l >> 4][col l & 15].
		// Always the full PM_K columns (zero past the last feature); the T tiles of a step are independent chains
		constexpr int UNR = T <= 1 ? PM_K / 4 : (T <= 3 ? 4 : (T <= (THREADS == 1024 ? 4 : 6) ? 2 : 1));
#pragma unroll UNR
		for (int ks = 0; ks < PM_K / 4; ks++)
		{
#pragma unroll
			for (int t = 0; t < T; t++)
			{
				const double a = sh.P[offA[t] + lbase + 4 * ks], b = sh.P[offB[t] + lbase + 4 * ks];
				acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[t], 0, 0, 0);
			}
		}
	}
	K9T(5);
	__syncthreads();
	K9T(1);
	k9_tile_end<T, THREADS / 64, THREADS>(sh, sh.P, ns, acc, ti, tj, wave, fused, ey, o, tab, val, mask, fallback, tile);
	K9T(6);
	K9T_FLUSH(0, 8);
	K9T_FLUSH(11, 13);
	K9T_COUNT(ns, T);
}

// the variant of pm_body for the tile's number of 16x16 output tiles per wave
template <int SMAX, int THREADS, class Fill>
__device__ __forceinline__ void k9_go(PmShared<SMAX>& sh, int ns, int f0, int f1, int jb, const int* __restrict__ fptr, const int* __restrict__ photo,
                                      const double* __restrict__ W, const double* __restrict__ LY, const unsigned long long* __restrict__ tab,
                                      const int* __restrict__ val, unsigned long long mask, const K9Out& o,
                                      unsigned char* __restrict__ fallback, const unsigned char* __restrict__ ces, int tile, Fill&& fill)
{
	constexpr int NW = THREADS / 64;
	const int NT = (6 * ns + 2 + 15) >> 4, tpw = (NT * (NT + 1) / 2 + NW - 1) / NW; // tiles per wave, uniform (strips: pm_body)
#define PM_GO(T) pm_body<T, SMAX, THREADS>(sh, ns, f0, f1, jb, fptr, photo, W, LY, tab, val, mask, o, fallback, ces, tile, fill)
	if constexpr (SMAX <= 8)
	{
		if (tpw <= 1) PM_GO(1);
		else if (tpw <= 2) PM_GO(2); // 3 strips: 6 tiles over 4 waves
		else PM_GO(3); // (8 poses: a fourth strip for the right-hand side's rows)
	}
	else if constexpr (SMAX <= 16 && THREADS == 256)
	{
		// (an instance per count: a wave of the T = 6 instance with four tiles to its name multiplied two more for nothing -- a third of the
		// matrix products of the commonest tiles, 11-13 poses, until round 5)
		if (tpw <= 1) PM_GO(1);
		else if (tpw <= 2) PM_GO(2);
		else if (tpw <= 3) PM_GO(3);
		else if (tpw <= 4) PM_GO(4); // 5 strips: 15 tiles over 4 waves
		else if (tpw <= 6) PM_GO(6); // 6 strips: 21 tiles over 4 waves
		else PM_GO(7); // (16 poses: a seventh strip)
	}
	else if constexpr (SMAX <= 16)
	{
		// eight waves: at most 28 tiles = 4 per wave
		if (tpw <= 1) PM_GO(1);
		else if (tpw <= 2) PM_GO(2);
		else if (tpw <= 3) PM_GO(3);
		else PM_GO(4);
	}
	else if constexpr (THREADS == 1024)
	{
		// sixteen waves, four to a SIMD, 128 registers each: a 32-slot panel's 91 output tiles = at most 6 per wave
		static_assert(SMAX <= PM_SMAX, "sixteen waves: the 32-slot variant only");
		if (tpw <= 1) PM_GO(1);
		else if (tpw <= 2) PM_GO(2);
		else if (tpw <= 3) PM_GO(3);
		else if (tpw <= 4) PM_GO(4);
		else if (tpw <= 5) PM_GO(5);
		else PM_GO(6);
	}
	else
	{
		// 32 / 48 / 64 slots: 512 threads -- eight waves, two to a SIMD, so that a wave may hold 256 registers: at most 13 output
		// tiles (104 accumulator registers) per wave and sweep, nothing spills (the 1024-thread work-groups of round 2 capped a
		// wave at 128 registers: 400-640 spilled registers in the 48 / 64-slot variants).  Wider panels take their output tiles
		// in sweeps over the tile's passes: 32 slots 78 tiles = 10 per wave, 48 slots 171 tiles = 2 sweeps of 11, 64 slots
		// 300 tiles = 3 sweeps of 13.
		constexpr int TS = SMAX <= PM_SMAX ? 10 : (SMAX <= PM_SMAX_BIG ? 12 : 13);
		if (tpw <= 1) PM_GO(1);
		else if (tpw <= 3) PM_GO(3);
		else if (tpw <= 6) PM_GO(6);
		else if (tpw <= 8) PM_GO(8); // (24-29 poses: 55 / 66 tiles over 8 waves)
		else if (tpw <= TS) PM_GO(TS);
		else
		{
			for (int q0 = 0; q0 < NW * tpw; q0 += NW * TS)
			{
				pm_body<TS, SMAX, THREADS>(sh, ns, f0, f1, jb, fptr, photo, W, LY, tab, val, mask, o, fallback, ces, tile, fill, q0);
				if (sh.bad) return; // (uniform: set before the barrier that ends the passes)
			}
		}
	}
#undef PM_GO
}

// The tile's poses -> slots, once per tile and level, ahead of the panel variants (k_schur_panel): which poses see the tile's
// features (a 64-entry hash table in LDS), the slot of every W block, which blocks repeat a (pose, feature) pair of the tile.
// The joins keep both blocks when a feature was seen from the hub pose on either side (the reference concatenates, Imp.cpp:1277;
// its pair loop adds them up): the first block of a pair is staged with plain stores, the repeats are added after it -- an LDS
// atomic add of a double costs ~4 clocks per LANE on this chip, and as the only way into the panel it was 40 % of the kernel.
// Leaves ns[tile] (poses; -poses / -1000: more than the widest panel takes / than the table holds -- the tile goes to k_schur_w),
// pose[tile][slot], eslot[block].  It is structure: a resident tree's later runs take it from the plan of the level.  As a
// launch of its own (round 4) every variant knows its tiles before it starts, so the variants of a level run side by side.
struct SlotShared {
	int hkey[PM_HASH];
	int hslot[PM_HASH];
	int pose_of[PM_SMAX_MAX];
	int nslots, bad;
	int fpt[PM_TILE + 1];
	unsigned char eslot[PM_MAXE];
};
__global__ void __launch_bounds__(PM_THREADS) k_schur_slots(int NF, const int* __restrict__ fptr, const int* __restrict__ photo, unsigned char* fallback, K9Cache kc)
{
	__shared__ SlotShared sh;
	const int tid = threadIdx.x;
	const int f0 = blockIdx.x * PM_TILE, f1 = min(f0 + PM_TILE, NF);
	for (int i = tid; i <= f1 - f0; i += PM_THREADS) sh.fpt[i] = fptr[f0 + i];
	if (tid < PM_HASH) { sh.hkey[tid] = -1; sh.hslot[tid] = -1; }
	if (tid == 0) { sh.nslots = 0; sh.bad = 0; }
	__syncthreads();
	const int jb = sh.fpt[0], je = sh.fpt[f1 - f0];
	for (int j = jb + tid; j < je; j += PM_THREADS)
	{
		const int key = photo[j];
		unsigned h = ((unsigned)key * 2654435761u) & (PM_HASH - 1);
		int probe = 0;
		for (; probe < PM_HASH; probe++)
		{
			const int cur = sh.hkey[h];
			if (cur == key) break;
			if (cur == -1)
			{
				const int old = atomicCAS(&sh.hkey[h], -1, key);
				if (old == -1 || old == key) break;
			}
			h = (h + 1) & (PM_HASH - 1);
		}
		if (probe == PM_HASH) sh.bad = 2; // table full: more than PM_HASH distinct poses
		if (j - jb < PM_MAXE) sh.eslot[j - jb] = (unsigned char)h; // table position now, slot number once slots are dealt
	}
	__syncthreads();
	// slots in the order of the table, not of arrival: the same tile gets the same slots in every run
	if (tid < PM_HASH)
	{
		const bool used = sh.hkey[tid] != -1;
		const unsigned long long m = __ballot(used);
		if (used)
		{
			const int id = __popcll(m & ((1ull << tid) - 1ull));
			sh.hslot[tid] = id;
			sh.pose_of[id] = sh.hkey[tid];
		}
		if (tid == 0) sh.nslots = __popcll(m);
	}
	__syncthreads();
	const int ns = sh.nslots;
	if (sh.bad == 2)
	{
		if (tid == 0) { fallback[blockIdx.x] = 1; kc.ns[blockIdx.x] = -1000; }
		return;
	}
	for (int e = tid; e < je - jb && e < PM_MAXE; e += PM_THREADS) sh.eslot[e] = (unsigned char)sh.hslot[sh.eslot[e]];
	__syncthreads();
	if (tid < f1 - f0)
	{
		// one lane per feature walks its run: a slot it has met before marks a repeat
		unsigned long long seen = 0ull;
		const int a = sh.fpt[tid] - jb, b = min(sh.fpt[tid + 1] - jb, PM_MAXE);
		for (int e = a; e < b; e++)
		{
			const int sl = sh.eslot[e];
			const unsigned long long bit = 1ull << sl;
			if (seen & bit) sh.eslot[e] = (unsigned char)(sl | PM_DUP);
			seen |= bit;
		}
	}
	__syncthreads();
	// (more poses than the widest panel takes -- its last strip has no room for the right-hand side's rows: k_schur_w)
	if (tid == 0) kc.ns[blockIdx.x] = ns > PmShared<PM_SMAX_MAX>::CAP ? -ns : ns;
	if (tid < ns) kc.pose[(size_t)blockIdx.x * PM_SMAX_MAX + tid] = sh.pose_of[tid];
	for (int e = tid; e < je - jb; e += PM_THREADS)
	{
		unsigned char v;
		if (e < PM_MAXE) v = sh.eslot[e];
		else
		{
			// (the blocks past the LDS list too)
			const int key = photo[jb + e];
			unsigned h = ((unsigned)key * 2654435761u) & (PM_HASH - 1);
			while (sh.hkey[h] != key) h = (h + 1) & (PM_HASH - 1);
			v = (unsigned char)(sh.hslot[h] | PM_DUP); // added after the first blocks, like a repeat
		}
		kc.eslot[jb + e] = v;
	}
}

// The tiles of the wide variants, listed in tile order (one work-group: a level has a few thousand tiles): a variant with an 80-157 KB
// panel takes its tiles off its list with a few hundred work-groups instead of starting one per tile of the level that leaves at
// once -- every one of those had to wait for a whole CU's LDS (the 64-slot variant's empty launch took 150 us of a level).
__global__ void __launch_bounds__(256) k_schur_lists(int ntiles, K9Cache kc)
{
	__shared__ int cnt[3][256];
	const int tid = threadIdx.x, per = (ntiles + 255) / 256, t0 = tid * per, t1 = min(t0 + per, ntiles);
	int c[3] = { 0, 0, 0 };
	for (int t = t0; t < t1; t++)
	{
		const int ns = kc.ns[t];
		if (ns > 16) c[ns <= PM_SMAX ? 0 : (ns <= PM_SMAX_BIG ? 1 : 2)]++;
	}
	for (int v = 0; v < 3; v++) cnt[v][tid] = c[v];
	__syncthreads();
	if (tid < 3)
	{
		int run = 0;
		for (int i = 0; i < 256; i++) { const int x = cnt[tid][i]; cnt[tid][i] = run; run += x; }
		kc.wcnt[tid] = run;
	}
	if (tid < 4) kc.wcnt[4 + tid] = 0; // the variants' cursors into their lists (a run that reads the lists from a plan: launch_schur_panel)
	__syncthreads();
	int pos[3] = { cnt[0][tid], cnt[1][tid], cnt[2][tid] };
	for (int t = t0; t < t1; t++)
	{
		const int ns = kc.ns[t];
		if (ns > 16)
		{
			const int v = ns <= PM_SMAX ? 0 : (ns <= PM_SMAX_BIG ? 1 : 2);
			kc.wlist[(size_t)v * ntiles + pos[v]++] = t;
		}
	}
}

template <int SMAX, int THREADS>
constexpr int k9_waves_per_simd() { return THREADS == 256 ? (SMAX <= 8 ? 4 : LSFM_K9_OCC16) : (SMAX <= 16 ? LSFM_K9_OCC16W * (THREADS / 256) : THREADS / 256 / 2); }

// what a tile keeps in LDS beside its run pointers and poses (called by pm_body behind the first pass's prefetch)
template <int SMAX, int THREADS>
__device__ __forceinline__ void k9_fill(PmShared<SMAX>& sh, int cns, int jb, int je, const K9Out& o, const K9Cache& kc,
                                        const unsigned long long* __restrict__ tab, const int* __restrict__ val, unsigned long long mask)
{
	const int tid = threadIdx.x;
	for (int e = tid; e < je - jb && e < PmShared<SMAX>::MAXE; e += THREADS) sh.eslot[e] = kc.eslot[jb + e];
	for (int i = tid; i < 6 * cns; i += THREADS) sh.sexp[i] = (short)o.sexp[6 * (size_t)sh.pose_of[i / 6] + i % 6];
	for (int i = tid; i < PmShared<SMAX>::PROWS; i += THREADS) sh.xs[i] = (o.xpose && i < 6 * cns) ? o.xpose[6 * (size_t)sh.pose_of[i / 6] + i % 6] : 0.0;
	for (int i = tid; i < cns; i += THREADS) sh.side[i] = o.pside ? (unsigned char)(o.pside[sh.pose_of[i]] & 1) : (unsigned char)0;
	if constexpr (PmShared<SMAX>::PLANNED)
	{
		// (from the last thread down: the first ones hold the loops above)
		for (int q = THREADS - 1 - tid; q < cns * (cns + 1) / 2; q += THREADS)
		{
			int sj = (int)((sqrtf(8.0f * q + 1.0f) - 1.0f) * 0.5f);
			while (sj * (sj + 1) / 2 > q) sj--;
			while ((sj + 1) * (sj + 2) / 2 <= q) sj++;
			const int si = q - sj * (sj + 1) / 2;
			sh.pslot[q] = pn_hash_find(tab, val, mask, sh.pose_of[si], sh.pose_of[sj]);
		}
	}
}

// One variant per panel width; a tile belongs to the narrowest variant that holds its poses.  The 8- and 16-slot variants are
// launched with one work-group per tile of the level (wlist == nullptr; `alone`: no wider variant is launched beside it -- a tile that
// exceeds the panel is flagged for k_schur_w, like the tiles the slots kernel gave up on); the 32-, 48- and 64-slot variants
// (LISTED) take their tiles off the list of their variant (k_schur_lists) with one work-group per CU; `lo`: tiles of at most that
// many poses belong to a narrower variant launched beside a one-work-group-per-tile launch.
// SMAX = 8 / 16: for levels whose systems have at most that many poses (the bottom of the tree: thousands of tiny joins).
// A tile is then all latency -- eight short passes, a handful of MFMAs -- and the smaller panel lets 6 / 3
// work-groups share a CU instead of 2.
template <int SMAX, int THREADS, bool LISTED>
__device__ __forceinline__ void
k9_kernel(int NF, const int* __restrict__ fptr, const int* __restrict__ photo, const double* __restrict__ W, const double* __restrict__ LY,
          const unsigned long long* __restrict__ tab, const int* __restrict__ val, unsigned long long mask,
          const K9Out& o, unsigned char* fallback, int alone, const int* __restrict__ wlist, const int* __restrict__ wcnt, int* cursor, int lo, const K9Cache& kc,
          PmShared<SMAX>& sh)
{
	const int tid = threadIdx.x;
	// (LISTED is a template parameter, not a test of wlist: the loop around the tile cost the 16-slot variant, which sits at its
	// register limit, ten spilled registers)
	if constexpr (!LISTED)
	{
		const int tile = blockIdx.x;
		const int cns = kc.ns[tile];
		if (cns >= 0 && cns <= lo) return; // (lo: widest panel of the narrower variants launched beside this one)
		if (cns < 0 || cns > PmShared<SMAX>::CAP)
		{
			// more poses than the hash table of the slots kernel holds, or -- where this is the only variant launched (levels of
			// small systems; a tile may straddle systems and see more poses than any one of them has) -- than the panel: k_schur_w
			if ((cns < 0 || alone) && tid == 0) fallback[tile] = 1;
			return;
		}
		K9T_DECL;
		const int f0 = tile * PM_TILE, f1 = min(f0 + PM_TILE, NF);
		for (int i = tid; i <= f1 - f0; i += THREADS) sh.fpt[i] = fptr[f0 + i];
		if (tid < cns) sh.pose_of[tid] = kc.pose[(size_t)tile * PM_SMAX_MAX + tid];
		if (tid == 0) { sh.nslots = cns; sh.bad = 0; }
		__syncthreads();
		const int jb = sh.fpt[0], je = sh.fpt[f1 - f0];
		K9T(0);
		K9T_FLUSH(0, 1);
		k9_go<SMAX, THREADS>(sh, cns, f0, f1, jb, fptr, photo, W, LY, tab, val, mask, o, fallback, kc.eslot, tile,
		                     [&]() { k9_fill<SMAX, THREADS>(sh, cns, jb, je, o, kc, tab, val, mask); });
	}
	else
	{
		__shared__ int s_it;
		const int nlist = *wcnt;
		// A short list -- the handful of 33-48-pose tiles of a top level, say -- would leave one work-group per tile walking its eight passes
		// alone while the launch behind it on the stream waits (150-200 us for ONE tile, 0.4 ms per tree): its tiles are cut into
		// parts of whole passes, each taken by a work-group of its own with the tile's slots.  The sums are integers: whatever is
		// added in how many pieces, S and E are the same; how a list is cut hangs on its length and the launch alone -- the same every run.
		const int room = (int)gridDim.x;
		const int parts = nlist * 8 <= room ? 8 : (nlist * 4 <= room ? 4 : (nlist * 2 <= room ? 2 : 1));
		const int plen = PM_TILE / parts;
		for (int round = 0;; round++)
		{
			// the next tile of the list, whoever comes first (tiles differ in work by an order of magnitude)
			if (round) __syncthreads(); // (the tile before is done with the panel and with s_it)
			if (tid == 0) s_it = atomicAdd(cursor, 1);
			__syncthreads();
			const int it = s_it;
			if (it >= nlist * parts) break;
			const int tile = wlist[it / parts];
			const int cns = kc.ns[tile];
			K9T_DECL;
			const int f0 = tile * PM_TILE + (it % parts) * plen, f1 = min(f0 + plen, NF);
			if (f0 >= f1) continue; // (the ragged end of the level's last tile; uniform)
			for (int i = tid; i <= f1 - f0; i += THREADS) sh.fpt[i] = fptr[f0 + i];
			if (tid < cns) sh.pose_of[tid] = kc.pose[(size_t)tile * PM_SMAX_MAX + tid];
			if (tid == 0) { sh.nslots = cns; sh.bad = 0; }
			__syncthreads();
			const int jb = sh.fpt[0], je = sh.fpt[f1 - f0];
			K9T(0);
			K9T_FLUSH(0, 1);
			k9_go<SMAX, THREADS>(sh, cns, f0, f1, jb, fptr, photo, W, LY, tab, val, mask, o, fallback, kc.eslot, tile,
			                     [&]() { k9_fill<SMAX, THREADS>(sh, cns, jb, je, o, kc, tab, val, mask); });
		}
	}
}

template <int SMAX, int THREADS, bool LISTED>
__global__ void __launch_bounds__(THREADS, (k9_waves_per_simd<SMAX, THREADS>()))
k_schur_panel(int NF, const int* __restrict__ fptr, const int* __restrict__ photo, const double* __restrict__ W, const double* __restrict__ LY,
              const unsigned long long* __restrict__ tab, const int* __restrict__ val, unsigned long long mask,
              K9Out o, unsigned char* fallback, int alone, const int* __restrict__ wlist, const int* __restrict__ wcnt, int* cursor, int lo, K9Cache kc)
{
	__shared__ PmShared<SMAX> sh;
	k9_kernel<SMAX, THREADS, LISTED>(NF, fptr, photo, W, LY, tab, val, mask, o, fallback, alone, wlist, wcnt, cursor, lo, kc, sh);
}

int schur_panel_tile() { return PM_TILE; }

#ifdef LSFM_K9_TIMING
extern "C" void lsfm_debug_k9(unsigned long long* out, int reset)
{
	(void)hipDeviceSynchronize();
	if (out) (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_k9_t), sizeof(unsigned long long) * 64);
	if (reset) { unsigned long long z[64] = { 0 }; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_k9_t), z, sizeof(z)); }
}
#endif

void launch_schur_slots(lsfm_context* ctx, int NF, const int* fptr, const int* photo, unsigned char* fallback, K9Cache kc)
{
	if (!NF) return;
	const int ntiles = (NF + PM_TILE - 1) / PM_TILE;
	hipLaunchKernelGGL(k_schur_slots, dim3(ntiles), dim3(PM_THREADS), 0, ctx->stream, NF, fptr, photo, fallback, kc);
	hipLaunchKernelGGL(k_schur_lists, dim3(1), dim3(256), 0, ctx->stream, ntiles, kc);
}

// kc: the tiles' slots (launch_schur_slots of this run, or the plan of the level).
void launch_schur_panel(lsfm_context* ctx, int NF, const int* fptr, const int* photo, const double* W, const double* LY,
                        const unsigned long long* tab, const int* val, unsigned long long mask, K9Out out, unsigned char* fallback,
                        int max_poses_per_system, K9Cache kc, bool fresh_lists)
{
	if (!NF) return;
	const int ntiles = (NF + PM_TILE - 1) / PM_TILE;
	const dim3 grid(ntiles);
	hipStream_t s = ctx->stream;
	const int* none = nullptr;
	static const bool hist = getenv("LSFM_K9_HIST") != nullptr; // diagnostic: poses per tile of every level, on stderr
	if (hist)
	{
		std::vector<int> h(ntiles);
		LSFM_CHECK_HIP(hipStreamSynchronize(s));
		LSFM_CHECK_HIP(hipMemcpy(h.data(), kc.ns, sizeof(int) * ntiles, hipMemcpyDeviceToHost));
		int cnt[66] = { 0 };
		for (int v : h) cnt[v < 0 ? 65 : std::min(v, 64)]++;
		fprintf(stderr, "K9 tiles %d (largest system %d poses):", ntiles, max_poses_per_system);
		for (int i = 0; i < 66; i++) if (cnt[i]) fprintf(stderr, " %d:%d", i == 65 ? -1 : i, cnt[i]);
		fprintf(stderr, "\n");
	}
	// no tile can be seen by more poses than its system has
	if (max_poses_per_system <= 8)
	{
		hipLaunchKernelGGL((k_schur_panel<8, PM_THREADS, false>), grid, dim3(PM_THREADS), 0, s, NF, fptr, photo, W, LY, tab, val, mask, out, fallback, 1, none, none, (int*)nullptr, 0, kc);
		return;
	}
	if (max_poses_per_system <= 16)
	{
		hipLaunchKernelGGL((k_schur_panel<16, LSFM_K9_T16, false>), grid, dim3(LSFM_K9_T16), 0, s, NF, fptr, photo, W, LY, tab, val, mask, out, fallback, 1, none, none, (int*)nullptr, 0, kc);
		return;
	}
	// By tile, not by level: most tiles of the upper levels are seen by a dozen poses (12.1 on average on the NC3500-like
	// set) and fit the 16-slot variant, which is three work-groups to a CU instead of two and a third less work per pass;
	// wider tiles go to the 32-, 48- and 64-slot variants, the rest to k_schur_w.  The slots kernel has told every variant its
	// tiles, so the variants of a level run side by side (until round 4: one after the other, each behind the tail of the one
	// before, every wide variant as one work-group per tile of the level -- nearly all of which left at once, but not before each
	// had waited for 80-157 KB of a CU's LDS: the 64-slot launch of an NC3500 level took 150 us to do nothing).  The wide variants
	// take their tiles off their lists with one work-group per CU, on the main stream; the 16-slot variant, one work-group per tile,
	// follows on the side stream.  K9 per tree: NC3500-like 7.7 -> 7.2 ms, synth-16k 83 -> 64, RS468-like 2.8 -> 3.1-3.3.
	// Measured and dropped: streams of their own for the variants (five streams on the context slowed EVERY launch of the run down,
	// 40 -> 55 ms per tree); the next-level stream (it has the pattern of the next level queued at this point); a head start for
	// part of the 32-slot list (7.9 -> 8.4); everything in one stream (8.5-9.1); half / twice as many work-groups on the lists.
	// LSFM_K9_SERIAL=1: one stream.
	static const bool serial = getenv("LSFM_K9_SERIAL") != nullptr;
	static const int ncu = []() { int d = 0, n = 0; (void)hipGetDevice(&d); (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, d); return n > 0 ? n : 256; }();
	hipStream_t s1 = serial ? s : ctx->stream2;
	if (!fresh_lists) fill_async(s, kc.wcnt + 4, 0, 4 * sizeof(int)); // the variants' cursors into their lists (k_schur_lists zeroes them when it has just run)
	if (!serial)
	{
		LSFM_CHECK_HIP(hipEventRecord(ctx->ev_k9[0], s));
		LSFM_CHECK_HIP(hipStreamWaitEvent(s1, ctx->ev_k9[0], 0));
	}
	const dim3 wgrid(std::min(ntiles, ncu));
	// (widest first: a work-group of the 64-slot variant needs a CU's whole LDS to start -- behind the others it would wait for the
	// 16-slot variant to drain even when its list is empty)
	hipLaunchKernelGGL((k_schur_panel<PM_SMAX_MAX, PM_WIDE, true>), wgrid, dim3(PM_WIDE), 0, s, NF, fptr, photo, W, LY, tab, val, mask, out, fallback, 0, kc.wlist + 2 * (size_t)ntiles, kc.wcnt + 2, kc.wcnt + 6, 0, kc);
	hipLaunchKernelGGL((k_schur_panel<PM_SMAX_BIG, PM_WIDE, true>), wgrid, dim3(PM_WIDE), 0, s, NF, fptr, photo, W, LY, tab, val, mask, out, fallback, 0, kc.wlist + ntiles, kc.wcnt + 1, kc.wcnt + 5, 0, kc);
	hipLaunchKernelGGL((k_schur_panel<PM_SMAX, LSFM_K9_T32, true>), wgrid, dim3(LSFM_K9_T32), 0, s, NF, fptr, photo, W, LY, tab, val, mask, out, fallback, 0, kc.wlist, kc.wcnt, kc.wcnt + 4, 0, kc);
	hipLaunchKernelGGL((k_schur_panel<16, LSFM_K9_T16, false>), grid, dim3(LSFM_K9_T16), 0, s1, NF, fptr, photo, W, LY, tab, val, mask, out, fallback, 0, none, none, (int*)nullptr, 0, kc);
	if (!serial)
	{
		LSFM_CHECK_HIP(hipEventRecord(ctx->ev_k9[1], s1));
		LSFM_CHECK_HIP(hipStreamWaitEvent(s, ctx->ev_k9[1], 0));
	}
}

} // namespace lsfm
