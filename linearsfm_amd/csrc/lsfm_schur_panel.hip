// K9, panel formulation on the matrix cores: S(p,q) -= sum_f W_pf V_f^-1 W_qf^T and E_p -= sum_f W_pf V_f^-1 eb_f
// (Imp.cpp:2244-2332).
//
// A tile of PM_TILE consecutive features is observed by a small set of poses (its ~12 hub poses plus the few frames
// that see it): at most PM_SMAX "slots".  With V_f^-1 = L_f L_f^T the tile's contribution is a dense symmetric rank-k
// update  P P^T,  P = [ W_sf L_f ]  (rows = 6 * slot + r, columns = 3 * feature + c, absent blocks zero), i.e. a real
// contraction over the 3 * PM_TILE feature columns: the panel is staged PM_PASS features at a time in LDS and the
// 16x16 tiles of the upper block triangle of P P^T are accumulated with v_mfma_f64_16x16x4_f64, the tiles dealt round
// robin to the four waves of the work-group.  No lane idles on an absent pose pair, the LDS traffic is two doubles per
// lane per 1024 multiply-adds, and every touched block of S leaves the work-group once.  The right-hand side part is
// the panel times y = L^T eb, one panel row per lane.
// Tiles with more than PM_SMAX poses (sub-map boundaries at the top of the tree can exceed it) or with a V^-1 that has
// no Cholesky factor are flagged and handled by the per-feature kernel k_schur_w.
#include "lsfm_device.hpp"
#include "lsfm_internal.hpp"
#include "lsfm_solve.hpp"

namespace lsfm {

#define PM_TILE 128
#define PM_PASS 16
#define PM_K (3 * PM_PASS)
#define PM_KS (PM_K + 1) /* odd row stride: the 16 rows x 2 k of a half-wave fall into distinct LDS banks */
#define PM_SMAX 32      /* slots of the common variant: 256 threads, two work-groups per CU */
#define PM_SMAX_BIG 48  /* slots of the variant for the tiles that exceed it: 1024 threads (16 waves share the 171 output tiles) */
#define PM_HASH 64
#define PM_THREADS 256
#define PM_MAXE 4096 /* W blocks of the tile whose slot is kept in LDS (one byte each); later ones probe the hash again */

typedef double v4d __attribute__((ext_vector_type(4)));

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
	int hkey[PM_HASH];
	int hslot[PM_HASH];
	int pose_of[SMAX];
	int nslots, bad;
	int fp[PM_PASS + 1];
	double Ls[PM_PASS * 6]; // l00 l10 l11 l20 l21 l22 of V^-1 = L L^T
	double ys[PM_K];        // L^T eb
	double P[6 * SMAX * PM_KS];
	unsigned char eslot[PM_MAXE]; // slot of the tile's W blocks, filled once: the passes do not touch photo[] again
};

// T = 16x16 tiles per wave (the work-group's upper-triangle tiles are dealt q = wave + NW t over its NW waves; slots
// past the last tile recompute tile (0,0) and are dropped)
template <int T, int SMAX, int THREADS>
__device__ __forceinline__ void pm_body(PmShared<SMAX>& sh, int ns, int f0, int f1, int jb, const int* __restrict__ fptr, const int* __restrict__ photo,
                                        const double* __restrict__ W, const double* __restrict__ IV, const double* __restrict__ eb,
                                        const unsigned long long* __restrict__ tab, const int* __restrict__ val, unsigned long long mask,
                                        double* __restrict__ S, double* __restrict__ E, unsigned char* __restrict__ fallback)
{
	const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6); // uniform: tile coordinates live in SGPRs
	const int rows = 6 * ns, NT = (rows + 15) >> 4, ntile = NT * (NT + 1) / 2;
	int ti[T], tj[T], offA[T], offB[T]; // wave-uniform
	const int lbase = (lane & 15) * PM_KS + (lane >> 4);
	v4d acc[T];
#pragma unroll
	for (int t = 0; t < T; t++)
	{
		const int q = wave + (THREADS / 64) * t;
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
	double eacc = 0.0;
	// W rows of the next pass are fetched into registers before the MFMA loop of the current one: the staging after the
	// barrier then works on LDS only (measured: staging with the global loads inside cost 4.5 of K9's 9.5 ms)
	// Staging is by feature: 16 lanes per feature of the pass walk the rows of its blocks (3 doubles each, consecutive
	// lanes on consecutive rows), so a lane knows its feature without searching the run pointers.
	constexpr int PF = T <= 9 ? 8 : (T <= 14 ? 2 : 1); // rows per lane held in flight (register budget of the variant)
	const int sfl = tid >> 4, sl16 = tid & 15;
	double pw[PF][3];
	auto prefetch = [&](int p0n) {
		const int nfn = min(PM_PASS, f1 - p0n);
		if (sfl < nfn)
		{
			const int qbn = fptr[p0n + sfl], nrown = (fptr[p0n + sfl + 1] - qbn) * 6;
#pragma unroll
			for (int i = 0; i < PF; i++)
			{
				const int w = sl16 + 16 * i;
				if (w < nrown)
				{
					const double* wr = W + (size_t)qbn * 18 + (size_t)w * 3; // row r of block e: (qb + e) * 18 + 3 r
					pw[i][0] = wr[0]; pw[i][1] = wr[1]; pw[i][2] = wr[2];
				}
			}
		}
	};
	if (f0 < f1) prefetch(f0);
	for (int p0 = f0; p0 < f1; p0 += PM_PASS)
	{
		const int nf = min(PM_PASS, f1 - p0);
		__syncthreads(); // the previous pass is fully consumed
		for (int q = tid; q < NT * 16 * PM_KS; q += THREADS) sh.P[q] = 0.0;
		if (tid <= nf) sh.fp[tid] = fptr[p0 + tid];
		if (tid < PM_PASS)
		{
			double l[6] = { 0, 0, 0, 0, 0, 0 }, y[3] = { 0, 0, 0 };
			if (tid < nf)
			{
				const double* a = IV + (size_t)(p0 + tid) * 9;
				const double* e = eb + (size_t)(p0 + tid) * 3;
				const double d0 = a[0];
				l[0] = sqrt(d0);
				l[1] = a[3] / l[0];
				l[3] = a[6] / l[0];
				const double d1 = a[4] - l[1] * l[1];
				l[2] = sqrt(d1);
				l[4] = (a[7] - l[3] * l[1]) / l[2];
				const double d2 = a[8] - l[3] * l[3] - l[4] * l[4];
				l[5] = sqrt(d2);
				if (!(d0 > 0.0) || !(d1 > 0.0) || !(d2 > 0.0)) sh.bad = 1;
				y[0] = l[0] * e[0] + l[1] * e[1] + l[3] * e[2];
				y[1] = l[2] * e[1] + l[4] * e[2];
				y[2] = l[5] * e[2];
			}
			for (int q = 0; q < 6; q++) sh.Ls[tid * 6 + q] = l[q];
			for (int q = 0; q < 3; q++) sh.ys[tid * 3 + q] = y[q];
		}
		__syncthreads();
		// stage P = W L
		if (sfl < nf)
		{
			const int qb = sh.fp[sfl], nrow = (sh.fp[sfl + 1] - qb) * 6;
			const double* l = &sh.Ls[sfl * 6];
			const double l0 = l[0], l1 = l[1], l2 = l[2], l3 = l[3], l4 = l[4], l5 = l[5];
			auto stage = [&](int w, double w0, double w1, double w2) {
				const int e = w / 6, r = w - 6 * e, j = qb + e;
				int sl;
				if (j - jb < PM_MAXE) sl = sh.eslot[j - jb];
				else
				{
					const int key = photo[j];
					unsigned h = ((unsigned)key * 2654435761u) & (PM_HASH - 1);
					while (sh.hkey[h] != key) h = (h + 1) & (PM_HASH - 1);
					sl = sh.hslot[h];
				}
				double* d = &sh.P[(6 * sl + r) * PM_KS + 3 * sfl];
				// two blocks of one (pose, feature) add up, as in the reference's pair loop: atomics
				__hip_atomic_fetch_add(d + 0, w0 * l0 + w1 * l1 + w2 * l3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
				__hip_atomic_fetch_add(d + 1, w1 * l2 + w2 * l4, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
				__hip_atomic_fetch_add(d + 2, w2 * l5, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
			};
#pragma unroll
			for (int i = 0; i < PF; i++)
			{
				const int w = sl16 + 16 * i;
				if (w < nrow) stage(w, pw[i][0], pw[i][1], pw[i][2]);
			}
			for (int w = sl16 + 16 * PF; w < nrow; w += 16)
			{
				const double* wr = W + (size_t)qb * 18 + (size_t)w * 3;
				stage(w, wr[0], wr[1], wr[2]);
			}
		}
		__syncthreads();
		if (p0 + PM_PASS < f1) prefetch(p0 + PM_PASS);
		// E_p -= W V^-1 eb = P y, Imp.cpp:2321-2328
		if (tid < rows)
		{
			const double* pr = &sh.P[tid * PM_KS];
			double s = 0.0;
			for (int k = 0; k < 3 * nf; k++) s = fma(pr[k], sh.ys[k], s);
			eacc -= s;
		}
		// P P^T, four feature columns per MFMA: lane l feeds A[row l & 15][k = l >> 4] and B[k = l >> 4][col l & 15]
		const int nks = (3 * nf + 3) >> 2;
		for (int ks = 0; ks < nks; ks++)
		{
#pragma unroll
			for (int t = 0; t < T; t++)
			{
				const double a = sh.P[offA[t] + lbase + 4 * ks], b = sh.P[offB[t] + lbase + 4 * ks];
				acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[t], 0, 0, 0);
			}
		}
	}
	__syncthreads();
	if (sh.bad)
	{
		if (tid == 0) fallback[blockIdx.x] = 1;
		return;
	}
	// ---- every touched block leaves the work-group once.  Slot of S for every slot pair, in the (now free) panel ----
	int* pslot = reinterpret_cast<int*>(sh.P);
	for (int q = tid; q < ns * ns; q += THREADS)
	{
		const int si = q / ns, sj = q - si * ns;
		pslot[q] = si <= sj ? pn_hash_find(tab, val, mask, sh.pose_of[si], sh.pose_of[sj]) : -1;
	}
	__syncthreads();
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
			const int slot = pslot[si * ns + sj];
			if (slot < 0) continue;
			double* d = S + (size_t)slot * 36;
			if (si == sj)
			{
				// a pose with itself is stored full; across a tile boundary only this half was computed
				atomic_add_f64(d + r * 6 + c, -v);
				if (ti[t] != tj[t]) atomic_add_f64(d + c * 6 + r, -v);
			}
			else
			{
				// stored orientation: rows = smaller pose index
				const bool up = sh.pose_of[si] <= sh.pose_of[sj];
				atomic_add_f64(d + (up ? r * 6 + c : c * 6 + r), -v);
			}
		}
	}
	if (tid < rows && eacc != 0.0) atomic_add_f64(E + (size_t)sh.pose_of[tid / 6] * 6 + tid % 6, eacc);
}

// `only` == nullptr: every tile; tiles with more than SMAX poses are flagged in `fallback`.  `only` != nullptr (the second
// pass with the larger variant): just the flagged tiles; a tile it can take is un-flagged, the rest stays for k_schur_w.
// SMAX = 8 / 16: for levels whose systems have at most that many poses (the bottom of the tree: thousands of tiny joins).
// A tile is then all latency -- hashing, eight short passes, a handful of MFMAs -- and the smaller panel lets 6 / 3
// work-groups share a CU instead of 2.
template <int SMAX, int THREADS>
__global__ void __launch_bounds__(THREADS, THREADS != 256 ? 1 : (SMAX <= 8 ? 6 : (SMAX <= 16 ? 3 : 2)))
k_schur_panel(int NF, const int* __restrict__ fptr, const int* __restrict__ photo, const double* __restrict__ W, const double* __restrict__ IV,
              const double* __restrict__ eb, const unsigned long long* __restrict__ tab, const int* __restrict__ val, unsigned long long mask,
              double* __restrict__ S, double* __restrict__ E, unsigned char* fallback, const unsigned char* only)
{
	if (only && !only[blockIdx.x]) return;
	__shared__ PmShared<SMAX> sh;
	const int tid = threadIdx.x;
	const int f0 = blockIdx.x * PM_TILE, f1 = min(f0 + PM_TILE, NF);
	const int jb = fptr[f0], je = fptr[f1];
	if (tid < PM_HASH) { sh.hkey[tid] = -1; sh.hslot[tid] = -1; }
	if (tid == 0) { sh.nslots = 0; sh.bad = 0; }
	__syncthreads();
	// ---- the tile's poses -> slots ----
	for (int j = jb + tid; j < je; j += THREADS)
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
	if (tid < PM_HASH && sh.hkey[tid] != -1)
	{
		const int id = atomicAdd(&sh.nslots, 1);
		sh.hslot[tid] = id;
		if (id < SMAX) sh.pose_of[id] = sh.hkey[tid];
	}
	__syncthreads();
	const int ns = sh.nslots;
	if (ns > SMAX || sh.bad == 2)
	{
		if (tid == 0) fallback[blockIdx.x] = 1;
		return;
	}
	if (only && tid == 0) fallback[blockIdx.x] = 0; // taken here (pm_body flags it again if a V^-1 has no Cholesky factor)
	for (int e = tid; e < je - jb && e < PM_MAXE; e += THREADS) sh.eslot[e] = (unsigned char)sh.hslot[sh.eslot[e]];
	// (visible to the passes through the barrier at the top of the first pass)
	constexpr int NW = THREADS / 64;
	const int NT = (6 * ns + 15) >> 4, tpw = (NT * (NT + 1) / 2 + NW - 1) / NW; // tiles per wave, uniform
#define PM_GO(T) pm_body<T, SMAX, THREADS>(sh, ns, f0, f1, jb, fptr, photo, W, IV, eb, tab, val, mask, S, E, fallback)
	if constexpr (SMAX <= 8)
	{
		if (tpw <= 1) PM_GO(1);
		else PM_GO(2); // 48 rows: 6 tiles over 4 waves
	}
	else if constexpr (SMAX <= 16)
	{
		if (tpw <= 1) PM_GO(1);
		else if (tpw <= 3) PM_GO(3);
		else PM_GO(6); // 96 rows: 21 tiles over 4 waves
	}
	else if constexpr (THREADS == 256)
	{
		if (tpw <= 1) PM_GO(1);
		else if (tpw <= 3) PM_GO(3);
		else if (tpw <= 6) PM_GO(6);
		else if (tpw <= 9) PM_GO(9);
		else if (tpw <= 14) PM_GO(14);
		else PM_GO(20);
	}
	else
	{
		if (tpw <= 6) PM_GO(6);
		else if (tpw <= 9) PM_GO(9);
		else PM_GO(11);
	}
#undef PM_GO
}

int schur_panel_tile() { return PM_TILE; }

void launch_schur_panel(lsfm_context* ctx, int NF, const int* fptr, const int* photo, const double* W, const double* IV, const double* eb,
                        const unsigned long long* tab, const int* val, unsigned long long mask, double* S, double* E, unsigned char* fallback,
                        int max_poses_per_system)
{
	if (!NF) return;
	const dim3 grid((NF + PM_TILE - 1) / PM_TILE);
	// no tile can be seen by more poses than its system has
	if (max_poses_per_system <= 8)
	{
		hipLaunchKernelGGL((k_schur_panel<8, PM_THREADS>), grid, dim3(PM_THREADS), 0, ctx->stream, NF, fptr, photo, W, IV, eb, tab, val, mask, S, E, fallback,
		                   (const unsigned char*)nullptr);
		return;
	}
	if (max_poses_per_system <= 16)
	{
		hipLaunchKernelGGL((k_schur_panel<16, PM_THREADS>), grid, dim3(PM_THREADS), 0, ctx->stream, NF, fptr, photo, W, IV, eb, tab, val, mask, S, E, fallback,
		                   (const unsigned char*)nullptr);
		return;
	}
	hipLaunchKernelGGL((k_schur_panel<PM_SMAX, PM_THREADS>), grid, dim3(PM_THREADS), 0, ctx->stream, NF, fptr, photo, W, IV, eb, tab, val, mask, S, E,
	                   fallback, (const unsigned char*)nullptr);
	// the tiles that exceed 32 poses (a path that revisits: the frames of two laps + the hub poses of every level)
	hipLaunchKernelGGL((k_schur_panel<PM_SMAX_BIG, 1024>), grid, dim3(1024), 0, ctx->stream, NF, fptr, photo, W, IV, eb, tab, val, mask, S, E, fallback,
	                   (const unsigned char*)fallback);
}

} // namespace lsfm
