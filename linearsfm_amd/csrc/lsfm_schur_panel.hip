// K9, panel formulation: S(p,q) -= sum_f W_pf V_f^-1 W_qf^T and E_p -= sum_f W_pf V_f^-1 eb_f  (Imp.cpp:2244-2332).
//
// A tile of PN_TILE consecutive features is observed by a small set of poses (its ~12 hub poses plus the few frames
// that see it): at most PN_SMAX "slots".  Instead of one lane per FEATURE adding each of its k_f(k_f+1)/2 products
// somewhere (atomics: to HBM 0.08 TB/s, to LDS 64-way same-address conflicts on the hub pairs), one lane owns one
// POSE PAIR of the tile and walks the tile's features, reading the W blocks from an LDS panel A[f][slot] (staged
// PN_PASS features at a time, absent blocks skipped through a presence mask).  Every pair block is accumulated in
// registers and leaves the work-group once, as 36 contiguous adds.  Tiles with more than PN_SMAX poses (sub-map
// boundaries at the top of the tree can exceed it) are flagged and handled by the per-feature kernel k_schur_w.
#include "lsfm_device.hpp"
#include "lsfm_internal.hpp"
#include "lsfm_solve.hpp"

namespace lsfm {

#define PN_TILE 128
#define PN_PASS 16
#define PN_SMAX 31
#define PN_HASH 64
#define PN_THREADS 256

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

// acc (6x6, rows = slot i, cols = slot j) += (A_i V^-1) A_j^T, row by row to keep few registers live
__device__ __forceinline__ void pn_accumulate(double* acc, const double* __restrict__ Ai, const double* __restrict__ Aj, const double* __restrict__ iv)
{
	double aj[18], v[9];
#pragma unroll
	for (int q = 0; q < 18; q++) aj[q] = Aj[q];
#pragma unroll
	for (int q = 0; q < 9; q++) v[q] = iv[q];
#pragma unroll
	for (int r = 0; r < 6; r++)
	{
		const double a0 = Ai[3 * r], a1 = Ai[3 * r + 1], a2 = Ai[3 * r + 2];
		const double t0 = a0 * v[0] + a1 * v[3] + a2 * v[6];
		const double t1 = a0 * v[1] + a1 * v[4] + a2 * v[7];
		const double t2 = a0 * v[2] + a1 * v[5] + a2 * v[8];
#pragma unroll
		for (int c = 0; c < 6; c++) acc[r * 6 + c] += t0 * aj[3 * c] + t1 * aj[3 * c + 1] + t2 * aj[3 * c + 2];
	}
}

__global__ void __launch_bounds__(PN_THREADS)
k_schur_panel(int NF, const int* __restrict__ fptr, const int* __restrict__ photo, const double* __restrict__ W, const double* __restrict__ IV,
              const double* __restrict__ eb, const unsigned long long* __restrict__ tab, const int* __restrict__ val, unsigned long long mask,
              double* __restrict__ S, double* __restrict__ E, unsigned char* __restrict__ fallback)
{
	__shared__ int hkey[PN_HASH];
	__shared__ int hslot[PN_HASH];
	__shared__ int pose_of[PN_SMAX + 1];
	__shared__ int nslots;
	__shared__ unsigned pres[PN_PASS];
	__shared__ double ivs[PN_PASS * 9];
	__shared__ double ebs[PN_PASS * 3];
	__shared__ double A[PN_PASS * PN_SMAX * 18];
	const int tid = threadIdx.x;
	const int f0 = blockIdx.x * PN_TILE, f1 = min(f0 + PN_TILE, NF);
	const int jb = fptr[f0], je = fptr[f1];
	if (tid < PN_HASH) { hkey[tid] = -1; hslot[tid] = -1; }
	if (tid == 0) nslots = 0;
	__syncthreads();
	// ---- the tile's poses -> slots ----
	for (int j = jb + tid; j < je; j += PN_THREADS)
	{
		const int key = photo[j];
		unsigned h = ((unsigned)key * 2654435761u) & (PN_HASH - 1);
		for (int probe = 0; probe < PN_HASH; probe++)
		{
			const int cur = hkey[h];
			if (cur == key) break;
			if (cur == -1)
			{
				const int old = atomicCAS(&hkey[h], -1, key);
				if (old == -1 || old == key) break;
			}
			h = (h + 1) & (PN_HASH - 1);
		}
	}
	__syncthreads();
	if (tid < PN_HASH && hkey[tid] != -1)
	{
		const int id = atomicAdd(&nslots, 1);
		hslot[tid] = id;
		if (id < PN_SMAX) pose_of[id] = hkey[tid];
	}
	__syncthreads();
	const int ns = nslots;
	if (ns > PN_SMAX)
	{
		// more than PN_HASH distinct poses also ends here: the table is then full, nslots = PN_HASH > PN_SMAX
		if (tid == 0) fallback[blockIdx.x] = 1;
		return;
	}
	// ---- pair tasks: lane t owns pairs t and t + PN_THREADS of the ns(ns+1)/2 slot pairs ----
	const int ntask = ns * (ns + 1) / 2;
	int ti[2], tj[2];
	bool used[2] = { false, false };
	double acc0[36], acc1[36], eacc[6];
	zero<36>(acc0); zero<36>(acc1); zero<6>(eacc);
#pragma unroll
	for (int u = 0; u < 2; u++)
	{
		const int pr = tid + u * PN_THREADS;
		int a = 0, b = 0;
		if (pr < ntask)
		{
			a = (int)((sqrt(8.0 * pr + 1.0) - 1.0) * 0.5);
			while (a * (a + 1) / 2 > pr) a--;
			while ((a + 1) * (a + 2) / 2 <= pr) a++;
			b = pr - a * (a + 1) / 2;
		}
		ti[u] = (pr < ntask) ? b : -1; // b <= a: slot pair (b, a)
		tj[u] = a;
	}
	bool eused = false;
	for (int p0 = f0; p0 < f1; p0 += PN_PASS)
	{
		const int p1 = min(p0 + PN_PASS, f1);
		__syncthreads(); // the previous pass is fully consumed
		for (int q = tid; q < PN_PASS * PN_SMAX * 18; q += PN_THREADS) A[q] = 0.0;
		if (tid < PN_PASS) pres[tid] = 0u;
		if (tid < (p1 - p0) * 9) ivs[tid] = IV[(size_t)p0 * 9 + tid];
		if (tid < (p1 - p0) * 3) ebs[tid] = eb[(size_t)p0 * 3 + tid];
		__syncthreads();
		// stage the W blocks of the pass; two blocks of one (pose, feature) add up, as in the reference's pair loop
		const int qb = fptr[p0], qe = fptr[p1];
		for (int j = qb + tid; j < qe; j += PN_THREADS)
		{
			// feature of entry j: the run that contains it (<= PN_PASS runs)
			int fl = 0;
			while (fl + 1 < p1 - p0 && fptr[p0 + fl + 1] <= j) fl++;
			const int key = photo[j];
			unsigned h = ((unsigned)key * 2654435761u) & (PN_HASH - 1);
			while (hkey[h] != key) h = (h + 1) & (PN_HASH - 1);
			const int sl = hslot[h];
			double* d = &A[(fl * PN_SMAX + sl) * 18];
			const double* w = W + (size_t)j * 18;
			const unsigned old = atomicOr(&pres[fl], 1u << sl);
			if (old & (1u << sl))
			{
				for (int q = 0; q < 18; q++) __hip_atomic_fetch_add(d + q, w[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
			}
			else
			{
				// first block of this (feature, pose): the cell was zeroed; a duplicate arriving concurrently adds atomically
				for (int q = 0; q < 18; q++) __hip_atomic_fetch_add(d + q, w[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
			}
		}
		__syncthreads();
		// ---- consume ----
		for (int fl = 0; fl < p1 - p0; fl++)
		{
			const unsigned m = pres[fl];
			const double* iv = &ivs[fl * 9];
			if (ti[0] >= 0 && ((m >> ti[0]) & 1u) && ((m >> tj[0]) & 1u))
			{
				pn_accumulate(acc0, &A[(fl * PN_SMAX + ti[0]) * 18], &A[(fl * PN_SMAX + tj[0]) * 18], iv);
				used[0] = true;
			}
			if (ti[1] >= 0 && ((m >> ti[1]) & 1u) && ((m >> tj[1]) & 1u))
			{
				pn_accumulate(acc1, &A[(fl * PN_SMAX + ti[1]) * 18], &A[(fl * PN_SMAX + tj[1]) * 18], iv);
				used[1] = true;
			}
			if (tid < ns && ((m >> tid) & 1u))
			{
				// E_p -= W V^-1 eb, Imp.cpp:2321-2328
				const double* a = &A[(fl * PN_SMAX + tid) * 18];
				const double e0 = ebs[fl * 3], e1 = ebs[fl * 3 + 1], e2 = ebs[fl * 3 + 2];
				const double y0 = iv[0] * e0 + iv[1] * e1 + iv[2] * e2, y1 = iv[3] * e0 + iv[4] * e1 + iv[5] * e2,
				             y2 = iv[6] * e0 + iv[7] * e1 + iv[8] * e2;
#pragma unroll
				for (int r = 0; r < 6; r++) eacc[r] -= a[3 * r] * y0 + a[3 * r + 1] * y1 + a[3 * r + 2] * y2;
				eused = true;
			}
		}
	}
	// ---- every touched block leaves the work-group once ----
#pragma unroll
	for (int u = 0; u < 2; u++)
	{
		if (!used[u]) continue;
		const double* acc = u ? acc1 : acc0;
		const int pa = pose_of[ti[u]], pb = pose_of[tj[u]];
		const int slot = pn_hash_find(tab, val, mask, pa, pb);
		double* d = S + (size_t)slot * 36;
		// acc = sum (A_i V^-1) A_j^T is the contribution to S(pa, pb); stored orientation: rows = smaller pose index
		if (pa <= pb) { for (int q = 0; q < 36; q++) atomic_add_f64(d + q, -acc[q]); }
		else { for (int r = 0; r < 6; r++) for (int c = 0; c < 6; c++) atomic_add_f64(d + c * 6 + r, -acc[r * 6 + c]); }
	}
	if (eused)
	{
		const int p = pose_of[tid];
		for (int r = 0; r < 6; r++) atomic_add_f64(E + (size_t)p * 6 + r, eacc[r]);
	}
}

int schur_panel_tile() { return PN_TILE; }

void launch_schur_panel(lsfm_context* ctx, int NF, const int* fptr, const int* photo, const double* W, const double* IV, const double* eb,
                        const unsigned long long* tab, const int* val, unsigned long long mask, double* S, double* E, unsigned char* fallback)
{
	if (NF)
		hipLaunchKernelGGL(k_schur_panel, dim3((NF + PN_TILE - 1) / PN_TILE), dim3(PN_THREADS), 0, ctx->stream, NF, fptr, photo, W, IV, eb, tab, val,
		                   mask, S, E, fallback);
}

} // namespace lsfm
