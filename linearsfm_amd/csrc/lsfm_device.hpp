// Device-side helpers: small fixed-size block algebra in registers, rotation helpers, wave-level reductions.
// gfx950 only: wavefront = 64 lanes.
#pragma once
#include <hip/hip_runtime.h>

#define LSFM_WAVE 64
#define LSFM_PI_REF 3.1415926 /* the reference's truncated literal, Imp.h:57 */

namespace lsfm {

// ---------------------------------------------------------------------------------------------------------
// block algebra (all row-major, sizes are compile-time so everything stays in registers)
// ---------------------------------------------------------------------------------------------------------
template <int N>
__device__ __forceinline__ void ld(double* dst, const double* __restrict__ src)
{
#pragma unroll
	for (int i = 0; i < N; i++) dst[i] = src[i];
}
template <int N>
__device__ __forceinline__ void ld(double* dst, const float* __restrict__ src) // an fp32 copy widened on load
{
#pragma unroll
	for (int i = 0; i < N; i++) dst[i] = (double)src[i];
}
template <int N>
__device__ __forceinline__ void st(double* __restrict__ dst, const double* src)
{
#pragma unroll
	for (int i = 0; i < N; i++) dst[i] = src[i];
}
template <int N>
__device__ __forceinline__ void zero(double* a)
{
#pragma unroll
	for (int i = 0; i < N; i++) a[i] = 0.0;
}

// C[RA x CB] (+)= A[RA x K] * B[K x CB]
template <int RA, int K, int CB, bool ACC>
__device__ __forceinline__ void mm(const double* A, const double* B, double* C)
{
#pragma unroll
	for (int i = 0; i < RA; i++)
#pragma unroll
		for (int j = 0; j < CB; j++)
		{
			double s = ACC ? C[i * CB + j] : 0.0;
#pragma unroll
			for (int k = 0; k < K; k++) s = fma(A[i * K + k], B[k * CB + j], s);
			C[i * CB + j] = s;
		}
}
// C[CA x CB] (+)= A^T * B with A[K x CA], B[K x CB]
template <int K, int CA, int CB, bool ACC>
__device__ __forceinline__ void mtm(const double* A, const double* B, double* C)
{
#pragma unroll
	for (int i = 0; i < CA; i++)
#pragma unroll
		for (int j = 0; j < CB; j++)
		{
			double s = ACC ? C[i * CB + j] : 0.0;
#pragma unroll
			for (int k = 0; k < K; k++) s = fma(A[k * CA + i], B[k * CB + j], s);
			C[i * CB + j] = s;
		}
}
// C[RA x RB] (+)= A * B^T with A[RA x K], B[RB x K]
template <int RA, int K, int RB, bool ACC>
__device__ __forceinline__ void mmt(const double* A, const double* B, double* C)
{
#pragma unroll
	for (int i = 0; i < RA; i++)
#pragma unroll
		for (int j = 0; j < RB; j++)
		{
			double s = ACC ? C[i * RB + j] : 0.0;
#pragma unroll
			for (int k = 0; k < K; k++) s = fma(A[i * K + k], B[j * K + k], s);
			C[i * RB + j] = s;
		}
}
template <int R, int C>
__device__ __forceinline__ void transpose(const double* A, double* AT)
{
#pragma unroll
	for (int i = 0; i < R; i++)
#pragma unroll
		for (int j = 0; j < C; j++) AT[j * R + i] = A[i * C + j];
}
__device__ __forceinline__ void mv3(const double* R, const double* v, double* o)
{
	o[0] = R[0] * v[0] + R[1] * v[1] + R[2] * v[2];
	o[1] = R[3] * v[0] + R[4] * v[1] + R[5] * v[2];
	o[2] = R[6] * v[0] + R[7] * v[1] + R[8] * v[2];
}

// 3x3 symmetric inverse written back symmetrised from the upper triangle (pba_inverseV, Imp.cpp:3022-3042)
__device__ __forceinline__ void inv3_sym(const double* a, double* o)
{
	double c00 = a[4] * a[8] - a[5] * a[7], c01 = a[5] * a[6] - a[3] * a[8], c02 = a[3] * a[7] - a[4] * a[6];
	double det = a[0] * c00 + a[1] * c01 + a[2] * c02, id = 1.0 / det;
	double i01 = (a[2] * a[7] - a[1] * a[8]) * id, i02 = (a[1] * a[5] - a[2] * a[4]) * id, i12 = (a[2] * a[3] - a[0] * a[5]) * id;
	o[0] = c00 * id; o[4] = (a[0] * a[8] - a[2] * a[6]) * id; o[8] = (a[0] * a[4] - a[1] * a[3]) * id;
	o[1] = o[3] = i01; o[2] = o[6] = i02; o[5] = o[7] = i12;
}

// ---------------------------------------------------------------------------------------------------------
// rotations (Imp.cpp:132-347)
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void mul33(const double* A, const double* B, double* C)
{
#pragma unroll
	for (int i = 0; i < 3; i++)
#pragma unroll
		for (int j = 0; j < 3; j++) C[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
}
// R3 = R1 * R2^T (lmj_TimesRRT, Imp.cpp:336-347)
__device__ __forceinline__ void times_rrt(double* R3, const double* R1, const double* R2)
{
#pragma unroll
	for (int i = 0; i < 3; i++)
#pragma unroll
		for (int j = 0; j < 3; j++) R3[3 * i + j] = R1[3 * i] * R2[3 * j] + R1[3 * i + 1] * R2[3 * j + 1] + R1[3 * i + 2] * R2[3 * j + 2];
}
// lmj_RMatrixYPR22, Imp.cpp:132-143
__device__ __forceinline__ void rmat_ypr(double* R, double Alpha, double Beta, double Gamma)
{
	double ca = cos(Alpha), sa = sin(Alpha), cb = cos(Beta), sb = sin(Beta), cg = cos(Gamma), sg = sin(Gamma);
	R[0] = cb * ca; R[1] = cb * sa; R[2] = -sb;
	R[3] = sg * sb * ca - cg * sa; R[4] = sg * sb * sa + cg * ca; R[5] = sg * cb;
	R[6] = cg * sb * ca + sg * sa; R[7] = cg * sb * sa - sg * ca; R[8] = cg * cb;
}
// lmj_InvRotMatrixYPR22, Imp.cpp:162-177
__device__ __forceinline__ void inv_rmat_ypr(const double* R, double& alpha, double& beta, double& gamma)
{
	beta = atan2(-R[2], sqrt(R[0] * R[0] + R[1] * R[1]));
	double cb = cos(beta);
	if (cb == 0) { alpha = 0; beta = LSFM_PI_REF / 2; gamma = atan2(R[1], R[4]); }
	else { alpha = atan2(R[1] / cb, R[0] / cb); gamma = atan2(R[5] / cb, R[8] / cb); }
}
// lmj_InvRotMatrixYPR22T, Imp.cpp:145-160
__device__ __forceinline__ void inv_rmat_ypr_T(const double* R, double& alpha, double& beta, double& gamma)
{
	beta = atan2(-R[6], sqrt(R[0] * R[0] + R[3] * R[3]));
	double cb = cos(beta);
	if (cb == 0) { alpha = 0; beta = LSFM_PI_REF / 2; gamma = atan2(R[3], R[4]); }
	else { alpha = atan2(R[3] / cb, R[0] / cb); gamma = atan2(R[7] / cb, R[8] / cb); }
}
// lmj_Rderivation, Imp.cpp:179-280
__device__ __forceinline__ void r_derivation(double Alpha, double Beta, double Gamma, double* R, double* dRA, double* dRB, double* dRG)
{
	double ca = cos(Alpha), sa = sin(Alpha), cb = cos(Beta), sb = sin(Beta), cg = cos(Gamma), sg = sin(Gamma);
	double RG[9] = { 1, 0, 0, 0, cg, sg, 0, -sg, cg };
	double RB[9] = { cb, 0, -sb, 0, 1, 0, sb, 0, cb };
	double RA[9] = { ca, sa, 0, -sa, ca, 0, 0, 0, 1 };
	double DG[9] = { 0, 0, 0, 0, -sg, cg, 0, -cg, -sg };
	double DB[9] = { -sb, 0, -cb, 0, 0, 0, cb, 0, -sb };
	double DA[9] = { -sa, ca, 0, -ca, -sa, 0, 0, 0, 0 };
	double tmp[9];
	R[0] = cb * ca; R[1] = cb * sa; R[2] = -sb;
	R[3] = sg * sb * ca - cg * sa; R[4] = sg * sb * sa + cg * ca; R[5] = sg * cb;
	R[6] = cg * sb * ca + sg * sa; R[7] = cg * sb * sa - sg * ca; R[8] = cg * cb;
	mul33(DG, RB, tmp); mul33(tmp, RA, dRG);
	mul33(RG, DB, tmp); mul33(tmp, RA, dRB);
	mul33(RG, RB, tmp); mul33(tmp, DA, dRA);
}
// Rates of the three angles that inv_rmat_ypr (TRANSPOSED: inv_rmat_ypr_T) reads off a rotation matrix R, when R moves at dR:
//   yaw = atan(R[a] / R[0]),  pitch = atan(-R[b] / hypot(R[0], R[a])),  roll = atan(R[c] / R[8])      (a, b, c) = (1, 2, 5) / (3, 6, 7)
// by the quotient and chain rules.  What lmj_dRi / lmj_dRiTT compute (Imp.cpp:282-307 / 309-334); the operations are kept in the
// reference's order -- the rates enter the Jacobians of every pose, and the fixtures of the real reference are compared at 1e-9.
__device__ __forceinline__ double ratio_rate(double num, double dnum, double den, double dden) { return (dnum * den - num * dden) / (den * den); }
__device__ __forceinline__ double atan_slope(double q) { return 1.0 / (1 + q * q); }
template <bool TRANSPOSED>
__device__ __forceinline__ void ypr_rates(double* rate, const double* dR, const double* R)
{
	constexpr int a = TRANSPOSED ? 3 : 1, b = TRANSPOSED ? 6 : 2, c = TRANSPOSED ? 7 : 5;
	const double yaw_q = R[a] / R[0], roll_q = R[c] / R[8];
	const double h2 = R[0] * R[0] + R[a] * R[a], h = sqrt(h2);
	const double pitch_q = -R[b] / h;
	const double dh = (1.0 / (2 * sqrt(h2))) * (2 * R[0] * dR[0] + 2 * R[a] * dR[a]);
	rate[0] = atan_slope(yaw_q) * ratio_rate(R[a], dR[a], R[0], dR[0]);
	rate[1] = atan_slope(pitch_q) * ((-dR[b] * h + R[b] * dh) / h2);
	rate[2] = atan_slope(roll_q) * ratio_rate(R[c], dR[c], R[8], dR[8]);
}

// ---------------------------------------------------------------------------------------------------------
// wave-level helpers
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
	for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, LSFM_WAVE);
	return v;
}

__device__ __forceinline__ void atomic_add_f64(double* p, double v)
{
	// hardware global_atomic_add_f64 (built with -munsafe-fp-atomics), agent scope, no return value used
	__hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---- order-independent sums -------------------------------------------------------------------------------------------------
// A sum that many work-groups add to with atomics lands in another order every run; in floating point that is another result every
// run.  Where a bound of every partial sum is known beforehand, the addends are instead rounded ONCE to a fixed-point grid far below
// fp64's resolution of that bound and added as 64-bit integers: integer addition is associative, the sum is the same bits whatever
// the order (the factorisation has done so since round 4; K9's sums since round 5).
__device__ __forceinline__ void atomic_add_i64(long long* p, long long v)
{
	__hip_atomic_fetch_add(reinterpret_cast<unsigned long long*>(p), (unsigned long long)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void lds_add_i64(long long* p, long long v)
{
	__hip_atomic_fetch_add(reinterpret_cast<unsigned long long*>(p), (unsigned long long)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// v 2^sh rounded to the nearest integer; bad: not finite, or beyond the 2^61 the bound leaves room for
__device__ __forceinline__ long long to_fixed(double v, int sh, bool& bad)
{
	const double q = ldexp(v, sh);
	if (!(fabs(q) < 2305843009213693952.0)) { bad = true; return 0; }
	return __double2ll_rn(q);
}
// two limbs, for sums whose bound is loose: hi in units of 2^-sh, lo in units of 2^-(sh + 40) -- 100 bits below the bound
__device__ __forceinline__ void to_fixed2(double v, int sh, long long& hi, long long& lo, bool& bad)
{
	const double q = ldexp(v, sh);
	if (!(fabs(q) < 2305843009213693952.0)) { bad = true; hi = 0; lo = 0; return; }
	const double h = rint(q);
	hi = (long long)h;
	lo = __double2ll_rn(ldexp(q - h, 40)); // (q - h is exact: |q| < 2^52 leaves a fraction fp64 holds, beyond that there is none)
}

// Adds vals[0..N) into dst[0..N) for every lane with valid==true.  The lanes of a wave that target the SAME dst (the common case
// for hub rows and per-map sums: neighbouring features share their map) are summed across the wave first and one wave instruction
// issues the atomics: 64x fewer atomics on the hot blocks.  Up to four distinct targets per wave are taken that way, one after the
// other (a wave that straddles a map boundary has two; until round 6 any second target sent ALL 64 lanes to N single-lane atomics
// each on the same few addresses -- at the lowest levels of a monocular tree, whose maps hold ~200 features, most waves straddle:
// k_tr_feat_post<2> took 4.7 ms at level 1 of a synth-16k tree against 1.3 at level 0); what is left after four adds lane by lane.
// Must be called by all 64 lanes (convergent).
template <int N>
__device__ __forceinline__ void wave_scatter_add(double* dst, const double* vals, bool valid)
{
	static_assert(N <= LSFM_WAVE, "one lane per value");
	unsigned long long todo = __ballot(valid);
	if (todo == 0ull) return;
	const int lane = threadIdx.x & (LSFM_WAVE - 1);
	const unsigned long long mine = (unsigned long long)(size_t)dst;
#pragma unroll 1
	for (int round = 0; round < 4 && todo; round++)
	{
		const int leader = __ffsll((long long)todo) - 1;
		const unsigned long long first = (unsigned long long)__shfl((long long)mine, leader, LSFM_WAVE);
		const bool same = valid && mine == first;
		const unsigned long long grp = __ballot(same);
		if (__popcll(grp) > 1)
		{
			// lane i keeps sum i: the N sums leave as ONE wave instruction over N contiguous doubles (one lane issuing N
			// single-lane atomics serialises on the same 64-byte lines: measured 1.3 ms for 7k waves on one 288-byte row)
			double keep = 0.0;
#pragma unroll
			for (int i = 0; i < N; i++)
			{
				const double sm = wave_sum(same ? vals[i] : 0.0);
				if (lane == i) keep = sm;
			}
			if (lane < N) atomic_add_f64(reinterpret_cast<double*>((size_t)first) + lane, keep);
		}
		else if (same)
		{
#pragma unroll
			for (int i = 0; i < N; i++) atomic_add_f64(dst + i, vals[i]);
		}
		todo &= ~grp;
	}
	if (valid && ((todo >> lane) & 1ull))
	{
#pragma unroll
		for (int i = 0; i < N; i++) atomic_add_f64(dst + i, vals[i]);
	}
}

// ---------------------------------------------------------------------------------------------------------
// work-group level scatter-add through a small LDS table (keys[cap] = -1 when empty, vals[cap*N])
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ int lds_slot(int* keys, int cap, int key)
{
	unsigned h = ((unsigned)key * 2654435761u) & (unsigned)(cap - 1);
	for (int probe = 0; probe < cap; probe++)
	{
		const int cur = keys[h];
		if (cur == key) return (int)h;
		if (cur == -1)
		{
			const int old = atomicCAS(&keys[h], -1, key);
			if (old == -1 || old == key) return (int)h;
		}
		h = (h + 1) & (unsigned)(cap - 1);
	}
	return -1;
}
__device__ __forceinline__ void lds_add_f64(double* p, double v)
{
	__hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// Adds x[0..N) to the accumulator of `key` for every valid lane: in the work-group's LDS table when the key fits,
// straight to gdst (global, atomics) otherwise.  Lanes of a wave that all carry the same key are summed across the wave
// first.  Must be called by all 64 lanes.  The table is flushed once per work-group by tile_flush().
template <int N>
__device__ __forceinline__ void tile_scatter_add(int* keys, double* vals, int cap, int key, double* gdst, const double* x, bool valid)
{
	const unsigned long long mask = __ballot(valid);
	if (mask == 0ull) return;
	const int leader = __ffsll((long long)mask) - 1;
	const int first = __shfl(key, leader, LSFM_WAVE);
	const bool uniform = __ballot(valid && key != first) == 0ull;
	if (uniform && __popcll(mask) > 1)
	{
		const int lane = threadIdx.x & (LSFM_WAVE - 1);
		int sl = -1;
		if (lane == leader) sl = lds_slot(keys, cap, key);
#pragma unroll
		for (int i = 0; i < N; i++)
		{
			const double s = wave_sum(valid ? x[i] : 0.0);
			if (lane == leader) { if (sl >= 0) lds_add_f64(vals + sl * N + i, s); else atomic_add_f64(gdst + i, s); }
		}
	}
	else if (valid)
	{
		const int sl = lds_slot(keys, cap, key);
#pragma unroll
		for (int i = 0; i < N; i++) { if (sl >= 0) lds_add_f64(vals + sl * N + i, x[i]); else atomic_add_f64(gdst + i, x[i]); }
	}
}
// The same, for keys that many lanes of a wave share (the hub pose of a map: every U block of the map adds to its row).  An LDS
// double-precision atomic serialises the lanes of one address at ~11 clocks each, CU-wide (tools/microbench/lds_ops.hip): 64
// lanes adding element i of their N numbers to the same accumulator, N times, is what made k_tr_ublocks 200 us per launch.  Here
// a lane parks its N numbers in its own LDS row `stage` and adds them in an order rotated by its lane number, so that the lanes of
// one key are spread over the N elements.  Must be called by all 64 lanes.
template <int N>
__device__ __forceinline__ void tile_scatter_add_rot(int* keys, double* vals, int cap, int key, double* gdst, const double* x, bool valid, double* stage)
{
	if (__ballot(valid) == 0ull) return;
	if (!valid) return;
	const int sl = lds_slot(keys, cap, key);
#pragma unroll
	for (int i = 0; i < N; i++) stage[i] = x[i];
	const int rot = (threadIdx.x & (LSFM_WAVE - 1)) % N;
#pragma unroll 1
	for (int e = 0; e < N; e++)
	{
		int i = e + rot;
		if (i >= N) i -= N;
		const double v = stage[i];
		if (sl >= 0) lds_add_f64(vals + sl * N + i, v); else atomic_add_f64(gdst + i, v);
	}
}
// after a __syncthreads(): every touched accumulator leaves the work-group once, N contiguous adds at gbase + key*N
template <int N>
__device__ __forceinline__ void tile_flush(const int* keys, const double* vals, int cap, double* gbase)
{
	for (int i = threadIdx.x; i < cap * N; i += blockDim.x)
	{
		const int k = keys[i / N];
		if (k >= 0) atomic_add_f64(gbase + (size_t)k * N + i % N, vals[i]);
	}
}

// ---------------------------------------------------------------------------------------------------------
// entry-parallel walk over the runs of a tile of features (W blocks of a feature are contiguous: run pointers fptr)
// ---------------------------------------------------------------------------------------------------------
// One lane per W block, consecutive lanes on consecutive blocks (coalesced), in rounds of blockDim.x blocks made of
// whole features (a longer run is cut into chunks).  entry(j, fl, out) computes TW values for block j of tile-local
// feature fl; they are summed per feature through LDS and handed to feat(fl, q, sum, first) once per feature and
// chunk (first = the chunk holds the feature's first block).  sFp: LDS, nft + 1 run pointers of the tile; sT: LDS,
// blockDim.x * TW doubles.  All threads of the work-group must call it.
template <int TW, class EntryFn, class FeatFn>
__device__ __forceinline__ void tile_runs(int nft, const int* sFp, double* sT, EntryFn entry, FeatFn feat)
{
	const int tid = threadIdx.x, nt = blockDim.x;
	int la = 0;
	while (la < nft)
	{
		const int e0 = sFp[la];
		int lo = la + 1, hi = nft; // largest lb with sFp[lb] - e0 <= nt (at least la + 1)
		while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (sFp[mid] - e0 <= nt) lo = mid; else hi = mid - 1; }
		const int lb = lo, e1 = sFp[lb];
		for (int ce0 = e0;; ce0 += nt)
		{
			const int ce1 = min(ce0 + nt, e1);
			const int j = ce0 + tid;
			if (j < ce1)
			{
				int l2 = la, h2 = lb - 1; // feature of block j: last fl with sFp[fl] <= j
				while (l2 < h2) { const int mid = (l2 + h2 + 1) >> 1; if (sFp[mid] <= j) l2 = mid; else h2 = mid - 1; }
				entry(j, l2, &sT[tid * TW]);
			}
			__syncthreads();
			for (int idx = tid; idx < (lb - la) * TW; idx += nt)
			{
				const int fl = la + idx / TW, q = idx % TW;
				const int r0 = max(sFp[fl], ce0) - ce0, r1 = min(sFp[fl + 1], ce1) - ce0;
				double sum = 0.0;
				for (int r = r0; r < r1; r++) sum += sT[r * TW + q];
				if (r1 > r0 || sFp[fl] >= ce0) feat(fl, q, sum, sFp[fl] >= ce0);
			}
			__syncthreads();
			if (ce1 >= e1) break;
		}
		la = lb;
	}
}

} // namespace lsfm
