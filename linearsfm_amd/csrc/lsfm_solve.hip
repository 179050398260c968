// Batched Schur complement + preconditioned CG + back-substitution for all joins of one tree level.
// Replaces lmj_solveLinearSFM{Stereo,Mono} (Imp.cpp:2119-2378 / 6756-7041): the reference builds S = U - W V^-1 W^T
// through a dense m x m byte mask and hands it to CHOLMOD; here
//   K7  k_vinv            V^-1 per feature                                   (pba_inverseV, Imp.cpp:3022-3042)
//   K8  k_pat_*           block pattern of S by hashing the pose pairs that share a feature + U's pattern
//                         (replaces smask, Imp.cpp:2131-2205), sorted into block-CSR
//   K9  k_schur_*         S(p,q) -= W_pf V_f^-1 W_qf^T, E_p -= W_pf V_f^-1 eb_f (Imp.cpp:2244-2332): k_schur_panel
//                         (lsfm_schur_panel.hip, fp64 MFMA on 128-feature tiles); k_schur_w here takes the tiles it flags
//   K10 k_spmv / k_pcg_*  Cholesky-preconditioned CG (lsfm_pcg.hip) on the symmetric 6x6-block system, all independent
//                         systems of the level together with per-system scalars (replaces Imp.cpp:2334-2361)
//   K11 k_backsub         features: x_f = V_f^-1 (eb_f - sum W_pf^T x_p)     (pba_solveFeatures, Imp.cpp:2980-3020)
// S is kept as its upper block triangle (diagonal blocks full); the SpMV reads every stored block once and adds its
// mirrored part through an LDS window of y (k_spmv).
#include <algorithm>
#include <cmath>

#include "lsfm_device.hpp"
#include "lsfm_internal.hpp"
#include "lsfm_join.hpp"
#include "lsfm_solve.hpp"

namespace lsfm {

static const unsigned long long HEMPTY = ~0ull;

// also what the panel kernel of K9 needs per feature: V^-1 = L L^T (l00 l10 l11 l20 l21 l22) and y = L^T eb, so that its
// passes find them ready instead of running a Cholesky with square roots and divisions on a dependent load each;
// l00 = NaN marks a V^-1 without a Cholesky factor (the tile then goes to k_schur_w)
// ymax (optional): per work-group the largest |L^T eb|^2 of its features -- what bounds K9's right-hand-side sums (k_schur_scale)
// rhs (fused right-hand side): eb is the V part only; also leaves uu = [L^-1 x_f of the End source | of the Cur source]
__global__ void __launch_bounds__(256) k_vinv(int NF, const double* __restrict__ V, const double* __restrict__ eb, double* __restrict__ IV, double* __restrict__ LY,
                                               double* __restrict__ ymax, RhsFused rhs, double* __restrict__ uu)
{
	__shared__ double wmax[4];
	int f = blockIdx.x * blockDim.x + threadIdx.x;
	double y2 = 0.0;
	if (f < NF)
	{
	double a[9], o[9];
	ld<9>(a, V + (size_t)f * 9);
	inv3_sym(a, o);
	st<9>(IV + (size_t)f * 9, o);
	const double* e = eb + (size_t)f * 3;
	double l[9];
	const double d0 = o[0];
	l[0] = sqrt(d0);
	l[1] = o[3] / l[0];
	l[3] = o[6] / l[0];
	const double d1 = o[4] - l[1] * l[1];
	l[2] = sqrt(d1);
	l[4] = (o[7] - l[3] * l[1]) / l[2];
	const double d2 = o[8] - l[3] * l[3] - l[4] * l[4];
	l[5] = sqrt(d2);
	l[6] = l[0] * e[0] + l[1] * e[1] + l[3] * e[2];
	l[7] = l[2] * e[1] + l[4] * e[2];
	l[8] = l[5] * e[2];
	if (!(d0 > 0.0) || !(d1 > 0.0) || !(d2 > 0.0)) l[0] = __builtin_nan("");
	st<9>(LY + (size_t)f * 9, l);
	y2 = l[6] * l[6] + l[7] * l[7] + l[8] * l[8];
	if (uu)
	{
#pragma unroll
		for (int sd = 0; sd < 2; sd++)
		{
			const int fs = sd ? rhs.srcC[f] : rhs.srcE[f];
			double u0 = 0.0, u1 = 0.0, u2 = 0.0;
			if (fs >= 0)
			{
				const double* x = rhs.feat_src + (size_t)fs * 3;
				u0 = x[0] / l[0];
				u1 = (x[1] - l[1] * u0) / l[2];
				u2 = (x[2] - l[3] * u0 - l[4] * u1) / l[5];
			}
			uu[(size_t)f * 6 + 3 * sd] = u0; uu[(size_t)f * 6 + 3 * sd + 1] = u1; uu[(size_t)f * 6 + 3 * sd + 2] = u2;
			y2 = fmax(y2, u0 * u0 + u1 * u1 + u2 * u2);
		}
	}
	if (!(y2 == y2)) y2 = 0.0; // (a feature without a factor goes to k_schur_w, which checks its own sums)
	}
	if (!ymax) return;
#pragma unroll
	for (int off = 32; off > 0; off >>= 1) y2 = fmax(y2, __shfl_xor(y2, off, LSFM_WAVE));
	if ((threadIdx.x & (LSFM_WAVE - 1)) == 0) wmax[threadIdx.x >> 6] = y2;
	__syncthreads();
	if (threadIdx.x == 0) ymax[blockIdx.x] = fmax(fmax(wmax[0], wmax[1]), fmax(wmax[2], wmax[3]));
}

// The scales of K9's fixed-point sums.  Per pose scalar i: sexp[i] with 2^sexp[i] > 2 sqrt(U_ii), U_ii the diagonal entry of the
// camera block of the joint information matrix (k_schur_u has put U into S: the diagonal block is the first of its row).  S = U - W
// V^-1 W^T is a Schur complement of a positive semi-definite matrix, so every partial sum over features of (W V^-1 W^T)_ij stays below
// sqrt(U_ii U_jj) (Cauchy-Schwarz on the semi-definite partial sums) -- known BEFORE K9 runs.  sexp[6 M] = ey with
// 2^ey > 2 sqrt(NF max_f |L_f^T eb_f|^2) >= 2 |L^T eb|: |(W V^-1 eb)_i| = |(W L)(L^T eb)|_i <= sqrt(U_ii) |L^T eb|.
__device__ __forceinline__ int half_exponent(double d) // smallest-ish e with 2^e > 2 sqrt(d); 0 for d <= 0 (an empty row: nothing is ever added)
{
	if (!(d > 0.0) || !(d < 1e300)) return 0;
	int k;
	(void)frexp(d, &k); // d = f 2^k, 1/2 <= f < 1: sqrt(d) < 2^((k + 1) >> 1)
	return ((k + 1) >> 1) + 1;
}
__global__ void __launch_bounds__(256) k_schur_scale(int M, const int* __restrict__ rowptr, const double* __restrict__ S, int NF, int nymax,
                                                      const double* __restrict__ ymax, const unsigned char* __restrict__ fixed, int* __restrict__ sexp,
                                                      const double* __restrict__ ea, double* __restrict__ E)
{
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i < 6 * M)
	{
		E[i] = ea[i]; // (the right-hand side starts as its pose part: a copy of its own until round 5)
		const int p = i / 6, r = i - 6 * p;
		// (a scalar removed from the system -- the Mono gauge -- has no bound and needs none: a scale beyond every addend, whatever lands
		// in its row and column rounds to nothing)
		sexp[i] = (fixed && fixed[i]) ? 400 : half_exponent(S[(size_t)rowptr[p] * 36 + 7 * r]);
	}
	if (blockIdx.x == 0)
	{
		__shared__ double wmax[4];
		double m = 0.0;
		for (int b = threadIdx.x; b < nymax; b += 256) m = fmax(m, ymax[b]);
#pragma unroll
		for (int off = 32; off > 0; off >>= 1) m = fmax(m, __shfl_xor(m, off, LSFM_WAVE));
		if ((threadIdx.x & (LSFM_WAVE - 1)) == 0) wmax[threadIdx.x >> 6] = m;
		__syncthreads();
		// (+ 20: with the fused right-hand side y gains P^T x_p, of the size of L^T eb but not bounded by it -- two limbs leave 100 bits)
		if (threadIdx.x == 0) sexp[6 * (size_t)M] = half_exponent((double)NF * fmax(fmax(wmax[0], wmax[1]), fmax(wmax[2], wmax[3]))) + 20;
	}
}
// S -= (the integer sums of W V^-1 W^T); E += (the two limbs of -W V^-1 eb); a poisoned level leaves NaN everywhere (its
// factorisation then reports the system, as a NaN in a floating-point sum did)
__global__ void k_schur_finish(int nnzb, int M, const unsigned long long* __restrict__ keys, const long long* __restrict__ acc, const int* __restrict__ sexp,
                               double* __restrict__ S, double* __restrict__ E)
{
	// acc: [poison word | nnzb * 36 sums of W V^-1 W^T | 6 M high limbs | 6 M low limbs]
	const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	const size_t ns = (size_t)nnzb * 36, ne = (size_t)M * 6;
	const bool poison = acc[0] != 0;
	acc += 1;
	if (i < ns)
	{
		const unsigned long long key = keys[i / 36];
		const int p = (int)(key >> 32), q = (int)(key & 0xffffffffull), rc = (int)(i % 36);
		const double v = ldexp((double)acc[i], sexp[6 * (size_t)p + rc / 6] + sexp[6 * (size_t)q + rc % 6] - 60);
		S[i] = poison ? __builtin_nan("") : S[i] - v;
	}
	else if (i < ns + ne)
	{
		const size_t k = i - ns;
		const int e = sexp[k] + sexp[ne] - 62;
		const double v = ldexp((double)acc[ns + k], e) + ldexp((double)acc[ns + ne + k], e - 40);
		E[k] = poison ? __builtin_nan("") : E[k] + v;
	}
}

// ---- hash set of block coordinates ------------------------------------------------------------------------
__device__ __forceinline__ unsigned long long mix64(unsigned long long x)
{
	x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33;
	return x;
}
__device__ __forceinline__ void hash_insert(unsigned long long* tab, unsigned long long mask, unsigned long long key, int* overflow)
{
	unsigned long long h = mix64(key) & mask;
	for (int probe = 0; probe < 4096; probe++)
	{
		unsigned long long cur = tab[h];
		if (cur == key) return;
		if (cur == HEMPTY)
		{
			unsigned long long old = atomicCAS(&tab[h], HEMPTY, key);
			if (old == HEMPTY || old == key) return;
		}
		h = (h + 1) & mask;
	}
	*overflow = 1;
}
__device__ __forceinline__ int hash_find(const unsigned long long* __restrict__ tab, const int* __restrict__ val, unsigned long long mask,
                                         unsigned long long key)
{
	unsigned long long h = mix64(key) & mask;
	for (int probe = 0; probe < 4096; probe++)
	{
		unsigned long long cur = tab[h];
		if (cur == key) return val[h];
		if (cur == HEMPTY) return -1;
		h = (h + 1) & mask;
	}
	return -1;
}
__device__ __forceinline__ unsigned long long pair_key(int p, int q)
{
	return p <= q ? (((unsigned long long)(unsigned)p << 32) | (unsigned)q) : (((unsigned long long)(unsigned)q << 32) | (unsigned)p);
}

__global__ void k_fill_u64(unsigned long long* p, size_t n, unsigned long long v)
{
	size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) p[i] = v;
}

__global__ void k_pat_insert_u(int NU, int M, const int* __restrict__ Ui, const int* __restrict__ Uj, unsigned long long* tab,
                               unsigned long long mask, int* overflow)
{
	int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i < NU) hash_insert(tab, mask, pair_key(Ui[i], Uj[i]), overflow);
	if (i < M) hash_insert(tab, mask, pair_key(i, i), overflow); // every block row owns its diagonal block
}

// one lane per feature; all pose pairs of its W run (Imp.cpp:2155-2173).  When the whole wave asks for the same
// pair (hub poses shared by neighbouring features) one lane inserts.
__global__ void __launch_bounds__(256)
k_pat_insert_w(int NF, const int* __restrict__ fptr, const int* __restrict__ photo, unsigned long long* tab, unsigned long long mask,
               int* overflow)
{
	int f = blockIdx.x * blockDim.x + threadIdx.x;
	const bool inb = f < NF;
	int j0 = 0, len = 0;
	if (inb) { j0 = fptr[f]; len = fptr[f + 1] - j0; }
	int maxlen = len;
#pragma unroll
	for (int off = 32; off > 0; off >>= 1) maxlen = max(maxlen, __shfl_xor(maxlen, off, LSFM_WAVE));
	const int lane = threadIdx.x & (LSFM_WAVE - 1);
	for (int a = 0; a < maxlen; a++)
	{
		const int pa = (inb && a < len) ? photo[j0 + a] : -1;
		for (int b = a + 1; b < maxlen; b++)
		{
			const bool v = inb && b < len;
			unsigned long long key = v ? pair_key(pa, photo[j0 + b]) : 0ull;
			unsigned long long m = __ballot(v);
			if (m == 0ull) continue;
			int leader = __ffsll((long long)m) - 1;
			unsigned long long first = (unsigned long long)__shfl((long long)key, leader, LSFM_WAVE);
			bool uniform = __ballot(v && key != first) == 0ull;
			if (uniform) { if (lane == leader) hash_insert(tab, mask, key, overflow); }
			else if (v) hash_insert(tab, mask, key, overflow);
		}
	}
}

// sum over the features of (run length)^2: the pose pairs of K9 (for its algorithmic flop count)
__global__ void __launch_bounds__(256) k_sum_run_squares(int NF, const int* __restrict__ fptr, unsigned long long* out)
{
	// a few hundred work-groups striding over the features, ONE atomic each (one per wave on a single address took 120 us)
	__shared__ unsigned long long part[4];
	unsigned long long v = 0;
	for (int f = blockIdx.x * blockDim.x + threadIdx.x; f < NF; f += gridDim.x * blockDim.x)
	{
		const unsigned long long k = (unsigned long long)(fptr[f + 1] - fptr[f]);
		v += k * k;
	}
#pragma unroll
	for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, LSFM_WAVE);
	if ((threadIdx.x & (LSFM_WAVE - 1)) == 0) part[threadIdx.x >> 6] = v;
	__syncthreads();
	if (threadIdx.x == 0) { v = part[0] + part[1] + part[2] + part[3]; if (v) atomicAdd(out, v); }
}

__global__ void k_pat_compact(size_t cap, const unsigned long long* __restrict__ tab, unsigned long long* __restrict__ list, int* __restrict__ count)
{
	size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= cap) return;
	unsigned long long k = tab[i];
	if (k != HEMPTY) list[atomicAdd(count, 1)] = k;
}

__global__ void k_pat_assign(int nnzb, const unsigned long long* __restrict__ sorted, const unsigned long long* __restrict__ tab,
                             int* __restrict__ val, unsigned long long mask, int* __restrict__ colidx)
{
	int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= nnzb) return;
	unsigned long long key = sorted[i];
	unsigned long long h = mix64(key) & mask;
	while (tab[h] != key) h = (h + 1) & mask;
	val[h] = i;
	colidx[i] = (int)(key & 0xffffffffull);
}

// rowptr[r] = first index whose key >= (r << 32)
__global__ void k_rowptr_from_keys(int rows, int n, const unsigned long long* __restrict__ sorted, int shift, int* __restrict__ rowptr)
{
	int r = blockIdx.x * blockDim.x + threadIdx.x;
	if (r > rows) return;
	unsigned long long target = (unsigned long long)(unsigned)r << shift;
	int lo = 0, hi = n;
	while (lo < hi) { int mid = (lo + hi) >> 1; if (sorted[mid] < target) lo = mid + 1; else hi = mid; }
	rowptr[r] = lo;
}

// ---- numeric Schur complement ------------------------------------------------------------------------------
// (one lane per block looks its place in S up; the 36 numbers of the work-group's blocks then leave by consecutive lanes on consecutive
// numbers -- one lane per block adding its 36 at a stride of 288 bytes was 63 us for the 37 MB of U at the top of an NC3500-like tree)
#define SCHUR_U_BLOCKS 256
__global__ void __launch_bounds__(SCHUR_U_BLOCKS) k_schur_u(int NU, const double* __restrict__ U, const int* __restrict__ Ui, const int* __restrict__ Uj,
                                                          const unsigned long long* __restrict__ tab, const int* __restrict__ val, unsigned long long mask,
                                                          double* __restrict__ S)
{
	__shared__ int s_slot[SCHUR_U_BLOCKS]; // slot of S, | 1 << 30: the block goes in transposed
	const int i0 = blockIdx.x * SCHUR_U_BLOCKS, nb = min(SCHUR_U_BLOCKS, NU - i0);
	if ((int)threadIdx.x < nb)
	{
		const int a = Ui[i0 + threadIdx.x], b = Uj[i0 + threadIdx.x];
		const int slot = hash_find(tab, val, mask, pair_key(a, b));
		s_slot[threadIdx.x] = slot < 0 ? -1 : (slot | (a <= b ? 0 : (1 << 30)));
	}
	__syncthreads();
	const double* u = U + (size_t)i0 * 36;
	for (int p = threadIdx.x; p < nb * 36; p += SCHUR_U_BLOCKS)
	{
		const int blk = p / 36, q = p - 36 * blk, sl = s_slot[blk];
		if (sl < 0) continue;
		const int slot = sl & ((1 << 30) - 1);
		const int qq = (sl >> 30) ? (q % 6) * 6 + q / 6 : q; // (u[r][c] -> s[c][r])
		atomic_add_f64(S + (size_t)slot * 36 + qq, u[p]);
	}
}

// K9, one lane per feature.  A tile of SCHUR_TILE consecutive features (one work-group) touches few distinct blocks
// of S (its ~10 hub poses squared plus the handful of poses that observe the tile), while every feature contributes
// k_f(k_f+1)/2 of them: all contributions are first summed in an LDS table keyed by the block's slot (ds_add_f64) and
// each touched block leaves the work-group once, as 36 contiguous atomics.  (Scattering 8-byte atomics straight to HBM
// ran at ~0.08 TB/s, MI355X_MICROARCH "Global float atomics": 64 lanes in 64 different rows.)
#define SCHUR_TILE 128
#define SCHUR_CAP 256 /* LDS block accumulators per work-group: 256 * 288 B = 72 KiB -> 2 work-groups per CU */
#define SCHUR_ECAP 64 /* LDS accumulators for E (per pose) */

__global__ void __launch_bounds__(SCHUR_TILE)
k_schur_w(int NF, const int* __restrict__ fptr, const int* __restrict__ photo, const double* __restrict__ W, const double* __restrict__ IV,
          const double* __restrict__ eb, const unsigned long long* __restrict__ tab, const int* __restrict__ val, unsigned long long mask,
          K9Out o, const unsigned char* __restrict__ only, const double* __restrict__ LY)
{
	if (only && !only[blockIdx.x / (LSFM_PM_TILE / SCHUR_TILE)]) return; // fallback pass: only the tiles the panel kernel could not take
	// (the same fixed-point sums as the panel kernel's, k_schur_scale: per feature here, so already the LDS tables are integers)
	__shared__ int skey[SCHUR_CAP];
	__shared__ int ekey[SCHUR_ECAP];
	__shared__ long long sval[SCHUR_CAP * 36];
	__shared__ long long evalv[SCHUR_ECAP * 12]; // high limbs, then low limbs
	const int tid = threadIdx.x;
	for (int i = tid; i < SCHUR_CAP; i += SCHUR_TILE) skey[i] = -1;
	for (int i = tid; i < SCHUR_ECAP; i += SCHUR_TILE) ekey[i] = -1;
	for (int i = tid; i < SCHUR_CAP * 36; i += SCHUR_TILE) sval[i] = 0;
	for (int i = tid; i < SCHUR_ECAP * 12; i += SCHUR_TILE) evalv[i] = 0;
	__syncthreads();
	const int f = blockIdx.x * SCHUR_TILE + tid;
	bool bad = false;
	const int ey = *o.ey;
	if (f < NF)
	{
		const int j0 = fptr[f], len = fptr[f + 1] - j0;
		double iv[9];
		ld<9>(iv, IV + (size_t)f * 9);
		double eb0 = eb[(size_t)f * 3], eb1 = eb[(size_t)f * 3 + 1], eb2 = eb[(size_t)f * 3 + 2];
		double xfs[2][3] = { { 0.0, 0.0, 0.0 }, { 0.0, 0.0, 0.0 } };
		if (o.xpose)
		{
			// fused right-hand side (K9Out): eb += W^T x_p over the run; the feature's estimate in either source map = L u
			for (int a = 0; a < len; a++)
			{
				const double* wa = W + (size_t)(j0 + a) * 18;
				const double* xp = o.xpose + (size_t)photo[j0 + a] * 6;
#pragma unroll
				for (int r = 0; r < 6; r++) { eb0 = fma(wa[3 * r], xp[r], eb0); eb1 = fma(wa[3 * r + 1], xp[r], eb1); eb2 = fma(wa[3 * r + 2], xp[r], eb2); }
			}
			const double* l = LY + (size_t)f * 9;
			if (!(l[0] == l[0])) bad = true; // (V^-1 without a factor: no y to leave -- the level is reported, as a NaN in S would be)
#pragma unroll
			for (int sd = 0; sd < 2; sd++)
			{
				const double* u = o.uu + (size_t)f * 6 + 3 * sd;
				xfs[sd][0] = l[0] * u[0]; xfs[sd][1] = l[1] * u[0] + l[2] * u[1]; xfs[sd][2] = l[3] * u[0] + l[4] * u[1] + l[5] * u[2];
			}
		}
		for (int a = 0; a < len; a++)
		{
			double WV[18], Wa[18];
			const int pa = photo[j0 + a];
			ld<18>(Wa, W + (size_t)(j0 + a) * 18);
			mm<6, 3, 3, false>(Wa, iv, WV); // W V^-1 (V^-1 symmetric), Imp.cpp:2260-2273
			int ea6[6];
#pragma unroll
			for (int r = 0; r < 6; r++) ea6[r] = o.sexp[(size_t)pa * 6 + r];
			{
				// E_p -= W V^-1 eb, Imp.cpp:2321-2328
				const int es = lds_slot(ekey, SCHUR_ECAP, pa);
#pragma unroll
				for (int r = 0; r < 6; r++)
				{
					double e = -(WV[3 * r] * eb0 + WV[3 * r + 1] * eb1 + WV[3 * r + 2] * eb2);
					if (o.xpose) { const double* xf = xfs[o.pside[pa] & 1]; e += Wa[3 * r] * xf[0] + Wa[3 * r + 1] * xf[1] + Wa[3 * r + 2] * xf[2]; }
					long long hi, lo;
					to_fixed2(e, 62 - ea6[r] - ey, hi, lo, bad);
					if (es >= 0) { lds_add_i64(&evalv[es * 6 + r], hi); lds_add_i64(&evalv[SCHUR_ECAP * 6 + es * 6 + r], lo); }
					else { atomic_add_i64(o.Ehi + (size_t)pa * 6 + r, hi); atomic_add_i64(o.Elo + (size_t)pa * 6 + r, lo); }
				}
			}
			for (int b = a; b < len; b++)
			{
				double Wb[18], T[36];
				const int pb = photo[j0 + b];
				ld<18>(Wb, W + (size_t)(j0 + b) * 18);
				mmt<6, 3, 6, false>(WV, Wb, T); // W_a V^-1 W_b^T = contribution to S(pa,pb), Imp.cpp:2283-2318
				const int slot = hash_find(tab, val, mask, pair_key(pa, pb));
				const int ls = lds_slot(skey, SCHUR_CAP, slot);
				long long* dl = ls >= 0 ? &sval[ls * 36] : nullptr;
				long long* dg = o.S + (size_t)slot * 36;
				int eb6[6];
#pragma unroll
				for (int c = 0; c < 6; c++) eb6[c] = o.sexp[(size_t)pb * 6 + c];
				// stored orientation: rows = the smaller pose index; a block of one pose with itself is stored full
				const bool same = (pa == pb), tr = pa > pb, twice = same && a != b;
#pragma unroll
				for (int r = 0; r < 6; r++)
#pragma unroll
					for (int c = 0; c < 6; c++)
					{
						// (+: the accumulators hold W V^-1 W^T, k_schur_finish subtracts)
						double x = T[r * 6 + c];
						if (twice) x += T[c * 6 + r];
						const int at = tr ? c * 6 + r : r * 6 + c;
						const long long q = to_fixed(x, 60 - ea6[r] - eb6[c], bad);
						if (dl) lds_add_i64(dl + at, q); else atomic_add_i64(dg + at, q);
					}
			}
		}
	}
	if (bad) atomic_add_i64(o.poison, 1);
	__syncthreads();
	for (int i = tid; i < SCHUR_CAP * 36; i += SCHUR_TILE)
	{
		const int k = skey[i / 36];
		if (k >= 0 && sval[i]) atomic_add_i64(o.S + (size_t)k * 36 + i % 36, sval[i]);
	}
	for (int i = tid; i < SCHUR_ECAP * 6; i += SCHUR_TILE)
	{
		const int k = ekey[i / 6];
		if (k >= 0)
		{
			if (evalv[i]) atomic_add_i64(o.Ehi + (size_t)k * 6 + i % 6, evalv[i]);
			if (evalv[SCHUR_ECAP * 6 + i]) atomic_add_i64(o.Elo + (size_t)k * 6 + i % 6, evalv[SCHUR_ECAP * 6 + i]);
		}
	}
}

// ---- K10a: y = S x on the upper-block storage, every stored block read ONCE --------------------------------
// A work-group owns SPT consecutive block rows.  Lane group g (8 lanes, lane r < 6 = scalar row r) takes blocks (p,q) of
// those rows: the row part B x_q goes to y_p and the mirrored part B^T x_p to y_q, both into an LDS window of y that
// covers the tile's rows and the SPW rows after them (S is a pose chain: q - p is small), flushed once with
// contiguous atomics.  Targets outside the window -- the columns of hub poses -- go through a small LDS table keyed
// by row.  Rows longer than SP_LONG blocks (the hub poses: up to a block per pose) are not walked by their own tile;
// instead every tile takes the hub blocks (h,q) whose COLUMN q it owns (their mirrored part lands in its window, their
// row part in the table under h).  So nothing is read twice, no wave scatters 8-byte atomics over 64 rows, and the
// index is just the upper block CSR plus the list of long rows.  (First version: a row-sorted index of both
// orientations -- every block read twice, one more sort per level.)   y must be zero on entry.
// dotw != null: dot[seg] += w . y (fused p.Ap).
#define SPT 16   /* block rows per work-group */
#define SPW 64   /* rows of the window after the tile */
#define SP_LONG 256
#define SP_FAR 64
#define SPMV_CACHED_BYTES ((size_t)96 << 20) /* a matrix up to this size is multiplied by k_spmv_gather (stays in L2 / Infinity Cache) */
__global__ void k_spmv_long_rows(int M, const int* __restrict__ rowptr, int* __restrict__ nlong, int* __restrict__ longrows)
{
	int r = blockIdx.x * blockDim.x + threadIdx.x;
	if (r < M && rowptr[r + 1] - rowptr[r] > SP_LONG) longrows[atomicAdd(nlong, 1)] = r;
}
__global__ void __launch_bounds__(256)
k_spmv(int M, const int* __restrict__ rowptr, const int* __restrict__ colidx, const int* __restrict__ nlong_p, const int* __restrict__ longrows,
       const double* __restrict__ S, const double* __restrict__ x, double* __restrict__ y, const unsigned char* __restrict__ fixed,
       const double* __restrict__ dotw, const int* __restrict__ pose_seg, double* __restrict__ dot, int dot_stride)
{
	__shared__ int pre[SPT + 1];               // prefix of the short rows' lengths
	__shared__ int rp[SPT + 1];
	__shared__ double ywin[(SPT + SPW) * 6];
	__shared__ int fkeys[SP_FAR];
	__shared__ double fvals[SP_FAR * 6];
	const int tid = threadIdx.x, g = tid >> 3, r = tid & 7, ng = blockDim.x >> 3;
	const int r0 = blockIdx.x * SPT, r1 = min(r0 + SPT, M), nr = r1 - r0;
	for (int i = tid; i < (SPT + SPW) * 6; i += blockDim.x) ywin[i] = 0.0;
	for (int i = tid; i < SP_FAR; i += blockDim.x) fkeys[i] = -1;
	for (int i = tid; i < SP_FAR * 6; i += blockDim.x) fvals[i] = 0.0;
	for (int i = tid; i <= nr; i += blockDim.x) rp[i] = rowptr[r0 + i];
	__syncthreads();
	if (tid == 0)
	{
		int acc = 0;
		for (int i = 0; i < nr; i++) { pre[i] = acc; const int len = rp[i + 1] - rp[i]; acc += len > SP_LONG ? 0 : len; }
		pre[nr] = acc;
	}
	__syncthreads();
	// target (row t, scalar r): window, table or global
	auto add = [&](int t, double v) {
		const int w = t - r0;
		if (w >= 0 && w < SPT + SPW) { lds_add_f64(&ywin[w * 6 + r], v); return; }
		const int sl = lds_slot(fkeys, SP_FAR, t);
		if (sl >= 0) { lds_add_f64(&fvals[sl * 6 + r], v); return; }
		// table full (a band wider than the window AND more far rows than slots): straight to memory, with its share of
		// the fused dot product
		if (fixed && fixed[(size_t)t * 6 + r]) return;
		atomic_add_f64(y + (size_t)t * 6 + r, v);
		if (dotw) atomic_add_f64(dot + (size_t)pose_seg[t] * dot_stride, dotw[(size_t)t * 6 + r] * v);
	};
	auto block = [&](int k, int p) {
		const int q = colidx[k];
		const double* blk = S + (size_t)k * 36;
		const double* xq = x + (size_t)q * 6;
		double sum = 0.0;
#pragma unroll
		for (int j = 0; j < 6; j++) sum = fma(blk[r * 6 + j], xq[j], sum);
		add(p, sum);
		if (q != p)
		{
			const double* xp = x + (size_t)p * 6;
			double t = 0.0;
#pragma unroll
			for (int j = 0; j < 6; j++) t = fma(blk[j * 6 + r], xp[j], t);
			add(q, t);
		}
	};
	// (1) the short rows of the tile
	const int nb = pre[nr];
	if (r < 6)
		for (int i = g; i < nb; i += ng)
		{
			int lo = 0, hi = nr - 1; // row of item i: last row with pre[row] <= i (rows of length 0 are skipped by the <=)
			while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (pre[mid] <= i) lo = mid; else hi = mid - 1; }
			block(rp[lo] + (i - pre[lo]), r0 + lo);
		}
	// (2) the long rows: their blocks in the tile's columns.  One lane per long row finds where the tile's columns start
	// in it (all searches run side by side), then the lane groups take the blocks
	__shared__ int lstart[SP_FAR];
	const int nlong = *nlong_p;
	for (int l0 = 0; l0 < nlong; l0 += SP_FAR)
	{
		const int nl = min(SP_FAR, nlong - l0);
		__syncthreads();
		if (tid < nl)
		{
			const int h = longrows[l0 + tid];
			int lo = rowptr[h], hi = rowptr[h + 1]; // first block of row h with column >= r0
			// a hub row is (nearly) dense from its diagonal on: try the position a dense row would have first
			const int guess = lo + max(r0 - h, 0);
			if (guess < hi && colidx[guess] >= r0 && (guess == lo || colidx[guess - 1] < r0)) lo = guess;
			else
				while (lo < hi) { const int mid = (lo + hi) >> 1; if (colidx[mid] < r0) lo = mid + 1; else hi = mid; }
			lstart[tid] = lo;
		}
		__syncthreads();
		if (r < 6)
			for (int li = 0; li < nl; li++)
			{
				const int h = longrows[l0 + li], he = rowptr[h + 1];
				for (int k = lstart[li] + g; k < he && colidx[k] < r1; k += ng) block(k, h);
			}
	}
	__syncthreads();
	// (3) flush: the window (contiguous), then the far rows
	double w = 0.0;
	int seg = 0;
	bool any = false;
	for (int i = tid; i < (SPT + SPW) * 6; i += blockDim.x)
	{
		const int t = r0 + i / 6;
		const double v = ywin[i];
		if (t < M && v != 0.0 && !(fixed && fixed[(size_t)t * 6 + i % 6]))
		{
			atomic_add_f64(y + (size_t)t * 6 + i % 6, v);
			if (dotw)
			{
				const int sg = pose_seg[t];
				if (any && sg != seg) { atomic_add_f64(dot + (size_t)seg * dot_stride, w); w = 0.0; }
				seg = sg; any = true;
				w = fma(dotw[(size_t)t * 6 + i % 6], v, w);
			}
		}
	}
	for (int i = tid; i < SP_FAR * 6; i += blockDim.x)
	{
		const int t = fkeys[i / 6];
		const double v = fvals[i];
		if (t >= 0 && v != 0.0 && !(fixed && fixed[(size_t)t * 6 + i % 6]))
		{
			atomic_add_f64(y + (size_t)t * 6 + i % 6, v);
			if (dotw)
			{
				const int sg = pose_seg[t];
				if (any && sg != seg) { atomic_add_f64(dot + (size_t)seg * dot_stride, w); w = 0.0; }
				seg = sg; any = true;
				w = fma(dotw[(size_t)t * 6 + i % 6], v, w);
			}
		}
	}
	if (dotw) wave_scatter_add<1>(dot + (size_t)seg * dot_stride, &w, any);
}

// ---- K10a, cache-resident variant ------------------------------------------------------------------------------
// The Schur matrix of a tree level is at most a few tens of MB: it stays in L2 / Infinity Cache between the handful of
// products a level needs, so reading a block twice costs no HBM traffic, and with a path that revisits the band
// assumption of k_spmv (q - p small, a few hub rows) no longer holds -- at the top of the NC3500-like tree the windowed
// kernel spent 130 us per product walking ~100 "long" rows in every work-group.  Here both orientations of every block
// are listed once, sorted by the row they contribute to (structure only: built with the pattern, kept by the plan):
// entry = (target row << 32 | block << 1 | transposed), other[e] = the pose whose x it multiplies.  The product is a
// segmented sum over that list: 8 entries per 8-lane group (lane r < 6 = scalar row r), 256 entries per work-group,
// summed in an LDS window over the work-group's rows (at most 256: every row holds its diagonal block) and flushed with
// one atomic per touched scalar -- perfectly balanced whatever the row lengths are.  y must be zero on entry.
#define SPG_EPG 8
#define SPG_ENT (32 * SPG_EPG) /* entries per work-group of 256 lanes */
__global__ void k_spmv_gather_keys(int nnzb, const unsigned long long* __restrict__ upper_keys, unsigned long long* __restrict__ ent,
                                   int* __restrict__ oth)
{
	const int k = blockIdx.x * blockDim.x + threadIdx.x;
	if (k >= nnzb) return;
	const unsigned long long key = upper_keys[k];
	const unsigned p = (unsigned)(key >> 32), q = (unsigned)(key & 0xffffffffull);
	ent[k] = ((unsigned long long)p << 32) | ((unsigned)k << 1);
	oth[k] = (int)q;
	// the mirrored part of an off-diagonal block; diagonal blocks leave a hole that sorts to the end
	ent[(size_t)nnzb + k] = p != q ? (((unsigned long long)q << 32) | ((unsigned)k << 1) | 1u) : ~0ull;
	oth[(size_t)nnzb + k] = (int)p;
}
__global__ void __launch_bounds__(256)
k_spmv_gather(int M, int nent, const unsigned long long* __restrict__ ent, const int* __restrict__ oth, const double* __restrict__ S,
              const double* __restrict__ x, double* __restrict__ y, const unsigned char* __restrict__ fixed, const double* __restrict__ dotw,
              const int* __restrict__ pose_seg, double* __restrict__ dot, int dot_stride)
{
	__shared__ double ywin[SPG_ENT * 6];
	const int tid = threadIdx.x, g = tid >> 3, r = tid & 7;
	const int eb = blockIdx.x * SPG_ENT;
	const int row0 = (int)(ent[eb] >> 32); // first entry of the work-group: never a hole while eb < number of real entries
	for (int i = tid; i < SPG_ENT * 6; i += 256) ywin[i] = 0.0;
	__syncthreads();
	const int e0 = eb + g * SPG_EPG;
	if (r < 6 && e0 < nent && row0 >= 0)
	{
		unsigned long long key[SPG_EPG];
		int ot[SPG_EPG];
#pragma unroll
		for (int i = 0; i < SPG_EPG; i++)
		{
			const bool in = e0 + i < nent;
			key[i] = in ? ent[e0 + i] : ~0ull;
			ot[i] = in ? oth[e0 + i] : 0;
		}
		double sum[SPG_EPG];
#pragma unroll
		for (int i = 0; i < SPG_EPG; i++)
		{
			sum[i] = 0.0;
			if (key[i] == ~0ull) continue;
			const unsigned kk = (unsigned)(key[i] & 0xffffffffull);
			const double* blk = S + (size_t)(kk >> 1) * 36;
			const double* xo = x + (size_t)ot[i] * 6;
			double s = 0.0;
			if (kk & 1u)
			{
#pragma unroll
				for (int j = 0; j < 6; j++) s = fma(blk[j * 6 + r], xo[j], s);
			}
			else
			{
#pragma unroll
				for (int j = 0; j < 6; j++) s = fma(blk[r * 6 + j], xo[j], s);
			}
			sum[i] = s;
		}
		auto put = [&](int row, double v) {
			if (row - row0 < SPG_ENT) { lds_add_f64(&ywin[(row - row0) * 6 + r], v); return; }
			// a row without any block of its own between row0 and here (no diagonal block: not a Schur system, but stay correct)
			if (fixed && fixed[(size_t)row * 6 + r]) return;
			atomic_add_f64(y + (size_t)row * 6 + r, v);
			if (dotw) atomic_add_f64(dot + (size_t)pose_seg[row] * dot_stride, dotw[(size_t)row * 6 + r] * v);
		};
		int cur = -1;
		double acc = 0.0;
#pragma unroll
		for (int i = 0; i < SPG_EPG; i++)
		{
			if (key[i] == ~0ull) continue;
			const int row = (int)(key[i] >> 32);
			if (row != cur)
			{
				if (cur >= 0) put(cur, acc);
				cur = row; acc = 0.0;
			}
			acc += sum[i];
		}
		if (cur >= 0) put(cur, acc);
	}
	__syncthreads();
	double w = 0.0;
	int seg = 0;
	bool any = false;
	if (row0 >= 0)
		for (int i = tid; i < SPG_ENT * 6; i += 256)
		{
			const int t = row0 + i / 6;
			const double v = ywin[i];
			if (t < M && v != 0.0 && !(fixed && fixed[(size_t)t * 6 + i % 6]))
			{
				atomic_add_f64(y + (size_t)t * 6 + i % 6, v);
				if (dotw)
				{
					const int sg = pose_seg[t];
					if (any && sg != seg) { atomic_add_f64(dot + (size_t)seg * dot_stride, w); w = 0.0; }
					seg = sg; any = true;
					w = fma(dotw[(size_t)t * 6 + i % 6], v, w);
				}
			}
		}
	if (dotw) wave_scatter_add<1>(dot + (size_t)seg * dot_stride, &w, any);
}

// K11 (Imp.cpp:2980-3020); features of carried maps keep their values
#define BSUB_TILE 128 /* features per work-group */
__global__ void __launch_bounds__(256)
k_backsub(int NF, const int* __restrict__ fptr, const int* __restrict__ photo, const double* __restrict__ W,
          const double* __restrict__ IV, const double* __restrict__ eb, const double* __restrict__ xp,
          const int* __restrict__ feat_seg, const unsigned char* __restrict__ active, double* __restrict__ xf,
          const double* __restrict__ LY, const double* __restrict__ xhat)
{
	// one lane per W block (coalesced): W^T x_p, summed per feature through LDS; then x_f = V^-1 (eb - sum)
	__shared__ int sFp[BSUB_TILE + 1];
	__shared__ double sT[256 * 3];
	__shared__ double sS[BSUB_TILE * 3];
	const int f0 = blockIdx.x * BSUB_TILE, nft = min(BSUB_TILE, NF - f0);
	for (int i = threadIdx.x; i <= nft; i += blockDim.x) sFp[i] = fptr[f0 + i];
	for (int i = threadIdx.x; i < nft * 3; i += blockDim.x) sS[i] = 0.0;
	__syncthreads();
	tile_runs<3>(nft, sFp, sT,
		[&](int j, int, double* out) {
			double w[18];
			ld<18>(w, W + (size_t)j * 18);
			const double* a = xp + (size_t)photo[j] * 6;
			double a6[6];
			ld<6>(a6, a);
			if (xhat)
			{
				// (fused right-hand side: the step from the poses' estimates, see below)
				const double* h = xhat + (size_t)photo[j] * 6;
#pragma unroll
				for (int r = 0; r < 6; r++) a6[r] -= h[r];
			}
#pragma unroll
			for (int c = 0; c < 3; c++)
			{
				double sacc = 0.0;
#pragma unroll
				for (int r = 0; r < 6; r++) sacc = fma(w[3 * r + c], a6[r], sacc);
				out[c] = sacc;
			}
		},
		[&](int fl, int q, double sum, bool) { sS[fl * 3 + q] += sum; });
	for (int fl = threadIdx.x; fl < nft; fl += blockDim.x)
	{
		const int f = f0 + fl;
		if (active && !active[feat_seg[f]]) continue;
		if (xhat)
		{
			// fused right-hand side: eb as a whole was never formed (eb here is its V part; the W part is W^T x^_p, x^_p the poses'
			// estimates): x_f = V^-1 (eb + W^T x^_p - W^T x_p) = L (L^T eb - L^T s), s = sum W^T (x_p - x^_p) over the feature's run
			const double* l = LY + (size_t)f * 9;
			const double s0 = sS[fl * 3], s1 = sS[fl * 3 + 1], s2 = sS[fl * 3 + 2];
			const double d0 = l[6] - (l[0] * s0 + l[1] * s1 + l[3] * s2), d1 = l[7] - (l[2] * s1 + l[4] * s2),
			             d2 = l[8] - l[5] * s2;
			xf[(size_t)f * 3] = l[0] * d0; xf[(size_t)f * 3 + 1] = l[1] * d0 + l[2] * d1; xf[(size_t)f * 3 + 2] = l[3] * d0 + l[4] * d1 + l[5] * d2;
			continue;
		}
		const double* iv = IV + (size_t)f * 9;
		const double d[3] = { eb[(size_t)f * 3] - sS[fl * 3], eb[(size_t)f * 3 + 1] - sS[fl * 3 + 1], eb[(size_t)f * 3 + 2] - sS[fl * 3 + 2] };
		for (int r = 0; r < 3; r++) xf[(size_t)f * 3 + r] = iv[3 * r] * d[0] + iv[3 * r + 1] * d[1] + iv[3 * r + 2] * d[2];
	}
}

// -------------------------------------------------------------------------------------------------------------
__global__ void k_low_words(int n, const unsigned long long* __restrict__ keys, int* __restrict__ out)
{
	int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) out[i] = (int)(keys[i] & 0xffffffffull);
}

void build_spmv_index(lsfm_context* ctx, SchurSystem& sy, const unsigned long long* sorted_upper, int* d_flags, int)
{
	hipStream_t s = ctx->stream;
	Arena& sc = ctx->scratch;
	const int M = sy.M, cnt = sy.nnzb;
	(void)d_flags;
	if (!sy.rowptr)
	{
		sy.rowptr = sc.alloc<int>(M + 1);
		hipLaunchKernelGGL(k_rowptr_from_keys, dim3((M + 1 + 255) / 256), dim3(256), 0, s, M, cnt, sorted_upper, 32, sy.rowptr);
	}
	if (!sy.colidx)
	{
		sy.colidx = sc.alloc<int>(cnt + 1);
		if (cnt) hipLaunchKernelGGL(k_low_words, dim3((cnt + 255) / 256), dim3(256), 0, s, cnt, sorted_upper, sy.colidx);
	}
	// the only extra index: the rows with more than SP_LONG blocks (hub poses)
	int* nl = sc.alloc<int>(1);
	sy.longrows = sc.alloc<int>(M + 1);
	dev_zero(ctx, nl, sizeof(int));
	if (M) hipLaunchKernelGGL(k_spmv_long_rows, dim3((M + 255) / 256), dim3(256), 0, s, M, sy.rowptr, nl, sy.longrows);
	sy.d_nlong = nl;
	// a matrix that stays in the caches: the list of both orientations sorted by target row (k_spmv_gather)
	sy.gent = nullptr; sy.goth = nullptr;
	const int variant = ctx->pcg.spmv_variant;
	if (cnt && variant != 1 && (variant == 2 || (size_t)cnt * 288 <= SPMV_CACHED_BYTES))
	{
		unsigned long long* ent = sc.alloc<unsigned long long>((size_t)2 * cnt);
		int* oth = sc.alloc<int>((size_t)2 * cnt);
		hipLaunchKernelGGL(k_spmv_gather_keys, dim3((cnt + 255) / 256), dim3(256), 0, s, cnt, sorted_upper, ent, oth);
		// by target row only (the high word; the sort is stable and the product is a segmented sum over rows: the order inside a
		// row is free); the holes (all bits set) end up last
		int rb = 1;
		while ((1 << rb) <= M) rb++;
		dev_sort_pairs_u64(ctx, ent, oth, (size_t)2 * cnt, 32 + rb, 32);
		sy.gent = ent; sy.goth = oth;
	}
}

// K7: V^-1 of every feature (values: once per run)
void schur_vinv(lsfm_context* ctx, const SolveIO& io, SchurSystem& sy)
{
	sy.IV = ctx->scratch.alloc<double>((size_t)io.NF * 9);
	sy.LY = ctx->scratch.alloc<double>((size_t)io.NF * 9);
	sy.ymax = ctx->scratch.alloc<double>((size_t)(io.NF + 255) / 256 + 1);
	sy.uu = nullptr;
	if (io.rhs) sy.uu = ctx->scratch.alloc<double>((size_t)io.NF * 6);
	if (io.NF) hipLaunchKernelGGL(k_vinv, dim3((io.NF + 255) / 256), dim3(256), 0, ctx->stream, io.NF, io.V, io.eb, sy.IV, sy.LY, sy.ymax, io.rhs ? *io.rhs : RhsFused(), sy.uu);
}

// ---- Pattern of S (hash of pose pairs, sorted key list, block CSR, SpMV index): depends on the index structure only ----
// In three steps so that the pair insertion can come from the joint map (build_schur_pattern) or, earlier, from what the
// joint map is made of (schur_pattern_early_issue): table set-up, [inserts], compaction (all enqueued); then -- after the
// one read-back of the count -- sort, block CSR, SpMV index.
struct PatternBuild {
	unsigned long long *tab = nullptr, *list = nullptr, *d_k2 = nullptr;
	int *hval = nullptr, *d_flags = nullptr; // [0] overflow, [1] count
	size_t cap = 0;
};
static size_t pattern_capacity(size_t NU, size_t M)
{
	// S has little more than U's pattern (the W-induced pairs are mostly hub links that U already holds): 4x head room over
	// NU + 8 M entries; a table that overflows is rebuilt larger
	size_t cap = 1024;
	while (cap < 4 * (NU + 8 * M + 64)) cap <<= 1;
	return cap;
}
static void pattern_begin(lsfm_context* ctx, size_t cap, PatternBuild& pb)
{
	Arena& sc = ctx->scratch;
	pb.cap = cap;
	pb.tab = sc.alloc<unsigned long long>(cap);
	pb.hval = sc.alloc<int>(cap);
	pb.list = sc.alloc<unsigned long long>(cap);
	pb.d_flags = sc.alloc<int>(4);
	dev_zero(ctx, pb.d_flags, 4 * sizeof(int));
	hipLaunchKernelGGL(k_fill_u64, dim3((unsigned)((cap + 255) / 256)), dim3(256), 0, ctx->stream, pb.tab, cap, HEMPTY);
}
static void pattern_compact(lsfm_context* ctx, PatternBuild& pb)
{
	hipLaunchKernelGGL(k_pat_compact, dim3((unsigned)((pb.cap + 255) / 256)), dim3(256), 0, ctx->stream, pb.cap, pb.tab, pb.list, pb.d_flags + 1);
}
// reads the count back (synchronises ctx->stream); false: the table overflowed or is more than half full
static bool pattern_count(lsfm_context* ctx, const PatternBuild& pb, int* cnt)
{
	int fl[2];
	d2h_ints(ctx, pb.d_flags, fl, 2);
	*cnt = fl[1];
	return !fl[0] && (size_t)fl[1] * 2 <= pb.cap;
}
// sorted key list + block CSR: all that the host's symbolic analysis and K9 wait for (enqueued, nothing read back)
static void pattern_finish(lsfm_context* ctx, const SolveIO& io, const PatternBuild& pb, int cnt, SchurSystem& sy)
{
	hipStream_t s = ctx->stream;
	Arena& sc = ctx->scratch;
	const int M = io.M;
	sy.M = M;
	sy.nnzb = cnt;
	const unsigned long long mask = (unsigned long long)(pb.cap - 1);
	// keys are (row << 32 | column) with both below M: two stable sorts over the bits in use (columns, then rows) instead of
	// one over all 64 -- a third of the passes
	int rb = 1;
	while ((1 << rb) <= M) rb++;
	dev_sort_keys_u64(ctx, pb.list, cnt, 0, rb);
	dev_sort_keys_u64(ctx, pb.list, cnt, 32, 32 + rb);
	sy.rowptr = sc.alloc<int>(M + 1);
	sy.colidx = sc.alloc<int>(cnt + 1);
	if (cnt) hipLaunchKernelGGL(k_pat_assign, dim3((cnt + 255) / 256), dim3(256), 0, s, cnt, pb.list, pb.tab, pb.hval, mask, sy.colidx);
	hipLaunchKernelGGL(k_rowptr_from_keys, dim3((M + 1 + 255) / 256), dim3(256), 0, s, M, cnt, pb.list, 32, sy.rowptr);
	sy.tab = pb.tab; sy.hval = pb.hval; sy.mask = mask;
	sy.upper_keys = pb.list;
	LSFM_CHECK_HIP(hipGetLastError());
}

__global__ void k_pat_insert_keys(int n, const unsigned long long* __restrict__ keys, unsigned long long* tab, unsigned long long mask, int* overflow);
// the pattern of the level below through the join's pose renumbering (PatternSeed)
__global__ void k_pat_insert_keys_remap(int n, const unsigned long long* __restrict__ keys, const int* __restrict__ pnew, const unsigned char* __restrict__ dropped,
                                        unsigned long long* tab, unsigned long long mask, int* overflow)
{
	int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	const int a = (int)(keys[i] >> 32), b = (int)(keys[i] & 0xffffffffull);
	if (dropped[a] || dropped[b]) return;
	hash_insert(tab, mask, pair_key(pnew[a], pnew[b]), overflow);
}
// ... and the pairs across the two sources of a matched joint feature
__global__ void __launch_bounds__(256)
k_pat_insert_cross_remap(PatternSeed sd, unsigned long long* tab, unsigned long long mask, int* overflow)
{
	// (the wave in step, like k_pat_insert_w: neighbouring features are seen by the same poses -- when every lane asks for the same
	// pair, one lane inserts it)
	const int nf = blockIdx.x * blockDim.x + threadIdx.x;
	int jE = 0, lenE = 0, jC = 0, lenC = 0;
	if (nf < sd.NFY)
	{
		const int fe = sd.srcE[nf], fc = sd.srcC[nf];
		if (fe >= 0 && fc >= 0) { jE = sd.fptr_in[fe]; lenE = sd.fptr_in[fe + 1] - jE; jC = sd.fptr_in[fc]; lenC = sd.fptr_in[fc + 1] - jC; }
	}
	int maxE = lenE, maxC = lenC;
#pragma unroll
	for (int off = 32; off > 0; off >>= 1) { maxE = max(maxE, __shfl_xor(maxE, off, LSFM_WAVE)); maxC = max(maxC, __shfl_xor(maxC, off, LSFM_WAVE)); }
	const int lane = threadIdx.x & (LSFM_WAVE - 1);
	for (int a = 0; a < maxE; a++)
	{
		int na = -1;
		if (a < lenE) { const int pa = sd.photo_in[jE + a]; if (!sd.dropped[pa]) na = sd.pnew[pa]; }
		for (int b = 0; b < maxC; b++)
		{
			int nb = -1;
			if (na >= 0 && b < lenC) { const int pb = sd.photo_in[jC + b]; if (!sd.dropped[pb]) nb = sd.pnew[pb]; }
			const bool v = nb >= 0;
			const unsigned long long key = v ? pair_key(na, nb) : 0ull;
			const unsigned long long m = __ballot(v);
			if (m == 0ull) continue;
			const int leader = __ffsll((long long)m) - 1;
			const unsigned long long first = (unsigned long long)__shfl((long long)key, leader, LSFM_WAVE);
			const bool uniform = __ballot(v && key != first) == 0ull;
			if (uniform) { if (lane == leader) hash_insert(tab, mask, key, overflow); }
			else if (v) hash_insert(tab, mask, key, overflow);
		}
	}
}
void build_schur_pattern(lsfm_context* ctx, const SolveIO& io, SchurSystem& sy)
{
	hipStream_t s = ctx->stream;
	Arena& sc = ctx->scratch;
	const int M = io.M, NF = io.NF;
	size_t cap = pattern_capacity(io.NU, M);
	for (int attempt = 0;; attempt++)
	{
		const size_t mk = sc.mark();
		PatternBuild pb;
		pattern_begin(ctx, cap, pb);
		const int nu = std::max(io.NU, M);
		if (nu) hipLaunchKernelGGL(k_pat_insert_u, dim3((nu + 255) / 256), dim3(256), 0, s, io.NU, M, io.Ui, io.Uj, pb.tab, (unsigned long long)(cap - 1), pb.d_flags);
		const PatternSeed* sd = (io.seed && io.seed->prev_keys && !ctx->comm) ? io.seed : nullptr;
		if (sd)
		{
			// (until round 5 every Mono level that analyses hashed every pose pair of every feature again: 66 GB and 28 ms of a
			// synth-16k tree, profiles/r04_pmc_traffic_summary_synth16k.json)
			if (sd->prev_nnzb) hipLaunchKernelGGL(k_pat_insert_keys_remap, dim3((sd->prev_nnzb + 255) / 256), dim3(256), 0, s, sd->prev_nnzb, sd->prev_keys, sd->pnew, sd->dropped, pb.tab, (unsigned long long)(cap - 1), pb.d_flags);
			if (sd->NFY) hipLaunchKernelGGL(k_pat_insert_cross_remap, dim3((sd->NFY + 255) / 256), dim3(256), 0, s, *sd, pb.tab, (unsigned long long)(cap - 1), pb.d_flags);
		}
		else if (NF) hipLaunchKernelGGL(k_pat_insert_w, dim3((NF + 255) / 256), dim3(256), 0, s, NF, io.fptr, io.photo, pb.tab, (unsigned long long)(cap - 1), pb.d_flags);
		pattern_compact(ctx, pb);
		int cnt = 0;
		bool ok = pattern_count(ctx, pb, &cnt);
		if (ctx->comm)
		{
			// feature-sharded run: the pattern of S is the UNION of what the ranks' slices induce.  Every rank learns every rank's
			// count (a vector with one slot per rank, summed), then every rank's keys (each writes its list at its offset of a
			// zeroed array, summed as integers), inserts them all and compacts again.  A rank whose table overflowed says so in
			// the count exchange and all of them start over with a larger table together.
			Comm& cm = *ctx->comm;
			cm.restart();
			long long* d_counts = cm.alloc<long long>(cm.world + 1);
			std::vector<long long> hc(cm.world + 1, 0);
			hc[cm.rank] = ok ? cnt : 0;
			hc[cm.world] = ok ? 0 : 1;
			h2d(ctx, d_counts, hc.data(), sizeof(long long) * hc.size());
			cm.allreduce(s, d_counts, hc.size(), LSFM_DTYPE_I64);
			d2h(ctx, hc.data(), d_counts, sizeof(long long) * hc.size());
			ok = hc[cm.world] == 0;
			if (ok)
			{
				long long total = 0, mine = 0;
				for (int r = 0; r < cm.world; r++) { if (r == cm.rank) mine = total; total += hc[r]; }
				if ((size_t)total * 2 > cap) ok = false; // (the same answer on every rank)
				else
				{
					unsigned long long* all = cm.alloc<unsigned long long>((size_t)total + 1);
					fill_async(s, all, 0, sizeof(unsigned long long) * (size_t)total);
					if (cnt) LSFM_CHECK_HIP(hipMemcpyAsync(all + mine, pb.list, sizeof(unsigned long long) * (size_t)cnt, hipMemcpyDeviceToDevice, s));
					cm.allreduce(s, all, (size_t)total, LSFM_DTYPE_I64);
					if (total) hipLaunchKernelGGL(k_pat_insert_keys, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, (int)total, all, pb.tab, (unsigned long long)(cap - 1), pb.d_flags);
					dev_zero(ctx, pb.d_flags + 1, sizeof(int));
					pattern_compact(ctx, pb);
					ok = pattern_count(ctx, pb, &cnt); // replicated from here on: the same table contents on every rank
				}
			}
		}
		if (ok)
		{
			pattern_finish(ctx, io, pb, cnt, sy);
			if (sd && getenv("LSFM_CHECK_MONO_SEED"))
			{
				// debug / test: the pattern built from the finished joint map must be contained in the seeded one (a pair the seeded
				// pattern lacked would lose its share of S without a word) -- and, the seed being exact, equal to it
				LSFM_CHECK_HIP(hipDeviceSynchronize());
				SolveIO plain = io;
				plain.seed = nullptr;
				SchurSystem ref;
				build_schur_pattern(ctx, plain, ref);
				std::vector<unsigned long long> a(sy.nnzb), b(ref.nnzb);
				d2h(ctx, a.data(), sy.upper_keys, a.size() * sizeof(unsigned long long));
				d2h(ctx, b.data(), ref.upper_keys, b.size() * sizeof(unsigned long long));
				size_t missing = 0, ia = 0;
				for (unsigned long long k : b)
				{
					while (ia < a.size() && a[ia] < k) ia++;
					if (ia == a.size() || a[ia] != k) missing++;
				}
				if (missing || a.size() != b.size())
					LSFM_FAIL(LSFM_ERR_INTERNAL, "seeded pattern of S (" + std::to_string(a.size()) + " blocks) against the joint map's (" + std::to_string(b.size()) + "): " +
					                                 std::to_string(missing) + " pair(s) missing");
			}
			build_spmv_index(ctx, sy, pb.list, pb.d_flags);
			return;
		}
		sc.release(mk);
		cap <<= 2;
		if (attempt > 10) LSFM_FAIL(LSFM_ERR_INTERNAL, "Schur pattern hash table kept overflowing");
	}
}

// ---- the same pattern, earlier ------------------------------------------------------------------------------------------
// The joint map of a Stereo join is the two input maps side by side: a joint feature is seen by the poses of its End source,
// the poses of its Cur source, and the hub pose of either map that is transformed on the way (the transform gives every feature
// of the map a block to the hub pose, Imp.cpp:1303-1309, and folds old blocks to it into that one); U' holds U's pairs and
// (k, hub) for every pose k of a transformed map (Imp.cpp:711-723).  All of that is known from the level's INPUT index arrays
// once the features are matched -- before the transform's block kernels have run.
__global__ void k_pat_insert_u_early(int NU, int M, const int* __restrict__ Ui, const int* __restrict__ Uj, const int* __restrict__ pose_map,
                                     const int* __restrict__ hub, unsigned long long* tab, unsigned long long mask, int* overflow)
{
	int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i < NU) hash_insert(tab, mask, pair_key(Ui[i], Uj[i]), overflow);
	if (i < M)
	{
		hash_insert(tab, mask, pair_key(i, i), overflow);
		const int h = hub[pose_map[i]];
		if (h >= 0) hash_insert(tab, mask, pair_key(i, h), overflow);
	}
}
__global__ void __launch_bounds__(256)
k_pat_insert_w_early(int NFY, const int* __restrict__ srcE, const int* __restrict__ srcC, const int* __restrict__ fptr, const int* __restrict__ photo,
                     const int* __restrict__ feat_map, const int* __restrict__ hub, unsigned long long* tab, unsigned long long mask, int* overflow)
{
	const int nf = blockIdx.x * blockDim.x + threadIdx.x;
	const bool inb = nf < NFY;
	int jE = 0, lenE = 0, hE = -1, jC = 0, lenC = 0, hC = -1;
	if (inb)
	{
		const int fe = srcE[nf], fc = srcC[nf];
		if (fe >= 0) { jE = fptr[fe]; lenE = fptr[fe + 1] - jE; hE = hub[feat_map[fe]]; }
		if (fc >= 0) { jC = fptr[fc]; lenC = fptr[fc + 1] - jC; hC = hub[feat_map[fc]]; }
	}
	const int nE = lenE + (hE >= 0 ? 1 : 0), len = nE + lenC + (hC >= 0 ? 1 : 0);
	auto pose_at = [&](int i) -> int {
		if (i < lenE) return photo[jE + i];
		if (i < nE) return hE;
		i -= nE;
		return i < lenC ? photo[jC + i] : hC;
	};
	int maxlen = len;
#pragma unroll
	for (int off = 32; off > 0; off >>= 1) maxlen = max(maxlen, __shfl_xor(maxlen, off, LSFM_WAVE));
	const int lane = threadIdx.x & (LSFM_WAVE - 1);
	for (int a = 0; a < maxlen; a++)
	{
		const int pa = (inb && a < len) ? pose_at(a) : -1;
		for (int b = a + 1; b < maxlen; b++)
		{
			const bool v = inb && b < len;
			unsigned long long key = v ? pair_key(pa, pose_at(b)) : 0ull;
			unsigned long long m = __ballot(v);
			if (m == 0ull) continue;
			int leader = __ffsll((long long)m) - 1;
			unsigned long long first = (unsigned long long)__shfl((long long)key, leader, LSFM_WAVE);
			bool uniform = __ballot(v && key != first) == 0ull;
			if (uniform) { if (lane == leader) hash_insert(tab, mask, key, overflow); }
			else if (v) hash_insert(tab, mask, key, overflow);
		}
	}
}

// the pairs across the two sources of a matched joint feature (everything else of the level's pattern is in the pattern of the
// level below or a hub link): (poses of End's run + End's hub) x (poses of Cur's run + Cur's hub)
__global__ void __launch_bounds__(256)
k_pat_insert_w_cross(int NFY, const int* __restrict__ srcE, const int* __restrict__ srcC, const int* __restrict__ fptr, const int* __restrict__ photo,
                     const int* __restrict__ feat_map, const int* __restrict__ hub, unsigned long long* tab, unsigned long long mask, int* overflow)
{
	const int nf = blockIdx.x * blockDim.x + threadIdx.x;
	if (nf >= NFY) return;
	const int fe = srcE[nf], fc = srcC[nf];
	if (fe < 0 || fc < 0) return;
	const int jE = fptr[fe], lenE = fptr[fe + 1] - jE, hE = hub[feat_map[fe]];
	const int jC = fptr[fc], lenC = fptr[fc + 1] - jC, hC = hub[feat_map[fc]];
	for (int a = 0; a <= lenE; a++)
	{
		const int pa = a < lenE ? photo[jE + a] : hE;
		if (pa < 0) continue;
		for (int b = 0; b <= lenC; b++)
		{
			const int pb = b < lenC ? photo[jC + b] : hC;
			if (pb >= 0) hash_insert(tab, mask, pair_key(pa, pb), overflow);
		}
	}
}
__global__ void k_pat_insert_keys(int n, const unsigned long long* __restrict__ keys, unsigned long long* tab, unsigned long long mask, int* overflow)
{
	int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) hash_insert(tab, mask, keys[i], overflow);
}

// ---- the pattern of the NEXT level's system, from this level's joint maps (prefetch_next_level, lsfm_pcg.hip) -------------
// hub pose of every map of the batch in the next level's transform: the pose whose id is the map's target reference
__global__ void k_pre_hubs(int M, const int* __restrict__ pose_id, const int* __restrict__ pose_map, const int* __restrict__ tref, int* __restrict__ hub)
{
	int k = blockIdx.x * blockDim.x + threadIdx.x;
	if (k >= M) return;
	const int b = pose_map[k];
	if (tref[b] >= 0 && pose_id[k] == tref[b]) hub[b] = k;
}
// pairs across the two maps of a pair, one lane per feature of the second map that has a match in the first
__global__ void __launch_bounds__(256)
k_pat_insert_w_cross_match(int NF, const int* __restrict__ feat_map, const int* __restrict__ match, const int* __restrict__ fptr,
                           const int* __restrict__ photo, const int* __restrict__ hub, unsigned long long* tab, unsigned long long mask, int* overflow)
{
	const int fc = blockIdx.x * blockDim.x + threadIdx.x;
	if (fc >= NF || !(feat_map[fc] & 1)) return;
	const int fe = match[fc];
	if (fe < 0) return;
	const int jE = fptr[fe], lenE = fptr[fe + 1] - jE, hE = hub[feat_map[fe]];
	const int jC = fptr[fc], lenC = fptr[fc + 1] - jC, hC = hub[feat_map[fc]];
	for (int a = 0; a <= lenE; a++)
	{
		const int pa = a < lenE ? photo[jE + a] : hE;
		if (pa < 0) continue;
		for (int b = 0; b <= lenC; b++)
		{
			const int pb = b < lenC ? photo[jC + b] : hC;
			if (pb >= 0) hash_insert(tab, mask, pair_key(pa, pb), overflow);
		}
	}
}
// which blocks of the batch survive the next level's transform as they are (k_tr_flags of lsfm_transform.hip, from the hubs alone)
__global__ void k_pre_flags(const int* __restrict__ Ui, const int* __restrict__ Uj, int NU, const int* __restrict__ photo, int NW,
                            const int* __restrict__ pose_map, const int* __restrict__ hub, int* __restrict__ keepU, int* __restrict__ keepW)
{
	int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i < NU) { const int a = Ui[i], b = Uj[i], h = hub[pose_map[a]]; keepU[i] = (h < 0 || (a != h && b != h)) ? 1 : 0; }
	if (i < NW) { const int k = photo[i], h = hub[pose_map[k]]; keepW[i] = (h < 0 || k != h) ? 1 : 0; }
	if (i == 0) { keepU[NU] = 0; keepW[NW] = 0; }
}
__global__ void k_pre_gather(const int* __restrict__ KU, const int* __restrict__ KW, const int* __restrict__ R, const int* __restrict__ uoff,
                             const int* __restrict__ woff, const int* __restrict__ foff, int B, int* __restrict__ out)
{
	int b = blockIdx.x * blockDim.x + threadIdx.x;
	if (b > B) return;
	out[b] = KU[uoff[b]];
	out[B + 1 + b] = KW[woff[b]];
	out[2 * (B + 1) + b] = R[foff[b]];
}
// ctx->stream / ctx->scratch name the stream and the arena the caller wants this on.  prev_keys: the pattern of the level that
// produced Y (every pair inside one of Y's maps).  false: nothing to build from.  counts (optional): what the next level's
// transform and join read back from the device -- kept-block prefixes at the map boundaries (U, then W: transform_batch's
// `cnt`) and the ranks of the unmatched features there (join_stereo_prepare's `rb`), 3 (B + 1) ints, valid after the
// caller's next synchronisation of the stream.
bool schur_pattern_prefetch(lsfm_context* ctx, const DevBatch& Y, const int* d_tref, const unsigned long long* prev_keys, int prev_nnzb, SchurSystem& sy,
                            std::vector<int>* counts, bool want_pattern, LevelIndex* keep)
{
	// prev_keys == null: the level that produced Y left no pattern (its systems were small enough for the dense path, which needs
	// none): the pairs inside every map of Y are then taken from Y's own W runs (k_pat_insert_w), as a level without a predecessor does
	if (!Y.M) return false;
	hipStream_t s = ctx->stream;
	Arena& sc = ctx->scratch;
	static const bool dbg = getenv("LSFM_DEBUG_SYNC") != nullptr;
	auto chk = [&](const char* what) {
		if (!dbg) return;
		hipError_t e = hipDeviceSynchronize();
		fprintf(stderr, "[prefetch] %s: %s\n", what, hipGetErrorString(e));
	};
	chk("before");
	int* hub = sc.alloc<int>(Y.B);
	fill_async(s, hub, 0xff, sizeof(int) * (size_t)Y.B);
	hipLaunchKernelGGL(k_pre_hubs, dim3((Y.M + 255) / 256), dim3(256), 0, s, Y.M, Y.pose_id, Y.pose_map, d_tref, hub);
	chk("hubs");
	int* match = sc.alloc<int>(Y.NF + 1);
	int* unm = sc.alloc<int>(Y.NF + 2);
	if (Y.NF) join_match_features(ctx, Y, match, unm);
	else dev_zero(ctx, unm, 2 * sizeof(int));
	chk("match");
	if (counts)
	{
		const int B = Y.B;
		int* keepU = sc.alloc<int>(Y.NU + 1); int* keepW = sc.alloc<int>(Y.NW + 1);
		int* KU = sc.alloc<int>(Y.NU + 2); int* KW = sc.alloc<int>(Y.NW + 2); int* R = sc.alloc<int>(Y.NF + 2);
		const int nmax = std::max(std::max(Y.NU, Y.NW), 1);
		hipLaunchKernelGGL(k_pre_flags, dim3((nmax + 255) / 256), dim3(256), 0, s, Y.Ui, Y.Uj, Y.NU, Y.photo, Y.NW, Y.pose_map, hub, keepU, keepW);
		dev_exclusive_scan(ctx, keepU, KU, Y.NU);
		dev_exclusive_scan(ctx, keepW, KW, Y.NW);
		dev_exclusive_scan(ctx, unm, R, Y.NF);
		int* d_off = sc.alloc<int>(2 * (B + 1));
		h2d(ctx, d_off, Y.u_off.data(), (B + 1) * sizeof(int));
		h2d(ctx, d_off + B + 1, Y.w_off.data(), (B + 1) * sizeof(int));
		int* d_cnt = sc.alloc<int>(3 * (B + 1));
		hipLaunchKernelGGL(k_pre_gather, dim3((B + 1 + 127) / 128), dim3(128), 0, s, KU, KW, R, d_off, d_off + B + 1, Y.d_feat_off, B, d_cnt);
		counts->resize(3 * (size_t)(B + 1));
		LSFM_CHECK_HIP(hipMemcpyAsync(counts->data(), d_cnt, counts->size() * sizeof(int), hipMemcpyDeviceToHost, s));
		if (keep)
		{
			// what the next level's transform and join would work out again from the same index arrays (they live in this arena until
			// the level after next prepares ITS successor)
			*keep = LevelIndex();
			keep->NU = Y.NU; keep->NW = Y.NW; keep->NF = Y.NF;
			keep->KU = KU; keep->KW = KW; keep->match = match; keep->R = R;
		}
	}
	if (!want_pattern)
	{
		// (the next level's systems are small: it needs the counts alone; they arrive with the caller's next synchronisation)
		LSFM_CHECK_HIP(hipStreamSynchronize(s));
		return true;
	}
	size_t cap = pattern_capacity(std::max((size_t)Y.NU + Y.M, (size_t)prev_nnzb + Y.M), Y.M);
	SolveIO io;
	io.M = Y.M;
	for (int attempt = 0;; attempt++)
	{
		const size_t mk = sc.mark();
		PatternBuild pb;
		pattern_begin(ctx, cap, pb);
		const unsigned long long mask = (unsigned long long)(cap - 1);
		const int nu = std::max(Y.NU, Y.M);
		chk("begin");
		hipLaunchKernelGGL(k_pat_insert_u_early, dim3((nu + 255) / 256), dim3(256), 0, s, Y.NU, Y.M, Y.Ui, Y.Uj, Y.pose_map, hub, pb.tab, mask, pb.d_flags);
		chk("insert_u");
		if (prev_keys && prev_nnzb) hipLaunchKernelGGL(k_pat_insert_keys, dim3((prev_nnzb + 255) / 256), dim3(256), 0, s, prev_nnzb, prev_keys, pb.tab, mask, pb.d_flags);
		if (!prev_keys && Y.NF) hipLaunchKernelGGL(k_pat_insert_w, dim3((Y.NF + 255) / 256), dim3(256), 0, s, Y.NF, Y.fptr, Y.photo, pb.tab, mask, pb.d_flags);
		chk("insert_keys");
		if (Y.NF) hipLaunchKernelGGL(k_pat_insert_w_cross_match, dim3((Y.NF + 255) / 256), dim3(256), 0, s, Y.NF, Y.feat_map, match, Y.fptr, Y.photo, hub, pb.tab, mask, pb.d_flags);
		chk("insert_cross");
		pattern_compact(ctx, pb);
		chk("compact");
		int cnt = 0;
		if (pattern_count(ctx, pb, &cnt))
		{
			pattern_finish(ctx, io, pb, cnt, sy);
			build_spmv_index(ctx, sy, pb.list, pb.d_flags);
			return true;
		}
		sc.release(mk);
		cap <<= 2;
		if (attempt > 10) return false;
	}
}

struct EarlyPattern { PatternBuild pb; int M = 0; };

void schur_pattern_early_issue(lsfm_context* ctx, const EarlyPatternIn& in)
{
	ctx->early.reset();
	auto ep = std::make_shared<EarlyPattern>();
	ep->M = in.M;
	// on the side stream, behind the point of the main stream where the matches and the hub poses are known (evC)
	LSFM_CHECK_HIP(hipStreamWaitEvent(ctx->stream3, ctx->evC, 0));
	std::swap(ctx->stream, ctx->stream3);
	try
	{
		hipStream_t s = ctx->stream;
		const size_t cap = pattern_capacity(std::max((size_t)in.NU + in.M, (size_t)in.prev_nnzb + in.M), in.M);
		pattern_begin(ctx, cap, ep->pb);
		const int nu = std::max(in.NU, in.M);
		const unsigned long long mask = (unsigned long long)(cap - 1);
		if (nu) hipLaunchKernelGGL(k_pat_insert_u_early, dim3((nu + 255) / 256), dim3(256), 0, s, in.NU, in.M, in.Ui, in.Uj, in.pose_map, in.hub, ep->pb.tab, mask, ep->pb.d_flags);
		if (in.prev_keys)
		{
			// the level below left its pattern: every pair inside one source map is in it; what is new are the pairs across
			if (in.prev_nnzb) hipLaunchKernelGGL(k_pat_insert_keys, dim3((in.prev_nnzb + 255) / 256), dim3(256), 0, s, in.prev_nnzb, in.prev_keys, ep->pb.tab, mask, ep->pb.d_flags);
			if (in.NFY) hipLaunchKernelGGL(k_pat_insert_w_cross, dim3((in.NFY + 255) / 256), dim3(256), 0, s, in.NFY, in.srcE, in.srcC, in.fptr, in.photo, in.feat_map, in.hub, ep->pb.tab, mask, ep->pb.d_flags);
		}
		else if (in.NFY) hipLaunchKernelGGL(k_pat_insert_w_early, dim3((in.NFY + 255) / 256), dim3(256), 0, s, in.NFY, in.srcE, in.srcC, in.fptr, in.photo, in.feat_map, in.hub, ep->pb.tab, mask, ep->pb.d_flags);
		pattern_compact(ctx, ep->pb);
		LSFM_CHECK_HIP(hipGetLastError());
	}
	catch (...) { std::swap(ctx->stream, ctx->stream3); throw; }
	std::swap(ctx->stream, ctx->stream3);
	ctx->early = ep;
}
void schur_pattern_early_drop(lsfm_context* ctx) { ctx->early.reset(); }

// Second half, on the stream ctx->stream currently names (the caller has swapped the side stream in): count, sort, block CSR.
// false: no early build in flight, or its table overflowed -- the caller builds the pattern from the joint map instead.
bool schur_pattern_early_finish(lsfm_context* ctx, const SolveIO& io, SchurSystem& sy)
{
	std::shared_ptr<void> keep = ctx->early;
	ctx->early.reset();
	EarlyPattern* ep = static_cast<EarlyPattern*>(keep.get());
	if (!ep || ep->M != io.M) return false;
	int cnt = 0;
	if (!pattern_count(ctx, ep->pb, &cnt)) return false;
	pattern_finish(ctx, io, ep->pb, cnt, sy);
	return true;
}
// after the keys have gone to the host: -- enqueued only -- the SpMV index, which nothing needs before the first product of the CG
void schur_pattern_early_extras(lsfm_context* ctx, const SolveIO& io, SchurSystem& sy)
{
	(void)io;
	build_spmv_index(ctx, sy, sy.upper_keys, nullptr);
}

void schur_pattern_only(lsfm_context* ctx, const SolveIO& io, int* nnzb, const int** rowptr, const int** colidx)
{
	SchurSystem sy;
	build_schur_pattern(ctx, io, sy);
	*nnzb = sy.nnzb; *rowptr = sy.rowptr; *colidx = sy.colidx;
}

// Values of S and E: enqueued without a host synchronisation (the caller overlaps host work with K9); the K9 launch is
// bracketed by the events ev2/ev3 of the context, read back by schur_values_stats() after the caller's next sync
void build_schur_values(lsfm_context* ctx, const SolveIO& io, SchurSystem& sy)
{
	hipStream_t s = ctx->stream;
	Arena& sc = ctx->scratch;
	const int M = io.M, NF = io.NF, cnt = sy.nnzb;
	const unsigned long long* tab = sy.tab;
	const int* hval = sy.hval;
	const unsigned long long mask = sy.mask;
	const int ntiles = (NF + SCHUR_TILE - 1) / SCHUR_TILE;
	unsigned char* fb;
	// K9 adds in fixed point (order-independent, lsfm_device.hpp): the accumulators of W V^-1 W^T, the two limbs of -W V^-1 eb and the
	// poison word, back to back; S starts as U, E as the pose part of the right-hand side, k_schur_finish takes the sums off them
	const size_t nacc = 1 + (size_t)cnt * 36 + (size_t)M * 12;
	if (ctx->comm)
	{
		// feature-sharded run: the accumulators live in the caller's buffer; those of S (and the poison word ahead of them) are
		// summed over the ranks as INTEGERS -- exact, so every rank holds the same bits of S (U is replicated: every rank starts S
		// from it).  The right-hand side's limbs stay this rank's own -- their scale hangs on the rank's features (k_schur_scale: ey)
		// -- and E, this rank's part of it (its features' share of the pose part, rhs kernels of the join, U's on rank 0; its
		// features' -W V^-1 eb), is summed as doubles once k_schur_finish has put it together.
		ctx->comm->restart();
		sy.acc = ctx->comm->alloc<long long>(nacc);
		sy.E = ctx->comm->alloc<double>((size_t)M * 6);
		fill_async(s, sy.acc, 0, nacc * sizeof(long long));
		ZeroSpan zs(sc);
		sy.S = sc.alloc<double>((size_t)cnt * 36);
		fb = sc.alloc<unsigned char>(ntiles + 1);
		zs.zero(s);
	}
	else
	{
		ZeroSpan zs(sc);
		sy.S = sc.alloc<double>((size_t)cnt * 36);
		fb = sc.alloc<unsigned char>(ntiles + 1); // tiles the panel kernel hands to the per-feature kernel
		sy.acc = sc.alloc<long long>(nacc);
		zs.zero(s);
		sy.E = sc.alloc<double>((size_t)M * 6);
	}
	sy.sexp = sc.alloc<int>((size_t)M * 6 + 1);
	if (io.NU) hipLaunchKernelGGL(k_schur_u, dim3((io.NU + 255) / 256), dim3(256), 0, s, io.NU, io.U, io.Ui, io.Uj, tab, hval, mask, sy.S);
	hipLaunchKernelGGL(k_schur_scale, dim3((6 * M + 255) / 256 + 1), dim3(256), 0, s, M, sy.rowptr, sy.S, NF, NF ? (NF + 255) / 256 : 0, sy.ymax, io.d_fixed, sy.sexp, io.ea, sy.E);
	K9Out ko;
	ko.poison = sy.acc; ko.S = sy.acc + 1; ko.Ehi = ko.S + (size_t)cnt * 36; ko.Elo = ko.Ehi + (size_t)M * 6;
	ko.sexp = sy.sexp; ko.ey = sy.sexp + (size_t)M * 6;
	if (io.rhs) { ko.xpose = io.rhs->pose_src; ko.pside = io.rhs->pose_map_src; ko.uu = sy.uu; }
	if (NF)
	{
		// bracketed by HIP events on this stream: live duration of the K9 launch for the roofline line of bench.py
		hipEvent_t e2 = nullptr, e3 = nullptr;
		if (ctx->stats) { e2 = ctx->pool_event(); e3 = ctx->pool_event(); LSFM_REC_T(e2, s); }
		static_assert(LSFM_PM_TILE % SCHUR_TILE == 0, "a tile of the panel kernel must be whole tiles of the fallback kernel");
		int most = 0;
		for (int r : io.seg_rows) most = std::max(most, r);
		// the per-tile slots of the panel variants: from the plan of the level, or worked out now (and left for the plan, if this
		// run makes one)
		bool fresh_lists = false;
		if (!sy.k9.ns)
		{
			fresh_lists = true;
			sy.k9.ns = sc.alloc<int>(ntiles + 1);
			sy.k9.pose = sc.alloc<int>((size_t)ntiles * 64 + 1);
			sy.k9.eslot = sc.alloc<unsigned char>((size_t)io.NW + 1);
			sy.k9.wlist = sc.alloc<int>((size_t)3 * ntiles + 1);
			sy.k9.wcnt = sc.alloc<int>(8);
			sy.k9_tiles = ntiles; sy.k9_NW = io.NW;
			launch_schur_slots(ctx, NF, io.fptr, io.photo, fb, sy.k9);
		}
		launch_schur_panel(ctx, NF, io.fptr, io.photo, io.W, sy.LY, tab, hval, mask, ko, fb, most, sy.k9, fresh_lists);
		hipLaunchKernelGGL(k_schur_w, dim3(ntiles), dim3(SCHUR_TILE), 0, s, NF, io.fptr, io.photo, io.W, sy.IV, io.eb, tab, hval, mask, ko, fb, sy.LY);
		if (ctx->stats)
		{
			LSFM_REC_T(e3, s);
			ctx->defer_time(e2, e3, &ctx->stats->schur_ms);
			ctx->stats->schur_launches++;
			// every input once (W block + photo index; V^-1, eb, run pointer per feature), every output once (S, E)
			ctx->stats->schur_bytes += (double)io.NW * (144 + 4) + (double)io.NF * (72 + 24 + 4) + (double)sy.nnzb * 288 + (double)io.M * 48;
			// algorithmic flops of K9 = 144 NW + 108 (sum of squared run lengths + NW): the sum is taken on the device and read with
			// the run's record at the end of a tree run (lsfm_tree_run adds 108 x that)
			ctx->stats->schur_flops += (double)io.NW * (144.0 + 108.0);
			// (the W part of the right-hand sides, when K9 takes it along: eF += W^T x_p and eP += W x_f, 36 multiply-adds a block,
			// and the estimates they read -- until round 5 a pass of its own, k_join_rhs_w)
			if (io.rhs) { ctx->stats->schur_flops += (double)io.NW * 72.0; ctx->stats->schur_bytes += (double)io.NF * (48 + 24) + (double)io.M * 48; }
			// (on the side stream, behind the 16-slot variant and its event: a number for the run's statistics has no place in the
			// chain of the main stream -- 9 us per level; lsfm_tree_run waits for that stream before it reads the record)
			if (ctx->in_tree_run && ctx->d_run)
			{
				LSFM_CHECK_HIP(hipEventRecord(ctx->ev_k9[0], s)); // (the run pointers are final at this point of the main stream)
				LSFM_CHECK_HIP(hipStreamWaitEvent(ctx->stream2, ctx->ev_k9[0], 0));
				hipLaunchKernelGGL(k_sum_run_squares, dim3(std::min((NF + 255) / 256, 512)), dim3(256), 0, ctx->stream2, NF, io.fptr, &ctx->d_run->k2);
			}
		}
	}
	if (ctx->comm) ctx->comm->allreduce(s, sy.acc, 1 + (size_t)cnt * 36, LSFM_DTYPE_I64);
	{
		const size_t n = (size_t)cnt * 36 + (size_t)M * 6;
		hipLaunchKernelGGL(k_schur_finish, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, cnt, M, sy.upper_keys, sy.acc, sy.sexp, sy.S, sy.E);
	}
	if (ctx->comm) ctx->comm->allreduce(s, sy.E, (size_t)M * 6, LSFM_DTYPE_F64);
	LSFM_CHECK_HIP(hipGetLastError());
}

void launch_spmv(lsfm_context* ctx, const SchurSystem& sy, const double* x, double* y, const unsigned char* fixed,
                 const double* dotw, const int* pose_seg, double* dot, int dot_stride)
{
	if (!sy.M) return;
	if (sy.gent)
	{
		// holes (the mirrored slots of the diagonal blocks) sort to the end: 2 nnzb - (diagonal blocks) real entries, at most 2 nnzb - 1
		const int nent = 2 * sy.nnzb;
		hipLaunchKernelGGL(k_spmv_gather, dim3((nent + SPG_ENT - 1) / SPG_ENT), dim3(256), 0, ctx->stream, sy.M, nent, sy.gent, sy.goth, sy.S, x, y, fixed,
		                   dotw, pose_seg, dot, dot_stride);
		return;
	}
	hipLaunchKernelGGL(k_spmv, dim3((sy.M + SPT - 1) / SPT), dim3(256), 0, ctx->stream, sy.M, sy.rowptr, sy.colidx, sy.d_nlong, sy.longrows, sy.S, x, y,
	                   fixed, dotw, pose_seg, dot, dot_stride);
}

// algorithmic bytes of one SpMV on the upper-block storage: blocks + column indices + row pointers + x and y
double spmv_bytes(const SchurSystem& sy) { return (double)sy.nnzb * (288 + 4) + 4.0 * (sy.M + 1) + 2.0 * 48 * sy.M; }

void launch_backsub(lsfm_context* ctx, const SolveIO& io, const SchurSystem& sy, const double* x)
{
	if (io.NF)
		hipLaunchKernelGGL(k_backsub, dim3((io.NF + BSUB_TILE - 1) / BSUB_TILE), dim3(256), 0, ctx->stream, io.NF, io.fptr, io.photo, io.W, sy.IV, io.eb, x,
		                   io.d_feat_seg, io.d_seg_active, io.x_feat, sy.LY, io.rhs ? io.rhs->pose_src : (const double*)nullptr);
}

void vinv_only(lsfm_context* ctx, int NF, const double* V, double* IV)
{
	if (!NF) return;
	const size_t mk = ctx->scratch.mark();
	double* LY = ctx->scratch.alloc<double>((size_t)NF * 9);
	double* eb = ctx->scratch.alloc<double>((size_t)NF * 3);
	dev_zero(ctx, eb, (size_t)NF * 3 * sizeof(double));
	hipLaunchKernelGGL(k_vinv, dim3((NF + 255) / 256), dim3(256), 0, ctx->stream, NF, V, eb, IV, LY, (double*)nullptr, RhsFused(), (double*)nullptr);
	LSFM_CHECK_HIP(hipStreamSynchronize(ctx->stream));
	ctx->scratch.release(mk);
}
void backsub_only(lsfm_context* ctx, int NF, const int* fptr, const int* photo, const double* W, const double* IV, const double* eb, const double* xp, double* xf)
{
	if (NF)
		hipLaunchKernelGGL(k_backsub, dim3((NF + BSUB_TILE - 1) / BSUB_TILE), dim3(256), 0, ctx->stream, NF, fptr, photo, W, IV, eb, xp,
		                   (const int*)nullptr, (const unsigned char*)nullptr, xf, (const double*)nullptr, (const double*)nullptr);
}

// y = S x for an externally supplied symmetric block matrix (upper block CSR): measurement entry of the C ABI
int spmv_external(lsfm_context* ctx, int m, const int* rowptr, const int* colidx, const double* val, const double* x, double* y, int reps,
                  double* avg_ms, double* bytes)
{
	hipStream_t s = ctx->stream;
	Arena& sc = ctx->scratch;
	size_t mk = sc.mark();
	const int nnzb = rowptr[m];
	SchurSystem sy;
	sy.M = m; sy.nnzb = nnzb;
	std::vector<unsigned long long> keys(nnzb);
	for (int p = 0; p < m; p++)
		for (int k = rowptr[p]; k < rowptr[p + 1]; k++) keys[k] = ((unsigned long long)(unsigned)p << 32) | (unsigned)colidx[k];
	unsigned long long* dk = sc.alloc<unsigned long long>(nnzb + 1);
	sy.S = sc.alloc<double>((size_t)nnzb * 36);
	double* dx = sc.alloc<double>((size_t)m * 6);
	double* dy = sc.alloc<double>((size_t)m * 6);
	int* d_flags = sc.alloc<int>(4);
	h2d(ctx, dk, keys.data(), (size_t)nnzb * sizeof(unsigned long long));
	h2d(ctx, sy.S, val, (size_t)nnzb * 36 * sizeof(double));
	h2d(ctx, dx, x, (size_t)m * 6 * sizeof(double));
	int nmir = 0;
	for (int p = 0; p < m; p++)
		for (int k = rowptr[p]; k < rowptr[p + 1]; k++) nmir += colidx[k] != p;
	build_spmv_index(ctx, sy, dk, d_flags, nmir);
	float total = 0;
	for (int k = 0; k < reps + 2; k++)
	{
		dev_zero(ctx, dy, (size_t)m * 6 * sizeof(double));
		LSFM_CHECK_HIP(hipEventRecord(ctx->ev0, s));
		launch_spmv(ctx, sy, dx, dy, nullptr, nullptr, nullptr, nullptr, 1);
		LSFM_CHECK_HIP(hipEventRecord(ctx->ev1, s));
		LSFM_CHECK_HIP(hipEventSynchronize(ctx->ev1));
		float t = 0;
		LSFM_CHECK_HIP(hipEventElapsedTime(&t, ctx->ev0, ctx->ev1));
		if (k >= 2) total += t;
	}
	d2h(ctx, y, dy, (size_t)m * 6 * sizeof(double));
	if (avg_ms) *avg_ms = reps > 0 ? total / reps : 0;
	if (bytes) *bytes = spmv_bytes(sy);
	sc.release(mk);
	return 0;
}

} // namespace lsfm
