// K10: preconditioned conjugate gradients on the Schur-reduced camera system S x = E, all independent systems of one
// tree level iterating together with per-system scalars.  Replaces cholmod_analyze / factorize / solve
// (pba_solveCholmod{LM,GN}, Imp.cpp:2380-2449 / 7043-7121).
//
// Preconditioner.  Block-Jacobi needs O(10 m) iterations on these matrices (a pose chain of length m plus one dense
// "hub" row per join of the tree; measured 84k iterations without convergence at m = 3499), so M is a sparse 6x6-block
// Cholesky factorisation of S itself, made cheap by the structure of the join tree:
//   * ordering: nested dissection along the tree.  The position of a pose in the level's pose array encodes the local
//     map that brought it, so an edge (p,q) of S crosses the cut of tree level bitlen(origin_p ^ origin_q).  The
//     endpoint of higher degree (the hub pose of that sub-map) goes into that level's separator; blocks are
//     eliminated by ascending separator level.  Measured fill 1.5x nnz(S), elimination-tree height ~ 15 per level.
//   * symbolic analysis (elimination tree, column patterns, tasks) on the host from the block pattern (a few hundred
//     KB), while the numeric Schur assembly runs on the device; numeric factorisation and triangular solves on the
//     device.  The elimination tree is ~140 columns high at the top join but only ~8 TASK levels deep: a sub-tree of
//     <= 32 columns, or a separator chain of the dissection, is one task, walked by one work-group; launches go by
//     task level.
// With the exact factor CG is iterative refinement: 2-3 iterations to 1e-12.
#include <algorithm>
#include <chrono>
#include <climits>
#include <cmath>
#include <cstdlib>
#include <numeric>
#include <queue>

#include "lsfm_device.hpp"
#include "lsfm_internal.hpp"
#include "lsfm_solve.hpp"
#include "lsfm_symbolic.hpp"

namespace lsfm {

struct PcgSeg {
	double rz[2];
	double pAp;
	double rr, ee, thresh, rr_prev;
	int done, its, row0, active;
	int slow, pad; // steps in a row that shrank the true residual by less than half
};
static_assert(sizeof(PcgSeg) % sizeof(double) == 0, "PcgSeg is strided in doubles by the fused dot products");
#define SEG_STRIDE ((int)(sizeof(PcgSeg) / sizeof(double)))

// ---------------------------------------------------------------------------------------------------------------
// sparse block Cholesky: device side
// ---------------------------------------------------------------------------------------------------------------
struct CholDev {
	int M = 0, nnzL = 0, nlevels = 0, tail_begin = 0; // columns [tail_begin, M) (in level order) run in one launch
	int* colptr = nullptr;  // [M+1]
	int* rowidx = nullptr;  // [nnzL] ascending inside a column, diagonal first
	int* perm = nullptr;    // [M] new -> old
	int* pinv = nullptr;    // [M] old -> new
	int* order = nullptr;   // [M] columns sorted by elimination-tree level
	std::vector<int> level_ptr; // host: order[level_ptr[l] .. level_ptr[l+1]) = columns of level l (before the tail)
	// tasks: connected pieces of the elimination tree that one work-group walks serially (small sub-trees, chains)
	int* task_cols = nullptr;          // [M] columns grouped by task, ascending inside a task
	int* task_ptr = nullptr;           // [ntasks+1] tasks ordered by task level
	std::vector<int> tlevel_ptr;       // host: tasks of task level l = [tlevel_ptr[l], tlevel_ptr[l+1])
	std::vector<int> tlevel_maxsize;   // host: most columns in a task of the level (LDS of the solve launches)
	int* col_task = nullptr;           // [M] task (position in task_ptr) of a column
	int* col_lpos = nullptr;           // [M] position of a column inside its task
	int* col_nin = nullptr;            // [M] leading rows of a column (below the diagonal) that belong to its own task
	std::vector<int> tlevel_col0;      // host: task_cols[tlevel_col0[l] .. tlevel_col0[l+1]) = columns of the level's tasks
	std::vector<int> tlevel_nsmall;    // host: the first tlevel_nsmall[l] tasks of level l fit LDS whole (small-task kernels)
	std::vector<int> tlevel_small_lds; // host: dynamic LDS bytes of the level's small-task launches
	std::vector<int> tlevel_outer;     // host: largest number of deferred update pairs of a column of the level
	// supernode groups: the columns above the leaf tasks, cut into runs of <= CHOL_GS consecutive columns of one
	// fundamental supernode (same rows below the run), ordered by group level (children before parents)
	int ngroups = 0;
	int *grp_c0 = nullptr, *grp_s = nullptr, *grp_nr = nullptr; // [ngroups] first column, columns, rows below the run
	std::vector<int> glevel_ptr;    // host: groups of level l = [glevel_ptr[l], glevel_ptr[l+1])
	std::vector<int> glevel_maxnr;  // host: most rows below a run of the level
	std::vector<int> glevel_maxs;   // host: most block columns of a run of the level (LDS of k_sn_panel)
	// distributed factorisation (lsfm_symbolic.hpp): owner of every column (-1: shared), null when off; the shared columns are the
	// last ones, from first_shared on (their blocks: from block shared_blk0 of L on)
	int* col_owner = nullptr;
	int first_shared = 0, shared_blk0 = 0;
	double work_total = 0, work_shared = 0;
	std::vector<char> glevel_owned, glevel_shared;
	int* blob = nullptr;    // all index arrays above are slices of this one allocation
	size_t blob_ints = 0;
	float *Lf = nullptr, *Dinvf = nullptr; // mixed precision: the factor rounded to fp32 for the triangular solves (null: fp64)
	double* wv = nullptr;   // [M*6] forward-solve results of the group columns (lsfm_pcg.hip k_sn_fwd / k_sn_bwd)
	double* Lg = nullptr;   // [nnzL*36] the factor of the supernode-group columns (same indexing as L; L keeps their unfactored blocks)
	float* Lgf = nullptr;   // mixed precision: its fp32 copy
	double* L = nullptr;    // [nnzL*36] block values, column major by blocks, each block row-major 6x6
	double* Dinv = nullptr; // [M*36] inverse of the diagonal Cholesky factors (lower triangular)
	double* diag0 = nullptr; // [M*6] diagonal of the scaled S as it was scattered (new numbering): what a pivot of the separators is held against
	double* dscale = nullptr; // [M*6] the scaling D^-1/2 (powers of two; new numbering): right-hand sides enter and solutions leave through it
	int* d_err = nullptr;
};

// Distributed factorisation (feature-sharded tree runs, lsfm_symbolic.hpp col_owner): which of a launch's work-groups take
// part -- the ones whose first column belongs to `want` (a rank's block, or -1: the shared separator columns).  col_owner == null: all.
struct OwnFilter {
	const int* col_owner = nullptr;
	int want = 0;
	__device__ __forceinline__ bool skip(int col) const { return col_owner && col_owner[col] != want; }
};
__device__ __forceinline__ int find_row(const int* __restrict__ rowidx, int lo, int hi, int target)
{
	while (lo < hi) { int mid = (lo + hi) >> 1; if (rowidx[mid] < target) lo = mid + 1; else hi = mid; }
	return lo;
}

// ---------------------------------------------------------------------------------------------------------------
// Scaled matrix, fixed-point accumulators.  The factorisation works on  D^-1/2 (P S P^T) D^-1/2  with D the diagonal of S rounded
// to powers of four: an exact scaling (no rounding: Cholesky commutes with it), after which every diagonal entry lies in
// [1/8, 1) and -- the matrix and all its Schur complements being positive definite -- every entry, and every partial sum of
// the updates  sum_k l_ik l_jk  an entry ever receives (Cauchy-Schwarz over any subset of the columns), lies in (-1, 1).
// The blocks of the supernode-group columns, the ones several work-groups of a launch add to, are therefore kept as 64-bit
// FIXED-POINT numbers in units of 2^-61 while they accumulate: integer atomics are associative, so the sum no longer depends
// on the order the atomics land in (two runs on the same S give the same factor bit for bit; round 3's fp64 atomics made the
// root system of a 16 384-map monocular tree come out indefinite in one run out of fifteen), and it is more accurate than
// the fp64 sum it replaces: every addend is rounded once to 2^-62 of the diagonal, instead of every partial sum to 2^-53 of
// its own size.  The panel kernel converts a block back when it loads it.  Leaf columns (one work-group owns each: no
// atomics) stay doubles.  Right-hand sides enter scaled (k_perm_in) and leave unscaled (k_perm_out_dot).
// ---------------------------------------------------------------------------------------------------------------
#define FX_ONE 0x1p61
#define FX_INV 0x1p-61
__device__ __forceinline__ long long fx_from(double v) { return __double2ll_rn(fmin(fmax(v, -2.0), 2.0) * FX_ONE); }
__device__ __forceinline__ double fx_to(long long a) { return (double)a * FX_INV; }
__device__ __forceinline__ void fx_atomic_sub(double* slot, double t)
{
	// (no return value used: global_atomic_add_u64 without a round trip)
	__hip_atomic_fetch_add(reinterpret_cast<unsigned long long*>(slot), (unsigned long long)fx_from(-t), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// power of two s with s^2 d in [1/8, 1): the scale of a scalar row / column whose diagonal entry is d
__device__ __forceinline__ int scale_exp(double d)
{
	if (!(d > 0) || !(d < 1e300)) return 0; // (not positive definite: reported by the pivot test)
	return (__builtin_amdgcn_frexp_exp(d) + 1) >> 1;
}

// A (upper blocks of S, old numbering) -> lower blocks of the scaled P S P^T in L's storage (fixed point in the group columns:
// col_task[j] >= ntask0), its diagonal to diag0, the scale factors to dscale
__global__ void k_chol_scatter(int nnzb, const unsigned long long* __restrict__ keys, const double* __restrict__ S, const int* __restrict__ srow,
                               const int* __restrict__ pinv, const int* __restrict__ colptr, const int* __restrict__ rowidx,
                               const unsigned char* __restrict__ fixed, const int* __restrict__ col_task, int ntask0, double* __restrict__ L,
                               double* __restrict__ diag0, double* __restrict__ dscale, const int* __restrict__ col_owner, int rank)
{
	int e = blockIdx.x * blockDim.x + threadIdx.x;
	if (e >= nnzb) return;
	const unsigned long long key = keys[e];
	const int p = (int)(key >> 32), q = (int)(key & 0xffffffffull);
	int i = pinv[p], j = pinv[q];
	bool tr = false; // stored block is S(p,q); the lower block L(i,j), i >= j, is S(perm i, perm j)
	if (i < j) { int t = i; i = j; j = t; tr = true; }
	const int pos = find_row(rowidx, colptr[j], colptr[j + 1], i);
	const double* s = S + (size_t)e * 36;
	double* d = L + (size_t)pos * 36;
	const int rowp = tr ? q : p, colp = tr ? p : q; // old indices of the block's rows / columns
	// scales from the diagonal blocks of S (the first block of every block row)
	const double* dr = S + (size_t)srow[rowp] * 36;
	const double* dc = S + (size_t)srow[colp] * 36;
	int kr[6], kc[6];
	for (int r = 0; r < 6; r++)
	{
		kr[r] = scale_exp((fixed && fixed[(size_t)rowp * 6 + r]) ? 1.0 : dr[r * 7]);
		kc[r] = scale_exp((fixed && fixed[(size_t)colp * 6 + r]) ? 1.0 : dc[r * 7]);
	}
	const bool fx = col_task[j] >= ntask0;
	// (distributed: a rank starts the columns of its own block from S, rank 0 the shared ones too; everybody else's stay zero --
	// the scaling and the diagonal are everybody's)
	const bool put = !col_owner || col_owner[j] == rank || (col_owner[j] < 0 && rank == 0);
	for (int r = 0; r < 6; r++)
		for (int c = 0; c < 6; c++)
		{
			double v = tr ? s[c * 6 + r] : s[r * 6 + c];
			if (fixed && (fixed[(size_t)rowp * 6 + r] || fixed[(size_t)colp * 6 + c])) v = (rowp == colp && r == c) ? 1.0 : 0.0;
			v = ldexp(v, -(kr[r] + kc[c]));
			if (put)
			{
				if (fx) reinterpret_cast<long long*>(d)[r * 6 + c] = fx_from(v);
				else d[r * 6 + c] = v;
			}
			if (i == j && r == c) { diag0[(size_t)j * 6 + r] = v; dscale[(size_t)j * 6 + r] = ldexp(1.0, -kr[r]); }
		}
}

// one work-group factors one block column: L_jj = chol(A_jj); L_ij = A_ij L_jj^-T; A_ik -= L_ij L_kj^T for the blocks
// below (right-looking; targets in other columns are updated atomically because the columns of one level run together)
// pivot block of column j (block c0 of Lb, memory or LDS): L_jj = chol(A_jj) in place, its inverse to Dinv and to sLi
// (LDS, 36 doubles); ends with a barrier
__device__ void chol_pivot(int j, int c0, double* L, double* __restrict__ Dinv, int* err, double* sLi)
{
	const int tid = threadIdx.x;
	if (tid < LSFM_WAVE)
	{
		// 6x6 Cholesky and its inverse by the first wave: lane 6 r + c holds element (r,c), pivots and columns travel
		// by shuffles (one thread doing it alone took ~3 us of the ~5 us column step on the critical path)
		const bool in = tid < 36;
		const int r = in ? tid / 6 : 0, c = in ? tid % 6 : 0;
		double a = in ? L[(size_t)c0 * 36 + tid] : 0.0;
		bool ok = true;
#pragma unroll
		for (int k = 0; k < 6; k++)
		{
			double d = __shfl(a, k * 6 + k, LSFM_WAVE);
			if (!(d > 0)) { ok = false; d = 1.0; }
			const double piv = sqrt(d);
			if (in && c == k) a = (r == k) ? piv : (r > k ? a / piv : a);
			const double lrk = __shfl(a, r * 6 + k, LSFM_WAVE), lck = __shfl(a, c * 6 + k, LSFM_WAVE);
			if (in && r > k && c > k) a -= lrk * lck;
		}
		if (c > r) a = 0.0;
		if (!ok && tid == 0) atomicExch(err, 1 + j);
		if (in) L[(size_t)c0 * 36 + tid] = a;
		// inverse, column `tid` per lane (lanes 0..5): L x = e_tid by forward substitution, L's entries broadcast
		double x[6];
#pragma unroll
		for (int i = 0; i < 6; i++)
		{
			double sacc = (i == tid) ? 1.0 : 0.0;
#pragma unroll
			for (int k = 0; k < i; k++) sacc -= __shfl(a, i * 6 + k, LSFM_WAVE) * x[k];
			x[i] = sacc / __shfl(a, i * 6 + i, LSFM_WAVE);
		}
		if (tid < 6)
		{
#pragma unroll
			for (int i = 0; i < 6; i++)
			{
				const double v = i >= tid ? x[i] : 0.0;
				sLi[i * 6 + tid] = v;
				Dinv[(size_t)j * 36 + i * 6 + tid] = v;
			}
		}
	}
	__syncthreads();
}
// L_ij = A_ij * Li^T for the n blocks below the pivot block c0: one thread per (block, row)
__device__ void chol_scale_column(int c0, int n, double* L, const double* sLi)
{
	for (int w = threadIdx.x; w < n * 6; w += blockDim.x)
	{
		double* blk = L + (size_t)(c0 + 1 + w / 6) * 36 + (w % 6) * 6;
		double a[6], o[6];
		for (int k = 0; k < 6; k++) a[k] = blk[k];
		for (int c = 0; c < 6; c++)
		{
			double s = 0;
			for (int k = 0; k <= c; k++) s = fma(a[k], sLi[c * 6 + k], s);
			o[c] = s;
		}
		for (int k = 0; k < 6; k++) blk[k] = o[k];
	}
}
__device__ void chol_factor_column(int j, const int* __restrict__ colptr, const int* __restrict__ rowidx, double* __restrict__ L,
                                   double* __restrict__ Dinv, int* err)
{
	__shared__ double sLi[36];
	const int c0 = colptr[j], n = colptr[j + 1] - c0 - 1;
	chol_pivot(j, c0, L, Dinv, err, sLi);
	chol_scale_column(c0, n, L, sLi);
	__syncthreads();
}

// one trailing update of column j: blocks a >= b below the diagonal give L_a L_b^T, subtracted from block (ra, rb)
template <bool ATOMIC>
__device__ __forceinline__ void chol_pair_update(int c0, int a, int b, const int* __restrict__ colptr, const int* __restrict__ rowidx, double* __restrict__ L)
{
	const int ra = rowidx[c0 + 1 + a], rb = rowidx[c0 + 1 + b];
	double La[36], Lb[36], T[36];
	ld<36>(La, L + (size_t)(c0 + 1 + a) * 36);
	ld<36>(Lb, L + (size_t)(c0 + 1 + b) * 36);
	mmt<6, 6, 6, false>(La, Lb, T);
	// the rows of column j from rb on are a subset of column rb's rows; when the two lists coincide (columns of one
	// separator: nested patterns) the target is at the same offset, else binary search
	const int cb = colptr[rb], nb = colptr[rb + 1] - cb;
	int pos = cb + (a - b);
	if (!(a - b < nb && rowidx[pos] == ra)) pos = find_row(rowidx, cb, cb + nb, ra);
	double* d = L + (size_t)pos * 36;
	if (ATOMIC) { for (int q = 0; q < 36; q++) atomic_add_f64(d + q, -T[q]); }
	else { for (int q = 0; q < 36; q++) d[q] -= T[q]; } // the caller owns the target column: one pair per target block
}
// the pairs whose target column rb is one of the first m rows (the rows inside the task): needed before the task's next column
__device__ void chol_column_update_inner(int j, int m, const int* __restrict__ colptr, const int* __restrict__ rowidx, double* __restrict__ L, int first, int stride)
{
	const int c0 = colptr[j], n = colptr[j + 1] - c0 - 1;
	for (int idx = first; idx < m * n; idx += stride)
	{
		const int b = idx / n, a = idx - b * n;
		if (a >= b) chol_pair_update<false>(c0, a, b, colptr, rowidx, L); // in-task columns: only this work-group touches them now
	}
}
// the pairs with b >= m: targets in columns outside the task, nobody inside the task waits for them.  Other columns
// update the same blocks, so these are atomics -- made contiguous: every lane parks its 6x6 product in LDS and the
// work-group adds block after block with consecutive lanes on consecutive doubles (one lane per block scatters 64
// rows per wave instruction: ~0.1 TB/s)
#define CHOL_OUT_THREADS 128
template <bool FX>
__device__ void chol_column_update_outer(int j, int m, const int* __restrict__ colptr, const int* __restrict__ rowidx, double* __restrict__ L, int first, int stride)
{
	__shared__ double sT[CHOL_OUT_THREADS * 37];
	__shared__ int spos[CHOL_OUT_THREADS];
	const int c0 = colptr[j], n = colptr[j + 1] - c0 - 1 - m;
	const int npairs = n * (n + 1) / 2;
	const int tid = threadIdx.x;
	for (int base = first; base < npairs; base += stride)
	{
		const int pr = base + tid;
		int pos = -1;
		if (pr < npairs)
		{
			int a = (int)((sqrt(8.0 * pr + 1.0) - 1.0) * 0.5);
			while (a * (a + 1) / 2 > pr) a--;
			while ((a + 1) * (a + 2) / 2 <= pr) a++;
			const int b = pr - a * (a + 1) / 2 + m;
			a += m;
			const int ra = rowidx[c0 + 1 + a], rb = rowidx[c0 + 1 + b];
			double La[36], Lb[36], T[36];
			ld<36>(La, L + (size_t)(c0 + 1 + a) * 36);
			ld<36>(Lb, L + (size_t)(c0 + 1 + b) * 36);
			mmt<6, 6, 6, false>(La, Lb, T);
			const int cb = colptr[rb], nb = colptr[rb + 1] - cb;
			pos = cb + (a - b);
			if (!(a - b < nb && rowidx[pos] == ra)) pos = find_row(rowidx, cb, cb + nb, ra);
			for (int q = 0; q < 36; q++) sT[tid * 37 + q] = T[q];
		}
		spos[tid] = pos;
		__syncthreads();
		for (int idx = tid; idx < CHOL_OUT_THREADS * 36; idx += CHOL_OUT_THREADS)
		{
			const int p = idx / 36, q = idx - p * 36;
			const int ps = spos[p];
			if (ps >= 0)
			{
				if (FX) fx_atomic_sub(L + (size_t)ps * 36 + q, sT[p * 37 + q]); // (targets are supernode-group columns: fixed point)
				else atomic_add_f64(L + (size_t)ps * 36 + q, -sT[p * 37 + q]);
			}
		}
		__syncthreads();
	}
}

// forward substitution, right-looking: y_j = Li_j v_j ; v_i -= L_ij y_j
__device__ void chol_fwd_column(int j, const int* __restrict__ colptr, const int* __restrict__ rowidx, const double* __restrict__ L,
                                const double* __restrict__ Dinv, double* __restrict__ v)
{
	const int c0 = colptr[j], n = colptr[j + 1] - c0 - 1;
	const int tid = threadIdx.x, nt = blockDim.x;
	__shared__ double sy[6];
	if (tid < 6)
	{
		const double* Li = Dinv + (size_t)j * 36;
		double s = 0;
		for (int k = 0; k <= tid; k++) s = fma(Li[tid * 6 + k], v[(size_t)j * 6 + k], s);
		sy[tid] = s;
	}
	__syncthreads();
	if (tid < 6) v[(size_t)j * 6 + tid] = sy[tid];
	for (int w = tid; w < n * 6; w += nt)
	{
		const int e = c0 + 1 + w / 6, r = w % 6;
		const double* blk = L + (size_t)e * 36 + r * 6;
		double s = 0;
		for (int k = 0; k < 6; k++) s = fma(blk[k], sy[k], s);
		atomic_add_f64(v + (size_t)rowidx[e] * 6 + r, -s);
	}
}
// backward substitution: x_j = Li_j^T (y_j - sum_i L_ij^T x_i)   (all i > j are final)
__device__ void chol_bwd_column(int j, const int* __restrict__ colptr, const int* __restrict__ rowidx, const double* __restrict__ L,
                                const double* __restrict__ Dinv, double* __restrict__ v)
{
	const int c0 = colptr[j], n = colptr[j + 1] - c0 - 1;
	const int tid = threadIdx.x, nt = blockDim.x;
	__shared__ double red[256];
	__shared__ double ss[6];
	const int c = tid % 6, g = tid / 6, ng = nt / 6;
	double s = 0;
	if (g < ng)
		for (int e = g; e < n; e += ng)
		{
			const double* blk = L + (size_t)(c0 + 1 + e) * 36;
			const double* xi = v + (size_t)rowidx[c0 + 1 + e] * 6;
			for (int r = 0; r < 6; r++) s = fma(blk[r * 6 + c], xi[r], s);
		}
	red[tid] = (g < ng) ? s : 0.0;
	__syncthreads();
	if (tid < 6)
	{
		double t = v[(size_t)j * 6 + tid];
		for (int k = 0; k < ng; k++) t -= red[k * 6 + tid];
		ss[tid] = t;
	}
	__syncthreads();
	if (tid < 6)
	{
		const double* Li = Dinv + (size_t)j * 36;
		double t = 0;
		for (int k = tid; k < 6; k++) t = fma(Li[k * 6 + tid], ss[k], t);
		v[(size_t)j * 6 + tid] = t;
	}
}
__global__ void __launch_bounds__(64) k_chol_fwd_level(const int* __restrict__ cols, const int* __restrict__ colptr, const int* __restrict__ rowidx,
                                                        const double* __restrict__ L, const double* __restrict__ Dinv, double* __restrict__ v)
{
	chol_fwd_column(cols[blockIdx.x], colptr, rowidx, L, Dinv, v);
}
__global__ void __launch_bounds__(64) k_chol_bwd_level(const int* __restrict__ cols, const int* __restrict__ colptr, const int* __restrict__ rowidx,
                                                        const double* __restrict__ L, const double* __restrict__ Dinv, double* __restrict__ v)
{
	chol_bwd_column(cols[blockIdx.x], colptr, rowidx, L, Dinv, v);
}
// tail: forward over the remaining columns in order, then straight back down over them
__global__ void __launch_bounds__(256) k_chol_solve_tail(int ncols, const int* __restrict__ cols, const int* __restrict__ colptr,
                                                          const int* __restrict__ rowidx, const double* __restrict__ L,
                                                          const double* __restrict__ Dinv, double* __restrict__ v)
{
	for (int k = 0; k < ncols; k++) { chol_fwd_column(cols[k], colptr, rowidx, L, Dinv, v); __threadfence(); __syncthreads(); }
	for (int k = ncols - 1; k >= 0; k--) { chol_bwd_column(cols[k], colptr, rowidx, L, Dinv, v); __threadfence(); __syncthreads(); }
}

// one work-group per task: its columns in ascending order (children before parents) / descending for the back solve
__device__ void chol_factor_task_global(int task, const int* __restrict__ task_ptr, const int* __restrict__ task_cols,
                                        const int* __restrict__ col_nin, const int* __restrict__ colptr,
                                        const int* __restrict__ rowidx, double* __restrict__ L, double* __restrict__ Dinv, int* err)
{
	const int b = task_ptr[task], e = task_ptr[task + 1];
	for (int k = b; k < e; k++)
	{
		const int j = task_cols[k];
		chol_factor_column(j, colptr, rowidx, L, Dinv, err);
		if (k + 1 < e)
		{
			chol_column_update_inner(j, col_nin[j], colptr, rowidx, L, threadIdx.x, blockDim.x);
			__threadfence();
			__syncthreads();
		}
	}
}
// the deferred updates of the level's tasks, into the columns above them: one column per blockIdx.x, pairs split over blockIdx.y
template <bool FX>
__global__ void __launch_bounds__(CHOL_OUT_THREADS) k_chol_update_outer(const int* __restrict__ cols, const int* __restrict__ col_nin,
                                                                         const int* __restrict__ colptr, const int* __restrict__ rowidx,
                                                                         double* __restrict__ L, OwnFilter of)
{
	const int j = cols[blockIdx.x];
	if (of.skip(j)) return;
	chol_column_update_outer<FX>(j, col_nin[j], colptr, rowidx, L, blockIdx.y * CHOL_OUT_THREADS, gridDim.y * CHOL_OUT_THREADS);
}
// Triangular solves by task.  The entries of v that belong to the task's own columns live in LDS while the work-group
// walks the task: a column step inside a task then costs LDS latency instead of a global atomic + fence round trip
// (measured 2.7 us per step, the critical path of the whole solve).  Rows outside the task (ancestors) are updated /
// read in global memory; nobody inside the task reads them.
// LDS per task column: its slice of v (6), the inverse pivot block (36) and three ints (column, first block, count):
// everything a column step needs except the sub-diagonal blocks themselves is fetched side by side before the walk
#define CHOL_TASK_LDS_PER_COL (6 * 8 + 36 * 8 + 3 * 4)
template <class FT>
__device__ __forceinline__ void chol_task_stage(int b, int e, const int* __restrict__ task_cols, const int* __restrict__ colptr,
                                                const FT* __restrict__ Dinv, const double* __restrict__ v, double* lv, double* sD, int* sj,
                                                int* sc0, int* sn)
{
	const int tid = threadIdx.x, nt = blockDim.x, nc = e - b;
	for (int q = tid; q < nc; q += nt)
	{
		const int j = task_cols[b + q];
		sj[q] = j;
		const int c0 = colptr[j];
		sc0[q] = c0; sn[q] = colptr[j + 1] - c0 - 1;
	}
	for (int q = tid; q < nc * 6; q += nt) lv[q] = v[(size_t)task_cols[b + q / 6] * 6 + q % 6];
	for (int q = tid; q < nc * 36; q += nt) sD[q] = (double)Dinv[(size_t)task_cols[b + q / 36] * 36 + q % 36];
	__syncthreads();
}
template <class FT>
__global__ void __launch_bounds__(256) k_chol_fwd_tasks(const int* __restrict__ task_ptr, const int* __restrict__ task_cols,
                                                         const int* __restrict__ col_task, const int* __restrict__ col_lpos, int task0,
                                                         const int* __restrict__ colptr, const int* __restrict__ rowidx, const FT* __restrict__ L,
                                                         const FT* __restrict__ Dinv, double* __restrict__ v, OwnFilter of)
{
	extern __shared__ double lds[];
	__shared__ double sy[6];
	const int b = task_ptr[blockIdx.x], e = task_ptr[blockIdx.x + 1], me = task0 + blockIdx.x, nc = e - b;
	if (of.skip(task_cols[b])) return;
	const int tid = threadIdx.x, nt = blockDim.x;
	double* lv = lds;
	double* sD = lds + 6 * nc;
	int* sj = reinterpret_cast<int*>(sD + 36 * nc);
	int *sc0 = sj + nc, *sn = sc0 + nc;
	chol_task_stage(b, e, task_cols, colptr, Dinv, v, lv, sD, sj, sc0, sn);
	for (int k = 0; k < nc; k++)
	{
		const int c0 = sc0[k], n = sn[k];
		if (tid < 6)
		{
			const double* Li = sD + k * 36;
			double s = 0;
			for (int q = 0; q <= tid; q++) s = fma(Li[tid * 6 + q], lv[k * 6 + q], s);
			sy[tid] = s;
		}
		__syncthreads();
		if (tid < 6) lv[k * 6 + tid] = sy[tid];
		for (int w = tid; w < n * 6; w += nt)
		{
			const int en = c0 + 1 + w / 6, r = w % 6;
			const FT* blk = L + (size_t)en * 36 + r * 6;
			double s = 0;
			for (int q = 0; q < 6; q++) s = fma((double)blk[q], sy[q], s);
			const int i = rowidx[en];
			if (col_task[i] == me) lds_add_f64(&lv[col_lpos[i] * 6 + r], -s);
			else atomic_add_f64(v + (size_t)i * 6 + r, -s);
		}
		__syncthreads();
	}
	for (int q = tid; q < nc * 6; q += nt) v[(size_t)sj[q / 6] * 6 + q % 6] = lv[q];
}
template <class FT>
__global__ void __launch_bounds__(256) k_chol_bwd_tasks(const int* __restrict__ task_ptr, const int* __restrict__ task_cols,
                                                         const int* __restrict__ col_task, const int* __restrict__ col_lpos, int task0,
                                                         const int* __restrict__ colptr, const int* __restrict__ rowidx, const FT* __restrict__ L,
                                                         const FT* __restrict__ Dinv, double* __restrict__ v, OwnFilter of)
{
	extern __shared__ double lds[];
	__shared__ double red[256];
	__shared__ double ss[6];
	const int b = task_ptr[blockIdx.x], e = task_ptr[blockIdx.x + 1], me = task0 + blockIdx.x, nc = e - b;
	if (of.skip(task_cols[b])) return;
	const int tid = threadIdx.x, nt = blockDim.x;
	const int c = tid % 6, g = tid / 6, ng = nt / 6;
	double* lv = lds;
	double* sD = lds + 6 * nc;
	int* sj = reinterpret_cast<int*>(sD + 36 * nc);
	int *sc0 = sj + nc, *sn = sc0 + nc;
	chol_task_stage(b, e, task_cols, colptr, Dinv, v, lv, sD, sj, sc0, sn);
	for (int k = nc - 1; k >= 0; k--)
	{
		const int c0 = sc0[k], n = sn[k];
		double s = 0;
		if (g < ng)
			for (int en = g; en < n; en += ng)
			{
				const FT* blk = L + (size_t)(c0 + 1 + en) * 36;
				const int i = rowidx[c0 + 1 + en];
				if (col_task[i] == me)
				{
					const double* xi = &lv[col_lpos[i] * 6];
					for (int r = 0; r < 6; r++) s = fma((double)blk[r * 6 + c], xi[r], s);
				}
				else
				{
					const double* xi = v + (size_t)i * 6;
					for (int r = 0; r < 6; r++) s = fma((double)blk[r * 6 + c], xi[r], s);
				}
			}
		red[tid] = (g < ng) ? s : 0.0;
		__syncthreads();
		if (tid < 6)
		{
			double t = lv[k * 6 + tid];
			for (int q = 0; q < ng; q++) t -= red[q * 6 + tid];
			ss[tid] = t;
		}
		__syncthreads();
		if (tid < 6)
		{
			const double* Li = sD + k * 36;
			double t = 0;
			for (int q = tid; q < 6; q++) t = fma(Li[q * 6 + tid], ss[q], t);
			lv[k * 6 + tid] = t;
		}
		__syncthreads();
	}
	for (int q = tid; q < nc * 6; q += nt) v[(size_t)sj[q / 6] * 6 + q % 6] = lv[q];
}

// ---------------------------------------------------------------------------------------------------------------
// Small tasks: everything the walk touches fits LDS.  The column steps of a task are a chain of dependent global
// round trips (pivot block, scaled column, updated targets: ~6 per column, 10-14 us measured); when all blocks of the
// task's columns (288 B each + row index) fit 60 KB they are fetched side by side once, the walk runs at LDS latency
// and the result is written back once.  The host puts the small tasks first in every task level.
// ---------------------------------------------------------------------------------------------------------------
struct SmallTask {
	int nc, nb;          // columns, blocks (pivot blocks included)
	double* sB;          // [nb * 36] blocks, column after column
	int* sR;             // [nb] row index of every block
	int *sj, *sc0, *sn, *sm, *sbo; // per column: global column, first global block, blocks below the pivot, in-task rows, first LDS block
};
__device__ __forceinline__ size_t small_task_bytes(int nc, int nb) { return (size_t)nb * (288 + 4) + (size_t)(5 * nc + 1) * 4 + 8; }
// lays the task out in LDS (base must be 8-byte aligned) and copies blocks and row indices in; ends with a barrier
__device__ void small_task_stage(SmallTask& t, char* base, int b0, int nc, const int* __restrict__ task_cols, const int* __restrict__ col_nin,
                                 const int* __restrict__ colptr, const int* __restrict__ rowidx, const double* __restrict__ L)
{
	const int tid = threadIdx.x, nt = blockDim.x;
	__shared__ int s_nb;
	t.nc = nc;
	int* meta = reinterpret_cast<int*>(base);
	t.sj = meta; t.sc0 = meta + nc; t.sn = meta + 2 * nc; t.sm = meta + 3 * nc; t.sbo = meta + 4 * nc; // sbo has nc + 1 entries
	for (int q = tid; q < nc; q += nt)
	{
		const int j = task_cols[b0 + q];
		t.sj[q] = j;
		const int c0 = colptr[j];
		t.sc0[q] = c0; t.sn[q] = colptr[j + 1] - c0 - 1; t.sm[q] = col_nin ? col_nin[j] : 0;
	}
	__syncthreads();
	if (tid == 0)
	{
		int acc = 0;
		for (int q = 0; q < nc; q++) { t.sbo[q] = acc; acc += 1 + t.sn[q]; }
		t.sbo[nc] = acc;
		s_nb = acc;
	}
	__syncthreads();
	t.nb = s_nb;
	size_t off = ((size_t)(5 * nc + 1) * 4 + 7) & ~(size_t)7;
	t.sB = reinterpret_cast<double*>(base + off);
	t.sR = reinterpret_cast<int*>(base + off + (size_t)t.nb * 288);
	for (int e = tid; e < t.nb; e += nt)
	{
		int lo = 0, hi = nc - 1; // column of LDS block e
		while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (t.sbo[mid] <= e) lo = mid; else hi = mid - 1; }
		t.sR[e] = rowidx[t.sc0[lo] + (e - t.sbo[lo])];
	}
	for (int i = tid; i < t.nb * 36; i += nt)
	{
		const int e = i / 36;
		int lo = 0, hi = nc - 1;
		while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (t.sbo[mid] <= e) lo = mid; else hi = mid - 1; }
		t.sB[i] = L[(size_t)(t.sc0[lo] + (e - t.sbo[lo])) * 36 + (i - e * 36)];
	}
	__syncthreads();
}
// LDS column of a global column index that belongs to the task (columns are ascending)
__device__ __forceinline__ int small_task_col(const SmallTask& t, int col)
{
	int lo = 0, hi = t.nc - 1;
	while (lo < hi) { const int mid = (lo + hi) >> 1; if (t.sj[mid] < col) lo = mid + 1; else hi = mid; }
	return lo;
}
__device__ void chol_factor_task_lds(int task, const int* __restrict__ task_ptr, const int* __restrict__ task_cols,
                                     const int* __restrict__ col_nin, const int* __restrict__ colptr,
                                     const int* __restrict__ rowidx, double* __restrict__ L, double* __restrict__ Dinv, int* err)
{
	extern __shared__ double lds_d[];
	__shared__ double sLi[36];
	const int b0 = task_ptr[task], nc = task_ptr[task + 1] - b0;
	const int tid = threadIdx.x, nt = blockDim.x;
	SmallTask t;
	small_task_stage(t, reinterpret_cast<char*>(lds_d), b0, nc, task_cols, col_nin, colptr, rowidx, L);
	for (int k = 0; k < nc; k++)
	{
		const int cl = t.sbo[k], n = t.sn[k], m = t.sm[k];
		chol_pivot(t.sj[k], cl, t.sB, Dinv, err, sLi); // pivot block in LDS; Dinv to memory; ends with a barrier
		chol_scale_column(cl, n, t.sB, sLi);
		__syncthreads();
		// updates into the task's own columns (b < m), all in LDS; one pair per target block
		for (int idx = tid; idx < m * n; idx += nt)
		{
			const int b = idx / n, a = idx - b * n;
			if (a < b) continue;
			double La[36], Lb[36], T[36];
			ld<36>(La, t.sB + (size_t)(cl + 1 + a) * 36);
			ld<36>(Lb, t.sB + (size_t)(cl + 1 + b) * 36);
			mmt<6, 6, 6, false>(La, Lb, T);
			const int ra = t.sR[cl + 1 + a], rb = t.sR[cl + 1 + b];
			const int lq = small_task_col(t, rb);
			const int cb = t.sbo[lq], nbk = 1 + t.sn[lq];
			int pos = cb + (a - b);
			if (!(a - b < nbk && t.sR[pos] == ra)) pos = find_row(t.sR, cb, cb + nbk, ra);
			double* d = t.sB + (size_t)pos * 36;
			for (int q = 0; q < 36; q++) d[q] -= T[q];
		}
		__syncthreads();
	}
	// the factor goes back to memory once (the deferred updates into ancestor columns read it there)
	for (int i = tid; i < t.nb * 36; i += nt)
	{
		const int e = i / 36;
		int lo = 0, hi = nc - 1;
		while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (t.sbo[mid] <= e) lo = mid; else hi = mid - 1; }
		L[(size_t)(t.sc0[lo] + (e - t.sbo[lo])) * 36 + (i - e * 36)] = t.sB[i];
	}
}

// one task level of the factorisation in one launch: the first nsmall tasks of the level fit LDS whole, the others walk
// their columns in memory; both kinds run side by side
__global__ void __launch_bounds__(256) k_chol_factor_level(int nsmall, const int* __restrict__ task_ptr, const int* __restrict__ task_cols,
                                                            const int* __restrict__ col_nin, const int* __restrict__ colptr,
                                                            const int* __restrict__ rowidx, double* __restrict__ L, double* __restrict__ Dinv, int* err, OwnFilter of)
{
	if (of.skip(task_cols[task_ptr[blockIdx.x]])) return;
	if ((int)blockIdx.x < nsmall) chol_factor_task_lds(blockIdx.x, task_ptr, task_cols, col_nin, colptr, rowidx, L, Dinv, err);
	else chol_factor_task_global(blockIdx.x, task_ptr, task_cols, col_nin, colptr, rowidx, L, Dinv, err);
}


// ---------------------------------------------------------------------------------------------------------------
// Supernode groups: the factorisation above the leaf tasks.  With a path that revisits, the separators of the
// dissection are 20-70 poses wide and every separator column has 100-300 blocks below it: walking such a chain
// column by column in one work-group (the round-1 scheme) left the chip idle -- 27 ms for the top join of the
// NC3500-like set.  A group is a run of s <= CHOL_GS consecutive columns of one fundamental supernode: column c0+t
// holds [its diagonal block, the s-1-t later columns of the run, the nr common rows below the run], so block
// (row i of the common rows, column t) sits at colptr[c0+t] + (s-t) + i: the run is a dense trapezoid in the block
// storage as it is.  Per group level (children before parents) two launches:
//   k_sn_panel   every work-group factors the s x s diagonal blocks in LDS (redundantly: the other CUs would idle)
//                and solves its 16 block rows of the panel against them:  X = A L_dd^-T
//   k_sn_update  one lane per pair (a >= b) of common rows: block (r_a, r_b) -= sum_t X[a,t] X[b,t]^T, left through LDS
//                as contiguous atomics (groups of one level share ancestors)
// ---------------------------------------------------------------------------------------------------------------
/* CHOL_GS (most block columns of a group, 8): lsfm_symbolic.hpp */
#define SN_RB 16                    /* block rows of the panel per work-group */
#define SN_XS (6 * CHOL_GS + 1)     /* odd row stride of the panel rows in LDS */
#define SN_THREADS 256               /* 96 lanes own rows; the rest is there to keep more loads in flight */
#define SN_LD 8                      /* loads in flight per lane in the copy loops (a dependent load costs ~1.5 us) */
#define SN_PT (SN_THREADS + 64)      /* k_sn_panel: one more wave, the pivot wave (look-ahead factorisation of the diagonal blocks) */
// 1 / sqrt(x) without the ~300-cycle IEEE sqrt + divide chains (they sat on the critical path of every column step):
// hardware estimate + three Newton steps (full double precision up to an ulp or two -- the factor is a preconditioner
// under iterative refinement)
__device__ __forceinline__ double fast_rsqrt(double x)
{
	double r = __builtin_amdgcn_rsq(x);
	const double h = 0.5 * x;
	r = r * fma(-h * r, r, 1.5);
	r = r * fma(-h * r, r, 1.5);
	r = r * fma(-h * r, r, 1.5);
	return r;
}
// the double that lane `lane` (a compile-time constant) of the wave holds in v, as a wave-uniform value in scalar registers:
// two v_readlane_b32.  LDS reads in which every lane asks for the same address were measured at ~28 clocks per wave instruction
// on this chip (four waves on a CU share the LDS pipe): the 36 pivot-row entries of a block of dot products cost 18 of them per
// wave, 2 400 clocks per block and CU; one lane each loading one entry and 72 v_readlane_b32 cost the wave ~300 clocks of its own SIMD
__device__ __forceinline__ double wave_bcast(double v, int lane)
{
	const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane), hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
	return __hiloint2double(hi, lo);
}
// the same to full double precision with a shorter dependent chain, for the pivot wave (the chain of six of them per diagonal
// block is what a block column costs when its dot products are short): one Halley step, cubic -- the hardware estimate is good to
// ~2^-23, the step leaves ~2^-66 -- five dependent operations instead of nine
__device__ __forceinline__ double fast_rsqrt_h(double x)
{
	const double y = __builtin_amdgcn_rsq(x);
	const double e = fma(-x, y * y, 1.0);
	return fma(y, e * fma(0.375, e, 0.5), y);
}
__device__ __forceinline__ int sn_idx(int s, int u, int t) { return t * s - t * (t - 1) / 2 + (u - t); }

// Dense s x s (blocks) Cholesky of the run's diagonal part, one lane per scalar row, left-looking by block columns: for
// block column t every lane i >= 6t forms  a[c] = A[i][6t+c] - sum_{j<6t} L[i][j] L[6t+c][j]  (own row from LDS with an
// odd stride, the six pivot rows broadcast); the six lanes of the block's own rows publish theirs as the 6x6 diagonal block
// D, everybody factors D in registers (56 flops: cheaper than a second barrier-separated phase) and finishes its row.
// Two barriers per block column instead of the ~6 of the block-by-block walk, no idle lanes: ~15 us for 96 x 96 instead
// of ~65.  The panel rows X = A L_dd^-T are the same recurrence on rows below the diagonal part: other waves of the work-group
// carry them along, block column by block column, on their own SIMDs.
// fv != null: the forward substitution of ONE right-hand side rides along (the first preconditioner application of a
// level, known before the factorisation starts): fv_g^T is one more panel row, so the recurrence leaves y_g = L_dd^-1 fv_g
// in it, and every work-group takes X y_g off fv at its common rows -- what k_sn_fwd does in a launch of its own per
// group level (25 of them at the top join).  y_g goes to fw for the backward substitution.
// FUSED = false: grid (groups, chunks of SN_RB block rows of the panel); k_sn_update follows with the rank update.
// FUSED = true:  grid (groups, pairs (ca >= cb) of chunks of SN_RB / 2 block rows): the work-group solves the panel rows of
//                BOTH chunks and subtracts their product X_ca X_cb^T from the ancestors itself -- T = X X^T is a dense
//                (48 x 6s) x (6s x 48) contraction on v_mfma_f64_16x16x4_f64, leaving as 36 contiguous atomics per block --
//                so a group level is ONE launch.  The unfactored blocks are only read from L and the factor goes to a
//                second array Lg (same indexing): no work-group overwrites what another one of the level still reads,
//                nothing is parked.  (The solves of chunks shared by several pairs are redundant, like the diagonal part:
//                latency, not work, is what a level costs.)  Used while a level's panels have few enough rows
//                (chol_factor); beyond that the pairs would take more rounds of work-groups than the two launches.
typedef double sn_v4d __attribute__((ext_vector_type(4)));
// Profiling aid (make K9_TIMING=1): lane 0 of the first chunk's work-group of every group adds up the shader clocks of the phases
// of k_sn_panel: [0] index set-up, [1] blocks -> LDS, [2] the column loop, [3] right-hand side + inverse diagonal + stores, [4] rank
// update (fused), [5] work-groups counted, [6] sum of s.  Compiled out otherwise.
#ifdef LSFM_K9_TIMING
__device__ unsigned long long g_sn_t[32];
#define SNT_DECL unsigned long long snt_prev = __builtin_readcyclecounter()
#define SNT(i) do { if (threadIdx.x == 0 && blockIdx.y == 0) { const unsigned long long n_ = __builtin_readcyclecounter(); atomicAdd(&g_sn_t[(FUSED ? 16 : 0) + (i)], n_ - snt_prev); snt_prev = n_; } } while (0)
// inside the column loop: summed in registers, flushed once after the loop (an atomic per mark would be waited for at the next barrier)
#define SNL_DECL unsigned long long snl_[2] = { 0, 0 }
#define SNL(i) do { if (threadIdx.x == 0 && blockIdx.y == 0) { const unsigned long long n_ = __builtin_readcyclecounter(); snl_[i] += n_ - snt_prev; snt_prev = n_; } } while (0)
#define SNL_FLUSH do { if (threadIdx.x == 0 && blockIdx.y == 0) { atomicAdd(&g_sn_t[(FUSED ? 16 : 0) + 8], snl_[0]); atomicAdd(&g_sn_t[(FUSED ? 16 : 0) + 9], snl_[1]); } \
	if (threadIdx.x == 128 && blockIdx.y == 0) for (int q_ = 0; q_ < 6; q_++) atomicAdd(&g_sn_t[(FUSED ? 16 : 0) + 10 + q_], snp_[q_]); } while (0)
// the same for the first panel lane (tid 128, a lane that works in every phase): [0] slot reads [1] finish arithmetic [2] writes [3] wait at the barrier after B
// [4] dot products [5] wait at the barrier after A
#define SNP_DECL unsigned long long snp_[6] = { 0, 0, 0, 0, 0, 0 }, snp_prev = __builtin_readcyclecounter()
#define SNP(i) do { if (threadIdx.x == 128 && blockIdx.y == 0) { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); const unsigned long long n_ = __builtin_readcyclecounter(); snp_[i] += n_ - snp_prev; snp_prev = n_; } } while (0)
extern "C" void lsfm_debug_sn(unsigned long long* out, int reset)
{
	(void)hipDeviceSynchronize();
	if (out) (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_sn_t), sizeof(unsigned long long) * 32);
	if (reset) { unsigned long long z[32] = { 0 }; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_sn_t), z, sizeof(z)); }
}
#else
#define SNT_DECL do { } while (0)
#define SNT(i) do { } while (0)
#define SNL_DECL do { } while (0)
#define SNL(i) do { } while (0)
#define SNL_FLUSH do { } while (0)
#define SNP_DECL do { } while (0)
#define SNP(i) do { } while (0)
#endif
template <bool FUSED>
__global__ void __launch_bounds__(SN_PT) k_sn_panel(const int* __restrict__ grp_c0, const int* __restrict__ grp_s, const int* __restrict__ grp_nr,
                                                          const int* __restrict__ colptr, double* __restrict__ L, double* __restrict__ Lg,
                                                          double* __restrict__ Dinv, int* err, const int* __restrict__ rowidx, double* __restrict__ fv,
                                                          double* __restrict__ fw, int smax, const double* __restrict__ diag0, double piv_floor, int* nfloor, OwnFilter of)
{
	// LDS by the widest run of the LEVEL (smax block columns), not by CHOL_GS: most levels of most systems hold runs of 1-6
	// columns, and at 150 KB a work-group had a CU to itself -- a level of 2 000 small work-groups took 8 rounds
	extern __shared__ double Ms[];
	const int LD = 6 * smax;                       // dense scalar rows of the diagonal part
	const int xs = ((LD + 3) & ~3) + 1;            // odd row stride, with room for the zero padding of the MFMA k step
	// rows 0 .. 6 GS - 1: L_dd (dense scalar rows); rows 6 GS ..: the panel rows of this work-group.  Lanes 0..95 own the
	// diagonal rows, lanes 128..223 (two other waves, other SIMDs) the panel rows: the same recurrence, in step
	__shared__ double sD[36];
	__shared__ double sLp[2 * 28];        // the pivot wave's L_tt (lower triangle, 21) and 1 / diag (6), two slots in turn
	__shared__ double sInvD[6 * CHOL_GS]; // 1 / L_kk of the run
	__shared__ double sFl[6 * CHOL_GS];   // diagonal of the (scaled) S at the run's columns: what a pivot is held against
	__shared__ int sSrc[CHOL_GS * (CHOL_GS + 1) / 2], sDst[CHOL_GS * (CHOL_GS + 1) / 2], sCol[CHOL_GS];
	__shared__ int sRow[SN_RB];  // common row (position below the run) of every panel slot, -1: empty slot
	__shared__ int spos[(SN_RB / 2) * (SN_RB / 2)]; // FUSED: block of L every (a, b) product goes to, -1: none
	double* const Ls = Ms;
	double* const Xs = Ms + LD * xs;
	const int g = blockIdx.x, c0 = grp_c0[g], s = grp_s[g], nr = grp_nr[g];
	if (of.skip(c0)) return;
	constexpr int HB = SN_RB / 2;
	int ca = 0, cb = 0;
	if constexpr (FUSED)
	{
		const int nch = (nr + HB - 1) / HB, p = blockIdx.y;
		if (p > 0 && p >= nch * (nch + 1) / 2) return;
		ca = (int)((sqrtf(8.0f * p + 1.0f) - 1.0f) * 0.5f);
		while (ca * (ca + 1) / 2 > p) ca--;
		while ((ca + 1) * (ca + 2) / 2 <= p) ca++;
		cb = p - ca * (ca + 1) / 2;
	}
	else if (blockIdx.y > 0 && (int)blockIdx.y * SN_RB >= nr) return;
	const bool diag_pair = !FUSED || ca == cb; // this work-group writes its (first) chunk's rows of the factor
	const int tid = threadIdx.x, nt = blockDim.x;
	SNT_DECL;
	const int nb = s * (s + 1) / 2, n6 = 6 * s;
	// where every block of the run's diagonal part sits in the block storage / in the dense rows (one lane per block)
	for (int t = tid; t < s; t += nt) sCol[t] = colptr[c0 + t];
	// (fetched here, once: in the column loop this load sat between the two barriers of every block column -- a memory round trip per column)
	for (int t = tid; t < 6 * s; t += nt) sFl[t] = diag0[(size_t)c0 * 6 + t];
	if (tid < SN_RB)
	{
		int row;
		if constexpr (FUSED) row = tid < HB ? ca * HB + tid : (ca == cb ? nr : cb * HB + (tid - HB));
		else row = blockIdx.y * SN_RB + tid;
		sRow[tid] = row < nr ? row : -1;
	}
	__syncthreads();
	for (int e = tid; e < nb; e += nt)
	{
		int t = 0;
		while (sn_idx(s, s - 1, t) < e) t++; // column of packed block e (s <= 16: a short scan)
		const int u = t + (e - sn_idx(s, t, t));
		sSrc[e] = (sCol[t] + (u - t)) * 36;
		sDst[e] = 6 * u * xs + 6 * t;
	}
	const int rows0 = sCol[s - 1] + 1; // the common rows: what the last column of the run holds below its diagonal
	int upd_pos = -1;
	if constexpr (FUSED)
	{
		// targets of the rank update, fetched now: the loads fly while the blocks arrive (the position is parked in LDS after them)
		if (tid < HB * HB)
		{
			const int a = tid / HB, b = tid - a * HB;
			const int ia = sRow[a], ib = ca == cb ? sRow[b] : sRow[HB + b];
			if (ia >= 0 && ib >= 0 && ia >= ib)
			{
				const int ra = rowidx[rows0 + ia], rb = rowidx[rows0 + ib];
				// the rows of the run from rb on are a subset of column rb's rows; nested patterns put the target at the same offset
				const int cbk = colptr[rb], nbk = colptr[rb + 1] - cbk;
				upd_pos = cbk + (ia - ib);
				if (!(ia - ib < nbk && rowidx[upd_pos] == ra)) upd_pos = find_row(rowidx, cbk, cbk + nbk, ra);
			}
		}
	}
	__syncthreads();
	SNT(0);
	// blocks -> dense rows, two numbers per load, SN_LD loads in flight per lane (a dependent load costs ~1.5 us).  The blocks are
	// the fixed-point accumulators of the group columns: converted as they are stored to LDS
	const int nd2 = nb * 18, np2 = SN_RB * s * 18; // pairs of numbers: diagonal part, panel slots
	const bool pivot_wave = tid >= SN_THREADS;
	if (pivot_wave && tid - SN_THREADS < 36) sD[tid - SN_THREADS] = fx_to(reinterpret_cast<const long long*>(L)[(size_t)sCol[0] * 36 + (tid - SN_THREADS)]);
	for (int base = 0; base < (pivot_wave ? 0 : nd2 + np2); base += SN_THREADS * SN_LD)
	{
		longlong2 v[SN_LD];
#pragma unroll
		for (int i = 0; i < SN_LD; i++)
		{
			const int q = base + i * SN_THREADS + tid;
			if (q < nd2) { const int e = q / 18; v[i] = *reinterpret_cast<const longlong2*>(L + (size_t)sSrc[e] + 2 * (q - e * 18)); }
			else if (q < nd2 + np2)
			{
				const int qq = q - nd2, blk = qq / 18, il = blk / s, t = blk - il * s, row = sRow[il];
				if (row >= 0) v[i] = *reinterpret_cast<const longlong2*>(L + (size_t)(sCol[t] + (s - t) + row) * 36 + 2 * (qq - blk * 18));
			}
		}
#pragma unroll
		for (int i = 0; i < SN_LD; i++)
		{
			const int q = base + i * SN_THREADS + tid;
			if (q < nd2) { const int e = q / 18, w = 2 * (q - e * 18); double* d = &Ls[sDst[e] + (w / 6) * xs + w % 6]; d[0] = fx_to(v[i].x); d[1] = fx_to(v[i].y); }
			else if (q < nd2 + np2)
			{
				const int qq = q - nd2, blk = qq / 18, w = 2 * (qq - blk * 18), il = blk / s, t = blk - il * s;
				if (sRow[il] >= 0)
				{
					double* d = &Xs[(6 * il + w / 6) * xs + 6 * t + w % 6];
					d[0] = fx_to(v[i].x); d[1] = fx_to(v[i].y);
				}
			}
		}
	}
	const int XR = LD + 6 * SN_RB; // row of the right-hand side, owned by lane 128 + 6 SN_RB
	const bool with_fv = fv && diag_pair;
	if (with_fv && tid < n6) Ms[XR * xs + tid] = fv[(size_t)c0 * 6 + tid];
	if constexpr (FUSED) { if (tid < HB * HB) spos[tid] = upd_pos; }
	bool bad = false;
	// the 6x6 Cholesky of the block in sD by every lane of the pivot wave alike, published in slot (t & 1): L_tt (lower triangle) and 1 / diag
	auto pivot_chol = [&](int t) {
		const int k0 = 6 * t, pl = tid - SN_THREADS;
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
		__builtin_amdgcn_wave_barrier(); // (one wave: its LDS accesses are served in order; the compiler must keep them so)
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
		double d[21], di[6];
#pragma unroll
		for (int r = 0; r < 6; r++)
#pragma unroll
			for (int c = 0; c <= r; c++) d[r * (r + 1) / 2 + c] = sD[r * 6 + c];
#pragma unroll
		for (int k = 0; k < 6; k++)
		{
			double pv = d[k * (k + 1) / 2 + k];
			// Modified Cholesky for the separators.  What is left of the last diagonal blocks of a top separator after everything
			// below them has been eliminated is, for the weakly observable directions of a long monocular chain (scale drift),
			// the difference of numbers a thousand to 1e13 times larger.  A pivot is taken by its magnitude, bounded below by
			// piv_floor x the entry the (scaled) S had: the factor is the exact factor of S plus a small perturbation in those one
			// or two directions, which the CG around it removes in a few steps.  Only a pivot that is negative on the scale of S
			// itself (or not a number) means the system is not positive definite.
			const double flr = sFl[k0 + k];
			const double fl = piv_floor * flr, neg = piv_floor > 0 ? -0.01 * flr : 0.0; // (piv_floor = 0: any non-positive pivot is an error)
			if (!(pv > fl))
			{
				if (!(pv == pv) || !(pv > neg) || !(fl > 0)) { bad = true; pv = 1.0; }
				else { pv = fmax(fabs(pv), fl); if (nfloor && pl == 0 && blockIdx.y == 0) atomicAdd(nfloor, 1); }
			}
#ifdef LSFM_DEBUG_PIVOT
			if (pl == 0 && blockIdx.y == 0 && c0 + t >= 16380)
				printf("[piv] col %d k %d pv %.6e fl %.3e raw %.6e\n", c0 + t, k, pv, fl, d[k * (k + 1) / 2 + k]);
#endif
			di[k] = fast_rsqrt_h(pv);
			d[k * (k + 1) / 2 + k] = pv * di[k];
#pragma unroll
			for (int r = k + 1; r < 6; r++) d[r * (r + 1) / 2 + k] *= di[k];
#pragma unroll
			for (int r = k + 1; r < 6; r++)
#pragma unroll
				for (int c = k + 1; c <= r; c++) d[r * (r + 1) / 2 + c] -= d[r * (r + 1) / 2 + k] * d[c * (c + 1) / 2 + k];
		}
		if (pl == 0)
		{
			double* slot = sLp + (t & 1) * 28;
#pragma unroll
			for (int q = 0; q < 21; q++) slot[q] = d[q];
#pragma unroll
			for (int k = 0; k < 6; k++) slot[21 + k] = di[k];
		}
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
		__builtin_amdgcn_wave_barrier();
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
	};
	// ... and its rows into the dense rows (nobody reads the diagonal block's place before the end of the loop; the lanes that own
	// these rows have nothing to do for column t), 1 / diag with them
	auto pivot_rows = [&](int t) {
		const int k0 = 6 * t, pl = tid - SN_THREADS;
		const double* slot = sLp + (t & 1) * 28;
		if (pl < 36)
		{
			const int r = pl / 6, c = pl - 6 * r;
			Ls[(k0 + r) * xs + k0 + c] = c <= r ? slot[r * (r + 1) / 2 + c] : 0.0;
			if (pl < 6) sInvD[k0 + pl] = slot[21 + pl];
		}
	};
	// the first diagonal block: fetched by the pivot wave itself before the block loads (sD), factored while they arrive
	if (pivot_wave) pivot_chol(0);
	__syncthreads();
	SNT(1);
	// row of Ms this lane owns (-1: none).  The lanes of the last wave (tid >= SN_THREADS) own none: it is the pivot wave
	const int ri = tid < LD ? tid : ((tid >= 128 && tid < 128 + 6 * SN_RB) ? LD + (tid - 128) : ((with_fv && tid == 128 + 6 * SN_RB) ? XR : -1));
	const bool panel_lane = ri >= LD && (ri == XR || sRow[(ri - LD) / 6] >= 0);
	// The column loop, with the factorisation of the 6x6 diagonal blocks taken off everybody's path (look-ahead).  Per block
	// column t a lane that owns a row below block t does  A(t): a = its six entries of the column minus the dot products with the
	// columns before;  B(t): finish them against L_tt.  L_tt = chol(D_tt) is a chain of six dependent reciprocal square roots:
	// round 3 had every lane run it between the two barriers of every column.  Now the PIVOT WAVE does: while the others are in
	// A(t) it forms D_tt itself from the finished rows of block t (36 lanes, one entry each, dot products of length 6t), factors
	// it, publishes L_tt and 1 / diag(L_tt) in one of two slots and puts the block's rows in place; B(t) is 27 numbers through
	// scalar registers and 27 multiply-adds.  Two barriers per column, as before.
	auto pivot_step = [&](int t) {
		const int k0 = 6 * t, pl = tid - SN_THREADS;
		if (pl < 36)
		{
			const int r = pl / 6, c = pl - 6 * r;
			const double* xr = &Ls[(k0 + r) * xs];
			const double* xc = &Ls[(k0 + c) * xs];
			double d0 = c <= r ? xr[k0 + c] : xc[k0 + r], d1 = 0.0, d2 = 0.0; // (the lower triangle of the symmetric block)
			for (int j = 0; j < k0; j += 6) // (twelve reads in flight per step: a dependent LDS read costs ~150 clocks)
			{
				double p[6], q[6];
#pragma unroll
				for (int k = 0; k < 6; k++) { p[k] = xr[j + k]; q[k] = xc[j + k]; }
				d0 = fma(-p[0], q[0], d0); d1 = fma(-p[1], q[1], d1); d2 = fma(-p[2], q[2], d2);
				d0 = fma(-p[3], q[3], d0); d1 = fma(-p[4], q[4], d1); d2 = fma(-p[5], q[5], d2);
			}
			sD[pl] = d0 + (d1 + d2);
		}
		pivot_chol(t);
		pivot_rows(t);
	};
	double a[6];
	// A(0): nothing before the first column
	if (!pivot_wave && (panel_lane || (ri >= 6 && ri < n6)))
	{
		const double* xi = &Ms[ri * xs];
#pragma unroll
		for (int c = 0; c < 6; c++) a[c] = xi[c];
	}
	if (pivot_wave) pivot_rows(0);
	__syncthreads();
	SNT(7); // (the first diagonal block's rows put in place)
	SNL_DECL;
	SNP_DECL;
	for (int t = 0; t < s; t++)
	{
		const int k0 = 6 * t;
		// ---- B(t): finish column t against the published L_tt (rows below block t; the block's own rows are the pivot wave's).  The
		// 27 numbers of the slot are the same for every lane: lane l loads number l, they arrive through scalar registers ----
		const bool mineB = !pivot_wave && (panel_lane || (ri >= k0 + 6 && ri < n6));
		if (__builtin_amdgcn_ballot_w64(mineB) != 0ull)
		{
			const double sv = sLp[(t & 1) * 28 + ((tid & 63) < 27 ? (tid & 63) : 27)];
			SNP(0);
#pragma unroll
			for (int c = 0; c < 6; c++)
			{
				double v = a[c];
#pragma unroll
				for (int k = 0; k < c; k++) v = fma(-a[k], wave_bcast(sv, c * (c + 1) / 2 + k), v);
				a[c] = v * wave_bcast(sv, 21 + c);
				__builtin_amdgcn_sched_barrier(0); // (a row of L_tt at a time in scalar registers)
			}
			SNP(1);
			if (mineB)
			{
				double* xo = &Ms[ri * xs + k0];
#pragma unroll
				for (int c = 0; c < 6; c++) xo[c] = a[c];
			}
			SNP(2);
		}
		__syncthreads();
		SNP(3);
		SNL(1);
		if (t + 1 < s)
		{
			// ---- the pivot wave: L of the next diagonal block; everybody else A(t + 1): the dot products of the next column ----
			const int k1 = k0 + 6;
			if (pivot_wave) pivot_step(t + 1);
			else
			{
				// rows below block t + 1.  The 36 entries of the pivot rows that a block of dot products needs are the same for every
				// lane: lane l < 36 of every wave loads entry l, they reach the multiply-adds through scalar registers (wave_bcast)
				const int k2 = k1 + 6;
				const bool mine = panel_lane || (ri >= k2 && ri < n6);
				if (__builtin_amdgcn_ballot_w64(mine) != 0ull) // (wave-uniform: every lane of the wave takes part in the loads)
				{
					const int lane = tid & 63, pe = lane < 36 ? lane : lane - 36 < 28 ? lane - 36 : 0;
					const double* pp = &Ls[(k1 + pe / 6) * xs + pe % 6];
					const double* xi = &Ms[(mine ? ri : 0) * xs];
#pragma unroll
					for (int c = 0; c < 6; c++) a[c] = xi[k1 + c];
					double pv = pp[0];
					for (int v = 0; v <= t; v++)
					{
						const double pn = pp[v < t ? 6 * (v + 1) : 0]; // (the next block's entry is under way while this one is used)
						double xv[6];
#pragma unroll
						for (int k = 0; k < 6; k++) xv[k] = xi[6 * v + k];
#pragma unroll
						for (int k = 0; k < 6; k++)
						{
							// (a column of the 6x6 at a time: six independent multiply-adds.  Scheduling barriers that keep the v_readlane of a
							// block from being hoisted all at once -- they need more scalar registers than there are -- were measured slower:
							// 4 400 / 3 840 clocks per column with one per row / per two columns against 3 260 without)
#pragma unroll
							for (int c = 0; c < 6; c++) a[c] = fma(-xv[k], wave_bcast(pv, c * 6 + k), a[c]);
						}
						pv = pn;
					}
				}
				SNP(4);
			}
			__syncthreads();
			SNP(5);
			SNL(0);
		}
	}
	SNL_FLUSH;
	SNT(2);
	if (bad && tid == SN_THREADS) atomicExch(err, 1 + c0);
	if (with_fv)
	{
		const double* yg = &Ms[XR * xs];
		if (blockIdx.y == 0 && tid < n6) fw[(size_t)c0 * 6 + tid] = yg[tid];
		const int slot = ri >= LD && ri != XR ? (ri - LD) / 6 : -1;
		if (panel_lane && slot >= 0 && (!FUSED || slot < HB))
		{
			const double* xr = &Ms[ri * xs];
			double o0 = 0.0, o1 = 0.0;
			for (int k = 0; k + 1 < n6; k += 2) { o0 = fma(xr[k], yg[k], o0); o1 = fma(xr[k + 1], yg[k + 1], o1); } // n6 is even
			const int r = (ri - LD) - 6 * slot;
			atomic_add_f64(fv + (size_t)rowidx[rows0 + sRow[slot]] * 6 + r, -(o0 + o1));
		}
	}
	if (blockIdx.y == 0)
	{
		// inverse of every diagonal 6x6 factor (the triangular solves use it): lane (t, c) solves L_tt x = e_c
		if (tid < n6)
		{
			const int t = tid / 6, c = tid - 6 * t;
			const double* dg = &Ls[(6 * t) * xs + 6 * t];
			double x[6];
#pragma unroll
			for (int r = 0; r < 6; r++)
			{
				double v = r == c ? 1.0 : 0.0;
#pragma unroll
				for (int k = 0; k < r; k++) v = fma(-dg[r * xs + k], x[k], v);
				x[r] = v * sInvD[6 * t + r];
			}
#pragma unroll
			for (int r = 0; r < 6; r++) Dinv[(size_t)(c0 + t) * 36 + r * 6 + c] = r >= c ? x[r] : 0.0;
		}
		// the factored diagonal blocks, to the factor's own array (the other work-groups of the group read the unfactored ones from L)
		for (int q = tid; q < nb * 36; q += nt)
		{
			const int e = q / 36, w = q - e * 36;
			Lg[(size_t)sSrc[e] + w] = Ls[sDst[e] + (w / 6) * xs + w % 6];
		}
	}
	// the solved panel rows X = A L_dd^-T of this work-group's own chunk
	if (diag_pair)
		for (int q = tid; q < (FUSED ? HB : SN_RB) * s * 18; q += nt)
		{
			const int blk = q / 18, w = 2 * (q - blk * 18), il = blk / s, t = blk - il * s, row = sRow[il];
			if (row < 0) continue;
			const double* x = &Xs[(6 * il + w / 6) * xs + 6 * t + w % 6];
			*reinterpret_cast<double2*>(Lg + (size_t)(sCol[t] + (s - t) + row) * 36 + w) = make_double2(x[0], x[1]);
		}
	SNT(3);
#ifdef LSFM_K9_TIMING
	if (threadIdx.x == 0 && blockIdx.y == 0) { atomicAdd(&g_sn_t[(FUSED ? 16 : 0) + 5], 1ull); atomicAdd(&g_sn_t[(FUSED ? 16 : 0) + 6], (unsigned long long)s); }
#endif
	if constexpr (FUSED)
	{
		if (nr == 0) return;
		// ---- rank update of the ancestors: T = X_ca X_cb^T on the matrix cores.  Rows of X past 6 s are padded with zeros up to
		// a multiple of 4 (the k step of the instruction); rows of empty slots hold stale numbers: their products are dropped ----
		const int n6r = (n6 + 3) & ~3;
		if (n6r > n6)
			for (int q = tid; q < 6 * SN_RB * (n6r - n6); q += nt) Xs[(q / (n6r - n6)) * xs + n6 + q % (n6r - n6)] = 0.0;
		__syncthreads(); // (also: the diagonal rows in Ls are no longer read -- the products land there)
		const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
		const double* XB = ca == cb ? Xs : Xs + 6 * HB * xs;
		constexpr int NTL = (6 * HB) / 16; // 16-row tiles per side: 3
		constexpr int TS = 6 * HB + 1;     // row stride of the products in LDS
		double* sT = Ms; // (over the diagonal rows, and into the X rows when the diagonal part is small: hence the barrier below)
		constexpr int TPW = (NTL * NTL + SN_PT / 64 - 1) / (SN_PT / 64);
		sn_v4d acc[TPW];
#pragma unroll
		for (int i = 0; i < TPW; i++)
		{
			acc[i] = (sn_v4d){ 0.0, 0.0, 0.0, 0.0 };
			const int q = wave + (SN_PT / 64) * i;
			if (q < NTL * NTL)
			{
				const int ti = q / NTL, tj = q - ti * NTL;
				const double* pa = &Xs[(16 * ti + (lane & 15)) * xs + (lane >> 4)];
				const double* pb = &XB[(16 * tj + (lane & 15)) * xs + (lane >> 4)];
				for (int ks = 0; ks < n6r; ks += 4) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(pa[ks], pb[ks], acc[i], 0, 0, 0);
			}
		}
		__syncthreads();
#pragma unroll
		for (int i = 0; i < TPW; i++)
		{
			const int q = wave + (SN_PT / 64) * i;
			if (q < NTL * NTL)
			{
				const int ti = q / NTL, tj = q - ti * NTL;
				// C/D of v_mfma_f64_16x16x4_f64: col = lane & 15, row = (lane >> 4) + 4 * reg
#pragma unroll
				for (int e = 0; e < 4; e++) sT[(16 * ti + (lane >> 4) + 4 * e) * TS + 16 * tj + (lane & 15)] = acc[i][e];
			}
		}
		__syncthreads();
		for (int idx = tid; idx < HB * HB * 36; idx += nt)
		{
			const int p = idx / 36, q = idx - p * 36, ps = spos[p];
			if (ps >= 0) fx_atomic_sub(L + (size_t)ps * 36 + q, sT[(6 * (p / HB) + q / 6) * TS + 6 * (p % HB) + q % 6]);
		}
		SNT(4);
	}
}

#define SN_KS 4                      /* lanes per pair of rows in k_sn_update: each takes every 4th column of the run */
#define SN_PAIRS (SN_THREADS / SN_KS)
// the rank update of the levels whose panels are too tall for the fused kernel: X is read from the factor's array Lg, the
// products are subtracted from the ancestors' (still unfactored) blocks in L
__global__ void __launch_bounds__(SN_THREADS) k_sn_update(const int* __restrict__ grp_c0, const int* __restrict__ grp_s, const int* __restrict__ grp_nr,
                                                           const int* __restrict__ colptr, const int* __restrict__ rowidx, double* __restrict__ L,
                                                           const double* __restrict__ Lg, OwnFilter of)
{
	__shared__ double sT[SN_PAIRS * 37];
	__shared__ int spos[SN_PAIRS];
	const int g = blockIdx.x, c0 = grp_c0[g], s = grp_s[g], nr = grp_nr[g];
	if (of.skip(c0)) return;
	const int npairs = nr * (nr + 1) / 2;
	const int tid = threadIdx.x;
	const int rows0 = colptr[c0 + s - 1] + 1; // the common rows: what the last column of the run holds below its diagonal
	const int pl = tid / SN_KS, sub = tid - pl * SN_KS; // pair of this lane inside the round, its share of the columns
	for (int base = blockIdx.y * SN_PAIRS; base < npairs; base += gridDim.y * SN_PAIRS)
	{
		for (int q = tid; q < SN_PAIRS * 37; q += SN_THREADS) sT[q] = 0.0;
		__syncthreads();
		const int pr = base + pl;
		if (pr < npairs)
		{
			int a = (int)((sqrt(8.0 * pr + 1.0) - 1.0) * 0.5);
			while (a * (a + 1) / 2 > pr) a--;
			while ((a + 1) * (a + 2) / 2 <= pr) a++;
			const int b = pr - a * (a + 1) / 2;
			double T[36];
			zero<36>(T);
			// a dependent chain of s block loads per pair cost ~2 us each: four lanes share the chain, their sums meet in LDS
			for (int t = sub; t < s; t += SN_KS)
			{
				const size_t cb = (size_t)colptr[c0 + t] + (s - t);
				double La[36], Lb[36];
				ld<36>(La, Lg + (cb + a) * 36);
				ld<36>(Lb, Lg + (cb + b) * 36);
				mmt<6, 6, 6, true>(La, Lb, T);
			}
			if (sub < s)
				for (int q = 0; q < 36; q++) lds_add_f64(&sT[pl * 37 + q], T[q]);
			if (sub == 0)
			{
				const int ra = rowidx[rows0 + a], rb = rowidx[rows0 + b];
				const int cbk = colptr[rb], nbk = colptr[rb + 1] - cbk;
				int pos = cbk + (a - b);
				if (!(a - b < nbk && rowidx[pos] == ra)) pos = find_row(rowidx, cbk, cbk + nbk, ra);
				spos[pl] = pos;
			}
		}
		else if (sub == 0) spos[pl] = -1;
		__syncthreads();
		for (int idx = tid; idx < SN_PAIRS * 36; idx += SN_THREADS)
		{
			const int p = idx / 36, q = idx - p * 36;
			const int ps = spos[p];
			if (ps >= 0) fx_atomic_sub(L + (size_t)ps * 36 + q, sT[p * 37 + q]);
		}
		__syncthreads();
	}
}
// The same rank update on the matrix cores, for the panels too tall for the fused kernel (a synth-16k Mono tree: 100-300 common
// rows per group at its upper levels -- k_sn_update, one scalar 6x6 product chain per PAIR of rows, was 11 % of that tree's
// device time and loaded every block of X once per partner row).  One work-group per pair (ca >= cb) of 8-block-row chunks of a
// group's solved panel X (read from Lg): both chunks go to LDS as dense scalar rows once, T = X_ca X_cb^T is a (48 x 6s) x (6s x
// 48) contraction on v_mfma_f64_16x16x4_f64 (nine 16x16 tiles over the four waves), and the 64 products leave as 36 contiguous
// atomics each -- the tail of k_sn_panel<true> without its redundant solves.  grid (groups, pairs of chunks of the level's
// tallest panel, capped: a work-group walks pairs gridDim.y apart).
__global__ void __launch_bounds__(SN_THREADS) k_sn_syrk(const int* __restrict__ grp_c0, const int* __restrict__ grp_s, const int* __restrict__ grp_nr,
                                                         const int* __restrict__ colptr, const int* __restrict__ rowidx, double* __restrict__ L,
                                                         const double* __restrict__ Lg, int smax, OwnFilter of)
{
	extern __shared__ double Ms[];
	constexpr int HB = SN_RB / 2, NTL = (6 * HB) / 16, TS = 6 * HB + 1, TPW = (NTL * NTL + SN_THREADS / 64 - 1) / (SN_THREADS / 64);
	__shared__ int spos[HB * HB];
	__shared__ int sCol[CHOL_GS];
	const int g = blockIdx.x, c0 = grp_c0[g], s = grp_s[g], nr = grp_nr[g];
	if (nr == 0 || of.skip(c0)) return;
	const int tid = threadIdx.x, nt = blockDim.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
	const int n6 = 6 * s, n6r = (n6 + 3) & ~3;
	const int xs = ((6 * smax + 3) & ~3) + 1; // odd row stride, room for the zero padding of the MFMA k step
	double* const XA = Ms;
	double* const XBs = Ms + 6 * HB * xs;
	const int nch = (nr + HB - 1) / HB, npairs = nch * (nch + 1) / 2;
	for (int t = tid; t < s; t += nt) sCol[t] = colptr[c0 + t];
	__syncthreads();
	const int rows0 = sCol[s - 1] + 1; // the common rows: what the last column of the run holds below its diagonal
	for (int p = blockIdx.y; p < npairs; p += gridDim.y)
	{
		int ca = (int)((sqrtf(8.0f * p + 1.0f) - 1.0f) * 0.5f);
		while (ca * (ca + 1) / 2 > p) ca--;
		while ((ca + 1) * (ca + 2) / 2 <= p) ca++;
		const int cb = p - ca * (ca + 1) / 2;
		const bool two = ca != cb;
		// targets of the 64 products (fetched first: the loads fly while the panel rows arrive)
		if (tid < HB * HB)
		{
			const int a = tid / HB, b = tid - a * HB;
			const int ia = ca * HB + a, ib = cb * HB + b;
			int pos = -1;
			if (ia < nr && ib < nr && ia >= ib)
			{
				const int ra = rowidx[rows0 + ia], rb = rowidx[rows0 + ib];
				const int cbk = colptr[rb], nbk = colptr[rb + 1] - cbk;
				pos = cbk + (ia - ib);
				if (!(ia - ib < nbk && rowidx[pos] == ra)) pos = find_row(rowidx, cbk, cbk + nbk, ra);
			}
			spos[tid] = pos;
		}
		// the two chunks of X as dense scalar rows: 2 x HB x s blocks, two doubles per load, SN_LD loads in flight per lane;
		// rows past the panel's end and the padding columns are zero
		const int np2 = (two ? 2 : 1) * HB * s * 18;
		for (int base = 0; base < np2; base += nt * SN_LD)
		{
			double2 v[SN_LD];
#pragma unroll
			for (int i = 0; i < SN_LD; i++)
			{
				const int q = base + i * nt + tid;
				v[i] = make_double2(0.0, 0.0);
				if (q < np2)
				{
					const int blk = q / 18, il = blk / s, t = blk - il * s;
					const int row = (il < HB ? ca * HB + il : cb * HB + (il - HB));
					if (row < nr) v[i] = *reinterpret_cast<const double2*>(Lg + (size_t)(sCol[t] + (s - t) + row) * 36 + 2 * (q - blk * 18));
				}
			}
#pragma unroll
			for (int i = 0; i < SN_LD; i++)
			{
				const int q = base + i * nt + tid;
				if (q < np2)
				{
					const int blk = q / 18, w = 2 * (q - blk * 18), il = blk / s, t = blk - il * s;
					double* d = &Ms[(6 * il + w / 6) * xs + 6 * t + w % 6];
					d[0] = v[i].x; d[1] = v[i].y;
				}
			}
		}
		if (n6r > n6)
			for (int q = tid; q < (two ? 2 : 1) * 6 * HB * (n6r - n6); q += nt) Ms[(q / (n6r - n6)) * xs + n6 + q % (n6r - n6)] = 0.0;
		__syncthreads();
		const double* XB = two ? XBs : XA;
		sn_v4d acc[TPW];
#pragma unroll
		for (int i = 0; i < TPW; i++)
		{
			acc[i] = (sn_v4d){ 0.0, 0.0, 0.0, 0.0 };
			const int q = wave + (SN_THREADS / 64) * i;
			if (q < NTL * NTL)
			{
				const int ti = q / NTL, tj = q - ti * NTL;
				const double* pa = &XA[(16 * ti + (lane & 15)) * xs + (lane >> 4)];
				const double* pb = &XB[(16 * tj + (lane & 15)) * xs + (lane >> 4)];
				for (int ks = 0; ks < n6r; ks += 4) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(pa[ks], pb[ks], acc[i], 0, 0, 0);
			}
		}
		__syncthreads(); // the products are staged over the panel rows
		double* sT = Ms;
#pragma unroll
		for (int i = 0; i < TPW; i++)
		{
			const int q = wave + (SN_THREADS / 64) * i;
			if (q < NTL * NTL)
			{
				const int ti = q / NTL, tj = q - ti * NTL;
				// C/D of v_mfma_f64_16x16x4_f64: col = lane & 15, row = (lane >> 4) + 4 * reg
#pragma unroll
				for (int e = 0; e < 4; e++) sT[(16 * ti + (lane >> 4) + 4 * e) * TS + 16 * tj + (lane & 15)] = acc[i][e];
			}
		}
		__syncthreads();
		for (int idx = tid; idx < HB * HB * 36; idx += nt)
		{
			const int pp = idx / 36, q = idx - pp * 36, ps = spos[pp];
			if (ps >= 0) fx_atomic_sub(L + (size_t)ps * 36 + q, sT[(6 * (pp / HB) + q / 6) * TS + 6 * (pp % HB) + q % 6]);
		}
		__syncthreads(); // spos and the staged products are overwritten by the next pair
	}
}
static size_t sn_syrk_lds(int smax)
{
	const size_t xs = ((6 * (size_t)smax + 3) & ~(size_t)3) + 1;
	return std::max((size_t)(6 * SN_RB) * xs, (size_t)(6 * SN_RB / 2) * (6 * SN_RB / 2 + 1)) * sizeof(double);
}
// dynamic LDS of k_sn_panel for a level whose widest run has smax block columns (the products of the rank update are staged over
// the same memory: at least 48 x 49 doubles)
static size_t sn_panel_lds(int smax)
{
	const size_t LD = 6 * (size_t)smax, xs = ((LD + 3) & ~(size_t)3) + 1;
	return std::max((LD + 6 * SN_RB + 1) * xs, (size_t)(6 * SN_RB / 2) * (6 * SN_RB / 2 + 1)) * sizeof(double);
}
// the factor of the group columns back into L (only for the fall-back solves that read one array: chol_apply)
__global__ void k_sn_merge(const int* __restrict__ grp_c0, const int* __restrict__ grp_s, const int* __restrict__ colptr, const double* __restrict__ Lg,
                           double* __restrict__ L)
{
	const int g = blockIdx.x, c0 = grp_c0[g], s = grp_s[g];
	const size_t b0 = (size_t)colptr[c0] * 36, b1 = (size_t)colptr[c0 + s] * 36;
	for (size_t q = b0 + threadIdx.x; q < b1; q += blockDim.x) L[q] = Lg[q];
}


// ---------------------------------------------------------------------------------------------------------------
// Triangular solves by supernode group (the columns above the leaf tasks; the leaf sub-trees keep k_chol_fwd/bwd_tasks).
// One work-group per group, one launch per group level.  Forward: y_g = L_dd^-1 v_g by the row-owner recurrence on the
// dense rows of L_dd in LDS (two barriers per block column), then v[r_i] -= X[i,:] y_g for the common rows, four lanes per
// row.  y_g goes to a second vector (another group of the level may still be adding to v_g's neighbours; nobody reads
// v_g after its own level).  Backward: z = y_g - X^T x[rows], x_g = L_dd^-T z, written into v.
// ---------------------------------------------------------------------------------------------------------------
template <class FT> struct SnPair;
template <> struct SnPair<double> { typedef double2 T; };
template <> struct SnPair<float> { typedef float2 T; };
template <class FT>
__device__ __forceinline__ void sn_load_diag(int s, int c0, const int* __restrict__ colptr, const FT* __restrict__ L, double* Ls, int* sSrc,
                                             int* sDst, int* sCol)
{
	const int tid = threadIdx.x, nt = blockDim.x, nb = s * (s + 1) / 2;
	for (int t = tid; t < s; t += nt) sCol[t] = colptr[c0 + t];
	__syncthreads();
	for (int e = tid; e < nb; e += nt)
	{
		int t = 0;
		while (sn_idx(s, s - 1, t) < e) t++;
		const int u = t + (e - sn_idx(s, t, t));
		sSrc[e] = (sCol[t] + (u - t)) * 36;
		sDst[e] = 6 * u * SN_XS + 6 * t;
	}
	__syncthreads();
	const int nd2 = nb * 18;
	for (int base = 0; base < nd2; base += nt * SN_LD)
	{
		typename SnPair<FT>::T v[SN_LD];
#pragma unroll
		for (int i = 0; i < SN_LD; i++)
		{
			const int q = base + i * nt + tid;
			if (q < nd2) { const int e = q / 18; v[i] = *reinterpret_cast<const typename SnPair<FT>::T*>(L + (size_t)sSrc[e] + 2 * (q - e * 18)); }
		}
#pragma unroll
		for (int i = 0; i < SN_LD; i++)
		{
			const int q = base + i * nt + tid;
			if (q < nd2) { const int e = q / 18, w = 2 * (q - e * 18); double* d = &Ls[sDst[e] + (w / 6) * SN_XS + w % 6]; d[0] = (double)v[i].x; d[1] = (double)v[i].y; }
		}
	}
	__syncthreads();
}

template <class FT>
__global__ void __launch_bounds__(SN_THREADS) k_sn_fwd(const int* __restrict__ grp_c0, const int* __restrict__ grp_s, const int* __restrict__ grp_nr,
                                                        const int* __restrict__ colptr, const int* __restrict__ rowidx, const FT* __restrict__ L,
                                                        const FT* __restrict__ Dinv, double* __restrict__ v, double* __restrict__ w, OwnFilter of)
{
	__shared__ double Ls[6 * CHOL_GS * SN_XS];
	__shared__ double sDi[CHOL_GS * 36];
	__shared__ double sA[6], sY[6 * CHOL_GS];
	__shared__ int sSrc[CHOL_GS * (CHOL_GS + 1) / 2], sDst[CHOL_GS * (CHOL_GS + 1) / 2], sCol[CHOL_GS];
	const int g = blockIdx.x, c0 = grp_c0[g], s = grp_s[g], nr = grp_nr[g];
	if (of.skip(c0)) return;
	const int tid = threadIdx.x, n6 = 6 * s;
	for (int q = tid; q < s * 36; q += SN_THREADS) sDi[q] = (double)Dinv[(size_t)c0 * 36 + q];
	double acc = tid < n6 ? v[(size_t)c0 * 6 + tid] : 0.0;
	sn_load_diag(s, c0, colptr, L, Ls, sSrc, sDst, sCol);
	for (int t = 0; t < s; t++)
	{
		const int k0 = 6 * t;
		if (tid >= k0 && tid < k0 + 6) sA[tid - k0] = acc;
		__syncthreads();
		if (tid >= k0 && tid < k0 + 6)
		{
			const int c = tid - k0;
			double y = 0.0;
			for (int k = 0; k <= c; k++) y = fma(sDi[t * 36 + c * 6 + k], sA[k], y);
			sY[tid] = y;
		}
		__syncthreads();
		if (tid >= k0 + 6 && tid < n6)
		{
			const double* lr = &Ls[tid * SN_XS + k0];
#pragma unroll
			for (int k = 0; k < 6; k++) acc = fma(-lr[k], sY[k0 + k], acc);
		}
	}
	if (tid < n6) w[(size_t)c0 * 6 + tid] = sY[tid];
	// the common rows: v[r_i] -= sum_t X[i,t] y_t, four lanes per row (each every 4th column of the run)
	const int rows0 = colptr[c0 + s - 1] + 1;
	const int sub = tid & 3;
	for (int i = tid >> 2; i < nr; i += SN_THREADS / 4)
	{
		double o[6] = { 0, 0, 0, 0, 0, 0 };
		for (int t = sub; t < s; t += 4)
		{
			const FT* blk = L + ((size_t)sCol[t] + (s - t) + i) * 36;
			double b[36];
			ld<36>(b, blk);
#pragma unroll
			for (int r = 0; r < 6; r++)
#pragma unroll
				for (int k = 0; k < 6; k++) o[r] = fma(b[r * 6 + k], sY[6 * t + k], o[r]);
		}
#pragma unroll
		for (int r = 0; r < 6; r++)
		{
			o[r] += __shfl_xor(o[r], 1, LSFM_WAVE);
			o[r] += __shfl_xor(o[r], 2, LSFM_WAVE);
		}
		if (sub == 0)
		{
			double* dst = v + (size_t)rowidx[rows0 + i] * 6;
#pragma unroll
			for (int r = 0; r < 6; r++) atomic_add_f64(dst + r, -o[r]);
		}
	}
}

template <class FT>
__global__ void __launch_bounds__(SN_THREADS) k_sn_bwd(const int* __restrict__ grp_c0, const int* __restrict__ grp_s, const int* __restrict__ grp_nr,
                                                        const int* __restrict__ colptr, const int* __restrict__ rowidx, const FT* __restrict__ L,
                                                        const FT* __restrict__ Dinv, double* __restrict__ v, const double* __restrict__ w, OwnFilter of)
{
	__shared__ double Ls[6 * CHOL_GS * SN_XS];
	__shared__ double sDi[CHOL_GS * 36];
	__shared__ double sA[6], sZ[6 * CHOL_GS], sX[6 * CHOL_GS];
	__shared__ int sSrc[CHOL_GS * (CHOL_GS + 1) / 2], sDst[CHOL_GS * (CHOL_GS + 1) / 2], sCol[CHOL_GS];
	const int g = blockIdx.x, c0 = grp_c0[g], s = grp_s[g], nr = grp_nr[g];
	if (of.skip(c0)) return;
	const int tid = threadIdx.x, n6 = 6 * s;
	for (int q = tid; q < s * 36; q += SN_THREADS) sDi[q] = (double)Dinv[(size_t)c0 * 36 + q];
	if (tid < n6) sZ[tid] = w[(size_t)c0 * 6 + tid];
	sn_load_diag(s, c0, colptr, L, Ls, sSrc, sDst, sCol); // (ends with a barrier: sZ, sDi, sCol visible)
	// z -= X^T x over the common rows (all final: they belong to higher levels), four lanes per row
	const int rows0 = colptr[c0 + s - 1] + 1;
	const int sub = tid & 3;
	for (int i = tid >> 2; i < nr; i += SN_THREADS / 4)
	{
		const double* xr = v + (size_t)rowidx[rows0 + i] * 6;
		double x6[6];
		ld<6>(x6, xr);
		for (int t = sub; t < s; t += 4)
		{
			const FT* blk = L + ((size_t)sCol[t] + (s - t) + i) * 36;
			double b[36];
			ld<36>(b, blk);
#pragma unroll
			for (int c = 0; c < 6; c++)
			{
				double o = 0.0;
#pragma unroll
				for (int r = 0; r < 6; r++) o = fma(b[r * 6 + c], x6[r], o);
				lds_add_f64(&sZ[6 * t + c], -o);
			}
		}
	}
	__syncthreads();
	// x_g = L_dd^-T z: lane (t, c) owns column 6t + c, block columns from the last to the first
	double acc = tid < n6 ? sZ[tid] : 0.0;
	for (int t = s - 1; t >= 0; t--)
	{
		const int k0 = 6 * t;
		if (tid >= k0 && tid < k0 + 6) sA[tid - k0] = acc;
		__syncthreads();
		if (tid >= k0 && tid < k0 + 6)
		{
			const int c = tid - k0;
			double x = 0.0;
			for (int k = c; k < 6; k++) x = fma(sDi[t * 36 + k * 6 + c], sA[k], x);
			sX[tid] = x;
		}
		__syncthreads();
		if (tid < k0)
		{
#pragma unroll
			for (int k = 0; k < 6; k++) acc = fma(-Ls[(k0 + k) * SN_XS + tid], sX[k0 + k], acc);
		}
	}
	if (tid < n6) v[(size_t)c0 * 6 + tid] = sX[tid];
}

__global__ void k_to_float(size_t n, const double* __restrict__ a, float* __restrict__ b)
{
	size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) b[i] = (float)a[i];
}

// v = D^-1/2 P r  (the factor is the scaled matrix's: k_chol_scatter)
__global__ void k_perm_in(int M, const int* __restrict__ perm, const double* __restrict__ r, const unsigned char* __restrict__ fixed,
                          const double* __restrict__ dscale, double* __restrict__ v, int zero_from)
{
	size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= (size_t)M * 6) return;
	const size_t src = (size_t)perm[i / 6] * 6 + i % 6;
	// (distributed: the shared rows collect the ranks' partial sums; only rank 0 starts them from the right-hand side: zero_from = first shared row elsewhere)
	v[i] = ((fixed && fixed[src]) || (long)(i / 6) >= (long)zero_from) ? 0.0 : r[src] * dscale[i];
}
// z = P^T D^-1/2 v ; rz[nxt] += r . z
__global__ void k_perm_out_dot(int M, const int* __restrict__ pinv, const double* __restrict__ v, const double* __restrict__ r,
                               const unsigned char* __restrict__ fixed, const double* __restrict__ dscale, const int* __restrict__ pose_seg,
                               double* __restrict__ z, double* dot, int dot_stride, const int* __restrict__ col_owner, int rank)
{
	int row = blockIdx.x * blockDim.x + threadIdx.x;
	const bool ok = row < M;
	double acc = 0;
	int sg = 0;
	if (ok)
	{
		sg = pose_seg[row];
		const double* src = v + (size_t)pinv[row] * 6;
		const double* sc = dscale + (size_t)pinv[row] * 6;
		for (int i = 0; i < 6; i++)
		{
			double zz = src[i] * sc[i];
			if (fixed && fixed[(size_t)row * 6 + i]) zz = 0.0;
			// (distributed: a rank holds the solution at its own block's columns, rank 0 at the shared ones too; the sum over the ranks is z)
			if (col_owner) { const int ow = col_owner[pinv[row]]; if (!(ow == rank || (ow < 0 && rank == 0))) zz = 0.0; }
			z[(size_t)row * 6 + i] = zz;
			acc = fma(zz, r[(size_t)row * 6 + i], acc);
		}
	}
	if (dot) wave_scatter_add<1>(dot + (size_t)sg * dot_stride, &acc, ok);
}
// rz[nxt] += r . z  alone (distributed: z is complete only after the ranks' parts have been summed)
__global__ void k_rz_dot(int M, const double* __restrict__ z, const double* __restrict__ r, const int* __restrict__ pose_seg, double* dot, int dot_stride)
{
	int row = blockIdx.x * blockDim.x + threadIdx.x;
	const bool ok = row < M;
	double acc = 0;
	int sg = 0;
	if (ok)
	{
		sg = pose_seg[row];
		for (int i = 0; i < 6; i++) acc = fma(z[(size_t)row * 6 + i], r[(size_t)row * 6 + i], acc);
	}
	wave_scatter_add<1>(dot + (size_t)sg * dot_stride, &acc, ok);
}

// ---------------------------------------------------------------------------------------------------------------
// host: ordering + symbolic factorisation
// ---------------------------------------------------------------------------------------------------------------
struct CholHostIn {
	std::vector<unsigned long long> keys; // sorted upper pattern of S
	std::vector<int> origin;              // local map that brought each pose
};
// the two small device -> host copies of the analysis (a synchronisation), separate from the host work so that the
// caller can enqueue the numeric Schur assembly in between and let it run under the symbolic factorisation
static void chol_fetch(lsfm_context* ctx, const SchurSystem& sy, const int* d_origin, CholHostIn& in)
{
	const int M = sy.M, nnzb = sy.nnzb;
	in.keys.resize(nnzb);
	in.origin.resize(M);
	if (d_origin) LSFM_CHECK_HIP(hipMemcpyAsync(in.origin.data(), d_origin, (size_t)M * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
	else std::iota(in.origin.begin(), in.origin.end(), 0);
	d2h(ctx, in.keys.data(), sy.upper_keys, (size_t)nnzb * sizeof(unsigned long long));
}

// value arrays of a factorisation (per run; the index arrays may come from a plan)
static void chol_alloc_values(lsfm_context* ctx, CholDev& ch)
{
	Arena& sc = ctx->scratch;
	ch.L = sc.alloc<double>((size_t)ch.nnzL * 36); ch.Dinv = sc.alloc<double>((size_t)ch.M * 36);
	ch.wv = sc.alloc<double>((size_t)ch.M * 6);
	ch.diag0 = sc.alloc<double>((size_t)ch.M * 6);
	ch.dscale = sc.alloc<double>((size_t)ch.M * 6);
	ch.Lg = ch.ngroups ? sc.alloc<double>((size_t)ch.nnzL * 36) : nullptr; // (every block of a group column is written by the factorisation)
	dev_zero(ctx, ch.L, (size_t)ch.nnzL * 36 * sizeof(double));
	static const bool digest = getenv("LSFM_FACTOR_DIGEST") != nullptr; // (the digest reads all of Lg: the leaf columns' slots, never written, must not be noise)
	if (digest && ch.Lg) dev_zero(ctx, ch.Lg, (size_t)ch.nnzL * 36 * sizeof(double));
}

// What a first solve of a tree level leaves for the next runs of the same tree (LevelPlan::solve): the block pattern of S
// with its hash index and the whole symbolic factorisation, in one device allocation of their own.
struct SolvePlan {
	SchurSystem sy; // index members only (S, E, IV are per run)
	CholDev ch;     // index members + host vectors (L, Lg, Dinv, d_err are per run)
	int its = 1;    // refinement steps the first run needed ...
	bool mixed = false; // ... with the preconditioner in this precision
	double rel_tol = 0; // ... to this relative residual
	bool small = false; // the plan of a level on the one-launch dense path: no pattern, no factorisation (lsfm_small.hip)
	char* mem = nullptr;
	~SolvePlan() { if (mem) (void)hipFree(mem); }
};
static std::shared_ptr<void> solve_plan_store(lsfm_context* ctx, const SchurSystem& sy, const CholDev& ch, int its)
{
	auto sp = std::make_shared<SolvePlan>();
	const size_t M = sy.M, nnzb = sy.nnzb, cap = (size_t)sy.mask + 1;
	if (getenv("LSFM_DEBUG") && sy.k9.ns && sy.k9_tiles > 0)
	{
		// census of the Schur tiles by the number of poses that see them (negative: no panel variant took the tile)
		std::vector<int> h(sy.k9_tiles);
		d2h(ctx, h.data(), sy.k9.ns, sizeof(int) * h.size());
		int b16 = 0, b32 = 0, b48 = 0, b64 = 0, b96 = 0, more = 0;
		for (int v : h) { const int n = v < 0 ? -v : v; (n <= 16 ? b16 : n <= 32 ? b32 : n <= 48 ? b48 : n <= 64 ? b64 : n <= 96 ? b96 : more)++; }
		fprintf(stderr, "[lsfm] Schur tiles by poses: <=16 %d, <=32 %d, <=48 %d, <=64 %d, <=96 %d, more (or > 64 distinct: hash full) %d\n", b16, b32, b48, b64, b96, more);
	}
	struct Item { const void* src; size_t bytes; void** dst; };
	SolvePlan& P = *sp;
	P.sy = sy; P.ch = ch; P.its = its; P.mixed = ch.Lf != nullptr; P.rel_tol = ctx->pcg.rel_tol;
	P.sy.S = nullptr; P.sy.E = nullptr; P.sy.IV = nullptr;
	P.ch.L = nullptr; P.ch.Dinv = nullptr; P.ch.diag0 = nullptr; P.ch.dscale = nullptr; P.ch.Lg = nullptr; P.ch.Lgf = nullptr; P.ch.d_err = nullptr; P.ch.wv = nullptr; P.ch.Lf = nullptr; P.ch.Dinvf = nullptr;
	std::vector<Item> items = {
		{ sy.rowptr, (M + 1) * 4, (void**)&P.sy.rowptr }, { sy.colidx, (nnzb + 1) * 4, (void**)&P.sy.colidx },
		{ sy.upper_keys, nnzb * 8, (void**)&P.sy.upper_keys }, { sy.longrows, (M + 1) * 4, (void**)&P.sy.longrows },
		{ sy.d_nlong, 4, (void**)&P.sy.d_nlong }, { sy.tab, cap * 8, (void**)&P.sy.tab }, { sy.hval, cap * 4, (void**)&P.sy.hval },
		{ ch.blob, ch.blob_ints * 4, (void**)&P.ch.blob },
		{ sy.gent, sy.gent ? nnzb * 16 : 0, (void**)&P.sy.gent }, { sy.goth, sy.goth ? nnzb * 8 : 0, (void**)&P.sy.goth },
		{ sy.k9.ns, sy.k9.ns ? (size_t)sy.k9_tiles * 4 : 0, (void**)&P.sy.k9.ns }, { sy.k9.pose, sy.k9.pose ? (size_t)sy.k9_tiles * 64 * 4 : 0, (void**)&P.sy.k9.pose },
		{ sy.k9.eslot, sy.k9.eslot ? (size_t)sy.k9_NW : 0, (void**)&P.sy.k9.eslot },
		{ sy.k9.wlist, sy.k9.wlist ? (size_t)sy.k9_tiles * 3 * 4 : 0, (void**)&P.sy.k9.wlist }, { sy.k9.wcnt, sy.k9.wcnt ? (size_t)32 : (size_t)0, (void**)&P.sy.k9.wcnt },
	};
	P.sy.k9.record = 0;
	size_t total = 0;
	for (const Item& it : items) total += (it.bytes + 255) & ~(size_t)255;
	LSFM_CHECK_HIP(hipMalloc((void**)&P.mem, total + 256));
	size_t off = 0;
	for (const Item& it : items)
	{
		if (it.bytes) LSFM_CHECK_HIP(hipMemcpyAsync(P.mem + off, it.src, it.bytes, hipMemcpyDeviceToDevice, ctx->stream));
		*it.dst = it.src ? P.mem + off : nullptr;
		off += (it.bytes + 255) & ~(size_t)255;
	}
	// the factorisation's index arrays are slices of the blob
	const ptrdiff_t shift = (char*)P.ch.blob - (char*)ch.blob;
	auto rebase = [&](int*& p) { if (p) p = (int*)((char*)p + shift); };
	rebase(P.ch.colptr); rebase(P.ch.rowidx); rebase(P.ch.perm); rebase(P.ch.pinv); rebase(P.ch.order); rebase(P.ch.task_cols);
	rebase(P.ch.task_ptr); rebase(P.ch.col_task); rebase(P.ch.col_lpos); rebase(P.ch.col_nin); rebase(P.ch.grp_c0); rebase(P.ch.grp_s);
	rebase(P.ch.grp_nr); rebase(P.ch.col_owner);
	LSFM_CHECK_HIP(hipStreamSynchronize(ctx->stream));
	return sp;
}

// symbolic analysis on the host (lsfm_symbolic.cpp), then every index array of it to the device in ONE copy
// index arrays of a symbolic factorisation to the device (ctx->scratch / ctx->stream as the caller has set them), host vectors
// copied: what a plan keeps
static void chol_upload_index(lsfm_context* ctx, const CholSymbolic& sym, CholDev& ch)
{
	Arena& sc = ctx->scratch;
	ch.M = sym.M; ch.nnzL = sym.nnzL; ch.nlevels = sym.nlevels; ch.tail_begin = sym.tail_begin;
	ch.level_ptr = sym.level_ptr;
	ch.tlevel_ptr = sym.tlevel_ptr; ch.tlevel_maxsize = sym.tlevel_maxsize; ch.tlevel_col0 = sym.tlevel_col0; ch.tlevel_nsmall = sym.tlevel_nsmall;
	ch.tlevel_small_lds = sym.tlevel_small_lds; ch.tlevel_outer = sym.tlevel_outer;
	ch.ngroups = sym.ngroups; ch.glevel_ptr = sym.glevel_ptr; ch.glevel_maxnr = sym.glevel_maxnr; ch.glevel_maxs = sym.glevel_maxs;
	ch.work_total = sym.work_total; ch.work_shared = sym.work_shared;
	ch.first_shared = sym.first_shared; ch.shared_blk0 = sym.colptr[sym.first_shared]; ch.glevel_owned = sym.glevel_owned; ch.glevel_shared = sym.glevel_shared;
	const struct { int** dst; const std::vector<int>* v; } parts[] = {
		{ &ch.grp_c0, &sym.grp_c0 }, { &ch.grp_s, &sym.grp_s }, { &ch.grp_nr, &sym.grp_nr }, { &ch.col_nin, &sym.col_nin }, { &ch.col_task, &sym.col_task },
		{ &ch.col_lpos, &sym.col_lpos }, { &ch.task_cols, &sym.task_cols }, { &ch.task_ptr, &sym.task_ptr }, { &ch.colptr, &sym.colptr },
		{ &ch.rowidx, &sym.rowidx }, { &ch.perm, &sym.perm }, { &ch.pinv, &sym.pinv }, { &ch.order, &sym.order }, { &ch.col_owner, &sym.col_owner } };
	size_t total = 0;
	for (const auto& pt : parts) total += pt.v->size();
	static thread_local std::vector<int> blob;
	blob.resize(total);
	int* d_blob = sc.alloc<int>(total);
	size_t off = 0;
	for (const auto& pt : parts)
	{
		if (!pt.v->empty()) memcpy(blob.data() + off, pt.v->data(), pt.v->size() * sizeof(int));
		*pt.dst = pt.v->empty() ? nullptr : d_blob + off;
		off += pt.v->size();
	}
	h2d(ctx, d_blob, blob.data(), total * sizeof(int));
	ch.blob = d_blob; ch.blob_ints = total;
}
static void chol_upload_symbolic(lsfm_context* ctx, const CholSymbolic& sym, CholDev& ch)
{
	Arena& sc = ctx->scratch;
	chol_upload_index(ctx, sym, ch);
	chol_alloc_values(ctx, ch);
	ch.d_err = sc.alloc<int>(1);
	dev_zero(ctx, ch.d_err, sizeof(int));
	if (getenv("LSFM_DEBUG") && sym.ngroups > 50)
	{
		int hist[CHOL_GS + 1] = { 0 };
		for (int g = 0; g < sym.ngroups; g++) hist[sym.grp_s[g]]++;
		fprintf(stderr, "[lsfm] groups %d, group levels %d, sizes:", sym.ngroups, (int)sym.glevel_ptr.size() - 1);
		for (int q = 1; q <= CHOL_GS; q++) fprintf(stderr, " %d", hist[q]);
		fprintf(stderr, " | groups per level:");
		for (size_t l = 0; l + 1 < sym.glevel_ptr.size(); l++) fprintf(stderr, " %d", sym.glevel_ptr[l + 1] - sym.glevel_ptr[l]);
		fprintf(stderr, "\n");
	}
}
static void chol_analyse(lsfm_context* ctx, const SchurSystem& sy, const CholHostIn& in, CholDev& ch)
{
	static thread_local CholSymbolic sym;
	chol_symbolic(in.keys.data(), sy.nnzb, in.origin.data(), sy.M, sym, ctx->comm ? ctx->comm->block_maps : 0);
	chol_upload_symbolic(ctx, sym, ch);
}

// ---- one level ahead ---------------------------------------------------------------------------------------------------------
// What a level that analyses needs from the host -- the pattern of its camera system and the symbolic factorisation --
// depends on index arrays only, and the index arrays of level L + 1's joint maps follow from level L's: the joint map of
// a pair is its two maps side by side (pose pairs inside a map: level L's pattern), plus the hub link of every pose of a
// map the transform re-expresses, plus the pairs across the two maps from the features they share.  So while the device
// factors and refines level L, stream3 puts level L + 1's pattern together from level L's joint maps and the host analyses
// it; level L + 1 finds both waiting and enqueues its factorisation right behind its Schur assembly.
struct PreLevel {
	SchurSystem sy;
	CholSymbolic sym;
	CholHostIn hin;          // what the symbolic analysis reads (kept here: it may run on the helper thread)
	int M = 0;
	bool on_worker = false;  // sym is being made by ctx->worker: wait() before it is read
	// the level's whole plan (counts in ctx->pre_plan already): what completes its solve part
	bool whole = false;
	int level = -1, its = 0;
};
static void pre_wait(lsfm_context* ctx, PreLevel* pl)
{
	if (pl && pl->on_worker) { ctx->worker->wait(); pl->on_worker = false; }
}
void prefetch_next_level(lsfm_context* ctx, const DevBatch& Y, const std::vector<int>& target_ref, int next_level, int step_hint)
{
	ctx->drop_prepared();
	static const bool on = !getenv("LSFM_NO_PREFETCH") && !getenv("LSFM_NO_EARLY_PATTERN");
	if (!on || !Y.M || Y.B < 2) return;
	// the next level's systems (pairs of Y's maps): small enough for the one-launch dense path?  Then it needs no pattern and no
	// symbolic factorisation, only -- to be enqueued without a host round trip -- its counts
	int most_next = 0;
	for (int b = 0; b < Y.B; b += 2) most_next = std::max(most_next, Y.pose_off[std::min(b + 2, Y.B)] - Y.pose_off[b]);
	static const bool no_small = getenv("LSFM_NO_SMALL") != nullptr;
	const bool next_small = ctx->small_max > 0 && !ctx->comm && !ctx->pcg.mixed && !no_small && small_solve_strips(most_next, ctx->small_max) > 0;
	// with the step count an earlier run left for that level, the level can run like a planned one (no round trip at all): then
	// its counts are prepared too.  (LSFM_CHECK_EARLY_PATTERN keeps to the path that compares the pattern.)
	static const bool plan_on = !getenv("LSFM_NO_PREPLAN");
	const bool whole = plan_on && step_hint > 0 && !getenv("LSFM_CHECK_EARLY_PATTERN");
	if (next_small && !whole) return; // (nothing to prepare: the level reads its counts back itself)
	ctx->mark("pre_start");
	auto pl = std::make_shared<PreLevel>();
	pl->M = Y.M;
	Arena& sa = ctx->sarena[next_level & 1];
	sa.reset();
	// The joint maps' index arrays are final at evY: the pattern kernels start there, beside the level's right-hand-side kernels and
	// K9.  Measured alternative (LSFM_PREFETCH_LATE=1): start them once K9 has left the main stream (evK), beside the
	// factorisation's chain of small launches -- K9 then runs undisturbed (0.66 -> 0.58 ms per level) but the host gets its
	// pattern 0.6 ms later at every level and the next level is enqueued late: 54.5 instead of 50.4 ms per tree.
	static const bool late_start = getenv("LSFM_PREFETCH_LATE") != nullptr;
	LSFM_CHECK_HIP(hipStreamWaitEvent(ctx->stream3, late_start ? ctx->evK : ctx->evY, 0));
	if (ctx->timeline_on) { (void)hipEventSynchronize(ctx->evY); ctx->mark("pre_evY"); }
	CholHostIn& hin = pl->hin;
	std::vector<int> counts;
	LevelIndex kept;
	bool ok = false;
	struct Swap { // this stretch runs on stream3 and allocates from the small arena of the level's parity
		lsfm_context* c; Arena& a;
		Swap(lsfm_context* x, Arena& y) : c(x), a(y) { std::swap(c->stream, c->stream3); std::swap(c->scratch, a); }
		~Swap() { std::swap(c->scratch, a); std::swap(c->stream, c->stream3); }
	};
	{
		Swap sw(ctx, sa);
		int* d_tref = ctx->scratch.alloc<int>(Y.B);
		h2d(ctx, d_tref, target_ref.data(), sizeof(int) * (size_t)Y.B);
		ok = schur_pattern_prefetch(ctx, Y, d_tref, ctx->solved_keys, ctx->solved_nnzb, pl->sy, whole ? &counts : nullptr, !next_small, whole ? &kept : nullptr);
		if (ok)
		{
			if (!next_small) chol_fetch(ctx, pl->sy, Y.pose_origin, hin); // (synchronises stream3: the counts have arrived too)
			LSFM_CHECK_HIP(hipEventRecord(ctx->evP, ctx->stream));
		}
	}
	ctx->mark("pre_pat");
	if (!ok) return;
	if (next_small)
	{
		// the plan of a small level is its counts: its solve is one launch that asks the host nothing
		const int B = Y.B;
		ctx->pre_plan.tr_cnt.assign(counts.begin(), counts.begin() + 2 * (B + 1));
		ctx->pre_plan.join_rb.assign(counts.begin() + 2 * (B + 1), counts.end());
		ctx->pre_plan.solve.reset();
		ctx->pre_plan.idx = kept;
		ctx->pre_plan.valid = true;
		ctx->pre_plan_level = next_level;
		ctx->mark("pre_plan");
		return;
	}
	// The symbolic factorisation is host work that only the level's FACTORISATION needs: it goes to the helper thread, and the
	// caller enqueues the next level's transform, join and Schur assembly meanwhile -- they need the counts only, which arrived
	// with the pattern.  (Done here, on this thread, the device sat idle 1-3 ms at every level boundary waiting for the next
	// level to be enqueued: 9 of an analysing run's 50 ms.)  LSFM_NO_WORKER=1: on this thread, as before.
	static const bool use_worker = !getenv("LSFM_NO_WORKER");
	if (use_worker)
	{
		if (!ctx->worker) ctx->worker.reset(new HostWorker());
		PreLevel* raw = pl.get(); // (kept alive by ctx->pre / ctx->pre_pending until pre_wait has returned)
		pl->on_worker = true;
		ctx->worker->run([raw]() { chol_symbolic(raw->hin.keys.data(), raw->sy.nnzb, raw->hin.origin.data(), raw->sy.M, raw->sym); });
	}
	else chol_symbolic(hin.keys.data(), pl->sy.nnzb, hin.origin.data(), pl->sy.M, pl->sym);
	ctx->mark("pre_sym");
	if (!whole)
	{
		ctx->pre = pl;
		return;
	}
	// the whole plan of the level: the counts as the host read them now; its solve part (index arrays of the factorisation to the
	// device) is completed by the level's solve_batch -> pre_plan_complete
	pl->whole = true; pl->level = next_level; pl->its = step_hint;
	const int B = Y.B;
	ctx->pre_plan.tr_cnt.assign(counts.begin(), counts.begin() + 2 * (B + 1));
	ctx->pre_plan.join_rb.assign(counts.begin() + 2 * (B + 1), counts.end());
	ctx->pre_plan.solve.reset();
	ctx->pre_plan.idx = kept;
	ctx->pre_plan.valid = true;
	ctx->pre_plan_level = next_level;
	ctx->pre_pending = pl;
	ctx->mark("pre_plan");
}
// the solve part of a plan made one level ahead: waits for the symbolic factorisation, sends its index arrays to the device (stream3,
// the small arena of the level's parity) and makes the main stream wait for them
static std::shared_ptr<void> pre_plan_complete(lsfm_context* ctx)
{
	std::shared_ptr<void> keep = ctx->pre_pending;
	ctx->pre_pending.reset();
	PreLevel* pl = static_cast<PreLevel*>(keep.get());
	pre_wait(ctx, pl);
	auto sp = std::make_shared<SolvePlan>();
	sp->sy = pl->sy;
	sp->its = pl->its; sp->mixed = ctx->pcg.mixed; sp->rel_tol = ctx->pcg.rel_tol;
	Arena& sa = ctx->sarena[pl->level & 1];
	{
		std::swap(ctx->stream, ctx->stream3); std::swap(ctx->scratch, sa);
		try { chol_upload_index(ctx, pl->sym, sp->ch); LSFM_CHECK_HIP(hipEventRecord(ctx->evP, ctx->stream)); }
		catch (...) { std::swap(ctx->scratch, sa); std::swap(ctx->stream, ctx->stream3); throw; }
		std::swap(ctx->scratch, sa); std::swap(ctx->stream, ctx->stream3);
	}
	LSFM_CHECK_HIP(hipStreamWaitEvent(ctx->stream, ctx->evP, 0));
	return sp;
}

// the supernode-group path of the triangular solves applies (chol_apply): the forward substitution can ride on the factorisation
static bool chol_group_solve(const CholDev& ch)
{
	static const bool on = !getenv("LSFM_LEVEL_SOLVE") && !getenv("LSFM_NO_GROUPS") && !getenv("LSFM_TASK_SOLVE") && !getenv("LSFM_NO_FUSED_FWD");
	return on && !ch.tlevel_ptr.empty() && (size_t)ch.tlevel_maxsize[0] * CHOL_TASK_LDS_PER_COL <= 56 * 1024;
}

// fwd_v != null (chol_group_solve(ch) holds): a right-hand side in elimination order; on return it holds what the forward
// substitution leaves (leaf columns in place, group columns in ch.wv) -- chol_apply(..., fwd_done) does the rest
// the scaled, permuted S into the factor's storage; also leaves the scaling (ch.dscale) that k_perm_in / k_perm_out_dot apply:
// before anything is permuted in
// the factorisation of this system is distributed over the ranks of a feature-sharded run (CholDev::col_owner)
static bool chol_distributed(const lsfm_context* ctx, const CholDev& ch)
{
	return ctx->comm && ctx->comm->world > 1 && ch.col_owner && ch.first_shared < ch.M && chol_group_solve(ch) && !getenv("LSFM_NO_GROUPS");
}
// sums `count` 8-byte numbers at p over the ranks (through the caller's buffer: p lives in this context's arenas)
static void comm_sum(lsfm_context* ctx, void* p, size_t count, int dtype)
{
	if (!count) return;
	Comm& cm = *ctx->comm;
	const size_t mk = cm.off;
	void* b = cm.alloc_bytes(count * 8);
	LSFM_CHECK_HIP(hipMemcpyAsync(b, p, count * 8, hipMemcpyDeviceToDevice, ctx->stream));
	cm.allreduce(ctx->stream, b, count, dtype);
	LSFM_CHECK_HIP(hipMemcpyAsync(p, b, count * 8, hipMemcpyDeviceToDevice, ctx->stream));
	cm.off = mk; // (consumed in stream order: the next sum may take the same place)
}

// the scaled, permuted S into the factor's storage; also leaves the scaling (ch.dscale) that k_perm_in / k_perm_out_dot apply:
// before anything is permuted in
static void chol_scatter(lsfm_context* ctx, const SchurSystem& sy, const unsigned char* fixed, CholDev& ch)
{
	static const bool groups = !getenv("LSFM_NO_GROUPS");
	// (the columns above the leaf tasks -- what the supernode groups factor -- accumulate in fixed point; LSFM_NO_GROUPS: none does)
	const int ntask0 = (groups && ch.tlevel_ptr.size() > 1) ? ch.tlevel_ptr[1] : INT_MAX;
	const bool dist = chol_distributed(ctx, ch);
	if (sy.nnzb)
		hipLaunchKernelGGL(k_chol_scatter, dim3((sy.nnzb + 127) / 128), dim3(128), 0, ctx->stream, sy.nnzb, sy.upper_keys, sy.S, sy.rowptr, ch.pinv, ch.colptr, ch.rowidx,
		                   fixed, ch.col_task, ntask0, ch.L, ch.diag0, ch.dscale, dist ? ch.col_owner : (const int*)nullptr, dist ? ctx->comm->rank : 0);
}
// Distributed (chol_distributed): phase 1 -- every rank factors the columns of its own block (leaf sub-trees, then its supernode
// groups level by level), whose updates into the shared separator columns it collects in its own copy of them; then the shared
// columns' accumulators -- 64-bit integers: the sum is exact and the same bits on every rank -- and the shared rows of the forward
// substitution's vector are summed over the ranks; phase 2 -- every rank factors the shared columns, alike.
static void chol_factor(lsfm_context* ctx, const SchurSystem& sy, const unsigned char* fixed, CholDev& ch, double* fwd_v = nullptr)
{
	hipStream_t s = ctx->stream;
	static const bool groups = !getenv("LSFM_NO_GROUPS");
	const bool dist = chol_distributed(ctx, ch);
	const OwnFilter mine{ dist ? ch.col_owner : nullptr, dist ? ctx->comm->rank : 0 }, shared{ dist ? ch.col_owner : nullptr, -1 };
	for (size_t l = 0; l + 1 < ch.tlevel_ptr.size(); l++)
	{
		if (groups && l > 0) break; // the chains above the leaf tasks go by supernode groups below
		const int n = ch.tlevel_ptr[l + 1] - ch.tlevel_ptr[l];
		if (!n) continue;
		static const bool use_small = !getenv("LSFM_NO_SMALL_TASKS");
		const int nsm = use_small ? ch.tlevel_nsmall[l] : 0;
		hipLaunchKernelGGL(k_chol_factor_level, dim3(n), dim3(l ? 256 : 128), nsm ? (size_t)ch.tlevel_small_lds[l] : 0, s, nsm, ch.task_ptr + ch.tlevel_ptr[l],
		                   ch.task_cols, ch.col_nin, ch.colptr, ch.rowidx, ch.L, ch.Dinv, ch.d_err, mine);
		const int c0 = ch.tlevel_col0[l], nc = ch.tlevel_col0[l + 1] - c0, mp = ch.tlevel_outer[l];
		if (mp > 0)
		{
			const dim3 grid(nc, std::min((mp + CHOL_OUT_THREADS - 1) / CHOL_OUT_THREADS, 64));
			if (groups) hipLaunchKernelGGL(k_chol_update_outer<true>, grid, dim3(CHOL_OUT_THREADS), 0, s, ch.task_cols + c0, ch.col_nin, ch.colptr, ch.rowidx, ch.L, mine);
			else hipLaunchKernelGGL(k_chol_update_outer<false>, grid, dim3(CHOL_OUT_THREADS), 0, s, ch.task_cols + c0, ch.col_nin, ch.colptr, ch.rowidx, ch.L, mine);
		}
	}
	if (fwd_v)
	{
		// the leaf sub-trees are factored: their part of the forward substitution, before the groups take theirs
		const int n0 = ch.tlevel_ptr.size() > 1 ? ch.tlevel_ptr[1] - ch.tlevel_ptr[0] : 0;
		const size_t lds0 = (size_t)ch.tlevel_maxsize[0] * CHOL_TASK_LDS_PER_COL + 8;
		if (n0) hipLaunchKernelGGL(k_chol_fwd_tasks<double>, dim3(n0), dim3(128), lds0, s, ch.task_ptr + ch.tlevel_ptr[0], ch.task_cols, ch.col_task, ch.col_lpos, ch.tlevel_ptr[0], ch.colptr, ch.rowidx, (const double*)ch.L, (const double*)ch.Dinv, fwd_v, mine);
	}
	if (groups)
	{
		// one launch per group level while the panels are short enough for the fused kernel (pairs of 8-row chunks: a panel of
		// 64 rows is 36 work-groups per group, each repeating the solve of its two chunks); taller ones take the panel kernel +
		// the rank-update kernel (a synth-16k Mono tree, whose upper levels have panels of 100-300 rows: 684 ms against 826
		// with everything fused)
		static const int fuse_max = getenv("LSFM_SN_FUSE_MAX") ? atoi(getenv("LSFM_SN_FUSE_MAX")) : 96; // (groups of <= 8 columns: 64 -> 96 rows, 9.5 -> 9.2 ms per NC3500 tree; 128 costs synth-16k 143 -> 169 ms)
		static const double piv_floor = getenv("LSFM_PIVOT_FLOOR") ? atof(getenv("LSFM_PIVOT_FLOOR")) : 1e-13; // (0: none)
		static const bool lds_set = []() {
			// (dynamic LDS beyond 64 KB has to be asked for once per kernel)
			(void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_sn_panel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sn_panel_lds(CHOL_GS));
			(void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_sn_panel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sn_panel_lds(CHOL_GS));
			(void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_sn_syrk), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sn_syrk_lds(CHOL_GS));
			return true;
		}();
		(void)lds_set;
		for (int phase = 0; phase < (dist ? 2 : 1); phase++)
		{
			const OwnFilter of = phase == 0 ? mine : shared;
			if (phase == 1)
			{
				comm_sum(ctx, ch.L + (size_t)ch.shared_blk0 * 36, ((size_t)ch.nnzL - ch.shared_blk0) * 36, LSFM_DTYPE_I64);
				if (fwd_v) comm_sum(ctx, fwd_v + (size_t)ch.first_shared * 6, ((size_t)ch.M - ch.first_shared) * 6, LSFM_DTYPE_F64);
			}
			for (size_t l = 0; l + 1 < ch.glevel_ptr.size(); l++)
			{
				const int g0 = ch.glevel_ptr[l], ng = ch.glevel_ptr[l + 1] - g0, mnr = ch.glevel_maxnr[l];
				if (!ng) continue;
				if (dist && !(phase == 0 ? ch.glevel_owned[l] : ch.glevel_shared[l])) continue;
				const int smax = l < ch.glevel_maxs.size() ? ch.glevel_maxs[l] : CHOL_GS;
				if (mnr <= fuse_max)
				{
					const int nch = (mnr + SN_RB / 2 - 1) / (SN_RB / 2);
					hipLaunchKernelGGL(k_sn_panel<true>, dim3(ng, std::max(1, nch * (nch + 1) / 2)), dim3(SN_PT), sn_panel_lds(smax), s, ch.grp_c0 + g0, ch.grp_s + g0, ch.grp_nr + g0,
					                   ch.colptr, ch.L, ch.Lg, ch.Dinv, ch.d_err, ch.rowidx, fwd_v, ch.wv, smax, ch.diag0, piv_floor, ctx->d_run ? &ctx->d_run->floored : nullptr, of);
					continue;
				}
				hipLaunchKernelGGL(k_sn_panel<false>, dim3(ng, std::max(1, (mnr + SN_RB - 1) / SN_RB)), dim3(SN_PT), sn_panel_lds(smax), s, ch.grp_c0 + g0, ch.grp_s + g0, ch.grp_nr + g0,
				                   ch.colptr, ch.L, ch.Lg, ch.Dinv, ch.d_err, ch.rowidx, fwd_v, ch.wv, smax, ch.diag0, piv_floor, ctx->d_run ? &ctx->d_run->floored : nullptr, of);
				static const bool scalar_update = getenv("LSFM_SN_SCALAR_UPDATE") != nullptr; // the round-2 kernel, kept for comparison
				if (scalar_update)
				{
					const long np = (long)mnr * (mnr + 1) / 2;
					hipLaunchKernelGGL(k_sn_update, dim3(ng, (unsigned)std::max<long>(1, std::min<long>((np + SN_PAIRS - 1) / SN_PAIRS, 4096))), dim3(SN_THREADS), 0, s,
					                   ch.grp_c0 + g0, ch.grp_s + g0, ch.grp_nr + g0, ch.colptr, ch.rowidx, ch.L, ch.Lg, of);
				}
				else
				{
					const long nch = (mnr + SN_RB / 2 - 1) / (SN_RB / 2), npair = nch * (nch + 1) / 2;
					hipLaunchKernelGGL(k_sn_syrk, dim3(ng, (unsigned)std::max<long>(1, std::min<long>(npair, 8192))), dim3(SN_THREADS), sn_syrk_lds(smax), s,
					                   ch.grp_c0 + g0, ch.grp_s + g0, ch.grp_nr + g0, ch.colptr, ch.rowidx, ch.L, ch.Lg, smax, of);
				}
			}
		}
		// the solves that walk columns by task or by level read ONE array: give them the group columns there
		if (ch.ngroups && !chol_group_solve(ch))
			hipLaunchKernelGGL(k_sn_merge, dim3(ch.ngroups), dim3(256), 0, s, ch.grp_c0, ch.grp_s, ch.colptr, ch.Lg, ch.L);
	}
}

// z = (L L^T)^-1 r in the original numbering, rz_dot[seg] += r . z
static void chol_perm_in(lsfm_context* ctx, const CholDev& ch, const double* r, const unsigned char* fixed, double* v)
{
	const size_t ns = (size_t)ch.M * 6;
	// (distributed: the shared rows start from the right-hand side on rank 0 only -- they collect the sum of the ranks' parts)
	const int zero_from = (chol_distributed(ctx, ch) && ctx->comm->rank != 0) ? ch.first_shared : INT_MAX;
	hipLaunchKernelGGL(k_perm_in, dim3((unsigned)((ns + 255) / 256)), dim3(256), 0, ctx->stream, ch.M, ch.perm, r, fixed, ch.dscale, v, zero_from);
}

// fwd_done: v already went through the forward substitution (chol_factor with fwd_v).
// Distributed: forward -- own columns, then the shared rows of v summed over the ranks, then the shared columns (alike on every
// rank); backward -- the shared columns, then the own ones; z is the sum of the ranks' parts (z: in the caller's exchange buffer).
static void chol_apply(lsfm_context* ctx, const CholDev& ch, const double* r, double* v, double* z, const unsigned char* fixed, const int* pose_seg,
                       double* dot, int dot_stride, bool fwd_done = false)
{
	hipStream_t s = ctx->stream;
	if (!fwd_done) chol_perm_in(ctx, ch, r, fixed, v);
	static const bool task_solve = !getenv("LSFM_LEVEL_SOLVE");
	int task_max = 0;
	for (int m : ch.tlevel_maxsize) task_max = std::max(task_max, m);
	static const bool group_solve = !getenv("LSFM_NO_GROUPS") && !getenv("LSFM_TASK_SOLVE");
	const OwnFilter all{};
	if (task_solve && group_solve && !ch.tlevel_ptr.empty() && (size_t)ch.tlevel_maxsize[0] * CHOL_TASK_LDS_PER_COL <= 56 * 1024)
	{
		// leaf sub-trees (task level 0) by task, everything above them by supernode group
		const int n0 = ch.tlevel_ptr.size() > 1 ? ch.tlevel_ptr[1] - ch.tlevel_ptr[0] : 0;
		const size_t lds0 = (size_t)ch.tlevel_maxsize[0] * CHOL_TASK_LDS_PER_COL + 8;
		const int ngl = (int)ch.glevel_ptr.size() - 1;
		const bool dist = chol_distributed(ctx, ch);
		const OwnFilter mine{ dist ? ch.col_owner : nullptr, dist ? ctx->comm->rank : 0 }, shared{ dist ? ch.col_owner : nullptr, -1 };
		// (Lx: the leaf columns' factor, in place in L; Gx: the group columns' factor, in its own array)
		auto sweep = [&](auto tag, const auto* Lx, const auto* Gx, const auto* Dx) {
			typedef decltype(tag) FT;
			if (!fwd_done)
			{
				if (n0) hipLaunchKernelGGL(k_chol_fwd_tasks<FT>, dim3(n0), dim3(128), lds0, s, ch.task_ptr + ch.tlevel_ptr[0], ch.task_cols, ch.col_task, ch.col_lpos, ch.tlevel_ptr[0], ch.colptr, ch.rowidx, Lx, Dx, v, mine);
				for (int phase = 0; phase < (dist ? 2 : 1); phase++)
				{
					if (phase == 1) comm_sum(ctx, v + (size_t)ch.first_shared * 6, ((size_t)ch.M - ch.first_shared) * 6, LSFM_DTYPE_F64);
					for (int l = 0; l < ngl; l++)
					{
						const int g0 = ch.glevel_ptr[l], ng = ch.glevel_ptr[l + 1] - g0;
						if (!ng || (dist && !(phase == 0 ? ch.glevel_owned[l] : ch.glevel_shared[l]))) continue;
						hipLaunchKernelGGL(k_sn_fwd<FT>, dim3(ng), dim3(SN_THREADS), 0, s, ch.grp_c0 + g0, ch.grp_s + g0, ch.grp_nr + g0, ch.colptr, ch.rowidx, Gx, Dx, v, ch.wv, phase == 0 ? mine : shared);
					}
				}
			}
			for (int phase = (dist ? 1 : 0); phase >= 0; phase--) // backward: the shared columns first
				for (int l = ngl - 1; l >= 0; l--)
				{
					const int g0 = ch.glevel_ptr[l], ng = ch.glevel_ptr[l + 1] - g0;
					if (!ng || (dist && !(phase == 0 ? ch.glevel_owned[l] : ch.glevel_shared[l]))) continue;
					hipLaunchKernelGGL(k_sn_bwd<FT>, dim3(ng), dim3(SN_THREADS), 0, s, ch.grp_c0 + g0, ch.grp_s + g0, ch.grp_nr + g0, ch.colptr, ch.rowidx, Gx, Dx, v, ch.wv, (dist && phase == 1) ? shared : mine);
				}
			if (n0) hipLaunchKernelGGL(k_chol_bwd_tasks<FT>, dim3(n0), dim3(128), lds0, s, ch.task_ptr + ch.tlevel_ptr[0], ch.task_cols, ch.col_task, ch.col_lpos, ch.tlevel_ptr[0], ch.colptr, ch.rowidx, Lx, Dx, v, mine);
		};
		if (ch.Lf) sweep(float(), (const float*)ch.Lf, (const float*)ch.Lgf, (const float*)ch.Dinvf); // mixed precision: the factor applied in fp32
		else sweep(double(), (const double*)ch.L, (const double*)ch.Lg, (const double*)ch.Dinv);
		if (dist)
		{
			hipLaunchKernelGGL(k_perm_out_dot, dim3((ch.M + 127) / 128), dim3(128), 0, s, ch.M, ch.pinv, v, r, fixed, ch.dscale, pose_seg, z, (double*)nullptr, dot_stride,
			                   ch.col_owner, ctx->comm->rank);
			ctx->comm->allreduce(s, z, (size_t)ch.M * 6, LSFM_DTYPE_F64);
			hipLaunchKernelGGL(k_rz_dot, dim3((ch.M + 127) / 128), dim3(128), 0, s, ch.M, z, r, pose_seg, dot, dot_stride);
		}
		else
			hipLaunchKernelGGL(k_perm_out_dot, dim3((ch.M + 127) / 128), dim3(128), 0, s, ch.M, ch.pinv, v, r, fixed, ch.dscale, pose_seg, z, dot, dot_stride, (const int*)nullptr, 0);
		return;
	}
	if (task_solve && (size_t)task_max * CHOL_TASK_LDS_PER_COL <= 56 * 1024) // a task's per-column data must fit LDS; else one launch per tree level
	{
		const int ntl = (int)ch.tlevel_ptr.size() - 1;
		for (int l = 0; l < ntl; l++)
		{
			const int n = ch.tlevel_ptr[l + 1] - ch.tlevel_ptr[l];
			if (n) hipLaunchKernelGGL(k_chol_fwd_tasks<double>, dim3(n), dim3(l ? 256 : 128), (size_t)ch.tlevel_maxsize[l] * CHOL_TASK_LDS_PER_COL + 8, s, ch.task_ptr + ch.tlevel_ptr[l], ch.task_cols, ch.col_task, ch.col_lpos, ch.tlevel_ptr[l], ch.colptr, ch.rowidx, ch.L, ch.Dinv, v, all);
		}
		for (int l = ntl - 1; l >= 0; l--)
		{
			const int n = ch.tlevel_ptr[l + 1] - ch.tlevel_ptr[l];
			if (n) hipLaunchKernelGGL(k_chol_bwd_tasks<double>, dim3(n), dim3(l ? 256 : 128), (size_t)ch.tlevel_maxsize[l] * CHOL_TASK_LDS_PER_COL + 8, s, ch.task_ptr + ch.tlevel_ptr[l], ch.task_cols, ch.col_task, ch.col_lpos, ch.tlevel_ptr[l], ch.colptr, ch.rowidx, ch.L, ch.Dinv, v, all);
		}
		hipLaunchKernelGGL(k_perm_out_dot, dim3((ch.M + 127) / 128), dim3(128), 0, s, ch.M, ch.pinv, v, r, fixed, ch.dscale, pose_seg, z, dot, dot_stride, (const int*)nullptr, 0);
		return;
	}
	for (int l = 0; l < ch.nlevels; l++)
	{
		const int n = ch.level_ptr[l + 1] - ch.level_ptr[l];
		if (n) hipLaunchKernelGGL(k_chol_fwd_level, dim3(n), dim3(64), 0, s, ch.order + ch.level_ptr[l], ch.colptr, ch.rowidx, ch.L, ch.Dinv, v);
	}
	if (ch.M - ch.tail_begin > 0)
		hipLaunchKernelGGL(k_chol_solve_tail, dim3(1), dim3(256), 0, s, ch.M - ch.tail_begin, ch.order + ch.tail_begin, ch.colptr, ch.rowidx, ch.L, ch.Dinv, v);
	for (int l = ch.nlevels - 1; l >= 0; l--)
	{
		const int n = ch.level_ptr[l + 1] - ch.level_ptr[l];
		if (n) hipLaunchKernelGGL(k_chol_bwd_level, dim3(n), dim3(64), 0, s, ch.order + ch.level_ptr[l], ch.colptr, ch.rowidx, ch.L, ch.Dinv, v);
	}
	hipLaunchKernelGGL(k_perm_out_dot, dim3((ch.M + 127) / 128), dim3(128), 0, s, ch.M, ch.pinv, v, r, fixed, ch.dscale, pose_seg, z, dot, dot_stride, (const int*)nullptr, 0);
}

// ---------------------------------------------------------------------------------------------------------------
// CG pieces
// ---------------------------------------------------------------------------------------------------------------
// (also: the product's accumulator y and the level's counters start from zero -- two fills of their own until round 5)
__global__ void k_x_init(int M, const double* __restrict__ x0, const unsigned char* __restrict__ fixed, double* __restrict__ x, double* __restrict__ y, int* __restrict__ misc)
{
	size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i < 4) misc[i] = 0;
	if (i >= (size_t)M * 6) return;
	y[i] = 0.0;
	double v = x0 ? x0[i] : 0.0;
	if (fixed && fixed[i]) v = 0.0;
	x[i] = v;
}

// r = E - y ; rr += r.r ; ee += E.E   (fixed scalars are not part of the system)
// (y is spent afterwards: it is left zeroed for the next product, which adds into it)
__global__ void k_pcg_resid(int M, const double* __restrict__ E, double* __restrict__ y, const int* __restrict__ pose_seg,
                            const unsigned char* __restrict__ fixed, double* __restrict__ r, PcgSeg* seg, int with_ee)
{
	int row = blockIdx.x * blockDim.x + threadIdx.x;
	bool v = row < M;
	double a[2] = { 0, 0 };
	int sg = 0;
	if (v)
	{
		sg = pose_seg[row];
		for (int i = 0; i < 6; i++)
		{
			const size_t o = (size_t)row * 6 + i;
			double e = E[o], d = e - y[o];
			y[o] = 0.0;
			if (fixed && fixed[o]) { e = 0; d = 0; }
			if (r) r[o] = d;
			a[0] += d * d; a[1] += e * e;
		}
		if (seg[sg].done) v = false; // converged systems are frozen
	}
	if (!with_ee) a[1] = 0.0;
	wave_scatter_add<2>(&seg[sg].rr, a, v); // rr, ee are adjacent
}

// p = z, per system: thresholds, convergence state
// (err / run: a non-positive pivot of the factorisation goes to the run's record here, when the level does not stop to read it)
__global__ void k_pcg_start(int nseg, PcgSeg* seg, const unsigned char* __restrict__ active, double rel_tol, int* ndone, const int* __restrict__ err, RunStatsDev* run)
{
	int s = blockIdx.x * blockDim.x + threadIdx.x;
	if (s == 0 && run && *err && !run->chol_err) run->chol_err = *err;
	if (s >= nseg) return;
	PcgSeg& g = seg[s];
	g.thresh = rel_tol * rel_tol * g.ee;
	g.pAp = 0; g.rz[1] = 0; g.its = 0;
	g.active = active ? active[s] : 1;
	g.done = (!g.active || !(g.rr > g.thresh)) ? 1 : 0;
	g.rr_prev = g.rr;
	// the record of the system's LAST true residual (seg[nseg + s]: what k_pcg_run_stats and the host's verdict read): the starting
	// point's, until a step's test replaces it (k_pcg_check) -- a system that starts below its bound is never touched again
	seg[nseg + s].rr = g.rr; seg[nseg + s].ee = g.ee;
	g.rr = 0;
	if (g.done) atomicAdd(ndone, 1);
}
__global__ void k_copy(size_t n, const double* __restrict__ a, double* __restrict__ b)
{
	size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) b[i] = a[i];
}

// alpha = rz/pAp ; x += alpha p ; y = 0.  The residual is then RECOMPUTED as E - S x (residual replacement in every
// iteration): with the exact factor as preconditioner CG acts as iterative refinement, and the recursively updated
// residual would hide the attainable accuracy (the camera systems reach condition numbers ~1e9).
__global__ void k_pcg_update1(int M, int cur, const int* __restrict__ pose_seg, double* __restrict__ x, const double* __restrict__ p,
                              double* __restrict__ y, const PcgSeg* __restrict__ seg)
{
	int row = blockIdx.x * blockDim.x + threadIdx.x;
	if (row >= M) return;
	const PcgSeg& g = seg[pose_seg[row]];
	for (int i = 0; i < 6; i++) y[(size_t)row * 6 + i] = 0.0;
	if (g.done) return;
	const double alpha = g.rz[cur] / g.pAp;
	for (int i = 0; i < 6; i++) { const size_t o = (size_t)row * 6 + i; x[o] += alpha * p[o]; }
}

// beta = rz[nxt]/rz[cur] ; p = z + beta p ; Ap = 0 ; per system: convergence test, reset accumulators
__global__ void k_pcg_update2(int M, int cur, const double* __restrict__ z, const int* __restrict__ pose_seg, double* __restrict__ p,
                              double* __restrict__ Ap, PcgSeg* seg, int* ndone)
{
	int row = blockIdx.x * blockDim.x + threadIdx.x;
	if (row >= M) return;
	const int sg = pose_seg[row];
	PcgSeg& g = seg[sg];
	const bool done = g.done;
	const double rzn = g.rz[cur ^ 1], rzc = g.rz[cur];
	for (int i = 0; i < 6; i++) Ap[(size_t)row * 6 + i] = 0.0;
	if (!done)
	{
		const double beta = rzn / rzc;
		for (int i = 0; i < 6; i++) { const size_t o = (size_t)row * 6 + i; p[o] = z[o] + beta * p[o]; }
	}
}
// after the residual is known and before the preconditioner is applied to it: convergence test per system
__global__ void k_pcg_check(int nseg, PcgSeg* seg, int* ndone)
{
	int s = blockIdx.x * blockDim.x + threadIdx.x;
	if (s >= nseg) return;
	PcgSeg& g = seg[s];
	if (g.done) return;
	const double rr = g.rr;
	g.its++;
	seg[nseg + s].rr = rr; // (the true residual r = E - S x of the iterate just formed: frozen systems keep the one they froze with)
	// converged; or the true residual stopped shrinking where a direct solve would leave it too (relative 1e-8: attainable
	// accuracy reached); or -- far above that -- three steps in a row that hardly moved it (a system this badly conditioned is
	// reported: the final check counts it as not converged).  One slow step alone does not end the refinement: the camera
	// systems of a deep monocular tree now and then take a step that gains little and go on to 1e-12 with the next.
	// (ten such steps, not three: the residuals of a CG are not monotone, and the top systems of a 16 384-map monocular tree --
	// conditioned ~1e10, factored with sums whose order changes from run to run -- were given up at 3e-8 in one run out of a
	// dozen where a few more steps take them to 1e-11; a system that ends above 1e-8 is a failure anyway, patience costs the
	// others nothing)
	const bool slow = !(rr < 0.25 * g.rr_prev);
	g.slow = slow ? g.slow + 1 : 0;
	if (!(rr > g.thresh) || !(rr == rr) || (slow && (!(rr > 1e-16 * g.ee) || g.slow >= 10))) { g.done = (rr == rr) ? 1 : 2; atomicAdd(ndone, 1); }
	g.rr_prev = rr;
}
// after update2 (separate launch: update2 reads the scalars of its system from every row): reset the accumulators
__global__ void k_pcg_reset(int nseg, int cur, PcgSeg* seg)
{
	int s = blockIdx.x * blockDim.x + threadIdx.x;
	if (s >= nseg) return;
	PcgSeg& g = seg[s];
	if (g.done) return;
	g.pAp = 0; g.rr = 0; g.rz[cur] = 0; // rz[cur] is the accumulator of the next iteration
}

// (a feature-sharded run forms the final residual once more, for the x every rank ends with: its record starts from zero)
__global__ void k_pcg_final_zero(int nseg, PcgSeg* fin)
{
	int s = blockIdx.x * blockDim.x + threadIdx.x;
	if (s < nseg) { fin[s].rr = 0.0; fin[s].ee = 0.0; }
}
// per-system outcome of a level into the run's device accumulators (a warm level does not stop to read them)
__global__ void k_pcg_run_stats(int nseg, const PcgSeg* __restrict__ seg, RunStatsDev* run)
{
	int g = blockIdx.x * blockDim.x + threadIdx.x;
	if (g >= nseg || !seg[g].active) return;
	const PcgSeg& fin = seg[nseg + g];
	const double rel = fin.ee > 0 ? sqrt(fin.rr / fin.ee) : 0.0;
	if (!(rel < 1e-8) || (seg[g].done != 1 && !(rel < 1e-9))) atomicAdd(&run->not_converged, 1);
	// (steps enqueued by a count from an earlier run: a system that had not met its stopping rule when they ran out and is not
	// within two orders of the target either -- the run is repeated asking after every step)
	if (seg[g].done == 0 && !(rel < 1e-10)) atomicAdd(&run->undone, 1);
	// max of non-negative doubles = max of their bit patterns
	atomicMax(reinterpret_cast<unsigned long long*>(&run->max_rel_residual), (unsigned long long)__double_as_longlong(rel == rel ? rel : 1e300));
}
// LSFM_DEBUG_CONV=1: sum and maximum of |a[i]| (what went into a large system and what its factorisation left)
__global__ void k_dbg_absstats(size_t n, const double* __restrict__ a, double* __restrict__ out)
{
	double s = 0.0, m = 0.0;
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
	{
		const double v = fabs(a[i]);
		s += v; if (v > m || !(v == v)) m = v == v ? v : 1e300;
	}
	atomic_add_f64(out, s);
	atomicMax(reinterpret_cast<unsigned long long*>(out + 1), (unsigned long long)__double_as_longlong(m));
}
// LSFM_DEBUG_CONV=1: the columns whose diagonal factor has an inverse beyond 1e3
__global__ void k_dbg_dinv(int M, const double* __restrict__ Dinv, const double* __restrict__ diag0, const int* __restrict__ colptr, int first_group_col)
{
	int j = blockIdx.x * blockDim.x + threadIdx.x;
	if (j >= M) return;
	double m = 0.0;
	for (int q = 0; q < 36; q++) m = fmax(m, fabs(Dinv[(size_t)j * 36 + q]));
	if (m > 1e3 || !(m == m))
		printf("[lsfm conv] column %d (%s, %d blocks): max |Dinv| %.3e, Dinv diag %.3e %.3e %.3e %.3e %.3e %.3e, diag0 %.3e %.3e %.3e %.3e %.3e %.3e\n", j,
		       j >= first_group_col ? "group" : "leaf", colptr[j + 1] - colptr[j], m, Dinv[(size_t)j * 36], Dinv[(size_t)j * 36 + 7], Dinv[(size_t)j * 36 + 14],
		       Dinv[(size_t)j * 36 + 21], Dinv[(size_t)j * 36 + 28], Dinv[(size_t)j * 36 + 35], diag0[j * 6], diag0[j * 6 + 1], diag0[j * 6 + 2], diag0[j * 6 + 3],
		       diag0[j * 6 + 4], diag0[j * 6 + 5]);
}
// LSFM_DEBUG_CONV=1: the systems a level leaves above 1e-9, with the state of their refinement
__global__ void k_pcg_debug(int nseg, int M, const PcgSeg* __restrict__ seg)
{
	int g = blockIdx.x * blockDim.x + threadIdx.x;
	if (g >= nseg || !seg[g].active) return;
	const PcgSeg& fin = seg[nseg + g];
	const double rel = fin.ee > 0 ? sqrt(fin.rr / fin.ee) : 0.0;
	if (!(rel < 1e-9))
		printf("[lsfm conv] M=%d nseg=%d system %d: rel %.3e its %d done %d slow %d rr_prev/ee %.3e thresh/ee %.3e ee %.3e pAp %.3e rz %.3e %.3e\n", M, nseg, g, rel, seg[g].its,
		       seg[g].done, seg[g].slow, seg[g].rr_prev / fin.ee, seg[g].thresh / fin.ee, fin.ee, seg[g].pAp, seg[g].rz[0], seg[g].rz[1]);
}
// LSFM_FACTOR_DIGEST=1: order-independent digest of an array of 8-byte words (a sum modulo 2^64 of position-mixed bit patterns)
__global__ void k_digest(size_t n, const unsigned long long* __restrict__ a, unsigned long long* out)
{
	unsigned long long h = 0;
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
	{
		unsigned long long b = a[i];
		if (b == 0x8000000000000000ull) b = 0; // (-0.0 == 0.0)
		b ^= b >> 31; b *= 0x9E3779B97F4A7C15ull * (2 * (unsigned long long)i + 1); b ^= b >> 29;
		h += b;
	}
	atomicAdd(out, h);
}
__global__ void k_digest_compare(const unsigned long long* d, int* mismatch)
{
	if (d[0] != d[1]) atomicAdd(mismatch, 1);
}
__global__ void k_chol_err_to_run(const int* err, RunStatsDev* run)
{
	if (*err && !run->chol_err) run->chol_err = *err;
}

int solve_batch(lsfm_context* ctx, const SolveIO& io)
{
	hipStream_t s = ctx->stream;
	Arena& sc = ctx->scratch;
	const int M = io.M, nseg = io.nseg;
	LevelPlan* lp = ctx->plan;
	// a plan made one level ahead whose symbolic factorisation may still be under way on the helper thread: the Schur assembly
	// needs the pattern only, so it is enqueued first
	PreLevel* pending = (lp && lp == &ctx->pre_plan && !lp->solve && ctx->pre_pending) ? static_cast<PreLevel*>(ctx->pre_pending.get()) : nullptr;
	if (!pending && ctx->pre_pending) { pre_wait(ctx, static_cast<PreLevel*>(ctx->pre_pending.get())); ctx->pre_pending.reset(); } // (not this level's: dropped)
	// a level of small systems (at most 16 poses each): assembled, factored and solved by one launch (lsfm_small.hip).  The pattern of
	// S and its symbolic analysis are still made -- the levels above build theirs on them, and a plan of the level keeps them
	int most_rows = 0;
	for (int rws : io.seg_rows) most_rows = std::max(most_rows, rws);
	const int strips = (ctx->small_max > 0 && !ctx->comm && !ctx->pcg.mixed && io.d_pose_off && io.d_feat_off && io.d_u_off && !getenv("LSFM_NO_SMALL"))
	                       ? small_solve_strips(most_rows, ctx->small_max) : 0;
	SolvePlan* sp = lp ? static_cast<SolvePlan*>(lp->solve.get()) : nullptr;
	// (a plan recorded on the other path -- lsfm_set_small_solve was changed between two runs of a resident tree -- is void)
	if (sp && sp->small != (strips > 0)) { lp->solve.reset(); sp = nullptr; }
	const bool warm = sp != nullptr || pending != nullptr; // pattern + symbolic factorisation known from an earlier run of the same tree level (or made one level ahead)
	hipEvent_t ea = ctx->pool_event(), eb = ctx->pool_event(), ec = ctx->pool_event(), ed = ctx->pool_event();
	LSFM_REC_T(ea, s); if (roctx().mark) roctx().mark("lsfm schur: begin");
	SchurSystem sy;
	CholDev ch;
	CholHostIn hin;
	const bool dbg = getenv("LSFM_DEBUG") != nullptr;
	auto wall = []() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
	double tw0 = 0, tw1 = 0;
	int* d_err = nullptr;
	int* d_small = nullptr; // [2] status of the small path + (as a double behind them) the level's largest relative residual
	hipEvent_t esm0 = nullptr, esm1 = nullptr;
	auto small_enqueue = [&]() {
		d_small = sc.alloc<int>(4);
		dev_zero(ctx, d_small, 4 * sizeof(int));
		if (ctx->stats) { esm0 = ctx->pool_event(); esm1 = ctx->pool_event(); LSFM_REC_T(esm0, s); }
		small_solve_launch(ctx, io, strips, d_small, reinterpret_cast<double*>(d_small + 2));
		if (ctx->stats) { LSFM_REC_T(esm1, s); ctx->defer_time(esm0, esm1, &ctx->stats->t_small_ms); ctx->stats->small_levels++; }
	};
	if (warm)
	{
		ctx->pattern_dep = false;
		sy = pending ? pending->sy : sp->sy;
		if (!strips)
		{
			schur_vinv(ctx, io, sy);
			build_schur_values(ctx, io, sy);
		}
		LSFM_REC_T(eb, s); if (roctx().mark) roctx().mark("lsfm factor + refine: begin");
		LSFM_CHECK_HIP(hipEventRecord(ctx->evK, s));
		if (pending)
		{
			ctx->mark("k9_enq");
			lp->solve = pre_plan_complete(ctx);
			sp = static_cast<SolvePlan*>(lp->solve.get());
			ctx->mark("sym_wait");
		}
		ch = sp->ch;
		if (!strips)
		{
			chol_alloc_values(ctx, ch);
			d_err = ch.d_err = sc.alloc<int>(1);
			dev_zero(ctx, d_err, sizeof(int));
		}
	}
	else
	{
		// pattern -> (copy it to the host) -> numeric assembly K9 enqueued -> symbolic factorisation on the host while K9
		// runs -> numeric factorisation
		// The pattern depends on index arrays only.  When the caller marked the point of the main stream where those were
		// complete (evA) and went on to enqueue its right-hand-side kernels, the pattern is built on the side stream next
		// to them; the values wait for both.
		static const bool side = !getenv("LSFM_NO_SIDE_STREAM");
		if (strips)
		{
			// no pattern of S, no symbolic factorisation: the dense path needs neither, and the level above builds its pattern from
			// its own joint maps when this one leaves none (schur_pattern_prefetch / k_pat_insert_w_early)
			ctx->pattern_dep = false;
			schur_pattern_early_drop(ctx);
			if (ctx->pre) { std::shared_ptr<void> keep = ctx->pre; ctx->pre.reset(); pre_wait(ctx, static_cast<PreLevel*>(keep.get())); }
			small_enqueue();
			LSFM_REC_T(eb, s);
			LSFM_CHECK_HIP(hipEventRecord(ctx->evK, s));
			goto small_tail;
		}
		schur_vinv(ctx, io, sy);
		bool have = false;
		std::shared_ptr<void> pre_keep = ctx->pre;
		ctx->pre.reset();
		PreLevel* pre = static_cast<PreLevel*>(pre_keep.get());
		// (drop_prepared() no longer sees the object: whatever throws below, it must not go while the helper thread still works on it)
		struct PreGuard { lsfm_context* c; PreLevel* p; ~PreGuard() { try { pre_wait(c, p); } catch (...) {} } } pre_guard{ ctx, pre };
		if (pre && !(pre->M == M && !ctx->comm)) pre_wait(ctx, pre); // (not used: nothing of it may go while the helper thread reads it)
		if (pre && pre->M == M && !ctx->comm)
		{
			// prepared while the level below was being solved: pattern (device) and symbolic factorisation (host)
			ctx->pattern_dep = false;
			schur_pattern_early_drop(ctx);
			{
				SchurSystem prepared = pre->sy; // the index members; V^-1 and its factor are this level's (schur_vinv above)
				prepared.IV = sy.IV; prepared.LY = sy.LY; prepared.ymax = sy.ymax; prepared.uu = sy.uu;
				sy = prepared;
			}
			have = true;
			LSFM_CHECK_HIP(hipStreamWaitEvent(s, ctx->evP, 0));
			if (getenv("LSFM_CHECK_EARLY_PATTERN"))
			{
				LSFM_CHECK_HIP(hipStreamSynchronize(s));
				SchurSystem ref;
				build_schur_pattern(ctx, io, ref);
				std::vector<unsigned long long> a(sy.nnzb), b(ref.nnzb);
				d2h(ctx, a.data(), sy.upper_keys, a.size() * sizeof(unsigned long long));
				d2h(ctx, b.data(), ref.upper_keys, b.size() * sizeof(unsigned long long));
				if (a != b) LSFM_FAIL(LSFM_ERR_INTERNAL, "prefetched pattern of S (" + std::to_string(a.size()) + " blocks) differs from the joint map's (" + std::to_string(b.size()) + ")");
			}
		}
		else pre = nullptr;
		if (have) {}
		else if (ctx->early && !ctx->comm)
		{
			// the pattern was put together on the side stream from the level's inputs while the transform ran (a Stereo level
			// that analyses): its second half, and the copy of it for the host's analysis, stay there
			ctx->pattern_dep = false;
			LSFM_CHECK_HIP(hipStreamWaitEvent(ctx->stream3, ctx->evC, 0)); // (recorded again once the joint run pointers were enqueued)
			std::swap(ctx->stream, ctx->stream3);
			ctx->mark("sv_start");
			try
			{
				have = schur_pattern_early_finish(ctx, io, sy);
				ctx->mark("pat_fin");
				if (have)
				{
					if (getenv("LSFM_CHECK_EARLY_PATTERN"))
					{
						// debug / test: the pattern built from the finished joint map must be the same one
						LSFM_CHECK_HIP(hipStreamSynchronize(ctx->stream3)); // (the main stream, swapped out: the joint map's index arrays)
						SchurSystem ref;
						build_schur_pattern(ctx, io, ref);
						std::vector<unsigned long long> a(sy.nnzb), b(ref.nnzb);
						d2h(ctx, a.data(), sy.upper_keys, a.size() * sizeof(unsigned long long));
						d2h(ctx, b.data(), ref.upper_keys, b.size() * sizeof(unsigned long long));
						if (a != b) LSFM_FAIL(LSFM_ERR_INTERNAL, "early pattern of S (" + std::to_string(a.size()) + " blocks) differs from the joint map's (" + std::to_string(b.size()) + ")");
					}
					chol_fetch(ctx, sy, io.d_pose_origin, hin);
					ctx->mark("fetch");
					schur_pattern_early_extras(ctx, io, sy);
					LSFM_CHECK_HIP(hipEventRecord(ctx->evB, ctx->stream));
				}
			}
			catch (...) { std::swap(ctx->stream, ctx->stream3); throw; }
			std::swap(ctx->stream, ctx->stream3);
			if (have) LSFM_CHECK_HIP(hipStreamWaitEvent(s, ctx->evB, 0));
		}
		if (have) {}
		else if (side && ctx->pattern_dep && !ctx->comm)
		{
			ctx->pattern_dep = false;
			LSFM_CHECK_HIP(hipStreamWaitEvent(ctx->stream2, ctx->evA, 0));
			std::swap(ctx->stream, ctx->stream2);
			try
			{
				build_schur_pattern(ctx, io, sy);
				chol_fetch(ctx, sy, io.d_pose_origin, hin);
				LSFM_CHECK_HIP(hipEventRecord(ctx->evB, ctx->stream));
			}
			catch (...) { std::swap(ctx->stream, ctx->stream2); throw; }
			std::swap(ctx->stream, ctx->stream2);
			LSFM_CHECK_HIP(hipStreamWaitEvent(s, ctx->evB, 0));
		}
		else
		{
			ctx->pattern_dep = false;
			build_schur_pattern(ctx, io, sy);
			chol_fetch(ctx, sy, io.d_pose_origin, hin);
		}
		build_schur_values(ctx, io, sy);
		LSFM_REC_T(eb, s); if (roctx().mark) roctx().mark("lsfm factor + refine: begin");
		LSFM_CHECK_HIP(hipEventRecord(ctx->evK, s));
		tw0 = wall();
		ctx->mark("k9_enq");
		if (pre) { pre_wait(ctx, pre); chol_upload_symbolic(ctx, pre->sym, ch); }
		else chol_analyse(ctx, sy, hin, ch);
		ctx->mark("analyse");
		tw1 = wall();
		d_err = ch.d_err;
	}
small_tail:
	if (strips)
	{
		if (warm) small_enqueue();
		LSFM_REC_T(ec, s);
		LSFM_REC_T(ed, s); if (roctx().mark) roctx().mark("lsfm solve: end");
		ctx->ev_solve_end = ed;
		ctx->solved_keys = nullptr; ctx->solved_nnzb = 0; // (no pattern left for the level above)
		if (ctx->stats) ctx->stats->pcg_iterations += 1;
		ctx->steps_used = 1;
		// (a plan made one level ahead is the run's own: nothing to record, nothing to stop for)
		const bool deferred = ctx->in_tree_run && ctx->d_run && (warm || !lp || lp == &ctx->pre_plan);
		if (deferred) return 0; // the kernel left its verdict in the run's device record (read at the end of the run)
		int hs[4];
		d2h_ints(ctx, d_small, hs, 4); // synchronises
		if (hs[1]) LSFM_FAIL(LSFM_ERR_NOT_SPD, "Schur system is not positive definite (system " + std::to_string(hs[1] - 1) + " of the level)");
		double mr;
		memcpy(&mr, hs + 2, sizeof mr);
		if (ctx->stats) ctx->stats->max_rel_residual = std::max(ctx->stats->max_rel_residual, mr);
		if (lp && !lp->solve && hs[0] == 0)
		{
			// the plan of a small level: nothing but the fact that it is one (the structure of its solve is the batch's offsets)
			auto small_plan = std::make_shared<SolvePlan>();
			small_plan->its = 1; small_plan->mixed = false; small_plan->rel_tol = ctx->pcg.rel_tol; small_plan->small = true;
			lp->solve = small_plan;
		}
		return hs[0];
	}
	// ---- CG set-up first: the residual of the starting point is the right-hand side of the first preconditioner
	// application, whose forward substitution rides on the factorisation (k_sn_panel) ----
	int* d_misc = sc.alloc<int>(4); // [1] ndone (zeroed by k_x_init)
	std::vector<PcgSeg> hseg(nseg);
	{
		int row = 0;
		for (int g = 0; g < nseg; g++) { memset(&hseg[g], 0, sizeof(PcgSeg)); hseg[g].row0 = row; row += io.seg_rows[g]; }
	}
	PcgSeg* seg = sc.alloc<PcgSeg>(2 * (size_t)nseg); // [nseg, 2 nseg): accumulators of the final residual check
	hseg.resize(2 * (size_t)nseg);
	std::copy(hseg.begin(), hseg.begin() + nseg, hseg.begin() + nseg);
	h2d(ctx, seg, hseg.data(), sizeof(PcgSeg) * 2 * nseg);
	double* x = io.x_pose;
	const size_t nscal = (size_t)M * 6;
	// (feature-sharded run: z is a sum over the ranks when the factorisation is distributed -- it lives in the exchange buffer)
	double* r = sc.alloc<double>(nscal); double* z = ctx->comm ? ctx->comm->alloc<double>(nscal) : sc.alloc<double>(nscal); double* p = sc.alloc<double>(nscal);
	double* Ap = sc.alloc<double>(nscal); double* v = sc.alloc<double>(nscal);
	const int nbr = (M + 127) / 128, nbs = (nseg + 127) / 128;
	const unsigned nbe = (unsigned)((nscal + 255) / 256);
	hipLaunchKernelGGL(k_x_init, dim3(std::max(1u, nbe)), dim3(256), 0, s, M, io.x0, io.d_fixed, x, Ap, d_misc);
	launch_spmv(ctx, sy, x, Ap, io.d_fixed, nullptr, nullptr, nullptr, 1);
	hipLaunchKernelGGL(k_pcg_resid, dim3(nbr), dim3(128), 0, s, M, sy.E, Ap, io.d_pose_seg, io.d_fixed, r, seg, 1);
	const bool mixed = ctx->pcg.mixed;
	const bool fused_fwd = !mixed && chol_group_solve(ch); // (mixed: the factor is applied from its fp32 copy, made after the factorisation)
	if (ctx->stats && chol_distributed(ctx, ch)) { ctx->stats->dist_solves++; ctx->stats->dist_work_total += ch.work_total; ctx->stats->dist_work_shared += ch.work_shared; }
	chol_scatter(ctx, sy, io.d_fixed, ch);
	if (fused_fwd) chol_perm_in(ctx, ch, r, io.d_fixed, v);
	chol_factor(ctx, sy, io.d_fixed, ch, fused_fwd ? v : nullptr);
	if (mixed)
	{
		// mixed precision (BASELINE configs[4]): the factor is rounded to fp32 once and applied from there; S, E, x and the
		// residual stay fp64 -- every refinement step corrects against r = E - S x in fp64
		const size_t nl = (size_t)ch.nnzL * 36, nd = (size_t)ch.M * 36;
		ch.Lf = sc.alloc<float>(nl); ch.Dinvf = sc.alloc<float>(nd);
		hipLaunchKernelGGL(k_to_float, dim3((unsigned)((nl + 255) / 256)), dim3(256), 0, s, nl, ch.L, ch.Lf);
		if (ch.Lg)
		{
			ch.Lgf = sc.alloc<float>(nl);
			hipLaunchKernelGGL(k_to_float, dim3((unsigned)((nl + 255) / 256)), dim3(256), 0, s, nl, ch.Lg, ch.Lgf);
		}
		hipLaunchKernelGGL(k_to_float, dim3((unsigned)((nd + 255) / 256)), dim3(256), 0, s, nd, ch.Dinv, ch.Dinvf);
	}
	static const bool digest = getenv("LSFM_FACTOR_DIGEST") != nullptr;
	if (digest && ctx->d_run)
	{
		auto dg = [&](const double* a, size_t n, unsigned long long* out) {
			if (a && n) hipLaunchKernelGGL(k_digest, dim3(256), dim3(256), 0, s, n, reinterpret_cast<const unsigned long long*>(a), out);
		};
		dg(sy.S, (size_t)sy.nnzb * 36, &ctx->d_run->s_digest);
		if (!ctx->comm)
		{
			// the SAME camera systems assembled a second time (U scatter, K9 with all its variants, the fallback kernel): their
			// work-groups land their sums in another order -- the bits of S and E must not depend on it (fixed-point sums,
			// lsfm_schur_panel.hip).  (The stage timings and flop counts of the run count this second assembly too: a debug mode.)
			SchurSystem sy2 = sy;
			build_schur_values(ctx, io, sy2);
			unsigned long long* d = sc.alloc<unsigned long long>(2);
			dev_zero(ctx, d, 2 * sizeof(unsigned long long));
			dg(sy.S, (size_t)sy.nnzb * 36, d); dg(sy.E, (size_t)M * 6, d);
			dg(sy2.S, (size_t)sy.nnzb * 36, d + 1); dg(sy2.E, (size_t)M * 6, d + 1);
			hipLaunchKernelGGL(k_digest_compare, dim3(1), dim3(1), 0, s, d, &ctx->d_run->s_rebuild_mismatch);
		}
		// leaf columns: factored in place in L; group columns: in Lg (their slots of L hold the spent accumulators: integers, summed alike)
		auto factor_digest = [&](unsigned long long* out) {
			dg(ch.Dinv, (size_t)ch.M * 36, out);
			dg(ch.L, (size_t)ch.nnzL * 36, out);
			if (ch.Lg) dg(ch.Lg, (size_t)ch.nnzL * 36, out);
		};
		factor_digest(&ctx->d_run->factor_digest);
		if (!ctx->comm)
		{
			// ... and the SAME system factored a second time (its work-groups will be scheduled differently, the atomics land in another
			// order): the two factors must be the same bits.  d[0], d[1]: the digests of this system's two factors alone
			unsigned long long* d = sc.alloc<unsigned long long>(2);
			dev_zero(ctx, d, 2 * sizeof(unsigned long long));
			factor_digest(d);
			dev_zero(ctx, ch.L, (size_t)ch.nnzL * 36 * sizeof(double));
			if (ch.Lg) dev_zero(ctx, ch.Lg, (size_t)ch.nnzL * 36 * sizeof(double));
			chol_scatter(ctx, sy, io.d_fixed, ch);
			chol_factor(ctx, sy, io.d_fixed, ch, nullptr);
			factor_digest(d + 1);
			hipLaunchKernelGGL(k_digest_compare, dim3(1), dim3(1), 0, s, d, &ctx->d_run->refactor_mismatch);
		}
	}
	if (dbg) { LSFM_CHECK_HIP(hipStreamSynchronize(s)); }
	double tw2 = wall();
	chol_apply(ctx, ch, r, v, z, io.d_fixed, io.d_pose_seg, &seg[0].rz[0], SEG_STRIDE, fused_fwd);
	hipLaunchKernelGGL(k_copy, dim3(nbe), dim3(256), 0, s, nscal, z, p);
	// (Ap is zero again: k_pcg_resid leaves it so)
	// inside a tree run the outcome of a level (a non-positive pivot, systems left above their bound, the largest residual) is
	// left in the run's device record and read once at the end of the run; a stage-level call, and a level whose structure is
	// being recorded as a plan, reads it here
	const bool deferred = warm || (ctx->in_tree_run && ctx->d_run && !lp);
	ctx->solved_keys = sy.upper_keys; ctx->solved_nnzb = sy.nnzb;
	auto check_factor = [&]() {
		const int cerr = d2h_int(ctx, d_err);
		if (cerr) LSFM_FAIL(LSFM_ERR_NOT_SPD, "Schur system is not positive definite (block column " + std::to_string(cerr - 1) + " of the factor)");
	};
	// (a feature-sharded run never throws for it in the middle of a pass: the ranks' factorisations are their own, and a rank that left
	// the pass alone would leave its peers in a sum it never joins -- the flags are exchanged at the end of the run)
	const bool err_to_run = deferred || (ctx->comm && ctx->d_run);
	hipLaunchKernelGGL(k_pcg_start, dim3(nbs), dim3(128), 0, s, nseg, seg, io.d_seg_active, ctx->pcg.rel_tol, d_misc + 1, d_err, err_to_run ? ctx->d_run : (RunStatsDev*)nullptr); // (a bad pivot: reported at the end of the run)

	// One refinement step: x += alpha p, true residual, convergence test per system (converged systems freeze), then the
	// preconditioner for the next step.  A first run reads the number of finished systems back after every step; a warm
	// run enqueues the steps the first run needed -- the device-side tests still freeze what is done, and whether every
	// system ended below its bound is read once at the end of the whole run.
	const int maxit = std::max(1, std::min(50, ctx->pcg.max_steps));
	// the step count was recorded with the preconditioner in this precision, for this tolerance, under this cap
	// ... or, in a run that analyses, what an earlier run of the same tree needed at this level (a guess about values, checked at
	// the end of the run like a plan's count)
	const bool hinted = !warm && deferred && ctx->step_hint > 0 && ctx->step_hint <= maxit;
	const bool planned_run = hinted || (warm && sp->mixed == mixed && sp->rel_tol == ctx->pcg.rel_tol && sp->its <= maxit);
	// (a run that counts its steps does not stop to ask before the first one either: systems that start below their bound
	// are frozen on the device, the step costs them nothing)
	int its = 0, ndone = 0;
	hipEvent_t es0 = ctx->pool_event(), es1 = ctx->pool_event(); // around one product S x of the refinement (lsfm_stats.spmv_ms)
	bool es_done = false;
	// (a level that needed two steps or more -- three with the fp32 preconditioner, where two is the rule -- is ill-conditioned enough
	// for its count to vary from run to run -- one synth-16k run in
	// eight asked for one more than the run before and had to be repeated as a whole: such levels get one step of margin; systems
	// that are done are frozen on the device, the extra step costs them the launches only)
	// A hinted level whose caller waits for the device at its end anyway (ctx->level_syncs: a Mono level that analyses) needs neither the
	// margin nor the repeat: it enqueues the steps the run before needed, asks ONCE whether every system is done, and goes on asking
	// step by step if not -- one synth-16k analysing run in eight to twenty was repeated as a whole (twice its time) until round 6.
	const bool ask_after = hinted && ctx->level_syncs && !ctx->comm;
	const int base_steps = hinted ? ctx->step_hint : (planned_run ? sp->its : maxit);
	const int planned = (planned_run && !ask_after && base_steps >= (mixed ? 3 : 2)) ? std::min(base_steps + 1, maxit) : base_steps;
	bool counting = planned_run; // (the steps are enqueued without asking)
	bool extended = false;
	// (an extension is for the run that needs ONE step more than the run before -- the counts scatter by one or two; a system that is
	// still not done three steps on has stalled where its true residual stops shrinking, and steps do not cure that: the run is
	// joined again, as before, which does -- the rounding falls differently.  Nor does an extension raise the hint: it would stay
	// raised, and every later run would pay for the one that stalled)
	int cap = maxit;
	while ((counting ? its < planned : (ndone < nseg && its < cap)))
	{
		const int cur = its & 1;
		launch_spmv(ctx, sy, p, Ap, io.d_fixed, p, io.d_pose_seg, &seg[0].pAp, SEG_STRIDE);
		hipLaunchKernelGGL(k_pcg_update1, dim3(nbr), dim3(128), 0, s, M, cur, io.d_pose_seg, x, p, Ap, seg);
		if (!es_done) LSFM_REC_T(es0, s);
		launch_spmv(ctx, sy, x, Ap, io.d_fixed, nullptr, nullptr, nullptr, 1);
		if (!es_done) { LSFM_REC_T(es1, s); es_done = true; }
		hipLaunchKernelGGL(k_pcg_resid, dim3(nbr), dim3(128), 0, s, M, sy.E, Ap, io.d_pose_seg, io.d_fixed, r, seg, 0);
		// the test comes before the preconditioner: the apply for a residual that already passed would be wasted
		hipLaunchKernelGGL(k_pcg_check, dim3(nbs), dim3(128), 0, s, nseg, seg, d_misc + 1);
		its++;
		if (counting)
		{
			if (its >= planned)
			{
				if (!ask_after) break;
				ndone = d2h_int(ctx, d_misc + 1);
				if (ndone >= nseg || its >= maxit) break;
				counting = false; // (a system needs more than the run before did: from here on like a run without a hint)
				extended = true;
				cap = std::min(maxit, planned + 3);
			}
		}
		else
		{
			ctx->mark("cg_enq");
			ndone = d2h_int(ctx, d_misc + 1);
			ctx->mark("cg_sync");
			if (its == 1 && !err_to_run) check_factor(); // (the stream is drained: this costs no second wait)
			if (ctx->comm && ctx->comm->world > 1)
			{
				// feature-sharded run: whether another step follows must be the same answer on every rank -- the next step holds sums
				// over the ranks when the factorisation is distributed (chol_apply), and the ranks' residuals, taken from floating-point
				// atomic sums, may differ in the last bits: at a threshold one rank would leave the loop for the sum of x while another
				// enters the sums of the preconditioner (advisor, round 4).  Any rank's doubt is everybody's: one 8-byte sum per step,
				// in runs that ask after every step only.
				Comm& cm = *ctx->comm;
				const size_t mk = cm.off;
				long long* d_more = cm.alloc<long long>(1);
				long long more = ndone >= nseg ? 0 : 1;
				LSFM_CHECK_HIP(hipMemcpyAsync(d_more, &more, sizeof more, hipMemcpyHostToDevice, s));
				LSFM_CHECK_HIP(hipStreamSynchronize(s));
				cm.allreduce(s, d_more, 1, LSFM_DTYPE_I64);
				LSFM_CHECK_HIP(hipMemcpyAsync(&more, d_more, sizeof more, hipMemcpyDeviceToHost, s));
				LSFM_CHECK_HIP(hipStreamSynchronize(s));
				cm.off = mk;
				if (!more) break;
				ndone = std::min(ndone, nseg - 1); // (a peer goes on: so does this rank -- its finished systems are frozen on the device)
			}
			else if (ndone >= nseg) break;
		}
		chol_apply(ctx, ch, r, v, z, io.d_fixed, io.d_pose_seg, &seg[0].rz[cur ^ 1], SEG_STRIDE);
		hipLaunchKernelGGL(k_pcg_update2, dim3(nbr), dim3(128), 0, s, M, cur, z, io.d_pose_seg, p, Ap, seg, d_misc + 1);
		hipLaunchKernelGGL(k_pcg_reset, dim3(nbs), dim3(128), 0, s, nseg, cur, seg);
	}
	if (dbg)
	{
		LSFM_CHECK_HIP(hipStreamSynchronize(s));
		fprintf(stderr, "[lsfm] solve M=%d nseg=%d nnzb=%d nnzL=%d etree levels=%d tail=%d task levels=%d group levels=%d %s| analyse %.2f ms, factor %.2f ms, cg(%d its) %.2f ms\n", M, nseg,
		        sy.nnzb, ch.nnzL, ch.nlevels, ch.M - ch.tail_begin, (int)ch.tlevel_ptr.size() - 1, (int)ch.glevel_ptr.size() - 1, warm ? "(plan) " : "", tw1 - tw0, tw2 - tw1, its, wall() - tw2);
	}
	if (ctx->comm)
	{
		// feature-sharded run: every rank solved the same system, but the factorisations add their updates in whatever order the
		// atomics land -- the solutions may differ in the last bit.  Rank 0's replaces everyone's, so that the replicated state
		// (and every decision taken from it) stays the same on all ranks.
		Comm& cm = *ctx->comm;
		double* xb = cm.alloc<double>(nscal);
		if (cm.rank == 0) LSFM_CHECK_HIP(hipMemcpyAsync(xb, x, nscal * sizeof(double), hipMemcpyDeviceToDevice, s));
		else fill_async(s, xb, 0, nscal * sizeof(double));
		cm.allreduce(s, xb, nscal, LSFM_DTYPE_F64);
		LSFM_CHECK_HIP(hipMemcpyAsync(x, xb, nscal * sizeof(double), hipMemcpyDeviceToDevice, s));
	}
	// ---- true residual, statistics.  Every step's test has left the system's last true residual r = E - S x in seg[nseg + g]
	// (k_pcg_start / k_pcg_check): the product and the residual kernel that formed it once more behind the loop (until round 6) are
	// gone -- except in a feature-sharded run, whose x has just been replaced by rank 0's.  One of the loop's products is timed with
	// HIP events (es0, es1).  Nothing here waits for the device before the back-substitution is enqueued ----
	const int nsample = 1;
	static const bool final_again = getenv("LSFM_FINAL_RESIDUAL") != nullptr; // (as until round 6: for comparison)
	if (ctx->comm || !es_done || final_again)
	{
		PcgSeg* seg2 = seg + nseg;
		if (!es_done) LSFM_REC_T(es0, s);
		launch_spmv(ctx, sy, x, Ap, io.d_fixed, nullptr, nullptr, nullptr, 1);
		if (!es_done) { LSFM_REC_T(es1, s); es_done = true; }
		hipLaunchKernelGGL(k_pcg_final_zero, dim3(nbs), dim3(128), 0, s, nseg, seg2);
		hipLaunchKernelGGL(k_pcg_resid, dim3(nbr), dim3(128), 0, s, M, sy.E, Ap, io.d_pose_seg, io.d_fixed, (double*)nullptr, seg2, 1);
	}
	LSFM_REC_T(ec, s); if (roctx().mark) roctx().mark("lsfm back-substitution: begin");
	launch_backsub(ctx, io, sy, x);
	LSFM_CHECK_HIP(hipGetLastError());
	LSFM_REC_T(ed, s); if (roctx().mark) roctx().mark("lsfm solve: end");
	ctx->ev_solve_end = ed;
	if (ctx->stats)
	{
		lsfm_stats* st = ctx->stats;
		ctx->defer_time(ea, eb, &st->t_schur_ms);
		ctx->defer_time(eb, ec, &st->t_pcg_ms);
		ctx->defer_time(ec, ed, &st->t_backsub_ms);
		ctx->defer_time(es0, es1, &st->spmv_ms);
		st->pcg_iterations += its;
		st->spmv_launches += nsample;
		st->spmv_bytes += nsample * spmv_bytes(sy);
		st->spmv_nnzb_upper_last = sy.nnzb; st->spmv_rows_last = M;
	}
	ctx->steps_used = planned_run ? 0 : std::max(its, 1);
	(void)extended;
	if (deferred)
	{
		if (warm && !planned_run) { sp->its = std::max(its, 1); sp->mixed = mixed; sp->rel_tol = ctx->pcg.rel_tol; } // precision / tolerance changed: the count was re-learnt
		static const bool dbg_conv = getenv("LSFM_DEBUG_CONV") != nullptr;
		if (dbg_conv) hipLaunchKernelGGL(k_pcg_debug, dim3(nbs), dim3(128), 0, s, nseg, M, seg);
		if (dbg_conv && M > 10000 && nseg == 1)
		{
			double* d = sc.alloc<double>(12);
			dev_zero(ctx, d, 12 * sizeof(double));
			auto st = [&](const double* a, size_t n, int k) { if (a && n) hipLaunchKernelGGL(k_dbg_absstats, dim3(512), dim3(256), 0, s, n, a, d + 2 * k); };
			st(sy.S, (size_t)sy.nnzb * 36, 0); st(sy.E, (size_t)M * 6, 1); st(ch.L, (size_t)ch.nnzL * 36, 2); st(ch.Lg, (size_t)ch.nnzL * 36, 3);
			st(ch.Dinv, (size_t)ch.M * 36, 4); st(x, nscal, 5);
			hipLaunchKernelGGL(k_dbg_dinv, dim3((ch.M + 255) / 256), dim3(256), 0, s, ch.M, ch.Dinv, ch.diag0, ch.colptr, 0);
			double h[12];
			d2h(ctx, h, d, sizeof h);
			fprintf(stderr, "[lsfm conv] root M=%d: |S| sum %.15e max %.6e  |E| sum %.15e  |L| sum %.12e max %.3e  |Lg| sum %.12e max %.3e  |Dinv| sum %.6e max %.3e  |x| sum %.12e max %.3e\n",
			        M, h[0], h[1], h[2], h[4], h[5], h[6], h[7], h[8], h[9], h[10], h[11]);
		}
		hipLaunchKernelGGL(k_pcg_run_stats, dim3(nbs), dim3(128), 0, s, nseg, seg, ctx->d_run);
		return 0; // the outcome is read at the end of the run (lsfm_tree_run)
	}
	std::vector<PcgSeg> hs2(2 * (size_t)nseg);
	d2h(ctx, hs2.data(), seg, sizeof(PcgSeg) * 2 * nseg); // synchronises
	int notconv = 0;
	double maxrel = 0;
	for (int g = 0; g < nseg; g++)
	{
		if (!hs2[g].active) continue;
		const PcgSeg& fin = hs2[nseg + g];
		const double rel = fin.ee > 0 ? sqrt(fin.rr / fin.ee) : 0.0;
		maxrel = std::max(maxrel, rel);
		// converged = stopped by the tolerance, or stopped by stagnation with a residual a direct solve would also leave
		if (!(rel < 1e-8) || (hs2[g].done != 1 && !(rel < 1e-9))) notconv++;
	}
	if (ctx->stats) ctx->stats->max_rel_residual = std::max(ctx->stats->max_rel_residual, maxrel);
	if (ctx->comm)
	{
		// feature-sharded run: the verdict (and with it whether this level keeps a plan, i.e. whether the NEXT run of the level is
		// warm) must be the same on every rank -- a rank that is cold alone would issue pattern all-reduces nobody joins.  Every
		// rank solved the same system; the residuals they computed differ in the last bit at most, but a count taken at a
		// threshold may: summed over the ranks, any rank's doubt is everybody's.
		Comm& cm = *ctx->comm;
		long long* d_v = cm.alloc<long long>(2);
		long long hv[2] = { notconv, 0 };
		LSFM_CHECK_HIP(hipMemcpyAsync(d_v, hv, sizeof hv, hipMemcpyHostToDevice, s));
		LSFM_CHECK_HIP(hipStreamSynchronize(s));
		cm.allreduce(s, d_v, 2, LSFM_DTYPE_I64);
		LSFM_CHECK_HIP(hipMemcpyAsync(hv, d_v, sizeof hv, hipMemcpyDeviceToHost, s));
		LSFM_CHECK_HIP(hipStreamSynchronize(s));
		notconv = (int)((hv[0] + cm.world - 1) / cm.world);
	}
	// what depends on the structure only stays with the tree level for its next runs
	if (lp && !lp->solve && notconv == 0) lp->solve = solve_plan_store(ctx, sy, ch, std::max(its, 1));
	return notconv;
}

} // namespace lsfm
