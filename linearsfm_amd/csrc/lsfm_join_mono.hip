// Batched pairwise join, Monocular: replaces lmj_LinearLS_PF3DMono (Imp.cpp:7282-7874) for all pairs of a tree level.
// Differences from the Stereo join (lsfm_join.hip):
//   * the two maps share TWO poses: Cur's reference pose and scale pose already exist in End (block indices posID1,
//     posID2 of the reference, Imp.cpp:7309-7314, 7383-7409).  Cur's copies are removed (m = m1 + m2 - 2), every index
//     of Cur is redirected -> the pose arrays are compacted;
//   * every U / W block that touches the reference pose is dropped (Imp.cpp:7482, 7531, 7619, 7678, 7774);
//   * Cur's (ScaP,ScaP) U block and Cur's W block (ScaP, f) of a shared feature are SUMMED into End's block when End has
//     one ("Fl/FlA", Imp.cpp:7484-7488, 7533-7547, 7621-7625, 7680-7700), otherwise appended;
//   * the scale pose's angles of both maps are unwrapped before they enter the right-hand side (Imp.cpp:7427-7465);
//   * the solve removes 7 scalars: the reference pose and the gauge-fixed translation of the scale pose
//     (lmj_solveLinearSFMMono, Imp.cpp:6981-7026), then sets stVal[Fix] = Sign.
#include <climits>
#include <cstdlib>

#include "lsfm_device.hpp"
#include "lsfm_internal.hpp"
#include "lsfm_join.hpp"
#include "lsfm_solve.hpp"

namespace lsfm {

struct MGroup {
	int pair;            // 1: two maps are joined, 0: carried map
	int ref, scap;       // pose ids shared by the two maps (= Cur's Ref / ScaP)
	int P1, P2, C1, C2;  // global (input) pose indices: End's reference / scale pose, Cur's copies of them
	int flU;             // input index of End's last (P2,P2) U block, -1 if none
	int fix, sign;       // End's Fix / Sign (gauge scalar and its value)
};

__global__ void k_mono_find(int M, const int* __restrict__ pose_id, const int* __restrict__ pose_map, MGroup* grp)
{
	int k = blockIdx.x * blockDim.x + threadIdx.x;
	if (k >= M) return;
	const int mp = pose_map[k];
	MGroup& g = grp[mp >> 1];
	const int id = pose_id[k];
	if (id == g.ref) { if (mp & 1) g.C1 = k; else g.P1 = k; }
	if (id == g.scap) { if (mp & 1) g.C2 = k; else g.P2 = k; }
}

// removed: Cur's copies of the shared poses (they become End's); dropped: the poses whose U / W blocks the join drops (the
// gauge-fixed reference pose on either side)
__global__ void k_mono_pose_flags(int M, const int* __restrict__ pose_map, const MGroup* __restrict__ grp, int* __restrict__ removed, unsigned char* __restrict__ dropped)
{
	int k = blockIdx.x * blockDim.x + threadIdx.x;
	if (k > M) return;
	if (k == M) { removed[M] = 0; return; }
	const MGroup& g = grp[pose_map[k] >> 1];
	removed[k] = (g.pair && (k == g.C1 || k == g.C2)) ? 1 : 0;
	dropped[k] = (g.pair && (k == g.P1 || k == g.C1)) ? 1 : 0;
}

// wrap-around of the scale pose's angles, in place on the copy of the prior poses (Imp.cpp:7427-7465)
__global__ void k_mono_wrap(int G, const MGroup* __restrict__ grp, double* __restrict__ prior)
{
	int g = blockIdx.x * blockDim.x + threadIdx.x;
	if (g >= G || !grp[g].pair) return;
	double* w1 = prior + (size_t)grp[g].P2 * 6 + 3;
	double* w2 = prior + (size_t)grp[g].C2 * 6 + 3;
	const double PI = LSFM_PI_REF;
	for (int i = 0; i < 3; i++)
	{
		int t;
		if (w1[i] > PI) { t = (int)(w1[i] / (2 * PI)); w1[i] -= (t + 1) * (2 * PI); }
		if (w1[i] < -PI) { t = (int)(w1[i] / (2 * PI)); w1[i] -= (t - 1) * (2 * PI); }
		if (w2[i] > PI) { t = (int)(w2[i] / (2 * PI)); w2[i] -= (t + 1) * (2 * PI); }
		if (w2[i] < -PI) { t = (int)(w2[i] / (2 * PI)); w2[i] -= (t - 1) * (2 * PI); }
		const double e = w2[i] - w1[i];
		if (e > PI) w2[i] -= 2 * PI; else if (e < -PI) w2[i] += 2 * PI;
	}
}

// new pose numbering + compacted pose arrays (values = priors; the solve overwrites them)
__global__ void k_mono_pose_remap(int M, const int* __restrict__ pose_map, const MGroup* __restrict__ grp, const int* __restrict__ R,
                                  const double* __restrict__ prior, const int* __restrict__ pose_id, const int* __restrict__ origin,
                                  int* __restrict__ pnew, double* __restrict__ pose_y, int* __restrict__ id_y, int* __restrict__ origin_y)
{
	int k = blockIdx.x * blockDim.x + threadIdx.x;
	if (k >= M) return;
	const MGroup& g = grp[pose_map[k] >> 1];
	if (g.pair && k == g.C1) { pnew[k] = g.P1 - R[g.P1]; return; }
	if (g.pair && k == g.C2) { pnew[k] = g.P2 - R[g.P2]; return; }
	const int n = k - R[k];
	pnew[k] = n;
	for (int i = 0; i < 6; i++) pose_y[(size_t)n * 6 + i] = prior[(size_t)k * 6 + i];
	id_y[n] = pose_id[k];
	origin_y[n] = origin[k];
}

// End's last (P2,P2) block (the reference keeps the LAST one it copied, Imp.cpp:7484-7488)
__global__ void k_mono_u_fl(int NU, const int* __restrict__ Ui, const int* __restrict__ Uj, const int* __restrict__ pose_map, MGroup* grp)
{
	int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= NU) return;
	const int mp = pose_map[Ui[i]];
	MGroup& g = grp[mp >> 1];
	if (g.pair && !(mp & 1) && Ui[i] == g.P2 && Uj[i] == g.P2) atomicMax(&g.flU, i);
}
// 1 = kept as its own block, 0 = dropped or summed into End's block
__global__ void k_mono_u_flags(int NU, const int* __restrict__ Ui, const int* __restrict__ Uj, const int* __restrict__ pose_map,
                               const MGroup* __restrict__ grp, int* __restrict__ keep)
{
	int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i > NU) return;
	if (i == NU) { keep[NU] = 0; return; }
	const int a = Ui[i], b = Uj[i], mp = pose_map[a];
	const MGroup& g = grp[mp >> 1];
	int k = 1;
	if (g.pair)
	{
		if (a == g.P1 || b == g.P1 || a == g.C1 || b == g.C1) k = 0;
		else if ((mp & 1) && a == g.C2 && b == g.C2 && g.flU >= 0) k = 0;
	}
	keep[i] = k;
}
// copies / sums the blocks and accumulates eP += U x, eP += U^T x (Imp.cpp:7480-7588) with the source maps' estimates
__global__ void k_mono_u_fill(int NU, const double* __restrict__ U, const int* __restrict__ Ui, const int* __restrict__ Uj,
                              const int* __restrict__ pose_map, const MGroup* __restrict__ grp, const int* __restrict__ keep,
                              const int* __restrict__ KU, const int* __restrict__ pnew, const double* __restrict__ prior,
                              double* __restrict__ Uy, int* __restrict__ Uiy, int* __restrict__ Ujy, double* __restrict__ eP, int add_rhs)
{
	int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= NU) return;
	const int a = Ui[i], b = Uj[i], mp = pose_map[a];
	const MGroup& g = grp[mp >> 1];
	const bool dropped = g.pair && (a == g.P1 || b == g.P1 || a == g.C1 || b == g.C1);
	if (dropped) return;
	double u[36];
	ld<36>(u, U + (size_t)i * 36);
	const int an = pnew[a], bn = pnew[b];
	if (keep[i])
	{
		const int pos = KU[i];
		// Uy is zero-initialised; the one block of a pair that also receives Cur's (ScaP,ScaP) block is accumulated
		if (g.pair && i == g.flU) { for (int q = 0; q < 36; q++) atomic_add_f64(Uy + (size_t)pos * 36 + q, u[q]); }
		else st<36>(Uy + (size_t)pos * 36, u);
		Uiy[pos] = an; Ujy[pos] = bn;
	}
	else
	{
		double* d = Uy + (size_t)KU[g.flU] * 36; // summed into End's (P2,P2) block
		for (int q = 0; q < 36; q++) atomic_add_f64(d + q, u[q]);
	}
	if (!add_rhs) return; // feature-sharded run: U is replicated, its part of the right-hand side is rank 0's
	const double* xb = prior + (size_t)b * 6;
	for (int r = 0; r < 6; r++)
	{
		double s = 0;
		for (int c = 0; c < 6; c++) s = fma(u[6 * r + c], xb[c], s);
		atomic_add_f64(eP + (size_t)an * 6 + r, s);
	}
	if (a != b)
	{
		const double* xa = prior + (size_t)a * 6;
		for (int c = 0; c < 6; c++)
		{
			double s = 0;
			for (int r = 0; r < 6; r++) s = fma(u[6 * r + c], xa[r], s);
			atomic_add_f64(eP + (size_t)bn * 6 + c, s);
		}
	}
}

// one lane per joint feature: how many W blocks survive from End's / Cur's run
__global__ void k_mono_w_count(int NFY, const int* __restrict__ srcE, const int* __restrict__ srcC, const int* __restrict__ fptr,
                               const int* __restrict__ photo, const int* __restrict__ feat_map_y, const MGroup* __restrict__ grp,
                               int* __restrict__ lens)
{
	int nf = blockIdx.x * blockDim.x + threadIdx.x;
	if (nf > NFY) return;
	if (nf == NFY) { lens[NFY] = 0; return; }
	const MGroup& g = grp[feat_map_y[nf]];
	int cnt = 0;
	bool hasP2 = false;
	const int fe = srcE[nf], fc = srcC[nf];
	if (fe >= 0)
		for (int j = fptr[fe]; j < fptr[fe + 1]; j++)
		{
			const int k = photo[j];
			if (g.pair && k == g.P1) continue;
			if (g.pair && k == g.P2) hasP2 = true;
			cnt++;
		}
	if (fc >= 0)
		for (int j = fptr[fc]; j < fptr[fc + 1]; j++)
		{
			const int k = photo[j];
			if (g.pair && k == g.C1) continue;
			if (g.pair && k == g.C2 && hasP2) continue; // summed into End's block
			cnt++;
		}
	lens[nf] = cnt;
}

// writes the joint run of every feature and the W part of the right-hand sides (Imp.cpp:7602-7821)
__global__ void k_mono_w_fill(int NFY, const int* __restrict__ srcE, const int* __restrict__ srcC, const int* __restrict__ fptr,
                              const int* __restrict__ photo, const double* __restrict__ W, const int* __restrict__ feat_map_y,
                              const MGroup* __restrict__ grp, const int* __restrict__ pnew, const double* __restrict__ prior,
                              const double* __restrict__ feat, const int* __restrict__ fptr_y, double* __restrict__ Wy, int* __restrict__ photo_y,
                              int* __restrict__ feature_y, double* __restrict__ eP, double* __restrict__ eF,
                              const int* __restrict__ feat_map_src, const double* __restrict__ W_alias, const int* __restrict__ alias)
{
	// eP: summed per work-group in an LDS table keyed by pose and flushed once (every feature adds to its hub poses:
	// global atomics on those few rows serialised -- 30 of the 55 ms of an RS468-like tree)
	constexpr int ECAP = 128;
	__shared__ int ekeys[ECAP];
	__shared__ double evals[ECAP * 6];
	for (int i = threadIdx.x; i < ECAP; i += blockDim.x) ekeys[i] = -1;
	for (int i = threadIdx.x; i < ECAP * 6; i += blockDim.x) evals[i] = 0.0;
	__syncthreads();
	const int nf = blockIdx.x * blockDim.x + threadIdx.x;
	const bool inb = nf < NFY;
	const MGroup& g = grp[feat_map_y[inb ? nf : 0]];
	int pos = inb ? fptr_y[nf] : 0, flpos = -1;
	double ef[3] = { 0, 0, 0 };
	for (int side = 0; side < 2 && inb; side++)
	{
		const int f = side ? srcC[nf] : srcE[nf];
		if (f < 0) continue;
		const double* xf = feat + (size_t)f * 3;
		// blocks of a map the transform passed through are still in the transform's input
		const int delta = alias ? alias[feat_map_src[f]] : INT_MIN;
		const double* Wsrc = delta != INT_MIN ? W_alias + (ptrdiff_t)delta * 18 : W;
		for (int j = fptr[f]; j < fptr[f + 1]; j++)
		{
			const int k = photo[j];
			if (g.pair && (k == g.P1 || k == g.C1)) continue;
			double w[18];
			ld<18>(w, Wsrc + (size_t)j * 18);
			const int kn = pnew[k];
			if (side == 1 && g.pair && k == g.C2 && flpos >= 0)
			{
				for (int q = 0; q < 18; q++) Wy[(size_t)flpos * 18 + q] += w[q]; // only this lane touches the feature's run
			}
			else
			{
				if (side == 0 && g.pair && k == g.P2) flpos = pos;
				st<18>(Wy + (size_t)pos * 18, w);
				photo_y[pos] = kn; feature_y[pos] = nf;
				pos++;
			}
			const double* xp = prior + (size_t)k * 6;
			const int es = lds_slot(ekeys, ECAP, kn);
			for (int r = 0; r < 6; r++)
			{
				const double y = w[3 * r] * xf[0] + w[3 * r + 1] * xf[1] + w[3 * r + 2] * xf[2];
				if (es >= 0) lds_add_f64(&evals[es * 6 + r], y); else atomic_add_f64(eP + (size_t)kn * 6 + r, y);
			}
			for (int c = 0; c < 3; c++)
				for (int r = 0; r < 6; r++) ef[c] = fma(w[3 * r + c], xp[r], ef[c]);
		}
	}
	if (inb) { eF[(size_t)nf * 3] += ef[0]; eF[(size_t)nf * 3 + 1] += ef[1]; eF[(size_t)nf * 3 + 2] += ef[2]; }
	__syncthreads();
	tile_flush<6>(ekeys, evals, ECAP, eP);
}

// ---- the same in two steps (the default): where every source block goes is index work, one lane per joint feature as above
// but ints only; the blocks themselves -- 144 bytes each -- are then moved one lane per SOURCE block, consecutive lanes
// on consecutive blocks, with the right-hand-side parts summed per source feature through LDS (tile_runs).  One lane per
// feature walking its 9-40 blocks, 144-byte loads a run length apart, was 7 % of an RS468-like tree. ----
#define MW_DROP (-1)            /* P1 / C1 block: gone, no contribution */
#define MW_SUMMED (-2)          /* Cur's block to C2: added into End's block to P2, contributes to the right-hand sides */
#define MW_TARGET (1 << 30)     /* End's last block to P2 of a feature whose Cur run holds C2 blocks: it adds them */
__global__ void k_mono_w_index(int NFY, const int* __restrict__ srcE, const int* __restrict__ srcC, const int* __restrict__ fptr,
                               const int* __restrict__ photo, const int* __restrict__ feat_map_y, const MGroup* __restrict__ grp,
                               const int* __restrict__ pnew, const int* __restrict__ fptr_y, int* __restrict__ photo_y,
                               int* __restrict__ feature_y, int* __restrict__ dst, int* __restrict__ jf)
{
	const int nf = blockIdx.x * blockDim.x + threadIdx.x;
	if (nf >= NFY) return;
	const MGroup& g = grp[feat_map_y[nf]];
	int pos = fptr_y[nf], flpos = -1, flj = -1;
	bool summed = false;
	for (int side = 0; side < 2; side++)
	{
		const int f = side ? srcC[nf] : srcE[nf];
		if (f < 0) continue;
		jf[f] = nf;
		for (int j = fptr[f]; j < fptr[f + 1]; j++)
		{
			const int k = photo[j];
			if (g.pair && (k == g.P1 || k == g.C1)) { dst[j] = MW_DROP; continue; }
			if (side == 1 && g.pair && k == g.C2 && flpos >= 0) { dst[j] = MW_SUMMED; summed = true; continue; }
			if (side == 0 && g.pair && k == g.P2) { flpos = pos; flj = j; }
			dst[j] = pos;
			photo_y[pos] = pnew[k]; feature_y[pos] = nf;
			pos++;
		}
	}
	if (summed) dst[flj] |= MW_TARGET;
}

#define MWC_TILE 128 /* source features per work-group */
__global__ void __launch_bounds__(256)
k_mono_w_copy(int NF, const int* __restrict__ fptr, const int* __restrict__ photo, const double* __restrict__ W, const int* __restrict__ feat_map_src,
              const double* __restrict__ W_alias, const int* __restrict__ alias, const int* __restrict__ dst, const int* __restrict__ jf,
              const int* __restrict__ srcC, const int* __restrict__ feat_map_y, const MGroup* __restrict__ grp, const int* __restrict__ pnew,
              const double* __restrict__ prior, const double* __restrict__ feat, double* __restrict__ Wy, double* __restrict__ eP,
              double* __restrict__ eF)
{
	constexpr int ECAP = 128;
	__shared__ int ekeys[ECAP];
	__shared__ double evals[ECAP * 6];
	__shared__ int sFp[MWC_TILE + 1];
	__shared__ double sT[256 * 3];
	const int f0 = blockIdx.x * MWC_TILE, nft = min(MWC_TILE, NF - f0);
	for (int i = threadIdx.x; i < ECAP; i += blockDim.x) ekeys[i] = -1;
	for (int i = threadIdx.x; i < ECAP * 6; i += blockDim.x) evals[i] = 0.0;
	for (int i = threadIdx.x; i <= nft; i += blockDim.x) sFp[i] = fptr[f0 + i];
	__syncthreads();
	// blocks of a map the transform passed through are still in the transform's input
	auto block_of = [&](int f, int j) -> const double* {
		const int delta = alias ? alias[feat_map_src[f]] : INT_MIN;
		return (delta != INT_MIN ? W_alias + (ptrdiff_t)delta * 18 : W) + (size_t)j * 18;
	};
	tile_runs<3>(nft, sFp, sT,
		[&](int j, int fl, double* out) {
			out[0] = 0.0; out[1] = 0.0; out[2] = 0.0;
			const int d = dst[j];
			if (d == MW_DROP) return;
			const int f = f0 + fl, k = photo[j], kn = pnew[k];
			double w[18];
			ld<18>(w, block_of(f, j));
			if (d >= 0)
			{
				const int pos = d & ~MW_TARGET;
				if (d & MW_TARGET)
				{
					// End's block to P2: Cur's block(s) to C2 of the same joint feature are summed into it (Imp.cpp:7619-7700)
					const int nf = jf[f], fc = srcC[nf], C2 = grp[feat_map_y[nf]].C2;
					double sum[18];
					ld<18>(sum, w);
					for (int j2 = fptr[fc]; j2 < fptr[fc + 1]; j2++)
						if (photo[j2] == C2)
						{
							double w2[18];
							ld<18>(w2, block_of(fc, j2));
							for (int q = 0; q < 18; q++) sum[q] += w2[q];
						}
					st<18>(Wy + (size_t)pos * 18, sum);
				}
				else st<18>(Wy + (size_t)pos * 18, w);
			}
			// right-hand sides: the block times ITS map's estimates (eP += W x_f, eF += W^T x_p)
			const double* xf = feat + (size_t)f * 3;
			const double* xp = prior + (size_t)k * 6;
			const int es = lds_slot(ekeys, ECAP, kn);
#pragma unroll
			for (int r = 0; r < 6; r++)
			{
				const double y = w[3 * r] * xf[0] + w[3 * r + 1] * xf[1] + w[3 * r + 2] * xf[2];
				if (es >= 0) lds_add_f64(&evals[es * 6 + r], y); else atomic_add_f64(eP + (size_t)kn * 6 + r, y);
			}
#pragma unroll
			for (int c = 0; c < 3; c++)
			{
				double sacc = 0.0;
#pragma unroll
				for (int r = 0; r < 6; r++) sacc = fma(w[3 * r + c], xp[r], sacc);
				out[c] = sacc;
			}
		},
		[&](int fl, int q, double sum, bool) {
			const int nf = jf[f0 + fl];
			if (nf >= 0 && sum != 0.0) atomic_add_f64(eF + (size_t)nf * 3 + q, sum); // End's and Cur's run of a joint feature both land here
		});
	tile_flush<6>(ekeys, evals, ECAP, eP); // tile_runs ends with a barrier
}

__global__ void k_mono_fixed(int G, const MGroup* __restrict__ grp, const int* __restrict__ pnew, unsigned char* __restrict__ fixed)
{
	int g = blockIdx.x * blockDim.x + threadIdx.x;
	if (g >= G || grp[g].P1 < 0 || grp[g].P2 < 0) return; // carried maps keep their own gauge rows out of the system too
	const int p1 = pnew[grp[g].P1], p2 = pnew[grp[g].P2];
	for (int i = 0; i < 6; i++) fixed[(size_t)p1 * 6 + i] = 1;
	fixed[(size_t)p2 * 6 + grp[g].fix] = 1;
}
// stVal[Fix] = Sign (Imp.cpp:7026); the reference pose stays at zero (Imp.cpp:7010-7021)
__global__ void k_mono_finish(int G, const MGroup* __restrict__ grp, const int* __restrict__ pnew, double* __restrict__ pose_y)
{
	int g = blockIdx.x * blockDim.x + threadIdx.x;
	if (g >= G || grp[g].P2 < 0) return;
	pose_y[(size_t)pnew[grp[g].P2] * 6 + grp[g].fix] = (double)grp[g].sign;
}
__global__ void k_fill_int(int n, int* p, int v)
{
	int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) p[i] = v;
}

void join_batch_mono(lsfm_context* ctx, Arena& ar, const DevBatch& in, DevBatch& out, double* eP_out, double* eF_out)
{
	hipStream_t s = ctx->stream;
	const int B = in.B, G = (B + 1) / 2, M = in.M;
	size_t smark = ctx->scratch.mark();
	Arena& sc = ctx->scratch;

	// ---- shared poses, pose compaction ----
	std::vector<MGroup> mg(G);
	int npair = 0;
	for (int g = 0; g < G; g++)
	{
		const int a = 2 * g, b = 2 * g + 1;
		MGroup& x = mg[g];
		x.pair = b < B; x.P1 = x.P2 = x.C1 = x.C2 = -1; x.flU = -1;
		x.ref = x.pair ? in.Ref[b] : in.Ref[a]; x.scap = x.pair ? in.ScaP[b] : in.ScaP[a];
		x.fix = in.Fix[a]; x.sign = in.Sign[a];
		npair += x.pair;
		if (x.pair && (in.Ref[a] != in.Ref[b] || in.ScaP[a] != in.ScaP[b]))
			LSFM_FAIL(LSFM_ERR_ARG, "Mono join: End must already be expressed in Cur's frame (Ref/ScaP differ)");
	}
	MGroup* d_mg = sc.alloc<MGroup>(G);
	h2d(ctx, d_mg, mg.data(), sizeof(MGroup) * G);
	int* removed = sc.alloc<int>(M + 1);
	unsigned char* dropped = sc.alloc<unsigned char>(M + 1);
	int* R = sc.alloc<int>(M + 2);
	int* pnew = sc.alloc<int>(M + 1);
	double* prior = sc.alloc<double>((size_t)M * 6);
	const int nbp = (M + 255) / 256;
	hipLaunchKernelGGL(k_mono_find, dim3(nbp), dim3(256), 0, s, M, in.pose_id, in.pose_map, d_mg);
	hipLaunchKernelGGL(k_mono_pose_flags, dim3((M + 256) / 256), dim3(256), 0, s, M, in.pose_map, d_mg, removed, dropped);
	dev_exclusive_scan(ctx, removed, R, M);
	LSFM_CHECK_HIP(hipMemcpyAsync(prior, in.pose, (size_t)M * 6 * sizeof(double), hipMemcpyDeviceToDevice, s));
	hipLaunchKernelGGL(k_mono_wrap, dim3((G + 127) / 128), dim3(128), 0, s, G, d_mg, prior);
	LevelPlan* plan = ctx->plan;
	const bool warm = ctx->warm(); // the structure of this level is known from an earlier run of the same tree: no round trips
	const int MY = M - 2 * npair;

	// ---- common features (K5), same as Stereo ----
	int* match = sc.alloc<int>(in.NF + 1);
	int* unm = sc.alloc<int>(in.NF + 2);
	int* RF = sc.alloc<int>(in.NF + 2);
	const int nb = (in.NF + 255) / 256;
	if (in.NF) join_match_features(ctx, in, match, unm);
	else dev_zero(ctx, unm, 2 * sizeof(int));
	dev_exclusive_scan(ctx, unm, RF, in.NF);
	int* d_rb = sc.alloc<int>(B + 1);
	hipLaunchKernelGGL(k_gather_at, dim3((B + 1 + 127) / 128), dim3(128), 0, s, RF, in.d_feat_off, B + 1, d_rb);
	// ---- U: which blocks survive (index work, needed on the host for the container sizes like the unmatched ranks) ----
	int* keepU = sc.alloc<int>(in.NU + 1);
	int* KU = sc.alloc<int>(in.NU + 2);
	if (in.NU) hipLaunchKernelGGL(k_mono_u_fl, dim3((in.NU + 255) / 256), dim3(256), 0, s, in.NU, in.Ui, in.Uj, in.pose_map, d_mg);
	hipLaunchKernelGGL(k_mono_u_flags, dim3((in.NU + 256) / 256), dim3(256), 0, s, in.NU, in.Ui, in.Uj, in.pose_map, d_mg, keepU);
	dev_exclusive_scan(ctx, keepU, KU, in.NU);
	std::vector<int> rb(B + 1), uo(B + 1);
	if (warm) { rb = plan->join_rb; uo = plan->join_uo; }
	else
	{
		// ONE read-back for everything the host needs of this level's structure: the pair records (are the shared poses there?), the
		// ranks of the unmatched features and the kept-U prefix at the map boundaries (three round trips before)
		int* d_uo = sc.alloc<int>(B + 1);
		int* d_ui = sc.alloc<int>(B + 1);
		h2d(ctx, d_ui, in.u_off.data(), (B + 1) * sizeof(int));
		hipLaunchKernelGGL(k_gather_at, dim3((B + 1 + 127) / 128), dim3(128), 0, s, KU, d_ui, B + 1, d_uo);
		const size_t nmg = sizeof(MGroup) * (size_t)G / sizeof(int);
		int* d_pack = sc.alloc<int>(nmg + 2 * (size_t)(B + 1));
		CopyBatch cp(ctx);
		cp.d2d(d_pack, d_mg, sizeof(MGroup) * (size_t)G);
		cp.d2d(d_pack + nmg, d_rb, sizeof(int) * (size_t)(B + 1));
		cp.d2d(d_pack + nmg + (B + 1), d_uo, sizeof(int) * (size_t)(B + 1));
		cp.flush();
		std::vector<int> hp(nmg + 2 * (size_t)(B + 1));
		d2h_ints(ctx, d_pack, hp.data(), hp.size());
		memcpy(mg.data(), hp.data(), sizeof(MGroup) * (size_t)G);
		std::copy(hp.begin() + nmg, hp.begin() + nmg + (B + 1), rb.begin());
		std::copy(hp.begin() + nmg + (B + 1), hp.end(), uo.begin());
		for (int g = 0; g < G; g++)
			if (mg[g].pair && (mg[g].P1 < 0 || mg[g].P2 < 0 || mg[g].C1 < 0 || mg[g].C2 < 0))
				LSFM_FAIL(LSFM_ERR_ARG, "Mono join: shared reference / scale pose missing in pair " + std::to_string(g));
		if (plan) { plan->join_rb = rb; plan->join_uo = uo; }
	}

	out = DevBatch();
	out.B = G; out.M = MY;
	out.pose_off.assign(G + 1, 0); out.feat_off.assign(G + 1, 0); out.u_off.assign(G + 1, 0); out.w_off.assign(G + 1, 0);
	out.Ref.resize(G); out.FRef.resize(G); out.ScaP.resize(G); out.Fix.resize(G); out.Sign.resize(G); out.FScaP.resize(G); out.FFix.resize(G);
	std::vector<JGroup> grp(G);
	std::vector<unsigned char> seg_active(G);
	std::vector<int> seg_rows(G);
	for (int g = 0; g < G; g++)
	{
		const int a = 2 * g, b = 2 * g + 1;
		const bool pair = b < B;
		JGroup& jg = grp[g];
		jg.F0E = in.feat_off[a]; jg.nE = in.feat_off[a + 1] - jg.F0E;
		jg.F0C = pair ? in.feat_off[b] : in.feat_off[a + 1]; jg.nC = pair ? in.feat_off[b + 1] - jg.F0C : 0;
		jg.FY0 = out.feat_off[g];
		jg.rC0 = pair ? rb[b] : 0;
		out.feat_off[g + 1] = jg.FY0 + jg.nE + (pair ? rb[b + 1] - rb[b] : 0);
		const int rows = (pair ? in.pose_off[b + 1] : in.pose_off[a + 1]) - in.pose_off[a] - (pair ? 2 : 0);
		out.pose_off[g + 1] = out.pose_off[g] + rows;
		seg_rows[g] = rows; seg_active[g] = pair ? 1 : 0;
		const int c = pair ? b : a; // Imp.cpp:7365-7373
		out.Ref[g] = in.Ref[c]; out.ScaP[g] = in.ScaP[c]; out.Fix[g] = in.Fix[c]; out.Sign[g] = in.Sign[c];
		out.FRef[g] = in.FRef[a]; out.FScaP[g] = in.FScaP[a]; out.FFix[g] = in.FFix[a];
	}
	out.NF = out.feat_off[G];
	const int NFY = out.NF;
	JGroup* d_grp = sc.alloc<JGroup>(G);
	h2d(ctx, d_grp, grp.data(), sizeof(JGroup) * G);

	out.pose = ar.alloc<double>((size_t)MY * 6); out.pose_id = ar.alloc<int>(MY); out.pose_origin = ar.alloc<int>(MY);
	out.feat = ar.alloc<double>((size_t)NFY * 3); out.feat_id = ar.alloc<int>(NFY);
	out.fptr = ar.alloc<int>(NFY + 1); out.V = ar.alloc<double>((size_t)NFY * 9);
	batch_set_offsets(ctx, ar, out);
	hipLaunchKernelGGL(k_mono_pose_remap, dim3(nbp), dim3(256), 0, s, M, in.pose_map, d_mg, R, prior, in.pose_id, in.pose_origin, pnew, out.pose,
	                   out.pose_id, out.pose_origin);

	double* eP = sc.alloc<double>((size_t)MY * 6);
	double* eF = sc.alloc<double>((size_t)NFY * 3);
	dev_zero(ctx, eP, (size_t)MY * 6 * sizeof(double)); // (eF, out.V and the source lists get their first values from k_join_features)

	// ---- U ----
	out.NU = uo[B];
	for (int g = 0; g <= G; g++) out.u_off[g] = uo[std::min(2 * g, B)];
	out.U = ar.alloc<double>((size_t)out.NU * 36); out.Ui = ar.alloc<int>(out.NU); out.Uj = ar.alloc<int>(out.NU);
	dev_zero(ctx, out.U, (size_t)out.NU * 36 * sizeof(double));
	if (in.NU)
		hipLaunchKernelGGL(k_mono_u_fill, dim3((in.NU + 127) / 128), dim3(128), 0, s, in.NU, in.U, in.Ui, in.Uj, in.pose_map, d_mg, keepU, KU, pnew,
		                   prior, out.U, out.Ui, out.Uj, eP, (!ctx->comm || ctx->comm->rank == 0) ? 1 : 0);

	// ---- features: V, run lengths, W ----
	int* newf = sc.alloc<int>(in.NF + 1);
	int* lenE = sc.alloc<int>(NFY + 1);
	int* lenC = sc.alloc<int>(NFY + 1);
	int* lens = sc.alloc<int>(NFY + 2);
	int* srcE = sc.alloc<int>(NFY + 1);
	int* srcC = sc.alloc<int>(NFY + 1);
	if (in.NF)
		for (int side = 0; side < 2; side++)
			hipLaunchKernelGGL(k_join_features, dim3(nb), dim3(256), 0, s, in.NF, in.feat_map, in.feat_id, in.feat, in.V, in.fptr, match, RF, d_grp,
			                   newf, lenE, lenC, out.V, eF, out.feat_id, out.feat, srcE, srcC, side);
	hipLaunchKernelGGL(k_mono_w_count, dim3((NFY + 256) / 256), dim3(256), 0, s, NFY, srcE, srcC, in.fptr, in.photo, out.feat_map, d_mg, lens);
	dev_exclusive_scan(ctx, lens, out.fptr, NFY);
	{
		// W offsets of the joint maps
		int* d_fo = sc.alloc<int>(G + 1);
		int* d_wo = sc.alloc<int>(G + 1);
		h2d(ctx, d_fo, out.feat_off.data(), (G + 1) * sizeof(int));
		if (warm) out.w_off = plan->join_wo;
		else
		{
			hipLaunchKernelGGL(k_gather_at, dim3((G + 1 + 127) / 128), dim3(128), 0, s, out.fptr, d_fo, G + 1, d_wo);
			d2h_ints(ctx, d_wo, out.w_off.data(), G + 1);
			if (plan) plan->join_wo = out.w_off;
		}
	}
	out.NW = out.w_off[G];
	out.W = ar.alloc<double>((size_t)out.NW * 18); out.photo = ar.alloc<int>(out.NW); out.feature = ar.alloc<int>(out.NW);
	static const bool one_lane_per_feature = getenv("LSFM_MONO_FILL_BY_FEATURE") != nullptr; // the round-1 kernel, kept for comparison
	if (NFY && one_lane_per_feature)
		hipLaunchKernelGGL(k_mono_w_fill, dim3((NFY + 127) / 128), dim3(128), 0, s, NFY, srcE, srcC, in.fptr, in.photo, in.W, out.feat_map, d_mg,
		                   pnew, prior, in.feat, out.fptr, out.W, out.photo, out.feature, eP, eF, in.feat_map, in.W_alias, in.d_alias);
	else if (NFY)
	{
		int* dst = sc.alloc<int>((size_t)in.NW + 1);
		int* jf = sc.alloc<int>((size_t)in.NF + 1);
		fill_async(s, dst, 0xff, sizeof(int) * (size_t)in.NW); // MW_DROP for blocks no joint feature claims
		fill_async(s, jf, 0xff, sizeof(int) * (size_t)in.NF);
		hipLaunchKernelGGL(k_mono_w_index, dim3((NFY + 255) / 256), dim3(256), 0, s, NFY, srcE, srcC, in.fptr, in.photo, out.feat_map, d_mg, pnew, out.fptr,
		                   out.photo, out.feature, dst, jf);
		// the joint maps' index arrays are final here (U's were written by k_mono_u_fill): a level that analyses builds the pattern of S
		// on the side stream from this point, beside the kernel that moves the W blocks and the right-hand sides (solve_batch)
		if (!warm && !eP_out && !eF_out && !ctx->comm)
		{
			LSFM_CHECK_HIP(hipEventRecord(ctx->evA, s));
			ctx->pattern_dep = true;
		}
		if (in.NF)
			hipLaunchKernelGGL(k_mono_w_copy, dim3((in.NF + MWC_TILE - 1) / MWC_TILE), dim3(256), 0, s, in.NF, in.fptr, in.photo, in.W, in.feat_map,
			                   in.W_alias, in.d_alias, dst, jf, srcC, out.feat_map, d_mg, pnew, prior, in.feat, out.W, eP, eF);
	}
	LSFM_CHECK_HIP(hipGetLastError());
	if (eP_out) d2h(ctx, eP_out, eP, (size_t)MY * 6 * sizeof(double));
	if (eF_out) d2h(ctx, eF_out, eF, (size_t)NFY * 3 * sizeof(double));

	// ---- solve with the 7 gauge scalars of every pair removed ----
	unsigned char* fixed = sc.alloc<unsigned char>((size_t)MY * 6 + 8);
	dev_zero(ctx, fixed, (size_t)MY * 6 + 8);
	hipLaunchKernelGGL(k_mono_fixed, dim3((G + 127) / 128), dim3(128), 0, s, G, d_mg, pnew, fixed);
	unsigned char* d_act = sc.alloc<unsigned char>(G);
	h2d(ctx, d_act, seg_active.data(), G);
	double* x0 = sc.alloc<double>((size_t)MY * 6);
	LSFM_CHECK_HIP(hipMemcpyAsync(x0, out.pose, (size_t)MY * 6 * sizeof(double), hipMemcpyDeviceToDevice, s));
	SolveIO io;
	io.M = MY; io.NF = NFY; io.NU = out.NU; io.NW = out.NW; io.nseg = G;
	io.d_pose_seg = out.pose_map; io.d_feat_seg = out.feat_map; io.d_seg_active = d_act;
	io.U = out.U; io.Ui = out.Ui; io.Uj = out.Uj; io.W = out.W; io.photo = out.photo; io.fptr = out.fptr; io.V = out.V;
	io.ea = eP; io.eb = eF; io.x0 = x0; io.d_fixed = fixed; io.d_pose_origin = out.pose_origin;
	io.x_pose = out.pose; io.x_feat = out.feat;
	io.seg_rows = seg_rows;
	{
		int most = 0;
		for (int r : seg_rows) most = std::max(most, r);
		if (ctx->small_max > 0 && small_solve_strips(most, ctx->small_max))
		{
			int* d_uo = sc.alloc<int>(G + 1);
			h2d(ctx, d_uo, out.u_off.data(), sizeof(int) * (size_t)(G + 1));
			io.d_pose_off = out.d_pose_off; io.d_feat_off = out.d_feat_off; io.d_u_off = d_uo;
		}
	}
	// the pattern of this level's system from the one below (a level that analyses; the level below left its pattern with its maps)
	PatternSeed seed;
	static const bool seed_on = !getenv("LSFM_NO_MONO_SEED");
	if (seed_on && !warm && in.s_keys && in.s_nnzb > 0 && !ctx->comm)
	{
		seed.prev_keys = in.s_keys; seed.prev_nnzb = in.s_nnzb; seed.pnew = pnew; seed.dropped = dropped;
		seed.NFY = NFY; seed.srcE = srcE; seed.srcC = srcC; seed.fptr_in = in.fptr; seed.photo_in = in.photo;
		io.seed = &seed;
	}
	ctx->solved_keys = nullptr; ctx->solved_nnzb = 0;
	ctx->level_syncs = !warm; // (this level waits for the device below: its refinement may ask once instead of guessing a margin)
	int rc;
	try { rc = solve_batch(ctx, io); }
	catch (...) { ctx->level_syncs = false; throw; }
	ctx->level_syncs = false;
	if (!warm && ctx->in_tree_run && ctx->solved_keys && ctx->solved_nnzb > 0 && !ctx->comm)
	{
		// ... and this level's pattern stays with its output for the level above
		unsigned long long* k = ar.alloc<unsigned long long>((size_t)ctx->solved_nnzb + 1);
		LSFM_CHECK_HIP(hipMemcpyAsync(k, ctx->solved_keys, (size_t)ctx->solved_nnzb * sizeof(unsigned long long), hipMemcpyDeviceToDevice, s));
		out.s_keys = k; out.s_nnzb = ctx->solved_nnzb;
	}
	hipLaunchKernelGGL(k_mono_finish, dim3((G + 127) / 128), dim3(128), 0, s, G, d_mg, pnew, out.pose);
	if (!warm) LSFM_CHECK_HIP(hipStreamSynchronize(s)); // a warm level is only enqueued: its scratch is reused in stream order
	sc.release(smark);
	if (rc > 0 && ctx->stats) ctx->stats->not_converged += rc;
	if (plan && !eP_out && !eF_out) plan->valid = true; // every stage of the level has left its structure behind
}

} // namespace lsfm
