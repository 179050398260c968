// Batched pairwise join of the maps of one tree level (Stereo): feature matching, merge of the two information
// matrices, right-hand side, then the batched solve.  Replaces lmj_LinearLS_PF3DStereo (Imp.cpp:2551-2978) for all
// pairs (2g, 2g+1) of the level at once.
//
// Because U/W carry GLOBAL pose indices, the joint pose set of a pair is simply the two maps' poses back to back
// (Imp.cpp:2626-2627 does the same copy), U is reused unchanged (Imp.cpp:2658-2735 only shifts Cur's indices by m1)
// and the join reduces to renumbering features:  End's features keep their order, Cur's unmatched features are
// appended (Imp.cpp:2630-2643), W runs of a shared feature are End's then Cur's (Imp.cpp:2761-2847), V is summed
// (Imp.cpp:2796-2800).  The reference finds common features with std::find over all of Cur's labels per End
// feature (O(n1*n2), Imp.cpp:2581-2599); here one hash join per level.
#include "lsfm_device.hpp"
#include "lsfm_internal.hpp"
#include <climits>
#include "lsfm_join.hpp"
#include "lsfm_solve.hpp"

namespace lsfm {

// Common features of a pair (K5): a hash join.  The features of the first map of every pair go into an open-addressing
// table keyed by (pair, label); the features of the second map probe it.  (The reference looks every End feature up in
// Cur with std::find, O(n1 n2), Imp.cpp:2581-2599; the first version here sorted all (pair, label, side) keys of the
// level, ~10 merge passes over up to 2.3 M keys.)  Labels are unique inside a map.
__device__ __forceinline__ unsigned long long jh_mix(unsigned long long x)
{
	x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33;
	return x;
}
__global__ void k_join_hash_insert(int NF, const int* __restrict__ feat_id, const int* __restrict__ feat_map, unsigned long long* tab,
                                   int* __restrict__ tval, unsigned long long mask)
{
	int f = blockIdx.x * blockDim.x + threadIdx.x;
	if (f >= NF) return;
	const int mp = feat_map[f];
	if (mp & 1) return;
	const unsigned long long key = ((unsigned long long)(mp >> 1) << 32) | (unsigned)feat_id[f];
	unsigned long long h = jh_mix(key) & mask;
	for (;;)
	{
		const unsigned long long old = atomicCAS(&tab[h], ~0ull, key);
		if (old == ~0ull || old == key) { tval[h] = f; return; }
		h = (h + 1) & mask;
	}
}
__global__ void k_join_hash_probe(int NF, const int* __restrict__ feat_id, const int* __restrict__ feat_map, const unsigned long long* __restrict__ tab,
                                  const int* __restrict__ tval, unsigned long long mask, int* __restrict__ match, int* __restrict__ unmatched)
{
	int f = blockIdx.x * blockDim.x + threadIdx.x;
	if (f >= NF) return;
	const int mp = feat_map[f];
	int mt = -1;
	if (mp & 1)
	{
		const unsigned long long key = ((unsigned long long)(mp >> 1) << 32) | (unsigned)feat_id[f];
		unsigned long long h = jh_mix(key) & mask;
		for (;;)
		{
			const unsigned long long cur = tab[h];
			if (cur == key) { mt = tval[h]; break; }
			if (cur == ~0ull) break;
			h = (h + 1) & mask;
		}
	}
	match[f] = mt;
	unmatched[f] = ((mp & 1) && mt < 0) ? 1 : 0;
	if (f == 0) unmatched[NF] = 0;
}
void join_match_features(lsfm_context* ctx, const DevBatch& in, int* match, int* unm)
{
	hipStream_t s = ctx->stream;
	size_t cap = 1024;
	while (cap < 2 * (size_t)in.NF) cap <<= 1;
	unsigned long long* tab = ctx->scratch.alloc<unsigned long long>(cap);
	int* tval = ctx->scratch.alloc<int>(cap);
	fill_async(s, tab, 0xff, cap * sizeof(unsigned long long));
	const int nb = (in.NF + 255) / 256;
	hipLaunchKernelGGL(k_join_hash_insert, dim3(nb), dim3(256), 0, s, in.NF, in.feat_id, in.feat_map, tab, tval, (unsigned long long)(cap - 1));
	hipLaunchKernelGGL(k_join_hash_probe, dim3(nb), dim3(256), 0, s, in.NF, in.feat_id, in.feat_map, tab, tval, (unsigned long long)(cap - 1), match, unm);
}

__global__ void k_gather_at(const int* __restrict__ src, const int* __restrict__ idx, int n, int* __restrict__ out)
{
	int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) out[i] = src[idx[i]];
}

// new feature numbers, merged V, V-part of eF, run lengths
__global__ void k_join_features(int NF, const int* __restrict__ feat_map, const int* __restrict__ feat_id, const double* __restrict__ feat,
                                const double* __restrict__ V, const int* __restrict__ fptr, const int* __restrict__ match,
                                const int* __restrict__ R, const JGroup* __restrict__ grp, int* __restrict__ newf, int* __restrict__ lenE,
                                int* __restrict__ lenC, double* __restrict__ Vy, double* __restrict__ eF, int* __restrict__ fid_y,
                                double* __restrict__ feat_y, int* __restrict__ srcE, int* __restrict__ srcC, int side)
{
	// two launches: side 0 = features of the first map of every pair, then side 1 = features of the second map, which add to
	// their match (each joint feature has one writer per launch: no atomics -- 12 scattered 8-byte atomics per feature cost 0.3 ms
	// per level).  The FIRST writer of a joint feature -- its End source, or its Cur source when it has no match -- stores, and
	// stores the other side's run length / source as absent: the joint arrays need no fill (until round 5: ~100 bytes per joint
	// feature of memset and four fill launches per level)
	int f = blockIdx.x * blockDim.x + threadIdx.x;
	if (f >= NF) return;
	int mp = feat_map[f];
	const JGroup& g = grp[mp >> 1];
	int nf;
	const bool cur = mp & 1;
	if ((int)cur != side) return;
	if (!cur) nf = g.FY0 + (f - g.F0E);
	else if (match[f] >= 0) nf = g.FY0 + (match[f] - g.F0E);
	else nf = g.FY0 + g.nE + (R[f] - g.rC0);
	newf[f] = nf;
	const bool first = !cur || match[f] < 0;
	int len = fptr[f + 1] - fptr[f];
	if (!cur) { lenE[nf] = len; lenC[nf] = 0; }
	else { lenC[nf] = len; if (first) lenE[nf] = 0; }
	if (srcE)
	{
		if (!cur) { srcE[nf] = f; srcC[nf] = -1; }
		else { srcC[nf] = f; if (first) srcE[nf] = -1; }
	}
	const double* v = V + (size_t)f * 9;
	const double* x = feat + (size_t)f * 3;
	// eF += V x  with each map's own estimate (Imp.cpp:2752-2757, 2802-2807, 2874-2879)
	if (first)
	{
		for (int i = 0; i < 9; i++) Vy[(size_t)nf * 9 + i] = v[i];
		for (int r = 0; r < 3; r++) eF[(size_t)nf * 3 + r] = v[3 * r] * x[0] + v[3 * r + 1] * x[1] + v[3 * r + 2] * x[2];
	}
	else
	{
		for (int i = 0; i < 9; i++) Vy[(size_t)nf * 9 + i] += v[i];
		for (int r = 0; r < 3; r++) eF[(size_t)nf * 3 + r] += v[3 * r] * x[0] + v[3 * r + 1] * x[1] + v[3 * r + 2] * x[2];
	}
	if (first)
	{
		fid_y[nf] = feat_id[f];
		feat_y[(size_t)nf * 3] = x[0]; feat_y[(size_t)nf * 3 + 1] = x[1]; feat_y[(size_t)nf * 3 + 2] = x[2];
	}
}

__global__ void k_add_lens(int n, const int* __restrict__ a, const int* __restrict__ b, int* __restrict__ out)
{
	int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) out[i] = a[i] + b[i];
	if (i == 0) out[n] = 0;
}

__global__ void k_join_wbase(int NF, const int* __restrict__ feat_map, const int* __restrict__ newf, const int* __restrict__ lenE,
                             const int* __restrict__ fptr_y, int* __restrict__ wbase)
{
	int f = blockIdx.x * blockDim.x + threadIdx.x;
	if (f >= NF) return;
	const int nf = newf[f];
	wbase[f] = fptr_y[nf] + ((feat_map[f] & 1) ? lenE[nf] : 0); // the second map's blocks follow the first map's
}
__global__ void k_join_wcopy(int NW, const double* __restrict__ W, const int* __restrict__ photo, const int* __restrict__ feature,
                             const int* __restrict__ fptr, const int* __restrict__ wbase, const int* __restrict__ newf,
                             double* __restrict__ Wy, int* __restrict__ photo_y, int* __restrict__ feature_y, int* __restrict__ srcf,
                             const double* __restrict__ W_alias, const int* __restrict__ alias, const int* __restrict__ feat_map)
{
	int j = blockIdx.x * blockDim.x + threadIdx.x;
	if (j >= NW) return;
	const int f = feature[j];
	const int dest = wbase[f] + (j - fptr[f]);
	// blocks of a map the transform passed through are still in the transform's input
	const int delta = alias ? alias[feat_map[f]] : INT_MIN;
	const double* w = delta != INT_MIN ? W_alias + (size_t)(j + delta) * 18 : W + (size_t)j * 18;
	double* o = Wy + (size_t)dest * 18;
	for (int i = 0; i < 18; i++) o[i] = w[i];
	photo_y[dest] = photo[j];
	feature_y[dest] = newf[f];
	srcf[dest] = f;
}

// right-hand side: eF += W^T x_pose, eP += W x_feat (each block with the estimates of the map it came from),
// Imp.cpp:2770-2786, 2822-2838, 2891-2906.  One lane per W block of the joint map (coalesced); the feature sums go
// through LDS per run, the pose sums through an LDS table flushed once per work-group.
#define RHS_TILE 128 /* joint features per work-group */
__global__ void __launch_bounds__(256)
k_join_rhs_w(int NFY, const int* __restrict__ fptr_y, const double* __restrict__ Wy, const int* __restrict__ photo_y,
             const int* __restrict__ srcf, const double* __restrict__ pose, const double* __restrict__ feat, double* __restrict__ eP,
             double* __restrict__ eF)
{
	constexpr int ECAP = 128;
	__shared__ int ekeys[ECAP];
	__shared__ double evals[ECAP * 6];
	__shared__ int sFp[RHS_TILE + 1];
	__shared__ double sT[256 * 3];
	const int f0 = blockIdx.x * RHS_TILE, nft = min(RHS_TILE, NFY - f0);
	for (int i = threadIdx.x; i < ECAP; i += blockDim.x) ekeys[i] = -1;
	for (int i = threadIdx.x; i < ECAP * 6; i += blockDim.x) evals[i] = 0.0;
	for (int i = threadIdx.x; i <= nft; i += blockDim.x) sFp[i] = fptr_y[f0 + i];
	__syncthreads();
	tile_runs<3>(nft, sFp, sT,
		[&](int j, int, double* out) {
			const int k = photo_y[j];
			double W[18], xp[6], xf[3];
			ld<18>(W, Wy + (size_t)j * 18);
			ld<6>(xp, pose + (size_t)k * 6);
			ld<3>(xf, feat + (size_t)srcf[j] * 3);
			const int sl = lds_slot(ekeys, ECAP, k);
#pragma unroll
			for (int r = 0; r < 6; r++)
			{
				const double y = W[3 * r] * xf[0] + W[3 * r + 1] * xf[1] + W[3 * r + 2] * xf[2];
				if (sl >= 0) lds_add_f64(&evals[sl * 6 + r], y); else atomic_add_f64(eP + (size_t)k * 6 + r, y);
			}
#pragma unroll
			for (int c = 0; c < 3; c++)
			{
				double sacc = 0.0;
#pragma unroll
				for (int r = 0; r < 6; r++) sacc = fma(W[3 * r + c], xp[r], sacc);
				out[c] = sacc;
			}
		},
		[&](int fl, int q, double sum, bool) { eF[(size_t)(f0 + fl) * 3 + q] += sum; });
	tile_flush<6>(ekeys, evals, ECAP, eP); // tile_runs ends with a barrier
}

// eP += U x, eP += U^T x for off-diagonal blocks, Imp.cpp:2666-2688
// (the blocks of a map to its hub pose follow each other and all add to the hub's six numbers: straight to memory that was thousands of
// atomics on one address -- 156 us on the largest level of an NC3500-like tree for 37 MB of blocks.  Now through the work-group's LDS
// table, a wave whose lanes share the pose summed first, every touched pose leaving the work-group once)
#define RHSU_CAP 256
__global__ void __launch_bounds__(128) k_join_rhs_u(int NU, const double* __restrict__ U, const int* __restrict__ Ui, const int* __restrict__ Uj,
                                                    const double* __restrict__ pose, double* __restrict__ eP)
{
	__shared__ int ekeys[RHSU_CAP];
	__shared__ double evals[RHSU_CAP * 6];
	for (int q = threadIdx.x; q < RHSU_CAP; q += blockDim.x) ekeys[q] = -1;
	for (int q = threadIdx.x; q < RHSU_CAP * 6; q += blockDim.x) evals[q] = 0.0;
	__syncthreads();
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	const bool have = i < NU;
	int a = 0, b = 0;
	double sa[6] = { 0, 0, 0, 0, 0, 0 }, sb[6] = { 0, 0, 0, 0, 0, 0 };
	if (have)
	{
		a = Ui[i]; b = Uj[i];
		double u[36];
		ld<36>(u, U + (size_t)i * 36);
		const double* xb = pose + (size_t)b * 6;
		for (int r = 0; r < 6; r++)
		{
			double s = 0;
			for (int c = 0; c < 6; c++) s = fma(u[6 * r + c], xb[c], s);
			sa[r] = s;
		}
		if (a != b)
		{
			const double* xa = pose + (size_t)a * 6;
			for (int c = 0; c < 6; c++)
			{
				double s = 0;
				for (int r = 0; r < 6; r++) s = fma(u[6 * r + c], xa[r], s);
				sb[c] = s;
			}
		}
	}
	tile_scatter_add<6>(ekeys, evals, RHSU_CAP, a, eP + (size_t)a * 6, sa, have);
	tile_scatter_add<6>(ekeys, evals, RHSU_CAP, b, eP + (size_t)b * 6, sb, have && a != b);
	__syncthreads();
	tile_flush<6>(ekeys, evals, RHSU_CAP, eP);
}

__global__ void k_shift_segments(int n, const int* __restrict__ map_of, int* __restrict__ seg)
{
	int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) seg[i] = map_of[i] >> 1;
}

// A Stereo join in two steps, so that a caller (the tree scheduler) can let the transform write its W blocks straight into
// the joint map: join_stereo_prepare() needs the input maps without their W values (labels, estimates, V, run pointers)
// and lays the joint map out; st.wbase[f] is where the run of input feature f starts in the joint W arrays.  Whoever
// fills out.W / photo / feature / st.srcf from there (k_join_wcopy here, or the transform's block kernel) is followed by
// join_stereo_finish(): right-hand sides and the solve.  Scratch taken in prepare is released by finish.
void join_stereo_prepare(lsfm_context* ctx, Arena& ar, const DevBatch& in, DevBatch& out, JoinState& st)
{
	hipStream_t s = ctx->stream;
	const int B = in.B, G = (B + 1) / 2;
	st.smark = ctx->scratch.mark();
	st.ar = &ar;

	// ---- common features (K5) ---- (from the level's plan when it holds them: LevelIndex, lsfm_internal.hpp)
	static const bool reuse_index = !getenv("LSFM_NO_INDEX_REUSE");
	LevelPlan* plan = ctx->plan;
	const int *match = nullptr, *R = nullptr;
	const int nb = (in.NF + 255) / 256;
	std::vector<int> rb(B + 1);
	if (reuse_index && ctx->warm() && plan->idx.match && plan->idx.R && plan->idx.NF == in.NF)
	{
		match = plan->idx.match; R = plan->idx.R;
		rb = plan->join_rb;
	}
	else
	{
		int* mt = ctx->scratch.alloc<int>(in.NF + 1);
		int* unm = ctx->scratch.alloc<int>(in.NF + 2);
		int* Rw = ctx->scratch.alloc<int>(in.NF + 2);
		if (in.NF) join_match_features(ctx, in, mt, unm);
		else dev_zero(ctx, unm, 2 * sizeof(int));
		dev_exclusive_scan(ctx, unm, Rw, in.NF);
		match = mt; R = Rw;
		if (ctx->warm()) rb = plan->join_rb; // known from an earlier run of the same tree
		else
		{
			// unmatched counts per map -> joint feature offsets (host)
			int* d_rb = ctx->scratch.alloc<int>(B + 1);
			hipLaunchKernelGGL(k_gather_at, dim3((B + 1 + 127) / 128), dim3(128), 0, s, R, in.d_feat_off, B + 1, d_rb);
			ctx->mark("jn_enq");
			d2h_ints(ctx, d_rb, rb.data(), B + 1);
			ctx->mark("jn_rb");
			if (plan) plan->join_rb = rb;
			if (reuse_index && plan && plan != &ctx->pre_plan && ctx->in_tree_run)
			{
				// a resident tree records the level: its later runs skip the matching and its scan
				plan->idx.match = level_index_keep(ctx, plan->idx, mt, (size_t)in.NF + 1);
				plan->idx.R = level_index_keep(ctx, plan->idx, Rw, (size_t)in.NF + 2);
				plan->idx.NF = in.NF;
			}
		}
	}

	out = DevBatch();
	out.B = G; out.M = in.M; out.NU = in.NU; out.NW = in.NW;
	out.pose_off.assign(G + 1, 0); out.feat_off.assign(G + 1, 0); out.u_off.assign(G + 1, 0); out.w_off.assign(G + 1, 0);
	out.Ref.resize(G); out.FRef.resize(G); out.ScaP.assign(G, 0); out.Fix.assign(G, 0); out.Sign.assign(G, 1); out.FScaP.assign(G, 0); out.FFix.assign(G, 0);
	std::vector<JGroup> grp(G);
	std::vector<unsigned char>& seg_active = st.seg_active;
	std::vector<int>& seg_rows = st.seg_rows;
	seg_active.assign(G, 0); seg_rows.assign(G, 0);
	for (int g = 0; g < G; g++)
	{
		const int a = 2 * g, b = 2 * g + 1;
		const bool pair = b < B;
		JGroup& jg = grp[g];
		jg.F0E = in.feat_off[a]; jg.nE = in.feat_off[a + 1] - jg.F0E;
		jg.F0C = pair ? in.feat_off[b] : in.feat_off[a + 1]; jg.nC = pair ? in.feat_off[b + 1] - jg.F0C : 0;
		jg.FY0 = out.feat_off[g];
		jg.rC0 = pair ? rb[b] : 0;
		const int nun = pair ? rb[b + 1] - rb[b] : 0;
		out.feat_off[g + 1] = jg.FY0 + jg.nE + nun;
		out.pose_off[g] = in.pose_off[a]; out.u_off[g] = in.u_off[a]; out.w_off[g] = in.w_off[a];
		out.Ref[g] = pair ? in.Ref[b] : in.Ref[a];     // Imp.cpp:2974
		out.FRef[g] = in.FRef[a];                      // Imp.cpp:2624
		seg_active[g] = pair ? 1 : 0;
		seg_rows[g] = (pair ? in.pose_off[b + 1] : in.pose_off[a + 1]) - in.pose_off[a];
	}
	out.pose_off[G] = in.M; out.u_off[G] = in.NU; out.w_off[G] = in.NW;
	out.NF = out.feat_off[G];
	const int NFY = out.NF;

	CopyBatch cb(ctx); // group records, offsets, label arrays, activity flags: one transfer, one kernel
	JGroup* d_grp = ctx->scratch.alloc<JGroup>(G);
	cb.h2d(d_grp, grp.data(), sizeof(JGroup) * G);

	// ---- joint arrays (K6) ----
	out.pose = ar.alloc<double>((size_t)in.M * 6);
	out.pose_id = ar.alloc<int>(in.M);
	out.pose_origin = ar.alloc<int>(in.M);
	cb.d2d(out.pose_origin, in.pose_origin, (size_t)in.M * sizeof(int));
	out.feat = ar.alloc<double>((size_t)NFY * 3);
	out.feat_id = ar.alloc<int>(NFY);
	out.U = ar.alloc<double>((size_t)in.NU * 36); out.Ui = ar.alloc<int>(in.NU); out.Uj = ar.alloc<int>(in.NU);
	out.W = ar.alloc<double>((size_t)in.NW * 18); out.photo = ar.alloc<int>(in.NW); out.feature = ar.alloc<int>(in.NW);
	out.fptr = ar.alloc<int>(NFY + 1);
	out.V = ar.alloc<double>((size_t)NFY * 9);
	batch_set_offsets(ctx, ar, out, &cb);
	cb.d2d(out.pose_id, in.pose_id, (size_t)in.M * sizeof(int));
	// (the activity flags of the pairs, for the solve: 4-byte units)
	st.d_act = reinterpret_cast<unsigned char*>(ctx->scratch.alloc<int>((G + 3) / 4 + 1));
	st.act_padded.assign(((size_t)G + 3) / 4 * 4, 0);
	std::copy(seg_active.begin(), seg_active.end(), st.act_padded.begin());
	cb.h2d(st.d_act, st.act_padded.data(), st.act_padded.size());
	cb.flush();
	batch_fill_maps(ctx, out);
	int* newf = st.newf = ctx->scratch.alloc<int>(in.NF + 1);
	int* lens = ctx->scratch.alloc<int>(NFY + 2);
	st.srcf = ctx->scratch.alloc<int>(in.NW + 1);
	// (every entry of lenE / lenC / eF / out.V gets its first value from k_join_features: only the pose part of the right-hand
	// side is an accumulator that starts from zero)
	int* lenE = st.lenE = ctx->scratch.alloc<int>(NFY + 1);
	int* lenC = ctx->scratch.alloc<int>(NFY + 1);
	double* eF = st.eF = ctx->scratch.alloc<double>((size_t)NFY * 3);
	double* eP = st.eP = ctx->scratch.alloc<double>((size_t)in.M * 6);
	dev_zero(ctx, eP, (size_t)in.M * 6 * sizeof(double));
	// a level that analyses, reached through the transform's hook: the sources of every joint feature, for the early pattern of S
	static const bool early_on = !getenv("LSFM_NO_EARLY_PATTERN");
	// (a level of small systems takes the dense path: no pattern at all)
	int most_rows = 0;
	for (int r : seg_rows) most_rows = std::max(most_rows, r);
	static const bool no_small = getenv("LSFM_NO_SMALL") != nullptr;
	const bool small_level = ctx->small_max > 0 && !ctx->comm && !ctx->pcg.mixed && !no_small && small_solve_strips(most_rows, ctx->small_max) > 0;
	const bool early = early_on && !small_level && !ctx->comm && !ctx->pre && ctx->tr_in && ctx->tr_hub && !ctx->warm() && ctx->tr_in->NF == in.NF && ctx->tr_in->M == in.M;
	// the W part of the right-hand sides left to the Schur assembly (lsfm_solve.hpp RhsFused): a level on the sparse pipeline, one GPU
	static const bool fuse_on = !getenv("LSFM_NO_FUSED_RHS");
	st.fuse_rhs = fuse_on && !small_level && !ctx->comm && !ctx->pcg.mixed;
	int *srcE = nullptr, *srcC = nullptr;
	if (early || st.fuse_rhs)
	{
		srcE = ctx->scratch.alloc<int>(NFY + 1); srcC = ctx->scratch.alloc<int>(NFY + 1); // (filled by k_join_features)
	}
	st.srcE = srcE; st.srcC = srcC;
	if (in.NF)
		for (int side = 0; side < 2; side++)
			hipLaunchKernelGGL(k_join_features, dim3(nb), dim3(256), 0, s, in.NF, in.feat_map, in.feat_id, in.feat, in.V, in.fptr, match, R, d_grp,
			                   newf, lenE, lenC, out.V, eF, out.feat_id, out.feat, srcE, srcC, side);
	if (early)
	{
		const DevBatch& X = *ctx->tr_in; // the level's input: its W runs and U blocks with the poses they had before the transform
		EarlyPatternIn ei;
		ei.M = in.M; ei.NFY = NFY; ei.NU = X.NU;
		ei.Ui = X.Ui; ei.Uj = X.Uj; ei.pose_map = X.pose_map; ei.hub = ctx->tr_hub;
		ei.fptr = X.fptr; ei.photo = X.photo; ei.feat_map = X.feat_map; ei.srcE = srcE; ei.srcC = srcC;
		ei.prev_keys = X.s_keys; ei.prev_nnzb = X.s_nnzb;
		LSFM_CHECK_HIP(hipEventRecord(ctx->evC, s));
		schur_pattern_early_issue(ctx, ei);
	}
	else schur_pattern_early_drop(ctx);
	hipLaunchKernelGGL(k_add_lens, dim3((NFY + 256) / 256), dim3(256), 0, s, NFY, lenE, lenC, lens);
	dev_exclusive_scan(ctx, lens, out.fptr, NFY);
	if (early) LSFM_CHECK_HIP(hipEventRecord(ctx->evC, s)); // the joint run pointers: the second half of the early pattern reads them
	// where the run of every input feature starts in the joint map
	st.wbase = ctx->scratch.alloc<int>(in.NF + 1);
	if (in.NF) hipLaunchKernelGGL(k_join_wbase, dim3(nb), dim3(256), 0, s, in.NF, in.feat_map, newf, lenE, out.fptr, st.wbase);
	LSFM_CHECK_HIP(hipGetLastError());
	ctx->mark("jn_prep");
}

void join_stereo_finish(lsfm_context* ctx, const DevBatch& in, DevBatch& out, JoinState& st, double* eP_out, double* eF_out)
{
	hipStream_t s = ctx->stream;
	const int G = out.B, NFY = out.NF;
	double *eP = st.eP, *eF = st.eF;
	int* srcf = st.srcf;
	const std::vector<unsigned char>& seg_active = st.seg_active;
	const std::vector<int>& seg_rows = st.seg_rows;
	// U is reused unchanged (global pose indices); copied here because the caller may still have been producing it
	// (the transform's U stage) while the joint map was laid out
	if (in.NU)
	{
		CopyBatch cu(ctx);
		cu.d2d(out.U, in.U, (size_t)in.NU * 36 * sizeof(double));
		cu.d2d(out.Ui, in.Ui, (size_t)in.NU * sizeof(int));
		cu.d2d(out.Uj, in.Uj, (size_t)in.NU * sizeof(int));
		cu.flush();
	}
	LSFM_CHECK_HIP(hipEventRecord(ctx->evY, s)); // the joint maps' index arrays are final (prefetch_next_level reads them)
	// everything the pattern of S needs is enqueued: the solve may build it beside the right-hand sides (unless it is under
	// way already: schur_pattern_early_issue)
	if (!eP_out && !eF_out && !ctx->early && !ctx->comm)
	{
		LSFM_CHECK_HIP(hipEventRecord(ctx->evA, s));
		ctx->pattern_dep = true;
	}
	// ---- right-hand sides ----
	// (the W part: a pass over W of its own -- unless the Schur assembly takes it along, K9Out; a caller that wants eP / eF gets them whole)
	const bool fuse_rhs = st.fuse_rhs && !eP_out && !eF_out;
	if (NFY && !fuse_rhs)
		hipLaunchKernelGGL(k_join_rhs_w, dim3((NFY + RHS_TILE - 1) / RHS_TILE), dim3(256), 0, s, NFY, out.fptr, out.W, out.photo, srcf, in.pose, in.feat, eP, eF);
	if (in.NU && (!ctx->comm || ctx->comm->rank == 0)) // (feature-sharded run: U is replicated, its part of the sum is rank 0's)
		hipLaunchKernelGGL(k_join_rhs_u, dim3((in.NU + 127) / 128), dim3(128), 0, s, in.NU, in.U, in.Ui, in.Uj, in.pose, eP);
	LSFM_CHECK_HIP(hipGetLastError());
	if (eP_out) d2h(ctx, eP_out, eP, (size_t)in.M * 6 * sizeof(double));
	if (eF_out) d2h(ctx, eF_out, eF, (size_t)NFY * 3 * sizeof(double));

	// ---- solve (K7-K11) ----
	unsigned char* d_act = st.d_act;
	SolveIO io;
	io.M = in.M; io.NF = NFY; io.NU = in.NU; io.NW = in.NW; io.nseg = G;
	io.d_pose_seg = out.pose_map; io.d_feat_seg = out.feat_map; io.d_seg_active = d_act;
	io.U = out.U; io.Ui = out.Ui; io.Uj = out.Uj; io.W = out.W; io.photo = out.photo; io.fptr = out.fptr; io.V = out.V;
	io.ea = eP; io.eb = eF; io.x0 = in.pose; io.d_fixed = nullptr; io.d_pose_origin = out.pose_origin;
	io.x_pose = out.pose; io.x_feat = out.feat;
	io.seg_rows = seg_rows;
	RhsFused rhs;
	if (fuse_rhs)
	{
		rhs.srcE = st.srcE; rhs.srcC = st.srcC; rhs.feat_src = in.feat; rhs.pose_src = in.pose; rhs.pose_map_src = in.pose_map;
		io.rhs = &rhs;
	}
	{
		// a level of small systems goes to the one-launch dense path, which walks the joins by their ranges (lsfm_small.hip)
		int most = 0;
		for (int r : seg_rows) most = std::max(most, r);
		if (ctx->small_max > 0 && small_solve_strips(most, ctx->small_max))
		{
			int* d_uo = ctx->scratch.alloc<int>(G + 1);
			h2d(ctx, d_uo, out.u_off.data(), sizeof(int) * (size_t)(G + 1));
			io.d_pose_off = out.d_pose_off; io.d_feat_off = out.d_feat_off; io.d_u_off = d_uo;
		}
	}
	const bool warm = ctx->warm();
	ctx->solved_keys = nullptr; ctx->solved_nnzb = 0;
	int rc = solve_batch(ctx, io);
	if (!warm && st.ar && ctx->in_tree_run && ctx->solved_keys && getenv("LSFM_NO_PREFETCH") && !getenv("LSFM_NO_EARLY_PATTERN"))
	{
		// the pattern of this level's system stays with its output for the level above (schur_pattern_early_issue)
		unsigned long long* k = st.ar->alloc<unsigned long long>((size_t)ctx->solved_nnzb + 1);
		LSFM_CHECK_HIP(hipMemcpyAsync(k, ctx->solved_keys, (size_t)ctx->solved_nnzb * sizeof(unsigned long long), hipMemcpyDeviceToDevice, s));
		out.s_keys = k; out.s_nnzb = ctx->solved_nnzb;
	}
	// a level of a tree run is only enqueued (its scratch is reused in stream order, errors are read at the end of the run); a
	// stage-level call stops here so that a failure surfaces at its stage
	if (!warm && !ctx->in_tree_run) LSFM_CHECK_HIP(hipStreamSynchronize(s));
	ctx->scratch.release(st.smark);
	if (rc > 0 && ctx->stats) ctx->stats->not_converged += rc;
	if (ctx->plan && !eP_out && !eF_out) ctx->plan->valid = true; // every stage of the level has left its structure behind
}

void join_batch_stereo(lsfm_context* ctx, Arena& ar, const DevBatch& in, DevBatch& out, double* eP_out, double* eF_out)
{
	JoinState st;
	join_stereo_prepare(ctx, ar, in, out, st);
	if (in.NW)
		hipLaunchKernelGGL(k_join_wcopy, dim3((in.NW + 255) / 256), dim3(256), 0, ctx->stream, in.NW, in.W, in.photo, in.feature, in.fptr, st.wbase,
		                   st.newf, out.W, out.photo, out.feature, st.srcf, in.W_alias, in.d_alias, in.feat_map);
	join_stereo_finish(ctx, in, out, st, eP_out, eF_out);
}

} // namespace lsfm
