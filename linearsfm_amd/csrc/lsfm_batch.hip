// Host maps (reference layout, Imp.h:75-178) <-> device batch (flat SoA with global indices).
#include <cstdlib>

#include "lsfm_device.hpp"
#include "lsfm_internal.hpp"

namespace lsfm {

__global__ void k_fill_segment_ids(const int* __restrict__ off, int B, int* __restrict__ seg, int total)
{
	int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= total) return;
	int lo = 0, hi = B; // largest b with off[b] <= i
	while (hi - lo > 1) { int mid = (lo + hi) >> 1; if (off[mid] <= i) lo = mid; else hi = mid; }
	seg[i] = lo;
}

void batch_set_offsets(lsfm_context* ctx, Arena& ar, DevBatch& b, CopyBatch* cb)
{
	b.d_pose_off = ar.alloc<int>(b.B + 1);
	b.d_feat_off = ar.alloc<int>(b.B + 1);
	b.pose_map = ar.alloc<int>(b.M);
	b.feat_map = ar.alloc<int>(b.NF);
	if (cb)
	{
		// the caller's batch carries the two offset arrays; it calls batch_fill_maps once the batch is flushed
		cb->h2d(b.d_pose_off, b.pose_off.data(), (b.B + 1) * sizeof(int));
		cb->h2d(b.d_feat_off, b.feat_off.data(), (b.B + 1) * sizeof(int));
		return;
	}
	h2d(ctx, b.d_pose_off, b.pose_off.data(), (b.B + 1) * sizeof(int));
	h2d(ctx, b.d_feat_off, b.feat_off.data(), (b.B + 1) * sizeof(int));
	batch_fill_maps(ctx, b);
}
// the map of every pose and of every feature, one launch (two until round 6: 5 us each in the chain of a level's small kernels)
__global__ void k_fill_segment_ids2(const int* __restrict__ offA, int* __restrict__ segA, int totalA, const int* __restrict__ offB,
                                    int* __restrict__ segB, int totalB, int B)
{
	int i = blockIdx.x * blockDim.x + threadIdx.x;
	const bool second = i >= totalA;
	if (second) i -= totalA;
	if (second && i >= totalB) return;
	const int* off = second ? offB : offA;
	int lo = 0, hi = B; // largest b with off[b] <= i
	while (hi - lo > 1) { int mid = (lo + hi) >> 1; if (off[mid] <= i) lo = mid; else hi = mid; }
	(second ? segB : segA)[i] = lo;
}
void batch_fill_maps(lsfm_context* ctx, DevBatch& b)
{
	const size_t n = (size_t)b.M + b.NF;
	if (n) hipLaunchKernelGGL(k_fill_segment_ids2, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, b.d_pose_off, b.pose_map, b.M, b.d_feat_off, b.feat_map, b.NF, b.B);
}

void batch_upload(lsfm_context* ctx, Arena& ar, const lsfm_map* maps, int N, bool mono, DevBatch& o)
{
	o = DevBatch();
	o.B = N;
	o.pose_off.assign(N + 1, 0); o.feat_off.assign(N + 1, 0); o.u_off.assign(N + 1, 0); o.w_off.assign(N + 1, 0);
	o.Ref.resize(N); o.FRef.resize(N); o.ScaP.assign(N, 0); o.Fix.assign(N, 0); o.Sign.assign(N, 1); o.FScaP.assign(N, 0); o.FFix.assign(N, 0);
	for (int k = 0; k < N; k++)
	{
		const lsfm_map& g = maps[k];
		if (g.m < 0 || g.n < 0 || g.nU < 0 || g.nW < 0) LSFM_FAIL(LSFM_ERR_ARG, "negative map size");
		o.pose_off[k + 1] = o.pose_off[k] + g.m; o.feat_off[k + 1] = o.feat_off[k] + g.n;
		o.u_off[k + 1] = o.u_off[k] + g.nU; o.w_off[k + 1] = o.w_off[k] + g.nW;
		o.Ref[k] = g.Ref; o.FRef[k] = g.FRef;
		if (mono) { o.ScaP[k] = g.ScaP; o.Fix[k] = g.Fix; o.Sign[k] = g.Sign; o.FScaP[k] = g.FScaP; o.FFix[k] = g.FFix; }
	}
	o.M = o.pose_off[N]; o.NF = o.feat_off[N]; o.NU = o.u_off[N]; o.NW = o.w_off[N];
	std::vector<double> pose((size_t)o.M * 6), feat((size_t)o.NF * 3);
	std::vector<int> pid(o.M), porg(o.M), fid(o.NF), Ui(o.NU), Uj(o.NU), photo(o.NW), feature(o.NW), fptr(o.NF + 1);
	for (int k = 0; k < N; k++)
	{
		const lsfm_map& g = maps[k];
		int po = o.pose_off[k], fo = o.feat_off[k], uo = o.u_off[k], wo = o.w_off[k];
		for (int i = 0; i < g.m; i++)
		{
			if (g.stno[6 * i] > 0) LSFM_FAIL(LSFM_ERR_ARG, "state label of a pose must be <= 0");
			pid[po + i] = -g.stno[6 * i];
			porg[po + i] = g.pose_origin ? g.pose_origin[i] : k;
			memcpy(&pose[(size_t)(po + i) * 6], g.stVal + 6 * i, 6 * sizeof(double));
		}
		for (int i = 0; i < g.n; i++)
		{
			if (g.stno[6 * g.m + 3 * i] <= 0) LSFM_FAIL(LSFM_ERR_ARG, "state label of a feature must be > 0");
			fid[fo + i] = g.stno[6 * g.m + 3 * i];
			memcpy(&feat[(size_t)(fo + i) * 3], g.stVal + 6 * g.m + 3 * i, 3 * sizeof(double));
		}
		for (int i = 0; i < g.nU; i++)
		{
			if (g.Ui[i] < 0 || g.Uj[i] >= g.m || g.Ui[i] > g.Uj[i]) LSFM_FAIL(LSFM_ERR_ARG, "U block coordinates must satisfy 0 <= Ui <= Uj < m");
			Ui[uo + i] = g.Ui[i] + po; Uj[uo + i] = g.Uj[i] + po;
		}
		// W must be sorted by feature and every feature must own at least one block (the reference's solver
		// derives the run lengths from feature[] under the same assumption, Imp.cpp:2134-2153)
		int j = 0;
		for (int f = 0; f < g.n; f++)
		{
			fptr[fo + f] = wo + j;
			int j0 = j;
			while (j < g.nW && g.feature[j] == f) j++;
			if (j == j0) LSFM_FAIL(LSFM_ERR_ARG, "every feature needs at least one W block, W sorted by feature");
		}
		if (j != g.nW) LSFM_FAIL(LSFM_ERR_ARG, "W is not sorted by feature");
		for (int i = 0; i < g.nW; i++)
		{
			if (g.photo[i] < 0 || g.photo[i] >= g.m) LSFM_FAIL(LSFM_ERR_ARG, "photo index out of range");
			photo[wo + i] = g.photo[i] + po; feature[wo + i] = g.feature[i] + fo;
		}
	}
	fptr[o.NF] = o.NW;
	o.pose = ar.alloc<double>((size_t)o.M * 6); o.pose_id = ar.alloc<int>(o.M); o.pose_origin = ar.alloc<int>(o.M);
	o.feat = ar.alloc<double>((size_t)o.NF * 3); o.feat_id = ar.alloc<int>(o.NF);
	o.U = ar.alloc<double>((size_t)o.NU * 36); o.Ui = ar.alloc<int>(o.NU); o.Uj = ar.alloc<int>(o.NU);
	o.W = ar.alloc<double>((size_t)o.NW * 18); o.photo = ar.alloc<int>(o.NW); o.feature = ar.alloc<int>(o.NW);
	o.fptr = ar.alloc<int>(o.NF + 1); o.V = ar.alloc<double>((size_t)o.NF * 9);
	h2d(ctx, o.pose, pose.data(), pose.size() * sizeof(double)); h2d(ctx, o.pose_id, pid.data(), pid.size() * sizeof(int));
	h2d(ctx, o.pose_origin, porg.data(), porg.size() * sizeof(int));
	h2d(ctx, o.feat, feat.data(), feat.size() * sizeof(double)); h2d(ctx, o.feat_id, fid.data(), fid.size() * sizeof(int));
	h2d(ctx, o.Ui, Ui.data(), Ui.size() * sizeof(int)); h2d(ctx, o.Uj, Uj.data(), Uj.size() * sizeof(int));
	h2d(ctx, o.photo, photo.data(), photo.size() * sizeof(int)); h2d(ctx, o.feature, feature.data(), feature.size() * sizeof(int));
	h2d(ctx, o.fptr, fptr.data(), fptr.size() * sizeof(int));
	// the big value arrays: every map's piece of an array, back to back, through the pinned ring -- one stream of large
	// copies per array kind (N pageable copies per kind cost ~0.9 s for 3499 maps: 42 000 small transfers in all)
	{
		std::vector<HostPiece> pu, pw, pv;
		pu.reserve(N); pw.reserve(N); pv.reserve(N);
		for (int k = 0; k < N; k++)
		{
			const lsfm_map& g = maps[k];
			pu.push_back(HostPiece{ g.U, (size_t)g.nU * 36 * sizeof(double) });
			pw.push_back(HostPiece{ g.W, (size_t)g.nW * 18 * sizeof(double) });
			pv.push_back(HostPiece{ g.V, (size_t)g.n * 9 * sizeof(double) });
		}
		h2d_gather(ctx, o.U, pu); h2d_gather(ctx, o.W, pw); h2d_gather(ctx, o.V, pv);
	}
	batch_set_offsets(ctx, ar, o);
}

// ---- packed maps: the hand-off of a tree node between GPUs ---------------------------------------------------------
size_t pack_layout(PackHeader& h)
{
	const size_t m = h.m, n = h.n, nU = h.nU, nW = h.nW;
	const size_t bytes[12] = { m * 48, n * 24, nU * 288, nW * 144, n * 72, m * 4, m * 4, n * 4, nU * 4, nU * 4, nW * 4, (n + 1) * 4 };
	size_t o = sizeof(PackHeader);
	for (int i = 0; i < 12; i++) { h.off[i] = o; o = (o + bytes[i] + 255) & ~(size_t)255; }
	h.total = o;
	return o;
}

__global__ void k_copy_add(const int* __restrict__ src, int n, int add, int* __restrict__ dst)
{
	int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) dst[i] = src[i] + add;
}

// map k of a batch -> one contiguous buffer (indices made local to the map)
void batch_pack_map(lsfm_context* ctx, const DevBatch& b, int k, bool mono, void* dst, size_t cap)
{
	hipStream_t s = ctx->stream;
	PackHeader h;
	memset(&h, 0, sizeof h);
	const int po = b.pose_off[k], fo = b.feat_off[k], uo = b.u_off[k], wo = b.w_off[k];
	h.magic = LSFM_PACK_MAGIC; h.version = 1; h.mono = mono;
	h.m = b.pose_off[k + 1] - po; h.n = b.feat_off[k + 1] - fo; h.nU = b.u_off[k + 1] - uo; h.nW = b.w_off[k + 1] - wo;
	h.Ref = b.Ref[k]; h.FRef = b.FRef[k]; h.ScaP = b.ScaP[k]; h.Fix = b.Fix[k]; h.Sign = b.Sign[k]; h.FScaP = b.FScaP[k]; h.FFix = b.FFix[k];
	if (pack_layout(h) > cap) LSFM_FAIL(LSFM_ERR_ARG, "export buffer too small");
	if (b.W_alias) LSFM_FAIL(LSFM_ERR_INTERNAL, "cannot pack a batch whose W blocks are aliased");
	char* d = static_cast<char*>(dst);
	h2d(ctx, d, &h, sizeof h);
	auto cp = [&](int slot, const void* src, size_t bytes) {
		if (bytes) LSFM_CHECK_HIP(hipMemcpyAsync(d + h.off[slot], src, bytes, hipMemcpyDeviceToDevice, s));
	};
	cp(0, b.pose + (size_t)po * 6, (size_t)h.m * 48); cp(1, b.feat + (size_t)fo * 3, (size_t)h.n * 24);
	cp(2, b.U + (size_t)uo * 36, (size_t)h.nU * 288); cp(3, b.W + (size_t)wo * 18, (size_t)h.nW * 144);
	cp(4, b.V + (size_t)fo * 9, (size_t)h.n * 72); cp(5, b.pose_id + po, (size_t)h.m * 4);
	cp(6, b.pose_origin + po, (size_t)h.m * 4); cp(7, b.feat_id + fo, (size_t)h.n * 4);
	auto sh = [&](int slot, const int* src, int cnt, int add) {
		if (cnt) hipLaunchKernelGGL(k_copy_add, dim3((cnt + 255) / 256), dim3(256), 0, s, src, cnt, add, reinterpret_cast<int*>(d + h.off[slot]));
	};
	sh(8, b.Ui + uo, h.nU, -po); sh(9, b.Uj + uo, h.nU, -po); sh(10, b.photo + wo, h.nW, -po); sh(11, b.fptr + fo, h.n + 1, -wo);
	LSFM_CHECK_HIP(hipGetLastError());
}

// ---- slices of a map by feature label (feature-sharded joins: lsfm_tree_export_slice_*) ------------------------------------
__device__ __forceinline__ int slice_of(int id, int G) { const int r = id % G; return r < 0 ? r + G : r; }
__global__ void k_slice_count(int NF, const int* __restrict__ feat_id, const int* __restrict__ fptr, int G, int* __restrict__ cnt)
{
	const int f = blockIdx.x * blockDim.x + threadIdx.x;
	if (f >= NF) return;
	const int g = slice_of(feat_id[f], G);
	atomicAdd(&cnt[g], 1);
	atomicAdd(&cnt[G + g], fptr[f + 1] - fptr[f]);
}
__global__ void k_slice_flags(int NF, const int* __restrict__ feat_id, const int* __restrict__ fptr, int G, int g, int* __restrict__ flag, int* __restrict__ wlen)
{
	const int f = blockIdx.x * blockDim.x + threadIdx.x;
	if (f > NF) return;
	const bool in = f < NF && slice_of(feat_id[f], G) == g;
	flag[f] = in ? 1 : 0;
	wlen[f] = in ? fptr[f + 1] - fptr[f] : 0;
}
__global__ void k_slice_features(int NF, const int* __restrict__ flag, const int* __restrict__ pos, const int* __restrict__ wpos, const double* __restrict__ feat,
                                 const double* __restrict__ V, const int* __restrict__ feat_id, double* __restrict__ ofeat, double* __restrict__ oV,
                                 int* __restrict__ oid, int* __restrict__ ofptr)
{
	const int f = blockIdx.x * blockDim.x + threadIdx.x;
	if (f > NF) return;
	if (f == NF) { ofptr[pos[NF]] = wpos[NF]; return; }
	if (!flag[f]) return;
	const int p = pos[f];
	for (int c = 0; c < 3; c++) ofeat[(size_t)p * 3 + c] = feat[(size_t)f * 3 + c];
	for (int c = 0; c < 9; c++) oV[(size_t)p * 9 + c] = V[(size_t)f * 9 + c];
	oid[p] = feat_id[f];
	ofptr[p] = wpos[f];
}
__global__ void __launch_bounds__(256)
k_slice_w(int NW, const int* __restrict__ feature, const int* __restrict__ fptr, const int* __restrict__ flag, const int* __restrict__ wpos,
          const double* __restrict__ W, const int* __restrict__ photo, double* __restrict__ oW, int* __restrict__ ophoto)
{
	const int j = blockIdx.x * blockDim.x + threadIdx.x;
	if (j >= NW) return;
	const int f = feature[j];
	if (!flag[f]) return;
	const int d = wpos[f] + (j - fptr[f]);
	double w[18];
	ld<18>(w, W + (size_t)j * 18);
	st<18>(oW + (size_t)d * 18, w);
	ophoto[d] = photo[j];
}

void batch_slice_counts(lsfm_context* ctx, const DevBatch& b, int G, std::vector<int>& nf, std::vector<int>& nw)
{
	if (b.B != 1) LSFM_FAIL(LSFM_ERR_ARG, "slices are cut from a single map");
	const size_t mk = ctx->scratch.mark();
	int* cnt = ctx->scratch.alloc<int>(2 * (size_t)G);
	dev_zero(ctx, cnt, sizeof(int) * 2 * (size_t)G);
	if (b.NF) hipLaunchKernelGGL(k_slice_count, dim3((b.NF + 255) / 256), dim3(256), 0, ctx->stream, b.NF, b.feat_id, b.fptr, G, cnt);
	std::vector<int> h(2 * (size_t)G);
	d2h_ints(ctx, cnt, h.data(), h.size());
	nf.assign(h.begin(), h.begin() + G);
	nw.assign(h.begin() + G, h.end());
	ctx->scratch.release(mk);
}

// pack `slice` of the batch's single map: all poses and U blocks, the slice's features in their order with V and their W runs
void batch_pack_slice(lsfm_context* ctx, const DevBatch& b, bool mono, int G, int g, int nf, int nw, void* dst, size_t cap)
{
	hipStream_t s = ctx->stream;
	if (b.B != 1) LSFM_FAIL(LSFM_ERR_ARG, "slices are cut from a single map");
	if (b.W_alias) LSFM_FAIL(LSFM_ERR_INTERNAL, "cannot pack a batch whose W blocks are aliased");
	PackHeader h;
	memset(&h, 0, sizeof h);
	h.magic = LSFM_PACK_MAGIC; h.version = 1; h.mono = mono;
	h.m = b.M; h.n = nf; h.nU = b.NU; h.nW = nw;
	h.Ref = b.Ref[0]; h.FRef = b.FRef[0]; h.ScaP = b.ScaP[0]; h.Fix = b.Fix[0]; h.Sign = b.Sign[0]; h.FScaP = b.FScaP[0]; h.FFix = b.FFix[0];
	if (pack_layout(h) > cap) LSFM_FAIL(LSFM_ERR_ARG, "export buffer too small");
	char* d = static_cast<char*>(dst);
	h2d(ctx, d, &h, sizeof h);
	auto cp = [&](int slot, const void* src, size_t bytes) {
		if (bytes) LSFM_CHECK_HIP(hipMemcpyAsync(d + h.off[slot], src, bytes, hipMemcpyDeviceToDevice, s));
	};
	// (a single map: its indices are local already)
	cp(0, b.pose, (size_t)h.m * 48); cp(2, b.U, (size_t)h.nU * 288); cp(5, b.pose_id, (size_t)h.m * 4); cp(6, b.pose_origin, (size_t)h.m * 4);
	cp(8, b.Ui, (size_t)h.nU * 4); cp(9, b.Uj, (size_t)h.nU * 4);
	const int NF = b.NF;
	int* flag = ctx->scratch.alloc<int>((size_t)NF + 2);
	int* wlen = ctx->scratch.alloc<int>((size_t)NF + 2);
	int* pos = ctx->scratch.alloc<int>((size_t)NF + 2);
	int* wpos = ctx->scratch.alloc<int>((size_t)NF + 2);
	hipLaunchKernelGGL(k_slice_flags, dim3((NF + 256) / 256), dim3(256), 0, s, NF, b.feat_id, b.fptr, G, g, flag, wlen);
	dev_exclusive_scan(ctx, flag, pos, NF);
	dev_exclusive_scan(ctx, wlen, wpos, NF);
	hipLaunchKernelGGL(k_slice_features, dim3((NF + 256) / 256), dim3(256), 0, s, NF, flag, pos, wpos, b.feat, b.V, b.feat_id,
	                   reinterpret_cast<double*>(d + h.off[1]), reinterpret_cast<double*>(d + h.off[4]), reinterpret_cast<int*>(d + h.off[7]),
	                   reinterpret_cast<int*>(d + h.off[11]));
	if (b.NW)
		hipLaunchKernelGGL(k_slice_w, dim3((b.NW + 255) / 256), dim3(256), 0, s, b.NW, b.feature, b.fptr, flag, wpos, b.W, b.photo,
		                   reinterpret_cast<double*>(d + h.off[3]), reinterpret_cast<int*>(d + h.off[10]));
	LSFM_CHECK_HIP(hipGetLastError());
}

// N packed maps (device) -> one batch with global indices in `ar`
void batch_unpack_maps(lsfm_context* ctx, Arena& ar, const void* const* packed, const PackHeader* hdr, int N, bool mono, DevBatch& o)
{
	hipStream_t s = ctx->stream;
	o = DevBatch();
	o.B = N;
	o.pose_off.assign(N + 1, 0); o.feat_off.assign(N + 1, 0); o.u_off.assign(N + 1, 0); o.w_off.assign(N + 1, 0);
	o.Ref.resize(N); o.FRef.resize(N); o.ScaP.assign(N, 0); o.Fix.assign(N, 0); o.Sign.assign(N, 1); o.FScaP.assign(N, 0); o.FFix.assign(N, 0);
	for (int k = 0; k < N; k++)
	{
		const PackHeader& h = hdr[k];
		o.pose_off[k + 1] = o.pose_off[k] + h.m; o.feat_off[k + 1] = o.feat_off[k] + h.n;
		o.u_off[k + 1] = o.u_off[k] + h.nU; o.w_off[k + 1] = o.w_off[k] + h.nW;
		o.Ref[k] = h.Ref; o.FRef[k] = h.FRef;
		if (mono) { o.ScaP[k] = h.ScaP; o.Fix[k] = h.Fix; o.Sign[k] = h.Sign; o.FScaP[k] = h.FScaP; o.FFix[k] = h.FFix; }
	}
	o.M = o.pose_off[N]; o.NF = o.feat_off[N]; o.NU = o.u_off[N]; o.NW = o.w_off[N];
	o.pose = ar.alloc<double>((size_t)o.M * 6); o.pose_id = ar.alloc<int>(o.M); o.pose_origin = ar.alloc<int>(o.M);
	o.feat = ar.alloc<double>((size_t)o.NF * 3); o.feat_id = ar.alloc<int>(o.NF);
	o.U = ar.alloc<double>((size_t)o.NU * 36); o.Ui = ar.alloc<int>(o.NU); o.Uj = ar.alloc<int>(o.NU);
	o.W = ar.alloc<double>((size_t)o.NW * 18); o.photo = ar.alloc<int>(o.NW); o.feature = ar.alloc<int>(o.NW);
	o.fptr = ar.alloc<int>(o.NF + 1); o.V = ar.alloc<double>((size_t)o.NF * 9);
	for (int k = 0; k < N; k++)
	{
		const PackHeader& h = hdr[k];
		const char* p = static_cast<const char*>(packed[k]);
		const int po = o.pose_off[k], fo = o.feat_off[k], uo = o.u_off[k], wo = o.w_off[k];
		auto cp = [&](void* dst, int slot, size_t bytes) {
			if (bytes) LSFM_CHECK_HIP(hipMemcpyAsync(dst, p + h.off[slot], bytes, hipMemcpyDeviceToDevice, s));
		};
		cp(o.pose + (size_t)po * 6, 0, (size_t)h.m * 48); cp(o.feat + (size_t)fo * 3, 1, (size_t)h.n * 24);
		cp(o.U + (size_t)uo * 36, 2, (size_t)h.nU * 288); cp(o.W + (size_t)wo * 18, 3, (size_t)h.nW * 144);
		cp(o.V + (size_t)fo * 9, 4, (size_t)h.n * 72); cp(o.pose_id + po, 5, (size_t)h.m * 4);
		cp(o.pose_origin + po, 6, (size_t)h.m * 4); cp(o.feat_id + fo, 7, (size_t)h.n * 4);
		auto sh = [&](int* dst, int slot, int cnt, int add) {
			if (cnt) hipLaunchKernelGGL(k_copy_add, dim3((cnt + 255) / 256), dim3(256), 0, s, reinterpret_cast<const int*>(p + h.off[slot]), cnt, add, dst);
		};
		sh(o.Ui + uo, 8, h.nU, po); sh(o.Uj + uo, 9, h.nU, po); sh(o.photo + wo, 10, h.nW, po);
		sh(o.fptr + fo, 11, h.n + (k == N - 1 ? 1 : 0), wo); // the last map also writes fptr[NF] = NW
	}
	// feature[] of every W block from the run pointers
	if (o.NW) hipLaunchKernelGGL(k_fill_segment_ids, dim3((o.NW + 255) / 256), dim3(256), 0, s, o.fptr, o.NF, o.feature, o.NW);
	LSFM_CHECK_HIP(hipGetLastError());
	batch_set_offsets(ctx, ar, o);
}

// order-independent sum of position-dependent 64-bit hashes of an int array
__global__ void k_digest_ints(const int* __restrict__ a, size_t n, unsigned long long salt, unsigned long long* __restrict__ out)
{
	const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	unsigned long long h = 0;
	if (i < n)
	{
		unsigned long long x = (salt + i) * 0x9e3779b97f4a7c15ull ^ (unsigned long long)(unsigned)a[i];
		x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33;
		h = x;
	}
#pragma unroll
	for (int off = 32; off > 0; off >>= 1) h += __shfl_xor(h, off, LSFM_WAVE);
	if ((threadIdx.x & (LSFM_WAVE - 1)) == 0 && h) atomicAdd(out, h);
}
unsigned long long batch_structure_digest(lsfm_context* ctx, const DevBatch& b)
{
	unsigned long long* d = ctx->scratch.alloc<unsigned long long>(1);
	dev_zero(ctx, d, sizeof *d);
	const struct { const int* p; size_t n; } arr[] = { { b.pose_id, (size_t)b.M }, { b.pose_origin, (size_t)b.M }, { b.feat_id, (size_t)b.NF },
		{ b.Ui, (size_t)b.NU }, { b.Uj, (size_t)b.NU }, { b.photo, (size_t)b.NW }, { b.fptr, (size_t)b.NF + 1 } };
	unsigned long long salt = 1;
	for (const auto& a : arr)
	{
		if (a.n) hipLaunchKernelGGL(k_digest_ints, dim3((unsigned)((a.n + 255) / 256)), dim3(256), 0, ctx->stream, a.p, a.n, salt << 40, d);
		salt++;
	}
	unsigned long long h = 0;
	d2h(ctx, &h, d, sizeof h);
	for (int k = 0; k < b.B; k++) // the host mirrors of the map records belong to the structure too
		for (int v : { b.Ref[k], b.FRef[k], b.ScaP[k], b.Fix[k], b.FScaP[k], b.FFix[k], b.pose_off[k + 1], b.feat_off[k + 1], b.u_off[k + 1], b.w_off[k + 1] })
			h = (h ^ (unsigned long long)(unsigned)v) * 0x100000001b3ull;
	return h;
}

template <class T> static T* host_alloc(size_t n) { return static_cast<T*>(malloc((n ? n : 1) * sizeof(T))); }

void batch_download_map(lsfm_context* ctx, const DevBatch& b, int k, bool mono, lsfm_map* g)
{
	memset(g, 0, sizeof *g);
	int po = b.pose_off[k], fo = b.feat_off[k];
	int m = b.pose_off[k + 1] - po, n = b.feat_off[k + 1] - fo;
	// u_off / w_off of the batch are kept on the host by every stage
	int uo = b.u_off[k], wo = b.w_off[k], nU = b.u_off[k + 1] - uo, nW = b.w_off[k + 1] - wo;
	g->m = m; g->n = n; g->nU = nU; g->nW = nW; g->Ref = b.Ref[k]; g->FRef = b.FRef[k];
	if (mono) { g->ScaP = b.ScaP[k]; g->Fix = b.Fix[k]; g->Sign = b.Sign[k]; g->FScaP = b.FScaP[k]; g->FFix = b.FFix[k]; }
	int r = 6 * m + 3 * n;
	g->stno = host_alloc<int>(r); g->stVal = host_alloc<double>(r);
	g->U = host_alloc<double>((size_t)nU * 36); g->Ui = host_alloc<int>(nU); g->Uj = host_alloc<int>(nU);
	g->W = host_alloc<double>((size_t)nW * 18); g->photo = host_alloc<int>(nW); g->feature = host_alloc<int>(nW);
	g->V = host_alloc<double>((size_t)n * 9); g->FBlock = host_alloc<int>(n);
	g->pose_origin = host_alloc<int>(m);
	if (b.pose_origin) d2h(ctx, g->pose_origin, b.pose_origin + po, m * sizeof(int));
	else for (int i = 0; i < m; i++) g->pose_origin[i] = k;
	std::vector<int> pid(m), fid(n), fptr(n + 1);
	d2h(ctx, pid.data(), b.pose_id + po, m * sizeof(int));
	d2h(ctx, fid.data(), b.feat_id + fo, n * sizeof(int));
	d2h(ctx, g->stVal, b.pose + (size_t)po * 6, (size_t)m * 6 * sizeof(double));
	d2h(ctx, g->stVal + 6 * m, b.feat + (size_t)fo * 3, (size_t)n * 3 * sizeof(double));
	for (int i = 0; i < m; i++) for (int c = 0; c < 6; c++) g->stno[6 * i + c] = -pid[i];
	for (int i = 0; i < n; i++) for (int c = 0; c < 3; c++) g->stno[6 * m + 3 * i + c] = fid[i];
	d2h(ctx, g->U, b.U + (size_t)uo * 36, (size_t)nU * 36 * sizeof(double));
	d2h(ctx, g->Ui, b.Ui + uo, nU * sizeof(int)); d2h(ctx, g->Uj, b.Uj + uo, nU * sizeof(int));
	for (int i = 0; i < nU; i++) { g->Ui[i] -= po; g->Uj[i] -= po; }
	d2h(ctx, g->W, b.W + (size_t)wo * 18, (size_t)nW * 18 * sizeof(double));
	d2h(ctx, g->photo, b.photo + wo, nW * sizeof(int)); d2h(ctx, g->feature, b.feature + wo, nW * sizeof(int));
	for (int i = 0; i < nW; i++) { g->photo[i] -= po; g->feature[i] -= fo; }
	d2h(ctx, g->V, b.V + (size_t)fo * 9, (size_t)n * 9 * sizeof(double));
	d2h(ctx, fptr.data(), b.fptr + fo, (n + 1) * sizeof(int));
	for (int i = 0; i < n; i++) g->FBlock[i] = fptr[i] - wo;
}

// ---- measurement: the W access patterns against the stream copy (lsfm_wstream_bench) ------------------------------
__global__ void __launch_bounds__(256) k_wstream_block18(long long n, const double* __restrict__ a, double* __restrict__ b)
{
	const long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x;
	if (j >= n) return;
	double w[18];
	ld<18>(w, a + j * 18);
	st<18>(b + j * 18, w);
}
__global__ void __launch_bounds__(256) k_wstream_block9x16(long long n, const double2* __restrict__ a, double2* __restrict__ b)
{
	const long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x;
	if (j >= n) return;
	double2 w[9];
#pragma unroll
	for (int i = 0; i < 9; i++) w[i] = a[j * 9 + i];
#pragma unroll
	for (int i = 0; i < 9; i++) b[j * 9 + i] = w[i];
}
__global__ void __launch_bounds__(256) k_wstream_linear(long long n2, const double2* __restrict__ a, double2* __restrict__ b)
{
	// four 16-byte pieces per lane, consecutive lanes on consecutive pieces
	const long long base = (long long)blockIdx.x * blockDim.x * 4 + threadIdx.x;
	double2 w[4];
#pragma unroll
	for (int i = 0; i < 4; i++) if (base + (long long)i * blockDim.x < n2) w[i] = a[base + (long long)i * blockDim.x];
#pragma unroll
	for (int i = 0; i < 4; i++) if (base + (long long)i * blockDim.x < n2) b[base + (long long)i * blockDim.x] = w[i];
}
int wstream_bench(lsfm_context* ctx, long long nblocks, int mode, int reps, double* avg_ms)
{
	hipStream_t s = ctx->stream;
	double* a = ctx->scratch.alloc<double>((size_t)nblocks * 18);
	double* b = ctx->scratch.alloc<double>((size_t)nblocks * 18);
	fill_async(s, a, 0, (size_t)nblocks * 144);
	float total = 0;
	for (int k = 0; k < reps + 2; k++)
	{
		LSFM_CHECK_HIP(hipEventRecord(ctx->ev0, s));
		if (mode == 0) hipLaunchKernelGGL(k_wstream_block18, dim3((unsigned)((nblocks + 255) / 256)), dim3(256), 0, s, nblocks, a, b);
		else if (mode == 1) hipLaunchKernelGGL(k_wstream_block9x16, dim3((unsigned)((nblocks + 255) / 256)), dim3(256), 0, s, nblocks, (const double2*)a, (double2*)b);
		else hipLaunchKernelGGL(k_wstream_linear, dim3((unsigned)((nblocks * 9 + 1023) / 1024)), dim3(256), 0, s, nblocks * 9, (const double2*)a, (double2*)b);
		LSFM_CHECK_HIP(hipEventRecord(ctx->ev1, s));
		LSFM_CHECK_HIP(hipEventSynchronize(ctx->ev1));
		float t = 0;
		LSFM_CHECK_HIP(hipEventElapsedTime(&t, ctx->ev0, ctx->ev1));
		if (k >= 2) total += t;
	}
	*avg_ms = total / reps;
	return LSFM_OK;
}

} // namespace lsfm

extern "C" void lsfm_map_release(lsfm_map* g)
{
	if (!g) return;
	free(g->stno); free(g->stVal); free(g->U); free(g->Ui); free(g->Uj); free(g->W); free(g->photo); free(g->feature);
	free(g->V); free(g->FBlock); free(g->pose_origin);
	memset(g, 0, sizeof *g);
}
