// Self-test of liblsfm_rccl on ONE GPU (a communicator of one rank): the feature-sharded top levels of a tree with RCCL as the
// all-reduce, driven from C++ alone -- what INTEGRATION.md section 3 shows a C host doing.  The local maps of a directory are
// joined twice: as one tree (lsfm_divide_conquer), and as two blocks whose roots are packed into slices (one slice: all
// features) and joined by a top tree that runs with lsfm_tree_set_comm -- every sum of its levels goes through ncclAllReduce on
// the library's stream.  The two results must agree.
//   lsfm_rccl_selftest -path <dir> -num <N> -type Stereo|Monocular
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/lsfm.h"
#include "../../include/lsfm_rccl.h"

#define CHECK(x) do { int rc__ = (x); if (rc__ < 0) { fprintf(stderr, "%s failed (%d): %s\n", #x, rc__, ctx ? lsfm_last_error(ctx) : ""); return 2; } } while (0)

int main(int argc, char** argv)
{
	const char* path = nullptr; int N = 0; bool mono = false;
	for (int i = 1; i + 1 < argc; i += 2)
	{
		if (!strcmp(argv[i], "-path")) path = argv[i + 1];
		else if (!strcmp(argv[i], "-num")) N = atoi(argv[i + 1]);
		else if (!strcmp(argv[i], "-type")) mono = !strcmp(argv[i + 1], "Monocular");
	}
	if (!path || N < 2) { fprintf(stderr, "usage: lsfm_rccl_selftest -path <dir> -num <N >= 2> -type Stereo|Monocular\n"); return 1; }
	lsfm_context* ctx = nullptr;
	std::vector<lsfm_map> maps(N);
	int failed = 0;
	if (lsfm_read_localmaps(path, 1, N, mono, 0, maps.data(), &failed)) { fprintf(stderr, "cannot read localmap_%d.txt\n", failed); return 1; }
	CHECK(lsfm_context_create(0, 0, &ctx));
	// pose origins: index of the local map in the WHOLE set (drives the elimination order; a block does not start at 0)
	std::vector<std::vector<int>> origin(N);
	for (int k = 0; k < N; k++) { origin[k].assign(maps[k].m, k); maps[k].pose_origin = origin[k].data(); }
	lsfm_map ref;
	lsfm_stats st;
	CHECK(lsfm_divide_conquer(ctx, maps.data(), N, mono, &ref, &st));
	int size = 1;
	while (size * 2 < N) size *= 2; // two blocks, the first a whole sub-tree (linearsfm_amd.distributed.shard_bounds(N, 2))
	const int lo[2] = { 0, size }, cnt[2] = { size, N - size };
	void* pack[2] = { nullptr, nullptr };
	for (int b = 0; b < 2; b++)
	{
		lsfm_tree* t = nullptr;
		CHECK(lsfm_tree_upload(ctx, maps.data() + lo[b], cnt[b], mono, &t));
		lsfm_tree_set_final_reanchor(t, 0); // the top tree's first level takes the odd root back to its first frame itself
		CHECK(lsfm_tree_run(ctx, t, &st));
		size_t bytes = 0;
		CHECK(lsfm_tree_export_slice_sizes(ctx, t, 1, &bytes));
		if (hipMalloc(&pack[b], bytes) != hipSuccess) return 2;
		CHECK(lsfm_tree_export_slice_dev(ctx, t, 1, 0, pack[b], bytes));
		lsfm_tree_free(ctx, t);
	}
	lsfm_tree* top = nullptr;
	CHECK(lsfm_tree_upload_dev(ctx, pack, 2, mono, &top));
	std::vector<unsigned char> id(lsfm_rccl_unique_id_bytes());
	if (lsfm_rccl_unique_id(id.data(), id.size())) { fprintf(stderr, "ncclGetUniqueId failed\n"); return 2; }
	lsfm_rccl* comm = nullptr;
	if (lsfm_rccl_create(id.data(), 0, 1, 0, (size_t)256 << 20, &comm)) { fprintf(stderr, "lsfm_rccl_create failed\n"); return 2; }
	CHECK(lsfm_rccl_attach(comm, top));
	double worst = 0;
	for (int run = 0; run < 3; run++) // an analysing run (the pattern union goes through the integer all-reduce), then planned ones
	{
		CHECK(lsfm_tree_run(ctx, top, &st));
		int m = 0, n = 0;
		CHECK(lsfm_tree_download_state(ctx, top, &m, &n, nullptr, nullptr, 0));
		if (m != ref.m || n != ref.n) { fprintf(stderr, "sizes differ: %d/%d poses, %d/%d features\n", m, ref.m, n, ref.n); return 3; }
		std::vector<int> stno(6 * (size_t)m + 3 * (size_t)n);
		std::vector<double> val(stno.size());
		CHECK(lsfm_tree_download_state(ctx, top, &m, &n, stno.data(), val.data(), stno.size()));
		for (size_t i = 0; i < stno.size(); i++)
		{
			if (stno[i] != ref.stno[i]) { fprintf(stderr, "labels differ at %zu\n", i); return 3; }
			worst = std::fmax(worst, std::fabs(val[i] - ref.stVal[i]) / std::fmax(1.0, std::fabs(ref.stVal[i])));
		}
	}
	long calls = 0; double elems = 0;
	lsfm_rccl_counters(comm, &calls, &elems);
	printf("lsfm_rccl_selftest: %d local maps (%s), blocks %d + %d, top tree under RCCL: %ld all-reduces, %.0f elements, max rel. difference to the single tree %.3e, "
	       "max rel. residual %.2e\n", N, mono ? "Monocular" : "Stereo", cnt[0], cnt[1], calls, elems, worst, st.max_rel_residual);
	lsfm_tree_free(ctx, top);
	lsfm_rccl_destroy(comm);
	for (void* p : pack) (void)hipFree(p);
	lsfm_map_release(&ref);
	for (auto& g : maps) { g.pose_origin = nullptr; lsfm_map_release(&g); }
	lsfm_context_destroy(ctx);
	return worst < (mono ? 1e-8 : 1e-9) && calls > 0 ? 0 : 3;
}
