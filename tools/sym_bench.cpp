// The symbolic analysis (lsfm_symbolic.cpp) on dumped inputs, away from the device:
//   LSFM_SYM_DUMP=<dir> python tools/timeline_run.py synth16k      (on the GPU box: sym_<M>_<nnzb>.bin per large system)
//   g++ -O3 -std=c++17 -pthread -Ilinearsfm_amd/csrc tools/sym_bench.cpp linearsfm_amd/csrc/lsfm_symbolic.cpp -o /tmp/sym_bench
//   LSFM_SYM_TIMING=1 /tmp/sym_bench sym_16386_822558.bin [reps]
// prints the average time of an analysis and a digest of everything it produces (an optimisation must leave the digest alone).
#include "lsfm_symbolic.hpp"

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

static unsigned long long mix(unsigned long long h, const std::vector<int>& v)
{
	for (int x : v) { h ^= (unsigned)x + 0x9E3779B97F4A7C15ull + (h << 6) + (h >> 2); }
	return h ^ v.size();
}

int main(int argc, char** argv)
{
	if (argc < 2) { fprintf(stderr, "usage: sym_bench <dump> [reps]\n"); return 2; }
	const int reps = argc > 2 ? atoi(argv[2]) : 5;
	FILE* f = fopen(argv[1], "rb");
	if (!f) { perror(argv[1]); return 1; }
	int hdr[4];
	if (fread(hdr, sizeof(int), 4, f) != 4) return 1;
	const int M = hdr[0], nnzb = hdr[1], block_maps = hdr[2];
	std::vector<unsigned long long> keys((size_t)nnzb);
	std::vector<int> origin((size_t)M);
	if (fread(keys.data(), 8, (size_t)nnzb, f) != (size_t)nnzb || fread(origin.data(), 4, (size_t)M, f) != (size_t)M) return 1;
	fclose(f);
	lsfm::CholSymbolic sym;
	lsfm::chol_symbolic(keys.data(), nnzb, origin.data(), M, sym, block_maps);
	const auto t0 = std::chrono::steady_clock::now();
	for (int r = 0; r < reps; r++) lsfm::chol_symbolic(keys.data(), nnzb, origin.data(), M, sym, block_maps);
	const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / reps;
	unsigned long long h = 0;
	for (const std::vector<int>* v : { &sym.perm, &sym.pinv, &sym.parent, &sym.ccount, &sym.colptr, &sym.rowidx, &sym.order, &sym.task_ptr, &sym.task_cols, &sym.col_task,
	                                   &sym.col_lpos, &sym.col_nin, &sym.grp_c0, &sym.grp_s, &sym.grp_nr, &sym.glevel_ptr, &sym.tlevel_ptr, &sym.level_ptr })
		h = mix(h, *v);
	printf("M %d nnzb %d: %.3f ms per analysis, nnzL %d, groups %d in %d levels, digest %016llx\n", M, nnzb, ms, sym.nnzL, sym.ngroups, (int)sym.glevel_ptr.size() - 1, h);
	return 0;
}
