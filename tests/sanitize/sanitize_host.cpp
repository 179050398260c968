// Sanitizer driver (CPU only, test infrastructure): the host-side code of the library that runs without a device --
// lsfm_io.cpp (reader, writers) and lsfm_symbolic.cpp (ordering + symbolic factorisation) -- and the oracle (oracle/lsfm_oracle.c,
// lsfm_chol.c), all compiled with -fsanitize=address,undefined by tests/test_sanitize_cpu.py and run over a small join tree:
//   sanitize_host <dir> <N> <Monocular|Stereo>
// reads <dir>/localmap_1..N.txt with the library's reader (one by one and as a threaded set) and with the oracle's, compares
// them, joins the tree with the oracle (serial and threaded), writes / re-reads the result with lsfm_write_localmap and the
// state / pose writers, and runs lsfm_symbolic_analyse on the final map's camera-system pattern (with and without origins).
// Any sanitizer report aborts the process (-fno-sanitize-recover): exit code != 0.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <set>
#include <string>
#include <vector>

#include "../../include/lsfm.h"
extern "C" {
#include "../../oracle/lsfm_oracle.h"
}

extern "C" void lsfm_map_release(lsfm_map* g) // (the library's lives in a .hip file; same body)
{
	if (!g) return;
	free(g->stno); free(g->stVal); free(g->U); free(g->Ui); free(g->Uj); free(g->W); free(g->photo); free(g->feature);
	free(g->V); free(g->FBlock); free(g->pose_origin);
	memset(g, 0, sizeof *g);
}

#define CHECK(c) do { if (!(c)) { fprintf(stderr, "sanitize_host: check failed: %s (line %d)\n", #c, __LINE__); return 1; } } while (0)

static bool same(const lsfm_map& a, const orc_map& b)
{
	if (a.m != b.m || a.n != b.n || a.nU != b.nU || a.nW != b.nW || a.Ref != b.Ref) return false;
	const int r = 6 * a.m + 3 * a.n;
	return !memcmp(a.stno, b.stno, sizeof(int) * r) && !memcmp(a.stVal, b.stVal, sizeof(double) * r) &&
	       !memcmp(a.U, b.U, sizeof(double) * 36 * a.nU) && !memcmp(a.W, b.W, sizeof(double) * 18 * a.nW) &&
	       !memcmp(a.V, b.V, sizeof(double) * 9 * a.n) && !memcmp(a.photo, b.photo, sizeof(int) * a.nW);
}

int main(int argc, char** argv)
{
	if (argc < 4) return 2;
	const std::string dir = argv[1];
	const int N = atoi(argv[2]);
	const int mono = !strcmp(argv[3], "Monocular");
	std::vector<lsfm_map> one(N), set(N);
	std::vector<orc_map> om(N);
	for (int k = 0; k < N; k++)
	{
		const std::string p = dir + "/localmap_" + std::to_string(k + 1) + ".txt";
		CHECK(lsfm_read_localmap(p.c_str(), mono, &one[k]) == 0);
		CHECK(orc_read_map(p.c_str(), mono, &om[k]) == 0);
		CHECK(same(one[k], om[k]));
	}
	int failed = -1;
	CHECK(lsfm_read_localmaps(dir.c_str(), 1, N, mono, 3, set.data(), &failed) == 0);
	for (int k = 0; k < N; k++) CHECK(same(set[k], om[k]));
	{
		std::vector<lsfm_map> more(N + 1);
		CHECK(lsfm_read_localmaps(dir.c_str(), 1, N + 1, mono, 2, more.data(), &failed) != 0 && failed == N + 1); // a missing file: nothing kept
		for (int k = 0; k <= N; k++) CHECK(more[k].stVal == nullptr);
	}
	// the oracle's tree, serial and on threads
	orc_map out, out2;
	double timing[4];
	{
		// (the tree consumes its input maps: each call gets copies)
		std::vector<orc_map> c1(N), c2(N);
		for (int k = 0; k < N; k++) { orc_map_copy(&c1[k], &om[k]); orc_map_copy(&c2[k], &om[k]); }
		CHECK(orc_divide_conquer(c1.data(), N, mono, &out, 0, timing) == 0);
		CHECK(orc_divide_conquer_omp(c2.data(), N, mono, &out2, 3, timing) == 0);
	}
	const int r = 6 * out.m + 3 * out.n;
	CHECK(out.m == out2.m && out.n == out2.n && !memcmp(out.stVal, out2.stVal, sizeof(double) * r));
	// writers: the final map in the local-map format and back, the state and pose files
	lsfm_map fin;
	memset(&fin, 0, sizeof fin);
	fin.Ref = out.Ref; fin.FRef = out.FRef; fin.m = out.m; fin.n = out.n; fin.nU = out.nU; fin.nW = out.nW;
	fin.ScaP = out.ScaP; fin.Fix = out.Fix; fin.Sign = out.Sign; fin.FScaP = out.FScaP; fin.FFix = out.FFix;
	fin.stno = out.stno; fin.stVal = out.stVal; fin.U = out.U; fin.Ui = out.Ui; fin.Uj = out.Uj; fin.W = out.W; fin.photo = out.photo;
	fin.feature = out.feature; fin.V = out.V; fin.FBlock = out.FBlock;
	const std::string fo = dir + "/final_map.txt";
	CHECK(lsfm_write_localmap(fo.c_str(), mono, &fin) == 0);
	lsfm_map back;
	CHECK(lsfm_read_localmap(fo.c_str(), mono, &back) == 0);
	CHECK(back.m == fin.m && back.n == fin.n && back.nW == fin.nW && !memcmp(back.W, fin.W, sizeof(double) * 18 * fin.nW) &&
	      !memcmp(back.stVal, fin.stVal, sizeof(double) * r));
	CHECK(lsfm_save_state((dir + "/state.txt").c_str(), fin.stVal, fin.stno, r) == 0);
	CHECK(lsfm_save_poses((dir + "/pose.txt").c_str(), (dir + "/feat.txt").c_str(), fin.stno, fin.stVal, r) == 0);
	CHECK(lsfm_save_state_bin((dir + "/state.bin").c_str(), fin.stVal, fin.stno, r) == 0);
	// binary cache of the set: write, read back (whole and a sub-range, threaded), a node with pose origins, and damaged files
	{
		const std::string cf = dir + "/set.lsfmbin";
		int cn = -1, cm = -1;
		CHECK(lsfm_mapset_info(cf.c_str(), &cn, &cm) != 0);
		CHECK(lsfm_write_mapset(cf.c_str(), set.data(), N, mono) == 0);
		CHECK(lsfm_mapset_info(cf.c_str(), &cn, &cm) == 0 && cn == N && cm == (mono ? 1 : 0));
		std::vector<lsfm_map> got(N);
		CHECK(lsfm_read_mapset(cf.c_str(), mono, 0, N, 3, got.data()) == 0);
		for (int k = 0; k < N; k++)
		{
			CHECK(got[k].m == set[k].m && got[k].nW == set[k].nW && !memcmp(got[k].W, set[k].W, sizeof(double) * 18 * set[k].nW) &&
			      !memcmp(got[k].stno, set[k].stno, sizeof(int) * (6 * set[k].m + 3 * set[k].n)) && got[k].pose_origin == nullptr);
			lsfm_map_release(&got[k]);
		}
		CHECK(lsfm_read_mapset(cf.c_str(), mono, 1, N - 1, 1, got.data()) == 0 && (N == 1 || got[0].m == set[1].m));
		for (int k = 0; k < N - 1; k++) lsfm_map_release(&got[k]);
		CHECK(lsfm_read_mapset(cf.c_str(), mono, 1, N, 1, got.data()) != 0);  // past the end
		CHECK(lsfm_read_mapset(cf.c_str(), !mono, 0, N, 1, got.data()) != 0); // the other map type
		std::vector<int> org(back.m, 3);
		lsfm_map node = back;
		node.pose_origin = org.data(); node.FRef = 1;
		CHECK(lsfm_write_mapset((dir + "/node.lsfmbin").c_str(), &node, 1, mono) == 0);
		lsfm_map nb;
		CHECK(lsfm_read_mapset((dir + "/node.lsfmbin").c_str(), mono, 0, 1, 0, &nb) == 0 && nb.pose_origin && nb.pose_origin[back.m - 1] == 3 && nb.FRef == 1);
		lsfm_map_release(&nb);
		// truncated at every 97th byte, and with a size field blown up: errors, no wild reads
		std::vector<char> raw;
		{ FILE* f = fopen(cf.c_str(), "rb"); CHECK(f != nullptr); char tmp[4096]; size_t n; while ((n = fread(tmp, 1, sizeof tmp, f)) > 0) raw.insert(raw.end(), tmp, tmp + n); fclose(f); }
		for (size_t cut = 0; cut < raw.size(); cut += std::max<size_t>(97, raw.size() / 64))
		{
			FILE* f = fopen((dir + "/cut.lsfmbin").c_str(), "wb"); CHECK(f != nullptr); fwrite(raw.data(), 1, cut, f); fclose(f);
			CHECK(lsfm_read_mapset((dir + "/cut.lsfmbin").c_str(), mono, 0, N, 2, got.data()) != 0);
		}
		{
			std::vector<char> bad = raw;
			const size_t rec0 = 32 + 8 * (size_t)(N + 1);
			int huge = 0x7fffffff;
			memcpy(bad.data() + rec0 + 4 * 10, &huge, 4); // nW of the first map
			FILE* f = fopen((dir + "/bad.lsfmbin").c_str(), "wb"); CHECK(f != nullptr); fwrite(bad.data(), 1, bad.size(), f); fclose(f);
			CHECK(lsfm_read_mapset((dir + "/bad.lsfmbin").c_str(), mono, 0, N, 1, got.data()) != 0);
		}
	}
	// symbolic analysis of the final map's camera system: pose pairs that share a feature + U's pattern, upper, diagonal first
	{
		const int m = fin.m;
		std::vector<std::set<int>> rows(m);
		for (int p = 0; p < m; p++) rows[p].insert(p);
		for (int k = 0; k < fin.nU; k++) rows[std::min(fin.Ui[k], fin.Uj[k])].insert(std::max(fin.Ui[k], fin.Uj[k]));
		for (int a = 0; a < fin.nW;)
		{
			int b = a;
			while (b < fin.nW && fin.feature[b] == fin.feature[a]) b++;
			for (int x = a; x < b; x++)
				for (int y = a; y < b; y++)
					if (fin.photo[x] <= fin.photo[y]) rows[fin.photo[x]].insert(fin.photo[y]);
			a = b;
		}
		std::vector<int> rowptr(m + 1, 0), colidx, origin(m);
		for (int p = 0; p < m; p++) { for (int q : rows[p]) colidx.push_back(q); rowptr[p + 1] = (int)colidx.size(); origin[p] = p < N ? p : N - 1; }
		std::vector<int> perm(m), colptr(m + 1), rowidx((size_t)m * (m + 1) / 2 + 1);
		int info[8];
		double ms = 0;
		for (int with_origin = 0; with_origin < 2; with_origin++)
		{
			CHECK(lsfm_symbolic_analyse(m, rowptr.data(), colidx.data(), with_origin ? origin.data() : nullptr, 2, perm.data(), colptr.data(), rowidx.data(),
			                            (int)rowidx.size(), info, &ms) == 0);
			CHECK(info[0] == colptr[m] && info[0] >= (int)colidx.size());
			std::vector<char> seen(m, 0);
			for (int i = 0; i < m; i++) { CHECK(perm[i] >= 0 && perm[i] < m && !seen[perm[i]]); seen[perm[i]] = 1; }
			for (int j = 0; j < m; j++) CHECK(rowidx[colptr[j]] == j);
		}
		CHECK(lsfm_symbolic_analyse(m, rowptr.data(), colidx.data(), nullptr, 1, nullptr, nullptr, rowidx.data(), 1, info, nullptr) != 0 || info[0] <= 1); // too small a buffer is refused
	}
	lsfm_map_release(&back);
	for (int k = 0; k < N; k++) { lsfm_map_release(&one[k]); lsfm_map_release(&set[k]); orc_map_free(&om[k]); }
	orc_map_free(&out); orc_map_free(&out2);
	printf("sanitize_host: ok (%d maps, %s, final map %d poses / %d features)\n", N, mono ? "Monocular" : "Stereo", fin.m, fin.n);
	return 0;
}
