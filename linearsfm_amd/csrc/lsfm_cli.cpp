// LinearSFM command line on top of liblsfm_hip: same flags, files and progress lines as the reference's console
// program (main: linux/src/LinearSFM/LinearSFM.cpp:9-18, parser: LinearSFMImp.cpp:7989-8087, help: 8089-8105).
//   LinearSFM -path <dir> -num <N> -type Monocular|Stereo [-p <poses>] [-f <features>] [-st <state>] [-help]
// Extra flags that do not collide with the reference's: -gpu <ordinal>, -tol <pcg rel tol>, -full <file> (final state
// at %.17g), -info <file> (final map WITH its information matrix in the local-map format), -stats 1 (timing breakdown
// on stderr: device stages of the join tree, then the wall seconds of every phase from files to files), -levels <L> -nodes <dir>
// (level checkpoint: stop after L tree levels and write the nodes of that level as <dir>/localmap_1.txt ...; a later run with
// -path <dir> -num <nodes> finishes the tree and gives the result of the uninterrupted run), -quiet 1 (no progress lines).
// -cache <file> (binary cache of the set: read instead of the text files when it holds -num maps of -type, written after the text files
// were parsed otherwise), -fullbin <file> (final state as raw doubles), -json <file> (the run's lsfm_stats and phase times as one JSON object),
// -gn <steps> (Gauss-Newton polish of the map-joining objective from the tree's result: lsfm_gn_polish; no reference counterpart).
#include <chrono>
#include <sys/stat.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/lsfm.h"

static void print_help()
{
	printf("Linear SFM Solution General Options\n");
	printf("\n");
	printf("-path			Set Data Path.\n");
	printf("-st            Set Path to Save Final State Vector\n");
	printf("-p			Set Path to Save Poses\n");
	printf("-f			Set Path to Save Features\n");
	printf("-num			Number of Initial Recontruction\n");
	printf("-type			Set Data Type.\n");
	printf("Data Type Listed As Following:\n");
	printf("			I  : Monocular\n");
	printf("			II : Stereo\n");
	printf("\n");
}

int main(int argc, char** argv)
{
	std::string path, st, pose, fea, full, info, nodes, cache, fullbin, json;
	int num = 0, type = -1, gpu = 0, want_stats = 0, levels = 0, quiet = 0, gn = 0;
	bool has_path = false, has_num = false;
	double tol = 0;
	for (int i = 1; i < argc; i++)
	{
		std::string name = argv[i];
		if (name[0] != '-') return 0; // each param has to start with at least one dash (Imp.cpp:8000-8002)
		size_t d = name.find_first_not_of('-');
		if (d != std::string::npos) name = name.substr(d);
		if (name == "help") { print_help(); return 0; }
		auto next = [&]() -> const char* { return (i + 1 < argc) ? argv[++i] : ""; };
		if (name == "path") { path = next(); has_path = true; }
		else if (name == "st") st = next();
		else if (name == "p") pose = next();
		else if (name == "f") fea = next();
		else if (name == "num") { num = atoi(next()); has_num = true; }
		else if (name == "type")
		{
			std::string v = next();
			if (v == "Monocular") type = 1;
			if (v == "Stereo") type = 0;
		}
		else if (name == "gpu") gpu = atoi(next());
		else if (name == "tol") tol = atof(next());
		else if (name == "full") full = next();
		else if (name == "info") info = next();
		else if (name == "stats") want_stats = atoi(next());
		else if (name == "levels") levels = atoi(next());
		else if (name == "nodes") nodes = next();
		else if (name == "quiet") quiet = atoi(next());
		else if (name == "cache") cache = next();
		else if (name == "fullbin") fullbin = next();
		else if (name == "json") json = next();
		else if (name == "gn") gn = atoi(next());
	}
	if (!has_path) { printf("LinerSFM Error: Please Input Right File Path:\n"); return 0; }
	if (!has_num) { printf("LinerSFM Error: Please Set Local Map Number:\n"); return 0; }
	if (type < 0) { printf("LinerSFM Error: Please Set Data Type:\n"); return 0; }
	if (num <= 0) { fprintf(stderr, "LinearSFM: -num must be positive (got %d)\n", num); return 1; }
	if ((levels > 0) != !nodes.empty() || levels < 0) { fprintf(stderr, "LinearSFM: -levels <L > 0> and -nodes <dir> go together\n"); return 1; }

	auto now = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
	const double w0 = now();
	std::vector<lsfm_map> maps(num);
	bool from_cache = false;
	// what a cache must have been made from to be believed: the resolved directory and size + modification time of its first n text files
	// (FNV-1a; 0 is "no stamp": never produced)
	auto source_stamp = [&](int n, bool* all_there) {
		unsigned long long h = 1469598103934665603ull;
		auto mix = [&](const void* p, size_t len) { const unsigned char* b = static_cast<const unsigned char*>(p); for (size_t i = 0; i < len; i++) { h ^= b[i]; h *= 1099511628211ull; } };
		char* rp = realpath(path.c_str(), nullptr);
		const std::string dir = rp ? rp : path;
		free(rp);
		mix(dir.data(), dir.size());
		*all_there = true;
		for (int k = 1; k <= n; k++)
		{
			struct stat st;
			const std::string fn = path + "/localmap_" + std::to_string(k) + ".txt";
			if (stat(fn.c_str(), &st) != 0) { *all_there = false; return 0ull; }
			const long long v[3] = { (long long)st.st_size, (long long)st.st_mtim.tv_sec, (long long)st.st_mtim.tv_nsec };
			mix(v, sizeof v);
		}
		return h ? h : 1ull;
	};
	if (!cache.empty())
	{
		// the binary cache of an earlier run over this set: used when it holds what is asked for AND was made from the text files as they
		// are now (its stamp); (re)written otherwise.  Text files that are gone cannot contradict it: the cache is then all there is.
		int cn = 0, cm = 0;
		if (lsfm_mapset_info(cache.c_str(), &cn, &cm) == LSFM_OK && cn >= num && cm == type)
		{
			unsigned long long have = 0;
			bool there = false;
			const unsigned long long want = source_stamp(cn, &there);
			(void)lsfm_mapset_stamp(cache.c_str(), &have, nullptr);
			if (there && have != want)
				fprintf(stderr, "LinearSFM: %s was not made from the text files under %s as they are now: reading them again\n", cache.c_str(), path.c_str());
			else from_cache = lsfm_read_mapset(cache.c_str(), type, 0, num, 0, maps.data()) == LSFM_OK;
		}
		else if (cn) fprintf(stderr, "LinearSFM: %s does not hold %d %s maps: reading the text files\n", cache.c_str(), num, type ? "Monocular" : "Stereo");
	}
	if (!from_cache)
	{
		// localmap_1.txt ... localmap_<num>.txt (Imp.cpp:125), parsed on all host cores
		int bad = 0;
		if (lsfm_read_localmaps(path.c_str(), 1, num, type, 0, maps.data(), &bad))
		{
			fprintf(stderr, "LinearSFM: cannot read %s/localmap_%d.txt\n", path.c_str(), bad);
			return 1;
		}
		if (!cache.empty())
		{
			bool there = false;
			const unsigned long long stamp = source_stamp(num, &there);
			if (lsfm_write_mapset(cache.c_str(), maps.data(), num, type) || lsfm_mapset_stamp(cache.c_str(), nullptr, &stamp))
				fprintf(stderr, "LinearSFM: cannot write %s\n", cache.c_str());
		}
	}
	const double w1 = now();
	lsfm_context* ctx = nullptr;
	int rc = lsfm_context_create(gpu, 0, &ctx);
	if (rc) { fprintf(stderr, "LinearSFM: no HIP device (rc=%d); this build has no CPU path\n", rc); return 2; }
	if (tol > 0) lsfm_set_pcg(ctx, tol, 0);
	// progress lines of the reference (Imp.cpp:1952, 1995)
	if (!quiet)
	{
		int cnt = num, L = 0;
		while (cnt > 1 && (levels <= 0 || L < levels))
		{
			int N2 = cnt % 2;
			cnt = (int)(cnt / 2.0 + 0.5);
			for (int i = 0; i < cnt; i++)
			{
				int NumLM = (i < cnt - 1 || N2 == 0) ? 2 : 1;
				for (int j = 0; j < NumLM; j++) printf("Join Level %d Local Map %d\n", L, 2 * i + j + 1);
				printf("Generate Level %d Local Map %d\n\n", L + 1, i + 1);
			}
			L++;
		}
	}
	// what lsfm_divide_conquer does, in its three steps so that each can be timed: maps to the device, the join tree (the
	// region the reference times), the final map back
	lsfm_map out;
	lsfm_stats stats;
	lsfm_tree* tree = nullptr;
	const double w2 = now();
	rc = lsfm_tree_upload(ctx, maps.data(), num, type, &tree);
	if (rc < 0) { fprintf(stderr, "LinearSFM: %s\n", lsfm_last_error(ctx)); return 3; }
	lsfm_tree_set_plans(tree, 0); // one run: nothing to keep for a next one
	if (levels > 0) lsfm_tree_set_stop_level(tree, levels);
	const double w3 = now();
	rc = lsfm_tree_run(ctx, tree, &stats);
	if (rc < 0) { fprintf(stderr, "LinearSFM: %s\n", lsfm_last_error(ctx)); return 3; }
	const double w4 = now();
	const int nnodes = lsfm_tree_node_count(ctx, tree);
	if (levels > 0 && nnodes > 1)
	{
		// level checkpoint: the nodes of the level the run ended at, named like a set of local maps
		for (int k = 0; k < nnodes; k++)
		{
			lsfm_map node;
			if (lsfm_tree_download_node(ctx, tree, k, &node) < 0) { fprintf(stderr, "LinearSFM: %s\n", lsfm_last_error(ctx)); return 3; }
			const std::string fn = nodes + "/localmap_" + std::to_string(k + 1) + ".txt";
			const int wrc = lsfm_write_localmap(fn.c_str(), type, &node);
			lsfm_map_release(&node);
			if (wrc) { fprintf(stderr, "LinearSFM: cannot write %s\n", fn.c_str()); return 1; }
		}
		printf("Stopped After Level %d: %d Nodes Written To %s\n", levels, nnodes, nodes.c_str());
		printf("Total Used Time:  %lf  sec\n\n", stats.t_total_ms * 1e-3);
		lsfm_tree_free(ctx, tree);
		for (auto& g : maps) lsfm_map_release(&g);
		lsfm_context_destroy(ctx);
		return rc == LSFM_NOT_CONVERGED ? 4 : 0;
	}
	{
		const int drc = lsfm_tree_download(ctx, tree, &out);
		if (drc < 0) { fprintf(stderr, "LinearSFM: %s\n", lsfm_last_error(ctx)); return 3; }
	}
	lsfm_tree_free(ctx, tree);
	const double w5 = now();
	// the reference solves directly and cannot end half-way; a system the refinement left above its residual bound is
	// reported and reflected in the exit code (the files are still written)
	const bool partial = rc == LSFM_NOT_CONVERGED;
	if (partial)
		fprintf(stderr, "LinearSFM: WARNING: %d camera system(s) not solved to the residual of a direct solve (max relative residual %.3e)\n",
		        stats.not_converged, stats.max_rel_residual);
	printf("Total Used Time:  %lf  sec\n\n", stats.t_total_ms * 1e-3); // Imp.cpp:2072
	if (gn > 0)
	{
		// -gn <steps>: Gauss-Newton polish of the map-joining objective over all local maps, from the tree's result (lsfm_gn_polish; the
		// reference has no such step -- without the flag the program is the reference's)
		std::vector<double> obj(gn + 1), gnorm(gn + 1);
		std::vector<int> halv(gn);
		const double g0 = now();
		const int grc = lsfm_gn_polish(ctx, maps.data(), num, type, &out, gn, obj.data(), gnorm.data(), halv.data());
		if (grc < 0) { fprintf(stderr, "LinearSFM: %s\n", lsfm_last_error(ctx)); return 3; }
		if (!quiet)
		{
			for (int i = 0; i <= gn; i++) printf("Gauss-Newton Step %d: Objective %.9e  Gradient %.3e\n", i, obj[i], gnorm[i]);
			printf("Gauss-Newton Used Time:  %lf  sec\n\n", now() - g0);
		}
	}
	if (want_stats)
		fprintf(stderr, "lsfm: total %.3f ms (transform %.3f, join %.3f [schur %.3f, pcg %.3f, backsub %.3f]), pcg its %ld, max rel resid %.2e, not converged %d, attempts %d\n",
		        stats.t_total_ms, stats.t_transform_ms, stats.t_join_ms, stats.t_schur_ms, stats.t_pcg_ms, stats.t_backsub_ms, stats.pcg_iterations,
		        stats.max_rel_residual, stats.not_converged, stats.attempts);
	const int r = 6 * out.m + 3 * out.n;
	if (!st.empty()) lsfm_save_state(st.c_str(), out.stVal, out.stno, r);
	if (!pose.empty() && !fea.empty()) lsfm_save_poses(pose.c_str(), fea.c_str(), out.stno, out.stVal, r); // only together (Imp.cpp:2078)
	if (!full.empty())
	{
		FILE* f = fopen(full.c_str(), "w");
		if (f) { for (int i = 0; i < r; i++) fprintf(f, "%d %.17g\n", out.stno[i], out.stVal[i]); fclose(f); }
	}
	if (!info.empty() && lsfm_write_localmap(info.c_str(), type, &out)) fprintf(stderr, "LinearSFM: cannot write %s\n", info.c_str());
	if (!fullbin.empty() && lsfm_save_state_bin(fullbin.c_str(), out.stVal, out.stno, r)) fprintf(stderr, "LinearSFM: cannot write %s\n", fullbin.c_str());
	if (!json.empty())
	{
		FILE* f = fopen(json.c_str(), "w");
		if (f)
		{
			fprintf(f, "{\"maps\": %d, \"type\": \"%s\", \"from_cache\": %s, \"poses\": %d, \"features\": %d, \"rc\": %d, \"t_total_ms\": %.6f, \"t_transform_ms\": %.6f, "
			           "\"t_join_ms\": %.6f, \"t_schur_ms\": %.6f, \"t_pcg_ms\": %.6f, \"t_backsub_ms\": %.6f, \"levels\": %d, \"joins\": %d, \"transforms\": %d, "
			           "\"pcg_iterations\": %ld, \"max_rel_residual\": %.6e, \"not_converged\": %d, \"attempts\": %d, "
			           "\"phases_s\": {\"read\": %.6f, \"context\": %.6f, \"upload\": %.6f, \"join_tree\": %.6f, \"download\": %.6f, \"write\": %.6f}}\n",
			        num, type ? "Monocular" : "Stereo", from_cache ? "true" : "false", out.m, out.n, rc, stats.t_total_ms, stats.t_transform_ms, stats.t_join_ms,
			        stats.t_schur_ms, stats.t_pcg_ms, stats.t_backsub_ms, stats.levels, stats.joins, stats.transforms, (long)stats.pcg_iterations,
			        stats.max_rel_residual, stats.not_converged, stats.attempts, w1 - w0, w2 - w1, w3 - w2, w4 - w3, w5 - w4, now() - w5);
			fclose(f);
		}
		else fprintf(stderr, "LinearSFM: cannot write %s\n", json.c_str());
	}
	if (want_stats)
		fprintf(stderr, "lsfm_e2e: read %.3f s, context %.3f s, upload %.3f s, join tree %.3f s, download %.3f s, write %.3f s\n", w1 - w0, w2 - w1, w3 - w2,
		        w4 - w3, w5 - w4, now() - w5);
	lsfm_map_release(&out);
	for (auto& g : maps) lsfm_map_release(&g);
	lsfm_context_destroy(ctx);
	return partial ? 4 : 0;
}
