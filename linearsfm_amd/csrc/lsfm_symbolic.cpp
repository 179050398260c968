// Ordering + symbolic factorisation of the camera system of a tree level, on the host (lsfm_symbolic.hpp).
//
//   * ordering: nested dissection along the join tree.  The local map that brought a pose ("origin") encodes where the
//     pose sits in the tree, so an edge (p,q) of S crosses the cut of tree level bitlen(origin_p ^ origin_q); the crossing
//     edges of a level lose one endpoint each to that level's separator (greedy vertex cover, top level first), blocks
//     are eliminated by ascending separator level.
//   * elimination tree, column counts and column patterns by row sub-tree walks.
//   * tasks (leaf sub-trees that fit LDS, chains above them) and supernode groups (runs of <= CHOL_GS columns of one
//     fundamental supernode) for the device kernels of lsfm_pcg.hip.
//
// This runs once per tree level inside the timed region of a run that analyses (the reference's cholmod_analyze_p runs
// once per join), so it is written for speed: one workspace kept between calls, counting sorts instead of comparison
// sorts, dense position maps instead of binary searches -- 3.6 ms -> 1 ms for the top join of the NC3500-like set
// (3499 poses, 65 k blocks of S, 130 k blocks of L), 0.4 -> 0.1 ms for a bottom level.  Results are identical to the
// round-2 code (checked block for block on the systems of the NC3500-like and RS468-like trees).
#include "lsfm_symbolic.hpp"

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <queue>
#include <thread>
#include <utility>

namespace lsfm {
// most block columns of a supernode group: CHOL_GS, or less (LSFM_GS, for measurements)
static const int gs_cap = []() { const char* e = getenv("LSFM_GS"); const int v = e ? atoi(e) : CHOL_GS; return v >= 1 && v <= CHOL_GS ? v : CHOL_GS; }();


namespace {

inline int bitlen(unsigned x) { return x ? 32 - __builtin_clz(x) : 0; }

struct Workspace {
	std::vector<int> sep, lev_cnt, edge_lev, edge_a, edge_b, eorder;
	std::vector<int> vid, verts, cdeg, start, adj, cur;
	std::vector<char> covered;
	std::vector<std::pair<int, int>> live;
	std::vector<std::vector<int>> bucket;
	std::vector<unsigned long long> heap;
	std::vector<int> rcnt, radj, fill, anc, mark, cfill, lev, lcount, lfill;
	std::vector<int> size, ntc, topchild, task, tlev;
};

Workspace& workspace()
{
	static thread_local Workspace w;
	return w;
}

} // namespace

// A few host threads that live as long as the process, for the chunks of independent systems of one analysis (threads started per
// call fault their fresh workspaces in under the process's memory lock, one after the other: no gain).  One analysis at a time uses
// them; a second caller (another context's helper thread) runs its chunks itself.
struct SymPool {
	std::vector<std::thread> th;
	std::mutex m, user;
	std::condition_variable cv, done_cv;
	const std::function<void(int)>* job = nullptr;
	int next = 0, total = 0, running = 0, epoch = 0;
	bool quit = false;
	~SymPool()
	{
		{ std::lock_guard<std::mutex> l(m); quit = true; }
		cv.notify_all();
		for (auto& t : th) if (t.joinable()) t.join();
	}
	void worker()
	{
		std::unique_lock<std::mutex> l(m);
		int seen = 0;
		for (;;)
		{
			cv.wait(l, [&] { return quit || (job && epoch != seen && next < total); });
			if (quit) return;
			while (job && next < total)
			{
				const int c = next++;
				running++;
				l.unlock();
				(*job)(c);
				l.lock();
				running--;
			}
			seen = epoch;
			if (running == 0) done_cv.notify_all();
		}
	}
	void run(int n, const std::function<void(int)>& fn)
	{
		std::unique_lock<std::mutex> u(user, std::try_to_lock);
		if (!u.owns_lock()) { for (int c = 0; c < n; c++) fn(c); return; }
		std::unique_lock<std::mutex> l(m);
		const int want = std::min(n - 1, 7);
		while ((int)th.size() < want) th.emplace_back([this] { worker(); });
		job = &fn; next = 0; total = n; epoch++;
		cv.notify_all();
		while (next < total) // the caller takes chunks too
		{
			const int c = next++;
			running++;
			l.unlock();
			fn(c);
			l.lock();
			running--;
		}
		done_cv.wait(l, [&] { return running == 0; });
		job = nullptr;
	}
};
static SymPool& sym_pool()
{
	static SymPool p;
	return p;
}

// ---- separators: greedy vertex cover of the crossing edges, top level first ---------------------------------------------------
// The crossing edges among keys[kb, ke) -- the rows of a set of whole systems -- bucketed by the tree level they cross.  An edge
// of level l joins two poses whose origins agree above bit l - 1: the edges of the levels <= split fall into independent GROUPS
// (origin >> split) -- no edge of those levels leaves its group, and what the levels above decided about a group's poses is final
// before the group starts -- so the groups' covers are worked out side by side (chol_symbolic: one task per group), each exactly as
// the single pass over all edges of a level would have: the greedy choice inside a group never looks at another group's degrees.
// (The top system of a 16 384-map monocular tree: 806 k crossing edges, 81 % of them below the top five cuts; its cover took 14 of
// the analysis's 30 ms on one thread while the device waited for the factorisation's index arrays.)
struct CoverJob {
	int kb = 0, ke = 0, ne = 0, split = 0, ngroups = 1, gbase = 0;
	std::vector<int> edge_a, edge_b; // the crossing edges, bucketed: the levels above split by level, then (group, level <= split)
	std::vector<int> high_ptr;  // [36]: the ranges of the levels above split
	std::vector<int> group_ptr; // [ngroups * (split + 1) + 1]: the ranges of (group, level <= split), behind the high ones
};

// one cut level: the uncovered ones of the (bucketed) edges [b0, b1) lose one endpoint each to the level's separator
static void cover_level(int l, const CoverJob& jb, int b0, int b1, std::vector<int>& sep, Workspace& w, int M)
{
	if (b0 == b1) return;
	if ((int)w.vid.size() != M) w.vid.assign(M, -1); // (left all -1 by every call)
	std::vector<std::pair<int, int>>& live = w.live;
	live.clear();
	for (int t = b0; t < b1; t++)
	{
		const int p = jb.edge_a[t], q = jb.edge_b[t];
		if (sep[p] < l && sep[q] < l) live.emplace_back(p, q);
	}
	if (live.empty()) return;
	// their endpoints, ascending (the tie-break of the heap below is the pose index), with a dense local numbering
	std::vector<int>& verts = w.verts;
	verts.clear();
	for (const auto& pq : live)
	{
		if (w.vid[pq.first] < 0) { w.vid[pq.first] = 0; verts.push_back(pq.first); }
		if (w.vid[pq.second] < 0) { w.vid[pq.second] = 0; verts.push_back(pq.second); }
	}
	std::sort(verts.begin(), verts.end());
	const int nv = (int)verts.size();
	for (int i = 0; i < nv; i++) w.vid[verts[i]] = i;
	w.cdeg.assign(nv, 0);
	for (const auto& pq : live) { w.cdeg[w.vid[pq.first]]++; w.cdeg[w.vid[pq.second]]++; }
	w.start.resize(nv + 1);
	w.start[0] = 0;
	for (int i = 0; i < nv; i++) w.start[i + 1] = w.start[i] + w.cdeg[i];
	w.adj.resize(w.start[nv]);
	w.cur.assign(nv, 0); // fill counters first, live degrees afterwards
	for (int e = 0; e < (int)live.size(); e++)
	{
		const int a = w.vid[live[e].first], b = w.vid[live[e].second];
		w.adj[w.start[a] + w.cur[a]++] = e;
		w.adj[w.start[b] + w.cur[b]++] = e;
	}
	w.covered.assign(live.size(), 0);
	// the pose with the most uncovered crossing edges first, the higher pose index on a tie: ONE max-heap of (degree, local
	// vertex number) keys (verts is ascending, so local order = pose order) whose entries are refreshed lazily -- covering an
	// edge only lowers the other endpoint's live degree; a popped entry whose degree is stale goes back with the live one.
	// (Round 3 pushed an entry into a per-degree heap at every decrement: ~1 M heap pushes for the top system of a
	// 16 384-map monocular tree, 57 of the 78 ms of its analysis.)  The vertex selected is the same one.
	for (int i = 0; i < nv; i++) w.cur[i] = w.cdeg[i];
	std::vector<unsigned long long>& hp = w.heap;
	hp.resize(nv);
	for (int i = 0; i < nv; i++) hp[i] = ((unsigned long long)(unsigned)w.cdeg[i] << 32) | (unsigned)i;
	std::make_heap(hp.begin(), hp.end());
	while (!hp.empty())
	{
		std::pop_heap(hp.begin(), hp.end());
		const unsigned long long key = hp.back();
		hp.pop_back();
		const int iv = (int)(key & 0xffffffffull), d = (int)(key >> 32);
		const int live_d = w.cur[iv];
		if (live_d <= 0) continue; // (all its edges are covered)
		if (live_d != d)
		{
			hp.push_back(((unsigned long long)(unsigned)live_d << 32) | (unsigned)iv);
			std::push_heap(hp.begin(), hp.end());
			continue;
		}
		const int v = verts[iv];
		sep[v] = l;
		for (int t = w.start[iv]; t < w.start[iv + 1]; t++)
		{
			const int e = w.adj[t];
			if (w.covered[e]) continue;
			w.covered[e] = 1;
			const int u = live[e].first == v ? live[e].second : live[e].first;
			--w.cur[w.vid[u]];
		}
		w.cur[iv] = 0;
	}
	for (int v : verts) w.vid[v] = -1;
}

// the edges of a job, bucketed (counting sort, order inside a bucket kept), and its top levels' separators -- the part that is one
// piece of work; `want_groups`: how many groups to cut the lower levels into (1: none); rows [r0, r1) are the job's
static void cover_prepare(const unsigned long long* keys, const int* origin, int M, int r0, int r1, std::vector<int>& sep, CoverJob& jb, int want_groups)
{
	const int nk = jb.ke - jb.kb;
	static const bool tm = getenv("LSFM_SYM_TIMING") != nullptr;
	auto now = []() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
	const double t0 = tm ? now() : 0;
	jb.edge_a.resize(nk); jb.edge_b.resize(nk);
	Workspace& w = workspace();
	// groups: origin >> split, as many as asked for (a power of two) when the job has the levels for it.  (No key crosses a job's
	// rows: the origins of its rows bound the levels of its edges.)
	unsigned omin = ~0u, omax = 0;
	for (int p = r0; p < r1; p++) { omin = std::min(omin, (unsigned)origin[p]); omax = std::max(omax, (unsigned)origin[p]); }
	int split = 0;
	if (want_groups > 1 && nk >= 100000 && r1 > r0)
	{
		int lg = 0;
		while ((1 << lg) < want_groups) lg++;
		split = bitlen(omin ^ omax) - lg;
		if (split < 2) split = 0;
	}
	jb.split = split;
	jb.gbase = split > 0 ? (int)(omin >> split) : 0;
	jb.ngroups = split > 0 ? (int)(omax >> split) - jb.gbase + 1 : 1;
	// buckets: the levels above split by level (34 .. split + 1: the order they are taken in does not matter for the layout), behind
	// them (group, level <= split)
	const int per = split + 1, nlow = jb.ngroups * per, NB = 35 + nlow;
	jb.high_ptr.assign(36, 0);
	jb.group_ptr.assign(nlow + 1, 0);
	std::vector<int>& bucket = w.edge_lev; // per key: < 0 diagonal, 0 .. 34 a high level, 35 + (group, level) a low one
	bucket.resize(nk);
	// a stable counting sort over ranges of the keys (one per thread when this is the one large job of the level: chol_symbolic):
	// every range counts per bucket, the running sums over (bucket, range) are every range's first places, every range writes in key order
	const int T = (want_groups > 1 && nk >= 100000) ? std::max(1, want_groups / 2) : 1;
	std::vector<int>& cnt = w.cdeg; // [T][NB]
	cnt.assign((size_t)T * NB, 0);
	const int kb = jb.kb, gbase = jb.gbase;
	const std::function<void(int)> classify = [&](int t) {
		const int e0 = (int)((long)nk * t / T), e1 = (int)((long)nk * (t + 1) / T);
		int* c = cnt.data() + (size_t)t * NB;
		for (int e = e0; e < e1; e++)
		{
			const unsigned long long k = keys[kb + e];
			const int p = (int)(k >> 32), q = (int)(k & 0xffffffffull);
			if (p == q) { bucket[e] = -1; continue; }
			const int l = bitlen((unsigned)(origin[p] ^ origin[q]));
			const int b = (l > split || split == 0) ? l : 35 + ((int)((unsigned)origin[p] >> split) - gbase) * per + l;
			bucket[e] = b;
			c[b]++;
		}
	};
	if (T > 1) sym_pool().run(T, classify); else classify(0);
	{
		int run = 0;
		for (int b = 0; b < NB; b++)
		{
			if (b < 35) jb.high_ptr[b] = run; else jb.group_ptr[b - 35] = run;
			for (int t = 0; t < T; t++) { const int x = cnt[(size_t)t * NB + b]; cnt[(size_t)t * NB + b] = run; run += x; }
			if (b == 34) jb.high_ptr[35] = run;
		}
		jb.group_ptr[nlow] = run;
		jb.ne = run;
	}
	const double t1 = tm ? now() : 0;
	const std::function<void(int)> place = [&](int t) {
		const int e0 = (int)((long)nk * t / T), e1 = (int)((long)nk * (t + 1) / T);
		int* pos = cnt.data() + (size_t)t * NB;
		for (int e = e0; e < e1; e++)
		{
			const int b = bucket[e];
			if (b < 0) continue;
			const unsigned long long k = keys[kb + e];
			const int at = pos[b]++;
			jb.edge_a[at] = (int)(k >> 32); jb.edge_b[at] = (int)(k & 0xffffffffull);
		}
	};
	if (T > 1) sym_pool().run(T, place); else place(0);
	const double t2 = tm ? now() : 0;
	for (int l = 33; l >= 1; l--)
		if (l > split || split == 0) cover_level(l, jb, jb.high_ptr[l], jb.high_ptr[l + 1], sep, w, M);
	if (tm && nk > 100000) fprintf(stderr, "[sym]     edges %.3f ms, buckets %.3f ms, top levels %.3f ms\n", t1 - t0, t2 - t1, now() - t2);
}
// the levels <= split of one group
static void cover_group(const CoverJob& jb, int g, std::vector<int>& sep, int M)
{
	Workspace& w = workspace();
	const int per = jb.split + 1;
	for (int l = jb.split; l >= 1; l--) cover_level(l, jb, jb.group_ptr[g * per + l], jb.group_ptr[g * per + l + 1], sep, w, M);
}

void chol_symbolic(const unsigned long long* keys, int nnzb, const int* origin, int M, CholSymbolic& ch, int block_maps)
{
	Workspace& w = workspace();
	static const bool sym_timing = getenv("LSFM_SYM_TIMING") != nullptr;
	auto tnow = []() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
	double tprev = tnow();
	auto tick = [&](const char* what) { if (sym_timing) { const double n = tnow(); fprintf(stderr, "[sym] %-28s %8.3f ms\n", what, n - tprev); tprev = n; } };
	if (const char* dump = getenv("LSFM_SYM_DUMP"))
	{
		// diagnostic: the inputs of the analyses of large systems, for work on this function away from the device (tools/sym_bench.cpp)
		if (nnzb >= 200000)
		{
			char path[512];
			snprintf(path, sizeof path, "%s/sym_%d_%d.bin", dump, M, nnzb);
			if (FILE* f = fopen(path, "wb"))
			{
				const int hdr[4] = { M, nnzb, block_maps, 0 };
				fwrite(hdr, sizeof(int), 4, f);
				fwrite(keys, sizeof(unsigned long long), (size_t)nnzb, f);
				fwrite(origin, sizeof(int), (size_t)M, f);
				fclose(f);
			}
		}
	}
	// ---- separators: greedy vertex cover of the crossing edges, top level first --------------------------------------
	std::vector<int>& sep = w.sep;
	sep.assign(M, 0);
	// Independent systems of the level (no edge between them) are analysed side by side: the rows are cut where no key crosses into
	// chunks of about equal numbers of keys, one host thread each for the vertex cover here and for the row sub-tree walks below
	// (elimination tree, column counts, column patterns: a row only ever touches columns of its own system).  A 16 384-map monocular
	// tree spent 330 ms of an analysing run's 570 in this function, on one thread, while the device waited.
	std::vector<int> cut_row(1, 0), cut_key(1, 0); // chunk c: rows [cut_row[c], cut_row[c+1]), keys [cut_key[c], cut_key[c+1])
	{
		static const int max_threads = getenv("LSFM_SYM_THREADS") ? std::max(1, atoi(getenv("LSFM_SYM_THREADS"))) : 8;
		const int want = (nnzb < 60000 || block_maps > 0) ? 1 : std::min<int>(max_threads, std::max(1u, std::thread::hardware_concurrency() / 2));
		if (want > 1)
		{
			const long target = (nnzb + want - 1) / want;
			int maxq = -1, row = -1;
			for (int e = 0; e <= nnzb; e++)
			{
				const int p = e < nnzb ? (int)(keys[e] >> 32) : M;
				if (p != row)
				{
					// rows row+1 .. p-1 hold no key (cannot happen: every row has its diagonal block); a cut before row p is valid when nothing reaches p
					if (row >= 0 && maxq < p && e - cut_key.back() >= target && p < M) { cut_row.push_back(p); cut_key.push_back(e); }
					row = p;
				}
				if (e < nnzb) maxq = std::max(maxq, (int)(keys[e] & 0xffffffffull));
			}
		}
		cut_row.push_back(M); cut_key.push_back(nnzb);
	}
	const int nchunk = (int)cut_row.size() - 1;
	auto par_chunks = [&](auto&& fn) {
		if (nchunk == 1) { fn(0); return; }
		sym_pool().run(nchunk, std::function<void(int)>([&fn](int c) { fn(c); }));
	};
	{
		// (the jobs keep their arrays between calls)
		static thread_local std::vector<CoverJob> jobs_tl;
		std::vector<CoverJob>& jobs = jobs_tl; // (a reference: the lambdas below run on other threads, whose own jobs_tl is not this one)
		if ((int)jobs.size() < nchunk) jobs.resize(nchunk);
		static const int max_threads = getenv("LSFM_SYM_THREADS") ? std::max(1, atoi(getenv("LSFM_SYM_THREADS"))) : 8;
		const int threads = block_maps > 0 ? 1 : std::min<int>(max_threads, std::max(1u, std::thread::hardware_concurrency() / 2));
		// twice as many pieces as threads between all the chunks: the groups of a job differ in work
		const int want_groups = threads > 1 ? std::max(1, 2 * threads / nchunk) : 1;
		par_chunks([&](int c) {
			const double t0 = sym_timing ? tnow() : 0;
			jobs[c].kb = cut_key[c]; jobs[c].ke = cut_key[c + 1];
			cover_prepare(keys, origin, M, cut_row[c], cut_row[c + 1], sep, jobs[c], want_groups);
			if (sym_timing) fprintf(stderr, "[sym]   chunk %d: rows %d..%d, %d keys, split at level %d into %d groups, top levels' cover %.3f ms\n", c, cut_row[c], cut_row[c + 1], cut_key[c + 1] - cut_key[c], jobs[c].split, jobs[c].ngroups, tnow() - t0);
		});
		tick("separators: top levels");
		std::vector<std::pair<int, int>> tasks;
		for (int c = 0; c < nchunk; c++)
			if (jobs[c].split > 0)
				for (int g = 0; g < jobs[c].ngroups; g++) tasks.emplace_back(c, g);
		if (!tasks.empty())
		{
			const std::function<void(int)> fn = [&](int t) {
				const double t0 = sym_timing ? tnow() : 0;
				cover_group(jobs[tasks[t].first], tasks[t].second, sep, M);
				if (sym_timing) fprintf(stderr, "[sym]   chunk %d group %d: %.3f ms\n", tasks[t].first, tasks[t].second, tnow() - t0);
			};
			if (tasks.size() == 1 || threads == 1) for (int t = 0; t < (int)tasks.size(); t++) fn(t);
			else sym_pool().run((int)tasks.size(), fn);
		}
	}
	tick("separators");
	// ---- permutation: by separator level, original order inside a level (a stable counting sort) ---------------------
	std::vector<int>&perm = ch.perm, &pinv = ch.pinv;
	perm.resize(M); pinv.resize(M);
	{
		int cnt[36] = { 0 };
		for (int i = 0; i < M; i++) cnt[sep[i] + 1]++;
		for (int l = 0; l < 35; l++) cnt[l + 1] += cnt[l];
		for (int i = 0; i < M; i++) perm[cnt[sep[i]]++] = i;
		for (int i = 0; i < M; i++) pinv[perm[i]] = i;
	}
	// ---- ownership (distributed factorisation): an edge between poses of different blocks crosses a cut of level > kb, and one of
	// its endpoints sits in that cut's separator -- so a pose with sep <= kb sees, and ever gets filled to, only poses of its own
	// block and separator poses above: its column is its block's; the columns with sep > kb (the tail of the ordering) are shared
	ch.col_owner.clear();
	ch.first_shared = M;
	if (block_maps > 0)
	{
		const int kb = bitlen((unsigned)(block_maps - 1)); // (blocks are 2^k local maps: origins of one block differ in their low kb bits only)
		ch.col_owner.resize(M);
		int fs = M;
		for (int j = 0; j < M; j++)
		{
			const int p = perm[j];
			const bool shared = sep[p] > kb;
			ch.col_owner[j] = shared ? -1 : origin[p] / block_maps;
			if (shared && j < fs) fs = j;
		}
		ch.first_shared = fs;
	}
	// ---- strict lower adjacency by row, new numbering ------------------------------------------------------------------
	std::vector<int>&rcnt = w.rcnt, &radj = w.radj, &fill = w.fill;
	static const int max_threads_a = getenv("LSFM_SYM_THREADS") ? std::max(1, atoi(getenv("LSFM_SYM_THREADS"))) : 8;
	const int athreads = (nnzb < 60000 || block_maps > 0) ? 1 : std::min<int>(max_threads_a, std::max(1u, std::thread::hardware_concurrency() / 2));
	if (athreads > 1 && nnzb >= 400000)
	{
		// a stable counting sort by row over ranges of the keys, one range per thread: every range counts its entries per row, the running
		// sums over (row, range) give every range its first place in every row, every range writes its entries in key order -- the rows
		// come out as the one-thread pass leaves them
		const int T = athreads;
		std::vector<int>& cnt = w.cfill; // [T][M]
		cnt.assign((size_t)T * M, 0);
		auto range = [&](int t, int& e0, int& e1) { e0 = (int)((long)nnzb * t / T); e1 = (int)((long)nnzb * (t + 1) / T); };
		sym_pool().run(T, std::function<void(int)>([&](int t) {
			int e0, e1;
			range(t, e0, e1);
			int* c = cnt.data() + (size_t)t * M;
			for (int e = e0; e < e1; e++)
			{
				const int p = (int)(keys[e] >> 32), q = (int)(keys[e] & 0xffffffffull);
				if (p != q) c[std::max(pinv[p], pinv[q])]++;
			}
		}));
		rcnt.resize(M + 1);
		int run = 0;
		for (int i = 0; i < M; i++)
		{
			rcnt[i] = run;
			for (int t = 0; t < T; t++) { const int x = cnt[(size_t)t * M + i]; cnt[(size_t)t * M + i] = run; run += x; }
		}
		rcnt[M] = run;
		radj.resize(run);
		sym_pool().run(T, std::function<void(int)>([&](int t) {
			int e0, e1;
			range(t, e0, e1);
			int* c = cnt.data() + (size_t)t * M;
			for (int e = e0; e < e1; e++)
			{
				const int p = (int)(keys[e] >> 32), q = (int)(keys[e] & 0xffffffffull);
				if (p == q) continue;
				const int a = std::min(pinv[p], pinv[q]), b = std::max(pinv[p], pinv[q]);
				radj[c[b]++] = a;
			}
		}));
	}
	else
	{
		rcnt.assign(M + 1, 0);
		for (int e = 0; e < nnzb; e++)
		{
			const int p = (int)(keys[e] >> 32), q = (int)(keys[e] & 0xffffffffull);
			if (p != q) rcnt[std::max(pinv[p], pinv[q]) + 1]++;
		}
		for (int i = 0; i < M; i++) rcnt[i + 1] += rcnt[i];
		radj.resize(rcnt[M]);
		fill.assign(M, 0);
		for (int e = 0; e < nnzb; e++)
		{
			const int p = (int)(keys[e] >> 32), q = (int)(keys[e] & 0xffffffffull);
			if (p == q) continue;
			const int a = std::min(pinv[p], pinv[q]), b = std::max(pinv[p], pinv[q]);
			radj[rcnt[b] + fill[b]++] = a;
		}
	}
	tick("perm + adjacency");
	// ---- elimination tree (ancestor path compression), column counts, column patterns by row sub-tree walks ----------
	std::vector<int>&parent = ch.parent, &ccount = ch.ccount, &anc = w.anc, &mark = w.mark;
	parent.assign(M, -1); anc.assign(M, -1);
	// the rows (new numbering) of every chunk, ascending: perm is sorted by (separator level, old index)
	std::vector<std::vector<int>> crow(nchunk);
	if (nchunk > 1)
	{
		std::vector<int> chunk_of(M);
		for (int c = 0; c < nchunk; c++) { crow[c].reserve(cut_row[c + 1] - cut_row[c]); for (int p = cut_row[c]; p < cut_row[c + 1]; p++) chunk_of[p] = c; }
		for (int k = 0; k < M; k++) crow[chunk_of[perm[k]]].push_back(k);
	}
	auto for_rows = [&](int c, auto&& body) {
		if (nchunk == 1) { for (int k = 0; k < M; k++) body(k); }
		else for (int k : crow[c]) body(k);
	};
	par_chunks([&](int c) {
		for_rows(c, [&](int k) {
			for (int t = rcnt[k]; t < rcnt[k + 1]; t++)
			{
				int i = radj[t];
				while (i != -1 && i < k) { const int nx = anc[i]; anc[i] = k; if (nx == -1) parent[i] = k; i = nx; }
			}
		});
	});
	tick("etree");
	// Column counts and column patterns: row k of L is the union of the tree paths from k's neighbours up to k (a row sub-tree walk
	// with a mark per visited column).  The rows are cut into RANGES, taken by the host threads in any order -- a range needs the
	// finished tree and a mark array of its own, nothing of another range: pass 1 counts what every range adds to every column
	// (cnt[range][column]), the running sums over the ranges give every range the place of ITS first entry in every column, pass 2
	// repeats the walk and writes -- the ranges ascend and so do the rows inside one, so a column's row indices come out ascending,
	// exactly as the one-thread walk over all rows leaves them.  (One system of 16 386 poses, 2.07 M blocks of L: 9 of its
	// analysis's 30 ms were these two walks on one thread.)
	std::vector<int>&colptr = ch.colptr, &rowidx = ch.rowidx;
	static const int max_threads_w = getenv("LSFM_SYM_THREADS") ? std::max(1, atoi(getenv("LSFM_SYM_THREADS"))) : 8;
	const int wthreads = (nnzb < 60000 || block_maps > 0) ? 1 : std::min<int>(max_threads_w, std::max(1u, std::thread::hardware_concurrency() / 2));
	// (a level of many independent systems is cut into chunks of whole systems instead, one thread each, below: no counters per range,
	// no sums over them -- 16 384 columns x 32 ranges of those cost a level of small systems more than its walks)
	if (wthreads > 1 && nchunk < 4)
	{
		// ranges of about equal numbers of adjacency entries, four per thread (the paths of the last rows -- the top separators -- are the long ones)
		const int R = 4 * wthreads;
		std::vector<int> rb(1, 0);
		{
			const long total = rcnt[M];
			for (int k = 0, r = 1; k < M && r < R; k++)
				if ((long)rcnt[k + 1] * R >= total * r) { if (k + 1 > rb.back() && k + 1 < M) rb.push_back(k + 1); r++; }
			rb.push_back(M);
		}
		const int nr = (int)rb.size() - 1;
		std::vector<int>& cnt = w.cfill; // [nr][M]
		cnt.assign((size_t)nr * M, 0);
		const std::function<void(int)> count_fn = [&](int r) {
			std::vector<int>& mk = workspace().mark;
			mk.assign(M, -1);
			int* c = cnt.data() + (size_t)r * M;
			for (int k = rb[r]; k < rb[r + 1]; k++)
			{
				mk[k] = k;
				for (int t = rcnt[k]; t < rcnt[k + 1]; t++)
					for (int i = radj[t]; mk[i] != k; i = parent[i]) { c[i]++; mk[i] = k; }
			}
		};
		sym_pool().run(nr, count_fn);
		tick("column counts");
		colptr.assign(M + 1, 0);
		ccount.resize(M);
		sym_pool().run(wthreads, std::function<void(int)>([&](int t) {
			const int j0 = (int)((long)M * t / wthreads), j1 = (int)((long)M * (t + 1) / wthreads);
			for (int r = 0; r < nr; r++)
			{
				int* c = cnt.data() + (size_t)r * M;
				if (r == 0) for (int j = j0; j < j1; j++) { const int x = c[j]; c[j] = 1; ccount[j] = 1 + x; } // (the pivot block first)
				else for (int j = j0; j < j1; j++) { const int x = c[j]; c[j] = ccount[j]; ccount[j] += x; }
			}
		}));
		for (int j = 0; j < M; j++) colptr[j + 1] = colptr[j] + ccount[j];
		rowidx.resize(colptr[M]);
		for (int j = 0; j < M; j++) rowidx[colptr[j]] = j;
		const std::function<void(int)> fill_fn = [&](int r) {
			std::vector<int>& mk = workspace().mark;
			mk.assign(M, -1);
			int* c = cnt.data() + (size_t)r * M;
			for (int k = rb[r]; k < rb[r + 1]; k++)
			{
				mk[k] = k;
				for (int t = rcnt[k]; t < rcnt[k + 1]; t++)
					for (int i = radj[t]; mk[i] != k; i = parent[i]) { rowidx[colptr[i] + c[i]++] = k; mk[i] = k; }
			}
		};
		sym_pool().run(nr, fill_fn);
	}
	else
	{
		// (a row only ever touches columns of its own system: the chunks share mark, ccount and cfill without meeting)
		mark.assign(M, -1); ccount.assign(M, 1);
		par_chunks([&](int c) {
			for_rows(c, [&](int k) {
				mark[k] = k;
				for (int t = rcnt[k]; t < rcnt[k + 1]; t++)
					for (int i = radj[t]; mark[i] != k; i = parent[i]) { ccount[i]++; mark[i] = k; }
			});
		});
		tick("column counts");
		colptr.assign(M + 1, 0);
		for (int j = 0; j < M; j++) colptr[j + 1] = colptr[j] + ccount[j];
		rowidx.resize(colptr[M]);
		w.cfill.assign(M, 1);
		for (int j = 0; j < M; j++) rowidx[colptr[j]] = j;
		std::fill(mark.begin(), mark.end(), -1);
		par_chunks([&](int c) {
			for_rows(c, [&](int k) {
				mark[k] = k;
				for (int t = rcnt[k]; t < rcnt[k + 1]; t++)
					for (int i = radj[t]; mark[i] != k; i = parent[i]) { rowidx[colptr[i] + w.cfill[i]++] = k; mark[i] = k; }
			});
		});
	}
	const int nnzL = colptr[M];
	ch.work_total = ch.work_shared = 0;
	for (int j = 0; j < M; j++)
	{
		const double w = 0.5 * (double)ccount[j] * (ccount[j] + 1);
		ch.work_total += w;
		if (j >= ch.first_shared) ch.work_shared += w;
	}
	tick("column patterns");
	// ---- level sets (height above the leaves); the narrow top (<= 2 columns per level) becomes the tail ---------------
	std::vector<int>& lev = w.lev;
	lev.assign(M, 0);
	int nlev = 0;
	for (int j = 0; j < M; j++)
	{
		if (parent[j] >= 0) lev[parent[j]] = std::max(lev[parent[j]], lev[j] + 1);
		nlev = std::max(nlev, lev[j] + 1);
	}
	std::vector<int>& lcount = w.lcount;
	lcount.assign(nlev + 1, 0);
	for (int j = 0; j < M; j++) lcount[lev[j] + 1]++;
	int tail_level = nlev;
	while (tail_level > 0 && lcount[tail_level] <= 2) tail_level--;
	for (int l = 0; l < nlev; l++) lcount[l + 1] += lcount[l];
	std::vector<int>& order = ch.order;
	order.resize(M);
	w.lfill.assign(nlev, 0);
	for (int j = 0; j < M; j++) order[lcount[lev[j]] + w.lfill[lev[j]]++] = j; // ascending j inside a level
	ch.M = M; ch.nnzL = nnzL; ch.nlevels = tail_level;
	ch.level_ptr.assign(lcount.begin(), lcount.begin() + tail_level + 1);
	ch.tail_begin = lcount[tail_level];
	// tail columns must be walked in ascending index (= a topological order), not level order
	std::sort(order.begin() + ch.tail_begin, order.end());
	// ---- tasks.  Sub-trees of at most task_x blocks are walked by one work-group each (task level 0); above them every
	// chain of the tree (a separator of the dissection: each column the only large child of the next) is one task,
	// levelled by the chains below it. ----
	static const int task_x = getenv("LSFM_TASK_X") ? atoi(getenv("LSFM_TASK_X")) : 90;
	ch.task_x = task_x;
	std::vector<int>&size = w.size, &ntc = w.ntc, &topchild = w.topchild, &task = w.task, &tlev = w.tlev;
	// "size" of a sub-tree = its blocks (pivot blocks included): a leaf task must fit LDS whole (small-task kernels)
	size.assign(ccount.begin(), ccount.end());
	ntc.assign(M, 0); topchild.assign(M, -1); task.assign(M, -1); tlev.clear();
	for (int j = 0; j < M; j++) if (parent[j] >= 0) size[parent[j]] += size[j];
	// (distributed factorisation: a shared column is never part of a leaf task -- those run before the ranks' sums are exchanged)
	for (int j = ch.first_shared; j < M; j++) size[j] = std::max(size[j], task_x + 1);
	for (int j = 0; j < M; j++)
		if (size[j] > task_x && parent[j] >= 0) { ntc[parent[j]]++; topchild[parent[j]] = j; }
	int ntasks = 0;
	// large columns, ascending: children first
	for (int j = 0; j < M; j++)
	{
		if (size[j] <= task_x) continue;
		if (ntc[j] == 1) { task[j] = task[topchild[j]]; continue; }
		task[j] = ntasks++;
		tlev.push_back(1);
	}
	// level of a chain = 1 + highest chain below it (ascending order sees the children first)
	for (int j = 0; j < M; j++)
	{
		if (size[j] <= task_x) continue;
		const int pj = parent[j];
		if (pj >= 0 && task[pj] != task[j]) tlev[task[pj]] = std::max(tlev[task[pj]], tlev[task[j]] + 1);
	}
	// small sub-trees, descending: parents first
	for (int j = M - 1; j >= 0; j--)
	{
		if (size[j] > task_x) continue;
		const int pj = parent[j];
		if (pj >= 0 && size[pj] <= task_x) task[j] = task[pj];
		else { task[j] = ntasks++; tlev.push_back(0); }
	}
	int ntl = 0;
	for (int t = 0; t < ntasks; t++) ntl = std::max(ntl, tlev[t] + 1);
	// order tasks by level, columns by (task order, ascending index)
	std::vector<int> tl_count(ntl + 1, 0), tpos(ntasks), tsize(ntasks, 0);
	for (int t = 0; t < ntasks; t++) tl_count[tlev[t] + 1]++;
	for (int l = 0; l < ntl; l++) tl_count[l + 1] += tl_count[l];
	// tasks whose blocks fit LDS whole go first in their level (they take the small-task kernels)
	std::vector<long> tblocks(ntasks, 0), tcolsn(ntasks, 0);
	for (int j = 0; j < M; j++) { tblocks[task[j]] += ccount[j]; tcolsn[task[j]]++; }
	auto task_lds = [&](int t) { return tblocks[t] * (288 + 4) + (5 * tcolsn[t] + 1) * 4 + 16; };
	const long small_cap = 60 * 1024;
	ch.tlevel_nsmall.assign(ntl, 0);
	ch.tlevel_small_lds.assign(ntl, 0);
	{
		std::vector<int> f(ntl, 0);
		for (int pass = 0; pass < 2; pass++)
			for (int t = 0; t < ntasks; t++)
			{
				const bool small = task_lds(t) <= small_cap;
				if (small != (pass == 0)) continue;
				tpos[t] = tl_count[tlev[t]] + f[tlev[t]]++;
				if (small) { ch.tlevel_nsmall[tlev[t]]++; ch.tlevel_small_lds[tlev[t]] = std::max(ch.tlevel_small_lds[tlev[t]], (int)task_lds(t)); }
			}
	}
	for (int j = 0; j < M; j++) tsize[tpos[task[j]]]++;
	std::vector<int>&tptr = ch.task_ptr, &tcols = ch.task_cols;
	tptr.assign(ntasks + 1, 0); tcols.resize(M);
	std::vector<int> tf(ntasks, 0);
	for (int t = 0; t < ntasks; t++) tptr[t + 1] = tptr[t] + tsize[t];
	for (int j = 0; j < M; j++) { const int t = tpos[task[j]]; tcols[tptr[t] + tf[t]++] = j; }
	ch.tlevel_ptr = tl_count;
	ch.tlevel_maxsize.assign(ntl, 0);
	std::vector<int>&ctask = ch.col_task, &clpos = ch.col_lpos;
	ctask.resize(M); clpos.resize(M);
	for (int t = 0; t < ntasks; t++)
		for (int k = tptr[t]; k < tptr[t + 1]; k++) { ctask[tcols[k]] = t; clpos[tcols[k]] = k - tptr[t]; }
	for (int l = 0; l < ntl; l++)
		for (int t = tl_count[l]; t < tl_count[l + 1]; t++) ch.tlevel_maxsize[l] = std::max(ch.tlevel_maxsize[l], tsize[t]);
	ch.tlevel_col0.assign(ntl + 1, 0);
	for (int l = 0; l <= ntl; l++) ch.tlevel_col0[l] = tptr[tl_count[l]];
	std::vector<int>& nin = ch.col_nin;
	nin.assign(M, 0);
	ch.tlevel_outer.assign(ntl, 0);
	for (int j = 0; j < M; j++)
	{
		int m = 0;
		while (colptr[j] + 1 + m < colptr[j + 1] && task[rowidx[colptr[j] + 1 + m]] == task[j]) m++;
		nin[j] = m;
		const int no = ccount[j] - 1 - m, l = tlev[task[j]];
		ch.tlevel_outer[l] = std::max(ch.tlevel_outer[l], no * (no + 1) / 2);
	}
	tick("levels + tasks");
	// ---- supernode groups over the large columns (the factorisation above the leaf tasks) ----------------------------
	{
		std::vector<int> grp(M, -1), gc0, gs, glev;
		for (int j = 0; j < M; j++)
		{
			if (size[j] <= task_x) continue;
			const bool join = j > 0 && size[j - 1] > task_x && parent[j - 1] == j && ccount[j - 1] == ccount[j] + 1 && gs[grp[j - 1]] < gs_cap &&
			                  j != ch.first_shared; // (a run never spans interior and shared columns)
			if (join) { grp[j] = grp[j - 1]; gs[grp[j]]++; }
			else { grp[j] = (int)gc0.size(); gc0.push_back(j); gs.push_back(1); glev.push_back(0); }
		}
		const int ng = (int)gc0.size();
		int ngl = 0;
		for (int g = 0; g < ng; g++) // ascending first column: children before parents
		{
			const int pj = parent[gc0[g] + gs[g] - 1];
			if (pj >= 0) glev[grp[pj]] = std::max(glev[grp[pj]], glev[g] + 1);
			ngl = std::max(ngl, glev[g] + 1);
		}
		std::vector<int> gl_count(ngl + 1, 0), gfill(ngl, 0);
		ch.grp_c0.resize(ng); ch.grp_s.resize(ng); ch.grp_nr.resize(ng);
		for (int g = 0; g < ng; g++) gl_count[glev[g] + 1]++;
		for (int l = 0; l < ngl; l++) gl_count[l + 1] += gl_count[l];
		ch.glevel_maxnr.assign(ngl, 0);
		ch.glevel_maxs.assign(ngl, 1);
		for (int g = 0; g < ng; g++)
		{
			const int at = gl_count[glev[g]] + gfill[glev[g]]++;
			ch.grp_c0[at] = gc0[g]; ch.grp_s[at] = gs[g]; ch.grp_nr[at] = ccount[gc0[g] + gs[g] - 1] - 1;
			ch.glevel_maxnr[glev[g]] = std::max(ch.glevel_maxnr[glev[g]], ch.grp_nr[at]);
			ch.glevel_maxs[glev[g]] = std::max(ch.glevel_maxs[glev[g]], gs[g]);
		}
		ch.ngroups = ng;
		ch.glevel_ptr = gl_count;
		ch.glevel_owned.assign(ngl, 0); ch.glevel_shared.assign(ngl, 0);
		for (int g = 0; g < ng; g++) (gc0[g] >= ch.first_shared ? ch.glevel_shared : ch.glevel_owned)[glev[g]] = 1;
	}
	tick("groups");
}

} // namespace lsfm

#include "../../include/lsfm.h"

extern "C" int lsfm_symbolic_analyse(int m, const int* rowptr, const int* colidx, const int* origin, int reps, int* perm, int* colptr, int* rowidx,
                                     int cap, int* info, double* avg_ms)
{
	if (m <= 0 || !rowptr || !colidx || rowptr[0] != 0) return LSFM_ERR_ARG;
	const int nnzb = rowptr[m];
	std::vector<unsigned long long> keys((size_t)nnzb);
	for (int p = 0; p < m; p++)
	{
		if (rowptr[p + 1] <= rowptr[p] || colidx[rowptr[p]] != p) return LSFM_ERR_ARG; // every block row starts with its diagonal block
		for (int k = rowptr[p]; k < rowptr[p + 1]; k++)
		{
			if (colidx[k] < p || colidx[k] >= m || (k > rowptr[p] && colidx[k] <= colidx[k - 1])) return LSFM_ERR_ARG;
			keys[k] = ((unsigned long long)(unsigned)p << 32) | (unsigned)colidx[k];
		}
	}
	std::vector<int> org(m);
	for (int p = 0; p < m; p++) org[p] = origin ? origin[p] : p;
	lsfm::CholSymbolic sym;
	if (reps < 1) reps = 1;
	lsfm::chol_symbolic(keys.data(), nnzb, org.data(), m, sym); // (workspace and code warm)
	const auto t0 = std::chrono::steady_clock::now();
	for (int r = 0; r < reps; r++) lsfm::chol_symbolic(keys.data(), nnzb, org.data(), m, sym);
	if (avg_ms) *avg_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / reps;
	if (info)
	{
		int height = 0, nsep = 0;
		{
			std::vector<int> lev(m, 0);
			for (int j = 0; j < m; j++)
			{
				if (sym.parent[j] >= 0) lev[sym.parent[j]] = std::max(lev[sym.parent[j]], lev[j] + 1);
				height = std::max(height, lev[j] + 1);
			}
			// poses eliminated after every pose of a lower separator level = the perm positions past the first non-leaf; count
			// the columns whose sub-tree is larger than a leaf task instead (what the group kernels factor)
			for (int j = 0; j < m; j++) nsep += sym.col_task[j] >= (sym.tlevel_ptr.size() > 1 ? sym.tlevel_ptr[1] : 0);
		}
		info[0] = sym.nnzL; info[1] = height; info[2] = sym.ngroups; info[3] = (int)sym.glevel_ptr.size() - 1;
		info[4] = sym.tlevel_ptr.size() > 1 ? sym.tlevel_ptr[1] : 0; info[5] = nsep; info[6] = 0; info[7] = 0;
	}
	if (perm) memcpy(perm, sym.perm.data(), (size_t)m * sizeof(int));
	if (colptr) memcpy(colptr, sym.colptr.data(), (size_t)(m + 1) * sizeof(int));
	if (rowidx)
	{
		if (cap < sym.nnzL) return LSFM_ERR_ARG;
		memcpy(rowidx, sym.rowidx.data(), (size_t)sym.nnzL * sizeof(int));
	}
	return LSFM_OK;
}
