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

// greedy vertex cover of the crossing edges among keys[kb, ke) -- the rows of a set of whole systems -- top level first: sep[pose] =
// level of the separator the pose was put into (0: none).  Writes sep at the poses of these rows only.
static void cover_edges(const unsigned long long* keys, int kb, int ke, const int* origin, int M, std::vector<int>& sep, Workspace& w)
{
	// the off-diagonal edges bucketed by the tree level they cross (counting sort, order inside a level kept)
	const int nk = ke - kb;
	w.edge_lev.resize(nk); w.edge_a.resize(nk); w.edge_b.resize(nk); w.eorder.resize(nk);
	w.lev_cnt.assign(35, 0);
	int ne = 0;
	for (int e = kb; e < ke; e++)
	{
		const int p = (int)(keys[e] >> 32), q = (int)(keys[e] & 0xffffffffull);
		if (p == q) continue;
		const int l = bitlen((unsigned)(origin[p] ^ origin[q]));
		w.edge_a[ne] = p; w.edge_b[ne] = q; w.edge_lev[ne] = l;
		w.lev_cnt[l + 1]++;
		ne++;
	}
	for (int l = 0; l < 34; l++) w.lev_cnt[l + 1] += w.lev_cnt[l];
	{
		int pos[35];
		for (int l = 0; l < 35; l++) pos[l] = w.lev_cnt[l];
		for (int e = 0; e < ne; e++) w.eorder[pos[w.edge_lev[e]]++] = e;
	}
	if ((int)w.vid.size() != M) w.vid.assign(M, -1); // (left all -1 by every call)
	std::vector<std::pair<int, int>> live;
	for (int l = 33; l >= 1; l--)
	{
		const int b0 = w.lev_cnt[l], b1 = w.lev_cnt[l + 1];
		if (b0 == b1) continue;
		// uncovered edges of the level
		live.clear();
		for (int t = b0; t < b1; t++)
		{
			const int e = w.eorder[t], p = w.edge_a[e], q = w.edge_b[e];
			if (sep[p] < l && sep[q] < l) live.emplace_back(p, q);
		}
		if (live.empty()) continue;
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
}

void chol_symbolic(const unsigned long long* keys, int nnzb, const int* origin, int M, CholSymbolic& ch, int block_maps)
{
	Workspace& w = workspace();
	static const bool sym_timing = getenv("LSFM_SYM_TIMING") != nullptr;
	auto tnow = []() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
	double tprev = tnow();
	auto tick = [&](const char* what) { if (sym_timing) { const double n = tnow(); fprintf(stderr, "[sym] %-28s %8.3f ms\n", what, n - tprev); tprev = n; } };
	// ---- separators: greedy vertex cover of the crossing edges, top level first --------------------------------------
	std::vector<int>& sep = w.sep;
	sep.assign(M, 0);
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
	par_chunks([&](int c) {
		const double t0 = sym_timing ? tnow() : 0;
		cover_edges(keys, cut_key[c], cut_key[c + 1], origin, M, sep, workspace());
		if (sym_timing) fprintf(stderr, "[sym]   chunk %d: rows %d..%d, %d keys, cover %.3f ms\n", c, cut_row[c], cut_row[c + 1], cut_key[c + 1] - cut_key[c], tnow() - t0);
	});
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
	mark.assign(M, -1); ccount.assign(M, 1);
	par_chunks([&](int c) {
		for_rows(c, [&](int k) {
			mark[k] = k;
			for (int t = rcnt[k]; t < rcnt[k + 1]; t++)
				for (int i = radj[t]; mark[i] != k; i = parent[i]) { ccount[i]++; mark[i] = k; }
		});
	});
	tick("column counts");
	std::vector<int>&colptr = ch.colptr, &rowidx = ch.rowidx;
	colptr.assign(M + 1, 0);
	for (int j = 0; j < M; j++) colptr[j + 1] = colptr[j] + ccount[j];
	const int nnzL = colptr[M];
	ch.work_total = ch.work_shared = 0;
	for (int j = 0; j < M; j++)
	{
		const double w = 0.5 * (double)ccount[j] * (ccount[j] + 1);
		ch.work_total += w;
		if (j >= ch.first_shared) ch.work_shared += w;
	}
	rowidx.resize(nnzL);
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
