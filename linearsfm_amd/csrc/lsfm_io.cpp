// File formats of the reference, host side: localmap_k.txt reader (Imp.cpp:3044-3132 / 6660-6754) and the result
// writers (Imp.cpp:2102-2117, 7876-7967), byte-compatible "%lf" output.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/lsfm.h"

namespace {
template <class T> T* xalloc(size_t n) { return static_cast<T*>(malloc((n ? n : 1) * sizeof(T))); }
}

// Reader.  The reference reads a local map token by token with fscanf (Imp.cpp:3044-3132 / 6660-6754); a 3499-map set
// is some GB of text and that loop is the largest wall-clock item outside the timed region.  Here a file is read in
// one piece and tokenised in place (integers and the common decimals by hand, exactly rounded, the rest by strtod: the values are
// bit-identical to "%lf"); lsfm_read_localmaps() spreads the files of a set over host threads.
#include <atomic>
#include <string>
#include <thread>

#define LSFM_NODE_MAGIC 1279870541 /* 'LSFM': first token of the tree-node trailer of a local-map file */

namespace {

struct Tok {
	const char* p;
	const char* end;
	bool ok = true;
	void skip() { while (p < end && (*p == ' ' || *p == '\n' || *p == '\t' || *p == '\r' || *p == '\f' || *p == '\v')) p++; }
	int geti()
	{
		skip();
		if (p >= end) { ok = false; return 0; }
		bool neg = false;
		if (*p == '-' || *p == '+') { neg = *p == '-'; p++; }
		if (p >= end || *p < '0' || *p > '9') { ok = false; return 0; }
		long long v = 0;
		while (p < end && *p >= '0' && *p <= '9') { v = v * 10 + (*p - '0'); p++; }
		return (int)(neg ? -v : v);
	}
	// Decimal -> double.  Fast path (Clinger): a mantissa of at most 2^53 and a power of ten up to 10^22 are exact
	// doubles, so one IEEE multiplication or division gives the correctly rounded value -- the same bits as "%lf".
	// Everything else (more digits, large exponents, inf/nan, hexadecimal) goes to strtod on a bounded copy.
	double slow(const char* start)
	{
		char tmp[512];
		size_t n = std::min<size_t>(sizeof tmp - 1, (size_t)(end - start));
		memcpy(tmp, start, n); tmp[n] = 0;
		char* e = nullptr;
		const double v = strtod(tmp, &e);
		if (e == tmp) { ok = false; return 0; }
		p = start + (e - tmp);
		return v;
	}
	double getd()
	{
		static const double p10[23] = { 1e0, 1e1, 1e2, 1e3, 1e4, 1e5, 1e6, 1e7, 1e8, 1e9, 1e10, 1e11, 1e12, 1e13, 1e14, 1e15, 1e16,
			                            1e17, 1e18, 1e19, 1e20, 1e21, 1e22 };
		skip();
		if (p >= end) { ok = false; return 0; }
		const char* start = p;
		const char* q = p;
		bool neg = false;
		if (*q == '-' || *q == '+') { neg = *q == '-'; q++; }
		unsigned long long m = 0;
		int nd = 0, e10 = 0;
		bool any = false;
		while (q < end && *q >= '0' && *q <= '9') { if (nd < 19) { m = m * 10 + (unsigned)(*q - '0'); if (m) nd++; } else e10++; q++; any = true; }
		if (q < end && *q == '.')
		{
			q++;
			while (q < end && *q >= '0' && *q <= '9') { if (nd < 19) { m = m * 10 + (unsigned)(*q - '0'); if (m) nd++; e10--; } q++; any = true; }
		}
		if (!any) return slow(start); // inf, nan, or not a number
		bool inexact = nd >= 19; // digits were dropped: let strtod round
		if (q < end && (*q == 'e' || *q == 'E'))
		{
			const char* r = q + 1;
			bool eneg = false;
			if (r < end && (*r == '-' || *r == '+')) { eneg = *r == '-'; r++; }
			if (r < end && *r >= '0' && *r <= '9')
			{
				int ev = 0;
				while (r < end && *r >= '0' && *r <= '9') { if (ev < 100000) ev = ev * 10 + (*r - '0'); r++; }
				e10 += eneg ? -ev : ev;
				q = r;
			}
		}
		else if (q < end && (*q == 'x' || *q == 'X')) return slow(start); // hexadecimal float
		if (inexact || m > (1ull << 53) || e10 > 22 || e10 < -22) return slow(start);
		double v = (double)m;
		if (e10 > 0) v *= p10[e10]; else if (e10 < 0) v /= p10[-e10];
		p = q;
		return neg ? -v : v;
	}
};

int parse_localmap(const char* buf, size_t len, int mono, lsfm_map* g)
{
	Tok t{ buf, buf + len };
	memset(g, 0, sizeof *g);
	g->Ref = t.geti();
	g->FRef = g->Ref;
	g->Sign = 1;
	if (mono)
	{
		g->ScaP = t.geti(); g->FScaP = g->ScaP;
		g->Fix = t.geti(); g->FFix = g->Fix;
		g->Sign = t.geti();
	}
	const int r = t.geti();
	if (!t.ok || r < 0) return LSFM_ERR_IO;
	g->stno = xalloc<int>(r); g->stVal = xalloc<double>(r);
	for (int i = 0; i < r && t.ok; i++) { g->stno[i] = t.geti(); g->stVal[i] = t.getd(); }
	g->m = t.geti(); g->n = t.geti(); g->nU = t.geti();
	if (!t.ok || g->nU < 0 || g->m < 0 || g->n < 0 || 6L * g->m + 3L * g->n != r) { lsfm_map_release(g); return LSFM_ERR_IO; }
	g->U = xalloc<double>((size_t)g->nU * 36); g->Ui = xalloc<int>(g->nU); g->Uj = xalloc<int>(g->nU);
	for (long i = 0; i < 36L * g->nU && t.ok; i++) g->U[i] = t.getd();
	for (int i = 0; i < g->nU && t.ok; i++) g->Ui[i] = t.geti();
	for (int i = 0; i < g->nU && t.ok; i++) g->Uj[i] = t.geti();
	g->nW = t.geti();
	if (!t.ok || g->nW < 0) { lsfm_map_release(g); return LSFM_ERR_IO; }
	g->W = xalloc<double>((size_t)g->nW * 18); g->photo = xalloc<int>(g->nW); g->feature = xalloc<int>(g->nW);
	for (long i = 0; i < 18L * g->nW && t.ok; i++) g->W[i] = t.getd();
	for (int i = 0; i < g->nW && t.ok; i++) g->photo[i] = t.geti();
	for (int i = 0; i < g->nW && t.ok; i++) g->feature[i] = t.geti();
	g->V = xalloc<double>((size_t)g->n * 9); g->FBlock = xalloc<int>(g->n);
	g->pose_origin = NULL;
	for (long i = 0; i < 9L * g->n && t.ok; i++) g->V[i] = t.getd();
	for (int i = 0; i < g->n && t.ok; i++) g->FBlock[i] = t.geti();
	if (!t.ok) { lsfm_map_release(g); return LSFM_ERR_IO; }
	// optional trailer of a TREE NODE written by lsfm_write_localmap (the reference's fscanf sequence ends with FBlock and never
	// looks further): LSFM_NODE_MAGIC, then FRef FScaP FFix and the m pose origins -- what a node that is joined further needs
	t.skip();
	if (t.p < t.end)
	{
		Tok u = t;
		if (u.geti() == LSFM_NODE_MAGIC && u.ok)
		{
			g->FRef = u.geti(); g->FScaP = u.geti(); g->FFix = u.geti();
			const int no = u.geti();
			if (!u.ok || (no != 0 && no != g->m)) { lsfm_map_release(g); return LSFM_ERR_IO; }
			if (no)
			{
				g->pose_origin = xalloc<int>(g->m);
				for (int i = 0; i < g->m && u.ok; i++) g->pose_origin[i] = u.geti();
			}
			if (!u.ok) { lsfm_map_release(g); return LSFM_ERR_IO; }
		}
	}
	return LSFM_OK;
}

} // namespace

extern "C" {

int lsfm_read_localmap(const char* path, int mono, lsfm_map* g)
{
	if (!path || !g) return LSFM_ERR_ARG;
	FILE* f = fopen(path, "rb");
	if (!f) return LSFM_ERR_IO;
	std::vector<char> buf;
	if (fseek(f, 0, SEEK_END) == 0)
	{
		const long sz = ftell(f);
		rewind(f);
		if (sz > 0) { buf.resize((size_t)sz); buf.resize(fread(buf.data(), 1, (size_t)sz, f)); }
	}
	if (buf.empty())
	{
		// not seekable (pipe): read in pieces
		char tmp[65536];
		size_t n;
		while ((n = fread(tmp, 1, sizeof tmp, f)) > 0) buf.insert(buf.end(), tmp, tmp + n);
	}
	fclose(f);
	return parse_localmap(buf.data(), buf.size(), mono, g);
}

// localmap_<first>.txt .. localmap_<first+count-1>.txt of a directory (naming of the reference, Imp.cpp:125) on
// `threads` host threads (<= 0: one per core, at most 32).  out[count] is filled in order; on failure everything read so
// far is released, LSFM_ERR_IO is returned and *failed (optional) is the lowest number of a file that could not be read.
int lsfm_read_localmaps(const char* dir, int first, int count, int mono, int threads, lsfm_map* out, int* failed)
{
	if (!dir || !out || count < 0) return LSFM_ERR_ARG;
	if (failed) *failed = 0;
	if (threads <= 0) threads = (int)std::min(32u, std::max(1u, std::thread::hardware_concurrency()));
	threads = std::max(1, std::min(threads, count));
	const int none = 0x7fffffff;
	std::atomic<int> next(0), bad(none);
	for (int k = 0; k < count; k++) memset(&out[k], 0, sizeof(lsfm_map));
	auto work = [&]() {
		for (;;)
		{
			const int k = next.fetch_add(1);
			if (k >= count) return;
			const std::string fn = std::string(dir) + "/localmap_" + std::to_string(first + k) + ".txt";
			if (lsfm_read_localmap(fn.c_str(), mono, &out[k]) != LSFM_OK)
			{
				// keep the lowest failing number, whatever the thread timing
				int cur = bad.load();
				while (first + k < cur && !bad.compare_exchange_weak(cur, first + k)) {}
			}
		}
	};
	std::vector<std::thread> pool;
	for (int t = 1; t < threads; t++) pool.emplace_back(work);
	work();
	for (auto& th : pool) th.join();
	if (bad.load() != none)
	{
		for (int k = 0; k < count; k++) lsfm_map_release(&out[k]);
		if (failed) *failed = bad.load();
		return LSFM_ERR_IO;
	}
	return LSFM_OK;
}

// A map in the local-map text format the readers take (Imp.cpp:3044-3132 / 6660-6754), values at %.17g so that
// writing and reading back is the identity.  The reference keeps the final map's information matrix (DOC.pdf p.1) but
// never writes it; with this any tree node (lsfm_tree_download, lsfm_join_*) can be stored and joined further later.
int lsfm_write_localmap(const char* path, int mono, const lsfm_map* g)
{
	if (!path || !g) return LSFM_ERR_ARG;
	FILE* f = fopen(path, "w");
	if (!f) return LSFM_ERR_IO;
	std::vector<char> big(1 << 20);
	setvbuf(f, big.data(), _IOFBF, big.size());
	const int r = 6 * g->m + 3 * g->n;
	fprintf(f, "%d\n", g->Ref);
	if (mono) fprintf(f, "%d\n%d\n%d\n", g->ScaP, g->Fix, g->Sign);
	fprintf(f, "%d\n", r);
	for (int i = 0; i < r; i++) fprintf(f, "%d %.17g\n", g->stno[i], g->stVal[i]);
	fprintf(f, "%d\n%d\n%d\n", g->m, g->n, g->nU);
	auto dbl = [&](const double* v, long n) { for (long i = 0; i < n; i++) fprintf(f, i + 1 < n ? "%.17g " : "%.17g", v[i]); fputc('\n', f); };
	auto itg = [&](const int* v, long n) { for (long i = 0; i < n; i++) fprintf(f, i + 1 < n ? "%d " : "%d", v[i]); fputc('\n', f); };
	dbl(g->U, 36L * g->nU); itg(g->Ui, g->nU); itg(g->Uj, g->nU);
	fprintf(f, "%d\n", g->nW);
	dbl(g->W, 18L * g->nW); itg(g->photo, g->nW); itg(g->feature, g->nW);
	dbl(g->V, 9L * g->n);
	if (g->FBlock) itg(g->FBlock, g->n);
	else
	{
		// first W block of every feature, -1 if none (Imp.h:75-121); W is sorted by feature
		std::vector<int> fb(g->n, -1);
		for (int j = g->nW - 1; j >= 0; j--) if (g->feature[j] >= 0 && g->feature[j] < g->n) fb[g->feature[j]] = j;
		itg(fb.data(), g->n);
	}
	// a node of a join tree that is to be joined further (its first frame is not its reference frame, or it carries the origins of its
	// poses): the trailer parse_localmap reads; the reference's reader stops at FBlock
	if (g->FRef != g->Ref || g->pose_origin || (mono && (g->FScaP != g->ScaP || g->FFix != g->Fix)))
	{
		fprintf(f, "%d\n%d %d %d\n%d\n", LSFM_NODE_MAGIC, g->FRef, mono ? g->FScaP : 0, mono ? g->FFix : 0, g->pose_origin ? g->m : 0);
		if (g->pose_origin) itg(g->pose_origin, g->m);
	}
	const bool ok = !ferror(f);
	return (fclose(f) == 0 && ok) ? LSFM_OK : LSFM_ERR_IO;
}

// Imp.cpp:2102-2117
int lsfm_save_state(const char* path, const double* st, const int* stno, int n)
{
	FILE* fp = fopen(path, "w");
	if (!fp) { printf("Please Input Path to Save Final State Vector!"); return LSFM_ERR_IO; }
	for (int i = 0; i < n; i++) fprintf(fp, "%d %lf\n", stno[i], st[i]);
	fclose(fp);
	return LSFM_OK;
}

// Imp.cpp:7876-7967: sorted by id; a repeated id keeps its last occurrence (std::map overwrite)
int lsfm_save_poses(const char* pose_path, const char* feat_path, const int* stno, const double* st, int n)
{
	if (!pose_path && !feat_path) return LSFM_OK;
	std::vector<std::pair<int, int> > P, F;
	for (int i = 0; i < n; i++)
	{
		if (stno[i] <= 0) { P.push_back(std::make_pair(-stno[i], i)); i += 5; }
		else { F.push_back(std::make_pair(stno[i], i)); i += 2; }
	}
	std::stable_sort(P.begin(), P.end());
	std::stable_sort(F.begin(), F.end());
	if (pose_path)
	{
		FILE* fp = fopen(pose_path, "w");
		if (!fp) return LSFM_ERR_IO;
		for (size_t i = 0; i < P.size(); i++)
		{
			if (i + 1 < P.size() && P[i + 1].first == P[i].first) continue;
			const double* p = st + P[i].second;
			fprintf(fp, "%d  %lf  %lf  %lf %lf  %lf  %lf\n", P[i].first, p[0], p[1], p[2], p[3], p[4], p[5]);
		}
		fclose(fp);
	}
	if (feat_path)
	{
		FILE* fp = fopen(feat_path, "w");
		if (!fp) return LSFM_ERR_IO;
		for (size_t i = 0; i < F.size(); i++)
		{
			if (i + 1 < F.size() && F[i + 1].first == F[i].first) continue;
			const double* p = st + F[i].second;
			fprintf(fp, "%d  %lf  %lf %lf\n", F[i].first, p[0], p[1], p[2]);
		}
		fclose(fp);
	}
	return LSFM_OK;
}

} // extern "C"
