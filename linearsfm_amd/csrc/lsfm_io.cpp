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

// ---- binary cache of a set of local maps --------------------------------------------------------------------
// One file for N maps: what the text reader produced, array by array, so that a second run over the same set skips the decimal
// conversion.  Layout (little endian, as the host): 32-byte file header { magic "LSFMSET1", int32 version, N, mono, 0 0 0 },
// N + 1 uint64 offsets of the maps' records (the last one = file size), then per map a 64-byte header of 16 int32 { Ref FRef ScaP Fix
// Sign FScaP FFix m n nU nW has_origin, 0 0 0 0 } followed by the int32 arrays stno[6m+3n] Ui[nU] Uj[nU] photo[nW] feature[nW]
// FBlock[n] pose_origin[m if has_origin], padding to 8 bytes, and the doubles stVal[6m+3n] U[36 nU] W[18 nW] V[9 n].
// Semantics identical to the text files: reading a cache gives the arrays lsfm_read_localmaps gives, bit for bit.
namespace {
const char SET_MAGIC[8] = { 'L', 'S', 'F', 'M', 'S', 'E', 'T', '1' };
struct SetHeader { char magic[8]; int version, N, mono, pad[3]; };
static_assert(sizeof(SetHeader) == 32, "file header");
size_t set_record_bytes(const lsfm_map* g)
{
	const size_t r = 6 * (size_t)g->m + 3 * (size_t)g->n;
	size_t ints = r + 2 * (size_t)g->nU + 2 * (size_t)g->nW + (size_t)g->n + (g->pose_origin ? (size_t)g->m : 0);
	ints += ints & 1;
	return 64 + ints * 4 + (r + 36 * (size_t)g->nU + 18 * (size_t)g->nW + 9 * (size_t)g->n) * 8;
}
}

int lsfm_write_mapset(const char* path, const lsfm_map* maps, int N, int mono)
{
	if (!path || !maps || N < 0) return LSFM_ERR_ARG;
	const std::string tmp = std::string(path) + ".tmp"; // a reader never sees a half-written cache
	FILE* f = fopen(tmp.c_str(), "wb");
	if (!f) return LSFM_ERR_IO;
	std::vector<char> big(4 << 20);
	setvbuf(f, big.data(), _IOFBF, big.size());
	SetHeader h;
	memset(&h, 0, sizeof h);
	memcpy(h.magic, SET_MAGIC, 8); h.version = 1; h.N = N; h.mono = mono ? 1 : 0;
	std::vector<unsigned long long> off((size_t)N + 1);
	off[0] = sizeof h + ((size_t)N + 1) * 8;
	for (int k = 0; k < N; k++) off[k + 1] = off[k] + set_record_bytes(&maps[k]);
	bool ok = fwrite(&h, sizeof h, 1, f) == 1 && fwrite(off.data(), 8, off.size(), f) == off.size();
	for (int k = 0; k < N && ok; k++)
	{
		const lsfm_map* g = &maps[k];
		const size_t r = 6 * (size_t)g->m + 3 * (size_t)g->n;
		const int hd[16] = { g->Ref, g->FRef, g->ScaP, g->Fix, g->Sign, g->FScaP, g->FFix, g->m, g->n, g->nU, g->nW, g->pose_origin ? 1 : 0, 0, 0, 0, 0 };
		ok = fwrite(hd, 4, 16, f) == 16;
		size_t ints = 0;
		auto wi = [&](const int* v, size_t n) { if (ok && n) ok = fwrite(v, 4, n, f) == n; ints += n; };
		auto wd = [&](const double* v, size_t n) { if (ok && n) ok = fwrite(v, 8, n, f) == n; };
		wi(g->stno, r); wi(g->Ui, g->nU); wi(g->Uj, g->nU); wi(g->photo, g->nW); wi(g->feature, g->nW);
		if (g->FBlock) wi(g->FBlock, g->n);
		else
		{
			std::vector<int> fb(g->n, -1);
			for (int j = g->nW - 1; j >= 0; j--) if (g->feature[j] >= 0 && g->feature[j] < g->n) fb[g->feature[j]] = j;
			wi(fb.data(), g->n);
		}
		if (g->pose_origin) wi(g->pose_origin, g->m);
		if (ints & 1) { const int z = 0; wi(&z, 1); }
		wd(g->stVal, r); wd(g->U, 36 * (size_t)g->nU); wd(g->W, 18 * (size_t)g->nW); wd(g->V, 9 * (size_t)g->n);
	}
	ok = ok && !ferror(f);
	ok = (fclose(f) == 0) && ok;
	if (!ok || rename(tmp.c_str(), path) != 0) { remove(tmp.c_str()); return LSFM_ERR_IO; }
	return LSFM_OK;
}

int lsfm_mapset_info(const char* path, int* N, int* mono)
{
	if (!path) return LSFM_ERR_ARG;
	FILE* f = fopen(path, "rb");
	if (!f) return LSFM_ERR_IO;
	SetHeader h;
	const bool ok = fread(&h, sizeof h, 1, f) == 1 && memcmp(h.magic, SET_MAGIC, 8) == 0 && h.version == 1 && h.N >= 0;
	fclose(f);
	if (!ok) return LSFM_ERR_IO;
	if (N) *N = h.N;
	if (mono) *mono = h.mono;
	return LSFM_OK;
}

// A cache says nothing of the text files it was made from unless its writer leaves a stamp (include/lsfm.h): 8 bytes of the header's padding
int lsfm_mapset_stamp(const char* path, unsigned long long* stamp, const unsigned long long* set_to)
{
	if (!path) return LSFM_ERR_ARG;
	FILE* f = fopen(path, set_to ? "r+b" : "rb");
	if (!f) return LSFM_ERR_IO;
	SetHeader h;
	bool ok = fread(&h, sizeof h, 1, f) == 1 && memcmp(h.magic, SET_MAGIC, 8) == 0 && h.version == 1 && h.N >= 0;
	if (ok && set_to)
	{
		memcpy(&h.pad[0], set_to, 8);
		ok = fseek(f, 0, SEEK_SET) == 0 && fwrite(&h, sizeof h, 1, f) == 1;
	}
	fclose(f);
	if (!ok) return LSFM_ERR_IO;
	if (stamp) memcpy(stamp, &h.pad[0], 8);
	return LSFM_OK;
}

// maps first .. first+count-1 (0-based) of a cache on `threads` host threads; everything is checked against the file size before
// a byte is copied (a truncated or foreign file is LSFM_ERR_IO, never a wild read)
int lsfm_read_mapset(const char* path, int mono, int first, int count, int threads, lsfm_map* out)
{
	if (!path || !out || first < 0 || count < 0) return LSFM_ERR_ARG;
	for (int k = 0; k < count; k++) memset(&out[k], 0, sizeof(lsfm_map));
	FILE* f = fopen(path, "rb");
	if (!f) return LSFM_ERR_IO;
	SetHeader h;
	std::vector<unsigned long long> off;
	bool ok = fread(&h, sizeof h, 1, f) == 1 && memcmp(h.magic, SET_MAGIC, 8) == 0 && h.version == 1 && h.N >= 0 && (h.mono != 0) == (mono != 0) &&
	          (long long)first + count <= h.N;
	long fsize = 0;
	if (ok)
	{
		off.resize((size_t)h.N + 1);
		ok = fread(off.data(), 8, off.size(), f) == off.size() && fseek(f, 0, SEEK_END) == 0 && (fsize = ftell(f)) > 0 && off[h.N] == (unsigned long long)fsize;
		for (int k = 0; k < h.N && ok; k++) ok = off[k] + 64 <= off[k + 1];
	}
	fclose(f);
	if (!ok) return LSFM_ERR_IO;
	if (threads <= 0) threads = (int)std::min(32u, std::max(1u, std::thread::hardware_concurrency()));
	threads = std::max(1, std::min(threads, count));
	std::atomic<int> next(0), bad(0);
	auto work = [&]() {
		FILE* fh = fopen(path, "rb");
		if (!fh) { bad = 1; return; }
		for (;;)
		{
			const int k = next.fetch_add(1);
			if (k >= count) break;
			lsfm_map* g = &out[k];
			const unsigned long long o = off[first + k], len = off[first + k + 1] - o;
			int hd[16];
			if (fseek(fh, (long)o, SEEK_SET) != 0 || fread(hd, 4, 16, fh) != 16) { bad = 1; continue; }
			g->Ref = hd[0]; g->FRef = hd[1]; g->ScaP = hd[2]; g->Fix = hd[3]; g->Sign = hd[4]; g->FScaP = hd[5]; g->FFix = hd[6];
			g->m = hd[7]; g->n = hd[8]; g->nU = hd[9]; g->nW = hd[10];
			const bool org = hd[11] != 0;
			if (g->m < 0 || g->n < 0 || g->nU < 0 || g->nW < 0) { memset(g, 0, sizeof *g); bad = 1; continue; }
			lsfm_map probe = *g;
			int one = 0;
			probe.pose_origin = org ? &one : nullptr;
			if (set_record_bytes(&probe) != len) { memset(g, 0, sizeof *g); bad = 1; continue; }
			const size_t r = 6 * (size_t)g->m + 3 * (size_t)g->n;
			size_t ints = 0;
			bool good = true;
			auto ri = [&](int*& v, size_t n) { v = xalloc<int>(n); good = good && v && (n == 0 || fread(v, 4, n, fh) == n); ints += n; };
			auto rd = [&](double*& v, size_t n) { v = xalloc<double>(n); good = good && v && (n == 0 || fread(v, 8, n, fh) == n); };
			ri(g->stno, r); ri(g->Ui, g->nU); ri(g->Uj, g->nU); ri(g->photo, g->nW); ri(g->feature, g->nW); ri(g->FBlock, g->n);
			if (org) ri(g->pose_origin, g->m);
			if (ints & 1) { int z; good = good && fread(&z, 4, 1, fh) == 1; }
			rd(g->stVal, r); rd(g->U, 36 * (size_t)g->nU); rd(g->W, 18 * (size_t)g->nW); rd(g->V, 9 * (size_t)g->n);
			if (!good) bad = 1;
		}
		fclose(fh);
	};
	std::vector<std::thread> pool;
	for (int t = 1; t < threads; t++) pool.emplace_back(work);
	work();
	for (auto& th : pool) th.join();
	if (bad.load())
	{
		for (int k = 0; k < count; k++) lsfm_map_release(&out[k]);
		return LSFM_ERR_IO;
	}
	return LSFM_OK;
}

// the state vector as raw doubles behind its labels (a dump for parity tooling that loses nothing and needs no parser):
// int32 n, int32 0, stno[n] int32 (+ one int32 of padding when n is odd), st[n] float64
int lsfm_save_state_bin(const char* path, const double* st, const int* stno, int n)
{
	if (!path || !st || !stno || n < 0) return LSFM_ERR_ARG;
	FILE* f = fopen(path, "wb");
	if (!f) return LSFM_ERR_IO;
	const int hd[2] = { n, 0 }, z = 0;
	bool ok = fwrite(hd, 4, 2, f) == 2 && (n == 0 || fwrite(stno, 4, (size_t)n, f) == (size_t)n);
	if (ok && (n & 1)) ok = fwrite(&z, 4, 1, f) == 1;
	ok = ok && (n == 0 || fwrite(st, 8, (size_t)n, f) == (size_t)n) && !ferror(f);
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
