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

extern "C" {

int lsfm_read_localmap(const char* path, int mono, lsfm_map* g)
{
	if (!path || !g) return LSFM_ERR_ARG;
	FILE* f = fopen(path, "r");
	if (!f) return LSFM_ERR_IO;
	memset(g, 0, sizeof *g);
	bool ok = true;
	int r = 0;
	ok &= fscanf(f, "%d", &g->Ref) == 1;
	g->FRef = g->Ref;
	g->Sign = 1;
	if (mono)
	{
		ok &= fscanf(f, "%d", &g->ScaP) == 1; g->FScaP = g->ScaP;
		ok &= fscanf(f, "%d", &g->Fix) == 1; g->FFix = g->Fix;
		ok &= fscanf(f, "%d", &g->Sign) == 1;
	}
	ok &= fscanf(f, "%d", &r) == 1;
	if (!ok || r < 0) { fclose(f); return LSFM_ERR_IO; }
	g->stno = xalloc<int>(r); g->stVal = xalloc<double>(r);
	for (int i = 0; i < r && ok; i++) ok &= fscanf(f, "%d %lf", &g->stno[i], &g->stVal[i]) == 2;
	ok &= fscanf(f, "%d", &g->m) == 1;
	ok &= fscanf(f, "%d", &g->n) == 1;
	ok &= fscanf(f, "%d", &g->nU) == 1;
	if (!ok || g->nU < 0 || 6 * g->m + 3 * g->n != r) { fclose(f); lsfm_map_release(g); return LSFM_ERR_IO; }
	g->U = xalloc<double>((size_t)g->nU * 36); g->Ui = xalloc<int>(g->nU); g->Uj = xalloc<int>(g->nU);
	for (long i = 0; i < 36L * g->nU && ok; i++) ok &= fscanf(f, "%lf", &g->U[i]) == 1;
	for (int i = 0; i < g->nU && ok; i++) ok &= fscanf(f, "%d", &g->Ui[i]) == 1;
	for (int i = 0; i < g->nU && ok; i++) ok &= fscanf(f, "%d", &g->Uj[i]) == 1;
	ok &= fscanf(f, "%d", &g->nW) == 1;
	if (!ok || g->nW < 0) { fclose(f); lsfm_map_release(g); return LSFM_ERR_IO; }
	g->W = xalloc<double>((size_t)g->nW * 18); g->photo = xalloc<int>(g->nW); g->feature = xalloc<int>(g->nW);
	for (long i = 0; i < 18L * g->nW && ok; i++) ok &= fscanf(f, "%lf", &g->W[i]) == 1;
	for (int i = 0; i < g->nW && ok; i++) ok &= fscanf(f, "%d", &g->photo[i]) == 1;
	for (int i = 0; i < g->nW && ok; i++) ok &= fscanf(f, "%d", &g->feature[i]) == 1;
	g->V = xalloc<double>((size_t)g->n * 9); g->FBlock = xalloc<int>(g->n);
	g->pose_origin = NULL;
	for (long i = 0; i < 9L * g->n && ok; i++) ok &= fscanf(f, "%lf", &g->V[i]) == 1;
	for (int i = 0; i < g->n && ok; i++) ok &= fscanf(f, "%d", &g->FBlock[i]) == 1;
	fclose(f);
	if (!ok) { lsfm_map_release(g); return LSFM_ERR_IO; }
	return LSFM_OK;
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
