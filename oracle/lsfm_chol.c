/* ORACLE -- TEST INFRASTRUCTURE ONLY.
 *
 * Sparse SPD direct solve standing where the reference calls CHOLMOD (cholmod_analyze[_p] / cholmod_factorize /
 * cholmod_solve(CHOLMOD_A), Imp.cpp:2380-2449, 7043-7121).  CHOLMOD itself (SuiteSparse; Windows tree pins
 * 1.6.0 headers only, no source) is not available in this image, so this is NOT the reference's arithmetic:
 * it is an up-looking sparse LL^T (the textbook algorithm: elimination tree, row pattern by tree reach,
 * one sparse triangular solve per row) on the same upper-triangular CSC input with a fill-reducing block
 * ordering.  The solution of an SPD system is unique, so only rounding differs.
 */
#include "lsfm_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static void* xm(size_t n) { void* p = malloc(n ? n : 1); if (!p) { fprintf(stderr, "oracle: out of memory\n"); exit(1); } return p; }

/* pattern of row k of L: nodes reachable in the elimination tree from the entries of column k of C (upper) */
static int ereach(const int* Cp, const int* Ci, int k, const int* parent, int* s, int* w, int n)
{
	int top = n, p, i, len;
	w[k] = k;
	for (p = Cp[k]; p < Cp[k + 1]; p++)
	{
		i = Ci[p];
		if (i > k) continue;
		for (len = 0; w[i] != k; i = parent[i]) { s[len++] = i; w[i] = k; }
		while (len > 0) s[--top] = s[--len];
	}
	return top;
}

static long symbolic_lnz(int n, const int* Cp, const int* Ci, int* parent, int* colcount)
{
	int *anc = xm(n * sizeof(int)), *s = xm(n * sizeof(int)), *w = xm(n * sizeof(int));
	int k, p, i, inext, top;
	long lnz = 0;
	for (k = 0; k < n; k++)
	{
		parent[k] = -1; anc[k] = -1;
		for (p = Cp[k]; p < Cp[k + 1]; p++)
		{
			i = Ci[p];
			while (i != -1 && i < k) { inext = anc[i]; anc[i] = k; if (inext == -1) parent[i] = k; i = inext; }
		}
	}
	for (k = 0; k < n; k++) { w[k] = -1; colcount[k] = 1; }
	for (k = 0; k < n; k++)
	{
		top = ereach(Cp, Ci, k, parent, s, w, n);
		for (; top < n; top++) colcount[s[top]]++;
	}
	for (k = 0; k < n; k++) lnz += colcount[k];
	free(anc); free(s); free(w);
	return lnz;
}

#define REAL double
#define RSQRT sqrt
#define SUF(name) name##_d
#include "lsfm_chol_num.inc"
#undef REAL
#undef RSQRT
#undef SUF
#define REAL long double
#define RSQRT sqrtl
#define SUF(name) name##_x
#include "lsfm_chol_num.inc"
#undef REAL
#undef RSQRT
#undef SUF

int orc_chol_solve(int n, const int* Ap, const int* Ai, const double* Ax, const int* perm, const double* b, double* x, long* lnz_out)
{
	return chol_solve_d(n, Ap, Ai, Ax, perm, b, x, lnz_out);
}
int orc_chol_solve_x(int n, const int* Ap, const int* Ai, const long double* Ax, const int* perm, const long double* b, long double* x, long* lnz_out)
{
	return chol_solve_x(n, Ap, Ai, Ax, perm, b, x, lnz_out);
}

/* Fill-reducing ordering of the block graph.  The Schur matrices of this path are "chain + hubs": every
 * transform makes one pose adjacent to its whole sub-map.  Eliminating by ascending initial degree (ties in
 * natural order) keeps the chain banded and pushes the hubs -- the separators of the join tree -- to the end.
 * It is compared with the natural order by symbolic fill and the better one is returned. */
typedef struct { int deg, idx; } degidx;
static int cmp_degidx(const void* a, const void* b)
{
	const degidx* x = a; const degidx* y = b;
	if (x->deg != y->deg) return x->deg < y->deg ? -1 : 1;
	return x->idx < y->idx ? -1 : (x->idx > y->idx);
}

static int bitlen(unsigned x) { int l = 0; while (x) { l++; x >>= 1; } return l; }

/* Three candidate orderings are compared by symbolic fill and the best is returned:
 *  natural, ascending degree, and a nested dissection along the join tree: block index p is the position of the
 *  pose in the joint state = (Stereo) the index of the local map that brought it, so the binary join tree is the
 *  bit structure of p.  An edge (p,q) crosses the cut of tree level bitlen(p^q); the endpoint of higher degree
 *  (the hub pose of that sub-map) is put into that level's separator (a vertex cover of the crossing edges);
 *  blocks are eliminated by ascending separator level. */
void orc_min_degree(int nb, const int* Ap, const int* Ai, int* perm)
{
	degidx* d = xm((nb + 1) * sizeof *d);
	int *pinv = xm((nb + 1) * sizeof(int)), *Cp, *Ci, *parent = xm((nb + 1) * sizeof(int)), *cc = xm((nb + 1) * sizeof(int));
	int *deg = calloc(nb + 1, sizeof(int)), *best = xm((nb + 1) * sizeof(int));
	double* Cx;
	int j, p, cand;
	long fill, best_fill = -1;
	for (j = 0; j < nb; j++)
		for (p = Ap[j]; p < Ap[j + 1]; p++)
			if (Ai[p] != j) { deg[j]++; deg[Ai[p]]++; }
	for (cand = 0; cand < 3; cand++)
	{
		for (j = 0; j < nb; j++) { d[j].deg = 0; d[j].idx = j; }
		if (cand == 1) for (j = 0; j < nb; j++) d[j].deg = deg[j];
		if (cand == 2)
			for (j = 0; j < nb; j++)
				for (p = Ap[j]; p < Ap[j + 1]; p++)
				{
					int i = Ai[p], l, v;
					if (i == j) continue;
					l = bitlen((unsigned)(i ^ j));
					v = (deg[i] > deg[j] || (deg[i] == deg[j] && i > j)) ? i : j;
					if (l > d[v].deg) d[v].deg = l;
				}
		qsort(d, nb, sizeof *d, cmp_degidx);
		for (j = 0; j < nb; j++) { perm[j] = d[j].idx; pinv[d[j].idx] = j; }
		symperm_d(nb, Ap, Ai, NULL, pinv, &Cp, &Ci, &Cx);
		fill = symbolic_lnz(nb, Cp, Ci, parent, cc);
		free(Cp); free(Ci);
		if (best_fill < 0 || fill < best_fill) { best_fill = fill; memcpy(best, perm, nb * sizeof(int)); }
		/* test hook: ORC_ORDER=0|1|2 forces one candidate (used to show how far two valid fp64 solves of the same
		 * system differ on ill-conditioned joins) */
		if (getenv("ORC_ORDER") && atoi(getenv("ORC_ORDER")) == cand) { memcpy(best, perm, nb * sizeof(int)); break; }
	}
	memcpy(perm, best, nb * sizeof(int));
	free(d); free(pinv); free(parent); free(cc); free(deg); free(best);
}
