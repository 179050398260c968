/* ORACLE -- TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C CPU restatement of the reference's hierarchical linear map-joining path
 * (/root/reference/linux/src/LinearSFMImp/LinearSFMImp.cpp, cited below as Imp.cpp:line).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this; the product
 * library (linearsfm_amd/csrc) never links, includes or calls it.
 *
 * Parity pin status (see DESIGN.md "Oracle"):
 *   - transform (Imp.cpp:349-1924, 3173-6509) and join assembly (Imp.cpp:2551-2965, 7282-7864):
 *     PINNED against the real reference code run in this container (oracle/_ref/ref_dump, fixtures under
 *     tests/golden/ made by tests/golden/make_golden.py).
 *   - Schur + direct solve (Imp.cpp:2119-2378, 6756-7041): the reference calls CHOLMOD, which is absent
 *     from this image and not vendored as source (only CHOLMOD 1.6.0 headers, windows/include/cholmod);
 *     no stand-in was written, so this stage is "parity unpinned" by reference outputs.  It is a
 *     mathematically unique SPD solve; the oracle restates the Schur loops and solves with its own
 *     sparse LL^T (lsfm_chol.c); tests check ||S x - E|| / ||E||.
 */
#ifndef LSFM_ORACLE_H
#define LSFM_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

/* Mirrors LocalMapInfoStereo / LocalMapInfo (Imp.h:75-178).  Mono-only fields are ignored for Stereo. */
typedef struct orc_map {
	int r, Ref, FRef;
	int m, n, nU, nW;
	int ScaP, Fix, Sign, FScaP, FFix;
	int* stno;      /* [6m+3n]                    */
	double* stVal;  /* [6m+3n]                    */
	double* U;      /* [nU*36] row-major 6x6      */
	int *Ui, *Uj;   /* [nU] Ui<=Uj, diag full sym */
	double* W;      /* [nW*18] row-major 6x3      */
	int *photo, *feature; /* [nW], sorted by feature */
	double* V;      /* [n*9]                      */
	int* FBlock;    /* [n] first W index of feature */
} orc_map;

void orc_map_free(orc_map* g);
void orc_map_copy(orc_map* dst, const orc_map* src);

/* Imp.cpp:3044-3132 / 6660-6754.  Returns 0 on success. */
int orc_read_map(const char* path, int mono, orc_map* g);
/* same token order as the readers, doubles at %.17g */
int orc_write_map(const char* path, int mono, const orc_map* g);

/* Imp.cpp:349-1924.  out is freshly allocated. */
void orc_transform_stereo(const orc_map* in, int Ref, orc_map* out);
/* Imp.cpp:3173-6509 */
void orc_transform_mono(const orc_map* in, int Ref, int ScaP, int Fix, orc_map* out);

/* Imp.cpp:2551-2965 (assembly part of lmj_LinearLS_PF3DStereo): builds joint arrays (joint->stVal is
 * allocated but not filled) and the right-hand sides eP[6m], eF[3n] (malloc'ed, caller frees). */
void orc_join_assemble_stereo(const orc_map* End, const orc_map* Cur, orc_map* joint, double** eP, double** eF);
/* Imp.cpp:7282-7864.  NOTE: like the reference it unwraps the scale-pose angles of End and Cur IN PLACE
 * (Imp.cpp:7427-7465).  solve_args[5] = {Ref(posID1), ScaP(pos1), Fix(posFix), Sign, FixBlk} (Imp.cpp:7860-7864) */
void orc_join_assemble_mono(orc_map* End, orc_map* Cur, orc_map* joint, double** eP, double** eF, int solve_args[5]);

/* Imp.cpp:2119-2378: Schur complement on features, solve, back-substitute.  Writes stVal[0..6m+3n).
 * V is left unchanged.  Returns 0, or >0 if the reduced system is not positive definite.
 * stats (optional, may be NULL): [0]=nuis (upper blocks of S), [1]=nnz(L) */
int orc_solve_stereo(double* stVal, const double* eb, const double* ea, const double* U, const double* W,
                     const double* V, const int* Ui, const int* Uj, const int* photo, const int* feature,
                     int m, int n, int nU, int nW, long* stats);
/* Imp.cpp:6756-7041 */
int orc_solve_mono(double* stVal, const double* eb, const double* ea, const double* U, const double* W,
                   const double* V, const int* Ui, const int* Uj, const int* photo, const int* feature,
                   int m, int n, int nU, int nW, int Ref, int ScaP, int Fix, int Sign, int FixBlk, long* stats);

/* Schur system only (for tests / PCG experiments): S as block-CRS upper (rowptr[m+1], colidx[nuis], val[nuis*36];
 * diagonal blocks hold their upper triangle only, as in the reference) and E[6m].  Arrays malloc'ed. */
void orc_schur(const double* eb, const double* ea, const double* U, const double* W, const double* V,
               const int* Ui, const int* Uj, const int* photo, const int* feature, int m, int n, int nU, int nW,
               int accumulate_u, int** rowptr, int** colidx, double** Sval, double** E, double** Vinv);

/* pba_solveFeatures (Imp.cpp:2980-3020) on its own: dpb[3n] from given pose values dpa[6m] and V^-1 (IV) */
void orc_solve_features(const double* W, const double* IV, const double* eb, const double* dpa, double* dpb, int n,
                        const int* photo, const int* feature, int nW);

/* the scalar CSC (upper, stype = 1) the reference hands to CHOLMOD, values in double: what pba_constructCSSLM / GN
 * (Imp.cpp:2451-2498 / 7123-7200) write from S.  skipblk / skipfix < 0: Stereo; Mono: block `Ref` and scalar `Fix` are
 * left out and the rest renumbered.  Arrays malloc'ed; returns the dimension. */
int orc_schur_csc(const double* S, const int* rowptr, const int* colidx, int m, int skipblk, int skipfix, int** Sp, int** Si,
                  double** Sx);

/* on = 1: every Schur complement, LL^T and back-substitution of the calls below runs in long double (64-bit mantissa:
 * oracle/lsfm_solve_num.inc, lsfm_chol_num.inc -- the same statements compiled for a second type), inputs and outputs
 * stay fp64.  Yardstick only: it says which of two fp64 answers is nearer the exact solution of the assembled system. */
void orc_set_extended(int on);

/* full joins: transform is NOT included.  End and Cur are consumed (freed), joint is produced. */
int orc_join_stereo(orc_map* End, orc_map* Cur, orc_map* joint);
int orc_join_mono(orc_map* End, orc_map* Cur, orc_map* joint);

/* Imp.cpp:1926-2063 / 6511-6630: binary-tree divide and conquer.  maps[0..N) are consumed; result in *out.
 * match_hash != 0 replaces the O(n1*n2) std::find feature matching by a sort-based one (same result).
 * timing (optional): [0]=total s, [1]=transform s, [2]=join-assembly s, [3]=solve s */
int orc_divide_conquer(orc_map* maps, int N, int mono, orc_map* out, int verbose, double* timing);
/* the same tree, the independent joins of a level on `nthreads` host threads (OpenMP); identical result; timing[0] only */
int orc_divide_conquer_omp(orc_map* maps, int N, int mono, orc_map* out, int nthreads, double* timing);

/* Imp.cpp:2102-2117 and 7876-7967 (byte-compatible "%lf" files) */
int orc_save_state(const char* path, const double* st, const int* stno, int n);
int orc_save_poses(const char* pose_path, const char* feat_path, const int* stno, const double* st, int n);

/* sparse SPD solve used by the solves above: upper-triangular CSC (stype=1 like the reference hands to CHOLMOD),
 * block_perm (length nb) may be NULL.  Returns 0 or the failing column+1. */
int orc_chol_solve(int n, const int* Ap, const int* Ai, const double* Ax, const int* perm, const double* b,
                   double* x, long* lnz_out);
int orc_chol_solve_x(int n, const int* Ap, const int* Ai, const long double* Ax, const int* perm, const long double* b,
                     long double* x, long* lnz_out);
/* minimum-degree ordering on a symmetric block pattern given as upper CSC (Ap[nb+1], Ai) */
void orc_min_degree(int nb, const int* Ap, const int* Ai, int* perm);

/* Gauss-Newton polish of the map-joining objective F(x) = sum_k ||x^_k - f_k(x)||^2_{I_k} over the N local maps (lsfm_gn.inc; the
 * reference has no such step: PARITY UNPINNED, property-checked).  G: the global state -- stno / stVal / m / n, Ref (its frame), Mono
 * also ScaP / Fix (gauge) -- e.g. the result of orc_divide_conquer; stVal is updated in place.  obj / gnorm: [iters + 1]. */
int orc_gn_polish(const orc_map* maps, int N, int mono, orc_map* G, int iters, double* obj, double* gnorm, int* halvings);
/* F and b = sum_k J_k^T I_k r_k (= -1/2 grad F) at G's state; grad[6m + 3n] may be NULL */
int orc_gn_objective(const orc_map* maps, int N, int mono, const orc_map* G, double* F, double* grad);
/* y = H v, H = sum_k J_k^T I_k J_k the matrix of a step at G's state (test entry) */
int orc_gn_hessian_times(const orc_map* maps, int N, int mono, const orc_map* G, const double* v, double* y);

void orc_set_match_hash(int on);
/* on = 0: orc_divide_conquer leaves the final map in the frame of its last join (a subtree root, Imp.cpp:2032),
 * instead of taking it back to its first frame (Imp.cpp:2039-2063) */
void orc_set_final_reanchor(int on);
/* Feature-sharded evaluation (checker of the multi-GPU top levels): this process holds slice `rank` of `world` of the features
 * of every map; fn sums `count` 8-byte elements at buf over the processes, in place (dtype 0: double, 1: 64-bit integer).
 * world <= 1 or fn == NULL: off. */
typedef void (*orc_reduce_fn)(void* buf, long count, int dtype);
void orc_set_comm(int rank, int world, orc_reduce_fn fn);

#ifdef __cplusplus
}
#endif
#endif
