/* lsfm.h -- C ABI of the MI355X-native LinearSFM hot path (liblsfm_hip.so).
 *
 * Every entry point replaces one public method of the reference's CLinearSFMImp
 * (/root/reference/linux/src/LinearSFMImp/LinearSFMImp.h, "Imp.h"; bodies in LinearSFMImp.cpp, "Imp.cpp").
 * Plain pointers and sizes only; all arrays are HOST arrays in the reference's own layout unless the name
 * says "dev".  Every function returns an int status: 0 ok, <0 invalid input / runtime failure, >0 numerical
 * (PCG not converged / not positive definite).  The reference's methods are void and ignore CHOLMOD's
 * status (Imp.cpp:2356, 7006).  There is NO CPU fallback: without a usable HIP device every call fails with
 * LSFM_ERR_NO_DEVICE.
 *
 * Map layout (= LocalMapInfoStereo / LocalMapInfo, Imp.h:75-178):
 *   stno[6m+3n]   labels: <=0 pose id (-stno, x6), >0 feature id (x3)
 *   stVal[6m+3n]  poses (tx ty tz alpha beta gamma) then features (x y z)
 *   U[nU*36]      6x6 row-major pose-pose information blocks, (Ui,Uj) block coordinates, Ui<=Uj, diagonal
 *                 blocks stored full; several entries with the same coordinates add up
 *   W[nW*18]      6x3 row-major pose-feature blocks, (photo, feature), sorted by feature, >=1 per feature
 *   V[n*9]        3x3 feature blocks;  FBlock[n] first W index of each feature
 *   Mono only:    ScaP (id of the scale pose), Fix (0..2 fixed translation axis of ScaP), Sign (+-1),
 *                 FScaP / FFix (those of the map's first frame)
 */
#ifndef LSFM_H
#define LSFM_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LSFM_OK 0
#define LSFM_ERR_ARG (-1)
#define LSFM_ERR_NO_DEVICE (-2)
#define LSFM_ERR_HIP (-3)
#define LSFM_ERR_OOM (-4)
#define LSFM_ERR_INTERNAL (-5)
#define LSFM_ERR_IO (-6)
#define LSFM_ERR_NOT_SPD (-7) /* a camera system whose factorisation met a pivot far below zero: the information matrices are not positive definite */
#define LSFM_NOT_CONVERGED 1

typedef struct lsfm_context lsfm_context; /* one per GPU / stream; thread-compatible, not thread-safe */

/* mirrors LocalMapInfoStereo / LocalMapInfo (Imp.h:75-178); arrays owned by whoever filled the struct */
typedef struct lsfm_map {
	int Ref, FRef, m, n, nU, nW;
	int ScaP, Fix, Sign, FScaP, FFix; /* Mono only */
	int* stno;
	double* stVal;
	double* U;
	int *Ui, *Uj;
	double* W;
	int *photo, *feature;
	double* V;
	int* FBlock;
	/* optional extension (NULL = absent): per pose, the index of the LOCAL MAP that brought it into the tree.  Only
	 * used to order the Cholesky preconditioner along the join tree; filled on outputs, honoured on inputs, so that a
	 * map produced by one lsfm_tree_run can seed another (multi-GPU subtree sharding). */
	int* pose_origin;
} lsfm_map;

typedef struct lsfm_stats {
	/* whole run */
	double t_total_ms;      /* region the reference times: Imp.cpp:1929 -> 2068 (all transforms + joins) */
	double t_transform_ms, t_join_ms, t_schur_ms, t_pcg_ms, t_backsub_ms;
	long pcg_iterations;    /* sum over levels of the iterations run (batched systems iterate together) */
	long spmv_launches;
	double spmv_ms;         /* HIP-event time of all SpMV launches, measured on the context's stream */
	double spmv_bytes;      /* algorithmic bytes of all SpMV launches ("K10a") */
	long spmv_nnzb_upper_last, spmv_rows_last; /* upper blocks / block rows of the last (top) system */
	double max_rel_residual;/* max over systems of ||E - S x|| / ||E|| at exit */
	int levels, joins, transforms;
	int not_converged;      /* number of systems that hit the iteration cap */
	/* the two instrumented kernels, each bracketed by HIP events on the context's stream (one bracket per tree level):
	 * K9 k_schur_panel (Schur assembly, all panel variants of a level + k_schur_w for the tiles none of them takes) and
	 * K3/K4 k_tr_entries (information transform, one lane per W block).  bytes = algorithmic bytes of those launches
	 *  */
	long schur_launches, trf_launches;
	double schur_ms, schur_bytes, trf_ms, trf_bytes;
	/* algorithmic flops of the K9 launches: per feature with k W blocks, k (108 + 36) for W V^-1 and the right-hand side and
	 * k (k + 1) / 2 * 216 for the pose pairs (Imp.cpp:2260-2328) -- at the top levels K9 is bound by the fp64 matrix rate */
	double schur_flops;
	/* wall ms lsfm_tree_upload took to bring the tree's N local maps into HBM (PCIe; never part of t_total_ms) */
	double upload_ms;
	/* how many times the tree was joined by this call: 1, or more when a run was repeated (a plan or step count of an earlier run
	 * that did not fit the values, a system left above its bound by an unlucky rounding of its factorisation) */
	int attempts;
	/* LSFM_FACTOR_DIGEST=1 in the environment (a debug / test aid, off by default: two more passes over every factor): sums
	 * modulo 2^64 of the mixed bit patterns of every camera system as assembled (s_digest) and of every Cholesky factor the run
	 * computed (factor_digest), over all levels.  The factorisation accumulates its updates in fixed point (integer atomics), so two
	 * runs with equal s_digest have equal factor_digest, whatever order the work-groups ran in. */
	unsigned long long s_digest, factor_digest;
	/* feature-sharded runs: tree levels whose camera systems were factored distributed over the ranks (lsfm_tree_set_comm_blocks) */
	int dist_solves;
	/* ... and, over those levels, the block products (6x6x6 multiply-adds) of the numeric factorisations: all of them / those of the
	 * shared separator columns, which every rank repeats (the replicated share: an Amdahl bound of the distributed solve) */
	double dist_work_total, dist_work_shared;
	/* LSFM_FACTOR_DIGEST=1: every camera system of the run is factored TWICE (the work-groups of the two factorisations are scheduled
	 * differently, their atomics land in another order); the number of systems whose two factors were not the same bits.  Must be 0:
	 * the accumulation is in fixed point. */
	int refactor_mismatch;
	/* LSFM_FACTOR_DIGEST=1: the camera systems of every level (S and the right-hand side E) are ASSEMBLED twice from the same joint
	 * maps (K9's work-groups land their sums in another order); the number of levels whose two assemblies were not the same bits.
	 * Must be 0 since round 5: K9 adds in fixed point (include: the per-feature fallback kernel). */
	int s_rebuild_mismatch;
	/* tree levels (or stage-level calls) whose camera systems -- at most 16 poses each -- were assembled, factored and solved by the
	 * one-launch dense path (one work-group per join, the system in LDS: lsfm_small.hip), and the HIP-event time of those launches */
	int small_levels;
	double t_small_ms;
} lsfm_stats;

/* ---- context ------------------------------------------------------------------------------------ */
/* device: HIP device ordinal.  arena_bytes: device memory reserved for maps and work space (0 = size on
 * demand from the inputs of the first call).  Fails with LSFM_ERR_NO_DEVICE when no GPU is usable. */
int lsfm_context_create(int device, size_t arena_bytes, lsfm_context** out);
void lsfm_context_destroy(lsfm_context* ctx);
/* Solver controls.  rel_tol: the refinement of a system stops when ||E - S x|| <= rel_tol * ||E|| (default 1e-12: the
 * residual level of a direct fp64 solve, which is what the reference computes) or when the true residual stops
 * shrinking.  max_steps: most refinement steps a system may take, 1 .. 50 (<= 0: the default, 50); a system that is
 * still above 1e-9 relative when the cap is reached counts as not converged. */
int lsfm_set_pcg(lsfm_context* ctx, double rel_tol, int max_steps);
/* Precision of the preconditioner.  mode 0 (default): fp64 throughout, the reference's arithmetic.  mode 1, "mixed": the
 * Cholesky factor of every camera system is kept and applied in fp32 (half the bytes of the triangular solves; the
 * factorisation itself runs in fp64 and is rounded once -- an fp32 elimination breaks down on these matrices), while S,
 * the right-hand side, the iterate and the residual r = E - S x stay fp64: every refinement step corrects against the
 * fp64 residual, and the stopping rule is the same relative residual as in mode 0, reached in more steps (2-3 instead
 * of 1).  BASELINE.json configs[4]. */
int lsfm_set_precision(lsfm_context* ctx, int mode);
/* Camera systems of a few poses (the lowest levels of a tree: thousands of independent joins a level) are assembled, factored (dense
 * Cholesky in LDS, one refinement step) and solved, features included, by ONE launch with one work-group per join instead of the level
 * pipeline's ~55 (Imp.cpp:2119-2378, run per join by the reference).  max_poses: the largest system that takes this path -- 0: none,
 * every level takes the sparse pipeline; at most 16 (what the kernel holds); default 5 (Stereo: the two lowest levels; where the path
 * beats the pipeline).  The tests compare the two paths at every size.  Feature-sharded runs and the fp32 preconditioner
 * always take the pipeline. */
int lsfm_set_small_solve(lsfm_context* ctx, int max_poses);
/* Which kernel multiplies by the Schur matrix in the CG.  0 (default): by size -- a matrix that stays in L2 / Infinity
 * Cache (every level of the named configurations) is multiplied from a row-sorted list of both orientations of its
 * blocks, a larger one streams its upper blocks from HBM once.  1: always the streaming kernel.  2: always the list.
 * A measurement / test knob; it applies to the systems analysed after the call (a recorded plan keeps its choice). */
int lsfm_set_spmv_variant(lsfm_context* ctx, int variant);
const char* lsfm_last_error(lsfm_context* ctx);
void* lsfm_stream(lsfm_context* ctx); /* hipStream_t the library launches on */

/* frees the arrays of a map that the LIBRARY allocated (outputs of the functions below) */
void lsfm_map_release(lsfm_map* g);

/* ---- the three methods the reference's scheduler calls ------------------------------------------ */
/* replaces CLinearSFMImp::lmj_Transform_PF3DStereo (Imp.h:206, Imp.cpp:349-1924): `in` expressed in the frame
 * of its pose `Ref`.  out: library-allocated. */
int lsfm_transform_stereo(lsfm_context* ctx, const lsfm_map* in, int Ref, lsfm_map* out);
/* replaces lmj_Transform_PF3DMono (Imp.h:218, Imp.cpp:3173-6509) */
int lsfm_transform_mono(lsfm_context* ctx, const lsfm_map* in, int Ref, int ScaP, int Fix, lsfm_map* out);

/* replaces lmj_LinearLS_PF3DStereo (Imp.h:207, Imp.cpp:2551-2978): joins End (already in Cur's frame) with Cur
 * and solves.  Unlike the reference (which frees both inputs, Imp.cpp:2937-2958, and publishes the result in
 * the member m_GMapS) inputs are left untouched and the joint map is returned in `joint`.
 * eP_out[6m] / eF_out[3n] (optional, may be NULL) receive the assembled right-hand sides. */
int lsfm_join_stereo(lsfm_context* ctx, const lsfm_map* End, const lsfm_map* Cur, lsfm_map* joint,
                     double* eP_out, double* eF_out);
/* replaces lmj_LinearLS_PF3DMono (Imp.h:221, Imp.cpp:7282-7874).  The reference unwraps the scale-pose angles of
 * both inputs in place (Imp.cpp:7427-7465); here the inputs stay const and the unwrapped values are used inside. */
int lsfm_join_mono(lsfm_context* ctx, const lsfm_map* End, const lsfm_map* Cur, lsfm_map* joint,
                   double* eP_out, double* eF_out);

/* replaces lmj_solveLinearSFMStereo (Imp.h:209, Imp.cpp:2119-2378), same argument list + context.
 * Schur complement on the features, preconditioned CG on the camera system (instead of CHOLMOD),
 * back-substitution.  Writes stVal[0..6m+3n).  x0 (optional, may be NULL): initial guess for the 6m pose
 * scalars.  V is read only (the reference inverts it in place and restores it). */
int lsfm_solve_stereo(lsfm_context* ctx, double* stVal, const double* eb, const double* ea, const double* U,
                      const double* W, const double* V, const int* Ui, const int* Uj, const int* photo,
                      const int* feature, int m, int n, int nU, int nW, const double* x0);
/* replaces lmj_solveLinearSFMMono (Imp.h:223, Imp.cpp:6756-7041); Ref/ScaP/Fix/Sign/FixBlk as at the reference's
 * call site Imp.cpp:7860-7864 (Ref = block index of the reference pose, ScaP = its scalar offset, Fix = scalar
 * index of the gauge-fixed translation). */
int lsfm_solve_mono(lsfm_context* ctx, double* stVal, const double* eb, const double* ea, const double* U,
                    const double* W, const double* V, const int* Ui, const int* Uj, const int* photo,
                    const int* feature, int m, int n, int nU, int nW, int Ref, int ScaP, int Fix, int Sign,
                    int FixBlk, const double* x0);

/* ---- Gauss-Newton polish of the map-joining objective (BASELINE.json north_star "plus the Gauss-Newton BA refinement") ----
 * NO reference counterpart: the reference joins once and has no iterative step (no loop and no residual anywhere in LinearSFMImp.cpp) --
 * PARITY UNPINNED; the tests hold it against the CPU checker's statement of the same steps and against properties (the objective never rises, its gradient falls, a minimiser is
 * a fixed point).  Minimises, over the global state x and ALL N local maps at once,
 *     F(x) = sum_k || x^_k - f_k(x) ||^2_{I_k}
 * x^_k / I_k: estimate / information matrix of local map k, f_k: the reference's own change of frame into map k's frame
 * (Imp.cpp:421-455, Mono 3268-3306) with the Jacobian the reference's transform forms (Imp.cpp:485-683, Mono 3383-3688) -- the objective
 * every join of the tree linearises once.  A step solves H d = b (H = sum J^T I J, b = sum J^T I r) through the same Schur + factorisation
 * as a tree level and takes x += a d, a = 1 halved (at most 8 times) while F does not fall; a step without decrease ends the run.
 * maps[N]: the local maps the tree was built from.  x: the global state to start from -- m, n, stno, stVal (updated in place; arrays of
 * the caller), Ref = the pose its frame is anchored at (Stereo: not in the state; a local map with that Ref is in the global frame),
 * Mono: ScaP / Fix (the gauge: the 6 scalars of pose Ref and scalar Fix of pose ScaP stay where they are), optional pose_origin --
 * e.g. the result of lsfm_tree_download / lsfm_divide_conquer.  type: 0 Stereo, 1 Monocular.  obj[iters + 1], gnorm[iters + 1]: F and
 * max |b_i| over the free scalars at the start and after every step; halvings[iters] (may be NULL; 9: no decrease found).
 * Returns LSFM_OK, LSFM_NOT_CONVERGED (a step's camera system stayed above its residual bound), < 0 errors (LSFM_ERR_ARG: a local
 * variable that is not in the global state, a global variable no map holds, a Stereo map that holds the global reference pose). */
int lsfm_gn_polish(lsfm_context* ctx, const lsfm_map* maps, int N, int type, lsfm_map* x, int iters, double* obj, double* gnorm, int* halvings);

/* replaces pba_inverseV (Imp.h:213, Imp.cpp:3022-3042): V^-1 of the n 3x3 feature blocks, IN PLACE like the reference's (which
 * inverts V in place and restores it afterwards, Imp.cpp:2210-2212, 2365): the upper triangle of the computed inverse, mirrored.
 * m is unused, as in the reference. */
int lsfm_inverse_v(lsfm_context* ctx, double* V, int m, int n);
/* replaces pba_solveFeatures (Imp.h:214, Imp.cpp:2980-3020), same argument list + context: the features' back-substitution
 * dpb_f = IV_f (eb_f - sum_p W_pf^T dpa_p) for given pose values dpa[6m]; IV[9n] as lsfm_inverse_v leaves it, W[18 nW] sorted by
 * feature with mapCor[f] blocks for feature f (the reference's mapPhoto), photo[nW] their poses.  ea is unused, as in the reference. */
int lsfm_solve_features(lsfm_context* ctx, const double* W, const double* IV, const double* ea, const double* eb, const double* dpa, double* dpb,
                        int m, int n, const int* mapCor, const int* photo);

/* Test / debug entry: the BLOCK PATTERN of the camera system S = U - W V^-1 W^T as the device builds it for a joint map
 * given by its index arrays alone (hash set of the pose pairs that share a feature, plus U's pattern; sorted into block
 * CSR).  It stands where the reference marks a dense m x m byte mask and scans it (Imp.cpp:2131-2205, sba_crsm_* 30-76)
 * and where pba_constructAuxCSS{LM,GN} (Imp.cpp:2529-2549 / 7248-7280) lists the same pattern for cholmod_amd.
 * Upper triangle, row by row, columns ascending, the diagonal block first: rowptr[m + 1], colidx[cap]; *nnzb receives the
 * number of blocks (LSFM_ERR_ARG when cap is too small). */
int lsfm_schur_pattern(lsfm_context* ctx, const int* Ui, const int* Uj, const int* photo, const int* feature, int m, int n, int nU, int nW,
                       int* rowptr, int* colidx, int cap, int* nnzb);

/* Test / measurement entry, NO device needed: the host half of the solver that stands where the reference calls cholmod_amd /
 * cholmod_analyze_p (Imp.cpp:2413, 2440 / 7081, 7112) -- nested-dissection ordering along the join tree + symbolic block
 * Cholesky factorisation of a camera system given by its upper block pattern (rowptr[m + 1], colidx: as lsfm_schur_pattern
 * returns it; every diagonal block present) and, per pose, the index of the local map that brought it (origin[m]; NULL: its
 * position).  Outputs (each optional): perm[m] (new -> old), colptr[m + 1] and rowidx[cap] = block CSC of L in the new
 * numbering (rows ascending, diagonal first); info[8] = { blocks of L, height of the elimination tree, supernode groups, group
 * levels, leaf tasks, poses in separators, 0, 0 }; *avg_ms = wall ms of one analysis, averaged over reps.  Returns LSFM_ERR_ARG
 * when cap is too small (info[0] still holds the size needed). */
int lsfm_symbolic_analyse(int m, const int* rowptr, const int* colidx, const int* origin, int reps, int* perm, int* colptr, int* rowidx,
                          int cap, int* info, double* avg_ms);

/* ---- the scheduler itself ------------------------------------------------------------------------ */
/* replaces lmj_PF3D_Divide_Conquer{Stereo,Mono} (Imp.h:205/220, Imp.cpp:1926-2063 / 6511-6630): hierarchical
 * join of maps[0..N) with the reference's binary-tree order; all joins of one tree level run as ONE batch on
 * the device.  maps are read only.  mono: 0 Stereo, 1 Monocular.  out: library-allocated final map.
 * Two phases so that a caller (bench) can time the device part with inputs resident in HBM:
 *   lsfm_tree_upload   copies the N maps to the device (PCIe), returns a handle
 *   lsfm_tree_run      runs the whole tree on the device (this is the region the reference times)
 *   lsfm_tree_download copies the final map back;  lsfm_tree_free releases the handle.
 * Lifetime rule: the result of lsfm_tree_run lives in the CONTEXT's arenas, which every other compute call on the same
 * context (another tree's upload or run, lsfm_transform_*, lsfm_join_*, lsfm_solve_*) reuses.  Download (or export) a
 * tree before the context does anything else; a download after such a call fails with LSFM_ERR_ARG instead of
 * returning overwritten memory.  The resident INPUT maps of a tree are its own: a tree can be run again at any time. */
typedef struct lsfm_tree lsfm_tree;
int lsfm_tree_upload(lsfm_context* ctx, const lsfm_map* maps, int N, int mono, lsfm_tree** out);
int lsfm_tree_run(lsfm_context* ctx, lsfm_tree* tree, lsfm_stats* stats);
/* on = 1 (default): the final map is re-expressed in its first frame (Imp.cpp:2039-2063).  on = 0: it is left in the
 * frame of its last join -- what the reference's loop holds for an intermediate tree node; used when the tree is a
 * SUBTREE whose root is joined further by another call (multi-GPU sharding). */
int lsfm_tree_set_final_reanchor(lsfm_tree* tree, int on);
/* on = 1 (default): the first lsfm_tree_run of a tree records what depends on its STRUCTURE only (container sizes, block
 * pattern of every Schur system, ordering / elimination tree / supernodes of its factorisation -- the reference repeats
 * that analysis in every join, cholmod_analyze_p, Imp.cpp:2440) and later runs of the same resident tree reuse it: they are
 * enqueued without a host <-> device round trip.  The numeric work of a run is the same either way.  on = 0: every run
 * analyses from scratch (what a first run costs; lsfm_stats.t_total_ms of the first run reports it too). */
int lsfm_tree_set_plans(lsfm_tree* tree, int on);
int lsfm_tree_download(lsfm_context* ctx, lsfm_tree* tree, lsfm_map* out);
/* Level checkpoint / resume (the reference keeps every node of the tree in RAM, m_LMsetS[i] = m_GMapS, Imp.cpp:2032, and
 * writes none).  lsfm_tree_set_stop_level(tree, L > 0): a run ends after L tree levels, leaving the ceil(N / 2^L) nodes of that level
 * -- each with its state, its information matrix, its first frame (FRef / FScaP / FFix) and the origins of its poses, odd-indexed ones
 * not yet taken back to their first frame: exactly what the reference's loop holds at that point (Imp.cpp:1997-2025 re-anchors while it
 * builds the NEXT level).  lsfm_tree_node_count: nodes of the level the last run ended at (1 after a whole tree; 0: not run /
 * overwritten).  lsfm_tree_download_node: node k of it (library-allocated, lsfm_map_release).  The nodes uploaded as the maps of a new
 * tree (lsfm_tree_upload: FRef, FScaP, FFix and pose_origin are honoured) and run to the end give the map the uninterrupted tree
 * gives; lsfm_write_localmap stores a node with a trailer the reference's reader never reaches, lsfm_read_localmap restores it.
 * L = 0 (default): the whole tree. */
int lsfm_tree_set_stop_level(lsfm_tree* tree, int levels);
int lsfm_tree_node_count(lsfm_context* ctx, lsfm_tree* tree);
int lsfm_tree_download_node(lsfm_context* ctx, lsfm_tree* tree, int k, lsfm_map* out);
/* the state vector of the final map alone (no information blocks): *m poses, *n features; stno / stVal (each optional, caller's
 * arrays of cap >= 6 m + 3 n entries) in the layout of lsfm_map -- both NULL: sizes only */
int lsfm_tree_download_state(lsfm_context* ctx, lsfm_tree* tree, int* m, int* n, int* stno, double* stVal, size_t cap);
void lsfm_tree_free(lsfm_context* ctx, lsfm_tree* tree);
/* ---- device-resident hand-off of a tree node (multi-GPU sub-tree sharding; no counterpart in the reference, whose
 * scheduler keeps every node in one process: m_LMsetS[i] = m_GMapS, Imp.cpp:2032) ---------------------------------
 * A finished tree's final map is PACKED into one contiguous device buffer (a 256-byte header, then the arrays
 * with map-local indices, each 256-byte aligned) that the caller owns and may move to another GPU by any means that
 * moves device bytes (RCCL send/recv, hipMemcpyPeer).  lsfm_tree_upload_dev builds the resident inputs of a new tree
 * from N such buffers on this context's device -- no host copy of the arrays, only the N headers are read back.
 *   lsfm_tree_export_size  bytes lsfm_tree_export_dev will write (0: tree not run / overwritten)
 *   lsfm_tree_export_dev   dst: device memory of >= cap bytes, accessible from the context's device
 *   lsfm_packed_size       total bytes of a packed map, from its first LSFM_PACK_HEADER_BYTES (256) bytes copied to the
 *                          host (0: not a pack -- magic, version, sizes or the array offsets stored in it do not fit)
 *   lsfm_tree_upload_dev   packed[k]: device pointer of packed map k (pose origins travel inside the pack) */
size_t lsfm_tree_export_size(lsfm_context* ctx, lsfm_tree* tree);
int lsfm_tree_export_dev(lsfm_context* ctx, lsfm_tree* tree, void* dst, size_t cap);
#define LSFM_PACK_HEADER_BYTES 256
size_t lsfm_packed_size(const void* host_header256);
int lsfm_tree_upload_dev(lsfm_context* ctx, const void* const* packed, int N, int mono, lsfm_tree** out);
/* new VALUES for the resident inputs of a tree made by lsfm_tree_upload_dev: N packed maps with the same sizes as the ones
 * it was built from -- the next step of a scheduler that joins the same sub-tree roots again.  Keeps the tree's
 * allocations; its plans (lsfm_tree_set_plans) are kept when the labels and index arrays are the same too (a digest of
 * them is taken on the device at upload and at reload) and dropped otherwise, so that a reload with another structure
 * costs an analysing run instead of a wrong result. */
int lsfm_tree_reload_dev(lsfm_context* ctx, lsfm_tree* tree, const void* const* packed, int N);

/* ---- feature-sharded joins: the top of the tree over several GPUs (no counterpart in the reference; SURVEY 8e "P2") --------
 * The Schur loop (Imp.cpp:2244-2332), the back-substitution (2980-3020) and the feature part of the information transform
 * (1270-1917) are sums over features.  G processes (one per GPU) each hold a SLICE of every map -- all poses and U blocks,
 * the features f with feat_id % G == slice together with their V and W blocks (the same feature of two maps lands in the
 * same slice, so common features still meet in a join) -- and run the SAME tree; three sums per level cross the GPUs:
 *   transform   the pose rows of I C and the hub-hub block (sum over the features of W C_f, C_f^T (I C)_f)
 *   join        S = U - sum_f W V^-1 W^T and E = eP - sum_f W V^-1 eF (U's part is taken by rank 0 alone); in a run that
 *               analyses also the union of the ranks' pose-pair patterns
 *   solve       the pose solution of rank 0 replaces everyone's (the replicated factorisations may differ in the last bit)
 * The library does not link a collective library: the caller hands over device memory `dev_buf` that the reduced arrays
 * live in and a function that sums `count` elements at `offset_bytes` of that buffer over all ranks, in place, ordered after
 * the work already enqueued on `hip_stream` and before the work enqueued on it afterwards (RCCL: ncclAllReduce on that stream;
 * torch.distributed: all_reduce under torch.cuda.ExternalStream(hip_stream)).  Returns 0 on success.  Every rank must call
 * lsfm_tree_run on its slice tree at the same time; the number and sizes of the calls are the same on every rank.
 * Every sum is announced by a sum of 4 int64 at offset 0 of the buffer (the first 256 bytes are the library's): {ranks that have
 * failed, ranks that have not, count, dtype}.  A rank whose run fails between two sums (out of memory, a HIP error, a buffer too
 * small) does not leave its peers waiting: it follows their headers, adds zeros to every sum they make and reaches the exchange of
 * the run's flags with them, where every rank learns of the failure -- lsfm_tree_run then returns an error on EVERY rank
 * (the failed rank its own, the others LSFM_ERR_INTERNAL "another rank ... failed").  Only a failure of `fn` itself (non-zero
 * return: the communicator is gone) cannot be matched; the caller must then tear the job down. */
#define LSFM_DTYPE_F64 0
#define LSFM_DTYPE_I64 1
typedef int (*lsfm_allreduce_fn)(void* user, size_t offset_bytes, size_t count, int dtype, void* hip_stream);
/* fn == NULL: off (the default).  world == 1 is allowed (every sum is then this rank's own: a way to exercise the caller's function) */
int lsfm_tree_set_comm(lsfm_tree* tree, int rank, int world, lsfm_allreduce_fn fn, void* user, void* dev_buf, size_t dev_bytes);
/* Distributes the POSE-side solve of such a tree as well (SURVEY 8e "P2"; the reference factors every camera system in one
 * CHOLMOD call, Imp.cpp:2444-2445).  block_maps = 2^k: rank r joined block r of block_maps consecutive local maps of the whole
 * tree before this tree took over (the pose origins inside the packs say which local map brought a pose).  The nested dissection
 * of every camera system follows the join tree, so a pose whose separator level lies inside one block is only ever coupled to
 * poses of that block and to the separators between blocks: rank r factors the columns of block r's poses and collects their
 * updates of the inter-block separator columns (64-bit fixed-point accumulators); those are summed over the ranks with ONE integer
 * all-reduce -- exact, so every rank holds the same bits -- and every rank factors the separators.  The triangular solves go the
 * same way (own columns, sum of the shared rows, shared columns | shared columns, own columns, sum of the solution).  0 (default):
 * every rank factors everything.  Set on every rank alike, before the first run. */
int lsfm_tree_set_comm_blocks(lsfm_tree* tree, int block_maps);
/* The final map of a finished tree cut into `nslices` packs (same format as lsfm_tree_export_dev: every pack holds ALL poses
 * and U blocks, and the features with feat_id % nslices == slice in their order, with their V and W blocks).
 *   lsfm_tree_export_slice_sizes  bytes of every slice pack, sizes[nslices]
 *   lsfm_tree_export_slice_dev    writes pack `slice` to dst (device memory of >= cap bytes) */
int lsfm_tree_export_slice_sizes(lsfm_context* ctx, lsfm_tree* tree, int nslices, size_t* sizes);
int lsfm_tree_export_slice_dev(lsfm_context* ctx, lsfm_tree* tree, int nslices, int slice, void* dst, size_t cap);

/* convenience: upload + run + download */
int lsfm_divide_conquer(lsfm_context* ctx, const lsfm_map* maps, int N, int mono, lsfm_map* out, lsfm_stats* stats);

/* ---- file formats (Imp.cpp:3044-3132, 6660-6754, 2102-2117, 7876-7967) ---------------------------- */
int lsfm_read_localmap(const char* path, int mono, lsfm_map* out); /* out: library-allocated */
/* dir/localmap_<first>.txt ... localmap_<first+count-1>.txt (the loop around lmj_readInformation*, Imp.cpp:125 naming)
 * on `threads` host threads (<= 0: one per core, at most 32); out[count] library-allocated, filled in order.  On a
 * failure nothing is kept and *failed (optional) is the number of the first file that could not be read. */
int lsfm_read_localmaps(const char* dir, int first, int count, int mono, int threads, lsfm_map* out, int* failed);
/* one map in the local-map text format at %.17g (write -> lsfm_read_localmap is the identity): stores a tree node with its
 * information matrix, which the reference computes (DOC.pdf p.1) but never writes */
int lsfm_write_localmap(const char* path, int mono, const lsfm_map* map);
/* Binary cache of a set of local maps ("optional binary cache with identical semantics" beside the parallel parser of
 * lmj_readInformation*'s files, Imp.cpp:3044-3132 / 6660-6754): one file, the arrays the text reader produced, bit for bit (layout:
 * lsfm_io.cpp).  lsfm_write_mapset writes maps[N] (atomically: a temporary file renamed); lsfm_mapset_info gives N and the type
 * (LSFM_ERR_IO: missing / not a cache); lsfm_read_mapset reads maps first .. first+count-1 (0-based) on `threads` host threads into
 * out[count] (library-allocated; LSFM_ERR_IO and nothing kept when the file is truncated, foreign or of the other map type). */
int lsfm_write_mapset(const char* path, const lsfm_map* maps, int N, int mono);
int lsfm_mapset_info(const char* path, int* N, int* mono);
/* Eight bytes of the cache's header are the writer's: a stamp of what the cache was made from (0: none).  set_to != NULL writes it, stamp
 * (optional) receives what the file holds afterwards.  The command line stamps a cache with a hash of the resolved -path and of the size and
 * modification time of every localmap_k.txt it parsed, and parses again when the files no longer match (advisor, round 4: a cache was trusted
 * whenever it held enough maps of the right type). */
int lsfm_mapset_stamp(const char* path, unsigned long long* stamp, const unsigned long long* set_to);
int lsfm_read_mapset(const char* path, int mono, int first, int count, int threads, lsfm_map* out);
int lsfm_save_state(const char* path, const double* st, const int* stno, int n);
/* the same state vector as raw doubles (parity tooling): int32 n, int32 0, stno[n] (+ 4 bytes of padding when n is odd),
 * st[n] float64 */
int lsfm_save_state_bin(const char* path, const double* st, const int* stno, int n);
int lsfm_save_poses(const char* pose_path, const char* feat_path, const int* stno, const double* st, int n);

/* ---- stand-alone kernel entry for measurement: y = S x on a symmetric 6x6-block matrix given as upper block
 * CSR (rowptr[m+1], colidx[nnzb], val[nnzb*36], diagonal blocks full); runs `reps` launches and returns the
 * average launch time in ms measured with HIP events on the context's stream. */
int lsfm_spmv_bench(lsfm_context* ctx, int m, const int* rowptr, const int* colidx, const double* val,
                    const double* x, double* y, int reps, double* avg_ms, double* algorithmic_bytes);

/* Measurement only: how fast this chip moves nblocks 6x3 blocks of W (144 bytes each) from one array to another with the
 * access pattern of the W kernels -- mode 0: one lane per block, the block as 18 doubles (lane stride 144 bytes, what
 * k_tr_entries / k_join_rhs_w / k_backsub do); mode 1: one lane per block, nine 16-byte loads; mode 2: consecutive lanes
 * on consecutive 16 bytes (the stream copy).  avg_ms over reps launches (HIP events); bytes moved = 2 * 144 * nblocks. */
int lsfm_wstream_bench(lsfm_context* ctx, long long nblocks, int mode, int reps, double* avg_ms);

/* Self-test of the library's own fill and small-copy kernels (they stand where hipMemsetAsync / hipMemcpyAsync stood until round 5:
 * every accumulator of the path is cleared and every index table arrives through them): fills of `cases` pseudo-random (offset, length,
 * byte) triples -- unaligned heads and tails included -- into a guarded buffer, host -> device copies of random lengths through the
 * pinned ring (one by one and as batches), each read back and compared.  Returns LSFM_OK, or LSFM_ERR_INTERNAL with the first
 * mismatch in lsfm_last_error. */
int lsfm_selftest_prims(lsfm_context* ctx, int cases, unsigned seed);

#ifdef __cplusplus
}
#endif
#endif
