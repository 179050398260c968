// TEST INFRASTRUCTURE ONLY (oracle side) -- never linked into or called by the product library.
//
// Harness around the REAL reference translation unit (/root/reference/linux/src/LinearSFMImp/
// LinearSFMImp.cpp), compiled where it lies by oracle/Makefile into oracle/_ref/.  The reference
// needs CHOLMOD (absent from this image, no stand-in is written), so a full reference build is
// impossible here.  What CAN be run are the CHOLMOD-free entry points:
//     lmj_readInformation{Stereo,Mono}      Imp.cpp:3044 / 6660   (file format)
//     lmj_Transform_PF3D{Stereo,Mono}       Imp.cpp:349  / 3173   (state + information transform)
//     lmj_LinearLS_PF3D{Stereo,Mono}        Imp.cpp:2551 / 7282   (join assembly) up to the point where
//                                           it calls lmj_solveLinearSFM{Stereo,Mono}
// The two lmj_solveLinearSFM* methods (Schur + CHOLMOD) are interposed by the capture hooks below:
// they only RECORD the arguments the reference assembled (joint U/W/V, eP, eF, index arrays) and
// return; nothing after that point is used.  libref_imp.so is linked lazily with the 12 cholmod_*
// symbols left undefined and they are never called (the CLinearSFMImp constructor, which calls
// cholmod_start, is bypassed by running methods on zeroed raw storage).
//
// From the solve stage (Imp.cpp:2119-2378 / 6756-7041) the pieces that do not touch CHOLMOD are public methods and are
// run for real by the "parts" mode on arrays handed over in a file:
//     pba_inverseV                          Imp.cpp:3022   V^-1 (Eigen 3x3 inverse, symmetrised write-back)
//     pba_solveFeatures                     Imp.cpp:2980   back-substitution of the features for given pose values
//     pba_constructAuxCSS{LM,GN}            Imp.cpp:2529 / 7248   block pattern handed to cholmod_amd
//     pba_constructCSS{LM,GN}               Imp.cpp:2451 / 7123   scalar CSC (upper) handed to cholmod_factorize; they walk
//                                           the S blocks through sba_crsm_elmidx (Imp.cpp:55-76, static, reached that way)
// pba_constructCSS* write through m_sparseS->x: the harness points m_sparseS at a cholmod_sparse STRUCT (vendored header
// type) whose x is a plain buffer -- no CHOLMOD function is called or replaced.  The Schur loop itself is inline in
// lmj_solveLinearSFM* between allocations and cholmod_start and cannot be called on its own.
//
// Usage:
//   ref_dump pair  Stereo|Monocular  A.txt B.txt out.bin     transform A to B's frame, assemble join
//   ref_dump trans Stereo A.txt Ref out.bin
//   ref_dump trans Monocular A.txt Ref ScaP Fix out.bin
//   ref_dump parts Stereo|Monocular  in.bin out.bin          in.bin (same tagged format): m n nW V W photo mapPhoto ea eb
//                                                            dpa rowptr colidx S [Ref ScaP Fix]
//   ref_dump save  both|pose|feat in.bin pose.txt feat.txt state.txt
//                                                            in.bin: stno stVal.  The REAL writers lmj_SaveStateVector
//                                                            (Imp.cpp:2102) and lmj_SavePoses_3DPF (Imp.cpp:7876; with
//                                                            "pose" / "feat" the other path is handed over as NULL) write the
//                                                            three text files: the bytes the product's writers are held to
// Output: tagged binary ("name dtype count\n" + raw little-endian payload), read by tests/refdump.py.
#define private public
#include "LinearSFMImp.h"
#undef private
#include <new>

static FILE* g_out = NULL;

static void put_d(const char* name, const double* p, long n)
{
	fprintf(g_out, "%s f8 %ld\n", name, n);
	if (n > 0) fwrite(p, sizeof(double), n, g_out);
}
static void put_i(const char* name, const int* p, long n)
{
	fprintf(g_out, "%s i4 %ld\n", name, n);
	if (n > 0) fwrite(p, sizeof(int), n, g_out);
}
static void put_1(const char* name, int v) { put_i(name, &v, 1); }

template <class M> static void dump_map(const char* pre, const M& g)
{
	char nm[128];
	int r = 6 * g.m + 3 * g.n;
#define NM(s) (snprintf(nm, sizeof nm, "%s.%s", pre, s), nm)
	put_1(NM("m"), g.m); put_1(NM("n"), g.n); put_1(NM("nU"), g.nU); put_1(NM("nW"), g.nW);
	put_1(NM("Ref"), g.Ref); put_1(NM("FRef"), g.FRef);
	put_i(NM("stno"), g.stno, r); put_d(NM("stVal"), g.stVal, r);
	put_d(NM("U"), g.U, 36L * g.nU); put_i(NM("Ui"), g.Ui, g.nU); put_i(NM("Uj"), g.Uj, g.nU);
	put_d(NM("W"), g.W, 18L * g.nW); put_i(NM("photo"), g.photo, g.nW); put_i(NM("feature"), g.feature, g.nW);
	put_d(NM("V"), g.V, 9L * g.n); put_i(NM("FBlock"), g.FBlock, g.n);
#undef NM
}

// ---- capture hooks: take precedence over libref_imp.so's own definitions (executable symbols are
// searched first), so the reference's lmj_LinearLS_* calls land here instead of Schur+CHOLMOD.
void CLinearSFMImp::lmj_solveLinearSFMStereo(double* stVal, double* eb, double* ea, double* U, double* W, double* V,
                                             int* Ui, int* Uj, int* photo, int* feature, int m, int n, int nU, int nW)
{
	put_1("solve.m", m); put_1("solve.n", n); put_1("solve.nU", nU); put_1("solve.nW", nW);
	put_d("solve.ea", ea, 6L * m); put_d("solve.eb", eb, 3L * n);
	put_d("solve.U", U, 36L * nU); put_i("solve.Ui", Ui, nU); put_i("solve.Uj", Uj, nU);
	put_d("solve.W", W, 18L * nW); put_i("solve.photo", photo, nW); put_i("solve.feature", feature, nW);
	put_d("solve.V", V, 9L * n);
	for (int i = 0; i < 6 * m + 3 * n; i++) stVal[i] = 0.0;
}

void CLinearSFMImp::lmj_solveLinearSFMMono(double* stVal, double* eb, double* ea, double* U, double* W, double* V,
                                           int* Ui, int* Uj, int* photo, int* feature, int m, int n, int nU, int nW,
                                           int Ref, int ScaP, int Fix, int Sign, int FixBlk)
{
	put_1("solve.m", m); put_1("solve.n", n); put_1("solve.nU", nU); put_1("solve.nW", nW);
	put_1("solve.Ref", Ref); put_1("solve.ScaP", ScaP); put_1("solve.Fix", Fix); put_1("solve.Sign", Sign);
	put_1("solve.FixBlk", FixBlk);
	put_d("solve.ea", ea, 6L * m); put_d("solve.eb", eb, 3L * n);
	put_d("solve.U", U, 36L * nU); put_i("solve.Ui", Ui, nU); put_i("solve.Uj", Uj, nU);
	put_d("solve.W", W, 18L * nW); put_i("solve.photo", photo, nW); put_i("solve.feature", feature, nW);
	put_d("solve.V", V, 9L * n);
	for (int i = 0; i < 6 * m + 3 * n; i++) stVal[i] = 0.0;
}

// ---- reader of the tagged binary (the format put_d / put_i write) ------------------------------------------
struct Blob { char name[128]; char dt[8]; long n; void* p; };
static int read_blobs(const char* path, Blob* b, int cap)
{
	FILE* f = fopen(path, "rb");
	if (!f) { perror(path); exit(1); }
	int k = 0;
	char line[256];
	while (k < cap && fgets(line, sizeof line, f))
	{
		if (sscanf(line, "%127s %7s %ld", b[k].name, b[k].dt, &b[k].n) != 3) break;
		size_t sz = (b[k].dt[0] == 'f' ? 8 : 4) * (size_t)b[k].n;
		b[k].p = malloc(sz ? sz : 1);
		if (sz && fread(b[k].p, 1, sz, f) != sz) { fprintf(stderr, "short read in %s\n", path); exit(1); }
		k++;
	}
	fclose(f);
	return k;
}
static Blob* find_blob(Blob* b, int n, const char* name)
{
	for (int i = 0; i < n; i++) if (strcmp(b[i].name, name) == 0) return &b[i];
	fprintf(stderr, "parts: array %s missing\n", name);
	exit(1);
	return NULL;
}
#define BD(name) ((double*)find_blob(bl, nb, name)->p)
#define BI(name) ((int*)find_blob(bl, nb, name)->p)

static void run_parts(CLinearSFMImp* imp, bool mono, const char* inp)
{
	static Blob bl[64];
	int nb = read_blobs(inp, bl, 64);
	const int m = BI("m")[0], n = BI("n")[0], nW = BI("nW")[0];
	// pba_inverseV works in place (the reference saves / restores V around it, Imp.cpp:2210-2212, 2365)
	double* IV = (double*)malloc(sizeof(double) * 9 * (n ? n : 1));
	memcpy(IV, BD("V"), sizeof(double) * 9 * n);
	imp->pba_inverseV(IV, m, n);
	put_d("parts.IV", IV, 9L * n);
	// pba_solveFeatures(W, IV, ea, eb, dpa, dpb, m, n, mapPhoto, photo)
	double* dpb = (double*)calloc(3 * (n ? n : 1), sizeof(double));
	imp->pba_solveFeatures(BD("W"), IV, BD("ea"), BD("eb"), BD("dpa"), dpb, m, n, BI("mapPhoto"), BI("photo"));
	put_d("parts.dpb", dpb, 3L * n);
	(void)nW;
	// the mask and the CRS index the reference's solver builds from it (Imp.cpp:2131-2205): here from the given pattern
	int *rowptr = BI("rowptr"), *colidx = BI("colidx");
	const int nuis = rowptr[m];
	char* smask = (char*)calloc((size_t)m * m, 1);
	for (int i = 0; i < m; i++) for (int k = rowptr[i]; k < rowptr[i + 1]; k++) smask[(size_t)i * m + colidx[k]] = 1;
	sba_crsm Sidxij;
	Sidxij.nr = Sidxij.nc = m; Sidxij.nnz = nuis;
	Sidxij.val = (int*)malloc(sizeof(int) * (nuis ? nuis : 1));
	for (int k = 0; k < nuis; k++) Sidxij.val[k] = k;
	Sidxij.colidx = colidx; Sidxij.rowptr = rowptr;
	int* Ap = (int*)calloc(m + 2, sizeof(int));
	int* Aii = (int*)calloc(nuis + 1, sizeof(int));
	const int dim = mono ? 6 * m - 7 : 6 * m;
	int* Sp = (int*)calloc(6 * m + 2, sizeof(int));
	int* Si = (int*)calloc((size_t)nuis * 36 + 1, sizeof(int));
	double* Sx = (double*)calloc((size_t)nuis * 36 + 1, sizeof(double));
	cholmod_sparse sp; // the vendored struct only; its x is where pba_constructCSS* write the values
	memset(&sp, 0, sizeof sp);
	sp.x = Sx; sp.p = Sp; sp.i = Si;
	imp->m_sparseS = &sp;
	if (!mono)
	{
		imp->pba_constructAuxCSSLM(Ap, Aii, m, smask);
		imp->pba_constructCSSLM(Si, Sp, NULL, BD("S"), Sidxij, false, m, smask);
		put_i("parts.Ap", Ap, m + 1);
		put_i("parts.Aii", Aii, Ap[m]);
	}
	else
	{
		const int Ref = BI("Ref")[0], ScaP = BI("ScaP")[0], Fix = BI("Fix")[0];
		imp->pba_constructAuxCSSGN(Ap, Aii, m, smask, Ref);
		imp->pba_constructCSSGN(Si, Sp, NULL, BD("S"), Sidxij, false, m, smask, Ref, ScaP, Fix);
		put_i("parts.Ap", Ap, m);
		put_i("parts.Aii", Aii, Ap[m - 1]);
	}
	put_i("parts.Sp", Sp, dim + 1);
	put_i("parts.Si", Si, Sp[dim]);
	put_d("parts.Sx", Sx, Sp[dim]);
}

static int run_save(CLinearSFMImp* imp, char** argv)
{
	static Blob bl[8];
	int nb = read_blobs(argv[3], bl, 8);
	Blob* bs = find_blob(bl, nb, "stno");
	const int n = (int)bs->n;
	char* pose = strcmp(argv[2], "feat") == 0 ? NULL : argv[4];
	char* feat = strcmp(argv[2], "pose") == 0 ? NULL : argv[5];
	imp->lmj_SaveStateVector(argv[6], BD("stVal"), BI("stno"), n);
	imp->lmj_SavePoses_3DPF(pose, feat, BI("stno"), BD("stVal"), n);
	return 0;
}

int main(int argc, char** argv)
{
	if (argc < 5) { fprintf(stderr, "usage: see header of ref_harness.cpp\n"); return 2; }
	// raw zeroed storage: the constructor (cholmod_start) must not run.
	void* raw = calloc(1, sizeof(CLinearSFMImp) + 64);
	CLinearSFMImp* imp = reinterpret_cast<CLinearSFMImp*>(raw);
	if (strcmp(argv[1], "save") == 0) return argc == 7 ? run_save(imp, argv) : 2;
	bool mono = strcmp(argv[2], "Monocular") == 0;
	const char* outp = argv[argc - 1];
	g_out = fopen(outp, "wb");
	if (!g_out) { perror(outp); return 1; }

	if (strcmp(argv[1], "trans") == 0)
	{
		if (!mono)
		{
			LocalMapInfoStereo A, E;
			imp->lmj_readInformationStereo(A, argv[3]);
			dump_map("in", A);
			imp->m_GMapS = A;
			imp->lmj_Transform_PF3DStereo(E, atoi(argv[4]));
			dump_map("out", E);
		}
		else
		{
			LocalMapInfo A, E;
			imp->lmj_readInformationMono(A, argv[3]);
			dump_map("in", A);
			put_1("in.ScaP", A.ScaP); put_1("in.Fix", A.Fix); put_1("in.Sign", A.Sign);
			imp->m_GMap = A;
			imp->lmj_Transform_PF3DMono(E, atoi(argv[4]), atoi(argv[5]), atoi(argv[6]));
			dump_map("out", E);
			put_1("out.ScaP", E.ScaP); put_1("out.Fix", E.Fix); put_1("out.Sign", E.Sign);
			put_1("out.FScaP", E.FScaP); put_1("out.FFix", E.FFix);
		}
	}
	else if (strcmp(argv[1], "pair") == 0)
	{
		if (!mono)
		{
			LocalMapInfoStereo A, B, E;
			imp->lmj_readInformationStereo(A, argv[3]);
			imp->lmj_readInformationStereo(B, argv[4]);
			imp->m_GMapS = A;
			imp->lmj_Transform_PF3DStereo(E, B.Ref);
			dump_map("end", E);
			imp->lmj_LinearLS_PF3DStereo(E, B);      // -> capture hook
			put_1("joint.m", imp->m_GMapS.m); put_1("joint.n", imp->m_GMapS.n);
			put_1("joint.Ref", imp->m_GMapS.Ref); put_1("joint.FRef", imp->m_GMapS.FRef);
			put_i("joint.stno", imp->m_GMapS.stno, imp->m_GMapS.r);
			put_i("joint.FBlock", imp->m_GMapS.FBlock, imp->m_GMapS.n);
		}
		else
		{
			LocalMapInfo A, B, E;
			imp->lmj_readInformationMono(A, argv[3]);
			imp->lmj_readInformationMono(B, argv[4]);
			imp->m_GMap = A;
			imp->lmj_Transform_PF3DMono(E, B.Ref, B.ScaP, B.Fix);
			dump_map("end", E);
			put_1("end.ScaP", E.ScaP); put_1("end.Fix", E.Fix); put_1("end.Sign", E.Sign);
			imp->lmj_LinearLS_PF3DMono(E, B);        // -> capture hook
			put_1("joint.m", imp->m_GMap.m); put_1("joint.n", imp->m_GMap.n);
			put_1("joint.nU", imp->m_GMap.nU); put_1("joint.nW", imp->m_GMap.nW);
			put_1("joint.Ref", imp->m_GMap.Ref); put_1("joint.FRef", imp->m_GMap.FRef);
			put_1("joint.ScaP", imp->m_GMap.ScaP); put_1("joint.Fix", imp->m_GMap.Fix);
			put_1("joint.Sign", imp->m_GMap.Sign); put_1("joint.FScaP", imp->m_GMap.FScaP);
			put_1("joint.FFix", imp->m_GMap.FFix);
			put_i("joint.stno", imp->m_GMap.stno, imp->m_GMap.r);
			put_i("joint.FBlock", imp->m_GMap.FBlock, imp->m_GMap.n);
		}
	}
	else if (strcmp(argv[1], "parts") == 0) run_parts(imp, mono, argv[3]);
	else { fprintf(stderr, "unknown mode %s\n", argv[1]); return 2; }
	fclose(g_out);
	return 0;
}
