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
// Usage:
//   ref_dump pair  Stereo|Monocular  A.txt B.txt out.bin     transform A to B's frame, assemble join
//   ref_dump trans Stereo A.txt Ref out.bin
//   ref_dump trans Monocular A.txt Ref ScaP Fix out.bin
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

int main(int argc, char** argv)
{
	if (argc < 5) { fprintf(stderr, "usage: see header of ref_harness.cpp\n"); return 2; }
	// raw zeroed storage: the constructor (cholmod_start) must not run.
	void* raw = calloc(1, sizeof(CLinearSFMImp) + 64);
	CLinearSFMImp* imp = reinterpret_cast<CLinearSFMImp*>(raw);
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
	else { fprintf(stderr, "unknown mode %s\n", argv[1]); return 2; }
	fclose(g_out);
	return 0;
}
