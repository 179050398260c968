/* ORACLE -- TEST INFRASTRUCTURE ONLY.  Command line with the reference's flags (Imp.cpp:7989-8105):
 *   lsfm_oracle -path <dir> -num <N> -type Monocular|Stereo [-p poses] [-f features] [-st state]
 * extra (oracle only): -hash 1 (sort-based feature matching instead of the O(n1*n2) find), -full <file>
 * (final state at %.17g), -quiet 1.  Used as the CPU baseline ("port") by bench.py.
 */
#include "lsfm_oracle.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

int main(int argc, char** argv)
{
	const char *path = NULL, *st = NULL, *pose = NULL, *fea = NULL, *type = NULL, *full = NULL;
	int num = 0, i, mono, quiet = 0, rc;
	orc_map* maps;
	orc_map out;
	double timing[4];
	for (i = 1; i < argc; i++)
	{
		const char* name = argv[i];
		if (name[0] != '-') return 0;
		while (*name == '-') name++;
		if (!strcmp(name, "help")) { printf("see oracle/lsfm_oracle_cli.c\n"); return 0; }
		if (i + 1 >= argc) break;
		if (!strcmp(name, "path")) path = argv[++i];
		else if (!strcmp(name, "st")) st = argv[++i];
		else if (!strcmp(name, "p")) pose = argv[++i];
		else if (!strcmp(name, "f")) fea = argv[++i];
		else if (!strcmp(name, "num")) num = atoi(argv[++i]);
		else if (!strcmp(name, "type")) type = argv[++i];
		else if (!strcmp(name, "hash")) orc_set_match_hash(atoi(argv[++i]));
		else if (!strcmp(name, "full")) full = argv[++i];
		else if (!strcmp(name, "quiet")) quiet = atoi(argv[++i]);
	}
	if (!path) { printf("LinerSFM Error: Please Input Right File Path:\n"); return 0; }
	if (!num) { printf("LinerSFM Error: Please Set Local Map Number:\n"); return 0; }
	if (!type || (strcmp(type, "Monocular") && strcmp(type, "Stereo"))) { printf("LinerSFM Error: Please Set Data Type:\n"); return 0; }
	mono = !strcmp(type, "Monocular");
	maps = calloc(num, sizeof *maps);
	for (i = 0; i < num; i++)
	{
		char fn[4096];
		snprintf(fn, sizeof fn, "%s/localmap_%d.txt", path, i + 1);
		if (orc_read_map(fn, mono, &maps[i])) { fprintf(stderr, "cannot read %s\n", fn); return 1; }
	}
	rc = orc_divide_conquer(maps, num, mono, &out, !quiet, timing);
	printf("Total Used Time:  %lf  sec\n\n", timing[0]);
	fprintf(stderr, "oracle timing: total %.6f s, transform %.6f s, join-assembly %.6f s, solve %.6f s, rc=%d\n",
	        timing[0], timing[1], timing[2], timing[3], rc);
	if (st) orc_save_state(st, out.stVal, out.stno, out.m * 6 + out.n * 3);
	if (pose && fea) orc_save_poses(pose, fea, out.stno, out.stVal, out.m * 6 + out.n * 3);
	if (full)
	{
		FILE* f = fopen(full, "w");
		if (f)
		{
			for (i = 0; i < out.m * 6 + out.n * 3; i++) fprintf(f, "%d %.17g\n", out.stno[i], out.stVal[i]);
			fclose(f);
		}
	}
	orc_map_free(&out);
	free(maps);
	return rc ? 3 : 0;
}
