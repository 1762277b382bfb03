/* oracle/ref_main.c -- TEST INFRASTRUCTURE ONLY (never linked into the product).
 *
 * Our own command dispatcher for the *reference's* object files, compiled by
 * oracle/Makefile from the sources where they lie under /root/reference.
 * The reference's own dispatcher (main.c) also pulls in bam2bam.c, which needs
 * libzmq headers this image lacks, so bam2bam.c is treated as unbuildable here
 * (see DESIGN.md "Oracle"); every other file of the per-read alignment path
 * compiles unmodified.  This driver only routes argv to the reference's own
 * sub-command entry points (prototypes: reference main.h:18-25).
 */
#include <stdio.h>
#include <string.h>

int bwa_index(int argc, char *argv[]);       /* reference bwtindex.c:39  */
int bwa_aln(int argc, char *argv[]);         /* reference bwtaln.c:299   */
int bwa_sai2sam_se(int argc, char *argv[]);  /* reference bwase.c:723    */
int bwa_sai2sam_pe(int argc, char *argv[]);  /* reference bwape.c:759    */

/* The reference prints this header line from main.c:41; bwase.c/bwape.c call it. */
void bwa_print_sam_PG(void)
{
	printf("@PG\tID:bwa\tPN:bwa\tVN:oracle-ref\n");
}

#ifndef REF_NO_MAIN
int main(int argc, char *argv[])
{
	int r = 1;
	if (argc < 2) {
		fprintf(stderr, "usage: bwa_ref <index|aln|samse|sampe> ...\n");
		return 1;
	}
	if (strcmp(argv[1], "index") == 0) r = bwa_index(argc - 1, argv + 1);
	else if (strcmp(argv[1], "aln") == 0) r = bwa_aln(argc - 1, argv + 1);
	else if (strcmp(argv[1], "samse") == 0) r = bwa_sai2sam_se(argc - 1, argv + 1);
	else if (strcmp(argv[1], "sampe") == 0) r = bwa_sai2sam_pe(argc - 1, argv + 1);
	else fprintf(stderr, "bwa_ref: unknown command %s\n", argv[1]);
	fflush(stdout);
	return r;
}
#endif
