/* oracle/nabwa_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement of the per-read alignment hot path of mpieva/network-aware-bwa
 * (SURVEY.md section 8a).  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this; the product (libnabwa.so) never does.
 * Parity of this restatement is PINNED: tests/test_oracle_*.py check it against the
 * golden vectors in tests/golden/ that were produced by the reference's own code
 * (oracle/_ref, built from /root/reference by oracle/Makefile), and, when oracle/_ref
 * is present, against the reference functions directly on fresh random inputs.
 */
#ifndef NABWA_ORACLE_H
#define NABWA_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* FM-index in the reference's on-disk layout (bwt.h:43-59, bwtio.c:161-204). */
typedef struct {
	uint32_t primary, L2[5], seq_len;
	uint64_t n_words;      /* u32 words of interleaved Occ + BWT */
	uint32_t *bwt;
	int sa_intv;
	uint32_t n_sa;
	uint32_t *sa;          /* sa[0] is a sentinel (treated as -1), bwt.c:80 */
} orc_bwt_t;

typedef struct { int64_t offset; int32_t len; int32_t n_ambs; char name[64]; } orc_ann_t;
typedef struct { int64_t offset; int32_t len; char amb; } orc_hole_t;

typedef struct {
	orc_bwt_t bwt[2];      /* [0] forward text, [1] reversed text */
	int64_t l_pac;
	uint32_t seed;
	int n_seqs, n_holes;
	orc_ann_t *anns;
	orc_hole_t *holes;
	uint8_t *pac;          /* 2-bit packed, 4 bases/byte, MSB first (bwtaln.h:33) */
} orc_index_t;

/* gap_opt_t, same field order and size as the reference (bwtaln.h:143-153): 64 bytes. */
typedef struct {
	int s_mm, s_gapo, s_gape;
	int mode;
	int indel_end_skip, max_del_occ, max_entries;
	float fnr;
	int max_diff, max_gapo, max_gape;
	int max_seed_diff, seed_len;
	int n_threads;
	int max_top2;
	int trim_qual;
} orc_opt_t;

#define ORC_MODE_GAPE     0x01
#define ORC_MODE_COMPREAD 0x02
#define ORC_MODE_LOGGAP   0x04
#define ORC_MODE_NONSTOP  0x10

/* one hit, bit-identical to bwt_aln1_t (bwtaln.h:41-45): 16 bytes */
typedef struct { uint32_t info; uint32_t k, l; int32_t score; } orc_aln_t;
#define ORC_ALN_MM(x)   ((x).info & 0xff)
#define ORC_ALN_GAPO(x) ((x).info >> 8 & 0xff)
#define ORC_ALN_GAPE(x) ((x).info >> 16 & 0xff)
#define ORC_ALN_A(x)    ((x).info >> 24 & 1)

/* counters of the reference algorithm's memory touches (algorithmic bytes) */
typedef struct {
	uint64_t n_bucket;     /* Occ-bucket touches */
	uint64_t n_sa;         /* bwt_sa calls */
	uint64_t n_pop, n_push;
} orc_counters_t;

orc_index_t *orc_index_load(const char *prefix, int with_sa, int with_pac);
/* wrap caller-owned arrays (bench: index synthesised in memory) */
orc_index_t *orc_index_wrap(const uint32_t *bwt0, uint64_t nw0, const uint32_t *bwt1, uint64_t nw1);
void orc_index_free(orc_index_t *ix);

uint32_t orc_occ(const orc_bwt_t *b, uint32_t k, int c);
void orc_occ4(const orc_bwt_t *b, uint32_t k, uint32_t cnt[4]);
void orc_2occ4(const orc_bwt_t *b, uint32_t k, uint32_t l, uint32_t ck[4], uint32_t cl[4]);
uint32_t orc_sa(const orc_bwt_t *b, uint32_t k);
/* bwt_cal_width (bwtaln.c:52-76) of one pass: w_out / bid_out get len + 1 entries */
void orc_cal_width(const orc_bwt_t *rb, int len, const uint8_t *str, uint32_t *w_out, int32_t *bid_out);
int orc_maxdiff(int l, double err, double thres);
void orc_default_opt(orc_opt_t *o);

/* bwa_cal_sa_reg_gap over a flat batch; see oracle/ref_harness.c for the argument meaning.
 * n_threads > 1 splits the batch over pthreads (CPU baseline); results are independent of it.
 * Returns total hits or -1 when aln_cap is too small.  ctr may be NULL. */
long orc_cal_sa_reg_gap(const orc_index_t *ix, const orc_opt_t *opt, int n, const int64_t *off,
						const uint8_t *seq, const uint8_t *rseq, int per_read,
						int32_t *n_aln, orc_aln_t *aln_out, long aln_cap, int32_t *max_entries,
						int n_threads, orc_counters_t *ctr);

/* drand48 replica (glibc: X' = 0x5DEECE66D * X + 0xB mod 2^48; srand48 seeds X = seed<<16 | 0x330E) */
typedef struct { uint64_t x; } orc_rng_t;
void orc_srand48(orc_rng_t *r, long seed);
double orc_drand48(orc_rng_t *r);

/* SE record after hit choice, position lookup, refinement -- the fields a SAM/BAM line is made of */
typedef struct {
	int type, strand, n_mm, n_gapo, n_gape, score;
	uint32_t sa, pos;
	uint32_t c1, c2;
	int mapQ, seQ;
	int len, full_len, clip_len;
	int n_cigar; uint16_t cigar[64];
	int nm; char md[512];
	int n_multi;
	struct { uint32_t pos; int gap, mm, strand; int n_cigar; uint16_t cigar[64]; } multi[16];
	/* derived the way bwa_print_sam1 does (bwase.c:458-571) */
	int flag, seqid, nn; int64_t rpos; char xt;
} orc_se_t;

/* Full SE chain for ONE read, consuming the RNG stream in call order:
 * bwa_aln2seq_core -> bwa_cal_pac_pos_core (+multi) -> bwa_refine_gapped -> flag/XT logic. */
void orc_se_finish(const orc_index_t *ix, const orc_opt_t *opt, orc_rng_t *rng, int len, int full_len,
				   const uint8_t *seq_rev, const uint8_t *rseq, int n_aln, const orc_aln_t *aln,
				   int n_occ, orc_se_t *out);

/* aln_global_core + path -> cigar32 (len<<4|op).  Returns the score. */
int orc_global(const uint8_t *s1, int l1, const uint8_t *s2, int l2, int gap_open, int gap_ext, int gap_end,
			   const int *matrix, int row, int band, uint32_t *cig_out, int *n_cig);

int orc_pac2real(const orc_index_t *ix, int64_t pac_coor, int len, int *seqid);

#ifdef __cplusplus
}
#endif
#endif
