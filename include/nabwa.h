/* include/nabwa.h -- C ABI of libnabwa.so: the MI355X-native per-read alignment hot path of
 * mpieva/network-aware-bwa, as a drop-in behind the functions `bwa bam2bam` / `bwa worker`
 * call per record.  Plain pointers and sizes only.
 *
 * All entry points return 0 on success or a negative NABWA_E* code; none falls back to a
 * CPU implementation: without a usable GPU they fail with NABWA_ENODEV.
 *
 * The reference has no FFI/plugin layer (it is one static binary), so the "binding" a
 * maintainer adds is a direct call from bam2bam.c -- shown in INTEGRATION.md.
 */
#ifndef NABWA_H
#define NABWA_H
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NABWA_OK        0
#define NABWA_ENODEV   -1   /* no HIP device / HIP call failed (message via nabwa_last_error) */
#define NABWA_EINVAL   -2   /* bad argument or unsupported option block */
#define NABWA_EIO      -3   /* index file missing / malformed */
#define NABWA_ENOMEM   -4
#define NABWA_ECAP     -5   /* caller-provided output capacity too small (n_aln[] / *n_rows say how much is needed) */
#define NABWA_EHITS    -6   /* some reads have more hit rows than the device-side result rows can be grown to (NABWA_ALNCAP2 = 1024 rows, searched
                             again with 16 x, 256 x, 4096 x that within NABWA_HIT_GROW_GB = 8 GB): their n_aln is 0, every other read is resolved
                             and can be fetched */

/* gap_opt_t -- identical layout to the reference's (bwtaln.h:143-153, 64 bytes); it is the
 * block `bwa worker` receives over the wire (bam2bam.c:1260-1263). */
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
} nabwa_gap_opt_t;

#define NABWA_MODE_GAPE     0x01   /* BWA_MODE_* bits, bwtaln.h:132-141 */
#define NABWA_MODE_COMPREAD 0x02
#define NABWA_MODE_LOGGAP   0x04
#define NABWA_MODE_NONSTOP  0x10

/* bwt_aln1_t -- identical layout (bwtaln.h:41-45, 16 bytes): {n_mm:8,n_gapo:8,n_gape:8,a:1}, k, l, score */
typedef struct { uint32_t info; uint32_t k, l; int32_t score; } nabwa_aln1_t;

typedef struct nabwa_index nabwa_index_t;   /* both FM-indexes (+SA, +pac) resident in HBM */

const char *nabwa_last_error(void);
int nabwa_device_count(void);

/* gap_init_opt (bwtaln.c:19-35) */
void nabwa_gap_init_opt(nabwa_gap_opt_t *opt);
/* bwa_cal_maxdiff (bwtaln.c:37-49); host-side floating point, kept bit-compatible */
int nabwa_cal_maxdiff(int len, double err, double thres);

/* Replaces init_genome_index (bam2bam.c:844-858): reads <prefix>.bwt/.rbwt (and .sa/.rsa when
 * with_sa, .pac when with_pac) in the reference's on-disk format (bwtio.c:161-204), uploads them
 * to `device` and re-packs the Occ arrays into 64-byte buckets there. */
int nabwa_index_load(const char *prefix, int device, int with_sa, int with_pac, nabwa_index_t **out);

/* Same, from arrays already in memory (is_device != 0: device pointers on `device`).
 * bwt0/bwt1: the content of a .bwt/.rbwt file as u32 words (5 header words + Occ-interleaved BWT).
 * sa0/sa1: content of .sa/.rsa files as u32 words (7 header words + samples), or NULL. */
int nabwa_index_from_arrays(int device, int is_device, const uint32_t *bwt0, uint64_t n_words0,
							const uint32_t *bwt1, uint64_t n_words1, const uint32_t *sa0, uint64_t n_sa_words0,
							const uint32_t *sa1, uint64_t n_sa_words1, nabwa_index_t **out);
void nabwa_index_destroy(nabwa_index_t *ix);
uint32_t nabwa_index_seq_len(const nabwa_index_t *ix, int which);
uint64_t nabwa_index_device_bytes(const nabwa_index_t *ix);

/* Batch form of bwa_cal_sa_reg_gap (bwtaln.c:93-142; callers bam2bam.c:616,676, bwtaln.c:235).
 *  seq/rseq : concatenated bwa_seq_t.seq / .rseq codes (0-3, 4=N) of n reads; off[i]..off[i+1]
 *             delimit read i (seq = read reversed, rseq = its reverse complement, bwaseqio.c:294-297).
 *  per_read : !=0 gives every read the option block the reference derives when called with
 *             n_seqs==1 (bam2bam); 0 derives it once from the longest read of the batch (bwa aln).
 *  n_aln[i], max_entries[i] : as bwa_seq_t.n_aln / .max_entries.
 *  aln_out  : rows of all reads back to back in read order (capacity aln_cap rows);
 *             *n_rows receives the total.  NABWA_ECAP if aln_cap is too small (n_aln[] is still valid). */
int nabwa_cal_sa_reg_gap(nabwa_index_t *ix, const nabwa_gap_opt_t *opt, int n, const int64_t *off,
						 const uint8_t *seq, const uint8_t *rseq, int per_read,
						 int32_t *n_aln, nabwa_aln1_t *aln_out, int64_t aln_cap, int64_t *n_rows,
						 int32_t *max_entries);

/* Device-resident variant for pipelines and for bench.py: upload once, run many times.
 * A batch owns the device copies of the reads, the per-lane search scratch and the outputs. */
typedef struct nabwa_batch nabwa_batch_t;
int nabwa_batch_create(nabwa_index_t *ix, const nabwa_gap_opt_t *opt, int n, const int64_t *off,
					   const uint8_t *seq, const uint8_t *rseq, int per_read, nabwa_batch_t **out);
/* enqueue the FM search of the whole batch on the batch's HIP stream (asynchronous) */
int nabwa_batch_run(nabwa_batch_t *b);
/* wait for completion; *n_second_pass = the number of reads the first pass handed on to kernel D (deep searches) */
int nabwa_batch_sync(nabwa_batch_t *b, int *n_second_pass);
/* HIP-event time of the most recent run of the dominant kernel (fm_search, first pass), ms */
float nabwa_batch_last_kernel_ms(nabwa_batch_t *b);
/* same for the width kernel (fm_width) that precedes it */
float nabwa_batch_last_width_ms(nabwa_batch_t *b);
/* same for kernel D (fm_deep: the reads the first pass handed on, one search per wavefront); 0 when it did not run */
float nabwa_batch_last_deep_ms(nabwa_batch_t *b);
/* untimed instrumented run: Occ-bucket touches the reference algorithm performs on this batch
 * (1 per bwt_occ/bwt_occ4 body, 1 per same-block bwt_2occ/bwt_2occ4), split into
 * those of bwt_match_gap (search kernel) and of the bwt_cal_width passes (width kernel) */
int nabwa_batch_count_touches(nabwa_batch_t *b, uint64_t *n_bucket, uint64_t *n_bucket_width);
int nabwa_batch_fetch(nabwa_batch_t *b, int32_t *n_aln, nabwa_aln1_t *aln_out, int64_t aln_cap, int64_t *n_rows,
					  int32_t *max_entries);
/* tests: what kernel W computes (the four bwt_cal_width passes, bwtaln.c:52-76,123-130) for reads [first, first + n): rows of
 * max_len + 1 widths and bound bytes (min(bid, 127) | (w[p-1] == w[p]) << 7) per read and strand; seed bounds seed_len + 1 wide */
int nabwa_batch_width_records(nabwa_batch_t *b, int first, int n, uint32_t *w_out, uint8_t *bid_out, uint8_t *seed_bid_out);
/* order-independent 64-bit checksum of (read id, row index, row) over all hits, computed on the device */
int nabwa_batch_checksum(nabwa_batch_t *b, uint64_t *sum, int64_t *n_rows);
void nabwa_batch_destroy(nabwa_batch_t *b);

/* ---- paired-end host pieces (config 3) --------------------------------------------------------- */
/* isize_info_t without the histogram pointer (bwape.h:16-20) */
typedef struct { double avg, std, ap_prior; uint32_t low, high, high_bayesian; } nabwa_isize_t;
/* infer_isize_hist (insert_size.c:50-139) on a histogram of 100000 u16 bins; 0 = usable, -1 = not (fields as the
 * reference leaves them: avg = std = -1, bounds 0) */
int nabwa_isize_infer(const uint16_t *hist, double ap_prior, int64_t L, nabwa_isize_t *out);
/* improve_isize_est (insert_size.c:141-165): the bin a record adds one to, or -1 */
int nabwa_isize_bin(int kind, int mapq0, int mapq1, uint32_t pos0, int len0, uint32_t pos1, int len1);
/* the bwa_seq_t fields pairing() reads and writes (bwape.c:180-293) */
typedef struct { uint32_t pos; int32_t strand, mapQ, seQ, len, full_len, n_mm, n_gapo, n_gape, score, extra_flag; } nabwa_pe_end_t;
/* pairing (bwape.c:180-293; caller finish_pair, bam2bam.c:768): hits[i] = text position << 32 | hit row << 1 | end
 * for every position of every hit row of both ends (sorted in place); returns the number of moved ends with mapQ > 0 */
int nabwa_pairing(nabwa_pe_end_t p[2], int n_hits, uint64_t *hits, const nabwa_aln1_t *rows0, const nabwa_aln1_t *rows1,
				  int max_isize, int s_mm, const nabwa_isize_t *ii);

/* ---- the reference's own record struct ------------------------------------------------------ */
/* bwa_seq_t (bwtaln.h:64-90), 200 bytes; bit-fields kept as the words the compiler packs them into:
 * bits0 = len:20 | strand:1 | type:2 | dummy:1 | extra_flag:8 ; bits1 = n_mm:8 | n_gapo:8 | n_gape:8 | mapQ:8 ;
 * c1c2seq = c1:28 | c2:28 | seQ:8 ; lenbits = full_len:20 | nm:12 */
typedef struct {
	char *name;
	uint8_t *seq, *rseq, *qual;
	uint32_t bits0, bits1;
	int32_t score, clip_len;
	int32_t n_aln; int32_t pad0;
	nabwa_aln1_t *aln;
	int32_t n_multi; int32_t pad1;
	void *multi;
	uint32_t sa, pos;
	uint64_t c1c2seq;
	int32_t n_cigar; int32_t pad2;
	uint16_t *cigar;
	int32_t tid;
	char bc[64];
	uint32_t lenbits;
	char *md;
	int32_t max_entries; int32_t pad3;
} nabwa_bwa_seq_t;

/* Drop-in for bwa_cal_sa_reg_gap(bwt, n_seqs, seqs, opt) (bwtaln.h:187; callers bam2bam.c:616,676, bwtaln.c:235)
 * on the reference's own records. */
int nabwa_bwa_cal_sa_reg_gap(nabwa_index_t *ix, int n_seqs, nabwa_bwa_seq_t *seqs, const nabwa_gap_opt_t *opt);

/* The encoding half of bam1_to_seq (bwaseqio.c:272-307) incl. seq_reverse (:91-108) and bwa_trim_read (:110-123):
 * codes = bases of the BAM record (0-3, 4 = N), qual = phred values or NULL; returns the (trimmed) length and fills
 * seq_out (read reversed) and rseq_out (its complement, i.e. the reverse complement of the read). */
int nabwa_encode_read(int full_len, const uint8_t *codes, const uint8_t *qual, int reverse, int trim_qual, int is_comp,
					  uint8_t *seq_out, uint8_t *rseq_out);

/* Batch form of bwt_sa (bwt.c:72-81; callers bam2bam.c:635-636,752,761,786, bwase.c:146,151):
 * sa_out[i] = SA value of row k[i] in index `which[i]` (0 forward, 1 reversed text). */
int nabwa_sa_lookup(nabwa_index_t *ix, int n, const uint8_t *which, const uint32_t *k, uint32_t *sa_out);

/* Batch form of aln_global_core + aln_path2cigar32 (stdaln.c:345-525, :1009-1039; caller
 * refine_gapped_core, bwase.c:212): task i aligns ref[ref_off[i]..ref_off[i+1]) (seq1) with
 * qry[qry_off[i]..qry_off[i+1]) (seq2), codes 0-4, under {gap_open, gap_ext, gap_end, 5x5 matrix,
 * band_width} (AlnParam, stdaln.h:86-95).  cigar32 rows are max_cigar wide (len<<4 | op);
 * n_cigar[i] > max_cigar means the row was truncated. */
int nabwa_global_align(int device, int n, const int64_t *ref_off, const uint8_t *ref, const int64_t *qry_off,
					   const uint8_t *qry, int gap_open, int gap_ext, int gap_end, const int *matrix25, int band,
					   int32_t *score, int32_t *n_cigar, uint32_t *cigar32, int max_cigar);

/* The alignment entry points keep their device working memory (score rows, traceback matrices) between calls; this frees it. */
void nabwa_dp_scratch_release(int device);

/* Batch form of aln_extend_core (stdaln.c:862-1007): left-anchored extension seeded with G0[i], then the
 * path by global alignment of the two prefixes (gap_end = -1, band doubled until the scores agree or it
 * exceeds the prefix lengths).  score[i] as the reference returns it (<= 0: no extension, n_cigar 0). */
int nabwa_extend_align(int device, int n, const int64_t *ref_off, const uint8_t *ref, const int64_t *qry_off,
					   const uint8_t *qry, int gap_open, int gap_ext, const int *matrix25, int band, const int32_t *G0,
					   int32_t *score, int32_t *n_cigar, uint32_t *cigar32, int max_cigar);

/* Batch form of aln_local_core (stdaln.c:529-761; caller bwa_sw_core, bwape.c:456, mate rescue) with
 * _thres = thres > 0: forward and reverse Smith-Waterman passes, then the path by global alignment of the
 * sub-matrix (gap_end = -1, doubling band).  coords rows: start_i, start_j, end_i, end_j (1-based);
 * subo (may be NULL): sub-optimal score as stdaln.c:700-709. */
int nabwa_local_align(int device, int n, const int64_t *ref_off, const uint8_t *ref, const int64_t *qry_off,
					  const uint8_t *qry, int gap_open, int gap_ext, const int *matrix25, int band, int thres,
					  int32_t *score, int32_t *coords, int32_t *subo, int32_t *n_cigar, uint32_t *cigar32, int max_cigar);

/* ---- single-end finishing chain: everything between the FM search and the BAM record -------- */
#define NABWA_MAX_CIGAR 64
#define NABWA_MAX_MD    512
#define NABWA_MAX_MULTI 16
typedef struct { uint32_t pos; int32_t gap, mm, strand, n_cigar; uint16_t cigar[NABWA_MAX_CIGAR]; } nabwa_multi_t;
typedef struct {
	int32_t type, strand, n_mm, n_gapo, n_gape, score;     /* bwa_seq_t fields of the same names (bwtaln.h:64-90) */
	uint32_t sa, pos, c1, c2;
	int32_t mapQ, seQ, len, full_len, clip_len;
	int32_t n_cigar; uint16_t cigar[NABWA_MAX_CIGAR];       /* bwa_cigar_t: op<<14 | len (bwtaln.h:47-56) */
	int32_t nm; char md[NABWA_MAX_MD];
	int32_t n_multi; nabwa_multi_t multi[NABWA_MAX_MULTI];
	/* what bwa_print_sam1 / bwa_update_bam1 derive (bwase.c:458-571, bam2bam.c:430-593) */
	int32_t flag, seqid, nn; int64_t rpos; char xt;
} nabwa_se_t;

/* Annotation side of the index (replaces bns_restore + bwt_restore_pac, bntseq.c:88-148): <prefix>.ann,
 * .amb and .pac are read into host memory and attached to the index. */
int nabwa_index_attach_reference(nabwa_index_t *ix, const char *prefix);
/* what bns_restore read (bntseq.c:88-139): contigs (name, offset in the concatenated reference, length), total length, srand48 seed */
int nabwa_index_n_contigs(const nabwa_index_t *ix);
int nabwa_index_contig(const nabwa_index_t *ix, int i, char *name, int name_cap, int64_t *offset, int32_t *len);
int nabwa_index_reference_info(const nabwa_index_t *ix, int64_t *l_pac, uint32_t *seed);
/* the same from memory (bntseq_t: contigs with offsets and lengths, bntamb1_t holes, the .pac bytes) */
int nabwa_index_set_reference(nabwa_index_t *ix, int64_t l_pac, uint32_t seed, int n_seqs, const char *const *names,
							  const int64_t *offsets, const int32_t *lens, int n_holes, const int64_t *hole_off,
							  const int32_t *hole_len, const char *hole_amb, const uint8_t *pac);

/* bwa_aln2seq_core (bwase.c:19-95) -> bwa_cal_pac_pos_core + multi-hit positions (bwase.c:139-181,
 * bam2bam.c:629-637) -> bwa_refine_gapped (bwase.c:356-423: gap refinement, MD/NM, trimmed-tail clip) for
 * n single-end reads IN RECORD ORDER.  The hit choice consumes the caller's drand48 stream: *rng48 is the
 * 48-bit state (srand48(seed) == ((uint64_t)seed << 16) | 0x330E) and is updated, so that successive
 * batches continue the reference's process-global stream.  SA lookups and the gap-refinement
 * DP run on the GPU in batches; the rest is host bookkeeping.
 * full_len[i] >= len: untrimmed read length (bwa_seq_t.full_len).  n_occ: max_occ_se (bam2bam.c:629; <= 15). */
int nabwa_se_finish(nabwa_index_t *ix, const nabwa_gap_opt_t *opt, int n, const int64_t *off, const uint8_t *seq,
					const uint8_t *rseq, const int32_t *full_len, const int32_t *n_aln, const nabwa_aln1_t *aln,
					int n_occ, uint64_t *rng48, nabwa_se_t *out);

/* The two halves of nabwa_se_finish on their own, as bam2bam's two passes need them:
 * nabwa_se_posn   = posn_singleton (bam2bam.c:622-641): hit choice on the caller's drand48 stream, bwt_sa batch, mapQ;
 * nabwa_se_refine = bwa_refine_gapped (bwase.c:356-423; finish_singleton, bam2bam.c:643-651) on positioned records + the
 *                   flag / contig / XT fields bwa_update_bam1 derives.  NABWA_ECAP if an MD string exceeds NABWA_MAX_MD. */
int nabwa_se_posn(nabwa_index_t *ix, const nabwa_gap_opt_t *opt, int n, const int64_t *off, const int32_t *full_len,
				  const int32_t *n_aln, const nabwa_aln1_t *aln, int n_occ, uint64_t *rng48, nabwa_se_t *out);
int nabwa_se_refine(nabwa_index_t *ix, int n, const int64_t *off, const uint8_t *seq, const uint8_t *rseq, nabwa_se_t *inout);
/* nabwa_se_posn with a hit-list bound per read (singletons: max_occ_se, ends of pairs: 0 -- one drand48 stream for a mixed file) */
int nabwa_se_posn_v(nabwa_index_t *ix, const nabwa_gap_opt_t *opt, int n, const int64_t *off, const int32_t *full_len,
					const int32_t *n_aln, const nabwa_aln1_t *aln, const uint8_t *n_occ_v, uint64_t *rng48, nabwa_se_t *out);

/* ---- paired-end chain (config 3) ------------------------------------------------------------ */
/* pe_opt_t (bwtaln.h:158-164), same layout; defaults as bwa_init_pe_opt (bwape.c:27-41) */
typedef struct {
	int32_t max_isize, force_isize, max_occ, max_occ_se, n_multi, N_multi, type, is_sw, is_preload;
	double ap_prior;
} nabwa_pe_opt_t;
void nabwa_pe_opt_default(nabwa_pe_opt_t *po);

/* one end of a pair: the single-end record plus what bwa_update_bam1 derives from the mate (bam2bam.c:430-525).
 * se.flag is the final SAM flag; extra_flag is bwa_seq_t.extra_flag (paired / read1 / read2 / proper pair). */
typedef struct {
	nabwa_se_t se;
	int32_t extra_flag, m_seqid, am;
	int32_t mapQ_paired;                                   /* se.mapQ as pairing left it; se.mapQ itself is cleared when the hit bridges two contigs (bam2bam.c:453-458) */
	int64_t m_rpos, isize;                                 /* mate position (1-based, on contig m_seqid), template length */
} nabwa_pe_t;

/* posn_pair (bam2bam.c:683-703) for n_pairs pairs: bwa_aln2seq + bwa_cal_pac_pos_core per end, IN RECORD ORDER
 * (pair 0 end 0, pair 0 end 1, pair 1 end 0, ...) on the caller's drand48 stream.  Reads, hit rows and records
 * are interleaved: index 2*pair + end.  Afterwards the caller bins insert sizes (nabwa_isize_bin on pos / len /
 * mapQ of the two ends) and, once all batches are in, calls nabwa_isize_infer -- the barrier of bam2bam. */
int nabwa_pe_posn(nabwa_index_t *ix, const nabwa_gap_opt_t *opt, int n_pairs, const int64_t *off, const int32_t *full_len,
				  const int32_t *n_aln, const nabwa_aln1_t *aln, uint64_t *rng48, nabwa_pe_t *out);

/* finish_pair (bam2bam.c:705-811) for the same pairs: hit enumeration (bwt_sa, GPU batch) + pairing (bwape.c:180-293),
 * multi-hit lists, mate rescue (bwa_paired_sw1 / bwa_sw_core, bwape.c:433-633; local alignments as one GPU batch),
 * bwa_refine_gapped on both ends (global alignments as one GPU batch), MD / NM, flags and mate fields.
 * ii: the read group's insert-size estimate; all zeros = none (null_ii, bam2bam.c:715).
 * n_tot / n_mapped (may be NULL): the counters of bwa_paired_sw1, [0] discordant pairs, [1] singletons. */
/* improve_isize_est (insert_size.c:141-165) for the positioned pairs of one batch: hist = 100000 uint16_t bins, wrapping as the
 * reference's do; call between nabwa_pe_posn and nabwa_isize_infer */
int nabwa_isize_add_pairs(int n_pairs, const nabwa_pe_t *recs, uint16_t *hist);
int nabwa_pe_finish(nabwa_index_t *ix, const nabwa_gap_opt_t *opt, const nabwa_pe_opt_t *popt, const nabwa_isize_t *ii,
					int n_pairs, const int64_t *off, const uint8_t *seq, const uint8_t *rseq, const int32_t *n_aln,
					const nabwa_aln1_t *aln, nabwa_pe_t *inout, uint64_t n_tot[2], uint64_t n_mapped[2]);
/* finish_pair's per-file position cache (kh_64_t *my_hash, bam2bam.c:707,741-757; one per pass 2 in `bam2bam -t 1`, :1186-1203): hit rows of
 * MIN_HASH_WIDTH = 1000 suffixes or more get their text positions once per file, under the key (k, l) alone -- later reads with the same
 * row take the positions of the first read that brought it, computed with THAT read's strand and length.  Pass the same cache to every
 * batch of a file, batches in input order, to get what the sequential reference writes; NULL = every row on its own (a cold cache). */
typedef struct nabwa_poscache nabwa_poscache_t;
nabwa_poscache_t *nabwa_poscache_create(void);
void nabwa_poscache_destroy(nabwa_poscache_t *c);
int64_t nabwa_poscache_size(const nabwa_poscache_t *c);      /* rows entered so far */
int nabwa_pe_finish_cached(nabwa_index_t *ix, const nabwa_gap_opt_t *opt, const nabwa_pe_opt_t *popt, const nabwa_isize_t *ii,
						   int n_pairs, const int64_t *off, const uint8_t *seq, const uint8_t *rseq, const int32_t *n_aln,
						   const nabwa_aln1_t *aln, nabwa_pe_t *inout, uint64_t n_tot[2], uint64_t n_mapped[2], nabwa_poscache_t *cache);

/* ---- the same phases on the reference's own records (bwa_seq_t), one call per phase and batch ---------------------
 * Each replaces the per-record function of the same role in bam2bam.c; fields are left as that function leaves them, and
 * multi / multi[].cigar / cigar / md are malloc'd for bwa_free_read_seq1 (bwaseqio.c:253-261) to free.
 *   nabwa_bwa_posn_se       posn_singleton (bam2bam.c:622-641): bwa_aln2seq_core(.., 1, max_occ_se), bwa_cal_pac_pos_core, multi positions
 *   nabwa_bwa_refine_gapped bwa_refine_gapped(bns, n, seqs, pac, ntbns) (bwase.h:16; bam2bam.c:649,799-800), seq un-reversed as there
 *   nabwa_bwa_posn_pe       posn_pair (bam2bam.c:683-703); records interleaved, 2 * pair + end
 *   nabwa_bwa_finish_pe     finish_pair up to bwa_update_bam1 (bam2bam.c:705-800): pairing (bwape.h:46), bwa_paired_sw1 (bwape.h:49),
 *                           multi lists, bwa_refine_gapped on both ends
 * rng48: the drand48 state (srand48(seed) == ((uint64_t)seed << 16) | 0x330E), consumed in record order. */
int nabwa_bwa_posn_se(nabwa_index_t *ix, const nabwa_gap_opt_t *opt, int max_occ_se, int n, nabwa_bwa_seq_t *seqs, uint64_t *rng48);
int nabwa_bwa_refine_gapped(nabwa_index_t *ix, int n, nabwa_bwa_seq_t *seqs);
int nabwa_bwa_posn_pe(nabwa_index_t *ix, const nabwa_gap_opt_t *opt, int n_pairs, nabwa_bwa_seq_t *seqs, uint64_t *rng48);
int nabwa_bwa_finish_pe(nabwa_index_t *ix, const nabwa_gap_opt_t *opt, const nabwa_pe_opt_t *popt, const nabwa_isize_t *ii,
						int n_pairs, nabwa_bwa_seq_t *seqs, uint64_t n_tot[2], uint64_t n_mapped[2]);
/* the same with finish_pair's my_hash argument (bam2bam.c:707): one cache per pass 2 of a file */
int nabwa_bwa_finish_pe_cached(nabwa_index_t *ix, const nabwa_gap_opt_t *opt, const nabwa_pe_opt_t *popt, const nabwa_isize_t *ii,
							   int n_pairs, nabwa_bwa_seq_t *seqs, uint64_t n_tot[2], uint64_t n_mapped[2], nabwa_poscache_t *cache);

/* ---- the batching front-end: BAM records in, BAM records out (what sits behind `bwa bam2bam` / `bwa worker`) --------
 * A batch = n_rec records as they stand in an uncompressed BAM stream (u32 block_size + block), in file order, mates adjacent.
 *   create : read_bam_pair's record logic (bwaseqio.c:340-494: singleton / pair, QC flag over mates, erase_unwanted_tags) and
 *            bam1_to_seq (bwaseqio.c:272-307)
 *   pass1  : pair_aln + pair_posn + improve_isize_est per logical record, in record order on the caller's drand48 stream
 *            (bam2bam.c:1143-1176, 608-703; insert_size.c:141-165)
 *   pass2  : pair_finish incl. bwa_update_bam1 (bam2bam.c:1178-1216, 643-658, 705-811, 430-593), each read group with its
 *            own insert-size estimate
 *   output : the records (uncompressed BAM bytes; BGZF and the header are the caller's, as they are bam2bam's own bgzf.c)
 * Between the passes of ALL batches: nabwa_isize_table_infer_all (infer_all_isizes, insert_size.c:167-173). */
typedef struct nabwa_isize_table nabwa_isize_table_t;
typedef struct nabwa_bam_batch nabwa_bam_batch_t;
nabwa_isize_table_t *nabwa_isize_table_create(double ap_prior, int64_t genome_len);
void nabwa_isize_table_destroy(nabwa_isize_table_t *t);
int nabwa_isize_table_infer_all(nabwa_isize_table_t *t);
/* the estimate of a read group; returns 1 and all zeros (null_ii, bam2bam.c:106) when it has none */
int nabwa_isize_table_get(const nabwa_isize_table_t *t, const char *rg, nabwa_isize_t *out);
/* histograms of another shard's pass 1 added in (N GPUs: the one cross-shard reduction of the pipeline) */
int nabwa_isize_table_merge(nabwa_isize_table_t *t, const nabwa_isize_table_t *other);
/* encode_iinfo / decode_iinfo (insert_size.c:185-213): the blob a `bwa worker` is sent; encode returns the size needed */
int64_t nabwa_isize_table_encode(const nabwa_isize_table_t *t, uint8_t *out, int64_t cap);
int nabwa_isize_table_decode(nabwa_isize_table_t *t, const uint8_t *in, int64_t n);
int nabwa_bam_batch_create(nabwa_index_t *ix, const nabwa_gap_opt_t *opt, const nabwa_pe_opt_t *popt, int n_rec,
						   const uint8_t *in, const int64_t *in_off, nabwa_bam_batch_t **out);
/* the same with bam2bam's record-handling switches (bam2bam.c:96-101,1988-1993):
 *   BROKEN_INPUT    --broken-input   : read_bam_pair's allow_broken (bwaseqio.c:345-410): wrong read 1 / read 2 flags of two reads of
 *                                      one name are set right, a paired read followed by another name (or by nothing) is discarded
 *   DROP_ALIGNED    --drop-aligned   : logical records with a read that is already mapped are left out (bwaseqio.c:466-474)
 *   SKIP_DUPLICATES --skip-duplicates: logical records with a read flagged as duplicate (0x400) are neither aligned nor counted
 *                                      for the insert size; they come out as they came in, less the erased tags (unique(), bam2bam.c:595-606)
 *   DEBUG           --debug-bam      : YQ:i = the most entries the search of the read held (bam2bam.c:433)
 *   ONLY_ALIGNED    --only-aligned   : output leaves out logical records with a read that stayed unmapped (pair_print_bam, bam2bam.c:911-925) */
#define NABWA_BAM_BROKEN_INPUT    1u
#define NABWA_BAM_DROP_ALIGNED    2u
#define NABWA_BAM_SKIP_DUPLICATES 4u
#define NABWA_BAM_DEBUG           8u
#define NABWA_BAM_ONLY_ALIGNED    16u
#define NABWA_BAM_ALL_FLAGS       31u
int nabwa_bam_batch_create_ex(nabwa_index_t *ix, const nabwa_gap_opt_t *opt, const nabwa_pe_opt_t *popt, uint32_t flags, int n_rec,
							  const uint8_t *in, const int64_t *in_off, nabwa_bam_batch_t **out);
/* optional, before pass 1 and from any thread: the FM search of the batch (the part of pass 1 that needs neither the random stream
 * nor the batches before it) on the GPU of the index the batch was created with; pass 1 then goes on from its rows.  This is how
 * one process keeps several GPUs busy: batches are dealt to index replicas, searched as they come, and passed in input order. */
int nabwa_bam_batch_search(nabwa_bam_batch_t *b);
int nabwa_bam_batch_pass1(nabwa_bam_batch_t *b, uint64_t *rng48, nabwa_isize_table_t *tab);
/* pass 2 keeps finish_pair's position cache (above) in the table: the batches of a file pass in input order, as the records of
 * `bam2bam -t 1` do */
int nabwa_bam_batch_pass2(nabwa_bam_batch_t *b, const nabwa_isize_table_t *tab, uint64_t n_tot[2], uint64_t n_mapped[2]);
int nabwa_bam_batch_output(const nabwa_bam_batch_t *b, uint8_t *out, int64_t cap, int64_t *out_off, int64_t *n_bytes);
int nabwa_bam_batch_counts(const nabwa_bam_batch_t *b, int *n_records, int *n_logical);
/* per logical record: 1 (a single read) or 2 (a pair); n_logical entries */
int nabwa_bam_batch_kinds(const nabwa_bam_batch_t *b, uint8_t *kind_out);
void nabwa_bam_batch_destroy(nabwa_bam_batch_t *b);

/* ---- 0MQ worker compatibility: the wire record, and a worker core that is independent of the transport ----------
 * `bwa bam2bam -p PORT` hands logical records to `bwa worker` processes as 0MQ messages and takes them back one phase further
 * (run_worker_thread, bam2bam.c:1387-1442); the same bytes, with a u32 length in front, are the records of its temporary file
 * between the passes (pair_print_custom / read_pair_custom, bam2bam.c:1099-1135).  One message (msg_init_from_pair, bam2bam.c:951-1006;
 * inverse pair_init_from_msg, :1040-1097):
 *    u64 recno (little endian) . u8 kind (0 end marker, 1 single read, 2 pair) . u8 phase (0 pristine, 1 aligned, 2 positioned, 3 finished)
 *    per read:  bam1_core_t as the HOST lays it out, 32 bytes (bamlite.h:44-53: word 2 = bin | qual << 16 | l_qname << 24, word 3 =
 *               flag | n_cigar << 16 -- not the order of a BAM file) . i32 data_len . data (name, CIGAR, bases, qualities, tags)
 *      phase positioned:  u8 strand << 4 | type . u8 n_mm, n_gapo, n_gape, seQ, mapQ . i32 len, clip_len, score, sa, c1, c2, pos, n_multi .
 *                         n_multi raw bwt_multi1_t of 16 bytes (u32 pos . u32 n_cigar:15 | gap:8 | mm:8 | strand:1 . a dead pointer)
 *      phases aligned, positioned:  i32 max_entries . i32 n_aln . n_aln raw bwt_aln1_t of 16 bytes
 * Parity unpinned: libzmq is absent from the image and bam2bam.c cannot be compiled, so no byte of a reference message exists to compare
 * with; the codec follows the cited lines, and tests/test_wire.py builds messages by hand from them. */
#define NABWA_KIND_EOF 0
#define NABWA_KIND_SINGLE 1
#define NABWA_KIND_PAIR 2
#define NABWA_PHASE_PRISTINE 0
#define NABWA_PHASE_ALIGNED 1
#define NABWA_PHASE_POSITIONED 2
#define NABWA_PHASE_FINISHED 3
typedef struct {
	uint8_t core[32];                       /* bam1_core_t, host layout */
	int32_t data_len; const uint8_t *data;  /* bam1_t.data */
	uint8_t strand, type, n_mm, n_gapo, n_gape, seQ, mapQ;                 /* positioned */
	int32_t len, clip_len, score; uint32_t sa, c1, c2, pos;
	int32_t n_multi; const uint8_t *multi;                                 /* n_multi x 16 raw bytes (may be unaligned) */
	int32_t max_entries, n_aln; const uint8_t *aln;                        /* aligned, positioned: n_aln x 16 raw bytes (may be unaligned) */
} nabwa_wire_read_t;
typedef struct { uint64_t recno; uint8_t kind, phase; nabwa_wire_read_t read[2]; } nabwa_wire_rec_t;
/* bytes the message of r takes */
int64_t nabwa_wire_size(const nabwa_wire_rec_t *r);
/* msg_init_from_pair: writes the message, returns its size; NABWA_ECAP if cap is too small, NABWA_EINVAL for a kind / phase that does not exist */
int64_t nabwa_wire_encode(const nabwa_wire_rec_t *r, uint8_t *out, int64_t cap);
/* pair_init_from_msg: the pointers of *out point into msg.  NABWA_EINVAL unless the message is consumed exactly (the reference exits there) */
int nabwa_wire_decode(const uint8_t *msg, int64_t len, nabwa_wire_rec_t *out);
/* a record as it stands in a BAM stream (u32 block_size, 32 bytes of core in file order, data) <-> core in host layout + data */
void nabwa_wire_core_from_bam(const uint8_t bam_core[32], uint8_t wire_core[32]);
void nabwa_wire_core_to_bam(const uint8_t wire_core[32], uint8_t bam_core[32]);
/* the reply to a worker's hello (run_config_service, bam2bam.c:1255-1266; read at :2260-2274): gap_opt_t . pe_opt_t . index prefix, not terminated */
int64_t nabwa_wire_config_encode(const nabwa_gap_opt_t *opt, const nabwa_pe_opt_t *popt, const char *prefix, uint8_t *out, int64_t cap);
int nabwa_wire_config_decode(const uint8_t *msg, int64_t len, nabwa_gap_opt_t *opt, nabwa_pe_opt_t *popt, char *prefix, int prefix_cap);

/* The state pass 1 leaves a batch in, read by read, and the way back: what lets a positioned record leave the process (the temporary
 * file between the passes, a worker's reply) and come back into a fresh batch made from the same records.
 *   nabwa_bam_batch_positioned : after pass 1; fills the positioned and aligned parts of out[0 .. n_records) (core / data are left alone),
 *                                pointers into the batch, valid until it is destroyed
 *   nabwa_bam_batch_restore    : on a batch just created from the same records: takes the state from in[0 .. n_records) instead of
 *                                searching and positioning; the batch is then where pass 1 would have left it (no random numbers drawn, no
 *                                insert sizes counted).  NABWA_EINVAL if a read's length disagrees with the batch's (other trimming options). */
int nabwa_bam_batch_positioned(nabwa_bam_batch_t *b, nabwa_wire_read_t *out);
int nabwa_bam_batch_restore(nabwa_bam_batch_t *b, const nabwa_wire_read_t *in);

/* The worker's side of the exchange without a socket in sight: messages in, messages out.
 *   pristine / aligned records : pair_aln + pair_posn (bam2bam.c:1414-1416) on the worker's own drand48 stream in arrival order
 *                                (srand48(bns->seed) at start, bam2bam.c:2284) -> positioned
 *   positioned records         : pair_finish with the insert-size estimates last set (bam2bam.c:1419) -> finished; without estimates the
 *                                record goes back as it came and counts as a failure (1024 of them end the worker, :1428-1433)
 *   finished records, end markers: go back as they came
 * Records are gathered into batches for the GPU; replies leave in arrival order.  A record sent twice (the master's resend loop,
 * bam2bam.c:1587-1596) is simply worked on twice, as the reference's workers do -- nothing is remembered between messages but the
 * random stream, the estimates and finish_pair's position cache. */
typedef struct nabwa_worker nabwa_worker_t;
int nabwa_worker_create(nabwa_index_t *ix, const nabwa_gap_opt_t *opt, const nabwa_pe_opt_t *popt, nabwa_worker_t **out);
void nabwa_worker_destroy(nabwa_worker_t *w);
/* the insert-size broadcast (`\2` + blob, handle_broadcast, bam2bam.c:2079-2097) or the reply to `\1nodename` (:2286-2299): decode_iinfo's blob */
int nabwa_worker_set_isize(nabwa_worker_t *w, const uint8_t *blob, int64_t n);
typedef int (*nabwa_send_fn)(void *ctx, const uint8_t *msg, int64_t len);                               /* 0 = sent */
/* returns 1 with a message (valid until the next call), 0 when none came within timeout_ms, -1 when the transport is closed */
typedef int (*nabwa_recv_fn)(void *ctx, const uint8_t **msg, int64_t *len, int timeout_ms);
/* one batch of messages: every one is answered through send, in the order given */
int nabwa_worker_process(nabwa_worker_t *w, int n_msg, const uint8_t *const *msgs, const int64_t *lens, nabwa_send_fn send, void *ctx);
/* bwa_worker_core's loop (bam2bam.c:2099-2176) over a transport: gathers up to max_batch messages (waiting linger_ms for more once one
 * is there), processes them, goes on; returns NABWA_OK when nothing came for idle_timeout_ms (the reference: 90 s) or the transport closed,
 * NABWA_EIO after 1024 positioned records without estimates */
typedef struct { int32_t max_batch, linger_ms, idle_timeout_ms; } nabwa_worker_opt_t;
int nabwa_worker_core(nabwa_worker_t *w, nabwa_recv_fn recv, nabwa_send_fn send, void *ctx, const nabwa_worker_opt_t *wo);
/* counters: records positioned, finished, bounced (no estimates), passed through */
void nabwa_worker_counts(const nabwa_worker_t *w, uint64_t out[4]);

/* Read-back of the index parts derived at load time (tests): what 0 = full SA, 1 = inverse SA, 2 = text bases (one per
 * word), 3 = interval-table entries {k, l} of the last level (two words per key), 4 = the table's depth T (one word). */
int nabwa_index_export(const nabwa_index_t *ix, int which, int what, uint64_t first, uint64_t n, uint32_t *out);

/* Rank primitives for tests: Occ of all four bases at rows k[i] (bwt_occ4, bwt.c:159-176). */
int nabwa_occ4(nabwa_index_t *ix, int which, int n, const uint32_t *k, uint32_t *cnt_out /* n x 4 */);

#ifdef __cplusplus
}
#endif
#endif
