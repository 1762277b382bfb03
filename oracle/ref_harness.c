/* oracle/ref_harness.c -- TEST INFRASTRUCTURE ONLY (never linked into the product).
 *
 * Flat-array wrappers around the *reference's own* functions so that Python
 * (ctypes) can drive them when generating golden vectors and when validating
 * the CPU restatement in oracle/nabwa_oracle.c.  Compiled by oracle/Makefile
 * against the reference headers in /root/reference (never copied).  Every
 * wrapper only marshals arguments; all arithmetic is the reference's.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include <math.h>
#include "bwt.h"
#include "bwtaln.h"
#include "bwtgap.h"
#include "bwase.h"
#include "bntseq.h"
#include "stdaln.h"
#include <pthread.h>

typedef struct {
	bwt_t *bwt[2];
	bntseq_t *bns;
	ubyte_t *pac;
} ref_index_t;

extern int g_log_n[256];
void bwase_initialize();

ref_index_t *ref_index_load(const char *prefix, int with_sa)
{
	ref_index_t *ix = (ref_index_t*)calloc(1, sizeof(ref_index_t));
	char *s = (char*)calloc(strlen(prefix) + 16, 1);
	strcpy(s, prefix); strcat(s, ".bwt");  ix->bwt[0] = bwt_restore_bwt(s, 0);
	strcpy(s, prefix); strcat(s, ".rbwt"); ix->bwt[1] = bwt_restore_bwt(s, 0);
	if (with_sa) {
		strcpy(s, prefix); strcat(s, ".sa");  bwt_restore_sa(s, ix->bwt[0], 0);
		strcpy(s, prefix); strcat(s, ".rsa"); bwt_restore_sa(s, ix->bwt[1], 0);
	}
	ix->bns = bns_restore(prefix);
	ix->pac = bwt_restore_pac(ix->bns, 0);
	bwase_initialize();
	free(s);
	return ix;
}

void ref_index_free(ref_index_t *ix)
{
	if (!ix) return;
	bwt_destroy(ix->bwt[0]); bwt_destroy(ix->bwt[1]);
	bwt_destroy_pac(ix->pac, ix->bns);
	bns_destroy(ix->bns);
	free(ix);
}

uint32_t ref_seq_len(ref_index_t *ix, int which) { return ix->bwt[which]->seq_len; }
uint32_t ref_primary(ref_index_t *ix, int which) { return ix->bwt[which]->primary; }
uint32_t ref_seed(ref_index_t *ix) { return ix->bns->seed; }

/* rank primitives (reference bwt.c:92-216) */
uint32_t ref_occ(ref_index_t *ix, int which, uint32_t k, int c) { return bwt_occ(ix->bwt[which], k, c); }
void ref_occ4(ref_index_t *ix, int which, uint32_t k, uint32_t cnt[4]) { bwt_occ4(ix->bwt[which], k, cnt); }
void ref_2occ(ref_index_t *ix, int which, uint32_t k, uint32_t l, int c, uint32_t *ok, uint32_t *ol)
{ bwt_2occ(ix->bwt[which], k, l, c, ok, ol); }
void ref_2occ4(ref_index_t *ix, int which, uint32_t k, uint32_t l, uint32_t ck[4], uint32_t cl[4])
{ bwt_2occ4(ix->bwt[which], k, l, ck, cl); }
uint32_t ref_sa(ref_index_t *ix, int which, uint32_t k) { return bwt_sa(ix->bwt[which], k); }
int ref_maxdiff(int l, double err, double thres) { return bwa_cal_maxdiff(l, err, thres); }

void ref_default_opt(gap_opt_t *o) { gap_opt_t *d = gap_init_opt(); *o = *d; free(d); }

/* bwa_cal_sa_reg_gap (reference bwtaln.c:93) over a flat batch.
 * seq/rseq: concatenated bwa_seq_t.seq / .rseq byte codes; off[i]..off[i+1] delimit read i.
 * per_read != 0 -> one call per read (what bam2bam.c:616 does), else one call for the batch
 * (what bwa aln does, bwtaln.c:235).  Output: n_aln[i], aln rows appended to aln_out
 * (4 x u32 per row: packed{n_mm,n_gapo,n_gape,a}, k, l, score), max_entries[i].
 * Returns total rows, or -1 if aln_cap is too small. */
long ref_cal_sa_reg_gap(ref_index_t *ix, const gap_opt_t *opt, int n, const int64_t *off,
						const uint8_t *seq, const uint8_t *rseq, int per_read,
						int32_t *n_aln, uint32_t *aln_out, long aln_cap, int32_t *max_entries)
{
	bwa_seq_t *s = (bwa_seq_t*)calloc(n, sizeof(bwa_seq_t));
	long tot = 0; int i;
	for (i = 0; i < n; ++i) {
		s[i].len = s[i].full_len = s[i].clip_len = (int)(off[i+1] - off[i]);
		s[i].seq = (ubyte_t*)(seq + off[i]);
		s[i].rseq = (ubyte_t*)(rseq + off[i]);
	}
	if (per_read) for (i = 0; i < n; ++i) bwa_cal_sa_reg_gap(ix->bwt, 1, s + i, opt);
	else bwa_cal_sa_reg_gap(ix->bwt, n, s, opt);
	for (i = 0; i < n; ++i) {
		n_aln[i] = s[i].n_aln;
		max_entries[i] = s[i].max_entries;
		if (tot + s[i].n_aln > aln_cap) { tot = -1; break; }
		if (s[i].n_aln) memcpy(aln_out + 4 * tot, s[i].aln, 16 * (size_t)s[i].n_aln);
		tot += s[i].n_aln;
	}
	for (i = 0; i < n; ++i) free(s[i].aln);
	free(s);
	return tot;
}

/* hit choice + position + mapQ for one SE read, in call order (global drand48 stream):
 * bwa_aln2seq_core (bwase.c:19) then bwa_cal_pac_pos_core (bwase.c:139) and the multi-hit
 * positions as bam2bam.c:629-637 / bwase.c:166-181 do.
 * out[0..11] = type,strand,n_mm,n_gapo,n_gape,score,sa,c1,c2,pos,mapQ,n_multi;
 * multi_out rows: pos(after SA lookup), gap, mm, strand. */
void ref_seed48(long seed) { srand48(seed); }
void ref_aln2pos_se(ref_index_t *ix, const gap_opt_t *opt, int len, int n_aln, const uint32_t *aln,
					int n_occ, int64_t *out, int64_t *multi_out)
{
	bwa_seq_t s; int j;
	memset(&s, 0, sizeof(s));
	s.len = s.full_len = s.clip_len = len;
	bwa_aln2seq_core(n_aln, (const bwt_aln1_t*)aln, &s, 1, n_occ);
	bwa_cal_pac_pos_core(ix->bwt[0], ix->bwt[1], &s, opt->max_diff, opt->fnr);
	for (j = 0; j < s.n_multi; ++j) {
		bwt_multi1_t *q = s.multi + j;
		if (q->strand) q->pos = bwt_sa(ix->bwt[0], q->pos);
		else q->pos = ix->bwt[1]->seq_len - (bwt_sa(ix->bwt[1], q->pos) + s.len);
		multi_out[4*j] = q->pos; multi_out[4*j+1] = q->gap; multi_out[4*j+2] = q->mm; multi_out[4*j+3] = q->strand;
	}
	out[0] = s.type; out[1] = s.strand; out[2] = s.n_mm; out[3] = s.n_gapo; out[4] = s.n_gape;
	out[5] = s.score; out[6] = s.sa; out[7] = s.c1; out[8] = s.c2; out[9] = s.pos; out[10] = s.mapQ;
	out[11] = s.n_multi;
	free(s.multi);
}

/* aln_global_core (stdaln.c:345) with an explicit parameter block.
 * matrix: row*row ints.  Returns score; cigar32 (len<<4|op) in cig_out, count in *n_cig;
 * path (i,j,ctype triples) in path_out if non-NULL. */
int ref_global(const uint8_t *s1, int l1, const uint8_t *s2, int l2, int gap_open, int gap_ext, int gap_end,
			   const int *matrix, int row, int band, uint32_t *cig_out, int *n_cig, int32_t *path_out, int *path_len)
{
	AlnParam ap; path_t *path; int score, i; uint32_t *c;
	ap.gap_open = gap_open; ap.gap_ext = gap_ext; ap.gap_end = gap_end;
	ap.matrix = (int*)matrix; ap.row = row; ap.band_width = band;
	path = (path_t*)calloc(l1 + l2 + 2, sizeof(path_t));
	score = aln_global_core((unsigned char*)s1, l1, (unsigned char*)s2, l2, &ap, path, path_len);
	c = aln_path2cigar32(path, *path_len, n_cig);
	for (i = 0; i < *n_cig; ++i) cig_out[i] = c[i];
	if (path_out) for (i = 0; i <= *path_len; ++i) {
		path_out[3*i] = path[i].i; path_out[3*i+1] = path[i].j; path_out[3*i+2] = path[i].ctype;
	}
	free(c); free(path);
	return score;
}

/* aln_local_core (stdaln.c:529) / aln_extend_core (stdaln.c:862), same marshalling. */
int ref_local(const uint8_t *s1, int l1, const uint8_t *s2, int l2, int gap_open, int gap_ext, int gap_end,
			  const int *matrix, int row, int band, int thres, uint32_t *cig_out, int *n_cig,
			  int32_t *path_out, int *path_len, int *subo)
{
	AlnParam ap; path_t *path; int score, i; uint32_t *c;
	ap.gap_open = gap_open; ap.gap_ext = gap_ext; ap.gap_end = gap_end;
	ap.matrix = (int*)matrix; ap.row = row; ap.band_width = band;
	path = (path_t*)calloc(l1 + l2 + 2, sizeof(path_t));
	score = aln_local_core((unsigned char*)s1, l1, (unsigned char*)s2, l2, &ap, path, path_len, thres, subo);
	c = aln_path2cigar32(path, *path_len, n_cig);
	for (i = 0; i < *n_cig; ++i) cig_out[i] = c[i];
	if (path_out) for (i = 0; i < *path_len; ++i) {
		path_out[3*i] = path[i].i; path_out[3*i+1] = path[i].j; path_out[3*i+2] = path[i].ctype;
	}
	free(c); free(path);
	return score;
}

int ref_extend(const uint8_t *s1, int l1, const uint8_t *s2, int l2, int gap_open, int gap_ext, int gap_end,
			   const int *matrix, int row, int band, int G0, uint32_t *cig_out, int *n_cig,
			   int32_t *path_out, int *path_len)
{
	AlnParam ap; path_t *path; int score, i; uint32_t *c;
	ap.gap_open = gap_open; ap.gap_ext = gap_ext; ap.gap_end = gap_end;
	ap.matrix = (int*)matrix; ap.row = row; ap.band_width = band;
	path = (path_t*)calloc(l1 + l2 + 2, sizeof(path_t));
	score = aln_extend_core((unsigned char*)s1, l1, (unsigned char*)s2, l2, &ap, path, path_len, G0, 0);
	c = aln_path2cigar32(path, *path_len, n_cig);
	for (i = 0; i < *n_cig; ++i) cig_out[i] = c[i];
	if (path_out) for (i = 0; i < *path_len; ++i) {
		path_out[3*i] = path[i].i; path_out[3*i+1] = path[i].j; path_out[3*i+2] = path[i].ctype;
	}
	free(c); free(path);
	return score;
}

/* bwa_refine_gapped (bwase.c:356) for one positioned SE read without multi hits.
 * seq/rseq as produced by bam1_to_seq (seq reversed; the function un-reverses it).
 * in:  strand,n_gapo,n_gape,pos,type ; out: pos', n_cigar, cigar u16[], nm, md (NUL-terminated). */
void ref_refine_one(ref_index_t *ix, int len, const uint8_t *seq, const uint8_t *rseq, int type, int strand,
					int n_mm, int n_gapo, int n_gape, uint32_t pos,
					uint32_t *pos_out, int *n_cigar, uint16_t *cigar_out, int *nm, char *md_out, int md_cap)
{
	bwa_seq_t s;
	memset(&s, 0, sizeof(s));
	s.len = s.full_len = s.clip_len = len;
	s.seq = (ubyte_t*)malloc(len + 1); memcpy(s.seq, seq, len);
	s.rseq = (ubyte_t*)malloc(len + 1); memcpy(s.rseq, rseq, len);
	s.type = type; s.strand = strand; s.n_mm = n_mm; s.n_gapo = n_gapo; s.n_gape = n_gape; s.pos = pos;
	bwa_refine_gapped(ix->bns, 1, &s, ix->pac, 0);
	*pos_out = s.pos; *n_cigar = s.n_cigar; *nm = s.nm;
	if (s.cigar) memcpy(cigar_out, s.cigar, 2 * s.n_cigar);
	md_out[0] = 0;
	if (s.md) { strncpy(md_out, s.md, md_cap - 1); md_out[md_cap - 1] = 0; }
	free(s.seq); free(s.rseq); free(s.cigar); free(s.md);
}

int ref_pac2real(ref_index_t *ix, int64_t pac_coor, int len, int32_t *seqid, int64_t *offset)
{
	int nn = bns_coor_pac2real(ix->bns, pac_coor, len, seqid);
	*offset = ix->bns->anns[*seqid].offset;
	return nn;
}

/* Wrap in-memory arrays holding the content of .bwt / .rbwt files (bench.py builds them on the GPU):
 * the same fields bwt_restore_bwt fills (bwtio.c:184-204). */
ref_index_t *ref_index_wrap(const uint32_t *bwt0, uint64_t nw0, const uint32_t *bwt1, uint64_t nw1)
{
	ref_index_t *ix = (ref_index_t*)calloc(1, sizeof(ref_index_t));
	const uint32_t *raw[2] = { bwt0, bwt1 }; uint64_t nw[2] = { nw0, nw1 }; int t;
	for (t = 0; t < 2; ++t) {
		bwt_t *b = (bwt_t*)calloc(1, sizeof(bwt_t));
		b->primary = raw[t][0];
		memcpy(b->L2 + 1, raw[t] + 1, 16);
		b->seq_len = b->L2[4];
		b->bwt_size = nw[t] - 5;
		b->bwt = (uint32_t*)(raw[t] + 5);
		bwt_gen_cnt_table(b);
		ix->bwt[t] = b;
	}
	return ix;
}
void ref_index_unwrap(ref_index_t *ix) { free(ix->bwt[0]); free(ix->bwt[1]); free(ix); }

/* bwa_cal_sa_reg_gap from n_threads host threads, each on a contiguous share of the reads, one call per
 * read as bam2bam's workers do (bam2bam.c:616; the function is re-entrant, SURVEY 8b "Threading").
 * Only n_aln[] and the hit count are kept; used as the CPU baseline of bench.py. */
typedef struct { ref_index_t *ix; const gap_opt_t *opt; int lo, hi; const int64_t *off; const uint8_t *seq, *rseq;
				 int32_t *n_aln; bwt_aln1_t **rows; } ref_job_t;
static void *ref_job(void *a)
{
	ref_job_t *J = (ref_job_t*)a; int i;
	for (i = J->lo; i < J->hi; ++i) {
		bwa_seq_t s; memset(&s, 0, sizeof(s));
		s.len = s.full_len = s.clip_len = (int)(J->off[i+1] - J->off[i]);
		s.seq = (ubyte_t*)(J->seq + J->off[i]); s.rseq = (ubyte_t*)(J->rseq + J->off[i]);
		bwa_cal_sa_reg_gap(J->ix->bwt, 1, &s, J->opt);
		J->n_aln[i] = s.n_aln; J->rows[i] = s.aln;
	}
	return 0;
}
long ref_cal_sa_reg_gap_mt(ref_index_t *ix, const gap_opt_t *opt, int n, const int64_t *off, const uint8_t *seq,
						   const uint8_t *rseq, int n_threads, int32_t *n_aln, uint32_t *aln_out, long aln_cap)
{
	ref_job_t *jobs = (ref_job_t*)calloc(n_threads, sizeof(ref_job_t));
	pthread_t *tid = (pthread_t*)calloc(n_threads, sizeof(pthread_t));
	bwt_aln1_t **rows = (bwt_aln1_t**)calloc(n ? n : 1, sizeof(*rows));
	long tot = 0; int t, i;
	for (t = 0; t < n_threads; ++t) {
		ref_job_t *J = jobs + t;
		J->ix = ix; J->opt = opt; J->off = off; J->seq = seq; J->rseq = rseq; J->n_aln = n_aln; J->rows = rows;
		J->lo = (int)((long)n * t / n_threads); J->hi = (int)((long)n * (t + 1) / n_threads);
		pthread_create(&tid[t], 0, ref_job, J);
	}
	for (t = 0; t < n_threads; ++t) pthread_join(tid[t], 0);
	for (i = 0; i < n; ++i) {
		if (tot >= 0 && tot + n_aln[i] <= aln_cap) { if (n_aln[i]) memcpy(aln_out + 4 * tot, rows[i], 16 * (size_t)n_aln[i]); tot += n_aln[i]; }
		else tot = -1;
		free(rows[i]);
	}
	free(rows); free(jobs); free(tid);
	return tot;
}

/* ------------------------------------------------------------------ paired-end pieces (bwape.c, insert_size.c) */
#include "bwape.h"

/* infer_isize_hist (insert_size.c:50-139; static) reached through infer_all_isizes (:167-173) on a table
 * with one read group.  hist: 100000 counts (the function frees it, so a copy is handed over).
 * out: avg, std, ap_prior, low, high, high_bayesian */
void ref_infer_isize(const uint16_t *hist, double ap_prior, int64_t L, double *out)
{
	khash_t(isize_infos) *h = kh_init(isize_infos);
	int ret; khiter_t it = kh_put(isize_infos, h, strdup("rg"), &ret);
	isize_info_t *ii = &kh_value(h, it);
	memset(ii, 0, sizeof(*ii));
	ii->hist = (unsigned short*)malloc(100000 * 2);
	memcpy(ii->hist, hist, 100000 * 2);
	infer_all_isizes(h, ap_prior, L);
	ii = &kh_value(h, it);
	out[0] = ii->avg; out[1] = ii->std; out[2] = ii->ap_prior; out[3] = ii->low; out[4] = ii->high; out[5] = ii->high_bayesian;
	free((char*)kh_key(h, it));
	kh_destroy(isize_infos, h);
}

/* pairing (bwape.c:180-293).  Per end e: n_aln[e] rows aln (4 x u32 each, bwt_aln1_t), and for every row its
 * text positions (flattened in hit_pos with hit_row giving the row index), as finish_pair builds d.arr
 * (bam2bam.c:737-767).  p_in per end: pos, strand, mapQ, seQ, len, full_len, n_mm, n_gapo, n_gape, score, extra_flag.
 * ii: avg, std, ap_prior, low, high, high_bayesian.  Returns cnt_chg; p_out same layout as p_in. */
int ref_pairing(const int *n_aln, const uint32_t *aln0, const uint32_t *aln1,
				int n_hit, const uint32_t *hit_pos, const int32_t *hit_row, const int32_t *hit_end,
				const int64_t *p_in, int max_isize, int pet_type, int s_mm, const double *iiv, int64_t *p_out)
{
	bwa_seq_t s[2], *p[2] = { &s[0], &s[1] };
	pe_data_t d; pe_opt_t *po = bwa_init_pe_opt(); isize_info_t ii; int e, i, r;
	bwase_initialize();
	memset(&d, 0, sizeof(d)); memset(s, 0, sizeof(s)); memset(&ii, 0, sizeof(ii));
	d.aln[0].a = (bwt_aln1_t*)aln0; d.aln[0].n = n_aln[0];
	d.aln[1].a = (bwt_aln1_t*)aln1; d.aln[1].n = n_aln[1];
	for (i = 0; i < n_hit; ++i) {
		uint64_t x = (uint64_t)hit_pos[i] << 32 | hit_row[i] << 1 | hit_end[i];
		kv_push(uint64_t, d.arr, x);
	}
	for (e = 0; e < 2; ++e) {
		const int64_t *q = p_in + 11 * e;
		s[e].pos = q[0]; s[e].strand = q[1]; s[e].mapQ = q[2]; s[e].seQ = q[3]; s[e].len = q[4]; s[e].full_len = q[5];
		s[e].n_mm = q[6]; s[e].n_gapo = q[7]; s[e].n_gape = q[8]; s[e].score = q[9]; s[e].extra_flag = q[10];
	}
	po->max_isize = max_isize; po->type = pet_type;
	ii.avg = iiv[0]; ii.std = iiv[1]; ii.ap_prior = iiv[2]; ii.low = iiv[3]; ii.high = iiv[4]; ii.high_bayesian = iiv[5];
	r = pairing(p, &d, po, s_mm, &ii);
	for (e = 0; e < 2; ++e) {
		int64_t *q = p_out + 11 * e;
		q[0] = s[e].pos; q[1] = s[e].strand; q[2] = s[e].mapQ; q[3] = s[e].seQ; q[4] = s[e].len; q[5] = s[e].full_len;
		q[6] = s[e].n_mm; q[7] = s[e].n_gapo; q[8] = s[e].n_gape; q[9] = s[e].score; q[10] = s[e].extra_flag;
	}
	kv_destroy(d.arr); free(po);
	return r;
}

/* ------------------------------------------------------------------ the per-pair chain of bam2bam
 * bam2bam.c itself cannot be compiled here (needs <zmq.h>), so this is OUR restatement of the glue in
 * posn_pair / finish_pair (bam2bam.c:683-811) -- a dozen calls in a fixed order -- with every call going to
 * the REFERENCE's own function (bwa_aln2seq*, bwa_cal_pac_pos_core, bwt_sa, pairing, bwa_paired_sw1,
 * bwa_refine_gapped).  The position cache (my_hash, bam2bam.c:741-757) only memoises bwt_sa and is left out. */
typedef struct { int n; bwa_seq_t *s; } ref_pe_batch_t;      /* s[2*i + end] */

ref_pe_batch_t *ref_pe_new(int n_pairs)
{
	ref_pe_batch_t *b = (ref_pe_batch_t*)calloc(1, sizeof(*b));
	b->n = n_pairs; b->s = (bwa_seq_t*)calloc(2 * (size_t)n_pairs, sizeof(bwa_seq_t));
	return b;
}
void ref_pe_set(ref_pe_batch_t *b, int pair, int end, int len, const uint8_t *seq, const uint8_t *rseq, int n_aln, const uint32_t *aln)
{
	bwa_seq_t *p = b->s + 2 * pair + end;
	p->len = p->full_len = p->clip_len = len;
	p->seq = (ubyte_t*)malloc(len + 1); memcpy(p->seq, seq, len);
	p->rseq = (ubyte_t*)malloc(len + 1); memcpy(p->rseq, rseq, len);
	p->n_aln = n_aln; p->aln = (bwt_aln1_t*)calloc(n_aln ? n_aln : 1, sizeof(bwt_aln1_t));
	if (n_aln) memcpy(p->aln, aln, 16 * (size_t)n_aln);
	p->extra_flag = SAM_FPD | (end ? SAM_FR2 : SAM_FR1);
}
/* pass 1 of every pair, in order (global drand48 stream): posn_pair, bam2bam.c:683-703 */
void ref_pe_posn(ref_pe_batch_t *b, ref_index_t *ix, const gap_opt_t *opt)
{
	int i, j;
	for (i = 0; i < b->n; ++i)
		for (j = 0; j < 2; ++j) {
			bwa_seq_t *p = b->s + 2 * i + j;
			p->n_multi = 0;
			bwa_aln2seq(p->n_aln, p->aln, p);
			bwa_cal_pac_pos_core(ix->bwt[0], ix->bwt[1], p, opt->max_diff, opt->fnr);
		}
}
/* pass 2 of every pair: finish_pair, bam2bam.c:705-811, up to (not including) bwa_update_bam1.  cache: finish_pair's my_hash
 * (bam2bam.c:741-757; one per pass 2 of a file, :1186-1203) in the reference's own hash map, or NULL for none: rows of MIN_HASH_WIDTH
 * suffixes or more take their positions from the first read that brought the same (k, l). */
KHASH_MAP_INIT_INT64(refpos, poslist_t)
void *ref_poscache_new(void) { return kh_init(refpos); }
void ref_poscache_free(void *c)
{
	kh_refpos_t *h = (kh_refpos_t*)c; khint_t it;
	for (it = kh_begin(h); it != kh_end(h); ++it) if (kh_exist(h, it)) free(kh_val(h, it).a);
	kh_destroy(refpos, h);
}
static void ref_pe_finish_impl(ref_pe_batch_t *b, ref_index_t *ix, const gap_opt_t *opt, const double *iiv, kh_refpos_t *my_hash);
void ref_pe_finish(ref_pe_batch_t *b, ref_index_t *ix, const gap_opt_t *opt, const double *iiv) { ref_pe_finish_impl(b, ix, opt, iiv, 0); }
void ref_pe_finish_cached(ref_pe_batch_t *b, ref_index_t *ix, const gap_opt_t *opt, const double *iiv, void *cache) { ref_pe_finish_impl(b, ix, opt, iiv, (kh_refpos_t*)cache); }
static void ref_pe_finish_impl(ref_pe_batch_t *b, ref_index_t *ix, const gap_opt_t *opt, const double *iiv, kh_refpos_t *my_hash)
{
	pe_opt_t *po = bwa_init_pe_opt(); isize_info_t ii; uint64_t n_tot[2] = {0, 0}, n_mapped[2] = {0, 0}; int i, j, k;
	memset(&ii, 0, sizeof(ii));
	ii.avg = iiv[0]; ii.std = iiv[1]; ii.ap_prior = iiv[2]; ii.low = iiv[3]; ii.high = iiv[4]; ii.high_bayesian = iiv[5];
	bwase_initialize();
	for (i = 0; i < b->n; ++i) {
		bwa_seq_t *p[2] = { b->s + 2 * i, b->s + 2 * i + 1 };
		pe_data_t d; memset(&d, 0, sizeof(d));
		for (j = 0; j < 2; ++j) { d.aln[j].a = p[j]->aln; d.aln[j].n = p[j]->n_aln; }
		if ((p[0]->type == BWA_TYPE_UNIQUE || p[0]->type == BWA_TYPE_REPEAT) && (p[1]->type == BWA_TYPE_UNIQUE || p[1]->type == BWA_TYPE_REPEAT)) {
			long long n_occ[2];
			for (j = 0; j < 2; ++j) { n_occ[j] = 0; for (k = 0; k < d.aln[j].n; ++k) n_occ[j] += d.aln[j].a[k].l - d.aln[j].a[k].k + 1; }
			if (n_occ[0] <= po->max_occ && n_occ[1] <= po->max_occ) {
				for (j = 0; j < 2; ++j)
					for (k = 0; k < d.aln[j].n; ++k) {
						bwt_aln1_t *r = d.aln[j].a + k; bwtint_t l;
						if (my_hash && r->l - r->k + 1 >= MIN_HASH_WIDTH) {
							int ret; khint_t iter = kh_put(refpos, my_hash, (uint64_t)r->k << 32 | r->l, &ret);
							if (ret) {
								poslist_t *z = &kh_val(my_hash, iter);
								z->n = r->l - r->k + 1; z->a = (bwtint_t*)malloc(sizeof(bwtint_t) * z->n);
								for (l = r->k; l <= r->l; ++l) z->a[l - r->k] = r->a ? bwt_sa(ix->bwt[0], l) : ix->bwt[1]->seq_len - (bwt_sa(ix->bwt[1], l) + p[j]->len);
							}
							for (l = 0; l < (bwtint_t)kh_val(my_hash, iter).n; ++l) {
								uint64_t x = kh_val(my_hash, iter).a[l];
								x = x << 32 | k << 1 | j;
								kv_push(uint64_t, d.arr, x);
							}
						} else for (l = r->k; l <= r->l; ++l) {
							uint64_t x = r->a ? bwt_sa(ix->bwt[0], l) : ix->bwt[1]->seq_len - (bwt_sa(ix->bwt[1], l) + p[j]->len);
							x = x << 32 | k << 1 | j;
							kv_push(uint64_t, d.arr, x);
						}
					}
				pairing(p, &d, po, opt->s_mm, &ii);
			}
		}
		for (j = 0; j < 2; ++j)
			if (p[j]->type != BWA_TYPE_NO_MATCH) {
				if (!(p[j]->extra_flag & SAM_FPP) && p[1-j]->type != BWA_TYPE_NO_MATCH)
					bwa_aln2seq_core(d.aln[j].n, d.aln[j].a, p[j], 0, p[j]->c1 + p[j]->c2 - 1 > po->N_multi ? po->n_multi : po->N_multi);
				else bwa_aln2seq_core(d.aln[j].n, d.aln[j].a, p[j], 0, po->n_multi);
				for (k = 0; k < p[j]->n_multi; ++k) {
					bwt_multi1_t *q = p[j]->multi + k;
					q->pos = q->strand ? bwt_sa(ix->bwt[0], q->pos) : ix->bwt[1]->seq_len - (bwt_sa(ix->bwt[1], q->pos) + p[j]->len);
				}
			}
		kv_destroy(d.arr);
		bwa_paired_sw1(ix->bns, ix->pac, p, po, &ii, n_tot, n_mapped);
		bwa_refine_gapped(ix->bns, 1, p[0], ix->pac, 0);
		bwa_refine_gapped(ix->bns, 1, p[1], ix->pac, 0);
	}
	free(po);
}
/* fields of one end: type,strand,n_mm,n_gapo,n_gape,score,sa,c1,c2,pos,mapQ,seQ,extra_flag,n_cigar,nm,n_multi,len ;
 * cigar (u16) ; md ; multi rows (pos,gap,mm,strand,n_cigar, then cigar padded to 16) */
void ref_pe_get(ref_pe_batch_t *b, int pair, int end, int64_t *f, uint16_t *cigar, char *md, int md_cap, int64_t *multi)
{
	const bwa_seq_t *p = b->s + 2 * pair + end; int k, c;
	f[0] = p->type; f[1] = p->strand; f[2] = p->n_mm; f[3] = p->n_gapo; f[4] = p->n_gape; f[5] = p->score; f[6] = p->sa;
	f[7] = p->c1; f[8] = p->c2; f[9] = p->pos; f[10] = p->mapQ; f[11] = p->seQ; f[12] = p->extra_flag;
	f[13] = p->cigar ? p->n_cigar : 0; f[14] = p->nm; f[15] = p->n_multi; f[16] = p->len;
	if (p->cigar) memcpy(cigar, p->cigar, 2 * p->n_cigar);
	md[0] = 0; if (p->md) { strncpy(md, p->md, md_cap - 1); md[md_cap - 1] = 0; }
	for (k = 0; k < p->n_multi; ++k) {
		const bwt_multi1_t *q = p->multi + k;
		multi[21 * k] = q->pos; multi[21 * k + 1] = q->gap; multi[21 * k + 2] = q->mm; multi[21 * k + 3] = q->strand;
		multi[21 * k + 4] = q->cigar ? q->n_cigar : 0;
		for (c = 0; c < 16; ++c) multi[21 * k + 5 + c] = (q->cigar && c < (int)q->n_cigar) ? q->cigar[c] : 0;
	}
}
void ref_pe_free(ref_pe_batch_t *b)
{
	int i;
	for (i = 0; i < 2 * b->n; ++i) bwa_free_read_seq1(b->s + i);
	free(b->s); free(b);
}

/* The insert-size estimate `sampe` makes from one batch of positioned pairs: bwape.c:74-175 is a static
 * function, so this is a RESTATEMENT of it over the harness batch (same libm, same expression order); it is
 * only used to reproduce the reference's `sampe` command end to end (make_golden.py checks the harness chain
 * against the SAM that command printed, which pins both this function and the chain glue above). */
static int cmp_u64(const void *a, const void *b) { uint64_t x = *(const uint64_t*)a, y = *(const uint64_t*)b; return x < y ? -1 : x > y; }
int ref_pe_isize_pairs(ref_pe_batch_t *b, double ap_prior, int64_t L, double *o)
{
	uint64_t x, *isz = (uint64_t*)calloc(b->n ? b->n : 1, 8), n_ap = 0; int n, i, tot = 0, p25, p75, max_len = 1, tmp;
	double avg = -1.0, std = -1.0, y, ap; uint32_t low = 0, high = 0, hb = 0;
	o[0] = o[1] = -1.0; o[2] = 0; o[3] = o[4] = o[5] = 0;
	for (i = 0; i < b->n; ++i) {
		const bwa_seq_t *p0 = b->s + 2 * i, *p1 = p0 + 1;
		if (p0->mapQ >= 20 && p1->mapQ >= 20) {
			x = (p0->pos < p1->pos) ? p1->pos + p1->len - p0->pos : p0->pos + p0->len - p1->pos;
			if (x < 100000) isz[tot++] = x;
		}
		if ((int)p0->len > max_len) max_len = p0->len;
		if ((int)p1->len > max_len) max_len = p1->len;
	}
	if (tot < 20) { free(isz); return -1; }
	qsort(isz, tot, 8, cmp_u64);
	p25 = isz[(int)(tot * 0.25 + 0.5)]; p75 = isz[(int)(tot * 0.75 + 0.5)];
	tmp = (int)(p25 - 2.0 * (p75 - p25) + .499);
	low = tmp > max_len ? tmp : max_len;
	high = (int)(p75 + 2.0 * (p75 - p25) + .499);
	for (i = 0, x = n = 0; i < tot; ++i) if (isz[i] >= low && isz[i] <= high) ++n, x += isz[i];
	avg = (double)x / n;
	for (i = 0; i < tot; ++i) if (isz[i] >= low && isz[i] <= high) { double t = (isz[i] - avg) * (isz[i] - avg); std += t; }
	std = sqrt(std / n);
	for (y = 1.0; y < 10.0; y += 0.01) if (.5 * erfc(y / M_SQRT2) < ap_prior / L * (y * std + avg)) break;
	hb = (uint32_t)(y * std + avg + .499);
	for (i = 0; i < tot; ++i) if (isz[i] > hb) ++n_ap;
	ap = .01 * (n_ap + .01) / tot; if (ap < ap_prior) ap = ap_prior;
	free(isz);
	o[2] = ap;
	if (isnan(std) || p75 > 100000) return -1;
	o[0] = avg; o[1] = std; o[3] = low; o[4] = high; o[5] = hb;
	return 0;
}

/* ------------------------------------------------------------------ the phases of bam2bam on the CALLER's bwa_seq_t records
 * (tests of the product's bwa_seq_t-level entry points: both sides get the same array).  Every call is the reference's. */
/* posn_singleton (bam2bam.c:622-641) for each record in order, then the bwa_refine_gapped of finish_singleton (:649) */
void ref_se_records(ref_index_t *ix, const gap_opt_t *opt, int max_occ_se, int n, bwa_seq_t *s)
{
	int i, j;
	bwase_initialize();
	for (i = 0; i < n; ++i) {
		bwa_seq_t *p = s + i;
		bwa_aln2seq_core(p->n_aln, p->aln, p, 1, max_occ_se);
		bwa_cal_pac_pos_core(ix->bwt[0], ix->bwt[1], p, opt->max_diff, opt->fnr);
		for (j = 0; j < p->n_multi; ++j) {
			bwt_multi1_t *q = p->multi + j;
			if (q->strand) q->pos = bwt_sa(ix->bwt[0], q->pos);
			else q->pos = ix->bwt[1]->seq_len - (bwt_sa(ix->bwt[1], q->pos) + p->len);
		}
	}
	bwa_refine_gapped(ix->bns, n, s, ix->pac, 0);
}
/* posn_pair for every pair, then finish_pair up to bwa_update_bam1 (the glue above) */
void ref_pe_records(ref_index_t *ix, const gap_opt_t *opt, const double *iiv, int n_pairs, bwa_seq_t *s)
{
	ref_pe_batch_t b; b.n = n_pairs; b.s = s;
	ref_pe_posn(&b, ix, opt);
	ref_pe_finish(&b, ix, opt, iiv);
}

/* ------------------------------------------------------------------ bench.py's CPU legs on a synthetic genome held in memory */
/* ref_index_wrap plus the suffix-array samples (content of .sa / .rsa: 7 header words, then the samples; bwtio.c:161-182), the
 * packed reference and a bntseq_t of n_contigs equal contigs (what bns_restore builds from .ann / .amb, bntseq.c:88-139). */
ref_index_t *ref_index_wrap_full(const uint32_t *bwt0, uint64_t nw0, const uint32_t *bwt1, uint64_t nw1,
								 const uint32_t *sa0, const uint32_t *sa1, const uint8_t *pac, int64_t l_pac, uint32_t seed, int n_contigs)
{
	ref_index_t *ix = ref_index_wrap(bwt0, nw0, bwt1, nw1);
	const uint32_t *sa[2] = { sa0, sa1 }; int t;
	for (t = 0; t < 2; ++t) {
		bwt_t *b = ix->bwt[t];
		b->sa_intv = sa[t][5];
		b->n_sa = (b->seq_len + b->sa_intv) / b->sa_intv;
		b->sa = (bwtint_t*)calloc(b->n_sa, sizeof(bwtint_t));
		b->sa[0] = (bwtint_t)-1;
		memcpy(b->sa + 1, sa[t] + 7, sizeof(bwtint_t) * (b->n_sa - 1));
	}
	ix->bns = (bntseq_t*)calloc(1, sizeof(bntseq_t));
	ix->bns->l_pac = l_pac; ix->bns->n_seqs = n_contigs; ix->bns->seed = seed;
	ix->bns->anns = (bntann1_t*)calloc(n_contigs, sizeof(bntann1_t));
	for (t = 0; t < n_contigs; ++t) {       /* equal contigs "synth1".. (a contig's length is an int32) */
		char nm[32]; snprintf(nm, sizeof nm, "synth%d", t + 1);
		ix->bns->anns[t].offset = l_pac * t / n_contigs; ix->bns->anns[t].len = (int32_t)(l_pac * (t + 1) / n_contigs - l_pac * t / n_contigs);
		ix->bns->anns[t].name = strdup(nm); ix->bns->anns[t].anno = strdup("");
	}
	ix->bns->n_holes = 0; ix->bns->ambs = 0;
	ix->pac = (ubyte_t*)pac;
	bwase_initialize();
	return ix;
}

/* The paired-end chain of bam2bam after the search, per pair: posn_pair in order on one drand48 stream (it is serial in the
 * reference too), then finish_pair (pairing, multi lists, bwa_paired_sw1, bwa_refine_gapped) on n_threads threads over
 * contiguous shares of the pairs -- pass 2 draws no random numbers.  Returns the seconds of (posn, finish). */
typedef struct { ref_pe_batch_t b; ref_index_t *ix; const gap_opt_t *opt; const double *iiv; } ref_pe_job_t;
static void *ref_pe_job(void *a) { ref_pe_job_t *J = (ref_pe_job_t*)a; ref_pe_finish(&J->b, J->ix, J->opt, J->iiv); return 0; }
#include <sys/time.h>
static double ref_now(void) { struct timeval tv; gettimeofday(&tv, 0); return tv.tv_sec + 1e-6 * tv.tv_usec; }
void ref_pe_chain_mt(ref_pe_batch_t *b, ref_index_t *ix, const gap_opt_t *opt, const double *iiv, int n_threads, double *secs)
{
	int t;
	double t0 = ref_now();
	ref_pe_posn(b, ix, opt);
	secs[0] = ref_now() - t0; t0 = ref_now();
	{
		ref_pe_job_t *jobs = (ref_pe_job_t*)calloc(n_threads, sizeof(*jobs));
		pthread_t *tid = (pthread_t*)calloc(n_threads, sizeof(pthread_t));
		for (t = 0; t < n_threads; ++t) {
			const int lo = (int)((long)b->n * t / n_threads), hi = (int)((long)b->n * (t + 1) / n_threads);
			jobs[t].b.n = hi - lo; jobs[t].b.s = b->s + 2 * (size_t)lo; jobs[t].ix = ix; jobs[t].opt = opt; jobs[t].iiv = iiv;
			pthread_create(&tid[t], 0, ref_pe_job, jobs + t);
		}
		for (t = 0; t < n_threads; ++t) pthread_join(tid[t], 0);
		free(jobs); free(tid);
	}
	secs[1] = ref_now() - t0;
}

/* The single-end chain of bam2bam after the search, for bench.py's end-to-end CPU leg: posn_singleton (bam2bam.c:622-641) for
 * every read in order on the one drand48 stream (serial in the reference too), then the bwa_refine_gapped of finish_singleton
 * (bam2bam.c:649) on n_threads threads over contiguous shares of the reads.  Every call is the reference's.
 * f rows (16 per read): type,strand,n_mm,n_gapo,n_gape,score,sa,c1,c2,pos,mapQ,n_multi,n_cigar,nm,len,0 ; cigar rows 64 u16 ; md rows
 * md_cap chars.  secs = seconds of (posn, refine). */
typedef struct { ref_index_t *ix; bwa_seq_t *s; int n; } ref_se_job_t;
static void *ref_se_job(void *a) { ref_se_job_t *J = (ref_se_job_t*)a; bwa_refine_gapped(J->ix->bns, J->n, J->s, J->ix->pac, 0); return 0; }
void ref_se_chain_mt(ref_index_t *ix, const gap_opt_t *opt, int max_occ_se, int n, const int64_t *off, const uint8_t *seq, const uint8_t *rseq,
					 const int32_t *n_aln, const uint32_t *aln, int n_threads, int64_t *f, uint16_t *cigar, char *md, int md_cap, double *secs)
{
	bwa_seq_t *s = (bwa_seq_t*)calloc(n ? n : 1, sizeof(bwa_seq_t)); int i, j, t; size_t row = 0; double t0;
	bwase_initialize();
	for (i = 0; i < n; ++i) {
		bwa_seq_t *p = s + i; const int len = (int)(off[i + 1] - off[i]);
		p->len = p->full_len = p->clip_len = len;
		p->seq = (ubyte_t*)malloc(len + 1); memcpy(p->seq, seq + off[i], len);
		p->rseq = (ubyte_t*)malloc(len + 1); memcpy(p->rseq, rseq + off[i], len);
		p->n_aln = n_aln[i]; p->aln = (bwt_aln1_t*)calloc(n_aln[i] ? n_aln[i] : 1, sizeof(bwt_aln1_t));
		if (n_aln[i]) memcpy(p->aln, aln + 4 * row, 16 * (size_t)n_aln[i]);
		row += n_aln[i];
	}
	t0 = ref_now();
	for (i = 0; i < n; ++i) {
		bwa_seq_t *p = s + i;
		bwa_aln2seq_core(p->n_aln, p->aln, p, 1, max_occ_se);
		bwa_cal_pac_pos_core(ix->bwt[0], ix->bwt[1], p, opt->max_diff, opt->fnr);
		for (j = 0; j < p->n_multi; ++j) {
			bwt_multi1_t *q = p->multi + j;
			if (q->strand) q->pos = bwt_sa(ix->bwt[0], q->pos);
			else q->pos = ix->bwt[1]->seq_len - (bwt_sa(ix->bwt[1], q->pos) + p->len);
		}
	}
	secs[0] = ref_now() - t0; t0 = ref_now();
	{
		ref_se_job_t *jobs = (ref_se_job_t*)calloc(n_threads, sizeof(*jobs));
		pthread_t *tid = (pthread_t*)calloc(n_threads, sizeof(pthread_t));
		for (t = 0; t < n_threads; ++t) {
			const int lo = (int)((long)n * t / n_threads), hi = (int)((long)n * (t + 1) / n_threads);
			jobs[t].ix = ix; jobs[t].s = s + lo; jobs[t].n = hi - lo;
			pthread_create(&tid[t], 0, ref_se_job, jobs + t);
		}
		for (t = 0; t < n_threads; ++t) pthread_join(tid[t], 0);
		free(jobs); free(tid);
	}
	secs[1] = ref_now() - t0;
	for (i = 0; i < n; ++i) {
		const bwa_seq_t *p = s + i; int64_t *o = f + 16 * (size_t)i;
		o[0] = p->type; o[1] = p->strand; o[2] = p->n_mm; o[3] = p->n_gapo; o[4] = p->n_gape; o[5] = p->score; o[6] = p->sa;
		o[7] = p->c1; o[8] = p->c2; o[9] = p->pos; o[10] = p->mapQ; o[11] = p->n_multi; o[12] = p->cigar ? p->n_cigar : 0; o[13] = p->nm;
		o[14] = p->len; o[15] = 0;
		if (p->cigar) memcpy(cigar + 64 * (size_t)i, p->cigar, 2 * (p->n_cigar < 64 ? p->n_cigar : 64));
		md[(size_t)md_cap * i] = 0;
		if (p->md) { strncpy(md + (size_t)md_cap * i, p->md, md_cap - 1); md[(size_t)md_cap * i + md_cap - 1] = 0; }
	}
	for (i = 0; i < n; ++i) bwa_free_read_seq1(s + i);
	free(s);
}

/* The iteration order of the reference's string hash set (khash.h, KHASH_SET_INIT_STR) after the keys are put in the order given:
 * what find_pp_tag's choice of the PP: value depends on (bam2bam.c:246-255; bam2bam.c itself cannot be compiled here).  order_out
 * receives the indexes of the distinct keys in slot order; returns their number. */
#include "khash.h"
KHASH_SET_INIT_STR(refwords)
int ref_khash_str_order(int n, const char *const *keys, int *order_out)
{
	kh_refwords_t *h = kh_init_refwords(); khint_t it; int r, i, m = 0;
	for (i = 0; i < n; ++i) kh_put_refwords(h, keys[i], &r);
	for (it = kh_begin(h); it != kh_end(h); ++it)
		if (kh_exist(h, it)) {
			for (i = 0; i < n; ++i) if (keys[i] == kh_key(h, it)) break;
			order_out[m++] = i;
		}
	kh_destroy_refwords(h);
	return m;
}
