// bwa_structs.hip -- the drop-in entry points that take the reference's own structs.
//
// nabwa_bwa_seq_t mirrors bwa_seq_t (bwtaln.h:64-90) field for field -- 200 bytes on LP64, offsets as
// measured on the compiled reference (seq 8, rseq 16, qual 24, bit-fields 32/36, score 40,
// clip_len 44, n_aln 48, aln 56, n_multi 64, multi 72, sa 80, pos 84, c1/c2/seQ 88, n_cigar 96, cigar 104,
// tid 112, bc 116, full_len/nm 180, md 184, max_entries 192) -- so an array of the reference's records can
// be handed over unchanged.  Only host marshalling happens here; the work is nabwa_cal_sa_reg_gap's.
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "../../include/nabwa.h"
#include "nabwa_internal.hpp"

static_assert(sizeof(nabwa_bwa_seq_t) == 200, "nabwa_bwa_seq_t must match bwa_seq_t (bwtaln.h:64-90)");
static_assert(offsetof(nabwa_bwa_seq_t, n_aln) == 48 && offsetof(nabwa_bwa_seq_t, aln) == 56, "bwa_seq_t layout");
static_assert(offsetof(nabwa_bwa_seq_t, sa) == 80 && offsetof(nabwa_bwa_seq_t, pos) == 84, "bwa_seq_t layout");
static_assert(offsetof(nabwa_bwa_seq_t, md) == 184 && offsetof(nabwa_bwa_seq_t, max_entries) == 192, "bwa_seq_t layout");

/* bwa_cal_sa_reg_gap (bwtaln.c:93-142) with the reference's signature, the bwt_t pair replaced by the
 * HBM-resident index.  Pre/post-conditions as documented at bwtaln.c:82-91: seq, rseq, len are read; n_aln,
 * aln (malloc'd; the caller frees it, as bwa_free_read_seq1 does), max_entries are filled; sa, type, c1, c2
 * are reset (bwtaln.c:113).  n_seqs > 1 derives the option block from the longest read of the call, as the
 * reference does (bwtaln.c:102-106). */
extern "C" int nabwa_bwa_cal_sa_reg_gap(nabwa_index_t *ix, int n_seqs, nabwa_bwa_seq_t *seqs, const nabwa_gap_opt_t *opt)
{
	if (!ix || !opt || n_seqs < 0 || (n_seqs && !seqs)) return nabwa_fail(NABWA_EINVAL, "null argument");
	std::vector<int64_t> off(n_seqs + 1, 0);
	for (int i = 0; i < n_seqs; ++i) off[i + 1] = off[i] + (seqs[i].bits0 & 0xfffffu);          /* len:20 */
	std::vector<uint8_t> s(off[n_seqs] + 1), r(off[n_seqs] + 1);
	for (int i = 0; i < n_seqs; ++i) {
		const size_t L = (size_t)(off[i + 1] - off[i]);
		if (L && (!seqs[i].seq || !seqs[i].rseq)) return nabwa_fail(NABWA_EINVAL, "bwa_seq_t without seq/rseq");
		if (L) { memcpy(&s[off[i]], seqs[i].seq, L); memcpy(&r[off[i]], seqs[i].rseq, L); }
	}
	std::vector<int32_t> n_aln(n_seqs ? n_seqs : 1), maxe(n_seqs ? n_seqs : 1);
	/* one search, two fetches: the first tells the number of rows, the second brings them (a fetch does not search again) */
	nabwa_batch_t *bt = 0;
	int rc = nabwa_batch_create(ix, opt, n_seqs, off.data(), s.data(), r.data(), /*per_read*/0, &bt);
	if (rc != NABWA_OK) return rc;
	rc = nabwa_batch_run(bt);
	if (rc == NABWA_OK) rc = nabwa_batch_sync(bt, 0);
	int64_t rows = 0;
	std::vector<nabwa_aln1_t> aln(1);
	if (rc == NABWA_OK) {
		rc = nabwa_batch_fetch(bt, n_aln.data(), 0, 0, &rows, maxe.data());
		if (rc == NABWA_ECAP && rows > 0) {
			aln.resize((size_t)rows);
			rc = nabwa_batch_fetch(bt, n_aln.data(), aln.data(), rows, &rows, maxe.data());
		}
	}
	nabwa_batch_destroy(bt);
	if (rc != NABWA_OK) return rc;
	int64_t a0 = 0;
	for (int i = 0; i < n_seqs; ++i) {
		nabwa_bwa_seq_t *p = seqs + i;
		p->sa = 0; p->bits0 &= ~(3u << 21);                                   /* type = BWA_TYPE_NO_MATCH */
		p->c1c2seq &= ~((1ull << 56) - 1);                                    /* c1 = c2 = 0, seQ kept */
		p->n_aln = 0; p->aln = 0;
		if ((p->bits0 & 0xfffffu) > 0) {
			/* the reference returns calloc'd storage of at least 4 rows even for 0 hits (bwtgap.c:113-114) */
			int m = 4; while (m < n_aln[i]) m <<= 1;
			p->aln = (nabwa_aln1_t*)calloc(m, sizeof(nabwa_aln1_t));
			if (n_aln[i]) memcpy(p->aln, &aln[a0], sizeof(nabwa_aln1_t) * n_aln[i]);
			p->n_aln = n_aln[i];
			p->max_entries = maxe[i];
		}
		a0 += n_aln[i];
	}
	return NABWA_OK;
}

/* bam1_to_seq's encoding step (bwaseqio.c:272-307) from already decoded base codes: given the read as it
 * sits in the BAM record (codes 0-3, 4 = N; `reverse` = the record's reverse-strand flag, which is undone,
 * bwaseqio.c:288-291), phred qualities (may be NULL) and trim_qual, produce seq (read reversed), rseq
 * (reverse complement, or a plain copy of seq without BWA_MODE_COMPREAD) and the trimmed length
 * (bwa_trim_read, bwaseqio.c:110-123; reads are never trimmed below BWA_MIN_RDLEN = 35).  Returns len. */
extern "C" int nabwa_encode_read(int full_len, const uint8_t *codes, const uint8_t *qual, int reverse, int trim_qual, int is_comp,
								 uint8_t *seq_out, uint8_t *rseq_out)
{
	std::vector<uint8_t> c(full_len ? full_len : 1), q(full_len ? full_len : 1);
	for (int i = 0; i < full_len; ++i) {
		const int j = reverse ? full_len - 1 - i : i;
		uint8_t b = codes[j];
		if (reverse && b < 4) b = 3 - b;
		c[i] = b; q[i] = qual ? qual[j] : 0;
	}
	int len = full_len;
	if (trim_qual >= 1 && qual) {
		int s = 0, mx = 0, max_l = full_len - 1;
		for (int l = full_len - 1; l >= 35 - 1; --l) {
			s += trim_qual - (int)q[l];
			if (s < 0) break;
			if (s > mx) { mx = s; max_l = l; }
		}
		len = max_l + 1;
	}
	for (int i = 0; i < len; ++i) {
		const uint8_t b = c[len - 1 - i];
		seq_out[i] = b;
		rseq_out[i] = is_comp ? (b < 4 ? 3 - b : b) : b;
	}
	return len;
}

/* ------------------------------------------------------------------------------------------------------------------
 * The phases after the search on the reference's own records: batch forms of what posn_singleton / finish_singleton /
 * posn_pair / finish_pair (bam2bam.c:622-811) do to bwa_seq_t, so that each of those functions becomes one call over the
 * records a batching front-end has gathered.  Ownership follows the reference: multi, multi[].cigar, cigar and md are
 * malloc'd here and freed by bwa_free_read_seq1 (bwaseqio.c:253-261).
 * ------------------------------------------------------------------------------------------------------------------ */
struct ref_multi1_t { uint32_t pos; uint32_t bits; uint16_t *cigar; };      /* bwt_multi1_t (bwtaln.h:58-62): n_cigar:15, gap:8, mm:8, strand:1 */
static_assert(sizeof(ref_multi1_t) == 16, "bwt_multi1_t is 16 bytes");

static void se_from_seq(const nabwa_bwa_seq_t &q, nabwa_se_t &s)
{
	memset(&s, 0, offsetof(nabwa_se_t, cigar));
	s.len = (int)(q.bits0 & 0xfffffu); s.strand = (int)(q.bits0 >> 20 & 1u); s.type = (int)(q.bits0 >> 21 & 3u);
	s.n_mm = (int)(q.bits1 & 0xffu); s.n_gapo = (int)(q.bits1 >> 8 & 0xffu); s.n_gape = (int)(q.bits1 >> 16 & 0xffu); s.mapQ = (int)(q.bits1 >> 24);
	s.score = q.score; s.clip_len = q.clip_len; s.sa = q.sa; s.pos = q.pos;
	s.c1 = (uint32_t)(q.c1c2seq & 0xfffffffull); s.c2 = (uint32_t)(q.c1c2seq >> 28 & 0xfffffffull); s.seQ = (int)(q.c1c2seq >> 56);
	s.full_len = (int)(q.lenbits & 0xfffffu); s.nm = (int)(q.lenbits >> 20);
	s.n_cigar = 0; s.md[0] = 0; s.flag = 0; s.seqid = 0; s.nn = 0; s.rpos = 0; s.xt = 0;
	s.n_multi = q.n_multi < 0 ? 0 : (q.n_multi > NABWA_MAX_MULTI ? NABWA_MAX_MULTI : q.n_multi);
	const ref_multi1_t *m = (const ref_multi1_t*)q.multi;
	for (int j = 0; j < s.n_multi; ++j) {
		s.multi[j].pos = m[j].pos; s.multi[j].gap = (int)(m[j].bits >> 15 & 0xffu); s.multi[j].mm = (int)(m[j].bits >> 23 & 0xffu);
		s.multi[j].strand = (int)(m[j].bits >> 31); s.multi[j].n_cigar = 0;
	}
}

/* the scalar fields of a record, and a fresh multi list (positions, no CIGARs yet) */
static void seq_scalars_from_se(const nabwa_se_t &s, nabwa_bwa_seq_t &q, int extra_flag)
{
	q.bits0 = ((uint32_t)s.len & 0xfffffu) | (uint32_t)(s.strand & 1) << 20 | (uint32_t)(s.type & 3) << 21 | (q.bits0 & (1u << 23)) | ((uint32_t)extra_flag & 0xffu) << 24;
	q.bits1 = ((uint32_t)s.n_mm & 0xffu) | ((uint32_t)s.n_gapo & 0xffu) << 8 | ((uint32_t)s.n_gape & 0xffu) << 16 | ((uint32_t)s.mapQ & 0xffu) << 24;
	q.score = s.score; q.clip_len = s.clip_len; q.sa = s.sa; q.pos = s.pos;
	q.c1c2seq = ((uint64_t)s.c1 & 0xfffffffull) | ((uint64_t)s.c2 & 0xfffffffull) << 28 | ((uint64_t)s.seQ & 0xffull) << 56;
	q.lenbits = ((uint32_t)s.full_len & 0xfffffu) | ((uint32_t)s.nm & 0xfffu) << 20;
}

static void seq_multi_from_se(const nabwa_se_t &s, nabwa_bwa_seq_t &q)
{
	ref_multi1_t *old = (ref_multi1_t*)q.multi;
	for (int j = 0; old && j < q.n_multi; ++j) free(old[j].cigar);
	free(old);
	q.multi = 0; q.n_multi = s.n_multi;
	if (s.n_multi == 0) return;
	ref_multi1_t *m = (ref_multi1_t*)calloc(s.n_multi, sizeof(ref_multi1_t));
	for (int j = 0; j < s.n_multi; ++j) {
		m[j].pos = s.multi[j].pos;
		m[j].bits = ((uint32_t)s.multi[j].n_cigar & 0x7fffu) | ((uint32_t)s.multi[j].gap & 0xffu) << 15 | ((uint32_t)s.multi[j].mm & 0xffu) << 23 | (uint32_t)(s.multi[j].strand & 1) << 31;
		if (s.multi[j].n_cigar) { m[j].cigar = (uint16_t*)malloc(2 * (size_t)s.multi[j].n_cigar); memcpy(m[j].cigar, s.multi[j].cigar, 2 * (size_t)s.multi[j].n_cigar); }
	}
	q.multi = m;
}

static void seq_alignment_from_se(const nabwa_se_t &s, nabwa_bwa_seq_t &q)
{
	free(q.cigar); q.cigar = 0; q.n_cigar = s.n_cigar;
	if (s.n_cigar) { q.cigar = (uint16_t*)malloc(2 * (size_t)s.n_cigar); memcpy(q.cigar, s.cigar, 2 * (size_t)s.n_cigar); }
	free(q.md); q.md = 0;
	if (s.type != 0) q.md = strdup(s.md);
}

/* flat views of n records: lengths, full lengths, hit rows back to back, and (when asked) the bases */
struct FlatRecs { std::vector<int64_t> off; std::vector<int32_t> full_len, n_aln; std::vector<nabwa_aln1_t> aln; std::vector<uint8_t> seq, rseq; };
static int flatten(int n, const nabwa_bwa_seq_t *seqs, bool want_seq, bool want_aln, FlatRecs &F)
{
	F.off.assign(n + 1, 0); F.full_len.assign(n ? n : 1, 0); F.n_aln.assign(n ? n : 1, 0);
	size_t rows = 0;
	for (int i = 0; i < n; ++i) {
		F.off[i + 1] = F.off[i] + (seqs[i].bits0 & 0xfffffu);
		F.full_len[i] = (int32_t)(seqs[i].lenbits & 0xfffffu);
		if (want_aln) { if (seqs[i].n_aln < 0 || (seqs[i].n_aln && !seqs[i].aln)) return nabwa_fail(NABWA_EINVAL, "bwa_seq_t without aln"); F.n_aln[i] = seqs[i].n_aln; rows += (size_t)seqs[i].n_aln; }
	}
	if (want_aln) {
		F.aln.resize(rows ? rows : 1);
		size_t a = 0;
		for (int i = 0; i < n; ++i) { if (seqs[i].n_aln) memcpy(&F.aln[a], seqs[i].aln, sizeof(nabwa_aln1_t) * (size_t)seqs[i].n_aln); a += (size_t)seqs[i].n_aln; }
	}
	if (want_seq) {
		F.seq.assign((size_t)F.off[n] + 1, 0); F.rseq.assign((size_t)F.off[n] + 1, 0);
		for (int i = 0; i < n; ++i) {
			const size_t L = (size_t)(F.off[i + 1] - F.off[i]);
			if (L && (!seqs[i].seq || !seqs[i].rseq)) return nabwa_fail(NABWA_EINVAL, "bwa_seq_t without seq/rseq");
			if (L) { memcpy(&F.seq[F.off[i]], seqs[i].seq, L); memcpy(&F.rseq[F.off[i]], seqs[i].rseq, L); }
		}
	}
	return NABWA_OK;
}

/* posn_singleton (bam2bam.c:622-641) for n records IN RECORD ORDER: bwa_aln2seq_core(n_aln, aln, p, 1, max_occ_se) on the
 * caller's drand48 stream, bwa_cal_pac_pos_core, the positions of the multi hits. */
extern "C" int nabwa_bwa_posn_se(nabwa_index_t *ix, const nabwa_gap_opt_t *opt, int max_occ_se, int n, nabwa_bwa_seq_t *seqs, uint64_t *rng48)
{
	if (!ix || !opt || !rng48 || n < 0 || (n && !seqs)) return nabwa_fail(NABWA_EINVAL, "null argument");
	FlatRecs F;
	int rc = flatten(n, seqs, false, true, F);
	if (rc != NABWA_OK) return rc;
	std::vector<nabwa_se_t> out(n ? n : 1);
	rc = nabwa_se_posn(ix, opt, n, F.off.data(), F.full_len.data(), F.n_aln.data(), F.aln.data(), max_occ_se, rng48, out.data());
	if (rc != NABWA_OK) return rc;
	for (int i = 0; i < n; ++i) {
		out[i].clip_len = seqs[i].clip_len; out[i].nm = (int)(seqs[i].lenbits >> 20);      /* not touched by this phase */
		seq_scalars_from_se(out[i], seqs[i], (int)(seqs[i].bits0 >> 24));
		seq_multi_from_se(out[i], seqs[i]);
	}
	return NABWA_OK;
}

/* bwa_refine_gapped(bns, n, seqs, pac, ntbns) (bwase.c:356-423; callers bam2bam.c:649,799-800) for n positioned records:
 * CIGARs of the gapped hits (main and multi), MD / NM, the quality-trimmed tail as a soft clip.  As in the reference, seq is
 * turned back into the read's own orientation (bwase.c:369) -- call it once per record. */
extern "C" int nabwa_bwa_refine_gapped(nabwa_index_t *ix, int n, nabwa_bwa_seq_t *seqs)
{
	if (!ix || n < 0 || (n && !seqs)) return nabwa_fail(NABWA_EINVAL, "null argument");
	FlatRecs F;
	int rc = flatten(n, seqs, true, false, F);
	if (rc != NABWA_OK) return rc;
	std::vector<nabwa_se_t> recs(n ? n : 1);
	for (int i = 0; i < n; ++i) se_from_seq(seqs[i], recs[i]);
	rc = nabwa_se_refine(ix, n, F.off.data(), F.seq.data(), F.rseq.data(), recs.data());
	if (rc != NABWA_OK) return rc;
	for (int i = 0; i < n; ++i) {
		nabwa_bwa_seq_t &q = seqs[i];
		const int L = (int)(q.bits0 & 0xfffffu);
		for (int a = 0, b = L - 1; a < b; ++a, --b) { const uint8_t t = q.seq[a]; q.seq[a] = q.seq[b]; q.seq[b] = t; }      /* seq_reverse(len, seq, 0) */
		const int mq = (int)(q.bits1 >> 24);                                 /* mapQ is bwa_update_bam1's to clear, not this phase's */
		recs[i].mapQ = mq;
		seq_scalars_from_se(recs[i], q, (int)(q.bits0 >> 24));
		seq_multi_from_se(recs[i], q);
		seq_alignment_from_se(recs[i], q);
	}
	return NABWA_OK;
}

/* posn_pair (bam2bam.c:683-703) for n_pairs pairs, records interleaved (2 * pair + end), in record order */
extern "C" int nabwa_bwa_posn_pe(nabwa_index_t *ix, const nabwa_gap_opt_t *opt, int n_pairs, nabwa_bwa_seq_t *seqs, uint64_t *rng48)
{
	if (!ix || !opt || !rng48 || n_pairs < 0 || (n_pairs && !seqs)) return nabwa_fail(NABWA_EINVAL, "null argument");
	const int n = 2 * n_pairs;
	FlatRecs F;
	int rc = flatten(n, seqs, false, true, F);
	if (rc != NABWA_OK) return rc;
	std::vector<nabwa_pe_t> out(n ? n : 1);
	rc = nabwa_pe_posn(ix, opt, n_pairs, F.off.data(), F.full_len.data(), F.n_aln.data(), F.aln.data(), rng48, out.data());
	if (rc != NABWA_OK) return rc;
	for (int i = 0; i < n; ++i) {
		out[i].se.clip_len = seqs[i].clip_len; out[i].se.nm = (int)(seqs[i].lenbits >> 20);
		seq_scalars_from_se(out[i].se, seqs[i], (int)(seqs[i].bits0 >> 24));
		seq_multi_from_se(out[i].se, seqs[i]);                              /* n_multi = 0 (bam2bam.c:692) */
	}
	return NABWA_OK;
}

/* finish_pair up to, not including, bwa_update_bam1 (bam2bam.c:705-800): hit enumeration + pairing, the multi-hit lists,
 * bwa_paired_sw1, bwa_refine_gapped on both ends.  ii: the read group's insert-size estimate (all zeros: none). */
extern "C" int nabwa_bwa_finish_pe(nabwa_index_t *ix, const nabwa_gap_opt_t *opt, const nabwa_pe_opt_t *popt, const nabwa_isize_t *ii,
								   int n_pairs, nabwa_bwa_seq_t *seqs, uint64_t n_tot[2], uint64_t n_mapped[2])
{
	return nabwa_bwa_finish_pe_cached(ix, opt, popt, ii, n_pairs, seqs, n_tot, n_mapped, 0);
}

extern "C" int nabwa_bwa_finish_pe_cached(nabwa_index_t *ix, const nabwa_gap_opt_t *opt, const nabwa_pe_opt_t *popt, const nabwa_isize_t *ii,
										  int n_pairs, nabwa_bwa_seq_t *seqs, uint64_t n_tot[2], uint64_t n_mapped[2], nabwa_poscache_t *cache)
{
	if (!ix || !opt || !popt || !ii || n_pairs < 0 || (n_pairs && !seqs)) return nabwa_fail(NABWA_EINVAL, "null argument");
	const int n = 2 * n_pairs;
	FlatRecs F;
	int rc = flatten(n, seqs, true, true, F);
	if (rc != NABWA_OK) return rc;
	std::vector<nabwa_pe_t> recs(n ? n : 1);
	for (int i = 0; i < n; ++i) {
		memset(&recs[i], 0, offsetof(nabwa_pe_t, se) + offsetof(nabwa_se_t, cigar));
		se_from_seq(seqs[i], recs[i].se);
		recs[i].extra_flag = (int)(seqs[i].bits0 >> 24); recs[i].m_seqid = 0; recs[i].am = 0; recs[i].mapQ_paired = 0; recs[i].m_rpos = 0; recs[i].isize = 0;
	}
	rc = nabwa_pe_finish_cached(ix, opt, popt, ii, n_pairs, F.off.data(), F.seq.data(), F.rseq.data(), F.n_aln.data(), F.aln.data(), recs.data(), n_tot, n_mapped, cache);
	if (rc != NABWA_OK) return rc;
	for (int i = 0; i < n; ++i) {
		nabwa_bwa_seq_t &q = seqs[i];
		const int L = (int)(q.bits0 & 0xfffffu);
		for (int a = 0, b = L - 1; a < b; ++a, --b) { const uint8_t t = q.seq[a]; q.seq[a] = q.seq[b]; q.seq[b] = t; }
		recs[i].se.mapQ = recs[i].mapQ_paired;                       /* the bridging rule is bwa_update_bam1's, not this phase's */
		seq_scalars_from_se(recs[i].se, q, recs[i].extra_flag);
		seq_multi_from_se(recs[i].se, q);
		seq_alignment_from_se(recs[i].se, q);
	}
	return NABWA_OK;
}
