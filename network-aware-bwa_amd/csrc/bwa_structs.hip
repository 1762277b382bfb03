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
