// pe_host.hip -- host-side pieces of the paired-end path (config 3): insert-size inference and pairing.
// Floating point stays on the host, in double, in the reference's expression order (SURVEY 7 "hard parts"),
// so that the thresholds it produces are the same numbers.
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <vector>
#include "../../include/nabwa.h"
#include "nabwa_internal.hpp"

#define MAX_ISIZE 100000          /* insert_size.c:47 */
#define OUTLIER_BOUND 2.0         /* bwape.h:34 */

/* infer_isize_hist (insert_size.c:50-139) on a histogram of outer distances: the quartiles give outlier bounds, the inliers a
 * mean and a standard deviation, and a scan over multiples of sigma gives the distance beyond which a pair is more likely
 * chimeric than merely long.  The doubles must come out bit for bit (they are thresholds of pairing and of the rescue windows),
 * so the arithmetic keeps the reference's types and order where a result depends on it:
 *   - the quartile of fraction f is the bin whose cumulative count first exceeds tot * f + 0.5;
 *   - the inlier sum adds hist[i] * i as a 32-bit int product (it wraps for bins the reference could not have filled either);
 *   - the sum of squared deviations starts from the -1.0 the field is initialised with (insert_size.c:60,106);
 *   - the scan steps y by 0.01 in double, from 1.0, and stops below 10.0.
 * The reference's skewness and kurtosis only go to its log line and are not computed.  0 = usable, -1 = not (fields as the
 * reference leaves them: avg = std = -1, bounds 0). */
extern "C" int nabwa_isize_infer(const uint16_t *hist, double ap_prior, int64_t L, nabwa_isize_t *ii)
{
	if (!hist || !ii) return nabwa_fail(NABWA_EINVAL, "null argument");
	ii->avg = ii->std = -1.0;
	ii->low = ii->high = ii->high_bayesian = 0;
	int tot = 0;
	for (int i = 0; i < MAX_ISIZE; ++i) tot += hist[i];
	if (tot < 20) return -1;                                   /* too few good pairs; ap_prior untouched */
	int q25 = 0, q75 = 0;
	{
		const double t25 = tot * 0.25 + 0.5, t75 = tot * 0.75 + 0.5;
		int below = 0;
		for (int i = 0; i < MAX_ISIZE; ++i) {
			const int upto = below + hist[i];
			if (below <= t25 && upto > t25) q25 = i;
			if (below <= t75 && upto > t75) q75 = i;
			below = upto;
		}
	}
	const int lo = (int)(q25 - OUTLIER_BOUND * (q75 - q25) + .499);
	ii->low = lo > 1 ? (uint32_t)lo : 1u;
	ii->high = (uint32_t)(int)(q75 + OUTLIER_BOUND * (q75 - q25) + .499);
	const uint32_t first = ii->low, last = ii->high < (uint32_t)(MAX_ISIZE - 1) ? ii->high : (uint32_t)(MAX_ISIZE - 1);
	int n_in = 0; uint64_t sum = 0;
	for (uint32_t i = first; i <= last; ++i) { n_in += hist[i]; sum += (uint64_t)(int64_t)(int32_t)((uint32_t)hist[i] * i); }
	ii->avg = (double)sum / n_in;
	for (uint32_t i = first; i <= last; ++i) { const double dev = (int)i - ii->avg; ii->std += dev * dev * hist[i]; }
	ii->std = sqrt(ii->std / n_in);
	double y = 1.0;
	while (y < 10.0 && !(.5 * erfc(y / M_SQRT2) < ap_prior / L * (y * ii->std + ii->avg))) y += 0.01;
	ii->high_bayesian = (uint32_t)(y * ii->std + ii->avg + .499);
	uint64_t beyond = 0;
	for (uint32_t i = ii->high_bayesian + 1; i < (uint32_t)MAX_ISIZE; ++i) beyond += hist[i];
	ii->ap_prior = .01 * (beyond + .01) / tot;
	if (ii->ap_prior < ap_prior) ii->ap_prior = ap_prior;
	if (isnan(ii->std) || q75 > MAX_ISIZE) {
		ii->low = ii->high = ii->high_bayesian = 0; ii->avg = ii->std = -1.0;
		return -1;
	}
	return 0;
}

/* improve_isize_est (insert_size.c:141-165) for one record: returns the histogram bin to bump, or -1.
 * kind 1 = single read (its length counts), 2 = pair (outer distance); both ends need mapQ >= 20. */
extern "C" int nabwa_isize_bin(int kind, int mapq0, int mapq1, uint32_t pos0, int len0, uint32_t pos1, int len1)
{
	if (kind < 1 || mapq0 < 20) return -1;
	if (kind > 1 && mapq1 < 20) return -1;
	const int len = kind == 1 ? len0 : (pos0 < pos1 ? (int)(pos1 + (uint32_t)len1 - pos0) : (int)(pos0 + (uint32_t)len0 - pos1));
	if (len < 0 || len >= MAX_ISIZE) return -1;
	return len;
}

extern "C" void nabwa_pe_opt_default(nabwa_pe_opt_t *po)            /* bwa_init_pe_opt, bwape.c:27-41 */
{
	if (!po) return;
	memset(po, 0, sizeof(*po));
	po->max_isize = 500; po->max_occ = 100000; po->max_occ_se = 3; po->n_multi = 3; po->N_multi = 10; po->type = 1; po->is_sw = 1; po->ap_prior = 1e-5;
}

static inline uint64_t mix_u64(uint64_t key)       /* the tie-breaking hash of pairing (bwape.c:43-54) */
{
	key += ~(key << 32); key ^= (key >> 22); key += ~(key << 13); key ^= (key >> 8);
	key += (key << 3); key ^= (key >> 15); key += ~(key << 27); key ^= (key >> 31);
	return key;
}

static int log_n(int n) { return n > 0 ? (int)(4.343 * log((double)n) + 0.5) : 0; }     /* g_log_n, bwase.c:613-617 */

/* pairing (bwape.c:180-293).  hits: every text position of every hit row of both ends as
 * pos << 32 | row << 1 | end (what finish_pair collects, bam2bam.c:737-767); rows: the two bwt_aln1_t arrays.
 * Sweep the sorted positions: a reverse-strand hit pairs with the last two forward-strand hits of the other
 * end; keep the best (and second best) pair by 10*(score sum) + insert-size penalty, ties broken by the hash.
 * Then derive the paired mapping qualities and move the ends that the best pair places elsewhere. */
extern "C" int nabwa_pairing(nabwa_pe_end_t p[2], int n_hits, uint64_t *hits, const nabwa_aln1_t *rows0, const nabwa_aln1_t *rows1,
							 int max_isize, int s_mm, const nabwa_isize_t *ii)
{
	const nabwa_aln1_t *rows[2] = { rows0, rows1 };
	int o_n = 0, subo_n = 0, cnt_chg = 0;
	uint64_t last_pos[2][2], o_pos[2] = { 0, 0 }, subo_score = ~0ull, o_score = ~0ull;
	int max_len = p[0].full_len; if (max_len < p[1].full_len) max_len = p[1].full_len;
	auto rowof = [&](uint64_t v) -> const nabwa_aln1_t& { return rows[v & 1][(uint32_t)v >> 1]; };
	auto consider = [&](uint64_t u, uint64_t v) {            /* v: reverse-strand hit; u: earlier forward hit of the mate */
		const uint32_t l = (uint32_t)((v >> 32) + (uint32_t)p[v & 1].len - (u >> 32));
		if (u != ~0ull && (v >> 32) > (u >> 32) && l >= (uint32_t)max_len
			&& ((ii->high && l <= ii->high_bayesian) || (ii->high == 0 && l <= (uint32_t)max_isize))) {
			uint64_t s = (uint64_t)(rowof(v).score + rowof(u).score);
			s *= 10;
			if (ii->high) s += (int)(-4.343 * log(.5 * erfc(M_SQRT1_2 * fabs(l - ii->avg) / ii->std)) + .499);
			s = s << 32 | (uint32_t)mix_u64((u >> 32) << 32 | (v >> 32));
			if (s >> 32 == o_score >> 32) ++o_n;
			else if (s >> 32 < o_score >> 32) { subo_n += o_n; o_n = 1; }
			else ++subo_n;
			if (s < o_score) { subo_score = o_score; o_score = s; o_pos[u & 1] = u; o_pos[v & 1] = v; }
			else if (s < subo_score) subo_score = s;
		}
	};
	std::sort(hits, hits + n_hits);
	for (int j = 0; j < 2; ++j) last_pos[j][0] = last_pos[j][1] = ~0ull;
	for (int i = 0; i < n_hits; ++i) {
		const uint64_t x = hits[i];
		if ((rowof(x).info >> 24 & 1) == 1) {                /* reverse strand: check against the mate's forward hits */
			const int y = 1 - (int)(x & 1);
			consider(last_pos[y][1], x);
			consider(last_pos[y][0], x);
		} else { last_pos[x & 1][0] = last_pos[x & 1][1]; last_pos[x & 1][1] = x; }
	}
	if (o_score != ~0ull) {
		int mapQ_p = 0, rr[2];
		if (o_n == 1) {
			if (subo_score == ~0ull) mapQ_p = 29;
			else if ((subo_score >> 32) - (o_score >> 32) > (uint64_t)(s_mm * 10)) mapQ_p = 23;
			else {
				const int n = subo_n > 255 ? 255 : subo_n;
				mapQ_p = (int)(((subo_score >> 32) - (o_score >> 32)) / 2) - log_n(n);
				if (mapQ_p < 0) mapQ_p = 0;
			}
		}
		rr[0] = rowof(o_pos[0]).info >> 24 & 1; rr[1] = rowof(o_pos[1]).info >> 24 & 1;
		const bool same0 = p[0].pos == (uint32_t)(o_pos[0] >> 32) && p[0].strand == rr[0];
		const bool same1 = p[1].pos == (uint32_t)(o_pos[1] >> 32) && p[1].strand == rr[1];
		if (same0 && same1) {
			if (p[0].mapQ > 0 && p[1].mapQ > 0) {
				int mapQ = p[0].mapQ + p[1].mapQ; if (mapQ > 60) mapQ = 60;
				p[0].mapQ = p[1].mapQ = mapQ;
			} else {
				if (p[0].mapQ == 0) p[0].mapQ = (mapQ_p + 7 < p[1].mapQ) ? mapQ_p + 7 : p[1].mapQ;
				if (p[1].mapQ == 0) p[1].mapQ = (mapQ_p + 7 < p[0].mapQ) ? mapQ_p + 7 : p[0].mapQ;
			}
		} else if (same0) { p[1].seQ = 0; p[1].mapQ = p[0].mapQ; if (p[1].mapQ > mapQ_p) p[1].mapQ = mapQ_p; }
		else if (same1) { p[0].seQ = 0; p[0].mapQ = p[1].mapQ; if (p[0].mapQ > mapQ_p) p[0].mapQ = mapQ_p; }
		else { p[0].seQ = p[1].seQ = 0; mapQ_p -= 20; if (mapQ_p < 0) mapQ_p = 0; p[0].mapQ = p[1].mapQ = mapQ_p; }
		for (int e = 0; e < 2; ++e) {                          /* __pairing_aux2 */
			const nabwa_aln1_t &r = rowof(o_pos[e]);
			nabwa_pe_end_t &q = p[e];
			q.extra_flag |= 2;                                 /* SAM_FPP */
			if (q.pos != (uint32_t)(o_pos[e] >> 32) || q.strand != (int)(r.info >> 24 & 1)) {
				q.n_mm = r.info & 0xff; q.n_gapo = r.info >> 8 & 0xff; q.n_gape = r.info >> 16 & 0xff; q.strand = r.info >> 24 & 1;
				q.score = r.score; q.pos = (uint32_t)(o_pos[e] >> 32);
				if (q.mapQ > 0) ++cnt_chg;
			}
		}
	}
	return cnt_chg;
}
