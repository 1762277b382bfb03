// pe_host.hip -- host-side pieces of the paired-end path (config 3): insert-size inference and pairing.
// Floating point stays on the host, in double, in the reference's expression order (SURVEY 7 "hard parts"),
// so that the thresholds it produces are the same numbers.
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <thread>
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

/* improve_isize_est over the positioned pairs of a batch (records interleaved 2 * pair + end, as nabwa_pe_posn leaves them):
 * every pair whose two ends have mapQ >= 20 adds one to the bin of its outer distance; bins are the reference's uint16_t and
 * wrap as they do there (insert_size.c:157). */
extern "C" int nabwa_isize_add_pairs(int n_pairs, const nabwa_pe_t *recs, uint16_t *hist)
{
	if (n_pairs < 0 || (n_pairs && !recs) || !hist) return nabwa_fail(NABWA_EINVAL, "null argument");
	/* the bins by all threads (two 3 KB records per pair: the next ones are asked for ahead), the counts by one: uint16 sums do not care for the order */
	std::vector<int32_t> bin((size_t)(n_pairs ? n_pairs : 1));
	int nt = (int)std::thread::hardware_concurrency(); if (nt < 1) nt = 1; if (nt > 16) nt = 16;
	if (getenv("NABWA_HOST_THREADS")) nt = atoi(getenv("NABWA_HOST_THREADS")) > 0 ? atoi(getenv("NABWA_HOST_THREADS")) : 1;
	if (n_pairs < 65536) nt = 1;
	auto work = [&](int t) {
		const size_t lo = (size_t)n_pairs * t / nt, hi = (size_t)n_pairs * (t + 1) / nt;
		for (size_t i = lo; i < hi; ++i) {
			if (i + 8 < hi) { __builtin_prefetch(&recs[2 * (i + 8)]); __builtin_prefetch(&recs[2 * (i + 8) + 1]); }
			const nabwa_se_t &a = recs[2 * i].se, &b = recs[2 * i + 1].se;
			bin[i] = nabwa_isize_bin(2, a.mapQ, b.mapQ, a.pos, a.len, b.pos, b.len);
		}
	};
	if (nt == 1) work(0);
	else { std::vector<std::thread> th; for (int t = 0; t < nt; ++t) th.emplace_back(work, t); for (auto &x : th) x.join(); }
	for (int i = 0; i < n_pairs; ++i) if (bin[(size_t)i] >= 0) ++hist[bin[(size_t)i]];
	return NABWA_OK;
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

/* ---- pairing (bwape.c:180-293) ----------------------------------------------------------------------------------------
 * Input: every text position of every hit row of both ends, tagged pos << 32 | row << 1 | end (what finish_pair collects,
 * bam2bam.c:737-767), and the two bwt_aln1_t arrays.  The candidates are walked in position order; a reverse-strand hit is
 * the right end of a possible FR pair and is tried against the two most recent forward-strand hits of the other end.  A pair
 * is ranked by (10 x summed alignment scores + insert-size penalty, then a hash of the two positions); the best pair moves
 * the ends, the runner-up and the number of equally / less good pairs give the pair's mapping quality. */
namespace {

struct PairRank {                       /* smaller is better; `none` = no pair seen */
	uint32_t cost, tie; bool none;
	bool better_than(const PairRank &o) const { return o.none || (!none && (cost < o.cost || (cost == o.cost && tie < o.tie))); }
};

struct PairSweep {
	const nabwa_pe_end_t *end; const nabwa_aln1_t *const *rows; const nabwa_isize_t *ii;
	uint32_t longest_read; int max_isize;
	uint64_t recent[2][2];              /* per end: the last two forward-strand candidates, [1] the newer */
	bool have[2][2];
	PairRank best, second; uint64_t best_of_end[2];
	int n_best, n_worse;                /* pairs at the best cost / at any higher cost */

	const nabwa_aln1_t &row(uint64_t c) const { return rows[c & 1][(uint32_t)c >> 1]; }
	static uint32_t where(uint64_t c) { return (uint32_t)(c >> 32); }
	bool reverse(uint64_t c) const { return (row(c).info >> 24 & 1) != 0; }

	void forward_seen(uint64_t c) { const int e = (int)(c & 1); recent[e][0] = recent[e][1]; have[e][0] = have[e][1]; recent[e][1] = c; have[e][1] = true; }

	void try_pair(bool present, uint64_t left, uint64_t right)
	{
		if (!present || where(right) <= where(left)) return;
		const uint32_t span = where(right) + (uint32_t)end[right & 1].len - where(left);      /* outer distance, 32-bit as bwtint_t */
		if (span < longest_read) return;
		const bool fits = ii->high ? span <= ii->high_bayesian : span <= (uint32_t)max_isize;
		if (!fits) return;
		uint64_t c = (uint64_t)(row(right).score + row(left).score);
		c *= 10;
		if (ii->high) c += (int)(-4.343 * log(.5 * erfc(M_SQRT1_2 * fabs(span - ii->avg) / ii->std)) + .499);
		PairRank r; r.cost = (uint32_t)c; r.tie = (uint32_t)mix_u64((uint64_t)where(left) << 32 | where(right)); r.none = false;
		if (!best.none && r.cost == best.cost) ++n_best;
		else if (best.none || r.cost < best.cost) { n_worse += n_best; n_best = 1; }
		else ++n_worse;
		if (r.better_than(best)) { second = best; best = r; best_of_end[left & 1] = left; best_of_end[right & 1] = right; }
		else if (r.better_than(second)) second = r;
	}

	void reverse_seen(uint64_t c) { const int mate = 1 - (int)(c & 1); try_pair(have[mate][1], recent[mate][1], c); try_pair(have[mate][0], recent[mate][0], c); }

	/* the most a moved end's mapping quality can be */
	int moved_end_quality(int s_mm) const
	{
		if (n_best != 1) return 0;
		if (second.none) return 29;
		/* the reference subtracts the upper words of two 64-bit keys; with "none" out of the way that is cost - cost */
		const uint64_t gap = (uint64_t)second.cost - (uint64_t)best.cost;
		if (gap > (uint64_t)(s_mm * 10)) return 23;
		const int n = n_worse > 255 ? 255 : n_worse;
		const int lg = n > 0 ? (int)(4.343 * log((double)n) + 0.5) : 0;          /* g_log_n[n], bwase.c:613-617 */
		const int q = (int)(gap / 2) - lg;
		return q < 0 ? 0 : q;
	}
};

}

extern "C" int nabwa_pairing(nabwa_pe_end_t p[2], int n_hits, uint64_t *hits, const nabwa_aln1_t *rows0, const nabwa_aln1_t *rows1,
							 int max_isize, int s_mm, const nabwa_isize_t *ii)
{
	const nabwa_aln1_t *rows[2] = { rows0, rows1 };
	PairSweep sw;
	sw.end = p; sw.rows = rows; sw.ii = ii; sw.max_isize = max_isize;
	sw.longest_read = (uint32_t)(p[0].full_len > p[1].full_len ? p[0].full_len : p[1].full_len);
	for (int e = 0; e < 2; ++e) { sw.have[e][0] = sw.have[e][1] = false; sw.recent[e][0] = sw.recent[e][1] = 0; sw.best_of_end[e] = 0; }
	sw.best.none = sw.second.none = true; sw.best.cost = sw.best.tie = sw.second.cost = sw.second.tie = 0;
	sw.n_best = sw.n_worse = 0;
	std::sort(hits, hits + n_hits);
	for (int i = 0; i < n_hits; ++i) {
		if (sw.reverse(hits[i])) sw.reverse_seen(hits[i]);
		else sw.forward_seen(hits[i]);
	}
	if (sw.best.none) return 0;

	const int cap = sw.moved_end_quality(s_mm);
	bool stays[2];
	for (int e = 0; e < 2; ++e) stays[e] = p[e].pos == PairSweep::where(sw.best_of_end[e]) && p[e].strand == (int)sw.reverse(sw.best_of_end[e]);
	if (stays[0] && stays[1]) {
		if (p[0].mapQ > 0 && p[1].mapQ > 0) {
			const int sum = p[0].mapQ + p[1].mapQ;
			p[0].mapQ = p[1].mapQ = sum > 60 ? 60 : sum;
		} else {                                              /* end 0 first: end 1 then sees end 0's new value, as in the reference */
			for (int e = 0; e < 2; ++e)
				if (p[e].mapQ == 0) p[e].mapQ = cap + 7 < p[1 - e].mapQ ? cap + 7 : p[1 - e].mapQ;
		}
	} else if (stays[0] || stays[1]) {
		const int fixed = stays[0] ? 0 : 1, moved = 1 - fixed;
		p[moved].seQ = 0;
		p[moved].mapQ = p[fixed].mapQ > cap ? cap : p[fixed].mapQ;
	} else {
		const int q = cap - 20 < 0 ? 0 : cap - 20;
		p[0].seQ = p[1].seQ = 0;
		p[0].mapQ = p[1].mapQ = q;
	}
	int n_moved_with_quality = 0;
	for (int e = 0; e < 2; ++e) {                              /* the ends take the pair's places */
		const nabwa_aln1_t &r = sw.row(sw.best_of_end[e]);
		nabwa_pe_end_t &q = p[e];
		q.extra_flag |= 2;                                     /* SAM_FPP */
		if (stays[e]) continue;
		q.n_mm = r.info & 0xff; q.n_gapo = r.info >> 8 & 0xff; q.n_gape = r.info >> 16 & 0xff; q.strand = r.info >> 24 & 1;
		q.score = r.score; q.pos = PairSweep::where(sw.best_of_end[e]);
		if (q.mapQ > 0) ++n_moved_with_quality;
	}
	return n_moved_with_quality;
}
