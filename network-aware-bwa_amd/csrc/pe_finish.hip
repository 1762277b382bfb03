// pe_finish.hip -- the paired-end chain between the FM search and the BAM records (config 3).
//
// What the reference does per pair (bam2bam.c:683-811):
//   posn_pair   : per end bwa_aln2seq (drand48 in record order) + bwa_cal_pac_pos_core            -> nabwa_pe_posn
//   (barrier: insert-size histogram over all pairs, insert_size.c)                                 -> nabwa_isize_bin / _infer
//   finish_pair : enumerate the text positions of every hit row of both ends (bwt_sa), pairing(),
//                 multi-hit lists, bwa_paired_sw1 (mate rescue by local alignment), bwa_refine_gapped
//                 on both ends, then the flag / mate fields of bwa_update_bam1                     -> nabwa_pe_finish
// Here each step runs over the whole batch: all bwt_sa walks as GPU batches, all rescue alignments as ONE
// batch of local alignments, all gap refinements as ONE batch of global alignments; the floating-point
// decisions (windows, log-odds) stay on the host in double, in the reference's expression order.
#include <hip/hip_runtime.h>
#include <limits.h>
#include <math.h>
#include <chrono>
#include <thread>
#include "finish_common.hpp"

#define F_PD 1
#define F_PP 2
#define F_SU 4
#define F_MU 8
#define F_SR 16
#define F_MR 32
#define F_R1 64
#define F_R2 128
#define SW_MIN_MATCH_LEN 20       /* bwape.h:36 */
#define SW_MIN_MAPQ 17            /* bwape.h:37 */

static inline nabwa_pe_t &PE(nabwa_pe_t *out, int pair, int end) { return out[2 * (size_t)pair + end]; }

int nabwa_se_posn_strided(nabwa_index_t *ix, const nabwa_gap_opt_t *opt, int n, const int64_t *off, const int32_t *full_len,
						  const int32_t *n_aln, const nabwa_aln1_t *aln, const uint8_t *n_occ_v, uint64_t *rng48, void *out_base, size_t stride);      /* se_finish.hip */

/* posn_pair (bam2bam.c:683-703) for n_pairs pairs; records and reads are interleaved: index 2*pair + end. */
extern "C" int nabwa_pe_posn(nabwa_index_t *ix, const nabwa_gap_opt_t *opt, int n_pairs, const int64_t *off, const int32_t *full_len,
							 const int32_t *n_aln, const nabwa_aln1_t *aln, uint64_t *rng48, nabwa_pe_t *out)
{
	if (!ix || !opt || !rng48 || n_pairs < 0 || (n_pairs && (!off || !n_aln || !out))) return nabwa_fail(NABWA_EINVAL, "null argument");
	const int n = 2 * n_pairs;
	/* the single-end part of posn_pair is posn_singleton without other hits listed (bam2bam.c:688-700): hit choice on the caller's
	 * drand48 stream, the bwt_sa batch, mapQ -- done by the single-end chain's code on these records in place (its head is theirs) */
	int rc = nabwa_se_posn_strided(ix, opt, n, off, full_len, n_aln, aln, 0, rng48, out, sizeof(nabwa_pe_t));
	if (rc != NABWA_OK) return rc;
	fin_parallel(fin_threads((size_t)n), (size_t)n, [&](int, size_t lo, size_t hi) {
		for (size_t i = lo; i < hi; ++i) {
			nabwa_pe_t &r = out[i];
			r.se.seqid = -1;
			r.extra_flag = F_PD | ((i & 1) ? F_R2 : F_R1);
			r.m_seqid = -1; r.m_rpos = 0; r.isize = 0; r.am = 0; r.mapQ_paired = 0;
		}
	});
	return NABWA_OK;
}

/* finish_pair's position cache (my_hash, bam2bam.c:741-757): the text positions of a hit row of MIN_HASH_WIDTH = 1000 suffixes or more are
 * computed once per file -- keyed by the row's (k, l) ALONE, with the strand and the read length of the read that brought the row first --
 * and handed to every later read with the same (k, l) as they are.  A later read of another length on the reverse strand (or, should two
 * rows of the two indexes ever share their numbers, of the other strand) therefore pairs on positions that are not its own: not a mere
 * memo, and part of what `bwa bam2bam -t 1` writes.  Positions are a function of (k, l, strand, length), so the cache holds only the
 * strand and length first seen; rows are entered in record order. */
#include <unordered_map>
#define MIN_HASH_WIDTH 1000       /* bwape.h:31 */
struct nabwa_poscache { std::unordered_map<uint64_t, uint32_t> first; };           /* (k << 32 | l) -> strand bit << 31 | read length */
extern "C" nabwa_poscache_t *nabwa_poscache_create(void) { return new nabwa_poscache(); }
extern "C" void nabwa_poscache_destroy(nabwa_poscache_t *c) { delete c; }
extern "C" int64_t nabwa_poscache_size(const nabwa_poscache_t *c) { return c ? (int64_t)c->first.size() : 0; }

/* finish_pair enumerates the hits of a pair only when both ends are mapped and neither has more than max_occ hits (bam2bam.c:726-738) */
static inline bool pair_is_enumerated(const nabwa_se_t &e0, const nabwa_se_t &e1, const nabwa_aln1_t *a0, int n0, const nabwa_aln1_t *a1, int n1, int max_occ, uint32_t *n_rows)
{
	if (!((e0.type == 1 || e0.type == 2) && (e1.type == 1 || e1.type == 2))) return false;
	long long o0 = 0, o1 = 0;
	for (int k = 0; k < n0; ++k) o0 += (long long)a0[k].l - a0[k].k + 1;
	for (int k = 0; k < n1; ++k) o1 += (long long)a1[k].l - a1[k].k + 1;
	if (o0 > max_occ || o1 > max_occ) return false;
	if (n_rows) *n_rows = (uint32_t)(o0 + o1);
	return true;
}

/* The wide rows of the pairs first[0..n) (record index of end 0; ends adjacent) enter the cache in this order.  The batch front-end calls
 * it with a batch's pairs in record order before it finishes them read group by read group: who is first with a row is then settled as
 * the sequential reference settles it, whatever order the groups are taken in. */
void nabwa_poscache_register(nabwa_poscache_t *cache, int max_occ, int n, const int *first, const int32_t *n_aln, const int64_t *row0,
							 const nabwa_aln1_t *rows, const nabwa_pe_t *res)
{
	if (!cache) return;
	for (int t = 0; t < n; ++t) {
		const int i = first[t];
		bool wide = false;
		for (int64_t r = row0[i]; r < row0[i] + n_aln[i] + n_aln[i + 1] && !wide; ++r) wide = rows[r].l - rows[r].k + 1 >= MIN_HASH_WIDTH;
		if (!wide) continue;
		if (!pair_is_enumerated(res[i].se, res[i + 1].se, rows + row0[i], n_aln[i], rows + row0[i + 1], n_aln[i + 1], max_occ, 0)) continue;
		for (int j = 0; j < 2; ++j)
			for (int k = 0; k < n_aln[i + j]; ++k) {
				const nabwa_aln1_t &r = rows[row0[i + j] + k];
				if (r.l - r.k + 1 >= MIN_HASH_WIDTH) cache->first.emplace((uint64_t)r.k << 32 | r.l, (r.info >> 24 & 1) << 31 | (uint32_t)res[i + j].se.len);
			}
	}
}

/* one mate-rescue attempt: align end `k` of pair `pair` inside [beg, beg+reglen) next to its mate (bwape.c:562-583) */
struct SwJob { int pair, k; int64_t beg; int ref_n; bool fwd; };

extern "C" int nabwa_pe_finish(nabwa_index_t *ix, const nabwa_gap_opt_t *opt, const nabwa_pe_opt_t *popt, const nabwa_isize_t *ii,
							   int n_pairs, const int64_t *off, const uint8_t *seq, const uint8_t *rseq, const int32_t *n_aln,
							   const nabwa_aln1_t *aln, nabwa_pe_t *out, uint64_t n_tot[2], uint64_t n_mapped[2])
{
	return nabwa_pe_finish_cached(ix, opt, popt, ii, n_pairs, off, seq, rseq, n_aln, aln, out, n_tot, n_mapped, 0);
}

extern "C" int nabwa_pe_finish_cached(nabwa_index_t *ix, const nabwa_gap_opt_t *opt, const nabwa_pe_opt_t *popt, const nabwa_isize_t *ii,
									  int n_pairs, const int64_t *off, const uint8_t *seq, const uint8_t *rseq, const int32_t *n_aln,
									  const nabwa_aln1_t *aln, nabwa_pe_t *out, uint64_t n_tot[2], uint64_t n_mapped[2], nabwa_poscache_t *cache)
{
	if (!ix || !opt || !popt || !ii || n_pairs < 0 || (n_pairs && (!off || !seq || !rseq || !n_aln || !out))) return nabwa_fail(NABWA_EINVAL, "null argument");
	if (!ix->ref) return nabwa_fail(NABWA_EINVAL, "index has no reference attached (nabwa_index_attach_reference)");
	if (popt->type != 1) return nabwa_fail(NABWA_EINVAL, "only BWA_PET_STD pairs are supported (no colour space)");
	if (popt->n_multi < 0 || popt->N_multi < 0 || popt->n_multi > NABWA_MAX_MULTI || popt->N_multi > NABWA_MAX_MULTI)
		return nabwa_fail(NABWA_EINVAL, "n_multi / N_multi outside 0..16");
	const nabwa_reference *R = ix->ref;
	const uint32_t rlen = ix->bwt[1].seq_len;
	const int n = 2 * n_pairs;
	const bool timing = getenv("NABWA_TIMING") != 0;
	auto now = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
	double t0 = now(), t1, t2, t3, t4;
	std::vector<size_t> a_off((size_t)n + 1, 0);
	for (int i = 0; i < n; ++i) a_off[i + 1] = a_off[i] + (size_t)n_aln[i];
	uint64_t tot_dummy[2] = { 0, 0 }, map_dummy[2] = { 0, 0 };
	if (!n_tot) n_tot = tot_dummy;
	if (!n_mapped) n_mapped = map_dummy;
	size_t n_hit_rows = 0, n_sw = 0, n_refine = 0, n_cand = 0;

	/* ---- A. pairing: text positions of every hit row of both ends, chunked so one bwt_sa batch stays bounded
	 *         (bam2bam.c:726-770; the position cache there: `cache`, above) */
	const size_t CHUNK_ROWS = 1u << 25;
	double ta[3] = { 0, 0, 0 }, tc[4] = { 0, 0, 0, 0 };
	/* which pairs are paired here -- both ends mapped, neither with more than max_occ hit rows -- and how many rows each brings:
	 * the records are read by all threads, the chunks are then cut from the counts alone */
	std::vector<uint32_t> prow((size_t)(n_pairs ? n_pairs : 1), 0);
	fin_parallel(fin_threads((size_t)n_pairs), (size_t)n_pairs, [&](int, size_t p_lo, size_t p_hi) {
		for (size_t pr = p_lo; pr < p_hi; ++pr) {
			if (pr + 8 < p_hi) { __builtin_prefetch(&PE(out, (int)pr + 8, 0)); __builtin_prefetch(&PE(out, (int)pr + 8, 1)); }      /* (3 KB records: the next ones asked for ahead) */
			uint32_t rows_here = 0;
			if (pair_is_enumerated(PE(out, pr, 0).se, PE(out, pr, 1).se, aln + a_off[2 * pr], n_aln[2 * pr], aln + a_off[2 * pr + 1], n_aln[2 * pr + 1], popt->max_occ, &rows_here))
				prow[pr] = rows_here;
		}
	});
	/* the position cache, in record order: for every wide row of a paired pair the (strand, length) its positions are computed with --
	 * the row's own when it is the first with its (k, l), else those of the first.  Only rows that end up with other values than their
	 * own are noted (`foreign`, by row index); wide rows are rare, so the pairs that have one are found by all threads first. */
	std::unordered_map<size_t, uint32_t> foreign;
	if (cache) {
		std::vector<uint8_t> wide((size_t)(n_pairs ? n_pairs : 1), 0);
		fin_parallel(fin_threads((size_t)n_pairs), (size_t)n_pairs, [&](int, size_t p_lo, size_t p_hi) {
			for (size_t pr = p_lo; pr < p_hi; ++pr) {
				if (!prow[pr]) continue;
				for (size_t r = a_off[2 * pr]; r < a_off[2 * pr + 2]; ++r) if (aln[r].l - aln[r].k + 1 >= MIN_HASH_WIDTH) { wide[pr] = 1; break; }
			}
		});
		for (size_t pr = 0; pr < (size_t)n_pairs; ++pr) {
			if (!wide[pr]) continue;
			for (int j = 0; j < 2; ++j)
				for (size_t r = a_off[2 * pr + j]; r < a_off[2 * pr + j + 1]; ++r) {
					if (aln[r].l - aln[r].k + 1 < MIN_HASH_WIDTH) continue;
					const uint32_t mine = (aln[r].info >> 24 & 1) << 31 | (uint32_t)PE(out, pr, j).se.len;
					auto it = cache->first.emplace((uint64_t)aln[r].k << 32 | aln[r].l, mine).first;
					if (it->second != mine) foreign[r] = it->second;
				}
		}
	}
	auto row_strand_len = [&](size_t r, uint32_t own_len, uint32_t &a, uint32_t &len) {      /* what the positions of hit row r are computed with */
		a = aln[r].info >> 24 & 1; len = own_len;
		if (!foreign.empty()) { auto it = foreign.find(r); if (it != foreign.end()) { a = it->second >> 31; len = it->second & 0x7fffffffu; } }
	};
	for (int p0 = 0; p0 < n_pairs;) {
		const double tA0 = now();
		std::vector<uint8_t> which; std::vector<uint32_t> rows; std::vector<size_t> pair_lo; std::vector<int> pairs;
		int p1 = p0;
		size_t n_rows = 0;
		for (; p1 < n_pairs && (n_rows < CHUNK_ROWS || pairs.empty()); ++p1) {
			if (!prow[(size_t)p1]) continue;
			pairs.push_back(p1); pair_lo.push_back(n_rows); n_rows += prow[(size_t)p1];
		}
		which.resize(n_rows); rows.resize(n_rows);
		fin_parallel(fin_threads(pairs.size()), pairs.size(), [&](int, size_t t_lo, size_t t_hi) {
			for (size_t t = t_lo; t < t_hi; ++t) {
				const int pr = pairs[t];
				size_t u = pair_lo[t];
				for (int j = 0; j < 2; ++j) {
					const nabwa_aln1_t *A = aln + a_off[2 * (size_t)pr + j];
					for (int k = 0; k < n_aln[2 * pr + j]; ++k) {
						uint32_t a, len_unused; row_strand_len(a_off[2 * (size_t)pr + j] + k, 0, a, len_unused);
						for (uint32_t l = A[k].k; ; ++l) { which[u] = a ? 0 : 1; rows[u] = l; ++u; if (l == A[k].l) break; }
					}
				}
			}
		});
		pair_lo.push_back(rows.size());
		n_hit_rows += rows.size();
		const double tA1 = now();
		std::vector<uint32_t> sa(rows.size());
		if (!rows.empty()) { int r = nabwa_sa_lookup(ix, (int)rows.size(), which.data(), rows.data(), sa.data()); if (r != NABWA_OK) return r; }
		const double tA2 = now();
		fin_parallel(fin_threads(pairs.size()), pairs.size(), [&](int, size_t t_lo, size_t t_hi) {      /* pairs are independent of each other */
		std::vector<uint64_t> hits;
		for (size_t t = t_lo; t < t_hi; ++t) {
			if (t + 8 < t_hi) { const int f = pairs[t + 8]; __builtin_prefetch(&PE(out, f, 0), 1); __builtin_prefetch(&PE(out, f, 1), 1); }
			const int pr = pairs[t];
			hits.clear();
			size_t u = pair_lo[t];
			for (int j = 0; j < 2; ++j) {
				const nabwa_aln1_t *A = aln + a_off[2 * (size_t)pr + j];
				for (int k = 0; k < n_aln[2 * pr + j]; ++k) {
					uint32_t a, len; row_strand_len(a_off[2 * (size_t)pr + j] + k, (uint32_t)PE(out, pr, j).se.len, a, len);
					for (uint32_t l = A[k].k; ; ++l, ++u) {
						const uint64_t x = (uint64_t)(uint32_t)(which[u] == 0 ? sa[u] : rlen - (sa[u] + len));
						hits.push_back(x << 32 | (uint64_t)(uint32_t)(k << 1) | (uint64_t)j);
						if (l == A[k].l) { ++u; break; }
					}
				}
			}
			nabwa_pe_end_t e[2];
			for (int j = 0; j < 2; ++j) {
				const nabwa_pe_t &r = PE(out, pr, j); const nabwa_se_t &s = r.se;
				e[j] = { s.pos, s.strand, s.mapQ, s.seQ, s.len, s.full_len, s.n_mm, s.n_gapo, s.n_gape, s.score, r.extra_flag };
			}
			nabwa_pairing(e, (int)hits.size(), hits.data(), aln + a_off[2 * (size_t)pr], aln + a_off[2 * (size_t)pr + 1], popt->max_isize, opt->s_mm, ii);
			for (int j = 0; j < 2; ++j) {
				nabwa_pe_t &r = PE(out, pr, j); nabwa_se_t &s = r.se;
				s.pos = e[j].pos; s.strand = e[j].strand; s.mapQ = e[j].mapQ; s.seQ = e[j].seQ; s.n_mm = e[j].n_mm; s.n_gapo = e[j].n_gapo;
				s.n_gape = e[j].n_gape; s.score = e[j].score; r.extra_flag = e[j].extra_flag;
			}
		}
		});
		ta[0] += tA1 - tA0; ta[1] += tA2 - tA1; ta[2] += now() - tA2;
		p0 = p1;
	}
	t1 = now();

	/* ---- B. multi-hit lists and their positions (bam2bam.c:773-790) */
	{
		std::vector<uint8_t> which; std::vector<uint32_t> rows; std::vector<int> look_rec, look_multi;
		const int ntb = fin_threads((size_t)n_pairs);
		struct MultiPart { std::vector<uint8_t> which; std::vector<uint32_t> rows; std::vector<int> look_rec, look_multi; };
		std::vector<MultiPart> parts((size_t)ntb);
		fin_parallel(ntb, (size_t)n_pairs, [&](int slice, size_t p_lo, size_t p_hi) {
		MultiPart &M = parts[(size_t)slice];
		std::vector<uint8_t> &which = M.which; std::vector<uint32_t> &rows = M.rows; std::vector<int> &look_rec = M.look_rec, &look_multi = M.look_multi;
		for (int pr = (int)p_lo; pr < (int)p_hi; ++pr)
			for (int j = 0; j < 2; ++j) {
				if (j == 0 && pr + 8 < (int)p_hi) for (int e = 0; e < 2; ++e) { const nabwa_pe_t *const f = &PE(out, pr + 8, e); __builtin_prefetch(f, 1); __builtin_prefetch(&f->se.n_multi, 1); __builtin_prefetch(&f->extra_flag); }
				nabwa_pe_t &r = PE(out, pr, j); nabwa_se_t &s = r.se;
				s.n_multi = 0;
				if (s.type == 0) continue;
				int nm = popt->n_multi;
				if (!(r.extra_flag & F_PP) && PE(out, pr, 1 - j).se.type != 0)
					nm = (int64_t)s.c1 + (int64_t)s.c2 - 1 > popt->N_multi ? popt->n_multi : popt->N_multi;
				list_multi(s, n_aln[2 * pr + j], aln + a_off[2 * (size_t)pr + j], nm);
				for (int z = 0; z < s.n_multi; ++z) {
					which.push_back(s.multi[z].strand ? 0 : 1); rows.push_back(s.multi[z].pos); look_rec.push_back(2 * pr + j); look_multi.push_back(z);
				}
			}
		});
		for (MultiPart &M : parts) {
			which.insert(which.end(), M.which.begin(), M.which.end()); rows.insert(rows.end(), M.rows.begin(), M.rows.end());
			look_rec.insert(look_rec.end(), M.look_rec.begin(), M.look_rec.end()); look_multi.insert(look_multi.end(), M.look_multi.begin(), M.look_multi.end());
		}
		std::vector<uint32_t> sa(rows.size());
		if (!rows.empty()) { int r = nabwa_sa_lookup(ix, (int)rows.size(), which.data(), rows.data(), sa.data()); if (r != NABWA_OK) return r; }
		for (size_t t = 0; t < rows.size(); ++t) {
			nabwa_se_t &s = out[look_rec[t]].se;
			s.multi[look_multi[t]].pos = which[t] == 0 ? sa[t] : rlen - (sa[t] + (uint32_t)s.len);
		}
	}
	t2 = now();

	/* ---- C. mate rescue (bwa_paired_sw1, bwape.c:519-633; bam2bam calls it unconditionally).
	 *         Quirk F4 kept: a pair with an unmapped end returns before anything is changed. */
	{
		static const int maq[25] = { 11,-19,-19,-19,-13, -19,11,-19,-19,-13, -19,-19,11,-19,-13, -19,-19,-19,11,-13, -13,-13,-13,-13,-13 };
		std::vector<SwJob> jobs; std::vector<int> cand;
		std::vector<int64_t> ro(1, 0), qo(1, 0); std::vector<uint8_t> rb, qb;
		std::vector<int64_t> begs((size_t)n, 0);                         /* beg[k] of every attempted end, by record index */
		std::vector<int> job_of((size_t)n, -1);
		{	/* which pairs are tried: unpaired, one end with a high mapping quality, both ends mapped (slices of pairs in threads) */
			const int ntc = fin_threads((size_t)n_pairs);
			std::vector<std::vector<int>> cparts((size_t)ntc);
			std::vector<uint64_t> tot0((size_t)ntc, 0), tot1((size_t)ntc, 0);
			fin_parallel(ntc, (size_t)n_pairs, [&](int slice, size_t p_lo, size_t p_hi) {
				for (int pr = (int)p_lo; pr < (int)p_hi; ++pr) {
					if (pr + 8 < (int)p_hi) for (int e = 0; e < 2; ++e) { const nabwa_pe_t *const f = &PE(out, pr + 8, e); __builtin_prefetch(f); __builtin_prefetch(&f->extra_flag); }
					const nabwa_pe_t &r0 = PE(out, pr, 0), &r1 = PE(out, pr, 1);
					if (!((r0.se.mapQ >= SW_MIN_MAPQ || r1.se.mapQ >= SW_MIN_MAPQ) && (r0.extra_flag & F_PP) == 0)) continue;
					const int single = (r0.se.type == 0 || r1.se.type == 0) ? 1 : 0;
					if (single) ++tot1[(size_t)slice]; else { ++tot0[(size_t)slice]; cparts[(size_t)slice].push_back(pr); }
				}
			});
			for (int t = 0; t < ntc; ++t) { n_tot[0] += tot0[(size_t)t]; n_tot[1] += tot1[(size_t)t]; cand.insert(cand.end(), cparts[(size_t)t].begin(), cparts[(size_t)t].end()); }
		}
		tc[0] = now() - t2; n_cand = cand.size();
		for (int pr : cand) {
			for (int k = 0; k < 2; ++k) {
				const nabwa_se_t &ref = PE(out, pr, 1 - k).se, &mate = PE(out, pr, k).se;
				int64_t a, b;
				if (ref.strand == 0) {                                   /* mate on the reverse strand, to the right (__set_rght_coor) */
					a = (int64_t)((double)(int64_t)ref.pos + ii->avg - 3 * ii->std - mate.len * 1.5);
					b = (int64_t)((double)a + 6 * ii->std + (double)(2 * mate.len));
					if (a < (int64_t)ref.pos + ref.len) a = (int64_t)(uint32_t)(ref.pos + (uint32_t)ref.len);
					if (b > R->l_pac) b = R->l_pac;
				} else {                                                 /* mate on the forward strand, to the left (__set_left_coor) */
					a = (int64_t)((double)((int64_t)ref.pos + ref.len) - ii->avg - 3 * ii->std - mate.len * 0.5);
					b = (int64_t)((double)a + 6 * ii->std + (double)(2 * mate.len));
					if (a < 0) a = 0;
					if (b > (int64_t)ref.pos) b = ref.pos;
				}
				begs[2 * (size_t)pr + k] = a;
				/* bwa_sw_core's guards (bwape.c:443-447) */
				const int reglen = (int)(b - a), len = mate.len;
				const uint32_t l_pac = (uint32_t)R->l_pac;
				if (reglen < SW_MIN_MATCH_LEN || (int64_t)l_pac - a < len) continue;
				const uint8_t *src = (ref.strand == 0 ? rseq : seq) + off[2 * pr + k];
				uint32_t x = 0;
				for (int z = 0; z < len; ++z) if (src[z] >= 4) ++x;
				if ((float)x / len >= 0.25 || len - (int)x < SW_MIN_MATCH_LEN) continue;
				int l = 0;
				for (uint32_t z = (uint32_t)a; l < reglen && z < l_pac; ++z, ++l) rb.push_back((uint8_t)pac_at(R, z));
				if (ref.strand == 0) qb.insert(qb.end(), src, src + len);
				else for (int z = len - 1; z >= 0; --z) qb.push_back(src[z]);        /* ->seq is stored reversed */
				ro.push_back((int64_t)rb.size()); qo.push_back((int64_t)qb.size());
				job_of[2 * (size_t)pr + k] = (int)jobs.size();
				jobs.push_back({ pr, k, a, l, ref.strand != 0 });
			}
		}
		tc[1] = now() - t2;
		n_sw = jobs.size();
		const int MAXC = NABWA_MAX_CIGAR - 2;
		std::vector<int32_t> sc(jobs.size()), co(jobs.size() * 4), nc(jobs.size()); std::vector<uint32_t> c32(jobs.size() * (size_t)MAXC);
		if (!jobs.empty()) {
			rb.push_back(0); qb.push_back(0);
			int r = nabwa_local_align(ix->device, (int)jobs.size(), ro.data(), rb.data(), qo.data(), qb.data(), 26, 9, maq, 50, 1,
									  sc.data(), co.data(), 0, nc.data(), c32.data(), MAXC);
			if (r != NABWA_OK) return r;
		}
		tc[2] = now() - t2;
		const int sw_isize_term = (int)(-4.343 * log(.5 * erfc(M_SQRT1_2 * 1.5) + .499));     /* bwape.c:593 */
		for (int pr : cand) {
			int n_cig[2] = { 0, 0 }, mq_adjust[2] = { 255, 255 }; uint16_t cig[2][NABWA_MAX_CIGAR]; uint32_t cnt[2] = { 0, 0 };
			int64_t beg[2]; bool have[2] = { false, false };
			for (int k = 0; k < 2; ++k) {
				beg[k] = begs[2 * (size_t)pr + k];
				const int t = job_of[2 * (size_t)pr + k];
				if (t < 0) continue;
				const nabwa_se_t &mate = PE(out, pr, k).se;
				const int len = mate.len;
				if (sc[t] < 0 || nc[t] < 1) continue;                       /* aln_local_core found nothing (or its "potential bug" branch) */
				if (nc[t] > MAXC) return nabwa_fail(NABWA_ECAP, "rescue CIGAR longer than NABWA_MAX_CIGAR");
				int m = nc[t]; uint16_t *cg = cig[k];
				int64_t x = 0, y = 0;
				for (int z = 0; z < m; ++z) {
					const uint32_t c = c32[(size_t)t * MAXC + z]; const int op = c & 0xf, l = c >> 4;
					cg[z] = CMAKE(op, l);
					if (op == 0) { x += l; y += l; } else if (op == 2) x += l; else y += l;
				}
				if (x < SW_MIN_MATCH_LEN || y < SW_MIN_MATCH_LEN) continue;
				/* first cell of the path (1-based i on the window, j on the read); a leading gap sits on row/column 0 of the sub-matrix */
				int pi = co[4 * t], pj = co[4 * t + 1];
				if (COP(cg[0]) == 1) pi -= 1; else if (COP(cg[0]) == 2) pj -= 1;
				const int end_j = co[4 * t + 3];
				beg[k] += (pi ? pi : 1) - 1;
				const int start = (pj ? pj : 1) - 1;
				if (start) { memmove(cg + 1, cg, 2 * (size_t)m); cg[0] = CMAKE(3, start); ++m; }
				if (end_j < len) cg[m++] = CMAKE(3, len - end_j);
				{	/* mismatches and gaps of the new alignment (bwape.c:495-513) */
					int n_mm = 0, n_gapo = 0, n_gape = 0;
					const uint8_t *rs = rb.data() + ro[t], *qs = qb.data() + qo[t];
					int64_t xx = pi ? pi - 1 : 0, yy = pj ? pj - 1 : 0;
					for (int z = 0; z < m; ++z) {
						const int op = COP(cg[z]), l = CLEN(cg[z]);
						if (op == 0) {
							for (int w = 0; w < l; ++w) if (rs[xx + w] < 4 && qs[yy + w] < 4 && rs[xx + w] != qs[yy + w]) ++n_mm;
							xx += l; yy += l;
						} else if (op == 2) { xx += l; ++n_gapo; n_gape += l - 1; }
						else if (op == 1) { yy += l; ++n_gapo; n_gape += l - 1; }
					}
					cnt[k] = (uint32_t)n_mm << 16 | n_gapo << 8 | n_gape;
				}
				n_cig[k] = m; have[k] = true;
				{	/* is the rescued placement more likely than the one the search found? (bwape.c:584-600) */
					int clip = 0;
					if (COP(cg[0]) == 3) clip += CLEN(cg[0]);
					if (COP(cg[m - 1]) == 3) clip += CLEN(cg[m - 1]);
					int s_old = (int)((mate.n_mm * 9 + mate.n_gapo * 13 + mate.n_gape * 2) / 3. * 8. + .499);
					int s_new = (int)(((cnt[k] >> 16) * 9 + (cnt[k] >> 8 & 0xff) * 13 + (cnt[k] & 0xff) * 2 + (uint32_t)clip * 3) / 3. * 8. + .499);
					const double so = (double)s_old + -4.343 * log(ii->ap_prior / R->l_pac);
					s_old = (so > -2147483649.0 && so < 2147483648.0) ? (int)so : INT_MIN;   /* ap_prior 0 (no estimate): cvttsd2si gives INT_MIN */
					s_new += sw_isize_term;
					if (s_old < s_new) { mq_adjust[k] = (int)((uint32_t)s_new - (uint32_t)s_old); have[k] = false; n_cig[k] = 0; }
					else mq_adjust[k] = s_old - s_new;
				}
			}
			int k = -1, mapQ = 0;
			nabwa_pe_t *p[2] = { &PE(out, pr, 0), &PE(out, pr, 1) };
			if (have[0] && have[1]) { k = p[0]->se.mapQ < p[1]->se.mapQ ? 0 : 1; mapQ = abs(p[1]->se.mapQ - p[0]->se.mapQ); }
			else if (have[0]) { k = 0; mapQ = p[1]->se.mapQ; }
			else if (have[1]) { k = 1; mapQ = p[0]->se.mapQ; }
			if (k >= 0 && (int64_t)p[k]->se.pos != beg[k]) {
				++n_mapped[0];
				nabwa_se_t &fix = p[k]->se, &ref = p[1 - k]->se;
				int tmp = ref.mapQ - fix.mapQ / 2 - 8;
				if (tmp <= 0) tmp = 1;
				if (mapQ > tmp) mapQ = tmp;
				fix.mapQ = ref.mapQ = mapQ & 0xff;
				fix.seQ = ref.seQ = ref.seQ < mapQ ? ref.seQ : (mapQ & 0xff);
				if (fix.mapQ > mq_adjust[k]) fix.mapQ = mq_adjust[k] & 0xff;
				if (fix.seQ > mq_adjust[k]) fix.seQ = mq_adjust[k] & 0xff;
				fix.n_cigar = n_cig[k]; memcpy(fix.cigar, cig[k], 2 * (size_t)n_cig[k]);
				fix.type = 3;                                          /* BWA_TYPE_MATESW */
				fix.pos = (uint32_t)beg[k];
				fix.seQ = ref.seQ;
				fix.strand = 1 - ref.strand;
				fix.n_mm = cnt[k] >> 16 & 0xff; fix.n_gapo = cnt[k] >> 8 & 0xff; fix.n_gape = cnt[k] & 0xff;
				p[0]->extra_flag |= F_PP; p[1]->extra_flag |= F_PP;
			}
		}
	}
	t3 = now();

	/* ---- D. gap refinement of both ends and their multi hits, one GPU batch (bwa_refine_gapped, bwase.c:356-381) */
	{
		int r = refine_batch(ix, out, sizeof(nabwa_pe_t), n, off, seq, rseq, &n_refine);
		if (r != NABWA_OK) return r;
	}
	t4 = now();

	/* ---- E. MD / NM / trimmed tail per end, then the flag and mate fields (bwase.c:399-419, bam2bam.c:430-525) */
	int md_over = 0;
	auto phaseE = [&](int lo, int hi) {
		std::vector<uint8_t> fwd;
		for (int pr = lo; pr < hi; ++pr) {
			/* (the pieces of a record lie far apart -- head, MD field and tail of a 3 KB record, its window in the packed reference: asked for ahead,
			 * as in se_finish.hip) */
			if (pr + 8 < hi) for (int j = 0; j < 2; ++j) { const nabwa_pe_t *const f = &PE(out, pr + 8, j); __builtin_prefetch(f, 1); __builtin_prefetch(f->se.md, 1); __builtin_prefetch(&f->se.flag, 1); __builtin_prefetch(&f->extra_flag, 1); }
			if (pr + 4 < hi) for (int j = 0; j < 2; ++j) { const nabwa_se_t &f = PE(out, pr + 4, j).se; if (f.type) { const uint8_t *const w = R->pac.data() + (f.pos >> 2); __builtin_prefetch(w); __builtin_prefetch(w + 32); } }
			for (int j = 0; j < 2; ++j) {
				nabwa_se_t &s = PE(out, pr, j).se;
				if (s.type != 0 && !md_and_trim(R, s, seq + off[2 * pr + j], rseq + off[2 * pr + j], fwd)) md_over = 1;
			}
			for (int j = 0; j < 2; ++j) {                              /* end 0 first, as bam2bam.c:804-805 */
				nabwa_pe_t &r = PE(out, pr, j); nabwa_se_t &p = r.se;
				const nabwa_se_t &mate = PE(out, pr, 1 - j).se;
				r.mapQ_paired = p.mapQ;
				if (p.type == 0 && mate.type == 0) {
					p.flag = (r.extra_flag & ~(F_PP | F_MU)) | F_SU | F_MU;
					p.seqid = -1; p.rpos = 0; p.nn = 0; p.xt = 0; r.m_seqid = -1; r.m_rpos = 0; r.isize = 0; r.am = 0;
					continue;
				}
				int jlen;
				if (p.type == 0) { p.pos = mate.pos; p.strand = mate.strand; r.extra_flag |= F_SU; jlen = 1; }
				else jlen = (int)(rec_pos_end(p) - p.pos);
				int flag = r.extra_flag, nn, seqid, m_seqid = -1;
				nn = pac2real(R, p.pos, jlen, &seqid);
				if (p.type != 0 && (int64_t)p.pos + jlen - R->anns[seqid].offset > R->anns[seqid].len) { flag |= F_SU; flag &= ~F_PP; p.mapQ = 0; }
				if (p.strand) flag |= F_SR;
				p.seqid = seqid; p.rpos = (int64_t)p.pos - R->anns[seqid].offset + 1;
				if (mate.type != 0) {
					r.am = mate.seQ < p.seQ ? mate.seQ : p.seQ;
					nn += pac2real(R, mate.pos, mate.len, &m_seqid);
					const int m_j = (int)(rec_pos_end(mate) - mate.pos);
					if ((int64_t)mate.pos + m_j - R->anns[m_seqid].offset > R->anns[m_seqid].len) { flag |= F_MU; flag &= ~F_PP; }
					if (mate.strand) flag |= F_MR;
					r.m_seqid = m_seqid; r.m_rpos = (int64_t)mate.pos - R->anns[m_seqid].offset + 1;
					if (p.type == 0 || seqid != m_seqid) r.isize = 0;
					else r.isize = (mate.strand ? rec_pos_end(mate) : (int64_t)mate.pos) - (p.strand ? rec_pos_end(p) : (int64_t)p.pos);
				} else { flag |= F_MU; flag &= ~F_PP; r.am = 0; r.m_seqid = seqid; r.m_rpos = p.rpos; r.isize = 0; }
				p.flag = flag; p.nn = nn;
				p.xt = p.type == 0 ? 0 : (nn > 10 ? 'N' : "NURM"[p.type]);
			}
		}
	};
	{
		int nt = (int)std::thread::hardware_concurrency(); if (nt < 1) nt = 1; if (nt > 16) nt = 16;
		if (getenv("NABWA_HOST_THREADS")) nt = std::max(1, atoi(getenv("NABWA_HOST_THREADS")));
		if (n_pairs < 4096) nt = 1;
		std::vector<std::thread> th;
		for (int t = 0; t < nt; ++t) th.emplace_back(phaseE, (int)((int64_t)n_pairs * t / nt), (int)((int64_t)n_pairs * (t + 1) / nt));
		for (auto &x : th) x.join();
	}
	if (md_over) return nabwa_fail(NABWA_ECAP, "MD string longer than NABWA_MAX_MD");
	if (timing) fprintf(stderr, "[nabwa] pe_finish pairing: rows collected %.3f s, bwt_sa %.3f s, pairing %.3f s; rescue: scan %.3f s, windows %.3f s (%zu candidates), local alignments %.3f s, applied %.3f s\n",
						ta[0], ta[1], ta[2], tc[0], tc[1] - tc[0], n_cand, tc[2] - tc[1], (t3 - t2) - tc[2]);
	if (timing) fprintf(stderr, "[nabwa] pe_finish %d pairs: pairing (%zu hit rows) %.3f s, multi %.3f s, mate rescue (%zu alignments) %.3f s, "
						"refinement (%zu jobs) %.3f s, md/flags %.3f s\n", n_pairs, n_hit_rows, t1 - t0, t2 - t1, n_sw, t3 - t2, n_refine, t4 - t3, now() - t4);
	return NABWA_OK;
}
