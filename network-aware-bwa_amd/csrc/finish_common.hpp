// finish_common.hpp -- host pieces shared by the single-end and paired-end finishing chains.
// Records are nabwa_se_t laid out with a caller-given stride (nabwa_pe_t starts with a nabwa_se_t).
#pragma once
#include <sys/time.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <memory>
#include <string>
#include <thread>
#include <vector>
#include "../../include/nabwa.h"
#include "nabwa_internal.hpp"

/* ------------------------------------------------------------------ small host pieces */

static inline double rng48_next(uint64_t *x)          /* drand48: X' = 0x5DEECE66D X + 0xB mod 2^48, result X'/2^48 */
{
	*x = (*x * 0x5DEECE66DULL + 0xBULL) & 0xFFFFFFFFFFFFULL;
	return (double)*x * (1.0 / 281474976710656.0);
}

static inline int pac_at(const nabwa_reference *R, int64_t k) { return R->pac[k >> 2] >> ((~k & 3) << 1) & 3; }

/* the four bases of a .pac byte as four bytes, the first base lowest: eight reference bases are compared with eight read codes
 * (one byte each) in one 64-bit XOR -- the MD string of a read that matches costs an eighth of the base-by-base walk */
struct PacExpand { uint32_t t[256]; PacExpand() { for (int b = 0; b < 256; ++b) t[b] = (uint32_t)(b >> 6 & 3) | (uint32_t)(b >> 4 & 3) << 8 | (uint32_t)(b >> 2 & 3) << 16 | (uint32_t)(b & 3) << 24; } };
static inline const uint32_t *pac_expand() { static const PacExpand x; return x.t; }
/* how many of the next n bases (n >= 0) from reference position pos on equal the read codes q[0..] before the first that does
 * not (an N in the read never equals); pos + n <= l_pac and no .amb hole in the stretch are the caller's */
static inline int pac_match_run(const nabwa_reference *R, int64_t pos, const uint8_t *q, int n)
{
	const uint32_t *const X = pac_expand();
	int z = 0;
	while (z < n && ((pos + z) & 3)) { if (pac_at(R, pos + z) != q[z]) return z; ++z; }
	for (; z + 8 <= n; z += 8) {
		const uint8_t *pb = R->pac.data() + ((pos + z) >> 2);
		const uint64_t r = (uint64_t)X[pb[0]] | (uint64_t)X[pb[1]] << 32;
		uint64_t qq; memcpy(&qq, q + z, 8);
		const uint64_t x = r ^ qq;
		if (x) return z + (__builtin_ctzll(x) >> 3);
	}
	while (z < n) { if (pac_at(R, pos + z) != q[z]) return z; ++z; }
	return n;
}

/* base of the reference at pos with the ambiguity codes of the .amb holes restored (bwase.c:239-251) */
static inline int ref_char(const nabwa_reference *R, int64_t pos)
{
	size_t lo = 0, hi = R->holes.size();
	while (lo < hi) {
		size_t mid = (lo + hi) / 2;
		if (pos >= R->holes[mid].offset + R->holes[mid].len) lo = mid + 1;
		else if (pos < R->holes[mid].offset) hi = mid;
		else return R->holes[mid].amb;
	}
	return pac_at(R, pos);
}

/* bns_coor_pac2real (bntseq.c:272-306): contig of a position and the number of ambiguous bases under [pos, pos+len) */
static inline int pac2real(const nabwa_reference *R, int64_t pos, int len, int *seqid)
{
	int left = 0, mid = 0, right = (int)R->anns.size(), nn = 0;
	while (left < right) {
		mid = (left + right) >> 1;
		if (pos >= R->anns[mid].offset) {
			if (mid == (int)R->anns.size() - 1) break;
			if (pos < R->anns[mid + 1].offset) break;
			left = mid + 1;
		} else right = mid;
	}
	*seqid = mid;
	left = 0; right = (int)R->holes.size();
	while (left < right) {
		const int m = (left + right) >> 1; const nabwa_hole &h = R->holes[m];
		if (pos >= h.offset + h.len) left = m + 1;
		else if (pos + len <= h.offset) right = m;
		else {
			if (pos >= h.offset) nn += h.offset + h.len < pos + len ? (int)(h.offset + h.len - pos) : len;
			else nn += h.offset + h.len < pos + len ? h.len : len - (int)(h.offset - pos);
			break;
		}
	}
	return nn;
}

#define COP(c) ((c) >> 14)
#define CLEN(c) ((c) & 0x3fff)
#define CMAKE(op, len) ((uint16_t)((op) << 14 | (len)))

/* one gap-refinement job: which record, main hit (-1) or multi index, query orientation, window */
struct RefineJob { int rec, multi, strand, ext, len; int64_t pos; int64_t win_lo; int win_n; };

/* MD string and NM of an alignment (bwa_cal_md1, bwase.c:253-315).  Written straight into the caller's field (no heap string, no
 * printf per run of matches); where no ambiguity hole of the .amb file touches the alignment's stretch of the reference -- one search
 * per record instead of one per base -- the bases come from the packed text directly. */
static inline bool make_md(const nabwa_reference *R, int n_cigar, const uint16_t *cigar, int len, uint32_t pos0, const uint8_t *q,
					char *md, int cap, int *nm_out)
{
	int64_t pos = pos0; int u = 0, nm = 0, y = 0;
	int o = 0; bool fits = true;                         /* characters written; false once the field is full */
	auto put = [&](char c) { if (o + 1 < cap) md[o++] = c; else fits = false; };
	auto flush_num = [&]() {
		char t[12]; int n = 0; unsigned v = (unsigned)u;
		do { t[n++] = (char)('0' + v % 10u); v /= 10u; } while (v);
		while (n) put(t[--n]);
	};
	auto base_chr = [](int c) -> char { return c > 3 ? (char)c : "ACGT"[c]; };
	/* the stretch of the reference this alignment can touch: pos0 .. pos0 + (M + D lengths) */
	int64_t span = len;
	if (n_cigar) { span = 0; for (int k = 0; k < n_cigar; ++k) { const int op = COP(cigar[k]); if (op == 0 || op == 2) span += CLEN(cigar[k]); } }
	bool plain = true;                                   /* no hole overlaps [pos0, pos0 + span) */
	{
		size_t lo = 0, hi = R->holes.size();
		while (lo < hi) { const size_t mid = (lo + hi) / 2; if (R->holes[mid].offset + R->holes[mid].len <= (int64_t)pos0) lo = mid + 1; else hi = mid; }
		if (lo < R->holes.size() && R->holes[lo].offset < (int64_t)pos0 + span) plain = false;
	}
#define MD_REF(p_) (plain ? pac_at(R, (p_)) : ref_char(R, (p_)))
	if (n_cigar) {
		for (int k = 0; k < n_cigar; ++k) {
			const int l = CLEN(cigar[k]), op = COP(cigar[k]);
			if (op == 0) {
				for (int z = 0; z < l && pos < R->l_pac; ++z, ++y, ++pos) {
					if (plain) {                                     /* the matching stretch up to the next difference, eight bases at a time */
						const int64_t left = R->l_pac - pos;
						const int run = pac_match_run(R, pos, q + y, (int64_t)(l - z) < left ? l - z : (int)left);
						u += run; z += run; y += run; pos += run;
						if (z >= l || pos >= R->l_pac) break;
					}
					const int c = MD_REF(pos);
					if (c > 3 || q[y] > 3 || c != q[y]) { flush_num(); put(base_chr(c)); ++nm; u = 0; } else ++u;
				}
			} else if (op == 1 || op == 3) { y += l; if (op == 1) nm += l; }
			else {
				flush_num(); put('^');
				for (int z = 0; z < l && pos < R->l_pac; ++z, ++pos) put(base_chr(MD_REF(pos)));
				u = 0; nm += l;
			}
		}
	} else {
		for (int z = 0; z < len; ++z, ++pos) {
			if (plain && pos + (len - z) <= R->l_pac) {
				const int run = pac_match_run(R, pos, q + z, len - z);
				u += run; z += run; pos += run;
				if (z >= len) break;
			}
			const int c = MD_REF(pos);
			if (c > 3 || q[z] > 3 || c != q[z]) { flush_num(); put(base_chr(c)); ++nm; u = 0; } else ++u;
		}
	}
#undef MD_REF
	flush_num();
	md[o < cap ? o : cap - 1] = 0;
	*nm_out = nm;
	return fits;                                         /* false: the string did not fit (the caller reports NABWA_ECAP, never a cut-off tag) */
}


static inline double fin_now() { struct timeval tv; gettimeofday(&tv, 0); return tv.tv_sec + 1e-6 * tv.tv_usec; }
static inline nabwa_se_t *rec_at(void *base, size_t stride, int i) { return (nabwa_se_t*)((char*)base + (size_t)i * stride); }

/* bwa_aln2seq_core with set_main (bwase.c:28-46): reservoir choice among the best-score rows with the caller's
 * drand48 stream; fills n_mm/n_gapo/n_gape/strand/score/sa/c1/c2/type. */
static inline void choose_main(nabwa_se_t &s, int na, const nabwa_aln1_t *A, uint64_t *rng48)
{
	if (na == 0) { s.type = 0; s.c1 = s.c2 = 0; return; }
	int cnt = 0, j;
	const int best = A[0].score;
	for (j = 0; j < na; ++j) {
		if (A[j].score > best) break;
		const uint32_t w = A[j].l - A[j].k + 1;
		if (rng48_next(rng48) * (double)(w + cnt) > (double)cnt) {
			s.n_mm = A[j].info & 0xff; s.n_gapo = A[j].info >> 8 & 0xff; s.n_gape = A[j].info >> 16 & 0xff;
			s.strand = A[j].info >> 24 & 1; s.score = A[j].score;
			s.sa = A[j].k + (uint32_t)((double)w * rng48_next(rng48));
		}
		cnt += w;
	}
	s.c1 = cnt & 0xfffffff;
	for (; j < na; ++j) cnt += A[j].l - A[j].k + 1;
	s.c2 = (cnt - s.c1) & 0xfffffff;
	s.type = s.c1 > 1 ? 2 : 1;                                             /* BWA_TYPE_REPEAT : BWA_TYPE_UNIQUE */
}

/* bwa_aln2seq_core with n_multi (bwase.c:48-94): when the read has at most n_multi+1 hit rows in total, list them
 * all but the chosen one (rows, still as BWT rows in .pos).  The sampling branch of the reference is unreachable
 * under that condition ("In fact, we never come here"), so no random numbers are drawn. */
static inline void list_multi(nabwa_se_t &s, int na, const nabwa_aln1_t *A, int n_multi)
{
	s.n_multi = 0;
	if (na == 0 || n_multi <= 0) return;
	uint64_t tot = 0;
	for (int j = 0; j < na; ++j) tot += A[j].l - A[j].k + 1;
	if (tot > (uint64_t)n_multi + 1) return;
	int z = 0;
	for (int j = 0; j < na; ++j)
		for (uint32_t r = A[j].k; r <= A[j].l; ++r) {
			if (r == s.sa) continue;
			if (z == n_multi) break;
			s.multi[z].pos = r; s.multi[z].gap = (A[j].info >> 8 & 0xff) + (A[j].info >> 16 & 0xff);
			s.multi[z].mm = A[j].info & 0xff; s.multi[z].strand = A[j].info >> 24 & 1; s.multi[z].n_cigar = 0; ++z;
		}
	s.n_multi = z;
}

/* bwa_approx_mapQ (bwase.c:113-122) */
static inline int approx_mapq(const nabwa_se_t &s, int md)
{
	if (s.c1 == 0) return 23;
	if (s.c1 > 1) return 0;
	if (s.n_mm == md) return 25;
	if (s.c2 == 0) return 37;
	const int nn = s.c2 >= 255 ? 255 : (int)s.c2; const int g = (int)(4.343 * log((double)nn) + 0.5);
	return 23 < g ? 0 : 23 - g;
}

/* pos_end (bwase.c:425-436) */
static inline int64_t rec_pos_end(const nabwa_se_t &s)
{
	if (!s.n_cigar) return (int64_t)s.pos + s.len;
	int64_t x = s.pos;
	for (int k = 0; k < s.n_cigar; ++k) { const int op = COP(s.cigar[k]); if (op == 0 || op == 2) x += CLEN(s.cigar[k]); }
	return x;
}

/* host threads of the finishing chains: slices of independent records */
static inline int fin_threads(size_t n)
{
	int nt = (int)std::thread::hardware_concurrency(); if (nt < 1) nt = 1; if (nt > 16) nt = 16;
	if (getenv("NABWA_HOST_THREADS")) nt = std::max(1, atoi(getenv("NABWA_HOST_THREADS")));
	if (n < 4096) nt = 1;
	return nt;
}
template <class F> static inline void fin_parallel(int nt, size_t count, F f)       /* f(slice, lo, hi) */
{
	if (nt <= 1) { f(0, (size_t)0, count); return; }
	std::vector<std::thread> th;
	for (int t = 0; t < nt; ++t) th.emplace_back([=]() { f(t, count * t / nt, count * (t + 1) / nt); });
	for (auto &x : th) x.join();
}

/* Gap refinement of every gapped hit of the batch (main hits and multi hits) as ONE batch of banded global
 * alignments on the GPU (refine_gapped_core, bwase.c:189-237; driver bwase.c:366-381: mate-rescued and
 * unmapped records are skipped).  The host work around the kernel -- finding the jobs, cutting their reference windows out
 * of the packed text, laying the reads out in alignment orientation, turning the paths into CIGARs -- runs on slices of the
 * records / jobs in threads. */
static inline int refine_batch(nabwa_index_t *ix, void *base, size_t stride, int n, const int64_t *off, const uint8_t *seq,
							   const uint8_t *rseq, size_t *n_jobs)
{
	const nabwa_reference *R = ix->ref;
	const bool timing = getenv("NABWA_TIMING") != 0;
	const double tr0 = fin_now();
	const int nt = fin_threads((size_t)n);
	std::vector<std::vector<RefineJob>> part((size_t)nt);
	fin_parallel(nt, (size_t)n, [&](int t, size_t lo, size_t hi) {
		std::vector<RefineJob> &v = part[(size_t)t];
		for (size_t i = lo; i < hi; ++i) {
			const nabwa_se_t &s = *rec_at(base, stride, (int)i);
			for (int j = 0; j < s.n_multi; ++j)
				if (s.multi[j].gap) v.push_back({ (int)i, j, s.multi[j].strand, (s.multi[j].strand ? 1 : -1) * s.multi[j].gap, s.len, s.multi[j].pos, 0, 0 });
			if (s.type != 0 && s.type != 3 && s.n_gapo) v.push_back({ (int)i, -1, s.strand, (s.strand ? 1 : -1) * (s.n_gapo + s.n_gape), s.len, s.pos, 0, 0 });
		}
	});
	std::vector<RefineJob> jobs;
	{ size_t tot = 0; for (auto &v : part) tot += v.size(); jobs.reserve(tot); for (auto &v : part) jobs.insert(jobs.end(), v.begin(), v.end()); }
	if (n_jobs) *n_jobs = jobs.size();
	if (jobs.empty()) return NABWA_OK;
	static const int maq[25] = { 11,-19,-19,-19,-13, -19,11,-19,-19,-13, -19,-19,11,-19,-13, -19,-19,-19,11,-13, -13,-13,-13,-13,-13 };  /* aln_sm_maq */
	const size_t nj = jobs.size();
	const int ntj = fin_threads(nj);
	std::vector<int64_t> ro(nj + 1, 0), qo(nj + 1, 0);
	for (size_t t = 0; t < nj; ++t) {                      /* the windows' extents (bwase.c:197-209), then where each starts in the batch */
		RefineJob &J = jobs[t];
		const int ref_len = J.len + abs(J.ext);
		const int64_t p = (uint32_t)J.pos > R->l_pac ? (int64_t)(int32_t)(uint32_t)J.pos : (int64_t)(uint32_t)J.pos;   /* bwase.c:197 */
		J.pos = p;
		int64_t lo, hi;
		if (J.ext > 0) { lo = std::max<int64_t>(p, 0); hi = std::min<int64_t>(p + ref_len, R->l_pac); }
		else { const int64_t x = p + J.len; lo = x - ref_len > 0 ? x - ref_len : 0; hi = std::min<int64_t>(x, R->l_pac); }
		J.win_lo = lo; J.win_n = hi > lo ? (int)(hi - lo) : 0;
		ro[t + 1] = ro[t] + J.win_n; qo[t + 1] = qo[t] + J.len;
	}
	/* (no zero fill: every byte that is read is written first -- 100 MB of it for 181 k jobs with the CIGAR rows below) */
	/* (and no fresh memory either: the three big blocks of a call -- windows, reads, CIGAR rows, 100 MB for 181 k jobs -- stay with the calling thread
	 * for its next batch; mapping, faulting in and unmapping them anew every call cost as much as cutting the windows) */
	struct Scratch { std::unique_ptr<uint8_t[]> p; size_t cap = 0; uint8_t *get(size_t m) { if (m > cap) { p.reset(); p.reset(new uint8_t[m + m / 4]); cap = m + m / 4; } return p.get(); } };
	static thread_local Scratch scr_r, scr_q, scr_c;
	struct { uint8_t *p; uint8_t *data() const { return p; } } rbuf{ scr_r.get((size_t)ro[nj] + 1) }, qbuf{ scr_q.get((size_t)qo[nj] + 1) };
	rbuf.p[ro[nj]] = 0; qbuf.p[qo[nj]] = 0;
	fin_parallel(ntj, nj, [&](int, size_t lo_t, size_t hi_t) {
		for (size_t t = lo_t; t < hi_t; ++t) {
			const RefineJob &J = jobs[t];
			uint8_t *rb = rbuf.data() + ro[t], *qb = qbuf.data() + qo[t];
			if (t + 8 < hi_t) { const RefineJob &F = jobs[t + 8]; const uint8_t *const w = R->pac.data() + (F.win_lo >> 2); __builtin_prefetch(w); __builtin_prefetch(w + 48); __builtin_prefetch((F.strand ? rseq : seq) + off[F.rec]); }
			{	/* the window out of the packed reference: four bases per byte of it through a table */
				static const struct Pac4 { uint32_t v[256]; Pac4() { for (int x = 0; x < 256; ++x) v[x] = (uint32_t)(x >> 6 & 3) | (uint32_t)(x >> 4 & 3) << 8 | (uint32_t)(x >> 2 & 3) << 16 | (uint32_t)(x & 3) << 24; } } pac4;
				int k = 0; int64_t g = J.win_lo;
				for (; k < J.win_n && (g & 3); ++k, ++g) rb[k] = (uint8_t)pac_at(R, g);
				const uint8_t *pb = R->pac.data() + (g >> 2);
				for (; k + 4 <= J.win_n; k += 4, g += 4, ++pb) memcpy(rb + k, &pac4.v[*pb], 4);
				for (; k < J.win_n; ++k, ++g) rb[k] = (uint8_t)pac_at(R, g);
			}
			/* query in alignment orientation: reverse strand = rseq, forward = the read itself (seq is stored reversed) */
			const uint8_t *src = (J.strand ? rseq : seq) + off[J.rec];
			if (J.strand) memcpy(qb, src, (size_t)J.len);
			else for (int k = 0; k < J.len; ++k) qb[k] = src[J.len - 1 - k];
		}
	});
	const double tr1 = fin_now();
	const int MAXC = NABWA_MAX_CIGAR;
	std::vector<int32_t> sc(nj), nc(nj);
	struct { uint32_t *p; uint32_t *get() const { return p; } uint32_t &operator[](size_t i) const { return p[i]; } } c32{ (uint32_t*)scr_c.get(nj * (size_t)MAXC * 4) };      /* row t: its first nc[t] words are valid */
	int r = nabwa_global_align(ix->device, (int)nj, ro.data(), rbuf.data(), qo.data(), qbuf.data(), 26, 9, 5, maq, 50,
							   sc.data(), nc.data(), c32.get(), MAXC);                       /* aln_param_bwa, stdaln.c:227 */
	if (r != NABWA_OK) return r;
	const double tr2 = fin_now();
	std::vector<int> bad((size_t)ntj, 0);
	fin_parallel(ntj, nj, [&](int slice, size_t lo_t, size_t hi_t) {
		for (size_t t = lo_t; t < hi_t; ++t) {
			const RefineJob &J = jobs[t];
			if (nc[t] > MAXC || nc[t] < 1) { bad[(size_t)slice] = 1; continue; }
			uint16_t cg[NABWA_MAX_CIGAR]; int m = nc[t]; int64_t p = J.pos;
			for (int k = 0; k < m; ++k) cg[k] = CMAKE(c32[t * MAXC + k] & 0xf, c32[t * MAXC + k] >> 4);
			if (J.ext < 0) {                       /* forward strand: the end was anchored, shift the start by the net indel */
				int d = 0;
				for (int k = 0; k < m; ++k) { if (COP(cg[k]) == 2) d -= CLEN(cg[k]); else if (COP(cg[k]) == 1) d += CLEN(cg[k]); }
				p += d;
			}
			if (COP(cg[0]) == 2) { p += CLEN(cg[0]); for (int k = 0; k + 1 < m; ++k) cg[k] = cg[k + 1]; --m; }
			if (COP(cg[m - 1]) == 2) --m;
			if (COP(cg[m - 1]) == 1) cg[m - 1] = CMAKE(3, CLEN(cg[m - 1]));
			if (COP(cg[0]) == 1) cg[0] = CMAKE(3, CLEN(cg[0]));
			nabwa_se_t &s = *rec_at(base, stride, J.rec);         /* (a record's main hit and its multi hits are different fields: no two jobs write the same bytes) */
			if (J.multi < 0) { s.pos = (uint32_t)p; s.n_cigar = m; memcpy(s.cigar, cg, 2 * m); }
			else { s.multi[J.multi].pos = (uint32_t)p; s.multi[J.multi].n_cigar = m; memcpy(s.multi[J.multi].cigar, cg, 2 * m); }
		}
	});
	for (int b : bad) if (b) return nabwa_fail(NABWA_ECAP, "refined CIGAR longer than NABWA_MAX_CIGAR");
	if (timing) fprintf(stderr, "[nabwa] refine_batch %zu jobs: jobs + windows %.3f s, nabwa_global_align %.3f s, CIGARs %.3f s\n", nj, tr1 - tr0, tr2 - tr1, fin_now() - tr2);
	return NABWA_OK;
}

/* MD / NM of a mapped record, then the quality-trimmed tail as a soft clip (bwase.c:399-419, :320-354).
 * `fwd` is scratch for the un-reversed read. */
static inline bool md_and_trim(const nabwa_reference *R, nabwa_se_t &s, const uint8_t *seq_i, const uint8_t *rseq_i, std::vector<uint8_t> &fwd)
{
	const int len = s.len;
	const uint8_t *q;
	if (s.strand) q = rseq_i;
	else { fwd.resize(len); for (int k = 0; k < len; ++k) fwd[k] = seq_i[len - 1 - k]; q = fwd.data(); }
	const bool md_fits = make_md(R, s.n_cigar, s.cigar, len, s.pos, q, s.md, NABWA_MAX_MD, &s.nm);
	if (len != s.full_len) {                                   /* bwa_correct_trimmed */
		const int clip = s.full_len - len;
		if (s.strand == 0) {
			if (s.n_cigar && COP(s.cigar[s.n_cigar - 1]) == 3) s.cigar[s.n_cigar - 1] += clip;
			else { if (s.n_cigar == 0) { s.n_cigar = 2; s.cigar[0] = CMAKE(0, len); } else ++s.n_cigar; s.cigar[s.n_cigar - 1] = CMAKE(3, clip); }
		} else {
			if (s.n_cigar && COP(s.cigar[0]) == 3) s.cigar[0] += clip;
			else {
				if (s.n_cigar == 0) { s.n_cigar = 2; s.cigar[1] = CMAKE(0, len); }
				else { ++s.n_cigar; memmove(s.cigar + 1, s.cigar, (s.n_cigar - 1) * 2); }
				s.cigar[0] = CMAKE(3, clip);
			}
		}
		s.len = s.full_len;
	}
	return md_fits;
}
