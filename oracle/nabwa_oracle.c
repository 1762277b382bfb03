/* oracle/nabwa_oracle.c -- TEST INFRASTRUCTURE ONLY (see nabwa_oracle.h).
 *
 * Plain-C restatement of the reference's per-read alignment path.  Each function
 * names the reference lines whose behaviour it restates.  Written from the
 * algorithm's definition, not transcribed: rank queries are evaluated "by
 * definition" over the 48-byte buckets, the priority stack is a vector per score,
 * the banded global DP is expressed through per-row [left,right] ranges.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <pthread.h>
#include "nabwa_oracle.h"

#define OCC_INTV 128u
#define NEG1 0xffffffffu

/* ------------------------------------------------------------------ index load */

static void *xread_all(const char *fn, size_t *size)
{
	FILE *f = fopen(fn, "rb");
	void *buf; long n;
	if (!f) { fprintf(stderr, "[oracle] cannot open %s\n", fn); abort(); }
	fseek(f, 0, SEEK_END); n = ftell(f); fseek(f, 0, SEEK_SET);
	buf = malloc(n ? n : 1);
	if (fread(buf, 1, n, f) != (size_t)n) { fprintf(stderr, "[oracle] short read %s\n", fn); abort(); }
	fclose(f);
	*size = n;
	return buf;
}

/* .bwt layout: primary, L2[1..4], then Occ-interleaved words (bwtio.c:184-204) */
static void load_bwt(const char *fn, orc_bwt_t *b)
{
	size_t sz; uint32_t *raw = (uint32_t*)xread_all(fn, &sz);
	memset(b, 0, sizeof(*b));
	b->primary = raw[0];
	memcpy(b->L2 + 1, raw + 1, 16);
	b->seq_len = b->L2[4];
	b->n_words = sz / 4 - 5;
	b->bwt = (uint32_t*)malloc(b->n_words * 4);
	memcpy(b->bwt, raw + 5, b->n_words * 4);
	free(raw);
}

/* .sa layout: primary, 4 skipped words, sa_intv, seq_len, then n_sa-1 samples (bwtio.c:161-182) */
static void load_sa(const char *fn, orc_bwt_t *b)
{
	size_t sz; uint32_t *raw = (uint32_t*)xread_all(fn, &sz);
	if (raw[0] != b->primary || raw[6] != b->seq_len) { fprintf(stderr, "[oracle] SA/BWT mismatch\n"); abort(); }
	b->sa_intv = raw[5];
	b->n_sa = (b->seq_len + b->sa_intv) / b->sa_intv;
	b->sa = (uint32_t*)malloc((size_t)b->n_sa * 4);
	b->sa[0] = NEG1;
	memcpy(b->sa + 1, raw + 7, (size_t)(b->n_sa - 1) * 4);
	free(raw);
}

orc_index_t *orc_index_load(const char *prefix, int with_sa, int with_pac)
{
	orc_index_t *ix = (orc_index_t*)calloc(1, sizeof(*ix));
	char fn[4096]; FILE *f; int i; long long xx; size_t sz;
	snprintf(fn, sizeof fn, "%s.bwt", prefix);  load_bwt(fn, &ix->bwt[0]);
	snprintf(fn, sizeof fn, "%s.rbwt", prefix); load_bwt(fn, &ix->bwt[1]);
	if (with_sa) {
		snprintf(fn, sizeof fn, "%s.sa", prefix);  load_sa(fn, &ix->bwt[0]);
		snprintf(fn, sizeof fn, "%s.rsa", prefix); load_sa(fn, &ix->bwt[1]);
	}
	/* .ann / .amb text formats (bntseq.c:88-133) */
	snprintf(fn, sizeof fn, "%s.ann", prefix);
	if ((f = fopen(fn, "r")) != 0) {
		unsigned seed;
		if (fscanf(f, "%lld%d%u", &xx, &ix->n_seqs, &seed) != 3) abort();
		ix->l_pac = xx; ix->seed = seed;
		ix->anns = (orc_ann_t*)calloc(ix->n_seqs, sizeof(orc_ann_t));
		for (i = 0; i < ix->n_seqs; ++i) {
			unsigned gi; char name[1024]; int c;
			if (fscanf(f, "%u%1023s", &gi, name) != 2) abort();
			memcpy(ix->anns[i].name, name, 63);
			while ((c = fgetc(f)) != '\n' && c != EOF) {}
			if (fscanf(f, "%lld%d%d", &xx, &ix->anns[i].len, &ix->anns[i].n_ambs) != 3) abort();
			ix->anns[i].offset = xx;
		}
		fclose(f);
		snprintf(fn, sizeof fn, "%s.amb", prefix);
		if ((f = fopen(fn, "r")) != 0) {
			int ns;
			if (fscanf(f, "%lld%d%d", &xx, &ns, &ix->n_holes) != 3) abort();
			ix->holes = (orc_hole_t*)calloc(ix->n_holes ? ix->n_holes : 1, sizeof(orc_hole_t));
			for (i = 0; i < ix->n_holes; ++i) {
				char s[64];
				if (fscanf(f, "%lld%d%63s", &xx, &ix->holes[i].len, s) != 3) abort();
				ix->holes[i].offset = xx; ix->holes[i].amb = s[0];
			}
			fclose(f);
		}
	}
	if (with_pac) {
		snprintf(fn, sizeof fn, "%s.pac", prefix);
		ix->pac = (uint8_t*)xread_all(fn, &sz);
	}
	return ix;
}

orc_index_t *orc_index_wrap(const uint32_t *bwt0, uint64_t nw0, const uint32_t *bwt1, uint64_t nw1)
{
	orc_index_t *ix = (orc_index_t*)calloc(1, sizeof(*ix));
	const uint32_t *raw[2] = { bwt0, bwt1 }; uint64_t nw[2] = { nw0, nw1 }; int t;
	for (t = 0; t < 2; ++t) {   /* same header as a .bwt file: primary, L2[1..4], words */
		orc_bwt_t *b = &ix->bwt[t];
		b->primary = raw[t][0];
		memcpy(b->L2 + 1, raw[t] + 1, 16);
		b->seq_len = b->L2[4];
		b->n_words = nw[t] - 5;
		b->bwt = (uint32_t*)(raw[t] + 5);
	}
	ix->n_seqs = -1; /* marks wrapped (not owned) arrays */
	return ix;
}

void orc_index_free(orc_index_t *ix)
{
	if (!ix) return;
	if (ix->n_seqs != -1) { free(ix->bwt[0].bwt); free(ix->bwt[1].bwt); }
	free(ix->bwt[0].sa); free(ix->bwt[1].sa);
	free(ix->anns); free(ix->holes); free(ix->pac); free(ix);
}

/* ------------------------------------------------------------------ rank primitives */

static __thread orc_counters_t *tl_ctr;   /* optional per-thread touch counters */
#define TOUCH_BUCKET() do { if (tl_ctr) ++tl_ctr->n_bucket; } while (0)

/* base j of the $-removed BWT string (bwt.h:61-66) */
static inline int b0(const orc_bwt_t *b, uint32_t j)
{
	const uint32_t *p = b->bwt + (uint64_t)(j / OCC_INTV) * 12 + 4;
	return p[(j % OCC_INTV) >> 4] >> ((~j & 15) << 1) & 3;
}

/* counts of all four bases in B0[start_of_bucket(j) .. j], plus the bucket's checkpoint.
 * Definition of Occ restated from bwt.c:92-115 / 159-176: the checkpoint holds the counts
 * before the bucket; bases sit 16 per word, first base in the top bits. */
static void occ4_at(const orc_bwt_t *b, uint32_t j, uint32_t cnt[4])
{
	const uint32_t *p = b->bwt + (uint64_t)(j / OCC_INTV) * 12;
	uint32_t r = j % OCC_INTV, w, n1 = 0, n2 = 0, n3 = 0;
	for (w = 0; w <= r >> 4; ++w) {
		/* keep the first nb bases of the word (they sit in the top bits) and count by bit planes */
		uint32_t nb = (w < r >> 4) ? 16 : (r & 15) + 1;
		uint32_t keep = nb == 16 ? 0xffffffffu : ~((1u << (32 - 2 * nb)) - 1);
		uint32_t x = p[4 + w], lo = x & keep & 0x55555555u, hi = (x >> 1) & (keep >> 1) & 0x55555555u;
		n3 += __builtin_popcount(hi & lo);
		n2 += __builtin_popcount(hi & ~lo);
		n1 += __builtin_popcount(lo & ~hi);
	}
	cnt[0] = p[0] + (r + 1 - n1 - n2 - n3); cnt[1] = p[1] + n1; cnt[2] = p[2] + n2; cnt[3] = p[3] + n3;
}

/* bwt_occ (bwt.c:92-115): k = -1 -> 0; rows >= primary shift by one because '$' is not stored */
uint32_t orc_occ(const orc_bwt_t *b, uint32_t k, int c)
{
	uint32_t cnt[4];
	if (k == b->seq_len) return b->L2[c+1] - b->L2[c];
	if (k == NEG1) return 0;
	if (k >= b->primary) --k;
	TOUCH_BUCKET();
	occ4_at(b, k, cnt);
	return cnt[c];
}

/* bwt_occ4 (bwt.c:159-176) */
void orc_occ4(const orc_bwt_t *b, uint32_t k, uint32_t cnt[4])
{
	if (k == NEG1) { cnt[0] = cnt[1] = cnt[2] = cnt[3] = 0; return; }
	if (k >= b->primary) --k;
	TOUCH_BUCKET();
	occ4_at(b, k, cnt);
}

static inline int same_bucket(const orc_bwt_t *b, uint32_t k, uint32_t l)
{
	uint32_t _k = k >= b->primary ? k - 1 : k, _l = l >= b->primary ? l - 1 : l;
	return !(_l / OCC_INTV != _k / OCC_INTV || k == NEG1 || l == NEG1);
}

/* bwt_2occ (bwt.c:118-153): values equal two bwt_occ calls; the touch count follows the
 * reference's fast path (one bucket when both rows share it) */
static void orc_2occ(const orc_bwt_t *b, uint32_t k, uint32_t l, int c, uint32_t *ok, uint32_t *ol)
{
	if (k == l) { *ok = *ol = orc_occ(b, k, c); return; }
	if (same_bucket(b, k, l)) {
		orc_counters_t *save = tl_ctr;
		TOUCH_BUCKET();
		tl_ctr = 0; *ok = orc_occ(b, k, c); *ol = orc_occ(b, l, c); tl_ctr = save;
	} else { *ok = orc_occ(b, k, c); *ol = orc_occ(b, l, c); }
}

/* bwt_2occ4 (bwt.c:179-216) */
void orc_2occ4(const orc_bwt_t *b, uint32_t k, uint32_t l, uint32_t ck[4], uint32_t cl[4])
{
	if (k == l) { orc_occ4(b, k, ck); memcpy(cl, ck, 16); return; }
	if (same_bucket(b, k, l)) {
		orc_counters_t *save = tl_ctr;
		TOUCH_BUCKET();
		tl_ctr = 0; orc_occ4(b, k, ck); orc_occ4(b, l, cl); tl_ctr = save;
	} else { orc_occ4(b, k, ck); orc_occ4(b, l, cl); }
}

/* bwt_sa / bwt_invPsi (bwt.c:72-81, bwt.h:71-75) */
uint32_t orc_sa(const orc_bwt_t *b, uint32_t k)
{
	uint32_t steps = 0;
	if (tl_ctr) ++tl_ctr->n_sa;
	while (k % b->sa_intv != 0) {
		++steps;
		if (k == b->primary) k = 0;
		else {
			int c = b0(b, k < b->primary ? k : k - 1);
			k = b->L2[c] + orc_occ(b, k, c);
		}
	}
	k /= b->sa_intv;
	return steps + (k ? b->sa[k] : NEG1);
}

/* ------------------------------------------------------------------ options, max_diff */

void orc_default_opt(orc_opt_t *o)   /* gap_init_opt, bwtaln.c:19-35 */
{
	memset(o, 0, sizeof(*o));
	o->s_mm = 3; o->s_gapo = 11; o->s_gape = 4;
	o->max_diff = -1; o->max_gapo = 1; o->max_gape = 6;
	o->indel_end_skip = 5; o->max_del_occ = 10; o->max_entries = 2000000;
	o->mode = ORC_MODE_GAPE | ORC_MODE_COMPREAD;
	o->seed_len = 32; o->max_seed_diff = 2;
	o->fnr = 0.04f; o->n_threads = 1; o->max_top2 = 30; o->trim_qual = 0;
}

/* bwa_cal_maxdiff (bwtaln.c:37-49).  The factorial is an int that wraps past 12!;
 * kept as 32-bit wrap-around to match what the compiled reference does. */
int orc_maxdiff(int l, double err, double thres)
{
	double elambda = exp(-l * err), sum = elambda, y = 1.0;
	uint32_t x = 1; int k;
	for (k = 1; k < 1000; ++k) {
		y *= l * err;
		x *= (uint32_t)k;
		sum += elambda * y / (int32_t)x;
		if (1.0 - sum < thres) return k;
	}
	return 2;
}

/* ------------------------------------------------------------------ width bounds */

typedef struct { uint32_t w; int bid; } width_t;

/* bwt_cal_width (bwtaln.c:52-76): exact backward search on the opposite-orientation index,
 * restarting (and bumping the bound) whenever the interval empties or an N is met. */
static void cal_width(const orc_bwt_t *rb, int len, const uint8_t *str, width_t *width)
{
	uint32_t k = 0, l = rb->seq_len, ok, ol; int i, bid = 0;
	for (i = 0; i < len; ++i) {
		int c = str[i];
		if (c < 4) {
			orc_2occ(rb, k - 1, l, c, &ok, &ol);
			k = rb->L2[c] + ok + 1;
			l = rb->L2[c] + ol;
		}
		if (k > l || c > 3) { k = 0; l = rb->seq_len; ++bid; }
		width[i].w = l - k + 1;
		width[i].bid = bid;
	}
	width[len].w = 0;
	width[len].bid = ++bid;
}

/* the same for the tests of kernel W: widths and bounds of one pass as two flat arrays of len + 1 entries */
void orc_cal_width(const orc_bwt_t *rb, int len, const uint8_t *str, uint32_t *w_out, int32_t *bid_out)
{
	width_t *w = (width_t*)calloc(len + 1, sizeof(width_t)); int i;
	cal_width(rb, len, str, w);
	for (i = 0; i <= len; ++i) { w_out[i] = w[i].w; bid_out[i] = w[i].bid; }
	free(w);
}

/* ------------------------------------------------------------------ priority stack */

enum { ST_M = 0, ST_I = 1, ST_D = 2 };

typedef struct {
	int score, a, i;
	int n_mm, n_gapo, n_gape, state;
	uint32_t k, l;
	int last_diff_pos;
} ent_t;

typedef struct { int n, m; ent_t *e; } lifo_t;
typedef struct { int n_stacks, best, n_entries; lifo_t *s; } pstack_t;

static pstack_t *pstack_new(int n_stacks)     /* gap_init_stack, bwtgap.c:13-27 */
{
	pstack_t *p = (pstack_t*)calloc(1, sizeof(*p));
	p->n_stacks = n_stacks;
	p->s = (lifo_t*)calloc(n_stacks, sizeof(lifo_t));
	return p;
}
static void pstack_free(pstack_t *p)
{
	int i;
	for (i = 0; i < p->n_stacks; ++i) free(p->s[i].e);
	free(p->s); free(p);
}
static void pstack_reset(pstack_t *p)          /* bwtgap.c:37-44 */
{
	int i;
	for (i = 0; i < p->n_stacks; ++i) p->s[i].n = 0;
	p->best = p->n_stacks; p->n_entries = 0;
}
static void pstack_push(pstack_t *p, const ent_t *e, orc_counters_t *ctr)   /* bwtgap.c:46-65 */
{
	lifo_t *q = p->s + e->score;
	if (q->n == q->m) { q->m = q->m ? q->m * 2 : 4; q->e = (ent_t*)realloc(q->e, q->m * sizeof(ent_t)); }
	q->e[q->n++] = *e;
	++p->n_entries;
	if (p->best > e->score) p->best = e->score;
	if (ctr) ++ctr->n_push;
}
static void pstack_pop(pstack_t *p, ent_t *e)   /* bwtgap.c:67-79: newest entry of the lowest score */
{
	lifo_t *q = p->s + p->best;
	*e = q->e[--q->n];
	--p->n_entries;
	if (p->n_entries == 0) p->best = p->n_stacks;
	else if (q->n == 0) {
		int i = p->best + 1;
		while (i < p->n_stacks && p->s[i].n == 0) ++i;
		p->best = i;
	}
}

static int ilog2(uint32_t v) { int c = 0; while (v >>= 1) ++c; return c; }   /* bwtgap.c:93-102 */

#define SCORE(o, m, g, e) ((m) * (o)->s_mm + (g) * (o)->s_gapo + (e) * (o)->s_gape)

typedef struct { orc_aln_t *a; int n, m; } hits_t;

static void child(pstack_t *st, const orc_opt_t *o, int a, int i, uint32_t k, uint32_t l,
				  int n_mm, int n_gapo, int n_gape, int state, int is_diff, orc_counters_t *ctr)
{
	ent_t e;
	e.score = SCORE(o, n_mm, n_gapo, n_gape); e.a = a; e.i = i; e.k = k; e.l = l;
	e.n_mm = n_mm & 0xff; e.n_gapo = n_gapo & 0xff; e.n_gape = n_gape & 0xff; e.state = state;
	e.last_diff_pos = is_diff ? i : 0;
	pstack_push(st, &e, ctr);
}

/* bwt_match_gap (bwtgap.c:104-266).  o->max_diff / o->seed_len are the per-read values
 * prepared by the driver.  Returns hits in discovery order. */
static void match_gap(const orc_bwt_t *const bwts[2], int len, const uint8_t *const seq[2], width_t *const w[2],
					  width_t *const seed_w[2], const orc_opt_t *o, hits_t *H, pstack_t *st, int *pmax_entries,
					  orc_counters_t *ctr)
{
	int best_score = SCORE(o, o->max_diff + 1, o->max_gapo + 1, o->max_gape + 1);
	int best_diff = o->max_diff + 1, max_diff = o->max_diff, best_cnt = 0, max_entries = 0, j, n_N = 0;
	const int gape_mode = o->mode & ORC_MODE_GAPE, nonstop = o->mode & ORC_MODE_NONSTOP;
	(void)best_diff;
	H->n = 0;
	for (j = 0; j < len; ++j) if (seq[0][j] > 3) ++n_N;        /* too many N: no search (:118-123) */
	if (n_N > max_diff) return;
	pstack_reset(st);
	child(st, o, 0, len, 0, bwts[0]->seq_len, 0, 0, 0, ST_M, 0, ctr);   /* both roots use bwts[0]->seq_len (:127-128) */
	child(st, o, 1, len, 0, bwts[0]->seq_len, 0, 0, 0, ST_M, 0, ctr);

	while (st->n_entries) {
		ent_t e; int a, i, m, m_seed = 0, hit = 0, allow_diff = 1, allow_M = 1, tmp;
		uint32_t k, l, ck[4], cl[4], occ;
		const orc_bwt_t *bwt; const uint8_t *str; width_t *width; const width_t *sw = 0;

		if (max_entries < st->n_entries) max_entries = st->n_entries;
		if (st->n_entries > o->max_entries) break;
		pstack_pop(st, &e);
		if (ctr) ++ctr->n_pop;
		k = e.k; l = e.l; a = e.a; i = e.i;
		if (!nonstop && e.score > best_score + o->s_mm) break;

		m = max_diff - (e.n_mm + e.n_gapo);
		if (gape_mode) m -= e.n_gape;
		if (m < 0) continue;
		bwt = bwts[1 - a]; str = seq[a]; width = w[a];
		if (seed_w) {
			sw = seed_w[a];
			m_seed = o->max_seed_diff - (e.n_mm + e.n_gapo);
			if (gape_mode) m_seed -= e.n_gape;
		}
		if (i > 0 && m < width[i-1].bid) continue;

		if (i == 0) hit = 1;
		else if (m == 0 && (e.state == ST_M || gape_mode || e.n_gape == o->max_gape)) {
			/* nothing more may differ: finish with an exact backward search (bwt.c:237-252) */
			int t; uint32_t ok, ol;
			for (t = i - 1; t >= 0; --t) {
				int c = str[t];
				if (c > 3) break;
				orc_2occ(bwt, k - 1, l, c, &ok, &ol);
				k = bwt->L2[c] + ok + 1; l = bwt->L2[c] + ol;
				if (k > l) break;
			}
			if (t >= 0) continue;
			hit = 1;
		}

		if (hit) {
			int score = SCORE(o, e.n_mm, e.n_gapo, e.n_gape), add = 1;
			if (H->n == 0) {
				best_score = score;
				best_diff = e.n_mm + e.n_gapo + (gape_mode ? e.n_gape : 0);
				if (!nonstop) max_diff = best_diff + 1 > o->max_diff ? o->max_diff : best_diff + 1;
			}
			if (score == best_score) best_cnt += l - k + 1;
			else if (best_cnt > o->max_top2) break;
			if (e.n_gapo) for (j = 0; j < H->n; ++j) if (H->a[j].k == k && H->a[j].l == l) { add = 0; break; }
			if (add) {
				/* gap_shadow (bwtgap.c:81-91): tighten the bounds left of the last difference */
				uint32_t x = l - k + 1, max = bwt->seq_len; int t, jj = 0;
				for (t = 0; t < e.last_diff_pos; ++t) {
					if (width[t].w > x) width[t].w -= x;
					else if (width[t].w == x) { width[t].bid = 1; width[t].w = max - (++jj); }
				}
				if (H->n == H->m) { H->m = H->m ? H->m * 2 : 4; H->a = (orc_aln_t*)realloc(H->a, H->m * sizeof(orc_aln_t)); }
				H->a[H->n].info = (uint32_t)e.n_mm | (uint32_t)e.n_gapo << 8 | (uint32_t)e.n_gape << 16 | (uint32_t)a << 24;
				H->a[H->n].k = k; H->a[H->n].l = l; H->a[H->n].score = score;
				++H->n;
			}
			continue;
		}

		--i;
		orc_2occ4(bwt, k - 1, l, ck, cl);
		occ = l - k + 1;
		if (i > 0) {
			int ii = i - (len - o->seed_len);
			if (width[i-1].bid > m - 1) allow_diff = 0;
			else if (width[i-1].bid == m - 1 && width[i].bid == m - 1 && width[i-1].w == width[i].w) allow_M = 0;
			if (seed_w && ii > 0) {
				if (sw[ii-1].bid > m_seed - 1) allow_diff = 0;
				else if (sw[ii-1].bid == m_seed - 1 && sw[ii].bid == m_seed - 1 && sw[ii-1].w == sw[ii].w) allow_M = 0;
			}
		}
		tmp = (o->mode & ORC_MODE_LOGGAP) ? ilog2(e.n_gape + e.n_gapo) / 2 + 1 : e.n_gapo + e.n_gape;
		if (allow_diff && i >= o->indel_end_skip + tmp && len - i >= o->indel_end_skip + tmp) {
			if (e.state == ST_M) {
				if (e.n_gapo < o->max_gapo) {
					child(st, o, a, i, k, l, e.n_mm, e.n_gapo + 1, e.n_gape, ST_I, 1, ctr);
					for (j = 0; j < 4; ++j) {
						uint32_t nk = bwt->L2[j] + ck[j] + 1, nl = bwt->L2[j] + cl[j];
						if (nk <= nl) child(st, o, a, i + 1, nk, nl, e.n_mm, e.n_gapo + 1, e.n_gape, ST_D, 1, ctr);
					}
				}
			} else if (e.state == ST_I) {
				if (e.n_gape < o->max_gape) child(st, o, a, i, k, l, e.n_mm, e.n_gapo, e.n_gape + 1, ST_I, 1, ctr);
			} else if (e.n_gape < o->max_gape) {
				if (e.n_gape + e.n_gapo < max_diff || occ < (uint32_t)o->max_del_occ)
					for (j = 0; j < 4; ++j) {
						uint32_t nk = bwt->L2[j] + ck[j] + 1, nl = bwt->L2[j] + cl[j];
						if (nk <= nl) child(st, o, a, i + 1, nk, nl, e.n_mm, e.n_gapo, e.n_gape + 1, ST_D, 1, ctr);
					}
			}
		}
		if (allow_diff && allow_M) {
			for (j = 1; j <= 4; ++j) {
				int c = (str[i] + j) & 3, is_mm = (j != 4 || str[i] > 3);
				uint32_t nk = bwt->L2[c] + ck[c] + 1, nl = bwt->L2[c] + cl[c];
				if (nk <= nl) child(st, o, a, i, nk, nl, e.n_mm + is_mm, e.n_gapo, e.n_gape, ST_M, is_mm, ctr);
			}
		} else if (str[i] < 4) {
			int c = str[i] & 3;
			uint32_t nk = bwt->L2[c] + ck[c] + 1, nl = bwt->L2[c] + cl[c];
			if (nk <= nl) child(st, o, a, i, nk, nl, e.n_mm, e.n_gapo, e.n_gape, ST_M, 0, ctr);
		}
	}
	*pmax_entries = max_entries;
}

/* ------------------------------------------------------------------ batch driver */

typedef struct {
	const orc_index_t *ix; const orc_opt_t *opt; int lo, hi; const int64_t *off;
	const uint8_t *seq, *rseq; int per_read, batch_max_len;
	int32_t *n_aln, *max_entries; orc_aln_t **rows; orc_counters_t ctr; int want_ctr;
} job_t;

/* bwa_cal_sa_reg_gap (bwtaln.c:93-142) for reads [lo,hi) */
static void *job_run(void *arg)
{
	job_t *J = (job_t*)arg;
	const orc_opt_t *opt = J->opt;
	const orc_bwt_t *bw[2] = { &J->ix->bwt[0], &J->ix->bwt[1] };
	orc_counters_t *ctr = J->want_ctr ? &J->ctr : 0;
	width_t *w[2] = { 0, 0 }, *sw[2]; int wcap = 0, i, last_ns = -1;
	pstack_t *st = 0; hits_t H = { 0, 0, 0 };
	tl_ctr = ctr;
	sw[0] = (width_t*)calloc(opt->seed_len + 1, sizeof(width_t));
	sw[1] = (width_t*)calloc(opt->seed_len + 1, sizeof(width_t));
	for (i = J->lo; i < J->hi; ++i) {
		int len = (int)(J->off[i+1] - J->off[i]), sizing_len = J->per_read ? len : J->batch_max_len, ns;
		const uint8_t *sq[2] = { J->seq + J->off[i], J->rseq + J->off[i] };
		orc_opt_t lo = *opt;
		/* stack sizing / max_gapo clamp use the longest read of the CALL (bwtaln.c:102-106) */
		if (opt->fnr > 0.0) lo.max_diff = orc_maxdiff(sizing_len, 0.02, opt->fnr);
		if (lo.max_diff < lo.max_gapo) lo.max_gapo = lo.max_diff;
		ns = SCORE(&lo, lo.max_diff + 1, lo.max_gapo + 1, lo.max_gape + 1);
		if (ns != last_ns) { if (st) pstack_free(st); st = pstack_new(ns); last_ns = ns; }
		J->n_aln[i] = 0; J->max_entries[i] = 0; J->rows[i] = 0;
		if (len <= 0) continue;
		if (len + 1 > wcap) {
			wcap = len + 1;
			w[0] = (width_t*)realloc(w[0], wcap * sizeof(width_t));
			w[1] = (width_t*)realloc(w[1], wcap * sizeof(width_t));
		}
		cal_width(bw[0], len, sq[0], w[0]);
		cal_width(bw[1], len, sq[1], w[1]);
		if (opt->fnr > 0.0) lo.max_diff = orc_maxdiff(len, 0.02, opt->fnr);
		lo.seed_len = opt->seed_len < len ? opt->seed_len : 0x7fffffff;
		if (len > opt->seed_len) {
			cal_width(bw[0], opt->seed_len, sq[0] + (len - opt->seed_len), sw[0]);
			cal_width(bw[1], opt->seed_len, sq[1] + (len - opt->seed_len), sw[1]);
		}
		match_gap(bw, len, sq, w, len <= opt->seed_len ? 0 : sw, &lo, &H, st, &J->max_entries[i], ctr);
		J->n_aln[i] = H.n;
		if (H.n) { J->rows[i] = (orc_aln_t*)malloc(H.n * sizeof(orc_aln_t)); memcpy(J->rows[i], H.a, H.n * sizeof(orc_aln_t)); }
	}
	if (st) pstack_free(st);
	free(H.a); free(w[0]); free(w[1]); free(sw[0]); free(sw[1]);
	tl_ctr = 0;
	return 0;
}

long orc_cal_sa_reg_gap(const orc_index_t *ix, const orc_opt_t *opt, int n, const int64_t *off,
						const uint8_t *seq, const uint8_t *rseq, int per_read,
						int32_t *n_aln, orc_aln_t *aln_out, long aln_cap, int32_t *max_entries,
						int n_threads, orc_counters_t *ctr)
{
	orc_aln_t **rows = (orc_aln_t**)calloc(n ? n : 1, sizeof(*rows));
	job_t *jobs; pthread_t *tid; int t, i, max_len = 0; long tot = 0;
	if (n_threads < 1) n_threads = 1;
	if (n_threads > n) n_threads = n ? n : 1;
	for (i = 0; i < n; ++i) if (off[i+1] - off[i] > max_len) max_len = (int)(off[i+1] - off[i]);
	jobs = (job_t*)calloc(n_threads, sizeof(job_t));
	tid = (pthread_t*)calloc(n_threads, sizeof(pthread_t));
	for (t = 0; t < n_threads; ++t) {
		job_t *J = jobs + t;
		J->ix = ix; J->opt = opt; J->off = off; J->seq = seq; J->rseq = rseq; J->per_read = per_read;
		J->batch_max_len = max_len; J->n_aln = n_aln; J->max_entries = max_entries; J->rows = rows;
		J->lo = (int)((long)n * t / n_threads); J->hi = (int)((long)n * (t + 1) / n_threads);
		J->want_ctr = ctr != 0;
		if (n_threads == 1) job_run(J); else pthread_create(&tid[t], 0, job_run, J);
	}
	if (n_threads > 1) for (t = 0; t < n_threads; ++t) pthread_join(tid[t], 0);
	if (ctr) {
		memset(ctr, 0, sizeof(*ctr));
		for (t = 0; t < n_threads; ++t) {
			ctr->n_bucket += jobs[t].ctr.n_bucket; ctr->n_sa += jobs[t].ctr.n_sa;
			ctr->n_pop += jobs[t].ctr.n_pop; ctr->n_push += jobs[t].ctr.n_push;
		}
	}
	for (i = 0; i < n; ++i) {
		if (tot >= 0 && tot + n_aln[i] <= aln_cap) {
			if (n_aln[i]) memcpy(aln_out + tot, rows[i], n_aln[i] * sizeof(orc_aln_t));
			tot += n_aln[i];
		} else tot = -1;
		free(rows[i]);
	}
	free(rows); free(jobs); free(tid);
	return tot;
}

/* ------------------------------------------------------------------ RNG */

void orc_srand48(orc_rng_t *r, long seed) { r->x = ((uint64_t)(uint32_t)seed << 16) | 0x330E; }
double orc_drand48(orc_rng_t *r)
{
	r->x = (r->x * 0x5DEECE66DULL + 0xB) & 0xFFFFFFFFFFFFULL;
	return (double)r->x / 281474976710656.0;   /* 2^48; exact since x < 2^48 */
}

/* ------------------------------------------------------------------ global DP */

#define NINF (-1073741823)
#define FM 0
#define FI 1
#define FD 2

/* aln_global_core (stdaln.c:345-525) stated through per-row column ranges:
 * row j covers columns max(0,j-b2) .. min(len1, j+b1-1); row 0 covers 0..b1-1.
 * Column 0 and column len1 / row len2 use the end-gap penalty (set_end_* :286-319). */
int orc_global(const uint8_t *s1, int l1, const uint8_t *s2, int l2, int gap_open, int gap_ext, int gap_end,
			   const int *matrix, int row, int band, uint32_t *cig_out, int *n_cig)
{
	int b1, b2, i, j, W = l1 + 1, score, n = 0;
	int *M[2], *I[2], *D[2];
	uint8_t *tb, ctype, type, *path;   /* tb: Mt | It<<2 | Dt<<4 */
	int plen = 0;
	const int end_pen = gap_end >= 0 ? gap_end : gap_ext;
	*n_cig = 0;
	if (l1 == 0 || l2 == 0) return 0;
	if (l1 > l2) { b1 = l1 - l2 + band; b2 = band; } else { b1 = band; b2 = l2 - l1 + band; }
	if (b1 > l1) b1 = l1;
	if (b2 > l2) b2 = l2;
	for (i = 0; i < 2; ++i) {
		M[i] = (int*)malloc(W * sizeof(int)); I[i] = (int*)malloc(W * sizeof(int)); D[i] = (int*)malloc(W * sizeof(int));
	}
	tb = (uint8_t*)calloc((size_t)(l2 + 1) * W, 1);
	/* row 0 */
	M[0][0] = 0; I[0][0] = D[0][0] = NINF;
	for (i = 1; i < b1; ++i) {
		int t; M[0][i] = I[0][i] = NINF;
		if (M[0][i-1] - gap_open > D[0][i-1]) { t = FM; D[0][i] = M[0][i-1] - gap_open - end_pen; }
		else { t = FD; D[0][i] = D[0][i-1] - end_pen; }
		tb[i] = t << 4;
	}
	for (j = 1; j <= l2; ++j) {
		int *cm = M[j&1], *ci = I[j&1], *cd = D[j&1], *pm = M[(j-1)&1], *pi = I[(j-1)&1], *pd = D[(j-1)&1];
		int left = j > b2 ? j - b2 : 0, right = j + b1 - 1 < l1 ? j + b1 - 1 : l1;
		const int *mat = matrix + s2[j-1] * row;
		const int dpen = (j == l2) ? end_pen : gap_ext;
		uint8_t *t = tb + (size_t)j * W;
		cm[left] = ci[left] = cd[left] = NINF;
		if (left == 0) {   /* column 0: end-gap insertion chain */
			if (pm[0] - gap_open > pi[0]) { t[0] = FM << 2; ci[0] = pm[0] - gap_open - end_pen; }
			else { t[0] = FI << 2; ci[0] = pi[0] - end_pen; }
		}
		for (i = left + 1; i <= right; ++i) {
			int mt, it = 0, dt, sc = mat[s1[i-1]];
			/* set_M: prefer M, then D over I on ties as the macro does (stdaln.c:260-275) */
			if (pm[i-1] >= pi[i-1]) {
				if (pm[i-1] >= pd[i-1]) { cm[i] = pm[i-1] + sc; mt = FM; } else { cm[i] = pd[i-1] + sc; mt = FD; }
			} else {
				if (pi[i-1] > pd[i-1]) { cm[i] = pi[i-1] + sc; mt = FI; } else { cm[i] = pd[i-1] + sc; mt = FD; }
			}
			/* I: from the row above; the band's right edge has no cell above unless it is column l1 */
			if (i == right && !(j + b1 - 1 > l1)) ci[i] = NINF;
			else {
				int ipen = (i == l1) ? end_pen : gap_ext;
				if (pm[i] - gap_open > pi[i]) { it = FM; ci[i] = pm[i] - gap_open - ipen; }
				else { it = FI; ci[i] = pi[i] - ipen; }
			}
			if (cm[i-1] - gap_open > cd[i-1]) { dt = FM; cd[i] = cm[i-1] - gap_open - dpen; }
			else { dt = FD; cd[i] = cd[i-1] - dpen; }
			t[i] = mt | it << 2 | dt << 4;
		}
	}
	/* backtrace (stdaln.c:487-514) */
	{
		int *lm = M[l2&1], *li = I[l2&1], *ld = D[l2&1];
		uint8_t q = tb[(size_t)l2 * W + l1];
		i = l1; j = l2;
		score = lm[l1]; type = q & 3; ctype = FM;
		if (li[l1] > score) { score = li[l1]; type = q >> 2 & 3; ctype = FI; }
		if (ld[l1] > score) { score = ld[l1]; type = q >> 4 & 3; ctype = FD; }
	}
	path = (uint8_t*)malloc(l1 + l2 + 2);
	path[plen++] = ctype;
	do {
		uint8_t q;
		if (ctype == FM) { --i; --j; } else if (ctype == FI) --j; else --i;
		q = tb[(size_t)j * W + i];
		ctype = type;
		type = (type == FM) ? (q & 3) : (type == FI) ? (q >> 2 & 3) : (q >> 4 & 3);
		path[plen++] = ctype;
	} while (i || j);
	--plen;   /* the entry written at (0,0) is not part of the path */
	/* aln_path2cigar32 (stdaln.c:1009-1039): run-length encode from the path's end */
	for (i = plen - 1; i >= 0; --i) {
		if (n && (cig_out[n-1] & 0xf) == path[i]) cig_out[n-1] += 1u << 4;
		else cig_out[n++] = 1u << 4 | path[i];
	}
	*n_cig = n;
	for (i = 0; i < 2; ++i) { free(M[i]); free(I[i]); free(D[i]); }
	free(tb); free(path);
	return score;
}

/* ------------------------------------------------------------------ SE finishing chain */

static const int sm_maq[25] = { 11,-19,-19,-19,-13, -19,11,-19,-19,-13, -19,-19,11,-19,-13,
								-19,-19,-19,11,-13, -13,-13,-13,-13,-13 };   /* stdaln.c:206-212 */

static inline int pac_base(const uint8_t *pac, int64_t k) { return pac[k >> 2] >> ((~k & 3) << 1) & 3; }

#define CIG_OP(c) ((c) >> 14)
#define CIG_LEN(c) ((c) & 0x3fff)
#define CIG(op, len) ((uint16_t)((op) << 14 | (len)))

/* refine_gapped_core (bwase.c:189-237).  seq is the query in alignment orientation. */
static int refine_core(int64_t l_pac, const uint8_t *pac, int len, const uint8_t *seq, uint32_t *pos_io, int ext,
					   uint16_t *cigar)
{
	int ref_len = len + abs(ext), l = 0, n_cig, n, i;
	int64_t k, pos = *pos_io > l_pac ? (int64_t)(int32_t)*pos_io : (int64_t)*pos_io;
	uint8_t *ref = (uint8_t*)calloc(ref_len + 1, 1);
	uint32_t c32[256];
	if (ext > 0) {
		for (k = pos; k < pos + ref_len && k < l_pac; ++k) ref[l++] = pac_base(pac, k);
	} else {
		int64_t x = pos + len;
		for (k = x - ref_len > 0 ? x - ref_len : 0; k < x && k < l_pac; ++k) ref[l++] = pac_base(pac, k);
	}
	orc_global(ref, l, seq, len, 26, 9, 5, sm_maq, 5, 50, c32, &n_cig);
	n = n_cig;
	for (i = 0; i < n; ++i) cigar[i] = CIG(c32[i] & 0xf, c32[i] >> 4);
	if (ext < 0) {   /* forward strand: the end was right, move the start by the net indel */
		int d = 0;
		for (i = 0; i < n; ++i) {
			if (CIG_OP(cigar[i]) == FD) d -= CIG_LEN(cigar[i]);
			else if (CIG_OP(cigar[i]) == FI) d += CIG_LEN(cigar[i]);
		}
		pos += d;
	}
	if (CIG_OP(cigar[0]) == FD) {
		pos += CIG_LEN(cigar[0]);
		for (i = 0; i < n - 1; ++i) cigar[i] = cigar[i+1];
		--n;
	}
	if (CIG_OP(cigar[n-1]) == FD) --n;
	if (CIG_OP(cigar[n-1]) == FI) cigar[n-1] = CIG(3, CIG_LEN(cigar[n-1]));
	if (CIG_OP(cigar[0]) == FI) cigar[0] = CIG(3, CIG_LEN(cigar[0]));
	*pos_io = (uint32_t)pos;
	free(ref);
	return n;
}

/* reference base at pos, restoring ambiguity codes from the .amb holes (bwase.c:239-268) */
static int ref_base(const orc_index_t *ix, int64_t pos)
{
	int lo = 0, hi = ix->n_holes;
	while (lo < hi) {
		int mid = (lo + hi) >> 1;
		if (pos >= ix->holes[mid].offset + ix->holes[mid].len) lo = mid + 1;
		else if (pos < ix->holes[mid].offset) hi = mid;
		else return ix->holes[mid].amb;
	}
	return pac_base(ix->pac, pos);
}

/* bwa_cal_md1 (bwase.c:253-315) */
static void cal_md(const orc_index_t *ix, int n_cigar, const uint16_t *cigar, int len, uint32_t pos0,
				   const uint8_t *seq, char *md, int *nm_out)
{
	int64_t pos = pos0; int u = 0, nm = 0, y = 0, z, k; char *o = md;
#define MD_BASE(c) ((c) > 3 ? (char)(c) : "ACGT"[c])
	if (n_cigar) {
		for (k = 0; k < n_cigar; ++k) {
			int l = CIG_LEN(cigar[k]), op = CIG_OP(cigar[k]);
			if (op == FM) {
				for (z = 0; z < l && pos < ix->l_pac; ++z, ++y, ++pos) {
					int c = ref_base(ix, pos);
					if (c > 3 || seq[y] > 3 || c != seq[y]) { o += sprintf(o, "%d%c", u, MD_BASE(c)); ++nm; u = 0; }
					else ++u;
				}
			} else if (op == FI || op == 3) {
				y += l;
				if (op == FI) nm += l;
			} else {
				o += sprintf(o, "%d^", u);
				for (z = 0; z < l && pos < ix->l_pac; ++z, ++pos) { int c = ref_base(ix, pos); *o++ = MD_BASE(c); }
				u = 0; nm += l;
			}
		}
	} else {
		for (z = 0; z < len; ++z, ++pos) {
			int c = ref_base(ix, pos);
			if (c > 3 || seq[z] > 3 || c != seq[z]) { o += sprintf(o, "%d%c", u, MD_BASE(c)); ++nm; u = 0; }
			else ++u;
		}
	}
	sprintf(o, "%d", u);
	*nm_out = nm;
}

/* bns_coor_pac2real (bntseq.c:272-306) */
int orc_pac2real(const orc_index_t *ix, int64_t pac_coor, int len, int *seqid)
{
	int left = 0, mid = 0, right = ix->n_seqs, nn = 0;
	while (left < right) {
		mid = (left + right) >> 1;
		if (pac_coor >= ix->anns[mid].offset) {
			if (mid == ix->n_seqs - 1) break;
			if (pac_coor < ix->anns[mid+1].offset) break;
			left = mid + 1;
		} else right = mid;
	}
	*seqid = mid;
	left = 0; right = ix->n_holes;
	while (left < right) {
		int m = (left + right) >> 1;
		const orc_hole_t *h = ix->holes + m;
		if (pac_coor >= h->offset + h->len) left = m + 1;
		else if (pac_coor + len <= h->offset) right = m;
		else {
			if (pac_coor >= h->offset) nn += h->offset + h->len < pac_coor + len ? h->offset + h->len - pac_coor : len;
			else nn += h->offset + h->len < pac_coor + len ? h->len : len - (h->offset - pac_coor);
			break;
		}
	}
	return nn;
}

static int g_logn(int n) { return (int)(4.343 * log(n) + 0.5); }   /* bwase.c:613-617 */

void orc_se_finish(const orc_index_t *ix, const orc_opt_t *opt, orc_rng_t *rng, int len, int full_len,
				   const uint8_t *seq_rev, const uint8_t *rseq, int n_aln, const orc_aln_t *aln,
				   int n_occ, orc_se_t *s)
{
	int i, j; uint8_t *fwd;
	memset(s, 0, sizeof(*s));
	s->len = len; s->full_len = full_len; s->clip_len = len;
	/* ---- bwa_aln2seq_core (bwase.c:19-95) */
	if (n_aln == 0) { s->type = 0; }
	else {
		int best = aln[0].score, cnt = 0;
		for (i = 0; i < n_aln; ++i) {
			const orc_aln_t *p = aln + i; uint32_t wdt = p->l - p->k + 1;
			if (p->score > best) break;
			if (orc_drand48(rng) * (wdt + cnt) > (double)cnt) {
				s->n_mm = ORC_ALN_MM(*p); s->n_gapo = ORC_ALN_GAPO(*p); s->n_gape = ORC_ALN_GAPE(*p);
				s->strand = ORC_ALN_A(*p); s->score = p->score;
				s->sa = p->k + (uint32_t)(wdt * orc_drand48(rng));
			}
			cnt += wdt;
		}
		s->c1 = cnt & 0xfffffff;
		for (; i < n_aln; ++i) cnt += aln[i].l - aln[i].k + 1;
		s->c2 = (cnt - s->c1) & 0xfffffff;
		s->type = s->c1 > 1 ? 2 : 1;
		if (n_occ) {
			uint64_t tot = 0; int z = 0, rest;
			for (i = 0; i < n_aln; ++i) tot += aln[i].l - aln[i].k + 1;
			if (tot <= (uint64_t)n_occ + 1) {
				rest = (int)tot;
				for (i = 0; i < n_aln; ++i) {
					const orc_aln_t *q = aln + i; uint32_t r;
					for (r = q->k; r <= q->l; ++r) {   /* all hits fit: list every row (bwase.c:66-74) */
						s->multi[z].pos = r; s->multi[z].gap = ORC_ALN_GAPO(*q) + ORC_ALN_GAPE(*q);
						s->multi[z].mm = ORC_ALN_MM(*q); s->multi[z].strand = ORC_ALN_A(*q); ++z;
					}
					rest -= q->l - q->k + 1;
				}
				for (i = j = 0; i < z; ++i) if (s->multi[i].pos != s->sa) s->multi[j++] = s->multi[i];
				s->n_multi = j < n_occ ? j : n_occ;
			}
		}
	}
	/* ---- bwa_cal_pac_pos_core (bwase.c:139-154) and multi positions (bwase.c:166-181) */
	if (s->type == 1 || s->type == 2) {
		int max_diff = opt->fnr > 0.0 ? orc_maxdiff(len, 0.02, opt->fnr) : opt->max_diff;
		if (s->strand) s->pos = orc_sa(&ix->bwt[0], s->sa);
		else s->pos = ix->bwt[1].seq_len - (orc_sa(&ix->bwt[1], s->sa) + len);
		/* bwa_approx_mapQ (bwase.c:113-122) */
		if (s->c1 == 0) s->mapQ = 23;
		else if (s->c1 > 1) s->mapQ = 0;
		else if (s->n_mm == max_diff) s->mapQ = 25;
		else if (s->c2 == 0) s->mapQ = 37;
		else { int n = s->c2 >= 255 ? 255 : (int)s->c2, g = g_logn(n); s->mapQ = 23 < g ? 0 : 23 - g; }
		s->seQ = s->mapQ;
	}
	for (j = 0; j < s->n_multi; ++j) {
		if (s->multi[j].strand) s->multi[j].pos = orc_sa(&ix->bwt[0], s->multi[j].pos);
		else s->multi[j].pos = ix->bwt[1].seq_len - (orc_sa(&ix->bwt[1], s->multi[j].pos) + len);
	}
	/* ---- bwa_refine_gapped (bwase.c:356-423); seq is un-reversed first */
	fwd = (uint8_t*)malloc(len + 1);
	for (i = 0; i < len; ++i) fwd[i] = seq_rev[len - 1 - i];
	for (j = 0; j < s->n_multi; ++j) {
		if (s->multi[j].gap == 0) continue;
		s->multi[j].n_cigar = refine_core(ix->l_pac, ix->pac, len, s->multi[j].strand ? rseq : fwd, &s->multi[j].pos,
										  (s->multi[j].strand ? 1 : -1) * s->multi[j].gap, s->multi[j].cigar);
	}
	if (s->type != 0 && s->n_gapo)
		s->n_cigar = refine_core(ix->l_pac, ix->pac, len, s->strand ? rseq : fwd, &s->pos,
								 (s->strand ? 1 : -1) * (s->n_gapo + s->n_gape), s->cigar);
	if (s->type != 0) cal_md(ix, s->n_cigar, s->cigar, len, s->pos, s->strand ? rseq : fwd, s->md, &s->nm);
	free(fwd);
	/* bwa_correct_trimmed (bwase.c:320-354) */
	if (len != full_len) {
		int clip = full_len - len;
		if (s->strand == 0) {
			if (s->n_cigar && CIG_OP(s->cigar[s->n_cigar-1]) == 3) s->cigar[s->n_cigar-1] += clip;
			else {
				if (s->n_cigar == 0) { s->n_cigar = 2; s->cigar[0] = CIG(0, len); } else ++s->n_cigar;
				s->cigar[s->n_cigar-1] = CIG(3, clip);
			}
		} else {
			if (s->n_cigar && CIG_OP(s->cigar[0]) == 3) s->cigar[0] += clip;
			else {
				if (s->n_cigar == 0) { s->n_cigar = 2; s->cigar[1] = CIG(0, len); }
				else { ++s->n_cigar; memmove(s->cigar + 1, s->cigar, (s->n_cigar - 1) * 2); }
				s->cigar[0] = CIG(3, clip);
			}
		}
		s->len = full_len;
	}
	/* ---- flag / coordinate / XT as bwa_print_sam1 derives them for SE (bwase.c:458-571) */
	s->flag = 0;
	if (s->type != 0) {
		int64_t end = s->pos; int reflen;
		if (s->n_cigar) { for (i = 0; i < s->n_cigar; ++i) { int op = CIG_OP(s->cigar[i]); if (op == 0 || op == 2) end += CIG_LEN(s->cigar[i]); } }
		else end = (int64_t)s->pos + s->len;
		reflen = (int)(end - s->pos);
		s->nn = orc_pac2real(ix, s->pos, reflen, &s->seqid);
		if ((int64_t)s->pos + reflen - ix->anns[s->seqid].offset > ix->anns[s->seqid].len) { s->flag |= 4; s->mapQ = 0; }
		if (s->strand) s->flag |= 16;
		s->rpos = (int64_t)s->pos - ix->anns[s->seqid].offset + 1;
		s->xt = "NURM"[s->type];
		if (s->nn > 10) s->xt = 'N';
	} else s->flag = 4;
}
