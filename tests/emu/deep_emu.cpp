// tests/emu/deep_emu.cpp -- TEST INFRASTRUCTURE ONLY: kernel D (network-aware-bwa_amd/csrc/fm_deep_body.hpp) compiled by g++ as a
// sequential emulation of one wavefront (NABWA_EMU, wave_spmd.hpp), so that its speculative rounds and ordered commits
// can be checked against the oracle here, under AddressSanitizer, without a GPU.  Nothing in libnabwa.so links this.
//
// Around the kernel body this file holds plain CPU stand-ins for what other GPU kernels produce for it: the 64-byte
// bucket array (repack_kernel, fm_index.hip) and the per-read width records (fm_width_kernel, fm_search.hip).
#define NABWA_EMU 1
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "../../network-aware-bwa_amd/csrc/fm_deep_body.hpp"

namespace {

uint32_t ref_base(const uint32_t *w, uint32_t j)          // base j of the reference's .bwt word stream (bwt.h:61-66)
{
	const uint32_t *p = w + (size_t)(j >> 7) * 12 + 4;
	return p[(j & 127u) >> 4] >> ((~j & 15u) << 1) & 3u;
}

struct EmuBwt { std::vector<uint4> bk; std::vector<uint32_t> sa, sa_full, isa, text; std::vector<uint2> kmer; DevBwt B; };

void build(EmuBwt &X, const uint32_t *words)
{
	memset(&X.B, 0, sizeof(X.B));
	X.B.primary = words[0]; X.B.L2[0] = 0; X.B.L2[1] = words[1]; X.B.L2[2] = words[2]; X.B.L2[3] = words[3]; X.B.seq_len = words[4];
	const uint32_t *w = words + 5, n = X.B.seq_len;
	const uint32_t nb = (uint32_t)(((uint64_t)n + NABWA_INTV - 1) / NABWA_INTV);
	X.bk.assign((size_t)nb * 4, make_uint4(0, 0, 0, 0));
	uint32_t cnt[4] = { 0, 0, 0, 0 };
	for (uint32_t b = 0; b < nb; ++b) {
		uint64_t lo[3] = { 0, 0, 0 }, hi[3] = { 0, 0, 0 };
		const uint32_t c0[4] = { cnt[0], cnt[1], cnt[2], cnt[3] };
		for (uint32_t r = 0; r < NABWA_INTV; ++r) {
			const uint32_t j = b * NABWA_INTV + r;
			if (j >= n) break;
			const uint32_t c = ref_base(w, j);
			++cnt[c];
			lo[r >> 6] |= (uint64_t)(c & 1u) << (r & 63u); hi[r >> 6] |= (uint64_t)(c >> 1) << (r & 63u);
		}
		uint4 *o = &X.bk[(size_t)b * 4];
		o[0] = make_uint4(c0[0], c0[1], c0[2], c0[3]);
		for (int g = 0; g < 3; ++g) o[1 + g] = make_uint4((uint32_t)lo[g], (uint32_t)(lo[g] >> 32), (uint32_t)hi[g], (uint32_t)(hi[g] >> 32));
	}
	X.B.bk = X.bk.data(); X.B.n_buckets = nb;
}

// the interval table of an index, levels 1 .. T back to back, as kmer_level_kernel (fm_index.hip) builds it: the children of every string
// of t - 1 symbols by one backward step, {1, 0} for a string that does not occur
void build_table(EmuBwt &X, int T)
{
	DevBwt &B = X.B;
	size_t n = 0;
	for (int t = 1; t <= T; ++t) n += (size_t)1 << (2 * t);
	X.kmer.assign(n, make_uint2(1u, 0u));
	size_t prev = 0, cur = 0;
	for (int t = 1; t <= T; ++t) {
		const size_t n_par = (size_t)1 << (2 * (t - 1));
		for (size_t idx = 0; idx < n_par; ++idx) {
			const uint2 par = t == 1 ? make_uint2(0u, B.seq_len) : X.kmer[prev + idx];
			if (par.x > par.y) continue;
			Occ4 ck, cl;
			nabwa_occ4_pair(B, par.x - 1u, par.y, ck, cl);
			for (int c = 0; c < 4; ++c) {
				const uint32_t k = B.L2[c] + ck.c[c] + 1u, l = B.L2[c] + cl.c[c];
				if (k <= l) X.kmer[cur + 4 * idx + c] = make_uint2(k, l);
			}
		}
		prev = cur; cur += (size_t)1 << (2 * t);
	}
	B.kmer_lo = X.kmer.data(); B.kmer = X.kmer.data() + prev; B.kmer_T = B.kmer_LW = (uint32_t)T;
}

// the text-mode companions of an index -- SA value of every row, its inverse, the text 2 bits per base -- from the BWT and the
// SA samples of the .sa file (7 header words, then SA[j * intv] for j >= 1): one LF walk per sample, as sa_fill_kernel does
void build_text(EmuBwt &X, const uint32_t *sa_words)
{
	DevBwt &B = X.B;
	B.sa_intv = sa_words[5];
	B.n_sa = (uint32_t)(((uint64_t)B.seq_len + B.sa_intv) / B.sa_intv);
	X.sa.assign(B.n_sa, 0xffffffffu);
	for (uint32_t j = 1; j < B.n_sa; ++j) X.sa[j] = sa_words[7 + j - 1];
	const size_t rows = (size_t)B.seq_len + 1, words = ((size_t)B.seq_len + 15) / 16 + 4;
	X.sa_full.assign(rows, 0); X.isa.assign(rows, 0); X.text.assign(words, 0);
	std::vector<uint8_t> tb(rows, 0);
	for (uint32_t j = 0; j < B.n_sa; ++j) {
		uint32_t row = j * B.sa_intv, v = j ? X.sa[j] : B.seq_len;
		X.sa_full[row] = j ? v : 0xffffffffu;
		X.isa[v] = row;
		for (;;) {
			if (row == B.primary) break;
			const uint32_t kp = row - (row > B.primary ? 1u : 0u);
			const uint32_t b = kp / NABWA_INTV, r = kp - b * NABWA_INTV;
			const uint4 *p = B.bk + (size_t)b * 4;
			const uint32_t c = nabwa_base_at(p[1], p[2], p[3], r);
			const Occ4 o = nabwa_count4(p[0], p[1], p[2], p[3], r);
			v -= 1u;
			tb[v] = (uint8_t)c;
			row = B.L2[c] + o.c[c];
			if (row % B.sa_intv == 0u) break;
			X.sa_full[row] = v; X.isa[v] = row;
		}
	}
	for (size_t j = 0; j < B.seq_len; ++j) X.text[j >> 4] |= (uint32_t)(tb[j] & 3u) << (2u * (j & 15u));
	B.sa = X.sa.data(); B.sa_full = X.sa_full.data(); B.isa = X.isa.data(); B.text = X.text.data();
}

// bwt_cal_width (bwtaln.c:52-76) into the record layout kernel W writes: widths, bound bytes min(bid,127) | (w[p-1]==w[p]) << 7
void cal_width(const DevBwt &B, int len, const uint8_t *str, uint32_t *wd, uint8_t *bd, unsigned long long *touches)
{
	uint32_t k = 0, l = B.seq_len, pw = 0; int bid = 0;
	for (int i = 0; i < len; ++i) {
		const uint32_t c = str[i];
		if (c < 4) {
			Occ4 ck, cl;
			nabwa_occ4_pair(B, k - 1u, l, ck, cl);
			*touches += ref_touches(B, k - 1u, l, false);
			k = B.L2[c] + ck.c[c] + 1u; l = B.L2[c] + cl.c[c];
		}
		if (k > l || c > 3) { k = 0; l = B.seq_len; ++bid; }
		const uint32_t wv = l - k + 1u;
		wd[i] = wv; bd[i] = (uint8_t)((bid > 127 ? 127 : bid) | ((i > 0 && wv == pw) ? 128 : 0));
		pw = wv;
	}
	++bid;
	wd[len] = 0; bd[len] = (uint8_t)((bid > 127 ? 127 : bid) | ((len > 0 && 0u == pw) ? 128 : 0));
}

int cal_maxdiff(int l, double err, double thres)       // bwa_cal_maxdiff (bwtaln.c:37-49)
{
	double elambda = exp(-l * err), sum = elambda, y = 1.0;
	uint32_t x = 1;
	for (int k = 1; k < 1000; ++k) {
		y *= l * err; x *= (uint32_t)k;
		sum += elambda * y / (int32_t)x;
		if (1.0 - sum < thres) return k;
	}
	return 2;
}

uint32_t align_up(uint32_t x, uint32_t a) { return (x + a - 1) / a * a; }

}  // namespace

struct emu_opt { int s_mm, s_gapo, s_gape, mode, indel_end_skip, max_del_occ, max_entries; float fnr; int max_diff, max_gapo, max_gape, max_seed_diff, seed_len, n_threads, max_top2, trim_qual; };

// knobs: [8] 1 = the read's own data in (emulated) LDS;  [0] max_lanes, [1] careful_all, [2] stage_k, [3] n_pages, [4] own_cap, [5] reads per wave (0: one wave takes all), [6] aln_cap,
// [7] text mode (needs sa0 / sa1: the .sa / .rsa file contents; touches are then not the reference's), [9] depth of the interval tables for key-form entries (0: none; the
// touch counter is then off, as in the product), [10] coop_lanes (DeepParams)
// stats: 16 words -- 8 as in DeepParams, [8] the reference's bucket touches in the width passes, [9] in bwt_match_gap, [10] expansions in key form, [11] their tail jumps / hits,
// [12] forced walks through the table, [13] forced walks on the text.  Returns 0.
extern "C" int emu_deep_search(const uint32_t *bwt0, const uint32_t *bwt1, const uint32_t *sa0, const uint32_t *sa1, const emu_opt *opt, int n, const int64_t *off,
							   const uint8_t *seq, const uint8_t *rseq, int per_read, const int *knobs,
							   int32_t *n_aln, uint32_t *rows /* n x aln_cap x 4 */, int32_t *max_ent, uint8_t *status,
							   unsigned long long *stats)
{
	EmuBwt X[2];
	build(X[0], bwt0); build(X[1], bwt1);
	if (knobs[7] && sa0 && sa1) { build_text(X[0], sa0); build_text(X[1], sa1); }
	if (knobs[9] > 0) { build_table(X[0], knobs[9]); build_table(X[1], knobs[9]); }
	int max_len = 0;
	for (int i = 0; i < n; ++i) if (off[i + 1] - off[i] > max_len) max_len = (int)(off[i + 1] - off[i]);
	DeepParams P;
	memset(&P, 0, sizeof(P));
	SearchParams &S = P.S;
	S.bwt[0] = X[0].B; S.bwt[1] = X[1].B;
	uint32_t ixtab[NABWA_IXTAB_WORDS];
	nabwa_ixtab_fill(ixtab, S.bwt);
	S.ixtab = ixtab;
	// reads, padded to 16-byte starts as pad_reads_kernel lays them out
	std::vector<int64_t> poff(n + 1, 0);
	for (int i = 0; i < n; ++i) poff[i + 1] = poff[i] + (off[i + 1] - off[i] + 15) / 16 * 16;
	std::vector<uint8_t> ps(poff[n] + 64, 4), pr(poff[n] + 64, 4);
	std::vector<int32_t> rd_len(n);
	std::vector<uint8_t> md(n), mg(n), nN(n);
	const int md_batch = opt->fnr > 0.0f ? cal_maxdiff(max_len, 0.02, opt->fnr) : opt->max_diff;
	uint32_t NS = 1;
	for (int i = 0; i < n; ++i) {
		const int L = (int)(off[i + 1] - off[i]);
		memcpy(&ps[poff[i]], seq + off[i], L); memcpy(&pr[poff[i]], rseq + off[i], L);
		rd_len[i] = L;
		const int d = opt->fnr > 0.0f ? cal_maxdiff(L, 0.02, opt->fnr) : opt->max_diff;
		const int sizing = per_read ? d : md_batch;
		int g = opt->max_gapo; if (sizing < g) g = sizing;
		md[i] = (uint8_t)d; mg[i] = (uint8_t)g;
		const uint32_t ns = (uint32_t)((d + 2) * opt->s_mm + (g + 1) * opt->s_gapo + (opt->max_gape + 1) * opt->s_gape + 1);
		if (ns > NS) NS = ns;
		int c = 0; for (int j = 0; j < L; ++j) if (seq[off[i] + j] > 3) ++c;
		nN[i] = (uint8_t)(c > 255 ? 255 : c);
	}
	S.seq = ps.data(); S.rseq = pr.data(); S.poff = poff.data(); S.rd_len = rd_len.data(); S.rd_maxdiff = md.data(); S.rd_maxgapo = mg.data();
	S.rd_nN = nN.data(); S.n = n;
	S.s_mm = opt->s_mm; S.s_gapo = opt->s_gapo; S.s_gape = opt->s_gape; S.mode = opt->mode; S.indel_end_skip = opt->indel_end_skip;
	S.max_del_occ = opt->max_del_occ; S.max_entries = opt->max_entries; S.max_gape = opt->max_gape; S.max_seed_diff = opt->max_seed_diff;
	S.seed_len = opt->seed_len; S.max_top2 = opt->max_top2; S.text_mode = knobs[7] ? 2 : 0;
	// width records (layout of nabwa_api.hip: layout())
	S.WL = align_up((uint32_t)max_len + 1, 16); S.WLB = S.WL + 16; S.SLB = align_up((uint32_t)(opt->seed_len > 65535 ? 0 : opt->seed_len) + 1, 16) + 16;
	S.woff_bid = 2 * S.WL * 4; S.woff_sbid = S.woff_bid + 2 * S.WLB; S.wstride = align_up(S.woff_sbid + 2 * S.SLB, 64);
	std::vector<uint8_t> wdata((size_t)(n ? n : 1) * S.wstride, 0);
	std::vector<uint32_t> tmpw(max_len + 2);
	unsigned long long wt = 0, st = 0;
	for (int i = 0; i < n; ++i) {
		uint8_t *rec = &wdata[(size_t)i * S.wstride];
		const int L = rd_len[i];
		if (L <= 0) continue;
		for (int x = 0; x < 2; ++x) {
			const uint8_t *str = (x ? pr.data() : ps.data()) + poff[i];
			cal_width(X[x].B, L, str, (uint32_t*)rec + x * S.WL, rec + S.woff_bid + x * S.WLB, &wt);
			if (L > opt->seed_len) cal_width(X[x].B, opt->seed_len, str + (L - opt->seed_len), tmpw.data(), rec + S.woff_sbid + x * S.SLB, &wt);
		}
	}
	S.wdata = wdata.data();
	const int aln_cap = knobs[6];
	S.n_aln = n_aln; S.max_ent = max_ent; S.status = status; S.aln = (uint4*)rows; S.aln_cap = aln_cap;
	unsigned int counter = 0, bump = 0;
	S.work_counter = &counter;
	S.touch_counter = knobs[9] > 0 ? 0 : &st;
	P.key_T = knobs[9] > 0 ? (uint32_t)knobs[9] : 0u;
	P.coop_lanes = (uint32_t)knobs[10];
	P.n_pages = (uint32_t)knobs[3]; P.own_cap = (uint32_t)knobs[4]; P.stage_k = (uint32_t)(knobs[2] < 1 ? 1 : (knobs[2] > (int)DEEP_STAGE_MAX ? (int)DEEP_STAGE_MAX : knobs[2]));
	P.max_lanes = knobs[0]; P.careful_all = knobs[1]; P.NS = NS; P.stats = stats;
	const int per_wave = knobs[5];
	const int n_waves = per_wave > 0 ? (n + per_wave - 1) / per_wave : 1;
	std::vector<uint4> pages((size_t)P.n_pages * DEEP_PAGE);
	std::vector<uint32_t> prev(P.n_pages, 0), own((size_t)n_waves * 2 * P.own_cap, 0);
	std::vector<uint4> stage((size_t)n_waves * 64 * P.stage_k * 4);
	P.rd_pl = align_up((uint32_t)(max_len > 0 ? max_len : 1), 16);
	P.lds_rd = knobs[8] ? 2u * S.WLB + 2u * S.SLB + 2u * P.rd_pl : 0u;
	std::vector<uint32_t> lds(DEEP_LDS_WORDS(NS, P.lds_rd) + 4);
	P.pages = pages.data(); P.page_prev = prev.data(); P.page_bump = &bump; P.own = own.data(); P.stage = stage.data();
	for (int w = 0; w < n_waves; ++w) {
		S.n = per_wave > 0 ? ((w + 1) * per_wave < n ? (w + 1) * per_wave : n) : n;
		if (P.lds_rd) deep_wave_body<true, true, true>(P, (uint32_t*)(((uintptr_t)lds.data() + 15) & ~(uintptr_t)15), (uint32_t)w);
		else deep_wave_body<true, false, true>(P, (uint32_t*)(((uintptr_t)lds.data() + 15) & ~(uintptr_t)15), (uint32_t)w);
		counter = (unsigned int)S.n;     /* (the wave's last, failed draw took a number: on the GPU all waves draw until the reads are gone) */
	}
	stats[8] = wt; stats[9] = st;
	return 0;
}
