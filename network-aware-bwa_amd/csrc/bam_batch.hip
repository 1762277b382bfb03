// bam_batch.hip -- the batching front-end behind `bwa bam2bam` / `bwa worker`: BAM records in, BAM records out.
//
// The reference handles one logical record (a singleton or a pair) at a time: read_bam_pair -> pair_aln -> pair_posn ->
// improve_isize_est -> [all records] -> infer_all_isizes -> pair_finish -> bwa_update_bam1 (bam2bam.c:1143-1216, 608-811,
// 430-593; bwaseqio.c:340-494; insert_size.c:141-213).  Here the same steps run over a BATCH of records:
//   create : split the records into singletons and pairs (read_bam_pair_core's rules), OR the QC flag over mates, erase the
//            tags the aligner regenerates (erase_unwanted_tags), encode the reads (bam1_to_seq incl. reverse flag and trimming)
//   pass 1 : bwa_cal_sa_reg_gap of every read [GPU, kernels W / S / D], the hit choice IN RECORD ORDER on the caller's drand48
//            stream (posn_singleton: bwa_aln2seq_core(.., 1, max_occ_se); posn_pair: bwa_aln2seq), all bwt_sa walks as one
//            GPU batch, mapQ, and the per-@RG insert-size histograms (improve_isize_est)
//   pass 2 : finish_singleton / finish_pair per read group with that group's estimate (pairing, mate rescue and gap
//            refinement as GPU batches inside nabwa_pe_finish / nabwa_se_refine), then bwa_update_bam1: flags, coordinates,
//            bin, CIGAR, reverse-complemented SEQ/QUAL, mate fields, tags in the reference's order and types
// All host code; the GPU work is what the entry points it calls do.  bam2bam.c itself cannot be compiled in the build
// container (<zmq.h>), so the BAM-specific bytes are checked through every field the reference's samse / sampe SAM exposes
// (tests/test_gpu_bam.py), not against a bam2bam run: DESIGN.md says so.
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <chrono>
#include <functional>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>
#include "../../include/nabwa.h"
#include "nabwa_internal.hpp"
#include "finish_common.hpp"

void nabwa_poscache_register(nabwa_poscache_t *cache, int max_occ, int n, const int *first, const int32_t *n_aln, const int64_t *row0,
							 const nabwa_aln1_t *rows, const nabwa_pe_t *res);                                                       /* pe_finish.hip */
/* the single-end chain on records of any stride whose head is a nabwa_se_t (se_finish.hip): this file works in place */
int nabwa_se_posn_strided(nabwa_index_t *ix, const nabwa_gap_opt_t *opt, int n, const int64_t *off, const int32_t *full_len,
						  const int32_t *n_aln, const nabwa_aln1_t *aln, const uint8_t *n_occ_v, uint64_t *rng48, void *out_base, size_t stride);
int nabwa_se_refine_strided(nabwa_index_t *ix, int n, const int64_t *off, const uint8_t *seq, const uint8_t *rseq, void *out_base, size_t stride);

/* slices of independent records on the host's threads */
static int bam_threads(size_t n)
{
	int nt = (int)std::thread::hardware_concurrency(); if (nt < 1) nt = 1; if (nt > 16) nt = 16;
	if (getenv("NABWA_HOST_THREADS")) nt = atoi(getenv("NABWA_HOST_THREADS")) > 0 ? atoi(getenv("NABWA_HOST_THREADS")) : 1;
	if (n < 8192) nt = 1;
	return nt;
}
static double bam_now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static void bam_parallel(size_t n, const std::function<void(int, size_t, size_t)> &f)
{
	const int nt = bam_threads(n);
	if (nt == 1) { f(0, 0, n); return; }
	std::vector<std::thread> th;
	for (int t = 0; t < nt; ++t) th.emplace_back(f, t, n * t / nt, n * (t + 1) / nt);
	for (auto &x : th) x.join();
}

#define F_PD 1
#define F_PP 2
#define F_SU 4
#define F_MU 8
#define F_SR 16
#define F_MR 32
#define F_R1 64
#define F_R2 128
#define F_SC 256
#define F_QC 512
#define F_DP 1024

/* ------------------------------------------------------------------ per-@RG insert-size table (insert_size.c:141-213) */

struct nabwa_isize_table {
	struct Rg { nabwa_isize_t ii; std::vector<uint16_t> hist; bool has_hist; };
	std::map<std::string, Rg> rg;        /* (the reference keeps a khash; its iteration order only decides the order of log lines) */
	double ap_prior; int64_t L;
	nabwa_poscache_t *poscache;          /* finish_pair's position cache of the file (bam2bam.c:1186-1203): lives as long as pass 2 does, like this table */
};

extern "C" nabwa_isize_table_t *nabwa_isize_table_create(double ap_prior, int64_t genome_len)
{
	nabwa_isize_table *t = new nabwa_isize_table();
	t->ap_prior = ap_prior; t->L = genome_len; t->poscache = nabwa_poscache_create();
	return t;
}
extern "C" void nabwa_isize_table_destroy(nabwa_isize_table_t *t) { if (t) nabwa_poscache_destroy(t->poscache); delete t; }

/* improve_isize_est (insert_size.c:141-165): one logical record's contribution.  The 16-bit bins wrap as the reference's do
 * (its "hit the ceiling" test compares an unsigned short with -1 and never fires). */
static nabwa_isize_table::Rg *isize_slot(nabwa_isize_table *t, const std::string &rg)       /* the read group's entry, made on first use */
{
	auto it = t->rg.find(rg);
	if (it == t->rg.end()) {
		nabwa_isize_table::Rg r; memset(&r.ii, 0, sizeof(r.ii)); r.hist.assign(100000, 0); r.has_hist = true;
		it = t->rg.emplace(rg, std::move(r)).first;
	}
	return &it->second;
}

/* infer_all_isizes (insert_size.c:167-173): every read group that still has its histogram gets its estimate */
extern "C" int nabwa_isize_table_infer_all(nabwa_isize_table_t *t)
{
	if (!t) return nabwa_fail(NABWA_EINVAL, "null argument");
	for (auto &kv : t->rg)
		if (kv.second.has_hist) {
			nabwa_isize_infer(kv.second.hist.data(), t->ap_prior, t->L, &kv.second.ii);
			kv.second.hist.clear(); kv.second.hist.shrink_to_fit(); kv.second.has_hist = false;
		}
	return NABWA_OK;
}

extern "C" int nabwa_isize_table_get(const nabwa_isize_table_t *t, const char *rg, nabwa_isize_t *out)
{
	if (!t || !rg || !out) return nabwa_fail(NABWA_EINVAL, "null argument");
	auto it = t->rg.find(rg);
	if (it == t->rg.end() || it->second.has_hist) { memset(out, 0, sizeof(*out)); return 1; }      /* null_ii (bam2bam.c:106,715) */
	*out = it->second.ii;
	return NABWA_OK;
}

extern "C" int nabwa_isize_table_merge(nabwa_isize_table_t *t, const nabwa_isize_table_t *other)      /* the host add between passes of N shards */
{
	if (!t || !other) return nabwa_fail(NABWA_EINVAL, "null argument");
	for (const auto &kv : other->rg) {
		if (!kv.second.has_hist) continue;
		auto it = t->rg.find(kv.first);
		if (it == t->rg.end()) { t->rg.emplace(kv.first, kv.second); continue; }
		if (!it->second.has_hist) continue;
		for (size_t b = 0; b < 100000; ++b) it->second.hist[b] = (uint16_t)(it->second.hist[b] + kv.second.hist[b]);
	}
	return NABWA_OK;
}

/* encode_iinfo / decode_iinfo (insert_size.c:185-213): the blob `bwa worker` receives -- per read group its name, NUL, then the
 * raw isize_info_t (a dead histogram pointer, then avg, std, ap_prior, low, high, high_bayesian: 48 bytes) */
extern "C" int64_t nabwa_isize_table_encode(const nabwa_isize_table_t *t, uint8_t *out, int64_t cap)
{
	if (!t) return nabwa_fail(NABWA_EINVAL, "null argument");
	int64_t need = 0;
	for (const auto &kv : t->rg) need += (int64_t)kv.first.size() + 1 + 8 + (int64_t)sizeof(nabwa_isize_t);
	if (!out || cap < need) return need;
	uint8_t *p = out;
	for (const auto &kv : t->rg) {
		memcpy(p, kv.first.c_str(), kv.first.size() + 1); p += kv.first.size() + 1;
		memset(p, 0, 8); p += 8;
		memcpy(p, &kv.second.ii, sizeof(nabwa_isize_t)); p += sizeof(nabwa_isize_t);
	}
	return need;
}
extern "C" int nabwa_isize_table_decode(nabwa_isize_table_t *t, const uint8_t *in, int64_t n)
{
	if (!t || (n && !in)) return nabwa_fail(NABWA_EINVAL, "null argument");
	const uint8_t *p = in, *q = in + n;
	while (p < q) {
		const size_t l = strnlen((const char*)p, (size_t)(q - p));
		if (p + l + 1 + 8 + sizeof(nabwa_isize_t) > q) return nabwa_fail(NABWA_EINVAL, "error when decoding isize info");
		nabwa_isize_table::Rg r; r.has_hist = false;
		memcpy(&r.ii, p + l + 1 + 8, sizeof(nabwa_isize_t));
		t->rg[std::string((const char*)p, l)] = r;
		p += l + 1 + 8 + sizeof(nabwa_isize_t);
	}
	return NABWA_OK;
}

/* ------------------------------------------------------------------ BAM records */

/* The bytes of one record.  They start out in the batch's arena, with room for what pass 2 adds (a million records = one
 * allocation, not a million), and move to the heap only if they outgrow that room. */
struct RecBuf {
	uint8_t *p; uint32_t n, cap; bool heap;
	RecBuf() : p(0), n(0), cap(0), heap(false) {}
	~RecBuf() { if (heap) free(p); }
	RecBuf(const RecBuf&) = delete;
	RecBuf &operator=(const RecBuf&) = delete;
	RecBuf(RecBuf &&o) noexcept : p(o.p), n(o.n), cap(o.cap), heap(o.heap) { o.p = 0; o.n = o.cap = 0; o.heap = false; }
	RecBuf &operator=(RecBuf &&o) noexcept
	{
		if (this != &o) { if (heap) free(p); p = o.p; n = o.n; cap = o.cap; heap = o.heap; o.p = 0; o.n = o.cap = 0; o.heap = false; }
		return *this;
	}
	uint8_t *data() { return p; }
	const uint8_t *data() const { return p; }
	size_t size() const { return n; }
	bool empty() const { return n == 0; }
	void place(uint8_t *at, size_t room, const uint8_t *src, size_t len) { if (heap) free(p); p = at; cap = (uint32_t)room; heap = false; n = (uint32_t)len; if (len) memcpy(p, src, len); }
	void grow(size_t need)
	{
		const size_t nc = need > 2 * (size_t)cap + 64 ? need : 2 * (size_t)cap + 64;
		uint8_t *q = (uint8_t*)malloc(nc);
		if (!q) throw std::bad_alloc();
		if (n) memcpy(q, p, n);
		if (heap) free(p);
		p = q; cap = (uint32_t)nc; heap = true;
	}
	void resize(size_t m) { if (m > cap) grow(m); n = (uint32_t)m; }
	void append(const void *b, size_t len) { if ((size_t)n + len > cap) grow((size_t)n + len); memcpy(p + n, b, len); n += (uint32_t)len; }
};
#define REC_ROOM 160u              /* bytes of room behind a record for the tags and the CIGAR pass 2 adds */

struct BamRec {                    /* one record, parsed: offsets are into `data` (everything after the 32 bytes of core) */
	int32_t tid, pos; uint32_t bin, mapq, l_qname, flag, n_cigar; int32_t l_qseq, mtid, mpos, isize;
	RecBuf data;                   /* qname, cigar, seq, qual, tags */
	const uint8_t *rg_p; uint32_t rg_n;      /* its read group (bam_get_rg), a view into data: found while the record is being parsed */
	size_t off_cigar() const { return l_qname; }
	size_t off_seq() const { return l_qname + 4 * (size_t)n_cigar; }
	size_t off_qual() const { return off_seq() + ((size_t)l_qseq + 1) / 2; }
	size_t off_aux() const { return off_qual() + (size_t)l_qseq; }
};

static bool parse_rec(const uint8_t *p, int64_t len, BamRec &r, uint8_t *room)       /* room: len - 36 + REC_ROOM bytes of the arena */
{
	if (len < 36) return false;
	uint32_t bs; memcpy(&bs, p, 4);
	if ((int64_t)bs + 4 != len || bs < 32) return false;
	uint32_t y, z;
	memcpy(&r.tid, p + 4, 4); memcpy(&r.pos, p + 8, 4); memcpy(&y, p + 12, 4); memcpy(&z, p + 16, 4);
	memcpy(&r.l_qseq, p + 20, 4); memcpy(&r.mtid, p + 24, 4); memcpy(&r.mpos, p + 28, 4); memcpy(&r.isize, p + 32, 4);
	r.bin = y >> 16; r.mapq = y >> 8 & 0xff; r.l_qname = y & 0xff; r.flag = z >> 16; r.n_cigar = z & 0xffff;
	r.data.place(room, (size_t)(len - 36) + REC_ROOM, p + 36, (size_t)(len - 36));
	if (r.l_qseq < 0 || r.off_aux() > r.data.size() || r.l_qname == 0) return false;
	if (r.data.data()[r.l_qname - 1] != 0) return false;        /* the name is compared as a C string (mates, bwaseqio.c:366): it must end inside l_qname */
	return true;
}

/* erase_unwanted_tags (bwaseqio.c:413-464): AM NM CM SM MD X0 X1 XA XC XG XM XN XO XT YQ go, everything else stays */
static bool erase_tags(BamRec &r)
{
	size_t p = r.off_aux(), q = p; const size_t end = r.data.size();
	uint8_t *d = r.data.data();
	while (p < end) {
		if (p + 3 > end) return false;
		bool keep = true;
		switch (d[p]) {
			case 'A': case 'S': case 'C': case 'N': keep = d[p + 1] != 'M'; break;
			case 'M': keep = d[p + 1] != 'D'; break;
			case 'X': keep = !(d[p + 1] && strchr("01ACGMNOT", d[p + 1])); break;
			case 'Y': keep = d[p + 1] != 'Q'; break;
		}
		size_t len = 3;
		switch (d[p + 2] & ~32) {
			case 'C': case 'A': len += 1; break;
			case 'S': len += 2; break;
			case 'I': case 'F': len += 4; break;
			case 'D': len += 8; break;
			case 'Z': case 'H': while (p + len < end && d[p + len]) ++len; ++len; break;
			case 'B': {
				if (p + 8 > end) return false;
				const size_t count = (size_t)d[p + 4] | (size_t)d[p + 5] << 8 | (size_t)d[p + 6] << 16 | (size_t)d[p + 7] << 24;
				len += 5;
				switch (d[p + 3] & ~32) { case 'C': case 'A': len += count; break; case 'S': len += 2 * count; break;
										  case 'I': case 'F': len += 4 * count; break; case 'D': len += 8 * count; break; }
				break;
			}
		}
		if (p + len > end) return false;
		if (keep) { memmove(d + q, d + p, len); q += len; }
		p += len;
	}
	r.data.resize(q);
	return true;
}

/* bam_get_rg (bamlite.c:157-190): the read group of a record, "" when it has none */
static std::pair<const uint8_t*, size_t> get_rg(const BamRec &r)       /* a view into the record */
{
	size_t p = r.off_aux(); const size_t end = r.data.size(); const uint8_t *d = r.data.data();
	while (p + 4 < end) {
		if (d[p] == 'R' && d[p + 1] == 'G') {
			if (d[p + 2] == 'Z') return { d + p + 3, strnlen((const char*)d + p + 3, end - p - 3) };
			if (d[p + 2] == 'A') return { d + p + 3, (size_t)1 };
		}
		switch (d[p + 2]) {
			case 'A': case 'C': case 'c': p += 4; break;
			case 'S': case 's': p += 5; break;
			case 'I': case 'i': case 'f': p += 7; break;
			case 'd': p += 11; break;
			case 'Z': case 'H': p += 3; while (p < end && d[p]) ++p; ++p; break;
			case 'B': {
				if (p + 8 > end) return { d, (size_t)0 };
				const size_t count = (size_t)d[p + 4] | (size_t)d[p + 5] << 8 | (size_t)d[p + 6] << 16 | (size_t)d[p + 7] << 24;
				size_t w = 1; switch (d[p + 3]) { case 's': case 'S': w = 2; break; case 'i': case 'I': case 'f': w = 4; break; case 'd': w = 8; break; }
				p += 8 + w * count; break;
			}
			default: return { d, (size_t)0 };
		}
	}
	return { d, (size_t)0 };
}

static inline int nib4(uint8_t v) { return ((v & 1) << 3) | ((v & 2) << 1) | ((v & 4) >> 1) | ((v & 8) >> 3); }   /* complement of a 4-bit base code = its bits reversed */

/* revcom_bam1 (bam2bam.c:335-362): flip the strand flag, reverse-complement SEQ, reverse QUAL */
/* both nibbles of a byte complemented (nib4), in place and swapped */
static const struct NibComp { uint8_t same[256], swap[256]; NibComp() { for (int x = 0; x < 256; ++x) { const int hi = nib4((uint8_t)(x >> 4)), lo = nib4((uint8_t)(x & 15));
	same[x] = (uint8_t)(hi << 4 | lo); swap[x] = (uint8_t)(lo << 4 | hi); } } } nib_comp;

static void revcom_rec(BamRec &r)
{
	r.flag ^= F_SR;
	const int L = r.l_qseq;
	uint8_t *s = r.data.data() + r.off_seq(), *q = r.data.data() + r.off_qual();
	/* byte by byte: with an even number of bases the bytes change places and their nibbles with them; with an odd number every byte of the
	 * result is put together from two neighbours (base L-1 sits alone in the top of the last byte, and the new last byte ends in a zero nibble) */
	const int nb = (L + 1) / 2;
	if (!(L & 1)) {
		int a = 0, b = nb - 1;
		for (; a < b; ++a, --b) { const uint8_t x = nib_comp.swap[s[a]], y = nib_comp.swap[s[b]]; s[a] = y; s[b] = x; }
		if (a == b) s[a] = nib_comp.swap[s[a]];
	} else if (nb) {
		uint8_t small[256]; std::vector<uint8_t> big;
		uint8_t *c = small;
		if (nb > (int)sizeof(small)) { big.resize((size_t)nb); c = big.data(); }
		for (int j = 0; j < nb; ++j) c[j] = nib_comp.same[s[j]];
		const int m = nb - 1;
		for (int j = 0; j < m; ++j) s[j] = (uint8_t)((c[m - j] & 0xF0) | (c[m - j - 1] & 0x0F));
		s[m] = (uint8_t)(c[0] & 0xF0);
	}
	for (int a = 0, b = L - 1; a < b; ++a, --b) { const uint8_t t = q[a]; q[a] = q[b]; q[b] = t; }
}

static inline uint32_t reg2bin(uint32_t beg, uint32_t end)      /* bam_reg2bin (bam2bam.c:324-333) */
{
	--end;
	if (beg >> 14 == end >> 14) return 4681 + (beg >> 14);
	if (beg >> 17 == end >> 17) return 585 + (beg >> 17);
	if (beg >> 20 == end >> 20) return 73 + (beg >> 20);
	if (beg >> 23 == end >> 23) return 9 + (beg >> 23);
	if (beg >> 26 == end >> 26) return 1 + (beg >> 26);
	return 0;
}

static void push_int(BamRec &r, char u, char v, int x) { const uint8_t b[7] = { (uint8_t)u, (uint8_t)v, 'i', (uint8_t)x, (uint8_t)(x >> 8), (uint8_t)(x >> 16), (uint8_t)(x >> 24) }; r.data.append(b, 7); }
static void push_char(BamRec &r, char u, char v, char c) { const uint8_t b[4] = { (uint8_t)u, (uint8_t)v, 'A', (uint8_t)c }; r.data.append(b, 4); }
static void push_str(BamRec &r, char u, char v, const char *s) { const uint8_t b[3] = { (uint8_t)u, (uint8_t)v, 'Z' }; r.data.append(b, 3); r.data.append(s, strlen(s) + 1); }

static void set_cigar(BamRec &r, int n, const uint32_t *c)       /* bam_resize_cigar + the copy (bam2bam.c:411-420,467-477) */
{
	const size_t at = r.off_cigar(), old_b = 4 * (size_t)r.n_cigar, new_b = 4 * (size_t)n, tail = r.data.size() - at - old_b;
	if (new_b > old_b) { r.data.resize(r.data.size() + (new_b - old_b)); memmove(r.data.data() + at + new_b, r.data.data() + at + old_b, tail); }
	else if (new_b < old_b) { memmove(r.data.data() + at + new_b, r.data.data() + at + old_b, tail); r.data.resize(r.data.size() - (old_b - new_b)); }
	if (n) memcpy(r.data.data() + at, c, new_b);
	r.n_cigar = (uint32_t)n;
}

/* bwa_update_bam1 (bam2bam.c:430-593).  p: this end's finished record; mate: the other end's (null for a singleton); pe: this
 * end's pair fields.  p / mate are in the state nabwa_se_refine / nabwa_pe_finish leave them in, i.e. with the side effects the
 * reference's calls have on them (an unmapped end takes its mate's place, a contig-bridging hit loses its mapQ) already applied. */
static void update_bam(BamRec &out, const nabwa_reference *R, const nabwa_se_t &p, const nabwa_se_t *mate, const nabwa_pe_t *pe,
					   int mode, int max_top2, int yq)
{
	if (p.clip_len < p.full_len) push_int(out, 'X', 'C', p.clip_len);
	if (yq) push_int(out, 'Y', 'Q', yq);                  /* --debug-bam: the most entries the search held (bam2bam.c:433) */
	if (p.type != 0 || (mate && mate->type != 0)) {
		if ((p.strand != 0) != ((out.flag & F_SR) != 0)) revcom_rec(out);
		out.flag &= ~(uint32_t)(F_PP | F_SU | F_MU | F_SC | F_MR);
		const int fl = p.flag;                 /* what the chain derived: proper pair, self / mate unmapped, mate strand */
		out.flag |= (uint32_t)(fl & (F_PP | F_SU | F_MU | F_MR));
		const int seqid = p.seqid;
		const int64_t off = R->anns[seqid].offset;
		out.tid = seqid; out.pos = (int32_t)((int64_t)p.pos - off);
		out.bin = reg2bin((uint32_t)((int64_t)p.pos - off), (uint32_t)(rec_pos_end(p) - off));
		out.mapq = (uint32_t)p.mapQ & 0xff;
		if (p.n_cigar) {
			uint32_t c[NABWA_MAX_CIGAR];
			for (int j = 0; j < p.n_cigar; ++j) c[j] = (uint32_t)CLEN(p.cigar[j]) << 4 | (uint32_t)"\000\001\002\004"[COP(p.cigar[j])];
			set_cigar(out, p.n_cigar, c);
		} else if (p.type == 0) set_cigar(out, 0, 0);
		else { const uint32_t c = (uint32_t)p.len << 4; set_cigar(out, 1, &c); }
		if (mate && mate->type != 0) { out.mtid = pe->m_seqid; out.mpos = (int32_t)(pe->m_rpos - 1); out.isize = (int32_t)pe->isize; }
		else if (mate) { out.mtid = seqid; out.mpos = (int32_t)((int64_t)p.pos - off); out.isize = 0; }
		else { out.mtid = -1; out.mpos = -1; out.isize = 0; }
		if (p.type != 0) {
			push_char(out, 'X', 'T', p.xt);
			push_int(out, (mode & NABWA_MODE_COMPREAD) ? 'N' : 'C', 'M', p.nm);
			if (p.nn) push_int(out, 'X', 'N', p.nn);
			if (mate) { push_int(out, 'S', 'M', p.seQ); push_int(out, 'A', 'M', pe->am); }
			if (p.type != 3) {                                     /* X0 / X1 do not exist for a mate-rescued alignment */
				push_int(out, 'X', '0', (int)p.c1);
				if ((int64_t)p.c1 <= (int64_t)max_top2) push_int(out, 'X', '1', (int)p.c2);
			}
			push_int(out, 'X', 'M', p.n_mm); push_int(out, 'X', 'O', p.n_gapo); push_int(out, 'X', 'G', p.n_gapo + p.n_gape);
			push_str(out, 'M', 'D', p.md);
			if (p.n_multi) {
				std::string xa; char buf[128];
				for (int i = 0; i < p.n_multi; ++i) {
					const nabwa_multi_t &q = p.multi[i];
					int64_t e = q.pos;
					if (q.n_cigar) { for (int k = 0; k < q.n_cigar; ++k) { const int op = COP(q.cigar[k]); if (op == 0 || op == 2) e += CLEN(q.cigar[k]); } } else e += p.len;
					int sid; pac2real(R, q.pos, (int)(e - q.pos), &sid);
					snprintf(buf, sizeof buf, "%s,%c%d,", R->anns[sid].name.c_str(), q.strand ? '-' : '+', (int)((int64_t)q.pos - R->anns[sid].offset + 1)); xa += buf;
					if (q.n_cigar) for (int k = 0; k < q.n_cigar; ++k) { snprintf(buf, sizeof buf, "%d%c", CLEN(q.cigar[k]), "MIDS"[COP(q.cigar[k])]); xa += buf; }
					else { snprintf(buf, sizeof buf, "%dM", p.len); xa += buf; }
					snprintf(buf, sizeof buf, ",%d;", q.gap + q.mm); xa += buf;
				}
				push_str(out, 'X', 'A', xa.c_str());
			}
		}
	} else {                       /* neither this read nor its mate has a match */
		out.tid = -1; out.pos = -1; out.bin = 0; out.mapq = 0; out.mtid = -1; out.mpos = -1; out.isize = 0;
		out.flag &= ~(uint32_t)(F_PP | F_MU | F_SC);
		out.flag |= F_SU;
		if (mate && mate->type == 0) out.flag |= F_MU;
		set_cigar(out, 0, 0);
	}
}

static void write_rec(const BamRec &r, uint8_t *o)
{
	const uint32_t bs = 32 + (uint32_t)r.data.size();
	const uint32_t y = r.bin << 16 | (r.mapq & 0xff) << 8 | (r.l_qname & 0xff), z = r.flag << 16 | (r.n_cigar & 0xffff);
	uint8_t h[36];
	memcpy(h, &bs, 4); memcpy(h + 4, &r.tid, 4); memcpy(h + 8, &r.pos, 4); memcpy(h + 12, &y, 4); memcpy(h + 16, &z, 4);
	memcpy(h + 20, &r.l_qseq, 4); memcpy(h + 24, &r.mtid, 4); memcpy(h + 28, &r.mpos, 4); memcpy(h + 32, &r.isize, 4);
	memcpy(o, h, 36); if (!r.data.empty()) memcpy(o + 36, r.data.data(), r.data.size());
}

/* ------------------------------------------------------------------ the batch */


static void *res_take(size_t bytes);
static void res_give(void *p, size_t bytes);
struct RawBytes {          /* bytes without the zero fill of std::vector (100 MB per million reads, written once by many threads); large blocks come from
                            * and go back to the pool of per-batch blocks below: no page faults, no unmapping from batch to batch */
	uint8_t *p; size_t n, cap; bool pooled;
	RawBytes() : p(0), n(0), cap(0), pooled(false) {}
	~RawBytes() { drop(); }
	RawBytes(const RawBytes&) = delete;
	RawBytes &operator=(const RawBytes&) = delete;
	void drop() { if (p) { if (pooled) res_give(p, cap); else free(p); } p = 0; n = cap = 0; pooled = false; }
	bool alloc(size_t m)
	{
		drop();
		cap = m ? m : 1; pooled = cap >= ((size_t)1 << 20);
		p = (uint8_t*)(pooled ? res_take(cap) : malloc(cap));
		n = m;
		if (!p) { cap = 0; pooled = false; }
		return p != 0;
	}
	uint8_t *data() { return p; }
	const uint8_t *data() const { return p; }
	const uint8_t *begin() const { return p; }
};

/* The per-read records of a batch (3 KB each: they end in fixed CIGAR / MD / multi-hit arrays) are never filled whole, but fresh
 * memory costs a page fault per record (0.4 s per million).  A streaming caller makes one batch after the other: the buffer of a
 * destroyed batch is kept (up to 16 GB) and handed to the next one. */
static std::mutex g_res_mu;
static std::vector<std::pair<void*, size_t>> g_res_idle;
static void *res_take(size_t bytes)
{
	{
		std::lock_guard<std::mutex> lk(g_res_mu);
		for (size_t i = 0; i < g_res_idle.size(); ++i)
			if (g_res_idle[i].second >= bytes && g_res_idle[i].second <= 2 * bytes + (1u << 20)) { void *p = g_res_idle[i].first; g_res_idle.erase(g_res_idle.begin() + i); return p; }
	}
	/* large blocks on 2 MB boundaries with the huge-page advice: the passes touch a line or two of every 3 KB record, and with
	 * 4 KB pages nearly each of those touches was a TLB miss as well */
	if (bytes >= ((size_t)64 << 20)) {
		void *p = 0;
		const size_t al = (size_t)2 << 20, sz = (bytes + al - 1) / al * al;
		if (posix_memalign(&p, al, sz) == 0) { (void)madvise(p, sz, MADV_HUGEPAGE); return p; }
	}
	return malloc(bytes);
}
static void res_give(void *p, size_t bytes)
{
	if (!p) return;
	std::lock_guard<std::mutex> lk(g_res_mu);
	size_t tot = bytes;
	for (auto &x : g_res_idle) tot += x.second;
	if (tot > ((size_t)16 << 30) || g_res_idle.size() >= 24) { free(p); return; }      /* (a pipeline holds four batches: two 3 GB record blocks and a dozen smaller ones come and go) */
	g_res_idle.push_back({ p, bytes });
}

/* The parsed records of a batch: pooled memory like the other per-batch blocks, constructed and destroyed by all threads (a std::vector of
 * a million records does both on one thread, zero fill and page faults included: 15 ms of a 60 ms create). */
struct RecArr {
	BamRec *p; size_t n, bytes;
	RecArr() : p(0), n(0), bytes(0) {}
	~RecArr() { clear(); }
	RecArr(const RecArr&) = delete;
	RecArr &operator=(const RecArr&) = delete;
	bool make(size_t m)
	{
		clear();
		bytes = sizeof(BamRec) * (m ? m : 1);
		p = (BamRec*)res_take(bytes);
		if (!p) { bytes = 0; return false; }
		n = m;
		BamRec *const q = p;
		bam_parallel(m, [q](int, size_t lo, size_t hi) { for (size_t i = lo; i < hi; ++i) new (q + i) BamRec(); });
		return true;
	}
	void clear()
	{
		if (p) {
			BamRec *const q = p;
			bam_parallel(n, [q](int, size_t lo, size_t hi) { for (size_t i = lo; i < hi; ++i) q[i].~BamRec(); });
			res_give(p, bytes);
		}
		p = 0; n = 0; bytes = 0;
	}
	size_t size() const { return n; }
	bool empty() const { return n == 0; }
	BamRec &operator[](size_t i) { return p[i]; }
	const BamRec &operator[](size_t i) const { return p[i]; }
	void swap(RecArr &o) { std::swap(p, o.p); std::swap(n, o.n); std::swap(bytes, o.bytes); }
};

/* hit rows as they come back from the device: no zero fill on one thread in front of the copy (a std::vector's resize), pooled like the rest.
 * Growing it loses what it held (every caller fills it whole afterwards); shrinking keeps it. */
struct RowArr {
	RawBytes raw; size_t n;
	RowArr() : n(0) {}
	bool resize(size_t m) { if (m * sizeof(nabwa_aln1_t) > raw.cap) { if (!raw.alloc(m * sizeof(nabwa_aln1_t))) { n = 0; return false; } } n = m; return true; }
	size_t size() const { return n; }
	nabwa_aln1_t *data() { return (nabwa_aln1_t*)raw.p; }
	const nabwa_aln1_t *data() const { return (const nabwa_aln1_t*)raw.p; }
};

struct nabwa_bam_batch {
	nabwa_index *ix; nabwa_gap_opt_t opt; nabwa_pe_opt_t popt;
	uint8_t *arena; size_t arena_bytes;            /* where the records' bytes live (pooled like res); declared before rec: it outlives the records */
	RecArr rec;                                    /* in logical-record order: singletons, and pairs as read 1, read 2 */
	std::vector<int> kind;                         /* per logical record: 1 or 2 */
	std::vector<int> first;                        /* per logical record: index of its first read */
	std::vector<int> rg;                           /* per logical record: its read group, an index into rg_names */
	std::vector<std::string> rg_names;
	std::vector<uint8_t> skip;                     /* per logical record: a flagged duplicate that passes through untouched (--skip-duplicates; unique(), bam2bam.c:595-606) */
	uint32_t flags;                                /* NABWA_BAM_* */
	std::vector<int64_t> off; RawBytes seq, rseq; std::vector<int32_t> full_len;     /* the encoded reads, one per BAM record */
	std::vector<int32_t> n_aln, max_ent; RowArr rows; std::vector<int64_t> row0;
	nabwa_pe_t *res;                               /* per read: the chain's record (singletons use .se only); raw memory: only what a phase fills is valid */
	int phase;                                     /* 0 created, 1 positioned, 2 finished */
	bool searched;                                 /* nabwa_bam_batch_search ran */
	std::vector<uint8_t> parked; std::vector<uint64_t> parked_at;     /* what pass 1 left in res, packed, while a batch with pairs waits for pass 2 */
	std::vector<uint8_t> wire_multi;                /* nabwa_bam_batch_positioned: the other hits of the reads as raw bwt_multi1_t */
	size_t res_bytes;
	nabwa_bam_batch() : arena(0), arena_bytes(0), flags(0), res(0), phase(0), searched(false), res_bytes(0) {}
	~nabwa_bam_batch() { res_give(res, res_bytes); rec.clear(); res_give(arena, arena_bytes); }
};

static const uint8_t nt16_nt4[16] = { 4, 0, 1, 4, 2, 4, 4, 4, 3, 4, 4, 4, 4, 4, 4, 4 };      /* bam_nt16_nt4_table (bwaseqio.c:10) */
/* the two bases of a byte at once, as they lie in a REVERSED read (the later base first), plain and complemented: one 16-bit store per strand for two bases */
static const struct Nt16Rev { uint16_t s[256], r[256]; Nt16Rev() { for (int x = 0; x < 256; ++x) { const uint8_t a = nt16_nt4[x >> 4], b = nt16_nt4[x & 15];
	s[x] = (uint16_t)(b | a << 8); r[x] = (uint16_t)((b < 4 ? 3 - b : b) | (a < 4 ? 3 - a : a) << 8); } } } nt16_rev;

extern "C" int nabwa_bam_batch_create(nabwa_index_t *ix, const nabwa_gap_opt_t *opt, const nabwa_pe_opt_t *popt, int n_rec,
									  const uint8_t *in, const int64_t *in_off, nabwa_bam_batch_t **out)
{
	return nabwa_bam_batch_create_ex(ix, opt, popt, 0, n_rec, in, in_off, out);
}

extern "C" int nabwa_bam_batch_create_ex(nabwa_index_t *ix, const nabwa_gap_opt_t *opt, const nabwa_pe_opt_t *popt, uint32_t flags, int n_rec,
										 const uint8_t *in, const int64_t *in_off, nabwa_bam_batch_t **out)
{
	if (!ix || !opt || !popt || !out || n_rec < 0 || (n_rec && (!in || !in_off))) return nabwa_fail(NABWA_EINVAL, "null argument");
	if (flags & ~(uint32_t)NABWA_BAM_ALL_FLAGS) return nabwa_fail(NABWA_EINVAL, "unknown NABWA_BAM_* flag");
	if (!ix->ref) return nabwa_fail(NABWA_EINVAL, "index has no reference attached (nabwa_index_attach_reference)");
	/* limits of the record format, refused here and not after the search: the other hits of a singleton (bam2bam.c:629) fill a fixed list */
	if (popt->max_occ_se < 0 || popt->max_occ_se + 1 > NABWA_MAX_MULTI) return nabwa_fail(NABWA_EINVAL, "max_occ_se (-D) outside 0..15");
	if (popt->n_multi < 0 || popt->N_multi < 0 || popt->n_multi > NABWA_MAX_MULTI || popt->N_multi > NABWA_MAX_MULTI) return nabwa_fail(NABWA_EINVAL, "n_multi / N_multi outside 0..16");
	if (opt->s_mm < 1 || opt->s_gapo < 1 || opt->s_gape < 1) return nabwa_fail(NABWA_EINVAL, "s_mm, s_gapo and s_gape must be >= 1 (-M / -O / -E 0 are not supported)");
	nabwa_bam_batch *b = new nabwa_bam_batch();
	b->ix = ix; b->opt = *opt; b->popt = *popt; b->phase = 0; b->flags = flags;
	const bool timing = getenv("NABWA_TIMING") != 0;
	const double tc0 = bam_now();
	std::vector<uint32_t> flag_or;
	if (!b->rec.make((size_t)n_rec)) { delete b; return nabwa_fail(NABWA_ENOMEM, "out of memory for the records"); }
	{
		for (int i = 0; i < n_rec; ++i) if (in_off[i + 1] - in_off[i] < 36 || in_off[i + 1] - in_off[i] > (int64_t)1 << 28) { delete b; return nabwa_fail(NABWA_EINVAL, "malformed BAM record"); }
		b->arena_bytes = (size_t)(n_rec ? in_off[n_rec] - in_off[0] : 0) + (size_t)n_rec * (REC_ROOM - 36) + 64;
		b->arena = (uint8_t*)res_take(b->arena_bytes);
		if (!b->arena) { b->arena_bytes = 0; delete b; return nabwa_fail(NABWA_ENOMEM, "out of memory for the records"); }
		std::vector<int> bad(bam_threads((size_t)n_rec), 0);
		flag_or.assign(bam_threads((size_t)n_rec) * 16, 0u);          /* (a line per thread) */
		bam_parallel((size_t)n_rec, [&](int t, size_t lo, size_t hi) {
			uint32_t fo = 0;
			for (size_t i = lo; i < hi; ++i)
			{
				/* one pass over a record while it is in the cache: parse, erase_unwanted_tags, bam_get_rg (neither depends on how the records
				 * pair up; a record the pairing drops has been cleaned in vain) */
				BamRec &r = b->rec[i];
				if (!parse_rec(in + in_off[i], in_off[i + 1] - in_off[i], r, b->arena + (in_off[i] - in_off[0]) + i * (size_t)(REC_ROOM - 36))) { bad[t] = 1; continue; }
				if (!erase_tags(r)) { bad[t] = 2; continue; }
				const auto v = get_rg(r);
				r.rg_p = v.first; r.rg_n = (uint32_t)v.second;
				fo |= r.flag;
			}
			flag_or[(size_t)t * 16] = fo;
		});
		for (int x : bad) if (x) { delete b; return nabwa_fail(NABWA_EINVAL, x == 2 ? "malformed tags in a BAM record" : "malformed BAM record"); }
	}
	const double tc1 = bam_now();
	/* logical records (read_bam_pair_core, bwaseqio.c:346-410): a paired read takes the next record as its mate -- same name,
	 * flags read 1 / read 2 in either order.  Anything else is an error, or with NABWA_BAM_BROKEN_INPUT (allow_broken) is mended as
	 * the reference mends it: wrong flags are set right, a paired read whose successor has another name is discarded and that
	 * successor starts the next logical record, a paired read with nothing after it is discarded.  NABWA_BAM_DROP_ALIGNED
	 * (read_bam_pair's ignore_aligned, bwaseqio.c:466-474) leaves out logical records any read of which is already mapped. */
	{
		const bool broken = (flags & NABWA_BAM_BROKEN_INPUT) != 0, drop = (flags & NABWA_BAM_DROP_ALIGNED) != 0, nodup = (flags & NABWA_BAM_SKIP_DUPLICATES) != 0;
		uint32_t any_flag = 0;
		for (size_t t = 0; t < flag_or.size(); t += 16) any_flag |= flag_or[t];
		std::vector<int> src;
		if (!(any_flag & F_PD) && !drop && !nodup) {
			/* single-end records only and nothing to leave out: every record is a logical record of its own */
			b->kind.assign((size_t)n_rec, 1); b->skip.assign((size_t)n_rec, 0); b->first.resize((size_t)n_rec);
			int *const fp = b->first.data();
			bam_parallel((size_t)n_rec, [fp](int, size_t lo, size_t hi) { for (size_t i = lo; i < hi; ++i) fp[i] = (int)i; });
			src.resize((size_t)n_rec);      /* (only its size is looked at) */
		} else {
		src.reserve(n_rec); b->kind.reserve(n_rec); b->first.reserve(n_rec); b->skip.reserve(n_rec);
		for (int i = 0; i < n_rec; ) {
			BamRec &r0 = b->rec[i];
			int k = 1;
			if (r0.flag & F_PD) {
				if (i + 1 >= n_rec) {
					if (broken) break;
					delete b; return nabwa_fail(NABWA_EINVAL, "a paired read at the end of the batch without its mate (keep mates in one batch)");
				}
				BamRec &r1 = b->rec[i + 1];
				const uint32_t f0 = r0.flag & (F_PD | F_R1 | F_R2), f1 = r1.flag & (F_PD | F_R1 | F_R2);
				if (strcmp((const char*)r0.data.data(), (const char*)r1.data.data()) != 0) {
					if (broken) { ++i; continue; }
					delete b; return nabwa_fail(NABWA_EINVAL, "lone mate: two paired reads whose names do not match");
				}
				if (f0 == (F_PD | F_R2) && f1 == (F_PD | F_R1)) std::swap(r0, r1);
				else if (!(f0 == (F_PD | F_R1) && f1 == (F_PD | F_R2))) {
					if (!broken) { delete b; return nabwa_fail(NABWA_EINVAL, "a pair whose read 1 / read 2 flags are wrong"); }
					r0.flag = (r0.flag & ~(uint32_t)F_R2) | F_PD | F_R1; r1.flag = (r1.flag & ~(uint32_t)F_R1) | F_PD | F_R2;
				}
				k = 2;
			}
			const uint32_t all = r0.flag & (k == 2 ? b->rec[i + 1].flag : ~0u), any = r0.flag | (k == 2 ? b->rec[i + 1].flag : 0u);
			if (!(drop && !(all & F_SU))) {
				if (k == 2) { BamRec &r1 = b->rec[i + 1]; r0.flag |= r1.flag & F_QC; r1.flag |= r0.flag & F_QC; }          /* either none or both pass QC (bwaseqio.c:486-489) */
				b->kind.push_back(k); b->first.push_back((int)src.size()); b->skip.push_back(nodup && (any & F_DP));
				for (int e = 0; e < k; ++e) src.push_back(i + e);
			}
			i += k;
		}
		}
		if ((int)src.size() != n_rec) {
			RecArr kept;
			if (!kept.make(src.size())) { delete b; return nabwa_fail(NABWA_ENOMEM, "out of memory for the records"); }
			for (size_t t = 0; t < src.size(); ++t) kept[t] = std::move(b->rec[src[t]]);
			b->rec.swap(kept);
			n_rec = (int)src.size();
		}
	}
	const double tc1a = bam_now();
	const double tc1b = bam_now();
	{
		const size_t nk = b->kind.size();
		b->rg.resize(nk);
		std::map<std::string, int> ids;
		/* "the same read group as the logical record before" by all threads (the records' bytes are touched there); the names that change, in order, by one */
		std::vector<uint8_t> same_rg(nk ? nk : 1, 0);
		bam_parallel(nk, [&](int, size_t lo, size_t hi) {
			for (size_t k = lo ? lo : 1; k < hi; ++k) {
				const BamRec &r0 = b->rec[b->first[k]], &rp = b->rec[b->first[k - 1]];
				same_rg[k] = r0.rg_n == rp.rg_n && !memcmp(r0.rg_p, rp.rg_p, r0.rg_n);
			}
		});
		for (size_t k = 0; k < nk; ++k) {
			if (same_rg[k]) { b->rg[k] = b->rg[k - 1]; continue; }
			const BamRec &r0 = b->rec[b->first[k]];
			const uint8_t *vp = r0.rg_p; const uint32_t vn = r0.rg_n;
			auto ins = ids.emplace(std::string((const char*)vp, vn), (int)b->rg_names.size());
			if (ins.second) b->rg_names.push_back(ins.first->first);
			b->rg[k] = ins.first->second;
		}
	}
	const double tc2 = bam_now();
	/* bam1_to_seq (bwaseqio.c:272-307): the (trimmed) lengths first, then every thread encodes its slice of the reads in place */
	b->off.assign(n_rec + 1, 0); b->full_len.assign(n_rec ? n_rec : 1, 0);
	{
		std::vector<int32_t> lens(n_rec ? n_rec : 1, 0);
		std::vector<uint8_t> rskip(n_rec ? n_rec : 1, 0);       /* a duplicate that is passed through is searched as a read without bases */
		for (size_t k = 0; k < b->kind.size(); ++k) if (b->skip[k]) for (int e = 0; e < b->kind[k]; ++e) rskip[b->first[k] + e] = 1;
		bam_parallel((size_t)n_rec, [&](int, size_t lo, size_t hi) {
			for (size_t i = lo; i < hi; ++i) {
				const BamRec &x = b->rec[i];
				const int L = x.l_qseq;
				int len = L;
				if (opt->trim_qual >= 1) {                    /* bwa_trim_read (bwaseqio.c:110-123) on phred + 33 capped at 126, in the read's own orientation */
					const bool rev = (x.flag & F_SR) != 0;
					const uint8_t *ql = x.data.data() + x.off_qual();
					int sc = 0, mx = 0, max_l = L - 1;
					for (int l = L - 1; l >= 35 - 1; --l) {
						const int jj = rev ? L - 1 - l : l; const int q = ql[jj] + 33 < 126 ? ql[jj] : 93;
						sc += opt->trim_qual - q;
						if (sc < 0) break;
						if (sc > mx) { mx = sc; max_l = l; }
					}
					len = max_l + 1;
				}
				lens[i] = rskip[i] ? 0 : len; b->full_len[i] = L;
			}
		});
		for (int i = 0; i < n_rec; ++i) b->off[i + 1] = b->off[i] + lens[i];
		if (!b->seq.alloc((size_t)b->off[n_rec] + 1) || !b->rseq.alloc((size_t)b->off[n_rec] + 1)) { delete b; return nabwa_fail(NABWA_ENOMEM, "out of memory for the reads"); }
		bam_parallel((size_t)n_rec, [&](int, size_t lo, size_t hi) {
			for (size_t i = lo; i < hi; ++i) {
				const BamRec &x = b->rec[i];
				const int L = x.l_qseq, len = lens[i];
				const bool rev = (x.flag & F_SR) != 0;
				const uint8_t *sq = x.data.data() + x.off_seq();
				uint8_t *s = b->seq.data() + b->off[i], *r = b->rseq.data() + b->off[i];
				/* base j of the read in its own orientation: a record that carries the reverse flag holds the reverse complement
				 * (bwaseqio.c:288-291); seq = the (trimmed) read reversed, rseq = its complement (bwaseqio.c:294-297) */
				if (!rev) {
					/* s[j] = code of base len-1-j; a byte of the record holds bases 2m (high nibble) and 2m+1: both codes from one table look-up,
					 * written back to front */
					int k = 0;
					for (; k + 1 < len; k += 2) {          /* bases k, k + 1 go to places len-1-k, len-2-k: the two bytes at len-2-k, the later base first */
						const uint8_t x = sq[k >> 1];
						memcpy(s + (len - 2 - k), &nt16_rev.s[x], 2);
						memcpy(r + (len - 2 - k), &nt16_rev.r[x], 2);
					}
					if (k < len) { const uint8_t v = nt16_nt4[sq[k >> 1] >> 4]; s[len - 1 - k] = v; r[len - 1 - k] = v < 4 ? 3 - v : v; }
				} else for (int j = 0; j < len; ++j) {
					const int k = len - 1 - j, jj = L - 1 - k;
					uint8_t v = nt16_nt4[sq[jj >> 1] >> ((~jj & 1) << 2) & 15];
					if (v < 4) v = 3 - v;
					s[j] = v; r[j] = v < 4 ? 3 - v : v;
				}
			}
		});
		b->seq.data()[b->off[n_rec]] = 0; b->rseq.data()[b->off[n_rec]] = 0;
	}
	if (timing) fprintf(stderr, "[nabwa] bam_batch_create %d records: parse %.3f s, pairing %.3f s, tag erase %.3f s, read groups %.3f s, bam1_to_seq %.3f s (%d threads)\n",
						n_rec, tc1 - tc0, tc1a - tc1, tc1b - tc1a, tc2 - tc1b, bam_now() - tc2, bam_threads((size_t)n_rec));
	*out = b;
	return NABWA_OK;
}

extern "C" void nabwa_bam_batch_destroy(nabwa_bam_batch_t *b) { delete b; }

/* A batch with pairs waits between the passes for the insert-size estimates of the whole input, and of its 3 KB per read pass 1
 * has filled some 70 bytes (the scalar head and, for singletons, the heads of the other hits).  These are packed and the block
 * goes back to the pool until pass 2 asks for it again: a file of 20 M paired reads waits in 1.4 GB instead of 64 GB. */
#define PARK_HEAD offsetof(nabwa_se_t, cigar)
#define PARK_MULTI offsetof(nabwa_multi_t, cigar)
static void park(nabwa_bam_batch *b)
{
	const size_t n = b->rec.size();
	b->parked_at.assign(n + 1, 0);
	for (size_t i = 0; i < n; ++i) b->parked_at[i + 1] = b->parked_at[i] + PARK_HEAD + 4 + PARK_MULTI * (size_t)b->res[i].se.n_multi;
	b->parked.resize(b->parked_at[n] ? b->parked_at[n] : 1);
	bam_parallel(n, [&](int, size_t lo, size_t hi) {
		for (size_t i = lo; i < hi; ++i) {
			const nabwa_se_t &s = b->res[i].se;
			uint8_t *o = b->parked.data() + b->parked_at[i];
			memcpy(o, &s, PARK_HEAD); memcpy(o + PARK_HEAD, &s.n_multi, 4);
			for (int j = 0; j < s.n_multi; ++j) memcpy(o + PARK_HEAD + 4 + PARK_MULTI * (size_t)j, &s.multi[j], PARK_MULTI);
		}
	});
	res_give(b->res, b->res_bytes);
	b->res = 0;
}
static bool unpark(nabwa_bam_batch *b)
{
	const size_t n = b->rec.size();
	b->res = (nabwa_pe_t*)res_take(b->res_bytes);
	if (!b->res) return false;
	bam_parallel(n, [&](int, size_t lo, size_t hi) {
		for (size_t i = lo; i < hi; ++i) {
			nabwa_pe_t &r = b->res[i]; nabwa_se_t &s = r.se;
			const uint8_t *o = b->parked.data() + b->parked_at[i];
			memcpy(&s, o, PARK_HEAD); memcpy(&s.n_multi, o + PARK_HEAD, 4);
			for (int j = 0; j < s.n_multi; ++j) memcpy(&s.multi[j], o + PARK_HEAD + 4 + PARK_MULTI * (size_t)j, PARK_MULTI);
			s.nm = 0; s.md[0] = 0; s.flag = 0; s.seqid = 0; s.nn = 0; s.rpos = 0; s.xt = 0;          /* as nabwa_se_posn leaves them */
			r.extra_flag = 0; r.m_seqid = 0; r.am = 0; r.mapQ_paired = 0; r.m_rpos = 0; r.isize = 0;
		}
	});
	std::vector<uint8_t>().swap(b->parked); std::vector<uint64_t>().swap(b->parked_at);
	return true;
}

/* the FM search of the batch's reads: the part of pass 1 that needs neither the random stream nor the other batches, so a caller
 * may run it ahead, from another thread and on the batch's own GPU, while pass 1 and pass 2 of earlier batches go on
 * (bwa_cal_sa_reg_gap is called with one read at a time in the reference, bam2bam.c:616,676 -> per_read = 1) */
extern "C" int nabwa_bam_batch_search(nabwa_bam_batch_t *b)
{
	if (!b) return nabwa_fail(NABWA_EINVAL, "null argument");
	if (b->phase != 0 || b->searched) return nabwa_fail(NABWA_EINVAL, "the batch has been searched already");
	const int n = (int)b->rec.size();
	b->n_aln.assign(n ? n : 1, 0); b->max_ent.assign(n ? n : 1, 0); b->row0.assign(n + 1, 0);
	int64_t n_rows = 0;
	nabwa_batch_t *sb = 0;
	int rc = nabwa_batch_create(b->ix, &b->opt, n, b->off.data(), b->seq.data(), b->rseq.data(), 1, &sb);
	if (rc != NABWA_OK) return rc;
	rc = nabwa_batch_run(sb);
	if (rc == NABWA_OK) rc = nabwa_batch_sync(sb, 0);
	if (rc == NABWA_OK) {
		/* one fetch where the rows fit a guess (most reads bring one row), a second one where they do not */
		if (!b->rows.resize((size_t)n + (size_t)n / 4 + 1024)) { nabwa_batch_destroy(sb); return nabwa_fail(NABWA_ENOMEM, "out of memory for the hit rows"); }
		rc = nabwa_batch_fetch(sb, b->n_aln.data(), b->rows.data(), (int64_t)b->rows.size(), &n_rows, b->max_ent.data());
		if (rc == NABWA_ECAP && n_rows > (int64_t)b->rows.size()) {
			if (!b->rows.resize((size_t)n_rows)) { nabwa_batch_destroy(sb); return nabwa_fail(NABWA_ENOMEM, "out of memory for the hit rows"); }
			rc = nabwa_batch_fetch(sb, b->n_aln.data(), b->rows.data(), n_rows, &n_rows, b->max_ent.data());
		}
		if (rc == NABWA_OK) (void)b->rows.resize(n_rows ? (size_t)n_rows : 1);
	}
	nabwa_batch_destroy(sb);
	if (rc == NABWA_OK) b->searched = true;
	return rc;
}

/* pass 1: pair_aln + pair_posn + improve_isize_est of every logical record (bam2bam.c:1143-1176) */
extern "C" int nabwa_bam_batch_pass1(nabwa_bam_batch_t *b, uint64_t *rng48, nabwa_isize_table_t *tab)
{
	if (!b || !rng48 || !tab) return nabwa_fail(NABWA_EINVAL, "null argument");
	if (b->phase != 0) return nabwa_fail(NABWA_EINVAL, "pass 1 already ran on this batch");
	const int n = (int)b->rec.size();
	const bool timing = getenv("NABWA_TIMING") != 0;
	const double tp0 = bam_now();
	int rc = b->searched ? NABWA_OK : nabwa_bam_batch_search(b);
	if (rc != NABWA_OK) return rc;
	for (int i = 0; i < n; ++i) b->row0[i + 1] = b->row0[i] + b->n_aln[i];
	for (size_t k = 0; k < b->kind.size(); ++k) if (b->skip[k]) for (int e = 0; e < b->kind[k]; ++e)
		if (b->n_aln[b->first[k] + e]) return nabwa_fail(NABWA_EINVAL, "internal: a read without bases came back with hits");
	/* posn_singleton / posn_pair in record order: singletons list up to max_occ_se other hits, ends of pairs none */
	std::vector<uint8_t> n_occ(n ? n : 1, 0);
	for (size_t k = 0; k < b->kind.size(); ++k) if (b->kind[k] == 1) n_occ[b->first[k]] = (uint8_t)b->popt.max_occ_se;
	res_give(b->res, b->res_bytes);
	b->res_bytes = sizeof(nabwa_pe_t) * (size_t)(n ? n : 1);
	b->res = (nabwa_pe_t*)res_take(b->res_bytes);
	if (!b->res) return nabwa_fail(NABWA_ENOMEM, "out of memory for the batch's records");
	const double tp1 = bam_now();
	rc = nabwa_se_posn_strided(b->ix, &b->opt, n, b->off.data(), b->full_len.data(), b->n_aln.data(), b->rows.data(), n_occ.data(), rng48, b->res, sizeof(nabwa_pe_t));
	if (rc != NABWA_OK) return rc;
	bam_parallel((size_t)n, [&](int, size_t lo, size_t hi) {
		for (size_t i = lo; i < hi; ++i) { nabwa_pe_t &r = b->res[i]; r.extra_flag = 0; r.m_seqid = 0; r.am = 0; r.mapQ_paired = 0; r.m_rpos = 0; r.isize = 0; }
	});
	/* improve_isize_est (insert_size.c:141-165): the bins by many threads, the counts in record order */
	{
		const size_t nk = b->kind.size();
		std::vector<int> bin(nk ? nk : 1, -1);
		bam_parallel(nk, [&](int, size_t lo, size_t hi) {
			for (size_t k = lo; k < hi; ++k) {
				if (b->skip[k]) continue;
				const int i = b->first[k];
				const nabwa_se_t &s0 = b->res[i].se;
				const nabwa_se_t &s1 = b->kind[k] == 2 ? b->res[i + 1].se : s0;
				bin[k] = nabwa_isize_bin(b->kind[k], s0.mapQ, s1.mapQ, s0.pos, s0.len, s1.pos, s1.len);
			}
		});
		std::vector<nabwa_isize_table::Rg*> slot(b->rg_names.size(), (nabwa_isize_table::Rg*)0);
		for (size_t k = 0; k < nk; ++k) {
			if (bin[k] < 0) continue;
			nabwa_isize_table::Rg *&r = slot[b->rg[k]];
			if (!r) r = isize_slot(tab, b->rg_names[b->rg[k]]);
			if (r->has_hist) r->hist[bin[k]] = (uint16_t)(r->hist[bin[k]] + 1);
		}
	}
	if (b->kind.size() != b->rec.size()) park(b);
	b->phase = 1;
	if (timing) fprintf(stderr, "[nabwa] bam_batch_pass1 %d records: search (upload, kernels, rows back) %.3f s, posn + insert-size bins %.3f s\n", n, tp1 - tp0, bam_now() - tp1);
	return NABWA_OK;
}

/* ---- the state pass 1 leaves, out of the batch and back into a fresh one (the wire record's positioned / aligned parts) */
extern "C" int nabwa_bam_batch_positioned(nabwa_bam_batch_t *b, nabwa_wire_read_t *out)
{
	if (!b || !out) return nabwa_fail(NABWA_EINVAL, "null argument");
	if (b->phase != 1) return nabwa_fail(NABWA_EINVAL, "the batch is not between the passes");
	if (!b->res && !unpark(b)) return nabwa_fail(NABWA_ENOMEM, "out of memory for the batch's records");
	const size_t n = b->rec.size();
	std::vector<size_t> m0(n + 1, 0);
	for (size_t i = 0; i < n; ++i) m0[i + 1] = m0[i] + (size_t)b->res[i].se.n_multi;
	b->wire_multi.assign(16 * (m0[n] ? m0[n] : 1), 0);
	for (size_t i = 0; i < n; ++i) {
		const nabwa_se_t &s = b->res[i].se; nabwa_wire_read_t &w = out[i];
		w.strand = (uint8_t)s.strand; w.type = (uint8_t)s.type; w.n_mm = (uint8_t)s.n_mm; w.n_gapo = (uint8_t)s.n_gapo; w.n_gape = (uint8_t)s.n_gape;
		w.seQ = (uint8_t)s.seQ; w.mapQ = (uint8_t)s.mapQ; w.len = s.len; w.clip_len = s.clip_len; w.score = s.score; w.sa = s.sa; w.c1 = s.c1; w.c2 = s.c2; w.pos = s.pos;
		w.n_multi = s.n_multi; w.multi = b->wire_multi.data() + 16 * m0[i];
		for (int j = 0; j < s.n_multi; ++j) {                     /* bwt_multi1_t (bwtaln.h:58-62): pos, n_cigar:15 | gap:8 | mm:8 | strand:1, a pointer that means nothing outside its process */
			uint8_t *o = b->wire_multi.data() + 16 * (m0[i] + (size_t)j);
			const uint32_t pos = s.multi[j].pos, bits = ((uint32_t)s.multi[j].gap & 0xff) << 15 | ((uint32_t)s.multi[j].mm & 0xff) << 23 | ((uint32_t)s.multi[j].strand & 1) << 31;
			memcpy(o, &pos, 4); memcpy(o + 4, &bits, 4);
		}
		w.max_entries = b->max_ent[i]; w.n_aln = b->n_aln[i]; w.aln = (const uint8_t*)(b->rows.data() + b->row0[i]);
	}
	return NABWA_OK;
}

extern "C" int nabwa_bam_batch_restore(nabwa_bam_batch_t *b, const nabwa_wire_read_t *in)
{
	if (!b || !in) return nabwa_fail(NABWA_EINVAL, "null argument");
	if (b->phase != 0 || b->searched) return nabwa_fail(NABWA_EINVAL, "restore needs a batch that was just created");
	const int n = (int)b->rec.size();
	b->n_aln.assign(n ? n : 1, 0); b->max_ent.assign(n ? n : 1, 0); b->row0.assign(n + 1, 0);
	for (int i = 0; i < n; ++i) {
		const nabwa_wire_read_t &w = in[i];
		if (w.n_aln < 0 || w.n_multi < 0 || w.n_multi > NABWA_MAX_MULTI || (w.n_aln && !w.aln) || (w.n_multi && !w.multi)) { char m[96]; snprintf(m, sizeof m, "record %d: counts out of range", i); return nabwa_fail(NABWA_EINVAL, "%s", m); }
		if ((int64_t)w.len != b->off[i + 1] - b->off[i]) { char m[160]; snprintf(m, sizeof m, "record %d: positioned with a length of %d, the batch has %lld (other trimming options?)", i, w.len, (long long)(b->off[i + 1] - b->off[i])); return nabwa_fail(NABWA_EINVAL, "%s", m); }
		b->n_aln[i] = w.n_aln; b->max_ent[i] = w.max_entries; b->row0[i + 1] = b->row0[i] + w.n_aln;
	}
	if (!b->rows.resize(b->row0[n] ? (size_t)b->row0[n] : 1)) return nabwa_fail(NABWA_ENOMEM, "out of memory for the hit rows");
	res_give(b->res, b->res_bytes);
	b->res_bytes = sizeof(nabwa_pe_t) * (size_t)(n ? n : 1);
	b->res = (nabwa_pe_t*)res_take(b->res_bytes);
	if (!b->res) return nabwa_fail(NABWA_ENOMEM, "out of memory for the batch's records");
	bam_parallel((size_t)n, [&](int, size_t lo, size_t hi) {
		for (size_t i = lo; i < hi; ++i) {
			const nabwa_wire_read_t &w = in[i]; nabwa_pe_t &r = b->res[i]; nabwa_se_t &s = r.se;
			if (w.n_aln) memcpy(b->rows.data() + b->row0[i], w.aln, 16 * (size_t)w.n_aln);
			memset(&s, 0, offsetof(nabwa_se_t, cigar));                            /* as nabwa_se_posn leaves a record */
			s.n_cigar = 0; s.nm = 0; s.md[0] = 0; s.flag = 0; s.seqid = 0; s.nn = 0; s.rpos = 0; s.xt = 0;
			s.type = w.type & 3; s.strand = w.strand & 1; s.n_mm = w.n_mm; s.n_gapo = w.n_gapo; s.n_gape = w.n_gape; s.score = w.score; s.sa = w.sa; s.pos = w.pos;
			s.c1 = w.c1 & 0xfffffff; s.c2 = w.c2 & 0xfffffff; s.mapQ = w.mapQ; s.seQ = w.seQ; s.len = w.len; s.clip_len = w.clip_len; s.full_len = b->full_len[i];
			s.n_multi = w.n_multi;
			for (int j = 0; j < w.n_multi; ++j) {
				uint32_t pos, bits; memcpy(&pos, w.multi + 16 * (size_t)j, 4); memcpy(&bits, w.multi + 16 * (size_t)j + 4, 4);
				s.multi[j].pos = pos; s.multi[j].gap = bits >> 15 & 0xff; s.multi[j].mm = bits >> 23 & 0xff; s.multi[j].strand = bits >> 31; s.multi[j].n_cigar = 0;
			}
			r.extra_flag = 0; r.m_seqid = 0; r.am = 0; r.mapQ_paired = 0; r.m_rpos = 0; r.isize = 0;
		}
	});
	b->searched = true; b->phase = 1;
	return NABWA_OK;
}

/* one working record to another: the scalars, and of the arrays what their counts say is filled */
static void copy_filled(nabwa_pe_t &d, const nabwa_pe_t &r)
{
	const nabwa_se_t &s = r.se; nabwa_se_t &o = d.se;
	memcpy(&o, &s, offsetof(nabwa_se_t, cigar));
	if (s.n_cigar > 0) memcpy(o.cigar, s.cigar, sizeof(uint16_t) * (size_t)(s.n_cigar < NABWA_MAX_CIGAR ? s.n_cigar : NABWA_MAX_CIGAR));
	o.nm = s.nm;
	memcpy(o.md, s.md, strnlen(s.md, NABWA_MAX_MD - 1) + 1);
	o.n_multi = s.n_multi;
	for (int j = 0; j < s.n_multi && j < NABWA_MAX_MULTI; ++j) {
		memcpy(&o.multi[j], &s.multi[j], offsetof(nabwa_multi_t, cigar));
		if (s.multi[j].n_cigar > 0) memcpy(o.multi[j].cigar, s.multi[j].cigar, sizeof(uint16_t) * (size_t)(s.multi[j].n_cigar < NABWA_MAX_CIGAR ? s.multi[j].n_cigar : NABWA_MAX_CIGAR));
	}
	o.flag = s.flag; o.seqid = s.seqid; o.nn = s.nn; o.rpos = s.rpos; o.xt = s.xt;
	d.extra_flag = r.extra_flag; d.m_seqid = r.m_seqid; d.am = r.am; d.mapQ_paired = r.mapQ_paired; d.m_rpos = r.m_rpos; d.isize = r.isize;
}

/* pass 2: pair_finish of every logical record (bam2bam.c:1178-1216, 643-658, 705-811) */
extern "C" int nabwa_bam_batch_pass2(nabwa_bam_batch_t *b, const nabwa_isize_table_t *tab, uint64_t n_tot[2], uint64_t n_mapped[2])
{
	if (!b || !tab) return nabwa_fail(NABWA_EINVAL, "null argument");
	if (b->phase != 1) return nabwa_fail(NABWA_EINVAL, "pass 2 needs a batch that went through pass 1 once");
	const nabwa_reference *R = b->ix->ref;
	const bool timing = getenv("NABWA_TIMING") != 0;
	const double tq0 = bam_now();
	if (!b->res && !unpark(b)) return nabwa_fail(NABWA_ENOMEM, "out of memory for the batch's records");
	/* ---- singletons: bwa_refine_gapped + what bwa_update_bam1 derives */
	{
		std::vector<int> idx;
		for (size_t k = 0; k < b->kind.size(); ++k) if (b->kind[k] == 1 && !b->skip[k]) idx.push_back(b->first[k]);
		if (idx.size() == b->rec.size() && !idx.empty()) {       /* a batch of singletons only: in place, no gathering */
			int rc = nabwa_se_refine_strided(b->ix, (int)idx.size(), b->off.data(), b->seq.data(), b->rseq.data(), b->res, sizeof(nabwa_pe_t));
			if (rc != NABWA_OK) return rc;
		} else if (!idx.empty()) {
			std::vector<int64_t> off(idx.size() + 1, 0); std::vector<uint8_t> sq, rq; std::vector<nabwa_se_t> se(idx.size());
			for (size_t t = 0; t < idx.size(); ++t) {
				const int i = idx[t]; const int64_t L = b->off[i + 1] - b->off[i];
				sq.insert(sq.end(), b->seq.begin() + b->off[i], b->seq.begin() + b->off[i] + L);
				rq.insert(rq.end(), b->rseq.begin() + b->off[i], b->rseq.begin() + b->off[i] + L);
				off[t + 1] = off[t] + L; memcpy(&se[t], &b->res[i].se, sizeof(nabwa_se_t));
			}
			sq.push_back(0); rq.push_back(0);
			int rc = nabwa_se_refine(b->ix, (int)idx.size(), off.data(), sq.data(), rq.data(), se.data());
			if (rc != NABWA_OK) return rc;
			for (size_t t = 0; t < idx.size(); ++t) memcpy(&b->res[idx[t]].se, &se[t], sizeof(nabwa_se_t));
		}
	}
	/* ---- pairs, one read group at a time with that group's estimate (pass 2 draws no random numbers: its order is free) */
	{
		std::map<std::string, std::vector<int>> groups;
		std::vector<int> in_order;
		for (size_t k = 0; k < b->kind.size(); ++k) if (b->kind[k] == 2 && !b->skip[k]) { groups[b->rg_names[b->rg[k]]].push_back(b->first[k]); in_order.push_back(b->first[k]); }
		/* who is first with a wide hit row is settled in record order (finish_pair's cache of positions, pe_finish.hip), not in group order */
		if (groups.size() > 1) nabwa_poscache_register(tab->poscache, b->popt.max_occ, (int)in_order.size(), in_order.data(), b->n_aln.data(), b->row0.data(), b->rows.data(), b->res);
		for (auto &g : groups) {
			nabwa_isize_t ii;
			nabwa_isize_table_get(tab, g.first.c_str(), &ii);
			const std::vector<int> &idx = g.second;
			const int np = (int)idx.size();
			if (2 * (size_t)np == b->rec.size()) {           /* the whole batch is pairs of this one group: in place */
				int rc = nabwa_pe_finish_cached(b->ix, &b->opt, &b->popt, &ii, np, b->off.data(), b->seq.data(), b->rseq.data(), b->n_aln.data(), b->rows.data(), b->res, n_tot, n_mapped, tab->poscache);
				if (rc != NABWA_OK) return rc;
				continue;
			}
			/* otherwise the group's reads are gathered, by all threads; of a 3 KB record only what is filled travels */
			const size_t nr = 2 * (size_t)np;
			std::vector<int64_t> off(nr + 1, 0), r0(nr + 1, 0); std::vector<int32_t> na(nr);
			for (int t = 0; t < np; ++t) for (int e = 0; e < 2; ++e) {
				const int i = idx[t] + e; const size_t q = 2 * (size_t)t + e;
				off[q + 1] = off[q] + (b->off[i + 1] - b->off[i]); na[q] = b->n_aln[i]; r0[q + 1] = r0[q] + b->n_aln[i];
			}
			RawBytes sq, rq, rowb;
			const size_t pe_bytes = sizeof(nabwa_pe_t) * nr;
			nabwa_pe_t *pe = (nabwa_pe_t*)res_take(pe_bytes);
			if (!pe || !sq.alloc((size_t)off[nr] + 1) || !rq.alloc((size_t)off[nr] + 1) || !rowb.alloc(sizeof(nabwa_aln1_t) * ((size_t)r0[nr] + 1))) { res_give(pe, pe_bytes); return nabwa_fail(NABWA_ENOMEM, "out of memory for a read group's pairs"); }
			nabwa_aln1_t *rows = (nabwa_aln1_t*)rowb.data();
			bam_parallel(nr, [&](int, size_t lo, size_t hi) {
				for (size_t q = lo; q < hi; ++q) {
					const int i = idx[q >> 1] + (int)(q & 1);
					memcpy(sq.data() + off[q], b->seq.data() + b->off[i], (size_t)(off[q + 1] - off[q]));
					memcpy(rq.data() + off[q], b->rseq.data() + b->off[i], (size_t)(off[q + 1] - off[q]));
					if (na[q]) memcpy(rows + r0[q], b->rows.data() + b->row0[i], sizeof(nabwa_aln1_t) * (size_t)na[q]);
					copy_filled(pe[q], b->res[i]);
				}
			});
			sq.data()[off[nr]] = 0; rq.data()[off[nr]] = 0; memset(&rows[r0[nr]], 0, sizeof(nabwa_aln1_t));
			int rc = nabwa_pe_finish_cached(b->ix, &b->opt, &b->popt, &ii, np, off.data(), sq.data(), rq.data(), na.data(), rows, pe, n_tot, n_mapped, tab->poscache);
			if (rc == NABWA_OK) bam_parallel(nr, [&](int, size_t lo, size_t hi) { for (size_t q = lo; q < hi; ++q) copy_filled(b->res[idx[q >> 1] + (int)(q & 1)], pe[q]); });
			res_give(pe, pe_bytes);
			if (rc != NABWA_OK) return rc;
		}
	}
	/* ---- bwa_update_bam1 */
	const double tq1 = bam_now();
	bam_parallel(b->kind.size(), [&](int, size_t lo, size_t hi) {
		for (size_t k = lo; k < hi; ++k) {
			/* (a record's pieces lie far apart -- its parsed head, the end of its bytes in the arena where the tags go, the head and the MD field of
			 * its 3 KB working record: asked for ahead, the ones behind a pointer once that pointer is at hand) */
			if (k + 16 < hi) { const int f = b->first[k + 16]; __builtin_prefetch(&b->rec[f], 1); __builtin_prefetch(&b->res[f].se); __builtin_prefetch(b->res[f].se.md); __builtin_prefetch(&b->res[f].se.flag); }
			if (k + 8 < hi) { const BamRec &fr = b->rec[b->first[k + 8]]; __builtin_prefetch(fr.data.p + fr.data.n, 1); __builtin_prefetch(fr.data.p + fr.l_qname, 1); }
			const int i = b->first[k];
			if (b->skip[k]) continue;
			const bool dbg = (b->flags & NABWA_BAM_DEBUG) != 0;
			if (b->kind[k] == 1) update_bam(b->rec[i], R, b->res[i].se, 0, 0, b->opt.mode, b->opt.max_top2, dbg ? b->max_ent[i] : 0);
			else {
				update_bam(b->rec[i], R, b->res[i].se, &b->res[i + 1].se, &b->res[i], b->opt.mode, b->opt.max_top2, dbg ? b->max_ent[i] : 0);
				update_bam(b->rec[i + 1], R, b->res[i + 1].se, &b->res[i].se, &b->res[i + 1], b->opt.mode, b->opt.max_top2, dbg ? b->max_ent[i + 1] : 0);
			}
		}
	});
	b->phase = 2;
	/* the records are complete: the 3 KB per read that led to them go back to the pool (a batch may wait long for its turn to be written) */
	res_give(b->res, b->res_bytes); b->res = 0;
	if (timing) fprintf(stderr, "[nabwa] bam_batch_pass2 %zu records: finishing chains %.3f s, bwa_update_bam1 %.3f s\n", b->rec.size(), tq1 - tq0, bam_now() - tq1);
	return NABWA_OK;
}

/* the records as they now are (after create: cleaned; after pass 2: aligned), in logical-record order; with NABWA_BAM_ONLY_ALIGNED
 * without the logical records a read of which is flagged unmapped (pair_print_bam, bam2bam.c:911-925); out_off then has one more
 * entry than records were written, and the rest of its n_rec + 1 entries repeat the end */
extern "C" int nabwa_bam_batch_output(const nabwa_bam_batch_t *b, uint8_t *out, int64_t cap, int64_t *out_off, int64_t *n_bytes)
{
	if (!b || !n_bytes) return nabwa_fail(NABWA_EINVAL, "null argument");
	const size_t n = b->rec.size();
	std::vector<int> pick;
	if ((b->flags & NABWA_BAM_ONLY_ALIGNED) && b->phase == 2) {
		pick.reserve(n);
		for (size_t k = 0; k < b->kind.size(); ++k) {
			const int i = b->first[k];
			bool keep = true;
			for (int e = 0; e < b->kind[k]; ++e) if (b->rec[i + e].flag & F_SU) keep = false;      /* (before pass 2 the flag is the input's) */
			if (keep) for (int e = 0; e < b->kind[k]; ++e) pick.push_back(i + e);
		}
	} else {       /* every record, in the order they lie in (logical records are consecutive) */
		pick.resize(n);
		int *const pp = pick.data();
		bam_parallel(n, [pp](int, size_t lo, size_t hi) { for (size_t t = lo; t < hi; ++t) pp[t] = (int)t; });
	}
	const size_t m = pick.size();
	std::vector<int64_t> at(n + 1, 0);
	bam_parallel(m, [&](int, size_t lo, size_t hi) { for (size_t t = lo; t < hi; ++t) at[t + 1] = 36 + (int64_t)b->rec[pick[t]].data.size(); });      /* the sizes by all threads ... */
	for (size_t t = 0; t < m; ++t) at[t + 1] += at[t];                                                                                                     /* ... their sums by one */
	for (size_t t = m; t < n; ++t) at[t + 1] = at[m];
	if (out_off) memcpy(out_off, at.data(), sizeof(int64_t) * (n + 1));
	*n_bytes = at[m];
	if (!out || cap < at[m]) return nabwa_fail(NABWA_ECAP, "output buffer too small");
	bam_parallel(m, [&](int, size_t lo, size_t hi) {
		for (size_t t = lo; t < hi; ++t) {
			if (t + 8 < hi) { const BamRec &fr = b->rec[pick[t + 8]]; __builtin_prefetch(fr.data.p); __builtin_prefetch(fr.data.p + 64); __builtin_prefetch(fr.data.p + 128); __builtin_prefetch(fr.data.p + 192); }
			if (t + 16 < hi) __builtin_prefetch(&b->rec[pick[t + 16]]);
			write_rec(b->rec[pick[t]], out + at[t]);
		}
	});
	return NABWA_OK;
}

extern "C" int nabwa_bam_batch_kinds(const nabwa_bam_batch_t *b, uint8_t *kind_out)
{
	if (!b || !kind_out) return nabwa_fail(NABWA_EINVAL, "null argument");
	for (size_t k = 0; k < b->kind.size(); ++k) kind_out[k] = (uint8_t)b->kind[k];
	return NABWA_OK;
}

extern "C" int nabwa_bam_batch_counts(const nabwa_bam_batch_t *b, int *n_records, int *n_logical)
{
	if (!b) return nabwa_fail(NABWA_EINVAL, "null argument");
	if (n_records) *n_records = (int)b->rec.size();
	if (n_logical) *n_logical = (int)b->kind.size();
	return NABWA_OK;
}
