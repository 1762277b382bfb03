// aln_main.cpp -- `nabwa_aln`: the reference's `bwa aln` command (bwtaln.c:178-395) on top of libnabwa.so.
//
//   nabwa_aln [options] <prefix> <in.fq>  >  out.sai
//
// Same option letters, same gap_opt_t header, same record stream: the .sai it writes is byte-identical to the one
// `bwa aln` writes for the same arguments (tests/test_gpu_aln_cli.py), so the reference's samse / sampe / bam2bam -0/-1/-2
// consume it unchanged.  SURVEY.md 8f-4.  Host side only: FASTA/FASTQ parsing, read encoding and the batch plan live
// here, every SA interval comes from the GPU through nabwa_cal_sa_reg_gap.  No CPU search path exists.
//
// Batch plan.  The reference calls bwa_cal_sa_reg_gap on 0x40000 reads at a time (bwtaln.c:207) and that chunking is
// visible in the output in one corner: max_gapo is clamped to the max_diff of the LONGEST read of the call
// (bwtaln.c:104-105).  We keep the chunk boundaries as bookkeeping, give the GPU runs of consecutive chunks whose clamp
// comes out the same (normally: everything that was read), and split only where it differs.
//
// BAM input (-b, with -0/-1/-2) is read through zlib as the reference does (bamlite.h:7-11).  One deliberate difference:
// resuming into an existing -f file (attempt_recovery, bwtaln.c:259-296) continues the record stream; the reference
// writes a second copy of the 64-byte header at the resume point (bwtaln.c:387 is unconditional), which makes the
// resumed file unreadable.  A resumed file here equals the file of an uninterrupted run.
#include <ctype.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>
#include <algorithm>
#include <condition_variable>
#include <functional>
#include <chrono>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>
#include "../../include/nabwa.h"

#define MODE_BAM_ANY   (0x20 | 0x40 | 0x80 | 0x100)   /* BWA_MODE_BAM*, bwtaln.h:137-140 */
#define MODE_CFY       0x08
#define MODE_IL13      0x200
#define MAX_BCLEN      63                              /* bwtaln.h:30 */
#define REF_CHUNK      0x40000                         /* reads per bwa_cal_sa_reg_gap call, bwtaln.c:207 */
#define MIN_RDLEN      35                              /* bwtaln.h:28 */
#define BARCODE_LOW_Q  13                              /* bwaseqio.c:170 */

// ---------------------------------------------------------------------------------------------------------------------
// FASTA / FASTQ records the way kseq_read delivers them (kseq.h:155-193): a record starts at the next '>' or '@';
// the name ends at the first white space, the rest of the line is the comment; sequence characters are gathered
// up to the next '>', '+' or '@' WHEREVER it stands; after a '+' line, quality characters (33..127) are gathered
// until there are as many as bases, and one more character is consumed.
struct Fastx {
	gzFile fp = nullptr;
	std::vector<unsigned char> own;    /* gzip / stdin: the read buffer */
	const unsigned char *data = nullptr;   /* what the scans run over: `own`, or the whole file when it is plain and mapped */
	size_t have = 0, at = 0, file_size = 0;
	int pending = 0;
	bool eof = false, mapped = false, hit_limit = false;
	std::string name, comment, seq, qual;
	unsigned char cls[256];            /* sequence bytes: 0 = base character (isgraph), 1 = skipped, 2 = ends the sequence ('>' '+' '@') */

	void tables()
	{
		for (int c = 0; c < 256; ++c) cls[c] = isgraph(c) ? 0 : 1;
		cls['>'] = cls['+'] = cls['@'] = 2;
	}
	bool open(const char *fn)
	{
		tables();
		if (strcmp(fn, "-") != 0 && !getenv("NABWA_ALN_BUF")) {          /* a plain regular file is mapped: no copies, and its parse can be split */
			const int fd = ::open(fn, O_RDONLY);
			struct stat st;
			if (fd >= 0 && fstat(fd, &st) == 0 && S_ISREG(st.st_mode) && st.st_size > 2) {
				unsigned char magic[2] = { 0, 0 };
				if (pread(fd, magic, 2, 0) == 2 && !(magic[0] == 0x1f && magic[1] == 0x8b)) {
					void *m = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
					if (m != MAP_FAILED) {
						(void)madvise(m, (size_t)st.st_size, MADV_SEQUENTIAL);
						data = (const unsigned char*)m; have = file_size = (size_t)st.st_size; at = 0; eof = true; mapped = true;
						::close(fd);
						return true;
					}
				}
			}
			if (fd >= 0) ::close(fd);
		}
		fp = strcmp(fn, "-") == 0 ? gzdopen(fileno(stdin), "r") : gzopen(fn, "r");
		if (fp) gzbuffer(fp, 1 << 20);
		const char *bs = getenv("NABWA_ALN_BUF");             /* (tests: a few bytes, so that every scan meets the end of the buffer) */
		own.resize(bs && atoi(bs) > 0 ? (size_t)atoi(bs) : (size_t)4 << 20);
		data = own.data();
		return fp != nullptr;
	}
	/* a second reader on the same mapped file: bytes [from, limit) */
	void view(const Fastx &whole, size_t from, size_t limit)
	{
		tables();
		data = whole.data; file_size = whole.file_size; have = limit; at = from; eof = true; mapped = true; pending = 0; hit_limit = false;
	}
	void close() { if (fp) gzclose(fp); fp = nullptr; if (mapped && data && have == file_size && !own.size()) { /* the mapping lives until exit */ } }
	/* where the next record starts: the header character of a FASTA record may already have been taken */
	size_t logical_pos() const { return at - (pending ? 1 : 0); }
	/* true when data[at .. have) holds at least one byte */
	bool more()
	{
		if (at < have) return true;
		if (mapped) { if (have < file_size) hit_limit = true; return false; }
		if (eof) return false;
		const int got = gzread(fp, own.data(), (unsigned)own.size());
		at = 0; have = got > 0 ? (size_t)got : 0;
		if (got <= 0) { eof = true; return false; }
		return true;
	}
	/* length of the sequence, -1 at the end of the input, -2 for a truncated quality string.  The scans below run over
	 * what is in the buffer and append whole runs; they consume exactly the bytes the character-at-a-time description
	 * above consumes (tests/test_aln_parser.py holds that description as code). */
	int next()
	{
		int c = -1;
		if (!pending) {
			for (;;) {
				if (!more()) return -1;
				const unsigned char *p = data + at, *e = data + have;
				while (p < e && *p != '>' && *p != '@') ++p;
				at = (size_t)(p - data);
				if (p < e) { ++at; break; }
			}
		}
		pending = 0;
		name.clear(); comment.clear(); seq.clear(); qual.clear();
		for (c = -1;;) {                                     /* name: up to the first white space */
			if (!more()) break;
			const unsigned char *p = data + at, *e = data + have, *q = p;
			while (q < e && !isspace(*q)) ++q;
			name.append((const char*)p, (size_t)(q - p));
			at = (size_t)(q - data);
			if (q < e) { c = *q; ++at; break; }
		}
		if (c == -1 && name.empty()) return -1;
		if (c != '\n' && c != -1)                            /* comment: the rest of the line */
			for (;;) {
				if (!more()) break;
				const unsigned char *p = data + at, *e = data + have;
				const unsigned char *q = (const unsigned char*)memchr(p, '\n', (size_t)(e - p));
				comment.append((const char*)p, (size_t)((q ? q : e) - p));
				at = (size_t)((q ? q + 1 : e) - data);
				if (q) break;
			}
		for (c = -1;;) {                                     /* sequence: runs of base characters up to '>', '+' or '@' */
			if (!more()) break;
			const unsigned char *e = data + have, *q = data + at;
			while (q < e) {
				const unsigned char k = cls[*q];
				if (k == 0) { const unsigned char *r = q; do ++q; while (q < e && cls[*q] == 0); seq.append((const char*)r, (size_t)(q - r)); }
				else if (k == 1) ++q;
				else { c = *q; break; }
			}
			at = (size_t)(q - data);
			if (c != -1) { ++at; break; }
		}
		if (c == '>' || c == '@') pending = c;
		if (c != '+') return (int)seq.size();
		for (;;) {                                           /* the rest of the '+' line */
			if (!more()) return -2;
			const unsigned char *p = data + at, *e = data + have;
			const unsigned char *q = (const unsigned char*)memchr(p, '\n', (size_t)(e - p));
			at = (size_t)((q ? q + 1 : e) - data);
			if (q) break;
		}
		for (;;) {                                           /* quality: characters 33..127 until there is one per base */
			if (qual.size() >= seq.size()) { if (more()) ++at; break; }       /* ... and the character after them goes too */
			if (!more()) break;
			const unsigned char *e = data + have, *q = data + at;
			size_t need = seq.size() - qual.size();
			while (q < e && need) {
				const unsigned char *r = q;
				while (q < e && (size_t)(q - r) < need && *q >= 33 && *q <= 127) ++q;
				qual.append((const char*)r, (size_t)(q - r)); need -= (size_t)(q - r);
				if (need && q < e) ++q;                      /* a character that is not a quality (line break): skipped */
			}
			at = (size_t)(q - data);
		}
		if (qual.size() != seq.size()) return -2;
		return (int)seq.size();
	}
};

static uint8_t NT4[256];               /* nst_nt4_table (bntseq.c:39-56); its 5 for '-' is "not a base" like 4 everywhere on this path */
static void nt4_init()
{
	memset(NT4, 4, sizeof NT4);
	NT4['A'] = NT4['a'] = 0; NT4['C'] = NT4['c'] = 1; NT4['G'] = NT4['g'] = 2; NT4['T'] = NT4['t'] = 3;
}

struct Batch {                      /* what one GPU call (or a few) consumes */
	std::vector<int64_t> off{0};
	std::vector<uint8_t> seq, rseq;
	std::vector<int> chunk_max_len;  /* longest read of each REF_CHUNK-sized piece */
	int n() const { return (int)off.size() - 1; }
};

/* BAM records as bwa_read_bam takes them (bwaseqio.c:125-168 over bamlite.c:73-155): any gzip container (BGZF is a
 * series of gzip members; the reference opens BAM with gzopen as well, bamlite.h:7-11), header skipped, then per record the
 * flag, the 4-bit bases and the qualities.  `which`: 1 = first reads of pairs, 2 = second reads, 4 = unpaired (bwtaln.c:167-172). */
/* a damaged BGZF block ends the run (the reader inflates blocks on several threads: no exit handlers under them) */
static void die(const char *what, const char *why) { fprintf(stderr, "[nabwa_aln] %s: %s\n", what, why); fflush(stderr); _exit(2); }
#include "bgzf_in.hpp"

struct BamReader {
	FILE *file = nullptr;
	std::unique_ptr<BamIn> fp;              /* BGZF blocks inflated many at a time; any other gzip stream, or none, as gzread takes it */
	int which = 7;
	std::vector<unsigned char> rec;

	bool get(void *dst, size_t n) { return n == 0 || fp->read(dst, n); }
	bool skip(size_t n) { unsigned char tmp[4096]; while (n) { const size_t k = n < sizeof tmp ? n : sizeof tmp; if (!get(tmp, k)) return false; n -= k; } return true; }
	bool open(const char *fn)
	{
		file = strcmp(fn, "-") == 0 ? stdin : fopen(fn, "rb");
		if (!file) return false;
		fp.reset(new BamIn(file, fn));
		char magic[4]; int32_t l_text = 0, n_ref = 0;
		if (!get(magic, 4) || memcmp(magic, "BAM\1", 4) != 0) { fprintf(stderr, "[nabwa_aln] invalid BAM binary header (this is not a BAM file).\n"); return false; }
		if (!get(&l_text, 4) || l_text < 0 || !skip((size_t)l_text) || !get(&n_ref, 4) || n_ref < 0) return false;
		for (int32_t i = 0; i < n_ref; ++i) { int32_t l_name = 0; if (!get(&l_name, 4) || l_name < 0 || !skip((size_t)l_name + 4)) return false; }
		return true;
	}
	void close() { fp.reset(); if (file && file != stdin) fclose(file); file = nullptr; }
	/* the next record that passes the selection: flag, number of bases, pointers to 4-bit bases and qualities; false at the end */
	bool next(unsigned *flag, int *l_seq, const unsigned char **bases, const unsigned char **qual)
	{
		for (;;) {
			int32_t block = 0; uint32_t x[8];
			if (!get(&block, 4) || block < 32 || !get(x, 32)) return false;
			rec.resize((size_t)block - 32 + 1);
			if (!get(rec.data(), (size_t)block - 32)) return false;
			const unsigned l_qname = x[2] & 0xffu, n_cigar = x[3] & 0xffffu; *flag = x[3] >> 16; *l_seq = (int)x[4];
			const size_t need = (size_t)l_qname + 4u * n_cigar + ((size_t)*l_seq + 1) / 2 + (size_t)*l_seq;
			if (*l_seq < 0 || need > (size_t)block - 32) return false;
			const bool paired = *flag & 1u;
			if (!(((which & 1) && paired && (*flag & 64u)) || ((which & 2) && paired && (*flag & 128u)) || ((which & 4) && !paired))) continue;
			*bases = rec.data() + l_qname + 4u * n_cigar; *qual = *bases + ((size_t)*l_seq + 1) / 2;
			return true;
		}
	}
};

struct Source {                     /* bwa_read_seq (bwaseqio.c:172-252) minus the bwa_seq_t records */
	Fastx fx;
	int mode, trim_qual;
	long n_trimmed = 0, n_tot = 0;

	BamReader *bam = nullptr;           /* -b: records come from here instead of fx */

	/* bwa_trim_read (bwaseqio.c:110-123) on ASCII qualities q[0..full) with the given offset: the length that is kept */
	static int trimmed_len(const unsigned char *q, int full, int trim_qual, int shift)
	{
		int sum = 0, best = 0, best_l = full - 1;
		for (int l = full - 1; l >= MIN_RDLEN - 1; --l) {
			sum += trim_qual - ((int)q[l] - shift);
			if (sum < 0) break;
			if (sum > best) { best = sum; best_l = l; }
		}
		return best_l + 1;
	}
	void append(Batch *b, const uint8_t *fwd, int len)   /* fwd: codes of the read as sequenced; stored reversed / reverse-complemented */
	{
		const size_t at = b->seq.size();
		b->seq.resize(at + len); b->rseq.resize(at + len);
		uint8_t *const ps = b->seq.data() + at, *const pr = b->rseq.data() + at;
		const uint8_t flip = (mode & NABWA_MODE_COMPREAD) ? 3 : 0;
		for (int i = 0; i < len; ++i) { const uint8_t c = fwd[len - 1 - i]; ps[i] = c; pr[i] = c < 4 ? c ^ flip : c; }
		if (b->n() % REF_CHUNK == 0) b->chunk_max_len.push_back(0);
		if (len > b->chunk_max_len.back()) b->chunk_max_len.back() = len;
		b->off.push_back((int64_t)(at + len));
	}
	/* bwa_read_bam (bwaseqio.c:125-168): no barcode, no Casava filter, empty reads are kept */
	bool one_bam(Batch *b)
	{
		static const uint8_t nt16_nt4[16] = { 4, 0, 1, 4, 2, 4, 4, 4, 3, 4, 4, 4, 4, 4, 4, 4 };
		unsigned flag; int l; const unsigned char *s4, *q;
		if (!bam->next(&flag, &l, &s4, &q)) return false;
		if (!b) return true;
		if (l > 65535) { fprintf(stderr, "[nabwa_aln] a read is longer than 65535 bases\n"); exit(1); }
		std::vector<uint8_t> code(l ? l : 1), qa(l ? l : 1);
		for (int i = 0; i < l; ++i) {
			code[i] = nt16_nt4[s4[i >> 1] >> 4 * (1 - (i & 1)) & 0xf];
			qa[i] = (uint8_t)((int)q[i] + 33 < 126 ? q[i] + 33 : 126);
		}
		if (flag & 16u) {                                           /* stored reverse-complemented: back to the read as sequenced */
			std::reverse(code.begin(), code.begin() + l); std::reverse(qa.begin(), qa.begin() + l);
			for (int i = 0; i < l; ++i) if (code[i] < 4) code[i] = 3 - code[i];
		}
		int len = l;
		if (trim_qual >= 1) { len = trimmed_len(qa.data(), l, trim_qual, 33); n_trimmed += l - len; }
		n_tot += l;
		append(b, code.data(), len);
		return true;
	}

	/* reads the next record that survives the filters; appends it to b unless b is null (skipping) */
	bool one(Batch *b)
	{
		if (bam) return one_bam(b);
		const int l_bc = (int)((unsigned)mode >> 24);
		for (;;) {
			if (fx.next() < 0) return false;
			if ((mode & MODE_CFY) && !fx.comment.empty()) {             /* Casava filter flag: "...:Y..." */
				const size_t p = fx.comment.find(':');
				if (p != std::string::npos && p + 1 < fx.comment.size() && fx.comment[p + 1] == 'Y') continue;
			}
			if ((int)fx.seq.size() <= l_bc) continue;                    /* nothing left after the barcode (also: empty reads) */
			break;
		}
		if (!b) return true;
		const int full = (int)fx.seq.size() - l_bc;
		const char *s = fx.seq.data() + l_bc;
		int len = full;
		if (!fx.qual.empty() && trim_qual >= 1) {                        /* bwa_trim_read, bwaseqio.c:110-123 */
			const int shift = 33 + ((mode & MODE_IL13) ? 31 : 0);
			const char *q = fx.qual.data() + l_bc;
			int sum = 0, best = 0, best_l = full - 1;
			for (int l = full - 1; l >= MIN_RDLEN - 1; --l) {
				sum += trim_qual - ((int)(unsigned char)q[l] - shift);
				if (sum < 0) break;
				if (sum > best) { best = sum; best_l = l; }
			}
			len = best_l + 1;
			n_trimmed += full - len;
		}
		n_tot += full;
		if (len > 65535) { fprintf(stderr, "[nabwa_aln] read '%s' is longer than 65535 bases\n", fx.name.c_str()); exit(1); }
		const bool comp = mode & NABWA_MODE_COMPREAD;
		const size_t at = b->seq.size();
		b->seq.resize(at + len); b->rseq.resize(at + len);
		uint8_t *const ps = b->seq.data() + at, *const pr = b->rseq.data() + at;
		const unsigned char *const last = (const unsigned char*)s + len - 1;
		const uint8_t flip = comp ? 3 : 0;
		for (int i = 0; i < len; ++i) {                                  /* seq: the read reversed; rseq: its reverse complement */
			const uint8_t c = NT4[last[-i]];
			ps[i] = c;
			pr[i] = c < 4 ? c ^ flip : c;                                /* 3 - c == c ^ 3 for 0..3 */
		}
		const int idx = b->n();
		if (idx % REF_CHUNK == 0) b->chunk_max_len.push_back(0);
		if (len > b->chunk_max_len.back()) b->chunk_max_len.back() = len;
		b->off.push_back((int64_t)(at + len));
		return true;
	}
};

// ---------------------------------------------------------------------------------------------------------------------
// From records to GPU batches.  Fragments of parsed reads (from one sequential parser, or from several parsers working on
// pieces of a mapped file) are appended to the batch under construction; a batch is handed on when it holds batch_reads
// reads (a multiple of the reference's chunk) or, for long reads, 1 Gi bases at a chunk boundary.
struct Assembler {
	long batch_reads;
	std::function<void(std::unique_ptr<Batch>)> emit;
	std::unique_ptr<Batch> cur{new Batch};

	void add(const Batch &f)
	{
		for (int i = 0; i < f.n(); ) {
			Batch &b = *cur;
			// as many reads of the fragment as the batch under construction still takes, bases copied in one piece
			int k = f.n() - i;
			if ((long)k > batch_reads - b.n()) k = (int)(batch_reads - b.n());
			const int to_chunk_end = REF_CHUNK - b.n() % REF_CHUNK;           /* the 1 Gi-base rule is looked at on chunk boundaries */
			if (k > to_chunk_end) k = to_chunk_end;
			const size_t o = (size_t)f.off[i], bytes = (size_t)(f.off[i + k] - f.off[i]), at = b.seq.size();
			b.seq.insert(b.seq.end(), f.seq.begin() + o, f.seq.begin() + o + bytes);
			b.rseq.insert(b.rseq.end(), f.rseq.begin() + o, f.rseq.begin() + o + bytes);
			if (b.n() % REF_CHUNK == 0) b.chunk_max_len.push_back(0);
			int mx = b.chunk_max_len.back();
			for (int j = 0; j < k; ++j) {
				const int len = (int)(f.off[i + j + 1] - f.off[i + j]);
				if (len > mx) mx = len;
				b.off.push_back((int64_t)at + (f.off[i + j + 1] - f.off[i]));
			}
			b.chunk_max_len.back() = mx;
			i += k;
			if (b.n() >= batch_reads || (b.seq.size() >= (1ull << 30) && b.n() % REF_CHUNK == 0)) flush();
		}
	}
	void flush()
	{
		if (!cur->n()) return;
		const size_t cap = cur->seq.size() + cur->seq.size() / 8;              /* the next batch will be about as large: no regrowth copies */
		emit(std::move(cur));
		cur.reset(new Batch);
		cur->seq.reserve(cap); cur->rseq.reserve(cap);
	}
};

/* Where a record may start at or after `from` (and before `end`): after a line break, '>' (a FASTA header wherever it
 * stands), or '@' whose line after next begins with '+' (the four-line FASTQ shape; a quality line may begin with '@' too,
 * and then the line after next is a sequence).  A guess -- the caller checks it against the parse that arrives there. */
static size_t guess_record_start(const unsigned char *d, size_t from, size_t end)
{
	for (size_t p = from; p < end; ) {
		const unsigned char *nl = (const unsigned char*)memchr(d + p, '\n', end - p);
		if (!nl) break;
		p = (size_t)(nl - d) + 1;
		if (p >= end) break;
		if (d[p] == '>') return p;
		if (d[p] == '@') {
			const unsigned char *l1 = (const unsigned char*)memchr(d + p, '\n', end - p);
			if (!l1) break;
			const unsigned char *l2 = (const unsigned char*)memchr(l1 + 1, '\n', (size_t)(d + end - (l1 + 1)));
			if (!l2) break;
			if (l2 + 1 < d + end && l2[1] == '+') return p;
		}
	}
	return end;
}

/* Everything the input still holds, as batches through `as`.  A mapped file is parsed a window at a time by several
 * parsers, each starting at a guessed record start; parser j's reads are taken when parser j-1 (whose own start was right)
 * stopped exactly where j began -- then j's parse is what the single sequential parse would have produced from there.  At
 * the first piece that does not line up, the rest of the window is dropped and the next window starts where the last
 * good parser stopped, which is a record boundary of the sequential parse by construction. */
static double dbg_now() { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; }

static void read_everything(Source &src, Assembler &as)
{
	int n_thr = getenv("NABWA_ALN_THREADS") ? atoi(getenv("NABWA_ALN_THREADS")) : (int)std::min(8u, std::max(1u, std::thread::hardware_concurrency()));
	if (!src.fx.mapped || n_thr < 2) {
		Batch frag;
		for (;;) {
			frag = Batch();
			while (frag.n() < 65536 && src.one(&frag)) {}
			if (frag.n() == 0) break;
			as.add(frag);
			if (frag.n() < 65536) break;
		}
		as.flush();
		return;
	}
	const size_t window = (getenv("NABWA_ALN_WINDOW") ? (size_t)atol(getenv("NABWA_ALN_WINDOW")) : (size_t)64 << 20);
	const unsigned char *d = src.fx.data; const size_t size = src.fx.file_size;
	size_t pos = src.fx.logical_pos();
	bool input_ended = false;
	struct Piece { Source s; Batch b; size_t stop_at = 0; bool ended = false; };
	std::vector<Piece> piece;
	while (pos < size && !input_ended) {
		const double t_start = dbg_now();
		const size_t wend = pos + window < size ? pos + window : size;
		std::vector<size_t> cut{pos};
		for (int k = 1; k < n_thr; ++k) {
			const size_t c = guess_record_start(d, pos + (wend - pos) / n_thr * k, wend);
			if (c > cut.back() && c < wend) cut.push_back(c);
		}
		cut.push_back(wend == size ? size : guess_record_start(d, wend, size));
		const int m = (int)cut.size() - 1;
		if ((int)piece.size() < m) piece.resize(m);
		std::vector<std::thread> th;
		for (int j = 0; j < m; ++j) {
			Piece &q = piece[j];
			q.b.off.assign(1, 0); q.b.seq.clear(); q.b.rseq.clear(); q.b.chunk_max_len.clear();     /* the buffers of the window before are used again */
			q.stop_at = 0; q.ended = false; q.s.n_trimmed = q.s.n_tot = 0;
			q.s.mode = src.mode; q.s.trim_qual = src.trim_qual;
			const size_t limit = cut[j + 1] + ((size_t)64 << 20) < size ? cut[j + 1] + ((size_t)64 << 20) : size;   /* a parser on a wrong start does not run to the end of the file */
			q.s.fx.view(src.fx, cut[j], limit);
			auto work = [&q, stop = cut[j + 1]]() {
				while (q.s.fx.logical_pos() < stop) if (!q.s.one(&q.b)) { q.ended = true; break; }
				q.stop_at = q.s.fx.logical_pos();
			};
			if (m == 1) work(); else th.emplace_back(work);
		}
		for (auto &x : th) x.join();
		const double t_parsed = dbg_now();
		size_t good_to = pos;
		for (int j = 0; j < m; ++j) {
			Piece &q = piece[j];
			if (q.s.fx.hit_limit) {          /* ran into its look-ahead limit: parse this stretch again without one, alone */
				q.b = Batch(); q.ended = false; q.s.n_trimmed = q.s.n_tot = 0;
				q.s.fx.view(src.fx, cut[j], size);
				while (q.s.fx.logical_pos() < cut[j + 1]) if (!q.s.one(&q.b)) { q.ended = true; break; }
				q.stop_at = q.s.fx.logical_pos();
			}
			as.add(q.b);
			src.n_trimmed += q.s.n_trimmed; src.n_tot += q.s.n_tot;
			good_to = q.stop_at;
			if (q.ended) { input_ended = true; break; }            /* end of the input, or a truncated quality string: reading stops for good */
			if (q.stop_at != cut[j + 1]) break;                   /* the next piece did not start on a record of this parse */
		}
		if (getenv("NABWA_ALN_DEBUG")) fprintf(stderr, "[nabwa_aln] window at %zu: %d pieces, accepted up to %zu of %zu; parse %.3f s, assemble %.3f s\n", pos, m, good_to, cut[m], t_parsed - t_start, dbg_now() - t_parsed);
		pos = good_to;
	}
	as.flush();
}

// ---------------------------------------------------------------------------------------------------------------------
struct Resume { int skip = 0; long at = 0; bool found = false; };

/* attempt_recovery (bwtaln.c:259-296): count the complete records of an earlier, interrupted run */
static Resume look_for_earlier_output(const char *fn, nabwa_gap_opt_t *opt)
{
	Resume r;
	FILE *f = fopen(fn, "rb");
	nabwa_gap_opt_t old;
	if (f && fread(&old, 1, sizeof(old), f) == sizeof(old)) {
		fprintf(stderr, "[nabwa_aln] %s exists, attempting recovery.\n", fn);
		std::vector<nabwa_aln1_t> rows;
		for (;;) {
			int32_t n_aln;
			r.at = ftell(f);
			if (fread(&n_aln, 1, 4, f) < 4 || n_aln < 0) break;
			rows.resize(n_aln ? n_aln : 1);
			if (n_aln && fread(rows.data(), sizeof(nabwa_aln1_t), n_aln, f) < (size_t)n_aln) break;
			++r.skip;
		}
		fprintf(stderr, "[nabwa_aln] %d records up to position %ld.\n", r.skip, r.at);
		*opt = old;
		r.found = true;
	}
	if (f) fclose(f);
	return r;
}

// The command line of `bwa aln` (bwtaln.c:303-340) as one table: option letter -> what it sets.  The getopt string, the
// parser and the usage text are all generated from it.
struct Opt {
	char letter;
	int nabwa_gap_opt_t::*field;       // integer option: the member it sets ...
	int set_bits, clear_bits;          // ... or flag: mode bits it sets / clears
	const char *arg, *help;
};
static const Opt OPTS[] = {
	{ 'n', nullptr, 0, 0, "NUM", "differences allowed: a count, or (with a '.') the fraction of reads that may be missed at 2% base error" },
	{ 'o', &nabwa_gap_opt_t::max_gapo, 0, 0, "INT", "gap opens allowed" },
	{ 'e', nullptr, 0, 0, "INT", "gap extensions allowed; -1: long gaps off, extensions count as differences" },
	{ 'i', &nabwa_gap_opt_t::indel_end_skip, 0, 0, "INT", "no indel within INT bases of the read ends" },
	{ 'd', &nabwa_gap_opt_t::max_del_occ, 0, 0, "INT", "a long deletion is only extended while the interval holds at most INT rows" },
	{ 'l', &nabwa_gap_opt_t::seed_len, 0, 0, "INT", "seed length" },
	{ 'k', &nabwa_gap_opt_t::max_seed_diff, 0, 0, "INT", "differences allowed in the seed" },
	{ 'm', &nabwa_gap_opt_t::max_entries, 0, 0, "INT", "a search is cut off beyond INT queued entries" },
	{ 't', &nabwa_gap_opt_t::n_threads, 0, 0, "INT", "written to the header; no other effect (as in the reference)" },
	{ 'M', &nabwa_gap_opt_t::s_mm, 0, 0, "INT", "mismatch penalty" },
	{ 'O', &nabwa_gap_opt_t::s_gapo, 0, 0, "INT", "gap open penalty" },
	{ 'E', &nabwa_gap_opt_t::s_gape, 0, 0, "INT", "gap extension penalty" },
	{ 'R', &nabwa_gap_opt_t::max_top2, 0, 0, "INT", "go on to sub-optimal hits only while there are at most INT best ones" },
	{ 'q', &nabwa_gap_opt_t::trim_qual, 0, 0, "INT", "trim the 3' end by quality INT (never below 35 bases)" },
	{ 'f', nullptr, 0, 0, "FILE", "write here instead of stdout; continues an interrupted FILE, renames 'x_' to 'x' when done" },
	{ 'B', nullptr, 0, 0, "INT", "the first INT bases are a barcode" },
	{ 'c', nullptr, 0, NABWA_MODE_COMPREAD, nullptr, "colour-space reads: reverse, do not complement" },
	{ 'L', nullptr, NABWA_MODE_LOGGAP, 0, nullptr, "log-scaled penalty for long deletions" },
	{ 'N', nullptr, NABWA_MODE_NONSTOP, 0, nullptr, "do not stop at the best score: every hit within the allowed differences" },
	{ 'I', nullptr, MODE_IL13, 0, nullptr, "qualities are Illumina 1.3+ (offset 64)" },
	{ 'Y', nullptr, MODE_CFY, 0, nullptr, "drop reads whose Casava comment says 'filtered'" },
	{ 'b', nullptr, 0x20, 0, nullptr, "the input is BAM (BGZF or plain gzip)" },
	{ '0', nullptr, 0x40, 0, nullptr, "with -b: unpaired reads only" },
	{ '1', nullptr, 0x80, 0, nullptr, "with -b: first reads of pairs only" },
	{ '2', nullptr, 0x100, 0, nullptr, "with -b: second reads of pairs only" },
};

/* bwa_open_reads (bwtaln.c:164-176): BAM with the read selection of -0 -1 -2 (none given: all), or FASTA/FASTQ */
static bool open_source(Source &src, BamReader &bam, const nabwa_gap_opt_t &opt, const char *fn)
{
	src.mode = opt.mode; src.trim_qual = opt.trim_qual;
	if (opt.mode & 0x20) {
		int which = ((opt.mode & 0x40) ? 4 : 0) | ((opt.mode & 0x80) ? 1 : 0) | ((opt.mode & 0x100) ? 2 : 0);
		bam.which = which ? which : 7;
		if (!bam.open(fn)) return false;
		src.bam = &bam;
		return true;
	}
	return src.fx.open(fn);
}

static int usage(const nabwa_gap_opt_t *o)
{
	fprintf(stderr, "\nUsage:   nabwa_aln [options] <prefix> <in.fq>   >   out.sai\n\n");
	for (const Opt &d : OPTS) {
		if (!d.help) continue;
		char dflt[32] = "";
		if (d.field) snprintf(dflt, sizeof dflt, " [%d]", o->*d.field);
		else if (d.letter == 'n') snprintf(dflt, sizeof dflt, " [%.2f]", o->fnr);
		fprintf(stderr, "         -%c %-5s %s%s\n", d.letter, d.arg ? d.arg : "", d.help, dflt);
	}
	fprintf(stderr, "\n");
	fprintf(stderr, "Environment: NABWA_DEVICES (GPUs to use, e.g. 0,1,2,3; default NABWA_DEVICE or 0), NABWA_ALN_BATCH (reads per GPU batch, 4194304),\n             NABWA_ALN_THREADS (parser threads for plain input, 8)\n\n");
	return 1;
}

int main(int argc, char *argv[])
{
	nabwa_gap_opt_t opt;
	nabwa_gap_init_opt(&opt);
	nt4_init();
	int c, opte = -1;
	const char *ofile = nullptr;
	Resume resume;
	std::string letters;
	for (const Opt &d : OPTS) { letters += d.letter; if (d.arg) letters += ':'; }
	while ((c = getopt(argc, argv, letters.c_str())) >= 0) {
		const Opt *d = nullptr;
		for (const Opt &x : OPTS) if (x.letter == c) d = &x;
		if (!d) return 1;
		if (d->field) { opt.*(d->field) = atoi(optarg); continue; }
		opt.mode = (opt.mode | d->set_bits) & ~d->clear_bits;
		switch (c) {           /* the four that are more than a field or a flag */
		case 'n':              /* "0.04" is a miss rate, "4" a count */
			if (strchr(optarg, '.')) { opt.fnr = (float)atof(optarg); opt.max_diff = -1; }
			else { opt.max_diff = atoi(optarg); opt.fnr = -1.0f; }
			break;
		case 'e': opte = atoi(optarg); break;
		case 'N': opt.max_top2 = 0x7fffffff; break;
		case 'B': opt.mode |= atoi(optarg) << 24; break;
		case 'f': ofile = optarg; resume = look_for_earlier_output(optarg, &opt); break;   /* options after -f still apply, as in the reference */
		default: break;
		}
	}
	if (opte > 0) { opt.max_gape = opte; opt.mode &= ~NABWA_MODE_GAPE; }
	if (optind + 2 > argc) return usage(&opt);
	if ((int)((unsigned)opt.mode >> 24) > MAX_BCLEN) { fprintf(stderr, "[nabwa_aln] the maximum barcode length is %d.\n", MAX_BCLEN); return 1; }
	if (opt.fnr > 0.0f)
		for (int i = 17, k = 0; i <= 250; ++i) {
			const int l = nabwa_cal_maxdiff(i, 0.02, opt.fnr);
			if (l != k) fprintf(stderr, "[nabwa_aln] %dbp reads: max_diff = %d\n", i, l);
			k = l;
		}
	const char *prefix = argv[optind], *reads = argv[optind + 1];

	// ---- NABWA_ALN_PARSE_ONLY: stop after the host side (no index, no GPU, no .sai) and say what the reads look like after
	// parsing, filtering, trimming and encoding: "reads N bases M fnv H" (=2: also one line per read "len fnv").  Lets the CPU
	// tests check this file's share of the work against an independent restatement, and times the parser.
	if (getenv("NABWA_ALN_PARSE_ONLY")) {
		const bool per_read = atoi(getenv("NABWA_ALN_PARSE_ONLY")) >= 2;
		Source src; BamReader bam;
		if (!open_source(src, bam, opt, reads)) { fprintf(stderr, "[nabwa_aln] fail to open file '%s'. Abort!\n", reads); return 2; }
		auto fnv = [](uint64_t h, const uint8_t *p, size_t n) { for (size_t i = 0; i < n; ++i) { h ^= p[i]; h *= 1099511628211ull; } return h; };
		uint64_t all = 1469598103934665603ull; long n_reads = 0, n_bases = 0;
		Assembler as;
		as.batch_reads = REF_CHUNK;
		as.emit = [&](std::unique_ptr<Batch> bp) {
			const Batch &b = *bp;
			for (int i = 0; i < b.n(); ++i) {
				const size_t o = (size_t)b.off[i], len = (size_t)(b.off[i + 1] - b.off[i]);
				n_bases += (long)len;
				if (!per_read) continue;                                  /* =1: counts only (timing the parser) */
				uint64_t h = fnv(fnv(1469598103934665603ull, b.seq.data() + o, len), b.rseq.data() + o, len);
				printf("%zu %016llx\n", len, (unsigned long long)h);
				all = fnv(all, (const uint8_t*)&h, 8);
			}
			n_reads += b.n();
		};
		read_everything(src, as);
		printf("reads %ld bases %ld fnv %016llx\n", n_reads, n_bases, (unsigned long long)all);
		return 0;
	}

	// ---- the reads: nothing is written before they can be opened
	Source src; BamReader bam;
	if (!open_source(src, bam, opt, reads)) { fprintf(stderr, "[nabwa_aln] fail to open file '%s'. Abort!\n", reads); return 2; }

	// ---- the index, one replica per GPU of NABWA_DEVICES ("0,1,2,3"; default: NABWA_DEVICE or 0): no GPU, no output
	std::vector<int> devices;
	if (getenv("NABWA_DEVICES")) {
		for (const char *q = getenv("NABWA_DEVICES"); *q; ) { char *e; const long d = strtol(q, &e, 10); if (e == q) break; devices.push_back((int)d); q = *e == ',' ? e + 1 : e; }
	}
	if (devices.empty()) devices.push_back(getenv("NABWA_DEVICE") ? atoi(getenv("NABWA_DEVICE")) : 0);
	const std::string sa_path = std::string(prefix) + ".sa", rsa_path = std::string(prefix) + ".rsa";
	const int with_sa = access(sa_path.c_str(), R_OK) == 0 && access(rsa_path.c_str(), R_OK) == 0;   /* optional: lets the library build its text-mode companions */
	std::vector<nabwa_index_t*> ixs(devices.size(), nullptr);
	{
		std::vector<std::string> err(devices.size());
		std::vector<std::thread> th;
		for (size_t g = 0; g < devices.size(); ++g)
			th.emplace_back([&, g]() { if (nabwa_index_load(prefix, devices[g], with_sa, 0, &ixs[g]) != NABWA_OK) { err[g] = nabwa_last_error(); ixs[g] = nullptr; } });
		for (auto &x : th) x.join();
		for (size_t g = 0; g < devices.size(); ++g)
			if (!ixs[g]) {
				fprintf(stderr, "[nabwa_aln] cannot set up the index on GPU %d: %s\n", devices[g], err[g].c_str());
				for (nabwa_index_t *p : ixs) if (p) nabwa_index_destroy(p);
				return 2;
			}
	}

	FILE *out = stdout;
	if (ofile) {
		out = fopen(ofile, resume.found ? "rb+" : "wb");
		if (!out) { fprintf(stderr, "[nabwa_aln] fail to open file '%s': ", ofile); perror(nullptr); return 2; }
		if (resume.found && fseek(out, resume.at, SEEK_SET) != 0) { fprintf(stderr, "[nabwa_aln] seek failed, aborting.\n"); return 2; }
	}
	if (!resume.found && fwrite(&opt, sizeof(opt), 1, out) != 1) { perror("[nabwa_aln] write"); return 2; }

	if (resume.skip) {
		fprintf(stderr, "[nabwa_aln] skipping %d sequences.\n", resume.skip);
		for (int i = 0; i < resume.skip; ++i)
			if (!src.one(nullptr)) { fprintf(stderr, "[nabwa_aln] EOF while skipping done work. Aborting.\n"); return 1; }
	}

	// ---- reader thread: parses and encodes the next batches while the GPU works on the current one
	long batch_reads = getenv("NABWA_ALN_BATCH") ? atol(getenv("NABWA_ALN_BATCH")) : (4l << 20);
	if (batch_reads < REF_CHUNK) batch_reads = REF_CHUNK;
	batch_reads -= batch_reads % REF_CHUNK;                               /* batches end on the reference's chunk boundaries */
	std::mutex mu; std::condition_variable cv;
	std::deque<std::unique_ptr<Batch>> ready; bool done = false;
	std::thread reader([&]() {
		Assembler as;
		as.batch_reads = batch_reads;
		as.emit = [&](std::unique_ptr<Batch> b) {
			std::unique_lock<std::mutex> lk(mu);
			cv.wait(lk, [&] { return ready.size() < 2; });
			ready.push_back(std::move(b));
			cv.notify_all();
		};
		read_everything(src, as);
		std::unique_lock<std::mutex> lk(mu);
		done = true;
		cv.notify_all();
	});

	// ---- one worker per GPU takes the batches as they come; the records leave in batch order (the .sai is positional)
	long tot = 0; int status = 0;
	/* a batch through one GPU: runs of chunks with the same max_gapo clamp -> one call each; the record stream (n_aln, then the
	 * rows; bwtaln.c:242-246) of the whole batch into obuf */
	auto search_batch = [&](nabwa_index_t *ix, const Batch &b, std::vector<char> &obuf) -> bool {
		std::vector<int32_t> n_aln, max_entries;
		std::vector<nabwa_aln1_t> rows;
		obuf.clear();
		const int n_chunks = (int)b.chunk_max_len.size();
		auto clamp_of = [&](int ch) {
			const int md = opt.fnr > 0.0f ? nabwa_cal_maxdiff(b.chunk_max_len[ch], 0.02, opt.fnr) : opt.max_diff;
			return md < opt.max_gapo ? md : opt.max_gapo;
		};
		for (int c0 = 0; c0 < n_chunks; ) {
			int c1 = c0 + 1;
			while (c1 < n_chunks && clamp_of(c1) == clamp_of(c0)) ++c1;
			const int r0 = c0 * REF_CHUNK, r1 = c1 * REF_CHUNK < b.n() ? c1 * REF_CHUNK : b.n(), n = r1 - r0;
			std::vector<int64_t> off(n + 1);
			const int64_t base = b.off[r0];
			for (int i = 0; i <= n; ++i) off[i] = b.off[r0 + i] - base;
			n_aln.resize(n); max_entries.resize(n);
			int64_t cap = (int64_t)n + n / 4 + 1024, n_rows = 0;
			int rc = NABWA_OK;
			for (int attempt = 0; attempt < 2; ++attempt) {      /* the second attempt has the row count the first one reported */
				rows.resize(cap);
				rc = nabwa_cal_sa_reg_gap(ix, &opt, n, off.data(), b.seq.data() + base, b.rseq.data() + base, 0,
										  n_aln.data(), rows.data(), cap, &n_rows, max_entries.data());
				if (rc != NABWA_ECAP || n_rows <= cap) break;
				cap = n_rows;
			}
			if (rc != NABWA_OK) { fprintf(stderr, "[nabwa_aln] GPU search failed: %s\n", nabwa_last_error()); return false; }
			const size_t at = obuf.size();
			obuf.resize(at + (size_t)n * 4 + (size_t)n_rows * sizeof(nabwa_aln1_t));
			char *w = obuf.data() + at; const nabwa_aln1_t *r = rows.data();
			for (int i = 0; i < n; ++i) {
				memcpy(w, &n_aln[i], 4); w += 4;
				memcpy(w, r, (size_t)n_aln[i] * sizeof(nabwa_aln1_t)); w += (size_t)n_aln[i] * sizeof(nabwa_aln1_t); r += n_aln[i];
			}
			c0 = c1;
		}
		return true;
	};
	struct Done { std::vector<char> bytes; int n_reads = 0; bool ok = false; };
	std::mutex omu; std::condition_variable ocv;
	std::map<long, Done> finished;            /* batch number -> its records, until the writer gets to it */
	long next_in = 0, next_out = 0; bool failed = false;
	std::vector<std::thread> workers;
	for (size_t g = 0; g < ixs.size(); ++g)
		workers.emplace_back([&, g]() {
			for (;;) {
				std::unique_ptr<Batch> b; long no;
				{
					std::unique_lock<std::mutex> lk(mu);
					cv.wait(lk, [&] { return !ready.empty() || done; });
					if (ready.empty()) return;
					b = std::move(ready.front()); ready.pop_front(); no = next_in++;
					cv.notify_all();
				}
				Done d; d.n_reads = b->n();
				{	/* not more than two finished batches per GPU wait for the writer */
					std::unique_lock<std::mutex> lk(omu);
					ocv.wait(lk, [&] { return failed || no - next_out < 2 * (long)ixs.size() + 2; });
					if (failed) continue;                                 /* drain the reader after a failure */
				}
				d.ok = search_batch(ixs[g], *b, d.bytes);
				std::unique_lock<std::mutex> lk(omu);
				if (!d.ok) failed = true;
				finished.emplace(no, std::move(d));
				ocv.notify_all();
			}
		});
	{	/* the writer: this thread */
		for (;;) {
			Done d;
			{
				std::unique_lock<std::mutex> lk(omu);
				bool readers_done = false;
				ocv.wait_for(lk, std::chrono::milliseconds(50), [&] { return finished.count(next_out) != 0 || failed; });
				if (finished.count(next_out) == 0) {
					if (failed) { status = 2; break; }
					lk.unlock();
					{ std::unique_lock<std::mutex> lk2(mu); readers_done = done && ready.empty() && next_in == next_out; }
					if (readers_done) break;
					continue;
				}
				d = std::move(finished[next_out]); finished.erase(next_out); ++next_out;
				ocv.notify_all();
			}
			if (!d.ok) { status = 2; break; }
			if (fwrite(d.bytes.data(), 1, d.bytes.size(), out) != d.bytes.size()) { perror("[nabwa_aln] write"); status = 2; std::unique_lock<std::mutex> lk(omu); failed = true; ocv.notify_all(); break; }
			tot += d.n_reads;
			fprintf(stderr, "[nabwa_aln] %ld sequences have been processed.\n", tot);
		}
		{ std::unique_lock<std::mutex> lk(omu); if (status) failed = true; ocv.notify_all(); }
	}
	for (auto &w : workers) w.join();
	reader.join();
	if (src.n_tot && opt.trim_qual >= 1) fprintf(stderr, "[nabwa_aln] %.1f%% bases are trimmed.\n", 100.0 * src.n_trimmed / src.n_tot);
	src.fx.close(); bam.close();
	for (nabwa_index_t *p : ixs) nabwa_index_destroy(p);
	if (fflush(out) != 0) status = status ? status : 2;
	if (out != stdout) fclose(out);
	if (status) return status;
	if (ofile) {                                                          /* final_rename (utils.c:159-173): "x.sai_" becomes "x.sai" once complete */
		std::string nf(ofile);
		size_t e = nf.size();
		while (e > 0 && nf[e - 1] == '_') --e;
		if (e > 0 && nf[e - 1] != '/' && e < nf.size()) {
			nf.resize(e);
			fprintf(stderr, "[nabwa_aln] finished, renaming %s to %s.\n", ofile, nf.c_str());
			rename(ofile, nf.c_str());
		}
	}
	fprintf(stderr, "[nabwa_aln] finished cleanly, shutting down.\n");
	return 0;
}
