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
// Not taken over: -b/-0/-1/-2 (BAM input is outside this path) are refused.  One deliberate difference:
// resuming into an existing -f file (attempt_recovery, bwtaln.c:259-296) continues the record stream; the reference
// writes a second copy of the 64-byte header at the resume point (bwtaln.c:387 is unconditional), which makes the
// resumed file unreadable.  A resumed file here equals the file of an uninterrupted run.
#include <ctype.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include <zlib.h>
#include <condition_variable>
#include <deque>
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
	std::vector<unsigned char> buf;
	int have = 0, at = 0, pending = 0;
	bool eof = false;
	std::string name, comment, seq, qual;
	unsigned char cls[256];            /* sequence bytes: 0 = base character (isgraph), 1 = skipped, 2 = ends the sequence ('>' '+' '@') */

	bool open(const char *fn)
	{
		fp = strcmp(fn, "-") == 0 ? gzdopen(fileno(stdin), "r") : gzopen(fn, "r");
		if (fp) gzbuffer(fp, 1 << 20);
		const char *bs = getenv("NABWA_ALN_BUF");             /* (tests: a few bytes, so that every scan meets the end of the buffer) */
		buf.resize(bs && atoi(bs) > 0 ? (size_t)atoi(bs) : (size_t)4 << 20);
		for (int c = 0; c < 256; ++c) cls[c] = isgraph(c) ? 0 : 1;
		cls['>'] = cls['+'] = cls['@'] = 2;
		return fp != nullptr;
	}
	void close() { if (fp) gzclose(fp); fp = nullptr; }
	/* true when buf[at .. have) holds at least one byte */
	bool more()
	{
		if (at < have) return true;
		if (eof) return false;
		have = gzread(fp, buf.data(), (unsigned)buf.size()); at = 0;
		if (have <= 0) { eof = true; have = 0; return false; }
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
				const unsigned char *p = buf.data() + at, *e = buf.data() + have;
				while (p < e && *p != '>' && *p != '@') ++p;
				at = (int)(p - buf.data());
				if (p < e) { ++at; break; }
			}
		}
		pending = 0;
		name.clear(); comment.clear(); seq.clear(); qual.clear();
		for (c = -1;;) {                                     /* name: up to the first white space */
			if (!more()) break;
			const unsigned char *p = buf.data() + at, *e = buf.data() + have, *q = p;
			while (q < e && !isspace(*q)) ++q;
			name.append((const char*)p, (size_t)(q - p));
			at = (int)(q - buf.data());
			if (q < e) { c = *q; ++at; break; }
		}
		if (c == -1 && name.empty()) return -1;
		if (c != '\n' && c != -1)                            /* comment: the rest of the line */
			for (;;) {
				if (!more()) break;
				const unsigned char *p = buf.data() + at, *e = buf.data() + have;
				const unsigned char *q = (const unsigned char*)memchr(p, '\n', (size_t)(e - p));
				comment.append((const char*)p, (size_t)((q ? q : e) - p));
				at = (int)((q ? q + 1 : e) - buf.data());
				if (q) break;
			}
		for (c = -1;;) {                                     /* sequence: runs of base characters up to '>', '+' or '@' */
			if (!more()) break;
			const unsigned char *e = buf.data() + have, *q = buf.data() + at;
			while (q < e) {
				const unsigned char k = cls[*q];
				if (k == 0) { const unsigned char *r = q; do ++q; while (q < e && cls[*q] == 0); seq.append((const char*)r, (size_t)(q - r)); }
				else if (k == 1) ++q;
				else { c = *q; break; }
			}
			at = (int)(q - buf.data());
			if (c != -1) { ++at; break; }
		}
		if (c == '>' || c == '@') pending = c;
		if (c != '+') return (int)seq.size();
		for (;;) {                                           /* the rest of the '+' line */
			if (!more()) return -2;
			const unsigned char *p = buf.data() + at, *e = buf.data() + have;
			const unsigned char *q = (const unsigned char*)memchr(p, '\n', (size_t)(e - p));
			at = (int)((q ? q + 1 : e) - buf.data());
			if (q) break;
		}
		for (;;) {                                           /* quality: characters 33..127 until there is one per base */
			if (qual.size() >= seq.size()) { if (more()) ++at; break; }       /* ... and the character after them goes too */
			if (!more()) break;
			const unsigned char *e = buf.data() + have, *q = buf.data() + at;
			size_t need = seq.size() - qual.size();
			while (q < e && need) {
				const unsigned char *r = q;
				while (q < e && (size_t)(q - r) < need && *q >= 33 && *q <= 127) ++q;
				qual.append((const char*)r, (size_t)(q - r)); need -= (size_t)(q - r);
				if (need && q < e) ++q;                      /* a character that is not a quality (line break): skipped */
			}
			at = (int)(q - buf.data());
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

struct Source {                     /* bwa_read_seq (bwaseqio.c:172-252) minus the bwa_seq_t records */
	Fastx fx;
	int mode, trim_qual;
	long n_trimmed = 0, n_tot = 0;

	/* reads the next record that survives the filters; appends it to b unless b is null (skipping) */
	bool one(Batch *b)
	{
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
	{ 'b', nullptr, 0x20, 0, nullptr, nullptr }, { '0', nullptr, 0x40, 0, nullptr, nullptr },      /* BAM input: accepted by the parser, refused below */
	{ '1', nullptr, 0x80, 0, nullptr, nullptr }, { '2', nullptr, 0x100, 0, nullptr, nullptr },
};

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
	fprintf(stderr, "         (-b -0 -1 -2: BAM input is not available in this tool)\n\n");
	fprintf(stderr, "Environment: NABWA_DEVICE (GPU ordinal, 0), NABWA_ALN_BATCH (reads per GPU batch, 4194304)\n\n");
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
	if (opt.mode & MODE_BAM_ANY) { fprintf(stderr, "[nabwa_aln] BAM input (-b -0 -1 -2) is not available in this tool\n"); return 1; }
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
		Source src;
		src.mode = opt.mode; src.trim_qual = opt.trim_qual;
		if (!src.fx.open(reads)) { fprintf(stderr, "[nabwa_aln] fail to open file '%s'. Abort!\n", reads); return 2; }
		auto fnv = [](uint64_t h, const uint8_t *p, size_t n) { for (size_t i = 0; i < n; ++i) { h ^= p[i]; h *= 1099511628211ull; } return h; };
		uint64_t all = 1469598103934665603ull; long n_reads = 0, n_bases = 0;
		for (;;) {
			Batch b;
			while (b.n() < REF_CHUNK && src.one(&b)) {}
			if (b.n() == 0) break;
			for (int i = 0; i < b.n(); ++i) {
				const size_t o = (size_t)b.off[i], len = (size_t)(b.off[i + 1] - b.off[i]);
				uint64_t h = fnv(fnv(1469598103934665603ull, b.seq.data() + o, len), b.rseq.data() + o, len);
				if (per_read) printf("%zu %016llx\n", len, (unsigned long long)h);
				all = fnv(all, (const uint8_t*)&h, 8); n_bases += (long)len;
			}
			n_reads += b.n();
		}
		printf("reads %ld bases %ld fnv %016llx\n", n_reads, n_bases, (unsigned long long)all);
		return 0;
	}

	// ---- the index: no GPU, no output
	nabwa_index_t *ix = nullptr;
	const int device = getenv("NABWA_DEVICE") ? atoi(getenv("NABWA_DEVICE")) : 0;
	const std::string sa_path = std::string(prefix) + ".sa", rsa_path = std::string(prefix) + ".rsa";
	const int with_sa = access(sa_path.c_str(), R_OK) == 0 && access(rsa_path.c_str(), R_OK) == 0;   /* optional: lets the library build its text-mode companions */
	if (nabwa_index_load(prefix, device, with_sa, 0, &ix) != NABWA_OK) {
		fprintf(stderr, "[nabwa_aln] cannot set up the index on GPU %d: %s\n", device, nabwa_last_error());
		return 2;
	}

	FILE *out = stdout;
	if (ofile) {
		out = fopen(ofile, resume.found ? "rb+" : "wb");
		if (!out) { fprintf(stderr, "[nabwa_aln] fail to open file '%s': ", ofile); perror(nullptr); return 2; }
		if (resume.found && fseek(out, resume.at, SEEK_SET) != 0) { fprintf(stderr, "[nabwa_aln] seek failed, aborting.\n"); return 2; }
	}
	if (!resume.found && fwrite(&opt, sizeof(opt), 1, out) != 1) { perror("[nabwa_aln] write"); return 2; }

	Source src;
	src.mode = opt.mode; src.trim_qual = opt.trim_qual;
	if (!src.fx.open(reads)) { fprintf(stderr, "[nabwa_aln] fail to open file '%s'. Abort!\n", reads); return 2; }
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
		for (bool eof = false; !eof; ) {
			std::unique_ptr<Batch> b(new Batch);
			while (b->n() < batch_reads) {
				if (!src.one(b.get())) { eof = true; break; }
				if (b->seq.size() >= (1ull << 30) && b->n() % REF_CHUNK == 0) break;   /* long reads: bound the bases of a batch as well */
			}
			std::unique_lock<std::mutex> lk(mu);
			cv.wait(lk, [&] { return ready.size() < 2; });
			if (b->n()) ready.push_back(std::move(b));
			if (eof) done = true;
			cv.notify_all();
		}
	});

	long tot = 0; int status = 0;
	std::vector<int32_t> n_aln, max_entries;
	std::vector<nabwa_aln1_t> rows;
	std::vector<char> obuf;
	for (;;) {
		std::unique_ptr<Batch> b;
		{
			std::unique_lock<std::mutex> lk(mu);
			cv.wait(lk, [&] { return !ready.empty() || done; });
			if (ready.empty()) break;
			b = std::move(ready.front()); ready.pop_front();
			cv.notify_all();
		}
		if (status) continue;                                             /* drain the reader after a failure */
		// runs of chunks with the same max_gapo clamp -> one GPU call each
		const int n_chunks = (int)b->chunk_max_len.size();
		auto clamp_of = [&](int ch) {
			const int md = opt.fnr > 0.0f ? nabwa_cal_maxdiff(b->chunk_max_len[ch], 0.02, opt.fnr) : opt.max_diff;
			return md < opt.max_gapo ? md : opt.max_gapo;
		};
		for (int c0 = 0; c0 < n_chunks && !status; ) {
			int c1 = c0 + 1;
			while (c1 < n_chunks && clamp_of(c1) == clamp_of(c0)) ++c1;
			const int r0 = c0 * REF_CHUNK, r1 = c1 * REF_CHUNK < b->n() ? c1 * REF_CHUNK : b->n(), n = r1 - r0;
			std::vector<int64_t> off(n + 1);
			const int64_t base = b->off[r0];
			for (int i = 0; i <= n; ++i) off[i] = b->off[r0 + i] - base;
			n_aln.resize(n); max_entries.resize(n);
			int64_t cap = (int64_t)n + n / 4 + 1024, n_rows = 0;
			int rc;
			for (;;) {
				rows.resize(cap);
				rc = nabwa_cal_sa_reg_gap(ix, &opt, n, off.data(), b->seq.data() + base, b->rseq.data() + base, 0,
										  n_aln.data(), rows.data(), cap, &n_rows, max_entries.data());
				if (rc != NABWA_ECAP) break;
				cap = 0; for (int i = 0; i < n; ++i) cap += n_aln[i];
			}
			if (rc != NABWA_OK) { fprintf(stderr, "[nabwa_aln] GPU search failed: %s\n", nabwa_last_error()); status = 2; break; }
			// the record stream: n_aln, then the rows (bwtaln.c:242-246)
			obuf.resize((size_t)n * 4 + (size_t)n_rows * sizeof(nabwa_aln1_t));
			char *w = obuf.data(); const nabwa_aln1_t *r = rows.data();
			for (int i = 0; i < n; ++i) {
				memcpy(w, &n_aln[i], 4); w += 4;
				memcpy(w, r, (size_t)n_aln[i] * sizeof(nabwa_aln1_t)); w += (size_t)n_aln[i] * sizeof(nabwa_aln1_t); r += n_aln[i];
			}
			if (fwrite(obuf.data(), 1, obuf.size(), out) != obuf.size()) { perror("[nabwa_aln] write"); status = 2; break; }
			tot += n;
			fprintf(stderr, "[nabwa_aln] %ld sequences have been processed.\n", tot);
			c0 = c1;
		}
	}
	reader.join();
	if (src.n_tot && opt.trim_qual >= 1) fprintf(stderr, "[nabwa_aln] %.1f%% bases are trimmed.\n", 100.0 * src.n_trimmed / src.n_tot);
	src.fx.close();
	nabwa_index_destroy(ix);
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
