// bam2bam_main.cpp -- `nabwa_bam2bam`: the command line of `bwa bam2bam -t 1` (bam2bam.c:1942-2098) on top of libnabwa.so.
//
//     nabwa_bam2bam -g PREFIX [alignment options of bwa bam2bam] [-f out.bam] in.bam
//
// BAM in (BGZF, any other gzip stream or none, as the reference's bamlite reads it; bgzf_in.hpp) -> both passes of the reference's sequential loop
// (bam2bam.c:1143-1216) through the batch front-end of the library (nabwa_bam_batch_*, bam_batch.hip) -> BGZF BAM out with
// the header bwa_print_bam_header writes (@HD VN:1.4, a new @PG chained to the old one, @SQ from the .ann file, the other old
// lines kept; bam2bam.c:164-301).  Host code only; the GPU work is the library's.  Not provided: the 0MQ master / worker modes
// (-p, `bwa worker`: libzmq is absent from the build image) and resuming from .sai files (-0 -1 -2) -- each is refused, none is
// silently ignored.  --only-aligned, --drop-aligned, --skip-duplicates, --broken-input and --debug-bam are the library's NABWA_BAM_* flags.
// -t is accepted and ignored (NABWA_DEVICES=0,1,... names the GPUs: an index replica on each, batches dealt to them in turn, searched as
// they come and passed in input order; default one GPU, NABWA_DEVICE or 0), --temp-dir likewise (the records wait in memory between the passes).
#include <getopt.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>
#include <sys/time.h>
#include <unistd.h>
#include <condition_variable>
#include <deque>
#include <memory>
#include <mutex>
#include <set>
#include <string>
#include <thread>
#include <vector>
#include "../../include/nabwa.h"

static const char *VERSION = "0.5.10-evan.6.3+nabwa";

/* any thread of the pipeline may end the run: no exit handlers (they would tear the GPU runtime down under the other threads) */
static void die(const char *what, const char *why) { fprintf(stderr, "[nabwa_bam2bam] %s: %s\n", what, why); fflush(stderr); _exit(1); }

/* ---------------------------------------------------------------- BGZF out (bgzf.c: blocks of <= 0xff00 input bytes, level 2) */
static void bgzf_block(const uint8_t *in, size_t n, int level, std::vector<uint8_t> &out)
{
	uint8_t buf[0x10000 + 64];
	z_stream zs; memset(&zs, 0, sizeof(zs));
	zs.next_in = (Bytef*)in; zs.avail_in = (uInt)n; zs.next_out = buf + 18; zs.avail_out = sizeof(buf) - 18 - 8;
	if (deflateInit2(&zs, level, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK || deflate(&zs, Z_FINISH) != Z_STREAM_END) die("BGZF", "deflate failed");
	const size_t clen = zs.total_out; deflateEnd(&zs);
	static const uint8_t hdr[16] = { 31, 139, 8, 4, 0, 0, 0, 0, 0, 255, 6, 0, 'B', 'C', 2, 0 };
	memcpy(buf, hdr, 16);
	const uint16_t bsize = (uint16_t)(clen + 25);
	buf[16] = (uint8_t)bsize; buf[17] = (uint8_t)(bsize >> 8);
	const uint32_t crc = (uint32_t)crc32(crc32(0, 0, 0), in, (uInt)n), isz = (uint32_t)n;
	uint8_t *t = buf + 18 + clen;
	memcpy(t, &crc, 4); memcpy(t + 4, &isz, 4);
	out.insert(out.end(), buf, buf + 18 + clen + 8);
}

struct BgzfOut {
	FILE *f; std::vector<uint8_t> pend; int level;
	void write(const void *p, size_t n) { const uint8_t *b = (const uint8_t*)p; pend.insert(pend.end(), b, b + n); if (pend.size() >= (64u << 20)) flush(false); }
	void flush(bool all)
	{
		const size_t BS = 0xff00;
		const size_t n_full = pend.size() / BS, n_blocks = all ? (pend.size() + BS - 1) / BS : n_full;
		if (!n_blocks) return;
		int nt = (int)std::thread::hardware_concurrency(); if (nt < 1) nt = 1; if (nt > 16) nt = 16;
		if ((size_t)nt > n_blocks) nt = (int)n_blocks;
		std::vector<std::vector<uint8_t>> parts(nt);
		std::vector<std::thread> th;
		for (int t = 0; t < nt; ++t) th.emplace_back([&, t]() {
			for (size_t k = n_blocks * t / nt; k < n_blocks * (t + 1) / nt; ++k) {
				const size_t o = k * BS, m = pend.size() - o < BS ? pend.size() - o : BS;
				bgzf_block(pend.data() + o, m, level, parts[t]);
			}
		});
		for (auto &x : th) x.join();
		for (auto &p : parts) if (!p.empty() && fwrite(p.data(), 1, p.size(), f) != p.size()) die("output", "write failed");
		const size_t done = n_blocks * BS < pend.size() ? n_blocks * BS : pend.size();
		pend.erase(pend.begin(), pend.begin() + done);
	}
	void close()
	{
		flush(true);
		static const uint8_t eof[28] = { 31, 139, 8, 4, 0, 0, 0, 0, 0, 255, 6, 0, 66, 67, 2, 0, 27, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0 };
		if (fwrite(eof, 1, 28, f) != 28 || fflush(f) != 0) die("output", "write failed");
		if (f != stdout) fclose(f);
	}
};

#include "bgzf_in.hpp"
#include "bam_header.hpp"

/* a bounded queue between two threads */
template <class T> struct Chan {
	std::mutex m; std::condition_variable cv; std::deque<T> q; size_t cap; bool closed;
	explicit Chan(size_t cap_) : cap(cap_), closed(false) {}
	void put(T &&x) { std::unique_lock<std::mutex> l(m); cv.wait(l, [&] { return q.size() < cap; }); q.push_back(std::move(x)); cv.notify_all(); }
	bool get(T &x) { std::unique_lock<std::mutex> l(m); cv.wait(l, [&] { return !q.empty() || closed; }); if (q.empty()) return false; x = std::move(q.front()); q.pop_front(); cv.notify_all(); return true; }
	void close() { std::unique_lock<std::mutex> l(m); closed = true; cv.notify_all(); }
};

/* ---------------------------------------------------------------- the header (bam2bam.c:164-301); find_pp_tag: bam_header.hpp */
static std::string header_text(nabwa_index_t *ix, const std::string &old, int argc, char **argv)
{
	std::string pp, id; bool has_pp;
	find_pp_tag(old, pp, id, has_pp);
	std::string t = "@HD\tVN:1.4\n@PG\tID:" + id + (has_pp ? "\tPP:" + pp : "") + "\tPN:bwa\tVN:" + VERSION + (argc ? "\tCL:" : "");
	for (int i = 0; i < argc; ++i) { t += argv[i]; t += i == argc - 1 ? '\n' : ' '; }
	const int ns = nabwa_index_n_contigs(ix);
	for (int i = 0; i < ns; ++i) { char name[1024]; int64_t off; int32_t len; nabwa_index_contig(ix, i, name, sizeof name, &off, &len); t += std::string("@SQ\tSN:") + name + "\tLN:" + std::to_string(len) + "\n"; }
	size_t p = 0;
	while (p < old.size() && old[p]) {
		size_t e = old.find('\n', p); if (e == std::string::npos) e = old.size();
		const bool boring = e - p >= 3 && old[p] == '@' && ((old[p + 1] == 'S' && old[p + 2] == 'Q') || (old[p + 1] == 'H' && old[p + 2] == 'D'));
		if (!boring) { t.append(old, p, e - p); t += '\n'; }
		p = e + 1;
	}
	return t;
}

/* ---------------------------------------------------------------- the temporary file between the passes (bam2bam.c:1099-1135, 1733-1758)
 * With --temp-dir a batch that has to wait for the insert-size estimates does not wait in memory: its records leave as the reference's
 * temporary file holds them -- u32 length + the message of msg_init_from_pair, positioned (or finished, for a batch of single reads
 * that only waits for its turn) -- and come back batch by batch for pass 2.  (Plain, not gzip: one core of deflate would pace the file.) */
struct Spilled { long at; int n_logical, n_records; bool finished; size_t dev; };
static void spill_batch(FILE *f, nabwa_bam_batch_t *b, bool finished, Spilled &S)
{
	int nr = 0, nl = 0; nabwa_bam_batch_counts(b, &nr, &nl);
	std::vector<uint8_t> kinds((size_t)(nl ? nl : 1));
	nabwa_bam_batch_kinds(b, kinds.data());
	int64_t nb = 0; std::vector<int64_t> oo((size_t)nr + 1, 0);
	nabwa_bam_batch_output(b, 0, 0, oo.data(), &nb);
	std::vector<uint8_t> ob((size_t)(nb ? nb : 1));
	if (nabwa_bam_batch_output(b, ob.data(), nb, oo.data(), &nb) != NABWA_OK) die("temporary file", nabwa_last_error());
	std::vector<nabwa_wire_read_t> st((size_t)(nr ? nr : 1));
	memset(st.data(), 0, sizeof(nabwa_wire_read_t) * st.size());
	if (!finished && nabwa_bam_batch_positioned(b, st.data()) != NABWA_OK) die("temporary file", nabwa_last_error());
	S.at = ftell(f); S.n_logical = nl; S.n_records = nr; S.finished = finished;
	std::vector<uint8_t> msg;
	int at = 0;
	for (int k = 0; k < nl; ++k) {
		nabwa_wire_rec_t r; memset(&r, 0, sizeof r);
		r.recno = (uint64_t)k; r.kind = kinds[(size_t)k]; r.phase = finished ? NABWA_PHASE_FINISHED : NABWA_PHASE_POSITIONED;
		for (int e = 0; e < r.kind; ++e, ++at) {
			r.read[e] = st[(size_t)at];
			const uint8_t *rec = ob.data() + oo[(size_t)at];
			nabwa_wire_core_from_bam(rec + 4, r.read[e].core);
			r.read[e].data = rec + 36; r.read[e].data_len = (int32_t)(oo[(size_t)at + 1] - oo[(size_t)at] - 36);
		}
		const int64_t need = nabwa_wire_size(&r);
		msg.resize((size_t)need + 4);
		const uint32_t len = (uint32_t)need;
		memcpy(msg.data(), &len, 4);
		if (nabwa_wire_encode(&r, msg.data() + 4, need) != need || fwrite(msg.data(), 1, msg.size(), f) != msg.size()) die("temporary file", "cannot write");
	}
}
/* one batch back: its records as a BAM stream and, unless it was finished, the state pass 1 left */
static void unspill_batch(FILE *f, const Spilled &S, std::vector<uint8_t> &stream, std::vector<int64_t> &off, std::vector<nabwa_wire_read_t> &st, std::vector<std::vector<uint8_t>> &keep)
{
	if (fseek(f, S.at, SEEK_SET) != 0) die("temporary file", "cannot seek");
	stream.clear(); off.assign(1, 0); st.clear(); keep.clear();
	for (int k = 0; k < S.n_logical; ++k) {
		uint32_t len = 0;
		if (fread(&len, 4, 1, f) != 1) die("temporary file", "truncated");
		keep.emplace_back((size_t)len);
		if (len && fread(keep.back().data(), 1, len, f) != len) die("temporary file", "truncated");
		nabwa_wire_rec_t r;
		if (nabwa_wire_decode(keep.back().data(), (int64_t)len, &r) != NABWA_OK) die("temporary file", nabwa_last_error());
		for (int e = 0; e < r.kind; ++e) {
			const nabwa_wire_read_t &x = r.read[e];
			const uint32_t bs = 32u + (uint32_t)x.data_len;
			const size_t at = stream.size();
			stream.resize(at + 4 + bs);
			memcpy(&stream[at], &bs, 4);
			nabwa_wire_core_to_bam(x.core, &stream[at + 4]);
			if (x.data_len) memcpy(&stream[at + 36], x.data, (size_t)x.data_len);
			off.push_back((int64_t)stream.size());
			st.push_back(x);
		}
	}
	if ((int)st.size() != S.n_records) die("temporary file", "a batch came back with another number of records");
}

int main(int argc, char **argv)
{
	const double t_main = now_s();
	uint32_t rec_flags = 0;                                    /* NABWA_BAM_*: --only-aligned, --drop-aligned, --debug-bam, --broken-input, --skip-duplicates */
	static struct option longopts[] = {
		{ "num-diff", 1, 0, 'n' }, { "max-gap-open", 1, 0, 'o' }, { "max-gap-extensions", 1, 0, 'e' }, { "indel-near-end", 1, 0, 'i' },
		{ "deletion-occurences", 1, 0, 'd' }, { "seed-length", 1, 0, 'l' }, { "seed-mismatches", 1, 0, 'k' }, { "queue-size", 1, 0, 'm' },
		{ "num-threads", 1, 0, 't' }, { "mismatch-penalty", 1, 0, 'M' }, { "gap-open-penalty", 1, 0, 'O' }, { "gap-extension-penalty", 1, 0, 'E' },
		{ "max-best-hits", 1, 0, 'R' }, { "trim-quality", 1, 0, 'q' }, { "log-gap-penalty", 0, 0, 'L' }, { "non-iterative", 0, 0, 'N' },
		{ "output", 1, 0, 'f' }, { "genome", 1, 0, 'g' }, { "only-aligned", 0, 0, 128 }, { "drop-aligned", 0, 0, 133 }, { "debug-bam", 0, 0, 129 },
		{ "broken-input", 0, 0, 130 }, { "skip-duplicates", 0, 0, 131 }, { "temp-dir", 1, 0, 132 }, { "max-insert-size", 1, 0, 'a' },
		{ "max-occurences", 1, 0, 'C' }, { "max-occurences-se", 1, 0, 'D' }, { "max-hits", 1, 0, 'h' }, { "max-discordant-hits", 1, 0, 'H' },
		{ "chimeric-rate", 1, 0, 'c' }, { "disable-sw", 0, 0, 's' }, { "disable-isize-estimate", 0, 0, 'A' }, { "listen-port", 1, 0, 'p' }, { 0, 0, 0, 0 } };
	nabwa_gap_opt_t go; nabwa_gap_init_opt(&go);
	nabwa_pe_opt_t po; nabwa_pe_opt_default(&po);
	const char *prefix = 0, *ofile = 0, *temp_dir = 0; int c, opte = -1;
	while ((c = getopt_long(argc, argv, "g:n:o:e:i:d:l:k:LR:m:t:NM:O:E:q:f:C:D:a:sc:h:H:Ap:0:1:2:", longopts, 0)) >= 0) {
		switch (c) {
			case 'g': prefix = optarg; break;
			case 'n': if (strstr(optarg, ".")) { go.fnr = (float)atof(optarg); go.max_diff = -1; } else { go.max_diff = atoi(optarg); go.fnr = -1.0f; } break;
			case 'o': go.max_gapo = atoi(optarg); break;
			case 'e': opte = atoi(optarg); break;
			case 'M': go.s_mm = atoi(optarg); break;
			case 'O': go.s_gapo = atoi(optarg); break;
			case 'E': go.s_gape = atoi(optarg); break;
			case 'd': go.max_del_occ = atoi(optarg); break;
			case 'i': go.indel_end_skip = atoi(optarg); break;
			case 'l': go.seed_len = atoi(optarg); break;
			case 'k': go.max_seed_diff = atoi(optarg); break;
			case 'm': go.max_entries = atoi(optarg); break;
			case 't': go.n_threads = atoi(optarg); break;
			case 'L': go.mode |= NABWA_MODE_LOGGAP; break;
			case 'R': go.max_top2 = atoi(optarg); break;
			case 'q': go.trim_qual = atoi(optarg); break;
			case 'N': go.mode |= NABWA_MODE_NONSTOP; go.max_top2 = 0x7fffffff; break;
			case 'f': ofile = optarg; break;
			case 'C': po.max_occ = atoi(optarg); break;
			case 'D': po.max_occ_se = atoi(optarg); break;
			case 'a': po.max_isize = atoi(optarg); break;
			case 's': po.is_sw = 0; break;
			case 'c': po.ap_prior = atof(optarg); break;
			case 'A': po.force_isize = 1; break;
			case 'h': po.n_multi = atoi(optarg); break;
			case 'H': po.N_multi = atoi(optarg); break;
			case 132: temp_dir = optarg; break;
			case 128: rec_flags |= NABWA_BAM_ONLY_ALIGNED; break;
			case 129: rec_flags |= NABWA_BAM_DEBUG; break;
			case 130: rec_flags |= NABWA_BAM_BROKEN_INPUT; break;
			case 131: rec_flags |= NABWA_BAM_SKIP_DUPLICATES; break;
			case 133: rec_flags |= NABWA_BAM_DROP_ALIGNED; break;
			case 'p': case '0': case '1': case '2':
				fprintf(stderr, "[nabwa_bam2bam] this option of bwa bam2bam is not provided (0MQ modes, .sai resume)\n");
				return 1;
			default: return 1;
		}
	}
	if (opte > 0) { go.max_gape = opte; go.mode &= ~NABWA_MODE_GAPE; }
	/* the library's limits, said before the index is loaded */
	if (po.max_occ_se < 0 || po.max_occ_se > NABWA_MAX_MULTI - 1) { fprintf(stderr, "[nabwa_bam2bam] -D %d: at most %d other hits of a single read are listed\n", po.max_occ_se, NABWA_MAX_MULTI - 1); return 1; }
	if (po.n_multi < 0 || po.n_multi > NABWA_MAX_MULTI || po.N_multi < 0 || po.N_multi > NABWA_MAX_MULTI) { fprintf(stderr, "[nabwa_bam2bam] -h / -H: 0..%d\n", NABWA_MAX_MULTI); return 1; }
	if (go.s_mm < 1 || go.s_gapo < 1 || go.s_gape < 1) { fprintf(stderr, "[nabwa_bam2bam] -M / -O / -E must be at least 1\n"); return 1; }
	if (optind + 1 > argc || !prefix) {
		fprintf(stderr, "\nUsage:   nabwa_bam2bam -g PREFIX [options of bwa bam2bam] [-f out.bam] <in.bam>\n\n");
		return 1;
	}
	/* one index replica per GPU of NABWA_DEVICES ("0,1,2,3"; default: NABWA_DEVICE or 0); batches are dealt to them in turn */
	std::vector<int> devices;
	if (getenv("NABWA_DEVICES"))
		for (const char *q = getenv("NABWA_DEVICES"); *q; ) { char *e; const long d = strtol(q, &e, 10); if (e == q) break; devices.push_back((int)d); q = *e == ',' ? e + 1 : e; }
	if (devices.empty()) devices.push_back(getenv("NABWA_DEVICE") ? atoi(getenv("NABWA_DEVICE")) : 0);
	std::vector<nabwa_index_t*> ixs(devices.size(), (nabwa_index_t*)0);
	{
		std::vector<std::string> err(devices.size());
		std::vector<std::thread> th;
		for (size_t g = 0; g < devices.size(); ++g)
			th.emplace_back([&, g]() { if (nabwa_index_load(prefix, devices[g], 1, 1, &ixs[g]) != NABWA_OK) { err[g] = nabwa_last_error(); ixs[g] = 0; } });
		for (auto &x : th) x.join();
		for (size_t g = 0; g < devices.size(); ++g) if (!ixs[g]) die("genome index", err[g].c_str());
	}
	nabwa_index_t *ix = ixs[0];
	int64_t genome_len = 0; uint32_t seed = 0;
	nabwa_index_reference_info(ix, &genome_len, &seed);
	fprintf(stderr, "[nabwa_bam2bam] genome length is %ld\n", (long)genome_len);

	/* ---- input: magic, header text, reference list (bamlite.c: bam_header_read) */
	FILE *inf = strcmp(argv[optind], "-") ? fopen(argv[optind], "rb") : stdin;
	if (!inf) die(argv[optind], "cannot open");
	BamIn in(inf, argv[optind]);
	auto rd = [&](void *p, size_t n) -> bool { return in.read(p, n); };
	char magic[4]; int32_t l_text = 0, n_ref = 0;
	if (!rd(magic, 4) || memcmp(magic, "BAM\1", 4) || !rd(&l_text, 4) || l_text < 0) die(argv[optind], "not a BAM file");
	std::string old(l_text, '\0');
	if (l_text && !rd(&old[0], l_text)) die(argv[optind], "truncated header");
	old.resize(strlen(old.c_str()));
	if (!rd(&n_ref, 4)) die(argv[optind], "truncated header");
	for (int i = 0; i < n_ref; ++i) { int32_t ln, tl; if (!rd(&ln, 4) || ln < 0) die(argv[optind], "truncated header"); std::vector<char> nm(ln); if (!rd(nm.data(), ln) || !rd(&tl, 4)) die(argv[optind], "truncated header"); }

	FILE *of = ofile ? fopen(ofile, "wb") : stdout;
	if (!of) die(ofile, "cannot create");
	BgzfOut out{ of, {}, 2 };
	{
		const std::string text = header_text(ix, old, argc, argv);
		const int32_t hl = (int32_t)text.size(), ns = nabwa_index_n_contigs(ix);
		out.write("BAM\1", 4); out.write(&hl, 4); out.write(text.data(), text.size()); out.write(&ns, 4);
		for (int i = 0; i < ns; ++i) { char name[1024]; int64_t off; int32_t len; nabwa_index_contig(ix, i, name, sizeof name, &off, &len); const int32_t nl = (int32_t)strlen(name) + 1; out.write(&nl, 4); out.write(name, nl); out.write(&len, 4); }
	}

	/* ---- a pipeline of threads: one reads and inflates the input and cuts it into batches of records (mates stay together), one
	 * parses them (create), one per GPU searches them, this one runs pass 1 over the batches in input order (and pass 2 at once
	 * for a batch of singletons, which is written at once unless pairs came before it), one collects the output records, one
	 * deflates and writes */
	const bool timing = getenv("NABWA_TIMING") != 0;
	const double t_loop = now_s();
	nabwa_isize_table_t *tab = nabwa_isize_table_create(po.ap_prior, genome_len);
	uint64_t rng = ((uint64_t)seed << 16) | 0x330E;            /* srand48(bns->seed), bam2bam.c:1745 */
	const long BATCH = getenv("NABWA_BAM_BATCH") ? atol(getenv("NABWA_BAM_BATCH")) : (1L << 20);
	struct InBatch { std::vector<uint8_t> buf; std::vector<int64_t> off; };
	Chan<InBatch> in_ch(2);
	struct OutBytes { std::unique_ptr<uint8_t[]> p; size_t n; };      /* no zero fill: the library writes every byte */
	Chan<OutBytes> out_ch(2);
	double t_read = 0, t_write = 0, t_wait_in = 0, t_lib = 0, t_wait_out = 0, t_call[5] = { 0, 0, 0, 0, 0 };      /* create, pass 1, pass 2, output, destroy */
	std::thread reader([&]() {
		const double t0 = now_s();
		InBatch cur; cur.off.assign(1, 0);
		bool hold_mate = false;                                   /* the last record is a paired read that waits for the record after it */
		size_t held_at = 0;                                       /* where it starts in buf */
		double t_blocked = 0;
		auto hand_over = [&]() {
			if (cur.off.size() > 1) { const double tb = now_s(); in_ch.put(std::move(cur)); t_blocked += now_s() - tb; }
			cur = InBatch(); cur.off.assign(1, 0);
		};
		for (;;) {
			if (in.need(4) == 0) break;
			uint32_t bs = 0;
			if (!in.read(&bs, 4) || bs < 32) die(argv[optind], "truncated record");
			std::vector<uint8_t> &buf = cur.buf;
			const size_t at = buf.size();
			if (buf.capacity() < at + 4 + bs) buf.reserve(buf.capacity() ? 2 * buf.capacity() + 4 + bs : (size_t)256 << 20);
			buf.resize(at + 4 + bs); memcpy(&buf[at], &bs, 4);
			if (!in.read(&buf[at + 4], bs)) die(argv[optind], "truncated record");
			uint32_t z; memcpy(&z, &buf[at + 16], 4);
			const bool paired = (z >> 16) & 1;
			{	/* the name is read as a C string below and by the library: it must lie, terminated, inside the record */
				const uint32_t l_qname = buf[at + 12];
				if (l_qname == 0 || bs < 32 + l_qname || buf[at + 36 + l_qname - 1] != 0) die(argv[optind], "damaged record (read name not terminated inside the record)");
			}
			cur.off.push_back((int64_t)buf.size());
			/* read_bam_pair_core's view of the stream (bwaseqio.c:345-410): a paired read takes the next record as its mate if the names
			 * agree; if they do not it is a lone mate (an error, or dropped with --broken-input) and the next record starts afresh */
			const bool mates = hold_mate && !strcmp((const char*)&buf[held_at + 36], (const char*)&buf[at + 36]);
			hold_mate = mates ? false : paired;
			held_at = at;
			if ((long)cur.off.size() - 1 >= BATCH && !hold_mate) hand_over();
		}
		hand_over();
		in_ch.close();
		t_read = now_s() - t0 - t_blocked;
	});
	std::thread writer([&]() {
		OutBytes o;
		while (out_ch.get(o)) { const double t0 = now_s(); out.write(o.p.get(), o.n); o.p.reset(); t_write += now_s() - t0; }
	});
	/* create (host work only) runs a batch ahead of the passes, output + destroy (host work only) a batch behind: everything that
	 * touches the GPU or draws random numbers stays on this thread, in input order */
	const size_t n_dev = ixs.size();
	Chan<nabwa_bam_batch_t*> done_ch(1);
	std::vector<std::unique_ptr<Chan<nabwa_bam_batch_t*>>> made_ch, found_ch;       /* per GPU: parsed batches, searched batches */
	for (size_t g = 0; g < n_dev; ++g) { made_ch.emplace_back(new Chan<nabwa_bam_batch_t*>(1)); found_ch.emplace_back(new Chan<nabwa_bam_batch_t*>(1)); }
	std::thread creator([&]() {
		InBatch ib;
		for (size_t k = 0; in_ch.get(ib); ++k) {
			const double t0 = now_s();
			nabwa_bam_batch_t *b = 0;
			if (nabwa_bam_batch_create_ex(ixs[k % n_dev], &go, &po, rec_flags, (int)ib.off.size() - 1, ib.buf.data(), ib.off.data(), &b) != NABWA_OK) die("input records", nabwa_last_error());
			std::vector<uint8_t>().swap(ib.buf);
			t_call[0] += now_s() - t0;
			made_ch[k % n_dev]->put(std::move(b));
		}
		for (auto &c : made_ch) c->close();
	});
	/* one thread per GPU searches that GPU's batches as they come (no random numbers, no order: the kernels of batch k + 1 run
	 * while batch k is positioned and finished); the main thread takes the searched batches in input order */
	std::vector<double> t_search(n_dev, 0.0);
	std::vector<std::thread> searchers;
	for (size_t g = 0; g < n_dev; ++g)
		searchers.emplace_back([&, g]() {
			nabwa_bam_batch_t *b;
			while (made_ch[g]->get(b)) {
				const double t0 = now_s();
				if (nabwa_bam_batch_search(b) != NABWA_OK) die("search", nabwa_last_error());
				t_search[g] += now_s() - t0;
				found_ch[g]->put(std::move(b));
			}
			found_ch[g]->close();
		});
	auto emit = [&](nabwa_bam_batch_t *b) {
		int64_t nb = 0;
		const double ta = now_s();
		nabwa_bam_batch_output(b, 0, 0, 0, &nb);
		OutBytes o; o.p.reset(new uint8_t[(size_t)(nb ? nb : 1)]); o.n = (size_t)nb;
		if (nabwa_bam_batch_output(b, o.p.get(), nb, 0, &nb) != NABWA_OK) die("output", nabwa_last_error());
		const double tb = now_s();
		nabwa_bam_batch_destroy(b);
		const double t0 = now_s();
		t_call[3] += tb - ta; t_call[4] += t0 - tb;
		out_ch.put(std::move(o));
		t_wait_out += now_s() - t0;
	};
	std::thread finisher([&]() { nabwa_bam_batch_t *b; while (done_ch.get(b)) emit(b); });
	std::vector<nabwa_bam_batch_t*> waiting;
	/* --temp-dir: what waits for pass 2 waits in a file there (made with mkstemp and unlinked at once, as the reference's is at exit) */
	FILE *spill = 0; std::vector<Spilled> spilled;
	if (temp_dir) {
		std::string tn = std::string(temp_dir) + "/nabwa_bam2bam_XXXXXX";
		const int fd = mkstemp(&tn[0]);
		if (fd < 0 || !(spill = fdopen(fd, "w+b"))) die(temp_dir, "cannot create a temporary file there");
		unlink(tn.c_str());
	}
	uint64_t n_tot[2] = { 0, 0 }, n_mapped[2] = { 0, 0 };
	long tot_seqs = 0; bool any_pairs = false;
	for (size_t k = 0; ; ++k) {
		nabwa_bam_batch_t *b = 0;
		const double t0 = now_s();
		if (!found_ch[k % n_dev]->get(b)) break;
		const double t1 = now_s();
		t_wait_in += t1 - t0;
		if (nabwa_bam_batch_pass1(b, &rng, tab) != NABWA_OK) die("pass 1", nabwa_last_error());
		const double td = now_s();
		t_call[1] += td - t1;
		int nr = 0, nl = 0; nabwa_bam_batch_counts(b, &nr, &nl);
		any_pairs |= nr != nl;
		tot_seqs += nr;
		fprintf(stderr, "[nabwa_bam2bam] pass 1: %ld sequences processed\n", tot_seqs);
		/* a batch of singletons needs no insert-size estimate: it is finished now, and written now unless pairs came before it */
		if (nr == nl && nabwa_bam_batch_pass2(b, tab, n_tot, n_mapped) != NABWA_OK) die("pass 2", nabwa_last_error());
		const double t2 = now_s();
		if (nr == nl) t_call[2] += t2 - td;
		if (nr == nl && !any_pairs) done_ch.put(std::move(b));
		else if (spill) { Spilled S; S.dev = k % n_dev; spill_batch(spill, b, nr == nl, S); spilled.push_back(S); nabwa_bam_batch_destroy(b); }
		else waiting.push_back(b);
		t_lib += t2 - t1;
	}
	reader.join(); creator.join();
	for (auto &x : searchers) x.join();
	if (inf != stdin) fclose(inf);
	/* ---- the barrier (infer_all_isizes), then pass 2 in input order; the output thread collects a batch while the next is finished */
	nabwa_isize_table_infer_all(tab);
	for (nabwa_bam_batch_t *b : waiting) {
		const double t1 = now_s();
		int nr = 0, nl = 0; nabwa_bam_batch_counts(b, &nr, &nl);
		if (nr != nl && nabwa_bam_batch_pass2(b, tab, n_tot, n_mapped) != NABWA_OK) die("pass 2", nabwa_last_error());
		t_lib += now_s() - t1; t_call[2] += now_s() - t1;
		done_ch.put(std::move(b));
	}
	if (spill) {
		std::vector<uint8_t> stream; std::vector<int64_t> off; std::vector<nabwa_wire_read_t> st; std::vector<std::vector<uint8_t>> keep;
		for (const Spilled &S : spilled) {
			const double t1 = now_s();
			unspill_batch(spill, S, stream, off, st, keep);
			if (S.finished) {                                     /* single reads that only waited for their turn: their records as they are */
				std::vector<int64_t> pick;
				size_t nb = 0;
				for (size_t i = 0; i + 1 < off.size(); ++i) {     /* --only-aligned (pair_print_bam, bam2bam.c:911-925) on the finished records */
					uint32_t z; memcpy(&z, &stream[(size_t)off[i] + 16], 4);
					if ((rec_flags & NABWA_BAM_ONLY_ALIGNED) && ((z >> 16) & 4)) continue;
					pick.push_back((int64_t)i); nb += (size_t)(off[i + 1] - off[i]);
				}
				OutBytes o; o.p.reset(new uint8_t[nb ? nb : 1]); o.n = nb;
				size_t w = 0;
				for (int64_t i : pick) { memcpy(o.p.get() + w, &stream[(size_t)off[(size_t)i]], (size_t)(off[(size_t)i + 1] - off[(size_t)i])); w += (size_t)(off[(size_t)i + 1] - off[(size_t)i]); }
				out_ch.put(std::move(o));
				continue;
			}
			nabwa_bam_batch_t *b = 0;
			if (nabwa_bam_batch_create_ex(ixs[S.dev], &go, &po, rec_flags & ~(uint32_t)(NABWA_BAM_BROKEN_INPUT | NABWA_BAM_DROP_ALIGNED), S.n_records, stream.data(), off.data(), &b) != NABWA_OK
				|| nabwa_bam_batch_restore(b, st.data()) != NABWA_OK || nabwa_bam_batch_pass2(b, tab, n_tot, n_mapped) != NABWA_OK) die("pass 2", nabwa_last_error());
			t_lib += now_s() - t1; t_call[2] += now_s() - t1;
			done_ch.put(std::move(b));
		}
		fclose(spill);
	}
	done_ch.close(); finisher.join();
	out_ch.close();
	writer.join();
	if (timing) fprintf(stderr, "[nabwa_bam2bam] timing: start-up (device, index, headers) %.3f s, records %.3f s\n", t_loop - t_main, now_s() - t_loop);
	if (timing) { double ts = 0; for (double x : t_search) ts += x; fprintf(stderr, "[nabwa_bam2bam] timing: search threads (%zu GPU%s) %.3f s busy\n", n_dev, n_dev > 1 ? "s" : "", ts); }
	if (timing) fprintf(stderr, "[nabwa_bam2bam] timing: library calls: create %.3f s, pass 1 %.3f s, pass 2 %.3f s, output %.3f s, destroy %.3f s\n", t_call[0], t_call[1], t_call[2], t_call[3], t_call[4]);
	if (timing) fprintf(stderr, "[nabwa_bam2bam] timing: reader thread %.3f s busy (%.3f s of it inflate, %s), passes 1 and 2 on this thread %.3f s (+ %.3f s waiting for input; the output thread waited %.3f s for the writer), writer thread %.3f s busy (deflate + write)\n",
						t_read, in.t_inflate, in.bgzf ? "BGZF blocks in parallel" : in.raw ? "not compressed" : "one gzip stream", t_lib, t_wait_in, t_wait_out, t_write);
	fprintf(stderr, "[nabwa_bam2bam] %ld sequences processed\n[nabwa_bam2bam] finished cleanly, shutting down.\n"
			"[bwa_paired_sw] %lld out of %lld Q%d singletons are mated.\n[bwa_paired_sw] %lld out of %lld Q%d discordant pairs are fixed.\n",
			tot_seqs, (long long)n_mapped[1], (long long)n_tot[1], 17, (long long)n_mapped[0], (long long)n_tot[0], 17);
	out.close();
	nabwa_isize_table_destroy(tab);
	for (nabwa_index_t *p : ixs) nabwa_index_destroy(p);
	/* final_rename (utils.c:159-173): every trailing '_' goes ("out.bam__" becomes "out.bam") once the file is complete -- unless nothing
	 * would be left of the name or of its last path component */
	if (ofile) {
		size_t e = strlen(ofile);
		const size_t l = e;
		while (e > 0 && ofile[e - 1] == '_') --e;
		if (e > 0 && ofile[e - 1] != '/' && e < l) { std::string to(ofile, e); fprintf(stderr, "[nabwa_bam2bam] finished, renaming %s to %s.\n", ofile, to.c_str()); if (rename(ofile, to.c_str()) != 0) die(ofile, "cannot rename"); }
	}
	return 0;
}
