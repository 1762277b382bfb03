// bgzf_in.hpp -- the plain bytes of a BGZF file, of any other gzip stream, or of a file that is not compressed: the input side of
// nabwa_bam2bam (what the reference reads through bamlite's gzread, bamlite.h:7-11).  Host code only; `die(what, why)` is the
// including program's way to end the run.
#pragma once
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <sys/time.h>
#include <zlib.h>
#include <thread>
#include <vector>

static double now_s() { struct timeval tv; gettimeofday(&tv, 0); return tv.tv_sec + 1e-6 * tv.tv_usec; }
static int io_threads() { int nt = (int)std::thread::hardware_concurrency(); if (nt < 1) nt = 1; if (nt > 16) nt = 16; return nt; }

/* ---------------------------------------------------------------- BAM in: the plain bytes of a BGZF file (bgzf.c: independent
 * gzip members of <= 64 KB that carry their own size in a "BC" extra field -- inflated many at a time, one thread per run of
 * blocks) or of any other gzip stream (what the reference's bamlite reads through gzread: one inflate stream, member after member) */
struct BamIn {
	FILE *f; const char *name; bool bgzf, raw, in_eof, z_open;
	std::vector<uint8_t> cmp; size_t lo, hi;          /* compressed bytes not yet inflated: cmp[lo, hi) */
	std::vector<uint8_t> plain; size_t pos;           /* inflated bytes, consumed up to pos */
	z_stream zs;
	double t_inflate;
	struct Blk { size_t src, n_src, dst; uint32_t isize, crc; };

	BamIn(FILE *f_, const char *name_) : f(f_), name(name_), bgzf(false), raw(false), in_eof(false), z_open(false), lo(0), hi(0), pos(0), t_inflate(0)
	{
		cmp.resize((size_t)32 << 20);
		more_input();
		size_t total;
		bgzf = block_at(lo, total) && total != 0;
		raw = hi - lo >= 2 && !(cmp[lo] == 31 && cmp[lo + 1] == 139);         /* not gzip at all: taken as it is, as gzread would */
		if (!bgzf && !raw) {
			memset(&zs, 0, sizeof(zs));
			if (inflateInit2(&zs, 15 + 32) != Z_OK) die(name, "inflateInit failed");
			z_open = true;
		}
	}
	~BamIn() { if (z_open) inflateEnd(&zs); }
	void more_input()
	{
		if (lo && lo == hi) lo = hi = 0;
		if (lo) { memmove(cmp.data(), cmp.data() + lo, hi - lo); hi -= lo; lo = 0; }
		while (!in_eof && hi < cmp.size()) {
			const size_t r = fread(cmp.data() + hi, 1, cmp.size() - hi, f);
			if (r == 0) { if (ferror(f)) die(name, "read error"); in_eof = true; }
			hi += r;
		}
	}
	/* is there a BGZF block header at cmp[o]?  total = the block's size (0: the header is not complete yet) */
	bool block_at(size_t o, size_t &total) const
	{
		total = 0;
		if (hi - o < 12) return hi - o == 0 ? false : (cmp[o] == 31);
		const uint8_t *h = cmp.data() + o;
		if (h[0] != 31 || h[1] != 139 || h[2] != 8 || !(h[3] & 4)) return false;
		const size_t xlen = h[10] | (size_t)h[11] << 8;
		if (hi - o < 12 + xlen) return true;
		for (size_t q = 12; q + 4 <= 12 + xlen; ) {
			const size_t sl = h[q + 2] | (size_t)h[q + 3] << 8;
			if (h[q] == 'B' && h[q + 1] == 'C' && sl == 2 && q + 6 <= 12 + xlen) { total = (size_t)(h[q + 4] | (size_t)h[q + 5] << 8) + 1; return total >= 12 + xlen + 8; }
			q += 4 + sl;
		}
		return false;
	}
	/* more plain bytes behind the unconsumed ones; false when the stream has ended */
	bool fill()
	{
		if (pos) { plain.erase(plain.begin(), plain.begin() + pos); pos = 0; }
		const double t0 = now_s();
		bool got = bgzf ? fill_bgzf() : raw ? fill_raw() : fill_stream();
		t_inflate += now_s() - t0;
		return got;
	}
	bool fill_bgzf()
	{
		for (;;) {
			std::vector<Blk> blk;
			size_t o = lo, out = plain.size();
			while (o < hi) {
				size_t total;
				if (!block_at(o, total)) die(name, "not a BGZF block where one should start");
				if (!total || hi - o < total) break;
				const size_t xlen = cmp[o + 10] | (size_t)cmp[o + 11] << 8;
				Blk b; b.src = o + 12 + xlen; b.n_src = total - 12 - xlen - 8; b.dst = out;
				memcpy(&b.crc, &cmp[o + total - 8], 4); memcpy(&b.isize, &cmp[o + total - 4], 4);
				if (b.isize > 0x10000) die(name, "a BGZF block of more than 64 KB");
				out += b.isize; o += total;
				blk.push_back(b);
			}
			if (blk.empty()) {
				if (in_eof) { if (lo != hi) die(name, "truncated BGZF block"); return false; }
				more_input();
				continue;
			}
			const size_t before = plain.size();
			plain.resize(out);
			int nt = io_threads(); if ((size_t)nt > blk.size()) nt = (int)blk.size();
			std::vector<int> bad(nt, 0);
			std::vector<std::thread> th;
			for (int t = 0; t < nt; ++t) th.emplace_back([&, t]() {
				z_stream z; memset(&z, 0, sizeof(z));
				if (inflateInit2(&z, -15) != Z_OK) { bad[t] = 1; return; }
				for (size_t k = blk.size() * t / nt; k < blk.size() * (t + 1) / nt; ++k) {
					const Blk &b = blk[k];
					uint8_t none;                                           /* an empty block (the end-of-file marker) still has a stream to check */
					z.next_in = cmp.data() + b.src; z.avail_in = (uInt)b.n_src; z.next_out = b.isize ? plain.data() + b.dst : &none; z.avail_out = b.isize;
					const int r = inflate(&z, Z_FINISH);
					if (r != Z_STREAM_END || z.avail_out != 0 || (uint32_t)crc32(crc32(0, 0, 0), b.isize ? plain.data() + b.dst : &none, b.isize) != b.crc) bad[t] = 1;
					inflateReset(&z);
				}
				inflateEnd(&z);
			});
			for (auto &x : th) x.join();
			for (int x : bad) if (x) die(name, "a BGZF block does not inflate to what its trailer says");
			lo = o;
			if (out != before) return true;           /* only empty blocks (the end-of-file marker): look further */
		}
	}
	bool fill_raw()
	{
		if (lo == hi) { more_input(); if (lo == hi) return false; }
		plain.insert(plain.end(), cmp.begin() + lo, cmp.begin() + hi);
		lo = hi;
		return true;
	}
	bool fill_stream()
	{
		const size_t STEP = (size_t)16 << 20;
		const size_t before = plain.size();
		plain.resize(before + STEP);
		zs.next_out = plain.data() + before; zs.avail_out = (uInt)STEP;
		while (zs.avail_out) {
			if (lo == hi) { more_input(); if (lo == hi) break; }
			zs.next_in = cmp.data() + lo; zs.avail_in = (uInt)(hi - lo);
			const int r = inflate(&zs, Z_NO_FLUSH);
			lo = hi - zs.avail_in;
			if (r == Z_STREAM_END) { inflateReset(&zs); continue; }          /* the next member, if there is one */
			if (r != Z_OK && r != Z_BUF_ERROR) die(name, "not a gzip stream, or a damaged one");
			if (r == Z_BUF_ERROR && zs.avail_in == 0 && in_eof && lo == hi) break;
		}
		plain.resize(before + STEP - zs.avail_out);
		return plain.size() != before;
	}
	/* at least n unconsumed bytes if the stream has them; returns how many there are */
	size_t need(size_t n) { while (plain.size() - pos < n) if (!fill()) break; return plain.size() - pos; }
	bool read(void *dst, size_t n) { if (need(n) < n) return false; memcpy(dst, plain.data() + pos, n); pos += n; return true; }
};

