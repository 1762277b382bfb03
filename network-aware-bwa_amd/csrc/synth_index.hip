// synth_index.hip -- BENCH / TEST INFRASTRUCTURE (libnabwa_synth.so), not part of the drop-in ABI.
//
// There is no network and no GRCh38 on the GPU box, and the reference's `bwa index` needs hours
// for 3.1 Gbp, so bench.py synthesises a genome of that size and builds both FM-indexes on the
// GPU: suffix array by one 63-bit-prefix radix sort plus prefix-doubling rounds over the rows
// that are still tied (hipCUB/rocPRIM device sort, scan, select), then BWT, Occ checkpoints and
// sampled SA written in the reference's own file layout (bwtio.c:161-204, bwtmisc.c:125-152).
// The arrays produced here go through the same nabwa_index_from_arrays() ingestion as a real
// index, and tests check them byte-for-byte against the reference-built toy index.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <stdint.h>
#include <stdio.h>
#include <string>

static thread_local std::string s_err;
extern "C" const char *nabwa_synth_last_error(void) { return s_err.c_str(); }
#define SCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
	char b_[512]; snprintf(b_, sizeof b_, "%s failed: %s (%s:%d)", #x, hipGetErrorString(e_), __FILE__, __LINE__); \
	s_err = b_; return -1; } } while (0)

#define GRID(n) dim3((unsigned)(((n) + 255) / 256 < 65536 * 16 ? ((n) + 255) / 256 : 65536 * 16)), dim3(256)
#define FOR_ALL(i, n) for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < (n); i += (size_t)gridDim.x * 256)

__device__ __forceinline__ uint64_t splitmix(uint64_t x)
{
	x += 0x9e3779b97f4a7c15ULL; x = (x ^ (x >> 30)) * 0xbf58476d1ce4e5b9ULL; x = (x ^ (x >> 27)) * 0x94d049bb133111ebULL;
	return x ^ (x >> 31);
}

// ---------------------------------------------------------------- synthetic text and reads

__global__ void text_kernel(uint8_t *t, size_t n, uint64_t seed)
{
	FOR_ALL(i, n) t[i] = (uint8_t)(splitmix(seed ^ (uint64_t)i * 0x2545F4914F6CDD1DULL) >> 40 & 3);
}

// copy n_dup segments of dup_len bases to other places (exact or ~1 % diverged) so that repeats exist;
// one block, segments in sequence, so the text is a pure function of the seed
__global__ void dup_kernel(uint8_t *t, size_t n, uint64_t seed, int n_dup, int dup_len)
{
	for (int d = 0; d < n_dup; ++d) {
		const size_t src = splitmix(seed + 2 * d) % (n - dup_len), dst = splitmix(seed + 2 * d + 1) % (n - dup_len);
		const bool diverge = d & 1;
		if (!(src + dup_len > dst && dst + dup_len > src)) {   // skip overlapping picks
			for (int j = threadIdx.x; j < dup_len; j += blockDim.x) {
				uint8_t c = t[src + j];
				if (diverge && splitmix(seed ^ (uint64_t)d << 32 ^ j) % 100 == 0) c = (c + 1 + splitmix(j) % 3) & 3;
				t[dst + j] = c;
			}
		}
		__syncthreads();
	}
}

extern "C" int nabwa_synth_text(int device, uint64_t n, uint64_t seed, int n_dup, int dup_len, uint8_t **d_text)
{
	SCHK(hipSetDevice(device));
	SCHK(hipMalloc(d_text, n + 64));
	SCHK(hipMemset(*d_text, 0, n + 64));
	hipLaunchKernelGGL(text_kernel, GRID(n), 0, 0, *d_text, (size_t)n, seed);
	if (n_dup > 0 && (uint64_t)dup_len * 4 < n) hipLaunchKernelGGL(dup_kernel, dim3(1), dim3(1024), 0, 0, *d_text, (size_t)n, seed * 7 + 1, n_dup, dup_len);
	SCHK(hipGetLastError());
	SCHK(hipDeviceSynchronize());
	return 0;
}

// ---------------------------------------------------------------- a genome with GRCh38-like repeat structure (bench.py --repeats)
// Uniform random text is the friendliest genome there is: every 16-mer is unique, one hit per read.  Real genomes are half
// repeats.  This model plants, deterministically (every family's copies sit in slots given by a bijection, families are
// written one after the other):
//   LINE-like : one 6 kb consensus, 60 000 copies truncated from the 5' end (500 .. 6000 bp kept), 3 .. 20 % divergence each
//   Alu-like  : one 300 bp consensus, 1 000 000 copies at 10 .. 15 % divergence, every tenth from a young subfamily at 1 .. 3 %
//   tandem    : 200 000 loci of a 1 .. 6 bp unit repeated over 30 .. 300 bp
// -- about 17 % of the genome; either strand.  Reads from young copies have several near-identical placements (hit lists,
// repeat mapQ, XA), reads from old ones make the search wade through partial matches.
__device__ __forceinline__ uint8_t mutate(uint8_t c, uint64_t h, uint32_t div_ppm)
{
	return (h % 1000000u) < div_ppm ? (uint8_t)((c + 1 + (h >> 32) % 3) & 3) : c;
}

__global__ void family_kernel(uint8_t *t, size_t n, uint64_t seed, int kind, size_t n_copies, size_t n_slots, size_t slot_bytes, uint64_t mult)
{
	FOR_ALL(c, n_copies) {
		const uint64_t h = splitmix(seed ^ (uint64_t)c * 0xD6E8FEB86659FD93ULL);
		const size_t slot = (size_t)(((uint64_t)c * mult + (uint64_t)kind * 7919u) % n_slots);      /* a bijection on the slots: mult is a prime that does not divide n_slots */
		const size_t dst = slot * slot_bytes + (kind == 2 ? slot_bytes / 2 : 0);
		const bool rev = h >> 63;
		if (kind == 2) {                        // tandem repeat: a unit of 1..6 bases over 30..300 bp
			const int unit = 1 + (int)(splitmix(h ^ 1) % 6), len = 30 + (int)(splitmix(h ^ 2) % 271);
			const uint64_t u = splitmix(h ^ 3);
			for (int j = 0; j < len && dst + j < n; ++j) t[dst + j] = (uint8_t)(u >> (2 * (j % unit)) & 3);
			continue;
		}
		const int cons_len = kind == 0 ? 6000 : 300;
		int keep = cons_len; uint32_t div;
		if (kind == 0) { keep = 500 + (int)(splitmix(h ^ 4) % 5501); div = 30000 + (uint32_t)(splitmix(h ^ 5) % 170001); }
		else div = (c % 10 == 0) ? 10000 + (uint32_t)(splitmix(h ^ 6) % 20001) : 100000 + (uint32_t)(splitmix(h ^ 6) % 50001);
		for (int j = 0; j < keep; ++j) {
			const int cj = cons_len - keep + j;                                   // truncated from the 5' end
			uint8_t b = (uint8_t)(splitmix(seed * 31 + (uint64_t)kind * 1000003u + (uint64_t)cj) >> 40 & 3);      // the family's consensus
			b = mutate(b, splitmix(h + 977 * (uint64_t)j), div);
			const size_t at = dst + (rev ? (size_t)(keep - 1 - j) : (size_t)j);
			if (at < n) t[at] = rev ? (uint8_t)(3 - b) : b;
		}
	}
}

extern "C" int nabwa_synth_text_repeats(int device, uint64_t n, uint64_t seed, uint8_t **d_text)
{
	SCHK(hipSetDevice(device));
	SCHK(hipMalloc(d_text, n + 64));
	SCHK(hipMemset(*d_text, 0, n + 64));
	hipLaunchKernelGGL(text_kernel, GRID(n), 0, 0, *d_text, (size_t)n, seed);
	if (n >= (1ull << 24)) {
		const double scale = (double)n / 3099734149.0;            // copy numbers of the GRCh38-sized genome, scaled
		const size_t n_line = (size_t)(60000 * scale), n_alu = (size_t)(1000000 * scale), n_tr = (size_t)(200000 * scale);
		size_t s8 = n / 8192, s1 = n / 1024;
		const uint64_t P = 1000003ull;
		if (s8 % P == 0) --s8;
		if (s1 % P == 0) --s1;
		if (n_line && n_line <= s8) hipLaunchKernelGGL(family_kernel, GRID(n_line), 0, 0, *d_text, (size_t)n, seed + 101, 0, n_line, s8, (size_t)8192, P);
		if (n_alu && n_alu <= s1) hipLaunchKernelGGL(family_kernel, GRID(n_alu), 0, 0, *d_text, (size_t)n, seed + 202, 1, n_alu, s1, (size_t)1024, P);
		if (n_tr && n_tr <= s1) hipLaunchKernelGGL(family_kernel, GRID(n_tr), 0, 0, *d_text, (size_t)n, seed + 303, 2, n_tr, s1, (size_t)1024, P);
	}
	SCHK(hipGetLastError());
	SCHK(hipDeviceSynchronize());
	return 0;
}

// one read of `len` bases from the window that starts at p (rev: its reverse complement), with substitutions and at most one
// 1-base indel; written as bwa_seq_t.seq (read reversed) and .rseq (reverse complement) codes (bwaseqio.c:294-297)
__device__ void gen_read(const uint8_t *t, size_t p, bool rev, uint64_t h, int len, uint32_t sub_ppm, uint32_t indel_ppm, uint8_t *s, uint8_t *q)
{
	const bool has_indel = (splitmix(h ^ 0x1234) % 1000000u) < indel_ppm;
	const int ipos = 15 + (int)(splitmix(h ^ 0x77) % (uint64_t)(len - 30));
	const bool is_del = splitmix(h ^ 0x99) & 1;
	for (int j = 0; j < len; ++j) {
		// read base j in read orientation
		int jj = rev ? len - 1 - j : j;              // position along the sampled window
		size_t src = p + jj;
		if (has_indel && jj >= ipos) src += is_del ? 1 : 0;
		uint8_t c = t[src];
		if (has_indel && !is_del && jj == ipos) c = (uint8_t)(splitmix(h ^ 0xabc) & 3);
		else if (has_indel && !is_del && jj > ipos) c = t[src - 1];
		if ((splitmix(h + 31 * (uint64_t)jj + 17) % 1000000u) < sub_ppm) c = (c + 1 + splitmix(h ^ jj) % 3) & 3;
		if (rev) c = 3 - c;
		s[len - 1 - j] = c;
		q[len - 1 - j] = 3 - c;
	}
}

// reads sampled from the text, fixed length, plus int64 offsets
__global__ void reads_kernel(const uint8_t *t, size_t n, int n_reads, int len, uint32_t sub_ppm, uint32_t indel_ppm, uint64_t seed,
							 uint8_t *seq, uint8_t *rseq, int64_t *off)
{
	FOR_ALL(r, (size_t)n_reads) {
		const uint64_t h = splitmix(seed ^ (uint64_t)r * 0x9E3779B97F4A7C15ULL);
		const size_t p = h % (n - len - 2);
		off[r] = (int64_t)r * len;
		gen_read(t, p, h >> 63, h, len, sub_ppm, indel_ppm, seq + (size_t)r * len, rseq + (size_t)r * len);
		if (r == (size_t)n_reads - 1) off[n_reads] = (int64_t)n_reads * len;
	}
}

extern "C" int nabwa_synth_reads(int device, const uint8_t *d_text, uint64_t n, int n_reads, int len, uint32_t sub_ppm,
								 uint32_t indel_ppm, uint64_t seed, uint8_t *d_seq, uint8_t *d_rseq, int64_t *d_off)
{
	SCHK(hipSetDevice(device));
	hipLaunchKernelGGL(reads_kernel, GRID((size_t)n_reads), 0, 0, d_text, (size_t)n, n_reads, len, sub_ppm, indel_ppm, seed,
					   d_seq, d_rseq, d_off);
	SCHK(hipGetLastError());
	SCHK(hipDeviceSynchronize());
	return 0;
}

// pairs: fragments of length ~ N(isize_mean, isize_sd) from either strand, read 1 from the fragment's start, read 2
// the reverse complement of its end (FR); reads interleaved, index 2 * pair + end
__global__ void pairs_kernel(const uint8_t *t, size_t n, int n_pairs, int len, uint32_t sub_ppm, uint32_t indel_ppm, float isize_mean, float isize_sd,
							 uint64_t seed, uint8_t *seq, uint8_t *rseq, int64_t *off)
{
	FOR_ALL(r, (size_t)n_pairs) {
		const uint64_t h = splitmix(seed ^ (uint64_t)r * 0x9E3779B97F4A7C15ULL);
		const float u1 = ((splitmix(h ^ 0x51) >> 11) + 1) * (1.0f / 9007199254740993.0f), u2 = (splitmix(h ^ 0x52) >> 11) * (1.0f / 9007199254740992.0f);
		int ins = (int)(isize_mean + isize_sd * sqrtf(-2.0f * logf(u1)) * cosf(6.2831853f * u2) + 0.5f);
		if (ins < len) ins = len;
		if (ins > 60000) ins = 60000;
		const size_t p = h % (n - (size_t)ins - 4);
		const bool flip = h >> 63;                       // the fragment comes from the reverse strand: read 1 is its far end
		for (int e = 0; e < 2; ++e) {
			const size_t rd = 2 * r + e;
			const bool far = (e == 1) != flip;               // this end covers the fragment's right end, reverse-complemented
			off[rd] = (int64_t)rd * len;
			gen_read(t, far ? p + ins - len - 1 : p, far, splitmix(h + 77 * (e + 1)), len, sub_ppm, indel_ppm, seq + rd * len, rseq + rd * len);
		}
		if (r == (size_t)n_pairs - 1) off[2 * (size_t)n_pairs] = (int64_t)2 * n_pairs * len;
	}
}

extern "C" int nabwa_synth_pairs(int device, const uint8_t *d_text, uint64_t n, int n_pairs, int len, uint32_t sub_ppm, uint32_t indel_ppm,
								 float isize_mean, float isize_sd, uint64_t seed, uint8_t *d_seq, uint8_t *d_rseq, int64_t *d_off)
{
	SCHK(hipSetDevice(device));
	hipLaunchKernelGGL(pairs_kernel, GRID((size_t)n_pairs), 0, 0, d_text, (size_t)n, n_pairs, len, sub_ppm, indel_ppm, isize_mean, isize_sd, seed,
					   d_seq, d_rseq, d_off);
	SCHK(hipGetLastError());
	SCHK(hipDeviceSynchronize());
	return 0;
}

// ---------------------------------------------------------------- suffix array

#define KCH 29   // characters packed into the first sort key: 58 bits + 5 bits of "valid" count

__global__ void key0_kernel(const uint8_t *t, size_t n, int reverse, uint64_t *keys, uint32_t *vals)
{
	FOR_ALL(i, n + 1) {
		uint64_t packed = 0;
		const size_t rem = n - i;
		const int valid = rem < KCH ? (int)rem : KCH;
		for (int u = 0; u < KCH; ++u) {
			uint64_t c = 0;
			if (u < valid) c = reverse ? t[n - 1 - (i + u)] : t[i + u];
			packed = packed << 2 | c;
		}
		keys[i] = packed << 5 | (uint64_t)valid;
		vals[i] = (uint32_t)i;
	}
}

// head[j] = j if row j starts a new group else 0 (then max-scanned into group starts)
__global__ void heads_kernel(const uint64_t *keys, size_t m, const uint32_t *slots, uint32_t *head)
{
	FOR_ALL(j, m) {
		const bool first = j == 0 || keys[j] != keys[j - 1];
		head[j] = first ? (slots ? slots[j] : (uint32_t)j) : 0u;
	}
}

// after the scan: gs[j] = first row of j's group.  Write ranks, and flag rows of non-singleton groups.
__global__ void ranks_kernel(const uint64_t *keys, size_t m, const uint32_t *slots, const uint32_t *gs, const uint32_t *sa_vals,
							 uint32_t *rank, uint32_t *group_start, uint8_t *tied)
{
	FOR_ALL(j, m) {
		const bool first = j == 0 || keys[j] != keys[j - 1];
		const bool last = j + 1 == m || keys[j + 1] != keys[j];
		const uint32_t row = slots ? slots[j] : (uint32_t)j;
		rank[sa_vals[j]] = gs[j];
		group_start[row] = gs[j];
		tied[row] = !(first && last);
	}
}

__global__ void key2_kernel(size_t m, const uint32_t *slots, const uint32_t *sa, const uint32_t *group_start, const uint32_t *rank,
							uint32_t h, uint64_t *keys, uint32_t *vals)
{
	FOR_ALL(t, m) {
		const uint32_t row = slots[t], i = sa[row];
		keys[t] = (uint64_t)group_start[row] << 32 | (uint64_t)rank[i + h];
		vals[t] = i;
	}
}

__global__ void writeback_kernel(size_t m, const uint32_t *slots, const uint32_t *vals, uint32_t *sa)
{
	FOR_ALL(t, m) sa[slots[t]] = vals[t];
}

struct MaxOp { __device__ __forceinline__ uint32_t operator()(uint32_t a, uint32_t b) const { return a > b ? a : b; } };

// ---------------------------------------------------------------- BWT / Occ / SA samples in the reference layout

// bw[r] = base preceding suffix SA[r] (4 marks the '$' row); also counts bases per 128-row block of B0 later
__global__ void bwt_kernel(const uint8_t *t, size_t n, int reverse, const uint32_t *sa, uint8_t *bw, uint32_t *primary)
{
	FOR_ALL(r, n + 1) {
		const uint32_t p = sa[r];
		if (p == 0) { bw[r] = 4; *primary = (uint32_t)r; }
		else bw[r] = reverse ? t[n - 1 - (p - 1)] : t[p - 1];
	}
}

// per 128-row block of B0 (the '$'-removed string): base counts, and the 8 packed words
__global__ void pack_kernel(const uint8_t *bw, size_t n, uint32_t primary, uint32_t *words, uint32_t *blk_cnt /* 4 arrays of nblk */,
							size_t nblk)
{
	FOR_ALL(b, nblk) {
		uint32_t c[4] = {0, 0, 0, 0};
		for (int w = 0; w < 8; ++w) {
			uint32_t x = 0; bool any = false;
			for (int u = 0; u < 16; ++u) {
				const size_t k = b * 128 + w * 16 + u;
				if (k < n) {
					const uint8_t base = bw[k < primary ? k : k + 1];
					x |= (uint32_t)base << ((15 - u) << 1);
					++c[base]; any = true;
				}
			}
			if (any) words[b * 12 + 4 + w] = x;
		}
		blk_cnt[b] = c[0]; blk_cnt[nblk + b] = c[1]; blk_cnt[2 * nblk + b] = c[2]; blk_cnt[3 * nblk + b] = c[3];
	}
}

__global__ void ckpt_kernel(const uint32_t *blk_excl /* 4 x (nblk+1) exclusive sums */, size_t nblk, size_t n, uint32_t *words)
{
	FOR_ALL(b, nblk + 1) {
		// checkpoint b sits before block b; the final one (totals) follows the last, possibly partial, block
		size_t pos = b * 12;
		if (b == nblk && (n & 127)) pos = (nblk - 1) * 12 + 4 + ((n & 127) + 15) / 16;
		for (int c = 0; c < 4; ++c) words[pos + c] = blk_excl[c * (nblk + 1) + b];
	}
}

__global__ void sa_sample_kernel(const uint32_t *sa, size_t n_sa, uint32_t intv, uint32_t *out)
{
	FOR_ALL(j, n_sa) if (j > 0) out[j - 1] = sa[j * intv];
}

extern "C" int nabwa_synth_free(void *p) { if (p) SCHK(hipFree(p)); return 0; }

// Build one FM-index of the text (reverse != 0: of the reversed text).  Outputs are DEVICE arrays
// holding exactly the bytes of a .bwt / .sa file written by the reference.
extern "C" int nabwa_synth_build_index(int device, const uint8_t *d_text, uint64_t n64, int reverse, int sa_intv,
									   uint32_t **d_bwt_words, uint64_t *n_bwt_words,
									   uint32_t **d_sa_words, uint64_t *n_sa_words, int verbose)
{
	SCHK(hipSetDevice(device));
	const size_t n = (size_t)n64, m = n + 1;
	if (n64 >= 0xfffffff0ull) { s_err = "text too long for 32-bit suffix indexes"; return -1; }
	uint64_t *kA = 0, *kB = 0; uint32_t *vA = 0, *vB = 0, *rank = 0, *gstart = 0, *head = 0, *slots = 0; uint8_t *tied = 0;
	uint32_t *d_count = 0; void *tmp = 0; size_t tmp_bytes = 0, need = 0;
	hipEvent_t e0, e1; SCHK(hipEventCreate(&e0)); SCHK(hipEventCreate(&e1)); SCHK(hipEventRecord(e0, 0));
	SCHK(hipMalloc(&kA, m * 8)); SCHK(hipMalloc(&kB, m * 8)); SCHK(hipMalloc(&vA, m * 4)); SCHK(hipMalloc(&vB, m * 4));
	hipLaunchKernelGGL(key0_kernel, GRID(m), 0, 0, d_text, n, reverse, kA, vA);
	{
		hipcub::DoubleBuffer<uint64_t> dk(kA, kB); hipcub::DoubleBuffer<uint32_t> dv(vA, vB);
		SCHK(hipcub::DeviceRadixSort::SortPairs(nullptr, need, dk, dv, m, 0, 63, 0));
		tmp_bytes = need; SCHK(hipMalloc(&tmp, tmp_bytes));
		SCHK(hipcub::DeviceRadixSort::SortPairs(tmp, tmp_bytes, dk, dv, m, 0, 63, 0));
		if (dk.Current() != kA) { uint64_t *x = kA; kA = kB; kB = x; }
		if (dv.Current() != vA) { uint32_t *x = vA; vA = vB; vB = x; }
	}
	// kA = sorted keys, vA = suffix array (rows 0..n)
	uint32_t *sa = vA;
	SCHK(hipMalloc(&rank, (m + 1) * 4)); SCHK(hipMalloc(&gstart, m * 4)); SCHK(hipMalloc(&head, m * 4));
	SCHK(hipMalloc(&tied, m)); SCHK(hipMalloc(&slots, m * 4)); SCHK(hipMalloc(&d_count, 8));
	auto ensure_tmp = [&](size_t want) -> int {
		if (want > tmp_bytes) { if (tmp) SCHK(hipFree(tmp)); tmp = 0; SCHK(hipMalloc(&tmp, want)); tmp_bytes = want; }
		return 0;
	};
	hipLaunchKernelGGL(heads_kernel, GRID(m), 0, 0, kA, m, (const uint32_t*)nullptr, head);
	SCHK(hipcub::DeviceScan::InclusiveScan(nullptr, need, head, head, MaxOp(), m, 0));
	if (ensure_tmp(need)) return -1;
	SCHK(hipcub::DeviceScan::InclusiveScan(tmp, tmp_bytes, head, head, MaxOp(), m, 0));
	hipLaunchKernelGGL(ranks_kernel, GRID(m), 0, 0, kA, m, (const uint32_t*)nullptr, head, sa, rank, gstart, tied);
	// the big key buffers are no longer needed at full size
	SCHK(hipDeviceSynchronize());
	SCHK(hipFree(kB)); kB = 0; SCHK(hipFree(kA)); kA = 0; SCHK(hipFree(vB)); vB = 0;
	// ---- prefix doubling over the rows that are still tied
	uint32_t h = KCH; int round = 0; size_t cap2 = 0;
	for (;;) {
		hipcub::CountingInputIterator<uint32_t> iota(0);
		SCHK(hipcub::DeviceSelect::Flagged(nullptr, need, iota, tied, slots, d_count, m, 0));
		if (ensure_tmp(need)) return -1;
		SCHK(hipcub::DeviceSelect::Flagged(tmp, tmp_bytes, iota, tied, slots, d_count, m, 0));
		uint32_t cnt = 0;
		SCHK(hipMemcpy(&cnt, d_count, 4, hipMemcpyDeviceToHost));
		if (verbose) fprintf(stderr, "[synth] %s index: round %d, h=%u, tied rows=%u\n", reverse ? "reverse" : "forward", round, h, cnt);
		if (cnt == 0) break;
		if ((uint64_t)h > n64) { s_err = "prefix doubling did not converge"; return -1; }
		if (cnt > cap2) {
			if (kA) { SCHK(hipFree(kA)); SCHK(hipFree(kB)); SCHK(hipFree(vB)); }
			cap2 = cnt;
			SCHK(hipMalloc(&kA, cap2 * 8)); SCHK(hipMalloc(&kB, cap2 * 8)); SCHK(hipMalloc(&vB, cap2 * 8));   // vB: two u32 buffers
		}
		uint32_t *v1 = vB, *v2 = vB + cap2;
		hipLaunchKernelGGL(key2_kernel, GRID((size_t)cnt), 0, 0, (size_t)cnt, slots, sa, gstart, rank, h, kA, v1);
		hipcub::DoubleBuffer<uint64_t> dk(kA, kB); hipcub::DoubleBuffer<uint32_t> dv(v1, v2);
		SCHK(hipcub::DeviceRadixSort::SortPairs(nullptr, need, dk, dv, (size_t)cnt, 0, 64, 0));
		if (ensure_tmp(need)) return -1;
		SCHK(hipcub::DeviceRadixSort::SortPairs(tmp, tmp_bytes, dk, dv, (size_t)cnt, 0, 64, 0));
		hipLaunchKernelGGL(writeback_kernel, GRID((size_t)cnt), 0, 0, (size_t)cnt, slots, dv.Current(), sa);
		hipLaunchKernelGGL(heads_kernel, GRID((size_t)cnt), 0, 0, dk.Current(), (size_t)cnt, slots, head);
		SCHK(hipcub::DeviceScan::InclusiveScan(nullptr, need, head, head, MaxOp(), (size_t)cnt, 0));
		if (ensure_tmp(need)) return -1;
		SCHK(hipcub::DeviceScan::InclusiveScan(tmp, tmp_bytes, head, head, MaxOp(), (size_t)cnt, 0));
		hipLaunchKernelGGL(ranks_kernel, GRID((size_t)cnt), 0, 0, dk.Current(), (size_t)cnt, slots, head, dv.Current(), rank, gstart, tied);
		SCHK(hipGetLastError());
		h *= 2; ++round;
	}
	if (kA) { SCHK(hipFree(kA)); SCHK(hipFree(kB)); SCHK(hipFree(vB)); kA = kB = 0; vB = 0; }
	SCHK(hipFree(rank)); SCHK(hipFree(gstart)); SCHK(hipFree(head)); SCHK(hipFree(slots)); SCHK(hipFree(tied));

	// ---- BWT, Occ-interleaved word stream (.bwt layout)
	uint8_t *bw = 0; uint32_t *d_primary = d_count;
	SCHK(hipMalloc(&bw, m + 1));
	hipLaunchKernelGGL(bwt_kernel, GRID(m), 0, 0, d_text, n, reverse, sa, bw, d_primary);
	uint32_t primary = 0;
	SCHK(hipMemcpy(&primary, d_primary, 4, hipMemcpyDeviceToHost));
	const size_t nblk = (n + 127) / 128;
	const uint64_t nw = 5 + (n + 15) / 16 + (nblk + 1) * 4;
	uint32_t *words = 0, *blk = 0, *blk_ex = 0;
	SCHK(hipMalloc(&words, nw * 4)); SCHK(hipMemset(words, 0, nw * 4));
	SCHK(hipMalloc(&blk, 4 * nblk * 4)); SCHK(hipMalloc(&blk_ex, 4 * (nblk + 1) * 4));
	hipLaunchKernelGGL(pack_kernel, GRID(nblk), 0, 0, bw, n, primary, words + 5, blk, nblk);
	for (int c = 0; c < 4; ++c) {
		SCHK(hipcub::DeviceScan::ExclusiveSum(nullptr, need, blk + c * nblk, blk_ex + c * (nblk + 1), nblk, 0));
		if (ensure_tmp(need)) return -1;
		SCHK(hipcub::DeviceScan::ExclusiveSum(tmp, tmp_bytes, blk + c * nblk, blk_ex + c * (nblk + 1), nblk, 0));
	}
	// totals = L2 increments: last exclusive value + last block count
	uint32_t tot[4], lastx[4], lastc[4];
	for (int c = 0; c < 4; ++c) {
		SCHK(hipMemcpy(&lastx[c], blk_ex + c * (nblk + 1) + (nblk - 1), 4, hipMemcpyDeviceToHost));
		SCHK(hipMemcpy(&lastc[c], blk + c * nblk + (nblk - 1), 4, hipMemcpyDeviceToHost));
		tot[c] = lastx[c] + lastc[c];
		SCHK(hipMemcpy(blk_ex + c * (nblk + 1) + nblk, &tot[c], 4, hipMemcpyHostToDevice));
	}
	hipLaunchKernelGGL(ckpt_kernel, GRID(nblk + 1), 0, 0, blk_ex, nblk, n, words + 5);
	uint32_t hdr[5] = { primary, tot[0], tot[0] + tot[1], tot[0] + tot[1] + tot[2], tot[0] + tot[1] + tot[2] + tot[3] };
	SCHK(hipMemcpy(words, hdr, 20, hipMemcpyHostToDevice));
	SCHK(hipFree(bw)); SCHK(hipFree(blk)); SCHK(hipFree(blk_ex));
	*d_bwt_words = words; *n_bwt_words = nw;

	// ---- sampled suffix array (.sa layout: primary, 4 skipped words, intv, seq_len, samples 1..n_sa-1)
	if (d_sa_words) {
		const size_t n_sa = (n + sa_intv) / sa_intv;
		uint32_t *sw = 0;
		SCHK(hipMalloc(&sw, (7 + n_sa) * 4));
		uint32_t sh[7] = { primary, hdr[1], hdr[2], hdr[3], hdr[4], (uint32_t)sa_intv, (uint32_t)n };
		SCHK(hipMemcpy(sw, sh, 28, hipMemcpyHostToDevice));
		hipLaunchKernelGGL(sa_sample_kernel, GRID(n_sa), 0, 0, sa, n_sa, (uint32_t)sa_intv, sw + 7);
		*d_sa_words = sw; *n_sa_words = 7 + n_sa - 1;
	}
	SCHK(hipGetLastError());
	SCHK(hipDeviceSynchronize());
	SCHK(hipFree(vA)); SCHK(hipFree(d_count)); if (tmp) SCHK(hipFree(tmp));
	SCHK(hipEventRecord(e1, 0)); SCHK(hipEventSynchronize(e1));
	float ms = 0; SCHK(hipEventElapsedTime(&ms, e0, e1));
	if (verbose) fprintf(stderr, "[synth] %s index of %zu bases built in %.2f s (primary %u)\n", reverse ? "reverse" : "forward", n, ms / 1e3, primary);
	SCHK(hipEventDestroy(e0)); SCHK(hipEventDestroy(e1));
	return 0;
}

// helpers for Python (ctypes) callers: device malloc / copies without torch
extern "C" int nabwa_synth_malloc(int device, uint64_t bytes, void **p) { SCHK(hipSetDevice(device)); SCHK(hipMalloc(p, bytes)); return 0; }
extern "C" int nabwa_synth_d2h(void *dst, const void *src, uint64_t bytes) { SCHK(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost)); return 0; }
extern "C" int nabwa_synth_h2d(void *dst, const void *src, uint64_t bytes) { SCHK(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice)); return 0; }

// ---------------------------------------------------------------- random-gather ceiling (micro-benchmark)
// What the memory system sustains for the access pattern of the FM search: every lane walks a chain of
// DEPENDENT random reads of `bytes_per_access` (16/32/64/128, naturally aligned) from a table far larger
// than the Infinity Cache, `chains` independent chains per lane.  Reported by bench.py next to the
// roofline as the attainable bound for 64-byte random gathers.
template <int VEC, int CHAINS>
__global__ __launch_bounds__(256) void gather_kernel(const uint4 *__restrict__ table, uint64_t n_units, int steps, uint64_t seed,
												 unsigned long long *sink)
{
	const uint64_t tid = (uint64_t)blockIdx.x * 256 + threadIdx.x;
	uint64_t idx[CHAINS]; uint32_t acc = 0;
#pragma unroll
	for (int c = 0; c < CHAINS; ++c) idx[c] = splitmix(seed + tid * CHAINS + c) % n_units;
	for (int s = 0; s < steps; ++s) {
		uint4 v[CHAINS][VEC];
#pragma unroll
		for (int c = 0; c < CHAINS; ++c)
#pragma unroll
			for (int u = 0; u < VEC; ++u) v[c][u] = table[idx[c] * VEC + u];
#pragma unroll
		for (int c = 0; c < CHAINS; ++c) {
			uint32_t x = 0;
#pragma unroll
			for (int u = 0; u < VEC; ++u) x += v[c][u].x ^ v[c][u].y ^ v[c][u].z ^ v[c][u].w;
			acc += x;
			idx[c] = splitmix(idx[c] * 0x9E3779B97F4A7C15ULL + x + s) % n_units;   // next address depends on the data
		}
	}
	if (acc == 0x12345678u) atomicAdd(sink, 1ull);
}

extern "C" int nabwa_synth_gather_bench(int device, uint64_t table_bytes, int bytes_per_access, int chains, int n_blocks, int steps,
										double *gbps, double *maccess_per_s)
{
	SCHK(hipSetDevice(device));
	uint4 *table = 0; unsigned long long *sink = 0;
	SCHK(hipMalloc(&table, table_bytes)); SCHK(hipMalloc(&sink, 8));
	hipLaunchKernelGGL(text_kernel, GRID(table_bytes), 0, 0, (uint8_t*)table, (size_t)table_bytes, 12345ull);
	SCHK(hipMemset(sink, 0, 8));
	const int vec = bytes_per_access / 16;
	const uint64_t n_units = table_bytes / bytes_per_access;
	hipEvent_t e0, e1; SCHK(hipEventCreate(&e0)); SCHK(hipEventCreate(&e1));
	for (int rep = 0; rep < 2; ++rep) {   // first repetition warms up
		SCHK(hipEventRecord(e0, 0));
#define GK(V, C) hipLaunchKernelGGL((gather_kernel<V, C>), dim3(n_blocks), dim3(256), 0, 0, table, n_units, steps, 777ull + rep, sink)
		if (vec == 1 && chains == 1) GK(1, 1); else if (vec == 2 && chains == 1) GK(2, 1); else if (vec == 4 && chains == 1) GK(4, 1);
		else if (vec == 8 && chains == 1) GK(8, 1); else if (vec == 4 && chains == 2) GK(4, 2); else if (vec == 2 && chains == 2) GK(2, 2);
		else if (vec == 4 && chains == 4) GK(4, 4); else if (vec == 8 && chains == 2) GK(8, 2); else if (vec == 1 && chains == 4) GK(1, 4);
		else { s_err = "unsupported gather_bench shape"; return -1; }
#undef GK
		SCHK(hipEventRecord(e1, 0)); SCHK(hipEventSynchronize(e1));
	}
	float ms = 0; SCHK(hipEventElapsedTime(&ms, e0, e1));
	const double acc = (double)n_blocks * 256 * chains * steps;
	*gbps = acc * bytes_per_access / (ms * 1e-3) / 1e9;
	*maccess_per_s = acc / (ms * 1e-3) / 1e6;
	SCHK(hipFree(table)); SCHK(hipFree(sink)); SCHK(hipEventDestroy(e0)); SCHK(hipEventDestroy(e1));
	return 0;
}
