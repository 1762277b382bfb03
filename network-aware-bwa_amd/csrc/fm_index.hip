// fm_index.hip -- index ingestion and the simple per-row kernels (gfx950).
//
//  * repack_kernel : reference .bwt word stream (4 Occ words + 8 BWT words per 128 rows,
//                    bwtmisc.c:125-152; 16 bases per word, first base in the top bits,
//                    bwt.h:61-66) -> 64-byte bit-plane buckets of 192 rows (nabwa_dev.hpp).
//  * sa_lookup_kernel : bwt_sa / bwt_invPsi (bwt.c:72-81, bwt.h:71-75), one row per lane.
//  * occ4_kernel : bwt_occ4 for tests.
#include "nabwa_dev.hpp"

// base j of B0 in the reference word stream
__device__ __forceinline__ uint32_t ref_base(const uint32_t *w, uint32_t j)
{
	const uint32_t *p = w + (size_t)(j >> 7) * 12 + 4;
	return p[(j & 127u) >> 4] >> ((~j & 15u) << 1) & 3u;
}

// one thread per output bucket; seq_len rows in total
__global__ __launch_bounds__(256) void repack_kernel(const uint32_t *__restrict__ w, uint32_t seq_len,
													 uint32_t n_buckets, uint4 *__restrict__ out)
{
	const uint32_t b = blockIdx.x * 256u + threadIdx.x;
	if (b >= n_buckets) return;
	const uint32_t j0 = b * NABWA_INTV;
	// checkpoint: the reference checkpoint of the enclosing 128-row block plus the rows in between
	uint32_t cnt[4] = {0, 0, 0, 0};
	if (j0 <= seq_len) {
		const uint32_t blk = j0 >> 7;
		const uint32_t *p = w + (size_t)blk * 12;
		cnt[0] = p[0]; cnt[1] = p[1]; cnt[2] = p[2]; cnt[3] = p[3];
		for (uint32_t j = blk << 7; j < j0; ++j) ++cnt[ref_base(w, j)];
	}
	uint64_t lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};
	for (uint32_t g = 0; g < 3; ++g) {
		// 64 rows = 4 reference words, each 16 bases
		for (uint32_t q = 0; q < 4; ++q) {
			const uint32_t j = j0 + g * 64u + q * 16u;
			if (j >= seq_len) break;
			const uint32_t x = w[(size_t)(j >> 7) * 12 + 4 + ((j & 127u) >> 4)];
			const uint32_t nb = seq_len - j < 16u ? seq_len - j : 16u;
			for (uint32_t t = 0; t < nb; ++t) {
				const uint32_t c = x >> ((15u - t) << 1) & 3u;
				lo[g] |= (uint64_t)(c & 1u) << (q * 16u + t);
				hi[g] |= (uint64_t)(c >> 1) << (q * 16u + t);
			}
		}
	}
	uint4 *o = out + (size_t)b * 4;
	o[0] = make_uint4(cnt[0], cnt[1], cnt[2], cnt[3]);
	o[1] = make_uint4((uint32_t)lo[0], (uint32_t)(lo[0] >> 32), (uint32_t)hi[0], (uint32_t)(hi[0] >> 32));
	o[2] = make_uint4((uint32_t)lo[1], (uint32_t)(lo[1] >> 32), (uint32_t)hi[1], (uint32_t)(hi[1] >> 32));
	o[3] = make_uint4((uint32_t)lo[2], (uint32_t)(lo[2] >> 32), (uint32_t)hi[2], (uint32_t)(hi[2] >> 32));
}

extern "C" void nabwa_launch_repack(const uint32_t *w, uint32_t seq_len, uint32_t n_buckets, uint4 *out, hipStream_t s)
{
	hipLaunchKernelGGL(repack_kernel, dim3((n_buckets + 255) / 256), dim3(256), 0, s, w, seq_len, n_buckets, out);
}

// SA[k]: walk LF until the row index is a multiple of sa_intv (one 64-byte bucket per step:
// the base at the row and its rank come from the same line), then add the steps taken.
__global__ __launch_bounds__(256) void sa_lookup_kernel(DevBwt B0, DevBwt B1, int n, const uint8_t *__restrict__ which,
													const uint32_t *__restrict__ kin, uint32_t *__restrict__ out)
{
	const int idx = blockIdx.x * 256 + threadIdx.x;
	if (idx >= n) return;
	const bool w1 = which[idx] != 0;
	const uint4 *bk = w1 ? B1.bk : B0.bk;
	const uint32_t *sa = w1 ? B1.sa : B0.sa;
	const uint32_t primary = w1 ? B1.primary : B0.primary;
	const uint32_t intv = w1 ? B1.sa_intv : B0.sa_intv;
	const uint32_t L0 = w1 ? B1.L2[0] : B0.L2[0], L1 = w1 ? B1.L2[1] : B0.L2[1];
	const uint32_t L2_ = w1 ? B1.L2[2] : B0.L2[2], L3 = w1 ? B1.L2[3] : B0.L2[3];
	uint32_t k = kin[idx], steps = 0;
	const uint32_t *full = w1 ? B1.sa_full : B0.sa_full;
	if (full) { out[idx] = full[k]; return; }
	while (k % intv != 0) {
		++steps;
		if (k == primary) { k = 0; continue; }
		const uint32_t kp = k - (k > primary ? 1u : 0u);   // row of B0 holding this row's base and rank
		const uint32_t b = kp / NABWA_INTV, r = kp - b * NABWA_INTV;
		const uint4 *p = bk + (size_t)b * 4;
		const uint4 q0 = p[0], q1 = p[1], q2 = p[2], q3 = p[3];
		const uint32_t c = nabwa_base_at(q1, q2, q3, r);
		const Occ4 o = nabwa_count4(q0, q1, q2, q3, r);
		const uint32_t Lc = c == 0 ? L0 : (c == 1 ? L1 : (c == 2 ? L2_ : L3));
		k = Lc + o.c[c];
	}
	k /= intv;
	out[idx] = steps + (k ? sa[k] : 0xffffffffu);
}

extern "C" void nabwa_launch_sa_lookup(const DevBwt *B, int n, const uint8_t *which, const uint32_t *k, uint32_t *out,
									   hipStream_t s)
{
	if (n <= 0) return;
	hipLaunchKernelGGL(sa_lookup_kernel, dim3((n + 255) / 256), dim3(256), 0, s, B[0], B[1], n, which, k, out);
}

__global__ __launch_bounds__(256) void occ4_kernel(DevBwt B, int n, const uint32_t *__restrict__ k, uint32_t *__restrict__ out)
{
	const int idx = blockIdx.x * 256 + threadIdx.x;
	if (idx >= n) return;
	const Occ4 o = nabwa_occ4(B, k[idx]);
	out[4 * idx] = o.c[0]; out[4 * idx + 1] = o.c[1]; out[4 * idx + 2] = o.c[2]; out[4 * idx + 3] = o.c[3];
}

extern "C" void nabwa_launch_occ4(const DevBwt *B, int n, const uint32_t *k, uint32_t *out, hipStream_t s)
{
	if (n <= 0) return;
	hipLaunchKernelGGL(occ4_kernel, dim3((n + 255) / 256), dim3(256), 0, s, *B, n, k, out);
}

// Interval table, one level per launch: level t holds the SA interval of every string of t symbols (key = symbols
// as base-4 digits, first consumed symbol most significant), computed from its parent at level t-1 by one backward
// step with the search's own recurrences (bwt.c:237-252): k' = C(c) + Occ(c, k-1) + 1, l' = C(c) + Occ(c, l).
__global__ __launch_bounds__(256) void kmer_level_kernel(DevBwt B, const uint2 *__restrict__ prev, uint2 *__restrict__ cur, uint64_t n_par)
{
	const uint64_t idx = (uint64_t)blockIdx.x * 256u + threadIdx.x;       // one parent per thread: its four children share the rank query
	if (idx >= n_par) return;
	const uint2 par = prev ? prev[idx] : make_uint2(0u, B.seq_len);
	uint2 out[4];
#pragma unroll
	for (int c = 0; c < 4; ++c) out[c] = make_uint2(1u, 0u);
	if (par.x <= par.y) {
		Occ4 ck, cl;
		nabwa_occ4_pair(B, par.x - 1u, par.y, ck, cl);
#pragma unroll
		for (int c = 0; c < 4; ++c) {
			const uint32_t k = B.L2[c] + ck.c[c] + 1u, l = B.L2[c] + cl.c[c];
			if (k <= l) out[c] = make_uint2(k, l);
		}
	}
	uint4 *const dst = (uint4*)(cur + 4 * idx);
	dst[0] = make_uint4(out[0].x, out[0].y, out[1].x, out[1].y);
	dst[1] = make_uint4(out[2].x, out[2].y, out[3].x, out[3].y);
}

extern "C" void nabwa_launch_kmer_level(const DevBwt *B, const uint2 *prev, uint2 *cur, uint64_t n_cur, hipStream_t s)
{
	const uint64_t n_par = n_cur / 4;
	hipLaunchKernelGGL(kmer_level_kernel, dim3((unsigned int)((n_par + 255) / 256)), dim3(256), 0, s, *B, prev, cur, n_par);
}

// Full suffix array, its inverse and the text from the BWT and the row-sampled SA (bwt.c:72-81 is the per-row
// walk this replaces).  One thread per SAMPLED row r (SA value v): LF(r) is the row of the suffix one position
// to the left, so walking LF hands out v-1, v-2, ... until the next sampled row, and the BWT character met at each
// row is the text base at that position.  LF is one cycle over all rows, so every row and every text position is
// written exactly once (row 0 stands for the empty suffix, position seq_len; LF(primary) = 0, bwt.h:71-75).
__global__ __launch_bounds__(256) void sa_fill_kernel(DevBwt B, uint32_t *__restrict__ sa_full, uint32_t *__restrict__ isa,
												  uint8_t *__restrict__ text_bytes)
{
	const uint32_t j = blockIdx.x * 256u + threadIdx.x;
	if (j >= B.n_sa) return;
	uint32_t row = j * B.sa_intv, v = j ? B.sa[j] : B.seq_len;
	sa_full[row] = j ? v : 0xffffffffu;
	isa[v] = row;
	for (;;) {
		if (row == B.primary) break;                          // the whole text: nothing to its left (LF = row 0, sampled)
		const uint32_t kp = row - (row > B.primary ? 1u : 0u);
		const uint32_t b = kp / NABWA_INTV, r = kp - b * NABWA_INTV;
		const uint4 *p = B.bk + (size_t)b * 4;
		const uint4 q0 = p[0], q1 = p[1], q2 = p[2], q3 = p[3];
		const uint32_t c = nabwa_base_at(q1, q2, q3, r);
		const Occ4 o = nabwa_count4(q0, q1, q2, q3, r);
		v -= 1u;
		text_bytes[v] = (uint8_t)c;
		row = B.L2[c] + o.c[c];
		if (row % B.sa_intv == 0u) break;                     // that row's own thread takes over
		sa_full[row] = v; isa[v] = row;
	}
}

__global__ __launch_bounds__(256) void text_pack_kernel(const uint8_t *__restrict__ bytes, uint32_t n, uint32_t n_words, uint32_t *__restrict__ out)
{
	const uint32_t w = blockIdx.x * 256u + threadIdx.x;
	if (w >= n_words) return;
	uint32_t x = 0;
	for (uint32_t t = 0; t < 16u; ++t) { const uint64_t j = (uint64_t)w * 16u + t; if (j < n) x |= (uint32_t)(bytes[j] & 3u) << (2u * t); }
	out[w] = x;
}

extern "C" void nabwa_launch_sa_fill(const DevBwt *B, uint32_t *sa_full, uint32_t *isa, uint8_t *text_bytes, hipStream_t s)
{
	hipLaunchKernelGGL(sa_fill_kernel, dim3((B->n_sa + 255) / 256), dim3(256), 0, s, *B, sa_full, isa, text_bytes);
}
extern "C" void nabwa_launch_text_pack(const uint8_t *bytes, uint32_t n, uint32_t n_words, uint32_t *out, hipStream_t s)
{
	hipLaunchKernelGGL(text_pack_kernel, dim3((n_words + 255) / 256), dim3(256), 0, s, bytes, n, n_words, out);
}
