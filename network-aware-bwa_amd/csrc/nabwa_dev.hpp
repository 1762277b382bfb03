// nabwa_dev.hpp -- device-side FM-index layout and rank primitives (gfx950).
//
// Beside the buckets each index carries an interval table for all 4^T strings of T symbols (T ~ log4(n) - 2),
// which lets an exact tail that starts within T symbols of the read end jump to depth T with ONE 8-byte load
// instead of up to T dependent rank queries (fm_search.hip, "tail jump").
//
// With 288 GB of HBM the index also carries the full suffix array, its inverse and the text (sa_full / isa / text,
// ~8.3 bytes per base and index): once an interval has shrunk to ONE row, "extend by the next read symbol" is a
// comparison with the text base in front of that suffix -- no rank query -- and rows <-> positions are single loads.
//
// HBM layout of one FM-index ("bucket array"): 64-byte buckets, 64-byte aligned, each
// covering NABWA_INTV = 192 consecutive rows of the $-removed BWT string B0:
//
//     u32 cnt[4]                 occurrences of A,C,G,T in B0[0 .. 192*b)
//     3 x { u64 lo ; u64 hi }    bit t of lo/hi = low/high bit of base 192*b + 64*g + t
//
// One rank query touches exactly one 64-byte line (one HBM burst); the bit-plane split makes
// "count all four bases up to row r" three masked popcounts per 64 rows with no shifting.
// The reference keeps 128 rows in 48 bytes (bwt.h:35,61-68, bwtmisc.c:125-152), which straddles
// 64-byte lines; the re-pack is done on the device at load time (fm_index.hip).
#pragma once
#ifdef NABWA_EMU
#include "emu_hip.hpp"   /* tests/emu/: the few vector types and bit intrinsics, for the CPU wave emulation of kernel D (wave_spmd.hpp) */
#else
#include <hip/hip_runtime.h>
#endif
#include <stdint.h>

#define NABWA_INTV 192u

struct DevBwt {
	const uint4 *bk;        // buckets (4 x uint4 each)
	const uint32_t *sa;     // SA samples, sa[j] = SA[j*sa_intv]; sa[0] is never read (bwt.c:80)
	uint32_t primary, seq_len, n_sa, sa_intv;
	uint32_t L2[4];         // cumulative base counts C(c)
	uint32_t n_buckets;
	uint32_t kmer_T;        // 0: no table.  Otherwise kmer[key] = SA interval {k, l} (k > l: empty) of the string whose
	const uint2 *kmer;      // T symbols, in the order the backward search consumes them, are the base-4 digits of key
	const uint2 *kmer_lo;   // levels 1..kmer_LW (= min(T, 12)) back to back, level t at offset (4^t - 4) / 3: the width passes'
	uint32_t kmer_LW, pad_; // first steps; the levels between LW and T only exist while the table is built
	                        // (first consumed symbol = most significant digit); built at load time (fm_index.hip)
	// "text mode" companions, all derived from the BWT + the SA samples at load time (fm_index.hip), or null:
	const uint32_t *sa_full;   // SA value of EVERY row (row 0: ~0u), so bwt_sa is one load
	const uint32_t *isa;       // row of every text position 0..seq_len (inverse of sa_full; isa[seq_len] = 0)
	const uint32_t *text;      // the indexed text itself, 2 bits per base, base j in word j>>4 at bits 2*(j&15)
};

#ifndef NABWA_EMU
// Kernel arguments reach a kernel as wide scalar loads (s_load_dwordx8 / x16), and a value that is part of such a tuple stays
// tied to it: when the kernel has more uniform values than scalar registers -- the search kernels do -- the register allocator
// spills the whole tuple and reads ALL its dwords back through v_readlane wherever one of them is used.  Copies made through an
// opaque move are values of their own: spilled and reloaded one by one, only where used.
__device__ __forceinline__ uint32_t own_u32(uint32_t x) { uint32_t v; asm volatile("s_mov_b32 %0, %1" : "=s"(v) : "s"(x)); return v; }
__device__ __forceinline__ uint64_t own_u64(uint64_t x) { uint64_t v; asm volatile("s_mov_b64 %0, %1" : "=s"(v) : "s"(x)); return v; }
template <class T> __device__ __forceinline__ void own(T &x)
{
	if (sizeof(T) == 4) { uint32_t u; __builtin_memcpy(&u, &x, 4); u = own_u32(u); __builtin_memcpy(&x, &u, 4); }
	else if (sizeof(T) == 8) { uint64_t u; __builtin_memcpy(&u, &x, 8); u = own_u64(u); __builtin_memcpy(&x, &u, 8); }
}
#endif

struct Occ4 { uint32_t c[4]; };

// counts of the four bases in rows [192*b, 192*b + r] of a bucket already in registers
__device__ __forceinline__ Occ4 nabwa_count4(const uint4 &q0, const uint4 &q1, const uint4 &q2, const uint4 &q3, uint32_t r)
{
	const uint32_t g = r >> 6, t = r & 63u;
	const uint64_t part = (2ull << t) - 1ull;            // t == 63 -> all ones
	const uint64_t m0 = g == 0 ? part : ~0ull;
	const uint64_t m1 = g == 0 ? 0ull : (g == 1 ? part : ~0ull);
	const uint64_t m2 = g == 2 ? part : 0ull;
	const uint64_t lo0 = ((uint64_t)q1.y << 32 | q1.x) & m0, hi0 = ((uint64_t)q1.w << 32 | q1.z) & m0;
	const uint64_t lo1 = ((uint64_t)q2.y << 32 | q2.x) & m1, hi1 = ((uint64_t)q2.w << 32 | q2.z) & m1;
	const uint64_t lo2 = ((uint64_t)q3.y << 32 | q3.x) & m2, hi2 = ((uint64_t)q3.w << 32 | q3.z) & m2;
	const uint32_t nlo = __popcll(lo0) + __popcll(lo1) + __popcll(lo2);
	const uint32_t nhi = __popcll(hi0) + __popcll(hi1) + __popcll(hi2);
	const uint32_t nb = __popcll(lo0 & hi0) + __popcll(lo1 & hi1) + __popcll(lo2 & hi2);
	Occ4 o;
	o.c[3] = q0.w + nb;
	o.c[2] = q0.z + (nhi - nb);
	o.c[1] = q0.y + (nlo - nb);
	o.c[0] = q0.x + (r + 1u - nlo - nhi + nb);
	return o;
}

// Occ of all four bases at BWT rows kq and lq, with the reference's conventions
// (bwt.c:159-216): row (u32)-1 gives zeros; rows >= primary shift down by one because '$'
// is not stored.  When both rows fall into one bucket only one 64-byte line is fetched.
__device__ __forceinline__ void nabwa_occ4_pair(const DevBwt &B, uint32_t kq, uint32_t lq, Occ4 &ck, Occ4 &cl)
{
	const uint32_t kp = kq - (kq >= B.primary ? 1u : 0u);
	const uint32_t lp = lq - (lq >= B.primary ? 1u : 0u);
	const bool kvalid = kq != 0xffffffffu, lvalid = lq != 0xffffffffu;
	const uint32_t bl = lvalid ? lp / NABWA_INTV : 0u, rl = lp - bl * NABWA_INTV;
	const uint4 *pl = B.bk + (size_t)bl * 4;
	const uint32_t bkk = kvalid ? kp / NABWA_INTV : bl, rk = kp - bkk * NABWA_INTV;
	uint4 a0 = pl[0], a1 = pl[1], a2 = pl[2], a3 = pl[3];
	if (lvalid) cl = nabwa_count4(a0, a1, a2, a3, rl);
	else { cl.c[0] = cl.c[1] = cl.c[2] = cl.c[3] = 0; }
	if (bkk != bl) {
		const uint4 *pk = B.bk + (size_t)bkk * 4;
		a0 = pk[0]; a1 = pk[1]; a2 = pk[2]; a3 = pk[3];
	}
	if (kvalid) ck = nabwa_count4(a0, a1, a2, a3, rk);
	else { ck.c[0] = ck.c[1] = ck.c[2] = ck.c[3] = 0; }
}

// single-row variant (bwt_occ4)
__device__ __forceinline__ Occ4 nabwa_occ4(const DevBwt &B, uint32_t kq)
{
	Occ4 o;
	if (kq == 0xffffffffu) { o.c[0] = o.c[1] = o.c[2] = o.c[3] = 0; return o; }
	const uint32_t kp = kq - (kq >= B.primary ? 1u : 0u);
	const uint32_t b = kp / NABWA_INTV, r = kp - b * NABWA_INTV;
	const uint4 *p = B.bk + (size_t)b * 4;
	return nabwa_count4(p[0], p[1], p[2], p[3], r);
}

// base stored at row j of B0 (bwt_B0, bwt.h:66) from a bucket in registers
__device__ __forceinline__ uint32_t nabwa_base_at(const uint4 &q1, const uint4 &q2, const uint4 &q3, uint32_t r)
{
	const uint32_t g = r >> 6, t = r & 63u;
	const uint4 q = g == 0 ? q1 : (g == 1 ? q2 : q3);
	const uint64_t lo = (uint64_t)q.y << 32 | q.x, hi = (uint64_t)q.w << 32 | q.z;
	return (uint32_t)(lo >> t & 1ull) | (uint32_t)(hi >> t & 1ull) << 1;
}

// bucket touches the REFERENCE algorithm performs for one (k-1, l) query: one per
// bwt_occ / bwt_occ4 body execution, one for a same-128-row-block pair (bwt.c:92-216)
__device__ __forceinline__ uint32_t ref_touches(const DevBwt &B, uint32_t kq, uint32_t lq, bool four)
{
	const uint32_t NEG = 0xffffffffu;
	const uint32_t bk = (kq == NEG || (!four && kq == B.seq_len)) ? 0u : 1u;
	const uint32_t bl = (lq == NEG || (!four && lq == B.seq_len)) ? 0u : 1u;
	if (kq == lq) return bk;
	const uint32_t _k = kq - (kq >= B.primary ? 1u : 0u), _l = lq - (lq >= B.primary ? 1u : 0u);
	if (!(_l >> 7 != _k >> 7 || kq == NEG || lq == NEG)) return 1u;
	return bk + bl;
}
