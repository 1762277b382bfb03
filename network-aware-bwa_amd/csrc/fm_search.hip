// fm_search.hip -- bounded-backtracking FM-index search, one read per lane (gfx950).
//
// Computes what bwa_cal_sa_reg_gap does per read (bwtaln.c:93-142): the four bwt_cal_width
// passes (bwtaln.c:52-76) and bwt_match_gap (bwtgap.c:104-266), reproducing the exact
// pop/push order of the reference's per-score LIFO stacks (bwtgap.c:46-79) because that order
// decides which hits are found and in which order they are reported.
//
// Execution model: two persistent kernels per batch.  Kernel W (fm_width_kernel) runs the width
// passes: one lane = one strand of one read, every lane executes the same code on every trip, so its
// waves are converged.  Kernel S (fm_search_kernel) runs the search: every lane is its own state machine
// (pop / expansion / exact-tail step / tail jump / text tail / hit) and performs at most ONE rank query --
// Occ of all four bases at rows (k-1, l), one or two 64-byte bucket fetches -- per trip.  A wave draws
// work in blocks from a global counter.  v2 had both phases in one loop and ran at 31 % VALU lane
// utilisation, issue-bound (profiles/r01_v2_pmc.json); hence the split.
//
// Memory discipline (v2).  With ~260 k reads in flight the caches hold ~128 B of L2 and ~1 KB of
// Infinity Cache per lane, so every touch of lane-private state in HBM costs a 64-byte
// transaction -- as much as the rank query it accompanies (v1 moved 4.5x the algorithmic bytes,
// profiles/r01_v1_pmc.json).  Therefore:
//   * the child that continues the current path (always the LAST push of an expansion) is kept
//     in registers as the "pending" entry (in arena format); it is the next pop whenever its score is
//     not above the lowest score in memory (for the match child: always) and is spilled otherwise, so
//     the observable pop order is unchanged;
//   * an arena entry is one 16-byte word with its list link embedded; head[score] lives in LDS
//     ([score][lane], 2 bytes each); a 64-bit register mask tracks the non-empty scores;
//   * read bases and the per-position bound bytes are read through 16-byte windows in LDS, filled by
//     global_load_lds_dwordx4, that stay valid for 8-16 steps of a descent;
//   * the test "w[i-1] == w[i]" (bwtgap.c:208,212) is precomputed into bit 7 of the bound byte,
//     so the 32-bit interval widths are only touched by gap_shadow (bwtgap.c:81-91);
//   * the width passes write their results in 16-byte chunks (4 widths / 16 bound bytes).
// Work avoidance (v5): exact tails near the read end jump through the interval table; an
// interval of one row is carried as a text position and extended by text comparison, several levels per
// trip; the gap children of an expansion are one grouped stack entry, materialised only if popped; the
// next pop is fetched ahead into LDS while a tail is walked; reads with an exact occurrence (kernel W's
// class) are searched first, by waves that refill all 64 lanes together.
// Bump allocation (arena = pushes that reach memory).  Reads that outgrow the arena, the per-read hit list or the trip budget
// are flagged and searched from the start by kernel D (fm_deep.hip: one search per wavefront, paged arenas) -- never on the CPU.
#include "nabwa_dev.hpp"
#include "fm_search.hpp"

#define ST_IDLE   0
#define ST_WIDTH  1
#define ST_POP    2
#define ST_EXACT  3
#define ST_EXPAND 4
#define ST_EXIT   5

#define STATE_M 0
#define STATE_I 1
#define STATE_D 2
#define STATE_GROUP 3   // arena only: the gap children of one expansion, see fm_search_kernel
#define GRP_EXT 0x100u
// Markers in the `l` word of a stack entry.  A key at depth T = 16 uses all 32 bits (TTTTTTTTTTTTTTTT = 0xffffffff), and the generic
// child test is "k' <= l'": KEYM must therefore be the largest value.  (With KEYM = 0xfffffffe the all-T child of the last table
// level failed that test and reads from poly-T stretches lost a hit -- found on the repeat-model genome, round 2.)  Text positions
// stay below 0xfffffff0 (text_ok).
#define TXM 0xfffffffeu   // l of an interval carried in text form (k = text position)
#define KEYM 0xffffffffu  // l of an interval carried in key form (k = path key: the reference symbols matched so far as base-4 digits)
#define LVO(t_) ((size_t)((((uint64_t)1 << (2 * (t_))) - 4ull) / 3ull))   /* offset of level t in the interval table */
#ifndef NABWA_W_WAVES
#define NABWA_W_WAVES 5   // kernel W: waves per SIMD the register budget is bounded for (102 VGPRs)
#endif
#ifndef NABWA_WORK_CHUNK
#define NABWA_WORK_CHUNK 64u
#endif
#ifndef NABWA_RUN_MAX
#define NABWA_RUN_MAX 4   // levels one text-form trip may walk (every lane of the wave waits for the longest walk)
#endif

// 16-byte register windows are kept as two separate 64-bit scalars and indexed with a select and a
// shift: a dynamically indexed uint4 (or a struct of two halves) makes hipcc keep the value in
// scratch and load one piece back.
__device__ __forceinline__ uint32_t byte_of(uint64_t lo, uint64_t hi, uint32_t idx)
{
	asm volatile("" : "+v"(lo), "+v"(hi));   // opaque at the USE: no dynamic vector extract, and no wait at the load
	const uint64_t h = (idx & 8u) ? hi : lo;
	return (uint32_t)(h >> ((idx & 7u) << 3)) & 0xffu;
}
#define WIN_SET(lo_, hi_, a_, b_, c_, d_) do { lo_ = (uint64_t)(b_) << 32 | (a_); hi_ = (uint64_t)(d_) << 32 | (c_); } while (0)

__device__ __forceinline__ void set_word(uint4 &q, uint32_t c, uint32_t v)
{
	q.x = c == 0 ? v : q.x; q.y = c == 1 ? v : q.y; q.z = c == 2 ? v : q.z; q.w = c == 3 ? v : q.w;
}

extern __shared__ uint16_t s_head[];   // search kernel, first pass only: [score][lane of the block]

#ifndef NABWA_MIN_WAVES
#define NABWA_MIN_WAVES 4   // 128 VGPRs: 4 blocks per CU (measured +10 % over 3)
#endif

// =====================================================================================
// Kernel W: the four bwt_cal_width passes (bwtaln.c:52-76,123-130) of every read.
// One lane = one strand of one read (work item 2 * read + strand): its full pass, then its seed pass; all lanes run
// the same code on every trip, so the wave stays converged; results go to the read's own record in HBM in 16-byte chunks:
//   Wd [2][WL]  u32 interval widths (only gap_shadow reads them later)
//   Bd [2][WLB] bound bytes: min(bid,127) | (w[i-1]==w[i]) << 7
//   SBd[2][SLB] the same for the seed passes
// =====================================================================================
template <bool COUNT>
__global__ __launch_bounds__(NABWA_SEARCH_BLOCK, NABWA_W_WAVES) void fm_width_kernel(const SearchParams P)
{
	const uint32_t lane = threadIdx.x & 63u;
	bool run = false, done = false;
	unsigned int w_next = 0, w_end = 0;                     // this wave's block of work items (item = 2 * read + strand)
	uint32_t rid = 0, x = 0; int len = 0, phase = 0, wi = 0, n = 0, sbase = 0, nN = 0;
	uint32_t sq_off = 0;                                    // this read's offset in the padded base arrays
#define WREC (P.wdata + (size_t)rid * P.wstride)
#define BX(f_) (x ? P.bwt[1].f_ : P.bwt[0].f_)              /* this lane's index: strand x is matched against bwt[x] (bwtaln.c:123-130) */
	// text mode (nabwa_dev.hpp): the pass has narrowed to ONE row, the suffix at text position tp; the next symbol
	// matches iff it equals the text base in front of it, and the width stays 1 (kk == ll is kept as it is)
	bool tmode = false; uint32_t tp = 0, twtag = 0xffffffffu; uint2 twin = make_uint2(0u, 0u);
	const bool text_ok = (P.text_mode & 1) && P.bwt[0].sa_full && P.bwt[1].sa_full;
	uint32_t kk = 0, ll = 0, pw = 0; int bid = 0;
	uint32_t wkey = 0xffffffffu; bool tok = false; int tbase = 0;   // interval-table key of the KT symbols from position tbase of this phase (the current restart point)
	const int KT = (int)P.bwt[0].kmer_T, LW = (int)P.bwt[0].kmer_LW;
	// after a restart (bwtaln.c:66-70) the pattern begins anew at the next position: its key comes from the packed read
	auto rekey = [&](int from) {
		tok = false; tbase = from;
		if (!KT || !P.rd_pack || from + 4 >= n) return;
		const int PW = P.pack_stride / 2;
		const uint32_t *const pk = P.rd_pack + (size_t)rid * P.pack_stride + x * PW;
		if (pk[PW - 1]) return;                                   // an N somewhere in this strand: step by step
		const int q = sbase + from;
		const uint32_t w0 = pk[q >> 4], w1 = pk[(q >> 4) + 1], sh = ((uint32_t)q & 15u) << 1;
		const uint32_t k16 = sh ? (w0 << sh) | (w1 >> (32u - sh)) : w0;
		wkey = k16 >> (2 * (16 - KT)); tok = true;
	};
	uint4 wacc = make_uint4(0, 0, 0, 0); uint64_t blo = 0, bhi = 0, slo = 0, shi = 0; int stag = -1;
	unsigned long long touches = 0;

	for (;;) {
		unsigned long long need = __ballot(!run && !done);
		// equal-length reads (P.w_sync): a wave takes 64 new items only when all its lanes are done, so every lane is at the
		// same position of the same phase and the wave executes ONE of the paths below per trip, not all of them
		if (P.w_sync && __ballot(run) != 0ull) need = 0ull;
		if (need) {
			// work items come in blocks of NABWA_WORK_CHUNK per wave: one atomic on the shared counter per block, not per
			// refill (a single address takes ~10^8 atomics per second)
			if (w_next == w_end) {
				unsigned int base = 0;
				if (lane == 0) base = atomicAdd(P.work_counter + 1, (unsigned int)NABWA_WORK_CHUNK);
				w_next = __builtin_amdgcn_readfirstlane(base); w_end = w_next + NABWA_WORK_CHUNK;
			}
			const unsigned int rank = (unsigned int)__popcll(need & ((1ull << lane) - 1ull)), avail = w_end - w_next;
			const unsigned int base = w_next;
			w_next += min((unsigned int)__popcll(need), avail);
			if (!run && !done && rank < avail) {
				const unsigned int idx = base + rank;
				if (idx < 2u * (unsigned int)P.n) {
					rid = P.ids ? (uint32_t)P.ids[idx >> 1] : idx >> 1; x = idx & 1u;
					const int64_t o = P.poff[rid];
					len = P.rd_len[rid];
					sq_off = (uint32_t)o;
					nN = 0; stag = -1;
					if (P.w_skip_clean && P.n_aln[rid] == 0) { /* its record is as the first run of this kernel left it */ }
					else if (len > 0) {
						run = true; phase = 0; wi = 0; n = len; sbase = 0; tmode = false;
						if (KT) { wkey = P.rd_key[6 * (size_t)rid + 2 + x]; tok = wkey != 0xffffffffu; }
						tbase = 0;
						kk = 0; ll = BX(seq_len); bid = 0; pw = 0; blo = bhi = 0;
					} else { if (x == 0u) P.rd_nN[rid] = 0; if (P.rd_cls) P.rd_cls[2 * (size_t)rid + x] = 0; }   // (an empty read: nothing to search)
				} else done = true;
			}
		}
		if (__ballot(!done) == 0ull) break;
		// output of position p: width, bound byte, 16-byte chunk flushes; `last` = the terminator {w = 0, bid = ++bid}
		// after the final position (bwtaln.c:73-74)
		auto out_pos = [&](int p, bool last) {
			uint8_t *const rec = WREC;
			if (last) ++bid;
			const uint32_t wv = last ? 0u : ll - kk + 1u;
			const uint32_t bv = (uint32_t)(bid > 127 ? 127 : bid) | ((p > 0 && wv == pw) ? 128u : 0u);
			set_word(wacc, (uint32_t)p & 3u, wv);
			{ const uint64_t sh = (uint64_t)bv << (((uint32_t)p & 7u) << 3); if (p & 8) bhi |= sh; else blo |= sh; }
			pw = wv;
			if (phase == 0 && ((p & 3) == 3 || last)) *(uint4*)((uint32_t*)rec + x * P.WL + (p & ~3)) = wacc;
			if ((p & 15) == 15 || last) {
				*(uint4*)(rec + (phase ? P.woff_sbid + x * P.SLB : P.woff_bid + x * P.WLB) + (p & ~15)) =
					make_uint4((uint32_t)blo, (uint32_t)(blo >> 32), (uint32_t)bhi, (uint32_t)(bhi >> 32));
				blo = bhi = 0;
			}
		};
		// bulk text trip: a full pass in text mode at a 16-position chunk boundary -- compare the chunk's 16 read symbols
		// with the 16 text bases to the left of the suffix as packed words; if all match, every width is 1, every bound
		// byte repeats (no restart, w[i-1] == w[i]) and the chunk's output is five 16-byte stores
		bool bulk = run && phase == 0 && tmode && (wi & 15) == 0 && wi + 16 < n && tp >= 16u && pw == 1u;
		if (bulk) {
			if ((wi >> 4) != stag) { const uint4 q = *(const uint4*)((x ? P.rseq : P.seq) + sq_off + wi); WIN_SET(slo, shi, q.x, q.y, q.z, q.w); stag = wi >> 4; }
			const uint32_t w0 = (tp - 16u) >> 4;
			const uint32_t *const txt = BX(text);
			const uint32_t t0 = txt[w0], t1 = txt[w0 + 1u];
			// 2 low bits of each of the 16 read bytes, byte j -> bits 2j
			auto squeeze = [](uint64_t v) -> uint32_t {
				uint64_t y = v & 0x0303030303030303ull;
				y = (y | y >> 6) & 0x000F000F000F000Full; y = (y | y >> 12) & 0x000000FF000000FFull; y = (y | y >> 24) & 0xFFFFull;
				return (uint32_t)y;
			};
			const bool clean = ((slo | shi) & 0xFCFCFCFCFCFCFCFCull) == 0ull;           // no N among them
			const uint32_t rd = squeeze(slo) | squeeze(shi) << 16;
			// text bases tp-16 .. tp-1, base tp-16+m at bits 2m; the read meets them from the top down
			const uint32_t tx = (uint32_t)(((uint64_t)t1 << 32 | t0) >> (((tp - 16u) & 15u) << 1));
			uint32_t rv = __brev(tx); rv = (rv >> 1 & 0x55555555u) | (rv & 0x55555555u) << 1;   // 2-bit groups reversed
			bulk = clean && rv == rd;
			if (bulk) {
				uint8_t *const rec = WREC;
				uint32_t *const wp = (uint32_t*)rec + x * P.WL + wi;
#pragma unroll
				for (int u = 0; u < 4; ++u) *(uint4*)(wp + 4 * u) = make_uint4(1u, 1u, 1u, 1u);
				const uint32_t b4 = ((uint32_t)(bid > 127 ? 127 : bid) | 128u) * 0x01010101u;
				*(uint4*)(rec + P.woff_bid + x * P.WLB + wi) = make_uint4(b4, b4, b4, b4);
				tp -= 16u; wi += 16;
			}
		}
		if (bulk) { /* chunk done */ }
		else if (run && tok && wi - tbase + 4 <= LW && wi + 4 < n) {
			// table trip: the intervals after wi+1 .. wi+4 symbols of this phase are entries of levels wi+1 .. wi+4 of the
			// interval table (fm_index.hip) -- four independent 8-byte loads, the low levels cache-resident -- instead
			// of four dependent rank queries.  An empty entry is the reference's restart (bwtaln.c:66-70): from there on
			// the prefix is no longer the read's, so the rest of the phase steps normally.
			uint2 tv[4];
			const uint2 *const lo = BX(kmer_lo);
#pragma unroll
			for (int u = 0; u < 4; ++u) {
				const int t = wi - tbase + u + 1;
				tv[u] = (lo + (size_t)((((uint64_t)1 << (2 * t)) - 4ull) / 3ull))[wkey >> (2 * (KT - t))];
			}
#pragma unroll
			for (int u = 0; u < 4; ++u) {
				kk = tv[u].x; ll = tv[u].y;
				const bool dead = kk > ll;
				if (dead) { kk = 0; ll = BX(seq_len); ++bid; }
				out_pos(wi, false);
				++wi;
				if (dead) { rekey(wi); break; }
			}
		} else if (run) {
			const int pos = sbase + wi;
			if ((pos >> 4) != stag) { const uint4 q = *(const uint4*)((x ? P.rseq : P.seq) + sq_off + (pos & ~15)); WIN_SET(slo, shi, q.x, q.y, q.z, q.w); stag = pos >> 4; }
			const int cc = (int)byte_of(slo, shi, (uint32_t)pos & 15u);
			// issue the loads of this step before consuming any (one memory latency per trip)
			uint4 qa[4], qb[4]; uint32_t rk = 0, rl = 0, sav = 0; bool kval = false, two = false;
#pragma unroll
			for (int u = 0; u < 4; ++u) { qa[u] = make_uint4(0, 0, 0, 0); qb[u] = make_uint4(0, 0, 0, 0); }
			const bool q = cc < 4 && !tmode;
			if (q) {
				const uint32_t primary = BX(primary), kq = kk - 1u, lq = ll;
				const uint32_t kp = kq - (kq >= primary ? 1u : 0u), lp = lq - (lq >= primary ? 1u : 0u);
				const uint32_t bl = lp / NABWA_INTV; rl = lp - bl * NABWA_INTV;
				kval = kq != 0xffffffffu;
				const uint32_t bkk = kval ? kp / NABWA_INTV : bl; rk = kp - bkk * NABWA_INTV;
				two = bkk != bl;
				const uint4 *const bkt = BX(bk);
				const uint4 *pl = bkt + (size_t)bl * 4, *pk = bkt + (size_t)bkk * 4;
#pragma unroll
				for (int u = 0; u < 4; ++u) qa[u] = pl[u];
				if (two) {
#pragma unroll
					for (int u = 0; u < 4; ++u) qb[u] = pk[u];
				}
				if (COUNT) touches += ref_touches(x ? P.bwt[1] : P.bwt[0], kq, lq, false);
				// one row left: fetch its text position alongside this step's query (the step moves it one to the left)
				if (text_ok && kk == ll) sav = BX(sa_full)[kk];
			}
			if (tmode && tp > 0u && ((tp - 1u) >> 5) != twtag) {
				twtag = (tp - 1u) >> 5;
				twin = *(const uint2*)(BX(text) + 2 * (size_t)twtag);
			}
			if (tmode) {
				bool ok = cc < 4 && tp > 0u;
				if (ok) {
					const uint32_t t = tp - 1u, wd = (t & 16u) ? twin.y : twin.x;
					ok = (wd >> ((t & 15u) << 1) & 3u) == (uint32_t)cc;
					if (ok) tp = t;
				}
				if (cc > 3 && x == 0u && phase == 0) ++nN;
				if (!ok) { tmode = false; kk = 0; ll = BX(seq_len); ++bid; rekey(wi + 1); }   // the restart of bwtaln.c:66-70
			} else {
				const bool was_one = text_ok && cc < 4 && kk == ll;
				if (cc < 4) {
					Occ4 ck, cl;
					ck.c[0] = ck.c[1] = ck.c[2] = ck.c[3] = 0;
					cl = nabwa_count4(qa[0], qa[1], qa[2], qa[3], rl);
					if (kval) {
						const uint4 s0 = two ? qb[0] : qa[0], s1 = two ? qb[1] : qa[1], s2 = two ? qb[2] : qa[2], s3 = two ? qb[3] : qa[3];
						ck = nabwa_count4(s0, s1, s2, s3, rk);
					}
					const uint32_t L2c = cc == 0 ? BX(L2[0]) : (cc == 1 ? BX(L2[1]) : (cc == 2 ? BX(L2[2]) : BX(L2[3])));
					const uint32_t ok = cc == 0 ? ck.c[0] : (cc == 1 ? ck.c[1] : (cc == 2 ? ck.c[2] : ck.c[3]));
					const uint32_t ol = cc == 0 ? cl.c[0] : (cc == 1 ? cl.c[1] : (cc == 2 ? cl.c[2] : cl.c[3]));
					kk = L2c + ok + 1u; ll = L2c + ol;
				} else if (x == 0u && phase == 0) ++nN;
				if (kk > ll || cc > 3) { kk = 0; ll = BX(seq_len); ++bid; rekey(wi + 1); }
				else if (was_one) { tmode = true; tp = sav - 1u; twtag = 0xffffffffu; }
			}
			out_pos(wi, false);
			if (wi + 1 == n) out_pos(wi + 1, true);
			++wi;
			if (wi == n) {
				if (phase == 0 && P.rd_cls) P.rd_cls[2 * (size_t)rid + x] = (uint8_t)(bid - 1 > 4 ? 4 : bid - 1);   // restarts of this strand's pass (bid counts the terminator too, bwtaln.c:66-74)
				if (phase == 0 && len > P.seed_len) {
					phase = 1; wi = 0; n = P.seed_len; sbase = len - P.seed_len; tmode = false;
					if (KT) { wkey = P.rd_key[6 * (size_t)rid + 4 + x]; tok = wkey != 0xffffffffu; }
					tbase = 0;
					kk = 0; ll = BX(seq_len); bid = 0; pw = 0;
				} else { if (x == 0u) P.rd_nN[rid] = (uint8_t)(nN > 255 ? 255 : nN); run = false; }
			}
		}
	}
	if (COUNT) {
		for (int o = 32; o > 0; o >>= 1) touches += __shfl_down(touches, o);
		if (lane == 0 && P.touch_counter) atomicAdd(P.touch_counter + 1, touches);
	}
#undef WREC
#undef BX
}
// =====================================================================================
// Kernel S: bwt_match_gap (bwtgap.c:104-266), one read per lane, exact pop/push order.
//
// Every trip of the wave loop has three phases so that it pays ONE memory latency:
//   1. decide (registers + LDS only): pop the pending entry or pick the arena slot to pop,
//      run the pre-checks that need no memory, choose what this lane does in this trip;
//   2. issue every global load the lane needs -- arena entry, bound-byte window, read-base
//      window, seed-bound window, one or two Occ buckets -- back to back, no use in between;
//   3. consume: park a fetched entry, finish the pre-check, take an exact-tail step, land a jump, expand, or record a hit.
// When the bound window is stale at pop time the rank query is issued speculatively together
// with the window; if the entry then turns out to be pruned (bwtgap.c:156) the counts are dropped.
// =====================================================================================
#define LS_IDLE  0
#define LS_POP   1   // needs to pop
#define LS_HAVE  2   // the entry popped from the arena in the previous trip is in e_*
#define LS_EXACT 3   // inside an exact tail (bwt.c:237-252), next position e_i - 1
#define LS_EXIT  4

template <bool COUNT>
__global__ __launch_bounds__(NABWA_SEARCH_BLOCK, NABWA_MIN_WAVES) void fm_search_kernel(const SearchParams P_)
{
	SearchParams P = P_;
	// the gap options as scalar values of their own (nabwa_dev.hpp: own): they arrive as one 256-bit tuple, which this kernel -- it has
	// more uniform values than scalar registers -- spilled whole and read back, all eight dwords, at 26 places.  (Doing the same for
	// EVERY argument removes more v_readlane instructions still, 492 -> 212, and makes the kernel 3 % slower; for the index
	// descriptors' scalars, their pointers, the batch's pointers or the per-read arrays' pointers alone: 1 - 7 % slower.  Measured, not kept.)
	own(P.n); own(P.s_mm); own(P.s_gapo); own(P.s_gape); own(P.mode); own(P.indel_end_skip); own(P.max_del_occ); own(P.max_entries);
	own(P.max_gape); own(P.max_seed_diff); own(P.seed_len); own(P.max_top2);
	const uint32_t NIL = 0xffffu;
	const uint32_t lane = threadIdx.x & 63u;
	const size_t slot = (size_t)blockIdx.x * NABWA_SEARCH_BLOCK + threadIdx.x;
	uint8_t *const sc = P.scratch + slot * P.lane_stride;
	uint4 *const ent = (uint4*)sc;
	const bool gape_mode = P.mode & 0x01, nonstop = P.mode & 0x10, loggap = P.mode & 0x04;

	int st = LS_IDLE;
	unsigned int w_next = 0, w_end = 0; bool w_sync = false;   // this wave's block of read numbers; lockstep mode
	// per-read
	uint32_t item = 0; int len = 0; uint32_t mdmg = 0;   // mdmg: this read's max_diff | max_gapo << 8
	uint32_t sq_off = 0;                                    // this read's offset in the padded base arrays
	const int KT = (int)P.bwt[0].kmer_T;                    // 0 in the touch-counting run
	// the hand-over budget: in a batch most of whose reads occur exactly on neither strand (the width passes counted them: *n_sync, the reads in front
	// of the class-0 ones in the work order) the searches are bushy ones -- kernel D's kind -- and go there sooner (2 x 150 bp at 2 %: S 133 + D 90 ms
	// at 2000 trips, S 17 + D 154 ms at 200; 100 bp at 0.2 %: 2000 is the optimum; profiles/r03_deep_variants.txt)
	uint32_t budget = P.trip_budget;
	if (!COUNT && P.n_sync && P.trip_budget_hard && 2u * (uint32_t)__builtin_amdgcn_readfirstlane((int)*P.n_sync) > (uint32_t)P.n) budget = P.trip_budget_hard;
#define RID item                                            /* results and width records are indexed by read */
#define REC (P.wdata + (size_t)RID * P.wstride)            /* this read's width record (kernel W) */
#define MD_READ ((int)(mdmg & 0xffu))
#define MG_READ ((int)(mdmg >> 8))
	// current interval
	uint32_t k = 0, l = 0;
	// search globals
	int max_diff = 0, best_score = 0, best_cnt = 0, n_aln = 0, max_ent = 0, n_entries = 0;
	uint32_t bump = 0; uint64_t mask_lo = 0; bool seeded = false; int status = 0;   // masks: non-empty scores; the first pass has at most 64 levels (host) and uses mask_lo only
	// current entry
	int e_i = 0, e_a = 0, e_mm = 0, e_go = 0, e_ge = 0, e_state = 0, e_ldp = 0, e_score = 0, m = 0;
	// pending entry: the last child pushed by the previous expansion, still in registers
	// (kept in the 16-byte arena format, so spilling it is one store and popping it is the arena unpack; p_score < 0: none)
	uint4 pe = make_uint4(0, 0, 0, 0); int p_score = -1;
#define p_valid (p_score >= 0)
	// register windows over read data / the width record
	// 16-byte windows over read data / the width record live in LDS (first pass), [lane][16 bytes], filled by
	// global_load_lds_dwordx4 (no registers in between; layout checked by profiles/probes/lds_dma_probe.hip) and read a byte at a time
	int sq_tag = -1;                     // 16 read bases
	int bw_base = -1, bw_a = -1;         // 16 bound bytes of strand bw_a from bw_base
	int sw_base = -1, sw_a = -1;         // same for the seed bounds
	unsigned long long touches = 0; uint32_t rd_touch = 0, rd_trips = 0;  // COUNT only
	uint32_t rk_kf = 0, rk_row2 = 0, rk_row1 = 0, rk_tx = 0, rk_pop = 0, rk_tail = 0, rk_jump = 0, rk_gap = 0;   // COUNT only: this read's trips by kind
	unsigned long long st_trips = 0, st_expand = 0, st_exact = 0, st_ent = 0, st_spec = 0, st_query = 0, st_two = 0, st_exit = 0, st_jump = 0, st_txe = 0, st_txt = 0;
	bool ovf = false;

	auto head_get = [&](int score) -> uint32_t {
		return (uint32_t)s_head[score * NABWA_SEARCH_BLOCK + threadIdx.x];
	};
	auto head_set = [&](int score, uint32_t v) {
		s_head[score * NABWA_SEARCH_BLOCK + threadIdx.x] = (uint16_t)v;
	};
	auto mask_has = [&](int score) -> bool { return mask_lo >> score & 1ull; };
	auto mask_set = [&](int score) { mask_lo |= 1ull << score; };
	auto mask_clr = [&](int score) { mask_lo &= ~(1ull << score); };
	auto mask_any = [&]() -> bool { return mask_lo != 0ull; };
	auto mask_first = [&]() -> int {          // lowest non-empty score, 0x7fffffff when none
		return mask_lo ? __ffsll((unsigned long long)mask_lo) - 1 : 0x7fffffff;
	};
	// After the first hit best_score is final (bwtgap.c:170), and the loop ends at the first pop whose score
	// exceeds best_score + s_mm (bwtgap.c:144): such a child can never be expanded.  It is still COUNTED
	// (n_entries feeds max_entries and the bwtgap.c:140 cut-off) but never written to the arena.
	auto never_popped = [&](int score) -> bool { return !nonstop && n_aln > 0 && score > best_score + P.s_mm; };
	// append an entry to the in-memory list of its score (n_entries is maintained by the callers)
	// the arena word of an entry: {k, l, i | last_diff_pos << 16, counters / state / strand (+ the list link, first pass)}
	auto mk_entry = [&](uint32_t nk, uint32_t nl, int ni, int nldp, int nmm, int ngo, int nge, int nstate, int na) -> uint4 {
		const uint32_t z = (uint32_t)ni | (uint32_t)nldp << 16;
		return make_uint4(nk, nl, z, (uint32_t)nmm << 16 | (uint32_t)ngo << 20 | (uint32_t)nge << 24 | (uint32_t)nstate << 29 | (uint32_t)na << 31);
	};
	auto push_mem = [&](int score, uint4 e) {
		if (ovf || never_popped(score)) return;
		if ((uint32_t)score >= P.NS) { ovf = true; return; }       // cannot happen (nabwa_api.hip sizes NS); the second pass would take over
		if (bump >= P.cap) { ovf = true; return; }
		const uint32_t s = bump++;
		const uint32_t prev = mask_has(score) ? head_get(score) : NIL;
		e.w |= prev;
		ent[s] = e;
		head_set(score, s);
		mask_set(score);
	};

	// first pass only: the entry at the head of the lowest non-empty score list is fetched ahead of its pop while the
	// lane walks an exact tail (a tail pushes nothing, so that head IS the next pop) and parked in LDS
	uint4 *const s_pf = (uint4*)(s_head + (size_t)P.NS * NABWA_SEARCH_BLOCK);
	uint2 *const s_key = (uint2*)(s_pf + NABWA_SEARCH_BLOCK);
	// Text mode (nabwa_dev.hpp).  An interval of ONE row is carried as {k = text position of that row's suffix, l = TXM}:
	// extending it by a symbol is a comparison with the text base to the left of the suffix, its only non-empty child
	// is the suffix one position further left, and rows are recovered (isa) only where one is reported.  Entries of
	// either form live side by side in the stacks.  s_tw: this lane's 16-base text word and its tag.
	uint2 *const s_tw = s_key + NABWA_SEARCH_BLOCK;
	uint32_t *const s_bw = (uint32_t*)(s_tw + NABWA_SEARCH_BLOCK), *const s_sw = s_bw + 4 * NABWA_SEARCH_BLOCK, *const s_sq = s_sw + 4 * NABWA_SEARCH_BLOCK;
	// s_fb: per strand (byte a) the lower bound of differences of the read's first len-KT symbols, bound[len-KT-1] -- what
	// decides whether every level down to depth KT is a forced match (see the tail jump); 0xff: unknown / no longer valid
	uint32_t *const s_fb = s_sq + 4 * NABWA_SEARCH_BLOCK;
	auto win_byte = [&](const uint32_t *W, uint32_t idx) -> uint32_t {
		return (uint32_t)((const uint8_t*)W)[(threadIdx.x << 4) + idx];
	};
	// 16 bytes from global memory into this lane's window slot
	auto win_load = [&](const void *src, uint32_t *W) {
		__builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)src,
										 (void __attribute__((address_space(3)))*)(W + ((threadIdx.x & ~63u) << 2)), 16, 0, 0);
	};
#define BW_BYTE(i_) win_byte(s_bw, (uint32_t)(i_))
#define SW_BYTE(i_) win_byte(s_sw, (uint32_t)(i_))
#define SQ_BYTE(i_) win_byte(s_sq, (uint32_t)(i_))
	// Key form.  With all levels of the interval table present, a gap-free entry of depth d <= KT (d = len - i reference symbols
	// matched, substitutions included) needs no interval at all: its path key addresses level d, its four children are the
	// 32 bytes at level d+1, its exact tail or forced walk lands at level KT under (key ++ the read's next symbols).  Such
	// entries are carried as {k = key, l = KEYM}, are expanded without a rank query, and take on rows (one table load)
	// when they are reported, reach depth KT, or yield a gap child that is popped.
	const bool kf_ok = KT > 0 && (int)P.bwt[0].kmer_LW == KT && (int)P.bwt[1].kmer_LW == KT && (P.text_mode & 4);
	const bool text_ok = (P.text_mode & 2) && P.bwt[0].sa_full && P.bwt[1].sa_full && P.bwt[0].seq_len < 0xfffffff0u;
	uint32_t pf_slot = NIL;
	bool finish = false;
	// unpack a popped arena entry into the current-entry registers and unlink it (bwtgap.c:66-79)
	auto unpack = [&](const uint4 &r) {
		k = r.x; l = r.y; e_i = (int)(r.z & 0xffffu); e_ldp = (int)(r.z >> 16);
		e_mm = (int)(r.w >> 16 & 15u); e_go = (int)(r.w >> 20 & 15u); e_ge = (int)(r.w >> 24 & 31u);
		e_state = (int)(r.w >> 29 & 3u); e_a = (int)(r.w >> 31);
	};
	auto take_entry = [&](const uint4 &r_ent) {
		unpack(r_ent);
		const uint32_t nx = r_ent.w & 0xffffu;
		head_set(e_score, nx);
		if (nx == NIL) mask_clr(e_score);
		--n_entries;
		if (!nonstop && e_score > best_score + P.s_mm) finish = true;    // bwtgap.c:144
	};

	for (;;) {
		finish = false;
		// ---------------------------------------------------------------- refill
		unsigned long long need = __ballot(st == LS_IDLE);
		const bool any_busy = __ballot(st != LS_IDLE && st != LS_EXIT) != 0ull;
		// reads whose search is known to run alike (an exact occurrence exists: kernel W's class 0, sorted to the end) are
		// taken 64 at a time by a wave with all lanes idle, so its lanes stay in step and its trips stay converged
		if ((P.sync_refill || w_sync) && any_busy) need = 0ull;
		if (need) {
			if (w_next == w_end) {          // blocks of read numbers per wave, as in kernel W
				unsigned int b0 = 0;
				if (lane == 0) b0 = atomicAdd(P.work_counter, (unsigned int)NABWA_WORK_CHUNK);
				w_next = __builtin_amdgcn_readfirstlane(b0); w_end = w_next + NABWA_WORK_CHUNK;
				// class 0 (end of the work order): lockstep.  Class 2+ (start of the work order): also 64 reads at a time without
				// refilling -- not for convergence but for the stragglers: the fewer lanes of a wave are still searching, the fewer
				// code paths a trip runs through, and the launch waits for exactly those reads (cnt[2] = n_sync[-8])
				w_sync = P.n_sync && (w_next >= *P.n_sync || w_next < P.n_sync[-8]);
			}
			const unsigned int rank = (unsigned int)__popcll(need & ((1ull << lane) - 1ull)), avail = w_end - w_next;
			const unsigned int base = w_next;
			w_next += min((unsigned int)__popcll(need), avail);
			if (st == LS_IDLE && rank < avail) {
				const unsigned int idx = base + rank;
				if (idx < (unsigned int)P.n) {
					const uint32_t rid = P.ids ? (uint32_t)P.ids[idx] : idx;
					item = rid;
					const int64_t o = P.poff[rid];
					len = P.rd_len[rid];
					sq_off = (uint32_t)o;
					mdmg = (uint32_t)P.rd_maxdiff[rid] | (uint32_t)P.rd_maxgapo[rid] << 8;
					if (KT) {
						s_key[threadIdx.x] = *(const uint2*)(P.rd_key + 6 * (size_t)rid);   // interval-table keys of the two strands
						uint32_t fb = 0xffffu;
						if (len > KT) { const uint8_t *const bb = REC + P.woff_bid + (len - KT - 1); fb = (uint32_t)(bb[0] & 127u) | (uint32_t)(bb[P.WLB] & 127u) << 8; }
						s_fb[threadIdx.x] = fb;
					}
					n_aln = 0; max_ent = 0; status = NABWA_ST_OK; rd_touch = 0; rd_trips = 0; ovf = false;
					rk_kf = rk_row2 = rk_row1 = rk_tx = rk_pop = rk_tail = rk_jump = rk_gap = 0;
					sq_tag = -1; bw_a = -1; sw_a = -1; p_score = -1; pf_slot = NIL;
					if (text_ok) s_tw[threadIdx.x] = make_uint2(0u, 0x7fffffffu);
					if (len > 0 && (int)P.rd_nN[rid] <= MD_READ) {      // too many N: no search (bwtgap.c:118-123)
						// ---- start of bwt_match_gap (bwtgap.c:104-128)
						seeded = len > P.seed_len;
						max_diff = MD_READ;
						best_score = (MD_READ + 1) * P.s_mm + (MG_READ + 1) * P.s_gapo + (P.max_gape + 1) * P.s_gape;
						best_cnt = 0;
						// roots: strand 0 is pushed first, strand 1 second -> strand 1 (pending) is expanded first
						bump = 0; mask_lo = 0ull;
						push_mem(0, mk_entry(0u, kf_ok ? KEYM : P.bwt[0].seq_len, len, 0, 0, 0, 0, STATE_M, 0));
						pe = mk_entry(0u, kf_ok ? KEYM : P.bwt[0].seq_len, len, 0, 0, 0, 0, STATE_M, 1); p_score = 0;
						n_entries = 2;
						st = LS_POP;
					} else { P.n_aln[item] = 0; P.max_ent[item] = 0; P.status[item] = NABWA_ST_OK; }
				} else st = LS_EXIT;
			}
		}
		if (__ballot(st != LS_EXIT) == 0ull) break;

		// ================================================================ phase 1: decide (no global loads)
		bool want_ent = false; uint32_t ent_slot = 0;
		bool have = st == LS_HAVE;
		if (st == LS_POP) {
			if (n_entries == 0) finish = true;
			else {
				if (max_ent < n_entries) max_ent = n_entries;
				if (n_entries > P.max_entries) finish = true;                 // bwtgap.c:140
				// a search that is still running after trip_budget trips is handed on to kernel D, which gives it a whole wave: the
				// launch cannot end before its longest lane does, and one lane walks a long search pop by pop
				else if (!COUNT && P.trip_budget && rd_trips > budget) { status = NABWA_ST_OVERFLOW; finish = true; }
			}
			if (!finish) {
				const int best_mem = mask_first();
				if (p_valid && p_score <= best_mem) {
					// the pending child is the newest entry of the lowest score: it is the pop
					unpack(pe);
					e_score = p_score; p_score = -1;
					--n_entries;
					if (!nonstop && e_score > best_score + P.s_mm) finish = true;    // bwtgap.c:144
					else have = true;
				} else if (best_mem == 0x7fffffff) {
					finish = true;      // only never-stored children are left: the reference pops one of them and stops (bwtgap.c:144)
				} else {
					if (p_valid) { push_mem(p_score, pe); p_score = -1; }
					if (ovf) { status = NABWA_ST_OVERFLOW; finish = true; }
					else {
						ent_slot = head_get(best_mem); e_score = best_mem;
						if (ent_slot == pf_slot) {                          // already here: pop it and go on in this trip
							const uint4 r = s_pf[threadIdx.x];
							take_entry(r);
							pf_slot = NIL;
							if (!finish) have = true;
						} else want_ent = true;
					}
				}
			}
		}
		// what this lane does with its current entry in this trip
		int kind = 0;                 // 1 expand, 2 exact-tail step, 3 hit without query (i == 0), 4 tail jump, 5 group member
		bool need_win = false, spec = false, forced = false; int win_hi = 0;
		int grp_c = 0, mlev = -1, mpost = 0; uint32_t midx = 0; bool kx = false;      // mlev/midx: table entry that turns a key-form entry into rows (kind 6)
		if (have && !finish && e_state == STATE_GROUP) {
			// pop ONE member of a gap group, newest first (deletion of T, G, C, A, then the insertion); the rest goes back
			// on the stack it came from, where it is again the top
			const uint32_t mk = (uint32_t)e_ldp & 0x1fu, ext = (uint32_t)e_ldp & GRP_EXT, more = (uint32_t)e_ldp >> 9;
			const int j = 31 - __clz((int)mk);
			const uint32_t rest = mk & ~(1u << j);
			if (rest) push_mem(e_score, mk_entry(k, l, e_i, (int)(rest | ext | more << 9), e_mm, e_go, e_ge, STATE_GROUP, e_a));
			else if (more) push_mem(e_score, mk_entry(k + 1u, l, e_i + 1, (int)(3u | (more - 1u) << 9), e_mm, e_go, e_ge, STATE_GROUP, e_a));   // (text form) the next older level
			if (ovf) { status = NABWA_ST_OVERFLOW; finish = true; }
			if (ext) ++e_ge; else ++e_go;
			if (l == KEYM) {                                                // key form: the parent (depth len - e_i - 1) or its child j-1, as rows
				const int dp = len - e_i - 1;
				kind = 6; st = LS_POP; have = false;
				if (j == 0) { mpost = 1; mlev = dp; midx = k; } else { mpost = 2; mlev = dp + 1; midx = k * 4u + (uint32_t)(j - 1); }
			}
			else if (j == 0) { e_state = STATE_I; e_ldp = e_i; }            // the insertion keeps the parent's interval
			else if (l == TXM) { k -= 1u; e_i += 1; e_state = STATE_D; e_ldp = e_i; }   // text form: the deleted base is the one to the left
			else { kind = 5; grp_c = j - 1; st = LS_POP; have = false; }    // a deletion: re-derive its interval
		}
		if (have && !finish) {
			st = LS_POP;
			m = max_diff - (e_mm + e_go); if (gape_mode) m -= e_ge;
			if (m >= 0) {
				const bool kf = l == KEYM;
				const int d = len - e_i;                                      // depth of a gap-free entry
				if (e_i == 0) { if (kf) { kind = 6; mlev = d; midx = k; } else kind = 3; }
				else {
					// the pre-check needs bound[e_i-1]; an expansion then needs bound[e_i-2] as well
					const bool tail = m == 0 && (e_state == STATE_M || gape_mode || e_ge == P.max_gape);
					win_hi = e_i - 1;
					const int win_lo = (tail || e_i < 2) ? e_i - 1 : e_i - 2;
					bool go = true;
					if (bw_a == e_a && win_lo >= bw_base && win_hi < bw_base + 16) {
						if (m < (int)(BW_BYTE(win_hi - bw_base) & 127u)) go = false;   // bwtgap.c:156
					} else { need_win = true; spec = true; }
					if (go) {
						const bool rdkey_ok = KT && (e_a ? s_key[threadIdx.x].y : s_key[threadIdx.x].x) != 0xffffffffu;
						// row form: the path is known when it is the read's own symbols with at most the one just consumed substituted
						const bool row_known = l != TXM && !kf && (e_go | e_ge) == 0 && e_state == STATE_M && (e_mm == 0 || (e_mm == 1 && e_ldp == e_i));
						if (tail) {                                 // nothing may differ any more: exact tail (bwt.c:237-252)
							kind = 2;                                   // its cursor is e_i: position e_i - 1 is consumed next
							// tail jump: the interval after KT symbols (the path so far, then the read's own) is one table entry away
							if (l == TXM) kind = 7;                  // text form: compare the rest of the read with the text
							else if (kf) { if (rdkey_ok) kind = 4; else { kind = 6; mlev = d; midx = k; } }
							else if (rdkey_ok && row_known && d <= KT) kind = 4;
						}
						else {
							// Forced matches.  Differences are allowed at a level only where the bound of the symbols still to
							// come is below m (bwtgap.c:204-207); the bounds never grow towards the read's start, and this entry
							// was not pruned, so bound[len-KT-1] == m means: on every level down to depth KT the only child is
							// the matching one, popped at once (same score, newest) -- an exact walk in all but name.  For an entry
							// whose path is known the table gives the interval it arrives with, or that it dies.
							forced = rdkey_ok && d < KT && (kf || row_known) && (int)(s_fb[threadIdx.x] >> (e_a << 3) & 0xffu) == m;
							if (forced) kind = 4;
							else if (kf && d >= KT) { kind = 6; mlev = d; midx = k; }     // depth KT: rows from here on
							else { kind = 1; --e_i; kx = kf; }
						}
					}
				}
			}
		} else if (st == LS_EXACT) kind = l == TXM ? 7 : 2;
		// a key-form entry turned into rows becomes the current entry of the next trip; a popped gap member takes its shape now
		auto rows_arrived = [&](uint32_t nk, uint32_t nl) {
			k = nk; l = nl;
			if (mpost == 1) { e_state = STATE_I; e_ldp = e_i; }
			else if (mpost == 2) { e_i += 1; e_state = STATE_D; e_ldp = e_i; }
			st = LS_HAVE;
		};
		if (kind == 6 && mlev == 0) { rows_arrived(0u, e_a ? P.bwt[0].seq_len : P.bwt[1].seq_len); kind = 0; }   // depth 0: every row

		const int qb = 1 - e_a;
		const bool tx = l == TXM && (kind == 1 || kind == 3 || kind == 7);
		const int spos = (kind == 2 || kind == 7) ? e_i - 1 : e_i;
		const int stag = (e_a << 20) | (spos >> 4);
		const bool need_seq = (kind == 1 || kind == 2 || kind == 7) && stag != sq_tag;
		bool query = (kind == 1 && !tx && !kx) || kind == 2 || kind == 5;
		// text word in front of the suffix (kinds 1 and 7 in text form)
		uint2 tw = make_uint2(0u, 0u); bool need_tw = false;
		if (tx && kind != 3 && k > 0u) {
			const uint32_t tag = ((k - 1u) >> 4) | (uint32_t)qb << 31;
			tw = s_tw[threadIdx.x];
			need_tw = tw.y != tag;
			tw.y = tag;
		}
		const bool to_text = kind == 1 && !tx && !kx && text_ok && k == l;   // one row left: its children go on in text form
		if (kind == 2 && !need_seq && SQ_BYTE((uint32_t)spos & 15u) > 3u) query = false;   // an N: no query
		const int ii = e_i - (len - P.seed_len);
		const bool use_seed = kind == 1 && e_i > 0 && seeded && ii > 0;
		const bool need_seed = use_seed && !(sw_a == e_a && ii - 1 >= sw_base && ii < sw_base + 16);

		// ================================================================ phase 2: issue every load
		uint4 r_ent = make_uint4(0, 0, 0, 0);
		if (need_win) {
			int base = (win_hi | 7) - 15; if (base < 0) base = 0;
			win_load(REC + P.woff_bid + e_a * P.WLB + base, s_bw); bw_base = base; bw_a = e_a;
		}
		if (need_seq) {
			win_load((e_a ? P.rseq : P.seq) + sq_off + (spos & ~15), s_sq); sq_tag = stag;
		}
		if (need_seed) {
			int base = (ii | 7) - 15; if (base < 0) base = 0;
			win_load(REC + P.woff_sbid + e_a * P.SLB + base, s_sw); sw_base = base; sw_a = e_a;
		}
		uint32_t r_x = 0u;       // text word, or the row of a reported suffix, or the position of the last row
		if (need_tw) r_x = (qb ? P.bwt[1].text : P.bwt[0].text)[(k - 1u) >> 4];
		else if (tx && kind == 3) r_x = (qb ? P.bwt[1].isa : P.bwt[0].isa)[k];
		else if (to_text) r_x = (qb ? P.bwt[1].sa_full : P.bwt[0].sa_full)[k];
		uint2 r_km = make_uint2(1u, 0u);
		if (kind == 4) {
			uint32_t key = e_a ? s_key[threadIdx.x].y : s_key[threadIdx.x].x;
			const int d = len - e_i;
			if (l == KEYM) {          // the path so far, then the read's own next KT - d symbols
				if (d > 0) { const uint32_t sh = 2u * (uint32_t)(KT - d); key = (k << sh) | (sh ? key & ((1u << sh) - 1u) : 0u); }
			} else if (e_mm) {     // the substituted symbol is the one k was derived with: k lies in (C(c), C(c+1)]
				const uint32_t c1 = qb ? P.bwt[1].L2[1] : P.bwt[0].L2[1], c2 = qb ? P.bwt[1].L2[2] : P.bwt[0].L2[2], c3 = qb ? P.bwt[1].L2[3] : P.bwt[0].L2[3];
				const uint32_t cs = (k > c1 ? 1u : 0u) + (k > c2 ? 1u : 0u) + (k > c3 ? 1u : 0u);
				const uint32_t sh = 2u * (uint32_t)(KT - d);
				key = (key & ~(3u << sh)) | cs << sh;
			}
			r_km = (qb ? P.bwt[1].kmer : P.bwt[0].kmer)[key];
		}
		if (kind == 6) r_km = ((qb ? P.bwt[1].kmer_lo : P.bwt[0].kmer_lo) + LVO(mlev))[midx];
		// the rank query: Occ of all four bases at rows k-1 and l of index qb (bwt.c:159-216 conventions)
		uint4 a0, a1, a2, a3, b0, b1, b2, b3; uint32_t rk = 0, rl = 0; bool kvalid = false, two = false;
		a0 = a1 = a2 = a3 = b0 = b1 = b2 = b3 = make_uint4(0, 0, 0, 0);
		if (kx) {       // key form: the four children of this level, 32 bytes of the next table level
			const uint4 *const ch = (const uint4*)((qb ? P.bwt[1].kmer_lo : P.bwt[0].kmer_lo) + LVO(len - e_i) + (size_t)k * 4);
			a0 = ch[0]; a1 = ch[1];
		}
		if (query) {
			const uint32_t primary = qb ? P.bwt[1].primary : P.bwt[0].primary;
			const uint4 *bk = qb ? P.bwt[1].bk : P.bwt[0].bk;
			const uint32_t kq = k - 1u, lq = l;
			const uint32_t kp = kq - (kq >= primary ? 1u : 0u), lp = lq - (lq >= primary ? 1u : 0u);
			const uint32_t bl = lp / NABWA_INTV; rl = lp - bl * NABWA_INTV;
			kvalid = kq != 0xffffffffu;
			const uint32_t bkk = kvalid ? kp / NABWA_INTV : bl; rk = kp - bkk * NABWA_INTV;
			two = bkk != bl;
			const uint4 *pl = bk + (size_t)bl * 4;
			a0 = pl[0]; a1 = pl[1]; a2 = pl[2]; a3 = pl[3];
			if (two) { const uint4 *pk = bk + (size_t)bkk * 4; b0 = pk[0]; b1 = pk[1]; b2 = pk[2]; b3 = pk[3]; }
		}

		// look-ahead pop (see s_pf): only from inside a tail, with nothing pending in registers
		bool pf_now = false; uint32_t pf_cand = NIL;
		if (want_ent) { pf_cand = ent_slot; pf_now = true; }                // a pop that was not fetched ahead: fetch now, pop next trip
		else if ((kind == 2 || kind == 4 || kind == 7) && !p_valid && mask_any()) {
			const int bm = mask_first();
			pf_cand = head_get(bm);
			pf_now = pf_cand != pf_slot;
		}
		if (pf_now) r_ent = ent[pf_cand];
		if (!COUNT && P.trip_budget && st != LS_IDLE && st != LS_EXIT) ++rd_trips;
		if (COUNT && st != LS_IDLE && st != LS_EXIT) {
			++rd_trips;
			if (kind == 1) { if (kx) ++rk_kf; else if (tx) ++rk_tx; else if (two) ++rk_row2; else ++rk_row1; if (e_go | e_ge) ++rk_gap; }
			else if (want_ent) ++rk_pop; else if (kind == 2 || kind == 7) ++rk_tail; else if (kind == 4) ++rk_jump;
		}
		if (COUNT) {   // trip statistics (instrumented build only): [2] trips, [3..] lane-trips by activity
			const unsigned long long bx = __ballot(kind == 1), be = __ballot(kind == 2), bm = __ballot(want_ent),
				bs = __ballot(spec), bq = __ballot(query), b2 = __ballot(query && two), bi = __ballot(st == LS_EXIT), bj = __ballot(kind == 4), bt = __ballot(tx && kind == 1), b7 = __ballot(kind == 7);
			if (lane == 0) {
				st_trips += 1; st_expand += __popcll(bx); st_exact += __popcll(be); st_ent += __popcll(bm); st_spec += __popcll(bs);
				st_query += __popcll(bq); st_two += __popcll(b2); st_exit += __popcll(bi); st_jump += __popcll(bj); st_txe += __popcll(bt); st_txt += __popcll(b7);
			}
		}
		asm volatile("" ::: "memory");   // keep every consumer below every load above (no block merging across)
		__builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): the window loads above wrote LDS behind the compiler's back
		// ================================================================ phase 3: consume
		if (pf_now) { s_pf[threadIdx.x] = r_ent; pf_slot = pf_cand; }
		if (spec && m < (int)(BW_BYTE(win_hi - bw_base) & 127u)) kind = 0;   // pruned after all (bwtgap.c:156)
		if (need_tw) { tw.x = r_x; s_tw[threadIdx.x] = tw; }
		int c = 4;
		if (kind == 1 || kind == 2) c = (int)SQ_BYTE((uint32_t)spos & 15u);
		if (kind == 2 && c > 3) query = false;
		Occ4 ck, cl;
		ck.c[0] = ck.c[1] = ck.c[2] = ck.c[3] = 0; cl = ck;
		if (query && kind) {
			cl = nabwa_count4(a0, a1, a2, a3, rl);
			if (kvalid) {       // one counting sequence for row k-1 too: pick its bucket first (16 selects < a third copy of the count)
				const uint4 s0 = two ? b0 : a0, s1 = two ? b1 : a1, s2 = two ? b2 : a2, s3 = two ? b3 : a3;
				ck = nabwa_count4(s0, s1, s2, s3, rk);
			}
			if (COUNT && kind != 5) rd_touch += ref_touches(qb ? P.bwt[1] : P.bwt[0], k - 1u, l, kind == 1);   // (the reference derived a group's members in the parent's query)
		}
		if (kx && kind == 1) {
			// key form: child j exists iff its table interval is not empty, and is again a key: present it to the generic
			// expansion below as "counts" (with C(.) = 0) that yield {key * 4 + j, KEYM}
			const uint32_t ek[4] = { a0.x, a0.z, a1.x, a1.z }, el[4] = { a0.y, a0.w, a1.y, a1.w };
#pragma unroll
			for (int j = 0; j < 4; ++j) { const bool ne = ek[j] <= el[j]; ck.c[j] = ne ? k * 4u + (uint32_t)j - 1u : 0u; cl.c[j] = ne ? KEYM : 0u; }
		}
		const bool tx1 = tx && kind == 1;
		if (tx1) {
			// the one row's only non-empty child is for the text base b to the left of the suffix, and it is the suffix
			// starting there: present it to the generic expansion below as "counts" (with C(.) = 0) that yield {k-1, TXM}
			const uint32_t b = k > 0u ? (tw.x >> (((k - 1u) & 15u) << 1) & 3u) : 4u;
#pragma unroll
			for (int j = 0; j < 4; ++j) { ck.c[j] = b == (uint32_t)j ? k - 2u : 0u; cl.c[j] = b == (uint32_t)j ? TXM : 0u; }
		}
		const bool syn = tx1 || (kx && kind == 1);
		const uint32_t L2q0 = syn ? 0u : (qb ? P.bwt[1].L2[0] : P.bwt[0].L2[0]), L2q1 = syn ? 0u : (qb ? P.bwt[1].L2[1] : P.bwt[0].L2[1]);
		const uint32_t L2q2 = syn ? 0u : (qb ? P.bwt[1].L2[2] : P.bwt[0].L2[2]), L2q3 = syn ? 0u : (qb ? P.bwt[1].L2[3] : P.bwt[0].L2[3]);
		const uint32_t seqlen_q = qb ? P.bwt[1].seq_len : P.bwt[0].seq_len;
#define L2Q(cc) ((cc) == 0 ? L2q0 : ((cc) == 1 ? L2q1 : ((cc) == 2 ? L2q2 : L2q3)))
#define CK(cc) ((cc) == 0 ? ck.c[0] : ((cc) == 1 ? ck.c[1] : ((cc) == 2 ? ck.c[2] : ck.c[3])))
#define CL(cc) ((cc) == 0 ? cl.c[0] : ((cc) == 1 ? cl.c[1] : ((cc) == 2 ? cl.c[2] : cl.c[3])))

		if (kind == 6) rows_arrived(r_km.x, r_km.y);
		else if (kind == 5) {                                   // the popped deletion: one more base of the reference, same read position
			k = L2Q(grp_c) + CK(grp_c) + 1u; l = L2Q(grp_c) + CL(grp_c);
			e_i += 1; e_state = STATE_D; e_ldp = e_i;
			st = LS_HAVE;
		} else if (kind == 4) {                                 // landed at depth KT (len > KT, so the tail goes on)
			k = r_km.x; l = r_km.y;
			if (k > l) st = LS_POP;
			else if (forced) { e_i = len - KT; e_ldp = 0; st = LS_HAVE; }   // arrives as the matching child it would have become
			else { e_i = len - KT; st = LS_EXACT; }
		} else if (kind == 7) {
			// exact tail in text form: read symbols e_i-1, e_i-2, ... against the text to the left, as far as both 16-base words reach
			uint32_t pp = k; bool fail = false;
			for (;;) {
				const uint32_t cc = SQ_BYTE((uint32_t)(e_i - 1) & 15u);
				if (cc > 3u || pp == 0u) { fail = true; break; }
				const uint32_t q = pp - 1u;
				if ((tw.x >> ((q & 15u) << 1) & 3u) != cc) { fail = true; break; }
				pp = q; --e_i;
				if (e_i == 0 || (e_i & 15) == 0 || (pp & 15u) == 0u) break;
			}
			k = pp;
			if (fail) st = LS_POP;
			else if (e_i == 0) st = LS_HAVE;                          // matched to the end: reported (as a row) in the next trip
			else st = LS_EXACT;
		} else if (kind == 2 || kind == 3) {
			bool hit = false;
			if (kind == 3 && tx) { k = r_x; l = r_x; }                // the row of the suffix (isa)
			if (kind == 3) hit = true;
			else if (c > 3) st = LS_POP;                          // an N in the tail: no match
			else {
				k = L2Q(c) + CK(c) + 1u; l = L2Q(c) + CL(c);
				if (k > l) st = LS_POP;
				else if (--e_i == 0) hit = true;
				else st = LS_EXACT;
			}
			if (hit) {
				// ---- hit bookkeeping (bwtgap.c:166-199)
				st = LS_POP;
				const int score = e_mm * P.s_mm + e_go * P.s_gapo + e_ge * P.s_gape;
				bool do_add = true;
				if (n_aln == 0) {
					best_score = score;
					const int best_diff = e_mm + e_go + (gape_mode ? e_ge : 0);
					if (!nonstop) max_diff = best_diff + 1 > MD_READ ? MD_READ : best_diff + 1;
				}
				if (score == best_score) best_cnt += (int)(l - k + 1u);
				else if (best_cnt > P.max_top2) { finish = true; do_add = false; }
				uint4 *const out = P.aln + (size_t)item * P.aln_cap;
				if (do_add && e_go) {
					for (int j = 0; j < n_aln; ++j) { const uint4 h = out[j]; if (h.y == k && h.z == l) { do_add = false; break; } }
				}
				if (do_add) {
					if (n_aln == P.aln_cap) { status = NABWA_ST_OVERFLOW; finish = true; }
					else {
						// gap_shadow (bwtgap.c:81-91) on this strand's bounds, positions < last_diff_pos;
						// 8 positions per trip; the weq flags of positions 1..last_diff_pos are refreshed
						const uint32_t x = l - k + 1u, mx = seqlen_q; uint32_t jj = 0, pw = 0;
						uint8_t *const rec = REC;
						uint32_t *const wp = (uint32_t*)rec + e_a * P.WL; uint8_t *const bp = rec + P.woff_bid + e_a * P.WLB;
						for (int t0 = 0; t0 <= e_ldp && e_ldp > 0; t0 += 8) {
							const uint4 w0 = *(const uint4*)(wp + t0), w1 = *(const uint4*)(wp + t0 + 4);
							const uint2 bq = *(const uint2*)(bp + t0);
							uint32_t wv[8] = { w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w };
							uint32_t bb[2] = { bq.x, bq.y };
#pragma unroll
							for (int u = 0; u < 8; ++u) {
								const int t = t0 + u;
								uint32_t bv = bb[u >> 2] >> ((u & 3) * 8) & 0xffu;
								if (t < e_ldp) {
									if (wv[u] > x) wv[u] -= x;
									else if (wv[u] == x) { bv = (bv & 128u) | 1u; wv[u] = mx - (++jj); }
								}
								if (t <= e_ldp) {
									bv = (bv & 127u) | ((t > 0 && wv[u] == pw) ? 128u : 0u);
									bb[u >> 2] = (bb[u >> 2] & ~(0xffu << ((u & 3) * 8))) | bv << ((u & 3) * 8);
								}
								pw = wv[u];
							}
							*(uint4*)(wp + t0) = make_uint4(wv[0], wv[1], wv[2], wv[3]);
							*(uint4*)(wp + t0 + 4) = make_uint4(wv[4], wv[5], wv[6], wv[7]);
							*(uint2*)(bp + t0) = make_uint2(bb[0], bb[1]);
						}
						bw_a = -1;   // the bound window may be stale now
						if (KT && e_ldp > len - KT - 1) s_fb[threadIdx.x] |= 0xffu << (e_a << 3);   // ... and so may the cached bound
						out[n_aln] = make_uint4((uint32_t)e_mm | (uint32_t)e_go << 8 | (uint32_t)e_ge << 16 | (uint32_t)e_a << 24,
												k, l, (uint32_t)score);
						++n_aln;
					}
				}
			}
		} else if (kind == 1) {
			// ---- node expansion (bwtgap.c:201-260); e_i is already decremented
			const uint32_t occ = tx1 ? 1u : l - k + 1u;                   // (key form: only read by deletion-state parents, which are never in key form)
			bool match_child = false;
			bool allow_diff = true, allow_M = true;
			if (e_i > 0) {
				const uint32_t B1 = BW_BYTE(e_i - 1 - bw_base), B0 = BW_BYTE(e_i - bw_base);
				const int b1 = (int)(B1 & 127u), b0v = (int)(B0 & 127u);
				if (b1 > m - 1) allow_diff = false;
				else if (b1 == m - 1 && b0v == m - 1 && (B0 & 128u)) allow_M = false;
				if (use_seed) {
					const uint32_t S1 = SW_BYTE(ii - 1 - sw_base), S0 = SW_BYTE(ii - sw_base);
					const int s1 = (int)(S1 & 127u), s0 = (int)(S0 & 127u);
					const int m_seed = m - max_diff + P.max_seed_diff;      // max_seed_diff - (the same differences), bwtgap.c:153
					if (s1 > m_seed - 1) allow_diff = false;
					else if (s1 == m_seed - 1 && s0 == m_seed - 1 && (S0 & 128u)) allow_M = false;
				}
			}
			// children go through a one-entry delay: the last one stays in registers as `pending`
			auto emit = [&](int score, uint32_t nk, uint32_t nl, int ni, int nldp, int nmm, int ngo, int nge, int nstate, int count) {
				if (p_valid) push_mem(p_score, pe);
				pe = mk_entry(nk, nl, ni, nldp, nmm & 0xff, ngo & 0xff, nge & 0xff, nstate, e_a); p_score = score;
				n_entries += count;
			};
			const int sc0 = e_mm * P.s_mm + e_go * P.s_gapo + e_ge * P.s_gape;
			int tmp = e_go + e_ge;
			if (loggap) { const uint32_t v = (uint32_t)(e_ge + e_go); tmp = (v ? 31 - __clz((int)v) : 0) / 2 + 1; }
			// Text-form fast path.  While the next read symbols equal the text in front of the suffix, a level of the search
			// tree is fully determined without memory: the only non-empty child is the matching one, which is the next pop
			// (same score, newest), and the gap children -- if the bounds allow differences there -- are one insertion and
			// one deletion.  So several levels are walked in this trip, as far as the 16-byte windows reach, and their gap
			// groups leave as ONE stack entry (`more` older levels behind the newest, see the pop side).
			bool ran = false;
			if (tx1 && e_state == STATE_M && n_entries + 64 <= P.max_entries) {
				const int ies = P.indel_end_skip + tmp, seed_lo = len - P.seed_len;
				const int m_seed = m - max_diff + P.max_seed_diff;
				int n = 0, icur = e_i, ilast = e_i; uint32_t pcur = k; bool grp_run = false;
				for (;;) {
					const uint32_t cc = SQ_BYTE((uint32_t)icur & 15u);
					if (cc > 3u || pcur == 0u) break;
					const uint32_t q = pcur - 1u;
					if ((tw.x >> ((q & 15u) << 1) & 3u) != cc) break;
					bool ad = true;
					const bool us = seeded && icur > 0 && icur - seed_lo > 0;
					if (icur > 0) {
						if ((int)(BW_BYTE(icur - 1 - bw_base) & 127u) > m - 1) ad = false;
						if (us && (int)(SW_BYTE(icur - seed_lo - 1 - sw_base) & 127u) > m_seed - 1) ad = false;
					}
					const bool g = ad && icur >= ies && len - icur >= ies && e_go < MG_READ;
					if (n == 0) grp_run = g; else if (g != grp_run) break;
					++n; pcur = q; ilast = icur;
					// may the child (position icur, suffix q) be popped and expanded in this same trip?
					if (n == NABWA_RUN_MAX || icur == 0) break;
					if (m < (int)(BW_BYTE(icur - 1 - bw_base) & 127u)) break;       // its pop prunes it (bwtgap.c:156)
					if ((icur & 15) == 0 || (q & 15u) == 0u) break;                  // read / text window ends
					if (icur - 1 > 0 && icur - 2 < bw_base) break;                   // bound window ends
					if (seeded && icur - 1 > 0 && icur - 1 - seed_lo > 0 && icur - 1 - seed_lo - 1 < sw_base) break;
					--icur;
				}
				if (n) {
					ran = true;
					if (grp_run) emit(sc0 + P.s_gapo, k - (uint32_t)(n - 1), TXM, ilast, (int)(3u | (uint32_t)(n - 1) << 9), e_mm, e_go, e_ge, STATE_GROUP, 2 * n);
					emit(sc0, pcur, TXM, ilast, 0, e_mm, e_go, e_ge, STATE_M, 1);
				}
			}
			if (ran) { /* children done */ }
			else if (allow_diff && e_i >= P.indel_end_skip + tmp && len - e_i >= P.indel_end_skip + tmp) {
				// The gap children of one expansion (an insertion and/or up to four deletions, bwtgap.c:216-240) share one
				// score and are pushed back to back, i.e. they are ADJACENT in that score's stack.  They travel as one
				// STATE_GROUP entry: the parent's interval and position plus a member mask (bit 0 the insertion, bit 1+j the
				// deletion of base j, GRP_EXT: gap extension of a deletion rather than gap open, bits 9..15: `more` older levels of a
				// text-form run, each {insertion, deletion}); a member is materialised --
				// the deletions by repeating the parent's rank query -- only if it is ever popped (see the pop side).  Most
				// never are: once a hit exists, scores above best_score + s_mm end the search (bwtgap.c:144).
				uint32_t dm = 0;
#pragma unroll
				for (int j = 0; j < 4; ++j) { const uint32_t nk = L2Q(j) + CK(j) + 1u, nl = L2Q(j) + CL(j); if (nk <= nl) dm |= 2u << j; }
				if (e_state == STATE_M) {
					if (e_go < MG_READ) emit(sc0 + P.s_gapo, k, l, e_i, (int)(dm | 1u), e_mm, e_go, e_ge, STATE_GROUP, __popc(dm) + 1);
				} else if (e_state == STATE_I) {
					if (e_ge < P.max_gape) emit(sc0 + P.s_gape, k, l, e_i, e_i, e_mm, e_go, e_ge + 1, STATE_I, 1);
				} else if (e_ge < P.max_gape) {
					if ((e_ge + e_go < max_diff || occ < (uint32_t)P.max_del_occ) && dm)
						emit(sc0 + P.s_gape, k, l, e_i, (int)(dm | GRP_EXT), e_mm, e_go, e_ge, STATE_GROUP, __popc(dm));
				}
			}
			if (ran) { /* children done */ }
			else if (allow_diff && allow_M) {
#pragma unroll
				for (int j = 1; j <= 4; ++j) {
					const int cc = (c + j) & 3; const bool is_mm = (j != 4 || c > 3);
					const uint32_t nk = L2Q(cc) + CK(cc) + 1u, nl = L2Q(cc) + CL(cc);
					if (nk <= nl) { emit(sc0 + (is_mm ? P.s_mm : 0), nk, nl, e_i, is_mm ? e_i : 0, e_mm + (is_mm ? 1 : 0), e_go, e_ge, STATE_M, 1); if (!is_mm) match_child = true; }
				}
			} else if (c < 4) {
				const uint32_t nk = L2Q(c) + CK(c) + 1u, nl = L2Q(c) + CL(c);
				if (nk <= nl) { emit(sc0, nk, nl, e_i, 0, e_mm, e_go, e_ge, STATE_M, 1); match_child = true; }
			}
			// the matching child of a one-row interval (always the last emit, still in the pending registers) goes on in text form
			if (to_text && match_child) { pe.x = r_x - 1u; pe.y = TXM; }
			if (ovf) { status = NABWA_ST_OVERFLOW; finish = true; }
		}
#undef L2Q
#undef CK
#undef CL

		if (finish) {
			P.n_aln[item] = n_aln; P.max_ent[item] = (!COUNT && status == NABWA_ST_OVERFLOW) ? (int)rd_trips : max_ent; P.status[item] = (uint8_t)status;      // (a search handed on: the trips it took here -- kernel D's statistics dump reads them; its own max_ent replaces this)
			if (COUNT && status == NABWA_ST_OK) touches += rd_touch;   // abandoned reads are counted by the wide pass
			if (COUNT && P.touch_counter && rd_trips > 8000u) {
				atomicAdd(P.touch_counter + 16, 1ull); atomicAdd(P.touch_counter + 17, (unsigned long long)rd_trips); atomicAdd(P.touch_counter + 18, (unsigned long long)rk_kf);
				atomicAdd(P.touch_counter + 19, (unsigned long long)rk_row2); atomicAdd(P.touch_counter + 20, (unsigned long long)rk_row1); atomicAdd(P.touch_counter + 21, (unsigned long long)rk_tx);
				atomicAdd(P.touch_counter + 22, (unsigned long long)rk_pop); atomicAdd(P.touch_counter + 23, (unsigned long long)rk_tail); atomicAdd(P.touch_counter + 24, (unsigned long long)rk_jump); atomicAdd(P.touch_counter + 25, (unsigned long long)rk_gap);
			}
			if (COUNT && P.touch_counter) { atomicMax(P.touch_counter + 13, (unsigned long long)rd_trips); if (rd_trips > 2000u) atomicAdd(P.touch_counter + 14, 1ull); if (rd_trips > 500u) atomicAdd(P.touch_counter + 15, 1ull); }
			st = LS_IDLE;
		}
	}
	if (COUNT) {
		for (int o = 32; o > 0; o >>= 1) touches += __shfl_down(touches, o);
		if (lane == 0 && P.touch_counter) {
			atomicAdd(P.touch_counter, touches);
			atomicAdd(P.touch_counter + 2, st_trips); atomicAdd(P.touch_counter + 3, st_expand); atomicAdd(P.touch_counter + 4, st_exact);
			atomicAdd(P.touch_counter + 5, st_ent); atomicAdd(P.touch_counter + 6, st_spec); atomicAdd(P.touch_counter + 7, st_query);
			atomicAdd(P.touch_counter + 8, st_two); atomicAdd(P.touch_counter + 9, st_exit); atomicAdd(P.touch_counter + 10, st_jump); atomicAdd(P.touch_counter + 11, st_txe); atomicAdd(P.touch_counter + 12, st_txt);
		}
	}
}

#undef p_valid
#undef BW_BYTE
#undef SW_BYTE
#undef SQ_BYTE
#undef RID
#undef REC
#undef MD_READ
#undef MG_READ
extern "C" void nabwa_launch_fm_width(const SearchParams *P, int n_blocks, hipStream_t s)
{
	if (P->touch_counter) hipLaunchKernelGGL((fm_width_kernel<true>), dim3(n_blocks), dim3(NABWA_SEARCH_BLOCK), 0, s, *P);
	else hipLaunchKernelGGL((fm_width_kernel<false>), dim3(n_blocks), dim3(NABWA_SEARCH_BLOCK), 0, s, *P);
}

extern "C" int nabwa_width_occupancy(void)
{
	int nb = 0;
	return hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, fm_width_kernel<false>, NABWA_SEARCH_BLOCK, 0) == hipSuccess ? nb : 0;
}

extern "C" void nabwa_launch_fm_search(const SearchParams *P, int n_blocks, hipStream_t s)
{
	const size_t lds = (size_t)P->NS * NABWA_SEARCH_BLOCK * 2 + NABWA_SEARCH_BLOCK * 84;
	if (P->touch_counter) hipLaunchKernelGGL((fm_search_kernel<true>), dim3(n_blocks), dim3(NABWA_SEARCH_BLOCK), lds, s, *P);
	else hipLaunchKernelGGL((fm_search_kernel<false>), dim3(n_blocks), dim3(NABWA_SEARCH_BLOCK), lds, s, *P);
}

extern "C" int nabwa_search_occupancy(int ns)
{
	int nb = 0;
	hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, fm_search_kernel<false>, NABWA_SEARCH_BLOCK,
																(size_t)ns * NABWA_SEARCH_BLOCK * 2 + NABWA_SEARCH_BLOCK * 84);
	return e == hipSuccess ? nb : 0;
}

// re-lay the reads out with every read starting on a 16-byte boundary (register windows load 16 bases), and form
// the interval-table keys: the first T symbols the search consumes (positions len-1 down to len-T) of each strand.
// 16 lanes per read, one 16-base chunk of both strands per lane and turn: loads and stores of a read are consecutive
// across its lanes (one thread per read walking bytes took 62 ms for 10 M reads; this form streams at HBM rate).
__global__ __launch_bounds__(256) void pad_reads_kernel(int n, const uint8_t *__restrict__ seq, const uint8_t *__restrict__ rseq,
													const int64_t *__restrict__ off, const int64_t *__restrict__ poff,
													uint8_t *__restrict__ pseq, uint8_t *__restrict__ prseq, int32_t *__restrict__ rd_len,
													uint32_t *__restrict__ rd_key, int T, int seed_len, uint32_t *__restrict__ rd_pack, int pack_stride,
													const uint8_t *__restrict__ md_tab, const uint8_t *__restrict__ mg_tab,
													uint8_t *__restrict__ rd_md, uint8_t *__restrict__ rd_mg)
{
	const int t = threadIdx.x & 15;
	const int i = blockIdx.x * 16 + (threadIdx.x >> 4);
	if (i >= n) return;                                          /* whole 16-lane groups leave together */
	const int64_t o = off[i], p = poff[i];
	const int L = (int)(off[i + 1] - o), NC = (int)(poff[i + 1] - p) >> 4;
	if (t == 0) { rd_len[i] = L; rd_md[i] = md_tab[L]; rd_mg[i] = mg_tab[L]; }      /* max_diff / max_gapo of a read follow from its length (host tables) */
	// both strands 2 bits per base as well, base j in word j>>4 from the TOP bits down, so that the 16 symbols from any position
	// are one 32-bit extract in table-key order; the word after the last says whether the strand holds an N (then: no keys)
	const int PW = pack_stride / 2;
	uint32_t *const pk = rd_pack ? rd_pack + (size_t)i * pack_stride : 0;
	const int n_turns = pk && PW - 1 > NC ? PW - 1 : NC;
	uint32_t anyN[2] = { 0u, 0u };
	for (int c = t; c < n_turns; c += 16) {
		for (int st = 0; st < 2; ++st) {
			const uint8_t *src = (st ? rseq : seq) + o + 16 * c;
			uint32_t w[4] = { 0x04040404u, 0x04040404u, 0x04040404u, 0x04040404u };
			const int have = L - 16 * c;                          /* bases of this chunk that exist */
			if (have >= 16) __builtin_memcpy(w, src, 16);
			else {
#pragma unroll
				for (int k = 0; k < 16; ++k) if (k < have) { const uint32_t v = src[k]; w[k >> 2] = (w[k >> 2] & ~(0xffu << (8 * (k & 3)))) | v << (8 * (k & 3)); }
			}
			if (c < NC) *(uint4*)((st ? prseq : pseq) + p + 16 * c) = make_uint4(w[0], w[1], w[2], w[3]);
			if (pk && c < PW - 1) {
				uint32_t x = 0;
#pragma unroll
				for (int k = 0; k < 16; ++k) {
					const uint32_t v = k < have ? (w[k >> 2] >> (8 * (k & 3)) & 0xffu) : 0u;
					anyN[st] |= v > 3u ? 1u : 0u;
					x |= (v & 3u) << (30 - 2 * k);
				}
				pk[st * PW + c] = x;
			}
		}
	}
	if (pk) {
		for (int m = 8; m; m >>= 1) { anyN[0] |= __shfl_xor(anyN[0], m, 16); anyN[1] |= __shfl_xor(anyN[1], m, 16); }
		if (t < 2) pk[t * PW + PW - 1] = anyN[t];
	}
	// keys (first consumed symbol = most significant digit): [0,1] kernel S, from position L-1 downwards, seq / rseq;
	// [2,3] kernel W full passes, from position 0 upwards; [4,5] kernel W seed passes, from position L-seed_len upwards
	if (t < 6) {
		const int v = t >> 1;
		uint32_t key = 0xffffffffu;
		if (T > 0 && L > T && (v < 2 || (L > seed_len && seed_len > T))) {
			const uint8_t *src = ((t & 1) ? rseq : seq) + o;
			uint32_t a = 0; bool ok = true;
			for (int q = 1; q <= T; ++q) {
				const int pos = v == 0 ? L - q : (v == 1 ? q - 1 : L - seed_len + q - 1);
				const uint32_t x = src[pos];
				ok = ok && x < 4u; a = a << 2 | (x & 3u);
			}
			if (ok) key = a;
		}
		rd_key[6 * (size_t)i + t] = key;
	}
}

extern "C" void nabwa_launch_pad_reads(int n, const uint8_t *seq, const uint8_t *rseq, const int64_t *off, const int64_t *poff,
									   uint8_t *pseq, uint8_t *prseq, int32_t *rd_len, uint32_t *rd_key, int T, int seed_len, uint32_t *rd_pack, int pack_stride,
									   const uint8_t *md_tab, const uint8_t *mg_tab, uint8_t *rd_md, uint8_t *rd_mg, hipStream_t s)
{
	if (n <= 0) return;
	hipLaunchKernelGGL(pad_reads_kernel, dim3((n + 15) / 16), dim3(256), 0, s, n, seq, rseq, off, poff, pseq, prseq, rd_len, rd_key, T, seed_len, rd_pack, pack_stride,
					   md_tab, mg_tab, rd_md, rd_mg);
}

// padded length of every read (starts go on 16-byte boundaries), and 0 after the last: input of the scan that gives the starts
__global__ __launch_bounds__(256) void padded_len_kernel(int n, const int64_t *__restrict__ off, int64_t *__restrict__ plen)
{
	const int i = blockIdx.x * 256 + threadIdx.x;
	if (i <= n) plen[i] = i < n ? (off[i + 1] - off[i] + 15) / 16 * 16 : 0;
}

extern "C" void nabwa_launch_padded_len(int n, const int64_t *off, int64_t *plen, hipStream_t s)
{
	hipLaunchKernelGGL(padded_len_kernel, dim3(n / 256 + 1), dim3(256), 0, s, n, off, plen);
}

__global__ __launch_bounds__(256) void collect_kernel(int n, const uint8_t *__restrict__ status, int32_t *__restrict__ ids,
												  unsigned int *__restrict__ count, int which)
{
	const int i = blockIdx.x * 256 + threadIdx.x;
	if (i < n && status[i] == which) ids[atomicAdd(count, 1u)] = i;
}

// The same, in the order kernel D should start them: its launch ends with its longest search, so the searches that look
// longest go first.  (Measured with NABWA_DEEP_DUMP + profiles/probes/deep_order_probe.py on the ancient-DNA workload: by the searches'
// round counts this key schedules no better than a random order, 1.19 x the bound, and a key of {no hit yet, max_diff, read length}
// reaches 1.07 -- but the launch got 4 % SLOWER with it: the longest search then runs its whole life beside a full machine, 21 us per
// round instead of 17 when it finishes alone.  Wave priority for the head of the list changed nothing.  Not kept.)  Key = max_diff - (the lower bound of the read's differences from kernel W: the smaller restart count of
// its two strands): the more differences the bounds leave open, the larger the tree; reads that filled the first pass's hit list
// (repeat families: hundreds of hit rows, every one-difference variant walked to the end) go before everything else.  One pass per
// key value, largest first.
__global__ __launch_bounds__(256) void collect_keyed_kernel(int n, const uint8_t *__restrict__ status, int32_t *__restrict__ ids,
															unsigned int *__restrict__ count, int which, const uint8_t *__restrict__ cls,
															const uint8_t *__restrict__ md, int lo, int hi, const int32_t *__restrict__ n_aln, int aln_cap, int max_key)
{
	const int i = blockIdx.x * 256 + threadIdx.x;
	if (i >= n || status[i] != which) return;
	const int c0 = cls[2 * (size_t)i], c1 = cls[2 * (size_t)i + 1];
	int k = (int)md[i] - (c0 < c1 ? c0 : c1);
	if (n_aln[i] >= aln_cap) k = max_key;          // the first pass filled its hit list: a read from a repeat family, the longest searches there are
	if (k >= lo && k <= hi) ids[atomicAdd(count, 1u)] = i;
}

extern "C" void nabwa_launch_collect_keyed(int n, const uint8_t *status, int32_t *ids, unsigned int *count, int which,
										   const uint8_t *cls, const uint8_t *md, int max_key, const int32_t *n_aln, int aln_cap, hipStream_t s)
{
	if (n <= 0) return;
	for (int key = max_key; key >= 0; --key)
		hipLaunchKernelGGL(collect_keyed_kernel, dim3((n + 255) / 256), dim3(256), 0, s, n, status, ids, count, which, cls, md,
						   key ? key : -0x7fffffff, key == max_key ? 0x7fffffff : key, n_aln, aln_cap, max_key);
}

// ids of the reads whose status is `which` (NABWA_ST_OVERFLOW after kernel S, NABWA_ST_POOL / NABWA_ST_HITCAP after kernel D)
extern "C" void nabwa_launch_collect(int n, const uint8_t *status, int32_t *ids, unsigned int *count, int which, hipStream_t s)
{
	if (n <= 0) return;
	hipLaunchKernelGGL(collect_kernel, dim3((n + 255) / 256), dim3(256), 0, s, n, status, ids, count, which);
}

__global__ __launch_bounds__(256) void scatter_wide_kernel(int n2, const int32_t *__restrict__ ids, const int32_t *__restrict__ n_aln2,
													   const int32_t *__restrict__ max_ent2, const uint8_t *__restrict__ status2,
													   int32_t *__restrict__ n_aln, int32_t *__restrict__ max_ent,
													   uint8_t *__restrict__ status, const int32_t *__restrict__ wide_idx)
{
	const int q = blockIdx.x * 256 + threadIdx.x;
	if (q >= n2) return;
	const int rid = ids[q], j = wide_idx[rid];       /* j: the read's row in the wide result arrays (assign_slots_kernel) */
	if (status2[j] == NABWA_ST_OK) { n_aln[rid] = n_aln2[j]; max_ent[rid] = max_ent2[j]; status[rid] = NABWA_ST_WIDE; }
	else { n_aln[rid] = 0; max_ent[rid] = max_ent2[j]; status[rid] = status2[j]; }     /* NABWA_ST_POOL / NABWA_ST_HITCAP: the host decides */
}

// results of the searches that were run again with longer hit lists (nabwa_batch_sync): the q-th search of the list -> its read; its
// rows are block + q * cap3, entered in the table of grown row blocks at slot0 + q, which the read's wide_idx names from now on
__global__ __launch_bounds__(256) void scatter_grown_kernel(int n2, const int32_t *__restrict__ ids, const int32_t *__restrict__ n_aln3,
															const int32_t *__restrict__ max_ent3, const uint8_t *__restrict__ status3,
															int32_t *__restrict__ n_aln, int32_t *__restrict__ max_ent, uint8_t *__restrict__ status,
															int32_t *__restrict__ wide_idx, const uint4 *block, size_t cap3, const uint4 **__restrict__ grown, int slot0)
{
	const int q = blockIdx.x * 256 + threadIdx.x;
	if (q >= n2) return;
	const int rid = ids[q];
	max_ent[rid] = max_ent3[q];
	if (status3[q] == NABWA_ST_OK) { n_aln[rid] = n_aln3[q]; status[rid] = NABWA_ST_GROWN; wide_idx[rid] = slot0 + q; grown[slot0 + q] = block + (size_t)q * cap3; }
	else { n_aln[rid] = 0; status[rid] = status3[q]; }
}
extern "C" void nabwa_launch_scatter_grown(int n2, const int32_t *ids, const int32_t *n_aln3, const int32_t *max_ent3, const uint8_t *status3,
										   int32_t *n_aln, int32_t *max_ent, uint8_t *status, int32_t *wide_idx, const uint4 *block, size_t cap3,
										   const uint4 **grown, int slot0, hipStream_t s)
{
	if (n2 <= 0) return;
	hipLaunchKernelGGL(scatter_grown_kernel, dim3((n2 + 255) / 256), dim3(256), 0, s, n2, ids, n_aln3, max_ent3, status3, n_aln, max_ent, status, wide_idx, block, cap3, grown, slot0);
}

// the reads that go to the wide passes get a row each in the wide result arrays; it stays theirs over the tiers
__global__ __launch_bounds__(256) void assign_slots_kernel(int n2, const int32_t *__restrict__ ids, int32_t *__restrict__ wide_idx)
{
	const int j = blockIdx.x * 256 + threadIdx.x;
	if (j < n2) wide_idx[ids[j]] = j;
}

extern "C" void nabwa_launch_assign_slots(int n2, const int32_t *ids, int32_t *wide_idx, hipStream_t s)
{
	if (n2 <= 0) return;
	hipLaunchKernelGGL(assign_slots_kernel, dim3((n2 + 255) / 256), dim3(256), 0, s, n2, ids, wide_idx);
}

extern "C" void nabwa_launch_scatter_wide(int n2, const int32_t *ids, const int32_t *n_aln2, const int32_t *max_ent2,
										  const uint8_t *status2, int32_t *n_aln, int32_t *max_ent, uint8_t *status,
										  int32_t *wide_idx, hipStream_t s)
{
	if (n2 <= 0) return;
	hipLaunchKernelGGL(scatter_wide_kernel, dim3((n2 + 255) / 256), dim3(256), 0, s, n2, ids, n_aln2, max_ent2, status2,
					   n_aln, max_ent, status, wide_idx);
}

__device__ __forceinline__ const uint4 *rows_of(int i, const uint4 *aln, int aln_cap, const uint8_t *status,
												const int32_t *wide_idx, const uint4 *aln2, int aln_cap2, const uint4 *const *grown)
{
	const int st = status[i];
	if (st == NABWA_ST_GROWN) return grown[wide_idx[i]];           /* a hit list that outgrew the wide rows: its own block (nabwa_batch_sync) */
	return st == NABWA_ST_WIDE ? aln2 + (size_t)wide_idx[i] * aln_cap2 : aln + (size_t)i * aln_cap;
}

// compaction: rows of read i go to out[row_off[i] ...]
__global__ __launch_bounds__(256) void gather_kernel(int n, const int32_t *__restrict__ n_aln, const uint32_t *__restrict__ row_off,
												 const uint4 *__restrict__ aln, int aln_cap, const uint8_t *__restrict__ status,
												 const int32_t *__restrict__ wide_idx, const uint4 *__restrict__ aln2, int aln_cap2,
												 const uint4 *const *__restrict__ grown, uint4 *__restrict__ out)
{
	const int i = blockIdx.x * 256 + threadIdx.x;
	if (i >= n) return;
	const int na = n_aln[i];
	const uint4 *src = rows_of(i, aln, aln_cap, status, wide_idx, aln2, aln_cap2, grown);
	uint4 *dst = out + row_off[i];
	for (int j = 0; j < na; ++j) dst[j] = src[j];
}

extern "C" void nabwa_launch_gather(int n, const int32_t *n_aln, const uint32_t *row_off, const uint4 *aln, int aln_cap,
									const uint8_t *status, const int32_t *wide_idx, const uint4 *aln2, int aln_cap2,
									const uint4 *const *grown, uint4 *out, hipStream_t s)
{
	if (n <= 0) return;
	hipLaunchKernelGGL(gather_kernel, dim3((n + 255) / 256), dim3(256), 0, s, n, n_aln, row_off, aln, aln_cap, status,
					   wide_idx, aln2, aln_cap2, grown, out);
}

// order-independent checksum over all hits: sum of a mix of (read, row index, row words)
__device__ __forceinline__ uint64_t mix64(uint64_t x)
{
	x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
	return x;
}

__global__ __launch_bounds__(256) void checksum_kernel(int n, const int32_t *__restrict__ n_aln, const uint4 *__restrict__ aln,
												   int aln_cap, const uint8_t *__restrict__ status, const int32_t *__restrict__ wide_idx,
												   const uint4 *__restrict__ aln2, int aln_cap2, const uint4 *const *__restrict__ grown,
												   unsigned long long *__restrict__ sum, unsigned long long *__restrict__ rows)
{
	const int i = blockIdx.x * 256 + threadIdx.x;
	uint64_t s = 0, r = 0;
	if (i < n) {
		const int na = n_aln[i];
		const uint4 *src = rows_of(i, aln, aln_cap, status, wide_idx, aln2, aln_cap2, grown);
		r = (uint64_t)na;
		s = mix64(((uint64_t)i << 20) ^ (uint64_t)na ^ 0x9e3779b97f4a7c15ULL);
		for (int j = 0; j < na; ++j) {
			const uint4 h = src[j];
			s += mix64(((uint64_t)i << 32 | (uint32_t)j) ^ mix64((uint64_t)h.x << 32 | h.y) ^ mix64((uint64_t)h.z << 32 | h.w) * 3ULL);
		}
	}
	for (int o = 32; o > 0; o >>= 1) { s += __shfl_down(s, o); r += __shfl_down(r, o); }
	if ((threadIdx.x & 63) == 0) { atomicAdd(sum, (unsigned long long)s); atomicAdd(rows, (unsigned long long)r); }
}

extern "C" void nabwa_launch_checksum(int n, const int32_t *n_aln, const uint4 *aln, int aln_cap, const uint8_t *status,
									  const int32_t *wide_idx, const uint4 *aln2, int aln_cap2,
									  const uint4 *const *grown, unsigned long long *sum, unsigned long long *rows, hipStream_t s)
{
	if (n <= 0) return;
	hipLaunchKernelGGL(checksum_kernel, dim3((n + 255) / 256), dim3(256), 0, s, n, n_aln, aln, aln_cap, status, wide_idx,
					   aln2, aln_cap2, grown, sum, rows);
}

// Work order of the search kernel by kernel W's class of a read = the smaller of its two strands' restart counts, i.e. a
// lower bound of the differences of its best hit: 0 (an exact occurrence exists), 1, 2+.  A search can take 10^4 dependent
// trips (tens of ms; the launch cannot end before its longest search does), so the long searches must START first:
// order 2+, 1, 0.  Class 0 reads run alike and close the launch in lockstep waves (64 reads at a time); the others go lane
// by lane.  (Tried: class 1 in lockstep waves -- its reads differ too much; a finer order 4+, 3, 2, 1, 0 -- the few hundred
// longest searches then sit in ONE wave's first block and run one after the other; those reads one per wave, or their waves
// at raised issue priority -- a trip is no shorter for it.  Mixed into the 2+ class they start in ~650 different waves at once.)
// cnt: [0..2] class sizes (pass 1), [5..7] cursors (pass 2), [10] = first work item of class 0 (*n_sync of the search
// kernel).  One wave handles 1024 consecutive reads with one atomic per class.
#define NABWA_N_CLASS 3
template <int PASS>
__global__ __launch_bounds__(256) void partition_kernel(int n, const uint8_t *__restrict__ cls, int32_t *__restrict__ ids, unsigned int *__restrict__ cnt)
{
	const unsigned int lane = threadIdx.x & 63u;
	const long wave = ((long)blockIdx.x * 256 + threadIdx.x) >> 6;
	const long first = wave * 1024;
	if (PASS == 2 && blockIdx.x == 0 && threadIdx.x == 0) { unsigned int x = 0; for (int q = 1; q < NABWA_N_CLASS; ++q) x += cnt[q]; cnt[10] = x; }
	if (first >= n) return;
	unsigned int num[NABWA_N_CLASS]; int cl[16];
#pragma unroll
	for (int q = 0; q < NABWA_N_CLASS; ++q) num[q] = 0;
#pragma unroll
	for (int g = 0; g < 16; ++g) {
		const long i = first + g * 64 + lane;
		const uint8_t c0 = i < n ? cls[2 * i] : 0, c1 = i < n ? cls[2 * i + 1] : 0;
		cl[g] = i < n ? (int)(c0 < c1 ? c0 : c1) : -1;
		if (cl[g] > NABWA_N_CLASS - 1) cl[g] = NABWA_N_CLASS - 1;
#pragma unroll
		for (int q = 0; q < NABWA_N_CLASS; ++q) num[q] += (unsigned int)__popcll(__ballot(cl[g] == q));
	}
	if (PASS == 1) {
		if (lane == 0) for (int q = 0; q < NABWA_N_CLASS; ++q) if (num[q]) atomicAdd(cnt + q, num[q]);
		return;
	}
	unsigned int base[NABWA_N_CLASS];
#pragma unroll
	for (int q = 0; q < NABWA_N_CLASS; ++q) base[q] = 0;
	if (lane == 0) {
		unsigned int start = 0;
		for (int q = NABWA_N_CLASS - 1; q >= 0; --q) { if (num[q]) base[q] = start + atomicAdd(cnt + 5 + q, num[q]); start += cnt[q]; }
	}
#pragma unroll
	for (int q = 0; q < NABWA_N_CLASS; ++q) base[q] = __shfl(base[q], 0);
#pragma unroll
	for (int g = 0; g < 16; ++g) {
		const long i = first + g * 64 + lane;
		const unsigned long long below = (1ull << lane) - 1ull;
#pragma unroll
		for (int q = 0; q < NABWA_N_CLASS; ++q) {
			const unsigned long long mq = __ballot(cl[g] == q);
			if (cl[g] == q) ids[base[q] + (unsigned int)__popcll(mq & below)] = (int32_t)i;
			base[q] += (unsigned int)__popcll(mq);
		}
	}
}

extern "C" void nabwa_launch_partition(int n, const uint8_t *cls, int32_t *ids, unsigned int *cnt, hipStream_t s)
{
	if (n <= 0) return;
	hipLaunchKernelGGL(partition_kernel<1>, dim3((n + 4095) / 4096), dim3(256), 0, s, n, cls, ids, cnt);
	hipLaunchKernelGGL(partition_kernel<2>, dim3((n + 4095) / 4096), dim3(256), 0, s, n, cls, ids, cnt);
}
