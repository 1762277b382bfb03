// fm_deep_body.hpp -- kernel D: bwt_match_gap (bwtgap.c:104-266) for the DEEP searches, one search per wavefront.
//
// Why.  One read per lane (kernel S, fm_search.hip) is right while searches are short: 260 k of them hide each other's
// latency.  A deep search (ancient-DNA options, reads with many differences: 10^5 .. 10^7 pops) is a dependent chain of
// pops on ONE lane at 2-3 us each, needs an arena of up to max_entries + 1 entries of its own, and the launch waits for
// the longest of them (round 1: 31 s for one read, 0.001 of the roofline).  Kernel D gives such a search a whole wave:
//
//   * The per-score LIFO stacks (bwtgap.c:46-79) are kept as they are in the reference -- one array per score -- but
//     paged: 4 KB pages of 256 entries from ONE device-wide pool, linked per score level.  A search holds the pages its
//     live entries need, nothing is sized for the worst case per lane, and nothing is re-run in growing tiers.
//   * A ROUND pops the top W <= 64 entries of the lowest non-empty score s at once, one per lane (lane 0 = the top).
//     Each lane runs its entry's CHAIN: pre-checks, hit test / exact tail, expansion; the matching child (same score,
//     pushed last, hence the very next pop of the reference) continues the chain in registers; every other child has a
//     score > s and is STAGED: the step files one 64-byte RECORD in the lane's own buffer (the intervals of the four
//     possible next symbols, the parent, which groups of children it pushes) and counts the children per class.  Between
//     two hits the reference's loop body is a pure function of the popped entry, so the W chains are independent -- this
//     is where the parallelism comes from.
//   * COMMIT restores the reference's order exactly: the reference would have run chain 0 to its end, then chain 1, ...
//     so the staged children go to their score levels lane by lane (a prefix sum over the lanes gives every lane its
//     slots; within a lane in chain order): the 64 lanes share out the round's records and turn each into its entries.  A chain that ends in a hit (the search state changes: max_diff, the width
//     bounds through gap_shadow, best_score) or that fills its staging buffer invalidates the lanes above it: they are
//     dropped (their entries are still on the stack) and popped again in the next round.
//   * The live-entry count before every pop (bwtgap.c:139-140: statistic and cut-off) is exact: a lane tracks the count
//     relative to its first pop and its peak, the prefix sum of the lanes' net changes gives the absolute values; when
//     the cut-off falls inside a round, the lanes before it are committed and the search goes on one pop per round
//     ("careful") until the cut-off is reached at a round's start.
//
// The same source is compiled for gfx950 (fm_deep.hip) and, with NABWA_EMU, as a CPU emulation of one wave
// (tests/emu/, test infrastructure): see wave_spmd.hpp.
#pragma once
#include "fm_deep.hpp"
#include "wave_spmd.hpp"

// the entry a lane's chain works on, packed as it is stored (two words of fields: five registers less per lane than one per field)
struct DeepLane { uint32_t k, l; uint32_t i : 16, ldp : 16; uint32_t mm : 8, go : 8, ge : 8, state : 2, a : 1, unused_ : 5; };

#ifdef NABWA_EMU
#define DEEP_FN static
DEEP_FN uint4 deep_ld_global16(const uint4 *p) { return *p; }
#define DEEP_ATOMIC_ADD_U64(p, v) (*(p) += (v))
#define DEEP_CLOCK() 0ull
#else
#define DEEP_FN __device__ __forceinline__
/* a bucket array reached through a pointer that was rebuilt from two words of the LDS table is, to the compiler, a pointer to anywhere
 * (flat loads: they wait on two counters and consult the LDS and scratch apertures first); it points to global memory and says so */
typedef uint32_t deep_u32x4 __attribute__((ext_vector_type(4)));
DEEP_FN uint4 deep_ld_global16(const uint4 *p)
{
	const deep_u32x4 v = *(const deep_u32x4 __attribute__((address_space(1)))*)(uintptr_t)p;
	return make_uint4(v.x, v.y, v.z, v.w);
}
#define DEEP_ATOMIC_ADD_U64(p, v) atomicAdd((p), (v))
#define DEEP_CLOCK() ((unsigned long long)wall_clock64())      /* 100 MHz */
#endif

DEEP_FN uint4 deep_pack(uint32_t k, uint32_t l, int i, int ldp, int mm, int go, int ge, int state, int a, uint32_t cls)
{
	return make_uint4(k, l, (uint32_t)i | (uint32_t)ldp << 16,
					  (uint32_t)mm | (uint32_t)go << 8 | (uint32_t)ge << 16 | (uint32_t)state << 24 | (uint32_t)a << 26 | cls << 27);
}

DEEP_FN void deep_unpack(const uint4 &r, DeepLane &e)
{
	e.k = r.x; e.l = r.y; e.i = r.z & 0xffffu; e.ldp = r.z >> 16;
	e.mm = r.w & 0xffu; e.go = r.w >> 8 & 0xffu; e.ge = r.w >> 16 & 0xffu;
	e.state = r.w >> 24 & 3u; e.a = r.w >> 26 & 1u; e.unused_ = 0u;
}

// nabwa_occ4_pair (nabwa_dev.hpp) with the loads of BOTH buckets issued before either is used: one memory latency per query
DEEP_FN void deep_occ4_pair(const DevBwt &B, uint32_t kq, uint32_t lq, Occ4 &ck, Occ4 &cl)
{
	const uint32_t kp = kq - (kq >= B.primary ? 1u : 0u), lp = lq - (lq >= B.primary ? 1u : 0u);
	const bool kvalid = kq != 0xffffffffu, lvalid = lq != 0xffffffffu;
	const uint32_t bl = lvalid ? lp / NABWA_INTV : 0u, rl = lp - bl * NABWA_INTV;
	const uint32_t bkk = kvalid ? kp / NABWA_INTV : bl, rk = kp - bkk * NABWA_INTV;
	const uint4 *const pl = B.bk + (size_t)bl * 4, *const pk = B.bk + (size_t)bkk * 4;
	// (fetching both buckets whether or not they are the same one saves the branch and a dozen moves -- and was 15 % slower on the ancient-DNA
	// workload: what the chain step pays for is memory requests, not instructions; measured in round 3, profiles/r03_deep_variants.txt)
	const uint4 a0 = deep_ld_global16(pl), a1 = deep_ld_global16(pl + 1), a2 = deep_ld_global16(pl + 2), a3 = deep_ld_global16(pl + 3);
	uint4 b0 = a0, b1 = a1, b2 = a2, b3 = a3;
	if (bkk != bl) { b0 = deep_ld_global16(pk); b1 = deep_ld_global16(pk + 1); b2 = deep_ld_global16(pk + 2); b3 = deep_ld_global16(pk + 3); }
	if (lvalid) cl = nabwa_count4(a0, a1, a2, a3, rl); else { cl.c[0] = cl.c[1] = cl.c[2] = cl.c[3] = 0; }
	if (kvalid) ck = nabwa_count4(b0, b1, b2, b3, rk); else { ck.c[0] = ck.c[1] = ck.c[2] = ck.c[3] = 0; }
}
// the 2 low bits of each of 8 bytes, byte j -> bits 2j
DEEP_FN uint32_t deep_squeeze(uint64_t v)
{
	uint64_t y = v & 0x0303030303030303ull;
	y = (y | y >> 6) & 0x000F000F000F000Full; y = (y | y >> 12) & 0x000000FF000000FFull; y = (y | y >> 24) & 0xFFFFull;
	return (uint32_t)y;
}
DEEP_FN int deep_ctz64(uint64_t m) { return __ffsll((unsigned long long)m) - 1; }
// element c of a four-element array by selects: a dynamically indexed register array would live in scratch memory
DEEP_FN uint32_t deep_sel4(const uint32_t (&a)[4], uint32_t c) { return c == 0u ? a[0] : (c == 1u ? a[1] : (c == 2u ? a[2] : a[3])); }

// the constants of the index an entry of strand a is searched on, from the wave's LDS table
DEEP_FN void deep_index_of(const uint32_t *s_bc, uint32_t a, DevBwt &B)
{
	const uint32_t *const c = s_bc + (a ? 0u : 12u);
	const uint4 lo = *(const uint4*)c, hi = *(const uint4*)(c + 4);
	B.bk = (const uint4*)(uintptr_t)((uint64_t)lo.y << 32 | lo.x); B.primary = lo.z; B.seq_len = lo.w;
	B.L2[0] = 0; B.L2[1] = hi.x; B.L2[2] = hi.y; B.L2[3] = hi.z;
}

// the interval table (all levels back to back, DEEP_LVO) of the index an entry of strand a is searched on
DEEP_FN const uint2 *deep_table_of(const uint32_t *s_bc, uint32_t a)
{
	const uint2 p = *(const uint2*)(s_bc + (a ? 8u : 20u));
	return (const uint2*)(uintptr_t)((uint64_t)p.y << 32 | p.x);
}
DEEP_FN uint2 deep_ld_global8(const uint2 *p)
{
#ifdef NABWA_EMU
	return *p;
#else
	typedef uint32_t u32x2_ __attribute__((ext_vector_type(2)));
	const u32x2_ v = *(const u32x2_ __attribute__((address_space(1)))*)(uintptr_t)p;
	return make_uint2(v.x, v.y);
#endif
}

// One wave: takes reads from the work counter until it runs out.  lds: 2 * NS + DEEP_NEWP words of this wave.
// PROF: the statistics (rounds, chains, phase clocks) cost scalar and vector registers, so the build without them is the one that runs
// unless NABWA_TIMING / NABWA_DEEP_STATS ask for them
// LDSM: the read's own data (bound bytes, seed bound bytes, bases) sit in the wave's LDS -- every read that fits (NABWA_DEEP_LDS_MAX); the other
// instantiation reads them where kernel W / the batch put them.  A template flag, not a runtime one: the chain step consults these
// bytes six times, and both ways of getting at them were in its code.
// COOP: with the wave-wide expansion of one-row chains (DF_COOP) -- for batches of reads long enough to have such chains (the host decides)
template <bool PROF, bool LDSM, bool COOP>
DEEP_FN void deep_wave_body(const DeepParams &P_, uint32_t *lds, uint32_t wave
#ifndef NABWA_EMU
							, const int ln
#endif
							)
{
	const DeepParams P = P_;          /* (a by-value copy: the kernel argument stays in scalar registers / the kernarg segment; splitting its tuples
	                                   * with nabwa_dev.hpp's own() as fm_search_kernel does was measured here too: no gain) */
	const SearchParams &S = P.S;
	// this wave's LDS: per score level the entry count and the top page; the pages a commit takes; and (lds_rd > 0) the read's
	// own data -- bound bytes, seed bound bytes, bases of both strands -- copied in when the read starts
	const uint32_t ns2 = (P.NS + 1u) & ~1u;
	uint32_t *const s_cnt = lds, *const s_top = lds + ns2, *const s_newp = lds + 2 * ns2;
	uint32_t *const s_off = s_newp + DEEP_NEWP;                           // 4 x 64 words: per lane the offsets of its records and of its children in the three classes (commit)
	// the constants of the two indexes, picked per lane by the strand of its entry: one 16-byte LDS read each instead of a chain of selects
	// over scalar registers that have long been spilled
	uint32_t *const s_bc = s_off + 256;
	uint8_t *const s_bb = (uint8_t*)(s_bc + DEEP_BC_WORDS), *const s_sb = s_bb + 2 * S.WLB, *const s_sq = s_sb + 2 * S.SLB;
	constexpr bool lds_mode = LDSM;
	const uint32_t PL = P.rd_pl;
	uint32_t *const own = P.own + (size_t)wave * 2 * P.own_cap, *const freep = own + P.own_cap;
	uint4 *const stage = P.stage + (size_t)wave * 64 * P.stage_k * 4;     // [lane][stage_k] records of 64 bytes
	const uint32_t K = P.stage_k;
	uint32_t n_own = 0, n_free = 0;
	const bool gape_mode = S.mode & 0x01, nonstop = S.mode & 0x10, loggap = S.mode & 0x04;
	unsigned long long st_rounds = 0, st_run = 0, st_commit = 0, st_steps = 0, st_careful = 0, st_pool = 0;
	unsigned long long st_maxclk = 0, st_maxrounds = 0, st_sumclk = 0;
	uint32_t coop_n = 0, coop_lv = 0, coop_skip = 0;      // the wave's one-row chains so far: taken over, their levels; steps since that stopped paying
	unsigned long long st_lanesteps = 0, st_coop = 0, st_cooplev = 0;      // ... wave-wide expansions of one-row chains, levels they took
	// (statistics build) per lane: expansions of entries in key form / their tail jumps and hits / records filed / children stored / entries pruned at the pop /
	// expansions / rank queries on two buckets / expansions that may push no difference (allow_diff false): in key form, on one row, on several rows
	LANE(uint32_t, pk_key); LANE(uint32_t, pk_ktl); LANE(uint32_t, pk_rec); LANE(uint32_t, pk_chl); LANE(uint32_t, pk_prn); LANE(uint32_t, pk_exp); LANE(uint32_t, pk_two);
	LANE(uint32_t, pk_fk); LANE(uint32_t, pk_f1); LANE(uint32_t, pk_fw);
	LANES { L(pk_key) = L(pk_ktl) = L(pk_rec) = L(pk_chl) = L(pk_prn) = L(pk_exp) = L(pk_two) = L(pk_fk) = L(pk_f1) = L(pk_fw) = 0u; }
	unsigned long long ph_pop = 0, ph_chain = 0, ph_tail = 0, ph_commit = 0, ph_hit = 0, st_tailit = 0;      // (statistics) time per phase of a round
	const bool prof = PROF && P.stats != 0;      // the longest single read of this wave: time, rounds; time in reads altogether
	const unsigned long long clk_start = DEEP_CLOCK();
	// text mode (nabwa_dev.hpp): an exact tail that has narrowed to ONE row is finished by comparing the read with the text
	const bool text_ok = (S.text_mode & 2) && S.bwt[0].sa_full && S.bwt[1].sa_full && S.bwt[0].isa && S.bwt[1].isa && S.bwt[0].text && S.bwt[1].text;
	LANE(int, ts); LANE(uint32_t, tpos);             // exact tails: where a parked tail stands, the text position of its one row
	LANE(uint32_t, ntl); LANE(uint32_t, ntx);      // statistics: rank steps / text finishes of this lane's exact tails
	LANES { L(ntl) = 0; L(ntx) = 0; }
	LANES { if (ln < 24) {      // s_bc[12 q + ..] = what a search on index q needs, out of the table the host made (SearchParams.ixtab): bucket array, primary,
		// seq_len, L2[1..3], a spare -- words 0, 1, 12 .. 17 of the index's block there --, the interval table (words 4, 5), two spares.
		// Entries of strand a search index 1 - a (bwtgap.c:149)
		const uint32_t q = (uint32_t)ln >= 12u ? 1u : 0u, w = (uint32_t)ln - 12u * q;
		s_bc[ln] = S.ixtab[q * NABWA_IXTAB_STRIDE + (w < 2u ? w : (w < 8u ? w + 10u : (w < 10u ? IX_KMER_LO + w - 8u : 17u)))];
	} }
	WAVE_SYNC();

	LANE(uint32_t, tu);       // scratch for broadcasts
	LANE(DeepLane, e);
	LANE(bool, act);
	LANE(int, flag);
	LANE(uint32_t, cc0); LANE(uint32_t, cc1); LANE(uint32_t, cc2);   // children staged (through records), per (canonical) class
	LANE(uint32_t, nrec);     // records this lane's chain has filed in this round
	LANE(bool, norun);        // this lane's entry takes its forced levels step by step (its row is the empty suffix's)
	LANE(int, rel);           // live entries relative to the count before this lane's first pop
	LANE(int, peak);          // the largest value `rel` had right before a pop
	LANE(uint32_t, d); LANE(uint32_t, off);
	LANE(int, nst);
	LANE(uint32_t, tch);      // bucket touches of the reference algorithm in this lane's chain (instrumented runs only)
	const bool counting = PROF && S.touch_counter != 0;      // (the touch-counting run uses the statistics instantiation: one per-lane counter less to carry)
	const uint32_t KT = counting ? 0u : P.key_T;             // key-form entries (fm_deep.hpp) while their strings are shorter than this; the counted touches are those of rows

	// the read's own data: from LDS, or (reads too long for it) from where kernel W / the batch put them
#define DEEP_BB(a_, p_) (lds_mode ? (uint32_t)s_bb[(uint32_t)(a_) * S.WLB + (uint32_t)(p_)] : (uint32_t)(rec + S.woff_bid)[(uint32_t)(a_) * S.WLB + (uint32_t)(p_)])
#define DEEP_SB(a_, p_) (lds_mode ? (uint32_t)s_sb[(uint32_t)(a_) * S.SLB + (uint32_t)(p_)] : (uint32_t)(rec + S.woff_sbid)[(uint32_t)(a_) * S.SLB + (uint32_t)(p_)])
#define DEEP_RD(a_, p_) (lds_mode ? (uint32_t)s_sq[(uint32_t)(a_) * PL + (uint32_t)(p_)] : (uint32_t)((a_) ? S.rseq : S.seq)[sq_off + (size_t)(p_)])
#define DEEP_RD16(a_, p_) (lds_mode ? *(const uint4*)(s_sq + (uint32_t)(a_) * PL + (uint32_t)(p_)) : *(const uint4*)(((a_) ? S.rseq : S.seq) + sq_off + (size_t)(p_)))

	// the bounds of the prefix still to match and of the seed for an expansion at read position p_ (bwtgap.c:205-215): ad_ = allow_diff, am_ = allow_M
#define DEEP_BOUNDS(a_, p_, m_, ms_, ad_, am_) do { \
		const bool in_ = (p_) > 0; \
		const uint32_t B1_ = DEEP_BB(a_, in_ ? (p_) - 1 : 0), B0_ = DEEP_BB(a_, p_); \
		const int b1_ = (int)(B1_ & 127u), b0_ = (int)(B0_ & 127u); \
		const bool no_d_ = in_ && b1_ > (m_) - 1, no_m_ = in_ && b1_ == (m_) - 1 && b0_ == (m_) - 1 && (B0_ & 128u) != 0u; \
		const int ii_ = (p_) - (len - S.seed_len); \
		const bool sd_ = seeded && in_ && ii_ > 0; \
		const uint32_t S1_ = DEEP_SB(a_, sd_ ? ii_ - 1 : 0), S0_ = DEEP_SB(a_, sd_ ? ii_ : 0); \
		const int s1_ = (int)(S1_ & 127u), s0_ = (int)(S0_ & 127u); \
		const bool no_ds_ = sd_ && s1_ > (ms_) - 1, no_ms_ = sd_ && s1_ == (ms_) - 1 && s0_ == (ms_) - 1 && (S0_ & 128u) != 0u; \
		ad_ = !(no_d_ || no_ds_); \
		am_ = !((!no_d_ && no_m_) || (!no_ds_ && no_ms_)); } while (0)
	// an expansion at read position p_ that may push no difference (bwtgap.c:205-215: allow_diff = 0), for an entry with m_ differences left (ms_ in the seed)
#define DEEP_FORCED(a_, p_, m_, ms_) ((p_) > 0 && ((int)(DEEP_BB(a_, (p_) - 1) & 127u) > (m_) - 1 || \
		(seeded && (p_) - (len - S.seed_len) > 0 && (int)(DEEP_SB(a_, (p_) - (len - S.seed_len) - 1) & 127u) > (ms_) - 1)))

	// n_new pages into s_newp[]: from this wave's free ones first, then from the pool; ok_ = false when the pool is dry
#define DEEP_ALLOC(n_new_, ok_) do { \
		const uint32_t nn_ = (n_new_); \
		const uint32_t take_ = nn_ < n_free ? nn_ : n_free; \
		LANES { for (uint32_t t_ = (uint32_t)ln; t_ < take_; t_ += 64u) s_newp[t_] = freep[n_free - 1u - t_]; } \
		n_free -= take_; \
		const uint32_t rest_ = nn_ - take_; \
		ok_ = true; \
		if (rest_) { \
			LANES { L(tu) = 0; if (ln == 0) L(tu) = ATOMIC_ADD_U32(P.page_bump, rest_); } \
			const uint32_t base_ = WUNI(WBCAST(tu, 0)); \
			if ((uint64_t)base_ + rest_ > P.n_pages || n_own + rest_ > P.own_cap) ok_ = false; \
			else { \
				LANES { for (uint32_t t_ = (uint32_t)ln; t_ < rest_; t_ += 64u) { s_newp[take_ + t_] = base_ + t_; own[n_own + t_] = base_ + t_; } } \
				n_own += rest_; \
			} \
		} \
		WAVE_SYNC(); } while (0)

	for (;;) {
		LANES { L(tu) = 0; if (ln == 0) L(tu) = ATOMIC_ADD_U32(S.work_counter, 1u); }
		const uint32_t idx = WUNI(WBCAST(tu, 0));
		if (idx >= (uint32_t)S.n) break;
		const uint32_t rid = WUNI(S.ids ? (uint32_t)S.ids[idx] : idx);
		const uint32_t item = WUNI(S.res_slot ? (uint32_t)S.res_slot[rid] : idx);
		const int len = WUNI(S.rd_len[rid]);
		const size_t sq_off = (size_t)WUNI((uint32_t)S.poff[rid]);      /* (< 4 Gi padded bases per batch: nabwa_batch_create) */
		const int MD = WUNI((int)S.rd_maxdiff[rid]), MG = WUNI((int)S.rd_maxgapo[rid]);
		uint8_t *const rec = S.wdata + (size_t)rid * S.wstride;
		uint4 *const out = S.aln + (size_t)item * S.aln_cap;
		int n_aln = 0, max_ent = 0, status = NABWA_ST_OK;
		unsigned long long rd_touch = 0;
		const unsigned long long clk0 = prof ? DEEP_CLOCK() : 0ull, rounds0 = st_rounds;

		if (len > 0 && (int)S.rd_nN[rid] <= MD) {          // too many N: no search (bwtgap.c:118-123)
			const bool seeded = len > S.seed_len;
			int max_diff = MD, best_cnt = 0, n_entries = 0;
			int best_score = (MD + 1) * S.s_mm + (MG + 1) * S.s_gapo + (S.max_gape + 1) * S.s_gape;
			LANES { for (uint32_t t = (uint32_t)ln; t < P.NS; t += 64u) { s_cnt[t] = 0; s_top[t] = DEEP_NIL; } }
			LANES { for (uint32_t t = (uint32_t)ln; t < n_own; t += 64u) freep[t] = own[t]; }      // every page this wave holds is free again
			n_free = n_own;
			if (lds_mode) {
				const uint32_t nb16 = (2u * S.WLB + 2u * S.SLB) / 16u, pl16 = ((uint32_t)len + 15u) / 16u;
				LANES {
					for (uint32_t t = (uint32_t)ln; t < nb16; t += 64u) ((uint4*)s_bb)[t] = ((const uint4*)(rec + S.woff_bid))[t];
					for (uint32_t t = (uint32_t)ln; t < 2u * pl16; t += 64u) {
						const uint32_t x = t >= pl16 ? 1u : 0u, c = t - x * pl16;
						((uint4*)(s_sq + x * PL))[c] = ((const uint4*)((x ? S.rseq : S.seq) + sq_off))[c];
					}
				}
			}
			WAVE_SYNC();
			bool done = false, careful = P.careful_all != 0;
			uint32_t cur = 0;                                 // no level below `cur` holds an entry
			// roots (bwtgap.c:127-128): strand 0 is pushed first, strand 1 second, so strand 1 is popped first
			bool got_page = false;
			DEEP_ALLOC(1u, got_page);
			if (!got_page) { status = NABWA_ST_POOL; done = true; if (PROF) ++st_pool; }
			else {
				const uint32_t p0 = WUNI(s_newp[0]);
				ONE_LANE {
					const uint32_t root_l = KT ? DEEP_KEYL : S.bwt[0].seq_len;      // the empty string: every row, or key form of length 0
					P.pages[(size_t)p0 * DEEP_PAGE + 0] = deep_pack(0u, root_l, len, 0, 0, 0, 0, DST_M, 0, 0u);
					P.pages[(size_t)p0 * DEEP_PAGE + 1] = deep_pack(0u, root_l, len, 0, 0, 0, 0, DST_M, 1, 0u);
					P.page_prev[p0] = DEEP_NIL; s_cnt[0] = 2; s_top[0] = p0;
				}
				WAVE_SYNC();
				n_entries = 2;
			}

			while (!done) {
				// ---------------------------------------------------------------- start of a round = the reference's loop head
				if (n_entries == 0) break;
				if (max_ent < n_entries) max_ent = n_entries;                     // bwtgap.c:139
				if (n_entries > S.max_entries) break;                             // bwtgap.c:140
				bool found = false;
				for (uint32_t base = cur; base < P.NS && !found; base += 64u) {
					const uint64_t mk = WBALLOT(base + (uint32_t)ln < P.NS && s_cnt[base + (uint32_t)ln] != 0u);
					if (mk) { cur = base + (uint32_t)deep_ctz64(mk); found = true; }
				}
				if (!found) break;     // only children that were counted but never stored are left: the reference pops one of them and stops (bwtgap.c:144)
				const unsigned long long pc0 = prof ? DEEP_CLOCK() : 0ull;
				const int s = (int)cur;
				if (!nonstop && n_aln > 0 && s > best_score + S.s_mm) break;      // bwtgap.c:144
				const uint32_t cs = WUNI(s_cnt[s]);
				uint32_t W = careful ? 1u : (uint32_t)P.max_lanes;
				if (W > cs) W = cs;
				const int T0 = s + S.s_mm, T1 = s + S.s_gapo, T2 = s + S.s_gape;
				const uint32_t can1 = T1 == T0 ? 0u : 1u, can2 = T2 == T0 ? 0u : (T2 == T1 ? can1 : 2u);
				const bool keep_all = nonstop || n_aln == 0;
				const bool keep0 = keep_all || T0 <= best_score + S.s_mm, keep1 = keep_all || T1 <= best_score + S.s_mm, keep2 = keep_all || T2 <= best_score + S.s_mm;
				const uint32_t topq = (cs - 1u) >> DEEP_PAGE_SH, top_pg = WUNI(s_top[s]);
				uint32_t prev_pg = DEEP_NIL;
				if (((cs - W) >> DEEP_PAGE_SH) != topq) prev_pg = WUNI(P.page_prev[top_pg]);
				LANES {
					L(act) = (uint32_t)ln < W; L(flag) = DF_NONE; L(tch) = 0; L(cc0) = L(cc1) = L(cc2) = 0; L(nrec) = 0; L(norun) = false; L(rel) = 0; L(peak) = 0;
					if (L(act)) {
						const uint32_t p = cs - 1u - (uint32_t)ln;
						const uint32_t pg = (p >> DEEP_PAGE_SH) == topq ? top_pg : prev_pg;
						deep_unpack(P.pages[(size_t)pg * DEEP_PAGE + (p & (DEEP_PAGE - 1u))], L(e));
					}
				}
				if (PROF) { ++st_rounds; st_run += W; if (careful) ++st_careful; }
#ifndef NABWA_EMU
				// (statistics build, NABWA_DEEP_HIST=2: rounds and wave-steps by the round's width -- 1, 2, 3-4, 5-8, ... 33-64 entries)
				const uint32_t wbin = W <= 1u ? 0u : 32u - (uint32_t)__clz((int)(W - 1u));
				if (PROF && P.stats && P.hist == 2) { ONE_LANE { atomicAdd(P.stats + 32 + wbin, 1ull); } }
#endif

				unsigned long long pc1 = 0; if (prof) { LANES { L(tu) = L(e).k; } (void)WUNI(WBCAST(tu, 0)); pc1 = DEEP_CLOCK(); ph_pop += pc1 - pc0; }
				// ---------------------------------------------------------------- the chains (and, when forced levels were walked on the text for
				// some of them, the chains again: those lanes come back with their entries at the end of the walk)
				unsigned long long pc2 = 0;
				for (;;) {
				while (WBALLOT(L(act)) != 0ull) {
					const uint32_t n_act = (uint32_t)__popcll((unsigned long long)WBALLOT(L(act)));
					// few chains left in this round: the long ones -- worth the wave's while (DF_COOP, below), as long as the chains it took so far
					// went on for 4 levels and more on average (reads of 50 bases do not: their chains end after a level or two, and the wave's
					// time is better spent on chain steps; asked again after a while -- a batch may mix libraries)
					if (COOP && coop_n >= 16u && coop_lv < 4u * coop_n && ++coop_skip >= 4096u) { coop_n = coop_lv = coop_skip = 0u; }
					const bool coop_ok = COOP && n_act <= P.coop_lanes && (coop_n < 16u || coop_lv >= 4u * coop_n);
					if (PROF) { ++st_steps; st_lanesteps += n_act; }
#ifndef NABWA_EMU
					if (PROF && P.stats && P.hist == 2) { ONE_LANE { atomicAdd(P.stats + 64 + wbin, 1ull); } }
#endif
					LANES { if (L(act)) {
						DeepLane &E = L(e);
						// ---- what the reference does with a popped entry (bwtgap.c:141-164)
						if (L(rel) > L(peak)) L(peak) = L(rel);
						L(rel) -= 1;
						const int m = max_diff - E.mm - E.go - (gape_mode ? E.ge : 0);
						// the fates of a popped entry, decided side by side (bwtgap.c:141-164): pruned (m < 0, or below the bound of the prefix
						// still to match, :156) / a hit (nothing left to match) / an exact tail (nothing may differ any more: bwt_match_exact_alt,
						// bwt.c:237-252 -- parked here and walked after the loop together with the other chains' tails: a tail is a run of
						// dependent loads, and inside this loop every other lane of the wave would wait for each of them) / an expansion
						const uint32_t bprev = DEEP_BB(E.a, E.i > 0 ? E.i - 1 : 0) & 127u;
						const bool go_on = m >= 0 && !(E.i > 0 && m < (int)bprev);
						const bool hit = go_on && E.i == 0;
						const bool tail = go_on && !hit && m == 0 && (E.state == DST_M || gape_mode || E.ge == S.max_gape);
						if (hit) L(flag) = DF_HIT; else if (tail) L(flag) = DF_TAIL;
						if (PROF && !go_on) L(pk_prn) += 1u;
						if (!go_on || hit || tail) L(act) = false;
						else {
							{
								// ---- expansion (bwtgap.c:201-260)
								const int i = E.i - 1;
								const bool kf = E.l >= DEEP_KEYL;                                // key form: the four extensions come out of the table
								const uint32_t kt = E.l & 0xffu;
								uint32_t occ;
								uint32_t nk0, nl0, nk1, nl1, nk2, nl2, nk3, nl3;
								if (PROF) L(pk_exp) += 1u;
#ifndef NABWA_EMU
								if (PROF && P.stats && P.hist == 1) { const uint32_t dd = (uint32_t)(len - i - 1) < 31u ? (uint32_t)(len - i - 1) : 31u; atomicAdd(P.stats + 32 + (kf ? 64u : (E.k == E.l ? 32u : 0u)) + dd, 1ull); }
#endif
								const int m_seed = S.max_seed_diff - E.mm - E.go - (gape_mode ? E.ge : 0);
								bool allow_diff = true, allow_M = true;
								{	// the bounds of the prefix still to match and of the seed (bwtgap.c:205-215), read with clamped positions and applied by
									// predicates: no branch, nothing to merge afterwards
									const bool in = i > 0;
									const uint32_t B1 = DEEP_BB(E.a, in ? i - 1 : 0), B0 = DEEP_BB(E.a, i);
									const int b1 = (int)(B1 & 127u), b0 = (int)(B0 & 127u);
									const bool no_d = in && b1 > m - 1, no_m = in && b1 == m - 1 && b0 == m - 1 && (B0 & 128u) != 0u;
									const int ii = i - (len - S.seed_len);
									const bool sd = seeded && in && ii > 0;
									const uint32_t S1 = DEEP_SB(E.a, sd ? ii - 1 : 0), S0 = DEEP_SB(E.a, sd ? ii : 0);
									const int s1 = (int)(S1 & 127u), s0 = (int)(S0 & 127u);
									const bool no_ds = sd && s1 > m_seed - 1, no_ms = sd && s1 == m_seed - 1 && s0 == m_seed - 1 && (S0 & 128u) != 0u;
									allow_diff = !(no_d || no_ds);
									allow_M = !((!no_d && no_m) || (!no_ds && no_ms));
								}
								// FORCED LEVELS.  Where no difference may be pushed (allow_diff = 0) an expansion pushes the matching child and nothing else
								// (bwtgap.c:252-258), and that child is the next pop: a stretch of such levels is an exact walk along the read, without a
								// record and with the live-entry count where it was.  It dies where the string stops occurring, where a symbol is an N, or
								// where the child is pruned at its pop (bwtgap.c:156) -- all without a trace -- and it ends in front of the first level that
								// may push a difference, at the read's end (a hit), where nothing may differ any more (an exact tail), or where the table ends.
								// An entry in key form takes the whole stretch with ONE load: the interval of its string plus the stretch's symbols.  An entry
								// on ONE row takes it from the text (parked like an exact tail: DF_RUN, walked in the loop behind this one, then back here).
								bool walked = false;
								if (!allow_diff && kf) {
									walked = true;
									if (PROF) L(pk_fk) += 1u;
									uint32_t key = E.k, t = kt; int p = i; bool die = false;
									for (;;) {
										const uint32_t c = DEEP_RD(E.a, p);
										if (c > 3u) { die = true; break; }
										key = key << 2 | c; ++t;
										if (p == 0) break;
										if (m < (int)(DEEP_BB(E.a, p - 1) & 127u)) { die = true; break; }
										if (m == 0 || t >= KT || !DEEP_FORCED(E.a, p - 1, m, m_seed)) break;
										--p;
									}
									if (!die) {
										const uint2 r = deep_ld_global8(deep_table_of(s_bc, E.a) + ((size_t)DEEP_LVO(t) + key));
										if (r.x > r.y) die = true;
										else {
											L(rel) += 1; E.i = p; E.ldp = 0; E.state = DST_M;
											if (t < KT) { E.k = key; E.l = DEEP_KEYL | t; } else { E.k = r.x; E.l = r.y; }
										}
									}
									if (die) L(act) = false;
								} else if (kf && coop_ok) {
									// (a chain in key form that may push differences, few chains left: the wave takes its levels down to the table's depth together -- the
									// two chains of a search's first round are such: a dozen steps with two lanes at work, a quarter of the PE workload's wave-steps)
									walked = true; L(flag) = DF_COOP; L(act) = false;
								} else if (text_ok && E.k == E.l && !L(norun) && (!allow_diff || coop_ok)) {
									// (forced levels: DF_RUN, every lane walks its own stretch; levels that may push differences: DF_COOP, the wave takes
									// the chain's next levels together, 64 at a time -- see behind the tails' loop)
									walked = true; L(flag) = allow_diff ? DF_COOP : DF_RUN; L(act) = false;
									if (PROF) L(pk_f1) += 1u;
								}
								if (!walked) {
									if (kf) {
										if (PROF) L(pk_key) += 1u;
										const uint2 *const tab = deep_table_of(s_bc, E.a);
										const uint4 *const ch = (const uint4*)(tab + ((size_t)DEEP_LVO(kt + 1u) + (size_t)E.k * 4u));
										const uint4 c01 = deep_ld_global16(ch), c23 = deep_ld_global16(ch + 1);
										nk0 = c01.x; nl0 = c01.y; nk1 = c01.z; nl1 = c01.w; nk2 = c23.x; nl2 = c23.y; nk3 = c23.z; nl3 = c23.w;
										occ = 0xffffffffu;
										// the one place an expansion asks how many rows its entry has (bwtgap.c:232: the extension of a deletion, once the gaps alone use up max_diff)
										if (E.state == DST_D && E.ge + E.go >= max_diff && kt) { const uint2 own = deep_ld_global8(tab + ((size_t)DEEP_LVO(kt) + E.k)); occ = own.y - own.x + 1u; }
									} else {
										DevBwt B;                                                       // the index searched: bwts[1 - a] (bwtgap.c:149)
										deep_index_of(s_bc, E.a, B);
										Occ4 ck, cl;
										deep_occ4_pair(B, E.k - 1u, E.l, ck, cl);
										if (counting) L(tch) += ref_touches(B, E.k - 1u, E.l, true);
										if (PROF && (E.k - 1u - (E.k - 1u >= B.primary ? 1u : 0u)) / NABWA_INTV != (E.l - (E.l >= B.primary ? 1u : 0u)) / NABWA_INTV) L(pk_two) += 1u;
										occ = E.l - E.k + 1u;
										// ---- the children (bwtgap.c:206-259).  Deletions, mismatches and the match over symbol x all have the interval
										// of "x in front of the suffixes": four intervals serve every child of this expansion.
										nk0 = B.L2[0] + ck.c[0] + 1u; nl0 = B.L2[0] + cl.c[0]; nk1 = B.L2[1] + ck.c[1] + 1u; nl1 = B.L2[1] + cl.c[1];
										nk2 = B.L2[2] + ck.c[2] + 1u; nl2 = B.L2[2] + cl.c[2]; nk3 = B.L2[3] + ck.c[3] + 1u; nl3 = B.L2[3] + cl.c[3];
									}
									if (PROF && !allow_diff) { if (kf) L(pk_fk) += 1u; else if (E.k == E.l) L(pk_f1) += 1u; else L(pk_fw) += 1u; }
									// The chain does not build the children: it files ONE 64-byte record -- the four intervals, the parent, which groups it
									// pushes and where in their classes' sequences they go -- and the commit, where all 64 lanes work whatever the
									// chains' lengths were, turns records into entries.  The counts are all the chain itself needs.
									const uint32_t vm = (nk0 <= nl0 ? 1u : 0u) | (nk1 <= nl1 ? 2u : 0u) | (nk2 <= nl2 ? 4u : 0u) | (nk3 <= nl3 ? 8u : 0u);
									const uint32_t nv = (uint32_t)__popc(vm);
									int tmp = E.go + E.ge;
									if (loggap) { const uint32_t v = (uint32_t)(E.ge + E.go); tmp = (v ? 31 - __clz((int)v) : 0) / 2 + 1; }
									uint32_t grp = 0, n_gap = 0, gcls = DCL_GO;      // DRG_* bits; children of the gap group; its class
									if (allow_diff && i >= S.indel_end_skip + tmp && len - i >= S.indel_end_skip + tmp) {
										if (E.state == DST_M) { if (E.go < MG) { grp = DRG_OPEN; n_gap = 1u + nv; } }                                 // the insertion, then the deletions
										else if (E.state == DST_I) { if (E.ge < S.max_gape) { grp = DRG_EXT_I; n_gap = 1u; gcls = DCL_GE; } }
										else if (E.ge < S.max_gape && (E.ge + E.go < max_diff || occ < (uint32_t)S.max_del_occ)) { grp = DRG_EXT_D; n_gap = nv; gcls = DCL_GE; }
									}
									const uint32_t c = DEEP_RD(E.a, i);
									uint32_t mmv = 0;                                   // symbols with a mismatch child
									bool match = false;
									if (allow_diff && allow_M) { mmv = c > 3u ? vm : vm & ~(1u << c); match = c <= 3u && (vm >> c & 1u); }
									else if (c < 4u) match = (vm >> c & 1u) != 0u;
									const uint32_t n_mm = (uint32_t)__popc(mmv);
									L(rel) += (int)(n_gap + n_mm);
									// a class that can never be popped (after the first hit the loop ends at the first pop above best_score + s_mm,
									// bwtgap.c:144) is counted, not stored
									const bool keep_gap = gcls == DCL_GO ? keep1 : keep2;
									if (!keep_gap) { grp = 0; n_gap = 0; }
									if (!keep0) mmv = 0;
									const uint32_t n_mm_st = keep0 ? n_mm : 0u;
									if (n_gap + n_mm_st) {
										const uint32_t gcn = gcls == DCL_GO ? can1 : can2;
										const uint32_t at_gap = gcn == 0u ? L(cc0) : (gcn == 1u ? L(cc1) : L(cc2));
										if (gcn == 0u) L(cc0) += n_gap; else if (gcn == 1u) L(cc1) += n_gap; else L(cc2) += n_gap;
										const uint32_t at_mm = L(cc0);
										L(cc0) += n_mm_st;
										uint4 *const rp = stage + ((size_t)ln * K + L(nrec)) * 4u;
										rp[0] = make_uint4(nk0, nk1, nk2, nk3);
										rp[1] = make_uint4(nl0, nl1, nl2, nl3);
										rp[2] = make_uint4(E.k, E.l, (uint32_t)i | vm << 16 | grp << 20 | mmv << 24 | gcn << 28,
														   (uint32_t)E.mm | (uint32_t)E.go << 8 | (uint32_t)E.ge << 16 | (uint32_t)E.a << 26 | (c & 7u) << 27);
										rp[3] = make_uint4(at_gap, at_mm, 0u, 0u);
										L(nrec) += 1u;
										if (PROF) { L(pk_rec) += 1u; L(pk_chl) += n_gap + n_mm_st; }
									}
									const uint32_t mk_ = c == 0u ? nk0 : (c == 1u ? nk1 : (c == 2u ? nk2 : nk3)), ml_ = c == 0u ? nl0 : (c == 1u ? nl1 : (c == 2u ? nl2 : nl3));
									// the matching child: same score, pushed last -> it is the reference's next pop: the chain goes on with it
									if (match) {
										L(rel) += 1; E.i = i; E.ldp = 0; E.state = DST_M;
										if (kf && kt + 1u < KT) { E.k = E.k << 2 | c; E.l = DEEP_KEYL | (kt + 1u); } else { E.k = mk_; E.l = ml_; }
									}
									else L(act) = false;
								}
								if (L(act) && (careful || L(nrec) >= K)) { L(flag) = DF_CONT; L(act) = false; }      // no room for another record: the chain goes on in the next round
							}
						}
					} }
					// a chain that ended in a hit or ran out of staging room: the lanes above it will be dropped
					const uint64_t sm = WBALLOT(L(flag) == DF_HIT || L(flag) == DF_CONT);
					if (sm) { const int j = deep_ctz64(sm); LANES { if (ln > j) { L(act) = false; if (L(flag) == DF_TAIL || L(flag) == DF_RUN || L(flag) == DF_COOP) L(flag) = DF_NONE; } } }
				}

				// ---------------------------------------------------------------- the parked exact tails, all together: every turn
				// of this loop is ONE memory round trip per lane -- a rank step over the next symbol (str[i-1]) while the interval
				// has several rows; once ONE row is left (text mode, nabwa_dev.hpp) its suffix's text position, then the comparison
				// of the i symbols still to match with the text right in front of it (str[j] against text[pos - i + j], 16 per
				// word pair, both packed low bits first), then the row of the extended suffix from the inverse suffix array
				if (prof) { pc2 = DEEP_CLOCK(); ph_chain += pc2 - pc1; }
				// (an entry in key form first takes its rows from the table: a tail jumps as far down as the table goes -- min(T - t, i) symbols
				// for one load --, a hit takes the rows of its own string)
				LANES {
					const bool kf = L(e).l >= DEEP_KEYL;
					L(ts) = L(flag) == DF_TAIL ? (kf ? 5 : ((text_ok && L(e).k == L(e).l) ? 1 : 0)) : (L(flag) == DF_HIT && kf ? 6 : (L(flag) == DF_COOP && kf ? 10 : (L(flag) == DF_RUN || L(flag) == DF_COOP ? 7 : -1)));
				}
				while (WBALLOT(L(ts) >= 0 && L(ts) != 10) != 0ull) {
					if (PROF) ++st_tailit;
					LANES { if (L(ts) >= 0 && L(ts) != 10) {
						DeepLane &E = L(e);
						const bool q1 = E.a == 0;
						bool fail = false, hit = false;
						if (L(ts) == 0 || L(ts) == 4) {
							DevBwt B;
							deep_index_of(s_bc, E.a, B);
							const uint32_t c = DEEP_RD(E.a, E.i - 1);
							if (c > 3u) fail = true;
							else {
								Occ4 ck, cl;
								deep_occ4_pair(B, E.k - 1u, E.l, ck, cl);
								if (counting) L(tch) += ref_touches(B, E.k - 1u, E.l, false);
								E.k = deep_sel4(B.L2, c) + deep_sel4(ck.c, c) + 1u; E.l = deep_sel4(B.L2, c) + deep_sel4(cl.c, c);
								if (PROF) L(ntl) += 1;
								if (E.k > E.l) fail = true;
								else if (--E.i == 0) hit = true;
								else if (L(ts) == 0 && text_ok && E.k == E.l) L(ts) = 1;
							}
						} else if (L(ts) == 1) {
							const uint32_t pos = (q1 ? S.bwt[1].sa_full : S.bwt[0].sa_full)[E.k];
							if (pos == 0xffffffffu) L(ts) = 4;                      // (the empty suffix: rank steps to the end)
							else if (pos < (uint32_t)E.i) fail = true;                // the text begins before the read does
							else { L(tpos) = pos; L(ts) = 2; }
						} else if (L(ts) == 2) {
							const uint32_t *const txt = q1 ? S.bwt[1].text : S.bwt[0].text;
							bool ok = true;
							for (int j0 = 0; ok && j0 < E.i; j0 += 16) {
								const uint4 q = DEEP_RD16(E.a, j0);
								const uint64_t lo = (uint64_t)q.y << 32 | q.x, hi = (uint64_t)q.w << 32 | q.z;
								const int nb = E.i - j0 < 16 ? E.i - j0 : 16;
								const uint64_t mlo = nb >= 8 ? ~0ull : (1ull << (8 * nb)) - 1ull, mhi = nb >= 16 ? ~0ull : (nb > 8 ? (1ull << (8 * (nb - 8))) - 1ull : 0ull);
								if (((lo & mlo) | (hi & mhi)) & 0xFCFCFCFCFCFCFCFCull) { ok = false; break; }      // an N never matches
								const uint32_t rd = deep_squeeze(lo) | deep_squeeze(hi) << 16;
								const uint32_t p0 = L(tpos) - (uint32_t)E.i + (uint32_t)j0, w0 = p0 >> 4;
								const uint32_t t0 = txt[w0], t1 = txt[w0 + 1u];
								const uint32_t tx = (uint32_t)(((uint64_t)t1 << 32 | t0) >> ((p0 & 15u) << 1));
								const uint32_t mask = nb >= 16 ? 0xffffffffu : (1u << (2 * nb)) - 1u;
								if ((rd ^ tx) & mask) ok = false;
							}
							if (PROF) L(ntx) += 1;
							if (ok) L(ts) = 3; else fail = true;
						} else if (L(ts) == 3) {
							E.k = E.l = (q1 ? S.bwt[1].isa : S.bwt[0].isa)[L(tpos) - (uint32_t)E.i];
							hit = true;
						} else if (L(ts) == 7) {
							// ---- forced levels of an entry on ONE row (DF_RUN): the text position of its suffix, ...
							const uint32_t pos = (q1 ? S.bwt[1].sa_full : S.bwt[0].sa_full)[E.k];
							if (pos == 0xffffffffu) { L(rel) += 1; L(act) = true; L(norun) = true; L(flag) = DF_NONE; L(ts) = -1; }      // (the empty suffix: this entry takes its levels by rank queries; its pop is undone)
							else { L(tpos) = pos; L(ts) = L(flag) == DF_COOP ? 10 : 8; }      // (10: waits for the wave, behind this loop)
						} else if (L(ts) == 8) {
							// ... the levels, up to 32 per turn: level after level what the chain step would do with an expansion that pushes its matching
							// child alone and with that child's pop -- the symbol in front of the suffix is the text's, ...
							const uint32_t *const txt = q1 ? S.bwt[1].text : S.bwt[0].text;
							const int m = max_diff - E.mm - E.go - (gape_mode ? E.ge : 0), m_seed = S.max_seed_diff - E.mm - E.go - (gape_mode ? E.ge : 0);
							const uint32_t tp = L(tpos), lo = tp >= 32u ? tp - 32u : 0u, w0 = lo >> 4, sh = (lo & 15u) << 1;
							const uint32_t t0 = txt[w0], t1 = txt[w0 + 1u], t2 = txt[w0 + 2u];
							const uint64_t a64 = (uint64_t)t1 << 32 | t0;
							const uint64_t win = sh ? (a64 >> sh | (uint64_t)t2 << (64u - sh)) : a64;      // text[lo + x] = (win >> 2 x) & 3, x < 32
							const uint32_t avail = tp - lo;
							int p = (int)E.i - 1; uint32_t d = 0; bool stop = false;
							for (;;) {
								if (d >= avail) { if (lo == 0u) fail = true; break; }      // the window is used up (or the text begins here: nothing in front of it)
								const uint32_t c = DEEP_RD(E.a, p), x = (uint32_t)(win >> ((tp - 1u - d - lo) << 1)) & 3u;
								if (c != x) { fail = true; break; }                       // (an N is 4: never equal)
								++d;
								if (p == 0) { stop = true; break; }
								if (m < (int)(DEEP_BB(E.a, p - 1) & 127u)) { fail = true; break; }
								if (m == 0 || !DEEP_FORCED(E.a, p - 1, m, m_seed)) { stop = true; break; }
								--p;
							}
							if (!fail) { L(tpos) = tp - d; E.i = E.i - d; if (stop) L(ts) = 9; }
						} else if (L(ts) == 9) {
							// ... and the row of the suffix the walk ended on: the chain goes on with it
							E.k = E.l = (q1 ? S.bwt[1].isa : S.bwt[0].isa)[L(tpos)];
							E.ldp = 0; E.state = DST_M;
							L(rel) += 1; L(ts) = -1;
							if (careful) L(flag) = DF_CONT; else { L(flag) = DF_NONE; L(act) = true; }
						} else {
							const uint2 *const tab = deep_table_of(s_bc, E.a);
							const uint32_t kt = E.l & 0xffu;
							uint32_t key = E.k, j = 0;
							if (PROF) L(pk_ktl) += 1u;
							if (L(ts) == 5) {
								j = KT - kt < (uint32_t)E.i ? KT - kt : (uint32_t)E.i;
								for (uint32_t u = 0; u < j; ++u) { const uint32_t c = DEEP_RD(E.a, E.i - 1 - (int)u); if (c > 3u) fail = true; key = key << 2 | (c & 3u); }
							}
							if (!fail) {
								const uint2 r = kt + j ? deep_ld_global8(tab + ((size_t)DEEP_LVO(kt + j) + key)) : make_uint2(0u, q1 ? S.bwt[1].seq_len : S.bwt[0].seq_len);
								E.k = r.x; E.l = r.y; E.i -= j;
								if (r.x > r.y) fail = true;
								else if (E.i == 0) hit = true;
								else L(ts) = (text_ok && r.x == r.y) ? 1 : 0;
							}
						}
						if (fail) { L(flag) = DF_NONE; L(ts) = -1; }
						else if (hit) { L(flag) = DF_HIT; L(ts) = -1; }
					} }
				}
				// ---------------------------------------------------------------- chains on ONE row, the wave together (DF_COOP).  The chain in front
				// of a search's first hit follows the read along one suffix of the text for a hundred levels and more, storing an insertion and a
				// deletion at every one of them: as chain steps that is a hundred wave-steps with one lane at work.  But everything a level does is
				// known from the text -- the one symbol in front of the suffix, hence the children that exist -- and from the read and its bounds;
				// what runs from level to level is only "the match went on" and sums.  So the wave takes such a chain 64 levels at a time, lane d
				// its level d: the entry there (row, position), what its pop finds (bwtgap.c:141-164), what its expansion pushes and stores; the
				// chain reaches level d when every level before it matched on (a ballot), the live-entry count, its peak, the classes' offsets
				// and the record slots are prefix sums over the levels reached, and the records leave together.  Then the chain's lane gets the
				// state the chain step would have left it in: ended, a hit, an exact tail, out of record room, or 64 levels further on.
				if (COOP) {
					uint64_t cm = WBALLOT(L(ts) == 10);
					if (cm) {
						const uint64_t sm2 = WBALLOT(L(flag) == DF_HIT || L(flag) == DF_CONT);
						int jstop = sm2 ? deep_ctz64(sm2) : 64;
						LANE(uint32_t, v_rk); LANE(uint32_t, v_el); LANE(uint32_t, v_rc); LANE(uint32_t, v_cl); LANE(uint32_t, v_inf); LANE(uint32_t, v_g); LANE(uint32_t, v_rec);
						LANE(uint32_t, v_a0); LANE(uint32_t, v_a1); LANE(uint32_t, v_a2); LANE(uint32_t, v_ng);
						LANE(uint32_t, x_g); LANE(uint32_t, x_f); LANE(uint32_t, x_0); LANE(uint32_t, x_1); LANE(uint32_t, x_2);
						while (cm) {
							const int j = deep_ctz64(cm);
							cm &= cm - 1ull;
							if (j > jstop) { LANES { if (ln == j) { L(flag) = DF_NONE; L(ts) = -1; } } continue; }      // a lane below it ended in a hit or is to be continued: dropped
							if (PROF) ++st_coop;
							// the chain: its entry (popped, past its pop's checks), its text position, its counts so far
							LANES { L(tu) = L(e).k; } const uint32_t ck = WUNI(WBCAST(tu, j));
							LANES { L(tu) = L(e).l; } const uint32_t cl0 = WUNI(WBCAST(tu, j));
							// the chain stands on one row (its levels come out of the text) or in key form (out of the interval table: level d's string is the
							// entry's and the read's next d symbols, its four possible extensions 32 bytes of the next table level -- down to the table's depth)
							const bool ckf = cl0 >= DEEP_KEYL;
							const uint32_t kt0 = cl0 & 0xffu;
							const uint32_t zlast = ckf ? (KT - 1u - kt0 < 63u ? KT - 1u - kt0 : 63u) : 63u;
							LANES { L(tu) = (uint32_t)L(e).i; } const int ci = (int)WUNI(WBCAST(tu, j));
							LANES { L(tu) = (uint32_t)L(e).mm | (uint32_t)L(e).go << 8 | (uint32_t)L(e).ge << 16 | (uint32_t)L(e).state << 24 | (uint32_t)L(e).a << 26; }
							const uint32_t cinfo = WUNI(WBCAST(tu, j));
							const uint32_t ctp = WUNI(WBCAST(tpos, j));
							const int crel = (int)WUNI(WBCAST(rel, j)), cpeak = (int)WUNI(WBCAST(peak, j));
							const uint32_t c0 = WUNI(WBCAST(cc0, j)), c1 = WUNI(WBCAST(cc1, j)), c2 = WUNI(WBCAST(cc2, j)), cnr = WUNI(WBCAST(nrec, j));
							const int cmm = (int)(cinfo & 0xffu), cgo = (int)(cinfo >> 8 & 0xffu), cge = (int)(cinfo >> 16 & 0xffu), cst = (int)(cinfo >> 24 & 3u);
							const uint32_t ca = cinfo >> 26 & 1u;
							const bool cq1 = ca == 0u;
							const uint2 *const ctab = deep_table_of(s_bc, ca);
							const uint32_t *const isa = cq1 ? S.bwt[1].isa : S.bwt[0].isa, *const txt = cq1 ? S.bwt[1].text : S.bwt[0].text;
							const int m = max_diff - cmm - cgo - (gape_mode ? cge : 0), m_seed = S.max_seed_diff - cmm - cgo - (gape_mode ? cge : 0);
							int tmp = cgo + cge;
							if (loggap) { const uint32_t v = (uint32_t)(cge + cgo); tmp = (v ? 31 - __clz((int)v) : 0) / 2 + 1; }
							// ---- lane d: level d.  kind: 0 expanded and matched on, 1 expanded and the chain ends, 2 dropped at the pop, 3 a hit, 4 an exact tail, 5 no such level
							LANES {
								const uint32_t d = (uint32_t)ln;
								uint32_t kind = 5u, rk = 0u, el = 0u, rc = 0u, cl = 0u, g = 0u, a0 = 0u, a1 = 0u, a2 = 0u, ng = 0u, rcd = 0u, inf = 0u;
								if ((int)d <= ci && (ckf ? d <= zlast : d <= ctp)) {
									const int Ei = ci - (int)d;
									const uint32_t tq = ckf ? 0u : ctp - d;
									if (ckf) {
										uint32_t key = ck;
										for (uint32_t u = 0; u < d; ++u) key = key << 2 | (DEEP_RD(ca, ci - 1 - (int)u) & 3u);
										rk = key; el = DEEP_KEYL | (kt0 + d);
									} else { rk = d == 0u ? ck : isa[tq]; el = rk; }
									const int st = d == 0u ? cst : DST_M;
									bool go_on = true, hit = false, tail = false;
									if (d > 0u) {
										go_on = m >= 0 && !(Ei > 0 && m < (int)(DEEP_BB(ca, Ei - 1) & 127u));
										hit = go_on && Ei == 0;
										tail = go_on && !hit && m == 0;
									}
									if (!go_on) kind = 2u; else if (hit) kind = 3u; else if (tail) kind = 4u;
									else {
										const int p = Ei - 1;
										const uint32_t c = DEEP_RD(ca, p);
										uint32_t vm, occ = 1u;
										if (ckf) {
											const uint4 *const ch = (const uint4*)(ctab + ((size_t)DEEP_LVO(kt0 + d + 1u) + (size_t)rk * 4u));
											const uint4 c01 = deep_ld_global16(ch), c23 = deep_ld_global16(ch + 1);
											vm = (c01.x <= c01.y ? 1u : 0u) | (c01.z <= c01.w ? 2u : 0u) | (c23.x <= c23.y ? 4u : 0u) | (c23.z <= c23.w ? 8u : 0u);
											occ = 0xffffffffu;
											if (st == DST_D && cge + cgo >= max_diff && kt0 + d) { const uint2 own = deep_ld_global8(ctab + ((size_t)DEEP_LVO(kt0 + d) + rk)); occ = own.y - own.x + 1u; }
											if (kt0 + d + 1u < KT) { rc = rk << 2 | (c & 3u); cl = DEEP_KEYL | (kt0 + d + 1u); }
											else { rc = c == 0u ? c01.x : (c == 1u ? c01.z : (c == 2u ? c23.x : c23.z)); cl = c == 0u ? c01.y : (c == 1u ? c01.w : (c == 2u ? c23.y : c23.w)); }
										} else {
											const bool have = tq >= 1u;
											uint32_t x = 0u;
											if (have) { x = txt[(tq - 1u) >> 4] >> (((tq - 1u) & 15u) << 1) & 3u; rc = isa[tq - 1u]; }
											cl = rc;
											vm = have ? 1u << x : 0u;
										}
										const uint32_t nv = (uint32_t)__popc(vm);
										bool allow_diff, allow_M;
										DEEP_BOUNDS(ca, p, m, m_seed, allow_diff, allow_M);
										uint32_t grp = 0u, n_gap = 0u, gcls = DCL_GO;
										if (allow_diff && p >= S.indel_end_skip + tmp && len - p >= S.indel_end_skip + tmp) {
											if (st == DST_M) { if (cgo < MG) { grp = DRG_OPEN; n_gap = 1u + nv; } }
											else if (st == DST_I) { if (cge < S.max_gape) { grp = DRG_EXT_I; n_gap = 1u; gcls = DCL_GE; } }
											else if (cge < S.max_gape && (cge + cgo < max_diff || occ < (uint32_t)S.max_del_occ)) { grp = DRG_EXT_D; n_gap = nv; gcls = DCL_GE; }
										}
										uint32_t mmv = 0u; bool match = false;
										if (allow_diff && allow_M) { mmv = c > 3u ? vm : vm & ~(1u << c); match = c <= 3u && (vm >> c & 1u); }
										else if (c < 4u) match = (vm >> c & 1u) != 0u;
										const uint32_t n_mm = (uint32_t)__popc(mmv);
										g = n_gap + n_mm;
										const bool keep_gap = gcls == DCL_GO ? keep1 : keep2;
										if (!keep_gap) { grp = 0u; n_gap = 0u; }
										if (!keep0) mmv = 0u;
										const uint32_t n_mm_st = keep0 ? n_mm : 0u;
										const uint32_t gcn = gcls == DCL_GO ? can1 : can2;
										ng = n_gap;
										a0 = (gcn == 0u ? n_gap : 0u) + n_mm_st; a1 = gcn == 1u ? n_gap : 0u; a2 = gcn == 2u ? n_gap : 0u;
										rcd = n_gap + n_mm_st ? 1u : 0u;
										inf = vm | grp << 4 | mmv << 8 | gcn << 12 | (c & 7u) << 14;
										kind = match ? 0u : 1u;
									}
								}
								L(v_rk) = rk; L(v_el) = el; L(v_rc) = rc; L(v_cl) = cl; L(v_inf) = inf | kind << 17; L(v_g) = g; L(v_rec) = rcd; L(v_a0) = a0; L(v_a1) = a1; L(v_a2) = a2; L(v_ng) = ng;
							}
							// ---- how far the chain gets: every level before it matched on; no further than its record room (or, careful, one pop)
							const uint64_t contm = WBALLOT((L(v_inf) >> 17 & 7u) == 0u);
							uint32_t z = ~contm ? (uint32_t)deep_ctz64(~contm) : 63u;
							if (z > zlast) z = zlast;                          // (the table ends there: a chain that got through goes on as rows, in chain steps)
							LANES { if ((uint32_t)ln > z) { L(v_g) = 0u; L(v_rec) = 0u; L(v_a0) = L(v_a1) = L(v_a2) = 0u; } }
							uint32_t tot_ = 0;
							WEXSCAN_U32(L(x_g), L(v_g), tot_); WEXSCAN_U32(L(x_f), L(v_rec), tot_);
							WEXSCAN_U32(L(x_0), L(v_a0), tot_); WEXSCAN_U32(L(x_1), L(v_a1), tot_); WEXSCAN_U32(L(x_2), L(v_a2), tot_);
							(void)tot_;
							const uint64_t fullm = WBALLOT((uint32_t)ln <= z && (L(v_inf) >> 17 & 7u) == 0u && (careful || cnr + L(x_f) + L(v_rec) >= K));
							const uint32_t le = fullm ? (uint32_t)deep_ctz64(fullm) : z;
							const bool cut = fullm != 0ull;
							if (PROF) st_cooplev += le + 1u;
							++coop_n; coop_lv += le + 1u;
							if (coop_n >= 0x40000000u) { coop_n >>= 1; coop_lv >>= 1; }
							int64_t mxr = (int64_t)cpeak;
							{ int64_t mx_ = 0; WMAX_I64(mx_, ((uint32_t)ln >= 1u && (uint32_t)ln <= le) ? (int64_t)crel + 1 + (int64_t)L(x_g) : (int64_t)cpeak); if (mx_ > mxr) mxr = mx_; }
							// ---- the records of the levels reached
							LANES {
								if ((uint32_t)ln <= le && L(v_rec)) {
									const uint32_t inf = L(v_inf), vm = inf & 15u, grp = inf >> 4 & 3u, mmv = inf >> 8 & 15u, gcn = inf >> 12 & 3u, c = inf >> 14 & 7u;
									const uint32_t rc = L(v_rc), rk = L(v_rk), el = L(v_el);
									const uint32_t at_gap = gcn == 0u ? c0 + L(x_0) : (gcn == 1u ? c1 + L(x_1) : c2 + L(x_2));
									const uint32_t at_mm = c0 + L(x_0) + (gcn == 0u ? L(v_ng) : 0u);
									const uint32_t p = (uint32_t)(ci - ln - 1);
									uint4 *const rp = stage + ((size_t)j * K + cnr + L(x_f)) * 4u;
									if (ckf) {                                          // the four intervals once more (they were not kept over the sums: eight registers a lane)
										const uint4 *const ch = (const uint4*)(ctab + ((size_t)DEEP_LVO((el & 0xffu) + 1u) + (size_t)rk * 4u));
										const uint4 c01 = deep_ld_global16(ch), c23 = deep_ld_global16(ch + 1);
										rp[0] = make_uint4(c01.x, c01.z, c23.x, c23.z);
										rp[1] = make_uint4(c01.y, c01.w, c23.y, c23.w);
									} else {
										rp[0] = make_uint4(vm & 1u ? rc : 1u, vm & 2u ? rc : 1u, vm & 4u ? rc : 1u, vm & 8u ? rc : 1u);
										rp[1] = make_uint4(vm & 1u ? rc : 0u, vm & 2u ? rc : 0u, vm & 4u ? rc : 0u, vm & 8u ? rc : 0u);
									}
									rp[2] = make_uint4(rk, el, p | vm << 16 | grp << 20 | mmv << 24 | gcn << 28,
													   (uint32_t)cmm | (uint32_t)cgo << 8 | (uint32_t)cge << 16 | ca << 26 | c << 27);
									rp[3] = make_uint4(at_gap, at_mm, 0u, 0u);
								}
							}
							// ---- the chain's lane: where the chain step would have left it
							const uint32_t e_kind = WUNI(WBCAST(v_inf, le)) >> 17 & 7u, e_rk = WUNI(WBCAST(v_rk, le)), e_rc = WUNI(WBCAST(v_rc, le)), e_el = WUNI(WBCAST(v_el, le)), e_cl = WUNI(WBCAST(v_cl, le));
							const uint32_t e_g = WUNI(WBCAST(v_g, le)), e_xg = WUNI(WBCAST(x_g, le)), e_nr = cnr + WUNI(WBCAST(x_f, le)) + WUNI(WBCAST(v_rec, le));
							const uint32_t n0 = c0 + WUNI(WBCAST(x_0, le)) + WUNI(WBCAST(v_a0, le)), n1 = c1 + WUNI(WBCAST(x_1, le)) + WUNI(WBCAST(v_a1, le)), n2 = c2 + WUNI(WBCAST(x_2, le)) + WUNI(WBCAST(v_a2, le));
							const int nrel = crel + (int)e_xg + (e_kind <= 1u ? (int)e_g : 0) + (e_kind == 0u ? 1 : 0);
							LANES { if (ln == j) {
								DeepLane &E = L(e);
								L(rel) = nrel; L(peak) = (int)mxr; L(cc0) = n0; L(cc1) = n1; L(cc2) = n2; L(nrec) = e_nr; L(ts) = -1; L(flag) = DF_NONE;
								if (e_kind == 0u) {                                       // its matching child is the next pop: out of room / careful (to be continued), or on it goes
									E.k = e_rc; E.l = e_cl; E.i = (uint32_t)(ci - (int)le - 1); E.ldp = 0; E.state = DST_M;
									if (cut) L(flag) = DF_CONT; else L(act) = true;
								} else if (e_kind == 3u || e_kind == 4u) {
									E.k = e_rk; E.l = e_el; E.i = (uint32_t)(ci - (int)le); E.ldp = 0; E.state = DST_M;
									L(flag) = e_kind == 3u ? DF_HIT : DF_TAIL;
								}
							} }
							if ((e_kind == 0u && cut) || e_kind == 3u) { if (j < jstop) jstop = j; }
							WAVE_SYNC();
						}
					}
				}
				if (WBALLOT(L(act) || L(flag) == DF_TAIL || (L(flag) == DF_HIT && L(e).l >= DEEP_KEYL)) == 0ull) break;      // (a hit found here in key form still takes its rows from the table: the tails' loop once more)
				if (prof) { pc1 = DEEP_CLOCK(); ph_tail += pc1 - pc2; }
				}

				unsigned long long pc3 = 0; if (prof) { pc3 = DEEP_CLOCK(); ph_tail += pc3 - pc2; }
				// ---------------------------------------------------------------- commit, in the reference's order
				const uint64_t stopm = WBALLOT(L(flag) != DF_NONE && (uint32_t)ln < W);
				int jl = stopm ? deep_ctz64(stopm) : (int)W - 1;            // the last lane whose chain counts
				{ uint32_t tot = 0; LANES { L(d) = ln <= jl ? (uint32_t)L(rel) : 0u; } WEXSCAN_U32(L(off), L(d), tot); (void)tot; }
				LANES { L(nst) = n_entries + (int)L(off); }                  // live entries before this lane's first pop
				const uint64_t over = WBALLOT(ln <= jl && (int64_t)L(nst) + L(peak) > (int64_t)S.max_entries);
				if (over) { careful = true; jl = deep_ctz64(over) - 1; }   // the cut-off (bwtgap.c:140) falls into that lane's chain
				if (jl < 0) { if (prof) ph_commit += DEEP_CLOCK() - pc3; continue; }
				{
					int64_t mx = 0;
					WMAX_I64(mx, ln <= jl ? (int64_t)L(nst) + L(peak) : (int64_t)0);
					if (mx > max_ent) max_ent = (int)mx;
				}
				LANES { L(tu) = (uint32_t)(L(nst) + L(rel)); }
				n_entries = (int)WUNI(WBCAST(tu, jl));
				if (PROF) st_commit += (unsigned)(jl + 1);
				if (counting) { uint32_t tt = 0; LANES { L(d) = ln <= jl ? L(tch) : 0u; } WEXSCAN_U32(L(off), L(d), tt); rd_touch += tt; }
				// the popped entries leave level s
				const uint32_t newc = cs - (uint32_t)(jl + 1);
				ONE_LANE { s_cnt[s] = newc; }
				if (newc == 0u || ((newc - 1u) >> DEEP_PAGE_SH) != topq) {      // its top page is empty now
					if (newc && prev_pg == DEEP_NIL) prev_pg = WUNI(P.page_prev[top_pg]);
					ONE_LANE { freep[n_free] = top_pg; s_top[s] = newc ? prev_pg : DEEP_NIL; }
					++n_free;
				}
				WAVE_SYNC();
				bool pool_fail = false;
				// the staged children go to their levels: lane by lane, within a lane in chain order.  Per (canonical) class: the
				// lanes' slots by prefix sum, the pages those slots need; then ONE pass over each lane's staging buffer
				uint32_t tot0 = 0, tot1 = 0, tot2 = 0;
				LANE(uint32_t, o0); LANE(uint32_t, o1); LANE(uint32_t, o2);
				LANES { L(d) = ln <= jl ? L(cc0) : 0u; } WEXSCAN_U32(L(o0), L(d), tot0);
				LANES { L(d) = ln <= jl ? L(cc1) : 0u; } WEXSCAN_U32(L(o1), L(d), tot1);
				LANES { L(d) = ln <= jl ? L(cc2) : 0u; } WEXSCAN_U32(L(o2), L(d), tot2);
				if ((tot0 && (uint32_t)T0 >= P.NS) || (tot1 && (uint32_t)T1 >= P.NS) || (tot2 && (uint32_t)T2 >= P.NS)) pool_fail = true;   // (cannot happen: the host sizes NS by the largest score an entry can have)
				uint32_t cT0 = 0, cT1 = 0, cT2 = 0, ot0 = DEEP_NIL, ot1 = DEEP_NIL, ot2 = DEEP_NIL, qn0 = 0, qn1 = 0, qn2 = 0, nn0 = 0, nn1 = 0, nn2 = 0;
				if (!pool_fail) {
#define DEEP_LEVEL(tot_, T_, cT_, ot_, qn_, nn_) if (tot_) { cT_ = WUNI(s_cnt[T_]); ot_ = WUNI(s_top[T_]); qn_ = cT_ ? ((cT_ - 1u) >> DEEP_PAGE_SH) + 1u : 0u; \
						const uint32_t ql_ = (cT_ + tot_ - 1u) >> DEEP_PAGE_SH; nn_ = ql_ + 1u > qn_ ? ql_ + 1u - qn_ : 0u; }
					DEEP_LEVEL(tot0, T0, cT0, ot0, qn0, nn0)
					DEEP_LEVEL(tot1, T1, cT1, ot1, qn1, nn1)
					DEEP_LEVEL(tot2, T2, cT2, ot2, qn2, nn2)
#undef DEEP_LEVEL
					if (nn0 + nn1 + nn2) { bool okp = false; DEEP_ALLOC(nn0 + nn1 + nn2, okp); if (!okp) pool_fail = true; }
				}
				if (!pool_fail) {
					const uint32_t nb1 = nn0, nb2 = nn0 + nn1;        // where each class's new pages start in s_newp[]
					ONE_LANE {
#define DEEP_LINK(tot_, T_, cT_, ot_, nn_, nb_) if (tot_) { for (uint32_t t = 0; t < nn_; ++t) P.page_prev[s_newp[nb_ + t]] = t ? s_newp[nb_ + t - 1u] : ot_; \
						s_cnt[T_] = cT_ + tot_; if (nn_) s_top[T_] = s_newp[nb_ + nn_ - 1u]; }
						DEEP_LINK(tot0, T0, cT0, ot0, nn0, 0u)
						DEEP_LINK(tot1, T1, cT1, ot1, nn1, nb1)
						DEEP_LINK(tot2, T2, cT2, ot2, nn2, nb2)
#undef DEEP_LINK
					}
					// the records become entries, the work spread evenly over the wave whatever the lanes' chains filed: record g of the
					// round belongs to the lane j with off_j <= g < off_j + nrec_j (a search in the prefix sums, which sit in LDS for it
					// together with the lanes' offsets in the three classes); a record's children go to consecutive places of their
					// class, in the order the reference pushes them
					uint32_t totr = 0;
					LANE(uint32_t, orc);
					LANES { L(d) = ln <= jl ? L(nrec) : 0u; } WEXSCAN_U32(L(orc), L(d), totr);
					LANES { s_off[ln] = ln <= jl ? L(orc) : totr; s_off[64 + ln] = L(o0); s_off[128 + ln] = L(o1); s_off[192 + ln] = L(o2); }
					WAVE_SYNC();
#define DEEP_PUT(cT_, ot_, qn_, nb_, g_, v_) do { const uint32_t w_ = (cT_) + (g_), q_ = w_ >> DEEP_PAGE_SH; \
						const uint32_t pg_ = q_ < (qn_) ? (ot_) : s_newp[(nb_) + q_ - (qn_)]; \
						P.pages[(size_t)pg_ * DEEP_PAGE + (w_ & (DEEP_PAGE - 1u))] = (v_); } while (0)
					LANES { for (uint32_t g = (uint32_t)ln; g < totr; g += 64u) {
						uint32_t j = 0;
						for (uint32_t stp = 32u; stp; stp >>= 1) if (s_off[j + stp] <= g) j += stp;
						const uint4 *const rp = stage + ((size_t)j * K + (g - s_off[j])) * 4u;
						const uint4 r0 = rp[0], r1 = rp[1], r2 = rp[2], r3 = rp[3];
						const uint32_t nkv[4] = { r0.x, r0.y, r0.z, r0.w }, nlv[4] = { r1.x, r1.y, r1.z, r1.w };
						const uint32_t ri = r2.z & 0xffffu, vm = r2.z >> 16 & 15u, grp = r2.z >> 20 & 15u, mmv = r2.z >> 24 & 15u, gcn = r2.z >> 28 & 3u;
						const int pmm = (int)(r2.w & 0xffu), pgo = (int)(r2.w >> 8 & 0xffu), pge = (int)(r2.w >> 16 & 0xffu), pa = (int)(r2.w >> 26 & 1u);
						const uint32_t c = r2.w >> 27 & 7u;
						// the children of a parent in key form stay in it while the table has a level below theirs (a deletion or a mismatch over x: the
						// parent's string and x); the insertion child has its parent's string
						const uint32_t kt1 = (r2.y & 0xffu) + 1u;
						const bool kfc = r2.y >= DEEP_KEYL && kt1 < KT;
						const uint32_t ckey = r2.x << 2, clen = DEEP_KEYL | kt1;
						if (grp) {                                    // the gap group: [the insertion,] [the deletions over the symbols that occur]
							const uint32_t gT = gcn == 0u ? cT0 : (gcn == 1u ? cT1 : cT2), gO = gcn == 0u ? ot0 : (gcn == 1u ? ot1 : ot2);
							const uint32_t gQ = gcn == 0u ? qn0 : (gcn == 1u ? qn1 : qn2), gN = gcn == 0u ? 0u : (gcn == 1u ? nb1 : nb2);
							uint32_t at = s_off[64u + 64u * gcn + j] + r3.x;
							const int cgo = pgo + (grp == DRG_OPEN ? 1 : 0), cge = pge + (grp == DRG_OPEN ? 0 : 1);
							if (grp != DRG_EXT_D) { DEEP_PUT(gT, gO, gQ, gN, at, deep_pack(r2.x, r2.y, (int)ri, (int)ri, pmm, cgo, cge, DST_I, pa, 0u)); ++at; }
							if (grp != DRG_EXT_I) {
#pragma unroll
								for (int x = 0; x < 4; ++x)
									if (vm >> x & 1u) { DEEP_PUT(gT, gO, gQ, gN, at, deep_pack(kfc ? ckey | (uint32_t)x : nkv[x], kfc ? clen : nlv[x], (int)ri + 1, (int)ri + 1, pmm, cgo, cge, DST_D, pa, 0u)); ++at; }
							}
						}
						if (mmv) {                                     // the mismatches, in the reference's order: (c + 1) & 3, (c + 2) & 3, ...
							uint32_t at = s_off[64u + j] + r3.y;
#pragma unroll
							for (uint32_t t = 1; t <= 4u; ++t) {
								const uint32_t x = (c + t) & 3u;
								if (mmv >> x & 1u) { DEEP_PUT(cT0, ot0, qn0, 0u, at, deep_pack(kfc ? ckey | x : deep_sel4(nkv, x), kfc ? clen : deep_sel4(nlv, x), (int)ri, (int)ri, pmm + 1, pgo, pge, DST_M, pa, 0u)); ++at; }
							}
						}
					} }
					WAVE_SYNC();
#undef DEEP_PUT
				}
				const int fl = WUNI(WBCAST(flag, jl));
				if (!pool_fail && !over && fl == DF_CONT) {
					// the chain of lane jl goes on in the next round: its current entry is the newest of level s again
					const uint32_t cT = WUNI(s_cnt[s]), old_top = WUNI(s_top[s]);
					const bool need = (cT & (DEEP_PAGE - 1u)) == 0u;
					bool okp = true;
					if (need) DEEP_ALLOC(1u, okp);
					if (!okp) pool_fail = true;
					else {
						const uint32_t pg = need ? WUNI(s_newp[0]) : old_top;
						ONE_LANE { if (need) { P.page_prev[pg] = old_top; s_top[s] = pg; } s_cnt[s] = cT + 1u; }
						LANES { if (ln == jl) P.pages[(size_t)pg * DEEP_PAGE + (cT & (DEEP_PAGE - 1u))] =
							deep_pack(L(e).k, L(e).l, L(e).i, L(e).ldp, L(e).mm, L(e).go, L(e).ge, L(e).state, L(e).a, 0u); }
						WAVE_SYNC();
					}
				}
				if (pool_fail) { status = NABWA_ST_POOL; if (PROF) ++st_pool; break; }
				unsigned long long pc4 = 0; if (prof) { pc4 = DEEP_CLOCK(); ph_commit += pc4 - pc3; }
				if (!over && fl == DF_HIT) {
					// ---- hit bookkeeping (bwtgap.c:166-199), the wave together
					LANES { L(tu) = L(e).k; } const uint32_t hk = WUNI(WBCAST(tu, jl));
					LANES { L(tu) = L(e).l; } const uint32_t hl = WUNI(WBCAST(tu, jl));
					LANES { L(tu) = (uint32_t)L(e).mm | (uint32_t)L(e).go << 8 | (uint32_t)L(e).ge << 16 | (uint32_t)L(e).a << 24; } const uint32_t hinfo = WUNI(WBCAST(tu, jl));
					LANES { L(tu) = (uint32_t)L(e).ldp; } const int h_ldp = (int)WUNI(WBCAST(tu, jl));
					const int h_mm = (int)(hinfo & 0xffu), h_go = (int)(hinfo >> 8 & 0xffu), h_ge = (int)(hinfo >> 16 & 0xffu), h_a = (int)(hinfo >> 24 & 1u);
					const int score = s;
					bool do_add = true;
					if (n_aln == 0) {
						best_score = score;
						const int best_diff = h_mm + h_go + (gape_mode ? h_ge : 0);
						if (!nonstop) max_diff = best_diff + 1 > MD ? MD : best_diff + 1;
					}
					if (score == best_score) best_cnt += (int)(hl - hk + 1u);
					else if (best_cnt > S.max_top2) { done = true; do_add = false; }
					if (do_add && h_go) {       // a gap in a tandem repeat finds the same interval again (bwtgap.c:179-183)
						LANES { L(tu) = 0u; for (int j = ln; j < n_aln; j += 64) { const uint4 h = out[j]; if (h.y == hk && h.z == hl) L(tu) = 1u; } }
						if (WBALLOT(L(tu) != 0u)) do_add = false;
					}
					if (do_add) {
						if (n_aln == S.aln_cap) { status = NABWA_ST_HITCAP; done = true; }
						else {
							// gap_shadow (bwtgap.c:81-91) on this strand's widths, positions < last_diff_pos, 64 at a time; then the
							// "w[t-1] == w[t]" bits of the bound bytes of positions 1 .. last_diff_pos are refreshed
							const uint32_t x = hl - hk + 1u, mx = h_a ? S.bwt[0].seq_len : S.bwt[1].seq_len;
							uint32_t *const wp = (uint32_t*)rec + (uint32_t)h_a * S.WL;
							uint8_t *const bp = lds_mode ? s_bb + (uint32_t)h_a * S.WLB : rec + S.woff_bid + (uint32_t)h_a * S.WLB;
							uint32_t jj = 0;
							for (int t0 = 0; t0 < h_ldp; t0 += 64) {
								const uint64_t eqm = WBALLOT(t0 + ln < h_ldp && wp[t0 + ln] == x);
								LANES {
									const int t = t0 + ln;
									if (t < h_ldp) {
										const uint32_t wv = wp[t];
										if (wv > x) wp[t] = wv - x;
										else if (wv == x) {
											const uint32_t rank = (uint32_t)__popcll((unsigned long long)(eqm & ((2ull << ln) - 1ull)));
											wp[t] = mx - (jj + rank); bp[t] = (uint8_t)((bp[t] & 128u) | 1u);
										}
									}
								}
								jj += (uint32_t)__popcll((unsigned long long)eqm);
							}
							WAVE_SYNC();
							for (int t0 = 1; t0 <= h_ldp && h_ldp > 0; t0 += 64) {
								LANES {
									const int t = t0 + ln;
									if (t <= h_ldp) bp[t] = (uint8_t)((bp[t] & 127u) | (wp[t] == wp[t - 1] ? 128u : 0u));
								}
							}
							ONE_LANE { out[n_aln] = make_uint4(hinfo, hk, hl, (uint32_t)score); }
							++n_aln;
							WAVE_SYNC();
						}
					}
					if (prof) ph_hit += DEEP_CLOCK() - pc4;
				}
			}
		}
		ONE_LANE { S.n_aln[item] = n_aln; S.max_ent[item] = max_ent; S.status[item] = (uint8_t)status; }
		if (PROF && P.rounds_out) { ONE_LANE { P.rounds_out[idx] = (uint32_t)(st_rounds - rounds0); } }
		if (prof) { const unsigned long long dt = DEEP_CLOCK() - clk0; st_sumclk += dt; if (dt > st_maxclk) st_maxclk = dt; if (st_rounds - rounds0 > st_maxrounds) st_maxrounds = st_rounds - rounds0; }
		if (counting && status == NABWA_ST_OK) { ONE_LANE { DEEP_ATOMIC_ADD_U64(S.touch_counter, rd_touch); } }
	}
	if (PROF && P.stats) {
		uint32_t t6 = 0, t7 = 0;
		LANES { L(d) = L(ntl); } WEXSCAN_U32(L(off), L(d), t6);
		LANES { L(d) = L(ntx); } WEXSCAN_U32(L(off), L(d), t7);
		uint32_t s_key = 0, s_ktl = 0, s_rec = 0, s_chl = 0, s_prn = 0, s_exp = 0, s_two = 0, s_fk = 0, s_f1 = 0, s_fw = 0;
		LANES { L(d) = L(pk_key); } WEXSCAN_U32(L(off), L(d), s_key);
		LANES { L(d) = L(pk_ktl); } WEXSCAN_U32(L(off), L(d), s_ktl);
		LANES { L(d) = L(pk_rec); } WEXSCAN_U32(L(off), L(d), s_rec);
		LANES { L(d) = L(pk_chl); } WEXSCAN_U32(L(off), L(d), s_chl);
		LANES { L(d) = L(pk_prn); } WEXSCAN_U32(L(off), L(d), s_prn);
		LANES { L(d) = L(pk_exp); } WEXSCAN_U32(L(off), L(d), s_exp);
		LANES { L(d) = L(pk_two); } WEXSCAN_U32(L(off), L(d), s_two);
		LANES { L(d) = L(pk_fk); } WEXSCAN_U32(L(off), L(d), s_fk);
		LANES { L(d) = L(pk_f1); } WEXSCAN_U32(L(off), L(d), s_f1);
		LANES { L(d) = L(pk_fw); } WEXSCAN_U32(L(off), L(d), s_fw);
		ONE_LANE {
#ifdef NABWA_EMU
			P.stats[10] += s_key; P.stats[11] += s_ktl; P.stats[12] += s_fk; P.stats[13] += s_f1; P.stats[14] += st_coop; P.stats[15] += st_cooplev;
			P.stats[0] += st_rounds; P.stats[1] += st_run; P.stats[2] += st_commit; P.stats[3] += st_steps; P.stats[4] += st_careful; P.stats[5] += st_pool; P.stats[6] += t6; P.stats[7] += t7;
#else
			atomicAdd(P.stats + 0, st_rounds); atomicAdd(P.stats + 1, st_run); atomicAdd(P.stats + 2, st_commit);
			atomicAdd(P.stats + 3, st_steps); atomicAdd(P.stats + 4, st_careful); atomicAdd(P.stats + 5, st_pool);
			atomicAdd(P.stats + 6, (unsigned long long)t6); atomicAdd(P.stats + 7, (unsigned long long)t7);
			atomicAdd(P.stats + 16, ph_pop); atomicAdd(P.stats + 17, ph_chain); atomicAdd(P.stats + 18, ph_tail); atomicAdd(P.stats + 19, ph_commit); atomicAdd(P.stats + 20, ph_hit); atomicAdd(P.stats + 21, st_tailit); atomicAdd(P.stats + 22, st_lanesteps); atomicAdd(P.stats + 23, (unsigned long long)s_key); atomicAdd(P.stats + 24, (unsigned long long)s_ktl);
			atomicAdd(P.stats + 25, (unsigned long long)s_rec); atomicAdd(P.stats + 26, (unsigned long long)s_chl); atomicAdd(P.stats + 27, (unsigned long long)s_prn); atomicAdd(P.stats + 28, (unsigned long long)s_exp); atomicAdd(P.stats + 29, (unsigned long long)s_two);
			atomicAdd(P.stats + 30, (unsigned long long)s_fk); atomicAdd(P.stats + 31, (unsigned long long)s_f1); atomicAdd(P.stats + 9, (unsigned long long)s_fw);
			atomicMax(P.stats + 10, st_maxclk); atomicMax(P.stats + 11, st_maxrounds); atomicAdd(P.stats + 12, st_sumclk); atomicMax(P.stats + 13, DEEP_CLOCK() - clk_start);
#endif
		}
	}
}
