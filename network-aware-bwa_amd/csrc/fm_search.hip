// fm_search.hip -- bounded-backtracking FM-index search, one read per lane (gfx950).
//
// Computes what bwa_cal_sa_reg_gap does per read (bwtaln.c:93-142): the four bwt_cal_width
// passes (bwtaln.c:52-76) and bwt_match_gap (bwtgap.c:104-266), reproducing the exact
// pop/push order of the reference's per-score LIFO stacks (bwtgap.c:46-79) because that order
// decides which hits are found and in which order they are reported.
//
// Execution model: a persistent grid; every lane runs its own state machine and draws the
// next read from a global work counter when it finishes one (wave ballot + one atomic per
// wave).  Every trip of the wave loop performs at most ONE "rank step" per lane -- Occ of all
// four bases at rows (k-1, l) of one index, one or two 64-byte bucket fetches -- whatever the
// lane is doing (width pass, exact tail match, node expansion), so lanes in different phases
// stay converged on the expensive part: the bucket gathers.
//
// Priority stack: entries live in a per-lane arena in HBM (16 B each); entries of one score
// form a linked list through link[]; head[score] is the newest entry, a 128-bit mask in
// registers tracks the non-empty scores, so "pop the newest entry of the lowest score"
// is one ctz + two dependent loads.  First pass: bump allocation (arena = total pushes);
// reads that outgrow it are flagged and re-run from scratch by the same kernel instantiated
// with slot reuse and an arena of max_entries+16 live entries (never on the CPU).
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

template <typename LinkT, bool REUSE, bool COUNT>
__global__ __launch_bounds__(NABWA_SEARCH_BLOCK) void fm_search_kernel(const SearchParams P)
{
	const LinkT NIL = (LinkT)~(LinkT)0;
	const uint32_t lane = threadIdx.x & 63u;
	const size_t slot = (size_t)blockIdx.x * NABWA_SEARCH_BLOCK + threadIdx.x;
	uint8_t *const sc = P.scratch + slot * P.lane_stride;
	uint4 *const ent = (uint4*)sc;
	LinkT *const lnk = (LinkT*)(sc + P.off_link);
	LinkT *const freel = (LinkT*)(sc + P.off_free);
	LinkT *const head = (LinkT*)(sc + P.off_head);
	uint32_t *const Wd = (uint32_t*)(sc + P.off_w);      // [2][WL] interval widths
	uint32_t *const SWd = (uint32_t*)(sc + P.off_sw);    // [2][SL] seed widths
	uint8_t *const Bd = sc + P.off_bid;                  // [2][WL] lower bounds
	uint8_t *const SBd = sc + P.off_sbid;                // [2][SL]
	const bool gape_mode = P.mode & 0x01, nonstop = P.mode & 0x10, loggap = P.mode & 0x04;

	int st = ST_IDLE;
	// per-read
	uint32_t item = 0; int len = 0, md_read = 0, mg_read = 0; const uint8_t *sq0 = 0, *sq1 = 0;
	// width passes
	int pass = 0, wi = 0, wbid = 0, nN = 0;
	// current interval / query
	uint32_t k = 0, l = 0;
	// search globals
	int max_diff = 0, best_score = 0, best_cnt = 0, n_aln = 0, max_ent = 0, n_entries = 0;
	uint32_t bump = 0, nfree = 0; uint64_t mask_lo = 0, mask_hi = 0; bool seeded = false; int status = 0;
	// current entry
	int e_i = 0, e_a = 0, e_mm = 0, e_go = 0, e_ge = 0, e_state = 0, e_ldp = 0, m = 0, m_seed = 0, xt = 0;
	unsigned long long touches = 0; uint32_t rd_touch = 0;   // COUNT only

	for (;;) {
		// ---------------------------------------------------------------- refill
		const unsigned long long need = __ballot(st == ST_IDLE);
		if (need) {
			unsigned int base = 0;
			if (lane == 0) base = atomicAdd(P.work_counter, (unsigned int)__popcll(need));
			base = __shfl(base, 0);
			if (st == ST_IDLE) {
				const unsigned int idx = base + (unsigned int)__popcll(need & ((1ull << lane) - 1ull));
				if (idx < (unsigned int)P.n) {
					item = idx;
					const uint32_t rid = P.ids ? (uint32_t)P.ids[idx] : idx;
					const int64_t o = P.off[rid];
					len = (int)(P.off[rid + 1] - o);
					sq0 = P.seq + o; sq1 = P.rseq + o;
					md_read = P.rd_maxdiff[rid]; mg_read = P.rd_maxgapo[rid];
					n_aln = 0; max_ent = 0; status = NABWA_ST_OK; nN = 0; rd_touch = 0;
					if (len > 0) { pass = 0; wi = 0; wbid = 0; k = 0; l = P.bwt[0].seq_len; st = ST_WIDTH; }
					else { P.n_aln[item] = 0; P.max_ent[item] = 0; P.status[item] = NABWA_ST_OK; }
				} else st = ST_EXIT;
			}
		}
		if (__ballot(st != ST_EXIT) == 0ull) break;

		bool finish = false;

		// ---------------------------------------------------------------- A: pop + pre-checks
		if (st == ST_POP) {
			if (n_entries == 0) finish = true;
			else {
				if (max_ent < n_entries) max_ent = n_entries;
				if (n_entries > P.max_entries) finish = true;
			}
			if (!finish) {
				const int best = mask_lo ? __ffsll((unsigned long long)mask_lo) - 1 : 64 + __ffsll((unsigned long long)mask_hi) - 1;
				const LinkT s = head[best];
				const uint4 e = ent[s];
				const LinkT nx = lnk[s];
				head[best] = nx;
				if (nx == NIL) { if (best < 64) mask_lo &= ~(1ull << best); else mask_hi &= ~(1ull << (best - 64)); }
				if (REUSE) freel[nfree++] = s;
				--n_entries;
				k = e.x; l = e.y;
				e_i = (int)(e.z & 0xffffu); e_ldp = (int)(e.z >> 16);
				e_mm = (int)(e.w & 0xffu); e_go = (int)(e.w >> 8 & 0xffu); e_ge = (int)(e.w >> 16 & 0xffu);
				e_state = (int)(e.w >> 24 & 3u); e_a = (int)(e.w >> 26 & 1u);
				if (!nonstop && best > best_score + P.s_mm) finish = true;      // bwtgap.c:144
				else {
					m = max_diff - (e_mm + e_go); if (gape_mode) m -= e_ge;
					m_seed = P.max_seed_diff - (e_mm + e_go); if (gape_mode) m_seed -= e_ge;
					bool skip = m < 0;
					if (!skip && e_i > 0 && m < (int)Bd[e_a * P.WL + e_i - 1]) skip = true;   // bwtgap.c:156
					if (!skip) {
						if (e_i == 0) {
							st = ST_EXACT; xt = -1;       // a hit as it stands; handled in stage C without a query
						} else if (m == 0 && (e_state == STATE_M || gape_mode || e_ge == P.max_gape)) {
							st = ST_EXACT; xt = e_i - 1;  // nothing may differ any more: exact tail (bwt.c:237-252)
						} else { st = ST_EXPAND; --e_i; }
					}
				}
			}
		}

		// ---------------------------------------------------------------- B: the rank step
		int qb = 0, c = 4; bool query = false;
		if (st == ST_WIDTH) {
			const int sbase = pass < 2 ? 0 : len - P.seed_len;
			qb = pass & 1;
			c = (qb ? sq1 : sq0)[sbase + wi];
			query = c < 4;
		} else if (st == ST_EXACT) {
			qb = 1 - e_a;
			if (xt >= 0) { c = (e_a ? sq1 : sq0)[xt]; query = c < 4; }
		} else if (st == ST_EXPAND) {
			qb = 1 - e_a; query = true;
			c = (e_a ? sq1 : sq0)[e_i];
		}
		Occ4 ck, cl;
		if (query) nabwa_occ4_pair(qb ? P.bwt[1] : P.bwt[0], k - 1u, l, ck, cl);
		if (COUNT && query) rd_touch += ref_touches(qb ? P.bwt[1] : P.bwt[0], k - 1u, l, st == ST_EXPAND);
		const uint32_t L2q0 = qb ? P.bwt[1].L2[0] : P.bwt[0].L2[0], L2q1 = qb ? P.bwt[1].L2[1] : P.bwt[0].L2[1];
		const uint32_t L2q2 = qb ? P.bwt[1].L2[2] : P.bwt[0].L2[2], L2q3 = qb ? P.bwt[1].L2[3] : P.bwt[0].L2[3];
		const uint32_t seqlen_q = qb ? P.bwt[1].seq_len : P.bwt[0].seq_len;
#define L2Q(cc) ((cc) == 0 ? L2q0 : ((cc) == 1 ? L2q1 : ((cc) == 2 ? L2q2 : L2q3)))
#define CK(cc) ((cc) == 0 ? ck.c[0] : ((cc) == 1 ? ck.c[1] : ((cc) == 2 ? ck.c[2] : ck.c[3])))
#define CL(cc) ((cc) == 0 ? cl.c[0] : ((cc) == 1 ? cl.c[1] : ((cc) == 2 ? cl.c[2] : cl.c[3])))

		// ---------------------------------------------------------------- C: consume the counts
		if (st == ST_WIDTH) {
			// one step of bwt_cal_width (bwtaln.c:52-76) on index `pass&1`
			const int n = pass < 2 ? len : P.seed_len;
			if (c < 4) { k = L2Q(c) + CK(c) + 1u; l = L2Q(c) + CL(c); }
			else if (pass == 0) ++nN;
			if (k > l || c > 3) { k = 0; l = seqlen_q; ++wbid; }
			uint32_t *wp = pass < 2 ? Wd + qb * P.WL : SWd + qb * P.SL;
			uint8_t *bp = pass < 2 ? Bd + qb * P.WL : SBd + qb * P.SL;
			wp[wi] = l - k + 1u; bp[wi] = (uint8_t)(wbid > 255 ? 255 : wbid);
			if (++wi == n) {
				++wbid; wp[n] = 0; bp[n] = (uint8_t)(wbid > 255 ? 255 : wbid);
				++pass;
				if (pass == 2 && len <= P.seed_len) pass = 4;
				if (pass < 4) { wi = 0; wbid = 0; k = 0; l = (pass & 1) ? P.bwt[1].seq_len : P.bwt[0].seq_len; }
				else {
					// ---- start of bwt_match_gap (bwtgap.c:104-128)
					seeded = len > P.seed_len;
					if (nN > md_read) finish = true;             // too many N: no search (bwtgap.c:118-123)
					else {
						max_diff = md_read;
						best_score = (md_read + 1) * P.s_mm + (mg_read + 1) * P.s_gapo + (P.max_gape + 1) * P.s_gape;
						best_cnt = 0;
						// roots: strand 0 pushed first, strand 1 second -> strand 1 is expanded first
						ent[0] = make_uint4(0u, P.bwt[0].seq_len, (uint32_t)len, 0u);
						ent[1] = make_uint4(0u, P.bwt[0].seq_len, (uint32_t)len, 1u << 26);
						lnk[0] = NIL; lnk[1] = (LinkT)0; head[0] = (LinkT)1;
						bump = 2; nfree = 0; n_entries = 2; mask_lo = 1ull; mask_hi = 0ull;
						st = ST_POP;
					}
				}
			}
		} else if (st == ST_EXACT) {
			bool hit = false;
			if (xt < 0) hit = true;
			else if (c > 3) st = ST_POP;                          // an N in the tail: no match
			else {
				k = L2Q(c) + CK(c) + 1u; l = L2Q(c) + CL(c);
				if (k > l) st = ST_POP;
				else if (--xt < 0) hit = true;
			}
			if (hit) {
				// ---- hit bookkeeping (bwtgap.c:166-199)
				st = ST_POP;
				const int score = e_mm * P.s_mm + e_go * P.s_gapo + e_ge * P.s_gape;
				bool do_add = true;
				if (n_aln == 0) {
					best_score = score;
					const int best_diff = e_mm + e_go + (gape_mode ? e_ge : 0);
					if (!nonstop) max_diff = best_diff + 1 > md_read ? md_read : best_diff + 1;
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
						// gap_shadow (bwtgap.c:81-91) on this strand's bounds, positions < last_diff_pos
						const uint32_t x = l - k + 1u, mx = seqlen_q; uint32_t jj = 0;
						uint32_t *wp = Wd + e_a * P.WL; uint8_t *bp = Bd + e_a * P.WL;
						for (int t0 = 0; t0 < e_ldp; t0 += 4) {
							uint4 w4 = *(const uint4*)(wp + t0);
							uint32_t wv[4] = { w4.x, w4.y, w4.z, w4.w };
#pragma unroll
							for (int u = 0; u < 4; ++u) {
								if (t0 + u < e_ldp) {
									if (wv[u] > x) wv[u] -= x;
									else if (wv[u] == x) { bp[t0 + u] = 1; wv[u] = mx - (++jj); }
								}
							}
							*(uint4*)(wp + t0) = make_uint4(wv[0], wv[1], wv[2], wv[3]);
						}
						out[n_aln] = make_uint4((uint32_t)e_mm | (uint32_t)e_go << 8 | (uint32_t)e_ge << 16 | (uint32_t)e_a << 24,
												k, l, (uint32_t)score);
						++n_aln;
					}
				}
			}
		} else if (st == ST_EXPAND) {
			// ---- node expansion (bwtgap.c:201-260); e_i is already decremented
			st = ST_POP;
			const uint32_t occ = l - k + 1u;
			const uint32_t *wp = Wd + e_a * P.WL; const uint8_t *bp = Bd + e_a * P.WL;
			bool allow_diff = true, allow_M = true;
			if (e_i > 0) {
				const int b1 = bp[e_i - 1], b0 = bp[e_i];
				if (b1 > m - 1) allow_diff = false;
				else if (b1 == m - 1 && b0 == m - 1 && wp[e_i - 1] == wp[e_i]) allow_M = false;
				const int ii = e_i - (len - P.seed_len);
				if (seeded && ii > 0) {
					const uint32_t *swp = SWd + e_a * P.SL; const uint8_t *sbp = SBd + e_a * P.SL;
					const int s1 = sbp[ii - 1], s0 = sbp[ii];
					if (s1 > m_seed - 1) allow_diff = false;
					else if (s1 == m_seed - 1 && s0 == m_seed - 1 && swp[ii - 1] == swp[ii]) allow_M = false;
				}
			}
			// children are appended to per-score lists; consecutive pushes of one score chain locally
			int cs = -1; LinkT ch = NIL; bool ovf = false;
			auto push = [&](int score, uint32_t nk, uint32_t nl, int ni, int nmm, int ngo, int nge, int nstate, bool is_diff) {
				if (ovf) return;
				if (score != cs) {
					if (cs >= 0) head[cs] = ch;
					cs = score;
					const bool has = score < 64 ? (mask_lo >> score & 1ull) : (mask_hi >> (score - 64) & 1ull);
					ch = has ? head[score] : NIL;
				}
				uint32_t s;
				if (REUSE && nfree) s = freel[--nfree];
				else { if (bump >= P.cap) { ovf = true; return; } s = bump++; }
				ent[s] = make_uint4(nk, nl, (uint32_t)ni | (uint32_t)(is_diff ? ni : 0) << 16,
									(uint32_t)(nmm & 0xff) | (uint32_t)(ngo & 0xff) << 8 | (uint32_t)(nge & 0xff) << 16 |
									(uint32_t)nstate << 24 | (uint32_t)e_a << 26);
				lnk[s] = ch; ch = (LinkT)s;
				if (score < 64) mask_lo |= 1ull << score; else mask_hi |= 1ull << (score - 64);
				++n_entries;
			};
			const int sc0 = e_mm * P.s_mm + e_go * P.s_gapo + e_ge * P.s_gape;
			int tmp = e_go + e_ge;
			if (loggap) { const uint32_t v = (uint32_t)(e_ge + e_go); tmp = (v ? 31 - __clz((int)v) : 0) / 2 + 1; }
			if (allow_diff && e_i >= P.indel_end_skip + tmp && len - e_i >= P.indel_end_skip + tmp) {
				if (e_state == STATE_M) {
					if (e_go < mg_read) {
						push(sc0 + P.s_gapo, k, l, e_i, e_mm, e_go + 1, e_ge, STATE_I, true);
#pragma unroll
						for (int j = 0; j < 4; ++j) {
							const uint32_t nk = L2Q(j) + CK(j) + 1u, nl = L2Q(j) + CL(j);
							if (nk <= nl) push(sc0 + P.s_gapo, nk, nl, e_i + 1, e_mm, e_go + 1, e_ge, STATE_D, true);
						}
					}
				} else if (e_state == STATE_I) {
					if (e_ge < P.max_gape) push(sc0 + P.s_gape, k, l, e_i, e_mm, e_go, e_ge + 1, STATE_I, true);
				} else if (e_ge < P.max_gape) {
					if (e_ge + e_go < max_diff || occ < (uint32_t)P.max_del_occ) {
#pragma unroll
						for (int j = 0; j < 4; ++j) {
							const uint32_t nk = L2Q(j) + CK(j) + 1u, nl = L2Q(j) + CL(j);
							if (nk <= nl) push(sc0 + P.s_gape, nk, nl, e_i + 1, e_mm, e_go, e_ge + 1, STATE_D, true);
						}
					}
				}
			}
			if (allow_diff && allow_M) {
#pragma unroll
				for (int j = 1; j <= 4; ++j) {
					const int cc = (c + j) & 3; const bool is_mm = (j != 4 || c > 3);
					const uint32_t nk = L2Q(cc) + CK(cc) + 1u, nl = L2Q(cc) + CL(cc);
					if (nk <= nl) push(sc0 + (is_mm ? P.s_mm : 0), nk, nl, e_i, e_mm + (is_mm ? 1 : 0), e_go, e_ge, STATE_M, is_mm);
				}
			} else if (c < 4) {
				const uint32_t nk = L2Q(c) + CK(c) + 1u, nl = L2Q(c) + CL(c);
				if (nk <= nl) push(sc0, nk, nl, e_i, e_mm, e_go, e_ge, STATE_M, false);
			}
			if (cs >= 0) head[cs] = ch;
			if (ovf) { status = NABWA_ST_OVERFLOW; finish = true; }
		}
#undef L2Q
#undef CK
#undef CL

		if (finish) {
			P.n_aln[item] = n_aln; P.max_ent[item] = max_ent; P.status[item] = (uint8_t)status;
			if (COUNT && status == NABWA_ST_OK) touches += rd_touch;   // abandoned reads are counted by the wide pass
			st = ST_IDLE;
		}
	}
	if (COUNT) {
		for (int o = 32; o > 0; o >>= 1) touches += __shfl_down(touches, o);
		if (lane == 0 && P.touch_counter) atomicAdd(P.touch_counter, touches);
	}
}

extern "C" void nabwa_launch_fm_search(const SearchParams *P, int n_blocks, int wide, hipStream_t s)
{
	if (P->touch_counter) {
		if (wide) hipLaunchKernelGGL((fm_search_kernel<uint32_t, true, true>), dim3(n_blocks), dim3(NABWA_SEARCH_BLOCK), 0, s, *P);
		else hipLaunchKernelGGL((fm_search_kernel<uint16_t, false, true>), dim3(n_blocks), dim3(NABWA_SEARCH_BLOCK), 0, s, *P);
	} else if (wide) hipLaunchKernelGGL((fm_search_kernel<uint32_t, true, false>), dim3(n_blocks), dim3(NABWA_SEARCH_BLOCK), 0, s, *P);
	else hipLaunchKernelGGL((fm_search_kernel<uint16_t, false, false>), dim3(n_blocks), dim3(NABWA_SEARCH_BLOCK), 0, s, *P);
}

extern "C" int nabwa_search_occupancy(int wide)
{
	int nb = 0;
	hipError_t e = wide ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, fm_search_kernel<uint32_t, true, false>, NABWA_SEARCH_BLOCK, 0)
						: hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, fm_search_kernel<uint16_t, false, false>, NABWA_SEARCH_BLOCK, 0);
	return e == hipSuccess ? nb : 0;
}

// ids of the reads whose first pass was abandoned (arena or hit list outgrown)
__global__ __launch_bounds__(256) void collect_kernel(int n, const uint8_t *__restrict__ status, int32_t *__restrict__ ids,
												  unsigned int *__restrict__ count)
{
	const int i = blockIdx.x * 256 + threadIdx.x;
	if (i < n && status[i] == NABWA_ST_OVERFLOW) ids[atomicAdd(count, 1u)] = i;
}

extern "C" void nabwa_launch_collect(int n, const uint8_t *status, int32_t *ids, unsigned int *count, hipStream_t s)
{
	if (n <= 0) return;
	hipLaunchKernelGGL(collect_kernel, dim3((n + 255) / 256), dim3(256), 0, s, n, status, ids, count);
}

#define NABWA_ST_WIDE 2   // result lives in the wide pass's buffers at wide_idx[read]

__global__ __launch_bounds__(256) void scatter_wide_kernel(int n2, const int32_t *__restrict__ ids, const int32_t *__restrict__ n_aln2,
													   const int32_t *__restrict__ max_ent2, const uint8_t *__restrict__ status2,
													   int32_t *__restrict__ n_aln, int32_t *__restrict__ max_ent,
													   uint8_t *__restrict__ status, int32_t *__restrict__ wide_idx)
{
	const int j = blockIdx.x * 256 + threadIdx.x;
	if (j >= n2) return;
	const int rid = ids[j];
	if (status2[j] == NABWA_ST_OK) { n_aln[rid] = n_aln2[j]; max_ent[rid] = max_ent2[j]; wide_idx[rid] = j; status[rid] = NABWA_ST_WIDE; }
	else { n_aln[rid] = 0; max_ent[rid] = max_ent2[j]; status[rid] = NABWA_ST_OVERFLOW; }
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
												const int32_t *wide_idx, const uint4 *aln2, int aln_cap2)
{
	return status[i] == NABWA_ST_WIDE ? aln2 + (size_t)wide_idx[i] * aln_cap2 : aln + (size_t)i * aln_cap;
}

// compaction: rows of read i go to out[row_off[i] ...]
__global__ __launch_bounds__(256) void gather_kernel(int n, const int32_t *__restrict__ n_aln, const uint32_t *__restrict__ row_off,
												 const uint4 *__restrict__ aln, int aln_cap, const uint8_t *__restrict__ status,
												 const int32_t *__restrict__ wide_idx, const uint4 *__restrict__ aln2, int aln_cap2,
												 uint4 *__restrict__ out)
{
	const int i = blockIdx.x * 256 + threadIdx.x;
	if (i >= n) return;
	const int na = n_aln[i];
	const uint4 *src = rows_of(i, aln, aln_cap, status, wide_idx, aln2, aln_cap2);
	uint4 *dst = out + row_off[i];
	for (int j = 0; j < na; ++j) dst[j] = src[j];
}

extern "C" void nabwa_launch_gather(int n, const int32_t *n_aln, const uint32_t *row_off, const uint4 *aln, int aln_cap,
									const uint8_t *status, const int32_t *wide_idx, const uint4 *aln2, int aln_cap2,
									uint4 *out, hipStream_t s)
{
	if (n <= 0) return;
	hipLaunchKernelGGL(gather_kernel, dim3((n + 255) / 256), dim3(256), 0, s, n, n_aln, row_off, aln, aln_cap, status,
					   wide_idx, aln2, aln_cap2, out);
}

// order-independent checksum over all hits: sum of a mix of (read, row index, row words)
__device__ __forceinline__ uint64_t mix64(uint64_t x)
{
	x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
	return x;
}

__global__ __launch_bounds__(256) void checksum_kernel(int n, const int32_t *__restrict__ n_aln, const uint4 *__restrict__ aln,
												   int aln_cap, const uint8_t *__restrict__ status, const int32_t *__restrict__ wide_idx,
												   const uint4 *__restrict__ aln2, int aln_cap2,
												   unsigned long long *__restrict__ sum, unsigned long long *__restrict__ rows)
{
	const int i = blockIdx.x * 256 + threadIdx.x;
	uint64_t s = 0, r = 0;
	if (i < n) {
		const int na = n_aln[i];
		const uint4 *src = rows_of(i, aln, aln_cap, status, wide_idx, aln2, aln_cap2);
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
									  unsigned long long *sum, unsigned long long *rows, hipStream_t s)
{
	if (n <= 0) return;
	hipLaunchKernelGGL(checksum_kernel, dim3((n + 255) / 256), dim3(256), 0, s, n, n_aln, aln, aln_cap, status, wide_idx,
					   aln2, aln_cap2, sum, rows);
}
