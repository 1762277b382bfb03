// dp_wave.hip -- the two Smith-Waterman forms of the path (aln_local_core for mate rescue, aln_extend_core), ONE WAVEFRONT PER TASK.
//
// What the reference computes (results only; stdaln.c:529-761 and :862-1007), stated as recurrences over a (read row j, window
// column i) matrix with gap penalties q (open) and r (extend), qr = q + r, and substitution scores s(j, i):
//
//   local, forward     H(j,i) = max(0, H(j-1,i-1) + s, V(j,i), G(j,i))
//                      V(j,i) = H(j-1,i) >= qr + 1 ? max(V(j-1,i) - r, H(j-1,i) - qr) : 0          (a vertical gap only leaves a cell worth it)
//                      G(j,i) = max over k < i of H(j,k) - qr - (i-1-k) r                          (horizontal gap)
//                      result: the best H, the FIRST cell in row-major order that has it, and every row's maximum
//   local, reverse     from the end cell back towards (1,1), the same H / G, V(j,i) = max(0, V(j+1,i) - r, H(j+1,i) - qr), over a band of
//                      columns (end, start] per row that the rows already done decide: `start` steps left when its cell is not worth
//                      a gap, `end` is as far left as the best cell so far could still be beaten from.  It stops at the first cell,
//                      in its scan order, whose score -- a new maximum -- is the forward score: that cell is the alignment's start.
//   extension          H(j,i) = max(H(j-1,i-1) ? H(j-1,i-1) + s : 0, V(j,i), G(j,i)), anchored in (0,0) with the seed score G0;
//                      V(j+1,i) = max(V(j,i) - r, H(j,i) - qr, 0), G likewise; columns [start, end) per row, from the first to three
//                      past the last cell that was positive in the row above, inside the caller's band.
//   Scores are kept in 16 bits there: whenever the best exceeds 32000 every stored value drops by 16000 (floored at 0) before
//   the next row.  That is part of the results (small cells vanish), so it is done here too.
//
// How it is computed here.  A lane per task walks these cells one at a time and keeps its rows in HBM; a wave per task keeps the
// rows in LDS and spends its 64 lanes on the cells:
//   * forward local pass: 64 consecutive rows move along the anti-diagonals (cell (j,i) needs H(j-1,i-1), H(j-1,i), V(j-1,i):
//     what the lane above produced one and two steps earlier, handed down by a lane shift); strip after strip of 64 rows, the
//     last row of a strip left in LDS for the next;
//   * every pass whose columns depend on the row before (reverse pass, extension) or that may need the 16-bit drop (reads beyond
//     ~2900 bases) goes ROW BY ROW, 64 columns per step: the cells' H without the horizontal gap all at once, the gap by a prefix
//     maximum over the lanes (G(i) = max_k (H0(k) + k r) - qr - (i-1) r: opening a gap from a cell that itself came out of a gap
//     never beats extending that gap, so H0 serves for H), "first cell with the best score" and the reverse pass's stop by ballots.
// A launch of two rescue alignments and a launch of 100 000 are the same kernel; windows too long for LDS keep their rows in HBM
// (template flag), same code.
#include <hip/hip_runtime.h>
#include <stdint.h>

#define DPW_NEG   (-(1 << 29))
#define OVF_LIMIT 32000          /* LOCAL_OVERFLOW_THRESHOLD, stdaln.c:252 */
#define OVF_STEP  16000          /* LOCAL_OVERFLOW_REDUCE, stdaln.c:253 */

#include "dp_params.hpp"
#include <stdlib.h>
#include <string.h>

__device__ __forceinline__ int dpw_max(int a, int b) { return a > b ? a : b; }
/* maximum over the wave */
__device__ __forceinline__ int dpw_wave_max(int v)
{
#pragma unroll
	for (int o = 32; o; o >>= 1) v = dpw_max(v, __shfl_xor(v, o));
	return v;
}
/* maximum over the lower lanes (lane 0: DPW_NEG) */
__device__ __forceinline__ int dpw_below_max(int v, int lane)
{
#pragma unroll
	for (int o = 1; o < 64; o <<= 1) { const int u = __shfl_up(v, o); if (lane >= o) v = dpw_max(v, u); }
	const int e = __shfl_up(v, 1);
	return lane ? e : DPW_NEG;
}
__device__ __forceinline__ int dpw_first_lane(bool c) { return __ffsll((unsigned long long)__ballot(c)) - 1; }
__device__ __forceinline__ int dpw_sub(const int *mat, int b) { return mat[b > 4 ? 4 : b]; }

// ---------------------------------------------------------------------------------------------------------------- local alignment
template <bool HBM>
__global__ __launch_bounds__(64) void dp_local_wave_kernel(const LocParams P)
{
	extern __shared__ int32_t wav_lds[];             // !HBM: two rows of W words, then the window's W bytes
	const int t = blockIdx.x, lane = threadIdx.x;
	if (t >= P.n) return;
	const uint8_t *s1g = P.ref + P.ref_off[t] - 1, *s2 = P.qry + P.qry_off[t] - 1;   // 1-based: s1[i], s2[j]
	const int l1 = (int)(P.ref_off[t + 1] - P.ref_off[t]), l2 = (int)(P.qry_off[t + 1] - P.qry_off[t]);
	int32_t *out = P.out + (size_t)t * 6;
	if (lane == 0) { out[0] = -1; out[1] = 0; out[2] = out[3] = out[4] = out[5] = 0; }
	if (l1 == 0 || l2 == 0) return;
	const int W = P.W;
	int32_t *rh = HBM ? P.eh + (size_t)t * 2 * W : wav_lds;     // row of H, row of V: indexes 0 .. l1 (+1)
	int32_t *rv = rh + W;
	const uint8_t *s1 = s1g;
	if (!HBM) {
		uint8_t *win = (uint8_t*)(wav_lds + 2 * (size_t)W);
		for (int i = lane; i <= l1; i += 64) win[i] = i ? s1g[i] : (uint8_t)4;
		s1 = win;
	}
	for (int i = lane; i < W; i += 64) { rh[i] = 0; rv[i] = 0; }
	__syncthreads();
	int32_t *suba = P.suba + (size_t)t * P.H;
	const int q = P.gap_open, r = P.gap_ext, qr = q + r;
	if (lane == 0) suba[0] = 0;
	int score_f = 0, end_i = 0, end_j = 0;

	if ((long long)l2 * P.max_score <= OVF_LIMIT && !P.row_forward) {
		// ---- forward pass along the anti-diagonals (no score of this task can reach the 16-bit drop)
		int best = 0, best_i = 0, best_j = 0;            // this lane's rows: the first cell with their best score
		for (int j0 = 1; j0 <= l2; j0 += 64) {
			const int j = j0 + lane;
			const bool live = j <= l2;
			const int *mat = P.matrix + (live ? s2[j] : 4) * 5;
			const int mt0 = mat[0], mt1 = mat[1], mt2 = mat[2], mt3 = mat[3], mt4 = mat[4];
			const bool writes = live && (lane == 63 || j == l2);      // the strip's last row: it is "the row above" for the next strip
			int left = 0, g = 0, rowmax = 0, diag = 0, o_h = 0, o_v = 0;
			for (int d = 0; d < l1 + 63; ++d) {
				const int i = d - lane + 1;
				int uh = __shfl_up(o_h, 1), uv = __shfl_up(o_v, 1);          // H(j-1, i), V(j-1, i): the lane above's last step
				const bool act = live && i >= 1 && i <= l1;
				if (lane == 0 && act) { uh = rh[i]; uv = rv[i]; }
				if (act) {
					const int b = s1[i];
					int h = diag + (b == 0 ? mt0 : (b == 1 ? mt1 : (b == 2 ? mt2 : (b == 3 ? mt3 : mt4))));
					if (h < 0) h = 0;
					if (left > 0) { g = dpw_max(g - r, left - qr); if (h < g) h = g; }
					int v = 0;
					if (uh >= qr + 1) { v = dpw_max(uv - r, uh - qr); if (h < v) h = v; }
					left = h;
					if (rowmax < h) rowmax = h;
					if (best < h) { best = h; best_i = i; best_j = j; }
					diag = uh; o_h = h; o_v = v;
					if (writes) { rh[i] = h; rv[i] = v; }
				}
			}
			if (live) suba[j] = rowmax;
			__syncthreads();
		}
		// the first cell, in row-major order, with the best score of all
		score_f = best; end_i = best_i; end_j = best_j;
		for (int o = 32; o; o >>= 1) {
			const int s_o = __shfl_xor(score_f, o), i_o = __shfl_xor(end_i, o), j_o = __shfl_xor(end_j, o);
			if (s_o > score_f || (s_o == score_f && (j_o < end_j || (j_o == end_j && i_o < end_i)))) { score_f = s_o; end_i = i_o; end_j = j_o; }
		}
		if (score_f == 0) { end_i = 0; end_j = 0; }      // no cell ever raised the running best above its start
	} else {
		// ---- forward pass row by row, with the 16-bit drop: rh[i] = H(j-1, i), rv[i] = V(j-1, i) when row j starts
		int base = 0; bool drop = false;
		for (int j = 1; j <= l2; ++j) {
			if (drop) {
				score_f -= OVF_STEP; base += OVF_STEP; drop = false;
				for (int i = lane; i <= l1; i += 64) { rh[i] = dpw_max(rh[i] - OVF_STEP, 0); rv[i] = dpw_max(rv[i] - OVF_STEP, 0); }
				__syncthreads();
			}
			const int *mat = P.matrix + s2[j] * 5;
			int reach = DPW_NEG, rowmax = 0, diag_in = 0;        // reach: max of H0(k) + k r over the columns done; diag_in: H(j-1, c0-1)
			for (int c0 = 1; c0 <= l1; c0 += 64) {
				const int i = c0 + lane;
				const bool act = i <= l1;
				const int up = act ? rh[i] : 0, upv = act ? rv[i] : 0;
				int dg = __shfl_up(up, 1);
				if (lane == 0) dg = diag_in;
				diag_in = __shfl(up, 63);
				int h = 0, v = 0, key = DPW_NEG;
				if (act) {
					h = dpw_max(dg + dpw_sub(mat, s1[i]), 0);
					if (up >= qr + 1) { v = dpw_max(upv - r, up - qr); h = dpw_max(h, v); }
					key = h + i * r;
				}
				const int before = dpw_max(dpw_below_max(key, lane), reach);
				if (act && before > DPW_NEG) h = dpw_max(h, before - qr - (i - 1) * r);
				reach = dpw_max(reach, dpw_wave_max(key));
				__syncthreads();                                 // every lane has read the row above before it is replaced
				if (act) { rh[i] = h; rv[i] = v; }
				const int cm = dpw_wave_max(act ? h : 0);
				if (cm > rowmax) rowmax = cm;
				if (cm > score_f) { score_f = cm; end_i = c0 + dpw_first_lane(act && h == cm); end_j = j; if (cm > OVF_LIMIT) drop = true; }
			}
			if (lane == 0) suba[j] = rowmax + base;
			__syncthreads();
		}
		score_f += base;
	}
	if (lane == 0) { out[0] = score_f; out[4] = end_i; out[5] = end_j; }
	if (score_f < P.thres || end_i == 0 || end_j == 0) return;

	// ---- reverse pass, row by row from the end cell: rh[x] = H(j+1, x), rv[x + 1] = V(j+1, x) when row j starts; columns (end, start]
	for (int i = lane; i <= end_i; i += 64) { rh[i] = 0; rv[i] = 0; }
	__syncthreads();
	int score_r = P.matrix[s1[end_i] * 5 + s2[end_j]];
	if (lane == 0) rh[end_i] = qr + score_r;                                 // so that the cell before it opens at its plain score
	int base = 0; bool drop = false;
	int start_i = end_i, start_j = end_j;
	int start = end_i - 1, end = end_i - 3;
	if (end <= 0) end = 0;
	__syncthreads();
	bool found = false;
	for (int j = end_j - 1; j >= 1 && !found; --j) {
		if (drop) {
			score_r -= OVF_STEP; base += OVF_STEP; drop = false;
			for (int x = start + 1 - lane; x >= end + 1; x -= 64) { rh[x] = dpw_max(rh[x] - OVF_STEP, 0); rv[x] = dpw_max(rv[x] - OVF_STEP, 0); }
			__syncthreads();
		}
		const int *mat = P.matrix + s2[j] * 5;
		int reach = DPW_NEG, right = 0;                      // right: H(j, c0 + 1), the cell done just before this step's first
		for (int c0 = start; c0 > end; c0 -= 64) {
			const int i = c0 - lane, p = start - i;              // p: the cell's place in the row's scan order
			const bool act = i > end;
			int h = 0, v = 0, key = DPW_NEG;
			if (act) {
				h = dpw_max(rh[i + 1] + dpw_sub(mat, s1[i]), 0);
				v = dpw_max(dpw_max(rv[i + 1] - r, rh[i] - qr), 0);
				h = dpw_max(h, v);
				key = h + p * r;
			}
			const int before = dpw_max(dpw_below_max(key, lane), reach);
			if (act && before > DPW_NEG) h = dpw_max(h, before - qr - (p - 1) * r);
			reach = dpw_max(reach, dpw_wave_max(key));
			int hr = __shfl_up(h, 1);
			if (lane == 0) hr = right;
			const int n_act = c0 - end < 64 ? c0 - end : 64;
			right = __shfl(h, n_act - 1);
			__syncthreads();                                     // the row below has been read
			if (act) { rh[i + 1] = hr; rv[i + 1] = v; }
			// a new best?  and is it the forward score: the first such cell in scan order ends the pass
			const int seen = dpw_max(dpw_below_max(act ? h : DPW_NEG, lane), score_r);
			const bool rises = act && h > seen;
			const int stop = dpw_first_lane(rises && h + base - qr == score_f);
			if (stop >= 0) { score_r = __shfl(h, stop); start_i = c0 - stop; start_j = j; found = true; break; }
			const int cm = dpw_wave_max(act ? h : DPW_NEG);
			if (cm > score_r) { score_r = cm; start_i = c0 - dpw_first_lane(act && h == cm); start_j = j; if (cm > OVF_LIMIT) drop = true; }
		}
		if (found) break;
		if (lane == 0) { rh[end + 1] = right; rv[end + 1] = 0; }         // the row's last cell; its V slot belongs to a column not done
		__syncthreads();
		if (rh[start] <= qr) --start;
		if (start <= 0) start = 0;
		end = start_i - (start_j - j) - (score_r + base + (start_j - j) * P.max_score) / r - 1;
		if (end <= 0) end = 0;
		__syncthreads();
	}
	if (lane == 0) { out[1] = score_r + base - qr; out[2] = start_i; out[3] = start_j; }
}

extern "C" size_t nabwa_dp_local_rows_bytes(int W) { return (size_t)W * 8; }      // HBM form: per task
extern "C" int nabwa_dp_local_fits_lds(int W)
{
	const char *e = getenv("NABWA_DP_ROWS");                 /* tests: "hbm" keeps the rows in HBM whatever the window */
	if (e && !strcmp(e, "hbm")) return 0;
	return (size_t)W * 9 + 16 <= 64000;
}

extern "C" void nabwa_launch_dp_local(const LocParams *P, hipStream_t s)
{
	if (P->n <= 0) return;
	if (nabwa_dp_local_fits_lds(P->W)) hipLaunchKernelGGL(dp_local_wave_kernel<false>, dim3(P->n), dim3(64), (size_t)P->W * 9 + 16, s, *P);
	else hipLaunchKernelGGL(dp_local_wave_kernel<true>, dim3(P->n), dim3(64), 0, s, *P);
}

// ---------------------------------------------------------------------------------------------------------------- extension
// rd[i] = H(j-1, i-1), rv[i] = V(j, i) when row j starts (the seed: rd[1] = G0).
template <bool HBM>
__global__ __launch_bounds__(64) void dp_extend_wave_kernel(const ExtParams P)
{
	extern __shared__ int32_t ext_lds[];             // !HBM: two rows of W words, then the window's W bytes
	const int t = blockIdx.x, lane = threadIdx.x;
	if (t >= P.n) return;
	const uint8_t *s1g = P.ref + P.ref_off[t] - 1, *s2 = P.qry + P.qry_off[t] - 1;   // 1-based
	const int l1 = (int)(P.ref_off[t + 1] - P.ref_off[t]), l2 = (int)(P.qry_off[t + 1] - P.qry_off[t]);
	if (lane == 0) { P.score[t] = -1; P.end_i[t] = 0; P.end_j[t] = 0; }
	if (l1 == 0 || l2 == 0) return;
	const int W = P.W;
	int32_t *rd = HBM ? (int32_t*)P.eh + (size_t)t * 2 * W : ext_lds;
	int32_t *rv = rd + W;
	const uint8_t *s1 = s1g;
	if (!HBM) {
		uint8_t *win = (uint8_t*)(ext_lds + 2 * (size_t)W);
		for (int i = lane; i <= l1; i += 64) win[i] = i ? s1g[i] : (uint8_t)4;
		s1 = win;
	}
	for (int i = lane; i < l1 + 2; i += 64) { rd[i] = 0; rv[i] = 0; }
	__syncthreads();
	if (lane == 0) rd[1] = P.g0[t] & 0xffff;                    // the seed sits in the 16 bits a score has there
	__syncthreads();
	const int r = P.gap_ext, qr = P.gap_open + P.gap_ext;
	int start = 1, end = 2, end_i = 0, end_j = 0, score = 0, base = 0; bool drop = false;
	for (int j = 1; j <= l2; ++j) {
		start = dpw_max(start, dpw_max(j - P.band, 1));
		{ int lim = j + P.band; if (lim > l1 + 1) lim = l1 + 1; if (lim < end) end = lim; }
		if (start >= end) break;
		if (drop) {
			score -= OVF_STEP; base += OVF_STEP; drop = false;
			for (int i = start + lane; i <= end; i += 64) { rd[i] = dpw_max(rd[i] - OVF_STEP, 0); rv[i] = dpw_max(rv[i] - OVF_STEP, 0); }
			__syncthreads();
		}
		const int *mat = P.matrix + s2[j] * 5;
		int reach = DPW_NEG, left = 0, first_pos = 0, last_pos = 0;    // left: H(j, c0 - 1)
		for (int c0 = start; c0 < end; c0 += 64) {
			const int i = c0 + lane;
			const bool act = i < end;
			int h = 0, v = 0, key = DPW_NEG;
			if (act) {
				const int d = rd[i];
				v = rv[i];
				h = d ? d + dpw_sub(mat, s1[i]) : 0;
				h = dpw_max(h, v);
				key = h + i * r;
			}
			// horizontal gap: max(0, max_k (H0(k) - qr, 0) - (i-1-k) r); a cell at 0 offers nothing that 0 does not
			const int before = dpw_max(dpw_below_max(key, lane), reach);
			if (act && before > DPW_NEG) h = dpw_max(h, before - qr - (i - 1) * r);
			reach = dpw_max(reach, dpw_wave_max(key));
			int hl = __shfl_up(h, 1);
			if (lane == 0) hl = left;
			const int n_act = end - c0 < 64 ? end - c0 : 64;
			left = __shfl(h, n_act - 1);
			__syncthreads();
			if (act) { rd[i] = hl; rv[i] = dpw_max(dpw_max(v - r, h - qr), 0); }
			const unsigned long long pos = (unsigned long long)__ballot(act && h > 0);
			if (pos) { if (!first_pos) first_pos = c0 + __ffsll(pos) - 1; last_pos = c0 + 63 - __clzll(pos); }
			const int cm = dpw_wave_max(act ? h : 0);
			if (cm > score) { score = cm; end_i = c0 + dpw_first_lane(act && h == cm); end_j = j; if (cm > OVF_LIMIT) drop = true; }
		}
		if (lane == 0) { rd[end] = left; rv[end] = 0; }
		__syncthreads();
		if (last_pos <= 0) break;                                // no cell of this row is positive: the extension has ended
		start = first_pos; end = last_pos + 3;
	}
	if (lane == 0) { P.score[t] = score + base - 1; P.end_i[t] = end_i; P.end_j[t] = end_j; }
}

extern "C" void nabwa_launch_dp_extend_fwd(const ExtParams *P, hipStream_t s)
{
	if (P->n <= 0) return;
	if (nabwa_dp_local_fits_lds(P->W)) hipLaunchKernelGGL(dp_extend_wave_kernel<false>, dim3(P->n), dim3(64), (size_t)P->W * 9 + 16, s, *P);
	else hipLaunchKernelGGL(dp_extend_wave_kernel<true>, dim3(P->n), dim3(64), 0, s, *P);
}
