// dp_global.hip -- banded global alignment with traceback, one (reference window, read) pair per lane.
//
// Semantics of aln_global_core (stdaln.c:345-525) + aln_path2cigar32 (stdaln.c:1009-1039):
// three-state (M/I/D) affine-gap DP over int32 scores, band b1/b2 derived from band_width and the
// length difference (stdaln.c:372-380), end-gap penalty on column 0, column len1 and row len2
// (set_end_I / set_end_D, stdaln.c:286-319), tie-breaking of set_M (M >= I, M >= D, else I > D:
// stdaln.c:260-275), backtrace preferring M, then I, then D on strictly greater (stdaln.c:491-493).
// Stated through per-row column ranges: row j covers columns max(0,j-b2) .. min(len1, j+b1-1);
// row 0 covers 0 .. b1-1.
//
// Layout: every lane owns two score rows (M,I,D) and a traceback matrix in HBM, interleaved by lane
// inside the wave ([cell][lane]) so that lanes walking the same cell index coalesce.
// Used for gap refinement (refine_gapped_core, bwase.c:189-237): only reads whose best hit has a gap
// open come here, so this kernel is small next to the FM search.
#include <hip/hip_runtime.h>
#include <stdint.h>

#define NINF (-1073741823)   // MINOR_INF, stdaln.h:84
#define FM 0
#define FI 1
#define FD 2

#include "dp_params.hpp"

// Two forms: LDSV = false keeps the score rows of a wave's 64 tasks interleaved in HBM (tens of
// thousands of tasks hide each other's round trips); LDSV = true is for a handful of tasks (the paths of mate rescue, a few per
// batch, each a chain of dependent row reads): eight tasks per block, rows and reference
// window in LDS.  The traceback matrix and the path stay in HBM in both (written once, read once along the path).
#define DP_SMALL_LANES 8
template <bool LDSV>
__global__ __launch_bounds__(LDSV ? DP_SMALL_LANES : 256) void dp_global_kernel(const DpParams P)
{
	extern __shared__ int32_t glo_lds[];             // LDSV: [6][W][8] row words, then 8 windows of W bytes
	constexpr int BT = LDSV ? DP_SMALL_LANES : 256, RS = LDSV ? DP_SMALL_LANES : 64;
	const int t = blockIdx.x * BT + threadIdx.x;
	const int lane = t & 63;
	const size_t wave = (size_t)(t >> 6);            // task t owns lane t % 64 of region t / 64 of the HBM scratch in either form
	if (t >= P.n) return;
	const uint8_t *s1 = P.ref + P.ref_off[t], *s2 = P.qry + P.qry_off[t];
	const int l1 = (int)(P.ref_off[t + 1] - P.ref_off[t]), l2 = (int)(P.qry_off[t + 1] - P.qry_off[t]);
	P.n_cigar[t] = 0; P.score[t] = 0;
	if (l1 == 0 || l2 == 0) return;
	const int W = P.W;
	int32_t *R = LDSV ? glo_lds + threadIdx.x : P.rows + wave * 6 * (size_t)W * 64 + lane;
	if (LDSV) {
		uint8_t *win = (uint8_t*)(glo_lds + 6 * (size_t)W * DP_SMALL_LANES) + (size_t)threadIdx.x * W;
		for (int i = 0; i < l1; ++i) win[i] = s1[i];
		s1 = win;
	}
	uint8_t *TB = P.tb + wave * (size_t)P.H * W * 64 + lane;
	uint8_t *PATH = P.path + wave * (size_t)(W + P.H) * 64 + lane;
#define ROW(arr, par, i) R[(((arr) * 2 + (par)) * (size_t)W + (i)) * RS]     // arr: 0 M, 1 I, 2 D; par: row parity
#define TBC(j, i) TB[((size_t)(j) * W + (i)) * 64]
	const int gap_open = P.gap_open, gap_ext = P.gap_ext;
	const int end_pen = P.gap_end >= 0 ? P.gap_end : P.gap_ext;
	int b1, b2;
	if (l1 > l2) { b1 = l1 - l2 + P.band; b2 = P.band; } else { b1 = P.band; b2 = l2 - l1 + P.band; }
	if (b1 > l1) b1 = l1;
	if (b2 > l2) b2 = l2;
	// row 0
	ROW(0, 0, 0) = 0; ROW(1, 0, 0) = NINF; ROW(2, 0, 0) = NINF;
	{
		int pm = 0, pd = NINF;
		for (int i = 1; i < b1; ++i) {
			int d, tt;
			if (pm - gap_open > pd) { tt = FM; d = pm - gap_open - end_pen; } else { tt = FD; d = pd - end_pen; }
			ROW(0, 0, i) = NINF; ROW(1, 0, i) = NINF; ROW(2, 0, i) = d;
			TBC(0, i) = (uint8_t)(tt << 4);
			pm = NINF; pd = d;
		}
	}
	for (int j = 1; j <= l2; ++j) {
		const int cur = j & 1, prv = cur ^ 1;
		const int left = j > b2 ? j - b2 : 0, right = j + b1 - 1 < l1 ? j + b1 - 1 : l1;
		const int *mat = P.matrix + s2[j - 1] * 5;
		const int mt0 = mat[0], mt1 = mat[1], mt2 = mat[2], mt3 = mat[3], mt4 = mat[4];      // this row's scores against A, C, G, T, N
		const int dpen = j == l2 ? end_pen : gap_ext;
		int cm_l = NINF, cd_l = NINF;          // M and D of the cell to the left in this row
		ROW(0, cur, left) = NINF; ROW(1, cur, left) = NINF; ROW(2, cur, left) = NINF; TBC(j, left) = 0;
		if (left == 0) {                       // column 0: end-gap insertion chain
			const int pm0 = ROW(0, prv, 0), pi0 = ROW(1, prv, 0);
			int v, tt;
			if (pm0 - gap_open > pi0) { tt = FM; v = pm0 - gap_open - end_pen; } else { tt = FI; v = pi0 - end_pen; }
			ROW(1, cur, 0) = v; TBC(j, 0) = (uint8_t)(tt << 2);
		}
		int pm_d = ROW(0, prv, left), pi_d = ROW(1, prv, left), pd_d = ROW(2, prv, left);   // diagonal cell (i-1) of the previous row
		// a lane's cells are a chain of dependent loads when taken one at a time (the previous row's three scores, the base, its
		// score): DPB columns are fetched together -- the previous row is not written in this row, so nothing read here is stale
#define DPB 4
		for (int i0 = left + 1; i0 <= right; i0 += DPB) {
			int um[DPB], ui[DPB], ud[DPB], ub[DPB];
#pragma unroll
			for (int u = 0; u < DPB; ++u) {
				const int i = i0 + u;
				um[u] = ui[u] = ud[u] = NINF; ub[u] = 4;
				if (i <= right) {
					ub[u] = s1[i - 1];
					// the cell above (row j-1, column i): needed for I, and it is the next column's diagonal
					const bool above = !(i == right && !(j + b1 - 1 > l1));
					if (above || i < right) { um[u] = ROW(0, prv, i); ui[u] = ROW(1, prv, i); ud[u] = ROW(2, prv, i); }
				}
			}
#pragma unroll
			for (int u = 0; u < DPB; ++u) {
				const int i = i0 + u;
				if (i > right) break;
				const int b = ub[u];
				const int sc = b == 0 ? mt0 : (b == 1 ? mt1 : (b == 2 ? mt2 : (b == 3 ? mt3 : mt4)));
				int m, mt, iv = NINF, it = 0, dv, dt;
				if (pm_d >= pi_d) { if (pm_d >= pd_d) { m = pm_d + sc; mt = FM; } else { m = pd_d + sc; mt = FD; } }
				else { if (pi_d > pd_d) { m = pi_d + sc; mt = FI; } else { m = pd_d + sc; mt = FD; } }
				const bool above = !(i == right && !(j + b1 - 1 > l1));
				const int pm_u = um[u], pi_u = ui[u], pd_u = ud[u];
				if (above) {
					const int ipen = i == l1 ? end_pen : gap_ext;
					if (pm_u - gap_open > pi_u) { it = FM; iv = pm_u - gap_open - ipen; } else { it = FI; iv = pi_u - ipen; }
				}
				if (cm_l - gap_open > cd_l) { dt = FM; dv = cm_l - gap_open - dpen; } else { dt = FD; dv = cd_l - dpen; }
				ROW(0, cur, i) = m; ROW(1, cur, i) = iv; ROW(2, cur, i) = dv;
				TBC(j, i) = (uint8_t)(mt | it << 2 | dt << 4);
				cm_l = m; cd_l = dv;
				pm_d = pm_u; pi_d = pi_u; pd_d = pd_u;
			}
		}
#undef DPB
	}
	// backtrace
	int i = l1, j = l2, score, type, ctype;
	{
		const int lm = ROW(0, l2 & 1, l1), li = ROW(1, l2 & 1, l1), ld = ROW(2, l2 & 1, l1);
		const uint8_t q = TBC(l2, l1);
		score = lm; type = q & 3; ctype = FM;
		if (li > score) { score = li; type = q >> 2 & 3; ctype = FI; }
		if (ld > score) { score = ld; type = q >> 4 & 3; ctype = FD; }
	}
	int plen = 0;
	PATH[(size_t)plen++ * 64] = (uint8_t)ctype;
	do {
		if (ctype == FM) { --i; --j; } else if (ctype == FI) --j; else --i;
		const uint8_t q = TBC(j, i);
		ctype = type;
		type = type == FM ? (q & 3) : (type == FI ? (q >> 2 & 3) : (q >> 4 & 3));
		PATH[(size_t)plen++ * 64] = (uint8_t)ctype;
	} while (i || j);
	--plen;                        // the entry written at (0,0) is not part of the path
	// run-length encode from the path's end: cigar32 = len<<4 | op
	// (operation k of task t at cigar[k * n_tasks + t]: the host fetches only as many operation slots as the longest CIGAR has)
	uint32_t *cg = P.cigar + t;
	const size_t cs = (size_t)P.n;
	int n = 0; uint32_t curc = 0;
	for (int p = plen - 1; p >= 0; --p) {
		const uint32_t op = PATH[(size_t)p * 64];
		if (n && (curc & 0xf) == op) curc += 1u << 4;
		else {
			if (n && n <= P.max_cigar) cg[(size_t)(n - 1) * cs] = curc;
			curc = 1u << 4 | op; ++n;
		}
	}
	if (n && n <= P.max_cigar) cg[(size_t)(n - 1) * cs] = curc;
	P.n_cigar[t] = n;              // n > max_cigar signals truncation to the host
	P.score[t] = score;
#undef ROW
#undef TBC
}

extern "C" void nabwa_launch_dp_global(const DpParams *P, hipStream_t s)
{
	if (P->n <= 0) return;
	const int small_max = getenv("NABWA_DP_SMALL") ? atoi(getenv("NABWA_DP_SMALL")) : 4096;           // tasks up to which the LDS form runs (0: never)
	const size_t lds = (size_t)P->W * DP_SMALL_LANES * 25;                                            // six rows of words + the window
	if (P->n <= small_max && lds <= 60000)
		hipLaunchKernelGGL(dp_global_kernel<true>, dim3((P->n + DP_SMALL_LANES - 1) / DP_SMALL_LANES), dim3(DP_SMALL_LANES), lds, s, *P);
	else
		hipLaunchKernelGGL(dp_global_kernel<false>, dim3((P->n + 255) / 256), dim3(256), 0, s, *P);
}

// aln_local_core and aln_extend_core (forward / reverse Smith-Waterman passes): dp_wave.hip, one wavefront per task.
