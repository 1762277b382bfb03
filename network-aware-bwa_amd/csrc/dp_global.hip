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

// ---- one wavefront per task (round 3).  One pair per lane makes every lane walk its band cell by cell through score rows in HBM: a chain of
// dependent loads per lane, 24 ms for the 181 k refinement jobs of a million pairs whatever the rows' width.  A row's cells do not depend on
// each other that way: M(j, i) and I(j, i) come out of row j - 1 alone, and D(j, i) -- the best of "open at i" and "extend D(j, i - 1)" along
// the row, with the reference's tie (extend unless opening is STRICTLY better) -- is a prefix maximum: with c(x) = M(j, i - 1) - gap_open +
// dpen x for the x-th cell of the row, T(x) = max(T(x - 1), c(x)), T(-1) = minus infinity, D = T(x) - dpen (x + 1), and the direction bit
// is "c(x) > T(x - 1)".  So the wave takes a row at a time, two cells per lane: M and I for all cells at once from the row before (in LDS), the
// D's by one scan over the lanes; the directions of the whole task stay in LDS -- four bits a cell (M's predecessor, "I was opened", "D was
// opened"), a lane's two cells one byte, a row's band cells side by side; what is in LDS per task bounds the tasks a CU works on together, and a
// task is a chain of a few hundred dependent steps --, the walk back and the run-length coding read them there.  Tasks too large for LDS take the
// one-pair-per-lane form.
__device__ __forceinline__ int dpw_excl_max(int v, int lane)          // maximum of v over the lower lanes; INT_MIN for lane 0
{
#pragma unroll
	for (int o = 1; o < 64; o <<= 1) { const int u = __shfl_up(v, o); if (lane >= o) v = u > v ? u : v; }
	const int e = __shfl_up(v, 1);
	return lane ? e : (int)0x80000000;
}

__device__ __forceinline__ uint32_t dpw_tb(const uint8_t *tb, const uint8_t *tbl, int WH, int j, int c)      // c: the cell's column counted from the row's left edge
{
	if (c == 0) return tbl[j];
	const uint32_t q = tb[(size_t)j * WH + ((c - 1) >> 1)] >> (((c - 1) & 1) << 2) & 15u;
	return (q & 3u) | (q & 4u ? (uint32_t)FI << 2 : 0u) | (q & 8u ? (uint32_t)FD << 4 : 0u);
}

__global__ __launch_bounds__(64) void dp_global_wave_kernel(const DpParams P)
{
	extern __shared__ int32_t gw_lds[];
	const int t = blockIdx.x, lane = threadIdx.x;
	const int W = P.W, H = P.H, WB = P.wb;
	int32_t *const rows = gw_lds;                                   // [parity][M, I, D][W]
	const int WH = (WB + 1) / 2;                                   // bytes of a row's directions: the cells right of its left edge, two a byte
	uint8_t *const tb = (uint8_t*)(rows + 6 * (size_t)W);          // [H][WH]: cell left_j + 1 + x of row j in nibble x & 1 of byte x >> 1: mt | it << 2 | dt << 3 (it, dt: opened = 0)
	uint8_t *const tbl = tb + (size_t)H * WH;                      // [H]: the row's left edge cell, a byte as the other form stores it
	uint8_t *const path = tbl + H;                                 // [W + H]
	uint8_t *const w1 = path + (W + H), *const w2 = w1 + W;        // the window and the read
	const uint8_t *const s1 = P.ref + P.ref_off[t], *const s2 = P.qry + P.qry_off[t];
	const int l1 = (int)(P.ref_off[t + 1] - P.ref_off[t]), l2 = (int)(P.qry_off[t + 1] - P.qry_off[t]);
	if (lane == 0) { P.n_cigar[t] = 0; P.score[t] = 0; }
	if (l1 == 0 || l2 == 0) return;
	for (int i = lane; i < l1; i += 64) w1[i] = s1[i];
	for (int i = lane; i < l2; i += 64) w2[i] = s2[i];
#define RW(arr, par, i) rows[((par) * 3 + (arr)) * W + (i)]
	const int gap_open = P.gap_open, gap_ext = P.gap_ext;
	const int end_pen = P.gap_end >= 0 ? P.gap_end : P.gap_ext;
	int b1, b2;
	if (l1 > l2) { b1 = l1 - l2 + P.band; b2 = P.band; } else { b1 = P.band; b2 = l2 - l1 + P.band; }
	if (b1 > l1) b1 = l1;
	if (b2 > l2) b2 = l2;
	// row 0: D(0, i) = -gap_open - i end_pen, opened at column 1 and extended from there (set_end_D along the row)
	for (int i = lane; i < b1; i += 64) { RW(0, 0, i) = i ? NINF : 0; RW(1, 0, i) = NINF; RW(2, 0, i) = i ? -gap_open - i * end_pen : NINF; }
	for (int y = lane; 2 * y + 1 < b1; y += 64) tb[y] = (uint8_t)((y == 0 ? 0 : 8) | (2 * y + 2 < b1 ? 8 << 4 : 0));      // cells x = 2y, 2y + 1 (columns x + 1): D extended, but for column 1: opened
	if (lane == 0) tbl[0] = 0;
	__syncthreads();
	for (int j = 1; j <= l2; ++j) {
		const int cur = j & 1, prv = cur ^ 1;
		const int left = j > b2 ? j - b2 : 0, right = j + b1 - 1 < l1 ? j + b1 - 1 : l1;
		const int *const mat = P.matrix + w2[j - 1] * 5;
		const int mt0 = mat[0], mt1 = mat[1], mt2 = mat[2], mt3 = mat[3], mt4 = mat[4];
		const int dpen = j == l2 ? end_pen : gap_ext;
		uint8_t *const tbr = tb + (size_t)j * WH;
		if (lane == 0) {
			RW(0, cur, left) = NINF; RW(1, cur, left) = NINF; RW(2, cur, left) = NINF; tbl[j] = 0;
			if (left == 0) {                   // column 0: end-gap insertion chain
				const int pm0 = RW(0, prv, 0), pi0 = RW(1, prv, 0);
				int v, tt;
				if (pm0 - gap_open > pi0) { tt = FM; v = pm0 - gap_open - end_pen; } else { tt = FI; v = pi0 - end_pen; }
				RW(1, cur, 0) = v; tbl[j] = (uint8_t)(tt << 2);
			}
		}
		const int ncell = right - left;            // the row's cells: i = left + 1 + x, x = 0 .. ncell - 1
		const bool edge = j + b1 - 1 <= l1;        // the row's last cell has nothing above it (the band's upper edge)
		int carry_t = NINF, carry_m = NINF;        // T of the cells so far; M of the cell before the chunk
		for (int xb = 0; xb < ncell; xb += 128) {
			int mv[2], iv[2], cv[2], bits[2];
#pragma unroll
			for (int u = 0; u < 2; ++u) {
				const int x = xb + 2 * lane + u, i = left + 1 + x;
				mv[u] = NINF; iv[u] = NINF; cv[u] = (int)0x80000000; bits[u] = 0;
				if (x < ncell) {
					const int pm_d = RW(0, prv, i - 1), pi_d = RW(1, prv, i - 1), pd_d = RW(2, prv, i - 1);
					const int b = w1[i - 1];
					const int sc = b == 0 ? mt0 : (b == 1 ? mt1 : (b == 2 ? mt2 : (b == 3 ? mt3 : mt4)));
					int m, mt;
					if (pm_d >= pi_d) { if (pm_d >= pd_d) { m = pm_d + sc; mt = FM; } else { m = pd_d + sc; mt = FD; } }
					else { if (pi_d > pd_d) { m = pi_d + sc; mt = FI; } else { m = pd_d + sc; mt = FD; } }
					int it = 0, v = NINF;
					if (!(i == right && edge)) {
						const int pm_u = RW(0, prv, i), pi_u = RW(1, prv, i);
						const int ipen = i == l1 ? end_pen : gap_ext;
						if (pm_u - gap_open > pi_u) { it = FM; v = pm_u - gap_open - ipen; } else { it = FI; v = pi_u - ipen; }
					}
					mv[u] = m; iv[u] = v; bits[u] = mt | (it == FI ? 4 : 0);
				}
			}
			// D along the row: the cell before x = 0 of a chunk is the last of the chunk before (or the band's left edge: minus infinity)
			const int m_before = __shfl_up(mv[1], 1);
			const int prev0 = lane ? m_before : carry_m;
			{
				const int x0 = xb + 2 * lane;
				if (x0 < ncell) cv[0] = prev0 - gap_open + dpen * x0;
				if (x0 + 1 < ncell) cv[1] = mv[0] - gap_open + dpen * (x0 + 1);
			}
			const int pair_max = cv[0] > cv[1] ? cv[0] : cv[1];
			const int below = dpw_excl_max(pair_max, lane);
			int e0 = below > carry_t ? below : carry_t;
			int dv[2];
#pragma unroll
			for (int u = 0; u < 2; ++u) {
				const int x = xb + 2 * lane + u;
				const int tx = cv[u] > e0 ? cv[u] : e0;
				if (x < ncell) { bits[u] |= cv[u] > e0 ? 0 : 8; dv[u] = tx - dpen * (x + 1); } else dv[u] = NINF;
				e0 = tx;
			}
#pragma unroll
			for (int u = 0; u < 2; ++u) {
				const int x = xb + 2 * lane + u, i = left + 1 + x;
				if (x < ncell) { RW(0, cur, i) = mv[u]; RW(1, cur, i) = iv[u]; RW(2, cur, i) = dv[u]; }
			}
			if (xb + 2 * lane < ncell) tbr[(xb >> 1) + lane] = (uint8_t)(bits[0] | bits[1] << 4);
			// what the next chunk starts from: T and M of this chunk's last cell (lane 63's second; a row of more than 128 cells only)
			carry_t = __shfl(e0, 63); carry_m = __shfl(mv[1], 63);
		}
		__syncthreads();
	}
	// the walk back (stdaln.c:484-503) and the run-length coding, by all lanes alike (LDS reads of one address), one lane storing
	int i = l1, j = l2, score, type, ctype;
// the byte the other form keeps for cell (j_, i_): mt | it << 2 | dt << 4
#define TBW(j_, i_) dpw_tb(tb, tbl, WH, (j_), (i_) - ((j_) > b2 ? (j_) - b2 : 0))
	{
		const int lm = RW(0, l2 & 1, l1), li = RW(1, l2 & 1, l1), ld = RW(2, l2 & 1, l1);
		const uint8_t q = TBW(l2, l1);
		score = lm; type = q & 3; ctype = FM;
		if (li > score) { score = li; type = q >> 2 & 3; ctype = FI; }
		if (ld > score) { score = ld; type = q >> 4 & 3; ctype = FD; }
	}
	int plen = 0;
	if (lane == 0) path[plen] = (uint8_t)ctype;
	++plen;
	do {
		if (ctype == FM) { --i; --j; } else if (ctype == FI) --j; else --i;
		const uint8_t q = TBW(j, i);
		ctype = type;
		type = type == FM ? (q & 3) : (type == FI ? (q >> 2 & 3) : (q >> 4 & 3));
		if (lane == 0) path[plen] = (uint8_t)ctype;
		++plen;
	} while (i || j);
	--plen;                        // the entry written at (0,0) is not part of the path
	__syncthreads();
	if (lane == 0) {
		uint32_t *const cg = P.cigar + t;
		const size_t cs = (size_t)P.n;
		int n = 0; uint32_t curc = 0;
		for (int p = plen - 1; p >= 0; --p) {
			const uint32_t op = path[p];
			if (n && (curc & 0xf) == op) curc += 1u << 4;
			else {
				if (n && n <= P.max_cigar) cg[(size_t)(n - 1) * cs] = curc;
				curc = 1u << 4 | op; ++n;
			}
		}
		if (n && n <= P.max_cigar) cg[(size_t)(n - 1) * cs] = curc;
		P.n_cigar[t] = n;
		P.score[t] = score;
	}
#undef RW
#undef TBW
}

extern "C" void nabwa_launch_dp_global(const DpParams *P, hipStream_t s)
{
	if (P->n <= 0) return;
	const int small_max = getenv("NABWA_DP_SMALL") ? atoi(getenv("NABWA_DP_SMALL")) : 4096;           // tasks up to which the LDS form runs (0: never)
	const size_t lds = (size_t)P->W * DP_SMALL_LANES * 25;                                            // six rows of words + the window
	/* one wavefront per task when a task's rows, direction bytes, path, window and read fit a block's LDS (a refinement job of 150 x 155: 21 KB) */
	const size_t wlds = P->wb > 0 ? (size_t)6 * P->W * 4 + (size_t)P->H * ((P->wb + 1) / 2 + 1) + (size_t)(P->W + P->H) * 2 + 16 : 0;
	if (wlds && wlds <= 64000 && !(getenv("NABWA_DP_WAVE") && atoi(getenv("NABWA_DP_WAVE")) == 0)) {
		hipLaunchKernelGGL(dp_global_wave_kernel, dim3(P->n), dim3(64), wlds, s, *P);
		return;
	}
	if (P->n <= small_max && lds <= 60000)
		hipLaunchKernelGGL(dp_global_kernel<true>, dim3((P->n + DP_SMALL_LANES - 1) / DP_SMALL_LANES), dim3(DP_SMALL_LANES), lds, s, *P);
	else
		hipLaunchKernelGGL(dp_global_kernel<false>, dim3((P->n + 255) / 256), dim3(256), 0, s, *P);
}

// aln_local_core and aln_extend_core (forward / reverse Smith-Waterman passes): dp_wave.hip, one wavefront per task.
