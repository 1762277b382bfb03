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

struct DpParams {
	int n;
	const int64_t *ref_off, *qry_off;
	const uint8_t *ref, *qry;
	int gap_open, gap_ext, gap_end, band;
	int matrix[25];
	int W;                 // max_l1 + 1
	int H;                 // max_l2 + 1
	int32_t *rows;         // per wave: [6][W][64]
	uint8_t *tb;           // per wave: [H][W][64], byte = Mt | It<<2 | Dt<<4
	uint8_t *path;         // per wave: [W+H][64]
	int32_t *score, *n_cigar; uint32_t *cigar; int max_cigar;
};

// Two forms, as for the local kernel below: LDSV = false keeps the score rows of a wave's 64 tasks interleaved in HBM (tens of
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

// ---------------------------------------------------------------------------------------------
// Forward pass of aln_extend_core (stdaln.c:862-976): left-anchored extension with a packed row
// eh[i] = h[j-1,i-1] << 16 | e[j,i], an adaptive column window [start, end) that follows the
// positive cells, 16-bit overflow rebasing (LOCAL_OVERFLOW_*, stdaln.c:230-231) and the seed score
// G0 in eh[1].  Produces score (+of_base-1) and the end cell; the path is then filled by the global
// kernel on the two prefixes with a doubling band (stdaln.c:985-1000), driven from the host.
// ---------------------------------------------------------------------------------------------
struct ExtParams {
	int n;
	const int64_t *ref_off, *qry_off;
	const uint8_t *ref, *qry;
	const int32_t *g0;
	int gap_open, gap_ext, band;
	int matrix[25];
	int W;                 // max_l1 + 2
	uint32_t *eh;          // per wave: [W][64]
	int32_t *score, *end_i, *end_j;
};

__global__ __launch_bounds__(256) void dp_extend_fwd_kernel(const ExtParams P)
{
	const int t = blockIdx.x * 256 + threadIdx.x;
	const int lane = threadIdx.x & 63;
	const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
	if (t >= P.n) return;
	const uint8_t *s1 = P.ref + P.ref_off[t] - 1, *s2 = P.qry + P.qry_off[t] - 1;   // 1-based as in the reference
	const int l1 = (int)(P.ref_off[t + 1] - P.ref_off[t]), l2 = (int)(P.qry_off[t + 1] - P.qry_off[t]);
	P.score[t] = -1; P.end_i[t] = 0; P.end_j[t] = 0;
	if (l1 == 0 || l2 == 0) return;
	uint32_t *EH = P.eh + wave * (size_t)P.W * 64 + lane;
#define E(i) EH[(size_t)(i) * 64]
	const int r = P.gap_ext, qr = P.gap_open + P.gap_ext;
	for (int i = 0; i < l1 + 2; ++i) E(i) = 0;
	E(1) = (uint32_t)P.g0[t] << 16;
	int start = 1, end = 2, end_i = 0, end_j = 0, score = 0, is_overflow = 0, of_base = 0;
	for (int j = 1; j <= l2; ++j) {
		int h1 = 0, f = 0;
		const int *mat = P.matrix + s2[j] * 5;
		int _start = j - P.band; if (_start < 1) _start = 1;
		if (_start > start) start = _start;
		int _end = j + P.band; if (_end > l1 + 1) _end = l1 + 1;
		if (_end < end) end = _end;
		if (start == end) break;
		if (is_overflow) {
			score -= 16000; of_base += 16000; is_overflow = 0;
			for (int i = start; i <= end; ++i) {
				const uint32_t v = E(i);
				int a = (int)(v >> 16), b = (int)(v & 0xffff);
				b = b < 16000 ? 0 : b - 16000;
				a = a < 16000 ? 0 : a - 16000;
				E(i) = (uint32_t)a << 16 | (uint32_t)b;
			}
		}
		_start = _end = 0;
		for (int i = start; i < end; ++i) {
			const uint32_t v = E(i);
			int h = (int)(v >> 16), e = (int)(v & 0xffff);
			uint32_t nv = (uint32_t)h1 << 16;
			h += h ? mat[s1[i]] : 0;
			h = h > e ? h : e;
			h = h > f ? h : f;
			h1 = h;
			if (h > 0) {
				if (_start == 0) _start = i;
				_end = i;
				if (score < h) { score = h; end_i = i; end_j = j; if (score > 32000) is_overflow = 1; }
			}
			h -= qr; h = h > 0 ? h : 0;
			e -= r; e = e > h ? e : h;
			f -= r; f = f > h ? f : h;
			E(i) = nv | (uint32_t)e;
		}
		E(end) = (uint32_t)h1 << 16;
		if (_end <= 0) break;
		start = _start; end = _end + 3;
	}
	P.score[t] = score + of_base - 1; P.end_i[t] = end_i; P.end_j[t] = end_j;
#undef E
}

extern "C" void nabwa_launch_dp_extend_fwd(const ExtParams *P, hipStream_t s)
{
	if (P->n <= 0) return;
	hipLaunchKernelGGL(dp_extend_fwd_kernel, dim3((P->n + 255) / 256), dim3(256), 0, s, *P);
}

// ---------------------------------------------------------------------------------------------
// aln_local_core (stdaln.c:529-761), passes 1 and 2: forward Smith-Waterman over a packed row
// eh[i] = h << 16 | e with 16-bit overflow rebasing, giving score_f and the end cell (plus the
// per-row maxima suba[] used for the sub-optimal score); then the reverse pass from the end cell
// over an adaptive column band [end, start], giving score_r and the start cell.  The path is
// filled by the global kernel on the sub-matrix with a doubling band (stdaln.c:723-735), driven
// from the host.  Used by mate rescue (bwa_sw_core, bwape.c:456).
// ---------------------------------------------------------------------------------------------
struct LocParams {
	int n;
	const int64_t *ref_off, *qry_off;
	const uint8_t *ref, *qry;
	int gap_open, gap_ext, thres;
	int matrix[25], max_score;
	int W;                 // max_l1 + 2
	int H;                 // max_l2 + 1
	int32_t *eh;           // per wave: [W][64]
	int32_t *suba;         // per task: [H]
	int32_t *out;          // per task: score_f, score_r, start_i, start_j, end_i, end_j
};

// Two forms.  LDSV = false: 256 tasks per block, the row words of a wave's 64 tasks interleaved in HBM -- right when tens of
// thousands of tasks hide each other's round trips (100 k tasks: 43 ms).  LDSV = true: a handful of tasks (mate rescue often
// has a few per batch) would spend a memory round trip per eight cells with nothing to hide it, so eight tasks share a block
// and keep their rows and their reference windows in LDS.  Measured for two tasks of 446 x 151: 28 -> 21 ms -- what is left is
// one lane walking 67 k cells by itself, which is what dp_local_wave_kernel below takes apart (4 ms); this form remains for the
// few-task launches whose scores could reach the reference's 16-bit rebasing (reads beyond ~2900 bases).
#define LOC_SMALL_LANES 8
template <bool LDSV>
__global__ __launch_bounds__(LDSV ? LOC_SMALL_LANES : 256) void dp_local_kernel(const LocParams P)
{
	extern __shared__ int32_t loc_lds[];             // LDSV: [W][8] row words, then 8 windows of W bytes
	constexpr int BT = LDSV ? LOC_SMALL_LANES : 256, ES = LDSV ? LOC_SMALL_LANES : 64;
	const int t = blockIdx.x * BT + threadIdx.x;
	const int lane = threadIdx.x & 63;
	const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
	if (t >= P.n) return;
	const uint8_t *s1 = P.ref + P.ref_off[t] - 1, *s2 = P.qry + P.qry_off[t] - 1;   // 1-based as in the reference
	const int l1 = (int)(P.ref_off[t + 1] - P.ref_off[t]), l2 = (int)(P.qry_off[t + 1] - P.qry_off[t]);
	int32_t *out = P.out + (size_t)t * 6;
	out[0] = -1; out[1] = 0; out[2] = out[3] = out[4] = out[5] = 0;
	if (l1 == 0 || l2 == 0) return;
	int32_t *EH = LDSV ? loc_lds + threadIdx.x : P.eh + wave * (size_t)P.W * 64 + lane;
	if (LDSV) {
		uint8_t *win = (uint8_t*)(loc_lds + (size_t)P.W * LOC_SMALL_LANES) + (size_t)threadIdx.x * P.W;
		win[0] = 4;                                      // (loaded with a block of cells at the window's edge, never used)
		for (int i = 1; i <= l1; ++i) win[i] = s1[i];
		s1 = win;
	}
	int32_t *suba = P.suba + (size_t)t * P.H;
#define E(i) EH[(size_t)(i) * ES]
	const int q = P.gap_open, r = P.gap_ext, qr = q + r, qr_shift = (qr + 1) << 16, tmp_len = l1 + 1;
	int end_i = 0, end_j = 0, score_f = 0, is_overflow = 0, of_base = 0;
	for (int i = 0; i < tmp_len; ++i) E(i) = 0;
	suba[0] = 0;
	// ---- forward pass
	for (int j = 1; j <= l2; ++j) {
		int subo = 0, last_h = 0, f = 0;
		const int *mat = P.matrix + s2[j] * 5;
		if (is_overflow) {
			score_f -= 16000; of_base += 16000; is_overflow = 0;
			for (int i = 0; i < tmp_len; ++i) {
				const int v = E(i); int a = v >> 16, b = v & 0xffff;
				b = b < 16000 ? 0 : b - 16000; a = a < 16000 ? 0 : a - 16000;
				E(i) = a << 16 | b;
			}
		}
		const int mt0 = mat[0], mt1 = mat[1], mt2 = mat[2], mt3 = mat[3], mt4 = mat[4];      // this row's scores against A, C, G, T, N
		int sv = E(0);                                   // *s with s = &eh[i-1]
		// LB cells at a time: their row words and bases are fetched together (a cell reads eh[i] before any cell of this row has
		// written it -- the writes go to eh[i-1] -- so the values are the ones the one-by-one loop would have seen)
#define LB 8
		for (int i0 = 1; i0 < tmp_len; i0 += LB) {
			int us[LB], ub[LB];
#pragma unroll
			for (int u = 0; u < LB; ++u) { us[u] = 0; ub[u] = 4; if (i0 + u < tmp_len) { us[u] = E(i0 + u); ub[u] = s1[i0 + u]; } }
#pragma unroll
			for (int u = 0; u < LB; ++u) {
				const int i = i0 + u;
				if (i >= tmp_len) break;
				const int sn = us[u], b = ub[u];                 // *(s+1)
				int curr_h = (sv >> 16) + (b == 0 ? mt0 : (b == 1 ? mt1 : (b == 2 ? mt2 : (b == 3 ? mt3 : mt4))));
				if (curr_h < 0) curr_h = 0;
				if (last_h > 0) { f = (f > last_h - q) ? f - r : last_h - qr; if (curr_h < f) curr_h = f; }
				if (sn >= qr_shift) {
					const int curr_last_h = sn >> 16;
					const int e = ((sv & 0xffff) > curr_last_h - q) ? (sv & 0xffff) - r : curr_last_h - qr;
					if (curr_h < e) curr_h = e;
					E(i - 1) = last_h << 16 | e;
				} else E(i - 1) = last_h << 16;
				last_h = curr_h;
				if (subo < curr_h) subo = curr_h;
				if (score_f < curr_h) { score_f = curr_h; end_i = i; end_j = j; if (score_f > 32000) is_overflow = 1; }
				sv = sn;
			}
		}
#undef LB
		E(l1) = last_h << 16;
		suba[j] = subo + of_base;
	}
	score_f += of_base;
	out[0] = score_f; out[4] = end_i; out[5] = end_j;
	if (score_f < P.thres || end_i == 0 || end_j == 0) return;
	// ---- reverse pass
	for (int i = end_i; i >= 0; --i) E(i) = 0;
	int score_r = P.matrix[s1[end_i] * 5 + s2[end_j]];
	is_overflow = of_base = 0;
	int start_i = end_i, start_j = end_j;
	E(end_i) = (qr + score_r) << 16;
	int start = end_i - 1, end = end_i - 3;
	if (end <= 0) end = 0;
	for (int j = end_j - 1; j != 0; --j) {
		int last_h = 0, f = 0;
		const int *mat = P.matrix + s2[j] * 5;
		if (is_overflow) {
			score_r -= 16000; of_base += 16000; is_overflow = 0;
			for (int i = start; i >= end; --i) {
				const int v = E(i + 1); int a = v >> 16, b = v & 0xffff;
				b = b < 16000 ? 0 : b - 16000; a = a < 16000 ? 0 : a - 16000;
				E(i + 1) = a << 16 | b;
			}
		}
		const int mt0 = mat[0], mt1 = mat[1], mt2 = mat[2], mt3 = mat[3], mt4 = mat[4];
		int i = start;
		// the same in blocks, downwards: cell i reads eh[i+1] and eh[i] and writes eh[i+1]; eh[i] is the next cell's eh[i+1], carried
		bool found = false;
		int sv = (i != end && i >= 0) ? E(i + 1) : 0;   // *s with s = &eh[i+1]
#define LB 8
		while (i != end && i >= 0 && !found) {
			int us[LB], ub[LB];
#pragma unroll
			for (int u = 0; u < LB; ++u) { us[u] = 0; ub[u] = 4; const int ii = i - u; if (ii >= 0 && (u == 0 || ii > end || end > i)) { us[u] = E(ii); ub[u] = s1[ii]; } }
#pragma unroll
			for (int u = 0; u < LB; ++u) {
				if (i == end || i < 0) break;
				const int sp = us[u], b = ub[u];         // *(s-1)
				int curr_h = (sv >> 16) + (b == 0 ? mt0 : (b == 1 ? mt1 : (b == 2 ? mt2 : (b == 3 ? mt3 : mt4))));
				if (curr_h < 0) curr_h = 0;
				if (last_h > 0) { f = (f > last_h - q) ? f - r : last_h - qr; if (curr_h < f) curr_h = f; }
				const int curr_last_h = sp >> 16;
				int e = ((sv & 0xffff) > curr_last_h - q) ? (sv & 0xffff) - r : curr_last_h - qr;
				if (e < 0) e = 0;
				if (curr_h < e) curr_h = e;
				E(i + 1) = last_h << 16 | e;
				last_h = curr_h;
				if (score_r < curr_h) {
					score_r = curr_h; start_i = i; start_j = j;
					if (score_r + of_base - qr == score_f) { j = 1; found = true; break; }
					if (score_r > 32000) is_overflow = 1;
				}
				sv = sp;
				--i;
			}
		}
#undef LB
		E(i + 1) = last_h << 16;                         // on the break, s was not advanced: same cell
		if ((E(start) >> 16) <= qr) --start;
		if (start <= 0) start = 0;
		end = start_i - (start_j - j) - (score_r + of_base + (start_j - j) * P.max_score) / r - 1;
		if (end <= 0) end = 0;
	}
	out[1] = score_r + of_base - qr; out[2] = start_i; out[3] = start_j;
#undef E
}

// The same for a FEW tasks, one WAVE per task.  A lane by itself walks the l1 x l2 cells of the forward pass one after the
// other (21 ms for 446 x 151 whatever the memory: one lane issues one instruction at a time); here the 64 lanes take 64
// consecutive rows and move along the anti-diagonals -- cell (j, i) needs h(j-1, i-1), h(j-1, i) and the vertical-gap value
// e(j-1, i), all of which the lane above produced one and two steps earlier and hands down by a lane shift -- strip after strip
// of 64 rows, the last row of a strip left in LDS for the first row of the next.  In terms of the reference's packed row
// (stdaln.c:579-640): eh[i-1] of the previous row is h(j-1, i-1) << 16 | e(j-1, i), eh[i] is h(j-1, i) << 16 | e(j-1, i+1), and the
// test `eh[i] >= (q + r + 1) << 16` is h(j-1, i) >= q + r + 1.  Row-major "first cell that reaches the best score" = the smallest
// row, then the smallest column, among the cells with the best score: a reduction over the lanes.  Only where the 16-bit
// rebasing of the reference cannot trigger (l2 * max_score <= 32000; the launch checks it).  The reverse pass -- a narrow,
// data-dependent band from the end cell -- is lane 0's, on a row in LDS.
__global__ __launch_bounds__(64) void dp_local_wave_kernel(const LocParams P)
{
	extern __shared__ int32_t wav_lds[];             // [W] h and [W] e of the row above the strip, [W] the reverse pass's row, W window bytes
	const int t = blockIdx.x, lane = threadIdx.x;
	if (t >= P.n) return;
	const uint8_t *s1g = P.ref + P.ref_off[t] - 1, *s2 = P.qry + P.qry_off[t] - 1;   // 1-based as in the reference
	const int l1 = (int)(P.ref_off[t + 1] - P.ref_off[t]), l2 = (int)(P.qry_off[t + 1] - P.qry_off[t]);
	int32_t *out = P.out + (size_t)t * 6;
	if (lane == 0) { out[0] = -1; out[1] = 0; out[2] = out[3] = out[4] = out[5] = 0; }
	if (l1 == 0 || l2 == 0) return;
	const int W = P.W;
	int32_t *bh = wav_lds, *be = wav_lds + W, *rev = wav_lds + 2 * (size_t)W;
	uint8_t *win = (uint8_t*)(wav_lds + 3 * (size_t)W);
	for (int i = lane; i <= l1; i += 64) { bh[i] = 0; be[i] = 0; win[i] = i ? s1g[i] : (uint8_t)4; }
	__syncthreads();
	const uint8_t *s1 = win;
	int32_t *suba = P.suba + (size_t)t * P.H;
	const int q = P.gap_open, r = P.gap_ext, qr = q + r;
	if (lane == 0) suba[0] = 0;
	// ---- forward pass
	int best = 0, best_i = 0, best_j = 0;            // this lane's rows: the first cell with their best score
	for (int j0 = 1; j0 <= l2; j0 += 64) {
		const int j = j0 + lane;
		const bool live = j <= l2;
		const int *mat = P.matrix + (live ? s2[j] : 4) * 5;
		const int mt0 = mat[0], mt1 = mat[1], mt2 = mat[2], mt3 = mat[3], mt4 = mat[4];
		const bool writes = live && (lane == 63 || j == l2);      // the strip's last row: it is "the row above" for the next strip
		int last_h = 0, f = 0, subo = 0, diag = 0, o_h = 0, o_e = 0;
		for (int d = 0; d < l1 + 63; ++d) {
			const int i = d - lane + 1;
			int uh = __shfl_up(o_h, 1), ue = __shfl_up(o_e, 1);          // h(j-1, i), e(j-1, i): the lane above's last step
			const bool act = live && i >= 1 && i <= l1;
			if (lane == 0 && act) { uh = bh[i]; ue = be[i]; }
			if (act) {
				const int b = s1[i];
				int curr_h = diag + (b == 0 ? mt0 : (b == 1 ? mt1 : (b == 2 ? mt2 : (b == 3 ? mt3 : mt4))));
				if (curr_h < 0) curr_h = 0;
				if (last_h > 0) { f = (f > last_h - q) ? f - r : last_h - qr; if (curr_h < f) curr_h = f; }
				int e = 0;
				if (uh >= qr + 1) { e = (ue > uh - q) ? ue - r : uh - qr; if (curr_h < e) curr_h = e; }
				last_h = curr_h;
				if (subo < curr_h) subo = curr_h;
				if (best < curr_h) { best = curr_h; best_i = i; best_j = j; }
				diag = uh; o_h = curr_h; o_e = e;
				if (writes) { bh[i] = curr_h; be[i] = e; }
			}
		}
		if (live) suba[j] = subo;
	}
	// the first cell, in row-major order, with the best score of all
	int score_f = best, end_i = best_i, end_j = best_j;
	for (int o = 32; o; o >>= 1) {
		const int s_o = __shfl_xor(score_f, o), i_o = __shfl_xor(end_i, o), j_o = __shfl_xor(end_j, o);
		if (s_o > score_f || (s_o == score_f && (j_o < end_j || (j_o == end_j && i_o < end_i)))) { score_f = s_o; end_i = i_o; end_j = j_o; }
	}
	if (score_f == 0) { end_i = 0; end_j = 0; }      // no cell ever raised the running best above its start
	if (lane != 0) return;
	out[0] = score_f; out[4] = end_i; out[5] = end_j;
	if (score_f < P.thres || end_i == 0 || end_j == 0) return;
	// ---- reverse pass (lane 0; no rebasing can trigger here either)
#define E(i) rev[(i)]
	for (int i = end_i; i >= 0; --i) E(i) = 0;
	int score_r = P.matrix[s1[end_i] * 5 + s2[end_j]];
	int start_i = end_i, start_j = end_j;
	E(end_i) = (qr + score_r) << 16;
	int start = end_i - 1, end = end_i - 3;
	if (end <= 0) end = 0;
	for (int j = end_j - 1; j != 0; --j) {
		int last_h = 0, f = 0;
		const int *mat = P.matrix + s2[j] * 5;
		int i;
		bool found = false;
		for (i = start; i != end && i >= 0; --i) {
			const int sv = E(i + 1), sp = E(i);
			const int b = s1[i];
			int curr_h = (sv >> 16) + mat[b > 4 ? 4 : b];
			if (curr_h < 0) curr_h = 0;
			if (last_h > 0) { f = (f > last_h - q) ? f - r : last_h - qr; if (curr_h < f) curr_h = f; }
			const int curr_last_h = sp >> 16;
			int e = ((sv & 0xffff) > curr_last_h - q) ? (sv & 0xffff) - r : curr_last_h - qr;
			if (e < 0) e = 0;
			if (curr_h < e) curr_h = e;
			E(i + 1) = last_h << 16 | e;
			last_h = curr_h;
			if (score_r < curr_h) {
				score_r = curr_h; start_i = i; start_j = j;
				if (score_r - qr == score_f) { j = 1; found = true; break; }
			}
		}
		(void)found;
		E(i + 1) = last_h << 16;                         // on the break, the cell was not left: same cell
		if ((E(start) >> 16) <= qr) --start;
		if (start <= 0) start = 0;
		end = start_i - (start_j - j) - (score_r + (start_j - j) * P.max_score) / r - 1;
		if (end <= 0) end = 0;
	}
	out[1] = score_r - qr; out[2] = start_i; out[3] = start_j;
#undef E
}

extern "C" void nabwa_launch_dp_local(const LocParams *P, hipStream_t s)
{
	if (P->n <= 0) return;
	const int small_max = getenv("NABWA_DP_SMALL") ? atoi(getenv("NABWA_DP_SMALL")) : 4096;     // tasks up to which the LDS form runs (0: never)
	const size_t lds = (size_t)P->W * LOC_SMALL_LANES * 5;                                            // row words + windows
	int max_score = 0;
	for (int k = 0; k < 25; ++k) if (P->matrix[k] > max_score) max_score = P->matrix[k];
	if (P->n <= small_max && (long long)P->H * max_score <= 32000 && (size_t)P->W * 13 + 16 <= 60000 && !getenv("NABWA_DP_NO_WAVE"))
		hipLaunchKernelGGL(dp_local_wave_kernel, dim3(P->n), dim3(64), (size_t)P->W * 13 + 16, s, *P);         // one wave per task
	else if (P->n <= small_max && lds <= 60000)
		hipLaunchKernelGGL(dp_local_kernel<true>, dim3((P->n + LOC_SMALL_LANES - 1) / LOC_SMALL_LANES), dim3(LOC_SMALL_LANES), lds, s, *P);
	else
		hipLaunchKernelGGL(dp_local_kernel<false>, dim3((P->n + 255) / 256), dim3(256), 0, s, *P);
}
