// se_finish.hip -- host side of the single-end finishing chain + the batched global-alignment entry.
//
// What the reference does per record between the FM search and the BAM record (bam2bam.c:622-657):
//   posn_singleton : bwa_aln2seq_core (bwase.c:19-95, consumes drand48 in record order),
//                    bwa_cal_pac_pos_core + multi-hit positions (bwase.c:139-154, bam2bam.c:633-637)
//   finish_singleton: bwa_refine_gapped (bwase.c:356-423): refine_gapped_core -> aln_global_core,
//                    bwa_cal_md1, bwa_correct_trimmed; then the flag / XT logic of bwa_update_bam1
//                    (bam2bam.c:430-593), which mirrors bwa_print_sam1 (bwase.c:458-571).
// Here the same chain runs over a whole batch: phase 1 on the host in record order (it is O(ns) per read
// and owns the RNG stream), all bwt_sa walks as ONE GPU batch, all gap refinements as ONE GPU batch of
// banded global alignments, the rest (CIGAR fix-ups, MD/NM, flags) on the host.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stddef.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <chrono>
#include <functional>
#include <mutex>
#include <thread>
#include "../../include/nabwa.h"
#include "nabwa_internal.hpp"
#include "finish_common.hpp"

#include "dp_params.hpp"


#define SCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
	char b_[512]; snprintf(b_, sizeof b_, "%s failed: %s (%s:%d)", #x, hipGetErrorString(e_), __FILE__, __LINE__); \
	return nabwa_fail(NABWA_ENODEV, "%s", b_); } } while (0)

/* Working memory of the alignment entry points: one grow-only block per device, kept between calls (hipMalloc / hipFree of the
 * traceback matrices -- 1.5 GB for 64 k pairs of 150 bases -- cost more than the kernels), handed out under a lock for the length
 * of one launch.  nabwa_dp_scratch_release gives it back. */
namespace {
struct DevArena { std::mutex mu; void *base = 0; size_t cap = 0; };
DevArena g_arena[16];
struct ArenaUse {
	DevArena &A; std::unique_lock<std::mutex> lk; size_t used = 0;
	explicit ArenaUse(int dev) : A(g_arena[dev & 15]), lk(A.mu) {}
	hipError_t reserve(size_t bytes)
	{
		used = 0;
		if (bytes <= A.cap) return hipSuccess;
		if (A.base) { (void)hipFree(A.base); A.base = 0; A.cap = 0; }
		const size_t c = bytes + bytes / 4;
		const hipError_t e = hipMalloc(&A.base, c);
		if (e == hipSuccess) A.cap = c;
		return e;
	}
	template <class T> T *take(size_t bytes) { T *p = (T*)((char*)A.base + used); used += (bytes + 255) & ~(size_t)255; return p; }
};
inline size_t up256(size_t b) { return (b + 255) & ~(size_t)255; }
}

extern "C" void nabwa_dp_scratch_release(int device)
{
	DevArena &A = g_arena[device & 15];
	std::lock_guard<std::mutex> lk(A.mu);
	if (A.base) { (void)hipFree(A.base); A.base = 0; A.cap = 0; }
}

/* ------------------------------------------------------------------ batched aln_global_core */

extern "C" int nabwa_global_align(int device, int n, const int64_t *ref_off, const uint8_t *ref, const int64_t *qry_off,
								  const uint8_t *qry, int gap_open, int gap_ext, int gap_end, const int *matrix25, int band,
								  int32_t *score, int32_t *n_cigar, uint32_t *cigar32, int max_cigar)
{
	if (n < 0 || (n && (!ref_off || !qry_off || !ref || !qry || !matrix25 || !score || !n_cigar || !cigar32)) || max_cigar < 1)
		return nabwa_fail(NABWA_EINVAL, "bad argument");
	if (n == 0) return NABWA_OK;
	if (nabwa_device_count() <= device) return nabwa_fail(NABWA_ENODEV, "no such HIP device");
	SCHK(hipSetDevice(device));
	const int CHUNK = 1 << 16;                     // tasks per launch: bounds the traceback scratch
	const bool timing = getenv("NABWA_TIMING") != 0 && n >= 1024;
	auto now = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
	double tg[4] = { 0, 0, 0, 0 };              // set-up, upload, kernel, download
	for (int c0 = 0; c0 < n; c0 += CHUNK) {
		const double tg0 = now();
		const int m = std::min(CHUNK, n - c0);
		int W = 1, H = 1; int64_t maxdiff = 0;
		for (int i = c0; i < c0 + m; ++i) {
			const int64_t a1 = ref_off[i + 1] - ref_off[i], a2 = qry_off[i + 1] - qry_off[i];
			W = std::max<int64_t>(W, a1 + 1);
			H = std::max<int64_t>(H, a2 + 1);
			maxdiff = std::max<int64_t>(maxdiff, a1 > a2 ? a1 - a2 : a2 - a1);
		}
		std::vector<int64_t> ro(m + 1), qo(m + 1);
		for (int i = 0; i <= m; ++i) { ro[i] = ref_off[c0 + i] - ref_off[c0]; qo[i] = qry_off[c0 + i] - qry_off[c0]; }
		const size_t waves = (size_t)((m + 255) / 256) * 4;
		DpParams P; memset(&P, 0, sizeof(P));
		ArenaUse A(device);
		const size_t sz[10] = { (size_t)(m + 1) * 8, (size_t)(m + 1) * 8, (size_t)ro[m] + 16, (size_t)qo[m] + 16, waves * 6 * (size_t)W * 64 * 4,
								waves * (size_t)H * W * 64, waves * (size_t)(W + H) * 64, (size_t)m * 4, (size_t)m * 4, (size_t)m * max_cigar * 4 };
		size_t need = 0; for (size_t z : sz) need += up256(z);
		SCHK(A.reserve(need));
		int64_t *d_ro = A.take<int64_t>(sz[0]), *d_qo = A.take<int64_t>(sz[1]); uint8_t *d_ref = A.take<uint8_t>(sz[2]), *d_qry = A.take<uint8_t>(sz[3]);
		P.rows = A.take<int32_t>(sz[4]); P.tb = A.take<uint8_t>(sz[5]); P.path = A.take<uint8_t>(sz[6]);
		P.score = A.take<int32_t>(sz[7]); P.n_cigar = A.take<int32_t>(sz[8]); P.cigar = A.take<uint32_t>(sz[9]);
		const double tg1 = now();
		SCHK(hipMemcpy(d_ro, ro.data(), (m + 1) * 8, hipMemcpyHostToDevice));
		SCHK(hipMemcpy(d_qo, qo.data(), (m + 1) * 8, hipMemcpyHostToDevice));
		if (ro[m]) SCHK(hipMemcpy(d_ref, ref + ref_off[c0], ro[m], hipMemcpyHostToDevice));
		if (qo[m]) SCHK(hipMemcpy(d_qry, qry + qry_off[c0], qo[m], hipMemcpyHostToDevice));
		P.n = m; P.ref_off = d_ro; P.qry_off = d_qo; P.ref = d_ref; P.qry = d_qry;
		P.gap_open = gap_open; P.gap_ext = gap_ext; P.gap_end = gap_end; P.band = band;
		memcpy(P.matrix, matrix25, sizeof(P.matrix));
		P.W = W; P.H = H; P.max_cigar = max_cigar;
		P.wb = (int)std::min<int64_t>(W, 2 * (int64_t)band + maxdiff + 1);
		const double tg2 = now();
		nabwa_launch_dp_global(&P, 0);
		SCHK(hipGetLastError());
		if (timing) SCHK(hipDeviceSynchronize());
		const double tg3 = now();
		SCHK(hipMemcpy(score + c0, P.score, (size_t)m * 4, hipMemcpyDeviceToHost));
		SCHK(hipMemcpy(n_cigar + c0, P.n_cigar, (size_t)m * 4, hipMemcpyDeviceToHost));
		/* the operations come slot by slot (dp_global_kernel): only the slots in use travel */
		int slots = 0;
		for (int i = 0; i < m; ++i) slots = std::max(slots, std::min(n_cigar[c0 + i], max_cigar));
		if (slots) {
			std::vector<uint32_t> cs((size_t)slots * m);
			SCHK(hipMemcpy(cs.data(), P.cigar, cs.size() * 4, hipMemcpyDeviceToHost));
			for (int i = 0; i < m; ++i) {
				uint32_t *dst = cigar32 + (size_t)(c0 + i) * max_cigar;
				const int k_n = std::min(n_cigar[c0 + i], max_cigar);
				for (int k = 0; k < k_n; ++k) dst[k] = cs[(size_t)k * m + i];
			}
		}
		tg[0] += tg1 - tg0; tg[1] += tg2 - tg1; tg[2] += tg3 - tg2; tg[3] += now() - tg3;
	}
	if (timing) fprintf(stderr, "[nabwa] global_align %d tasks: set-up %.4f s, upload %.4f s, kernel %.4f s, download %.4f s\n", n, tg[0], tg[1], tg[2], tg[3]);
	return NABWA_OK;
}

/* ------------------------------------------------------------------ batched aln_extend_core */



extern "C" int nabwa_extend_align(int device, int n, const int64_t *ref_off, const uint8_t *ref, const int64_t *qry_off,
								  const uint8_t *qry, int gap_open, int gap_ext, const int *matrix25, int band, const int32_t *G0,
								  int32_t *score, int32_t *n_cigar, uint32_t *cigar32, int max_cigar)
{
	if (n < 0 || (n && (!ref_off || !qry_off || !ref || !qry || !matrix25 || !G0 || !score || !n_cigar || !cigar32)) || max_cigar < 1 || band < 1)
		return nabwa_fail(NABWA_EINVAL, "bad argument");
	if (n == 0) return NABWA_OK;
	if (nabwa_device_count() <= device) return nabwa_fail(NABWA_ENODEV, "no such HIP device");
	SCHK(hipSetDevice(device));
	/* forward pass on the GPU */
	int W = 2;
	for (int i = 0; i < n; ++i) W = std::max<int64_t>(W, ref_off[i + 1] - ref_off[i] + 2);
	std::vector<int32_t> fs(n), ei(n), ej(n);
	{
		ExtParams P; memset(&P, 0, sizeof(P));
		ArenaUse A(device);
		const size_t sz[9] = { (size_t)(n + 1) * 8, (size_t)(n + 1) * 8, (size_t)ref_off[n] + 16, (size_t)qry_off[n] + 16, (size_t)n * 4,
							   nabwa_dp_local_fits_lds(W) ? 256 : (size_t)n * nabwa_dp_local_rows_bytes(W), (size_t)n * 4, (size_t)n * 4, (size_t)n * 4 };
		size_t need = 0; for (size_t z : sz) need += up256(z);
		SCHK(A.reserve(need));
		int64_t *d_ro = A.take<int64_t>(sz[0]), *d_qo = A.take<int64_t>(sz[1]); uint8_t *d_ref = A.take<uint8_t>(sz[2]), *d_qry = A.take<uint8_t>(sz[3]);
		int32_t *d_g0 = A.take<int32_t>(sz[4]);
		P.eh = A.take<uint32_t>(sz[5]); P.score = A.take<int32_t>(sz[6]); P.end_i = A.take<int32_t>(sz[7]); P.end_j = A.take<int32_t>(sz[8]);
		SCHK(hipMemcpy(d_ro, ref_off, (size_t)(n + 1) * 8, hipMemcpyHostToDevice));
		SCHK(hipMemcpy(d_qo, qry_off, (size_t)(n + 1) * 8, hipMemcpyHostToDevice));
		if (ref_off[n]) SCHK(hipMemcpy(d_ref, ref, ref_off[n], hipMemcpyHostToDevice));
		if (qry_off[n]) SCHK(hipMemcpy(d_qry, qry, qry_off[n], hipMemcpyHostToDevice));
		SCHK(hipMemcpy(d_g0, G0, (size_t)n * 4, hipMemcpyHostToDevice));
		P.n = n; P.ref_off = d_ro; P.qry_off = d_qo; P.ref = d_ref; P.qry = d_qry; P.g0 = d_g0;
		P.gap_open = gap_open; P.gap_ext = gap_ext; P.band = band; memcpy(P.matrix, matrix25, sizeof(P.matrix)); P.W = W;
		nabwa_launch_dp_extend_fwd(&P, 0);
		SCHK(hipGetLastError());
		SCHK(hipMemcpy(fs.data(), P.score, (size_t)n * 4, hipMemcpyDeviceToHost));
		SCHK(hipMemcpy(ei.data(), P.end_i, (size_t)n * 4, hipMemcpyDeviceToHost));
		SCHK(hipMemcpy(ej.data(), P.end_j, (size_t)n * 4, hipMemcpyDeviceToHost));
	}
	/* path: global alignment of the two prefixes with gap_end = -1 and a doubling band (stdaln.c:985-1000) */
	std::vector<int> act;
	for (int i = 0; i < n; ++i) { score[i] = fs[i]; n_cigar[i] = 0; if (fs[i] > 0) act.push_back(i); }
	for (int bw = band; !act.empty(); bw <<= 1) {
		std::vector<int64_t> ro(act.size() + 1, 0), qo(act.size() + 1, 0); std::vector<uint8_t> rb, qb;
		for (size_t t = 0; t < act.size(); ++t) {
			const int i = act[t];
			rb.insert(rb.end(), ref + ref_off[i], ref + ref_off[i] + ei[i]);
			qb.insert(qb.end(), qry + qry_off[i], qry + qry_off[i] + ej[i]);
			ro[t + 1] = (int64_t)rb.size(); qo[t + 1] = (int64_t)qb.size();
		}
		rb.push_back(0); qb.push_back(0);
		std::vector<int32_t> sg(act.size()), nc(act.size()); std::vector<uint32_t> cg(act.size() * (size_t)max_cigar);
		int r = nabwa_global_align(device, (int)act.size(), ro.data(), rb.data(), qo.data(), qb.data(), gap_open, gap_ext, -1,
								   matrix25, bw, sg.data(), nc.data(), cg.data(), max_cigar);
		if (r != NABWA_OK) return r;
		std::vector<int> next;
		for (size_t t = 0; t < act.size(); ++t) {
			const int i = act[t], jmax = std::max(ei[i], ej[i]);
			if (sg[t] == fs[i] || bw > jmax) {
				score[i] = sg[t]; n_cigar[i] = nc[t];
				memcpy(cigar32 + (size_t)i * max_cigar, cg.data() + t * (size_t)max_cigar, (size_t)std::min(nc[t], max_cigar) * 4);
			} else next.push_back(i);
		}
		act.swap(next);
	}
	return NABWA_OK;
}

/* ------------------------------------------------------------------ batched aln_local_core */



extern "C" int nabwa_local_align(int device, int n, const int64_t *ref_off, const uint8_t *ref, const int64_t *qry_off,
								 const uint8_t *qry, int gap_open, int gap_ext, const int *matrix25, int band, int thres,
								 int32_t *score, int32_t *coords /* n x 4: start_i,start_j,end_i,end_j (1-based) */, int32_t *subo,
								 int32_t *n_cigar, uint32_t *cigar32, int max_cigar)
{
	if (n < 0 || (n && (!ref_off || !qry_off || !ref || !qry || !matrix25 || !score || !coords || !n_cigar || !cigar32)) || max_cigar < 1 || band < 1 || thres < 1)
		return nabwa_fail(NABWA_EINVAL, "bad argument");
	if (n == 0) return NABWA_OK;
	if (nabwa_device_count() <= device) return nabwa_fail(NABWA_ENODEV, "no such HIP device");
	SCHK(hipSetDevice(device));
	auto now = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
	const bool timing = getenv("NABWA_TIMING") != 0;
	const double tl0 = now();
	double tl1 = 0, tl2 = 0, tl3 = 0;
	int W = 2, H = 2, max_score = 0;
	for (int i = 0; i < n; ++i) { W = std::max<int64_t>(W, ref_off[i + 1] - ref_off[i] + 2); H = std::max<int64_t>(H, qry_off[i + 1] - qry_off[i] + 1); }
	for (int i = 0; i < 25; ++i) max_score = std::max(max_score, matrix25[i]);
	std::vector<int32_t> o((size_t)n * 6), sub((size_t)n * H);
	{
		LocParams P; memset(&P, 0, sizeof(P));
		ArenaUse A(device);
		const size_t sz[7] = { (size_t)(n + 1) * 8, (size_t)(n + 1) * 8, (size_t)ref_off[n] + 16, (size_t)qry_off[n] + 16,
							   nabwa_dp_local_fits_lds(W) ? 256 : (size_t)n * nabwa_dp_local_rows_bytes(W), (size_t)n * H * 4, (size_t)n * 24 };
		size_t need = 0; for (size_t z : sz) need += up256(z);
		SCHK(A.reserve(need));
		int64_t *d_ro = A.take<int64_t>(sz[0]), *d_qo = A.take<int64_t>(sz[1]); uint8_t *d_ref = A.take<uint8_t>(sz[2]), *d_qry = A.take<uint8_t>(sz[3]);
		P.eh = A.take<int32_t>(sz[4]); P.suba = A.take<int32_t>(sz[5]); P.out = A.take<int32_t>(sz[6]);
		SCHK(hipMemcpy(d_ro, ref_off, (size_t)(n + 1) * 8, hipMemcpyHostToDevice));
		SCHK(hipMemcpy(d_qo, qry_off, (size_t)(n + 1) * 8, hipMemcpyHostToDevice));
		if (ref_off[n]) SCHK(hipMemcpy(d_ref, ref, ref_off[n], hipMemcpyHostToDevice));
		if (qry_off[n]) SCHK(hipMemcpy(d_qry, qry, qry_off[n], hipMemcpyHostToDevice));
		P.n = n; P.ref_off = d_ro; P.qry_off = d_qo; P.ref = d_ref; P.qry = d_qry;
		P.gap_open = gap_open; P.gap_ext = gap_ext; P.thres = thres; memcpy(P.matrix, matrix25, 100); P.max_score = max_score; P.W = W; P.H = H;
		P.row_forward = getenv("NABWA_DP_FORWARD") && !strcmp(getenv("NABWA_DP_FORWARD"), "rows");
		tl1 = now();
		nabwa_launch_dp_local(&P, 0);
		SCHK(hipGetLastError());
		if (timing) { SCHK(hipDeviceSynchronize()); tl2 = now(); }
		SCHK(hipMemcpy(o.data(), P.out, (size_t)n * 24, hipMemcpyDeviceToHost));
		SCHK(hipMemcpy(sub.data(), P.suba, (size_t)n * H * 4, hipMemcpyDeviceToHost));
	}
	tl3 = now();
	std::vector<int> act;
	for (int i = 0; i < n; ++i) {
		const int32_t *v = &o[(size_t)i * 6];
		const int l2 = (int)(qry_off[i + 1] - qry_off[i]);
		score[i] = v[0]; n_cigar[i] = 0;
		coords[4 * i] = v[2]; coords[4 * i + 1] = v[3]; coords[4 * i + 2] = v[4]; coords[4 * i + 3] = v[5];
		if (subo) subo[i] = 0;
		if (l2 == 0 || ref_off[i + 1] == ref_off[i]) { score[i] = -1; continue; }
		if (v[0] < thres || v[4] == 0 || v[5] == 0) continue;
		if (subo) {                                             /* stdaln.c:700-709 */
			int tmp2 = 0, tmp = (int)(v[3] - .33 * (v[5] - v[3]) + .499);
			const int32_t *sa = &sub[(size_t)i * H];
			for (int j = 1; j <= tmp; ++j) if (tmp2 < sa[j]) tmp2 = sa[j];
			tmp = (int)(v[5] + .33 * (v[5] - v[3]) + .499);
			for (int j = tmp; j <= l2; ++j) if (tmp2 < sa[j]) tmp2 = sa[j];
			subo[i] = tmp2;
		}
		act.push_back(i);
	}
	/* path: global alignment of the sub-matrix, gap_end = -1, doubling band (stdaln.c:723-735) */
	for (int bw = band; !act.empty(); bw <<= 1) {
		std::vector<int64_t> ro(act.size() + 1, 0), qo(act.size() + 1, 0); std::vector<uint8_t> rb, qb;
		for (size_t t = 0; t < act.size(); ++t) {
			const int i = act[t]; const int32_t *v = &o[(size_t)i * 6];
			rb.insert(rb.end(), ref + ref_off[i] + v[2] - 1, ref + ref_off[i] + v[4]);
			qb.insert(qb.end(), qry + qry_off[i] + v[3] - 1, qry + qry_off[i] + v[5]);
			ro[t + 1] = (int64_t)rb.size(); qo[t + 1] = (int64_t)qb.size();
		}
		rb.push_back(0); qb.push_back(0);
		std::vector<int32_t> sg(act.size()), nc(act.size()); std::vector<uint32_t> cg(act.size() * (size_t)max_cigar);
		int r = nabwa_global_align(device, (int)act.size(), ro.data(), rb.data(), qo.data(), qb.data(), gap_open, gap_ext, -1,
								   matrix25, bw, sg.data(), nc.data(), cg.data(), max_cigar);
		if (r != NABWA_OK) return r;
		std::vector<int> next;
		for (size_t t = 0; t < act.size(); ++t) {
			const int i = act[t]; const int32_t *v = &o[(size_t)i * 6];
			const int jmax = std::max(v[4] - v[2], v[5] - v[3]) + 1;
			if (sg[t] == v[1] || sg[t] == v[0] || bw > jmax) {
				score[i] = (v[1] > sg[t] && v[0] > sg[t]) ? -1 : sg[t];     /* "potential bug" branch, stdaln.c:737-740 */
				n_cigar[i] = nc[t];
				memcpy(cigar32 + (size_t)i * max_cigar, cg.data() + t * (size_t)max_cigar, (size_t)std::min(nc[t], max_cigar) * 4);
			} else next.push_back(i);
		}
		act.swap(next);
	}
	if (timing) fprintf(stderr, "[nabwa] local_align %d tasks (window %d x %d): set-up + upload %.4f s, kernel %.4f s, download %.4f s, paths (global alignments) %.4f s\n", n, W, H, tl1 - tl0, tl2 - tl1, tl3 - tl2, now() - tl3);
	return NABWA_OK;
}

/* ------------------------------------------------------------------ reference annotations */

extern "C" int nabwa_index_attach_reference(nabwa_index_t *ix, const char *prefix)
{
	if (!ix || !prefix) return nabwa_fail(NABWA_EINVAL, "null argument");
	nabwa_reference *R = new nabwa_reference();
	std::string p(prefix);
	FILE *f = fopen((p + ".ann").c_str(), "r");                       /* format: bns_dump / bns_restore_core, bntseq.c:63-117 */
	long long xx; int n_seqs = 0; unsigned seed = 0;
	if (!f || fscanf(f, "%lld%d%u", &xx, &n_seqs, &seed) != 3) { if (f) fclose(f); delete R; return nabwa_fail(NABWA_EIO, "cannot read %s.ann", prefix); }
	R->l_pac = xx; R->seed = seed;
	for (int i = 0; i < n_seqs; ++i) {
		unsigned gi; char name[1024]; int c; nabwa_ann a;
		if (fscanf(f, "%u%1023s", &gi, name) != 2) { fclose(f); delete R; return nabwa_fail(NABWA_EIO, "malformed %s.ann", prefix); }
		while ((c = fgetc(f)) != '\n' && c != EOF) {}
		if (fscanf(f, "%lld%d%d", &xx, &a.len, &a.n_ambs) != 3) { fclose(f); delete R; return nabwa_fail(NABWA_EIO, "malformed %s.ann", prefix); }
		a.offset = xx; a.name = name;
		R->anns.push_back(a);
	}
	fclose(f);
	f = fopen((p + ".amb").c_str(), "r");
	int ns = 0, nh = 0;
	if (!f || fscanf(f, "%lld%d%d", &xx, &ns, &nh) != 3) { if (f) fclose(f); delete R; return nabwa_fail(NABWA_EIO, "cannot read %s.amb", prefix); }
	for (int i = 0; i < nh; ++i) {
		char s[64]; nabwa_hole h;
		if (fscanf(f, "%lld%d%63s", &xx, &h.len, s) != 3) { fclose(f); delete R; return nabwa_fail(NABWA_EIO, "malformed %s.amb", prefix); }
		h.offset = xx; h.amb = s[0];
		R->holes.push_back(h);
	}
	fclose(f);
	f = fopen((p + ".pac").c_str(), "rb");
	if (!f) { delete R; return nabwa_fail(NABWA_EIO, "cannot read %s.pac", prefix); }
	fseek(f, 0, SEEK_END); long sz = ftell(f); fseek(f, 0, SEEK_SET);
	R->pac.resize(sz + 8);
	if (fread(R->pac.data(), 1, sz, f) != (size_t)sz || sz < R->l_pac / 4) { fclose(f); delete R; return nabwa_fail(NABWA_EIO, "short %s.pac", prefix); }
	fclose(f);
	delete ix->ref;
	ix->ref = R;
	return NABWA_OK;
}

/* The same from memory: one or more contigs (names, offsets, lengths), ambiguity holes, the packed bases (.pac layout: 4 per
 * byte, first base in the top bits).  bench.py attaches its synthetic genome this way. */
extern "C" int nabwa_index_set_reference(nabwa_index_t *ix, int64_t l_pac, uint32_t seed, int n_seqs, const char *const *names,
										  const int64_t *offsets, const int32_t *lens, int n_holes, const int64_t *hole_off,
										  const int32_t *hole_len, const char *hole_amb, const uint8_t *pac)
{
	if (!ix || l_pac < 0 || n_seqs < 1 || !offsets || !lens || !pac || (n_holes && (!hole_off || !hole_len || !hole_amb))) return nabwa_fail(NABWA_EINVAL, "bad argument");
	nabwa_reference *R = new nabwa_reference();
	R->l_pac = l_pac; R->seed = seed;
	for (int i = 0; i < n_seqs; ++i) { nabwa_ann a; a.offset = offsets[i]; a.len = lens[i]; a.n_ambs = 0; a.name = names && names[i] ? names[i] : "seq"; R->anns.push_back(a); }
	for (int i = 0; i < n_holes; ++i) { nabwa_hole h; h.offset = hole_off[i]; h.len = hole_len[i]; h.amb = hole_amb[i]; R->holes.push_back(h); }
	R->pac.assign(pac, pac + (l_pac + 3) / 4);
	R->pac.resize(R->pac.size() + 8);
	delete ix->ref;
	ix->ref = R;
	return NABWA_OK;
}

extern "C" int nabwa_index_n_contigs(const nabwa_index_t *ix) { return ix && ix->ref ? (int)ix->ref->anns.size() : 0; }
extern "C" int nabwa_index_contig(const nabwa_index_t *ix, int i, char *name, int name_cap, int64_t *offset, int32_t *len)
{
	if (!ix || !ix->ref || i < 0 || i >= (int)ix->ref->anns.size()) return nabwa_fail(NABWA_EINVAL, "no such contig");
	const nabwa_ann &a = ix->ref->anns[i];
	if (name && name_cap > 0) snprintf(name, (size_t)name_cap, "%s", a.name.c_str());
	if (offset) *offset = a.offset;
	if (len) *len = a.len;
	return NABWA_OK;
}
extern "C" int nabwa_index_reference_info(const nabwa_index_t *ix, int64_t *l_pac, uint32_t *seed)
{
	if (!ix || !ix->ref) return nabwa_fail(NABWA_EINVAL, "index has no reference attached");
	if (l_pac) *l_pac = ix->ref->l_pac;
	if (seed) *seed = ix->ref->seed;
	return NABWA_OK;
}

/* ------------------------------------------------------------------ the chain */

/* host threads of the finishing chains: slices of independent records */
static int host_threads(int n)
{
	int nt = (int)std::thread::hardware_concurrency(); if (nt < 1) nt = 1; if (nt > 16) nt = 16;
	if (getenv("NABWA_HOST_THREADS")) nt = std::max(1, atoi(getenv("NABWA_HOST_THREADS")));
	if (n < 4096) nt = 1;
	return nt;
}
static void in_threads(int nt, size_t count, const std::function<void(size_t, size_t)> &f)
{
	if (nt == 1) { f(0, count); return; }
	std::vector<std::thread> th;
	for (int t = 0; t < nt; ++t) th.emplace_back(f, count * t / nt, count * (t + 1) / nt);
	for (auto &x : th) x.join();
}

/* posn_singleton (bam2bam.c:622-641) for n reads in record order: bwa_aln2seq_core with the caller's drand48 stream, all
 * bwt_sa walks of the batch (main hits and multi hits) as one GPU batch, bwa_approx_mapQ */
static int se_posn_impl(nabwa_index_t *ix, const nabwa_gap_opt_t *opt, int n, const int64_t *off, const int32_t *full_len,
						const int32_t *n_aln, const nabwa_aln1_t *aln, int n_occ, const uint8_t *n_occ_v, uint64_t *rng48, void *out_base, size_t stride);

extern "C" int nabwa_se_posn(nabwa_index_t *ix, const nabwa_gap_opt_t *opt, int n, const int64_t *off, const int32_t *full_len,
							 const int32_t *n_aln, const nabwa_aln1_t *aln, int n_occ, uint64_t *rng48, nabwa_se_t *out)
{
	return se_posn_impl(ix, opt, n, off, full_len, n_aln, aln, n_occ, 0, rng48, out, sizeof(nabwa_se_t));
}

/* the same with a bound per read: a file that mixes singletons (max_occ_se other hits listed, bam2bam.c:629) and ends of pairs
 * (none, bam2bam.c:692-694) is still ONE drand48 stream in record order */
extern "C" int nabwa_se_posn_v(nabwa_index_t *ix, const nabwa_gap_opt_t *opt, int n, const int64_t *off, const int32_t *full_len,
							   const int32_t *n_aln, const nabwa_aln1_t *aln, const uint8_t *n_occ_v, uint64_t *rng48, nabwa_se_t *out)
{
	if (n && !n_occ_v) return nabwa_fail(NABWA_EINVAL, "null argument");
	return se_posn_impl(ix, opt, n, off, full_len, n_aln, aln, 0, n_occ_v, rng48, out, sizeof(nabwa_se_t));
}

/* records of any stride whose head is a nabwa_se_t (nabwa_pe_t starts with one): the batch front-end works in place */
int nabwa_se_posn_strided(nabwa_index_t *ix, const nabwa_gap_opt_t *opt, int n, const int64_t *off, const int32_t *full_len,
						  const int32_t *n_aln, const nabwa_aln1_t *aln, const uint8_t *n_occ_v, uint64_t *rng48, void *out_base, size_t stride)
{
	return se_posn_impl(ix, opt, n, off, full_len, n_aln, aln, 0, n_occ_v, rng48, out_base, stride);
}

static int se_posn_impl(nabwa_index_t *ix, const nabwa_gap_opt_t *opt, int n, const int64_t *off, const int32_t *full_len,
						const int32_t *n_aln, const nabwa_aln1_t *aln, int n_occ, const uint8_t *n_occ_v, uint64_t *rng48, void *out_base, size_t stride)
{
	if (!ix || !opt || !rng48 || n < 0 || (n && (!off || !n_aln || !out_base))) return nabwa_fail(NABWA_EINVAL, "null argument");
#define out_at(i_) (*rec_at(out_base, stride, (int)(i_)))
	if (n_occ < 0 || n_occ + 1 > NABWA_MAX_MULTI) return nabwa_fail(NABWA_EINVAL, "n_occ outside 0..15");
	if (n_occ_v) for (int i = 0; i < n; ++i) if (n_occ_v[i] + 1 > NABWA_MAX_MULTI) return nabwa_fail(NABWA_EINVAL, "n_occ outside 0..15");
	const uint32_t rlen = ix->bwt[1].seq_len;
	const bool timing = getenv("NABWA_TIMING") != 0;
	auto now = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
	const double t0 = now();
	/* ---- host: hit choice with the caller's RNG stream (bwase.c:19-95).  The stream is consumed in record order and how many
	 * numbers a read takes depends on the numbers themselves (a second draw follows every accepted row), so ONE light serial pass
	 * runs the generator alone over the batch -- two multiplications per draw, no record touched -- and notes its state at the slice
	 * boundaries; the slices then do the whole choice in threads, each from its own state, and draw exactly what that pass drew. */
	const int nt0 = host_threads(n);
	std::vector<uint64_t> slice_state((size_t)nt0 + 1, 0); std::vector<size_t> slice_a0((size_t)nt0 + 1, 0);
	{
		uint64_t st = *rng48; size_t a0 = 0; int next = 0;
		for (int i = 0; i <= n; ++i) {
			while (next <= nt0 && (size_t)i == (size_t)n * next / nt0) { slice_state[next] = st; slice_a0[next] = a0; ++next; }
			if (i == n) break;
			const int na = n_aln[i]; const nabwa_aln1_t *A = aln + a0;
			a0 += na;
			if (na == 0) continue;
			int cnt = 0; const int best = A[0].score;
			for (int j = 0; j < na; ++j) {                       /* choose_main's draws, nothing else */
				if (A[j].score > best) break;
				const uint32_t w = A[j].l - A[j].k + 1;
				if (rng48_next(&st) * (double)(w + cnt) > (double)cnt) (void)rng48_next(&st);
				cnt += w;
			}
		}
		*rng48 = st;
	}
	struct Part { std::vector<uint8_t> which; std::vector<uint32_t> rows; std::vector<int> look_rec, look_multi; };
	std::vector<Part> parts((size_t)nt0);
	{
		std::vector<std::thread> th;
		auto work = [&](int t) {
			Part &Q = parts[(size_t)t];
			const size_t lo = (size_t)n * t / nt0, hi = (size_t)n * (t + 1) / nt0;
			Q.which.reserve((hi - lo) + (hi - lo) / 4); Q.rows.reserve((hi - lo) + (hi - lo) / 4); Q.look_rec.reserve((hi - lo) + (hi - lo) / 4); Q.look_multi.reserve((hi - lo) + (hi - lo) / 4);
			uint64_t st = slice_state[(size_t)t]; size_t a0 = slice_a0[(size_t)t];
			for (size_t i = lo; i < hi; ++i) {
				if (i + 8 < hi) { nabwa_se_t *const f = &out_at(i + 8); __builtin_prefetch(f, 1); __builtin_prefetch(&f->nm, 1); __builtin_prefetch(&f->n_multi, 1); __builtin_prefetch(&f->flag, 1); }      /* (the fields written below, in a 3 KB record) */
				nabwa_se_t &s = out_at(i);
				memset(&s, 0, offsetof(nabwa_se_t, cigar));          /* the scalar head; arrays are only valid up to their counts */
				s.n_cigar = 0; s.nm = 0; s.md[0] = 0; s.n_multi = 0; s.flag = 0; s.seqid = 0; s.nn = 0; s.rpos = 0; s.xt = 0;
				const int len = (int)(off[i + 1] - off[i]);
				s.len = len; s.clip_len = len; s.full_len = full_len ? full_len[i] : len;
				const nabwa_aln1_t *A = aln + a0; const int na = n_aln[i];
				a0 += na;
				if (na == 0) continue;
				choose_main(s, na, A, &st);
				list_multi(s, na, A, n_occ_v ? (int)n_occ_v[i] : n_occ);
				Q.which.push_back(s.strand ? 0 : 1); Q.rows.push_back(s.sa); Q.look_rec.push_back((int)i); Q.look_multi.push_back(-1);
				for (int j = 0; j < s.n_multi; ++j) {
					Q.which.push_back(s.multi[j].strand ? 0 : 1); Q.rows.push_back(s.multi[j].pos); Q.look_rec.push_back((int)i); Q.look_multi.push_back(j);
				}
			}
		};
		if (nt0 == 1) work(0);
		else { for (int t = 0; t < nt0; ++t) th.emplace_back(work, t); for (auto &x : th) x.join(); }
	}
	std::vector<uint8_t> which; std::vector<uint32_t> rows;           /* SA lookups: [main of each mapped read][multi...] */
	std::vector<int> look_rec, look_multi;
	{
		size_t tot = 0; for (Part &Q : parts) tot += Q.rows.size();
		which.reserve(tot); rows.reserve(tot); look_rec.reserve(tot); look_multi.reserve(tot);
		for (Part &Q : parts) {
			which.insert(which.end(), Q.which.begin(), Q.which.end()); rows.insert(rows.end(), Q.rows.begin(), Q.rows.end());
			look_rec.insert(look_rec.end(), Q.look_rec.begin(), Q.look_rec.end()); look_multi.insert(look_multi.end(), Q.look_multi.begin(), Q.look_multi.end());
		}
	}
	const double t1 = now();
	/* ---- GPU: all bwt_sa walks of the batch (bwt.c:72-81) */
	std::vector<uint32_t> sa(rows.size());
	if (!rows.empty()) {
		int r = nabwa_sa_lookup(ix, (int)rows.size(), which.data(), rows.data(), sa.data());
		if (r != NABWA_OK) return r;
	}
	const int nt = host_threads(n);
	/* positions (bwase.c:146-151, bam2bam.c:635-636): every looked-up row belongs to one record field, so slices are independent */
	in_threads(nt, rows.size(), [&](size_t lo, size_t hi) {
		for (size_t t = lo; t < hi; ++t) {
			nabwa_se_t &s = out_at(look_rec[t]);
			const uint32_t p = which[t] == 0 ? sa[t] : rlen - (sa[t] + (uint32_t)s.len);
			if (look_multi[t] < 0) s.pos = p; else s.multi[look_multi[t]].pos = p;
		}
	});
	/* bwa_approx_mapQ (bwase.c:113-122); max_diff of a read follows from its length: one table instead of a Poisson sum per read */
	int longest = 0;
	for (int i = 0; i < n; ++i) if ((int)(off[i + 1] - off[i]) > longest) longest = (int)(off[i + 1] - off[i]);      /* (= the records' len, without touching a million records on one thread) */
	std::vector<int> md_of(longest + 1, opt->max_diff);
	if (opt->fnr > 0.0f) for (int L = 0; L <= longest; ++L) md_of[L] = nabwa_cal_maxdiff(L, 0.02, opt->fnr);
	in_threads(nt, (size_t)n, [&](size_t lo, size_t hi) {
		for (size_t i = lo; i < hi; ++i) {
			if (i + 8 < hi) __builtin_prefetch(&out_at(i + 8), 1);
			nabwa_se_t &s = out_at(i);
			if (s.type == 0) continue;
			const int q = approx_mapq(s, md_of[s.len]);
			s.mapQ = s.seQ = q;
		}
	});
	if (timing) fprintf(stderr, "[nabwa] se_posn %d reads: hit choice %.3f s, bwt_sa batch (%zu rows) + positions + mapQ %.3f s\n", n, t1 - t0, rows.size(), now() - t1);
	return NABWA_OK;
}
#undef out_at

/* The non-BAM part of finish_singleton (bam2bam.c:643-651) for n positioned records: bwa_refine_gapped (bwase.c:356-423 --
 * refine_gapped_core of every gapped hit as ONE GPU batch of banded global alignments, bwa_cal_md1, bwa_correct_trimmed), then
 * the flag / contig / XT logic bwa_update_bam1 applies to a single-end record (bam2bam.c:430-525). */
int nabwa_se_refine_strided(nabwa_index_t *ix, int n, const int64_t *off, const uint8_t *seq, const uint8_t *rseq, void *out_base, size_t stride);
extern "C" int nabwa_se_refine(nabwa_index_t *ix, int n, const int64_t *off, const uint8_t *seq, const uint8_t *rseq, nabwa_se_t *out)
{
	return nabwa_se_refine_strided(ix, n, off, seq, rseq, out, sizeof(nabwa_se_t));
}

int nabwa_se_refine_strided(nabwa_index_t *ix, int n, const int64_t *off, const uint8_t *seq, const uint8_t *rseq, void *out_base, size_t stride)
{
	if (!ix || n < 0 || (n && (!off || !seq || !rseq || !out_base))) return nabwa_fail(NABWA_EINVAL, "null argument");
	if (!ix->ref) return nabwa_fail(NABWA_EINVAL, "index has no reference attached (nabwa_index_attach_reference)");
	const nabwa_reference *R = ix->ref;
	const bool timing = getenv("NABWA_TIMING") != 0;
	auto now = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
	const double t2 = now();
	size_t n_jobs = 0;
	{
		int r = refine_batch(ix, out_base, stride, n, off, seq, rseq, &n_jobs);
		if (r != NABWA_OK) return r;
	}
	const double t3 = now();
	/* host threads: MD / NM, trimmed tail, flags (bwase.c:253-354, :458-571); records are independent */
	int md_over = 0;
	auto phase4 = [&](int lo, int hi) {
		std::vector<uint8_t> fwd;
		for (int i = lo; i < hi; ++i) {
			/* what a record needs lies far apart -- its head and its MD field in a 3 KB record, its window in 775 MB of packed reference --, and
			 * every piece was a cache miss in turn: the record 16 ahead and the window of the record 8 ahead are asked for now */
			if (i + 16 < hi) { const nabwa_se_t *const f = rec_at(out_base, stride, i + 16); __builtin_prefetch(f); __builtin_prefetch(f->md, 1); __builtin_prefetch(&f->flag, 1); }
			if (i + 8 < hi) { const nabwa_se_t *const f = rec_at(out_base, stride, i + 8); if (f->type) { const uint8_t *const w = R->pac.data() + (f->pos >> 2); __builtin_prefetch(w); __builtin_prefetch(w + 32); } }
			nabwa_se_t &s = *rec_at(out_base, stride, i);
			if (s.type == 0) { s.flag = 4; continue; }
			if (!md_and_trim(R, s, seq + off[i], rseq + off[i], fwd)) md_over = 1;
			int64_t end = s.pos;
			if (s.n_cigar) { for (int k = 0; k < s.n_cigar; ++k) { const int op = COP(s.cigar[k]); if (op == 0 || op == 2) end += CLEN(s.cigar[k]); } }
			else end += s.len;
			const int reflen = (int)(end - s.pos);
			s.nn = pac2real(R, s.pos, reflen, &s.seqid);
			s.flag = 0;
			if ((int64_t)s.pos + reflen - R->anns[s.seqid].offset > R->anns[s.seqid].len) { s.flag |= 4; s.mapQ = 0; }   /* bridges two contigs */
			if (s.strand) s.flag |= 16;
			s.rpos = (int64_t)s.pos - R->anns[s.seqid].offset + 1;
			s.xt = s.nn > 10 ? 'N' : "NURM"[s.type];
		}
	};
	in_threads(host_threads(n), (size_t)n, [&](size_t lo, size_t hi) { phase4((int)lo, (int)hi); });
	if (md_over) return nabwa_fail(NABWA_ECAP, "MD string longer than NABWA_MAX_MD");
	if (timing) fprintf(stderr, "[nabwa] se_refine %d reads: refinement (%zu jobs) %.3f s, md/flags %.3f s\n", n, n_jobs, t3 - t2, now() - t3);
	return NABWA_OK;
}

extern "C" int nabwa_se_finish(nabwa_index_t *ix, const nabwa_gap_opt_t *opt, int n, const int64_t *off, const uint8_t *seq,
							   const uint8_t *rseq, const int32_t *full_len, const int32_t *n_aln, const nabwa_aln1_t *aln,
							   int n_occ, uint64_t *rng48, nabwa_se_t *out)
{
	if (!ix || !opt || !rng48 || n < 0 || (n && (!off || !seq || !rseq || !n_aln || !out))) return nabwa_fail(NABWA_EINVAL, "null argument");
	if (!ix->ref) return nabwa_fail(NABWA_EINVAL, "index has no reference attached (nabwa_index_attach_reference)");
	int r = nabwa_se_posn(ix, opt, n, off, full_len, n_aln, aln, n_occ, rng48, out);
	if (r == NABWA_OK) r = nabwa_se_refine(ix, n, off, seq, rseq, out);
	return r;
}
