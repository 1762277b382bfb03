// nabwa_api.hip -- host side of libnabwa.so: the C ABI declared in include/nabwa.h.
// Plain HIP runtime calls; no torch, no CPU fallback of the compute path.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <chrono>
#include <mutex>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>
#include "../../include/nabwa.h"
#include "fm_search.hpp"
#include "fm_deep.hpp"
#include "nabwa_internal.hpp"

extern "C" {
void nabwa_launch_repack(const uint32_t *w, uint32_t seq_len, uint32_t n_buckets, uint4 *out, hipStream_t s);
void nabwa_launch_kmer_level(const DevBwt *B, const uint2 *prev, uint2 *cur, uint64_t n_cur, hipStream_t s);
void nabwa_launch_sa_fill(const DevBwt *B, uint32_t *sa_full, uint32_t *isa, uint8_t *text_bytes, hipStream_t s);
void nabwa_launch_text_pack(const uint8_t *bytes, uint32_t n, uint32_t n_words, uint32_t *out, hipStream_t s);
void nabwa_launch_sa_lookup(const DevBwt *B, int n, const uint8_t *which, const uint32_t *k, uint32_t *out, hipStream_t s);
void nabwa_launch_occ4(const DevBwt *B, int n, const uint32_t *k, uint32_t *out, hipStream_t s);
void nabwa_launch_fm_search(const SearchParams *P, int n_blocks, hipStream_t s);
void nabwa_launch_fm_width(const SearchParams *P, int n_blocks, hipStream_t s);
int nabwa_width_occupancy(void);
void nabwa_launch_checksum(int n, const int32_t *n_aln, const uint4 *aln, int aln_cap, const uint8_t *status,
						   const int32_t *wide_idx, const uint4 *aln2, int aln_cap2, const uint4 *const *grown,
						   unsigned long long *sum, unsigned long long *rows, hipStream_t s);
void nabwa_launch_collect(int n, const uint8_t *status, int32_t *ids, unsigned int *count, int which, hipStream_t s);
void nabwa_launch_fm_deep(const DeepParams *P, int n_waves, hipStream_t s);
void nabwa_launch_collect_keyed(int n, const uint8_t *status, int32_t *ids, unsigned int *count, int which,
								const uint8_t *cls, const uint8_t *md, int max_key, const int32_t *n_aln, int aln_cap, hipStream_t s);
int nabwa_deep_occupancy(int ns, int lds_rd);
void nabwa_launch_assign_slots(int n2, const int32_t *ids, int32_t *wide_idx, hipStream_t s);
void nabwa_launch_scatter_grown(int n2, const int32_t *ids, const int32_t *n_aln3, const int32_t *max_ent3, const uint8_t *status3,
								 int32_t *n_aln, int32_t *max_ent, uint8_t *status, int32_t *wide_idx, const uint4 *block, size_t cap3,
								 const uint4 **grown, int slot0, hipStream_t s);
void nabwa_launch_scatter_wide(int n2, const int32_t *ids, const int32_t *n_aln2, const int32_t *max_ent2,
							   const uint8_t *status2, int32_t *n_aln, int32_t *max_ent, uint8_t *status,
							   int32_t *wide_idx, hipStream_t s);
void nabwa_launch_gather(int n, const int32_t *n_aln, const uint32_t *row_off, const uint4 *aln, int aln_cap,
						 const uint8_t *status, const int32_t *wide_idx, const uint4 *aln2, int aln_cap2,
						 const uint4 *const *grown, uint4 *out, hipStream_t s);
int nabwa_search_occupancy(int ns);
void nabwa_launch_partition(int n, const uint8_t *cls, int32_t *ids, unsigned int *cnt, hipStream_t s);
void nabwa_launch_padded_len(int n, const int64_t *off, int64_t *plen, hipStream_t s);
void nabwa_launch_pad_reads(int n, const uint8_t *seq, const uint8_t *rseq, const int64_t *off, const int64_t *poff,
							uint8_t *pseq, uint8_t *prseq, int32_t *rd_len, uint32_t *rd_key, int T, int seed_len, uint32_t *rd_pack, int pack_stride,
							const uint8_t *md_tab, const uint8_t *mg_tab, uint8_t *rd_md, uint8_t *rd_mg, hipStream_t s);
}

static thread_local std::string g_err;
static int fail(int code, const char *fmt, const char *a = "")
{
	char buf[512]; snprintf(buf, sizeof buf, fmt, a); g_err = buf; return code;
}
int nabwa_fail(int code, const char *fmt, const char *a) { return fail(code, fmt, a); }
#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
	char b_[512]; snprintf(b_, sizeof b_, "%s failed: %s (%s:%d)", #x, hipGetErrorString(e_), __FILE__, __LINE__); \
	g_err = b_; return NABWA_ENODEV; } } while (0)

extern "C" const char *nabwa_last_error(void) { return g_err.c_str(); }

extern "C" int nabwa_device_count(void)
{
	int n = 0;
	if (hipGetDeviceCount(&n) != hipSuccess) return 0;
	return n;
}

extern "C" void nabwa_gap_init_opt(nabwa_gap_opt_t *o)   /* gap_init_opt, bwtaln.c:19-35 */
{
	memset(o, 0, sizeof(*o));
	o->s_mm = 3; o->s_gapo = 11; o->s_gape = 4;
	o->max_diff = -1; o->max_gapo = 1; o->max_gape = 6;
	o->indel_end_skip = 5; o->max_del_occ = 10; o->max_entries = 2000000;
	o->mode = NABWA_MODE_GAPE | NABWA_MODE_COMPREAD;
	o->seed_len = 32; o->max_seed_diff = 2;
	o->fnr = 0.04f; o->n_threads = 1; o->max_top2 = 30; o->trim_qual = 0;
}

/* bwa_cal_maxdiff (bwtaln.c:37-49): Poisson tail with the reference's int factorial, which
 * wraps past 12!; evaluated on the host in double, same expression order. */
extern "C" int nabwa_cal_maxdiff(int l, double err, double thres)
{
	double elambda = exp(-l * err), sum = elambda, y = 1.0;
	uint32_t x = 1;
	for (int k = 1; k < 1000; ++k) {
		y *= l * err;
		x *= (uint32_t)k;
		sum += elambda * y / (int32_t)x;
		if (1.0 - sum < thres) return k;
	}
	return 2;
}

/* ------------------------------------------------------------------ index */

static int build_one(nabwa_index *ix, int t_, const uint32_t *words, uint64_t n_words, bool on_device,
					 const uint32_t *sa_words, uint64_t n_sa_words)
{
	uint32_t hdr[5];
	if (n_words < 5) return fail(NABWA_EIO, "bwt array too short");
	if (on_device) HIPCHK(hipMemcpy(hdr, words, 20, hipMemcpyDeviceToHost)); else memcpy(hdr, words, 20);
	DevBwt &B = ix->bwt[t_];
	memset(&B, 0, sizeof(B));
	B.primary = hdr[0]; B.L2[0] = 0; B.L2[1] = hdr[1]; B.L2[2] = hdr[2]; B.L2[3] = hdr[3]; B.seq_len = hdr[4];
	/* (seq_len+15)/16 BWT words plus (seq_len+127)/128+1 checkpoints of 4 words (bwtmisc.c:130-131) */
	const uint64_t expect = ((uint64_t)B.seq_len + 15) / 16 + (((uint64_t)B.seq_len + 127) / 128 + 1) * 4;
	if (n_words - 5 < expect) return fail(NABWA_EIO, "bwt array shorter than its seq_len implies");
	B.n_buckets = (uint32_t)(((uint64_t)B.seq_len + NABWA_INTV - 1) / NABWA_INTV);
	uint32_t *raw = 0;
	const uint32_t *src = words + 5;
	if (!on_device) {
		HIPCHK(hipMalloc(&raw, (n_words - 5) * 4));
		HIPCHK(hipMemcpy(raw, words + 5, (n_words - 5) * 4, hipMemcpyHostToDevice));
		src = raw;
	}
	HIPCHK(hipMalloc(&ix->bk[t_], (size_t)B.n_buckets * 64));
	nabwa_launch_repack(src, B.seq_len, B.n_buckets, ix->bk[t_], 0);
	HIPCHK(hipGetLastError());
	HIPCHK(hipDeviceSynchronize());
	if (raw) HIPCHK(hipFree(raw));
	B.bk = ix->bk[t_];
	ix->bytes += (uint64_t)B.n_buckets * 64;
	{	/* interval table, ALL levels 1..T back to back (level t at offset (4^t - 4) / 3): T = floor(log4(seq_len)) + 1 (about a
		 * quarter row per key at the last level: most walks that the table replaces die inside it), at most 16 and no more
		 * than 40 % of the free HBM; NABWA_KMER_T overrides (0 = off).  GRCh38: T = 16, 46 GB per index.  The search keeps
		 * every gap-free entry of depth <= T as its path KEY and takes children, tails and forced walks from here. */
		int T = 0;
		for (uint64_t x = B.seq_len; x >= 4; x >>= 2) ++T;
		T += 1;
		const char *e = getenv("NABWA_KMER_T");
		if (e) T = atoi(e);
		if (T > 16) T = 16;
		size_t free_b = 0, total_b = 0;
		HIPCHK(hipMemGetInfo(&free_b, &total_b));
		auto table_entries = [](int t) { size_t x = 0; for (int u = 1; u <= t; ++u) x += (size_t)1 << (2 * u); return x; };
		/* both directions must get the same depth (the search runs without tables otherwise): the first one built decides, leaving
		 * room for the second; the second takes that depth, and says so loudly if it cannot */
		if (ix->kmer_T_pick < 0) {
			while (T > 12 && 2 * table_entries(T) * 8 > free_b / 5 * 3) --T;
			ix->kmer_T_pick = T;
		} else {
			T = ix->kmer_T_pick;
			if (T >= 1 && table_entries(T) * 8 > free_b / 10 * 9) {
				fprintf(stderr, "[nabwa] WARNING: no device memory for the second interval table of depth %d (%zu MB free): the search runs WITHOUT interval tables "
								"(several times slower); free device memory or set NABWA_KMER_T lower\n", T, free_b >> 20);
				T = 0;
			}
		}
		if (T >= 1) {
			const size_t lo_n = table_entries(T);
			HIPCHK(hipMalloc(&ix->kmer[t_], lo_n * 8));
			uint2 *prev = 0, *cur = ix->kmer[t_];
			for (int t = 1; t <= T; ++t) {
				nabwa_launch_kmer_level(&B, prev, cur, (uint64_t)1 << (2 * t), 0);
				prev = cur; cur += (size_t)1 << (2 * t);
			}
			HIPCHK(hipGetLastError());
			HIPCHK(hipDeviceSynchronize());
			ix->bytes += lo_n * 8;
			B.kmer = prev; B.kmer_T = (uint32_t)T; B.kmer_lo = ix->kmer[t_]; B.kmer_LW = (uint32_t)T;
		}
	}
	if (sa_words) {
		uint32_t sh[7];
		if (n_sa_words < 7) return fail(NABWA_EIO, "sa array too short");
		if (on_device) HIPCHK(hipMemcpy(sh, sa_words, 28, hipMemcpyDeviceToHost)); else memcpy(sh, sa_words, 28);
		if (sh[0] != B.primary || sh[6] != B.seq_len) return fail(NABWA_EIO, "SA-BWT inconsistency");   /* bwtio.c:169,173 */
		B.sa_intv = sh[5];
		B.n_sa = (uint32_t)(((uint64_t)B.seq_len + B.sa_intv) / B.sa_intv);
		if (n_sa_words - 7 < (uint64_t)B.n_sa - 1) return fail(NABWA_EIO, "sa array shorter than n_sa");
		HIPCHK(hipMalloc(&ix->sa[t_], (size_t)B.n_sa * 4));
		HIPCHK(hipMemset(ix->sa[t_], 0xff, 4));
		HIPCHK(hipMemcpy(ix->sa[t_] + 1, sa_words + 7, (size_t)(B.n_sa - 1) * 4,
						 on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice));
		B.sa = ix->sa[t_];
		ix->bytes += (uint64_t)B.n_sa * 4;
		const char *tm = getenv("NABWA_TEXT_MODE");
		if (!(tm && atoi(tm) == 0)) {      /* full SA + inverse + text, ~8.3 B per base (NABWA_TEXT_MODE=0: keep the samples only) */
			const size_t rows = (size_t)B.seq_len + 1, words = ((size_t)B.seq_len + 15) / 16 + 4;
			uint8_t *tb = 0;
			HIPCHK(hipMalloc(&ix->sa_full[t_], rows * 4)); HIPCHK(hipMalloc(&ix->isa[t_], rows * 4));
			HIPCHK(hipMalloc(&ix->text[t_], words * 4)); HIPCHK(hipMalloc(&tb, rows));
			nabwa_launch_sa_fill(&B, ix->sa_full[t_], ix->isa[t_], tb, 0);
			nabwa_launch_text_pack(tb, B.seq_len, (uint32_t)words, ix->text[t_], 0);
			HIPCHK(hipGetLastError());
			HIPCHK(hipDeviceSynchronize());
			HIPCHK(hipFree(tb));
			B.sa_full = ix->sa_full[t_]; B.isa = ix->isa[t_]; B.text = ix->text[t_];
			ix->bytes += rows * 8 + words * 4;
		}
	}
	return NABWA_OK;
}

static int env_int(const char *name, int dflt)
{
	const char *s = getenv(name);
	return s && *s ? atoi(s) : dflt;
}

/* Working buffers come from a per-index pool: a streaming caller makes one batch after the other, of about the same
 * size, and hipMalloc of the search arena (tens of GB) costs 0.5 - 1 s each time -- more than the search itself.
 * A released buffer is kept (up to NABWA_POOL_GB, default 80: a 10 M-read batch holds about 20 GB, kernel D's page pool 32 GB) and handed to the next request it fits within 25 %;
 * everything cached goes back to the driver when an allocation fails and when the index is destroyed. */
struct nabwa_dev_pool {
	std::mutex mu;
	struct Blk { void *p; size_t bytes; };
	std::vector<Blk> idle;                         /* oldest first */
	std::unordered_map<void*, size_t> live;
	size_t idle_bytes = 0, limit = 0;
	/* staged uploads (pageable caller memory -> pinned slots -> HBM), set up by the first large upload */
	enum { UP_THREADS = 4, UP_SLOT = 32 << 20 };
	uint8_t *pin = 0; hipStream_t up_stream[UP_THREADS] = {}; hipEvent_t up_ev[UP_THREADS][2] = {};
};

/* hipMemcpy from pageable memory runs at ~15 GB/s here (one staging thread inside the runtime); four host threads
 * copying into their own pinned slots while the previous slot is in flight reach the link rate.  Jobs: {dst, src, bytes}. */
struct UploadJob { void *dst; const void *src; size_t bytes; };
static hipError_t staged_upload(nabwa_index *ix, const UploadJob *jobs, int n_jobs)
{
	nabwa_dev_pool *pl = ix->pool;
	size_t total = 0;
	for (int j = 0; j < n_jobs; ++j) total += jobs[j].bytes;
	const int T = nabwa_dev_pool::UP_THREADS; const size_t SLOT = nabwa_dev_pool::UP_SLOT;
	if (total < ((size_t)env_int("NABWA_STAGED_MIN_MB", 256) << 20) || total == 0) {       /* small: not worth four threads */
		for (int j = 0; j < n_jobs; ++j)
			if (jobs[j].bytes) { hipError_t e = hipMemcpy(jobs[j].dst, jobs[j].src, jobs[j].bytes, hipMemcpyHostToDevice); if (e != hipSuccess) return e; }
		return hipSuccess;
	}
	std::lock_guard<std::mutex> lk(pl->mu);                  /* one staged upload at a time per index */
	if (!pl->pin) {
		hipError_t e = hipHostMalloc((void**)&pl->pin, (size_t)T * 2 * SLOT, hipHostMallocDefault);
		if (e != hipSuccess) { pl->pin = 0; return e; }
		for (int t = 0; t < T; ++t) {
			if ((e = hipStreamCreateWithFlags(&pl->up_stream[t], hipStreamNonBlocking)) != hipSuccess) return e;
			for (int k = 0; k < 2; ++k) if ((e = hipEventCreateWithFlags(&pl->up_ev[t][k], hipEventDisableTiming)) != hipSuccess) return e;
		}
	}
	struct Piece { uint8_t *dst; const uint8_t *src; size_t bytes; };
	std::vector<Piece> pieces;
	for (int j = 0; j < n_jobs; ++j)
		for (size_t o = 0; o < jobs[j].bytes; o += SLOT)
			pieces.push_back({ (uint8_t*)jobs[j].dst + o, (const uint8_t*)jobs[j].src + o, jobs[j].bytes - o < SLOT ? jobs[j].bytes - o : SLOT });
	hipError_t err[T];
	std::vector<std::thread> th;
	for (int t = 0; t < T; ++t) {
		err[t] = hipSuccess;
		th.emplace_back([&, t]() {
			hipError_t e = hipSetDevice(ix->device);
			int used = 0;
			for (size_t i = t; i < pieces.size() && e == hipSuccess; i += T, ++used) {
				const int k = used & 1;
				uint8_t *slot = pl->pin + ((size_t)t * 2 + k) * SLOT;
				if (used >= 2) e = hipEventSynchronize(pl->up_ev[t][k]);       /* the copy that last used this slot is done */
				if (e != hipSuccess) break;
				memcpy(slot, pieces[i].src, pieces[i].bytes);
				e = hipMemcpyAsync(pieces[i].dst, slot, pieces[i].bytes, hipMemcpyHostToDevice, pl->up_stream[t]);
				if (e == hipSuccess) e = hipEventRecord(pl->up_ev[t][k], pl->up_stream[t]);
			}
			const hipError_t e2 = hipStreamSynchronize(pl->up_stream[t]);
			err[t] = e != hipSuccess ? e : e2;
		});
	}
	for (auto &x : th) x.join();
	for (int t = 0; t < T; ++t) if (err[t] != hipSuccess) return err[t];
	return hipSuccess;
}

static void pool_flush(nabwa_dev_pool *pl)         /* caller holds the lock */
{
	for (auto &k : pl->idle) (void)hipFree(k.p);
	pl->idle.clear(); pl->idle_bytes = 0;
}

static hipError_t pool_malloc(nabwa_index *ix, void **out, size_t bytes)
{
	nabwa_dev_pool *pl = ix->pool;
	if (bytes == 0) bytes = 1;
	const size_t gran = bytes >= (8u << 20) ? (2u << 20) : 256;
	const size_t need = (bytes + gran - 1) / gran * gran;
	std::lock_guard<std::mutex> lk(pl->mu);
	size_t best = pl->idle.size();
	for (size_t i = 0; i < pl->idle.size(); ++i)
		if (pl->idle[i].bytes >= need && pl->idle[i].bytes <= need + need / 4 + 4096 && (best == pl->idle.size() || pl->idle[i].bytes < pl->idle[best].bytes)) best = i;
	if (best != pl->idle.size()) {
		*out = pl->idle[best].p; pl->live[*out] = pl->idle[best].bytes; pl->idle_bytes -= pl->idle[best].bytes;
		pl->idle.erase(pl->idle.begin() + best);
		return hipSuccess;
	}
	hipError_t e = hipMalloc(out, need);
	if (e != hipSuccess && !pl->idle.empty()) { (void)hipGetLastError(); pool_flush(pl); e = hipMalloc(out, need); }
	if (e == hipSuccess) pl->live[*out] = need;
	return e;
}

static hipError_t pool_free(nabwa_index *ix, void *p)
{
	if (!p) return hipSuccess;
	nabwa_dev_pool *pl = ix->pool;
	std::lock_guard<std::mutex> lk(pl->mu);
	auto it = pl->live.find(p);
	if (it == pl->live.end()) return hipFree(p);
	const size_t bytes = it->second;
	pl->live.erase(it);
	if (bytes > pl->limit) return hipFree(p);
	pl->idle.push_back({ p, bytes }); pl->idle_bytes += bytes;
	while (pl->idle_bytes > pl->limit) {            /* the oldest go first */
		(void)hipFree(pl->idle.front().p); pl->idle_bytes -= pl->idle.front().bytes; pl->idle.erase(pl->idle.begin());
	}
	return hipSuccess;
}

extern "C" int nabwa_index_from_arrays(int device, int is_device, const uint32_t *bwt0, uint64_t nw0,
									   const uint32_t *bwt1, uint64_t nw1, const uint32_t *sa0, uint64_t ns0,
									   const uint32_t *sa1, uint64_t ns1, nabwa_index_t **out)
{
	if (!out || !bwt0 || !bwt1) return fail(NABWA_EINVAL, "null argument");
	if (nabwa_device_count() <= device) return fail(NABWA_ENODEV, "no such HIP device");
	HIPCHK(hipSetDevice(device));
	nabwa_index *ix = new nabwa_index();
	ix->pool = new nabwa_dev_pool();
	ix->pool->limit = (size_t)env_int("NABWA_POOL_GB", 80) << 30;
	memset(ix->bwt, 0, sizeof(ix->bwt)); ix->bk[0] = ix->bk[1] = 0; ix->sa[0] = ix->sa[1] = 0; ix->kmer[0] = ix->kmer[1] = 0; ix->kmer_top[0] = ix->kmer_top[1] = 0; for (int t = 0; t < 2; ++t) ix->sa_full[t] = ix->isa[t] = ix->text[t] = 0; ix->bytes = 0; ix->ref = 0;
	ix->device = device;
	int r = build_one(ix, 0, bwt0, nw0, is_device != 0, sa0, ns0);
	if (r == NABWA_OK) r = build_one(ix, 1, bwt1, nw1, is_device != 0, sa1, ns1);
	if (r != NABWA_OK) { nabwa_index_destroy(ix); return r; }
	*out = ix;
	return NABWA_OK;
}

static bool slurp(const std::string &fn, std::vector<uint32_t> &v)
{
	FILE *f = fopen(fn.c_str(), "rb");
	if (!f) return false;
	fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET);
	v.resize((n + 3) / 4);
	bool ok = fread(v.data(), 1, n, f) == (size_t)n;
	fclose(f);
	return ok;
}

extern "C" int nabwa_index_load(const char *prefix, int device, int with_sa, int with_pac, nabwa_index_t **out)
{
	if (!prefix || !out) return fail(NABWA_EINVAL, "null argument");
	std::vector<uint32_t> b0, b1, s0, s1;
	std::string p(prefix);
	if (!slurp(p + ".bwt", b0)) return fail(NABWA_EIO, "cannot read %s.bwt", prefix);
	if (!slurp(p + ".rbwt", b1)) return fail(NABWA_EIO, "cannot read %s.rbwt", prefix);
	if (with_sa) {
		if (!slurp(p + ".sa", s0)) return fail(NABWA_EIO, "cannot read %s.sa", prefix);
		if (!slurp(p + ".rsa", s1)) return fail(NABWA_EIO, "cannot read %s.rsa", prefix);
	}
	int r = nabwa_index_from_arrays(device, 0, b0.data(), b0.size(), b1.data(), b1.size(),
									with_sa ? s0.data() : 0, s0.size(), with_sa ? s1.data() : 0, s1.size(), out);
	if (r == NABWA_OK && with_pac) {
		r = nabwa_index_attach_reference(*out, prefix);
		if (r != NABWA_OK) { nabwa_index_destroy(*out); *out = 0; }
	}
	return r;
}

extern "C" void nabwa_index_destroy(nabwa_index_t *ix)
{
	if (!ix) return;
	(void)hipSetDevice(ix->device);
	for (int t = 0; t < 2; ++t) { if (ix->bk[t]) (void)hipFree(ix->bk[t]); if (ix->sa[t]) (void)hipFree(ix->sa[t]); if (ix->kmer[t]) (void)hipFree(ix->kmer[t]); if (ix->kmer_top[t]) (void)hipFree(ix->kmer_top[t]);
		if (ix->sa_full[t]) (void)hipFree(ix->sa_full[t]); if (ix->isa[t]) (void)hipFree(ix->isa[t]); if (ix->text[t]) (void)hipFree(ix->text[t]); }
	if (ix->pool) {
		{ std::lock_guard<std::mutex> lk(ix->pool->mu); pool_flush(ix->pool); }
		if (ix->pool->pin) (void)hipHostFree(ix->pool->pin);
		for (int t = 0; t < nabwa_dev_pool::UP_THREADS; ++t) {
			if (ix->pool->up_stream[t]) (void)hipStreamDestroy(ix->pool->up_stream[t]);
			for (int k = 0; k < 2; ++k) if (ix->pool->up_ev[t][k]) (void)hipEventDestroy(ix->pool->up_ev[t][k]);
		}
		delete ix->pool;
	}
	delete ix->ref;
	delete ix;
}

/* Read-back of the derived index parts (tests): what 0 = sa_full[first..), 1 = isa[first..), 2 = text bases first.. (one per
 * word), 3 = interval table, level kmer_T: entry pairs {k, l} of keys first.. (2 words each), 4 = kmer_T (one word). */
extern "C" int nabwa_index_export(const nabwa_index_t *ix, int which, int what, uint64_t first, uint64_t n, uint32_t *out)
{
	if (!ix || !out || which < 0 || which > 1) return fail(NABWA_EINVAL, "bad argument");
	HIPCHK(hipSetDevice(ix->device));
	const DevBwt &B = ix->bwt[which];
	if (what == 4) { out[0] = B.kmer_T; return NABWA_OK; }
	if (what == 3) {
		if (!B.kmer || first + n > (1ull << (2 * B.kmer_T))) return fail(NABWA_EINVAL, "no interval table / out of range");
		HIPCHK(hipMemcpy(out, B.kmer + first, n * 8, hipMemcpyDeviceToHost));
		return NABWA_OK;
	}
	if (!B.sa_full) return fail(NABWA_EINVAL, "index has no text-mode companions (no SA given, or NABWA_TEXT_MODE=0)");
	if (what == 0 || what == 1) {
		if (first + n > (uint64_t)B.seq_len + 1) return fail(NABWA_EINVAL, "out of range");
		HIPCHK(hipMemcpy(out, (what ? B.isa : B.sa_full) + first, n * 4, hipMemcpyDeviceToHost));
		return NABWA_OK;
	}
	if (what == 2) {
		if (first + n > (uint64_t)B.seq_len) return fail(NABWA_EINVAL, "out of range");
		const uint64_t w0 = first / 16, w1 = (first + n + 15) / 16;
		std::vector<uint32_t> w(w1 - w0);
		HIPCHK(hipMemcpy(w.data(), B.text + w0, (w1 - w0) * 4, hipMemcpyDeviceToHost));
		for (uint64_t j = 0; j < n; ++j) { const uint64_t p = first + j; out[j] = w[p / 16 - w0] >> (2 * (p & 15)) & 3u; }
		return NABWA_OK;
	}
	return fail(NABWA_EINVAL, "unknown part");
}

extern "C" uint32_t nabwa_index_seq_len(const nabwa_index_t *ix, int which) { return ix->bwt[which & 1].seq_len; }
extern "C" uint64_t nabwa_index_device_bytes(const nabwa_index_t *ix) { return ix->bytes; }

/* ------------------------------------------------------------------ batch */

struct nabwa_batch {
	nabwa_index *ix;
	nabwa_gap_opt_t opt;
	int n;
	hipStream_t stream;
	hipEvent_t ev0, ev1, evw;
	float last_ms;
	// device inputs
	uint8_t *d_seq, *d_rseq, *d_md, *d_mg; int64_t *d_poff; int32_t *d_len; uint32_t *d_key, *d_pack; int pack_stride; uint8_t *d_cls; int32_t *d_perm; unsigned int *d_ncls; int max_len;
	// first pass
	SearchParams P; int class_sort; uint32_t NS_wide; int n_blocks, n_blocks_w; uint8_t *d_scratch, *d_wdata, *d_nN; float last_ms_w;
	int32_t *d_naln, *d_maxent, *d_wide_idx; uint8_t *d_status; uint4 *d_aln;
	unsigned int *d_counter, *d_novf; int32_t *d_ovf_ids;
	uint8_t *grown[8]; int n_grown;           // row blocks of the reads whose hit lists outgrew the wide rows (nabwa_batch_sync)
	const uint4 **d_grown_tab; int grown_cap, grown_used;      // device table: slot -> rows of one such read (wide_idx of a NABWA_ST_GROWN read)
	// wide pass (allocated on demand)
	int n2, aln_cap2; uint8_t *d_scratch2; size_t scratch2_bytes; int32_t *d_naln2, *d_maxent2; uint8_t *d_status2; uint4 *d_aln2;
	int unresolved;
	unsigned long long *d_sum;
	// kernel D (deep searches): page pool, per-wave page lists and staging, counters; allocated on demand, kept for the next run
	uint4 *d_pages; uint32_t *d_page_prev, *d_deep_own; uint4 *d_deep_stage; unsigned long long *d_deep_ctr;
	uint32_t *d_ixtab;
	size_t deep_pages, deep_own_words, deep_stage_ent;
	hipEvent_t evd0, evd1; float last_ms_deep; int deep_ran, deep_only;
	int deep_cfg; uint32_t deep_K, deep_lds_rd, deep_rd_pl; size_t deep_n_pages; uint64_t deep_cap_pages; long deep_waves_max;
};

static uint32_t align_up(uint32_t x, uint32_t a) { return (x + a - 1) / a * a; }

static void layout(SearchParams &P, uint32_t cap, int max_len, int seed_len, uint32_t NS)
{
	// per-read width record
	P.WL = align_up((uint32_t)max_len + 1, 16);
	P.WLB = P.WL + 16;
	P.SLB = align_up((uint32_t)(max_len > seed_len ? seed_len : 0) + 1, 16) + 16;      /* seed bounds exist only for reads longer than the seed (bwtaln.c:126-130) */
	P.woff_bid = 2 * P.WL * 4;
	P.woff_sbid = P.woff_bid + 2 * P.WLB;
	P.wstride = align_up(P.woff_sbid + 2 * P.SLB, 64);
	// per-lane search scratch
	P.cap = cap; P.NS = NS;
	P.lane_stride = align_up((size_t)cap * 16, 64);
}

extern "C" void nabwa_batch_destroy(nabwa_batch_t *b)
{
	if (!b) return;
	(void)hipSetDevice(b->ix->device);
	void *ptrs[] = { b->d_pack, b->d_cls, b->d_perm, b->d_ncls, b->d_key, b->d_wdata, b->d_nN, b->d_seq, b->d_rseq, b->d_md, b->d_mg, b->d_poff, b->d_len, b->d_scratch, b->d_naln, b->d_maxent, b->d_wide_idx,
					 b->d_status, b->d_aln, b->d_counter, b->d_novf, b->d_ovf_ids, b->d_scratch2, b->d_naln2, b->d_maxent2,
					 b->d_status2, b->d_aln2, b->d_sum, b->d_pages, b->d_page_prev, b->d_deep_own, b->d_deep_stage, b->d_deep_ctr, b->d_ixtab };
	if (b->stream) (void)hipStreamSynchronize(b->stream);      /* the buffers go back to the pool, not to the driver: nothing may still use them */
	for (void *p : ptrs) if (p) (void)pool_free(b->ix, p);
	for (int t = 0; t < b->n_grown; ++t) (void)pool_free(b->ix, b->grown[t]);
	if (b->d_grown_tab) (void)pool_free(b->ix, (void*)b->d_grown_tab);
	if (b->ev0) (void)hipEventDestroy(b->ev0);
	if (b->ev1) (void)hipEventDestroy(b->ev1);
	if (b->evw) (void)hipEventDestroy(b->evw);
	if (b->evd0) (void)hipEventDestroy(b->evd0);
	if (b->evd1) (void)hipEventDestroy(b->evd1);
	if (b->stream) (void)hipStreamDestroy(b->stream);
	delete b;
}

#define BCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
	char b_[512]; snprintf(b_, sizeof b_, "%s failed: %s (%s:%d)", #x, hipGetErrorString(e_), __FILE__, __LINE__); \
	g_err = b_; nabwa_batch_destroy(b); return NABWA_ENODEV; } } while (0)

extern "C" int nabwa_batch_create(nabwa_index_t *ix, const nabwa_gap_opt_t *opt, int n, const int64_t *off,
								  const uint8_t *seq, const uint8_t *rseq, int per_read, nabwa_batch_t **out)
{
	if (!ix || !opt || !off || !out || n < 0 || (n && (!seq || !rseq))) return fail(NABWA_EINVAL, "null argument");
	HIPCHK(hipSetDevice(ix->device));
	const bool timing = getenv("NABWA_TIMING") != 0;
	auto now = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
	const double tc0 = now();
	// ---- per-read option derivation, on the host in double (bwtaln.c:102-106,125)
	// One threaded pass over the read boundaries: validity, the lengths that occur, the padded size.  Everything per read
	// that follows from its length alone (max_diff, max_gapo) is a table over lengths, applied on the device.
	int max_len = 0, min_len = 65535; int64_t padded_total = 0; bool bad_len = false;
	std::vector<uint8_t> seen(65536, 0);
	{
		const int NT = n >= (1 << 20) ? 4 : 1;
		struct Part { int mx = 0, mn = 65535; int64_t padded = 0; bool bad = false; std::vector<uint8_t> seen; };
		std::vector<Part> part(NT);
		std::vector<std::thread> th;
		for (int t = 0; t < NT; ++t) {
			part[t].seen.assign(65536, 0);
			auto work = [&, t]() {
				Part &q = part[t];
				const int64_t i0 = (int64_t)n * t / NT, i1 = (int64_t)n * (t + 1) / NT;
				for (int64_t i = i0; i < i1; ++i) {
					const int64_t L = off[i + 1] - off[i];
					if (L < 0 || L > 65535) { q.bad = true; continue; }
					q.seen[L] = 1; q.padded += (L + 15) / 16 * 16;
					if (L > q.mx) q.mx = (int)L;
					if (L < q.mn) q.mn = (int)L;
				}
			};
			if (NT == 1) work(); else th.emplace_back(work);
		}
		for (auto &x : th) x.join();
		for (auto &q : part) {
			bad_len |= q.bad; padded_total += q.padded;
			if (q.mx > max_len) max_len = q.mx;
			if (q.mn < min_len) min_len = q.mn;
			for (int L = 0; L < 65536; ++L) seen[L] |= q.seen[L];
		}
	}
	if (bad_len) return fail(NABWA_EINVAL, "read length outside 0..65535");
	std::vector<int> md_of(max_len + 1, opt->max_diff);
	if (opt->fnr > 0.0f) for (int L = 0; L <= max_len; ++L) md_of[L] = nabwa_cal_maxdiff(L, 0.02, opt->fnr);
	std::vector<uint8_t> md_tab(max_len + 1, 0), mg_tab(max_len + 1, 0);     /* by read length: what the search of such a read runs with */
	uint32_t NS = 1;
	int mdx = 0, mgx = 0;
	for (int L = 0; L <= max_len; ++L) {
		if (!seen[L]) continue;
		const int md_sizing = md_of[per_read ? L : max_len];
		int g = opt->max_gapo; if (md_sizing < g) g = md_sizing;
		const int d = md_of[L];
		/* kernel D's entries keep n_mm / n_gapo / n_gape in 8 bits each, the bound bytes of kernel W hold min(bid, 127) */
		if (d < 0 || d > 126 || g < 0 || g > 255) return fail(NABWA_EINVAL, "max_diff > 126 or max_gapo > 255 (unsupported)");
		md_tab[L] = (uint8_t)d; mg_tab[L] = (uint8_t)g;
		if (d > mdx) mdx = d;
		if (g > mgx) mgx = g;
		const long ns = (long)(md_sizing + 1) * opt->s_mm + (long)(g + 1) * opt->s_gapo + (long)(opt->max_gape + 1) * opt->s_gape;
		if (ns > (long)NS) NS = (uint32_t)ns;
	}
	/* First pass: score levels that can actually hold an entry.  A child is only made of a parent that passed the
	 * m >= 0 test (bwtgap.c:152-154), so it has at most max_diff + 1 counted differences, at most max_gapo opens and
	 * max_gape extensions; its score is the largest index the per-score lists are addressed with.  (The formula above
	 * is the reference's initial best_score, a value that is compared, never an index; the second pass still sizes by it.) */
	uint32_t NS1 = 1;
	{
		const bool gape_counts = opt->mode & NABWA_MODE_GAPE;
		for (int a = 0; a <= mdx + 1; ++a)
			for (int g = 0; g <= mgx; ++g)
				for (int e = 0; e <= (g ? opt->max_gape : 0); ++e) {
					if (a + g + (gape_counts ? e : 0) > mdx + 1) continue;
					const long sc = (long)a * opt->s_mm + (long)g * opt->s_gapo + (long)e * opt->s_gape;
					if (sc + 1 > (long)NS1) NS1 = (uint32_t)(sc + 1);
				}
		if (NS1 > NS) NS1 = NS;
	}
	/* the reference packs the score into 11 bits (bwtgap.c:58) */
	if (NS > 2048 || opt->s_mm < 0 || opt->s_gapo < 0 || opt->s_gape < 0 || opt->max_gape < 0 || opt->max_gape > 255)
		return fail(NABWA_EINVAL, "option block needs more than 2048 score levels (the reference's own limit)");
	/* first-pass (kernel S) arena entries keep n_mm / n_gapo in 4 bits and n_gape in 5, and it tracks 64 score levels:
	 * option blocks beyond that go to kernel D whole */
	const bool deep_only = mdx > 14 || mgx > 15 || opt->max_gape > 31 || NS1 > 64;
	if (opt->seed_len < 0) return fail(NABWA_EINVAL, "negative seed_len");
	/* refused before any work, not when the first read reaches kernel D in the middle of a file: there a chain's matching child must be
	 * the only child of its own score (fm_deep_body.hpp), so every penalty has to be positive (-M / -O / -E 0 have no use in practice) */
	if (opt->s_mm < 1 || opt->s_gapo < 1 || opt->s_gape < 1) return fail(NABWA_EINVAL, "s_mm, s_gapo and s_gape must be >= 1 (-M / -O / -E 0 are not supported)");

	const double tc1 = now();
	nabwa_batch *b = new nabwa_batch();
	memset(b, 0, sizeof(*b));
	b->ix = ix; b->opt = *opt; b->n = n; b->deep_only = deep_only ? 1 : 0;
	/* a stream that does not synchronise with the legacy default stream: the finishing chains of another batch (bwt_sa batches, the
	 * alignment kernels: default stream, synchronous copies) run from another thread while this batch's search kernels do; everything in
	 * this file orders its own work on b->stream explicitly.  NABWA_STREAM_BLOCKING=1: the old kind, for comparison. */
	if (env_int("NABWA_STREAM_BLOCKING", 0)) BCHK(hipStreamCreate(&b->stream));
	else BCHK(hipStreamCreateWithFlags(&b->stream, hipStreamNonBlocking));
	BCHK(hipEventCreate(&b->ev0));
	BCHK(hipEventCreate(&b->ev1));
	BCHK(hipEventCreate(&b->evw));
	BCHK(hipEventCreate(&b->evd0));
	BCHK(hipEventCreate(&b->evd1));
	// reads: upload as given, then re-lay out on the device with 16-byte aligned starts
	const size_t nb = (size_t)off[n] > 0 ? (size_t)off[n] : 1, pnb = (size_t)padded_total + 64;
	if ((uint64_t)pnb >= (1ull << 32)) { nabwa_batch_destroy(b); return fail(NABWA_EINVAL, "batch holds 4 Gi padded bases or more: split it (lane state keeps a 32-bit read offset)"); }
	b->max_len = max_len;
	BCHK(pool_malloc(b->ix, (void**)&b->d_seq, pnb)); BCHK(pool_malloc(b->ix, (void**)&b->d_rseq, pnb));
	BCHK(pool_malloc(b->ix, (void**)&b->d_poff, (size_t)(n + 1) * 8)); BCHK(pool_malloc(b->ix, (void**)&b->d_len, (size_t)(n ? n : 1) * 4));
	BCHK(pool_malloc(b->ix, (void**)&b->d_md, n ? n : 1)); BCHK(pool_malloc(b->ix, (void**)&b->d_mg, n ? n : 1)); BCHK(pool_malloc(b->ix, (void**)&b->d_key, (size_t)(n ? n : 1) * 24));
	b->pack_stride = 2 * ((max_len + 15) / 16 + 2);
	BCHK(pool_malloc(b->ix, (void**)&b->d_pack, (size_t)(n ? n : 1) * b->pack_stride * 4));
	BCHK(pool_malloc(b->ix, (void**)&b->d_cls, (size_t)(n ? n : 1) * 2)); BCHK(pool_malloc(b->ix, (void**)&b->d_perm, (size_t)(n ? n : 1) * 4)); BCHK(pool_malloc(b->ix, (void**)&b->d_ncls, 64));
	if (n == 0) BCHK(hipMemset(b->d_poff, 0, 8));
	if (n) {
		uint8_t *raw_s = 0, *raw_r = 0; int64_t *raw_off = 0;
		BCHK(pool_malloc(b->ix, (void**)&raw_s, nb)); BCHK(pool_malloc(b->ix, (void**)&raw_r, nb)); BCHK(pool_malloc(b->ix, (void**)&raw_off, (size_t)(n + 1) * 8));
		const UploadJob jobs[3] = { { raw_s, seq, (size_t)off[n] }, { raw_r, rseq, (size_t)off[n] }, { raw_off, off, (size_t)(n + 1) * 8 } };
		const double tu0 = now();
		BCHK(staged_upload(ix, jobs, 3));
		if (timing) fprintf(stderr, "[nabwa] upload of %.2f GB: %.3f s (%.3f s into batch_create)\n", 2e-9 * (double)off[n], now() - tu0, tu0 - tc0);
		// padded starts: exclusive scan of the padded lengths, on the device
		int64_t *plen = 0; void *d_tmp = 0; size_t tmp_bytes = 0; uint8_t *d_tab = 0;
		BCHK(pool_malloc(b->ix, (void**)&plen, (size_t)(n + 1) * 8));
		nabwa_launch_padded_len(n, raw_off, plen, b->stream);
		BCHK(hipcub::DeviceScan::ExclusiveSum(nullptr, tmp_bytes, plen, b->d_poff, n + 1, b->stream));
		BCHK(pool_malloc(b->ix, &d_tmp, tmp_bytes ? tmp_bytes : 16));
		BCHK(hipcub::DeviceScan::ExclusiveSum(d_tmp, tmp_bytes, plen, b->d_poff, n + 1, b->stream));
		BCHK(pool_malloc(b->ix, (void**)&d_tab, 2 * (size_t)(max_len + 1)));
		BCHK(hipMemcpyAsync(d_tab, md_tab.data(), max_len + 1, hipMemcpyHostToDevice, b->stream));
		BCHK(hipMemcpyAsync(d_tab + max_len + 1, mg_tab.data(), max_len + 1, hipMemcpyHostToDevice, b->stream));
		nabwa_launch_pad_reads(n, raw_s, raw_r, raw_off, b->d_poff, b->d_seq, b->d_rseq, b->d_len, b->d_key,
							   ix->bwt[0].kmer_T == ix->bwt[1].kmer_T ? (int)ix->bwt[0].kmer_T : 0, opt->seed_len, b->d_pack, b->pack_stride,
							   d_tab, d_tab + max_len + 1, b->d_md, b->d_mg, b->stream);
		BCHK(hipStreamSynchronize(b->stream));
		if (timing) fprintf(stderr, "[nabwa] re-layout kernel done %.3f s into batch_create\n", now() - tc0);
		BCHK(pool_free(b->ix, raw_s)); BCHK(pool_free(b->ix, raw_r)); BCHK(pool_free(b->ix, raw_off));
		BCHK(pool_free(b->ix, plen)); BCHK(pool_free(b->ix, d_tmp)); BCHK(pool_free(b->ix, d_tab));
	}

	const double tc2 = now();
	SearchParams &P = b->P;
	memset(&P, 0, sizeof(P));
	P.bwt[0] = ix->bwt[0]; P.bwt[1] = ix->bwt[1];
	{	/* the per-index constants the search kernels pick per lane, as a table for their LDS (fm_search.hpp) */
		uint32_t tab[NABWA_IXTAB_WORDS];
		nabwa_ixtab_fill(tab, ix->bwt);
		BCHK(pool_malloc(b->ix, (void**)&b->d_ixtab, sizeof tab));
		BCHK(hipMemcpy(b->d_ixtab, tab, sizeof tab, hipMemcpyHostToDevice));
		P.ixtab = b->d_ixtab;
	}
	P.seq = b->d_seq; P.rseq = b->d_rseq; P.poff = b->d_poff; P.rd_len = b->d_len; P.rd_maxdiff = b->d_md; P.rd_maxgapo = b->d_mg; P.rd_key = b->d_key; P.rd_pack = b->d_pack; P.pack_stride = b->pack_stride;
	P.text_mode = env_int("NABWA_TEXT_KERNELS", 7);      /* bit 0: text mode in the width kernel, bit 1: in the search kernel, bit 2: key form in the search kernel */
	if (ix->bwt[0].kmer_T != ix->bwt[1].kmer_T) P.bwt[0].kmer_T = P.bwt[1].kmer_T = 0;
	P.ids = 0; P.n = n;
	P.s_mm = opt->s_mm; P.s_gapo = opt->s_gapo; P.s_gape = opt->s_gape; P.mode = opt->mode;
	P.indel_end_skip = opt->indel_end_skip; P.max_del_occ = opt->max_del_occ; P.max_entries = opt->max_entries;
	P.max_gape = opt->max_gape; P.max_seed_diff = opt->max_seed_diff; P.seed_len = opt->seed_len; P.max_top2 = opt->max_top2;
	int cap1 = env_int("NABWA_CAP1", 4096);
	if (cap1 < 16) cap1 = 16;
	if (cap1 > 65534) cap1 = 65534;
	layout(P, (uint32_t)cap1, max_len, opt->seed_len, deep_only ? 1u : NS1);
	b->NS_wide = NS;
	P.aln_cap = env_int("NABWA_ALNCAP1", 16);
	P.sync_refill = env_int("NABWA_SYNC_REFILL", 0);
	/* trips after which kernel S hands a search on to kernel D (0: never).  Seeded searches (the default options): measured 10 M x 100 bp,
	 * 0 -> 123.6 ms per pass, 2000 -> 108.9 (1111 reads handed on, kernel D 2 ms), 1000 -> 116.2, 500 -> 189.6; 2 x 150 bp pairs: 2000 is the
	 * optimum too.  Without a seed (reads no longer than seed_len: the ancient-DNA options) the searches that do not end early are deep, and
	 * every trip kernel S spends on them is spent again by kernel D: 6.25 M reads, 300 / 1000 / 2000 / 5000 -> 2.29 / 2.20 / 2.20 / 2.04 M reads/s */
	P.trip_budget = (uint32_t)env_int("NABWA_TRIP_BUDGET", max_len > opt->seed_len ? 2000 : 300);
	P.trip_budget_hard = (uint32_t)env_int("NABWA_TRIP_BUDGET_HARD", max_len > opt->seed_len ? 200 : (int)P.trip_budget);      /* (a batch of reads that mostly occur on neither strand: fm_search.hip) */
	if (getenv("NABWA_TRIP_BUDGET") && !getenv("NABWA_TRIP_BUDGET_HARD")) P.trip_budget_hard = P.trip_budget;                       /* (a sweep of the one knob means the one budget) */
	b->class_sort = env_int("NABWA_CLASS_SORT", 1);
	{
		P.w_sync = (n > 0 && min_len == max_len) ? env_int("NABWA_W_SYNC", 1) : 0;
	}
	if (P.aln_cap < 1) P.aln_cap = 1;

	hipDeviceProp_t prop;
	BCHK(hipGetDeviceProperties(&prop, ix->device));
	int occ = nabwa_search_occupancy(deep_only ? 1 : (int)NS1);
	if (occ < 1) occ = 1;
	const int occ_env = env_int("NABWA_BLOCKS_PER_CU", 0);
	if (occ_env > 0) occ = occ_env;
	long blocks = (long)prop.multiProcessorCount * occ;
	const long need = ((long)n + NABWA_SEARCH_BLOCK - 1) / NABWA_SEARCH_BLOCK;
	if (blocks > need) blocks = need;
	if (blocks < 1) blocks = 1;
	b->n_blocks = (int)blocks;
	if (getenv("NABWA_TIMING")) fprintf(stderr, "[nabwa] search kernel: %u score levels, %d blocks per CU (LDS %zu B per block), %ld blocks\n", NS1, occ,
										(size_t)NS1 * NABWA_SEARCH_BLOCK * 2 + NABWA_SEARCH_BLOCK * 80, blocks);
	BCHK(pool_malloc(b->ix, (void**)&b->d_scratch, (size_t)blocks * NABWA_SEARCH_BLOCK * P.lane_stride));
	int occw = nabwa_width_occupancy(); if (occw < 1) occw = 1;
	long blocks_w = (long)prop.multiProcessorCount * occw;
	if (blocks_w > 2 * need) blocks_w = 2 * need;        /* kernel W: one lane per strand of a read */
	if (blocks_w < 1) blocks_w = 1;
	b->n_blocks_w = (int)blocks_w;
	BCHK(pool_malloc(b->ix, (void**)&b->d_wdata, (size_t)(n ? n : 1) * P.wstride));
	BCHK(pool_malloc(b->ix, (void**)&b->d_nN, n ? n : 1));
	P.wdata = b->d_wdata; P.rd_nN = b->d_nN;
	const size_t n1 = n ? n : 1;
	BCHK(pool_malloc(b->ix, (void**)&b->d_naln, n1 * 4)); BCHK(pool_malloc(b->ix, (void**)&b->d_maxent, n1 * 4)); BCHK(pool_malloc(b->ix, (void**)&b->d_wide_idx, n1 * 4));
	BCHK(pool_malloc(b->ix, (void**)&b->d_status, n1)); BCHK(pool_malloc(b->ix, (void**)&b->d_aln, n1 * (size_t)P.aln_cap * 16));
	BCHK(pool_malloc(b->ix, (void**)&b->d_counter, 16)); BCHK(pool_malloc(b->ix, (void**)&b->d_novf, 4)); BCHK(pool_malloc(b->ix, (void**)&b->d_ovf_ids, n1 * 4));
	BCHK(pool_malloc(b->ix, (void**)&b->d_sum, 256));
	P.scratch = b->d_scratch; P.n_aln = b->d_naln; P.max_ent = b->d_maxent; P.status = b->d_status; P.aln = b->d_aln;
	P.work_counter = b->d_counter;
	*out = b;
	if (timing) fprintf(stderr, "[nabwa] batch_create %d reads: host option derivation %.3f s, read upload + device layout %.3f s, working buffers %.3f s\n", n, tc1 - tc0, tc2 - tc1, now() - tc2);
	return NABWA_OK;
}

extern "C" int nabwa_batch_run(nabwa_batch_t *b)
{
	if (!b) return fail(NABWA_EINVAL, "null batch");
	HIPCHK(hipSetDevice(b->ix->device));
	b->unresolved = 0;
	if (b->n == 0) return NABWA_OK;
	if (b->n_grown) {          /* row blocks of the previous run's longest hit lists: that run's results are gone with this one */
		HIPCHK(hipStreamSynchronize(b->stream));
		for (int t = 0; t < b->n_grown; ++t) HIPCHK(pool_free(b->ix, b->grown[t]));
		b->n_grown = 0; b->grown_used = 0;
	}
	HIPCHK(hipMemsetAsync(b->d_counter, 0, 16, b->stream));
	HIPCHK(hipMemsetAsync(b->d_novf, 0, 4, b->stream));
	HIPCHK(hipEventRecord(b->evw, b->stream));
	SearchParams PW = b->P; PW.ids = 0; PW.rd_cls = b->class_sort ? b->d_cls : 0;
	nabwa_launch_fm_width(&PW, b->n_blocks_w, b->stream);
	if (b->class_sort) {      /* work order of the search: reads with an exact occurrence first, in lockstep waves */
		HIPCHK(hipMemsetAsync(b->d_ncls, 0, 64, b->stream));
		nabwa_launch_partition(b->n, b->d_cls, b->d_perm, b->d_ncls, b->stream);
	}
	HIPCHK(hipEventRecord(b->ev0, b->stream));
	SearchParams PS = b->P; PS.ids = b->class_sort ? b->d_perm : 0; PS.n_sync = b->class_sort ? b->d_ncls + 10 : 0;
	if (!b->deep_only) nabwa_launch_fm_search(&PS, b->n_blocks, b->stream);
	else {      /* option blocks the first-pass kernel's compact entries cannot hold: every read goes to kernel D */
		HIPCHK(hipMemsetAsync(b->d_status, NABWA_ST_OVERFLOW, b->n, b->stream));
		HIPCHK(hipMemsetAsync(b->d_naln, 0, (size_t)b->n * 4, b->stream));
	}
	HIPCHK(hipEventRecord(b->ev1, b->stream));
	/* the reads the first pass hands on, in the order kernel D should start them (largest-looking searches first) */
	if (b->class_sort && env_int("NABWA_DEEP_ORDER", 1)) nabwa_launch_collect_keyed(b->n, b->d_status, b->d_ovf_ids, b->d_novf, NABWA_ST_OVERFLOW, b->d_cls, b->d_md, 7, b->d_naln, b->P.aln_cap, b->stream);
	else nabwa_launch_collect(b->n, b->d_status, b->d_ovf_ids, b->d_novf, NABWA_ST_OVERFLOW, b->stream);
	HIPCHK(hipGetLastError());
	return NABWA_OK;
}

extern "C" int nabwa_batch_sync(nabwa_batch_t *b, int *n_second_pass)
{
	if (!b) return fail(NABWA_EINVAL, "null batch");
	HIPCHK(hipSetDevice(b->ix->device));
	if (n_second_pass) *n_second_pass = 0;
	if (b->n == 0) return NABWA_OK;
	unsigned int novf = 0;
	HIPCHK(hipMemcpyAsync(&novf, b->d_novf, 4, hipMemcpyDeviceToHost, b->stream));
	HIPCHK(hipStreamSynchronize(b->stream));
	HIPCHK(hipEventElapsedTime(&b->last_ms, b->ev0, b->ev1));
	HIPCHK(hipEventElapsedTime(&b->last_ms_w, b->evw, b->ev0));
	if (n_second_pass) *n_second_pass = (int)novf;
	if (novf == 0) return NABWA_OK;
	// ---- the flagged reads go to kernel D (fm_deep_body.hpp): one search per wavefront, arenas paged out of one pool.
	// (Round 1 re-ran them from scratch on one lane each in tiers of growing per-lane arenas: 22 k reads/s on the
	// ancient-DNA workload, the launch as long as its longest search.)  Optionally the first-pass kernel runs once more
	// before that with the largest arena its 16-bit links address (NABWA_TIER_A=1).
	const bool timing = getenv("NABWA_TIMING") != 0;
	auto now = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
	unsigned int cur = novf;
	b->deep_ran = 0; b->last_ms_deep = 0.f;
	// the kernel-W records of the listed reads again: a search edits them in place (gap_shadow)
	auto rebuild_widths = [&](const SearchParams &Q, unsigned int cnt, bool after_first_pass) {
		SearchParams QW = Q; QW.touch_counter = 0; QW.rd_cls = 0; QW.n_sync = 0; QW.w_sync = 0;
		/* only a search that found a hit has edited its record; after kernel D has had a read (the guaranteed pass, the searches with longer
		 * hit lists) its record is rebuilt whatever it found */
		QW.w_skip_clean = after_first_pass && env_int("NABWA_W_SKIP_CLEAN", 1) ? 1 : 0;
		QW.n_aln = b->d_naln; QW.max_ent = b->d_maxent; QW.status = b->d_status; QW.aln = b->d_aln; QW.aln_cap = b->P.aln_cap;
		long bw2 = (2 * (long)cnt + NABWA_SEARCH_BLOCK - 1) / NABWA_SEARCH_BLOCK;     /* one lane per strand */
		if (bw2 > b->n_blocks_w) bw2 = b->n_blocks_w;
		(void)hipMemsetAsync(b->d_counter, 0, 16, b->stream);       /* kernel W draws its work items from counter [1] */
		nabwa_launch_fm_width(&QW, (int)bw2, b->stream);
	};
	auto recollect = [&](int which, unsigned int *left) -> int {
		HIPCHK(hipMemsetAsync(b->d_novf, 0, 4, b->stream));
		nabwa_launch_collect(b->n, b->d_status, b->d_ovf_ids, b->d_novf, which, b->stream);
		HIPCHK(hipMemcpyAsync(left, b->d_novf, 4, hipMemcpyDeviceToHost, b->stream));
		HIPCHK(hipStreamSynchronize(b->stream));
		return NABWA_OK;
	};
	if (env_int("NABWA_TIER_A", 0) && b->P.cap < 65534 && b->deep_only == 0) {
		const double tt0 = now();
		SearchParams Q = b->P;
		layout(Q, 65534u, b->max_len, b->opt.seed_len, b->P.NS);
		long blocks = ((long)cur + NABWA_SEARCH_BLOCK - 1) / NABWA_SEARCH_BLOCK;
		size_t budget = (size_t)env_int("NABWA_WIDE_GB", 32) << 30;
		const long fit = (long)(budget / ((size_t)NABWA_SEARCH_BLOCK * Q.lane_stride));
		if (blocks > fit) blocks = fit < 1 ? 1 : fit;
		const size_t need = (size_t)blocks * NABWA_SEARCH_BLOCK * Q.lane_stride;
		if (b->scratch2_bytes < need) {
			if (b->d_scratch2) { HIPCHK(pool_free(b->ix, b->d_scratch2)); b->d_scratch2 = 0; b->scratch2_bytes = 0; }
			HIPCHK(pool_malloc(b->ix, (void**)&b->d_scratch2, need));
			b->scratch2_bytes = need;
		}
		Q.scratch = b->d_scratch2; Q.ids = b->d_ovf_ids; Q.n = (int)cur; Q.n_sync = 0; Q.w_sync = 0;
		rebuild_widths(Q, cur, false);
		HIPCHK(hipMemsetAsync(b->d_counter, 0, 16, b->stream));
		nabwa_launch_fm_search(&Q, (int)blocks, b->stream);
		HIPCHK(hipGetLastError());
		unsigned int left = 0;
		int r = recollect(NABWA_ST_OVERFLOW, &left);
		if (r != NABWA_OK) return r;
		if (timing) fprintf(stderr, "[nabwa] first-pass kernel again, arena of 65534 entries: %u reads on %ld blocks, %u left, %.3f s\n", cur, blocks, left, now() - tt0);
		cur = left;
	}
	if (cur) {
		/* a chain's matching child must be the only child of its own score (fm_deep_body.hpp) */
		if (b->opt.s_mm < 1 || b->opt.s_gapo < 1 || b->opt.s_gape < 1) return fail(NABWA_EINVAL, "deep searches need s_mm, s_gapo, s_gape >= 1");
		const double tt0 = now();
		const uint32_t NS = b->NS_wide;
		// rows of the wide result arrays for the reads that are left
		if (b->n2 < (int)cur || b->aln_cap2 != env_int("NABWA_ALNCAP2", 1024)) {
			void *old[] = { b->d_naln2, b->d_maxent2, b->d_status2, b->d_aln2 };
			for (void *p : old) if (p) (void)pool_free(b->ix, p);
			b->d_naln2 = b->d_maxent2 = 0; b->d_status2 = 0; b->d_aln2 = 0;
			b->aln_cap2 = env_int("NABWA_ALNCAP2", 1024);
			if (b->aln_cap2 < 1) b->aln_cap2 = 1;
			HIPCHK(pool_malloc(b->ix, (void**)&b->d_naln2, (size_t)cur * 4)); HIPCHK(pool_malloc(b->ix, (void**)&b->d_maxent2, (size_t)cur * 4));
			HIPCHK(pool_malloc(b->ix, (void**)&b->d_status2, cur)); HIPCHK(pool_malloc(b->ix, (void**)&b->d_aln2, (size_t)cur * b->aln_cap2 * 16));
			b->n2 = (int)cur;
		}
		nabwa_launch_assign_slots((int)cur, b->d_ovf_ids, b->d_wide_idx, b->stream);
		// kernel D's launch shape and pool size: worked out once per batch (device queries cost as much as a small launch)
		if (!b->deep_cfg) {
			hipDeviceProp_t prop;
			HIPCHK(hipGetDeviceProperties(&prop, b->ix->device));
			uint32_t K = (uint32_t)env_int("NABWA_DEEP_STAGE", (int)DEEP_STAGE_MAX);
			if (K < 1u) K = 1u;
			if (K > DEEP_STAGE_MAX) K = DEEP_STAGE_MAX;
			// the read's own data (bound bytes, seed bound bytes, both strands' bases) sits in the wave's LDS when it is small enough
			const uint32_t rd_pl = align_up((uint32_t)(b->max_len > 0 ? b->max_len : 1), 16);
			uint32_t lds_rd = 2u * b->P.WLB + 2u * b->P.SLB + 2u * rd_pl;
			if (lds_rd > (uint32_t)env_int("NABWA_DEEP_LDS_MAX", 6144)) lds_rd = 0;
			int occ = nabwa_deep_occupancy((int)NS, (int)lds_rd);
			if (occ < 1) occ = 1;
			if (env_int("NABWA_DEEP_WAVES_PER_CU", 0) > 0) occ = env_int("NABWA_DEEP_WAVES_PER_CU", 0);
			size_t budget = (size_t)env_int("NABWA_DEEP_GB", 32) << 30;
			{
				size_t fr = 0, tot = 0;
				HIPCHK(hipMemGetInfo(&fr, &tot));
				size_t avail = fr;
				{ std::lock_guard<std::mutex> lk(b->ix->pool->mu); avail += b->ix->pool->idle_bytes; }
				avail = avail > ((size_t)6 << 30) ? avail - ((size_t)6 << 30) : ((size_t)64 << 20);
				if (budget > avail) budget = avail;
			}
			if (getenv("NABWA_DEEP_PAGES")) budget = (size_t)env_int("NABWA_DEEP_PAGES", 64) * ((size_t)DEEP_PAGE * 16 + 4);     /* (tests: a pool that runs dry) */
			size_t n_pages = budget / ((size_t)DEEP_PAGE * 16 + 4);
			if (n_pages > 0xfffffff0ull) n_pages = 0xfffffff0ull;
			if (n_pages < 2) n_pages = 2;
			// pages one search can hold at most: its live entries are bounded by the cut-off (bwtgap.c:140) plus one round's
			// children, and every score level may have a partly filled page
			uint64_t cap_pages = ((uint64_t)(b->opt.max_entries > 0 ? b->opt.max_entries : 0) + 9ull * 64ull * K + 2) / DEEP_PAGE + NS + 4;
			if (cap_pages > n_pages) cap_pages = n_pages;
			b->deep_K = K; b->deep_lds_rd = lds_rd; b->deep_rd_pl = rd_pl; b->deep_n_pages = n_pages; b->deep_cap_pages = cap_pages;
			b->deep_waves_max = (long)prop.multiProcessorCount * occ;
			b->deep_cfg = 1;
		}
		const uint32_t K = b->deep_K, lds_rd = b->deep_lds_rd, rd_pl = b->deep_rd_pl;
		const size_t n_pages = b->deep_n_pages; const uint64_t cap_pages = b->deep_cap_pages;
		long n_waves = b->deep_waves_max;
		if (n_waves > (long)cur) n_waves = (long)cur;
		if (b->deep_pages < n_pages) {
			if (b->d_pages) { HIPCHK(pool_free(b->ix, b->d_pages)); HIPCHK(pool_free(b->ix, b->d_page_prev)); b->d_pages = 0; b->d_page_prev = 0; b->deep_pages = 0; }
			HIPCHK(pool_malloc(b->ix, (void**)&b->d_pages, n_pages * DEEP_PAGE * 16)); HIPCHK(pool_malloc(b->ix, (void**)&b->d_page_prev, n_pages * 4));
			b->deep_pages = n_pages;
		}
		const size_t own_words = (size_t)n_waves * 2 * cap_pages, stage_ent = (size_t)n_waves * 64 * K * 4;
		if (b->deep_own_words < own_words) {
			if (b->d_deep_own) HIPCHK(pool_free(b->ix, b->d_deep_own));
			b->d_deep_own = 0; b->deep_own_words = 0;
			HIPCHK(pool_malloc(b->ix, (void**)&b->d_deep_own, own_words * 4)); b->deep_own_words = own_words;
		}
		if (b->deep_stage_ent < stage_ent) {
			if (b->d_deep_stage) HIPCHK(pool_free(b->ix, b->d_deep_stage));
			b->d_deep_stage = 0; b->deep_stage_ent = 0;
			HIPCHK(pool_malloc(b->ix, (void**)&b->d_deep_stage, stage_ent * 16)); b->deep_stage_ent = stage_ent;
		}
		if (!b->d_deep_ctr) HIPCHK(pool_malloc(b->ix, (void**)&b->d_deep_ctr, 1024));
		DeepParams D;
		memset(&D, 0, sizeof(D));
		D.S = b->P;
		D.S.ids = b->d_ovf_ids; D.S.n_sync = 0; D.S.w_sync = 0; D.S.res_slot = b->d_wide_idx;
		D.S.n_aln = b->d_naln2; D.S.max_ent = b->d_maxent2; D.S.status = b->d_status2; D.S.aln = b->d_aln2; D.S.aln_cap = b->aln_cap2;
		D.pages = b->d_pages; D.page_prev = b->d_page_prev; D.n_pages = (uint32_t)n_pages;
		D.page_bump = (unsigned int*)(b->d_deep_ctr + 8);
		D.own = b->d_deep_own; D.stage = b->d_deep_stage; D.stage_k = K; D.NS = NS; D.lds_rd = lds_rd; D.rd_pl = rd_pl;
		D.careful_all = env_int("NABWA_DEEP_CAREFUL", 0); D.max_lanes = env_int("NABWA_DEEP_LANES", 64);
		/* key-form entries (fm_deep.hpp): both indexes carry interval tables of one depth, and no row number reaches the form's marker;
		 * NABWA_DEEP_KEYFORM=0 keeps every entry as rows (A/B runs, and what the touch-counting run does anyway) */
		{
			const DevBwt &B0 = b->P.bwt[0], &B1 = b->P.bwt[1];
			const bool ok = B0.kmer_T > 0 && B0.kmer_T == B1.kmer_T && B0.kmer_LW == B0.kmer_T && B1.kmer_LW == B1.kmer_T && B0.kmer_lo && B1.kmer_lo &&
							B0.seq_len < DEEP_KEYL - 1u && B1.seq_len < DEEP_KEYL - 1u && (b->P.text_mode & 4) && env_int("NABWA_DEEP_KEYFORM", 1);
			D.key_T = ok ? B0.kmer_T : 0u;
		}
		if (D.max_lanes < 1) D.max_lanes = 1;
		if (D.max_lanes > 64) D.max_lanes = 64;

		D.stats = timing || getenv("NABWA_DEEP_STATS") ? b->d_deep_ctr : 0;
		D.hist = env_int("NABWA_DEEP_HIST", 0);
		/* the wave-wide expansion of one-row chains pays where chains are long: reads of 100 bases and more (PE D -24 %); on reads of 50-76 bases
		 * its chains end after a level or two, and the kernel built without it is the faster one (profiles/r03_deep_variants.txt) */
		D.coop_lanes = (uint32_t)env_int("NABWA_DEEP_COOP", b->max_len >= 90 ? 4 : 0);
		/* NABWA_DEEP_DUMP=<file> (investigations of the work order): per search of the first launch its read, length, max_diff, the width
		 * passes' restart classes, what kernel S saw of it (trips, hits) and the rounds kernel D needed -- int32 x 8 per search */
		const char *dump_path = getenv("NABWA_DEEP_DUMP");
		std::vector<int32_t> dump_ids, dump_trips, dump_naln; uint32_t *d_rounds = 0;
		if (dump_path) {
			D.stats = b->d_deep_ctr;
			dump_ids.resize(cur); dump_trips.resize(b->n); dump_naln.resize(b->n);
			HIPCHK(hipStreamSynchronize(b->stream));
			HIPCHK(hipMemcpy(dump_ids.data(), b->d_ovf_ids, (size_t)cur * 4, hipMemcpyDeviceToHost));
			HIPCHK(hipMemcpy(dump_trips.data(), b->d_maxent, (size_t)b->n * 4, hipMemcpyDeviceToHost));
			HIPCHK(hipMemcpy(dump_naln.data(), b->d_naln, (size_t)b->n * 4, hipMemcpyDeviceToHost));
			HIPCHK(pool_malloc(b->ix, (void**)&d_rounds, (size_t)cur * 4));
			HIPCHK(hipMemsetAsync(d_rounds, 0, (size_t)cur * 4, b->stream));
			D.rounds_out = d_rounds;
		}
		// pass 1: as many waves as fit the CUs, pages on demand; pass 2 (only if the pool ran dry under some reads): as many
		// waves as the pool can serve in the worst case
		unsigned int todo = cur, n_pool = 0;
		for (int pass = 0; pass < 2 && todo; ++pass) {
			long waves = n_waves, own_cap = (long)cap_pages;
			if (pass == 1) {
				waves = (long)(n_pages / cap_pages);
				if (waves < 1) waves = 1;
				if (waves > n_waves) waves = n_waves;
			}
			if (waves > (long)todo) waves = (long)todo;
			D.S.n = (int)todo; D.own_cap = (uint32_t)own_cap;
			rebuild_widths(D.S, todo, pass == 0);
			HIPCHK(hipMemsetAsync(b->d_counter, 0, 16, b->stream));
			HIPCHK(hipMemsetAsync(b->d_deep_ctr, 0, 1024, b->stream));
			D.S.work_counter = b->d_counter;
			if (pass == 0) HIPCHK(hipEventRecord(b->evd0, b->stream));
			nabwa_launch_fm_deep(&D, (int)waves, b->stream);
			if (pass == 0) HIPCHK(hipEventRecord(b->evd1, b->stream));
			nabwa_launch_scatter_wide((int)todo, b->d_ovf_ids, b->d_naln2, b->d_maxent2, b->d_status2,
									  b->d_naln, b->d_maxent, b->d_status, b->d_wide_idx, b->stream);
			HIPCHK(hipGetLastError());
			int r = recollect(NABWA_ST_POOL, &n_pool);
			if (r != NABWA_OK) return r;
			if (timing) {
				unsigned long long st[128];
				HIPCHK(hipMemcpy(st, b->d_deep_ctr, 1024, hipMemcpyDeviceToHost));
				fprintf(stderr, "[nabwa] kernel D%s: %u reads on %ld waves (%zu pages of 4 KB, %u handed out), %u left for the guaranteed pass, %.3f s; rounds %llu, chains run %llu / committed %llu, wave-steps %llu, careful rounds %llu, exact tails: %llu rank steps, %llu finished by text; longest read %.3f s / %llu rounds, all reads %.1f wave-s, longest wave %.3f s\n",
						pass ? " (guaranteed pass)" : "", todo, waves, n_pages, (unsigned int)(st[8] & 0xffffffffu), n_pool, now() - tt0, st[0], st[1], st[2], st[3], st[4], st[6], st[7], st[10] * 1e-8, st[11], st[12] * 1e-8, st[13] * 1e-8);
				fprintf(stderr, "[nabwa] kernel D phases (wave-s): pop %.1f, chains %.1f, exact tails %.1f (%llu turns), commit %.1f, hit bookkeeping %.1f; active lanes per chain step %.1f\n",
						st[16] * 1e-8, st[17] * 1e-8, st[18] * 1e-8, st[21], st[19] * 1e-8, st[20] * 1e-8, st[3] ? (double)st[22] / (double)st[3] : 0.0);
				fprintf(stderr, "[nabwa] kernel D lane-steps %llu: pruned at the pop %llu, expansions %llu (in key form %llu, on two buckets %llu), records %llu, children stored %llu; key-form tails / hits %llu; expansions without a difference allowed: %llu in key form, %llu on one row, %llu on several\n",
						st[22], st[27], st[28], st[23], st[29], st[25], st[26], st[24], st[30], st[31], st[9]);
				if (env_int("NABWA_DEEP_HIST", 0) == 2) {
					fprintf(stderr, "[nabwa] kernel D rounds by width (1, 2, 3-4, 5-8, 9-16, 17-32, 33-64 entries):");
					for (int d = 0; d < 7; ++d) fprintf(stderr, " %llu", st[32 + d]);
					fprintf(stderr, "; their wave-steps:");
					for (int d = 0; d < 7; ++d) fprintf(stderr, " %llu", st[64 + d]);
					fprintf(stderr, "\n");
				}
				if (env_int("NABWA_DEEP_HIST", 0) == 1) for (int h = 0; h < 3; ++h) {
					fprintf(stderr, "[nabwa] kernel D expansions by depth (read symbols consumed), %s:", h == 0 ? "rows, several" : (h == 1 ? "rows, one" : "key form"));
					for (int d = 0; d < 32; ++d) fprintf(stderr, " %llu", st[32 + 32 * h + d]);
					fprintf(stderr, "\n");
				}
			}
			if (pass == 0 && dump_path) {
				std::vector<uint32_t> rounds(dump_ids.size());
				std::vector<uint8_t> cls((size_t)b->n * 2), md((size_t)b->n); std::vector<int32_t> lens((size_t)b->n);
				HIPCHK(hipMemcpy(rounds.data(), d_rounds, rounds.size() * 4, hipMemcpyDeviceToHost));
				if (b->d_cls) HIPCHK(hipMemcpy(cls.data(), b->d_cls, cls.size(), hipMemcpyDeviceToHost));
				HIPCHK(hipMemcpy(md.data(), b->d_md, md.size(), hipMemcpyDeviceToHost));
				HIPCHK(hipMemcpy(lens.data(), b->d_len, lens.size() * 4, hipMemcpyDeviceToHost));
				FILE *f = fopen(dump_path, "wb");
				if (f) {
					for (size_t t = 0; t < dump_ids.size(); ++t) {
						const int32_t r = dump_ids[t];
						const int32_t row[8] = { r, lens[r], (int32_t)md[r], (int32_t)cls[2 * (size_t)r], (int32_t)cls[2 * (size_t)r + 1], dump_trips[r], dump_naln[r], (int32_t)rounds[t] };
						fwrite(row, 4, 8, f);
					}
					fclose(f);
				}
				{	/* <file>.all: per read of the batch its two restart classes and whether kernel S handed it on */
					std::vector<uint8_t> st((size_t)b->n);
					HIPCHK(hipMemcpy(st.data(), b->d_status, st.size(), hipMemcpyDeviceToHost));
					FILE *g = fopen((std::string(dump_path) + ".all").c_str(), "wb");
					if (g) { for (int i = 0; i < b->n; ++i) { const uint8_t row[4] = { cls[2 * (size_t)i], cls[2 * (size_t)i + 1], (uint8_t)(st[i] != NABWA_ST_OK), md[i] }; fwrite(row, 1, 4, g); } fclose(g); }
				}
				HIPCHK(pool_free(b->ix, d_rounds)); D.rounds_out = 0;
			}
			todo = n_pool;
		}
		b->deep_ran = 1;
		HIPCHK(hipEventElapsedTime(&b->last_ms_deep, b->evd0, b->evd1));
		if (todo) { b->unresolved = (int)todo; return fail(NABWA_ENOMEM, "kernel D: the page pool cannot hold one worst-case search (raise NABWA_DEEP_GB or lower max_entries)"); }
		unsigned int n_hit = 0;
		int r = recollect(NABWA_ST_HITCAP, &n_hit);
		if (r != NABWA_OK) return r;
		/* A hit list that outgrew the wide rows: the reference's list grows without bound (bwtgap.c:186-190), so those searches run
		 * again with 16 x the rows, then 256 x ... in a block of their own (NABWA_HIT_GROW steps, NABWA_HIT_GROW_GB at most); a table
		 * on the device names the rows of every read resolved that way (status NABWA_ST_GROWN, wide_idx = its slot there). */
		size_t cap3 = (size_t)b->aln_cap2;
		for (int grow = 0; n_hit && grow < env_int("NABWA_HIT_GROW", 3) && b->n_grown < 8; ++grow) {
			cap3 *= 16;
			const size_t bytes = (size_t)n_hit * cap3 * 16;
			if (bytes > ((size_t)env_int("NABWA_HIT_GROW_GB", 8) << 30) || cap3 > 0x7fffffffu) break;
			uint8_t *raw = 0; int32_t *n3 = 0, *m3 = 0; uint8_t *s3 = 0;
			HIPCHK(pool_malloc(b->ix, (void**)&raw, bytes));
			b->grown[b->n_grown++] = raw;
			HIPCHK(pool_malloc(b->ix, (void**)&n3, (size_t)n_hit * 4)); HIPCHK(pool_malloc(b->ix, (void**)&m3, (size_t)n_hit * 4)); HIPCHK(pool_malloc(b->ix, (void**)&s3, n_hit));
			uint8_t *const base = raw;
			if (!b->d_grown_tab || (b->grown_used == 0 && b->grown_cap < 8 * (int)n_hit + 8)) {      /* every step resolves or repeats reads of the first step's list */
				if (b->d_grown_tab) { HIPCHK(hipStreamSynchronize(b->stream)); HIPCHK(pool_free(b->ix, (void*)b->d_grown_tab)); b->d_grown_tab = 0; }
				b->grown_cap = 8 * (int)n_hit + 8;
				HIPCHK(pool_malloc(b->ix, (void**)&b->d_grown_tab, (size_t)b->grown_cap * 8));
				b->grown_used = 0;
			}
			if (b->grown_used + (int)n_hit > b->grown_cap) break;
			DeepParams G = D;
			G.S.res_slot = 0; G.S.n_aln = n3; G.S.max_ent = m3; G.S.status = s3; G.S.aln = (uint4*)base; G.S.aln_cap = (int)cap3;
			G.rounds_out = 0;
			long waves = (long)(n_pages / cap_pages);          /* as many searches at a time as the pool can hold in the worst case */
			if (waves < 1) waves = 1;
			if (waves > n_waves) waves = n_waves;
			if (waves > (long)n_hit) waves = (long)n_hit;
			G.S.n = (int)n_hit; G.own_cap = (uint32_t)cap_pages;
			rebuild_widths(G.S, n_hit, false);
			HIPCHK(hipMemsetAsync(b->d_counter, 0, 16, b->stream));
			HIPCHK(hipMemsetAsync(b->d_deep_ctr, 0, 1024, b->stream));
			G.S.work_counter = b->d_counter;
			nabwa_launch_fm_deep(&G, (int)waves, b->stream);
			nabwa_launch_scatter_grown((int)n_hit, b->d_ovf_ids, n3, m3, s3, b->d_naln, b->d_maxent, b->d_status, b->d_wide_idx, (const uint4*)base, cap3, b->d_grown_tab, b->grown_used, b->stream);
			b->grown_used += (int)n_hit;
			HIPCHK(hipGetLastError());
			HIPCHK(hipStreamSynchronize(b->stream));
			HIPCHK(pool_free(b->ix, n3)); HIPCHK(pool_free(b->ix, m3)); HIPCHK(pool_free(b->ix, s3));
			if (timing) fprintf(stderr, "[nabwa] kernel D, hit lists beyond %d rows: %u reads searched again with %zu rows each\n", b->aln_cap2, n_hit, cap3);
			r = recollect(NABWA_ST_HITCAP, &n_hit);
			if (r != NABWA_OK) return r;
		}
		b->unresolved = (int)n_hit;
		if (n_hit) return fail(NABWA_EHITS, "reads with more hit rows than the grown lists hold (NABWA_ALNCAP2 x 16^NABWA_HIT_GROW within NABWA_HIT_GROW_GB): their n_aln is reported as 0, every other read is resolved");
	}
	return NABWA_OK;
}

extern "C" float nabwa_batch_last_kernel_ms(nabwa_batch_t *b) { return b ? b->last_ms : 0.f; }
extern "C" float nabwa_batch_last_width_ms(nabwa_batch_t *b) { return b ? b->last_ms_w : 0.f; }
extern "C" float nabwa_batch_last_deep_ms(nabwa_batch_t *b) { return b ? b->last_ms_deep : 0.f; }

/* One extra, untimed run of both passes with the instrumented kernel: total Occ-bucket touches the
 * REFERENCE algorithm performs on this batch (the "algorithmic bytes" of the roofline are 48 B each). */
extern "C" int nabwa_batch_count_touches(nabwa_batch_t *b, uint64_t *n_bucket, uint64_t *n_bucket_width)
{
	if (!b || !n_bucket) return fail(NABWA_EINVAL, "null argument");
	HIPCHK(hipSetDevice(b->ix->device));
	HIPCHK(hipMemsetAsync(b->d_sum, 0, 256, b->stream));
	b->P.touch_counter = b->d_sum;
	/* the reference walks every exact tail row by row: count with the tail jump off (NABWA_TRIP_STATS=jump keeps it
	 * on to profile the production trips; the touch totals are then not the reference's) */
	const uint32_t kt0 = b->P.bwt[0].kmer_T, kt1 = b->P.bwt[1].kmer_T; const int tm0 = b->P.text_mode;
	const char *ts = getenv("NABWA_TRIP_STATS");
	if (!(ts && strcmp(ts, "jump") == 0)) { b->P.bwt[0].kmer_T = b->P.bwt[1].kmer_T = 0; b->P.text_mode = 0; }
	int r = nabwa_batch_run(b);
	if (r == NABWA_OK) r = nabwa_batch_sync(b, 0);
	b->P.touch_counter = 0;
	b->P.bwt[0].kmer_T = kt0; b->P.bwt[1].kmer_T = kt1; b->P.text_mode = tm0;
	if (r != NABWA_OK) return r;
	unsigned long long v[2] = { 0, 0 };
	HIPCHK(hipMemcpy(v, b->d_sum, 16, hipMemcpyDeviceToHost));
	*n_bucket = v[0];                          /* search kernel (bwt_match_gap) */
	if (n_bucket_width) *n_bucket_width = v[1];  /* width kernel (bwt_cal_width) */
	if (getenv("NABWA_TRIP_STATS") && b->class_sort) {
		unsigned int c[12];
		HIPCHK(hipMemcpy(c, b->d_ncls, 48, hipMemcpyDeviceToHost));
		fprintf(stderr, "[nabwa] read classes by restarts 0 / 1 / 2+: %u / %u / %u\n", c[0], c[1], c[2]);
	}
	if (getenv("NABWA_TRIP_STATS")) {
		unsigned long long t[32];
		HIPCHK(hipMemcpy(t, b->d_sum, 256, hipMemcpyDeviceToHost));
		fprintf(stderr, "[nabwa] search kernel: wave-trips %llu; lane-trips: expand %llu exact %llu entry-load %llu spec %llu query %llu two-bucket %llu exited %llu tail-jump %llu text-expand %llu text-tail %llu; longest read %llu trips, %llu reads over 2000 trips, %llu over 500\n",
				t[2], t[3], t[4], t[5], t[6], t[7], t[8], t[9], t[10], t[11], t[12], t[13], t[14], t[15]);
		fprintf(stderr, "[nabwa] the %llu reads over 8000 trips: %llu trips = expansions key-form %llu, rows two-bucket %llu, rows one-bucket %llu, text %llu (of all: %llu with gaps); pops %llu, tail steps %llu, jumps %llu\n",
				t[16], t[17], t[18], t[19], t[20], t[21], t[25], t[22], t[23], t[24]);
	}
	return NABWA_OK;
}

/* Tests: the records kernel W writes for reads [first, first + n), unpacked: per read and strand the len + 1 interval widths
 * and lower bounds of the full pass (bwt_cal_width, bwtaln.c:52-76, 123-124) and the seed_len + 1 bounds of the seed pass
 * (:126-130; only for reads longer than the seed).  Runs kernel W alone on a fresh record.  Rows are max_len + 1 wide. */
extern "C" int nabwa_batch_width_records(nabwa_batch_t *b, int first, int n, uint32_t *w_out, uint8_t *bid_out, uint8_t *seed_bid_out)
{
	if (!b || first < 0 || n < 0 || first + n > b->n || (n && (!w_out || !bid_out))) return fail(NABWA_EINVAL, "bad argument");
	if (n == 0) return NABWA_OK;
	HIPCHK(hipSetDevice(b->ix->device));
	HIPCHK(hipMemsetAsync(b->d_counter, 0, 16, b->stream));
	SearchParams PW = b->P; PW.ids = 0; PW.rd_cls = 0; PW.touch_counter = 0;
	nabwa_launch_fm_width(&PW, b->n_blocks_w, b->stream);
	HIPCHK(hipGetLastError());
	std::vector<uint8_t> rec((size_t)n * b->P.wstride);
	HIPCHK(hipMemcpyAsync(rec.data(), b->d_wdata + (size_t)first * b->P.wstride, rec.size(), hipMemcpyDeviceToHost, b->stream));
	HIPCHK(hipStreamSynchronize(b->stream));
	const int W = b->max_len + 1, SW = b->opt.seed_len + 1;
	for (int i = 0; i < n; ++i) {
		const uint8_t *r = rec.data() + (size_t)i * b->P.wstride;
		for (int x = 0; x < 2; ++x) {
			memcpy(w_out + ((size_t)i * 2 + x) * W, (const uint32_t*)r + x * b->P.WL, 4 * (size_t)W);
			memcpy(bid_out + ((size_t)i * 2 + x) * W, r + b->P.woff_bid + x * b->P.WLB, (size_t)W);
			if (seed_bid_out && b->max_len > b->opt.seed_len && SW <= (int)b->P.SLB) memcpy(seed_bid_out + ((size_t)i * 2 + x) * SW, r + b->P.woff_sbid + x * b->P.SLB, (size_t)SW);
		}
	}
	return NABWA_OK;
}

extern "C" int nabwa_batch_checksum(nabwa_batch_t *b, uint64_t *sum, int64_t *n_rows)
{
	if (!b) return fail(NABWA_EINVAL, "null batch");
	HIPCHK(hipSetDevice(b->ix->device));
	unsigned long long h[2] = { 0, 0 };
	HIPCHK(hipMemsetAsync(b->d_sum, 0, 16, b->stream));
	nabwa_launch_checksum(b->n, b->d_naln, b->d_aln, b->P.aln_cap, b->d_status, b->d_wide_idx, b->d_aln2, b->aln_cap2, b->d_grown_tab,
						  b->d_sum, b->d_sum + 1, b->stream);
	HIPCHK(hipMemcpyAsync(h, b->d_sum, 16, hipMemcpyDeviceToHost, b->stream));
	HIPCHK(hipStreamSynchronize(b->stream));
	if (sum) *sum = h[0];
	if (n_rows) *n_rows = (int64_t)h[1];
	return NABWA_OK;
}

extern "C" int nabwa_batch_fetch(nabwa_batch_t *b, int32_t *n_aln, nabwa_aln1_t *aln_out, int64_t aln_cap, int64_t *n_rows,
								 int32_t *max_entries)
{
	if (!b || !n_aln) return fail(NABWA_EINVAL, "null argument");
	HIPCHK(hipSetDevice(b->ix->device));
	if (n_rows) *n_rows = 0;
	if (b->n == 0) return NABWA_OK;
	// device-side compaction: exclusive scan of n_aln, then gather rows
	uint32_t *d_off = 0; void *d_tmp = 0; size_t tmp_bytes = 0; uint4 *d_rows = 0;
	HIPCHK(pool_malloc(b->ix, (void**)&d_off, (size_t)(b->n + 1) * 4));
	HIPCHK(hipcub::DeviceScan::ExclusiveSum(nullptr, tmp_bytes, (const uint32_t*)b->d_naln, d_off, b->n, b->stream));
	HIPCHK(pool_malloc(b->ix, (void**)&d_tmp, tmp_bytes ? tmp_bytes : 16));
	HIPCHK(hipcub::DeviceScan::ExclusiveSum(d_tmp, tmp_bytes, (const uint32_t*)b->d_naln, d_off, b->n, b->stream));
	uint32_t last_off = 0; int32_t last_n = 0;
	HIPCHK(hipMemcpyAsync(&last_off, d_off + (b->n - 1), 4, hipMemcpyDeviceToHost, b->stream));
	HIPCHK(hipMemcpyAsync(&last_n, b->d_naln + (b->n - 1), 4, hipMemcpyDeviceToHost, b->stream));
	HIPCHK(hipMemcpyAsync(n_aln, b->d_naln, (size_t)b->n * 4, hipMemcpyDeviceToHost, b->stream));
	if (max_entries) HIPCHK(hipMemcpyAsync(max_entries, b->d_maxent, (size_t)b->n * 4, hipMemcpyDeviceToHost, b->stream));
	HIPCHK(hipStreamSynchronize(b->stream));
	const int64_t total = (int64_t)last_off + last_n;
	if (n_rows) *n_rows = total;
	int rc = NABWA_OK;
	if (total > aln_cap || (total && !aln_out)) rc = fail(NABWA_ECAP, "aln_cap too small");
	else if (total) {
		HIPCHK(pool_malloc(b->ix, (void**)&d_rows, (size_t)total * 16));
		nabwa_launch_gather(b->n, b->d_naln, d_off, b->d_aln, b->P.aln_cap, b->d_status, b->d_wide_idx, b->d_aln2, b->aln_cap2, b->d_grown_tab,
							d_rows, b->stream);
		HIPCHK(hipMemcpyAsync(aln_out, d_rows, (size_t)total * 16, hipMemcpyDeviceToHost, b->stream));
		HIPCHK(hipStreamSynchronize(b->stream));
		HIPCHK(pool_free(b->ix, d_rows));
	}
	HIPCHK(pool_free(b->ix, d_off)); HIPCHK(pool_free(b->ix, d_tmp));
	return rc;
}

extern "C" int nabwa_cal_sa_reg_gap(nabwa_index_t *ix, const nabwa_gap_opt_t *opt, int n, const int64_t *off,
									const uint8_t *seq, const uint8_t *rseq, int per_read,
									int32_t *n_aln, nabwa_aln1_t *aln_out, int64_t aln_cap, int64_t *n_rows,
									int32_t *max_entries)
{
	nabwa_batch_t *b = 0;
	const bool timing = getenv("NABWA_TIMING") != 0;
	auto now = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
	const double t0 = now();
	if (n_rows) *n_rows = 0;
	int r = nabwa_batch_create(ix, opt, n, off, seq, rseq, per_read, &b);
	if (r != NABWA_OK) return r;
	const double t1 = now();
	r = nabwa_batch_run(b);
	if (r == NABWA_OK) r = nabwa_batch_sync(b, 0);
	const double t2 = now();
	if (r == NABWA_OK) r = nabwa_batch_fetch(b, n_aln, aln_out, aln_cap, n_rows, max_entries);
	else if (r == NABWA_EHITS) {      /* a few reads have more hit rows than NABWA_ALNCAP2: every other read's result is still handed out */
		const std::string msg = g_err;
		const int r2 = nabwa_batch_fetch(b, n_aln, aln_out, aln_cap, n_rows, max_entries);
		if (r2 != NABWA_OK) r = r2; else g_err = msg;
	}
	const double t3 = now();
	nabwa_batch_destroy(b);
	if (timing) fprintf(stderr, "[nabwa] cal_sa_reg_gap %d reads: upload + layout %.3f s, kernels %.3f s, compaction + download %.3f s, release %.3f s\n",
						n, t1 - t0, t2 - t1, t3 - t2, now() - t3);
	return r;
}

/* ------------------------------------------------------------------ bwt_sa / occ batches */

extern "C" int nabwa_sa_lookup(nabwa_index_t *ix, int n, const uint8_t *which, const uint32_t *k, uint32_t *sa_out)
{
	if (!ix || n < 0 || (n && (!which || !k || !sa_out))) return fail(NABWA_EINVAL, "null argument");
	if (!ix->bwt[0].sa || !ix->bwt[1].sa) return fail(NABWA_EINVAL, "index was loaded without suffix arrays");
	if (n == 0) return NABWA_OK;
	HIPCHK(hipSetDevice(ix->device));
	uint8_t *dw = 0; uint32_t *dk = 0, *dout = 0;
	/* (buffers from the pool kept with the index: this is called once per batch by the finishing chains) */
	HIPCHK(pool_malloc(ix, (void**)&dw, (size_t)n)); HIPCHK(pool_malloc(ix, (void**)&dk, (size_t)n * 4)); HIPCHK(pool_malloc(ix, (void**)&dout, (size_t)n * 4));
	HIPCHK(hipMemcpy(dw, which, n, hipMemcpyHostToDevice));
	HIPCHK(hipMemcpy(dk, k, (size_t)n * 4, hipMemcpyHostToDevice));
	nabwa_launch_sa_lookup(ix->bwt, n, dw, dk, dout, 0);
	HIPCHK(hipGetLastError());
	HIPCHK(hipMemcpy(sa_out, dout, (size_t)n * 4, hipMemcpyDeviceToHost));
	HIPCHK(pool_free(ix, dw)); HIPCHK(pool_free(ix, dk)); HIPCHK(pool_free(ix, dout));
	return NABWA_OK;
}

extern "C" int nabwa_occ4(nabwa_index_t *ix, int which, int n, const uint32_t *k, uint32_t *cnt_out)
{
	if (!ix || n < 0 || (n && (!k || !cnt_out))) return fail(NABWA_EINVAL, "null argument");
	if (n == 0) return NABWA_OK;
	HIPCHK(hipSetDevice(ix->device));
	uint32_t *dk = 0, *dout = 0;
	HIPCHK(hipMalloc(&dk, (size_t)n * 4)); HIPCHK(hipMalloc(&dout, (size_t)n * 16));
	HIPCHK(hipMemcpy(dk, k, (size_t)n * 4, hipMemcpyHostToDevice));
	nabwa_launch_occ4(&ix->bwt[which & 1], n, dk, dout, 0);
	HIPCHK(hipGetLastError());
	HIPCHK(hipMemcpy(cnt_out, dout, (size_t)n * 16, hipMemcpyDeviceToHost));
	HIPCHK(hipFree(dk)); HIPCHK(hipFree(dout));
	return NABWA_OK;
}
