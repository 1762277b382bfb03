// wire_worker.cpp -- the network half of the drop-in: the record `bwa bam2bam` and `bwa worker` exchange, and a worker
// core over any transport.  Host code on top of the library's own C ABI (include/nabwa.h); no socket library is needed or used --
// libzmq is absent from the image, INTEGRATION.md shows the few zmq_msg_* lines that go around these calls.
//
// Reference: msg_init_from_pair / pair_init_from_msg (bam2bam.c:951-1097), run_config_service (:1238-1286), run_worker_thread
// (:1387-1442), handle_broadcast / bwa_worker_core (:2079-2176), bwa_worker (:2213-2309).
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>
#include "../../include/nabwa.h"
#include "nabwa_internal.hpp"

namespace {
struct Writer {
	uint8_t *p;
	void u8(uint32_t v) { *p++ = (uint8_t)v; }
	void u32(uint32_t v) { for (int k = 0; k < 4; ++k) *p++ = (uint8_t)(v >> (8 * k)); }
	void u64(uint64_t v) { u32((uint32_t)v); u32((uint32_t)(v >> 32)); }
	void bytes(const void *q, size_t n) { if (n) memcpy(p, q, n); p += n; }
};
struct Reader {
	const uint8_t *p, *end; bool ok;
	bool room(size_t n) { if ((size_t)(end - p) < n) { ok = false; return false; } return true; }
	uint32_t u8() { return room(1) ? *p++ : 0; }
	uint32_t u32() { if (!room(4)) return 0; uint32_t v = 0; for (int k = 0; k < 4; ++k) v |= (uint32_t)p[k] << (8 * k); p += 4; return v; }
	uint64_t u64() { const uint64_t lo = u32(), hi = u32(); return hi << 32 | lo; }
	const uint8_t *bytes(size_t n) { if (!room(n)) return 0; const uint8_t *q = p; p += n; return q; }
};
inline bool has_positioned(int phase) { return phase == NABWA_PHASE_POSITIONED; }
inline bool has_aligned(int phase) { return phase == NABWA_PHASE_ALIGNED || phase == NABWA_PHASE_POSITIONED; }
inline bool good_header(const nabwa_wire_rec_t *r) { return r->kind <= NABWA_KIND_PAIR && r->phase <= NABWA_PHASE_FINISHED; }
}

extern "C" int64_t nabwa_wire_size(const nabwa_wire_rec_t *r)
{
	if (!r || !good_header(r)) return NABWA_EINVAL;
	int64_t n = 10;
	for (int i = 0; i < r->kind; ++i) {
		const nabwa_wire_read_t &x = r->read[i];
		n += 32 + 4 + (int64_t)x.data_len;
		if (has_positioned(r->phase)) n += 38 + 16 * (int64_t)x.n_multi;
		if (has_aligned(r->phase)) n += 8 + 16 * (int64_t)x.n_aln;
	}
	return n;
}

extern "C" int64_t nabwa_wire_encode(const nabwa_wire_rec_t *r, uint8_t *out, int64_t cap)
{
	const int64_t need = nabwa_wire_size(r);
	if (need < 0) return nabwa_fail(NABWA_EINVAL, "a record kind / phase that does not exist");
	for (int i = 0; i < r->kind; ++i) {
		const nabwa_wire_read_t &x = r->read[i];
		if (x.data_len < 0 || (x.data_len && !x.data) || x.n_multi < 0 || x.n_aln < 0 ||
			(has_positioned(r->phase) && x.n_multi && !x.multi) || (has_aligned(r->phase) && x.n_aln && !x.aln)) return nabwa_fail(NABWA_EINVAL, "a read with a count but no data");
	}
	if (!out || cap < need) return NABWA_ECAP;
	Writer w = { out };
	w.u64(r->recno); w.u8(r->kind); w.u8(r->phase);
	for (int i = 0; i < r->kind; ++i) {
		const nabwa_wire_read_t &x = r->read[i];
		w.bytes(x.core, 32); w.u32((uint32_t)x.data_len); w.bytes(x.data, (size_t)x.data_len);
		if (has_positioned(r->phase)) {
			w.u8((uint32_t)(x.strand << 4 | x.type)); w.u8(x.n_mm); w.u8(x.n_gapo); w.u8(x.n_gape); w.u8(x.seQ); w.u8(x.mapQ);
			w.u32((uint32_t)x.len); w.u32((uint32_t)x.clip_len); w.u32((uint32_t)x.score); w.u32(x.sa); w.u32(x.c1); w.u32(x.c2); w.u32(x.pos);
			w.u32((uint32_t)x.n_multi); w.bytes(x.multi, 16 * (size_t)x.n_multi);
		}
		if (has_aligned(r->phase)) { w.u32((uint32_t)x.max_entries); w.u32((uint32_t)x.n_aln); w.bytes(x.aln, 16 * (size_t)x.n_aln); }
	}
	return need;
}

extern "C" int nabwa_wire_decode(const uint8_t *msg, int64_t len, nabwa_wire_rec_t *out)
{
	if (!msg || !out || len < 10) return nabwa_fail(NABWA_EINVAL, "a message shorter than its fixed part");
	memset(out, 0, sizeof *out);
	Reader rd = { msg, msg + len, true };
	out->recno = rd.u64(); out->kind = (uint8_t)rd.u8(); out->phase = (uint8_t)rd.u8();
	if (!good_header(out)) return nabwa_fail(NABWA_EINVAL, "a record kind / phase that does not exist");
	for (int i = 0; i < out->kind && rd.ok; ++i) {
		nabwa_wire_read_t &x = out->read[i];
		const uint8_t *c = rd.bytes(32);
		if (c) memcpy(x.core, c, 32);
		x.data_len = (int32_t)rd.u32();
		if (x.data_len < 0) { rd.ok = false; break; }
		x.data = rd.bytes((size_t)x.data_len);
		if (has_positioned(out->phase)) {
			const uint32_t st = rd.u8();
			x.strand = (uint8_t)(st >> 4); x.type = (uint8_t)(st & 3);      /* the reference stores the whole byte into a two-bit field (bam2bam.c:1059) */
			x.n_mm = (uint8_t)rd.u8(); x.n_gapo = (uint8_t)rd.u8(); x.n_gape = (uint8_t)rd.u8(); x.seQ = (uint8_t)rd.u8(); x.mapQ = (uint8_t)rd.u8();
			x.len = (int32_t)rd.u32(); x.clip_len = (int32_t)rd.u32(); x.score = (int32_t)rd.u32(); x.sa = rd.u32(); x.c1 = rd.u32(); x.c2 = rd.u32(); x.pos = rd.u32();
			x.n_multi = (int32_t)rd.u32();
			if (x.n_multi < 0 || (int64_t)x.n_multi * 16 > len) { rd.ok = false; break; }
			x.multi = rd.bytes(16 * (size_t)x.n_multi);
		}
		if (has_aligned(out->phase)) {
			x.max_entries = (int32_t)rd.u32(); x.n_aln = (int32_t)rd.u32();
			if (x.n_aln < 0 || (int64_t)x.n_aln * 16 > len) { rd.ok = false; break; }
			x.aln = rd.bytes(16 * (size_t)x.n_aln);
		}
	}
	if (!rd.ok || rd.p != rd.end) return nabwa_fail(NABWA_EINVAL, "a message whose length does not fit its content");
	return NABWA_OK;
}

/* bam1_core_t in memory (bamlite.h:44-53) against the eight words of a BAM file's record: words 2 and 3 pack their fields the other
 * way round (the reader takes them apart, bamlite.c bam_read1: bin = x >> 16, qual = x >> 8 & 0xff, l_qname = x & 0xff; flag = x >> 16, n_cigar = x & 0xffff) */
extern "C" void nabwa_wire_core_from_bam(const uint8_t bam_core[32], uint8_t wire_core[32])
{
	uint32_t w[8]; memcpy(w, bam_core, 32);
	const uint32_t bin = w[2] >> 16, qual = w[2] >> 8 & 0xff, l_qname = w[2] & 0xff, flag = w[3] >> 16, n_cigar = w[3] & 0xffff;
	w[2] = bin | qual << 16 | l_qname << 24; w[3] = flag | n_cigar << 16;
	memcpy(wire_core, w, 32);
}
extern "C" void nabwa_wire_core_to_bam(const uint8_t wire_core[32], uint8_t bam_core[32])
{
	uint32_t w[8]; memcpy(w, wire_core, 32);
	const uint32_t bin = w[2] & 0xffff, qual = w[2] >> 16 & 0xff, l_qname = w[2] >> 24, flag = w[3] & 0xffff, n_cigar = w[3] >> 16;
	w[2] = bin << 16 | qual << 8 | l_qname; w[3] = flag << 16 | n_cigar;
	memcpy(bam_core, w, 32);
}

extern "C" int64_t nabwa_wire_config_encode(const nabwa_gap_opt_t *opt, const nabwa_pe_opt_t *popt, const char *prefix, uint8_t *out, int64_t cap)
{
	if (!opt || !popt || !prefix) return nabwa_fail(NABWA_EINVAL, "null argument");
	static_assert(sizeof(nabwa_gap_opt_t) == 64 && sizeof(nabwa_pe_opt_t) == 48, "gap_opt_t / pe_opt_t travel as they lie in memory (bam2bam.c:1260-1263)");
	const int64_t need = 64 + 48 + (int64_t)strlen(prefix);
	if (!out || cap < need) return NABWA_ECAP;
	memcpy(out, opt, 64); memcpy(out + 64, popt, 48); memcpy(out + 112, prefix, strlen(prefix));
	return need;
}
extern "C" int nabwa_wire_config_decode(const uint8_t *msg, int64_t len, nabwa_gap_opt_t *opt, nabwa_pe_opt_t *popt, char *prefix, int prefix_cap)
{
	if (!msg || !opt || !popt || !prefix) return nabwa_fail(NABWA_EINVAL, "null argument");
	if (len < 112) return nabwa_fail(NABWA_EINVAL, "a configuration reply shorter than the two option blocks");          /* the reference exits (bam2bam.c:2265-2268) */
	if (len - 112 + 1 > prefix_cap) return NABWA_ECAP;
	memcpy(opt, msg, 64); memcpy(popt, msg + 64, 48); memcpy(prefix, msg + 112, (size_t)(len - 112)); prefix[len - 112] = 0;
	return NABWA_OK;
}

// ---------------------------------------------------------------------------------------------------------------- the worker
struct nabwa_worker {
	nabwa_index_t *ix; nabwa_gap_opt_t opt; nabwa_pe_opt_t popt;
	uint64_t rng48;                                  /* the worker's own drand48 stream (srand48(bns->seed), bam2bam.c:2284) */
	nabwa_isize_table_t *isize;                      /* the estimates last received (g_iinfos), with finish_pair's position cache; 0 = none yet */
	nabwa_isize_table_t *scratch;                    /* pass 1 counts insert sizes into a table; in network mode that is the master's job */
	int64_t genome_len;
	uint64_t counts[4]; int failures;
};

extern "C" int nabwa_worker_create(nabwa_index_t *ix, const nabwa_gap_opt_t *opt, const nabwa_pe_opt_t *popt, nabwa_worker_t **out)
{
	if (!ix || !opt || !popt || !out) return nabwa_fail(NABWA_EINVAL, "null argument");
	int64_t l_pac = 0; uint32_t seed = 0;
	int rc = nabwa_index_reference_info(ix, &l_pac, &seed);
	if (rc != NABWA_OK) return rc;
	nabwa_worker *w = new nabwa_worker();
	w->ix = ix; w->opt = *opt; w->popt = *popt; w->genome_len = l_pac;
	w->rng48 = ((uint64_t)seed << 16) | 0x330E;
	w->isize = 0; w->scratch = nabwa_isize_table_create(popt->ap_prior, l_pac);
	memset(w->counts, 0, sizeof w->counts); w->failures = 0;
	*out = w;
	return NABWA_OK;
}
extern "C" void nabwa_worker_destroy(nabwa_worker_t *w)
{
	if (!w) return;
	if (w->isize) nabwa_isize_table_destroy(w->isize);
	nabwa_isize_table_destroy(w->scratch);
	delete w;
}
extern "C" void nabwa_worker_counts(const nabwa_worker_t *w, uint64_t out[4]) { if (w && out) memcpy(out, w->counts, sizeof w->counts); }

extern "C" int nabwa_worker_set_isize(nabwa_worker_t *w, const uint8_t *blob, int64_t n)
{
	if (!w || n < 0 || (n && !blob)) return nabwa_fail(NABWA_EINVAL, "null argument");
	if (n == 0) return NABWA_OK;                      /* an empty reply: the master has no estimates yet (bam2bam.c:2294-2299) */
	nabwa_isize_table_t *t = nabwa_isize_table_create(w->popt.ap_prior, w->genome_len);
	const int rc = nabwa_isize_table_decode(t, blob, n);
	if (rc != NABWA_OK) { nabwa_isize_table_destroy(t); return rc; }
	if (w->isize) nabwa_isize_table_destroy(w->isize);      /* (the reference keeps its position cache per thread over such an update; a new table starts a new one) */
	w->isize = t;
	return NABWA_OK;
}

namespace {
/* the records of a group of messages as one BAM stream, reads in the order of the messages */
struct Stream { std::vector<uint8_t> buf; std::vector<int64_t> off; };
void append_read(Stream &S, const nabwa_wire_read_t &x)
{
	const uint32_t bs = 32 + (uint32_t)x.data_len;
	const size_t at = S.buf.size();
	S.buf.resize(at + 4 + bs);
	memcpy(&S.buf[at], &bs, 4);
	nabwa_wire_core_to_bam(x.core, &S.buf[at + 4]);
	if (x.data_len) memcpy(&S.buf[at + 36], x.data, (size_t)x.data_len);
	S.off.push_back((int64_t)S.buf.size());
}
struct BatchGuard { nabwa_bam_batch_t *b; ~BatchGuard() { if (b) nabwa_bam_batch_destroy(b); } };
}

extern "C" int nabwa_worker_process(nabwa_worker_t *w, int n_msg, const uint8_t *const *msgs, const int64_t *lens, nabwa_send_fn send, void *ctx)
{
	if (!w || n_msg < 0 || (n_msg && (!msgs || !lens)) || !send) return nabwa_fail(NABWA_EINVAL, "null argument");
	std::vector<nabwa_wire_rec_t> rec((size_t)n_msg);
	std::vector<int> to_posn, to_finish;
	for (int m = 0; m < n_msg; ++m) {
		int rc = nabwa_wire_decode(msgs[m], lens[m], &rec[m]);
		if (rc != NABWA_OK) return rc;
		if (rec[m].kind == NABWA_KIND_EOF) continue;
		if (rec[m].phase == NABWA_PHASE_PRISTINE || rec[m].phase == NABWA_PHASE_ALIGNED) to_posn.push_back(m);
		else if (rec[m].phase == NABWA_PHASE_POSITIONED) { if (w->isize) to_finish.push_back(m); else { ++w->failures; ++w->counts[2]; } }
	}
	std::vector<std::vector<uint8_t>> reply((size_t)n_msg);          /* empty: the message goes back as it came */
	/* one group at a time: its reads as a BAM stream -> a batch of the front-end; a group's logical records are its messages, in order */
	auto run_group = [&](const std::vector<int> &grp, bool finish) -> int {
		if (grp.empty()) return NABWA_OK;
		Stream S; S.off.push_back(0);
		std::vector<nabwa_wire_read_t> state;
		for (int m : grp) for (int e = 0; e < rec[m].kind; ++e) { append_read(S, rec[m].read[e]); state.push_back(rec[m].read[e]); }
		const int n_reads = (int)state.size();
		BatchGuard G = { 0 };
		int rc = nabwa_bam_batch_create(w->ix, &w->opt, &w->popt, n_reads, S.buf.data(), S.off.data(), &G.b);
		if (rc != NABWA_OK) return rc;
		int nr = 0, nl = 0;
		nabwa_bam_batch_counts(G.b, &nr, &nl);
		if (nr != n_reads || nl != (int)grp.size()) return nabwa_fail(NABWA_EINVAL, "a message whose reads do not form the logical record it says it is");
		if (!finish) {
			/* a record that arrives `aligned` is searched again: same index, same options, same rows (the search is deterministic) */
			rc = nabwa_bam_batch_pass1(G.b, &w->rng48, w->scratch);
			if (rc != NABWA_OK) return rc;
			rc = nabwa_bam_batch_positioned(G.b, state.data());
		} else {
			rc = nabwa_bam_batch_restore(G.b, state.data());
			if (rc != NABWA_OK) return rc;
			uint64_t tot[2] = { 0, 0 }, mp[2] = { 0, 0 };
			rc = nabwa_bam_batch_pass2(G.b, w->isize, tot, mp);
		}
		if (rc != NABWA_OK) return rc;
		/* the records as they now stand (tags erased after create, rewritten after pass 2) */
		int64_t nb = 0;
		std::vector<int64_t> oo((size_t)n_reads + 1, 0);
		nabwa_bam_batch_output(G.b, 0, 0, oo.data(), &nb);
		std::vector<uint8_t> ob((size_t)(nb ? nb : 1));
		rc = nabwa_bam_batch_output(G.b, ob.data(), nb, oo.data(), &nb);
		if (rc != NABWA_OK) return rc;
		int at = 0;
		for (int m : grp) {
			nabwa_wire_rec_t o = rec[m];
			o.phase = finish ? NABWA_PHASE_FINISHED : NABWA_PHASE_POSITIONED;
			for (int e = 0; e < o.kind; ++e, ++at) {
				nabwa_wire_read_t &x = o.read[e];
				if (!finish) x = state[(size_t)at];                        /* the positioned and aligned parts */
				const uint8_t *r = ob.data() + oo[at];
				nabwa_wire_core_from_bam(r + 4, x.core);
				x.data = r + 36; x.data_len = (int32_t)(oo[at + 1] - oo[at] - 36);
			}
			const int64_t need = nabwa_wire_size(&o);
			reply[(size_t)m].resize((size_t)need);
			if (nabwa_wire_encode(&o, reply[(size_t)m].data(), need) != need) return nabwa_fail(NABWA_EINVAL, "internal: a reply that cannot be encoded");
			++w->counts[finish ? 1 : 0];
		}
		return NABWA_OK;
	};
	int rc = run_group(to_posn, false);
	if (rc != NABWA_OK) return rc;
	rc = run_group(to_finish, true);
	if (rc != NABWA_OK) return rc;
	for (int m = 0; m < n_msg; ++m) {
		const bool same = reply[(size_t)m].empty();
		if (same && !(rec[m].phase == NABWA_PHASE_POSITIONED && rec[m].kind != NABWA_KIND_EOF)) ++w->counts[3];
		if (send(ctx, same ? msgs[m] : reply[(size_t)m].data(), same ? lens[m] : (int64_t)reply[(size_t)m].size()) != 0) return nabwa_fail(NABWA_EIO, "the transport refused a reply");
	}
	return NABWA_OK;
}

extern "C" int nabwa_worker_core(nabwa_worker_t *w, nabwa_recv_fn recv, nabwa_send_fn send, void *ctx, const nabwa_worker_opt_t *wo)
{
	if (!w || !recv || !send) return nabwa_fail(NABWA_EINVAL, "null argument");
	const int max_batch = wo && wo->max_batch > 0 ? wo->max_batch : 1 << 16;
	const int linger = wo && wo->linger_ms >= 0 ? wo->linger_ms : 5;
	const int idle = wo && wo->idle_timeout_ms > 0 ? wo->idle_timeout_ms : 90000;            /* timeout, bam2bam.c:10 */
	std::vector<std::vector<uint8_t>> held;
	for (;;) {
		held.clear();
		const uint8_t *m = 0; int64_t len = 0;
		int got = recv(ctx, &m, &len, idle);
		if (got <= 0) return NABWA_OK;                           /* nothing for `idle` ms, or the transport is gone: the worker's clean ends */
		held.emplace_back(m, m + len);
		while ((int)held.size() < max_batch && (got = recv(ctx, &m, &len, linger)) > 0) held.emplace_back(m, m + len);
		std::vector<const uint8_t*> ptr(held.size()); std::vector<int64_t> lens(held.size());
		for (size_t i = 0; i < held.size(); ++i) { ptr[i] = held[i].data(); lens[i] = (int64_t)held[i].size(); }
		const int rc = nabwa_worker_process(w, (int)held.size(), ptr.data(), lens.data(), send, ctx);
		if (rc != NABWA_OK) return rc;
		if (w->failures >= 1024) return nabwa_fail(NABWA_EIO, "1024 positioned records came without insert-size estimates: suspected communication problem");
		if (got < 0) return NABWA_OK;
	}
}
