// bam_header.hpp -- the text header `bwa bam2bam` writes (bam2bam.c:164-301): @HD, a new @PG chained to the old ones, @SQ from the
// index, then every old line except @HD / @SQ.  Host code, used by bam2bam_main.cpp and (alone) by a CPU test.
//
// The one subtle part is WHICH old @PG the new one names as its predecessor: the reference collects the IDs and the PP values in two
// string sets and takes "the first ID that nobody links to" in the ITERATION ORDER OF ITS HASH SET (bam2bam.c:246-255) -- with one
// unlinked @PG, as there should be, any order gives the same answer; with several the answer follows the slots of klib's khash
// (khash.h: X31 string hash, table sizes from a prime list, double hashing, rehash in place).  WordSlots below lays the same keys
// out in the same slots; tests/test_bam_header.py compares its order with the reference's own khash.h on random ID sets.
#pragma once
#include <stdint.h>
#include <string>
#include <vector>

struct WordSlots {
	std::vector<std::string> key; std::vector<uint8_t> full;          /* one entry per slot */
	uint32_t n = 0, limit = 0;                                           /* keys held; grow when n reaches limit */
	static uint32_t hash(const std::string &s)                         /* h = 31 h + c over the bytes as (signed) chars, the first byte as it is */
	{ uint32_t h = 0; bool first = true; for (char c : s) { const uint32_t v = (uint32_t)(int)(signed char)c; h = first ? v : h * 31u + v; first = false; } return h; }
	static uint32_t next_size(uint32_t at_least)                         /* the smallest table size of the list that is > at_least - 1 ... */
	{
		static const uint32_t P[] = { 3u, 11u, 23u, 53u, 97u, 193u, 389u, 769u, 1543u, 3079u, 6151u, 12289u, 24593u, 49157u, 98317u, 196613u, 393241u, 786433u,
			1572869u, 3145739u, 6291469u, 12582917u, 25165843u, 50331653u, 100663319u, 201326611u, 402653189u, 805306457u, 1610612741u, 3221225473u, 4294967291u };
		for (uint32_t p : P) if (p > at_least) return p;
		return 4294967291u;
	}
	/* first free slot on the key's probe path through a table of m slots */
	static uint32_t probe(const std::string &s, uint32_t m, const std::vector<uint8_t> &used)
	{
		const uint32_t h = hash(s); uint32_t i = h % m; const uint32_t step = 1 + h % (m - 1);
		while (used[i]) i = i + step >= m ? i + step - m : i + step;
		return i;
	}
	void grow()
	{
		const uint32_t old_m = (uint32_t)key.size(), m = next_size(old_m);
		std::vector<std::string> nk(m); std::vector<uint8_t> used(m, 0), waiting(full);      /* waiting: old slots whose key has not moved yet */
		/* the reference rehashes inside one array: a key that lands on a slot whose old key has not moved yet takes the slot and sends
		 * that key on its way next -- the order of arrival decides who gets a contested slot */
		for (uint32_t j = 0; j < old_m; ++j) {
			if (!waiting[j]) continue;
			std::string cur = key[j]; waiting[j] = 0;
			for (;;) {
				const uint32_t i = probe(cur, m, used);
				used[i] = 1;
				if (i < old_m && waiting[i]) { nk[i] = cur; cur = key[i]; waiting[i] = 0; }
				else { nk[i] = cur; break; }
			}
		}
		key.swap(nk); full.swap(used);
		limit = (uint32_t)(m * 0.77 + 0.5);
	}
	bool has(const std::string &s) const
	{
		const uint32_t m = (uint32_t)key.size();
		if (!m) return false;
		const uint32_t h = hash(s); uint32_t i = h % m; const uint32_t step = 1 + h % (m - 1), first = i;
		while (full[i] && key[i] != s) { i = i + step >= m ? i + step - m : i + step; if (i == first) return false; }
		return full[i] != 0;
	}
	void add(const std::string &s)
	{
		if (n >= limit) grow();
		const uint32_t m = (uint32_t)key.size(), h = hash(s); uint32_t i = h % m; const uint32_t step = 1 + h % (m - 1);
		while (full[i] && key[i] != s) i = i + step >= m ? i + step - m : i + step;
		if (!full[i]) { key[i] = s; full[i] = 1; ++n; }
	}
};

/* find_pp_tag (bam2bam.c:212-271): pp = the first @PG ID, in the reference's set order, that no PP names; id = "bwa", "bwa-1", ... */
static inline void find_pp_tag(const std::string &h, std::string &pp, std::string &id, bool &has_pp)
{
	WordSlots present, linked;
	size_t p = 0;
	while (p < h.size() && h[p]) {
		size_t e = h.find('\n', p); if (e == std::string::npos) e = h.size();
		if (h.compare(p, 3, "@PG") == 0) {
			size_t q = p;
			while (q < e) {
				size_t fe = h.find('\t', q); if (fe == std::string::npos || fe > e) fe = e;
				/* the reference looks for "ID:" / "PP:" at the start of the line and of every tab-separated field (the line starts with "@PG") */
				if (fe - q >= 3 && h[q + 2] == ':' && ((h[q] == 'I' && h[q + 1] == 'D') || (h[q] == 'P' && h[q + 1] == 'P')))
					(h[q] == 'I' ? present : linked).add(h.substr(q + 3, fe - q - 3));
				q = fe + 1;
			}
		}
		p = e + 1;
	}
	has_pp = false;
	for (size_t i = 0; i < present.key.size(); ++i) if (present.full[i] && !linked.has(present.key[i])) { pp = present.key[i]; has_pp = true; break; }
	id = "bwa";
	for (int n = 1; present.has(id); ++n) id = "bwa-" + std::to_string(n);
}
