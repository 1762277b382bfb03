// fm_search.hpp -- parameter block shared by the search kernel and the host API.
#pragma once
#include <stdint.h>
#include "nabwa_dev.hpp"

#define NABWA_SEARCH_BLOCK 256
#define NABWA_ST_OK        0
#define NABWA_ST_OVERFLOW  1   // arena or hit list outgrown in the first pass (kernel S): the read goes to kernel D
#define NABWA_ST_WIDE      2   // resolved by kernel D: the result lives in the wide result arrays at wide_idx[read]
#define NABWA_ST_POOL      3   // kernel D: the page pool ran dry under this read -- it is run again in the guaranteed pass
#define NABWA_ST_HITCAP    4   // kernel D: more hit rows than the wide result rows (NABWA_ALNCAP2): searched again with longer lists
#define NABWA_ST_GROWN     5   // resolved by such a second search: the rows are the block grown[wide_idx[read]] names

struct SearchParams {
	DevBwt bwt[2];
	// reads (device): codes 0-3, 4 = N; seq = read reversed, rseq = reverse complement
	const uint8_t *seq, *rseq;                // padded layout: read i starts at poff[i], a multiple of 16
	const int64_t *poff;
	const int32_t *rd_len;
	const uint8_t *rd_maxdiff, *rd_maxgapo;   // per read: max_diff and clamped max_gapo (host-side FP, bwtaln.c:104-105,125)
	const int32_t *ids;                       // work item -> read id (kernel S: the class-sorted work order; kernel D: its list), or null
	const int32_t *res_slot;                  // kernel D: read id -> row of its result arrays, or null = list position
	int n;
	// gap_opt_t fields that are uniform over the batch
	int s_mm, s_gapo, s_gape, mode, indel_end_skip, max_del_occ, max_entries, max_gape, max_seed_diff, seed_len, max_top2;
	// per-READ width record written by kernel W: [2][WL] u32 widths, then [2][WLB] and [2][SLB] bound bytes
	uint8_t *wdata;
	size_t wstride;
	uint32_t woff_bid, woff_sbid;
	uint32_t WL, WLB, SLB;
	int text_mode;                            // 0: never leave the FM-index (the touch-counting run); 1: text mode where the index has it
	const uint32_t *rd_key;                   // per read six interval-table keys [6*rid + ..] (see pad_reads_kernel), ~0u = none
	uint8_t *rd_nN;                           // per read: number of N in the read, saturated at 255
	// per-lane scratch of kernel S: the arena
	uint8_t *scratch;
	size_t lane_stride;
	uint32_t cap, NS;
	// outputs, indexed by work item
	int32_t *n_aln, *max_ent;
	uint8_t *status;
	uint4 *aln;
	int aln_cap;
	unsigned int *work_counter;               // [0] kernel S, [1] kernel W
	uint32_t trip_budget;                     // kernel S: trips after which a search is handed on to kernel D (0: never)
	uint32_t trip_budget_hard;                // ... in a batch most of whose reads have no exact occurrence on either strand (*n_sync > n / 2: the searches are kernel D's kind)
	int sync_refill;                          // experiment knob (NABWA_SYNC_REFILL): every wave refills only when all its lanes are idle
	const uint32_t *rd_pack; int pack_stride;  // both strands of every read 2 bits per base (pad_reads_kernel); words per read
	int w_sync;                               // kernel W: lockstep waves (all reads of the batch have one length)
	int w_skip_clean;                         // kernel W run again over the reads kernel S handed on: a search that found no hit has not edited its record (gap_shadow, bwtgap.c:81-91) -- n_aln[read] == 0: nothing to rebuild
	uint8_t *rd_cls;                          // kernel W -> partition: per strand the restarts of its width pass, clipped to 4
	const unsigned int *n_sync;               // work items from *n_sync on are class-0 reads: their waves run in lockstep (see fm_search_kernel, partition_kernel)
	unsigned long long *touch_counter;        // non-null: also count the reference algorithm's bucket touches
	const uint32_t *ixtab;                    // NABWA_IXTAB_WORDS words (device): per index what a lane picks by the strand of its entry -- see below
};

/* The per-index constants a lane of kernel D selects by its entry's strand, as a table the kernel copies into LDS (fm_deep_body.hpp): as
 * selects over the kernel's arguments they were a dozen scalar registers live across the chain loop -- spilled, and read back lane by lane
 * (v_readlane) at every use.  (The same for kernel S -- all its index pointers through such a table -- took its spilled scalars from 126 to 73
 * and made it 6 % SLOWER: an LDS round trip in front of every load.  Measured in round 3, not kept.)  Per index NABWA_IXTAB_STRIDE words: */
#define NABWA_IXTAB_STRIDE 24
#define NABWA_IXTAB_WORDS  48
#define IX_BK 0                  /* pointers: two words each, low word first */
#define IX_KMER 2
#define IX_KMER_LO 4
#define IX_TEXT 6
#define IX_ISA 8
#define IX_SA_FULL 10
#define IX_PRIMARY 12
#define IX_SEQ_LEN 13
#define IX_L2 13                 /* IX_L2 + c for c = 1..3 */
static inline void nabwa_ixtab_fill(uint32_t *tab, const DevBwt *bwt /* [2] */)
{
	for (int q = 0; q < 2; ++q) {
		uint32_t *o = tab + NABWA_IXTAB_STRIDE * q;
		const uint64_t p[6] = { (uint64_t)(uintptr_t)bwt[q].bk, (uint64_t)(uintptr_t)bwt[q].kmer, (uint64_t)(uintptr_t)bwt[q].kmer_lo,
								(uint64_t)(uintptr_t)bwt[q].text, (uint64_t)(uintptr_t)bwt[q].isa, (uint64_t)(uintptr_t)bwt[q].sa_full };
		for (int f = 0; f < 6; ++f) { o[2 * f] = (uint32_t)p[f]; o[2 * f + 1] = (uint32_t)(p[f] >> 32); }
		o[IX_PRIMARY] = bwt[q].primary; o[IX_SEQ_LEN] = bwt[q].seq_len; o[IX_L2 + 1] = bwt[q].L2[1]; o[IX_L2 + 2] = bwt[q].L2[2]; o[IX_L2 + 3] = bwt[q].L2[3];
		for (int f = 17; f < NABWA_IXTAB_STRIDE; ++f) o[f] = 0;
	}
}
