// fm_deep.hpp -- parameter block of kernel D (fm_deep_body.hpp / fm_deep.hip), shared with the host API.
#pragma once
#include "fm_search.hpp"

#define DEEP_PAGE_SH 8u
#define DEEP_PAGE    (1u << DEEP_PAGE_SH)    /* entries per page */
#define DEEP_NIL     0xffffffffu
#define DEEP_NEWP    192u                      /* pages one commit can need at most: 64 lanes x stage_k <= 48 records x <= 9 children / 256 + one partly filled page per class <= 111 */
#define DEEP_STAGE_MAX 48u                      /* records a chain may file per round (64 B each): 64 lanes x 48 x 9 children stay within DEEP_NEWP pages */
#define DRG_OPEN  1u                          /* a record's gap group: gap open = the insertion + the deletions; extension of an insertion; of a deletion */
#define DRG_EXT_I 2u
#define DRG_EXT_D 3u
#define DST_M 0
#define DST_I 1
#define DST_D 2
#define DF_NONE 0
#define DF_HIT  1
#define DF_CONT 2
#define DF_TAIL 3      /* the chain ends in an exact tail that is still to be walked */
#define DCL_MM 0u      /* child classes = the three scores a chain at score s pushes to: s + s_mm, s + s_gapo, s + s_gape */
#define DCL_GO 1u
#define DCL_GE 2u

#define DEEP_BC_WORDS 16u                       /* per index 8 words: bucket array (2), primary, seq_len, L2[1..3], one spare -- picked from SearchParams.ixtab */
#define DEEP_LDS_WORDS(ns_, rd_) (2u * (((ns_) + 1u) & ~1u) + DEEP_NEWP + 256u + DEEP_BC_WORDS + ((rd_) + 3u) / 4u)

struct DeepParams {
	SearchParams S;                  // index, reads, width records, options, outputs (n_aln / max_ent / status / aln by work item or res_slot)
	uint4 *pages;                    // the pool: n_pages x 256 entries x 16 B
	uint32_t *page_prev;             // per page: the page below it in its level's stack
	uint32_t n_pages;
	unsigned int *page_bump;         // pages handed out so far (a wave keeps what it took and re-uses it for its next reads)
	uint32_t *own;                   // per wave 2 x own_cap ids: the pages it holds, and those of them that are free
	uint32_t own_cap;
	uint4 *stage;                    // per wave [64][stage_k] records of 64 bytes: what the chains of the running round push, lane by lane
	uint32_t stage_k;
	uint32_t NS;                     // score levels
	uint32_t lds_rd, rd_pl;          // bytes of LDS for the read's own data (2 WLB + 2 SLB + 2 rd_pl; 0: it stays in global memory), stride of a strand's bases there
	                                 // LDS per wave: DEEP_LDS_WORDS(NS, lds_rd) words
	int careful_all, max_lanes;      // test knobs: every round one pop; lanes a round may use (production: 0, 64)
	uint32_t *rounds_out;            // or null (statistics build): per work item the rounds its search took
	unsigned long long *stats;       // or null: [0] rounds, [1] lane-chains run, [2] chains committed, [3] chain steps, [4] careful rounds, [5] pool failures, [6] rank steps and [7] text finishes of exact tails (lane counts)
};
