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
#define DF_RUN  4      /* the chain stands in front of forced levels (no difference may be pushed) on ONE row: walked on the text, then the chain goes on */
#define DF_COOP 5      /* ... on ONE row in front of levels that may push differences: the wave takes the chain's next levels together (one level per lane) */
#define DCL_MM 0u      /* child classes = the three scores a chain at score s pushes to: s + s_mm, s + s_gapo, s + s_gape */
#define DCL_GO 1u
#define DCL_GE 2u

#define DEEP_BC_WORDS 24u                       /* per index 12 words: bucket array (2), primary, seq_len, L2[1..3], one spare, interval table (2), two spare -- picked from SearchParams.ixtab */
/* KEY FORM of an entry (as in kernel S, fm_search.hip): while the string of reference symbols an entry stands for -- matches, mismatches and
 * deleted symbols; an insertion adds none -- is shorter than the interval table is deep, the entry carries that string (k = its symbols as
 * base-4 digits, first one most significant; l = DEEP_KEYL | its length t) instead of the string's rows.  Its four possible extensions are
 * 32 consecutive bytes of level t + 1 of the table: one load instead of the rank query's two buckets, and no counting.  Rows are taken from
 * the table where the reference's results show them: a hit, level T, an exact tail (one jump down to level T), the occurrence test of a
 * deletion's extension.  An index of 0xffffff00 rows or more is searched without it (the host sets key_T = 0), so l tells the forms apart. */
#define DEEP_KEYL 0xffffff00u
#define DEEP_LVO(t_) ((0x55555555u & ((1u << (2u * (t_) - 2u)) - 1u)) << 2)      /* entries in front of level t of the table, 1 <= t <= 16: (4^t - 4) / 3 */
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
	uint32_t key_T;                  // depth of the interval tables the searches may use for key-form entries (0: rows only -- no tables, or the touch-counting run)
	uint32_t coop_lanes;             // a chain on one row is taken over by the whole wave (64 levels at a time) when no more than this many chains of the round are still running (0: never)
	int hist;                        // statistics build: also the expansions by depth and form (stats[32 ..])
	int careful_all, max_lanes;      // test knobs: every round one pop; lanes a round may use (production: 0, 64)
	uint32_t *rounds_out;            // or null (statistics build): per work item the rounds its search took
	unsigned long long *stats;       // or null: [0] rounds, [1] lane-chains run, [2] chains committed, [3] chain steps, [4] careful rounds, [5] pool failures, [6] rank steps and [7] text finishes of exact tails (lane counts)
};
