"""The structs of include/nabwa.h against the reference's own headers, by the C compiler: sizes and member offsets of
gap_opt_t / bwt_aln1_t / bwa_seq_t / pe_opt_t / isize_info_t / bwt_multi1_t (bwtaln.h, bwape.h), and INTEGRATION.md's C
snippets compiled verbatim against both sets of headers.  Needs /root/reference (the build container); skipped elsewhere."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="the reference's headers are not on this machine")

PRELUDE = r'''
#include <stdint.h>
#include <stddef.h>
#include <stdlib.h>
#include <string.h>
#include "bwtaln.h"
#include "bwase.h"
#include "bwape.h"
#include "bntseq.h"
#include "utils.h"
#include "nabwa.h"
'''

ASSERTS = PRELUDE + r'''
#define SAME_SIZE(a, b) _Static_assert(sizeof(a) == sizeof(b), "sizeof " #a " vs " #b)
#define SAME_OFF(a, fa, b, fb) _Static_assert(offsetof(a, fa) == offsetof(b, fb), "offsetof " #a "." #fa)
SAME_SIZE(nabwa_gap_opt_t, gap_opt_t);
SAME_OFF(nabwa_gap_opt_t, s_mm, gap_opt_t, s_mm); SAME_OFF(nabwa_gap_opt_t, s_gapo, gap_opt_t, s_gapo); SAME_OFF(nabwa_gap_opt_t, s_gape, gap_opt_t, s_gape);
SAME_OFF(nabwa_gap_opt_t, mode, gap_opt_t, mode); SAME_OFF(nabwa_gap_opt_t, indel_end_skip, gap_opt_t, indel_end_skip);
SAME_OFF(nabwa_gap_opt_t, max_del_occ, gap_opt_t, max_del_occ); SAME_OFF(nabwa_gap_opt_t, max_entries, gap_opt_t, max_entries);
SAME_OFF(nabwa_gap_opt_t, fnr, gap_opt_t, fnr); SAME_OFF(nabwa_gap_opt_t, max_diff, gap_opt_t, max_diff); SAME_OFF(nabwa_gap_opt_t, max_gapo, gap_opt_t, max_gapo);
SAME_OFF(nabwa_gap_opt_t, max_gape, gap_opt_t, max_gape); SAME_OFF(nabwa_gap_opt_t, max_seed_diff, gap_opt_t, max_seed_diff);
SAME_OFF(nabwa_gap_opt_t, seed_len, gap_opt_t, seed_len); SAME_OFF(nabwa_gap_opt_t, n_threads, gap_opt_t, n_threads);
SAME_OFF(nabwa_gap_opt_t, max_top2, gap_opt_t, max_top2); SAME_OFF(nabwa_gap_opt_t, trim_qual, gap_opt_t, trim_qual);
SAME_SIZE(nabwa_aln1_t, bwt_aln1_t);
SAME_OFF(nabwa_aln1_t, k, bwt_aln1_t, k); SAME_OFF(nabwa_aln1_t, l, bwt_aln1_t, l); SAME_OFF(nabwa_aln1_t, score, bwt_aln1_t, score);
SAME_SIZE(nabwa_bwa_seq_t, bwa_seq_t);
SAME_OFF(nabwa_bwa_seq_t, name, bwa_seq_t, name); SAME_OFF(nabwa_bwa_seq_t, seq, bwa_seq_t, seq); SAME_OFF(nabwa_bwa_seq_t, rseq, bwa_seq_t, rseq);
SAME_OFF(nabwa_bwa_seq_t, qual, bwa_seq_t, qual); SAME_OFF(nabwa_bwa_seq_t, score, bwa_seq_t, score); SAME_OFF(nabwa_bwa_seq_t, clip_len, bwa_seq_t, clip_len);
SAME_OFF(nabwa_bwa_seq_t, n_aln, bwa_seq_t, n_aln); SAME_OFF(nabwa_bwa_seq_t, aln, bwa_seq_t, aln); SAME_OFF(nabwa_bwa_seq_t, n_multi, bwa_seq_t, n_multi);
SAME_OFF(nabwa_bwa_seq_t, multi, bwa_seq_t, multi); SAME_OFF(nabwa_bwa_seq_t, sa, bwa_seq_t, sa); SAME_OFF(nabwa_bwa_seq_t, pos, bwa_seq_t, pos);
SAME_OFF(nabwa_bwa_seq_t, n_cigar, bwa_seq_t, n_cigar); SAME_OFF(nabwa_bwa_seq_t, cigar, bwa_seq_t, cigar); SAME_OFF(nabwa_bwa_seq_t, tid, bwa_seq_t, tid);
SAME_OFF(nabwa_bwa_seq_t, bc, bwa_seq_t, bc); SAME_OFF(nabwa_bwa_seq_t, md, bwa_seq_t, md); SAME_OFF(nabwa_bwa_seq_t, max_entries, bwa_seq_t, max_entries);
/* the bit-field words: the member before and after pin where they sit */
_Static_assert(offsetof(nabwa_bwa_seq_t, bits0) == offsetof(bwa_seq_t, qual) + 8 && offsetof(nabwa_bwa_seq_t, bits1) + 4 == offsetof(bwa_seq_t, score), "len/strand/type/extra_flag and n_mm/../mapQ words");
_Static_assert(offsetof(nabwa_bwa_seq_t, c1c2seq) == offsetof(bwa_seq_t, pos) + 4 && offsetof(nabwa_bwa_seq_t, lenbits) + 4 == offsetof(bwa_seq_t, md) - 0 - 0 || offsetof(nabwa_bwa_seq_t, lenbits) + 8 == offsetof(bwa_seq_t, md), "c1/c2/seQ and full_len/nm words");
SAME_SIZE(nabwa_pe_opt_t, pe_opt_t);
SAME_OFF(nabwa_pe_opt_t, max_isize, pe_opt_t, max_isize); SAME_OFF(nabwa_pe_opt_t, force_isize, pe_opt_t, force_isize); SAME_OFF(nabwa_pe_opt_t, max_occ, pe_opt_t, max_occ);
SAME_OFF(nabwa_pe_opt_t, max_occ_se, pe_opt_t, max_occ_se); SAME_OFF(nabwa_pe_opt_t, n_multi, pe_opt_t, n_multi); SAME_OFF(nabwa_pe_opt_t, N_multi, pe_opt_t, N_multi);
SAME_OFF(nabwa_pe_opt_t, type, pe_opt_t, type); SAME_OFF(nabwa_pe_opt_t, is_sw, pe_opt_t, is_sw); SAME_OFF(nabwa_pe_opt_t, is_preload, pe_opt_t, is_preload);
SAME_OFF(nabwa_pe_opt_t, ap_prior, pe_opt_t, ap_prior);
/* isize_info_t = the histogram pointer, then nabwa_isize_t (bwape.h:16-20): &ii->avg is a nabwa_isize_t */
#define ISZ_OFF(f) _Static_assert(offsetof(nabwa_isize_t, f) + offsetof(isize_info_t, avg) == offsetof(isize_info_t, f), "isize_info_t." #f)
ISZ_OFF(avg); ISZ_OFF(std); ISZ_OFF(ap_prior); ISZ_OFF(low); ISZ_OFF(high); ISZ_OFF(high_bayesian);
_Static_assert(sizeof(nabwa_isize_t) + offsetof(isize_info_t, avg) == sizeof(isize_info_t), "isize_info_t size");
_Static_assert(sizeof(bwt_multi1_t) == 16 && offsetof(bwt_multi1_t, cigar) == 8, "bwt_multi1_t as bwa_structs.hip reads it");
int main(void) { return 0; }
'''


def cc(src, tmp_path, name, extra=()):
    f = tmp_path / name
    f.write_text(src)
    r = subprocess.run(["gcc", "-std=gnu11", "-fgnu89-inline", "-w", "-fsyntax-only", "-I" + REF, "-I" + os.path.join(ROOT, "include"), str(f)] + list(extra),
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]


def test_struct_layouts_match_the_reference_headers(tmp_path):
    cc(ASSERTS, tmp_path, "layout.c")


def test_integration_md_snippets_compile_against_the_reference_headers(tmp_path):
    """every ```c block of INTEGRATION.md, in order, as one translation unit after the reference's headers and nabwa.h; the
    globals bam2bam.c keeps (bam2bam.c:88-101) are declared extern, as a maintainer's file would see them"""
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    blocks = re.findall(r"```c\n(.*?)```", text, re.S)
    assert len(blocks) >= 4
    glue = (PRELUDE + "extern gap_opt_t *gap_opt; extern pe_opt_t *pe_opt; extern bntseq_t *bns; extern bwt_t *bwt[2];\n"
            "extern int only_aligned, debug_bam, broken_input, drop_aligned, skip_duplicates;      /* bam2bam.c:96-101 */\n")
    cc(glue + "\n".join(blocks) + "\nint main(void) { return 0; }\n", tmp_path, "integration.c")
