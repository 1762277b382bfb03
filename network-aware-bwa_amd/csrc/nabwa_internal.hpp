// nabwa_internal.hpp -- host-side structures shared by the translation units of libnabwa.so
#pragma once
#include <stdint.h>
#include <string>
#include <vector>
#include "fm_search.hpp"

struct nabwa_ann { int64_t offset; int32_t len, n_ambs; std::string name; };    // bntann1_t, bntseq.h:39-45
struct nabwa_hole { int64_t offset; int32_t len; char amb; };                   // bntamb1_t, bntseq.h:47-51

struct nabwa_reference {          // bntseq_t + the packed reference (bntseq.h:53-61, bwtio.c pac)
	int64_t l_pac; uint32_t seed;
	std::vector<nabwa_ann> anns;
	std::vector<nabwa_hole> holes;
	std::vector<uint8_t> pac;     // 2 bits per base, 4 bases per byte, first base in the top bits (bwtaln.h:33)
};

struct nabwa_index {
	int device;
	DevBwt bwt[2];
	uint4 *bk[2];
	uint32_t *sa[2];
	uint32_t *sa_full[2], *isa[2], *text[2];   // text-mode companions (nabwa_dev.hpp), null when switched off
	uint2 *kmer[2], *kmer_top[2];  // interval table: levels 1..LW back to back; level T on its own when T > LW (else inside the former)
	uint64_t bytes;
	int kmer_T_pick = -1;           // depth of the interval tables, decided when the first direction is built
	nabwa_reference *ref;
	struct nabwa_dev_pool *pool;   // released working buffers of earlier batches, kept for the next one (nabwa_api.hip)
};

int nabwa_fail(int code, const char *fmt, const char *a = "");

