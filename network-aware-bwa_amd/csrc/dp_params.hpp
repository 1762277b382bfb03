// dp_params.hpp -- launch parameters of the alignment kernels (dp_global.hip, dp_wave.hip), shared with the host code that fills them
// (se_finish.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

struct DpParams {
	int n;
	const int64_t *ref_off, *qry_off;
	const uint8_t *ref, *qry;
	int gap_open, gap_ext, gap_end, band;
	int matrix[25];
	int W;                 // max_l1 + 1
	int H;                 // max_l2 + 1
	int wb;                // cells a row's band can hold at most, + 1: min(W, 2 band + max |l1 - l2| + 1) -- the width of a task's direction rows in the wave-per-task form (0: not known, that form is not used)
	int32_t *rows;         // per wave: [6][W][64]
	uint8_t *tb;           // per wave: [H][W][64], byte = Mt | It<<2 | Dt<<4
	uint8_t *path;         // per wave: [W+H][64]
	int32_t *score, *n_cigar; uint32_t *cigar; int max_cigar;
};

struct LocParams {
	int n;
	const int64_t *ref_off, *qry_off;
	const uint8_t *ref, *qry;
	int gap_open, gap_ext, thres;
	int matrix[25], max_score;
	int W;                 // max_l1 + 2
	int H;                 // max_l2 + 1
	int row_forward;       // tests: take the forward pass row by row (the form with the 16-bit drop) whatever the read length
	int32_t *eh;           // rows in HBM (windows too long for LDS): per task 2 x W words
	int32_t *suba;         // per task: [H] row maxima
	int32_t *out;          // per task: score_f, score_r, start_i, start_j, end_i, end_j
};

struct ExtParams {
	int n;
	const int64_t *ref_off, *qry_off;
	const uint8_t *ref, *qry;
	const int32_t *g0;
	int gap_open, gap_ext, band;
	int matrix[25];
	int W;                 // max_l1 + 2
	uint32_t *eh;          // rows in HBM (windows too long for LDS): per task 2 x W words
	int32_t *score, *end_i, *end_j;
};

extern "C" void nabwa_launch_dp_global(const DpParams *P, hipStream_t s);
extern "C" void nabwa_launch_dp_local(const LocParams *P, hipStream_t s);
extern "C" void nabwa_launch_dp_extend_fwd(const ExtParams *P, hipStream_t s);
extern "C" int nabwa_dp_local_fits_lds(int W);             /* do a task's two rows and its window fit LDS? else the rows live in HBM */
extern "C" size_t nabwa_dp_local_rows_bytes(int W);        /* HBM form: bytes per task */
