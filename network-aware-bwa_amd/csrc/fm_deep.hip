// fm_deep.hip -- kernel D for gfx950: bwt_match_gap (bwtgap.c:104-266) for the deep searches, one search per wavefront.
// The kernel body lives in fm_deep_body.hpp (design notes there); this file is its __global__ entry and launcher.
// Block = ONE wave (64 threads): everything the wave shares (per-score counts and top pages) is its block's LDS, and
// __syncthreads() is the wave's own memory fence.  Persistent grid: n_waves blocks draw reads from a counter.
#include "fm_deep_body.hpp"

#ifndef NABWA_DEEP_WAVES
#define NABWA_DEEP_WAVES 4    // waves per SIMD the register budget is bounded for: 128 VGPRs (22 spilled dwords), 16 searches per CU.  The trade-off was
                              // measured twice.  With every child built and stored inside the chain step (about 180 live registers): 2 / 3 / 4 / 5 waves ->
                              // 1074 / 1488 / 2383 / 4611 wave-seconds for 1 M aDNA reads, i.e. 0.52 / 0.48 / 0.58 / 0.90 s of work per wave -- three waves
                              // won, every spilled dword being a scratch access in the same in-order queue as the gathers.  With records (165 registers
                              // unbounded): four waves take 9 % off the paired-end workload's kernel time against three.
#endif

extern __shared__ __attribute__((aligned(16))) uint32_t s_deep[];

template <bool PROF, bool LDSM, bool COOP>
// The statistics instantiation (phase clocks, touch counts: never timed) gets the whole register file: at 128 registers it spilled some 70 of
// them -- among them the registers that hold spilled SCALAR values lane by lane -- and round 3 saw that build fault on the GPU
// (HSA_STATUS_ERROR_MEMORY_APERTURE_VIOLATION, a pointer read back wrong) while the same source passes in the CPU emulation and in the build
// without statistics.  With two waves per SIMD nothing of it is spilled to memory.
__global__ __launch_bounds__(64, PROF ? 2 : NABWA_DEEP_WAVES) void fm_deep_kernel(const DeepParams P)
{
	deep_wave_body<PROF, LDSM, COOP>(P, s_deep, blockIdx.x, (int)(threadIdx.x & 63u));
}

extern "C" void nabwa_launch_fm_deep(const DeepParams *P, int n_waves, hipStream_t s)
{
	const size_t lds = (size_t)DEEP_LDS_WORDS(P->NS, P->lds_rd) * 4u;
	const bool prof = P->stats || P->S.touch_counter, ldsm = P->lds_rd != 0u, coop = P->coop_lanes != 0u;
#define DEEP_GO(a_, b_, c_) hipLaunchKernelGGL((fm_deep_kernel<a_, b_, c_>), dim3(n_waves), dim3(64), lds, s, *P)
	if (prof) { if (ldsm) { if (coop) DEEP_GO(true, true, true); else DEEP_GO(true, true, false); } else { if (coop) DEEP_GO(true, false, true); else DEEP_GO(true, false, false); } }
	else { if (ldsm) { if (coop) DEEP_GO(false, true, true); else DEEP_GO(false, true, false); } else { if (coop) DEEP_GO(false, false, true); else DEEP_GO(false, false, false); } }
#undef DEEP_GO
}

extern "C" int nabwa_deep_occupancy(int ns, int lds_rd)
{
	int nb = 0;
	return hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, lds_rd ? fm_deep_kernel<false, true, true> : fm_deep_kernel<false, false, true>, 64, (size_t)DEEP_LDS_WORDS((unsigned)ns, (unsigned)lds_rd) * 4u) == hipSuccess ? nb : 0;
}
