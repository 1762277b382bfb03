// wave_spmd.hpp -- the handful of wave-level operations kernel D (fm_deep.hip) is written in.
//
// Kernel D runs ONE search per 64-lane wavefront; its code alternates between per-lane sections and wave-level
// steps (ballot, prefix sum over lanes, broadcast).  On the GPU the macros below are the gfx950 wave intrinsics.
// With NABWA_EMU defined (tests/emu/ only -- test infrastructure, never part of libnabwa.so) the same kernel body is
// compiled by g++ as a sequential emulation of one wave: per-lane variables become arrays of 64, per-lane sections
// become loops.  That build exists so that the kernel's ordering logic can be checked against the oracle, under
// AddressSanitizer, in a container without a GPU (GPU sanitizers are not available on the pool).
#pragma once
#include <stdint.h>

#ifndef NABWA_EMU
// ---------------------------------------------------------------------------------------- gfx950
#define LANE(T, name)            T name                      /* a per-lane variable */
#define L(name)                  name
#define LANES                    if (true)                   /* a per-lane section; `ln` is the lane number */
#define ONE_LANE                 if (ln == 0)                /* side effects that depend on wave-uniform values only */
#define WBALLOT(expr)            ((uint64_t)__ballot(expr))
#define WBCAST(name, lane_)      __shfl(name, (int)(lane_))  /* value of the per-lane variable `name` in one lane */
/* Lanes of one wave exchange data through LDS and global memory (one lane stores, another loads later).  A wave's memory
 * instructions reach LDS and its CU's vector cache in program order, so wavefront scope needs no wait and no cache action --
 * only the compiler must not move accesses across the point (LLVM AMDGPU memory model: a wavefront-scope fence emits nothing).
 * (A workgroup barrier here drained every outstanding store first: several store latencies per round.) */
#define WAVE_SYNC()              do { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); } while (0)
#define WUNI(x)                  __builtin_amdgcn_readfirstlane(x)   /* a value all lanes hold: keep it in a scalar register */
#define ATOMIC_ADD_U32(p, v)     atomicAdd((p), (v))
__device__ __forceinline__ uint32_t wave_incl_scan_u32(uint32_t v)
{
	const int ln_ = (int)(threadIdx.x & 63u);
#pragma unroll
	for (int o = 1; o < 64; o <<= 1) { const uint32_t u = (uint32_t)__shfl_up((int)v, o); if (ln_ >= o) v += u; }
	return v;
}
__device__ __forceinline__ int64_t wave_max_i64(int64_t v)
{
#pragma unroll
	for (int o = 32; o > 0; o >>= 1) { const int64_t u = __shfl_xor(v, o); v = u > v ? u : v; }
	return v;
}
/* dst (per lane) = sum of src over the lower lanes, total (uniform) = sum over all lanes */
#define WEXSCAN_U32(dst, src, total) do { const uint32_t in_ = (src); const uint32_t inc_ = wave_incl_scan_u32(in_); \
	dst = inc_ - in_; total = (uint32_t)__builtin_amdgcn_readlane((int)inc_, 63); } while (0)
#define WMAX_I64(total, expr)    do { total = wave_max_i64((int64_t)(expr)); } while (0)
#else
// ---------------------------------------------------------------------------------------- CPU emulation of one wave
#define LANE(T, name)            T name[64]
#define L(name)                  name[ln]
#define LANES                    for (int ln = 0; ln < 64; ++ln)
#define ONE_LANE                 if (true)
#define WBALLOT(expr)            ([&]() -> uint64_t { uint64_t b_ = 0; for (int ln = 0; ln < 64; ++ln) if (expr) b_ |= 1ull << ln; return b_; }())
#define WBCAST(name, lane_)      name[(lane_)]
#define WAVE_SYNC()              do { } while (0)
#define WUNI(x)                  (x)
#define ATOMIC_ADD_U32(p, v)     emu_atomic_add((p), (v))
static inline unsigned int emu_atomic_add(unsigned int *p, unsigned int v) { const unsigned int o = *p; *p = o + v; return o; }
#define WEXSCAN_U32(dst, src, total) do { uint32_t a_ = 0; for (int ln = 0; ln < 64; ++ln) { const uint32_t in_ = (src); dst = a_; a_ += in_; } total = a_; } while (0)
#define WMAX_I64(total, expr)    do { int64_t a_ = INT64_MIN; for (int ln = 0; ln < 64; ++ln) { const int64_t v_ = (int64_t)(expr); if (v_ > a_) a_ = v_; } total = a_; } while (0)
#endif
