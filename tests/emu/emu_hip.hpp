// tests/emu/emu_hip.hpp -- TEST INFRASTRUCTURE ONLY.  What nabwa_dev.hpp / fm_deep_body.hpp need from the HIP headers,
// for the g++ build that emulates one wavefront of kernel D on the CPU (see network-aware-bwa_amd/csrc/wave_spmd.hpp).
#pragma once
#include <stdint.h>
#include <stddef.h>
struct uint2 { uint32_t x, y; };
struct uint4 { uint32_t x, y, z, w; };
static inline uint2 make_uint2(uint32_t x, uint32_t y) { uint2 r = { x, y }; return r; }
static inline uint4 make_uint4(uint32_t x, uint32_t y, uint32_t z, uint32_t w) { uint4 r = { x, y, z, w }; return r; }
#define __device__
#define __forceinline__ inline
static inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
static inline int __ffsll(unsigned long long v) { return __builtin_ffsll((long long)v); }
static inline int __popc(unsigned int v) { return __builtin_popcount(v); }
static inline int __clz(int v) { return v ? __builtin_clz((unsigned)v) : 32; }
