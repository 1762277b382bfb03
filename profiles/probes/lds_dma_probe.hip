// Probe: layout and divergence behaviour of global_load_lds_dwordx4 on gfx950 (one 16-byte load per lane, straight into LDS).
// hipcc --offload-arch=gfx950 -O3 lds_dma_probe.hip -o lds_dma_probe && ./lds_dma_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
extern __shared__ uint32_t sm[];
__global__ void k(const uint4 *g, uint32_t *out, int n, int mask_mod)
{
	const int i = blockIdx.x * 256 + threadIdx.x;
	uint32_t *base = sm + (threadIdx.x & ~63) * 4;
	for (int w = 0; w < 4; ++w) sm[threadIdx.x * 4 + w] = 0xdeadbeefu;
	__syncthreads();
	if (i < n && (threadIdx.x % mask_mod) == 0)
		__builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(g + (n - 1 - i)), (void __attribute__((address_space(3)))*)base, 16, 0, 0);
	__builtin_amdgcn_s_waitcnt(0);
	__syncthreads();
	for (int w = 0; w < 4; ++w) out[i * 4 + w] = sm[threadIdx.x * 4 + w];
}
int main()
{
	const int n = 512;
	std::vector<uint32_t> h(n * 4), o(n * 4);
	for (int i = 0; i < n * 4; ++i) h[i] = i;
	uint4 *g; uint32_t *d;
	hipMalloc(&g, n * 16); hipMalloc(&d, n * 16);
	hipMemcpy(g, h.data(), n * 16, hipMemcpyHostToDevice);
	for (int mm : { 1, 3 }) {
		hipLaunchKernelGGL(k, dim3(2), dim3(256), 4096, 0, g, d, n, mm);
		hipMemcpy(o.data(), d, n * 16, hipMemcpyDeviceToHost);
		int bad = 0;
		for (int i = 0; i < n; ++i) for (int w = 0; w < 4; ++w) {
			const uint32_t want = (i % 256 % mm) == 0 ? (uint32_t)((n - 1 - i) * 4 + w) : 0xdeadbeefu;
			if (o[i * 4 + w] != want) { if (bad < 5) printf("mask_mod %d lane %d word %d: got %u want %u\n", mm, i, w, o[i * 4 + w], want); ++bad; }
		}
		printf("mask_mod %d: %s (lane-consecutive 16-byte layout, inactive lanes untouched)\n", mm, bad ? "MISMATCH" : "ok");
	}
	return 0;
}
