// Microbenchmark: what does a vector memory instruction cost when only k of the 64 lanes
// are active, for scattered 16-byte loads?  (decides how much lane divergence costs the
// memory pipeline in the anchor scan)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void k_loads(const uint4 *buf, uint32_t mask_lines, int iters, int active, uint32_t *out) {
	uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
	uint32_t lane = threadIdx.x & 63;
	uint32_t x = tid * 2654435761u + 12345u;
	uint32_t acc = 0;
	if (lane < (uint32_t)active) {
		for (int i = 0; i < iters; ++i) {
			x = x * 1664525u + 1013904223u;
			uint32_t line = (x >> 8) & mask_lines;        // 64-byte line index
			uint4 v = buf[(size_t)line * 4 + (x & 3u)];   // one 16-byte piece of it
			acc += v.x ^ v.y ^ v.z ^ v.w;
			x += acc & 1u; // dependent: one load in flight per lane
		}
	}
	if (acc == 0x12345678u) out[tid] = acc;
}

int main() {
	const size_t big = (size_t)1 << 30, small = (size_t)2 << 20; // 1 GiB (HBM), 2 MiB (L2)
	uint4 *buf; uint32_t *out;
	CK(hipMalloc(&buf, big)); CK(hipMemset(buf, 1, big)); CK(hipMalloc(&out, 1 << 24));
	hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
	const int blocks = 256 * 8, iters = 2000;
	for (int pass = 0; pass < 2; ++pass) {
		size_t bytes = pass ? small : big;
		uint32_t mask = (uint32_t)(bytes / 64 - 1);
		for (int active : {1, 2, 4, 8, 16, 32, 64}) {
			k_loads<<<blocks, 256>>>(buf, mask, 10, active, out);
			CK(hipEventRecord(a));
			k_loads<<<blocks, 256>>>(buf, mask, iters, active, out);
			CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
			float ms; CK(hipEventElapsedTime(&ms, a, b));
			double winstr = (double)blocks * 4 * iters;
			printf("%s active %2d: %.3f ms, %.1f ns per wave-load per CU-slot, %.2f G lane-loads/s, %.1f cycles/wave-load/CU\n",
				   pass ? "L2 " : "HBM", active, ms, ms * 1e6 / winstr * 256, winstr * active / ms / 1e6,
				   ms * 1e-3 * 2.4e9 / (winstr / 256));
		}
	}
	return 0;
}
