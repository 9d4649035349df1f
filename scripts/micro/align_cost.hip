// Microbenchmark: cost of scattered 16-byte loads by alignment (all 64 lanes active, L2-resident buffer)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int BYTES>
__global__ __launch_bounds__(256) void k_loads(const uint8_t *buf, uint32_t mask_lines, int iters, int offset, uint32_t *out) {
	uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
	uint32_t x = tid * 2654435761u + 12345u;
	uint32_t acc = 0;
	for (int i = 0; i < iters; ++i) {
		x = x * 1664525u + 1013904223u;
		uint32_t line = (x >> 8) & mask_lines;
		const uint8_t *p = buf + (size_t)line * 64 + offset;
		if (BYTES == 16) { uint4 v; __builtin_memcpy(&v, p, 16); acc += v.x ^ v.y ^ v.z ^ v.w; }
		else if (BYTES == 8) { uint2 v; __builtin_memcpy(&v, p, 8); acc += v.x ^ v.y; }
		else { uint32_t v; __builtin_memcpy(&v, p, 4); acc += v; }
		x += acc & 1u;
	}
	if (acc == 0x12345678u) out[tid] = acc;
}

int main() {
	const size_t bytes = (size_t)2 << 20;
	uint8_t *buf; uint32_t *out;
	CK(hipMalloc(&buf, bytes + 256)); CK(hipMemset(buf, 1, bytes + 256)); CK(hipMalloc(&out, 1 << 24));
	hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
	const int blocks = 256 * 8, iters = 1000;
	uint32_t mask = (uint32_t)(bytes / 64 - 1);
	for (int sz : {16, 8, 4}) for (int off : {0, 1, 2, 4, 8, 16, 20, 33, 48, 52, 56, 60}) {
		auto run = [&](int it) {
			if (sz == 16) k_loads<16><<<blocks, 256>>>(buf, mask, it, off, out);
			else if (sz == 8) k_loads<8><<<blocks, 256>>>(buf, mask, it, off, out);
			else k_loads<4><<<blocks, 256>>>(buf, mask, it, off, out);
		};
		run(10);
		CK(hipEventRecord(a)); run(iters); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
		float ms; CK(hipEventElapsedTime(&ms, a, b));
		double lane_loads = (double)blocks * 256 * iters;
		printf("size %2d offset %2d: %.3f ms  %.1f G lane-loads/s\n", sz, off, ms, lane_loads / ms / 1e6);
	}
	return 0;
}
