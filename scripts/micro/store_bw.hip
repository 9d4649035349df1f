// Microbenchmark: streaming-store bandwidth of a 128 MiB table by store width
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
template <typename T> __global__ __launch_bounds__(256) void k_fill(T *dst, size_t count, uint32_t v) {
	size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
	T x; __builtin_memset(&x, 0, sizeof x); *(uint32_t *)&x = v;
	for (; i < count; i += stride) dst[i] = x;
}
int main() {
	const size_t bytes = (size_t)128 << 20;
	void *buf; CK(hipMalloc(&buf, bytes));
	hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
	for (int blocks : {2048, 8192, 65536}) for (int w : {4, 8, 16}) {
		auto run = [&]() {
			if (w == 4) k_fill<uint32_t><<<blocks, 256>>>((uint32_t *)buf, bytes / 4, 7);
			else if (w == 8) k_fill<uint2><<<blocks, 256>>>((uint2 *)buf, bytes / 8, 7);
			else k_fill<uint4><<<blocks, 256>>>((uint4 *)buf, bytes / 16, 7);
		};
		run(); CK(hipEventRecord(a)); for (int i = 0; i < 5; ++i) run(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
		float ms; CK(hipEventElapsedTime(&ms, a, b));
		printf("blocks %6d width %2d: %.1f us per 128 MiB, %.2f TB/s\n", blocks, w, ms * 200, bytes * 5 / ms / 1e9);
	}
	CK(hipEventRecord(a)); for (int i = 0; i < 5; ++i) CK(hipMemsetAsync(buf, 0, bytes)); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
	float ms; CK(hipEventElapsedTime(&ms, a, b));
	printf("hipMemset: %.1f us per 128 MiB\n", ms * 200);
	return 0;
}
