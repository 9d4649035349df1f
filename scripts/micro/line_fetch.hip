// Microbenchmark: how fast does the chip deliver WHOLE cache lines at random addresses to single
// lanes, by the way they are asked for?  (decides how pass A of the anchor scan should fetch the
// lines of its per-lane sequence streams)
//
// Every lane follows a dependent chain of random line addresses (the next address depends on the
// loaded data), like a chain of the scan.  Modes:
//   0  one 16-byte piece of a random 64-B line per step            (what pass A did in round 1)
//   1  all four pieces of a random 64-B line, four loads by the same lane, back to back
//   2  all eight pieces of a random 128-B line, eight loads by the same lane
//   3  as 1, through LDS-DMA (global_load_lds_dwordx4, no data VGPRs) and one ds_read_b128
//   4  as 2, through LDS-DMA
//   5  four neighbouring lanes share a random 64-B line (one piece each: a fully coalesced request)
//   6  eight neighbouring lanes share a random 128-B line
//   7, 8, 9  as 0, 2, 4 with every address 3 bytes off a 16-byte boundary (the subject side of the scan reads
//      the packed text at byte granularity)
// Reported: lines/s, bytes/s of lines, for footprints 1 GiB (HBM), 128 MiB (Infinity Cache) and
// 16 MiB (half of the L2s), at 8 and at `occ` resident wavefronts per SIMD (LDS padding).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

typedef __attribute__((address_space(1))) const uint4 *g_u4p;
typedef __attribute__((address_space(1))) const uint8_t *g_u8;
__device__ __forceinline__ uint4 ld16u(g_u8 p) { // any byte address
	uint4 v;
	__builtin_memcpy(&v, p, 16);
	return v;
}
__device__ __forceinline__ uint4 ld16(g_u4p p) {
	uint4 v;
	__builtin_memcpy(&v, p, 16);
	return v;
}
typedef __attribute__((address_space(3))) void *lds_p;

template <int MODE>
__global__ __launch_bounds__(256) void k_lines(const uint4 *buf_, uint32_t mask_bytes, int iters, uint32_t *out) {
	extern __shared__ uint4 s_dyn[]; // [piece][thread] for the LDS-DMA modes; otherwise only occupancy padding
	g_u4p buf = (g_u4p)buf_;
	const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
	const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
	uint32_t x = tid * 2654435761u + 12345u;
	if (MODE == 5) x = (tid >> 2) * 2654435761u + 12345u;
	if (MODE == 6) x = (tid >> 3) * 2654435761u + 12345u;
	uint32_t acc = 0;
	constexpr int PIECES = (MODE == 1 || MODE == 3) ? 4 : (MODE == 2 || MODE == 4 || MODE == 8 || MODE == 9) ? 8 : 1;
	constexpr uint32_t LINE = (MODE == 2 || MODE == 4 || MODE == 6 || MODE == 8 || MODE == 9) ? 128u : 64u;
	for (int i = 0; i < iters; ++i) {
		x = x * 1664525u + 1013904223u;
		const uint32_t line = ((x >> 4) & mask_bytes) & ~(LINE - 1u); // byte offset of the line
		if constexpr (MODE == 0) {
			const uint4 v = ld16(buf + (line >> 4) + (x & 3u));
			acc += v.x ^ v.y ^ v.z ^ v.w;
		} else if constexpr (MODE == 1 || MODE == 2) {
			uint4 v[PIECES];
#pragma unroll
			for (int k = 0; k < PIECES; ++k) v[k] = ld16(buf + (line >> 4) + k);
#pragma unroll
			for (int k = 0; k < PIECES; ++k) acc += v[k].x ^ v[k].w;
		} else if constexpr (MODE == 3 || MODE == 4) {
			// piece k of this wave's lanes lands at s_dyn[(wave * PIECES + k) * 64 + lane]
#pragma unroll
			for (int k = 0; k < PIECES; ++k)
				__builtin_amdgcn_global_load_lds((g_u4p)(buf + (line >> 4) + k), (lds_p)(s_dyn + (wave * PIECES + k) * 64), 16, 0, 0);
			asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
			const uint4 v = s_dyn[(wave * PIECES + (x & (PIECES - 1))) * 64 + lane];
			acc += v.x ^ v.w;
		} else if constexpr (MODE == 7) {
			const uint4 v = ld16u((g_u8)buf + line + 16 * (x & 3u) + 3);
			acc += v.x ^ v.y ^ v.z ^ v.w;
		} else if constexpr (MODE == 8) {
			uint4 v[PIECES];
#pragma unroll
			for (int k = 0; k < PIECES; ++k) v[k] = ld16u((g_u8)buf + line + 16 * k + 3);
#pragma unroll
			for (int k = 0; k < PIECES; ++k) acc += v[k].x ^ v[k].w;
		} else if constexpr (MODE == 9) {
#pragma unroll
			for (int k = 0; k < PIECES; ++k)
				__builtin_amdgcn_global_load_lds((g_u8)buf + line + 16 * k + 3, (lds_p)(s_dyn + (wave * PIECES + k) * 64), 16, 0, 0);
			asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
			const uint4 v = s_dyn[(wave * PIECES + (x & (PIECES - 1))) * 64 + lane];
			acc += v.x ^ v.w;
		} else if constexpr (MODE == 5) {
			const uint4 v = ld16(buf + (line >> 4) + (lane & 3u));
			acc += v.x ^ v.w;
			acc = (uint32_t)__shfl((int)acc, (int)(lane & ~3u)); // the group stays on one chain
		} else {
			const uint4 v = ld16(buf + (line >> 4) + (lane & 7u));
			acc += v.x ^ v.w;
			acc = (uint32_t)__shfl((int)acc, (int)(lane & ~7u));
		}
		x += acc & 1u; // dependent: the next address needs this step's data
	}
	if (acc == 0x12345678u) out[tid] = acc;
}

template <int MODE>
static void run(const char *what, const uint4 *buf, size_t bytes, int blocks_per_cu, uint32_t *out, hipEvent_t a, hipEvent_t b) {
	const int blocks = 256 * 8 * 4, iters = 400;
	constexpr int PIECES = (MODE == 1 || MODE == 3) ? 4 : (MODE == 2 || MODE == 4 || MODE == 8 || MODE == 9) ? 8 : 1;
	constexpr uint32_t LINE = (MODE == 2 || MODE == 4 || MODE == 6 || MODE == 8 || MODE == 9) ? 128u : 64u;
	size_t lds = (MODE == 3 || MODE == 4 || MODE == 9) ? (size_t)256 * PIECES * 16 : 0;
	const size_t want = (size_t)160 * 1024 / blocks_per_cu; // LDS per block that admits exactly blocks_per_cu
	if (blocks_per_cu < 8 && lds < want - 1024) lds = want - 1024;
	CK(hipFuncSetAttribute((const void *)k_lines<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
	k_lines<MODE><<<blocks, 256, lds>>>(buf, (uint32_t)(bytes - 1), 10, out);
	CK(hipEventRecord(a));
	k_lines<MODE><<<blocks, 256, lds>>>(buf, (uint32_t)(bytes - 1), iters, out);
	CK(hipEventRecord(b));
	CK(hipEventSynchronize(b));
	float ms;
	CK(hipEventElapsedTime(&ms, a, b));
	const double chains = (double)blocks * 256 / (MODE == 5 ? 4 : MODE == 6 ? 8 : 1);
	const double lines = chains * iters;
	printf("%-34s occ %d lds %6zu: %8.3f ms  %7.2f G lines/s  %7.2f TB/s of %3u-B lines  (%.0f ns per dependent step)\n", what, blocks_per_cu, lds,
		   ms, lines / ms / 1e6, lines * LINE / ms / 1e9, LINE, ms * 1e6 / iters / ((double)blocks / (256.0 * blocks_per_cu)));
}

int main() {
	const size_t big = (size_t)1 << 30;
	uint4 *buf;
	uint32_t *out;
	CK(hipMalloc(&buf, big));
	CK(hipMemset(buf, 1, big));
	CK(hipMalloc(&out, 1 << 26));
	hipEvent_t a, b;
	CK(hipEventCreate(&a));
	CK(hipEventCreate(&b));
	if (getenv("LINE_FETCH_UNALIGNED")) { // the unaligned modes only
		const size_t bytes = (size_t)1 << 30;
		for (int occ : {8, 3}) {
			run<0>("16-B piece of a 64-B line", buf, bytes - 256, occ, out, a, b);
			run<7>("16 B, 3 bytes off alignment", buf, bytes - 256, occ, out, a, b);
			run<2>("128-B line, 8 loads of one lane", buf, bytes - 256, occ, out, a, b);
			run<8>("128 B from 3 bytes off, 8 loads", buf, bytes - 256, occ, out, a, b);
			run<4>("128-B line, LDS-DMA", buf, bytes - 256, occ, out, a, b);
			run<9>("128 B from 3 bytes off, LDS-DMA", buf, bytes - 256, occ, out, a, b);
		}
		return 0;
	}
	for (size_t bytes : {(size_t)1 << 30, (size_t)128 << 20, (size_t)16 << 20}) {
		printf("---- footprint %zu MiB\n", bytes >> 20);
		for (int occ : {8, 4, 3}) {
			run<0>("16-B piece of a 64-B line", buf, bytes, occ, out, a, b);
			run<1>("64-B line, 4 loads of one lane", buf, bytes, occ, out, a, b);
			run<2>("128-B line, 8 loads of one lane", buf, bytes, occ, out, a, b);
			run<3>("64-B line, LDS-DMA", buf, bytes, occ, out, a, b);
			run<4>("128-B line, LDS-DMA", buf, bytes, occ, out, a, b);
			run<5>("64-B line shared by 4 lanes", buf, bytes, occ, out, a, b);
			run<6>("128-B line shared by 8 lanes", buf, bytes, occ, out, a, b);
		}
	}
	return 0;
}
