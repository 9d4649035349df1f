// sa_device.hip — the suffix array of a subject text, built on the device.
//
// Replaces divsufsort() at src/esa.c:303 where it bounds the whole job: on the host one
// 9.8 M-character text costs a core 0.4 s alone and 1.3 s when 29 sort at once
// (host_sais.cpp); here 1.6 ms (4.2 M: 0.8 ms, 10^8: 13.9 ms; measured on MI355X,
// profiles/r02_sa_device.txt).  A suffix array is unique, so any correct sorter gives the
// same downstream bits; tests/test_esa_gpu.py compares this one with the host sorter entry
// for entry on every kind of text.
//
// Prefix doubling on radix sorts (Manber-Myers / Larsson-Sadakane ranks, with the suffixes
// that are already in their final place dropped from every later round):
//   round 0   sort all suffixes by their first 16 symbols as 2-bit codes -- 32-bit keys, four 8-bit passes of the
//             radix sort over 8 bytes per suffix (round 4: 3 bits per symbol, 48-bit keys, six passes over 12 bytes).
//             A suffix with a symbol that is no nucleotide among its first 16 (a separator ! # ; or the end of the
//             text: "special"; sixteen per separator) gets zeros from that symbol on: it lands at the head of its
//             prefix's bucket, which is its place unless a suffix "prefix AAAA..." or another special has the same
//             key -- a tie like any other.  The sorted indices ARE the suffix array but for the ties (a few
//             suffixes in a thousand in random-like DNA): those are picked out in one pass over the keys; no ranks,
//             no group heads, no scatter for the others;
//   round 1   the suffixes left are keyed by (their group, where their first non-nucleotide is, which it is, the
//             symbols of the TEXT behind it -- or behind the first 16 -- as many as the key has room for, 3 bits
//             each): specials before the suffixes they tied with, in their true order; ordered by 27 symbols
//             (9.8 M characters);
//   round r   a suffix i that still shares its place with others -- inside a repeat -- is keyed by
//             (group of i, group of i + h): sorting those keys orders it by 2h symbols (the ranks
//             of ALL suffixes are written once, when the first such round begins).  Groups are contiguous in
//             the array and keep their span, so the sorted suffixes go back to the slots the unsorted ones
//             came from.
// Random-like DNA is done after round 1; repeats take log2(length / 27) more rounds over the suffixes inside
// them only.  Sorting and scans are rocPRIM's (hipcub front end); the kernels here make keys, pick out the ties,
// group heads and ranks.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <algorithm>

#include "andi_dev.h"
#include "sa_device.h"

namespace {

#define SA_BLOCK 256
#define SA_SYMS 16 /* symbols of a round-0 key: the K <= 13 symbols the scan index's records want and up to four behind them */
#define SA_TILE (4 * SA_BLOCK) /* slots of a block of the selection */
#define SA_TRY(call)                      \
	do {                                  \
		hipError_t e__ = (call);          \
		if (e__ != hipSuccess) return e__; \
	} while (0)

// order-preserving 3-bit code of a text byte (NUL ! # ; A C G T in byte order); 8 = not in the alphabet
__device__ __forceinline__ uint32_t order_code(uint32_t ch) {
	if (ch >= 'A') {
		const uint32_t x = ch & 6u, c = (x ^ (x >> 1)) >> 1; // A0 C1 G2 T3, as nt_code
		const bool known = ch == 'A' || ch == 'C' || ch == 'G' || ch == 'T';
		return known ? 4u + c : 8u;
	}
	return ch == 0 ? 0u : ch == '!' ? 1u : ch == '#' ? 2u : ch == ';' ? 3u : 8u;
}

// the first 16 symbols of suffix i (positions >= n read the NUL padding): their 2-bit codes, the first in the top bits, zeros
// from the first symbol on that is no nucleotide; jj: where that is (16: nowhere), sep: which (NUL 0 ! 1 # 2 ; 3); bad & 8:
// a byte outside the alphabet
struct First16 {
	uint32_t code, jj, sep, bad;
};
__device__ __forceinline__ First16 first16(const uint8_t *__restrict__ S, size_t i) {
	g_u8p p = (g_u8p)S + i;
	const uint64_t w0 = ld_u64_unaligned(p), w1 = ld_u64_unaligned(p + 8);
	First16 f;
	f.code = 0, f.jj = 16, f.sep = 0, f.bad = 0;
#pragma unroll
	for (int j = 0; j < 16; ++j) {
		const uint32_t c = order_code((uint32_t)((j < 8 ? w0 : w1) >> (8 * (j & 7))) & 0xffu);
		f.bad |= c;
		if (f.jj == 16 && c < 4u) f.jj = (uint32_t)j, f.sep = c;
		f.code = (f.code << 2) | (f.jj == 16 ? (c & 3u) : 0u);
	}
	return f;
}

// round 0: key of suffix i; the special suffixes are listed.  A thread makes the keys of FOUR consecutive suffixes from the 19 symbols they
// span (first16 per suffix classified every byte sixteen times: 200 vector instructions per suffix, 80 us per 9.8 M characters -- a sixth of
// the sorter's time; round 6: the bytes' codes once, the keys cut out of the 38 bits they fill)
__global__ __launch_bounds__(SA_BLOCK) void k_sa_keys0(const uint8_t *__restrict__ S, int32_t n, uint32_t *__restrict__ key,
														uint32_t *__restrict__ val, int32_t *__restrict__ foreign, uint32_t *__restrict__ special,
														uint32_t *__restrict__ nspecial) {
	const int64_t i0 = 4 * ((int64_t)blockIdx.x * SA_BLOCK + threadIdx.x);
	if (i0 >= n) return;
	g_u8p p = (g_u8p)S + i0; // (positions >= n read the NUL padding)
	const uint64_t w[3] = {ld_u64_unaligned(p), ld_u64_unaligned(p + 8), ld_u64_unaligned(p + 16)};
	uint64_t Y = 0;           // the 2-bit codes of the 19 symbols, the first on top
	uint32_t stop = 0, f8 = 0; // bit j: symbol j is no nucleotide (NUL ! # ;) / a byte outside the alphabet
#pragma unroll
	for (int j = 0; j < 19; ++j) {
		const uint32_t c = order_code((uint32_t)(w[j >> 3] >> (8 * (j & 7))) & 0xffu);
		Y = (Y << 2) | (c & 3u);
		stop |= (c < 4u ? 1u : 0u) << j, f8 |= (c == 8u ? 1u : 0u) << j;
	}
	uint32_t k[4];
#pragma unroll
	for (int s = 0; s < 4; ++s) {
		const uint32_t ms = (stop >> s) & 0xffffu, jj = ms ? (uint32_t)__builtin_ctz(ms) : 16u; // the first of the suffix's 16 symbols that is no nucleotide
		const uint32_t y = (uint32_t)(Y >> (2 * (3 - s)));
		k[s] = jj == 0 ? 0u : (jj < 16 ? y & (~0u << (2 * (16 - jj))) : y); // (zeros from that symbol on, as first16)
		if (i0 + s < n) {
			if ((f8 >> s) & 1u) *foreign = 1; // (a byte is reported by the suffix that starts with it)
			if (jj < 16) special[atomicAdd(nspecial, 1u)] = (uint32_t)(i0 + s);
		}
	}
	if (i0 + 3 < n) {
		*(uint4 *)(key + i0) = make_uint4(k[0], k[1], k[2], k[3]);
		*(uint4 *)(val + i0) = make_uint4((uint32_t)i0, (uint32_t)i0 + 1u, (uint32_t)i0 + 2u, (uint32_t)i0 + 3u);
	} else {
		for (int s = 0; s < 4 && i0 + s < n; ++s) key[i0 + s] = k[s], val[i0 + s] = (uint32_t)(i0 + s);
	}
}

// The record of a suffix for the scan index's builder (esa_build.hip: suffix_rec -- the 2-bit code of its first K
// characters, the number of leading nucleotides, the separator class behind them) and the up to four nucleotides behind
// the first K (rec2: their codes, the first in the low bits, | how many << 8), from the text
__device__ __forceinline__ uint32_t record_of(const uint8_t *__restrict__ S, size_t i, int K, uint32_t &r2) {
	g_u8p p = (g_u8p)S + i;
	const uint64_t w0 = ld_u64_unaligned(p), w1 = ld_u64_unaligned(p + 8);
	uint32_t y = 0, v = 16, sym_v = 0;
#pragma unroll
	for (int j = 0; j < 16; ++j) {
		const uint32_t c = order_code((uint32_t)((j < 8 ? w0 : w1) >> (8 * (j & 7))) & 0xffu) & 7u;
		if (v == 16 && c < 4u) v = (uint32_t)j, sym_v = c;
		y = (y << 2) | (c & 3u);
	}
	const uint32_t lite = (uint32_t)(SA_SYMS - K) < 4u ? (uint32_t)(SA_SYMS - K) : 4u;
	const uint32_t nval = v <= (uint32_t)K ? 0u : (v - (uint32_t)K < lite ? v - (uint32_t)K : lite);
	uint32_t codes = 0;
	for (uint32_t j = 0; j < nval; ++j) codes |= ((y >> (30 - 2 * ((uint32_t)K + j))) & 3u) << (2 * j);
	r2 = codes | (nval << 8);
	uint32_t sep = 0;
	if (v < (uint32_t)K) sep = sym_v == 1 ? 1u : (sym_v == 3 ? 2u : 3u);
	else v = (uint32_t)K;
	return ((y >> (32 - 2 * K)) << 6) | (sep << 4) | v;
}

// The sorted round-0 keys hold the first 16 symbols of every suffix IN SUFFIX-ARRAY ORDER: the records of all suffixes that
// are no specials (theirs are written from the text: k_sa_special_records and k_sa_apply) -- a sequential read for
// k_probe_table instead of one random access into the text per suffix.
__global__ __launch_bounds__(SA_BLOCK) void k_sa_records(const uint32_t *__restrict__ key, int32_t n, int K, uint32_t *__restrict__ rec, uint16_t *__restrict__ rec2) {
	const int64_t i = (int64_t)blockIdx.x * SA_BLOCK + threadIdx.x;
	if (i >= n) return;
	const uint32_t y = key[i];
	if (rec2) {
		const uint32_t nval = (uint32_t)(SA_SYMS - K) < 4u ? (uint32_t)(SA_SYMS - K) : 4u;
		uint32_t codes = 0;
		for (uint32_t j = 0; j < nval; ++j) codes |= ((y >> (30 - 2 * ((uint32_t)K + j))) & 3u) << (2 * j);
		rec2[i] = (uint16_t)(codes | (nval << 8));
	}
	rec[i] = ((y >> (32 - 2 * K)) << 6) | (uint32_t)K;
}

// ... of the specials whose key is theirs alone (they are in place: found by their key); those that tied get theirs when
// they are placed
__global__ __launch_bounds__(SA_BLOCK) void k_sa_special_records(const uint8_t *__restrict__ S, const uint32_t *__restrict__ special,
																  const uint32_t *__restrict__ nspecial, const uint32_t *__restrict__ key, int32_t n, int K,
																  uint32_t *__restrict__ rec, uint16_t *__restrict__ rec2) {
	const uint32_t ns = *nspecial;
	for (uint32_t t = blockIdx.x * SA_BLOCK + threadIdx.x; t < ns; t += gridDim.x * SA_BLOCK) {
		const uint32_t i = special[t], k = first16(S, i).code;
		uint32_t lo = 0, hi = (uint32_t)n; // the first slot whose key is >= k
		while (lo < hi) {
			const uint32_t mid = lo + ((hi - lo) >> 1);
			if (key[mid] < k) lo = mid + 1; else hi = mid;
		}
		if (lo + 1 < (uint32_t)n && key[lo + 1] == k) continue;
		uint32_t r2;
		rec[lo] = record_of(S, i, K, r2);
		if (rec2) rec2[lo] = (uint16_t)r2;
	}
}

// round 0, behind the sort: the slots that share their key with a neighbour, in order, as (position << 32 | suffix).  Three
// steps: a count per block of SA_TILE slots, the exclusive sums of the counts, the slots written.
__device__ __forceinline__ bool open0(const uint32_t *__restrict__ key, uint32_t n, uint32_t t) {
	if (t >= n) return false;
	const uint32_t k = key[t];
	return (t > 0 && key[t - 1] == k) || (t + 1 < n && key[t + 1] == k);
}
__global__ __launch_bounds__(SA_BLOCK) void k_sa_open_count(const uint32_t *__restrict__ key, uint32_t n, uint32_t *__restrict__ bcount) {
	__shared__ uint32_t s_sum;
	if (threadIdx.x == 0) s_sum = 0;
	__syncthreads();
	uint32_t c = 0;
#pragma unroll
	for (uint32_t k = 0; k < SA_TILE / SA_BLOCK; ++k) c += (uint32_t)__builtin_popcountll(__ballot(open0(key, n, blockIdx.x * SA_TILE + k * SA_BLOCK + threadIdx.x)));
	if ((threadIdx.x & 63u) == 0) atomicAdd(&s_sum, c);
	__syncthreads();
	if (threadIdx.x == 0) bcount[blockIdx.x] = s_sum;
}
__global__ void k_sa_open_total(const uint32_t *__restrict__ bcount, const uint32_t *__restrict__ boff, uint32_t nb, uint32_t *__restrict__ total) {
	*total = boff[nb - 1] + bcount[nb - 1];
}
__global__ __launch_bounds__(SA_BLOCK) void k_sa_open_write(const uint32_t *__restrict__ key, const int32_t *__restrict__ SA, uint32_t n,
															 const uint32_t *__restrict__ boff, uint64_t *__restrict__ slots) {
	__shared__ uint32_t s_wave[SA_TILE / 64];
	const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
	bool mine[SA_TILE / SA_BLOCK];
	uint64_t b[SA_TILE / SA_BLOCK];
#pragma unroll
	for (uint32_t k = 0; k < SA_TILE / SA_BLOCK; ++k) {
		mine[k] = open0(key, n, blockIdx.x * SA_TILE + k * SA_BLOCK + threadIdx.x);
		b[k] = __ballot(mine[k]);
		if (lane == 0) s_wave[k * (SA_BLOCK / 64) + wave] = (uint32_t)__builtin_popcountll(b[k]);
	}
	__syncthreads();
	uint32_t before = boff[blockIdx.x];
#pragma unroll
	for (uint32_t k = 0; k < SA_TILE / SA_BLOCK; ++k) {
		for (uint32_t w = 0; w < SA_BLOCK / 64; ++w) {
			const uint32_t cnt = s_wave[k * (SA_BLOCK / 64) + w];
			if (w == wave && mine[k]) {
				const uint32_t t = blockIdx.x * SA_TILE + k * SA_BLOCK + threadIdx.x;
				slots[before + (uint32_t)__builtin_popcountll(b[k] & ((1ull << lane) - 1ull))] = ((uint64_t)t << 32) | (uint32_t)SA[t];
			}
			before += cnt;
		}
	}
}

// ... heads of their groups: hv[t] = position + 1 of slot t if its round-0 key differs from its predecessor's (a max-scan spreads it)
__global__ __launch_bounds__(SA_BLOCK) void k_sa_heads0(const uint32_t *__restrict__ key0, const uint64_t *__restrict__ slots, uint32_t m,
														 uint32_t *__restrict__ hv) {
	const uint32_t t = blockIdx.x * SA_BLOCK + threadIdx.x;
	if (t >= m) return;
	const uint32_t pos = (uint32_t)(slots[t] >> 32);
	hv[t] = (pos == 0 || key0[pos] != key0[pos - 1]) ? pos + 1 : 0;
}

// round 1: key of an open slot = (its group, where its first non-nucleotide is -- 16: not among the first 16 --, which it is,
// the `syms` symbols of the text behind that one, or behind the first 16; 3 bits each).  Within a group of equal round-0 keys
// that is the suffixes' order: a special's zeros stand for a separator where the others have A.
__global__ __launch_bounds__(SA_BLOCK) void k_sa_keys_text(const uint64_t *__restrict__ slots, const uint32_t *__restrict__ grp, uint32_t m,
															const uint8_t *__restrict__ S, uint32_t syms, uint64_t *__restrict__ key,
															uint32_t *__restrict__ val) {
	const uint32_t t = blockIdx.x * SA_BLOCK + threadIdx.x;
	if (t >= m) return;
	const uint32_t idx = (uint32_t)slots[t];
	const First16 f = first16(S, idx);
	const uint32_t off = f.jj < 16 ? f.jj + 1 : 16u;
	g_u8p p = (g_u8p)S + (size_t)idx + off; // (behind the text: its zero padding, ANDI_PAD bytes)
	const uint64_t w0 = ld_u64_unaligned(p), w1 = ld_u64_unaligned(p + 8);
	uint64_t k = ((uint64_t)grp[t] << 7) | (f.jj << 2) | f.sep; // (grp: head position + 1)
	for (uint32_t j = 0; j < syms; ++j) {
		const uint64_t w = j < 8 ? w0 : w1;
		k = (k << 3) | (order_code((uint32_t)(w >> (8 * (j & 7))) & 0xffu) & 7u);
	}
	key[t] = k;
	val[t] = idx;
}

// heads of the groups of equal keys among m sorted slots; slot t lies at position pos(t) of the
// suffix array: hv[t] = position of the head if t is one, else 0 (a max-scan spreads it)
__global__ __launch_bounds__(SA_BLOCK) void k_sa_heads(const uint64_t *__restrict__ key, const uint64_t *__restrict__ slots,
														uint32_t m, uint32_t *__restrict__ hv) {
	const uint32_t t = blockIdx.x * SA_BLOCK + threadIdx.x;
	if (t >= m) return;
	const bool head = t == 0 || key[t] != key[t - 1];
	hv[t] = head ? (uint32_t)(slots[t] >> 32) + 1 : 0; // + 1: position 0 must win the max-scan too
}

// place the sorted suffixes (with their records, from the text: a group's suffixes need not share what the records hold),
// give them their group (the head's position) as rank, and mark the slots whose group has more than one member
// (*special_open: one of those is a special suffix -- the round behind round 1 then counts on fewer sorted symbols)
__global__ __launch_bounds__(SA_BLOCK) void k_sa_apply(const uint64_t *__restrict__ key, const uint32_t *__restrict__ val,
														const uint64_t *__restrict__ slots, const uint32_t *__restrict__ grp,
														uint32_t m, int32_t *__restrict__ SA, uint32_t *__restrict__ rank,
														uint64_t *__restrict__ slots_out, uint8_t *__restrict__ open, const uint8_t *__restrict__ S, int K,
														uint32_t *__restrict__ rec, uint16_t *__restrict__ rec2, int32_t *__restrict__ special_open) {
	const uint32_t t = blockIdx.x * SA_BLOCK + threadIdx.x;
	if (t >= m) return;
	const uint32_t pos = (uint32_t)(slots[t] >> 32), idx = val[t];
	SA[pos] = (int32_t)idx;
	if (rank) rank[idx] = grp[t] - 1;
	if (rec) {
		uint32_t r2;
		rec[pos] = record_of(S, idx, K, r2);
		if (rec2) rec2[pos] = (uint16_t)r2;
	}
	const bool head = t == 0 || key[t] != key[t - 1];
	const bool next_head = t + 1 == m || key[t + 1] != key[t];
	const bool is_open = !(head && next_head);
	open[t] = is_open ? 1 : 0;
	slots_out[t] = ((uint64_t)pos << 32) | idx;
	if (special_open && is_open && first16(S, idx).jj < 16) *special_open = 1;
}

// rounds r >= 2: key of an open slot = (its group, the group of the suffix h symbols further on)
__global__ __launch_bounds__(SA_BLOCK) void k_sa_keys(const uint64_t *__restrict__ slots, uint32_t m, const uint32_t *__restrict__ rank,
													   int32_t n, uint32_t h, int bits, uint64_t *__restrict__ key,
													   uint32_t *__restrict__ val) {
	const uint32_t t = blockIdx.x * SA_BLOCK + threadIdx.x;
	if (t >= m) return;
	const uint32_t idx = (uint32_t)slots[t];
	const uint64_t far = (uint64_t)idx + h < (uint64_t)n ? (uint64_t)rank[idx + h] + 1 : 0; // past the end: before everything
	key[t] = ((uint64_t)rank[idx] << bits) | far;
	val[t] = idx;
}

// the ranks of all suffixes as they stand when the doubling rounds begin: a suffix's position (the groups still open are
// written behind this, k_sa_ranks)
__global__ __launch_bounds__(SA_BLOCK) void k_sa_rank_fill(const int32_t *__restrict__ SA, int32_t n, uint32_t *__restrict__ rank) {
	const int64_t i = (int64_t)blockIdx.x * SA_BLOCK + threadIdx.x;
	if (i < n) rank[SA[i]] = (uint32_t)i;
}
__global__ __launch_bounds__(SA_BLOCK) void k_sa_ranks(const uint32_t *__restrict__ val, const uint32_t *__restrict__ grp, uint32_t m,
														uint32_t *__restrict__ rank) {
	const uint32_t t = blockIdx.x * SA_BLOCK + threadIdx.x;
	if (t < m) rank[val[t]] = grp[t] - 1;
}

} // namespace

size_t andi_sa_device_workspace(int32_t n) {
	size_t sort_t = 0, sort0_t = 0, scan_t = 0, sel_t = 0;
	uint64_t *k = nullptr;
	uint32_t *v = nullptr;
	uint8_t *f = nullptr;
	(void)hipcub::DeviceRadixSort::SortPairs(nullptr, sort_t, k, k, v, v, n, 0, 64, (hipStream_t)0);
	(void)hipcub::DeviceRadixSort::SortPairs(nullptr, sort0_t, v, v, v, v, n, 0, 32, (hipStream_t)0);
	(void)hipcub::DeviceScan::InclusiveScan(nullptr, scan_t, v, v, hipcub::Max(), n, (hipStream_t)0);
	(void)hipcub::DeviceSelect::Flagged(nullptr, sel_t, k, f, k, v, n, (hipStream_t)0);
	size_t tmp = std::max(std::max(sort_t, sort0_t), std::max(scan_t, sel_t));
	tmp = (tmp + 255) & ~(size_t)255;
	const size_t N = ((size_t)n + 63) & ~(size_t)63;
	// key x2, val x2, slots x2, rank, hv, grp, open, count, temp
	return tmp + N * (8 * 2 + 4 * 2 + 8 * 2 + 4 + 4 + 4 + 1) + 4096;
}

hipError_t andi_sa_device(const uint8_t *S, int32_t n, int32_t *SA, void *workspace, size_t workspace_bytes,
						  int32_t *h_pinned4, hipStream_t st, int *rounds_out, uint32_t *rec, int recK, uint16_t *rec2) {
	if (n <= 0) return hipSuccess;
	if (workspace_bytes < andi_sa_device_workspace(n)) return hipErrorInvalidValue;
	const size_t N = ((size_t)n + 63) & ~(size_t)63;
	char *p = (char *)workspace;
	auto take = [&](size_t bytes) {
		char *r = p;
		p += (bytes + 255) & ~(size_t)255;
		return r;
	};
	uint64_t *keyA = (uint64_t *)take(N * 8), *keyB = (uint64_t *)take(N * 8);
	uint32_t *valA = (uint32_t *)take(N * 4), *valB = (uint32_t *)take(N * 4);
	uint64_t *slotA = (uint64_t *)take(N * 8), *slotB = (uint64_t *)take(N * 8);
	uint32_t *rank = (uint32_t *)take(N * 4), *hv = (uint32_t *)take(N * 4), *grp = (uint32_t *)take(N * 4);
	uint8_t *open = (uint8_t *)take(N);
	uint32_t *d_count = (uint32_t *)take(256); // [0] open slots, [1] a byte outside the alphabet, [2] an open slot is a special suffix, [3] special suffixes
	int32_t *d_foreign = (int32_t *)(d_count + 1), *d_special_open = (int32_t *)(d_count + 2);
	uint32_t *d_nspecial = d_count + 3;
	void *tmp = p;
	size_t tmp_bytes = workspace_bytes - (size_t)(p - (char *)workspace);

	int bits = 1; // of a rank + 1
	while (((uint64_t)1 << bits) < (uint64_t)n + 2) ++bits;
	auto blocks = [](uint32_t m) { return (m + SA_BLOCK - 1) / SA_BLOCK; };
	auto read_count = [&](uint32_t &m) -> hipError_t { // open slots, foreign flag, an open special
		SA_TRY(hipMemcpyAsync(h_pinned4, d_count, 16, hipMemcpyDeviceToHost, st));
		SA_TRY(hipStreamSynchronize(st));
		if (h_pinned4[1]) return hipErrorInvalidSymbol; // a byte outside the alphabet: the caller reports it (the scan refuses such a text anyway)
		m = (uint32_t)h_pinned4[0];
		return hipSuccess;
	};
	int rounds = 0;
	uint32_t m = 0, h = SA_SYMS;
	// ---- round 0: 32-bit keys; the sorted indices go straight into SA; the slots that share their key are picked out of the keys
	{
		uint32_t *key0A = (uint32_t *)keyA, *key0B = (uint32_t *)keyB, *special = hv, *bcount = grp, *boff = rank; // (buffers of the later rounds)
		const uint32_t nb = ((uint32_t)n + SA_TILE - 1) / SA_TILE;
		SA_TRY(hipMemsetAsync(d_count, 0, 64, st));
		k_sa_keys0<<<blocks(((uint32_t)n + 3u) / 4u), SA_BLOCK, 0, st>>>(S, n, key0A, valA, d_foreign, special, d_nspecial);
		SA_TRY(hipGetLastError());
		size_t tb = tmp_bytes;
		SA_TRY(hipcub::DeviceRadixSort::SortPairs(tmp, tb, key0A, key0B, valA, (uint32_t *)SA, n, 0, 32, st));
		if (rec) {
			k_sa_records<<<blocks((uint32_t)n), SA_BLOCK, 0, st>>>(key0B, n, recK, rec, rec2);
			k_sa_special_records<<<64, SA_BLOCK, 0, st>>>(S, special, d_nspecial, key0B, n, recK, rec, rec2);
		}
		k_sa_open_count<<<nb, SA_BLOCK, 0, st>>>(key0B, (uint32_t)n, bcount);
		SA_TRY(hipGetLastError());
		tb = tmp_bytes;
		SA_TRY(hipcub::DeviceScan::ExclusiveSum(tmp, tb, bcount, boff, (int)nb, st));
		k_sa_open_total<<<1, 1, 0, st>>>(bcount, boff, nb, d_count);
		k_sa_open_write<<<nb, SA_BLOCK, 0, st>>>(key0B, SA, (uint32_t)n, boff, slotB);
		SA_TRY(hipGetLastError());
		SA_TRY(read_count(m));
		++rounds;
	}
	// ---- round 1: the text behind the first SA_SYMS symbols (behind a special's separator)
	if (m) {
		const uint32_t room = (uint32_t)(64 - 7 - bits) / 3u, syms = room < 16u ? room : 16u;
		k_sa_heads0<<<blocks(m), SA_BLOCK, 0, st>>>((const uint32_t *)keyB, slotB, m, hv);
		SA_TRY(hipGetLastError());
		size_t tb = tmp_bytes;
		SA_TRY(hipcub::DeviceScan::InclusiveScan(tmp, tb, hv, grp, hipcub::Max(), (int)m, st));
		k_sa_keys_text<<<blocks(m), SA_BLOCK, 0, st>>>(slotB, grp, m, S, syms, keyA, valA);
		SA_TRY(hipGetLastError());
		tb = tmp_bytes;
		SA_TRY(hipcub::DeviceRadixSort::SortPairs(tmp, tb, keyA, keyB, valA, valB, (int)m, 0, bits + 7 + 3 * (int)syms, st));
		k_sa_heads<<<blocks(m), SA_BLOCK, 0, st>>>(keyB, slotB, m, hv);
		SA_TRY(hipGetLastError());
		tb = tmp_bytes;
		SA_TRY(hipcub::DeviceScan::InclusiveScan(tmp, tb, hv, grp, hipcub::Max(), (int)m, st));
		k_sa_apply<<<blocks(m), SA_BLOCK, 0, st>>>(keyB, valB, slotB, grp, m, SA, nullptr, slotA, open, S, recK, rec, rec2, d_special_open);
		SA_TRY(hipGetLastError());
		tb = tmp_bytes;
		SA_TRY(hipcub::DeviceSelect::Flagged(tmp, tb, slotA, open, slotB, d_count, (int)m, st));
		const uint32_t m1 = m;
		SA_TRY(read_count(m));
		++rounds;
		// what every group still open is sorted by: 16 + syms symbols, a special's group by those behind its separator at least
		h = h_pinned4[2] ? syms + 1 : SA_SYMS + syms;
		if (m) { // repeats: the doubling rounds want every suffix's rank
			k_sa_rank_fill<<<blocks((uint32_t)n), SA_BLOCK, 0, st>>>(SA, n, rank);
			k_sa_ranks<<<blocks(m1), SA_BLOCK, 0, st>>>(valB, grp, m1, rank);
			SA_TRY(hipGetLastError());
		}
	}
	// ---- rounds r >= 2: (group, group h further on)
	while (m) {
		k_sa_keys<<<blocks(m), SA_BLOCK, 0, st>>>(slotB, m, rank, n, h, bits, keyA, valA);
		SA_TRY(hipGetLastError());
		size_t tb = tmp_bytes;
		SA_TRY(hipcub::DeviceRadixSort::SortPairs(tmp, tb, keyA, keyB, valA, valB, (int)m, 0, 2 * bits, st));
		k_sa_heads<<<blocks(m), SA_BLOCK, 0, st>>>(keyB, slotB, m, hv);
		SA_TRY(hipGetLastError());
		tb = tmp_bytes;
		SA_TRY(hipcub::DeviceScan::InclusiveScan(tmp, tb, hv, grp, hipcub::Max(), (int)m, st));
		k_sa_apply<<<blocks(m), SA_BLOCK, 0, st>>>(keyB, valB, slotB, grp, m, SA, rank, slotA, open, S, recK, rec, rec2, nullptr);
		SA_TRY(hipGetLastError());
		tb = tmp_bytes;
		SA_TRY(hipcub::DeviceSelect::Flagged(tmp, tb, slotA, open, slotB, d_count, (int)m, st));
		SA_TRY(read_count(m));
		++rounds;
		if (h > (1u << 30)) return hipErrorUnknown; // (cannot happen: distinct suffixes differ within n symbols)
		h *= 2;
	}
	if (rounds_out) *rounds_out = rounds;
	return hipSuccess;
}
