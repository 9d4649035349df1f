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
//   round 0   sort all suffixes by their first SA_SYMS = 16 symbols (3 bits each: the text's alphabet
//             NUL ! # ; A C G T in byte order, 48-bit keys; 21 symbols in 63 bits until round 3);
//   round r   a suffix i that still shares its place with others is keyed by
//             (group of i, group of i + h), h = 16 * 2^(r-1): sorting those keys orders it by
//             2h symbols.  Groups are contiguous in the array and keep their span, so the
//             sorted suffixes go back to the slots the unsorted ones came from.
// Random-like DNA is done after round 1 (a few suffixes in a thousand share 16 symbols); repeats take log2(length / 16) more
// rounds over the suffixes inside them only.  Sorting, scans and compaction are rocPRIM's
// (hipcub front end); the kernels here make keys, group heads and ranks.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <algorithm>

#include "andi_dev.h"
#include "sa_device.h"

namespace {

#define SA_BLOCK 256
// symbols of a round-0 key, 3 bits each.  16 (48-bit keys: six 8-bit passes of the radix sort instead of eight) still
// holds the K <= 13 symbols the scan index's records want; in random-like DNA of 10^7 characters 0.2 % of the
// suffixes then share their key with another and go through round 1 (21 symbols: a dozen suffixes -- and two more passes
// over all of them: 1.82 -> 1.67 ms per 9.8 M characters as the bench set's staging has them; 13 symbols, five passes: 1.84)
#ifndef SA_SYMS
#define SA_SYMS 16
#endif
#define SA_KEY_BITS (3 * SA_SYMS)
#define SA_TRY(call)                      \
	do {                                  \
		hipError_t e__ = (call);          \
		if (e__ != hipSuccess) return e__; \
	} while (0)

// order-preserving 3-bit code of a text byte; 8 = not in the alphabet
__device__ __forceinline__ uint32_t order_code(uint32_t ch) {
	if (ch >= 'A') {
		const uint32_t x = ch & 6u, c = (x ^ (x >> 1)) >> 1; // A0 C1 G2 T3, as nt_code
		const bool known = ch == 'A' || ch == 'C' || ch == 'G' || ch == 'T';
		return known ? 4u + c : 8u;
	}
	return ch == 0 ? 0u : ch == '!' ? 1u : ch == '#' ? 2u : ch == ';' ? 3u : 8u;
}

// round 0: key of suffix i = its first SA_SYMS symbols (positions >= n read the NUL padding: code 0,
// below every symbol, so a suffix that is a prefix of another sorts first)
__global__ __launch_bounds__(SA_BLOCK) void k_sa_keys0(const uint8_t *__restrict__ S, int32_t n, uint64_t *__restrict__ key,
														uint32_t *__restrict__ val, int32_t *__restrict__ foreign) {
	const int64_t i = (int64_t)blockIdx.x * SA_BLOCK + threadIdx.x;
	if (i >= n) return;
	g_u8p p = (g_u8p)S + i;
	const uint64_t w0 = ld_u64_unaligned(p), w1 = ld_u64_unaligned(p + 8), w2 = SA_SYMS > 16 ? ld_u64_unaligned(p + 16) : 0ull;
	uint64_t k = 0;
	uint32_t bad = 0;
#pragma unroll
	for (int j = 0; j < SA_SYMS; ++j) {
		const uint64_t w = j < 8 ? w0 : j < 16 ? w1 : w2;
		const uint32_t c = order_code((uint32_t)(w >> (8 * (j & 7))) & 0xffu);
		bad |= c;
		k = (k << 3) | (c & 7u);
	}
	if ((bad & 8u) && order_code(S[i]) == 8u) *foreign = 1; // (a byte is reported by the suffix that starts with it)
	key[i] = k;
	val[i] = (uint32_t)i;
}

// heads of the groups of equal keys among m sorted slots; slot t lies at position pos(t) of the
// suffix array: hv[t] = position of the head if t is one, else 0 (a max-scan spreads it)
__global__ __launch_bounds__(SA_BLOCK) void k_sa_heads(const uint64_t *__restrict__ key, const uint64_t *__restrict__ slots,
														uint32_t m, uint32_t *__restrict__ hv) {
	const uint32_t t = blockIdx.x * SA_BLOCK + threadIdx.x;
	if (t >= m) return;
	const bool head = t == 0 || key[t] != key[t - 1];
	const uint32_t pos = slots ? (uint32_t)(slots[t] >> 32) : t;
	hv[t] = head ? pos + 1 : 0; // + 1: position 0 must win the max-scan too
}

// place the sorted suffixes, give them their group (the head's position) as rank, and mark the
// slots whose group has more than one member
__global__ __launch_bounds__(SA_BLOCK) void k_sa_apply(const uint64_t *__restrict__ key, const uint32_t *__restrict__ val,
														const uint64_t *__restrict__ slots, const uint32_t *__restrict__ grp,
														uint32_t m, int32_t *__restrict__ SA, uint32_t *__restrict__ rank,
														uint64_t *__restrict__ slots_out, uint8_t *__restrict__ open) {
	const uint32_t t = blockIdx.x * SA_BLOCK + threadIdx.x;
	if (t >= m) return;
	const uint32_t pos = slots ? (uint32_t)(slots[t] >> 32) : t, idx = val[t];
	SA[pos] = (int32_t)idx;
	rank[idx] = grp[t] - 1;
	const bool head = t == 0 || key[t] != key[t - 1];
	const bool next_head = t + 1 == m || key[t + 1] != key[t];
	open[t] = (head && next_head) ? 0 : 1;
	slots_out[t] = ((uint64_t)pos << 32) | idx;
}

// The sorted round-0 keys hold the first SA_SYMS symbols of every suffix IN SUFFIX-ARRAY ORDER.  The scan index's
// builder (esa_build.hip: k_probe_table) wants, per suffix, the 2-bit code of its first K characters, the number of
// leading nucleotides and the separator class behind them -- its record, suffix_rec there -- which it would
// otherwise gather from the text, one random access per suffix.  Made here from the keys, the records are a
// sequential read for it.  (Symbol codes: NUL 0, '!' 1, '#' 2, ';' 3, A C G T 4..7; symbol j in bits 62-3j..60-3j.)
__global__ __launch_bounds__(SA_BLOCK) void k_sa_records(const uint64_t *__restrict__ key, int32_t n, int K, uint32_t *__restrict__ rec, uint16_t *__restrict__ rec2) {
	const int64_t i = (int64_t)blockIdx.x * SA_BLOCK + threadIdx.x;
	if (i >= n) return;
	const uint64_t k = key[i] << (63 - SA_KEY_BITS); // symbol j in bits 62-3j..60-3j, as with 21 symbols
	const uint64_t other = ~k & (0x4924924924924924ull & ~((1ull << (63 - SA_KEY_BITS)) - 1ull)); // top bit of a symbol clear: not a nucleotide
	uint32_t v = other ? ((uint32_t)__builtin_clzll(other) - 1u) / 3u : (uint32_t)SA_SYMS;
	if (rec2) { // the nucleotides behind the first K symbols, as far as the key holds them (four at most): how many, their codes (first in the low bits)
		const uint32_t lite = (uint32_t)(SA_SYMS - K) < 4u ? (uint32_t)(SA_SYMS - K) : 4u;
		const uint32_t nval = v <= (uint32_t)K ? 0u : (v - (uint32_t)K < lite ? v - (uint32_t)K : lite);
		uint32_t codes = 0;
		for (uint32_t j = 0; j < nval; ++j) codes |= ((uint32_t)(k >> (60 - 3 * ((uint32_t)K + j))) & 3u) << (2 * j);
		rec2[i] = (uint16_t)(codes | (nval << 8));
	}
	uint32_t sep = 0;
	if (v < (uint32_t)K) {
		const uint32_t sym = (uint32_t)(k >> (60 - 3 * v)) & 7u;
		sep = sym == 1 ? 1u : (sym == 3 ? 2u : 3u);
	} else {
		v = (uint32_t)K;
	}
	uint32_t y = 0; // 2-bit codes of the first 16 symbols, the first in the top two bits
#pragma unroll
	for (int j = 0; j < 16; ++j) y |= ((uint32_t)(k >> (60 - 3 * j)) & 3u) << (30 - 2 * j);
	rec[i] = ((y >> (32 - 2 * K)) << 6) | (sep << 4) | v;
}

// round r >= 1: key of an open slot = (its group, the group of the suffix h symbols further on)
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

} // namespace

size_t andi_sa_device_workspace(int32_t n) {
	size_t sort_t = 0, scan_t = 0, sel_t = 0;
	uint64_t *k = nullptr;
	uint32_t *v = nullptr;
	uint8_t *f = nullptr;
	(void)hipcub::DeviceRadixSort::SortPairs(nullptr, sort_t, k, k, v, v, n, 0, 64, (hipStream_t)0);
	(void)hipcub::DeviceScan::InclusiveScan(nullptr, scan_t, v, v, hipcub::Max(), n, (hipStream_t)0);
	(void)hipcub::DeviceSelect::Flagged(nullptr, sel_t, k, f, k, v, n, (hipStream_t)0);
	size_t tmp = std::max(sort_t, std::max(scan_t, sel_t));
	tmp = (tmp + 255) & ~(size_t)255;
	const size_t N = ((size_t)n + 63) & ~(size_t)63;
	// key x2, val x2, slots x2, rank, hv, grp, open, count, temp
	return tmp + N * (8 * 2 + 4 * 2 + 8 * 2 + 4 + 4 + 4 + 1) + 4096;
}

hipError_t andi_sa_device(const uint8_t *S, int32_t n, int32_t *SA, void *workspace, size_t workspace_bytes,
						  int32_t *h_pinned2, hipStream_t st, int *rounds_out, uint32_t *rec, int recK, uint16_t *rec2) {
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
	uint32_t *d_count = (uint32_t *)take(256);
	int32_t *d_foreign = (int32_t *)(d_count + 1);
	void *tmp = p;
	size_t tmp_bytes = workspace_bytes - (size_t)(p - (char *)workspace);

	int bits = 1; // of a rank + 1
	while (((uint64_t)1 << bits) < (uint64_t)n + 2) ++bits;
	auto blocks = [](uint32_t m) { return (m + SA_BLOCK - 1) / SA_BLOCK; };

	SA_TRY(hipMemsetAsync(d_count, 0, 64, st));
	k_sa_keys0<<<blocks((uint32_t)n), SA_BLOCK, 0, st>>>(S, n, keyA, valA, d_foreign);
	SA_TRY(hipGetLastError());
	uint32_t m = (uint32_t)n, h = SA_SYMS;
	const uint64_t *slots = nullptr; // round 0: slot t is position t
	int rounds = 0;
	for (;;) {
		size_t tb = tmp_bytes;
		SA_TRY(hipcub::DeviceRadixSort::SortPairs(tmp, tb, keyA, keyB, valA, valB, (int)m, 0, rounds == 0 ? SA_KEY_BITS : 2 * bits, st));
		if (rounds == 0 && rec) k_sa_records<<<blocks(m), SA_BLOCK, 0, st>>>(keyB, n, recK, rec, rec2);
		k_sa_heads<<<blocks(m), SA_BLOCK, 0, st>>>(keyB, slots, m, hv);
		SA_TRY(hipGetLastError());
		tb = tmp_bytes;
		SA_TRY(hipcub::DeviceScan::InclusiveScan(tmp, tb, hv, grp, hipcub::Max(), (int)m, st));
		k_sa_apply<<<blocks(m), SA_BLOCK, 0, st>>>(keyB, valB, slots, grp, m, SA, rank, slotA, open);
		SA_TRY(hipGetLastError());
		tb = tmp_bytes;
		SA_TRY(hipcub::DeviceSelect::Flagged(tmp, tb, slotA, open, slotB, d_count, (int)m, st));
		SA_TRY(hipMemcpyAsync(h_pinned2, d_count, 8, hipMemcpyDeviceToHost, st)); // open slots, foreign flag
		SA_TRY(hipStreamSynchronize(st));
		++rounds;
		if (h_pinned2[1]) return hipErrorInvalidSymbol; // a byte outside the alphabet: the caller reports it (the scan refuses such a text anyway)
		m = (uint32_t)h_pinned2[0];
		if (m == 0) break;
		slots = slotB;
		k_sa_keys<<<blocks(m), SA_BLOCK, 0, st>>>(slotB, m, rank, n, h, bits, keyA, valA);
		SA_TRY(hipGetLastError());
		if (h > (1u << 30)) return hipErrorUnknown; // (cannot happen: distinct suffixes differ within n symbols)
		h *= 2;
	}
	if (rounds_out) *rounds_out = rounds;
	return hipSuccess;
}
