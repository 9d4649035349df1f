// esa_build.h — launch interface of the device index build (esa_build.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define ANDI_MIN_LEVELS 6 /* 64^5 > 2^30 covers every int32-indexable text */

// 64-ary pyramid of minima over LCP[0..n]; lv[0] is LCP itself.
struct MinTree {
	const int32_t *lv[ANDI_MIN_LEVELS];
	int32_t cnt[ANDI_MIN_LEVELS];
	int levels;
};

struct EsaBuildArgs {
	const uint8_t *S;   // n + 1 + pad
	const int32_t *SA;  // n
	int32_t *LCP;       // n + 1   (out)
	int32_t *CLD;       // n + 1   (out; doubles as PLCP scratch)
	uint8_t *FVC;       // n       (out)
	int4 *tab;          // 4^10    (out)
	uint2 *deep;        // 4^deepK (out)
	uint8_t *N0, *N1;   // 4-bit symbols of the text, two alignments (out; see andi_dev.h)
	uint32_t *P;        // the text bit-sliced (out; EsaDev.P): what the wavefront kernels stream
	int32_t *flags;     // 4 ints  (out)
	const uint32_t *rec; // the suffixes' records in suffix-array order if the device sorter made them (sa_device.hip), else null
	const uint16_t *rec2; // ... and the symbols behind their first deepK (the short extended form of DEEP_SINGLE), or null
	int32_t deepK;
	int32_t *min_scratch; // andi_min_tree_entries(n) ints
	int32_t n;
};

// one subject of a batched scan-index build
struct AndiIndexBatchItem {
	const uint8_t *S;  // text (pack_symbols' source)
	const int32_t *SA;
	uint2 *deep;
	uint8_t *N0, *N1;
	uint32_t *P; // as EsaBuildArgs.P
	int32_t *flags;
	const uint32_t *rec; // as EsaBuildArgs.rec
	const uint16_t *rec2;
	int32_t n, deepK;
	int32_t single_ext; // the form of this subject's entries of K-mers that occur once (andi_index_single_ext)
};

size_t andi_min_tree_entries(int32_t n);
// the form of the entries of K-mers that occur once the index builds launched now write (andi_dev.h: DEEP_SINGLE):
// 0 plain, 1 extended (13 symbols behind the occurrence, gathered from the text), 2 short extended (up to 4, from the
// device sorter's keys: no gather; subjects whose suffix array came from the host get plain entries)
int andi_index_single_ext(size_t queries, bool sorted_on_device);
// the scan indexes of `count` subjects (device array of items) in two launches; max_n = the longest text
hipError_t andi_launch_index_build_batch(const AndiIndexBatchItem *d_items, uint32_t count, int32_t max_n, hipStream_t st);
// (scan_lane.hip) packed symbols of the items' texts, `bytes` source bytes each at most (shorter texts stop at their own end)
hipError_t andi_launch_pack_symbols_batch(const AndiIndexBatchItem *d_items, uint32_t count, size_t bytes, hipStream_t st);
hipError_t andi_launch_pack_planes_batch(const AndiIndexBatchItem *d_items, uint32_t count, size_t max_n, hipStream_t st); // N0 -> P of every item
// reference arrays LCP, CLD, FVC, tab (esa_init_LCP/_CLD/_FVC/_cache)
hipError_t andi_launch_esa_build(const EsaBuildArgs &a, hipStream_t st);
// scan index: deep, side, flags from S and SA alone
hipError_t andi_launch_index_build(const EsaBuildArgs &a, int single_ext, hipStream_t st);
// seq_subject_init for a sequence that already lies in device memory (the query pool): RS = revcomp(q) '#' q into a subject's
// text buffer (the caller zeroes what lies behind 2 len + 1); the number of G and C of every sequence of a pool (calc_gc)
hipError_t andi_launch_rs_from_query(uint8_t *RS, const uint8_t *q, uint32_t len, hipStream_t st);
hipError_t andi_launch_gc_counts(const uint8_t *pool, const uint64_t *d_off, const uint32_t *d_len, uint32_t nq, uint32_t longest,
								 unsigned long long *d_counts, hipStream_t st);
// contig separators ('!') per sequence of a pool (counts zeroed by the caller)
hipError_t andi_launch_sep_counts(const uint8_t *pool, const uint64_t *d_off, const uint32_t *d_len, uint32_t nq, uint32_t longest,
								  uint32_t *d_counts, hipStream_t st);
