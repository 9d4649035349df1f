// esa_build.hip — enhanced-suffix-array construction on the device (gfx950).
//
// Input: RS and its suffix array, already in HBM.  Output: LCP, CLD, FVC and
// the 10-mer interval table, i.e. everything esa_init() builds after
// esa_init_SA (src/esa.c:254-277).  All kernels are HBM/L2-bound integer
// gathers and scatters; none is GEMM-shaped.
//
//   K1a phi_scatter   PHI[SA[r]] = SA[r-1]                 (src/esa.c:396-400)
//   K1b plcp_chunks   permuted LCP, Kasai amortisation restarted per chunk
//                                                           (src/esa.c:402-417)
//   K1c lcp_fvc       LCP[r] = PLCP[SA[r]], FVC[r] = S[SA[r]+LCP[r]]
//                                                   (src/esa.c:420-422, 229-245)
//   K2a min_tree      64-ary min pyramid over LCP
//   K2b child_table   CLD from nearest-smaller-value searches  (src/esa.c:312-363)
//   K4  kmer_table    4^10 interval table, one thread per 10-mer (src/esa.c:73-215)
// and the scan index (see "scan index" below), built from RS and SA alone:
//   pack_symbols (scan_lane.hip)  the text as 4-bit symbols in two alignments (N0, N1)
//   probe_table                   4^K outcome table: a block takes 512 suffix-array gaps, makes the suffixes'
//                                 records itself (one 8-byte gather each) and writes the table piece it owns
// (the suffix array itself comes from the host or from sa_device.hip)
#include "andi_dev.h"
#include "esa_build.h"
#include "scan.h"
#include "knobs.h"

#include <cstring>

#define CHECK_LAUNCH()                                                                             \
	do {                                                                                           \
		hipError_t e_ = hipGetLastError();                                                         \
		if (e_ != hipSuccess) return e_;                                                           \
	} while (0)

// ---------------------------------------------------------------- K1a
__global__ __launch_bounds__(256) void k_phi_scatter(const int32_t *__restrict__ SA,
													 int32_t *__restrict__ phi, int32_t n) {
	int32_t r = blockIdx.x * blockDim.x + threadIdx.x;
	if (r >= n) return;
	int32_t prev = r ? SA[r - 1] : -1;
	phi[SA[r]] = prev;
}

// ---------------------------------------------------------------- K1b
// The reference walks the text once, carrying h = max(PLCP[t-1]-1, 0) as a
// lower bound (src/esa.c:402-417).  The bound holds from any starting point,
// so the text is cut into chunks of PLCP_CHUNK positions; each thread restarts
// with h = 0 and pays one full comparison (8 bytes per step) at its chunk head.
#define PLCP_CHUNK 32
__global__ __launch_bounds__(256) void k_plcp_chunks(const uint8_t *__restrict__ S,
													 int32_t *__restrict__ phi_plcp, int32_t n) {
	int64_t chunk = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	int64_t t0 = chunk * PLCP_CHUNK;
	if (t0 >= n) return;
	int32_t t1 = (int32_t)((t0 + PLCP_CHUNK < n) ? t0 + PLCP_CHUNK : n);
	uint32_t h = 0;
	for (int32_t t = (int32_t)t0; t < t1; ++t) {
		int32_t prev = phi_plcp[t];
		if (prev < 0) { // the lexicographically smallest suffix has no predecessor
			phi_plcp[t] = -1;
			continue;
		}
		// two distinct suffixes differ at or before the NUL at S[n]
		g_u8p a = (g_u8p)S + prev + h, b = (g_u8p)S + t + h;
		for (;;) {
			uint64_t x = ld_u64_unaligned(a) ^ ld_u64_unaligned(b);
			if (x) {
				h += (uint32_t)(__builtin_ctzll(x) >> 3);
				break;
			}
			h += 8, a += 8, b += 8;
		}
		phi_plcp[t] = (int32_t)h;
		h = h ? h - 1 : 0;
	}
}

// ---------------------------------------------------------------- K1c (+K3)
__global__ __launch_bounds__(256) void k_lcp_fvc(const uint8_t *__restrict__ S,
												 const int32_t *__restrict__ SA,
												 const int32_t *__restrict__ plcp,
												 int32_t *__restrict__ LCP, uint8_t *__restrict__ FVC,
												 int32_t n) {
	int32_t r = blockIdx.x * blockDim.x + threadIdx.x;
	if (r > n) return;
	if (r == n) {
		LCP[n] = -1;
		return;
	}
	int32_t sa = SA[r];
	int32_t l = r ? plcp[sa] : -1;
	LCP[r] = l;
	FVC[r] = S[sa + l]; // r = 0 reads S[SA[0]-1] like the reference; never consulted
}

// ---------------------------------------------------------------- K2a
// out[b] = min(in[64b .. 64b+63]) over the entries that exist.
__global__ __launch_bounds__(256) void k_min64(const int32_t *__restrict__ in, int32_t count,
											   int32_t *__restrict__ out) {
	int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	int32_t v = gid < count ? in[gid] : INT32_MAX;
	for (int off = 32; off; off >>= 1) {
		int32_t o = __shfl_xor(v, off);
		v = o < v ? o : v;
	}
	if ((threadIdx.x & 63) == 0 && gid < count) out[gid >> 6] = v;
}

// nearest p < q with lv0[p] <= v.  lv0[0] = -1 guarantees termination.
__device__ __forceinline__ int32_t nearest_left_le(const MinTree &t, int32_t q, int32_t v) {
	int32_t cur = q - 1;
	int lvl = 0;
	for (;;) {
		bool found = false;
		for (;;) {
			if (t.lv[lvl][cur] <= v) {
				found = true;
				break;
			}
			if ((cur & 63) == 0) break;
			--cur;
		}
		if (found) break;
		cur = (cur >> 6) - 1; // previous 64-group, one level up
		++lvl;
	}
	while (lvl > 0) {
		--lvl;
		cur = (cur << 6) + 63;
		while (t.lv[lvl][cur] > v) --cur;
	}
	return cur;
}

// nearest b > q with lv0[b] < v (v >= 0).  lv0[n] = -1 guarantees termination.
__device__ __forceinline__ int32_t nearest_right_lt(const MinTree &t, int32_t q, int32_t v) {
	int32_t cur = q + 1;
	int lvl = 0;
	for (;;) {
		bool found = false;
		for (;;) {
			if (t.lv[lvl][cur] < v) {
				found = true;
				break;
			}
			if ((cur & 63) == 63) break;
			++cur;
		}
		if (found) break;
		cur = (cur >> 6) + 1;
		++lvl;
	}
	while (lvl > 0) {
		--lvl;
		cur = cur << 6;
		while (t.lv[lvl][cur] >= v) ++cur;
	}
	return cur;
}

// ---------------------------------------------------------------- K2b
// The reference fills CLD with one stack sweep (src/esa.c:336-359).  Every
// slot it writes has a closed form: for q in 1..n let v = LCP[q],
//   a = nearest index left of q with LCP <= v,
//   b = nearest index right of q with LCP <  v.
//   LCP[a] == v        -> CLD[a]   = q   (next l-index of a)
//   LCP[a] <= LCP[b]   -> CLD[b-1] = q   ("up" of b: q is the leftmost minimum
//                                          of the block left of b)
//   otherwise          -> CLD[a]   = q   ("down" of a)
// q = n gives CLD[0] = n (src/esa.c:329).  Slots nobody writes stay -1
// (uninitialised in the reference and never read).
__global__ __launch_bounds__(256) void k_child_table(MinTree t, int32_t *__restrict__ CLD,
													 int32_t n) {
	int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (gid >= n) return;
	int32_t q = (int32_t)gid + 1;
	const int32_t *LCP = t.lv[0];
	int32_t v = LCP[q];
	int32_t a = nearest_left_le(t, q, v);
	int32_t la = LCP[a];
	if (la == v) {
		CLD[a] = q;
		return;
	}
	int32_t b = nearest_right_lt(t, q, v);
	if (la <= LCP[b]) {
		CLD[b - 1] = q;
	} else {
		CLD[a] = q;
	}
}

// ---------------------------------------------------------------- K4
// The reference fills the table by a depth-first walk over ACGT prefixes
// (esa_init_cache_dfs, src/esa.c:103-191).  The value a given 10-mer ends up
// with depends only on the decisions taken along its own characters, so each
// thread replays that walk for one code: absent child -> parent interval;
// singleton -> depth fixed to pos+1; interval deeper than one character but
// shallower than 10 -> parent for every 10-mer except the single existing
// elongation, which is followed (and cut at a separator).
__global__ __launch_bounds__(256) void k_kmer_table(EsaDev Ed, int4 *__restrict__ tab,
													int32_t *__restrict__ flags) {
	uint32_t code = blockIdx.x * blockDim.x + threadIdx.x;
	if (code >= (1u << (2 * ANDI_CACHE_K))) return;
	const EsaG E = esa_global(Ed);
	auto sym = [&](int pos) { return code_nt(code >> (2 * (ANDI_CACHE_K - 1 - pos))); };

	Ival in = esa_root(E);
	int pos = 0;
	Ival out;
	for (;;) {
		if (pos >= ANDI_CACHE_K) {
			out = in;
			break;
		}
		Ival ij = esa_child(E, in, sym(pos));
		if (ival_empty(ij)) {
			out = in;
			break;
		}
		if (ij.i == ij.j) {
			ij.l = pos + 1;
			out = ij;
			break;
		}
		if (ij.l <= pos + 1) {
			in = ij;
			++pos;
			continue;
		}
		if (ij.l >= ANDI_CACHE_K) {
			out = in;
			break;
		}
		// one elongation of length ij.l exists below this prefix
		g_u8p suf = E.S + E.SA[ij.i];
		int k = pos + 1;
		bool decided = false;
		for (; k < ij.l; ++k) {
			uint8_t e = suf[k];
			if (!is_acgt(e)) { // separator inside the interval's label
				out = ij;
				decided = true;
				// such an entry makes get_match_cached differ from the true
				// longest match
				flags[2] = 1; // (pinned host memory: plain idempotent store)
				break;
			}
			if (e != sym(k)) {
				out = in;
				decided = true;
				break;
			}
		}
		if (decided) break;
		in = ij;
		pos = ij.l;
	}
	tab[code] = make_int4(out.l, out.i, out.j, out.m);
}

// ================================================================ scan index
// The anchor scan needs, per probe, only (match length, unique?, position).
// The probe table answers that from the first K query characters.  It is built
// bottom-up from the suffix array in two streaming passes; the only random
// accesses are one 8-byte read of the packed text per suffix.
//
// A suffix's record (made on the fly by suffix_rec, kept in LDS per block) describes suffix SA[r]: bits 31..6 the 2-bit code of its first K
// characters (first character most significant, garbage past the valid part),
// bits 5..4 what follows the valid part (0 ACGT/none, 1 '!', 2 ';', 3 other),
// bits 3..0 v = number of leading ACGT characters, capped at K.
#define REC_V(x) ((x)&15u)
#define REC_SEP(x) (((x) >> 4) & 3u)
#define REC_CODE(x) ((x) >> 6)

// rec of suffix r: one 8-byte read of the text's 4-bit symbols (N0, see andi_dev.h) at the
// suffix's position -- the only random access of the build -- holds its first 15 symbols:
// K-mer code, number of leading nucleotides and the separator behind them all come from
// that word.
__device__ __forceinline__ uint32_t suffix_rec(const uint8_t *__restrict__ N0, const int32_t *__restrict__ SA,
											   int32_t r, int K) {
	const uint32_t p = (uint32_t)SA[r];
	const uint64_t w = ld_u64_unaligned((g_u8p)N0 + (p >> 1)) >> (4 * (p & 1u)); // symbol i at bits 4i..4i+3
	const uint64_t stop = (w & 0x4444444444444444ull) | (1ull << 62); // not a nucleotide (bit 2); symbol 15 is outside
	uint32_t v = (uint32_t)__builtin_ctzll(stop) >> 2;                // leading ACGT characters, <= 15
	uint32_t sep = 0;
	if (v < (uint32_t)K) {
		const uint32_t c = (uint32_t)(w >> (4 * v)) & 7u; // 4 '!', 5 ';', 6 '#', 7 NUL
		sep = c == 4 ? 1u : (c == 5 ? 2u : 3u);
	} else {
		v = (uint32_t)K;
	}
	auto squeeze = [](uint32_t x) { // 8 nibbles -> 8 x 2 bits, first symbol in the low bits
		x &= 0x33333333u;
		x = (x | (x >> 2)) & 0x0f0f0f0fu;
		x = (x | (x >> 4)) & 0x00ff00ffu;
		x = (x | (x >> 8)) & 0x0000ffffu;
		return x;
	};
	uint32_t y = __brev(squeeze((uint32_t)w) | (squeeze((uint32_t)(w >> 32)) << 16));
	y = ((y & 0x55555555u) << 1) | ((y >> 1) & 0x55555555u); // first symbol in the top two bits
	return ((y >> (32 - 2 * K)) << 6) | (sep << 4) | v;
}

// leading characters two suffixes share, counting ACGT only, capped at K
__device__ __forceinline__ uint32_t rec_lcp(uint32_t a, uint32_t b, int K) {
	uint32_t x = REC_CODE(a) ^ REC_CODE(b);
	uint32_t same = x ? (uint32_t)(__builtin_clz(x) - (32 - 2 * K)) >> 1 : (uint32_t)K;
	uint32_t va = REC_V(a), vb = REC_V(b);
	uint32_t m = va < vb ? va : vb;
	return same < m ? same : m;
}

// leading characters the K-mer `code` shares with a suffix
__device__ __forceinline__ uint32_t rec_lcp_code(uint32_t code, uint32_t a, int K) {
	uint32_t x = code ^ REC_CODE(a);
	uint32_t same = x ? (uint32_t)(__builtin_clz(x) - (32 - 2 * K)) >> 1 : (uint32_t)K;
	uint32_t va = REC_V(a);
	return same < va ? same : va;
}

// One thread per gap r = 0..n of the suffix array (between suffix r-1 and r) works out
// what the gap owns:
//  (a) if suffix r starts a run of suffixes with the same valid K-mer, that K-mer's
//      entry: SINGLE (position) or MULTI (first SA index and run length);
//  (b) every K-mer that sorts strictly inside the gap is absent from RS: its longest
//      match is the longer of its common prefixes with the two neighbours,
//      FINAL(l, unique, SA index);
//  (c) it detects what can make the reference's 10-mer table differ from the true
//      longest match: a prefix w (1..8 ACGT characters) whose every occurrence is
//      followed by the same separator, at least twice (SURVEY.md appendix C.11;
//      this test is a superset of the exact condition) -> flags[0].
// The codes owned by consecutive gaps are consecutive ranges (absent codes of gap r,
// then the K-mer of suffix r), so a block of 256 gaps owns one contiguous piece of the
// table.  The block writes that piece together, entry t by thread t mod 256 (coalesced,
// no divergent per-gap loops); a binary search over the gaps' offsets tells an entry
// which gap it belongs to.
#ifndef PT_BLOCK
#define PT_BLOCK 256
#endif
#ifndef PT_GAPS
#define PT_GAPS 3 /* gaps per thread (independent loads in flight): bench set 2.91 / 2.33 / 2.24 / 2.54 ms at 1 / 2 / 3 / 4 (round 6; LDS 20.6 KB per block at 3) */
#endif
#define PT_TILE (PT_BLOCK * PT_GAPS)
#define PT_WORDS (PT_TILE / 64) /* 64-bit ballots that cover the tile: gap i = u * PT_BLOCK + thread is bit i & 63 of word i >> 6 */
#define PT_RANKED 4096          /* entries of a block whose owners are found by rank (the block's piece is seldom longer: 768 gaps; a table a symbol deeper than the text asks for -- subjects that meet a thousand queries -- has four entries per gap) */
// inclusive prefix sums over the 64 lanes of a wavefront in six DPP additions: shifts by 1, 2, 4, 8 inside the rows
// of 16 lanes, then lane 15 of a row into the next row, lane 31 into the upper half
__device__ __forceinline__ uint32_t wave_scan_incl(uint32_t v) {
	v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false); // row_shr:1
	v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false); // row_shr:2
	v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false); // row_shr:4
	v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false); // row_shr:8
	v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false); // row_bcast:15 into rows 1 and 3
	v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false); // row_bcast:31 into rows 2 and 3
	return v;
}

// the first code the gap behind the suffix with record L can own (has: there is such a suffix): behind a full K-mer
// the next code; "w <sep>" sorts before every K-mer that starts with w
__device__ __forceinline__ uint32_t first_code(bool has, uint32_t L, int K) {
	const uint32_t v = REC_V(L), sh = 2 * ((uint32_t)K - v);
	const uint32_t c = ((REC_CODE(L) >> sh) << sh) + (v == (uint32_t)K ? 1u : 0u);
	return has ? c : 0u;
}

__device__ __forceinline__ void probe_table_block(const uint8_t *__restrict__ N0, const int32_t *__restrict__ SA,
												  const uint32_t *__restrict__ REC, const uint16_t *__restrict__ REC2, uint2 *__restrict__ deep,
												  int32_t *__restrict__ flags, int32_t n, int K, int single_ext, uint32_t block) {
	__shared__ uint32_t s_first[PT_TILE];   // first code a gap owns (one that owns nothing: its successor's)
	__shared__ uint32_t s_absent[PT_TILE];  // number of absent codes it owns (they come first)
	__shared__ uint32_t s_h[PT_TILE + 2];   // s_h[k + 1]: characters the suffixes r0 + k - 1 and r0 + k share
	__shared__ uint2 s_present[PT_TILE];    // entry of the K-mer of suffix r, if the gap owns it
	__shared__ uint32_t s_rec[PT_TILE + 5]; // rec of the suffixes r0 - 2 .. r0 + PT_TILE (two more words: the run test below reads ahead of what it uses)
	__shared__ uint64_t s_some[PT_WORDS];   // gap i owns entries
	__shared__ uint32_t s_bits[PT_RANKED / 32]; // bit t: an owning gap's entries start at t
	__shared__ uint32_t s_before[PT_RANKED / 32]; // set bits in the words before this one
	__shared__ uint16_t s_owner[PT_TILE];   // the owning gaps, in order

	const uint32_t r0 = block * PT_TILE; // (n < 2^31: gaps and suffixes in 32 bits)
	const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
	// the suffixes' records: read in order if the device sorter left them (REC), else made here from one gather each
	auto make_rec = [&](int32_t j) { return REC ? REC[j] : suffix_rec(N0, SA, j, K); };
	uint32_t mine2[PT_GAPS]; // the short extended form: what the sorter's keys held behind this thread's suffixes' K-mers (read with the records)
	{
		uint32_t mine[PT_GAPS];
#pragma unroll
		for (int u = 0; u < PT_GAPS; ++u) {
			const uint32_t g = r0 + threadIdx.x + u * PT_BLOCK;
			mine[u] = g < (uint32_t)n ? make_rec((int32_t)g) : 0u;
			mine2[u] = (single_ext == 2 && REC2 && g < (uint32_t)n) ? REC2[g] : 0u;
		}
#pragma unroll
		for (int u = 0; u < PT_GAPS; ++u) s_rec[threadIdx.x + u * PT_BLOCK + 2] = mine[u];
	}
	if (threadIdx.x < 2) s_rec[threadIdx.x] = r0 + threadIdx.x >= 2 ? make_rec((int32_t)(r0 + threadIdx.x - 2)) : 0u;
	if (threadIdx.x == 2) s_rec[PT_TILE + 2] = r0 + PT_TILE < (uint32_t)n ? make_rec((int32_t)(r0 + PT_TILE)) : 0u;
	if (threadIdx.x < PT_RANKED / 32) s_bits[threadIdx.x] = 0;
	__syncthreads();
	auto rec = [&](int32_t j) { // 0 <= j < n; inside the block's range from LDS
		const uint32_t k = (uint32_t)j - r0 + 2; // (j >= r0 - 2 wraps to a small number; anything else to a large one)
		return k < PT_TILE + 3 ? s_rec[k] : make_rec(j);
	};
	const uint32_t full = (uint32_t)K;
	if (threadIdx.x == 0) s_h[0] = r0 >= 2 ? rec_lcp(s_rec[0], s_rec[1], K) : 0u;
	if (threadIdx.x == 1) s_h[PT_TILE + 1] = r0 + PT_TILE < (uint32_t)n ? rec_lcp(s_rec[PT_TILE + 1], s_rec[PT_TILE + 2], K) : 0u;

	uint32_t counts[PT_GAPS], firsts[PT_GAPS];
#pragma unroll
	for (int u = 0; u < PT_GAPS; ++u) {
		const uint32_t i = threadIdx.x + u * PT_BLOCK; // gap r0 + i
		const uint32_t gid = r0 + i;
		const bool live = gid <= (uint32_t)n;
		const int32_t r = (int32_t)(live ? gid : 0);
		const bool hasL = live && r > 0, hasR = live && r < n;
		// (the records around the gap and the three behind it, and the suffix's position: fetched whatever the gap turns out to be --
		// a branch around every one of these loads cost more than the loads)
		const uint32_t Lr = s_rec[i + 1], Rr = s_rec[i + 2], a1 = s_rec[i + 3], a2 = s_rec[i + 4], a3 = s_rec[i + 5];
		const uint32_t L = hasL ? Lr : 0u, R = hasR ? Rr : 0u;
		const uint32_t sa_r = hasR ? (uint32_t)SA[r] : 0u;
		const uint32_t h = (hasL && hasR) ? rec_lcp(L, R, K) : 0u;
		s_h[i + 1] = h;
		uint32_t absent = 0, first = 1u << (2 * K), owns_present = 0; // (gaps beyond the text: behind every code)
		uint2 present = make_uint2(0, 0);

		if (live) {
			// (a) present K-mers
			if (hasR && REC_V(R) == full && !(hasL && L == R)) {
				// the end of the run of suffixes with this K-mer: inside the tile the first gap behind r that is not
				// inside a run; a run that leaves the tile is followed by binary search over the suffix array --
				// records of one K-mer are equal, later ones greater -- so that a K-mer with 10^6 occurrences (a
				// homopolymer, a satellite) does not serialise on one lane
				// the end of the run of suffixes with this K-mer: runs are short in genomes, so the next three records
				// are looked at together; a longer run is walked a little further and then followed by binary search
				// over the suffix array -- records of one K-mer are equal, later ones greater -- so that a K-mer with
				// 10^6 occurrences (a homopolymer, a satellite) does not serialise on one lane
				const bool in_tile = i + 5 < PT_TILE + 3; // (records of suffixes beyond the text are 0: never equal to R)
				const int32_t m = a1 != R ? 0 : (a2 != R ? 1 : (a3 != R ? 2 : 3));
				int32_t j = in_tile ? r + m : r;
				const bool more = !in_tile || m == 3;
				if (more) {
					while (j + 1 < n && j - r < 16 && rec(j + 1) == R) ++j;
					if (j - r == 16 && j + 1 < n && rec(j + 1) == R) {
						int32_t lo = j + 1, hi = n - 1; // rec(lo) == R; find the last index with rec == R
						while (lo < hi) {
							const int32_t mid = lo + ((hi - lo + 1) >> 1);
							if (rec(mid) == R) lo = mid; else hi = mid - 1;
						}
						j = lo;
					}
				}
				if (j == r) {
					// the K-mer occurs once.  For the scan in rounds (scan_rounds.hip) its entry also carries the
					// (up to 13) nucleotides that follow it in the text, so that a chance match is settled
					// without touching the text (costs the build a second gather: +11 %)
					const uint32_t pos = sa_r, e0 = pos + full;
					if (single_ext == 2 && REC2) { // the short extended form: what the sorter's keys held behind the K-mer
						const uint32_t w = mine2[u];
						present = make_uint2(pos, DEEP_SINGLE | ((w >> 8) << 2) | ((w & 0xffu) << 6));
					} else if (single_ext != 1) {
						present = make_uint2(pos, DEEP_SINGLE | (1u << 2) | (full << 8));
					} else {
						const uint64_t w = ld_u64_unaligned((g_u8p)N0 + (e0 >> 1)) >> (4 * (e0 & 1u)); // 15 symbols from e0 on
						uint32_t nval = (uint32_t)__builtin_ctzll((w & 0x4444444444444444ull) | (1ull << 52)) >> 2; // <= 13
						auto squeeze = [](uint32_t x) { // 8 nibbles -> 8 x 2 bits, first symbol in the low bits
							x &= 0x33333333u;
							x = (x | (x >> 2)) & 0x0f0f0f0fu;
							x = (x | (x >> 4)) & 0x00ff00ffu;
							x = (x | (x >> 8)) & 0x0000ffffu;
							return x;
						};
						const uint32_t ext = (squeeze((uint32_t)w) | (squeeze((uint32_t)(w >> 32)) << 16)) & ((1u << (2 * nval)) - 1u);
						present = make_uint2(pos, DEEP_SINGLE | (nval << 2) | (ext << 6));
					}
				} else {
					const bool fits = (uint32_t)(j - r) < (1u << 24);
					present = make_uint2(fits ? (uint32_t)r : 0u, fits ? DEEP_MULTI | ((uint32_t)(j - r) << 8) : (uint32_t)DEEP_SEARCH);
				}
				owns_present = 1;
			}

			// (c) closed run of suffixes "w <sep>" with 1 <= |w| <= 8 (a wavefront none of whose suffixes is followed by a '!' or ';' skips it)
			if (__builtin_amdgcn_ballot_w64(hasR && (REC_SEP(R) == 1 || REC_SEP(R) == 2)) != 0 && hasR) {
				uint32_t k = REC_V(R), sp = REC_SEP(R);
				if (k >= 1 && k <= 8 && k < full && (sp == 1 || sp == 2) && (!hasL || h < k)) {
					int32_t j = r;
					while (j + 1 < n) {
						uint32_t X = rec(j + 1);
						if (REC_V(X) == k && REC_SEP(X) == sp && rec_lcp(R, X, K) == k) ++j; else break;
					}
					if (j > r && (j + 1 == n || rec_lcp(R, rec(j + 1), K) < k)) flags[0] = 1; // (pinned host memory: plain idempotent store)
				}
			}

			// (b) absent K-mers inside this gap
			const int32_t lo = (int32_t)first_code(hasL, L, K); // codes are below 4^13: 32 bits do
			const uint32_t shR = 2 * (full - REC_V(R)); // (0 for a full K-mer: the code itself; selects, not branches)
			const int32_t hi = hasR ? (int32_t)((REC_CODE(R) >> shR) << shR) - 1 : (int32_t)((1u << (2 * K)) - 1u);
			if (lo <= hi) absent = (uint32_t)(hi - lo + 1);
			first = (uint32_t)lo; // (a gap that owns only its suffix's K-mer: lo is that K-mer; one that owns nothing: where the next one starts)
		}
		s_first[i] = first, s_absent[i] = absent, s_present[i] = present;
		counts[u] = absent + owns_present, firsts[u] = first;
		const uint64_t b = __ballot(counts[u] != 0);
		if (lane == 0) s_some[u * (PT_BLOCK / 64) + wave] = b;
	}

	// The codes the gaps own are consecutive ranges (see above), so a gap's entries start at its first code: the block's
	// piece of the table is [first code of its first gap, first code of the next block's first gap), entry t of the
	// piece is code base + t, and no prefix sums over the gaps' counts are needed.
	__syncthreads();
	const uint32_t base = first_code(r0 > 0, s_rec[1], K);
	const uint32_t total = (r0 + PT_TILE <= (uint32_t)n ? first_code(true, s_rec[PT_TILE + 1], K) : (1u << (2 * K))) - base;
	// the owners of the first PT_RANKED entries by rank: a bit where an owning gap's entries start (starts are
	// distinct), the owning gaps listed in order; the owner of entry t is then number (set bits up to t) of the list
	uint32_t owners_before; // owning gaps in the ballot words before word `lane` (lane < PT_WORDS)
	{
		const uint32_t mine = lane < PT_WORDS ? (uint32_t)__builtin_popcountll(s_some[lane]) : 0u;
		owners_before = wave_scan_incl(mine) - mine;
	}
#pragma unroll
	for (int u = 0; u < PT_GAPS; ++u) {
		const uint32_t word = u * (PT_BLOCK / 64) + wave;
		const uint32_t before = (uint32_t)__builtin_amdgcn_readlane((int)owners_before, __builtin_amdgcn_readfirstlane((int)word)); // (word is wave-uniform: v_readlane, not a trip through the LDS crossbar)
		if (counts[u] == 0) continue;
		const uint32_t i = threadIdx.x + u * PT_BLOCK, off = firsts[u] - base;
		s_owner[before + (uint32_t)__builtin_popcountll(s_some[word] & ((1ull << lane) - 1ull))] = (uint16_t)i;
		if (off < PT_RANKED) atomicOr(&s_bits[off >> 5], 1u << (off & 31u));
	}
	__syncthreads();
	if (wave == 0) { // set bits before each word of s_bits (64 words per pass of one wavefront)
		uint32_t carry = 0;
		for (uint32_t w0 = 0; w0 < PT_RANKED / 32; w0 += 64) {
			const uint32_t mine = (uint32_t)__builtin_popcount(s_bits[w0 + lane]), incl = wave_scan_incl(mine);
			s_before[w0 + lane] = carry + incl - mine;
			carry += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
		}
	}
	__syncthreads();

	// what entry t of the piece holds, a its owner (ONE store per entry, whatever it is: 256 consecutive entries by one instruction -- two stores
	// under complementary masks wrote every line of the table in two pieces)
	auto write_entry = [&](uint32_t t, uint32_t a) {
		const uint32_t c = base + t;
		const bool is_present = c - s_first[a] >= s_absent[a]; // the K-mer of suffix r itself
		const uint2 pres = s_present[a];
		const uint32_t rr = r0 + a;
		const bool gL = rr > 0, gR = rr < (uint32_t)n;
		// is the left (right) neighbour the only suffix sharing a given prefix length with it?
		// lcp(suffix rr-2, rr-1) and lcp(suffix rr, rr+1) are the neighbouring gaps' h (0 outside the text)
		const uint32_t gl = s_rec[a + 1], gr = s_rec[a + 2], hll = s_h[a], hrr = s_h[a + 2];
		const uint32_t xL = rec_lcp_code(c, gl, K), xR = rec_lcp_code(c, gr, K); // (both computed, then selected: no branch around seven instructions)
		const uint32_t lL = gL ? xL : 0u, lR = gR ? xR : 0u;
		// (selects: both neighbours sharing l characters -- or l == 0: every suffix does -- is the third case)
		const bool left = lL > lR, right = lR > lL;
		const uint32_t l = left ? lL : lR;
		const uint32_t idx = left ? rr - 1 : (right ? rr : 0u);
		const bool alone = left ? (rr < 2 || hll < l) : (rr + 1 >= (uint32_t)n || hrr < l);
		const uint32_t uniq = ((left || right) && l != 0) ? (alone ? 1u : 0u) : (n == 1 ? 1u : 0u);
		const uint2 val = is_present ? pres : make_uint2(idx, DEEP_FINAL | (uniq << 2) | (l << 8));
		__builtin_nontemporal_store(((unsigned long long)val.y << 32) | val.x, (unsigned long long *)(deep + c));
	};
	// (two loops: the ranked entries -- all of them, as a rule -- in one without the binary search's branch and inner loop)
	const uint32_t ranked = total < PT_RANKED ? total : PT_RANKED;
	for (uint32_t t = threadIdx.x; t < ranked; t += PT_BLOCK) {
		const uint32_t word = t >> 5;
		write_entry(t, s_owner[s_before[word] + (uint32_t)__builtin_popcount(s_bits[word] & (0xffffffffu >> (31u - (t & 31u)))) - 1u]);
	}
	for (uint32_t t = PT_RANKED + threadIdx.x; t < total; t += PT_BLOCK) {
		// the gap that owns code c: the last one whose first code is <= c (gaps that own nothing share their successor's first code)
		const uint32_t c = base + t;
		uint32_t a = 0, b = PT_TILE; // invariant: s_first[a] <= c < s_first[b]
		while (b - a > 1) {
			const uint32_t mid = (a + b) >> 1;
			if (s_first[mid] <= c) a = mid; else b = mid;
		}
		write_entry(t, a);
	}
}

__global__ __launch_bounds__(PT_BLOCK) void k_probe_table(const uint8_t *__restrict__ N0, const int32_t *__restrict__ SA,
														  const uint32_t *__restrict__ rec, const uint16_t *__restrict__ rec2, uint2 *__restrict__ deep,
														  int32_t *__restrict__ flags, int32_t n, int K, int single_ext) {
	probe_table_block(N0, SA, rec, rec2, deep, flags, n, K, single_ext, blockIdx.x);
}

// the tables of several subjects in one launch (blockIdx.y = subject): no launch gaps, one tail
__global__ __launch_bounds__(PT_BLOCK) void k_probe_table_batch(const AndiIndexBatchItem *__restrict__ items) {
	const AndiIndexBatchItem it = items[blockIdx.y];
	if ((uint64_t)blockIdx.x * PT_TILE > (uint64_t)it.n) return;
	probe_table_block(it.N0, it.SA, it.rec, it.rec2, it.deep, it.flags, it.n, it.deepK, it.single_ext, blockIdx.x);
}

// ---------------------------------------------------------------- host side
size_t andi_min_tree_entries(int32_t n) {
	size_t total = 0;
	size_t cnt = (size_t)n + 1;
	while (cnt > 64) {
		cnt = (cnt + 63) / 64;
		total += cnt;
	}
	return total ? total : 1;
}

// The entries of K-mers that occur once also carry the (up to 13) nucleotides behind the occurrence when a scan is going to
// read them: pass A in rounds (scan_rounds.hip) and pass A with one wavefront per chain (scan_coop.hip).  Every scan
// understands both forms (the position is in the same place); the subject's handle remembers which it has.
int andi_index_single_ext(size_t queries, bool sorted_on_device) { // queries: how many the subject is going to meet (0: unknown)
	const int coop = andi_coop_enabled();
	if (coop == 0) return 0;
	if (const char *f = andi_knob(KNOB_SINGLE_EXT)) { // (experiments: 0, 1, 2; anything else is ignored -- the scan decodes an entry by the form its handle records)
		const int v = atoi(f);
		if (v >= 0 && v <= 2 && (v != 2 || sorted_on_device)) return v;
	}
	// Pass A by wavefronts reads them: a chance occurrence of a K-mer off the window's diagonal is then settled by the
	// entry instead of a look at the text with the parked lanes (bench set: pass A 4.95 -> 4.71 ms).  Four symbols settle
	// all but one chance occurrence in 256 -- pass A is as fast as with thirteen (bench set 4.71 / 4.67 ms, C4 shape
	// 28.7 / 28.7) -- and come with the device sorter's records; a subject whose suffix array came from the host gets the
	// long form (a gather from the text: + 20 % of the build) where it pays: the scan forced to that kernel, or
	// hundreds of queries.
	if (sorted_on_device) return 2;
	return (coop > 0 || queries >= 256) ? 1 : 0;
}

hipError_t andi_launch_index_build(const EsaBuildArgs &a, int single_ext, hipStream_t st) {
	const int32_t n = a.n;
	hipError_t e;
	// symbols for the lane scan: the text, its NUL and 64 bytes of the zero padding behind it
	e = andi_launch_pack_symbols(a.S, (size_t)n + 1 + 64, a.N0, a.N1, a.flags + 1, st);
	if (e != hipSuccess) return e;
	// ... and bit-sliced, for the wavefront kernels' streams (round 6: once per subject here, not in front of every launch of theirs)
	if (a.P) e = andi_launch_pack_planes(a.N0, (size_t)n + 1 + 64, a.P, st);
	if (e != hipSuccess) return e;
	k_probe_table<<<(unsigned)(((int64_t)n + 1 + PT_TILE - 1) / PT_TILE), PT_BLOCK, 0, st>>>(a.N0, a.SA, a.rec, a.rec2, a.deep, a.flags, n,
																				  a.deepK, single_ext);
	CHECK_LAUNCH();
	return hipSuccess;
}

hipError_t andi_launch_index_build_batch(const AndiIndexBatchItem *d_items, uint32_t count, int32_t max_n, hipStream_t st) {
	if (count == 0) return hipSuccess;
	hipError_t e = andi_launch_pack_symbols_batch(d_items, count, (size_t)max_n + 1 + 64, st);
	if (e != hipSuccess) return e;
	e = andi_launch_pack_planes_batch(d_items, count, (size_t)max_n, st);
	if (e != hipSuccess) return e;
	const dim3 grid((unsigned)(((int64_t)max_n + 1 + PT_TILE - 1) / PT_TILE), count);
	k_probe_table_batch<<<grid, PT_BLOCK, 0, st>>>(d_items);
	CHECK_LAUNCH();
	return hipSuccess;
}

hipError_t andi_launch_esa_build(const EsaBuildArgs &a, hipStream_t st) {
	const int32_t n = a.n;
	const int B = 256;
	auto blocks = [&](int64_t items) { return (unsigned)((items + B - 1) / B); };

	// K1: LCP via PHI/PLCP; the PLCP scratch is the CLD buffer's first n ints
	int32_t *plcp = a.CLD;
	k_phi_scatter<<<blocks(n), B, 0, st>>>(a.SA, plcp, n);
	CHECK_LAUNCH();
	int64_t chunks = ((int64_t)n + PLCP_CHUNK - 1) / PLCP_CHUNK;
	k_plcp_chunks<<<blocks(chunks), B, 0, st>>>(a.S, plcp, n);
	CHECK_LAUNCH();
	k_lcp_fvc<<<blocks((int64_t)n + 1), B, 0, st>>>(a.S, a.SA, plcp, a.LCP, a.FVC, n);
	CHECK_LAUNCH();

	// K2: min pyramid, then the child table
	MinTree t;
	t.lv[0] = a.LCP;
	t.cnt[0] = n + 1;
	t.levels = 1;
	int32_t *next = a.min_scratch;
	while (t.cnt[t.levels - 1] > 64 && t.levels < ANDI_MIN_LEVELS) {
		int32_t in_cnt = t.cnt[t.levels - 1];
		int32_t out_cnt = (in_cnt + 63) / 64;
		k_min64<<<blocks(in_cnt), B, 0, st>>>(t.lv[t.levels - 1], in_cnt, next);
		CHECK_LAUNCH();
		t.lv[t.levels] = next;
		t.cnt[t.levels] = out_cnt;
		next += out_cnt;
		t.levels++;
	}
	hipError_t e = hipMemsetAsync(a.CLD, 0xff, ((size_t)n + 1) * sizeof(int32_t), st);
	if (e != hipSuccess) return e;
	k_child_table<<<blocks(n), B, 0, st>>>(t, a.CLD, n);
	CHECK_LAUNCH();

	// K4: 10-mer interval table
	EsaDev E;
	memset(&E, 0, sizeof E);
	E.S = a.S, E.SA = a.SA, E.LCP = a.LCP, E.CLD = a.CLD, E.FVC = a.FVC, E.tab = a.tab;
	E.flags = a.flags;
	E.n = n, E.mode = ANDI_MODE_REFERENCE;
	k_kmer_table<<<blocks(1 << (2 * ANDI_CACHE_K)), B, 0, st>>>(E, a.tab, a.flags);
	CHECK_LAUNCH();
	return hipSuccess;
}

// ------------------------------------------------------------------ subjects from sequences that are already in HBM
// seq_subject_init (src/sequence.c:210-219) for a sequence that lies in the query pool: RS = revcomp(S) '#' S
// (catcomp, src/sequence.c:177-190; revcomp, 143-168: A<->T, C<->G, the contig separator '!' becomes ';') written
// straight into the subject's slot -- what andi_hip_subject_prepare (host_seq.c) makes on the host and the seam used to
// upload a second time.  One thread per aligned word of RS; the bytes behind the text are the slot's zero padding.
__global__ __launch_bounds__(256) void k_rs_from_query(uint8_t *__restrict__ RS, const uint8_t *__restrict__ q, uint32_t len) {
	const uint64_t w = (uint64_t)blockIdx.x * 256 + threadIdx.x, o = 4 * w, n = 2 * (uint64_t)len + 1;
	if (o >= n) return;
	uint32_t word = 0;
#pragma unroll
	for (uint32_t k = 0; k < 4; ++k) {
		const uint64_t p = o + k;
		uint32_t c = 0;
		if (p < len) {
			c = q[len - 1 - p];
			c = c < 'A' ? (uint32_t)';' : (c ^ ((c & 2u) ? 4u : 21u));
		} else if (p == len) {
			c = '#';
		} else if (p < n) {
			c = q[p - len - 1];
		}
		word |= c << (8 * k);
	}
	((uint32_t *)RS)[w] = word;
}

// calc_gc's numerator (src/sequence.c:197-208) for every sequence of the pool: counts[q] = number of G and C.  Blocks of 64 KiB
// of one sequence; 16 bytes per thread and step, one atomic per wavefront.
#define GC_CHUNK 65536u
__global__ __launch_bounds__(256) void k_gc_counts(const uint8_t *__restrict__ pool, const uint64_t *__restrict__ off, const uint32_t *__restrict__ len,
													unsigned long long *__restrict__ counts) {
	const uint32_t qi = blockIdx.y, L = len[qi];
	const uint64_t c0 = (uint64_t)blockIdx.x * GC_CHUNK;
	if (c0 >= L) return;
	const uint8_t *s = pool + off[qi]; // (256-byte aligned; the pool is zero behind every sequence)
	const uint64_t c1 = c0 + GC_CHUNK < L ? c0 + GC_CHUNK : L;
	uint32_t cnt = 0;
	for (uint64_t p = c0 + 16ull * threadIdx.x; p < c1; p += 16ull * 256) {
		const uint4 v = *(const uint4 *)(s + p);
		const uint32_t ws[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
		for (int j = 0; j < 4; ++j)
#pragma unroll
			for (int b = 0; b < 4; ++b) {
				const uint32_t c = (ws[j] >> (8 * b)) & 0xffu;
				cnt += (p + 4 * j + b < c1) && ((c & 0xfbu) == 0x43u); // 'C' 0x43, 'G' 0x47
			}
	}
	for (int o = 32; o > 0; o >>= 1) cnt += __shfl_down(cnt, o);
	if ((threadIdx.x & 63u) == 0 && cnt) atomicAdd(&counts[qi], (unsigned long long)cnt);
}

// contig separators ('!', joined contigs: src/sequence.c:78-125) of every sequence of the pool: every one is a place where the diagonal
// of a pair breaks -- the routing of the scan wants to know how often (scan_lane.hip: k_pair_estimate)
__global__ __launch_bounds__(256) void k_sep_counts(const uint8_t *__restrict__ pool, const uint64_t *__restrict__ off, const uint32_t *__restrict__ len,
													 uint32_t *__restrict__ counts) {
	const uint32_t qi = blockIdx.y, L = len[qi];
	const uint64_t c0 = (uint64_t)blockIdx.x * GC_CHUNK;
	if (c0 >= L) return;
	const uint8_t *s = pool + off[qi];
	const uint64_t c1 = c0 + GC_CHUNK < L ? c0 + GC_CHUNK : L;
	uint32_t cnt = 0;
	for (uint64_t p = c0 + 16ull * threadIdx.x; p < c1; p += 16ull * 256) {
		const uint4 v = *(const uint4 *)(s + p);
		const uint32_t ws[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
		for (int j = 0; j < 4; ++j) {
			// bytes below 'A' (0x41) inside the sequence are separators; a zero-byte test on (w & 0xc0c0c0c0) would count NULs too, but
			// the sequence holds none before its end: positions at and beyond c1 are masked
#pragma unroll
			for (int b = 0; b < 4; ++b) cnt += (p + 4 * j + b < c1) && (((ws[j] >> (8 * b)) & 0xffu) == (uint32_t)'!');
		}
	}
	for (int o = 32; o > 0; o >>= 1) cnt += __shfl_down(cnt, o);
	if ((threadIdx.x & 63u) == 0 && cnt) atomicAdd(&counts[qi], cnt);
}

hipError_t andi_launch_sep_counts(const uint8_t *pool, const uint64_t *d_off, const uint32_t *d_len, uint32_t nq, uint32_t longest,
								  uint32_t *d_counts, hipStream_t st) {
	if (!nq || !longest) return hipSuccess;
	for (uint32_t q0 = 0; q0 < nq; q0 += 65535u) {
		const uint32_t cnt = nq - q0 < 65535u ? nq - q0 : 65535u;
		k_sep_counts<<<dim3((longest + GC_CHUNK - 1) / GC_CHUNK, cnt), 256, 0, st>>>(pool, d_off + q0, d_len + q0, d_counts + q0);
		CHECK_LAUNCH();
	}
	return hipSuccess;
}

hipError_t andi_launch_rs_from_query(uint8_t *RS, const uint8_t *q, uint32_t len, hipStream_t st) {
	const uint64_t words = (2 * (uint64_t)len + 1 + 3) / 4;
	k_rs_from_query<<<(unsigned)((words + 255) / 256), 256, 0, st>>>(RS, q, len);
	CHECK_LAUNCH();
	return hipSuccess;
}

hipError_t andi_launch_gc_counts(const uint8_t *pool, const uint64_t *d_off, const uint32_t *d_len, uint32_t nq, uint32_t longest,
								 unsigned long long *d_counts, hipStream_t st) {
	if (!nq || !longest) return hipSuccess;
	for (uint32_t q0 = 0; q0 < nq; q0 += 65535u) { // (grid.y is 16 bits)
		const uint32_t cnt = nq - q0 < 65535u ? nq - q0 : 65535u;
		k_gc_counts<<<dim3((longest + GC_CHUNK - 1) / GC_CHUNK, cnt), 256, 0, st>>>(pool, d_off + q0, d_len + q0, d_counts + q0);
		CHECK_LAUNCH();
	}
	return hipSuccess;
}
