// andi_dev.h — device-side view of one subject's index and the primitives shared
// by the index-build and scan kernels (gfx950).
//
// Layout in HBM per subject (all hipMalloc'ed, see api.hip):
//   S    uint8[n+1+PAD]  RS, NUL at n, zero padding so wide loads never fault
//   SA   int32[n]        suffix array (host-built, src/esa.c:294-304)
// scan index (what the anchor scan uses; built by k_pack_symbols + k_probe_table):
//   deep uint2[4^K]      probe table: for every ACGT K-mer the outcome of the
//                        longest-match search as far as the K-mer alone decides
//                        it, so most probes cost one random access
//   N0   uint8[n/2+..]   the text as 4-bit symbols (A C G T ! ; # NUL = 0..7), symbol i in
//                        the low (i even) or high half of byte i/2; N1 the same shifted by
//                        one symbol (byte b = symbols 2b-1, 2b), so that a 16-byte load can
//                        start at any symbol.  256 readable bytes in front, 384 behind.
//   flags int32[4]       [0] != 0: some 10-mer table entry may span a separator
//                        (SURVEY.md appendix C.11): the reference's cached lookup
//                        is then not the true longest match and the scan follows
//                        the reference walk below instead of the probe table
//                        [1] != 0: the text holds a byte outside {A C G T ! ; # NUL};
//                        the packed-symbol scan is then not applicable
//                        [2] set by the 10-mer table kernel when it really
//                        produced such an entry (diagnostic)
// reference arrays (esa_s, src/esa.h:42-59; built on request or when flags[0]):
//   LCP  int32[n+1]      LCP[0]=LCP[n]=-1          (K1, src/esa.c:373-426)
//   CLD  int32[n+1]      child table                (K2, src/esa.c:312-363)
//   FVC  uint8[n]        S[SA[i]+LCP[i]]            (K3, src/esa.c:229-245)
//   tab  int4[4^10]      10-mer interval table {l,i,j,m} (K4, src/esa.c:73-215)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define ANDI_CACHE_K 10
#define ANDI_PAD 2048 /* bytes of zero padding behind every byte pool */
#define ANDI_MAX_DEEP_K 13
#define ANDI_NIB_BACK 384  /* readable bytes behind the packed symbols of a text */
#define ANDI_NIB_FRONT 256 /* and in front of them */

struct EsaDev {
	const uint8_t *S;
	const int32_t *SA;
	const int32_t *LCP; // reference arrays: null until built
	const int32_t *CLD;
	const uint8_t *FVC;
	const int4 *tab; // x=l y=i z=j w=m
	const uint2 *deep;
	const int32_t *flags;
	const uint8_t *N0, *N1; // nibble-packed text, two alignments (scan_lane.hip)
	const uint16_t *R2;     // per suffix, in suffix-array order: the (up to min(4, 16 - K)) nucleotides behind its first K symbols: count << 8 | 2-bit codes, first in the low bits (the device sorter's by-product; null: not there)
	const uint32_t *P;      // the text bit-sliced (coop_pool.h): block b = words 3b, 3b+1, 3b+2 = bit 0, 1, 2 of the symbols 32b ... 32b+31; a block of padding in front
	int32_t n;
	int32_t thr;
	int32_t deepK;
	int32_t mode; // ANDI_MODE_*
	int32_t deep_ext; // the entries of K-mers that occur once carry the nucleotides behind the occurrence (DEEP_SINGLE): 1 up to 13 of them, 2 up to min(4, 16 - K)
};

#define ANDI_MODE_PROBE 0     /* probe table + suffix-array search (true longest match) */
#define ANDI_MODE_REFERENCE 1 /* 10-mer table + child-table walk, exactly as src/esa.c */

// probe-table entry: x = payload, y = kind | unique << 2 | l << 8
#define DEEP_FINAL 0u  /* K-mer absent: match length l < K; x = SA index of the one suffix if unique */
#define DEEP_SINGLE 1u /* K-mer occurs once: x = its position in RS; in the extended form (EsaDev.deep_ext) y >> 2 & 15 = v <= 13 nucleotides follow it, y >> 6 = their 2-bit codes (first in the low bits) */
#define DEEP_MULTI 2u  /* K-mer occurs more than once: x = first SA index, y >> 8 = run length - 1 */
#define DEEP_SEARCH 3u /* (run too long to encode) search the whole suffix array */

// Device code addresses the index through global-address-space pointers so
// the compiler emits global_load (not flat_load) for them.
#define ANDI_GLOBAL __attribute__((address_space(1)))
typedef ANDI_GLOBAL const uint8_t *g_u8p;
typedef ANDI_GLOBAL const int32_t *g_i32p;
typedef ANDI_GLOBAL const int4 *g_i4p;
typedef ANDI_GLOBAL const uint2 *g_u2p;
typedef ANDI_GLOBAL const uint32_t *g_u32p;
typedef ANDI_GLOBAL const uint16_t *g_u16p;

struct EsaG {
	g_u8p S;
	g_i32p SA, LCP, CLD;
	g_u8p FVC;
	g_i4p tab;
	g_u2p deep;
	g_u8p N0, N1;
	g_u32p P;
	g_u16p R2;
	int32_t n, thr, deepK, mode, deep_ext;
};

__device__ __forceinline__ EsaG esa_global(const EsaDev &e) {
	EsaG g;
	g.S = (g_u8p)e.S, g.SA = (g_i32p)e.SA, g.LCP = (g_i32p)e.LCP, g.CLD = (g_i32p)e.CLD;
	g.FVC = (g_u8p)e.FVC, g.tab = (g_i4p)e.tab;
	g.deep = (g_u2p)e.deep;
	g.N0 = (g_u8p)e.N0, g.N1 = (g_u8p)e.N1;
	g.P = (g_u32p)e.P;
	g.R2 = (g_u16p)e.R2;
	g.n = e.n, g.thr = e.thr, g.deepK = e.deepK, g.mode = e.mode, g.deep_ext = e.deep_ext;
	return g;
}

struct Ival { // lcp_inter_t, src/esa.h:25-34
	int32_t l, i, j, m;
};

__device__ __forceinline__ bool ival_empty(const Ival &v) {
	return v.i == -1 && v.j == -1;
}

__device__ __forceinline__ uint32_t ld_u32_unaligned(g_u8p p) {
	uint32_t v;
	__builtin_memcpy(&v, p, 4);
	return v;
}

__device__ __forceinline__ uint64_t ld_u64_unaligned(g_u8p p) {
	uint64_t v;
	__builtin_memcpy(&v, p, 8);
	return v;
}

// true for A C G T; everything else in the alphabet {\0 ! # ;} is below 'A'
__device__ __forceinline__ bool is_acgt(uint8_t c) {
	return c >= 'A';
}

// char2code, src/esa.c:49-58 == nucl2bit, src/model.c:295-299: A0 C1 G2 T3
__device__ __forceinline__ uint32_t nt_code(uint8_t c) {
	uint32_t x = c & 6u;
	return (x ^ (x >> 1)) >> 1;
}

__device__ __forceinline__ uint8_t code_nt(uint32_t code) {
	return (uint8_t)((0x54474341u >> (8 * (code & 3u))) & 0xffu); // "ACGT"
}

__device__ __forceinline__ uint4 ld_u128_unaligned(g_u8p p) {
	uint4 v;
	__builtin_memcpy(&v, p, 16);
	return v;
}

// Lanes are organised in groups of G consecutive lanes (G = 1, 2, 4 ... 64).
// All lanes of a group hold the same chain state and execute the same control
// flow; different groups of one wavefront may diverge.
template <int G>
struct Group {
	static_assert(G >= 1 && G <= 64 && (G & (G - 1)) == 0, "group size must be a power of two");
	static __device__ __forceinline__ uint32_t sub() { return __lane_id() & (G - 1); }
	static __device__ __forceinline__ uint32_t base() { return __lane_id() & ~(uint32_t)(G - 1); }
	// the group's slice of a wave-wide ballot
	static __device__ __forceinline__ uint64_t slice(uint64_t ballot) {
		if constexpr (G == 64) return ballot;
		return (ballot >> base()) & ((1ull << G) - 1);
	}
};

// byte index of the first non-zero byte of the 16-byte value x, 16 if none
__device__ __forceinline__ uint32_t first_diff_byte(uint4 x) {
	if (x.x) return (uint32_t)__builtin_ctz(x.x) >> 3;
	if (x.y) return 4 + ((uint32_t)__builtin_ctz(x.y) >> 3);
	if (x.z) return 8 + ((uint32_t)__builtin_ctz(x.z) >> 3);
	if (x.w) return 12 + ((uint32_t)__builtin_ctz(x.w) >> 3);
	return 16;
}

// Length of the common prefix of a[0..maxlen) and b[0..maxlen).
// G == 1: one thread, 8 bytes per step.  G > 1: the G lanes of a group call with
// identical arguments and share the work, 16 bytes per lane and step (one
// 16*G-byte window of each string); the result is uniform within the group.
// Both strings must be readable ANDI_PAD bytes past maxlen.
template <int G>
__device__ __forceinline__ uint32_t common_prefix(g_u8p a, g_u8p b,
												  uint32_t maxlen) {
	if constexpr (G == 1) {
		uint32_t k = 0;
		while (k < maxlen) {
			uint64_t x = ld_u64_unaligned(a + k) ^ ld_u64_unaligned(b + k);
			if (x) {
				k += (uint32_t)(__builtin_ctzll(x) >> 3);
				return k < maxlen ? k : maxlen;
			}
			k += 8;
		}
		return maxlen;
	} else {
		const uint32_t sub = Group<G>::sub();
		for (uint32_t done = 0; done < maxlen; done += 16 * G) {
			uint32_t off = done + 16 * sub;
			uint4 wa = ld_u128_unaligned(a + off), wb = ld_u128_unaligned(b + off);
			uint4 x = make_uint4(wa.x ^ wb.x, wa.y ^ wb.y, wa.z ^ wb.z, wa.w ^ wb.w);
			uint32_t f = first_diff_byte(x);
			uint64_t diff = Group<G>::slice(__ballot(f < 16));
			if (diff) {
				uint32_t first = (uint32_t)__builtin_ctzll(diff);
				uint32_t ff = (uint32_t)__shfl((int)f, (int)(Group<G>::base() + first));
				uint32_t pos = done + 16 * first + ff;
				return pos < maxlen ? pos : maxlen;
			}
		}
		return maxlen;
	}
}

__device__ __forceinline__ Ival esa_root(const EsaG &E) {
	Ival r;
	r.i = 0;
	r.j = E.n - 1;
	r.m = E.CLD[E.n - 1]; // L(CLD, n), src/esa.c:82-83
	r.l = E.LCP[r.m];
	return r;
}

// get_interval, src/esa.c:441-511: the child of `ij` whose next character is a.
__device__ __forceinline__ Ival esa_child(const EsaG &E, Ival ij, uint8_t a) {
	int32_t i = ij.i;
	const int32_t j = ij.j;
	if (i == j) {
		if (E.S[E.SA[i] + ij.l] != a) ij.i = ij.j = -1;
		return ij;
	}
	int32_t m = ij.m;
	const int32_t l = ij.l;
	uint8_t c = E.S[E.SA[i] + l];
	for (;;) {
		if (c == a) {
			Ival r;
			if (i != m - 1) {
				int32_t nm = E.CLD[m - 1];
				r.i = i, r.j = m - 1, r.m = nm, r.l = E.LCP[nm];
			} else {
				r.i = i, r.j = i, r.m = -1, r.l = E.LCP[i];
			}
			return r;
		}
		if (c > a) break;
		i = m;
		if (i == j) break;
		m = E.CLD[m];
		if (E.LCP[m] != l) break;
		c = E.FVC[i];
	}
	bool hit = (i != ij.i) ? (E.FVC[i] == a) : (E.S[E.SA[i] + l] == a);
	if (hit) {
		ij.i = i;
		ij.l = E.LCP[m];
		ij.m = m;
	} else {
		ij.i = ij.j = -1;
	}
	return ij;
}

// get_match_from, src/esa.c:531-601
template <int G>
__device__ __forceinline__ Ival esa_match_from(const EsaG &E, g_u8p q, uint32_t qlen,
											   int32_t k, Ival ij) {
	if (ival_empty(ij)) return ij;
	if (ij.i == ij.j) {
		// singleton: plain extension along the one suffix.  RS's NUL can never
		// equal a query byte, so the `S[p+k]` stop of the reference is implied.
		uint32_t from = (uint32_t)ij.l;
		if (from < qlen)
			from += common_prefix<G>(q + from, E.S + E.SA[ij.i] + from, qlen - from);
		ij.l = (int32_t)from;
		return ij;
	}
	Ival res = ij;
	do {
		ij = esa_child(E, ij, q[k]);
		if (ival_empty(ij)) {
			res.l = k;
			return res;
		}
		res.i = ij.i;
		res.j = ij.j;
		int32_t lim = (int32_t)qlen;
		if (ij.i < ij.j && ij.l < lim) lim = ij.l;
		++k;
		if (k < lim) {
			k += (int32_t)common_prefix<G>(q + k, E.S + E.SA[ij.i] + k, (uint32_t)(lim - k));
			if (k < lim) {
				res.l = k;
				return res;
			}
		}
	} while (k < (int32_t)qlen);
	res.l = (int32_t)qlen;
	return res;
}

// get_match, src/esa.c:615-624
template <int G>
__device__ __forceinline__ Ival esa_match(const EsaG &E, g_u8p q, uint32_t qlen) {
	return esa_match_from<G>(E, q, qlen, 0, esa_root(E));
}

// 2-bit code of the first 10 characters, first character most significant
// (src/esa.c:639-643); returns false if any of them is not ACGT.
__device__ __forceinline__ bool kmer10_code(g_u8p q, uint32_t &code) {
	uint32_t w0 = ld_u32_unaligned(q), w1 = ld_u32_unaligned(q + 4), w2 = ld_u32_unaligned(q + 8);
	w2 &= 0x0000ffffu;
	// every ACGT byte has bit 6 set, no separator has
	bool ok = ((w0 & 0x40404040u) == 0x40404040u) && ((w1 & 0x40404040u) == 0x40404040u) &&
			  ((w2 & 0x00004040u) == 0x00004040u);
	auto pack4 = [](uint32_t w) {
		uint32_t x = w & 0x06060606u;
		x ^= x >> 1;
		x = (x >> 1) & 0x03030303u; // 2-bit code per byte, byte 0 = first char
		return ((x & 0xffu) << 6) | (((x >> 8) & 0xffu) << 4) | (((x >> 16) & 0xffu) << 2) |
			   (x >> 24);
	};
	code = (pack4(w0) << 12) | (pack4(w1) << 4) | (pack4(w2) >> 4);
	return ok;
}

// get_match_cached, src/esa.c:636-656
template <int G>
__device__ __forceinline__ Ival esa_match_cached(const EsaG &E, g_u8p q, uint32_t qlen) {
	if (qlen <= ANDI_CACHE_K) return esa_match<G>(E, q, qlen);
	uint32_t code;
	if (!kmer10_code(q, code)) return esa_match<G>(E, q, qlen);
	uint4 t = ld_u128_unaligned((g_u8p)(E.tab + code)); // {l,i,j,m}
	Ival ij;
	ij.l = (int32_t)t.x, ij.i = (int32_t)t.y, ij.j = (int32_t)t.z, ij.m = (int32_t)t.w;
	if (ival_empty(ij)) return esa_match<G>(E, q, qlen);
	return esa_match_from<G>(E, q, qlen, ij.l, ij);
}

// 2-bit code of the first K (<= 16) characters of q, first character most
// significant; false if one of them is not ACGT.  Reads 16 bytes.
__device__ __forceinline__ bool kmer_code_from(uint4 w, int K, uint32_t &code) {
	auto pack4 = [](uint32_t v) {
		uint32_t x = v & 0x06060606u;
		x ^= x >> 1;
		x = (x >> 1) & 0x03030303u;
		return ((x & 0xffu) << 6) | (((x >> 8) & 0xffu) << 4) | (((x >> 16) & 0xffu) << 2) | (x >> 24);
	};
	auto need = [K](int d) { // bit-6 mask of the characters of dword d that belong to the K-mer
		int c = K - 4 * d;
		c = c < 0 ? 0 : (c > 4 ? 4 : c);
		return c == 4 ? 0x40404040u : (0x40404040u & ((1u << (8 * c)) - 1u));
	};
	bool ok = ((w.x & need(0)) == need(0)) && ((w.y & need(1)) == need(1)) &&
			  ((w.z & need(2)) == need(2)) && ((w.w & need(3)) == need(3));
	uint32_t c32 = (pack4(w.x) << 24) | (pack4(w.y) << 16) | (pack4(w.z) << 8) | pack4(w.w);
	code = c32 >> (32 - 2 * K);
	return ok;
}

__device__ __forceinline__ bool kmer_code(g_u8p q, int K, uint32_t &code) {
	return kmer_code_from(ld_u128_unaligned(q), K, code);
}

// What dist_anchor needs from get_match_cached (src/process.c:113-123): the
// match length, whether the match is unique (inter.i == inter.j) and SA[inter.i].
struct Probe {
	uint32_t len;
	uint32_t pos;
	bool unique;
};

// Longest match of q[0..qlen) given that exactly the suffixes SA[lo..hi] start
// with q[0..k): the search of get_match_from (src/esa.c:531-601) without the
// child table.  All suffixes of the range share c = lcp(first, last) characters;
// q is compared against that label, then the range is narrowed by q[c] with two
// binary searches (suffixes are sorted by their character at depth c).
template <int G>
__device__ __forceinline__ Probe sa_range_match(const EsaG &E, g_u8p q, uint32_t qlen, int32_t lo,
												int32_t hi, uint32_t k) {
	Probe r;
	for (;;) {
		int32_t pl = E.SA[lo];
		if (lo == hi) { // one suffix left: extend along it
			r.len = k < qlen ? k + common_prefix<G>(q + k, E.S + pl + k, qlen - k) : k;
			r.pos = (uint32_t)pl, r.unique = true;
			return r;
		}
		r.pos = (uint32_t)pl, r.unique = false;
		if (k >= qlen) {
			r.len = qlen;
			return r;
		}
		int32_t ph = E.SA[hi];
		// two distinct suffixes differ at or before the shorter one's NUL
		uint32_t room = (uint32_t)E.n - (uint32_t)(pl > ph ? pl : ph);
		uint32_t c = k < room ? k + common_prefix<G>(E.S + pl + k, E.S + ph + k, room - k) : k;
		uint32_t lim = c < qlen ? c : qlen;
		uint32_t m = k < lim ? k + common_prefix<G>(q + k, E.S + pl + k, lim - k) : k;
		if (m < lim || m >= qlen) { // mismatch inside the shared label, or query exhausted
			r.len = m;
			return r;
		}
		// m == c < qlen: narrow by the next query character
		const uint8_t ch = q[c];
		int32_t a = lo, b = hi + 1; // first index with char >= ch
		while (a < b) {
			int32_t mid = a + ((b - a) >> 1);
			if (E.S[E.SA[mid] + c] < ch) a = mid + 1; else b = mid;
		}
		int32_t first = a;
		b = hi + 1; // first index with char > ch
		while (a < b) {
			int32_t mid = a + ((b - a) >> 1);
			if (E.S[E.SA[mid] + c] <= ch) a = mid + 1; else b = mid;
		}
		if (first == a) { // no suffix continues with ch
			r.len = c;
			return r;
		}
		lo = first, hi = a - 1, k = c + 1;
	}
}

// The probe of one chain step in ANDI_MODE_REFERENCE: the reference's own walk
// (get_match_cached, src/esa.c:636-656) reduced to what dist_anchor reads of it.
// (ANDI_MODE_PROBE lives in scan.hip / scan_lane.hip: probe_step, lane_probe.)
template <int G>
__device__ __forceinline__ Probe esa_probe(const EsaG &E, g_u8p q, uint32_t qlen) {
	Probe r;
	Ival m = esa_match_cached<G>(E, q, qlen);
	r.len = m.l <= 0 ? 0u : (uint32_t)m.l;
	r.unique = m.i == m.j;
	r.pos = (uint32_t)E.SA[m.i];
	return r;
}
