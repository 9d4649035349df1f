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
//   K4b probe_table   4^K outcome table for K = 11..13, one thread per K-mer
#include "andi_dev.h"
#include "esa_build.h"

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
				// longest match: remember it
				atomicOr(&flags[0], 1);
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

// ---------------------------------------------------------------- K4b
// Probe table.  For every ACGT K-mer w: how far does the longest-match search
// (get_match, src/esa.c:531-624) get on w alone?
//   w absent      -> FINAL: the match length l < K, uniqueness and SA[i] of the
//                    matched prefix are the final answer for every query that
//                    starts with w;
//   w occurs once -> SINGLE: its position; the query is extended along it;
//   w occurs more -> MULTI: the lcp-interval of w, the search resumes there.
// Semantics are those of the true longest match; they coincide with the
// reference's get_match_cached unless flags[0] is set, in which case the scan
// ignores this table.  The walk starts from the 10-mer table entry.
__global__ __launch_bounds__(256) void k_probe_table(EsaDev Ed, uint2 *__restrict__ deep,
													 int4 *__restrict__ side,
													 int32_t *__restrict__ flags) {
	const int K = Ed.deepK;
	uint32_t code = blockIdx.x * blockDim.x + threadIdx.x;
	if (code >= (1u << (2 * K))) return;
	if (flags[0]) return; // 10-mer table is not the true longest match here
	const EsaG E = esa_global(Ed);
	auto sym = [&](int pos) { return code_nt(code >> (2 * (K - 1 - pos))); };
	auto emit = [&](uint32_t kind, uint32_t unique, uint32_t l, uint32_t x) {
		deep[code] = make_uint2(x, kind | (unique << 2) | (l << 8));
	};

	uint4 t = ld_u128_unaligned((g_u8p)(E.tab + (code >> (2 * (K - ANDI_CACHE_K)))));
	Ival in;
	in.l = (int32_t)t.x, in.i = (int32_t)t.y, in.j = (int32_t)t.z, in.m = (int32_t)t.w;
	int pos = in.l; // characters of w matched so far; == lcp of `in` unless singleton
	for (;;) {
		if (in.i == in.j) { // one suffix left: compare the rest of w against it
			int32_t suf = E.SA[in.i];
			while (pos < K && E.S[suf + pos] == sym(pos)) ++pos;
			if (pos == K) {
				emit(DEEP_SINGLE, 1, (uint32_t)K, (uint32_t)suf);
			} else {
				emit(DEEP_FINAL, 1, (uint32_t)pos, (uint32_t)suf);
			}
			return;
		}
		if (pos >= K) { // w is a proper prefix of (or equal to) the interval's label
			uint32_t slot = (uint32_t)atomicAdd(&flags[1], 1);
			if (slot < (uint32_t)Ed.side_cap) {
				side[slot] = make_int4(in.l, in.i, in.j, in.m);
				emit(DEEP_MULTI, 0, (uint32_t)K, slot);
			} else {
				emit(DEEP_FALLBACK, 0, 0, 0);
			}
			return;
		}
		// here pos == in.l: branch on the next character
		Ival ij = esa_child(E, in, sym(pos));
		if (ival_empty(ij)) {
			emit(DEEP_FINAL, 0, (uint32_t)pos, (uint32_t)E.SA[in.i]);
			return;
		}
		++pos;
		if (ij.i < ij.j) { // verify the label up to the child's depth
			int32_t suf = E.SA[ij.i];
			int lim = ij.l < K ? ij.l : K;
			while (pos < lim && E.S[suf + pos] == sym(pos)) ++pos;
			if (pos < lim) {
				emit(DEEP_FINAL, 0, (uint32_t)pos, (uint32_t)suf);
				return;
			}
		}
		in = ij;
	}
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
	E.S = a.S, E.SA = a.SA, E.LCP = a.LCP, E.CLD = a.CLD, E.FVC = a.FVC, E.tab = a.tab;
	E.deep = a.deep, E.side = a.side, E.flags = a.flags;
	E.n = n, E.thr = 0, E.deepK = a.deepK, E.side_cap = a.side_cap;
	e = hipMemsetAsync(a.flags, 0, 4 * sizeof(int32_t), st);
	if (e != hipSuccess) return e;
	k_kmer_table<<<blocks(1 << (2 * ANDI_CACHE_K)), B, 0, st>>>(E, a.tab, a.flags);
	CHECK_LAUNCH();

	// K4b: probe table
	if (a.deep && a.deepK > 0) {
		k_probe_table<<<blocks((int64_t)1 << (2 * a.deepK)), B, 0, st>>>(E, a.deep, a.side, a.flags);
		CHECK_LAUNCH();
	}
	return hipSuccess;
}
