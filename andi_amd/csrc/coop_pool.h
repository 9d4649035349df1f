// coop_pool.h -- pass A with one wavefront per chain, the window's walks POOLED through global memory (mode P).
// Included by scan_coop.hip inside its namespace; it shares that file's mode G and its helpers.
//
// coop_window (mode W) keeps a window of 8192 positions in LDS -- mismatch bits, the query's 2-bit codes, the heads'
// records -- and walks the window's heads (42 on average on the bench set) with 64 lanes: 16 lanes at work per trip
// (profiles/r06_coop/work_split_bench.txt).  LDS is what bounds the window (profiles/r05_ring/README.md).  Here the window
// is up to POOL_MW = 131072 positions long and lives in a per-wavefront scratch in global memory, and the work on it
// is three sweeps of the same wavefront, each depending on POSITIONS alone:
//
//   S  stream the window along the diagonal, 2048 positions per round: mismatch bits -> scratch; every mismatch is
//      counted as a single-position gap at once (speculatively: what the chain turns out not to reach is taken back
//      in sweep R); heads (a mismatch followed by < thr equal symbols and preceded by >= thr) are listed with a
//      RECORD of what their walk reads: the 64 mismatch bits and the 64 query symbols (2-bit codes) behind them.
//   W  the walks of ALL heads of the window, 64 per trip, a lane that is done takes the next head: the record is the
//      lane's only load besides the probe table's entry (walks that leave their record read the scratch / the query).
//   R  the chain hops from head to head (64 heads per round, ballots); the stretches behind the heads it came by are
//      or-ed into the window's bits (atomic or: two stretches may meet in a word; round 5 kept them in a second bitmap,
//      zeroed and read back per window: C4 shape 25.7 -> 23.6 ms without it); one pass over the bits counts the anchors
//      (src/model.c:247-253).
//
// The states and counts are those of the sequential loop (src/process.c:141-214), as coop_window's: tests/test_coop_gpu.py
// runs both against the oracle and against each other.  LogDet / ANI (EXACT) stay with coop_window.

#ifndef POOL_MW_POS
#define POOL_MW_POS 131072
#endif
constexpr uint32_t POOL_MW = POOL_MW_POS;       // positions of a window at most (a multiple of 2048)
constexpr uint32_t POOL_CHUNK_HEADS = 128;      // heads of one round of 2048 positions (more: as above)

struct __attribute__((aligned(16))) PoolRec { // what a head's walk reads
	uint32_t q2[4];   // the query's symbols at positions pos + 1 ... pos + 64, bit-sliced: [0], [1] = bit 0 of the 64 symbols, [2], [3] = bit 1
	uint32_t bits[2]; // the mismatch bits of those positions
	uint32_t pos, dirty; // dirty: one of those 64 query symbols is no nucleotide ('!' of joined contigs): the walk's probes from the record go the long way
};
struct __attribute__((aligned(16))) PoolRes { // what it found
	uint32_t pos, ha, hend, flag; // the head; landing position; length of the anchor landed on (W_LUCKY: not known); pool_pack_flag(W_* flags, equal symbols of the stretch)
};
// The flag word of a result: bits 0-4 the W_* status bits, 5-6 the anchors met off the diagonal (W_NX_SHIFT; at most COOP_MAX_X = 3),
// bit 7: bits 8-31 hold the equal symbols of the walk's stretch by nucleotide, six bits each -- an ordinary stretch of at most
// POOL_EQ_MAX positions inside a record without separators (the walk has the head's record in registers when it ends; until
// round 6 sweep R fetched every record a second time for them: 32 bytes per head)
constexpr uint32_t POOL_EQ_MAX = 63, POOL_EQ_VALID = 0x80u;
static_assert(COOP_MAX_X <= 3, "two bits of the packed flag");
__device__ __forceinline__ uint32_t pool_pack_flag(uint32_t res, uint32_t eq, bool eq_valid) {
	return (res & 0x1fu) | (((res >> W_NX_SHIFT) & 3u) << 5) | (eq_valid ? POOL_EQ_VALID : 0u) | (eq << 8);
}
__device__ __forceinline__ uint32_t pool_flag_of(uint32_t w) {
	return (w & 0x1fu) | (((w >> 5) & 3u) << W_NX_SHIFT);
}
struct PoolScratch { // a window's scratch in global memory, one per resident wavefront
	uint32_t *bits;  // [64 maxchunks + 64] bit (x - wbase): query symbol x != subject symbol x + dg
	uint32_t *ebits; // (-DPOOL_SEPARATE_EBITS, the A/B build of round 5's layout: [64 maxchunks + 64] sweep R's stretches in a bitmap of their own; by default they are or-ed into `bits` and this is null)
	PoolRec *rec;    // [hc]
	PoolRes *res;    // [hc]
	uint32_t hc;        // heads of a window that are walked (more: nothing is decided from the first one dropped on)
	uint32_t maxchunks; // rounds of 2048 positions of a window at most
};
__host__ __device__ inline size_t pool_scratch_bytes(uint32_t maxchunks, uint32_t hc) {
#ifdef POOL_SEPARATE_EBITS
	return 2 * (size_t)(64 * maxchunks + 64) * sizeof(uint32_t) + (size_t)hc * (sizeof(PoolRec) + sizeof(PoolRes));
#else
	return (size_t)(64 * maxchunks + 64) * sizeof(uint32_t) + (size_t)hc * (sizeof(PoolRec) + sizeof(PoolRes));
#endif
}
__device__ __forceinline__ PoolScratch pool_scratch_at(void *base, size_t idx, uint32_t maxchunks, uint32_t hc) {
	char *p = (char *)base + idx * pool_scratch_bytes(maxchunks, hc);
	PoolScratch g;
	g.bits = (uint32_t *)p, p += (size_t)(64 * maxchunks + 64) * sizeof(uint32_t);
#ifdef POOL_SEPARATE_EBITS
	g.ebits = (uint32_t *)p, p += (size_t)(64 * maxchunks + 64) * sizeof(uint32_t);
#else
	g.ebits = nullptr;
#endif
	g.rec = (PoolRec *)p, p += (size_t)hc * sizeof(PoolRec);
	g.res = (PoolRes *)p;
	g.hc = hc, g.maxchunks = maxchunks;
	return g;
}
struct PoolLds {
	uint32_t mring[256];          // sweep S: the bits of the last four rounds (word w at w & 255)
	uint32_t qring[512];          // and the query's 2-bit codes (two words per word of bits)
	uint32_t dring[256];          // ... and which of its symbols are no nucleotides (joined contigs' '!'): for the heads' records
	uint16_t hl[POOL_CHUNK_HEADS]; // the heads of a round, offsets into it
	uint32_t pl[64], pp[64]; // mode G: the block of probes
	uint32_t hist[16];
};

// the lanes of the wavefront hand data over through GLOBAL memory between the sweeps: the stores must have been written
// through before another lane's load is issued (loads and stores of a wavefront are not kept in order with each other)
__device__ __forceinline__ void pool_sync() {
	__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
	__builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ uint32_t pool_ld_bits(const uint32_t *p) {
	return *p;
}

// first set bit of the window's bits at or after position x (x >= wbase), NOPOS if none (all lanes)
__device__ __forceinline__ uint32_t pool_next_bit(const PoolScratch *G, uint32_t nwords, uint32_t wbase, uint32_t x) {
	const uint32_t lane = __lane_id(), o = x - wbase;
	for (uint32_t w0 = o >> 5; w0 < nwords; w0 += 64) {
		const uint32_t w = w0 + lane;
		uint32_t v = w < nwords ? G->bits[w] : 0u;
		if (w == (o >> 5)) v &= ~0u << (o & 31u);
		const uint64_t hit = __ballot(v != 0);
		if (hit) {
			const uint32_t l = (uint32_t)__builtin_ctzll(hit);
			const uint32_t vl = lane_read(v, l);
			return uni(wbase + 32 * (w0 + l) + (uint32_t)__builtin_ctz(vl));
		}
	}
	return NOPOS;
}

// last set bit before position x (x > wbase), NOPOS if none (all lanes; the bits as the device holds them now)
__device__ __forceinline__ uint32_t pool_prev_bit(const PoolScratch *G, uint32_t wbase, uint32_t x) {
	const uint32_t lane = __lane_id(), o = x - wbase;
	if (o == 0) return NOPOS;
	const uint32_t top = (o - 1) >> 5;
	for (int32_t w0 = (int32_t)top; w0 >= 0; w0 -= 64) {
		const int32_t w = w0 - (int32_t)lane;
		uint32_t v = w >= 0 ? pool_ld_bits(&G->bits[w]) : 0u;
		if (w == (int32_t)top && ((o - 1) & 31u) != 31u) v &= (2u << ((o - 1) & 31u)) - 1u;
		const uint64_t hit = __ballot(v != 0);
		if (hit) {
			const uint32_t l = (uint32_t)__builtin_ctzll(hit);
			const uint32_t vl = lane_read(v, l);
			return uni(wbase + 32 * (uint32_t)(w0 - (int32_t)l) + 31u - (uint32_t)__builtin_clz(vl));
		}
	}
	return NOPOS;
}

// the same for ONE lane (the anchor before a head whose walk met anchors off the diagonal: rare)
__device__ __forceinline__ uint32_t pool_prev_bit_lane(const PoolScratch *G, uint32_t wbase, uint32_t x) {
	uint32_t o = x - wbase;
	if (o == 0) return NOPOS;
	int32_t w = (int32_t)((o - 1) >> 5);
	uint32_t v = pool_ld_bits(&G->bits[w]);
	if (((o - 1) & 31u) != 31u) v &= (2u << ((o - 1) & 31u)) - 1u;
	while (v == 0 && w > 0) v = pool_ld_bits(&G->bits[--w]);
	return v ? wbase + 32 * (uint32_t)w + 31u - (uint32_t)__builtin_clz(v) : NOPOS;
}

// substitutions of Q[q..q+len) against S[s..s+len) taken back from the counts (all lanes, 32 positions each; what sweep S counted of
// a stretch that is counted nowhere)
__device__ __forceinline__ void pool_uncount_coop(const PairCtx &c, lds_u32 *hist, uint32_t q, uint32_t s, uint32_t len) {
	const uint32_t lane = __lane_id(), qe = q & ~1u, end = q + len;
	const int64_t dg = (int64_t)s - (int64_t)q;
	for (uint32_t base = qe; base < end; base += 64 * WNT) {
		const uint32_t x0 = base + WNT * lane;
		if (x0 >= end) continue;
		const uint4 qv = ld_query(c, x0), sv = ld_subject_guarded(c, (int64_t)x0 + dg);
		const uint32_t lo = q > x0 ? q - x0 : 0u, hi = end - x0 < WNT ? end - x0 : WNT;
		for (uint32_t j = lo >> 3; 8 * j < hi; ++j) {
			const uint32_t a = lo > 8 * j ? lo - 8 * j : 0u, b = hi - 8 * j < 8 ? hi - 8 * j : 8u;
			const uint32_t qw = pick(qv, j), sw = pick(sv, j), dw = neq8(qw, sw) >> 3;
			for (uint32_t ne = symbol_range(a, b) & ~(qw >> 2) & ~(sw >> 2) & dw; ne; ne &= ne - 1) {
				const uint32_t k = (uint32_t)__builtin_ctz(ne);
				lds_add(&hist[(((sw >> k) & 3u) << 2) | ((qw >> k) & 3u)], 0u - 1u);
			}
		}
	}
}

// the equal nucleotides of Q[q..q+len) against S[s..s+len) into the diagonal cells (all lanes, 32 positions each; a stretch longer
// than its head's record)
__device__ __forceinline__ void pool_count_equal_coop(const PairCtx &c, uint32_t q, uint32_t s, uint32_t len, uint32_t &n0, uint32_t &n1, uint32_t &n2, uint32_t &n3) {
	const uint32_t lane = __lane_id(), qe = q & ~1u, end = q + len;
	const int64_t dg = (int64_t)s - (int64_t)q;
	for (uint32_t base = qe; base < end; base += 64 * WNT) {
		const uint32_t x0 = base + WNT * lane;
		if (x0 >= end) continue;
		const uint4 qv = ld_query(c, x0), sv = ld_subject_guarded(c, (int64_t)x0 + dg);
		const uint32_t lo = q > x0 ? q - x0 : 0u, hi = end - x0 < WNT ? end - x0 : WNT;
#pragma unroll
		for (uint32_t j = 0; j < 4; ++j) {
			const uint32_t a = lo > 8 * j ? (lo - 8 * j < 8 ? lo - 8 * j : 8u) : 0u, b = hi > 8 * j ? (hi - 8 * j < 8 ? hi - 8 * j : 8u) : 0u;
			if (a >= b) continue;
			const uint32_t qw = pick(qv, j), sw = pick(sv, j), dw = neq8(qw, sw) >> 3;
			const uint32_t eq = symbol_range(a, b) & ~(qw >> 2) & ~(sw >> 2) & ~dw, b0 = qw, b1 = qw >> 1;
			n0 += (uint32_t)__builtin_popcount(eq & ~(b0 | b1)), n1 += (uint32_t)__builtin_popcount(eq & b0 & ~b1);
			n2 += (uint32_t)__builtin_popcount(eq & b1 & ~b0), n3 += (uint32_t)__builtin_popcount(eq & b0 & b1);
		}
	}
}

// The texts bit-sliced (andi_dev.h: EsaDev.P): 32 symbols of the query from x0 (a multiple of 32) / of the subject from any offset s
// (>= -32; at and beyond the text's end: NUL, all ones) as bit 0, 1, 2 of the symbols
struct Planes {
	uint32_t b0, b1, b2;
};
__device__ __forceinline__ Planes ld_query_planes(const PairCtx &c, uint32_t x0) {
	const g_u32p p = c.Qp + 3 * (x0 >> 5);
	Planes r;
	r.b0 = p[0], r.b1 = p[1], r.b2 = p[2];
	return r;
}
__device__ __forceinline__ Planes ld_subject_planes(const PairCtx &c, int64_t s) {
	Planes r;
	r.b0 = r.b1 = r.b2 = ~0u;
	if (s >= (int64_t)c.E.n) return r;
	const int32_t blk = (int32_t)(s >> 5);
	const uint32_t sh = (uint32_t)s & 31u;
	const g_u32p p = c.E.P + 3 * blk; // (blk >= -1: a block of padding lies in front)
	const uint32_t a0 = p[0], a1 = p[1], a2 = p[2], n0 = p[3], n1 = p[4], n2 = p[5];
	r.b0 = __builtin_amdgcn_alignbit(n0, a0, sh), r.b1 = __builtin_amdgcn_alignbit(n1, a1, sh), r.b2 = __builtin_amdgcn_alignbit(n2, a2, sh);
	return r;
}
// The same two fetches as the words come from memory -- no use of a loaded value, so that the loads of round t + 1 stay in flight
// while round t is worked on (ld_subject_planes shifts its words at once: the wait for them then stands right behind the loads,
// and sweep S paid the stream's whole latency every round; round 6) -- and what turns them into planes.
typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
typedef uint32_t u32x3_t __attribute__((ext_vector_type(3)));
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
struct RawPlanes { // (whole registers tuples as the loads deliver them: taken apart only where they are used -- a struct of nine words was
	u32x3_t q;     // copied word by word at the loop's head, with the wait for the loads in front of the copies)
	u32x4_t a;     // subject block: planes 0, 1, 2 and plane 0 of the next block
	u32x2_t n;     // planes 1, 2 of the next block
};
__device__ __forceinline__ void ld_raw_planes(const PairCtx &c, uint32_t x0, int64_t s, RawPlanes &r) {
	r.q = *(const __attribute__((address_space(1))) u32x3_t *)(c.Qp + 3 * (x0 >> 5));
	r.a = (u32x4_t)(~0u), r.n = (u32x2_t)(~0u); // (at and beyond the text's end: NUL, all ones)
	if (s < (int64_t)c.E.n) {
		const g_u32p p = c.E.P + 3 * (int32_t)(s >> 5); // (block >= -1: a block of padding lies in front)
		r.a = *(const __attribute__((address_space(1))) u32x4_t *)p;
		r.n = *(const __attribute__((address_space(1))) u32x2_t *)(p + 4);
	}
}
__device__ __forceinline__ void planes_of(const RawPlanes &r, uint32_t sh, Planes &q, Planes &s) {
	q.b0 = r.q.x, q.b1 = r.q.y, q.b2 = r.q.z;
	s.b0 = __builtin_amdgcn_alignbit(r.a.w, r.a.x, sh), s.b1 = __builtin_amdgcn_alignbit(r.n.x, r.a.y, sh), s.b2 = __builtin_amdgcn_alignbit(r.n.y, r.a.z, sh);
}
// the substitutions among the positions `mm` (query symbol != subject symbol, both nucleotides) by kind, into twelve counters:
// kind k = 3 * (query nucleotide) + (rank of the subject's among the three others) -- no loop over the mismatches, no LDS
struct SubstAcc {
	uint32_t n[12];
};
__device__ __forceinline__ void subst_count(SubstAcc &acc, uint32_t mm, const Planes &q, const Planes &s) {
	const uint32_t qA = mm & ~q.b1 & ~q.b0, qC = mm & ~q.b1 & q.b0, qG = mm & q.b1 & ~q.b0, qT = mm & q.b1 & q.b0;
	const uint32_t sA = ~s.b1 & ~s.b0, sC = ~s.b1 & s.b0, sG = s.b1 & ~s.b0, sT = s.b1 & s.b0;
	acc.n[0] += (uint32_t)__builtin_popcount(qA & sC), acc.n[1] += (uint32_t)__builtin_popcount(qA & sG), acc.n[2] += (uint32_t)__builtin_popcount(qA & sT);
	acc.n[3] += (uint32_t)__builtin_popcount(qC & sA), acc.n[4] += (uint32_t)__builtin_popcount(qC & sG), acc.n[5] += (uint32_t)__builtin_popcount(qC & sT);
	acc.n[6] += (uint32_t)__builtin_popcount(qG & sA), acc.n[7] += (uint32_t)__builtin_popcount(qG & sC), acc.n[8] += (uint32_t)__builtin_popcount(qG & sT);
	acc.n[9] += (uint32_t)__builtin_popcount(qT & sA), acc.n[10] += (uint32_t)__builtin_popcount(qT & sC), acc.n[11] += (uint32_t)__builtin_popcount(qT & sG);
}
// the counters of all lanes into the 4 x 4 counts (cell = subject nucleotide << 2 | query nucleotide, src/model.c:309-337), added or taken back
__device__ __forceinline__ void subst_flush(const SubstAcc &acc, lds_u32 *hist, bool add) {
	const uint32_t lane = __lane_id();
#pragma unroll
	for (int k = 0; k < 12; ++k) {
		const uint32_t qn = (uint32_t)k / 3, r = (uint32_t)k % 3, sn = r + (r >= qn ? 1u : 0u);
		const uint32_t v = wave_sum(acc.n[k]);
		if (lane == 0 && v) lds_add(&hist[(sn << 2) | qn], add ? v : 0u - v);
	}
}

// 16 bits -> the even bits of a word (bit k to bit 2k)
__device__ __forceinline__ uint32_t spread16(uint32_t x) {
	x &= 0xffffu;
	x = (x | (x << 8)) & 0x00ff00ffu;
	x = (x | (x << 4)) & 0x0f0f0f0fu;
	x = (x | (x << 2)) & 0x33333333u;
	return (x | (x << 1)) & 0x55555555u;
}

#ifdef POOL_KNOCK /* diagnostic builds: parts of the kernel switched off at run time (ANDI_KNOCK=bits) to count their instructions; results are then wrong */
#define PKNOCK(bit) ((a.knock >> (bit)) & 1u)
#else
#define PKNOCK(bit) false
#endif

// ------------------------------------------------------------------ mode P
// A window: what sweep S leaves for the sweeps W and R (wave-uniform)
struct PoolWin {
	uint32_t e0, nchunks, nheads, last_mm, f_cap, clean;
};

// Sweep S.  The chain stands at a canonical state of diagonal dg, as for coop_window; `chunks`: rounds of 2048 positions this
// window may take.
__device__ __forceinline__ void pool_stream(const ScanArgs &a, const PairCtx &c, Chain &ch, PoolLds &L, const PoolScratch *G, uint32_t end, uint32_t chunks, PoolWin &pw) {
	const uint32_t lane = __lane_id(), thr = c.thr;
	const ChainState &st = ch.st;
	const int64_t dg = (int64_t)st.lastS - (int64_t)st.lastQ;
	const uint32_t e0 = st.lastQ + st.lastLen;
	const uint32_t wbase = e0 & ~31u;
	lds_u32 *hist = (lds_u32 *)L.hist;
	// the window: up to the round behind the segment's end (an anchor that begins inside the segment may end there) and the query's
	uint32_t nchunks = (end + 2048u - wbase + 2047u) / 2048u;
	{
		const uint32_t qch = (c.qlen + 64u - wbase + 2047u) / 2048u;
		if (qch < nchunks) nchunks = qch;
		if (chunks < nchunks) nchunks = chunks;
		if (G->maxchunks < nchunks) nchunks = G->maxchunks;
	}
	const uint32_t nwords = 64 * nchunks, Wp = 2048 * nchunks, wend = wbase + Wp;
	(void)Wp;

	TICK(tph);
	// ---- sweep S
	uint32_t nheads = 0, f_cap = NOPOS, last_mm = e0, dirty = 0;
	uint32_t ring_dirty = 0xfu; // which of the ring's four rounds of L.dring may hold a separator's bit (a window begins with whatever the last one left)
	bool win_dirty = false;     // the window has shown a separator so far (wave-uniform)
	bool heads_on = true;
	{
		Planes qv, sv;
		qv.b0 = qv.b1 = qv.b2 = sv.b0 = sv.b1 = sv.b2 = 0;
		SubstAcc acc; // every mismatch a single-position gap (model_count of one position, src/model.c:309-337): sweep R takes back what is not
#pragma unroll
		for (int k = 0; k < 12; ++k) acc.n[k] = 0;
		// (a lane's subject offset x0 + dg keeps its low five bits from round to round: rounds are 2048 positions apart)
		const uint32_t sh = (uint32_t)((int64_t)(wbase + WNT * lane) + dg) & 31u;
		RawPlanes raw;
		raw.q = (u32x3_t)(0u), raw.a = (u32x4_t)(0u), raw.n = (u32x2_t)(0u);
		auto fetch = [&](uint32_t t) {
			const uint32_t x0 = wbase + 2048 * t + WNT * lane;
			if (t < nchunks && x0 < c.qlen) ld_raw_planes(c, x0, (int64_t)x0 + dg, raw);
		};
		fetch(0);
		for (uint32_t t = 0; t <= nchunks; ++t) {
			if (t < nchunks) {
				planes_of(raw, sh, qv, sv);
				fetch(t + 1); // (in flight while this round is worked on)
				const uint32_t x0 = wbase + 2048 * t + WNT * lane;
				uint32_t m = ~0u, mc = 0; // positions at and beyond the query's end: lcp() stops there; mc: the mismatches that are counted
				uint2 codes = make_uint2(0, 0); // (the query's planes b0, b1: what the heads' records are cut from)
				uint32_t dirty_w = 0;           // the word's query symbols that are no nucleotides
				if (x0 < c.qlen) {
					m = (qv.b0 ^ sv.b0) | (qv.b1 ^ sv.b1) | (qv.b2 ^ sv.b2);
					codes = make_uint2(qv.b0, qv.b1);
					mc = m & ~(qv.b2 | sv.b2); // both nucleotides (src/model.c:318-320)
					uint32_t inq = ~0u;
					if (c.qlen - x0 < WNT) inq = ~(~0u << (c.qlen - x0)), m |= ~inq, mc &= inq;
					dirty_w = qv.b2 & inq;
					dirty |= dirty_w;
				}
				if (x0 <= e0 && e0 - x0 < WNT) m &= ~0u << (e0 - x0), mc &= ~0u << (e0 - x0); // (what lies before the anchor is none of the window's business)
				if (x0 + WNT <= e0) m = 0, mc = 0;
				G->bits[64 * t + lane] = m;
				L.mring[(64 * t + lane) & 255u] = m;
				*(uint2 *)&L.qring[(2 * (64 * t + lane)) & 511u] = codes;
				{ // (the separators' ring is written only where it holds or held one: whole genomes never touch it)
					const bool any_now = __any(dirty_w != 0);
					if (any_now || ((ring_dirty >> (t & 3u)) & 1u)) L.dring[(64 * t + lane) & 255u] = dirty_w;
					ring_dirty = (ring_dirty & ~(1u << (t & 3u))) | (any_now ? 1u << (t & 3u) : 0u);
					win_dirty = win_dirty || any_now;
				}
				if (!PKNOCK(0)) subst_count(acc, mc, qv, sv);
				{
					const uint64_t any = __ballot(m != 0);
					if (any) {
						const uint32_t l = 63u - (uint32_t)__builtin_clzll(any);
						const uint32_t ml = lane_read(m, l);
						last_mm = uni(wbase + 2048 * t + 32 * l + 31u - (uint32_t)__builtin_clz(ml));
					}
				}
			}
			wave_sync();
			if (t >= 1 && heads_on && !PKNOCK(1)) { // the heads of round T: its words and their neighbours are in the ring
				const uint32_t T = t - 1, w = 64 * T + lane;
				const uint32_t cur = L.mring[w & 255u], nxt = w + 1 < nwords ? L.mring[(w + 1) & 255u] : 0u, prv = w ? L.mring[(w - 1) & 255u] : 0u;
				// a mismatch among the next thr positions / among the thr positions before: smear the bits over thr - 1 more positions
				uint32_t A = __builtin_amdgcn_alignbit(nxt, cur, 1), An = nxt >> 1;   // bit x: position x + 1
				uint32_t B = __builtin_amdgcn_alignbit(cur, prv, 31), Bp = prv << 1;  // bit x: position x - 1
				uint32_t have = 1;
				while (2 * have <= thr) {
					A |= __builtin_amdgcn_alignbit(An, A, have), An |= An >> have;
					B |= __builtin_amdgcn_alignbit(B, Bp, 32 - have), Bp |= Bp << have;
					have *= 2;
				}
				if (have < thr) {
					const uint32_t r = thr - have;
					A |= __builtin_amdgcn_alignbit(An, A, r);
					B |= __builtin_amdgcn_alignbit(B, Bp, 32 - r);
				}
				uint32_t live = ~0u; // a chain that stands behind position x >= end - 1 has left the segment
				const uint32_t x0 = wbase + 32 * w;
				if (x0 + 1 >= end) live = 0;
				else if (end - 1 - x0 < 32) live = (1u << (end - 1 - x0)) - 1u;
				const uint32_t hmask = cur & A & ~B & live;
				const uint32_t nh = (uint32_t)__builtin_popcount(hmask);
				uint32_t hb = wave_scan_add(nh); // (inclusive; made exclusive below)
				const uint32_t total = lane_read(hb, 63);
				hb -= nh;
				if (total) {
					if (total > POOL_CHUNK_HEADS || nheads + total > G->hc) { // dropped: nothing is decided from this round's first head on
						const uint32_t first = uni(wave_min(hmask ? x0 + (uint32_t)__builtin_ctz(hmask) : NOPOS));
						if (first < f_cap) f_cap = first;
						heads_on = false;
					} else {
						for (uint32_t hm = hmask; hm; hm &= hm - 1) L.hl[hb++] = (uint16_t)(32 * lane + (uint32_t)__builtin_ctz(hm));
						wave_sync();
						const uint32_t rbase = nheads; // this round's records go behind those of the rounds before
						for (uint32_t i0 = 0; i0 < total; i0 += 64) {
							const uint32_t i = i0 + lane;
							const bool valid = i < total;
							const uint32_t e = valid ? wbase + 2048 * T + L.hl[i] : 0u;
							const bool ok = valid && e + 65u <= wend; // (its record lies inside the window)
							if (ok) {
								const uint32_t b0 = e + 1 - wbase, wq = b0 >> 5, sh = b0 & 31u, cw = 2 * wq;
								const uint32_t r0 = L.mring[wq & 255u], r1 = L.mring[(wq + 1) & 255u], r2 = L.mring[(wq + 2) & 255u];
								const uint32_t c0 = L.qring[cw & 511u], c1 = L.qring[(cw + 1) & 511u], c2 = L.qring[(cw + 2) & 511u],
											   c3 = L.qring[(cw + 3) & 511u], c4 = L.qring[(cw + 4) & 511u], c5 = L.qring[(cw + 5) & 511u];
								PoolRec rc; // (the ring holds the query's planes: words 2w, 2w + 1 = bit 0, bit 1 of the symbols of bits-word w)
								rc.q2[0] = __builtin_amdgcn_alignbit(c2, c0, sh), rc.q2[1] = __builtin_amdgcn_alignbit(c4, c2, sh);
								rc.q2[2] = __builtin_amdgcn_alignbit(c3, c1, sh), rc.q2[3] = __builtin_amdgcn_alignbit(c5, c3, sh);
								rc.bits[0] = __builtin_amdgcn_alignbit(r1, r0, sh), rc.bits[1] = __builtin_amdgcn_alignbit(r2, r1, sh);
								rc.pos = e;
								rc.dirty = win_dirty ? L.dring[wq & 255u] | L.dring[(wq + 1) & 255u] | L.dring[(wq + 2) & 255u] : 0u; // (a separator in the three words the record is cut from: a superset of its 64 positions' own is enough)
								G->rec[rbase + i] = rc;
							}
							const uint64_t okm = __ballot(ok), bad = __ballot(valid && !ok);
							if (bad) { // (sorted: the first of them is the lowest) too near the window's end: nothing is decided from there on
								const uint32_t first = lane_read(e, __builtin_ctzll(bad));
								if (first < f_cap) f_cap = first;
								heads_on = false;
							}
							nheads += (uint32_t)__builtin_popcountll(okm);
						}
					}
				}
			}
			wave_sync();
		}
		if (lane < 4) G->bits[nwords + lane] = 0; // (behind the window nothing is known)
		subst_flush(acc, hist, true);
	}
	if (PKNOCK(2)) nheads = 0;
	ch.blk_base = NOPOS;
	ch.stuck = 0;
	CSTAT(CS_WINDOWS, 1);
	CSTAT(CS_HEADS, nheads);
	pw.e0 = e0, pw.nchunks = nchunks, pw.nheads = nheads, pw.last_mm = last_mm, pw.f_cap = f_cap, pw.clean = __any(dirty != 0) ? 0u : 1u;
	pool_sync();
	TOCK(tph, PH_STREAM);
}

// Sweeps W and R of the window pw.  Returns true if the chain moved; st is a genuine loop-top state either way.
__device__ __forceinline__ bool pool_resolve(const ScanArgs &a, const PairCtx &c, Chain &ch, PoolLds &L, const PoolScratch *G, uint32_t end, const PoolWin &pw, bool &through, uint32_t (&same)[4]) {
	const uint32_t lane = __lane_id(), thr = c.thr, n = (uint32_t)c.E.n;
	ChainState &st = ch.st;
	const int64_t dg = (int64_t)st.lastS - (int64_t)st.lastQ;
	const uint32_t sd = st.lastS - st.lastQ;
	const uint32_t e0 = pw.e0, wbase = e0 & ~31u, nchunks = pw.nchunks, nheads = pw.nheads, last_mm = pw.last_mm, f_cap = pw.f_cap;
	const uint32_t nwords = 64 * nchunks, Wp = 2048 * nchunks, wend = wbase + Wp;
	const bool clean = pw.clean != 0;
	lds_u32 *hist = (lds_u32 *)L.hist;
	through = false;
	TICK(tph);
	// ---- sweep W: the walks, one lane each; a lane that is done takes the next head (as coop_window's, the window's
	// bits and codes from the head's record)
	{
		uint32_t hk = NOPOS, e = 0, p = 0, Xq = 0, Xs = 0, Xl = 0, nX = 0, next_head = 0;
		uint32_t mx = 0, mn = 0, mq = 0;
		uint32_t cq0 = 0, cq1 = 0, cq2 = 0, cq3 = 0, cb0 = 0, cb1 = 0; // the head's record
		bool hclean = true;                                            // ... holds no separator (a window without one: every record)
		bool parked = false;
		// equal symbols from pp on along the diagonal as far as 32 bits show; seen: a mismatch of the window ends them
		auto run_ahead = [&](uint32_t pp, bool &seen) {
			const uint32_t oc = pp - (e + 1), o = pp - wbase;
			uint32_t v;
			if (oc < 32) {
				v = __builtin_amdgcn_alignbit(cb1, cb0, oc);
			} else if (oc == 32) {
				v = cb1;
			} else {
				const uint32_t wi = o >> 5;
				const uint32_t lo = wi < nwords ? G->bits[wi] : 0u, hi = wi + 1 < nwords ? G->bits[wi + 1] : 0u;
				v = __builtin_amdgcn_alignbit(hi, lo, o & 31u);
			}
			const uint32_t r = v ? (uint32_t)__builtin_ctz(v) : 32u;
			seen = v != 0 && o + r < Wp;
			return r;
		};
		auto generic_probe = [&](uint32_t pp) { return coop_generic_probe(c, pp); };
		auto settle = [&](uint32_t res, uint32_t ra, uint32_t rlen, const Probe &pr, bool have) { // (as coop_window's)
			if (!res && have) {
				if (pr.unique && pr.len >= thr) {
					if (pr.pos == p + sd) {
						const bool same_side = (p + sd < c.border) == (e + sd <= c.border);
						if (!same_side || (Xl && Xl >= 2 * thr))
							res = W_BREAK;
						else
							res = W_OK | (rlen == NOPOS ? W_LUCKY : 0u) | (Xl ? W_HADX : 0u) | (nX << W_NX_SHIFT), ra = p, rlen = rlen == NOPOS ? 0u : pr.len;
					} else {
						const uint32_t endS = Xs + Xl, endQ = Xq + Xl;
						if (Xl && ((pr.pos > endS && p - endQ == pr.pos - endS && (pr.pos < c.border) == (Xs < c.border)) || Xl >= 2 * thr))
							res = W_BREAK;
						else if (++nX > COOP_MAX_X)
							res = W_BREAK;
						else
							Xq = p, Xs = pr.pos, Xl = pr.len;
					}
				}
				if (!res) p += pr.len + 1;
			}
			if (res) {
				if ((res & W_LUCKY) && (ra + sd < c.border) != (e + sd <= c.border)) res = W_BREAK;
				// the equal symbols of the stretch (e, ra) by nucleotide, for sweep R: from the record, if the stretch lies inside it
				uint32_t eq = 0;
				bool eq_valid = false;
				{
					const uint32_t len = ra - e - 1;
					if ((res & W_STATUS) == W_OK && !(res & W_HADX) && hclean && len - 1u < POOL_EQ_MAX) {
						eq_valid = true;
						uint32_t eqlo = ~cb0, eqhi = ~cb1; // the equal positions among the len <= 63 behind the head
						if (len < 32) eqlo &= (1u << len) - 1u, eqhi = 0;
						else eqhi &= (1u << (len - 32)) - 1u;
						const uint32_t n0 = (uint32_t)__builtin_popcount(eqlo & ~cq2 & ~cq0) + (uint32_t)__builtin_popcount(eqhi & ~cq3 & ~cq1);
						const uint32_t n1 = (uint32_t)__builtin_popcount(eqlo & ~cq2 & cq0) + (uint32_t)__builtin_popcount(eqhi & ~cq3 & cq1);
						const uint32_t n2 = (uint32_t)__builtin_popcount(eqlo & cq2 & ~cq0) + (uint32_t)__builtin_popcount(eqhi & cq3 & ~cq1);
						const uint32_t n3 = (uint32_t)__builtin_popcount(eqlo & cq2 & cq0) + (uint32_t)__builtin_popcount(eqhi & cq3 & cq1);
						eq = n0 | (n1 << 6) | (n2 << 12) | (n3 << 18);
					}
				}
				PoolRes rs;
				rs.pos = e, rs.ha = ra, rs.hend = rlen, rs.flag = pool_pack_flag(res, eq, eq_valid);
				G->res[hk] = rs;
				hk = NOPOS;
			}
		};
		// (the parked lanes' turn in front of the loop of the ordinary trips, as in coop_window)
		bool due = false;
		for (;;) {
			if (due) {
				due = false;
				if (hk != NOPOS && parked) {
					uint32_t res = 0, ra = 0, rlen = 0;
					Probe pr;
					pr.len = 0, pr.pos = 0, pr.unique = false;
					bool have = false;
					bool long_diag = false;
					if (mn) {
						bool seen;
						const uint32_t r = run_ahead(p, seen);
						have = coop_probe_multi<false>(c, p, sd, mx, mn, mq, r, seen, pr, long_diag); // (the sorter's records were the ordinary trip's own try: what they left open is looked up in the texts)
					}
					// (measured: without lane_probe in this loop -- such a probe ending the window at its head instead -- the kernel fits 64 registers
					// with 19 spilled, but eight wavefronts per SIMD are no faster than six at equal work, and the windows cut short cost
					// sweep S five times over: profiles/r07_pool/no_generic_probe_occupancy.txt)
					if (!have) pr = generic_probe(p), long_diag = false;
					have = true, parked = false;
#ifdef ANDI_COOP_STATS
					atomicAdd(&g_coop_stats[CS_PROBES], 1ull);
#endif
					if (long_diag && pr.unique) {
						if (wend - p >= 32) {
							const bool same_side = (p + sd < c.border) == (e + sd <= c.border);
							res = (!same_side || (Xl && Xl >= 2 * thr)) ? W_BREAK : (W_OK | W_LUCKY | (Xl ? W_HADX : 0u) | (nX << W_NX_SHIFT));
							ra = p;
						} else {
							pr = generic_probe(p);
						}
					}
					settle(res, ra, rlen, pr, have);
				}
			}
			for (;;) {
				const uint64_t idle = __ballot(hk == NOPOS);
				if (idle && next_head < nheads) {
					const uint32_t my = next_head + (uint32_t)__builtin_popcountll(idle & ((1ull << lane) - 1ull));
					if (hk == NOPOS && my < nheads) {
						const PoolRec rc = G->rec[my];
						hk = my, e = rc.pos, p = e + 1, Xl = 0, nX = 0, parked = false;
						cq0 = rc.q2[0], cq1 = rc.q2[1], cq2 = rc.q2[2], cq3 = rc.q2[3], cb0 = rc.bits[0], cb1 = rc.bits[1];
						hclean = clean || rc.dirty == 0;
					}
					next_head += (uint32_t)__builtin_popcountll(idle);
				}
				const uint64_t busy = __ballot(hk != NOPOS), waiting = __ballot(hk != NOPOS && parked);
				if (!busy) break;
				const bool service = waiting && ((uint32_t)__builtin_popcountll(waiting) >= COOP_PARK || waiting == busy);
#ifdef ANDI_COOP_STATS
				if (lane == (uint32_t)__builtin_ctzll(__ballot(1))) {
					const uint32_t nb = (uint32_t)__builtin_popcountll(service ? waiting : busy & ~waiting);
					atomicAdd(&g_coop_trip_lanes[service ? 1 : 0][nb ? 32 - __builtin_clz(nb) : 0], 1ull);
					atomicAdd(&g_coop_stats[service ? CS_SERVICE : CS_TRIPS], 1ull);
					atomicAdd(&g_coop_stats[service ? CS_SERVICE_LANES : CS_LANE_STEPS], (unsigned long long)__builtin_popcountll(service ? waiting : busy & ~waiting));
				}
#endif
				if (service) {
					due = true;
					break;
				}
				if (hk == NOPOS || parked) continue;
				uint32_t res = 0, ra = 0, rlen = 0;
				Probe pr;
				pr.len = 0, pr.pos = 0, pr.unique = false;
				bool have = false;
				const uint32_t o = p - wbase;
				const bool inwin = p < end && o < Wp;
				bool seen;
				const uint32_t r = run_ahead(inwin ? p : e + 1, seen);
				const uint32_t left = Wp - (inwin ? o : 0u);
				const bool near = Xl == 0 && p + sd < n && p - e <= thr;
				const bool run_ok = seen ? r >= thr : left >= thr;
				res = !inwin ? (Xl ? W_BREAK : (p >= end ? W_EXIT : W_OPEN))
					  : (near && run_ok) ? (W_OK | W_LUCKY)
					  : (near && !seen && left < thr) ? W_OPEN : 0u;
				ra = p;
				if (inwin && Xl) {
					const uint32_t adv = p - Xq;
					if (Xs + adv < n && adv - Xl <= thr) {
						const uint32_t qa = p & ~1u;
						const uint4 d = neq32(ld_query(c, qa), ld_subject_guarded(c, (int64_t)qa + ((int64_t)Xs - (int64_t)Xq)));
						uint32_t l = first_from(d, p & 1u) - (p & 1u);
						if (l > c.qlen - p) l = c.qlen - p;
						if (l >= thr) res = W_BREAK;
					}
				}
				if (inwin && !res) { // the probe: K-mer and the 16 symbols behind it from the record (a walk that has left it: from the query)
					bool on_diag = false;
					mn = 0;
					have = false;
					// (joined contigs -- a window with separators: round 5 sent EVERY probe of such a window to lane_probe, and a set of
					// 100-contig assemblies scanned at half the speed of the same genomes whole; now only a record / a fetch that holds one)
					if (p + 32 <= c.qlen && o + 32 <= Wp) {
						const uint32_t oc = p - (e + 1);
						uint32_t lo = 0, hi = 0;
						bool codes_ok;
						if (oc <= 36) {
							// the record holds the symbols bit-sliced: bit 0 and bit 1 of the 32 from p on, interleaved to 2-bit codes
							const uint32_t pa = (uint32_t)((((uint64_t)cq1 << 32) | cq0) >> oc), pb = (uint32_t)((((uint64_t)cq3 << 32) | cq2) >> oc);
							lo = spread16(pa) | (spread16(pb) << 1), hi = spread16(pa >> 16) | (spread16(pb >> 16) << 1);
							codes_ok = hclean;
						} else {
							const uint4 qv = ld_query(c, p & ~1u); // 32 symbols from an even position on
							const uint32_t s0 = squeeze_codes(qv.x) | (squeeze_codes(qv.y) << 16), s1 = squeeze_codes(qv.z) | (squeeze_codes(qv.w) << 16);
							lo = __builtin_amdgcn_alignbit(s1, s0, 2 * (p & 1u)), hi = s1 >> (2 * (p & 1u)); // (31 symbols are enough: K + 16 <= 29)
							codes_ok = clean || ((qv.x | qv.y | qv.z | qv.w) & 0x44444444u) == 0;
						}
						if (codes_ok) have = coop_probe_codes(c, p, sd, lo, hi, pr, on_diag, mx, mn, mq);
						else WHY(CS_WHY_PRE);
					} else {
						WHY(CS_WHY_PRE);
					}
					if (!have && mn) { // a K-mer with a few occurrences: the sorter's records settle it as a rule (as coop_window's walks)
						bool long_diag;
						if (coop_probe_r2(c, p, sd, mx, mn, mq, r, seen, pr, long_diag)) {
							have = true;
							if (long_diag && pr.unique) { // the diagonal's occurrence is the longest, longer than the bits at hand show
								if (wend - p >= 32) {
									const bool same_side = (p + sd < c.border) == (e + sd <= c.border);
									res = (!same_side || (Xl && Xl >= 2 * thr)) ? W_BREAK : (W_OK | W_LUCKY | (Xl ? W_HADX : 0u) | (nX << W_NX_SHIFT));
								} else {
									have = false; // (the window's end: lane_probe's)
								}
							}
						}
					}
					if (on_diag) {
						pr.unique = true, pr.pos = p + sd, pr.len = r;
						if (!seen && left < 32) have = false;
						else if (!seen) rlen = NOPOS;
					}
					parked = !have;
#ifdef ANDI_COOP_STATS
					if (have) atomicAdd(&g_coop_stats[CS_PROBES], 1ull);
#endif
				}
				settle(res, ra, rlen, pr, have);
			}
			if (!due) break;
		}
	}
	pool_sync();

	TOCK(tph, PH_WALKS);
	// ---- sweep R.  Where the window's knowledge ends:
	uint32_t F = last_mm; // behind the last mismatch nothing is known
	if (f_cap < F) F = f_cap;
	{
		const int64_t b = (int64_t)c.border - dg; // '#': the next anchor lies on the other strand, no right anchor (src/process.c:162)
		if (b >= (int64_t)e0 && b < (int64_t)F) F = (uint32_t)b;
		if (end - 1 < F) { // a chain that stands behind a position >= end - 1 has left the segment
			const uint32_t fe = pool_next_bit(G, nwords, wbase, end - 1 > e0 ? end - 1 : e0);
			if (fe != NOPOS && fe < F) F = fe;
		}
	}
	// what sweep S counted of positions the chain does not reach is taken back: the mismatches of [from, wend)
	auto take_back = [&](uint32_t from) {
		if (PKNOCK(4) || (from - wbase) / 2048u >= nchunks) return;
		SubstAcc acc;
#pragma unroll
		for (int k = 0; k < 12; ++k) acc.n[k] = 0;
		for (uint32_t t = (from - wbase) / 2048u; t < nchunks; ++t) {
			const uint32_t x0 = wbase + 2048 * t + WNT * lane;
			if (x0 >= c.qlen || x0 + WNT <= from) continue;
			const Planes qv = ld_query_planes(c, x0), sv = ld_subject_planes(c, (int64_t)x0 + dg);
			uint32_t mc = ((qv.b0 ^ sv.b0) | (qv.b1 ^ sv.b1) | (qv.b2 ^ sv.b2)) & ~(qv.b2 | sv.b2);
			if (c.qlen - x0 < WNT) mc &= ~(~0u << (c.qlen - x0));
			if (from > x0) mc &= ~0u << (from - x0);
			subst_count(acc, mc, qv, sv);
		}
		subst_flush(acc, hist, false);
	};
	if (e0 >= F) {
		take_back(e0);
		wave_sync();
		return false;
	}
	// The chain hops from head to head; between them every mismatch is followed by a lucky anchor (coop_window's scheme, 64
	// heads per round).  A head the chain came by ("on the path"): the stretch up to its walk's landing is gap positions --
	// or-ed into the bits; its equal symbols are counted here (the mismatches were, in sweep S), from the head's record.
	// Walks that met anchors off the diagonal (rare): the stretch is counted nowhere (its mismatches are taken back) and the
	// anchor before the head only under the conditions of src/process.c:176-186.
	uint32_t cur = e0, kcur = 0; // (cur: after a run of ordinary hops only a lower bound -- where the last anchor BEGINS; the heads it is compared with are mismatches, and an anchor holds none)
	uint32_t have_hop = 0, hop_ha = 0, hop_x = 0; // the last head the chain hopped from: where its anchor begins, W_HADX
	uint32_t extra_anchors = 0, sub_q = 0, sub_r = 0; // (per lane)
	uint32_t eq0 = 0, eq1 = 0, eq2 = 0, eq3 = 0;      // (per lane) equal pairs in the stretches, by nucleotide: summed over the lanes once per window
	// (an LDS add of every lane to one cell is turned into a scalar loop over the lanes by the compiler: 1500 scalar instructions per round of heads)
	bool done = false;
#ifdef POOL_SEPARATE_EBITS
	for (uint32_t w = lane; w < nwords + 4; w += 64) G->ebits[w] = 0;
	pool_sync();
#define POOL_GAPBITS ebits
#else
	// The stretches are or-ed into the window's mismatch bits themselves (round 6; until then a second bitmap, zeroed and read
	// back per window: 32 KB of the scratch traffic of a window of 131072 positions).  Every reader of the bits inside this sweep
	// decides the same with or without a stretch's bits, whichever it happens to see: the first bit at or after a landing
	// (pool_next_bit) is a mismatch -- stretches start at heads, which are mismatches --, and the last bit before a head or
	// before `cur` (pool_prev_bit*) is compared with where the hop before landed: a bit inside that hop's stretch lies before
	// its landing like the mismatches it covers.  Sweep S writes every word anew for the next window.
#define POOL_GAPBITS bits
#endif
	for (uint32_t base = 0; base < nheads && !done; base += 64) {
		const uint32_t k = base + lane;
		const bool valid = k < nheads;
		PoolRes rs;
		rs.pos = NOPOS, rs.ha = 0, rs.hend = 0, rs.flag = 0;
		if (valid) rs = G->res[k];
		const uint32_t pos = rs.pos, fl = pool_flag_of(rs.flag), la = rs.ha, eqw = rs.flag >> 8; // (eqw: the stretch's equal symbols, from the walk)
		const bool eq_valid = (rs.flag & POOL_EQ_VALID) != 0;
		// Where the anchor the walk landed on ends is the chain's next stand.  A probe's anchor: the result says; a lucky anchor's end
		// is the next mismatch behind the landing -- looked up only where it matters: the next head, a mismatch itself, bounds it
		const bool landed = (fl & W_STATUS) == W_OK, lucky = landed && (fl & W_LUCKY);
		const uint32_t endk = landed && !lucky ? la + rs.hend : NOPOS;
		uint32_t nxtpos = lane_above(pos);
		if (lane == 63) nxtpos = k + 1 < nheads ? G->res[k + 1].pos : NOPOS;
		const bool unusual = valid && (!landed || pos >= F || nxtpos < la || (lucky ? nxtpos >= F : (nxtpos < endk || endk >= F)));
		bool onpath = false;
		const uint32_t carry_have = have_hop, carry_ha = hop_ha, carry_x = hop_x; // (the hop before this round's first)
		if (kcur < base) kcur = base;
		{ // heads the chain has jumped over already
			const uint64_t at = __ballot(valid && pos >= cur);
			const uint32_t first = at ? (uint32_t)__builtin_ctzll(at) : 64u;
			if (base + first > kcur) kcur = base + first;
		}
		while (!done && kcur < base + 64 && kcur < nheads) {
			CSTAT(CS_HOPS, 1);
			const uint32_t lo = kcur - base;
			const uint64_t ev = __ballot(unusual) & (~0ull << lo);
			const uint32_t j = ev ? (uint32_t)__builtin_ctzll(ev) : 64u; // every head before it: hopped, on to the next
			if (lane >= lo && lane < j && valid) onpath = true;
			if (j > lo) {
				const uint32_t last = (j < 64 ? j : 64u) - 1u;
				const uint32_t lastv = nheads - base - 1u < last ? nheads - base - 1u : last;
				cur = lane_read(la, lastv); // (a lower bound, see above)
				have_hop = 1, hop_ha = cur, hop_x = lane_read(fl, lastv) & W_HADX;
			}
			if (j >= 64) {
				kcur = base + 64;
				break;
			}
			const uint32_t pj = lane_read(pos, j), lj = lane_read(la, j), fj = lane_read(fl, j);
			uint32_t ej = lane_read(endk, j);
			if (pj < F && (fj & W_STATUS) == W_OK && (fj & W_LUCKY)) ej = pool_next_bit(G, nwords, wbase, lj);
			if (pj >= F) { // the chain reaches a position where the window's knowledge ends before this head
				cur = F, done = true;
			} else if (ej == NOPOS) { // its walk did not land (or where the anchor ends is not in the window): the chain stops at the head
				cur = pj, done = true;
				if ((fj & W_STATUS) == W_BREAK) ch.stuck = 1; // (scan_coop.hip: Chain.stuck)
			} else { // hopped; the chain stands where the anchor ends
				if (lane == j) onpath = true;
				cur = ej;
				have_hop = 1, hop_ha = lj, hop_x = fj & W_HADX;
				if (ej >= F) {
					done = true;
				} else {
					const uint64_t at = __ballot(valid && pos >= ej) & (~0ull << j);
					kcur = base + (at ? (uint32_t)__builtin_ctzll(at) : 64u);
				}
			}
		}
		// the heads of this round the chain came by
		const uint64_t onm = __ballot(onpath);
#ifdef ANDI_COOP_STATS
		CSTAT(CS_ONPATH, __builtin_popcountll(onm));
		CSTAT(CS_X, __builtin_popcountll(__ballot(onpath && (fl & W_HADX))));
#endif
		if (onm) {
			// the hop before each: the head on the path before it
			const uint64_t before_me = onm & ((1ull << lane) - 1ull);
			const uint32_t pl = before_me ? 63u - (uint32_t)__builtin_clzll(before_me) : 0u;
			const uint32_t s_ha = (uint32_t)__shfl((int)la, (int)pl), s_x = (uint32_t)__shfl((int)fl, (int)pl) & W_HADX;
			const uint32_t pred_have = before_me ? 1u : carry_have, pred_ha = before_me ? s_ha : carry_ha, pred_x = before_me ? s_x : carry_x;
			if (!PKNOCK(6)) {
				// (1) the stretches [o0, o1) into the bitmap of gap positions: three words as a rule (a record's 64 positions)
				const uint32_t o0 = pos - wbase, o1 = la - wbase, wd0 = o0 >> 5;
				if (onpath && !PKNOCK(7)) {
#pragma unroll
					for (int t = 0; t < 3; ++t) {
						const uint32_t wd = wd0 + t;
						uint32_t m = ~0u;
						if (t == 0) m &= ~0u << (o0 & 31u);
						if (32 * wd + 32 > o1) m &= (1u << (o1 & 31u)) - 1u;
						if (32 * wd < o1) (void)__hip_atomic_fetch_or(&G->POOL_GAPBITS[wd], m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); // (two stretches may meet in a word)
					}
					for (uint32_t wd = wd0 + 3; 32 * wd < o1; ++wd) {
						uint32_t m = ~0u;
						if (32 * wd + 32 > o1) m &= (1u << (o1 & 31u)) - 1u;
						(void)__hip_atomic_fetch_or(&G->POOL_GAPBITS[wd], m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
					}
				}
				// (2) the equal symbols of an ordinary stretch (its mismatches were counted in sweep S): from the head's record; stretches
				// longer than a record (one head in thirty) with all lanes, one after the other -- a lane's own loop over their text
				// was run by every round of 64 heads
				const uint32_t len = la - pos - 1;
				const bool ord = onpath && !(fl & W_HADX) && len != 0, fast = ord && eq_valid; // (fast: the walk counted them, as pool_pack_flag says)
				if (fast && !PKNOCK(8)) eq0 += eqw & 63u, eq1 += (eqw >> 6) & 63u, eq2 += (eqw >> 12) & 63u, eq3 += (eqw >> 18) & 63u;
				for (uint64_t sl = PKNOCK(9) ? 0ull : __ballot(ord && !fast); sl; sl &= sl - 1) {
					const uint32_t l = (uint32_t)__builtin_ctzll(sl);
					const uint32_t q0 = lane_read(pos, l) + 1, ln = lane_read(len, l);
					pool_count_equal_coop(c, q0, (uint32_t)((int64_t)q0 + dg), ln, eq0, eq1, eq2, eq3);
				}
				// (3) a walk that met anchors off the diagonal (rare): its stretch is counted nowhere -- the mismatches are taken back --, the anchor
				// before the head only under the conditions of src/process.c:176-186
				for (uint64_t hx = __ballot(onpath && (fl & W_HADX)); hx; hx &= hx - 1) {
					const uint32_t l = (uint32_t)__builtin_ctzll(hx);
					const uint32_t q0 = lane_read(pos, l), ln = lane_read(la, l) - q0;
					pool_uncount_coop(c, hist, q0, (uint32_t)((int64_t)q0 + dg), ln);
				}
				if (onpath && (fl & W_HADX)) {
					// the anchor before the head: the one the hop before landed on if no mismatch lies between, else the one behind the last mismatch
					uint32_t a_before, lw_before = 1;
					if (pos == e0) {
						a_before = st.lastQ, lw_before = st.lwra;
					} else {
						const uint32_t pm = pool_prev_bit_lane(G, wbase, pos);
						if (pred_have && (pm == NOPOS || pm < pred_ha)) a_before = pred_ha, lw_before = pred_x ? 0u : 1u;
						else a_before = pm + 1;
					}
					// (the counting pass below counts every anchor that ends at a gap: this one is taken back if it does not count)
					if (!(lw_before || pos - a_before >= 2 * thr)) sub_q += (pos - a_before) >> 2, sub_r += (pos - a_before) & 3u;
					extra_anchors += (fl >> W_NX_SHIFT) & 7u;
				}
			}
		}
	}
	if (!done) cur = F; // past the last head: lucky anchors up to where the window's knowledge ends
	if (cur == e0) {
		take_back(e0);
		wave_sync();
		return false;
	}
	CSTAT(CS_MOVED, 1);
	CSTAT(CS_COVERED, cur - e0);
	pool_sync(); // (the stretches are in G->ebits: the atomics are done; the loads below read the device's copy, not this CU's cache)
	// the anchor that ends at cur: the one the last hop landed on if no mismatch lies between, else the one behind the mismatch before cur
	uint32_t aQ, lw = 1;
	{
		const uint32_t pm = pool_prev_bit(G, wbase, cur < wend ? cur : wend); // (an anchor may end behind the window: nothing is known there, no bit is set)
		if (have_hop && (pm == NOPOS || pm < hop_ha)) aQ = hop_ha, lw = hop_x ? 0u : 1u;
		else aQ = pm + 1;
	}
	TOCK(tph, PH_HOPS);
	// ---- the anchors: one ends at every position that starts a gap (positions e0 ... cur - 1 of the bits, stretches included);
	// 8192 positions per round, lane l takes the four words of positions 128 l ...
	{
		uint32_t q_acc = 0, r_acc = 0, n_acc = 0;
		uint32_t carry_before = st.lastQ + 1u; // (the last position in no anchor before the round's words) + 2: the anchor before e0 starts at lastQ
		uint32_t carry_top = 0;
		const uint32_t span = (cur < wend ? cur : wend) - wbase; // (behind the window no bit is set)
		// (the words of round w0 + 256 are fetched while round w0 is counted -- two 8-byte loads past this CU's cache per lane and round:
		// the stretches were or-ed in at the device's level)
		typedef unsigned long long u64x2_t __attribute__((ext_vector_type(2)));
		auto load_gapbits = [&](uint32_t w0) {
			u64x2_t v = (u64x2_t)(0ull);
			if (32 * w0 < span && w0 + 4 * lane < nwords) {
				const unsigned long long *ep = (const unsigned long long *)&G->POOL_GAPBITS[w0 + 4 * lane];
				v.x = __hip_atomic_load(ep, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), v.y = __hip_atomic_load(ep + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			}
			return v;
		};
		u64x2_t ev_next = load_gapbits(0);
		for (uint32_t w0 = 0; 32 * w0 < span && !PKNOCK(3); w0 += 256) {
			uint32_t u[4], before = 0;
#ifdef POOL_SEPARATE_EBITS
			const uint4 mv = *(const uint4 *)&G->bits[w0 + 4 * lane]; // (64 words of padding behind the window's)
#else
			const uint4 mv = make_uint4(0, 0, 0, 0); // (the bits with the stretches in them: ev, read past this CU's cache)
#endif
			const uint4 ev = make_uint4((uint32_t)ev_next.x, (uint32_t)(ev_next.x >> 32), (uint32_t)ev_next.y, (uint32_t)(ev_next.y >> 32));
			ev_next = load_gapbits(w0 + 256);
#pragma unroll
			for (int j = 0; j < 4; ++j) {
				const uint32_t w = w0 + 4 * lane + j, x0 = wbase + 32 * w;
				uint32_t rm = 0;
				if (x0 < cur && x0 + WNT > e0) {
					rm = ~0u;
					if (e0 > x0) rm &= ~0u << (e0 - x0);
					if (cur - x0 < WNT) rm &= (1u << (cur - x0)) - 1u;
				}
				u[j] = rm && w < nwords ? (pick(mv, j) | pick(ev, j)) & rm : 0u; // (an anchor may end behind the window: no bits there)
				if (u[j]) before = x0 + 31u - (uint32_t)__builtin_clz(u[j]) + 2u;
			}
			const uint32_t scan = wave_scan_max(before);
			before = lane_below(scan);
			if (lane == 0 || before < carry_before) before = carry_before;
			uint32_t prev_top = lane_below((u[3] >> 31));
			if (lane == 0) prev_top = carry_top;
#pragma unroll
			for (int j = 0; j < 4; ++j) {
				const uint32_t x0 = wbase + 32 * (w0 + 4 * lane + j), uu = u[j];
				uint32_t gs = uu & ~((uu << 1) | prev_top); // gap starts: a position in no anchor whose predecessor is in one
				n_acc += (uint32_t)__builtin_popcount(gs);
				for (; gs; gs &= gs - 1) {
					const uint32_t b = (uint32_t)__builtin_ctz(gs), below = uu & ((1u << b) - 1u);
					const uint32_t pv = below ? x0 + 31u - (uint32_t)__builtin_clz(below) : before - 2u; // the last position before x0 + b in no anchor
					const uint32_t len = x0 + b - 1u - pv; // (pv may be lastQ - 1 = -1: unsigned wrap is fine)
					q_acc += len >> 2, r_acc += len & 3u;
				}
				if (uu) before = x0 + 31u - (uint32_t)__builtin_clz(uu) + 2u;
				prev_top = uu >> 31;
			}
			{
				const uint32_t top = lane_read(scan, 63);
				if (top > carry_before) carry_before = top;
				carry_top = lane_read((u[3] >> 31), 63);
			}
		}
		ch.quarter += wave_sum(q_acc) - wave_sum(sub_q), ch.rest += wave_sum(r_acc) - wave_sum(sub_r);
		same[0] += wave_sum(eq0), same[1] += wave_sum(eq1), same[2] += wave_sum(eq2), same[3] += wave_sum(eq3);
		const uint32_t nodes = wave_sum(n_acc);
		CSTAT(CS_NODES, nodes);
		ch.anchors += nodes + wave_sum(extra_anchors);
	}
	TOCK(tph, PH_FINAL);
	take_back(cur);
	TOCK(tph, PH_STRETCH);
	through = !done || cur + 256 >= wend; // (the chain got to the window's end: the next window may be longer)
	st.p = cur + 1, st.lastS = (uint32_t)((int64_t)aQ + dg), st.lastQ = aQ, st.lastLen = cur - aQ, st.lwra = lw;
	wave_sync();
	return true;
}

__device__ __forceinline__ bool pool_window(const ScanArgs &a, const PairCtx &c, Chain &ch, PoolLds &L, const PoolScratch *G, uint32_t end, uint32_t chunks, bool &through, uint32_t (&same)[4]) {
	PoolWin pw;
	pool_stream(a, c, ch, L, G, end, chunks, pw);
	return pool_resolve(a, c, ch, L, G, end, pw, through, same);
}

// ------------------------------------------------------------------ the kernel: persistent wavefronts take the segments in order
#ifndef POOL_OCC
#define POOL_OCC 7 /* wavefronts per SIMD: 72 registers, 19 spilled -- round 5: bench set 5.66 / 5.09 / 4.92 ms at 4 / 5 / 6; round 6, first session: 7 and 8 a fifth and a third
                      slower than 6 (45 / 108 spilled); with sweep W's loop split (19 / 34 spilled): C4 shape 19.8 / 19.8 / 22.9 ms at 6 / 7 / 8, C3-like 1.63 / 1.56 / 1.82 */
#endif
// One segment: mode G until the chain has a diagonal, windows while it stays on it.
__device__ __forceinline__ void pool_segment(const ScanArgs &a, PoolLds &L, const PoolScratch *G, uint32_t sub, uint32_t wseg) {
	const uint32_t lane = __lane_id();
	if (a.subjects[sub].mode != ANDI_MODE_PROBE) return;
	const uint32_t qidx = uni(a.seg2query[wseg]);
	if (a.self[sub] == (int64_t)qidx) return;
	const uint32_t seg_in_q = wseg - uni(a.qseg_start[qidx]);
	PairCtx c = make_ctx(a, sub, qidx);
	const uint32_t start = seg_in_q * a.seg, end = start + a.seg < c.qlen ? start + a.seg : c.qlen;
	const size_t slot = (size_t)sub * a.total_segs + wseg;
	const uint32_t n = (uint32_t)c.E.n, thr = c.thr;
	uint8_t *route = a.route ? a.pair_class + (size_t)sub * a.nq + qidx : nullptr;
	auto give_up = [&]() {
		if (lane == 0) a.restitch_count[ANDI_ROUTE_ANY_LEFT] = 1;
		if (lane == 0) __hip_atomic_store(route, (uint8_t)(__hip_atomic_load(route, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) | ANDI_ROUTE_LEFT), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	};
	auto given_up = [&]() { return route && (uni(__hip_atomic_load(route, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) & (ANDI_ROUTE_COOP | ANDI_ROUTE_LEFT)) != ANDI_ROUTE_COOP; };
	if (given_up()) return;
	wave_sync();
	Chain ch;
	uint32_t my_g = 0;
	PoolWin pw;
	if (lane < 16) L.hist[lane] = 0;
	ch.st = seg_in_q == 0 ? initial_state() : cold_state(start, n);
	ch.quarter = ch.rest = ch.anchors = ch.marked = 0, ch.blk_base = NOPOS, ch.stuck = 0;
	ChainState &st = ch.st;
	wave_sync();

	CSTAT(CS_SEGMENTS, 1);
	TICK(tall);
	uint32_t same[4] = {0, 0, 0, 0}; // equal pairs the windows found in gaps, by nucleotide
	// How long the next window should be (rounds of 2048 positions): sweep S streams a window to its end before anything is known
	// about it, so a window that ends early -- the chain leaves its diagonal: an indel, the separator of a joined contig -- has
	// streamed the rest for nothing.  A window the chain went through doubles the next one; one it did not go through sets the next to
	// what it covered (as a power of two, four rounds at least).  Round 5 always began again at the longest window: genomes of 100
	// contigs each (a diagonal every 10 000 positions) scanned 17 times slower than the same genomes whole -- 384 against 20 ms for the
	// C4 shape -- and now 2.2 times (profiles/r07_pool/join_ab.txt).  The hint outlives the windows' loop: mode G takes over at every
	// break, and the windows behind it begin where the last ones left off.
	// (the persistent wavefronts begin their segments with windows of different lengths -- 64, 24, 40, 56 rounds by their number -- so that
	// the six of a SIMD do not all stream, or all walk, at the same time: C4 shape 20.36 -> 20.08 ms; the bench set forced onto this kernel
	// 4.30 -> 4.36: profiles/r07_pool/stagger_ab.txt)
	uint32_t hint_chunks = a.pool_first == 64u ? (blockIdx.x & 3u) == 0 ? 64u : (blockIdx.x & 3u) == 1 ? 24u : (blockIdx.x & 3u) == 2 ? 40u : 56u : a.pool_first;
	// windows one after the other while the chain stays canonical on the diagonal and moves; false: the pair was handed back
	auto windows = [&]() {
		// (behind an anchor of thousands of symbols the next mismatch is far: genomes 1e-5 apart -- a window finds one or two, and nothing is
		// decided behind the last: short windows first)
		uint32_t chunks = st.lastLen >= 4096 ? 4u : hint_chunks;
		bool through = false;
		for (;;) {
			if (!(st.p < end && st.lastQ + st.lastLen < c.qlen)) break;
			const uint32_t from = st.lastQ + st.lastLen;
			pool_stream(a, c, ch, L, G, end, chunks, pw);
			const bool moved = pool_resolve(a, c, ch, L, G, end, pw, through, same);
#ifdef POOL_FIXED_WINDOWS /* (A/B: round 5's policy) */
			if (!through) {
				chunks = a.pool_first;
			} else
#endif
			if (moved && through) {
				chunks = 2 * chunks < G->maxchunks ? 2 * chunks : G->maxchunks;
			} else {
				const uint32_t covered = moved ? st.p - from : 0u;
				chunks = 4;
				while (chunks < G->maxchunks && 2048u * chunks < covered) chunks *= 2;
			}
			hint_chunks = chunks;
			if (!moved) break;
			if (given_up()) return false;
			if (ch.stuck) break; // (the chain stands at a head whose walk gave up: mode G's step, not another window at the same place)
		}
		return true;
	};
	while (st.p < end) {
		CSTAT(CS_G_STEPS, 1);
		++my_g;
		if (route) {
			if (my_g > a.route_giveup + (a.seg >> 12)) return give_up();
			if (given_up()) return;
		}
		// ---- one step of mode G (src/process.c:153-197), as k_coop_cold's
		bool found = false, lucky = false;
		uint32_t curS = 0, curLen = 0;
		if (lucky_applies(st, n, thr)) {
			curS = st.lastS + (st.p - st.lastQ);
			curLen = coop_lcp(c, st.p, curS, c.qlen - st.p, route ? COOP_TRIAL_LCP : ~0u);
			if (curLen == NOPOS) return give_up();
			found = lucky = curLen >= thr;
		}
		if (!found) {
			if (ch.blk_base == NOPOS || st.p < ch.blk_base || st.p - ch.blk_base >= 64) {
				const uint32_t p = st.p + lane;
				CSTAT(CS_BLOCKS, 1);
				Probe pr;
				pr.len = 0, pr.pos = 0, pr.unique = false;
				if (p < c.qlen) {
					LWin w;
					w.q0 = EMPTY, w.dg = NO_DIAG;
					pr = lane_probe(c, p, w);
				}
				wave_sync();
				L.pl[lane] = pr.len | (pr.unique ? 0x80000000u : 0u), L.pp[lane] = pr.pos;
				ch.blk_base = st.p;
				wave_sync();
			}
			const uint32_t v = uni(L.pl[st.p - ch.blk_base]);
			curLen = v & 0x7fffffffu, curS = uni(L.pp[st.p - ch.blk_base]);
			found = (v >> 31) && curLen >= thr;
		}
		if (found) {
			coop_account<false>(c, ch, L, curS);
			st.lastS = curS, st.lastQ = st.p, st.lastLen = curLen;
		}
		st.p += curLen + 1;
		if (found) {
			wave_sync();
			coop_note_anchor(a, slot, ch, L);
		}
		if (found && lucky && !windows()) return;
	}
	TOCK(tall, 7);
	wave_sync();
#ifdef ANDI_COOP_STATS
	if (lane == 0) atomicMax(&g_coop_max[0], my_g);
#endif
	(void)my_g;
	// ---- what pass B reads (scan.h)
	if (lane == 0) {
		ColdMark *m = a.marks + slot * ANDI_COLD_MARKS;
		if (!ch.marked) m->st.pad[0] = 0; // unused mark
		ChainState out = st;
		out.pad[0] = 0, out.pad[1] = ch.anchors < 255 ? ch.anchors : 255, out.pad[2] = 0;
		a.cold_exit[slot] = out;
		a.exit_p[slot] = st.p;
	}
	if (lane < 16) {
		uint32_t v = L.hist[lane];
		if (lane == 0 || lane == 5 || lane == 10 || lane == 15) v += ch.quarter;
		if (lane == 15) v += ch.rest;
		v += lane == 0 ? same[0] : lane == 5 ? same[1] : lane == 10 ? same[2] : lane == 15 ? same[3] : 0u;
		a.cold_counts[slot * 16 + lane] = v;
	}
}

// the kernel: persistent wavefronts, a scratch each, take the segments in order
__global__ __launch_bounds__(64, POOL_OCC) void k_pool_cold(ScanArgs a) {
	__shared__ PoolLds L;
	const PoolScratch Gs = pool_scratch_at(a.pool_scratch, blockIdx.x, a.pool_maxchunks, a.pool_hc);
	const uint32_t items = a.total_segs * a.nsub;
	uint32_t item = blockIdx.x; // the first one; then whatever comes next
	while (item < items) {
		const uint32_t row = item / a.total_segs;
		pool_segment(a, L, &Gs, a.sub_order ? uni(a.sub_order[row]) : row, item % a.total_segs);
		wave_sync();
		uint32_t nx = 0;
		if (__lane_id() == 0) nx = atomicAdd(a.pool_ticket, 1u);
		item = gridDim.x + lane_read(nx, 0);
	}
}

