// scan_lane.hip — passes A and B of the anchor scan (dist_anchor, src/process.c:141-214)
// with ONE LANE PER CHAIN on nibble-packed sequences.
//
// scan.hip runs a chain on a group of G lanes that share every byte comparison.
// A chain step is mostly control (src/process.c:153-197), and the groups of a
// wavefront diverge, so there a wavefront pays the whole step for 64/G chains.
// Here every lane owns a chain (segment) of its own: 64 chains per wavefront pay
// each instruction once.  What makes that affordable is the packing: a symbol is
// 4 bits (A C G T ! ; # NUL = 0..7, so equal bytes <=> equal nibbles), a 16-byte
// load is a window of 32 symbols, and a lane compares, counts and forms K-mer
// codes on its own registers with shifts, masks and population counts -- no
// cross-lane traffic at all.  The subject text is kept in two alignments (N0,
// N1 = N0 shifted by one symbol) so that a window of the subject can start at
// any symbol while the query window starts at an even one.
//
// The chain logic, the probe table and the three-pass scheme (cold chains,
// stitch, reduce) are those of scan.hip; the results are bit-identical.
#include "lane_chain.h"
#include "knobs.h"

#include <algorithm>

namespace {

// Adaptive mode, step 1: one wavefront per pair samples the longest match at 64 evenly
// spaced query positions.  Their mean estimates the distance between mismatches, i.e.
// how long the true and the cold chain of a segment take to meet; the pair's segments
// are made seg_factor times that (as a power-of-two multiple of seg0, at most 8 seg0).
// Short segments keep the lanes of a wavefront close together in memory and the work
// items small; long ones keep the stitching of low-divergence pairs cheap.
// (A block of several wavefronts per pair: every one of them takes the samples of the mean -- the same, so that all know
// what the pair is a candidate for -- and its share of the further samples of a routed call: the kernel's time is the
// chain of dependent probes of one wavefront, 185 us with one wavefront per pair, 100 us with eight.)
[[maybe_unused]] constexpr uint32_t EST_WAVES = 4; // (calls of up to 4096 pairs)
constexpr uint32_t EST_WAVES_FEW = 8; // (calls of up to 1024 pairs: a round of further samples per wavefront)
// (Calls of more pairs than that, `many`: the device is full with one wavefront per pair; four pairs share a block.  What
// the pairs add up -- the layout's wavefront counts -- is k_pair_totals' business: 90 000 pairs adding to three words one
// by one from here took a millisecond, a million 30 ms.)
__global__ __launch_bounds__(64 * EST_WAVES_FEW) void k_pair_estimate(ScanArgs a, bool many) {
	__shared__ uint32_t s_shorts, s_runs;
	const uint32_t est_waves = many ? 1u : blockDim.x >> 6;
	const uint32_t pair = many ? blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6) : blockIdx.x, wave = many ? 0u : threadIdx.x >> 6;
	const uint32_t lane = threadIdx.x & 63u;
	if (pair >= a.nsub * a.nq) return; // (many)
	const uint32_t sub = pair / a.nq, qidx = pair % a.nq;
	const bool first = many ? lane == 0 : threadIdx.x == 0; // who writes the pair's results
	if (a.self[sub] == (int64_t)qidx) {
		if (first) a.pair_class[pair] = 0, a.pair_waves[pair] = 0;
		return;
	}
	if (!many) { // (the same branch in every wavefront of the block)
		if (threadIdx.x == 0) s_shorts = 0, s_runs = 0;
		__syncthreads();
	}
	PairCtx c = make_ctx(a, sub, qidx);
	// (routed calls, short queries: 32 or 16 samples -- a sample per kbp still; calls of thousands of such queries are
	// mostly sampling otherwise)
	const uint32_t nl_shift = !a.route || c.qlen >= 32768u ? 6u : c.qlen >= 16384u ? 5u : 4u, nl = 1u << nl_shift;
	const uint32_t p = (uint32_t)(((uint64_t)(2 * lane + 1) * c.qlen) >> (nl_shift + 1));
	LWin w;
	w.q0 = EMPTY, w.dg = NO_DIAG;
	// a pair whose mean is seg0 * 8 / factor gets the longest segments anyway: samples are cut at twice
	// that (a long match is followed 32 symbols per round trip, and the wavefront waits for its longest)
	// (a routed call with one segment length for its lanes asks only whether the mean is below ANDI_SPARSE_MATCH: three round trips)
	const uint32_t cap = a.adaptive ? 16 * a.seg0 / (a.seg_factor ? a.seg_factor : 1) / 2 + 32 : 96u;
	Probe r;
	r.len = 0;
	if (lane < nl) r = lane_probe(c, p, w, cap);
	uint32_t sum = r.len < cap ? r.len : cap;
#pragma unroll
	for (int d = 32; d; d >>= 1) sum += (uint32_t)__shfl_xor((int)sum, d);
	sum <<= 6u - nl_shift; // (as of 64 samples)
	// A pair whose matches are long on average may still spend most of its chain steps where it has none: 10 % of
	// unrelated sequence (genomic islands: a step every 13 nucleotides there) are most of the steps of a pair with
	// a mismatch every 500, and probing is what k_lane_quad is slow at (the realistic set: pass A 11.2 ms with
	// such pairs in k_lane_quad, 10.6 ms without).  So a candidate is sampled at 192 more positions, and goes to
	// k_lane_quad only if its short matches are no more than its mean explains (matches end at random: a fraction
	// 1 - exp(-threshold / mean) of the positions sees less than the threshold), within two standard deviations.
	bool islands = false, guess = false, far_clean = false;
	const bool quad_cand = (sum >> 6) >= a.quad_min_match && (sum >> 6) < ANDI_ISLAND_MEAN_MAX && a.quad_min_match != 0;
	// routed calls: would the pair suit pass A by wavefronts?  (one whole segment of that kernel's at least)
	// (one whole segment of that kernel's at least, or a query of a few windows: with many short queries a wavefront's chain is a query)
	// Joined contigs: every separator of the query or of the subject ends the pair's diagonal (the genomes' contigs are cut at different
	// places), and a wavefront kernel pays a window for every end.  C4 shape of genomes of 100 / 20 contigs, per call: k_pool_cold 117 / 48 ms,
	// k_coop_cold 87 / 37, the lane scan 54 / 44 (whole genomes: 20 by k_pool_cold); bench set of 100-contig genomes (an end every 24 500
	// positions): lanes 8.2, k_coop_cold 9.5 ms.  So by the mean distance between the ends: below 20480 positions -- two of k_coop_cold's
	// windows -- the lane scan, below 262144 -- two windows of k_pool_cold -- no candidate of that kernel (C4 shape of 8-contig genomes, an end
	// every 131 000 positions: k_pool_cold 33.7, k_coop_cold 30.0 ms; profiles/r07_pool/join_routing.txt, join_sweep.txt).  (32768 until
	// k_coop_cold's second session of round 6: with that kernel a sixth faster the C4 shape of 40-contig genomes -- an end every 26 000
	// positions -- takes 41.3 ms by wavefronts against 50.9 by lanes, the bench set of 100-contig genomes -- 24 500 -- 10.6 either way, the C4
	// shape of 100-contig genomes -- 10 500 -- 71 against 59: profiles/r07_coop/join_threshold.txt.)
	uint32_t break_dist = ~0u;
	if (a.qsep && a.self[sub] >= 0) {
		const uint32_t ends = a.qsep[qidx] + a.qsep[(uint32_t)a.self[sub]];
		if (ends) break_dist = c.qlen / ends;
	}
	const bool coop_cand = a.route && (sum >> 6) < 512u && c.qlen >= (a.route_seg < ANDI_ROUTE_MIN_QLEN ? a.route_seg : ANDI_ROUTE_MIN_QLEN) && break_dist >= 20480u; // (matches of 512 symbols and more on average: k_lane_quad's, always)
	if (coop_cand) { // (wave-uniform)
		// Unrelated stretches are contiguous: where a sample sees less than a threshold's worth of matching symbols, four
		// more are taken, 128 symbols apart.  Five short ones in a row come about by chance at the fifth power of the rate of
		// short samples -- a pair 3 % apart sees a third of its samples short anyway, one 6 % apart two thirds: 1 % and 10 %
		// of them five times --, inside a stretch without homology every time: 512 samples tell 10 % of unrelated
		// sequence from none at six standard deviations whatever the divergence.  (The excess of short samples over what
		// the mean match length explains, the test of k_lane_quad's candidates below, is too weak here: it missed one
		// structured pair in six and suspected one clean pair in nine; three in a row miss the pairs 5 % and more apart.)
		const uint32_t T = c.thr + 3; // (beyond what chance matches reach: a 13-mer occurs in a 10 Mbp text one time in seven)
		// (calls of thousands of pairs: 256 samples -- one wavefront per pair there, and the chain of its dependent probes is
		// what the kernel takes: 0.77 -> 0.4 ms for the C4 shape's 24 680 pairs; 10 % of unrelated sequence still stand out
		// by four standard deviations)
		// (short queries: a sample per 2048 symbols at most -- the follow-ups of one span 640)
		uint32_t nrounds = est_waves > 1 ? 8u : 4u;
		while (nrounds > 1 && c.qlen < nrounds * (64u * 2048u)) nrounds >>= 1;
		const uint32_t nsamples = nl * nrounds;
		uint32_t shorts = 0, runs = 0;
		for (uint32_t k = wave; k < nrounds; k += est_waves) {
			const uint32_t pk = (uint32_t)(((uint64_t)(2 * nrounds * lane + 2 * k + 1) * c.qlen) / (2u * nl * nrounds));
			bool all_short = lane < nl;
			for (uint32_t j = 0; j < 5 && all_short; ++j) {
				const uint32_t pj = pk + 128 * j;
				LWin wk;
				wk.q0 = EMPTY, wk.dg = NO_DIAG;
				all_short = pj + T < c.qlen && lane_probe(c, pj, wk, T + 1).len < T;
				if (j == 0 && all_short) ++shorts;
			}
			if (all_short) ++runs;
		}
#pragma unroll
		for (int d = 32; d; d >>= 1) shorts += (uint32_t)__shfl_xor((int)shorts, d), runs += (uint32_t)__shfl_xor((int)runs, d);
		if (!many) {
			if (lane == 0) atomicAdd(&s_shorts, shorts), atomicAdd(&s_runs, runs);
			__syncthreads(); // (coop_cand is the same in all wavefronts of the block: they took the same samples)
			shorts = s_shorts, runs = s_runs;
		}
		// The rate f0 of short samples OUTSIDE unrelated stretches: from the short samples that did not turn into a run --
		// f0 - f0^5 of the samples in homologous sequence, hardly any inside a stretch without homology (Newton on the lower
		// branch; pairs beyond 6 % or so have no solution there and are taken for suspicious: the lane scan's anyway).
		// Taking the rate from all short samples let the stretches themselves raise the expectation: one structured pair
		// in ninety passed, and ground through its islands in generic steps.
		const float ns = (float)nsamples;
		float g = (float)(shorts - runs) / ns, f0;
		if (g > 0.53f) g = 0.53f;
		f0 = g;
		for (int it = 0; it < 5; ++it) {
			const float f4 = f0 * f0 * f0 * f0;
			f0 -= (f0 - f0 * f4 - g) / (1.f - 5.f * f4);
		}
		const float f5 = f0 * f0 * f0 * f0 * f0, expect = ns * f5;
		islands = (float)runs > expect + 3.f * sqrtf(expect * (1.f - f5)) + 3.f;
		// ... or the pair is simply far apart: runs of five as often as the rate of ALL short samples gives by itself (a clean pair 8 %
		// apart: 76 % short, 25 % runs; a pair 3 % apart with a fifth of unrelated sequence: 51 % short, 20 % runs where that rate
		// explains 4 %).  Such a pair stays the lane scan's -- beyond 6 % the equation above has no well-conditioned solution --, but it is
		// no pair with unrelated stretches.
		if (islands) {
			const float fa = (float)shorts / ns, fa5 = fa * fa * fa * fa * fa, most = ns * fa5;
			far_clean = (float)runs <= 2.f * most + 10.f; // (clean pairs 4-8 % apart show up to 1.8 times the runs that rate gives)
		}
		// (small calls: pairs so far apart that nearly every sample is short -- runs of five tell nothing there -- are the
		// wavefront kernel's: what it hands back costs a small call less than the lanes' chains cost every such pair)
		// (... and so are the pairs near the top of f0 - f0^5, 4 ... 7 % apart, where the rate has no well-conditioned solution:
		// nine clean pairs of ninety 6 % apart were suspected and made pass A of 10 x 1 Mbp 1.65 instead of 0.6 ms)
		// (Unless the runs exceed even what the rate of ALL short samples explains -- an upper bound of f0 that the stretches
		// themselves raise, good enough up to 7 % or so: 12 x 1 Mbp of structured genomes went by wavefronts without it,
		// 7.3 ms per call against 4.0; what passes both is left to the kernel's own limit of generic steps.)
		if (a.route_all_few && ((sum >> 6) < ANDI_SPARSE_MATCH || (float)(shorts - runs) >= 0.45f * ns)) {
			const float fa = (float)shorts / ns, fa5 = fa * fa * fa * fa * fa, most = ns * fa5;
			islands = (float)runs > most + 3.f * sqrtf(most * (1.f - fa5)) + 3.f;
			guess = !islands; // (k_pair_route: only in calls whose other pairs show no unrelated stretches either)
		}
	} else if (quad_cand && nl == 64) { // (wave-uniform; calls that are not routed, and the pairs with the longest matches)
		uint32_t shorts = r.len < c.thr ? 1u : 0u;
		for (uint32_t k = 1; k < 4; ++k) {
			const uint32_t pk = (uint32_t)(((uint64_t)(8 * lane + 2 * k + 1) * c.qlen) >> 9); // between the first samples
			LWin wk;
			wk.q0 = EMPTY, wk.dg = NO_DIAG;
			const Probe rk = lane_probe(c, pk, wk, c.thr + 1);
			shorts += rk.len < c.thr ? 1u : 0u;
		}
#pragma unroll
		for (int d = 32; d; d >>= 1) shorts += (uint32_t)__shfl_xor((int)shorts, d);
		const float expect = 256.f * (1.f - __expf(-(float)c.thr / (float)(sum >> 6)));
		islands = (float)shorts > expect + 2.f * sqrtf(expect * (1.f - expect / 256.f)) + 2.f;
	}
	if (first) {
		const uint32_t want = (sum >> 6) * a.seg_factor; // mean match length * factor
		uint32_t cls = 0;
		while (cls < a.max_class && (a.seg0 << cls) < want) ++cls;
		const uint32_t seg = a.seg0 << cls, nseg = (c.qlen + seg - 1) / seg;
		// bit 7: the pair's matches are long enough for pass A with the streams fetched by quads (k_lane_quad)
		// (routed calls) pairs the lane scan is better at -- matches hardly reaching the anchor threshold (divergence beyond
		// some 6 %: their chains probe at nearly every step and meet their neighbours' slowly; with long segments pass B has
		// few lanes for those replays) -- take it where they are more than a tenth of the call; a few of them ride along
		// with the wavefront kernel (k_pair_route).  (Pairs of k_lane_quad's class -- mean match 128 ... 511 -- were treated
		// the same way at first: the wavefront kernel is the faster one for them, route_soft_match.)
		// (a pair that is merely far apart: a candidate of the wavefront kernel like the pairs small calls cannot judge -- not in a call
		// that has pairs with unrelated stretches (k_pair_route: it may be one of them after all, and would be handed back), and a soft
		// one: the lane scan's where such pairs are many.  The bench set's 18 farthest pairs ride along with the wavefront kernel
		// instead of keeping the lane kernels busy beside it for 1.7 ms: step 7.63 -> 7.50 ms)
		const bool far = islands && far_clean;
		const bool soft = (sum >> 6) < ANDI_SPARSE_MATCH || (sum >> 6) >= a.route_soft_match || far;
		if (far) islands = false, guess = true;
		const bool quad = (sum >> 6) >= a.quad_min_match && !(islands && (sum >> 6) < ANDI_ISLAND_MEAN_MAX);
		a.pair_class[pair] = (uint8_t)(cls | (quad ? 0x80u : 0u) | (coop_cand && !islands ? ANDI_ROUTE_COOP : 0u) | (soft ? ANDI_ROUTE_SOFT : 0u) |
										 (coop_cand && islands ? ANDI_ROUTE_LEFT : 0u) | (coop_cand && guess ? ANDI_ROUTE_GUESS : 0u) |
										 (a.route && (sum >> 6) >= a.pool_match && (sum >> 6) < 4096u && break_dist >= 262144u ? ANDI_ROUTE_POOLCAND : 0u));
		a.pair_waves[pair] = (nseg + 63) / 64;
		// (scan.h: sub_order; not in calls of thousands of pairs -- a few subjects with thousands of queries each, alike: their costs stay
		// zero and the order is the subjects' own; 24 680 additions to eight words took 0.07 ms)
		if (a.sub_cost && !many) atomicAdd(&a.sub_cost[sub], (float)c.qlen / (float)((sum >> 6) < 8u ? 8u : (sum >> 6)));
		if (islands) atomicAdd(&a.restitch_count[ANDI_STRUCT_WAVES], (nseg + 63) / 64); // (k_pair_route: a call of structured genomes?)
	}
}

// step 2: exclusive prefix sums of the pairs' wavefront counts, any number of pairs: sums of
// 1024 pairs per block, a scan of those sums by one block, the scan inside every block
__device__ __forceinline__ uint32_t block_scan_1024(uint32_t v, uint32_t *s_part) { // inclusive, 1024 threads
	s_part[threadIdx.x] = v;
	__syncthreads();
	for (uint32_t d = 1; d < 1024; d <<= 1) { // Hillis-Steele
		const uint32_t o = threadIdx.x >= d ? s_part[threadIdx.x - d] : 0;
		__syncthreads();
		s_part[threadIdx.x] += o;
		__syncthreads();
	}
	return s_part[threadIdx.x];
}

__global__ __launch_bounds__(1024) void k_pair_block_sums(ScanArgs a) {
	__shared__ uint32_t s_part[1024];
	const uint32_t P = a.nsub * a.nq, i = blockIdx.x * 1024 + threadIdx.x;
	const uint32_t incl = block_scan_1024(i < P ? a.pair_waves[i] : 0u, s_part);
	if (threadIdx.x == 1023) a.pair_bsum[blockIdx.x] = incl;
}

__global__ __launch_bounds__(1024) void k_pair_block_offsets(ScanArgs a) { // one block: exclusive scan of the block sums, in place
	__shared__ uint32_t s_part[1024];
	const uint32_t P = a.nsub * a.nq, nb = (P + 1023) / 1024, per = (nb + 1023) / 1024;
	const uint32_t first = threadIdx.x * per, last = first + per < nb ? first + per : nb;
	uint32_t sum = 0;
	for (uint32_t i = first; i < last; ++i) sum += a.pair_bsum[i];
	uint32_t run = block_scan_1024(sum, s_part) - sum;
	for (uint32_t i = first; i < last; ++i) {
		const uint32_t v = a.pair_bsum[i];
		a.pair_bsum[i] = run;
		run += v;
	}
	if (threadIdx.x == 1023) a.pair_wave0[P] = s_part[1023]; // wavefronts in use
}

__global__ __launch_bounds__(1024) void k_pair_offsets(ScanArgs a) {
	__shared__ uint32_t s_part[1024];
	const uint32_t P = a.nsub * a.nq, i = blockIdx.x * 1024 + threadIdx.x;
	const uint32_t v = i < P ? a.pair_waves[i] : 0u;
	const uint32_t incl = block_scan_1024(v, s_part);
	if (i < P) {
		const uint32_t w0 = a.pair_bsum[blockIdx.x] + incl - v;
		a.pair_wave0[i] = w0;
		if ((a.pair_class[i] & 0x80u) && v) { // its wavefronts go on k_lane_quad's list
			uint32_t *list = (uint32_t *)a.defer_list;
			const uint32_t base = atomicAdd(&a.restitch_count[ANDI_QUAD_WAVES], v);
			for (uint32_t k = 0; k < v; ++k) list[base + k] = w0 + k;
		}
	}
}

// routed calls: the wavefronts of all pairs, of those the lane scan would like (not marked for the wavefront kernel, or
// soft), and of those it gets whatever the others do -- what k_pair_route decides by.  (A sum per block: a million pairs
// adding to three words one by one took 30 ms.)
__global__ __launch_bounds__(1024) void k_pair_totals(ScanArgs a) {
	__shared__ uint32_t s_sum[4];
	const uint32_t P = a.nsub * a.nq, pair = blockIdx.x * 1024 + threadIdx.x;
	if (threadIdx.x < 4) s_sum[threadIdx.x] = 0;
	__syncthreads();
	uint32_t all = 0, sparse = 0, hard = 0, isl = 0;
	if (pair < P) {
		const uint32_t cls = a.pair_class[pair];
		all = a.pair_waves[pair]; // (none for a query that is the subject itself)
		hard = (cls & ANDI_ROUTE_COOP) ? 0u : all;
		sparse = (cls & ANDI_ROUTE_SOFT) ? all : hard;
		isl = (cls & ANDI_ROUTE_LEFT) ? all : 0u; // (until k_pair_route: unrelated stretches seen)
	}
#pragma unroll
	for (int d = 32; d; d >>= 1) {
		all += (uint32_t)__shfl_xor((int)all, d);
		sparse += (uint32_t)__shfl_xor((int)sparse, d);
		hard += (uint32_t)__shfl_xor((int)hard, d);
		isl += (uint32_t)__shfl_xor((int)isl, d);
	}
	if ((threadIdx.x & 63u) == 0) atomicAdd(&s_sum[0], all), atomicAdd(&s_sum[1], sparse), atomicAdd(&s_sum[2], hard), atomicAdd(&s_sum[3], isl);
	__syncthreads();
	if (threadIdx.x == 0) {
		if (s_sum[0]) atomicAdd(&a.restitch_count[ANDI_ALL_WAVES], s_sum[0]);
		if (s_sum[1]) atomicAdd(&a.restitch_count[ANDI_SPARSE_WAVES], s_sum[1]);
		if (s_sum[2]) atomicAdd(&a.restitch_count[ANDI_HARD_WAVES], s_sum[2]);
		if (s_sum[3]) atomicAdd(&a.restitch_count[ANDI_ISLAND_WAVES], s_sum[3]);
	}
}

// routed calls: which pairs take pass A by wavefronts (they get no wavefronts in the lane layout)
__device__ __forceinline__ uint32_t route_pair(const ScanArgs &a, uint32_t pair, uint32_t &pool_segs, uint32_t &coop_segs) {
	uint32_t cls = a.pair_class[pair];
	const bool lanes_few = 10 * a.restitch_count[ANDI_SPARSE_WAVES] <= a.restitch_count[ANDI_ALL_WAVES]; // (the lane scan's and those it would like)
	// (small calls: the soft pairs are the wavefront kernel's as well -- 3 x 1 Mbp 10 % apart, BASELINE's configs[0]: pass A
	// 0.34 ms by wavefronts, 2.0 ms by lanes, whose chains take a segment's few hundred steps one after the other)
	if ((cls & ANDI_ROUTE_COOP) && (cls & ANDI_ROUTE_SOFT) && !lanes_few && !a.route_all_few) cls &= ~ANDI_ROUTE_COOP;
	// (small calls: the pairs the sampling could not judge stay with the lane scan where a twentieth of the call shows
	// unrelated stretches -- structured genomes: 12 x 1 Mbp took 5.4 ms with such pairs tried by wavefronts and handed
	// back, 3.9 ms by lanes)
	if ((cls & ANDI_ROUTE_GUESS) && 20 * a.restitch_count[ANDI_ISLAND_WAVES] > a.restitch_count[ANDI_ALL_WAVES]) cls &= ~ANDI_ROUTE_COOP;
	// A call of structured genomes -- a third of its wavefronts belong to pairs with unrelated stretches (and not merely far apart) --
	// gives the suspected pairs segments twice as long: what such pairs cost is pass B, where every segment boundary inside a stretch
	// without homology is a replay of hundreds of steps by one lane, and pass A loses nothing (the structured set's passes B/C 6.8 ->
	// 4.4 ms, its step 19.0 -> 16.6).  All of them or none: with the segment lengths mixed pass A was the slower for it (10.1 against
	// 9.4 ms), and clean pairs 4-10 % apart -- suspected too -- lose 5-10 % with long segments where they are most of a call.
	if ((cls & (ANDI_ROUTE_LEFT | ANDI_ROUTE_GUESS)) && !(cls & ANDI_ROUTE_COOP) && (cls & 3u) < a.max_class && 3 * a.restitch_count[ANDI_STRUCT_WAVES] >= a.restitch_count[ANDI_ALL_WAVES]) {
		++cls;
		const uint32_t seg = a.seg0 << (cls & 3u);
		a.pair_waves[pair] = ((a.qlen[pair % a.nq] + seg - 1) / seg + 63) / 64;
	}
	const bool pool = (cls & ANDI_ROUTE_POOLCAND) != 0;
	cls &= ~(ANDI_ROUTE_SOFT | ANDI_ROUTE_LEFT | ANDI_ROUTE_GUESS | ANDI_ROUTE_POOLCAND);
	// Small calls (route_all_few): pass A by wavefronts takes a fraction of a millisecond, and a lane's chain over one
	// segment as long as it ever does -- six pairs of 9900 left to the lane scan (100 x 30 kbp) made pass A 1.1 ms instead
	// of 0.6.  Where the lane scan's pairs are that few, the wavefront kernel takes them all (and hands back what it must).
	if (a.route_all_few && 20 * a.restitch_count[ANDI_HARD_WAVES] <= a.restitch_count[ANDI_ALL_WAVES] && a.self[pair / a.nq] != (int64_t)(pair % a.nq))
		cls |= ANDI_ROUTE_COOP;
	if (cls & ANDI_ROUTE_COOP) { // (which of the wavefront kernels: scan.h)
		const uint32_t segs = (a.qlen[pair % a.nq] + a.route_seg - 1) / a.route_seg;
		coop_segs += segs, pool_segs += pool ? segs : 0u;
	}
	a.pair_class[pair] = (uint8_t)cls;
	if (cls & ANDI_ROUTE_COOP) a.pair_waves[pair] = 0;
	return a.pair_waves[pair];
}

__global__ __launch_bounds__(256) void k_pair_route(ScanArgs a) {
	const uint32_t P = a.nsub * a.nq, pair = blockIdx.x * 256 + threadIdx.x;
	uint32_t mine = 0, pool_segs = 0, coop_segs = 0; // wavefronts the pair keeps in the lane layout; the wavefront kernel's segments
	if (pair < P) mine = route_pair(a, pair, pool_segs, coop_segs);
#pragma unroll
	for (int d = 32; d; d >>= 1)
		mine += (uint32_t)__shfl_xor((int)mine, d), pool_segs += (uint32_t)__shfl_xor((int)pool_segs, d), coop_segs += (uint32_t)__shfl_xor((int)coop_segs, d);
	if ((threadIdx.x & 63u) == 0 && mine) atomicAdd(&a.restitch_count[ANDI_LANE_WAVES], mine); // (the host looks: which kernel goes first)
	if ((threadIdx.x & 63u) == 0 && coop_segs) atomicAdd(&a.restitch_count[ANDI_COOP_SEGS], coop_segs), atomicAdd(&a.restitch_count[ANDI_POOL_SEGS], pool_segs); // (... and which wavefront kernel)
}

// routed calls: the subjects by falling cost (scan.h: sub_order); one block
__global__ __launch_bounds__(1024) void k_sub_order(ScanArgs a) {
	for (uint32_t s = threadIdx.x; s < a.nsub; s += 1024) {
		if (a.nsub > 8192u) { // (ranking is quadratic: calls of that many subjects take them as they come)
			a.sub_order[s] = s;
			continue;
		}
		const float c = a.sub_cost[s];
		uint32_t rank = 0;
		for (uint32_t t = 0; t < a.nsub; ++t) {
			const float o = a.sub_cost[t];
			rank += (o > c || (o == c && t < s)) ? 1u : 0u;
		}
		a.sub_order[rank] = s;
	}
}

// routed calls, after pass A: the pairs the wavefront kernel handed back get the wavefronts of the second lane layout
// (a: that layout's arguments), every other pair none
__global__ __launch_bounds__(256) void k_pair_leftover(ScanArgs a) {
	const uint32_t P = a.nsub * a.nq, pair = blockIdx.x * 256 + threadIdx.x;
	if (pair >= P) return;
	const uint32_t sub = pair / a.nq, qidx = pair % a.nq;
	uint32_t cls = a.pair_class[pair], waves = 0;
	if (a.self[sub] != (int64_t)qidx && (cls & ANDI_ROUTE_COOP) && (cls & ANDI_ROUTE_LEFT)) {
		cls = (cls & ~(ANDI_ROUTE_COOP | ANDI_ROUTE_LEFT)) | ANDI_ROUTE_L2;
		const uint32_t seg = a.seg0 << (cls & 3u), nseg = (a.qlen[qidx] + seg - 1) / seg;
		a.pair_class[pair] = (uint8_t)cls;
		waves = (nseg + 63) / 64;
	}
	a.pair_waves[pair] = waves;
}

// routed calls: who took what (the context's counters, read with the timings)
__global__ __launch_bounds__(256) void k_route_count(ScanArgs a) {
	const uint32_t P = a.nsub * a.nq, pair = blockIdx.x * 256 + threadIdx.x;
	unsigned long long nt[2] = {0, 0};
	uint32_t back = 0;
	if (pair < P && a.self[pair / a.nq] != (int64_t)(pair % a.nq)) {
		const uint32_t cls = a.pair_class[pair];
		nt[(cls & ANDI_ROUTE_COOP) ? 0 : 1] = a.qlen[pair % a.nq];
		back = (cls & ANDI_ROUTE_L2) ? 1u : 0u;
	}
	// (a sum per wavefront: 90 000 pairs adding to two words one by one took a millisecond)
#pragma unroll
	for (int d = 32; d; d >>= 1) {
		nt[0] += (unsigned long long)__shfl_xor((long long)nt[0], d);
		nt[1] += (unsigned long long)__shfl_xor((long long)nt[1], d);
		back += (uint32_t)__shfl_xor((int)back, d);
	}
	if ((threadIdx.x & 63u) == 0) {
		if (nt[0]) atomicAdd(&a.route_nt[0], nt[0]);
		if (nt[1]) atomicAdd(&a.route_nt[1], nt[1]);
		if (back) atomicAdd(&a.route_nt[2], (unsigned long long)back);
	}
}

// ------------------------------------------------------------------ pass A

__device__ __forceinline__ uint32_t dpp_quad_floor(uint32_t v, int ctrl) { // quad_perm: broadcast lane `ctrl`, or lane ^ 1 (4), lane ^ 2 (5)
	switch (ctrl) {
		case 0: return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x00, 0xF, 0xF, true);
		case 1: return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x55, 0xF, 0xF, true);
		case 2: return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xAA, 0xF, 0xF, true);
		case 3: return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xFF, 0xF, 0xF, true);
		case 4: return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true);
		default: return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, true);
	}
}


__device__ __forceinline__ bool pos_in(const LWin &w, uint32_t p) {
	return w.q0 != EMPTY && p >= w.q0 && p - w.q0 < WNT;
}

// Pass A with ALL streaming along diagonals done by QUADS of lanes (ANDI_LANE_STREAM=2).
//
// The floors (k_stream_floor, profiles/r02_stream/floor.txt): a lane that reads its own two streams 16 bytes at a
// time is one request per load, 70-110 G of which the memory pipeline takes per second -- 3.2 ms for the bench set
// with nothing but the comparison, and lane_step's pass A runs at 87 % of that request rate; four neighbouring
// lanes reading 64 consecutive bytes are ONE request: 0.8 ms, 4 x 4 transposes included.
//
// Every trip of the wavefront's loop a lane says where it looks next -- at the start of a step: the gap behind its
// last anchor and the position p on the lucky diagonal (or p alone, for the probe's K-mer); in the middle of a
// comparison: where that stands -- and its quad fetches the 128 symbols of its two streams from there (four rounds:
// the four lanes read 64 consecutive bytes of one lane's stream each; a transpose over the quad by DPP hands every
// lane its own).  Then the lane goes on with its step out of those registers: lcp() of lucky_anchor
// (src/process.c:59-65, 82-100) or of the one occurrence a probe found, window by window, until it ends or the
// registers do (it is then resumed on the next trip); when it has ended, the rest of the step -- the probe if the
// lucky attempt failed, the counting, the bookkeeping -- is lane_step's code.  A probe follows a match only 64
// symbols by itself (lane_probe's cap); one that goes on is handed to the quads.
#define QUAD_STAGE (4 * 65) /* uint4 per wavefront: four rounds of 64 pieces, rows one piece apart from a multiple of the banks */
template <bool EXACT>
__device__ __forceinline__ void lane_cold_quad(const ScanArgs &a, const LaneItem &it, uint32_t *s_hist, uint4 *s_stage) {
	Tally tally;
	tally_begin<1>(tally, s_hist + threadIdx.x);
	PairCtx c = make_ctx(a, it.sub, it.qidx);
	const uint32_t n = (uint32_t)c.E.n, thr = c.thr, qi = threadIdx.x & 3u;
	ChainState st = it.seg_in_q == 0 ? initial_state() : cold_state(it.start, n);
	LWin w;
	w.q0 = EMPTY, w.dg = NO_DIAG;
	const size_t slot = it.slot;
	ColdMark *marks = a.marks + slot * ANDI_COLD_MARKS;
	uint32_t anchors = 0;
	uint4 Bq[4], Bs[4]; // 128 symbols of the query from b0 (even) and of the subject against them on diagonal bdg
	uint32_t b0 = EMPTY;
	int32_t bdg = NO_DIAG;
	// the comparison in progress: 0 none, 1 lucky_anchor's, 2 along the one occurrence a probe found
	uint32_t mode = 0, curS = 0, curLen = 0;
	bool accounted = false;
	bool active = it.valid && st.p < it.end;
	constexpr uint32_t PROBE_CAP = 64;
	// A match longer than a segment is found by the cold chain of every segment it covers, each from its own start,
	// and each would follow it to its end: m segments, m^2 / 2 segments' worth of comparing.  But the lane to the
	// right holds the next segment of the same pair: once a comparison has reached the segment's end, and the
	// neighbour's first anchor starts right there on the same diagonal, the rest of the match is that anchor.
	uint32_t fQ = EMPTY, fS = 0, fLen = 0; // the chain's first anchor

	// make w the piece of the registers that holds query position x (false: they do not hold it)
	auto point_w_at = [&](uint32_t x) -> bool {
		bool hit = false;
		if (b0 == EMPTY) return false;
#pragma unroll
		for (int k = 0; k < 4; ++k) {
			const uint32_t w0 = b0 + WNT * k;
			if (x - w0 < WNT) {
				w.q0 = w0, w.q = Bq[k], w.dg = bdg, hit = true;
				if (bdg != NO_DIAG) w.s = Bs[k];
			}
		}
		return hit;
	};

	while (__any(active)) {
#ifdef ANDI_LANE_STATS
		{ // trips of the wavefront's loop (x4), lanes at work in them (x5), lanes in the middle of a comparison (x6)
			const uint64_t on = __ballot(active), mid = __ballot(active && mode != 0);
			if (__lane_id() == (uint32_t)__builtin_ctzll(on)) {
				STAT(ST_X4);
				atomicAdd(&g_lane_stats[ST_X5], (unsigned long long)__builtin_popcountll(on));
				atomicAdd(&g_lane_stats[ST_X6], (unsigned long long)__builtin_popcountll(mid));
			}
		}
#endif
		const bool has_nb = (threadIdx.x & 63u) != 63u;
		uint32_t nbQ = (uint32_t)__shfl_down((int)fQ, 1), nbS = (uint32_t)__shfl_down((int)fS, 1), nbLen = (uint32_t)__shfl_down((int)fLen, 1);
		bool nb_end = false;
		const bool at_end = active && mode != 0 && st.p + curLen >= it.end;
		if (at_end && !has_nb && it.end < c.qlen) { // the next segment is another wavefront's: what its chain has published
			const unsigned long long pub = __hip_atomic_load(a.first_pub + slot + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			nbQ = pub != ~0ull ? it.end : EMPTY, nbS = (uint32_t)(pub >> 32), nbLen = (uint32_t)pub;
		}
		if (at_end && it.end < c.qlen && nbQ == it.end && nbS - nbQ == curS - st.p) { // (it.end < qlen: the lane to the right may hold another query's first segment)
			curLen = (it.end - st.p) + nbLen, nb_end = true;
			STAT(ST_FINAL_SA); // (diagnostic builds count the matches ended by the neighbour in this slot)
		}
		// ---- where does the lane look next?
		uint32_t want = 0;
		int32_t wdg = NO_DIAG;
		if (active && !nb_end) {
			if (mode == 0) { // the top of a step (src/process.c:153)
				accounted = false;
				if (lucky_applies(st, n, thr)) {
					const uint32_t gap = st.p - st.lastQ - st.lastLen;
					curS = st.lastS + (st.p - st.lastQ), curLen = 0, mode = 1;
					want = st.p - (gap < 16 ? gap : 16), wdg = (int32_t)(curS - st.p);
					STAT(ST_LUCKY_TRY);
				} else {
					want = st.p; // the probe's K-mer
				}
			} else {
				want = st.p + curLen, wdg = (int32_t)(curS - st.p);
			}
		}
		// ---- its registers, filled by its quad when one of the quad's lanes lacks what it wants
		const bool lacks = active && !nb_end && (b0 == EMPTY || want < b0 || want - b0 >= 4 * WNT || (wdg != NO_DIAG && wdg != bdg));
		if (__any(lacks)) {
			const uint64_t lb = __ballot(lacks);
			if ((lb >> (__lane_id() & ~3u)) & 0xFu) { // (uniform in the quad: all its active lanes take new registers from where they look)
				const uint32_t nb0 = want & ~1u;
				const uint64_t qbase = (uint64_t)(uintptr_t)c.Qn;
				STAT(ST_LCP_RELOAD);
#ifdef ANDI_LANE_STATS
				if (__lane_id() == (uint32_t)__builtin_ctzll(__ballot(1))) STAT(ST_X7); // trips with a fetch
#endif
#pragma unroll
				for (int r = 0; r < 4; ++r) {
					const bool o_on = dpp_quad_floor(active ? 1u : 0u, r) != 0;
					const bool o_s = dpp_quad_floor(wdg != NO_DIAG ? 1u : 0u, r) != 0;
					const uint32_t o_q = dpp_quad_floor(nb0, r), o_sa = dpp_quad_floor(nb0 + (uint32_t)wdg, r);
					const uint64_t o_base = dpp_quad_floor((uint32_t)qbase, r) | ((uint64_t)dpp_quad_floor((uint32_t)(qbase >> 32), r) << 32);
					Bq[r] = Bs[r] = make_uint4(0, 0, 0, 0);
					if (o_on) {
						Bq[r] = ld_u128_unaligned((g_u8p)(uintptr_t)o_base + ((o_q + WNT * qi) >> 1)); // (the owner's query; the subject is the wavefront's)
						if (o_s) Bs[r] = ld_subject(c, (int32_t)(o_sa + WNT * qi));
					}
				}
				// every lane its own: piece i of owner r came to lane i in round r.  Through LDS -- written as loaded
				// (round r, lane l at [65 r + l]: contiguous), read back by the owner (its round, its quad's four
				// lanes; the odd row length spreads the owners over the banks) -- the transposition costs the vector
				// ALUs, which bound this kernel, nothing (two 4 x 4 transposes over the quad by DPP: 128 instructions).
				uint4 *stage = s_stage + (threadIdx.x >> 6) * QUAD_STAGE;
				const uint32_t ln = threadIdx.x & 63u;
#pragma unroll
				for (int r = 0; r < 4; ++r) stage[65 * r + ln] = Bq[r];
#pragma unroll
				for (int k = 0; k < 4; ++k) Bq[k] = stage[65 * qi + (ln & ~3u) + k];
#pragma unroll
				for (int r = 0; r < 4; ++r) stage[65 * r + ln] = Bs[r];
#pragma unroll
				for (int k = 0; k < 4; ++k) Bs[k] = stage[65 * qi + (ln & ~3u) + k];
				b0 = active ? nb0 : EMPTY, bdg = wdg;
			}
		}
		if (!active) continue;

		// ---- the comparison, out of the registers
		bool ended = mode == 0 || nb_end; // (no comparison to make: straight to the probe; or the neighbour knew the rest)
		if (mode != 0 && !nb_end) {
			const uint32_t maxlen = c.qlen - st.p;
			uint32_t pos = st.p + curLen;
#pragma unroll
			for (int k = 0; k < 4; ++k) {
				const uint32_t w0 = b0 + WNT * k;
				if (!ended && pos - w0 < WNT) {
					const uint4 d = neq32(Bq[k], Bs[k]);
					const uint32_t o = pos - w0, f = first_from(d, o);
					curLen += f - o, pos += f - o;
					if (f < WNT || curLen >= maxlen) ended = true;
				}
			}
			if (curLen > maxlen) curLen = maxlen;
			if (!ended && !accounted && mode == 1 && curLen >= thr && maxlen >= thr) {
				// certainly an anchor already: count the gap behind it while the registers hold it (src/process.c:157-190)
				if (!point_w_at(st.lastQ + st.lastLen)) w.q0 = EMPTY;
				lane_account<EXACT>(c, st, tally, w, curS);
				accounted = true;
			}
		}
		if (!ended) continue; // resumed on the next trip

		// ---- the rest of the step (src/process.c:113-197)
		STAT(ST_STEP);
		bool found = mode != 0 && curLen >= thr; // (mode 2: the occurrence is the only one)
		if (!found && mode != 2) {
			if (!point_w_at(st.p)) w.q0 = EMPTY; // the probe starts from the window that holds p
			Probe pr = lane_probe(c, st.p, w, PROBE_CAP);
			if (pr.len >= PROBE_CAP && PROBE_CAP < c.qlen - st.p) {
				if (pr.unique) { // the match goes on: follow it with the quads
					curS = pr.pos, curLen = PROBE_CAP, mode = 2, accounted = false;
					continue;
				}
				pr = lane_probe(c, st.p, w); // several occurrences match that far: the whole answer, by the lane itself
			}
			curS = pr.pos, curLen = pr.len;
			found = pr.unique && curLen >= thr;
		}
		if (found) {
			if (!accounted) {
				if (mode != 0) (void)point_w_at(st.lastQ + st.lastLen); // the gap behind a lucky anchor is in the registers (lane_count_gap checks)
				lane_account<EXACT>(c, st, tally, w, curS);
			}
			st.lastS = curS, st.lastQ = st.p, st.lastLen = curLen;
		}
		st.p += curLen + 1;
		mode = 0;
		if (found && ++anchors == 1) {
			*(uint4 *)marks[0].first = make_uint4(st.lastQ, st.lastS, st.lastLen, 0);
			fQ = st.lastQ, fS = st.lastS, fLen = st.lastLen;
			if ((threadIdx.x & 63u) == 0 && fQ == it.start)
				__hip_atomic_store(a.first_pub + slot, ((unsigned long long)fS << 32) | fLen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		}
		if (found && anchors >= 2 && anchors < 2 + ANDI_COLD_MARKS) { // remember the state after anchors 2, 3, 4
			ColdMark *m = marks + (anchors - 2);
			ChainState ms = st;
			ms.pad[0] = 1;
			m->st = ms;
			uint32_t v[16];
#pragma unroll
			for (int t = 0; t < 16; ++t) v[t] = tally.hist[t * BLOCK];
			v[0] += tally.quarter + tally.same[0], v[5] += tally.quarter + tally.same[1];
			v[10] += tally.quarter + tally.same[2], v[15] += tally.quarter + tally.rest + tally.same[3];
			uint4 *mc = (uint4 *)m->counts;
#pragma unroll
			for (int t = 0; t < 4; ++t) mc[t] = make_uint4(v[4 * t], v[4 * t + 1], v[4 * t + 2], v[4 * t + 3]);
		}
		active = st.p < it.end;
	}
	if (!it.valid) return;
	for (uint32_t k = anchors < 2 ? 0 : anchors - 1; k < ANDI_COLD_MARKS; ++k) marks[k].st.pad[0] = 0; // unused marks

	st.pad[1] = anchors < 255 ? anchors : 255;
	a.cold_exit[slot] = st;
	a.exit_p[slot] = st.p;
	tally_finish<1>(tally);
	uint32_t out[16];
#pragma unroll
	for (int t = 0; t < 16; ++t) out[t] = tally.hist[t * BLOCK];
	uint4 *dst = (uint4 *)(a.cold_counts + slot * 16);
#pragma unroll
	for (int t = 0; t < 4; ++t) dst[t] = make_uint4(out[4 * t], out[4 * t + 1], out[4 * t + 2], out[4 * t + 3]);
}

template <bool EXACT>
__global__ __launch_bounds__(BLOCK, 4) void k_lane_quad(ScanArgs a) { // (4 wavefronts per SIMD, whatever the block)
	__shared__ uint32_t s_hist[16 * BLOCK];
	__shared__ uint4 s_stage[WAVES_PER_BLOCK * QUAD_STAGE];
	if (!a.adaptive && a.subjects[blockIdx.y].mode != ANDI_MODE_PROBE) return;
	// a.quad_listed: the kernel's wavefronts take the list of those that have work here in order (k_pair_offsets made
	// it), the others return at once; else (experiments) every wavefront of the call looks whether its pair is this
	// kernel's.  The kernel runs beside k_lane_cold on a device that k_lane_cold's blocks fill; taking its own
	// wavefronts in the call's order, its work trailed behind -- its blocks are large and find a place late -- and
	// ended up running alone after k_lane_cold had finished (C4 shape: 4.5 of 41.7 ms).  From the list, all of it is
	// dispatched first.
	uint32_t wave = ~0u;
	if (a.quad_listed) {
		const uint32_t k = blockIdx.x * WAVES_PER_BLOCK + (threadIdx.x >> 6);
		if (k >= a.restitch_count[ANDI_QUAD_WAVES]) return;
		wave = ((const uint32_t *)a.defer_list)[k];
	}
	LaneItem it = lane_item(a, wave);
	if (!lane_is_mine(a, it, true)) it.valid = false; // (a wavefront is one pair's: all or none)
	if (!__any(it.valid)) return;
	lane_cold_quad<EXACT>(a, it, s_hist, s_stage);
}


// The lanes of a wavefront take consecutive segments of one query, so they see the
// same divergence and stay in step.  (Persistent lanes that fetch their next segment
// from a counter when done were measured 15-50 % slower: they mix pairs of different
// divergence in one wavefront.)
template <bool EXACT, int OCC, bool PER_PAIR>
__global__ __launch_bounds__(BLOCK, OCC) void k_lane_cold(ScanArgs a) {
	__shared__ uint32_t s_hist[16 * BLOCK];
	if (!PER_PAIR && a.subjects[blockIdx.y].mode != ANDI_MODE_PROBE) return;
	const LaneItem it = lane_item<BLOCK, PER_PAIR ? 1 : 2>(a);
	if (!it.valid || !lane_is_mine(a, it, false)) return;
	Tally tally;
	tally_begin<1>(tally, s_hist + threadIdx.x);

	PairCtx c = make_ctx(a, it.sub, it.qidx);
	ChainState st = it.seg_in_q == 0 ? initial_state() : cold_state(it.start, (uint32_t)c.E.n);
	LWin w;
	w.q0 = EMPTY, w.dg = NO_DIAG;
	const size_t slot = it.slot;
	auto marks = [&]() { return a.marks + slot * ANDI_COLD_MARKS; }; // (formed where it is used: not held through the loop)
	uint32_t anchors = 0;
	while (st.p < it.end) {
		bool found;
#ifdef ANDI_LANE_STATS
		{ // trips of the wavefront's loop, and those with at most 32 / 16 / 8 lanes still at work
			const uint64_t on = __ballot(1);
			if (__lane_id() == (uint32_t)__builtin_ctzll(on)) {
				const int k = __builtin_popcountll(on);
				STAT(ST_X0);
				if (k <= 32) STAT(ST_X1);
				if (k <= 16) STAT(ST_X2);
				if (k <= 8) STAT(ST_X3);
			}
		}
#endif
		st = lane_step<EXACT>(c, st, tally, w, found);
		if (found && ++anchors == 1) *(uint4 *)marks()[0].first = make_uint4(st.lastQ, st.lastS, st.lastLen, 0);
		if (found && anchors >= 2 && anchors < 2 + ANDI_COLD_MARKS) { // remember the state after anchors 2, 3, 4
			ColdMark *m = marks() + (anchors - 2);
			ChainState ms = st;
			ms.pad[0] = 1;
			m->st = ms;
			uint32_t v[16];
#pragma unroll
			for (int t = 0; t < 16; ++t) v[t] = tally.hist[t * BLOCK];
			v[0] += tally.quarter + tally.same[0], v[5] += tally.quarter + tally.same[1];
			v[10] += tally.quarter + tally.same[2], v[15] += tally.quarter + tally.rest + tally.same[3];
			uint4 *mc = (uint4 *)m->counts;
#pragma unroll
			for (int t = 0; t < 4; ++t) mc[t] = make_uint4(v[4 * t], v[4 * t + 1], v[4 * t + 2], v[4 * t + 3]);
		}
	}
	for (uint32_t k = anchors < 2 ? 0 : anchors - 1; k < ANDI_COLD_MARKS; ++k) marks()[k].st.pad[0] = 0; // unused marks

	st.pad[1] = anchors < 255 ? anchors : 255;
	a.cold_exit[slot] = st;
	a.exit_p[slot] = st.p;
	tally_finish<1>(tally);
	uint32_t out[16];
#pragma unroll
	for (int t = 0; t < 16; ++t) out[t] = tally.hist[t * BLOCK];
	uint4 *dst = (uint4 *)(a.cold_counts + slot * 16);
#pragma unroll
	for (int t = 0; t < 4; ++t) dst[t] = make_uint4(out[4 * t], out[4 * t + 1], out[4 * t + 2], out[4 * t + 3]);
}

// ------------------------------------------------------------------ pass B
#ifdef ANDI_LANE_STATS
__device__ unsigned long long g_stitch_hist[4][16], g_stitch_steps[4], g_stitch_exit[4][8][2], g_stitch_phase[4][8];
__device__ unsigned int g_stitch_max[4];
#endif
// as stitch_segment in scan.hip: the true chain (entering in state T) is replayed next
// to the segment's cold chain until both are in the same state
// LISTED: the segments an earlier launch put on the list
// One segment, entered by the true chain in state T: what it counts there (owned), the state it leaves in (E, and
// true_exit) and the entry that was used (used_entry: pass C verifies it against the predecessor's true exit).
// KIND: 0 first stage, 1 first stage's list, 3 a stretch stitched again (statistics; 1 and 3 have no budget).
// false: the replay is too long for this launch and was put on the list.
template <bool EXACT, int KIND>
__device__ __forceinline__ bool stitch_one(const ScanArgs &a, const LaneItem &it, const PairCtx &c, ChainState T,
										   uint32_t (*s_hist)[16 * BLOCK], ChainState &E) {
	constexpr bool LISTED = KIND != 0;
	const size_t slot = it.slot;
	uint32_t steps = 0; // chain steps replayed so far
#ifdef ANDI_LANE_STATS
	int exit_kind = 0; // 0 left for the list, 1 shortcut, 2 first anchor, 3 mark, 4 met, 5 position, 6 on its own
	struct StepRec { // (diagnostic builds: how long the replays of each kind of launch are)
		const uint32_t &s;
		const int &xk;
		__device__ ~StepRec() {
			const int b = s ? 32 - __builtin_clz(s) : 0;
			atomicAdd(&g_stitch_exit[KIND][xk][0], 1ull);
			atomicAdd(&g_stitch_exit[KIND][xk][1], (unsigned long long)s);
			atomicAdd(&g_stitch_hist[KIND][b < 15 ? b : 15], 1ull);
			atomicAdd(&g_stitch_steps[KIND], (unsigned long long)s);
			atomicMax(&g_stitch_max[KIND], s);
		}
	} step_rec{steps, exit_kind};
#endif
	bool early = false;
	auto over_budget = [&]() { // (not LISTED) too long for this launch: leave it to the listed one
		if (++steps <= ANDI_STITCH_FIRST || LISTED) return false;
		// Past ANDI_STITCH_FIRST steps a replay leaves early if its wavefront is nearly empty by now -- but only if the
		// list is going to be long anyway: a launch over a short list costs its longest replay (0.1 ms on the bench
		// set, where the few stragglers ride along with the other wavefronts for nothing), a long one frees the
		// wavefronts that one or two replays would hold for the whole budget (the realistic set: 11.8 -> 9.4 ms).
		if (steps == ANDI_STITCH_FIRST + 1) early = atomicAdd(&a.restitch_count[ANDI_STRAGGLERS], 1u) >= ANDI_STITCH_MANY;
		if (steps <= ANDI_STITCH_BUDGET && !(early && __builtin_popcountll(__ballot(1)) <= ANDI_STITCH_FEW)) return false;
		a.defer_list[atomicAdd(a.defer_count, 1u)] = (unsigned long long)slot;
		return true;
	};
	const uint32_t *coldCounts = a.cold_counts + slot * 16;
	uint32_t *owned = a.owned + slot * 16;
	a.used_entry[slot] = T; // verified in pass C
	ChainState C = cold_state(it.start, (uint32_t)c.E.n);
	Tally tT, tC;
	tally_begin<1>(tT, s_hist[0] + threadIdx.x);
	tally_begin<1>(tC, s_hist[1] + threadIdx.x);
	LWin w; // shared by both chains: they run next to each other
	w.q0 = EMPTY, w.dg = NO_DIAG;
	bool found;

	// pass A's marks: the states the cold chain was in right after its first anchors
	const ColdMark *marks = a.marks + slot * ANDI_COLD_MARKS;
	ChainState M[ANDI_COLD_MARKS];
	uint32_t lastMarkP = 0;
	bool anyMark = false;
#pragma unroll
	for (int k = 0; k < ANDI_COLD_MARKS; ++k) {
		M[k] = marks[k].st;
		if (M[k].pad[0]) anyMark = true, lastMarkP = M[k].p;
	}
	// Shortcut: the true chain enters right behind an anchor that ends where the cold chain's
	// 1st anchor ends (the same match, found from further left -- the usual case when an
	// anchor crosses the segment's start).  Both chains then take the same steps up to and
	// including the cold chain's 2nd anchor, after which their states are equal; they differ
	// only in what that step counts for the anchor before it (src/process.c:176-186), and
	// for the models that split an anchor's length evenly that is known without replaying.
	if constexpr (!EXACT) {
		if (M[0].pad[0]) {
			const uint4 f1 = *(const uint4 *)marks[0].first; // pos_Q, pos_S, length
			if (T.p == f1.x + f1.z + 1 && T.lastQ + T.lastLen == f1.x + f1.z && T.lastS + T.lastLen == f1.y + f1.z &&
				T.p < it.end) {
				const bool right = M[0].lwra != 0; // the 2nd anchor was a right anchor (same test for both chains)
				const uint32_t lenC = (right || f1.z >= 2 * c.thr) ? f1.z : 0u;
				const uint32_t lenT = (right || T.lwra || T.lastLen >= 2 * c.thr) ? T.lastLen : 0u;
				const uint32_t dq = (lenT >> 2) - (lenC >> 2), dr = (lenT & 3u) - (lenC & 3u); // model_count_equal's split
				for (int t = 0; t < 16; ++t) {
					uint32_t v = coldCounts[t];
					if (t == 0 || t == 5 || t == 10) v += dq;
					if (t == 15) v += dq + dr;
					owned[t] = v;
				}
				a.true_exit[slot] = E = a.cold_exit[slot];
				STAT(ST_X0);
#ifdef ANDI_LANE_STATS
				exit_kind = 1;
#endif
				return true;
			}
		}
	}
	// The replay is ONE loop with one place where a chain steps (lane_step is most of this kernel's code, and the lanes
	// of a wavefront are in different phases of their replays: with a loop per phase a trip of the wavefront went through
	// up to five copies of it, one after the other).  Its phases:
	//
	// Phase 0: the cold chain's first anchor lies far ahead -- the segment starts in a stretch without homology
	// (an island, an unrelated contig).  Until a chain finds an anchor, and once lucky_anchor's precondition is
	// gone, its positions depend on its position alone: both chains step (the one behind) until they stand at the
	// same position; from there the true chain's steps are the cold chain's, so it arrives at the cold chain's
	// first anchor just as that did and is put there, with its own memory, instead of being replayed through
	// the stretch.
	//
	// Phase 1: only the true chain runs, until it is in a state the cold chain was in right after one of its first
	// anchors (pass A's marks).  The state right behind the cold chain's FIRST anchor is a mark too, and one that costs
	// nothing to keep: the cold chain has counted nothing by then (its first anchor is no right anchor and has no anchor
	// before it to count), and it leaves with last_was_right_anchor clear.  A true chain that finds that anchor from the
	// same position -- in a stretch without homology the two chains fall in with each other's positions after a few
	// steps, and chance anchors come every few hundred nucleotides -- and is no right anchor of what it remembers is in
	// that state: it need not walk on to the cold chain's second anchor.
	//
	// Phase 2 (no mark was hit): both chains, the one that is behind steps, until they meet.
	// Without an anchor ahead of it a chain's positions depend on its position alone (once lucky_anchor's
	// precondition is gone): when the two chains stand at the same position and the cold chain has found all
	// the anchors it finds in this segment, the true chain's remaining steps are the cold chain's -- whatever
	// the two remember.  That settles the segments of a stretch without homology after a few steps.
	// Chains that have not met after ANDI_STITCH_TOGETHER steps are in a stretch in which they will not soon
	// (without homology the step lengths hardly vary and the two keep leapfrogging; chance anchors of one are
	// not the other's): the true chain then runs alone to the segment's end -- half the steps of replaying both.
	uint32_t foundC = 0; // anchors the replayed cold chain has behind it
	const ChainState coldExit = a.cold_exit[slot];
	const uint32_t totalC = coldExit.pad[1], nS = (uint32_t)c.E.n;
	const bool knownC = totalC >= 1 && totalC != ANDI_ANCHORS_UNKNOWN;
	uint4 f1 = make_uint4(0, 0, 0, 0); // the cold chain's 1st anchor: pos_Q, pos_S, length
	if (knownC) f1 = *(const uint4 *)marks[0].first;
	ChainState MF = initial_state();
	bool hasF = false;
	if (knownC) {
		MF.p = f1.x + f1.z + 1, MF.lastS = f1.y, MF.lastQ = f1.x, MF.lastLen = f1.z;
		hasF = f1.x + f1.z + 1 < it.end; // (an anchor that leaves the segment is the cold exit: nothing to gain)
		if (hasF && MF.p > lastMarkP) lastMarkP = MF.p;
	}
	enum { PH0, PH1, PH2_ENTER, PH2, PH2_ALONE };
	// (a true chain that comes from an anchor nearby falls in with the marks at once: no phase 0)
	uint32_t phase = (knownC && f1.x > T.p + 4 * WNT && !lucky_applies(T, nS, c.thr)) ? PH0 : PH1;
	int hit = -1;
	bool synced = false, psynced = false;
	uint32_t together = 0;
	for (;;) {
		// (phase 0 begins whenever its condition comes to hold: a true chain that enters right behind an anchor -- the usual
		// case -- is past lucky_anchor's precondition after a step or two; left in phase 1 it walked all the way to the
		// cold chain's first anchor on its own, 55 steps on average for 106 K of the realistic set's listed replays)
		if (phase == PH1 && knownC && foundC == 0 && f1.x > T.p + 4 * WNT && C.p < f1.x && !lucky_applies(T, nS, c.thr)) phase = PH0;
		if (phase == PH0) {
			if (!(T.p < f1.x && C.p < f1.x)) {
				phase = PH1;
			} else if (T.p == C.p && !lucky_applies(T, nS, c.thr)) {
				T.p = f1.x;
#ifdef ANDI_LANE_STATS
				atomicAdd(&g_stitch_phase[KIND][6], 1ull);
#endif
				lane_account<EXACT>(c, T, tT, w, f1.y);
				T.lastS = f1.y, T.lastQ = f1.x, T.lastLen = f1.z;
				T.p += f1.z + 1;
				phase = PH1;
			}
		}
		if (phase == PH1) {
			if (anyMark || hasF) {
#pragma unroll
				for (int k = 0; k < ANDI_COLD_MARKS; ++k)
					if (hit < 0 && M[k].pad[0] && same_state(T, M[k])) hit = k;
				if (hit < 0 && hasF && same_state(T, MF)) hit = ANDI_COLD_MARKS;
				if (hit >= 0) break;
				if (T.p >= it.end || T.p > lastMarkP) phase = PH2_ENTER;
			} else {
				phase = PH2_ENTER;
			}
		}
		if (phase == PH2_ENTER) {
			STAT(ST_SEARCH); // (diagnostic builds: segments that reach phase 2)
			if (M[0].pad[0] && T.p >= M[0].p) { // the cold chain need not be replayed up to its mark
				C = M[0];
				foundC = 2;
				for (int t = 0; t < 16; ++t) tC.hist[t * BLOCK] = marks[0].counts[t];
			}
			phase = PH2;
		}
		if (phase == PH2) {
			if (same_state(T, C)) {
				synced = true;
				break;
			}
			if (together >= ANDI_STITCH_TOGETHER) {
				phase = PH2_ALONE;
			} else {
				if (T.p == C.p && totalC < 255 && foundC == totalC && !lucky_applies(T, nS, c.thr) && !lucky_applies(C, nS, c.thr)) {
					psynced = true;
					break;
				}
				if (T.p >= it.end) break;
			}
		}
		if (phase == PH2_ALONE && T.p >= it.end) break;
		if (over_budget()) return false;
		// who steps: the one that is behind (phases 0 and 2) -- but in phase 2 a cold chain with no anchor left to find is
		// only needed to learn where the true chain falls in with it, and that cannot happen while the true chain runs on
		// lucky anchors (through a repeat, where the cold chain would crawl from probe to probe)
		const bool stepT = phase == PH0 ? T.p <= C.p
						 : phase == PH2 ? (C.p >= it.end || T.p <= C.p || (totalC < 255 && foundC == totalC && lucky_applies(T, nS, c.thr)))
						 : true;
		Tally tx = stepT ? tT : tC;
#ifdef ANDI_LANE_STATS
		atomicAdd(&g_stitch_phase[KIND][phase == PH0 ? (stepT ? 0 : 1) : phase == PH1 ? 2 : phase == PH2 ? (stepT ? 3 : 4) : 5], 1ull);
#endif
		ChainState nx = lane_step<EXACT>(c, stepT ? T : C, tx, w, found);
		if (stepT) {
			T = nx, tT = tx;
		} else {
			C = nx, tC = tx;
			if (found) ++foundC;
		}
		if (phase == PH0 && found) phase = PH1; // the true chain found an anchor of its own, or the cold chain its first: go on as usual
		if (phase == PH2) ++together;
	}
	if (hit >= 0) {
		tally_finish<1>(tT);
#ifdef ANDI_LANE_STATS
		if (hit == ANDI_COLD_MARKS) STAT(ST_X1); else STAT(ST_X2);
		exit_kind = hit == ANDI_COLD_MARKS ? 2 : 3;
		atomicAdd(&g_lane_stats[ST_X7], (unsigned long long)steps);
#endif
		if (hit == ANDI_COLD_MARKS) {
			for (int t = 0; t < 16; ++t) owned[t] = tT.hist[t * BLOCK] + coldCounts[t];
		} else {
			const uint32_t *markCounts = marks[hit].counts;
			for (int t = 0; t < 16; ++t) owned[t] = tT.hist[t * BLOCK] + coldCounts[t] - markCounts[t];
		}
		a.true_exit[slot] = E = a.cold_exit[slot];
		return true;
	}
#ifdef ANDI_LANE_STATS
	if (synced) STAT(ST_X3); else if (psynced) STAT(ST_X4); else STAT(ST_X5);
	exit_kind = synced ? 4 : psynced ? 5 : 6;
	atomicAdd(&g_lane_stats[(synced || psynced) ? ST_X7 : ST_X6], (unsigned long long)steps);
#endif
	tally_finish<1>(tT);
	tally_finish<1>(tC);
	// from the meeting point on, the cold chain's trajectory is the true one
	for (int t = 0; t < 16; ++t) {
		uint32_t v = tT.hist[t * BLOCK];
		if (synced || psynced) v += coldCounts[t] - tC.hist[t * BLOCK];
		owned[t] = v;
	}
	if (psynced) T.p = coldExit.p; // same steps from here on, the true chain's own memory
	a.true_exit[slot] = E = synced ? coldExit : T;
	if (!synced && !psynced) atomicAdd(&a.restitch_count[ANDI_RESTITCH_ROUNDS], 1u); // its successor's assumed entry is at stake
	return true;
}

// first stage: every segment, entered in the state pass B assumes for it (scan_dev.h: assumed_entry)
template <bool EXACT, bool LISTED>
__device__ __forceinline__ void stitch_item(const ScanArgs &a, const LaneItem &it, uint32_t (*s_hist)[16 * BLOCK]) {
	const size_t slot = it.slot;
	if (it.seg_in_q == 0) { // the first segment's "cold" chain is the true chain
		const uint32_t *coldCounts = a.cold_counts + slot * 16;
		uint32_t *owned = a.owned + slot * 16;
		a.true_exit[slot] = a.cold_exit[slot];
		for (int t = 0; t < 16; ++t) owned[t] = coldCounts[t];
		return;
	}
	const PairCtx c = make_ctx(a, it.sub, it.qidx);
	ChainState E;
	(void)stitch_one<EXACT, LISTED ? 1 : 0>(a, it, c, assumed_entry(a, slot - it.seg_in_q, it.seg_in_q, it.seg, c.qlen), s_hist, E);
}

// Stitching again.  A segment is BAD if the state it was entered in (used_entry) is not the one the true chain left
// its predecessor in (true_exit) -- pass C's check.  Bad segments come in stretches: the successor of a segment whose
// true chain left on its own, stitched again, usually leaves in yet another state, and so on until the chain falls in
// with a cold chain again (a repeat the true chain crosses on lucky anchors, src/process.c:95-97; the edge of an
// island).  k_stitch_heads lists the first segment of every stretch; one lane per stretch then stitches it again,
// segment after segment, each entered in the state the one before was just left in, until a segment is entered in
// the state that was used for it before.  It stops short of the next stretch's first segment (another lane's), whose
// entry it may have changed: the next round finds that, or pass C does.  (Rounds that stitched every bad segment with
// its predecessor's exit as it stood -- one more segment of every stretch per round -- took 3 x 0.9 ms on the realistic
// set and left 970 segments to pass C.)
__global__ __launch_bounds__(BLOCK) void k_stitch_heads(ScanArgs a) {
	if (a.restitch_count[a.restitch_round > 0 ? a.restitch_round - 1 : ANDI_RESTITCH_ROUNDS] == 0) return; // (nothing left on its own / nothing was stitched again: nothing to do)
	if (!a.adaptive && a.subjects[blockIdx.y].mode != ANDI_MODE_PROBE) return;
	const LaneItem it = lane_item(a);
	// (consecutive lanes have consecutive segments in both layouts: the predecessor's answer comes from the lane before,
	// except for a wavefront's first lane)
	const bool bad = it.valid && it.seg_in_q >= 1 && !same_state(a.true_exit[it.slot - 1], a.used_entry[it.slot]);
	bool pred_bad = __shfl_up((int)bad, 1) != 0;
	if (!it.valid) return;
	if (__lane_id() == 0) pred_bad = it.seg_in_q >= 2 && !same_state(a.true_exit[it.slot - 2], a.used_entry[it.slot - 1]);
	if (it.seg_in_q < 2) pred_bad = false;
	a.stretch_bad[it.slot] = bad ? 1 : 0;
	if (bad && !pred_bad) {
		a.defer_list[atomicAdd(a.defer_count, 1u)] = (unsigned long long)it.slot;
		atomicAdd(&a.restitch_count[a.restitch_round], 1u);
	}
}

template <bool EXACT>
__device__ __forceinline__ void stitch_stretch(const ScanArgs &a, LaneItem it, uint32_t (*s_hist)[16 * BLOCK]) {
	const PairCtx c = make_ctx(a, it.sub, it.qidx);
	// the segment before `it` is settled: the true chain leaves it in state E; was_bad: it was bad when the round began
	ChainState E = a.true_exit[it.slot - 1];
	bool was_bad = true; // (so that the stretch's own first segment is not taken for another's)
	for (;;) {
		const bool bad = a.stretch_bad[it.slot] != 0;
		if (bad && !was_bad) return; // the next stretch's first segment: another lane's
		if (same_state(E, a.used_entry[it.slot])) {
			// the chain enters this segment as was assumed (now): what pass B recorded for it holds.  If it was not bad
			// before either, the stretch ends here (a bad segment further on is another stretch's first); if it was, the
			// segments behind it are still this lane's: no other looks at them in this round
			if (!bad) return;
			E = a.true_exit[it.slot];
		} else {
			// (A segment that was not bad and lies right before another stretch's first one is left alone: that stretch's
			// lane has read this segment's true exit as its entry state, perhaps at this moment -- rewriting it under
			// its eyes could hand it a torn state.  The next round finds the segment.)
			if (!bad && it.end < c.qlen && a.stretch_bad[it.slot + 1] != 0) return;
			E.pad[0] = E.pad[1] = E.pad[2] = 0;
			const ChainState T = E;
			(void)stitch_one<EXACT, 3>(a, it, c, T, s_hist, E);
		}
		if (it.end >= c.qlen) return; // the query's last segment
		was_bad = bad;
		it.slot += 1, it.seg_in_q += 1, it.start += it.seg;
		it.end = it.start + it.seg < c.qlen ? it.start + it.seg : c.qlen;
	}
}

// STAGE 0: every segment; 1: the segments stage 0 put on the list; 2: the stretches k_stitch_heads listed
template <bool EXACT, int STAGE>
__global__ __launch_bounds__(BLOCK, 4) void k_lane_stitch(ScanArgs a) { // (4 wavefronts per SIMD with 30 spilled registers: 1.33 -> 1.25 ms for passes B/C; 5 with 175: 1.78 ms)
	__shared__ uint32_t s_hist[2][16 * BLOCK];
	if constexpr (STAGE != 0) {
		// As few segments per wavefront as the list's length allows (at least ANDI_LISTED_LANES): their replays
		// are long and all different, a full wavefront of them would take its 64 lanes' paths one after the
		// other at every step, and the device is nearly empty while this launch runs
		const uint32_t count = *a.defer_count, waves = gridDim.x * WAVES_PER_BLOCK;
		uint32_t lanes = (count + waves - 1) / waves;
		lanes = lanes < ANDI_LISTED_LANES ? ANDI_LISTED_LANES : lanes > 64 ? 64 : lanes;
		if ((threadIdx.x & 63u) >= lanes) return;
		const uint32_t stride = waves * lanes;
		for (uint32_t idx = (blockIdx.x * WAVES_PER_BLOCK + (threadIdx.x >> 6)) * lanes + (threadIdx.x & 63u); idx < count; idx += stride) {
			const LaneItem it = lane_item_of_slot(a, a.defer_list[idx]);
			if constexpr (STAGE == 1) stitch_item<EXACT, true>(a, it, s_hist); else stitch_stretch<EXACT>(a, it, s_hist);
		}
	} else {
		if (!a.adaptive && a.subjects[blockIdx.y].mode != ANDI_MODE_PROBE) return;
		uint32_t seg_w = ~0u;
		if (!a.adaptive && a.stitch_lanes) { // (few segments: some lanes of every wavefront, scan.h)
			if ((threadIdx.x & 63u) >= a.stitch_lanes) return;
			seg_w = (blockIdx.x * WAVES_PER_BLOCK + (threadIdx.x >> 6)) * a.stitch_lanes + (threadIdx.x & 63u);
		}
		const LaneItem it = lane_item(a, ~0u, seg_w);
		if (it.valid) stitch_item<EXACT, false>(a, it, s_hist);
	}
}

// ------------------------------------------------------------------ packing
__device__ __forceinline__ uint32_t symbol_of(uint8_t ch) {
	return ch >= 'A' ? nt_code(ch) : (ch == '!' ? 4u : ch == ';' ? 5u : ch == '#' ? 6u : 7u);
}

__device__ __forceinline__ bool in_alphabet(uint8_t ch) {
	return ch == 'A' || ch == 'C' || ch == 'G' || ch == 'T' || ch == '!' || ch == ';' || ch == '#' || ch == 0;
}

// word j of N0 = symbols 8j .. 8j+7, word j of N1 = symbols 8j-1 .. 8j+6.  One thread
// packs 16 bytes (two words of each copy) through a 256-entry table in LDS.
__device__ __forceinline__ void pack_symbols_block(const uint8_t *__restrict__ src, int64_t pairs, uint2 *__restrict__ N0,
													uint2 *__restrict__ N1, int32_t *__restrict__ foreign) {
	__shared__ uint8_t lut[256]; // symbol | 0x80 if the byte is outside the alphabet
	lut[threadIdx.x] = (uint8_t)(symbol_of((uint8_t)threadIdx.x) | (in_alphabet((uint8_t)threadIdx.x) ? 0u : 0x80u));
	__syncthreads();
	const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (j >= pairs) return;
	const uint4 v = ld_u128_unaligned((g_u8p)src + 16 * j);
	const uint32_t in[4] = {v.x, v.y, v.z, v.w};
	uint32_t w[2] = {0, 0}, bad = 0;
#pragma unroll
	for (int k = 0; k < 16; ++k) {
		const uint32_t e = lut[(in[k >> 2] >> (8 * (k & 3))) & 0xffu];
		w[k >> 3] |= (e & 7u) << (4 * (k & 7));
		bad |= e;
	}
	if ((bad & 0x80u) && foreign) *foreign = 1; // a byte outside the alphabet: only the byte kernels are exact
	N0[j] = make_uint2(w[0], w[1]);
	if (N1) {
		const uint32_t prev = j ? (lut[src[16 * j - 1]] & 7u) : 7u;
		N1[j] = make_uint2((w[0] << 4) | prev, (w[1] << 4) | (w[0] >> 28));
	}
}

__global__ __launch_bounds__(256) void k_pack_symbols(const uint8_t *__restrict__ src, int64_t pairs,
													  uint2 *__restrict__ N0, uint2 *__restrict__ N1,
													  int32_t *__restrict__ foreign) {
	pack_symbols_block(src, pairs, N0, N1, foreign);
}

// the other way round: the bytes of a pool from its 4-bit symbols (the seam's queries come packed from the host; the
// rare byte-wise paths still read bytes).  One thread: 16 symbols -> 16 bytes.
__global__ __launch_bounds__(256) void k_unpack_symbols(const uint2 *__restrict__ N0, int64_t pairs, uint4 *__restrict__ dst) {
	const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (j >= pairs) return;
	const uint2 v = N0[j];
	const uint64_t letters = 0x00233b2154474341ull; // "ACGT!;#\0": byte k = the letter of symbol k
	uint32_t out[4];
#pragma unroll
	for (int k = 0; k < 4; ++k) {
		const uint32_t w = (k < 2 ? v.x : v.y) >> (16 * (k & 1)); // four symbols
		uint32_t o = 0;
#pragma unroll
		for (int t = 0; t < 4; ++t) o |= (uint32_t)((letters >> (8 * ((w >> (4 * t)) & 7u))) & 0xffu) << (8 * t);
		out[k] = o;
	}
	dst[j] = make_uint4(out[0], out[1], out[2], out[3]);
}

// 32 symbols (16 bytes of 4-bit symbols) -> bit 0, 1, 2 of each as three words (EsaDev.P)
__device__ __forceinline__ void planes_of(const uint4 v, uint32_t *out) {
	auto squeeze = [](uint32_t x) { // bit 4k + 3 of a word -> bit k
		x = (x >> 3) & 0x11111111u;
		x = (x | (x >> 3)) & 0x03030303u;
		x = (x | (x >> 6)) & 0x000f000fu;
		return (x | (x >> 12)) & 0xffu;
	};
#pragma unroll
	for (int b = 0; b < 3; ++b) {
		const int sh = 3 - b;
		out[b] = squeeze(v.x << sh) | (squeeze(v.y << sh) << 8) | (squeeze(v.z << sh) << 16) | (squeeze(v.w << sh) << 24);
	}
}

__global__ __launch_bounds__(256) void k_pack_planes(const uint4 *__restrict__ N0, int64_t blocks, uint32_t *__restrict__ planes) {
	const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (j >= blocks) return;
	uint32_t o[3];
	planes_of(N0[j], o);
	planes[3 * j] = o[0], planes[3 * j + 1] = o[1], planes[3 * j + 2] = o[2];
}

__global__ __launch_bounds__(256) void k_pack_planes_batch(const AndiIndexBatchItem *__restrict__ items) {
	const AndiIndexBatchItem it = items[blockIdx.y];
	if (!it.P || !it.N0) return;
	const int64_t blocks = ((int64_t)it.n + 1 + 64 + 31) / 32, j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (j >= blocks) return;
	uint32_t o[3];
	planes_of(((const uint4 *)it.N0)[j], o);
	it.P[3 * j] = o[0], it.P[3 * j + 1] = o[1], it.P[3 * j + 2] = o[2];
}

__global__ __launch_bounds__(256) void k_pack_symbols_batch(const AndiIndexBatchItem *__restrict__ items) {
	const AndiIndexBatchItem it = items[blockIdx.y];
	pack_symbols_block(it.S, ((int64_t)it.n + 1 + 64 + 15) / 16, (uint2 *)it.N0, (uint2 *)it.N1, it.flags + 1);
}

} // namespace

#ifdef ANDI_QUAD_TU
// This file compiled a second time with one wavefront per block (-DANDI_QUAD_TU -DWAVES_PER_BLOCK=1) for k_lane_quad
// alone: `count` blocks, block i = wavefront i of the list.  A block of four wavefronts holds its place until the
// slowest of them is done, and the chains of this kernel differ much in length (C3-like set 3.0 -> 2.8 ms, genomes
// 1e-5 apart 1.83 -> 1.55 ms).
hipError_t andi_launch_lane_quad_small(const ScanArgs &a0, uint32_t count, hipStream_t st) { // per-pair segments only
	if (count == 0) return hipSuccess;
	ScanArgs a = a0;
	a.quad_listed = 1;
	if (a.exact_equal)
		k_lane_quad<true><<<count, BLOCK, 0, st>>>(a);
	else
		k_lane_quad<false><<<count, BLOCK, 0, st>>>(a);
	return hipGetLastError();
}
#else

hipError_t andi_launch_pack_symbols_batch(const AndiIndexBatchItem *d_items, uint32_t count, size_t bytes, hipStream_t st) {
	const int64_t pairs = (int64_t)((bytes + 15) / 16);
	if (pairs == 0 || count == 0) return hipSuccess;
	k_pack_symbols_batch<<<dim3((unsigned)((pairs + 255) / 256), count), 256, 0, st>>>(d_items);
	CHECK_LAUNCH();
	return hipSuccess;
}

hipError_t andi_launch_unpack_symbols(const uint8_t *N0, size_t bytes, uint8_t *dst, hipStream_t st) { // bytes: of dst, a multiple of 16
	const int64_t pairs = (int64_t)(bytes / 16);
	if (pairs == 0) return hipSuccess;
	k_unpack_symbols<<<(unsigned)((pairs + 255) / 256), 256, 0, st>>>((const uint2 *)N0, pairs, (uint4 *)dst);
	CHECK_LAUNCH();
	return hipSuccess;
}

hipError_t andi_launch_pack_planes(const uint8_t *N0, size_t symbols, uint32_t *planes, hipStream_t st) {
	const int64_t blocks = (int64_t)((symbols + 31) / 32);
	if (blocks == 0) return hipSuccess;
	k_pack_planes<<<(unsigned)((blocks + 255) / 256), 256, 0, st>>>((const uint4 *)N0, blocks, planes);
	CHECK_LAUNCH();
	return hipSuccess;
}

hipError_t andi_launch_pack_planes_batch(const AndiIndexBatchItem *d_items, uint32_t count, size_t max_n, hipStream_t st) {
	const int64_t blocks = (int64_t)((max_n + 1 + 64 + 31) / 32);
	if (blocks == 0 || count == 0) return hipSuccess;
	k_pack_planes_batch<<<dim3((unsigned)((blocks + 255) / 256), count), 256, 0, st>>>(d_items);
	CHECK_LAUNCH();
	return hipSuccess;
}

hipError_t andi_launch_pack_symbols(const uint8_t *src, size_t bytes, uint8_t *N0, uint8_t *N1,
									int32_t *foreign, hipStream_t st) {
	const int64_t pairs = (int64_t)((bytes + 15) / 16);
	if (pairs == 0) return hipSuccess;
	k_pack_symbols<<<(unsigned)((pairs + 255) / 256), 256, 0, st>>>(src, pairs, (uint2 *)N0, (uint2 *)N1, foreign);
	CHECK_LAUNCH();
	return hipSuccess;
}

static int lane_occupancy(bool per_pair) { // waves per SIMD pass A is compiled for (experiments: ANDI_LANE_OCC)
	const char *e = andi_knob(KNOB_LANE_OCC);
	int v = e ? atoi(e) : 0;
	if (v == 6 || v == 7 || v == 8) return v;
	// per-pair segments: 8 (64 registers, two spilled outside the loop): 6.29 ms against 6.41 at 7 and 6.6 at 6 on the
	// bench set.  One segment length (the query's base and length are per lane): seven registers would be spilled
	// at 8 -- 7.69 against 6.90 ms at 7 (bench set, 4096-symbol segments), 2.67 against 2.50 (2000 x 16.5 kbp)
	return per_pair ? 8 : 7;
}

template <bool EXACT>
static hipError_t lane_cold(const ScanArgs &a, dim3 grid, hipStream_t st) {
	const bool blocks4 = andi_knob(KNOB_QUAD_UNLISTED) != nullptr; // (experiments: k_lane_quad's wavefronts in the call's order)
	const bool quads = a.adaptive && a.quad_min_match != 0xffffffffu;
	const bool side = quads && a.side_stream && !andi_knob(KNOB_NO_SIDE_STREAM);
	// With the side stream: k_lane_cold goes first, the host then reads how many wavefronts k_lane_quad's list has
	// (four bytes over the side stream, while k_lane_cold runs: the device is never idle -- a wait for the layout
	// BEFORE pass A cost it 0.1 ms) and launches k_lane_quad with exactly that many single-wavefront blocks, or
	// not at all: blocks that only find out that they have nothing to do are not free either (bench set + 0.1 ms).
	const bool counted = side && a.h_quad_waves && !blocks4 && !andi_knob(KNOB_QUAD_BLOCKS4);
	if (quads) { // the pairs with long matches, beside the others
		(void)hipMemsetAsync(a.first_pub, 0xff, (size_t)64 * a.max_waves * sizeof(unsigned long long), st);
		if (side) {
			(void)hipEventRecord(a.side_fork, st);
			(void)hipStreamWaitEvent(a.side_stream, a.side_fork, 0);
		}
		if (!counted) { // blocks of four wavefronts over the whole grid: those beyond the list return at once
			ScanArgs b = a;
			b.quad_listed = blocks4 ? 0 : 1;
			k_lane_quad<EXACT><<<grid, BLOCK, 0, side ? a.side_stream : st>>>(b);
			if (side) (void)hipEventRecord(a.side_join, a.side_stream);
		}
	}
	if (a.adaptive) {
		switch (lane_occupancy(true)) {
			case 6: k_lane_cold<EXACT, 6, true><<<grid, BLOCK, 0, st>>>(a); break;
			case 7: k_lane_cold<EXACT, 7, true><<<grid, BLOCK, 0, st>>>(a); break;
			default: k_lane_cold<EXACT, 8, true><<<grid, BLOCK, 0, st>>>(a); break;
		}
	} else {
		switch (lane_occupancy(false)) {
			case 6: k_lane_cold<EXACT, 6, false><<<grid, BLOCK, 0, st>>>(a); break;
			case 7: k_lane_cold<EXACT, 7, false><<<grid, BLOCK, 0, st>>>(a); break;
			default: k_lane_cold<EXACT, 8, false><<<grid, BLOCK, 0, st>>>(a); break;
		}
	}
	if (counted) {
		(void)hipMemcpyAsync(a.h_quad_waves, a.restitch_count + ANDI_QUAD_WAVES, sizeof(uint32_t), hipMemcpyDeviceToHost, a.side_stream);
		hipError_t e = hipStreamSynchronize(a.side_stream);
		if (e != hipSuccess) return e;
		e = andi_launch_lane_quad_small(a, *a.h_quad_waves, a.side_stream);
		if (e != hipSuccess) return e;
		(void)hipEventRecord(a.side_join, a.side_stream);
	}
	if (side) (void)hipStreamWaitEvent(st, a.side_join, 0);
	return hipGetLastError();
}

static hipError_t pair_offsets(const ScanArgs &a, hipStream_t st) {
	const uint32_t P = a.nsub * a.nq;
	const unsigned nb = (P + 1023) / 1024;
	k_pair_block_sums<<<nb, 1024, 0, st>>>(a);
	CHECK_LAUNCH();
	k_pair_block_offsets<<<1, 1024, 0, st>>>(a);
	CHECK_LAUNCH();
	k_pair_offsets<<<nb, 1024, 0, st>>>(a);
	CHECK_LAUNCH();
	return hipSuccess;
}

hipError_t andi_launch_pair_layout(const ScanArgs &a, hipStream_t st) {
	const uint32_t P = a.nsub * a.nq;
	(void)hipMemsetAsync(a.restitch_count, 0, 16 * sizeof(uint32_t), st); // k_lane_quad's list is empty, no wavefronts counted
	if (a.sub_cost) (void)hipMemsetAsync(a.sub_cost, 0, a.nsub * sizeof(float), st);
	if (P <= 4096) k_pair_estimate<<<P, P <= 1024 ? 64 * EST_WAVES_FEW : 64 * EST_WAVES, 0, st>>>(a, false);
	else k_pair_estimate<<<(P + 3) / 4, 256, 0, st>>>(a, true);
	CHECK_LAUNCH();
	if (a.route) {
		k_pair_totals<<<(P + 1023) / 1024, 1024, 0, st>>>(a);
		CHECK_LAUNCH();
		k_pair_route<<<(P + 255) / 256, 256, 0, st>>>(a);
		CHECK_LAUNCH();
		if (a.sub_order) {
			k_sub_order<<<1, 1024, 0, st>>>(a);
			CHECK_LAUNCH();
		}
	}
	return a.adaptive ? pair_offsets(a, st) : hipSuccess; // (a routed call with one segment length for its lanes: the marks only)
}

hipError_t andi_launch_pair_leftover(const ScanArgs &a2, hipStream_t st) {
	const uint32_t P = a2.nsub * a2.nq;
	(void)hipMemsetAsync(a2.restitch_count + ANDI_QUAD_WAVES, 0, 3 * sizeof(uint32_t), st);
	k_pair_leftover<<<(P + 255) / 256, 256, 0, st>>>(a2);
	CHECK_LAUNCH();
	return a2.adaptive ? pair_offsets(a2, st) : hipSuccess;
}

hipError_t andi_launch_route_count(const ScanArgs &a, hipStream_t st) {
	const uint32_t P = a.nsub * a.nq;
	k_route_count<<<(P + 255) / 256, 256, 0, st>>>(a);
	CHECK_LAUNCH();
	return hipSuccess;
}

static dim3 lane_grid(const ScanArgs &a) {
	if (a.adaptive) return dim3((a.max_waves + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK, 1);
	return dim3((a.total_segs + BLOCK - 1) / BLOCK, a.nsub);
}

hipError_t andi_launch_lane_cold(const ScanArgs &a, hipStream_t st) {
	dim3 grid = lane_grid(a);
	hipError_t e = a.exact_equal ? lane_cold<true>(a, grid, st) : lane_cold<false>(a, grid, st);
#ifdef ANDI_LANE_STATS
	if (e == hipSuccess && andi_knob(KNOB_LANE_STATS)) {
		static const char *names[24] = {"steps", "lcp_reload", "lcp_slide", "probes", "probe_reload", "table", "final_sa",
										"single", "ext_loop", "multi", "multi_cand", "search", "gap_reload", "gap_words",
										"substitutions", "lucky_tries", "x0", "x1", "x2", "x3", "x4", "x5", "x6", "x7"};
		unsigned long long h[24];
		(void)hipStreamSynchronize(st);
		(void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_lane_stats), sizeof h);
		for (int k = 0; k < 24; ++k) fprintf(stderr, "lane_stats %-14s %llu\n", names[k], h[k]);
		memset(h, 0, sizeof h);
		(void)hipMemcpyToSymbol(HIP_SYMBOL(g_lane_stats), h, sizeof h);
	}
#endif
	return e;
}

hipError_t andi_launch_lane_stitch(const ScanArgs &a0, hipStream_t st) {
	hipStream_t st_ = st;
	(void)st_;
	dim3 grid = lane_grid(a0);
	ScanArgs a = a0;
	hipError_t e = hipMemsetAsync(a.restitch_count, 0, 16 * sizeof(uint32_t), st);
	if (e != hipSuccess) return e;
	// every stage: the launch over all segments, then the segments it put on the list (a grid for half of all
	// segments at most: blocks beyond the list's length return at once; the list cannot be longer than the
	// segments that have a predecessor)
	const size_t slots = a.adaptive ? (size_t)64 * a.max_waves : (size_t)a.nsub * a.total_segs;
	const uint32_t per_block = WAVES_PER_BLOCK * ANDI_LISTED_LANES;
	const unsigned lblocks = (unsigned)std::min<size_t>((slots + per_block - 1) / per_block, ANDI_LISTED_BLOCKS); // (strides over the list)
	// (every stage counts its list in a word of its own among those the memset above has just cleared -- 8, and 13 ... 15 for the
	// rounds, which only the layout and pass A use otherwise: no memset between the stages, four launches less per layout)
	static_assert(ANDI_RESTITCH_ROUNDS <= 3, "a list counter per round: restitch_count[13 ... 15]");
	auto stage = [&](auto main_kernel, auto listed_kernel, dim3 main_grid, uint32_t counter) {
		a.defer_count = a.restitch_count + counter;
		main_kernel<<<main_grid, BLOCK, 0, st>>>(a);
		listed_kernel<<<lblocks, BLOCK, 0, st>>>(a);
	};
	// the first launch over few segments of one length: some lanes of every wavefront (scan.h: stitch_lanes), 8192 wavefronts or so
	dim3 grid0 = grid;
	a.stitch_lanes = 0;
	if (!a.adaptive && slots < (size_t)64 * 8192) {
		uint32_t L = 1;
		while ((size_t)L * 8192 < slots) L *= 2;
		if (L < 64) a.stitch_lanes = L, grid0 = dim3((a.total_segs + WAVES_PER_BLOCK * L - 1) / (WAVES_PER_BLOCK * L), a.nsub);
	}
	if (a.exact_equal)
		stage(k_lane_stitch<true, 0>, k_lane_stitch<true, 1>, grid0, 8);
	else
		stage(k_lane_stitch<false, 0>, k_lane_stitch<false, 1>, grid0, 8);
#ifdef ANDI_LANE_STATS
	if (andi_knob(KNOB_LANE_STATS)) { // pass B's first stage: how its segments were settled, and the chain steps that took
		static const char *names[8] = {"entered behind the cold chain's first anchor", "met behind its first anchor", "met at a mark",
									   "met in phase 2", "position-synced", "left on their own", "steps of those that left on their own", "steps of the others"};
		unsigned long long h[24];
		(void)hipStreamSynchronize(st);
		(void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_lane_stats), sizeof h);
		for (int k = 0; k < 8; ++k) fprintf(stderr, "stitch_stats %-48s %llu\n", names[k], h[ST_X0 + k]);
		fprintf(stderr, "stitch_stats %-48s %llu\n", "reached phase 2", h[ST_SEARCH]);
		memset(h, 0, sizeof h);
		(void)hipMemcpyToSymbol(HIP_SYMBOL(g_lane_stats), h, sizeof h);
	}
#endif
	const bool again = !andi_knob(KNOB_NO_RESTITCH);
	for (uint32_t r = 0; again && r < ANDI_RESTITCH_ROUNDS; ++r) {
		a.restitch_round = r;
		if (a.exact_equal)
			stage(k_stitch_heads, k_lane_stitch<true, 2>, grid, 13 + r);
		else
			stage(k_stitch_heads, k_lane_stitch<false, 2>, grid, 13 + r);
	}
#ifdef ANDI_LANE_STATS
	if (andi_knob(KNOB_LANE_STATS)) { // replays by length (steps: 0, 1, 2-3, 4-7, ...) per kind of launch
		static const char *kinds[4] = {"first stage", "first stage, listed", "-", "stretches stitched again (segments)"};
		unsigned long long h[4][16], st[4];
		unsigned int mx[4];
		(void)hipStreamSynchronize(st_);
		(void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_stitch_hist), sizeof h);
		(void)hipMemcpyFromSymbol(st, HIP_SYMBOL(g_stitch_steps), sizeof st);
		(void)hipMemcpyFromSymbol(mx, HIP_SYMBOL(g_stitch_max), sizeof mx);
		{
			uint32_t rc[16];
			(void)hipMemcpy(rc, a.restitch_count, sizeof rc, hipMemcpyDeviceToHost);
			fprintf(stderr, "stitch_rounds stretches listed per round: %u %u %u; left on their own: %u\n", rc[0], rc[1], rc[2], rc[ANDI_RESTITCH_ROUNDS]);
		}
		{
			static const char *xn[8] = {"left for the list", "shortcut", "first anchor", "mark", "met", "position", "on its own", "-"};
			unsigned long long x[4][8][2];
			(void)hipMemcpyFromSymbol(x, HIP_SYMBOL(g_stitch_exit), sizeof x);
			for (int k = 0; k < 4; ++k)
				for (int e = 0; e < 7; ++e)
					if (x[k][e][0]) fprintf(stderr, "stitch_exit kind %d %-18s %9llu replays %10llu steps\n", k, xn[e], x[k][e][0], x[k][e][1]);
			memset(x, 0, sizeof x);
			(void)hipMemcpyToSymbol(HIP_SYMBOL(g_stitch_exit), x, sizeof x);
			unsigned long long ph[4][8];
			(void)hipMemcpyFromSymbol(ph, HIP_SYMBOL(g_stitch_phase), sizeof ph);
			for (int k = 0; k < 4; ++k)
				fprintf(stderr, "stitch_phase kind %d steps: phase 0 true %llu cold %llu, phase 1 %llu, phase 2 true %llu cold %llu, alone %llu; put at the first anchor %llu\n", k,
						ph[k][0], ph[k][1], ph[k][2], ph[k][3], ph[k][4], ph[k][5], ph[k][6]);
			memset(ph, 0, sizeof ph);
			(void)hipMemcpyToSymbol(HIP_SYMBOL(g_stitch_phase), ph, sizeof ph);
		}
		for (int k = 0; k < 4; ++k) {
			fprintf(stderr, "stitch_len %-22s steps %llu max %u  by log2:", kinds[k], st[k], mx[k]);
			for (int b = 0; b < 16; ++b) fprintf(stderr, " %llu", h[k][b]);
			fprintf(stderr, "\n");
		}
		memset(h, 0, sizeof h), memset(st, 0, sizeof st), memset(mx, 0, sizeof mx);
		(void)hipMemcpyToSymbol(HIP_SYMBOL(g_stitch_hist), h, sizeof h);
		(void)hipMemcpyToSymbol(HIP_SYMBOL(g_stitch_steps), st, sizeof st);
		(void)hipMemcpyToSymbol(HIP_SYMBOL(g_stitch_max), mx, sizeof mx);
	}
#endif
	return hipGetLastError();
}
#endif // ANDI_QUAD_TU
