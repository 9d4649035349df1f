// scan.h — launch interface of the anchor-scan kernels (scan.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "andi_dev.h"
#include "andi_hip.h"

// Loop-top state of dist_anchor's while loop (src/process.c:153): the query
// position about to be examined plus everything the next iterations read.
struct __attribute__((aligned(16))) ChainState {
	uint32_t p;       // this_match.pos_Q
	uint32_t lastS;   // last_match.pos_S
	uint32_t lastQ;   // last_match.pos_Q
	uint32_t lastLen; // last_match.length
	uint32_t lwra;    // last_was_right_anchor
	uint32_t pad[3];  // [0]: a mark is valid (ColdMark); [1] in a cold exit: anchors the cold chain found in its
					  // segment (capped at 255; ANDI_ANCHORS_UNKNOWN: not recorded)
};
#define ANDI_ANCHORS_UNKNOWN 0xffffffffu

// What pass A of the lane scan remembers of a cold chain besides its exit: the state and
// the counts right after its 2nd (.. ANDI_COLD_MARKS + 1 th) anchor.  The true chain
// usually falls in step with the cold chain there; pass B then need not replay the cold
// chain.  (1, 2 and 3 marks measured: the first one does nearly all of it.)
#define ANDI_COLD_MARKS 1
struct __attribute__((aligned(16))) ColdMark {
	ChainState st; // st.pad[0] != 0: valid
	uint32_t counts[16];
	uint32_t first[4]; // the cold chain's 1st anchor: pos_Q, pos_S, length (mark 0 only)
};

struct ScanArgs {
	const EsaDev *subjects; // [nsub] device array
	const int64_t *self;    // [nsub] query index of the subject itself or -1
	uint32_t nsub;
	// query pool
	const uint8_t *qpool;
	const uint8_t *qnib;  // the pool as 4-bit symbols: sequence q starts at qnib + qoff[q] / 2
	const uint32_t *qplanes; // the pool bit-sliced (andi_dev.h: EsaDev.P): sequence q starts at block qoff[q] / 32
	const uint64_t *qoff; // [nq]
	const uint32_t *qlen; // [nq]
	const uint32_t *qsep; // [nq] contig separators ('!') of every query, or null (k_sep_counts; the routing keeps pairs whose diagonals break often away from the wavefront kernels)
	uint32_t nq;
	// segmentation of the queries into work items
	const uint32_t *qseg_start; // [nq+1]
	const uint32_t *seg2query;  // [total_segs]
	uint32_t total_segs;
	uint32_t seg; // nucleotides per segment
	// per (subject, segment) scratch
	ChainState *cold_exit; // state when the cold chain leaves the segment
	uint32_t *exit_p;      // its position alone, densely (entry_source reads eight of them per segment)
	uint32_t *cold_counts; // [..][16] counts the cold chain added inside the segment
	ColdMark *marks;       // [..][ANDI_COLD_MARKS] (lane scan only)
	ChainState *true_exit; // state of the true chain when it leaves the segment
	ChainState *used_entry; // the state pass B let the true chain enter the segment in (pass C verifies it)
	uint32_t *restitch_count; // [r]: stretches stitched again in round r (lane scan); [ANDI_RESTITCH_ROUNDS]: true chains that left their segment on their own
	uint32_t restitch_round;
	// pass B's first launch over a layout of few segments (one segment length: the wavefront kernel's): stitch_lanes < 64 lanes of
	// every wavefront take a segment each -- 31 000 replays of up to 48 dependent steps in 490 full wavefronts took 154 us of the bench
	// step, every step of a wavefront paying for the paths of all its lanes (0: all 64)
	uint32_t stitch_lanes;
	// Segments whose stitching takes more than ANDI_STITCH_BUDGET chain steps -- or more than ANDI_STITCH_FIRST
	// when no more than ANDI_STITCH_FEW lanes of their wavefront are still at it and the call has had more than
	// ANDI_STITCH_MANY such replays -- are put on a list and stitched
	// by a second launch, few of them per wavefront: one long replay (through a repeat, along the edge of an
	// island) would otherwise hold up the 63 settled segments of its wavefront.
	unsigned long long *defer_list; // slots
	uint32_t *defer_count;
	uint32_t defer_base; // first list entry of this launch
	// k_lane_quad: where a cold chain's first anchor starts at its segment's start, (pos_S << 32 | length), published for
	// the chain of the segment before it -- in another wavefront -- whose match may run on into it (all ones: none yet)
	unsigned long long *first_pub;
	uint8_t *stretch_bad; // pass B, stitching again (k_stitch_heads): the segment's entry was not its predecessor's true exit when the round began (lies in first_pub's memory, idle by then)
	uint32_t *owned;       // [..][16] counts the true chain adds inside the segment
	andi_hip_model *M;     // [nsub][nq]
	unsigned long long *fixups;
	int exact_equal;   // LogDet/ANI: count the nucleotides of every anchor (src/model.c:256-278)
	int any_reference; // some subject is in ANDI_MODE_REFERENCE: launch the reference-walk kernels too
	// Per-pair segment lengths (lane scan, all subjects on the probe table).  A pair is
	// subject * nq + query; its segments are seg0 << pair_class long and occupy the slots
	// 64 * pair_wave0[pair] ..., i.e. whole wavefronts; pair_wave0[nsub * nq] = wavefronts in use.
	int adaptive;
	uint32_t seg0;
	uint32_t seg_factor; // a pair's segments are at least this many mean match lengths long
	uint32_t max_class;  // longest class in use (<= 3): kept small enough that the call still has ~2^19 chains
	uint8_t *pair_class;
	uint32_t *pair_waves; // scratch: wavefronts per pair
	uint32_t *pair_wave0;
	uint32_t *pair_bsum;  // scratch: sums / offsets of 1024 pairs each
	uint32_t max_waves;   // upper bound (every pair in class 0): the grid
	// Routed calls: pass A by wavefronts takes the subjects heaviest first (sub_order[k]: the subject the k-th row of its grid works
	// on; null: as they come).  A wavefront's segment takes a millisecond and a launch is three or four rounds of them: what the
	// device does while the last ones finish depends on which they are (the bench set's 29 subjects in order of rising
	// divergence 4.82 ms, as they come 4.50, heaviest first 4.39).  sub_cost: what k_pair_estimate's samples say a subject's pairs
	// cost -- nucleotides over mean match length, summed.
	float *sub_cost;
	uint32_t *sub_order;
	// Pairs whose sampled mean match length is at least quad_min_match (per-pair segment lengths only) take pass A with
	// the streams fetched by quads of lanes (scan_lane.hip: k_lane_quad), the others lane_step's; 0xffffffff: none do
	uint32_t quad_min_match;
	// the wavefronts of those pairs, listed by k_pair_offsets (in pass B's list, idle during pass A; their number
	// at restitch_count[ANDI_QUAD_WAVES]): k_lane_quad's wavefronts take them in order
	uint32_t quad_listed; // k_lane_quad takes its wavefronts from that list
	uint32_t *h_quad_waves; // pinned host word the list's length is copied to (host side only)
	// host side only: a second stream and two events, so that pass A's two kernels (k_lane_quad for the pairs with long
	// matches, k_lane_cold for the others) share the device instead of each ending in a tail of its own
	hipStream_t side_stream;
	hipEvent_t side_fork, side_join;
	// Pass A of a call is ROUTED PER PAIR (route != 0; the engine's choice from 2^25 query symbols x subjects; api.hip).  k_pair_estimate samples every
	// pair; k_pair_route marks the pairs that suit pass A by wavefronts (scan_coop.hip) -- matches neither long (k_lane_quad's
	// class) nor hardly reaching the anchor threshold, unless such pairs are few; no runs of short matches (unrelated
	// stretches) -- with ANDI_ROUTE_COOP in their pair_class byte: they get no wavefronts in the lane layout, whose kernels
	// run beside that kernel.  A wavefront of k_coop_cold that meets what it is slow at all the same (a stretch without
	// homology, a match longer than its segment) marks its pair ANDI_ROUTE_LEFT and returns, as do the pair's other
	// wavefronts when they see the mark; such pairs (rare) get a lane layout of their own afterwards (ANDI_ROUTE_L2).
	// Passes B and C run once per layout, each on its own pairs.
	int route; // 0: no routing (every pair takes the call's one pass A); else the layout these arguments describe: ANDI_LAYOUT_*
	uint32_t route_seg;           // the wavefront kernel's segment length in a routed call
	uint32_t reduce_threads;      // pass C: threads of a pair's block (64 where no query has more than 64 segments, else the lane scan's block)
	uint32_t route_all_few;       // small calls: where the lane scan's pairs would be few, the wavefront kernel takes every pair (k_pair_route)
	uint32_t route_soft_match;    // mean sampled match from which a pair is better off with k_lane_quad if such pairs are many (k_pair_estimate)
	uint32_t route_giveup;        // generic steps in one of its segments beyond which a wavefront hands its pair back (scan_coop.hip: COOP_TRIAL_G; ANDI_COOP_GIVEUP: tests)
	unsigned long long *route_nt; // [3]: query nucleotides of the pairs whose pass A ran by wavefronts / by lanes, pairs handed back
	int coop;            // pass A with one wavefront per chain (scan_coop.hip): one segment length, RAW/JC/Kimura, probe-table subjects
	// ... with the windows' walks pooled through global memory (coop_pool.h: k_pool_cold; the models that split an anchor's length
	// evenly): a scratch per resident wavefront, pool_waves of them; pool_ticket: the next segment to take (zeroed per launch)
	void *pool_scratch;   // the wavefronts' scratches
	uint32_t *pool_ticket;
	uint32_t pool_waves;
	size_t pool_bytes;    // host side: what the context holds behind pool_ticket's 4096 bytes
	uint32_t pool_maxchunks, pool_hc; // a scratch's size: rounds of 2048 positions of a window, heads of a window
	// routed calls: WHICH wavefront kernel -- k_pool_cold where the pairs whose sampled mean match is pool_match ... 4095 (few heads per position:
	// streaming decides) hold at least half of the wavefront kernel's segments, k_coop_cold otherwise (many heads: its windows in LDS walk them at
	// less cost; near-identical genomes: a window sees one mismatch).  One kernel per call: side by side they were slower than either.
	uint32_t pool_match;
	int pool_use; // host side, routed calls: this call's wavefront kernel is k_pool_cold (the pairs with long sampled matches hold most of its wavefronts)
	uint32_t pool_first; // rounds of 2048 positions of a chain's first window (doubled after every window the chain got through)
	uint32_t knock;      // diagnostic builds (-DANDI_LANE_STATS): parts of pass A switched off to time them (results are then wrong)
};

// 4-bit symbols of `bytes` source bytes (a NUL-padded pool or text) into N0 and, if not
// null, the one-symbol-shifted copy N1; `bytes` is rounded up to a multiple of 16; *foreign is
// set to 1 if a byte is none of A C G T ! ; # NUL (the packed scan is then not applicable)
hipError_t andi_launch_unpack_symbols(const uint8_t *N0, size_t bytes, uint8_t *dst, hipStream_t st); // a byte pool from its 4-bit symbols
hipError_t andi_launch_pack_symbols(const uint8_t *src, size_t bytes, uint8_t *N0, uint8_t *N1,
									int32_t *foreign, hipStream_t st);
// the bit-sliced form of 4-bit symbols (EsaDev.P): `symbols` of them from N0 (rounded up to blocks of 32) into planes
hipError_t andi_launch_pack_planes(const uint8_t *N0, size_t symbols, uint32_t *planes, hipStream_t st);
// adaptive mode: sample every pair's match lengths, choose its segment length (and, in a routed call, its pass A), lay out the slots
hipError_t andi_launch_pair_layout(const ScanArgs &a, hipStream_t st);
// routed calls: the second lane layout (a2: the pairs pass A by wavefronts handed back); who took what, for the timings
hipError_t andi_launch_pair_leftover(const ScanArgs &a2, hipStream_t st);
hipError_t andi_launch_route_count(const ScanArgs &a, hipStream_t st);
hipError_t andi_launch_lane_cold(const ScanArgs &a, hipStream_t st);
// pass A with one wavefront per chain (scan_coop.hip)
int andi_coop_enabled(void); // 0: off (ANDI_COOP=0); < 0: the engine chooses -- tiny calls every pair, others routed per pair (the default); n = 2, 4, 8: every pair, windows of 2048 n symbols
hipError_t andi_launch_coop_cold(const ScanArgs &a, hipStream_t st);
int andi_coop_wants_pool(const ScanArgs &a); // a launch of pass A by wavefronts that k_pool_cold would take if the context had its scratch
int andi_coop_will_pool(const ScanArgs &a); // that launch is k_pool_cold's (coop_pool.h), not k_coop_cold's
size_t andi_pool_scratch_bytes(int device, uint32_t *waves); // the pooled kernels' scratch (4096 bytes for the ticket in front); 0: those kernels are off (ANDI_POOL=0)
hipError_t andi_launch_lane_stitch(const ScanArgs &a, hipStream_t st);
hipError_t andi_launch_lane_quad_small(const ScanArgs &a, uint32_t count, hipStream_t st); // k_lane_quad, one wavefront per block (scan_lane.hip compiled a second time)
// Pass B again for the segments that were entered in a state their predecessor's true chain did not leave in (a true
// chain that runs on lucky anchors through a repeat in which cold chains find nothing unique): per round, k_stitch_heads
// lists the first segment of every stretch of such segments and one lane per stretch stitches it to its end
// (scan_lane.hip).  One round settles all but the stretches that ran into each other; rounds after one that listed
// nothing return at once.
#ifndef ANDI_RESTITCH_ROUNDS
#define ANDI_RESTITCH_ROUNDS 3 /* (realistic set: 18 163 stretches in round 1, 86 in round 2, none in round 3) */
#endif
#ifndef ANDI_STITCH_FIRST
#define ANDI_STITCH_FIRST 12
#endif
#ifndef ANDI_STITCH_FEW
#define ANDI_STITCH_FEW 12
#endif
#ifndef ANDI_STITCH_MANY
#define ANDI_STITCH_MANY 4096
#endif
static_assert(ANDI_RESTITCH_ROUNDS < 8, "restitch_count[]: rounds 0 .. ROUNDS at the front, defer_count at 8, ANDI_ROUTE_ANY_LEFT at 11, ANDI_STRAGGLERS at 12, ANDI_QUAD_WAVES at 13, ANDI_SPARSE_WAVES at 14, ANDI_ALL_WAVES at 15");
#define ANDI_ROUTE_COOP 0x40u /* pair_class: the pair takes pass A by wavefronts (routed calls) */
#define ANDI_ROUTE_LEFT 0x20u /* ... which handed it back */
#define ANDI_ROUTE_SOFT 0x10u /* (k_pair_estimate to k_pair_route) the lane scan is better at it, if such pairs are more than a few */
#define ANDI_ROUTE_GUESS 0x04u /* (k_pair_estimate to k_pair_route, small calls) marked for the wavefront kernel although the sampling cannot judge the pair: not where the call has pairs with unrelated stretches */
/* ONE BIT, TWO MEANINGS, kept apart in time (the class byte has no bit to spare): ANDI_ROUTE_POOLCAND is written by k_pair_estimate and
 * read AND CLEARED for every pair by k_pair_route (scan_lane.hip: route_pair) before anything reads ANDI_ROUTE_L2, which only k_pair_leftover
 * sets, behind pass A.  A reader of ANDI_ROUTE_L2 placed between k_pair_estimate and k_pair_route -- or a path that skips k_pair_route --
 * would take pool candidates for handed-back pairs and drop them from the lane layout: do not add one. */
#define ANDI_ROUTE_L2 0x08u   /* (from k_pair_leftover on) a pair handed back: in the second lane layout */
#define ANDI_ROUTE_POOLCAND 0x08u /* (k_pair_estimate to k_pair_route ONLY) mean sampled match in [ScanArgs.pool_match, 4096): a pair that suits k_pool_cold (coop_pool.h) */
#define ANDI_STRUCT_WAVES 4 /* restitch_count[this] during the layout of a routed call: wavefronts of pairs with unrelated stretches that are not merely far apart (k_pair_estimate) */
#define ANDI_POOL_SEGS 6 /* restitch_count[this] after k_pair_route: segments of the wavefront kernel's pairs that suit k_pool_cold */
#define ANDI_COOP_SEGS 7 /* ... of all its pairs */
#define ANDI_LAYOUT_LANES 1   /* ScanArgs.route: the lane scan's pairs */
#define ANDI_LAYOUT_LANES2 2  /* the pairs pass A by wavefronts handed back */
#define ANDI_LAYOUT_COOP 3    /* the wavefront kernel's pairs (one segment length) */
#define ANDI_ROUTE_ANY_LEFT 11 /* restitch_count[this] of the wavefront kernel's layout: some pair was handed back */
#define ANDI_ROUTE_MIN_QLEN 8192u /* queries shorter than this (and than a segment of the wavefront kernel) are the lane scan's: 1000 x 5 kbp 11.6 ms by wavefronts, 9.3 by lanes; 300 x 10 kbp 1.85 against 2.65 */
#define ANDI_HARD_WAVES 9 /* restitch_count[this] during the layout: wavefronts of pairs the wavefront kernel is no candidate for (beside ANDI_SPARSE_WAVES, which includes the soft ones) */
#define ANDI_ISLAND_WAVES 11 /* restitch_count[this] of the lane layout during the layout: wavefronts of pairs in which the sampling saw unrelated stretches (ANDI_ROUTE_LEFT is their mark until k_pair_route) */
#define ANDI_LANE_WAVES 10 /* restitch_count[this] after the layout of a routed call: wavefronts of the pairs the lane scan keeps (k_pair_route) */
#define ANDI_ALL_WAVES 15      /* restitch_count[this] during the layout: wavefronts of all pairs (beside ANDI_SPARSE_WAVES) */
#define ANDI_QUAD_WAVES 13 /* restitch_count[this] during pass A: wavefronts on k_lane_quad's list */
#define ANDI_SPARSE_WAVES 14 /* restitch_count[this] during pass A: wavefronts of pairs whose sampled mean match is below ANDI_SPARSE_MATCH */
#define ANDI_SPARSE_MATCH 19u
#define ANDI_STRAGGLERS 12 /* restitch_count[this]: replays of the call so far that went past ANDI_STITCH_FIRST steps */
#ifndef ANDI_STITCH_BUDGET
#define ANDI_STITCH_BUDGET 48
#endif
#ifndef ANDI_LISTED_LANES
#define ANDI_LISTED_LANES 2
#endif
#ifndef ANDI_LISTED_BLOCKS
#define ANDI_LISTED_BLOCKS 1024 /* one round of the device at 4 wavefronts per SIMD; 4096: passes B/C 15.5 ms on the realistic set, 1024: 13.3, 512: 13.7 */
#endif
#ifndef ANDI_STITCH_TOGETHER
#define ANDI_STITCH_TOGETHER 40 /* steps of pass B's phase 2 in which the cold chain is replayed beside the true one */
#endif
hipError_t andi_launch_scan_cold(const ScanArgs &a, hipStream_t st);
hipError_t andi_launch_scan_stitch(const ScanArgs &a, hipStream_t st);
hipError_t andi_launch_scan_reduce(const ScanArgs &a, hipStream_t st);
hipError_t andi_launch_match_positions(const EsaDev &E, const uint8_t *q, uint32_t qlen,
									   uint32_t first, uint32_t count, int cached,
									   andi_hip_interval *out, hipStream_t st);
