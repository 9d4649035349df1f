// knobs.h — the library's environment switches, read ONCE: when the library first looks at any of them, and again only
// on andi_hip_reload_knobs() (include/andi_hip.h).  No getenv on the scan / build path.
//
// The shipped library (libandihip.so) knows SEVEN of them: where its device memory comes from, which pass A kernels it may
// use, the reference's own walk for every subject, how rows are gathered, and a trace of the seam.  Everything else --
// experiment switches and test hooks: forced layouts, segment lengths, thresholds, kernels switched off -- exists only in
// the build the test suite and scripts/dev load (libandihip_test.so, -DANDI_TEST_HOOKS); in the shipped library
// andi_knob() of such a name is a constant nullptr and the code behind it folds away.
#pragma once

#define ANDI_KNOB_LIST_SHIPPED(X) X(ARENA_KEEP) X(ARENA_MB) X(COOP) X(E2E_TRACE) X(FORCE_REFERENCE) X(GATHER) X(POOL)
#define ANDI_KNOB_LIST_HOOKS(X)                                                                                            \
	X(COOP_GIVEUP) X(COOP_SEG) X(COOP_STATS) X(DEBUG_STITCH) X(DEEP_K) X(FORCE_ADAPTIVE) X(KNOCK) X(LANE_OCC) X(LANE_STATS)      \
	X(NO_RESTITCH) X(NO_SIDE_STREAM) X(NO_SORTED_RECORDS) X(POOL_FIRST) X(POOL_MATCH) X(QUAD_BLOCKS4) X(QUERIES_BYTES) X(QUERIES_PACKED)       \
	X(QUAD_MATCH) X(QUAD_UNLISTED) X(ROUTE_SMALL) X(ROUTE_SOFT) X(ROUTE_TINY) X(SEG0) X(SORT_WIDTH) X(SEG_FACTOR) X(SINGLE_EXT) X(UNIFORM_SEGMENTS) X(UPLOAD_MIN_MB)
#define ANDI_KNOB_LIST(X) ANDI_KNOB_LIST_SHIPPED(X) ANDI_KNOB_LIST_HOOKS(X)

enum AndiKnob {
#define X(n) KNOB_##n,
	ANDI_KNOB_LIST(X)
#undef X
	KNOB_COUNT
};
#define X(n) +1
constexpr int ANDI_KNOBS_SHIPPED = 0 ANDI_KNOB_LIST_SHIPPED(X); // (they come first)
#undef X

// the value of ANDI_<name> as it was when the knobs were read; nullptr if it was not set
const char *andi_knob_value(AndiKnob k);
inline const char *andi_knob(AndiKnob k) {
#ifndef ANDI_TEST_HOOKS
	if ((int)k >= ANDI_KNOBS_SHIPPED) return nullptr;
#endif
	return andi_knob_value(k);
}
