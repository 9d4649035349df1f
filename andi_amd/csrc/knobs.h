// knobs.h — the library's environment switches (experiments, diagnostics, test hooks), read ONCE: when the library first
// looks at any of them, and again only on andi_hip_reload_knobs() (include/andi_hip.h; the tests call it after changing
// the environment under a live context).  No getenv on the scan / build path.
#pragma once

#define ANDI_KNOB_LIST(X)                                                                                                  \
	X(ARENA_KEEP) X(ARENA_MB) X(COOP) X(COOP_GIVEUP) X(COOP_SEG) X(COOP_STATS) X(DEBUG_STITCH) X(DEEP_K) X(E2E_TRACE) X(FORCE_ADAPTIVE)  \
	X(FORCE_REFERENCE) X(GATHER) X(KNOCK) X(LANE_OCC) X(LANE_STATS) X(NO_RESTITCH) X(NO_SIDE_STREAM)      \
	X(NO_SORTED_RECORDS) X(POOL) X(POOL_FIRST) X(QUAD_BLOCKS4) X(QUERIES_BYTES) X(QUERIES_PACKED) X(QUAD_MATCH) X(QUAD_UNLISTED) X(ROUTE_SMALL) X(ROUTE_SOFT) X(ROUTE_TINY) X(SEG0) X(SEG_FACTOR) X(SINGLE_EXT) X(UNIFORM_SEGMENTS)

enum AndiKnob {
#define X(n) KNOB_##n,
	ANDI_KNOB_LIST(X)
#undef X
	KNOB_COUNT
};

// the value of ANDI_<name> as it was when the knobs were read; nullptr if it was not set
const char *andi_knob(AndiKnob k);
