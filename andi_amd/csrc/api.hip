// api.hip — the C-ABI of libandihip.so (include/andi_hip.h): device objects,
// staging, kernel orchestration and the one-call replacement of
// distMatrix/distMatrixLM (src/dist_hack.h:34-96).
#include <hip/hip_runtime.h>
#include <rccl/rccl.h> // types only: librccl is loaded on demand (dlopen), see rccl() below
#include <dlfcn.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <mutex>
#include <string>
#include <thread>
#include <sys/mman.h>
#include <vector>

#include "andi_dev.h"
#include "andi_hip.h"
#include "bootstrap.h"
#include "dev_arena.h"
#include "esa_build.h"
#include "knobs.h"
#include "sa_device.h"
#include "scan.h"

// ------------------------------------------------------------------ knobs (knobs.h)
namespace {
// The switches as they were when last read: an immutable snapshot behind an atomic pointer.  andi_hip_reload_knobs
// publishes a new one and never frees the old (a handful of them in a test session): a string andi_knob returned stays
// valid whatever other threads do.
struct KnobSnapshot {
	std::string value[KNOB_COUNT];
	bool set[KNOB_COUNT];
	KnobSnapshot() {
		static const char *names[KNOB_COUNT] = {
#define X(n) "ANDI_" #n,
			ANDI_KNOB_LIST(X)
#undef X
		};
		for (int k = 0; k < KNOB_COUNT; ++k) {
#ifndef ANDI_TEST_HOOKS
			if (k >= ANDI_KNOBS_SHIPPED) { // (the shipped library does not even look)
				set[k] = false;
				continue;
			}
#endif
			const char *v = getenv(names[k]);
			set[k] = v != nullptr;
			value[k] = v ? v : "";
		}
	}
};
std::atomic<const KnobSnapshot *> g_knobs{nullptr};
const KnobSnapshot &knob_store() {
	const KnobSnapshot *s = g_knobs.load(std::memory_order_acquire);
	if (!s) { // (read when the library first looks)
		const KnobSnapshot *fresh = new KnobSnapshot();
		if (g_knobs.compare_exchange_strong(s, fresh, std::memory_order_acq_rel))
			s = fresh;
		else
			delete fresh;
	}
	return *s;
}
} // namespace

const char *andi_knob_value(AndiKnob k) {
	const KnobSnapshot &s = knob_store();
	return s.set[k] ? s.value[k].c_str() : nullptr;
}


// segment length when the caller passes 0: short enough that one scan launch has
// several hundred thousand chains, long enough that stitching stays a few per cent
#define ANDI_MIN_SEGMENT 4096u
#define ANDI_MAX_SEGMENT 65536u
#define ANDI_TARGET_CHAINS (1u << 22) /* at most about this many chains per call (308 bytes of scratch each) */
#define ANDI_MIN_CHAINS (1u << 19)    /* and long segments only while the call keeps this many */
// per-pair segment lengths: classes seg/2, seg, 2 seg, 4 seg of the call's length, as long as the scratch they
// need (whole wavefronts per pair) stays a fraction of the device's memory
#define ANDI_ADAPTIVE_MAX_PAIRS (1u << 22)
#define ANDI_ROUTE_MIN_NT (1u << 18) /* query symbols x subjects from which pass A of a call is routed per pair */
#define ANDI_ROUTE_TINY_NT (1u << 25) /* ... below which it is not routed but takes pass A by wavefronts for every pair */
#define ANDI_ROUTE_SMALL_NT (1ull << 30) /* ... below which pass A by wavefronts takes a millisecond or less: a few pairs left to the lane scan would take longer (k_pair_route) */
// scratch per (subject, segment): three states, two count vectors, the marks, the exit position, a list slot, a published anchor
#define ANDI_SLOT_BYTES (3 * sizeof(ChainState) + 2 * 16 * sizeof(uint32_t) + ANDI_COLD_MARKS * sizeof(ColdMark) + 4 + 8 + 8)

static_assert(sizeof(andi_hip_model) == 68, "struct model must be 17 x u32 (src/model.h:52-57)");
static_assert(sizeof(andi_hip_interval) == 16, "lcp_inter_t is 4 x int32 (src/esa.h:25-34)");
static_assert(sizeof(ChainState) == 32, "ChainState is padded to 32 bytes");
static_assert(sizeof(ColdMark) == 112, "ColdMark is a state, 16 counts and the first anchor");

struct EventPair {
	hipEvent_t a, b;
	int kind; // 0 build, 1 scan, 2 stitch
};

struct andi_hip_ctx {
	int device = 0;
	size_t queries_hint = 0; // andi_hip_ctx_expect_queries
	hipStream_t stream = nullptr;
	hipStream_t side_stream = nullptr; // pass A's second kernel runs beside the first
	hipEvent_t side_fork = nullptr, side_join = nullptr;
	std::string err;
	// scan scratch
	void *scratch = nullptr;
	size_t scratch_bytes = 0;
	// descriptor staging (pinned host + device), guarded by desc_done
	void *desc_host = nullptr;
	void *desc_dev = nullptr;
	size_t desc_bytes = 0;
	hipEvent_t desc_done = nullptr;
	unsigned long long *d_fixups = nullptr;
	// batched index builds: items (pinned host + device), guarded by ib_done
	void *ib_host = nullptr, *ib_dev = nullptr;
	size_t ib_cap = 0;
	hipEvent_t ib_done = nullptr;
	// index builds queued since the last scan looked at their flags (pinned host words the build kernels write)
	hipEvent_t built = nullptr;
	bool builds_pending = false;
	// device suffix sorter: workspace, two pinned ints
	void *sa_ws = nullptr;
	size_t sa_ws_bytes = 0;
	int32_t *sa_pinned = nullptr;
	int stream_prio = 0; // of stream and side_stream (host_pool: they go back there)
	uint32_t *h_quad_waves = nullptr; // pinned: the length of k_lane_quad's list of a scan call
	hipStream_t coop_stream = nullptr; // routed scan calls: pass A by wavefronts runs beside the lane scan's kernels
	hipEvent_t coop_fork = nullptr, coop_join = nullptr, l2_fork = nullptr, l2_join = nullptr;
	uint32_t *h_any_left = nullptr;    // pinned: [0] the wavefront kernel handed some pair back, [1 + k] the layout's counter restitch_count[k] (ANDI_LANE_WAVES: wavefronts of the lane layout, ...)
	void *pool_scratch = nullptr;      // pass A by wavefronts with pooled walks (coop_pool.h): a scratch per resident wavefront
	size_t pool_bytes = 0;
	uint32_t pool_waves = 0;
	bool pool_failed = false;          // its allocation failed once: not tried again by this context
	void *scratch2 = nullptr;          // the second lane layout (those pairs), grown on demand
	size_t scratch2_bytes = 0;
	unsigned long long *d_route = nullptr; // routed scan calls: query nucleotides whose pass A ran by wavefronts / by lanes, pairs handed back (read with the timings)
	std::vector<EventPair> pending;
	andi_hip_timings acc{};
};

struct andi_hip_esa {
	uint8_t *S = nullptr;
	int32_t *SA = nullptr, *LCP = nullptr, *CLD = nullptr;
	uint8_t *FVC = nullptr;
	int4 *tab = nullptr;
	int32_t *min_scratch = nullptr;
	uint2 *deep = nullptr;
	uint8_t *Nraw = nullptr;              // 4-bit symbols for the lane scan: N0 and N1 with their padding
	uint8_t *N0 = nullptr, *N1 = nullptr;
	uint32_t *Praw = nullptr, *P = nullptr; // the text bit-sliced (EsaDev.P; packed from N0 when a scan call wants it), a block of padding in front
	uint32_t *rec = nullptr;    // the suffixes' records in suffix-array order, left by the device sorter (sa_device.hip) for the index build
	bool rec_valid = false;
	uint16_t *rec2 = nullptr;   // ... and the symbols behind their first deepK (same validity)
	int32_t *flags = nullptr;   // device, 4 ints
	int32_t *h_flags = nullptr; // the same 4 ints as the host sees them (flags live in pinned host memory)
	int32_t deepK = 0;
	int32_t deepK_cap = 0; // the depth the table was allocated for
	int32_t n = 0;
	int32_t thr = 0;
	size_t cap = 0;     // characters the buffers were sized for (>= n)
	size_t ref_cap = 0; // same for the reference arrays
	bool ref_built = false;   // LCP, CLD, FVC, tab valid
	bool index_built = false; // deep, flags valid
	int deep_ext = 0;         // the form of the table's entries of K-mers that occur once (andi_dev.h: 0 plain, 1 extended, 2 short extended)
	size_t bytes = 0;
};

struct andi_hip_queries {
	uint8_t *pool = nullptr;
	uint8_t *nib = nullptr;       // the pool as 4-bit symbols
	uint32_t *planes = nullptr;   // ... bit-sliced (EsaDev.P)
	int32_t *h_foreign = nullptr; // pinned: set if the pool holds bytes outside the alphabet
	uint64_t *d_off = nullptr;
	uint32_t *d_len = nullptr;
	uint32_t *d_sep = nullptr;    // contig separators of every sequence (k_sep_counts: for the routing of the scan)
	std::vector<uint64_t> off;
	std::vector<uint32_t> len;
	size_t nq = 0;
	uint64_t total_nt = 0;
	// segmentation cache
	uint32_t seg = 0;
	uint32_t *d_qseg_start = nullptr;
	uint32_t *d_seg2query = nullptr;
	uint32_t total_segs = 0;
	// a second one: the long segments of pass A by wavefronts (scan_coop.hip), kept beside the call's own so that a
	// call that falls back to the lane scan does not cut the queries anew every time
	uint32_t c_seg = 0;
	uint32_t *c_qseg_start = nullptr;
	uint32_t *c_seg2query = nullptr;
	uint32_t c_total_segs = 0;
};

namespace {

void set_err(char *buf, size_t len, const char *fmt, ...) {
	if (!buf || !len) return;
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(buf, len, fmt, ap);
	va_end(ap);
}

int fail(andi_hip_ctx *ctx, const char *what, hipError_t e) {
	if (ctx) {
		ctx->err = std::string(what) + ": " + hipGetErrorString(e);
	}
	return 1;
}

#define HIP_TRY(ctx, call)                                                                         \
	do {                                                                                           \
		hipError_t e__ = (call);                                                                   \
		if (e__ != hipSuccess) return fail((ctx), #call, e__);                                     \
	} while (0)

template <typename T>
hipError_t dmalloc(T **p, size_t count) {
	return andi_arena::dev_malloc((void **)p, count * sizeof(T)); // (out of large chunks: dev_arena.h)
}

void resolve_events(andi_hip_ctx *ctx) {
	for (auto &ev : ctx->pending) {
		float ms = 0.f;
		if (hipEventSynchronize(ev.b) == hipSuccess && hipEventElapsedTime(&ms, ev.a, ev.b) == hipSuccess) {
			if (ev.kind == 0) {
				ctx->acc.build_ms += ms;
				ctx->acc.build_launches++;
			} else if (ev.kind == 1) {
				ctx->acc.scan_ms += ms;
				ctx->acc.scan_launches++;
			} else {
				ctx->acc.stitch_ms += ms;
				ctx->acc.stitch_launches++;
			}
		}
		(void)hipEventDestroy(ev.a);
		(void)hipEventDestroy(ev.b);
	}
	ctx->pending.clear();
}

struct Timed {
	andi_hip_ctx *ctx;
	EventPair ev;
	bool ok;
	Timed(andi_hip_ctx *c, int kind) : ctx(c), ok(false) {
		ev.kind = kind;
		if (hipEventCreate(&ev.a) != hipSuccess) return;
		if (hipEventCreate(&ev.b) != hipSuccess) {
			(void)hipEventDestroy(ev.a);
			return;
		}
		ok = hipEventRecord(ev.a, c->stream) == hipSuccess;
	}
	void stop() {
		if (!ok) return;
		(void)hipEventRecord(ev.b, ctx->stream);
		ctx->pending.push_back(ev);
		ok = false;
		if (ctx->pending.size() > 256) resolve_events(ctx);
	}
	~Timed() { // an error exit before stop(): the events go with the timer
		if (!ok) return;
		(void)hipEventDestroy(ev.a);
		(void)hipEventDestroy(ev.b);
	}
	Timed(const Timed &) = delete;
	Timed &operator=(const Timed &) = delete;
};

EsaDev esa_view(const andi_hip_esa *e, int mode) {
	EsaDev v;
	v.S = e->S, v.SA = e->SA, v.LCP = e->LCP, v.CLD = e->CLD, v.FVC = e->FVC, v.tab = e->tab;
	v.deep = e->deep, v.flags = e->flags;
	v.N0 = e->N0, v.N1 = e->N1, v.P = e->P;
	v.R2 = e->rec_valid ? e->rec2 : nullptr; // (made by the device sorter for this K: api.hip esa_sort_suffixes)
	v.n = e->n, v.thr = e->thr, v.deepK = e->deepK, v.mode = mode, v.deep_ext = e->deep_ext;
	return v;
}

// probe table depth: smallest K with 4^K >= n, within [4, 13]
// (queries: how many queries a subject of this context will meet, 0 = unknown.  A table one level deeper answers more
// probes without touching the text -- pass A of a C4-shaped call 53.2 -> 50.0 ms -- and costs its build four times the
// stores -- 0.06 -> 0.19 ms per 4.2 M-character subject: it pays from about a thousand queries per subject on.)
int pick_deep_k(size_t n, size_t queries) {
	int K = 4;
	while (K < ANDI_MAX_DEEP_K && ((size_t)1 << (2 * K)) < n) ++K;
	if (queries >= 1024 && K < ANDI_MAX_DEEP_K) ++K;
	if (const char *ev = andi_knob(KNOB_DEEP_K)) {
		int v = atoi(ev);
		if (v >= 4 && v <= ANDI_MAX_DEEP_K) K = v;
	}
	return K;
}

EsaBuildArgs build_args(const andi_hip_esa *e) {
	EsaBuildArgs a;
	a.S = e->S, a.SA = e->SA, a.LCP = e->LCP, a.CLD = e->CLD, a.FVC = e->FVC, a.tab = e->tab;
	a.deep = e->deep, a.flags = e->flags, a.deepK = e->deepK;
	a.rec = e->rec_valid ? e->rec : nullptr;
	a.rec2 = e->rec_valid ? e->rec2 : nullptr;
	a.N0 = e->N0, a.N1 = e->N1, a.P = e->P;
	a.min_scratch = e->min_scratch;
	a.n = e->n;
	return a;
}

} // namespace

extern "C" {

int andi_hip_abi_version(void) {
	return ANDI_HIP_ABI_VERSION;
}

// Streams and the seam's pinned upload buffer are kept from one call to the next (like the arena's chunks: andi_hip_trim
// gives them back): creating a stream takes 3 ms on this runtime -- nine per call of andi_hip_dist_matrix, 30 of a warm
// call's 105 ms -- and pinning 20 MB another 4.
namespace host_pool {
struct IdleStream {
	int device, prio;
	hipStream_t s;
};
struct IdlePinned {
	void *p;
	size_t bytes;
};
struct IdleScratch { // k_pool_cold's scratch (0.8 GB on a 256-CU part: a mapping of its own, not the arena's)
	int device;
	void *p;
	size_t bytes;
};
// The pool's state lives on the heap and is never destroyed (as dev_arena.h's arenas): a context may be released during or
// after the destruction of this library's statics, and its streams come back here.
struct State {
	std::mutex mu;
	std::vector<IdleStream> streams;
	std::vector<IdlePinned> pinned;
	std::vector<IdleScratch> scratch;
	// small pinned blocks (flags and counters the kernels write and the host reads: 16 ... 256 bytes each) out of slabs of
	// 64 KiB: a context and its subjects took two dozen hipHostMalloc / hipHostFree of a few words per call of the seam,
	// 5 of a warm call's 36 ms
	std::vector<char *> slabs;
	std::vector<void *> free_words;
	size_t words_out = 0;
};
static State &state() {
	static State *s = new State;
	return *s;
}
constexpr size_t PINNED_KEEP = (size_t)256 << 20; // bytes of pinned buffers kept at most

static hipError_t stream_get(hipStream_t *out, int device, int prio) {
	{
		std::lock_guard<std::mutex> lk(state().mu);
		for (size_t i = 0; i < state().streams.size(); ++i)
			if (state().streams[i].device == device && state().streams[i].prio == prio) {
				*out = state().streams[i].s;
				state().streams.erase(state().streams.begin() + (long)i);
				return hipSuccess;
			}
	}
	return hipStreamCreateWithPriority(out, hipStreamNonBlocking, prio);
}
static void stream_put(hipStream_t s, int device, int prio) { // (idle: the caller has synchronised it)
	if (!s) return;
	std::lock_guard<std::mutex> lk(state().mu);
	state().streams.push_back({device, prio, s});
}
static hipError_t pinned_get(void **out, size_t bytes) {
	{
		std::lock_guard<std::mutex> lk(state().mu);
		size_t best = state().pinned.size();
		for (size_t i = 0; i < state().pinned.size(); ++i)
			if (state().pinned[i].bytes >= bytes && state().pinned[i].bytes <= 2 * bytes + 4096 && (best == state().pinned.size() || state().pinned[i].bytes < state().pinned[best].bytes)) best = i;
		if (best != state().pinned.size()) {
			*out = state().pinned[best].p;
			state().pinned.erase(state().pinned.begin() + (long)best);
			return hipSuccess;
		}
	}
	return hipHostMalloc(out, bytes, hipHostMallocDefault);
}
static void pinned_put(void *p, size_t bytes) {
	if (!p) return;
	{
		std::lock_guard<std::mutex> lk(state().mu);
		size_t held = 0;
		for (const IdlePinned &b : state().pinned) held += b.bytes;
		if (held + bytes <= PINNED_KEEP) {
			state().pinned.push_back({p, bytes});
			return;
		}
	}
	(void)hipHostFree(p);
}
constexpr size_t WORD_BYTES = 256, SLAB_BYTES = 65536;
static void *word_get() { // 256 zeroed bytes of pinned host memory (device-visible: unified addressing), 256-byte aligned
	std::lock_guard<std::mutex> lk(state().mu);
	State &S = state();
	if (S.free_words.empty()) {
		char *slab = nullptr;
		if (hipHostMalloc((void **)&slab, SLAB_BYTES, hipHostMallocDefault) != hipSuccess) {
			(void)hipGetLastError();
			return nullptr;
		}
		S.slabs.push_back(slab);
		for (size_t o = SLAB_BYTES; o >= WORD_BYTES; o -= WORD_BYTES) S.free_words.push_back(slab + o - WORD_BYTES);
	}
	void *p = S.free_words.back();
	S.free_words.pop_back();
	++S.words_out;
	memset(p, 0, WORD_BYTES);
	return p;
}
static void word_put(void *p) {
	if (!p) return;
	std::lock_guard<std::mutex> lk(state().mu);
	state().free_words.push_back(p);
	--state().words_out;
}

// the pooled wavefront kernel's scratch: one per device is kept from context to context (a context of the seam lives for one
// call: 0.8 GB of hipMalloc + hipFree per call and device otherwise); andi_hip_trim returns it
static void *scratch_get(int device, size_t bytes) {
	{
		std::lock_guard<std::mutex> lk(state().mu);
		auto &v = state().scratch;
		for (size_t i = 0; i < v.size(); ++i)
			if (v[i].device == device && v[i].bytes == bytes) {
				void *p = v[i].p;
				v.erase(v.begin() + (long)i);
				return p;
			}
	}
	void *p = nullptr;
	if (hipMalloc(&p, bytes) != hipSuccess) {
		(void)hipGetLastError();
		return nullptr;
	}
	return p;
}
static void scratch_put(int device, void *p, size_t bytes) { // (idle: the caller has waited for the kernels that used it)
	if (!p) return;
	{
		std::lock_guard<std::mutex> lk(state().mu);
		auto &v = state().scratch;
		bool have = false;
		for (const IdleScratch &x : v) have = have || x.device == device;
		if (!have) {
			v.push_back({device, p, bytes});
			return;
		}
	}
	(void)hipFree(p);
}
static bool any() {
	std::lock_guard<std::mutex> lk(state().mu);
	return !state().streams.empty() || !state().pinned.empty() || !state().scratch.empty() || !state().slabs.empty();
}
static size_t trim() { // (the caller restores the current device); returns the device bytes given back
	std::vector<IdleStream> st;
	std::vector<IdlePinned> pb;
	std::vector<IdleScratch> sc;
	{
		std::lock_guard<std::mutex> lk(state().mu);
		st.swap(state().streams), pb.swap(state().pinned), sc.swap(state().scratch);
		if (state().words_out == 0) { // (slabs with blocks still out stay)
			for (char *slab : state().slabs) pb.push_back({slab, SLAB_BYTES});
			state().slabs.clear(), state().free_words.clear();
		}
	}
	for (const IdleStream &x : st)
		if (hipSetDevice(x.device) == hipSuccess) (void)hipStreamDestroy(x.s);
	for (const IdlePinned &b : pb) (void)hipHostFree(b.p);
	size_t freed = 0;
	for (const IdleScratch &x : sc)
		if (hipSetDevice(x.device) == hipSuccess && hipFree(x.p) == hipSuccess) freed += x.bytes;
	return freed;
}
} // namespace host_pool

size_t andi_hip_trim(void) {
	if (!andi_arena::any_chunks() && !host_pool::any()) return 0; // (a process that never used the library's device memory: no HIP call at all)
	int ndev = 0, cur = 0;
	if (hipGetDeviceCount(&ndev) != hipSuccess) {
		(void)hipGetLastError();
		return 0;
	}
	(void)hipGetDevice(&cur);
	size_t freed = 0;
	for (int d = 0; d < ndev && d < 64; ++d) {
		if (!andi_arena::has_chunks(d)) continue; // (a device the library never used is not touched)
		if (hipSetDevice(d) != hipSuccess) continue;
		freed += andi_arena::trim(d);
	}
	freed += host_pool::trim(); // (idle streams, pinned upload buffers, the pooled kernel's scratch)
	(void)hipSetDevice(cur);
	return freed;
}

void andi_hip_default_opts(andi_hip_opts *o) {
	if (!o) return;
	memset(o, 0, sizeof *o);
	o->p_value = 0.025; // ANCHOR_P_VALUE, src/andi.c:48
	o->model = ANDI_M_JC;
	o->device = 0;
	o->host_threads = 0;
	o->low_memory = 0;
	o->segment = 0;
	o->sa_on_host = 0;
	o->num_gpus = 1;
	o->devices = NULL;
}

int andi_hip_device_count(void) {
	int count = 0;
	return hipGetDeviceCount(&count) == hipSuccess && count > 0 ? count : 0;
}

// high_priority: the context's streams are served before those of other contexts on the device (the staging stage of
// andi_hip_dist_matrix: its short kernels must not queue behind the workgroups of a scan that fills the device)
static int ctx_create(andi_hip_ctx **out, int device, char *errbuf, size_t errlen, bool high_priority) {
	if (!out) return 1;
	*out = nullptr;
	int count = 0;
	hipError_t e = hipGetDeviceCount(&count);
	if (e != hipSuccess || count <= 0) {
		set_err(errbuf, errlen, "no HIP device available (%s); the anchor-distance engine has no CPU path",
				e != hipSuccess ? hipGetErrorString(e) : "device count 0");
		return 1;
	}
	if (device < 0 || device >= count) {
		set_err(errbuf, errlen, "HIP device %d out of range (have %d)", device, count);
		return 1;
	}
	e = hipSetDevice(device);
	if (e != hipSuccess) {
		set_err(errbuf, errlen, "hipSetDevice(%d): %s", device, hipGetErrorString(e));
		return 1;
	}
	auto *ctx = new andi_hip_ctx;
	ctx->device = device;
	andi_arena::retain(device); // (released in andi_hip_ctx_destroy)
	int prio = 0;
	if (high_priority) {
		int least = 0, greatest = 0;
		if (hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess) prio = greatest;
	}
	ctx->stream_prio = prio;
	e = host_pool::stream_get(&ctx->stream, device, prio);
	if (e == hipSuccess) e = hipEventCreateWithFlags(&ctx->side_fork, hipEventDisableTiming);
	if (e == hipSuccess) e = hipEventCreateWithFlags(&ctx->side_join, hipEventDisableTiming);
	if (e == hipSuccess) e = hipEventCreateWithFlags(&ctx->desc_done, hipEventDisableTiming);
	if (e == hipSuccess) e = hipEventCreateWithFlags(&ctx->built, hipEventDisableTiming);
	if (e == hipSuccess) e = hipMalloc((void **)&ctx->d_fixups, sizeof(unsigned long long));
	if (e == hipSuccess) e = hipMemset(ctx->d_fixups, 0, sizeof(unsigned long long));
	if (e == hipSuccess && !(ctx->h_quad_waves = (uint32_t *)host_pool::word_get())) e = hipErrorOutOfMemory;
	if (e == hipSuccess) e = hipEventCreateWithFlags(&ctx->coop_fork, hipEventDisableTiming);
	if (e == hipSuccess) e = hipEventCreateWithFlags(&ctx->coop_join, hipEventDisableTiming);
	if (e == hipSuccess) e = hipEventCreateWithFlags(&ctx->l2_fork, hipEventDisableTiming);
	if (e == hipSuccess) e = hipEventCreateWithFlags(&ctx->l2_join, hipEventDisableTiming);
	if (e == hipSuccess && !(ctx->h_any_left = (uint32_t *)host_pool::word_get())) e = hipErrorOutOfMemory;
	if (e == hipSuccess) e = hipMalloc((void **)&ctx->d_route, 4 * sizeof(unsigned long long));
	if (e == hipSuccess) e = hipMemset(ctx->d_route, 0, 4 * sizeof(unsigned long long));
	if (e != hipSuccess) {
		set_err(errbuf, errlen, "context setup: %s", hipGetErrorString(e));
		andi_hip_ctx_destroy(ctx);
		return 1;
	}
	*out = ctx;
	return 0;
}

int andi_hip_ctx_create(andi_hip_ctx **out, int device, char *errbuf, size_t errlen) {
	return ctx_create(out, device, errbuf, errlen, false);
}

void andi_hip_ctx_expect_queries(andi_hip_ctx *ctx, size_t queries) {
	if (ctx) ctx->queries_hint = queries;
}

void andi_hip_ctx_destroy(andi_hip_ctx *ctx) {
	if (!ctx) return;
	(void)hipSetDevice(ctx->device);
	if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
	resolve_events(ctx);
	if (ctx->scratch) (void)andi_arena::dev_free(ctx->scratch);
	if (ctx->desc_dev) (void)andi_arena::dev_free(ctx->desc_dev);
	if (ctx->desc_host) host_pool::pinned_put(ctx->desc_host, ctx->desc_bytes);
	if (ctx->d_fixups) (void)andi_arena::dev_free(ctx->d_fixups);
	if (ctx->ib_dev) (void)andi_arena::dev_free(ctx->ib_dev);
	if (ctx->ib_host) host_pool::pinned_put(ctx->ib_host, ctx->ib_cap * sizeof(AndiIndexBatchItem));
	if (ctx->ib_done) (void)hipEventDestroy(ctx->ib_done);
	if (ctx->built) (void)hipEventDestroy(ctx->built);
	if (ctx->sa_ws) (void)andi_arena::dev_free(ctx->sa_ws);
	host_pool::word_put(ctx->sa_pinned);
	host_pool::word_put(ctx->h_quad_waves);
	if (ctx->d_route) (void)andi_arena::dev_free(ctx->d_route);
	if (ctx->scratch2) (void)andi_arena::dev_free(ctx->scratch2);
	if (ctx->pool_scratch) { // (kept for the device's next context: host_pool)
		(void)hipDeviceSynchronize();
		host_pool::scratch_put(ctx->device, ctx->pool_scratch, ctx->pool_bytes + 4096);
	}
	host_pool::word_put(ctx->h_any_left);
	if (ctx->coop_stream) {
		(void)hipStreamSynchronize(ctx->coop_stream);
		host_pool::stream_put(ctx->coop_stream, ctx->device, 0);
	}
	if (ctx->coop_fork) (void)hipEventDestroy(ctx->coop_fork);
	if (ctx->coop_join) (void)hipEventDestroy(ctx->coop_join);
	if (ctx->l2_fork) (void)hipEventDestroy(ctx->l2_fork);
	if (ctx->l2_join) (void)hipEventDestroy(ctx->l2_join);
	if (ctx->desc_done) (void)hipEventDestroy(ctx->desc_done);
	if (ctx->side_stream) {
		(void)hipStreamSynchronize(ctx->side_stream);
		host_pool::stream_put(ctx->side_stream, ctx->device, ctx->stream_prio);
	}
	if (ctx->side_fork) (void)hipEventDestroy(ctx->side_fork);
	if (ctx->side_join) (void)hipEventDestroy(ctx->side_join);
	if (ctx->stream) {
		(void)hipStreamSynchronize(ctx->stream); // (events resolved above may have left work behind them)
		host_pool::stream_put(ctx->stream, ctx->device, ctx->stream_prio);
	}
	andi_arena::release(ctx->device); // (the device's last context gives its free chunks back)
	delete ctx;
}

const char *andi_hip_last_error(const andi_hip_ctx *ctx) {
	return ctx ? ctx->err.c_str() : "no context";
}

int andi_hip_sync(andi_hip_ctx *ctx) {
	if (!ctx) return 1;
	HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
	return 0;
}

int andi_hip_dev_alloc(andi_hip_ctx *ctx, size_t bytes, void **dptr) {
	if (!ctx || !dptr) return 1;
	HIP_TRY(ctx, hipSetDevice(ctx->device));
	HIP_TRY(ctx, hipMalloc(dptr, bytes ? bytes : 1));
	return 0;
}

void andi_hip_dev_free(andi_hip_ctx *ctx, void *dptr) {
	if (!ctx || !dptr) return;
	(void)hipSetDevice(ctx->device);
	(void)andi_arena::dev_free(dptr);
}

int andi_hip_copy_to_host(andi_hip_ctx *ctx, void *dst, const void *src, size_t bytes) {
	if (!ctx) return 1;
	HIP_TRY(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
	HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
	return 0;
}

// ------------------------------------------------------------------ subjects
// Allocate a subject slot able to hold an RS of up to `cap` characters.
static int esa_reserve(andi_hip_ctx *ctx, size_t cap, andi_hip_esa **out) {
	auto *e = new andi_hip_esa;
	e->cap = cap;
	hipError_t err = hipSuccess;
	auto chk = [&](hipError_t x) {
		if (err == hipSuccess) err = x;
	};
	e->deepK_cap = pick_deep_k(cap, ctx->queries_hint);
	const size_t deep_entries = (size_t)1 << (2 * e->deepK_cap);
	chk(dmalloc(&e->S, cap + 1 + ANDI_PAD));
	chk(dmalloc(&e->SA, cap + 8)); // +8: the scan reads the occurrences of a repeated K-mer eight entries at a time
	chk(dmalloc(&e->deep, deep_entries + 2)); // +2: entries are fetched with 16-byte loads
	// symbols: [front][N0: cap/2 + 1 + back][front][N1: same]; N0 and N1 start on a 256-byte boundary, the
	// paddings (NUL symbols) let whole lines of the text be fetched around any window the scan may ask for
	const size_t nib_part = (cap / 2 + 1 + ANDI_NIB_BACK + 255) & ~(size_t)255;
	chk(dmalloc(&e->Nraw, 2 * (ANDI_NIB_FRONT + nib_part)));
	if (err == hipSuccess) {
		e->N0 = e->Nraw + ANDI_NIB_FRONT, e->N1 = e->Nraw + ANDI_NIB_FRONT + nib_part + ANDI_NIB_FRONT;
		chk(hipMemsetAsync(e->Nraw, 0x77, 2 * (ANDI_NIB_FRONT + nib_part), ctx->stream));
	}
	{ // the text bit-sliced: 12 bytes per 32 symbols, a block in front, the padding all ones (NUL symbols)
		const size_t blocks = (cap + 1 + 4096) / 32 + 4;
		chk(dmalloc(&e->Praw, 3 * blocks));
		if (err == hipSuccess) {
			e->P = e->Praw + 3;
			chk(hipMemsetAsync(e->Praw, 0xff, 3 * blocks * sizeof(uint32_t), ctx->stream));
		}
	}
	// flags: pinned host memory the kernels write directly (rare, idempotent plain stores) --
	// no per-build memset or copy; the host reads them after a stream synchronisation
	if (!(e->h_flags = (int32_t *)host_pool::word_get())) chk(hipErrorOutOfMemory);
	if (err == hipSuccess) chk(hipHostGetDevicePointer((void **)&e->flags, e->h_flags, 0));
	e->bytes = (cap + 1 + ANDI_PAD) + 4 * cap + 8 * deep_entries + 80 +
			   2 * (ANDI_NIB_FRONT + nib_part);
	if (err != hipSuccess) {
		andi_hip_esa_free(ctx, e);
		return fail(ctx, "allocating a subject", err);
	}
	*out = e;
	return 0;
}

// Put a (new) subject into a slot: uploads only.  The caller may release RS/SA
// as soon as this returns.
static int esa_upload(andi_hip_ctx *ctx, andi_hip_esa *e, const char *RS, const int32_t *SA, size_t n,
					  size_t threshold, hipEvent_t done = nullptr) { // done: do not wait -- the event says when RS (and SA) may be reused
	if (n > e->cap) {
		ctx->err = "subject does not fit its slot";
		return 1;
	}
	e->n = (int32_t)n;
	e->thr = (int32_t)threshold;
	e->deepK = std::min(pick_deep_k(n, ctx->queries_hint), e->deepK_cap);
	e->ref_built = e->index_built = false;
	e->rec_valid = false;
	// the flags are functions of the text and its suffix array: cleared here, only ever set by the builds
	hipError_t err = done ? hipSuccess : hipStreamSynchronize(ctx->stream); // (the slot is the caller's: nothing of it is in flight)
	memset(e->h_flags, 0, 4 * sizeof(int32_t));
	if (err == hipSuccess) err = hipMemsetAsync(e->S + n, 0, 1 + ANDI_PAD, ctx->stream);
	if (err == hipSuccess) err = hipMemcpyAsync(e->S, RS, n, hipMemcpyHostToDevice, ctx->stream);
	if (err == hipSuccess && SA)
		err = hipMemcpyAsync(e->SA, SA, n * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream);
	if (err == hipSuccess) err = done ? hipEventRecord(done, ctx->stream) : hipStreamSynchronize(ctx->stream);
	if (err != hipSuccess) return fail(ctx, "uploading a subject", err);
	return 0;
}

// seq_subject_init (src/sequence.c:210-219) for a sequence that is already resident as a query: RS is written into the slot
// by a kernel from the query pool (esa_build.hip: k_rs_from_query) -- no host pass over the sequence, no upload.  The
// threshold is the caller's (min_anchor_length on the host, from the device's G+C count: queries_gc_counts).
static int esa_from_query(andi_hip_ctx *ctx, andi_hip_esa *e, const andi_hip_queries *Q, size_t i, size_t threshold) {
	const size_t len = Q->len[i], n = 2 * len + 1;
	if (n > e->cap) {
		ctx->err = "subject does not fit its slot";
		return 1;
	}
	e->n = (int32_t)n;
	e->thr = (int32_t)threshold;
	e->deepK = std::min(pick_deep_k(n, ctx->queries_hint), e->deepK_cap);
	e->ref_built = e->index_built = false;
	e->rec_valid = false;
	memset(e->h_flags, 0, 4 * sizeof(int32_t)); // (the slot is the caller's: nothing of it is in flight)
	hipError_t err = hipMemsetAsync(e->S + (n & ~(size_t)3), 0, (n & 3) + 1 + ANDI_PAD, ctx->stream);
	if (err == hipSuccess) err = andi_launch_rs_from_query(e->S, Q->pool + Q->off[i], (uint32_t)len, ctx->stream);
	if (err != hipSuccess) return fail(ctx, "writing a subject from its resident sequence", err);
	return 0;
}

// calc_gc's numerators (src/sequence.c:197-208) of all staged sequences, counted where they lie
static int queries_gc_counts(andi_hip_ctx *ctx, const andi_hip_queries *Q, std::vector<unsigned long long> &out) {
	HIP_TRY(ctx, hipSetDevice(ctx->device));
	unsigned long long *d = nullptr;
	HIP_TRY(ctx, dmalloc(&d, Q->nq));
	uint32_t longest = 0;
	for (uint32_t l : Q->len) longest = std::max(longest, l);
	out.assign(Q->nq, 0);
	hipError_t err = hipMemsetAsync(d, 0, Q->nq * sizeof(unsigned long long), ctx->stream);
	if (err == hipSuccess) err = andi_launch_gc_counts(Q->pool, Q->d_off, Q->d_len, (uint32_t)Q->nq, longest, d, ctx->stream);
	if (err == hipSuccess) err = hipMemcpyAsync(out.data(), d, Q->nq * sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream);
	if (err == hipSuccess) err = hipStreamSynchronize(ctx->stream);
	(void)andi_arena::dev_free(d);
	if (err != hipSuccess) return fail(ctx, "counting G+C of the staged sequences", err);
	return 0;
}

// esa_init_SA (src/esa.c:294-304) on the device: the text is in the slot, the suffix array is built there
static int esa_sort_suffixes(andi_hip_ctx *ctx, andi_hip_esa *e) {
	const size_t need = andi_sa_device_workspace(e->n);
	if (ctx->sa_ws_bytes < need) {
		HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
		if (ctx->sa_ws) (void)andi_arena::dev_free(ctx->sa_ws);
		ctx->sa_ws = nullptr, ctx->sa_ws_bytes = 0;
		HIP_TRY(ctx, andi_arena::dev_malloc(&ctx->sa_ws, need));
		ctx->sa_ws_bytes = need;
	}
	if (!ctx->sa_pinned && !(ctx->sa_pinned = (int32_t *)host_pool::word_get())) HIP_TRY(ctx, hipErrorOutOfMemory);
	if (!e->rec && !andi_knob(KNOB_NO_SORTED_RECORDS)) { // (experiments: the index build then gathers from the text, as with a host-made suffix array)
		HIP_TRY(ctx, andi_arena::dev_malloc((void **)&e->rec, (e->cap + 8) * sizeof(uint32_t)));
		HIP_TRY(ctx, andi_arena::dev_malloc((void **)&e->rec2, (e->cap + 8) * sizeof(uint16_t)));
		e->bytes += (e->cap + 8) * (sizeof(uint32_t) + sizeof(uint16_t));
	}
	const auto t0 = std::chrono::steady_clock::now();
	int rounds = 0;
	hipError_t err = andi_sa_device(e->S, e->n, e->SA, ctx->sa_ws, ctx->sa_ws_bytes, ctx->sa_pinned, ctx->stream, &rounds, e->rec, e->deepK, e->rec2);
	e->rec_valid = err == hipSuccess && e->rec != nullptr;
	if (err == hipErrorInvalidSymbol) {
		ctx->err = "a subject holds a byte outside {A,C,G,T,!,;,#}";
		return 1;
	}
	if (err != hipSuccess) return fail(ctx, "building the suffix array", err);
	ctx->acc.sa_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
	ctx->acc.sa_builds++;
	ctx->acc.sa_rounds += (uint64_t)rounds;
	return 0;
}

int andi_hip_esa_stage_text(andi_hip_ctx *ctx, const char *RS, size_t n, size_t threshold, andi_hip_esa **out) {
	if (!ctx || !RS || !out || n == 0 || n >= (size_t)INT32_MAX) {
		if (ctx) ctx->err = "andi_hip_esa_stage_text: bad arguments";
		return 1;
	}
	HIP_TRY(ctx, hipSetDevice(ctx->device));
	andi_hip_esa *e = nullptr;
	if (esa_reserve(ctx, n, &e)) return 1;
	if (esa_upload(ctx, e, RS, nullptr, n, threshold) || esa_sort_suffixes(ctx, e)) {
		andi_hip_esa_free(ctx, e);
		return 1;
	}
	*out = e;
	return 0;
}

int andi_hip_esa_download_sa(andi_hip_ctx *ctx, const andi_hip_esa *e, int32_t *SA) {
	if (!ctx || !e || !SA) return 1;
	HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
	HIP_TRY(ctx, hipMemcpy(SA, e->SA, (size_t)e->n * sizeof(int32_t), hipMemcpyDeviceToHost));
	return 0;
}

int andi_hip_esa_stage(andi_hip_ctx *ctx, const char *RS, const int32_t *SA, size_t n,
					   size_t threshold, andi_hip_esa **out) {
	if (!ctx || !RS || !SA || !out || n == 0 || n >= (size_t)INT32_MAX) {
		if (ctx) ctx->err = "andi_hip_esa_stage: bad arguments";
		return 1;
	}
	HIP_TRY(ctx, hipSetDevice(ctx->device));
	andi_hip_esa *e = nullptr;
	if (esa_reserve(ctx, n, &e)) return 1;
	if (esa_upload(ctx, e, RS, SA, n, threshold)) {
		andi_hip_esa_free(ctx, e);
		return 1;
	}
	*out = e;
	return 0;
}

static int ensure_reference_buffers(andi_hip_ctx *ctx, andi_hip_esa *e) {
	if (e->LCP && e->ref_cap >= (size_t)e->n) return 0;
	if (e->LCP) { // slot reused for a longer subject
		HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
		(void)andi_arena::dev_free(e->LCP), (void)andi_arena::dev_free(e->CLD), (void)andi_arena::dev_free(e->FVC), (void)andi_arena::dev_free(e->tab);
		(void)andi_arena::dev_free(e->min_scratch);
		e->LCP = e->CLD = nullptr, e->FVC = nullptr, e->tab = nullptr, e->min_scratch = nullptr;
	}
	const size_t n = e->cap;
	e->ref_cap = n;
	const size_t tab_entries = (size_t)1 << (2 * ANDI_CACHE_K);
	const size_t mins = andi_min_tree_entries((int32_t)n);
	hipError_t err = hipSuccess;
	auto chk = [&](hipError_t x) {
		if (err == hipSuccess) err = x;
	};
	chk(dmalloc(&e->LCP, n + 1));
	chk(dmalloc(&e->CLD, n + 1));
	chk(dmalloc(&e->FVC, n + ANDI_PAD));
	chk(dmalloc(&e->tab, tab_entries));
	chk(dmalloc(&e->min_scratch, mins));
	if (err != hipSuccess) return fail(ctx, "allocating the reference arrays", err);
	e->bytes += 8 * (n + 1) + n + ANDI_PAD + 16 * tab_entries + 4 * mins;
	return 0;
}

int andi_hip_esa_build(andi_hip_ctx *ctx, andi_hip_esa *e) {
	if (!ctx || !e) return 1;
	HIP_TRY(ctx, hipSetDevice(ctx->device));
	if (ensure_reference_buffers(ctx, e)) return 1;
	Timed t(ctx, 0);
	hipError_t err = andi_launch_esa_build(build_args(e), ctx->stream);
	t.stop();
	if (err == hipSuccess) err = hipEventRecord(ctx->built, ctx->stream);
	if (err != hipSuccess) return fail(ctx, "andi_hip_esa_build", err);
	e->ref_built = true, ctx->builds_pending = true;
	return 0;
}

int andi_hip_esa_build_index(andi_hip_ctx *ctx, andi_hip_esa *e) {
	if (!ctx || !e) return 1;
	HIP_TRY(ctx, hipSetDevice(ctx->device));
	Timed t(ctx, 0);
	const int ext = andi_index_single_ext(ctx->queries_hint, e->rec_valid);
	hipError_t err = andi_launch_index_build(build_args(e), ext, ctx->stream);
	t.stop();
	if (err == hipSuccess) err = hipEventRecord(ctx->built, ctx->stream);
	if (err != hipSuccess) return fail(ctx, "andi_hip_esa_build_index", err);
	e->index_built = true, e->deep_ext = (ext == 2 && !e->rec_valid) ? 0 : ext, ctx->builds_pending = true; // (the short form forced on a host-made suffix array: plain)
	return 0;
}

int andi_hip_esa_build_index_batch(andi_hip_ctx *ctx, andi_hip_esa *const *esas, size_t count) {
	if (!ctx || !esas || count == 0 || count > 65535) {
		if (ctx) ctx->err = "andi_hip_esa_build_index_batch: bad arguments";
		return 1;
	}
	HIP_TRY(ctx, hipSetDevice(ctx->device));
	if (ctx->ib_cap < count) {
		HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
		if (ctx->ib_dev) (void)andi_arena::dev_free(ctx->ib_dev);
		if (ctx->ib_host) host_pool::pinned_put(ctx->ib_host, ctx->ib_cap * sizeof(AndiIndexBatchItem));
		ctx->ib_dev = ctx->ib_host = nullptr, ctx->ib_cap = 0;
		const size_t cap = std::max<size_t>(count, 64);
		HIP_TRY(ctx, andi_arena::dev_malloc(&ctx->ib_dev, cap * sizeof(AndiIndexBatchItem)));
		HIP_TRY(ctx, host_pool::pinned_get(&ctx->ib_host, cap * sizeof(AndiIndexBatchItem)));
		if (!ctx->ib_done) HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->ib_done, hipEventDisableTiming));
		ctx->ib_cap = cap;
	} else {
		HIP_TRY(ctx, hipEventSynchronize(ctx->ib_done)); // the previous batch's items have been copied
	}
	auto *items = (AndiIndexBatchItem *)ctx->ib_host;
	int32_t max_n = 0;
	for (size_t k = 0; k < count; ++k) {
		andi_hip_esa *e = esas[k];
		if (!e) {
			ctx->err = "andi_hip_esa_build_index_batch: null subject";
			return 1;
		}
		items[k].S = e->S, items[k].SA = e->SA, items[k].deep = e->deep, items[k].N0 = e->N0, items[k].N1 = e->N1, items[k].P = e->P;
		items[k].flags = e->flags, items[k].n = e->n, items[k].deepK = e->deepK;
		items[k].rec = e->rec_valid ? e->rec : nullptr;
		items[k].rec2 = e->rec_valid ? e->rec2 : nullptr;
		items[k].single_ext = andi_index_single_ext(ctx->queries_hint, e->rec_valid);
		max_n = std::max(max_n, e->n);
	}
	Timed t(ctx, 0);
	hipError_t err = hipMemcpyAsync(ctx->ib_dev, ctx->ib_host, count * sizeof(AndiIndexBatchItem), hipMemcpyHostToDevice, ctx->stream);
	if (err == hipSuccess) err = hipEventRecord(ctx->ib_done, ctx->stream);
	if (err == hipSuccess) err = andi_launch_index_build_batch((const AndiIndexBatchItem *)ctx->ib_dev, (uint32_t)count, max_n, ctx->stream);
	t.stop();
	if (err == hipSuccess) err = hipEventRecord(ctx->built, ctx->stream);
	if (err != hipSuccess) return fail(ctx, "andi_hip_esa_build_index_batch", err);
	for (size_t k = 0; k < count; ++k) {
		const int ext = items[k].single_ext;
		esas[k]->index_built = true, esas[k]->deep_ext = (ext == 2 && !esas[k]->rec_valid) ? 0 : ext;
	}
	ctx->builds_pending = true;
	return 0;
}

int andi_hip_esa_flags(andi_hip_ctx *ctx, const andi_hip_esa *e, int32_t *out4) {
	if (!ctx || !e || !out4) return 1;
	HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
	memcpy(out4, e->h_flags, 4 * sizeof(int32_t));
	return 0;
}

int andi_hip_esa_download(andi_hip_ctx *ctx, const andi_hip_esa *e, int32_t *LCP, int32_t *CLD,
						  uint8_t *FVC, andi_hip_interval *cache) {
	if (!ctx || !e) return 1;
	if (!e->ref_built) {
		ctx->err = "andi_hip_esa_download: reference arrays not built (call andi_hip_esa_build)";
		return 1;
	}
	const size_t n = (size_t)e->n;
	HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
	if (LCP) HIP_TRY(ctx, hipMemcpy(LCP, e->LCP, (n + 1) * 4, hipMemcpyDeviceToHost));
	if (CLD) HIP_TRY(ctx, hipMemcpy(CLD, e->CLD, (n + 1) * 4, hipMemcpyDeviceToHost));
	if (FVC) HIP_TRY(ctx, hipMemcpy(FVC, e->FVC, n, hipMemcpyDeviceToHost));
	if (cache)
		HIP_TRY(ctx, hipMemcpy(cache, e->tab, ((size_t)16 << (2 * ANDI_CACHE_K)), hipMemcpyDeviceToHost));
	return 0;
}

int andi_hip_esa_download_index(andi_hip_ctx *ctx, const andi_hip_esa *e, uint32_t *table, int *K) {
	if (!ctx || !e) return 1;
	if (!e->index_built) {
		ctx->err = "andi_hip_esa_download_index: scan index not built (call andi_hip_esa_build_index)";
		return 1;
	}
	if (K) *K = e->deepK;
	HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
	if (table) HIP_TRY(ctx, hipMemcpy(table, e->deep, (size_t)8 << (2 * e->deepK), hipMemcpyDeviceToHost));
	return 0;
}

int andi_hip_esa_single_form(const andi_hip_esa *e) { return e && e->index_built ? e->deep_ext : 0; }

void andi_hip_esa_free(andi_hip_ctx *ctx, andi_hip_esa *e) {
	if (!e) return;
	if (ctx) (void)hipSetDevice(ctx->device);
	(void)hipDeviceSynchronize(); // once for the handle's ten buffers: nothing in flight uses them when they are handed out again
	void *bufs[] = {e->S, e->SA, e->LCP, e->CLD, e->FVC, e->tab, e->deep, e->Nraw, e->Praw, e->rec, e->rec2, e->min_scratch};
	for (void *b : bufs) (void)andi_arena::dev_free(b, false);
	host_pool::word_put(e->h_flags);
	delete e;
}

size_t andi_hip_esa_bytes(const andi_hip_esa *e) {
	return e ? e->bytes : 0;
}

// ------------------------------------------------------------------ queries
// the contig separators of every staged sequence (d_off, d_len and the byte pool are queued on the stream): one pass over the pool
static hipError_t queries_count_separators(andi_hip_ctx *ctx, andi_hip_queries *q) {
	hipError_t err = dmalloc(&q->d_sep, q->nq);
	if (err != hipSuccess) return err;
	uint32_t longest = 0;
	for (uint32_t l : q->len) longest = std::max(longest, l);
	err = hipMemsetAsync(q->d_sep, 0, q->nq * sizeof(uint32_t), ctx->stream);
	if (err == hipSuccess) err = andi_launch_sep_counts(q->pool, q->d_off, q->d_len, (uint32_t)q->nq, longest, q->d_sep, ctx->stream);
	return err;
}

int andi_hip_queries_stage(andi_hip_ctx *ctx, const andi_hip_seq *seqs, size_t n,
						   andi_hip_queries **out) {
	if (!ctx || !seqs || !out || n == 0 || n >= (size_t)UINT32_MAX) {
		if (ctx) ctx->err = "andi_hip_queries_stage: bad arguments";
		return 1;
	}
	HIP_TRY(ctx, hipSetDevice(ctx->device));
	auto *q = new andi_hip_queries;
	q->nq = n;
	q->off.resize(n);
	q->len.resize(n);
	uint64_t cursor = 0;
	for (size_t i = 0; i < n; ++i) {
		if (!seqs[i].seq || seqs[i].len == 0 || seqs[i].len > (size_t)(INT32_MAX - 1) / 2) {
			ctx->err = "andi_hip_queries_stage: empty or oversized sequence";
			delete q;
			return 1;
		}
		q->off[i] = cursor;
		q->len[i] = (uint32_t)seqs[i].len;
		q->total_nt += seqs[i].len;
		cursor += (seqs[i].len + 1 + 255) & ~(uint64_t)255; // NUL; starts on 256-byte boundaries (128 of the packed pool: a cache line)
	}
	const size_t pool_bytes = cursor + ANDI_PAD;
	hipError_t err = hipSuccess;
	auto chk = [&](hipError_t x) {
		if (err == hipSuccess) err = x;
	};
	chk(dmalloc(&q->pool, pool_bytes));
	chk(dmalloc(&q->nib, pool_bytes / 2 + 64));
	chk(dmalloc(&q->planes, 3 * (pool_bytes / 32) + 16));
	if (!(q->h_foreign = (int32_t *)host_pool::word_get())) chk(hipErrorOutOfMemory);
	int32_t *d_foreign = nullptr;
	chk(dmalloc(&d_foreign, 1));
	chk(dmalloc(&q->d_off, n));
	chk(dmalloc(&q->d_len, n));
	if (err == hipSuccess) err = hipMemsetAsync(q->pool, 0, pool_bytes, ctx->stream);
	uint64_t threaded_from = (uint64_t)1 << 31;
	if (const char *um = andi_knob(KNOB_UPLOAD_MIN_MB)) // (tests: the threaded path on small sets)
		if (atoi(um) >= 0) threaded_from = (uint64_t)atoi(um) << 20;
	if (err == hipSuccess && q->total_nt >= threaded_from) {
		// Gigabytes of sequences in pageable memory (BASELINE's config 3: 6.5 GB): one hipMemcpyAsync per sequence went through the
		// runtime's staging at 10-13 GB/s (0.5-0.6 of the 10.5 s of the 3085 x 3085 matrix, before anything else can begin: now 0.35 s).  Four
		// host threads instead, each copying its share chunk by chunk into one of its two pinned buffers while the other's
		// transfer runs on a stream of its own.
		err = hipStreamSynchronize(ctx->stream); // (the pool's zeroes first: the copies run on other streams)
		constexpr size_t CHUNK = (size_t)8 << 20;
		const int nt = 4;
		std::atomic<size_t> next_seq{0};
		std::atomic<int> bad{0};
		auto work = [&]() {
			(void)hipSetDevice(ctx->device);
			hipStream_t st = nullptr;
			void *pin = nullptr;
			hipEvent_t ev[2] = {nullptr, nullptr};
			bool ok = host_pool::stream_get(&st, ctx->device, 0) == hipSuccess && host_pool::pinned_get(&pin, 2 * CHUNK) == hipSuccess &&
					  hipEventCreateWithFlags(&ev[0], hipEventDisableTiming) == hipSuccess && hipEventCreateWithFlags(&ev[1], hipEventDisableTiming) == hipSuccess;
			size_t k = 0; // chunks this thread has sent
			while (ok && !bad.load()) {
				const size_t i = next_seq.fetch_add(1);
				if (i >= n) break;
				for (size_t o = 0; o < seqs[i].len && ok; o += CHUNK, ++k) {
					const size_t len = std::min(CHUNK, seqs[i].len - o);
					char *buf = (char *)pin + (k & 1) * CHUNK;
					if (k >= 2) ok = hipEventSynchronize(ev[k & 1]) == hipSuccess; // (the transfer that last read this buffer)
					memcpy(buf, seqs[i].seq + o, len);
					ok = ok && hipMemcpyAsync(q->pool + q->off[i] + o, buf, len, hipMemcpyHostToDevice, st) == hipSuccess && hipEventRecord(ev[k & 1], st) == hipSuccess;
				}
			}
			if (st) ok = hipStreamSynchronize(st) == hipSuccess && ok;
			if (!ok) bad.store(1);
			for (hipEvent_t e : ev)
				if (e) (void)hipEventDestroy(e);
			if (pin) host_pool::pinned_put(pin, 2 * CHUNK);
			if (st) host_pool::stream_put(st, ctx->device, 0);
		};
		std::vector<std::thread> ts;
		for (int t = 1; t < nt; ++t) ts.emplace_back(work);
		work();
		for (auto &t : ts) t.join();
		if (bad.load() && err == hipSuccess) err = hipErrorUnknown;
	} else {
		for (size_t i = 0; i < n && err == hipSuccess; ++i)
			err = hipMemcpyAsync(q->pool + q->off[i], seqs[i].seq, seqs[i].len, hipMemcpyHostToDevice,
								 ctx->stream);
	}
	if (err == hipSuccess)
		err = hipMemcpyAsync(q->d_off, q->off.data(), n * 8, hipMemcpyHostToDevice, ctx->stream);
	if (err == hipSuccess)
		err = hipMemcpyAsync(q->d_len, q->len.data(), n * 4, hipMemcpyHostToDevice, ctx->stream);
	// 4-bit symbols of the whole pool (pool_bytes is a multiple of 16)
	if (err == hipSuccess) err = hipMemsetAsync(d_foreign, 0, sizeof(int32_t), ctx->stream);
	if (err == hipSuccess)
		err = andi_launch_pack_symbols(q->pool, pool_bytes, q->nib, nullptr, d_foreign, ctx->stream);
	if (err == hipSuccess) err = andi_launch_pack_planes(q->nib, pool_bytes, q->planes, ctx->stream);
	if (err == hipSuccess) err = queries_count_separators(ctx, q);
	if (err == hipSuccess)
		err = hipMemcpyAsync(q->h_foreign, d_foreign, sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream);
	if (err == hipSuccess) err = hipStreamSynchronize(ctx->stream);
	(void)andi_arena::dev_free(d_foreign);
	if (err != hipSuccess) {
		andi_hip_queries_free(ctx, q);
		return fail(ctx, "andi_hip_queries_stage", err);
	}

	*out = q;
	return 0;
}

void andi_hip_queries_free(andi_hip_ctx *ctx, andi_hip_queries *q) {
	if (!q) return;
	if (ctx) (void)hipSetDevice(ctx->device);
	(void)hipDeviceSynchronize();
	void *bufs[] = {q->pool, q->nib, q->planes, q->d_off, q->d_len, q->d_sep, q->d_qseg_start, q->d_seg2query, q->c_qseg_start, q->c_seg2query};
	for (void *b : bufs) (void)andi_arena::dev_free(b, false);
	host_pool::word_put(q->h_foreign);
	delete q;
}

// 4-bit symbols of a byte string, as the device's pack kernel makes them (scan_lane.hip: symbol_of, in_alphabet): byte j of
// `out` = symbol 2j | symbol 2j+1 << 4, the NUL behind an odd length included; (len + 1) / 2 bytes.  Eight bytes at a time
// where they are all nucleotides (what genomes are made of); returns 1 if a byte lies outside {A,C,G,T,!,;,#,NUL}.
extern "C" int andi_hip_pack_symbols(const unsigned char *src, size_t len, unsigned char *out) {
	static const struct Lut {
		uint8_t v[256];
		Lut() {
			for (int c = 0; c < 256; ++c) {
				const uint8_t ch = (uint8_t)c;
				const uint32_t sym = ch >= 'A' ? (uint32_t)(((ch & 6u) ^ ((ch & 6u) >> 1)) >> 1) : (ch == '!' ? 4u : ch == ';' ? 5u : ch == '#' ? 6u : 7u);
				const bool in = ch == 'A' || ch == 'C' || ch == 'G' || ch == 'T' || ch == '!' || ch == ';' || ch == '#' || ch == 0;
				v[c] = (uint8_t)(sym | (in ? 0u : 0x80u));
			}
		}
	} lut;
	uint32_t bad = 0;
	size_t k = 0;
	const uint64_t ones = 0x0101010101010101ull;
	for (; k + 8 <= len; k += 8) {
		uint64_t x;
		memcpy(&x, src + k, 8);
		const uint64_t t = ((x >> 1) ^ (x >> 2)) & (3 * ones); // a nucleotide's code, per byte: ((c & 6) ^ ((c & 6) >> 1)) >> 1
		const uint64_t b0 = t & ones, b1 = (t >> 1) & ones;
		const uint64_t mc = b0 & ~b1, mg = b1 & ~b0, mt = b0 & b1;      // C, G, T (no carries: the sums stay below 0x55)
		const uint64_t expect = 0x41 * ones + 2 * mc + 6 * mg + 0x13 * mt; // the letter that code belongs to
		if (x == expect) {                                                // eight nucleotides: four bytes of symbols
			uint64_t z = (t | (t >> 4)) & 0x00ff00ff00ff00ffull;            // 16-bit lanes: symbol 2j | symbol 2j+1 << 4
			z = (z | (z >> 8)) & 0x0000ffff0000ffffull;
			const uint32_t o = (uint32_t)(z | (z >> 16));
			memcpy(out + k / 2, &o, 4);
		} else {
			for (size_t j = k; j < k + 8; j += 2) {
				const uint32_t lo = lut.v[src[j]], hi = lut.v[src[j + 1]];
				bad |= lo | hi;
				out[j / 2] = (uint8_t)((lo & 7u) | ((hi & 7u) << 4));
			}
		}
	}
	for (; k + 1 < len; k += 2) {
		const uint32_t lo = lut.v[src[k]], hi = lut.v[src[k + 1]];
		bad |= lo | hi;
		out[k / 2] = (uint8_t)((lo & 7u) | ((hi & 7u) << 4));
	}
	if (k < len) { // an odd length: the NUL behind the string is the last byte's other symbol
		const uint32_t lo = lut.v[src[k]];
		bad |= lo;
		out[k / 2] = (uint8_t)((lo & 7u) | (7u << 4));
	}
	return (bad & 0x80u) ? 1 : 0;
}

// The seam's queries, packed ONCE on the host (round 4): every device uploads the 4-bit pool -- a quarter of what the
// byte pool and its packed copy were, from one host copy shared by the device threads -- and unpacks the bytes the rare
// byte-wise paths read (k_unpack_symbols).  C4's 6.5 GB of queries took 0.35 s per device as bytes from pageable memory.
struct PackedQueries {
	std::vector<uint64_t> off;
	std::vector<uint32_t> len;
	uint64_t total_nt = 0;
	size_t pool_bytes = 0;
	uint8_t *nib = nullptr; // pool_bytes / 2 bytes: the pool as the device's pack kernel would leave it
	int foreign = 0;        // a byte outside the alphabet (the scan refuses the queries then)
	std::string err;
	std::atomic<int> users{0}; // devices that have not staged yet: the last one lets the host copy go (gigabytes: not at the call's end)
	void release() {
		free(nib);
		nib = nullptr;
	}
	~PackedQueries() { release(); }
};

static int pack_queries_host(const andi_hip_seq *seqs, size_t n, int threads, PackedQueries &P) {
	if (!seqs || n == 0 || n >= (size_t)UINT32_MAX) {
		P.err = "andi_hip_queries_stage: bad arguments";
		return 1;
	}
	P.off.resize(n), P.len.resize(n);
	uint64_t cursor = 0;
	for (size_t i = 0; i < n; ++i) { // (the layout of andi_hip_queries_stage)
		if (!seqs[i].seq || seqs[i].len == 0 || seqs[i].len > (size_t)(INT32_MAX - 1) / 2) {
			P.err = "andi_hip_queries_stage: empty or oversized sequence";
			return 1;
		}
		P.off[i] = cursor, P.len[i] = (uint32_t)seqs[i].len, P.total_nt += seqs[i].len;
		cursor += (seqs[i].len + 1 + 255) & ~(uint64_t)255;
	}
	P.pool_bytes = cursor + ANDI_PAD;
	// (2 MiB-aligned and advised as huge pages: 256 threads touching gigabytes of fresh 4 KiB pages queue up in the kernel --
	// C5's 6.4 GB took 1.3 s to pack that way)
	const size_t nib_bytes = (P.pool_bytes / 2 + ((size_t)2 << 20) - 1) & ~(((size_t)2 << 20) - 1);
	P.nib = (uint8_t *)aligned_alloc((size_t)2 << 20, nib_bytes);
#ifdef MADV_HUGEPAGE
	if (P.nib) (void)madvise(P.nib, nib_bytes, MADV_HUGEPAGE);
#endif
	if (!P.nib) {
		P.err = "andi_hip_dist_matrix: out of host memory for the packed queries";
		return 1;
	}
	std::atomic<size_t> next{0};
	std::atomic<int> foreign{0};
	auto work = [&]() {
		for (;;) {
			const size_t i = next.fetch_add(1);
			if (i >= n) return;
			const size_t len = seqs[i].len, end = (i + 1 < n ? (size_t)P.off[i + 1] : P.pool_bytes) / 2;
			uint8_t *dst = P.nib + P.off[i] / 2; // (offsets are multiples of 256)
			const size_t w = (len + 1) / 2;
			if (andi_hip_pack_symbols((const unsigned char *)seqs[i].seq, len, dst)) foreign.store(1);
			memset(dst + w, 0x77, end - (P.off[i] / 2 + w)); // NUL, NUL up to the next sequence (the pool's end)
		}
	};
	// (two dozen threads keep up with the host's memory; all 256 cores packing starved the devices' context creation, which
	// runs beside this: C5's contexts 1.3 -> 2.8 s)
	const int nt = std::max(1, std::min(std::min(threads, 24), (int)std::min<size_t>(n, 256)));
	std::vector<std::thread> ts;
	for (int t = 1; t < nt; ++t) ts.emplace_back(work);
	work();
	for (auto &t : ts) t.join();
	P.foreign = foreign.load();
	return 0;
}

static int queries_stage_packed(andi_hip_ctx *ctx, const PackedQueries &P, andi_hip_queries **out) {
	HIP_TRY(ctx, hipSetDevice(ctx->device));
	auto *q = new andi_hip_queries;
	const size_t n = P.off.size();
	q->nq = n, q->off = P.off, q->len = P.len, q->total_nt = P.total_nt;
	hipError_t err = hipSuccess;
	auto chk = [&](hipError_t x) {
		if (err == hipSuccess) err = x;
	};
	chk(dmalloc(&q->pool, P.pool_bytes));
	chk(dmalloc(&q->nib, P.pool_bytes / 2 + 64));
	chk(dmalloc(&q->planes, 3 * (P.pool_bytes / 32) + 16));
	if (!(q->h_foreign = (int32_t *)host_pool::word_get())) chk(hipErrorOutOfMemory);
	chk(dmalloc(&q->d_off, n));
	chk(dmalloc(&q->d_len, n));
	if (err == hipSuccess) *q->h_foreign = P.foreign;
	if (err == hipSuccess) err = hipMemcpyAsync(q->nib, P.nib, P.pool_bytes / 2, hipMemcpyHostToDevice, ctx->stream);
	if (err == hipSuccess) err = andi_launch_unpack_symbols(q->nib, P.pool_bytes, q->pool, ctx->stream);
	if (err == hipSuccess) err = andi_launch_pack_planes(q->nib, P.pool_bytes, q->planes, ctx->stream);
	if (err == hipSuccess) err = hipMemcpyAsync(q->d_off, q->off.data(), n * 8, hipMemcpyHostToDevice, ctx->stream);
	if (err == hipSuccess) err = hipMemcpyAsync(q->d_len, q->len.data(), n * 4, hipMemcpyHostToDevice, ctx->stream);
	if (err == hipSuccess) err = queries_count_separators(ctx, q);
	if (err == hipSuccess) err = hipStreamSynchronize(ctx->stream);
	if (err != hipSuccess) {
		andi_hip_queries_free(ctx, q);
		return fail(ctx, "andi_hip_dist_matrix: staging the packed queries", err);
	}
	*out = q;
	return 0;
}

static int ensure_segmentation(andi_hip_ctx *ctx, andi_hip_queries *q, uint32_t seg, bool coop = false) {
	uint32_t &have = coop ? q->c_seg : q->seg, &total_out = coop ? q->c_total_segs : q->total_segs;
	uint32_t *&d_start = coop ? q->c_qseg_start : q->d_qseg_start, *&d_s2q = coop ? q->c_seg2query : q->d_seg2query;
	if (have == seg && d_start) return 0;
	HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
	(void)andi_arena::dev_free(d_start);
	(void)andi_arena::dev_free(d_s2q);
	d_start = d_s2q = nullptr;
	std::vector<uint32_t> start(q->nq + 1);
	uint64_t total = 0;
	for (size_t i = 0; i < q->nq; ++i) {
		start[i] = (uint32_t)total;
		total += (q->len[i] + (uint64_t)seg - 1) / seg;
		if (total >= UINT32_MAX) {
			ctx->err = "too many scan segments; raise opts.segment";
			return 1;
		}
	}
	start[q->nq] = (uint32_t)total;
	std::vector<uint32_t> s2q((size_t)total);
	for (size_t i = 0; i < q->nq; ++i)
		for (uint32_t w = start[i]; w < start[i + 1]; ++w) s2q[w] = (uint32_t)i;
	HIP_TRY(ctx, dmalloc(&d_start, q->nq + 1));
	HIP_TRY(ctx, dmalloc(&d_s2q, (size_t)total));
	HIP_TRY(ctx, hipMemcpy(d_start, start.data(), (q->nq + 1) * 4, hipMemcpyHostToDevice));
	HIP_TRY(ctx, hipMemcpy(d_s2q, s2q.data(), (size_t)total * 4, hipMemcpyHostToDevice));
	have = seg;
	total_out = (uint32_t)total;
	return 0;
}

int andi_hip_match_positions(andi_hip_ctx *ctx, const andi_hip_esa *esa, const andi_hip_queries *q,
							 size_t qidx, size_t first, size_t count, int cached,
							 andi_hip_interval *out_host) {
	if (!ctx || !esa || !q || !out_host || qidx >= q->nq || !esa->ref_built) {
		if (ctx) ctx->err = "andi_hip_match_positions: bad arguments (reference arrays not built?)";
		return 1;
	}
	if (first + count > q->len[qidx]) {
		ctx->err = "andi_hip_match_positions: range beyond the query";
		return 1;
	}
	if (count == 0) return 0;
	HIP_TRY(ctx, hipSetDevice(ctx->device));
	andi_hip_interval *d_out = nullptr;
	HIP_TRY(ctx, dmalloc(&d_out, count));
	hipError_t e = andi_launch_match_positions(esa_view(esa, ANDI_MODE_REFERENCE), q->pool + q->off[qidx], q->len[qidx],
											   (uint32_t)first, (uint32_t)count, cached, d_out,
											   ctx->stream);
	if (e == hipSuccess)
		e = hipMemcpyAsync(out_host, d_out, count * sizeof(andi_hip_interval), hipMemcpyDeviceToHost,
						   ctx->stream);
	if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
	(void)andi_arena::dev_free(d_out);
	if (e != hipSuccess) return fail(ctx, "andi_hip_match_positions", e);
	return 0;
}

// ------------------------------------------------------------------ scan
int andi_hip_scan_rows(andi_hip_ctx *ctx, andi_hip_esa *const *subjects, const int64_t *self,
					   size_t nsub, const andi_hip_queries *q_const, int model, uint32_t segment,
					   andi_hip_model *M_dev) {
	if (!ctx || !subjects || !q_const || !M_dev || nsub == 0 || nsub > 65535) {
		if (ctx) ctx->err = "andi_hip_scan_rows: bad arguments";
		return 1;
	}
	if (model < ANDI_M_RAW || model > ANDI_M_ANI) {
		ctx->err = "andi_hip_scan_rows: unknown model";
		return 1;
	}
	auto *q = const_cast<andi_hip_queries *>(q_const);
	if (*q->h_foreign) { // the reference's reader never produces that (src/sequence.c:260-282)
		ctx->err = "andi_hip_scan_rows: a query holds a byte outside {A,C,G,T,!}";
		return 1;
	}
	HIP_TRY(ctx, hipSetDevice(ctx->device));
	// the streams only scans use (a context that stages or uploads never asks for them: a stream costs 3 ms to create)
	if (!ctx->side_stream) HIP_TRY(ctx, host_pool::stream_get(&ctx->side_stream, ctx->device, ctx->stream_prio));
	if (!ctx->coop_stream) HIP_TRY(ctx, host_pool::stream_get(&ctx->coop_stream, ctx->device, 0));
	// segment == 0: the engine chooses.  With the lane scan and a moderate number of pairs
	// the segment length is chosen per pair (scan_lane.hip: k_pair_estimate); otherwise one
	// length for the call.
	// Pass A with one wavefront per chain (scan_coop.hip) for the models that split an anchor's length evenly and
	// thresholds a 32-symbol window can decide.  ANDI_COOP=n: the call's pass A, one (long) segment length for the call.
	// Unset: large calls are ROUTED PER PAIR (scan.h) -- the pairs whose sampled matches suit that kernel take it, on a
	// segmentation of its own; the others, and the pairs it hands back, take the lane scan; passes B and C run per layout.
	const int coop_mode = andi_coop_enabled();
	int coop_ok = coop_mode != 0 && !andi_knob(KNOB_FORCE_REFERENCE);
	for (size_t s = 0; s < nsub && coop_ok; ++s)
		if (!subjects[s] || subjects[s]->thr < 2 || subjects[s]->thr > 30) coop_ok = 0;
	const uint64_t call_nt = q->total_nt * (uint64_t)nsub;
	// TINY calls (less than two rounds of wavefronts on 2048-symbol segments): that kernel for every pair, without the
	// sampling -- whatever a pair is like, a wavefront's chain over 2048 symbols is no longer than a lane's over 4096, and
	// the device has the wavefronts to spare (structured genomes 3 x 1 Mbp ... 5 x 1.3 Mbp: 2.1 ... 3.3 ms by wavefronts,
	// 2.65 ... 3.4 by lanes; clean ones 3 x 1 Mbp: 0.34 against 0.54 routed, 0.96 by lanes)
	uint64_t tiny_nt = ANDI_ROUTE_TINY_NT;
	if (const char *rt = andi_knob(KNOB_ROUTE_TINY)) // (tests, experiments: log2 of that size; 1: no call is tiny, small ones are routed)
		if (atoi(rt) > 0 && atoi(rt) < 63) tiny_nt = 1ull << atoi(rt);
	bool tiny = coop_ok && coop_mode < 0 && segment == 0 && call_nt >= ANDI_ROUTE_MIN_NT && call_nt < tiny_nt &&
				!andi_knob(KNOB_UNIFORM_SEGMENTS) && !andi_knob(KNOB_FORCE_ADAPTIVE);
	// A wavefront needs far fewer chains in flight than a lane, and every segment costs it a cold start of a dozen
	// dependent round trips: segments as long as leave the device four rounds of wavefronts (24576), 32768 ... 524288
	// symbols (measured: bench set 5.57 / 5.39 / 5.31 / 5.34 ms at 32768 / 65536 / 131072 / 262144, C4 shape 38.6 / 33.5 /
	// 32.6 / 32.6 / 35.4 / 44.0 ms at 32768 / 131072 / 262144 / 524288 / 2^20 / 2^21 -- whole queries: pairs differ too much)
	// Small calls: shorter segments still, as long as the device has one round of wavefronts (2048 symbols at least) --
	// 3 x 1 Mbp (BASELINE's configs[0]): pass A 0.10 ms by wavefronts against 0.66 ms by lanes; 100 x 30 kbp 0.94 against
	// 2.35 ms per call (profiles/r05_small_calls.txt).
	uint32_t coop_seg = 524288;
	while (coop_seg > 32768 && q->total_nt * (uint64_t)nsub / coop_seg < 24576) coop_seg /= 2;
	while (coop_seg > 2048 && q->total_nt * (uint64_t)nsub / coop_seg < 12000u) coop_seg /= 2; // (8 x 1 Mbp: 0.38 ms at 4096 -- 15 600 wavefronts --, 0.46 at 2048; 12 x 1 Mbp: 0.65 at 8192 -- 17 600 --, 0.74 at 4096)
	if (const char *cs = andi_knob(KNOB_COOP_SEG)) // experiments
		if (atoi(cs) >= 64) coop_seg = (uint32_t)atoi(cs);
	// (the smallest calls -- a few launches' worth of work -- keep the lane scan: routing costs them the sampling kernel
	// and two looks of the host at the device)
	bool routed = coop_ok && coop_mode < 0 && segment == 0 && call_nt >= ANDI_ROUTE_MIN_NT &&
				  nsub * q->nq <= ANDI_ADAPTIVE_MAX_PAIRS && !andi_knob(KNOB_UNIFORM_SEGMENTS) && !andi_knob(KNOB_FORCE_ADAPTIVE);
	if (routed || tiny) { // (queries shorter than the wavefront kernel takes -- k_pair_estimate -- are the lane scan's: where they are most of the call, all of it)
		uint64_t cand_nt = 0;
		for (size_t i = 0; i < q->nq; ++i)
			if (q->len[i] >= std::min(coop_seg, ANDI_ROUTE_MIN_QLEN)) cand_nt += q->len[i];
		if (2 * cand_nt < q->total_nt) routed = tiny = false;
	}
	if (tiny) routed = false;
	const int coop = coop_ok && (coop_mode > 0 || tiny);
	const bool want_adaptive = !coop && segment == 0 && nsub * q->nq <= ANDI_ADAPTIVE_MAX_PAIRS &&
							   !andi_knob(KNOB_UNIFORM_SEGMENTS);
	if (segment == 0 && coop) segment = coop_seg;
	if (segment == 0) {
		uint64_t nt = q->total_nt * (uint64_t)nsub;
		segment = ANDI_MIN_SEGMENT;
		while (segment < ANDI_MAX_SEGMENT && nt / segment > ANDI_TARGET_CHAINS) segment *= 2;
	}
	if (ensure_segmentation(ctx, q, segment)) return 1;

	// descriptors: [EsaDev x nsub][int64 x nsub]
	const size_t desc_need = nsub * (sizeof(EsaDev) + sizeof(int64_t));
	if (ctx->desc_bytes < desc_need) {
		HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
		if (ctx->desc_dev) (void)andi_arena::dev_free(ctx->desc_dev);
		if (ctx->desc_host) host_pool::pinned_put(ctx->desc_host, ctx->desc_bytes);
		ctx->desc_dev = ctx->desc_host = nullptr;
		ctx->desc_bytes = 0;
		const size_t desc_cap = std::max<size_t>(desc_need, 4096); // (one size for small calls: the pool of pinned buffers hands it back)
		HIP_TRY(ctx, andi_arena::dev_malloc(&ctx->desc_dev, desc_cap));
		HIP_TRY(ctx, host_pool::pinned_get(&ctx->desc_host, desc_cap));
		ctx->desc_bytes = desc_cap;
	} else {
		HIP_TRY(ctx, hipEventSynchronize(ctx->desc_done)); // previous upload consumed
	}
	auto *h_esa = (EsaDev *)ctx->desc_host;
	auto *h_self = (int64_t *)(h_esa + nsub);
	uint64_t pairs = 0, nt = 0;
	int any_reference = 0;
	// the index builds must have finished: their flags decide which walk is exact.  (Only the builds this context
	// has queued since its last scan are waited for -- not whatever else is on the stream; subjects built by another
	// context are the caller's to have synchronised, as before.)
	if (ctx->builds_pending) {
		HIP_TRY(ctx, hipEventSynchronize(ctx->built));
		ctx->builds_pending = false;
	}
	for (size_t s = 0; s < nsub; ++s) {
		andi_hip_esa *e = subjects[s];
		if (!e || (!e->index_built && !e->ref_built)) {
			ctx->err = "andi_hip_scan_rows: subject index not built";
			return 1;
		}
		int mode = ANDI_MODE_PROBE;
		if (!e->index_built || e->h_flags[0] != 0 || andi_knob(KNOB_FORCE_REFERENCE)) {
			// a 10-mer table entry may span a separator: only the reference's
			// own walk reproduces get_match_cached there
			mode = ANDI_MODE_REFERENCE;
			if (!e->ref_built && andi_hip_esa_build(ctx, e)) return 1;
			ctx->acc.reference_subjects++;
			any_reference = 1;
		}
		if (e->index_built && e->h_flags[1]) {
			ctx->err = "andi_hip_scan_rows: a subject holds a byte outside {A,C,G,T,!,;,#}";
			return 1;
		}
		h_esa[s] = esa_view(e, mode);
		h_self[s] = self ? self[s] : -1;
		bool has_self = h_self[s] >= 0 && (size_t)h_self[s] < q->nq;
		pairs += q->nq - (has_self ? 1 : 0);
		nt += q->total_nt - (has_self ? q->len[(size_t)h_self[s]] : 0);
	}
	HIP_TRY(ctx, hipMemcpyAsync(ctx->desc_dev, ctx->desc_host, desc_need, hipMemcpyHostToDevice,
								ctx->stream));
	HIP_TRY(ctx, hipEventRecord(ctx->desc_done, ctx->stream));

	if (any_reference) routed = false;
	// scratch: per (subject, segment) two states and two count vectors
	bool adaptive = want_adaptive && !any_reference;
	uint32_t seg0 = segment / 2; // classes: 1/2, 1, 2, 4 times the call's segment length
	if (const char *e0 = andi_knob(KNOB_SEG0)) { // experiments: shortest segment of the adaptive classes
		if (atoi(e0) >= 64) seg0 = (uint32_t)atoi(e0);
	}
	uint64_t max_waves = 0; // adaptive: wavefronts (64 segments of one pair) if every pair had the shortest segments
	if (adaptive) {
		for (size_t s = 0; s < nsub; ++s)
			for (size_t i = 0; i < q->nq; ++i) {
				if (h_self[s] == (int64_t)i) continue;
				max_waves += ((q->len[i] + (uint64_t)seg0 - 1) / seg0 + 63) / 64;
			}
		// a pair occupies whole wavefronts: with queries of a few segments most lanes would idle, and the
		// scratch must stay a fraction of the device's memory -- one segment length for the call then
		uint64_t used = 0;
		for (size_t i = 0; i < q->nq; ++i) used += (q->len[i] + (uint64_t)seg0 - 1) / seg0;
		used *= nsub;
		size_t free_b = 0, total_b = 0;
		// (a routed call may need a second lane layout as large as the first for the pairs handed back: counted here, so that
		// a call that fits keeps fitting when that happens)
		const size_t want_b = (size_t)64 * max_waves * ANDI_SLOT_BYTES * (routed ? 2 : 1);
		const bool fits = max_waves < (1u << 26) && // (the device is asked only when the scratch would have to grow)
						  (want_b <= ctx->scratch_bytes || hipMemGetInfo(&free_b, &total_b) != hipSuccess || want_b < free_b / 2 + ctx->scratch_bytes);
		if (!fits || (10 * used < 7 * 64 * max_waves && !andi_knob(KNOB_FORCE_ADAPTIVE))) adaptive = false, max_waves = 0;
	}
	const size_t pairs_all = nsub * q->nq;
	if (routed && ensure_segmentation(ctx, q, coop_seg, true)) return 1;
	const size_t slots = adaptive ? (size_t)64 * max_waves : nsub * (size_t)q->total_segs;
	const size_t slots2 = routed ? nsub * (size_t)q->c_total_segs : 0; // (the wavefront kernel's layout, beside the lane scan's)
	const size_t need = (slots + slots2) * ANDI_SLOT_BYTES + 256 +
						(adaptive || routed ? pairs_all * 9 + 64 + (pairs_all / 1024 + 2) * 4 + 16 + nsub * 8 + 32 : 0);
	if (ctx->scratch_bytes < need) {
		HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
		if (ctx->scratch) (void)andi_arena::dev_free(ctx->scratch);
		ctx->scratch = nullptr;
		ctx->scratch_bytes = 0;
		HIP_TRY(ctx, andi_arena::dev_malloc(&ctx->scratch, need));
		ctx->scratch_bytes = need;
	}

	ScanArgs a;
	a.subjects = (const EsaDev *)ctx->desc_dev;
	a.self = (const int64_t *)((const EsaDev *)ctx->desc_dev + nsub);
	a.nsub = (uint32_t)nsub;
	a.qpool = q->pool, a.qnib = q->nib, a.qplanes = q->planes, a.qoff = q->d_off, a.qlen = q->d_len, a.qsep = q->d_sep, a.nq = (uint32_t)q->nq;
	a.qseg_start = q->d_qseg_start, a.seg2query = q->d_seg2query;
	a.total_segs = q->total_segs, a.seg = segment;
	char *p = (char *)ctx->scratch;
	auto carve = [&p](ScanArgs &x, size_t n) { // the per-slot arrays of a layout of n slots
		x.cold_exit = (ChainState *)p;
		p += n * sizeof(ChainState);
		x.true_exit = (ChainState *)p;
		p += n * sizeof(ChainState);
		x.used_entry = (ChainState *)p;
		p += n * sizeof(ChainState);
		x.cold_counts = (uint32_t *)p;
		p += n * 16 * sizeof(uint32_t);
		x.owned = (uint32_t *)p;
		p += n * 16 * sizeof(uint32_t);
		x.marks = (ColdMark *)p;
		p += n * ANDI_COLD_MARKS * sizeof(ColdMark);
		x.exit_p = (uint32_t *)p;
		p += n * sizeof(uint32_t);
		p = (char *)(((uintptr_t)p + 15) & ~(uintptr_t)15);
		x.restitch_count = (uint32_t *)p;
		x.restitch_round = 0;
		x.defer_count = x.restitch_count + 8;
		p += 64;
		x.defer_list = (unsigned long long *)p;
		p += n * sizeof(unsigned long long);
		x.first_pub = (unsigned long long *)p; // (k_lane_quad, per-pair segment lengths only)
		x.stretch_bad = (uint8_t *)p;          // (pass B: a byte per slot, while first_pub is idle)
		p += n * sizeof(unsigned long long);
	};
	carve(a, slots);
	a.adaptive = adaptive ? 1 : 0;
	a.seg0 = seg0, a.max_waves = (uint32_t)max_waves;
	a.max_class = 0; // long segments must not leave the device short of chains
	while (a.max_class < 3 && nt / ((uint64_t)seg0 << (a.max_class + 1)) >= ANDI_MIN_CHAINS) a.max_class++;
	a.pair_waves = (uint32_t *)p;
	a.pair_wave0 = a.pair_waves + pairs_all;
	a.pair_bsum = a.pair_wave0 + pairs_all + 1;
	a.pair_class = (uint8_t *)(a.pair_bsum + pairs_all / 1024 + 2);
	a.sub_cost = nullptr, a.sub_order = nullptr;
	if (routed) { // (the order in which pass A by wavefronts takes the subjects: scan.h)
		a.sub_cost = (float *)(((uintptr_t)(a.pair_class + pairs_all) + 15) & ~(uintptr_t)15);
		a.sub_order = (uint32_t *)(a.sub_cost + nsub);
	}
	{
		const char *f = andi_knob(KNOB_SEG_FACTOR);
		a.seg_factor = f && atoi(f) > 0 ? (uint32_t)atoi(f) : 16u; // measured best of 8/16/32 with seg0 = 2048
	}
	a.M = M_dev;
	a.fixups = ctx->d_fixups;
	a.any_reference = any_reference;
	{
		const char *qm = andi_knob(KNOB_QUAD_MATCH); // experiments: mean match length from which a pair goes to k_lane_quad (0: all, -1: none)
		a.quad_min_match = qm ? (uint32_t)atoi(qm) : 128u;
		a.quad_listed = 0;
		a.side_stream = ctx->side_stream, a.side_fork = ctx->side_fork, a.side_join = ctx->side_join;
		a.h_quad_waves = ctx->h_quad_waves;
		const char *kn = andi_knob(KNOB_KNOCK);
		a.knock = kn ? (uint32_t)atoi(kn) : 0u;
	}
	a.coop = coop && !a.adaptive;
	a.pool_scratch = nullptr, a.pool_ticket = nullptr, a.pool_waves = 0, a.pool_bytes = 0;
	a.pool_maxchunks = a.pool_hc = 0, a.pool_first = 0, a.pool_use = 0;
	{
		const char *pm = andi_knob(KNOB_POOL_MATCH); // (experiments: mean sampled match from which a routed pair's wavefront kernel is k_pool_cold)
		a.pool_match = pm && atoi(pm) >= 0 ? (uint32_t)atoi(pm) : 48u;
	}
	// Pooled walks (k_pool_cold): the scratch of the resident wavefronts -- 0.8 GB on a 256-CU part, a mapping of its own -- is
	// taken only by a call that is going to run that kernel (andi_coop_wants_pool: known behind the look at the layout in a
	// routed call), from the device's idle one if a destroyed context left it (host_pool), once per context; a context whose
	// attempt failed does not try again (the windows then stay in LDS: k_coop_cold).
	auto give_pool_scratch = [&](ScanArgs &x) {
		if (!andi_coop_wants_pool(x)) return;
		if (!ctx->pool_scratch && !ctx->pool_failed) {
			uint32_t waves = 0;
			const size_t bytes = andi_pool_scratch_bytes(ctx->device, &waves);
			if (bytes && (ctx->pool_scratch = host_pool::scratch_get(ctx->device, bytes)))
				ctx->pool_waves = waves, ctx->pool_bytes = bytes - 4096;
			else
				ctx->pool_failed = true;
		}
		if (!ctx->pool_scratch) return;
		x.pool_ticket = (uint32_t *)ctx->pool_scratch, x.pool_scratch = (char *)ctx->pool_scratch + 4096, x.pool_waves = ctx->pool_waves, x.pool_bytes = ctx->pool_bytes;
	};
	a.route = routed ? ANDI_LAYOUT_LANES : 0, a.route_seg = coop_seg, a.route_nt = ctx->d_route;
	uint32_t longest_q = 0;
	for (size_t i = 0; i < q->nq; ++i) longest_q = std::max(longest_q, (uint32_t)q->len[i]);
	a.reduce_threads = (longest_q + (a.adaptive ? seg0 : segment) - 1) / (a.adaptive ? seg0 : segment) <= 64 ? 64u : 0u;
	{
		const char *rs = andi_knob(KNOB_ROUTE_SMALL); // (experiments: log2 of the size below which a call is small)
		const uint64_t small_nt = rs && atoi(rs) > 0 && atoi(rs) < 63 ? 1ull << atoi(rs) : ANDI_ROUTE_SMALL_NT;
		a.route_all_few = q->total_nt * (uint64_t)nsub < small_nt ? 1u : 0u;
	}
	{
		const char *gu = andi_knob(KNOB_COOP_GIVEUP); // (tests: hand pairs back early, so that the second lane layout runs)
		a.route_giveup = gu && atoi(gu) > 0 ? (uint32_t)atoi(gu) : a.route_all_few ? 256u : 1024u; // (small calls route pairs the sampling cannot judge: a lower limit)
		const char *sm = andi_knob(KNOB_ROUTE_SOFT); // (experiments)
		a.route_soft_match = sm && atoi(sm) > 0 ? (uint32_t)atoi(sm) : 512u; // (128 = k_lane_quad's class: tree-structured set 38.1 -> 39.4 % of the roofline at 512, C3-like 45.6 -> 48.1 %, C4 shape the same)
	}
	a.exact_equal = (model == ANDI_M_LOGDET || model == ANDI_M_ANI) ? 1 : 0; // src/model.c:247

	if (routed) {
		// The call's pairs are routed (scan.h): the wavefront kernel's layout b beside the lane scan's a.  The pairs are
		// sampled and routed; pass A by wavefronts (on a stream of its own) runs beside the lane scan's kernels; the pairs
		// it handed back -- rare: the host looks -- get a second lane layout a2; passes B and C once per layout.
		ScanArgs b = a;
		b.adaptive = 0, b.coop = 1, b.route = ANDI_LAYOUT_COOP;
		b.qseg_start = q->c_qseg_start, b.seg2query = q->c_seg2query, b.total_segs = q->c_total_segs, b.seg = coop_seg;
		b.reduce_threads = (longest_q + coop_seg - 1) / coop_seg <= 64 ? 64u : 0u;
		p = (char *)(a.sub_order + nsub);
		p = (char *)(((uintptr_t)p + 15) & ~(uintptr_t)15);
		carve(b, slots2);
		hipError_t e;
		// every error exit below first waits for the streams this branch forks work onto: their kernels read the scratch
		// the next call may regrow, and the context's teardown waits for ctx->stream only
		auto bail = [&](const char *what, hipError_t err) {
			(void)hipStreamSynchronize(ctx->coop_stream);
			(void)hipStreamSynchronize(ctx->side_stream);
			(void)hipStreamSynchronize(ctx->stream);
			return fail(ctx, what, err);
		};
		{
			Timed t(ctx, 2);
			e = hipMemsetAsync(b.restitch_count, 0, 16 * sizeof(uint32_t), ctx->stream);
			if (e == hipSuccess) e = andi_launch_pair_layout(a, ctx->stream);
			t.stop();
			if (e != hipSuccess) return bail("scan layout", e);
		}
		if (andi_knob(KNOB_DEBUG_STITCH)) { // diagnostics: how the pairs were routed
			std::vector<uint8_t> cls(pairs_all);
			(void)hipStreamSynchronize(ctx->stream);
			(void)hipMemcpy(cls.data(), a.pair_class, pairs_all, hipMemcpyDeviceToHost);
			size_t n_coop = 0, n_quad = 0, n_other = 0;
			for (size_t i = 0; i < pairs_all; ++i)
				if (h_self[i / q->nq] != (int64_t)(i % q->nq)) (cls[i] & ANDI_ROUTE_COOP ? n_coop : cls[i] & 0x80u ? n_quad : n_other)++;
			fprintf(stderr, "route: %zu pairs by wavefronts, %zu k_lane_quad's class, %zu other lanes (unrelated stretches suspected / short query / many pairs far apart)\n", n_coop, n_quad, n_other);
		}
		bool any_left = false;
		{
			Timed t(ctx, 1);
			// Which of the two goes first: the lane scan's kernels where its pairs are few -- behind the wavefront kernel a
			// handful of lane blocks (four wavefronts and their LDS on one CU at once) found no place until that kernel's
			// tail and ended 0.2 ms after everything else --, the wavefront kernel where they are many (the tree-structured
			// set, the C4 shape: 0.4 and 1.5 ms the other way round).  The host looks at the layout (one word).
			// (Small calls do not look: the wavefront kernel first, the lane layout's passes whether it has pairs or not --
			// a look costs them 40 us of their few hundred.)
			const bool look = !a.route_all_few;
			for (int k = 0; k < 16; ++k) ctx->h_any_left[1 + k] = 0;
			ctx->h_any_left[1 + ANDI_LANE_WAVES] = 1;
			e = hipSuccess;
			if (look) e = hipMemcpyAsync(ctx->h_any_left + 1, a.restitch_count, 16 * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream);
			if (look && e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
			const bool lanes_first = look && e == hipSuccess && (uint64_t)ctx->h_any_left[1 + ANDI_LANE_WAVES] * 20 < ctx->h_any_left[1 + ANDI_ALL_WAVES];
			// which wavefront kernel: the pooled one where the pairs that suit it hold at least half of the segments (scan.h)
			b.pool_use = look && e == hipSuccess && 2 * (uint64_t)ctx->h_any_left[1 + ANDI_POOL_SEGS] >= ctx->h_any_left[1 + ANDI_COOP_SEGS] && ctx->h_any_left[1 + ANDI_COOP_SEGS] != 0;
			give_pool_scratch(b);
			if (e == hipSuccess) e = hipEventRecord(ctx->coop_fork, ctx->stream);
			if (e == hipSuccess) e = hipStreamWaitEvent(ctx->coop_stream, ctx->coop_fork, 0);
			if (e == hipSuccess && lanes_first) e = andi_launch_scan_cold(a, ctx->stream);
			if (e == hipSuccess) e = andi_launch_coop_cold(b, ctx->coop_stream);
			if (andi_coop_will_pool(b)) ctx->acc.pool_calls++;
			if (e == hipSuccess) e = hipEventRecord(ctx->coop_join, ctx->coop_stream);
			if (e == hipSuccess && !lanes_first) e = andi_launch_scan_cold(a, ctx->stream);
			if (e == hipSuccess) e = hipStreamWaitEvent(ctx->stream, ctx->coop_join, 0);
			if (e == hipSuccess) e = hipMemcpyAsync(ctx->h_any_left, b.restitch_count + ANDI_ROUTE_ANY_LEFT, sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream);
			if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
			if (e != hipSuccess) return bail("scan pass A", e);
			any_left = ctx->h_any_left[0] != 0;
			const bool any_lanes = ctx->h_any_left[1 + ANDI_LANE_WAVES] != 0; // (no pair in the lane layout: its passes B and C have nothing to do)
			ScanArgs a2 = a;
			if (any_left) { // the pairs handed back: a lane layout of their own (as large as the first at most)
				const size_t need2 = slots * ANDI_SLOT_BYTES + 256 + pairs_all * 9 + 64 + (pairs_all / 1024 + 2) * 4 + 16;
				if (ctx->scratch2_bytes < need2) {
					if (ctx->scratch2) (void)andi_arena::dev_free(ctx->scratch2);
					ctx->scratch2 = nullptr, ctx->scratch2_bytes = 0;
					e = andi_arena::dev_malloc(&ctx->scratch2, need2);
					if (e != hipSuccess) return bail("scratch of the second lane layout", e);
					ctx->scratch2_bytes = need2;
				}
				p = (char *)ctx->scratch2;
				carve(a2, slots);
				a2.route = ANDI_LAYOUT_LANES2;
				a2.pair_waves = (uint32_t *)p;
				a2.pair_wave0 = a2.pair_waves + pairs_all;
				a2.pair_bsum = a2.pair_wave0 + pairs_all + 1;
				a2.side_stream = nullptr; // (its own kernels one after the other: it runs on the side stream itself, below)
			}
			t.stop();
			// Passes B and C once per layout, side by side (each is a chain of small launches); the pairs handed back take
			// their pass A at the head of their chain.
			Timed t2(ctx, 2);
			e = hipEventRecord(ctx->l2_fork, ctx->stream);
			if (e == hipSuccess) e = hipStreamWaitEvent(ctx->coop_stream, ctx->l2_fork, 0);
			if (e == hipSuccess) e = andi_launch_scan_stitch(b, ctx->coop_stream);
			if (e == hipSuccess) e = andi_launch_scan_reduce(b, ctx->coop_stream);
			if (e == hipSuccess) e = hipEventRecord(ctx->coop_join, ctx->coop_stream);
			if (e == hipSuccess && any_left) {
				e = hipStreamWaitEvent(ctx->side_stream, ctx->l2_fork, 0);
				if (e == hipSuccess) e = andi_launch_pair_leftover(a2, ctx->side_stream);
				if (e == hipSuccess) e = andi_launch_scan_cold(a2, ctx->side_stream);
				if (e == hipSuccess) e = andi_launch_scan_stitch(a2, ctx->side_stream);
				if (e == hipSuccess) e = andi_launch_scan_reduce(a2, ctx->side_stream);
				if (e == hipSuccess) e = hipEventRecord(ctx->l2_join, ctx->side_stream);
			}
			if (e == hipSuccess && any_lanes) e = andi_launch_scan_stitch(a, ctx->stream);
			if (e == hipSuccess && any_lanes) e = andi_launch_scan_reduce(a, ctx->stream);
			if (e == hipSuccess) e = hipStreamWaitEvent(ctx->stream, ctx->coop_join, 0);
			if (e == hipSuccess && any_left) e = hipStreamWaitEvent(ctx->stream, ctx->l2_join, 0);
			if (e == hipSuccess) e = andi_launch_route_count(a, ctx->stream);
			t2.stop();
			if (e != hipSuccess) return bail("scan passes B/C", e);
		}
		ctx->acc.coop_calls++;
		ctx->acc.routed_calls++;
	} else {
		if (a.adaptive) {
			Timed t(ctx, 2);
			hipError_t e = andi_launch_pair_layout(a, ctx->stream);
			t.stop();
			if (e != hipSuccess) return fail(ctx, "scan layout", e);
		}
		give_pool_scratch(a); // (a call whose pass A is the wavefront kernel's for every pair: ANDI_COOP=n, tiny calls)
		{
			Timed t(ctx, 1);
			hipError_t e = andi_launch_scan_cold(a, ctx->stream);
			t.stop();
			if (e != hipSuccess) return fail(ctx, "scan pass A", e);
			if (a.coop) ctx->acc.coop_calls++;
			if (a.coop && andi_coop_will_pool(a)) ctx->acc.pool_calls++;
		}
		Timed t(ctx, 2);
		hipError_t e = andi_launch_scan_stitch(a, ctx->stream);
		if (e == hipSuccess) e = andi_launch_scan_reduce(a, ctx->stream);
		t.stop();
		if (e != hipSuccess) return fail(ctx, "scan passes B/C", e);
	}
	if (andi_knob(KNOB_DEBUG_STITCH)) { // diagnostics: segments stitched again per round, length of the last stage's list
		uint32_t h[16];
		(void)hipStreamSynchronize(ctx->stream);
		(void)hipMemcpy(h, a.restitch_count, sizeof h, hipMemcpyDeviceToHost);
		fprintf(stderr, "stitch: %zu slots; true chains that left on their own %u; stitched again in rounds: %u %u %u; last list %u\n", slots,
				h[ANDI_RESTITCH_ROUNDS], h[0], h[1], h[2], h[8]);
	}
	(adaptive ? ctx->acc.adaptive_calls : ctx->acc.uniform_calls)++;
	ctx->acc.scan_pairs += pairs;
	ctx->acc.scan_query_nt += nt;
	return 0;
}

// ------------------------------------------------------------------ the measured copy ceiling (bench.py: roofline.measured_copy_GBps)
namespace {
__global__ __launch_bounds__(256) void k_stream_copy(const uint4 *__restrict__ src, uint4 *__restrict__ dst, size_t n16) {
	const size_t stride = (size_t)gridDim.x * 256;
	for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += stride) {
		uint4 v;
		v.x = __builtin_nontemporal_load(&src[i].x), v.y = __builtin_nontemporal_load(&src[i].y);
		v.z = __builtin_nontemporal_load(&src[i].z), v.w = __builtin_nontemporal_load(&src[i].w);
		__builtin_nontemporal_store(v.x, &dst[i].x), __builtin_nontemporal_store(v.y, &dst[i].y);
		__builtin_nontemporal_store(v.z, &dst[i].z), __builtin_nontemporal_store(v.w, &dst[i].w);
	}
}
} // namespace

int andi_hip_copy_ceiling(andi_hip_ctx *ctx, size_t bytes, int reps, double *gbps) {
	if (!ctx || !gbps || bytes < 4096 || reps < 1) {
		if (ctx) ctx->err = "andi_hip_copy_ceiling: bad arguments";
		return 1;
	}
	HIP_TRY(ctx, hipSetDevice(ctx->device));
	const size_t n16 = bytes / 16;
	uint4 *src = nullptr, *dst = nullptr;
	hipEvent_t e0 = nullptr, e1 = nullptr;
	hipError_t err = hipMalloc((void **)&src, n16 * 16);
	if (err == hipSuccess) err = hipMalloc((void **)&dst, n16 * 16);
	if (err == hipSuccess) err = hipMemsetAsync(src, 1, n16 * 16, ctx->stream);
	if (err == hipSuccess) err = hipEventCreate(&e0);
	if (err == hipSuccess) err = hipEventCreate(&e1);
	int cus = 256;
	(void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, ctx->device);
	const unsigned grid = (unsigned)std::min<size_t>((n16 + 255) / 256, (size_t)cus * 32);
	float ms = 0;
	if (err == hipSuccess) {
		k_stream_copy<<<grid, 256, 0, ctx->stream>>>(src, dst, n16); // (untimed: first touch)
		err = hipEventRecord(e0, ctx->stream);
		for (int r = 0; r < reps && err == hipSuccess; ++r) k_stream_copy<<<grid, 256, 0, ctx->stream>>>(src, dst, n16);
		if (err == hipSuccess) err = hipEventRecord(e1, ctx->stream);
		if (err == hipSuccess) err = hipEventSynchronize(e1);
		if (err == hipSuccess) err = hipEventElapsedTime(&ms, e0, e1);
	}
	if (e0) (void)hipEventDestroy(e0);
	if (e1) (void)hipEventDestroy(e1);
	if (src) (void)hipFree(src);
	if (dst) (void)hipFree(dst);
	if (err != hipSuccess) return fail(ctx, "andi_hip_copy_ceiling", err);
	*gbps = ms > 0 ? 2.0 * (double)(n16 * 16) * reps / (ms * 1e-3) / 1e9 : 0.0;
	return 0;
}

// ------------------------------------------------------------------ bootstrap
int andi_hip_bootstrap(andi_hip_ctx *ctx, const andi_hip_model *M, size_t n, uint64_t seed,
					   size_t replicates, andi_hip_model *B) {
	if (!ctx || !M || !B || n == 0 || n > 65535) {
		if (ctx) ctx->err = "andi_hip_bootstrap: bad arguments";
		return 1;
	}
	if (replicates == 0) return 0;
	HIP_TRY(ctx, hipSetDevice(ctx->device));
	const size_t one = n * n * sizeof(andi_hip_model);
	andi_hip_model *dM = nullptr, *dB = nullptr;
	hipError_t e = hipMalloc((void **)&dM, one);
	if (e == hipSuccess) e = hipMalloc((void **)&dB, one * replicates);
	if (e == hipSuccess) e = hipMemcpyAsync(dM, M, one, hipMemcpyHostToDevice, ctx->stream);
	if (e == hipSuccess)
		e = andi_launch_bootstrap(dM, dB, (uint32_t)n, (uint32_t)replicates, seed, ctx->stream);
	if (e == hipSuccess) e = hipMemcpyAsync(B, dB, one * replicates, hipMemcpyDeviceToHost, ctx->stream);
	if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
	(void)andi_arena::dev_free(dM);
	(void)andi_arena::dev_free(dB);
	if (e != hipSuccess) return fail(ctx, "andi_hip_bootstrap", e);
	return 0;
}

void andi_hip_reload_knobs(void) { g_knobs.store(new KnobSnapshot(), std::memory_order_release); }

int andi_hip_timings_get(andi_hip_ctx *ctx, andi_hip_timings *t) {
	if (!ctx || !t) return 1;
	HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
	resolve_events(ctx);
	unsigned long long fx = 0, rt[4] = {0, 0, 0, 0};
	HIP_TRY(ctx, hipMemcpy(&fx, ctx->d_fixups, sizeof fx, hipMemcpyDeviceToHost));
	HIP_TRY(ctx, hipMemcpy(rt, ctx->d_route, sizeof rt, hipMemcpyDeviceToHost));
	ctx->acc.fixups = fx;
	ctx->acc.coop_query_nt = rt[0], ctx->acc.lane_query_nt = rt[1], ctx->acc.coop_fallbacks = rt[2];
	*t = ctx->acc;
	return 0;
}

void andi_hip_timings_reset(andi_hip_ctx *ctx) {
	if (!ctx) return;
	(void)hipStreamSynchronize(ctx->stream);
	resolve_events(ctx);
	(void)hipMemset(ctx->d_fixups, 0, sizeof(unsigned long long));
	(void)hipMemset(ctx->d_route, 0, 4 * sizeof(unsigned long long));
	ctx->acc = andi_hip_timings{};
}

} // extern "C"

// ------------------------------------------------------------------ the seam
// distMatrix / distMatrixLM, src/dist_hack.h:34-96: for every subject build the
// index and compare every other sequence against it.
//
// The rows of the matrix (one subject against every query) are independent given
// the subject's index.  Every device of the call owns a contiguous block of rows
// (block sizes differ by at most one) and is driven by one host thread with its own
// context: all queries staged once, a set of subject slots reused batch after batch,
// its rows kept in HBM.  A pool of host threads shared by all devices prepares RS and
// the suffix array (seq_subject_init + esa_init_SA) in the order the devices will
// ask for them.  The one exchange of the job is the gather of the row blocks on the
// first device -- RCCL send/recv over xGMI, every peer on its own link -- followed by
// one copy of the matrix to the host.  (One device, several contexts on one device,
// or no usable RCCL: every block is copied to the host matrix directly.)
namespace {
struct Prepared {
	size_t idx = 0;
	char *RS = nullptr;
	size_t n = 0, thr = 0;
	std::vector<int32_t> SA;
	int rc = 0;
};

// librccl is loaded when a call first spans several devices: single-device users (and processes
// that carry another copy of RCCL, like PyTorch's) never touch it
struct Rccl {
	void *lib = nullptr;
	ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
	ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
	ncclResult_t (*GroupStart)() = nullptr;
	ncclResult_t (*GroupEnd)() = nullptr;
	ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
	ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
	const char *(*GetErrorString)(ncclResult_t) = nullptr;
	bool ok = false;
};

Rccl &rccl() {
	static Rccl r;
	static std::once_flag once;
	std::call_once(once, [] {
		const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
		for (const char *nm : names)
			if ((r.lib = dlopen(nm, RTLD_NOW | RTLD_LOCAL))) break;
		if (!r.lib) return;
		r.CommInitAll = (decltype(r.CommInitAll))dlsym(r.lib, "ncclCommInitAll");
		r.CommDestroy = (decltype(r.CommDestroy))dlsym(r.lib, "ncclCommDestroy");
		r.GroupStart = (decltype(r.GroupStart))dlsym(r.lib, "ncclGroupStart");
		r.GroupEnd = (decltype(r.GroupEnd))dlsym(r.lib, "ncclGroupEnd");
		r.Send = (decltype(r.Send))dlsym(r.lib, "ncclSend");
		r.Recv = (decltype(r.Recv))dlsym(r.lib, "ncclRecv");
		r.GetErrorString = (decltype(r.GetErrorString))dlsym(r.lib, "ncclGetErrorString");
		r.ok = r.CommInitAll && r.CommDestroy && r.GroupStart && r.GroupEnd && r.Send && r.Recv && r.GetErrorString;
	});
	return r;
}

thread_local char g_last_gather[200] = "none"; // how the calling thread's last andi_hip_dist_matrix call collected its rows (diagnostic)

void row_block(size_t total, size_t parts, size_t k, size_t &first, size_t &last) { // as andi_amd/shard.py: row_block
	const size_t base = total / parts, extra = total % parts;
	first = k * base + std::min(k, extra);
	last = first + base + (k < extra ? 1 : 0);
}
} // namespace

extern "C" {

const char *andi_hip_last_gather(void) {
	return g_last_gather;
}

void andi_hip_row_block(size_t total, size_t parts, size_t k, size_t *first, size_t *last) {
	size_t f = 0, l = 0;
	if (parts && k < parts) row_block(total, parts, k, f, l);
	if (first) *first = f;
	if (last) *last = l;
}

int andi_hip_dist_matrix(andi_hip_model *M, const andi_hip_seq *seqs, size_t n,
						 const andi_hip_opts *opts_in, char *errbuf, size_t errlen) {
	if (!M || !seqs || n == 0) {
		set_err(errbuf, errlen, "andi_hip_dist_matrix: bad arguments");
		return 1;
	}
	andi_hip_opts o;
	if (opts_in) {
		o = *opts_in;
	} else {
		andi_hip_default_opts(&o);
	}
	for (size_t i = 0; i < n; ++i) {
		if (!seqs[i].seq || seqs[i].len == 0) {
			set_err(errbuf, errlen, "sequence %zu is empty", i); // src/andi.c:302-304
			return 1;
		}
		if (seqs[i].len > (size_t)(INT32_MAX - 1) / 2) { // src/andi.c:296-300
			set_err(errbuf, errlen, "sequence %zu is too long. The technical limit is %zu.", i,
					(size_t)(INT32_MAX - 1) / 2);
			return 1;
		}
	}

	// ---- the devices of the call
	std::vector<int> devs;
	{
		int visible = 0;
		hipError_t e = hipGetDeviceCount(&visible);
		if (e != hipSuccess || visible <= 0) {
			set_err(errbuf, errlen, "no HIP device available (%s); the anchor-distance engine has no CPU path",
					e != hipSuccess ? hipGetErrorString(e) : "device count 0");
			return 1;
		}
		if (o.devices && o.num_gpus > 0) {
			devs.assign(o.devices, o.devices + o.num_gpus);
		} else {
			const int want = o.num_gpus < 0 ? visible - o.device : (o.num_gpus == 0 ? 1 : o.num_gpus);
			for (int k = 0; k < want; ++k) devs.push_back(o.device + k);
		}
		for (int d : devs)
			if (d < 0 || d >= visible) {
				set_err(errbuf, errlen, "HIP device %d out of range (have %d)", d, visible);
				return 1;
			}
		if (devs.empty()) {
			set_err(errbuf, errlen, "andi_hip_dist_matrix: no device selected");
			return 1;
		}
		if (devs.size() > n) devs.resize(n); // at least one row each
	}
	const size_t ndev = devs.size();
	bool distinct = true;
	for (size_t a = 0; a < ndev; ++a)
		for (size_t b = a + 1; b < ndev; ++b) distinct = distinct && devs[a] != devs[b];
	const char *gather_env = andi_knob(KNOB_GATHER);
	// RCCL gather: several distinct devices (or forced, to exercise the path on the devices there are -- with contexts that
	// share a device the communicators cannot be made: the route's fallback, every block copied from HBM directly, runs),
	// and the matrix fits next to the rest
	bool use_rccl = (ndev > 1 && distinct && !(gather_env && !strcmp(gather_env, "direct"))) ||
					(gather_env && !strcmp(gather_env, "rccl"));
	if (use_rccl && n * n * sizeof(andi_hip_model) > ((size_t)32 << 30)) use_rccl = false;
	if (use_rccl && !rccl().ok) use_rccl = false;

	// ---- shared host pool: subject preparation + suffix sorting (the role of the OpenMP
	// subject loop, src/dist_hack.h:46-52), in the order the devices consume, bounded look-ahead
	size_t longest = 0;
	for (size_t i = 0; i < n; ++i) longest = std::max(longest, seqs[i].len);
	const size_t rs_cap = 2 * longest + 1;
	std::vector<size_t> first(ndev), last(ndev);
	size_t max_rows = 0;
	for (size_t d = 0; d < ndev; ++d) {
		row_block(n, ndev, d, first[d], last[d]);
		max_rows = std::max(max_rows, last[d] - first[d]);
	}
	std::vector<size_t> order; // subjects in the order they are needed
	order.reserve(n);
	for (size_t k = 0; k < max_rows; ++k)
		for (size_t d = 0; d < ndev; ++d)
			if (first[d] + k < last[d]) order.push_back(first[d] + k);

	int threads = o.host_threads > 0 ? o.host_threads : (int)std::thread::hardware_concurrency();
	if (threads < 1) threads = 1;
	if ((size_t)threads > n) threads = (int)n;
	// Every subject is also a query, and the queries are staged in HBM before the first batch: unless the suffix arrays are
	// the host's (sa_on_host: the sorter needs RS where it runs), a device writes RS = revcomp(S) '#' S into the subject's
	// slot itself from its query pool (esa_from_query) and the host computes only min_anchor_length from the device's G+C
	// counts -- no host pass over the sequences, no second upload of what is already resident (round 5's trace of the bench
	// set's warm call: host pool 5.5 ms + subject uploads 12.9 ms of 54).
	const bool dev_prep = !o.sa_on_host;
	const size_t batch_max = o.low_memory ? 1 : 8;
	const size_t window = (size_t)threads + ndev * batch_max + 1;

	std::mutex mu;
	std::condition_variable cv;
	std::deque<Prepared *> ready; // any order
	std::atomic<size_t> next{0};
	size_t consumed = 0; // subjects taken by the devices, guarded by mu
	bool abort_flag = false;
	std::string first_error;
	size_t rows_done = 0; // guarded by mu (progress)

	auto fail_all = [&](const std::string &msg) {
		std::lock_guard<std::mutex> lk(mu);
		if (!abort_flag) first_error = msg;
		abort_flag = true;
		cv.notify_all();
	};

	auto worker = [&]() {
		for (;;) {
			const size_t pos = next.fetch_add(1);
			if (pos >= n) return;
			{
				std::unique_lock<std::mutex> lk(mu);
				cv.wait(lk, [&] { return abort_flag || pos < consumed + window; });
				if (abort_flag) return;
			}
			const size_t i = order[pos];
			Prepared *p = nullptr;
			try {
				p = new Prepared;
				p->idx = i;
				double gc;
				p->rc = andi_hip_subject_prepare(seqs[i].seq, seqs[i].len, o.p_value, &p->RS, &p->n, &gc, &p->thr);
				if (!p->rc && o.sa_on_host) {
					p->SA.resize(p->n);
					p->rc = andi_hip_suffix_array((const unsigned char *)p->RS, p->SA.data(), (int32_t)p->n);
				}
			} catch (...) { // out of memory: report it as the reference does (src/dist_hack.h:53)
				if (p) {
					andi_hip_free(p->RS);
					delete p;
				}
				char msg[96];
				snprintf(msg, sizeof msg, "Failed to create index for sequence %zu.", i);
				fail_all(msg);
				return;
			}
			{
				std::lock_guard<std::mutex> lk(mu);
				ready.push_back(p);
			}
			cv.notify_all();
		}
	};

	auto take = [&](size_t i) -> Prepared * { // blocks until subject i is prepared; null if the call was aborted
		std::unique_lock<std::mutex> lk(mu);
		Prepared *p = nullptr;
		cv.wait(lk, [&] {
			if (abort_flag) return true;
			for (auto *c : ready)
				if (c->idx == i) return true;
			return false;
		});
		if (abort_flag) return nullptr;
		for (auto it = ready.begin(); it != ready.end(); ++it)
			if ((*it)->idx == i) {
				p = *it;
				ready.erase(it);
				break;
			}
		return p;
	};

	// ---- one driver per device.  Two stages, two contexts (streams) and two sets of subject slots per device: while the
	// scan of one batch of subjects runs, a second thread stages the next -- upload, suffix arrays, index builds -- as
	// the reference's threads build one subject's index while others scan (src/dist_hack.h:46-52).
	struct Dev {
		andi_hip_ctx *ctx = nullptr;  // scans, row copies
		andi_hip_ctx *prep = nullptr; // suffix arrays, index builds
		andi_hip_ctx *up = nullptr;   // uploads (a thread and a stream of their own: the copies of batch k + 1 run beside the sorts of batch k)
		std::vector<andi_hip_ctx *> sorters; // suffix sorts of a batch's subjects side by side (streams and workspaces of their own)
		andi_hip_queries *Q = nullptr;
		andi_hip_model *d_rows = nullptr; // rccl: the whole row block; direct: one batch of rows
		size_t pinned_bytes = 0;
		char *pinned = nullptr;           // staging buffers for RS (two: one is filled while the other's copy runs): uploads from pinned memory go through the DMA engines, beside a scan
		hipEvent_t pinned_free[2] = {nullptr, nullptr};
		std::vector<andi_hip_esa *> slots; // sets x batch
	};
	std::vector<Dev> dv(ndev);

	const bool trace = andi_knob(KNOB_E2E_TRACE) != nullptr; // diagnostics: where the call's wall time goes (device 0's driver)
	auto now_ms = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
	const double t_call = now_ms();
	// the queries as 4-bit symbols, packed once for all devices while their contexts come up (ANDI_QUERIES_BYTES: every
	// device uploads the bytes and packs them itself, as up to round 3)
	// One device: the bytes as they lie (measured on one GPU, same box: C4's queries 0.51 s as bytes, 0.10 s packed -- but the
	// pack's pass over the host's memory and the release of its copy gave the 0.3 s back; C5 was slower packed).
	const bool pack_on_host = andi_knob(KNOB_QUERIES_BYTES) == nullptr && (ndev > 1 || andi_knob(KNOB_QUERIES_PACKED) != nullptr);
	PackedQueries PQ;
	PQ.users.store((int)ndev);
	std::mutex pq_mu;
	std::condition_variable pq_cv;
	bool pq_done = false;
	int pq_rc = 0;
	std::thread packer;
	if (pack_on_host)
		packer = std::thread([&] {
			const int rc = pack_queries_host(seqs, n, threads, PQ);
			std::lock_guard<std::mutex> lk(pq_mu);
			pq_rc = rc, pq_done = true;
			pq_cv.notify_all();
		});
	auto drive = [&](size_t d) {
		Dev &D = dv[d];
		char eb[256] = "";
		double t_last = now_ms(), acc_wait = 0, acc_scan = 0, acc_copy = 0;
		auto lap = [&](double &acc) {
			const double t = now_ms();
			acc += t - t_last, t_last = t;
		};
		double t_ctx = 0, t_queries = 0, t_slots = 0;
		auto bail = [&](const char *what, andi_hip_ctx *cx) {
			char msg[512];
			snprintf(msg, sizeof msg, "%s (device %d): %s", what, devs[d], cx ? andi_hip_last_error(cx) : eb);
			fail_all(msg);
		};
		if (andi_hip_ctx_create(&D.ctx, devs[d], eb, sizeof eb)) return bail("creating a context", nullptr);
		if (ctx_create(&D.prep, devs[d], eb, sizeof eb, true)) return bail("creating a context", nullptr);
		if (!dev_prep) {
			if (ctx_create(&D.up, devs[d], eb, sizeof eb, true)) return bail("creating a context", nullptr);
			andi_hip_ctx_expect_queries(D.up, n - 1);
		}
		// A suffix sort is two dozen launches with two or three host round trips between them (sa_device.hip): 0.73 ms per
		// 9.8 M characters of which the device is busy half.  The subjects of a batch are sorted by up to four host threads,
		// each with a stream and a workspace of its own, so one subject's small launches and waits hide behind another's
		// radix passes.
		size_t sort_width = dev_prep && !o.low_memory ? std::min<size_t>(4, std::min(batch_max, last[d] - first[d])) : 1;
		if (const char *sw = andi_knob(KNOB_SORT_WIDTH)) // (experiments)
			if (atoi(sw) >= 1 && atoi(sw) <= 8) sort_width = std::min<size_t>((size_t)atoi(sw), std::min(batch_max, last[d] - first[d]));
		// (a workspace of 45 bytes per character each: together at most one chunk of the arena -- eight of them for 9.8 M characters pushed a
		// 29-genome call past the 8 GiB the arena keeps from call to call, and every call paid the driver for its chunks again: 37 -> 177 ms;
		// two sorters measured like four, profiles/r07_seam/)
		while (sort_width > 1 && andi_sa_device_workspace((int32_t)rs_cap) * sort_width > ((size_t)2 << 30)) --sort_width;
		for (size_t w = 1; w < sort_width; ++w) {
			andi_hip_ctx *cx = nullptr;
			if (ctx_create(&cx, devs[d], eb, sizeof eb, true)) return bail("creating a context", nullptr);
			andi_hip_ctx_expect_queries(cx, n - 1);
			D.sorters.push_back(cx);
		}
		andi_hip_ctx_expect_queries(D.ctx, n - 1);
		andi_hip_ctx_expect_queries(D.prep, n - 1);
		lap(t_ctx);
		const size_t rows = last[d] - first[d];
		// Subject slots: device buffers sized for the longest genome, reused batch after batch (no
		// allocation inside the loop).  Several subjects per scan call keep the GPU filled; low_memory
		// keeps one index resident at a time, which is what distMatrixLM trades (src/dist_hack.h:14-16).
		size_t batch = batch_max < rows ? batch_max : rows;
		auto sets_for = [&](size_t bt) { // (low_memory: one index resident at a time)
			const size_t nb = (rows + bt - 1) / bt;
			return o.low_memory ? (size_t)1 : (nb > 2 && !dev_prep ? (size_t)3 : (nb > 1 ? (size_t)2 : (size_t)1)); // (the third set is the uploads')
		};
		{
			size_t free_b = 0, total_b = 0;
			if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
				// a slot: text + padding, suffix array, the records of the device sorter, the packed text twice, the probe table
				const size_t per_slot = 14 * rs_cap + ((size_t)8 << (2 * pick_deep_k(rs_cap, D.ctx->queries_hint))) + (1 << 20);
				while (batch > 1 && sets_for(batch) * batch * per_slot > free_b / (2 * ndev)) batch /= 2; // (as many sets as the batches will really have)
			}
		}
		const size_t nbatches = (rows + batch - 1) / batch;
		// Three sets of slots: while batch k is scanned, batch k + 2 is uploaded (no compute units needed) and batch k + 1
		// is ready; the device's COMPUTE alternates strictly -- suffix sorts and index builds of batch k + 1, then the scan
		// of batch k -- because side by side the staging kernels starve behind the workgroups of a scan that fills the
		// device (sorts of 8 subjects: 6 ms alone, 38 ms beside a scan, on a high-priority stream as on a plain one).
		const size_t sets = sets_for(batch);
		D.slots.assign(sets * batch, nullptr);
		if (pack_on_host) {
			{
				std::unique_lock<std::mutex> lk(pq_mu);
				pq_cv.wait(lk, [&] { return pq_done; });
			}
			if (pq_rc) {
				snprintf(eb, sizeof eb, "%s", PQ.err.c_str());
				return bail("staging queries", nullptr);
			}
			const int rc = queries_stage_packed(D.ctx, PQ, &D.Q);
			if (PQ.users.fetch_sub(1) == 1) PQ.release(); // (every device has its copy)
			if (rc) return bail("staging queries", D.ctx);
		} else if (andi_hip_queries_stage(D.ctx, seqs, n, &D.Q)) {
			return bail("staging queries", D.ctx);
		}
		std::vector<unsigned long long> gcs; // G+C of every sequence (calc_gc, src/sequence.c:197-208)
		if (dev_prep && queries_gc_counts(D.ctx, D.Q, gcs)) return bail("staging queries", D.ctx);
		lap(t_queries);
		for (size_t b = 0; b < sets * batch; ++b)
			if (esa_reserve(D.prep, rs_cap, &D.slots[b])) return bail("allocating subject slots", D.prep);
		if (andi_hip_sync(D.prep)) return bail("allocating subject slots", D.prep);
		double t_reserve = 0, t_pinned = 0;
		lap(t_reserve);
		D.pinned_bytes = 2 * (rs_cap + 64);
		if (dev_prep || host_pool::pinned_get((void **)&D.pinned, D.pinned_bytes) != hipSuccess) D.pinned = nullptr; // (then from where RS lies)
		if (D.pinned && (hipEventCreateWithFlags(&D.pinned_free[0], hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&D.pinned_free[1], hipEventDisableTiming) != hipSuccess)) {
			host_pool::pinned_put(D.pinned, D.pinned_bytes);
			D.pinned = nullptr;
		}
		lap(t_pinned);
		if (andi_hip_dev_alloc(D.ctx, (use_rccl ? rows : batch) * n * sizeof(andi_hip_model), (void **)&D.d_rows)) return bail("row buffer", D.ctx);
		lap(t_slots);
		t_slots += t_reserve + t_pinned;
		if (trace && d == 0) fprintf(stderr, "andi_hip_dist_matrix trace: slots = device buffers of %zu slots %.1f ms + pinned upload buffer %.1f ms + row buffer %.1f ms\n", sets * batch, t_reserve, t_pinned, t_slots - t_reserve - t_pinned);

		// hand-over between the two stages
		std::mutex pm;
		std::condition_variable pcv;
		size_t prepared = 0, scanned = 0; // batches staged / scanned so far
		bool prep_failed = false;
		double p_take = 0, p_upload = 0, p_sort = 0, p_build = 0;

		size_t uploaded = dev_prep ? nbatches : 0; // batches whose texts are on the device (written there by the staging thread itself: all of them)
		// the device's compute alternates between the stages where a slot set is free for it: stage k + 1, then scan k
		const bool alternate = sets >= (dev_prep ? (size_t)2 : (size_t)3);
		auto give_up = [&]() {
			std::lock_guard<std::mutex> lk(pm);
			prep_failed = true;
			pcv.notify_all();
		};
		auto upload = [&]() { // the upload thread of this device: texts from the host pool into the slot sets, a batch ahead of the sorts
			(void)hipSetDevice(devs[d]);
			double tl = now_ms();
			auto plap = [&](double &acc) {
				const double t = now_ms();
				acc += t - tl, tl = t;
			};
			size_t nup = 0; // texts uploaded so far
			for (size_t k = 0; k < nbatches; ++k) {
				{
					std::unique_lock<std::mutex> lk(pm);
					pcv.wait(lk, [&] { return k < scanned + sets || prep_failed; }); // its set of slots is free again
					if (prep_failed) return;
				}
				tl = now_ms();
				const size_t i0 = first[d] + k * batch, nb = std::min(batch, last[d] - i0);
				andi_hip_esa **set = D.slots.data() + (k % sets) * batch;
				for (size_t b = 0; b < nb; ++b) { // uploads: beside whatever the device computes
					Prepared *p = take(i0 + b);
					if (!p) return give_up();
					plap(p_take);
					bool ok = true;
					if (p->rc) {
						char msg[96];
						snprintf(msg, sizeof msg, "Failed to create index for sequence %zu.", i0 + b); // src/dist_hack.h:53
						fail_all(msg);
						ok = false;
					}
					// through one of two pinned buffers: the next text is copied into the other while this one's transfer runs
					const bool two = D.pinned && !o.sa_on_host; // (a suffix array from the host is pageable memory: that copy waits anyway)
					const size_t pb = nup++ & 1;
					char *pin = D.pinned ? D.pinned + pb * (rs_cap + 64) : nullptr;
					if (ok && two && nup > 2 && hipEventSynchronize(D.pinned_free[pb]) != hipSuccess) bail("staging subject", D.up), ok = false;
					const char *src = p->RS;
					if (ok && pin) memcpy(pin, p->RS, p->n), src = pin;
					if (ok && esa_upload(D.up, set[b], src, o.sa_on_host ? p->SA.data() : nullptr, p->n, p->thr, two ? D.pinned_free[pb] : nullptr)) bail("staging subject", D.up), ok = false;
					plap(p_upload);
					andi_hip_free(p->RS);
					delete p;
					{
						std::lock_guard<std::mutex> lk(mu);
						++consumed;
					}
					cv.notify_all();
					if (!ok) return give_up();
				}
				if (andi_hip_sync(D.up)) { // (the batch's transfers)
					bail("staging subject", D.up);
					return give_up();
				}
				plap(p_upload);
				{
					std::lock_guard<std::mutex> lk(pm);
					uploaded = k + 1;
				}
				pcv.notify_all();
			}
		};
		auto stage = [&]() { // the staging thread of this device: suffix sorts and index builds
			(void)hipSetDevice(devs[d]);
			double tl = now_ms();
			auto plap = [&](double &acc) {
				const double t = now_ms();
				acc += t - tl, tl = t;
			};
			for (size_t k = 0; k < nbatches; ++k) {
				{ // the batch's texts are there; with three sets the device's compute is this batch's once the scan of batch k - 2 is done
					std::unique_lock<std::mutex> lk(pm);
					pcv.wait(lk, [&] { return (uploaded > k && (dev_prep ? k < scanned + sets : (sets < 3 || k < scanned + 2))) || prep_failed; });
					if (prep_failed) return;
				}
				tl = now_ms();
				const size_t i0 = first[d] + k * batch, nb = std::min(batch, last[d] - i0);
				andi_hip_esa **set = D.slots.data() + (k % sets) * batch;
				// RS from the resident sequence (the threshold on the host, same libm: src/sequence.c:210-219), then its suffix array
				auto text_and_sort = [&](andi_hip_ctx *cx, size_t b) -> const char * {
					const size_t i = i0 + b, len = seqs[i].len;
					const size_t thr = andi_hip_min_anchor_length(o.p_value, (double)gcs[i] / len, 2 * len + 1);
					if (esa_from_query(cx, set[b], D.Q, i, thr)) return "staging subject";
					if (esa_sort_suffixes(cx, set[b])) return "suffix array";
					return nullptr;
				};
				if (dev_prep) {
					const size_t width = std::min(nb, D.sorters.size() + 1);
					std::atomic<size_t> next_b{0};
					std::mutex em;
					const char *what = nullptr;
					andi_hip_ctx *where = nullptr;
					auto sort_some = [&](andi_hip_ctx *cx) {
						(void)hipSetDevice(devs[d]);
						for (;;) {
							const size_t b = next_b.fetch_add(1);
							if (b >= nb) break;
							const char *w = text_and_sort(cx, b);
							if (w) {
								std::lock_guard<std::mutex> lk(em);
								if (!what) what = w, where = cx;
								next_b.store(nb);
								break;
							}
						}
						if (andi_hip_sync(cx)) {
							std::lock_guard<std::mutex> lk(em);
							if (!what) what = "suffix array", where = cx;
						}
					};
					std::vector<std::thread> helpers;
					for (size_t w = 1; w < width; ++w) helpers.emplace_back(sort_some, D.sorters[w - 1]);
					sort_some(D.prep);
					for (auto &t : helpers) t.join();
					if (what) {
						bail(what, where);
						return give_up();
					}
					plap(p_sort);
				}
				if (andi_hip_esa_build_index_batch(D.prep, set, nb) || andi_hip_sync(D.prep)) {
					bail("index build", D.prep);
					return give_up();
				}
				plap(p_build);
				{
					std::lock_guard<std::mutex> lk(pm);
					prepared = k + 1;
				}
				pcv.notify_all();
			}
		};
		std::thread uploader;
		if (!dev_prep) uploader = std::thread(upload);
		std::thread stager(stage);

		std::vector<int64_t> self(batch);
		bool failed = false;
		t_last = now_ms();
		for (size_t k = 0; k < nbatches && !failed; ++k) {
			{
				std::unique_lock<std::mutex> lk(pm);
				pcv.wait(lk, [&] { return (prepared > k && (!alternate || prepared > k + 1 || prepared == nbatches)) || prep_failed; });
				if (prepared <= k) break; // (the staging thread has reported why)
			}
			lap(acc_wait);
			const size_t i0 = first[d] + k * batch, nb = std::min(batch, last[d] - i0);
			andi_hip_esa **set = D.slots.data() + (k % sets) * batch;
			for (size_t b = 0; b < nb; ++b) self[b] = (int64_t)(i0 + b);
			andi_hip_model *dst = use_rccl ? D.d_rows + (i0 - first[d]) * n : D.d_rows;
			if (andi_hip_scan_rows(D.ctx, set, self.data(), nb, D.Q, o.model, o.segment, dst)) bail("scan", D.ctx), failed = true;
			if (!failed && trace) (void)andi_hip_sync(D.ctx);
			lap(acc_scan);
			if (!failed && !use_rccl && andi_hip_copy_to_host(D.ctx, M + i0 * n, dst, nb * n * sizeof(andi_hip_model))) bail("row copy", D.ctx), failed = true;
			if (!failed && use_rccl && andi_hip_sync(D.ctx)) bail("scan", D.ctx), failed = true; // the slots are reused
			lap(acc_copy);
			{
				std::lock_guard<std::mutex> lk(pm);
				scanned = k + 1;
				if (failed) prep_failed = true;
			}
			pcv.notify_all();
			if (!failed && o.progress) {
				std::lock_guard<std::mutex> lk(mu);
				rows_done += nb;
				o.progress(rows_done * (n - 1), n * n - n, o.ud);
			}
		}
		{
			std::lock_guard<std::mutex> lk(pm);
			if (scanned < nbatches) prep_failed = true; // (release the staging thread)
		}
		pcv.notify_all();
		stager.join();
		if (uploader.joinable()) uploader.join();
		if (trace && d == 0)
			fprintf(stderr, "andi_hip_dist_matrix trace (ms): contexts %.1f, queries %.1f, slots %.1f | staging thread: waiting for the host pool %.1f, subject %s %.1f, suffix arrays %.1f, index builds %.1f | scan thread: waiting for staged subjects %.1f, scans %.1f, row copies %.1f; driver total %.1f (%zu batches of %zu, %zu slot sets)\n",
					t_ctx, t_queries, t_slots, p_take, dev_prep ? "texts written on the device" : "uploads", p_upload, p_sort, p_build, acc_wait, acc_scan, acc_copy, now_ms() - t_call, nbatches, batch, sets);
	};

	std::vector<std::thread> pool, drivers;
	for (int t = 0; t < threads && !dev_prep; ++t) pool.emplace_back(worker);
	if (ndev == 1) {
		drive(0); // the calling thread, as before
	} else {
		for (size_t d = 0; d < ndev; ++d) drivers.emplace_back(drive, d);
		for (auto &t : drivers) t.join();
	}
	{
		std::lock_guard<std::mutex> lk(mu);
		consumed = n; // release any waiting worker
		if (abort_flag) next.store(n);
	}
	cv.notify_all();
	for (auto &t : pool) t.join();
	if (packer.joinable()) packer.join();
	for (auto *p : ready) {
		andi_hip_free(p->RS);
		delete p;
	}
	int rc = abort_flag ? 1 : 0;
	const double t_drivers_done = now_ms();

	// ---- the gather: row blocks to the first device over RCCL, one copy to the host
	snprintf(g_last_gather, sizeof g_last_gather, "%s", use_rccl ? "rccl" : "direct");
	if (!rc && use_rccl) {
		Rccl &R = rccl();
		std::vector<ncclComm_t> comms(ndev, nullptr);
		andi_hip_model *d_full = nullptr;
		std::string err;
		auto nccl_ok = [&](ncclResult_t r, const char *what) {
			if (r == ncclSuccess) return true;
			if (err.empty()) err = std::string(what) + ": " + R.GetErrorString(r);
			return false;
		};
		// one process, one node: the communicators bootstrap over the loopback interface unless the caller chose one;
		// the caller's environment is put back as it was (a later multi-node initialisation in this process must not
		// inherit the loopback)
		// (RCCL takes the interface from the process environment and from nowhere else: two of this library's calls are
		// kept apart by a lock; a caller whose OTHER threads read or write the environment meanwhile sets
		// NCCL_SOCKET_IFNAME itself before its first call -- the library then leaves the environment alone, andi_hip.h)
		static std::mutex env_lock;
		bool ok;
		{
			std::lock_guard<std::mutex> guard(env_lock);
			const bool had_ifname = getenv("NCCL_SOCKET_IFNAME") != nullptr;
			if (!had_ifname) setenv("NCCL_SOCKET_IFNAME", "lo", 0);
			ok = nccl_ok(R.CommInitAll(comms.data(), (int)ndev, devs.data()), "ncclCommInitAll");
			if (!had_ifname) unsetenv("NCCL_SOCKET_IFNAME");
		}
		if (ok && hipSetDevice(devs[0]) != hipSuccess) ok = false, err = "hipSetDevice";
		if (ok && hipMalloc((void **)&d_full, n * n * sizeof(andi_hip_model)) != hipSuccess) ok = false, err = "allocating the gathered matrix";
		if (ok) {
			ok = nccl_ok(R.GroupStart(), "ncclGroupStart");
			for (size_t d = 1; d < ndev && ok; ++d) {
				const size_t bytes = (last[d] - first[d]) * n * sizeof(andi_hip_model);
				// (every call with the device of its communicator current)
				ok = hipSetDevice(devs[d]) == hipSuccess &&
					 nccl_ok(R.Send(dv[d].d_rows, bytes, ncclUint8, 0, comms[d], dv[d].ctx->stream), "ncclSend") &&
					 hipSetDevice(devs[0]) == hipSuccess &&
					 nccl_ok(R.Recv(d_full + first[d] * n, bytes, ncclUint8, (int)d, comms[0], dv[0].ctx->stream), "ncclRecv");
			}
			if (!nccl_ok(R.GroupEnd(), "ncclGroupEnd")) ok = false;
		}
		if (ok) { // the first device's own block, then everything to the host
			hipError_t e = hipSetDevice(devs[0]);
			if (e == hipSuccess)
				e = hipMemcpyAsync(d_full + first[0] * n, dv[0].d_rows, (last[0] - first[0]) * n * sizeof(andi_hip_model),
								   hipMemcpyDeviceToDevice, dv[0].ctx->stream);
			for (size_t d = 1; d < ndev && e == hipSuccess; ++d) {
				e = hipSetDevice(devs[d]);
				if (e == hipSuccess) e = hipStreamSynchronize(dv[d].ctx->stream);
			}
			if (e == hipSuccess) e = hipSetDevice(devs[0]);
			if (e == hipSuccess) e = hipStreamSynchronize(dv[0].ctx->stream);
			if (e == hipSuccess) e = hipMemcpy(M, d_full, n * n * sizeof(andi_hip_model), hipMemcpyDeviceToHost);
			if (e != hipSuccess) ok = false, err = std::string("gathering the matrix: ") + hipGetErrorString(e);
		}
		for (auto cm : comms)
			if (cm) (void)R.CommDestroy(cm);
		if (d_full) {
			(void)hipSetDevice(devs[0]);
			(void)andi_arena::dev_free(d_full);
		}
		if (!ok) { // RCCL unusable on this box: the rows are still in HBM -- copy every block to the host directly
			snprintf(g_last_gather, sizeof g_last_gather, "direct (rccl: %.160s)", err.c_str());
			for (size_t d = 0; d < ndev && !rc; ++d)
				if (andi_hip_copy_to_host(dv[d].ctx, M + first[d] * n, dv[d].d_rows, (last[d] - first[d]) * n * sizeof(andi_hip_model))) {
					first_error = std::string("row copy: ") + andi_hip_last_error(dv[d].ctx);
					rc = 1;
				}
		}
	}
	if (rc) set_err(errbuf, errlen, "%s", first_error.empty() ? "andi_hip_dist_matrix failed" : first_error.c_str());
	const double t_gathered = now_ms();

	for (auto &D : dv) {
		if (!D.ctx) {
			for (auto *cx : D.sorters) andi_hip_ctx_destroy(cx);
			if (D.prep) andi_hip_ctx_destroy(D.prep);
			if (D.up) andi_hip_ctx_destroy(D.up);
			continue;
		}
		for (auto *cx : D.sorters) andi_hip_ctx_destroy(cx);
		for (auto *e : D.slots)
			if (e) andi_hip_esa_free(D.ctx, e);
		if (D.d_rows) andi_hip_dev_free(D.ctx, D.d_rows);
		if (D.Q) andi_hip_queries_free(D.ctx, D.Q);
		if (D.pinned) host_pool::pinned_put(D.pinned, D.pinned_bytes);
		for (hipEvent_t ev : D.pinned_free)
			if (ev) (void)hipEventDestroy(ev);
		if (D.prep) andi_hip_ctx_destroy(D.prep);
		if (D.up) andi_hip_ctx_destroy(D.up);
		andi_hip_ctx_destroy(D.ctx);
	}
	if (trace) fprintf(stderr, "andi_hip_dist_matrix trace: call total %.1f ms (gather %.1f, slots, queries and contexts released %.1f)\n", now_ms() - t_call, t_gathered - t_drivers_done, now_ms() - t_gathered);
	return rc;
}

} // extern "C"
