/*
 * andi_hip.h — C-ABI of libandihip.so, the MI355X (gfx950) engine behind
 * andi's anchor-distance hot path.
 *
 * The reference has no FFI for this path; its seam is internal:
 * calculate_distances() (src/process.c:230) allocates the n*n `struct model`
 * matrix and calls distMatrix()/distMatrixLM() (src/dist_hack.h:34), which per
 * subject run seq_subject_init (src/sequence.c:210), esa_init
 * (src/esa.c:254) and, per query, dist_anchor (src/process.c:141).  Every
 * entry point below names the reference function it replaces.  All
 * signatures are plain C: pointers, sizes, PODs.  Functions return 0 on
 * success; on failure they return non-zero and write a message to
 * errbuf (when given).  Nothing here falls back to a CPU implementation of a
 * device step: without a usable HIP device the device entry points fail.
 */
#ifndef ANDI_HIP_H
#define ANDI_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ANDI_HIP_ABI_VERSION 5

/* enum in src/global.h:50 */
enum { ANDI_M_RAW = 0, ANDI_M_JC = 1, ANDI_M_KIMURA = 2, ANDI_M_LOGDET = 3, ANDI_M_ANI = 4 };

/* seq_t (src/sequence.h:18-25) without the name: NUL-terminated, over
 * {A,C,G,T,!} as produced by normalize() (src/sequence.c:260-282).  Any other
 * byte is refused by the scan (andi_hip_scan_rows / andi_hip_dist_matrix return an
 * error): the device index codes a symbol in 2 or 4 bits. */
typedef struct {
	const char *seq;
	size_t len;
} andi_hip_seq;

/* struct model (src/model.h:52-57): counts[4*from+to], then the query length.
 * 68 bytes, no padding. */
typedef struct {
	uint32_t counts[16];
	uint32_t seq_len;
} andi_hip_model;

/* lcp_inter_t (src/esa.h:25-34) */
typedef struct {
	int32_t l, i, j, m;
} andi_hip_interval;

/* The globals the reference reads deep inside the path, made explicit:
 * ANCHOR_P_VALUE (src/andi.c:48), MODEL (src/andi.c:50; selects the equal-run
 * attribution of src/model.c:247), THREADS (src/andi.c:46), F_LOW_MEMORY
 * (src/global.h:63), the progress line of src/dist_hack.h:40-43,74-87. */
typedef struct {
	double p_value;
	int model;
	int device;        /* first HIP device ordinal */
	int host_threads;  /* <=0: all cores (suffix sorting pool, shared by the devices) */
	int low_memory;    /* bound the number of resident subject indexes */
	uint32_t segment;  /* query nucleotides per scan work item; 0 = default */
	void (*progress)(size_t done, size_t total, void *ud); /* serialised; called from the devices' driver threads */
	void *ud;
	int sa_on_host;    /* 0: suffix arrays are built on the device (sa_device.hip); 1: by the host pool
	                    * (andi_hip_suffix_array), as the reference does with libdivsufsort (src/esa.c:303) */
	/* The N x N loop is tiled over the GPUs of the node by rows (the parallel loop of
	 * src/dist_hack.h:46-47): num_gpus devices device, device + 1, ...; 0 or 1 = one device; < 0 = all
	 * visible devices from `device` on.  Every device owns a contiguous block of subject rows; the row
	 * blocks are gathered on the first device with RCCL (send/recv over xGMI) and copied to M once. */
	int num_gpus;
	const int *devices; /* optional: exactly these num_gpus ordinals (an ordinal may repeat: several contexts
	                     * on one device, rows then go to M directly) */
} andi_hip_opts;

void andi_hip_default_opts(andi_hip_opts *o);
int andi_hip_abi_version(void);

/* Device memory: the library carves its buffers out of large chunks (2 GiB, ANDI_ARENA_MB).  Up to ANDI_ARENA_KEEP MiB of
 * them (default 8192; 0: none, as up to ABI 3) stay with the process when the last context on a device is destroyed, so
 * that the next call of the seam does not pay the driver for them again (on some hosts 0.3 s per call for a 29-genome
 * job); what a larger job took beyond that goes back with its last context.  This gives every chunk nobody holds a block of
 * back to the driver -- all devices the library has used; waits for each --, and returns the bytes released; the idle streams
 * and pinned upload buffers destroyed contexts left behind (kept for the next call as well: a stream costs this runtime 3 ms to
 * create, nine per call of the seam) go with them.  A process that never used the library makes no HIP call here. */
size_t andi_hip_trim(void);

/* Host helper: the 4-bit symbols of a byte string as the engine keeps its texts (A C G T ! ; # NUL = 0 ... 7; byte j of out
 * = symbol 2j | symbol 2j+1 << 4, the NUL behind an odd length included): (len + 1) / 2 bytes.  Returns 1 if a byte lies
 * outside that alphabet (src/sequence.c:260-282 never produces one).  The seam packs its queries with it, once for all
 * devices; no GPU is touched. */
int andi_hip_pack_symbols(const unsigned char *src, size_t len, unsigned char *out);

/* ------------------------------------------------------------------ */
/* The seam: replaces distMatrix / distMatrixLM (src/dist_hack.h:34-96) */
/* as called from calculate_distances (src/process.c:247-251).         */
/* M is caller-owned, n*n row-major, M[i*n+j] = subject i vs query j,  */
/* diagonal = {counts[0]=9, seq_len=9} (src/dist_hack.h:61-64).        */
/* ------------------------------------------------------------------ */
int andi_hip_dist_matrix(andi_hip_model *M, const andi_hip_seq *seqs, size_t n,
						 const andi_hip_opts *opts, char *errbuf, size_t errlen);
/* how the calling thread's last call collected its rows: "rccl", "direct (...)" (diagnostic; per thread).
 * A call that spans several devices initialises RCCL communicators, and RCCL reads the bootstrap interface from the
 * process environment only: unless NCCL_SOCKET_IFNAME is set, the library sets it to "lo" around ncclCommInitAll and
 * takes it back (its own calls serialised by a lock).  A caller with other threads that touch the environment sets
 * NCCL_SOCKET_IFNAME itself beforehand: the library then never writes the environment. */
const char *andi_hip_last_gather(void);
/* The seam's tiling of the parallel subject loop (src/dist_hack.h:46-47) over `parts` devices: part k owns the
 * contiguous rows [*first, *last) of `total`; sizes differ by at most one, the longer blocks come first.  A caller that
 * runs one process per GPU instead (bench.py --gpus N, andi_amd/shard.py) partitions with the same rule.  No GPU is touched. */
void andi_hip_row_block(size_t total, size_t parts, size_t k, size_t *first, size_t *last);

/* ------------------------------------------------------------------ */
/* Host pieces of the path (stay on the host, same libm)               */
/* ------------------------------------------------------------------ */
/* seq_subject_init (src/sequence.c:210-219): RS = revcomp(S) '#' S '\0'
 * (malloc'ed, free with andi_hip_free), RSlen = 2*len+1, gc, threshold. */
int andi_hip_subject_prepare(const char *seq, size_t len, double p_value,
							 char **RS, size_t *RSlen, double *gc, size_t *threshold);
void andi_hip_free(void *p);
/* min_anchor_length / shustring_cum_prob (src/sequence.c:296-304,353-373) */
size_t andi_hip_min_anchor_length(double p, double g, size_t l);
double andi_hip_shustring_cum_prob(size_t x, double p, size_t l);
/* divsufsort() as called at src/esa.c:303: T[0..n) unsigned bytes, T[n]
 * must be readable; SA[0..n).  Re-entrant. */
int andi_hip_suffix_array(const unsigned char *T, int32_t *SA, int32_t n);
/* which sorter that is: libdivsufsort where the host has it (loaded at first use with dlopen -- the reference's own,
 * configure.ac:33-38; ANDI_HIP_NO_DIVSUFSORT in the environment keeps it away), else the built-in linear-time SA-IS */
const char *andi_hip_suffix_sorter(void);
/* model_average / model_coverage / estimate_* (src/model.c:39-210) */
andi_hip_model andi_hip_model_average(const andi_hip_model *a, const andi_hip_model *b);
double andi_hip_model_coverage(const andi_hip_model *m);
double andi_hip_estimate(const andi_hip_model *m, int model);
/* print_distances (src/io.c:246-322) into a caller buffer: PHYLIP text of the
 * n*n matrix.  names[i] as seq_t.name.  Returns the number of bytes needed
 * (excluding NUL); writes at most cap.  *warn_flags gets bit0 = a NaN was
 * reported, bit1 = coverage < 0.2 reported; warning lines go to warnbuf. */
size_t andi_hip_format_distances(const andi_hip_model *M, const char *const *names, size_t n,
								 int model, int extra_verbose, int truncate_names, int warnings,
								 char *out, size_t cap, char *warnbuf, size_t warncap,
								 int *warn_flags);

/* ------------------------------------------------------------------ */
/* Device-resident objects                                             */
/* ------------------------------------------------------------------ */
typedef struct andi_hip_ctx andi_hip_ctx;         /* device + streams + scratch */
typedef struct andi_hip_esa andi_hip_esa;         /* one subject's esa_s (src/esa.h:42-59) in HBM */
typedef struct andi_hip_queries andi_hip_queries; /* all query sequences in HBM */

int andi_hip_device_count(void); /* visible HIP devices; 0 if none (or no usable runtime) */
int andi_hip_ctx_create(andi_hip_ctx **ctx, int device, char *errbuf, size_t errlen);
void andi_hip_ctx_destroy(andi_hip_ctx *ctx);
/* How many queries the subjects staged in this context from now on will be scanned against (0 = unknown, the
 * default).  Decides the depth of their probe tables: from 1024 queries on, one level deeper than the text's length
 * asks for (4x the table, built once per subject; fewer text accesses per probe, paid back over the queries).
 * andi_hip_dist_matrix sets it itself (n - 1: distMatrix compares every sequence with every other,
 * src/dist_hack.h:59-68).  Results do not depend on it. */
void andi_hip_ctx_expect_queries(andi_hip_ctx *ctx, size_t queries);
const char *andi_hip_last_error(const andi_hip_ctx *ctx);
int andi_hip_sync(andi_hip_ctx *ctx);

/* Upload RS (n bytes + NUL) and its suffix array (esa_init_SA's output,
 * src/esa.c:294-304).  Host→device copies only. */
int andi_hip_esa_stage(andi_hip_ctx *ctx, const char *RS, const int32_t *SA, size_t n,
					   size_t threshold, andi_hip_esa **out);
/* The same with the suffix array built on the device (esa_init_SA, src/esa.c:294-304, without the host:
 * prefix doubling on radix sorts): only RS is uploaded.  Synchronous. */
int andi_hip_esa_stage_text(andi_hip_ctx *ctx, const char *RS, size_t n, size_t threshold, andi_hip_esa **out);
/* Test hook: the suffix array as the device holds it, n entries. */
int andi_hip_esa_download_sa(andi_hip_ctx *ctx, const andi_hip_esa *esa, int32_t *SA);
/* esa_init_LCP, _CLD, _FVC, _cache (src/esa.c:373-426, 312-363, 229-245,
 * 73-215) as HIP kernels on the context's stream (asynchronous): the
 * reference's own arrays, bit for bit. */
int andi_hip_esa_build(andi_hip_ctx *ctx, andi_hip_esa *esa);
/* The index the anchor scan uses in place of those arrays: a 4^K table holding,
 * per K-mer, the outcome of get_match_cached (src/esa.c:636-656) as far as
 * the K-mer decides it, built from RS and SA alone (asynchronous).  When the
 * build finds that the reference's 10-mer table may contain an entry spanning
 * a separator (flags[0], see andi_hip_esa_flags), andi_hip_scan_rows builds
 * the reference arrays for that subject and follows the reference's walk. */
int andi_hip_esa_build_index(andi_hip_ctx *ctx, andi_hip_esa *esa);
/* The same for several staged subjects in two launches (no launch gaps, one tail). */
int andi_hip_esa_build_index_batch(andi_hip_ctx *ctx, andi_hip_esa *const *esas, size_t count);
/* flags[0]: see above; flags[2]: the 10-mer table kernel really produced such an
 * entry (only after andi_hip_esa_build). */
int andi_hip_esa_flags(andi_hip_ctx *ctx, const andi_hip_esa *esa, int32_t *flags4);
/* Test hook: copy the built arrays back.  Any pointer may be NULL.
 * LCP/CLD have n+1 entries, FVC n, cache 4^10. */
int andi_hip_esa_download(andi_hip_ctx *ctx, const andi_hip_esa *esa, int32_t *LCP,
						  int32_t *CLD, uint8_t *FVC, andi_hip_interval *cache);
/* Test hook: the scan index's probe table (what the scan consults in place of get_match_cached,
 * src/esa.c:636-656), 4^K entries of two 32-bit words {x, y}; *K receives the depth.  y & 3 is the kind:
 * 0 FINAL -- the K-mer is absent from RS, its longest match is y >> 8 (< K) characters, unique iff y & 4,
 * and then x is the suffix-array index of the one suffix; 1 SINGLE -- it occurs once, at RS offset x;
 * 2 MULTI -- it occurs (y >> 8) + 1 times, at the suffix-array indices x, x + 1, ...; 3 -- search the
 * whole suffix array.  `table` may be NULL (only K is wanted); it needs 8 << 2K bytes. */
int andi_hip_esa_download_index(andi_hip_ctx *ctx, const andi_hip_esa *esa, uint32_t *table, int *K);
/* the form of the table's entries of K-mers that occur once (test hook): 0 plain, 1 the up to 13 nucleotides behind the
 * occurrence in the entry, 2 the up to min(4, 16 - K) the device sorter's keys held (the default for subjects sorted on the device) */
int andi_hip_esa_single_form(const andi_hip_esa *esa);
void andi_hip_esa_free(andi_hip_ctx *ctx, andi_hip_esa *esa);
size_t andi_hip_esa_bytes(const andi_hip_esa *esa);

int andi_hip_queries_stage(andi_hip_ctx *ctx, const andi_hip_seq *seqs, size_t n,
						   andi_hip_queries **out);
void andi_hip_queries_free(andi_hip_ctx *ctx, andi_hip_queries *q);

/* get_match_cached / get_match (src/esa.c:615-656) for `count` consecutive
 * suffixes of query `qidx`: out[k] = match of Q[first+k ..] (fields l,i,j;
 * m = SA[i], the pos_S dist_anchor would use, src/process.c:120). */
int andi_hip_match_positions(andi_hip_ctx *ctx, const andi_hip_esa *esa,
							 const andi_hip_queries *q, size_t qidx, size_t first, size_t count,
							 int cached, andi_hip_interval *out_host);

/* dist_anchor (src/process.c:141-214) for every (subject, query) pair of
 * `nsub` staged subjects whose index is built (andi_hip_esa_build_index, or
 * andi_hip_esa_build for the reference walk) against all queries.  self[s] = index of the
 * query that is subject s itself (gets the diagonal placeholder) or -1.
 * M_dev: device pointer, nsub * nq models, row s = subject s.  Asynchronous
 * on the context's stream. */
int andi_hip_scan_rows(andi_hip_ctx *ctx, andi_hip_esa *const *subjects,
					   const int64_t *self, size_t nsub, const andi_hip_queries *q, int model,
					   uint32_t segment, andi_hip_model *M_dev);

/* calculate_bootstrap (src/process.c:289-321): `replicates` resampled matrices
 * from M (host, n*n) into B (host, replicates*n*n).  For every pair i < j the
 * summed counts model_average(M(i,j), M(j,i)) are redrawn from a multinomial
 * (model_bootstrap, src/model.c:222-232), mirrored, diagonal {counts[0]=1,
 * seq_len=1}.  Deterministic in (seed, replicate, i, j).  The reference seeds
 * GSL from the clock, so only the distribution can be compared, not the draws. */
int andi_hip_bootstrap(andi_hip_ctx *ctx, const andi_hip_model *M, size_t n, uint64_t seed,
					   size_t replicates, andi_hip_model *B);

/* plain device memory helpers so callers need no HIP headers */
int andi_hip_dev_alloc(andi_hip_ctx *ctx, size_t bytes, void **dptr);
void andi_hip_dev_free(andi_hip_ctx *ctx, void *dptr);
int andi_hip_copy_to_host(andi_hip_ctx *ctx, void *dst_host, const void *src_dev, size_t bytes);

/* The measured device-copy ceiling the roofline is reported beside: `reps` passes of a 16-bytes-per-lane
 * streaming copy kernel of the engine's own over `bytes` (source and destination allocated here, non-temporal loads and
 * stores, grid-stride), timed with HIP events on the context's stream; *gbps = read + written bytes per second / 1e9. */
int andi_hip_copy_ceiling(andi_hip_ctx *ctx, size_t bytes, int reps, double *gbps);

/* Kernel timing, measured with HIP events on the stream the kernels run on.
 * Accumulates since the last reset; read after andi_hip_sync(). */
typedef struct {
	double build_ms;      /* index builds: the scan index (andi_hip_esa_build_index), the reference arrays K1-K4 (andi_hip_esa_build) */
	uint64_t build_launches;
	double scan_ms;       /* anchor scan pass A (the dominant kernel) */
	uint64_t scan_launches;
	double stitch_ms;     /* passes B + C */
	uint64_t stitch_launches;
	uint64_t scan_query_nt; /* sum of query lengths over scanned pairs */
	uint64_t scan_pairs;
	uint64_t fixups;      /* segments whose speculative entry state was wrong */
	uint64_t reference_subjects; /* subjects scanned with the reference walk */
	double sa_ms;            /* suffix arrays built on the device (wall time of the calls: they synchronise per round) */
	uint64_t sa_builds;
	uint64_t sa_rounds;      /* sorting rounds of those builds */
	uint64_t adaptive_calls; /* scan calls that chose the segment length per pair */
	uint64_t uniform_calls;  /* ... one segment length for the call */
	uint64_t coop_calls;     /* scan calls in which pass A by wavefronts (scan_coop.hip) ran: for every pair (ANDI_COOP=n; by default calls of 2^18 ... 2^25 query symbols x subjects) or for the pairs routed to it */
	uint64_t coop_fallbacks; /* routed calls: PAIRS that kernel handed back to the lane scan (a stretch without homology, a match longer than a segment) */
	uint64_t routed_calls;   /* scan calls whose pass A was routed per pair (by default calls of 2^25 query symbols x subjects and more) */
	uint64_t coop_query_nt;  /* routed calls: query nucleotides of the pairs whose pass A ran by wavefronts ... */
	uint64_t lane_query_nt;  /* ... and by lanes */
	uint64_t pool_calls;     /* scan calls whose pass A by wavefronts was k_pool_cold's (the windows' walks pooled through global memory), not k_coop_cold's (ABI 5) */
} andi_hip_timings;

/* The library's ANDI_* environment switches (experiments, diagnostics: INTEGRATION.md lists them) are read once, when
 * the library first looks at one; this reads them again (the tests change them under a live context).  Not to be
 * called while another thread is inside the library. */
void andi_hip_reload_knobs(void);
int andi_hip_timings_get(andi_hip_ctx *ctx, andi_hip_timings *t);
void andi_hip_timings_reset(andi_hip_ctx *ctx);

#ifdef __cplusplus
}
#endif
#endif
