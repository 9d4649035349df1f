/*
 * andi_oracle.h — CPU restatement of the andi anchor-distance hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product (andi_amd/, include/,
 * libandihip.so) may include, link or call this.  Allowed users: tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg.
 *
 * Every function cites the reference file:line (relative to the upstream
 * EvolBioInf/andi v1.15 tree) whose behaviour it restates.  Parity pins: see
 * the header comment of andi_oracle.c.
 */
#ifndef ANDI_ORACLE_H
#define ANDI_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* model enum, global.h:50 */
enum { ORC_M_RAW = 0, ORC_M_JC, ORC_M_KIMURA, ORC_M_LOGDET, ORC_M_ANI };

/* struct model, model.h:52-57 — 17 x u32, no padding */
typedef struct {
	uint32_t counts[16];
	uint32_t seq_len;
} orc_model;

/* lcp_inter_t, esa.h:25-34 (field order l,i,j,m) */
typedef struct {
	int32_t l, i, j, m;
} orc_interval;

/* esa_s, esa.h:42-59 */
typedef struct {
	const char *S; /* RS, NUL-terminated */
	int32_t *SA;
	int32_t *LCP; /* len+1 entries */
	int32_t len;
	orc_interval *cache; /* 4^10 entries */
	char *FVC;
	int32_t *CLD; /* len+1 entries */
} orc_esa;

/* seq_subject, sequence.h:32-45 */
typedef struct {
	char *RS;
	size_t RSlen;
	double gc;
	size_t threshold;
} orc_subject;

/* counters used to pin dist_anchor against the instrumented reference run
 * recorded in SURVEY.md §6.2 */
typedef struct {
	uint64_t iterations;   /* while-loop trips, process.c:153 */
	uint64_t esa_probes;   /* get_match_cached calls, process.c:117 */
	uint64_t lucky_tries;  /* lcp() evaluated, process.c:95 */
	uint64_t lucky_hits;   /* lucky_anchor returned true */
	uint64_t anchor_pairs; /* right-anchor branch taken, process.c:163 */
	uint64_t gap_chars;    /* total len passed to model_count */
} orc_scan_stats;

/* ---- sequence.c ------------------------------------------------------ */
size_t orc_normalize(char *s, int *non_acgt);                  /* sequence.c:260-282 */
char *orc_revcomp(const char *s, size_t len);                  /* sequence.c:143-168 */
char *orc_catcomp(const char *s, size_t len);                  /* sequence.c:177-190 */
double orc_gc(const char *s, size_t len);                      /* sequence.c:197-208 */
size_t orc_binomial(size_t n, size_t k);                       /* sequence.c:315-335 */
double orc_shustring_cum_prob(size_t x, double p, size_t l);   /* sequence.c:353-373 */
size_t orc_min_anchor_length(double p, double g, size_t l);    /* sequence.c:296-304 */
int orc_subject_init(orc_subject *sub, const char *s, size_t len,
					 double p_value);                          /* sequence.c:210-219 */
void orc_subject_free(orc_subject *sub);

/* ---- esa.c ----------------------------------------------------------- */
int orc_suffix_array(const unsigned char *T, int32_t *SA, int32_t n); /* stands in for divsufsort, esa.c:303 */
int orc_esa_init(orc_esa *E, const orc_subject *sub);          /* esa.c:254-277 */
/* same, but takes a caller-provided suffix array (copied) */
int orc_esa_init_with_sa(orc_esa *E, const orc_subject *sub, const int32_t *SA);
void orc_esa_free(orc_esa *E);
orc_interval orc_get_match(const orc_esa *E, const char *q, size_t qlen);        /* esa.c:615-624 */
orc_interval orc_get_match_cached(const orc_esa *E, const char *q, size_t qlen); /* esa.c:636-656 */

/* ---- model.c --------------------------------------------------------- */
void orc_model_count(orc_model *m, const char *s, const char *q, size_t len);     /* model.c:309-337 */
void orc_model_count_equal(orc_model *m, const char *s, size_t len, int model);   /* model.c:246-279 */
orc_model orc_model_average(const orc_model *a, const orc_model *b);              /* model.c:39-46 */
size_t orc_model_total(const orc_model *m);                                       /* model.c:54-60 */
double orc_model_coverage(const orc_model *m);                                    /* model.c:68-73 */
double orc_estimate(const orc_model *m, int model);                               /* model.c:81-208 */

/* ---- process.c ------------------------------------------------------- */
orc_model orc_dist_anchor(const orc_esa *E, const char *q, size_t qlen,
						  size_t threshold, int model,
						  orc_scan_stats *stats /* may be NULL */);     /* process.c:141-214 */

/* dist_hack.h:34-96 + process.c:230-251: fill n*n row-major M.
 * seqs[i] NUL-terminated over {A,C,G,T,!}.  threads<=0 → all cores.
 * times_out (may be NULL): [0]=index build seconds (sum over subjects, wall
 * inside the parallel loop), [1]=scan seconds (same convention). */
int orc_dist_matrix(orc_model *M, const char *const *seqs, const size_t *lens,
					size_t n, double p_value, int model, int threads,
					double *times_out);

/* scan only: all queries against one prepared ESA (for the timed CPU
 * baseline). */
void orc_scan_row(orc_model *row, const orc_esa *E, size_t threshold,
				  const char *const *seqs, const size_t *lens, size_t n,
				  size_t self, int model, int threads);

/* ---- bootstrap (model.c:222-232, process.c:289-321) -------------------- */
/* GSL is absent from the image: its published conditional-binomial multinomial is restated; the reference seeds from the
 * clock, so the distribution -- not a stream -- is what can be compared (parity unpinned, see andi_oracle.c). */
uint64_t orc_ran_binomial(uint64_t *rng, double p, uint64_t n);                          /* gsl_ran_binomial's law */
void orc_ran_multinomial(uint64_t *rng, size_t K, uint64_t N, const double *p, uint32_t *n); /* gsl_ran_multinomial, model.c:229 */
orc_model orc_model_bootstrap(orc_model datum, uint64_t *rng);                           /* model.c:222-232 */
void orc_bootstrap_matrix(orc_model *B, const orc_model *M, size_t n, uint64_t seed);    /* process.c:299-316, one replicate */

#ifdef __cplusplus
}
#endif
#endif
