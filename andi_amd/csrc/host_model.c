/*
 * host_model.c — what happens to the integer counts after the device scan:
 * symmetrisation, distance estimators and the PHYLIP printer.  Kept on the
 * host so RAW/JC/Kimura distances are bit-identical to the reference given
 * identical counts.  Follows src/model.c:39-210 and
 * src/io.c:246-338.
 */
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "andi_hip.h"

/* cell index = 4*from + to, A C G T = 0 1 2 3 (src/model.h:14-32) */
#define CELL(f, t) (4 * (f) + (t))
enum { nA, nC, nG, nT };

/* model_average, src/model.c:39-46 — element-wise sum, seq_len included */
andi_hip_model andi_hip_model_average(const andi_hip_model *a, const andi_hip_model *b) {
	andi_hip_model r = *a;
	for (int k = 0; k < 16; k++) r.counts[k] += b->counts[k];
	r.seq_len += b->seq_len;
	return r;
}

/* model_total, src/model.c:54-60 */
static size_t total(const andi_hip_model *m) {
	size_t t = 0;
	for (int k = 0; k < 16; k++) t += m->counts[k];
	return t;
}

/* model_coverage, src/model.c:68-73 */
double andi_hip_model_coverage(const andi_hip_model *m) {
	return (double)total(m) / (double)m->seq_len;
}

static size_t off_diagonal(const andi_hip_model *m) {
	size_t t = 0;
	for (int f = 0; f < 4; f++)
		for (int g = 0; g < 4; g++)
			if (f != g) t += m->counts[CELL(f, g)];
	return t;
}

/* estimate_RAW, src/model.c:81-92 */
static double raw(const andi_hip_model *m) {
	size_t nucl = total(m);
	size_t snps = off_diagonal(m);
	if (nucl <= 3) return NAN;
	return (double)snps / (double)nucl;
}

/* estimate_JC, src/model.c:100-106 */
static double jc(const andi_hip_model *m) {
	double d = raw(m);
	d = -0.75 * log(1.0 - (4.0 / 3.0) * d);
	return d <= 0.0 ? 0.0 : d;
}

/* estimate_KIMURA, src/model.c:113-127 */
static double kimura(const andi_hip_model *m) {
	size_t nucl = total(m);
	size_t ts = (size_t)m->counts[CELL(nA, nG)] + m->counts[CELL(nG, nA)] +
				m->counts[CELL(nC, nT)] + m->counts[CELL(nT, nC)];
	size_t tv = off_diagonal(m) - ts;
	double P = (double)ts / (double)nucl;
	double Q = (double)tv / (double)nucl;
	double w = 1.0 - 2.0 * P - Q;
	double d = -0.25 * log((1.0 - 2.0 * Q) * w * w);
	return d <= 0.0 ? 0.0 : d;
}

/* estimate_LOGDET, src/model.c:155-199.  The 4x4 determinant is expanded in
 * the same term order as the reference so the double matches. */
static double logdet(const andi_hip_model *m) {
	double nucl = (double)total(m);
	double P[16];
	for (int k = 0; k < 16; k++) P[k] = m->counts[k] / nucl;

	double lg = 0.0;
	for (int f = 0; f < 4; f++) {
		size_t s = 0;
		for (int g = 0; g < 4; g++) s += m->counts[CELL(f, g)];
		double term = log(s / nucl);
		lg = f ? lg + term : term;
	}
	for (int g = 0; g < 4; g++) {
		size_t s = 0;
		for (int f = 0; f < 4; f++) s += m->counts[CELL(f, g)];
		lg = lg + log(s / nucl);
	}

#define p(f, g) P[CELL(n##f, n##g)]
	double det = p(A, A) * p(C, C) * (p(G, G) * p(T, T) - p(T, G) * p(G, T)) -
				 p(A, A) * p(C, G) * (p(G, C) * p(T, T) - p(T, C) * p(G, T)) +
				 p(A, A) * p(C, T) * (p(G, C) * p(T, G) - p(T, C) * p(G, G)) -

				 p(A, C) * p(C, A) * (p(G, G) * p(T, T) - p(T, G) * p(G, T)) +
				 p(A, C) * p(C, G) * (p(G, A) * p(T, T) - p(T, A) * p(G, T)) -
				 p(A, C) * p(C, T) * (p(G, A) * p(T, G) - p(T, A) * p(G, G)) +

				 p(A, G) * p(C, A) * (p(G, C) * p(T, T) - p(T, C) * p(G, T)) -
				 p(A, G) * p(C, C) * (p(G, A) * p(T, T) - p(T, A) * p(G, T)) +
				 p(A, G) * p(C, T) * (p(G, A) * p(T, C) - p(T, A) * p(G, C)) -

				 p(A, T) * p(C, A) * (p(G, C) * p(T, G) - p(T, C) * p(G, G)) +
				 p(A, T) * p(C, C) * (p(G, A) * p(T, G) - p(T, A) * p(G, G)) -
				 p(A, T) * p(C, G) * (p(G, A) * p(T, C) - p(T, A) * p(G, C));
#undef p
	double d = -0.25 * (log(det) - 0.5 * lg);
	return d <= 0.0 ? 0.0 : d;
}

/* estimate_ANI, src/model.c:207-210 */
static double ani(const andi_hip_model *m) {
	return (1.0 - raw(m)) * 100;
}

/* dispatch as in print_distances, src/io.c:259-268 (JC is the default) */
double andi_hip_estimate(const andi_hip_model *m, int model) {
	switch (model) {
		case ANDI_M_RAW: return raw(m);
		case ANDI_M_KIMURA: return kimura(m);
		case ANDI_M_LOGDET: return logdet(m);
		case ANDI_M_ANI: return ani(m);
		default: return jc(m);
	}
}

typedef struct {
	char *buf;
	size_t cap, len;
} sink;

static void put(sink *s, const char *fmt, ...) {
	va_list ap;
	va_start(ap, fmt);
	size_t room = s->len < s->cap ? s->cap - s->len : 0;
	int w = vsnprintf(room ? s->buf + s->len : NULL, room, fmt, ap);
	va_end(ap);
	if (w > 0) s->len += (size_t)w;
}

/* print_distances, src/io.c:246-322 — same averaging rule, scientific-notation
 * switch, warnings and row format, written into buffers instead of stdout /
 * stderr so the caller decides where they go.
 *
 * Row-parallel (BASELINE's config 3 is a 3085 x 3085 matrix: 9.5 M estimates and as many printf conversions, seconds on one
 * thread beside a 13 s matrix): the rows are dealt to a pool of threads twice -- first the distances, the warnings of
 * each row and whether any distance calls for the scientific format (a property of the WHOLE matrix, src/io.c:283), then
 * the rows' text -- and put together in row order, so the bytes are those of the sequential loop. */
typedef struct {
	const andi_hip_model *M;
	const char *const *names;
	size_t n;
	int model, extra_verbose, truncate_names, warnings;
	double *D;
	sink *row_warn, *row_text; /* one growing buffer per row */
	int *row_flags, *row_sci;
	int scientific;
	int phase;
	size_t next; /* atomic: the next block of rows */
} fmt_job;

static void put_grow(sink *s, const char *fmt, ...) { /* a sink that owns its buffer */
	for (;;) {
		va_list ap;
		va_start(ap, fmt);
		const size_t room = s->cap - s->len;
		const int w = vsnprintf(room ? s->buf + s->len : NULL, room, fmt, ap);
		va_end(ap);
		if (w < 0) return;
		if ((size_t)w < room) {
			s->len += (size_t)w;
			return;
		}
		s->cap = 2 * s->cap + (size_t)w + 64;
		s->buf = realloc(s->buf, s->cap);
		if (!s->buf) abort();
	}
}

static void fmt_rows(fmt_job *job, size_t i0, size_t i1) {
	const andi_hip_model *M = job->M;
	const size_t n = job->n;
	for (size_t i = i0; i < i1; i++) {
		if (job->phase == 0) {
			int flags = 0, sci = 0;
			sink *w = &job->row_warn[i];
			for (size_t j = 0; j < n; j++) {
				andi_hip_model datum = M[i * n + j];
				if (!job->extra_verbose) datum = andi_hip_model_average(&M[i * n + j], &M[j * n + i]);
				double d = job->D[i * n + j] = i == j ? 0.0 : andi_hip_estimate(&datum, job->model);
				if (d > 0 && d < 0.001) sci = 1;
				if (isnan(d) && job->warnings) {
					flags |= 1;
					put_grow(w,
							 "For the two sequences '%s' and '%s' the distance computation failed and "
							 "is reported as nan. Please refer to the documentation for further "
							 "details.\n",
							 job->names[i], job->names[j]);
				}
				if (!isnan(d) && i < j && job->warnings) {
					double c1 = andi_hip_model_coverage(&M[i * n + j]);
					double c2 = andi_hip_model_coverage(&M[j * n + i]);
					if (c1 < 0.2 || c2 < 0.2) {
						flags |= 2;
						put_grow(w,
								 "For the two sequences '%s' and '%s' very little homology was found "
								 "(%f and %f, respectively).\n",
								 job->names[i], job->names[j], c1, c2);
					}
				}
			}
			job->row_flags[i] = flags, job->row_sci[i] = sci;
		} else {
			sink *o = &job->row_text[i];
			put_grow(o, job->truncate_names ? "%-10.10s" : "%-10s", job->names[i]);
			const char *f = job->scientific ? " %1.4e" : " %1.4f";
			for (size_t j = 0; j < n; j++) put_grow(o, f, job->D[i * n + j]);
			put_grow(o, "\n");
		}
	}
}

#include <pthread.h>
#include <unistd.h>
static void *fmt_worker(void *arg) {
	fmt_job *job = arg;
	for (;;) {
		const size_t i0 = __atomic_fetch_add(&job->next, 16, __ATOMIC_RELAXED);
		if (i0 >= job->n) break;
		fmt_rows(job, i0, i0 + 16 < job->n ? i0 + 16 : job->n);
	}
	return NULL;
}

static void fmt_run(fmt_job *job, int phase) {
	job->phase = phase, job->next = 0;
	long procs = sysconf(_SC_NPROCESSORS_ONLN);
	size_t nt = procs > 0 ? (size_t)procs : 1;
	if (nt > 32) nt = 32;
	if (nt > job->n / 64 + 1) nt = job->n / 64 + 1; /* (small matrices: the calling thread alone) */
	pthread_t tid[32];
	size_t started = 0;
	for (size_t t = 1; t < nt; t++)
		if (pthread_create(&tid[started], NULL, fmt_worker, job) == 0) started++;
	fmt_worker(job);
	for (size_t t = 0; t < started; t++) pthread_join(tid[t], NULL);
}

size_t andi_hip_format_distances(const andi_hip_model *M, const char *const *names, size_t n,
								 int model, int extra_verbose, int truncate_names, int warnings,
								 char *out, size_t cap, char *warnbuf, size_t warncap,
								 int *warn_flags) {
	fmt_job job = {M, names, n, model, extra_verbose, truncate_names, warnings, NULL, NULL, NULL, NULL, NULL, 0, 0, 0};
	job.D = malloc((n ? n * n : 1) * sizeof *job.D);
	job.row_warn = calloc(n ? n : 1, sizeof *job.row_warn);
	job.row_text = calloc(n ? n : 1, sizeof *job.row_text);
	job.row_flags = calloc(n ? n : 1, sizeof *job.row_flags);
	job.row_sci = calloc(n ? n : 1, sizeof *job.row_sci);
	if (!job.D || !job.row_warn || !job.row_text || !job.row_flags || !job.row_sci) {
		free(job.D), free(job.row_warn), free(job.row_text), free(job.row_flags), free(job.row_sci);
		return 0;
	}
	sink o = {out, cap, 0}, w = {warnbuf, warncap, 0};
	int flags = 0;

	fmt_run(&job, 0);
	for (size_t i = 0; i < n; i++) job.scientific |= job.row_sci[i], flags |= job.row_flags[i];
	fmt_run(&job, 1);

	put(&o, "%zu\n", n);
	for (size_t i = 0; i < n; i++) {
		const sink *r = &job.row_text[i], *rw = &job.row_warn[i];
		if (o.len < o.cap && r->len) memcpy(o.buf + o.len, r->buf, r->len < o.cap - o.len ? r->len : o.cap - o.len);
		o.len += r->len;
		if (w.len < w.cap && rw->len) memcpy(w.buf + w.len, rw->buf, rw->len < w.cap - w.len ? rw->len : w.cap - w.len);
		w.len += rw->len;
		free(r->buf), free(rw->buf);
	}
	free(job.D), free(job.row_warn), free(job.row_text), free(job.row_flags), free(job.row_sci);
	if (out && cap) out[o.len < cap ? o.len : cap - 1] = '\0';
	if (warnbuf && warncap) warnbuf[w.len < warncap ? w.len : warncap - 1] = '\0';
	if (warn_flags) *warn_flags = flags;
	return o.len;
}
