/*
 * host_seq.c — the host side of subject preparation.
 *
 * These steps stay on the CPU by design: they are O(len)
 * byte passes and a handful of libm calls whose doubles must come out of the
 * same libm as the reference's.  Follows src/sequence.c:143-219, 296-373.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "andi_hip.h"

void andi_hip_free(void *p) {
	free(p);
}

/* binomial_coefficient, src/sequence.c:315-335 (size_t arithmetic, the
 * multiply-then-divide order matters for the integer results) */
static size_t choose(size_t n, size_t k) {
	if (n == 0 || k > n) return 0;
	if (k == 0 || k == n) return 1;
	if (n - k < k) k = n - k;
	size_t acc = 1;
	for (size_t i = 1; i <= k; i++) {
		acc *= n - k + i;
		acc /= i;
	}
	return acc;
}

/* shustring_cum_prob, src/sequence.c:353-373: P{shustring length <= x} for a
 * random sequence of length l with half-GC-content p.  The grouping of the
 * floating-point products is the reference's. */
double andi_hip_shustring_cum_prob(size_t x, double p, size_t l) {
	double xd = (double)x;
	double ld = (double)l;
	double sum = 0.0;
	for (size_t k = 0; k <= x; k++) {
		double kd = (double)k;
		double t = pow(p, kd) * pow(0.5 - p, xd - kd);
		sum += pow(2, xd) * (t * pow(1 - t, ld)) * (double)choose(x, k);
		if (sum >= 1.0) return 1.0;
	}
	return sum;
}

/* min_anchor_length, src/sequence.c:296-304 */
size_t andi_hip_min_anchor_length(double p, double g, size_t l) {
	size_t x = 1;
	for (; andi_hip_shustring_cum_prob(x, g / 2, l) < 1 - p; x++) {
	}
	return x;
}

/* seq_subject_init, src/sequence.c:210-219, with calc_gc (197-208), revcomp
 * (143-168) and catcomp (177-190) folded into two passes over the sequence. */
int andi_hip_subject_prepare(const char *seq, size_t len, double p_value, char **RS_out,
							 size_t *RSlen_out, double *gc_out, size_t *threshold_out) {
	if (!seq || !RS_out || len == 0) return 1;
	char *rs = malloc(2 * len + 2);
	if (!rs) return 2;

	size_t gc = 0;
	char *fwd = rs + len + 1;
	for (size_t i = 0; i < len; i++) {
		char c = seq[i];
		gc += (c == 'G') | (c == 'C');
		fwd[i] = c;
		/* complement: A<->T via ^21, C<->G via ^4; separators (< 'A') map to ';' */
		rs[len - 1 - i] = c < 'A' ? ';' : (char)(c ^ ((c & 2) ? 4 : 21));
	}
	rs[len] = '#';
	rs[2 * len + 1] = '\0';

	size_t rslen = 2 * len + 1;
	double g = (double)gc / len;
	*RS_out = rs;
	if (RSlen_out) *RSlen_out = rslen;
	if (gc_out) *gc_out = g;
	if (threshold_out) *threshold_out = andi_hip_min_anchor_length(p_value, g, rslen);
	return 0;
}
