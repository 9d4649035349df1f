/*
 * andi_oracle.c — CPU restatement (plain C) of the andi hot path:
 * subject preparation → enhanced suffix array → anchor scan → 4x4 counts →
 * distance estimators.
 *
 * TEST INFRASTRUCTURE ONLY (see andi_oracle.h).  This is a "port"-kind oracle:
 * the upstream tree cannot be built in this image without writing stand-ins
 * (it needs libdivsufsort, GSL and an autoconf-generated config.h, all absent),
 * so the algorithm is restated here function by function with the reference
 * file:line each one follows.
 *
 * PARITY PINS (tests/test_oracle_pins.py):
 *   - test/test_seq.c:34,69   RS strings for "ACGTTGCA" and "ACGT!TGCA", gc
 *   - test/test_process.c:16-29  minimality of min_anchor_length
 *   - test/test_esa.c:32-44,107-192  cached == uncached on (l,i,j) for the two
 *     200-nt fixtures, hand-picked strings and all 4^11 11-mers; match is a
 *     true, maximal prefix match
 *   - SURVEY.md §8c worked example (SA/LCP/FVC/get_match for TGCAACGT#ACGTTGCA)
 *   - SURVEY.md §6.2 / BASELINE.md §2: numbers the unmodified reference
 *     produced in this image on inputs from its own generator
 *     (test/test_fasta.cxx, rebuilt into oracle/_ref/): loop iterations, probe
 *     / lucky / pair / gap-character counts of dist_anchor for -s 42 -l 1e6
 *     at d = 0.1 / 0.01 / 0.001, and the 4-decimal PHYLIP distances for
 *     -s 1729 -l 1e6 -d 0.1 -d 0.1 under RAW and JC.
 * Bootstrap (model.c:222-232): GSL's published multinomial restated at the end of this file; no pin anywhere
 * (the reference seeds from the clock and has no test): "parity unpinned", the distribution is what is compared.
 */
#include "andi_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_CACHE_K 10 /* CACHE_LENGTH, esa.c:35 */

/* ====================================================================== */
/* sequence.c                                                             */
/* ====================================================================== */

/* sequence.c:260-282 — keep ACGT!, upper-case acgt, drop the rest. */
size_t orc_normalize(char *s, int *non_acgt) {
	char *w = s;
	int dropped = 0;
	for (const char *r = s; *r; ++r) {
		char c = *r;
		if (c == 'A' || c == 'C' || c == 'G' || c == 'T' || c == '!') {
			*w++ = c;
		} else if (c == 'a' || c == 'c' || c == 'g' || c == 't') {
			*w++ = (char)(c - 'a' + 'A');
		} else {
			dropped = 1;
		}
	}
	*w = '\0';
	if (non_acgt && dropped) *non_acgt = 1;
	return (size_t)(w - s);
}

/* sequence.c:143-168 — reverse complement; bytes below 'A' become ';'. */
char *orc_revcomp(const char *s, size_t len) {
	char *out = malloc(len + 1);
	if (!out) return NULL;
	for (size_t k = 0; k < len; ++k) {
		char c = s[len - 1 - k];
		if (c < 'A') {
			out[k] = ';';
		} else {
			out[k] = (char)(c ^ ((c & 2) ? 4 : 21));
		}
	}
	out[len] = '\0';
	return out;
}

/* sequence.c:177-190 — RS = revcomp(S) '#' S '\0'. */
char *orc_catcomp(const char *s, size_t len) {
	char *rev = orc_revcomp(s, len);
	if (!rev) return NULL;
	char *rs = realloc(rev, 2 * len + 2);
	if (!rs) {
		free(rev);
		return NULL;
	}
	rs[len] = '#';
	memcpy(rs + len + 1, s, len);
	rs[2 * len + 1] = '\0';
	return rs;
}

/* sequence.c:197-208 */
double orc_gc(const char *s, size_t len) {
	size_t gc = 0;
	for (const char *p = s; *p; ++p) gc += (*p == 'G' || *p == 'C');
	return (double)gc / len;
}

/* sequence.c:315-335 */
size_t orc_binomial(size_t n, size_t k) {
	if (n == 0 || k > n) return 0;
	if (k == 0 || k == n) return 1;
	if (k > n - k) k = n - k;
	size_t r = 1;
	for (size_t i = 1; i <= k; ++i) {
		r *= n - k + i;
		r /= i;
	}
	return r;
}

/* sequence.c:353-373 — the association of the products is kept as in the
 * reference so the doubles come out bit-identical. */
double orc_shustring_cum_prob(size_t x, double p, size_t l) {
	const double xx = (double)x, ll = (double)l;
	double s = 0.0;
	for (size_t k = 0; k <= x; ++k) {
		double kk = (double)k;
		double t = pow(p, kk) * pow(0.5 - p, xx - kk);
		s += pow(2, xx) * (t * pow(1 - t, ll)) * (double)orc_binomial(x, k);
		if (s >= 1.0) {
			s = 1.0;
			break;
		}
	}
	return s;
}

/* sequence.c:296-304 */
size_t orc_min_anchor_length(double p, double g, size_t l) {
	size_t x = 1;
	while (orc_shustring_cum_prob(x, g / 2, l) < 1 - p) ++x;
	return x;
}

/* sequence.c:210-219 */
int orc_subject_init(orc_subject *sub, const char *s, size_t len, double p_value) {
	sub->gc = orc_gc(s, len);
	sub->RS = orc_catcomp(s, len);
	if (!sub->RS) return 1;
	sub->RSlen = 2 * len + 1;
	sub->threshold = orc_min_anchor_length(p_value, sub->gc, sub->RSlen);
	return 0;
}

void orc_subject_free(orc_subject *sub) {
	free(sub->RS);
	memset(sub, 0, sizeof *sub);
}

/* ====================================================================== */
/* suffix array — stands in for divsufsort (esa.c:303).  A suffix array is  */
/* unique, so any correct sorter yields the same downstream bits.  This one */
/* is a re-entrant multikey quicksort over unsigned bytes; T[n] must be 0.  */
/* ====================================================================== */

static inline unsigned mk_key(const unsigned char *T, int32_t s, int32_t d) {
	return T[s + d];
}

static void mk_insertion(const unsigned char *T, int32_t *a, int32_t n, int32_t d) {
	for (int32_t i = 1; i < n; ++i) {
		int32_t v = a[i];
		int32_t j = i;
		while (j > 0) {
			const unsigned char *x = T + a[j - 1] + d, *y = T + v + d;
			while (*x == *y) { /* distinct suffixes always differ before/at the NUL */
				++x;
				++y;
			}
			if (*x < *y) break;
			a[j] = a[j - 1];
			--j;
		}
		a[j] = v;
	}
}

static void mk_sort(const unsigned char *T, int32_t *a, int32_t n, int32_t d) {
	while (n > 1) {
		if (n < 12) {
			mk_insertion(T, a, n, d);
			return;
		}
		/* median of three */
		unsigned k0 = mk_key(T, a[0], d), k1 = mk_key(T, a[n / 2], d),
				 k2 = mk_key(T, a[n - 1], d);
		unsigned pv = k0 < k1 ? (k1 < k2 ? k1 : (k0 < k2 ? k2 : k0))
							  : (k0 < k2 ? k0 : (k1 < k2 ? k2 : k1));
		/* three-way partition: [0,lt) < pv, [lt,gt) == pv, [gt,n) > pv */
		int32_t lt = 0, gt = n, i = 0;
		while (i < gt) {
			unsigned k = mk_key(T, a[i], d);
			if (k < pv) {
				int32_t t = a[lt];
				a[lt] = a[i];
				a[i] = t;
				++lt;
				++i;
			} else if (k > pv) {
				--gt;
				int32_t t = a[gt];
				a[gt] = a[i];
				a[i] = t;
			} else {
				++i;
			}
		}
		mk_sort(T, a, lt, d);
		mk_sort(T, a + gt, n - gt, d);
		/* continue with the equal part one character deeper; a NUL pivot can
		 * only be shared by one suffix, so this terminates. */
		a += lt;
		n = gt - lt;
		if (pv == 0) return;
		++d;
	}
}

int orc_suffix_array(const unsigned char *T, int32_t *SA, int32_t n) {
	if (!T || !SA || n < 0) return -1;
	/* one counting pass on the first byte keeps the recursion shallow */
	int32_t cnt[257] = {0};
	for (int32_t i = 0; i < n; ++i) cnt[T[i] + 1]++;
	for (int c = 0; c < 256; ++c) cnt[c + 1] += cnt[c];
	int32_t start[257];
	memcpy(start, cnt, sizeof start);
	for (int32_t i = 0; i < n; ++i) SA[cnt[T[i]]++] = i;
	for (int c = 0; c < 256; ++c) {
		int32_t lo = start[c], hi = start[c + 1];
		if (hi - lo > 1) mk_sort(T, SA + lo, hi - lo, 1);
	}
	return 0;
}

/* ====================================================================== */
/* esa.c                                                                  */
/* ====================================================================== */

/* esa.c:49-58 */
static inline int code_of(char c) {
	switch (c) {
		case 'A': return 0;
		case 'C': return 1;
		case 'G': return 2;
		case 'T': return 3;
	}
	return -1;
}

/* esa.c:373-426 — LCP through the PHI/PLCP arrays. */
static int build_lcp(orc_esa *E) {
	const char *S = E->S;
	const int32_t *SA = E->SA;
	const int32_t n = E->len;
	if (!S || !SA || n == 0) return 1;
	int32_t *LCP = E->LCP = malloc(((size_t)n + 1) * sizeof *LCP);
	int32_t *phi = malloc((size_t)n * sizeof *phi);
	if (!LCP || !phi) {
		free(phi);
		return 1;
	}
	LCP[0] = LCP[n] = -1;
	phi[SA[0]] = -1;
	for (int32_t r = 1; r < n; ++r) phi[SA[r]] = SA[r - 1];
	/* phi is overwritten in place by the permuted LCP */
	int64_t h = 0;
	for (int32_t t = 0; t < n; ++t) {
		int32_t prev = phi[t];
		if (prev == -1) {
			phi[t] = -1;
			continue;
		}
		while (S[prev + h] == S[t + h]) ++h;
		phi[t] = (int32_t)h;
		if (--h < 0) h = 0;
	}
	for (int32_t r = 1; r < n; ++r) LCP[r] = phi[SA[r]];
	free(phi);
	return 0;
}

/* esa.c:312-363 — child table in one array: CLD[i] = next l-index or "down",
 * CLD[k-1] = "up" of k.  Slots the sweep never writes are left at -1 here
 * (uninitialised in the reference; never consulted). */
static int build_cld(orc_esa *E) {
	const int32_t n = E->len;
	const int32_t *LCP = E->LCP;
	int32_t *CLD = E->CLD = malloc(((size_t)n + 1) * sizeof *CLD);
	int32_t *stk_idx = malloc(((size_t)n + 1) * sizeof *stk_idx);
	int32_t *stk_lcp = malloc(((size_t)n + 1) * sizeof *stk_lcp);
	if (!CLD || !stk_idx || !stk_lcp) {
		free(stk_idx);
		free(stk_lcp);
		return 1;
	}
	memset(CLD, 0xff, ((size_t)n + 1) * sizeof *CLD);
	CLD[0] = n;
	int64_t top = 0;
	stk_idx[0] = 0;
	stk_lcp[0] = -1;
	for (int32_t k = 1; k <= n; ++k) {
		const int32_t cur = LCP[k];
		while (cur < stk_lcp[top]) {
			int32_t li = stk_idx[top], ll = stk_lcp[top];
			--top;
			while (stk_lcp[top] == ll) { /* chain equal-lcp entries */
				CLD[stk_idx[top]] = li;
				li = stk_idx[top];
				--top;
			}
			if (cur < stk_lcp[top]) {
				CLD[stk_idx[top]] = li; /* down */
			} else {
				CLD[k - 1] = li; /* up */
			}
		}
		++top;
		stk_idx[top] = k;
		stk_lcp[top] = cur;
	}
	free(stk_idx);
	free(stk_lcp);
	return 0;
}

/* esa.c:229-245 — FVC[i] = S[SA[i] + LCP[i]] (i = 0 reads S[SA[0]-1]; that
 * slot is never consulted). */
static int build_fvc(orc_esa *E) {
	const int32_t n = E->len;
	char *F = E->FVC = malloc((size_t)n);
	if (!F) return 1;
	for (int32_t i = 0; i < n; ++i) F[i] = E->S[E->SA[i] + E->LCP[i]];
	return 0;
}

/* esa.c:441-511 */
static orc_interval child_interval(const orc_esa *E, orc_interval ij, char a) {
	const char *S = E->S;
	const int32_t *SA = E->SA, *LCP = E->LCP, *CLD = E->CLD;
	const char *FVC = E->FVC;
	int32_t i = ij.i;
	const int32_t j = ij.j;

	if (i == j) {
		if (S[SA[i] + ij.l] != a) ij.i = ij.j = -1;
		return ij;
	}

	int32_t m = ij.m;
	const int32_t l = ij.l;
	char c = S[SA[i] + l];
	for (;;) {
		if (c == a) {
			orc_interval r;
			if (i != m - 1) {
				int32_t nm = CLD[m - 1];
				r.i = i, r.j = m - 1, r.m = nm, r.l = LCP[nm];
			} else {
				r.i = i, r.j = i, r.m = -1, r.l = LCP[i];
			}
			return r;
		}
		if (c > a) break;
		i = m;
		if (i == j) break;
		m = CLD[m];
		if (LCP[m] != l) break;
		c = FVC[i];
	}

	int hit = (i != ij.i) ? (FVC[i] == a) : (S[SA[i] + l] == a);
	if (hit) {
		ij.i = i;
		ij.l = LCP[m];
		ij.m = m;
	} else {
		ij.i = ij.j = -1;
	}
	return ij;
}

/* esa.c:531-601 */
static orc_interval match_from(const orc_esa *E, const char *q, size_t qlen,
							   int32_t k, orc_interval ij) {
	if (ij.i == -1 && ij.j == -1) return ij;
	const char *S = E->S;
	const int32_t *SA = E->SA;

	if (ij.i == ij.j) {
		int32_t p = SA[ij.i];
		size_t kk = (size_t)ij.l;
		while (kk < qlen && S[p + kk] && S[p + kk] == q[kk]) ++kk;
		ij.l = (int32_t)kk;
		return ij;
	}

	orc_interval res = ij;
	do {
		ij = child_interval(E, ij, q[k]);
		if (ij.i == -1 && ij.j == -1) {
			res.l = k;
			return res;
		}
		res.i = ij.i;
		res.j = ij.j;

		int32_t lim = (int32_t)qlen;
		if (ij.i < ij.j && ij.l < lim) lim = ij.l;
		++k;
		for (int32_t p = SA[ij.i]; k < lim; ++k) {
			if (S[p + k] != q[k]) {
				res.l = k;
				return res;
			}
		}
	} while (k < (int32_t)qlen);
	res.l = (int32_t)qlen;
	return res;
}

static inline orc_interval root_interval(const orc_esa *E) {
	orc_interval r;
	r.i = 0;
	r.j = E->len - 1;
	r.m = E->CLD[E->len - 1]; /* L(CLD, len) */
	r.l = E->LCP[r.m];
	return r;
}

/* esa.c:615-624 */
orc_interval orc_get_match(const orc_esa *E, const char *q, size_t qlen) {
	if (!E || !q || !E->len || !E->SA || !E->LCP || !E->S || !E->CLD) {
		orc_interval bad = {-1, -1, -1, -1};
		return bad;
	}
	return match_from(E, q, qlen, 0, root_interval(E));
}

/* esa.c:636-656 */
orc_interval orc_get_match_cached(const orc_esa *E, const char *q, size_t qlen) {
	if (qlen <= ORC_CACHE_K) return orc_get_match(E, q, qlen);
	long code = 0;
	for (int t = 0; t < ORC_CACHE_K && code >= 0; ++t) {
		int c = code_of(q[t]);
		code = c < 0 ? -1 : ((code << 2) | c);
	}
	if (code < 0) return orc_get_match(E, q, qlen);
	orc_interval ij = E->cache[code];
	if (ij.i == -1 && ij.j == -1) return orc_get_match(E, q, qlen);
	return match_from(E, q, qlen, ij.l, ij);
}

/* esa.c:193-215 — write `v` into every table slot whose 10-mer starts with
 * the `depth` characters in str. */
static void cache_fill(orc_esa *E, const char *str, size_t depth, orc_interval v) {
	size_t code = 0;
	for (size_t t = 0; t < depth; ++t) code = (code << 2) | (size_t)code_of(str[t]);
	size_t span = (size_t)1 << (2 * (ORC_CACHE_K - depth));
	orc_interval *dst = E->cache + code * span;
	for (size_t t = 0; t < span; ++t) dst[t] = v;
}

/* esa.c:103-191 — depth-first walk over ACGT prefixes up to length 10 */
static void cache_dfs(orc_esa *E, char *str, size_t pos, orc_interval in) {
	static const char ACGT[4] = {'A', 'C', 'G', 'T'};
	if (pos >= ORC_CACHE_K || (in.i == -1 && in.j == -1)) {
		cache_fill(E, str, pos < ORC_CACHE_K ? pos : ORC_CACHE_K, in);
		return;
	}
	for (int code = 0; code < 4; ++code) {
		str[pos] = ACGT[code];
		orc_interval ij = child_interval(E, in, str[pos]);

		if (ij.i == -1 && ij.j == -1) { /* prefix+char absent: keep parent */
			cache_fill(E, str, pos + 1, in);
			continue;
		}
		if (ij.i == ij.j) { /* singleton: depth is exactly pos+1 */
			ij.l = (int32_t)(pos + 1);
			cache_fill(E, str, pos + 1, ij);
			continue;
		}
		if (ij.l <= (int32_t)(pos + 1)) { /* usual case */
			cache_dfs(E, str, pos + 1, ij);
			continue;
		}
		if ((size_t)ij.l >= ORC_CACHE_K) { /* deeper than the table: stop */
			cache_fill(E, str, pos + 1, in);
			continue;
		}
		/* the interval is deeper than one character but still inside the
		 * table: everything below str[0..pos] gets the parent, then the one
		 * existing elongation is followed. */
		cache_fill(E, str, pos + 1, in);
		int sep = 0;
		size_t k = pos + 1;
		for (; k < (size_t)ij.l; ++k) {
			char c = E->S[E->SA[ij.i] + k];
			if (code_of(c) < 0) {
				sep = 1;
				break;
			}
			str[k] = c;
		}
		if (sep) {
			cache_fill(E, str, k, ij);
		} else {
			cache_dfs(E, str, k, ij);
		}
	}
}

/* esa.c:73-88 */
static int build_cache(orc_esa *E) {
	E->cache = malloc(((size_t)1 << (2 * ORC_CACHE_K)) * sizeof *E->cache);
	if (!E->cache) return 1;
	char str[ORC_CACHE_K + 1];
	str[ORC_CACHE_K] = '\0';
	cache_dfs(E, str, 0, root_interval(E));
	return 0;
}

/* esa.c:254-277 */
static int esa_finish(orc_esa *E) {
	if (build_lcp(E)) return 1;
	if (build_cld(E)) return 1;
	if (build_fvc(E)) return 1;
	if (build_cache(E)) return 1;
	return 0;
}

int orc_esa_init(orc_esa *E, const orc_subject *sub) {
	if (!E || !sub || !sub->RS) return 1;
	memset(E, 0, sizeof *E);
	E->S = sub->RS;
	E->len = (int32_t)sub->RSlen;
	E->SA = malloc((size_t)E->len * sizeof *E->SA);
	if (!E->SA) return 1;
	if (orc_suffix_array((const unsigned char *)E->S, E->SA, E->len)) return 1;
	return esa_finish(E);
}

int orc_esa_init_with_sa(orc_esa *E, const orc_subject *sub, const int32_t *SA) {
	if (!E || !sub || !sub->RS || !SA) return 1;
	memset(E, 0, sizeof *E);
	E->S = sub->RS;
	E->len = (int32_t)sub->RSlen;
	E->SA = malloc((size_t)E->len * sizeof *E->SA);
	if (!E->SA) return 1;
	memcpy(E->SA, SA, (size_t)E->len * sizeof *E->SA);
	return esa_finish(E);
}

void orc_esa_free(orc_esa *E) {
	free(E->SA);
	free(E->LCP);
	free(E->CLD);
	free(E->cache);
	free(E->FVC);
	memset(E, 0, sizeof *E);
}

/* ====================================================================== */
/* model.c                                                                */
/* ====================================================================== */

/* model.c:295-299: A0 C1 G2 T3 from bits 1..2 */
static inline unsigned nt2bits(unsigned char c) {
	c &= 6;
	c ^= c >> 1;
	return c >> 1;
}

/* model.c:309-337 */
void orc_model_count(orc_model *m, const char *s, const char *q, size_t len) {
	size_t local[16] = {0};
	for (size_t t = 0; t < len; ++t) {
		char a = s[t], b = q[t];
		if (a < 'A' || b < 'A') continue;
		local[(nt2bits((unsigned char)a) << 2) + nt2bits((unsigned char)b)]++;
	}
	for (int t = 0; t < 16; ++t) m->counts[t] += (uint32_t)local[t];
}

/* model.c:246-279 */
void orc_model_count_equal(orc_model *m, const char *s, size_t len, int model) {
	if (model == ORC_M_RAW || model == ORC_M_JC || model == ORC_M_KIMURA) {
		size_t q = len / 4;
		m->counts[0] += (uint32_t)q;
		m->counts[5] += (uint32_t)q;
		m->counts[10] += (uint32_t)q;
		m->counts[15] += (uint32_t)(q + (len & 3));
		return;
	}
	size_t local[4] = {0};
	for (size_t t = 0; t < len; ++t) {
		char c = s[t];
		if (c < 'A') continue;
		local[(c >> 1) & 3]++; /* A0 C1 T2 G3 */
	}
	m->counts[0] += (uint32_t)local[0];
	m->counts[5] += (uint32_t)local[1];
	m->counts[10] += (uint32_t)local[3];
	m->counts[15] += (uint32_t)local[2];
}

/* model.c:39-46 */
orc_model orc_model_average(const orc_model *a, const orc_model *b) {
	orc_model r = *a;
	for (int t = 0; t < 16; ++t) r.counts[t] += b->counts[t];
	r.seq_len += b->seq_len;
	return r;
}

/* model.c:54-60 */
size_t orc_model_total(const orc_model *m) {
	size_t tot = 0;
	for (int t = 0; t < 16; ++t) tot += m->counts[t];
	return tot;
}

/* model.c:68-73 */
double orc_model_coverage(const orc_model *m) {
	return (double)orc_model_total(m) / (double)m->seq_len;
}

static size_t sum_cells(const orc_model *m, const int *cells, int n) {
	size_t t = 0;
	for (int k = 0; k < n; ++k) t += m->counts[cells[k]];
	return t;
}

/* cell = 4*from + to */
enum { AA, AC, AG, AT, CA, CC, CG, CT, GA, GC, GG, GT, TA, TC, TG, TT };

/* model.c:81-92 */
static double est_raw(const orc_model *m) {
	static const int snp[12] = {AC, AG, AT, CA, CG, CT, GA, GC, GT, TA, TC, TG};
	size_t nucl = orc_model_total(m);
	size_t snps = sum_cells(m, snp, 12);
	if (nucl <= 3) return NAN;
	return (double)snps / (double)nucl;
}

/* model.c:100-106 */
static double est_jc(const orc_model *m) {
	double d = est_raw(m);
	d = -0.75 * log(1.0 - (4.0 / 3.0) * d);
	return d <= 0.0 ? 0.0 : d;
}

/* model.c:113-127 */
static double est_kimura(const orc_model *m) {
	static const int ts[4] = {AG, GA, CT, TC};
	static const int tv[8] = {AC, CA, AT, TA, GC, CG, GT, TG};
	size_t nucl = orc_model_total(m);
	double P = (double)sum_cells(m, ts, 4) / (double)nucl;
	double Q = (double)sum_cells(m, tv, 8) / (double)nucl;
	double tmp = 1.0 - 2.0 * P - Q;
	double d = -0.25 * log((1.0 - 2.0 * Q) * tmp * tmp);
	return d <= 0.0 ? 0.0 : d;
}

/* model.c:155-199 */
static double est_logdet(const orc_model *m) {
	double nucl = (double)orc_model_total(m);
	double P[16];
	for (int t = 0; t < 16; ++t) P[t] = m->counts[t] / nucl;
	double ld = 0.0;
	/* row sums then column sums, same order of additions as the reference */
	for (int r = 0; r < 4; ++r) {
		int cells[4] = {4 * r, 4 * r + 1, 4 * r + 2, 4 * r + 3};
		double term = log(sum_cells(m, cells, 4) / nucl);
		ld = r == 0 ? term : ld + term;
	}
	for (int c = 0; c < 4; ++c) {
		int cells[4] = {c, 4 + c, 8 + c, 12 + c};
		ld = ld + log(sum_cells(m, cells, 4) / nucl);
	}
	double det =
		P[AA] * P[CC] * (P[GG] * P[TT] - P[TG] * P[GT]) -
		P[AA] * P[CG] * (P[GC] * P[TT] - P[TC] * P[GT]) +
		P[AA] * P[CT] * (P[GC] * P[TG] - P[TC] * P[GG]) -

		P[AC] * P[CA] * (P[GG] * P[TT] - P[TG] * P[GT]) +
		P[AC] * P[CG] * (P[GA] * P[TT] - P[TA] * P[GT]) -
		P[AC] * P[CT] * (P[GA] * P[TG] - P[TA] * P[GG]) +

		P[AG] * P[CA] * (P[GC] * P[TT] - P[TC] * P[GT]) -
		P[AG] * P[CC] * (P[GA] * P[TT] - P[TA] * P[GT]) +
		P[AG] * P[CT] * (P[GA] * P[TC] - P[TA] * P[GC]) -

		P[AT] * P[CA] * (P[GC] * P[TG] - P[TC] * P[GG]) +
		P[AT] * P[CC] * (P[GA] * P[TG] - P[TA] * P[GG]) -
		P[AT] * P[CG] * (P[GA] * P[TC] - P[TA] * P[GC]);
	double d = -0.25 * (log(det) - 0.5 * ld);
	return d <= 0.0 ? 0.0 : d;
}

/* model.c:207-210 */
static double est_ani(const orc_model *m) {
	return (1.0 - est_raw(m)) * 100;
}

double orc_estimate(const orc_model *m, int model) {
	switch (model) {
		case ORC_M_RAW: return est_raw(m);
		case ORC_M_KIMURA: return est_kimura(m);
		case ORC_M_LOGDET: return est_logdet(m);
		case ORC_M_ANI: return est_ani(m);
		default: return est_jc(m);
	}
}

/* ====================================================================== */
/* process.c                                                              */
/* ====================================================================== */

typedef struct {
	size_t pos_S, pos_Q, length;
} anchor_t;

/* process.c:141-214 (with lcp 59-65, lucky_anchor 82-100, anchor 113-123) */
orc_model orc_dist_anchor(const orc_esa *E, const char *q, size_t qlen,
						  size_t thr, int model, orc_scan_stats *st) {
	orc_model ret;
	memset(&ret, 0, sizeof ret);
	ret.seq_len = (uint32_t)qlen;

	anchor_t cur = {0, 0, 0}, last = {0, 0, 0};
	int last_right = 0;
	const size_t n = (size_t)E->len;
	const size_t border = n / 2;
	const char *S = E->S;

	while (cur.pos_Q < qlen) {
		if (st) st->iterations++;
		int found = 0;

		/* lucky anchor: continue on the diagonal of the last anchor */
		size_t advance = cur.pos_Q - last.pos_Q;
		size_t gap = advance - last.length;
		size_t try_S = last.pos_S + advance;
		if (try_S < n && gap <= thr) {
			size_t rem = qlen - cur.pos_Q, len = 0;
			const char *a = q + cur.pos_Q, *b = S + try_S;
			while (len < rem && a[len] == b[len]) ++len;
			cur.pos_S = try_S;
			cur.length = len;
			found = len >= thr;
			if (st) {
				st->lucky_tries++;
				st->lucky_hits += (uint64_t)found;
			}
		}
		if (!found) {
			orc_interval in = orc_get_match_cached(E, q + cur.pos_Q, qlen - cur.pos_Q);
			if (st) st->esa_probes++;
			cur.pos_S = (size_t)E->SA[in.i];
			cur.length = in.l <= 0 ? 0 : (size_t)in.l;
			found = in.i == in.j && cur.length >= thr;
		}

		if (found) {
			size_t end_S = last.pos_S + last.length;
			size_t end_Q = last.pos_Q + last.length;
			if (cur.pos_S > end_S && cur.pos_Q - end_Q == cur.pos_S - end_S &&
				(cur.pos_S < border) == (last.pos_S < border)) {
				orc_model_count_equal(&ret, q + last.pos_Q, last.length, model);
				orc_model_count(&ret, S + end_S, q + end_Q, cur.pos_Q - end_Q);
				if (st) {
					st->anchor_pairs++;
					st->gap_chars += cur.pos_Q - end_Q;
				}
				last_right = 1;
			} else {
				if (last_right || last.length >= thr * 2) {
					orc_model_count_equal(&ret, q + last.pos_Q, last.length, model);
				}
				last_right = 0;
			}
			last = cur;
		}
		cur.pos_Q += cur.length + 1;
	}

	if (last.length >= qlen) { /* identical sequences */
		orc_model_count_equal(&ret, q, qlen, model);
		return ret;
	}
	if (last_right || last.length >= thr * 2) {
		orc_model_count_equal(&ret, q + last.pos_Q, last.length, model);
	}
	return ret;
}

static double now_s(void) {
	struct timespec ts;
	clock_gettime(CLOCK_MONOTONIC, &ts);
	return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

void orc_scan_row(orc_model *row, const orc_esa *E, size_t threshold,
				  const char *const *seqs, const size_t *lens, size_t n,
				  size_t self, int model, int threads) {
#ifdef _OPENMP
	if (threads <= 0) threads = omp_get_num_procs();
#else
	(void)threads;
#endif
#pragma omp parallel for num_threads(threads) schedule(dynamic, 1)
	for (size_t j = 0; j < n; ++j) {
		if (j == self) {
			memset(&row[j], 0, sizeof row[j]);
			row[j].seq_len = 9; /* dist_hack.h:61-64 */
			row[j].counts[0] = 9;
			continue;
		}
		row[j] = orc_dist_anchor(E, seqs[j], lens[j], threshold, model, NULL);
	}
}

/* dist_hack.h:34-96 (subject-parallel flavour) */
int orc_dist_matrix(orc_model *M, const char *const *seqs, const size_t *lens,
					size_t n, double p_value, int model, int threads,
					double *times_out) {
#ifdef _OPENMP
	if (threads <= 0) threads = omp_get_num_procs();
#else
	(void)threads;
#endif
	int failed = 0;
	double t_build = 0.0, t_scan = 0.0;
#pragma omp parallel for num_threads(threads) schedule(dynamic, 1) reduction(+ : t_build, t_scan)
	for (size_t i = 0; i < n; ++i) {
		orc_subject sub;
		orc_esa E;
		double t0 = now_s();
		if (orc_subject_init(&sub, seqs[i], lens[i], p_value) || orc_esa_init(&E, &sub)) {
#pragma omp atomic write
			failed = 1;
			continue;
		}
		double t1 = now_s();
		orc_scan_row(M + i * n, &E, sub.threshold, seqs, lens, n, i, model, 1);
		double t2 = now_s();
		t_build += t1 - t0;
		t_scan += t2 - t1;
		orc_esa_free(&E);
		orc_subject_free(&sub);
	}
	if (times_out) {
		times_out[0] = t_build;
		times_out[1] = t_scan;
	}
	return failed;
}

/* ====================================================================== */
/* bootstrap: model.c:222-232 + process.c:289-321                         */
/* ====================================================================== */
/*
 * model_bootstrap calls gsl_ran_multinomial(RNG, 16, N, p, counts) (model.c:229).  GSL is a third-party dependency
 * that is absent from the image (configure.ac:22-27, unpinned system library); its PUBLISHED algorithm (GSL manual,
 * "The Multinomial Distribution"; randist/multinomial.c, after C.S. Davis, "The computer generation of multinomial
 * random variates", Comp. Stat. Data Anal. 16 (1993) 205-217) is the conditional-binomial construction:
 *
 *     norm = sum_k p[k];  sum_p = 0;  sum_n = 0;
 *     for k = 0 .. K-1:
 *         n[k] = p[k] > 0 ? Binomial(p[k] / (norm - sum_p), N - sum_n) : 0;
 *         sum_p += p[k];  sum_n += n[k];
 *
 * restated here in the same order and the same double arithmetic.  The reference seeds GSL's generator from the
 * clock (andi.c:272-279) and holds no test for the bootstrap, so no stream can be reproduced: PARITY UNPINNED, the
 * DISTRIBUTION is what tests/test_bootstrap_gpu.py compares (chi-square of the device's cells against these draws and
 * against the exact binomial marginals).  The binomial sampler is exact inversion outward from the mode (any order of
 * enumerating the support gives Binomial(n, p) exactly; O(sqrt(n p q)) terms per draw), the generator splitmix64.
 */
static uint64_t orc_rng_next(uint64_t *s) {
	uint64_t z = (*s += 0x9E3779B97F4A7C15ull);
	z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
	z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
	return z ^ (z >> 31);
}
static double orc_rng_uniform(uint64_t *s) { /* (0, 1) */
	return ((double)(orc_rng_next(s) >> 11) + 0.5) * (1.0 / 9007199254740992.0);
}

uint64_t orc_ran_binomial(uint64_t *rng, double p, uint64_t n) {
	if (n == 0 || p <= 0.0) return 0;
	if (p >= 1.0) return n;
	const double q = 1.0 - p, nd = (double)n;
	double md = floor((nd + 1.0) * p);
	if (md > nd) md = nd;
	const double fm = exp(lgamma(nd + 1.0) - lgamma(md + 1.0) - lgamma(nd - md + 1.0) + md * log(p) + (nd - md) * log(q));
	double u = orc_rng_uniform(rng);
	/* enumerate m, m+1, m-1, m+2, m-2, ...; f(k+1) = f(k) (n-k)/(k+1) p/q, f(k-1) = f(k) k/(n-k+1) q/p */
	double up = md, dn = md, fu = fm, fd = fm;
	u -= fm;
	if (u < 0.0) return (uint64_t)md;
	for (;;) {
		int moved = 0;
		if (up < nd) {
			fu *= (nd - up) / (up + 1.0) * (p / q);
			up += 1.0;
			moved = 1;
			u -= fu;
			if (u < 0.0) return (uint64_t)up;
		}
		if (dn > 0.0) {
			fd *= dn / (nd - dn + 1.0) * (q / p);
			dn -= 1.0;
			moved = 1;
			u -= fd;
			if (u < 0.0) return (uint64_t)dn;
		}
		if (!moved || (fu < 1e-300 && fd < 1e-300)) return (uint64_t)md; /* rounding left a sliver of u: the mode */
	}
}

/* gsl_ran_multinomial as called at model.c:229 (K cells, N trials, weights p, result n) */
void orc_ran_multinomial(uint64_t *rng, size_t K, uint64_t N, const double *p, uint32_t *n) {
	double norm = 0.0, sum_p = 0.0;
	uint64_t sum_n = 0;
	for (size_t k = 0; k < K; ++k) norm += p[k];
	for (size_t k = 0; k < K; ++k) {
		if (p[k] > 0.0) {
			n[k] = (uint32_t)orc_ran_binomial(rng, p[k] / (norm - sum_p), N - sum_n);
		} else {
			n[k] = 0;
		}
		sum_p += p[k];
		sum_n += n[k];
	}
}

/* model_bootstrap, model.c:222-232 */
orc_model orc_model_bootstrap(orc_model datum, uint64_t *rng) {
	size_t nucl = orc_model_total(&datum);
	double p[16];
	for (size_t i = 0; i < 16; ++i) p[i] = datum.counts[i] / (double)nucl;
	orc_ran_multinomial(rng, 16, nucl, p, datum.counts);
	return datum;
}

/* calculate_bootstrap's loop body for ONE replicate, process.c:299-316: B is n*n, diagonal {counts[0] = 1, seq_len = 1},
 * B(j,i) = B(i,j) = model_bootstrap(model_average(M(i,j), M(j,i))) */
void orc_bootstrap_matrix(orc_model *B, const orc_model *M, size_t n, uint64_t seed) {
	uint64_t rng = seed;
	for (size_t i = 0; i < n; ++i) {
		for (size_t j = i; j < n; ++j) {
			if (i == j) {
				memset(&B[i * n + j], 0, sizeof(orc_model));
				B[i * n + j].counts[0] = 1;
				B[i * n + j].seq_len = 1;
				continue;
			}
			orc_model datum = orc_model_average(&M[i * n + j], &M[j * n + i]);
			if (orc_model_total(&datum) == 0) { /* (0/0 weights: GSL would be handed NaNs; an empty pair stays empty) */
				B[j * n + i] = B[i * n + j] = datum;
				continue;
			}
			datum = orc_model_bootstrap(datum, &rng);
			B[j * n + i] = B[i * n + j] = datum;
		}
	}
}
