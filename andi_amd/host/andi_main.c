/*
 * andi_main.c — command-line front end with andi's interface (multi-FASTA in,
 * PHYLIP matrix out; same options, warnings and exit codes as src/andi.c:63-394,
 * input handling of src/io.c:103-233 and src/sequence.c:78-125,234-282), driving
 * the MI355X engine through the C-ABI in include/andi_hip.h.  Written for this
 * project; only the observable behaviour follows the reference.
 */
#define _GNU_SOURCE
#include <err.h>
#include <ctype.h>
#include <errno.h>
#include <fcntl.h>
#include <getopt.h>
#include <limits.h>
#include <pthread.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <strings.h>
#include <time.h>
#include <unistd.h>

#include "andi_hip.h"

#define PROGRAM_VERSION "0.1"

typedef struct {
	char *name;
	char *seq;
	size_t len;
} genome;

typedef struct {
	genome *v;
	size_t n, cap;
} genome_list;

static int soft_error = 0; /* F_SOFT_ERROR, src/global.h:85-99 */
static int saw_non_acgt = 0;

#define soft_warnx(...)                                                                            \
	do {                                                                                           \
		soft_error = 1;                                                                            \
		warnx(__VA_ARGS__);                                                                        \
	} while (0)

static void *xmalloc(size_t n) {
	void *p = malloc(n ? n : 1);
	if (!p) err(errno, "Out of memory");
	return p;
}

static void push_genome(genome_list *l, genome g) {
	if (l->n == l->cap) {
		l->cap = l->cap ? l->cap * 3 / 2 + 1 : 4;
		l->v = realloc(l->v, l->cap * sizeof *l->v);
		if (!l->v) err(errno, "Out of memory");
	}
	l->v[l->n++] = g;
}

/* What reading ONE input file produced.  The files are read by a pool of threads (3085 assemblies, 6.5 GB: BASELINE's
 * config 3 -- SURVEY.md 8 f2 names ingest as the next wall-clock term); the main thread takes the results in the order of
 * the command line, so sequences, messages and flags come out as if the files had been read one after the other. */
typedef struct {
	genome_list seqs;
	char *msgs; /* the lines warnx()/warn() would have printed, in order */
	size_t mlen, mcap;
	int soft, non_acgt;
} file_result;

static void fmsg(file_result *r, int with_errno, const char *fmt, ...) {
	char line[4096];
	const int saved = errno;
	int k = snprintf(line, sizeof line, "%s: ", program_invocation_short_name);
	va_list ap;
	va_start(ap, fmt);
	k += vsnprintf(line + k, sizeof line - (size_t)k - 2, fmt, ap);
	va_end(ap);
	if (k > (int)sizeof line - 2) k = (int)sizeof line - 2;
	if (with_errno) k += snprintf(line + k, sizeof line - (size_t)k - 1, ": %s", strerror(saved));
	if (k > (int)sizeof line - 2) k = (int)sizeof line - 2;
	line[k++] = '\n';
	if (r->mlen + (size_t)k + 1 > r->mcap) {
		r->mcap = 2 * r->mcap + (size_t)k + 256;
		r->msgs = realloc(r->msgs, r->mcap);
		if (!r->msgs) err(errno, "Out of memory");
	}
	memcpy(r->msgs + r->mlen, line, (size_t)k);
	r->mlen += (size_t)k;
	r->msgs[r->mlen] = '\0';
}
#define file_soft_warnx(r, ...)                                                                    \
	do {                                                                                           \
		(r)->soft = 1;                                                                             \
		fmsg((r), 0, __VA_ARGS__);                                                                 \
	} while (0)

/* normalize, src/sequence.c:260-282: keep ACGT and '!', upper-case acgt, drop the rest (by table: a byte maps to itself,
 * to its upper case, or to 0 = dropped) */
static unsigned char norm_table[256];
static void norm_table_init(void) {
	const char *keep = "ACGT!";
	for (const char *k = keep; *k; k++) norm_table[(unsigned char)*k] = (unsigned char)*k;
	norm_table['a'] = 'A', norm_table['c'] = 'C', norm_table['g'] = 'G', norm_table['t'] = 'T';
}
static size_t normalize(char *s, size_t len, int *dropped) {
	char *w = s;
	int lost = 0;
	for (size_t i = 0; i < len; i++) {
		const unsigned char c = norm_table[(unsigned char)s[i]];
		*w = (char)c;
		w += c != 0;
		lost |= c == 0;
	}
	*w = '\0';
	if (lost) *dropped = 1;
	return (size_t)(w - s);
}

/* a reader thread's buffer, reused from file to file (read(), not a fresh allocation per file) */
typedef struct {
	char *buf;
	size_t cap;
} read_buffer;

static char *slurp(const char *file_name, size_t *len_out, read_buffer *rb) {
	const int fd = strcmp(file_name, "-") ? open(file_name, O_RDONLY) : STDIN_FILENO;
	if (fd < 0) return NULL;
	if (!rb->buf) rb->cap = (size_t)4 << 20, rb->buf = xmalloc(rb->cap + 1);
	size_t len = 0;
	int bad = 0;
	for (;;) {
		if (len == rb->cap) {
			rb->cap *= 2;
			rb->buf = realloc(rb->buf, rb->cap + 1);
			if (!rb->buf) err(errno, "Out of memory");
		}
		const ssize_t got = read(fd, rb->buf + len, rb->cap - len);
		if (got < 0) {
			if (errno == EINTR) continue;
			bad = 1;
			break;
		}
		if (got == 0) break;
		len += (size_t)got;
	}
	const int saved = errno;
	if (fd != STDIN_FILENO) close(fd);
	if (bad) {
		errno = saved;
		return NULL;
	}
	rb->buf[len] = '\0';
	*len_out = len;
	return rb->buf;
}

/* read_fasta, src/io.c:196-233: every record of the file becomes one sequence.  The record grammar and the
 * messages are those of the reference's parser (libs/pfasta.c:304-480): the file must start with '>'; the name is
 * the word behind it, the rest of the line a comment; the sequence is every following word of a line that starts
 * with a letter, '-' or '*', across blank lines and blanks inside lines, CR LF or LF; anything else must be the
 * next record's '>'.  A malformed record ends the file with a message (records read before it are kept). */
#define IS_SPACE(c) (((c) >= '\t' && (c) <= '\r') || (c) == ' ')
static void read_fasta(const char *file_name, file_result *res, read_buffer *rb) {
	genome_list *out = &res->seqs;
	size_t len = 0;
	char *text = slurp(file_name, &len, rb);
	if (!text) {
		res->soft = 1;
		fmsg(res, 1, "%s", file_name);
		return;
	}
	if (len == 0) {
		file_soft_warnx(res, "%s: File is empty.", file_name);
		return;
	}
	if (text[0] != '>') {
		file_soft_warnx(res, "%s: File must start with '>'.", file_name);
		return;
	}
	char *p = text, *end = text + len;
	size_t line = 1;
	while (p < end) {
		if (*p != '>') {
			file_soft_warnx(res, "%s: Expected '>' but found '%c' on line %zu.", file_name, *p, line);
			break;
		}
		/* name */
		char *h = ++p;
		while (p < end && !IS_SPACE(*p)) p++;
		if (p == end) {
			file_soft_warnx(res, "%s: Unexpected EOF in name on line %zu.", file_name, line);
			break;
		}
		if (p == h) {
			file_soft_warnx(res, "%s: Empty name on line %zu.", file_name, line);
			break;
		}
		char *name_end = p;
		/* comment: the rest of the line */
		p = memchr(p, '\n', (size_t)(end - p));
		if (!p) {
			file_soft_warnx(res, "%s: Unexpected EOF in comment on line %zu.", file_name, line);
			break;
		}
		/* sequence: first where it ends (the next line that starts with something else), then one copy of its words */
		while (p < end && IS_SPACE(*p)) line += *p++ == '\n';
		char *s0 = p;
		size_t n = 0;
		while (p < end && (isalpha((unsigned char)*p) || *p == '-' || *p == '*')) {
			char *w = p;
			while (p < end && !IS_SPACE(*p)) p++;
			n += (size_t)(p - w);
			while (p < end && IS_SPACE(*p)) line += *p++ == '\n';
		}
		if (n == 0) {
			file_soft_warnx(res, "%s: Empty sequence on line %zu.", file_name, line);
			break;
		}
		genome g;
		g.seq = xmalloc(n + 1);
		size_t k = 0;
		for (char *q = s0; q < p;) { /* the words again: copied, then normalised in place */
			char *w = q;
			while (q < p && !IS_SPACE(*q)) q++;
			memcpy(g.seq + k, w, (size_t)(q - w));
			k += (size_t)(q - w);
			while (q < p && IS_SPACE(*q)) q++;
		}
		g.name = strndup(h, (size_t)(name_end - h));
		if (!g.name) err(errno, "Out of memory");
		g.len = normalize(g.seq, n, &res->non_acgt);
		if (g.len + 4096 < n) { /* (much was dropped: give the memory back) */
			char *fit = realloc(g.seq, g.len + 1);
			if (fit) g.seq = fit;
		}
		push_genome(out, g);
	}
}

/* read_fasta_join + dsa_join, src/io.c:159-194, src/sequence.c:78-125: all records of
 * a file joined by '!', named after the file without directory and extension */
static void read_fasta_join(const char *file_name, file_result *res, read_buffer *rb) {
	file_result one = {0};
	read_fasta(file_name, &one, rb);
	res->msgs = one.msgs, res->mlen = one.mlen, res->mcap = one.mcap, res->soft = one.soft, res->non_acgt = one.non_acgt;
	genome_list single = one.seqs;
	if (single.n == 0) return;
	size_t total = 0;
	for (size_t i = 0; i < single.n; i++) total += single.v[i].len + 1;
	genome g;
	g.seq = xmalloc(total);
	char *w = g.seq;
	for (size_t i = 0; i < single.n; i++) {
		if (i) *w++ = '!';
		memcpy(w, single.v[i].seq, single.v[i].len);
		w += single.v[i].len;
	}
	*w = '\0';
	g.len = total - 1;
	const char *left = strrchr(file_name, '/');
	left = left ? left + 1 : file_name;
	const char *dot = strchrnul(left, '.');
	g.name = strndup(left, (size_t)(dot - left));
	if (!g.name) err(errno, "Out of memory");
	push_genome(&res->seqs, g);
	for (size_t i = 0; i < single.n; i++) {
		free(single.v[i].name);
		free(single.v[i].seq);
	}
	free(single.v);
}

/* the input files, read by up to `threads` threads (each takes the next unread file); results in command-line order */
typedef struct {
	char **files;
	size_t nfiles;
	int join;
	file_result *results;
	size_t next; /* atomic */
} read_job;

static void *read_worker(void *arg) {
	read_job *job = arg;
	read_buffer rb = {0};
	for (;;) {
		const size_t i = __atomic_fetch_add(&job->next, 1, __ATOMIC_RELAXED);
		if (i >= job->nfiles) break;
		if (job->join) read_fasta_join(job->files[i], &job->results[i], &rb);
		else read_fasta(job->files[i], &job->results[i], &rb);
	}
	free(rb.buf);
	return NULL;
}

static void read_all_files(char **files, size_t nfiles, int join, int threads, genome_list *all) {
	norm_table_init();
	read_job job = {files, nfiles, join, calloc(nfiles ? nfiles : 1, sizeof(file_result)), 0};
	if (!job.results) err(errno, "Out of memory");
	size_t nt = threads > 0 ? (size_t)threads : 1;
	if (nt > nfiles) nt = nfiles;
	if (nt > 64) nt = 64; /* (the files come from one file system: more readers do not make it faster) */
	pthread_t *tid = xmalloc(nt * sizeof *tid);
	size_t started = 0;
	for (size_t t = 1; t < nt; t++)
		if (pthread_create(&tid[started], NULL, read_worker, &job) == 0) started++;
	read_worker(&job);
	for (size_t t = 0; t < started; t++) pthread_join(tid[t], NULL);
	free(tid);
	for (size_t i = 0; i < nfiles; i++) {
		file_result *r = &job.results[i];
		if (r->msgs) fputs(r->msgs, stderr);
		free(r->msgs);
		soft_error |= r->soft;
		saw_non_acgt |= r->non_acgt;
		for (size_t k = 0; k < r->seqs.n; k++) push_genome(all, r->seqs.v[k]);
		free(r->seqs.v);
	}
	free(job.results);
}

/* read_into_string_vector, src/io.c:103-144 */
static void read_file_of_filenames(const char *file_name, char ***names, size_t *n, size_t *cap) {
	FILE *f = strcmp(file_name, "-") ? fopen(file_name, "r") : stdin;
	if (!f) {
		soft_error = 1;
		warn("%s", file_name);
		return;
	}
	char *line = NULL;
	size_t bufsz = 0;
	while (getline(&line, &bufsz, f) != -1) {
		char *nl = strchr(line, '\n');
		if (nl) *nl = '\0';
		if (!*line) continue;
		if (*n == *cap) {
			*cap = *cap ? *cap * 2 : 16;
			*names = realloc(*names, *cap * sizeof **names);
			if (!*names) err(errno, "Out of memory");
		}
		(*names)[(*n)++] = strdup(line);
	}
	free(line);
	if (f != stdin) fclose(f);
}

static void usage(int status) {
	static const char str[] =
		"Usage: andi-hip [OPTIONS...] FILES...\n"
		"\tFILES... can be any sequence of FASTA files.\n"
		"\tUse '-' as file name to read from stdin.\n"
		"Options:\n"
		"  -b, --bootstrap=INT  Print additional bootstrap matrices\n"
		"      --file-of-filenames=FILE  Read additional filenames from FILE; one per line\n"
		"  -j, --join           Treat all sequences from one file as a single genome\n"
		"  -l, --low-memory     Use less memory at the cost of speed\n"
		"  -m, --model=MODEL    Pick an evolutionary model of 'Raw', 'JC', 'Kimura', 'LogDet', 'ANI'; "
		"default: JC\n"
		"  -p FLOAT             Significance of an anchor; default: 0.025\n"
		"      --progress=WHEN  Print a progress bar 'always', 'never', or 'auto'; default: auto\n"
		"  -t, --threads=INT    Set the number of host threads; by default, all processors are used\n"
		"      --truncate-names Truncate names to ten characters\n"
		"  -v, --verbose        Prints additional information\n"
		"  -h, --help           Display this help and exit\n"
		"      --version        Output version information and acknowledgments\n";
	fputs(str, status == EXIT_SUCCESS ? stdout : stderr);
	exit(status);
}

static void version(void) {
	printf("andi-hip " PROGRAM_VERSION " (command-line interface of andi 1.15)\n"
		   "Anchor distances on AMD MI355X through libandihip (ABI %d).\n\n"
		   "Acknowledgments:\n"
		   "1) Method: Haubold, B. Kl\xc3\xb6tzl, F. and Pfaffelhuber, P. (2015). Fast and accurate estimation of "
		   "evolutionary distances between closely related genomes, Bioinformatics.\n"
		   "2) Bootstrapping: Kl\xc3\xb6tzl, F. and Haubold, B. (2016). Support Values for Genome Phylogenies, "
		   "Life 6.1.\n",
		   andi_hip_abi_version());
	exit(EXIT_SUCCESS);
}

static size_t progress_n = 0;
static void progress_cb(size_t done, size_t total, void *ud) {
	(void)ud; /* src/dist_hack.h:74-87 */
	fprintf(stderr, "\rComparing %zu sequences: %5.1f%% (%zu/%zu)", progress_n,
			total ? 100.0 * (double)done / (double)total : 100.0, done, total);
}

static void print_matrix(const andi_hip_model *M, const genome *g, size_t n, int model, int vv,
						 int truncate, int warnings) {
	const char **names = xmalloc(n * sizeof *names);
	for (size_t i = 0; i < n; i++) names[i] = g[i].name;
	/* the warnings' buffer grows on demand (a line per pair at worst -- n^2 x 512 bytes up front would be 4.9 GB for
	 * BASELINE's 3085 genomes): a buffer that came back full is doubled and the call repeated */
	size_t cap = 64 + n * (300 + 16 * n), wcap = (size_t)1 << 16;
	char *out = xmalloc(cap), *wbuf = xmalloc(wcap);
	int flags = 0;
	for (;;) {
		const size_t need = andi_hip_format_distances(M, names, n, model, vv, truncate, warnings, out, cap, wbuf, wcap, &flags);
		const int out_short = need >= cap, warn_short = strlen(wbuf) + 1 >= wcap; /* (long names: the call says how much it needs) */
		if (!out_short && !warn_short) break;
		if (out_short) free(out), cap = need + 1, out = xmalloc(cap);
		if (warn_short) free(wbuf), wcap *= 4, wbuf = xmalloc(wcap);
	}
	for (char *line = strtok(wbuf, "\n"); line; line = strtok(NULL, "\n")) soft_warnx("%s", line);
	fputs(out, stdout);
	free(out);
	free(wbuf);
	free(names);
}

int main(int argc, char *argv[]) {
	static const struct option long_options[] = {{"version", no_argument, NULL, 0},
												 {"truncate-names", no_argument, NULL, 0},
												 {"file-of-filenames", required_argument, NULL, 0},
												 {"progress", optional_argument, NULL, 0},
												 {"help", no_argument, NULL, 'h'},
												 {"verbose", no_argument, NULL, 'v'},
												 {"join", no_argument, NULL, 'j'},
												 {"low-memory", no_argument, NULL, 'l'},
												 {"threads", required_argument, NULL, 't'},
												 {"bootstrap", required_argument, NULL, 'b'},
												 {"model", required_argument, NULL, 'm'},
												 {0, 0, 0, 0}};
	andi_hip_opts opts;
	andi_hip_default_opts(&opts);
	long procs = sysconf(_SC_NPROCESSORS_ONLN);
	opts.host_threads = procs > 0 ? (int)procs : 1;
	/* one GPU unless asked: ANDI_HIP_GPUS=k tiles the rows of the matrix over the first k visible GPUs, ANDI_HIP_GPUS=all over
	 * all of them (the gather between distinct devices has not run on hardware yet: opt-in until it has) */
	opts.num_gpus = 1;
	if (getenv("ANDI_HIP_GPUS")) {
		if (!strcmp(getenv("ANDI_HIP_GPUS"), "all")) opts.num_gpus = -1;
		else if (atoi(getenv("ANDI_HIP_GPUS")) > 0) opts.num_gpus = atoi(getenv("ANDI_HIP_GPUS"));
	}
	int verbose = 0, join = 0, truncate = 0;
	unsigned long bootstrap = 0;
	enum { P_AUTO, P_NEVER, P_ALWAYS } progress = P_AUTO;
	char **files = NULL;
	size_t nfiles = 0, files_cap = 0;

	for (;;) {
		int idx = 0;
		int c = getopt_long(argc, argv, "jvht:p:m:b:l", long_options, &idx);
		if (c == -1) break;
		switch (c) {
			case 0: {
				const char *o = long_options[idx].name;
				if (!strcmp(o, "version")) version();
				if (!strcmp(o, "truncate-names")) truncate = 1;
				if (!strcmp(o, "file-of-filenames")) read_file_of_filenames(optarg, &files, &nfiles, &files_cap);
				if (!strcmp(o, "progress")) {
					if (!optarg || !strcasecmp(optarg, "always")) progress = P_ALWAYS;
					else if (!strcasecmp(optarg, "auto")) progress = P_AUTO;
					else if (!strcasecmp(optarg, "never")) progress = P_NEVER;
					else
						warnx("invalid argument to --progress '%s'. Expected one of 'auto', 'always', or "
							  "'never'.", optarg);
				}
				break;
			}
			case 'h': usage(EXIT_SUCCESS); break;
			case 'v': verbose++; break;
			case 'l': opts.low_memory = 1; break;
			case 'j': join = 1; break;
			case 'p': {
				errno = 0;
				char *end;
				double v = strtod(optarg, &end);
				if (errno || end == optarg || *end) {
					soft_warnx("Expected a floating point number for -p argument, but '%s' was given. "
							   "Skipping argument.", optarg);
				} else if (v <= 0.0 || v >= 1.0) {
					soft_warnx("A probability should be a value between 0 and 1, exclusive; Ignoring -p %f "
							   "argument.", v);
				} else {
					opts.p_value = v;
				}
				break;
			}
			case 't': {
				errno = 0;
				char *end;
				unsigned long t = strtoul(optarg, &end, 10);
				if (errno || end == optarg || *end) {
					warnx("Expected a number for -t argument, but '%s' was given. Ignoring -t argument.", optarg);
				} else if (procs > 0 && t > (unsigned long)procs) {
					warnx("The number of threads to be used, is greater than the number of available "
						  "processors; Ignoring -t %lu argument.", t);
				} else if (t > 0) {
					opts.host_threads = (int)t;
				}
				break;
			}
			case 'b': {
				errno = 0;
				char *end;
				unsigned long b = strtoul(optarg, &end, 10);
				if (errno || end == optarg || *end || b == 0) {
					soft_warnx("Expected a positive number for -b argument, but '%s' was given. Ignoring -b "
							   "argument.", optarg);
				} else {
					bootstrap = b - 1; /* -b N prints N matrices in total, src/andi.c:198 */
				}
				break;
			}
			case 'm':
				if (!strcasecmp(optarg, "RAW")) opts.model = ANDI_M_RAW;
				else if (!strcasecmp(optarg, "JC")) opts.model = ANDI_M_JC;
				else if (!strcasecmp(optarg, "KIMURA")) opts.model = ANDI_M_KIMURA;
				else if (!strcasecmp(optarg, "LOGDET")) opts.model = ANDI_M_LOGDET;
				else if (!strcasecmp(optarg, "ANI")) opts.model = ANDI_M_ANI;
				else soft_warnx("Ignoring argument for --model. Expected Raw, JC, Kimura, LogDet or ANI");
				break;
			default: usage(EXIT_FAILURE);
		}
	}
	for (int i = optind; i < argc; i++) {
		if (nfiles == files_cap) {
			files_cap = files_cap ? files_cap * 2 : 16;
			files = realloc(files, files_cap * sizeof *files);
			if (!files) err(errno, "Out of memory");
		}
		files[nfiles++] = strdup(argv[i]);
	}
	if (join && nfiles == 0) errx(1, "In join mode at least one filename needs to be supplied.");
	if (nfiles < (size_t)(join ? 2 : 1)) {
		if (isatty(STDIN_FILENO)) usage(EXIT_FAILURE);
		files = realloc(files, (nfiles + 1) * sizeof *files);
		files[nfiles++] = strdup("-");
	}

	/* ANDI_HIP_CLI_TRACE=1: where the wall time goes -- reading the input, the matrix, printing it (stderr; scripts/full_size.py --cli) */
	const int cli_trace = getenv("ANDI_HIP_CLI_TRACE") != NULL;
	struct timespec ts0, ts1, ts2, ts3;
	clock_gettime(CLOCK_MONOTONIC, &ts0);
	genome_list all = {0};
	read_all_files(files, nfiles, join, opts.host_threads, &all);
	clock_gettime(CLOCK_MONOTONIC, &ts1);
	const size_t n = all.n;
	if (n < 2)
		errx(1, "I am truly sorry, but with less than two sequences (%zu given) there is nothing to compare.", n);
	if (saw_non_acgt)
		warnx("The input sequences contained characters other than acgtACGT. These were automatically "
			  "stripped to ensure correct results.");
	int any_short = 0;
	const size_t limit = (INT_MAX - 1) / 2;
	for (size_t i = 0; i < n; i++) {
		const genome *g = &all.v[i];
		if (truncate && strlen(g->name) > 10)
			warnx("The sequence name '%s' is longer than ten characters. It will be truncated in the output "
				  "to '%.10s'.", g->name, g->name);
		if (g->len > limit) errx(1, "The sequence %s is too long. The technical limit is %zu.", g->name, limit);
		if (g->len == 0) errx(1, "The sequence %s is empty.", g->name);
		if (g->len < 1000) any_short = 1;
	}
	if (any_short)
		soft_warnx("One of the given input sequences is shorter than a thousand nucleotides. This may result "
				   "in inaccurate distances. Try an alignment instead.");

	if (progress == P_AUTO) progress = isatty(STDERR_FILENO) ? P_ALWAYS : P_NEVER;
	if (progress == P_ALWAYS) {
		progress_n = n;
		opts.progress = progress_cb;
		progress_cb(0, n * n - n, NULL);
	}

	/* calculate_distances, src/process.c:230-270 */
	if (SIZE_MAX / sizeof(andi_hip_model) / n < n) errx(1, "Comparison is limited to fewer sequences (%zu given).", n);
	andi_hip_model *M = malloc(n * n * sizeof *M);
	if (!M) err(errno, "Could not allocate enough memory for the comparison matrix. Try using --join or --low-memory.");
	andi_hip_seq *in = xmalloc(n * sizeof *in);
	for (size_t i = 0; i < n; i++) {
		in[i].seq = all.v[i].seq;
		in[i].len = all.v[i].len;
	}
	char msg[512];
	if (andi_hip_dist_matrix(M, in, n, &opts, msg, sizeof msg)) errx(1, "%s", msg);
	if (progress == P_ALWAYS) fprintf(stderr, ", done.\n");
	clock_gettime(CLOCK_MONOTONIC, &ts2);

	print_matrix(M, all.v, n, opts.model, verbose >= 2, truncate, 1);
	if (cli_trace) {
		fflush(stdout);
		clock_gettime(CLOCK_MONOTONIC, &ts3);
#define SECS(a, b) ((double)((b).tv_sec - (a).tv_sec) + 1e-9 * (double)((b).tv_nsec - (a).tv_nsec))
		size_t nt_total = 0;
		for (size_t i = 0; i < n; i++) nt_total += all.v[i].len;
		fprintf(stderr, "andi-hip trace: %zu sequences, %zu nucleotides from %zu files: ingest %.3f s, matrix %.3f s, print %.3f s\n", n, nt_total,
				nfiles, SECS(ts0, ts1), SECS(ts1, ts2), SECS(ts2, ts3));
	}
	if (verbose) { /* print_coverages, src/io.c:329-338 */
		printf("\nCoverage:\n");
		for (size_t i = 0; i < n; i++) {
			for (size_t j = 0; j < n; j++) printf("%1.4e ", andi_hip_model_coverage(&M[i * n + j]));
			printf("\n");
		}
	}
	if (bootstrap) { /* calculate_bootstrap, src/process.c:289-321 */
		andi_hip_ctx *ctx = NULL;
		andi_hip_model *B = malloc(bootstrap * n * n * sizeof *B);
		if (!B || andi_hip_ctx_create(&ctx, opts.device, msg, sizeof msg) ||
			andi_hip_bootstrap(ctx, M, n, (uint64_t)time(NULL), bootstrap, B)) {
			soft_warnx("Bootstrapping failed.");
		} else {
			for (unsigned long b = 0; b < bootstrap; b++)
				print_matrix(B + b * n * n, all.v, n, opts.model, verbose >= 2, truncate, 0);
		}
		if (ctx) andi_hip_ctx_destroy(ctx);
		free(B);
	}
	free(M);
	free(in);
	return soft_error ? EXIT_FAILURE : EXIT_SUCCESS;
}
