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
#include <getopt.h>
#include <limits.h>
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

/* normalize, src/sequence.c:260-282: keep ACGT and '!', upper-case acgt, drop the rest */
static size_t normalize(char *s) {
	char *w = s;
	for (const char *r = s; *r; r++) {
		switch (*r) {
			case 'A': case 'C': case 'G': case 'T': case '!': *w++ = *r; break;
			case 'a': *w++ = 'A'; break;
			case 'c': *w++ = 'C'; break;
			case 'g': *w++ = 'G'; break;
			case 't': *w++ = 'T'; break;
			default: saw_non_acgt = 1; break;
		}
	}
	*w = '\0';
	return (size_t)(w - s);
}

static char *slurp(const char *file_name, size_t *len_out) {
	FILE *f = strcmp(file_name, "-") ? fopen(file_name, "r") : stdin;
	if (!f) return NULL;
	size_t cap = 1 << 16, len = 0;
	char *buf = xmalloc(cap + 1);
	for (;;) {
		size_t got = fread(buf + len, 1, cap - len, f);
		len += got;
		if (got == 0) break;
		if (len == cap) {
			cap *= 2;
			buf = realloc(buf, cap + 1);
			if (!buf) err(errno, "Out of memory");
		}
	}
	int bad = ferror(f);
	if (f != stdin) fclose(f);
	if (bad) {
		free(buf);
		return NULL;
	}
	buf[len] = '\0';
	*len_out = len;
	return buf;
}

/* read_fasta, src/io.c:196-233: every record of the file becomes one sequence.  The record grammar and the
 * messages are those of the reference's parser (libs/pfasta.c:304-480): the file must start with '>'; the name is
 * the word behind it, the rest of the line a comment; the sequence is every following word of a line that starts
 * with a letter, '-' or '*', across blank lines and blanks inside lines, CR LF or LF; anything else must be the
 * next record's '>'.  A malformed record ends the file with a message (records read before it are kept). */
#define IS_SPACE(c) (((c) >= '\t' && (c) <= '\r') || (c) == ' ')
static void read_fasta(const char *file_name, genome_list *out) {
	size_t len = 0;
	char *text = slurp(file_name, &len);
	if (!text) {
		soft_error = 1;
		warn("%s", file_name);
		return;
	}
	if (len == 0) {
		soft_warnx("%s: File is empty.", file_name);
		free(text);
		return;
	}
	if (text[0] != '>') {
		soft_warnx("%s: File must start with '>'.", file_name);
		free(text);
		return;
	}
	char *p = text, *end = text + len;
	size_t line = 1;
	while (p < end) {
		if (*p != '>') {
			soft_warnx("%s: Expected '>' but found '%c' on line %zu.", file_name, *p, line);
			break;
		}
		/* name */
		char *h = ++p;
		while (p < end && !IS_SPACE(*p)) p++;
		if (p == end) {
			soft_warnx("%s: Unexpected EOF in name on line %zu.", file_name, line);
			break;
		}
		if (p == h) {
			soft_warnx("%s: Empty name on line %zu.", file_name, line);
			break;
		}
		char *name_end = p;
		/* comment: the rest of the line */
		while (p < end && *p != '\n') p++;
		if (p == end) {
			soft_warnx("%s: Unexpected EOF in comment on line %zu.", file_name, line);
			break;
		}
		/* sequence */
		while (p < end && IS_SPACE(*p)) line += *p++ == '\n';
		genome g;
		g.seq = xmalloc((size_t)(end - p) + 1);
		size_t n = 0;
		while (p < end && (isalpha((unsigned char)*p) || *p == '-' || *p == '*')) {
			while (p < end && !IS_SPACE(*p)) g.seq[n++] = *p++;
			while (p < end && IS_SPACE(*p)) line += *p++ == '\n';
		}
		if (n == 0) {
			soft_warnx("%s: Empty sequence on line %zu.", file_name, line);
			free(g.seq);
			break;
		}
		g.seq[n] = '\0';
		g.name = strndup(h, (size_t)(name_end - h));
		if (!g.name) err(errno, "Out of memory");
		g.len = normalize(g.seq);
		char *fit = realloc(g.seq, g.len + 1);
		if (fit) g.seq = fit;
		push_genome(out, g);
	}
	free(text);
}

/* read_fasta_join + dsa_join, src/io.c:159-194, src/sequence.c:78-125: all records of
 * a file joined by '!', named after the file without directory and extension */
static void read_fasta_join(const char *file_name, genome_list *out) {
	genome_list single = {0};
	read_fasta(file_name, &single);
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
	push_genome(out, g);
	for (size_t i = 0; i < single.n; i++) {
		free(single.v[i].name);
		free(single.v[i].seq);
	}
	free(single.v);
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
	size_t cap = 64 + n * (300 + 16 * n), wcap = 4096 + n * n * 512;
	char *out = xmalloc(cap), *wbuf = xmalloc(wcap);
	int flags = 0;
	size_t need = andi_hip_format_distances(M, names, n, model, vv, truncate, warnings, out, cap, wbuf, wcap, &flags);
	if (need >= cap) { /* long names: the call says how much it needs */
		free(out);
		cap = need + 1;
		out = xmalloc(cap);
		andi_hip_format_distances(M, names, n, model, vv, truncate, warnings, out, cap, wbuf, wcap, &flags);
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

	genome_list all = {0};
	for (size_t i = 0; i < nfiles; i++) {
		if (join) read_fasta_join(files[i], &all);
		else read_fasta(files[i], &all);
	}
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

	print_matrix(M, all.v, n, opts.model, verbose >= 2, truncate, 1);
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
