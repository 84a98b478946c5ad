/* Times the symbol a user of the reference calls for a batch: kiwi_analyze_m (reader -> receiver, results delivered in input order;
 * include/kiwi_capi.h <- reference include/kiwi/capi.h).  The lines of a UTF-8 file are analysed `passes` + 1 times (the first pass is an untimed
 * warm-up: device blocks, pinned buffers, host pool); the receiver does what a client must do with a result -- ask for its size and the number
 * of tokens of the best analysis, close it.  Prints one JSON line.  usage: capi_bench <model> <text file> [passes] [top_n]
 * Used by bench.py (`capi` block of its output line); built by tools/Makefile against THIS repo's header, so it travels to the GPU box. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include "kiwi_capi.h"

typedef struct { char** lines; int* lens; int n; long tokens; } corpus_t;

static int reader(int id, char* buffer, void* user)
{
	corpus_t* c = (corpus_t*)user;
	if (id >= c->n) return 0;
	if (!buffer) return c->lens[id];
	memcpy(buffer, c->lines[id], (size_t)c->lens[id]);
	return 0;
}

static int receiver(int id, kiwi_res_h r, void* user)
{
	corpus_t* c = (corpus_t*)user;
	(void)id;
	if (kiwi_res_size(r) > 0) c->tokens += kiwi_res_word_num(r, 0);
	kiwi_res_close(r);
	return 0;
}

static double now(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; }

int main(int argc, char** argv)
{
	corpus_t c = { 0, 0, 0, 0 };
	char* line = 0; size_t cap = 0; ssize_t len;
	kiwi_analyze_option_t opt;
	kiwi_h k;
	FILE* f;
	int passes = argc > 3 ? atoi(argv[3]) : 5, topN = argc > 4 ? atoi(argv[4]) : 1, p, done = 0;
	double t0, t1;
	if (argc < 3) { fprintf(stderr, "usage: capi_bench <model> <text file> [passes] [top_n]\n"); return 2; }
	f = fopen(argv[2], "rb");
	if (!f) { perror(argv[2]); return 2; }
	while ((len = getline(&line, &cap, f)) >= 0)
	{
		while (len && (line[len - 1] == '\n' || line[len - 1] == '\r')) line[--len] = 0;
		if (!len) continue;
		c.lines = (char**)realloc(c.lines, sizeof(char*) * (size_t)(c.n + 1));
		c.lens = (int*)realloc(c.lens, sizeof(int) * (size_t)(c.n + 1));
		c.lines[c.n] = strdup(line); c.lens[c.n] = (int)len; ++c.n;
	}
	fclose(f);
	k = kiwi_init(argv[1], -1, KIWI_BUILD_DEFAULT, 0);      /* num_threads -1: "as many as the machine has" (capi.h:594), what a throughput client asks for; 0 would keep every host stage on the calling thread */
	if (!k) { fprintf(stderr, "kiwi_init: %s\n", kiwi_error()); return 1; }
	memset(&opt, 0, sizeof(opt));
	opt.match_options = (1 << 0) | (1 << 1) | (1 << 2) | (1 << 3) | (1 << 4) | (1 << 5) | (1 << 23) | (1 << 16);   /* KIWI_MATCH_ALL_WITH_NORMALIZING */
	if (kiwi_analyze_m(k, reader, receiver, &c, topN, opt) < 0) { fprintf(stderr, "kiwi_analyze_m: %s\n", kiwi_error()); return 1; }
	c.tokens = 0;
	t0 = now();
	for (p = 0; p < passes; ++p)
	{
		done = kiwi_analyze_m(k, reader, receiver, &c, topN, opt);
		if (done < 0) { fprintf(stderr, "kiwi_analyze_m: %s\n", kiwi_error()); return 1; }
	}
	t1 = now();
	printf("{\"symbol\": \"kiwi_analyze_m\", \"value\": %.1f, \"unit\": \"sentences/s\", \"lines\": %d, \"passes\": %d, \"ms_per_pass\": %.3f, \"tokens_per_pass\": %ld, \"top_n\": %d}\n",
		(double)done * passes / (t1 - t0), done, passes, 1000.0 * (t1 - t0) / passes, c.tokens / (passes ? passes : 1), topN);
	return kiwi_close(k);
}
