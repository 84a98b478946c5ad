/* A client written against Kiwi's C API only (include/kiwi_capi.h <- reference include/kiwi/capi.h), in the shape of the
 * reference's own C test (reference test/test_c.cpp:48-71): lines of a UTF-8 file go through kiwi_analyze_m with a
 * reader / receiver pair; every token is printed as  line <tab> form <tab> tag <tab> position <tab> length <tab> score.
 * Built and run by tests/test_gpu_capi.py::test_c_client_program.  usage: client <model> <text file> */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef KIWI_CLIENT_REFERENCE_HEADER
#include <kiwi/capi.h>      /* the reference's own header: -I/root/reference/include */
#else
#include "kiwi_capi.h"
#endif

typedef struct { char** lines; int n; } corpus_t;

static int reader(int id, char* buffer, void* user)
{
	corpus_t* c = (corpus_t*)user;
	if (id >= c->n) return 0;
	if (!buffer) return (int)strlen(c->lines[id]);
	memcpy(buffer, c->lines[id], strlen(c->lines[id]));
	return 0;
}

static int receiver(int id, kiwi_res_h r, void* user)
{
	(void)user;
	if (kiwi_res_size(r) > 0)
	{
		int n = kiwi_res_word_num(r, 0), j;
		for (j = 0; j < n; ++j)
			printf("%d\t%s\t%s\t%d\t%d\t%.9g\n", id, kiwi_res_form(r, 0, j), kiwi_res_tag(r, 0, j), kiwi_res_position(r, 0, j), kiwi_res_length(r, 0, j), (double)kiwi_res_score(r, 0, j));
		printf("%d\t#\t%.9g\n", id, (double)kiwi_res_prob(r, 0));
	}
	kiwi_res_close(r);
	return 0;
}

int main(int argc, char** argv)
{
	corpus_t c = { 0, 0 };
	char* line = 0; size_t cap = 0; ssize_t len;
	kiwi_analyze_option_t opt;
	kiwi_h k;
	FILE* f;
	int done;
	if (argc < 3) { fprintf(stderr, "usage: client <model> <text file>\n"); return 2; }
	f = fopen(argv[2], "rb");
	if (!f) { perror(argv[2]); return 2; }
	while ((len = getline(&line, &cap, f)) >= 0)
	{
		while (len && (line[len - 1] == '\n' || line[len - 1] == '\r')) line[--len] = 0;
		if (!len) continue;
		c.lines = (char**)realloc(c.lines, sizeof(char*) * (size_t)(c.n + 1));
		c.lines[c.n++] = strdup(line);
	}
	fclose(f);
	k = kiwi_init(argv[1], 0, KIWI_BUILD_DEFAULT, 0);
	if (!k) { fprintf(stderr, "kiwi_init: %s\n", kiwi_error()); return 1; }
	memset(&opt, 0, sizeof(opt));
	opt.match_options = (1 << 0) | (1 << 1) | (1 << 2) | (1 << 3) | (1 << 4) | (1 << 5) | (1 << 23) | (1 << 16);   /* KIWI_MATCH_ALL_WITH_NORMALIZING */
	done = kiwi_analyze_m(k, reader, receiver, &c, 1, opt);
	if (done < 0) { fprintf(stderr, "kiwi_analyze_m: %s\n", kiwi_error()); return 1; }
	fprintf(stderr, "analysed %d lines\n", done);
	return kiwi_close(k);
}
