/*
 * ntedit_oracle_main.c -- command-line driver for the CPU oracle.
 * TEST INFRASTRUCTURE ONLY.  Mirrors the reference's flag surface
 * (ntedit.cpp:135-169, 2276-2364) at -t 1 so that golden outputs can be
 * produced with the same command lines a reference user would type.
 */
#include "ntedit_oracle.h"

#include <getopt.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

static const char shortopts[] = "t:f:s:k:z:b:r:v:d:i:X:Y:x:y:m:c:j:s:e:a:l:p:q:";

static const char*
basename_of(const char* p)
{
	const char* s = strrchr(p, '/');
	return s ? s + 1 : p;
}

int
main(int argc, char** argv)
{
	ora_params p;
	ora_params_default(&p);
	const char *draft = NULL, *bfpath = NULL, *bfrep = NULL, *prefix = NULL, *annot_path = NULL;
	int report = 0;
	static const struct option longopts[] = { { "report", no_argument, NULL, 1000 },
		                                      { NULL, 0, NULL, 0 } };
	for (int c; (c = getopt_long(argc, argv, shortopts, longopts, NULL)) != -1;) {
		switch (c) {
		case 'f':
			draft = optarg;
			break;
		case 'r':
			bfpath = optarg;
			break;
		case 'e':
			bfrep = optarg;
			break;
		case 'b':
			prefix = optarg;
			break;
		case 'z':
			p.min_contig_len = (unsigned)strtoul(optarg, NULL, 10);
			break;
		case 'i':
			p.max_insertions = (unsigned)strtoul(optarg, NULL, 10);
			break;
		case 'd':
			p.max_deletions = (unsigned)strtoul(optarg, NULL, 10);
			break;
		case 'x':
			p.missing_threshold = strtof(optarg, NULL);
			break;
		case 'y':
			p.edit_threshold = strtof(optarg, NULL);
			break;
		case 'X':
			p.missing_ratio = strtof(optarg, NULL);
			p.use_ratio = 1;
			break;
		case 'Y':
			p.edit_ratio = strtof(optarg, NULL);
			p.use_ratio = 1;
			break;
		case 'j':
			p.jump = (unsigned)strtoul(optarg, NULL, 10);
			break;
		case 'm':
			p.mode = atoi(optarg);
			break;
		case 's':
			p.snv = atoi(optarg);
			break;
		case 'a':
			p.mask = atoi(optarg);
			break;
		case 'p':
			p.min_threshold = (unsigned)strtoul(optarg, NULL, 10);
			break;
		case 'q':
			p.max_threshold = (unsigned)strtoul(optarg, NULL, 10);
			break;
		case 1000:
			report = 1;
			break;
		case 'l':
			annot_path = optarg;
			break;
		case 't':
		case 'k':
		case 'c':
		case 'v':
			break; /* accepted, no effect on this path */
		default:
			return 2;
		}
	}
	if (!draft || !bfpath) {
		fprintf(stderr, "usage: ntedit_oracle -f draft.fa -r filter.bf [-e rep.bf] [-b prefix] ...\n");
		return 2;
	}
	ora_bf bloom, bloomrep;
	memset(&bloomrep, 0, sizeof bloomrep);
	if (ora_bf_load(&bloom, bfpath)) {
		fprintf(stderr, "ntedit_oracle: error: cannot load Bloom filter %s\n", bfpath);
		return 1;
	}
	p.secbf = bfrep != NULL;
	ora_params_finalize(&p, &bloom);
	if (bfrep) {
		if (ora_bf_load(&bloomrep, bfrep)) {
			fprintf(stderr, "ntedit_oracle: error: cannot load secondary Bloom filter %s\n", bfrep);
			return 1;
		}
		if (bloomrep.k != p.k) {
			fprintf(stderr, "ntedit_oracle: error: secondary Bloom filter k size differs\n");
			return 1;
		}
	}
	char defprefix[4096];
	if (!prefix) {
		/* ntedit.cpp:2496-2502 */
		snprintf(
		    defprefix,
		    sizeof defprefix,
		    "%s_k%u_z%u_r%s_i%u_d%u_m%d",
		    basename_of(draft),
		    p.k,
		    p.min_contig_len,
		    basename_of(bfpath),
		    p.max_insertions,
		    p.max_deletions,
		    p.mode);
		prefix = defprefix;
	}
	uint64_t bases = 0;
	struct timespec t0, t1;
	clock_gettime(CLOCK_MONOTONIC, &t0);
	ora_annot* annot = annot_path ? ora_annot_load(annot_path) : NULL;
	int rc = ora_polish_file_vcf(draft, &p, &bloom, bfrep ? &bloomrep : NULL, prefix, &bases, annot);
	ora_annot_free(annot);
	clock_gettime(CLOCK_MONOTONIC, &t1);
	if (report) {
		double s = (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
		printf(
		    "{\"bases\": %llu, \"seconds\": %.6f, \"rolls\": %llu, \"contains\": %llu, "
		    "\"bitreads\": %llu}\n",
		    (unsigned long long)bases,
		    s,
		    (unsigned long long)ora_ctr.rolls,
		    (unsigned long long)ora_ctr.contains,
		    (unsigned long long)ora_ctr.bitreads);
	}
	ora_bf_free(&bloom);
	if (bfrep) {
		ora_bf_free(&bloomrep);
	}
	return rc ? 1 : 0;
}
