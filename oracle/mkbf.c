/*
 * mkbf.c -- build a btllib-style k-mer Bloom filter file from FASTA input(s).
 * TEST INFRASTRUCTURE ONLY (fixtures for the oracle and the parity tests).
 * Behavioural model: src/ntedit_make_genome_bf.cpp:143-162 (insert every
 * k-mer of every sequence), with the array size given explicitly.
 *   usage: mkbf -k K -g HASHES -s BYTES [-C] -o out.bf in.fa[.gz] ...
 */
#include "ntedit_oracle.h"

#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include <zlib.h>

int
main(int argc, char** argv)
{
	unsigned k = 25, h = 3;
	unsigned long long bytes = 1 << 23;
	int counting = 0;
	const char* out = NULL;
	for (int c; (c = getopt(argc, argv, "k:g:s:o:C")) != -1;) {
		switch (c) {
		case 'k':
			k = (unsigned)atoi(optarg);
			break;
		case 'g':
			h = (unsigned)atoi(optarg);
			break;
		case 's':
			bytes = strtoull(optarg, NULL, 10);
			break;
		case 'o':
			out = optarg;
			break;
		case 'C':
			counting = 1;
			break;
		default:
			return 2;
		}
	}
	if (!out || optind >= argc) {
		fprintf(stderr, "usage: mkbf -k K -g H -s BYTES [-C] -o out.bf in.fa ...\n");
		return 2;
	}
	ora_bf bf;
	if (ora_bf_init(&bf, bytes, h, k, counting)) {
		return 1;
	}
	for (int a = optind; a < argc; a++) {
		gzFile f = gzopen(argv[a], "r");
		if (!f) {
			fprintf(stderr, "mkbf: cannot open %s\n", argv[a]);
			return 1;
		}
		size_t cap = 1 << 20, n = 0;
		char* seq = (char*)malloc(cap);
		char line[1 << 16];
		while (1) {
			char* got = gzgets(f, line, sizeof line);
			if (!got || line[0] == '>') {
				if (n) {
					ora_bf_insert_seq(&bf, seq, n);
				}
				n = 0;
				if (!got) {
					break;
				}
				/* swallow the rest of a long header line */
				while (got && !strchr(line, '\n')) {
					got = gzgets(f, line, sizeof line);
				}
				continue;
			}
			size_t l = strlen(line);
			while (l && (line[l - 1] == '\n' || line[l - 1] == '\r')) {
				l--;
			}
			if (n + l > cap) {
				cap = (n + l) * 2;
				seq = (char*)realloc(seq, cap);
			}
			memcpy(seq + n, line, l);
			n += l;
		}
		free(seq);
		gzclose(f);
	}
	int rc = ora_bf_save(&bf, out);
	ora_bf_free(&bf);
	return rc ? 1 : 0;
}
