/*
 * ntedit_oracle.c -- CPU restatement of ntEdit v2.1.1's hot path.
 * TEST INFRASTRUCTURE ONLY (see ntedit_oracle.h for the parity statement).
 *
 * Written from the behaviour of /root/reference/ntedit.cpp; each function
 * cites the reference lines it restates.  Data structures are plain C arrays.
 *
 * Defined behaviour where the reference has undefined behaviour:
 *   (U1) reading newSeq[i] with i >= newSeq.size() (e.g. ntedit.cpp:1481,1486,
 *        914) yields an "unset" node (node_type -1, c 0).
 *   (U2) best_sub_base / altbaseN are read uninitialised in -m 2 corner cases
 *        (ntedit.cpp:1881-1885,2019); here they start as 0.
 *   (U3) base_*_hash on non-ACGT characters: btllib uses tetramer tables whose
 *        behaviour on IUPAC input is unknown here; we define the seed hash as
 *        the XOR of per-character SEED_TAB terms (identical for ACGT input and
 *        consistent with the rolling update).
 *   (U4) writeEditsToFile reads contigSeq.at(s_pos - insertion length) for an insertion
 *        row (ntedit.cpp:957); when insertions accumulated to more bases than lie in
 *        front of them that index wraps and .at() throws (the reference terminates).
 *        Here the row is written with 'N' as its draft base.
 */
#include "ntedit_oracle.h"

#include <ctype.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <zlib.h>

__thread ora_counters ora_ctr = { 0, 0, 0 }; /* per thread (ora_polish_batch_flat_mt) */

/* ------------------------------------------------------------------ ntHash2 */
/* btllib nthash_consts: SEED_A/C/G/T, SEED_N = 0, CP_OFF = 7, MULTISHIFT = 27,
 * MULTISEED = 0x90b45d39fb6da1fa [published ntHash2 constants] */
#define SEED_A 0x3c8bfbb395c60474ULL
#define SEED_C 0x3193c18562a02b4cULL
#define SEED_G 0x20323ed082572324ULL
#define SEED_T 0x295549f54be24456ULL
#define MULTISEED 0x90b45d39fb6da1faULL
#define MULTISHIFT 27
#define CP_OFF 0x07

/* SEED_TAB[256]: letters map to their seed, slots 0..7 hold the complement
 * seeds addressed by (c & CP_OFF); everything else is SEED_N = 0. */
uint64_t
ora_seed(unsigned char c)
{
	switch (c) {
	case 'A':
	case 'a':
		return SEED_A;
	case 'C':
	case 'c':
		return SEED_C;
	case 'G':
	case 'g':
		return SEED_G;
	case 'T':
	case 't':
	case 'U':
	case 'u':
		return SEED_T;
	case 1: /* 'A' & 7 -> complement T */
		return SEED_T;
	case 3: /* 'C' & 7 -> G */
		return SEED_G;
	case 4: /* 'T' & 7 -> A */
	case 5: /* 'U' & 7 -> A */
		return SEED_A;
	case 7: /* 'G' & 7 -> C */
		return SEED_C;
	default:
		return 0;
	}
}

/* split rotate left by one: bits 0..32 and 33..63 rotate independently */
uint64_t
ora_srol(uint64_t x)
{
	uint64_t m = ((x & 0x8000000000000000ULL) >> 30) | ((x & 0x100000000ULL) >> 32);
	return ((x << 1) & 0xFFFFFFFDFFFFFFFFULL) | m;
}

uint64_t
ora_sror(uint64_t x)
{
	uint64_t m = ((x & 0x200000000ULL) << 30) | ((x & 1ULL) << 32);
	return ((x >> 1) & 0xFFFFFFFEFFFFFFFFULL) | m;
}

/* split rotate by d: low 33 bits rotate by d%33, high 31 bits by d%31 */
uint64_t
ora_srol_n(uint64_t x, unsigned d)
{
	const uint64_t lo_mask = 0x1FFFFFFFFULL;
	uint64_t lo = x & lo_mask;
	uint64_t hi = x >> 33;
	unsigned dl = d % 33, dh = d % 31;
	if (dl) {
		lo = ((lo << dl) | (lo >> (33 - dl))) & lo_mask;
	}
	if (dh) {
		hi = ((hi << dh) | (hi >> (31 - dh))) & 0x7FFFFFFFULL;
	}
	return (hi << 33) | lo;
}

uint64_t
ora_base_forward_hash(const char* s, unsigned k)
{
	uint64_t h = 0;
	for (unsigned i = 0; i < k; i++) {
		h = ora_srol(h) ^ ora_seed((unsigned char)s[i]);
	}
	return h;
}

uint64_t
ora_base_reverse_hash(const char* s, unsigned k)
{
	uint64_t h = 0;
	for (unsigned i = 0; i < k; i++) {
		h ^= ora_srol_n(ora_seed((unsigned char)s[i] & CP_OFF), i);
	}
	return h;
}

uint64_t
ora_next_forward_hash(uint64_t fh, unsigned k, unsigned char out, unsigned char in)
{
	return ora_srol(fh) ^ ora_seed(in) ^ ora_srol_n(ora_seed(out), k);
}

uint64_t
ora_next_reverse_hash(uint64_t rh, unsigned k, unsigned char out, unsigned char in)
{
	uint64_t h = rh ^ ora_srol_n(ora_seed(in & CP_OFF), k) ^ ora_seed(out & CP_OFF);
	return ora_sror(h);
}

void
ora_extend_hashes(uint64_t base, unsigned k, unsigned h, uint64_t* hv)
{
	hv[0] = base;
	for (unsigned i = 1; i < h; i++) {
		uint64_t t = base * ((uint64_t)i ^ ((uint64_t)k * MULTISEED));
		t ^= t >> MULTISHIFT;
		hv[i] = t;
	}
}

/* the three adapters of ntedit.cpp:403-452 */
static void
NTMC64_seed(const char* kmer, unsigned k, unsigned m, uint64_t* fh, uint64_t* rh, uint64_t* hv)
{
	*fh = ora_base_forward_hash(kmer, k);
	*rh = ora_base_reverse_hash(kmer, k);
	ora_extend_hashes(*fh + *rh, k, m, hv);
}

static void
NTMC64_roll(
    unsigned char out,
    unsigned char in,
    unsigned k,
    unsigned m,
    uint64_t* fh,
    uint64_t* rh,
    uint64_t* hv)
{
	*fh = ora_next_forward_hash(*fh, k, out, in);
	*rh = ora_next_reverse_hash(*rh, k, out, in);
	ora_extend_hashes(*fh + *rh, k, m, hv);
	ora_ctr.rolls++;
}

static void
NTMC64_changelast(
    unsigned char out,
    unsigned char in,
    unsigned k,
    unsigned m,
    uint64_t* fh,
    uint64_t* rh,
    uint64_t* hv)
{
	*fh ^= ora_seed(out);
	*fh ^= ora_seed(in);
	*rh ^= ora_srol_n(ora_seed(out & CP_OFF), k - 1);
	*rh ^= ora_srol_n(ora_seed(in & CP_OFF), k - 1);
	ora_extend_hashes(*fh + *rh, k, m, hv);
}

/* ------------------------------------------------------------- Bloom filter */
int
ora_bf_init(ora_bf* bf, uint64_t bytes, unsigned hash_num, unsigned k, int counting)
{
	memset(bf, 0, sizeof(*bf));
	/* btllib rounds the array up to a multiple of 8 bytes */
	bytes = (bytes + 7) / 8 * 8;
	bf->data = (uint8_t*)calloc(bytes, 1);
	if (!bf->data) {
		return -1;
	}
	bf->bytes = bytes;
	bf->bits = bytes * 8;
	bf->hash_num = hash_num;
	bf->k = k;
	bf->counting = counting;
	bf->owns = 1;
	return 0;
}

void
ora_bf_free(ora_bf* bf)
{
	if (bf->owns) {
		free(bf->data);
	}
	memset(bf, 0, sizeof(*bf));
}

unsigned
ora_bf_contains(const ora_bf* bf, const uint64_t* hv)
{
	ora_ctr.contains++;
	if (bf->counting) {
		unsigned mn = 255;
		for (unsigned i = 0; i < bf->hash_num; i++) {
			unsigned c = bf->data[hv[i] % bf->bytes];
			ora_ctr.bitreads++;
			if (c < mn) {
				mn = c;
			}
		}
		return mn;
	}
	for (unsigned i = 0; i < bf->hash_num; i++) {
		uint64_t n = hv[i] % bf->bits;
		ora_ctr.bitreads++;
		if (!((bf->data[n >> 3] >> (n & 7)) & 1)) {
			return 0;
		}
	}
	return 1;
}

void
ora_bf_insert(ora_bf* bf, const uint64_t* hv)
{
	if (bf->counting) {
		unsigned mn = 255;
		for (unsigned i = 0; i < bf->hash_num; i++) {
			unsigned c = bf->data[hv[i] % bf->bytes];
			if (c < mn) {
				mn = c;
			}
		}
		if (mn == 255) {
			return;
		}
		for (unsigned i = 0; i < bf->hash_num; i++) {
			uint8_t* p = &bf->data[hv[i] % bf->bytes];
			if (*p == mn) {
				*p = (uint8_t)(mn + 1);
			}
		}
		return;
	}
	for (unsigned i = 0; i < bf->hash_num; i++) {
		uint64_t n = hv[i] % bf->bits;
		bf->data[n >> 3] |= (uint8_t)(1u << (n & 7));
	}
}

static int
is_acgt(unsigned char c)
{
	c = (unsigned char)toupper(c);
	return c == 'A' || c == 'C' || c == 'G' || c == 'T';
}

void
ora_bf_insert_seq(ora_bf* bf, const char* seq, size_t len)
{
	unsigned k = bf->k;
	uint64_t hv[64];
	uint64_t fh = 0, rh = 0;
	size_t run = 0; /* number of consecutive ACGT chars ending at i */
	for (size_t i = 0; i < len; i++) {
		if (!is_acgt((unsigned char)seq[i])) {
			run = 0;
			continue;
		}
		run++;
		if (run == k) {
			fh = ora_base_forward_hash(seq + i + 1 - k, k);
			rh = ora_base_reverse_hash(seq + i + 1 - k, k);
		} else if (run > k) {
			fh = ora_next_forward_hash(fh, k, (unsigned char)seq[i - k], (unsigned char)seq[i]);
			rh = ora_next_reverse_hash(rh, k, (unsigned char)seq[i - k], (unsigned char)seq[i]);
		} else {
			continue;
		}
		ora_extend_hashes(fh + rh, k, bf->hash_num, hv);
		ora_bf_insert(bf, hv);
	}
}

/* btllib-style file: "[BTLKmerBloomFilter_vN]\nkey = value\n...[HeaderEnd]\n" + raw array.
 * (format restated from memory: PARITY UNPINNED; the loader accepts keys in
 * any order and any _vN suffix) */
int
ora_bf_save(const ora_bf* bf, const char* path)
{
	FILE* f = fopen(path, "wb");
	if (!f) {
		return -1;
	}
	fprintf(
	    f,
	    "[%s]\nbytes = %llu\nhash_fn = \"ntHash_v2\"\nhash_num = %u\nk = %u\n[HeaderEnd]\n",
	    bf->counting ? "BTLKmerCountingBloomFilter_v5" : "BTLKmerBloomFilter_v6",
	    (unsigned long long)bf->bytes,
	    bf->hash_num,
	    bf->k);
	size_t w = fwrite(bf->data, 1, bf->bytes, f);
	fclose(f);
	return w == bf->bytes ? 0 : -1;
}

int
ora_bf_load(ora_bf* bf, const char* path)
{
	memset(bf, 0, sizeof(*bf));
	FILE* f = fopen(path, "rb");
	if (!f) {
		return -1;
	}
	char line[512];
	int first = 1, ok = 0;
	unsigned long long bytes = 0;
	unsigned hash_num = 0, k = 0;
	int counting = 0;
	while (fgets(line, sizeof line, f)) {
		if (first) {
			first = 0;
			if (strncmp(line, "[BTL", 4) != 0) {
				break;
			}
			counting = strstr(line, "Counting") != NULL;
			continue;
		}
		if (strncmp(line, "[HeaderEnd]", 11) == 0) {
			ok = 1;
			break;
		}
		char key[64];
		char val[256];
		if (sscanf(line, " %63[^ =] = %255[^\n]", key, val) == 2) {
			if (!strcmp(key, "bytes")) {
				bytes = strtoull(val, NULL, 10);
			} else if (!strcmp(key, "hash_num")) {
				hash_num = (unsigned)strtoul(val, NULL, 10);
			} else if (!strcmp(key, "k")) {
				k = (unsigned)strtoul(val, NULL, 10);
			}
		}
	}
	if (!ok || !bytes || !hash_num) {
		fclose(f);
		return -2;
	}
	if (ora_bf_init(bf, bytes, hash_num, k, counting)) {
		fclose(f);
		return -3;
	}
	size_t r = fread(bf->data, 1, bytes, f);
	fclose(f);
	if (r != bytes) {
		ora_bf_free(bf);
		return -4;
	}
	/* btllib's file constructor takes the header's size as it is (array_size = bytes, array_bits =
	 * bytes * 8); only its BUILD constructor rounds up to 8 bytes (ora_bf_init) */
	bf->bytes = bytes;
	bf->bits = bytes * 8;
	return 0;
}

/* --------------------------------------------------------------- parameters */
void
ora_params_default(ora_params* p)
{
	/* ntedit.cpp:99-133 */
	memset(p, 0, sizeof(*p));
	p->jump = 3;
	p->min_contig_len = 100;
	p->max_insertions = 5;
	p->max_deletions = 5;
	p->edit_threshold = 9.0f;
	p->missing_threshold = 5.0f;
	p->edit_ratio = 0.5f;
	p->missing_ratio = 0.5f;
	p->use_ratio = 0;
	p->mode = 0;
	p->min_threshold = 1;
	p->max_threshold = 255;
}

int
ora_params_finalize(ora_params* p, const ora_bf* bloom)
{
	/* ntedit.cpp:2411-2413 */
	if (p->snv) {
		p->max_insertions = 0;
		p->max_deletions = 0;
	}
	/* ntedit.cpp:2439,2448,2450-2451 (-c is overwritten) */
	p->h = bloom->hash_num;
	p->k = bloom->k;
	p->insertion_cap = (unsigned)((float)p->k * 1.5f);
	/* ntedit.cpp:2453-2458 */
	if (!bloom->counting && p->min_threshold != 1) {
		p->min_threshold = 1;
	}
	/* ntedit.cpp:2467-2475: the x/y range check is a contradiction and never fires */
	/* ntedit.cpp:2478-2493 */
	if ((p->max_insertions == 0 && p->max_deletions > 0) ||
	    (p->max_insertions == 1 && p->max_deletions > 1)) {
		p->max_deletions = p->max_insertions;
	}
	if (p->max_insertions > 5) {
		p->max_insertions = 5;
	}
	if (p->max_deletions > 10) {
		p->max_deletions = 10;
	}
	return 0;
}

/* -------------------------------------------------------- edited-seq "rope" */
/* ntedit.cpp:613-620 */
typedef struct
{
	int node_type; /* -1 unset, 0 position range, 1 character */
	size_t s_pos, e_pos;
	unsigned char c;
	unsigned num_support;
} seqNode;

typedef struct
{
	seqNode* v;
	size_t size, cap;
} nodeVec;

static const seqNode UNSET_NODE = { -1, 0, 0, 0, 0 };

static void
nv_push(nodeVec* nv, seqNode n)
{
	if (nv->size == nv->cap) {
		nv->cap = nv->cap ? nv->cap * 2 : 16;
		nv->v = (seqNode*)realloc(nv->v, nv->cap * sizeof(seqNode));
	}
	nv->v[nv->size++] = n;
}

/* (U1) out-of-range reads give an unset node */
static seqNode
nv_get(const nodeVec* nv, size_t i)
{
	return i < nv->size ? nv->v[i] : UNSET_NODE;
}

static void
nv_put(nodeVec* nv, size_t idx, seqNode n)
{
	/* "if idx < size assign else push_back" idiom of makeInsertion */
	if (idx < nv->size) {
		nv->v[idx] = n;
	} else {
		nv_push(nv, n);
	}
}

/* substitution records: ntedit.cpp:599-611 + std::queue */
typedef struct
{
	unsigned pos;
	unsigned char draft_char, sub_base;
	unsigned num_support;
	unsigned char altbase1;
	unsigned altsupp1;
	unsigned char altbase2;
	unsigned altsupp2;
	unsigned char altbase3;
	unsigned altsupp3;
} sRec;

typedef struct
{
	sRec* v;
	size_t size, cap, head;
} recQueue;

static void
rq_push(recQueue* q, sRec r)
{
	if (q->size == q->cap) {
		q->cap = q->cap ? q->cap * 2 : 64;
		q->v = (sRec*)realloc(q->v, q->cap * sizeof(sRec));
	}
	q->v[q->size++] = r;
}

/* ntedit.cpp:486-499 */
static int
isATGCBase(unsigned char C)
{
	return C == 'A' || C == 'T' || C == 'G' || C == 'C';
}

static int
isAcceptedBase(unsigned char C)
{
	return C == 'A' || C == 'T' || C == 'G' || C == 'C' || C == 'R' || C == 'Y' || C == 'S' ||
	       C == 'W' || C == 'K' || C == 'M' || C == 'B' || C == 'D' || C == 'H' || C == 'V';
}

/* ntedit.cpp:501-520 */
static char
RC(unsigned char C)
{
	switch (C) {
	case 'A':
	case 'a':
		return 'T';
	case 'T':
	case 't':
		return 'A';
	case 'G':
	case 'g':
		return 'C';
	case 'C':
	case 'c':
		return 'G';
	default:
		return 'N';
	}
}

/* ntedit.cpp:524-545 */
static unsigned
findFirstAcceptedKmer(unsigned b_i, const char* seq, unsigned len, unsigned k)
{
	for (unsigned i = b_i; (uint64_t)i + k < len;) {
		if (isAcceptedBase((unsigned char)toupper((unsigned char)seq[i]))) {
			int good = 1;
			for (unsigned j = i + 1; j < i + k; j++) {
				if (!isAcceptedBase((unsigned char)toupper((unsigned char)seq[j]))) {
					good = 0;
					i = j + 1;
					break;
				}
			}
			if (good) {
				return i;
			}
		} else {
			i++;
		}
	}
	return len - 1;
}

/* ntedit.cpp:561-596: KMP failure function test for "s is a power of a word" */
static int
isRepeatInsertion(const char* s, int n)
{
	if (n <= 0) {
		/* reference: lps(0) then lps[0]=0 is UB on an empty string; the call
		 * sites only reach here with n >= 1 except prev_insertion=="" at
		 * ntedit.cpp:1319, where size()+indel >= k cannot hold for k > 5. */
		return 0;
	}
	int* lps = (int*)malloc(sizeof(int) * (size_t)n);
	int len = 0, i = 1;
	lps[0] = 0;
	while (i < n) {
		if (s[i] == s[len]) {
			len++;
			lps[i] = len;
			i++;
		} else if (len != 0) {
			len = lps[len - 1];
		} else {
			lps[i] = 0;
			i++;
		}
	}
	len = lps[n - 1];
	free(lps);
	return len > 0 && n % (n - len) == 0;
}

/* ntedit.cpp:625-714 */
static void
makeInsertion(
    unsigned* t_node_index,
    unsigned insert_pos,
    const char* ins,
    unsigned n_ins,
    unsigned num_support,
    nodeVec* nv)
{
	seqNode orig = nv_get(nv, *t_node_index);
	seqNode to_insert[16];
	for (unsigned i = 0; i < n_ins; i++) {
		to_insert[i].node_type = 1;
		to_insert[i].s_pos = 0;
		to_insert[i].e_pos = 0;
		to_insert[i].c = (unsigned char)ins[i];
		to_insert[i].num_support = num_support;
	}
	if ((orig.node_type == 0 && insert_pos <= orig.s_pos) || orig.node_type == 1) {
		/* gather nodes following this insertion, blank them, re-append after */
		size_t i = *t_node_index;
		size_t n_re = 0, cap_re = 16;
		seqNode* re = (seqNode*)malloc(cap_re * sizeof(seqNode));
		while (i < nv->size && nv->v[i].node_type != -1) {
			if (n_re == cap_re) {
				cap_re *= 2;
				re = (seqNode*)realloc(re, cap_re * sizeof(seqNode));
			}
			re[n_re++] = nv->v[i];
			nv->v[i].node_type = -1;
			i++;
		}
		for (unsigned q = 0; q < n_ins; q++) {
			nv_put(nv, *t_node_index + q, to_insert[q]);
		}
		for (size_t q = 0; q < n_re; q++) {
			nv_put(nv, *t_node_index + n_ins + q, re[q]);
		}
		free(re);
	} else if (orig.node_type == 0) {
		/* split the position node */
		seqNode after;
		after.node_type = 0;
		after.s_pos = insert_pos;
		after.e_pos = orig.e_pos;
		after.c = 0;
		after.num_support = 0;
		nv->v[*t_node_index].e_pos = (size_t)insert_pos - 1;
		for (unsigned q = 0; q < n_ins; q++) {
			nv_put(nv, *t_node_index + q + 1, to_insert[q]);
		}
		nv_put(nv, *t_node_index + n_ins + 1, after);
		(*t_node_index)++;
	}
}

/* ntedit.cpp:719-809 */
static void
makeDeletion(unsigned* t_node_index, unsigned* pos, unsigned num_del, unsigned num_support, nodeVec* nv)
{
	seqNode orig = nv_get(nv, *t_node_index);
	if (orig.node_type == 0) {
		unsigned leftover_del = 0;
		if (*pos <= orig.s_pos) {
			if ((size_t)*pos + num_del <= orig.e_pos) {
				nv->v[*t_node_index].s_pos = (size_t)*pos + num_del;
				nv->v[*t_node_index].num_support = num_support;
				*pos = (unsigned)nv->v[*t_node_index].s_pos;
				return;
			}
			leftover_del = (unsigned)((size_t)*pos + num_del - orig.e_pos);
			*pos = (unsigned)(orig.e_pos + 1);
			size_t i = (size_t)*t_node_index + 1;
			while (i < nv->size && nv->v[i].node_type != -1) {
				nv->v[i - 1] = nv->v[i];
				nv->v[i].node_type = -1;
				i++;
			}
		} else {
			if ((size_t)*pos + num_del <= orig.e_pos) {
				seqNode split;
				split.node_type = 0;
				split.s_pos = (size_t)*pos + num_del;
				split.e_pos = orig.e_pos;
				split.c = 0;
				split.num_support = num_support;
				nv->v[*t_node_index].e_pos = (size_t)*pos - 1;
				*pos = (unsigned)split.s_pos;
				(*t_node_index)++;
				nv_put(nv, *t_node_index, split);
				return;
			}
			leftover_del = (unsigned)((size_t)*pos + num_del - orig.e_pos);
			nv->v[*t_node_index].e_pos = (size_t)*pos - 1;
			*pos = (unsigned)(orig.e_pos + 1);
			(*t_node_index)++;
		}
		if (leftover_del > 0) {
			if (*t_node_index < nv->size && nv->v[*t_node_index].node_type != -1) {
				if (nv->v[*t_node_index].node_type == 0) {
					*pos = (unsigned)nv->v[*t_node_index].s_pos;
				}
				makeDeletion(t_node_index, pos, leftover_del, num_support, nv);
			}
		}
	} else if (orig.node_type == 1) {
		size_t i = *t_node_index;
		unsigned leftover_del = num_del;
		while (i < nv->size && nv->v[i].node_type == 1 && leftover_del > 0) {
			nv->v[i].node_type = -1;
			leftover_del--;
			i++;
		}
		size_t j = *t_node_index;
		while (i < nv->size && nv->v[i].node_type != -1) {
			nv->v[j] = nv->v[i];
			nv->v[i].node_type = -1;
			i++;
			j++;
		}
		if (leftover_del > 0) {
			if (*t_node_index < nv->size && nv->v[*t_node_index].node_type != -1) {
				if (nv->v[*t_node_index].node_type == 0) {
					*pos = (unsigned)nv->v[*t_node_index].s_pos;
				}
				makeDeletion(t_node_index, pos, leftover_del, num_support, nv);
			}
		}
	}
}

/* ntedit.cpp:812-823 (contigSeq.at() would throw past the end; see header) */
static unsigned char
getCharacter(unsigned pos, seqNode node, const char* seq, unsigned len)
{
	if (node.node_type == 0) {
		return pos < len ? (unsigned char)seq[pos] : 0;
	}
	if (node.node_type == 1) {
		return node.c;
	}
	return 0;
}

/* ntedit.cpp:826-844 */
static void
increment(unsigned* pos, unsigned* node_index, const nodeVec* nv)
{
	seqNode node = nv_get(nv, *node_index);
	if (node.node_type == 0) {
		(*pos)++;
		if (*pos > node.e_pos) {
			(*node_index)++;
			if (*node_index < nv->size && nv->v[*node_index].node_type == 0) {
				*pos = (unsigned)nv->v[*node_index].s_pos;
			}
		}
	} else if (node.node_type == 1) {
		(*node_index)++;
		if (*node_index < nv->size && nv->v[*node_index].node_type == 0) {
			*pos = (unsigned)nv->v[*node_index].s_pos;
		}
	}
}

/* ntedit.cpp:848-903; returns 1 and fills kmer[k] on success, else 0 (""),
 * in which case h_seq_i = t_seq_i = len */
static int
findAcceptedKmer(
    unsigned* h_seq_i,
    unsigned* t_seq_i,
    unsigned* h_node_index,
    unsigned* t_node_index,
    const char* seq,
    unsigned len,
    const nodeVec* nv,
    unsigned k,
    char* kmer)
{
	seqNode curr_node = nv_get(nv, *t_node_index);
	unsigned temp_t_node_index = *t_node_index;
	unsigned temp_h_node_index;
	unsigned i = *t_seq_i;
	while (i < len && temp_t_node_index < nv->size && nv->v[temp_t_node_index].node_type != -1) {
		unsigned char c = getCharacter(i, curr_node, seq, len);
		if (isAcceptedBase((unsigned char)toupper(c))) {
			unsigned n = 0;
			kmer[n++] = (char)c;
			temp_h_node_index = temp_t_node_index;
			unsigned j = i;
			increment(&j, &temp_t_node_index, nv);
			while (j < len && temp_t_node_index < nv->size &&
			       nv->v[temp_t_node_index].node_type != -1) {
				curr_node = nv->v[temp_t_node_index];
				c = getCharacter(j, curr_node, seq, len);
				if (!isAcceptedBase((unsigned char)toupper(c))) {
					i = j;
					break;
				}
				kmer[n++] = (char)c;
				if (n == k) {
					break;
				}
				increment(&j, &temp_t_node_index, nv);
			}
			if (n == k) {
				*h_seq_i = i;
				*t_seq_i = j;
				*h_node_index = temp_h_node_index;
				*t_node_index = temp_t_node_index;
				return 1;
			}
		}
		increment(&i, &temp_t_node_index, nv);
	}
	*h_seq_i = len;
	*t_seq_i = len;
	return 0;
}

/* ntedit.cpp:907-922; writes the string into out (cap bytes), returns length */
static unsigned
getPrevInsertion(unsigned t_seq_i, unsigned t_node_index, const nodeVec* nv, char* out, unsigned cap)
{
	unsigned n = 0;
	seqNode tn = nv_get(nv, t_node_index);
	if ((t_node_index < nv->size && tn.node_type == 0 && t_seq_i == tn.s_pos) ||
	    tn.node_type == 1) {
		t_node_index--;
	}
	while (t_node_index < nv->size && nv->v[t_node_index].node_type == 1) {
		if (n + 1 < cap) {
			out[n++] = RC(nv->v[t_node_index].c);
		}
		t_node_index--;
	}
	out[n] = 0;
	return n;
}

/* ntedit.cpp:1216-1247 */
static int
roll(
    unsigned* h_seq_i,
    unsigned* t_seq_i,
    unsigned* h_node_index,
    unsigned* t_node_index,
    const char* seq,
    unsigned len,
    const nodeVec* nv,
    unsigned char* charOut,
    unsigned char* charIn)
{
	if (*h_seq_i >= len || *h_node_index >= nv->size) {
		return 0;
	}
	*charOut = getCharacter(*h_seq_i, nv->v[*h_node_index], seq, len);
	increment(h_seq_i, h_node_index, nv);
	if (*t_seq_i >= len || *t_node_index >= nv->size) {
		return 0;
	}
	increment(t_seq_i, t_node_index, nv);
	if (*t_seq_i >= len || *t_node_index >= nv->size) {
		return 0;
	}
	*charIn = getCharacter(*t_seq_i, nv->v[*t_node_index], seq, len);
	return 1;
}

/* ------------------------------------------------ variant annotation map (-l) */
/* ntedit.cpp:2261-2274,2524-2562: every line of the -l VCF with >= 8 tab-separated fields
 * maps "CHROM>REF POS ALT" (concatenated, no blanks) to its INFO field */
struct ora_annot
{
	char** key;
	char** val;
	size_t n, cap;
	int sorted;
};

ora_annot*
ora_annot_load(const char* path)
{
	gzFile f = gzopen(path, "r"); /* plain or gzipped (the reference uses Boost for .gz) */
	if (!f) {
		return NULL;
	}
	ora_annot* m = (ora_annot*)calloc(1, sizeof(*m));
	size_t cap = 1 << 16;
	char* line = (char*)malloc(cap);
	for (;;) {
		size_t n = 0;
		int got = 0;
		while (gzgets(f, line + n, (int)(cap - n))) {
			got = 1;
			n += strlen(line + n);
			if (n && line[n - 1] == '\n') {
				break;
			}
			if (cap - n < 2) {
				cap *= 2;
				line = (char*)realloc(line, cap);
			}
		}
		if (!got) {
			break;
		}
		while (n && (line[n - 1] == '\n')) {
			line[--n] = 0;
		}
		/* split on tabs */
		char* tok[9];
		int nt = 0;
		char* s = line;
		tok[nt++] = s;
		for (char* q = line; *q; q++) {
			if (*q == '\t') {
				*q = 0;
				if (nt < 9) {
					tok[nt++] = q + 1;
				} else {
					nt++;
				}
			}
		}
		if (nt >= 8) {
			size_t kl = strlen(tok[0]) + 1 + strlen(tok[3]) + strlen(tok[1]) + strlen(tok[4]) + 1;
			char* key = (char*)malloc(kl);
			snprintf(key, kl, "%s>%s%s%s", tok[0], tok[3], tok[1], tok[4]);
			if (m->n == m->cap) {
				m->cap = m->cap ? m->cap * 2 : 1024;
				m->key = (char**)realloc(m->key, m->cap * sizeof(char*));
				m->val = (char**)realloc(m->val, m->cap * sizeof(char*));
			}
			m->key[m->n] = key;
			m->val[m->n] = strdup(tok[7]);
			m->n++;
		}
	}
	free(line);
	gzclose(f);
	/* later duplicates overwrite earlier ones (std::map operator[]=): stable sort, keep the last */
	if (m->n) {
		size_t* idx = (size_t*)malloc(m->n * sizeof(size_t));
		for (size_t i = 0; i < m->n; i++) {
			idx[i] = i;
		}
		for (size_t i = 1; i < m->n; i++) { /* simple binary-insertion sort keeps it dependency-free */
			size_t v = idx[i];
			size_t lo = 0, hi = i;
			while (lo < hi) {
				size_t mid = (lo + hi) / 2;
				if (strcmp(m->key[idx[mid]], m->key[v]) <= 0) {
					lo = mid + 1;
				} else {
					hi = mid;
				}
			}
			memmove(idx + lo + 1, idx + lo, (i - lo) * sizeof(size_t));
			idx[lo] = v;
		}
		char** k2 = (char**)malloc(m->n * sizeof(char*));
		char** v2 = (char**)malloc(m->n * sizeof(char*));
		for (size_t i = 0; i < m->n; i++) {
			k2[i] = m->key[idx[i]];
			v2[i] = m->val[idx[i]];
		}
		free(m->key);
		free(m->val);
		free(idx);
		m->key = k2;
		m->val = v2;
	}
	m->sorted = 1;
	return m;
}

void
ora_annot_free(ora_annot* m)
{
	if (!m) {
		return;
	}
	for (size_t i = 0; i < m->n; i++) {
		free(m->key[i]);
		free(m->val[i]);
	}
	free(m->key);
	free(m->val);
	free(m);
}

/* value of the LAST entry with this key, or NULL (empty values count as absent, like
 * clinvar[id].empty() in the reference) */
static const char*
annot_get(const ora_annot* m, const char* key)
{
	if (!m || !m->n) {
		return NULL;
	}
	size_t lo = 0, hi = m->n;
	while (lo < hi) { /* upper bound */
		size_t mid = (lo + hi) / 2;
		if (strcmp(m->key[mid], key) <= 0) {
			lo = mid + 1;
		} else {
			hi = mid;
		}
	}
	if (lo == 0 || strcmp(m->key[lo - 1], key) != 0) {
		return NULL;
	}
	return m->val[lo - 1][0] ? m->val[lo - 1] : NULL;
}

/* --------------------------------------------------------- polishing context */
typedef struct
{
	const ora_params* p;
	FILE* vcf;
	const ora_annot* annot;
	const ora_bf* bloom;
	const ora_bf* bloomrep;
	char* seq;
	unsigned len;
	nodeVec nv;
	recQueue subs;
	uint64_t hVal[64];
} ctx_t;

static int
bloom_has(const ctx_t* c)
{
	/* BFWrapper::contains, ntedit.cpp:368-371 */
	return ora_bf_contains(c->bloom, c->hVal) > 0;
}

static unsigned
bloom_count(const ctx_t* c)
{
	/* BFWrapper::get_count, ntedit.cpp:373-376 */
	return c->bloom->counting ? ora_bf_contains(c->bloom, c->hVal) : 1;
}

/* ntedit.cpp:465-473 */
static int
is_kmer_solid(const ctx_t* c)
{
	int solid_if_reg = !c->p->secbf || !(ora_bf_contains(c->bloomrep, c->hVal) > 0);
	int solid_if_count = !c->bloom->counting || (bloom_count(c) <= c->p->max_threshold &&
	                                             bloom_count(c) >= c->p->min_threshold);
	return solid_if_reg && solid_if_count;
}

static int
cmp_u8(const void* a, const void* b)
{
	return (int)*(const uint8_t*)a - (int)*(const uint8_t*)b;
}

/* ntedit.cpp:455-463 */
static unsigned
median_u8(uint8_t* v, size_t n)
{
	if (n > 0) {
		qsort(v, n, 1, cmp_u8);
		return v[n / 2];
	}
	return 0;
}

/* the three float threshold tests (ntedit.cpp:1531-1535, 1659-1663/1992-1997, 1867-1872) */
static int
edit_ok_sub_ins(const ora_params* p, unsigned check_present)
{
	return (!p->use_ratio && (float)check_present >= ((float)p->k / p->edit_threshold)) ||
	       (p->use_ratio && (float)check_present >= ((float)p->k / p->jump) * p->edit_ratio);
}

static int
edit_ok_del(const ora_params* p, unsigned check_present)
{
	return (!p->use_ratio && (float)check_present >= ((float)p->k / p->edit_threshold)) ||
	       (p->use_ratio &&
	        (float)check_present >= (1 + ((float)p->k / p->jump)) * p->edit_ratio);
}

static int
missing_ok(const ora_params* p, unsigned check_missing)
{
	return (!p->use_ratio && (float)check_missing >= ((float)p->k / p->missing_threshold)) ||
	       (p->use_ratio && (float)check_missing >= ((float)p->k / p->jump) * p->missing_ratio);
}

/* i-th insertion candidate for an index base: the enumeration of
 * multi_possible_bases (ntedit.cpp:203-348) is "index base followed by every
 * string over A<C<G<T of length 0,1,2,3,4 in length-then-lexicographic order";
 * generated here instead of stored. Returns the length. */
static unsigned
insertion_candidate(unsigned char index_char, unsigned i, char* out)
{
	static const char alphabet[4] = { 'A', 'C', 'G', 'T' };
	unsigned extra = 0, first = 0, count = 1;
	while (i >= first + count) {
		first += count;
		count *= 4;
		extra++;
	}
	unsigned r = i - first;
	out[0] = (char)index_char;
	for (unsigned q = 0; q < extra; q++) {
		out[extra - q] = alphabet[r & 3];
		r >>= 2;
	}
	out[extra + 1] = 0;
	return extra + 1;
}

static const unsigned num_tries[6] = { 0, 1, 5, 21, 85, 341 }; /* ntedit.cpp:172 */

/* substitution candidate lists (ntedit.cpp:180-199) */
static unsigned
candidate_bases(int snv, unsigned char draft_char, unsigned char* out)
{
	const char* s = "";
	if (snv) {
		switch (draft_char) {
		case 'A':
			s = "TCG";
			break;
		case 'T':
			s = "ACG";
			break;
		case 'C':
			s = "ATG";
			break;
		case 'G':
			s = "ATC";
			break;
		case 'R':
		case 'Y':
		case 'S':
		case 'W':
		case 'K':
		case 'M':
		case 'B':
		case 'D':
		case 'H':
		case 'V':
		case 'N':
			s = "ATCG";
			break;
		default:
			s = "";
		}
	} else {
		switch (draft_char) {
		case 'A':
			s = "TCG";
			break;
		case 'T':
			s = "ACG";
			break;
		case 'C':
			s = "ATG";
			break;
		case 'G':
			s = "ATC";
			break;
		case 'R':
			s = "TC";
			break;
		case 'Y':
			s = "AG";
			break;
		case 'S':
			s = "AT";
			break;
		case 'W':
			s = "CG";
			break;
		case 'K':
			s = "AC";
			break;
		case 'M':
			s = "TG";
			break;
		case 'B':
			s = "A";
			break;
		case 'D':
			s = "C";
			break;
		case 'H':
			s = "G";
			break;
		case 'V':
			s = "T";
			break;
		case 'N':
			s = "ATCG";
			break;
		default:
			s = "";
		}
	}
	unsigned n = 0;
	while (s[n]) {
		out[n] = (unsigned char)s[n];
		n++;
	}
	return n;
}

/* ntedit.cpp:1451-1545; deleted_bases is appended to (caller passes empty) */
static int
tryDeletion(
    ctx_t* c,
    unsigned char draft_char,
    unsigned num_deletions,
    unsigned h_seq_i,
    unsigned t_seq_i,
    unsigned h_node_index,
    unsigned t_node_index,
    uint64_t fhVal,
    uint64_t rhVal,
    char* deleted_bases)
{
	const ora_params* p = c->p;
	uint64_t temp_fhVal = fhVal, temp_rhVal = rhVal;
	unsigned temp_h_seq_i = h_seq_i, temp_t_seq_i = t_seq_i;
	unsigned temp_h_node_index = h_node_index, temp_t_node_index = t_node_index;
	unsigned char charOut = 0, charIn = 0;
	unsigned nd = 0;
	for (unsigned i = 0; i < num_deletions; i++) {
		deleted_bases[nd++] =
		    (char)getCharacter(temp_t_seq_i, nv_get(&c->nv, temp_t_node_index), c->seq, c->len);
		increment(&temp_t_seq_i, &temp_t_node_index, &c->nv);
	}
	deleted_bases[nd] = 0;
	NTMC64_changelast(
	    draft_char,
	    getCharacter(temp_t_seq_i, nv_get(&c->nv, temp_t_node_index), c->seq, c->len),
	    p->k,
	    p->h,
	    &temp_fhVal,
	    &temp_rhVal,
	    c->hVal);
	unsigned check_present = 0;
	if (bloom_has(c) && is_kmer_solid(c)) {
		check_present++;
	}
	for (unsigned k = 1; k <= (p->k - 2) && temp_h_seq_i < c->len; k++) {
		if (roll(
		        &temp_h_seq_i,
		        &temp_t_seq_i,
		        &temp_h_node_index,
		        &temp_t_node_index,
		        c->seq,
		        c->len,
		        &c->nv,
		        &charOut,
		        &charIn)) {
			NTMC64_roll(charOut, charIn, p->k, p->h, &temp_fhVal, &temp_rhVal, c->hVal);
			if (k % p->jump == 0 && bloom_has(c) && is_kmer_solid(c)) {
				check_present++;
			}
		}
	}
	if (edit_ok_del(p, check_present)) {
		return (int)check_present;
	}
	return 0;
}

typedef struct
{
	unsigned best_edit_type;
	char best_indel[16];
	char alt_indel[16];
	unsigned char best_sub_base;
	unsigned best_num_support;
	unsigned char altbase1, altbase2, altbase3;
	unsigned altsupp1, altsupp2, altsupp3;
} best_t;

/* ntedit.cpp:1548-1744 */
static int
tryIndels(
    ctx_t* c,
    unsigned char draft_char,
    unsigned char index_char,
    unsigned* num_deletions,
    unsigned h_seq_i,
    unsigned t_seq_i,
    unsigned h_node_index,
    unsigned t_node_index,
    uint64_t fhVal,
    uint64_t rhVal,
    best_t* b)
{
	const ora_params* p = c->p;
	uint64_t temp_fhVal, temp_rhVal;
	unsigned temp_h_seq_i, temp_t_seq_i, temp_h_node_index, temp_t_node_index;
	unsigned temp_best_num_support = 0, temp_alt_num_support = 0;
	char temp_best_indel[16] = "", temp_alt_indel[16] = "";
	unsigned temp_best_edit_type = 0;
	unsigned char charIn = 0, charOut = 0;

	for (unsigned i = 0; i < num_tries[p->max_insertions]; i++) {
		char insertion_bases[16];
		unsigned n_ins = insertion_candidate(index_char, i, insertion_bases);
		insertion_bases[n_ins++] = (char)draft_char;
		insertion_bases[n_ins] = 0;

		temp_fhVal = fhVal;
		temp_rhVal = rhVal;
		temp_h_seq_i = h_seq_i;
		temp_t_seq_i = t_seq_i;
		temp_h_node_index = h_node_index;
		temp_t_node_index = t_node_index;

		NTMC64_changelast(draft_char, index_char, p->k, p->h, &temp_fhVal, &temp_rhVal, c->hVal);
		unsigned check_present = 0;
		unsigned k = 0;
		for (; k < n_ins - 1 && temp_h_seq_i < c->len; k++) {
			NTMC64_roll(
			    getCharacter(temp_h_seq_i, nv_get(&c->nv, temp_h_node_index), c->seq, c->len),
			    (unsigned char)insertion_bases[k + 1],
			    p->k,
			    p->h,
			    &temp_fhVal,
			    &temp_rhVal,
			    c->hVal);
			increment(&temp_h_seq_i, &temp_h_node_index, &c->nv);
			if (k % p->jump == 0 && bloom_has(c) && is_kmer_solid(c)) {
				check_present++;
			}
		}
		for (; k < p->k - 1 && temp_h_seq_i < c->len; k++) {
			if (roll(
			        &temp_h_seq_i,
			        &temp_t_seq_i,
			        &temp_h_node_index,
			        &temp_t_node_index,
			        c->seq,
			        c->len,
			        &c->nv,
			        &charOut,
			        &charIn)) {
				NTMC64_roll(charOut, charIn, p->k, p->h, &temp_fhVal, &temp_rhVal, c->hVal);
				if (k % p->jump == 0 && bloom_has(c) && is_kmer_solid(c)) {
					check_present++;
				}
			}
		}
		insertion_bases[--n_ins] = 0; /* pop_back */
		if (edit_ok_sub_ins(p, check_present)) {
			if (p->mode == 0) {
				b->best_edit_type = 2;
				strcpy(b->best_indel, insertion_bases);
				b->best_num_support = check_present;
				return 1;
			}
			if (p->mode == 1 || p->mode == 2) {
				if (check_present >= temp_best_num_support) {
					if (temp_best_num_support) {
						strcpy(temp_alt_indel, temp_best_indel);
						temp_alt_num_support = temp_best_num_support;
					}
					temp_best_edit_type = 2;
					strcpy(temp_best_indel, insertion_bases);
					temp_best_num_support = check_present;
				}
			}
		}

		if (*num_deletions <= p->max_deletions) {
			char deleted_bases[16];
			unsigned del_support = (unsigned)tryDeletion(
			    c,
			    draft_char,
			    *num_deletions,
			    h_seq_i,
			    t_seq_i,
			    h_node_index,
			    t_node_index,
			    fhVal,
			    rhVal,
			    deleted_bases);
			if (del_support > 0) {
				if (p->mode == 0) {
					b->best_edit_type = 3;
					strcpy(b->best_indel, deleted_bases);
					b->best_num_support = del_support;
					return 1;
				}
				if (p->mode == 1 || p->mode == 2) {
					if (del_support >= temp_best_num_support) {
						if (temp_best_num_support) {
							strcpy(temp_alt_indel, temp_best_indel);
							temp_alt_num_support = temp_best_num_support;
						}
						temp_best_edit_type = 3;
						strcpy(temp_best_indel, deleted_bases);
						temp_best_num_support = del_support;
					}
				}
			}
			(*num_deletions)++;
		}
	}

	if (temp_best_num_support > 0) {
		if ((p->mode == 2 && temp_best_num_support > b->best_num_support) || p->mode == 1) {
			b->best_edit_type = temp_best_edit_type;
			strcpy(b->best_indel, temp_best_indel);
			b->best_num_support = temp_best_num_support;
			strcpy(b->alt_indel, temp_alt_indel);
			b->altsupp1 = temp_alt_num_support;
		}
		return 1;
	}
	return 0;
}

/* removal of a previous insertion run (ntedit.cpp:1321-1334 / 1352-1366) */
static void
drop_prev_insertion(ctx_t* c, unsigned t_seq_i, unsigned t_node_index, unsigned count)
{
	nodeVec* nv = &c->nv;
	unsigned j = 1;
	seqNode tn = nv_get(nv, t_node_index);
	if (tn.node_type == 0 && t_seq_i == tn.s_pos) {
		j = 0;
	}
	for (unsigned i = count; i > 0; i--) {
		size_t dst = (size_t)t_node_index - i; /* wraps like the reference if i > t_node_index */
		if ((size_t)t_node_index + j < nv->size && nv->v[t_node_index + j].node_type != -1) {
			if (dst < nv->size) {
				nv->v[dst] = nv->v[t_node_index + j];
			}
			nv->v[t_node_index + j].node_type = -1;
			j++;
		} else if (dst < nv->size) {
			nv->v[dst].node_type = -1;
		}
	}
}

/* ntedit.cpp:1250-1448 */
static void
makeEdit(
    ctx_t* c,
    unsigned char draft_char,
    best_t* b,
    unsigned* h_seq_i,
    unsigned* t_seq_i,
    unsigned* h_node_index,
    unsigned* t_node_index,
    uint64_t* fhVal,
    uint64_t* rhVal)
{
	const ora_params* p = c->p;
	int skipped_repeat = 0;
	seqNode tNode = nv_get(&c->nv, *t_node_index);
	char kmer[256];
	switch (b->best_edit_type) {
	case 1:
		if (tNode.node_type == 0) {
			c->seq[*t_seq_i] = (char)b->best_sub_base;
			sRec s;
			memset(&s, 0, sizeof s);
			s.draft_char = draft_char;
			s.pos = *t_seq_i;
			s.sub_base = b->best_sub_base;
			s.num_support = b->best_num_support;
			if (b->altsupp1 && b->altbase1 != b->best_sub_base) {
				s.altbase1 = b->altbase1;
				s.altsupp1 = b->altsupp1;
			}
			if (b->altsupp2 && b->altbase2 != b->altbase1) {
				s.altbase2 = b->altbase2;
				s.altsupp2 = b->altsupp2;
			}
			if (b->altsupp3 && b->altbase3 != b->altbase2) {
				s.altbase3 = b->altbase3;
				s.altsupp3 = b->altsupp3;
			}
			rq_push(&c->subs, s);
		} else if (tNode.node_type == 1) {
			c->nv.v[*t_node_index].c = b->best_sub_base;
		}
		NTMC64_changelast(draft_char, b->best_sub_base, p->k, p->h, fhVal, rhVal, c->hVal);
		break;
	case 2: {
		unsigned cap = (unsigned)c->nv.size + 32;
		char* prev = (char*)malloc(cap + 16);
		unsigned n_prev = getPrevInsertion(*t_seq_i, *t_node_index, &c->nv, prev, cap);
		unsigned n_ind = (unsigned)strlen(b->best_indel);
		if (n_prev + n_ind >= p->k) {
			if (isRepeatInsertion(prev, (int)n_prev) || n_prev + n_ind >= p->insertion_cap) {
				drop_prev_insertion(c, *t_seq_i, *t_node_index, n_prev);
				if (findAcceptedKmer(
				        h_seq_i, t_seq_i, h_node_index, t_node_index, c->seq, c->len, &c->nv, p->k, kmer)) {
					NTMC64_seed(kmer, p->k, p->h, fhVal, rhVal, c->hVal);
				} else {
					/* reference hashes the first k bytes of "" (UB); define as zeros */
					memset(kmer, 0, sizeof kmer);
					NTMC64_seed(kmer, p->k, p->h, fhVal, rhVal, c->hVal);
				}
				skipped_repeat = 1;
			} else {
				for (unsigned w = 0; w < n_ind; w++) {
					memmove(prev + 1, prev, n_prev + 1);
					prev[0] = RC((unsigned char)b->best_indel[w]);
					n_prev++;
					if (isRepeatInsertion(prev, (int)n_prev)) {
						drop_prev_insertion(c, *t_seq_i, *t_node_index, n_prev - w);
						if (findAcceptedKmer(
						        h_seq_i,
						        t_seq_i,
						        h_node_index,
						        t_node_index,
						        c->seq,
						        c->len,
						        &c->nv,
						        p->k,
						        kmer)) {
							NTMC64_seed(kmer, p->k, p->h, fhVal, rhVal, c->hVal);
						} else {
							memset(kmer, 0, sizeof kmer);
							NTMC64_seed(kmer, p->k, p->h, fhVal, rhVal, c->hVal);
						}
						skipped_repeat = 1;
					}
				}
			}
		}
		free(prev);
		if (!skipped_repeat) {
			makeInsertion(t_node_index, *t_seq_i, b->best_indel, n_ind, b->best_num_support, &c->nv);
			NTMC64_changelast(
			    draft_char, (unsigned char)b->best_indel[0], p->k, p->h, fhVal, rhVal, c->hVal);
		}
		break;
	}
	case 3:
		makeDeletion(
		    t_node_index, t_seq_i, (unsigned)strlen(b->best_indel), b->best_num_support, &c->nv);
		NTMC64_changelast(
		    draft_char,
		    getCharacter(*t_seq_i, nv_get(&c->nv, *t_node_index), c->seq, c->len),
		    p->k,
		    p->h,
		    fhVal,
		    rhVal,
		    c->hVal);
		break;
	case 0:
		if (p->mask) {
			if (tNode.node_type == 0) {
				c->seq[*t_seq_i] = (char)tolower(draft_char);
			} else if (tNode.node_type == 1) {
				c->nv.v[*t_node_index].c = (unsigned char)tolower(draft_char);
			}
			NTMC64_changelast(
			    draft_char, (unsigned char)tolower(draft_char), p->k, p->h, fhVal, rhVal, c->hVal);
		}
		if (p->snv) {
			sRec s;
			memset(&s, 0, sizeof s);
			s.draft_char = draft_char;
			s.pos = *t_seq_i;
			s.sub_base = draft_char;
			s.num_support = b->best_num_support;
			s.altbase1 = b->altbase1;
			s.altsupp1 = b->altsupp1;
			s.altbase2 = b->altbase2;
			s.altsupp2 = b->altsupp2;
			s.altbase3 = b->altbase3;
			s.altsupp3 = b->altsupp3;
			if (b->altsupp1) {
				rq_push(&c->subs, s);
			}
		}
		break;
	default:
		break;
	}
}

/* ntedit.cpp:925-1213 (FASTA + TSV only; the VCF body is outside the parity contract) */
/* "^" + annotation of the variant id, or "^NA" (ntedit.cpp:964-969 and the like) */
static void
vcf_annot(FILE* vcf, const ora_annot* m, const char* id)
{
	const char* v = annot_get(m, id);
	fprintf(vcf, "^%s", v ? v : "NA");
}

static void
upper_into(char* dst, const char* src, size_t n)
{
	for (size_t i = 0; i < n; i++) {
		dst[i] = (char)toupper((unsigned char)src[i]);
	}
	dst[n] = 0;
}

/* the substitution line of the VCF (ntedit.cpp:986-1162) */
static void
vcf_substitution(FILE* vcf, const ora_annot* m, const char* hdr, const ora_params* p, const sRec* s)
{
	const int snv_mode_no_edit = !(p->snv && s->draft_char == s->sub_base);
	char base[8];
	char support[64];
	char annot[4][16];
	(void)annot;
	size_t idcap = strlen(hdr) + 64;
	char* id = (char*)malloc(idcap);
	/* collected annotations are written at the end, in the reference's order */
	const char* ann[3];
	char* ann_owned[3] = { NULL, NULL, NULL };
	int n_ann = 0;
#define ADD_ANN(idstr)                                                                           \
	do {                                                                                         \
		const char* v_ = annot_get(m, (idstr));                                                  \
		ann[n_ann++] = v_ ? v_ : "NA";                                                           \
	} while (0)
	snprintf(base, sizeof base, "%c", s->sub_base);
	snprintf(support, sizeof support, "%u", s->num_support);
	const char D = (char)toupper(s->draft_char);
	snprintf(id, idcap, "%s>%c%u%c", hdr, D, s->pos + 1, D);
	ADD_ANN(id);
	if (snv_mode_no_edit) {
		snprintf(id, idcap, "%s>%c%u%c", hdr, D, s->pos + 1, (char)toupper((unsigned char)base[0]));
		ADD_ANN(id);
	}
	unsigned char ab[3];
	unsigned as[3];
	int na = 0;
	if (s->altsupp1 > 0) {
		ab[na] = s->altbase1;
		as[na++] = s->altsupp1;
	}
	if (s->altsupp2 > 0) {
		ab[na] = s->altbase2;
		as[na++] = s->altsupp2;
	}
	if (s->altsupp3 > 0) {
		ab[na] = s->altbase3;
		as[na++] = s->altsupp3;
	}
	unsigned curr_best = 0;
	char best_alt_base = '1';
	const char* genotype;
	if (na) {
		if (p->snv) {
			if (!snv_mode_no_edit) {
				for (int i = 0; i < na; i++) {
					if (as[i] > curr_best) {
						curr_best = as[i];
						best_alt_base = (char)ab[i];
					}
				}
				snprintf(base, sizeof base, "%c", best_alt_base);
				snprintf(id, idcap, "%s>%c%u%c", hdr, D, s->pos + 1, (char)toupper((unsigned char)best_alt_base));
				ADD_ANN(id);
				snprintf(support, sizeof support, "%u,%u", s->num_support, curr_best);
				genotype = "0/1";
			} else {
				int ref = 0;
				for (int i = 0; i < na; i++) {
					if (s->draft_char == ab[i]) {
						curr_best = as[i];
						ref = 1;
						break;
					}
					if (as[i] > curr_best) {
						curr_best = as[i];
						best_alt_base = (char)ab[i];
					}
				}
				if (ref) {
					snprintf(support, sizeof support, "%u,%u", curr_best, s->num_support);
					genotype = "0/1";
				} else {
					genotype = "1/2";
					snprintf(support, sizeof support, "%u,%u", s->num_support, curr_best);
					snprintf(base, sizeof base, "%c,%c", s->sub_base, best_alt_base);
					snprintf(id, idcap, "%s>%c%u%c", hdr, D, s->pos + 1, (char)toupper((unsigned char)best_alt_base));
					ADD_ANN(id);
				}
			}
		} else {
			for (int i = 0; i < na; i++) {
				if (s->draft_char == ab[i]) {
					continue;
				}
				if (as[i] > curr_best) {
					curr_best = as[i];
					best_alt_base = (char)ab[i];
				}
			}
			genotype = "1/2";
			snprintf(support, sizeof support, "%u,%u", s->num_support, curr_best);
			snprintf(base, sizeof base, "%c,%c", s->sub_base, best_alt_base);
			snprintf(id, idcap, "%s>%c%u%c", hdr, D, s->pos + 1, (char)toupper((unsigned char)best_alt_base));
			ADD_ANN(id);
		}
	} else {
		genotype = "1/1";
	}
#undef ADD_ANN
	fprintf(vcf, "%s\t%u\t.\t%c\t%s\t.\tPASS\tAD=%s", hdr, s->pos + 1, s->draft_char, base, support);
	for (int i = 0; i < n_ann; i++) {
		fprintf(vcf, "^%s", ann[i]);
	}
	fprintf(vcf, "\tGT\t%s\n", genotype);
	(void)ann_owned;
	free(id);
}

static void
writeEditsToFile(FILE* fa, FILE* tsv, const char* hdr, ctx_t* c)
{
	const ora_params* p = c->p;
	FILE* vcf = c->vcf;
	nodeVec* nv = &c->nv;
	recQueue* q = &c->subs;
	const char* seq = c->seq;
	if (fa) {
		fprintf(fa, ">%s\n", hdr);
	}
	size_t node_index = 0;
	char* insertion_bases = (char*)malloc(nv->size + 2);
	size_t n_ins = 0;
	int num_support = -1;
	unsigned pos = 0;
	seqNode curr = nv_get(nv, node_index);
	while (node_index < nv->size && curr.node_type != -1) {
		if (curr.node_type == 0) {
			if (n_ins) {
				/* (U4) */
				unsigned char draft_char = curr.s_pos >= n_ins ? (unsigned char)seq[curr.s_pos - n_ins] : (unsigned char)'N';
				insertion_bases[n_ins] = 0;
				if (tsv) {
					fprintf(tsv, "%s\t%u\t%c\t+%s\t%d\n", hdr, pos, draft_char, insertion_bases, num_support);
				}
				if (vcf) {
					/* ntedit.cpp:954-977 */
					size_t idcap = strlen(hdr) + n_ins + 64;
					char* id = (char*)malloc(idcap);
					char* up = (char*)malloc(n_ins + 2);
					upper_into(up, insertion_bases, n_ins);
					snprintf(id, idcap, "%s>%c%u%c%s", hdr, (char)toupper(draft_char), pos, (char)toupper(draft_char), up);
					fprintf(vcf, "%s\t%u\t.\t%c\t%c%s\t.\tPASS\tAD=%d", hdr, pos, draft_char, draft_char, insertion_bases, num_support);
					vcf_annot(vcf, c->annot, id);
					fprintf(vcf, "\tGT\t1/1\n");
					free(id);
					free(up);
				}
				n_ins = 0;
				num_support = -1;
			}
			while (q->head < q->size && q->v[q->head].pos <= curr.e_pos) {
				const sRec* s = &q->v[q->head];
				int snv_mode_no_edit = !(p->snv && s->draft_char == s->sub_base);
				if (snv_mode_no_edit && tsv) {
					fprintf(tsv, "%s\t%u\t%c\t%c\t%u", hdr, s->pos + 1, s->draft_char, s->sub_base, s->num_support);
					if (s->altsupp1 > 0) {
						fprintf(tsv, "\t%c\t%u", s->altbase1, s->altsupp1);
					}
					if (s->altsupp2 > 0) {
						fprintf(tsv, "\t%c\t%u", s->altbase2, s->altsupp2);
					}
					if (s->altsupp3 > 0) {
						fprintf(tsv, "\t%c\t%u", s->altbase3, s->altsupp3);
					}
					fputc('\n', tsv);
				}
				if (vcf) {
					vcf_substitution(vcf, c->annot, hdr, p, s);
				}
				q->head++;
			}
			if (fa) {
				fwrite(seq + curr.s_pos, 1, curr.e_pos - curr.s_pos + 1, fa);
			}
			pos = (unsigned)(curr.e_pos + 1);
		} else if (curr.node_type == 1) {
			insertion_bases[n_ins++] = (char)curr.c;
			if (num_support == -1) {
				num_support = (int)curr.num_support;
			}
			if (fa) {
				fputc(curr.c, fa);
			}
		}
		node_index++;
		if (node_index < nv->size) {
			curr = nv->v[node_index];
			if (curr.node_type == 0 && curr.s_pos != pos) {
				if (tsv) {
					fprintf(tsv, "%s\t%u\t%c\t-", hdr, pos, seq[pos]);
					fwrite(seq + pos, 1, curr.s_pos - pos, tsv);
					fprintf(tsv, "\t%u\n", curr.num_support);
				}
				if (vcf && pos > 0) {
					/* ntedit.cpp:1184-1208 */
					size_t dl = (size_t)(curr.s_pos - pos) + 1;
					size_t idcap = strlen(hdr) + dl + 64;
					char* id = (char*)malloc(idcap);
					char* up = (char*)malloc(dl + 1);
					upper_into(up, seq + pos - 1, dl);
					snprintf(id, idcap, "%s>%s%u%c", hdr, up, pos, (char)toupper((unsigned char)seq[pos - 1]));
					fprintf(vcf, "%s\t%u\t.\t", hdr, pos);
					fwrite(seq + pos - 1, 1, dl, vcf);
					fprintf(vcf, "\t%c\t.\tPASS\tAD=%u", seq[pos - 1], curr.num_support);
					vcf_annot(vcf, c->annot, id);
					fprintf(vcf, "\tGT\t1/1\n");
					free(id);
					free(up);
				}
			}
		}
	}
	if (fa) {
		fputc('\n', fa);
	}
	free(insertion_bases);
}

void
ora_write_tsv_header(FILE* tsv, const ora_params* p, const ora_bf* bloom)
{
	/* ntedit.cpp:2175-2188 */
	fprintf(tsv, "ID\tbpPosition+1\tOriginalBase\tNewBase\t");
	if (bloom->counting) {
		fprintf(tsv, "Coverage (max 255)");
	} else {
		fprintf(tsv, "Support %u-mer (out of %g)", p->k, ceil((double)p->k / (double)p->jump));
	}
	const char* alt = bloom->counting ? "Coverage" : "Support";
	fprintf(tsv, "\tAlt.Base1\tAlt.%s1\tAlt.Base2\tAlt.%s2\tAlt.Base3\tAlt.%s3\n", alt, alt, alt);
}

void
ora_polish_contig(
    const char* hdr,
    char* seq,
    unsigned len,
    const ora_params* p,
    const ora_bf* bloom,
    const ora_bf* bloomrep,
    FILE* fa,
    FILE* tsv)
{
	ora_polish_contig_vcf(hdr, seq, len, p, bloom, bloomrep, fa, tsv, NULL, NULL);
}

/* ntedit.cpp:1747-2151 */
void
ora_polish_contig_vcf(
    const char* hdr,
    char* seq,
    unsigned len,
    const ora_params* p,
    const ora_bf* bloom,
    const ora_bf* bloomrep,
    FILE* fa,
    FILE* tsv,
    FILE* vcf,
    const ora_annot* annot)
{
	ctx_t cx;
	ctx_t* c = &cx;
	memset(c, 0, sizeof(*c));
	c->vcf = vcf;
	c->annot = annot;
	c->p = p;
	c->bloom = bloom;
	c->bloomrep = bloomrep;
	c->seq = seq;
	c->len = len;

	uint64_t fhVal = 0, rhVal = 0;
	unsigned char charIn = 0, charOut = 0, draft_char;

	unsigned h_seq_i = findFirstAcceptedKmer(0, seq, len, p->k);
	unsigned t_seq_i = h_seq_i + p->k - 1;
	if (h_seq_i + p->k - 1 < len) {
		NTMC64_seed(seq + h_seq_i, p->k, p->h, &fhVal, &rhVal, c->hVal);
		charIn = (unsigned char)seq[t_seq_i];
	}

	seqNode root;
	root.node_type = 0;
	root.s_pos = 0;
	root.e_pos = (size_t)len - 1;
	root.c = 0;
	root.num_support = 0;
	nv_push(&c->nv, root);
	unsigned h_node_index = 0, t_node_index = 0;

	int continue_edit = 1;
	do {
		if (h_seq_i + p->k - 1 >= len) {
			break;
		}
		if (p->snv || !bloom_has(c) || (bloom->counting && bloom_count(c) < p->min_threshold)) {
			uint64_t temp_fhVal = fhVal, temp_rhVal = rhVal;
			unsigned temp_h_seq_i = h_seq_i, temp_t_seq_i = t_seq_i;
			unsigned temp_h_node_index = h_node_index, temp_t_node_index = t_node_index;

			draft_char = (unsigned char)toupper(charIn);

			unsigned check_missing = 0, check_there = 0, check_there_median = 0;
			uint8_t med_vec[256];
			size_t n_med = 0;
			int do_not_fix = 0;

			for (unsigned k = 0; k < p->k && temp_h_seq_i < len; k++) {
				if (roll(
				        &temp_h_seq_i,
				        &temp_t_seq_i,
				        &temp_h_node_index,
				        &temp_t_node_index,
				        seq,
				        len,
				        &c->nv,
				        &charOut,
				        &charIn)) {
					NTMC64_roll(charOut, charIn, p->k, p->h, &temp_fhVal, &temp_rhVal, c->hVal);
					if (!isAcceptedBase((unsigned char)toupper(charIn))) {
						do_not_fix = 1;
						break;
					}
					if (k % p->jump == 0 && !bloom_has(c)) {
						check_missing++;
					} else if (
					    isATGCBase(draft_char) && k % p->jump == 0 && bloom_has(c) &&
					    (!bloom->counting || bloom_count(c) >= p->min_threshold)) {
						check_there++;
						if (bloom->counting && n_med < sizeof med_vec) {
							med_vec[n_med++] = (uint8_t)bloom_count(c);
						}
					}
				} else {
					do_not_fix = 1;
					break;
				}
			}
			if (bloom->counting) {
				check_there_median = median_u8(med_vec, n_med);
			}
			if (p->snv ||
			    (!do_not_fix && (missing_ok(p, check_missing) ||
			                     (bloom->counting && check_there_median < p->min_threshold)))) {
				unsigned num_deletions = 1;
				best_t b;
				memset(&b, 0, sizeof b); /* (U2) */

				if (p->snv) {
					if (edit_ok_sub_ins(p, check_there)) {
						b.best_sub_base = draft_char;
						b.best_num_support = bloom->counting ? check_there_median : check_there;
					}
				}

				unsigned char cand[4];
				unsigned n_cand = candidate_bases(p->snv, draft_char, cand);
				for (unsigned ci = 0; ci < n_cand; ci++) {
					unsigned char sub_base = cand[ci];
					temp_fhVal = fhVal;
					temp_rhVal = rhVal;
					NTMC64_changelast(
					    draft_char, sub_base, p->k, p->h, &temp_fhVal, &temp_rhVal, c->hVal);
					if ((bloom_has(c) && is_kmer_solid(c)) || p->mode == 2) {
						temp_h_node_index = h_node_index;
						temp_t_node_index = t_node_index;
						temp_h_seq_i = h_seq_i;
						temp_t_seq_i = t_seq_i;

						seqNode tn = nv_get(&c->nv, t_node_index);
						if (tn.node_type == 0) {
							seq[temp_t_seq_i] = (char)sub_base;
						} else if (tn.node_type == 1) {
							c->nv.v[t_node_index].c = sub_base;
						}
						unsigned check_present = 0;
						for (unsigned k = 0; k < p->k && temp_h_seq_i < len && temp_t_seq_i < len;
						     k++) {
							if (roll(
							        &temp_h_seq_i,
							        &temp_t_seq_i,
							        &temp_h_node_index,
							        &temp_t_node_index,
							        seq,
							        len,
							        &c->nv,
							        &charOut,
							        &charIn)) {
								NTMC64_roll(
								    charOut, charIn, p->k, p->h, &temp_fhVal, &temp_rhVal, c->hVal);
								if (k % p->jump == 0 && bloom_has(c) && is_kmer_solid(c)) {
									check_present++;
								}
							} else {
								break;
							}
						}
						/* revert (note: writes the UPPER-cased draft char) */
						if (tn.node_type == 0) {
							seq[t_seq_i] = (char)draft_char;
						} else if (tn.node_type == 1) {
							c->nv.v[t_node_index].c = draft_char;
						}

						if (edit_ok_sub_ins(p, check_present)) {
							if (check_present >= b.best_num_support) {
								if (b.altsupp2) {
									b.altbase3 = b.altbase2;
									b.altsupp3 = b.altsupp2;
								}
								if (b.altsupp1) {
									b.altbase2 = b.altbase1;
									b.altsupp2 = b.altsupp1;
								}
								if (b.best_num_support) {
									b.altsupp1 = b.best_num_support;
									b.altbase1 = b.best_sub_base;
								}
								b.best_edit_type = 1;
								b.best_sub_base = sub_base;
								b.best_num_support = check_present;
							} else {
								if (!b.altsupp1) {
									b.altbase1 = sub_base;
									b.altsupp1 = check_present;
								} else if (!b.altsupp2) {
									if (check_present < b.altsupp1) {
										b.altbase2 = sub_base;
										b.altsupp2 = check_present;
									} else {
										b.altbase2 = b.altbase1;
										b.altsupp2 = b.altsupp1;
										b.altbase1 = sub_base;
										b.altsupp1 = check_present;
									}
								} else if (!b.altsupp3) {
									if (check_present < b.altsupp2) {
										b.altbase3 = sub_base;
										b.altsupp3 = check_present;
									} else if (check_present < b.altsupp1) {
										b.altbase3 = b.altbase2;
										b.altsupp3 = b.altsupp2;
										b.altbase2 = sub_base;
										b.altsupp2 = check_present;
									} else {
										b.altbase3 = b.altbase2;
										b.altsupp3 = b.altsupp2;
										b.altbase2 = b.altbase1;
										b.altsupp2 = b.altsupp1;
										b.altbase1 = sub_base;
										b.altsupp1 = check_present;
									}
								}
							}
							if (p->mode == 0 || p->mode == 1) {
								continue;
							}
						}
						if (p->mode == 2 || b.best_edit_type != 1) {
							if (tryIndels(
							        c,
							        draft_char,
							        sub_base,
							        &num_deletions,
							        h_seq_i,
							        t_seq_i,
							        h_node_index,
							        t_node_index,
							        fhVal,
							        rhVal,
							        &b)) {
								if (p->mode == 0 || p->mode == 1) {
									break;
								}
							}
						}
					}
				}

				makeEdit(
				    c, draft_char, &b, &h_seq_i, &t_seq_i, &h_node_index, &t_node_index, &fhVal, &rhVal);
			}
		}
		/* roll and skip over k-mers containing non-accepted bases */
		int target_t_seq_i = -1;
		do {
			if (roll(
			        &h_seq_i,
			        &t_seq_i,
			        &h_node_index,
			        &t_node_index,
			        seq,
			        len,
			        &c->nv,
			        &charOut,
			        &charIn)) {
				if (!isAcceptedBase((unsigned char)toupper(charIn))) {
					target_t_seq_i = (int)t_seq_i + (int)p->k;
				}
				NTMC64_roll(charOut, charIn, p->k, p->h, &fhVal, &rhVal, c->hVal);
			} else {
				continue_edit = 0;
				break;
			}
		} while (target_t_seq_i >= 0 && (int)t_seq_i != target_t_seq_i);
	} while (continue_edit);

	writeEditsToFile(fa, tsv, hdr, c);
	free(c->nv.v);
	free(c->subs.v);
}

/* ----------------------------------------------------------- step-1 screen */
void
ora_screen(const char* seq, size_t len, const ora_bf* bloom, uint64_t* bitmap)
{
	unsigned k = bloom->k;
	size_t nwords = (len + 63) / 64;
	memset(bitmap, 0, nwords * sizeof(uint64_t));
	uint64_t hv[64];
	uint64_t fh = 0, rh = 0;
	size_t run = 0;
	for (size_t i = 0; i < len; i++) {
		if (!isAcceptedBase((unsigned char)toupper((unsigned char)seq[i]))) {
			run = 0;
			continue;
		}
		run++;
		if (run == k) {
			fh = ora_base_forward_hash(seq + i + 1 - k, k);
			rh = ora_base_reverse_hash(seq + i + 1 - k, k);
		} else if (run > k) {
			fh = ora_next_forward_hash(fh, k, (unsigned char)seq[i - k], (unsigned char)seq[i]);
			rh = ora_next_reverse_hash(rh, k, (unsigned char)seq[i - k], (unsigned char)seq[i]);
		} else {
			continue;
		}
		ora_extend_hashes(fh + rh, k, bloom->hash_num, hv);
		if (!(ora_bf_contains(bloom, hv) > 0)) {
			size_t s = i + 1 - k;
			bitmap[s >> 6] |= 1ULL << (s & 63);
		}
	}
}

/* ------------------------------------------------------------ FASTA reader */
/* kseq semantics (lib/kseq.h:176-215): name = header up to first whitespace,
 * comment = rest of the header line, sequence = all following lines
 * concatenated until a line starting with '>' / '@' / '+'. */
typedef struct
{
	gzFile f;
	char* line;
	size_t cap;
	int have_hdr;
	char* hdr_line;
} fa_reader;

static int
gz_getline(gzFile f, char** buf, size_t* cap)
{
	size_t n = 0;
	for (;;) {
		if (*cap - n < 2) {
			*cap = *cap ? *cap * 2 : 1 << 16;
			*buf = (char*)realloc(*buf, *cap);
		}
		if (!gzgets(f, *buf + n, (int)(*cap - n > 0x40000000 ? 0x40000000 : *cap - n))) {
			return n ? (int)1 : 0;
		}
		n += strlen(*buf + n);
		if (n && (*buf)[n - 1] == '\n') {
			(*buf)[--n] = 0;
			if (n && (*buf)[n - 1] == '\r') {
				(*buf)[--n] = 0;
			}
			return 1;
		}
		if (gzeof(f)) {
			return 1;
		}
	}
}

void
ora_write_vcf_header(FILE* vcf, const char* draft_path)
{
	/* ntedit.cpp:2192-2211 */
	time_t now = time(NULL);
	struct tm* ltm = localtime(&now);
	fprintf(vcf, "##fileformat=VCFv4.2\n##fileDate=%04d%02d%02d\n##source=ntEdit v2.1.1\n##reference=file:%s\n",
	        1900 + ltm->tm_year, 1 + ltm->tm_mon, ltm->tm_mday, draft_path);
	fprintf(vcf, "##FORMAT=<ID=GT,Number=1,Type=String,Description=\"Genotype\">\n");
	fprintf(vcf, "##INFO=<ID=AD,Number=2,Type=Integer,Description=\"Kmer Depth\">\n");
	fprintf(vcf, "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tINTEGRATION\n");
}

int
ora_polish_file(
    const char* draft_path,
    const ora_params* p,
    const ora_bf* bloom,
    const ora_bf* bloomrep,
    const char* prefix,
    uint64_t* bases_out)
{
	return ora_polish_file_vcf(draft_path, p, bloom, bloomrep, prefix, bases_out, NULL);
}

int
ora_polish_file_vcf(
    const char* draft_path,
    const ora_params* p,
    const ora_bf* bloom,
    const ora_bf* bloomrep,
    const char* prefix,
    uint64_t* bases_out,
    const ora_annot* annot)
{
	gzFile f = gzopen(draft_path, "r");
	if (!f) {
		return -1;
	}
	char path[4096];
	snprintf(path, sizeof path, "%s_edited.fa", prefix);
	FILE* fa = fopen(path, "w");
	snprintf(path, sizeof path, "%s_changes.tsv", prefix);
	FILE* tsv = fopen(path, "w");
	snprintf(path, sizeof path, "%s_variants.vcf", prefix);
	FILE* vcf = fopen(path, "w");
	if (!fa || !tsv || !vcf) {
		return -2;
	}
	ora_write_tsv_header(tsv, p, bloom);
	ora_write_vcf_header(vcf, draft_path);

	char* line = NULL;
	size_t cap = 0;
	char* hdr = NULL;
	char* seq = NULL;
	size_t seq_len = 0, seq_cap = 0;
	int in_rec = 0, fastq_skip = 0;
	uint64_t bases = 0;
	for (;;) {
		int got = gz_getline(f, &line, &cap);
		int is_hdr = got && (line[0] == '>' || line[0] == '@') && !fastq_skip;
		if (!got || is_hdr) {
			if (in_rec) {
				/* emit previous record: name + " " + comment (ntedit.cpp:2224-2229) */
				if (seq_len >= p->min_contig_len) {
					if (!seq) {
						seq = (char*)calloc(1, 1);
					}
					seq[seq_len] = 0;
					ora_polish_contig_vcf(hdr, seq, (unsigned)seq_len, p, bloom, bloomrep, fa, tsv, vcf, annot);
					bases += seq_len;
				}
			}
			if (!got) {
				break;
			}
			/* parse header: name up to first whitespace; one separator skipped */
			free(hdr);
			char* s = line + 1;
			size_t nl = 0;
			while (s[nl] && !isspace((unsigned char)s[nl])) {
				nl++;
			}
			size_t tl = strlen(s);
			hdr = (char*)malloc(tl + 2);
			memcpy(hdr, s, nl);
			hdr[nl] = 0;
			if (s[nl] && s[nl + 1]) {
				/* comment present and non-empty */
				hdr[nl] = ' ';
				strcpy(hdr + nl + 1, s + nl + 1);
			}
			in_rec = 1;
			seq_len = 0;
			continue;
		}
		if (!in_rec) {
			continue;
		}
		if (fastq_skip) {
			/* quality lines: consume until we have seq_len characters */
			size_t l = strlen(line);
			if (l >= (size_t)fastq_skip) {
				fastq_skip = 0;
			} else {
				fastq_skip -= (int)l;
			}
			continue;
		}
		if (line[0] == '+') {
			fastq_skip = (int)seq_len;
			if (!fastq_skip) {
				fastq_skip = 0;
			}
			continue;
		}
		size_t l = strlen(line);
		if (seq_len + l + 1 > seq_cap) {
			seq_cap = (seq_len + l + 1) * 2;
			seq = (char*)realloc(seq, seq_cap);
		}
		memcpy(seq + seq_len, line, l);
		seq_len += l;
	}
	free(line);
	free(hdr);
	free(seq);
	gzclose(f);
	fclose(fa);
	fclose(tsv);
	fclose(vcf);
	if (bases_out) {
		*bases_out = bases;
	}
	return 0;
}

/* counting-filter variant of the screen: absent <=> min counter is 0 or below min_threshold
 * (ntedit.cpp:1806-1807) */
void
ora_screen_counting_flat(
    const char* seq,
    size_t len,
    const uint8_t* bf_data,
    uint64_t bf_bytes,
    unsigned hash_num,
    unsigned k,
    unsigned min_threshold,
    uint64_t* bitmap)
{
	ora_bf bf;
	memset(&bf, 0, sizeof bf);
	bf.data = (uint8_t*)bf_data;
	bf.bytes = bf_bytes;
	bf.bits = bf_bytes * 8;
	bf.hash_num = hash_num;
	bf.k = k;
	bf.counting = 1;
	size_t nwords = (len + 63) / 64;
	memset(bitmap, 0, nwords * sizeof(uint64_t));
	uint64_t hv[64];
	uint64_t fh = 0, rh = 0;
	size_t run = 0;
	for (size_t i = 0; i < len; i++) {
		if (!isAcceptedBase((unsigned char)toupper((unsigned char)seq[i]))) {
			run = 0;
			continue;
		}
		run++;
		if (run == k) {
			fh = ora_base_forward_hash(seq + i + 1 - k, k);
			rh = ora_base_reverse_hash(seq + i + 1 - k, k);
		} else if (run > k) {
			fh = ora_next_forward_hash(fh, k, (unsigned char)seq[i - k], (unsigned char)seq[i]);
			rh = ora_next_reverse_hash(rh, k, (unsigned char)seq[i - k], (unsigned char)seq[i]);
		} else {
			continue;
		}
		ora_extend_hashes(fh + rh, k, hash_num, hv);
		unsigned c = ora_bf_contains(&bf, hv);
		if (c == 0 || c < min_threshold) {
			size_t s = i + 1 - k;
			bitmap[s >> 6] |= 1ULL << (s & 63);
		}
	}
}

/* flat-argument wrapper for ctypes callers (tests) */
void
ora_screen_flat(
    const char* seq,
    size_t len,
    const uint8_t* bf_data,
    uint64_t bf_bytes,
    unsigned hash_num,
    unsigned k,
    uint64_t* bitmap)
{
	ora_bf bf;
	memset(&bf, 0, sizeof bf);
	bf.data = (uint8_t*)bf_data;
	bf.bytes = bf_bytes;
	bf.bits = bf_bytes * 8;
	bf.hash_num = hash_num;
	bf.k = k;
	ora_screen(seq, len, &bf, bitmap);
}

/* flat-argument batch driver for ctypes callers (tests, bench cpu_baseline):
 * polishes the contigs of a packed batch (include/ntedit_hip.h layout) at -t 1.
 * fa_path / tsv_path may be NULL. Returns bases processed. */
uint64_t
ora_polish_batch_flat(
    const char* bases,
    const uint64_t* offsets,
    const uint32_t* lens,
    const char* const* names,
    uint32_t n_contigs,
    const uint8_t* bf_data,
    uint64_t bf_bytes,
    unsigned hash_num,
    unsigned k,
    const uint8_t* rep_data,
    uint64_t rep_bytes,
    unsigned rep_hash_num,
    const ora_params* params,
    const char* fa_path,
    const char* tsv_path)
{
	ora_bf bf, rep;
	memset(&bf, 0, sizeof bf);
	memset(&rep, 0, sizeof rep);
	bf.data = (uint8_t*)bf_data;
	bf.bytes = bf_bytes;
	bf.bits = bf_bytes * 8;
	bf.hash_num = hash_num;
	bf.k = k;
	if (rep_data) {
		rep.data = (uint8_t*)rep_data;
		rep.bytes = rep_bytes;
		rep.bits = rep_bytes * 8;
		rep.hash_num = rep_hash_num;
		rep.k = k;
	}
	ora_params p = *params;
	p.secbf = rep_data != NULL;
	ora_params_finalize(&p, &bf);
	FILE* fa = fa_path ? fopen(fa_path, "w") : NULL;
	FILE* tsv = tsv_path ? fopen(tsv_path, "w") : NULL;
	if (tsv) {
		ora_write_tsv_header(tsv, &p, &bf);
	}
	uint64_t total = 0;
	for (uint32_t i = 0; i < n_contigs; i++) {
		char* seq = (char*)malloc((size_t)lens[i] + 1);
		memcpy(seq, bases + offsets[i], lens[i]);
		seq[lens[i]] = 0;
		ora_polish_contig(names ? names[i] : "c", seq, lens[i], &p, &bf, rep_data ? &rep : NULL, fa, tsv);
		total += lens[i];
		free(seq);
	}
	if (fa) {
		fclose(fa);
	}
	if (tsv) {
		fclose(tsv);
	}
	return total;
}


/* ------------------------------------------------------------------ multi-threaded driver
 * The reference parallelises across contigs only (OpenMP loop, ntedit.cpp:2213-2253: every thread
 * takes the next record under a critical section and polishes it).  Same scheme with pthreads.
 * Two users: the cpu_baseline leg of bench.py (nothing written, contigs handed out in input order like
 * the reference) and the full-size parity tests (ora_polish_batch_flat_mt_files: every contig's
 * _edited.fa record / _changes.tsv rows / _variants.vcf rows are kept in memory and written in INPUT
 * order once all threads are done, i.e. the files of the reference at -t 1). */
#include <pthread.h>

typedef struct
{
	char* p;
	size_t n;
} mt_buf;

typedef struct
{
	const char* bases;
	const uint64_t* offsets;
	const uint32_t* lens;
	const char* const* names;
	uint32_t n_contigs;
	const ora_params* p;
	const ora_bf* bf;
	const ora_bf* rep;
	const uint32_t* order; /* hand-out order (NULL = input order) */
	mt_buf* fa;            /* per contig, NULL = discard */
	mt_buf* tsv;
	mt_buf* vcf;
	uint32_t next; /* next contig to hand out */
	uint64_t total;
	pthread_mutex_t mu;
} mt_job;

static void*
mt_worker(void* arg)
{
	mt_job* j = (mt_job*)arg;
	for (;;) {
		pthread_mutex_lock(&j->mu);
		uint32_t i = j->next < j->n_contigs ? j->next++ : UINT32_MAX;
		pthread_mutex_unlock(&j->mu);
		if (i == UINT32_MAX) {
			return NULL;
		}
		if (j->order) {
			i = j->order[i];
		}
		char* seq = (char*)malloc((size_t)j->lens[i] + 1);
		memcpy(seq, j->bases + j->offsets[i], j->lens[i]);
		seq[j->lens[i]] = 0;
		FILE* fa = j->fa ? open_memstream(&j->fa[i].p, &j->fa[i].n) : NULL;
		FILE* tsv = j->tsv ? open_memstream(&j->tsv[i].p, &j->tsv[i].n) : NULL;
		FILE* vcf = j->vcf ? open_memstream(&j->vcf[i].p, &j->vcf[i].n) : NULL;
		ora_polish_contig_vcf(j->names ? j->names[i] : "c", seq, j->lens[i], j->p, j->bf, j->rep, fa, tsv, vcf, NULL);
		if (fa) {
			fclose(fa);
		}
		if (tsv) {
			fclose(tsv);
		}
		if (vcf) {
			fclose(vcf);
		}
		free(seq);
		pthread_mutex_lock(&j->mu);
		j->total += j->lens[i];
		pthread_mutex_unlock(&j->mu);
	}
}

static void
mt_run(mt_job* j, unsigned n_threads)
{
	pthread_mutex_init(&j->mu, NULL);
	if (n_threads < 1) {
		n_threads = 1;
	}
	pthread_t* th = (pthread_t*)calloc(n_threads, sizeof(pthread_t));
	for (unsigned t = 0; t < n_threads; t++) {
		pthread_create(&th[t], NULL, mt_worker, j);
	}
	for (unsigned t = 0; t < n_threads; t++) {
		pthread_join(th[t], NULL);
	}
	free(th);
	pthread_mutex_destroy(&j->mu);
}

/* the in-memory primary filter of the ora_*_flat* entry points is a KmerCountingBloomFilter8 (ntedit.cpp:357-361):
 * set before the call, process-wide (test infrastructure) */
static int flat_primary_counting = 0;

void
ora_set_flat_counting(int counting)
{
	flat_primary_counting = counting;
}

static void
flat_bf(ora_bf* bf, const uint8_t* data, uint64_t bytes, unsigned hash_num, unsigned k)
{
	memset(bf, 0, sizeof *bf);
	bf->data = (uint8_t*)data;
	bf->bytes = bytes;
	bf->bits = bytes * 8;
	bf->hash_num = hash_num;
	bf->k = k;
}

uint64_t
ora_polish_batch_flat_mt(
    const char* bases,
    const uint64_t* offsets,
    const uint32_t* lens,
    uint32_t n_contigs,
    const uint8_t* bf_data,
    uint64_t bf_bytes,
    unsigned hash_num,
    unsigned k,
    const ora_params* params,
    unsigned n_threads)
{
	ora_bf bf;
	flat_bf(&bf, bf_data, bf_bytes, hash_num, k);
	ora_params p = *params;
	p.secbf = 0;
	ora_params_finalize(&p, &bf);
	mt_job j;
	memset(&j, 0, sizeof j);
	j.bases = bases;
	j.offsets = offsets;
	j.lens = lens;
	j.n_contigs = n_contigs;
	j.p = &p;
	j.bf = &bf;
	mt_run(&j, n_threads);
	return j.total;
}

static const uint32_t* mt_sort_lens;

static int
mt_by_len_desc(const void* a, const void* b)
{
	const uint32_t x = *(const uint32_t*)a, y = *(const uint32_t*)b;
	if (mt_sort_lens[x] != mt_sort_lens[y]) {
		return mt_sort_lens[x] > mt_sort_lens[y] ? -1 : 1;
	}
	return x < y ? -1 : (x > y);
}

uint64_t
ora_polish_batch_flat_mt_files(
    const char* bases,
    const uint64_t* offsets,
    const uint32_t* lens,
    const char* const* names,
    uint32_t n_contigs,
    const uint8_t* bf_data,
    uint64_t bf_bytes,
    unsigned hash_num,
    unsigned k,
    const uint8_t* rep_data,
    uint64_t rep_bytes,
    unsigned rep_hash_num,
    const ora_params* params,
    unsigned n_threads,
    const char* fa_path,
    const char* tsv_path,
    const char* vcf_path)
{
	ora_bf bf, rep;
	flat_bf(&bf, bf_data, bf_bytes, hash_num, k);
	bf.counting = flat_primary_counting;
	flat_bf(&rep, rep_data, rep_bytes, rep_hash_num, k);
	ora_params p = *params;
	p.secbf = rep_data != NULL;
	ora_params_finalize(&p, &bf);
	mt_job j;
	memset(&j, 0, sizeof j);
	j.bases = bases;
	j.offsets = offsets;
	j.lens = lens;
	j.names = names;
	j.n_contigs = n_contigs;
	j.p = &p;
	j.bf = &bf;
	j.rep = rep_data ? &rep : NULL;
	/* longest contig first: the wall time of a checker run should not hang on a 50 Mbp contig
	 * that happened to come last (the OUTPUT order is the input order either way) */
	uint32_t* order = (uint32_t*)malloc(((size_t)n_contigs + 1) * sizeof(uint32_t));
	for (uint32_t i = 0; i < n_contigs; i++) {
		order[i] = i;
	}
	mt_sort_lens = lens;
	qsort(order, n_contigs, sizeof(uint32_t), mt_by_len_desc);
	j.order = order;
	j.fa = fa_path ? (mt_buf*)calloc((size_t)n_contigs + 1, sizeof(mt_buf)) : NULL;
	j.tsv = tsv_path ? (mt_buf*)calloc((size_t)n_contigs + 1, sizeof(mt_buf)) : NULL;
	j.vcf = vcf_path ? (mt_buf*)calloc((size_t)n_contigs + 1, sizeof(mt_buf)) : NULL;
	mt_run(&j, n_threads);
	free(order);
	const char* paths[3] = { fa_path, tsv_path, vcf_path };
	mt_buf* bufs[3] = { j.fa, j.tsv, j.vcf };
	for (int s = 0; s < 3; s++) {
		if (!paths[s]) {
			continue;
		}
		FILE* f = fopen(paths[s], "w");
		if (f) {
			if (s == 1) {
				ora_write_tsv_header(f, &p, &bf);
			}
			for (uint32_t i = 0; i < n_contigs; i++) {
				if (bufs[s][i].n) {
					fwrite(bufs[s][i].p, 1, bufs[s][i].n, f);
				}
			}
			fclose(f);
		}
		for (uint32_t i = 0; i < n_contigs; i++) {
			free(bufs[s][i].p);
		}
		free(bufs[s]);
	}
	return j.total;
}

/* ---- the tables of ntedit.cpp:172-348 as this restatement holds them, in the canonical text form of
 * tests/tools/reference_tables.py ("num_tries ...", "polish X cands", "snv X cands", "multi X cand cand ..."): the CPU tier
 * compares it with the text extracted from the reference's source, where that is present.  Returns the length written
 * (without the terminating 0), or -1 when cap is too small. */
long
ora_tables_dump(char* out, size_t cap)
{
	size_t n = 0;
#define ORA_PUT(...)                                                   \
	do {                                                               \
		int w_ = snprintf(out + n, n < cap ? cap - n : 0, __VA_ARGS__); \
		if (w_ < 0 || n + (size_t)w_ >= cap) {                         \
			return -1;                                                 \
		}                                                              \
		n += (size_t)w_;                                               \
	} while (0)
	ORA_PUT("num_tries");
	for (int i = 0; i < 6; i++) {
		ORA_PUT(" %u", num_tries[i]);
	}
	ORA_PUT("\n");
	static const char letters[] = "ATCGRYSWKMBDHVN";
	for (int snv = 0; snv < 2; snv++) {
		for (const char* c = letters; *c; c++) {
			unsigned char cand[8];
			unsigned nc = candidate_bases(snv, (unsigned char)*c, cand);
			ORA_PUT("%s %c ", snv ? "snv" : "polish", *c);
			for (unsigned q = 0; q < nc; q++) {
				ORA_PUT("%c", cand[q]);
			}
			ORA_PUT("\n");
		}
	}
	for (const char* c = "ACGT"; *c; c++) {
		ORA_PUT("multi %c", *c);
		for (unsigned i = 0; i < num_tries[5]; i++) {
			char ins[8];
			insertion_candidate((unsigned char)*c, i, ins);
			ORA_PUT(" %s", ins);
		}
		ORA_PUT("\n");
	}
#undef ORA_PUT
	return (long)n;
}
