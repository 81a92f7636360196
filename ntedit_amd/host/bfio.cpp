#include "bfio.h"

#include <cstdlib>
#include <cstring>

namespace nte_host {

FILE*
bf_open(const char* path, BfHeader* h, const char** why)
{
	const char* dummy;
	if (!why) {
		why = &dummy;
	}
	*why = "cannot open the file";
	FILE* f = fopen(path, "rb");
	if (!f) {
		return nullptr;
	}
	*why = "not a btllib Bloom filter file (no [BTL...BloomFilter...] signature line)";
	char line[1024];
	bool first = true, ended = false;
	*h = BfHeader();
	while (fgets(line, sizeof line, f)) {
		if (first) {
			first = false;
			if (strncmp(line, "[BTL", 4) != 0 || !strstr(line, "BloomFilter")) {
				break;
			}
			h->counting = strstr(line, "Counting") != nullptr;
			continue;
		}
		if (strncmp(line, "[HeaderEnd]", 11) == 0) {
			ended = true;
			break;
		}
		char* eq = strchr(line, '=');
		if (!eq) {
			continue;
		}
		*eq = 0;
		char* key = line;
		while (*key == ' ' || *key == '\t') {
			key++;
		}
		char* ke = eq;
		while (ke > key && (ke[-1] == ' ' || ke[-1] == '\t')) {
			*--ke = 0;
		}
		const char* val = eq + 1;
		if (!strcmp(key, "bytes")) {
			h->bytes = strtoull(val, nullptr, 10);
		} else if (!strcmp(key, "hash_num")) {
			h->hash_num = (uint32_t)strtoul(val, nullptr, 10);
		} else if (!strcmp(key, "k")) {
			h->k = (uint32_t)strtoul(val, nullptr, 10);
		} else if (!strcmp(key, "hash_fn")) {
			const char* q = strchr(val, '"');
			size_t n = 0;
			if (q) {
				for (q++; *q && *q != '"' && n + 1 < sizeof h->hash_fn; q++) {
					h->hash_fn[n++] = *q;
				}
			}
			h->hash_fn[n] = 0;
		}
	}
	const char* bad = nullptr;
	if (first) {
		bad = "empty file";
	} else if (!ended) {
		bad = *why; // (signature missing) or:
		if (h->bytes || h->hash_num || h->k) {
			bad = "header without [HeaderEnd]";
		}
	} else if (h->bytes == 0) {
		bad = "header has no `bytes`";
	} else if (h->hash_num == 0) {
		bad = "header has no `hash_num`";
	} else if (h->k == 0) {
		bad = "header has no `k` (a plain btllib BloomFilter, not a k-mer Bloom filter)";
	} else if (h->hash_fn[0] && strcmp(h->hash_fn, "ntHash_v2") != 0) {
		bad = "hash_fn is not \"ntHash_v2\" (filter written with another hash function)";
	}
	if (bad) {
		*why = bad;
		fclose(f);
		return nullptr;
	}
	*why = nullptr;
	return f;
}

int
bf_save(const char* path, const BfHeader& h, const uint8_t* data)
{
	FILE* f = fopen(path, "wb");
	if (!f) {
		return -1;
	}
	fprintf(
	    f,
	    "[%s]\nbytes = %llu\nhash_fn = \"ntHash_v2\"\nhash_num = %u\nk = %u\n[HeaderEnd]\n",
	    h.counting ? "BTLKmerCountingBloomFilter_v5" : "BTLKmerBloomFilter_v6",
	    (unsigned long long)h.bytes,
	    h.hash_num,
	    h.k);
	size_t w = fwrite(data, 1, h.bytes, f);
	int rc = fclose(f);
	return (w == h.bytes && rc == 0) ? 0 : -1;
}

} // namespace nte_host
