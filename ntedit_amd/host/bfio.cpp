#include "bfio.h"

#include <cstdlib>
#include <cstring>

namespace nte_host {

FILE*
bf_open(const char* path, BfHeader* h)
{
	FILE* f = fopen(path, "rb");
	if (!f) {
		return nullptr;
	}
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
		}
	}
	if (!ended || h->bytes == 0 || h->hash_num == 0) {
		fclose(f);
		return nullptr;
	}
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
