// dump_ours.cpp -- this repository's answers for the verification kit (tests/tools/verify_btllib.sh), computed with
// the PRODUCT's primitives: ntedit_amd/csrc/nte_common.h (the arithmetic the HIP kernels run, compiled here for the
// host) and ntedit_amd/host/bfio.cpp (the .bf reader / writer).  Same commands and the same output, line for line, as
// dump_btllib.cpp; no GPU needed.
#include "../../../ntedit_amd/csrc/nte_common.h"
#include "../../../ntedit_amd/host/bfio.h"
#include "../../../ntedit_amd/host/params.h"

#include <cinttypes>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>

using namespace nte;

static std::vector<std::string>
read_lines(const char* path)
{
	std::vector<std::string> out;
	std::ifstream f(path);
	std::string l;
	while (std::getline(f, l)) {
		if (!l.empty()) {
			out.push_back(l);
		}
	}
	return out;
}

// base_forward_hash / base_reverse_hash as the product seeds an event (raw-byte seeds: defined for every byte)
static HashState
seed(const char* s, unsigned k)
{
	HashState h = { 0, 0 };
	for (unsigned i = 0; i < k; i++) {
		h.fh = srol1(h.fh) ^ seed_fwd_raw((u8)s[i]);
	}
	for (unsigned i = k; i > 0; i--) {
		h.rh = srol1(h.rh) ^ seed_rev_raw((u8)s[i - 1]);
	}
	return h;
}

static DevParams
params(unsigned k, unsigned h)
{
	ntedit_hip_params hp;
	nte_host::params_default(&hp);
	DevParams p;
	nte_host::make_dev_params(hp, k < 12 ? 12 : k, h, false, &p, false); // (thresholds are irrelevant here)
	p.k = k;
	for (unsigned i = 0; i < MAX_HASHES; i++) {
		p.mul[i] = (u64)i ^ ((u64)k * MULTISEED);
	}
	return p;
}

static std::vector<std::pair<std::string, std::string>>
read_fasta(const char* path)
{
	std::vector<std::pair<std::string, std::string>> out;
	std::ifstream f(path);
	std::string l;
	while (std::getline(f, l)) {
		if (!l.empty() && l[0] == '>') {
			out.push_back({ l.substr(1), "" });
		} else if (!out.empty()) {
			out.back().second += l;
		}
	}
	return out;
}

int
main(int argc, char** argv)
{
	if (argc >= 5 && !strcmp(argv[1], "hashes")) {
		const unsigned k = (unsigned)atoi(argv[3]), h = (unsigned)atoi(argv[4]);
		const DevParams p = params(k, h);
		for (const std::string& s : read_lines(argv[2])) {
			if (s.size() < k) {
				continue;
			}
			HashState r = { 0, 0 };
			for (size_t i = 0; i + k <= s.size(); i++) {
				const HashState sd = seed(s.data() + i, k);
				if (i == 0) {
					r = sd;
				} else {
					hash_roll_raw(r, k, (u8)s[i - 1], (u8)s[i + k - 1]);
				}
				HashState cl = sd;
				hash_changelast_raw(cl, k, (u8)s[i + k - 1], (u8)'A');
				printf("%zu seed %016" PRIx64 " %016" PRIx64 " roll %016" PRIx64 " %016" PRIx64 " h", i, sd.fh, sd.rh, r.fh, r.rh);
				for (unsigned q = 0; q < h; q++) {
					printf(" %016" PRIx64, hash_extend(sd.fh + sd.rh, p, q));
				}
				printf(" lastA");
				for (unsigned q = 0; q < h; q++) {
					printf(" %016" PRIx64, hash_extend(cl.fh + cl.rh, p, q));
				}
				printf("\n");
			}
			printf("--\n");
		}
		return 0;
	}
	if (argc >= 7 && !strcmp(argv[1], "build")) {
		const unsigned k = (unsigned)atoi(argv[3]), h = (unsigned)atoi(argv[4]);
		u64 bytes = strtoull(argv[5], nullptr, 10);
		bytes = (bytes + 7) / 8 * 8; // (btllib's constructor rounds up to whole 64-bit words)
		const DevParams p = params(k, h);
		std::vector<u8> data(bytes, 0);
		Filter f;
		f.data = data.data();
		filter_set_size(f, bytes * 8);
		f.hash_num = h;
		f.counting = 0;
		for (const auto& rec : read_fasta(argv[2])) {
			const std::string& s = rec.second;
			// every k-mer made of ACGT only (what the insert kernel k_screen<..., INSERT> takes)
			size_t good = 0;
			for (size_t i = 0; i < s.size(); i++) {
				good = char_code((u8)s[i]) <= 3 ? good + 1 : 0;
				if (good >= k) {
					const HashState sd = seed(s.data() + i + 1 - k, k);
					for (unsigned q = 0; q < h; q++) {
						const u64 n = filter_slot(f, hash_extend(sd.fh + sd.rh, p, q));
						data[n >> 3] |= (u8)(1u << (n & 7));
					}
				}
			}
		}
		nte_host::BfHeader hd;
		hd.bytes = bytes;
		hd.hash_num = h;
		hd.k = k;
		hd.counting = false;
		if (nte_host::bf_save(argv[6], hd, data.data())) {
			return 1;
		}
		u64 occ = 0;
		for (u8 b : data) {
			occ += (u64)__builtin_popcount(b);
		}
		double fpr = 1.0;
		for (unsigned q = 0; q < h; q++) {
			fpr *= (double)occ / (double)(bytes * 8);
		}
		printf("bytes %zu fpr %.9g\n", (size_t)bytes, fpr);
		return 0;
	}
	if (argc >= 4 && !strcmp(argv[1], "query")) {
		nte_host::BfHeader hd;
		const char* why = nullptr;
		FILE* fp = nte_host::bf_open(argv[2], &hd, &why);
		if (!fp) {
			fprintf(stderr, "%s: %s\n", argv[2], why ? why : "?");
			return 1;
		}
		std::vector<u8> data(hd.bytes);
		if (fread(data.data(), 1, hd.bytes, fp) != hd.bytes) {
			return 1;
		}
		fclose(fp);
		const DevParams p = params(hd.k, hd.hash_num);
		Filter f;
		f.data = data.data();
		filter_set_size(f, hd.counting ? hd.bytes : hd.bytes * 8);
		f.hash_num = hd.hash_num;
		f.counting = hd.counting ? 1 : 0;
		if (!hd.counting) {
			printf("k %u h %u bytes %zu\n", hd.k, hd.hash_num, (size_t)hd.bytes);
		}
		for (const std::string& s : read_lines(argv[3])) {
			for (size_t i = 0; i + hd.k <= s.size(); i++) {
				const HashState sd = seed(s.data() + i, hd.k);
				if (hd.counting) {
					printf("%u\n", filter_min_count(f, p, sd.fh + sd.rh));
				} else {
					printf("%d\n", filter_contains(f, p, sd) ? 1 : 0);
				}
			}
			printf("--\n");
		}
		return 0;
	}
	fprintf(stderr, "usage: dump_ours hashes|build|query ... (see dump_btllib.cpp)\n");
	return 2;
}
