// dump_btllib.cpp -- the REAL btllib's answers for the verification kit (tests/tools/verify_btllib.sh).
// Needs a btllib installation (e.g. `conda install -c bioconda btllib`); it cannot be built in the development
// image, which is exactly why hashing + the .bf format are "parity unpinned" (oracle header).
// It makes the very calls ntedit.cpp makes (ntedit.cpp:24-26, 357-362, 368-376, 412-415, 428-431, 444-451):
//   hashes <seq.txt> <k> <h>          one line per k-mer start of every line of seq.txt: seeded and rolled
//                                     forward / reverse hashes, the h extended hashes, and the change-last update
//   build <genome.fa> <k> <h> <bytes> <out.bf>     KmerBloomFilter built with btllib, saved with btllib
//   query <filter.bf> <seq.txt>       contains() of every k-mer of seq.txt against a filter file (ours or btllib's)
#include <btllib/bloom_filter.hpp>
#include <btllib/counting_bloom_filter.hpp>
#include <btllib/nthash.hpp>
#include <btllib/seq_reader.hpp>

#include <cinttypes>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>

namespace hi = btllib::hashing_internals;

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

int
main(int argc, char** argv)
{
	if (argc >= 5 && !strcmp(argv[1], "hashes")) {
		const unsigned k = (unsigned)atoi(argv[3]), h = (unsigned)atoi(argv[4]);
		std::vector<uint64_t> hv(h), hv2(h);
		for (const std::string& s : read_lines(argv[2])) {
			if (s.size() < k) {
				continue;
			}
			uint64_t fh = 0, rh = 0;
			for (size_t i = 0; i + k <= s.size(); i++) {
				// seeded (NTMC64(kmerSeq, ...), ntedit.cpp:403-416)
				const uint64_t sfh = hi::base_forward_hash(s.data() + i, k);
				const uint64_t srh = hi::base_reverse_hash(s.data() + i, k);
				if (i == 0) {
					fh = sfh;
					rh = srh;
				} else {
					// rolled (NTMC64(charOut, charIn, ...), ntedit.cpp:418-432)
					fh = hi::next_forward_hash(fh, k, (unsigned char)s[i - 1], (unsigned char)s[i + k - 1]);
					rh = hi::next_reverse_hash(rh, k, (unsigned char)s[i - 1], (unsigned char)s[i + k - 1]);
				}
				hi::extend_hashes(hi::canonical(sfh, srh), k, h, hv.data());
				// change of the last base to 'A' (NTMC64_changelast, ntedit.cpp:434-452)
				uint64_t cf = sfh, cr = srh;
				const unsigned char out = (unsigned char)s[i + k - 1], in = 'A';
				cf ^= hi::SEED_TAB[out];
				cf ^= hi::SEED_TAB[in];
				cr ^= hi::srol_table(out & hi::CP_OFF, k - 1);
				cr ^= hi::srol_table(in & hi::CP_OFF, k - 1);
				hi::extend_hashes(hi::canonical(cf, cr), k, h, hv2.data());
				printf("%zu seed %016" PRIx64 " %016" PRIx64 " roll %016" PRIx64 " %016" PRIx64 " h", i, sfh, srh, fh, rh);
				for (unsigned q = 0; q < h; q++) {
					printf(" %016" PRIx64, hv[q]);
				}
				printf(" lastA");
				for (unsigned q = 0; q < h; q++) {
					printf(" %016" PRIx64, hv2[q]);
				}
				printf("\n");
			}
			printf("--\n");
		}
		return 0;
	}
	if (argc >= 7 && !strcmp(argv[1], "build")) {
		const unsigned k = (unsigned)atoi(argv[3]), h = (unsigned)atoi(argv[4]);
		const size_t bytes = (size_t)strtoull(argv[5], nullptr, 10);
		btllib::KmerBloomFilter bf(bytes, h, k);
		btllib::SeqReader reader(argv[2], btllib::SeqReader::Flag::LONG_MODE);
		for (const auto record : reader) {
			bf.insert(record.seq); // (what src/ntedit_make_genome_bf.cpp:143-157 does)
		}
		bf.save(argv[6]);
		printf("bytes %zu fpr %.9g\n", (size_t)bf.get_bytes(), bf.get_fpr());
		return 0;
	}
	if (argc >= 4 && !strcmp(argv[1], "query")) {
		const bool counting = btllib::BloomFilter::check_file_signature(argv[2], btllib::KMER_COUNTING_BLOOM_FILTER_SIGNATURE);
		if (counting) {
			btllib::KmerCountingBloomFilter8 cbf(argv[2]);
			const unsigned k = cbf.get_k(), h = cbf.get_hash_num();
			std::vector<uint64_t> hv(h);
			for (const std::string& s : read_lines(argv[3])) {
				for (size_t i = 0; i + k <= s.size(); i++) {
					hi::extend_hashes(hi::canonical(hi::base_forward_hash(s.data() + i, k), hi::base_reverse_hash(s.data() + i, k)), k, h, hv.data());
					printf("%u\n", (unsigned)cbf.contains(hv.data()));
				}
				printf("--\n");
			}
			return 0;
		}
		btllib::KmerBloomFilter bf(argv[2]);
		const unsigned k = bf.get_k(), h = bf.get_hash_num();
		std::vector<uint64_t> hv(h);
		printf("k %u h %u bytes %zu\n", k, h, (size_t)bf.get_bytes());
		for (const std::string& s : read_lines(argv[3])) {
			for (size_t i = 0; i + k <= s.size(); i++) {
				hi::extend_hashes(hi::canonical(hi::base_forward_hash(s.data() + i, k), hi::base_reverse_hash(s.data() + i, k)), k, h, hv.data());
				printf("%d\n", bf.contains(hv.data()) ? 1 : 0);
			}
			printf("--\n");
		}
		return 0;
	}
	fprintf(stderr, "usage: dump_btllib hashes|build|query ... (see the header comment)\n");
	return 2;
}
