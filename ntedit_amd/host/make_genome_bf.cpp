// make_genome_bf.cpp -- `ntedit-make-genome-bf` for the MI355X hot path: builds the k-mer Bloom
// filter of one or more genome FASTA files in HBM and saves it in btllib's file format.
//
// Keeps the reference tool's command line and console output (src/ntedit_make_genome_bf.cpp:49-166):
//   --genome FILE [FILE ...]  -k K  [--fpr 0.01] [--hashes 3] [-o genome_bf.bf]
//   [--bf BYTES] [--num_elements N] [-t THREADS]
// Sizing follows get_bf_size (ntedit_make_genome_bf.cpp:41-47); every all-ACGT k-mer of every
// record at least k long is inserted (143-157) by the same rolling-hash kernel that screens a draft
// (ntedit_hip_filter_insert).  -t is accepted; the k-mers are hashed on the GPU.
#include "../../include/ntedit_hip.h"
#include "fasta.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <iostream>
#include <string>
#include <vector>

static void
log_info(const std::string& msg)
{
	// btllib::log_info: "[<local time>] [INFO] <msg>" on stderr
	char ts[64];
	time_t now = time(nullptr);
	strftime(ts, sizeof ts, "%Y-%m-%d %H:%M:%S", localtime(&now));
	std::cerr << "[" << ts << "] [INFO] " << msg << std::endl;
}

// ntedit_make_genome_bf.cpp:41-47 (Broder & Mitzenmacher 2004, via ntHits)
static uint64_t
get_bf_size(uint64_t num_elements, double num_hashes, double fpr)
{
	const double r = -num_hashes / log(1.0 - exp(log(fpr) / num_hashes));
	const uint64_t m = (uint64_t)(ceil((double)num_elements * r) / 8u);
	return m;
}

static void
usage(const char* why)
{
	if (why) {
		std::cerr << why << std::endl;
	}
	std::cerr << "Usage: make_genome_bf [--help] --genome VAR... -k VAR [--fpr VAR] [--hashes VAR] [-o VAR] [--bf VAR] "
	             "[--num_elements VAR] [-t VAR]\n\n"
	             "Optional arguments:\n"
	             "  -h, --help      shows help message and exits\n"
	             "  --genome        Input genome fasta file [nargs: 1 or more] [required]\n"
	             "  -k              k-mer size (bp) [required]\n"
	             "  --fpr           False positive rate for Bloom filter [default: 0.01]\n"
	             "  --hashes        Number of hash functions [default: 3]\n"
	             "  -o              Name for output Bloom filter [default: \"genome_bf.bf\"]\n"
	             "  --bf            Bloom filter size in bytes (optional)\n"
	             "  --num_elements  Approximate number of elements for Bloom filter (used for calculating Bloom filter "
	             "size, optional)\n"
	             "  -t              Number of threads [default: 12]\n";
}

static bool
is_option(const char* a)
{
	return a[0] == '-' && a[1] != 0 && !(a[1] >= '0' && a[1] <= '9');
}

int
main(int argc, char** argv)
{
	std::vector<std::string> genome_files;
	unsigned k = 0, hashes = 3, num_threads = 12;
	bool have_k = false, have_bf = false, have_ne = false;
	double fpr = 0.01;
	uint64_t bf_bytes = 0, num_elements = 0;
	std::string out_file = "genome_bf.bf";
	for (int i = 1; i < argc; i++) {
		const std::string a = argv[i];
		auto value = [&](const char* name) -> const char* {
			if (i + 1 >= argc) {
				usage((std::string("Too few arguments for '") + name + "'.").c_str());
				exit(1);
			}
			return argv[++i];
		};
		if (a == "-h" || a == "--help") {
			usage(nullptr);
			return 0;
		} else if (a == "--genome") {
			while (i + 1 < argc && !is_option(argv[i + 1])) {
				genome_files.push_back(argv[++i]);
			}
		} else if (a == "-k") {
			k = (unsigned)strtoul(value("-k"), nullptr, 10);
			have_k = true;
		} else if (a == "--fpr") {
			fpr = strtod(value("--fpr"), nullptr);
		} else if (a == "--hashes") {
			hashes = (unsigned)strtoul(value("--hashes"), nullptr, 10);
		} else if (a == "-o") {
			out_file = value("-o");
		} else if (a == "--bf") {
			bf_bytes = strtoull(value("--bf"), nullptr, 10);
			have_bf = true;
		} else if (a == "--num_elements") {
			num_elements = strtoull(value("--num_elements"), nullptr, 10);
			have_ne = true;
		} else if (a == "-t") {
			num_threads = (unsigned)strtoul(value("-t"), nullptr, 10);
		} else {
			usage(("Unknown argument: " + a).c_str());
			return 1;
		}
	}
	if (genome_files.empty()) {
		usage("--genome: 1 or more argument(s) expected. 0 provided.");
		return 1;
	}
	if (!have_k) {
		usage("-k: required.");
		return 1;
	}

	std::cout << "Parameters:" << std::endl;
	std::cout << "\t\t--genome ";
	for (const std::string& g : genome_files) {
		std::cout << g << " ";
	}
	std::cout << std::endl;
	std::cout << "\t\t-t " << num_threads << std::endl;
	std::cout << "\t\t-k " << k << std::endl;
	std::cout << "\t\t--fpr " << fpr << std::endl;
	std::cout << "\t\t--hashes " << hashes << std::endl;
	std::cout << "\t\t-o " << out_file << std::endl;

	uint64_t bf_size;
	if (have_bf) {
		bf_size = bf_bytes;
		std::cout << "\t\t--bf " << bf_size << std::endl;
	} else if (have_ne) {
		std::cout << "\t\t--num_elements " << num_elements << std::endl;
		bf_size = get_bf_size(num_elements, hashes, fpr);
	} else {
		std::cout << "Calculating BF size based on input genome size" << std::endl;
		uint64_t genome_size = 0;
		for (const std::string& g : genome_files) {
			nte_host::FastaReader reader(g.c_str());
			std::string hdr, seq;
			while (reader.ok() && reader.next(hdr, seq)) {
				genome_size += seq.size();
				seq.clear();
			}
		}
		std::cout << "Genome size (bp): " << genome_size << std::endl;
		bf_size = get_bf_size(genome_size, hashes, fpr);
	}
	std::cout << "BF size (bytes): " << bf_size << std::endl;

	ntedit_hip_ctx* ctx = nullptr;
	if (ntedit_hip_create(0, &ctx) != 0) {
		std::cerr << "make_genome_bf: error: " << (ctx ? ntedit_hip_last_error(ctx) : "no HIP device") << std::endl;
		return 1;
	}
	if (ntedit_hip_filter_alloc(ctx, NTEDIT_FILTER_PRIMARY, bf_size, hashes, k) != 0) {
		std::cerr << "make_genome_bf: error: " << ntedit_hip_last_error(ctx) << std::endl;
		return 1;
	}
	// records are concatenated, '\n' between them (no k-mer spans a separator), and handed to
	// the insert kernel a few hundred MB at a time
	const size_t FLUSH = 512u << 20;
	std::string blob, hdr;
	blob.reserve(FLUSH + (64 << 20));
	auto flush = [&]() {
		if (blob.empty()) {
			return;
		}
		if (ntedit_hip_filter_insert(ctx, NTEDIT_FILTER_PRIMARY, blob.data(), blob.size(), 0) != 0) {
			std::cerr << "make_genome_bf: error: " << ntedit_hip_last_error(ctx) << std::endl;
			exit(1);
		}
		blob.clear();
	};
	for (const std::string& g : genome_files) {
		log_info("Reading " + g);
		nte_host::FastaReader reader(g.c_str());
		if (!reader.ok()) {
			std::cerr << "make_genome_bf: error: cannot open " << g << std::endl;
			return 1;
		}
		for (;;) {
			const size_t before = blob.size();
			if (!reader.next(hdr, blob)) {
				break;
			}
			if (blob.size() - before >= k) { // ntedit_make_genome_bf.cpp:152
				blob.push_back('\n');
			} else {
				blob.resize(before);
			}
			if (blob.size() >= FLUSH) {
				flush();
			}
		}
	}
	flush();

	uint64_t occupied = 0, slots = 0;
	if (ntedit_hip_filter_occupancy(ctx, NTEDIT_FILTER_PRIMARY, &occupied, &slots) != 0) {
		std::cerr << "make_genome_bf: error: " << ntedit_hip_last_error(ctx) << std::endl;
		return 1;
	}
	// btllib BloomFilter::get_fpr(): occupancy ^ hash_num
	std::cout << "Bloom filter FPR: " << pow((double)occupied / (double)slots, (double)hashes) << std::endl;

	log_info("Saving Bloom filter");
	if (ntedit_hip_filter_save_file(ctx, NTEDIT_FILTER_PRIMARY, out_file.c_str()) != 0) {
		std::cerr << "make_genome_bf: error: cannot write " << out_file << std::endl;
		return 1;
	}
	log_info("Done!");
	ntedit_hip_destroy(ctx);
	return 0;
}
