// fasta_abi.cpp -- the host binary's FASTA ingest behind the C ABI (ntedit_hip_fasta_*), so that every driver of
// the hot path (the `ntedit` binary, python -m ntedit_amd.run, a reference maintainer's own main) reads a draft
// with ONE implementation of kseq's record semantics (lib/kseq.h:176-215 as used at ntedit.cpp:2223-2230):
// the mapped multi-threaded reader for plain and BGZF multi-FASTA, the streaming kseq restatement for everything
// else (ordinary gzip -- found by its magic bytes, not by the file name --, FASTQ, CR line ends, text in front of
// the first record).  A sequence that holds a NUL byte ends there (contigSeq = seq->seq.s, ntedit.cpp:2230).
#include "../../include/ntedit_hip.h"
#include "fasta.h"
#include "fasta_map.h"

#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

struct ntedit_hip_fasta
{
	std::vector<std::string> headers;
	std::vector<uint64_t> offs;
	std::vector<uint64_t> lens;
	std::string blob; // every sequence followed by '\n': the batch layout of ntedit_hip_polish_batch
};

static void
set_err(char* err, size_t cap, const std::string& text)
{
	if (err && cap) {
		snprintf(err, cap, "%s", text.c_str());
	}
}

extern "C" {

int
ntedit_hip_fasta_load(const char* path, uint64_t min_len, unsigned threads, ntedit_hip_fasta** out, char* err, size_t errcap)
{
	if (!path || !out) {
		set_err(err, errcap, "fasta_load: bad argument");
		return NTEDIT_E_ARG;
	}
	*out = nullptr;
	if (threads == 0) {
		threads = std::thread::hardware_concurrency();
		threads = threads > 16 ? 16 : (threads < 1 ? 1 : threads);
	}
	// (multi-GB containers are filled below: an exception must not cross the C ABI)
	ntedit_hip_fasta* f = nullptr;
	try {
	f = new ntedit_hip_fasta();
	nte_host::FastaMap fmap(path, threads, threads > 64 ? 64 : threads);
	if (fmap.ok()) {
		const size_t N = fmap.records();
		fmap.measure(0, N);
		std::vector<size_t> pick;
		uint64_t total = 0;
		for (size_t i = 0; i < N; i++) {
			if (fmap.length(i) >= min_len) {
				pick.push_back(i);
				f->offs.push_back(total);
				f->lens.push_back(fmap.length(i));
				f->headers.push_back(fmap.header(i));
				total += fmap.length(i) + 1;
			}
		}
		f->blob.resize(total);
		std::vector<char*> dst(pick.size());
		for (size_t j = 0; j < pick.size(); j++) {
			dst[j] = &f->blob[f->offs[j]];
			f->blob[f->offs[j] + f->lens[j]] = '\n';
		}
		fmap.copy(pick.data(), dst.data(), pick.size());
	} else {
		nte_host::FastaReader reader(path);
		if (!reader.ok()) {
			set_err(err, errcap, std::string("`") + path + "': cannot open");
			delete f;
			return NTEDIT_E_IO;
		}
		std::string hdr;
		for (;;) {
			const size_t before = f->blob.size();
			if (!reader.next(hdr, f->blob)) {
				break;
			}
			const void* z = memchr(f->blob.data() + before, 0, f->blob.size() - before);
			if (z) {
				f->blob.resize((size_t)((const char*)z - f->blob.data()));
			}
			const size_t len = f->blob.size() - before;
			if (len < min_len) {
				f->blob.resize(before);
				continue;
			}
			f->offs.push_back(before);
			f->lens.push_back(len);
			f->headers.push_back(hdr);
			f->blob.push_back('\n');
		}
		if (reader.io_error()) {
			set_err(err, errcap, std::string("`") + path + "': " + reader.io_error_text());
			delete f;
			return NTEDIT_E_IO;
		}
	}
	} catch (const std::bad_alloc&) {
		delete f;
		set_err(err, errcap, std::string("`") + path + "': out of memory while loading the draft");
		return NTEDIT_E_IO;
	} catch (const std::exception& ex) {
		delete f;
		set_err(err, errcap, std::string("`") + path + "': " + ex.what());
		return NTEDIT_E_IO;
	} catch (...) {
		delete f;
		set_err(err, errcap, std::string("`") + path + "': unexpected failure while loading the draft");
		return NTEDIT_E_IO;
	}
	*out = f;
	return 0;
}

uint64_t
ntedit_hip_fasta_count(const ntedit_hip_fasta* f)
{
	return f ? f->headers.size() : 0;
}

const char*
ntedit_hip_fasta_blob(const ntedit_hip_fasta* f, uint64_t* nbytes)
{
	if (!f) {
		return nullptr;
	}
	if (nbytes) {
		*nbytes = f->blob.size();
	}
	return f->blob.data();
}

int
ntedit_hip_fasta_record(const ntedit_hip_fasta* f, uint64_t i, const char** header, uint64_t* header_len, uint64_t* offset, uint64_t* len)
{
	if (!f || i >= f->headers.size()) {
		return NTEDIT_E_ARG;
	}
	if (header) {
		*header = f->headers[i].data();
	}
	if (header_len) {
		*header_len = f->headers[i].size();
	}
	if (offset) {
		*offset = f->offs[i];
	}
	if (len) {
		*len = f->lens[i];
	}
	return 0;
}

void
ntedit_hip_fasta_free(ntedit_hip_fasta* f)
{
	delete f;
}

} // extern "C"
