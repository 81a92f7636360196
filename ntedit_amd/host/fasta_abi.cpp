// fasta_abi.cpp -- the host binary's FASTA ingest behind the C ABI (ntedit_hip_fasta_*), so that every driver of
// the hot path (the `ntedit` binary, python -m ntedit_amd.run, a reference maintainer's own main) reads a draft
// with ONE implementation of kseq's record semantics (lib/kseq.h:176-215 as used at ntedit.cpp:2223-2230):
// the mapped multi-threaded reader for plain and BGZF multi-FASTA, the streaming kseq restatement for everything
// else (ordinary gzip -- found by its magic bytes, not by the file name --, FASTQ, CR line ends, text in front of
// the first record).  A sequence that holds a NUL byte ends there (contigSeq = seq->seq.s, ntedit.cpp:2230).
#include "../../include/ntedit_hip.h"
#include "fasta.h"
#include "fasta_map.h"

#include "../csrc/nte_common.h"

#include <atomic>
#include <cstdio>
#include <cstring>
#include <mutex>
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

// ---- the packed form of a batch (include/ntedit_hip.h)
uint64_t
ntedit_hip_packed_size(uint64_t n)
{
	return (n + 31) / 32 * 16 + (n + 127) / 128 * 16;
}

int
ntedit_hip_pack_bases(const char* bases, uint64_t n, void* packed, unsigned threads)
{
	if ((n && !bases) || !packed) {
		return NTEDIT_E_ARG;
	}
	// per byte: low nibble = code, bit 4 = lower-case letter, bit 7 = the packed form cannot carry this byte
	static uint8_t lut[256];
	static std::once_flag once;
	std::call_once(once, []() {
		for (int c = 0; c < 256; c++) {
			const uint8_t code = nte::char_code((uint8_t)c);
			lut[c] = (uint8_t)(code | ((c >= 'a' && c <= 'z') ? 0x10 : 0) | (nte::is_exotic((uint8_t)c) ? 0x80 : 0));
		}
	});
	uint8_t* codes = (uint8_t*)packed;
	uint8_t* cases = codes + (n + 31) / 32 * 16;
	memset(cases, 0, (size_t)((n + 127) / 128 * 16));
	if (n & 1) {
		codes[n / 2] = 0xF0; // (the odd tail's high nibble; the padding behind it is never read as bases)
	}
	if (threads == 0) {
		threads = std::thread::hardware_concurrency();
		threads = threads > 16 ? 16 : (threads < 1 ? 1 : threads);
	}
	const uint64_t unit = 1u << 20; // (a multiple of 128: no two threads share a byte of either section)
	const uint64_t n_units = (n + unit - 1) / unit;
	if (threads > n_units) {
		threads = (unsigned)(n_units ? n_units : 1);
	}
	std::atomic<uint64_t> next(0);
	std::atomic<int> bad(0);
	auto work = [&]() {
		for (;;) {
			const uint64_t u = next.fetch_add(1);
			if (u >= n_units || bad.load(std::memory_order_relaxed)) {
				return;
			}
			const uint64_t a = u * unit, b = a + unit < n ? a + unit : n;
			const uint8_t* src = (const uint8_t*)bases;
			uint8_t flags = 0;
			uint64_t i = a;
			for (; i + 8 <= b; i += 8) {
				uint8_t t[8];
				for (int q = 0; q < 8; q++) {
					t[q] = lut[src[i + q]];
				}
				flags |= t[0] | t[1] | t[2] | t[3] | t[4] | t[5] | t[6] | t[7];
				codes[i / 2] = (uint8_t)((t[0] & 15) | (t[1] << 4));
				codes[i / 2 + 1] = (uint8_t)((t[2] & 15) | (t[3] << 4));
				codes[i / 2 + 2] = (uint8_t)((t[4] & 15) | (t[5] << 4));
				codes[i / 2 + 3] = (uint8_t)((t[6] & 15) | (t[7] << 4));
				cases[i / 8] = (uint8_t)(((t[0] >> 4) & 1) | ((t[1] >> 3) & 2) | ((t[2] >> 2) & 4) | ((t[3] >> 1) & 8) | (t[4] & 16) |
				                         ((t[5] << 1) & 32) | ((t[6] << 2) & 64) | ((t[7] << 3) & 128));
			}
			for (; i < b; i++) { // (the batch's last bytes)
				const uint8_t t = lut[src[i]];
				flags |= t;
				if (i & 1) {
					codes[i / 2] = (uint8_t)((codes[i / 2] & 15) | (t << 4));
				} else {
					codes[i / 2] = (uint8_t)((codes[i / 2] & 0xF0) | (t & 15));
				}
				cases[i / 8] |= (uint8_t)(((t >> 4) & 1) << (i & 7));
			}
			if (flags & 0x80) {
				bad.store(1);
			}
		}
	};
	std::vector<std::thread> pool;
	for (unsigned t = 1; t < threads; t++) {
		pool.emplace_back(work);
	}
	work();
	for (std::thread& t : pool) {
		t.join();
	}
	return bad.load() ? 1 : 0;
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
