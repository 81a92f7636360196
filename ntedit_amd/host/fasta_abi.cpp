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
#include <cstdlib>
#include <immintrin.h>
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
	// ntedit_hip_fasta_open: the mapped file and its record index instead of a blob (offs[] = ~0); sequences are read
	// on demand (ntedit_hip_fasta_read)
	nte_host::FastaMap* map = nullptr;
	std::vector<size_t> map_idx; // record i of this handle = record map_idx[i] of the file
	~ntedit_hip_fasta() { delete map; }
};

static void
set_err(char* err, size_t cap, const std::string& text)
{
	if (err && cap) {
		snprintf(err, cap, "%s", text.c_str());
	}
}

extern "C" {

// The draft's INDEX without its bases: the file is mapped, its records are found and measured by `threads` threads, and
// nothing is copied -- a rank of a multi-GPU run plans the partition from the lengths and then reads its own pieces
// (and the 64 KB windows its cuts are looked for in) with ntedit_hip_fasta_read.  Inputs the mapped reader does not
// take (single-stream gzip, FASTQ, CR line ends, ...) are loaded whole, as ntedit_hip_fasta_load does: same calls,
// same results, the bases then come out of memory.
int
ntedit_hip_fasta_open(const char* path, uint64_t min_len, unsigned threads, ntedit_hip_fasta** out, char* err, size_t errcap)
{
	if (!path || !out) {
		set_err(err, errcap, "fasta_open: bad argument");
		return NTEDIT_E_ARG;
	}
	*out = nullptr;
	unsigned t = threads;
	if (t == 0) {
		t = std::thread::hardware_concurrency();
		t = t > 16 ? 16 : (t < 1 ? 1 : t);
	}
	try {
		nte_host::FastaMap* m = new nte_host::FastaMap(path, t, t > 64 ? 64 : t);
		if (m->ok()) {
			ntedit_hip_fasta* f = new ntedit_hip_fasta();
			f->map = m;
			const size_t N = m->records();
			m->measure(0, N);
			for (size_t i = 0; i < N; i++) {
				if (m->length(i) >= min_len) {
					f->map_idx.push_back(i);
					f->offs.push_back(~0ULL);
					f->lens.push_back(m->length(i));
					f->headers.push_back(m->header(i));
				}
			}
			*out = f;
			return 0;
		}
		delete m;
	} catch (...) {
		set_err(err, errcap, std::string("`") + path + "': failure while indexing the draft");
		return NTEDIT_E_IO;
	}
	return ntedit_hip_fasta_load(path, min_len, threads, out, err, errcap);
}

// bases [start, start + n) of record i of a handle of ntedit_hip_fasta_open / _load to dst
int
ntedit_hip_fasta_read(const ntedit_hip_fasta* f, uint64_t i, uint64_t start, uint64_t n, char* dst)
{
	if (!f || i >= f->lens.size() || start > f->lens[i] || n > f->lens[i] - start || (n && !dst)) {
		return NTEDIT_E_ARG;
	}
	if (n == 0) {
		return 0;
	}
	if (f->map) {
		f->map->copy_range(f->map_idx[i], start, n, dst);
	} else {
		memcpy(dst, f->blob.data() + f->offs[i] + start, (size_t)n);
	}
	return 0;
}

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

} // extern "C"

unsigned nte_host_threads_setting(); // (nte_api_results.inc: the ntedit_hip_set_host_threads() value, 0 = unset)

// 32 bases per step: classification by two 16-entry shuffles on the case-folded letter (0x40..0x5F), the nibbles of
// two neighbours joined by one multiply-add, the case bits by a byte mask.  Same bytes out as the table loop below
// (tests/test_pack.py runs both, threads | 1 << 31 = the table loop); ~10 GB/s per thread where that loop does 1.  Handles [a, b) in whole steps of 32,
// returns where it stopped; *bad != 0: a byte the packed form cannot carry (nte::is_exotic).
__attribute__((target("avx2"))) static uint64_t
pack_avx2(const uint8_t* src, uint64_t a, uint64_t b, uint8_t* codes, uint8_t* cases, int* bad)
{
	// codes of '@' A B C D E F G H I J K L M N O / P Q R S T U V W X Y Z [ \\ ] ^ _   (nte::char_code; 15 = none)
	const __m256i tab_lo = _mm256_setr_epi8(15, 0, 10, 1, 11, 15, 15, 2, 12, 15, 15, 8, 15, 9, 15, 15, 15, 0, 10, 1, 11, 15, 15, 2, 12, 15, 15, 8, 15, 9, 15, 15);
	const __m256i tab_hi = _mm256_setr_epi8(15, 15, 4, 6, 3, 15, 13, 7, 15, 5, 15, 15, 15, 15, 15, 15, 15, 15, 4, 6, 3, 15, 13, 7, 15, 5, 15, 15, 15, 15, 15, 15);
	// a byte without a code whose ntHash seeds are not zero: (c & 7) in {1, 3, 4, 5, 7} (nte::seed_rev_raw; the bytes
	// with a forward seed -- U, u, 1, 3, 4, 5, 7 -- are among them)
	const __m256i tab_exo = _mm256_setr_epi8(0, -1, 0, -1, -1, -1, 0, -1, 0, 0, 0, 0, 0, 0, 0, 0, 0, -1, 0, -1, -1, -1, 0, -1, 0, 0, 0, 0, 0, 0, 0, 0);
	const __m256i c_df = _mm256_set1_epi8((char)0xDF), c_e0 = _mm256_set1_epi8((char)0xE0), c_40 = _mm256_set1_epi8(0x40);
	const __m256i c_0f = _mm256_set1_epi8(0x0F), c_10 = _mm256_set1_epi8(0x10), c_07 = _mm256_set1_epi8(0x07);
	const __m256i c_60 = _mm256_set1_epi8(0x60), c_7b = _mm256_set1_epi8(0x7B), c_15 = _mm256_set1_epi8(15);
	const __m256i mul = _mm256_set1_epi16(0x1001); // (even byte x 1 + odd byte x 16)
	__m256i exo_acc = _mm256_setzero_si256();
	uint64_t i = a;
	for (; i + 32 <= b; i += 32) {
		const __m256i x = _mm256_loadu_si256((const __m256i*)(src + i));
		const __m256i lower = _mm256_and_si256(_mm256_cmpgt_epi8(x, c_60), _mm256_cmpgt_epi8(c_7b, x)); // 'a'..'z'
		const __m256i u = _mm256_and_si256(x, c_df);
		const __m256i in_row = _mm256_cmpeq_epi8(_mm256_and_si256(u, c_e0), c_40); // 0x40..0x5F
		const __m256i lo = _mm256_shuffle_epi8(tab_lo, _mm256_and_si256(u, c_0f));
		const __m256i hi = _mm256_shuffle_epi8(tab_hi, _mm256_and_si256(u, c_0f));
		const __m256i is_hi = _mm256_cmpeq_epi8(_mm256_and_si256(u, c_10), c_10);
		__m256i code = _mm256_blendv_epi8(lo, hi, is_hi);
		code = _mm256_blendv_epi8(c_15, code, in_row);
		const __m256i none = _mm256_cmpeq_epi8(code, c_15);
		exo_acc = _mm256_or_si256(exo_acc, _mm256_and_si256(none, _mm256_shuffle_epi8(tab_exo, _mm256_and_si256(x, c_07))));
		const __m256i pairs = _mm256_maddubs_epi16(code, mul);                          // 16 x (lo | hi << 4), as 16-bit
		const __m256i packed = _mm256_permute4x64_epi64(_mm256_packus_epi16(pairs, pairs), 0xD8); // low 16 bytes: in order
		_mm_storeu_si128((__m128i*)(codes + i / 2), _mm256_castsi256_si128(packed));
		const uint32_t m = (uint32_t)_mm256_movemask_epi8(lower);
		memcpy(cases + i / 8, &m, 4);
	}
	*bad = !_mm256_testz_si256(exo_acc, exo_acc);
	return i;
}

extern "C" {

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
		// the ntedit_hip_set_host_threads() setting (the threads the renderer uses); unset: the machine's, at most 16
		threads = nte_host_threads_setting();
		if (threads == 0) {
			threads = std::thread::hardware_concurrency();
			threads = threads > 16 ? 16 : (threads < 1 ? 1 : threads);
		}
	}
	// (threads with bit 31 set: the table loop only -- the tests run both forms against each other)
	const bool use_avx2 = __builtin_cpu_supports("avx2") && !(threads & 0x80000000u);
	threads &= 0x7FFFFFFFu;
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
			if (use_avx2 && b - a >= 32) {
				int bad32 = 0;
				i = pack_avx2(src, a, b, codes, cases, &bad32);
				flags |= bad32 ? 0x80 : 0;
			}
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
