#include "fasta_map.h"

#include <atomic>
#include <cctype>
#include <cstring>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <thread>
#include <unistd.h>

namespace nte_host {

namespace {

template<typename F>
void
parallel_for(unsigned threads, size_t n, F f)
{
	if (threads <= 1 || n <= 1) {
		for (size_t i = 0; i < n; i++) {
			f(i);
		}
		return;
	}
	std::atomic<size_t> next(0);
	std::vector<std::thread> pool;
	const unsigned T = threads < n ? threads : (unsigned)n;
	for (unsigned t = 0; t < T; t++) {
		pool.emplace_back([&]() {
			for (;;) {
				const size_t i = next.fetch_add(1);
				if (i >= n) {
					return;
				}
				f(i);
			}
		});
	}
	for (std::thread& t : pool) {
		t.join();
	}
}

} // namespace

FastaMap::FastaMap(const char* path, unsigned threads)
  : threads_(threads ? threads : 1)
{
	fd_ = open(path, O_RDONLY);
	if (fd_ < 0) {
		return;
	}
	struct stat st;
	if (fstat(fd_, &st) != 0 || !S_ISREG(st.st_mode) || st.st_size < 2) {
		return;
	}
	size_ = (uint64_t)st.st_size;
	void* p = mmap(nullptr, size_, PROT_READ, MAP_PRIVATE, fd_, 0);
	if (p == MAP_FAILED) {
		return;
	}
	data_ = (const char*)p;
	(void)madvise(p, size_, MADV_WILLNEED);
	if (data_[0] != '>') {
		return; // gzip (1f 8b), FASTQ, leading text: the streaming reader's business
	}
	// ---- record starts: T regions, each scanned line by line
	const size_t n_regions = threads_ > 1 ? (size_t)threads_ * 4 : 1;
	const uint64_t step = (size_ + n_regions - 1) / n_regions;
	std::vector<std::vector<uint64_t>> found(n_regions);
	std::atomic<bool> plain(true);
	parallel_for(threads_, n_regions, [&](size_t ri) {
		const uint64_t a = ri * step, b = a + step < size_ ? a + step : size_;
		if (a >= b) {
			return;
		}
		if (memchr(data_ + a, 0, b - a)) {
			plain = false; // (a sequence with an embedded NUL ends there in the reference)
			return;
		}
		const char* p = data_ + a;
		const char* e = data_ + b;
		while (p < e && plain.load(std::memory_order_relaxed)) {
			const char* q = (const char*)memchr(p, '\n', (size_t)(e - p));
			if (!q) {
				break;
			}
			if (q > data_ && q[-1] == '\r') {
				plain = false;
				return;
			}
			const uint64_t nx = (uint64_t)(q + 1 - data_);
			if (nx < size_) {
				const char c = data_[nx];
				if (c == '>') {
					found[ri].push_back(nx);
				} else if (c == '@' || c == '+') {
					plain = false; // FASTQ, or a sequence line kseq would take for one
					return;
				}
			}
			p = q + 1;
		}
	});
	if (!plain) {
		return;
	}
	if (size_ && data_[size_ - 1] == '\r') {
		return;
	}
	std::vector<uint64_t> starts;
	starts.push_back(0);
	for (const auto& v : found) {
		starts.insert(starts.end(), v.begin(), v.end());
	}
	recs_.resize(starts.size());
	for (size_t i = 0; i < starts.size(); i++) {
		Rec& r = recs_[i];
		r.start = starts[i];
		r.end = i + 1 < starts.size() ? starts[i + 1] : size_;
		const char* nl = (const char*)memchr(data_ + r.start, '\n', (size_t)(r.end - r.start));
		r.seq = nl ? (uint64_t)(nl + 1 - data_) : r.end;
		r.len = ~0ULL;
	}
	if (!recs_.empty() && recs_.back().start + 1 >= size_) {
		recs_.pop_back(); // a lone '>' at the very end of the file: kseq finds no name to read and stops
	}
	ok_ = true;
}

FastaMap::~FastaMap()
{
	if (data_) {
		munmap((void*)data_, size_);
	}
	if (fd_ >= 0) {
		close(fd_);
	}
}

std::string
FastaMap::header(size_t i) const
{
	// the header line without '>' and '\n': name = up to the first whitespace character; if there is more,
	// comment = the rest of the line (lib/kseq.h:189-190) -- same as FastaReader::next() on a line without CR / NUL
	const Rec& r = recs_[i];
	const char* p = data_ + r.start + 1;
	uint64_t n = r.seq - (r.start + 1);
	if (n && p[n - 1] == '\n') {
		n--;
	}
	uint64_t nl = 0;
	while (nl < n && !isspace((unsigned char)p[nl])) {
		nl++;
	}
	std::string h(p, nl);
	if (nl < n) {
		const uint64_t cl = n - (nl + 1);
		if (cl) {
			h.push_back(' ');
			h.append(p + nl + 1, cl);
		}
	}
	return h;
}

void
FastaMap::measure(size_t first, size_t count)
{
	parallel_for(threads_, count, [&](size_t j) {
		Rec& r = recs_[first + j];
		if (r.len != ~0ULL) {
			return;
		}
		const char* p = data_ + r.seq;
		const char* e = data_ + r.end;
		uint64_t nl = 0;
		while (p < e) {
			const char* q = (const char*)memchr(p, '\n', (size_t)(e - p));
			if (!q) {
				break;
			}
			nl++;
			p = q + 1;
		}
		r.len = (r.end - r.seq) - nl;
	});
}

void
FastaMap::copy(const size_t* idx, char* const* dst, size_t n) const
{
	// large records first would balance better; records of a batch are few hundred at most and the pool is dynamic
	parallel_for(threads_, n, [&](size_t j) {
		const Rec& r = recs_[idx[j]];
		const char* p = data_ + r.seq;
		const char* e = data_ + r.end;
		char* d = dst[j];
		while (p < e) {
			const char* q = (const char*)memchr(p, '\n', (size_t)(e - p));
			const size_t len = q ? (size_t)(q - p) : (size_t)(e - p);
			memcpy(d, p, len);
			d += len;
			p += len + 1;
		}
	});
}

} // namespace nte_host
