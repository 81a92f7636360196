#include "fasta_map.h"

#include <atomic>
#include <cctype>
#include <cstring>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <thread>
#include <unistd.h>
#include <zlib.h>

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

FastaMap::FastaMap(const char* path, unsigned threads, unsigned inflate_threads)
  : threads_(threads ? threads : 1)
  , inflate_threads_(inflate_threads ? inflate_threads : threads_)
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
	if (size_ >= 28 && (unsigned char)data_[0] == 0x1f && (unsigned char)data_[1] == 0x8b && !inflate_bgzf()) {
		return; // ordinary gzip, or a damaged file: the streaming reader reads (and reports) those
	}
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

// BGZF (the SAM specification, section 4.1): a series of gzip members, each with FLG = FEXTRA and an extra subfield
// 'B','C' of two bytes holding the member's total size - 1; CRC32 and ISIZE close each member, an empty member ends
// the file.  Anything that does not fit this exactly makes the function return false with data_/size_ untouched.
bool
FastaMap::inflate_bgzf()
{
	struct Block
	{
		uint64_t in, out; // offsets of the deflate data / of the inflated bytes
		uint32_t n_in, n_out, crc;
	};
	const unsigned char* d = (const unsigned char*)data_;
	std::vector<Block> blocks;
	uint64_t o = 0, total = 0;
	auto u16 = [&](uint64_t at) { return (uint32_t)d[at] | (uint32_t)d[at + 1] << 8; };
	auto u32 = [&](uint64_t at) { return u16(at) | u16(at + 2) << 16; };
	while (o < size_) {
		if (size_ - o < 12 + 6 + 8 || d[o] != 0x1f || d[o + 1] != 0x8b || d[o + 2] != 8 || d[o + 3] != 4) {
			return false;
		}
		const uint32_t xlen = u16(o + 10);
		if (12 + (uint64_t)xlen + 8 > size_ - o) {
			return false;
		}
		uint32_t bsize = 0;
		bool have = false;
		for (uint64_t x = o + 12; x + 4 <= o + 12 + xlen;) {
			const uint32_t slen = u16(x + 2);
			if (d[x] == 'B' && d[x + 1] == 'C' && slen == 2 && x + 6 <= o + 12 + xlen) {
				bsize = u16(x + 4);
				have = true;
			}
			x += 4 + (uint64_t)slen;
		}
		const uint64_t member = (uint64_t)bsize + 1;
		if (!have || member < 12 + (uint64_t)xlen + 8 || member > size_ - o) {
			return false;
		}
		Block b;
		b.in = o + 12 + xlen;
		b.n_in = (uint32_t)(member - 12 - xlen - 8);
		b.crc = u32(o + member - 8);
		b.n_out = u32(o + member - 4);
		b.out = total;
		if (b.n_out > 65536) {
			return false;
		}
		total += b.n_out;
		blocks.push_back(b);
		o += member;
	}
	if (total < 2) {
		return false;
	}
	void* buf = mmap(nullptr, total, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
	if (buf == MAP_FAILED) {
		return false;
	}
	(void)madvise(buf, total, MADV_HUGEPAGE);
	std::atomic<bool> good(true);
	const size_t GROUP = 64; // blocks per task
	parallel_for(inflate_threads_, (blocks.size() + GROUP - 1) / GROUP, [&](size_t g) {
		z_stream zs;
		memset(&zs, 0, sizeof zs);
		if (inflateInit2(&zs, -15) != Z_OK) {
			good = false;
			return;
		}
		const size_t hi = (g + 1) * GROUP < blocks.size() ? (g + 1) * GROUP : blocks.size();
		for (size_t i = g * GROUP; i < hi && good.load(std::memory_order_relaxed); i++) {
			const Block& b = blocks[i];
			Bytef* dst = (Bytef*)buf + b.out;
			zs.next_in = const_cast<Bytef*>(d + b.in);
			zs.avail_in = b.n_in;
			zs.next_out = dst;
			zs.avail_out = b.n_out;
			const int rc = inflate(&zs, Z_FINISH);
			if (rc != Z_STREAM_END || zs.avail_out != 0 || zs.avail_in != 0 ||
			    (uint32_t)crc32(crc32(0L, Z_NULL, 0), dst, b.n_out) != b.crc) {
				good = false;
			}
			inflateReset(&zs);
		}
		inflateEnd(&zs);
	});
	if (!good) {
		munmap(buf, total);
		return false;
	}
	munmap((void*)data_, size_);
	data_ = (const char*)buf;
	size_ = total;
	return true;
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

void
FastaMap::copy_range(size_t i, uint64_t start, uint64_t n, char* dst) const
{
	const Rec& r = recs_[i];
	const char* p = data_ + r.seq;
	const char* e = data_ + r.end;
	uint64_t skip = start;
	while (p < e && n) {
		const char* q = (const char*)memchr(p, '\n', (size_t)(e - p));
		uint64_t len = q ? (uint64_t)(q - p) : (uint64_t)(e - p);
		const char* line = p;
		p += len + 1;
		if (skip >= len) {
			skip -= len;
			continue;
		}
		line += skip;
		len -= skip;
		skip = 0;
		if (len > n) {
			len = n;
		}
		memcpy(dst, line, (size_t)len);
		dst += len;
		n -= len;
	}
}

} // namespace nte_host
