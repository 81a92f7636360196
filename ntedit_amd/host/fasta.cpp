#include "fasta.h"
#include "gunzip.h"

#include <atomic>
#include <cctype>
#include <cstdlib>
#include <cstring>
#include <sys/stat.h>

namespace nte_host {

static std::atomic<int> g_gzip_through_zlib{ 0 };

int
set_gzip_through_zlib(int on)
{
	return g_gzip_through_zlib.exchange(on ? 1 : 0);
}

// large reads go straight from the file into the block when the input is not compressed
// (zlib copies directly once a request is at least twice its own buffer)
static const int BUFSZ = 4 << 20;

// (every block has room for the decoder's window in front of it and for an overrunning match behind it)
unsigned char*
FastaReader::data_(int s) const
{
	return slot_[s] + Gunzip::WINDOW;
}

FastaReader::FastaReader(const char* path)
  : f_(nullptr)
  , gz_(nullptr)
  , buf_(nullptr)
  , begin_(0)
  , end_(0)
  , eof_(false)
  , last_char_(0)
  , hit_nl_(false)
  , failed_(false)
  , head_(0)
  , tail_(0)
  , crc_done_(0)
  , cur_(-1)
  , stop_(false)
{
	for (int i = 0; i < NSLOTS; i++) {
		slot_[i] = nullptr;
		slot_len_[i] = 0;
		slot_eof_[i] = slot_member_end_[i] = false;
		slot_crc_[i] = slot_isize_[i] = 0;
	}
	struct stat st;
	if (!g_gzip_through_zlib.load() && stat(path, &st) == 0 && S_ISREG(st.st_mode)) {
		gz_ = new Gunzip();
		if (!gz_->open(path)) {
			delete gz_; // not a gzip stream (or not readable: gzopen says so below)
			gz_ = nullptr;
		}
	}
	if (!gz_) {
		f_ = gzopen(path, "r");
	}
	if (f_ || gz_) {
		if (f_) {
			gzbuffer(f_, 1 << 17);
		}
		for (int i = 0; i < NSLOTS; i++) {
			slot_[i] = (unsigned char*)malloc(Gunzip::WINDOW + BUFSZ + Gunzip::SLACK);
		}
		io_ = std::thread([this]() { io_loop_(); });
		if (gz_) {
			crc_ = std::thread([this]() { crc_loop_(); });
		}
	}
}

FastaReader::~FastaReader()
{
	if (io_.joinable()) {
		{
			std::lock_guard<std::mutex> lk(mu_);
			stop_ = true;
		}
		cv_.notify_all();
		io_.join();
		if (crc_.joinable()) {
			crc_.join();
		}
	}
	if (f_) {
		gzclose(f_);
	}
	delete gz_;
	for (int i = 0; i < NSLOTS; i++) {
		free(slot_[i]);
	}
}

// I/O thread: reads (and, for .gz, inflates) the file block by block ahead of the parser
void
FastaReader::io_loop_()
{
	for (;;) {
		int s;
		{
			std::unique_lock<std::mutex> lk(mu_);
			cv_.wait(lk, [&]() { return stop_ || (head_ - tail_ < NSLOTS && (!gz_ || head_ - crc_done_ < NSLOTS)); });
			if (stop_) {
				return;
			}
			s = (int)(head_ % NSLOTS);
		}
		int n;
		bool bad = false, member_end = false;
		unsigned crc = 0, isize = 0;
		std::string why;
		if (gz_) {
			n = (int)gz_->read(data_(s), BUFSZ);
			member_end = gz_->member_end();
			crc = gz_->member_crc();
			isize = gz_->member_isize();
			if (gz_->failed()) {
				bad = true;
				why = gz_->error();
				n = 0;
			}
		} else {
			n = gzread(f_, data_(s), BUFSZ);
			bad = n < 0;
			if (n < BUFSZ) {
				// a short or failed read: end of file, or a stream that broke (zlib reports a truncated
				// .gz as Z_BUF_ERROR and corrupt data as Z_DATA_ERROR only through gzerror)
				int errnum = Z_OK;
				const char* msg = gzerror(f_, &errnum);
				if (errnum != Z_OK && errnum != Z_STREAM_END) {
					bad = true;
					why = msg ? msg : "read error";
				}
			}
			if (n < 0) {
				n = 0;
			}
		}
		const bool eof = n == 0 && !member_end;
		{
			std::lock_guard<std::mutex> lk(mu_);
			if (bad) {
				io_error_ = true;
				io_error_text_ = why.empty() ? "read error" : why;
			}
			slot_len_[s] = n;
			slot_eof_[s] = eof;
			slot_member_end_[s] = member_end;
			slot_crc_[s] = crc;
			slot_isize_[s] = isize;
			head_++;
		}
		cv_.notify_all();
		if (eof) {
			return; // end of file (an empty block marks it)
		}
	}
}

// checksum thread (our own decoder only): CRC-32 and length of every gzip member against its trailer, block by
// block behind the I/O thread -- at the decoder's speed the checksum would otherwise take a third of its time
void
FastaReader::crc_loop_()
{
	unsigned long running = crc32(0L, Z_NULL, 0);
	unsigned long long total = 0;
	for (;;) {
		int s, n;
		bool eof, member_end;
		unsigned crc, isize;
		{
			std::unique_lock<std::mutex> lk(mu_);
			cv_.wait(lk, [&]() { return stop_ || crc_done_ < head_; });
			if (stop_) {
				return;
			}
			s = (int)(crc_done_ % NSLOTS);
			n = slot_len_[s];
			eof = slot_eof_[s];
			member_end = slot_member_end_[s];
			crc = slot_crc_[s];
			isize = slot_isize_[s];
		}
		const char* why = nullptr;
		if (!eof) {
			running = crc32(running, data_(s), (unsigned)n);
			total += (unsigned long long)n;
			if (member_end) {
				if ((unsigned)running != crc) {
					why = "incorrect data check";
				} else if ((unsigned)(total & 0xffffffffull) != isize) {
					why = "incorrect length check";
				}
				running = crc32(0L, Z_NULL, 0);
				total = 0;
			}
		}
		{
			std::lock_guard<std::mutex> lk(mu_);
			if (why && !io_error_) {
				io_error_ = true;
				io_error_text_ = why;
			}
			crc_done_++;
		}
		cv_.notify_all();
		if (eof) {
			return;
		}
	}
}

bool
FastaReader::fill_()
{
	if (eof_) {
		return false;
	}
	std::unique_lock<std::mutex> lk(mu_);
	for (;;) {
		if (cur_ >= 0) {
			tail_++; // the block the parser has finished with goes back to the I/O thread
			cur_ = -1;
			cv_.notify_all();
		}
		cv_.wait(lk, [&]() { return head_ > tail_; });
		const int s = (int)(tail_ % NSLOTS);
		if (slot_eof_[s]) {
			// (what io_error() says must be final when next() reports the end: wait for the last checksum)
			cv_.wait(lk, [&]() { return !gz_ || crc_done_ >= head_; });
			eof_ = true;
			begin_ = end_ = 0;
			return false;
		}
		cur_ = s;
		if (slot_len_[s] > 0) {
			break;
		}
		// (an empty block that is not the last one: a gzip member without data)
	}
	buf_ = data_(cur_);
	begin_ = 0;
	end_ = slot_len_[cur_];
	return true;
}

int
FastaReader::getc_()
{
	if (begin_ >= end_ && !fill_()) {
		return -1;
	}
	return buf_[begin_++];
}

// appends the rest of the current line (up to, not including, '\n') to `out`; with strip_cr,
// then drops one trailing '\r' if the string that starts at out[base] is longer than one
// character (kseq's ks_getuntil2 rule for line reads, applied to the accumulated string)
bool
FastaReader::getline_(std::string& out, size_t base, bool strip_cr)
{
	bool any = false;
	hit_nl_ = false;
	for (;;) {
		if (begin_ >= end_ && !fill_()) {
			break;
		}
		any = true;
		const unsigned char* s = buf_ + begin_;
		const size_t avail = (size_t)(end_ - begin_);
		const unsigned char* nl = (const unsigned char*)memchr(s, '\n', avail);
		const size_t n = nl ? (size_t)(nl - s) : avail;
		out.append((const char*)s, n);
		begin_ += (int)n + (nl ? 1 : 0);
		if (nl) {
			hit_nl_ = true;
			break;
		}
	}
	if (strip_cr && out.size() - base > 1 && out.back() == '\r') {
		out.pop_back();
	}
	return any;
}

bool
FastaReader::next(std::string& header, std::string& seq)
{
	int c;
	if (failed_ || !ok()) {
		return false;
	}
	if (last_char_ == 0) {
		while ((c = getc_()) != -1 && c != '>' && c != '@') {
		}
		if (c == -1) {
			return false;
		}
		last_char_ = c;
	}
	line_.clear();
	if (!getline_(line_, 0, false) && eof_) {
		return false;
	}
	// name = up to the first whitespace character; if that character was not the newline,
	// comment = the rest of the line (lib/kseq.h:189-190).  Both are consumed as C strings
	// (ntedit.cpp:2223-2226).
	size_t nl = 0;
	while (nl < line_.size() && !isspace((unsigned char)line_[nl])) {
		nl++;
	}
	header.assign(line_, 0, strnlen(line_.data(), nl));
	if (nl < line_.size()) {
		size_t cl = line_.size() - (nl + 1);
		if (cl > 1 && line_.back() == '\r') {
			cl--;
		}
		if (cl) {
			header.push_back(' ');
			header.append(line_, nl + 1, strnlen(line_.data() + nl + 1, cl));
		}
	}
	const size_t base = seq.size();
	last_char_ = 0;
	while ((c = getc_()) != -1 && c != '>' && c != '+' && c != '@') {
		if (c == '\n') {
			continue;
		}
		seq.push_back((char)c);
		getline_(seq, base, true);
	}
	if (c == '>' || c == '@') {
		last_char_ = c;
	}
	if (c == '+') {
		// FASTQ: skip the rest of the '+' line, then quality lines until there are as many
		// quality characters as bases (lib/kseq.h:205-212)
		const size_t want = seq.size() - base;
		line_.clear();
		getline_(line_, 0, false);
		if (!hit_nl_) {
			failed_ = true; // kseq_read() = -2 (no quality string): the reference stops reading
			seq.resize(base);
			return false;
		}
		line_.clear();
		while (getline_(line_, 0, true) && line_.size() < want) {
		}
		last_char_ = 0;
		if (line_.size() != want) {
			failed_ = true; // kseq_read() = -2 (truncated quality string)
			seq.resize(base);
			return false;
		}
	}
	return true;
}

} // namespace nte_host
