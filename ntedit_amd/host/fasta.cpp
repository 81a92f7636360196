#include "fasta.h"

#include <cctype>
#include <cstdlib>
#include <cstring>

namespace nte_host {

// large reads go straight from the file into the block when the input is not compressed
// (zlib copies directly once a request is at least twice its own buffer)
static const int BUFSZ = 4 << 20;

FastaReader::FastaReader(const char* path)
  : f_(gzopen(path, "r"))
  , buf_(nullptr)
  , begin_(0)
  , end_(0)
  , eof_(false)
  , last_char_(0)
  , hit_nl_(false)
  , failed_(false)
  , head_(0)
  , tail_(0)
  , cur_(-1)
  , stop_(false)
{
	for (int i = 0; i < NSLOTS; i++) {
		slot_[i] = nullptr;
		slot_len_[i] = 0;
	}
	if (f_) {
		gzbuffer(f_, 1 << 17);
		for (int i = 0; i < NSLOTS; i++) {
			slot_[i] = (unsigned char*)malloc(BUFSZ);
		}
		io_ = std::thread([this]() { io_loop_(); });
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
	}
	if (f_) {
		gzclose(f_);
	}
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
			cv_.wait(lk, [&]() { return stop_ || head_ - tail_ < NSLOTS; });
			if (stop_) {
				return;
			}
			s = (int)(head_ % NSLOTS);
		}
		int n = gzread(f_, slot_[s], BUFSZ);
		bool bad = n < 0;
		std::string why;
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
		{
			std::lock_guard<std::mutex> lk(mu_);
			if (bad) {
				io_error_ = true;
				io_error_text_ = why.empty() ? "read error" : why;
			}
			slot_len_[s] = n;
			head_++;
		}
		cv_.notify_all();
		if (n == 0) {
			return; // end of file (an empty block marks it)
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
	if (cur_ >= 0) {
		tail_++; // the block the parser has finished with goes back to the I/O thread
		cur_ = -1;
		cv_.notify_all();
	}
	cv_.wait(lk, [&]() { return head_ > tail_; });
	const int s = (int)(tail_ % NSLOTS);
	if (slot_len_[s] <= 0) {
		eof_ = true;
		begin_ = end_ = 0;
		return false;
	}
	cur_ = s;
	buf_ = slot_[s];
	begin_ = 0;
	end_ = slot_len_[s];
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
	if (failed_ || !f_) {
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
