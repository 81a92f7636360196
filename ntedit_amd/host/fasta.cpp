#include "fasta.h"

#include <cctype>
#include <cstdlib>

namespace nte_host {

static const int BUFSZ = 1 << 18;

FastaReader::FastaReader(const char* path)
  : f_(gzopen(path, "r"))
  , buf_((unsigned char*)malloc(BUFSZ))
  , begin_(0)
  , end_(0)
  , eof_(false)
  , last_char_(0)
{
	if (f_) {
		gzbuffer(f_, 1 << 20);
	}
}

FastaReader::~FastaReader()
{
	if (f_) {
		gzclose(f_);
	}
	free(buf_);
}

int
FastaReader::getc_()
{
	if (begin_ >= end_) {
		if (eof_) {
			return -1;
		}
		begin_ = 0;
		end_ = gzread(f_, buf_, BUFSZ);
		if (end_ <= 0) {
			end_ = 0;
			eof_ = true;
			return -1;
		}
	}
	return buf_[begin_++];
}

// reads up to (not including) the next '\n'; strips one trailing '\r'
bool
FastaReader::getline_(std::string& out, bool append)
{
	if (!append) {
		out.clear();
	}
	bool any = false;
	for (;;) {
		if (begin_ >= end_) {
			if (eof_) {
				break;
			}
			begin_ = 0;
			end_ = gzread(f_, buf_, BUFSZ);
			if (end_ <= 0) {
				end_ = 0;
				eof_ = true;
				break;
			}
		}
		any = true;
		int i = begin_;
		while (i < end_ && buf_[i] != '\n') {
			i++;
		}
		out.append((const char*)buf_ + begin_, (size_t)(i - begin_));
		begin_ = i + 1;
		if (i < end_) {
			break;
		}
	}
	if (out.size() > 1 && out.back() == '\r') {
		out.pop_back();
	}
	return any;
}

bool
FastaReader::next(std::string& header, std::string& seq)
{
	int c;
	if (last_char_ == 0) {
		while ((c = getc_()) != -1 && c != '>' && c != '@') {
		}
		if (c == -1) {
			return false;
		}
		last_char_ = c;
	}
	std::string line;
	if (!getline_(line, false) && eof_) {
		return false;
	}
	// name up to the first whitespace; comment = rest of the line
	size_t nl = 0;
	while (nl < line.size() && !isspace((unsigned char)line[nl])) {
		nl++;
	}
	header.assign(line, 0, nl);
	if (nl + 1 < line.size()) {
		header.push_back(' ');
		header.append(line, nl + 1, std::string::npos);
	}
	seq.clear();
	last_char_ = 0;
	while ((c = getc_()) != -1 && c != '>' && c != '+' && c != '@') {
		if (c == '\n') {
			continue;
		}
		seq.push_back((char)c);
		getline_(seq, true);
	}
	if (c == '>' || c == '@') {
		last_char_ = c;
	}
	if (c == '+') {
		// FASTQ: skip the '+' line and as many quality characters as bases
		std::string q;
		getline_(q, false);
		size_t have = 0;
		while (have < seq.size() && getline_(q, false)) {
			have += q.size();
		}
	}
	return true;
}

} // namespace nte_host
