// fasta.h -- streaming FASTA/FASTQ (optionally gzipped) reader for the host
// driver.  Record semantics follow what the reference gets from kseq
// (lib/kseq.h:176-215 as used at ntedit.cpp:2223-2230): name = header text up
// to the first whitespace, comment = the rest of the header line, sequence =
// all following lines concatenated.  Own implementation on top of zlib.
#pragma once
#include <string>
#include <zlib.h>

namespace nte_host {

class FastaReader
{
  public:
	explicit FastaReader(const char* path);
	~FastaReader();
	bool ok() const { return f_ != nullptr; }
	// reads the next record; header = name [+ " " + comment]; the sequence is APPENDED to seq
	// (so a batch can be assembled without an intermediate copy); false at EOF
	bool next(std::string& header, std::string& seq);

  private:
	bool fill_();
	int getc_();
	bool getline_(std::string& out, size_t base, bool strip_cr);
	std::string line_;
	gzFile f_;
	unsigned char* buf_;
	int begin_, end_;
	bool eof_;
	int last_char_;
	bool hit_nl_, failed_;
};

} // namespace nte_host
