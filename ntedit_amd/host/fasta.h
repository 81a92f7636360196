// fasta.h -- streaming FASTA/FASTQ (optionally gzipped) reader for the host
// driver.  Record semantics follow what the reference gets from kseq
// (lib/kseq.h:176-215 as used at ntedit.cpp:2223-2230): name = header text up
// to the first whitespace, comment = the rest of the header line, sequence =
// all following lines concatenated.  Own implementation; the file is read (and inflated)
// block by block by a second thread, ahead of the parser.  A regular file that starts with
// the gzip magic is inflated by the decoder of gunzip.h (its member checksums verified by a
// third thread); everything else (plain text, pipes) goes through zlib's gzread.
#pragma once
#include <condition_variable>
#include <mutex>
#include <string>
#include <thread>
#include <zlib.h>

namespace nte_host {

class Gunzip;
// 1: inflate .gz inputs with zlib's gzread instead of the decoder of gunzip.h (A/B runs, tests); returns the old value
int set_gzip_through_zlib(int on);

class FastaReader
{
  public:
	explicit FastaReader(const char* path);
	~FastaReader();
	bool ok() const { return f_ != nullptr || gz_ != nullptr; }
	// set once the input turned out to be unreadable half-way (corrupt or truncated .gz, I/O error): next()
	// then returns false as at the end of the file, and the caller must not treat what it got as the whole draft
	// (kseq, lib/kseq.h:103-107, treats a failed read as the end of the input; a polished genome that is silently shorter is worse)
	bool io_error() const { return io_error_; }
	const std::string& io_error_text() const { return io_error_text_; }
	// reads the next record; header = name [+ " " + comment]; the sequence is APPENDED to seq
	// (so a batch can be assembled without an intermediate copy); false at EOF
	bool next(std::string& header, std::string& seq);

  private:
	bool fill_();
	int getc_();
	bool getline_(std::string& out, size_t base, bool strip_cr);
	std::string line_;
	void io_loop_();
	void crc_loop_();
	unsigned char* data_(int s) const;
	gzFile f_;
	Gunzip* gz_; // set: the file is a gzip stream inflated by our own decoder (f_ is not used then)
	unsigned char* buf_; // the block being parsed

	int begin_, end_;
	bool eof_;
	int last_char_;
	bool hit_nl_, failed_;
	// blocks are read (and inflated) ahead of the parser by a second thread
	static const int NSLOTS = 4;
	unsigned char* slot_[NSLOTS];
	int slot_len_[NSLOTS];
	bool slot_eof_[NSLOTS];        // the block that marks the end of the input
	bool slot_member_end_[NSLOTS]; // (gz_) the block ends a gzip member whose trailer said slot_crc_ / slot_isize_
	unsigned slot_crc_[NSLOTS], slot_isize_[NSLOTS];
	unsigned long long head_, tail_; // blocks produced / consumed
	unsigned long long crc_done_;    // (gz_) blocks the checksum thread is through with
	int cur_;
	bool stop_;
	bool io_error_ = false; // written by the I/O thread before it publishes the final (empty) block
	std::string io_error_text_;
	std::mutex mu_;
	std::condition_variable cv_;
	std::thread io_;
	std::thread crc_;
};

} // namespace nte_host
