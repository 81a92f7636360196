// fasta_map.h -- whole-file, multi-threaded FASTA ingest for plain multi-FASTA files, uncompressed or BGZF
// (bgzip: gzip members of <= 64 KiB that carry their own compressed size, so they can be found without inflating
// and inflated independently -- on all threads, where one zlib stream manages 0.35 Gbases/s).
//
// The streaming reader (fasta.h) parses with one thread: memchr + one copy per line, ~0.18 s per Gbase, and is
// the critical path of the `ntedit` binary end to end once the GPU does a Gbase in 60 ms.  A plain FASTA file can be
// taken apart in parallel: the file is mapped, T threads find the record starts ('>' at the start of a line) in T
// regions of it, and then measure / copy whole records concurrently straight into the batch buffer.
// Record semantics are the reference's (kseq, lib/kseq.h:176-215 as used at ntedit.cpp:2223-2230); files that use
// anything the simple rules below do not cover -- ordinary (single-stream) gzip, FASTQ ('@' / '+' at a line start), CR line ends, NUL bytes,
// text in front of the first '>' -- are refused (ok() == false) and go through the streaming reader, which restates
// kseq character by character.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace nte_host {

class FastaMap
{
  public:
	FastaMap(const char* path, unsigned threads, unsigned inflate_threads = 0);
	~FastaMap();
	FastaMap(const FastaMap&) = delete;
	FastaMap& operator=(const FastaMap&) = delete;
	bool ok() const { return ok_; }
	size_t records() const { return recs_.size(); }
	// name [+ " " + comment], exactly what FastaReader::next() hands out
	std::string header(size_t i) const;
	// sequence lengths of records [first, first + count), measured by `threads` threads (cached)
	void measure(size_t first, size_t count);
	uint64_t length(size_t i) const { return recs_[i].len; }
	// copies the sequences of records idx[0..n) to dst[j] (each length(idx[j]) bytes), concurrently
	void copy(const size_t* idx, char* const* dst, size_t n) const;
	// bases [start, start + n) of record i to dst (one thread: the line breaks in front of `start` are walked, ~10 GB/s)
	void copy_range(size_t i, uint64_t start, uint64_t n, char* dst) const;

  private:
	struct Rec
	{
		uint64_t start;   // offset of '>'
		uint64_t seq;     // offset of the first byte behind the header line
		uint64_t end;     // offset of the next record's '>' (or the file size)
		uint64_t len;     // sequence bytes (~0 = not measured yet)
	};
	bool inflate_bgzf(); // data_/size_ (the mapped file) -> an anonymous mapping holding the inflated text
	const char* data_ = nullptr;
	uint64_t size_ = 0;
	int fd_ = -1;
	bool ok_ = false;
	unsigned threads_ = 1;
	unsigned inflate_threads_ = 1;
	std::vector<Rec> recs_;
};

} // namespace nte_host
