// gunzip.h -- streaming gzip (RFC 1952) / DEFLATE (RFC 1951) decoder for single-stream .gz drafts.
//
// The reference reads a gzipped draft through zlib's gzread (lib/kseq.h:44-50, KSEQ_INIT(gzFile, gzread) at
// ntedit.cpp:25); zlib's inflate decodes one symbol per table walk and tops out at a third of a Gbase/s, which
// makes the inflate thread the slowest stage of the whole polishing run on the reference demo's own input format.
// This decoder is written for that stream shape (long dynamic-Huffman blocks, four hot literals, short matches):
// a 64-bit bit buffer refilled without branches, a run table that yields the literals in front of a match together
// with the match's length in one look-up, matches copied eight bytes at a time.  It produces the output block by block
// (FastaReader's 4 MiB slots) and keeps the 32 KiB window between calls.
//
// What it returns is what gzread would: concatenated members are decoded one after the other, bytes after the last
// member that do not start another one are ignored; a damaged or truncated stream fails.  The member's CRC-32 and
// length are handed to the caller (member_end()), which checks them on another thread.
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>

namespace nte_host {

class Gunzip
{
  public:
	static constexpr size_t WINDOW = 32768;
	// room the caller keeps writable beyond `want` bytes of a read() (a match may overrun the request)
	static constexpr size_t SLACK = 320;

	// in_bytes: size of the buffer the compressed file is read through (tests make it small: every refill is a seam)
	explicit Gunzip(size_t in_bytes = 4u << 20);
	~Gunzip();
	Gunzip(const Gunzip&) = delete;
	Gunzip& operator=(const Gunzip&) = delete;

	// true: the file is open and starts with the gzip magic (anything else: use another reader)
	bool open(const char* path);
	// inflates into dst until `want` bytes or more have been produced, a member ends, or the stream fails.
	// dst[-WINDOW .. -1] must be writable (the window is copied there) and dst[0 .. want + SLACK) too.
	// Returns the bytes produced; 0 without member_end(): the end of the input, or failed().
	size_t read(unsigned char* dst, size_t want);
	// after a read(): the bytes just returned were the last of a gzip member; its trailer fields
	bool member_end() const { return member_end_; }
	uint32_t member_crc() const { return member_crc_; }
	uint32_t member_isize() const { return member_isize_; }
	bool failed() const { return failed_; }
	const std::string& error() const { return error_; }

  private:
	enum Stage
	{
		ST_HEADER,  // at a gzip member header (or the end of the file)
		ST_BLOCK,   // at a DEFLATE block header
		ST_STORED,  // inside a stored block (stored_left_ bytes to copy)
		ST_HUFFMAN, // inside a Huffman block (tables built)
		ST_TRAILER, // after the final block of a member
		ST_END
	};
	bool fill_input_();
	bool grow_input_();
	bool fail_(const char* what);
	bool parse_header_();
	bool parse_block_();
	void refill_();
	size_t avail_in_() const { return (size_t)(in_end_ - in_); }

	size_t inbuf_size_, in_low_;
	int fd_;
	bool file_eof_;
	unsigned char* inbuf_;
	const unsigned char* in_;
	const unsigned char* in_end_; // true end of the data in inbuf_ (zero padding follows)
	uint64_t bits_;
	unsigned nbits_;
	Stage stage_;
	bool final_block_;
	size_t stored_left_;
	uint64_t member_out_; // bytes of the current member produced so far
	bool any_member_;
	bool member_end_;
	uint32_t member_crc_, member_isize_;
	bool failed_;
	std::string error_;
	unsigned char* window_; // the last WINDOW bytes produced (valid: min(WINDOW, member_out_))
	uint32_t* litlen_;
	uint32_t* dist_;
	unsigned run_bits_;                     // index bits of the run table of the current block
	size_t block_bytes_, prev_block_bytes_; // output of the current / the previous Huffman block
};

} // namespace nte_host
