#include "gunzip.h"

#include <cerrno>
#include <cstdlib>
#include <cstring>
#include <fcntl.h>
#include <unistd.h>

namespace nte_host {

namespace {

constexpr size_t INPAD = 64;         // zero bytes kept behind the data: the bit reader loads eight bytes at a time
constexpr size_t IN_LOW = 256 << 10; // the buffer is topped up below this (a member header or a block header fits)

// entry of a decode table: bits 0-4 = code bits to drop, bit 5 end of block, bit 6 literal, bit 7 sub-table,
// bits 8-12 = extra bits (or the sub-table's index bits), bits 16-31 = literal / base value / sub-table offset
constexpr uint32_t F_EOB = 1u << 5, F_LIT = 1u << 6, F_SUB = 1u << 7;
constexpr unsigned LITLEN_ROOT = 11, DIST_ROOT = 8, PRE_ROOT = 7;
constexpr unsigned LITLEN_ENTRIES = (1u << LITLEN_ROOT) + 288 * 16;
constexpr unsigned DIST_ENTRIES = (1u << DIST_ROOT) + 32 * 128;
// the run table: index = the next FAST_BITS bits of the stream, entry = what they decode to as far as the codes lie
// completely inside them: up to four literals, then possibly a length code (bits 0-7 = bits to drop, bits 8-10 =
// literals, bit 11 = a length follows: base in bits 16-24, extra bits in bits 25-27;
// bits 32-63 = the literals in output order).  A draft is four letters and a newline with codes of two or three
// bits and a match every few bases: one look-up yields the literals in front of a match and the match's length,
// where the symbol-by-symbol walk is a dependent table load per base.  The table is filled for every block, so its
// width follows the size of the blocks: FAST_BITS when the previous block was a long one (zlib's are 16 K symbols,
// 80 KB of a draft: 1,024 entries cost 3 % of the block's time), SMALL_FAST_BITS behind a short one.
constexpr unsigned FAST_BITS = 10, SMALL_FAST_BITS = 6;
constexpr unsigned FAST_ENTRIES = 1u << FAST_BITS;
constexpr size_t LONG_BLOCK = 32768;
constexpr uint64_t P_NLIT = 0x700, P_LEN = 0x800;

const uint16_t LEN_BASE[29] = { 3,  4,  5,  6,  7,  8,  9,  10, 11,  13,  15,  17,  19,  23, 27,
	                            31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258 };
const uint8_t LEN_EXTRA[29] = { 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0 };
const uint16_t DIST_BASE[30] = { 1,   2,   3,   4,   5,   7,    9,    13,   17,   25,   33,   49,   65,    97,    129,
	                             193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577 };
const uint8_t DIST_EXTRA[30] = { 0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13 };
const uint8_t PRE_ORDER[19] = { 16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15 };

inline uint64_t
load64(const unsigned char* p)
{
	uint64_t v;
	memcpy(&v, p, 8);
	return v; // (little-endian host)
}

inline unsigned
reverse_bits(unsigned code, unsigned len)
{
	unsigned r = 0;
	for (unsigned i = 0; i < len; i++) {
		r = (r << 1) | ((code >> i) & 1u);
	}
	return r;
}

enum Kind
{
	K_PRE,
	K_LITLEN,
	K_DIST
};

inline uint32_t
payload(Kind kind, unsigned sym, bool* valid)
{
	*valid = true;
	if (kind == K_PRE) {
		return (uint32_t)sym << 16;
	}
	if (kind == K_LITLEN) {
		if (sym < 256) {
			return F_LIT | ((uint32_t)sym << 16);
		}
		if (sym == 256) {
			return F_EOB;
		}
		if (sym > 285) {
			*valid = false; // 286 and 287 take part in the fixed code but never occur
			return 0;
		}
		return ((uint32_t)LEN_BASE[sym - 257] << 16) | ((uint32_t)LEN_EXTRA[sym - 257] << 8);
	}
	if (sym > 29) {
		*valid = false;
		return 0;
	}
	return ((uint32_t)DIST_BASE[sym] << 16) | ((uint32_t)DIST_EXTRA[sym] << 8);
}

// canonical Huffman code of lens[0..n) -> two-level decode table (first level: root bits; codes longer than that
// go through a sub-table per first-level prefix).  false: the lengths are over-subscribed, or incomplete in a way
// zlib rejects too (inftrees.c accepts an incomplete code only if it is a single code of length 1)
bool
build_table(const uint8_t* lens, unsigned n, uint32_t* table, unsigned root, unsigned max_entries, Kind kind)
{
	unsigned count[16] = { 0 };
	for (unsigned i = 0; i < n; i++) {
		count[lens[i]]++;
	}
	memset(table, 0, sizeof(uint32_t) << root);
	if (count[0] == n) {
		return kind != K_PRE; // no codes at all: every lookup is an invalid code
	}
	int left = 1;
	unsigned maxlen = 0;
	for (unsigned l = 1; l <= 15; l++) {
		left = (left << 1) - (int)count[l];
		if (left < 0) {
			return false;
		}
		if (count[l]) {
			maxlen = l;
		}
	}
	if (left > 0 && (kind == K_PRE || maxlen != 1)) {
		return false;
	}
	// symbols in canonical order (by length, then by value)
	uint16_t sorted[288];
	unsigned offs[17];
	offs[1] = 0;
	for (unsigned l = 1; l <= 15; l++) {
		offs[l + 1] = offs[l] + count[l];
	}
	const unsigned n_coded = offs[16];
	{
		unsigned at[16];
		for (unsigned l = 1; l <= 15; l++) {
			at[l] = offs[l];
		}
		for (unsigned i = 0; i < n; i++) {
			if (lens[i]) {
				sorted[at[lens[i]]++] = (uint16_t)i;
			}
		}
	}
	// their codes: consecutive within a length, shifted left by one at every step to the next length
	uint16_t codes[288];
	uint8_t clen[288];
	{
		unsigned code = 0, at = 0;
		for (unsigned l = 1; l <= 15; l++) {
			for (unsigned c = 0; c < count[l]; c++) {
				codes[at] = (uint16_t)code++;
				clen[at++] = (uint8_t)l;
			}
			code <<= 1;
		}
	}
	unsigned next_free = 1u << root;
	unsigned i = 0;
	while (i < n_coded) {
		const unsigned len = clen[i];
		if (len <= root) {
			bool valid;
			const uint32_t pay = payload(kind, sorted[i], &valid);
			if (valid) {
				const uint32_t e = pay | len;
				for (unsigned idx = reverse_bits(codes[i], len); idx < (1u << root); idx += 1u << len) {
					table[idx] = e;
				}
			}
			i++;
			continue;
		}
		// the codes that share this one's first `root` bits are the next ones in canonical order, and the last of
		// them is the longest: it sizes the sub-table
		const unsigned prefix = codes[i] >> (len - root);
		unsigned j = i;
		while (j < n_coded && (unsigned)(codes[j] >> (clen[j] - root)) == prefix) {
			j++;
		}
		const unsigned sub_bits = clen[j - 1] - root;
		if (next_free + (1u << sub_bits) > max_entries) {
			return false;
		}
		uint32_t* sub = table + next_free;
		memset(sub, 0, sizeof(uint32_t) << sub_bits);
		table[reverse_bits(prefix, root)] = F_SUB | ((uint32_t)next_free << 16) | (sub_bits << 8) | root;
		next_free += 1u << sub_bits;
		for (; i < j; i++) {
			bool valid;
			const uint32_t pay = payload(kind, sorted[i], &valid);
			if (!valid) {
				continue;
			}
			const unsigned rest = clen[i] - root;
			const uint32_t e = pay | rest;
			for (unsigned idx = reverse_bits(codes[i], clen[i]) >> root; idx < (1u << sub_bits); idx += 1u << rest) {
				sub[idx] = e;
			}
		}
	}
	return true;
}

// one entry of the run table, from the literal/length table `lt` (an index whose upper bits are zero finds the entry of
// a code that fits into the known bits)
uint64_t
run_entry(const uint32_t* lt, unsigned idx, unsigned width)
{
	unsigned used = 0, cnt = 0;
	uint64_t lits = 0, tail = 0;
	for (;;) {
		const uint32_t e = lt[idx >> used];
		const unsigned l = e & 31;
		if ((e & (F_SUB | F_EOB)) || l == 0 || used + l > width) {
			break;
		}
		if (e & F_LIT) {
			if (cnt == 4) {
				break;
			}
			lits |= (uint64_t)((e >> 16) & 0xff) << (8 * cnt);
			cnt++;
			used += l;
			continue;
		}
		tail = P_LEN | ((uint64_t)(e >> 16) << 16) | ((uint64_t)((e >> 8) & 7) << 25);
		used += l;
		break;
	}
	return (lits << 32) | tail | (cnt << 8) | used;
}

} // namespace

Gunzip::Gunzip(size_t in_bytes)
  : inbuf_size_(in_bytes < 2048 ? 2048 : in_bytes)
  , in_low_(IN_LOW < inbuf_size_ / 2 ? IN_LOW : inbuf_size_ / 2)
  , fd_(-1)
  , file_eof_(false)
  , inbuf_(nullptr)
  , in_(nullptr)
  , in_end_(nullptr)
  , bits_(0)
  , nbits_(0)
  , stage_(ST_HEADER)
  , final_block_(false)
  , stored_left_(0)
  , member_out_(0)
  , any_member_(false)
  , member_end_(false)
  , member_crc_(0)
  , member_isize_(0)
  , failed_(false)
  , window_(nullptr)
  , litlen_(nullptr)
  , dist_(nullptr)
  , run_bits_(FAST_BITS)
  , block_bytes_(0)
  , prev_block_bytes_(LONG_BLOCK)
{
}

Gunzip::~Gunzip()
{
	if (fd_ >= 0) {
		close(fd_);
	}
	free(inbuf_);
	free(window_);
	free(litlen_);
	free(dist_);
}

bool
Gunzip::fail_(const char* what)
{
	failed_ = true;
	if (error_.empty()) {
		error_ = what;
	}
	stage_ = ST_END;
	return false;
}

// tops the input buffer up
bool
Gunzip::fill_input_()
{
	if (file_eof_ || avail_in_() >= in_low_) {
		return true;
	}
	if (in_ > in_end_) {
		return true; // (only at the end of the file)
	}
	// (the whole bytes the bit reader holds stay in front of in_: a stored block or a trailer hands them back)
	const size_t back = (size_t)(nbits_ >> 3) < (size_t)(in_ - inbuf_) ? (size_t)(nbits_ >> 3) : (size_t)(in_ - inbuf_);
	const size_t keep = avail_in_() + back;
	memmove(inbuf_, in_ - back, keep);
	size_t have = keep;
	while (have < inbuf_size_) {
		const ssize_t r = ::read(fd_, inbuf_ + have, inbuf_size_ - have);
		if (r < 0) {
			if (errno == EINTR) {
				continue;
			}
			return fail_("read error");
		}
		if (r == 0) {
			file_eof_ = true;
			break;
		}
		have += (size_t)r;
	}
	in_ = inbuf_ + back;
	in_end_ = inbuf_ + have;
	memset(inbuf_ + have, 0, INPAD);
	return true;
}

bool
Gunzip::open(const char* path)
{
	fd_ = ::open(path, O_RDONLY);
	if (fd_ < 0) {
		return false;
	}
	inbuf_ = (unsigned char*)malloc(inbuf_size_ + INPAD);
	window_ = (unsigned char*)calloc(WINDOW, 1);
	// (the run table lives behind the literal/length table: FAST_ENTRIES 8-byte entries)
	litlen_ = (uint32_t*)malloc(sizeof(uint32_t) * LITLEN_ENTRIES + sizeof(uint64_t) * FAST_ENTRIES + 8);
	dist_ = (uint32_t*)malloc(sizeof(uint32_t) * DIST_ENTRIES);
	if (!inbuf_ || !window_ || !litlen_ || !dist_) {
		return false;
	}
	in_ = in_end_ = inbuf_;
	memset(inbuf_, 0, INPAD);
	if (!fill_input_()) {
		return false;
	}
	return avail_in_() >= 2 && in_[0] == 0x1f && in_[1] == 0x8b;
}

inline void
Gunzip::refill_()
{
	bits_ |= load64(in_) << nbits_;
	in_ += (63 - nbits_) >> 3;
	nbits_ |= 56;
}

// a larger input buffer with more of the file in it (parse_header_: a header field longer than the buffer)
bool
Gunzip::grow_input_()
{
	const size_t off = (size_t)(in_ - inbuf_), have0 = (size_t)(in_end_ - inbuf_);
	const size_t ns = inbuf_size_ * 2;
	// FEXTRA is at most 64 KiB by format; a file name or comment that has no end within megabytes is damage (or hostile):
	// fail instead of pulling the whole file into memory (gzread would skip it byte by byte; nobody names a draft that way)
	if (ns > ((size_t)16 << 20)) {
		return fail_("invalid header (a field of several megabytes)");
	}
	unsigned char* nb = (unsigned char*)realloc(inbuf_, ns + INPAD);
	if (!nb) {
		return fail_("out of memory");
	}
	inbuf_ = nb;
	inbuf_size_ = ns;
	size_t have = have0;
	while (have < inbuf_size_) {
		const ssize_t r = ::read(fd_, inbuf_ + have, inbuf_size_ - have);
		if (r < 0) {
			if (errno == EINTR) {
				continue;
			}
			return fail_("read error");
		}
		if (r == 0) {
			file_eof_ = true;
			break;
		}
		have += (size_t)r;
	}
	in_ = inbuf_ + off;
	in_end_ = inbuf_ + have;
	memset(inbuf_ + have, 0, INPAD);
	return true;
}

// RFC 1952 member header.  At the end of the file, or at bytes that do not start another member (zlib's gzread
// ignores those), the stream is over.
bool
Gunzip::parse_header_()
{
	if (!fill_input_()) {
		return false;
	}
	const size_t av = avail_in_();
	if (av == 0 || (any_member_ && (av < 2 || in_[0] != 0x1f || in_[1] != 0x8b))) {
		stage_ = ST_END;
		return true;
	}
	// A header field (FEXTRA, FNAME, FCOMMENT) may be longer than what is buffered: gzread takes any length, so the
	// parse is repeated with more of the file until it is complete or the file is over.
	const unsigned char* p = in_;
	for (;;) {
	p = in_;
	const unsigned char* const e = in_end_;
	bool short_of_input = false;
	do {
	if ((size_t)(e - p) < 10) {
		short_of_input = true;
		break;
	}
	if (p[0] != 0x1f || p[1] != 0x8b) {
		return fail_("incorrect header check");
	}
	if (p[2] != 8) {
		return fail_("unknown compression method");
	}
	const unsigned flg = p[3];
	if (flg & 0xe0) {
		return fail_("unknown header flags set");
	}
	p += 10;
	if (flg & 4) {
		if (e - p < 2) {
			short_of_input = true;
			break;
		}
		const size_t xlen = p[0] | ((size_t)p[1] << 8);
		p += 2;
		if ((size_t)(e - p) < xlen) {
			short_of_input = true;
			break;
		}
		p += xlen;
	}
	for (unsigned bit = 8; bit <= 16; bit <<= 1) { // FNAME, FCOMMENT
		if (flg & bit) {
			const void* z = memchr(p, 0, (size_t)(e - p));
			if (!z) {
				short_of_input = true;
			break;
			}
			p = (const unsigned char*)z + 1;
		}
	}
	if (flg & 2) {
		if (e - p < 2) {
			short_of_input = true;
			break;
		}
		p += 2; // (header CRC: not checked, as zlib's gzread does not either)
	}
	} while (false);
	if (!short_of_input) {
		break;
	}
	if (file_eof_) {
		return fail_("unexpected end of file");
	}
	if (!grow_input_()) {
		return false;
	}
	}
	in_ = p;
	bits_ = 0;
	nbits_ = 0;
	any_member_ = true;
	member_out_ = 0;
	stage_ = ST_BLOCK;
	return true;
}

// DEFLATE block header: stored length, or the code tables of a Huffman block
bool
Gunzip::parse_block_()
{
	if (!fill_input_()) {
		return false;
	}
	auto need = [&](unsigned n) {
		if (nbits_ < n) {
			refill_();
		}
	};
	auto take = [&](unsigned n) -> unsigned {
		const unsigned v = (unsigned)(bits_ & ((1ull << n) - 1));
		bits_ >>= n;
		nbits_ -= n;
		return v;
	};
	auto overrun = [&]() { return in_ > in_end_ && (size_t)(in_ - in_end_) * 8 > nbits_; };
	need(3);
	final_block_ = take(1) != 0;
	const unsigned type = take(2);
	if (type == 0) {
		// stored: back to a byte boundary, LEN, NLEN
		take(nbits_ & 7);
		in_ -= nbits_ >> 3;
		bits_ = 0;
		nbits_ = 0;
		if (in_ > in_end_ || avail_in_() < 4) {
			return fail_("unexpected end of file");
		}
		const unsigned len = in_[0] | ((unsigned)in_[1] << 8);
		const unsigned nlen = in_[2] | ((unsigned)in_[3] << 8);
		if ((len ^ 0xffffu) != nlen) {
			return fail_("invalid stored block lengths");
		}
		in_ += 4;
		stored_left_ = len;
		stage_ = ST_STORED;
		return true;
	}
	if (type == 3) {
		return fail_("invalid block type");
	}
	uint8_t lens[288 + 32];
	unsigned nlit, ndist;
	if (type == 1) {
		nlit = 288;
		ndist = 32;
		for (unsigned i = 0; i < 288; i++) {
			lens[i] = i < 144 ? 8 : i < 256 ? 9 : i < 280 ? 7 : 8;
		}
		for (unsigned i = 0; i < 32; i++) {
			lens[288 + i] = 5;
		}
	} else {
		need(14);
		nlit = take(5) + 257;
		ndist = take(5) + 1;
		const unsigned ncode = take(4) + 4;
		if (nlit > 286 || ndist > 30) {
			return fail_("too many length or distance symbols");
		}
		uint8_t pre[19] = { 0 };
		for (unsigned i = 0; i < ncode; i++) {
			need(3);
			pre[PRE_ORDER[i]] = (uint8_t)take(3);
		}
		uint32_t pretab[1u << PRE_ROOT];
		if (!build_table(pre, 19, pretab, PRE_ROOT, 1u << PRE_ROOT, K_PRE)) {
			return fail_("invalid code lengths set");
		}
		unsigned i = 0;
		while (i < nlit + ndist) {
			need(7 + 7);
			const uint32_t e = pretab[bits_ & ((1u << PRE_ROOT) - 1)];
			if ((e & 31) == 0) {
				return fail_("invalid code lengths set");
			}
			take(e & 31);
			const unsigned sym = e >> 16;
			if (sym < 16) {
				lens[i++] = (uint8_t)sym;
				continue;
			}
			unsigned rep, val = 0;
			if (sym == 16) {
				if (i == 0) {
					return fail_("invalid bit length repeat");
				}
				val = lens[i - 1];
				rep = 3 + take(2);
			} else if (sym == 17) {
				rep = 3 + take(3);
			} else {
				rep = 11 + take(7);
			}
			if (i + rep > nlit + ndist) {
				return fail_("invalid bit length repeat");
			}
			while (rep--) {
				lens[i++] = (uint8_t)val;
			}
		}
		if (overrun()) {
			return fail_("unexpected end of file");
		}
		if (lens[256] == 0) {
			return fail_("invalid code -- missing end-of-block");
		}
		// the distance lengths follow the literal/length ones directly: move them to their own place
		uint8_t dl[32];
		memcpy(dl, lens + nlit, ndist);
		memcpy(lens + 288, dl, ndist);
	}
	if (!build_table(lens, nlit, litlen_, LITLEN_ROOT, LITLEN_ENTRIES, K_LITLEN)) {
		return fail_("invalid literal/lengths set");
	}
	if (!build_table(lens + 288, ndist, dist_, DIST_ROOT, DIST_ENTRIES, K_DIST)) {
		return fail_("invalid distances set");
	}
	{
		run_bits_ = prev_block_bytes_ >= LONG_BLOCK ? FAST_BITS : SMALL_FAST_BITS;
		uint64_t* const fast = (uint64_t*)(litlen_ + LITLEN_ENTRIES + (LITLEN_ENTRIES & 1));
		for (unsigned idx = 0; idx < (1u << run_bits_); idx++) {
			fast[idx] = run_entry(litlen_, idx, run_bits_);
		}
		block_bytes_ = 0;
	}
	stage_ = ST_HUFFMAN;
	return true;
}

size_t
Gunzip::read(unsigned char* dst, size_t want)
{
	member_end_ = false;
	if (failed_ || stage_ == ST_END) {
		return 0;
	}
	const size_t hist0 = member_out_ < WINDOW ? (size_t)member_out_ : WINDOW;
	memcpy(dst - WINDOW, window_, WINDOW);
	unsigned char* out = dst;
	unsigned char* const out_end = dst + want;
	const uint64_t* const fast = (const uint64_t*)(litlen_ + LITLEN_ENTRIES + (LITLEN_ENTRIES & 1));
	while (out < out_end && !failed_) {
		if (stage_ == ST_HEADER) {
			if (out > dst) {
				break; // (a member's bytes are returned on their own: its CRC belongs to them)
			}
			if (!parse_header_() || stage_ == ST_END) {
				break;
			}
		} else if (stage_ == ST_BLOCK) {
			if (!parse_block_()) {
				break;
			}
		} else if (stage_ == ST_STORED) {
			if (!fill_input_()) {
				break;
			}
			size_t n = stored_left_;
			if (n > (size_t)(out_end - out)) {
				n = (size_t)(out_end - out);
			}
			if (n > avail_in_()) {
				n = avail_in_();
				if (n == 0) {
					fail_("unexpected end of file");
					break;
				}
			}
			memcpy(out, in_, n);
			out += n;
			in_ += n;
			stored_left_ -= n;
			if (stored_left_ == 0) {
				stage_ = final_block_ ? ST_TRAILER : ST_BLOCK;
			}
		} else if (stage_ == ST_HUFFMAN) {
			if (!fill_input_()) {
				break;
			}
			// while more of the file is to come the loop stops 32 bytes short of the buffered data: one pass refills
			// up to three times, each refill moves on by up to seven bytes and loads eight, and none of that may
			// reach the zero padding and take it for data; at the end of the file the loop runs into the padding and
			// the overrun test below tells a complete stream from a truncated one
			const unsigned char* const in_lim = file_eof_ ? in_end_ + 8 : in_end_ - 32;
			const uint32_t* const lt = litlen_;
			const uint32_t* const dt = dist_;
			uint64_t bits = bits_;
			unsigned nbits = nbits_;
			const unsigned char* in = in_;
			bool block_done = false;
			const char* err = nullptr;
#define NTE_REFILL()                                                                                                       \
	do {                                                                                                                   \
		bits |= load64(in) << nbits;                                                                                       \
		in += (63 - nbits) >> 3;                                                                                           \
		nbits |= 56;                                                                                                       \
	} while (0)
#define NTE_DROP(n)                                                                                                        \
	do {                                                                                                                   \
		bits >>= (n);                                                                                                      \
		nbits -= (n);                                                                                                      \
	} while (0)
			// One pass of the loop = the literals in front of a match and the match: up to three look-ups in the run
			// table (33 bits), the length's extra bits, a refill, the distance.
#define NTE_RUN(p)                                                                                                         \
	do {                                                                                                                   \
		p = fast[bits & fmask];                                                                                            \
		memcpy(out, (const unsigned char*)&p + 4, 4);                                                                      \
		out += (p >> 8) & 7;                                                                                               \
		NTE_DROP((unsigned)p & 0xff);                                                                                      \
	} while (0)
			const uint64_t fmask = (1u << run_bits_) - 1;
			unsigned char* const out0 = out;
			while (out < out_end && in <= in_lim) {
				NTE_REFILL();
				uint64_t p;
				NTE_RUN(p);
				if (!(p & P_LEN) && (p & P_NLIT)) {
					NTE_RUN(p);
					if (!(p & P_LEN) && (p & P_NLIT)) {
						NTE_RUN(p);
						if (!(p & P_LEN) && (p & P_NLIT)) {
							continue; // a long run of literals
						}
					}
				}
				unsigned lbase, xl;
				if (p & P_LEN) {
					lbase = (unsigned)(p >> 16) & 0x1ff;
					xl = (unsigned)(p >> 25) & 7;
				} else {
					// neither a literal nor a length that the run table knows: a long code, or the end of the block
					NTE_REFILL();
					uint32_t e = lt[bits & ((1u << LITLEN_ROOT) - 1)];
					if (e & F_SUB) {
						NTE_DROP(LITLEN_ROOT);
						e = lt[(e >> 16) + (bits & ((1u << ((e >> 8) & 31)) - 1))];
					}
					const unsigned l = e & 31;
					if (l == 0) {
						err = "invalid literal/length code";
						break;
					}
					NTE_DROP(l);
					if (e & F_LIT) {
						*out++ = (unsigned char)(e >> 16);
						continue;
					}
					if (e & F_EOB) {
						block_done = true;
						break;
					}
					lbase = e >> 16;
					xl = (e >> 8) & 31;
				}
				const unsigned length = lbase + (unsigned)(bits & ((1u << xl) - 1));
				NTE_DROP(xl);
				if (nbits < 32) { // (a distance code with its extra bits takes up to 28 bits)
					NTE_REFILL();
				}
				uint32_t d = dt[bits & ((1u << DIST_ROOT) - 1)];
				if (d & F_SUB) {
					NTE_DROP(DIST_ROOT);
					d = dt[(d >> 16) + (bits & ((1u << ((d >> 8) & 31)) - 1))];
				}
				const unsigned dl = d & 31;
				if (dl == 0) {
					err = "invalid distance code";
					break;
				}
				NTE_DROP(dl);
				const unsigned xd = (d >> 8) & 31;
				const size_t distance = (d >> 16) + (size_t)(bits & ((1u << xd) - 1));
				NTE_DROP(xd);
				if (distance > hist0 + (size_t)(out - dst)) {
					err = "invalid distance too far back";
					break;
				}
				const unsigned char* src = out - distance;
				unsigned char* o = out;
				out += length;
				if (distance >= 8) {
					// eight bytes at a time, in order: a chunk's source lies before the chunk already written
					memcpy(o, src, 8);
					memcpy(o + 8, src + 8, 8);
					if (length > 16) {
						o += 16;
						src += 16;
						do {
							memcpy(o, src, 8);
							o += 8;
							src += 8;
						} while (o < out);
					}
				} else if (distance == 1) {
					memset(o, *src, length);
				} else {
					do {
						*o++ = *src++;
					} while (o < out);
				}
			}
#undef NTE_RUN
#undef NTE_REFILL
#undef NTE_DROP
			bits_ = bits;
			nbits_ = nbits;
			in_ = in;
			block_bytes_ += (size_t)(out - out0);
			if (err) {
				fail_(err);
				break;
			}
			if (in_ > in_end_ && (size_t)(in_ - in_end_) * 8 > nbits_) {
				fail_("unexpected end of file"); // symbols were decoded from the padding
				break;
			}
			if (block_done) {
				prev_block_bytes_ = block_bytes_;
				stage_ = final_block_ ? ST_TRAILER : ST_BLOCK;
			} else if (out < out_end && file_eof_ && in_ > in_lim) {
				fail_("unexpected end of file");
				break;
			}
		} else if (stage_ == ST_TRAILER) {
			// back to a byte boundary; CRC-32 and ISIZE of the member
			const unsigned drop = nbits_ & 7;
			bits_ >>= drop;
			nbits_ -= drop;
			in_ -= nbits_ >> 3;
			bits_ = 0;
			nbits_ = 0;
			if (in_ > in_end_) {
				fail_("unexpected end of file");
				break;
			}
			if (!fill_input_()) {
				break;
			}
			if (avail_in_() < 8) {
				fail_("unexpected end of file");
				break;
			}
			member_crc_ = in_[0] | ((uint32_t)in_[1] << 8) | ((uint32_t)in_[2] << 16) | ((uint32_t)in_[3] << 24);
			member_isize_ = in_[4] | ((uint32_t)in_[5] << 8) | ((uint32_t)in_[6] << 16) | ((uint32_t)in_[7] << 24);
			in_ += 8;
			member_end_ = true;
			stage_ = ST_HEADER;
			break;
		}
	}
	if (failed_) {
		return 0;
	}
	const size_t n = (size_t)(out - dst);
	member_out_ += n;
	// the window for the next call: the last WINDOW bytes of [dst - WINDOW, out)
	memcpy(window_, out - WINDOW, WINDOW);
	if (member_end_) {
		member_out_ = 0; // (members do not share history)
	}
	return n;
}

} // namespace nte_host
