// render.cpp -- see render.h.  Host-side.  Every contig is rendered independently (worker
// threads) into text buffers plus a list of spans of the draft; the calling thread writes
// them to the three streams in input order.
#include "render.h"

#include <cctype>
#include <cmath>
#include <cstring>
#include <ctime>
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <thread>
#include <string>
#include <cerrno>
#include <chrono>
#include <cstdlib>
#include <sys/uio.h>
#include <unistd.h>
#include <zlib.h>

namespace nte_host {

using nte::Item;

// ------------------------------------------------------------ -l annotations
class Annotations
{
  public:
	std::map<std::string, std::string> m;
	// "^" + INFO of the variant id, or "^NA" (clinvar[id].empty() -> NA in the reference)
	void put(std::string& vcf, const std::string& id) const
	{
		auto it = m.find(id);
		vcf.push_back('^');
		if (it != m.end() && !it->second.empty()) {
			vcf += it->second.c_str();
		} else {
			vcf += "NA";
		}
	}
};

Annotations*
annotations_load(const char* path)
{
	gzFile f = gzopen(path, "r");
	if (!f) {
		return nullptr;
	}
	Annotations* a = new Annotations();
	std::string line;
	char buf[1 << 16];
	auto flush = [&]() {
		// vcf_entry_to_map (ntedit.cpp:2261-2274): >= 8 tab-separated fields
		std::string tok[8];
		size_t nt = 0, start = 0;
		for (size_t i = 0; i <= line.size() && nt < 8; i++) {
			if (i == line.size() || line[i] == '\t') {
				tok[nt++] = line.substr(start, i - start);
				start = i + 1;
			}
		}
		if (nt >= 8) {
			a->m[tok[0] + ">" + tok[3] + tok[1] + tok[4]] = tok[7];
		}
		line.clear();
	};
	while (gzgets(f, buf, sizeof buf)) {
		line += buf;
		if (!line.empty() && line.back() == '\n') {
			line.pop_back();
			flush();
		}
	}
	if (!line.empty()) {
		flush();
	}
	gzclose(f);
	return a;
}

void
annotations_free(Annotations* a)
{
	delete a;
}

static void
put_annot(std::string& vcf, const Annotations* a, const std::string& id)
{
	if (a) {
		a->put(vcf, id);
	} else {
		vcf += "^NA";
	}
}

static std::string
upper(const char* s, size_t n)
{
	std::string r(s, n);
	for (char& c : r) {
		c = (char)toupper((unsigned char)c);
	}
	return r;
}

// decimal text of an unsigned / signed integer (what %u / %d print)
static void
put_u(std::string& o, uint64_t v)
{
	char tmp[24];
	int n = 0;
	do {
		tmp[n++] = (char)('0' + v % 10);
		v /= 10;
	} while (v);
	while (n) {
		o.push_back(tmp[--n]);
	}
}

static void
put_i(std::string& o, int64_t v)
{
	if (v < 0) {
		o.push_back('-');
		put_u(o, (uint64_t)(-v));
	} else {
		put_u(o, (uint64_t)v);
	}
}

namespace {

struct RNode
{
	int type;
	uint32_t s_pos, e_pos;
	uint8_t c;
	uint32_t support;
};

struct RSub
{
	uint32_t pos;
	uint8_t draft, sub;
	uint32_t support;
	uint8_t a1, a2, a3;
	uint32_t s1, s2, s3;
};

// one piece of a contig's _edited.fa record: a span of the (possibly modified) draft, or
// of the record's own small text (header, inserted bases, newlines)
struct Piece
{
	const char* p; // nullptr: `off` indexes ContigOut::text
	size_t off;
	size_t n;
};

// everything one work unit (a run of consecutive contigs) contributes to the three output streams
struct ContigOut
{
	// scratch of the contig being rendered
	std::vector<RNode> nodes;
	std::vector<RSub> subs;
	std::vector<char> seq; // private copy once a character changes
	bool terminated = false;
	// output of the unit
	std::vector<Piece> fa;
	std::string text, tsv, vcf;
	std::deque<std::vector<char>> seq_keep; // modified contigs the pieces point into
	RenderStats st;
	int rc = 0;
	bool ready = false;
	size_t fa_bytes = 0;              // bytes of the unit's FASTA pieces so far
	std::vector<uint64_t> sizes;      // 4 per job of the unit: contig, fa / tsv / vcf bytes (RenderOptions::out_sizes)
	std::vector<ntedit_hip_edit> edits; // RenderOptions::edits
	std::string edit_pool;
	// the unit's FASTA pieces copied into one stretch by the render thread (multi-threaded rendering): the writer then
	// hands the kernel a few megabytes per call instead of a thousand spans of a kilobase
	char* flat = nullptr;
	size_t flat_n = 0, flat_cap = 0;

	void reset()
	{
		flat_n = 0;
		fa.clear();
		text.clear();
		tsv.clear();
		vcf.clear();
		seq_keep.clear();
		st = RenderStats();
		rc = 0;
		fa_bytes = 0;
		sizes.clear();
		edits.clear();
		edit_pool.clear();
	}
	void begin_contig()
	{
		nodes.clear();
		subs.clear();
		seq.clear();
		terminated = false;
	}
	void fa_text(const char* s, size_t n)
	{
		if (!fa.empty() && !fa.back().p && fa.back().off + fa.back().n == text.size()) {
			fa.back().n += n;
		} else {
			Piece pc = { nullptr, text.size(), n };
			fa.push_back(pc);
		}
		text.append(s, n);
		fa_bytes += n;
	}
	void fa_char(char c) { fa_text(&c, 1); }
	void fa_span(const char* s, size_t n)
	{
		Piece pc = { s, 0, n };
		fa.push_back(pc);
		fa_bytes += n;
	}
};

// the substitution line of _variants.vcf (ntedit.cpp:986-1162)
void
write_vcf_substitution(std::string& vcf, const std::string& H, const RSub& s, const RenderOptions& opt, bool is_edit, uint64_t off)
{
	std::string base(1, (char)s.sub);
	std::string support = std::to_string(s.support);
	const char D = (char)toupper(s.draft);
	const std::string pos1 = std::to_string((uint64_t)s.pos + 1 + off);
	std::vector<std::string> ids;
	ids.push_back(H + ">" + D + pos1 + D);
	if (is_edit) {
		ids.push_back(H + ">" + D + pos1 + (char)toupper((unsigned char)base[0]));
	}
	uint8_t ab[3];
	uint32_t as[3];
	int na = 0;
	if (s.s1 > 0) {
		ab[na] = s.a1;
		as[na++] = s.s1;
	}
	if (s.s2 > 0) {
		ab[na] = s.a2;
		as[na++] = s.s2;
	}
	if (s.s3 > 0) {
		ab[na] = s.a3;
		as[na++] = s.s3;
	}
	uint32_t curr_best = 0;
	char best_alt_base = '1';
	const char* genotype = "1/1";
	if (na) {
		if (opt.snv) {
			if (!is_edit) {
				for (int i = 0; i < na; i++) {
					if (as[i] > curr_best) {
						curr_best = as[i];
						best_alt_base = (char)ab[i];
					}
				}
				base = std::string(1, best_alt_base);
				ids.push_back(H + ">" + D + pos1 + (char)toupper((unsigned char)best_alt_base));
				support += "," + std::to_string(curr_best);
				genotype = "0/1";
			} else {
				bool ref = false;
				for (int i = 0; i < na; i++) {
					if (s.draft == ab[i]) {
						curr_best = as[i];
						ref = true;
						break;
					}
					if (as[i] > curr_best) {
						curr_best = as[i];
						best_alt_base = (char)ab[i];
					}
				}
				if (ref) {
					support = std::to_string(curr_best) + "," + support;
					genotype = "0/1";
				} else {
					genotype = "1/2";
					support += "," + std::to_string(curr_best);
					base += ",";
					base += best_alt_base;
					ids.push_back(H + ">" + D + pos1 + (char)toupper((unsigned char)best_alt_base));
				}
			}
		} else {
			for (int i = 0; i < na; i++) {
				if (s.draft == ab[i]) {
					continue;
				}
				if (as[i] > curr_best) {
					curr_best = as[i];
					best_alt_base = (char)ab[i];
				}
			}
			genotype = "1/2";
			support += "," + std::to_string(curr_best);
			base += ",";
			base += best_alt_base;
			ids.push_back(H + ">" + D + pos1 + (char)toupper((unsigned char)best_alt_base));
		}
	}
	// "%s\t%u\t.\t%c\t%s\t.\tPASS\tAD=%s"
	vcf += H;
	vcf.push_back('\t');
	put_u(vcf, (uint64_t)s.pos + 1 + off);
	vcf += "\t.\t";
	vcf.push_back((char)s.draft);
	vcf.push_back('\t');
	vcf += base.c_str(); // (%s: an alternate base of 0 ends the string, as in the reference)
	vcf += "\t.\tPASS\tAD=";
	vcf += support;
	for (const std::string& id : ids) {
		put_annot(vcf, opt.annot, id);
	}
	vcf += "\tGT\t";
	vcf += genotype;
	vcf.push_back('\n');
}

// Renders one record the way writeEditsToFile walks the rope (ntedit.cpp:936-1212).
// sg (may be nullptr): the record is one segment of a contig that was cut for multi-GPU sharding -- its
// positions are shifted by sg->pos_offset, a segment that is not the first has no header line, one that is
// not the last no closing newline (the caller has already taken the look-ahead halo off the last node).
// want_edits: every TSV row is also handed on as an ntedit_hip_edit record (entry index ci).
void
write_contig(const char* hdr, const char* seq, ContigOut& o, bool want_fa, bool want_tsv, bool want_vcf, const RenderOptions& opt,
             const ntedit_hip_segment* sg, uint32_t ci, bool want_edits, uint32_t part_flags = 0)
{
	const std::vector<RNode>& nodes = o.nodes;
	const std::vector<RSub>& subs = o.subs;
	RenderStats* st = &o.st;
	std::string& tsv = o.tsv;
	std::string& vcf = o.vcf;
	const std::string H(hdr);
	const uint64_t off = sg ? sg->pos_offset : 0;
	const uint32_t sflags = (sg ? sg->flags : 0) | part_flags; // (part_flags: a part of a contig rendered by itself, SubRange)
	auto add_edit = [&](uint8_t kind, uint32_t dpos, const char* bases, size_t nb, uint32_t support, uint8_t draft, uint8_t nbase,
	                    const RSub* alt) {
		ntedit_hip_edit e;
		memset(&e, 0, sizeof e);
		e.contig = ci;
		e.draft_pos = (uint32_t)(dpos + off);
		e.bases_off = (uint32_t)o.edit_pool.size();
		e.len = (uint16_t)(nb > 0xFFFF ? 0xFFFF : nb);
		e.support = (uint16_t)(support > 0xFFFF ? 0xFFFF : support);
		e.kind = kind;
		e.draft_base = draft;
		e.new_base = nbase;
		if (alt) {
			if (alt->s1 > 0) {
				e.alt_base[e.n_alt] = alt->a1;
				e.alt_support[e.n_alt++] = (uint8_t)alt->s1;
			}
			if (alt->s2 > 0) {
				e.alt_base[e.n_alt] = alt->a2;
				e.alt_support[e.n_alt++] = (uint8_t)alt->s2;
			}
			if (alt->s3 > 0) {
				e.alt_base[e.n_alt] = alt->a3;
				e.alt_support[e.n_alt++] = (uint8_t)alt->s3;
			}
		}
		if (bases && nb) {
			o.edit_pool.append(bases, nb);
		}
		o.edits.push_back(e);
	};
	if (want_fa && !(sflags & NTEDIT_SEG_NO_HEADER)) {
		o.fa_char('>');
		o.fa_text(H.data(), H.size());
		o.fa_char('\n');
	}
	size_t qi = 0;
	std::string ins;
	int num_support = -1;
	uint32_t pos = 0;
	size_t ni = 0;
	const size_t nn = nodes.size();
	while (ni < nn && nodes[ni].type != -1) {
		const RNode& cur = nodes[ni];
		if (cur.type == 0) {
			if (!ins.empty()) {
				// (U4, oracle/ntedit_oracle.c) more inserted bases than bases in front of them: the
				// reference's .at() throws there; the row gets 'N'
				unsigned char draft_char = cur.s_pos >= ins.size() ? (unsigned char)seq[cur.s_pos - ins.size()] : (unsigned char)'N';
				if (want_tsv) {
					// "%s\t%u\t%c\t+%s\t%d\n" (%s stops at an embedded NUL, like the reference's c_str())
					tsv += H;
					tsv.push_back('\t');
					put_u(tsv, pos + off);
					tsv.push_back('\t');
					tsv.push_back((char)draft_char);
					tsv += "\t+";
					tsv += ins.c_str();
					tsv.push_back('\t');
					put_i(tsv, num_support);
					tsv.push_back('\n');
				}
				if (want_vcf) {
					// ntedit.cpp:954-977
					const char D = (char)toupper(draft_char);
					vcf += H;
					vcf.push_back('\t');
					put_u(vcf, pos + off);
					vcf += "\t.\t";
					vcf.push_back((char)draft_char);
					vcf.push_back('\t');
					vcf.push_back((char)draft_char);
					vcf += ins.c_str();
					vcf += "\t.\tPASS\tAD=";
					put_i(vcf, num_support);
					put_annot(vcf, opt.annot, H + ">" + D + std::to_string(pos + off) + D + upper(ins.data(), ins.size()));
					vcf += "\tGT\t1/1\n";
				}
				if (want_edits) {
					add_edit(NTEDIT_EDIT_INS, pos, ins.data(), ins.size(), (uint32_t)(num_support < 0 ? 0 : num_support), draft_char, 0, nullptr);
				}
				st->insertions++;
				ins.clear();
				num_support = -1;
			}
			while (qi < subs.size() && subs[qi].pos <= cur.e_pos) {
				const RSub& s = subs[qi];
				const bool is_edit = !(opt.snv && s.draft == s.sub); // "snv_mode_no_edit" in the reference
				if (want_vcf) {
					write_vcf_substitution(vcf, H, s, opt, is_edit, off);
				}
				if (want_edits) {
					add_edit(is_edit ? NTEDIT_EDIT_SUB : NTEDIT_EDIT_SNV_KEPT, s.pos, nullptr, 1, s.support, s.draft, s.sub, &s);
				}
				if (want_tsv && is_edit) {
					tsv += H;
					tsv.push_back('\t');
					put_u(tsv, (uint64_t)s.pos + 1 + off);
					tsv.push_back('\t');
					tsv.push_back((char)s.draft);
					tsv.push_back('\t');
					tsv.push_back((char)s.sub);
					tsv.push_back('\t');
					put_u(tsv, s.support);
					if (s.s1 > 0) {
						tsv.push_back('\t');
						tsv.push_back((char)s.a1);
						tsv.push_back('\t');
						put_u(tsv, s.s1);
					}
					if (s.s2 > 0) {
						tsv.push_back('\t');
						tsv.push_back((char)s.a2);
						tsv.push_back('\t');
						put_u(tsv, s.s2);
					}
					if (s.s3 > 0) {
						tsv.push_back('\t');
						tsv.push_back((char)s.a3);
						tsv.push_back('\t');
						put_u(tsv, s.s3);
					}
					tsv.push_back('\n');
				}
				if (is_edit) {
					st->substitutions++;
				}
				qi++;
			}
			if (want_fa) {
				o.fa_span(seq + cur.s_pos, (size_t)cur.e_pos - cur.s_pos + 1);
			}
			pos = cur.e_pos + 1;
		} else if (cur.type == 1) {
			ins.push_back((char)cur.c);
			if (num_support == -1) {
				num_support = (int)cur.support;
			}
			if (want_fa) {
				o.fa_char((char)cur.c);
			}
		}
		ni++;
		if (ni < nn) {
			const RNode& nx = nodes[ni];
			if (nx.type == 0 && nx.s_pos != pos) {
				if (want_tsv) {
					// "%s\t%u\t%c\t-" + the deleted bases + "\t%u\n"
					tsv += H;
					tsv.push_back('\t');
					put_u(tsv, pos + off);
					tsv.push_back('\t');
					tsv.push_back(seq[pos]);
					tsv += "\t-";
					tsv.append(seq + pos, (size_t)nx.s_pos - pos);
					tsv.push_back('\t');
					put_u(tsv, nx.support);
					tsv.push_back('\n');
				}
				if (want_vcf && pos > 0) {
					// ntedit.cpp:1184-1208
					const size_t dl = (size_t)(nx.s_pos - pos) + 1;
					vcf += H;
					vcf.push_back('\t');
					put_u(vcf, pos + off);
					vcf += "\t.\t";
					vcf.append(seq + pos - 1, dl);
					vcf.push_back('\t');
					vcf.push_back(seq[pos - 1]);
					vcf += "\t.\tPASS\tAD=";
					put_u(vcf, nx.support);
					put_annot(vcf, opt.annot,
					          H + ">" + upper(seq + pos - 1, dl) + std::to_string(pos + off) + (char)toupper((unsigned char)seq[pos - 1]));
					vcf += "\tGT\t1/1\n";
				}
				if (want_edits) {
					add_edit(NTEDIT_EDIT_DEL, pos, seq + pos, (size_t)nx.s_pos - pos, nx.support, (uint8_t)seq[pos], 0, nullptr);
				}
				st->deletions++;
			}
		}
	}
	if (want_fa && !(sflags & NTEDIT_SEG_NO_NEWLINE)) {
		o.fa_char('\n');
	}
}

struct BatchView
{
	const Item* arena;
	size_t arena_items;
	const uint32_t* ev_first;
	const char* bases;
	const uint64_t* offsets;
	const uint32_t* lens;
	const char* const* names;
	bool want_fa, want_tsv, want_vcf, want_edits;
	RenderOptions opt;
};

// A part of a contig that is rendered by itself (large contigs: one render thread per contig would leave the others idle
// behind a chromosome).  The contig is cut in front of events that start where the serial run is certainly clean and
// untouched: no earlier event's run -- applied or not -- comes within PART_MARGIN bases of the cut (render_batch picks the
// cuts).  A part = the draft positions [pa, pb) and the events that start in them; positions stay the contig's.
struct SubRange
{
	uint32_t pa, pb; // pb = the contig's length for its last part
	bool first, last;
};

// events [ev, ev_end) of contig ci -> o
void
render_contig(const BatchView& v, uint32_t ci, size_t ev, size_t ev_end, ContigOut& cs, const SubRange* sr = nullptr)
{
	cs.begin_contig();
	const char* seq = v.bases + v.offsets[ci];
	const uint32_t len = v.lens[ci];
	const uint32_t pa = sr ? sr->pa : 0;
	const uint32_t part_flags = sr ? ((sr->first ? 0u : (uint32_t)NTEDIT_SEG_NO_HEADER) | (sr->last ? 0u : (uint32_t)NTEDIT_SEG_NO_NEWLINE)) : 0u;
	const Item* arena = v.arena;
	const size_t arena_items = v.arena_items;
	const ntedit_hip_segment* sg = v.opt.segments ? &v.opt.segments[ci] : nullptr;
	const uint32_t halo = sg ? sg->halo : 0;
	if (sg && ((sg->flags & NTEDIT_SEG_SKIP) || halo > len)) {
		if (!(sg->flags & NTEDIT_SEG_SKIP)) {
			cs.rc = -8;
		}
		return; // (an entry that was superseded by a re-run: no output at all)
	}
	RNode root = { 0, pa, len ? len - 1 : 0, 0, 0 };
	cs.nodes.push_back(root);
	// (a part's copy of modified bases: its own positions and a margin in front of them, addressed by contig position)
	const uint32_t ca = sr ? (pa >= 64 ? pa - 64 : 0) : 0, cb = sr && !sr->last ? sr->pb : len;
	uint32_t cover = 0;
	bool any = false;
	for (; ev < ev_end; ev++) {
		const uint32_t fc = v.ev_first[ev];
		if (fc == nte::NONE32) {
			continue;
		}
		const Item& hdr = arena[(size_t)fc * nte::CHUNK_ITEMS + 1]; // (bounds checked by the caller)
		const uint32_t start = hdr.w[1], cover_end = hdr.w[2];
		if (start < cover) {
			continue; // overtaken by an earlier event's serial run
		}
		if (hdr.w[3] & nte::EV_UNFINISHED) {
			cs.rc = -6; // a parked event must have been re-run before it can be applied
			return;
		}
		cover = cover_end;
		cs.st.events_applied++;
		any = true;
		// walk the chunk chain
		bool first_node = true;
		uint32_t chunk = fc;
		bool first_chunk = true;
		while (chunk != nte::NONE32) {
			if ((size_t)(chunk + 1) * nte::CHUNK_ITEMS > arena_items) {
				cs.rc = -1;
				return;
			}
			const Item* c = arena + (size_t)chunk * nte::CHUNK_ITEMS;
			uint32_t next = c[0].w[0], cnt = c[0].w[1];
			if (cnt > nte::CHUNK_ITEMS) {
				cs.rc = -3;
				return;
			}
			for (uint32_t i = first_chunk ? 2 : 1; i < cnt; i++) {
				const Item& it = c[i];
				switch (it.w[0] & 0xFF) {
				case nte::TAG_NODE: {
					if (cs.terminated) {
						break;
					}
					RNode n;
					n.type = (int)(int8_t)((it.w[0] >> 8) & 0xFF);
					n.c = (uint8_t)((it.w[0] >> 16) & 0xFF);
					n.s_pos = it.w[1];
					n.e_pos = it.w[2];
					n.support = it.w[3];
					if (first_node) {
						// the event's rope starts with (its view of) the open node
						// that currently ends the contig's rope
						first_node = false;
						RNode& open = cs.nodes.back();
						if (n.type == 0 && n.s_pos == 0) {
							n.s_pos = open.s_pos;
							n.support = open.support;
						}
						open = n;
						if (n.type == -1) {
							cs.terminated = true;
						}
					} else {
						cs.nodes.push_back(n);
						if (n.type == -1) {
							cs.terminated = true;
						}
					}
					break;
				}
				case nte::TAG_SUB: {
					RSub s;
					s.draft = (uint8_t)((it.w[0] >> 8) & 0xFF);
					s.sub = (uint8_t)((it.w[0] >> 16) & 0xFF);
					s.support = (it.w[0] >> 24) & 0xFF;
					s.pos = it.w[1];
					s.a1 = (uint8_t)(it.w[2] & 0xFF);
					s.s1 = (it.w[2] >> 8) & 0xFF;
					s.a2 = (uint8_t)((it.w[2] >> 16) & 0xFF);
					s.s2 = (it.w[2] >> 24) & 0xFF;
					s.a3 = (uint8_t)(it.w[3] & 0xFF);
					s.s3 = (it.w[3] >> 8) & 0xFF;
					cs.subs.push_back(s);
					break;
				}
				case nte::TAG_MOD: {
					if (cs.seq.empty()) {
						cs.seq.assign(seq + ca, seq + cb);
					}
					if (it.w[1] >= ca && it.w[1] < cb) {
						cs.seq[it.w[1] - ca] = (char)((it.w[0] >> 8) & 0xFF);
					} else if (sr && it.w[1] < len) {
						cs.rc = -9; // a cut that an event reaches across: render_batch's margin is meant to rule that out
						return;
					}
					break;
				}
				default:
					cs.rc = -4;
					return;
				}
			}
			first_chunk = false;
			chunk = next;
		}
	}
	if (halo) {
		// The last `halo` bases of the entry are look-ahead room that belongs to the next segment of the
		// contig.  The cut is only valid if the serial run was clean again in front of it: every applied
		// event ended at or before the cut, the rope ends in the open position node, nothing behind the
		// cut was touched.
		RNode& last = cs.nodes.back();
		if (cover > len - halo || cs.terminated || last.type != 0 || last.e_pos != len - 1 || last.s_pos >= len - halo) {
			cs.rc = -7;
			return;
		}
		last.e_pos = len - 1 - halo;
	}
	if (sr && !sr->last) {
		// the part ends where the next one begins: in the open node, behind every applied run
		RNode& last = cs.nodes.back();
		if (cover > sr->pb || cs.terminated || last.type != 0 || last.e_pos != len - 1 || last.s_pos > sr->pb) {
			cs.rc = -9;
			return;
		}
		last.e_pos = sr->pb - 1; // (s_pos == pb: an empty node, zero bytes)
	}
	const char* out_seq = seq;
	if (!cs.seq.empty()) {
		// the record's pieces will point into the modified copy: park it with the unit
		cs.seq_keep.emplace_back(std::move(cs.seq));
		cs.seq.clear();
		out_seq = cs.seq_keep.back().data() - ca; // (addressed by contig position; only [ca, cb) is ever read)
	}
	if (!any) {
		// untouched contig (or part): header + sequence + newline
		if (v.want_fa) {
			if ((!sg || !(sg->flags & NTEDIT_SEG_NO_HEADER)) && !(part_flags & NTEDIT_SEG_NO_HEADER)) {
				cs.fa_char('>');
				cs.fa_text(v.names[ci], strlen(v.names[ci]));
				cs.fa_char('\n');
			}
			const uint32_t end = sr && !sr->last ? sr->pb : len - halo;
			cs.fa_span(seq + pa, end - pa);
			if ((!sg || !(sg->flags & NTEDIT_SEG_NO_NEWLINE)) && !(part_flags & NTEDIT_SEG_NO_NEWLINE)) {
				cs.fa_char('\n');
			}
		}
		return;
	}
	write_contig(v.names[ci], out_seq, cs, v.want_fa, v.want_tsv, v.want_vcf, v.opt, sg, ci, v.want_edits, part_flags);
}

// Buffers for the flattened FASTA output of the render units, kept across render_batch calls: a fresh 8 MB buffer per
// slot and call is two thousand page faults each.
class FlatPool
{
  public:
	// best fit (first fit handed a 300 MB buffer to a 1 MB unit while the next large unit allocated another)
	char* take(size_t bytes, size_t* cap)
	{
		{
			std::lock_guard<std::mutex> lk(mu_);
			size_t best = bufs_.size();
			for (size_t i = 0; i < bufs_.size(); i++) {
				if (bufs_[i].second >= bytes && (best == bufs_.size() || bufs_[i].second < bufs_[best].second)) {
					best = i;
				}
			}
			if (best < bufs_.size()) {
				char* p = bufs_[best].first;
				*cap = bufs_[best].second;
				held_ -= bufs_[best].second;
				bufs_[best] = bufs_.back();
				bufs_.pop_back();
				return p;
			}
		}
		const size_t want = bytes + bytes / 4 + (1u << 20);
		char* p = (char*)malloc(want);
		*cap = p ? want : 0;
		return p;
	}
	// kept for the next unit while the pool holds fewer than 96 buffers and less than 2 GiB; the rest goes back to the system
	void give(char* p, size_t cap)
	{
		if (!p) {
			return;
		}
		std::lock_guard<std::mutex> lk(mu_);
		if (bufs_.size() < 96 && held_ + cap <= (size_t(2) << 30)) {
			bufs_.emplace_back(p, cap);
			held_ += cap;
		} else {
			free(p);
		}
	}

  private:
	std::mutex mu_;
	std::vector<std::pair<char*, size_t>> bufs_;
	size_t held_ = 0;
};
FlatPool g_flat_pool;

// copies the unit's FASTA pieces into o.flat (render thread); false: no memory -- the writer gathers the pieces itself
bool
flatten_fa(ContigOut& o)
{
	if (o.fa_bytes > o.flat_cap) {
		g_flat_pool.give(o.flat, o.flat_cap);
		o.flat = g_flat_pool.take(o.fa_bytes, &o.flat_cap);
		if (!o.flat) {
			o.flat_cap = 0;
			return false;
		}
	}
	char* d = o.flat;
	for (const Piece& pc : o.fa) {
		if (pc.n) {
			memcpy(d, pc.p ? pc.p : o.text.data() + pc.off, pc.n);
			d += pc.n;
		}
	}
	o.flat_n = (size_t)(d - o.flat);
	return true;
}

// _changes.tsv and _variants.vcf written by a thread of their own while the calling thread writes _edited.fa: three
// files, three inode locks -- the FASTA stream (nine tenths of the bytes) no longer waits for the other two.
class SideWriter
{
  public:
	SideWriter(FILE* tsv, FILE* vcf)
	  : tsv_(tsv)
	  , vcf_(vcf)
	{}
	// false: no thread could be started -- the caller writes the small streams itself
	bool start()
	{
		try {
			th_ = std::thread([this]() { run(); });
		} catch (...) {
			return false;
		}
		return true;
	}
	void push(std::string& tsv, std::string& vcf)
	{
		Item it;
		it.tsv.swap(tsv);
		it.vcf.swap(vcf);
		{
			// (bounded: with -s 1 the VCF strings would pile up if that file were slower than the FASTA stream)
			std::unique_lock<std::mutex> lk(mu_);
			cv_room_.wait(lk, [&]() { return q_.size() < MAX_QUEUED; });
			q_.push_back(std::move(it));
		}
		cv_.notify_one();
	}
	// everything queued is in the streams' buffers; != 0: a write failed (disk full, ...)
	int finish()
	{
		if (th_.joinable()) {
			{
				std::lock_guard<std::mutex> lk(mu_);
				done_ = true;
			}
			cv_.notify_one();
			th_.join();
		}
		return failed_.load() ? -5 : 0;
	}
	~SideWriter() { (void)finish(); }

  private:
	static constexpr size_t MAX_QUEUED = 256;
	struct Item
	{
		std::string tsv, vcf;
	};
	void run()
	{
		for (;;) {
			Item it;
			{
				std::unique_lock<std::mutex> lk(mu_);
				cv_.wait(lk, [&]() { return done_ || !q_.empty(); });
				if (q_.empty()) {
					return;
				}
				it = std::move(q_.front());
				q_.pop_front();
			}
			cv_room_.notify_one();
			if (tsv_ && !it.tsv.empty() && fwrite(it.tsv.data(), 1, it.tsv.size(), tsv_) != it.tsv.size()) {
				failed_.store(true);
			}
			if (vcf_ && !it.vcf.empty() && fwrite(it.vcf.data(), 1, it.vcf.size(), vcf_) != it.vcf.size()) {
				failed_.store(true);
			}
		}
	}
	FILE* tsv_;
	FILE* vcf_;
	std::mutex mu_;
	std::condition_variable cv_, cv_room_;
	std::deque<Item> q_;
	bool done_ = false;
	std::atomic<bool> failed_{ false };
	std::thread th_;
};

int
emit_contig(ContigOut& o, FILE* fa, FILE* tsv, FILE* vcf, RenderStats* st, const RenderOptions& opt, uint32_t first_contig, SideWriter* side = nullptr)
{
	(void)first_contig;
	for (size_t i = 0; opt.out_sizes && i + 3 < o.sizes.size(); i += 4) {
		// (contig, fa, tsv, vcf bytes): added up, the parts of a contig that was cut come in several units
		uint64_t* d = opt.out_sizes + (size_t)o.sizes[i] * 3;
		d[0] += o.sizes[i + 1];
		d[1] += o.sizes[i + 2];
		d[2] += o.sizes[i + 3];
	}
	if (opt.edits && !o.edits.empty()) {
		const size_t base = opt.edit_pool->size();
		opt.edit_pool->append(o.edit_pool);
		const size_t at = opt.edits->size();
		opt.edits->insert(opt.edits->end(), o.edits.begin(), o.edits.end());
		for (size_t i = at; i < opt.edits->size(); i++) {
			(*opt.edits)[i].bases_off += (uint32_t)base;
		}
	}
	if (fa && o.flat_n) {
		// (flattened by its render thread: one stretch, a few megabytes)
		fflush(fa);
		const int fd = fileno(fa);
		size_t done = 0;
		while (done < o.flat_n) {
			const ssize_t w = write(fd, o.flat + done, o.flat_n - done);
			if (w < 0) {
				if (errno == EINTR) {
					continue;
				}
				return -5;
			}
			done += (size_t)w;
		}
	} else if (fa && !o.fa.empty()) {
		// the record is a gather of draft spans: hand them to the kernel as they are instead
		// of copying everything through the stream's buffer first
		fflush(fa);
		const int fd = fileno(fa);
		std::vector<iovec> iov;
		iov.reserve(o.fa.size() < 1024 ? o.fa.size() : 1024);
		size_t i = 0;
		while (i < o.fa.size()) {
			iov.clear();
			size_t want = 0;
			while (i < o.fa.size() && iov.size() < 1024) {
				const Piece& pc = o.fa[i++];
				if (pc.n) {
					iovec v;
					v.iov_base = const_cast<char*>(pc.p ? pc.p : o.text.data() + pc.off);
					v.iov_len = pc.n;
					iov.push_back(v);
					want += pc.n;
				}
			}
			size_t first = 0;
			while (want) {
				const ssize_t w = writev(fd, iov.data() + first, (int)(iov.size() - first));
				if (w < 0) {
					if (errno == EINTR) {
						continue;
					}
					return -5;
				}
				want -= (size_t)w;
				size_t left = (size_t)w;
				while (left && first < iov.size()) { // (a short write: resume inside the vector)
					if (left >= iov[first].iov_len) {
						left -= iov[first].iov_len;
						first++;
					} else {
						iov[first].iov_base = (char*)iov[first].iov_base + left;
						iov[first].iov_len -= left;
						left = 0;
					}
				}
			}
		}
	}
	if (side) {
		// (the small streams go to the side writer, in unit order: the unit's strings are taken out of the slot)
		side->push(o.tsv, o.vcf);
	} else {
		if (tsv && !o.tsv.empty() && fwrite(o.tsv.data(), 1, o.tsv.size(), tsv) != o.tsv.size()) {
			return -5;
		}
		if (vcf && !o.vcf.empty() && fwrite(o.vcf.data(), 1, o.vcf.size(), vcf) != o.vcf.size()) {
			return -5;
		}
	}
	st->events_applied += o.st.events_applied;
	st->substitutions += o.st.substitutions;
	st->insertions += o.st.insertions;
	st->deletions += o.st.deletions;
	return 0;
}

} // namespace

// The serial-order filter alone (what render_contig does before it renders): per entry, where the run of
// the last applied event ended.
int
cover_ends(const Item* arena, size_t arena_items, const uint32_t* ev_first, size_t n_events, uint32_t n_contigs, uint32_t* out)
{
	for (uint32_t i = 0; i < n_contigs; i++) {
		out[i] = 0;
	}
	for (size_t ev = 0; ev < n_events; ev++) {
		const uint32_t fc = ev_first[ev];
		if (fc == nte::NONE32) {
			continue;
		}
		if ((size_t)fc * nte::CHUNK_ITEMS + 1 >= arena_items) {
			return -1;
		}
		const Item& h = arena[(size_t)fc * nte::CHUNK_ITEMS + 1];
		if (h.w[0] >= n_contigs) {
			return -2;
		}
		if (h.w[1] < out[h.w[0]]) {
			continue; // overtaken by an earlier event's serial run
		}
		out[h.w[0]] = h.w[2];
	}
	return 0;
}

// Would render_contig accept entry i as a segment whose last halos[i] bases are look-ahead room?  The same
// predicate, from the node stream alone (no rendering): every applied event ended at or before the cut, the rope was
// not terminated and ends in the open position node, which starts in front of the cut.  halos[i] == 0: always yes.
int
cuts_ok(const Item* arena, size_t arena_items, const uint32_t* ev_first, size_t n_events, uint32_t n_contigs, const uint32_t* lens, const uint32_t* halos,
        uint8_t* ok)
{
	struct St
	{
		uint32_t cover = 0;
		bool terminated = false;
		int type = 0;
		uint32_t s_pos = 0, e_pos = 0;
	};
	std::vector<St> st(n_contigs);
	for (uint32_t i = 0; i < n_contigs; i++) {
		st[i].e_pos = lens[i] ? lens[i] - 1 : 0;
	}
	for (size_t ev = 0; ev < n_events; ev++) {
		const uint32_t fc = ev_first[ev];
		if (fc == nte::NONE32) {
			continue;
		}
		if ((size_t)fc * nte::CHUNK_ITEMS + 1 >= arena_items) {
			return -1;
		}
		const Item& h = arena[(size_t)fc * nte::CHUNK_ITEMS + 1];
		if (h.w[0] >= n_contigs) {
			return -2;
		}
		St& c = st[h.w[0]];
		if (h.w[1] < c.cover) {
			continue; // overtaken by an earlier event's serial run
		}
		c.cover = h.w[2];
		bool first_node = true, first_chunk = true;
		uint32_t chunk = fc;
		while (chunk != nte::NONE32) {
			if ((size_t)(chunk + 1) * nte::CHUNK_ITEMS > arena_items) {
				return -1;
			}
			const Item* ch = arena + (size_t)chunk * nte::CHUNK_ITEMS;
			const uint32_t next = ch[0].w[0], cnt = ch[0].w[1];
			if (cnt > nte::CHUNK_ITEMS) {
				return -3;
			}
			for (uint32_t i = first_chunk ? 2 : 1; i < cnt; i++) {
				const Item& it = ch[i];
				if ((it.w[0] & 0xFF) != nte::TAG_NODE || c.terminated) {
					continue;
				}
				const int type = (int)(int8_t)((it.w[0] >> 8) & 0xFF);
				uint32_t s_pos = it.w[1];
				if (first_node) {
					first_node = false;
					if (type == 0 && s_pos == 0) {
						s_pos = c.s_pos;
					}
				}
				c.type = type;
				c.s_pos = s_pos;
				c.e_pos = it.w[2];
				if (type == -1) {
					c.terminated = true;
				}
			}
			first_chunk = false;
			chunk = next;
		}
	}
	for (uint32_t i = 0; i < n_contigs; i++) {
		const uint32_t halo = halos[i], len = lens[i];
		const St& c = st[i];
		ok[i] = halo == 0 || (halo <= len && !(c.cover > len - halo || c.terminated || c.type != 0 || c.e_pos != len - 1 || c.s_pos >= len - halo));
	}
	return 0;
}

void
write_vcf_header(FILE* vcf, const char* draft_filename)
{
	time_t now = time(nullptr);
	tm* ltm = localtime(&now);
	fprintf(vcf, "##fileformat=VCFv4.2\n##fileDate=%04d%02d%02d\n##source=ntEdit v2.1.1\n##reference=file:%s\n",
	        1900 + ltm->tm_year, 1 + ltm->tm_mon, ltm->tm_mday, draft_filename);
	fputs("##FORMAT=<ID=GT,Number=1,Type=String,Description=\"Genotype\">\n", vcf);
	fputs("##INFO=<ID=AD,Number=2,Type=Integer,Description=\"Kmer Depth\">\n", vcf);
	fputs("#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tINTEGRATION\n", vcf);
}

void
write_tsv_header(FILE* tsv, uint32_t k, uint32_t jump, bool counting)
{
	// ntedit.cpp:2175-2188
	fputs("ID\tbpPosition+1\tOriginalBase\tNewBase\t", tsv);
	if (counting) {
		fputs("Coverage (max 255)", tsv);
	} else {
		fprintf(tsv, "Support %u-mer (out of %g)", k, std::ceil((double)k / (double)jump));
	}
	const char* alt = counting ? "Coverage" : "Support";
	fprintf(tsv, "\tAlt.Base1\tAlt.%s1\tAlt.Base2\tAlt.%s2\tAlt.Base3\tAlt.%s3\n", alt, alt, alt);
}

int
render_batch(
    const Item* arena,
    size_t arena_items,
    const uint32_t* ev_first,
    size_t n_events,
    const char* bases,
    const uint64_t* offsets,
    const uint32_t* lens,
    const char* const* names,
    uint32_t n_contigs,
    FILE* fa,
    FILE* tsv,
    RenderStats* stats,
    FILE* vcf,
    const RenderOptions* opt_in)
{
	RenderStats local;
	RenderStats* st = stats ? stats : &local;
	BatchView v;
	v.arena = arena;
	v.arena_items = arena_items;
	v.ev_first = ev_first;
	v.bases = bases;
	v.offsets = offsets;
	v.lens = lens;
	v.names = names;
	v.want_fa = fa != nullptr;
	v.want_tsv = tsv != nullptr;
	v.want_vcf = vcf != nullptr;
	v.opt = opt_in ? *opt_in : RenderOptions();
	v.want_edits = v.opt.edits != nullptr && v.opt.edit_pool != nullptr;

	// events of every contig: [ev_begin[ci], ev_begin[ci + 1]) -- ev_begin[ci] = the first event with output whose header
	// names contig ci or a later one (events without output in front of it stay with the contig before; nobody reads
	// them).  One header read per event: 4.4 M dependent reads per 3 Gbp when done by one thread in front of everything
	// else, so the events are cut into ranges that are scanned concurrently; the ranges' contig changes are merged.
	std::vector<size_t> ev_begin((size_t)n_contigs + 1, 0);
	struct CutCand
	{
		size_t ev;       // the event in front of which the contig may be cut
		uint32_t contig;
		uint32_t pos;    // its start
		uint32_t seen;   // highest run end among the range's earlier events of the contig
	};
	std::vector<std::vector<CutCand>> cand;
	std::vector<std::vector<std::pair<uint32_t, uint32_t>>> range_max;
	std::vector<CutCand> cuts; // accepted, in order
	{
		unsigned P = v.opt.threads ? v.opt.threads : std::thread::hardware_concurrency();
		P = P > 16 ? 16 : (P < 1 ? 1 : P);
		if (n_events < 65536 && v.opt.unit_bases > 1) {
			P = 1; // (tests render with units of one base: their few events are scanned in ranges all the same)
		}
		if (P > n_events) {
			P = n_events ? (unsigned)n_events : 1;
		}
		struct Change
		{
			size_t ev;
			uint32_t contig;
		};
		std::vector<std::vector<Change>> found(P);
		std::vector<int> bad(P, 0);
		const bool split = v.opt.part_margin > 0 && !v.opt.segments && v.opt.threads != 1;
		const uint64_t part_bases = v.opt.unit_bases;
		range_max.assign(P, std::vector<std::pair<uint32_t, uint32_t>>()); // per range: (contig, highest run end seen in it)
		cand.assign(P, std::vector<CutCand>());
		auto scan = [&](unsigned t) {
			const size_t a = n_events * t / P, b = n_events * (t + 1) / P;
			uint32_t prev = 0;
			bool have = false;
			for (size_t ev = a; ev < b; ev++) {
				const uint32_t fc = ev_first[ev];
				if (fc == nte::NONE32) {
					continue; // an event without output
				}
				if ((size_t)fc * nte::CHUNK_ITEMS + 1 >= arena_items) {
					bad[t] = -1;
					return;
				}
				const Item& hd = arena[(size_t)fc * nte::CHUNK_ITEMS + 1];
				const uint32_t c = hd.w[0];
				if (!have || c != prev) {
					if ((have && c < prev) || c >= n_contigs) {
						bad[t] = -2; // events must arrive in contig order
						return;
					}
					found[t].push_back(Change{ ev, c });
					prev = c;
					have = true;
					if (split) {
						range_max[t].emplace_back(c, 0u);
					}
				}
				if (split && lens[c] >= 2 * part_bases) {
					// a place to cut the contig: an event that starts a part's length behind the last cut, with no
					// run of THIS range's earlier events of the contig within the margin (earlier ranges: the merge)
					uint32_t& mx = range_max[t].back().second;
					const uint32_t start = hd.w[1], run_end = hd.w[2];
					std::vector<CutCand>& cc = cand[t];
					const uint64_t last_cut = !cc.empty() && cc.back().contig == c ? cc.back().pos : 0;
					if (start >= last_cut + part_bases && (uint64_t)mx + v.opt.part_margin <= start && (uint64_t)start + part_bases / 2 <= lens[c] &&
					    !(hd.w[3] & (nte::EV_UNFINISHED | nte::EV_TERMINAL))) {
						cc.push_back(CutCand{ ev, c, start, mx });
					}
					mx = run_end > mx ? run_end : mx;
				} else if (split) {
					uint32_t& mx = range_max[t].back().second;
					mx = hd.w[2] > mx ? hd.w[2] : mx;
				}
			}
		};
		if (P == 1) {
			scan(0);
		} else {
			// (a thread that cannot be started -- std::system_error must not cross the C ABI with joinable threads behind
			// it -- leaves its range to this one)
			std::vector<std::thread> th;
			std::vector<unsigned> left;
			for (unsigned t = 1; t < P; t++) {
				try {
					th.emplace_back(scan, t);
				} catch (...) {
					left.push_back(t);
				}
			}
			scan(0);
			for (unsigned t : left) {
				scan(t);
			}
			for (std::thread& t : th) {
				t.join();
			}
		}
		const size_t UNSET = ~(size_t)0;
		std::fill(ev_begin.begin(), ev_begin.end(), UNSET);
		uint32_t last = 0;
		bool any = false;
		for (unsigned t = 0; t < P; t++) {
			if (bad[t]) {
				return bad[t];
			}
			for (const Change& ch : found[t]) {
				if (any && ch.contig < last) {
					return -2;
				}
				if (!any || ch.contig != last) {
					ev_begin[ch.contig] = ch.ev;
				}
				last = ch.contig;
				any = true;
			}
		}
		ev_begin[n_contigs] = n_events;
		for (size_t ci = n_contigs; ci-- > 0;) {
			if (ev_begin[ci] == UNSET) {
				ev_begin[ci] = ev_begin[ci + 1]; // a contig without events
			}
		}
		// cuts: a candidate stands if the runs of the EARLIER ranges' events of its contig keep the margin as well, and if
		// it is a part's length behind the cut accepted before it (ranges know only their own)
		if (split) {
			uint32_t carry_contig = 0xFFFFFFFFu, carry_max = 0;
			for (unsigned t = 0; t < P; t++) {
				size_t ri = 0;
				uint32_t cur_contig = 0xFFFFFFFFu, before = 0; // run ends of earlier ranges for the contig the range opens with
				for (const CutCand& cc : cand[t]) {
					while (ri < range_max[t].size() && range_max[t][ri].first != cc.contig) {
						ri++;
					}
					if (cc.contig != cur_contig) {
						cur_contig = cc.contig;
						before = (ri == 0 && carry_contig == cc.contig) ? carry_max : 0;
					}
					const uint64_t last_cut = !cuts.empty() && cuts.back().contig == cc.contig ? cuts.back().pos : 0;
					if ((uint64_t)before + v.opt.part_margin <= cc.pos && cc.pos >= last_cut + part_bases) {
						cuts.push_back(cc);
					}
				}
				// what this range leaves behind for the next one
				if (!range_max[t].empty()) {
					const std::pair<uint32_t, uint32_t>& lastc = range_max[t].back();
					uint32_t m = lastc.second;
					if (range_max[t].size() == 1 && carry_contig == lastc.first) {
						m = m > carry_max ? m : carry_max;
					}
					carry_contig = lastc.first;
					carry_max = m;
				}
			}
		}
	}

	// work units: runs of consecutive contigs of about a megabase (one hand-over, one write and one set of buffers per
	// unit, not per contig: fragmented assemblies have millions of contigs) -- or ONE part of a contig that was cut
	struct Job
	{
		uint32_t ci;
		size_t ev_a, ev_b;
		SubRange sr;
		bool whole;
	};
	std::vector<Job> jobs;
	std::vector<uint32_t> unit_begin; // into jobs
	{
		uint64_t acc = 0;
		uint32_t cnt = 0;
		size_t qi = 0;
		for (uint32_t ci = 0; ci < n_contigs; ci++) {
			size_t q1 = qi;
			while (q1 < cuts.size() && cuts[q1].contig == ci) {
				q1++;
			}
			if (q1 > qi) {
				// parts: [0, cut 0), [cut 0, cut 1), ..., [last cut, len)
				uint32_t pa = 0;
				size_t ea = ev_begin[ci];
				for (size_t q = qi; q <= q1; q++) {
					Job j;
					j.ci = ci;
					j.whole = false;
					j.ev_a = ea;
					j.ev_b = q < q1 ? cuts[q].ev : ev_begin[ci + 1];
					j.sr.pa = pa;
					j.sr.pb = q < q1 ? cuts[q].pos : lens[ci];
					j.sr.first = q == qi;
					j.sr.last = q == q1;
					unit_begin.push_back((uint32_t)jobs.size());
					jobs.push_back(j);
					pa = j.sr.pb;
					ea = j.ev_b;
				}
				qi = q1;
				acc = v.opt.unit_bases; // (the next contig opens a unit of its own)
				continue;
			}
			if (jobs.empty() || acc >= v.opt.unit_bases || cnt >= 8192) {
				unit_begin.push_back((uint32_t)jobs.size());
				acc = 0;
				cnt = 0;
			}
			Job j;
			j.ci = ci;
			j.whole = true;
			j.ev_a = ev_begin[ci];
			j.ev_b = ev_begin[ci + 1];
			j.sr = SubRange{ 0, lens[ci], true, true };
			jobs.push_back(j);
			acc += lens[ci];
			cnt++;
		}
		unit_begin.push_back((uint32_t)jobs.size());
	}
	const uint32_t n_units = (uint32_t)unit_begin.size() - 1;
	auto render_unit = [&](uint32_t u, ContigOut& o) {
		o.reset();
		for (uint32_t ji = unit_begin[u]; ji < unit_begin[u + 1] && !o.rc; ji++) {
			const Job& j = jobs[ji];
			const size_t f0 = o.fa_bytes, t0 = o.tsv.size(), v0 = o.vcf.size();
			render_contig(v, j.ci, j.ev_a, j.ev_b, o, j.whole ? nullptr : &j.sr);
			if (v.opt.out_sizes) {
				o.sizes.push_back(j.ci);
				o.sizes.push_back(o.fa_bytes - f0);
				o.sizes.push_back(o.tsv.size() - t0);
				o.sizes.push_back(o.vcf.size() - v0);
			}
		}
	};

	unsigned T = v.opt.threads;
	if (T == 0) {
		T = std::thread::hardware_concurrency();
		if (T > 8) {
			T = 8;
		}
	}
	if (T > n_units) {
		T = n_units;
	}
	if (T <= 1) {
		ContigOut o;
		for (uint32_t u = 0; u < n_units; u++) {
			render_unit(u, o);
			if (o.rc) {
				return o.rc;
			}
			if (int e = emit_contig(o, fa, tsv, vcf, st, v.opt, unit_begin[u])) {
				return e;
			}
		}
		return 0;
	}

	// workers render units (claimed in input order) into a ring of slots; this thread
	// writes slot after slot, in input order
	const uint32_t W = 2 * T + 2;
	std::vector<ContigOut> slots(W);
	std::atomic<uint32_t> next_u(0);
	std::mutex mu;
	std::condition_variable cv_ready, cv_free;
	uint32_t written = 0;
	bool abort = false;
	auto worker = [&]() {
		for (;;) {
			const uint32_t u = next_u.fetch_add(1);
			if (u >= n_units) {
				return;
			}
			{
				std::unique_lock<std::mutex> lk(mu);
				cv_free.wait(lk, [&]() { return abort || u < written + W; });
				if (abort) {
					return;
				}
			}
			ContigOut& o = slots[u % W];
			render_unit(u, o);
			if (fa && !o.rc && o.fa_bytes) {
				(void)flatten_fa(o);
			}
			{
				std::lock_guard<std::mutex> lk(mu);
				o.ready = true;
			}
			cv_ready.notify_all();
		}
	};
	std::vector<std::thread> pool;
	for (unsigned t = 0; t < T; t++) {
		pool.emplace_back(worker);
	}
	int rc = 0;
	const bool timing = getenv("NTEDIT_HIP_DEBUG") != nullptr;
	double s_wait = 0, s_emit = 0;
	// (a FASTA stream next to at least one of the small ones: those get a writer of their own)
	std::unique_ptr<SideWriter> side;
	if (fa && (tsv || vcf)) {
		side.reset(new SideWriter(tsv, vcf));
		if (!side->start()) {
			side.reset(); // (no thread to be had: the small streams are written inline)
		}
	}
	SideWriter* side_p = side.get();
	for (uint32_t u = 0; u < n_units; u++) {
		ContigOut& o = slots[u % W];
		const auto tw0 = std::chrono::steady_clock::now();
		{
			std::unique_lock<std::mutex> lk(mu);
			cv_ready.wait(lk, [&]() { return o.ready; });
		}
		const auto tw1 = std::chrono::steady_clock::now();
		if (o.rc) {
			rc = o.rc;
			break;
		}
		if ((rc = emit_contig(o, fa, tsv, vcf, st, v.opt, unit_begin[u], side_p))) {
			break;
		}
		if (timing) {
			s_wait += std::chrono::duration<double>(tw1 - tw0).count();
			s_emit += std::chrono::duration<double>(std::chrono::steady_clock::now() - tw1).count();
		}
		{
			std::lock_guard<std::mutex> lk(mu);
			o.ready = false;
			written = u + 1;
		}
		cv_free.notify_all();
	}
	{
		std::lock_guard<std::mutex> lk(mu);
		abort = true; // (no-op after a complete run: every worker has left its loop)
	}
	cv_free.notify_all();
	for (std::thread& t : pool) {
		t.join();
	}
	for (ContigOut& o : slots) {
		g_flat_pool.give(o.flat, o.flat_cap);
		o.flat = nullptr;
		o.flat_cap = 0;
	}
	if (side) {
		const int src = side->finish(); // (everything queued is in the streams' buffers before the caller closes them)
		if (!rc) {
			rc = src;
		}
	}
	if (timing) {
		fprintf(stderr, "[ntedit_hip] render: %u units (%zu contig cuts) on %u threads, writer waited %.3f s for units, wrote for %.3f s\n", n_units, cuts.size(), T, s_wait, s_emit);
	}
	return rc;
}

} // namespace nte_host
