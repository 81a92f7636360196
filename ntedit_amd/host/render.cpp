// render.cpp -- see render.h.  Host-side.  Every contig is rendered independently (worker
// threads) into text buffers plus a list of spans of the draft; the calling thread writes
// them to the three streams in input order.
#include "render.h"

#include <cctype>
#include <cmath>
#include <cstring>
#include <ctime>
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
	std::vector<uint64_t> sizes;      // 3 per contig of the unit: fa / tsv / vcf bytes (RenderOptions::out_sizes)
	std::vector<ntedit_hip_edit> edits; // RenderOptions::edits
	std::string edit_pool;

	void reset()
	{
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
             const ntedit_hip_segment* sg, uint32_t ci, bool want_edits)
{
	const std::vector<RNode>& nodes = o.nodes;
	const std::vector<RSub>& subs = o.subs;
	RenderStats* st = &o.st;
	std::string& tsv = o.tsv;
	std::string& vcf = o.vcf;
	const std::string H(hdr);
	const uint64_t off = sg ? sg->pos_offset : 0;
	const uint32_t sflags = sg ? sg->flags : 0;
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

// events [ev, ev_end) of contig ci -> o
void
render_contig(const BatchView& v, uint32_t ci, size_t ev, size_t ev_end, ContigOut& cs)
{
	cs.begin_contig();
	const char* seq = v.bases + v.offsets[ci];
	const uint32_t len = v.lens[ci];
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
	RNode root = { 0, 0, len ? len - 1 : 0, 0, 0 };
	cs.nodes.push_back(root);
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
						cs.seq.assign(seq, seq + len);
					}
					if (it.w[1] < len) {
						cs.seq[it.w[1]] = (char)((it.w[0] >> 8) & 0xFF);
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
	const char* out_seq = seq;
	if (!cs.seq.empty()) {
		// the record's pieces will point into the modified copy: park it with the unit
		cs.seq_keep.emplace_back(std::move(cs.seq));
		cs.seq.clear();
		out_seq = cs.seq_keep.back().data();
	}
	if (!any) {
		// untouched contig: header + sequence + newline
		if (v.want_fa) {
			if (!sg || !(sg->flags & NTEDIT_SEG_NO_HEADER)) {
				cs.fa_char('>');
				cs.fa_text(v.names[ci], strlen(v.names[ci]));
				cs.fa_char('\n');
			}
			cs.fa_span(seq, len - halo);
			if (!sg || !(sg->flags & NTEDIT_SEG_NO_NEWLINE)) {
				cs.fa_char('\n');
			}
		}
		return;
	}
	write_contig(v.names[ci], out_seq, cs, v.want_fa, v.want_tsv, v.want_vcf, v.opt, sg, ci, v.want_edits);
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
	if (opt.out_sizes && !o.sizes.empty()) {
		memcpy(opt.out_sizes + (size_t)first_contig * 3, o.sizes.data(), o.sizes.size() * sizeof(uint64_t));
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
	if (fa && !o.fa.empty()) {
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

	// events of every contig: [ev_begin[ci], ev_begin[ci + 1])
	std::vector<size_t> ev_begin((size_t)n_contigs + 1, 0);
	{
		size_t ev = 0;
		for (uint32_t ci = 0; ci < n_contigs; ci++) {
			ev_begin[ci] = ev;
			while (ev < n_events) {
				const uint32_t fc = ev_first[ev];
				if (fc == nte::NONE32) {
					ev++; // an event without output
					continue;
				}
				if ((size_t)fc * nte::CHUNK_ITEMS + 1 >= arena_items) {
					return -1;
				}
				const uint32_t c = arena[(size_t)fc * nte::CHUNK_ITEMS + 1].w[0];
				if (c != ci) {
					if (c < ci) {
						return -2; // events must arrive in contig order
					}
					break;
				}
				ev++;
			}
		}
		ev_begin[n_contigs] = ev;
	}

	// work units: runs of consecutive contigs of about a megabase (one hand-over, one gather write and one
	// set of buffers per unit, not per contig: fragmented assemblies have millions of contigs)
	std::vector<uint32_t> unit_begin;
	{
		uint64_t acc = 0;
		uint32_t cnt = 0;
		for (uint32_t ci = 0; ci < n_contigs; ci++) {
			if (ci == 0 || acc >= v.opt.unit_bases || cnt >= 8192) {
				unit_begin.push_back(ci);
				acc = 0;
				cnt = 0;
			}
			acc += lens[ci];
			cnt++;
		}
		unit_begin.push_back(n_contigs);
	}
	const uint32_t n_units = (uint32_t)unit_begin.size() - 1;
	auto render_unit = [&](uint32_t u, ContigOut& o) {
		o.reset();
		for (uint32_t ci = unit_begin[u]; ci < unit_begin[u + 1] && !o.rc; ci++) {
			const size_t f0 = o.fa_bytes, t0 = o.tsv.size(), v0 = o.vcf.size();
			render_contig(v, ci, ev_begin[ci], ev_begin[ci + 1], o);
			if (v.opt.out_sizes) {
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
	if (side) {
		const int src = side->finish(); // (everything queued is in the streams' buffers before the caller closes them)
		if (!rc) {
			rc = src;
		}
	}
	if (timing) {
		fprintf(stderr, "[ntedit_hip] render: %u units on %u threads, writer waited %.3f s for units, wrote for %.3f s\n", n_units, T, s_wait, s_emit);
	}
	return rc;
}

} // namespace nte_host
