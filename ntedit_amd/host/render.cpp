// render.cpp -- see render.h.  Host-side, single pass over the batch.
#include "render.h"

#include <cmath>
#include <cstring>
#include <string>

namespace nte_host {

using nte::Item;

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

struct ContigState
{
	std::vector<RNode> nodes;
	std::vector<RSub> subs;
	std::vector<char> seq; // private copy once a character changes
	bool terminated = false;
};

// Writes one record the way writeEditsToFile walks the rope (ntedit.cpp:936-1212).
void
write_contig(
    const char* hdr,
    const char* seq,
    const std::vector<RNode>& nodes,
    const std::vector<RSub>& subs,
    FILE* fa,
    FILE* tsv,
    RenderStats* st)
{
	if (fa) {
		fputc('>', fa);
		fputs(hdr, fa);
		fputc('\n', fa);
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
				unsigned char draft_char = (unsigned char)seq[cur.s_pos - ins.size()];
				if (tsv) {
					fprintf(tsv, "%s\t%u\t%c\t+%s\t%d\n", hdr, pos, draft_char, ins.c_str(), num_support);
				}
				st->insertions++;
				ins.clear();
				num_support = -1;
			}
			while (qi < subs.size() && subs[qi].pos <= cur.e_pos) {
				const RSub& s = subs[qi];
				if (tsv) {
					fprintf(tsv, "%s\t%u\t%c\t%c\t%u", hdr, s.pos + 1, s.draft, s.sub, s.support);
					if (s.s1 > 0) {
						fprintf(tsv, "\t%c\t%u", s.a1, s.s1);
					}
					if (s.s2 > 0) {
						fprintf(tsv, "\t%c\t%u", s.a2, s.s2);
					}
					if (s.s3 > 0) {
						fprintf(tsv, "\t%c\t%u", s.a3, s.s3);
					}
					fputc('\n', tsv);
				}
				st->substitutions++;
				qi++;
			}
			if (fa) {
				fwrite(seq + cur.s_pos, 1, (size_t)cur.e_pos - cur.s_pos + 1, fa);
			}
			pos = cur.e_pos + 1;
		} else if (cur.type == 1) {
			ins.push_back((char)cur.c);
			if (num_support == -1) {
				num_support = (int)cur.support;
			}
			if (fa) {
				fputc(cur.c, fa);
			}
		}
		ni++;
		if (ni < nn) {
			const RNode& nx = nodes[ni];
			if (nx.type == 0 && nx.s_pos != pos) {
				if (tsv) {
					fprintf(tsv, "%s\t%u\t%c\t-", hdr, pos, seq[pos]);
					fwrite(seq + pos, 1, (size_t)nx.s_pos - pos, tsv);
					fprintf(tsv, "\t%u\n", nx.support);
				}
				st->deletions++;
			}
		}
	}
	if (fa) {
		fputc('\n', fa);
	}
}

} // namespace

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
    RenderStats* stats)
{
	RenderStats local;
	RenderStats* st = stats ? stats : &local;
	size_t ev = 0;
	ContigState cs;
	for (uint32_t ci = 0; ci < n_contigs; ci++) {
		const char* seq = bases + offsets[ci];
		const uint32_t len = lens[ci];
		cs.nodes.clear();
		cs.subs.clear();
		cs.seq.clear();
		cs.terminated = false;
		RNode root = { 0, 0, len ? len - 1 : 0, 0, 0 };
		cs.nodes.push_back(root);
		uint32_t cover = 0;
		bool any = false;
		while (ev < n_events) {
			uint32_t fc = ev_first[ev];
			if ((size_t)fc * nte::CHUNK_ITEMS + 1 >= arena_items) {
				return -1;
			}
			const Item& hdr = arena[(size_t)fc * nte::CHUNK_ITEMS + 1];
			if (hdr.w[0] != ci) {
				if (hdr.w[0] < ci) {
					return -2; // events must arrive in contig order
				}
				break;
			}
			ev++;
			const uint32_t start = hdr.w[1], cover_end = hdr.w[2];
			if (start < cover) {
				continue; // overtaken by an earlier event's serial run
			}
			cover = cover_end;
			st->events_applied++;
			any = true;
			// walk the chunk chain
			bool first_node = true;
			uint32_t chunk = fc;
			bool first_chunk = true;
			while (chunk != nte::NONE32) {
				if ((size_t)(chunk + 1) * nte::CHUNK_ITEMS > arena_items) {
					return -1;
				}
				const Item* c = arena + (size_t)chunk * nte::CHUNK_ITEMS;
				uint32_t next = c[0].w[0], cnt = c[0].w[1];
				if (cnt > nte::CHUNK_ITEMS) {
					return -3;
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
						return -4;
					}
				}
				first_chunk = false;
				chunk = next;
			}
		}
		const char* out_seq = cs.seq.empty() ? seq : cs.seq.data();
		if (!any) {
			// untouched contig: header + sequence + newline
			if (fa) {
				fputc('>', fa);
				fputs(names[ci], fa);
				fputc('\n', fa);
				fwrite(seq, 1, len, fa);
				fputc('\n', fa);
			}
			continue;
		}
		write_contig(names[ci], out_seq, cs.nodes, cs.subs, fa, tsv, st);
	}
	return 0;
}

} // namespace nte_host
