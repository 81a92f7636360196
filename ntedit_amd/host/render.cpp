// render.cpp -- see render.h.  Host-side, single pass over the batch.
#include "render.h"

#include <cctype>
#include <cmath>
#include <cstring>
#include <ctime>
#include <map>
#include <string>
#include <zlib.h>

namespace nte_host {

using nte::Item;

// ------------------------------------------------------------ -l annotations
class Annotations
{
  public:
	std::map<std::string, std::string> m;
	// "^" + INFO of the variant id, or "^NA" (clinvar[id].empty() -> NA in the reference)
	void put(FILE* vcf, const std::string& id) const
	{
		auto it = m.find(id);
		fputc('^', vcf);
		if (it != m.end() && !it->second.empty()) {
			fputs(it->second.c_str(), vcf);
		} else {
			fputs("NA", vcf);
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
put_annot(FILE* vcf, const Annotations* a, const std::string& id)
{
	if (a) {
		a->put(vcf, id);
	} else {
		fputs("^NA", vcf);
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

// the substitution line of _variants.vcf (ntedit.cpp:986-1162)
void
write_vcf_substitution(FILE* vcf, const std::string& H, const RSub& s, const RenderOptions& opt, bool is_edit)
{
	std::string base(1, (char)s.sub);
	std::string support = std::to_string(s.support);
	const char D = (char)toupper(s.draft);
	const std::string pos1 = std::to_string(s.pos + 1);
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
	fprintf(vcf, "%s\t%u\t.\t%c\t%s\t.\tPASS\tAD=%s", H.c_str(), s.pos + 1, s.draft, base.c_str(), support.c_str());
	for (const std::string& id : ids) {
		put_annot(vcf, opt.annot, id);
	}
	fprintf(vcf, "\tGT\t%s\n", genotype);
}

// Writes one record the way writeEditsToFile walks the rope (ntedit.cpp:936-1212).
void
write_contig(
    const char* hdr,
    const char* seq,
    const std::vector<RNode>& nodes,
    const std::vector<RSub>& subs,
    FILE* fa,
    FILE* tsv,
    RenderStats* st,
    FILE* vcf,
    const RenderOptions& opt)
{
	const std::string H(hdr);
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
				if (vcf) {
					// ntedit.cpp:954-977
					const char D = (char)toupper(draft_char);
					fprintf(vcf, "%s\t%u\t.\t%c\t%c%s\t.\tPASS\tAD=%d", hdr, pos, draft_char, draft_char, ins.c_str(), num_support);
					put_annot(vcf, opt.annot, H + ">" + D + std::to_string(pos) + D + upper(ins.data(), ins.size()));
					fputs("\tGT\t1/1\n", vcf);
				}
				st->insertions++;
				ins.clear();
				num_support = -1;
			}
			while (qi < subs.size() && subs[qi].pos <= cur.e_pos) {
				const RSub& s = subs[qi];
				const bool is_edit = !(opt.snv && s.draft == s.sub); // "snv_mode_no_edit" in the reference
				if (vcf) {
					write_vcf_substitution(vcf, H, s, opt, is_edit);
				}
				if (tsv && is_edit) {
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
				if (is_edit) {
					st->substitutions++;
				}
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
				if (vcf && pos > 0) {
					// ntedit.cpp:1184-1208
					const size_t dl = (size_t)(nx.s_pos - pos) + 1;
					fprintf(vcf, "%s\t%u\t.\t", hdr, pos);
					fwrite(seq + pos - 1, 1, dl, vcf);
					fprintf(vcf, "\t%c\t.\tPASS\tAD=%u", seq[pos - 1], nx.support);
					put_annot(vcf, opt.annot,
					          H + ">" + upper(seq + pos - 1, dl) + std::to_string(pos) + (char)toupper((unsigned char)seq[pos - 1]));
					fputs("\tGT\t1/1\n", vcf);
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
	const RenderOptions opt = opt_in ? *opt_in : RenderOptions();
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
		write_contig(names[ci], out_seq, cs.nodes, cs.subs, fa, tsv, st, vcf, opt);
	}
	return 0;
}

} // namespace nte_host
