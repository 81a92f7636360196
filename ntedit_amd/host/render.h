// render.h -- host side of the hot path's output: applies the edit records
// produced by the event machine to the draft and writes _edited.fa and
// _changes.tsv exactly as the reference's writeEditsToFile does
// (ntedit.cpp:925-1213, headers 2175-2188).
#pragma once
#include "../csrc/nte_common.h"
#include "../../include/ntedit_hip.h" // ntedit_hip_segment, ntedit_hip_edit (plain C structs)

#include <cstdio>
#include <string>
#include <cstdint>
#include <vector>

namespace nte_host {

struct RenderStats
{
	uint64_t events_applied = 0;
	uint64_t substitutions = 0;
	uint64_t insertions = 0;
	uint64_t deletions = 0;
};

// -l annotation map (ntedit.cpp:2261-2274,2524-2562): "CHROM>REF POS ALT" -> INFO
class Annotations;
Annotations* annotations_load(const char* path); // plain or gzipped VCF; nullptr if unreadable
void annotations_free(Annotations* a);

struct RenderOptions
{
	bool snv = false;               // -s 1: "no edit" substitution records go to the VCF only
	const Annotations* annot = nullptr;
	unsigned threads = 0;           // work units rendered concurrently (0 = up to 8, 1 = in the calling thread)
#ifndef NTE_RENDER_UNIT_BASES
#define NTE_RENDER_UNIT_BASES (1u << 20)
#endif
	unsigned unit_bases = NTE_RENDER_UNIT_BASES; // a work unit = consecutive contigs of about this many bases (and the size of a large contig's parts)
	// > 0: contigs of several units are rendered in parts, cut in front of events that no earlier event's run comes within
	// this many bases of (k + max deletions + slack: what an event may touch behind the end of its run); 0: whole contigs
	unsigned part_margin = 0;
	// multi-GPU sharding: entry i is a segment of a larger contig (nullptr: every entry is a whole contig)
	const ntedit_hip_segment* segments = nullptr;
	uint64_t* out_sizes = nullptr;  // 3 per entry: bytes the entry added to the fa / tsv / vcf streams
	// every _changes.tsv row (and, with -s 1, the VCF-only records) as POD records, in output order
	std::vector<ntedit_hip_edit>* edits = nullptr;
	std::string* edit_pool = nullptr; // inserted / deleted bases the records point into
};

// arena:    host copy of the chunk arena
// ev_first: first chunk of every event, ordered by global start position (contig order, then
//           position); nte::NONE32 entries (events without output) are skipped
// fa / tsv: may be nullptr (that stream is skipped)
// returns 0, or: -7 a segment's cut is not event-free (RenderOptions::segments), -8 bad segment descriptor,
// -6 an event parked by the budget was not re-run, -1..-5 malformed records / write error
int render_batch(
    const nte::Item* arena,
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
    FILE* vcf = nullptr,
    const RenderOptions* opt = nullptr);

// per entry: where the serial run of its last applied event ended (0: no applied event)
int cover_ends(const nte::Item* arena, size_t arena_items, const uint32_t* ev_first, size_t n_events, uint32_t n_contigs, uint32_t* out);
int cuts_ok(const nte::Item* arena, size_t arena_items, const uint32_t* ev_first, size_t n_events, uint32_t n_contigs, const uint32_t* lens,
            const uint32_t* halos, uint8_t* ok);

// ntedit.cpp:2192-2211
void write_vcf_header(FILE* vcf, const char* draft_filename);

void write_tsv_header(FILE* tsv, uint32_t k, uint32_t jump, bool counting);

} // namespace nte_host
