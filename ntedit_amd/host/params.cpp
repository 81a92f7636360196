#include "params.h"

#include <cstdio>
#include <cstring>

namespace nte_host {

void
params_default(ntedit_hip_params* p)
{
	// ntedit.cpp:99-133
	memset(p, 0, sizeof(*p));
	p->min_contig_len = 100;
	p->max_insertions = 5;
	p->max_deletions = 5;
	p->edit_threshold = 9.0f;
	p->missing_threshold = 5.0f;
	p->edit_ratio = 0.5f;
	p->missing_ratio = 0.5f;
	p->use_ratio = 0;
	p->jump = 3;
	p->mode = 0;
	p->snv = 0;
	p->mask = 0;
	p->min_threshold = 1;
	p->max_threshold = 255;
}

static void
append(char* warn, size_t cap, const char* msg)
{
	if (!warn || !cap) {
		return;
	}
	size_t l = strlen(warn);
	if (l + 1 < cap) {
		snprintf(warn + l, cap - l, "%s", msg);
	}
}

void
params_clamp(ntedit_hip_params* p, char* warn, size_t cap)
{
	if (warn && cap) {
		warn[0] = 0;
	}
	// ntedit.cpp:2411-2413
	if (p->snv) {
		p->max_insertions = 0;
		p->max_deletions = 0;
	}
	// ntedit.cpp:2467-2475: the x/y range test is a conjunction of
	// contradictory terms and can never fire -- x and y are used as given.
	// ntedit.cpp:2478-2483
	if ((p->max_insertions == 0 && p->max_deletions > 0) ||
	    (p->max_insertions == 1 && p->max_deletions > 1)) {
		append(
		    warn,
		    cap,
		    "ntEdit v2.1.1: warning: i and d parameter combination is not possible; d was set to the "
		    "value of i.\n");
		p->max_deletions = p->max_insertions;
	}
	// ntedit.cpp:2485-2493 (the reference prints these without a newline)
	if (p->max_insertions > 5) {
		append(warn, cap, "ntEdit v2.1.1: warning: i parameter too high, adjusting to maximum -i 5");
		p->max_insertions = 5;
	}
	if (p->max_deletions > 10) {
		append(warn, cap, "ntEdit v2.1.1: warning: d parameter too high, adjusting to maximum -d 10");
		p->max_deletions = 10;
	}
}

template<typename F>
static uint32_t
first_count(uint32_t k, F ok)
{
	for (uint32_t c = 0; c <= k + 2; c++) {
		if (ok(c)) {
			return c;
		}
	}
	return 0xFFFFFFFFu;
}

int
make_dev_params(
    const ntedit_hip_params& hp,
    uint32_t k,
    uint32_t hash_num,
    bool secbf,
    nte::DevParams* out,
    bool counting)
{
	static const uint32_t num_tries[nte::MAX_INSERTION + 1] = { 0, 1, 5, 21, 85, 341 }; // ntedit.cpp:172
	if (k < 12 || k > 200 || hash_num == 0 || hash_num > nte::MAX_HASHES || hp.jump == 0 ||
	    hp.max_insertions > nte::MAX_INSERTION || hp.max_deletions > nte::MAX_DELETION || hp.mode < 0 || hp.mode > 2) {
		return NTEDIT_E_ARG;
	}
	nte::DevParams d;
	memset(&d, 0, sizeof d);
	d.k = k;
	d.h = hash_num;
	d.jump = hp.jump;
	// ntedit.cpp:2411-2413: -s 1 switches the indel sweep off
	d.ins_tries = hp.snv ? 0 : num_tries[hp.max_insertions];
	d.max_deletions = hp.snv ? 0 : hp.max_deletions;
	d.mode = (uint32_t)hp.mode;
	d.mask = hp.mask ? 1 : 0;
	d.secbf = secbf ? 1 : 0;
	d.counting = counting ? 1 : 0;
	d.snv = hp.snv ? 1 : 0;
	// ntedit.cpp:2453-2458: -p only exists for counting filters
	d.min_thr = counting ? hp.min_threshold : 1;
	d.max_thr = counting ? hp.max_threshold : 255;
	// ntedit.cpp:2450-2451 (-c is parsed, then overwritten by k*1.5)
	d.insertion_cap = (uint32_t)((float)k * 1.5f);
	const float fk = (float)k;
	const bool ur = hp.use_ratio != 0;
	const float x = hp.missing_threshold, y = hp.edit_threshold;
	const float X = hp.missing_ratio, Y = hp.edit_ratio;
	const uint32_t jump = hp.jump;
	d.thr_missing = first_count(k, [&](uint32_t c) {
		return (!ur && (float)c >= (fk / x)) || (ur && (float)c >= ((fk / jump) * X));
	});
	d.thr_edit = first_count(k, [&](uint32_t c) {
		return (!ur && (float)c >= (fk / y)) || (ur && (float)c >= ((fk / jump)) * Y);
	});
	d.thr_edit_del = first_count(k, [&](uint32_t c) {
		return (!ur && (float)c >= (fk / y)) || (ur && (float)c >= (1 + (fk / jump)) * Y);
	});
	uint32_t g = hp.start_grid ? hp.start_grid : 256;
	if (g & (g - 1)) {
		return NTEDIT_E_ARG;
	}
	d.start_grid = g;
	uint32_t w = hp.node_window ? hp.node_window : 6 * k + 96;
	if (w < 4 * k + 64) {
		w = 4 * k + 64;
	}
	d.node_window = w;
	// speculative events are cut off after this many positions; far above any realistic chain of edits,
	// and always beyond the next forced event start
	uint32_t budget = hp.event_budget ? hp.event_budget : 2048;
	if (budget < 2 * g) {
		budget = 2 * g;
	}
	d.event_budget = budget;
	d.lanes = 2;     // runs of failing positions one position per lane (nte_machine.h, run_lanes)
	d.defer_run = 8; // ... when the clean run goes on for at least 8 more positions (3 Gbp: 2 / 4 / 8 / 16 -> machine 32.4 / 32.7 / 32.0 / 32.7 ms)
	d.defer_fail = 3; // (round 6; 3 Gbp genome-like draft: 0 / 2 / 3 / 4 / 8 -> machine 98.3 / 80.7 / 80.1 / 82.0 / 83.8 ms; i.i.d. draft 33.9 / 36.7 / 33.0 / 32.9 / 32.8)
	d.defer_fail_snv = 0;
	d.inline_tries = 8; // (3 Gbp bench, round 6 with the wavefront-per-event kernel at 168 registers: 2 / 4 / 8 / 12 / 16 tries -> machine 33.5 / 31.7 / 30.9 / 31.0 / 31.4 ms, genome-like 72.4 / 71.9 / 72.2 / 73.1 / 74.8; rounds 3-5: 16)
	for (uint32_t i = 0; i < nte::MAX_HASHES; i++) {
		d.mul[i] = (uint64_t)i ^ ((uint64_t)k * nte::MULTISEED);
	}
	*out = d;
	return 0;
}

} // namespace nte_host
