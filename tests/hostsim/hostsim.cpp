// hostsim.cpp -- TEST-ONLY host build of the device-side control logic.
//
// The event machine (ntedit_amd/csrc/nte_machine.h) and the hashing / filter
// arithmetic (nte_common.h) are written as host+device code.  This file
// compiles them with the host compiler and drives them with a plain loop
// (screen -> event starts -> one machine run per event -> render), so that
// the CPU test tier can check the product's control logic and host renderer
// against the oracle without a GPU.  It is never linked into the shipped
// library and is not a fallback: ntedit_amd refuses to run without the HIP
// extension.
#define NTE_COUNTERS 1
#include "../../ntedit_amd/csrc/nte_machine.h"
namespace nte { WorkCounters g_wc; }
#include "../../ntedit_amd/host/params.h"
#include "../../ntedit_amd/host/render.h"
#include "../../ntedit_amd/host/resolve.h"

#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

using namespace nte;

static Filter
make_filter(const uint8_t* data, uint64_t nbytes, uint32_t hash_num, bool counting = false)
{
	Filter f;
	f.data = data;
	filter_set_size(f, counting ? nbytes : nbytes * 8);
	f.hash_num = hash_num;
	f.counting = counting ? 1 : 0;
	return f;
}

// one event through one instantiation of the machine (lanes: the wavefront-per-event launch's run_lanes() path)
struct MachineOut
{
	u32 flags, fc, raw_first, cover_end, nsize, nbase;
	unsigned long long cycles;
};

template<u32 CFG>
static MachineOut
run_machine(const EventEnv& env, u32 start, bool lanes)
{
	MachineT<CFG> m(env);
	MachineOut o;
	o.cover_end = start;
	const unsigned long long t0 = __builtin_ia32_rdtsc();
	if (lanes) {
		m.template run<true>(start, o.cover_end);
	} else {
		m.template run<false>(start, o.cover_end);
	}
	o.cycles = __builtin_ia32_rdtsc() - t0;
	o.flags = m.flags;
	o.raw_first = m.first_chunk;
	o.fc = m.finish(start, o.cover_end);
	o.nsize = m.nsize;
	o.nbase = m.nbase;
	return o;
}

// same contract as the screening kernel
static void
sim_screen(const u8* seq, u64 n, const Filter& f, const DevParams& p, const u64* tab, u64* bitmap)
{
	memset(bitmap, 0, ((n + 63) / 64) * 8);
	HashState hs = { 0, 0 };
	u64 good = 0;
	for (u64 i = 0; i < n; i++) {
		u8 in = char_code(seq[i]);
		u8 out = i >= p.k ? char_code(seq[i - p.k]) : CODE_BAD;
		hash_roll(hs, tab, out, in);
		good = in == CODE_BAD ? 0 : good + 1;
		if (good >= p.k) {
			if (p.snv || filter_screen_absent(f, p, hs)) {
				u64 s = i + 1 - p.k;
				bitmap[s >> 6] |= 1ULL << (s & 63);
			}
		}
	}
}

// host twin of k_assess (nte_assess.hip): the run map = the absent bitmap minus the positions whose clean-state
// assessment cannot do anything; the very function the kernel calls (Machine::assess_gate) on a window of one position
static void
sim_assess(const u8* seq, u64 n, const Filter& f, const Filter& fr, const DevParams& p, const u64* tab, const u64* bitmap, u64* runmap)
{
	const u32 K = p.k + p.max_deletions + 1;
	std::vector<u8> win(p.k + K + 8);
	EventEnv env;
	memset(&env, 0, sizeof env);
	env.seq = seq;
	env.batch_end = seq + n;
	env.bitmap = bitmap;
	env.runmap = bitmap;
	env.tab = tab;
	env.p = &p;
	env.bloom = f;
	env.rep = fr;
	env.win = win.data();
	env.win_stride = 1;
	env.wave_size = 1;
	memcpy(runmap, bitmap, ((n + 63) / 64) * 8);
	for (u64 g = 0; g < n; g++) {
		if (!bit_absent(bitmap, g)) {
			continue;
		}
		bool clear = g + 2 * p.k <= n; // (the k-mer and the k rolls the gate makes; k_assess: `need`)
		for (u32 i = 0; clear && i < 2 * p.k; i++) {
			const u8 code = char_code(seq[g + i]);
			clear = code != CODE_BAD;
			win[i] = code;
		}
		MachineT<0> m(env);
		m.win_off = 0;
		m.win_ok = true;
		if (!clear) {
			// (a non-accepted character or the end of the batch within reach: the k-mer's own codes only)
			for (u32 i = 0; i < p.k; i++) {
				win[i] = char_code(seq[g + i]);
			}
			m.hs = m.seed_from_window();
			if (!m.assess_gate_kmer_only()) {
				runmap[g >> 6] &= ~(1ULL << (g & 63));
			}
			continue;
		}
		m.hs = m.seed_from_window();
		if (!m.assess_gate(g, seq[g + p.k - 1])) {
			runmap[g >> 6] &= ~(1ULL << (g & 63));
		}
	}
}

extern "C" int
hostsim_screen(
    const char* bases,
    uint64_t n,
    const uint8_t* bf,
    uint64_t bf_bytes,
    uint32_t hash_num,
    uint32_t k,
    uint64_t* bitmap,
    int counting,
    uint32_t min_threshold)
{
	ntedit_hip_params hp;
	nte_host::params_default(&hp);
	hp.min_threshold = min_threshold;
	DevParams p;
	if (nte_host::make_dev_params(hp, k, hash_num, false, &p, counting != 0)) {
		return -1;
	}
	u64 tab[TAB_WORDS];
	build_seed_tables(k, tab);
	Filter f = make_filter(bf, bf_bytes, hash_num, counting != 0);
	sim_screen((const u8*)bases, n, f, p, tab, bitmap);
	return 0;
}

// TEST-ONLY: extras for the NEXT hostsim_polish call (cleared by it): segment descriptors, per-entry output
// sizes and cover ends (the multi-GPU gather's inputs), flag 1 = no TSV / VCF header lines (shard files),
// and a file that receives the ntedit_hip_edit records: u64 count | records | u64 pool bytes | pool.
static const ntedit_hip_segment* g_x_segments = nullptr;
static uint64_t* g_x_sizes = nullptr;
static uint32_t* g_x_covers = nullptr;
static unsigned g_x_flags = 0;
static uint8_t* g_x_cuts_ok = nullptr; // per entry: nte_host::cuts_ok (= ntedit_hip_result_cuts_ok) for g_x_segments
static std::string g_x_edits_path;

extern "C" void
hostsim_set_cuts_ok(uint8_t* ok)
{
	g_x_cuts_ok = ok;
}

extern "C" void
hostsim_set_render_extras(const ntedit_hip_segment* segments, uint64_t* sizes, uint32_t* covers, unsigned flags, const char* edits_path)
{
	g_x_segments = segments;
	g_x_sizes = sizes;
	g_x_covers = covers;
	g_x_flags = flags;
	g_x_edits_path = edits_path ? edits_path : "";
}

extern "C" int
hostsim_polish(
    const char* bases,
    uint64_t n,
    const uint64_t* offsets,
    const uint32_t* lens,
    const char* const* names,
    uint32_t n_contigs,
    const uint8_t* bf,
    uint64_t bf_bytes,
    uint32_t hash_num,
    uint32_t k,
    const uint8_t* rep,
    uint64_t rep_bytes,
    uint32_t rep_hash_num,
    const ntedit_hip_params* hp,
    const char* fa_path,
    const char* tsv_path,
    uint64_t* n_events_out,
    uint64_t* n_applied_out,
    int counting,
    int rep_counting,
    const char* vcf_path,
    const char* annot_path)
{
	DevParams p;
	int rc = nte_host::make_dev_params(*hp, k, hash_num, rep != nullptr, &p, counting != 0);
	if (rc) {
		return rc;
	}
	u64 tab[TAB_WORDS];
	build_seed_tables(k, tab);
	Filter f = make_filter(bf, bf_bytes, hash_num, counting != 0);
	Filter fr = make_filter(rep, rep_bytes, rep ? rep_hash_num : 0, rep_counting != 0);

	std::vector<u64> bitmap((n + 63) / 64 + 1);
	sim_screen((const u8*)bases, n, f, p, tab, bitmap.data());

	// the run map (k_assess): by default where the GPU path computes one (HOSTSIM_ASSESS=0 / 1: never / always)
	std::vector<u64> runmap_store;
	const u64* runmap = bitmap.data();
	{
		bool assess = p.snv || counting != 0;
		if (const char* it = getenv("HOSTSIM_ASSESS")) {
			assess = atoi(it) != 0;
		}
		if ((p.snv || counting != 0) && (p.k - 1) / p.jump + 1 > 32) {
			assess = false; // (Machine::lane_counts_fit: the per-lane step 2 keeps at most 32 counts)
		}
		if (assess) {
			runmap_store.assign(bitmap.size(), 0);
			sim_assess((const u8*)bases, n, f, fr, p, tab, bitmap.data(), runmap_store.data());
			runmap = runmap_store.data();
		}
	}
	// event starts, in position order
	std::vector<u64> events;
	for (u64 g = 0; g < n; g++) {
		if (is_event_start(runmap, g, p.start_grid)) {
			events.push_back(g);
		}
	}
	// arena: generous; like ntedit_hip_polish_batch, a batch that still runs out of it is run again
	// with four times as much (up to 4 times)
	u32 arena_chunks = (u32)(events.size() * 4 + n + 1024);
	std::vector<Item> arena;
	u32 arena_next = 0;
	std::vector<u32> ev_first;
	bool overflow = false;
	bool arena_full = false;
	for (int attempt = 0; attempt < 5; attempt++, arena_chunks *= 4) {
	arena.assign((size_t)arena_chunks * CHUNK_ITEMS, Item());
	arena_next = 0;
	overflow = false;
	arena_full = false;
	ev_first.clear();
	std::vector<Node> nodes(p.node_window);
	std::vector<u32> ov_pos(p.node_window);
	std::vector<u8> ov_chr(p.node_window);
	std::vector<u8> win(2 * p.k + p.max_deletions + 8 + 32 + 64);
	std::vector<u8> prev(p.node_window);
	std::vector<int16_t> lps(p.node_window);
	// contig of every event
	std::vector<u32> ev_contig(events.size());
	{
		u32 ci = 0;
		for (size_t i = 0; i < events.size(); i++) {
			while (ci + 1 < n_contigs && offsets[ci + 1] <= events[i]) {
				ci++;
			}
			ev_contig[i] = ci;
		}
	}
	// one event, the way a launch with the given budget runs it (pass 1 + pass 2 of the two-pass scheme)
	auto run_event = [&](size_t idx, u32 budget) -> u32 {
		const u64 g = events[idx];
		const u32 ci = ev_contig[idx];
		DevParams pe = p;
		pe.event_budget = budget;
		if (const char* it = getenv("HOSTSIM_INLINE_TRIES")) {
			pe.inline_tries = (u32)atoi(it);
		}
		if (const char* it = getenv("HOSTSIM_LANES")) { // 0: position by position, 1: clean runs one position per lane, 2: (default) also behind substitutions
			pe.lanes = (u32)atoi(it);
		}
		if (const char* it = getenv("HOSTSIM_DEFER_RUN")) {
			pe.defer_run = (u32)atoi(it);
		}
		if (const char* it = getenv("HOSTSIM_DEFER_FAIL")) {
			pe.defer_fail = (u32)atoi(it);
		}
		EventEnv env;
		env.seq = (const u8*)bases + offsets[ci];
		env.batch_end = (const u8*)bases + n;
		env.len = lens[ci];
		env.contig = ci;
		env.gbase = offsets[ci];
		env.bitmap = bitmap.data();
		env.runmap = runmap;
		env.tab = tab;
		env.p = &pe;
		env.bloom = f;
		env.rep = fr;
		env.nodes = nodes.data();
		env.ov_pos = ov_pos.data();
		env.ov_chr = ov_chr.data();
		env.win = getenv("HOSTSIM_NO_WINDOW") ? nullptr : win.data();
		env.win_stride = 1;
		env.prev = prev.data();
		env.lps = lps.data();
		env.arena = arena.data();
		env.arena_next = &arena_next;
		env.arena_chunks = arena_chunks;
		env.defer_sweeps = getenv("HOSTSIM_TWO_PASS") != nullptr;
		env.wave_size = 1;
		u32 start = (u32)(g - offsets[ci]);
		// the instantiation of the machine the GPU path would launch for this configuration (HOSTSIM_CFG=0: the general one)
		u32 cfg = machine_cfg_of(pe, f, fr);
		cfg = (cfg & 15u) == 15u ? 15u : (cfg & 11u) == 11u ? 11u : 0u;
		if (const char* it = getenv("HOSTSIM_CFG")) {
			cfg = atoi(it) ? cfg : 0u;
		}
		MachineOut o = cfg == 15 ? run_machine<15>(env, start, !env.defer_sweeps) : cfg == 11 ? run_machine<11>(env, start, !env.defer_sweeps) : run_machine<0>(env, start, !env.defer_sweeps);
		if (getenv("HOSTSIM_HIST")) { fprintf(stderr, "EVT %u %u %llu %d\n", start, o.cover_end, o.cycles, (int)(o.raw_first != NONE32)); }
		if (o.flags & (EV_OVERFLOW | EV_ARENA_FULL)) {
			overflow = true;
			arena_full = arena_full || (o.flags & EV_ARENA_FULL);
		}
		if (o.flags & EV_DEFERRED) {
			// second pass: the same event again, sweeps allowed
			env.defer_sweeps = false;
			o = cfg == 15 ? run_machine<15>(env, start, true) : cfg == 11 ? run_machine<11>(env, start, true) : run_machine<0>(env, start, true);
			if (o.flags & (EV_OVERFLOW | EV_ARENA_FULL)) {
				overflow = true;
				arena_full = arena_full || (o.flags & EV_ARENA_FULL);
			}
			return o.fc;
		}
		u32 fc = o.fc;
		if (getenv("HOSTSIM_DEBUG")) {
			if (fc != NONE32) { const Item* c = arena.data() + (size_t)fc*CHUNK_ITEMS; for (u32 i=0;i<c[0].w[1];i++) fprintf(stderr,"   item %u: %08x %u %u %u\n", i, c[i].w[0], c[i].w[1], c[i].w[2], c[i].w[3]); }
			fprintf(stderr, "event contig %u start %u cover_end %u flags %u first_chunk %d nsize %u nbase %u\n", ci, start, o.cover_end, o.flags, (int)fc, o.nsize, o.nbase);
		}
		return fc;
	};
	std::vector<u32> first_by_event(events.size(), NONE32);
	for (size_t i = 0; i < events.size(); i++) {
		first_by_event[i] = run_event(i, p.event_budget);
	}
	// parked events the serial order needs: re-run to completion (what ntedit_hip_polish_batch does)
	if (!overflow) {
		nte_host::Resolver rs(arena.data(), arena.size(), first_by_event.data(), events.size());
		std::vector<u32> rerun;
		if (!rs.start(rerun)) {
			return -7;
		}
		unsigned rounds = 0;
		u32 wide_budget = p.event_budget ? p.event_budget : 1;
		while (!rerun.empty() && !overflow) {
			// (the plan of PolishRun::collect: one parked event per contig and round at first, then all of them with a growing budget)
			const bool wide = rounds >= 6;
			if (wide) {
				rerun.clear();
				rs.parked_behind(rerun);
				wide_budget = wide_budget < (1u << 26) ? wide_budget * 16u : 0u;
			}
			for (u32 i : rerun) {
				first_by_event[i] = run_event(i, wide ? wide_budget : 0);
			}
			rerun.clear();
			rounds++;
			if (overflow) {
				break; // a re-run ran out of arena / rope window: the whole batch again with more room (as the C ABI does)
			}
			if (!rs.resume(rerun, wide)) {
				return -7;
			}
		}
		if (getenv("HOSTSIM_DEBUG") || getenv("HOSTSIM_COUNTERS")) {
			fprintf(stderr, "RESOLVE rounds %u\n", rounds);
		}
	}
	ev_first = first_by_event; // (one entry per event, NONE32 where there is no output, as the C ABI hands them on)
	if (!(overflow && arena_full)) {
		break;
	}
	} // attempt
	if (getenv("HOSTSIM_COUNTERS")) {
		fprintf(stderr, "COUNTERS events %zu probes %llu slow_rolls %llu ins_cands %llu del_cands %llu sweeps %llu lane_batches %llu lane_positions %llu lane_walked %llu lane_edits %llu\n",
		        events.size(), g_wc.probes, g_wc.slow_rolls, g_wc.ins_cands, g_wc.del_cands, g_wc.sweeps, g_wc.lane_batches, g_wc.lane_positions, g_wc.lane_walked, g_wc.lane_edits);
	}
	if (overflow) {
		return NTEDIT_E_OVERFLOW;
	}
	FILE* fa = fa_path ? fopen(fa_path, "w") : nullptr;
	FILE* tsv = tsv_path ? fopen(tsv_path, "w") : nullptr;
	if (tsv && !(g_x_flags & 1)) {
		nte_host::write_tsv_header(tsv, k, hp->jump, counting != 0);
	}
	FILE* vcf = vcf_path ? fopen(vcf_path, "w") : nullptr;
	if (vcf && !(g_x_flags & 1)) {
		nte_host::write_vcf_header(vcf, "draft");
	}
	nte_host::RenderOptions ropt;
	std::vector<ntedit_hip_edit> x_edits;
	std::string x_pool;
	ropt.segments = g_x_segments;
	ropt.out_sizes = g_x_sizes;
	if (g_x_covers) {
		// (the serial-order filter alone, as ntedit_hip_result_cover_ends does it)
		if (nte_host::cover_ends(arena.data(), arena.size(), ev_first.data(), ev_first.size(), n_contigs, g_x_covers)) {
			return -9;
		}
	}
	if (g_x_cuts_ok && g_x_segments) {
		std::vector<uint32_t> halos(n_contigs);
		for (uint32_t i = 0; i < n_contigs; i++) {
			halos[i] = g_x_segments[i].halo;
		}
		if (nte_host::cuts_ok(arena.data(), arena.size(), ev_first.data(), ev_first.size(), n_contigs, lens, halos.data(), g_x_cuts_ok)) {
			return -9;
		}
	}
	g_x_cuts_ok = nullptr;
	if (g_x_sizes) {
		memset(g_x_sizes, 0, (size_t)n_contigs * 3 * sizeof(uint64_t));
	}
	const std::string x_edits_path = g_x_edits_path;
	if (!x_edits_path.empty()) {
		ropt.edits = &x_edits;
		ropt.edit_pool = &x_pool;
	}
	hostsim_set_render_extras(nullptr, nullptr, nullptr, 0, nullptr);
	if (const char* rt = getenv("HOSTSIM_RENDER_THREADS")) {
		ropt.threads = (unsigned)atoi(rt);
	}
	// tests: one contig per work unit unless told otherwise, so that the few-contig cases still go through
	// the concurrent renderer
	ropt.unit_bases = getenv("HOSTSIM_RENDER_UNIT") ? (unsigned)atoi(getenv("HOSTSIM_RENDER_UNIT")) : 1;
	// contigs rendered in parts wherever a cut is allowed (unit of 1 base: every event that keeps the margin opens a part)
	// -- the library does this for contigs of several megabases; HOSTSIM_RENDER_PARTS=0 turns it off
	if (!getenv("HOSTSIM_RENDER_PARTS") || atoi(getenv("HOSTSIM_RENDER_PARTS"))) {
		ropt.part_margin = k + hp->max_deletions + 48;
	}
	ropt.snv = hp->snv != 0;
	nte_host::Annotations* ann = annot_path ? nte_host::annotations_load(annot_path) : nullptr;
	ropt.annot = ann;
	nte_host::RenderStats st;
	rc = nte_host::render_batch(
	    arena.data(),
	    arena.size(),
	    ev_first.data(),
	    ev_first.size(),
	    bases,
	    offsets,
	    lens,
	    names,
	    n_contigs,
	    fa,
	    tsv,
	    &st,
	    vcf,
	    &ropt);
	if (!x_edits_path.empty()) {
		if (FILE* ef = fopen(x_edits_path.c_str(), "wb")) {
			const uint64_t ne = x_edits.size(), np = x_pool.size();
			fwrite(&ne, 8, 1, ef);
			fwrite(x_edits.data(), sizeof(ntedit_hip_edit), x_edits.size(), ef);
			fwrite(&np, 8, 1, ef);
			fwrite(x_pool.data(), 1, x_pool.size(), ef);
			fclose(ef);
		}
	}
	if (vcf) {
		fclose(vcf);
	}
	nte_host::annotations_free(ann);
	if (fa) {
		fclose(fa);
	}
	if (tsv) {
		fclose(tsv);
	}
	if (n_events_out) {
		*n_events_out = events.size();
	}
	if (n_applied_out) {
		*n_applied_out = st.events_applied;
	}
	return rc;
}


// TEST-ONLY: dumps what the host FASTA/FASTQ reader (ntedit_amd/host/fasta.cpp) yields for a file:
// "<header length> <sequence length>\n<header>\n<sequence>\n" per record
#include "../../ntedit_amd/host/fasta.h"
extern "C" int
hostsim_fasta_dump(const char* in_path, const char* out_path)
{
	nte_host::FastaReader r(in_path);
	if (!r.ok()) {
		return -1;
	}
	FILE* o = fopen(out_path, "wb");
	if (!o) {
		return -2;
	}
	std::string hdr, blob;
	int n = 0;
	for (;;) {
		const size_t before = blob.size();
		if (!r.next(hdr, blob)) {
			break;
		}
		fprintf(o, "%zu %zu\n", hdr.size(), blob.size() - before);
		fwrite(hdr.data(), 1, hdr.size(), o);
		fputc('\n', o);
		fwrite(blob.data() + before, 1, blob.size() - before, o);
		fputc('\n', o);
		n++;
	}
	fclose(o);
	return n;
}

// TEST-ONLY: reads a file to its end with the streaming reader; 1 if it reported a damaged input (io_error), 0 if not, -1 cannot open
extern "C" int
hostsim_fasta_io_error(const char* in_path)
{
	nte_host::FastaReader r(in_path);
	if (!r.ok()) {
		return -1;
	}
	std::string hdr, blob;
	while (r.next(hdr, blob)) {
		blob.clear();
	}
	return r.io_error() ? 1 : 0;
}

// TEST-ONLY: 1 = the streaming reader inflates through zlib instead of host/gunzip.cpp; returns the old setting
extern "C" int
hostsim_gzip_through_zlib(int on)
{
	return nte_host::set_gzip_through_zlib(on);
}

// TEST-ONLY: inflates a .gz file with host/gunzip.cpp alone into out_path, `block` bytes per read() call (so that tests
// can put the block boundaries anywhere), the compressed file read through a buffer of in_bytes (0: the default).  Returns the bytes written; -1 not a gzip file, -2 output, -3 the stream
// failed (what it produced up to there is in the file), -4 a member's CRC-32 or length does not match its trailer
#include "../../ntedit_amd/host/gunzip.h"
#include <zlib.h>
extern "C" long long
hostsim_gunzip_buffers(const char* in_path, const char* out_path, unsigned block, unsigned in_bytes)
{
	nte_host::Gunzip g(in_bytes ? in_bytes : (4u << 20));
	if (!g.open(in_path)) {
		return -1;
	}
	FILE* o = fopen(out_path, "wb");
	if (!o) {
		return -2;
	}
	std::vector<unsigned char> buf(nte_host::Gunzip::WINDOW + (size_t)block + nte_host::Gunzip::SLACK);
	unsigned char* dst = buf.data() + nte_host::Gunzip::WINDOW;
	long long total = 0;
	unsigned long crc = crc32(0L, Z_NULL, 0);
	unsigned long long member = 0;
	bool mismatch = false;
	for (;;) {
		const size_t n = g.read(dst, block);
		if (n == 0 && !g.member_end()) {
			break;
		}
		fwrite(dst, 1, n, o);
		total += (long long)n;
		crc = crc32(crc, dst, (unsigned)n);
		member += n;
		if (g.member_end()) {
			if ((unsigned)crc != g.member_crc() || (unsigned)(member & 0xffffffffull) != g.member_isize()) {
				mismatch = true;
			}
			crc = crc32(0L, Z_NULL, 0);
			member = 0;
		}
	}
	fclose(o);
	return g.failed() ? -3 : mismatch ? -4 : total;
}

extern "C" long long
hostsim_gunzip(const char* in_path, const char* out_path, unsigned block)
{
	return hostsim_gunzip_buffers(in_path, out_path, block, 0);
}

// TEST-ONLY: the same dump through the mapped, multi-threaded reader (ntedit_amd/host/fasta_map.cpp);
// -1 when it refuses the file (then the streaming reader is the one that parses it)
#include "../../ntedit_amd/host/fasta_map.h"
extern "C" int
hostsim_fasta_map_dump(const char* in_path, const char* out_path, unsigned threads)
{
	nte_host::FastaMap m(in_path, threads);
	if (!m.ok()) {
		return -1;
	}
	FILE* o = fopen(out_path, "wb");
	if (!o) {
		return -2;
	}
	m.measure(0, m.records());
	std::vector<size_t> idx(m.records());
	std::vector<std::string> seqs(m.records());
	std::vector<char*> dst(m.records());
	for (size_t i = 0; i < m.records(); i++) {
		idx[i] = i;
		seqs[i].resize(m.length(i));
		dst[i] = seqs[i].empty() ? nullptr : &seqs[i][0];
	}
	std::vector<char> dummy(1);
	for (auto& d : dst) {
		if (!d) {
			d = dummy.data();
		}
	}
	m.copy(idx.data(), dst.data(), idx.size());
	for (size_t i = 0; i < m.records(); i++) {
		const std::string hdr = m.header(i);
		fprintf(o, "%zu %zu\n", hdr.size(), seqs[i].size());
		fwrite(hdr.data(), 1, hdr.size(), o);
		fputc('\n', o);
		fwrite(seqs[i].data(), 1, seqs[i].size(), o);
		fputc('\n', o);
	}
	fclose(o);
	return (int)m.records();
}

// TEST-ONLY: nte::filter_slot (the device's hv % slots without a division) for n hash values
extern "C" void
hostsim_filter_slots(unsigned long long slots, const unsigned long long* hv, unsigned long long n, unsigned long long* out)
{
	nte::Filter f;
	memset(&f, 0, sizeof f);
	nte::filter_set_size(f, slots);
	for (unsigned long long i = 0; i < n; i++) {
		out[i] = nte::filter_slot(f, hv[i]);
	}
}

// The tables of ntedit.cpp:172-348 as the PRODUCT's sources hold them -- MachineT::candidate_bases / insertion_candidate
// (nte_machine_position.inc / _rope.inc, the very functions the kernels run) and params.cpp's num_tries, read through
// DevParams::ins_tries -- in the canonical text form of tests/tools/reference_tables.py.  Length written or -1.
extern "C" long
hostsim_tables_dump(char* out, size_t cap)
{
	size_t n = 0;
	auto put = [&](const char* fmt, auto... a) -> bool {
		int w = snprintf(out + n, n < cap ? cap - n : 0, fmt, a...);
		if (w < 0 || n + (size_t)w >= cap) {
			return false;
		}
		n += (size_t)w;
		return true;
	};
	if (!put("num_tries")) {
		return -1;
	}
	for (uint32_t i = 0; i <= 5; i++) {
		ntedit_hip_params hp;
		nte_host::params_default(&hp);
		hp.max_insertions = i;
		hp.max_deletions = i < 2 ? i : hp.max_deletions;
		DevParams d;
		if (nte_host::make_dev_params(hp, 25, 3, false, &d) != 0 || !put(" %u", d.ins_tries)) {
			return -1;
		}
	}
	put("%c", (int)10);
	for (int snv = 0; snv < 2; snv++) {
		for (const char* c = "ATCGRYSWKMBDHVN"; *c; c++) {
			u8 cand[8];
			const u32 nc = MachineT<0>::candidate_bases((u8)*c, snv != 0, cand);
			if (!put("%s %c ", snv ? "snv" : "polish", *c)) {
				return -1;
			}
			for (u32 q = 0; q < nc; q++) {
				put("%c", cand[q]);
			}
			put("%c", (int)10);
		}
	}
	for (const char* c = "ACGT"; *c; c++) {
		if (!put("multi %c", *c)) {
			return -1;
		}
		for (u32 i = 0; i < 341; i++) {
			u8 ins[INDEL_BYTES + 1];
			const u32 m = MachineT<0>::insertion_candidate((u8)*c, i, ins);
			ins[m] = 0;
			if (!put(" %s", (const char*)ins)) {
				return -1;
			}
		}
		put("%c", (int)10);
	}
	return (long)n;
}
