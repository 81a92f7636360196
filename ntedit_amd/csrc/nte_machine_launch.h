// nte_machine_launch.h -- launch interface of the event-machine kernels.
#pragma once
#include "nte_machine.h"

#include <hip/hip_runtime.h>

namespace nte {

struct MachineArgs
{
	const u8* seq;
	u64 n_bytes;        // batch bytes
	const u64* offsets; // start of every contig in the batch
	const u32* lens;
	u32 n_contigs;
	const u64* bitmap;
	const u64* runmap; // see EventEnv
	const u64* events;
	u64 n_events;
	const u64* tabs;
	DevParams p;
	Filter bloom, rep;
	// per-thread workspace slabs
	Node* ws_nodes;
	u32* ws_ov_pos;
	u8* ws_ov_chr;
	u8* ws_prev;
	int16_t* ws_lps;
	u8* ws_win;     // used when the windows do not fit in LDS
	u32 win_bytes;  // window bytes per thread
	u32 win_in_lds;
	// wavefront-per-event kernel only: the per-event workspace (rope nodes, overlay, scratch
	// strings) lives in LDS too, lds_slab bytes per event starting at byte lds_ws_off of the
	// dynamic LDS (0 = use the global slabs)
	u32 lds_ws_off;
	u32 lds_slab;
	// output
	Item* arena;
	u32* arena_next;
	u32 arena_chunks;
	u32* first_chunk; // per event
	u32* status;      // OR of EV_OVERFLOW / EV_ARENA_FULL seen
	// two-pass launch: pass 1 (defer = 1) postpones events that need an indel sweep by
	// appending their index to `deferred`; pass 2 runs exactly that list (ev_list)
	u32 defer;
	const u32* ev_list; // nullptr = all events 0..n_events-1
	u32* deferred;
	u32* n_deferred;
	u32* n_unfinished; // events parked by the budget (p.event_budget) in this launch
	// per event, written for every event a launch runs to its end: where its serial run ended, as a byte
	// position of the batch (contig offset + cover_end), and its flags (EV_UNFINISHED: the end is unknown).
	// The launch driver uses them to skip events that start inside an earlier event's run (nullptr: not kept)
	u64* ev_cover;
	u8* ev_flags;
	// dynamic work distribution: every worker (thread / wavefront) takes the next event from this
	// counter (zeroed before the launch) -- event costs are heavy-tailed, a static split leaves most of
	// the chip waiting for the unluckiest worker
	u32* work_counter;
	u32 cfg; // which instantiation runs it (machine_cfg_pick; 0 = the general one)
};

constexpr int MACHINE_TPB = 256;

// k_assess (nte_assess.hip): the run map of the positions [pos_begin, pos_end) of the batch, pos_begin a multiple of
// the tile (assess_tile()), n_tiles = tiles of that range
struct AssessArgs
{
	const u8* seq;
	u64 n_bytes;
	const u64* bitmap;
	u64* runmap;
	const u64* tabs;
	DevParams p;
	Filter bloom, rep;
	u64 pos_begin, pos_end, n_tiles;
	const u32* cand_map; // -s 1, plain filter: 4 bits per position (nte_bin_wc.inc MODE 1), or null
};
void launch_k_assess(unsigned blocks, hipStream_t stream, const AssessArgs& a);
int assess_tile();

// The kernels exist once per machine configuration (MachineCfg bits; nte_machine_thread.hip / nte_machine_wave.hip are
// compiled with -DNTE_CFG=<bits>): 0 = any configuration, 11 = -m 0 without -a, plain power-of-two filters (a secondary
// filter allowed), 15 = the same without a secondary filter.
#define NTE_MACHINE_CFGS(X) X(0) X(11) X(15)
#define NTE_DECL_MACHINE(C)                                                                                              \
	void launch_k_machine_thread_cfg##C(unsigned blocks, size_t dyn_lds, hipStream_t stream, const MachineArgs& a);     \
	void launch_k_machine_wave_cfg##C(unsigned blocks, size_t dyn_lds, hipStream_t stream, const MachineArgs& a);       \
	void machine_wave_profile_cfg##C(unsigned long long out[64]);                                                       \
	unsigned long long machine_thread_gathers_cfg##C();                                                                 \
	unsigned machine_thread_evlog_cfg##C(unsigned long long* out, unsigned cap);                                         \
	unsigned machine_wave_evlog_cfg##C(unsigned long long* out, unsigned cap);
NTE_MACHINE_CFGS(NTE_DECL_MACHINE)
#undef NTE_DECL_MACHINE

// the instantiation a launch uses: the most specific one its configuration allows
inline u32
machine_cfg_pick(const MachineArgs& a)
{
	const u32 cfg = machine_cfg_of(a.p, a.bloom, a.rep);
	return (cfg & 15u) == 15u ? 15u : (cfg & 11u) == 11u ? 11u : 0u;
}

// one thread per event (pass 1 / single pass)
inline void
launch_k_machine_thread(unsigned blocks, size_t dyn_lds, hipStream_t stream, const MachineArgs& a)
{
	switch (a.cfg) {
#define NTE_CASE(C) case C: launch_k_machine_thread_cfg##C(blocks, dyn_lds, stream, a); break;
		NTE_MACHINE_CFGS(NTE_CASE)
#undef NTE_CASE
	default: launch_k_machine_thread_cfg0(blocks, dyn_lds, stream, a);
	}
}

// one wavefront per event (sweep-only second pass)
inline void
launch_k_machine_wave(unsigned blocks, size_t dyn_lds, hipStream_t stream, const MachineArgs& a)
{
	switch (a.cfg) {
#define NTE_CASE(C) case C: launch_k_machine_wave_cfg##C(blocks, dyn_lds, stream, a); break;
		NTE_MACHINE_CFGS(NTE_CASE)
#undef NTE_CASE
	default: launch_k_machine_wave_cfg0(blocks, dyn_lds, stream, a);
	}
}

// the device code's candidate tables, raw (nte_machine_thread.hip: k_tables; 30 * 8 + 4 * 341 * 8 bytes)
constexpr size_t TABLES_RAW_BYTES = 30 * 8 + 4 * 341 * 8;
void launch_k_tables(hipStream_t stream, u8* out);

// lanes per event in that kernel (a 256-thread block runs 256 / group events at a time)
int machine_wave_group();

inline unsigned long long
machine_thread_gathers()
{
	unsigned long long v = 0;
#define NTE_SUM(C) v += machine_thread_gathers_cfg##C();
	NTE_MACHINE_CFGS(NTE_SUM)
#undef NTE_SUM
	return v;
}

// the long events of the launches since the last call (profile build), of the thread- / wavefront-per-event kernels
inline unsigned
machine_evlog(bool wave, unsigned long long* out, unsigned cap)
{
	unsigned n = 0;
#define NTE_SUM(C) n += wave ? machine_wave_evlog_cfg##C(out + 4 * n, cap - n) : machine_thread_evlog_cfg##C(out + 4 * n, cap - n);
	NTE_MACHINE_CFGS(NTE_SUM)
#undef NTE_SUM
	return n;
}

inline void
machine_wave_profile(unsigned long long out[64])
{
	for (int i = 0; i < 64; i++) {
		out[i] = 0;
	}
	unsigned long long t[64];
#define NTE_SUM(C) machine_wave_profile_cfg##C(t); for (int i = 0; i < 64; i++) { out[i] += t[i]; }
	NTE_MACHINE_CFGS(NTE_SUM)
#undef NTE_SUM
}

} // namespace nte
