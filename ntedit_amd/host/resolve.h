// resolve.h -- which speculative events the serial order keeps, and which parked ones it needs.
//
// Events are independent runs of the serial machine started at every position where the reference's
// main loop could be in a clean state.  Walking a contig's events in position order,
// an event is applied iff it starts at or behind the end of the previous applied event's run
// (cover_end); everything that starts inside that run was speculation.  An event that was parked by
// the launch budget (EV_UNFINISHED) has no valid cover_end: if the walk reaches it as an APPLIED
// event it has to be re-run to completion before the rest of its contig can be decided.
#pragma once
#include "../csrc/nte_common.h"

#include <cstddef>
#include <cstdint>
#include <vector>

namespace nte_host {

class Resolver
{
  public:
	// first[i] = first arena chunk of event i (nte::NONE32: the event produced nothing), events in
	// global position order; both arrays may be updated between start() and resume()
	Resolver(const nte::Item* arena, size_t arena_items, const uint32_t* first, size_t n_events)
	  : arena_(arena)
	  , arena_items_(arena_items)
	  , first_(first)
	  , n_(n_events)
	{}
	void rebind(const nte::Item* arena, size_t arena_items, const uint32_t* first)
	{
		arena_ = arena;
		arena_items_ = arena_items;
		first_ = first;
	}
	// first round: every contig.  Appends the events that must be re-run; false on a malformed arena.
	bool start(std::vector<uint32_t>& rerun)
	{
		blocked_.clear();
		size_t i = 0;
		while (i < n_) {
			const nte::Item* h = header(i);
			if (!h) {
				if (first_[i] != nte::NONE32) {
					return false;
				}
				i++;
				continue;
			}
			i = walk(i, rerun);
		}
		return true;
	}
	// after the events handed out last time were re-run (their records replaced): carry on.
	// may_park: a re-run with a (larger) budget may have parked an event again -- it is handed out once more
	bool resume(std::vector<uint32_t>& rerun, bool may_park = false)
	{
		std::vector<size_t> todo;
		todo.swap(blocked_);
		for (size_t i : todo) {
			const nte::Item* h = header(i);
			if (!h || ((h->w[3] & nte::EV_UNFINISHED) && !may_park)) {
				return false; // the re-run must have completed the event
			}
			walk(i, rerun);
		}
		return true;
	}
	// Every parked event behind the points where the walk is waiting: the events a widened re-run takes on at once
	// instead of one per contig and round (most of them speculation, as in the first launch).
	void parked_behind(std::vector<uint32_t>& out) const
	{
		for (size_t b : blocked_) {
			const uint32_t contig = header(b)->w[0];
			for (size_t i = b; i < n_; i++) {
				const nte::Item* h = header(i);
				if (!h) {
					continue;
				}
				if (h->w[0] != contig) {
					break;
				}
				if (h->w[3] & nte::EV_UNFINISHED) {
					out.push_back((uint32_t)i);
				}
			}
		}
	}

  private:
	const nte::Item* header(size_t i) const
	{
		const uint32_t fc = first_[i];
		if (fc == nte::NONE32 || (size_t)fc * nte::CHUNK_ITEMS + 1 >= arena_items_) {
			return nullptr;
		}
		return arena_ + (size_t)fc * nte::CHUNK_ITEMS + 1;
	}
	// event i opens (or continues) its contig as an applied event; returns the index of the first
	// event of the next contig
	size_t walk(size_t i, std::vector<uint32_t>& rerun)
	{
		const uint32_t contig = header(i)->w[0];
		uint32_t cover = 0; // (a blocked event that is walked again starts at or behind every earlier cover)
		bool waiting = false;
		for (; i < n_; i++) {
			const nte::Item* h = header(i);
			if (!h) {
				continue;
			}
			if (h->w[0] != contig) {
				break;
			}
			if (waiting || h->w[1] < cover) {
				continue; // inside an earlier run (or behind an undecided one)
			}
			if (h->w[3] & nte::EV_UNFINISHED) {
				rerun.push_back((uint32_t)i);
				blocked_.push_back(i);
				waiting = true;
				continue;
			}
			cover = h->w[2];
		}
		return i;
	}

	const nte::Item* arena_;
	size_t arena_items_;
	const uint32_t* first_;
	size_t n_;
	std::vector<size_t> blocked_;
};

} // namespace nte_host
