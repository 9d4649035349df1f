// dev_arena.h -- device memory for the engine's buffers out of large chunks.
//
// Every subject brings four to eight buffers (text, suffix array, probe table, packed symbols ...), every call its
// scratch.  As separate hipMallocs they are separate mappings, and pass A -- scattered 16-byte loads from the packed
// texts of some two hundred pairs at a time -- runs 3 % slower on them than on the same buffers carved out of a few
// large allocations (bench set: 6.02 -> 5.84 ms; chunks of 256 MiB ... 16 GiB measured alike, skewing the buffers'
// start addresses without packing them does nothing: it is the number of mappings, not their alignment).  So the
// engine's device buffers come from chunks of ANDI_ARENA_MB MiB (default 2048; 0: plain hipMalloc), first fit with
// coalescing; a request of more than half a chunk, or one that no chunk can be found or made for, goes to hipMalloc.
// One arena per device, shared by the contexts on it.  The chunks outlive the contexts (round 4): a process that calls
// the seam again finds them -- on some boxes of the pool a fresh hipMalloc of the 6.5 GB a 29-genome job takes costs
// 0.3 s (the driver clears what it hands out), every call, which was three quarters of the seam's wall time there;
// andi_hip_trim() gives the unused chunks back (the Python binding calls it at exit); at most ANDI_ARENA_KEEP MiB
// (default 8192) stay behind a device's last context, 0 releases everything as before.  Freeing waits for the device like hipFree does, so a block is never handed out again while a kernel may
// still use it.
#pragma once
#include <hip/hip_runtime.h>
#include "knobs.h"

#include <cstdlib>
#include <iterator>
#include <map>
#include <mutex>
#include <unordered_map>
#include <vector>

namespace andi_arena {

struct Chunk {
	char *base = nullptr;
	size_t size = 0, used = 0;
	std::map<size_t, size_t> free_blocks; // offset -> length, coalesced
};

struct Arena {
	std::mutex mu;
	std::vector<Chunk> chunks;
	std::unordered_map<void *, size_t> live; // block -> its (rounded) length
	int contexts = 0;
};

inline Arena &of_device(int dev) {
	static Arena *arenas = new Arena[64]; // (never destroyed: a context may be released after the statics of this library)
	return arenas[dev >= 0 && dev < 64 ? dev : 0];
}

inline size_t chunk_bytes() {
	static const size_t v = [] {
		const char *e = andi_knob(KNOB_ARENA_MB);
		const long mb = e ? atol(e) : 2048;
		return mb > 0 ? (size_t)mb << 20 : (size_t)0;
	}();
	return v;
}

constexpr size_t GRAIN = 4096;

inline hipError_t dev_malloc(void **p, size_t bytes) {
	const size_t chunk = chunk_bytes(), need = (bytes + GRAIN - 1) & ~(GRAIN - 1);
	if (chunk == 0 || bytes == 0 || need > chunk / 2) return hipMalloc(p, bytes ? bytes : 1);
	int dev = 0;
	if (hipGetDevice(&dev) != hipSuccess) return hipMalloc(p, bytes);
	Arena &A = of_device(dev);
	std::lock_guard<std::mutex> lock(A.mu);
	for (int attempt = 0; attempt < 2; ++attempt) {
		for (Chunk &c : A.chunks) {
			for (auto it = c.free_blocks.begin(); it != c.free_blocks.end(); ++it) {
				if (it->second < need) continue;
				const size_t off = it->first, len = it->second;
				c.free_blocks.erase(it);
				if (len > need) c.free_blocks[off + need] = len - need;
				c.used += need;
				*p = c.base + off;
				A.live[*p] = need;
				return hipSuccess;
			}
		}
		if (attempt == 1) break;
		Chunk c;
		if (hipMalloc((void **)&c.base, chunk) != hipSuccess) {
			(void)hipGetLastError(); // (not an error of the caller's: the request itself may still fit)
			break;
		}
		c.size = chunk;
		c.free_blocks[0] = chunk;
		A.chunks.push_back(std::move(c));
	}
	return hipMalloc(p, bytes);
}

// dev_free(p, false): the caller has waited for the device already (a handle's buffers are freed together: one wait, not ten)
inline hipError_t dev_free(void *p, bool wait = true) {
	if (!p) return hipSuccess;
	int cur = 0;
	(void)hipGetDevice(&cur);
	// the block's own arena: the current device's as a rule, but a handle may be freed while another device is current
	for (int probe = 0; probe < 64; ++probe) {
		const int dev = probe == 0 ? cur : (probe <= cur ? probe - 1 : probe);
		Arena &A = of_device(dev);
		std::unique_lock<std::mutex> lock(A.mu);
		auto it = A.live.find(p);
		if (it == A.live.end()) continue;
		const size_t len = it->second;
		A.live.erase(it);
		lock.unlock();
		if (wait) {
			if (dev != cur) (void)hipSetDevice(dev);
			(void)hipDeviceSynchronize(); // as hipFree: nothing in flight uses the block when it is handed out again
			if (dev != cur) (void)hipSetDevice(cur);
		}
		lock.lock();
		for (Chunk &c : A.chunks) {
			if ((char *)p < c.base || (char *)p >= c.base + c.size) continue;
			size_t off = (size_t)((char *)p - c.base), n = len;
			auto next = c.free_blocks.lower_bound(off);
			if (next != c.free_blocks.end() && off + n == next->first) { // joins the free block behind it
				n += next->second;
				next = c.free_blocks.erase(next);
			}
			if (next != c.free_blocks.begin()) { // and the one before it
				auto prev = std::prev(next);
				if (prev->first + prev->second == off) {
					off = prev->first, n += prev->second;
					c.free_blocks.erase(prev);
				}
			}
			c.free_blocks[off] = n;
			c.used -= len;
			return hipSuccess;
		}
		return hipSuccess; // (unreachable: a live block lies in a chunk)
	}
	return hipFree(p);
}

// the contexts on a device share its arena; the last one to go releases the chunks nobody holds a block of
inline void retain(int dev) {
	Arena &A = of_device(dev);
	std::lock_guard<std::mutex> lock(A.mu);
	++A.contexts;
}

inline size_t keep_bytes() { // ANDI_ARENA_KEEP: MiB of unused chunks that outlive a device's last context; 0: none
	// (the switch was a boolean once: 1 ... 15 -- "true" in whatever spelling -- mean the default, 8192, not a few MiB)
	const char *e = andi_knob(KNOB_ARENA_KEEP);
	const long mb = e ? atol(e) : 8192;
	return mb >= 1 && mb < 16 ? (size_t)8192 << 20 : mb > 0 ? (size_t)mb << 20 : (size_t)0;
}

inline bool has_chunks(int dev) {
	Arena &A = of_device(dev);
	std::lock_guard<std::mutex> lock(A.mu);
	return !A.chunks.empty();
}

inline bool any_chunks() {
	for (int d = 0; d < 64; ++d)
		if (has_chunks(d)) return true;
	return false;
}

// the chunks of a device nobody holds a block of go back to the driver; returns the bytes given back
// (the device must be current.  The chunks leave the arena under its lock; the wait for the device and the hipFrees --
// slow calls -- happen outside it, so that other contexts' allocations are not held up)
inline size_t trim(int dev) {
	Arena &A = of_device(dev);
	std::vector<Chunk> gone;
	{
		std::lock_guard<std::mutex> lock(A.mu);
		for (size_t i = 0; i < A.chunks.size();) {
			if (A.chunks[i].used == 0) {
				gone.push_back(std::move(A.chunks[i]));
				A.chunks.erase(A.chunks.begin() + (long)i);
			} else {
				++i;
			}
		}
	}
	if (gone.empty()) return 0;
	(void)hipDeviceSynchronize(); // (as hipFree: nothing in flight lies in a chunk that goes)
	size_t freed = 0;
	for (Chunk &c : gone) {
		freed += c.size;
		(void)hipFree(c.base);
	}
	return freed;
}

inline void release(int dev) {
	Arena &A = of_device(dev);
	std::lock_guard<std::mutex> lock(A.mu);
	if (--A.contexts > 0) return;
	A.contexts = 0;
	// What stays with the arena after a device's last context (andi_hip_trim or the process's end releases it): at most
	// ANDI_ARENA_KEEP MiB (default 8192 -- the 6.5 GB of a 29-genome job, whose hipMalloc costs 0.3 s per call on some
	// boxes; 0: nothing); a large job's memory goes back to the driver with its last context
	size_t keep = keep_bytes(), kept = 0;
	for (size_t i = 0; i < A.chunks.size();) {
		if (A.chunks[i].used == 0 && kept + A.chunks[i].size > keep) {
			(void)hipFree(A.chunks[i].base);
			A.chunks.erase(A.chunks.begin() + (long)i);
		} else {
			if (A.chunks[i].used == 0) kept += A.chunks[i].size;
			++i;
		}
	}
}

} // namespace andi_arena
