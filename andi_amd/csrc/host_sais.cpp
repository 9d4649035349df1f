// host_sais.cpp — suffix array construction on the host.
//
// Replaces the single third-party call on the path: divsufsort() at
// src/esa.c:303 (libdivsufsort, saidx_t = int32_t; not installed in this
// image).  A suffix array is unique, so any correct sorter gives the same
// downstream bits.  This is a linear-time induced-sorting construction
// (SA-IS, Nong/Zhang/Chan 2009) written for this project: re-entrant, no
// globals, unsigned-byte order, virtual sentinel smaller than every symbol.
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "andi_hip.h"

namespace {

struct TypeBits {
	std::vector<uint64_t> w;
	explicit TypeBits(size_t n) : w((n + 63) / 64, 0) {}
	inline bool get(int32_t i) const { return (w[(size_t)i >> 6] >> (i & 63)) & 1; }
	inline void set(int32_t i) { w[(size_t)i >> 6] |= (uint64_t)1 << (i & 63); }
};

template <typename Sym>
void bucket_bounds(const Sym *T, int32_t n, int32_t K, std::vector<int32_t> &cnt) {
	cnt.assign((size_t)K + 1, 0);
	for (int32_t i = 0; i < n; ++i) cnt[(size_t)T[i] + 1]++;
	for (int32_t c = 0; c < K; ++c) cnt[(size_t)c + 1] += cnt[c];
}

// One full round of induced sorting.  On entry SA holds the seeded LMS
// suffixes at the tails of their buckets and -1 elsewhere.
template <typename Sym>
void induce(const Sym *T, int32_t *SA, int32_t n, int32_t K, const TypeBits &isS,
			const std::vector<int32_t> &bounds, std::vector<int32_t> &ptr) {
	// L-type pass, left to right.  The virtual sentinel suffix comes first and
	// induces position n-1 (always L-type).
	for (int32_t c = 0; c < K; ++c) ptr[c] = bounds[c];
	SA[ptr[T[n - 1]]++] = n - 1;
	for (int32_t i = 0; i < n; ++i) {
		int32_t j = SA[i];
		if (j > 0 && !isS.get(j - 1)) SA[ptr[T[j - 1]]++] = j - 1;
	}
	// every S-type suffix is re-induced below; drop the seeds
	for (int32_t i = 0; i < n; ++i) {
		int32_t j = SA[i];
		if (j >= 0 && isS.get(j)) SA[i] = -1;
	}
	// S-type pass, right to left.
	for (int32_t c = 0; c < K; ++c) ptr[c] = bounds[c + 1];
	for (int32_t i = n - 1; i >= 0; --i) {
		int32_t j = SA[i];
		if (j > 0 && isS.get(j - 1)) SA[--ptr[T[j - 1]]] = j - 1;
	}
}

template <typename Sym>
void sais(const Sym *T, int32_t *SA, int32_t n, int32_t K) {
	if (n <= 0) return;
	if (n == 1) {
		SA[0] = 0;
		return;
	}
	TypeBits isS((size_t)n);
	// position n-1 is L-type (followed by the sentinel)
	for (int32_t i = n - 2; i >= 0; --i) {
		if (T[i] < T[i + 1] || (T[i] == T[i + 1] && isS.get(i + 1))) isS.set(i);
	}
	auto is_lms = [&](int32_t i) { return i > 0 && isS.get(i) && !isS.get(i - 1); };

	std::vector<int32_t> bounds, ptr((size_t)K);
	bucket_bounds(T, n, K, bounds);

	// Step 1: sort LMS substrings by one round of induced sorting.
	for (int32_t i = 0; i < n; ++i) SA[i] = -1;
	for (int32_t c = 0; c < K; ++c) ptr[c] = bounds[c + 1];
	int32_t n_lms = 0;
	for (int32_t i = n - 1; i > 0; --i) {
		if (is_lms(i)) {
			SA[--ptr[T[i]]] = i;
			++n_lms;
		}
	}
	induce(T, SA, n, K, isS, bounds, ptr);

	if (n_lms == 0) return; // text is one L-run (e.g. "dcba"): already sorted

	// Step 2: name the LMS substrings in their sorted order.
	std::vector<int32_t> lms_sorted;
	lms_sorted.reserve((size_t)n_lms);
	for (int32_t i = 0; i < n; ++i)
		if (SA[i] >= 0 && is_lms(SA[i])) lms_sorted.push_back(SA[i]);

	std::vector<int32_t> name_of((size_t)n / 2 + 1, -1); // indexed by pos/2 (LMS positions are ≥2 apart)
	int32_t names = 0, prev = -1;
	for (int32_t k = 0; k < n_lms; ++k) {
		int32_t p = lms_sorted[k];
		bool diff = prev < 0;
		if (!diff) {
			// compare LMS substrings starting at prev and p
			for (int32_t d = 0;; ++d) {
				int32_t a = prev + d, b = p + d;
				if (a >= n || b >= n) { // one runs into the sentinel
					diff = true;
					break;
				}
				if (T[a] != T[b] || isS.get(a) != isS.get(b)) {
					diff = true;
					break;
				}
				if (d > 0 && (is_lms(a) || is_lms(b))) {
					diff = !(is_lms(a) && is_lms(b));
					break;
				}
			}
		}
		if (diff) ++names;
		name_of[(size_t)p / 2] = names - 1;
		prev = p;
	}

	// Step 3: order of the LMS suffixes, recursively if names collide.
	std::vector<int32_t> lms_pos; // text order
	lms_pos.reserve((size_t)n_lms);
	for (int32_t i = 1; i < n; ++i)
		if (is_lms(i)) lms_pos.push_back(i);

	std::vector<int32_t> order((size_t)n_lms);
	if (names < n_lms) {
		std::vector<int32_t> T1((size_t)n_lms), SA1((size_t)n_lms);
		for (int32_t k = 0; k < n_lms; ++k) T1[k] = name_of[(size_t)lms_pos[k] / 2];
		sais<int32_t>(T1.data(), SA1.data(), n_lms, names);
		for (int32_t k = 0; k < n_lms; ++k) order[k] = lms_pos[SA1[k]];
	} else {
		for (int32_t k = 0; k < n_lms; ++k) order[name_of[(size_t)lms_pos[k] / 2]] = lms_pos[k];
	}
	std::vector<int32_t>().swap(name_of);

	// Step 4: seed the sorted LMS suffixes and induce everything.
	for (int32_t i = 0; i < n; ++i) SA[i] = -1;
	for (int32_t c = 0; c < K; ++c) ptr[c] = bounds[c + 1];
	for (int32_t k = n_lms - 1; k >= 0; --k) {
		int32_t p = order[k];
		SA[--ptr[T[p]]] = p;
	}
	induce(T, SA, n, K, isS, bounds, ptr);
}

} // namespace

// libdivsufsort -- the sorter the reference links (configure.ac:33-38, called at src/esa.c:303; saidx_t = int32_t) -- is an
// OPTIONAL link: where the host has it, it is loaded at first use and does the sorting (a suffix array is unique: the bits
// downstream are the same); where it is absent (this project's image) or ANDI_HIP_NO_DIVSUFSORT is set, the SA-IS above does.
#include <dlfcn.h>
#include <mutex>
namespace {
typedef int32_t (*divsufsort_fn)(const unsigned char *, int32_t *, int32_t);
divsufsort_fn g_divsufsort = nullptr;
const char *g_sorter = "SA-IS (built in)";
void resolve_sorter() {
	static std::once_flag once;
	std::call_once(once, [] {
		if (getenv("ANDI_HIP_NO_DIVSUFSORT")) return;
		const char *names[] = {"libdivsufsort.so.3", "libdivsufsort.so"};
		for (const char *nm : names) {
			void *h = dlopen(nm, RTLD_NOW | RTLD_LOCAL);
			if (!h) continue;
			if (auto f = (divsufsort_fn)dlsym(h, "divsufsort")) {
				g_divsufsort = f, g_sorter = "libdivsufsort (divsufsort, loaded at run time)";
				return;
			}
			dlclose(h);
		}
	});
}
} // namespace

extern "C" const char *andi_hip_suffix_sorter(void) {
	resolve_sorter();
	return g_sorter;
}

extern "C" int andi_hip_suffix_array(const unsigned char *T, int32_t *SA, int32_t n) {
	if (!T || !SA || n < 0) return -1;
	resolve_sorter();
	if (g_divsufsort && n > 0) return g_divsufsort(T, SA, n) == 0 ? 0 : -2; // src/esa.c:303-304
	try {
		sais<unsigned char>(T, SA, n, 256);
	} catch (...) {
		return -2;
	}
	return 0;
}
