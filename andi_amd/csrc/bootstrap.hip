// bootstrap.hip — K8: pairwise bootstrap of the substitution-count matrix.
//
// calculate_bootstrap (src/process.c:289-321) resamples, for every unordered pair
// i < j, the summed counts model_average(M(i,j), M(j,i)) from a multinomial with
// N = total count and p = counts / N (model_bootstrap, src/model.c:222-232,
// gsl_ran_multinomial), mirrors the result to (j,i) and sets the diagonal to
// {counts[0] = 1, seq_len = 1}.  The reference draws from a global GSL generator
// seeded with time(NULL) and has no test for it, so there is nothing to be
// bit-identical to ("parity unpinned"); what is kept is the distribution.  Here
// every (replicate, pair) owns a counter-based Philox stream, so the result is a
// pure function of (seed, replicate, i, j) whatever the launch geometry, and all
// replicates of a matrix are one launch.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "andi_hip.h"
#include "bootstrap.h"

namespace {

// Philox4x32-10 (Salmon et al., SC'11)
struct Philox {
	uint32_t ctr[4], key[2], out[4];
	int have;
	__device__ Philox(uint64_t seed, uint64_t stream, uint64_t sub) : have(0) {
		ctr[0] = 0, ctr[1] = (uint32_t)sub, ctr[2] = (uint32_t)stream, ctr[3] = (uint32_t)(stream >> 32);
		key[0] = (uint32_t)seed, key[1] = (uint32_t)(seed >> 32);
	}
	__device__ void round(uint32_t *c, const uint32_t *k) {
		uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
		uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k[0], n1 = (uint32_t)p1;
		uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k[1], n3 = (uint32_t)p0;
		c[0] = n0, c[1] = n1, c[2] = n2, c[3] = n3;
	}
	__device__ void refill() {
		uint32_t c[4] = {ctr[0], ctr[1], ctr[2], ctr[3]}, k[2] = {key[0], key[1]};
		for (int r = 0; r < 10; ++r) {
			round(c, k);
			k[0] += 0x9E3779B9u, k[1] += 0xBB67AE85u;
		}
		out[0] = c[0], out[1] = c[1], out[2] = c[2], out[3] = c[3];
		ctr[0]++; // 2^32 blocks per stream is far more than one pair ever needs
		have = 4;
	}
	__device__ double uniform() { // (0, 1), 53 random bits
		if (have < 2) refill();
		uint64_t hi = out[--have], lo = out[--have];
		uint64_t bits = ((hi << 32) | lo) >> 11;
		return ((double)bits + 0.5) * (1.0 / 9007199254740992.0);
	}
};

__device__ double stirling_tail(double k) { // log(k!) - [Stirling's leading terms]
	const double tab[10] = {0.0810614667953272, 0.0413406959554092, 0.0276779256849983, 0.02079067210376509,
							0.0166446911898211, 0.0138761288230707, 0.0118967099458917, 0.0104112652619720,
							0.00925546218271273, 0.00833056343336287};
	if (k <= 9) return tab[(int)k];
	double kp1sq = (k + 1) * (k + 1);
	return (1.0 / 12 - (1.0 / 360 - 1.0 / 1260 / kp1sq) / kp1sq) / (k + 1);
}

// Binomial(n, p), exact: waiting-time method for small n*p, BTRS (Hoermann 1993,
// "The generation of binomial random variates") otherwise.
__device__ uint64_t binomial(Philox &g, uint64_t n, double p) {
	if (n == 0 || p <= 0.0) return 0;
	if (p >= 1.0) return n;
	const bool flip = p > 0.5;
	if (flip) p = 1.0 - p;
	uint64_t k;
	const double nd = (double)n;
	if (nd * p < 10.0) {
		const double lq = log1p(-p);
		double sum = 0.0;
		k = 0;
		for (;;) {
			sum += ceil(log(g.uniform()) / lq);
			if (sum > nd) break;
			++k;
		}
	} else {
		const double q = 1.0 - p, spq = sqrt(nd * p * q);
		const double b = 1.15 + 2.53 * spq, a = -0.0873 + 0.0248 * b + 0.01 * p;
		const double c = nd * p + 0.5, vr = 0.92 - 4.2 / b, alpha = (2.83 + 5.1 / b) * spq;
		const double r = p / q, m = floor((nd + 1) * p);
		for (;;) {
			double u = g.uniform() - 0.5, v = g.uniform();
			double us = 0.5 - fabs(u);
			double kd = floor((2 * a / us + b) * u + c);
			if (kd < 0 || kd > nd) continue;
			if (us >= 0.07 && v <= vr) {
				k = (uint64_t)kd;
				break;
			}
			v = log(v * alpha / (a / (us * us) + b));
			double ub = (m + 0.5) * log((m + 1) / (r * (nd - m + 1))) +
						(nd + 1) * log((nd - m + 1) / (nd - kd + 1)) +
						(kd + 0.5) * log(r * (nd - kd + 1) / (kd + 1)) + stirling_tail(m) +
						stirling_tail(nd - m) - stirling_tail(kd) - stirling_tail(nd - kd);
			if (v <= ub) {
				k = (uint64_t)kd;
				break;
			}
		}
	}
	return flip ? n - k : k;
}

// One thread per (replicate, unordered pair incl. diagonal).
__global__ __launch_bounds__(256) void k_bootstrap(const andi_hip_model *__restrict__ M,
												   andi_hip_model *__restrict__ B, uint32_t n,
												   uint32_t replicates, uint64_t seed) {
	const uint64_t per = (uint64_t)n * (n + 1) / 2;
	uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (gid >= per * replicates) return;
	const uint32_t rep = (uint32_t)(gid / per);
	uint64_t t = gid % per;
	// unrank t -> (i, j), i <= j, row-major over the upper triangle
	uint32_t i = 0;
	{
		// rows have n, n-1, ... entries; solve by a short search from an estimate
		double est = ((2.0 * n + 1) - sqrt((2.0 * n + 1) * (2.0 * n + 1) - 8.0 * (double)t)) / 2.0;
		i = est > 0 ? (uint32_t)est : 0;
		auto start = [n](uint64_t r) { return r * n - r * (r - 1) / 2; };
		while (i > 0 && start(i) > t) --i;
		while (i + 1 < n && start(i + 1) <= t) ++i;
		t -= start(i);
	}
	const uint32_t j = i + (uint32_t)t;
	andi_hip_model *out = B + (size_t)rep * n * n;
	andi_hip_model res;
	if (i == j) { // src/process.c:303-306
		for (int c = 0; c < 16; ++c) res.counts[c] = 0;
		res.counts[0] = 1, res.seq_len = 1;
		out[(size_t)i * n + i] = res;
		return;
	}
	const andi_hip_model a = M[(size_t)i * n + j], b = M[(size_t)j * n + i];
	uint64_t counts[16], total = 0;
	for (int c = 0; c < 16; ++c) { // model_average, src/model.c:39-46
		counts[c] = (uint64_t)(uint32_t)(a.counts[c] + b.counts[c]);
		total += counts[c];
	}
	res.seq_len = a.seq_len + b.seq_len;
	// multinomial by conditional binomials (the construction gsl_ran_multinomial uses)
	Philox g(seed, (uint64_t)rep, (uint64_t)i * n + j);
	uint64_t left = total, mass = total;
	for (int c = 0; c < 16; ++c) {
		uint64_t draw = 0;
		if (counts[c] > 0 && left > 0)
			draw = counts[c] >= mass ? left : binomial(g, left, (double)counts[c] / (double)mass);
		res.counts[c] = (uint32_t)draw;
		left -= draw;
		mass -= counts[c];
	}
	out[(size_t)i * n + j] = res;
	out[(size_t)j * n + i] = res; // src/process.c:314
}

} // namespace

hipError_t andi_launch_bootstrap(const andi_hip_model *M_dev, andi_hip_model *B_dev, uint32_t n,
								 uint32_t replicates, uint64_t seed, hipStream_t st) {
	const uint64_t items = (uint64_t)n * (n + 1) / 2 * replicates;
	if (items == 0) return hipSuccess;
	k_bootstrap<<<(unsigned)((items + 255) / 256), 256, 0, st>>>(M_dev, B_dev, n, replicates, seed);
	return hipGetLastError();
}
