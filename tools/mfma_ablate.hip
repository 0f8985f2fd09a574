// tools/mfma_ablate.hip -- where does score_mfma_kernel spend its time?
// Times the product kernel (ABL = 0) and variants with one part removed (wrong results; see the
// ABL switches in kern_score_mfma.h) on synthetic genotypes of the bench workload's shape.
//   make -C saigegds_amd/csrc ablate && ./tools/mfma_ablate [N=430000] [M=50000] [reps=5]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <cmath>
#include <vector>

#define WAVE 64
#define LO_MASK 0x55555555u
#include "../saigegds_amd/csrc/kern_synth.h"
#define MF_KERNEL_ONLY
#define SGX_MAX_COEFF 16
#include "../saigegds_amd/csrc/kern_score_mfma.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int ABL, bool WIDE = false, int NAF = 4>
static float run(const uint8_t *G, size_t bpv, int M, const MfTab &tb, dim3 grid, int tps, int *acc, int reps)
{
	grid.x = (M + 16 * NAF * MF_WAVES - 1) / (16 * NAF * MF_WAVES);
	hipEvent_t a, b;
	(void)hipEventCreate(&a); (void)hipEventCreate(&b);
	const size_t lds = (size_t)2 * 16 * 64 * 16 + ((ABL & 512) ? 16 * 1024 : 0);
	float best = 1e30f;
	for (int r = 0; r < reps + 1; r++) {
		(void)hipMemsetAsync(acc, 0, (size_t)M * (64 + 48) * sizeof(int), 0);
		(void)hipEventRecord(a, 0);
		hipLaunchKernelGGL((score_mfma_kernel<3, true, WIDE, ABL, NAF>), grid, dim3(WAVE * MF_WAVES), lds, 0, G, bpv, M, tb, tps, acc, 64 + 48);
		(void)hipEventRecord(b, 0);
		(void)hipEventSynchronize(b);
		float ms = 0;
		(void)hipEventElapsedTime(&ms, a, b);
		if (r > 0) best = std::min(best, ms);
	}
	return best;
}

#ifndef MF_STAMP_EXTRA
#define MF_STAMP_EXTRA 0
#endif
int main(int argc, char **argv)
{
	const int N = argc > 1 ? atoi(argv[1]) : 430000, M = argc > 2 ? atoi(argv[2]) : 50000, reps = argc > 3 ? atoi(argv[3]) : 5;
	const size_t bpv = (size_t)((N + 511) / 512) * 128;
	uint8_t *G, *Fl; uint32_t *thr; int *acc;
	MfTab tb{};
	const int NCOL = 64, NACC = 64 + 48;   // the K = 3 layout: 3 value fragments + the bit-1 fragment
	tb.ntile = 2 * ((N + 511) / 512);
	CK(hipMalloc((void **)&G, (size_t)M * bpv));
	CK(hipMalloc((void **)&thr, (size_t)M * 3 * sizeof(uint32_t)));
	const size_t dbg_bytes = (size_t)64 << 20;
	CK(hipMalloc((void **)&acc, (size_t)M * NACC * sizeof(int) + dbg_bytes));
	const size_t flb = (size_t)tb.ntile * 16 * NCOL * 16;
	CK(hipMalloc((void **)&Fl, flb));
	std::vector<uint8_t> hf(flb);
	uint64_t x = 12345;
	for (auto &v : hf) { x = splitmix64(x); v = (uint8_t)x; }
	CK(hipMemcpy(Fl, hf.data(), flb, hipMemcpyHostToDevice));
	tb.Fl = Fl;
	std::vector<uint32_t> ht((size_t)M * 3);
	for (int j = 0; j < M; j++) {       // same law as saigegds_amd/synth.py: MAF 10^U(-3.3,-0.3), missing 1e-3
		x = splitmix64(x);
		const double p = std::pow(10.0, -3.3 + 3.0 * ((x >> 11) / 9007199254740992.0));
		ht[3 * j] = (uint32_t)std::min(4294967295.0, std::floor((1 - p) * (1 - p) * 4294967296.0));
		ht[3 * j + 1] = (uint32_t)std::min(4294967295.0, std::floor((1 - p * p) * 4294967296.0));
		ht[3 * j + 2] = (uint32_t)(1e-3 * 4294967296.0);
	}
	CK(hipMemcpy(thr, ht.data(), ht.size() * 4, hipMemcpyHostToDevice));
	hipLaunchKernelGGL(synth2b_kernel, dim3(8, M), dim3(256), 0, 0, G, bpv, N, (size_t)M, (uint64_t)0, (uint64_t)7, thr);
	CK(hipDeviceSynchronize());
	hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, 0));
	const int vt = (M + MF_VPB - 1) / MF_VPB;
	int sk = std::max(1, (pr.multiProcessorCount * (8 / MF_WAVES) * 4 + vt / 2) / vt);
	if (sk >= 6) sk = (sk + 7) & ~7;
	if (getenv("SK")) sk = atoi(getenv("SK"));
	sk = std::min(sk, tb.ntile);
	int tps = (tb.ntile + sk - 1) / sk;
	tps += tps & 1;
	sk = (tb.ntile + tps - 1) / tps;
	const dim3 grid(vt, sk);
	printf("N=%d M=%d grid=(%d,%d) tiles/split=%d bytes=%.3f GB\n", N, M, vt, sk, tps, (double)M * bpv / 1e9);
#define RUN(A, what) { const float ms = run<A>(G, bpv, M, tb, grid, tps, acc, reps); \
	printf("ABL=%2d %-34s %7.3f ms  %6.0f GB/s\n", A, what, ms, (double)M * bpv / ms / 1e6); }
	RUN(0, "product kernel")
	RUN(1, "- missing plane")
	RUN(2, "- bit-1 MFMA")
	RUN(4, "- unpack")
	RUN(16, "- A loads")
	RUN(32, "- B DMA")
	RUN(48, "- A loads, B DMA")
	RUN(63, "value-plane MFMA only")
	RUN(64, "memory system only")
	RUN(96, "A loads + barriers only")
	RUN(2048, "loads of a tile in one burst")
#define RUNW(A, what) { const float ms = run<A, true>(G, bpv, M, tb, grid, tps, acc, reps); \
	printf("ABL=%2d wide rows: %-22s %7.3f ms  %6.0f GB/s\n", A, what, ms, (double)M * bpv / ms / 1e6); }
	if (tps % 2 == 0 && tb.ntile % 2 == 0) {
#define RUN3(A, W, what) { const float ms = run<A, W, 3>(G, bpv, M, tb, grid, tps, acc, reps); \
	printf("ABL=%2d 3 A fragments, 3 waves/SIMD%s: %-22s %7.3f ms  %6.0f GB/s\n", A, W ? ", wide" : "", what, ms, (double)M * bpv / ms / 1e6); }
#define RUN2(A, W, what) { const float ms = run<A, W, 2>(G, bpv, M, tb, grid, tps, acc, reps); \
	printf("ABL=%2d 2 A fragments, 4 waves/SIMD%s: %-22s %7.3f ms  %6.0f GB/s\n", A, W ? ", wide" : "", what, ms, (double)M * bpv / ms / 1e6); }
		RUN2(0, false, "product kernel")
		RUN2(1, false, "- missing plane")
		RUN2(64, false, "memory system only")
		RUN2(0, true, "product kernel")
		RUN3(0, false, "product kernel")
		RUN3(1, false, "- missing plane")
		RUN3(64, false, "memory system only")
		RUN3(512, false, "tiled row loads")
		RUN3(513, false, "tiled, - missing plane")
		RUN3(576, false, "tiled, memory system only")
		RUN3(0, true, "product kernel")
		RUN3(1, true, "- missing plane")
		RUN3(64, true, "memory system only")
		RUNW(0, "product kernel")
		RUNW(1, "- missing plane")
		RUNW(64, "memory system only")
		RUNW(96, "A loads + barriers")
		RUNW(2048, "loads in one burst")
	}
	{   // where a wave's cycles go (s_memtime stamps; the stamps themselves cost a few %)
		const float ms = run<1024 + MF_STAMP_EXTRA>(G, bpv, M, tb, grid, tps, acc, 1);
		const size_t nw = (size_t)grid.x * grid.y * MF_WAVES;
		std::vector<unsigned long long> d(nw * 4);
		CK(hipMemcpy(d.data(), (char *)acc + (size_t)M * NACC * sizeof(int), nw * 32, hipMemcpyDeviceToHost));
		double w = 0, i = 0, c = 0, n = 0;
		for (size_t k = 0; k < nw; k++) { w += d[4 * k]; i += d[4 * k + 1]; c += d[4 * k + 2]; n += d[4 * k + 3]; }
		printf("stamped run %.3f ms: per wave and tile  barrier+wait %.0f  issue %.0f  compute %.0f cycles (tiles/wave %.1f)\n",
			ms, w / n, i / n, c / n, n / nw);
	}
	return 0;
}
