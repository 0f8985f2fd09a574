// tools/score3_bench.hip -- validation and timing of score3_kernel (saigegds_amd/csrc/kern_score3.h) alone.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -o tools/score3_bench tools/score3_bench.hip
//   ./tools/score3_bench check            small shapes against a CPU sum (every instantiation below)
//   ./tools/score3_bench [N=430000] [M=50000] [reps=5]
// Inputs are random tiled blocks (codes 0/1/2, 1e-3 missing) and random limb tiles: the kernel's time does
// not depend on the values.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <vector>

#define S3_KERNEL_ONLY
#include "../saigegds_amd/csrc/kern_score3.h"
#include "../saigegds_amd/csrc/kern_lists.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

static uint64_t sm64(uint64_t &x) { x += 0x9E3779B97F4A7C15ull; uint64_t z = x; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }

__global__ void fill_codes(uint32_t *dst, size_t ndw, uint64_t seed, uint32_t miss16 = 66)
{
	for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < ndw; i += (size_t)gridDim.x * blockDim.x) {
		uint64_t x = seed + i * 0x9E3779B97F4A7C15ull;
		uint32_t w = 0;
		for (int s = 0; s < 16; s++) {
			x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 27; x *= 0x94D049BB133111EBull; x ^= x >> 31;
			const uint32_t u = (uint32_t)(x >> 40) & 0xFFFF;       // 16-bit uniform
			const uint32_t code = u < miss16 ? 3u : (u < 50000 ? 0u : (u < 62000 ? 1u : 2u));
			w |= code << (2 * s);
		}
		dst[i] = w;
	}
}

struct Shape { int N; size_t M; };

template <int NBF, int NAF, int WAVES, int NLA, int NLB, int DA, int DB, int ABL, int NCB = 1, int NBUF = 2, int RM = 0, bool MISS = false>
static float run(const char *name, const uint8_t *A, const uint8_t *Fl, int ntile, size_t M, int wg_per_cu, int n_cu, int *out, size_t out_ints,
	int reps, S3Plan *plan_out = nullptr, size_t bpv = 0)
{
	const int grid = n_cu * wg_per_cu;
	constexpr int NCV = WAVES / NCB, NBW = MISS ? 2 * NBF - 1 : (NBF + NCB - 1) / NCB;
	const S3Plan pl = s3_plan(M, ntile, grid, NAF * NCV, bpv);
	const size_t need = (size_t)pl.ng * pl.ipg * WAVES * NAF * NBW * 256;
	if (need > out_ints) { fprintf(stderr, "%s: out buffer too small (%zu > %zu)\n", name, need, out_ints); exit(1); }
	if (plan_out) *plan_out = pl;
	const size_t lds = ((size_t)(DB + 1) * 4 * NBF + (size_t)(DA + 1) * NCV * NAF * (RM == 1 ? 2 : 1)) * 1024;
	if (lds > 163840) { printf("%-40s skipped: %zu B of LDS\n", name, lds); return 0; }
	auto kern = score3_kernel<NBF, NAF, WAVES, NLA, NLB, DA, DB, ABL, NCB, NBUF, RM, MISS>;
	static unsigned long long *stamps = nullptr;
	if (!stamps) CK(hipMalloc((void **)&stamps, 16 * 4096));
	CK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
	hipEvent_t a, b;
	CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
	float best = 1e30f;
	for (int rr = 0; rr < reps + 1; rr++) {
		CK(hipEventRecord(a, 0));
		hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * (WAVES + NLA + NLB)), lds, 0, A, Fl, pl, out, stamps);
		CK(hipEventRecord(b, 0));
		CK(hipEventSynchronize(b));
		CK(hipGetLastError());
		float ms = 0;
		CK(hipEventElapsedTime(&ms, a, b));
		if (rr > 0 || reps == 0) best = std::min(best, ms);
	}
	CK(hipEventDestroy(a)); CK(hipEventDestroy(b));
	if (reps > 0) {
		const double bytes = (double)M * (ntile * 64.0);
		double ghz = 0;
		if (ABL & 16) {
			std::vector<unsigned long long> hs(2 * grid);
			CK(hipMemcpy(hs.data(), stamps, hs.size() * 8, hipMemcpyDeviceToHost));
			std::vector<double> c;
			for (int b2 = 0; b2 < grid; b2++) if (hs[2 * b2 + 1]) c.push_back((double)hs[2 * b2] / (double)hs[2 * b2 + 1] * 0.1);
			std::sort(c.begin(), c.end());
			if (!c.empty()) ghz = c[c.size() / 2];
		}
		printf("%-40s %s NBF=%2d NAF=%d NC=%d/%d NL=%d+%d D=%d/%d buf %d ABL=%2d  items/grp=%4d f=%2d  %7.3f ms  %6.0f GB/s  %.3f of 8 TB/s",
			name, MISS ? "rows3" : RM == 2 ? "lines" : RM ? "rows " : "tiles", NBF, NAF, WAVES, NCB, NLA, NLB, DA, DB, NBUF, ABL, pl.ipg, pl.f, best, bytes / best / 1e6, bytes / best / 1e6 / 8000.0);
		if (ABL & 16) printf("  clock %.3f GHz", ghz);
		printf("\n");
		fflush(stdout);
	}
	return best;
}

// CPU check: sums of the item slabs per (variant, column) against the direct sum
template <int NBF, int NAF, int WAVES, int NLA, int NLB, int DA, int DB, int NCB = 1, int NBUF = 2, int RM = 0, bool MISS = false>
static int check(const char *name, int N, size_t M, int wg_per_cu, int n_cu, size_t row_pad = 0, uint32_t miss16 = 300)
{
	const int ntile = 2 * ((N + 511) / 512);
	const size_t nfrag = (M + 15) / 16, abytes = nfrag * (size_t)ntile * 1024;
	const int NCOL = 16 * NBF;
	const size_t flbytes = (size_t)ntile * 16 * NCOL * 16;
	std::vector<uint8_t> hA(abytes), hF(flbytes);
	uint64_t x = 77 + N + M;
	for (auto &v : hF) v = (uint8_t)sm64(x);
	// codes: variant v, sample s
	std::vector<uint8_t> code((size_t)nfrag * 16 * ((size_t)ntile * 256), 0);
	for (size_t v = 0; v < nfrag * 16; v++)
		for (size_t s = 0; s < (size_t)ntile * 256; s++) {
			const uint32_t u = (uint32_t)(sm64(x) & 0xFFFF);
			code[v * ntile * 256 + s] = (v < M && s < (size_t)N) ? (u < miss16 ? 3 : (u < 40000 ? 0 : (u < 56000 ? 1 : 2))) : (uint8_t)(sm64(x) & 3);   // padding holds garbage codes
		}
	// padding samples (>= N) must see zero limbs, as sgx_init writes them
	for (int t = 0; t < ntile; t++)
		for (int g16 = 0; g16 < 16; g16++)
			for (int e = 0; e < 16; e++) {
				const size_t s = (size_t)t * 256 + g16 * 16 + e;
				if (s >= (size_t)N)
					for (int c = 0; c < NCOL; c++) hF[((size_t)(t * 16 + g16) * NCOL + c) * 16 + s3_pos(e)] = 0;
			}
	for (size_t v = 0; v < nfrag * 16; v++)
		for (size_t p = 0; p < (size_t)ntile * 4; p++) {
			uint32_t w[4] = {0, 0, 0, 0};
			for (int u = 0; u < 4; u++)
				for (int e = 0; e < 16; e++) w[u] |= (uint32_t)code[v * ntile * 256 + p * 64 + u * 16 + e] << (2 * e);
			memcpy(&hA[s3_piece_off(v, p, ntile)], w, 16);
		}
	uint8_t *dA, *dF; int *dO;
	// row-major form: exactly M rows (no padding rows: a read past the last row would fault or show)
	const size_t bpv = (size_t)ntile * 64 + row_pad;
	if (RM) {
		hA.assign(M * bpv, 0xFF);                      // (the bytes between the rows hold missing codes: never read)
		for (size_t v = 0; v < M; v++)
			for (size_t p = 0; p < (size_t)ntile * 4; p++) {
				uint32_t w[4] = {0, 0, 0, 0};
				for (int u = 0; u < 4; u++)
					for (int e = 0; e < 16; e++) w[u] |= (uint32_t)code[v * ntile * 256 + p * 64 + u * 16 + e] << (2 * e);
				memcpy(&hA[v * bpv + p * 16], w, 16);
			}
	}
	const size_t abytes_dev = RM ? M * bpv : abytes;
	CK(hipMalloc((void **)&dA, abytes_dev)); CK(hipMalloc((void **)&dF, flbytes));
	CK(hipMemcpy(dA, hA.data(), abytes_dev, hipMemcpyHostToDevice));
	CK(hipMemcpy(dF, hF.data(), flbytes, hipMemcpyHostToDevice));
	const int grid = n_cu * wg_per_cu;
	constexpr int NCV = WAVES / NCB, NBW = MISS ? 2 * NBF - 1 : (NBF + NCB - 1) / NCB;
	const S3Plan pl0 = s3_plan(M, ntile, grid, NAF * NCV);
	const size_t oints = (size_t)pl0.ng * pl0.ipg * WAVES * NAF * NBW * 256;
	CK(hipMalloc((void **)&dO, oints * 4));
	CK(hipMemset(dO, 0xCD, oints * 4));
	S3Plan pl;
	run<NBF, NAF, WAVES, NLA, NLB, DA, DB, 0, NCB, NBUF, RM, MISS>(name, dA, dF, ntile, M, wg_per_cu, n_cu, dO, oints, 0, &pl, bpv);
	std::vector<int> hO(oints);
	CK(hipMemcpy(hO.data(), dO, oints * 4, hipMemcpyDeviceToHost));
	long long bad = 0;
	for (size_t v = 0; v < M; v++) {
		const int vtile = (int)(v / (16 * (size_t)pl.fpw)), within = (int)(v % (16 * (size_t)pl.fpw));
		const int vg = within / (16 * NAF), f = (within / 16) % NAF, row = within % 16, kg = row / 4, reg = row % 4;
		for (int c = 0; c < NCOL; c++) {
			long long ref = 0;
			for (size_t s = 0; s < (size_t)ntile * 256; s++) {
				const int cd = code[v * ntile * 256 + s];
				const int t = (int)(s / 256), g16 = (int)(s % 256) / 16, e = (int)(s % 16);
				const int a = ((c >= NCOL - 16) ? (cd & 2) : cd) * s3_scale(e);
				ref += (long long)a * (int8_t)hF[((size_t)(t * 16 + g16) * NCOL + c) * 16 + s3_pos(e)];
			}
			long long got = 0;
			for (int g = 0; g < pl.ng; g++) {
				int first, count;
				s3_items_of(pl, vtile, g, first, count);
				for (int id = first; id < first + count; id++)
					got += hO[(((size_t)id * WAVES + (vg * NCB + (c / 16) / NBW)) * NAF + f) * NBW * 256 + (size_t)((c / 16) % NBW) * 256 + reg * 64 + kg * 16 + (c % 16)];
			}
			if (got != ref) { if (bad < 5) fprintf(stderr, "%s: variant %zu col %d: got %lld want %lld\n", name, v, c, got, ref); bad++; }
		}
		// the missing plane against the value columns (slots NBF .. 2 NBF - 2 of the slab)
		for (int c = 0; MISS && c < NCOL - 16; c++) {
			long long ref = 0;
			for (size_t s = 0; s < (size_t)ntile * 256; s++) {
				const int cd = code[v * ntile * 256 + s];
				const int t = (int)(s / 256), g16 = (int)(s % 256) / 16, e = (int)(s % 16);
				ref += (long long)((cd == 3) * s3_scale(e)) * (int8_t)hF[((size_t)(t * 16 + g16) * NCOL + c) * 16 + s3_pos(e)];
			}
			long long got = 0;
			for (int g = 0; g < pl.ng; g++) {
				int first, count;
				s3_items_of(pl, vtile, g, first, count);
				for (int id = first; id < first + count; id++)
					got += hO[(((size_t)id * WAVES + vg) * NAF + f) * NBW * 256 + (size_t)(NBF + c / 16) * 256 + reg * 64 + kg * 16 + (c % 16)];
			}
			if (got != ref) { if (bad < 5) fprintf(stderr, "%s: variant %zu missing-plane col %d: got %lld want %lld\n", name, v, c, got, ref); bad++; }
		}
	}
	printf("check %s %-30s N=%d M=%zu ntile=%d ng=%d wpg=%d rf=%d rem=%d f=%d: %s\n", MISS ? "rows3" : RM ? "rows " : "tiles", name, N, M, ntile, pl.ng, pl.wpg, pl.rf, pl.rem, pl.f, bad ? "FAILED" : "ok");
	CK(hipFree(dA)); CK(hipFree(dF)); CK(hipFree(dO));
	return bad ? 1 : 0;
}

// the one-pass list builder (kern_lists.h) on row-major rows: timing, and the lists against a CPU walk
static int lists_bench(const uint8_t *rows, size_t bpv, int N, size_t M, int ntile, int reps, bool verify)
{
	S3Lists L{};
	const size_t cap = std::max<size_t>(M * std::max<size_t>(64, (size_t)N / 128), (size_t)S3_NSUB * 256);
	L.idx_cap = (unsigned)cap; L.ld = M; L.nr = s3_nranges(ntile);
	s3_lists_setup(L, ntile, (M * (size_t)L.nr + 3) / 4);
	CK(hipMalloc((void **)&L.idx, cap * 4));
	CK(hipMalloc((void **)&L.cursor, S3_NSUB * S3_CURSOR_STRIDE * 4));
	CK(hipMalloc((void **)&L.lstart, S3_NR * M * 4)); CK(hipMalloc((void **)&L.lcnt, S3_NR * M * 4));
	CK(hipMalloc((void **)&L.nzp, S3_NR * M * 4)); CK(hipMalloc((void **)&L.n2p, S3_NR * M * 4));
	uint8_t *copy = nullptr;
	CK(hipMalloc((void **)&copy, M * bpv));
	hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
	const dim3 grid((unsigned)((M + 3) / 4), (unsigned)L.nr);
	for (int mode = 0; mode < 2; mode++) {
		float best = 1e30f;
		for (int r = 0; r < reps + 1; r++) {
			CK(hipMemset(L.cursor, 0, S3_NSUB * S3_CURSOR_STRIDE * 4));
			CK(hipEventRecord(a, 0));
			if (mode == 0) hipLaunchKernelGGL((s3_lists_kernel<8, false, false>), grid, dim3(256), 0, 0, rows, bpv, N, (int)M, 0, ntile, L, (uint8_t *)nullptr, (size_t)0);
			else hipLaunchKernelGGL((s3_lists_kernel<8, true, true>), grid, dim3(256), 0, 0, rows, bpv, N, (int)M, 0, ntile, L, copy, bpv);
			CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b)); CK(hipGetLastError());
			float ms; CK(hipEventElapsedTime(&ms, a, b));
			if (r > 0 || reps == 0) best = std::min(best, ms);
		}
		unsigned long long used = 0;
		{ std::vector<unsigned> hc(S3_NSUB * S3_CURSOR_STRIDE); CK(hipMemcpy(hc.data(), L.cursor, hc.size() * 4, hipMemcpyDeviceToHost)); for (int q = 0; q < S3_NSUB; q++) used += hc[q * S3_CURSOR_STRIDE]; }
		printf("lists kernel %-14s N=%d M=%zu: %7.3f ms  %6.0f GB/s read  (%llu entries listed, pool %zu)\n", mode ? "(+ copy)" : "(lists only)",
			N, M, best, (double)M * ntile * 64 / best / 1e6, used, cap);
	}
	int bad = 0;
	if (verify) {
		std::vector<uint8_t> hr(M * bpv), hc(M * bpv);
		CK(hipMemcpy(hr.data(), rows, M * bpv, hipMemcpyDeviceToHost));
		CK(hipMemcpy(hc.data(), copy, M * bpv, hipMemcpyDeviceToHost));
		std::vector<unsigned> st((size_t)L.nr * M), idx(cap); std::vector<int> cn((size_t)L.nr * M), nz((size_t)L.nr * M), n2((size_t)L.nr * M);
		CK(hipMemcpy(st.data(), L.lstart, st.size() * 4, hipMemcpyDeviceToHost));
		CK(hipMemcpy(cn.data(), L.lcnt, cn.size() * 4, hipMemcpyDeviceToHost));
		CK(hipMemcpy(nz.data(), L.nzp, nz.size() * 4, hipMemcpyDeviceToHost));
		CK(hipMemcpy(n2.data(), L.n2p, n2.size() * 4, hipMemcpyDeviceToHost));
		CK(hipMemcpy(idx.data(), L.idx, cap * 4, hipMemcpyDeviceToHost));
		for (size_t v = 0; v < M && bad < 5; v++) {
			if (memcmp(&hr[v * bpv], &hc[v * bpv], (size_t)ntile * 64)) { fprintf(stderr, "lists: copy of row %zu differs\n", v); bad++; }
			for (int g = 0; g < L.nr; g++) {
				std::vector<unsigned> want; int wz = 0, w2 = 0;
				const int s0 = s3_range_t0(g, ntile, L.nr) * 256, s1 = std::min(N, s3_range_t0(g + 1, ntile, L.nr) * 256);
				for (int s = s0; s < s1; s++) {
					const int c = (hr[v * bpv + s / 4] >> (2 * (s % 4))) & 3;
					if (c == 3) want.push_back((unsigned)s);
					wz += c != 0; w2 += c == 2;
				}
				const size_t e = (size_t)g * M + v;
				bool ok = cn[e] == (int)want.size() && nz[e] == wz && n2[e] == w2;
				if (ok) { std::vector<unsigned> got(idx.begin() + st[e], idx.begin() + st[e] + want.size()); std::sort(got.begin(), got.end()); ok = got == want; }
				if (!ok) { fprintf(stderr, "lists: variant %zu range %d: count %d want %zu, nz %d want %d, n2 %d want %d\n", v, g, cn[e], want.size(), nz[e], wz, n2[e], w2); bad++; }
			}
		}
		printf("lists check N=%d M=%zu: %s\n", N, M, bad ? "FAILED" : "ok");
	}
	CK(hipFree(L.idx)); CK(hipFree(L.cursor)); CK(hipFree(L.lstart)); CK(hipFree(L.lcnt)); CK(hipFree(L.nzp)); CK(hipFree(L.n2p)); CK(hipFree(copy));
	return bad;
}

// A stand-in for the other lane's cumulant pass: one workgroup of 512 threads per CU holding `lds` bytes of LDS and
// ~2 NV registers per lane, busy with FP64 FMAs for `us` microseconds.  Beside it: how fast does the list pass go?
template <int NV>
__global__ void __launch_bounds__(512) occupy_kernel(double *out, unsigned long long ticks, double seed)
{
	extern __shared__ double occ_lds[];
	double acc[NV];
#pragma unroll
	for (int i = 0; i < NV; i++) acc[i] = seed + i + threadIdx.x;
	occ_lds[threadIdx.x] = seed;
	const unsigned long long t0 = __builtin_readcyclecounter();
	while (__builtin_readcyclecounter() - t0 < ticks) {
#pragma unroll
		for (int i = 0; i < NV; i++) acc[i] = __builtin_fma(acc[i], 1.0000001, acc[(i + 1) % NV] * 1e-9);
	}
	double sum = occ_lds[(threadIdx.x + 1) & 511];
#pragma unroll
	for (int i = 0; i < NV; i++) sum += acc[i];
	if (sum == 12345.678) out[blockIdx.x] = sum;
}

template <int NV>
static void beside(const char *what, size_t lds, int n_cu, hipStream_t s_occ, hipStream_t s_list, const uint8_t *rows, size_t bpv, int N, size_t M, int ntile, S3Lists L, int P, const long long *Q, long long *part, double *dummy)
{
	CK(hipFuncSetAttribute((const void *)occupy_kernel<NV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
	hipEvent_t a, b, c, d; CK(hipEventCreate(&a)); CK(hipEventCreate(&b)); CK(hipEventCreate(&c)); CK(hipEventCreate(&d));
	for (int order = 0; order < 2; order++) {
		CK(hipDeviceSynchronize());
		const unsigned long long ticks = 100000000ull * 8 / 10000;      // 0.8 ms at the 100 MHz of s_memrealtime... (readcyclecounter: shader clock; see the printed time)
		auto occ = [&]() { CK(hipEventRecord(a, s_occ)); hipLaunchKernelGGL((occupy_kernel<NV>), dim3(n_cu), dim3(512), lds, s_occ, dummy, ticks * 21, 1.0); CK(hipEventRecord(b, s_occ)); };
		auto lst = [&]() { CK(hipEventRecord(c, s_list)); hipLaunchKernelGGL((s3_lists_t3_kernel<8, 8>), dim3((unsigned)((M + 3) / 4), (unsigned)L.nr), dim3(256), 0, s_list, rows, bpv, N, (int)M, ntile, L, P, Q, part); CK(hipEventRecord(d, s_list)); };
		if (order == 0) { occ(); lst(); } else { lst(); occ(); }
		CK(hipDeviceSynchronize()); CK(hipGetLastError());
		float mo, ml; CK(hipEventElapsedTime(&mo, a, b)); CK(hipEventElapsedTime(&ml, c, d));
		printf("beside %-34s (%s first): stand-in %6.3f ms, list pass %6.3f ms\n", what, order ? "list" : "stand-in", mo, ml);
	}
}

// the fused list + T3 pass of row-major calls (kern_lists.h): timing, counts and sums against a CPU walk
template <int PP>
static int lists_t3_bench(const uint8_t *rows, size_t bpv, int N, size_t M, int ntile, int reps, bool verify, int P)
{
	S3Lists L{};
	L.ld = M; L.nr = s3_nranges(ntile);
	s3_lists_setup(L, ntile, (M * (size_t)L.nr + 3) / 4);
	CK(hipMalloc((void **)&L.lcnt, S3_NR * M * 4));
	long long *Q, *part;
	const size_t nq = (size_t)ntile * 256 * P;
	CK(hipMalloc((void **)&Q, nq * 8));
	CK(hipMalloc((void **)&part, (size_t)L.nr * M * P * 2 * 8));
	std::vector<long long> hq(nq);
	{ uint64_t x = 99; for (auto &v : hq) v = (long long)(sm64(x) >> 8) - (1ll << 55); }
	CK(hipMemcpy(Q, hq.data(), nq * 8, hipMemcpyHostToDevice));
	hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
	const unsigned grid = (unsigned)(((M + 3) / 4) * (size_t)L.nr);
	int bad = 0;
	for (int mode = 0; mode < 1; mode++) {
		float best = 1e30f;
		CK(hipMemset(part, 0xEE, (size_t)L.nr * M * P * 2 * 8));
		for (int r = 0; r < reps + 1; r++) {
			CK(hipEventRecord(a, 0));
			hipLaunchKernelGGL((s3_lists_t3_kernel<8, PP>), dim3((unsigned)((M + 3) / 4), (unsigned)L.nr), dim3(256), 0, 0, rows, bpv, N, (int)M, ntile, L, P, Q, part);
			CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b)); CK(hipGetLastError());
			float ms; CK(hipEventElapsedTime(&ms, a, b));
			if (r > 0 || reps == 0) best = std::min(best, ms);
		}
		printf("list + T3 pass N=%d M=%zu P=%d: %7.3f ms  %6.0f GB/s read\n", N, M, P, best, (double)M * ntile * 64 / best / 1e6);
		if (!verify) continue;
		std::vector<uint8_t> hr(M * bpv);
		CK(hipMemcpy(hr.data(), rows, M * bpv, hipMemcpyDeviceToHost));
		std::vector<int> cn((size_t)L.nr * M);
		std::vector<long long> hp((size_t)L.nr * M * P * 2);
		CK(hipMemcpy(cn.data(), L.lcnt, cn.size() * 4, hipMemcpyDeviceToHost));
		CK(hipMemcpy(hp.data(), part, hp.size() * 8, hipMemcpyDeviceToHost));
		size_t over = 0;
		for (size_t v = 0; v < M && bad < 5; v++)
			for (int g = 0; g < L.nr; g++) {
				const int s0 = s3_range_t0(g, ntile, L.nr) * 256, s1 = std::min(N, s3_range_t0(g + 1, ntile, L.nr) * 256);
				std::vector<long long> hi(P, 0), lo(P, 0);
				int want = 0;
				for (int s = s0; s < s1; s++)
					if (((hr[v * bpv + s / 4] >> (2 * (s % 4))) & 3) == 3) {
						want++;
						for (int c = 0; c < P; c++) { const long long q = hq[(size_t)s * P + c]; hi[c] += q >> 32; lo[c] += q & 0xFFFFFFFFll; }
					}
				const size_t e = (size_t)g * M + v;
				if (want > S3_LT_CAP) { over++; if (cn[e] != -1) { fprintf(stderr, "list+T3: variant %zu range %d: %d entries, count %d (want -1)\n", v, g, want, cn[e]); bad++; } continue; }
				bool ok = cn[e] == want;
				for (int c = 0; ok && c < P; c++) ok = hp[(e * P + c) * 2] == hi[c] && hp[(e * P + c) * 2 + 1] == lo[c];
				if (!ok) { fprintf(stderr, "list+T3: variant %zu range %d: count %d want %d, or sums differ\n", v, g, cn[e], want); bad++; }
			}
		printf("list + T3 check N=%d M=%zu P=%d (%zu segments beyond the cap): %s\n", N, M, P, over, bad ? "FAILED" : "ok");
	}
	CK(hipFree(L.lcnt)); CK(hipFree(Q)); CK(hipFree(part));
	return bad;
}

int main(int argc, char **argv)
{
	hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, 0));
	const int n_cu = pr.multiProcessorCount;
	if (argc > 1 && !strcmp(argv[1], "check")) {
		int bad = 0;
#ifndef S3_BENCH_SMALL
		// grids far smaller than the chip so that rounds, leftovers and pieces all occur
		bad += check<4, 4, 8, 3, 1, 2, 2>("k3 naf4 d2", 5000, 700, 1, 8);
		bad += check<4, 4, 8, 3, 1, 3, 1>("k3 naf4 big grid", 3000, 300, 1, n_cu);
		bad += check<4, 3, 8, 3, 1, 4, 1>("k3 naf3 d4/1", 4100, 1000, 1, 16);
		bad += check<4, 6, 4, 3, 1, 4, 1>("k3 naf6 4+3+1", 4100, 1000, 1, 16);
		bad += check<4, 4, 4, 1, 1, 1, 1>("k3 naf4 d1", 1000, 130, 1, 8);
		bad += check<4, 4, 8, 2, 2, 2, 2>("k3 naf4 tiny N", 100, 50, 1, 8);
		bad += check<4, 2, 12, 3, 1, 3, 2>("k3 naf2 12 waves", 9000, 2100, 1, 8);
		bad += check<11, 4, 4, 3, 1, 3, 1>("k13 naf4 4+3+1", 2500, 800, 1, 8);
		bad += check<6, 3, 8, 3, 1, 3, 1>("k5 naf3", 2100, 900, 1, 8);
		bad += check<2, 4, 8, 3, 1, 3, 2>("quant naf4", 2100, 900, 1, 8);
		bad += check<13, 3, 4, 2, 2, 3, 1>("k16 naf3 4+2+2", 1500, 500, 1, 8);
		bad += check<8, 4, 4, 3, 1, 3, 1>("k8 naf4 4+3+1", 1500, 500, 1, 8);
		bad += check<12, 3, 4, 2, 2, 3, 1>("k13 nbf12 naf3 4+2+2", 1500, 500, 1, 8);
		bad += check<12, 6, 4, 2, 2, 3, 1, 2>("k13 nbf12 naf6 2x2", 1500, 700, 1, 8);
		bad += check<11, 6, 4, 2, 2, 3, 1, 2>("k13 nbf11 naf6 2x2 (6 + 5)", 2100, 700, 1, 8);
		bad += check<12, 5, 4, 2, 2, 3, 1, 2>("k13 nbf12 naf5 2x2", 1500, 700, 1, 8);
		bad += check<8, 6, 4, 2, 2, 3, 1, 2>("k8 naf6 2x2", 1500, 700, 1, 8);
		bad += check<6, 6, 8, 3, 1, 3, 1, 2>("k5 naf6 4x2", 2100, 900, 1, 8);
		bad += check<16, 4, 4, 2, 2, 3, 1, 2>("k16 nbf16 naf4 2x2", 1100, 300, 1, 8);
		bad += check<12, 6, 4, 2, 2, 3, 1, 2, 4>("k13 nbf12 naf6 2x2 4 buffers", 1500, 700, 1, 8);
		bad += check<12, 3, 8, 2, 2, 3, 1, 2>("k13 nbf12 naf3 8 consumers 4x2", 1500, 900, 1, 8);
		bad += check<13, 3, 8, 2, 2, 3, 1, 2>("k16 nbf13 naf3 8 consumers 4x2", 1100, 900, 1, 8);
		bad += check<8, 4, 8, 2, 2, 2, 1, 2>("k8 nbf8 naf4 8 consumers 4x2", 1500, 900, 1, 8);
		bad += check<12, 3, 4, 2, 2, 3, 1, 1, 4>("k13 nbf12 naf3 4 buffers", 1500, 500, 1, 8);
		bad += check<11, 4, 4, 2, 2, 3, 1, 1, 3>("k13 nbf11 naf4 3 buffers", 1500, 500, 1, 8);
		bad += check<13, 3, 4, 2, 2, 3, 1, 1, 5>("k16 nbf13 naf3 5 buffers", 1500, 500, 1, 8);
		bad += check<6, 3, 8, 3, 1, 3, 1, 1, 3>("k5 nbf6 naf3 3 buffers", 2100, 900, 1, 8);
#endif
		// row-major rows (the caller's layout, no tiles): M not a multiple of 16, strides beyond the row
		bad += check<4, 4, 8, 3, 1, 1, 1, 1, 2, 1>("k3 naf4 d1/1", 5000, 700, 1, 8);
		bad += check<4, 3, 8, 3, 1, 1, 2, 1, 2, 1>("k3 naf3 d1/2, M = 693, stride + 64", 5000, 693, 1, 8, 64);
		bad += check<4, 4, 8, 3, 1, 1, 1, 1, 2, 1>("k3 naf4 big grid", 3000, 301, 1, n_cu, 16);
		bad += check<4, 2, 8, 2, 2, 2, 2, 1, 2, 1>("k3 naf2 d2/2 tiny N", 100, 50, 1, 8);
		bad += check<4, 4, 8, 2, 2, 1, 1, 1, 2, 1>("k3 naf4 one variant", 700, 1, 1, 8, 128);
		bad += check<11, 4, 4, 3, 1, 1, 1, 1, 2, 1>("k13 naf4 4+3+1", 2500, 803, 1, 8);
		bad += check<2, 4, 8, 3, 1, 1, 2, 1, 2, 1>("quant naf4", 2100, 900, 1, 8);
		bad += check<13, 3, 4, 2, 2, 1, 1, 1, 2, 1>("k16 naf3 4+2+2", 1500, 499, 1, 8, 192);
		bad += check<4, 4, 8, 3, 1, 1, 1, 1, 2, 1>("k3 naf4 d1/1, long rows", 70000, 100, 1, 8, 128);
		bad += check<4, 2, 8, 3, 1, 2, 2, 1, 2, 1>("k3 naf2 d2/2, long rows", 70000, 500, 1, 8);
		bad += check<6, 3, 8, 3, 1, 1, 1, 1, 2, 1>("k5 naf3 d1/1", 9000, 1000, 1, 16);
		// every form of the product's table (kern_score3.h S3_FOR_EACH_NBF)
#define CHKP(NBF, NAF, NC, NLA, NLB, DA, DB) bad += check<NBF, NAF, NC, NLA, NLB, DA, DB, 1, 2, 1>("product form", 2100 + 37 * NBF, 600 + NBF, 1, 8, (NBF & 1) * 64);
		S3_FOR_EACH_NBF(CHKP)
#undef CHKP
#define CHKM(NBF, NAF, NC, NLA, NLB, DA, DB) bad += check<NBF, NAF, NC, NLA, NLB, DA, DB, 1, 2, 1, true>("three-plane form", 1900 + 41 * NBF, 500 + NBF, 1, 8, (NBF & 1) * 64);
		S3_FOR_EACH_NBF_MISS(CHKM)
#undef CHKM
		bad += check<4, 2, 8, 3, 1, 2, 2, 1, 2, 1, true>("three planes, 1 missing code in 65536", 5000, 700, 1, 8, 0, 1);
		bad += check<4, 2, 8, 3, 1, 2, 2, 1, 2, 1, true>("three planes, no missing code", 3000, 300, 1, 8, 64, 0);
		bad += check<4, 2, 8, 3, 1, 2, 2, 1, 2, 1, true>("three planes, 5 % missing", 3000, 300, 1, 8, 0, 3277);
		return bad ? 1 : 0;
	}
	const int N = argc > 1 ? atoi(argv[1]) : 430000;
	const size_t M = argc > 2 ? (size_t)atoll(argv[2]) : 50000;
	const int reps = argc > 3 ? atoi(argv[3]) : 5;
	const int ntile = 2 * ((N + 511) / 512);
	const size_t abytes = s3_block_bytes(M, ntile);
	uint8_t *A, *Fl; int *out;
	CK(hipMalloc((void **)&A, abytes));
	fill_codes<<<4096, 256>>>((uint32_t *)A, abytes / 4, 12345);
	if (getenv("ZERO_A")) CK(hipMemset(A, 0, abytes));
	const int NBFMAX = 16;
	const size_t flb = (size_t)ntile * 16 * 16 * NBFMAX * 16;
	CK(hipMalloc((void **)&Fl, flb));
	{ std::vector<uint8_t> hf(flb); uint64_t x = 5; for (auto &v : hf) v = getenv("ZERO_B") ? 0 : (uint8_t)sm64(x); CK(hipMemcpy(Fl, hf.data(), flb, hipMemcpyHostToDevice)); }
	const size_t oints = (size_t)1 << 30;       // 4 GiB of slabs: enough for every variant below
	CK(hipMalloc((void **)&out, oints * 4));
	CK(hipDeviceSynchronize());
	printf("N=%d M=%zu ntile=%d rows %.3f GB, %d CUs\n", N, M, ntile, (double)M * ntile * 64 / 1e9, n_cu);
#define R3(NBF, NAF, NC, NLA, NLB, DA, DB, ABL, NCB, NBUF, name) do { \
\
	run<NBF, NAF, NC, NLA, NLB, DA, DB, ABL, NCB, NBUF>(name, A, Fl, ntile, M, 1, n_cu, out, oints, reps); } while (0)
#define R2(NBF, NAF, NC, NLA, NLB, DA, DB, ABL, NCB, name) R3(NBF, NAF, NC, NLA, NLB, DA, DB, ABL, NCB, 2, name)
#define R(NBF, NAF, NC, NLA, NLB, DA, DB, ABL, name) R2(NBF, NAF, NC, NLA, NLB, DA, DB, ABL, 1, name)
	if (getenv("RM")) {
		// row-major rows against tiles, the product forms of K = 3, quantitative, K = 13; N = 50 000 by the command line
		const size_t bpv = (size_t)ntile * 64;
		uint8_t *Ar;
		CK(hipMalloc((void **)&Ar, M * bpv));
		fill_codes<<<4096, 256>>>((uint32_t *)Ar, M * bpv / 4, 12345, getenv("MISS16") ? (uint32_t)atoi(getenv("MISS16")) : 66u);
		CK(hipDeviceSynchronize());
		{
			// small shapes against a CPU walk (N not a multiple of 64, a long-range shape), then the timing
			uint8_t *Sm; const int n2 = 5003, nt2 = 2 * ((n2 + 511) / 512); const size_t m2 = 333, bp2 = (size_t)nt2 * 64 + 64;
			CK(hipMalloc((void **)&Sm, m2 * bp2));
			fill_codes<<<256, 256>>>((uint32_t *)Sm, m2 * bp2 / 4, 777);
			CK(hipDeviceSynchronize());
			int badl = lists_bench(Sm, bp2, n2, m2, nt2, 0, true);
			CK(hipFree(Sm));
			const int n3 = 600000, nt3 = 2 * ((n3 + 511) / 512); const size_t m3 = 40, bp3 = (size_t)nt3 * 64;
			CK(hipMalloc((void **)&Sm, m3 * bp3));
			fill_codes<<<256, 256>>>((uint32_t *)Sm, m3 * bp3 / 4, 778);
			CK(hipDeviceSynchronize());
			badl += lists_bench(Sm, bp3, n3, m3, nt3, 0, true);
			CK(hipFree(Sm));
			if (badl) return 1;
			lists_bench(Ar, bpv, N, M, ntile, reps, false);
			if (getenv("LT3")) {
				// small shapes against the CPU (ragged N, a long-range shape, dense missing codes), then the timing
				const int mrates[3] = {66, 655, 6000};
				for (int q = 0; q < 3; q++) {
					const int n4 = 70001 + 4099 * q, nt4 = 2 * ((n4 + 511) / 512); const size_t m4 = 131, bp4 = (size_t)nt4 * 64 + 128;
					CK(hipMalloc((void **)&Sm, m4 * bp4));
					fill_codes<<<256, 256>>>((uint32_t *)Sm, m4 * bp4 / 4, 900 + q, mrates[q]);
					CK(hipDeviceSynchronize());
					badl += lists_t3_bench<8>(Sm, bp4, n4, m4, nt4, 0, true, 8);
					badl += lists_t3_bench<16>(Sm, bp4, n4, m4, nt4, 0, true, 12);
					badl += lists_t3_bench<32>(Sm, bp4, n4, m4, nt4, 0, true, 28);
					badl += lists_t3_bench<64>(Sm, bp4, n4, m4, nt4, 0, true, 34);
					CK(hipFree(Sm));
				}
				{
					const int n5 = 600000, nt5 = 2 * ((n5 + 511) / 512); const size_t m5 = 24, bp5 = (size_t)nt5 * 64;
					CK(hipMalloc((void **)&Sm, m5 * bp5));
					fill_codes<<<256, 256>>>((uint32_t *)Sm, m5 * bp5 / 4, 779, 20);
					CK(hipDeviceSynchronize());
					badl += lists_t3_bench<8>(Sm, bp5, n5, m5, nt5, 0, true, 8);
					CK(hipFree(Sm));
				}
				if (badl) return 1;
				lists_t3_bench<8>(Ar, bpv, N, M, ntile, reps, false, 8);
				lists_t3_bench<32>(Ar, bpv, N, M, ntile, reps, false, 28);
				if (getenv("BESIDE")) {
					S3Lists L{};
					L.ld = M; L.nr = s3_nranges(ntile);
					s3_lists_setup(L, ntile, (M * (size_t)L.nr + 3) / 4);
					CK(hipMalloc((void **)&L.lcnt, S3_NR * M * 4));
					long long *Q, *part; double *dummy;
					CK(hipMalloc((void **)&Q, (size_t)ntile * 256 * 8 * 8)); CK(hipMemset(Q, 1, (size_t)ntile * 256 * 8 * 8));
					CK(hipMalloc((void **)&part, (size_t)L.nr * M * 8 * 2 * 8)); CK(hipMalloc((void **)&dummy, 4096 * 8));
					hipStream_t so, sl; int lo, hi; CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
					const int mode = atoi(getenv("BESIDE"));      // 1: stand-in low / list high priority, 2: both normal, 3: stand-in high / list low
					CK(hipStreamCreateWithPriority(&so, hipStreamNonBlocking, mode == 1 ? lo : mode == 3 ? hi : 0)); CK(hipStreamCreateWithPriority(&sl, hipStreamNonBlocking, mode == 1 ? hi : mode == 3 ? lo : 0));
					printf("priorities: range %d (low) .. %d (high), mode %d\n", lo, hi, mode);
					beside<46>("186 regs, 154 KiB LDS", 154 * 1024, n_cu, so, sl, Ar, bpv, N, M, ntile, L, 8, Q, part, dummy);
					beside<46>("186 regs, 128 KiB LDS", 128 * 1024, n_cu, so, sl, Ar, bpv, N, M, ntile, L, 8, Q, part, dummy);
					beside<46>("186 regs, 8 KiB LDS", 8 * 1024, n_cu, so, sl, Ar, bpv, N, M, ntile, L, 8, Q, part, dummy);
					beside<28>("64 regs, 154 KiB LDS", 154 * 1024, n_cu, so, sl, Ar, bpv, N, M, ntile, L, 8, Q, part, dummy);
					beside<28>("64 regs, 8 KiB LDS", 8 * 1024, n_cu, so, sl, Ar, bpv, N, M, ntile, L, 8, Q, part, dummy);
				}
				return 0;
			}
		}
#define R0(NBF, NAF, NC, NLA, NLB, DA, DB, ABL, name) run<NBF, NAF, NC, NLA, NLB, DA, DB, ABL, 1, 2, 0>(name, A, Fl, ntile, M, 1, n_cu, out, oints, reps)
#define R1(NBF, NAF, NC, NLA, NLB, DA, DB, ABL, name) run<NBF, NAF, NC, NLA, NLB, DA, DB, ABL, 1, 2, 1>(name, Ar, Fl, ntile, M, 1, n_cu, out, oints, reps, nullptr, bpv)
		R0(4, 4, 8, 3, 1, 2, 2, 0, "k3 naf4 8+3+1 d2/2");
		R0(4, 4, 8, 3, 1, 2, 1, 0, "k3 naf4 8+3+1 d2/1");
		R0(4, 3, 8, 3, 1, 3, 2, 0, "k3 naf3 8+3+1 d3/2");
		R0(4, 4, 8, 3, 1, 2, 2, 1, "k3 naf4 memory only");
		R1(4, 4, 8, 3, 1, 1, 1, 0, "k3 naf4 8+3+1 pairs d1/1 (160 KiB)");
		R1(4, 4, 8, 2, 2, 1, 1, 0, "k3 naf4 8+2+2 pairs d1/1 (160 KiB)");
		R1(4, 3, 8, 3, 1, 1, 2, 0, "k3 naf3 8+3+1 pairs d1/2");
		R1(4, 3, 8, 2, 2, 1, 2, 0, "k3 naf3 8+2+2 pairs d1/2");
		R1(4, 2, 8, 3, 1, 2, 2, 0, "k3 naf2 8+3+1 pairs d2/2");
		R1(4, 2, 12, 3, 1, 1, 2, 0, "k3 naf2 12+3+1 pairs d1/2");
		R1(4, 4, 8, 3, 1, 1, 1, 1, "k3 naf4 pairs d1/1 memory only");
		R1(4, 3, 8, 3, 1, 1, 2, 1, "k3 naf3 pairs d1/2 memory only");
#define R3M(NBF, NAF, NC, NLA, NLB, DA, DB, ABL, name) run<NBF, NAF, NC, NLA, NLB, DA, DB, ABL, 1, 2, 1, true>(name, Ar, Fl, ntile, M, 1, n_cu, out, oints, reps, nullptr, bpv)
		R3M(4, 2, 8, 3, 1, 2, 2, 0, "k3 three planes naf2 8+3+1 d2/2");
		R3M(4, 2, 8, 3, 1, 2, 1, 0, "k3 three planes naf2 8+3+1 d2/1");
		R3M(4, 2, 8, 2, 2, 2, 2, 0, "k3 three planes naf2 8+2+2 d2/2");
		R3M(4, 4, 4, 3, 1, 2, 2, 0, "k3 three planes naf4 4+3+1 d2/2");
		R3M(4, 2, 8, 3, 1, 2, 2, 1, "k3 three planes naf2 memory only");
#define R3M3(NBF, NAF, NC, NLA, NLB, DA, DB, ABL, name) run<NBF, NAF, NC, NLA, NLB, DA, DB, ABL, 1, 3, 1, true>(name, Ar, Fl, ntile, M, 1, n_cu, out, oints, reps, nullptr, bpv)
		if (getenv("NBUF3")) {
			if (check<4, 2, 8, 3, 1, 2, 2, 1, 3, 1, true>("three planes, B reads two chunks ahead", 2300, 777, 1, 8, 64)) return 1;
			for (int rep = 0; rep < 3; rep++) {
				R3M(4, 2, 8, 3, 1, 2, 2, 0, "k3 three planes naf2 8+3+1 d2/2");
				R3M3(4, 2, 8, 3, 1, 2, 2, 0, "k3 three planes naf2 8+3+1 d2/2 NBUF 3");
			}
			return 0;
		}
#ifdef S3_BENCH_SWEEPS   /* the shapes of the three-plane K = 3 kernel that were measured and not adopted (tools/README.md): -DS3_BENCH_SWEEPS, SWEEP3 / SWEEP4 / SWEEP5=1 */
		if (getenv("SWEEP3")) {
			R3M(4, 3, 8, 3, 1, 1, 1, 0, "k3 three planes naf3 8+3+1 d1/1");
			R3M(4, 3, 8, 3, 1, 1, 2, 0, "k3 three planes naf3 8+3+1 d1/2");
			R3M(4, 3, 8, 2, 2, 1, 1, 0, "k3 three planes naf3 8+2+2 d1/1");
			R3M(4, 4, 4, 2, 2, 2, 2, 0, "k3 three planes naf4 4+2+2 d2/2");
			R3M(4, 4, 4, 3, 1, 3, 2, 0, "k3 three planes naf4 4+3+1 d3/2");
			R3M(4, 2, 8, 3, 1, 3, 1, 0, "k3 three planes naf2 8+3+1 d3/1");
			R3M(4, 2, 12, 3, 1, 1, 1, 0, "k3 three planes naf2 12+3+1 d1/1");
			R3M(4, 3, 4, 3, 1, 2, 2, 0, "k3 three planes naf3 4+3+1 d2/2");
			R3M(4, 4, 8, 3, 1, 1, 1, 0, "k3 three planes naf4 8+3+1 d1/1");
		}
		if (getenv("SWEEP5")) {
			// NAF = 3 at two waves per SIMD (256 registers): 8 waves per workgroup
			int badc = 0;
			badc += check<4, 3, 6, 1, 1, 2, 2, 1, 2, 1, true>("three planes naf3 6+1+1", 2300, 777, 1, 8, 64);
			badc += check<4, 3, 5, 2, 1, 2, 2, 1, 2, 1, true>("three planes naf3 5+2+1", 2300, 777, 1, 8, 64);
			if (badc) return 1;
			for (int rep = 0; rep < 2; rep++) {
				R3M(4, 2, 8, 3, 1, 2, 2, 0, "k3 three planes naf2 8+3+1 d2/2");
				R3M(4, 3, 6, 1, 1, 2, 2, 0, "k3 three planes naf3 6+1+1 d2/2");
				R3M(4, 3, 6, 1, 1, 1, 2, 0, "k3 three planes naf3 6+1+1 d1/2");
				R3M(4, 3, 5, 2, 1, 2, 2, 0, "k3 three planes naf3 5+2+1 d2/2");
				R3M(4, 3, 5, 2, 1, 3, 2, 0, "k3 three planes naf3 5+2+1 d3/2");
				R3M(4, 3, 6, 1, 1, 2, 2, 1, "k3 three planes naf3 6+1+1 memory only");
			}
			return 0;
		}
		if (getenv("SWEEP4")) {
			int badc = 0;
			badc += check<4, 4, 5, 2, 1, 2, 1, 1, 2, 1, true>("three planes naf4 5+2+1", 2300, 777, 1, 8, 64);
			badc += check<4, 4, 6, 1, 1, 1, 2, 1, 2, 1, true>("three planes naf4 6+1+1", 2300, 777, 1, 8, 64);
			badc += check<4, 2, 8, 2, 2, 2, 2, 1, 2, 1, true>("three planes naf2 8+2+2", 2300, 777, 1, 8, 64);
			if (badc) return 1;
			R3M(4, 2, 8, 2, 2, 2, 2, 0, "k3 three planes naf2 8+2+2 d2/2");
			R3M(4, 4, 5, 2, 1, 2, 1, 0, "k3 three planes naf4 5+2+1 d2/1");
			R3M(4, 4, 5, 2, 1, 1, 2, 0, "k3 three planes naf4 5+2+1 d1/2");
			R3M(4, 4, 6, 1, 1, 1, 2, 0, "k3 three planes naf4 6+1+1 d1/2");
			R3M(4, 4, 6, 1, 1, 1, 1, 0, "k3 three planes naf4 6+1+1 d1/1");
			R3M(4, 4, 4, 2, 2, 2, 1, 0, "k3 three planes naf4 4+2+2 d2/1");
			R3M(4, 4, 4, 3, 1, 2, 1, 0, "k3 three planes naf4 4+3+1 d2/1");
			R3M(4, 2, 8, 2, 2, 2, 1, 0, "k3 three planes naf2 8+2+2 d2/1");
			R3M(4, 2, 8, 2, 2, 3, 1, 0, "k3 three planes naf2 8+2+2 d3/1");
			return 0;
		}
#endif
		R3M(2, 4, 8, 3, 1, 1, 2, 0, "quant three planes naf4");
		R3M(6, 3, 4, 2, 2, 2, 1, 0, "k5 three planes naf3 4+2+2");
		R3M(12, 1, 4, 2, 2, 2, 1, 0, "k13 three planes naf1 4+2+2");
		R0(2, 4, 8, 3, 1, 3, 2, 0, "quant naf4 8+3+1 d3/2");
		R1(2, 4, 8, 3, 1, 1, 2, 0, "quant naf4 8+3+1 pairs d1/2");
		R1(2, 4, 8, 3, 1, 1, 3, 0, "quant naf4 8+3+1 pairs d1/3");
		R0(12, 3, 4, 2, 2, 3, 1, 0, "k13 nbf12 naf3 4+2+2 d3/1");
		R1(12, 3, 4, 2, 2, 1, 1, 0, "k13 nbf12 naf3 4+2+2 pairs d1/1");
		R0(6, 3, 8, 3, 1, 3, 1, 0, "k5 naf3 8+3+1 d3/1");
		R1(6, 3, 8, 3, 1, 1, 1, 0, "k5 naf3 8+3+1 pairs d1/1");
		return 0;
	}
#ifndef S3_BENCH_SMALL
	if (getenv("ONLY13")) {
		R2(12, 6, 4, 2, 2, 3, 1, 0, 2, "k13 nbf12 naf6 2x2");
		return 0;
	}
	if (getenv("W8")) {
		// eight consumer waves (two per SIMD) as variant groups x column groups
		R2(12, 6, 4, 2, 2, 3, 1, 16, 2, "k13 nbf12 naf6 4 consumers 2x2");
		R2(12, 3, 8, 3, 1, 3, 1, 16, 2, "k13 nbf12 naf3 8 consumers 4x2 +3+1");
		R2(12, 3, 8, 2, 2, 3, 1, 16, 2, "k13 nbf12 naf3 8 consumers 4x2 +2+2");
		R2(12, 3, 8, 1, 3, 3, 1, 16, 2, "k13 nbf12 naf3 8 consumers 4x2 +1+3");
		R3(12, 3, 8, 2, 2, 3, 1, 16, 2, 3, "k13 nbf12 naf3 8 consumers 4x2 +2+2, 3 buffers");
		R2(12, 4, 8, 2, 2, 2, 1, 16, 2, "k13 nbf12 naf4 8 consumers 4x2 +2+2 d2/1");
		R2(12, 6, 8, 2, 2, 3, 1, 16, 4, "k13 nbf12 naf6 8 consumers 2x4 +2+2");
		R2(12, 2, 8, 2, 2, 3, 1, 16, 1, "k13 nbf12 naf2 8 consumers 8x1 +2+2");
		R2(11, 3, 8, 2, 2, 3, 1, 16, 2, "k13 nbf11 naf3 8 consumers 4x2");
		R2(8, 4, 8, 2, 2, 2, 1, 16, 2, "k8 nbf8 naf4 8 consumers 4x2 d2/1");
		R2(8, 3, 8, 2, 2, 3, 1, 16, 2, "k8 nbf8 naf3 8 consumers 4x2");
		R(8, 4, 4, 2, 2, 3, 1, 16, "k8 nbf8 naf4 4+2+2 (one group)");
		R2(16, 3, 8, 2, 2, 2, 1, 16, 2, "nbf16 naf3 8 consumers 4x2 d2/1");
		R2(16, 2, 8, 2, 2, 3, 1, 16, 2, "nbf16 naf2 8 consumers 4x2");
		R2(13, 3, 8, 2, 2, 3, 1, 16, 2, "k16 nbf13 naf3 8 consumers 4x2");
		return 0;
	}
	if (getenv("ABL13")) {
		R2(12, 6, 4, 2, 2, 3, 1, 16, 2, "full");
		R2(12, 6, 4, 2, 2, 3, 1, 2 | 16, 2, "no row DMA");
		R2(12, 6, 4, 2, 2, 3, 1, 4 | 16, 2, "no B DMA");
		R2(12, 6, 4, 2, 2, 3, 1, 1 | 2 | 4 | 16, 2, "LDS reads + barrier only");
		R2(12, 6, 4, 2, 2, 3, 1, 1 | 4 | 16, 2, "row DMA + LDS reads, no arithmetic");
		R2(12, 6, 4, 2, 2, 3, 1, 1 | 2 | 16, 2, "B DMA + LDS reads, no arithmetic");
		R2(12, 6, 4, 2, 2, 3, 1, 1 | 2 | 8 | 16, 2, "B DMA only (no LDS reads of B, no arithmetic)");
		R2(12, 6, 4, 2, 2, 3, 1, 1 | 2 | 4 | 8 | 16, 2, "barrier + A reads only");
		R2(12, 6, 4, 2, 2, 3, 1, 2 | 4 | 8 | 16, 2, "arithmetic only (no DMA, no B reads)");
		R2(12, 6, 4, 1, 3, 3, 1, 1 | 2 | 8 | 16, 2, "B DMA only, 3 B loaders");
		R2(12, 6, 4, 1, 1, 3, 1, 1 | 2 | 8 | 16, 2, "B DMA only, 1 B loader");
		return 0;
	}
	if (getenv("ONLY3")) {
		R(4, 4, 8, 3, 1, 2, 2, 0, "k3 naf4 8+3+1 d2/2");
		return 0;
	}
	if (getenv("WIDE")) {
		// many covariates: column groups (NCB > 1) and deeper B prefetch (NBUF) against the round-3 forms
		R(12, 3, 4, 2, 2, 3, 1, 16, "k13 nbf12 naf3 4+2+2 (one group)");
		R3(12, 3, 4, 2, 2, 3, 1, 16, 1, 3, "k13 nbf12 naf3, 3 buffers");
		R3(12, 3, 4, 2, 2, 3, 1, 16, 1, 4, "k13 nbf12 naf3, 4 buffers");
		R3(12, 3, 4, 2, 2, 3, 1, 16, 1, 6, "k13 nbf12 naf3, 6 buffers");
		R2(12, 6, 4, 2, 2, 3, 1, 16, 2, "k13 nbf12 naf6 2x2");
		R3(12, 6, 4, 2, 2, 3, 1, 16, 2, 3, "k13 nbf12 naf6 2x2, 3 buffers");
		R3(12, 6, 4, 2, 2, 3, 1, 16, 2, 4, "k13 nbf12 naf6 2x2, 4 buffers");
		R3(12, 6, 4, 2, 2, 3, 1, 1 | 16, 2, 4, "k13 nbf12 naf6 2x2, 4 buffers, memory system only");
		R3(12, 5, 4, 2, 2, 3, 1, 16, 2, 4, "k13 nbf12 naf5 2x2, 4 buffers");
		R(11, 4, 4, 2, 2, 3, 1, 16, "k13 nbf11 naf4 4+2+2 (one group)");
		R3(11, 4, 4, 2, 2, 3, 1, 16, 1, 3, "k13 nbf11 naf4, 3 buffers");
		R(8, 4, 4, 2, 2, 3, 1, 16, "k8 nbf8 naf4 4+2+2 (one group)");
		R3(8, 4, 4, 2, 2, 3, 1, 16, 1, 3, "k8 nbf8 naf4, 3 buffers");
		R3(8, 4, 4, 2, 2, 3, 1, 16, 1, 4, "k8 nbf8 naf4, 4 buffers");
		R(6, 3, 8, 3, 1, 3, 1, 16, "k5 nbf6 naf3 8+3+1 (one group)");
		R3(6, 3, 8, 3, 1, 3, 1, 16, 1, 3, "k5 nbf6 naf3, 3 buffers");
		R(13, 3, 4, 2, 2, 3, 1, 16, "k16 nbf13 naf3 (one group)");
		R3(13, 3, 4, 2, 2, 3, 1, 16, 1, 4, "k16 nbf13 naf3, 4 buffers");
		R(16, 2, 4, 2, 2, 3, 1, 16, "nbf16 naf2 (one group)");
		R3(16, 2, 4, 2, 2, 3, 1, 16, 1, 4, "nbf16 naf2, 4 buffers");
		R3(16, 4, 4, 2, 2, 3, 1, 16, 2, 4, "nbf16 naf4 2x2, 4 buffers");
		return 0;
	}
	// K = 3 (3 value fragments + bit-1)
	if (getenv("SHORT")) {
		R(4, 4, 8, 3, 1, 2, 2, 16, "k3 naf4 8+3+1 d2/2");
		R(4, 3, 8, 2, 2, 4, 1, 16, "k3 naf3 8+2+2 d4/1");
		R(4, 6, 4, 3, 1, 4, 1, 16, "k3 naf6 4+3+1 d4/1");
		R(4, 6, 4, 2, 2, 4, 1, 16, "k3 naf6 4+2+2 d4/1");
		R(4, 4, 8, 3, 1, 3, 1, 6 | 16, "k3 naf4 no DMA");
		R(4, 6, 4, 3, 1, 4, 1, 6 | 16, "k3 naf6 no DMA");
		R(4, 4, 8, 3, 1, 3, 1, 1 | 16, "k3 naf4 memory system only");
		R(2, 4, 8, 3, 1, 3, 2, 16, "quant naf4 8+3+1 d3/2");
		R(11, 4, 4, 2, 2, 3, 1, 16, "k13 naf4 4+2+2 d3/1");
		return 0;
	}
	R(4, 4, 8, 3, 1, 2, 2, 16, "k3 naf4 8+3+1 d2/2");
	R(4, 4, 8, 3, 1, 3, 1, 16, "k3 naf4 8+3+1 d3/1");
	R(4, 4, 8, 4, 0 + 1, 3, 1, 0, "k3 naf4 8+4+1 d3/1");
	R(4, 3, 8, 3, 1, 4, 1, 16, "k3 naf3 8+3+1 d4/1");
	R(4, 3, 8, 3, 1, 3, 2, 0, "k3 naf3 8+3+1 d3/2");
	R(4, 3, 8, 3, 1, 4, 2, 0, "k3 naf3 8+3+1 d4/2");
	R(4, 3, 8, 6, 2, 4, 1, 0, "k3 naf3 8+6+2 d4/1");
	R(4, 3, 8, 2, 2, 4, 1, 0, "k3 naf3 8+2+2 d4/1");
	R(4, 2, 12, 3, 1, 4, 1, 16, "k3 naf2 12+3+1 d4/1");
	R(4, 2, 12, 3, 1, 4, 2, 0, "k3 naf2 12+3+1 d4/2");
	R(4, 2, 8, 3, 1, 6, 2, 0, "k3 naf2 8+3+1 d6/2");
	R(4, 4, 8, 3, 1, 3, 1, 1 | 16, "k3 naf4 memory system only");
	R(4, 4, 8, 3, 1, 3, 1, 6 | 16, "k3 naf4 no DMA");
	R(4, 3, 8, 3, 1, 4, 1, 1 | 16, "k3 naf3 memory system only");
	R(4, 3, 8, 3, 1, 4, 1, 6 | 16, "k3 naf3 no DMA");
	R(4, 3, 8, 3, 1, 4, 1, 2 | 16, "k3 naf3 no row DMA");
	// quantitative (value fragment + bit-1: 2 fragments)
	R(2, 4, 8, 3, 1, 3, 2, 0, "quant naf4 8+3+1 d3/2");
	R(2, 4, 8, 3, 1, 4, 1, 0, "quant naf4 8+3+1 d4/1");
	// K = 5, 8, 13, 16
	R(6, 3, 8, 3, 1, 3, 1, 0, "k5 naf3 8+3+1 d3/1");
	R(6, 3, 8, 2, 2, 3, 1, 0, "k5 naf3 8+2+2 d3/1");
	R(8, 4, 4, 3, 1, 3, 1, 0, "k8 naf4 4+3+1 d3/1");
	R(8, 4, 4, 2, 2, 3, 1, 0, "k8 naf4 4+2+2 d3/1");
	R(11, 4, 4, 2, 2, 3, 1, 16, "k13 naf4 4+2+2 d3/1");
	R(11, 4, 4, 3, 1, 3, 1, 0, "k13 naf4 4+3+1 d3/1");
	R(11, 4, 4, 2, 2, 3, 1, 6 | 16, "k13 naf4 no DMA");
	R(13, 3, 4, 2, 2, 3, 1, 0, "k16 naf3 4+2+2 d3/1");
	// N = 50 000 shape is run by passing N on the command line
#endif
	return 0;
}
