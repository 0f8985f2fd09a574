// kern_spa4.h -- SPA stage v4: the saddlepoint sums of a flagged variant as a power series in t.
// Part of libsaigehip.so (single translation unit: saigehip.hip).
#pragma once
#include "kern_spa4_tab.h"

// What the reference evaluates per Newton step and root (SPATest.cpp:42-83) are sums over the
// carriers I of the variant,
//     Korg(t) = sum_I log(1 - mu_i + mu_i e^{g_i t}),   K1 = Korg',   K2 = Korg'',
// i.e. the cumulant generating function of sum_I g_i Bernoulli(mu_i) and its derivatives.  Its
// Taylor coefficients are the cumulants
//     kappa_n = sum_I g_i^n c_n(mu_i),      c_2 = mu (1 - mu),  c_{n+1} = mu (1 - mu) dc_n/dmu,
// and the series converges for |g_i t| < sqrt(logit(mu_i)^2 + pi^2) (the nearest zero of
// 1 - mu + mu e^x), i.e. with ratio <= |g t| / pi per term.  At biobank sizes g_i ~ 1/sqrt(2 N maf)
// is small: for every variant with maf above ~1 % (98 % of all carriers of a scan at N = 430 000)
// max |g_i t| stays under ~0.7 and sixteen cumulants give Korg, K1 and K2 to better than 1e-13 --
// under the rounding noise of the reference's own double sums (tests/diagnostics/cumulant_proto.py).
//
// So the stage is ONE pass over the carriers of the flagged variants (spa4_moments: the cumulant
// sums beside the sums the scalars need anyway), after which every Newton evaluation of
// getroot_K1_fast, both roots, and the Lugannani-Rice tail are polynomial evaluations by one thread
// per variant (spa4_solve).  No carrier list is stored and none is re-read.
//
// A variant leaves this path for the exact exp/log sums (spa5_kernel mode 1 on its carrier list, or the
// dense spa_kernel) when
//   * any point the root search evaluates has gmax |t| > spa_xmax (= a quarter of the smallest
//     convergence radius over the model's mu_i), or
//   * there the last two terms of the K2 series (the slowest of the three) exceed SPA4_TAIL_TOL of its
//     sum, or
//   * the g_pos / g_neg bound test of kern_spa2.h is not decisive (-> spa_kernel).
// Those are the rare variants (few carriers, large g): their lists are short.  Results do not depend on
// which path a variant takes beyond ~1e-12 (tests: test_baseline_c3_exact_path_agrees).
//
// Variants with at most SPA5_NNZ carriers never enter the per-segment pass: a workgroup per variant
// (spa5_kernel) builds the carrier list once and runs the same series on it (mode 0), or the exact sweeps
// where the predicted max |g t| is beyond the series' reach (mode 1).
//
//   spa4_moments  workgroups pull (sample segment, slice of the flagged variants) items from a queue;
//                 the segment's X rows and mu are staged in LDS, then one wave per variant compacts
//                 its carriers through an LDS queue and accumulates, 64 carriers at a time,
//                 sum mu G, sum b, sum max/min(adj, 0), kappa_1, kappa_2, max |adj| and
//                 kappa'_n = sum (adj ts)^n c_n(mu)/n!  (n = 3..NC; ts = a power of two near the
//                 expected root, so the scaled sums decay like the series itself)
//   spa4_solve    per variant: ordered sums over the segments, the scalars of saige_main.cpp:369-381,
//                 the cutoff exit (SPATest.cpp:319-321), both root searches (root_feed / root_step of
//                 kern_spa2.h, fed from the series), tail probabilities, SE, the output row

// waves of spa4_moments' one workgroup per CU: 8 = two per SIMD at 168 registers.  Alone, 12 waves are faster
// at K <= 8 (C3: 0.87 ms against 0.95 for tier A; 16 would need 128 registers and spill), but three waves of
// 168 registers leave a SIMD nothing: with two, the small kernels of the step in flight on the other lane
// (sparse pass, reduction, epilogue, solves) find room beside the pass instead of waiting for it, and the scan
// as a whole is ~2 % faster on two lanes (same box, C3: 2.55 -> 2.50 ms per step; K = 5: 3.42 -> 3.29).  With
// many covariates a wave's state (c[K], a table row per carrier in flight) does not fit 168 registers at
// three waves anyway: K = 13 ran 2.56 ms at 8 waves and 3.24 at 12.
#ifndef SPA4_WAVES_LOWK
#define SPA4_WAVES_LOWK 8
#endif
__host__ __device__ constexpr int spa4_waves(int K) { return K <= 8 ? SPA4_WAVES_LOWK : 8; }
#define SPA4_NCA 12              /* cumulants carried for the variants of tier A (small g t: most carriers) */
#define SPA4_NCB SPA4_NC         /* ... of tier B */
#define SPA4_NSMAX (SPA4_NC + 5) /* partial sums per (variant, segment) of the wider tier */
// The last two terms of the K2 series may be this fraction of its sum.  K2 enters the tail
// probability only through log(v/w)/w: measured over the bench workload the p-value moves by less
// than 1/80 of that fraction (tests/diagnostics/cumulant_proto.py), i.e. stays under ~1e-12.
#define SPA4_TAIL_TOL 1e-10

// T_n = y^n c_n(mu)/n! for n = 3..NC added to acc[n - 3];  y = adj * ts.
// The polynomials in u are summed over explicit powers of u (one multiply-add per coefficient with
// the coefficient as the constant operand), not by Horner's rule, whose running value would need a
// register copy of every coefficient first.  The first step, C1 u + C0, is written as the three-address
// v_fma_f64 (C1 in scalar registers, C0 in vector registers, both loop-invariant): left to itself the compiler
// copies C0 into the result register and adds C1 u with the two-address form -- one v_mov_b64 per polynomial,
// ten of the ~85 vector instructions a carrier costs at twelve cumulants.
__device__ __forceinline__ double spa4_fma_svv(double a_sgpr, double x, double c_vgpr)
{
	double r;
	asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "s"(a_sgpr), "v"(x), "v"(c_vgpr));
	return r;
}

// g = max(g, |a|) in one instruction (fmax on a loop-carried value costs a second one: the compiler cannot see that
// it is canonical and quiets it first)
__device__ __forceinline__ void spa4_max_abs(double &g, double a)
{
	asm("v_max_f64 %0, %0, |%1|" : "+v"(g) : "v"(a));
}

template <int NC>
__device__ __forceinline__ void spa4_cum_terms(double y, double u, double d, double *acc)
{
	constexpr int ND = (NC - 2) / 2 + 1;          // powers of u needed
	double up[ND];                 // u^k
	up[0] = 1.0; up[1] = u;
#pragma unroll
	for (int k = 2; k < ND; k++) up[k] = up[k - 1] * u;
	const double y2 = y * y;
	double pe = y2 * u;            // y^n u      (even n, starting at n = 2)
	double po = pe * y * d;        // y^n u d    (odd n, starting at n = 3)
#pragma unroll
	for (int n = 3; n <= NC; n++) {
		const int deg = (n - 2) / 2;      // degree of the polynomial in u
		double b;
		if (deg == 0) b = SPA4_CUM[n][0];
		else {
			b = spa4_fma_svv(SPA4_CUM[n][1], u, SPA4_CUM[n][0]);
#pragma unroll
			for (int k = 2; k <= deg; k++) b = fma(SPA4_CUM[n][k], up[k], b);
		}
		if (n & 1) {
			acc[n - 3] = fma(po, b, acc[n - 3]);
			po *= y2;
		} else {
			pe *= y2;
			acc[n - 3] = fma(pe, b, acc[n - 3]);
		}
	}
}

// Sum V doubles per lane over the wave with ~V shuffles instead of 6 V: at every step a lane keeps
// half of its values and hands the other half to its partner (xor 32, 16, ..), the last steps add
// the lanes that hold the same value.  Afterwards x[0] of lane l is the total of value idx (returned;
// idx >= V: padding).  The tree is fixed, so the result does not depend on scheduling.
//   V in 17..24: halves 24 -> 12 -> 6 -> 3 (padded to 4) -> 2 -> 1, idx = 12 b5 + 6 b4 + 3 b3 + 2 b2 + b1
//                (b2 = b1 = 1 is padding)
//   V in  9..16: halves 16 -> 8 -> 4 -> 2 -> 1,                     idx = 8 b5 + 4 b4 + 2 b3 + b2
template <int V>
__device__ __forceinline__ int wave_reduce_scatter(double (&x)[V], int lane)
{
	static_assert(V <= 24 && V > 8, "written for 9..24 values");
	constexpr bool WIDE = V > 16;
	constexpr int NST = WIDE ? 5 : 4;
	int idx = 0;
#pragma unroll
	for (int st = 0; st < NST; st++) {
		const int o = 32 >> st;
		const int n = WIDE ? (st == 0 ? 24 : st == 1 ? 12 : st == 2 ? 6 : st == 3 ? 4 : 2) : (16 >> st);   // values held
		const int half = n / 2;
		const bool up = (lane & o) != 0;
#pragma unroll
		for (int i = 0; i < half; i++) {
			// slots past the V values of the first step, and the pad slot of the wide form's fourth, hold zero
			const bool hi_real = st == 0 ? (i + half < V) : !(WIDE && st == 3 && i + half == 3);
			const double xh = hi_real ? x[i + half] : 0.0;
			const double send = up ? x[i] : xh;
			const double keep = up ? xh : x[i];
			x[i] = keep + __shfl_xor(send, o, WAVE);
		}
		idx += up ? (WIDE ? (st == 0 ? 12 : st == 1 ? 6 : st == 2 ? 3 : st == 3 ? 2 : 1) : (8 >> st)) : 0;
	}
	x[0] += __shfl_xor(x[0], 1, WAVE);
	if (!WIDE) x[0] += __shfl_xor(x[0], 2, WAVE);
	return (WIDE && (lane & 6) == 6) ? 24 : idx;
}

#define SPA4_VPER 128            /* flagged variants whose parameters a workgroup holds in LDS at a time (at most) */

// Variants per item: SPA4_VPER when that still gives every workgroup a few (segment, slice) items; with few
// segments (N = 50 000: 13) smaller slices, whole rounds of the workgroup's waves, so that the items
// outnumber the workgroups ~8 to 1 (with 128 the 143 items of a C2 block left 113 of 256 CUs idle).
__device__ __forceinline__ int spa4_slice(int nflag, int nseg, int nwg, int waves)
{
	const int want = (8 * nwg + nseg - 1) / nseg;                      // slices wanted
	int v = (nflag + want - 1) / want;
	v = (v + waves - 1) / waves * waves;
	return min(SPA4_VPER, max(v, waves));
}

// entries of a wave's leftover queue
__host__ __device__ constexpr int spa4_qcap(int K) { return spa_seg(K) / 2 < 8192 / spa4_waves(K) ? spa_seg(K) / 2 : (8192 / spa4_waves(K)) & ~63; }

// dynamic LDS of spa4_moments<K>: the segment's table + the parameter slice + a queue per wave
__host__ __device__ constexpr size_t spa4_lds_bytes(int K)
{
	return (size_t)spa_seg(K) * ((K + 2) & ~1) * 8 + (size_t)SPA4_VPER * (8 + 8 * (K + 6)) + (size_t)spa4_waves(K) * spa4_qcap(K) * 2;
}

// The flagged variants of a call sit in recs[] in two ranges: tier A from slot 0 upwards (counters[0]
// of them), tier B from slot btop - 1 downwards (counters[7]); rec index of the v-th of a tier:
__device__ __forceinline__ int spa4_rec(int tier, int btop, int v) { return tier ? btop - 1 - v : v; }

// The segment's table rows -> LDS: whole KiB pieces by LDS-DMA (no round trip through registers, all of a
// wave's pieces in flight at once), the tail of the last segment by ordinary copies.  The barrier that
// follows in the callers (__syncthreads) drains the DMA.
template <int NWAVES>
__device__ __forceinline__ void spa4_stage_table(double *tab, const double *src, int ndouble)
{
	const int tid = threadIdx.x, lane = tid & (WAVE - 1), wid = tid / WAVE;
	const int npiece = (ndouble * 8) / 1024;
	const uint8_t *s8 = reinterpret_cast<const uint8_t *>(src);
	uint8_t *d8 = reinterpret_cast<uint8_t *>(tab);
	for (int k = wid; k < npiece; k += NWAVES)
		__builtin_amdgcn_global_load_lds(
			(const __attribute__((address_space(1))) void *)(s8 + (size_t)k * 1024 + lane * 16),
			(__attribute__((address_space(3))) void *)(d8 + k * 1024), 16, 0, 0);
	for (int i = npiece * 128 + tid; i < ndouble; i += WAVE * NWAVES) tab[i] = src[i];
}

// The last workgroup to leave a moments kernel puts the item queue back to zero for the next launch on the
// stream: cursor[0] = next item, cursor[1] = workgroups that are through.
__device__ __forceinline__ void spa4_queue_done(int *cursor)
{
	if (threadIdx.x == 0 && atomicAdd(cursor + 1, 1) == (int)gridDim.x - 1) {
		__hip_atomic_store(cursor, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		__hip_atomic_store(cursor + 1, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	}
}

// One workgroup per CU.  Item = (sample segment, slice of SPA4_VPER flagged variants): the segment's X
// rows and mu and the slice's parameters are staged in LDS; one wave per variant, a lane owns SEG/64
// consecutive samples of the segment and walks its carriers in lock step with the other lanes.
template <int K, int NC>
__global__ void __launch_bounds__(WAVE * spa4_waves(K))
spa4_moments(RowsRef rr, DevModel md, int nseg, int tier, int btop, int v0, int vcap,
	const SpaRec *__restrict__ recs, const int *__restrict__ counters, double *__restrict__ segpart, int abl, int *__restrict__ cursor)
{
	constexpr int SEG = spa_seg(K), KP = (K + 2) & ~1, NS = NC + 5;
	constexpr int LDW = SEG / 16 / WAVE > 0 ? SEG / 16 / WAVE : 1;     // dwords (of 16 samples) per lane
	constexpr int NLANE = SEG / 16 / LDW;                              // lanes that own samples
	static_assert(LDW == 1 || LDW == 2 || LDW == 4, "segment sizes 512..4096");
	extern __shared__ __attribute__((aligned(16))) uint8_t fill_smem[];
	double *tab = reinterpret_cast<double *>(fill_smem);                                  // [SEG][KP]
	double *pd = tab + (size_t)SEG * KP;                                                  // [K + 6][VPER]: inv, ts, c[K], lut[4]
	int *pj = reinterpret_cast<int *>(pd + (size_t)(K + 6) * SPA4_VPER);                 // [VPER] row, [VPER] flip mask
	const int N = md.N, tid = threadIdx.x, lane = tid & (WAVE - 1), wid = tid / WAVE;
	uint16_t *q = reinterpret_cast<uint16_t *>(pj + 2 * SPA4_VPER) + wid * spa4_qcap(K);      // this wave's queue: sample in the segment | code << 14
	const int nflag = min(counters[tier ? 7 : 0] - v0, vcap);
	if (nflag <= 0) return;
	// the (segment, slice) items in segment-major order, pulled off a queue (cursor[0]): the items differ in
	// weight (carriers per slice) and a static share left workgroups idle behind the slowest; the workgroups
	// running at any moment work on the same few segments, so the tables they restage come from L2
	const int vper = spa4_slice(nflag, nseg, gridDim.x, spa4_waves(K));
	const int nslice = (nflag + vper - 1) / vper;
	const int nitem = nseg * nslice;
	__shared__ int sh_it;
	int seg = -1;
	size_t row_off = 0;                                                           // this lane's bytes of a row (piece 16 B = 64 samples)
	bool mine = false;
	int samp0 = 0;                                                                // first sample of the lane
	auto load_row = [&](int vl) -> uint4 {                                      // .x .. of the first LDW are used
		uint4 w = make_uint4(0u, 0u, 0u, 0u);
		if (mine && !(abl & 16)) {
			const uint8_t *p = rr.base + rr_piece(rr, (size_t)pj[vl], row_off >> 4) + (row_off & 15);
			if (LDW == 4) w = *reinterpret_cast<const uint4 *>(p);
			else if (LDW == 2) { const uint2 t = *reinterpret_cast<const uint2 *>(p); w.x = t.x; w.y = t.y; }
			else w.x = *reinterpret_cast<const uint32_t *>(p);
		}
		return w;
	};
	for (;;) {
		__syncthreads();                     // the previous item's readers are done
		if (tid == 0) sh_it = atomicAdd(cursor, 1);
		__syncthreads();
		const int it = sh_it;
		if (it >= nitem) break;
		const int sg = it / nslice, vb = (it - sg * nslice) * vper;
		const int nv = min(vper, nflag - vb);
		if (sg != seg) {
			seg = sg;
			const int rows = min(SEG, N - seg * SEG);
			spa4_stage_table<spa4_waves(K)>(tab, md.XM + (size_t)seg * SEG * KP, rows * KP);
			row_off = ((size_t)seg * (SEG / 16) + (size_t)lane * LDW) * 4;
			mine = lane < NLANE && row_off + 4 * LDW <= rr_row_bytes(rr);
			samp0 = seg * SEG + lane * LDW * 16;
		}
		for (int i = tid; i < nv; i += WAVE * spa4_waves(K)) {
			const SpaRec &r = recs[spa4_rec(tier, btop, v0 + vb + i)];
			pj[i] = r.j;
			pj[SPA4_VPER + i] = r.minus ? (int)0xAAAAAAAAu : 0;
			pd[i] = 1 / sqrt(r.AC2);
			pd[SPA4_VPER + i] = r.tscale;
#pragma unroll
			for (int a = 0; a < K; a++) pd[(2 + a) * SPA4_VPER + i] = r.c[a];
#pragma unroll
			for (int a = 0; a < 4; a++) pd[(2 + K + a) * SPA4_VPER + i] = r.lut[a];
		}
		__syncthreads();
		// rows two variants ahead of the one being worked on
		uint4 w0 = make_uint4(0u, 0u, 0u, 0u), w1 = w0;
		if (wid < nv) w0 = load_row(wid);
		if (wid + spa4_waves(K) < nv) w1 = load_row(wid + spa4_waves(K));
		for (int vl = wid; vl < nv; vl += spa4_waves(K)) {
			const uint4 wv = w0;
			w0 = w1;
			if (vl + 2 * spa4_waves(K) < nv) w1 = load_row(vl + 2 * spa4_waves(K));
			const double inv = pd[vl], ts = pd[SPA4_VPER + vl];
			const uint32_t zx = (uint32_t)pj[SPA4_VPER + vl];
			double c[K];
#pragma unroll
			for (int a = 0; a < K; a++) c[a] = pd[(2 + a) * SPA4_VPER + vl];
			const double *lut = pd + (2 + K) * SPA4_VPER + vl;       // dosage of code k at lut[k * VPER]
			// 0 sum mu*G, 1 sum b, 2 sum max(adj,0), 3 sum min(adj,0), 4 sum adj*mu, 5 sum adj^2 mu(1-mu),
			// 6.. kappa'_3..NC; max |adj| apart
			double acc[NS - 1], gmax = 0;
#pragma unroll
			for (int a = 0; a < NS - 1; a++) acc[a] = 0;
			// carrier masks: bit b of lo -> dword (b & 1), sample b >> 1 of it; hi the same for dwords 2, 3
			const uint32_t wd0 = wv.x, wd1 = wv.y, wd2 = wv.z, wd3 = wv.w;
			auto nzm = [&](uint32_t w, int k) -> uint32_t {
				return (mine && k < LDW) ? nz_fields((w ^ zx) & keep_mask(N - samp0 - 16 * k)) : 0u;
			};
			uint32_t lo = nzm(wd0, 0) | (nzm(wd1, 1) << 1), hi = nzm(wd2, 2) | (nzm(wd3, 3) << 1);
			if (abl & 4) lo = hi = 0;
			// Software pipeline: the table row and the dosage of a lane's next carrier are read from
			// LDS before the arithmetic of the current one, into the other of two register sets.  The reads
			// are UNCONDITIONAL (a lane without a carrier left reads its first sample's row and uses nothing
			// of it) and the bit walk is selects only: with the reads inside a branch the compiler cannot count
			// how many LDS operations are in flight at the use of the OTHER set and waits for all of them,
			// i.e. for the reads it has just issued (lgkmcnt(2..0) where it now emits lgkmcnt(5..3)).  Worth
			// nothing measurable at two waves per SIMD (0.99 ms either way: the other wave covers the wait),
			// kept because the loop is 20 instructions shorter.
			struct Car { double x[K + 1], G; bool ok; };
			const uint64_t wlo = ((uint64_t)wd1 << 32) | wd0, whi = ((uint64_t)wd3 << 32) | wd2;
			auto fetch = [&](Car &cr) {
				const bool inlo = lo != 0;
				const uint32_t m = inlo ? lo : hi;
				cr.ok = m != 0;
				const int b = cr.ok ? __ffs(m) - 1 : 0;
				const uint32_t m2 = m & (m - 1);
				lo = inlo ? m2 : lo;
				hi = inlo ? hi : m2;
				const int dw = b & 1, sid = b >> 1;                              // dword of the pair, sample in it
				const uint64_t wsel = inlo ? wlo : whi;
				const uint32_t code = (uint32_t)(wsel >> (32 * dw + 2 * sid)) & 3u;
				const double *x = tab + (size_t)((lane * LDW + (inlo ? 0 : 2) + dw) * 16 + sid) * KP;
#pragma unroll
				for (int a = 0; a <= K; a++) cr.x[a] = x[a];
				cr.G = lut[code * SPA4_VPER];
			};
			auto work = [&](const Car &cr) {
				double bb = 0;
#pragma unroll
				for (int a = 0; a < K; a++) bb = fma(cr.x[a], c[a], bb);
				const double mui = cr.x[K], G = cr.G;
				const double adj = (G - bb) * inv;
				const double u = mui * (1 - mui);
				acc[0] = fma(mui, G, acc[0]);
				acc[1] += bb;
				acc[2] += fmax(adj, 0.0); acc[3] += fmin(adj, 0.0);      // (two selects of a 64-bit pair each way cost seven instructions)
				acc[4] = fma(adj, mui, acc[4]);
				acc[5] = fma(adj * adj, u, acc[5]);
				spa4_max_abs(gmax, adj);
				// (ablation bit 1, "no cumulant terms", is a build option: tested at run time it is a scalar branch
				// per carrier that cuts the carrier's arithmetic into three scheduling regions)
#ifdef SPA4_ABLATE_TERMS
				if (!(abl & 1))
#endif
				spa4_cum_terms<NC>(adj * ts, u, 1 - 2 * mui, &acc[6]);
			};
			// A lane owns 64 samples, so the lanes' carrier counts differ (binomial): walking them in lock
			// step to the largest count would leave ~40 % of the lane-steps empty.  Instead: T =
			// ceil(mean count) lock-step rounds, then whatever the busier lanes have left goes through
			// the wave's LDS queue and is shared out evenly, 64 carriers a round.  (abl & 256: everything
			// the queue holds goes through it -- measured, no faster: filling the queue costs what the
			// idle lane-steps do.)
			const int ctot = wave_total_i(__popc(lo) + __popc(hi));
			const int T = (abl & 256) ? max(0, (ctot - spa4_qcap(K) + WAVE - 1) / WAVE) : (ctot + WAVE - 1) / WAVE;
			// exactly T fetches, none of them inside a branch of the loop body
			Car ca, cb;
			ca.ok = cb.ok = false;
			if (T > 0) {
				fetch(ca);
				int it = 1;
				for (; it + 1 < T; it += 2) {
					fetch(cb);
					if (ca.ok) work(ca);
					fetch(ca);
					if (cb.ok) work(cb);
				}
				if (it < T) {
					fetch(cb);
					if (ca.ok) work(ca);
					if (cb.ok) work(cb);
				} else if (ca.ok) work(ca);
			}
			if (__ballot((lo | hi) != 0)) {
				const int rem = __popc(lo) + __popc(hi);
				const int incl = wave_scan_incl_i(rem);
				const int nq = min(__builtin_amdgcn_readlane(incl, WAVE - 1), spa4_qcap(K));
				int o2 = incl - rem;
				while ((lo | hi) != 0 && o2 < spa4_qcap(K)) {
					const bool inlo = lo != 0;
					const int b = __ffs(inlo ? lo : hi) - 1;
					if (inlo) lo &= lo - 1; else hi &= hi - 1;
					const int dwi = (inlo ? 0 : 2) + (b & 1), sid = b >> 1;
					const uint32_t wsel = dwi == 0 ? wd0 : dwi == 1 ? wd1 : dwi == 2 ? wd2 : wd3;
					q[o2++] = (uint16_t)((lane * LDW + dwi) * 16 + sid) | (uint16_t)(((wsel >> (2 * sid)) & 3u) << 14);
				}
				__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // the wave's own LDS writes, in order
				__builtin_amdgcn_wave_barrier();
				auto fetch_q = [&](Car &cr, int k) {            // unconditional reads, as in fetch (nq > 0 here)
					cr.ok = k < nq;
					const uint32_t e = q[min(k, nq - 1)];
					const double *x = tab + (size_t)(e & 0x3FFFu) * KP;
#pragma unroll
					for (int a = 0; a <= K; a++) cr.x[a] = x[a];
					cr.G = lut[(e >> 14) * SPA4_VPER];
				};
				fetch_q(ca, lane);
				for (int k0 = 0; k0 < nq; k0 += 2 * WAVE) {          // (wave-uniform bounds)
					fetch_q(cb, k0 + WAVE + lane);
					if (ca.ok) work(ca);
					fetch_q(ca, k0 + 2 * WAVE + lane);
					if (cb.ok) work(cb);
				}
				__builtin_amdgcn_wave_barrier();      // queue reads stay before the next variant's writes
				// (more left than the queue holds: a segment of near-complete carriers behind a very
				// uneven start -- finish in lock step)
				fetch(ca);
				while (__ballot(ca.ok)) {
					fetch(cb);
					if (ca.ok) work(ca);
					ca = cb;
				}
			}
			const int idx = wave_reduce_scatter(acc, lane);
			gmax = wave_max_nonneg_to_last(gmax);
			// [variant][segment][NS]: the sums of a (variant, segment) leave in one store, and spa4_solve reads a
			// variant's segments as one contiguous run
			const size_t base = ((size_t)(vb + vl) * nseg + seg) * NS;
			if (!(lane & (NS - 1 > 16 ? 1 : 3)) && idx < NS - 1) segpart[base + idx] = acc[0];
			if (lane == WAVE - 1) segpart[base + (NS - 1)] = gmax;
		}
	}
	spa4_queue_done(cursor);
}

// Dosage rows (RAW bytes / doubles, kern_spa.h load_dosage): the same sums, but a row of imputed
// dosages is dense -- nearly every sample is a "carrier" -- so there is nothing to compact: in step j
// lane l takes sample 64 j + l of the segment (coalesced row loads), skipping the exact zeros.
template <int K, int NC, int INPUT>
__global__ void __launch_bounds__(WAVE * spa4_waves(K))
spa4_moments_ds(const void *__restrict__ rows, size_t row_bytes, DevModel md, int nseg, int tier, int btop, int v0, int vcap,
	const SpaRec *__restrict__ recs, const int *__restrict__ counters, double *__restrict__ segpart, int *__restrict__ cursor)
{
	constexpr int SEG = spa_seg(K), KP = (K + 2) & ~1, NS = NC + 5;
	extern __shared__ __attribute__((aligned(16))) uint8_t fill_smem[];
	double *tab = reinterpret_cast<double *>(fill_smem);                                  // [SEG][KP]
	double *pd = tab + (size_t)SEG * KP;                                                  // [K + 6][VPER]: inv, ts, c[K], lut[4]
	int *pj = reinterpret_cast<int *>(pd + (size_t)(K + 6) * SPA4_VPER);                 // [VPER] row, [VPER] flip flag
	const int N = md.N, tid = threadIdx.x, lane = tid & (WAVE - 1), wid = tid / WAVE;
	const int nflag = min(counters[tier ? 7 : 0] - v0, vcap);
	if (nflag <= 0) return;
	const int vper = spa4_slice(nflag, nseg, gridDim.x, spa4_waves(K));
	const int nslice = (nflag + vper - 1) / vper;
	const int nitem = nseg * nslice;
	__shared__ int sh_it;
	int seg = -1;
	for (;;) {
		__syncthreads();                     // the previous item's readers are done
		if (tid == 0) sh_it = atomicAdd(cursor, 1);
		__syncthreads();
		const int it = sh_it;
		if (it >= nitem) break;
		const int sg = it / nslice, vb = (it - sg * nslice) * vper;
		const int nv = min(vper, nflag - vb);
		if (sg != seg) {
			seg = sg;
			const int nrow = min(SEG, N - seg * SEG);
			spa4_stage_table<spa4_waves(K)>(tab, md.XM + (size_t)seg * SEG * KP, nrow * KP);
		}
		for (int i = tid; i < nv; i += WAVE * spa4_waves(K)) {
			const SpaRec &r = recs[spa4_rec(tier, btop, v0 + vb + i)];
			pj[i] = r.j;
			pj[SPA4_VPER + i] = r.minus;
			pd[i] = 1 / sqrt(r.AC2);
			pd[SPA4_VPER + i] = r.tscale;
#pragma unroll
			for (int a = 0; a < K; a++) pd[(2 + a) * SPA4_VPER + i] = r.c[a];
#pragma unroll
			for (int a = 0; a < 4; a++) pd[(2 + K + a) * SPA4_VPER + i] = r.lut[a];
		}
		__syncthreads();
		const int s_lo = seg * SEG, nstep = (min(SEG, N - s_lo) + WAVE - 1) / WAVE;
		for (int vl = wid; vl < nv; vl += spa4_waves(K)) {
			const double inv = pd[vl], ts = pd[SPA4_VPER + vl];
			const double imp = pd[(2 + K + 3) * SPA4_VPER + vl];      // lut[3]: the imputed value, flipped already
			const bool minus = pj[SPA4_VPER + vl] != 0;
			const uint8_t *row = reinterpret_cast<const uint8_t *>(rows) + (size_t)pj[vl] * row_bytes;
			double c[K];
#pragma unroll
			for (int a = 0; a < K; a++) c[a] = pd[(2 + a) * SPA4_VPER + vl];
			double acc[NS - 1], gmax = 0;
#pragma unroll
			for (int a = 0; a < NS - 1; a++) acc[a] = 0;
			auto dosage = [&](int j) -> double {               // load_dosage of kern_spa.h on the staged parameters
				const int i = s_lo + j * WAVE + lane;
				if (i >= N) return 0.0;
				if (INPUT == IN_U8) {
					const uint8_t v = row[i];
					return (v == 0xFF) ? imp : (minus ? 2.0 - (double)v : (double)v);
				} else {
					const double v = reinterpret_cast<const double *>(row)[i];
					return !isfinite(v) ? imp : (minus ? 2.0 - v : v);
				}
			};
			double gn = dosage(0);
			for (int j = 0; j < nstep; j++) {
				const double G = gn;
				if (j + 1 < nstep) gn = dosage(j + 1);
				if (G != 0) {
					const double *x = tab + (size_t)(j * WAVE + lane) * KP;
					double bb = 0;
#pragma unroll
					for (int a = 0; a < K; a++) bb = fma(x[a], c[a], bb);
					const double mui = x[K];
					const double adj = (G - bb) * inv;
					const double u = mui * (1 - mui);
					acc[0] = fma(mui, G, acc[0]);
					acc[1] += bb;
					acc[2] += fmax(adj, 0.0); acc[3] += fmin(adj, 0.0);      // (two selects of a 64-bit pair each way cost seven instructions)
					acc[4] = fma(adj, mui, acc[4]);
					acc[5] = fma(adj * adj, u, acc[5]);
					spa4_max_abs(gmax, adj);
					spa4_cum_terms<NC>(adj * ts, u, 1 - 2 * mui, &acc[6]);
				}
			}
			const int idx = wave_reduce_scatter(acc, lane);
			gmax = wave_max_nonneg_to_last(gmax);
			// [variant][segment][NS]: the sums of a (variant, segment) leave in one store, and spa4_solve reads a
			// variant's segments as one contiguous run
			const size_t base = ((size_t)(vb + vl) * nseg + seg) * NS;
			if (!(lane & (NS - 1 > 16 ? 1 : 3)) && idx < NS - 1) segpart[base + idx] = acc[0];
			if (lane == WAVE - 1) segpart[base + (NS - 1)] = gmax;
		}
	}
	spa4_queue_done(cursor);
}

// the carrier series of one variant
template <int NC>
struct Spa4Series {
	double k1, k2;               // kappa_1 = sum g mu, kappa_2 = sum g^2 mu (1 - mu)
	double kp[NC - 2];           // kappa'_n = kappa_n ts^n / n!, n = 3..NC
	double ts, gmax;
};

// K1 (without "- q"), K2 and Korg sums at t (SPATest.cpp:49,64,79); false when the series cannot
// be trusted there
template <int NC>
__device__ __forceinline__ bool spa4_eval(const Spa4Series<NC> &S, double xmax, double t, double &K1s, double &K2s, double &K0s)
{
	const double its = 1 / S.ts, tau = t * its;       // ts is a power of two
	double p0 = 0, p1 = 0, p2 = 0;
#pragma unroll
	for (int n = NC; n >= 3; n--) {
		const double kn = S.kp[n - 3];
		p0 = fma(p0, tau, kn);
		p1 = fma(p1, tau, (double)n * kn);
		p2 = fma(p2, tau, (double)(n * (n - 1)) * kn);
	}
	const double tau2 = tau * tau;
	K0s = fma(t, S.k1, fma(0.5 * t * t, S.k2, p0 * tau2 * tau));
	K1s = S.k1 + fma(t, S.k2, p1 * tau2 * its);
	K2s = fma(p2 * tau, its * its, S.k2);
	// the last two terms of the K2 series, tau^(NC-2) and tau^(NC-3)
	double tp = tau;                    // tau^(NC-3)
#pragma unroll
	for (int k = 1; k < NC - 3; k++) tp *= tau;
	const double tail = (fabs((double)(NC * (NC - 1)) * S.kp[NC - 3] * tp * tau) +
		fabs((double)((NC - 1) * (NC - 2)) * S.kp[NC - 4] * tp)) * its * its;
	return S.gmax * fabs(t) <= xmax && tail <= SPA4_TAIL_TOL * fabs(K2s) && isfinite(K2s);
}

// getroot_K1_fast (SPATest.cpp:139-184) on the series; false = the series does not hold somewhere
template <int NC>
__device__ __forceinline__ bool spa4_root(const Spa4Series<NC> &S, double xmax, double q, double NAmu, double NAsigma, RootState &s)
{
	root_begin(s, q, 0, 0);
	root_feed(s, S.k1, S.k2, NAmu, NAsigma, 0.0, true);      // t = 0: K1 = kappa_1, K2 = kappa_2, Korg = 0
	while (s.active) {
		double K1s, K2s, K0s;
		if (!spa4_eval(S, xmax, s.tnew, K1s, K2s, K0s)) return false;
		root_feed(s, K1s, K2s, NAmu, NAsigma, K0s, true);
	}
	return true;
}

template <int K, int NC>
__device__ __forceinline__ void spa4_solve_one(const DevModel &md, int nseg, int tier, int btop, int v0, int vcap, int v, int lane,
	SpaRec *__restrict__ recs, int *__restrict__ counters, const double *__restrict__ segpart,
	int *__restrict__ fb_dense, int *__restrict__ fb_exact, double *__restrict__ out8, int force_dense, int force_exact);

// one wave per flagged variant of the tier's round [v0, v0 + vcap).  A tier-A variant whose series
// is too short is handed to tier B (a copy of its record at the end of that range); from tier B it goes
// to the exact kernels.
template <int K, int NC>
__global__ void __launch_bounds__(256)
spa4_solve(DevModel md, int nseg, int tier, int btop, int v0, int vcap, SpaRec *__restrict__ recs, int *__restrict__ counters,
	const double *__restrict__ segpart, int *__restrict__ fb_dense, int *__restrict__ fb_exact,
	double *__restrict__ out8, int force_dense, int force_exact)
{
	// one wave per variant: the lanes share the segments' partial sums (fixed tree), then all of
	// them run the scalar part on identical values and lane 0 writes
	const int lane = threadIdx.x & (WAVE - 1);
	const int nflag = min(counters[tier ? 7 : 0] - v0, vcap);
	for (int v = (blockIdx.x * blockDim.x + threadIdx.x) / WAVE; v < nflag; v += gridDim.x * blockDim.x / WAVE)
		spa4_solve_one<K, NC>(md, nseg, tier, btop, v0, vcap, v, lane, recs, counters, segpart, fb_dense, fb_exact, out8,
			force_dense, force_exact);
}

template <int K, int NC>
__device__ __forceinline__ void spa4_solve_one(const DevModel &md, int nseg, int tier, int btop, int v0, int vcap, int v, int lane,
	SpaRec *__restrict__ recs, int *__restrict__ counters, const double *__restrict__ segpart,
	int *__restrict__ fb_dense, int *__restrict__ fb_exact, double *__restrict__ out8, int force_dense, int force_exact)
{
	constexpr int NS = NC + 5;
	const int ri = spa4_rec(tier, btop, v0 + v);
	const SpaRec r = recs[ri];
	double a[NS];
#pragma unroll
	for (int x = 0; x < NS; x++) a[x] = 0;
	for (int s = lane; s < nseg; s += WAVE) {
		const double *p = segpart + ((size_t)v * nseg + s) * NS;
#pragma unroll
		for (int x = 0; x < NS - 1; x++) a[x] += p[x];
		a[NS - 1] = fmax(a[NS - 1], p[NS - 1]);
	}
#pragma unroll
	for (int x = 0; x < NS - 1; x++) a[x] = wave_sum(a[x]);
#pragma unroll
	for (int o = 32; o > 0; o >>= 1) a[NS - 1] = fmax(a[NS - 1], __shfl_xor(a[NS - 1], o, WAVE));
	// Lanes 0 and 1 go on with identical values: each runs one of the two root searches (q and 2 m1 - q),
	// lane 0 collects and writes.
	if (lane > 1) return;
	// scalars of saige_main.cpp:369-381
	const double inv = 1 / sqrt(r.AC2);
	double xmu_c = 0, xsum_c = 0;
#pragma unroll
	for (int x = 0; x < K; x++) { xmu_c = fma(md.Xmu[x], r.c[x], xmu_c); xsum_c = fma(md.Xsum[x], r.c[x], xsum_c); }
	const double Tstat = r.S * inv;                       // q - m1, :380
	const double var2 = r.var2 / r.AC2, var1 = var2 * md.r;
	const double m1 = (a[0] - xmu_c) * inv;
	const double qtilde = Tstat / sqrt(var1) * sqrt(var2) + m1;   // :381
	const double sdev = qtilde - m1, qinv = -sdev + m1;
	const double pn_in = d_pchisq1_upper(sdev * sdev / var2);
	if (fabs(qtilde - m1) / sqrt(var2) < 2.0) {           // SPATest.cpp:319-321 (the score epilogue takes
		if (lane == 0) spa_write_row(r, Tstat, var1, pn_in, true, out8);  // these out already; kept for rounding at the edge)
		return;
	}
	// g_pos / g_neg bound test (kern_spa2.h)
	const double nb = (xsum_c - a[1]) * inv;
	const double L = a[2] + fmax(-nb, 0.0), U = a[3] + fmin(-nb, 0.0);
	const double mar = 1e-9 * (fabs(L) + fabs(U) + fabs(qtilde) + fabs(qinv));
	if (force_dense || !(qtilde < L - mar && qtilde > U + mar && qinv < L - mar && qinv > U + mar)) {
		if (lane == 0) fb_dense[atomicAdd(&counters[2], 1)] = ri;
		return;
	}
	Spa4Series<NC> S;
	S.k1 = a[4]; S.k2 = a[5]; S.ts = r.tscale; S.gmax = a[NS - 1];
#pragma unroll
	for (int x = 0; x < NC - 2; x++) S.kp[x] = a[6 + x];
	const double NAmu = m1 - a[4], NAsigma = var2 - a[5];
	const double qmine = lane == 0 ? qtilde : qinv;
	RootState sm;
	const bool ok_mine = !force_exact && spa4_root(S, md.spa_xmax, qmine, NAmu, NAsigma, sm);
	double p_mine = 0;
	if (ok_mine && sm.converged) p_mine = lugannani_rice(sm.root, sm.Kcur, sm.K2cur, qmine, NAmu, NAsigma);
	const int st_mine = (ok_mine ? 1 : 0) | (ok_mine && sm.converged ? 2 : 0);
	const double p_other = __shfl(p_mine, 1, WAVE);         // lane 0 reads lane 1's root
	const int st_other = __shfl(st_mine, 1, WAVE);
	if (lane != 0) return;
	if (!(st_mine & 1) || !(st_other & 1)) {
		if (tier == 0 && !force_exact) {
			const int slot = atomicAdd(&counters[7], 1);       // on to the longer series
			atomicAdd(&counters[6], 1);
			recs[btop - 1 - slot] = r;
		} else {
			fb_exact[atomicAdd(&counters[4], 1)] = ri;      // the exact list (the per-variant series is this same series)
		}
		return;
	}
	double pval;
	bool converged = true;
	if ((st_mine & 2) && (st_other & 2)) {
		pval = fabs(p_mine) + fabs(p_other);
		if (pval != 0 && pn_in / pval > 1000) pval = pn_in;   // SPATest.cpp:368-371
	} else {
		pval = pn_in;
		converged = false;
	}
	spa_write_row(r, Tstat, var1, pval, converged, out8);
}

// ---------------------------------------------------------------------------
// Exact path for the variants the series does not cover (rare variants: few carriers, large g t; or
// a strong signal): one workgroup of 16 waves per variant, workgroups pull variants from a queue,
// the longer lists first.  The workgroup scans the packed row into an index list (ascending sample
// order), turns it into the (adj, mu) list (gathers of the X rows) with the carrier sums, and then
// runs Saddle_Prob_Fast exactly as the reference does -- every K1/K2 evaluation one sweep of the
// workgroup over its list, both roots in the same sweep, Korg with the sweep that is predicted to be
// the last (kern_spa2.h).  The lists live in the workgroup's slice of global scratch (N entries: any
// variant fits).

#define SPA5_BIG 2048            /* lists longer than this are started first (they set the kernel's tail) */

// K1, K2 (SPATest.cpp:64,79-80) and Korg (:49) terms of one carrier at t; the Korg term switched at run time (workgroup-uniform)
__device__ __forceinline__ void cgf_terms_rt(double g, double m, double t, bool with_k, double &k1, double &k2, double &k0)
{
	const double om = 1 - m, mg = m * g, c2 = om * mg * g;
	const double e = fast_exp(-g * t);
	const double d = fma(om, e, m);
	const double rr = isfinite(d) ? fast_rcp(d) : 0.0;
	k1 = fma(mg, rr, k1);
	const double tt = c2 * e * rr * rr;
	if (isfinite(tt)) k2 += tt;
	if (with_k) k0 += isfinite(d) ? fma(g, t, fast_log(d)) : fast_log(fma(m, fast_exp(g * t), om));
}

// bytes of scratch per workgroup: (adj, mu) list + index list, N entries each
__host__ __device__ inline size_t spa5_wg_bytes(int N) { return (((size_t)N + 63) & ~(size_t)63) * (16 + 4); }

// this thread's share of the cumulant sums over a variant's list (a call: the polynomial constants stay
// out of the caller's registers)
template <int BLOCK>
__device__ __noinline__ void spa5_cum_sweep(const double2 *__restrict__ glist, int nnz, double ts, double *kp_out)
{
	constexpr int NC = SPA4_NCB;
	double kp[NC - 1];
#pragma unroll
	for (int a = 0; a < NC - 1; a++) kp[a] = 0;
	double2 gn = (int)threadIdx.x < nnz ? glist[threadIdx.x] : make_double2(0.0, 0.5);
	for (int k = threadIdx.x; k < nnz; k += BLOCK) {
		const double2 gm = gn;
		if (k + BLOCK < nnz) gn = glist[k + BLOCK];                  // one step ahead
		const double mui = gm.y;
		spa4_cum_terms<NC>(gm.x * ts, mui * (1 - mui), 1 - 2 * mui, kp);
		kp[NC - 2] = fmax(kp[NC - 2], fabs(gm.x));
	}
#pragma unroll
	for (int a = 0; a < NC - 1; a++) kp_out[a] = kp[a];
}

// thread 0 of spa5_series: both root searches and the tail on the cumulant sums in arg[]
__device__ __noinline__ int spa5_series_solve(const double *arg, const SpaRec *__restrict__ rec, double *__restrict__ out8)
{
	constexpr int NC = SPA4_NCB;
	Spa4Series<NC> S;
#pragma unroll
	for (int a = 0; a < NC - 2; a++) S.kp[a] = arg[a];
	S.k1 = arg[NC - 2]; S.k2 = arg[NC - 1]; S.ts = arg[NC]; S.gmax = arg[NC + 1];
	const double xmax = arg[NC + 2], qtilde = arg[NC + 3], qinv = arg[NC + 4], NAmu = arg[NC + 5], NAsigma = arg[NC + 6];
	const double pn_in = arg[NC + 7], Tstat = arg[NC + 8], var1 = arg[NC + 9];
	RootState s1, s2;
	if (!spa4_root(S, xmax, qtilde, NAmu, NAsigma, s1) || !spa4_root(S, xmax, qinv, NAmu, NAsigma, s2)) return 0;
	double pval;
	bool converged = true;
	if (s1.converged && s2.converged) {
		const double p1 = lugannani_rice(s1.root, s1.Kcur, s1.K2cur, qtilde, NAmu, NAsigma);
		const double p2 = lugannani_rice(s2.root, s2.Kcur, s2.K2cur, qinv, NAmu, NAsigma);
		pval = fabs(p1) + fabs(p2);
		if (pval != 0 && pn_in / pval > 1000) pval = pn_in;   // SPATest.cpp:368-371
	} else {
		pval = pn_in;
		converged = false;
	}
	const SpaRec rr = *rec;
	spa_write_row(rr, Tstat, var1, pval, converged, out8);
	return 1;
}

// The series on a variant's (adj, mu) list (spa5_kernel): one sweep for the cumulant sums, then thread 0
// runs both root searches and the tail on them.  Returns (to every thread) whether the row was written;
// false = some evaluation point lies outside what sixteen cumulants cover: the caller goes on with the
// exact sweeps.  sh: NC + 16 + (NC - 1) * (BLOCK / 64) doubles of shared memory.
template <int BLOCK>
__device__ __forceinline__ bool spa5_series(const double2 *__restrict__ glist, int nnz, double ts, double xmax,
	double k1, double k2, double qtilde, double qinv, double NAmu, double NAsigma, double pn_in,
	double Tstat, double var1, const SpaRec *__restrict__ rec, double *__restrict__ out8, double *sh, int *sh_flag, int *prof = nullptr)
{
	constexpr int NC = SPA4_NCB;
	double kp[NC - 1];                 // kappa'_3..NC, and max |adj| in the last slot
#ifdef SPA5_PROF
	long long tp_ = wall_clock64();
#define SPA5_TS(ph) do { if (threadIdx.x == 0) { const long long now_ = wall_clock64(); atomicAdd(&prof[ph], (int)(now_ - tp_)); tp_ = now_; } } while (0)
#else
#define SPA5_TS(ph) do { } while (0)
#endif
	spa5_cum_sweep<BLOCK>(glist, nnz, ts, kp);
	SPA5_TS(0);
	// Only thread 0 needs the totals: a reduce-scatter inside each wave (wave_reduce_scatter: ~NC shuffles
	// instead of 6 NC), the waves' partial sums through LDS, NC - 2 threads add them up
	constexpr int NWV = BLOCK / WAVE;
	double gmax = kp[NC - 2];
	const int lane = threadIdx.x & (WAVE - 1), wv = threadIdx.x / WAVE;
	double kr[NC - 2];
#pragma unroll
	for (int a = 0; a < NC - 2; a++) kr[a] = kp[a];
	const int idx = wave_reduce_scatter(kr, lane);
#pragma unroll
	for (int o = 32; o > 0; o >>= 1) gmax = fmax(gmax, __shfl_xor(gmax, o, WAVE));
	double *part = sh + NC + 16;                                  // [NC - 2][NWV] partial sums, [NWV] maxima
	if (!(lane & 3) && idx < NC - 2) part[idx * NWV + wv] = kr[0];
	if (lane == 1) part[(NC - 2) * NWV + wv] = gmax;
	__syncthreads();
	double *arg = sh;
	if (threadIdx.x < NC - 2) {
		double t = 0;
		for (int w = 0; w < NWV; w++) t += part[threadIdx.x * NWV + w];
		arg[threadIdx.x] = t;
	}
	__syncthreads();
	SPA5_TS(1);
	if (threadIdx.x == 0) {
		for (int w = 0; w < NWV; w++) gmax = fmax(gmax, part[(NC - 2) * NWV + w]);
		// the scalar part as a call: its registers (two root searches on sixteen cumulants) stay out of
		// the workgroup's budget
		arg[NC - 2] = k1; arg[NC - 1] = k2; arg[NC] = ts; arg[NC + 1] = gmax; arg[NC + 2] = xmax;
		arg[NC + 3] = qtilde; arg[NC + 4] = qinv; arg[NC + 5] = NAmu; arg[NC + 6] = NAsigma;
		arg[NC + 7] = pn_in; arg[NC + 8] = Tstat; arg[NC + 9] = var1;
		*sh_flag = spa5_series_solve(arg, rec, out8);
	}
	SPA5_TS(2);
	__syncthreads();
	const bool done = *sh_flag != 0;
	__syncthreads();                   // sh and the flag are free again
	return done;
}

// MODE 0: the series on the list; variants it does not cover are appended to `todo_next` (counters[4]).
// MODE 1: the exact sweeps (over that list).
// BLOCK: 512 threads per variant, or 128 where the packed row is short enough for four workgroups per CU
// (small N: the kernel is bound by the number of variants in flight, not by the work in one).
// Registers: the 512-thread form is two waves per SIMD at up to 256 registers -- one workgroup per CU then holds
// the CU's whole register file while it waits on its ~10 dependent phases.  Capped at 128 (SPA5_REG_WAVES 4) half
// of every register file would stay free for the other lane's small kernels, but the kernel spills and the
// SPA stage alone goes from 1.26 to 1.65 ms (C3 2.24 -> 2.53-2.57 ms per step): measured, not adopted.
#ifndef SPA5_REG_WAVES
#define SPA5_REG_WAVES 2
#endif
#define SPA5_WAVES_PER_SIMD(BLOCK) ((BLOCK) == 512 ? SPA5_REG_WAVES : ((BLOCK) / 64 + 3) / 4)
#ifdef SPA5_PROF   /* phase times of spa5_kernel in 10-ns ticks -> counters[8 + 8 MODE + phase] (diagnostic build only) */
#define SPA5_T(ph) do { if (tid == 0) { const long long now_ = wall_clock64(); atomicAdd(&counters[8 + 8 * MODE + (ph)], (int)(now_ - tprev_)); tprev_ = now_; } } while (0)
#else
#define SPA5_T(ph) do { } while (0)
#endif
template <int K, int INPUT, int MODE, int BLOCK>
__global__ void __launch_bounds__(BLOCK, SPA5_WAVES_PER_SIMD(BLOCK))
spa5_kernel(RowsRef rr, DevModel md, const SpaRec *__restrict__ recs,
	int *__restrict__ counters, const int *__restrict__ todo, int *__restrict__ todo_next, int *__restrict__ cursor,
	int *__restrict__ fb_dense, uint8_t *__restrict__ scratch, double *__restrict__ out8, int force_dense, int force_exact,
	size_t lds_row_bytes, int only, size_t wg_stride, int nslot)
{
	constexpr int KP = (K + 2) & ~1, NW = BLOCK / WAVE;
	extern __shared__ __attribute__((aligned(16))) uint8_t fill_smem[];
	uint4 *rows_lds = reinterpret_cast<uint4 *>(fill_smem);
	__shared__ double sh[SPA4_NCB * NW + SPA4_NCB + 16];
	__shared__ int sh_flag;
	__shared__ int shi[NW];
	__shared__ int sh_vi, sh_v;
	__shared__ SpaRec sh_rec;
	const int N = md.N, tid = threadIdx.x, lane = tid & (WAVE - 1), wid = tid / WAVE;
	// only = 2: the block's carrier lists (rr.cptr) are not used (diagnostic: every variant scans its row); 0: used
	double2 *glist = reinterpret_cast<double2 *>(scratch + (size_t)blockIdx.x * wg_stride);
	uint32_t *ilist = reinterpret_cast<uint32_t *>(glist + (((size_t)N + 63) & ~(size_t)63));
	// length of the list: counters[3] / [4], or a copy taken when the list was complete (a launch that runs
	// beside a kernel that still appends to it)
	const int ntodo = counters[nslot];
	// pieces of 64 samples: a uint4 of a packed row, or 64 dosages
	const int nvec = INPUT == IN_2BIT ? (int)(min((size_t)((N + 63) >> 6) * 16, rr_row_bytes(rr)) / 16) : (N + 63) >> 6;
	const int per = ((nvec + NW - 1) / NW + WAVE - 1) & ~(WAVE - 1);          // pieces per wave, whole wave steps
	for (;;) {
		__syncthreads();                         // sh_vi, shi and the lists of the previous variant are free
		// wave 0 takes the next variant off the queue and its record into LDS (one coalesced read; the
		// record's fields are then a few cycles away instead of a global latency each).  Two rounds
		// over the list: the long lists first, then the short ones.
		if (tid < WAVE) {
			int vi0 = 0;
			if (tid == 0) vi0 = atomicAdd(cursor, 1);
			vi0 = __builtin_amdgcn_readfirstlane(vi0);
			int v0 = -1;
			if (vi0 < 2 * ntodo) {
				v0 = todo[vi0 < ntodo ? vi0 : vi0 - ntodo];
				const double *src = reinterpret_cast<const double *>(recs + v0);
				for (int i = tid; i < (int)(sizeof(SpaRec) / 8); i += WAVE) reinterpret_cast<double *>(&sh_rec)[i] = src[i];
			}
			if (tid == 0) { sh_vi = vi0; sh_v = v0; }
		}
		__syncthreads();
		const int vi = sh_vi;
		if (vi >= 2 * ntodo) break;
#ifdef SPA5_PROF
		long long tprev_ = wall_clock64();
#endif
		const bool big_round = vi < ntodo;
		const int v = sh_v;
		const SpaRec &r = sh_rec;
		if ((r.nnz > SPA5_BIG) != big_round) continue;
		auto row_piece = [&](int p) -> uint4 { return *reinterpret_cast<const uint4 *>(rr.base + rr_piece(rr, (size_t)r.j, (size_t)p)); };
		const uint32_t zx = r.minus ? 0xAAAAAAAAu : 0u;
		const double lut0 = r.lut[0], lut1 = r.lut[1], lut2 = r.lut[2], lut3 = r.lut[3];
		// The block's carrier list of the variant, when it has one for this orientation: sample | code << 30 in
		// ascending order, r.nnz entries -- what the three phases below build from the row otherwise.
		const bool listed = INPUT == IN_2BIT && only != 2 && rr.cptr != nullptr && rr.corient[r.j] == (r.minus ? 2 : 1);
		const uint32_t *il = listed ? rr.cidx + rr.cptr[r.j] : ilist;
		int nnz = listed ? r.nnz : 0;
		if (!listed) {
			// ---- index list: sample | code << 30, ascending.  Wave w owns pieces [w per, (w+1) per).
			// ww/z: codes and carrier bits of piece p (2-bit rows); dosage rows: z[0], z[1] = a 64-bit mask of
			// the piece's non-zero dosages, bit s = sample 64 p + s
			// 2-bit rows that fit go through LDS: one streaming copy with many loads in flight instead of two
			// latency-bound passes over global memory (rows_lds: dynamic shared memory of nvec uint4, or none)
			const bool staged = INPUT == IN_2BIT && lds_row_bytes >= (size_t)nvec * 16;
			if (staged) {
				constexpr int UN = 8;
				for (int p0 = tid; p0 < nvec; p0 += UN * BLOCK) {
					uint4 t[UN];
	#pragma unroll
					for (int j = 0; j < UN; j++) { const int p = p0 + j * BLOCK; t[j] = p < nvec ? row_piece(p) : make_uint4(0u, 0u, 0u, 0u); }
	#pragma unroll
					for (int j = 0; j < UN; j++) { const int p = p0 + j * BLOCK; if (p < nvec) rows_lds[p] = t[j]; }
				}
				__syncthreads();
			}
			SPA5_T(0);
			auto masks = [&](int p, uint32_t (&ww)[4], uint32_t (&z)[4]) -> int {
				int cnt = 0;
				if (INPUT == IN_2BIT) {
					uint4 w = make_uint4(0u, 0u, 0u, 0u);
					if (p < nvec) w = staged ? rows_lds[p] : row_piece(p);
					ww[0] = w.x; ww[1] = w.y; ww[2] = w.z; ww[3] = w.w;
	#pragma unroll
					for (int k = 0; k < 4; k++) { z[k] = nz_fields((ww[k] ^ zx) & keep_mask(N - p * 64 - 16 * k)); cnt += __popc(z[k]); }
				} else {
					z[0] = z[1] = z[2] = z[3] = 0;
					if (p < nvec) {
						for (int s2 = 0; s2 < 64; s2++) {
							const int i = p * 64 + s2;
							if (i < N && load_dosage<INPUT>(rr, i, r) != 0) { z[s2 >> 5] |= 1u << (s2 & 31); cnt++; }
						}
					}
				}
				return cnt;
			};
			int cnt_w = 0;
			for (int it = 0; it < per; it += WAVE) {
				uint32_t ww[4], z[4];
				cnt_w += masks(wid * per + it + lane, ww, z);
			}
			cnt_w = wave_sum_i(cnt_w);
			if (lane == 0) shi[wid] = cnt_w;
			__syncthreads();
			int o_w = 0;
	#pragma unroll
			for (int w = 0; w < NW; w++) { if (w < wid) o_w += shi[w]; nnz += shi[w]; }
			SPA5_T(1);
			for (int it = 0; it < per; it += WAVE) {
				uint32_t ww[4], z[4];
				const int p = wid * per + it + lane;
				const int cnt = masks(p, ww, z);
				if (!__ballot(cnt != 0)) continue;
				int incl = cnt;
	#pragma unroll
				for (int o = 1; o < WAVE; o <<= 1) {
					const int up = __shfl_up(incl, o, WAVE);
					if (lane >= o) incl += up;
				}
				int o2 = o_w + incl - cnt;
				o_w += __shfl(incl, WAVE - 1, WAVE);
				if (INPUT == IN_2BIT) {
	#pragma unroll
					for (int k = 0; k < 4; k++) {
						uint32_t zz = z[k];
						while (zz) {
							const int b = __ffs(zz) - 1;
							zz &= zz - 1;
							ilist[o2++] = (uint32_t)(p * 64 + 16 * k + (b >> 1)) | (((ww[k] >> b) & 3u) << 30);
						}
					}
				} else {
	#pragma unroll
					for (int k = 0; k < 2; k++) {
						uint32_t zz = z[k];
						while (zz) {
							const int b = __ffs(zz) - 1;
							zz &= zz - 1;
							ilist[o2++] = (uint32_t)(p * 64 + 32 * k + b);
						}
					}
				}
			}
			__syncthreads();                         // publishes the index list to the workgroup
			SPA5_T(2);
		}
		// ---- (adj, mu) list + carrier sums (kern_spa2.h)
		const double inv = 1 / sqrt(r.AC2);
		double c[K];
#pragma unroll
		for (int a = 0; a < K; a++) c[a] = r.c[a];
		double a6[6] = {0, 0, 0, 0, 0, 0};
		// (UNG independent index reads, then UNG independent row gathers per thread and step: the loop is
		// bound by the two global latencies in a row, not by its arithmetic)
		constexpr int UNG = K <= 4 ? 4 : 2;
		for (int k0 = tid; k0 < ((force_exact & 128) ? 0 : nnz); k0 += UNG * BLOCK) {
			uint32_t e[UNG];
			double xv[UNG][KP];
#pragma unroll
			for (int j = 0; j < UNG; j++) e[j] = (k0 + j * BLOCK < nnz) ? il[k0 + j * BLOCK] : 0u;
#pragma unroll
			for (int j = 0; j < UNG; j++) {
				const double *x = md.XM + (size_t)(e[j] & 0x3FFFFFFFu) * KP;
#pragma unroll
				for (int a = 0; a < KP; a += 2) {
					const double2 t2 = *reinterpret_cast<const double2 *>(x + a);
					xv[j][a] = t2.x; xv[j][a + 1] = t2.y;
				}
			}
#pragma unroll
			for (int j = 0; j < UNG; j++) {
				const int k = k0 + j * BLOCK;
				if (k >= nnz) continue;
				const uint32_t code = e[j] >> 30;
				const double G = INPUT == IN_2BIT ? ((code & 2u) ? ((code & 1u) ? lut3 : lut2) : ((code & 1u) ? lut1 : lut0))
					: load_dosage<INPUT>(rr, (int)(e[j] & 0x3FFFFFFFu), r);
				double b = 0;
#pragma unroll
				for (int a = 0; a < K; a++) b = fma(xv[j][a], c[a], b);
				const double mui = xv[j][K];
				const double adj = (G - b) * inv;
				glist[k] = make_double2(adj, mui);
				a6[0] = fma(mui, G, a6[0]);
				a6[1] += b;
				if (adj > 0) a6[2] += adj; else a6[3] += adj;
				a6[4] = fma(adj, mui, a6[4]);
				a6[5] = fma(adj * adj, mui * (1 - mui), a6[5]);
			}
		}
		block_sum<6, BLOCK>(a6, sh);        // its barriers also publish the list
		SPA5_T(3);
		// ---- scalars of saige_main.cpp:369-381, then Saddle_Prob_Fast
		double xmu_c = 0, xsum_c = 0;
#pragma unroll
		for (int a = 0; a < K; a++) { xmu_c = fma(md.Xmu[a], c[a], xmu_c); xsum_c = fma(md.Xsum[a], c[a], xsum_c); }
		const double m1 = (a6[0] - xmu_c) * inv;
		const double Tstat = r.S * inv;
		const double var2 = r.var2 / r.AC2, var1 = var2 * md.r;
		const double qtilde = Tstat / sqrt(var1) * sqrt(var2) + m1;
		const double sdev = qtilde - m1, qinv = -sdev + m1;
		const double pn_in = d_pchisq1_upper(sdev * sdev / var2);
		double pval;
		bool converged = true;
		if (fabs(qtilde - m1) / sqrt(var2) < 2.0) {
			pval = pn_in;
		} else {
			const double nb = (xsum_c - a6[1]) * inv;
			const double L = a6[2] + fmax(-nb, 0.0), U = a6[3] + fmin(-nb, 0.0);
			const double mar = 1e-9 * (fabs(L) + fabs(U) + fabs(qtilde) + fabs(qinv));
			if (force_dense || !(qtilde < L - mar && qtilde > U + mar && qinv < L - mar && qinv > U + mar)) {
				if (tid == 0) fb_dense[atomicAdd(&counters[2], 1)] = v;
				continue;
			}
			const double NAmu = m1 - a6[4], NAsigma = var2 - a6[5];
			if (MODE == 0) {
				// the series: one sweep over the list instead of one per Newton step
				if ((force_exact & 1) || !spa5_series<BLOCK>(glist, nnz, r.tscale, md.spa_xmax, a6[4], a6[5], qtilde, qinv, NAmu, NAsigma,
						pn_in, Tstat, var1, &sh_rec, out8, sh, &sh_flag, counters + 13)) {
					if (tid == 0) todo_next[atomicAdd(&counters[4], 1)] = v;     // on to the exact exp/log sweeps
				}
				SPA5_T(4);
				continue;
			}
			RootState s1, s2;
			root_begin(s1, qtilde, L, U);
			root_begin(s2, qinv, L, U);
			root_feed(s1, a6[4], a6[5], NAmu, NAsigma, 0.0, true);      // t = 0 needs no sweep: exp(0) = 1
			root_feed(s2, a6[4], a6[5], NAmu, NAsigma, 0.0, true);
			// one sweep: K1, K2 (and Korg where wanted) of the active roots
			auto sweep = [&](bool a1, bool a2, bool k1w, bool k2w, double t1, double t2, double (&sv)[6]) {
#pragma unroll
				for (int a = 0; a < 6; a++) sv[a] = 0;
				double2 gn = tid < nnz ? glist[tid] : make_double2(0.0, 0.5);            // g = 0 adds exactly 0
				for (int k0 = tid; k0 < nnz; k0 += BLOCK) {
					const double2 gm = gn;
					if (k0 + BLOCK < nnz) gn = glist[k0 + BLOCK];              // one step ahead
					if (a1) cgf_terms_rt(gm.x, gm.y, t1, k1w, sv[0], sv[1], sv[2]);
					if (a2) cgf_terms_rt(gm.x, gm.y, t2, k2w, sv[3], sv[4], sv[5]);
				}
				block_sum<6, BLOCK>(sv, sh);
			};
			while ((s1.active || s2.active) && !(force_exact & 64)) {
				const bool a1 = s1.active, a2 = s2.active, k1w = s1.want_k, k2w = s2.want_k;
				double sv[6];
				sweep(a1, a2, k1w, k2w, s1.tnew, s2.tnew, sv);
				if (a1) root_feed(s1, sv[0], sv[1], NAmu, NAsigma, sv[2], k1w);
				if (a2) root_feed(s2, sv[3], sv[4], NAmu, NAsigma, sv[5], k2w);
			}
			if (s1.converged && s2.converged) {
				if (!(s1.k_ok && s2.k_ok)) {
					// a search ended at a point evaluated without Korg (root_step's guess was wrong)
					double sv[6];
					sweep(!s1.k_ok, !s2.k_ok, true, true, s1.root, s2.root, sv);
					if (!s1.k_ok) s1.Kcur = sv[2];
					if (!s2.k_ok) s2.Kcur = sv[5];
				}
				const double p1 = lugannani_rice(s1.root, s1.Kcur, s1.K2cur, qtilde, NAmu, NAsigma);
				const double p2 = lugannani_rice(s2.root, s2.Kcur, s2.K2cur, qinv, NAmu, NAsigma);
				pval = fabs(p1) + fabs(p2);
				if (pval != 0 && pn_in / pval > 1000) pval = pn_in;   // SPATest.cpp:368-371
			} else {
				pval = pn_in;
				converged = false;
			}
		}
		if (tid == 0) spa_write_row(r, Tstat, var1, pval, converged, out8);
		SPA5_T(5);
	}
}
