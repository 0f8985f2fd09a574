// kern_spa3.h -- SPA stage v3 for 2-bit genotypes: level-synchronous Newton.
// Part of libsaigehip.so (single translation unit: saigehip.hip).
#pragma once

// The per-variant cost of the saddlepoint stage is proportional to the number
// of carriers (10 ... 320 000 at N = 430K), so "one workgroup per variant"
// (kern_spa2.h) leaves the chip waiting for the few largest variants.  Here the
// Newton iteration of ALL flagged variants advances in lock step and every
// launch is balanced over fixed-size pieces of the carrier lists:
//
//   spa3_count  carriers per (variant, sample segment); segment = as many samples as have their
//               X rows and mu in 128 KiB of LDS (4096 at K = 3)
//   spa3_plan   per variant: scalars that need no carrier pass (m1, q~, the
//               cutoff exit of SPATest.cpp:319-321),
//               exclusive scan over its segments, arena allocation
//   spa3_fill   per (segment, slice of the variants): the segment's X rows and mu are staged in
//               LDS once, then one wave per variant turns its carriers into (adj, mu) entries at
//               their final arena position (ascending sample order) and forms the carrier sums
//   spa3_head   per variant: ordered sums, g_pos/g_neg bound test, the Newton
//               step at t = 0, chunk descriptors
//   repeat L times
//     spa3_pass     per chunk of 4096 carriers: K1, K2, Korg partial sums of the
//                   roots still searching (SPATest.cpp:49,64,79-80)
//     spa3_advance  per variant: ordered sum of its chunks' partials, then
//                   getroot_K1_fast's step (root_feed, kern_spa2.h)
//   spa3_finish   per variant: Lugannani-Rice (SPATest.cpp:211-230), SE, row
//
// Two sweeps over the carrier arena are saved against the plain scheme:
//  * t = 0 needs no pass: exp(0) = 1 turns the K1/K2 sums into sum g mu and
//    sum g^2 mu(1-mu), which the extraction has anyway, and Korg(0) = 0.
//  * Korg rides along in the pass that is predicted to be the last of a search:
//    getroot_K1_fast returns the LAST EVALUATED point as the root
//    (SPATest.cpp:161-165,180), so its Korg is then already there; spa3_korg
//    serves the searches whose end was not predicted.
// At N = 430K Newton needs ~3 evaluations per root after t = 0, i.e. 3 passes.
// (Evaluating the first Newton point inside spa3_fill -- it is known from m1 --
// was tried and lost: that kernel is compaction-shaped, its lanes are poorly
// filled, and the exp/log work cost 1.7 ms to save a 0.6 ms sweep.)
//
// Variants that do not fit the arena, need the exact dense g_pos/g_neg pass, or
// are still iterating after L levels are appended to fallback lists and handled
// by spa2_kernel / spa_kernel afterwards -- same algorithm, same results.

#define SPA3_CHUNK 4096      /* carriers per chunk                     */
#define SPA3_BLOCK 256
#define SPA3_MLP 4           /* independent 16-byte loads in flight per thread */
#define SPA3_TAB_BYTES (128 * 1024)   /* LDS for the X rows + mu of one sample segment */
#define SPA3_FILL_WAVES 16
#define SPA3_NPART 6         /* doubles per chunk partial: K1a K2a Ka K1b K2b Kb */
#define SPA3_NSEGP 6         /* doubles per (variant, segment): the carrier sums */

struct SpaHead {
	int state;        // 0 finished / handed over, 1 searching, 2 both roots converged, 3 not converged
	int nnz;          // carriers; < 0: no list (-1 arena full -> spa2, -2 finished in spa3_plan)
	int c0, nchunks;
	int pad0_, pad1_;
	unsigned long long off;     // first list entry in the arena
	double m1, Tstat, var2, var1, qtilde, qinv, pn_in, NAmu, NAsigma;
	RootState s1, s2;
};

struct ChunkDesc { int v; int k; };

// final row of a variant that went through Saddle_Prob_Fast (saige_main.cpp:390-403)
__device__ __forceinline__ void spa_write_row(const SpaRec &r, double Tstat, double var1, double pval,
	bool converged, double *__restrict__ out8)
{
	if (pval == 0 && r.p_noadj > 0) { pval = r.p_noadj; converged = false; }
	double beta = (Tstat / var1) / sqrt(r.AC2);
	if (r.minus) beta = -beta;
	double *o = out8 + (size_t)r.j * 8;
	o[3] = beta;
	o[4] = fabs(beta / d_qnorm(pval / 2));
	o[5] = pval;
	o[7] = converged ? 1.0 : 0.0;
}

// K1, K2 (SPATest.cpp:64,79-80) and Korg (:49) terms of one carrier at t
template <bool WITH_K>
__device__ __forceinline__ void cgf_terms(double g, double m, double t, double &k1, double &k2, double &k0)
{
	const double om = 1 - m, mg = m * g, c2 = om * mg * g;
	const double e = fast_exp(-g * t);
	const double d = fma(om, e, m);
	const double rr = isfinite(d) ? fast_rcp(d) : 0.0;
	k1 = fma(mg, rr, k1);
	const double tt = c2 * e * rr * rr;
	if (isfinite(tt)) k2 += tt;
	// log(1 - m + m e^{gt}) = g t + log((1-m) e^{-gt} + m): reuses the exponential;
	// when e^{-gt} overflows the reference's own form is evaluated instead
	if (WITH_K) k0 += isfinite(d) ? fma(g, t, fast_log(d)) : fast_log(fma(m, fast_exp(g * t), om));
}

// counters: [0] n_spa [1] n_valid [2] n_dense_fallback [3] n_spa2_fallback
//           [4] chunk cursor [6] number of valid chunk descriptors

// samples per extraction segment for K covariates: rows of (K + 2) & ~1 doubles, a power of two
// of them in SPA3_TAB_BYTES, at most 4096
__host__ __device__ constexpr int spa3_seg(int K)
{
	const int row = ((K + 2) & ~1) * 8;
	return row <= 32 ? 4096 : row <= 64 ? 2048 : row <= 128 ? 1024 : 512;
}

// one wave per (variant, segment)
__global__ void __launch_bounds__(256)
spa3_count(const uint8_t *__restrict__ packed, size_t bpv, int N, int nseg, int seg_samples,
	const SpaRec *__restrict__ recs, const int *__restrict__ counters, int *__restrict__ segcnt)
{
	const int lane = threadIdx.x & (WAVE - 1);
	const int gw = (blockIdx.x * blockDim.x + threadIdx.x) / WAVE, nw = gridDim.x * blockDim.x / WAVE;
	const int nflag = counters[0];
	const int nitem = nflag * nseg;
	const int ndw = (N + 15) >> 4, segdw = seg_samples / 16;
	for (int wi = gw; wi < nitem; wi += nw) {
		const int seg = wi / nflag, v = wi - seg * nflag;   // segment-major order
		const uint32_t *row = reinterpret_cast<const uint32_t *>(packed + (size_t)recs[v].j * bpv);
		const uint32_t zx = recs[v].minus ? 0xAAAAAAAAu : 0u;
		int cnt = 0;
		for (int dd = lane; dd < segdw; dd += WAVE) {
			const int d = seg * segdw + dd;
			const uint32_t w = (d < ndw) ? row[d] : 0u;
			cnt += __popc(nz_fields((w ^ zx) & keep_mask(N - d * 16)));
		}
		cnt = wave_sum_i(cnt);
		if (lane == 0) segcnt[v * nseg + seg] = cnt;
	}
}

template <int K>
__global__ void __launch_bounds__(256)
spa3_plan(DevModel md, int nseg, const SpaRec *__restrict__ recs, int *__restrict__ counters,
	unsigned long long *__restrict__ cursor, unsigned long long arena_cap, int *__restrict__ segcnt,
	SpaHead *__restrict__ heads, int *__restrict__ fb_spa2, double *__restrict__ out8)
{
	const int v = blockIdx.x * blockDim.x + threadIdx.x;
	if (v >= counters[0]) return;
	const SpaRec r = recs[v];
	SpaHead h;
	h.state = 0; h.c0 = 0; h.nchunks = 0; h.pad0_ = h.pad1_ = 0; h.off = 0;
	h.m1 = h.qtilde = h.qinv = h.pn_in = h.NAmu = h.NAsigma = 0;
	const double inv = 1 / sqrt(r.AC2);
	h.Tstat = r.S * inv;                       // q - m1, saige_main.cpp:380
	h.var2 = r.var2 / r.AC2;                   // :378
	h.var1 = h.var2 * md.r;                    // :379
	bool done = false;
	if (r.has_gmu) {
		double xmu_c = 0;
#pragma unroll
		for (int a = 0; a < K; a++) xmu_c = fma(md.Xmu[a], r.c[a], xmu_c);
		h.m1 = (r.sum_gmu - xmu_c) * inv;
		h.qtilde = h.Tstat / sqrt(h.var1) * sqrt(h.var2) + h.m1;   // :381
		const double s = h.qtilde - h.m1;
		h.qinv = -s + h.m1;
		h.pn_in = d_pchisq1_upper(s * s / h.var2);
		if (fabs(h.qtilde - h.m1) / sqrt(h.var2) < 2.0) {
			spa_write_row(r, h.Tstat, h.var1, h.pn_in, true, out8);   // SPATest.cpp:319-321
			done = true;
		}
	}
	int run = 0;
	for (int s = 0; s < nseg; s++) { const int c = segcnt[v * nseg + s]; segcnt[v * nseg + s] = run; run += c; }
	h.nnz = run;
	if (done) {
		h.nnz = -2;
	} else {
		const unsigned long long off = atomicAdd(cursor, (unsigned long long)run);
		h.off = off;
		if (off + (unsigned long long)run > arena_cap) {
			h.nnz = -1;                                    // no list: the per-workgroup kernel takes it
			fb_spa2[atomicAdd(&counters[3], 1)] = v;
		}
	}
	heads[v] = h;
}

// grid-stride over (segment, slice of the flagged variants), segment-major; block = SPA3_FILL_WAVES
// waves; dynamic LDS = the segment's table + one 1024-entry queue per wave.
template <int K>
__global__ void __launch_bounds__(WAVE * SPA3_FILL_WAVES)
spa3_fill(const uint8_t *__restrict__ packed, size_t bpv, DevModel md, int nseg, int nslice,
	const SpaRec *__restrict__ recs, const int *__restrict__ counters, const int *__restrict__ segoff,
	const SpaHead *__restrict__ heads, double2 *__restrict__ arena, double *__restrict__ segpart)
{
	constexpr int SEG = spa3_seg(K), KP = (K + 2) & ~1;
	constexpr int SUBDW = SEG / 16 < WAVE ? SEG / 16 : WAVE, NSUB = SEG / 16 / SUBDW;   // dwords per sub-step (one per lane)
	constexpr int FMLP = K <= 4 ? 4 : (K <= 8 ? 2 : 1);   // carriers per lane in flight (registers: K + 1 doubles each)
	extern __shared__ __attribute__((aligned(16))) uint8_t fill_smem[];
	double *tab = reinterpret_cast<double *>(fill_smem);                               // [SEG][KP]
	uint16_t *queues = reinterpret_cast<uint16_t *>(fill_smem + (size_t)SEG * KP * 8);    // [waves][1024]
	const int N = md.N, tid = threadIdx.x, lane = tid & (WAVE - 1), wid = tid / WAVE;
	uint16_t *q = queues + wid * 1024;       // sample within the segment (12 bits) | code << 14
	const int nflag = counters[0];
	const int vper = (nflag + nslice - 1) / nslice;
	const int ndw = (N + 15) >> 4;
	for (int item = blockIdx.x; item < nseg * nslice; item += gridDim.x) {
		const int seg = item / nslice, sl = item - seg * nslice;
		const int vbeg = sl * vper, vend = min(nflag, vbeg + vper);
		if (vbeg >= vend) continue;
		__syncthreads();                     // the previous item's readers of the table are done
		{
			const int rows = min(SEG, N - seg * SEG);
			const double2 *src = reinterpret_cast<const double2 *>(md.XM + (size_t)seg * SEG * KP);
			double2 *dst = reinterpret_cast<double2 *>(tab);
			for (int i = tid; i < rows * (KP / 2); i += WAVE * SPA3_FILL_WAVES) dst[i] = src[i];
		}
		__syncthreads();
		// this lane's dwords of a variant's segment (one per sub-step), fetched one variant ahead
		auto load_row = [&](int v, uint32_t (&wv)[NSUB]) {
			const uint32_t *row = reinterpret_cast<const uint32_t *>(packed + (size_t)recs[v].j * bpv);
#pragma unroll
			for (int sub = 0; sub < NSUB; sub++) {
				const int d = seg * (SEG / 16) + sub * SUBDW + lane;
				wv[sub] = (lane < SUBDW && d < ndw) ? row[d] : 0u;
			}
		};
		uint32_t wcur[NSUB], wnxt[NSUB];
		if (vbeg + wid < vend) load_row(vbeg + wid, wcur);
		for (int v = vbeg + wid; v < vend; v += SPA3_FILL_WAVES) {
			if (v + SPA3_FILL_WAVES < vend) load_row(v + SPA3_FILL_WAVES, wnxt);
			uint32_t wrow[NSUB];
#pragma unroll
			for (int sub = 0; sub < NSUB; sub++) { wrow[sub] = wcur[sub]; wcur[sub] = wnxt[sub]; }
			const SpaHead *hd = heads + v;
			if (hd->nnz < 0) continue;
			const SpaRec r = recs[v];
			const int it = v * nseg + seg;
			const double inv = 1 / sqrt(r.AC2);
			const uint32_t zx = r.minus ? 0xAAAAAAAAu : 0u;
			double c[K];
#pragma unroll
			for (int a = 0; a < K; a++) c[a] = r.c[a];
			double2 *lst = arena + hd->off + (unsigned long long)segoff[it];
			// 0 sum mu*G, 1 sum b, 2 sum max(adj,0), 3 sum min(adj,0), 4 sum adj*mu, 5 sum adj^2 mu(1-mu)
			double a12[SPA3_NSEGP];
#pragma unroll
			for (int a = 0; a < SPA3_NSEGP; a++) a12[a] = 0;
			int run = 0;
#pragma unroll
			for (int sub = 0; sub < NSUB; sub++) {
				// 16 SUBDW samples: one dword per lane -> carriers, compacted through the wave's queue
				const int d = seg * (SEG / 16) + sub * SUBDW + lane;
				const bool mine = lane < SUBDW && d < ndw;       // (a flipped variant turns an absent dword into carriers)
				const uint32_t w = wrow[sub];
				uint32_t z = mine ? nz_fields((w ^ zx) & keep_mask(N - d * 16)) : 0u;
				const int cnt = __popc(z);
				int incl = cnt;
#pragma unroll
				for (int o = 1; o < WAVE; o <<= 1) {
					const int up = __shfl_up(incl, o, WAVE);
					if (lane >= o) incl += up;
				}
				const int total = __shfl(incl, WAVE - 1, WAVE);
				int o2 = incl - cnt;
				while (z) {
					const int b = __ffs(z) - 1;
					z &= z - 1;
					q[o2++] = (uint16_t)((sub * SUBDW + lane) * 16 + (b >> 1)) | (uint16_t)(((w >> b) & 3u) << 14);
				}
				__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // the wave's own LDS writes, in order
				__builtin_amdgcn_wave_barrier();
				for (int k0 = lane; k0 < total; k0 += FMLP * WAVE) {
					// FMLP carriers per lane at a time: their queue and table reads are in flight together
					uint32_t e[FMLP];
#pragma unroll
					for (int j = 0; j < FMLP; j++) e[j] = (k0 + j * WAVE < total) ? q[k0 + j * WAVE] : 0u;
					double xr[FMLP][K + 1];
#pragma unroll
					for (int j = 0; j < FMLP; j++) {
						const double *x = tab + (size_t)(e[j] & 0x3FFFu) * KP;
#pragma unroll
						for (int a = 0; a <= K; a++) xr[j][a] = x[a];
					}
#pragma unroll
					for (int j = 0; j < FMLP; j++) {
						const int k = k0 + j * WAVE;
						if (k >= total) continue;
						const double G = sel4(r.lut, e[j] >> 14);
						double b = 0;
#pragma unroll
						for (int a = 0; a < K; a++) b = fma(xr[j][a], c[a], b);
						const double mui = xr[j][K];
						const double adj = (G - b) * inv;
						lst[run + k] = make_double2(adj, mui);
						a12[0] = fma(mui, G, a12[0]);
						a12[1] += b;
						if (adj > 0) a12[2] += adj; else a12[3] += adj;
						a12[4] = fma(adj, mui, a12[4]);
						a12[5] = fma(adj * adj, mui * (1 - mui), a12[5]);
					}
				}
				__builtin_amdgcn_wave_barrier();   // queue reads above stay before the next sub-step's writes
				run += total;
			}
#pragma unroll
			for (int a = 0; a < SPA3_NSEGP; a++) a12[a] = wave_sum(a12[a]);
			if (lane == 0) {
#pragma unroll
				for (int a = 0; a < SPA3_NSEGP; a++) segpart[(size_t)it * SPA3_NSEGP + a] = a12[a];
			}
		}
	}
}

template <int K>
__global__ void __launch_bounds__(256)
spa3_head(DevModel md, int nseg, const SpaRec *__restrict__ recs, int *__restrict__ counters,
	const double *__restrict__ segpart, SpaHead *__restrict__ heads, ChunkDesc *__restrict__ chunks,
	int chunk_cap, int *__restrict__ fb_dense, int *__restrict__ fb_spa2, double *__restrict__ out8,
	int force_dense)
{
	const int v = blockIdx.x * blockDim.x + threadIdx.x;
	if (v >= counters[0]) return;
	SpaHead h = heads[v];
	if (h.nnz < 0) { heads[v].nnz = 0; return; }
	const SpaRec r = recs[v];
	double a[SPA3_NSEGP];
#pragma unroll
	for (int q = 0; q < SPA3_NSEGP; q++) a[q] = 0;
	for (int s = 0; s < nseg; s++)
#pragma unroll
		for (int q = 0; q < SPA3_NSEGP; q++) a[q] += segpart[((size_t)v * nseg + s) * SPA3_NSEGP + q];
	const double inv = 1 / sqrt(r.AC2);
	double xmu_c = 0, xsum_c = 0;
#pragma unroll
	for (int q = 0; q < K; q++) { xmu_c = fma(md.Xmu[q], r.c[q], xmu_c); xsum_c = fma(md.Xsum[q], r.c[q], xsum_c); }
	const int nnz = h.nnz;
	if (!r.has_gmu) {
		// gather score path: m1 only now (carrier sum of mu*G), saige_main.cpp:378-381
		h.m1 = (a[0] - xmu_c) * inv;
		h.qtilde = h.Tstat / sqrt(h.var1) * sqrt(h.var2) + h.m1;
		const double s = h.qtilde - h.m1;
		h.qinv = -s + h.m1;
		h.pn_in = d_pchisq1_upper(s * s / h.var2);
	}
	h.NAmu = h.m1 - a[4];
	h.NAsigma = h.var2 - a[5];
	h.state = 0;
	if (fabs(h.qtilde - h.m1) / sqrt(h.var2) < 2.0) {
		spa_write_row(r, h.Tstat, h.var1, h.pn_in, true, out8);   // SPATest.cpp:319-321
	} else {
		const double nb = (xsum_c - a[1]) * inv;
		const double L = a[2] + fmax(-nb, 0.0), U = a[3] + fmin(-nb, 0.0);
		const double mar = 1e-9 * (fabs(L) + fabs(U) + fabs(h.qtilde) + fabs(h.qinv));
		if (force_dense || !(h.qtilde < L - mar && h.qtilde > U + mar && h.qinv < L - mar && h.qinv > U + mar)) {
			fb_dense[atomicAdd(&counters[2], 1)] = v;
		} else {
			const int nch = (nnz + SPA3_CHUNK - 1) / SPA3_CHUNK;
			const int c0 = atomicAdd(&counters[4], nch);
			if (c0 < 0 || c0 + nch > chunk_cap) {
				// table full: descriptors [c0, ..) stay unwritten and outside counters[6]
				fb_spa2[atomicAdd(&counters[3], 1)] = v;
			} else {
				atomicMax(&counters[6], c0 + nch);     // chunks the pass kernels may touch
				h.c0 = c0; h.nchunks = nch;
				root_begin(h.s1, h.qtilde, L, U);
				root_begin(h.s2, h.qinv, L, U);
				// t = 0 needs no pass: exp(0) = 1 turns the K1 and K2 sums into
				// sum g mu and sum g^2 mu (1-mu), and Korg(0) = 0
				root_feed(h.s1, a[4], a[5], h.NAmu, h.NAsigma, 0.0, true);
				root_feed(h.s2, a[4], a[5], h.NAmu, h.NAsigma, 0.0, true);
				h.state = (h.s1.active || h.s2.active) ? 1 : ((h.s1.converged && h.s2.converged) ? 2 : 3);
				for (int k = 0; k < nch; k++) { chunks[c0 + k].v = v; chunks[c0 + k].k = k; }
			}
		}
	}
	heads[v] = h;
}

// one workgroup per chunk; partial[chunk] = {K1, K2, Korg at t1, K1, K2, Korg at t2}
__global__ void __launch_bounds__(SPA3_BLOCK)
spa3_pass(const int *__restrict__ counters, const ChunkDesc *__restrict__ chunks,
	const SpaHead *__restrict__ heads, const double2 *__restrict__ arena, double *__restrict__ partial)
{
	__shared__ double sh[SPA3_NPART * (SPA3_BLOCK / WAVE)];
	const int nchunk = counters[6];
	for (int ci = blockIdx.x; ci < nchunk; ci += gridDim.x) {
		const ChunkDesc cd = chunks[ci];
		const SpaHead *h = heads + cd.v;
		if (h->state != 1) continue;
		const bool a1 = h->s1.active, a2 = h->s2.active;
		const bool k1w = h->s1.want_k, k2w = h->s2.want_k;
		const double t1 = h->s1.tnew, t2 = h->s2.tnew;
		const int beg = cd.k * SPA3_CHUNK, end = min(h->nnz, beg + SPA3_CHUNK);
		const double2 *lst = arena + h->off;
		double v[SPA3_NPART] = {0, 0, 0, 0, 0, 0};
		for (int k0 = beg + threadIdx.x; k0 < end; k0 += SPA3_MLP * SPA3_BLOCK) {
			double2 gm[SPA3_MLP];
#pragma unroll
			for (int j = 0; j < SPA3_MLP; j++) {   // independent loads in flight
				const int k = k0 + j * SPA3_BLOCK;
				gm[j] = (k < end) ? lst[k] : make_double2(0.0, 0.5);   // g = 0 adds exactly 0
			}
#pragma unroll
			for (int j = 0; j < SPA3_MLP; j++) {
				if (a1) { if (k1w) cgf_terms<true>(gm[j].x, gm[j].y, t1, v[0], v[1], v[2]); else cgf_terms<false>(gm[j].x, gm[j].y, t1, v[0], v[1], v[2]); }
				if (a2) { if (k2w) cgf_terms<true>(gm[j].x, gm[j].y, t2, v[3], v[4], v[5]); else cgf_terms<false>(gm[j].x, gm[j].y, t2, v[3], v[4], v[5]); }
			}
		}
		block_sum<SPA3_NPART, SPA3_BLOCK>(v, sh);
		if (threadIdx.x < SPA3_NPART) partial[(size_t)ci * SPA3_NPART + threadIdx.x] = v[threadIdx.x];
	}
}

// one thread per flagged variant: consume the partial sums, take the Newton step
__global__ void __launch_bounds__(256)
spa3_advance(int *__restrict__ counters, SpaHead *__restrict__ heads, const double *__restrict__ partial)
{
	const int v = blockIdx.x * blockDim.x + threadIdx.x;
	if (v >= counters[0]) return;
	SpaHead *h = heads + v;
	if (h->state != 1) return;
	double s[SPA3_NPART] = {0, 0, 0, 0, 0, 0};
	for (int k = 0; k < h->nchunks; k++)
#pragma unroll
		for (int q = 0; q < SPA3_NPART; q++) s[q] += partial[(size_t)(h->c0 + k) * SPA3_NPART + q];
	RootState s1 = h->s1, s2 = h->s2;
	if (s1.active) { const bool kv = s1.want_k; root_feed(s1, s[0], s[1], h->NAmu, h->NAsigma, s[2], kv); }
	if (s2.active) { const bool kv = s2.want_k; root_feed(s2, s[3], s[4], h->NAmu, h->NAsigma, s[5], kv); }
	h->s1 = s1; h->s2 = s2;
	if (!s1.active && !s2.active) h->state = (s1.converged && s2.converged) ? 2 : 3;
}

// Korg at the root for the searches that ended at a point evaluated without it
// (the guess in root_step was wrong); partial[chunk] = {.., .., Ka, .., .., Kb}
__global__ void __launch_bounds__(SPA3_BLOCK)
spa3_korg(const int *__restrict__ counters, const ChunkDesc *__restrict__ chunks,
	const SpaHead *__restrict__ heads, const double2 *__restrict__ arena, double *__restrict__ partial)
{
	__shared__ double sh[2 * (SPA3_BLOCK / WAVE)];
	const int nchunk = counters[6];
	for (int ci = blockIdx.x; ci < nchunk; ci += gridDim.x) {
		const ChunkDesc cd = chunks[ci];
		const SpaHead *h = heads + cd.v;
		if (h->state != 2 || (h->s1.k_ok && h->s2.k_ok)) continue;
		const bool n1 = !h->s1.k_ok, n2 = !h->s2.k_ok;
		const double t1 = h->s1.root, t2 = h->s2.root;
		const int beg = cd.k * SPA3_CHUNK, end = min(h->nnz, beg + SPA3_CHUNK);
		const double2 *lst = arena + h->off;
		double v[2] = {0, 0};
		for (int k = beg + threadIdx.x; k < end; k += SPA3_BLOCK) {
			const double2 gm = lst[k];
			const double g = gm.x, m = gm.y, om = 1 - m;
			if (n1) v[0] += fast_log(fma(m, fast_exp(g * t1), om));
			if (n2) v[1] += fast_log(fma(m, fast_exp(g * t2), om));
		}
		block_sum<2, SPA3_BLOCK>(v, sh);
		if (threadIdx.x == 0) { partial[(size_t)ci * SPA3_NPART + 2] = v[0]; partial[(size_t)ci * SPA3_NPART + 5] = v[1]; }
	}
}

// one thread per flagged variant: tail probabilities and the output row;
// variants still searching go to the spa2 fallback list
__global__ void __launch_bounds__(256)
spa3_finish(int *__restrict__ counters, SpaHead *__restrict__ heads, const double *__restrict__ partial,
	const SpaRec *__restrict__ recs, int *__restrict__ fb_spa2, double *__restrict__ out8)
{
	const int v = blockIdx.x * blockDim.x + threadIdx.x;
	if (v >= counters[0]) return;
	SpaHead *h = heads + v;
	if (h->state == 0) return;
	if (h->state == 1) { fb_spa2[atomicAdd(&counters[3], 1)] = v; h->state = 0; return; }
	const SpaRec r = recs[v];
	double pval;
	bool converged = true;
	if (h->state == 2) {
		double ka = h->s1.Kcur, kb = h->s2.Kcur;
		if (!h->s1.k_ok || !h->s2.k_ok) {      // sums of spa3_korg
			double sa = 0, sb = 0;
			for (int k = 0; k < h->nchunks; k++) {
				sa += partial[(size_t)(h->c0 + k) * SPA3_NPART + 2];
				sb += partial[(size_t)(h->c0 + k) * SPA3_NPART + 5];
			}
			if (!h->s1.k_ok) ka = sa;
			if (!h->s2.k_ok) kb = sb;
		}
		const double p1 = lugannani_rice(h->s1.root, ka, h->s1.K2cur, h->qtilde, h->NAmu, h->NAsigma);
		const double p2 = lugannani_rice(h->s2.root, kb, h->s2.K2cur, h->qinv, h->NAmu, h->NAsigma);
		pval = fabs(p1) + fabs(p2);
		if (pval != 0 && h->pn_in / pval > 1000) pval = h->pn_in;   // SPATest.cpp:368-371
	} else {
		pval = h->pn_in;
		converged = false;
	}
	spa_write_row(r, h->Tstat, h->var1, pval, converged, out8);
	h->state = 0;
}


// ---------------------------------------------------------------------------
// Dosage rows (RAW bytes / doubles) through the same level-synchronous stage: the list of a row
// holds its non-zero (imputed, flipped) dosages -- for imputed data nearly every sample.
// spa3_count_ds / spa3_fill_ds stand in for spa3_count / spa3_fill; plan, head, the Newton levels
// and finish are shared.  INPUT is IN_U8 or IN_F64 (kern_spa.h).

template <int INPUT>
__global__ void __launch_bounds__(256)
spa3_count_ds(const void *__restrict__ rows, size_t row_bytes, int N, int nseg, int seg_samples,
	const SpaRec *__restrict__ recs, const int *__restrict__ counters, int *__restrict__ segcnt)
{
	const int lane = threadIdx.x & (WAVE - 1);
	const int gw = (blockIdx.x * blockDim.x + threadIdx.x) / WAVE, nw = gridDim.x * blockDim.x / WAVE;
	const int nflag = counters[0];
	const int nitem = nflag * nseg;
	for (int wi = gw; wi < nitem; wi += nw) {
		const int seg = wi / nflag, v = wi - seg * nflag;   // segment-major order
		const SpaRec &r = recs[v];
		const void *row = reinterpret_cast<const uint8_t *>(rows) + (size_t)r.j * row_bytes;
		const int s0 = seg * seg_samples, s1 = min(N, s0 + seg_samples);
		int cnt = 0;
		for (int i = s0 + lane; i < s1; i += WAVE) cnt += (load_dosage<INPUT>(row, i, r) != 0);
		cnt = wave_sum_i(cnt);
		if (lane == 0) segcnt[v * nseg + seg] = cnt;
	}
}

template <int K, int INPUT>
__global__ void __launch_bounds__(WAVE * SPA3_FILL_WAVES)
spa3_fill_ds(const void *__restrict__ rows, size_t row_bytes, DevModel md, int nseg, int nslice,
	const SpaRec *__restrict__ recs, const int *__restrict__ counters, const int *__restrict__ segoff,
	const SpaHead *__restrict__ heads, double2 *__restrict__ arena, double *__restrict__ segpart)
{
	constexpr int SEG = spa3_seg(K), KP = (K + 2) & ~1, PF = 8;   // PF dosages per lane in flight
	extern __shared__ __attribute__((aligned(16))) uint8_t fill_smem[];
	double *tab = reinterpret_cast<double *>(fill_smem);                               // [SEG][KP]
	const int N = md.N, tid = threadIdx.x, lane = tid & (WAVE - 1), wid = tid / WAVE;
	const int nflag = counters[0];
	const int vper = (nflag + nslice - 1) / nslice;
	for (int item = blockIdx.x; item < nseg * nslice; item += gridDim.x) {
		const int seg = item / nslice, sl = item - seg * nslice;
		const int vbeg = sl * vper, vend = min(nflag, vbeg + vper);
		if (vbeg >= vend) continue;
		__syncthreads();                     // the previous item's readers of the table are done
		const int rows_here = min(SEG, N - seg * SEG);
		{
			const double2 *src = reinterpret_cast<const double2 *>(md.XM + (size_t)seg * SEG * KP);
			double2 *dst = reinterpret_cast<double2 *>(tab);
			for (int i = tid; i < rows_here * (KP / 2); i += WAVE * SPA3_FILL_WAVES) dst[i] = src[i];
		}
		__syncthreads();
		for (int v = vbeg + wid; v < vend; v += SPA3_FILL_WAVES) {
			const SpaHead *hd = heads + v;
			if (hd->nnz < 0) continue;
			const SpaRec r = recs[v];
			const int it = v * nseg + seg;
			const void *row = reinterpret_cast<const uint8_t *>(rows) + (size_t)r.j * row_bytes;
			const double inv = 1 / sqrt(r.AC2);
			double c[K];
#pragma unroll
			for (int a = 0; a < K; a++) c[a] = r.c[a];
			double2 *lst = arena + hd->off + (unsigned long long)segoff[it];
			double a12[SPA3_NSEGP];
#pragma unroll
			for (int a = 0; a < SPA3_NSEGP; a++) a12[a] = 0;
			int run = 0;
			for (int b0 = 0; b0 < rows_here; b0 += PF * WAVE) {
				double g[PF];
#pragma unroll
				for (int j = 0; j < PF; j++) {
					const int ii = b0 + j * WAVE + lane;
					g[j] = (ii < rows_here) ? load_dosage<INPUT>(row, seg * SEG + ii, r) : 0.0;
				}
#pragma unroll
				for (int j = 0; j < PF; j++) {
					const int ii = b0 + j * WAVE + lane;
					const bool carrier = g[j] != 0;
					const unsigned long long mask = __ballot(carrier);
					if (carrier) {
						const int pos = run + __popcll(mask & ((1ull << lane) - 1ull));
						const double *x = tab + (size_t)ii * KP;
						double b = 0;
#pragma unroll
						for (int a = 0; a < K; a++) b = fma(x[a], c[a], b);
						const double mui = x[K];
						const double adj = (g[j] - b) * inv;
						lst[pos] = make_double2(adj, mui);
						a12[0] = fma(mui, g[j], a12[0]);
						a12[1] += b;
						if (adj > 0) a12[2] += adj; else a12[3] += adj;
						a12[4] = fma(adj, mui, a12[4]);
						a12[5] = fma(adj * adj, mui * (1 - mui), a12[5]);
					}
					run += __popcll(mask);
				}
			}
#pragma unroll
			for (int a = 0; a < SPA3_NSEGP; a++) a12[a] = wave_sum(a12[a]);
			if (lane == 0) {
#pragma unroll
				for (int a = 0; a < SPA3_NSEGP; a++) segpart[(size_t)it * SPA3_NSEGP + a] = a12[a];
			}
		}
	}
}
