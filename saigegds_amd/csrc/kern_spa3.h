// kern_spa3.h -- SPA stage v3 for 2-bit genotypes: level-synchronous Newton.
// Part of libsaigehip.so (single translation unit: saigehip.hip).
#pragma once

// The per-variant cost of the saddlepoint stage is proportional to the number
// of carriers (10 ... 320 000 at N = 430K), so "one workgroup per variant"
// (kern_spa2.h) leaves the chip waiting for the few largest variants.  Here the
// Newton iteration of ALL flagged variants advances in lock step and each
// level is one balanced launch over fixed-size chunks of the carrier lists:
//
//   spa3_count/plan/fill/head  carrier lists (adj, mu) of all flagged variants
//                 appended to one global arena; scalars; cutoff exit; bound
//                 test for g_pos/g_neg (kern_spa2.h); chunk descriptors
//   repeat L times
//     spa3_pass     one workgroup per chunk: partial K1/K2 sums of the roots
//                   that are still searching (SPATest.cpp:64,79-80)
//     spa3_advance  one thread per variant: ordered sum of its chunks' partials
//                   (deterministic), then getroot_K1_fast's step (root_feed)
//   spa3_korg     per chunk: sum log(1-mu+mu e^{gt}) at both roots (SPATest.cpp:49)
//   spa3_finish   per variant: Lugannani-Rice, SE, output row
//
// Variants that do not fit the arena, need the exact dense g_pos/g_neg pass, or
// are still iterating after L levels are appended to fallback lists and handled
// by spa2_kernel / spa_kernel afterwards -- same algorithm, same results.

#define SPA3_CHUNK 4096      /* carriers per chunk                     */
#define SPA3_BLOCK 256

struct SpaHead {
	int state;        // 0 finished / handed over, 1 searching, 2 both roots converged, 3 not converged
	int nnz, c0, nchunks;
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

// counters: [0] n_spa [1] n_valid [2] n_dense_fallback [3] n_spa2_fallback
//           [4] chunk cursor [6] number of valid chunk descriptors
//
// Extraction is split so that every launch is balanced over (variant, segment)
// work items, a segment being SPA3_SEG consecutive samples of one flagged row:
//   spa3_count  carriers per (variant, segment)
//   spa3_plan   per variant: exclusive scan over its segments, arena allocation
//   spa3_fill   per (variant, segment): gather X_i, mu_i -> (adj, mu) entries at
//               their final position (ascending sample order), partial sums
//   spa3_head   per variant: ordered sum of the partials, scalars, cutoff exit,
//               g_pos/g_neg bound test, root_begin, chunk descriptors

#define SPA3_SEG 8192        /* samples per extraction segment = 512 dwords */

__global__ void __launch_bounds__(256)
spa3_count(const uint8_t *__restrict__ packed, size_t bpv, int N, int nseg,
	const SpaRec *__restrict__ recs, const int *__restrict__ counters, int *__restrict__ segcnt)
{
	__shared__ int shi[4];
	const int tid = threadIdx.x, lane = tid & (WAVE - 1), wid = tid / WAVE;
	const int nitem = counters[0] * nseg;
	const int ndw = (N + 15) >> 4;
	const int nflag = counters[0];
	for (int wi = blockIdx.x; wi < nitem; wi += gridDim.x) {
		const int seg = wi / nflag, v = wi - seg * nflag;   // segment-major order
		const int it = v * nseg + seg;
		const int minus = recs[v].minus;
		const uint32_t *row = reinterpret_cast<const uint32_t *>(packed + (size_t)recs[v].j * bpv);
		const uint32_t zx = minus ? 0xAAAAAAAAu : 0u;
		int cnt = 0;
#pragma unroll
		for (int u = 0; u < 2; u++) {
			const int d = seg * (SPA3_SEG / 16) + 2 * tid + u;
			const uint32_t w = (d < ndw) ? row[d] : 0u;
			cnt += __popc(nz_fields((w ^ zx) & keep_mask(N - d * 16)));
		}
		cnt = wave_sum_i(cnt);
		__syncthreads();
		if (lane == 0) shi[wid] = cnt;
		__syncthreads();
		if (tid == 0) segcnt[it] = shi[0] + shi[1] + shi[2] + shi[3];
	}
}

__global__ void __launch_bounds__(256)
spa3_plan(int nseg, const SpaRec *__restrict__ recs, int *__restrict__ counters,
	unsigned long long *__restrict__ cursor, unsigned long long arena_cap, int *__restrict__ segcnt,
	SpaHead *__restrict__ heads, int *__restrict__ fb_spa2)
{
	const int v = blockIdx.x * blockDim.x + threadIdx.x;
	if (v >= counters[0]) return;
	int run = 0;
	for (int s = 0; s < nseg; s++) { const int c = segcnt[v * nseg + s]; segcnt[v * nseg + s] = run; run += c; }
	SpaHead *h = heads + v;
	h->nnz = run; h->c0 = 0; h->nchunks = 0; h->state = 0;
	const unsigned long long off = atomicAdd(cursor, (unsigned long long)run);
	h->off = off;
	if (off + (unsigned long long)run > arena_cap) {
		h->nnz = -1;                                    // no list: the per-workgroup kernel takes it
		fb_spa2[atomicAdd(&counters[3], 1)] = v;
	}
}

template <int K>
__global__ void __launch_bounds__(256)
spa3_fill(const uint8_t *__restrict__ packed, size_t bpv, DevModel md, int nseg,
	const SpaRec *__restrict__ recs, const int *__restrict__ counters, const int *__restrict__ segoff,
	const SpaHead *__restrict__ heads, double2 *__restrict__ arena, double *__restrict__ segpart)
{
	constexpr int BLOCK = 256, NW = BLOCK / WAVE;
	constexpr int KP = (K + 2) & ~1;
	__shared__ double sh[8 * NW];
	__shared__ int shi[NW];
	__shared__ uint32_t qidx[SPA3_SEG];
	const int N = md.N, tid = threadIdx.x, lane = tid & (WAVE - 1), wid = tid / WAVE;
	const int nitem = counters[0] * nseg;
	const int ndw = (N + 15) >> 4;
	const int nflag = counters[0];
	for (int wi = blockIdx.x; wi < nitem; wi += gridDim.x) {
		// segment-major: workgroups running together gather the same 8192 rows of XM (L2)
		const int seg = wi / nflag, v = wi - seg * nflag;
		const int it = v * nseg + seg;
		if (heads[v].nnz < 0) continue;
		const SpaRec r = recs[v];
		const uint32_t *row = reinterpret_cast<const uint32_t *>(packed + (size_t)r.j * bpv);
		const double inv = 1 / sqrt(r.AC2);
		const uint32_t zx = r.minus ? 0xAAAAAAAAu : 0u;
		double c[K];
#pragma unroll
		for (int a = 0; a < K; a++) c[a] = r.c[a];
		const int d0 = seg * (SPA3_SEG / 16) + 2 * tid;
		uint32_t w[2], nz[2];
		int cnt = 0;
#pragma unroll
		for (int u = 0; u < 2; u++) {
			w[u] = (d0 + u < ndw) ? row[d0 + u] : 0u;
			nz[u] = nz_fields((w[u] ^ zx) & keep_mask(N - (d0 + u) * 16));
			cnt += __popc(nz[u]);
		}
		int incl = cnt;
#pragma unroll
		for (int o = 1; o < WAVE; o <<= 1) {
			const int up = __shfl_up(incl, o, WAVE);
			if (lane >= o) incl += up;
		}
		__syncthreads();                     // previous item's readers of shi / qidx are done
		if (lane == WAVE - 1) shi[wid] = incl;
		__syncthreads();
		int wbase = 0, total = 0;
#pragma unroll
		for (int ww = 0; ww < NW; ww++) { if (ww < wid) wbase += shi[ww]; total += shi[ww]; }
		int o2 = wbase + incl - cnt;
#pragma unroll
		for (int u = 0; u < 2; u++) {
			uint32_t z = nz[u];
			while (z) {
				const int b = __ffs(z) - 1;
				z &= z - 1;
				qidx[o2++] = (uint32_t)((d0 + u) * 16 + (b >> 1)) | (((w[u] >> b) & 3u) << 30);
			}
		}
		__syncthreads();
		double2 *lst = arena + heads[v].off + (unsigned long long)segoff[it];
		double a6[6] = {0, 0, 0, 0, 0, 0};
		for (int k = tid; k < total; k += BLOCK) {
			const uint32_t e = qidx[k];
			const int i = (int)(e & 0x3FFFFFFFu);
			const double G = sel4(r.lut, e >> 30);
			const double *x = md.XM + (size_t)i * KP;
			double xv[KP];
#pragma unroll
			for (int a = 0; a < KP; a += 2) {
				const double2 t2 = *reinterpret_cast<const double2 *>(x + a);
				xv[a] = t2.x; xv[a + 1] = t2.y;
			}
			double b = 0;
#pragma unroll
			for (int a = 0; a < K; a++) b = fma(xv[a], c[a], b);
			const double mui = xv[K];
			const double adj = (G - b) * inv;
			lst[k] = make_double2(adj, mui);
			a6[0] = fma(mui, G, a6[0]);
			a6[1] += b;
			if (adj > 0) a6[2] += adj; else a6[3] += adj;
			a6[4] = fma(adj, mui, a6[4]);
			a6[5] = fma(adj * adj, mui * (1 - mui), a6[5]);
		}
		block_sum<6, BLOCK>(a6, sh);
		if (tid < 6) segpart[(size_t)it * 6 + tid] = a6[tid];
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
	double a6[6] = {0, 0, 0, 0, 0, 0};
	for (int s = 0; s < nseg; s++)
#pragma unroll
		for (int a = 0; a < 6; a++) a6[a] += segpart[((size_t)v * nseg + s) * 6 + a];
	const double inv = 1 / sqrt(r.AC2);
	double xmu_c = 0, xsum_c = 0;
#pragma unroll
	for (int a = 0; a < K; a++) { xmu_c = fma(md.Xmu[a], r.c[a], xmu_c); xsum_c = fma(md.Xsum[a], r.c[a], xsum_c); }
	const int nnz = h.nnz;
	h.m1 = (a6[0] - xmu_c) * inv;
	h.Tstat = r.S * inv;
	h.var2 = r.var2 / r.AC2;
	h.var1 = h.var2 * md.r;
	h.qtilde = h.Tstat / sqrt(h.var1) * sqrt(h.var2) + h.m1;
	const double s = h.qtilde - h.m1;
	h.qinv = -s + h.m1;
	h.pn_in = d_pchisq1_upper(s * s / h.var2);
	h.NAmu = h.m1 - a6[4];
	h.NAsigma = h.var2 - a6[5];
	h.state = 0;
	if (fabs(h.qtilde - h.m1) / sqrt(h.var2) < 2.0) {
		spa_write_row(r, h.Tstat, h.var1, h.pn_in, true, out8);   // SPATest.cpp:319-321
	} else {
		const double nb = (xsum_c - a6[1]) * inv;
		const double L = a6[2] + fmax(-nb, 0.0), U = a6[3] + fmin(-nb, 0.0);
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
				// the evaluation at t = 0 needs no pass: exp(0) = 1 turns the K1 and K2
				// sums (SPATest.cpp:64,79) into sum g mu and sum g^2 mu (1-mu)
				root_feed(h.s1, a6[4], a6[5], h.NAmu, h.NAsigma);
				root_feed(h.s2, a6[4], a6[5], h.NAmu, h.NAsigma);
				h.state = (h.s1.active || h.s2.active) ? 1 : ((h.s1.converged && h.s2.converged) ? 2 : 3);
				for (int k = 0; k < nch; k++) { chunks[c0 + k].v = v; chunks[c0 + k].k = k; }
			}
		}
	}
	heads[v] = h;
}

// one workgroup per chunk; partial[chunk] = {K1(t1), K2(t1), K1(t2), K2(t2)} over its carriers
__global__ void __launch_bounds__(SPA3_BLOCK)
spa3_pass(const int *__restrict__ counters, const ChunkDesc *__restrict__ chunks,
	const SpaHead *__restrict__ heads, const double2 *__restrict__ arena, double4 *__restrict__ partial)
{
	__shared__ double sh[4 * (SPA3_BLOCK / WAVE)];
	const int nchunk = counters[6];
	for (int ci = blockIdx.x; ci < nchunk; ci += gridDim.x) {
		const ChunkDesc cd = chunks[ci];
		const SpaHead *h = heads + cd.v;
		if (h->state != 1) continue;
		const bool a1 = h->s1.active, a2 = h->s2.active;
		const double t1 = h->s1.tnew, t2 = h->s2.tnew;
		const int beg = cd.k * SPA3_CHUNK, end = min(h->nnz, beg + SPA3_CHUNK);
		const double2 *lst = arena + h->off;
		double v[4] = {0, 0, 0, 0};
		for (int k0 = beg + threadIdx.x; k0 < end; k0 += 4 * SPA3_BLOCK) {
			double2 gm4[4];
#pragma unroll
			for (int j = 0; j < 4; j++) {          // four independent loads in flight
				const int k = k0 + j * SPA3_BLOCK;
				gm4[j] = (k < end) ? lst[k] : make_double2(0.0, 0.5);   // g = 0 adds exactly 0
			}
#pragma unroll
			for (int j = 0; j < 4; j++) {
				const double g = gm4[j].x, m = gm4[j].y, om = 1 - m;
				const double mg = m * g, c2 = om * mg * g;
				if (a1) {
					const double e = fast_exp(-g * t1);
					const double d = fma(om, e, m);
					const double rr = isfinite(d) ? fast_rcp(d) : 0.0;
					v[0] = fma(mg, rr, v[0]);
					const double tt = c2 * e * rr * rr;
					if (isfinite(tt)) v[1] += tt;
				}
				if (a2) {
					const double e = fast_exp(-g * t2);
					const double d = fma(om, e, m);
					const double rr = isfinite(d) ? fast_rcp(d) : 0.0;
					v[2] = fma(mg, rr, v[2]);
					const double tt = c2 * e * rr * rr;
					if (isfinite(tt)) v[3] += tt;
				}
			}
		}
		block_sum<4, SPA3_BLOCK>(v, sh);
		if (threadIdx.x == 0) partial[ci] = make_double4(v[0], v[1], v[2], v[3]);
	}
}

// one thread per flagged variant: consume the partial sums, take the Newton step
__global__ void __launch_bounds__(256)
spa3_advance(int *__restrict__ counters, SpaHead *__restrict__ heads, const double4 *__restrict__ partial)
{
	const int v = blockIdx.x * blockDim.x + threadIdx.x;
	if (v >= counters[0]) return;
	SpaHead *h = heads + v;
	if (h->state != 1) return;
	double s[4] = {0, 0, 0, 0};
	for (int k = 0; k < h->nchunks; k++) {
		const double4 p = partial[h->c0 + k];
		s[0] += p.x; s[1] += p.y; s[2] += p.z; s[3] += p.w;
	}
	RootState s1 = h->s1, s2 = h->s2;
	if (s1.active) root_feed(s1, s[0], s[1], h->NAmu, h->NAsigma);
	if (s2.active) root_feed(s2, s[2], s[3], h->NAmu, h->NAsigma);
	h->s1 = s1; h->s2 = s2;
	if (!s1.active && !s2.active) h->state = (s1.converged && s2.converged) ? 2 : 3;
}

// Korg partial sums at both roots for variants whose two searches converged
__global__ void __launch_bounds__(SPA3_BLOCK)
spa3_korg(const int *__restrict__ counters, const ChunkDesc *__restrict__ chunks,
	const SpaHead *__restrict__ heads, const double2 *__restrict__ arena, double4 *__restrict__ partial)
{
	__shared__ double sh[2 * (SPA3_BLOCK / WAVE)];
	const int nchunk = counters[6];
	for (int ci = blockIdx.x; ci < nchunk; ci += gridDim.x) {
		const ChunkDesc cd = chunks[ci];
		const SpaHead *h = heads + cd.v;
		if (h->state != 2) continue;
		const double t1 = h->s1.root, t2 = h->s2.root;
		const int beg = cd.k * SPA3_CHUNK, end = min(h->nnz, beg + SPA3_CHUNK);
		const double2 *lst = arena + h->off;
		double v[2] = {0, 0};
		for (int k0 = beg + threadIdx.x; k0 < end; k0 += 4 * SPA3_BLOCK) {
			double2 gm4[4];
#pragma unroll
			for (int j = 0; j < 4; j++) {
				const int k = k0 + j * SPA3_BLOCK;
				gm4[j] = (k < end) ? lst[k] : make_double2(0.0, 0.5);   // log(1) = 0
			}
#pragma unroll
			for (int j = 0; j < 4; j++) {
				const double g = gm4[j].x, m = gm4[j].y, om = 1 - m;
				v[0] += fast_log(fma(m, fast_exp(g * t1), om));
				v[1] += fast_log(fma(m, fast_exp(g * t2), om));
			}
		}
		block_sum<2, SPA3_BLOCK>(v, sh);
		if (threadIdx.x == 0) partial[ci] = make_double4(v[0], v[1], 0, 0);
	}
}

// one thread per flagged variant: tail probabilities and the output row;
// variants still searching go to the spa2 fallback list
__global__ void __launch_bounds__(256)
spa3_finish(int *__restrict__ counters, SpaHead *__restrict__ heads, const double4 *__restrict__ partial,
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
		double k1 = 0, k2 = 0;
		for (int k = 0; k < h->nchunks; k++) { const double4 p = partial[h->c0 + k]; k1 += p.x; k2 += p.y; }
		const double p1 = lugannani_rice(h->s1.root, k1, h->s1.K2cur, h->qtilde, h->NAmu, h->NAsigma);
		const double p2 = lugannani_rice(h->s2.root, k2, h->s2.K2cur, h->qinv, h->NAmu, h->NAsigma);
		pval = fabs(p1) + fabs(p2);
		if (pval != 0 && h->pn_in / pval > 1000) pval = h->pn_in;   // SPATest.cpp:368-371
	} else {
		pval = h->pn_in;
		converged = false;
	}
	spa_write_row(r, h->Tstat, h->var1, pval, converged, out8);
	h->state = 0;
}
