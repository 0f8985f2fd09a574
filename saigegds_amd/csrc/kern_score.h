// kern_score.h -- score kernels (2-bit and dosage inputs)
// Part of libsaigehip.so (single translation unit: saigehip.hip).
#pragma once

// ---------------------------------------------------------------------------
// Score kernel, 2-bit input.  One workgroup per variant.
//   pass 1: popcount the codes -> AC, Num -> filter, flip, dosage table
//   pass 2: for each carrier gather F[i] and accumulate the P sums
// The row is read twice; the second read is served by L2.

template <int P, int BLOCK>
__global__ void __launch_bounds__(BLOCK)
score2b_kernel(RowsRef rr, int M, DevModel md,
	SpaRec *__restrict__ recs, int *__restrict__ counters, double *__restrict__ out8,
	uint8_t *__restrict__ valid, const int *__restrict__ list, int list_counter, int btop, int *__restrict__ fb_series, int *__restrict__ fb_exact)
{
	__shared__ double sh[P * (BLOCK / WAVE)];
	__shared__ int shi[3 * (BLOCK / WAVE)];
	// list: the variants to take (counters[list_counter] of them, grid-stride); else variant = workgroup
	const int nwork = list ? counters[list_counter] : M;
	for (int wi = blockIdx.x; wi < nwork; wi += gridDim.x) {
	const int j = list ? list[wi] : wi;
	const int N = md.N, tid = threadIdx.x;
	const int lane = tid & (WAVE - 1), wid = tid / WAVE;
	constexpr int NW = BLOCK / WAVE;
	const int nvec = (N + 63) >> 6;
	auto row_piece = [&](int v) -> uint4 { return *reinterpret_cast<const uint4 *>(rr.base + rr_piece(rr, (size_t)j, (size_t)v)); };
	__syncthreads();                     // the previous variant's readers of sh / shi are done

	// ---- pass 1 ----
	int n1 = 0, n2 = 0, n3 = 0;
	for (int v = tid; v < nvec; v += BLOCK) {
		const uint4 q = row_piece(v);
		const uint32_t ww[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
		for (int d = 0; d < 4; d++) {
			const uint32_t w = ww[d] & keep_mask(N - (v * 64 + d * 16));
			const uint32_t lo = w & LO_MASK, hi = (w >> 1) & LO_MASK;
			n3 += __popc(lo & hi);
			n1 += __popc(lo & ~hi);
			n2 += __popc(hi & ~lo);
		}
	}
	n1 = wave_sum_i(n1); n2 = wave_sum_i(n2); n3 = wave_sum_i(n3);
	if (lane == 0) { shi[wid] = n1; shi[NW + wid] = n2; shi[2 * NW + wid] = n3; }
	__syncthreads();
	n1 = n2 = n3 = 0;
#pragma unroll
	for (int w = 0; w < NW; w++) { n1 += shi[w]; n2 += shi[NW + w]; n3 += shi[2 * NW + w]; }

	const VarHead h = make_head(md, double(n1 + 2 * n2), N - n3);
	if (!h.pass) {
		if (tid == 0) { nan_row(out8 + (size_t)j * 8); valid[j] = 0; }
		continue;
	}

	// ---- pass 2 ----
	double acc[P];
#pragma unroll
	for (int a = 0; a < P; a++) acc[a] = 0;
	const uint32_t zx = h.minus ? 0xAAAAAAAAu : 0u;   // xor that maps the zero-dosage code to 0
	const double *__restrict__ F = md.F;
	for (int v = tid; v < nvec; v += BLOCK) {
		const uint4 q = row_piece(v);
		const uint32_t ww[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
		for (int d = 0; d < 4; d++) {
			const int base = v * 64 + d * 16;
			const uint32_t km = keep_mask(N - base);
			const uint32_t w = ww[d];
			uint32_t nz = nz_fields((w ^ zx) & km) ;
			while (nz) {
				const int b = __ffs(nz) - 1;
				nz &= nz - 1;
				const double g = sel4(h.lut, (w >> b) & 3u);
				const double *f = F + (size_t)(base + (b >> 1)) * P;
#pragma unroll
				for (int a = 0; a < P - 2; a += 2) {
					const double2 t = *reinterpret_cast<const double2 *>(f + a);
					acc[a] = fma(g, t.x, acc[a]);
					acc[a + 1] = fma(g, t.y, acc[a + 1]);
				}
				const double2 t = *reinterpret_cast<const double2 *>(f + P - 2);
				acc[P - 2] = fma(g, t.x, acc[P - 2]);
				acc[P - 1] = fma(g * g, t.y, acc[P - 1]);
			}
		}
	}
	block_sum<P, BLOCK>(acc, sh);

	if (tid == 0) {
		double *o = out8 + (size_t)j * 8;
		double cbuf[KMAX], pn, Ssc, v2sc;
		valid[j] = 1;
		if (score_epilogue<(P - 2) / 2>(md, h, acc, o, cbuf, &pn, &Ssc, &v2sc)) {
			spa_push<(P - 2) / 2>(md, recs, counters, btop, fb_series, fb_exact, j, h.minus, h.minus ? (2 * h.Num - h.AC) : h.AC,
				h.minus ? (N - n2) : (n1 + n2 + n3), h.lut, pn, Ssc, v2sc, cbuf);
		}
		atomicAdd(&counters[1], 1);
	}
	}
}

// ---------------------------------------------------------------------------
// Score kernel, dosage input (RAW bytes or doubles), get_ds branches
// saige_main.cpp:171-183.  One workgroup per variant, two passes like above.
// Carriers = entries with (imputed, flipped) dosage != 0.
//   T = uint8_t : 0xFF missing;  T = double : non-finite missing

template <typename T> __device__ __forceinline__ bool ds_missing(T v);
template <> __device__ __forceinline__ bool ds_missing<uint8_t>(uint8_t v) { return v == 0xFF; }
template <> __device__ __forceinline__ bool ds_missing<double>(double v) { return !isfinite(v); }

template <int P, int BLOCK, typename T>
__global__ void __launch_bounds__(BLOCK)
score_ds_kernel(const T *__restrict__ ds, int M, DevModel md, SpaRec *__restrict__ recs,
	int *__restrict__ counters, double *__restrict__ out8, uint8_t *__restrict__ valid)
{
	__shared__ double sh[P * (BLOCK / WAVE)];
	const int j = blockIdx.x;
	if (j >= M) return;
	const int N = md.N, tid = threadIdx.x;
	const T *row = ds + (size_t)j * N;

	double hd[2] = {0, 0};  // sum, count
	for (int i = tid; i < N; i += BLOCK) {
		const T v = row[i];
		if (!ds_missing<T>(v)) { hd[0] += (double)v; hd[1] += 1.0; }
	}
	block_sum<2, BLOCK>(hd, sh);
	VarHead h = make_head(md, hd[0], (int)hd[1]);
	if (!h.pass) {
		if (tid == 0) { nan_row(out8 + (size_t)j * 8); valid[j] = 0; }
		return;
	}
	const double imp = 2 * h.AF;
	double acc[P];
#pragma unroll
	for (int a = 0; a < P; a++) acc[a] = 0;
	const double *__restrict__ F = md.F;
	for (int i = tid; i < N; i += BLOCK) {
		const T v = row[i];
		double g = ds_missing<T>(v) ? imp : (double)v;
		if (h.minus) g = 2 - g;
		if (g != 0) {
			const double *f = F + (size_t)i * P;
#pragma unroll
			for (int a = 0; a < P - 2; a += 2) {
				const double2 t = *reinterpret_cast<const double2 *>(f + a);
				acc[a] = fma(g, t.x, acc[a]);
				acc[a + 1] = fma(g, t.y, acc[a + 1]);
			}
			const double2 t = *reinterpret_cast<const double2 *>(f + P - 2);
			acc[P - 2] = fma(g, t.x, acc[P - 2]);
			acc[P - 1] = fma(g * g, t.y, acc[P - 1]);
		}
	}
	block_sum<P, BLOCK>(acc, sh);
	if (tid == 0) {
		double *o = out8 + (size_t)j * 8;
		double cbuf[KMAX], pn, Ssc, v2sc;
		valid[j] = 1;
		if (score_epilogue<(P - 2) / 2>(md, h, acc, o, cbuf, &pn, &Ssc, &v2sc)) {
			const int slot = atomicAdd(&counters[0], 1);
			SpaRec r;
			r.j = j; r.minus = h.minus; r.AC2 = h.minus ? (2 * h.Num - h.AC) : h.AC;
			r.nnz = 0; r.has_gmu = 0; r.sum_gmu = 0;
			r.p_noadj = pn; r.S = Ssc; r.var2 = v2sc; r.tscale = spa_tscale(Ssc, v2sc, r.AC2, md.r);
			// dosage rows carry real values: lut[3] holds the imputed value, the
			// SPA kernel re-reads the row itself
			for (int a = 0; a < 4; a++) r.lut[a] = h.lut[a];
			
#pragma unroll
			for (int a = 0; a < KMAX; a++) r.c[a] = cbuf[a];
			recs[slot] = r;
		}
		atomicAdd(&counters[1], 1);
	}
}


// ---------------------------------------------------------------------------
// Score kernels for dosage rows, tiled: one pass, the score vectors shared through LDS.
//
// score_ds_kernel above gathers the P score values of a sample (72 B at K = 3) once per variant
// and sample and reads each row twice, i.e. ~10 B of F and 2 B of row per byte of input.  Here a
// workgroup takes 32 variants and one of `nsplit` sample ranges, walks the range in tiles whose F
// rows are staged in LDS once for all 32, and makes ONE pass by carrying, per variant and column,
//     V = sum_valid g f     U = sum_valid (2 - g) f     T3 = sum_missing f
// (g^2 resp. (2 - g)^2 for the last column), from which the epilogue takes the imputed sums in
// either allele orientation without cancellation:  V + imp T3   or   U + (2 - imp) T3.
// Partial sums go to part[split][variant][3P + 2]; score_ds_tile_epilogue adds them in split
// order (deterministic) and finishes the row.  The sample split keeps the chip busy when a call
// brings few rows (host-staged dosage blocks, burden rows).
#define DS_TILE_VB 32        /* variants per workgroup */
#define DS_TILE_LPV 8        /* lanes per variant      */

template <int P, typename T>
__global__ void __launch_bounds__(256)
score_ds_tile_kernel(const T *__restrict__ ds, int M, DevModel md, int per_split, double *__restrict__ part)
{
	constexpr int BLOCK = 256, LPV = DS_TILE_LPV, VB = DS_TILE_VB, NP = 3 * P + 2;
	constexpr int TS = ((4096 / P) & ~63) < 64 ? 64 : ((4096 / P) & ~63);   // samples per tile: <= 32 KiB of F
	__shared__ __attribute__((aligned(16))) double ftile[TS * P];
	const int N = md.N, tid = threadIdx.x;
	const int vl = tid / LPV, l = tid % LPV;
	const int j = blockIdx.x * VB + vl;
	const bool live = j < M;
	const T *row = ds + (size_t)(live ? j : 0) * N;
	const int s0 = blockIdx.y * per_split, s1 = min(N, s0 + per_split);
	double V[P], U[P], T3[P];
#pragma unroll
	for (int a = 0; a < P; a++) { V[a] = 0; U[a] = 0; T3[a] = 0; }
	double sumg = 0, nvalid = 0;
	for (int t0 = s0; t0 < s1; t0 += TS) {
		const int ts = min(TS, s1 - t0);
		__syncthreads();
		{
			const double2 *src = reinterpret_cast<const double2 *>(md.F + (size_t)t0 * P);
			double2 *dst = reinterpret_cast<double2 *>(ftile);
			for (int i = tid; i < ts * (P / 2); i += BLOCK) dst[i] = src[i];
		}
		__syncthreads();
		if (live) {
			for (int i = l; i < ts; i += LPV) {
				const T v = row[t0 + i];
				const double *f = ftile + i * P;
				if (ds_missing<T>(v)) {
#pragma unroll
					for (int a = 0; a < P; a++) T3[a] += f[a];
				} else {
					const double g = (double)v, h2 = 2 - g;
					sumg += g; nvalid += 1;
#pragma unroll
					for (int a = 0; a < P - 1; a++) { V[a] = fma(g, f[a], V[a]); U[a] = fma(h2, f[a], U[a]); }
					V[P - 1] = fma(g * g, f[P - 1], V[P - 1]);
					U[P - 1] = fma(h2 * h2, f[P - 1], U[P - 1]);
				}
			}
		}
	}
	// the LPV lanes of a variant are consecutive lanes of one wave
#pragma unroll
	for (int o = LPV / 2; o > 0; o >>= 1) {
		sumg += __shfl_xor(sumg, o, WAVE);
		nvalid += __shfl_xor(nvalid, o, WAVE);
#pragma unroll
		for (int a = 0; a < P; a++) {
			V[a] += __shfl_xor(V[a], o, WAVE);
			U[a] += __shfl_xor(U[a], o, WAVE);
			T3[a] += __shfl_xor(T3[a], o, WAVE);
		}
	}
	if (!live || l != 0) return;
	double *o = part + ((size_t)blockIdx.y * M + j) * NP;
	o[0] = sumg; o[1] = nvalid;
#pragma unroll
	for (int a = 0; a < P; a++) { o[2 + a] = V[a]; o[2 + P + a] = U[a]; o[2 + 2 * P + a] = T3[a]; }
}

template <int P>
__global__ void __launch_bounds__(256)
score_ds_tile_epilogue(int M, DevModel md, int nsplit, const double *__restrict__ part,
	SpaRec *__restrict__ recs, int *__restrict__ counters, double *__restrict__ out8, uint8_t *__restrict__ valid)
{
	constexpr int NP = 3 * P + 2;
	const int j = blockIdx.x * blockDim.x + threadIdx.x;
	if (j >= M) return;
	double t[NP];
#pragma unroll
	for (int a = 0; a < NP; a++) t[a] = 0;
	for (int s = 0; s < nsplit; s++) {
		const double *p = part + ((size_t)s * M + j) * NP;
#pragma unroll
		for (int a = 0; a < NP; a++) t[a] += p[a];
	}
	const VarHead h = make_head(md, t[0], (int)t[1]);
	double *o = out8 + (size_t)j * 8;
	if (!h.pass) { nan_row(o); valid[j] = 0; return; }
	const double imp = h.minus ? 2 - 2 * h.AF : 2 * h.AF;       // the imputed value in the tested orientation
	const double *V = t + 2, *U = t + 2 + P, *T3 = t + 2 + 2 * P;
	double acc[P];
#pragma unroll
	for (int a = 0; a < P - 1; a++) acc[a] = (h.minus ? U[a] : V[a]) + imp * T3[a];
	acc[P - 1] = (h.minus ? U[P - 1] : V[P - 1]) + imp * imp * T3[P - 1];
	double cbuf[KMAX], pn, Ssc, v2sc;
	valid[j] = 1;
	if (score_epilogue<(P - 2) / 2>(md, h, acc, o, cbuf, &pn, &Ssc, &v2sc)) {
		const int slot = atomicAdd(&counters[0], 1);
		SpaRec r;
		r.j = j; r.minus = h.minus; r.AC2 = h.minus ? (2 * h.Num - h.AC) : h.AC;
		r.nnz = 0; r.has_gmu = 0; r.sum_gmu = 0;
		r.p_noadj = pn; r.S = Ssc; r.var2 = v2sc; r.tscale = spa_tscale(Ssc, v2sc, r.AC2, md.r);
		// dosage rows carry real values: lut[3] holds the imputed value, the SPA kernel re-reads the row
		for (int a = 0; a < 4; a++) r.lut[a] = h.lut[a];
		
#pragma unroll
			for (int a = 0; a < KMAX; a++) r.c[a] = cbuf[a];
		recs[slot] = r;
	}
	atomicAdd(&counters[1], 1);
}
