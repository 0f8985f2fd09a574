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
score2b_kernel(const uint8_t *__restrict__ packed, size_t bpv, int M, DevModel md,
	SpaRec *__restrict__ recs, int *__restrict__ counters, double *__restrict__ out8,
	uint8_t *__restrict__ valid)
{
	__shared__ double sh[P * (BLOCK / WAVE)];
	__shared__ int shi[3 * (BLOCK / WAVE)];
	const int j = blockIdx.x;
	if (j >= M) return;
	const int N = md.N, tid = threadIdx.x;
	const int lane = tid & (WAVE - 1), wid = tid / WAVE;
	constexpr int NW = BLOCK / WAVE;
	const uint4 *row = reinterpret_cast<const uint4 *>(packed + (size_t)j * bpv);
	const int nvec = (N + 63) >> 6;

	// ---- pass 1 ----
	int n1 = 0, n2 = 0, n3 = 0;
	for (int v = tid; v < nvec; v += BLOCK) {
		const uint4 q = row[v];
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
		return;
	}

	// ---- pass 2 ----
	double acc[P];
#pragma unroll
	for (int a = 0; a < P; a++) acc[a] = 0;
	const uint32_t zx = h.minus ? 0xAAAAAAAAu : 0u;   // xor that maps the zero-dosage code to 0
	const double *__restrict__ F = md.F;
	for (int v = tid; v < nvec; v += BLOCK) {
		const uint4 q = row[v];
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
		if (score_epilogue(md, h, acc, o, cbuf, &pn, &Ssc, &v2sc)) {
			const int slot = atomicAdd(&counters[0], 1);
			SpaRec r;
			r.j = j; r.minus = h.minus; r.AC2 = h.minus ? (2 * h.Num - h.AC) : h.AC;
			r.nnz = h.minus ? (N - n2) : (n1 + n2 + n3); r.has_gmu = 0; r.sum_gmu = 0;
			r.p_noadj = pn; r.S = Ssc; r.var2 = v2sc;
			for (int a = 0; a < 4; a++) r.lut[a] = h.lut[a];
			for (int a = 0; a < KMAX; a++) r.c[a] = (a < md.K) ? cbuf[a] : 0.0;
			recs[slot] = r;
		}
		atomicAdd(&counters[1], 1);
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
		if (score_epilogue(md, h, acc, o, cbuf, &pn, &Ssc, &v2sc)) {
			const int slot = atomicAdd(&counters[0], 1);
			SpaRec r;
			r.j = j; r.minus = h.minus; r.AC2 = h.minus ? (2 * h.Num - h.AC) : h.AC;
			r.nnz = 0; r.has_gmu = 0; r.sum_gmu = 0;
			r.p_noadj = pn; r.S = Ssc; r.var2 = v2sc;
			// dosage rows carry real values: lut[3] holds the imputed value, the
			// SPA kernel re-reads the row itself
			for (int a = 0; a < 4; a++) r.lut[a] = h.lut[a];
			for (int a = 0; a < KMAX; a++) r.c[a] = (a < md.K) ? cbuf[a] : 0.0;
			recs[slot] = r;
		}
		atomicAdd(&counters[1], 1);
	}
}
