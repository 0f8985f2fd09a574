// kern_burden.h -- weighted collapse of the variants of a unit into one dosage row
// (ds_mat_burden, reference src/saige_main.cpp:526-610, RAW genotype input).
// Part of libsaigehip.so (single translation unit: saigehip.hip).
#pragma once

// row r = sum over its entries e in [row_ptr[r], row_ptr[r+1]) of lut[e][code of variant var_idx[e]].
// The host puts weight, mean imputation and the minor-allele flip into the 4-entry table
//     lut = w * {0, 1, 2, m}   or   w * {2, 1, 0, 2 - m},   m = sum / n of the variant,
// so a sample's value is accumulated in the reference's order (variants in unit order) with the
// reference's own products s[j] * w.
// grid = (ceil(ndw / 256), n_rows), block = 256; one dword (16 samples) per thread.
__global__ void __launch_bounds__(256)
burden_collapse_kernel(const uint8_t *__restrict__ packed, size_t bpv, int N,
	const long long *__restrict__ row_ptr, const int *__restrict__ var_idx,
	const double *__restrict__ lut, double *__restrict__ out, size_t out_stride)
{
	const int d = blockIdx.x * blockDim.x + threadIdx.x;
	const int ndw = (N + 15) >> 4;
	if (d >= ndw) return;
	const size_t r = blockIdx.y;
	double acc[16];
#pragma unroll
	for (int s = 0; s < 16; s++) acc[s] = 0;
	for (long long e = row_ptr[r]; e < row_ptr[r + 1]; e++) {
		const uint32_t w = reinterpret_cast<const uint32_t *>(packed + (size_t)var_idx[e] * bpv)[d];
		const double l0 = lut[4 * e], l1 = lut[4 * e + 1], l2 = lut[4 * e + 2], l3 = lut[4 * e + 3];
#pragma unroll
		for (int s = 0; s < 16; s++) {
			const uint32_t c = (w >> (2 * s)) & 3u;
			acc[s] += (c == 0) ? l0 : (c == 1) ? l1 : (c == 2) ? l2 : l3;
		}
	}
	double *o = out + r * out_stride + (size_t)d * 16;
#pragma unroll
	for (int s = 0; s < 16; s++)
		if (d * 16 + s < N) o[s] = acc[s];
}

// per-variant counts of a 2-bit matrix: n_valid (codes != 3) and the allele sum (seqSetFilterCond's
// maf / missing-rate inputs, R/saige_main.r:314-321).  One workgroup per variant.
__global__ void __launch_bounds__(256)
geno_stats_kernel(const uint8_t *__restrict__ packed, size_t bpv, int N, int *__restrict__ n_valid,
	int *__restrict__ allele_sum)
{
	__shared__ int sh[2][4];
	const size_t j = blockIdx.x;
	const uint32_t *row = reinterpret_cast<const uint32_t *>(packed + j * bpv);
	const int ndw = (N + 15) >> 4;
	int nv = 0, sm = 0;
	for (int d = threadIdx.x; d < ndw; d += 256) {
		const uint32_t km = keep_mask(N - d * 16);
		const uint32_t w = row[d];
		const uint32_t lo = w & LO_MASK & km, hi = (w >> 1) & LO_MASK & km;
		const uint32_t miss = lo & hi;
		nv += __popc(km & LO_MASK) - __popc(miss);
		sm += __popc(lo & ~miss) + 2 * __popc(hi & ~miss);
	}
	nv = wave_sum_i(nv); sm = wave_sum_i(sm);
	const int lane = threadIdx.x & (WAVE - 1), wid = threadIdx.x / WAVE;
	if (lane == 0) { sh[0][wid] = nv; sh[1][wid] = sm; }
	__syncthreads();
	if (threadIdx.x == 0) {
		n_valid[j] = sh[0][0] + sh[0][1] + sh[0][2] + sh[0][3];
		allele_sum[j] = sh[1][0] + sh[1][1] + sh[1][2] + sh[1][3];
	}
}
