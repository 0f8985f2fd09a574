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
