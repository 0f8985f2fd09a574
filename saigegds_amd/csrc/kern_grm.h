// kern_grm.h -- implicit-GRM operator of the null-model fit on 2-bit genotypes.
// Part of libsaigehip.so (single translation unit: saigehip.hip).
#pragma once

// Reference: src/saige_fitnull.cpp
//   saige_store_2b_geno   :159-230  standardised-genotype table, diag(GRM)
//   get_crossprod_b_grm   :435-536  out = G'(G b)/M over the packed genotypes
//   get_diag_sigma        :542-559, get_crossprod :564-576, PCG_diag_sigma :581-614
// G is the M x N matrix of standardised genotypes of the GRM markers,
//   G[v,i] = lut_v[code]:  lut_v = {-2af, 1-2af, 2-2af, 0(missing)} * inv_v,
//   inv_v = 1/sqrt(2 af (1-af))  (0 for a monomorphic marker, :195-197).
// With l0 = -2 af inv:  lut_v[code] = l0_v + code * inv_v for code 0..2, so both
// halves of the product are sums of integer codes times one real vector:
//   pass 1 (per marker, over samples):   dot_v = l0_v (sum b - T3b_v) + inv_v (V_v - 3 T3b_v)
//        V_v = sum_i code_vi b_i,  T3b_v = sum_{missing} b_i
//   pass 2 (per sample, over markers):   M out_i = C0 + X_i - Gam_i
//        x_v = dot_v inv_v,  C0 = sum_v dot_v l0_v,
//        X_i = sum_v code_vi x_v,  Gam_i = sum_{missing} (3 - 2 af_v) x_v
// Both are evaluated by score_mfma_kernel<1,false> (code plane + missing plane)
// on the marker-major matrix and on its 2-bit transpose, with b / (x, gam)
// converted to 56-bit fixed-point limbs first, so each pass is one streaming
// sweep of the packed matrix with exact integer accumulation.

#define GRM_NCOL 16          /* B tile columns: one value fragment               */
#define GRM_NACC 32          /* ints per row: value fragment + missing fragment  */

// ---- per-marker statistics: af, inv, l0  (:181-203); one workgroup per marker
__global__ void __launch_bounds__(256)
grm_marker_stats(const uint8_t *__restrict__ packed, size_t bpv, int N, size_t M,
	double *__restrict__ af_out, double *__restrict__ inv_out, double *__restrict__ l0_out)
{
	__shared__ int shi[8];
	const size_t v = blockIdx.x;
	if (v >= M) return;
	const int tid = threadIdx.x, lane = tid & (WAVE - 1), wid = tid / WAVE;
	const uint32_t *row = reinterpret_cast<const uint32_t *>(packed + v * bpv);
	const int ndw = (N + 15) >> 4;
	int nvalid = 0, sum = 0;
	for (int d = tid; d < ndw; d += 256) {
		const uint32_t w = row[d];
		const uint32_t km = keep_mask(N - d * 16);
		const uint32_t lo = w & LO_MASK & km, hi = (w >> 1) & LO_MASK & km;
		const int n3 = __popc(lo & hi), n1 = __popc(lo & ~hi), n2 = __popc(hi & ~lo);
		nvalid += min(16, N - d * 16) - n3;
		sum += n1 + 2 * n2;
	}
	nvalid = wave_sum_i(nvalid); sum = wave_sum_i(sum);
	if (lane == 0) { shi[wid] = nvalid; shi[4 + wid] = sum; }
	__syncthreads();
	if (tid == 0) {
		const int nv = shi[0] + shi[1] + shi[2] + shi[3], sm = shi[4] + shi[5] + shi[6] + shi[7];
		double af = double(sm) / (2 * nv);
		double inv = 1 / sqrt(2 * af * (1 - af));
		if (!isfinite(af) || !isfinite(inv)) af = inv = 0;
		af_out[v] = af; inv_out[v] = inv; l0_out[v] = (0 - 2 * af) * inv;
	}
}

// ---- 2-bit transpose: src [R][src_bpv] (C columns) -> dst [C][dst_bpv] (R columns)
// one workgroup per tile of 64 source rows x 256 source columns
__global__ void __launch_bounds__(256)
transpose_2bit(const uint8_t *__restrict__ src, size_t src_bpv, size_t R, int C,
	uint8_t *__restrict__ dst, size_t dst_bpv)
{
	__shared__ uint32_t tile[64][17];      // 64 rows x 16 dwords (+1 pad)
	const int tid = threadIdx.x;
	const size_t r0 = (size_t)blockIdx.y * 64;
	const int c0 = blockIdx.x * 256;        // first source column; 16 dwords per row
	for (int k = tid; k < 64 * 16; k += 256) {
		const int rr = k >> 4, d = k & 15;
		uint32_t w = 0;
		if (r0 + rr < R && (size_t)(c0 / 4 + d * 4 + 4) <= src_bpv)
			w = *reinterpret_cast<const uint32_t *>(src + (r0 + rr) * src_bpv + c0 / 4 + d * 4);
		tile[rr][d] = w;
	}
	__syncthreads();
	// output: 256 destination rows (source columns) x 64 destination columns = 4 dwords each
	for (int k = tid; k < 256 * 4; k += 256) {
		const int cc = k >> 2, q = k & 3;   // destination row c0+cc, its dword q (16 source rows)
		if (c0 + cc >= C) continue;
		uint32_t w = 0;
#pragma unroll
		for (int s = 0; s < 16; s++) {
			const uint32_t code = (tile[16 * q + s][cc >> 4] >> (2 * (cc & 15))) & 3u;
			w |= code << (2 * s);
		}
		*reinterpret_cast<uint32_t *>(dst + (size_t)(c0 + cc) * dst_bpv + r0 / 4 + q * 4) = w;
	}
}

// ---- diag(GRM)_i = (1/M) sum_v lut_v[code_vi]^2  (:205-227); one wave per sample
// on the sample-major matrix
__global__ void __launch_bounds__(256)
grm_diag_kernel(const uint8_t *__restrict__ gt, size_t bpvM, int N, size_t M,
	const double *__restrict__ inv, const double *__restrict__ l0, double *__restrict__ diag)
{
	const int lane = threadIdx.x & (WAVE - 1);
	const int i = blockIdx.x * (blockDim.x / WAVE) + threadIdx.x / WAVE;
	if (i >= N) return;
	const uint32_t *row = reinterpret_cast<const uint32_t *>(gt + (size_t)i * bpvM);
	const size_t ndw = (M + 15) >> 4;
	double s = 0;
	for (size_t d = lane; d < ndw; d += WAVE) {
		const uint32_t w = row[d];
		for (int k = 0; k < 16; k++) {
			const size_t v = d * 16 + k;
			if (v >= M) break;
			const uint32_t code = (w >> (2 * k)) & 3u;
			const double g = (code == 3u) ? 0.0 : fma((double)code, inv[v], l0[v]);
			s = fma(g, g, s);
		}
	}
	s = wave_sum(s);
	if (lane == 0) diag[i] = s / (double)M;
}

// ---- fixed-point conversion of a real vector into one block of 7 limb columns
__global__ void __launch_bounds__(256)
absmax_kernel(const double *__restrict__ x, size_t n, unsigned long long *__restrict__ out)
{
	double m = 0;
	for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
		const double a = fabs(x[i]);
		m = (a > m) ? a : m;      // NaN never wins
	}
#pragma unroll
	for (int o = 32; o > 0; o >>= 1) { const double t = __shfl_xor(m, o, WAVE); m = (t > m) ? t : m; }
	// non-negative doubles order like their bit patterns
	if ((threadIdx.x & (WAVE - 1)) == 0) atomicMax(out, (unsigned long long)__double_as_longlong(m));
}

// exponent e with max|x| * 2^e < 2^54
__device__ __forceinline__ int limb_scale(unsigned long long maxbits)
{
	const double mx = __longlong_as_double((long long)maxbits);
	if (!(mx > 0) || !isfinite(mx)) return 0;
	return 54 - __builtin_amdgcn_frexp_exp(mx);
}

// x[n] -> limb digits in columns [col0, col0+7) of the tile image Fl[ngrp_pad][GRM_NCOL][16];
// the whole column range is written (zeros beyond n), other columns are left alone
__global__ void __launch_bounds__(256)
limbs_kernel(const double *__restrict__ x, size_t n, size_t n_pad, int col0,
	const unsigned long long *__restrict__ maxbits, uint8_t *__restrict__ Fl)
{
	const int e = limb_scale(*maxbits);
	for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n_pad; i += (size_t)gridDim.x * blockDim.x) {
		long long q = 0;
		if (i < n) {
			const double v = x[i];
			q = isfinite(v) ? __double2ll_rn(ldexp(v, e)) : 0;
		}
		uint8_t *base = Fl + ((i >> 4) * GRM_NCOL + col0) * 16 + mf_pos((int)(i & 15));
		long long rem = q;
#pragma unroll
		for (int l = 0; l < MF_NLIMB; l++) {
			const long long d = (l < MF_NLIMB - 1) ? (((rem + 128) & 255) - 128) : rem;
			rem = (rem - d) >> 8;
			base[l * 16] = (uint8_t)(int8_t)d;
		}
	}
}

// deterministic block partial sums: out[blockIdx] = sum of a[i] (* b[i])
template <bool WITH_B>
__global__ void __launch_bounds__(256)
dot_partial_kernel(const double *__restrict__ a, const double *__restrict__ b, size_t n,
	double *__restrict__ out)
{
	__shared__ double sh[4];
	double s[1] = {0};
	for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
		s[0] = WITH_B ? fma(a[i], b[i], s[0]) : s[0] + a[i];
	block_sum<1, 256>(s, sh);
	if (threadIdx.x == 0) out[blockIdx.x] = s[0];
}

// ---- pass-1 epilogue: dot_v -> x_v, gam_v, and the C0 partial sums
__global__ void __launch_bounds__(256)
grm_dot_epilogue(size_t M, const int *__restrict__ acc, const unsigned long long *__restrict__ maxb,
	double sum_b, const double *__restrict__ af, const double *__restrict__ inv,
	const double *__restrict__ l0, double *__restrict__ xv, double *__restrict__ gv,
	double *__restrict__ c0_partial)
{
	__shared__ double sh[4];
	const int e = limb_scale(*maxb);
	double c0[1] = {0};
	for (size_t v = blockIdx.x * (size_t)blockDim.x + threadIdx.x; v < M; v += (size_t)gridDim.x * blockDim.x) {
		const int *a = acc + v * GRM_NACC;
		const HiLo V = mf_limbs(a), T3 = mf_limbs(a + GRM_NCOL);
		const double t3 = ldexp(hl_to_double(T3), -e);
		const double vw = ldexp(hl_to_double(hl_axpy(-3, T3, V)), -e);     // sum over codes 1,2 of code*b
		const double dot = l0[v] * (sum_b - t3) + inv[v] * vw;
		const double x = dot * inv[v];
		xv[v] = x;
		gv[v] = (3 - 2 * af[v]) * x;
		c0[0] = fma(dot, l0[v], c0[0]);
	}
	block_sum<1, 256>(c0, sh);
	if (threadIdx.x == 0) c0_partial[blockIdx.x] = c0[0];
}

// ---- pass-2 epilogue: out_i = (C0 + X_i - Gam_i) / M
__global__ void __launch_bounds__(256)
grm_out_epilogue(int N, size_t M, const int *__restrict__ acc, const unsigned long long *__restrict__ maxx,
	const unsigned long long *__restrict__ maxg, double C0, double *__restrict__ out)
{
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= N) return;
	const int ex = limb_scale(*maxx), eg = limb_scale(*maxg);
	const int *a = acc + (size_t)i * GRM_NACC;
	const double X = ldexp(hl_to_double(mf_limbs(a)), -ex);                       // code plane x x limbs
	const double G = ldexp(hl_to_double(mf_limbs(a + GRM_NCOL + MF_NLIMB)), -eg);  // missing plane x gam limbs
	out[i] = (C0 + X - G) / (double)M;
}

// ---- vector kernels of PCG_diag_sigma (:581-614)
// minv = 1 / max(tau0/w + tau1*diag, 1e-4)   (get_diag_sigma :542-559)
__global__ void pcg_minv_kernel(int n, const double *w, const double *diag, double tau0, double tau1, double *minv)
{
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	double v = tau0 / w[i] + tau1 * diag[i];
	if (v < 1e-4) v = 1e-4;
	minv[i] = 1 / v;
}

// r = b, z = minv*r, p = z, x = 0
__global__ void pcg_init_kernel(int n, const double *b, const double *minv, double *r, double *z, double *p, double *x)
{
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	const double ri = b[i];
	r[i] = ri; z[i] = minv[i] * ri; p[i] = z[i]; x[i] = 0;
}

// Ap = tau0 * p / w + tau1 * gp   (get_crossprod :564-576); gp may be NULL when tau1 == 0
__global__ void pcg_ap_kernel(int n, const double *p, const double *w, const double *gp, double tau0, double tau1, double *Ap)
{
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	const double base = tau0 * (p[i] * (1 / w[i]));
	Ap[i] = gp ? base + tau1 * gp[i] : base;
}

// x += a p; r -= a Ap; z = minv r
__global__ void pcg_update_kernel(int n, double a, const double *p, const double *Ap, const double *minv,
	double *x, double *r, double *z)
{
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	x[i] += a * p[i];
	const double ri = r[i] - a * Ap[i];
	r[i] = ri; z[i] = minv[i] * ri;
}

// p = z + bet p
__global__ void pcg_dir_kernel(int n, double bet, const double *z, double *p)
{
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	p[i] = z[i] + bet * p[i];
}
