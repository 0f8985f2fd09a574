// dev_common.h -- device model, Rmath stand-ins, reductions, 2-bit helpers, score epilogue
// Part of libsaigehip.so (single translation unit: saigehip.hip).
#pragma once
#include "s3_layout.h"

// ---------------------------------------------------------------------------
// Rows of a block of variants as the kernels see them: row-major (dosage rows, 2-bit rows of the FP64 test
// hook) or the tiled layout of a genotype block (s3_layout.h).  Everything reads 2-bit rows in 16-byte pieces
// (64 samples).
struct RowsRef {
	const uint8_t *base;
	size_t bpv;       // row-major: bytes per row
	int ntile;        // tiled: 256-sample tiles per row; 0 = row-major
	// genotype blocks: the carrier lists of the rare variants (kern_score3.h, s3_ingest_clist_kernel); else null
	const unsigned *cptr, *cidx;
	const uint8_t *corient;
};
// byte offset of piece p of row j
__device__ __forceinline__ size_t rr_piece(const RowsRef &rr, size_t j, size_t p)
{
	return rr.ntile ? s3_piece_off(j, p, rr.ntile) : j * rr.bpv + p * 16;
}
// bytes of a 2-bit row that may be read
__device__ __forceinline__ size_t rr_row_bytes(const RowsRef &rr) { return rr.ntile ? (size_t)rr.ntile * 64 : rr.bpv; }
// start of row j of a row-major block (dosage rows)
__device__ __forceinline__ const uint8_t *rr_row(const RowsRef &rr, size_t j) { return rr.base + j * rr.bpv; }

// ---------------------------------------------------------------------------
// device-side model

struct DevModel {
	int N, K, P, quant;
	double tau0, r;
	double thr_maf, thr_mac, thr_missing, thr_spa;
	const double *F;    // [N][P] score vectors
	const double *X;    // [N][K] t_X
	const double *y;    // [N]
	const double *mu;   // [N]
	const double *mu2;  // [N]
	const double *XM;   // [N][(K+2)&~1]: X_i (K), mu_i, pad -- one gather per carrier in the SPA stage
	double XVX[KMAX * KMAX];
	double S_a[KMAX];
	double Xmu[KMAX];   // X' mu
	double Xsum[KMAX];  // X' 1
	double spa_xmax;    // series SPA stage (kern_spa4.h): largest max_i |g_i t| it accepts
	double Xabs[KMAX];  // max_i |X[i,k]|
};

// a variant handed from the score stage to the SPA stage
struct SpaRec {
	int j;            // variant index in the block
	int minus;        // AF > 0.5
	int nnz;          // carriers: samples whose (imputed, flipped) dosage is non-zero
	int has_gmu;      // (unused)
	double lut[4];    // dosage value per 2-bit code after impute + flip
	double AC2;       // allele count of the tested (minor) allele
	double p_noadj;
	double S;         // score sum (y-mu).adj, unscaled
	double var2;      // sum mu2 adj^2, unscaled (no variance ratio)
	double sum_gmu;   // sum_i G_i mu_i over the (imputed, flipped) dosages
	double tscale;    // power of two near the first Newton point of the root search (kern_spa4.h)
	double c[KMAX];   // c' = XVX_inv_XV * G
};

// 2^floor(log2 |t|) of the first Newton point t = (q~ - m1)/var2 of getroot_K1_fast, from the score
// stage's sums: q~ - m1 = Tstat/sqrt(r), Tstat = S/sqrt(AC2), var2 = var2_score/AC2
__device__ __forceinline__ double spa_tscale(double S, double var2, double AC2, double r)
{
	const double t = fabs(S) * sqrt(AC2) / (sqrt(r) * var2);
	int e = (t > 0 && isfinite(t)) ? ilogb(t) : 0;
	e = max(-400, min(400, e));
	return ldexp(1.0, e);
}

// ---------------------------------------------------------------------------
// device math: Rmath stand-ins (see oracle/saige_oracle.c for the CPU twins)

__device__ __forceinline__ double d_pchisq1_upper(double x)
{
	if (isnan(x)) return x;
	if (x <= 0) return 1.0;
	return erfc(sqrt(x * 0.5));
}

__device__ __forceinline__ double d_pnorm_upper(double z) { return 0.5 * erfc(z * M_SQRT1_2); }
__device__ __forceinline__ double d_pnorm_lower(double z) { return 0.5 * erfc(-z * M_SQRT1_2); }

__device__ __forceinline__ double d_sign(double x)
{
	if (isnan(x)) return x;
	return (x > 0) ? 1.0 : ((x == 0) ? 0.0 : -1.0);
}

// qnorm(p, 0, 1, lower, log=FALSE): Wichura AS241 PPND16, as Rmath's qnorm5
__device__ double d_qnorm(double p)
{
	if (isnan(p)) return p;
	if (p < 0 || p > 1) return NAN;
	if (p == 0) return -INFINITY;
	if (p == 1) return INFINITY;
	double q = p - 0.5, r, val;
	if (fabs(q) <= 0.425) {
		r = 0.180625 - q * q;
		val = q * (((((((r * 2509.0809287301226727 +
			33430.575583588128105) * r + 67265.770927008700853) * r +
			45921.953931549871457) * r + 13731.693765509461125) * r +
			1971.5909503065514427) * r + 133.14166789178437745) * r +
			3.387132872796366608)
			/ (((((((r * 5226.495278852854561 +
			28729.085735721942674) * r + 39307.89580009271061) * r +
			21213.794301586595867) * r + 5394.1960214247511077) * r +
			687.1870074920579083) * r + 42.313330701600911252) * r + 1.0);
		return val;
	}
	r = (q < 0) ? p : (1.0 - p);
	r = sqrt(-log(r));
	if (r <= 5.0) {
		r -= 1.6;
		val = (((((((r * 7.7454501427834140764e-4 +
			0.0227238449892691845833) * r + 0.24178072517745061177) * r +
			1.27045825245236838258) * r + 3.64784832476320460504) * r +
			5.7694972214606914055) * r + 4.6303378461565452959) * r +
			1.42343711074968357734)
			/ (((((((r * 1.05075007164441684324e-9 +
			5.475938084995344946e-4) * r + 0.0151986665636164571966) * r +
			0.14810397642748007459) * r + 0.68976733498510000455) * r +
			1.6763848301838038494) * r + 2.05319162663775882187) * r + 1.0);
	} else {
		r -= 5.0;
		val = (((((((r * 2.01033439929228813265e-7 +
			2.71155556874348757815e-5) * r + 0.0012426609473880784386) * r +
			0.026532189526576123093) * r + 0.29656057182850489123) * r +
			1.7848265399172913358) * r + 5.4637849111641143699) * r +
			6.6579046435011037772)
			/ (((((((r * 2.04426310338993978564e-15 +
			1.4215117583164458887e-7) * r + 1.8463183175100546818e-5) * r +
			7.868691311456132591e-4) * r + 0.0148753612908506148525) * r +
			0.13692988092273580531) * r + 0.59983220655588793769) * r + 1.0);
	}
	if (q < 0.0) val = -val;
	return val;
}

// ---------------------------------------------------------------------------
// fast double-precision reciprocal / exp / log of the saddlepoint sums (self-tested by sgx_selftest)

// reciprocal of d > 0 to double precision without the IEEE division sequence
__device__ __forceinline__ double fast_rcp(double d)
{
	double r = __builtin_amdgcn_rcp(d);
	r = fma(fma(-d, r, 1.0), r, r);
	r = fma(fma(-d, r, 1.0), r, r);
	return r;
}

// exp(x) in double precision: x = n ln2 + r, |r| <= ln2/2, degree-13 Taylor
// polynomial (truncation 4e-18), scaled by 2^n.  Relative error ~2e-16; overflow
// gives +inf and underflow 0 as exp() does.  About half the instructions of the
// device-library exp.
__device__ __forceinline__ double fast_exp(double x)
{
	const double n = rint(x * 1.4426950408889634074);
	double r = fma(-n, 6.93147180369123816490e-01, x);     // ln2 hi
	r = fma(-n, 1.90821492927058770002e-10, r);             // ln2 lo
	double p = 1.6059043836821613e-10;                      // 1/13!
	p = fma(p, r, 2.08767569878681e-09);
	p = fma(p, r, 2.505210838544172e-08);
	p = fma(p, r, 2.755731922398589e-07);
	p = fma(p, r, 2.7557319223985893e-06);
	p = fma(p, r, 2.48015873015873e-05);
	p = fma(p, r, 1.984126984126984e-04);
	p = fma(p, r, 1.388888888888889e-03);
	p = fma(p, r, 8.333333333333333e-03);
	p = fma(p, r, 4.1666666666666664e-02);
	p = fma(p, r, 1.6666666666666666e-01);
	p = fma(p, r, 0.5);
	p = fma(p, r, 1.0);
	p = fma(p, r, 1.0);
	double y = ldexp(p, (int)n);
	y = (x > 709.782712893384) ? INFINITY : y;
	y = (x < -745.1332191019412) ? 0.0 : y;
	return (x != x) ? x : y;
}

// log(x) for x > 0: x = 2^e m, m in [sqrt(1/2), sqrt(2)), log m = 2 atanh(s),
// s = (m-1)/(m+1), |s| <= 0.1716, series to s^21 (truncation 2e-17 relative).
__device__ __forceinline__ double fast_log(double x)
{
	if (!(x > 0) || !isfinite(x)) return log(x);             // 0, negative, inf, NaN: library semantics
	int e = __builtin_amdgcn_frexp_exp(x);
	double m = __builtin_amdgcn_frexp_mant(x);               // [0.5, 1)
	const bool lo = m < 0.70710678118654752440;
	m = lo ? m + m : m;
	e = lo ? e - 1 : e;
	const double s = (m - 1.0) * fast_rcp(m + 1.0);
	const double z = s * s;
	double p = 1.0 / 21.0;
	p = fma(p, z, 1.0 / 19.0);
	p = fma(p, z, 1.0 / 17.0);
	p = fma(p, z, 1.0 / 15.0);
	p = fma(p, z, 1.0 / 13.0);
	p = fma(p, z, 1.0 / 11.0);
	p = fma(p, z, 1.0 / 9.0);
	p = fma(p, z, 1.0 / 7.0);
	p = fma(p, z, 1.0 / 5.0);
	p = fma(p, z, 1.0 / 3.0);
	p = fma(p, z, 1.0);
	const double de = (double)e;
	return fma(de, 6.93147180369123816490e-01, fma(2.0 * s, p, de * 1.90821492927058770002e-10));
}

// ---------------------------------------------------------------------------
// wavefront / workgroup reductions (deterministic order)

__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
	for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, WAVE);
	return v;
}

__device__ __forceinline__ int wave_sum_i(int v)
{
#pragma unroll
	for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, WAVE);
	return v;
}

// Inclusive prefix sum over the 64 lanes by data-parallel-primitive moves (gfx9 DPP: shifts inside the rows of
// 16 lanes, then the two row broadcasts) -- six vector instructions and no trip through the LDS crossbar, where
// the shuffle form pays six dependent ds_bpermute latencies.  Lane 63 holds the total.
__device__ __forceinline__ int wave_scan_incl_i(int v)
{
	v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);   // row_shr:1
	v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);   // row_shr:2
	v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);   // row_shr:4
	v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);   // row_shr:8
	v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);   // row_bcast:15 -> rows 1, 3
	v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);   // row_bcast:31 -> rows 2, 3
	return v;
}
// the wave's total of v, wave-uniform (a scalar register)
__device__ __forceinline__ int wave_total_i(int v) { return __builtin_amdgcn_readlane(wave_scan_incl_i(v), WAVE - 1); }
// maximum of non-negative doubles over the wave by the same moves; lane 63 holds it (lanes a move does not
// reach see 0)
template <int CTRL, int ROWS>
__device__ __forceinline__ double dpp_f64(double v)
{
	const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROWS, 0xf, false);
	const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROWS, 0xf, false);
	return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_max_nonneg_to_last(double v)
{
	v = fmax(v, dpp_f64<0x111, 0xf>(v));
	v = fmax(v, dpp_f64<0x112, 0xf>(v));
	v = fmax(v, dpp_f64<0x114, 0xf>(v));
	v = fmax(v, dpp_f64<0x118, 0xf>(v));
	v = fmax(v, dpp_f64<0x142, 0xa>(v));
	v = fmax(v, dpp_f64<0x143, 0xc>(v));
	return v;
}

// Sum NV doubles per thread over the workgroup; every thread gets the totals.
// sh must hold NV * (BLOCK/64) doubles.  Two barriers.
template <int NV, int BLOCK>
__device__ __forceinline__ void block_sum(double (&v)[NV], double *sh)
{
	constexpr int NW = BLOCK / WAVE;
	const int lane = threadIdx.x & (WAVE - 1), wid = threadIdx.x / WAVE;
#pragma unroll
	for (int a = 0; a < NV; a++) {
		double t = wave_sum(v[a]);
		if (lane == 0) sh[a * NW + wid] = t;
	}
	__syncthreads();
#pragma unroll
	for (int a = 0; a < NV; a++) {
		double t = 0;
#pragma unroll
		for (int w = 0; w < NW; w++) t += sh[a * NW + w];
		v[a] = t;
	}
	__syncthreads();
}

// ---------------------------------------------------------------------------
// 2-bit helpers.  A dword holds 16 samples; code c of sample s = (w >> 2s) & 3.

#define LO_MASK 0x55555555u

// bit 2s set iff code of sample s != 0
__device__ __forceinline__ uint32_t nz_fields(uint32_t w) { return (w | (w >> 1)) & LO_MASK; }

// 4-entry table lookup without dynamic register indexing
__device__ __forceinline__ double sel4(const double (&l)[4], uint32_t code)
{
	const double a = (code & 1u) ? l[1] : l[0];
	const double b = (code & 1u) ? l[3] : l[2];
	return (code & 2u) ? b : a;
}

// keep only the first `keep` samples of a dword (keep in [0,16])
__device__ __forceinline__ uint32_t keep_mask(int keep)
{
	return (keep >= 16) ? 0xFFFFFFFFu : ((keep <= 0) ? 0u : ((1u << (2 * keep)) - 1u));
}

// ---------------------------------------------------------------------------
// Filter + dosage table shared by every input format.
//   saige_main.cpp:288-295 (bin) / :197-204 (quant); vectorization.cpp:186-205
struct VarHead {
	double AF, AC, mac;
	int Num, minus, pass;
	double lut[4];
};

__device__ __forceinline__ VarHead make_head(const DevModel &md, double AC, int Num)
{
	VarHead h;
	const int N = md.N;
	h.AC = AC; h.Num = Num;
	h.AF = (Num > 0) ? (AC / (2 * Num)) : NAN;
	const double maf = fmin(h.AF, 1 - h.AF);
	h.mac = fmin(AC, 2 * Num - AC);
	const double missing = double(N - Num) / N;
	h.pass = (Num > 0) && (maf > 0) && (maf >= md.thr_maf) && (h.mac >= md.thr_mac) &&
		(missing <= md.thr_missing);
	h.minus = h.AF > 0.5;
	const double imp = 2 * h.AF;
	if (h.minus) { h.lut[0] = 2; h.lut[1] = 1; h.lut[2] = 0; h.lut[3] = 2 - imp; }
	else         { h.lut[0] = 0; h.lut[1] = 1; h.lut[2] = 2; h.lut[3] = imp; }
	return h;
}

// Series SPA stage (kern_spa4.h): a flagged variant starts in tier A (short series) while the predicted
// max_i |g_i t| is small: t ~ 1.5 x the first Newton point, |g_i| <= (2 + sum_k |c'_k| max_i |X_ik|) / sqrt(AC2).
#define SPA4_TIER_X 0.45
#define SPA5_NNZ 8192            /* variants with at most this many carriers go to the per-variant kernels (measured: N = 430 000 equal from 4 096 to 16 384, slower below; N = 50 000 best from 1 024 to 8 192, 9 % over 16 384) */
#define SPA5_EXACT_X 1.2         /* ... and straight to the exact sweeps beyond this predicted max |g t| */
// 0, 1: tiers A, B of the per-segment moments kernels;  2: per-variant kernel, series on the list;
// 3: per-variant kernel, exact exp/log sweeps
template <int K>
__device__ __forceinline__ int spa_tier(const DevModel &md, int nnz, double S, double var2, double AC2, const double (&c)[KMAX])
{
	double bmax = 0;
#pragma unroll
	for (int k = 0; k < K; k++) bmax += fabs(c[k]) * md.Xabs[k];
	const double t1 = fabs(S) * sqrt(AC2) / (sqrt(md.r) * var2);
	const double x = 1.5 * t1 * (2.0 + bmax) / sqrt(AC2);
	// Few carriers: one workgroup builds the variant's list once (a pass per sample segment would be
	// all overhead) and runs the series or, where that is hopeless, the exact sweeps on it.
	if (nnz <= SPA5_NNZ) return (x <= SPA5_EXACT_X) ? 2 : 3;
	return (x <= SPA4_TIER_X) ? 0 : 1;
}

// Record of a flagged variant into recs[] (written field by field: no copy of the record on the stack):
// tier A from slot 0 upwards (counters[0]), tier B from slot btop - 1 downwards (counters[7]), the
// variants of the per-variant kernels from slot btop upwards (counters[5]; their indices also onto the list
// of their kernel: fb_series / counters[3] or fb_exact / counters[4]).  btop = 0: one range only.
template <int K>
__device__ __forceinline__ void spa_push(const DevModel &md, SpaRec *recs, int *counters, int btop, int *fb_series,
	int *fb_exact, int j, int minus, double AC2, int nnz, const double (&lut)[4], double pn, double S, double var2,
	const double (&c)[KMAX])
{
	const int tier = btop > 0 ? spa_tier<K>(md, nnz, S, var2, AC2, c) : 0;
	int slot;
	if (tier >= 2) {
		slot = btop + atomicAdd(&counters[5], 1);
		if (tier == 2) fb_series[atomicAdd(&counters[3], 1)] = slot;
		else fb_exact[atomicAdd(&counters[4], 1)] = slot;
	} else {
		slot = tier ? btop - 1 - atomicAdd(&counters[7], 1) : atomicAdd(&counters[0], 1);
	}
	SpaRec *r = recs + slot;
	r->j = j; r->minus = minus; r->nnz = nnz; r->has_gmu = 0;
#pragma unroll
	for (int k = 0; k < 4; k++) r->lut[k] = lut[k];
	r->AC2 = AC2; r->p_noadj = pn; r->S = S; r->var2 = var2; r->sum_gmu = 0;
	r->tscale = spa_tscale(S, var2, AC2, md.r);
#pragma unroll
	for (int k = 0; k < KMAX; k++) r->c[k] = c[k];
}

// Score epilogue: from the P reduced sums to the output row; returns 1 when the
// variant has to go through the SPA stage.
//   binary saige_main.cpp:313-356, quantitative :225-272
template <int K>
__device__ __forceinline__ int score_epilogue(const DevModel &md, const VarHead &h, const double (&acc)[2 * K + 2],
	double *out, double (&c_out)[KMAX], double *p_noadj_out, double *S_out, double *var2_out)
{
	const double s = acc[2 * K], w = acc[2 * K + 1];
	double quad = 0, ec = 0, sac = 0;
#pragma unroll
	for (int a = 0; a < K; a++) {
		const double ca = acc[a];
#pragma unroll
		for (int b = 0; b < K; b++) quad += ca * acc[b] * md.XVX[a * K + b];
		ec += acc[K + a] * ca;
		sac += md.S_a[a] * ca;
	}
	const double var2 = quad + w - 2 * ec;
	const double S = s - sac;
	double pval, beta;
	if (md.quant) {
		const double inv_sqrt_mac = 1.0 / sqrt(h.mac), inv_mac = 1.0 / h.mac;
		const double var1 = var2 * inv_mac * md.r;
		const double Tstat = S * inv_sqrt_mac / md.tau0;
		pval = d_pchisq1_upper(Tstat * Tstat / var1);
		beta = Tstat / var1 * inv_sqrt_mac;
	} else {
		const double var1 = var2 * md.r;
		pval = d_pchisq1_upper(S * S / var1);
		beta = S / var1;
	}
	out[0] = h.AF; out[1] = h.mac; out[2] = h.Num;
	if (!md.quant) {
		const int converged = isfinite(pval);
		if (converged && pval <= md.thr_spa) {
			out[6] = pval;
			// Saddle_Prob_Fast's first test (SPATest.cpp:319-321): |q - m1| / sqrt(var1) < cutoff = 2
			// returns pval_noadj.  With q~ - m1 = Tstat / sqrt(var1) * sqrt(var2) (saige_main.cpp:381)
			// it needs nothing but the score stage's sums -- no carrier pass for the variants between
			// |z| = 1.96 and 2 (9 % of the flagged ones).
			const double AC2 = h.minus ? (2 * h.Num - h.AC) : h.AC;
			const double Tstat = S / sqrt(AC2), v2 = var2 / AC2, v1 = v2 * md.r;
			const double sdev = Tstat / sqrt(v1) * sqrt(v2);
			if (fabs(sdev) / sqrt(v2) < 2.0) {
				const double pn = d_pchisq1_upper(sdev * sdev / v2);
				double bs = (Tstat / v1) / sqrt(AC2);          // saige_main.cpp:392
				if (h.minus) bs = -bs;
				out[3] = bs;
				out[4] = fabs(bs / d_qnorm(pn / 2));
				out[5] = pn;
				out[7] = 1.0;
				return 0;
			}
#pragma unroll
			for (int a = 0; a < KMAX; a++) c_out[a] = a < K ? acc[a < K ? a : 0] : 0.0;
			*p_noadj_out = pval; *S_out = S; *var2_out = var2;
			return 1;
		}
		out[6] = pval; out[7] = converged ? 1.0 : 0.0;
	} else {
		out[6] = NAN; out[7] = NAN;
	}
	if (h.minus) beta = -beta;
	out[3] = beta;
	out[4] = fabs(beta / d_qnorm(pval / 2));
	out[5] = pval;
	return 0;
}

__device__ __forceinline__ void nan_row(double *out)
{
	const double n = NAN;
#pragma unroll
	for (int c = 0; c < 8; c++) out[c] = n;
}
