// saigehip.hip -- MI355X (gfx950) kernels + C ABI of the SAIGEgds single-variant
// scan.  Interface and reference citations: include/saigehip.h; design,
// data layout and rooflines: DESIGN.md.
//
// Math of the score stage.  For a variant with (imputed, flipped) dosage G the
// reference computes (src/saige_main.cpp:300-350, either branch)
//     adj = G - X c',  c' = (X'VX)^-1 X'V G,  S = sum y_mu*adj,
//     var = r * sum mu2*adj^2.
// Both branches reduce to four carrier-only sums over I = {i : G_i != 0}:
//     c'_k = sum_I G_i A[i,k]          A = t_XVX_inv_XV
//     e_k  = sum_I G_i mu2_i X[i,k]
//     s    = sum_I G_i y_mu_i
//     w    = sum_I G_i^2 mu2_i
//     var2 = c'^T XVX c' + w - 2 e.c'        S = s - S_a.c'
// (quantitative: mu2 == 1).  The per-sample vector F[i] = [A[i,:], mu2_i X[i,:],
// y_mu_i, mu2_i] (P = 2K+2 doubles, one 16-byte aligned row) is what a carrier
// costs in memory traffic.

#include <hip/hip_runtime.h>
#include <math.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>

#include "saigehip.h"

#define KMAX SGX_MAX_COEFF
#define WAVE 64

// ---------------------------------------------------------------------------
// error handling

static thread_local std::string g_err;

static int fail(int code, const char *fmt, ...)
{
	char buf[512];
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(buf, sizeof buf, fmt, ap);
	va_end(ap);
	g_err = buf;
	return code;
}

#define HIPCHK(expr)                                                              \
	do {                                                                          \
		hipError_t e_ = (expr);                                                   \
		if (e_ != hipSuccess)                                                     \
			return fail(SGX_EHIP, "%s failed: %s (%s:%d)", #expr,                 \
				hipGetErrorString(e_), __FILE__, __LINE__);                       \
	} while (0)

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
	double XVX[KMAX * KMAX];
	double S_a[KMAX];
};

// a variant handed from the score stage to the SPA stage
struct SpaRec {
	int j;            // variant index in the block
	int minus;        // AF > 0.5
	double lut[4];    // dosage value per 2-bit code after impute + flip
	double AC2;       // allele count of the tested (minor) allele
	double p_noadj;
	double c[KMAX];   // c' = XVX_inv_XV * G
};

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

// Score epilogue: from the P reduced sums to the output row; returns 1 when the
// variant has to go through the SPA stage.
//   binary saige_main.cpp:313-356, quantitative :225-272
__device__ int score_epilogue(const DevModel &md, const VarHead &h, const double *acc,
	double *out, double *c_out, double *p_noadj_out)
{
	const int K = md.K;
	const double *c = acc, *e = acc + K;
	const double s = acc[2 * K], w = acc[2 * K + 1];
	double quad = 0, ec = 0, sac = 0;
	for (int a = 0; a < K; a++) {
		const double ca = c[a];
		for (int b = 0; b < K; b++) quad += ca * c[b] * md.XVX[a * K + b];
		ec += e[a] * ca;
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
			for (int a = 0; a < K; a++) c_out[a] = c[a];
			*p_noadj_out = pval;
			out[6] = pval;
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
		double cbuf[KMAX], pn;
		valid[j] = 1;
		if (score_epilogue(md, h, acc, o, cbuf, &pn)) {
			const int slot = atomicAdd(&counters[0], 1);
			SpaRec r;
			r.j = j; r.minus = h.minus; r.AC2 = h.minus ? (2 * h.Num - h.AC) : h.AC;
			r.p_noadj = pn;
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
		double cbuf[KMAX], pn;
		valid[j] = 1;
		if (score_epilogue(md, h, acc, o, cbuf, &pn)) {
			const int slot = atomicAdd(&counters[0], 1);
			SpaRec r;
			r.j = j; r.minus = h.minus; r.AC2 = h.minus ? (2 * h.Num - h.AC) : h.AC;
			r.p_noadj = pn;
			// dosage rows carry real values: lut[3] holds the imputed value, the
			// SPA kernel re-reads the row itself
			for (int a = 0; a < 4; a++) r.lut[a] = h.lut[a];
			for (int a = 0; a < KMAX; a++) r.c[a] = (a < md.K) ? cbuf[a] : 0.0;
			recs[slot] = r;
		}
		atomicAdd(&counters[1], 1);
	}
}

// ---------------------------------------------------------------------------
// SPA stage.  One workgroup per flagged variant.
//   A. dense pass over all N samples (saige_main.cpp:359-385):
//        adj_i = (G_i - X_i.c') / sqrt(AC2);  q, m1, var2;  g_pos, g_neg
//      and compaction of the carriers' (adj, mu) into a private list
//      (SPATest.cpp:324-345), deterministic order (ascending sample index).
//   B. Saddle_Prob_Fast (SPATest.cpp:299-374): two safeguarded Newton root
//      searches (:139-184) whose K1/K2 sums run over the list with the whole
//      workgroup, then the Lugannani-Rice tail (:211-230).
// All threads execute the scalar control flow redundantly on identical values.

enum { IN_2BIT = 0, IN_U8 = 1, IN_F64 = 2 };

template <int INPUT>
__device__ __forceinline__ double load_dosage(const void *row, int i, const SpaRec &r)
{
	if (INPUT == IN_2BIT) {
		const uint8_t b = reinterpret_cast<const uint8_t *>(row)[i >> 2];
		return sel4(r.lut, (b >> ((i & 3) * 2)) & 3u);
	} else if (INPUT == IN_U8) {
		const uint8_t v = reinterpret_cast<const uint8_t *>(row)[i];
		double g = (v == 0xFF) ? r.lut[3] : (r.minus ? 2.0 - (double)v : (double)v);
		return g;
	} else {
		const double v = reinterpret_cast<const double *>(row)[i];
		double g = !isfinite(v) ? r.lut[3] : (r.minus ? 2.0 - v : v);
		return g;
	}
}

// K1 (without "- q") and K2 sums at t over the compact list
template <int BLOCK>
__device__ __forceinline__ void cgf_pass(double t, int nnz, const double *__restrict__ gl,
	const double *__restrict__ ml, double *sh, double &K1s, double &K2s)
{
	double v[2] = {0, 0};
	for (int k = threadIdx.x; k < nnz; k += BLOCK) {
		const double g = gl[k], m = ml[k], om = 1 - m;
		const double e = exp(-g * t);
		const double d = om * e + m;
		v[0] += m * g / d;                       // SPATest.cpp:64
		const double t2 = (om * m * g * g * e) / (d * d);   // :79
		if (isfinite(t2)) v[1] += t2;            // :80
	}
	block_sum<2, BLOCK>(v, sh);
	K1s = v[0]; K2s = v[1];
}

template <int BLOCK>
__device__ __forceinline__ double korg_pass(double t, int nnz, const double *__restrict__ gl,
	const double *__restrict__ ml, double *sh)
{
	double v[1] = {0};
	for (int k = threadIdx.x; k < nnz; k += BLOCK) {
		const double g = gl[k], m = ml[k];
		v[0] += log(1 - m + m * exp(g * t));     // SPATest.cpp:49
	}
	block_sum<1, BLOCK>(v, sh);
	return v[0];
}

// getroot_K1_fast (SPATest.cpp:139-184).  Returns root; K2 at the root in k2_root.
template <int BLOCK>
__device__ double getroot_fast(double g_pos, double g_neg, double q, double NAmu, double NAsigma,
	int nnz, const double *gl, const double *ml, double *sh, bool &converged, double &k2_root)
{
	const double tol = 0.0001220703125;   // DBL_EPSILON^(1/4), SPATest.cpp:87
	const int maxiter = 1000;
	k2_root = 0;
	if (q >= g_pos || q <= g_neg) { converged = true; return INFINITY; }
	double t = 0, root = 0, K1s, K2s;
	cgf_pass<BLOCK>(t, nnz, gl, ml, sh, K1s, K2s);
	double K1_eval = (K1s - q) + NAmu + NAsigma * t;
	double prevJump = INFINITY;
	converged = false;
	for (int it = 1; it <= maxiter; it++) {
		const double K2_eval = K2s + NAsigma;
		double tnew = t - K1_eval / K2_eval;
		if (!isfinite(tnew)) break;
		if (fabs(tnew - t) < tol) { converged = true; break; }
		double K1n, K2n;
		cgf_pass<BLOCK>(tnew, nnz, gl, ml, sh, K1n, K2n);
		double newK1 = (K1n - q) + NAmu + NAsigma * tnew;
		if (d_sign(K1_eval) != d_sign(newK1)) {
			if (fabs(tnew - t) > prevJump - tol) {
				tnew = t + d_sign(newK1 - K1_eval) * prevJump * 0.5;
				cgf_pass<BLOCK>(tnew, nnz, gl, ml, sh, K1n, K2n);
				newK1 = (K1n - q) + NAmu + NAsigma * tnew;
				prevJump *= 0.5;
			} else {
				prevJump = fabs(tnew - t);
			}
		}
		root = t = tnew;
		K1_eval = newK1;
		K2s = K2n;
	}
	k2_root = K2s;   // K2 at t == root
	return root;
}

// get_saddle_prob_fast (SPATest.cpp:211-230); k2s = K2 sum at t (already known)
template <int BLOCK>
__device__ double saddle_prob_fast(double t, double k2s, double q, double NAmu, double NAsigma,
	int nnz, const double *gl, const double *ml, double *sh)
{
	if (!isfinite(t)) return 0;
	const double K = korg_pass<BLOCK>(t, nnz, gl, ml, sh) + NAmu * t + 0.5 * NAsigma * t * t;
	const double k2 = k2s + NAsigma;
	double pval = 0;
	if (isfinite(K) && isfinite(k2)) {
		const double w = d_sign(t) * sqrt(2 * (t * q - K));
		const double v = t * sqrt(k2);
		const double z = w + log(v / w) / w;
		if (z > 0) pval = d_pnorm_upper(z);
		else pval = -d_pnorm_lower(z);
	}
	return pval;
}

template <int K, int BLOCK, int INPUT>
__global__ void __launch_bounds__(BLOCK)
spa_kernel(const void *__restrict__ rows, size_t row_bytes, DevModel md,
	const SpaRec *__restrict__ recs, const int *__restrict__ counters,
	double *__restrict__ scratch, size_t scratch_stride, double *__restrict__ out8)
{
	constexpr int NW = BLOCK / WAVE;
	__shared__ double sh[8 * NW];
	__shared__ int shc[NW];
	const int N = md.N, tid = threadIdx.x;
	const int lane = tid & (WAVE - 1), wid = tid / WAVE;
	const int nflag = counters[0];
	double *gl = scratch + (size_t)blockIdx.x * scratch_stride;
	double *ml = gl + scratch_stride / 2;
	// contiguous sample segment per wave, multiple of 64
	const int seg = (((N + NW - 1) / NW) + 63) & ~63;
	const int s0 = wid * seg, s1 = min(N, s0 + seg);

	for (int v = blockIdx.x; v < nflag; v += gridDim.x) {
		const SpaRec r = recs[v];
		const void *row = reinterpret_cast<const uint8_t *>(rows) + (size_t)r.j * row_bytes;
		const double inv = 1 / sqrt(r.AC2);
		double c[K];
#pragma unroll
		for (int a = 0; a < K; a++) c[a] = r.c[a];

		// ---- A1: carriers per wave segment -> list offsets
		int cnt = 0;
		for (int i = s0 + lane; i < s1; i += WAVE) cnt += (load_dosage<INPUT>(row, i, r) != 0);
		cnt = wave_sum_i(cnt);
		__syncthreads();            // previous variant's readers of shc/list are done
		if (lane == 0) shc[wid] = cnt;
		__syncthreads();
		int base = 0, nnz = 0;
#pragma unroll
		for (int w = 0; w < NW; w++) { if (w < wid) base += shc[w]; nnz += shc[w]; }

		// ---- A2: dense pass
		double a7[7] = {0, 0, 0, 0, 0, 0, 0};  // q, m1, var2, g_pos, g_neg, sum g*mu, sum g^2 mu(1-mu)
		for (int i0 = s0; i0 < s1; i0 += WAVE) {
			const int i = i0 + lane;
			const bool in = i < s1;
			double G = 0, adj = 0, mui = 0;
			if (in) {
				G = load_dosage<INPUT>(row, i, r);
				const double *x = md.X + (size_t)i * K;
				double d = 0;
#pragma unroll
				for (int a = 0; a < K; a++) d = fma(x[a], c[a], d);
				adj = (G - d) * inv;
				mui = md.mu[i];
				a7[0] = fma(md.y[i], adj, a7[0]);
				a7[1] = fma(mui, adj, a7[1]);
				a7[2] = fma(md.mu2[i] * adj, adj, a7[2]);
				if (adj > 0) a7[3] += adj; else a7[4] += adj;
			}
			const bool carrier = in && (G != 0);
			const unsigned long long mask = __ballot(carrier);
			if (carrier) {
				const int pos = base + __popcll(mask & ((1ull << lane) - 1ull));
				gl[pos] = adj; ml[pos] = mui;
				a7[5] = fma(adj, mui, a7[5]);
				a7[6] = fma(adj * adj, mui * (1 - mui), a7[6]);
			}
			base += __popcll(mask);
		}
		block_sum<7, BLOCK>(a7, sh);   // its barriers also publish the list to the workgroup

		// ---- B: saige_main.cpp:379-395 + Saddle_Prob_Fast
		const double q = a7[0], m1 = a7[1], var2 = a7[2], g_pos = a7[3], g_neg = a7[4];
		const double var1 = var2 * md.r;
		const double Tstat = q - m1;
		const double qtilde = Tstat / sqrt(var1) * sqrt(var2) + m1;
		const double s = qtilde - m1;
		const double qinv = -s + m1;
		const double pn_in = d_pchisq1_upper(s * s / var2);
		double pval;
		bool converged = true;
		if (fabs(qtilde - m1) / sqrt(var2) < 2.0) {
			pval = pn_in;
		} else {
			const double NAmu = m1 - a7[5], NAsigma = var2 - a7[6];
			bool conv1, conv2;
			double k2r1, k2r2;
			const double root1 = getroot_fast<BLOCK>(g_pos, g_neg, qtilde, NAmu, NAsigma, nnz, gl, ml, sh, conv1, k2r1);
			const double root2 = getroot_fast<BLOCK>(g_pos, g_neg, qinv, NAmu, NAsigma, nnz, gl, ml, sh, conv2, k2r2);
			if (conv1 && conv2) {
				const double p1 = saddle_prob_fast<BLOCK>(root1, k2r1, qtilde, NAmu, NAsigma, nnz, gl, ml, sh);
				const double p2 = saddle_prob_fast<BLOCK>(root2, k2r2, qinv, NAmu, NAsigma, nnz, gl, ml, sh);
				pval = fabs(p1) + fabs(p2);
				// SPATest.cpp:368-371: the cutoff doubles until |z| < cutoff, the
				// roots do not change, so the loop always ends in pval_noadj
				if (pval != 0 && pn_in / pval > 1000) pval = pn_in;
			} else {
				pval = pn_in;
				converged = false;
			}
		}
		if (pval == 0 && r.p_noadj > 0) { pval = r.p_noadj; converged = false; }
		if (tid == 0) {
			double beta = (Tstat / var1) / sqrt(r.AC2);
			if (r.minus) beta = -beta;
			double *o = out8 + (size_t)r.j * 8;
			o[3] = beta;
			o[4] = fabs(beta / d_qnorm(pval / 2));
			o[5] = pval;
			o[7] = converged ? 1.0 : 0.0;
		}
	}
}

// ---------------------------------------------------------------------------
// Synthetic 2-bit genotypes (bench / tests): see saigehip.h, sgx_synth_2bit_dev

__host__ __device__ __forceinline__ uint64_t splitmix64(uint64_t x)
{
	x += 0x9E3779B97F4A7C15ull;
	x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
	x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
	return x ^ (x >> 31);
}

__global__ void __launch_bounds__(256)
synth2b_kernel(uint8_t *__restrict__ packed, size_t bpv, int N, size_t M, uint64_t first,
	uint64_t seed, const uint32_t *__restrict__ thr)
{
	const size_t j = blockIdx.y;
	if (j >= M) return;
	const uint32_t t0 = thr[3 * j], t1 = thr[3 * j + 1], tm = thr[3 * j + 2];
	const uint64_t key = splitmix64(seed ^ splitmix64(first + j));
	uint32_t *row = reinterpret_cast<uint32_t *>(packed + j * bpv);
	const int nd = (int)(bpv / 4);
	for (int d = blockIdx.x * blockDim.x + threadIdx.x; d < nd; d += gridDim.x * blockDim.x) {
		uint32_t w = 0;
#pragma unroll
		for (int s = 0; s < 16; s++) {
			const int i = d * 16 + s;
			if (i < N) {
				const uint64_t x = splitmix64(key + (uint64_t)i);
				const uint32_t u = (uint32_t)(x >> 32), m = (uint32_t)x;
				uint32_t code = (u < t0) ? 0u : ((u < t1) ? 1u : 2u);
				if (m < tm) code = 3u;
				w |= code << (2 * s);
			}
		}
		row[d] = w;
	}
}

// ---------------------------------------------------------------------------
// host side

struct sgx_handle {
	int device = 0;
	hipStream_t stream = nullptr;
	DevModel md{};
	double *dF = nullptr, *dX = nullptr, *dy = nullptr, *dmu = nullptr, *dmu2 = nullptr;
	// per-call workspace
	SpaRec *recs = nullptr; size_t recs_cap = 0;
	int *counters = nullptr;          // [0] n_spa, [1] n_valid
	int *h_counters = nullptr;        // pinned
	double *scratch = nullptr; size_t scratch_stride = 0; int spa_grid = 0;
	// host-pointer staging
	uint8_t *stage_in = nullptr; size_t stage_in_cap = 0;
	double *stage_out = nullptr; uint8_t *stage_valid = nullptr; size_t stage_out_cap = 0;
	hipEvent_t ev[3] = {nullptr, nullptr, nullptr};
	sgx_stats stats{};
	bool stats_pending = false;
};

static int set_dev(sgx_handle *h)
{
	HIPCHK(hipSetDevice(h->device));
	return SGX_OK;
}

extern "C" const char *sgx_version(void) { return "saigehip 0.1 (gfx950)"; }
extern "C" const char *sgx_last_error(void) { return g_err.c_str(); }

extern "C" int sgx_device_count(void)
{
	int n = 0;
	if (hipGetDeviceCount(&n) != hipSuccess) return 0;
	return n;
}

extern "C" size_t sgx_row_stride(int32_t n_samp)
{
	return (size_t)((n_samp + 63) / 64) * 16;
}

static double thr_or(double v, double dflt) { return std::isfinite(v) ? v : dflt; }

extern "C" int sgx_set_thresholds(sgx_handle *h, double maf, double mac, double missing,
	double spa_pval)
{
	if (!h) return fail(SGX_EINVAL, "sgx_set_thresholds: NULL handle");
	// saige_main.cpp:108-115
	h->md.thr_maf = thr_or(maf, -1);
	h->md.thr_mac = thr_or(mac, -1);
	h->md.thr_missing = thr_or(missing, 1);
	h->md.thr_spa = thr_or(spa_pval, 0.05);
	return SGX_OK;
}

template <typename T>
static int dev_upload(T **dst, const std::vector<T> &src)
{
	HIPCHK(hipMalloc((void **)dst, src.size() * sizeof(T)));
	HIPCHK(hipMemcpy(*dst, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice));
	return SGX_OK;
}

extern "C" int sgx_init(const sgx_model *m, int device, sgx_handle **out)
{
	if (!m || !out) return fail(SGX_EINVAL, "sgx_init: NULL argument");
	*out = nullptr;
	const int N = m->n_samp, K = m->n_coeff;
	if (N <= 0) return fail(SGX_EINVAL, "sgx_init: n_samp = %d", N);
	if (K < 1 || K > KMAX)
		return fail(SGX_EINVAL, "sgx_init: n_coeff = %d, supported 1..%d", K, KMAX);
	if (m->trait != SGX_TRAIT_BINARY && m->trait != SGX_TRAIT_QUANT)
		return fail(SGX_EINVAL, "sgx_init: invalid trait %d", m->trait);
	if (!m->y || !m->mu || !m->y_mu || !m->mu2 || !m->t_XVX_inv_XV || !m->t_X || !m->XVX || !m->S_a)
		return fail(SGX_EINVAL, "sgx_init: NULL model array");
	int ndev = 0;
	if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
		return fail(SGX_ENODEV, "sgx_init: no HIP device available");
	if (device < 0 || device >= ndev)
		return fail(SGX_EINVAL, "sgx_init: device %d out of range (0..%d)", device, ndev - 1);

	sgx_handle *h = new sgx_handle();
	h->device = device;
	int rc = set_dev(h);
	if (rc) { delete h; return rc; }
	const int P = 2 * K + 2;
	const bool quant = m->trait == SGX_TRAIT_QUANT;
	std::vector<double> F((size_t)N * P), X((size_t)N * K), y(m->y, m->y + N),
		mu(m->mu, m->mu + N), mu2(m->mu2, m->mu2 + N);
	for (int i = 0; i < N; i++) {
		const double w = quant ? 1.0 : m->mu2[i];   // quantitative: plain sums, saige_main.cpp:227-228
		double *f = &F[(size_t)i * P];
		for (int k = 0; k < K; k++) {
			f[k] = m->t_XVX_inv_XV[(size_t)i * K + k];
			f[K + k] = w * m->t_X[(size_t)i * K + k];
			X[(size_t)i * K + k] = m->t_X[(size_t)i * K + k];
		}
		f[2 * K] = m->y_mu[i];
		f[2 * K + 1] = w;
	}
	DevModel &md = h->md;
	md.N = N; md.K = K; md.P = P; md.quant = quant;
	md.tau0 = m->tau[0]; md.r = m->var_ratio;
	sgx_set_thresholds(h, m->maf, m->mac, m->missing, m->spa_pval);
	for (int a = 0; a < K * K; a++) md.XVX[a] = m->XVX[a];
	for (int a = 0; a < K; a++) md.S_a[a] = m->S_a[a];
#define TRY(x) do { rc = (x); if (rc) { sgx_free(h); return rc; } } while (0)
	TRY(dev_upload(&h->dF, F));
	TRY(dev_upload(&h->dX, X));
	TRY(dev_upload(&h->dy, y));
	TRY(dev_upload(&h->dmu, mu));
	TRY(dev_upload(&h->dmu2, mu2));
	md.F = h->dF; md.X = h->dX; md.y = h->dy; md.mu = h->dmu; md.mu2 = h->dmu2;
	hipError_t e;
#define TRYH(x) do { e = (x); if (e != hipSuccess) { sgx_free(h); return fail(SGX_EHIP, "%s: %s", #x, hipGetErrorString(e)); } } while (0)
	TRYH(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
	TRYH(hipMalloc((void **)&h->counters, 4 * sizeof(int)));
	TRYH(hipHostMalloc((void **)&h->h_counters, 4 * sizeof(int), hipHostMallocDefault));
	for (int i = 0; i < 3; i++) TRYH(hipEventCreate(&h->ev[i]));
	// SPA scratch: one (adj, mu) list of N entries per resident workgroup
	hipDeviceProp_t prop;
	TRYH(hipGetDeviceProperties(&prop, device));
	h->spa_grid = prop.multiProcessorCount * 4;
	h->scratch_stride = 2 * (((size_t)N + 63) & ~(size_t)63);
	TRYH(hipMalloc((void **)&h->scratch, h->scratch_stride * sizeof(double) * h->spa_grid));
#undef TRY
#undef TRYH
	*out = h;
	return SGX_OK;
}

extern "C" void sgx_free(sgx_handle *h)
{
	if (!h) return;
	(void)hipSetDevice(h->device);
	if (h->stream) (void)hipStreamSynchronize(h->stream);
	(void)hipFree(h->dF); (void)hipFree(h->dX); (void)hipFree(h->dy);
	(void)hipFree(h->dmu); (void)hipFree(h->dmu2);
	(void)hipFree(h->recs); (void)hipFree(h->counters); (void)hipFree(h->scratch);
	(void)hipFree(h->stage_in); (void)hipFree(h->stage_out); (void)hipFree(h->stage_valid);
	if (h->h_counters) (void)hipHostFree(h->h_counters);
	for (int i = 0; i < 3; i++) if (h->ev[i]) (void)hipEventDestroy(h->ev[i]);
	if (h->stream) (void)hipStreamDestroy(h->stream);
	delete h;
}

static int ensure_recs(sgx_handle *h, size_t n)
{
	if (n <= h->recs_cap) return SGX_OK;
	HIPCHK(hipStreamSynchronize(h->stream));
	if (h->recs) HIPCHK(hipFree(h->recs));
	h->recs = nullptr; h->recs_cap = 0;
	HIPCHK(hipMalloc((void **)&h->recs, n * sizeof(SpaRec)));
	h->recs_cap = n;
	return SGX_OK;
}

// ---- kernel dispatch over the compile-time K ------------------------------

#define FOR_EACH_K(X) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15) X(16)

template <int INPUT>
static int launch_scan(sgx_handle *h, const void *rows, size_t row_bytes, size_t M,
	double *out8, uint8_t *valid)
{
	const DevModel &md = h->md;
	constexpr int SB = 256, PB = 512;
	hipStream_t st = h->stream;
	HIPCHK(hipMemsetAsync(h->counters, 0, 4 * sizeof(int), st));
	HIPCHK(hipEventRecord(h->ev[0], st));
	const dim3 grid((unsigned)M);
	switch (md.K) {
#define CASE(KK)                                                                             \
	case KK:                                                                                 \
		if (INPUT == IN_2BIT)                                                                \
			hipLaunchKernelGGL((score2b_kernel<2 * KK + 2, SB>), grid, dim3(SB), 0, st,     \
				(const uint8_t *)rows, row_bytes, (int)M, md, h->recs, h->counters, out8, valid); \
		else if (INPUT == IN_U8)                                                             \
			hipLaunchKernelGGL((score_ds_kernel<2 * KK + 2, SB, uint8_t>), grid, dim3(SB), 0, st, \
				(const uint8_t *)rows, (int)M, md, h->recs, h->counters, out8, valid);      \
		else                                                                                 \
			hipLaunchKernelGGL((score_ds_kernel<2 * KK + 2, SB, double>), grid, dim3(SB), 0, st, \
				(const double *)rows, (int)M, md, h->recs, h->counters, out8, valid);       \
		break;
		FOR_EACH_K(CASE)
#undef CASE
	default: return fail(SGX_EINVAL, "unsupported K=%d", md.K);
	}
	HIPCHK(hipGetLastError());
	HIPCHK(hipEventRecord(h->ev[1], st));
	h->stats.score_launches = 1;
	h->stats.spa_launches = 0;
	if (!md.quant) {
		const dim3 sgrid((unsigned)std::min<size_t>(M, (size_t)h->spa_grid));
		switch (md.K) {
#define CASE(KK)                                                                             \
	case KK:                                                                                 \
		hipLaunchKernelGGL((spa_kernel<KK, PB, INPUT>), sgrid, dim3(PB), 0, st, rows,        \
			row_bytes, md, h->recs, h->counters, h->scratch, h->scratch_stride, out8);       \
		break;
			FOR_EACH_K(CASE)
#undef CASE
		}
		HIPCHK(hipGetLastError());
		h->stats.spa_launches = 1;
	}
	HIPCHK(hipEventRecord(h->ev[2], st));
	HIPCHK(hipMemcpyAsync(h->h_counters, h->counters, 4 * sizeof(int), hipMemcpyDeviceToHost, st));
	h->stats.n_variants = M;
	h->stats_pending = true;
	return SGX_OK;
}

extern "C" int sgx_sync(sgx_handle *h)
{
	if (!h) return fail(SGX_EINVAL, "sgx_sync: NULL handle");
	int rc = set_dev(h);
	if (rc) return rc;
	HIPCHK(hipStreamSynchronize(h->stream));
	if (h->stats_pending) {
		h->stats.n_spa = (uint64_t)h->h_counters[0];
		h->stats.n_valid = (uint64_t)h->h_counters[1];
		float a = 0, b = 0, c = 0;
		(void)hipEventElapsedTime(&a, h->ev[0], h->ev[1]);
		(void)hipEventElapsedTime(&b, h->ev[1], h->ev[2]);
		(void)hipEventElapsedTime(&c, h->ev[0], h->ev[2]);
		h->stats.ms_score = a; h->stats.ms_spa = b; h->stats.ms_total = c;
		h->stats_pending = false;
	}
	return SGX_OK;
}

extern "C" int sgx_get_stats(sgx_handle *h, sgx_stats *st)
{
	if (!h || !st) return fail(SGX_EINVAL, "sgx_get_stats: NULL argument");
	int rc = sgx_sync(h);
	if (rc) return rc;
	*st = h->stats;
	return SGX_OK;
}

extern "C" int sgx_scan_2bit_dev(sgx_handle *h, const uint8_t *packed_dev, size_t bpv,
	size_t M, double *out8_dev, uint8_t *valid_dev)
{
	if (!h) return fail(SGX_EINVAL, "sgx_scan_2bit_dev: NULL handle");
	if (M == 0) return SGX_OK;
	if (!packed_dev || !out8_dev || !valid_dev)
		return fail(SGX_EINVAL, "sgx_scan_2bit_dev: NULL buffer");
	if (M > 0x7fffffffu) return fail(SGX_EINVAL, "sgx_scan_2bit_dev: too many variants in one call");
	if (bpv % 16 != 0 || bpv < sgx_row_stride(h->md.N))
		return fail(SGX_EINVAL, "Invalid length of dosages: bytes_per_variant=%zu, need a multiple of 16 >= %zu",
			bpv, sgx_row_stride(h->md.N));
	if (((uintptr_t)packed_dev & 15u) != 0)
		return fail(SGX_EINVAL, "sgx_scan_2bit_dev: packed_dev must be 16-byte aligned");
	int rc = set_dev(h);
	if (rc) return rc;
	rc = ensure_recs(h, M);
	if (rc) return rc;
	return launch_scan<IN_2BIT>(h, packed_dev, bpv, M, out8_dev, valid_dev);
}

static int ensure_stage(sgx_handle *h, size_t in_bytes, size_t M)
{
	if (in_bytes > h->stage_in_cap) {
		if (h->stage_in) HIPCHK(hipFree(h->stage_in));
		h->stage_in = nullptr; h->stage_in_cap = 0;
		HIPCHK(hipMalloc((void **)&h->stage_in, in_bytes));
		h->stage_in_cap = in_bytes;
	}
	if (M > h->stage_out_cap) {
		if (h->stage_out) HIPCHK(hipFree(h->stage_out));
		if (h->stage_valid) HIPCHK(hipFree(h->stage_valid));
		h->stage_out = nullptr; h->stage_valid = nullptr; h->stage_out_cap = 0;
		HIPCHK(hipMalloc((void **)&h->stage_out, M * 8 * sizeof(double)));
		HIPCHK(hipMalloc((void **)&h->stage_valid, M));
		h->stage_out_cap = M;
	}
	return SGX_OK;
}

// host-buffer scans run in chunks so the staging area stays bounded
static const size_t STAGE_BYTES = (size_t)1 << 30;

template <int INPUT>
static int scan_host(sgx_handle *h, const void *rows, size_t src_row_bytes, size_t dev_row_bytes,
	size_t M, double *out8, uint8_t *valid)
{
	if (!h) return fail(SGX_EINVAL, "scan: NULL handle");
	if (M == 0) return SGX_OK;
	if (!rows || !out8 || !valid) return fail(SGX_EINVAL, "scan: NULL buffer");
	int rc = set_dev(h);
	if (rc) return rc;
	size_t chunk = std::max<size_t>(1, STAGE_BYTES / dev_row_bytes);
	chunk = std::min(chunk, M);
	rc = ensure_stage(h, chunk * dev_row_bytes, chunk);
	if (rc) return rc;
	rc = ensure_recs(h, chunk);
	if (rc) return rc;
	sgx_stats total{};
	for (size_t off = 0; off < M; off += chunk) {
		const size_t m = std::min(chunk, M - off);
		const uint8_t *src = reinterpret_cast<const uint8_t *>(rows) + off * src_row_bytes;
		if (src_row_bytes == dev_row_bytes) {
			HIPCHK(hipMemcpyAsync(h->stage_in, src, m * dev_row_bytes, hipMemcpyHostToDevice, h->stream));
		} else {
			if (dev_row_bytes > src_row_bytes)
				HIPCHK(hipMemsetAsync(h->stage_in, 0, m * dev_row_bytes, h->stream));
			HIPCHK(hipMemcpy2DAsync(h->stage_in, dev_row_bytes, src, src_row_bytes,
				std::min(src_row_bytes, dev_row_bytes), m, hipMemcpyHostToDevice, h->stream));
		}
		rc = launch_scan<INPUT>(h, h->stage_in, dev_row_bytes, m, h->stage_out, h->stage_valid);
		if (rc) return rc;
		HIPCHK(hipMemcpyAsync(out8 + off * 8, h->stage_out, m * 8 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
		HIPCHK(hipMemcpyAsync(valid + off, h->stage_valid, m, hipMemcpyDeviceToHost, h->stream));
		rc = sgx_sync(h);
		if (rc) return rc;
		total.n_variants += h->stats.n_variants; total.n_valid += h->stats.n_valid;
		total.n_spa += h->stats.n_spa; total.ms_score += h->stats.ms_score;
		total.ms_spa += h->stats.ms_spa; total.ms_total += h->stats.ms_total;
		total.score_launches += h->stats.score_launches; total.spa_launches += h->stats.spa_launches;
	}
	h->stats = total;
	return SGX_OK;
}

extern "C" int sgx_scan_2bit(sgx_handle *h, const uint8_t *packed, size_t bpv, size_t M,
	double *out8, uint8_t *valid)
{
	if (h && bpv < (size_t)(h->md.N + 3) / 4)
		return fail(SGX_EINVAL, "Invalid length of dosages: bytes_per_variant=%zu < ceil(N/4)=%zu",
			bpv, (size_t)(h->md.N + 3) / 4);
	return scan_host<IN_2BIT>(h, packed, bpv, h ? sgx_row_stride(h->md.N) : 0, M, out8, valid);
}

extern "C" int sgx_scan_u8(sgx_handle *h, const uint8_t *dosage, size_t M, double *out8, uint8_t *valid)
{
	const size_t rb = h ? (size_t)h->md.N : 0;
	return scan_host<IN_U8>(h, dosage, rb, rb, M, out8, valid);
}

extern "C" int sgx_scan_f64(sgx_handle *h, const double *dosage, size_t M, double *out8, uint8_t *valid)
{
	const size_t rb = h ? (size_t)h->md.N * sizeof(double) : 0;
	return scan_host<IN_F64>(h, dosage, rb, rb, M, out8, valid);
}

extern "C" int sgx_synth_2bit_dev(sgx_handle *h, uint8_t *packed_dev, size_t bpv, int32_t n_samp,
	size_t M, uint64_t first_variant, uint64_t seed, const uint32_t *thr_dev)
{
	if (!h || !packed_dev || !thr_dev) return fail(SGX_EINVAL, "sgx_synth_2bit_dev: NULL argument");
	if (bpv % 4 != 0 || bpv < (size_t)(n_samp + 3) / 4)
		return fail(SGX_EINVAL, "sgx_synth_2bit_dev: bad bytes_per_variant %zu", bpv);
	int rc = set_dev(h);
	if (rc) return rc;
	const size_t MAXY = 32768;
	for (size_t off = 0; off < M; off += MAXY) {
		const size_t m = std::min(MAXY, M - off);
		const int nd = (int)(bpv / 4);
		const dim3 grid((unsigned)std::min(64, (nd + 255) / 256), (unsigned)m);
		hipLaunchKernelGGL(synth2b_kernel, grid, dim3(256), 0, h->stream, packed_dev + off * bpv,
			bpv, (int)n_samp, m, first_variant + off, seed, thr_dev + 3 * off);
		HIPCHK(hipGetLastError());
	}
	return SGX_OK;
}
