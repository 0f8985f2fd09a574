// kern_spa2.h -- SPA stage v2 for 2-bit genotypes: carrier-only extraction,
// both Newton root searches fused into shared passes over the carrier list.
// Part of libsaigehip.so (single translation unit: saigehip.hip).
#pragma once

// What the reference does per flagged variant (saige_main.cpp:356-395,
// SPATest.cpp:299-374) and how it is restated here without a pass over all N:
//
//   adj_i = (G_i - X_i.c') / sqrt(AC2)
//   Tstat = q - m1 = sum (y-mu) adj          = S / sqrt(AC2)      (score stage)
//   var2  = sum mu2 adj^2                     = var2_score / AC2   (score stage)
//   m1    = sum mu adj = (sum_I mu_i G_i - (X'mu).c') / sqrt(AC2)  (carriers + constant)
//   NAmu, NAsigma, the (adj, mu) list                               (carriers)
//   g_pos = sum_{adj>0} adj, g_neg = sum_{adj<=0} adj are used only in the test
//   "q >= g_pos || q <= g_neg" (SPATest.cpp:145).  Non-carriers have
//   adj_i = -X_i.c'/sqrt(AC2) whose SUM is known from constants, which bounds
//       g_pos >= L = sum_I max(adj,0) + max(-nb,0),
//       g_neg <= U = sum_I min(adj,0) + min(-nb,0),   nb = ((X'1).c' - sum_I X_i.c')/sqrt(AC2).
//   If U < q < L (with a safety margin) for both roots the test is false for
//   certain; otherwise the variant is handed to the v1 kernel (kern_spa.h), which
//   makes the exact dense pass.  The semantics are unchanged either way.

struct RootState {
	double q;          // right-hand side of K1(t) = q
	double t, root;    // current iterate / value returned by getroot_K1_fast
	double K1_eval;    // K1_adj(t) + NAmu + NAsigma t
	double K2cur;      // K2 sum at t
	double Kcur;       // Korg sum at t (valid when k_ok)
	double prevJump;
	double st_prev;    // size of the previous Newton step (for the last-evaluation guess only)
	double tnew;       // point being evaluated
	int it;
	int phase;         // 0 first evaluation at t=0, 1 Newton candidate, 2 bisected candidate
	bool active;       // needs another evaluation at tnew
	bool converged;
	bool want_k;       // the evaluation at tnew should include Korg (it is probably the last one)
	bool k_ok;         // Kcur belongs to the current t
};

#define SPA_TOL 0.0001220703125   /* DBL_EPSILON^(1/4), SPATest.cpp:87 */
#define SPA_MAXITER 1000          /* SPATest.cpp:88 */

// start of getroot_K1_fast, SPATest.cpp:145-154
__device__ __forceinline__ void root_begin(RootState &s, double q, double g_pos_lb, double g_neg_ub)
{
	s.q = q; s.t = 0; s.root = 0; s.K1_eval = 0; s.K2cur = 0; s.Kcur = 0; s.prevJump = INFINITY;
	s.tnew = 0; s.it = 1; s.phase = 0; s.active = true; s.converged = false;
	s.want_k = false; s.k_ok = false; s.st_prev = 0;
	(void)g_pos_lb; (void)g_neg_ub;
}

// loop head, SPATest.cpp:155-165: Newton step from (t, K1_eval, K2cur)
__device__ __forceinline__ void root_step(RootState &s, double NAsigma)
{
	if (s.it > SPA_MAXITER) { s.active = false; return; }   // loop exhausted, converged stays false
	const double K2_eval = s.K2cur + NAsigma;
	const double tnew = s.t - s.K1_eval / K2_eval;
	if (!isfinite(tnew)) { s.active = false; return; }
	if (fabs(tnew - s.t) < SPA_TOL) { s.converged = true; s.active = false; return; }
	s.tnew = tnew; s.phase = 1; s.active = true;
	// Newton converges quadratically: if the step after this one is predicted to fall under
	// the tolerance, this evaluation is the last and its point becomes the root, so Korg is
	// wanted with it (a wrong guess only costs time: spa3_korg covers the rest)
	// Steps shrink like st' ~ C st^2 with C ~ st / st_prev^2 once two steps are known.  A guess on
	// the generous side is cheap (one log per carrier), a missed one costs a sweep over the list.
	const double st = fabs(tnew - s.t);
	const double next = (s.st_prev > 0) ? st * st * st / (s.st_prev * s.st_prev) : st * st / (8.0 * fmax(fabs(tnew), 1e-3));
	s.want_k = next < 1.5 * SPA_TOL;
	s.st_prev = st;
}

// consume the sums evaluated at s.tnew, SPATest.cpp:152,166-181
__device__ __forceinline__ void root_feed(RootState &s, double K1s, double K2s, double NAmu, double NAsigma,
	double Ks = 0.0, bool k_valid = false)
{
	if (s.phase == 0) {
		s.K1_eval = (K1s - s.q) + NAmu + NAsigma * s.t;
		s.K2cur = K2s;
		s.Kcur = Ks; s.k_ok = k_valid;
		root_step(s, NAsigma);
		return;
	}
	const double newK1 = (K1s - s.q) + NAmu + NAsigma * s.tnew;
	if (s.phase == 1 && d_sign(s.K1_eval) != d_sign(newK1)) {
		if (fabs(s.tnew - s.t) > s.prevJump - SPA_TOL) {
			s.tnew = s.t + d_sign(newK1 - s.K1_eval) * s.prevJump * 0.5;
			s.prevJump *= 0.5;
			s.phase = 2;          // re-evaluate at the bisected point
			s.want_k = false;
			return;
		}
		s.prevJump = fabs(s.tnew - s.t);
	}
	s.root = s.t = s.tnew;
	s.K1_eval = newK1;
	s.K2cur = K2s;
	s.Kcur = Ks; s.k_ok = k_valid;
	s.it++;
	root_step(s, NAsigma);
}

__global__ void fastmath_selftest_kernel(const double *x, double *y, int nt)
{
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i < nt) y[i] = fast_exp(x[i]);
	else if (i < 2 * nt) y[i] = fast_log(x[i]);
}

// One pass over the list: K1 and K2 sums (SPATest.cpp:64,79-80) at t1 (root 1)
// and/or t2 (root 2).  a1/a2 are wave-uniform.
template <int BLOCK>
__device__ __forceinline__ void cgf_pass2(bool a1, bool a2, double t1, double t2, int nnz,
	const double *__restrict__ gl, const double *__restrict__ ml, double *sh, double (&o)[4])
{
	double v[4] = {0, 0, 0, 0};
	for (int k = threadIdx.x; k < nnz; k += BLOCK) {
		const double g = gl[k], m = ml[k], om = 1 - m;
		const double mg = m * g, c2 = om * mg * g;
		if (a1) {
			const double e = exp(-g * t1);
			const double d = fma(om, e, m);
			const double r = isfinite(d) ? fast_rcp(d) : 0.0;
			v[0] = fma(mg, r, v[0]);
			const double t = c2 * e * r * r;
			if (isfinite(t)) v[1] += t;
		}
		if (a2) {
			const double e = exp(-g * t2);
			const double d = fma(om, e, m);
			const double r = isfinite(d) ? fast_rcp(d) : 0.0;
			v[2] = fma(mg, r, v[2]);
			const double t = c2 * e * r * r;
			if (isfinite(t)) v[3] += t;
		}
	}
	block_sum<4, BLOCK>(v, sh);
#pragma unroll
	for (int a = 0; a < 4; a++) o[a] = v[a];
}

// Korg sums (SPATest.cpp:49) at both roots
template <int BLOCK>
__device__ __forceinline__ void korg_pass2(bool a1, bool a2, double t1, double t2, int nnz,
	const double *__restrict__ gl, const double *__restrict__ ml, double *sh, double (&o)[2])
{
	double v[2] = {0, 0};
	for (int k = threadIdx.x; k < nnz; k += BLOCK) {
		const double g = gl[k], m = ml[k], om = 1 - m;
		if (a1) v[0] += log(fma(m, exp(g * t1), om));
		if (a2) v[1] += log(fma(m, exp(g * t2), om));
	}
	block_sum<2, BLOCK>(v, sh);
	o[0] = v[0]; o[1] = v[1];
}

// tail of get_saddle_prob_fast, SPATest.cpp:216-229, Korg/K2 sums given
__device__ __forceinline__ double lugannani_rice(double t, double Ksum, double k2s, double q,
	double NAmu, double NAsigma)
{
	const double K = Ksum + NAmu * t + 0.5 * NAsigma * t * t;
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

#define SPA2_QCAP 16   /* carriers a thread can queue per chunk = samples per dword */

template <int K, int BLOCK>
__global__ void __launch_bounds__(BLOCK)
spa2_kernel(const uint8_t *__restrict__ packed, size_t bpv, DevModel md,
	const SpaRec *__restrict__ recs, int *__restrict__ counters, int counter_slot,
	const int *__restrict__ rec_index, int *__restrict__ fallback,
	double *__restrict__ scratch, size_t scratch_stride, double *__restrict__ out8)
{
	constexpr int NW = BLOCK / WAVE;
	constexpr int KP = (K + 2) & ~1;           // row of XM: X_i (K), mu_i, pad
	__shared__ double sh[8 * NW];
	__shared__ int shi[NW];
	__shared__ uint32_t qidx[BLOCK * SPA2_QCAP];
	const int N = md.N, tid = threadIdx.x;
	const int lane = tid & (WAVE - 1), wid = tid / WAVE;
	const int nflag = counters[counter_slot];
	double *gl = scratch + (size_t)blockIdx.x * scratch_stride;
	double *ml = gl + scratch_stride / 2;
	const int ndw = (N + 15) >> 4;

	for (int vi = blockIdx.x; vi < nflag; vi += gridDim.x) {
		const int v = rec_index ? rec_index[vi] : vi;
		const SpaRec r = recs[v];
		const uint32_t *row = reinterpret_cast<const uint32_t *>(packed + (size_t)r.j * bpv);
		const double inv = 1 / sqrt(r.AC2);
		const uint32_t zx = r.minus ? 0xAAAAAAAAu : 0u;
		double c[K];
#pragma unroll
		for (int a = 0; a < K; a++) c[a] = r.c[a];

		// ---- carriers: (adj, mu) list + carrier sums
		// a6: sum mu*G, sum b, sum max(adj,0), sum min(adj,0), sum adj*mu, sum adj^2 mu(1-mu)
		double a6[6] = {0, 0, 0, 0, 0, 0};
		int nnz = 0;
		for (int d0 = 0; d0 < ndw; d0 += BLOCK) {
			const int d = d0 + tid;
			const uint32_t w = (d < ndw) ? row[d] : 0u;
			uint32_t nz = nz_fields((w ^ zx) & keep_mask(N - d * 16));
			const int cnt = __popc(nz);
			int incl = cnt;                           // wave inclusive scan
#pragma unroll
			for (int o = 1; o < WAVE; o <<= 1) {
				const int up = __shfl_up(incl, o, WAVE);
				if (lane >= o) incl += up;
			}
			if (lane == WAVE - 1) shi[wid] = incl;
			__syncthreads();
			int wbase = 0, total = 0;
#pragma unroll
			for (int ww = 0; ww < NW; ww++) { if (ww < wid) wbase += shi[ww]; total += shi[ww]; }
			int off = wbase + incl - cnt;
			while (nz) {
				const int b = __ffs(nz) - 1;
				nz &= nz - 1;
				qidx[off++] = (uint32_t)(d * 16 + (b >> 1)) | (((w >> b) & 3u) << 30);
			}
			__syncthreads();
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
				gl[nnz + k] = adj; ml[nnz + k] = mui;
				a6[0] = fma(mui, G, a6[0]);
				a6[1] += b;
				if (adj > 0) a6[2] += adj; else a6[3] += adj;
				a6[4] = fma(adj, mui, a6[4]);
				a6[5] = fma(adj * adj, mui * (1 - mui), a6[5]);
			}
			nnz += total;
			__syncthreads();
		}
		block_sum<6, BLOCK>(a6, sh);     // barriers also publish the list

		// ---- scalars of saige_main.cpp:369-381
		double xmu_c = 0, xsum_c = 0;
#pragma unroll
		for (int a = 0; a < K; a++) { xmu_c = fma(md.Xmu[a], c[a], xmu_c); xsum_c = fma(md.Xsum[a], c[a], xsum_c); }
		const double m1 = (a6[0] - xmu_c) * inv;
		const double Tstat = r.S * inv;
		const double var2 = r.var2 / r.AC2;
		const double var1 = var2 * md.r;
		const double qtilde = Tstat / sqrt(var1) * sqrt(var2) + m1;
		// Saddle_Prob_Fast(qtilde, m1, var2, ...)
		const double s = qtilde - m1;
		const double qinv = -s + m1;
		const double pn_in = d_pchisq1_upper(s * s / var2);
		double pval;
		bool converged = true, need_fallback = false;
		if (fabs(qtilde - m1) / sqrt(var2) < 2.0) {
			pval = pn_in;
		} else {
			const double nb = (xsum_c - a6[1]) * inv;
			const double L = a6[2] + fmax(-nb, 0.0), U = a6[3] + fmin(-nb, 0.0);
			const double mar = 1e-9 * (fabs(L) + fabs(U) + fabs(qtilde) + fabs(qinv));
			if (!(qtilde < L - mar && qtilde > U + mar && qinv < L - mar && qinv > U + mar)) {
				need_fallback = true;
				pval = pn_in;
			} else {
				const double NAmu = m1 - a6[4], NAsigma = var2 - a6[5];
				RootState s1, s2;
				root_begin(s1, qtilde, L, U);
				root_begin(s2, qinv, L, U);
				while (s1.active || s2.active) {
					double o[4];
					cgf_pass2<BLOCK>(s1.active, s2.active, s1.tnew, s2.tnew, nnz, gl, ml, sh, o);
					if (s1.active) root_feed(s1, o[0], o[1], NAmu, NAsigma);
					if (s2.active) root_feed(s2, o[2], o[3], NAmu, NAsigma);
				}
				if (s1.converged && s2.converged) {
					double ko[2];
					korg_pass2<BLOCK>(true, true, s1.root, s2.root, nnz, gl, ml, sh, ko);
					const double p1 = lugannani_rice(s1.root, ko[0], s1.K2cur, qtilde, NAmu, NAsigma);
					const double p2 = lugannani_rice(s2.root, ko[1], s2.K2cur, qinv, NAmu, NAsigma);
					pval = fabs(p1) + fabs(p2);
					if (pval != 0 && pn_in / pval > 1000) pval = pn_in;   // SPATest.cpp:368-371
				} else {
					pval = pn_in;
					converged = false;
				}
			}
		}
		if (need_fallback) {
			if (tid == 0) fallback[atomicAdd(&counters[2], 1)] = v;
			continue;
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
