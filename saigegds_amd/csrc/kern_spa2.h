// kern_spa2.h -- pieces shared by the SPA kernels: the carrier-only restatement of the scalars,
// getroot_K1_fast as a state machine, the CGF terms, Lugannani-Rice, the output row.
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
	// wanted with it (a wrong guess only costs time: one more sweep covers the rest)
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


// samples per segment of the SPA stage for K covariates: rows of (K + 2) & ~1 doubles, a power of two
// of them in SPA_TAB_KB KiB of LDS
#ifndef SPA_TAB_KB
#define SPA_TAB_KB 128
#endif
__host__ __device__ constexpr int spa_seg(int K)
{
	const int row = ((K + 2) & ~1) * 8;
	const int full = row <= 32 ? 4096 : row <= 64 ? 2048 : row <= 128 ? 1024 : 512;      // 128 KiB
	return full * SPA_TAB_KB / 128;
}

