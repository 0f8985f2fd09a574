/*
 * saige_oracle.c -- CPU restatement of the SAIGEgds single-variant scan.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the parity oracle for the HIP path:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load it.  The product (saigegds_amd/, libsaigehip.so) never calls it.
 *
 * It restates, in plain C99 double precision and in the reference's own
 * evaluation order, the per-variant routine reached from
 * seqAssocGLMM_SPA() (reference file:line cited at every function):
 *     src/saige_main.cpp:162-186   get_ds
 *     src/saige_main.cpp:189-276   single_test_quant
 *     src/saige_main.cpp:279-407   single_test_bin
 *     src/SPATest.cpp:42-83        Korg / K1_adj / K2
 *     src/SPATest.cpp:139-184      getroot_K1_fast
 *     src/SPATest.cpp:211-230      get_saddle_prob_fast
 *     src/SPATest.cpp:299-374      Saddle_Prob_Fast
 *     src/vectorization.cpp:186-582  the f64_* primitives used above
 *
 * Pinning: the oracle reproduces inst/unitTests/saige_pval.rds and
 * saige_pval_quant.rds (10 000 variants each) from saige_model*.rds and
 * grm1k_10k_snp.gds -- see tests/test_oracle_golden.py.  Branches those
 * goldens do not reach (AF>0.5 flip, missing genotypes, root=Inf, bisection
 * safeguard, non-convergence) are "parity unpinned": the reference objects
 * need R headers this image lacks, so they cannot be built here.
 *
 * Third-party arithmetic restated (Rmath, not vendored in the reference):
 *   Rf_pchisq(x, 1, upper)  -> erfc(sqrt(x/2))
 *   Rf_pnorm5(z, 0, 1, ..)  -> 0.5*erfc(-+z/sqrt2)
 *   Rf_qnorm5(p, 0, 1, lower) -> Wichura AS241 PPND16 (the algorithm Rmath uses)
 *   Rf_sign(x)              -> -1/0/+1 (NaN stays NaN)
 */
#include <math.h>
#include <float.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "saige_oracle.h"

/* Accumulator type of every sum.  The oracle proper uses double (as the
 * reference); tests also build a long-double twin (liboracle_ld.so) to tell
 * the reference's own rounding noise from a real discrepancy. */
#ifndef ORC_ACC
#define ORC_ACC double
#endif

/* ------------------------------------------------------------------ */
/* Rmath stand-ins                                                      */

double orc_pchisq1_upper(double x)
{
	if (isnan(x)) return x;
	if (x <= 0) return 1.0;
	return erfc(sqrt(x * 0.5));
}

/* pnorm(z, 0, 1, lower_tail, log=FALSE) */
double orc_pnorm(double z, int lower)
{
	if (isnan(z)) return z;
	return lower ? 0.5 * erfc(-z * M_SQRT1_2) : 0.5 * erfc(z * M_SQRT1_2);
}

static double orc_sign(double x)
{
	if (isnan(x)) return x;
	return (x > 0) ? 1.0 : ((x == 0) ? 0.0 : -1.0);
}

/* qnorm(p, 0, 1, lower_tail=TRUE, log=FALSE): AS241 PPND16 */
double orc_qnorm(double p)
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

/* ------------------------------------------------------------------ */
/* vectorization.cpp primitives (plain loops, IEEE order)               */

/* vectorization.cpp:186-205 */
static void f64_af_ac_impute(double *ds, size_t n, double *AF, double *AC,
	int *Num, int *buf_idx)
{
	ORC_ACC sum = 0;
	int num = 0, *pIdx = buf_idx;
	for (size_t i = 0; i < n; i++) {
		if (isfinite(ds[i])) { sum += ds[i]; num++; }
		else *pIdx++ = (int)i;
	}
	*AF = (num > 0) ? (sum / (2 * num)) : NAN;
	*AC = sum; *Num = num;
	if (num < (int)n) {
		double d = *AF * 2;
		for (; buf_idx < pIdx; ) ds[*buf_idx++] = d;
	}
}

/* vectorization.cpp:209-215 */
static size_t f64_nonzero_index(size_t n, const double *x, int *idx)
{
	size_t n_i = 0;
	for (size_t j = 0; j < n; j++)
		if (x[j] != 0) idx[n_i++] = (int)j;
	return n_i;
}

/* vectorization.cpp:309-406: p = X(m x n, column = sample) * y, skipping y==0 */
static void f64_mul_mat_vec(size_t n, size_t m, const double *x, const double *y, double *p)
{
	ORC_ACC acc[64] = {0};
	for (size_t k = 0; k < n; k++, x += m) {
		double alpha = y[k];
		if (alpha != 0)
			for (size_t i = 0; i < m; i++) acc[i] += alpha * x[i];
	}
	for (size_t i = 0; i < m; i++) p[i] = (double)acc[i];
}

/* vectorization.cpp:410-499 */
static void f64_mul_mat_vec_sp(size_t n_idx, const int *idx, size_t m,
	const double *x, const double *y, double *p)
{
	ORC_ACC acc[64] = {0};
	for (size_t k = 0; k < n_idx; k++) {
		size_t i = (size_t)idx[k];
		double alpha = y[i];
		const double *xx = &x[m * i];
		for (size_t j = 0; j < m; j++) acc[j] += alpha * xx[j];
	}
	for (size_t j = 0; j < m; j++) p[j] = (double)acc[j];
}

/* vectorization.cpp:503-514 */
static void f64_mul_mat_vec_sub(size_t n, const int *idx, size_t m,
	const double *x, const double *y, double *p)
{
	for (size_t i = 0; i < n; i++) {
		size_t k = (size_t)idx[i];
		const double *xx = &x[m * k];
		ORC_ACC sum = 0;
		for (size_t j = 0; j < m; j++) sum += y[j] * xx[j];
		p[i] = sum;
	}
}

/* vectorization.cpp:518-567 */
static void f64_sub_mul_mat_vec(size_t n, size_t m, const double *x,
	const double *y, const double *z, double *p)
{
	for (size_t i = 0; i < n; i++, y += m) {
		ORC_ACC sum = 0;
		for (size_t j = 0; j < m; j++) sum += y[j] * z[j];
		p[i] = x[i] - sum;
	}
}

/* vectorization.cpp:571-582 */
static double f64_sum_mat_vec(size_t n, const double *x, const double *y)
{
	ORC_ACC sum = 0;
	for (size_t i = 0; i < n; i++) {
		const double *xx = &x[n * i], a = y[i];
		for (size_t j = 0; j < n; j++) sum += a * y[j] * xx[j];
	}
	return sum;
}

/* ------------------------------------------------------------------ */
/* SPATest.cpp                                                          */

/* SPATest.cpp:42-52 */
static double Korg(double t, size_t n_g, const double mu[], const double g[])
{
	ORC_ACC sum = 0;
	for (size_t i = 0; i < n_g; i++) {
		double m_i = mu[i];
		sum += log(1 - m_i + m_i * exp(g[i] * t));
	}
	return sum;
}

/* SPATest.cpp:56-67 */
static double K1_adj(double t, size_t n_g, const double mu[], const double g[], double q)
{
	ORC_ACC sum = 0;
	for (size_t i = 0; i < n_g; i++) {
		double m_i = mu[i], g_i = g[i];
		sum += m_i * g_i / ((1 - m_i) * exp(-g_i * t) + m_i);
	}
	return sum - q;
}

/* SPATest.cpp:71-83 */
static double K2(double t, size_t n_g, const double mu[], const double g[])
{
	ORC_ACC sum = 0;
	for (size_t i = 0; i < n_g; i++) {
		double m_i = mu[i], one_m_i = 1 - m_i;
		double g_i = g[i], exp_i = exp(-g_i * t);
		double d = one_m_i * exp_i + m_i;
		double v = (one_m_i * m_i * g_i * g_i * exp_i) / (d * d);
		if (isfinite(v)) sum += v;
	}
	return sum;
}

/* SPATest.cpp:139-184 */
static void getroot_K1_fast(double g_pos, double g_neg, double *root, int *n_iter_out,
	int *converged, double init, size_t n_nonzero, const double mu[],
	const double g[], double q, double NAmu, double NAsigma, orc_trace *tr)
{
	const double tol = sqrt(sqrt(DBL_EPSILON));  /* SPATest.cpp:87 */
	const int maxiter = 1000;                     /* SPATest.cpp:88 */
	if (q >= g_pos || q <= g_neg) {
		*root = INFINITY; *n_iter_out = 0; *converged = 1;
		if (tr) tr->root_inf++;
		return;
	}
	double t = init;
	*root = init;
	double K1_eval = K1_adj(t, n_nonzero, mu, g, q) + NAmu + NAsigma * t;
	double prevJump = INFINITY;
	*converged = 0;
	int it;
	for (it = 1; it <= maxiter; it++) {
		double K2_eval = K2(t, n_nonzero, mu, g) + NAsigma;
		double tnew = t - K1_eval / K2_eval;
		if (!isfinite(tnew)) break;
		if (fabs(tnew - t) < tol) { *converged = 1; break; }
		double newK1 = K1_adj(tnew, n_nonzero, mu, g, q) + NAmu + NAsigma * tnew;
		if (orc_sign(K1_eval) != orc_sign(newK1)) {
			if (fabs(tnew - t) > prevJump - tol) {
				tnew = t + orc_sign(newK1 - K1_eval) * prevJump * 0.5;
				newK1 = K1_adj(tnew, n_nonzero, mu, g, q) + NAmu + NAsigma * tnew;
				prevJump *= 0.5;
				if (tr) tr->bisect++;
			} else {
				prevJump = fabs(tnew - t);
			}
		}
		*root = t = tnew;
		K1_eval = newK1;
	}
	*n_iter_out = it;
	if (tr) { tr->newton_iters += it; if (!*converged) tr->not_converged++; }
}

/* SPATest.cpp:211-230 */
static double get_saddle_prob_fast(double t, size_t n_nonzero, const double mu[],
	const double g[], double q, double NAmu, double NAsigma)
{
	if (!isfinite(t)) return 0;
	double K  = Korg(t, n_nonzero, mu, g) + NAmu * t + 0.5 * NAsigma * t * t;
	double k2 = K2(t, n_nonzero, mu, g) + NAsigma;
	double pval = 0;
	if (isfinite(K) && isfinite(k2)) {
		double w = orc_sign(t) * sqrt(2 * (t * q - K));
		double v = t * sqrt(k2);
		double z = w + log(v / w) / w;
		if (z > 0)
			pval = orc_pnorm(z, 0);
		else
			pval = -orc_pnorm(z, 1);
	}
	return pval;
}

/* SPATest.cpp:299-374 */
double orc_saddle_prob_fast(double q, double m1, double var1, size_t n_g,
	const double mu[], const double g[], size_t n_nonzero,
	const int nonzero_idx[], double cutoff, int *converged, double buf_spa[],
	double *p_noadj, orc_trace *tr)
{
	double s = q - m1;
	double qinv = -s + m1;
	double pval_noadj = orc_pchisq1_upper(s * s / var1);
	double pval;
	ORC_ACC NAmu = 0, NAsigma = 0;
	ORC_ACC g_pos = 0, g_neg = 0;
	int init = 0;

	if (p_noadj) *p_noadj = pval_noadj;
	while (1) {
		*converged = 1;
		if (cutoff < 0.1) cutoff = 0.1;
		if (fabs(q - m1) / sqrt(var1) < cutoff) {
			pval = pval_noadj;
			if (tr) tr->cutoff_exit++;
		} else {
			if (!init) {
				init = 1;
				for (size_t i = 0; i < n_g; i++) {
					double v = g[i];
					if (v > 0) g_pos += v; else g_neg += v;
				}
				NAmu = m1; NAsigma = var1;
				for (size_t i = 0; i < n_nonzero; i++) {
					size_t k = (size_t)nonzero_idx[i];
					double g_k, mu_k;
					buf_spa[i] = g_k = g[k];
					buf_spa[i + n_nonzero] = mu_k = mu[k];
					NAmu -= g_k * mu_k;
					NAsigma -= g_k * g_k * mu_k * (1 - mu_k);
				}
				g = &buf_spa[0]; mu = &buf_spa[n_nonzero];
			}
			double root1, root2;
			int ni1, ni2, conv1, conv2;
			getroot_K1_fast(g_pos, g_neg, &root1, &ni1, &conv1, 0, n_nonzero,
				mu, g, q, NAmu, NAsigma, tr);
			getroot_K1_fast(g_pos, g_neg, &root2, &ni2, &conv2, 0, n_nonzero,
				mu, g, qinv, NAmu, NAsigma, tr);
			if (conv1 && conv2) {
				double p1 = get_saddle_prob_fast(root1, n_nonzero, mu, g, q, NAmu, NAsigma);
				double p2 = get_saddle_prob_fast(root2, n_nonzero, mu, g, qinv, NAmu, NAsigma);
				pval = fabs(p1) + fabs(p2);
				if (tr) tr->spa_done++;
			} else {
				pval = pval_noadj;
				*converged = 0;
				break;
			}
		}
		if (pval != 0 && pval_noadj / pval > 1000) {
			cutoff *= 2;
			if (tr) tr->cutoff_doubled++;
		} else
			break;
	}
	return pval;
}

/* ------------------------------------------------------------------ */
/* saige_main.cpp                                                       */

struct orc_model {
	int n, k;
	int quant;
	double thr_maf, thr_mac, thr_missing, thr_pval_spa;
	double tau[2], var_ratio;
	double *y, *mu, *y_mu, *mu2;
	double *t_XXVX_inv, *XV, *t_XVX_inv_XV, *t_X;  /* K x N, sample-major */
	double *XVX, *S_a;
	/* scratch (saige_main.cpp:82-89) */
	double *buf_dosage, *buf_coeff, *buf_adj_g, *buf_B, *buf_g_tilde, *buf_X1, *buf_spa;
	int *buf_index;
};

static double *dupd(const double *p, size_t n)
{
	double *r = (double *)malloc(sizeof(double) * (n ? n : 1));
	if (p) memcpy(r, p, sizeof(double) * n);
	return r;
}

/* saige_main.cpp:103-150 saige_score_test_init (arrays are copied here) */
orc_model *orc_model_new(int n, int k, int quant, const double *tau,
	const double *y, const double *mu, const double *y_mu, const double *mu2,
	const double *t_XXVX_inv, const double *XV, const double *t_XVX_inv_XV,
	const double *t_X, const double *XVX, const double *S_a, double var_ratio,
	double maf, double mac, double missing, double spa_pval)
{
	orc_model *m = (orc_model *)calloc(1, sizeof(orc_model));
	m->n = n; m->k = k; m->quant = quant;
	m->thr_maf = isfinite(maf) ? maf : -1;               /* :108-109 */
	m->thr_mac = isfinite(mac) ? mac : -1;               /* :110-111 */
	m->thr_missing = isfinite(missing) ? missing : 1;    /* :112-113 */
	m->thr_pval_spa = isfinite(spa_pval) ? spa_pval : 0.05;  /* :114-115 */
	m->tau[0] = tau[0]; m->tau[1] = tau[1];
	m->var_ratio = var_ratio;
	size_t N = (size_t)n, K = (size_t)k;
	m->y = dupd(y, N); m->mu = dupd(mu, N); m->y_mu = dupd(y_mu, N); m->mu2 = dupd(mu2, N);
	m->t_XXVX_inv = dupd(t_XXVX_inv, N * K); m->XV = dupd(XV, N * K);
	m->t_XVX_inv_XV = dupd(t_XVX_inv_XV, N * K); m->t_X = dupd(t_X, N * K);
	m->XVX = dupd(XVX, K * K); m->S_a = dupd(S_a, K);
	m->buf_dosage = dupd(NULL, N); m->buf_coeff = dupd(NULL, K);
	m->buf_adj_g = dupd(NULL, N); m->buf_B = dupd(NULL, N);
	m->buf_g_tilde = dupd(NULL, N); m->buf_X1 = dupd(NULL, K);
	m->buf_spa = dupd(NULL, 2 * N);
	m->buf_index = (int *)malloc(sizeof(int) * (N ? N : 1));
	return m;
}

void orc_model_free(orc_model *m)
{
	if (!m) return;
	free(m->y); free(m->mu); free(m->y_mu); free(m->mu2);
	free(m->t_XXVX_inv); free(m->XV); free(m->t_XVX_inv_XV); free(m->t_X);
	free(m->XVX); free(m->S_a);
	free(m->buf_dosage); free(m->buf_coeff); free(m->buf_adj_g); free(m->buf_B);
	free(m->buf_g_tilde); free(m->buf_X1); free(m->buf_spa); free(m->buf_index);
	free(m);
}

static inline double sq(double v) { return v * v; }

/* saige_main.cpp:189-276.  G is modified in place (impute, flip). */
static int single_test_quant(orc_model *M, double G[], double out[6])
{
	const size_t n = (size_t)M->n, K = (size_t)M->k;
	int *idx = M->buf_index;
	double AF, AC; int Num;
	f64_af_ac_impute(G, n, &AF, &AC, &Num, idx);
	const double maf = fmin(AF, 1 - AF);
	const double mac = fmin(AC, 2 * Num - AC);
	const double missing = (double)(n - (size_t)Num) / n;
	if (!((Num > 0) && (maf > 0) && (maf >= M->thr_maf) &&
		(mac >= M->thr_mac) && (missing <= M->thr_missing)))
		return 0;
	int minus = (AF > 0.5);
	if (minus) for (size_t i = 0; i < n; i++) G[i] = 2 - G[i];

	double pval, beta;
	const double inv_sqrt_mac = 1.0 / sqrt(mac);
	const double inv_mac = 1.0 / mac;
	if (maf < 0.05) {
		size_t nnz = f64_nonzero_index(n, G, idx);
		f64_mul_mat_vec_sp(nnz, idx, K, M->t_XVX_inv_XV, G, M->buf_coeff);
		f64_mul_mat_vec_sub(nnz, idx, K, M->t_X, M->buf_coeff, M->buf_B);
		for (size_t i = 0; i < nnz; i++) M->buf_g_tilde[i] = G[idx[i]] - M->buf_B[i];
		ORC_ACC var2 = f64_sum_mat_vec(K, M->XVX, M->buf_coeff);
		for (size_t i = 0; i < nnz; i++) var2 += sq(M->buf_g_tilde[i]) - sq(M->buf_B[i]);
		double var1 = var2 * inv_mac * M->var_ratio;
		ORC_ACC S1 = 0;
		for (size_t i = 0; i < nnz; i++) S1 += M->y_mu[idx[i]] * M->buf_g_tilde[i];
		f64_mul_mat_vec_sp(nnz, idx, K, M->t_X, M->y_mu, M->buf_X1);
		ORC_ACC S2 = 0;
		for (size_t i = 0; i < K; i++) S2 += (M->buf_X1[i] - M->S_a[i]) * M->buf_coeff[i];
		double Tstat = (S1 + S2) * inv_sqrt_mac / M->tau[0];
		pval = orc_pchisq1_upper(Tstat * Tstat / var1);
		beta = Tstat / var1 * inv_sqrt_mac;
	} else {
		f64_mul_mat_vec(n, K, M->XV, G, M->buf_coeff);
		f64_sub_mul_mat_vec(n, K, G, M->t_XXVX_inv, M->buf_coeff, M->buf_adj_g);
		ORC_ACC S = 0, var = 0;  /* f64_dot_sp, vectorization.cpp:281-291 */
		for (size_t i = 0; i < n; i++) {
			S += M->y_mu[i] * M->buf_adj_g[i];
			var += M->buf_adj_g[i] * M->buf_adj_g[i];
		}
		double Tstat = S * inv_sqrt_mac / M->tau[0];
		var *= inv_mac * M->var_ratio;
		pval = orc_pchisq1_upper(Tstat * Tstat / var);
		beta = Tstat / var * inv_sqrt_mac;
	}
	if (minus) beta = -beta;
	double SE = fabs(beta / orc_qnorm(pval / 2));
	out[0] = AF; out[1] = mac; out[2] = Num; out[3] = beta; out[4] = SE; out[5] = pval;
	return 1;
}

/* saige_main.cpp:279-407.  G is modified in place (impute, flip). */
static int single_test_bin(orc_model *M, double G[], double out[8], orc_trace *tr)
{
	const size_t n = (size_t)M->n, K = (size_t)M->k;
	int *idx = M->buf_index;
	double AF, AC; int Num;
	f64_af_ac_impute(G, n, &AF, &AC, &Num, idx);
	const double maf = fmin(AF, 1 - AF);
	const double mac = fmin(AC, 2 * Num - AC);
	const double missing = (double)(n - (size_t)Num) / n;
	if (!((Num > 0) && (maf > 0) && (maf >= M->thr_maf) &&
		(mac >= M->thr_mac) && (missing <= M->thr_missing)))
		return 0;
	int minus = (AF > 0.5);
	if (minus) for (size_t i = 0; i < n; i++) G[i] = 2 - G[i];

	double pval_noadj, beta;
	size_t nnz = 0;
	const int is_sparse = maf < 0.05;
	if (is_sparse) {
		nnz = f64_nonzero_index(n, G, idx);
		f64_mul_mat_vec_sp(nnz, idx, K, M->t_XVX_inv_XV, G, M->buf_coeff);
		f64_mul_mat_vec_sub(nnz, idx, K, M->t_X, M->buf_coeff, M->buf_B);
		for (size_t i = 0; i < nnz; i++) M->buf_g_tilde[i] = G[idx[i]] - M->buf_B[i];
		ORC_ACC var2 = f64_sum_mat_vec(K, M->XVX, M->buf_coeff);
		for (size_t i = 0; i < nnz; i++)
			var2 += (sq(M->buf_g_tilde[i]) - sq(M->buf_B[i])) * M->mu2[idx[i]];
		double var1 = var2 * M->var_ratio;
		ORC_ACC S1 = 0;
		for (size_t i = 0; i < nnz; i++) S1 += M->y_mu[idx[i]] * M->buf_g_tilde[i];
		f64_mul_mat_vec_sp(nnz, idx, K, M->t_X, M->y_mu, M->buf_X1);
		ORC_ACC S2 = 0;
		for (size_t i = 0; i < K; i++) S2 += (M->buf_X1[i] - M->S_a[i]) * M->buf_coeff[i];
		double S = S1 + S2;
		pval_noadj = orc_pchisq1_upper(S * S / var1);
		beta = S / var1;
		if (tr) tr->sparse_path++;
	} else {
		f64_mul_mat_vec(n, K, M->XV, G, M->buf_coeff);
		f64_sub_mul_mat_vec(n, K, G, M->t_XXVX_inv, M->buf_coeff, M->buf_adj_g);
		ORC_ACC S = 0, var = 0;  /* f64_dot_sp2, vectorization.cpp:295-305 */
		for (size_t i = 0; i < n; i++) {
			S += M->y_mu[i] * M->buf_adj_g[i];
			var += M->mu2[i] * M->buf_adj_g[i] * M->buf_adj_g[i];
		}
		var *= M->var_ratio;
		pval_noadj = orc_pchisq1_upper(S * S / var);
		beta = S / var;
		if (tr) tr->dense_path++;
	}

	double pval = pval_noadj;
	int converged = isfinite(pval_noadj) != 0;
	if (converged && (pval_noadj <= M->thr_pval_spa)) {
		if (tr) tr->spa_entered++;
		if (is_sparse) {
			f64_mul_mat_vec_sp(nnz, idx, K, M->XV, G, M->buf_coeff);
			f64_sub_mul_mat_vec(n, K, G, M->t_XXVX_inv, M->buf_coeff, M->buf_adj_g);
		}
		double AC2 = minus ? (2 * Num - AC) : AC;
		double sc = 1 / sqrt(AC2);
		for (size_t i = 0; i < n; i++) M->buf_adj_g[i] *= sc;       /* f64_mul */
		ORC_ACC q = 0;
		for (size_t i = 0; i < n; i++) q += M->y[i] * M->buf_adj_g[i];  /* f64_dot */
		ORC_ACC m1 = 0, var2 = 0;
		for (size_t i = 0; i < n; i++) {                              /* f64_dot_sp2 */
			m1 += M->mu[i] * M->buf_adj_g[i];
			var2 += M->mu2[i] * M->buf_adj_g[i] * M->buf_adj_g[i];
		}
		double var1 = var2 * M->var_ratio;
		double Tstat = q - m1;
		double qtilde = Tstat / sqrt(var1) * sqrt(var2) + m1;
		if (!is_sparse) nnz = f64_nonzero_index(n, G, idx);
		pval = orc_saddle_prob_fast(qtilde, m1, var2, n, M->mu, M->buf_adj_g,
			nnz, idx, 2, &converged, M->buf_spa, NULL, tr);
		if (pval == 0 && pval_noadj > 0) { pval = pval_noadj; converged = 0; }
		beta = (Tstat / var1) / sqrt(AC2);
	}
	if (minus) beta = -beta;
	double SE = fabs(beta / orc_qnorm(pval / 2));
	out[0] = AF; out[1] = mac; out[2] = Num; out[3] = beta; out[4] = SE;
	out[5] = pval; out[6] = pval_noadj; out[7] = converged ? 1 : 0;
	if (tr && minus) tr->flipped++;
	return 1;
}

/* ------------------------------------------------------------------ */
/* drivers: one call per variant block                                  */

static void nan_row(double *o, int ncol) { for (int c = 0; c < ncol; c++) o[c] = NAN; }

static int run_one(orc_model *M, double *G, double *o, orc_trace *tr)
{
	if (M->quant) {
		double r[6];
		int ok = single_test_quant(M, G, r);
		if (ok) { memcpy(o, r, sizeof r); o[6] = NAN; o[7] = NAN; }
		return ok;
	}
	return single_test_bin(M, G, o, tr);
}

/* REALSXP branch of get_ds (saige_main.cpp:173-174): doubles, NaN = missing */
int orc_scan_f64(orc_model *M, const double *dosage, size_t n_variants,
	double *out8, uint8_t *valid, orc_trace *tr)
{
	const size_t n = (size_t)M->n;
	for (size_t j = 0; j < n_variants; j++) {
		memcpy(M->buf_dosage, dosage + j * n, sizeof(double) * n);
		nan_row(out8 + 8 * j, 8);
		valid[j] = (uint8_t)run_one(M, M->buf_dosage, out8 + 8 * j, tr);
	}
	return 0;
}

/* RAWSXP branch of get_ds (saige_main.cpp:179-182): bytes, 0xFF = missing */
int orc_scan_u8(orc_model *M, const uint8_t *dosage, size_t n_variants,
	double *out8, uint8_t *valid, orc_trace *tr)
{
	const size_t n = (size_t)M->n;
	for (size_t j = 0; j < n_variants; j++) {
		const uint8_t *p = dosage + j * n;
		for (size_t i = 0; i < n; i++)
			M->buf_dosage[i] = (p[i] != 0xFF) ? (double)p[i] : NAN;
		nan_row(out8 + 8 * j, 8);
		valid[j] = (uint8_t)run_one(M, M->buf_dosage, out8 + 8 * j, tr);
	}
	return 0;
}

/* 2-bit packed rows (4 samples per byte, LSB first; code 3 = missing): the
 * block format of the HIP boundary, decoded to what seqApply(.useraw=NA)
 * would hand to get_ds as RAW 0/1/2/0xFF. */
int orc_scan_2bit(orc_model *M, const uint8_t *packed, size_t bytes_per_variant,
	size_t n_variants, double *out8, uint8_t *valid, orc_trace *tr)
{
	const size_t n = (size_t)M->n;
	for (size_t j = 0; j < n_variants; j++) {
		const uint8_t *p = packed + j * bytes_per_variant;
		for (size_t i = 0; i < n; i++) {
			unsigned c = (p[i >> 2] >> (2 * (i & 3))) & 3u;
			M->buf_dosage[i] = (c != 3u) ? (double)c : NAN;
		}
		nan_row(out8 + 8 * j, 8);
		valid[j] = (uint8_t)run_one(M, M->buf_dosage, out8 + 8 * j, tr);
	}
	return 0;
}
