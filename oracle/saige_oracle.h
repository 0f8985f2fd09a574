/* saige_oracle.h -- interface of the CPU parity oracle (test infrastructure;
 * see the header of saige_oracle.c for scope, citations and pinning). */
#ifndef SAIGE_ORACLE_H
#define SAIGE_ORACLE_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_model orc_model;

/* branch counters, so tests can show which reference branches a case reached */
typedef struct orc_trace {
	long sparse_path, dense_path, flipped, spa_entered, cutoff_exit, spa_done;
	long cutoff_doubled, root_inf, bisect, not_converged, newton_iters;
} orc_trace;

double orc_pchisq1_upper(double x);
double orc_pnorm(double z, int lower);
double orc_qnorm(double p);

orc_model *orc_model_new(int n, int k, int quant, const double *tau,
	const double *y, const double *mu, const double *y_mu, const double *mu2,
	const double *t_XXVX_inv, const double *XV, const double *t_XVX_inv_XV,
	const double *t_X, const double *XVX, const double *S_a, double var_ratio,
	double maf, double mac, double missing, double spa_pval);
void orc_model_free(orc_model *m);

double orc_saddle_prob_fast(double q, double m1, double var1, size_t n_g,
	const double mu[], const double g[], size_t n_nonzero,
	const int nonzero_idx[], double cutoff, int *converged, double buf_spa[],
	double *p_noadj, orc_trace *tr);

/* out8: n_variants x 8 doubles [AF, mac, num, beta, SE, pval, pval_noadj,
 * converged] (quantitative: last two NaN); valid[j]=0 when the variant is
 * filtered (the reference returns NULL), its row is NaN. */
int orc_scan_f64(orc_model *M, const double *dosage, size_t n_variants,
	double *out8, uint8_t *valid, orc_trace *tr);
int orc_scan_u8(orc_model *M, const uint8_t *dosage, size_t n_variants,
	double *out8, uint8_t *valid, orc_trace *tr);
int orc_scan_2bit(orc_model *M, const uint8_t *packed, size_t bytes_per_variant,
	size_t n_variants, double *out8, uint8_t *valid, orc_trace *tr);

/* implicit-GRM operator of the null-model fit (grm_oracle.c) */
typedef struct orc_grm orc_grm;
orc_grm *orc_grm_new(const uint8_t *packed, size_t bpv, int n_samp, size_t n_markers);
void orc_grm_free(orc_grm *G);
void orc_grm_diag(const orc_grm *G, double *out);
void orc_grm_crossprod(orc_grm *G, const double *b, double *out);
int orc_grm_pcg(orc_grm *G, const double *w, const double *tau, const double *b,
	int maxiter, double tol, double *x);

#ifdef __cplusplus
}
#endif
#endif
