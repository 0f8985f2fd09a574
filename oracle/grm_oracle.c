/*
 * grm_oracle.c -- CPU restatement of the implicit-GRM operator of the
 * null-model fit (TEST INFRASTRUCTURE ONLY; see saige_oracle.c for the rules).
 *
 * Follows, single-threaded and in plain C, reference src/saige_fitnull.cpp:
 *     saige_store_2b_geno   :159-230  (std-genotype table :181-203, diag(GRM) :205-227)
 *     get_crossprod_b_grm   :435-536  (dense packed branch :486-519, normalisation :523-535)
 *     get_diag_sigma        :542-559
 *     get_crossprod         :564-576
 *     PCG_diag_sigma        :581-614
 * Parity unpinned: the reference's only test of this code is the whole-model
 * comparison test.saige_fit_null_model (tolerance 1e-4, needs R's RNG stream),
 * which cannot run here; the GPU operator is compared with this restatement.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "saige_oracle.h"

struct orc_grm {
	size_t n, m, bpv;
	uint8_t *g;        /* m rows of bpv bytes */
	double *lut;       /* 4 per marker */
	double *diag;      /* n */
	double *buf;       /* n */
};

static inline double sq(double v) { return v * v; }

orc_grm *orc_grm_new(const uint8_t *packed, size_t bpv, int n_samp, size_t n_markers)
{
	orc_grm *G = (orc_grm *)calloc(1, sizeof(orc_grm));
	G->n = (size_t)n_samp; G->m = n_markers; G->bpv = bpv;
	G->g = (uint8_t *)malloc(bpv * n_markers);
	memcpy(G->g, packed, bpv * n_markers);
	G->lut = (double *)malloc(sizeof(double) * 4 * n_markers);
	G->diag = (double *)calloc(G->n, sizeof(double));
	G->buf = (double *)malloc(sizeof(double) * G->n);
	for (size_t i = 0; i < n_markers; i++) {       /* :181-203 */
		const uint8_t *g = G->g + bpv * i;
		int n_valid = 0, sum = 0;
		for (size_t j = 0; j < G->n; j++) {         /* only real samples are counted */
			unsigned c = (g[j >> 2] >> (2 * (j & 3))) & 3u;
			if (c < 3) { n_valid++; sum += (int)c; }
		}
		double af = (double)sum / (2 * n_valid);
		double inv = 1 / sqrt(2 * af * (1 - af));
		if (!isfinite(af) || !isfinite(inv)) af = inv = 0;
		double *p = &G->lut[4 * i];
		p[0] = (0 - 2 * af) * inv; p[1] = (1 - 2 * af) * inv;
		p[2] = (2 - 2 * af) * inv; p[3] = 0;
	}
	for (size_t i = 0; i < n_markers; i++) {       /* :207-226 */
		const uint8_t *g = G->g + bpv * i;
		const double *base = G->lut + 4 * i;
		for (size_t j = 0; j < G->n; j++)
			G->diag[j] += sq(base[(g[j >> 2] >> (2 * (j & 3))) & 3u]);
	}
	for (size_t j = 0; j < G->n; j++) G->diag[j] *= 1.0 / (double)n_markers;   /* :227 */
	return G;
}

void orc_grm_free(orc_grm *G)
{
	if (!G) return;
	free(G->g); free(G->lut); free(G->diag); free(G->buf); free(G);
}

void orc_grm_diag(const orc_grm *G, double *out) { memcpy(out, G->diag, sizeof(double) * G->n); }

/* :435-536, dense packed branch */
void orc_grm_crossprod(orc_grm *G, const double *b, double *out)
{
	memset(out, 0, sizeof(double) * G->n);
	for (size_t i = 0; i < G->m; i++) {
		const uint8_t *g = G->g + G->bpv * i;
		const double *base = G->lut + 4 * i;
		double dot = 0;                                   /* :492-505 */
		for (size_t j = 0; j < G->n; j++)
			dot += base[(g[j >> 2] >> (2 * (j & 3))) & 3u] * b[j];
		for (size_t j = 0; j < G->n; j++)                 /* :507-519 */
			out[j] += dot * base[(g[j >> 2] >> (2 * (j & 3))) & 3u];
	}
	for (size_t j = 0; j < G->n; j++) out[j] *= 1.0 / (double)G->m;   /* :533 */
}

/* :564-576 */
static void get_crossprod(orc_grm *G, const double *b, const double *w, const double *tau, double *out)
{
	if (tau[1] == 0) {
		for (size_t j = 0; j < G->n; j++) out[j] = tau[0] * (b[j] * (1 / w[j]));
	} else {
		orc_grm_crossprod(G, b, G->buf);
		for (size_t j = 0; j < G->n; j++) out[j] = tau[0] * (b[j] * (1 / w[j])) + tau[1] * G->buf[j];
	}
}

/* :581-614; returns the number of iterations */
int orc_grm_pcg(orc_grm *G, const double *w, const double *tau, const double *b,
	int maxiter, double tol, double *x)
{
	const size_t n = G->n;
	double *r = (double *)malloc(sizeof(double) * n * 5);
	double *minv = r + n, *z = r + 2 * n, *p = r + 3 * n, *Ap = r + 4 * n;
	for (size_t j = 0; j < n; j++) {                    /* get_diag_sigma :542-559 */
		double v = tau[0] / w[j] + tau[1] * G->diag[j];
		if (v < 1e-4) v = 1e-4;
		minv[j] = 1 / v;
		r[j] = b[j]; z[j] = minv[j] * r[j]; p[j] = z[j]; x[j] = 0;
	}
	int iter = 0;
	for (;;) {
		double rr = 0;
		for (size_t j = 0; j < n; j++) rr += r[j] * r[j];
		if (!(iter < maxiter && rr > tol)) break;
		iter++;
		get_crossprod(G, p, w, tau, Ap);
		double rz = 0, pAp = 0;
		for (size_t j = 0; j < n; j++) { rz += r[j] * z[j]; pAp += p[j] * Ap[j]; }
		const double a = rz / pAp;
		double rz1 = 0;
		for (size_t j = 0; j < n; j++) {
			x[j] += a * p[j];
			r[j] -= a * Ap[j];
			z[j] = minv[j] * r[j];
			rz1 += z[j] * r[j];
		}
		const double bet = rz1 / rz;
		for (size_t j = 0; j < n; j++) p[j] = z[j] + bet * p[j];
	}
	free(r);
	return iter;
}
