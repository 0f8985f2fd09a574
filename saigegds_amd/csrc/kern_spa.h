// kern_spa.h -- SPA stage v1: dense pass + compaction + Newton (fallback path and dosage inputs)
// Part of libsaigehip.so (single translation unit: saigehip.hip).
#pragma once

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

// dosage of sample i of row r.j (2-bit rows: either layout; dosage rows: row-major)
template <int INPUT>
__device__ __forceinline__ double load_dosage(const RowsRef &rr, int i, const SpaRec &r)
{
	const uint8_t *row = rr.base + (INPUT == IN_2BIT ? 0 : (size_t)r.j * rr.bpv);
	if (INPUT == IN_2BIT) {
		const uint8_t b = rr.base[rr_piece(rr, (size_t)r.j, (size_t)(i >> 6)) + ((i & 63) >> 2)];
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
		const double mg = m * g, e = fast_exp(-g * t);
		const double d = fma(om, e, m);
		const double rr = isfinite(d) ? fast_rcp(d) : 0.0;
		v[0] = fma(mg, rr, v[0]);                // SPATest.cpp:64
		const double t2 = om * mg * g * e * rr * rr;   // :79
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
		v[0] += fast_log(fma(m, fast_exp(g * t), 1 - m));   // SPATest.cpp:49
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
spa_kernel(RowsRef rr, DevModel md,
	const SpaRec *__restrict__ recs, const int *__restrict__ counters, int counter_slot,
	const int *__restrict__ rec_index, double *__restrict__ scratch, size_t scratch_stride,
	double *__restrict__ out8)
{
	constexpr int NW = BLOCK / WAVE;
	__shared__ double sh[8 * NW];
	__shared__ int shc[NW];
	const int N = md.N, tid = threadIdx.x;
	const int lane = tid & (WAVE - 1), wid = tid / WAVE;
	const int nflag = counters[counter_slot];
	double *gl = scratch + (size_t)blockIdx.x * scratch_stride;
	double *ml = gl + scratch_stride / 2;
	// contiguous sample segment per wave, multiple of 64
	const int seg = (((N + NW - 1) / NW) + 63) & ~63;
	const int s0 = wid * seg, s1 = min(N, s0 + seg);

	for (int v = blockIdx.x; v < nflag; v += gridDim.x) {
		const SpaRec r = recs[rec_index ? rec_index[v] : v];
		const double inv = 1 / sqrt(r.AC2);
		double c[K];
#pragma unroll
		for (int a = 0; a < K; a++) c[a] = r.c[a];

		// ---- A1: carriers per wave segment -> list offsets
		int cnt = 0;
		for (int i = s0 + lane; i < s1; i += WAVE) cnt += (load_dosage<INPUT>(rr, i, r) != 0);
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
				G = load_dosage<INPUT>(rr, i, r);
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
