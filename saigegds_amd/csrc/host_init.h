// host_init.h -- library / model lifetime: sgx_init (the fixed-point limb tiles), thresholds, layout, sgx_free, per-call workspace.
// Part of libsaigehip.so: included by saigehip.hip (one translation unit), not a header of its own.
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
	return (size_t)((n_samp + 511) / 512) * 128;  // whole pairs of 256-sample tiles = whole 128-B lines (kern_score_mfma.h)
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
	for (sgx_handle *t : h->twins) if (t) { int rc = sgx_set_thresholds(t, maf, mac, missing, spa_pval); if (rc) return rc; }
	return SGX_OK;
}

template <typename T>
static int dev_upload(T **dst, const std::vector<T> &src)
{
	HIPCHK(hipMalloc((void **)dst, src.size() * sizeof(T)));
	HIPCHK(hipMemcpy(*dst, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice));
	return SGX_OK;
}

// XVXi with  t_XVX_inv_XV[i,:] = X[i,:] XVXi  for a quantitative model (weights 1; reference
// R/assoc_single.r:33-41 builds t_XVX_inv_XV from the same inverse).  Start from the inverse of
// XVX = X'X, then correct it by the least-squares residual against the model's own matrix, so that
// c' = XVXi e matches the reference's sum_i G_i t_XVX_inv_XV[i,:] to rounding even when X'X is badly
// conditioned.  False (-> the c' columns are carried as for binary traits) if the model's matrix is
// not such an image.
static bool fit_xvx_inverse(const sgx_model *m, double *out)
{
	const int N = m->n_samp, K = m->n_coeff;
	typedef long double LD;
	auto invert = [&](std::vector<LD> a, std::vector<LD> &inv) -> bool {   // Gauss-Jordan, partial pivoting
		inv.assign((size_t)K * K, 0);
		for (int i = 0; i < K; i++) inv[(size_t)i * K + i] = 1;
		for (int c = 0; c < K; c++) {
			int pv = c;
			for (int r = c + 1; r < K; r++) if (fabsl(a[(size_t)r * K + c]) > fabsl(a[(size_t)pv * K + c])) pv = r;
			if (!(fabsl(a[(size_t)pv * K + c]) > 0)) return false;
			for (int x = 0; x < K; x++) { std::swap(a[(size_t)c * K + x], a[(size_t)pv * K + x]); std::swap(inv[(size_t)c * K + x], inv[(size_t)pv * K + x]); }
			const LD d = 1 / a[(size_t)c * K + c];
			for (int x = 0; x < K; x++) { a[(size_t)c * K + x] *= d; inv[(size_t)c * K + x] *= d; }
			for (int r = 0; r < K; r++) if (r != c) {
				const LD f = a[(size_t)r * K + c];
				if (f == 0) continue;
				for (int x = 0; x < K; x++) { a[(size_t)r * K + x] -= f * a[(size_t)c * K + x]; inv[(size_t)r * K + x] -= f * inv[(size_t)c * K + x]; }
			}
		}
		return true;
	};
	std::vector<LD> A((size_t)K * K), M0, G((size_t)K * K, 0), Gi, B((size_t)K * K, 0), M((size_t)K * K);
	for (int a = 0; a < K * K; a++) A[a] = m->XVX[a];
	if (!invert(A, M0)) return false;
	for (int i = 0; i < N; i++) {        // R = t_XVX_inv_XV - X M0;  G = X'X;  B = X'R
		const double *x = m->t_X + (size_t)i * K;
		LD r[SGX_MAX_COEFF];
		for (int k = 0; k < K; k++) {
			LD t = m->t_XVX_inv_XV[(size_t)i * K + k];
			for (int b = 0; b < K; b++) t -= (LD)x[b] * M0[(size_t)b * K + k];
			r[k] = t;
		}
		for (int a = 0; a < K; a++)
			for (int b = 0; b < K; b++) { G[(size_t)a * K + b] += (LD)x[a] * x[b]; B[(size_t)a * K + b] += (LD)x[a] * r[b]; }
	}
	if (!invert(G, Gi)) return false;
	for (int a = 0; a < K; a++)
		for (int b = 0; b < K; b++) {
			LD d = 0;
			for (int x = 0; x < K; x++) d += Gi[(size_t)a * K + x] * B[(size_t)x * K + b];
			M[(size_t)a * K + b] = M0[(size_t)a * K + b] + d;
		}
	// the fit must reproduce the model's matrix to rounding
	LD worst = 0, scale = 0;
	for (int i = 0; i < N; i++) {
		const double *x = m->t_X + (size_t)i * K;
		for (int k = 0; k < K; k++) {
			LD t = 0;
			for (int b = 0; b < K; b++) t += (LD)x[b] * M[(size_t)b * K + k];
			worst = std::max(worst, fabsl(t - (LD)m->t_XVX_inv_XV[(size_t)i * K + k]));
			scale = std::max(scale, fabsl((LD)m->t_XVX_inv_XV[(size_t)i * K + k]));
		}
	}
	if (!(worst <= 1e-13L * scale)) return false;
	// c'_x = sum_y XVXi[x*K + y] e_y  with  c' = M' e
	for (int a = 0; a < K; a++)
		for (int b = 0; b < K; b++) out[(size_t)b * K + a] = (double)M[(size_t)a * K + b];
	for (int a = 0; a < K * K; a++) if (!std::isfinite(out[a])) return false;
	return true;
}

// stream, events, counters and the SPA buffers whose size does not depend on the call
static int alloc_workspace(sgx_handle *h)
{
	const int N = h->md.N;
	{
		// The step's critical path is the score chain (it streams the genotypes; the next step's chain cannot start
		// before this one's ends), the SPA stage hides under the other lane's chain: two priorities (round 4: kernel
		// traces showed the list pass stretched from 0.98 to 1.44 ms and 20-us solve kernels waiting 0.7 ms behind it)
		int least = 0, greatest = 0;
		HIPCHK(hipDeviceGetStreamPriorityRange(&least, &greatest));
		HIPCHK(hipStreamCreateWithPriority(&h->stream, hipStreamNonBlocking, least));
		HIPCHK(hipStreamCreateWithPriority(&h->hstream, hipStreamNonBlocking, greatest));
		HIPCHK(hipStreamCreateWithPriority(&h->s3_side, hipStreamNonBlocking, greatest));
	}
	HIPCHK(hipEventCreateWithFlags(&h->s3_fork, hipEventDisableTiming));
	HIPCHK(hipEventCreateWithFlags(&h->s3_join, hipEventDisableTiming));
	HIPCHK(hipMalloc((void **)&h->counters, 24 * sizeof(int)));
	HIPCHK(hipHostMalloc((void **)&h->h_counters, 24 * sizeof(int), hipHostMallocDefault));
	for (int i = 0; i < 3; i++) HIPCHK(hipEventCreate(&h->ev[i]));
	for (int i = 0; i < 2; i++) HIPCHK(hipEventCreate(&h->evk[i]));
	HIPCHK(hipEventCreate(&h->ev_lists));
	// SPA scratch: one (adj, mu) list of N entries per resident workgroup
	hipDeviceProp_t prop;
	HIPCHK(hipGetDeviceProperties(&prop, h->device));
	h->spa_grid = prop.multiProcessorCount * 2;
	h->n_cu = prop.multiProcessorCount;
	h->scratch_stride = 2 * (((size_t)N + 63) & ~(size_t)63);
	HIPCHK(hipMalloc((void **)&h->scratch, h->scratch_stride * sizeof(double) * h->spa_grid));
	if (!h->md.quant) {
		// workgroups of the per-variant kernels, each with its scratch lists: 4 per CU where a packed row is
		// short (128-thread workgroups, see the launch), else one
		h->nwg5 = ((size_t)((N + 63) / 64) * 16 <= 32 * 1024) ? h->n_cu * 4 : h->n_cu;
		HIPCHK(hipMalloc((void **)&h->scr5, (size_t)h->nwg5 * spa5_wg_bytes(N)));
		HIPCHK(hipMalloc((void **)&h->cur5, 8 * sizeof(int)));   // [0], [1] spa5_kernel queues; [2], [3] spa4_moments' item queue; [4], [5] spa5_kernel on the blocks' lists
	}
	return SGX_OK;
}

// Diagnostic of sgx_init (SAIGEHIP_CHECK_MODEL=1).  R/assoc_single.r:28-48 builds, per sample i with no-K weight
// V_i,  XV[:,i] = V_i t_X[:,i]  and  t_XVX_inv_XV[:,i] = V_i t_XXVX_inv[:,i]:  V_i is read off the largest
// entry of t_X[:,i] and both relations are held to 1e-8 of the column's largest entry.
static int check_model_consistency(const sgx_model *m)
{
	const int N = m->n_samp, K = m->n_coeff;
	if (!m->t_XXVX_inv || !m->XV)
		return fail(SGX_EINVAL, "sgx_init: SAIGEHIP_CHECK_MODEL needs t_XXVX_inv and XV");
	for (int i = 0; i < N; i++) {
		const double *x = m->t_X + (size_t)i * K, *xv = m->XV + (size_t)i * K,
			*a = m->t_XXVX_inv + (size_t)i * K, *av = m->t_XVX_inv_XV + (size_t)i * K;
		int k0 = 0;
		double sx = 0, sa = 0;
		for (int k = 0; k < K; k++) {
			if (std::fabs(x[k]) > std::fabs(x[k0])) k0 = k;
			sx = std::max(sx, std::fabs(xv[k])); sa = std::max(sa, std::fabs(av[k]));
		}
		if (x[k0] == 0) continue;
		const double V = xv[k0] / x[k0];
		for (int k = 0; k < K; k++) {
			if (std::fabs(xv[k] - V * x[k]) > 1e-8 * sx)
				return fail(SGX_EINVAL, "sgx_init: XV[%d,%d] = %g is not V t_X = %g", k, i, xv[k], V * x[k]);
			if (std::fabs(av[k] - V * a[k]) > 1e-8 * sa)
				return fail(SGX_EINVAL, "sgx_init: t_XVX_inv_XV[%d,%d] = %g is not V t_XXVX_inv = %g",
					k, i, av[k], V * a[k]);
		}
	}
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
	// t_XXVX_inv and XV are not read by the scan (the carrier formulation needs t_X, t_XVX_inv_XV, XVX and
	// S_a only).  SAIGEHIP_CHECK_MODEL=1 holds them against the arrays that ARE read, so that a
	// caller whose five K x N arrays do not belong together is told instead of getting one branch's algebra.
	{ const char *e = getenv("SAIGEHIP_CHECK_MODEL");
	  if (e && e[0] == '1') { int rc = check_model_consistency(m); if (rc) return rc; } }
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
	const int KP = (K + 2) & ~1;
	std::vector<double> F((size_t)N * P), X((size_t)N * K), y(m->y, m->y + N),
		mu(m->mu, m->mu + N), mu2(m->mu2, m->mu2 + N), XM((size_t)N * KP, 0.0);
	long double xmu[KMAX] = {0}, xsum[KMAX] = {0};
	for (int i = 0; i < N; i++) {
		const double w = quant ? 1.0 : m->mu2[i];   // quantitative: plain sums, saige_main.cpp:227-228
		double *f = &F[(size_t)i * P];
		for (int k = 0; k < K; k++) {
			f[k] = m->t_XVX_inv_XV[(size_t)i * K + k];
			f[K + k] = w * m->t_X[(size_t)i * K + k];
			X[(size_t)i * K + k] = m->t_X[(size_t)i * K + k];
			XM[(size_t)i * KP + k] = m->t_X[(size_t)i * K + k];
			xmu[k] += (long double)m->t_X[(size_t)i * K + k] * m->mu[i];
			xsum[k] += (long double)m->t_X[(size_t)i * K + k];
		}
		XM[(size_t)i * KP + K] = m->mu[i];
		f[2 * K] = m->y_mu[i];
		f[2 * K + 1] = w;
	}
	// Fixed-point limb tiles of the MFMA score path (kern_score3.h; kern_score_mfma.h "Limb counts"): ONE
	// group of up to 15 value fragments + the bit-1 fragment, and the table Q of the same values as int64
	// for the sparse pass over the missing genotypes.  Sample x at an odd position of its dword is used by
	// the kernel where it stands, two bits up (s3_scale): its value is a multiple of 4 and its digits carry
	// value / 4.
	std::vector<int8_t> Fl;
	std::vector<long long> Qt;
	if ((double)N * 4.0 * 384.0 < 2147483647.0) {
		MfEpi &ep = h->mfe;
		const int CS = 2 * K, CW = 2 * K + 1;       // s, and the column that carries G^2 (w)
		const int ngrp = (N + 15) / 16;
		const int ntile = 2 * ((ngrp + 31) / 32);   // whole 128-B lines of a row-major row
		const size_t ngrp_pad = (size_t)ntile * 16;
		// the columns: s, w first, then e, then c'.  Quantitative traits: the weights are 1
		// (saige_main.cpp:227-228), so w is the constant column (one limb) and t_XVX_inv_XV = X (X'X)^-1
		// makes c' a K x K image of e = sum G X: the c' columns are not carried, the epilogue forms
		// c' = XVXi e (XVXi fitted to the model's own t_XVX_inv_XV, fit_xvx_inverse).  Binary traits keep
		// them: there the two weight vectors (no-K V in t_XVX_inv_XV, GLMM mu2 in e) differ.
		ep.derive_c = quant && fit_xvx_inverse(m, ep.XVXi) ? 1 : 0;
		std::vector<int> order = {CS, CW};
		for (int k = 0; k < K; k++) order.push_back(K + k);
		if (!ep.derive_c) for (int k = 0; k < K; k++) order.push_back(k);
		for (int k = 0; k < K; k++) { ep.cgrp[k] = 0; ep.ccol[k] = 0; ep.climb[k] = 0; }
		// Limb counts follow the measured dynamic range of each column.  A column is quantised against
		// its largest entry, so an entry of typical size keeps 8 nl - 2 - log2(max / typical) bits (two
		// fewer at the odd positions): the reduced widths of kern_score_mfma.h "Limb counts" hold for
		// covariates whose largest value is a few times the typical one (max / mean|.| = 5.6 for a standard
		// normal column at N = 430 000) and are widened for heavy-tailed ones; beyond 2^22 no width is
		// enough and the model takes the FP64 gather kernels instead of the MFMA path.
		bool range_ok = true;
		auto limbs_for = [&](int c) -> int {
			if (c == CW && quant) return 1;
			long double sum = 0; double mx = 0;
			for (int i = 0; i < N; i++) { const double a = std::fabs(F[(size_t)i * P + c]); sum += a; mx = std::max(mx, a); }
			const double range = sum > 0 ? mx / (double)(sum / N) : 1.0;
			if (!(range <= 4194304.0)) range_ok = false;
			int nl = c >= 2 * K ? MF_NLIMB : (c >= K ? MF_LIMB_E : MF_LIMB_A);
			if (range > 64.0) nl = std::max(nl, MF_LIMB_E);
			if (range > 16384.0) nl = MF_NLIMB;
			// Small models: a sum over a handful of carriers does not average the quantisation away, and the
			// odd sample positions keep two bits fewer (s3_scale) -- one more limb costs nothing that matters
			// at these sizes (constructed inputs at N = 200: p-value 1.2e-10 off with the reduced widths)
			if (N < 16384) nl = std::min(MF_NLIMB, nl + 1);
			return nl;
		};
		int used = 1;                               // column 0 .. : values; the constant column last
		for (int c : order) {
			const int nl = limbs_for(c);
			ep.cgrp[c] = 0; ep.ccol[c] = (unsigned char)(used - 1); ep.climb[c] = (unsigned char)nl;
			used += nl;
		}
		if (range_ok) {
		const int nbfv = (used + 15) / 16;          // value fragments (<= 15: 2 K x 7 + 7 + 7 + 1 <= 239 columns)
		h->mf_nbfv[0] = nbfv;
		ep.ngroups = 1;
		ep.goff[0] = 0;
		ep.gncol[0] = 16 * (nbfv + 1);
		ep.acc_stride = ep.gncol[0];
		ep.col_ones = used - 1;                     // after the value columns
		ep.col_b1 = 16 * nbfv;
		h->mf[0].ntile = ntile;
		Fl.assign(ngrp_pad * ep.gncol[0] * 16, 0);
		Qt.assign((size_t)N * P, 0);
		auto at = [&](int i, int col) -> int8_t & {
			return Fl[((size_t)(i / 16) * ep.gncol[0] + col) * 16 + s3_pos(i % 16)];
		};
		for (int c = 0; c < P; c++) {
			const int cc = ep.ccol[c], nl = ep.climb[c];
			if (nl == 0) { ep.escale[c] = 0; ep.ftot_hi[c] = ep.ftot_lo[c] = 0; continue; }   // derived column
			double mx = 0;
			for (int i = 0; i < N; i++) mx = std::max(mx, std::fabs(F[(size_t)i * P + c]));
			int ex = 0;
			if (mx > 0) (void)std::frexp(mx, &ex);
			ep.escale[c] = (quant && c == CW) ? 2 : 8 * nl - 2 - ex;   // quantitative w = 1 exactly: value 4
			__int128 tot = 0;
			for (int i = 0; i < N; i++) {
				const int sc = s3_scale(i % 16);
				const long long d0 = std::llrint(std::ldexp(F[(size_t)i * P + c], ep.escale[c]) / sc);   // digits hold value / scale
				const long long q = d0 * sc;
				Qt[(size_t)i * P + c] = q;
				tot += q;
				long long rem = d0;
				for (int l = 0; l < nl; l++) {
					long long d = (l < nl - 1) ? (((rem + 128) & 255) - 128) : rem;
					rem = (rem - d) >> 8;
					at(i, cc + l) = (int8_t)d;
					if (c == CW) at(i, ep.col_b1 + l) = (int8_t)d;
				}
			}
			const __int128 two32 = ((__int128)1) << 32;
			__int128 hi = tot / two32, lo = tot - hi * two32;
			if (lo < 0) { lo += two32; hi -= 1; }
			ep.ftot_hi[c] = (long long)hi; ep.ftot_lo[c] = (long long)lo;
		}
		for (int i = 0; i < N; i++) {               // the constant column: 4 per allele at every position
			const int8_t d = (int8_t)(4 / s3_scale(i % 16));
			at(i, ep.col_ones) = d;
			at(i, ep.col_b1 + ep.climb[CW]) = d;
		}
		h->mf_ok = true;
		}
	}
	DevModel &md = h->md;
	md.N = N; md.K = K; md.P = P; md.quant = quant;
	md.tau0 = m->tau[0]; md.r = m->var_ratio;
	sgx_set_thresholds(h, m->maf, m->mac, m->missing, m->spa_pval);
	{
		// series SPA stage: a quarter of the smallest convergence radius sqrt(logit(mu)^2 + pi^2)
		// of log(1 - mu + mu e^x) over the model's fitted values (kern_spa4.h)
		double l2min = INFINITY;
		for (int i = 0; i < N; i++) {
			const double mi = m->mu[i];
			if (mi > 0 && mi < 1) { const double lg = std::log(mi / (1 - mi)); l2min = std::min(l2min, lg * lg); }
		}
		md.spa_xmax = std::isfinite(l2min) ? 0.25 * std::sqrt(l2min + M_PI * M_PI) : 0.0;
	}
	for (int k = 0; k < K; k++) {
		double mx = 0;
		for (int i = 0; i < N; i++) mx = std::max(mx, std::fabs(m->t_X[(size_t)i * K + k]));
		md.Xabs[k] = mx;
	}
	for (int a = 0; a < K * K; a++) md.XVX[a] = m->XVX[a];
	for (int a = 0; a < K; a++) { md.S_a[a] = m->S_a[a]; md.Xmu[a] = (double)xmu[a]; md.Xsum[a] = (double)xsum[a]; }
#define TRY(x) do { rc = (x); if (rc) { sgx_free(h); return rc; } } while (0)
	TRY(dev_upload(&h->dF, F));
	TRY(dev_upload(&h->dX, X));
	TRY(dev_upload(&h->dy, y));
	TRY(dev_upload(&h->dmu, mu));
	TRY(dev_upload(&h->dmu2, mu2));
	TRY(dev_upload(&h->dXM, XM));
	if (h->mf_ok) {
		std::vector<uint8_t> Flu(Fl.begin(), Fl.end());
		TRY(dev_upload(&h->dFl, Flu));
		h->mf[0].Fl = h->dFl;
		TRY(dev_upload(&h->dQ, Qt));
	}
	md.F = h->dF; md.X = h->dX; md.y = h->dy; md.mu = h->dmu; md.mu2 = h->dmu2; md.XM = h->dXM;
	rc = alloc_workspace(h);
	if (rc) { sgx_free(h); return rc; }
#undef TRY
	*out = h;
	return SGX_OK;
}

// limb counts of the fixed-point score columns [c' (K), e (K), s, w] and the number of column
// groups; 0 groups = the model takes the FP64 gather kernels
extern "C" int sgx_score_layout(sgx_handle *h, int32_t *limbs, int32_t n_limbs, int32_t *n_groups)
{
	if (!h || !n_groups) return fail(SGX_EINVAL, "sgx_score_layout: NULL argument");
	*n_groups = h->mf_ok ? h->mfe.ngroups : 0;
	for (int c = 0; limbs && c < n_limbs; c++) limbs[c] = (h->mf_ok && c < h->md.P) ? h->mfe.climb[c] : 0;
	return SGX_OK;
}

extern "C" void sgx_free(sgx_handle *h)
{
	if (!h) return;
	(void)hipSetDevice(h->device);
	if (h->stream) (void)hipStreamSynchronize(h->stream);
	for (sgx_handle *&t : h->twins) if (t) { sgx_free(t); t = nullptr; }
	if (!h->shares_model) {
		(void)hipFree(h->dF); (void)hipFree(h->dX); (void)hipFree(h->dy);
		(void)hipFree(h->dmu); (void)hipFree(h->dmu2); (void)hipFree(h->dXM); (void)hipFree(h->dFl); (void)hipFree(h->dQ);
	}
	(void)hipFree(h->fallback); (void)hipFree(h->fb_spa2); (void)hipFree(h->fb_x2);
	(void)hipFree(h->s3_slabs); (void)hipFree(h->s3_t3); (void)hipFree(h->s3_ovf);
	for (int b = 0; b < 2; b++) if (h->tmp_blk[b]) { sgx_block_free(h->tmp_blk[b]); h->tmp_blk[b] = nullptr; }
	(void)hipFree(h->mf_acc); (void)hipFree(h->seg4); (void)hipFree(h->scr5); (void)hipFree(h->cur5);
	(void)hipFree(h->recs); (void)hipFree(h->counters); (void)hipFree(h->scratch);
	for (int b = 0; b < 2; b++) {
		(void)hipFree(h->pipe_in[b]); (void)hipFree(h->pipe_pk[b]); (void)hipFree(h->pipe_out[b]); (void)hipFree(h->pipe_valid[b]);
		if (h->pin_out[b]) (void)hipHostFree(h->pin_out[b]);
		if (h->pin_valid[b]) (void)hipHostFree(h->pin_valid[b]);
	}
	(void)hipFree(h->pipe_flag);
	if (h->h_pipe_flag) (void)hipHostFree(h->h_pipe_flag);
	if (h->ev_h2d) (void)hipEventDestroy(h->ev_h2d);
	for (int k = 0; k < 2; k++) { if (h->ev_copy[k]) (void)hipEventDestroy(h->ev_copy[k]); if (h->ev_done[k]) (void)hipEventDestroy(h->ev_done[k]); }
	if (h->cstream) (void)hipStreamDestroy(h->cstream);
	(void)hipFree(h->stage_in); (void)hipFree(h->stage_out); (void)hipFree(h->stage_valid); (void)hipFree(h->stage_pk); (void)hipFree(h->ds_part);
	if (h->h_counters) (void)hipHostFree(h->h_counters);
	for (int i = 0; i < 3; i++) if (h->ev[i]) (void)hipEventDestroy(h->ev[i]);
	for (int i = 0; i < 2; i++) if (h->evk[i]) (void)hipEventDestroy(h->evk[i]);
	if (h->ev_lists) (void)hipEventDestroy(h->ev_lists);
	if (h->s3_side) { (void)hipStreamSynchronize(h->s3_side); (void)hipStreamDestroy(h->s3_side); }
	if (h->hstream) { (void)hipStreamSynchronize(h->hstream); (void)hipStreamDestroy(h->hstream); }
	if (h->s3_fork) (void)hipEventDestroy(h->s3_fork);
	if (h->s3_join) (void)hipEventDestroy(h->s3_join);
	if (h->stream) (void)hipStreamDestroy(h->stream);
	delete h;
}

static int ensure_recs(sgx_handle *h, size_t n)
{
	if (n <= h->recs_cap) return SGX_OK;
	HIPCHK(hipStreamSynchronize(h->stream));
	if (h->recs) HIPCHK(hipFree(h->recs));
	if (h->fallback) HIPCHK(hipFree(h->fallback));
	h->recs = nullptr; h->fallback = nullptr; h->recs_cap = 0;
	HIPCHK(hipMalloc((void **)&h->recs, 3 * n * sizeof(SpaRec)));   // tier ranges A and B (+ handed-on copies), exact range (dev_common.h)
	HIPCHK(hipMalloc((void **)&h->fallback, n * sizeof(int)));
	if (!h->md.quant) {
		if (h->fb_spa2) HIPCHK(hipFree(h->fb_spa2));
		h->fb_spa2 = nullptr;
		h->nseg = (h->md.N + spa_seg(h->md.K) - 1) / spa_seg(h->md.K);
		HIPCHK(hipMalloc((void **)&h->fb_spa2, n * sizeof(int)));
		if (h->fb_x2) HIPCHK(hipFree(h->fb_x2));
		h->fb_x2 = nullptr;
		HIPCHK(hipMalloc((void **)&h->fb_x2, n * sizeof(int)));
		if (h->seg4) HIPCHK(hipFree(h->seg4));
		h->seg4 = nullptr;
		// flagged variants per round of the series SPA stage: a block of the usual 50 000 variants in one
		// round (a second, normally empty round costs four kernel launches per step)
		h->vcap4 = (int)std::min<size_t>(n, 65536);
		h->nround4 = (int)((n + h->vcap4 - 1) / h->vcap4);
		HIPCHK(hipMalloc((void **)&h->seg4, (size_t)h->nseg * SPA4_NSMAX * h->vcap4 * sizeof(double)));
	}
	if (h->mf_ok) {
		if (h->mf_acc) HIPCHK(hipFree(h->mf_acc));
		h->mf_acc = nullptr;
		HIPCHK(hipMalloc((void **)&h->mf_acc, n * (size_t)(2 * h->mfe.acc_stride - 16) * sizeof(int)));   // (three-plane form: 2 NBF - 1 fragment slots)
	}
	h->recs_cap = n;
	return SGX_OK;
}
