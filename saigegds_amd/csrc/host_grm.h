// host_grm.h -- implicit-GRM operator of the null-model fit (kern_grm.h).
// Part of libsaigehip.so: included by saigehip.hip (one translation unit), not a header of its own.

// ===========================================================================
// Implicit-GRM operator of the null-model fit (kern_grm.h)

struct sgx_grm {
	int device = 0;
	hipStream_t stream = nullptr;
	int N = 0; size_t M = 0;
	size_t bpvN = 0, bpvM = 0;             // row strides of G (marker-major) and Gt (sample-major)
	uint8_t *G = nullptr, *Gt = nullptr;
	double *af = nullptr, *inv = nullptr, *l0 = nullptr, *diag = nullptr;
	MfTab tbN{}, tbM{};                    // limb tiles over samples / over markers
	uint8_t *FlN = nullptr, *FlM = nullptr;
	int *accV = nullptr, *accS = nullptr;  // [M][32], [N][32]
	double *xv = nullptr, *gv = nullptr;   // [M]
	unsigned long long *maxb = nullptr;    // [3]: b, x, gam
	double *part = nullptr, *h_part = nullptr;   // 256 block partials (device / pinned)
	double *vb = nullptr, *vout = nullptr; // [N] staging for host-pointer calls
	double *r = nullptr, *z = nullptr, *p = nullptr, *x = nullptr, *Ap = nullptr, *minv = nullptr, *w = nullptr;
	int n_cu = 256;
};

#define GRM_RED_BLOCKS 256

static int grm_sum(sgx_grm *g, const double *a, const double *b, size_t n, double *out)
{
	if (b) hipLaunchKernelGGL((dot_partial_kernel<true>), dim3(GRM_RED_BLOCKS), dim3(256), 0, g->stream, a, b, n, g->part);
	else hipLaunchKernelGGL((dot_partial_kernel<false>), dim3(GRM_RED_BLOCKS), dim3(256), 0, g->stream, a, b, n, g->part);
	HIPCHK(hipGetLastError());
	HIPCHK(hipMemcpyAsync(g->h_part, g->part, GRM_RED_BLOCKS * sizeof(double), hipMemcpyDeviceToHost, g->stream));
	HIPCHK(hipStreamSynchronize(g->stream));
	double s = 0;
	for (int i = 0; i < GRM_RED_BLOCKS; i++) s += g->h_part[i];   // fixed order
	*out = s;
	return SGX_OK;
}

static dim3 grm_mfma_grid(const sgx_grm *g, size_t rows, int ntile, int *tps)
{
	return mf_grid(g->n_cu, rows, ntile, tps);
}

// out = G'(G b)/M, device vectors (get_crossprod_b_grm, saige_fitnull.cpp:435-536)
static int grm_matvec_dev(sgx_grm *g, const double *b, double *out)
{
	hipStream_t st = g->stream;
	const size_t N = (size_t)g->N, M = g->M;
	const size_t lds = (size_t)2 * 16 * GRM_NCOL * 16;
	double sum_b = 0, C0 = 0;
	int rc = grm_sum(g, b, nullptr, N, &sum_b);
	if (rc) return rc;
	// ---- pass 1: per marker, over samples
	HIPCHK(hipMemsetAsync(g->maxb, 0, 3 * sizeof(unsigned long long), st));
	hipLaunchKernelGGL(absmax_kernel, dim3(GRM_RED_BLOCKS), dim3(256), 0, st, b, N, g->maxb);
	hipLaunchKernelGGL(limbs_kernel, dim3(512), dim3(256), 0, st, b, N, (size_t)g->tbN.ntile * 256, 0, g->maxb, g->FlN);
	HIPCHK(hipMemsetAsync(g->accV, 0, M * GRM_NACC * sizeof(int), st));
	int tps = 0;
	dim3 grid = grm_mfma_grid(g, M, g->tbN.ntile, &tps);
	hipLaunchKernelGGL((score_mfma_kernel<1, false, true>), grid, dim3(WAVE * MF_WAVES), lds, st, g->G, g->bpvN, (int)M, g->tbN, tps, g->accV, GRM_NACC);
	hipLaunchKernelGGL(grm_dot_epilogue, dim3(GRM_RED_BLOCKS), dim3(256), 0, st, M, g->accV, g->maxb, sum_b,
		g->af, g->inv, g->l0, g->xv, g->gv, g->part);
	HIPCHK(hipGetLastError());
	HIPCHK(hipMemcpyAsync(g->h_part, g->part, GRM_RED_BLOCKS * sizeof(double), hipMemcpyDeviceToHost, st));
	HIPCHK(hipStreamSynchronize(st));
	for (int i = 0; i < GRM_RED_BLOCKS; i++) C0 += g->h_part[i];
	// ---- pass 2: per sample, over markers
	hipLaunchKernelGGL(absmax_kernel, dim3(GRM_RED_BLOCKS), dim3(256), 0, st, g->xv, M, g->maxb + 1);
	hipLaunchKernelGGL(absmax_kernel, dim3(GRM_RED_BLOCKS), dim3(256), 0, st, g->gv, M, g->maxb + 2);
	hipLaunchKernelGGL(limbs_kernel, dim3(512), dim3(256), 0, st, g->xv, M, (size_t)g->tbM.ntile * 256, 0, g->maxb + 1, g->FlM);
	hipLaunchKernelGGL(limbs_kernel, dim3(512), dim3(256), 0, st, g->gv, M, (size_t)g->tbM.ntile * 256, MF_NLIMB, g->maxb + 2, g->FlM);
	HIPCHK(hipMemsetAsync(g->accS, 0, N * GRM_NACC * sizeof(int), st));
	grid = grm_mfma_grid(g, N, g->tbM.ntile, &tps);
	hipLaunchKernelGGL((score_mfma_kernel<1, false, true>), grid, dim3(WAVE * MF_WAVES), lds, st, g->Gt, g->bpvM, g->N, g->tbM, tps, g->accS, GRM_NACC);
	hipLaunchKernelGGL(grm_out_epilogue, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, st, g->N, M, g->accS,
		g->maxb + 1, g->maxb + 2, C0, out);
	HIPCHK(hipGetLastError());
	return SGX_OK;
}

extern "C" void sgx_grm_free(sgx_grm *g)
{
	if (!g) return;
	(void)hipSetDevice(g->device);
	if (g->stream) (void)hipStreamSynchronize(g->stream);
	void *ptrs[] = {g->G, g->Gt, g->af, g->inv, g->l0, g->diag, g->FlN, g->FlM, g->accV, g->accS, g->xv, g->gv,
		g->maxb, g->part, g->vb, g->vout, g->r, g->z, g->p, g->x, g->Ap, g->minv, g->w};
	for (void *p : ptrs) (void)hipFree(p);
	if (g->h_part) (void)hipHostFree(g->h_part);
	if (g->stream) (void)hipStreamDestroy(g->stream);
	delete g;
}

// saige_store_2b_geno (saige_fitnull.cpp:159-230): packed = n_markers rows of
// bytes_per_marker bytes (>= ceil(N/4)), 2-bit codes 0/1/2 = allele count, 3 = missing
static int grm_init_impl(const uint8_t *packed, size_t bytes_per_marker, int32_t n_samp,
	size_t n_markers, int device, sgx_grm **out, hipMemcpyKind kind);

extern "C" int sgx_grm_init(const uint8_t *packed, size_t bytes_per_marker, int32_t n_samp,
	size_t n_markers, int device, sgx_grm **out)
{
	return grm_init_impl(packed, bytes_per_marker, n_samp, n_markers, device, out, hipMemcpyHostToDevice);
}

// same, the packed matrix already resident in this GPU's HBM (it is copied)
extern "C" int sgx_grm_init_dev(const uint8_t *packed_dev, size_t bytes_per_marker, int32_t n_samp,
	size_t n_markers, int device, sgx_grm **out)
{
	return grm_init_impl(packed_dev, bytes_per_marker, n_samp, n_markers, device, out, hipMemcpyDeviceToDevice);
}

static int grm_init_impl(const uint8_t *packed, size_t bytes_per_marker, int32_t n_samp,
	size_t n_markers, int device, sgx_grm **out, hipMemcpyKind kind)
{
	if (!packed || !out) return fail(SGX_EINVAL, "sgx_grm_init: NULL argument");
	*out = nullptr;
	if (n_samp <= 0 || n_markers == 0) return fail(SGX_EINVAL, "sgx_grm_init: empty genotype matrix");
	if (bytes_per_marker < (size_t)(n_samp + 3) / 4)
		return fail(SGX_EINVAL, "sgx_grm_init: bytes_per_marker=%zu < ceil(N/4)", bytes_per_marker);
	if ((double)n_markers * 384.0 >= 2147483647.0 || (double)n_samp * 384.0 >= 2147483647.0)
		return fail(SGX_EINVAL, "sgx_grm_init: matrix too large for int32 limb sums");
	int ndev = 0;
	if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(SGX_ENODEV, "sgx_grm_init: no HIP device available");
	if (device < 0 || device >= ndev) return fail(SGX_EINVAL, "sgx_grm_init: device %d out of range", device);
	sgx_grm *g = new sgx_grm();
	g->device = device;
	hipError_t e;
#define GTRY(x) do { e = (x); if (e != hipSuccess) { sgx_grm_free(g); return fail(SGX_EHIP, "%s: %s", #x, hipGetErrorString(e)); } } while (0)
	GTRY(hipSetDevice(device));
	hipDeviceProp_t prop;
	GTRY(hipGetDeviceProperties(&prop, device));
	g->n_cu = prop.multiProcessorCount;
	const size_t N = (size_t)n_samp, M = n_markers;
	g->N = n_samp; g->M = M;
	g->bpvN = sgx_row_stride(n_samp);
	g->bpvM = (size_t)((M + 511) / 512) * 128;
	GTRY(hipStreamCreateWithFlags(&g->stream, hipStreamNonBlocking));
	GTRY(hipMalloc((void **)&g->G, M * g->bpvN));
	GTRY(hipMalloc((void **)&g->Gt, N * g->bpvM));
	GTRY(hipMemsetAsync(g->G, 0, M * g->bpvN, g->stream));
	GTRY(hipMemsetAsync(g->Gt, 0, N * g->bpvM, g->stream));
	GTRY(hipMemcpy2DAsync(g->G, g->bpvN, packed, bytes_per_marker, std::min(bytes_per_marker, g->bpvN), M,
		kind, g->stream));
	for (double **p : {&g->af, &g->inv, &g->l0, &g->xv, &g->gv}) GTRY(hipMalloc((void **)p, M * sizeof(double)));
	for (double **p : {&g->diag, &g->vb, &g->vout, &g->r, &g->z, &g->p, &g->x, &g->Ap, &g->minv, &g->w})
		GTRY(hipMalloc((void **)p, N * sizeof(double)));
	GTRY(hipMalloc((void **)&g->maxb, 3 * sizeof(unsigned long long)));
	GTRY(hipMalloc((void **)&g->part, GRM_RED_BLOCKS * sizeof(double)));
	GTRY(hipHostMalloc((void **)&g->h_part, GRM_RED_BLOCKS * sizeof(double), hipHostMallocDefault));
	auto mk = [&](MfTab &tb, size_t n, uint8_t **Fl) -> hipError_t {
		tb = MfTab{};
		tb.ntile = 2 * (int)((n + 511) / 512);
		const size_t bytes = (size_t)tb.ntile * 16 * GRM_NCOL * 16;
		hipError_t ee = hipMalloc((void **)Fl, bytes);
		if (ee != hipSuccess) return ee;
		ee = hipMemsetAsync(*Fl, 0, bytes, g->stream);
		tb.Fl = *Fl;
		return ee;
	};
	GTRY(mk(g->tbN, N, &g->FlN));
	GTRY(mk(g->tbM, M, &g->FlM));
	GTRY(hipMalloc((void **)&g->accV, M * GRM_NACC * sizeof(int)));
	GTRY(hipMalloc((void **)&g->accS, N * GRM_NACC * sizeof(int)));
	// marker statistics, transpose, diag(GRM)
	hipLaunchKernelGGL(grm_marker_stats, dim3((unsigned)M), dim3(256), 0, g->stream, g->G, g->bpvN, g->N, M, g->af, g->inv, g->l0);
	hipLaunchKernelGGL(transpose_2bit, dim3((unsigned)((N + 255) / 256), (unsigned)((M + 63) / 64)), dim3(256), 0, g->stream,
		g->G, g->bpvN, M, g->N, g->Gt, g->bpvM);
	hipLaunchKernelGGL(grm_diag_kernel, dim3((unsigned)((N + 3) / 4)), dim3(256), 0, g->stream, g->Gt, g->bpvM, g->N, M,
		g->inv, g->l0, g->diag);
	GTRY(hipGetLastError());
	GTRY(hipStreamSynchronize(g->stream));
#undef GTRY
	*out = g;
	return SGX_OK;
}

extern "C" int sgx_grm_diag(sgx_grm *g, double *diag_out)
{
	if (!g || !diag_out) return fail(SGX_EINVAL, "sgx_grm_diag: NULL argument");
	HIPCHK(hipSetDevice(g->device));
	HIPCHK(hipMemcpy(diag_out, g->diag, (size_t)g->N * sizeof(double), hipMemcpyDeviceToHost));
	return SGX_OK;
}

// get_crossprod_b_grm: out = GRM b, host vectors of length N
extern "C" int sgx_grm_crossprod(sgx_grm *g, const double *b, double *out)
{
	if (!g || !b || !out) return fail(SGX_EINVAL, "sgx_grm_crossprod: NULL argument");
	HIPCHK(hipSetDevice(g->device));
	const size_t nb = (size_t)g->N * sizeof(double);
	HIPCHK(hipMemcpyAsync(g->vb, b, nb, hipMemcpyHostToDevice, g->stream));
	int rc = grm_matvec_dev(g, g->vb, g->vout);
	if (rc) return rc;
	HIPCHK(hipMemcpyAsync(out, g->vout, nb, hipMemcpyDeviceToHost, g->stream));
	HIPCHK(hipStreamSynchronize(g->stream));
	return SGX_OK;
}

// PCG_diag_sigma (saige_fitnull.cpp:581-614): solves (tau0 diag(1/w) + tau1 GRM) x = b
extern "C" int sgx_grm_pcg(sgx_grm *g, const double *w, const double *tau, const double *b,
	int maxiter, double tol, double *x_out, int *iters_out)
{
	if (!g || !w || !tau || !b || !x_out) return fail(SGX_EINVAL, "sgx_grm_pcg: NULL argument");
	HIPCHK(hipSetDevice(g->device));
	hipStream_t st = g->stream;
	const int n = g->N;
	const size_t nb = (size_t)n * sizeof(double);
	const dim3 gr((unsigned)((n + 255) / 256)), bl(256);
	const double tau0 = tau[0], tau1 = tau[1];
	HIPCHK(hipMemcpyAsync(g->w, w, nb, hipMemcpyHostToDevice, st));
	HIPCHK(hipMemcpyAsync(g->vb, b, nb, hipMemcpyHostToDevice, st));
	hipLaunchKernelGGL(pcg_minv_kernel, gr, bl, 0, st, n, g->w, g->diag, tau0, tau1, g->minv);
	hipLaunchKernelGGL(pcg_init_kernel, gr, bl, 0, st, n, g->vb, g->minv, g->r, g->z, g->p, g->x);
	int iter = 0, rc;
	double rr = 0, rz = 0;
	if ((rc = grm_sum(g, g->r, g->r, n, &rr))) return rc;
	if ((rc = grm_sum(g, g->r, g->z, n, &rz))) return rc;
	while (iter < maxiter && rr > tol) {
		iter++;
		const double *gp = nullptr;
		if (tau1 != 0) {                       // get_crossprod :569-575
			if ((rc = grm_matvec_dev(g, g->p, g->vout))) return rc;
			gp = g->vout;
		}
		hipLaunchKernelGGL(pcg_ap_kernel, gr, bl, 0, st, n, g->p, g->w, gp, tau0, tau1, g->Ap);
		double pAp = 0;
		if ((rc = grm_sum(g, g->p, g->Ap, n, &pAp))) return rc;
		const double a = rz / pAp;
		hipLaunchKernelGGL(pcg_update_kernel, gr, bl, 0, st, n, a, g->p, g->Ap, g->minv, g->x, g->r, g->z);
		double rz1 = 0;
		if ((rc = grm_sum(g, g->z, g->r, n, &rz1))) return rc;
		const double bet = rz1 / rz;
		hipLaunchKernelGGL(pcg_dir_kernel, gr, bl, 0, st, n, bet, g->z, g->p);
		rz = rz1;
		if ((rc = grm_sum(g, g->r, g->r, n, &rr))) return rc;
	}
	HIPCHK(hipGetLastError());
	HIPCHK(hipMemcpyAsync(x_out, g->x, nb, hipMemcpyDeviceToHost, st));
	HIPCHK(hipStreamSynchronize(st));
	if (iters_out) *iters_out = iter;
	return SGX_OK;
}

// out = GRM b with b, out device vectors of N doubles (asynchronous until sgx_grm_sync)
extern "C" int sgx_grm_crossprod_dev(sgx_grm *g, const double *b_dev, double *out_dev)
{
	if (!g || !b_dev || !out_dev) return fail(SGX_EINVAL, "sgx_grm_crossprod_dev: NULL argument");
	HIPCHK(hipSetDevice(g->device));
	return grm_matvec_dev(g, b_dev, out_dev);
}

extern "C" int sgx_grm_sync(sgx_grm *g)
{
	if (!g) return fail(SGX_EINVAL, "sgx_grm_sync: NULL handle");
	HIPCHK(hipSetDevice(g->device));
	HIPCHK(hipStreamSynchronize(g->stream));
	return SGX_OK;
}
