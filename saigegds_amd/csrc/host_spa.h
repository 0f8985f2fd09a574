// host_spa.h -- kernel dispatch over the compile-time K: the SPA stage of a call, the FP64 score kernels (dosage rows).
// Part of libsaigehip.so: included by saigehip.hip (one translation unit), not a header of its own.

// ---- kernel dispatch over the compile-time K ------------------------------

#define FOR_EACH_K(X) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15) X(16)

// Sample splits of the MFMA kernels: a few rounds of the workgroups a CU holds (wg_per_cu), and a
// multiple of 8 splits when there are that many, so that each XCD works on whole splits
// (kern_score_mfma.h).  vpb: variants per workgroup.
static dim3 mf_grid(int n_cu, size_t rows, int ntile, int *tps, int vpb = MF_VPB, int wg_per_cu = 2)
{
	const int vt = (int)((rows + vpb - 1) / vpb);
	int sk = std::max(1, (n_cu * wg_per_cu * 4 + vt / 2) / vt);
	sk = std::min(sk, std::max(1, ntile / 24));   // a split shorter than ~24 tiles is mostly prologue and atomics
	if (sk >= 6) sk = (sk + 7) & ~7;
	sk = std::min(sk, std::max(1, ntile / 2));
	*tps = (ntile + sk - 1) / sk;
	*tps += *tps & 1;                         // even tile ranges (wide-row kernel)
	sk = (ntile + *tps - 1) / *tps;
	return dim3((unsigned)vt, (unsigned)sk);
}

// SPA stage of the flagged variants of a call (their records are in h->recs): the series kernels
// (kern_spa4.h), the per-variant kernels and the exact dense pass.  rr: the call's rows.
// lazy_dense (device-resident calls, whose results are read after a sync): the exact dense pass -- normally
// without a single variant -- is not launched here.  An empty launch of its 512-thread workgroups at the end of
// every step still has to wait for room on a CU beside the other lane's contraction kernel or cumulant pass
// (0.1-1.2 ms in kernel traces), and with it the lane's completion and its next step.  The next sync of the lane
// reads the step's counters and launches the pass if a variant asked for it (sync_lane).
template <int INPUT>
static int launch_spa(sgx_handle *h, RowsRef rr, size_t M, double *out8, bool lazy_dense = false)
{
	const DevModel &md = h->md;
	constexpr int PB = 512;
	hipStream_t st = h->stream;
	h->stats.spa_launches = 0;
	if (!md.quant) {
		const dim3 sgrid((unsigned)std::min<size_t>(M, (size_t)h->spa_grid));
		switch (md.K) {
#define MOMENTS(KK, NCX, TIER, RD)                                                               \
	do {                                                                                         \
		if (INPUT == IN_2BIT)                                                                    \
			hipLaunchKernelGGL((spa4_moments<KK, NCX>), dim3((unsigned)h->n_cu), \
				dim3(WAVE * spa4_waves(KK)), fl, st, rr, md, h->nseg,  \
				TIER, btop, (RD) * h->vcap4, h->vcap4, h->recs, h->counters, h->seg4, h->spa_abl, h->cur5 + 2); \
		else                                                                                     \
			hipLaunchKernelGGL((spa4_moments_ds<KK, NCX, (INPUT == IN_2BIT ? IN_U8 : INPUT)>),   \
				dim3((unsigned)h->n_cu), dim3(WAVE * spa4_waves(KK)), fl, st, (const void *)rr.base, rr.bpv, md, \
				h->nseg, TIER, btop, (RD) * h->vcap4, h->vcap4, h->recs, h->counters, h->seg4, h->cur5 + 2);  \
		hipLaunchKernelGGL((spa4_solve<KK, NCX>), gsolve, dim3(256), 0, st, md, h->nseg, TIER,   \
			btop, (RD) * h->vcap4, h->vcap4, h->recs, h->counters, h->seg4, h->fallback,         \
			h->fb_x2, out8, h->force_dense ? 1 : 0, h->force_exact ? 1 : 0);                     \
	} while (0)
#define CASE(KK)                                                                             \
	case KK:                                                                                 \
		if (h->force_v1 && INPUT != IN_2BIT) {                                               \
			hipLaunchKernelGGL((spa_kernel<KK, PB, INPUT>), sgrid, dim3(PB), 0, st, rr,    \
				md, h->recs, h->counters, 0, (const int *)nullptr, h->scratch,    \
				h->scratch_stride, out8);                                                    \
		} else {                                                                             \
			/* series SPA stage (kern_spa4.h): rounds of at most vcap4 flagged variants;     \
			   tier A (short series), then tier B with what tier A handed on */              \
			const size_t fl = spa4_lds_bytes(KK);                                            \
			if (!h->mom_attr_set[INPUT]) {                                                   \
				const void *fa = INPUT == IN_2BIT ? (const void *)spa4_moments<KK, SPA4_NCA> \
					: (const void *)spa4_moments_ds<KK, SPA4_NCA, (INPUT == IN_2BIT ? IN_U8 : INPUT)>; \
				const void *fb = INPUT == IN_2BIT ? (const void *)spa4_moments<KK, SPA4_NCB> \
					: (const void *)spa4_moments_ds<KK, SPA4_NCB, (INPUT == IN_2BIT ? IN_U8 : INPUT)>; \
				HIPCHK(hipFuncSetAttribute(fa, hipFuncAttributeMaxDynamicSharedMemorySize, (int)fl)); \
				HIPCHK(hipFuncSetAttribute(fb, hipFuncAttributeMaxDynamicSharedMemorySize, (int)fl)); \
				h->mom_attr_set[INPUT] = true;                                               \
			}                                                                                \
			const int nround = (int)((M + h->vcap4 - 1) / h->vcap4);                         \
			const int btop = (int)(2 * M);                                                   \
			const dim3 gsolve((unsigned)std::min((h->vcap4 + 3) / 4, 4 * h->n_cu));   /* a wave per variant, grid-stride */ \
			/* what the series does not cover: exact exp/log sums, one workgroup per variant; \
			   then the exact dense g_pos / g_neg pass */                                    \
			/* a packed row in LDS when it fits; short rows: 128 threads per variant, 4 workgroups per CU */ \
			const size_t rowb5 = (size_t)((md.N + 63) / 64) * 16;                            \
			const size_t l5 = (INPUT == IN_2BIT && rowb5 <= 120 * 1024) ? rowb5 : 0;         \
			const bool small5 = INPUT == IN_2BIT && rowb5 <= 32 * 1024;                      \
			if (l5 > 48 * 1024 && !h->spa5_attr_set[INPUT]) {                                \
				HIPCHK(hipFuncSetAttribute((const void *)spa5_kernel<KK, INPUT, 0, 512>,     \
					hipFuncAttributeMaxDynamicSharedMemorySize, (int)l5));                   \
				HIPCHK(hipFuncSetAttribute((const void *)spa5_kernel<KK, INPUT, 1, 512>,     \
					hipFuncAttributeMaxDynamicSharedMemorySize, (int)l5));                   \
				h->spa5_attr_set[INPUT] = true;                                              \
			}                                                                                \
			const int fx5 = (h->force_exact ? 1 : 0) | (h->spa_abl & ~1);                    \
			for (int rd = 0; rd < nround; rd++) MOMENTS(KK, SPA4_NCA, 0, rd);                \
			for (int rd = 0; rd < nround; rd++) MOMENTS(KK, SPA4_NCB, 1, rd);                \
			/* (genotype blocks carry the carrier lists of the rare variants, rr.cptr: the kernels walk those \
			   instead of scanning the row; spa_abl & 512 makes them scan, as for row-major input) */ \
			const int only5 = (INPUT == IN_2BIT && rr.cptr != nullptr && (h->spa_abl & 512)) ? 2 : 0; \
			const size_t ws5 = spa5_wg_bytes(md.N);                                          \
			if (small5) {                                                                    \
				/* (2-bit rows only: the constant keeps the other inputs' 128-thread forms uninstantiated) */ \
				constexpr int IN5 = INPUT == IN_2BIT ? INPUT : IN_2BIT;                      \
				hipLaunchKernelGGL((spa5_kernel<KK, IN5, 0, 128>), dim3((unsigned)h->nwg5), dim3(128), \
					l5, st, rr, md, h->recs, h->counters, h->fb_spa2, h->fb_x2, h->cur5, \
					h->fallback, h->scr5, out8, h->force_dense ? 1 : 0, fx5, l5, only5, ws5, 3); \
				hipLaunchKernelGGL((spa5_kernel<KK, IN5, 1, 128>), dim3((unsigned)h->nwg5), dim3(128), \
					l5, st, rr, md, h->recs, h->counters, h->fb_x2, h->fb_x2, h->cur5 + 1, \
					h->fallback, h->scr5, out8, h->force_dense ? 1 : 0, fx5, l5, only5, ws5, 4); \
			} else {                                                                         \
				hipLaunchKernelGGL((spa5_kernel<KK, INPUT, 0, 512>), dim3((unsigned)h->n_cu), dim3(512), \
					l5, st, rr, md, h->recs, h->counters, h->fb_spa2, h->fb_x2, h->cur5, \
					h->fallback, h->scr5, out8, h->force_dense ? 1 : 0, fx5, l5, only5, ws5, 3); \
				hipLaunchKernelGGL((spa5_kernel<KK, INPUT, 1, 512>), dim3((unsigned)h->n_cu), dim3(512), \
					l5, st, rr, md, h->recs, h->counters, h->fb_x2, h->fb_x2, h->cur5 + 1, \
					h->fallback, h->scr5, out8, h->force_dense ? 1 : 0, fx5, l5, only5, ws5, 4); \
			}                                                                                \
			if (!lazy_dense)                                                                 \
				hipLaunchKernelGGL((spa_kernel<KK, PB, INPUT>), sgrid, dim3(PB), 0, st, rr,  \
					md, h->recs, h->counters, 2, h->fallback, h->scratch,                    \
					h->scratch_stride, out8);                                                \
		}                                                                                    \
		break;
			FOR_EACH_K(CASE)
#undef CASE
#undef MOMENTS
		}
		HIPCHK(hipGetLastError());
		h->stats.spa_launches = (h->force_v1 && INPUT != IN_2BIT) ? 1u : (uint32_t)(4 * ((M + h->vcap4 - 1) / h->vcap4) + 3);
		if (lazy_dense && !(h->force_v1 && INPUT != IN_2BIT)) { h->pend_dense.active = true; h->pend_dense.rr = rr; h->pend_dense.M = M; h->pend_dense.out8 = out8; }
	}
	return SGX_OK;
}

// the dense pass a lazy call left out, for the variants on its fallback list (counters[2] of that call)
static int launch_pending_dense(sgx_handle *h)
{
	const DevModel &md = h->md;
	const RowsRef rr = h->pend_dense.rr;
	double *out8 = h->pend_dense.out8;
	const dim3 sgrid((unsigned)std::min<size_t>(h->pend_dense.M, (size_t)h->spa_grid));
	switch (md.K) {
#define DCASE(KK) case KK: hipLaunchKernelGGL((spa_kernel<KK, 512, IN_2BIT>), sgrid, dim3(512), 0, h->stream, rr, md, h->recs, \
		h->counters, 2, h->fallback, h->scratch, h->scratch_stride, out8); break;
		FOR_EACH_K(DCASE)
#undef DCASE
	}
	HIPCHK(hipGetLastError());
	return SGX_OK;
}

// Score stage by the FP64 kernels (dosage rows; 2-bit rows of the "score_v1" hook or of a model the
// fixed-point form does not hold), then the SPA stage.  Row-major rows.
template <int INPUT>
static int launch_scan(sgx_handle *h, const void *rows, size_t row_bytes, size_t M,
	double *out8, uint8_t *valid)
{
	const DevModel &md = h->md;
	constexpr int SB = 256;
	hipStream_t st = h->stream;
	const RowsRef rr{reinterpret_cast<const uint8_t *>(rows), row_bytes, 0, nullptr, nullptr, nullptr};
	HIPCHK(hipMemsetAsync(h->counters, 0, 24 * sizeof(int), st));
	if (h->cur5) HIPCHK(hipMemsetAsync(h->cur5, 0, 8 * sizeof(int), st));
	HIPCHK(hipEventRecord(h->ev[0], st));
	{
	const dim3 grid((unsigned)M);
		switch (md.K) {
	#define CASE(KK)                                                                             \
		case KK:                                                                                 \
			if (INPUT == IN_2BIT)                                                                \
				hipLaunchKernelGGL((score2b_kernel<2 * KK + 2, SB>), grid, dim3(SB), 0, st,     \
					rr, (int)M, md, h->recs, h->counters, out8, valid, (const int *)nullptr, 0, 0, (int *)nullptr, (int *)nullptr); \
			else if (KK <= 8 && !h->force_v1) {                                                  \
				/* tiled one-pass kernels: 32 variants x a sample range per workgroup */         \
				constexpr int PT = (KK <= 8) ? 2 * KK + 2 : 4;                                   \
				const int vb = (int)((M + DS_TILE_VB - 1) / DS_TILE_VB);                         \
				int ns = std::max(1, std::min((4 * h->n_cu + vb - 1) / vb, (md.N + 4095) / 4096)); \
				int per = (((md.N + ns - 1) / ns) + 63) & ~63;                                   \
				ns = (md.N + per - 1) / per;                                                     \
				const size_t need = (size_t)ns * M * (3 * PT + 2) * sizeof(double);              \
				if (need > h->ds_part_cap) {                                                     \
					HIPCHK(hipStreamSynchronize(st));                                            \
					if (h->ds_part) HIPCHK(hipFree(h->ds_part));                                 \
					h->ds_part = nullptr; h->ds_part_cap = 0;                                    \
					HIPCHK(hipMalloc((void **)&h->ds_part, need));                               \
					h->ds_part_cap = need;                                                       \
				}                                                                                \
				const dim3 gt((unsigned)vb, (unsigned)ns);                                       \
				if (INPUT == IN_U8)                                                              \
					hipLaunchKernelGGL((score_ds_tile_kernel<PT, uint8_t>), gt, dim3(256), 0, st, \
						(const uint8_t *)rows, (int)M, md, per, h->ds_part);                     \
				else                                                                             \
					hipLaunchKernelGGL((score_ds_tile_kernel<PT, double>), gt, dim3(256), 0, st, \
						(const double *)rows, (int)M, md, per, h->ds_part);                      \
				hipLaunchKernelGGL((score_ds_tile_epilogue<PT>), dim3((unsigned)((M + 255) / 256)), dim3(256), 0, st, \
					(int)M, md, ns, h->ds_part, h->recs, h->counters, out8, valid);              \
			} else if (INPUT == IN_U8)                                                           \
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
	}
	HIPCHK(hipGetLastError());
	HIPCHK(hipEventRecord(h->ev[1], st));
	h->stats.score_launches = 1;
	int rc = launch_spa<INPUT>(h, rr, M, out8);
	if (rc) return rc;
	HIPCHK(hipEventRecord(h->ev[2], st));
	HIPCHK(hipMemcpyAsync(h->h_counters, h->counters, 24 * sizeof(int), hipMemcpyDeviceToHost, st));
	h->stats.n_variants = M;
	h->stats_pending = true;
	return SGX_OK;
}
