// host_blocks.h -- genotype blocks: create / load / lists (kern_lists.h).
// Part of libsaigehip.so: included by saigehip.hip (one translation unit), not a header of its own.

// ---------------------------------------------------------------------------
// Genotype blocks (kern_score3.h)

// entries of a block's pools for max_variants rows of n_samp samples
static size_t block_idx_cap(int32_t n_samp, size_t max_variants)
{
	// missing genotypes: room for max(64, N / 128) per variant on average (0.8 % at large N), at least a few
	// segments per sub-pool; variants that find the pool full take the FP64 kernel
	return std::min<size_t>(std::max<size_t>(max_variants * std::max<size_t>(64, (size_t)n_samp / 128), (size_t)S3_NSUB * 256), 0xF0000000u);
}
static size_t block_cidx_cap(int32_t n_samp, size_t max_variants, size_t cavg)
{
	// carrier lists: 1536 entries per variant on average (a log-uniform MAF spectrum from 5e-4 lists ~40 % of the
	// variants at N = 430 000 with ~3 000 carriers each)
	return std::min<size_t>(max_variants * std::min<size_t>(cavg, (size_t)n_samp), 0xF0000000u);
}
#define SGX_CLIST_AVG 1536

extern "C" size_t sgx_block_bytes(int32_t n_samp, size_t max_variants)
{
	if (n_samp <= 0 || max_variants == 0) return 0;
	const int ntile = 2 * ((n_samp + 511) / 512), nr = s3_nranges(ntile);
	return max_variants * (size_t)ntile * 64 + block_idx_cap(n_samp, max_variants) * 4 + block_cidx_cap(n_samp, max_variants, SGX_CLIST_AVG) * 4 +
		max_variants * ((size_t)nr * 16 + 5 + 13) + (size_t)S3_NSUB * S3_CURSOR_STRIDE * 4 + 8;
}

extern "C" void sgx_block_free(sgx_block *b)
{
	if (!b) return;
	(void)hipSetDevice(b->device);
	if (b->last_read && b->was_read) (void)hipEventSynchronize(b->last_read);    // scans that read it are done
	(void)hipFree(b->rows); (void)hipFree(b->idx); (void)hipFree(b->cursor); (void)hipFree(b->lstart); (void)hipFree(b->lcnt);
	(void)hipFree(b->nzp); (void)hipFree(b->n2p); (void)hipFree(b->n3); (void)hipFree(b->ovf);
	(void)hipFree(b->nzv); (void)hipFree(b->n2v); (void)hipFree(b->cptr); (void)hipFree(b->cidx); (void)hipFree(b->corient);
	(void)hipFree(b->info);
	if (b->h_info) (void)hipHostFree(b->h_info);
	if (b->ready) (void)hipEventDestroy(b->ready);
	if (b->last_read) (void)hipEventDestroy(b->last_read);
	delete b;
}

// cavg: carrier-list entries per variant on average (resident blocks); lists_only: the scratch of a row-major scan
static int block_create(int32_t n_samp, size_t max_variants, int device, bool lists_only, size_t cavg, sgx_block **out)
{
	*out = nullptr;
	if (n_samp <= 0 || max_variants == 0 || max_variants > 0x7fffffffu / S3_NR)
		return fail(SGX_EINVAL, "sgx_block_create: n_samp = %d, max_variants = %zu", n_samp, max_variants);
	int ndev = 0;
	if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(SGX_ENODEV, "sgx_block_create: no HIP device available");
	if (device < 0 || device >= ndev) return fail(SGX_EINVAL, "sgx_block_create: device %d out of range", device);
	sgx_block *b = new sgx_block();
	b->device = device; b->N = n_samp; b->ntile = 2 * ((n_samp + 511) / 512); b->nr = s3_nranges(b->ntile); b->cap = max_variants;
	b->lists_only = lists_only;
	b->bpv = (size_t)b->ntile * 64;
	b->idx_cap = lists_only ? 0 : block_idx_cap(n_samp, max_variants);    // (the row-major calls gather on the spot: no pool)
	b->cidx_cap = lists_only ? 0 : block_cidx_cap(n_samp, max_variants, cavg);
	const size_t nrc = (size_t)b->nr * max_variants;
	hipError_t e = hipSetDevice(device);
	if (e == hipSuccess && !lists_only) e = hipMalloc((void **)&b->rows, max_variants * b->bpv);
	if (e == hipSuccess && !lists_only) e = hipMalloc((void **)&b->idx, b->idx_cap * sizeof(unsigned));
	if (e == hipSuccess) e = hipMalloc((void **)&b->cursor, (size_t)S3_NSUB * S3_CURSOR_STRIDE * sizeof(unsigned));
	if (e == hipSuccess) e = hipMemset(b->cursor, 0, (size_t)S3_NSUB * S3_CURSOR_STRIDE * sizeof(unsigned));
	if (e == hipSuccess && !lists_only) e = hipMalloc((void **)&b->lstart, nrc * sizeof(unsigned));
	if (e == hipSuccess) e = hipMalloc((void **)&b->lcnt, nrc * sizeof(int));
	if (e == hipSuccess) e = hipMalloc((void **)&b->n3, max_variants * sizeof(int));
	if (e == hipSuccess) e = hipMalloc((void **)&b->ovf, max_variants);
	if (e == hipSuccess) e = hipMalloc((void **)&b->info, 2 * sizeof(int));
	if (e == hipSuccess) e = hipMemset(b->info, 0, 2 * sizeof(int));
	if (e == hipSuccess) e = hipHostMalloc((void **)&b->h_info, 2 * sizeof(int), hipHostMallocDefault);
	if (!lists_only) {
		if (e == hipSuccess) e = hipMalloc((void **)&b->nzp, nrc * sizeof(int));
		if (e == hipSuccess) e = hipMalloc((void **)&b->n2p, nrc * sizeof(int));
		if (e == hipSuccess) e = hipMalloc((void **)&b->nzv, max_variants * sizeof(int));
		if (e == hipSuccess) e = hipMalloc((void **)&b->n2v, max_variants * sizeof(int));
		if (e == hipSuccess) e = hipMalloc((void **)&b->cptr, (max_variants + 1) * sizeof(unsigned));
		if (e == hipSuccess) e = hipMalloc((void **)&b->cidx, std::max<size_t>(b->cidx_cap, 1) * sizeof(unsigned));
		if (e == hipSuccess) e = hipMalloc((void **)&b->corient, max_variants);
	}
	if (e == hipSuccess) e = hipEventCreateWithFlags(&b->ready, hipEventDisableTiming);
	if (e == hipSuccess) e = hipEventCreateWithFlags(&b->last_read, hipEventDisableTiming);
	if (e != hipSuccess) { sgx_block_free(b); return fail(e == hipErrorOutOfMemory ? SGX_ENOMEM : SGX_EHIP, "sgx_block_create: %s", hipGetErrorString(e)); }
	*out = b;
	return SGX_OK;
}

extern "C" int sgx_block_create(int32_t n_samp, size_t max_variants, int device, sgx_block **out)
{
	if (!out) return fail(SGX_EINVAL, "sgx_block_create: NULL argument");
	return block_create(n_samp, max_variants, device, false, SGX_CLIST_AVG, out);
}

// test hook: a resident block whose carrier lists hold `clist_avg` entries per variant on average (the later
// variants of a block go unlisted and have their rows scanned by the SPA kernels)
extern "C" int sgx_block_create_ex(int32_t n_samp, size_t max_variants, int device, long long clist_avg, sgx_block **out)
{
	if (!out || clist_avg < 0) return fail(SGX_EINVAL, "sgx_block_create_ex: bad argument");
	return block_create(n_samp, max_variants, device, false, (size_t)clist_avg, out);
}

static S3Lists block_lists(const sgx_block *b)
{
	S3Lists L{};
	L.idx = b->idx; L.idx_cap = (unsigned)b->idx_cap; L.cursor = b->cursor; L.lstart = b->lstart; L.lcnt = b->lcnt;
	L.nzp = b->nzp; L.n2p = b->n2p; L.ld = b->cap; L.nr = b->nr;
	s3_lists_setup(L, b->ntile, (b->cap * (size_t)b->nr + 3) / 4);
	return L;
}
static RowsRef block_rows(const sgx_block *b)
{
	if (b->lists_only) return RowsRef{b->ext_rows, b->ext_bpv, 0, nullptr, nullptr, nullptr};
	return RowsRef{b->rows, b->bpv, 0, b->cptr, b->cidx, b->corient};
}

// rows [v_first, v_first + m) of the block from row-major device rows: ONE pass over the rows lists their missing
// genotypes (and, into a resident block, copies them and counts the carriers); any number of calls, then
// block_finish once
static int block_put_rows(sgx_block *b, const uint8_t *rows_dev, size_t bpv, size_t v_first, size_t m, hipStream_t st)
{
	const dim3 grid((unsigned)((m + 3) / 4), (unsigned)b->nr);
	const S3Lists L = block_lists(b);
	if (b->lists_only)
		hipLaunchKernelGGL((s3_lists_kernel<8, false, false>), grid, dim3(256), 0, st, rows_dev, bpv, b->N, (int)m, (int)v_first, b->ntile, L,
			(uint8_t *)nullptr, (size_t)0);
	else
		hipLaunchKernelGGL((s3_lists_kernel<8, true, true>), grid, dim3(256), 0, st, rows_dev, bpv, b->N, (int)m, (int)v_first, b->ntile, L,
			b->rows, b->bpv);
	HIPCHK(hipGetLastError());
	return SGX_OK;
}

static int block_finish(sgx_block *b, size_t M, hipStream_t st)
{
	const S3Lists L = block_lists(b);
	hipLaunchKernelGGL(s3_lists_finish_kernel, dim3((unsigned)((std::max<size_t>(M, S3_NSUB) + 255) / 256)), dim3(256), 0, st, (int)M, L, b->n3, b->ovf,
		b->lists_only ? (int *)nullptr : b->nzv, b->lists_only ? (int *)nullptr : b->n2v, b->lists_only ? (int *)nullptr : b->info);
	if (!b->lists_only) {
		hipLaunchKernelGGL(s3_ingest_clist_count_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, st, (int)M, b->N, SPA5_NNZ, b->nzv, b->n2v, b->n3, b->corient);
		hipLaunchKernelGGL(s3_ingest_clist_kernel, dim3(1), dim3(1024), 0, st, (int)M, (unsigned)b->cidx_cap, b->nzv, b->cptr, b->corient);
		hipLaunchKernelGGL((s3_clist_fill_kernel<8>), dim3((unsigned)((M * (size_t)b->nr + 3) / 4)), dim3(256), 0, st, b->rows, b->bpv, b->N, (int)M, b->ntile, L,
			b->corient, b->cptr, b->cidx);
	}
	HIPCHK(hipGetLastError());
	if (!b->lists_only) {
		HIPCHK(hipMemcpyAsync(b->h_info, b->info, 2 * sizeof(int), hipMemcpyDeviceToHost, st));
		HIPCHK(hipMemsetAsync(b->info, 0, 2 * sizeof(int), st));
		b->info_read = false;
	}
	HIPCHK(hipEventRecord(b->ready, st));
	b->M = M;
	return SGX_OK;
}

// a (re)load of a block: the stream that writes it waits for the scans that still read it; a lane of this handle
// whose deferred dense pass (launch_spa, lazy_dense) still points at the block is brought to its end first
static int sync_lane(sgx_handle *h);
static int block_begin_load(sgx_handle *h, sgx_block *b, hipStream_t st)
{
	sgx_handle *p = h->owner ? h->owner : h;
	sgx_handle *lanes[4] = {p, p->twins[0], p->twins[1], p->twins[2]};
	for (sgx_handle *l : lanes) if (l && l->pend_dense.active && l->pend_dense.blk == b) { int rc = sync_lane(l); if (rc) return rc; }
	if (b->was_read) HIPCHK(hipStreamWaitEvent(st, b->last_read, 0));
	b->M = 0;
	return SGX_OK;
}

static int check_block_args(sgx_handle *h, sgx_block *b, size_t bpv, size_t M, const char *who)
{
	if (!h || !b) return fail(SGX_EINVAL, "%s: NULL argument", who);
	if (b->lists_only) return fail(SGX_EINVAL, "%s: not a resident block", who);
	if (b->device != h->device) return fail(SGX_EINVAL, "%s: block and handle are on different devices", who);
	if (M == 0 || M > b->cap) return fail(SGX_EINVAL, "%s: %zu variants, the block holds up to %zu", who, M, b->cap);
	if (bpv % 16 != 0 || bpv < (size_t)b->ntile * 64)
		return fail(SGX_EINVAL, "Invalid length of dosages: bytes_per_variant=%zu, need a multiple of 16 >= %zu", bpv, (size_t)b->ntile * 64);
	return SGX_OK;
}

extern "C" int sgx_block_load_dev(sgx_handle *h, sgx_block *b, const uint8_t *packed_dev, size_t bpv, size_t M)
{
	int rc = check_block_args(h, b, bpv, M, "sgx_block_load_dev");
	if (rc) return rc;
	if (!packed_dev || ((uintptr_t)packed_dev & 15u)) return fail(SGX_EINVAL, "sgx_block_load_dev: packed_dev must be a 16-byte aligned device pointer");
	rc = set_dev(h);
	if (rc) return rc;
	rc = block_begin_load(h, b, h->stream);
	if (rc) return rc;
	rc = block_put_rows(b, packed_dev, bpv, 0, M, h->stream);
	if (rc) return rc;
	return block_finish(b, M, h->stream);
}

extern "C" size_t sgx_block_variants(const sgx_block *b) { return b ? b->M : 0; }
