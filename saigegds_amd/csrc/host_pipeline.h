// host_pipeline.h -- host-buffer scans: the PCIe pipeline, block loads from host rows, dosage inputs, burden rows.
// Part of libsaigehip.so: included by saigehip.hip (one translation unit), not a header of its own.

static const size_t STAGE_BYTES = (size_t)1 << 30;    // burden rows are made and scanned in chunks of this size

static int ensure_stage(sgx_handle *h, size_t in_bytes, size_t M)
{
	if (in_bytes > h->stage_in_cap) {
		if (h->stage_in) HIPCHK(hipFree(h->stage_in));
		h->stage_in = nullptr; h->stage_in_cap = 0;
		HIPCHK(hipMalloc((void **)&h->stage_in, in_bytes));
		h->stage_in_cap = in_bytes;
	}
	if (M > h->stage_out_cap) {
		if (h->stage_out) HIPCHK(hipFree(h->stage_out));
		if (h->stage_valid) HIPCHK(hipFree(h->stage_valid));
		h->stage_out = nullptr; h->stage_valid = nullptr; h->stage_out_cap = 0;
		HIPCHK(hipMalloc((void **)&h->stage_out, M * 8 * sizeof(double)));
		HIPCHK(hipMalloc((void **)&h->stage_valid, M));
		h->stage_out_cap = M;
	}
	return SGX_OK;
}

// ---------------------------------------------------------------------------
// Host-buffer scans: a two-stage pipeline over chunks of the caller's block.  While chunk i is being
// computed on the handle's stream, chunk i + 1 crosses PCIe on a second stream into the other input
// buffer; results come back through pinned memory, so no copy of the caller's pageable buffers ever
// waits for a kernel.  RAW / INTEGER dosages that are hard calls (0, 1, 2, missing) are packed to
// 2-bit rows on the device (kern_pack.h) and take the MFMA path.
enum { IN_I32 = 3 };
static const size_t PIPE_BYTES = (size_t)512 << 20;      // device bytes of one chunk's input rows ("pipe_mb" option)

static int ensure_pipe(sgx_handle *h, size_t in_bytes, size_t pk_bytes, size_t M)
{
	if (!h->cstream) {
		HIPCHK(hipStreamCreateWithFlags(&h->cstream, hipStreamNonBlocking));
		HIPCHK(hipEventCreateWithFlags(&h->ev_h2d, hipEventDisableTiming));
		for (int k = 0; k < 2; k++) { HIPCHK(hipEventCreateWithFlags(&h->ev_copy[k], hipEventDisableTiming)); HIPCHK(hipEventCreateWithFlags(&h->ev_done[k], hipEventDisableTiming)); }
		HIPCHK(hipMalloc((void **)&h->pipe_flag, sizeof(int)));
		HIPCHK(hipHostMalloc((void **)&h->h_pipe_flag, sizeof(int), hipHostMallocDefault));
	}
	if (in_bytes > h->pipe_in_cap) {
		for (int b = 0; b < 2; b++) { if (h->pipe_in[b]) HIPCHK(hipFree(h->pipe_in[b])); h->pipe_in[b] = nullptr; }
		h->pipe_in_cap = 0;
		for (int b = 0; b < 2; b++) HIPCHK(hipMalloc((void **)&h->pipe_in[b], in_bytes));
		h->pipe_in_cap = in_bytes;
	}
	if (pk_bytes > h->pipe_pk_cap) {
		for (int b = 0; b < 2; b++) { if (h->pipe_pk[b]) HIPCHK(hipFree(h->pipe_pk[b])); h->pipe_pk[b] = nullptr; }
		h->pipe_pk_cap = 0;
		for (int b = 0; b < 2; b++) HIPCHK(hipMalloc((void **)&h->pipe_pk[b], pk_bytes));
		h->pipe_pk_cap = pk_bytes;
	}
	if (M > h->pipe_out_cap) {
		for (int b = 0; b < 2; b++) {
			if (h->pipe_out[b]) HIPCHK(hipFree(h->pipe_out[b]));
			if (h->pipe_valid[b]) HIPCHK(hipFree(h->pipe_valid[b]));
			if (h->pin_out[b]) HIPCHK(hipHostFree(h->pin_out[b]));
			if (h->pin_valid[b]) HIPCHK(hipHostFree(h->pin_valid[b]));
			h->pipe_out[b] = nullptr; h->pipe_valid[b] = nullptr; h->pin_out[b] = nullptr; h->pin_valid[b] = nullptr;
		}
		h->pipe_out_cap = 0;
		for (int b = 0; b < 2; b++) {
			HIPCHK(hipMalloc((void **)&h->pipe_out[b], M * 8 * sizeof(double)));
			HIPCHK(hipMalloc((void **)&h->pipe_valid[b], M));
			HIPCHK(hipHostMalloc((void **)&h->pin_out[b], M * 8 * sizeof(double), hipHostMallocDefault));
			HIPCHK(hipHostMalloc((void **)&h->pin_valid[b], M, hipHostMallocDefault));
		}
		h->pipe_out_cap = M;
	}
	return SGX_OK;
}

template <int INPUT>
static int scan_host(sgx_handle *h, const void *rows, size_t src_row_bytes, size_t dev_row_bytes,
	size_t M, double *out8, uint8_t *valid)
{
	if (!h) return fail(SGX_EINVAL, "scan: NULL handle");
	if (M == 0) return SGX_OK;
	if (!rows || !out8 || !valid) return fail(SGX_EINVAL, "scan: NULL buffer");
	int rc = set_dev(h);
	if (rc) return rc;
	rc = sync_lane(h);                        // anything queued on this handle before is done
	if (rc) return rc;
	h->last_issued = h;                       // sgx_get_stats: this call, not an earlier one on the twin lane
	const int N = h->md.N;
	const size_t pk_row = sgx_row_stride(N);
	const bool can_pack = (INPUT == IN_U8 || INPUT == IN_I32) && h->mf_ok && !h->force_v1;
	// a chunk's rows on the device: as they arrive (+ the doubles INTEGER rows may have to become)
	const size_t per_row = dev_row_bytes + (INPUT == IN_I32 ? (size_t)N * sizeof(double) : 0);
	size_t chunk = std::min(M, std::max<size_t>(1, (h->pipe_bytes ? h->pipe_bytes : PIPE_BYTES) / per_row));
	if (can_pack) chunk = std::min<size_t>(chunk, 65535);       // pack_rows_2bit: grid.y = rows
	const size_t f64_off = (chunk * dev_row_bytes + 15) & ~(size_t)15;   // INTEGER rows that are not hard calls: their doubles
	rc = ensure_pipe(h, f64_off + (INPUT == IN_I32 ? chunk * (size_t)N * sizeof(double) : 0), can_pack ? chunk * pk_row : 0, chunk);
	if (rc) return rc;
	rc = ensure_recs(h, chunk);
	if (rc) return rc;
	// 2-bit rows (as they come, or packed from hard calls) take the MFMA path, the lists of a chunk per pipeline buffer
	const bool blocks = (INPUT == IN_2BIT || can_pack) && h->mf_ok && !h->force_v1;
	if (blocks) for (int b = 0; b < 2; b++) { rc = ensure_tmp_block(h, b, chunk); if (rc) return rc; }
	sgx_stats total{};
	auto harvest = [&](size_t off, size_t m, int b) -> int {       // chunk [off, off + m) of buffer b is done
		int r2 = sync_lane(h);
		if (r2) return r2;
		memcpy(out8 + off * 8, h->pin_out[b], m * 8 * sizeof(double));
		memcpy(valid + off, h->pin_valid[b], m);
		const sgx_stats &x = h->stats;
		total.n_variants += x.n_variants; total.n_valid += x.n_valid; total.n_spa += x.n_spa;
		total.n_spa_dense += x.n_spa_dense; total.n_spa_slow += x.n_spa_slow;
		total.ms_score += x.ms_score; total.ms_spa += x.ms_spa; total.ms_total += x.ms_total;
		total.ms_kernel += x.ms_kernel; total.ms_lists += x.ms_lists;
		total.score_launches += x.score_launches; total.spa_launches += x.spa_launches;
		total.three_plane = std::max(total.three_plane, x.three_plane); total.n_unlisted += x.n_unlisted; total.n_guarded += x.n_guarded;   // (any chunk)
		return SGX_OK;
	};
	size_t prev_off = 0, prev_m = 0;
	int i = 0;
	for (size_t off = 0; off < M; off += chunk, i++) {
		const size_t m = std::min(chunk, M - off);
		const int b = i & 1;
		// ---- chunk i over PCIe on the copy stream (buffer b was last used by chunk i - 2: done)
		const uint8_t *src = reinterpret_cast<const uint8_t *>(rows) + off * src_row_bytes;
		if (src_row_bytes == dev_row_bytes) {
			HIPCHK(hipMemcpyAsync(h->pipe_in[b], src, m * dev_row_bytes, hipMemcpyHostToDevice, h->cstream));
		} else {
			if (dev_row_bytes > src_row_bytes) HIPCHK(hipMemsetAsync(h->pipe_in[b], 0, m * dev_row_bytes, h->cstream));
			HIPCHK(hipMemcpy2DAsync(h->pipe_in[b], dev_row_bytes, src, src_row_bytes,
				std::min(src_row_bytes, dev_row_bytes), m, hipMemcpyHostToDevice, h->cstream));
		}
		bool packed_ok = false;
		if (can_pack) {
			HIPCHK(hipMemsetAsync(h->pipe_flag, 0, sizeof(int), h->cstream));
			const dim3 g((unsigned)std::min<size_t>(64, (pk_row / 4 + 255) / 256), (unsigned)m);
			if (INPUT == IN_U8)
				hipLaunchKernelGGL((pack_rows_2bit<uint8_t>), g, dim3(256), 0, h->cstream,
					(const uint8_t *)h->pipe_in[b], N, h->pipe_pk[b], pk_row, h->pipe_flag);
			else
				hipLaunchKernelGGL((pack_rows_2bit<int>), g, dim3(256), 0, h->cstream,
					(const int *)h->pipe_in[b], N, h->pipe_pk[b], pk_row, h->pipe_flag);
			HIPCHK(hipGetLastError());
			HIPCHK(hipMemcpyAsync(h->h_pipe_flag, h->pipe_flag, sizeof(int), hipMemcpyDeviceToHost, h->cstream));
			HIPCHK(hipStreamSynchronize(h->cstream));
			packed_ok = *h->h_pipe_flag == 0;
		}
		double *as_f64 = nullptr;
		if (INPUT == IN_I32 && !packed_ok) {
			as_f64 = reinterpret_cast<double *>(h->pipe_in[b] + f64_off);
			hipLaunchKernelGGL(i32_rows_to_f64, dim3(1024), dim3(256), 0, h->cstream,
				(const int *)h->pipe_in[b], m * (size_t)N, as_f64);
			HIPCHK(hipGetLastError());
		}
		const bool as_block = blocks && (INPUT == IN_2BIT || packed_ok);
		HIPCHK(hipEventRecord(h->ev_h2d, h->cstream));
		// ---- chunk i - 1 has been computing meanwhile: collect it
		if (prev_m) { rc = harvest(prev_off, prev_m, b ^ 1); if (rc) return rc; }
		// ---- compute chunk i, results to pinned memory.  The chunk's 2-bit rows go into the buffer's block on
		// the COMPUTE stream: on the copy stream the 2.5 ms of ingest sat between two 9.5-ms copies and the
		// link idled a fifth of the time (43 GB/s; the next copy now starts as this one ends).
		HIPCHK(hipStreamWaitEvent(h->stream, h->ev_h2d, 0));
		HIPCHK(hipStreamWaitEvent(h->hstream, h->ev_h2d, 0));
		if (as_block) rc = scan_rows_dev(h, b, INPUT == IN_2BIT ? h->pipe_in[b] : h->pipe_pk[b], INPUT == IN_2BIT ? dev_row_bytes : pk_row, m,
			h->pipe_out[b], h->pipe_valid[b], false);
		else if (INPUT == IN_2BIT) rc = launch_scan<IN_2BIT>(h, h->pipe_in[b], dev_row_bytes, m, h->pipe_out[b], h->pipe_valid[b]);
		else if (packed_ok) rc = launch_scan<IN_2BIT>(h, h->pipe_pk[b], pk_row, m, h->pipe_out[b], h->pipe_valid[b]);
		else if (INPUT == IN_U8) rc = launch_scan<IN_U8>(h, h->pipe_in[b], dev_row_bytes, m, h->pipe_out[b], h->pipe_valid[b]);
		else if (INPUT == IN_I32) rc = launch_scan<IN_F64>(h, as_f64, (size_t)N * sizeof(double), m, h->pipe_out[b], h->pipe_valid[b]);
		else rc = launch_scan<IN_F64>(h, h->pipe_in[b], dev_row_bytes, m, h->pipe_out[b], h->pipe_valid[b]);
		if (rc) return rc;
		HIPCHK(hipMemcpyAsync(h->pin_out[b], h->pipe_out[b], m * 8 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
		HIPCHK(hipMemcpyAsync(h->pin_valid[b], h->pipe_valid[b], m, hipMemcpyDeviceToHost, h->stream));
		prev_off = off; prev_m = m;
	}
	rc = harvest(prev_off, prev_m, (i - 1) & 1);
	if (rc) return rc;
	h->stats = total;
	return SGX_OK;
}

// Page-locked host memory for the caller's block buffers: copies from it run at the full PCIe rate
// and truly asynchronously (a pageable source is staged by the runtime at ~50 GB/s).
extern "C" void *sgx_host_alloc(size_t bytes)
{
	void *p = nullptr;
	if (bytes == 0 || hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) {
		(void)fail(SGX_ENOMEM, "sgx_host_alloc: cannot pin %zu bytes", bytes);
		return nullptr;
	}
	return p;
}

extern "C" void sgx_host_free(void *p)
{
	if (p) (void)hipHostFree(p);
}

// Rows in host memory into a block: chunks cross PCIe on the copy stream while the previous chunk is being
// rearranged on the handle's stream; the lists are made once at the end.  Returns when the rows have left
// the caller's buffer; the block is ready for sgx_scan_block on this handle (same stream).
extern "C" int sgx_block_load(sgx_handle *h, sgx_block *b, const uint8_t *packed, size_t bpv, size_t M)
{
	if (!h || !b) return fail(SGX_EINVAL, "sgx_block_load: NULL argument");
	if (!packed) return fail(SGX_EINVAL, "sgx_block_load: NULL buffer");
	if (b->lists_only) return fail(SGX_EINVAL, "sgx_block_load: not a resident block");
	if (b->device != h->device) return fail(SGX_EINVAL, "sgx_block_load: block and handle are on different devices");
	if (M == 0 || M > b->cap) return fail(SGX_EINVAL, "sgx_block_load: %zu variants, the block holds up to %zu", M, b->cap);
	if (bpv < (size_t)(b->N + 3) / 4)
		return fail(SGX_EINVAL, "Invalid length of dosages: bytes_per_variant=%zu < ceil(N/4)=%zu", bpv, (size_t)(b->N + 3) / 4);
	int rc = set_dev(h);
	if (rc) return rc;
	const size_t dev_row = (size_t)b->ntile * 64;
	size_t chunk = std::max<size_t>(16, ((h->pipe_bytes ? h->pipe_bytes : PIPE_BYTES) / dev_row) & ~(size_t)15);
	chunk = std::min(chunk, (M + 15) & ~(size_t)15);
	rc = ensure_pipe(h, chunk * dev_row, 0, 1);
	if (rc) return rc;
	rc = block_begin_load(h, b, h->stream);
	if (rc) return rc;
	int i = 0;
	for (size_t off = 0; off < M; off += chunk, i++) {
		const size_t m = std::min(chunk, M - off);
		const int k = i & 1;
		if (i >= 2) HIPCHK(hipStreamWaitEvent(h->cstream, h->ev_done[k], 0));      // the buffer's previous chunk has been read
		const uint8_t *src = packed + off * bpv;
		if (bpv == dev_row) {
			HIPCHK(hipMemcpyAsync(h->pipe_in[k], src, m * dev_row, hipMemcpyHostToDevice, h->cstream));
		} else {
			if (dev_row > bpv) HIPCHK(hipMemsetAsync(h->pipe_in[k], 0, m * dev_row, h->cstream));
			HIPCHK(hipMemcpy2DAsync(h->pipe_in[k], dev_row, src, bpv, std::min(bpv, dev_row), m, hipMemcpyHostToDevice, h->cstream));
		}
		HIPCHK(hipEventRecord(h->ev_copy[k], h->cstream));
		HIPCHK(hipStreamWaitEvent(h->stream, h->ev_copy[k], 0));
		rc = block_put_rows(b, h->pipe_in[k], dev_row, off, m, h->stream);
		if (rc) return rc;
		HIPCHK(hipEventRecord(h->ev_done[k], h->stream));
	}
	rc = block_finish(b, M, h->stream);
	if (rc) return rc;
	HIPCHK(hipStreamSynchronize(h->cstream));      // the caller's buffer is free
	HIPCHK(hipStreamSynchronize(h->stream));
	return SGX_OK;
}

extern "C" int sgx_scan_2bit(sgx_handle *h, const uint8_t *packed, size_t bpv, size_t M,
	double *out8, uint8_t *valid)
{
	if (h && bpv < (size_t)(h->md.N + 3) / 4)
		return fail(SGX_EINVAL, "Invalid length of dosages: bytes_per_variant=%zu < ceil(N/4)=%zu",
			bpv, (size_t)(h->md.N + 3) / 4);
	return scan_host<IN_2BIT>(h, packed, bpv, h ? sgx_row_stride(h->md.N) : 0, M, out8, valid);
}

extern "C" int sgx_scan_u8(sgx_handle *h, const uint8_t *dosage, size_t M, double *out8, uint8_t *valid)
{
	const size_t rb = h ? (size_t)h->md.N : 0;
	return scan_host<IN_U8>(h, dosage, rb, rb, M, out8, valid);
}

extern "C" int sgx_scan_i32(sgx_handle *h, const int32_t *dosage, size_t M, double *out8, uint8_t *valid)
{
	const size_t rb = h ? (size_t)h->md.N * sizeof(int32_t) : 0;
	return scan_host<IN_I32>(h, dosage, rb, rb, M, out8, valid);
}

extern "C" int sgx_scan_f64(sgx_handle *h, const double *dosage, size_t M, double *out8, uint8_t *valid)
{
	const size_t rb = h ? (size_t)h->md.N * sizeof(double) : 0;
	return scan_host<IN_F64>(h, dosage, rb, rb, M, out8, valid);
}

// Burden rows from 2-bit genotypes, then the single-variant test on each row
// (saige_burden_test_bin/quant and the burden halves of ACAT-V / ACAT-O, saige_main.cpp:615-976)
extern "C" int sgx_burden_2bit(sgx_handle *h, const uint8_t *packed, size_t bpv, size_t n_variants,
	size_t n_rows, const int64_t *row_ptr, const int32_t *var_idx, const double *lut,
	double *out8, uint8_t *valid)
{
	if (!h) return fail(SGX_EINVAL, "sgx_burden_2bit: NULL handle");
	if (n_rows == 0) return SGX_OK;
	if (!packed || !row_ptr || !var_idx || !lut || !out8 || !valid)
		return fail(SGX_EINVAL, "sgx_burden_2bit: NULL buffer");
	const int N = h->md.N;
	if (bpv < (size_t)(N + 3) / 4)
		return fail(SGX_EINVAL, "Invalid length of dosages: bytes_per_variant=%zu < ceil(N/4)=%zu", bpv, (size_t)(N + 3) / 4);
	const int64_t nnz = row_ptr[n_rows];
	if (row_ptr[0] != 0 || nnz < 0) return fail(SGX_EINVAL, "sgx_burden_2bit: bad row_ptr");
	for (size_t r = 0; r < n_rows; r++)
		if (row_ptr[r + 1] < row_ptr[r]) return fail(SGX_EINVAL, "sgx_burden_2bit: row_ptr not ascending");
	for (int64_t e = 0; e < nnz; e++)
		if (var_idx[e] < 0 || (size_t)var_idx[e] >= n_variants)
			return fail(SGX_EINVAL, "sgx_burden_2bit: variant index %d out of range", var_idx[e]);
	int rc = set_dev(h);
	if (rc) return rc;
	h->last_issued = h;
	// device copies: packed rows (4-byte aligned stride), CSR, tables
	const size_t dbpv = ((size_t)(N + 15) / 16) * 4;
	const size_t o_ptr = (n_variants * dbpv + 15) & ~(size_t)15;
	const size_t o_idx = (o_ptr + (n_rows + 1) * sizeof(long long) + 15) & ~(size_t)15;
	const size_t o_lut = (o_idx + (size_t)std::max<int64_t>(nnz, 1) * sizeof(int) + 15) & ~(size_t)15;
	const size_t need = o_lut + (size_t)std::max<int64_t>(nnz, 1) * 4 * sizeof(double);
	if (need > h->stage_pk_cap) {
		HIPCHK(hipStreamSynchronize(h->stream));
		if (h->stage_pk) HIPCHK(hipFree(h->stage_pk));
		h->stage_pk = nullptr; h->stage_pk_cap = 0;
		HIPCHK(hipMalloc((void **)&h->stage_pk, need));
		h->stage_pk_cap = need;
	}
	HIPCHK(hipMemsetAsync(h->stage_pk, 0, n_variants * dbpv, h->stream));
	HIPCHK(hipMemcpy2DAsync(h->stage_pk, dbpv, packed, bpv, std::min(bpv, dbpv), n_variants, hipMemcpyHostToDevice, h->stream));
	std::vector<long long> rp(row_ptr, row_ptr + n_rows + 1);
	HIPCHK(hipMemcpyAsync(h->stage_pk + o_ptr, rp.data(), rp.size() * sizeof(long long), hipMemcpyHostToDevice, h->stream));
	if (nnz > 0) {
		HIPCHK(hipMemcpyAsync(h->stage_pk + o_idx, var_idx, (size_t)nnz * sizeof(int), hipMemcpyHostToDevice, h->stream));
		HIPCHK(hipMemcpyAsync(h->stage_pk + o_lut, lut, (size_t)nnz * 4 * sizeof(double), hipMemcpyHostToDevice, h->stream));
	}
	HIPCHK(hipStreamSynchronize(h->stream));      // rp is a local
	const size_t row_bytes = (size_t)N * sizeof(double);
	size_t chunk = std::max<size_t>(1, STAGE_BYTES / row_bytes);
	chunk = std::min<size_t>(std::min(chunk, n_rows), 65535);       // grid.y of burden_collapse_kernel
	rc = ensure_stage(h, chunk * row_bytes, chunk);
	if (rc) return rc;
	rc = ensure_recs(h, chunk);
	if (rc) return rc;
	sgx_stats total{};
	const int ndw = (N + 15) >> 4;
	for (size_t off = 0; off < n_rows; off += chunk) {
		const size_t m = std::min(chunk, n_rows - off);
		hipLaunchKernelGGL(burden_collapse_kernel, dim3((unsigned)((ndw + 255) / 256), (unsigned)m), dim3(256), 0, h->stream,
			h->stage_pk, dbpv, N, reinterpret_cast<const long long *>(h->stage_pk + o_ptr) + off,
			reinterpret_cast<const int *>(h->stage_pk + o_idx), reinterpret_cast<const double *>(h->stage_pk + o_lut),
			reinterpret_cast<double *>(h->stage_in), (size_t)N);
		HIPCHK(hipGetLastError());
		rc = launch_scan<IN_F64>(h, h->stage_in, row_bytes, m, h->stage_out, h->stage_valid);
		if (rc) return rc;
		HIPCHK(hipMemcpyAsync(out8 + off * 8, h->stage_out, m * 8 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
		HIPCHK(hipMemcpyAsync(valid + off, h->stage_valid, m, hipMemcpyDeviceToHost, h->stream));
		rc = sgx_sync(h);
		if (rc) return rc;
		total.n_variants += h->stats.n_variants; total.n_valid += h->stats.n_valid;
		total.n_spa += h->stats.n_spa; total.n_spa_dense += h->stats.n_spa_dense; total.n_spa_slow += h->stats.n_spa_slow;
		total.ms_score += h->stats.ms_score; total.ms_spa += h->stats.ms_spa; total.ms_total += h->stats.ms_total;
		total.score_launches += h->stats.score_launches; total.spa_launches += h->stats.spa_launches;
	}
	h->stats = total;
	return SGX_OK;
}
