// host_scan.h -- the block scan (contraction kernel forms, sparse pass, epilogue), lanes, the row-major device call.
// Part of libsaigehip.so: included by saigehip.hip (one translation unit), not a header of its own.

template <typename T>
static int ensure_buf(sgx_handle *h, T **p, size_t *cap, size_t need)
{
	if (need <= *cap) return SGX_OK;
	HIPCHK(hipStreamSynchronize(h->stream));
	if (h->hstream) HIPCHK(hipStreamSynchronize(h->hstream));
	if (h->s3_side) HIPCHK(hipStreamSynchronize(h->s3_side));
	if (*p) HIPCHK(hipFree(*p));
	*p = nullptr; *cap = 0;
	HIPCHK(hipMalloc((void **)p, need * sizeof(T)));
	*cap = need;
	return SGX_OK;
}

// Scan of a block (resident, or the lists of a row-major call with the caller's rows) on this lane's stream: sparse
// pass over the missing genotypes, contraction, reduction, epilogue, the FP64 kernel for what the lists do not
// cover, SPA stage.
// miss: the three-plane form -- the sums over the missing samples come out of the contraction kernel, the block's lists
// are not read (and need not exist)
// t3_done: the per-range sums over the missing samples are in h->s3_t3 already (scan_rows_dev's fused list + T3 pass)
static int launch_block_scan(sgx_handle *h, const sgx_block *b, size_t M, double *out8, uint8_t *valid, bool lazy_dense = false, bool miss = false,
	bool t3_done = false)
{
	const DevModel &md = h->md;
	const MfEpi &ep = h->mfe;
	hipStream_t st = h->hstream;
	const int NBF = h->mf_nbfv[0] + 1;
	const int grid = std::max(8, h->n_cu & ~7);
	const RowsRef rr = block_rows(b);
	const S3Lists L = block_lists(b);
	S3Plan pl{};
	int NCW = 0, NAFW = 0;
	const int slots = miss ? 2 * NBF - 1 : NBF;      // fragment slots of a variant's row of limb sums
	h->used_miss = miss;
	HIPCHK(hipStreamWaitEvent(st, b->ready, 0));
	HIPCHK(hipEventRecord(h->ev[0], st));            // (counters and queue cursors: zeroed by s3_reduce_kernel)
	int rc = ensure_buf(h, &h->s3_t3, &h->s3_t3_cap, (size_t)b->nr * M * md.P * 2);            // per-range sums over the missing samples (the epilogue adds them up)
	if (rc) return rc;
	rc = ensure_buf(h, &h->s3_ovf, &h->s3_ovf_cap, M);
	if (rc) return rc;
	if (miss) {
		switch (NBF) {
#define S3CASE(NBF_, NAF_, NC_, NLA_, NLB_, DA_, DB_)                                                         \
		case NBF_: {                                                                                          \
			NCW = NC_; NAFW = NAF_;                                                                           \
			pl = s3_plan(M, b->ntile, grid, NAF_ * NC_, rr.bpv);                                              \
			rc = ensure_buf(h, &h->s3_slabs, &h->s3_slabs_cap, (size_t)pl.ng * pl.ipg * NC_ * NAF_ * (2 * NBF_ - 1) * 256); \
			if (rc) return rc;                                                                                \
			const size_t lds = s3_lds_bytes(NBF_, NAF_, NC_, DA_, DB_);                                       \
			auto kern = score3_kernel<NBF_, NAF_, NC_, NLA_, NLB_, DA_, DB_, 0, 1, 2, 1, true>;                \
			if (!h->s3_attr_miss[NBF_]) {                                                                     \
				HIPCHK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
				h->s3_attr_miss[NBF_] = true;                                                                 \
			}                                                                                                 \
			HIPCHK(hipEventRecord(h->evk[0], st));                                                            \
			hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(64 * (NC_ + NLA_ + NLB_)), lds, st,            \
				rr.base, (const uint8_t *)h->dFl, pl, h->s3_slabs, (unsigned long long *)nullptr);            \
			HIPCHK(hipEventRecord(h->evk[1], st));                                                            \
			h->evk_set = true;                                                                                \
		} break;
			S3_FOR_EACH_NBF_MISS(S3CASE)
#undef S3CASE
		default: return fail(SGX_EINVAL, "score3: %d B fragments not supported", NBF);
		}
		HIPCHK(hipGetLastError());
	} else {
	// Sums over the missing samples, on the side stream, FIRST; the contraction kernel waits for them:
	//  * with few fragments (3 waves of ~154 registers per SIMD) the pass finds no room beside a resident
	//    contraction workgroup; launched second it would wait for the kernel's end;
	//  * from 7 fragments on (2 waves of <= 216 registers) one wave of the pass fits per SIMD, but in the kernel's
	//    shadow it slows the kernel by what it saves (K = 13, same box: 6.13 / 6.20 ms per step first, 6.23 / 6.23 after);
	//  * launched together onto an idle GPU the pass's 25 000 small workgroups and the kernel's 256 persistent ones
	//    fight for the CUs (kernel traces: 1.9 ms for the kernel and 1.1 ms for the pass in those steps).
	// (round 3, tools/README.md: the three orders measured)
	if (!t3_done) {
	HIPCHK(hipEventRecord(h->s3_fork, st));                  // (the side stream starts where this stream stands NOW)
	HIPCHK(hipStreamWaitEvent(h->s3_side, h->s3_fork, 0));
	{
		hipStream_t s2 = h->s3_side;
		const int PP = md.P <= 8 ? 8 : md.P <= 16 ? 16 : md.P <= 32 ? 32 : 64;
		const int tpw = 64 / PP;
		const unsigned chunks = (unsigned)((M + 4 * tpw - 1) / (4 * tpw));
		const dim3 g3(chunks * (unsigned)b->nr);
		if (PP == 8) hipLaunchKernelGGL(s3_t3_kernel<8>, g3, dim3(256), 0, s2, (int)M, md.P, h->dQ, L, h->s3_t3);
		else if (PP == 16) hipLaunchKernelGGL(s3_t3_kernel<16>, g3, dim3(256), 0, s2, (int)M, md.P, h->dQ, L, h->s3_t3);
		else if (PP == 32) hipLaunchKernelGGL(s3_t3_kernel<32>, g3, dim3(256), 0, s2, (int)M, md.P, h->dQ, L, h->s3_t3);
		else hipLaunchKernelGGL(s3_t3_kernel<64>, g3, dim3(256), 0, s2, (int)M, md.P, h->dQ, L, h->s3_t3);
		HIPCHK(hipGetLastError());
		HIPCHK(hipEventRecord(h->s3_join, s2));
	}
	HIPCHK(hipStreamWaitEvent(st, h->s3_join, 0));
	}
	switch (NBF) {
#define S3CASE(NBF_, NAF_, NC_, NLA_, NLB_, DA_, DB_)                                                         \
	case NBF_: {                                                                                          \
		NCW = NC_; NAFW = NAF_;                                                                           \
		pl = s3_plan(M, b->ntile, grid, NAF_ * NC_, rr.bpv);                                              \
		rc = ensure_buf(h, &h->s3_slabs, &h->s3_slabs_cap, (size_t)pl.ng * pl.ipg * NC_ * NAF_ * NBF_ * 256); \
		if (rc) return rc;                                                                                \
		const size_t lds = s3_lds_bytes(NBF_, NAF_, NC_, DA_, DB_);                                       \
		auto kern = score3_kernel<NBF_, NAF_, NC_, NLA_, NLB_, DA_, DB_, 0, 1, 2, 1>;                      \
		if (!h->s3_attr[NBF_]) {                                                                          \
			HIPCHK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
			h->s3_attr[NBF_] = true;                                                                      \
		}                                                                                                 \
		HIPCHK(hipEventRecord(h->evk[0], st));                                                            \
		hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(64 * (NC_ + NLA_ + NLB_)), lds, st,            \
			rr.base, (const uint8_t *)h->dFl, pl, h->s3_slabs, (unsigned long long *)nullptr);            \
		HIPCHK(hipEventRecord(h->evk[1], st));                                                            \
		h->evk_set = true;                                                                                \
	} break;
		S3_FOR_EACH_NBF(S3CASE)
#undef S3CASE
	default: return fail(SGX_EINVAL, "score3: %d B fragments not supported", NBF);
	}
	HIPCHK(hipGetLastError());
	}
	const int acc_stride = 16 * slots;
	{
		const int per = NCW * NAFW * slots * 256;
		hipLaunchKernelGGL(s3_reduce_kernel, dim3((unsigned)((per / 4 + 255) / 256), (unsigned)pl.vt), dim3(256), 0, st,
			pl, (int)M, NCW, NAFW, slots, 1, h->s3_slabs, h->mf_acc, acc_stride, h->counters, h->cur5);
	}
	const int btop = md.quant ? 0 : (int)(2 * M);
	switch (md.K) {
#define ECASE(KK) case KK:                                                                     \
	hipLaunchKernelGGL((score3_epilogue<KK>), dim3((unsigned)((M + s3e_vb(KK) - 1) / s3e_vb(KK))), dim3(s3e_vb(KK)), 0, st, (int)M, md, ep, h->mf_acc, acc_stride, \
		miss ? 16 * NBF : 0, h->s3_t3, b->nr, L.lcnt, L.ld, h->s3_ovf, h->recs, h->counters, btop, h->fb_spa2, h->fb_x2, out8, valid, h->guard_tol, (h->owner ? h->owner : h)->spa_abl >> 16); \
	hipLaunchKernelGGL((score2b_kernel<2 * KK + 2, 256>), dim3((unsigned)std::min<size_t>(M, 4 * (size_t)h->n_cu)), dim3(256), 0, st, \
		rr, (int)M, md, h->recs, h->counters, out8, valid, (const int *)h->s3_ovf, 23, btop, h->fb_spa2, h->fb_x2); \
	break;
	FOR_EACH_K(ECASE)
#undef ECASE
	default: return fail(SGX_EINVAL, "score3: unsupported K=%d", md.K);
	}
	HIPCHK(hipGetLastError());
	HIPCHK(hipEventRecord(h->ev[1], st));
	h->stats.score_launches = 5;
	st = h->stream;                                  // the SPA stage: low priority, behind the score chain
	HIPCHK(hipStreamWaitEvent(st, h->ev[1], 0));
	rc = launch_spa<IN_2BIT>(h, rr, M, out8, lazy_dense);
	if (rc) return rc;
	if (lazy_dense) h->pend_dense.blk = b;
	HIPCHK(hipEventRecord(h->ev[2], st));
	HIPCHK(hipEventRecord(b->last_read, st));
	const_cast<sgx_block *>(b)->was_read = true;
	HIPCHK(hipMemcpyAsync(h->h_counters, h->counters, 24 * sizeof(int), hipMemcpyDeviceToHost, st));
	h->stats.n_variants = M;
	h->stats_pending = true;
	return SGX_OK;
}

// picks the lane of the next device-resident call (two lanes alternate) and makes it ready for M variants
static int next_lane(sgx_handle *h, size_t M, sgx_handle **lane_out)
{
	sgx_handle *lane = h, *other = nullptr;
	if (h->n_lanes > 1) {
		lane = h->next_lane ? h->twins[h->next_lane - 1] : h;
		other = h->last_issued;                    // the lane of the previous call
		h->next_lane = (h->next_lane + 1) % h->n_lanes;
	}
	int rc = sync_lane(lane);            // the lane's previous call is done: keep its stats (events are reused)
	if (rc) return rc;
	h->last_issued = lane;
	rc = ensure_recs(lane, M);
	if (rc) return rc;
	// score stages do not overlap: this one starts after the other lane's has ended
	if (other && other != lane && other->stats_pending) HIPCHK(hipStreamWaitEvent(lane->hstream, other->ev[1], 0));
	*lane_out = lane;
	return SGX_OK;
}

extern "C" int sgx_scan_block(sgx_handle *h, const sgx_block *b, double *out8_dev, uint8_t *valid_dev)
{
	if (!h || !b) return fail(SGX_EINVAL, "sgx_scan_block: NULL argument");
	if (!out8_dev || !valid_dev) return fail(SGX_EINVAL, "sgx_scan_block: NULL buffer");
	if (b->lists_only) return fail(SGX_EINVAL, "sgx_scan_block: not a resident block");
	if (b->M == 0) return SGX_OK;
	if (b->device != h->device) return fail(SGX_EINVAL, "sgx_scan_block: block and handle are on different devices");
	if (b->N != h->md.N) return fail(SGX_EINVAL, "sgx_scan_block: the block holds rows of %d samples, the model has %d", b->N, h->md.N);
	int rc = set_dev(h);
	if (rc) return rc;
	sgx_handle *lane = nullptr;
	rc = next_lane(h, b->M, &lane);
	if (rc) return rc;
	if (!h->mf_ok || h->force_v1) {
		// FP64 kernels on the tiled rows (test hook; models outside the fixed-point form's range)
		hipStream_t st = lane->stream;
		HIPCHK(hipStreamWaitEvent(st, b->ready, 0));
		HIPCHK(hipMemsetAsync(lane->counters, 0, 24 * sizeof(int), st));
		if (lane->cur5) HIPCHK(hipMemsetAsync(lane->cur5, 0, 8 * sizeof(int), st));
		HIPCHK(hipEventRecord(lane->ev[0], st));
		RowsRef rr = block_rows(b);
		rr.cptr = nullptr; rr.cidx = nullptr; rr.corient = nullptr;
		switch (lane->md.K) {
#define VCASE(KK) case KK: hipLaunchKernelGGL((score2b_kernel<2 * KK + 2, 256>), dim3((unsigned)b->M), dim3(256), 0, st, rr, (int)b->M, lane->md, \
	lane->recs, lane->counters, out8_dev, valid_dev, (const int *)nullptr, 0, 0, (int *)nullptr, (int *)nullptr); break;
		FOR_EACH_K(VCASE)
#undef VCASE
		}
		HIPCHK(hipGetLastError());
		HIPCHK(hipEventRecord(lane->ev[1], st));
		lane->stats.score_launches = 1;
		rc = launch_spa<IN_2BIT>(lane, rr, b->M, out8_dev);
		if (rc) return rc;
		HIPCHK(hipEventRecord(lane->ev[2], st));
		HIPCHK(hipEventRecord(b->last_read, st));
		const_cast<sgx_block *>(b)->was_read = true;
		HIPCHK(hipMemcpyAsync(lane->h_counters, lane->counters, 24 * sizeof(int), hipMemcpyDeviceToHost, st));
		lane->stats.n_variants = b->M;
		lane->stats_pending = true;
		return SGX_OK;
	}
	// a block with many missing genotypes (or variants its pool had no room for) takes the three-plane form
	sgx_block *bw = const_cast<sgx_block *>(b);
	if (!bw->info_read) {
		HIPCHK(hipEventSynchronize(b->ready));
		const double frac = 64.0 * (double)b->h_info[0] / ((double)b->M * (double)b->N);
		bw->dense = frac > SGX_DENSE_ON || (size_t)b->h_info[1] * 32 > b->M;
		bw->info_read = true;
	}
	const bool miss = h->dense_opt >= 0 ? h->dense_opt != 0 : b->dense;
	return launch_block_scan(lane, b, b->M, out8_dev, valid_dev, true, miss);
}

// Which form of the contraction kernel a row-major call takes.  The two-plane form needs the positions of the missing
// genotypes -- a pass over the rows (s3_lists_t3_kernel) whose cost grows with their number; the three-plane form needs
// none, at ~1.7 x the MFMAs.  Large kernels of different streams do not share the machine (tools/README.md, round 4: a
// kernel that holds every CU keeps the other queue's out), so what counts is the sum of the two kernels' times:
//   K = 3 binary (4 B fragments), N = 430 000: list pass 0.89 + two planes 1.04 ms against three planes 1.55-1.75 ms:
//     step 3.19-3.23 -> 2.88-3.01 ms (same box), N = 50 000: 0.676 -> 0.620; K = 2: 3.11 -> 2.84
//   K = 4 (5 fragments): 3.56 against 3.63, K = 5: 3.94 against 4.02 ms: the two-plane form, until more than ~0.5 % of
//     the genotypes are missing -- where the sparse sums have grown to the difference and the segments' room
//     (S3_LT_CAP entries) is about to overflow.
static bool rows_take_three_planes(const sgx_handle *lane)
{
	const sgx_handle *p = lane->owner ? lane->owner : lane;
	if (p->dense_opt >= 0) return p->dense_opt != 0;
	return lane->mf_nbfv[0] + 1 <= 4 || p->dense_mode;
}

// the lists of this lane's row-major calls (the rows stay where the caller has them)
static int ensure_tmp_block(sgx_handle *lane, int which, size_t M)
{
	sgx_block *&tb = lane->tmp_blk[which];
	if (tb && tb->cap >= M) return SGX_OK;
	HIPCHK(hipStreamSynchronize(lane->stream));
	HIPCHK(hipStreamSynchronize(lane->hstream));
	if (tb) { sgx_block_free(tb); tb = nullptr; }
	return block_create(lane->md.N, M, lane->device, true, 0, &tb);
}

// row-major rows on the device -> table: one pass over the rows for the lists of the missing genotypes, then the
// scan reads the rows where they are
static int scan_rows_dev(sgx_handle *lane, int which, const uint8_t *rows_dev, size_t bpv, size_t M, double *out8, uint8_t *valid, bool lazy_dense)
{
	int rc = ensure_tmp_block(lane, which, M);
	if (rc) return rc;
	sgx_block *tb = lane->tmp_blk[which];
	tb->ext_rows = rows_dev; tb->ext_bpv = bpv;
	if (rows_take_three_planes(lane)) return launch_block_scan(lane, tb, M, out8, valid, lazy_dense, true);
	// one pass over the rows: the missing genotypes of every (range, variant), their sums of Q gathered on the spot
	rc = ensure_buf(lane, &lane->s3_t3, &lane->s3_t3_cap, (size_t)tb->nr * M * lane->md.P * 2);
	if (rc) return rc;
	HIPCHK(hipEventRecord(lane->ev_lists, lane->hstream));
	lane->lists_timed = true;
	{
		const S3Lists L = block_lists(tb);
		const int P = lane->md.P, PP = P <= 8 ? 8 : P <= 16 ? 16 : P <= 32 ? 32 : 64;
		const dim3 grid((unsigned)((M + 3) / 4), (unsigned)tb->nr);
		hipStream_t st = lane->hstream;
		if (PP == 8) hipLaunchKernelGGL((s3_lists_t3_kernel<8, 8>), grid, dim3(256), 0, st, rows_dev, bpv, tb->N, (int)M, tb->ntile, L, P, lane->dQ, lane->s3_t3);
		else if (PP == 16) hipLaunchKernelGGL((s3_lists_t3_kernel<8, 16>), grid, dim3(256), 0, st, rows_dev, bpv, tb->N, (int)M, tb->ntile, L, P, lane->dQ, lane->s3_t3);
		else if (PP == 32) hipLaunchKernelGGL((s3_lists_t3_kernel<8, 32>), grid, dim3(256), 0, st, rows_dev, bpv, tb->N, (int)M, tb->ntile, L, P, lane->dQ, lane->s3_t3);
		else hipLaunchKernelGGL((s3_lists_t3_kernel<8, 64>), grid, dim3(256), 0, st, rows_dev, bpv, tb->N, (int)M, tb->ntile, L, P, lane->dQ, lane->s3_t3);
		HIPCHK(hipGetLastError());
	}
	// (no kernel between this pass and the contraction: the epilogue adds up the ranges' sums and counts itself)
	return launch_block_scan(lane, tb, M, out8, valid, lazy_dense, false, true);
}


extern "C" int sgx_scan_2bit_dev(sgx_handle *h, const uint8_t *packed_dev, size_t bpv,
	size_t M, double *out8_dev, uint8_t *valid_dev)
{
	if (!h) return fail(SGX_EINVAL, "sgx_scan_2bit_dev: NULL handle");
	if (M == 0) return SGX_OK;
	if (!packed_dev || !out8_dev || !valid_dev)
		return fail(SGX_EINVAL, "sgx_scan_2bit_dev: NULL buffer");
	if (M > 0x7fffffffu / S3_NR) return fail(SGX_EINVAL, "sgx_scan_2bit_dev: too many variants in one call");
	if (bpv % 64 != 0 || bpv < sgx_row_stride(h->md.N))
		return fail(SGX_EINVAL, "Invalid length of dosages: bytes_per_variant=%zu, need a multiple of 64 >= %zu",
			bpv, sgx_row_stride(h->md.N));
	if (((uintptr_t)packed_dev & 15u) != 0)
		return fail(SGX_EINVAL, "sgx_scan_2bit_dev: packed_dev must be 16-byte aligned");
	int rc = set_dev(h);
	if (rc) return rc;
	sgx_handle *lane = nullptr;
	rc = next_lane(h, M, &lane);
	if (rc) return rc;
	if (!h->mf_ok || h->force_v1) return launch_scan<IN_2BIT>(lane, packed_dev, bpv, M, out8_dev, valid_dev);
	return scan_rows_dev(lane, 0, packed_dev, bpv, M, out8_dev, valid_dev, true);
}
