// host_lanes.h -- options, lanes, synchronisation and stats.
// Part of libsaigehip.so: included by saigehip.hip (one translation unit), not a header of its own.
// Tuning / test hooks.  Unknown names are an error.
extern "C" int sgx_set_option(sgx_handle *h, const char *name, long long value)
{
	if (!h || !name) return fail(SGX_EINVAL, "sgx_set_option: NULL argument");
	const std::string n(name);
	if (n == "score_v1") h->force_v1 = value != 0;
	else if (n == "force_dense") h->force_dense = value != 0;
	else if (n == "spa_exact") h->force_exact = value != 0;
	else if (n == "pipe_mb") { if (value < 0 || value > 65536) return fail(SGX_EINVAL, "pipe_mb out of range"); h->pipe_bytes = (size_t)value << 20; return SGX_OK; }
	else if (n == "spa_abl") h->spa_abl = (int)value;
	else if (n == "guard_exp") { if (value < 0 || value > 300) return fail(SGX_EINVAL, "guard_exp must be 0..300"); h->guard_tol = std::pow(10.0, -(double)value); }
	else if (n == "three_plane") { if (value < -1 || value > 1) return fail(SGX_EINVAL, "three_plane must be -1 (automatic), 0 or 1"); h->dense_opt = (int)value; return SGX_OK; }
	else if (n == "lanes") {
		if (value < 1 || value > 4) return fail(SGX_EINVAL, "lanes must be 1..4");
		if (h->owner) return fail(SGX_EINVAL, "lanes: not on a twin");
		int rc = sgx_sync(h);
		if (rc) return rc;
		for (int i = (int)value - 1; i < 3; i++) if (h->twins[i]) { sgx_free(h->twins[i]); h->twins[i] = nullptr; }
		for (int i = 0; i < (int)value - 1; i++) {
			if (h->twins[i]) continue;
			sgx_handle *t = new sgx_handle();
			t->device = h->device; t->md = h->md; t->mf_ok = h->mf_ok; t->mfe = h->mfe;
			for (int g = 0; g < MF_MAXG; g++) { t->mf[g] = h->mf[g]; t->mf_nbfv[g] = h->mf_nbfv[g]; }
			t->dF = h->dF; t->dX = h->dX; t->dy = h->dy; t->dmu = h->dmu; t->dmu2 = h->dmu2; t->dXM = h->dXM; t->dFl = h->dFl; t->dQ = h->dQ;
			t->shares_model = true; t->owner = h;
			t->force_dense = h->force_dense; t->force_v1 = h->force_v1; t->force_exact = h->force_exact;
			t->spa_abl = h->spa_abl; t->guard_tol = h->guard_tol;
			rc = set_dev(t);
			if (!rc) rc = alloc_workspace(t);
			if (rc) { sgx_free(t); return rc; }
			h->twins[i] = t;
		}
		h->n_lanes = (int)value; h->next_lane = 0; h->last_issued = nullptr;
		return SGX_OK;
	}
	else return fail(SGX_EINVAL, "sgx_set_option: unknown option '%s'", name);
	for (sgx_handle *t : h->twins) if (t) { int rc = sgx_set_option(t, name, value); if (rc) return rc; }
	return SGX_OK;
}

// wait for this lane's work and turn its events / counters into stats
static int sync_lane(sgx_handle *h)
{
	int rc = set_dev(h);
	if (rc) return rc;
	HIPCHK(hipStreamSynchronize(h->stream));
	if (h->pend_dense.active) {
		h->pend_dense.active = false;
		if (h->stats_pending && h->h_counters[2] > 0) {
			rc = launch_pending_dense(h);
			if (rc) return rc;
			HIPCHK(hipStreamSynchronize(h->stream));
		}
	}
	if (h->stats_pending) {
		h->stats.n_spa = (uint64_t)(h->h_counters[0] + h->h_counters[7] - h->h_counters[6] + h->h_counters[5]);   // the tiers (handed-on copies once) + straight to exact
		h->stats.n_valid = (uint64_t)h->h_counters[1];
		h->stats.n_spa_dense = (uint64_t)h->h_counters[2];
		h->stats.n_spa_slow = (uint64_t)h->h_counters[4];
		h->stats.three_plane = h->used_miss ? 1u : 0u;
		h->stats.n_guarded = (uint32_t)h->h_counters[21];
		h->stats.n_unlisted = (uint32_t)h->h_counters[23] - h->stats.n_guarded;
		{
			// the step's missing genotypes (census of the epilogue, units of 64) decide the form of the NEXT row-major calls
			sgx_handle *p = h->owner ? h->owner : h;
			const double frac = 64.0 * (double)h->h_counters[22] / ((double)std::max<uint64_t>(1, h->stats.n_variants) * (double)h->md.N);
			const bool over = (uint64_t)h->stats.n_unlisted * 32 > h->stats.n_variants;
			if (!h->used_miss && (frac > SGX_DENSE_ON || over)) p->dense_mode = true;
			else if (h->used_miss && frac < SGX_DENSE_OFF) p->dense_mode = false;
		}
#ifdef SPA5_PROF
		fprintf(stderr, "routing: tier A %d, tier B %d (of them handed on by A: %d), per-variant kernels %d (series list %d, exact list %d), dense %d\n",
			h->h_counters[0], h->h_counters[7], h->h_counters[6], h->h_counters[5], h->h_counters[3], h->h_counters[4], h->h_counters[2]);
		fprintf(stderr, "spa5 phases (10 ns ticks summed over variants): series kernel %d variants: stage %d count %d index %d gather %d series %d (sweep %d sum %d solve %d) | exact kernel %d variants: stage %d count %d index %d gather %d - sweeps %d\n",
			h->h_counters[3], h->h_counters[8], h->h_counters[9], h->h_counters[10], h->h_counters[11], h->h_counters[12], h->h_counters[13], h->h_counters[14], h->h_counters[15],
			h->h_counters[4], h->h_counters[16], h->h_counters[17], h->h_counters[18], h->h_counters[19], h->h_counters[21]);
#endif
		float a = 0, b = 0, c = 0;
		(void)hipEventElapsedTime(&a, h->ev[0], h->ev[1]);
		(void)hipEventElapsedTime(&b, h->ev[1], h->ev[2]);
		(void)hipEventElapsedTime(&c, h->ev[0], h->ev[2]);
		h->stats.ms_score = a; h->stats.ms_spa = b; h->stats.ms_total = c;
		float k = 0;
		if (h->evk_set) (void)hipEventElapsedTime(&k, h->evk[0], h->evk[1]);
		h->stats.ms_kernel = k; h->evk_set = false;
		float l = 0;
		if (h->lists_timed) { (void)hipEventElapsedTime(&l, h->ev_lists, h->ev[0]); h->stats.ms_total += l; }
		h->stats.ms_lists = l; h->lists_timed = false;
		h->stats_pending = false;
		sgx_handle *p = h->owner ? h->owner : h;
		const sgx_stats &x = h->stats;
		p->total.n_variants += x.n_variants; p->total.n_valid += x.n_valid; p->total.n_spa += x.n_spa;
		p->total.n_spa_dense += x.n_spa_dense; p->total.n_spa_slow += x.n_spa_slow;
		p->total.ms_score += x.ms_score; p->total.ms_spa += x.ms_spa; p->total.ms_total += x.ms_total; p->total.ms_kernel += x.ms_kernel;
		p->total.ms_lists += x.ms_lists;
		p->total.score_launches += x.score_launches; p->total.spa_launches += x.spa_launches;
		p->total.three_plane += x.three_plane; p->total.n_unlisted += x.n_unlisted; p->total.n_guarded += x.n_guarded;
		p->total_calls++;
	}
	return SGX_OK;
}

extern "C" int sgx_sync(sgx_handle *h)
{
	if (!h) return fail(SGX_EINVAL, "sgx_sync: NULL handle");
	int rc = sync_lane(h);
	for (sgx_handle *t : h->twins) if (!rc && t) rc = sync_lane(t);
	return rc;
}

extern "C" int sgx_get_stats(sgx_handle *h, sgx_stats *st)
{
	if (!h || !st) return fail(SGX_EINVAL, "sgx_get_stats: NULL argument");
	int rc = sgx_sync(h);
	if (rc) return rc;
	*st = (h->last_issued && h->last_issued != h) ? h->last_issued->stats : h->stats;   // the most recent call
	return SGX_OK;
}

extern "C" int sgx_get_stats_total(sgx_handle *h, sgx_stats *st, uint64_t *n_calls, int reset)
{
	if (!h || !st) return fail(SGX_EINVAL, "sgx_get_stats_total: NULL argument");
	int rc = sgx_sync(h);
	if (rc) return rc;
	*st = h->total;
	if (n_calls) *n_calls = h->total_calls;
	if (reset) { h->total = sgx_stats{}; h->total_calls = 0; }
	return SGX_OK;
}
