// host_util.h -- GDS dBit2 decoder, genotype counts, synthetic genotypes, self-test.
// Part of libsaigehip.so: included by saigehip.hip (one translation unit), not a header of its own.

// ---------------------------------------------------------------------------
// Host-side decoder of SeqArray's genotype/data (dBit2 [variant][sample][ploidy], 2 bits per allele, LSB
// first) into 2-bit dosage rows: code = number of non-reference alleles, 3 = missing (any allele missing) --
// what seqGetData(gds, "$dosage_alt") yields and seqApply hands the reference as RAW (R/assoc_single.r:202-221).
// alleles: the bytes that hold variants [0, m) starting at bit `bit0` of the first byte (rows are 4 n_samp
// bits and need not be whole bytes).  sel: n_sel sample indices to keep, in the order wanted (NULL: all).
// out: m rows of out_stride bytes; bytes beyond a row's codes are zeroed.  Rows are split over `threads`
// host threads (0 = one per hardware thread, at most 16).
extern "C" int sgx_decode_dbit2(const uint8_t *alleles, size_t bit0, int32_t n_samp, size_t m,
	const int64_t *sel, int32_t n_sel, uint8_t *out, size_t out_stride, int threads)
{
	if (!alleles || !out) return fail(SGX_EINVAL, "sgx_decode_dbit2: NULL buffer");
	if (n_samp <= 0 || (sel && n_sel <= 0)) return fail(SGX_EINVAL, "sgx_decode_dbit2: no samples");
	const size_t n_out = sel ? (size_t)n_sel : (size_t)n_samp, nb = (n_out + 3) / 4;
	if (out_stride < nb) return fail(SGX_EINVAL, "sgx_decode_dbit2: out_stride %zu < %zu", out_stride, nb);
	if (sel) for (int32_t k = 0; k < n_sel; k++)
		if (sel[k] < 0 || sel[k] >= n_samp) return fail(SGX_EINVAL, "sgx_decode_dbit2: sample index %lld out of range", (long long)sel[k]);
	// nibble (two allele codes of one sample) -> dosage code
	uint8_t nib[16];
	for (int v = 0; v < 16; v++) {
		const int a0 = v & 3, a1 = v >> 2;
		nib[v] = (a0 == 3 || a1 == 3) ? 3 : (uint8_t)((a0 != 0) + (a1 != 0));
	}
	// two bytes (four samples) -> one packed byte
	static std::vector<uint8_t> lut16;
	static std::once_flag once;
	std::call_once(once, [&]() {
		lut16.resize(65536);
		for (int w = 0; w < 65536; w++)
			lut16[w] = (uint8_t)(nib[w & 15] | (nib[(w >> 4) & 15] << 2) | (nib[(w >> 8) & 15] << 4) | (nib[w >> 12] << 6));
	});
	const size_t row_bits = (size_t)n_samp * 4;
	int T = threads > 0 ? threads : (int)std::min<unsigned>(16u, std::max(1u, std::thread::hardware_concurrency()));
	T = (int)std::min<size_t>((size_t)T, std::max<size_t>(1, m));
	auto work = [&](size_t r0, size_t r1) {
		for (size_t r = r0; r < r1; r++) {
			const size_t b0 = bit0 + r * row_bits;
			uint8_t *o = out + r * out_stride;
			if (!sel && (b0 & 7) == 0) {
				const uint8_t *p = alleles + (b0 >> 3);
				const size_t full = (size_t)n_samp / 4;
				for (size_t k = 0; k < full; k++) o[k] = lut16[(size_t)p[2 * k] | ((size_t)p[2 * k + 1] << 8)];
				if (full < nb) {                                  // the last 1..3 samples
					uint8_t v = 0;
					for (size_t s = 4 * full; s < (size_t)n_samp; s++) {
						const size_t bit = b0 + 4 * s;
						v |= (uint8_t)(nib[(alleles[bit >> 3] >> (bit & 7)) & 15] << (2 * (s & 3)));
					}
					o[full] = v;
				}
			} else {
				for (size_t k = 0; k < nb; k++) {
					uint8_t v = 0;
					for (size_t q = 0; q < 4 && 4 * k + q < n_out; q++) {
						const size_t s = sel ? (size_t)sel[4 * k + q] : 4 * k + q;
						const size_t bit = b0 + 4 * s;               // a nibble never straddles a byte: bit0 and 4 s are multiples of 4
						v |= (uint8_t)(nib[(alleles[bit >> 3] >> (bit & 7)) & 15] << (2 * q));
					}
					o[k] = v;
				}
			}
			if (out_stride > nb) memset(o + nb, 0, out_stride - nb);
		}
	};
	if ((bit0 & 3) != 0) return fail(SGX_EINVAL, "sgx_decode_dbit2: bit0 must be a multiple of 4");
	if (T <= 1) { work(0, m); return SGX_OK; }
	std::vector<std::thread> th;
	for (int t = 0; t < T; t++) th.emplace_back(work, m * t / T, m * (t + 1) / T);
	for (auto &x : th) x.join();
	return SGX_OK;
}

// per-variant n_valid and allele sum of a host 2-bit matrix (no model handle needed)
extern "C" int sgx_geno_stats_2bit(const uint8_t *packed, size_t bpv, int32_t n_samp, size_t n_variants,
	int device, int32_t *n_valid, int32_t *allele_sum)
{
	if (n_variants == 0) return SGX_OK;
	if (!packed || !n_valid || !allele_sum) return fail(SGX_EINVAL, "sgx_geno_stats_2bit: NULL buffer");
	if (n_samp <= 0 || bpv < (size_t)(n_samp + 3) / 4)
		return fail(SGX_EINVAL, "Invalid length of dosages: bytes_per_variant=%zu < ceil(N/4)", bpv);
	HIPCHK(hipSetDevice(device));
	const size_t dbpv = ((size_t)(n_samp + 15) / 16) * 4;
	const size_t chunk = std::max<size_t>(1, std::min<size_t>(n_variants, ((size_t)1 << 30) / dbpv));
	uint8_t *dpk = nullptr; int *dn = nullptr, *ds = nullptr;
	HIPCHK(hipMalloc((void **)&dpk, chunk * dbpv));
	hipError_t e = hipMalloc((void **)&dn, chunk * sizeof(int));
	if (e == hipSuccess) e = hipMalloc((void **)&ds, chunk * sizeof(int));
	int rc = SGX_OK;
	for (size_t off = 0; off < n_variants && e == hipSuccess; off += chunk) {
		const size_t m = std::min(chunk, n_variants - off);
		e = hipMemset(dpk, 0, m * dbpv);
		if (e == hipSuccess) e = hipMemcpy2D(dpk, dbpv, packed + off * bpv, bpv, std::min(bpv, dbpv), m, hipMemcpyHostToDevice);
		if (e != hipSuccess) break;
		hipLaunchKernelGGL(geno_stats_kernel, dim3((unsigned)m), dim3(256), 0, 0, dpk, dbpv, (int)n_samp, dn, ds);
		e = hipGetLastError();
		if (e == hipSuccess) e = hipMemcpy(n_valid + off, dn, m * sizeof(int), hipMemcpyDeviceToHost);
		if (e == hipSuccess) e = hipMemcpy(allele_sum + off, ds, m * sizeof(int), hipMemcpyDeviceToHost);
	}
	(void)hipFree(dpk); (void)hipFree(dn); (void)hipFree(ds);
	if (e != hipSuccess) rc = fail(SGX_EHIP, "sgx_geno_stats_2bit: %s", hipGetErrorString(e));
	return rc;
}

extern "C" int sgx_synth_2bit_dev(sgx_handle *h, uint8_t *packed_dev, size_t bpv, int32_t n_samp,
	size_t M, uint64_t first_variant, uint64_t seed, const uint32_t *thr_dev)
{
	if (!h || !packed_dev || !thr_dev) return fail(SGX_EINVAL, "sgx_synth_2bit_dev: NULL argument");
	if (bpv % 4 != 0 || bpv < (size_t)(n_samp + 3) / 4)
		return fail(SGX_EINVAL, "sgx_synth_2bit_dev: bad bytes_per_variant %zu", bpv);
	int rc = set_dev(h);
	if (rc) return rc;
	const size_t MAXY = 32768;
	for (size_t off = 0; off < M; off += MAXY) {
		const size_t m = std::min(MAXY, M - off);
		const int nd = (int)(bpv / 4);
		const dim3 grid((unsigned)std::min(64, (nd + 255) / 256), (unsigned)m);
		hipLaunchKernelGGL(synth2b_kernel, grid, dim3(256), 0, h->stream, packed_dev + off * bpv,
			bpv, (int)n_samp, m, first_variant + off, seed, thr_dev + 3 * off);
		HIPCHK(hipGetLastError());
	}
	return SGX_OK;
}

// Checks the operand/result lane maps of v_mfma_i32_16x16x64_i8 that the MFMA
// score path relies on, with asymmetric integer data.
extern "C" int sgx_selftest(int device)
{
	HIPCHK(hipSetDevice(device));
	std::vector<int8_t> A(16 * 64), B(64 * 16);
	std::vector<int> D(256), R(256, 0);
	uint64_t x = 12345;
	for (auto &v : A) { x = splitmix64(x); v = (int8_t)(x & 3); }
	for (auto &v : B) { x = splitmix64(x); v = (int8_t)(x & 0xFF); }
	for (int i = 0; i < 16; i++) for (int j = 0; j < 16; j++) for (int k = 0; k < 64; k++)
		R[i * 16 + j] += (int)A[i * 64 + k] * (int)B[k * 16 + j];
	int8_t *dA, *dB; int *dD;
	HIPCHK(hipMalloc((void **)&dA, A.size())); HIPCHK(hipMalloc((void **)&dB, B.size()));
	HIPCHK(hipMalloc((void **)&dD, 256 * sizeof(int)));
	HIPCHK(hipMemcpy(dA, A.data(), A.size(), hipMemcpyHostToDevice));
	HIPCHK(hipMemcpy(dB, B.data(), B.size(), hipMemcpyHostToDevice));
	hipLaunchKernelGGL(mfma_selftest_kernel, dim3(1), dim3(64), 0, 0, dA, dB, dD);
	HIPCHK(hipDeviceSynchronize());
	HIPCHK(hipMemcpy(D.data(), dD, 256 * sizeof(int), hipMemcpyDeviceToHost));
	(void)hipFree(dA); (void)hipFree(dB); (void)hipFree(dD);
	for (int i = 0; i < 256; i++)
		if (D[i] != R[i]) return fail(SGX_EHIP, "sgx_selftest: MFMA i8 lane map mismatch at %d: %d != %d", i, D[i], R[i]);

	// fast_exp / fast_log (kern_spa2.h) against the host libm
	const int NT = 8192;
	std::vector<double> xin(2 * NT), yout(2 * NT);
	for (int i = 0; i < NT; i++) {
		x = splitmix64(x);
		const double u = (double)(x >> 11) / 9007199254740992.0;
		xin[i] = (i < 16) ? (double[]){0.0, -0.0, 1.0, -1.0, 709.7, 709.79, -745.0, -745.2, -708.4, 1e-300, -1e-300, 0.34657, -0.34657, 50.0, -50.0, 710.0}[i]
			: (i & 1 ? -745.0 + u * 1455.0 : -2.0 + 4.0 * u);
		x = splitmix64(x);
		const double w = (double)(x >> 11) / 9007199254740992.0;
		xin[NT + i] = (i < 8) ? (double[]){1.0, 0.5, 2.0, 0.70710678118654746, 0.70710678118654757, 1.0000000000000002, 0.99999999999999989, 1e308}[i]
			: (i & 1 ? std::pow(10.0, -300.0 + 600.0 * w) : 1.0 + (w - 0.5) * std::pow(10.0, -(double)(i % 16)));
	}
	double *dx, *dy;
	HIPCHK(hipMalloc((void **)&dx, xin.size() * sizeof(double)));
	HIPCHK(hipMalloc((void **)&dy, xin.size() * sizeof(double)));
	HIPCHK(hipMemcpy(dx, xin.data(), xin.size() * sizeof(double), hipMemcpyHostToDevice));
	hipLaunchKernelGGL(fastmath_selftest_kernel, dim3((2 * NT + 255) / 256), dim3(256), 0, 0, dx, dy, NT);
	HIPCHK(hipDeviceSynchronize());
	HIPCHK(hipMemcpy(yout.data(), dy, yout.size() * sizeof(double), hipMemcpyDeviceToHost));
	(void)hipFree(dx); (void)hipFree(dy);
	for (int i = 0; i < 2 * NT; i++) {
		const double ref = (i < NT) ? std::exp(xin[i]) : std::log(xin[i]);
		const double got = yout[i];
		const bool same = (ref == got) || (std::isnan(ref) && std::isnan(got));
		// log near 1 is tiny: allow an absolute 4e-17 there, else 1.5e-15 relative
		const double tol = 1.5e-15 * std::fabs(ref) + ((i >= NT) ? 4e-17 : 0.0) + ((i < NT && ref < 1e-300) ? 1e-320 : 0.0);
		if (!same && !(std::fabs(got - ref) <= tol))
			return fail(SGX_EHIP, "sgx_selftest: fast %s(%.17g) = %.17g, libm %.17g", i < NT ? "exp" : "log", xin[i], got, ref);
	}
	return SGX_OK;
}
