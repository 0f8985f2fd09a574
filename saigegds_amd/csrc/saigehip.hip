// saigehip.hip -- MI355X (gfx950) kernels + C ABI of the SAIGEgds single-variant
// scan.  Interface and reference citations: include/saigehip.h; design,
// data layout and rooflines: DESIGN.md.
//
// Math of the score stage.  For a variant with (imputed, flipped) dosage G the
// reference computes (src/saige_main.cpp:300-350, either branch)
//     adj = G - X c',  c' = (X'VX)^-1 X'V G,  S = sum y_mu*adj,
//     var = r * sum mu2*adj^2.
// Both branches reduce to four carrier-only sums over I = {i : G_i != 0}:
//     c'_k = sum_I G_i A[i,k]          A = t_XVX_inv_XV
//     e_k  = sum_I G_i mu2_i X[i,k]
//     s    = sum_I G_i y_mu_i
//     w    = sum_I G_i^2 mu2_i
//     var2 = c'^T XVX c' + w - 2 e.c'        S = s - S_a.c'
// (quantitative: mu2 == 1).  The per-sample vector F[i] = [A[i,:], mu2_i X[i,:],
// y_mu_i, mu2_i] (P = 2K+2 doubles, one 16-byte aligned row) is what a carrier
// costs in memory traffic.

#include <hip/hip_runtime.h>
#include <math.h>
#include <cmath>
#include <algorithm>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "saigehip.h"

#define KMAX SGX_MAX_COEFF
#define WAVE 64

// ---------------------------------------------------------------------------
// error handling

static thread_local std::string g_err;

static int fail(int code, const char *fmt, ...)
{
	char buf[512];
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(buf, sizeof buf, fmt, ap);
	va_end(ap);
	g_err = buf;
	return code;
}

#define HIPCHK(expr)                                                              \
	do {                                                                          \
		hipError_t e_ = (expr);                                                   \
		if (e_ != hipSuccess)                                                     \
			return fail(SGX_EHIP, "%s failed: %s (%s:%d)", #expr,                 \
				hipGetErrorString(e_), __FILE__, __LINE__);                       \
	} while (0)

#include "dev_common.h"
#include "kern_score.h"
#include "kern_score_mfma.h"
#include "kern_score3.h"
#include "kern_spa.h"
#include "kern_spa2.h"
#include "kern_spa4.h"
#include "kern_synth.h"
#include "kern_grm.h"
#include "kern_burden.h"
#include "kern_pack.h"

// ---------------------------------------------------------------------------
// host side

// A block of variants: the 2-bit rows as they came (row-major: the contraction kernel's loaders read them as they
// are) with the sparse side the scan needs -- the positions of the missing genotypes and, in a resident block,
// the carrier lists of the rare variants (kern_lists.h).  Depends on the number of samples only: one block can be
// scanned with any model of that many samples.  lists_only: the scratch of the row-major scan calls -- the rows
// stay where the caller has them (ext_rows), no carrier lists.
struct sgx_block {
	int device = 0;
	int N = 0, ntile = 0, nr = 1;
	size_t cap = 0;              // variants it can hold
	size_t M = 0;                // variants loaded
	bool lists_only = false;
	uint8_t *rows = nullptr;     // [cap][bpv] the block's copy of the rows
	size_t bpv = 0;              // bytes per row of that copy = sgx_row_stride(N)
	const uint8_t *ext_rows = nullptr; size_t ext_bpv = 0;   // lists_only: the rows of the scan in flight
	// missing genotypes (S3Lists)
	unsigned *idx = nullptr; size_t idx_cap = 0;
	unsigned *cursor = nullptr;  // [S3_NSUB x S3_CURSOR_STRIDE]
	unsigned *lstart = nullptr;  // [nr][cap]
	int *lcnt = nullptr;         // [nr][cap]
	int *nzp = nullptr, *n2p = nullptr;    // [nr][cap] non-zero codes / codes 2 per (range, variant) (resident blocks)
	int *n3 = nullptr;           // [cap] listed missing genotypes per variant
	uint8_t *ovf = nullptr;      // [cap] 1 = not listed (the pool was full): the scan takes the FP64 kernel for it
	// carrier lists of the rare variants (at most SPA5_NNZ carriers): what the per-variant SPA kernels walk
	int *nzv = nullptr, *n2v = nullptr;    // [cap] non-zero codes / codes 2 per variant (load-time scratch)
	unsigned *cptr = nullptr;    // [cap + 1] start of a variant's list in cidx
	unsigned *cidx = nullptr;    // sample | code << 30, ascending per variant
	size_t cidx_cap = 0;
	uint8_t *corient = nullptr;  // [cap] 0 no list, 1 list of the non-zero codes, 2 of the codes other than 2 (AF > 0.5)
	hipEvent_t ready = nullptr;  // recorded behind the last load: scans on other streams wait for it
	hipEvent_t last_read = nullptr;   // recorded behind the last scan that reads the block: a reload waits for it
	bool was_read = false;
	// census of the last load (s3_lists_finish_kernel): [0] listed missing genotypes / 64, [1] variants the pool had no room for
	int *info = nullptr, *h_info = nullptr;
	bool info_read = false, dense = false;
};

struct sgx_handle {
	int device = 0;
	hipStream_t stream = nullptr;
	DevModel md{};
	double *dF = nullptr, *dX = nullptr, *dy = nullptr, *dmu = nullptr, *dmu2 = nullptr, *dXM = nullptr;
	// per-call workspace
	SpaRec *recs = nullptr; size_t recs_cap = 0;
	int *fallback = nullptr;          // rec indices that need the exact dense pass
	int *fb_spa2 = nullptr;           // rec indices for the per-variant kernel (series on the variant's list)
	int *fb_x2 = nullptr;             // ... of those, the ones that need the exact sweeps
	int nseg = 0;                     // sample segments of the SPA stage
	bool force_dense = false;         // test hook: every SPA variant takes the exact dense pass
	// series SPA stage (kern_spa4.h)
	double *seg4 = nullptr;           // [vcap4][nseg][NC + 5] partial sums of one round of flagged variants
	int vcap4 = 0, nround4 = 0;
	bool spa5_attr_set[3] = {false, false, false};
	bool mom_attr_set[3] = {false, false, false};   // per input type: the moments kernels' dynamic LDS size has been raised
	uint8_t *scr5 = nullptr; int *cur5 = nullptr; int nwg5 = 0;   // spa5_kernel: per-workgroup lists, queue cursor
	int spa_abl = 0;                  // timing experiments (wrong results)
	bool force_exact = false;         // test hook: every SPA variant takes the exact exp/log kernels
	// exact-integer MFMA score path (kern_score_mfma.h)
	bool mf_ok = false;
	MfTab mf[MF_MAXG]{};              // one limb table per column group
	int mf_nbfv[MF_MAXG]{};
	MfEpi mfe{};
	uint8_t *dFl = nullptr;
	long long *dQ = nullptr;          // [N][P] the fixed-point score values as int64 (s3_t3_kernel)
	int *mf_acc = nullptr;
	// score3 (kern_score3.h): item slabs, partial sums over the missing samples, variants for the FP64 kernel
	int *s3_slabs = nullptr; size_t s3_slabs_cap = 0;
	long long *s3_t3 = nullptr; size_t s3_t3_cap = 0;
	int *s3_ovf = nullptr; size_t s3_ovf_cap = 0;
	bool s3_attr[17] = {false};       // per NBF: dynamic LDS size raised
	bool s3_attr_miss[17] = {false};  // ... of the three-plane form
	hipStream_t hstream = nullptr;    // the score chain of a block scan (list pass, sparse pass, contraction, reduction, epilogue): HIGH priority,
	                                  // so that it is not slowed by the SPA kernels of the other lane's step it runs beside (h->stream: low)
	hipStream_t s3_side = nullptr;    // the sparse pass over the missing genotypes (beside the list pass's tail; joined before the contraction kernel)
	hipEvent_t s3_fork = nullptr, s3_join = nullptr;
	sgx_block *tmp_blk[2] = {nullptr, nullptr};   // row-major calls: the rows are ingested into a block first
	int n_cu = 256;
	int *counters = nullptr;          // [0] n_spa, [1] n_valid, [2] n_fallback
	int *h_counters = nullptr;        // pinned
	double *scratch = nullptr; size_t scratch_stride = 0; int spa_grid = 0;
	// host-pointer staging
	uint8_t *stage_in = nullptr; size_t stage_in_cap = 0;
	// pipelined host-buffer scans (scan_host): two input buffers, results through pinned memory
	hipStream_t cstream = nullptr;    // copies of the block that is NOT being computed
	size_t pipe_bytes = 0;            // test hook: chunk size of the pipeline (0 = PIPE_BYTES)
	uint8_t *pipe_in[2] = {nullptr, nullptr}; size_t pipe_in_cap = 0;
	uint8_t *pipe_pk[2] = {nullptr, nullptr}; size_t pipe_pk_cap = 0;       // packed 2-bit rows made on the device
	double *pipe_out[2] = {nullptr, nullptr}; uint8_t *pipe_valid[2] = {nullptr, nullptr}; size_t pipe_out_cap = 0;
	double *pin_out[2] = {nullptr, nullptr}; uint8_t *pin_valid[2] = {nullptr, nullptr};   // pinned host
	int *pipe_flag = nullptr, *h_pipe_flag = nullptr;
	hipEvent_t ev_h2d = nullptr;
	hipEvent_t ev_copy[2] = {nullptr, nullptr}, ev_done[2] = {nullptr, nullptr};   // sgx_block_load: chunk copied / chunk read
	uint8_t *stage_pk = nullptr; size_t stage_pk_cap = 0;   // burden: packed rows, CSR and tables
	double *ds_part = nullptr; size_t ds_part_cap = 0;       // dosage score kernels: per-split partial sums
	double *stage_out = nullptr; uint8_t *stage_valid = nullptr; size_t stage_out_cap = 0;
	hipEvent_t ev[3] = {nullptr, nullptr, nullptr};
	hipEvent_t evk[2] = {nullptr, nullptr};    // around the contraction kernel alone (stats.ms_kernel)
	bool evk_set = false;
	hipEvent_t ev_lists = nullptr;             // in front of the list pass of a row-major call (stats.ms_lists = ev_lists .. ev[0])
	bool lists_timed = false;
	sgx_stats stats{};
	bool force_v1 = false;            // "score_v1" option: gather kernel instead of the MFMA path
	bool stats_pending = false;
	// dense g_pos / g_neg fallback of the last device-resident call, launched by the next sync if that call turned
	// out to need it (launch_spa, lazy_dense)
	struct { bool active = false; RowsRef rr{}; size_t M = 0; double *out8 = nullptr; const sgx_block *blk = nullptr; } pend_dense;
	// Lanes ("lanes" option, 1..SGX_MAX_LANES): device-resident scans go round-robin over this handle and
	// its twins, each with its own stream and workspace (the model arrays are shared), so that the SPA stage
	// of one block of variants runs while the score stage of the next one streams the genotypes, and -- where
	// the blocks are small (N = 50 000) -- the many short kernels of a step find others to run beside.
	// Score stages never overlap each other (the later one waits for the earlier one's event).
	sgx_handle *twins[3] = {nullptr, nullptr, nullptr};   // owned by the primary handle
	int n_lanes = 1;
	sgx_handle *owner = nullptr;      // set in a twin
	bool shares_model = false;        // twin: dF .. dFl belong to the owner
	// primary: the three-plane form of the contraction kernel (no lists of the missing genotypes, cost independent of
	// the missing rate) for calls that would build lists -- set when a finished step listed more than SGX_DENSE_ON of
	// its genotypes as missing (or overflowed the pool), cleared when a three-plane step counted fewer than SGX_DENSE_OFF
	bool dense_mode = false;
	int dense_opt = -1;               // "three_plane" option: -1 automatic, 0 never, 1 always
	// bound on the z-score's move by the fixed-point columns' quantisation beyond which a variant is scored by the FP64
	// kernel (score3_epilogue): 2e-11 keeps the p-value inside 1e-10 relative with room; "guard_exp" option: 10^-x
	double guard_tol = 2e-11;
	bool used_miss = false;           // this lane's call in flight took the three-plane form
	int next_lane = 0;                // primary: which lane takes the next _dev call
	sgx_handle *last_issued = nullptr;// primary: lane of the most recent call
	sgx_stats total{};                // primary: sums over harvested calls (sgx_get_stats_total)
	uint64_t total_calls = 0;
};

#define SGX_DENSE_ON  0.005       /* see rows_take_three_planes */
#define SGX_DENSE_OFF 0.003

static int set_dev(sgx_handle *h)
{
	HIPCHK(hipSetDevice(h->device));
	return SGX_OK;
}

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
	L.nsub = (int)std::max<size_t>(1, std::min<size_t>(S3_NSUB, (b->cap * (size_t)b->nr + 3) / 4));
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
	const unsigned grid = (unsigned)(((m + 3) / 4) * (size_t)b->nr);
	const S3Lists L = block_lists(b);
	if (b->lists_only)
		hipLaunchKernelGGL((s3_lists_kernel<8, false, false>), dim3(grid), dim3(256), 0, st, rows_dev, bpv, b->N, (int)m, (int)v_first, b->ntile, L,
			(uint8_t *)nullptr, (size_t)0);
	else
		hipLaunchKernelGGL((s3_lists_kernel<8, true, true>), dim3(grid), dim3(256), 0, st, rows_dev, bpv, b->N, (int)m, (int)v_first, b->ntile, L,
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
	int rc = ensure_buf(h, &h->s3_t3, &h->s3_t3_cap, (size_t)(b->nr + 1) * M * md.P * 2);      // per-range partials, then the totals
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
	if (t3_done) {
		const size_t n3e = M * (size_t)md.P * 2;
		hipLaunchKernelGGL(s3_t3_sum_kernel, dim3((unsigned)((n3e + 255) / 256)), dim3(256), 0, st, n3e, b->nr, h->s3_t3, h->s3_t3 + (size_t)b->nr * n3e);
		HIPCHK(hipGetLastError());
	} else {
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
		const size_t n3e = M * (size_t)md.P * 2;
		hipLaunchKernelGGL(s3_t3_sum_kernel, dim3((unsigned)((n3e + 255) / 256)), dim3(256), 0, s2, n3e, b->nr, h->s3_t3, h->s3_t3 + (size_t)b->nr * n3e);
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
	hipLaunchKernelGGL((score3_epilogue<KK>), dim3((unsigned)((M + 255) / 256)), dim3(256), 0, st, (int)M, md, ep, h->mf_acc, acc_stride, \
		miss ? 16 * NBF : 0, h->s3_t3 + (size_t)b->nr * M * md.P * 2, b->n3, b->ovf, h->s3_ovf, h->recs, h->counters, btop, h->fb_spa2, h->fb_x2, out8, valid, h->guard_tol); \
	hipLaunchKernelGGL((score2b_kernel<2 * KK + 2, 256>), dim3((unsigned)std::min<size_t>(M, 4 * (size_t)h->n_cu)), dim3(256), 0, st, \
		rr, (int)M, md, h->recs, h->counters, out8, valid, (const int *)h->s3_ovf, 23, btop, h->fb_spa2, h->fb_x2); \
	break;
	FOR_EACH_K(ECASE)
#undef ECASE
	default: return fail(SGX_EINVAL, "score3: unsupported K=%d", md.K);
	}
	HIPCHK(hipGetLastError());
	HIPCHK(hipEventRecord(h->ev[1], st));
	h->stats.score_launches = 6;
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

// Which form of the contraction kernel a call takes.  The two-plane form needs the positions of the missing genotypes
// (a pass over the rows, or a resident block's lists) and a sparse pass whose cost grows with their number; the
// three-plane form needs neither, at ~1.7 x the MFMAs.  Measured (tools/README.md, round 4): up to 3 B fragments
// (quantitative traits, K <= 2) the three-plane kernel costs what the two-plane kernel does and saves the list pass;
// from 4 fragments on it pays once more than ~0.5 % of the genotypes are missing -- where the sparse pass has grown
// to the difference and the pool of the lists (0.8 %) is about to overflow.
static bool rows_take_three_planes(const sgx_handle *lane)
{
	const sgx_handle *p = lane->owner ? lane->owner : lane;
	if (p->dense_opt >= 0) return p->dense_opt != 0;
	return lane->mf_nbfv[0] + 1 <= 3 || p->dense_mode;
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
	rc = ensure_buf(lane, &lane->s3_t3, &lane->s3_t3_cap, (size_t)(tb->nr + 1) * M * lane->md.P * 2);
	if (rc) return rc;
	HIPCHK(hipEventRecord(lane->ev_lists, lane->hstream));
	lane->lists_timed = true;
	{
		const S3Lists L = block_lists(tb);
		const int P = lane->md.P, PP = P <= 8 ? 8 : P <= 16 ? 16 : P <= 32 ? 32 : 64;
		const dim3 grid((unsigned)(((M + 3) / 4) * (size_t)tb->nr));
		hipStream_t st = lane->hstream;
		if (PP == 8) hipLaunchKernelGGL((s3_lists_t3_kernel<8, 8>), grid, dim3(256), 0, st, rows_dev, bpv, tb->N, (int)M, tb->ntile, L, P, lane->dQ, lane->s3_t3);
		else if (PP == 16) hipLaunchKernelGGL((s3_lists_t3_kernel<8, 16>), grid, dim3(256), 0, st, rows_dev, bpv, tb->N, (int)M, tb->ntile, L, P, lane->dQ, lane->s3_t3);
		else if (PP == 32) hipLaunchKernelGGL((s3_lists_t3_kernel<8, 32>), grid, dim3(256), 0, st, rows_dev, bpv, tb->N, (int)M, tb->ntile, L, P, lane->dQ, lane->s3_t3);
		else hipLaunchKernelGGL((s3_lists_t3_kernel<8, 64>), grid, dim3(256), 0, st, rows_dev, bpv, tb->N, (int)M, tb->ntile, L, P, lane->dQ, lane->s3_t3);
		HIPCHK(hipGetLastError());
	}
	rc = block_finish(tb, M, lane->hstream);
	if (rc) return rc;
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
