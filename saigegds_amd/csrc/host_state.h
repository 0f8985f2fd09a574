// host_state.h -- the library's host-side state: genotype blocks and handles (lanes, workspaces, streams).
// Part of libsaigehip.so: included by saigehip.hip (one translation unit), not a header of its own.
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
