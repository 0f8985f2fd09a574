/*
 * saigehip.h -- C ABI of libsaigehip.so, the MI355X (gfx950) implementation of
 * the SAIGEgds single-variant association scan.
 *
 * Drop-in boundary.  In the reference the R driver seqAssocGLMM_SPA()
 * (R/assoc_single.r:92-334) reaches native code through three .Call entry
 * points, the last two ONCE PER VARIANT from SeqArray::seqApply
 * (R/assoc_single.r:207,218 via .cfunction, R/saige_main.r:22-30):
 *
 *   SEXP saige_score_test_init (SEXP model)    src/saige_main.cpp:103-150
 *   SEXP saige_score_test_bin  (SEXP dosage)   src/saige_main.cpp:437-462
 *   SEXP saige_score_test_quant(SEXP dosage)   src/saige_main.cpp:413-434
 *
 * A per-variant callback cannot feed a GPU, so this ABI keeps the meaning of
 * those three calls and changes the granularity to BLOCKS of variants:
 *
 *   sgx_init        <- saige_score_test_init : model arrays by pointer+length,
 *                      copied to HBM (the reference borrows R-owned memory)
 *   sgx_scan_2bit   <- saige_score_test_bin/quant over a block; genotypes are
 *                      2-bit codes, what seqApply(.useraw=NA) yields as RAW
 *                      0/1/2/0xFF, packed 4 samples per byte
 *   sgx_scan_u8     <- the RAWSXP branch of get_ds  (saige_main.cpp:179-182)
 *   sgx_scan_f64    <- the REALSXP branch of get_ds (saige_main.cpp:173-174)
 *
 * Results: one row of 8 doubles per variant,
 *   [AF.alt, mac, num, beta, SE, pval, p.norm, converged]
 * exactly the NumericVector(8) of saige_score_test_bin (saige_main.cpp:453-458);
 * for quantitative traits the reference returns 6 values (:427-430) and columns
 * 6,7 are NaN here.  A variant rejected by the MAF/MAC/missing filter
 * (saige_main.cpp:288-292, :197-201; reference returns R_NilValue) gets
 * valid[j]=0 and a NaN row.
 *
 * No R, torch or HIP types appear in the signatures.  All functions return 0 on
 * success or a negative SGX_E* code; sgx_last_error() gives the message (the R
 * glue turns it into stop(), as BEGIN_RCPP/END_RCPP does in the reference).
 */
#ifndef SAIGEHIP_H
#define SAIGEHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SGX_OK          0
#define SGX_EINVAL     -1   /* bad argument (length, alignment, NULL)          */
#define SGX_EHIP       -2   /* HIP runtime error                               */
#define SGX_ENOMEM     -3   /* allocation failed                               */
#define SGX_ENODEV     -4   /* no usable gfx950 device                         */

#define SGX_MAX_COEFF  16   /* K = nrow(XV) supported by the compiled kernels  */
#define SGX_TRAIT_BINARY 0
#define SGX_TRAIT_QUANT  1

/* 2-bit genotype code (alt-allele dosage, SeqArray "$dosage_alt"); sample i of a
 * variant row lives in bits 2*(i%4)..2*(i%4)+1 of byte i/4. */
#define SGX_GENO_MISSING 3

typedef struct sgx_handle sgx_handle;

/* The list .init_nullmod builds (R/assoc_single.r:28-66) as read by
 * saige_score_test_init (src/saige_main.cpp:106-130).  K x N matrices are
 * column-major, i.e. the K values of sample i are contiguous at [K*i].
 *
 * t_XXVX_inv and XV are part of the reference's list and are accepted here so
 * that the caller passes the list as it is, but the scan does not read them:
 * both reference branches (saige_main.cpp:237-262 sparse, :263-292 dense) are
 * computed from t_X, t_XVX_inv_XV, XVX and S_a.  They may be
 * NULL.  With SAIGEHIP_CHECK_MODEL=1 in the environment sgx_init holds them
 * against t_X / t_XVX_inv_XV (XV = V t_X, t_XVX_inv_XV = V t_XXVX_inv with one
 * weight V_i per sample, 1e-8) and returns SGX_EINVAL naming the first entry
 * that disagrees -- for callers who assemble the arrays themselves. */
typedef struct sgx_model {
	int32_t n_samp;              /* N  = length(y)                  :117 */
	int32_t n_coeff;             /* K  = nrow(XV)                   :118 */
	int32_t trait;               /* SGX_TRAIT_*                          */
	int32_t reserved;
	double tau[2];               /* variance components             :119 */
	double var_ratio;            /* var.ratio                       :130 */
	double maf;                  /* thresholds; NaN -> -1/-1/1/0.05 :108-115 */
	double mac;
	double missing;
	double spa_pval;
	const double *y;             /* N                               :120 */
	const double *mu;            /* N                               :121 */
	const double *y_mu;          /* N   y - mu                      :122 */
	const double *mu2;           /* N   mu*(1-mu)                   :123 */
	const double *t_XXVX_inv;    /* K x N   NOT READ by the scan    :124 */
	const double *XV;            /* K x N   NOT READ by the scan    :125 */
	const double *t_XVX_inv_XV;  /* K x N                           :126 */
	const double *XVX;           /* K x K                           :127 */
	const double *t_X;           /* K x N                           :128 */
	const double *S_a;           /* K                               :129 */
} sgx_model;

/* Counters and device timings of the most recent scan call on a handle. */
typedef struct sgx_stats {
	uint64_t n_variants;   /* variants in the call                            */
	uint64_t n_valid;      /* passed the filter                               */
	uint64_t n_spa;        /* pval_noadj <= spa.pval, handed to the SPA stage */
	uint64_t n_spa_dense;  /* of those, needed the exact dense g_pos/g_neg pass */
	uint64_t n_spa_slow;   /* of those, through the exact exp/log kernel (not the series) */
	float ms_score;        /* HIP-event time of the score kernel(s), ms       */
	float ms_spa;          /* HIP-event time of the SPA kernel(s), ms         */
	float ms_total;        /* first launch .. last launch complete, ms        */
	uint32_t score_launches;
	uint32_t spa_launches;
	float ms_kernel;       /* HIP-event time of the genotype-streaming kernel alone (score3_kernel), ms;
	                          0 where the scan took the FP64 kernels                                      */
	float ms_lists;        /* row-major calls on the two-plane form: HIP-event time of the pass that finds the
	                          missing genotypes (in front of the score stage; part of ms_total, not of
	                          ms_score); 0 for the three-plane form and for sgx_scan_block, whose lists were
	                          made when the block was loaded                                             */
	uint32_t three_plane;  /* 1: the call took the three-plane form of the contraction kernel (the sums over the
	                          missing samples from a third MFMA plane: no lists, no sparse pass; chosen for
	                          few score columns or many missing genotypes); totals: number of such calls     */
	uint32_t n_unlisted;   /* variants whose missing genotypes found the pool of the lists full (FP64 kernel)  */
	uint32_t n_guarded;    /* variants whose a-posteriori bound on the fixed-point columns' quantisation (its
	                          effect on the z-score) exceeded the guard: scored by the FP64 kernel  */
} sgx_stats;

/* Library / device ------------------------------------------------------- */
const char *sgx_version(void);
const char *sgx_last_error(void);
int sgx_device_count(void);
/* verifies the v_mfma_i32_16x16x64_i8 lane maps the score kernel relies on and
 * the accuracy of the device exp/log used by the SPA stage */
int sgx_selftest(int device);

/* Model lifetime (one handle per GPU; handles are independent, no globals). */
int  sgx_init(const sgx_model *model, int device, sgx_handle **out);
void sgx_free(sgx_handle *h);
int  sgx_set_thresholds(sgx_handle *h, double maf, double mac, double missing,
	double spa_pval);

/* Fixed-point layout sgx_init chose for the exact-integer score stage: limbs[c] = bytes per entry of
 * score column c in the order c' (K columns), e (K), s, w; 0 for a column that is derived instead
 * of carried.  Columns of heavy-tailed covariates get more limbs than ordinary ones; n_groups = 0
 * means the model's dynamic range exceeds what the fixed-point form holds and the scan uses the
 * FP64 kernels. */
int sgx_score_layout(sgx_handle *h, int32_t *limbs, int32_t n_limbs, int32_t *n_groups);

/* Scan a block of variants held in HOST memory.  packed: n_variants rows of
 * bytes_per_variant bytes (>= ceil(N/4)).  out8: n_variants*8 doubles.
 * valid: n_variants bytes.  Synchronous (chunks cross PCIe while the previous chunk is scanned). */
int sgx_scan_2bit(sgx_handle *h, const uint8_t *packed, size_t bytes_per_variant,
	size_t n_variants, double *out8, uint8_t *valid);

/* Page-locked host memory for block buffers handed to the host-buffer scans: from it the chunks cross
 * PCIe at the full link rate and asynchronously (pageable memory works too, staged by the runtime). */
void *sgx_host_alloc(size_t bytes);
void  sgx_host_free(void *p);

/* Same, all buffers already resident in this GPU's HBM (device pointers): the headline path.
 * packed must be 16-byte aligned and bytes_per_variant a multiple of 64 with
 * bytes_per_variant >= sgx_row_stride(N) = 128*ceil(N/512) (rows that start on 128-byte lines are read
 * as whole lines; a stride that is a multiple of 64 only still works, slower).  The rows are read WHERE
 * THEY ARE, twice: one pass lists the positions of the missing genotypes (the sparse form of
 * f64_af_ac_impute's walk, src/vectorization.cpp:186-205), then the contraction kernel streams them;
 * nothing is copied or rearranged.  Asynchronous on one of the handle's streams: the rows, out8_dev and
 * valid_dev must stay untouched until sgx_sync(), which is also what makes results and stats readable. */
int sgx_scan_2bit_dev(sgx_handle *h, const uint8_t *packed_dev,
	size_t bytes_per_variant, size_t n_variants, double *out8_dev,
	uint8_t *valid_dev);

/* Genotype blocks: rows resident on the device for any number of scans ------------------------------
 * The reference streams one variant at a time out of the GDS file into its C++ code
 * (R/assoc_single.r:202-209, seqApply) and finds, per variant and per phenotype, the missing genotypes
 * (f64_af_ac_impute) and the carriers (f64_nonzero_index, src/vectorization.cpp:209-215).  A BLOCK keeps
 * up to max_variants rows in HBM together with what does not depend on the model: the rows themselves
 * (row-major, as they came), the positions of their missing genotypes, and the carrier lists (sample and
 * code, ascending) of every variant with at most 8 192 carriers in the orientation the scan will use
 * (non-zero codes, or the codes other than 2 where the alt allele is the major one).  One loaded block
 * can be scanned with any number of models (phenotypes): each scan then streams the rows once.
 *   sgx_block_create    device storage for up to max_variants rows of n_samp samples
 *   sgx_block_bytes     what that takes (about 1.2 x the packed rows at large N)
 *   sgx_block_load_dev  rows already in this GPU's memory (bytes_per_variant a multiple of 16,
 *                       >= sgx_row_stride(n_samp), 16-byte aligned); asynchronous on the handle's stream;
 *                       a load waits for the scans that still read the block
 *   sgx_block_load      rows in host memory (>= ceil(n_samp / 4) bytes each), through the pinned pipeline
 *   sgx_scan_block      the scan; asynchronous like sgx_scan_2bit_dev (same lanes, stats, sgx_sync)
 *   sgx_block_create_ex test hook: carrier lists with room for clist_avg entries per variant on average
 * A block must not be freed, and its result buffers not read, before sgx_sync() on every handle that
 * scanned it.  Variants whose missing genotypes find the block's pool full (a list entry per missing
 * genotype, room for 0.8 % of the block at large N) are scanned by the FP64 kernels instead: same
 * results, slower.  Rare variants beyond a full carrier list (1 536 entries per variant of the block
 * on average) have their rows scanned by the SPA kernels, as sgx_scan_2bit_dev's rows are: same results. */
typedef struct sgx_block sgx_block;
size_t sgx_block_bytes(int32_t n_samp, size_t max_variants);
int  sgx_block_create(int32_t n_samp, size_t max_variants, int device, sgx_block **out);
int  sgx_block_create_ex(int32_t n_samp, size_t max_variants, int device, long long clist_avg, sgx_block **out);
void sgx_block_free(sgx_block *b);
int  sgx_block_load_dev(sgx_handle *h, sgx_block *b, const uint8_t *packed_dev,
	size_t bytes_per_variant, size_t n_variants);
int  sgx_block_load(sgx_handle *h, sgx_block *b, const uint8_t *packed,
	size_t bytes_per_variant, size_t n_variants);
size_t sgx_block_variants(const sgx_block *b);
int  sgx_scan_block(sgx_handle *h, const sgx_block *b, double *out8_dev, uint8_t *valid_dev);

/* Dosage inputs in HOST memory, one row of N values per variant, the three branches of get_ds
 * (src/saige_main.cpp:171-183):
 *   u8 : 0..254, 0xFF = missing (RAWSXP :179-182);  i32: NA_INTEGER (INT_MIN) = missing (INTSXP
 *   :175-178);  f64: NaN/Inf = missing (REALSXP :173-174).
 * u8 / i32 blocks that hold hard calls only (0, 1, 2, missing) are packed to 2-bit rows on the device
 * and take the same kernels as sgx_scan_2bit; anything else takes the dosage kernels.
 * All host-buffer scans are pipelined: the next chunk of the block crosses PCIe while the current
 * one is computed ("pipe_mb" option = chunk size). */
int sgx_scan_u8(sgx_handle *h, const uint8_t *dosage, size_t n_variants,
	double *out8, uint8_t *valid);
int sgx_scan_i32(sgx_handle *h, const int32_t *dosage, size_t n_variants,
	double *out8, uint8_t *valid);
int sgx_scan_f64(sgx_handle *h, const double *dosage, size_t n_variants,
	double *out8, uint8_t *valid);

/* Aggregate tests: n_rows burden rows from 2-bit genotypes in HOST memory, then the
 * single-variant test on every row (replaces ds_mat_burden + single_test_bin/quant inside
 * saige_burden_test_*, saige_acatv_test_bin and saige_acato_test_bin, src/saige_main.cpp:526-976).
 * Row r = sum over its entries e in [row_ptr[r], row_ptr[r+1]) of lut[4e + code], code = the 2-bit
 * genotype of variant var_idx[e] (row of `packed`); the caller folds weight, mean imputation and
 * the flip to the minor allele into lut (see saigegds_amd/aggregate.py).  out8 / valid as above,
 * one row per burden row.  Synchronous. */
int sgx_burden_2bit(sgx_handle *h, const uint8_t *packed, size_t bytes_per_variant,
	size_t n_variants, size_t n_rows, const int64_t *row_ptr, const int32_t *var_idx,
	const double *lut, double *out8, uint8_t *valid);

/* Host-side decoder of SeqArray's genotype/data node (dBit2 [variant][sample][ploidy]) into the 2-bit dosage
 * rows of sgx_scan_2bit / sgx_block_load: code = number of non-reference alleles, 3 = missing -- SeqArray's
 * "$dosage_alt", what seqApply(.useraw=NA) hands saige_score_test_bin as RAW (R/assoc_single.r:202-221).
 * alleles: the node's bytes from the one that holds bit `bit0` (a multiple of 4) of the first wanted
 * variant; sel: sample indices to keep, in the order wanted (NULL = all n_samp); out: m rows of out_stride
 * bytes (a pinned block buffer, say).  threads: host threads to split the rows over (0 = automatic).
 * Needs no GPU. */
int sgx_decode_dbit2(const uint8_t *alleles, size_t bit0, int32_t n_samp, size_t m,
	const int64_t *sel, int32_t n_sel, uint8_t *out, size_t out_stride, int threads);

/* Per-variant counts of a HOST 2-bit matrix on GPU `device`: n_valid[j] = samples with a call,
 * allele_sum[j] = their alt-allele count -- the inputs of the maf / missing-rate variant filter of
 * seqFitNullGLMM_SPA (seqSetFilterCond, R/saige_main.r:314-321).  Needs no model handle. */
int sgx_geno_stats_2bit(const uint8_t *packed, size_t bytes_per_variant, int32_t n_samp,
	size_t n_variants, int device, int32_t *n_valid, int32_t *allele_sum);

/* Tuning / test hooks (per handle; there are no process-wide switches): "spa_exact" (every flagged variant
 * through the exact exp/log SPA kernel instead of the cumulant series), "force_dense" (exact g_pos/g_neg
 * pass for every SPA variant), "score_v1" (FP64 gather score kernel instead of the MFMA path), "lanes"
 * (1..4: successive sgx_scan_block / sgx_scan_2bit_dev calls go round-robin over that many streams with
 * their own workspace, so the SPA stage of one block runs under the score stage of the next; call
 * sgx_sync() before reading any output), "pipe_mb" (MiB of input rows per chunk of a host-buffer scan;
 * 0 = default 512), "spa_abl" (diagnostic bits; 512: the SPA kernels scan the rows of a block instead of
 * walking its carrier lists), "three_plane" (-1 automatic -- row-major calls take the three-plane form of the
 * contraction kernel for models of up to four B fragments (K <= 3 binary, any quantitative K <= 2) and otherwise once
 * more than ~0.5 % of the genotypes are missing, resident blocks by the census of their load; 0 / 1: never / always), "guard_exp" (x: the fixed-point guard at 10^-x instead of 2e-11; 0 = off, 300 = every
 * variant through the FP64 kernel).  Results never depend on them beyond rounding (1e-12). */
int sgx_set_option(sgx_handle *h, const char *name, long long value);

int sgx_sync(sgx_handle *h);
int sgx_get_stats(sgx_handle *h, sgx_stats *st);          /* the most recent call */
/* sums over all calls completed since the last reset (ms_* are per-stage event times: with two
 * lanes they overlap in wall time) */
int sgx_get_stats_total(sgx_handle *h, sgx_stats *st, uint64_t *n_calls, int reset);

/* Device-side helpers for the benchmark / synthetic GDS generator ---------- */

/* Smallest legal bytes_per_variant for sgx_scan_2bit_dev. */
size_t sgx_row_stride(int32_t n_samp);

/* Fill packed_dev (n_variants rows of bytes_per_variant) with synthetic 2-bit
 * genotypes: sample i of variant (first_variant+j) draws one 64-bit
 * counter-based random word (splitmix64 of seed, variant, sample); the top 32
 * bits u pick the code: u < thr[3j] -> 0, u < thr[3j+1] -> 1, else 2; the low
 * 32 bits v mark it missing when v < thr[3j+2].  thr_dev: n_variants*3 uint32
 * in device memory.  The same function in numpy: saigegds_amd/synth.py. */
int sgx_synth_2bit_dev(sgx_handle *h, uint8_t *packed_dev, size_t bytes_per_variant,
	int32_t n_samp, size_t n_variants, uint64_t first_variant, uint64_t seed,
	const uint32_t *thr_dev);

/* Null-model fit: implicit GRM on 2-bit packed genotypes ------------------------
 * The operator inside seqFitNullGLMM_SPA()'s AI-REML/PCG loop:
 *   sgx_grm_init       <- saige_store_2b_geno    src/saige_fitnull.cpp:159-230
 *   sgx_grm_diag       <- buf_diag_grm           :205-227
 *   sgx_grm_crossprod  <- get_crossprod_b_grm    :435-536   out = G'(G b)/M
 *   sgx_grm_pcg        <- PCG_diag_sigma         :581-614   (tau0 diag(1/w) + tau1 GRM) x = b
 * packed: n_markers rows of bytes_per_marker bytes, 4 samples per byte (as above);
 * vectors are host arrays of n_samp doubles. */
typedef struct sgx_grm sgx_grm;
int  sgx_grm_init(const uint8_t *packed, size_t bytes_per_marker, int32_t n_samp,
	size_t n_markers, int device, sgx_grm **out);
/* packed_dev in this GPU's HBM (copied); b_dev/out_dev device vectors */
int  sgx_grm_init_dev(const uint8_t *packed_dev, size_t bytes_per_marker, int32_t n_samp,
	size_t n_markers, int device, sgx_grm **out);
int  sgx_grm_crossprod_dev(sgx_grm *g, const double *b_dev, double *out_dev);
int  sgx_grm_sync(sgx_grm *g);
void sgx_grm_free(sgx_grm *g);
int  sgx_grm_diag(sgx_grm *g, double *diag_out);
int  sgx_grm_crossprod(sgx_grm *g, const double *b, double *out);
int  sgx_grm_pcg(sgx_grm *g, const double *w, const double *tau, const double *b,
	int maxiter, double tol, double *x_out, int *iters_out);

#ifdef __cplusplus
}
#endif
#endif /* SAIGEHIP_H */
