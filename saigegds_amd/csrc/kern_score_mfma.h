// kern_score_mfma.h -- score stage for 2-bit genotypes as an EXACT integer
// contraction on the matrix cores (v_mfma_i32_16x16x64_i8).
// Part of libsaigehip.so (single translation unit: saigehip.hip).
#pragma once
#include <type_traits>

// Why a matrix formulation at all: with FP64 the dense score sums cost
// 2(2K+2) flop per (variant, sample) against 0.25 B of input, i.e. they are
// FP64-bound at ~11 M variants/s, and skipping zeros turns them
// into a 64 B gather per carrier.  The sums are, however, integer-weighted:
//     sum_i code_i * F[i,c],   code_i in {0,1,2,3}.
// F is converted ONCE (sgx_init) to 56-bit fixed point per column,
//     F[i,c] ~ q[i,c] * 2^-e_c,   q = sum_{l<7} d_l 256^l,  d_l in [-128,127],
// so that every sum is an exact int32 dot product per limb
//     S[c,l] = sum_i code_i * d_l[i,c]              (|S| <= 384 N < 2^31)
// which v_mfma_i32_16x16x64_i8 evaluates at ~16x the FP64 FMA rate, with no
// rounding anywhere: the reduction order (and the split of N across workgroups,
// merged by integer atomics) cannot change a single bit.  The quantisation
// error is 2^-55 of the column maximum per entry, below the rounding error of
// a double-precision dot product.
//
// Planes.  A = raw 2-bit codes (0,1,2,3) against all limb columns gives
//     V[c] = T1 + 2 T2 + 3 T3      (T_g = sum of q over samples with code g)
// A' = bit 1 of the code against the mu2 limbs gives  B1 = T2 + T3  (mu2 only),
// and A'' = [code == 3] against the value columns gives T3[c] and n3; that plane
// is only multiplied in fragments in which a wave sees a missing code.  From
// these, in integer arithmetic,
//     W[c] = V[c] - 3 T3[c] = T1 + 2 T2,   H2 = B1 - T3[mu2] = T2[mu2],
//     AC = V[ones] - 3 n3,   Num = N - n3,
// and with imp = 2 AF the sums the epilogue needs (dev_common.h):
//     no flip:  sum G F = W + imp T3          sum G^2 mu2 = W + 2 H2 + imp^2 T3
//     flip:     sum G F = 2 Ftot - W - imp T3
//               sum G^2 mu2 = 4 (Ftot - S1 - H2 - T3) + S1 + (2-imp)^2 T3,  S1 = W - 2 H2.

typedef int v4i __attribute__((ext_vector_type(4)));

#define MF_NAF 4             /* A fragments (16 variants each) per wave      */
#ifndef MF_WAVES
#define MF_WAVES 4           /* waves per workgroup -> 256 variants          */
#endif
#define MF_VPW (16 * MF_NAF) /* variants per wave                            */
#define MF_VPB (MF_VPW * MF_WAVES)
#define MF_NLIMB 7           /* limbs of a full-precision column (56-bit fixed point)              */
#define MF_LIMB_A 5          /* limbs of the t_XVX_inv_XV columns (c'): see "Limb counts" below     */
#define MF_LIMB_E 6          /* limbs of the w X columns (e)                                        */
#define MF_GLIMBS 64         /* limb columns per column group = 4 B fragments                       */
#define MF_MAXP (2 * SGX_MAX_COEFF + 2)   /* value columns: c' (K), e (K), s, w                     */
#define MF_MAXG 4

// what the contraction kernel needs of one column group
struct MfTab {
	const uint8_t *Fl;         // [ngrp_pad][NCOL][16] int8 limb digits, sample order mf_pos()
	int ntile;                 // number of 256-sample tiles = ngrp_pad / 16
};

// Limb counts.  s = G.(y-mu) and w = sum w_i G_i^2 carry 7 limbs (56 bits: below the rounding of a
// double-precision dot product).  The covariate projections enter the statistics only through
//     var2 = c'.XVX.c' + w - 2 e.c'      S = s - S_a.c'
// where c' = (X'VX)^-1 X'V G is O(AF) for the intercept direction and O(1/sqrt(N)) otherwise,
// XVX c' - e vanishes when the two weight vectors (no-K V, GLMM mu2) agree, and S_a = X'(y-mu) is
// ~0 at the fit.  e therefore carries 6 limbs (48 bits) and the t_XVX_inv_XV columns 5 (40 bits):
// measured effect on var2 <= 5e-14 relative and on S/sqrt(var2) <= 6e-16 (golden model and the
// N = 100 000 synthetic model, tools/limb_sim.py) -- under the rounding noise of the reference's
// own double sums, and 2000 x under the 1e-10 parity bar.  With K = 3 the value
// columns then fill exactly 3 B fragments: 3*5 + 3*6 + 7 + 7 + 1 = 48.
//
// Column groups.  A workgroup's accumulators hold up to 4 B fragments (64 limb columns); with
// more covariates the columns are cut into groups (never inside a column) and the contraction
// kernel runs once per group over the same packed rows.  Group 0 carries s, w, the constant-1
// column and the bit-1 fragment.  Accumulator row of a variant = the groups' rows one after the
// other: goff[g] ints in, group g has gncol[g] ints of value (+ bit-1) sums followed by
// 16 * nbfv[g] ints of missing-plane sums.
struct MfEpi {
	int ngroups, acc_stride;
	int goff[MF_MAXG], gncol[MF_MAXG];
	int col_ones;              // group 0: column of the constant 1
	int col_b1;                // group 0: first column of the w limbs in the bit-1 fragment
	unsigned char cgrp[MF_MAXP], ccol[MF_MAXP], climb[MF_MAXP];   // per value column: group, first limb column, limbs
	int derive_c;              // quantitative traits: c' = XVXi e instead of carried columns (climb = 0)
	double XVXi[SGX_MAX_COEFF * SGX_MAX_COEFF];
	int escale[MF_MAXP];       // F = q * 2^-escale
	long long ftot_hi[MF_MAXP];// sum_i q[i,c] = hi * 2^32 + lo
	long long ftot_lo[MF_MAXP];
};

// Sample order inside a group of 16.  One dword of the packed row holds 16 codes,
// code s in bits 2s..2s+1.  The A fragment takes them as 4 dwords of 4 bytes; with
//     val[t] = (w >> 2t) & 0x03030303      (byte j of val[t] = code of sample 4j + t)
// the unpack is two VALU operations per dword, and the B tiles simply store the 16
// samples of a group in that same order: sample s at byte mf_pos(s) = 4 (s & 3) + (s >> 2).
__host__ __device__ __forceinline__ int mf_pos(int s) { return ((s & 3) << 2) | (s >> 2); }

// value plane (codes 0..3) and twice their bit 1 (0/2)
__device__ __forceinline__ void mf_unpack(uint32_t w, v4i &val, v4i &b1)
{
	val[0] = (int)(w & 0x03030303u);
	val[1] = (int)((w >> 2) & 0x03030303u);
	val[2] = (int)((w >> 4) & 0x03030303u);
	val[3] = (int)((w >> 6) & 0x03030303u);
#pragma unroll
	for (int k = 0; k < 4; k++) b1[k] = (int)((uint32_t)val[k] & 0x02020202u);   // the plane is 2*[code>=2]
}

// grid = (variant tiles of MF_VPB, sample splits); block = 64 * MF_WAVES.
// Accumulator row of a variant: [0, ncol) value plane + bit-1 fragment,
// [ncol, ncol + 16 nbfv) the same value columns summed over MISSING samples only
// (plane [code == 3]); that plane is multiplied only in fragments that contain
// a missing code, which a wave decides with one ballot.
// HAS_B1 = false drops the bit-1 plane and its fragment (the implicit-GRM products
// of kern_grm.h only need the code plane and the missing plane).
// WIDE: a lane fetches 2 x 16 B of each of its rows per PAIR of tiles (the two halves of one
// 128-B line, by back-to-back instructions) instead of 16 B per tile: half as many lines in flight
// per byte.  Needs bpv % 128 == 0 and even tile ranges (t0, t1 - t0, tb.ntile).
// ABL: switches of the timing tool tools/mfma_ablate.hip (wrong results; the product uses 0):
//   1 no missing plane, 2 no bit-1 MFMA, 4 no unpack, 16 no A loads, 32 no B DMA,
//   64 no arithmetic at all (loads, DMA, LDS reads and barriers only),
//   512 row loads in the pattern of a tiled layout (narrow rows only; the data is then not the variants'),
//   1024 s_memtime stamps per tile, 2048 all loads and DMA of a tile issued in one burst at its start
// NAF: A fragments per wave.  4 (2 waves per SIMD at <= 256 registers) or 3 (3 waves per SIMD at <= 168).
template <int NBFV, bool HAS_B1, bool WIDE = false, int ABL = 0, int NAF = MF_NAF>
__global__ void __launch_bounds__(WAVE * MF_WAVES, NAF == 4 ? 2 : NAF == 3 ? 3 : 4)   /* waves per SIMD */
score_mfma_kernel(const uint8_t *__restrict__ packed, size_t bpv, int M, MfTab tb,
	int tiles_per_split, int *__restrict__ accbuf, int acc_stride)
{
	constexpr int NBF = NBFV + (HAS_B1 ? 1 : 0);
	constexpr int NCOL = 16 * NBF;
	constexpr int TILE_BYTES = 16 * NCOL * 16;
	constexpr int NDMA = (TILE_BYTES / 1024 + MF_WAVES - 1) / MF_WAVES;   // DMA instructions per wave and tile
	constexpr int AW = WIDE ? 2 : 1;                                      // 16-B pieces of a row held per lane
	constexpr bool SPREAD = !(ABL & (2048 | 16));   // one VMEM instruction per MFMA group (else: a burst at the tile start)
	static_assert(AW * NAF + NDMA <= 4 * NAF, "more loads per tile than MFMA groups");
	extern __shared__ __attribute__((aligned(16))) uint8_t smem[];   // 2 x TILE_BYTES, nothing else
	uint8_t *ldsB = smem;

	const int tid = threadIdx.x, lane = tid & (WAVE - 1), wid = tid / WAVE;
	const int r = lane & 15, kg = lane >> 4;
	// Workgroups are handed to the 8 XCDs round-robin in dispatch order, so those with the same
	// (linear id % 8) share an L2.  Give each such group whole sample splits: its resident
	// workgroups then stream the same B tiles, which stay in that L2, while every packed row
	// is still read once.  (A placement guess only: correctness does not depend on it.)
	unsigned vt_idx = blockIdx.x, split = blockIdx.y;
	{
		const unsigned VT = gridDim.x, L = blockIdx.y * VT + blockIdx.x, full = gridDim.y & ~7u;
		if (L < full * VT) { const unsigned q = L >> 3; split = 8 * (q / VT) + (L & 7); vt_idx = q % VT; }
	}
	const int vbase = vt_idx * (16 * NAF * MF_WAVES) + wid * 16 * NAF;
	const int t0 = split * tiles_per_split;
	const int t1 = min(tb.ntile, t0 + tiles_per_split);

	v4i acc[NAF][NBF], accm[NAF][NBFV];
#pragma unroll
	for (int f = 0; f < NAF; f++) {
#pragma unroll
		for (int b = 0; b < NBF; b++) acc[f][b] = (v4i){0, 0, 0, 0};
#pragma unroll
		for (int b = 0; b < NBFV; b++) accm[f][b] = (v4i){0, 0, 0, 0};
	}
	bool saw_missing = false;   // wave-uniform

	// this lane's 16 B of each of its rows in tile t: dwords 16t+4kg .. +3.  Row pointers advance by
	// 64 B per tile; bpv is a multiple of 64 that covers every tile (sgx_row_stride), rows past M
	// are clamped (their sums are never stored)
	const uint8_t *rowp[NAF];
#pragma unroll
	for (int f = 0; f < NAF; f++)
		rowp[f] = packed + (size_t)min(vbase + 16 * f + r, M - 1) * bpv + (size_t)t0 * 64 + 16 * kg;
	int tcur_abl = t0;
	auto load_A1 = [&](uint4 &dst, int f, int piece) {
		if (ABL & 16) { dst = make_uint4(t0 + f, lane, wid, 0x01010101u); return; }
		if (ABL & 512) {   // timing only: the access pattern of a tiled layout (one contiguous KiB per fragment and tile)
			dst = *reinterpret_cast<const uint4 *>(packed + ((size_t)tcur_abl * ((M + 15) / 16) + (vbase / 16 + f)) * 1024 + lane * 16);
			if (f == NAF - 1) tcur_abl++;
			return;
		}
		// (measured with the streaming hint, round 4: 4.46 -> 5.04 ms per product -- a lane's 16 bytes are a quarter of a
		// 64-byte piece that four instructions share; the hint drops the line between them)
		dst = *reinterpret_cast<const uint4 *>(rowp[f] + 64 * piece);
		if (piece == AW - 1) rowp[f] += 64 * AW;
	};
	// one KiB of B tile t -> LDS buffer (t & 1) by LDS-DMA, lane-linear
	auto dma_B1 = [&](int t, int i) {
		const int k = wid + i * MF_WAVES;
		if ((ABL & 32) || k >= TILE_BYTES / 1024) return;
		const uint8_t *src = tb.Fl + (size_t)t * TILE_BYTES;
		uint8_t *dst = ldsB + (size_t)(t & 1) * TILE_BYTES;
		__builtin_amdgcn_global_load_lds(
			(const __attribute__((address_space(1))) void *)(src + (size_t)k * 1024 + lane * 16),
			(__attribute__((address_space(3))) void *)(dst + k * 1024), 16, 0, 0);
	};

	uint4 acur[NAF][AW], anxt[NAF][AW];
	if (t0 < t1) {
#pragma unroll
		for (int f = 0; f < NAF; f++)
#pragma unroll
			for (int p = 0; p < AW; p++) load_A1(acur[f][p], f, p);
#pragma unroll
		for (int i = 0; i < NDMA; i++) dma_B1(t0, i);
	}
	// ABL & 1024: s_memtime stamps -> cycles spent waiting at the tile barrier / issuing loads / computing
	unsigned long long st_wait = 0, st_issue = 0, st_comp = 0, st_prev = 0;
#define MF_STAMP(x) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(x) :: "memory")
	if (ABL & 1024) MF_STAMP(st_prev);

	// one 256-sample tile; HH = which of the AW row pieces it reads
	auto tile = [&](int t, auto HH) {
		constexpr int hh = decltype(HH)::value;
		__syncthreads();   // tile t landed (each wave drained its own DMA), tile t-1 fully consumed
		unsigned long long sa = 0, sb = 0;
		if (ABL & 1024) { __builtin_amdgcn_sched_barrier(0); MF_STAMP(sa); __builtin_amdgcn_sched_barrier(0); st_wait += sa - st_prev; }
		const bool more_B = t + 1 < t1;                  // there is a next tile
		const bool more_A = t + (AW - hh) < t1;          // there is a next row piece set (fetched in the hh = 0 tile)
		// loads of this tile, spread over its MFMA groups: the A pieces of the next tile (pair), then the next B tile
		auto vmem_slot = [&](int idx) {
			if (hh == 0 && idx < AW * NAF) {
				if (more_A) load_A1(anxt[idx / AW][idx % AW], idx / AW, idx % AW);
			} else {
				const int i = idx - (hh == 0 ? AW * NAF : 0);
				if (i < NDMA && more_B) dma_B1(t + 1, i);
			}
			__builtin_amdgcn_sched_barrier(0);
		};
		if (!SPREAD) {
#pragma unroll
			for (int idx = 0; idx < 4 * NAF; idx++) vmem_slot(idx);
		}
		if (ABL & 1024) { __builtin_amdgcn_sched_barrier(0); MF_STAMP(sb); __builtin_amdgcn_sched_barrier(0); st_issue += sb - sa; }
		const uint8_t *bt = ldsB + (size_t)(t & 1) * TILE_BYTES;
#pragma unroll
		for (int u = 0; u < 4; u++) {
			const int g = 4 * kg + u;
			v4i bfrag[NBF];
#pragma unroll
			for (int b = 0; b < NBF; b++)
				bfrag[b] = *reinterpret_cast<const v4i *>(bt + ((size_t)(g * NCOL + b * 16 + r)) * 16);
#pragma unroll
			for (int f = 0; f < NAF; f++) {
				const uint4 aw = acur[f][hh];
				const uint32_t w = (u == 0) ? aw.x : (u == 1) ? aw.y : (u == 2) ? aw.z : aw.w;
				if (SPREAD) vmem_slot(u * NAF + f);
				if (ABL & 64) { acc[f][0][0] ^= (int)w ^ bfrag[0][0]; continue; }
				v4i val, b1;
				if (ABL & 4) { val = (v4i){(int)w, (int)w, (int)w, (int)w}; b1 = val; }
				else mf_unpack(w, val, b1);
#pragma unroll
				for (int b = 0; b < NBFV; b++)
					acc[f][b] = __builtin_amdgcn_mfma_i32_16x16x64_i8(val, bfrag[b], acc[f][b], 0, 0, 0);
				if (HAS_B1 && !(ABL & 2)) acc[f][NBF - 1] = __builtin_amdgcn_mfma_i32_16x16x64_i8(b1, bfrag[NBF - 1], acc[f][NBF - 1], 0, 0, 0);
				// samples beyond N have all-zero limbs, so stray codes there add nothing
				const uint32_t m3 = w & (w >> 1) & LO_MASK;
				if (!(ABL & 1) && __ballot(m3 != 0)) {
					saw_missing = true;
					v4i ms;
#pragma unroll
					for (int k = 0; k < 4; k++) ms[k] = (int)((m3 >> (2 * k)) & 0x01010101u);
#pragma unroll
					for (int b = 0; b < NBFV; b++)
						accm[f][b] = __builtin_amdgcn_mfma_i32_16x16x64_i8(ms, bfrag[b], accm[f][b], 0, 0, 0);
				}
			}
		}
		if (ABL & 1024) { __builtin_amdgcn_sched_barrier(0); MF_STAMP(st_prev); __builtin_amdgcn_sched_barrier(0); st_comp += st_prev - sb; }
		if (hh == AW - 1) {
#pragma unroll
			for (int f = 0; f < NAF; f++)
#pragma unroll
				for (int p = 0; p < AW; p++) acur[f][p] = anxt[f][p];
		}
	};
	if (WIDE) {
		for (int t = t0; t < t1; t += 2) {
			tile(t, std::integral_constant<int, 0>());
			tile(t + 1, std::integral_constant<int, AW - 1>());
		}
	} else {
		for (int t = t0; t < t1; t++) tile(t, std::integral_constant<int, 0>());
	}
	if ((ABL & 1024) && lane == 0) {
		unsigned long long *dbg = reinterpret_cast<unsigned long long *>(accbuf + (size_t)M * acc_stride) +
			((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * MF_WAVES + wid) * 4;
		dbg[0] = st_wait; dbg[1] = st_issue; dbg[2] = st_comp; dbg[3] = (unsigned long long)(t1 - t0);
	}

	// ---- results: integer atomics (exact, order-independent)
#pragma unroll
	for (int f = 0; f < NAF; f++) {
#pragma unroll
		for (int reg = 0; reg < 4; reg++) {
			const int v = vbase + 16 * f + kg * 4 + reg;
			if (v < M) {
				int *dst = accbuf + (size_t)v * acc_stride;
#pragma unroll
				for (int b = 0; b < NBF; b++) atomicAdd(&dst[b * 16 + r], acc[f][b][reg]);
				if (saw_missing) {
#pragma unroll
					for (int b = 0; b < NBFV; b++)
						if (accm[f][b][reg] != 0) atomicAdd(&dst[NCOL + b * 16 + r], accm[f][b][reg]);
				}
			}
		}
	}
}

#ifndef MF_KERNEL_ONLY   /* tools/mfma_ablate.hip takes the contraction kernel alone */
// value = hi * 2^32 + lo, both parts small enough to be exact in a double
struct HiLo { long long hi, lo; };
__device__ __forceinline__ double hl_to_double(HiLo x) { return (double)x.hi * 4294967296.0 + (double)x.lo; }
__device__ __forceinline__ HiLo hl(long long hi, long long lo) { HiLo x; x.hi = hi; x.lo = lo; return x; }
__device__ __forceinline__ HiLo hl_axpy(long long a, HiLo x, HiLo y) { return hl(a * x.hi + y.hi, a * x.lo + y.lo); }

// limb sums of one column (nl <= 7 limbs) -> HiLo
__device__ __forceinline__ HiLo mf_limbs(const int *a, int nl = MF_NLIMB)
{
	long long lo = 0, hi = 0;
#pragma unroll
	for (int l = 3; l >= 0; l--) lo = lo * 256 + (l < nl ? a[l] : 0);
#pragma unroll
	for (int l = MF_NLIMB - 1; l >= 4; l--) hi = hi * 256 + (l < nl ? a[l] : 0);
	return hl(hi, lo);
}

#endif /* MF_KERNEL_ONLY */

// Lane-map self-test of v_mfma_i32_16x16x64_i8 with asymmetric integer data:
//   A[row l&15][k = 16(l>>4)+j], B[k = 16(l>>4)+j][col l&15], D[(l>>4)*4+reg][l&15]
__global__ void mfma_selftest_kernel(const int8_t *A, const int8_t *B, int *D)
{
	const int lane = threadIdx.x, r = lane & 15, kg = lane >> 4;
	v4i a, b, c = {0, 0, 0, 0};
	for (int k = 0; k < 4; k++) {
		int av = 0, bv = 0;
		for (int j = 0; j < 4; j++) {
			av |= (int)(uint8_t)A[r * 64 + 16 * kg + 4 * k + j] << (8 * j);
			bv |= (int)(uint8_t)B[(16 * kg + 4 * k + j) * 16 + r] << (8 * j);
		}
		a[k] = av; b[k] = bv;
	}
	c = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c, 0, 0, 0);
	for (int reg = 0; reg < 4; reg++) D[(kg * 4 + reg) * 16 + r] = c[reg];
}
