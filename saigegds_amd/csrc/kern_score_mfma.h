// kern_score_mfma.h -- score stage for 2-bit genotypes as an EXACT integer
// contraction on the matrix cores (v_mfma_i32_16x16x64_i8).
// Part of libsaigehip.so (single translation unit: saigehip.hip).
#pragma once

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
// and the rare missing entries (code 3) are summed exactly on a side path,
// T3[c] and n3.  From these, in integer arithmetic,
//     W[c] = V[c] - 3 T3[c] = T1 + 2 T2,   H2 = B1 - T3[mu2] = T2[mu2],
//     AC = V[ones] - 3 n3,   Num = N - n3,
// and with imp = 2 AF the sums the epilogue needs (dev_common.h):
//     no flip:  sum G F = W + imp T3          sum G^2 mu2 = W + 2 H2 + imp^2 T3
//     flip:     sum G F = 2 Ftot - W - imp T3
//               sum G^2 mu2 = 4 (Ftot - S1 - H2 - T3) + S1 + (2-imp)^2 T3,  S1 = W - 2 H2.

typedef int v4i __attribute__((ext_vector_type(4)));

#define MF_NAF 4             /* A fragments (16 variants each) per wave      */
#define MF_WAVES 4           /* waves per workgroup -> 256 variants          */
#define MF_VPW (16 * MF_NAF) /* variants per wave                            */
#define MF_VPB (MF_VPW * MF_WAVES)
#define MF_NLIMB 7
#define MF_MAXP 10           /* 2K+2 supported by the MFMA path (K <= 4)     */
#define MF_QCAP 128          /* per-wave queue of missing entries            */

struct MfTab {
	const uint8_t *Fl;         // [ngrp_pad][ncol][16] int8 limb digits, sample-fastest
	const unsigned long long *Fq;   // [N][P] fixed-point values (two's complement int64)
	int ncol;                  // 16 * (nbfv + 1)
	int nbfv;                  // B fragments used with the value plane
	int col_ones;              // column of the constant 1
	int col_b1;                // first column of the mu2 limbs in the bit-1 fragment
	int ntile;                 // number of 256-sample tiles = ngrp_pad / 16
	int escale[MF_MAXP];       // F = q * 2^-escale
	long long ftot_hi[MF_MAXP];     // sum_i q[i,c] = hi * 2^32 + lo
	long long ftot_lo[MF_MAXP];
};

// 16 two-bit codes -> 16 bytes (value plane) and twice their bit 1 (0/2 bytes)
__device__ __forceinline__ void mf_unpack(uint32_t w, v4i &val, v4i &b1)
{
#pragma unroll
	for (int k = 0; k < 4; k++) {
		const uint32_t t = (w >> (8 * k)) & 0xFFu;
		const uint32_t y = (t | (t << 12)) & 0x000F000Fu;
		const uint32_t z = (y | (y << 6)) & 0x03030303u;
		val[k] = (int)z;
		b1[k] = (int)(z & 0x02020202u);   // bit 1 kept in place: the plane is 2*[code>=2]
	}
}

// Flush the wave's queue of missing entries: one entry per lane, gather the
// fixed-point row and add it to the wave's LDS table (exact integer adds).
template <int P>
__device__ __forceinline__ void mf_flush_missing(int cnt, const uint32_t *q, const MfTab &tb,
	unsigned long long *t3lo, long long *t3hi, int *n3, int lane)
{
	for (int k = lane; k < cnt; k += WAVE) {
		const uint32_t e = q[k];
		const int vl = (int)(e >> 26), i = (int)(e & 0x03FFFFFFu);
		const unsigned long long *fq = tb.Fq + (size_t)i * P;
#pragma unroll
		for (int c = 0; c < P; c++) {
			const unsigned long long x = fq[c];
			atomicAdd(&t3lo[vl * P + c], (unsigned long long)(uint32_t)x);
			atomicAdd((unsigned long long *)&t3hi[vl * P + c], (unsigned long long)((long long)x >> 32));
		}
		atomicAdd(&n3[vl], 1);
	}
}

// grid = (variant tiles of MF_VPB, sample splits); block = 64 * MF_WAVES
template <int NBFV, int P>
__global__ void __launch_bounds__(WAVE * MF_WAVES, 2)
score_mfma_kernel(const uint8_t *__restrict__ packed, size_t bpv, int M, int N, MfTab tb,
	int tiles_per_split, int *__restrict__ accbuf, unsigned long long *__restrict__ t3g_lo,
	long long *__restrict__ t3g_hi, int *__restrict__ n3g)
{
	constexpr int NBF = NBFV + 1;
	constexpr int NCOL = 16 * NBF;
	constexpr int TILE_BYTES = 16 * NCOL * 16;
	extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
	uint8_t *ldsB = smem;                                              // 2 x TILE_BYTES (double buffer)
	unsigned long long *t3lo = reinterpret_cast<unsigned long long *>(smem + 2 * TILE_BYTES);   // [WAVES][64][P]
	long long *t3hi = reinterpret_cast<long long *>(t3lo + MF_WAVES * MF_VPW * P);
	int *n3 = reinterpret_cast<int *>(t3hi + MF_WAVES * MF_VPW * P);   // [WAVES][64]
	uint32_t *mq = reinterpret_cast<uint32_t *>(n3 + MF_WAVES * MF_VPW);   // [WAVES][QCAP]

	const int tid = threadIdx.x, lane = tid & (WAVE - 1), wid = tid / WAVE;
	const int r = lane & 15, kg = lane >> 4;
	const int vbase = blockIdx.x * MF_VPB + wid * MF_VPW;
	const int t0 = blockIdx.y * tiles_per_split;
	const int t1 = min(tb.ntile, t0 + tiles_per_split);

	unsigned long long *my_lo = t3lo + wid * MF_VPW * P;
	long long *my_hi = t3hi + wid * MF_VPW * P;
	int *my_n3 = n3 + wid * MF_VPW;
	uint32_t *my_q = mq + wid * MF_QCAP;
	for (int k = lane; k < MF_VPW * P; k += WAVE) { my_lo[k] = 0; my_hi[k] = 0; }
	my_n3[lane] = 0;
	int qn = 0;   // wave-uniform queue fill

	v4i acc[MF_NAF][NBF];
#pragma unroll
	for (int f = 0; f < MF_NAF; f++)
#pragma unroll
		for (int b = 0; b < NBF; b++) acc[f][b] = (v4i){0, 0, 0, 0};

	const uint8_t *rowp[MF_NAF];
	bool vok[MF_NAF];
#pragma unroll
	for (int f = 0; f < MF_NAF; f++) {
		const int v = vbase + 16 * f + r;
		vok[f] = v < M;
		rowp[f] = packed + (size_t)(vok[f] ? v : 0) * bpv;
	}

	// B tile t -> LDS buffer (t & 1) by LDS-DMA: 1 KiB per wave-instruction, lane-linear
	auto issue_B = [&](int t) {
		const uint8_t *src = tb.Fl + (size_t)t * TILE_BYTES;
		uint8_t *dst = ldsB + (size_t)(t & 1) * TILE_BYTES;
#pragma unroll
		for (int k = wid; k < TILE_BYTES / 1024; k += MF_WAVES)
			__builtin_amdgcn_global_load_lds(
				(const __attribute__((address_space(1))) void *)(src + (size_t)k * 1024 + lane * 16),
				(__attribute__((address_space(3))) void *)(dst + k * 1024), 16, 0, 0);
	};
	// this lane's 64 samples of each of its 4 variants in tile t: dwords 16t+4kg .. +3
	auto load_A = [&](int t, uint4 (&a)[MF_NAF]) {
		const size_t boff = ((size_t)16 * t + 4 * kg) * 4;
#pragma unroll
		for (int f = 0; f < MF_NAF; f++) {
			a[f] = make_uint4(0, 0, 0, 0);
			if (vok[f] && boff + 16 <= bpv) a[f] = *reinterpret_cast<const uint4 *>(rowp[f] + boff);
		}
	};

	uint4 acur[MF_NAF], anxt[MF_NAF];
	if (t0 < t1) { load_A(t0, acur); issue_B(t0); }
	for (int t = t0; t < t1; t++) {
		__syncthreads();   // tile t landed (each wave drained its own DMA), tile t-1 fully consumed
		if (t + 1 < t1) { load_A(t + 1, anxt); issue_B(t + 1); }
		const uint8_t *bt = ldsB + (size_t)(t & 1) * TILE_BYTES;
#pragma unroll
		for (int u = 0; u < 4; u++) {
			const int g = 4 * kg + u;
			v4i bfrag[NBF];
#pragma unroll
			for (int b = 0; b < NBF; b++)
				bfrag[b] = *reinterpret_cast<const v4i *>(bt + ((size_t)(g * NCOL + b * 16 + r)) * 16);
#pragma unroll
			for (int f = 0; f < MF_NAF; f++) {
				const uint32_t w = (u == 0) ? acur[f].x : (u == 1) ? acur[f].y : (u == 2) ? acur[f].z : acur[f].w;
				v4i val, b1;
				mf_unpack(w, val, b1);
#pragma unroll
				for (int b = 0; b < NBFV; b++)
					acc[f][b] = __builtin_amdgcn_mfma_i32_16x16x64_i8(val, bfrag[b], acc[f][b], 0, 0, 0);
				acc[f][NBFV] = __builtin_amdgcn_mfma_i32_16x16x64_i8(b1, bfrag[NBFV], acc[f][NBFV], 0, 0, 0);
				// missing entries (code 3) -> wave queue
				uint32_t m = w & (w >> 1) & LO_MASK;
				const int sbase = (16 * t + g) * 16;
				if (sbase + 16 > N) m &= keep_mask(N - sbase);
				unsigned long long any = __ballot(m != 0);
				while (any) {
					const bool has = m != 0;
					if (has) {
						const int b = __ffs(m) - 1;
						m &= m - 1;
						const int pos = qn + __popcll(any & ((1ull << lane) - 1ull));
						my_q[pos] = ((uint32_t)(16 * f + r) << 26) | (uint32_t)(sbase + (b >> 1));
					}
					qn += __popcll(any);
					any = __ballot(m != 0);
					if (qn > MF_QCAP - WAVE) {
						mf_flush_missing<P>(qn, my_q, tb, my_lo, my_hi, my_n3, lane);
						qn = 0;
					}
				}
			}
		}
#pragma unroll
		for (int f = 0; f < MF_NAF; f++) acur[f] = anxt[f];
	}
	mf_flush_missing<P>(qn, my_q, tb, my_lo, my_hi, my_n3, lane);

	// ---- results: integer atomics (exact, order-independent)
#pragma unroll
	for (int f = 0; f < MF_NAF; f++) {
#pragma unroll
		for (int reg = 0; reg < 4; reg++) {
			const int v = vbase + 16 * f + kg * 4 + reg;
			if (v < M) {
#pragma unroll
				for (int b = 0; b < NBF; b++)
					atomicAdd(&accbuf[(size_t)v * NCOL + b * 16 + r], acc[f][b][reg]);
			}
		}
	}
	{
		const int v = vbase + lane;
		const int c3 = my_n3[lane];
		if (v < M && c3 > 0) {
			atomicAdd(&n3g[v], c3);
#pragma unroll
			for (int c = 0; c < P; c++) {
				atomicAdd(&t3g_lo[(size_t)v * P + c], my_lo[lane * P + c]);
				atomicAdd((unsigned long long *)&t3g_hi[(size_t)v * P + c], (unsigned long long)my_hi[lane * P + c]);
			}
		}
	}
}

// value = hi * 2^32 + lo, both parts small enough to be exact in a double
struct HiLo { long long hi, lo; };
__device__ __forceinline__ double hl_to_double(HiLo x) { return (double)x.hi * 4294967296.0 + (double)x.lo; }
__device__ __forceinline__ HiLo hl(long long hi, long long lo) { HiLo x; x.hi = hi; x.lo = lo; return x; }
__device__ __forceinline__ HiLo hl_axpy(long long a, HiLo x, HiLo y) { return hl(a * x.hi + y.hi, a * x.lo + y.lo); }

// limb sums of one column -> HiLo
__device__ __forceinline__ HiLo mf_limbs(const int *a)
{
	long long lo = 0, hi = 0;
#pragma unroll
	for (int l = 3; l >= 0; l--) lo = lo * 256 + a[l];
#pragma unroll
	for (int l = MF_NLIMB - 1; l >= 4; l--) hi = hi * 256 + a[l];
	return hl(hi, lo);
}

// one thread per variant: integer recombination, then the common epilogue
template <int P>
__global__ void __launch_bounds__(256)
score_mfma_epilogue(int M, DevModel md, MfTab tb, const int *__restrict__ accbuf,
	const unsigned long long *__restrict__ t3g_lo, const long long *__restrict__ t3g_hi,
	const int *__restrict__ n3g, SpaRec *__restrict__ recs, int *__restrict__ counters,
	double *__restrict__ out8, uint8_t *__restrict__ valid)
{
	const int j = blockIdx.x * blockDim.x + threadIdx.x;
	if (j >= M) return;
	const int *a = accbuf + (size_t)j * tb.ncol;
	const int n3 = n3g[j];
	const int N = md.N;
	const long long AC = (long long)a[tb.col_ones] - 3ll * n3;
	const int n2 = a[tb.col_b1 + MF_NLIMB] / 2 - n3;       // bit-1 plane (0/2) against the ones column
	const int n1 = (int)(AC - 2ll * n2);
	const VarHead h = make_head(md, (double)AC, N - n3);
	double *o = out8 + (size_t)j * 8;
	if (!h.pass) { nan_row(o); valid[j] = 0; return; }
	const double imp = 2 * h.AF;
	double acc[P];
	HiLo Wm = hl(0, 0), T3m = hl(0, 0);
#pragma unroll
	for (int c = 0; c < P; c++) {
		const HiLo V = mf_limbs(a + c * MF_NLIMB);
		const HiLo T3 = hl(t3g_hi[(size_t)j * P + c], (long long)t3g_lo[(size_t)j * P + c]);
		const HiLo W = hl_axpy(-3, T3, V);
		const double t3d = hl_to_double(T3);
		double s;
		if (!h.minus) s = hl_to_double(W) + imp * t3d;
		else s = hl_to_double(hl(2 * tb.ftot_hi[c] - W.hi, 2 * tb.ftot_lo[c] - W.lo)) - imp * t3d;
		acc[c] = ldexp(s, -tb.escale[c]);
		if (c == P - 1) { Wm = W; T3m = T3; }
	}
	{   // last column carries G^2
		const HiLo B2 = mf_limbs(a + tb.col_b1);          // = 2 (T2 + T3), the plane holds 0/2
		const HiLo H2 = hl(B2.hi / 2 - T3m.hi, B2.lo / 2 - T3m.lo);   // every limb sum of that plane is even
		const double t3d = hl_to_double(T3m);
		double w;
		if (!h.minus) {
			w = hl_to_double(hl_axpy(2, H2, Wm)) + imp * imp * t3d;
		} else {
			const HiLo S1 = hl_axpy(-2, H2, Wm);
			// 4 (Ftot - S1 - H2 - T3) + S1
			const HiLo R = hl(tb.ftot_hi[P - 1] - S1.hi - H2.hi - T3m.hi, tb.ftot_lo[P - 1] - S1.lo - H2.lo - T3m.lo);
			w = hl_to_double(hl_axpy(4, R, S1)) + (2 - imp) * (2 - imp) * t3d;
		}
		acc[P - 1] = ldexp(w, -tb.escale[P - 1]);
	}
	double cbuf[KMAX], pn, Ssc, v2sc;
	valid[j] = 1;
	if (score_epilogue(md, h, acc, o, cbuf, &pn, &Ssc, &v2sc)) {
		const int slot = atomicAdd(&counters[0], 1);
		SpaRec rr;
		rr.j = j; rr.minus = h.minus; rr.AC2 = h.minus ? (2 * h.Num - h.AC) : h.AC;
		rr.nnz = h.minus ? (N - n2) : (n1 + n2 + n3); rr.pad_ = 0;
		rr.p_noadj = pn; rr.S = Ssc; rr.var2 = v2sc;
		for (int k = 0; k < 4; k++) rr.lut[k] = h.lut[k];
		for (int k = 0; k < KMAX; k++) rr.c[k] = (k < md.K) ? cbuf[k] : 0.0;
		recs[slot] = rr;
	}
	atomicAdd(&counters[1], 1);
}

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
