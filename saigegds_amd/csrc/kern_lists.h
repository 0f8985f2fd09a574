// kern_lists.h -- the sparse side of a block of ROW-MAJOR 2-bit rows, built in one pass over the rows:
// the positions of the missing genotypes (what s3_t3_kernel gathers over) and, for resident blocks, the
// carrier lists of the rare variants (what spa5_kernel walks).  Replaces round 3's two walks over a tiled
// copy of the rows (tile transpose + counts, then the fill).
// Part of libsaigehip.so; also included alone by tools/score3_bench.hip (needs only s3_layout.h).
//
// The reference finds both per variant inside the per-variant call: f64_af_ac_impute walks the N doubles for
// the missing ones (src/vectorization.cpp:186-205), f64_nonzero_index lists the carriers (:209-215).
#pragma once
#include "s3_layout.h"

// inclusive prefix sum over the 64 lanes by DPP moves (as dev_common.h wave_scan_incl_i); lane 63 holds the total
__device__ __forceinline__ int lst_scan_incl(int v)
{
	v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);   // row_shr:1
	v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);   // row_shr:2
	v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);   // row_shr:4
	v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);   // row_shr:8
	v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);   // row_bcast:15 -> rows 1, 3
	v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);   // row_bcast:31 -> rows 2, 3
	return v;
}

// a 16-byte piece of a row, read once: the streaming hint keeps the rows from pushing the table Q (which the same
// kernel gathers from, a few MB per sample range) out of the L2
__device__ __forceinline__ uint4 lst_load_stream(const void *p)
{
	typedef uint32_t lst_u4 __attribute__((ext_vector_type(4)));
	const lst_u4 v = __builtin_nontemporal_load(reinterpret_cast<const lst_u4 *>(p));
	return make_uint4(v.x, v.y, v.z, v.w);
}

__device__ __forceinline__ void lst_store_stream(void *p, const uint4 &w)
{
	typedef uint32_t lst_u4 __attribute__((ext_vector_type(4)));
	lst_u4 v; v.x = w.x; v.y = w.y; v.z = w.z; v.w = w.w;
	__builtin_nontemporal_store(v, reinterpret_cast<lst_u4 *>(p));
}

__device__ __forceinline__ uint32_t lst_keep(int keep)
{
	return (keep >= 16) ? 0xFFFFFFFFu : ((keep <= 0) ? 0u : ((1u << (2 * keep)) - 1u));
}

// Per-block list state on the device (all of it written by the kernels below; the cursors zeroed by the host
// when the block is created and by s3_lists_finish_kernel after every load).
//   lstart / lcnt  [nr][ld]: where in `idx` the missing genotypes of (sample range g, variant v) are listed and
//                  how many; lcnt = -1: the pool was full (the variant takes the FP64 kernel)
//   idx            the pool: sample indices of a (g, v) segment (in the wave's lane order); segments in the order the waves
//                  finished (one atomic add per segment: T3 sums are exact integers, any order gives the same bits).
//                  The pool is cut into S3_NSUB sub-pools with a cursor each (a workgroup's waves take the sub-pool
//                  blockIdx % S3_NSUB): 800 000 atomic adds on ONE address serialise at ~11 ns each (9.6 ms for a
//                  block of 50 000 variants, measured), on 1024 lines they cost nothing measurable.
//   nzp / n2p      [nr][ld] non-zero codes / codes 2 per (g, v) (resident blocks only: the carrier lists' sizes)
#define S3_NSUB 1024
#define S3_CURSOR_STRIDE 32        /* unsigned per cursor: one 128-byte line each */
struct S3Lists {
	unsigned *idx;
	unsigned idx_cap;              // entries of the whole pool; a sub-pool holds idx_cap / nsub
	unsigned *cursor;              // [S3_NSUB * S3_CURSOR_STRIDE] entries handed out per sub-pool
	unsigned *lstart;
	int *lcnt;
	int *nzp, *n2p;        // may be null
	size_t ld;             // leading dimension of the [nr][ld] arrays (>= the block's variants)
	int nr;                // sample ranges = s3_nranges(ntile)
	int nsub;              // sub-pools in use: a power of two <= min(S3_NSUB, workgroups of a full load), so that small blocks keep large sub-pools
	unsigned subcap;       // idx_cap / nsub
	int rt[S3_NR + 1];     // rt[g] = s3_range_t0(g, ntile, nr): the kernels below are short, a 64-bit division per wave showed in their time
};
// the host's part of the struct: ranges and sub-pools (idx_cap, nr set; full_wg = workgroups of a full load)
static inline void s3_lists_setup(S3Lists &L, int ntile, size_t full_wg)
{
	for (int g = 0; g <= S3_NR; g++) L.rt[g] = s3_range_t0(g < L.nr ? g : L.nr, ntile, L.nr);
	int ns = 1;
	while (ns * 2 <= S3_NSUB && (size_t)ns * 2 <= full_wg) ns *= 2;
	L.nsub = ns;
	L.subcap = L.idx_cap / (unsigned)ns;
}

// One wave per (variant v, sample range g), four variants of a range per workgroup: the range's 16-byte pieces (64 samples) are read with all lanes'
// loads in flight at once (MAXLD uint4 per lane: 8 KiB per wave; a range of N = 430 000 is 6.6 KiB), the
// missing codes counted, ONE atomic add reserves the segment, the lanes write their sample indices behind a
// wave prefix (a lane's entries are contiguous: the segment is in lane order, not in sample order -- T3 does
// not care).  The kernel has to stay under the time the rows take to stream (0.9 ms per 5.4 GB): a piece's four
// masks are folded into one 64-bit word (bit 2 s + h: sample s + 16 h of dword pair h) so that the bit walk is one
// loop per piece, and the mask of the samples >= N is applied only by the waves whose range holds the row's end.
// COPY: the pieces are also stored to dst (rows of dst_bpv bytes: the resident block's copy).  COUNTS: non-zero
// codes and codes 2 per (g, v) (the sizes of the carrier lists).
// A range longer than the registers hold (N > 16 x MAXLD x 4096) is walked twice (the second time from L2).
// rows: bpv >= 64 ntile, 16-byte aligned.  v_first: the block's index of rows[0] (chunked host loads).
template <int MAXLD, bool COPY, bool COUNTS>
__global__ void __launch_bounds__(256)
s3_lists_kernel(const uint8_t *__restrict__ rows, size_t bpv, int N, int m, int v_first, int ntile, S3Lists L,
	uint8_t *__restrict__ dst, size_t dst_bpv)
{
	const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
	// workgroup = (range g, four consecutive variants): its four segments are reserved by ONE atomic add and lie
	// behind each other, so that a wave of the T3 pass (eight consecutive variants of a range) reads two runs
	// grid = (groups of four variants, ranges)
	const int g = (int)blockIdx.y, vl = (int)blockIdx.x * 4 + wid, v = v_first + vl;
	const bool live = vl < m;
	const int t0 = L.rt[g], t1 = L.rt[g + 1];
	const int npiece = live ? (t1 - t0) * 4 : 0, p0 = t0 * 4;   // pieces of the range; first piece of the row
	const uint8_t *src = rows + (size_t)(live ? vl : 0) * bpv + (size_t)t0 * 64;
	uint8_t *out = COPY ? dst + (size_t)(live ? v : 0) * dst_bpv + (size_t)t0 * 64 : nullptr;
	const size_t e = (size_t)g * L.ld + v;
	__shared__ int sh_tot[4];
	__shared__ unsigned sh_base;
	const unsigned sub = (blockIdx.y * gridDim.x + blockIdx.x) & (unsigned)(L.nsub - 1), subcap = L.subcap;
	const bool tail = (p0 + npiece) * 64 > N;                   // (wave-uniform) the range reaches past the last sample
	// the piece with the samples >= N cleared
	auto clip = [&](uint4 wv, int p) -> uint4 {
		if (!tail) return wv;
		const int k0 = N - (p0 + p) * 64;
		return make_uint4(wv.x & lst_keep(k0), wv.y & lst_keep(k0 - 16), wv.z & lst_keep(k0 - 32), wv.w & lst_keep(k0 - 48));
	};
	// missing codes of a (clipped) piece as one 64-bit mask: bit 32 h + 2 s + u: sample 32 h + 16 u + s
	auto miss64 = [&](const uint4 &wv) -> unsigned long long {
		const uint32_t a = wv.x & (wv.x >> 1) & 0x55555555u, b = wv.y & (wv.y >> 1) & 0x55555555u;
		const uint32_t c = wv.z & (wv.z >> 1) & 0x55555555u, d = wv.w & (wv.w >> 1) & 0x55555555u;
		return ((unsigned long long)(c | (d << 1)) << 32) | (a | (b << 1));
	};
	auto emit = [&](unsigned long long mm, int p, unsigned o) -> unsigned {
		while (mm) {
			const int b = __ffsll((long long)mm) - 1;
			mm &= mm - 1;
			L.idx[o++] = (unsigned)((p0 + p) * 64 + (b >> 5) * 32 + (b & 1) * 16 + ((b & 31) >> 1));
		}
		return o;
	};
	auto counts = [&](const uint4 &wv, int &cz, int &c2) {
		const uint32_t d[4] = {wv.x, wv.y, wv.z, wv.w};
#pragma unroll
		for (int u = 0; u < 4; u++) {
			cz += __popc((d[u] | (d[u] >> 1)) & 0x55555555u);
			c2 += __popc(~d[u] & (d[u] >> 1) & 0x55555555u);
		}
	};
	// the workgroup's four segments in one reservation (every wave of the workgroup gets here)
	auto reserve = [&](int tot, unsigned &start) -> bool {
		if (lane == 0) sh_tot[wid] = tot;
		__syncthreads();
		const int all = sh_tot[0] + sh_tot[1] + sh_tot[2] + sh_tot[3];
		// (a workgroup that holds a row of missing codes would take its three neighbours down with it: beyond an
		// eighth of the sub-pool every wave reserves for itself)
		const bool together = (unsigned)all <= subcap / 8;
		if (threadIdx.x == 0) sh_base = (together && all > 0) ? atomicAdd(L.cursor + (size_t)sub * S3_CURSOR_STRIDE, (unsigned)all) : 0u;
		__syncthreads();
		unsigned base = sh_base;
		bool ok;
		if (together) {
			ok = (unsigned long long)base + (unsigned long long)all <= (unsigned long long)subcap;
			start = sub * subcap + base + (unsigned)((wid > 0 ? sh_tot[0] : 0) + (wid > 1 ? sh_tot[1] : 0) + (wid > 2 ? sh_tot[2] : 0));
		} else {
			base = 0;
			const bool fits = (unsigned)tot <= subcap;          // (a segment larger than the sub-pool does not touch the cursor)
			if (lane == 0 && tot > 0 && fits) base = atomicAdd(L.cursor + (size_t)sub * S3_CURSOR_STRIDE, (unsigned)tot);
			base = (unsigned)__builtin_amdgcn_readfirstlane((int)base);
			ok = fits && (unsigned long long)base + (unsigned long long)tot <= (unsigned long long)subcap;
			start = sub * subcap + base;
		}
		if (lane == 0 && live) { L.lstart[e] = ok ? start : 0u; L.lcnt[e] = ok ? tot : -1; }
		return ok && tot > 0;
	};
	if (npiece <= MAXLD * 64) {
		unsigned long long mk[MAXLD];
		int c = 0, cz = 0, c2 = 0;
#pragma unroll
		for (int k = 0; k < MAXLD; k++) {
			const int p = k * 64 + lane;
			uint4 wv = make_uint4(0u, 0u, 0u, 0u);
			if (p < npiece) wv = lst_load_stream(src + (size_t)p * 16);
			if (COPY) { if (p < npiece) lst_store_stream(out + (size_t)p * 16, wv); }
			wv = clip(wv, p);
			mk[k] = miss64(wv);
			c += __popcll(mk[k]);
			if (COUNTS) counts(wv, cz, c2);
		}
		const int incl = lst_scan_incl(c);
		const int tot = __builtin_amdgcn_readlane(incl, 63);
		if (COUNTS) {
			cz = __builtin_amdgcn_readlane(lst_scan_incl(cz), 63);
			c2 = __builtin_amdgcn_readlane(lst_scan_incl(c2), 63);
			if (lane == 0 && live) { L.nzp[e] = cz; L.n2p[e] = c2; }
		}
		unsigned start;
		if (!reserve(tot, start)) return;
		unsigned o = start + (unsigned)(incl - c);
#pragma unroll
		for (int k = 0; k < MAXLD; k++) o = emit(mk[k], k * 64 + lane, o);
		return;
	}
	// long ranges: count, reserve, then walk again
	int c = 0, cz = 0, c2 = 0;
	for (int p = lane; p < npiece; p += 64) {
		uint4 wv = *reinterpret_cast<const uint4 *>(src + (size_t)p * 16);
		if (COPY) *reinterpret_cast<uint4 *>(out + (size_t)p * 16) = wv;
		wv = clip(wv, p);
		c += __popcll(miss64(wv));
		if (COUNTS) counts(wv, cz, c2);
	}
	const int incl = lst_scan_incl(c);
	const int tot = __builtin_amdgcn_readlane(incl, 63);
	if (COUNTS) {
		cz = __builtin_amdgcn_readlane(lst_scan_incl(cz), 63);
		c2 = __builtin_amdgcn_readlane(lst_scan_incl(c2), 63);
		if (lane == 0 && live) { L.nzp[e] = cz; L.n2p[e] = c2; }
	}
	unsigned start;
	if (!reserve(tot, start)) return;
	unsigned o = start + (unsigned)(incl - c);
	for (int p = lane; p < npiece; p += 64)
		o = emit(miss64(clip(*reinterpret_cast<const uint4 *>(src + (size_t)p * 16), p)), p, o);
}

// The row-major call's form of the two kernels (list pass + sparse T3 pass) in one: the wave that finds the missing
// genotypes of (variant v, range g) does not list them in global memory but compacts them into its 1 KiB of LDS
// and gathers their rows of Q right away (PP lanes per entry, the columns; 64 / PP entries per step), so that the
// rows are still read once, the sparse sums cost the gathers' latency under the stream instead of a kernel of
// their own (0.33 ms at N = 430 000), and no pool is needed.  part[g][v][c] = {hi, lo} as s3_t3_kernel leaves them;
// lcnt[g][v] = the count, or -1 when the segment holds more than S3_LT_CAP entries (the variant then takes the
// FP64 kernel; a block with that many missing genotypes is the three-plane form's business).
#define S3_LT_CAP 256
// Measured on the way (tools/README.md, round 4): the wave's prologue held two 64-bit divisions (the range's first
// tile, the workgroup's place in a 1-D grid) -- now grid = (four variants, range) and the ranges' first tiles come from
// the host; the rows are read with the streaming hint (lst_load_stream), or they push Q out of the L2 and 40 % of the
// gathers go to memory (1.08 -> 0.91 ms per 5.4 GB at N = 430 000); a form with a third fewer vector instructions
// (detect pieces with a missing code first, compact them, read them again) was slower, its second read missing the L2.
template <int MAXLD, int PP>
__global__ void __launch_bounds__(256)
s3_lists_t3_kernel(const uint8_t *__restrict__ rows, size_t bpv, int N, int m, int ntile, S3Lists L,
	int P, const long long *__restrict__ Q, long long *__restrict__ part)
{
	constexpr int TPE = 64 / PP;
	__shared__ unsigned ent[4][S3_LT_CAP];
	const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
	const int g = (int)blockIdx.y, v = (int)blockIdx.x * 4 + wid;
	if (v >= m) return;                                        // (no workgroup barrier below)
	const int t0 = L.rt[g], t1 = L.rt[g + 1];
	const int npiece = (t1 - t0) * 4, p0 = t0 * 4;
	const uint8_t *src = rows + (size_t)v * bpv + (size_t)t0 * 64;
	const size_t e = (size_t)g * L.ld + v;
	const bool tail = (p0 + npiece) * 64 > N;
	auto clip = [&](uint4 wv, int p) -> uint4 {
		if (!tail) return wv;
		const int k0 = N - (p0 + p) * 64;
		return make_uint4(wv.x & lst_keep(k0), wv.y & lst_keep(k0 - 16), wv.z & lst_keep(k0 - 32), wv.w & lst_keep(k0 - 48));
	};
	auto miss64 = [&](const uint4 &wv) -> unsigned long long {
		const uint32_t a = wv.x & (wv.x >> 1) & 0x55555555u, b = wv.y & (wv.y >> 1) & 0x55555555u;
		const uint32_t c = wv.z & (wv.z >> 1) & 0x55555555u, d = wv.w & (wv.w >> 1) & 0x55555555u;
		return ((unsigned long long)(c | (d << 1)) << 32) | (a | (b << 1));
	};
	auto emit = [&](unsigned long long mm, int p, int o) -> int {
		while (mm) {
			const int b = __ffsll((long long)mm) - 1;
			mm &= mm - 1;
			if (o < S3_LT_CAP) ent[wid][o] = (unsigned)((p0 + p) * 64 + (b >> 5) * 32 + (b & 1) * 16 + ((b & 31) >> 1));
			o++;
		}
		return o;
	};
	int tot = 0;
	if (npiece <= MAXLD * 64) {
		unsigned long long mk[MAXLD];
		int c = 0;
#pragma unroll
		for (int k = 0; k < MAXLD; k++) {
			const int p = k * 64 + lane;
			uint4 wv = make_uint4(0u, 0u, 0u, 0u);
			if (p < npiece) wv = lst_load_stream(src + (size_t)p * 16);
			mk[k] = miss64(clip(wv, p));
			c += __popcll(mk[k]);
		}
		const int incl = lst_scan_incl(c);
		tot = __builtin_amdgcn_readlane(incl, 63);
		int o = incl - c;
#pragma unroll
		for (int k = 0; k < MAXLD; k++) o = emit(mk[k], k * 64 + lane, o);
	} else {
		for (int pb = 0; pb < npiece; pb += 64) {             // (wave-uniform bounds)
			const int p = pb + lane;
			uint4 wv = make_uint4(0u, 0u, 0u, 0u);
			if (p < npiece) wv = lst_load_stream(src + (size_t)p * 16);
			const unsigned long long mm = miss64(clip(wv, p));
			const int c = __popcll(mm), incl = lst_scan_incl(c);
			emit(mm, p, tot + incl - c);
			tot += __builtin_amdgcn_readlane(incl, 63);
		}
	}
	const bool ok = tot <= S3_LT_CAP;
	if (lane == 0) L.lcnt[e] = ok ? tot : -1;
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // the wave's own LDS writes, in order
	__builtin_amdgcn_wave_barrier();
	const int c = lane % PP, t = lane / PP;
	long long hi = 0, lo = 0;
	if (ok) {
#pragma unroll 4
		for (int eb = 0; eb < tot; eb += TPE) {
			const int ei = eb + t;
			if (ei < tot && c < P) {
				const long long q = Q[(size_t)ent[wid][ei] * P + c];
				hi += q >> 32; lo += q & 0xFFFFFFFFll;
			}
		}
	}
#pragma unroll
	for (int o = PP; o < 64; o <<= 1) { hi += __shfl_xor(hi, o, 64); lo += __shfl_xor(lo, o, 64); }
	if (t == 0 && c < P) {
		long long *o = part + (((size_t)g * m + v) * P + c) * 2;
		o[0] = hi; o[1] = lo;
	}
}

// per variant: n3 = its listed missing genotypes, ovf = 1 when a range found the pool full; with the carrier
// counts (resident blocks) also nzv / n2v, the inputs of s3_ingest_clist_count_kernel
__global__ void __launch_bounds__(256)
s3_lists_finish_kernel(int M, S3Lists L, int *__restrict__ n3, uint8_t *__restrict__ ovf, int *__restrict__ nzv, int *__restrict__ n2v,
	int *__restrict__ info /* may be null: [0] += listed missing genotypes / 64, [1] += variants with a full pool */)
{
	const int v = blockIdx.x * 256 + threadIdx.x;
	if (v < S3_NSUB) L.cursor[(size_t)v * S3_CURSOR_STRIDE] = 0u;   // the pool is handed out: the next load starts at the sub-pools' heads (grid >= S3_NSUB threads)
	if (v >= M) return;
	int t = 0, z = 0, c2 = 0;
	bool over = false;
	for (int g = 0; g < L.nr; g++) {
		const int c = L.lcnt[(size_t)g * L.ld + v];
		over |= c < 0;
		t += c < 0 ? 0 : c;
		if (nzv) { z += L.nzp[(size_t)g * L.ld + v]; c2 += L.n2p[(size_t)g * L.ld + v]; }
	}
	n3[v] = t;
	ovf[v] = over ? 1 : 0;
	if (nzv) { nzv[v] = z; n2v[v] = c2; }
	if (info) {
		if (t >= 64) atomicAdd(info, t >> 6);
		else if (t > 0 && (v & 63) == 0) atomicAdd(info, 1);      // (short rows: a coarse count is enough)
		if (over) atomicAdd(info + 1, 1);
	}
}

// Carrier lists of the variants s3_ingest_clist_kernel gave room to (corient 1: the non-zero codes, 2: the codes
// other than 2), from the block's row-major rows: wave per (variant, range), ascending samples, sample | code << 30
// at cidx[cptr[v] + carriers of the earlier ranges ..).  Variants without a list cost nothing (no row is read).
template <int MAXLD>
__global__ void __launch_bounds__(256)
s3_clist_fill_kernel(const uint8_t *__restrict__ rows, size_t bpv, int N, int M, int ntile, S3Lists L,
	const uint8_t *__restrict__ corient, const unsigned *__restrict__ cptr, unsigned *__restrict__ cidx)
{
	const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
	const long long w = (long long)blockIdx.x * 4 + wid;
	if (w >= (long long)M * L.nr) return;
	const int v = (int)(w / L.nr), g = (int)(w % L.nr);
	const int orient = corient[v];
	if (!orient) return;
	const int t0 = s3_range_t0(g, ntile, L.nr), t1 = s3_range_t0(g + 1, ntile, L.nr);
	const int npiece = (t1 - t0) * 4, p0 = t0 * 4;
	const uint8_t *src = rows + (size_t)v * bpv + (size_t)t0 * 64;
	const uint32_t zx = orient == 2 ? 0xAAAAAAAAu : 0u;
	// carriers of the earlier ranges: non-zero codes, or (flipped) the samples of the range that are not code 2
	unsigned o = cptr[v];
	for (int gg = 0; gg < g; gg++) {
		if (orient == 1) o += (unsigned)L.nzp[(size_t)gg * L.ld + v];
		else {
			const int s0 = s3_range_t0(gg, ntile, L.nr) * 256, s1 = min(N, s3_range_t0(gg + 1, ntile, L.nr) * 256);
			o += (unsigned)(max(0, s1 - s0) - L.n2p[(size_t)gg * L.ld + v]);
		}
	}
	for (int pb = 0; pb < npiece; pb += MAXLD * 64) {
		uint4 held[MAXLD];
#pragma unroll
		for (int k = 0; k < MAXLD; k++) {
			const int p = pb + k * 64 + lane;
			held[k] = p < npiece ? *reinterpret_cast<const uint4 *>(src + (size_t)p * 16) : make_uint4(zx, zx, zx, zx);
		}
#pragma unroll
		for (int k = 0; k < MAXLD; k++) {
			const int p = pb + k * 64 + lane;
			if (pb + k * 64 >= npiece) break;                 // (wave-uniform)
			const uint32_t d[4] = {held[k].x, held[k].y, held[k].z, held[k].w};
			uint32_t z[4];
			int c = 0;
#pragma unroll
			for (int u = 0; u < 4; u++) {
				const uint32_t x = (d[u] ^ zx) & lst_keep(p < npiece ? N - (p0 + p) * 64 - 16 * u : 0);
				z[u] = (x | (x >> 1)) & 0x55555555u;
				c += __popc(z[u]);
			}
			const int incl = lst_scan_incl(c);
			unsigned oo = o + (unsigned)(incl - c);
#pragma unroll
			for (int u = 0; u < 4; u++) {
				uint32_t mm = z[u];
				while (mm) {
					const int b = __ffs(mm) - 1;
					mm &= mm - 1;
					cidx[oo++] = (unsigned)((p0 + p) * 64 + u * 16 + (b >> 1)) | (((d[u] >> b) & 3u) << 30);
				}
			}
			o += (unsigned)__builtin_amdgcn_readlane(incl, 63);
		}
	}
}
