// kern_score3.h -- score stage v3: the exact-integer contraction of kern_score_mfma.h as a PERSISTENT
// streaming kernel over genotype blocks in the library's tiled device layout.
// Part of libsaigehip.so; also included alone by tools/score3_bench.hip (S3_KERNEL_ONLY).
//
// What changed against score_mfma_kernel (round 2: 1.6 ms per 50 000 variants at N = 430 000, issue-bound):
//   * Device layout ("block"): 16 variants x 256 samples = one contiguous KiB, [fragment][tile][lane][16 B];
//     a wave's row load is 8 whole 128-B lines instead of 16 half lines of 16 different rows.
//   * No missing plane.  The rows keep their code 3, the value plane still sums V = T1 + 2 T2 + 3 T3, but
//     T3 (the sums over the MISSING samples) comes from a sparse pass over the block's list of missing
//     genotypes (s3_t3_kernel): exact int64 sums of the same fixed-point values.  The contraction kernel
//     loses 1.9 of its 5.9 MFMAs per fragment-step, the ballot, the second unpack and 36 accumulators.
//   * Any K <= 16 in ONE pass: up to 13 B fragments per tile (accumulators 4 NAF NBF registers); the rows
//     are streamed once, not once per column group.
//   * Persistent workgroups with a static item list instead of a (variant tile, split) grid: no tail of
//     half-empty rounds, the first tiles of the next item are in flight while the current one ends, and
//     partial sums leave by plain stores into per-item slabs (no atomics, no memset of the accumulators).
//   * Counted waits: the B tile (LDS-DMA) of step n + 1 is issued FIRST in step n, then the row loads of
//     step n + D; the top of a step waits for vmcnt(NAF (D - 1)), so row loads stay in flight across the
//     tile barrier (__syncthreads() drained every load at every tile).
#pragma once
#include <type_traits>

#ifndef S3_V4I_DEFINED
#define S3_V4I_DEFINED
typedef int s3_v4i __attribute__((ext_vector_type(4)));
#endif

#include "s3_layout.h"

// LDS reads the compiler does not see as such (see the kernel), and the waits that go with them: the
// wait is tied to the registers it covers so that no use is scheduled in front of it
// (functions, not macros: clang rejects asm operands that name a lambda's captured variables)
template <int OFF> __device__ __forceinline__ void s3_ds_read(s3_v4i &dst, uint32_t addr)
{
	asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF));
}
template <int N> __device__ __forceinline__ void s3_lgkm_wait(s3_v4i &reg) { asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(reg) : "n"(N)); }
__device__ __forceinline__ void s3_tie(s3_v4i &reg) { asm volatile("" : "+v"(reg)); }
#define S3_DS_READ(dst, addr, off) s3_ds_read<(off)>(dst, addr)
#define S3_LGKM_WAIT(n, reg) s3_lgkm_wait<(n)>(reg)
#define S3_TIE(reg) s3_tie(reg)

template <int I, int N, typename F>
__device__ __forceinline__ void s3_static_for(F &&f)
{
	if constexpr (I < N) { f(std::integral_constant<int, I>()); s3_static_for<I + 1, N>(f); }
}

#ifndef S3_AUX_ROWS
#define S3_AUX_ROWS 2            /* cache policy of the row loaders' DMA: nt (streaming) -- the rows are read once and should not push the B tiles (L2) out; 1-6 % by form, tools/README.md */
#endif
#define S3_WAITCNT_VM(n) __builtin_amdgcn_s_waitcnt(((n) & 0xF) | (((n) >> 4) << 14) | (0x7 << 4) | (0xF << 8))

// Unpack of one dword (16 codes) of a row piece into the two A operands (4 dwords each), as micro-operations
// that the consumer loop places one by one between its MFMAs (every vector instruction costs the SIMD four
// issue cycles, an MFMA eight of its sixteen: at K = 3 the kernel lives on how few of them there are).
//   value plane : byte = code x s_t            b1 plane : byte = 2 [code >= 2] x s_t
// with the per-position scale s_t = (1, 4, 1, 4) for the four codes of a byte: positions 1 and 3 keep their
// code where it stands (bits 2-3 after a shift by 0 or 4), so a dword takes ONE shift and eight masks -- nine
// operations against eleven for unit scales (three shifts, four masks, four masks).  The limb tiles carry
// q_x / s_t(x) for sample x (sgx_init quantises those samples to multiples of 4: two bits of 40 to 56).
template <int OP>
__device__ __forceinline__ void s3_unpack_op(uint32_t w, uint32_t &w4, s3_v4i &val, s3_v4i &b1)
{
	if constexpr (OP == 0) w4 = w >> 4;
	else if constexpr (OP == 1) val[0] = (int)(w & 0x03030303u);
	else if constexpr (OP == 2) val[1] = (int)(w & 0x0C0C0C0Cu);
	else if constexpr (OP == 3) val[2] = (int)(w4 & 0x03030303u);
	else if constexpr (OP == 4) val[3] = (int)(w4 & 0x0C0C0C0Cu);
	else if constexpr (OP == 5) b1[0] = (int)(w & 0x02020202u);
	else if constexpr (OP == 6) b1[1] = (int)(w & 0x08080808u);
	else if constexpr (OP == 7) b1[2] = (int)(w4 & 0x02020202u);
	else if constexpr (OP == 8) b1[3] = (int)(w4 & 0x08080808u);
}
// The three-plane form (MISS): beside the value and bit-1 planes the plane [code == 3] x s_t, multiplied into
// EVERY value column: T3 (the sums over the missing samples) then comes out of the contraction itself and no
// list of the missing genotypes is needed -- one read of the rows, at any missing rate.  15 operations per dword:
// the nine above, then  t = w >> 1,  t4 = w4 >> 1  and four three-input ANDs (v_bitop3_b32 on gfx950)
// w & t & mask: bit 2 s of w & (w >> 1) says that code s is 3; bytes 1 / 4 at the even / odd positions, the
// position scales again.  (Sixteen with t = w & (w >> 1) formed first: one two-input AND more than the fused form.)
#define S3_UNPACK3_OPS 15
template <int OP>
__device__ __forceinline__ void s3_unpack3_op(uint32_t w, uint32_t &w4, uint32_t &t, uint32_t &t4, s3_v4i &val, s3_v4i &b1, s3_v4i &mis)
{
	if constexpr (OP < 9) s3_unpack_op<OP>(w, w4, val, b1);
	else if constexpr (OP == 9) t = w >> 1;
	else if constexpr (OP == 10) t4 = w4 >> 1;
	else if constexpr (OP == 11) mis[0] = (int)(w & t & 0x01010101u);
	else if constexpr (OP == 12) mis[1] = (int)(w & t & 0x04040404u);
	else if constexpr (OP == 13) mis[2] = (int)(w4 & t4 & 0x01010101u);
	else if constexpr (OP == 14) mis[3] = (int)(w4 & t4 & 0x04040404u);
}

// NBF: B fragments per tile (value fragments + the bit-1 fragment, the LAST one).  NAF: A fragments (16
// variants) per consumer wave.  NC consumer waves + NLA row-loader waves + NLB B-loader waves per workgroup
// (one workgroup per CU).
//
// Roles.  Every byte reaches the matrix cores through LDS: the row pieces of a tile (NC NAF KiB, from HBM) are
// LDS-DMA'd DA tiles ahead into a ring of DA + 1 slots by the ROW LOADERS, its B tile (4 NBF KiB, from L2) DB
// tiles ahead into DB + 1 slots by the B LOADERS; they do nothing else (buffer_load .. lds with a scalar
// offset per piece: no vector instruction at all).  The CONSUMER waves issue no vector-memory instruction in
// the loop (LDS reads, unpack, MFMA only), so a full memory queue never stalls a wave that has MFMAs to
// issue.  (One symmetric wave type doing both: 1.22 ms against 0.90 for its arithmetic alone and 0.95 for its
// memory traffic alone.)  One s_barrier per tile joins them: a loader arrives once its pieces of the tile
// have landed (counted vmcnt: the pieces of its younger tiles stay in flight across the barrier), after it
// the loaders refill the slot the consumers have just left.  Row and B loaders are different waves because
// vmcnt counts in order: behind the same counter the rows could not run further ahead than the B tiles.
// LDS per workgroup: (DB + 1) x 4 NBF + (DA + 1) x NC NAF KiB.
// ABL (timing tool only, wrong results): 1 no unpack/MFMA, 2 no row DMA, 4 no B DMA, 8 no LDS reads of B,
//   16 clock stamps (s_memtime / s_memrealtime per workgroup behind the slabs)
// RM = 1: the rows are the caller's ROW-MAJOR 2-bit rows (pl.bpv bytes each, pl.nrow of them), not tiles.  A tile of
//   a row is 64 B -- half a cache line -- and 16 half lines of 16 rows per wave instruction read 15-20 % slower than
//   a tile's contiguous KiB, whole lines of 8 rows at the tiles' rate (measured: tools/README.md, round 4).  So the
//   row ring holds PAIRS of tiles (DA pairs ahead, DA + 1 slots of 2 NPA KiB): a row loader's DMA instruction
//   fetches the 128-byte line (tiles t, t + 1) of rows 8 h .. 8 h + 7 of a fragment into KiB 2 p + h of the slot,
//   lanes 8 i .. 8 i + 7 = the eight 16-byte pieces of row 8 h + i, in the order c = j ^ sigma(i, h),
//   sigma = (i >> 1) | (h << 2), c = 4 (tile parity) + kg: with that order the consumers' ds_read_b128 lane groups
//   (16 lanes = the 64-sample pieces of 16 rows at one kg) touch every bank once.  The consumers' only change is
//   the address of a row piece (per tile parity).  B tiles and the barrier stay per tile.  (Staging the lines
//   through the loaders' registers and ds_write_b128 into a single-tile ring was built first: same memory rate,
//   but the LDS store path -- ~80 B per clock and CU -- cost 0.18 ms of a 1.07-ms kernel.)
//   (RM = 2: timing experiment of the tool, wrong results.)
// MISS: the three-plane form (s3_unpack3_op): a consumer wave keeps a second set of accumulators, the missing plane
//   against the NBF - 1 value fragments; its slab is NAF x (2 NBF - 1) fragment slots, the missing plane's behind
//   the NBF of the two-plane form.  NCB = 1 only.
template <int NBF, int NAF, int NC, int NLA, int NLB, int DA, int DB, int ABL = 0, int NCB = 1, int NBUF_ = 2, int RM = 0, bool MISS = false>
__global__ void __launch_bounds__(64 * (NC + NLA + NLB), (NC + NLA + NLB + 3) / 4)
score3_kernel(const uint8_t *__restrict__ A, const uint8_t *__restrict__ Fl, S3Plan pl, int *__restrict__ out, unsigned long long *__restrict__ stamps)
{
#if __HIP_DEVICE_COMPILE__      /* (the host pass only needs the stub; it does not know the buffer-resource builtins) */
	static_assert(NC % NCB == 0 && NCB >= 1 && NCB <= 4, "consumer waves = variant groups x column groups");
	constexpr int NCV = NC / NCB;                             // variant groups: consumer wave wid = (vg, cg) = (wid / NCB, wid % NCB)
	constexpr int NBWMAX = (NBF + NCB - 1) / NCB;             // B fragments of a column group (the last group may hold fewer)
	constexpr int NCOL = 16 * NBF;
	constexpr int TILE_BYTES = 16 * NCOL * 16;
	constexpr int NPB = TILE_BYTES / 1024;                    // KiB pieces of a B tile = 4 NBF
	constexpr int NPA = NCV * NAF;                            // KiB pieces of a tile's rows
	constexpr int RA = DA + 1, RB = DB + 1;                   // ring slots (RM = 1: the row ring's slots are PAIRS of tiles)
	constexpr int SLOT_A = (RM == 1 ? 2 : 1) * NPA * 1024;
	static_assert(DA >= 1 && DB >= 1, "at least one tile ahead");
	extern __shared__ __attribute__((aligned(16))) uint8_t s3_smem[];   // RB x TILE_BYTES (B), then RA x SLOT_A (rows)

	const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
	const int x = blockIdx.x & 7, y = blockIdx.x >> 3;
	const int g = x % pl.ng, i = y * (8 / pl.ng) + x / pl.ng;
	const int nk = pl.rf + (i < pl.rem * pl.f ? 1 : 0);     // items of this workgroup
	// tile ranges in whole PAIRS of tiles (ntile is even): a pair of tiles is one 128-byte line of a row-major row
	const int T0 = 2 * (int)((long long)g * (pl.ntile / 2) / pl.ng), T1 = 2 * (int)((long long)(g + 1) * (pl.ntile / 2) / pl.ng);
	unsigned long long st0 = 0, sr0 = 0;
	if (ABL & 16) { st0 = __builtin_amdgcn_s_memtime(); sr0 = __builtin_amdgcn_s_memrealtime(); }

	struct Pos { int k, t, t1, vtile, id; };                // wave-uniform stream positions
	auto pos_set = [&](Pos &p, int k) {
		p.k = k;
		if (k >= nk) return;
		if (k < pl.rf) { p.vtile = k * pl.wpg + i; p.t = T0; p.t1 = T1; p.id = g * pl.ipg + p.vtile; }
		else {
			const int q = i % pl.f, vv = i / pl.f, len = T1 - T0;
			p.vtile = pl.rf * pl.wpg + vv;
			p.t = T0 + 2 * (q * (len / 2) / pl.f); p.t1 = T0 + 2 * ((q + 1) * (len / 2) / pl.f);
			p.id = g * pl.ipg + pl.rf * pl.wpg + i;
		}
	};
	auto pos_next = [&](Pos &p) { if (++p.t >= p.t1) pos_set(p, p.k + 1); };
	Pos pc;
	pos_set(pc, 0);

	if (wid >= NC) {
		// ---------------------------------------------------------------- loaders
		const bool rows = wid < NC + NLA;
		const int l = __builtin_amdgcn_readfirstlane(rows ? wid - NC : wid - NC - NLA);
		const int voff = (rows ? s3_dma_lane(lane) : lane) * 16;   // rows: LDS slot `lane` receives the KiB's slot of (variant r, piece kg)
		// row-major rows: LDS slot `lane` receives 16 B of row (lane & 15), piece (lane >> 4); in the block's last
		// fragment the rows past the end are the last row again (their sums are never read)
		const unsigned rm_bpv = (unsigned)pl.bpv;
		Pos pa = pc;
		int sl = 0;                                           // slot of the next tile to issue
		int ahead = 0;                                        // tiles issued beyond the one the next barrier releases
		if (rows && RM == 1) {
			// ---- row-major rows: pairs of tiles, two DMA instructions (rows 0-7, rows 8-15) per fragment and pair
			constexpr int PLO = NPA / NLA, NHI = NPA % NLA;       // loaders l < NHI take PLO + 1 fragments
			static_assert(2 * (PLO + 1) * (DA > 1 ? DA - 1 : 1) < 64, "vmcnt range");
			// lane = 8 i + j: row 8 h + i of the fragment, piece c = j ^ sigma(i, h) of its line
			const int li = lane >> 3, lj = lane & 7;
			const int rlast = pl.nrow - 1 - 16 * (pl.nfrag - 1);       // last row of the block's last fragment
			unsigned vo[2], vol[2];                                    // per-lane byte offsets (h = 0, 1); in the last fragment
#pragma unroll
			for (int h = 0; h < 2; h++) {
				const unsigned c = (unsigned)(lj ^ ((li >> 1) | (h << 2)));
				vo[h] = (unsigned)(8 * h + li) * rm_bpv + c * 16u;
				vol[h] = (unsigned)min(8 * h + li, rlast) * rm_bpv + c * 16u;
			}
			auto issue = [&]() {
				const size_t f0 = (size_t)pa.vtile * pl.fpw;
				const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void *)(A + f0 * 16 * (size_t)pl.bpv), 0, 0xFFFFFFFFu, 0x00020000);
				const int lastf = pl.nfrag - 1 - (int)f0;      // past the end: the last fragment again (never stored)
#pragma unroll
				for (int j = 0; j <= PLO; j++) {
					const int p = l + j * NLA;
					if (p >= NPA || (ABL & 2)) break;
					const int pp = min(p, lastf);
					// (fpw fragments x 16 rows x bpv bytes: far below 4 GiB)
					const int so = (int)((unsigned)pp * 16u * rm_bpv + (unsigned)(pa.t >> 1) * 128u);
#pragma unroll
					for (int h = 0; h < 2; h++)
						__builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (__attribute__((address_space(3))) void *)(s3_smem + RB * TILE_BYTES + sl * SLOT_A + (2 * p + h) * 1024),
							16, (int)(pp == lastf ? vol[h] : vo[h]), so, 0, S3_AUX_ROWS);
				}
				sl = sl + 1 == RA ? 0 : sl + 1;
				pos_next(pa); pos_next(pa);
			};
#pragma unroll
			for (int d = 0; d < DA; d++) if (pa.k < nk) { issue(); ahead++; }
			// a pair per iteration, its two barriers written out (every item is whole pairs): the wait belongs to the first.
			// (With one barrier in the loop and the wait under `if (even tile)` the compiler issued s_waitcnt vmcnt(0) in front
			// of EVERY barrier: the pair fetched behind an even barrier then had one tile step to land, not two.)
			while (pc.k < nk) {
				// the pair this barrier opens has landed; `ahead - 1` younger pairs may fly
				if (ahead == DA && DA > 1) { if (l < NHI) S3_WAITCNT_VM(2 * (PLO + 1) * (DA - 1)); else S3_WAITCNT_VM(2 * PLO * (DA - 1)); }
				else S3_WAITCNT_VM(0);
				__builtin_amdgcn_s_barrier();
				// the consumers have left the pair before this one: its slot takes the pair DA ahead
				ahead--;
				if (pa.k < nk) { issue(); ahead++; }
				pos_next(pc);
				__builtin_amdgcn_s_barrier();                     // the pair's second tile: nothing to wait for, nothing to issue
				pos_next(pc);
			}
		} else if (rows) {
			constexpr int PLO = NPA / NLA, NHI = NPA % NLA;       // loaders l < NHI take PLO + 1 pieces
			static_assert((PLO + 1) * (DA - 1) < 64, "vmcnt range");
			auto issue = [&]() {
				// rows of the item's variant tile: a descriptor at its first fragment, the fragment and the tile
				// in the scalar offset (fpw fragments x ntile KiB: far below 4 GiB)
				const size_t f0 = (size_t)pa.vtile * pl.fpw;
				const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(
					(void *)(A + (RM ? f0 * 16 * (size_t)pl.bpv : f0 * pl.ntile * 1024)), 0, 0xFFFFFFFFu, 0x00020000);
				const int lastf = pl.nfrag - 1 - (int)f0;      // past the end: the last fragment again (never stored)
#pragma unroll
				for (int j = 0; j <= PLO; j++) {
					const int p = l + j * NLA;
					if (p >= NPA || (ABL & 2)) break;
					if constexpr (RM == 1) {
						// (not reached: RM = 1 has its own loop below)
					} else if constexpr (RM == 2) {
						// (timing experiment, wrong results: the same bytes as whole 128-B lines of 8 rows per instruction)
						const int pp = min(p, lastf);
						__builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (__attribute__((address_space(3))) void *)(s3_smem + RB * TILE_BYTES + sl * SLOT_A + p * 1024),
							16, (int)((unsigned)(lane >> 3) * rm_bpv + (unsigned)(lane & 7) * 16u),
							(int)(((unsigned)pp * 16u + (unsigned)(pa.t & 1) * 8u) * rm_bpv + (unsigned)(pa.t >> 1) * 128u), 0, 0);
					} else
					__builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (__attribute__((address_space(3))) void *)(s3_smem + RB * TILE_BYTES + sl * SLOT_A + p * 1024),
						16, voff, (min(p, lastf) * pl.ntile + pa.t) * 1024, 0, 0);
				}
				sl = sl + 1 == RA ? 0 : sl + 1;
				pos_next(pa);
			};
#pragma unroll
			for (int d = 0; d < DA; d++) if (pa.k < nk) { issue(); ahead++; }
			while (pc.k < nk) {
				// the pieces of the tile this barrier releases have landed; `ahead - 1` younger tiles may fly
				if (ahead == DA && DA > 1) { if (l < NHI) S3_WAITCNT_VM((PLO + 1) * (DA - 1)); else S3_WAITCNT_VM(PLO * (DA - 1)); }
				else S3_WAITCNT_VM(0);
				__builtin_amdgcn_s_barrier();
				ahead--;
				if (pa.k < nk) { issue(); ahead++; }
				pos_next(pc);
			}
		} else {
			constexpr int PLO = NPB / NLB, NHI = NPB % NLB;
			static_assert((PLO + 1) * (DB - 1) < 64, "vmcnt range");
			const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void *)Fl, 0, 0xFFFFFFFFu, 0x00020000);
			auto issue = [&]() {
#pragma unroll
				for (int j = 0; j <= PLO; j++) {
					const int b = l + j * NLB;
					if (b >= NPB || (ABL & 4)) break;
					__builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (__attribute__((address_space(3))) void *)(s3_smem + sl * TILE_BYTES + b * 1024),
						16, voff, pa.t * TILE_BYTES + b * 1024, 0, 0);
				}
				sl = sl + 1 == RB ? 0 : sl + 1;
				pos_next(pa);
			};
#pragma unroll
			for (int d = 0; d < DB; d++) if (pa.k < nk) { issue(); ahead++; }
			while (pc.k < nk) {
				if (ahead == DB && DB > 1) { if (l < NHI) S3_WAITCNT_VM((PLO + 1) * (DB - 1)); else S3_WAITCNT_VM(PLO * (DB - 1)); }
				else S3_WAITCNT_VM(0);
				__builtin_amdgcn_s_barrier();
				ahead--;
				if (pa.k < nk) { issue(); ahead++; }
				pos_next(pc);
			}
		}
		return;
	}

	// -------------------------------------------------------------------- consumer
	const int r = lane & 15, kg = lane >> 4;
	const int vg = __builtin_amdgcn_readfirstlane(wid / NCB), cgr = __builtin_amdgcn_readfirstlane(wid % NCB);
	// LDS byte addresses of this lane's 16 B of a row piece and of its B fragment (sample group 4 kg, column r)
	const uint32_t smem_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t *)s3_smem;
	const uint32_t ring_lds = smem_lds + RB * TILE_BYTES + vg * (NAF * 1024) + lane * 16;
	// row-major rows (RM = 1): piece (r, kg) of the pair's tile of parity par sits in KiB 2 p + (r >> 3) of the pair's
	// slot at 16-byte position 8 i + ((4 par + kg) ^ sigma(i, h)), i = r & 7, h = r >> 3 (see the loaders)
	const int rm_sig = ((r & 7) >> 1) | ((r >> 3) << 2);
	const uint32_t ring_rm = smem_lds + RB * TILE_BYTES + vg * (NAF * 2048) + (uint32_t)(r >> 3) * 1024u + (uint32_t)(r & 7) * 128u;
	const uint32_t ring_rm0 = ring_rm + (uint32_t)(kg ^ rm_sig) * 16u, ring_rm1 = ring_rm + (uint32_t)((4 + kg) ^ rm_sig) * 16u;

	// The body of a consumer wave of column group CG: B fragments [B0, B0 + NBW) of the tile against the NAF row
	// pieces of its variant group.  With NCB > 1 the waves of a variant group read the SAME row pieces and unpack
	// them redundantly, each multiplying them into its own share of the columns: a wave's LDS reads per MFMA fall
	// from (NAF + 4 NBF) / (4 NAF NBF) KiB to (NAF' + 4 NBF / NCB) / (4 NAF' NBF / NCB) with NAF' = NCB NAF pieces
	// at the same number of accumulators -- at K >= 9 the LDS port (B reads + the DMA's writes), not the matrix
	// pipe, was what the one-column-group form waited for.
	auto consume = [&](auto CG) {
		constexpr int cg = decltype(CG)::value;
		constexpr int B0 = cg * NBWMAX;
		constexpr int NBW = (NBF - B0 < NBWMAX) ? NBF - B0 : NBWMAX;
		constexpr bool HASB1 = B0 + NBW == NBF;                 // the bit-1 fragment is the LAST of the tile
		static_assert(NBW >= 1, "empty column group");
		const uint32_t bt_lds = smem_lds + (4 * kg * NCOL + r) * 16 + B0 * 256;

		s3_v4i acc[NAF][NBW];
#pragma unroll
		for (int f = 0; f < NAF; f++)
#pragma unroll
			for (int b = 0; b < NBW; b++) acc[f][b] = (s3_v4i){0, 0, 0, 0};

		// B fragments are read one chunk (<= BCH fragments) ahead of the MFMAs that use them; per dword step u the
		// NAF NBW MFMAs run fragment-major (a B fragment feeds NAF consecutive MFMAs) and the unpack operations --
		// the b1 planes of this step, then the value planes of the next -- are dealt out behind them, each group
		// fenced so that the compiler keeps the order.
		constexpr int BCH = NBUF_ > 2 ? 1 : NCB > 1 ? (NBW <= 2 ? NBW : 2) : (NBF <= 4 ? NBF : (NBF >= 9 ? 2 : (NBF % 3 == 0 ? 3 : 4)));
		// NBUF register buffers of one chunk each: the reads run NBUF - 1 chunks ahead of the MFMAs.  (With one
		// consumer wave per SIMD and all of them reading in step behind the tile barrier an LDS read takes ~270
		// cycles to come back: one chunk of two fragments ahead = 2 NAF MFMAs is not enough at NAF <= 6.)
		constexpr int NBUF = NBUF_;
		// PIPE: two sets of value planes, the next dword's made evenly between ALL MFMAs of the current one (few
		// fragments: the unpack is a large share of a step).  Otherwise ONE set leaves the registers to the
		// accumulators, and the next dword's planes of row piece f are made right behind the last MFMA that reads
		// the current ones (those of the wave's last value fragment), under the MFMAs that follow.
		constexpr bool PIPE = NCB == 1 && NBF <= 8;
		constexpr int NCH = (NBW + BCH - 1) / BCH;              // chunks per dword step
		constexpr int NM = NAF * NBW;                           // MFMAs per dword step
		constexpr int NV = HASB1 ? 4 * NAF : 0, NW = 5 * NAF;   // b1 operations of a step, value operations of the next
		constexpr int BLV = HASB1 ? NBW - 2 : NBW - 1;          // the wave's last value fragment (-1: none)
		// b1 planes of this dword: dealt over the MFMAs before the b1 fragment -- without PIPE before the last value
		// fragment, whose slots carry the next dword's value planes (they overwrite w4)
		constexpr int MB = HASB1 ? ((!PIPE && NBW >= 3) ? (NBW - 2) * NAF : (NBW - 1) * NAF) : 0;
		static_assert(!HASB1 || NBW >= 2 || NCB == 1, "a column group of the b1 fragment alone has no slot for its unpack");

		int sa = 0, sb = 0;        // ring slots of the current tile (rows, B)
		int par = 0;               // parity of the tile in the workgroup's stream (every item is whole pairs)
		while (pc.k < nk) {
			__builtin_amdgcn_sched_barrier(0);
			__builtin_amdgcn_s_barrier();
			__builtin_amdgcn_sched_barrier(0);
			// (LDS reads by inline asm with counted lgkmcnt waits: behind an LDS-DMA the compiler puts vmcnt(0) in
			// front of every LDS read it can see)
			const uint32_t a_addr = (RM == 1 ? (par ? ring_rm1 : ring_rm0) : ring_lds) + (uint32_t)(sa * SLOT_A);
			constexpr int AFS = RM == 1 ? 2048 : 1024;            // bytes between a wave's row pieces
			const uint32_t b_addr = bt_lds + (uint32_t)(sb * TILE_BYTES);
			s3_v4i aw[NAF];
			s3_static_for<0, NAF>([&](auto F) { constexpr int f = decltype(F)::value; S3_DS_READ(aw[f], a_addr, f * AFS); });
			s3_v4i bf[NBUF][BCH];
			auto read_chunk = [&](auto CI) {
				constexpr int ci = decltype(CI)::value, u = ci / NCH, b0 = (ci % NCH) * BCH;
				s3_static_for<0, BCH>([&](auto J) {
					constexpr int j = decltype(J)::value;
					if constexpr (b0 + j < NBW) {
						if (ABL & 8) bf[ci % NBUF][j] = (s3_v4i){u, j, r, sa};
						else S3_DS_READ(bf[ci % NBUF][j], b_addr, u * NCOL * 16 + (b0 + j) * 256);
					}
				});
			};
			constexpr int nread_last = NBW - (NCH - 1) * BCH;       // fragments of a dword step's last chunk
			// fragment reads of chunks [c0, c1) of the tile
			auto nreads = [](int c0, int c1) constexpr { int n = 0; for (int c = c0; c < c1 && c < 4 * NCH; c++) n += (c % NCH == NCH - 1) ? nread_last : BCH; return n; };
			s3_static_for<0, NBUF - 1>([&](auto C) { if constexpr (decltype(C)::value < 4 * NCH) read_chunk(C); });
			// the row pieces are back (the first B chunks may still be in flight): value planes of dword 0
			if (!(ABL & 8)) { S3_LGKM_WAIT(nreads(0, NBUF - 1), aw[0]); } else { S3_LGKM_WAIT(0, aw[0]); }
#pragma unroll
			for (int f = 1; f < NAF; f++) S3_TIE(aw[f]);
			s3_v4i val[PIPE ? 2 : 1][NAF], b1[NAF];
			uint32_t w4[PIPE ? 2 : 1][NAF];       // the dword shifted by 4: of the current dword (its b1 planes) and of the next (its value planes)
			// (without PIPE only those of row piece 0: piece f + 1 follows behind the first MFMA of piece f)
			if (!(ABL & 1)) {
				s3_static_for<0, (PIPE ? NW : 5)>([&](auto O) {
					constexpr int o = decltype(O)::value, f = o / 5, op = o % 5;
					s3_unpack_op<op>((uint32_t)aw[f][0], w4[0][f], val[0][f], b1[f]);
				});
			}
			__builtin_amdgcn_sched_barrier(0);
			// Without PIPE the dword step's LAST chunk runs piece-major when it holds two fragments (both are in
			// registers): the planes of piece f are free two MFMAs earlier, and their replacement is spread
			// over two slots instead of five operations behind one MFMA.
			constexpr bool FM = !PIPE && BCH == 2 && nread_last == 2;
			constexpr int M0 = (NCH - 1) * BCH * NAF;              // first slot of the last chunk
			s3_static_for<0, 4>([&](auto U) {
				constexpr int u = decltype(U)::value;
				constexpr int vb = PIPE ? (u & 1) : 0, vn = PIPE ? ((u + 1) & 1) : 0;      // value-plane sets of this dword / the next
				s3_static_for<0, NM>([&](auto MI) {
					constexpr int m = decltype(MI)::value;
					constexpr bool fm = FM && m >= M0;
					constexpr int b = fm ? (NCH - 1) * BCH + (m - M0) % 2 : m / NAF, f = fm ? (m - M0) / 2 : m % NAF;
					constexpr int ch = b / BCH, ci = u * NCH + ch, j = b % BCH;
					if constexpr (m == ch * BCH * NAF) {
						// entering a chunk: start the one NBUF - 1 ahead, then wait for this one
						if constexpr (ci + NBUF - 1 < 4 * NCH) read_chunk(std::integral_constant<int, ci + NBUF - 1>());
						constexpr int inflight = (ABL & 8) ? 0 : nreads(ci + 1, ci + NBUF);
						constexpr int nb = (ch == NCH - 1) ? nread_last : BCH;
						S3_LGKM_WAIT(inflight, bf[ci % NBUF][0]);
#pragma unroll
						for (int jj = 1; jj < nb; jj++) S3_TIE(bf[ci % NBUF][jj]);
					}
					if (ABL & 1) { if (f == 0) acc[0][b][1] ^= bf[ci % NBUF][j][0] ^ aw[b % NAF][u]; }
					else if (HASB1 && b == NBW - 1) acc[f][b] = __builtin_amdgcn_mfma_i32_16x16x64_i8(b1[f], bf[ci % NBUF][j], acc[f][b], 0, 0, 0);
					else acc[f][b] = __builtin_amdgcn_mfma_i32_16x16x64_i8(val[vb][f], bf[ci % NBUF][j], acc[f][b], 0, 0, 0);
					if (!(ABL & 1)) {
						// value planes of the tile's first dword, row piece f + 1
						if constexpr (!PIPE && u == 0 && b == 0 && f + 1 < NAF) {
							s3_static_for<0, 5>([&](auto O) {
								constexpr int op = decltype(O)::value;
								s3_unpack_op<op>((uint32_t)aw[f + 1][0], w4[0][f + 1], val[0][f + 1], b1[f + 1]);
							});
						}
						if constexpr (m < MB) {
							s3_static_for<m * NV / MB, (m + 1) * NV / MB>([&](auto O) {
								constexpr int o = decltype(O)::value, ff = o / 4, op = 5 + o % 4;
								s3_unpack_op<op>((uint32_t)aw[ff][u], w4[vb][ff], val[vb][ff], b1[ff]);
							});
						}
						// value planes of the next dword
						if constexpr (PIPE && u < 3) {
							s3_static_for<m * NW / NM, (m + 1) * NW / NM>([&](auto O) {
								constexpr int o = decltype(O)::value, ff = o / 5, op = o % 5;
								s3_unpack_op<op>((uint32_t)aw[ff][u + 1], w4[vn][ff], val[vn][ff], b1[ff]);
							});
						}
						if constexpr (!PIPE && u < 3) {
							// behind the last MFMA that reads val[f] (that of fragment BLV); with the b1 fragment behind it
							// in a piece-major chunk: two operations there, three behind the b1 MFMA
							constexpr int lo = !fm ? ((b == BLV || (BLV < 0 && b == 0)) ? 0 : 5) : (HASB1 ? (j == 0 ? 0 : 2) : (j == 1 ? 0 : 5));
							constexpr int hi = !fm ? 5 : (HASB1 && j == 0 ? 2 : 5);
							s3_static_for<lo, hi>([&](auto O) {
								constexpr int op = decltype(O)::value;
								s3_unpack_op<op>((uint32_t)aw[f][u + 1], w4[0][f], val[0][f], b1[f]);
							});
						}
					}
					__builtin_amdgcn_sched_barrier(0);
				});
			});
			if (pc.t + 1 == pc.t1) {
				// the item is complete: its slab [wave][f][b][reg][lane] leaves by plain stores (256 B per instruction)
				int *dst = out + ((size_t)pc.id * NC + wid) * (NAF * NBWMAX * 256) + lane;
#pragma unroll
				for (int f = 0; f < NAF; f++)
#pragma unroll
					for (int b = 0; b < NBW; b++)
#pragma unroll
						for (int reg = 0; reg < 4; reg++) {
							dst[((f * NBWMAX + b) * 4 + reg) * 64] = acc[f][b][reg];
							acc[f][b][reg] = 0;
						}
			}
			pos_next(pc);
			if (RM != 1 || par) sa = sa + 1 == RA ? 0 : sa + 1;         // (a pair's slot serves two tiles)
			sb = sb + 1 == RB ? 0 : sb + 1;
			par ^= 1;
		}
	};
	// ---- the three-plane consumer (MISS): per dword step the MFMAs run (fragment, plane)-major -- B fragment b feeds the
	// NAF value MFMAs, then the NAF missing-plane MFMAs, the bit-1 fragment its NAF -- and the 15 NAF operations that
	// make the NEXT dword's planes (a second set of plane registers) are dealt out evenly behind them.
	auto consume3 = [&]() {
		static_assert(!MISS || NCB == 1, "the three-plane form has one column group");
		constexpr int NBUF = NBUF_;                           // chunk buffers: the B reads run NBUF - 1 chunks ahead of their MFMAs
		constexpr int NBV = NBF - 1;                          // value fragments
		constexpr int NVF = 2 * NBV + 1;                      // (fragment, plane) pairs of a dword step
		constexpr int NM = NAF * NVF;                         // MFMAs per dword step
		constexpr int NOPT = S3_UNPACK3_OPS * NAF;            // operations per dword
		constexpr int BCH = NBF <= 4 ? NBF : 2, NCH = (NBF + BCH - 1) / BCH;
		const uint32_t bt_lds = smem_lds + (4 * kg * NCOL + r) * 16;
		s3_v4i acc[NAF][NBF], accm[NAF][NBV > 0 ? NBV : 1];
#pragma unroll
		for (int f = 0; f < NAF; f++) {
#pragma unroll
			for (int b = 0; b < NBF; b++) acc[f][b] = (s3_v4i){0, 0, 0, 0};
#pragma unroll
			for (int b = 0; b < NBV; b++) accm[f][b] = (s3_v4i){0, 0, 0, 0};
		}
		int sa = 0, sb = 0, par = 0;
		while (pc.k < nk) {
			__builtin_amdgcn_sched_barrier(0);
			__builtin_amdgcn_s_barrier();
			__builtin_amdgcn_sched_barrier(0);
			const uint32_t a_addr = (RM == 1 ? (par ? ring_rm1 : ring_rm0) : ring_lds) + (uint32_t)(sa * SLOT_A);
			constexpr int AFS = RM == 1 ? 2048 : 1024;
			const uint32_t b_addr = bt_lds + (uint32_t)(sb * TILE_BYTES);
			s3_v4i aw[NAF];
			s3_static_for<0, NAF>([&](auto F) { constexpr int f = decltype(F)::value; S3_DS_READ(aw[f], a_addr, f * AFS); });
			s3_v4i bf[NBUF][BCH];
			auto read_chunk = [&](auto CI) {
				constexpr int ci = decltype(CI)::value, u = ci / NCH, b0 = (ci % NCH) * BCH;
				s3_static_for<0, BCH>([&](auto J) {
					constexpr int j = decltype(J)::value;
					if constexpr (b0 + j < NBF) S3_DS_READ(bf[ci % NBUF][j], b_addr, u * NCOL * 16 + (b0 + j) * 256);
				});
			};
			constexpr int nread_last = NBF - (NCH - 1) * BCH;
			auto nreads = [](int c0, int c1) constexpr { int n = 0; for (int c = c0; c < c1 && c < 4 * NCH; c++) n += (c % NCH == NCH - 1) ? nread_last : BCH; return n; };
			s3_static_for<0, NBUF - 1>([&](auto C) { if constexpr (decltype(C)::value < 4 * NCH) read_chunk(C); });
			S3_LGKM_WAIT(nreads(0, NBUF - 1), aw[0]);
#pragma unroll
			for (int f = 1; f < NAF; f++) S3_TIE(aw[f]);
			s3_v4i val[2][NAF], b1[2][NAF], mis[2][NAF];
			uint32_t w4[NAF], tt[NAF], m4[NAF];
			s3_static_for<0, NOPT>([&](auto O) {
				constexpr int o = decltype(O)::value, f = o / S3_UNPACK3_OPS, op = o % S3_UNPACK3_OPS;
				s3_unpack3_op<op>((uint32_t)aw[f][0], w4[f], tt[f], m4[f], val[0][f], b1[0][f], mis[0][f]);
			});
			__builtin_amdgcn_sched_barrier(0);
			s3_static_for<0, 4>([&](auto U) {
				constexpr int u = decltype(U)::value, vb = u & 1, vn = (u + 1) & 1;
				s3_static_for<0, NM>([&](auto MI) {
					constexpr int m = decltype(MI)::value, vf = m / NAF, f = m % NAF;
					constexpr int b = vf < 2 * NBV ? vf / 2 : NBV;            // B fragment
					constexpr int plane = vf < 2 * NBV ? (vf & 1) : 2;        // 0 value, 1 missing, 2 bit-1
					constexpr int ch = b / BCH, ci = u * NCH + ch, j = b % BCH;
					if constexpr (f == 0 && plane != 1 && b % BCH == 0) {
						// entering a chunk: start the next one, then wait for this one
						if constexpr (ci + NBUF - 1 < 4 * NCH) read_chunk(std::integral_constant<int, ci + NBUF - 1>());
						constexpr int inflight = nreads(ci + 1, ci + NBUF);
						constexpr int nb = (ch == NCH - 1) ? nread_last : BCH;
						S3_LGKM_WAIT(inflight, bf[ci % NBUF][0]);
#pragma unroll
						for (int jj = 1; jj < nb; jj++) S3_TIE(bf[ci % NBUF][jj]);
					}
					if constexpr (plane == 0) acc[f][b] = __builtin_amdgcn_mfma_i32_16x16x64_i8(val[vb][f], bf[ci % NBUF][j], acc[f][b], 0, 0, 0);
					else if constexpr (plane == 1) accm[f][b] = __builtin_amdgcn_mfma_i32_16x16x64_i8(mis[vb][f], bf[ci % NBUF][j], accm[f][b], 0, 0, 0);
					else acc[f][b] = __builtin_amdgcn_mfma_i32_16x16x64_i8(b1[vb][f], bf[ci % NBUF][j], acc[f][b], 0, 0, 0);
					if constexpr (u < 3) {
						s3_static_for<m * NOPT / NM, (m + 1) * NOPT / NM>([&](auto O) {
							constexpr int o = decltype(O)::value, ff = o / S3_UNPACK3_OPS, op = o % S3_UNPACK3_OPS;
							s3_unpack3_op<op>((uint32_t)aw[ff][u + 1], w4[ff], tt[ff], m4[ff], val[vn][ff], b1[vn][ff], mis[vn][ff]);
						});
					}
					__builtin_amdgcn_sched_barrier(0);
				});
			});
			if (pc.t + 1 == pc.t1) {
				// the item's slab: [wave][f][slot < 2 NBF - 1][reg][lane], the missing plane's fragments behind the NBF
				int *dst = out + ((size_t)pc.id * NC + wid) * (NAF * NVF * 256) + lane;
#pragma unroll
				for (int f = 0; f < NAF; f++) {
#pragma unroll
					for (int b = 0; b < NBF; b++)
#pragma unroll
						for (int reg = 0; reg < 4; reg++) { dst[((f * NVF + b) * 4 + reg) * 64] = acc[f][b][reg]; acc[f][b][reg] = 0; }
#pragma unroll
					for (int b = 0; b < NBV; b++)
#pragma unroll
						for (int reg = 0; reg < 4; reg++) { dst[((f * NVF + NBF + b) * 4 + reg) * 64] = accm[f][b][reg]; accm[f][b][reg] = 0; }
				}
			}
			pos_next(pc);
			if (RM != 1 || par) sa = sa + 1 == RA ? 0 : sa + 1;
			sb = sb + 1 == RB ? 0 : sb + 1;
			par ^= 1;
		}
	};
	if constexpr (MISS) consume3();
	else if constexpr (NCB == 1) consume(std::integral_constant<int, 0>());
	else s3_static_for<0, NCB>([&](auto CG) { if (cgr == decltype(CG)::value) consume(CG); });
	if ((ABL & 16) && tid == 0) {
		stamps[2 * blockIdx.x] = __builtin_amdgcn_s_memtime() - st0;
		stamps[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - sr0;
	}
#endif
}

// Per NBF the product's instantiation of score3_kernel on row-major rows (RM = 1): fragments per consumer wave,
// consumer / row-loader / B-loader waves, pairs of row tiles ahead, B tiles ahead -- what fits 160 KiB of LDS
// ((DB + 1) x 4 NBF + (DA + 1) x 2 NC NAF KiB) and the registers of that many waves, the fastest of the forms
// measured with tools/score3_bench (tools/README.md).
#define S3_FOR_EACH_NBF(X) \
	X(2, 4, 8, 3, 1, 1, 2) X(3, 4, 8, 3, 1, 1, 1) X(4, 4, 8, 3, 1, 1, 1) X(5, 3, 8, 3, 1, 1, 2) X(6, 3, 8, 3, 1, 1, 1) \
	X(7, 4, 4, 2, 2, 2, 1) X(8, 4, 4, 2, 2, 2, 1) X(9, 4, 4, 2, 2, 1, 1) X(10, 4, 4, 2, 2, 1, 1) X(11, 4, 4, 2, 2, 1, 1) \
	X(12, 3, 4, 2, 2, 1, 1) X(13, 3, 4, 2, 2, 1, 1) X(14, 2, 4, 2, 2, 2, 1) X(15, 2, 4, 2, 2, 1, 1) X(16, 2, 4, 2, 2, 1, 1)
// ... and of the three-plane form (MISS: no lists of the missing genotypes; 4 NAF (2 NBF - 1) accumulators per wave)
#define S3_FOR_EACH_NBF_MISS(X) \
	X(2, 4, 8, 3, 1, 1, 2) X(3, 3, 8, 3, 1, 1, 2) X(4, 2, 8, 3, 1, 2, 2) X(5, 2, 8, 3, 1, 2, 2) X(6, 3, 4, 2, 2, 2, 1) \
	X(7, 2, 4, 2, 2, 2, 1) X(8, 2, 4, 2, 2, 2, 1) X(9, 2, 4, 2, 2, 2, 1) X(10, 2, 4, 2, 2, 2, 1) X(11, 1, 4, 2, 2, 2, 1) \
	X(12, 1, 4, 2, 2, 2, 1) X(13, 1, 4, 2, 2, 2, 1) X(14, 1, 4, 2, 2, 2, 1) X(15, 1, 4, 2, 2, 2, 1) X(16, 1, 4, 2, 2, 2, 1)
static inline size_t s3_lds_bytes(int NBF, int NAF, int NC, int DA, int DB) { return ((size_t)(DB + 1) * 4 * NBF + (size_t)(DA + 1) * 2 * NC * NAF) * 1024; }

#ifndef S3_KERNEL_ONLY   /* tools/score3_bench.hip takes the contraction kernel alone */
// ===========================================================================================================
// Around the contraction kernel: the offsets of the carrier lists (the lists themselves: kern_lists.h), the sparse
// T3 pass over the missing genotypes, the reduction of the item slabs and the epilogue.
#include "kern_lists.h"

// exclusive prefix over the threads of a 1024-thread workgroup (sh: 1024 values)
__device__ __forceinline__ unsigned long long s3_block_excl(unsigned long long x, unsigned long long *sh)
{
	const int tid = threadIdx.x;
	sh[tid] = x;
	__syncthreads();
	for (int o = 1; o < 1024; o <<= 1) {
		const unsigned long long y = tid >= o ? sh[tid - o] : 0;
		__syncthreads();
		sh[tid] += y;
		__syncthreads();
	}
	const unsigned long long incl = sh[tid];
	__syncthreads();
	return incl - x;
}

// ---- carrier lists of the rare variants (the input of the per-variant SPA kernels, kern_spa4.h): a variant
// with at most `lim` carriers -- non-zero codes where the alt allele is the minor one, codes other than 2 where
// the scan will flip it (make_head: AF > 0.5, from the same integer counts) -- gets its carriers listed in
// ascending sample order at cidx[cptr[v] .. cptr[v + 1]) as sample | code << 30; corient[v] = 1 / 2 says which
// orientation the list is for, 0 = no list (too many carriers, or the block's list is full: the kernels then
// scan the variant's row as they do for row-major input).
__global__ void __launch_bounds__(256)
s3_ingest_clist_count_kernel(int M, int N, int lim, int *__restrict__ nzv /* in: non-zero codes; out: listed carriers */,
	const int *__restrict__ n2v, const int *__restrict__ n3, uint8_t *__restrict__ corient)
{
	const int v = blockIdx.x * 256 + threadIdx.x;
	if (v >= M) return;
	const int nz = nzv[v], n2 = n2v[v], m3 = n3[v];
	const long long AC = (long long)(nz - n2 - m3) + 2ll * n2;
	const bool minus = (double)AC / (2.0 * (double)(N - m3)) > 0.5;      // make_head (dev_common.h)
	const int nnz = minus ? N - n2 : nz;
	const int orient = (N - m3 > 0 && nnz <= lim) ? (minus ? 2 : 1) : 0;
	corient[v] = (uint8_t)orient;
	nzv[v] = orient ? nnz : 0;
}

// one workgroup: the offsets in variant order; a variant that no longer fits the block's list goes unlisted
// (and, the sum only growing, so do the later ones that do not fit; their offsets stay valid, empty ranges)
__global__ void __launch_bounds__(1024)
s3_ingest_clist_kernel(int M, unsigned cidx_cap, const int *__restrict__ cn, unsigned *__restrict__ cptr, uint8_t *__restrict__ corient)
{
	__shared__ unsigned long long sh[1024];
	const int tid = threadIdx.x, per = (M + 1023) / 1024, v0 = tid * per, v1 = min(M, v0 + per);
	unsigned long long mine = 0;
	for (int v = v0; v < v1; v++) mine += (unsigned long long)cn[v];
	unsigned long long base = s3_block_excl(mine, sh);
	for (int v = v0; v < v1; v++) {
		const int n = cn[v];
		const bool fits = base + (unsigned long long)n <= (unsigned long long)cidx_cap;
		cptr[v] = (unsigned)min(base, (unsigned long long)cidx_cap);
		if (!fits) corient[v] = 0;
		else base += (unsigned long long)n;
	}
	if (tid == 1023) cptr[M] = (unsigned)min(base, (unsigned long long)cidx_cap);
}

// ---- T3: sums of the fixed-point score values over a variant's missing samples, per sample range.
// Q: [N][P] int64, the values the limb tiles hold (digits x position scale).  A task = (range g, variant v),
// taken by PP lanes: lane c of the task gathers column c of every listed sample (a row of Q is P x 8 contiguous
// bytes) and keeps hi = sum q >> 32, lo = sum q & 0xFFFFFFFF -- exact, whatever the order.  A workgroup works
// on one range (blockIdx % nr): with blocks dealt round-robin over the XCDs an L2 sees two of sixteen ranges of Q
// (1/8 of the table).  part: [nr][M][P][2] int64.  The lists: kern_lists.h.
template <int PP>
__global__ void __launch_bounds__(256, PP <= 16 ? 8 : 7)    /* few enough registers (64 / 72) to sit beside score3_kernel's workgroup on a CU */
s3_t3_kernel(int M, int P, const long long *__restrict__ Q, S3Lists L, long long *__restrict__ part)
{
	constexpr int TPW = 64 / PP;                       // tasks per wave
	constexpr int NB = 1;                              // index loads in flight per lane: NB x PP gathers behind them
	const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
	const int g = blockIdx.x % L.nr, chunk = blockIdx.x / L.nr;
	const unsigned *__restrict__ idx = L.idx;
	const int c = lane % PP, tk = lane / PP;
	const int v = (chunk * 4 + wid) * TPW + tk;
	unsigned e0 = 0, e1 = 0;
	if (v < M) { e0 = L.lstart[(size_t)g * L.ld + v]; e1 = e0 + (unsigned)max(0, L.lcnt[(size_t)g * L.ld + v]); }
	long long hi = 0, lo = 0;
	const int gbase = lane - c;                        // first lane of this task
	for (unsigned e = e0; __any(e < e1); e += NB * PP) {
		unsigned mine[NB];
#pragma unroll
		for (int k = 0; k < NB; k++) mine[k] = (e + k * PP + c < e1) ? idx[e + k * PP + c] : 0xFFFFFFFFu;
		constexpr int UN = PP <= 16 ? 8 : 4;           // gathers issued back to back (PP is a multiple of 8)
		constexpr int UNR = PP <= 16 ? 2 : 1;
#pragma unroll
		for (int k = 0; k < NB; k++)
#pragma unroll UNR
			for (int j0 = 0; j0 < PP; j0 += UN) {
				long long q[UN];
#pragma unroll
				for (int j = 0; j < UN; j++) {
					const unsigned s = (unsigned)__shfl((int)mine[k], gbase + j0 + j, 64);
					q[j] = (s != 0xFFFFFFFFu && c < P) ? Q[(size_t)s * P + c] : 0;
				}
#pragma unroll
				for (int j = 0; j < UN; j++) { hi += q[j] >> 32; lo += q[j] & 0xFFFFFFFFll; }
			}
	}
	if (v < M && c < P) {
		long long *o = part + (((size_t)g * M + v) * P + c) * 2;
		o[0] = hi; o[1] = lo;
	}
}

// ---- the item slabs of score3_kernel -> one row of limb sums per variant (the layout the epilogue reads:
// accbuf[v * stride + 16 b + r]).  grid = (elements of a variant tile's slab / 256, variant tiles).
// A slab is [consumer wave = (variant group, column group)][f][b < NBW][reg][lane]: NCW waves, NCB column groups
// of NBW = ceil(NBF / NCB) fragment slots each (the last group's spare slots are never written).
__global__ void __launch_bounds__(256)
s3_reduce_kernel(S3Plan pl, int M, int NCW, int NAF, int NBF, int NCB, const int *__restrict__ slabs, int *__restrict__ accbuf, int stride,
	int *__restrict__ counters, int *__restrict__ cursors)
{
	// the step's counters and queue cursors start at zero: nothing before the epilogue touches them, and two
	// memset launches at the head of every step were two more waits for a free CU beside the other lane's kernels
	if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x < 32) {
		if (threadIdx.x < 24) counters[threadIdx.x] = 0;
		if (threadIdx.x < 8 && cursors) cursors[threadIdx.x] = 0;
	}
	const int NBW = (NBF + NCB - 1) / NCB;
	// a thread takes four consecutive slab elements: the same (fragment, register, variant), four columns -- one
	// 16-byte load per item and one 16-byte store (a quarter of the load instructions of the one-element form)
	const int e = (blockIdx.x * 256 + threadIdx.x) * 4, per = NCW * NAF * NBW * 256, vtile = blockIdx.y;
	if (e >= per) return;
	const int lane = e & 63, reg = (e >> 6) & 3, fb = e >> 8;           // fb = (wave NAF + f) NBW + b
	const int bw = fb % NBW, wf = fb / NBW;                             // wf = wave NAF + f
	const int wave = wf / NAF, f = wf - wave * NAF;
	const int b = (wave % NCB) * NBW + bw;                              // fragment of the tile
	if (b >= NBF) return;
	const int v = (vtile * pl.fpw + (wave / NCB) * NAF + f) * 16 + (lane >> 4) * 4 + reg;
	if (v >= M) return;
	s3_v4i sum = (s3_v4i){0, 0, 0, 0};
	for (int g = 0; g < pl.ng; g++) {
		int first, count;
		s3_items_of(pl, vtile, g, first, count);
		for (int id = first; id < first + count; id++) sum += *reinterpret_cast<const s3_v4i *>(slabs + (size_t)id * per + e);
	}
	*reinterpret_cast<s3_v4i *>(accbuf + (size_t)v * stride + b * 16 + (lane & 15)) = sum;      // (stride is a multiple of 16 ints)
}

// ---- epilogue: the limb sums of a variant -> its row of the table (or its place in the SPA stage's lists).
// A workgroup takes s3e_vb(K) variants in three steps:
//  0. a thread per variant (as many threads as variants): missing genotypes (the per-range counts of the list pass / the block, or the missing plane's
//     constant column), allele counts, filters (make_head).  Variants whose missing genotypes are not listed go onto
//     `ovf_list` (counters[23]) for the FP64 kernel.
//  1. a thread per (variant, score column): integer recombination of the column's limb sums with the sums over the missing
//     samples -- the per-range partials of the sparse pass are added here (no kernel of their own), or the missing plane's
//     limbs -- and the constant column worth 4 per allele (the position scales want a multiple of 4).  Consecutive lanes
//     read consecutive addresses, every load of a thread is in flight before the first is used.  (One thread per variant
//     doing the columns one after the other waited for each of its 50 - 460 loads in turn: 60 us at K = 3, 0.65 ms at K = 13.)
//  2. a thread per variant again: the guard on the fixed-point columns, the score test, the SPA stage's record.
// (The per-variant steps keep every wave of the workgroup busy: with 64 variants per 256 threads step 2 ran on the first
// wave of every workgroup only -- on one SIMD of each CU -- and took 0.59 ms instead of 0.03.)
__host__ __device__ constexpr int s3e_vb(int K) { return K <= 8 ? 256 : 128; }      // variants = threads per workgroup
template <int K>
__global__ void __launch_bounds__(256)
score3_epilogue(int M, DevModel md, MfEpi ep, const int *__restrict__ accbuf, int acc_stride, int miss_off,
	const long long *__restrict__ t3part /* [nr][M][P][2] per-range sums over the missing samples */, int nr,
	const int *__restrict__ lcnt /* [nr][ld] listed missing genotypes per range, -1: not listed */, size_t ld, int *__restrict__ ovf_list,
	SpaRec *__restrict__ recs, int *__restrict__ counters, int btop, int *__restrict__ fb_series, int *__restrict__ fb_exact,
	double *__restrict__ out8, uint8_t *__restrict__ valid, double guard_tol, int dbg = 0)
{
	constexpr int P = 2 * K + 2, CW = P - 1;     // score columns c' (K), e (K), s, w; column CW carries G^2
	constexpr int S3E_VB = s3e_vb(K);
	__shared__ double s_acc[S3E_VB][P | 1];
	__shared__ double s_imp[S3E_VB];
	__shared__ int s_n[S3E_VB][4];                // n1, n2, n3, state (0: done with, 1: minor allele alt, 2: flipped)
	__shared__ long long s_ftot[P][2];
	__shared__ int s_col[P][3];                   // first limb column, limbs, scale exponent
	const int tid = threadIdx.x, j0 = blockIdx.x * S3E_VB;
	const int N = md.N;
	if (dbg & 1) {
		if (tid == 0)
#pragma unroll
			for (int c = 0; c < P; c++) {
				s_col[c][0] = ep.ccol[c]; s_col[c][1] = ep.climb[c]; s_col[c][2] = ep.escale[c];
				s_ftot[c][0] = ep.ftot_hi[c]; s_ftot[c][1] = ep.ftot_lo[c];
			}
	} else
	for (int c = tid; c < P; c += S3E_VB) {
		s_col[c][0] = ep.ccol[c]; s_col[c][1] = ep.climb[c]; s_col[c][2] = ep.escale[c];
		s_ftot[c][0] = ep.ftot_hi[c]; s_ftot[c][1] = ep.ftot_lo[c];
	}
	// ---- 0. per variant
	{
		const int j = j0 + tid;
		const int *a0 = accbuf + (size_t)min(j, M - 1) * acc_stride;
		int n3 = 0;
		bool over = false;
		if (miss_off) n3 = a0[miss_off + ep.col_ones] / 4;
		else {
			for (int g0 = 0; g0 < nr; g0 += 8) {
				int cn[8];
#pragma unroll
				for (int u = 0; u < 8; u++) cn[u] = lcnt[(size_t)min(g0 + u, nr - 1) * ld + min(j, M - 1)];
#pragma unroll
				for (int u = 0; u < 8; u++) if (g0 + u < nr) { over |= cn[u] < 0; n3 += cn[u] < 0 ? 0 : cn[u]; }
			}
		}
		if (j >= M) n3 = 0;
		{
			// census of the step's missing genotypes (units of 64) for the host's choice between the two forms
			const int tot = wave_total_i(n3);
			if ((tid & 63) == 0 && tot >= 64) atomicAdd(&counters[22], tot >> 6);
		}
		int state = 0, n1 = 0, n2 = 0;
		double imp = 0;
		if (dbg & 8) over = false;
		if (j < M) {
			if (over) ovf_list[atomicAdd(&counters[23], 1)] = j;
			else {
				const long long AC = (long long)(a0[ep.col_ones] / 4) - 3ll * n3;
				n2 = a0[ep.col_b1 + ep.climb[CW]] / 8 - n3;    // bit-1 plane (0/2) against the constant column (4)
				n1 = (int)(AC - 2ll * n2);
				const VarHead h = make_head(md, (double)AC, N - n3);
				if (!h.pass) { nan_row(out8 + (size_t)j * 8); valid[j] = 0; }
				else { state = h.minus ? 2 : 1; imp = 2 * h.AF; }
			}
		}
		s_n[tid][0] = n1; s_n[tid][1] = n2; s_n[tid][2] = n3; s_n[tid][3] = state;
		s_imp[tid] = imp;
	}
	__syncthreads();
	// ---- 1. per (variant, column)
	for (int task = tid; task < S3E_VB * P; task += S3E_VB) {
		const int vl = task / P, c = task - vl * P, j = j0 + vl;
		const int state = s_n[vl][3];
		if (!state) continue;
		const int cc = s_col[c][0], nl = s_col[c][1];
		if (nl == 0) { s_acc[vl][c] = 0; continue; }           // derived below
		const int *a0 = accbuf + (size_t)j * acc_stride;
		const bool minus = state == 2;
		const double imp = s_imp[vl];
		// every load first: the column's limbs (those beyond nl read the last one again), the missing samples' sums
		int lv[MF_NLIMB], lm[MF_NLIMB], lb[MF_NLIMB];
#pragma unroll
		for (int l = 0; l < MF_NLIMB; l++) lv[l] = a0[cc + min(l, nl - 1)];
		HiLo T3 = hl(0, 0);
		if (miss_off) {
#pragma unroll
			for (int l = 0; l < MF_NLIMB; l++) lm[l] = a0[miss_off + cc + min(l, nl - 1)];
		} else if (!(dbg & 2)) {
			const long long *p = t3part + ((size_t)j * P + c) * 2;
			const size_t gs = (size_t)M * P * 2;
			for (int g0 = 0; g0 < nr; g0 += 8) {
				long long ph[8], pl[8];
#pragma unroll
				for (int u = 0; u < 8; u++) { const long long *q = p + (size_t)min(g0 + u, nr - 1) * gs; ph[u] = q[0]; pl[u] = q[1]; }
#pragma unroll
				for (int u = 0; u < 8; u++) if (g0 + u < nr) { T3.hi += ph[u]; T3.lo += pl[u]; }
			}
		}
		if (c == CW) {
#pragma unroll
			for (int l = 0; l < MF_NLIMB; l++) lb[l] = a0[ep.col_b1 + min(l, nl - 1)];
		}
		auto limbs = [&](const int (&x)[MF_NLIMB]) -> HiLo {
			long long lo = 0, hi = 0;
#pragma unroll
			for (int l = 3; l >= 0; l--) lo = lo * 256 + (l < nl ? x[l] : 0);
#pragma unroll
			for (int l = MF_NLIMB - 1; l >= 4; l--) hi = hi * 256 + (l < nl ? x[l] : 0);
			return hl(hi, lo);
		};
		const HiLo V = limbs(lv);
		if (miss_off) T3 = limbs(lm);
		const HiLo W = hl_axpy(-3, T3, V);
		const double t3d = hl_to_double(T3);
		double r;
		if (c != CW) {
			double sm;
			if (!minus) sm = hl_to_double(W) + imp * t3d;
			else sm = hl_to_double(hl(2 * s_ftot[c][0] - W.hi, 2 * s_ftot[c][1] - W.lo)) - imp * t3d;
			r = ldexp(sm, -s_col[c][2]);
		} else {
			const HiLo B2 = limbs(lb);                           // = 2 (T2 + T3), the plane holds 0/2
			const HiLo H2 = hl(B2.hi / 2 - T3.hi, B2.lo / 2 - T3.lo);   // every limb sum of that plane is even
			double w;
			if (!minus) {
				w = hl_to_double(hl_axpy(2, H2, W)) + imp * imp * t3d;
			} else {
				const HiLo S1 = hl_axpy(-2, H2, W);
				const HiLo R = hl(s_ftot[CW][0] - S1.hi - H2.hi - T3.hi, s_ftot[CW][1] - S1.lo - H2.lo - T3.lo);
				w = hl_to_double(hl_axpy(4, R, S1)) + (2 - imp) * (2 - imp) * t3d;
			}
			r = ldexp(w, -s_col[CW][2]);
		}
		s_acc[vl][c] = r;
	}
	__syncthreads();
	// ---- 2. per variant
	if (dbg & 4) return;
	const int j = j0 + tid;
	if (j >= M || !s_n[tid][3]) return;
	const int n1 = s_n[tid][0], n2 = s_n[tid][1], n3 = s_n[tid][2];
	const long long AC = (long long)n1 + 2ll * n2;
	const VarHead h = make_head(md, (double)AC, N - n3);
	const double imp = 2 * h.AF;
	double *o = out8 + (size_t)j * 8;
	double acc[P];
#pragma unroll
	for (int c = 0; c < P; c++) acc[c] = s_acc[tid][c];
	if (ep.derive_c) {
#pragma unroll
		for (int x = 0; x < K; x++) {
			double cx = 0;
#pragma unroll
			for (int y = 0; y < K; y++) cx = fma(ep.XVXi[x * K + y], acc[K + y], cx);
			acc[x] = cx;
		}
	}
	// ---- a-posteriori bound on what the columns' quantisation can have done to this variant's z-score
	// (VERDICT r03, weak 2).  An entry of column c is off by up to ~1 unit 2^-escale[c] (odd sample positions carry
	// multiples of 4 units), a sum over the carriers by ~sqrt(sum G^2) units; through  var2 = c'XVXc' + w - 2 e.c'  and
	// S = s - S_a.c'  (first order) that moves  z = S / sqrt(r var2)  by dz.  Ill-conditioned designs -- covariate
	// projections that cancel across columns -- show up as a large |XVX c' - e| or |S_a| against what var2 is made of.  Beyond
	// guard_tol the variant is scored by the FP64 kernel from the unquantised vectors instead.
	{
		const double g2 = (double)n1 + 4.0 * n2 + (h.minus ? (2 - imp) * (2 - imp) : imp * imp) * n3;
		const double spread = 8.0 * sqrt(fmax(g2, 1.0));             // (8: safety over the root-sum-square)
		double dcol[P];
#pragma unroll
		for (int c = 0; c < P; c++) dcol[c] = (ep.climb[c] && !(md.quant && c == CW)) ? spread * ldexp(1.0, -ep.escale[c]) : 0.0;   // (quantitative w = 1: exact)
		// (quantitative traits with derived c' = XVXi e: c' is a function of e, var2 = w - e'XVXi e and S = s - (XVXi'S_a).e,
		// so d var2 = -2 c'.de + dw and dS = ds - (XVXi'S_a).de: no term in dc' of its own)
		double sat[K];
#pragma unroll
		for (int y = 0; y < K; y++) {
			double t = 0;
#pragma unroll
			for (int x = 0; x < K; x++) t = fma(md.S_a[x], ep.XVXi[x * K + y], t);
			sat[y] = t;
		}
		// d var2 = 2 (XVX c' - e).dc' - 2 c'.de + dw  (first order; XVX c' - e vanishes where the two weight vectors agree)
		double dquad = 0, dec = 0, dS = dcol[2 * K], quad = 0, ec = 0, sac = 0;
#pragma unroll
		for (int a = 0; a < K; a++) {
			double t = 0;
#pragma unroll
			for (int b = 0; b < K; b++) t = fma(md.XVX[a * K + b], acc[b], t);
			quad = fma(t, acc[a], quad);
			dquad += 2 * dcol[a] * fabs(t - acc[K + a]);
			ec = fma(acc[K + a], acc[a], ec);
			dec += dcol[K + a] * fabs(acc[a]);
			sac = fma(md.S_a[a], acc[a], sac);
			dS += ep.derive_c ? fabs(sat[a]) * dcol[K + a] : fabs(md.S_a[a]) * dcol[a];
		}
		const double var2 = quad + acc[CW] - 2 * ec, S = acc[2 * K] - sac;
		const double dvar = dquad + 2 * dec + dcol[CW];
		const double sd = sqrt(md.r * fmax(var2, 0.0) * (md.quant ? 1.0 / fmax(h.mac, 1.0) : 1.0)) * (md.quant ? md.tau0 * sqrt(fmax(h.mac, 1.0)) : 1.0);
		const double z = fabs(S) / sd;
		const double dz = dS / sd + 0.5 * z * dvar / var2;
		if (!(dz * fmax(1.0, z) <= guard_tol)) {                        // (also: var2 <= 0 or not finite)
			ovf_list[atomicAdd(&counters[23], 1)] = j;
			atomicAdd(&counters[21], 1);
			return;
		}
	}
	double cbuf[KMAX], pn, Ssc, v2sc;
	valid[j] = 1;
	if (score_epilogue<(P - 2) / 2>(md, h, acc, o, cbuf, &pn, &Ssc, &v2sc)) {
		spa_push<K>(md, recs, counters, btop, fb_series, fb_exact, j, h.minus, h.minus ? (2 * h.Num - h.AC) : h.AC,
			h.minus ? (N - n2) : (n1 + n2 + n3), h.lut, pn, Ssc, v2sc, cbuf);
	}
	{
		// one add per wave (50 000 adds of single lanes on this one address took 0.59 ms: ~12 ns each)
		const unsigned long long act = __ballot(true);
		if (__builtin_amdgcn_mbcnt_hi((uint32_t)(act >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)act, 0u)) == 0u) atomicAdd(&counters[1], (int)__popcll(act));
	}
}
#endif /* S3_KERNEL_ONLY */
