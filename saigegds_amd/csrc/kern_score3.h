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

// ---- work decomposition (host-computed, by value) ------------------------------------------------
// The grid has `grid` workgroups (a multiple of 8).  Workgroups with equal (blockIdx % 8) % ng share a
// TILE GROUP g: a contiguous range of 256-sample tiles [g ntile / ng, (g + 1) ntile / ng), so that an
// XCD (blocks are dealt round-robin over the 8 XCDs: a placement guess, never correctness) streams the
// same B tiles from its L2.  Inside a group the wpg workgroups take the variant tiles round-robin: rf
// full rounds, then the rem leftover variant tiles cut into f tile sub-ranges each so that the last
// round is spread over (nearly) all workgroups.  Every item writes its own slab of partial sums.
struct S3Plan {
	int ntile;      // 256-sample tiles of a row
	int nfrag;      // 16-variant fragments = ceil(M / 16)
	int fpw;        // fragments per workgroup = NAF * WAVES of the instantiation
	int vt;         // variant tiles = ceil(nfrag / fpw)
	int ng;         // tile groups: 8, 4, 2 or 1
	int wpg;        // workgroups per group = grid / ng
	int rf;         // full rounds = vt / wpg
	int rem;        // leftover variant tiles = vt % wpg
	int f;          // pieces per leftover variant tile (0 if rem == 0)
	int ipg;        // items per group = rf * wpg + rem * f
};

static inline S3Plan s3_plan(size_t M, int ntile, int grid, int fpw)
{
	S3Plan p{};
	p.ntile = ntile;
	p.nfrag = (int)((M + 15) / 16);
	p.fpw = fpw;
	p.vt = (p.nfrag + fpw - 1) / fpw;
	p.ng = 8;
	while (p.ng > 1 && ntile / p.ng < 8) p.ng >>= 1;
	p.wpg = grid / p.ng;
	p.rf = p.vt / p.wpg;
	p.rem = p.vt % p.wpg;
	p.f = 0;
	if (p.rem) {
		const int bylen = (ntile / p.ng) / 4 > 0 ? (ntile / p.ng) / 4 : 1;    // a piece is at least ~4 tiles
		p.f = p.wpg / p.rem < bylen ? p.wpg / p.rem : bylen;
		if (p.f < 1) p.f = 1;
	}
	p.ipg = p.rf * p.wpg + p.rem * p.f;
	return p;
}

// items of variant tile `vtile` in group g: ids [first, first + count)
__host__ __device__ __forceinline__ void s3_items_of(const S3Plan &p, int vtile, int g, int &first, int &count)
{
	if (vtile < p.rf * p.wpg) { first = g * p.ipg + vtile; count = 1; }
	else { first = g * p.ipg + p.rf * p.wpg + (vtile - p.rf * p.wpg) * p.f; count = p.f; }
}

// bytes of one tiled block of M variants: nfrag fragments x ntile KiB
static inline size_t s3_block_bytes(size_t M, int ntile) { return ((M + 15) / 16) * (size_t)ntile * 1024; }

// byte offset of the 16-B piece p (64 samples) of variant j in the tiled layout
__host__ __device__ __forceinline__ size_t s3_piece_off(size_t j, size_t p, int ntile)
{
	return ((j >> 4) * (size_t)ntile + (p >> 2)) * 1024 + (((p & 3) << 4) + (j & 15)) * 16;
}

// Sample order inside a group of 16 (as kern_score_mfma.h mf_pos): byte j of (w >> 2t) & 0x03030303 is
// the code of sample 4 j + t, and the B tiles store the 16 samples of a group in that order.
__host__ __device__ __forceinline__ int s3_pos(int s) { return ((s & 3) << 2) | (s >> 2); }

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
#define S3_NOPS 9
// scale of the code at position e (0..15) of a dword: byte e / 4 ... no: code e sits in bits 2e, 2e + 1, i.e.
// byte e >> 2, position e & 3 of that byte
__host__ __device__ __forceinline__ int s3_scale(int e) { return (e & 1) ? 4 : 1; }
// byte of the A operand (and of a B tile's 16-byte group) that holds code e of a dword: operand dword t holds,
// in byte j, the code of byte j of the packed dword at position t -> code e = 4 j + t sits at byte 4 t + j
// (the same order as kern_score_mfma.h mf_pos)

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
template <int NBF, int NAF, int NC, int NLA, int NLB, int DA, int DB, int ABL = 0>
__global__ void __launch_bounds__(64 * (NC + NLA + NLB), (NC + NLA + NLB + 3) / 4)
score3_kernel(const uint8_t *__restrict__ A, const uint8_t *__restrict__ Fl, S3Plan pl, int *__restrict__ out, unsigned long long *__restrict__ stamps)
{
#if __HIP_DEVICE_COMPILE__      /* (the host pass only needs the stub; it does not know the buffer-resource builtins) */
	constexpr int NCOL = 16 * NBF;
	constexpr int TILE_BYTES = 16 * NCOL * 16;
	constexpr int NPB = TILE_BYTES / 1024;                    // KiB pieces of a B tile = 4 NBF
	constexpr int NPA = NC * NAF;                             // KiB pieces of a tile's rows
	constexpr int RA = DA + 1, RB = DB + 1;                   // ring slots
	constexpr int SLOT_A = NPA * 1024;
	static_assert(DA >= 1 && DB >= 1, "at least one tile ahead");
	extern __shared__ __attribute__((aligned(16))) uint8_t s3_smem[];   // RB x TILE_BYTES (B), then RA x SLOT_A (rows)

	const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
	const int x = blockIdx.x & 7, y = blockIdx.x >> 3;
	const int g = x % pl.ng, i = y * (8 / pl.ng) + x / pl.ng;
	const int nk = pl.rf + (i < pl.rem * pl.f ? 1 : 0);     // items of this workgroup
	const int T0 = (int)((long long)g * pl.ntile / pl.ng), T1 = (int)((long long)(g + 1) * pl.ntile / pl.ng);
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
			p.t = T0 + q * len / pl.f; p.t1 = T0 + (q + 1) * len / pl.f;
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
		const int voff = lane * 16;
		Pos pa = pc;
		int sl = 0;                                           // slot of the next tile to issue
		int ahead = 0;                                        // tiles issued beyond the one the next barrier releases
		if (rows) {
			constexpr int PLO = NPA / NLA, NHI = NPA % NLA;       // loaders l < NHI take PLO + 1 pieces
			static_assert((PLO + 1) * (DA - 1) < 64, "vmcnt range");
			auto issue = [&]() {
				// rows of the item's variant tile: a descriptor at its first fragment, the fragment and the tile
				// in the scalar offset (fpw fragments x ntile KiB: far below 4 GiB)
				const size_t f0 = (size_t)pa.vtile * pl.fpw;
				const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void *)(A + f0 * pl.ntile * 1024), 0, 0xFFFFFFFFu, 0x00020000);
				const int lastf = pl.nfrag - 1 - (int)f0;      // past the end: the last fragment again (never stored)
#pragma unroll
				for (int j = 0; j <= PLO; j++) {
					const int p = l + j * NLA;
					if (p >= NPA || (ABL & 2)) break;
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
	// LDS byte addresses of this lane's 16 B of a row piece and of its B fragment (sample group 4 kg, column r)
	const uint32_t smem_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t *)s3_smem;
	const uint32_t ring_lds = smem_lds + RB * TILE_BYTES + wid * (NAF * 1024) + lane * 16;
	const uint32_t bt_lds = smem_lds + (4 * kg * NCOL + r) * 16;

	s3_v4i acc[NAF][NBF];
#pragma unroll
	for (int f = 0; f < NAF; f++)
#pragma unroll
		for (int b = 0; b < NBF; b++) acc[f][b] = (s3_v4i){0, 0, 0, 0};

	// B fragments are read one chunk (<= BCH fragments) ahead of the MFMAs that use them; per dword step u the
	// NAF NBF MFMAs run fragment-major (a B fragment feeds NAF consecutive MFMAs) and the 9 NAF unpack
	// operations -- the b1 planes of this step, then the value planes of the next -- are dealt out evenly
	// behind them, each group fenced so that the compiler keeps the order.
	constexpr int BCH = NBF <= 4 ? NBF : (NBF >= 9 ? 2 : (NBF % 3 == 0 ? 3 : 4));
	// with many B fragments the unpack is a few per cent of a step: one set of value planes, made at the
	// step's start, leaves the registers to the accumulators
	constexpr bool PIPE = NBF <= 8;
	constexpr int NCH = (NBF + BCH - 1) / BCH;              // chunks per dword step
	constexpr int NM = NAF * NBF;                           // MFMAs per dword step
	constexpr int NV = 4 * NAF, NW = 5 * NAF;               // b1 operations of a step, value operations of the next

	int sa = 0, sb = 0;        // ring slots of the current tile (rows, B)
	while (pc.k < nk) {
		__builtin_amdgcn_sched_barrier(0);
		__builtin_amdgcn_s_barrier();
		__builtin_amdgcn_sched_barrier(0);
		// (LDS reads by inline asm with counted lgkmcnt waits: behind an LDS-DMA the compiler puts vmcnt(0) in
		// front of every LDS read it can see)
		const uint32_t a_addr = ring_lds + (uint32_t)(sa * SLOT_A);
		const uint32_t b_addr = bt_lds + (uint32_t)(sb * TILE_BYTES);
		s3_v4i aw[NAF];
		s3_static_for<0, NAF>([&](auto F) { constexpr int f = decltype(F)::value; S3_DS_READ(aw[f], a_addr, f * 1024); });
		s3_v4i bf[2][BCH];
		auto read_chunk = [&](auto CI) {
			constexpr int ci = decltype(CI)::value, u = ci / NCH, b0 = (ci % NCH) * BCH;
			s3_static_for<0, BCH>([&](auto J) {
				constexpr int j = decltype(J)::value;
				if constexpr (b0 + j < NBF) {
					if (ABL & 8) bf[ci & 1][j] = (s3_v4i){u, j, r, sa};
					else S3_DS_READ(bf[ci & 1][j], b_addr, u * NCOL * 16 + (b0 + j) * 256);
				}
			});
		};
		constexpr int nread_last = NBF - (NCH - 1) * BCH;       // fragments of a dword step's last chunk
		read_chunk(std::integral_constant<int, 0>());
		// the row pieces are back (the first B chunk may still be in flight): value planes of dword 0
		if (!(ABL & 8)) { S3_LGKM_WAIT(BCH < NBF ? BCH : NBF, aw[0]); } else { S3_LGKM_WAIT(0, aw[0]); }
#pragma unroll
		for (int f = 1; f < NAF; f++) S3_TIE(aw[f]);
		s3_v4i val[PIPE ? 2 : 1][NAF], b1[NAF];
		uint32_t w4[PIPE ? 2 : 1][NAF];       // the dword shifted by 4: of the current dword (its b1 planes) and of the next (its value planes)
		if (!(ABL & 1)) {
			s3_static_for<0, NW>([&](auto O) {
				constexpr int o = decltype(O)::value, f = o / 5, op = o % 5;
				s3_unpack_op<op>((uint32_t)aw[f][0], w4[0][f], val[0][f], b1[f]);
			});
		}
		__builtin_amdgcn_sched_barrier(0);
		s3_static_for<0, 4>([&](auto U) {
			constexpr int u = decltype(U)::value;
			constexpr int vb = PIPE ? (u & 1) : 0, vn = PIPE ? ((u + 1) & 1) : 0;      // value-plane sets of this dword / the next
			if constexpr (!PIPE && u > 0) {
				if (!(ABL & 1)) {
					s3_static_for<0, NW>([&](auto O) {
						constexpr int o = decltype(O)::value, f = o / 5, op = o % 5;
						s3_unpack_op<op>((uint32_t)aw[f][u], w4[0][f], val[0][f], b1[f]);
					});
				}
				__builtin_amdgcn_sched_barrier(0);
			}
			s3_static_for<0, NM>([&](auto MI) {
				constexpr int m = decltype(MI)::value, b = m / NAF, f = m % NAF, ch = b / BCH, ci = u * NCH + ch, j = b % BCH;
				if constexpr (f == 0 && j == 0) {
					// entering a chunk: start the next one, then wait for this one
					if constexpr (ci + 1 < 4 * NCH) read_chunk(std::integral_constant<int, ci + 1>());
					constexpr int inflight = (ci + 1 < 4 * NCH && !(ABL & 8)) ? ((ci + 1) % NCH == NCH - 1 ? nread_last : BCH) : 0;
					constexpr int nb = (ch == NCH - 1) ? nread_last : BCH;
					S3_LGKM_WAIT(inflight, bf[ci & 1][0]);
#pragma unroll
					for (int jj = 1; jj < nb; jj++) S3_TIE(bf[ci & 1][jj]);
				}
				if (ABL & 1) { if (f == 0) acc[0][b][1] ^= bf[ci & 1][j][0] ^ aw[b % NAF][u]; }
				else if (b == NBF - 1) acc[f][b] = __builtin_amdgcn_mfma_i32_16x16x64_i8(b1[f], bf[ci & 1][j], acc[f][b], 0, 0, 0);
				else acc[f][b] = __builtin_amdgcn_mfma_i32_16x16x64_i8(val[vb][f], bf[ci & 1][j], acc[f][b], 0, 0, 0);
				if (!(ABL & 1)) {
					// b1 planes of this dword: all dealt before the first MFMA of the b1 fragment
					constexpr int MB = (NBF - 1) * NAF;
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
				}
				__builtin_amdgcn_sched_barrier(0);
			});
		});
		if (pc.t + 1 == pc.t1) {
			// the item is complete: its slab [wave][f][b][reg][lane] leaves by plain stores (256 B per instruction)
			int *dst = out + ((size_t)pc.id * NC + wid) * (NAF * NBF * 256) + lane;
#pragma unroll
			for (int f = 0; f < NAF; f++)
#pragma unroll
				for (int b = 0; b < NBF; b++)
#pragma unroll
					for (int reg = 0; reg < 4; reg++) {
						dst[((f * NBF + b) * 4 + reg) * 64] = acc[f][b][reg];
						acc[f][b][reg] = 0;
					}
		}
		pos_next(pc);
		sa = sa + 1 == RA ? 0 : sa + 1;
		sb = sb + 1 == RB ? 0 : sb + 1;
	}
	if ((ABL & 16) && tid == 0) {
		stamps[2 * blockIdx.x] = __builtin_amdgcn_s_memtime() - st0;
		stamps[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - sr0;
	}
#endif
}
