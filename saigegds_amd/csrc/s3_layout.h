// s3_layout.h -- the tiled device layout of genotype blocks and the work decomposition of score3_kernel
// (host and device; no HIP specifics).  Part of libsaigehip.so.
#pragma once
#include <stddef.h>
#include <stdint.h>

#ifndef __host__
#define __host__
#define __device__
#endif
#ifndef __forceinline__
#define __forceinline__ inline
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
	// row-major input (score3_kernel<.., RM = true>): the rows as the caller holds them, no tiles
	int nrow;       // rows (variants) that exist: the loaders never touch a row >= nrow
	unsigned long long bpv;   // bytes per row (a multiple of 16, >= 64 ntile)
};

static inline S3Plan s3_plan(size_t M, int ntile, int grid, int fpw, size_t bpv = 0)
{
	S3Plan p{};
	p.ntile = ntile;
	p.nrow = (int)M;
	p.bpv = bpv;
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

// byte offset of the 16-B piece p (64 samples) of variant j in the tiled layout: inside the KiB of (fragment,
// tile) the 64 bytes of a variant are contiguous, so whoever reads ONE row (the SPA kernels) touches whole
// 64-byte sectors; the contraction kernel's loader permutes the lanes of its DMA instead (s3_dma_lane)
__host__ __device__ __forceinline__ size_t s3_piece_off(size_t j, size_t p, int ntile)
{
	return ((j >> 4) * (size_t)ntile + (p >> 2)) * 1024 + (((j & 15) << 2) + (p & 3)) * 16;
}
// 16-byte slot of the KiB that the consumer lane (r = lane & 15: variant, kg = lane >> 4: 64-sample piece) wants
__host__ __device__ __forceinline__ int s3_dma_lane(int lane) { return ((lane & 15) << 2) | (lane >> 4); }

// Sample order inside a group of 16 (as kern_score_mfma.h mf_pos): byte j of (w >> 2t) & 0x03030303 is
// the code of sample 4 j + t, and the B tiles store the 16 samples of a group in that order.
__host__ __device__ __forceinline__ int s3_pos(int s) { return ((s & 3) << 2) | (s >> 2); }

// scale of the code at position e (0..15) of a dword in the A operands of score3_kernel (s3_unpack_op): the
// codes at odd positions are used where they stand, two bits up -- their limb digits carry q / 4
__host__ __device__ __forceinline__ int s3_scale(int e) { return (e & 1) ? 4 : 1; }

// sample ranges of the missing-genotype lists (kern_lists.h, s3_t3_kernel): range g of nr = tiles [g ntile / nr,
// (g + 1) ntile / nr).  nr = 16 for long rows (a workgroup of the T3 pass gathers from 1/16 of the table Q, so an
// XCD's L2 sees 1/8 of it), fewer for short ones (a wave of the list builder wants ~8 KiB of a row: at
// N = 50 000 sixteen ranges were 800 000 waves of 780 bytes, bound by the launch rate).
#define S3_NR 16
__host__ __device__ __forceinline__ int s3_nranges(int ntile) { const int n = (ntile * 64 + 8191) / 8192; return n < 1 ? 1 : (n > S3_NR ? S3_NR : n); }
__host__ __device__ __forceinline__ int s3_range_t0(int g, int ntile, int nr) { return (int)((long long)g * ntile / nr); }
