// kern_synth.h -- synthetic 2-bit genotype generator
// Part of libsaigehip.so (single translation unit: saigehip.hip).
#pragma once

// ---------------------------------------------------------------------------
// Synthetic 2-bit genotypes (bench / tests): see saigehip.h, sgx_synth_2bit_dev

__host__ __device__ __forceinline__ uint64_t splitmix64(uint64_t x)
{
	x += 0x9E3779B97F4A7C15ull;
	x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
	x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
	return x ^ (x >> 31);
}

__global__ void __launch_bounds__(256)
synth2b_kernel(uint8_t *__restrict__ packed, size_t bpv, int N, size_t M, uint64_t first,
	uint64_t seed, const uint32_t *__restrict__ thr)
{
	const size_t j = blockIdx.y;
	if (j >= M) return;
	const uint32_t t0 = thr[3 * j], t1 = thr[3 * j + 1], tm = thr[3 * j + 2];
	const uint64_t key = splitmix64(seed ^ splitmix64(first + j));
	uint32_t *row = reinterpret_cast<uint32_t *>(packed + j * bpv);
	const int nd = (int)(bpv / 4);
	for (int d = blockIdx.x * blockDim.x + threadIdx.x; d < nd; d += gridDim.x * blockDim.x) {
		uint32_t w = 0;
#pragma unroll
		for (int s = 0; s < 16; s++) {
			const int i = d * 16 + s;
			if (i < N) {
				const uint64_t x = splitmix64(key + (uint64_t)i);
				const uint32_t u = (uint32_t)(x >> 32), m = (uint32_t)x;
				uint32_t code = (u < t0) ? 0u : ((u < t1) ? 1u : 2u);
				if (m < tm) code = 3u;
				w |= code << (2 * s);
			}
		}
		row[d] = w;
	}
}
