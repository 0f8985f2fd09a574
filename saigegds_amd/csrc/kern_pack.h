// kern_pack.h -- ingest: dosage rows as they arrive over PCIe -> 2-bit rows for the MFMA path
// Part of libsaigehip.so (single translation unit: saigehip.hip).
#pragma once

// What seqApply(.useraw = NA) hands the reference is RAW 0/1/2/0xFF for hard calls (get_ds,
// saige_main.cpp:179-182) or INTEGER with NA_INTEGER (:175-178).  Both are packed on the device to
// the 2-bit rows of sgx_scan_2bit (code = dosage, 3 = missing); a value outside {0, 1, 2, missing}
// raises `flag`, and the block then takes the dosage kernels unpacked.
// One thread per 16 samples = one dword of a packed row; grid.y = row.
template <typename T>
__global__ void __launch_bounds__(256)
pack_rows_2bit(const T *__restrict__ rows, int N, uint8_t *__restrict__ packed, size_t bpv, int *__restrict__ flag)
{
	const size_t j = blockIdx.y;
	const int nd = (int)(bpv / 4);
	const T *row = rows + j * (size_t)N;
	uint32_t *out = reinterpret_cast<uint32_t *>(packed + j * bpv);
	bool bad = false;
	for (int d = blockIdx.x * blockDim.x + threadIdx.x; d < nd; d += gridDim.x * blockDim.x) {
		uint32_t w = 0;
#pragma unroll
		for (int s = 0; s < 16; s++) {
			const int i = d * 16 + s;
			if (i < N) {
				const T v = row[i];
				uint32_t code;
				if (sizeof(T) == 1) code = ((uint8_t)v == 0xFF) ? 3u : (uint32_t)(uint8_t)v;
				else code = ((int)v == (int)0x80000000) ? 3u : (uint32_t)(int)v;      // NA_INTEGER
				const bool miss = sizeof(T) == 1 ? ((uint8_t)v == 0xFF) : ((int)v == (int)0x80000000);
				if (!miss && code > 2u) { bad = true; code = 0; }
				w |= code << (2 * s);
			}
		}
		out[d] = w;
	}
	if (bad) atomicOr(flag, 1);
}

// INTEGER dosages that are not all 0/1/2/NA: to doubles (NaN = missing) for the dosage kernels
__global__ void __launch_bounds__(256)
i32_rows_to_f64(const int *__restrict__ rows, size_t n, double *__restrict__ out)
{
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
		const int v = rows[i];
		out[i] = (v == (int)0x80000000) ? NAN : (double)v;
	}
}
