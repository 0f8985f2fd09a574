// tools/fp64_rate.hip -- issue rate of FP64 vector instructions on one SIMD (cycles per wave64 instruction)
//   hipcc -O3 --offload-arch=gfx950 -o tools/fp64_rate tools/fp64_rate.hip && ./tools/fp64_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

template <int OP>
__global__ void __launch_bounds__(256) rate_kernel(double *out, unsigned long long *cyc, int iters, double seed)
{
	double a[8];
	for (int i = 0; i < 8; i++) a[i] = seed + threadIdx.x * 1e-3 + i;
	const double m = 1.0000001, c = 1e-9;
	unsigned long long t0 = __builtin_readcyclecounter();
	for (int it = 0; it < iters; it++) {
#pragma unroll
		for (int r = 0; r < 4; r++) {
#pragma unroll
			for (int i = 0; i < 8; i++) {
				if (OP == 0) a[i] = fma(a[i], m, c);
				else if (OP == 1) a[i] = a[i] * m;
				else if (OP == 2) a[i] = a[i] + c;
				else if (OP == 3) { float f = (float)a[i]; f = fmaf(f, 1.0000001f, 1e-9f); a[i] = f; }   // (f32 reference incl. cvt: not clean)
				else if (OP == 4) a[i] = fmax(a[i], c) ;
			}
		}
	}
	unsigned long long t1 = __builtin_readcyclecounter();
	double s = 0;
	for (int i = 0; i < 8; i++) s += a[i];
	out[blockIdx.x * blockDim.x + threadIdx.x] = s;
	if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int OP>
static void run(const char *name, int waves_per_simd)
{
	double *out; unsigned long long *cyc;
	hipMalloc((void **)&out, 1 << 24); hipMalloc((void **)&cyc, 1 << 16);
	hipDeviceProp_t pr; hipGetDeviceProperties(&pr, 0);
	const int iters = 20000, nblk = pr.multiProcessorCount * waves_per_simd;   // 256 threads = 4 waves = 1 per SIMD
	hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
	rate_kernel<OP><<<nblk, 256>>>(out, cyc, 100, 1.0);
	hipEventRecord(a, 0);
	rate_kernel<OP><<<nblk, 256>>>(out, cyc, iters, 1.0);
	hipEventRecord(b, 0); hipEventSynchronize(b);
	float ms; hipEventElapsedTime(&ms, a, b);
	unsigned long long hc[16]; hipMemcpy(hc, cyc, sizeof(hc), hipMemcpyDeviceToHost);
	const double ninst = (double)iters * 32 * waves_per_simd;    // wave instructions per SIMD
	printf("%-10s %d wave(s)/SIMD: %.3f ms, %.2f ns per wave-instruction per SIMD, s_memtime cycles/inst %.2f (100 MHz counter?) \n",
		name, waves_per_simd, ms, ms * 1e6 / ninst, (double)hc[0] / (iters * 32.0));
}

int main()
{
	for (int w = 1; w <= 2; w++) {
		run<0>("v_fma_f64", w);
		run<1>("v_mul_f64", w);
		run<2>("v_add_f64", w);
		run<4>("v_max_f64", w);
	}
	return 0;
}
