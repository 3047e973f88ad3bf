// Microbenchmark: issue cost (shader cycles per wave64 instruction) of fp64 / fp32 VALU ops, v_readlane and LDS broadcast
// reads on gfx950, for 1 and 2 waves per SIMD. Build: hipcc --offload-arch=gfx950 -O3 -o fp64_rate fp64_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define N_IT 256
template <int MODE>
__global__ void __launch_bounds__(64) k(double* out, unsigned long long* cyc, double a0)
{
	double a[16];
#pragma unroll
	for (int i = 0; i < 16; ++i) a[i] = a0 + i + threadIdx.x;
	float f[16];
#pragma unroll
	for (int i = 0; i < 16; ++i) f[i] = (float)a[i];
	__shared__ double lds[256];
	lds[threadIdx.x] = a0; lds[threadIdx.x + 64] = a0; __syncthreads();
	const double b = a0 * 1.0000001, c = a0 * 0.25;
	unsigned long long t0 = __builtin_readcyclecounter();
	for (int it = 0; it < N_IT; ++it) {
#pragma unroll
		for (int i = 0; i < 16; ++i) {
			if (MODE == 0) a[i] = __builtin_fma(a[i], b, c);
			if (MODE == 1) a[i] = a[i] * b;
			if (MODE == 2) a[i] = a[i] + c;
			if (MODE == 3) f[i] = __builtin_fmaf(f[i], (float)b, (float)c);
			if (MODE == 4) { int lo = __builtin_amdgcn_readlane(__double2loint(a[i]), i); a[i] += lo; }  // readlane + cvt + add
			if (MODE == 5) a[i] += lds[(it + i) & 127];   // LDS broadcast read + add
			if (MODE == 6) { a[i] = a[i] * b; a[i] = a[i] + c; }   // dependent mul+add pairs (16 independent chains)
		}
	}
	unsigned long long t1 = __builtin_readcyclecounter();
	double s = 0; for (int i = 0; i < 16; ++i) s += a[i] + f[i];
	out[blockIdx.x * 64 + threadIdx.x] = s;
	if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int MODE>
void run(const char* name, int blocks, int ops_per_it)
{
	double* out; unsigned long long* cyc;
	hipMalloc(&out, sizeof(double) * 64 * blocks); hipMalloc(&cyc, sizeof(unsigned long long) * blocks);
	k<MODE><<<blocks, 64>>>(out, cyc, 1.0); hipDeviceSynchronize();
	k<MODE><<<blocks, 64>>>(out, cyc, 1.0); hipDeviceSynchronize();
	std::vector<unsigned long long> h(blocks);
	hipMemcpy(h.data(), cyc, sizeof(unsigned long long) * blocks, hipMemcpyDeviceToHost);
	double avg = 0; for (auto v : h) avg += v; avg /= blocks;
	printf("%-34s blocks=%5d: %.2f cycles per instruction-slot (%d slots/iter)\n", name, blocks, avg / (N_IT * ops_per_it), ops_per_it);
	hipFree(out); hipFree(cyc);
}
int main()
{
	for (int blocks : {256, 1024, 2048, 4096}) {   // 1 wave/CU, 1 wave/SIMD, 2 waves/SIMD, 4 waves/SIMD
		run<0>("v_fma_f64", blocks, 16);
		run<1>("v_mul_f64", blocks, 16);
		run<2>("v_add_f64", blocks, 16);
		run<6>("v_mul_f64 + v_add_f64 (dependent)", blocks, 32);
		run<3>("v_fma_f32", blocks, 16);
		run<4>("readlane+cvt+add", blocks, 16);
		run<5>("lds read + add_f64", blocks, 16);
	}
	return 0;
}
