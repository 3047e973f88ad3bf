// Microbenchmark: cost per "broadcast one lane's double to the wave + FMA" entry, the unit of work of the register LDL^T / triangular
// solves (dtrl_kernel_fast.h), for the candidate broadcast mechanisms on gfx950, at 1 and 2 waves per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 -o bcast_fma bcast_fma.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define N_IT 128
#define NE 16
__device__ __forceinline__ double bcast(double v, int src)
{
	int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
	int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
	return __hiloint2double(hi, lo);
}
template <int MODE>
__global__ void __launch_bounds__(64) k(double* out, unsigned long long* cyc, double a0)
{
	double h[NE];
#pragma unroll
	for (int i = 0; i < NE; ++i) h[i] = a0 + i + threadIdx.x;
	__shared__ double lds[128];
	const double lik = a0 * 1e-3 + threadIdx.x * 1e-6;
	double ak = a0 + threadIdx.x;
	unsigned long long t0 = __builtin_readcyclecounter();
	for (int it = 0; it < N_IT; ++it) {
		if (MODE == 0) {          // compiler-scheduled readlane pairs + fma
#pragma unroll
			for (int i = 0; i < NE; ++i) h[i] = __builtin_fma(-lik, bcast(ak, i + 1), h[i]);
		}
		if (MODE == 1) {          // batches of 4 broadcasts, then 4 FMAs
#pragma unroll
			for (int i = 0; i < NE; i += 4) {
				double b[4];
#pragma unroll
				for (int j = 0; j < 4; ++j) b[j] = bcast(ak, i + j + 1);
				__builtin_amdgcn_sched_barrier(0);
#pragma unroll
				for (int j = 0; j < 4; ++j) h[i + j] = __builtin_fma(-lik, b[j], h[i + j]);
				__builtin_amdgcn_sched_barrier(0);
			}
		}
		if (MODE == 2) {          // column parked in LDS, uniform-address 8-byte reads
			lds[threadIdx.x] = ak; __syncthreads();
#pragma unroll
			for (int i = 0; i < NE; ++i) h[i] = __builtin_fma(-lik, lds[i + 1], h[i]);
			__syncthreads();
		}
		if (MODE == 3) {          // column parked in LDS, uniform-address 16-byte reads
			lds[threadIdx.x] = ak; __syncthreads();
			const double2* l2 = reinterpret_cast<const double2*>(lds);
#pragma unroll
			for (int i = 0; i < NE; i += 2) { const double2 v = l2[i / 2 + 1]; h[i] = __builtin_fma(-lik, v.x, h[i]); h[i + 1] = __builtin_fma(-lik, v.y, h[i + 1]); }
			__syncthreads();
		}
		if (MODE == 4) {          // FMA only (no broadcast): the floor
#pragma unroll
			for (int i = 0; i < NE; ++i) h[i] = __builtin_fma(-lik, ak, h[i]);
		}
		if (MODE == 5) {          // readlane pairs only
			double s = 0;
#pragma unroll
			for (int i = 0; i < NE; ++i) { int lo = __builtin_amdgcn_readlane(__double2loint(ak), i + 1); int hi = __builtin_amdgcn_readlane(__double2hiint(ak), i + 2); h[i] = __hiloint2double(hi ^ __double2hiint(h[i]), lo); }
		}
		ak = h[it & 1 ? 3 : 5] * 0.5;   // next column depends on this one's result
	}
	unsigned long long t1 = __builtin_readcyclecounter();
	double s = 0; for (int i = 0; i < NE; ++i) s += h[i];
	out[blockIdx.x * 64 + threadIdx.x] = s;
	if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int MODE>
void run(const char* name, int blocks)
{
	double* out; unsigned long long* cyc;
	hipMalloc(&out, sizeof(double) * 64 * blocks); hipMalloc(&cyc, sizeof(unsigned long long) * blocks);
	k<MODE><<<blocks, 64>>>(out, cyc, 1.0); hipDeviceSynchronize();
	k<MODE><<<blocks, 64>>>(out, cyc, 1.0); hipDeviceSynchronize();
	std::vector<unsigned long long> h(blocks);
	hipMemcpy(h.data(), cyc, sizeof(unsigned long long) * blocks, hipMemcpyDeviceToHost);
	double avg = 0; for (auto v : h) avg += v; avg /= blocks;
	printf("%-44s blocks=%5d: %.2f ticks per entry\n", name, blocks, avg / (N_IT * NE));
	hipFree(out); hipFree(cyc);
}
int main()
{
	for (int blocks : {256, 1024, 2048}) {   // 1 wave/CU, 1 wave/SIMD, 2 waves/SIMD
		run<4>("fma only", blocks);
		run<5>("2 readlanes + xor (no fma)", blocks);
		run<0>("2 readlanes + fma (compiler order)", blocks);
		run<1>("2 readlanes + fma, batches of 4", blocks);
		run<2>("LDS column, ds_read_b64 + fma", blocks);
		run<3>("LDS column, ds_read_b128 + 2 fma", blocks);
	}
	return 0;
}
