// Cost of a device-wide barrier inside one launch (would a single persistent launch per trainer step beat ~45 small launches of 6-20 us?).
// W workgroups x 256 threads, 200 barriers (monotone counter in device memory: release fence + atomic add + acquire spin by one thread, workgroup barriers
// around it, acquire fence by every wave), each followed by a token write/read across workgroups to verify visibility across the 8 XCDs' L2s.
// Build: hipcc --offload-arch=gfx950 -O3 -o grid_barrier grid_barrier.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <chrono>

__device__ __forceinline__ void grid_sync(unsigned* ctr, unsigned nblocks, unsigned& epoch)
{
	__syncthreads();
	if (threadIdx.x == 0) {
		const unsigned target = (++epoch) * nblocks;
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
		__hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
	} else ++epoch;
	__syncthreads();
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
}

__global__ void __launch_bounds__(256) k_barriers(unsigned* ctr, float* data, int n_bar, int* errors)
{
	unsigned epoch = 0;
	const unsigned nb = gridDim.x;
	for (int b = 0; b < n_bar; ++b) {
		// every workgroup writes a token, after the barrier reads its neighbour's (a workgroup on another XCD: blockIdx + 1 goes to the next XCD)
		if (threadIdx.x == 0) data[blockIdx.x] = static_cast<float>(b * 1000 + blockIdx.x);
		grid_sync(ctr, nb, epoch);
		if (threadIdx.x == 0) { const unsigned o = (blockIdx.x + 1) % nb; if (data[o] != static_cast<float>(b * 1000 + o)) atomicAdd(errors, 1); }
		grid_sync(ctr, nb, epoch);
	}
}

int main()
{
	unsigned* ctr; float* data; int* err;
	hipMalloc(&ctr, 4); hipMalloc(&data, 4 * 4096); hipMalloc(&err, 4);
	for (int W : {32, 96, 256, 512}) {
		for (int rep = 0; rep < 2; ++rep) {
			hipMemset(ctr, 0, 4); hipMemset(err, 0, 4); hipDeviceSynchronize();
			const int n_bar = 100;
			const auto t0 = std::chrono::steady_clock::now();
			hipLaunchKernelGGL(k_barriers, dim3(W), dim3(256), 0, 0, ctr, data, n_bar, err);
			hipDeviceSynchronize();
			const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
			int e = 0; hipMemcpy(&e, err, 4, hipMemcpyDeviceToHost);
			if (rep) printf("%4d workgroups: %d barriers in %.1f us -> %.2f us per barrier (incl. launch ~10 us); visibility errors %d\n", W, 2 * n_bar, us, us / (2 * n_bar), e);
		}
	}
	return 0;
}
