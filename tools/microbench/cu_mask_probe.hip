// Probe: which compute units does a CU-masked stream use, and does a small kernel on an unmasked stream find the units the mask left free while a
// long-running kernel fills every wavefront slot of the masked ones? For reserve = 0, 1, 2, 4, 8 units per XCD (the engine's DTRL_RESERVE_CUS layout: the top
// 8 k mask bits): (1) a hog kernel (2048 workgroups x 64 threads, 20 KB LDS each = 8 per CU, spinning ~3 ms) on the masked stream records its (XCC, SE, CU)
// ids; (2) 20 small kernels (96 workgroups x 256 threads, 16 KB LDS) back to back on a plain stream while the hog runs: their total wall time, and where
// they ran. Build: hipcc --offload-arch=gfx950 -O3 -o cu_mask_probe cu_mask_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <set>
#include <vector>
#include <chrono>
#include <thread>

__device__ __forceinline__ unsigned where()
{
	unsigned hw, xcc;
	asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
	asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
	const unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 0x1, se = (hw >> 13) & 0x7;
	return ((xcc & 0xf) << 12) | (se << 8) | (sh << 4) | cu;
}

__global__ void __launch_bounds__(64) hog(unsigned* out, long long ticks)
{
	__shared__ float pad[5 * 1024];
	pad[threadIdx.x] = 1.0f;
	const long long t0 = wall_clock64();
	while (wall_clock64() - t0 < ticks) { pad[threadIdx.x] += 1.0f; }
	if (threadIdx.x == 0) out[blockIdx.x] = where() | (pad[0] > 0 ? 0u : 1u << 31);
}

__global__ void __launch_bounds__(256) small(unsigned* out, int iter)
{
	__shared__ float pad[4 * 1024];
	pad[threadIdx.x] = 1.0f;
	__syncthreads();
	float s = 0;
	for (int i = 0; i < 256; ++i) s += pad[(threadIdx.x + i) & 1023];
	if (threadIdx.x == 0) out[iter * gridDim.x + blockIdx.x] = where() | (s > 0 ? 0u : 1u << 31);
}

__global__ void __launch_bounds__(64) hog2(long long ticks, long long* start_stamp, unsigned* out = nullptr)
{
	if (out && threadIdx.x == 0) out[blockIdx.x] = where();
	__shared__ float pad[5 * 1024];
	const long long t0 = wall_clock64();
	if (blockIdx.x == 0 && threadIdx.x == 0 && start_stamp) *start_stamp = t0;
	pad[threadIdx.x] = 1.0f;
	while (wall_clock64() - t0 < ticks) pad[threadIdx.x] += 1.0f;
	if (pad[threadIdx.x] < 0) __builtin_trap();
}
__global__ void __launch_bounds__(256) stamp(long long* out)
{
	__shared__ float pad[4 * 1024];
	pad[threadIdx.x] = 1.0f;
	__syncthreads();
	if (blockIdx.x == 0 && threadIdx.x == 0) *out = wall_clock64() + (pad[1] > 2.0f ? 1 : 0);
}

int main()
{
	hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
	const int cus = prop.multiProcessorCount;
	printf("compute units %d, wall clock %d kHz\n", cus, prop.clockRate);
	unsigned *d_hog, *d_small;
	hipMalloc(&d_hog, 4 * 4096); hipMalloc(&d_small, 4 * 96 * 20);
	const long long ticks = 300000;   // wall_clock64 runs at 100 MHz -> 3 ms
	for (int reserve : {0, 1, 2, 4, 8}) {
		hipStream_t hs, ss;
		std::vector<uint32_t> mask((cus + 31) / 32, 0u);
		for (int b = 0; b < cus - 8 * reserve; ++b) mask[b / 32] |= 1u << (b % 32);
		if (reserve) hipExtStreamCreateWithCUMask(&hs, (uint32_t)mask.size(), mask.data()); else hipStreamCreateWithFlags(&hs, hipStreamNonBlocking);
		hipStreamCreateWithFlags(&ss, hipStreamNonBlocking);
		for (int rep = 0; rep < 2; ++rep) {
			hipMemset(d_hog, 0, 4 * 4096); hipMemset(d_small, 0, 4 * 96 * 20);
			hipDeviceSynchronize();
			hipLaunchKernelGGL(hog, dim3(2048), dim3(64), 0, hs, d_hog, ticks);
			hipLaunchKernelGGL(hog, dim3(2048), dim3(64), 0, hs, d_hog + 2048, ticks);     // a second launch queued behind, as the other env group's is
			std::this_thread::sleep_for(std::chrono::microseconds(500));
			const auto t0 = std::chrono::steady_clock::now();
			for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(small, dim3(96), dim3(256), 0, ss, d_small, i);
			hipStreamSynchronize(ss);
			const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
			hipDeviceSynchronize();
			if (rep == 0) continue;
			std::vector<unsigned> h(4096), s(96 * 20);
			hipMemcpy(h.data(), d_hog, 4 * 4096, hipMemcpyDeviceToHost); hipMemcpy(s.data(), d_small, 4 * 96 * 20, hipMemcpyDeviceToHost);
			std::set<unsigned> hset(h.begin(), h.begin() + 2048), sset(s.begin(), s.end());
			int per_xcc_h[16] = {0}, per_xcc_s[16] = {0}, overlap = 0;
			for (unsigned v : hset) per_xcc_h[(v >> 12) & 0xf]++;
			for (unsigned v : sset) { per_xcc_s[(v >> 12) & 0xf]++; overlap += hset.count(v); }
			printf("reserve %d per XCD: hog on %3zu distinct CUs (per XCC:", reserve, hset.size());
			for (int x = 0; x < 8; ++x) printf(" %d", per_xcc_h[x]);
			printf("); 20 small kernels while it runs: %8.1f us total, on %3zu distinct CUs (per XCC:", us, sset.size());
			for (int x = 0; x < 8; ++x) printf(" %d", per_xcc_s[x]);
			printf("), %d of them also used by the hog\n", overlap);
			if (reserve == 4) { printf("   small kernels' CUs (xcc.se.sh.cu):"); for (unsigned v : sset) printf(" %u.%u.%u.%u", (v >> 12) & 0xf, (v >> 8) & 0x7, (v >> 4) & 1, v & 0xf); printf("\n"); }
		}
		hipStreamDestroy(hs); hipStreamDestroy(ss);
	}
	// (3) the engine's situation: TWO masked streams, the second launch not queued behind the first but actively waiting for wavefront slots in its own
	// hardware queue; the small kernels on each of 10 plain streams in turn (does a launch stalled in dispatch block other queues that share its pipe?)
	{
		const int reserve = 4;
		hipStream_t ha, hb, ss[10];
		std::vector<uint32_t> mask((cus + 31) / 32, 0u);
		for (int b = 0; b < cus - 8 * reserve; ++b) mask[b / 32] |= 1u << (b % 32);
		hipExtStreamCreateWithCUMask(&ha, (uint32_t)mask.size(), mask.data()); hipExtStreamCreateWithCUMask(&hb, (uint32_t)mask.size(), mask.data());
		for (auto& st : ss) hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
		for (int k = 0; k < 10; ++k) {
			hipDeviceSynchronize();
			hipLaunchKernelGGL(hog, dim3(2048), dim3(64), 0, ha, d_hog, ticks);
			hipLaunchKernelGGL(hog, dim3(2048), dim3(64), 0, hb, d_hog + 2048, ticks);
			std::this_thread::sleep_for(std::chrono::microseconds(500));
			const auto t0 = std::chrono::steady_clock::now();
			for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(small, dim3(96), dim3(256), 0, ss[k], d_small, i);
			hipStreamSynchronize(ss[k]);
			const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
			printf("two masked streams (reserve 4), both launches active; 20 small kernels on plain stream %d: %8.1f us\n", k, us);
		}
		hipDeviceSynchronize();
	}
	// (4) narrowing down why a stamp kernel behind the occupants waits when part (3)'s small kernels did not
	{
		const int reserve = 4;
		hipStream_t hm[2], ss[6];
		std::vector<uint32_t> mask((cus + 31) / 32, 0u);
		for (int b = 0; b < cus - 8 * reserve; ++b) mask[b / 32] |= 1u << (b % 32);
		for (auto& st : hm) hipExtStreamCreateWithCUMask(&st, (uint32_t)mask.size(), mask.data());
		for (auto& st : ss) hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
		long long* stamps; hipHostMalloc(&stamps, 8 * 16, hipHostMallocMapped | hipHostMallocCoherent);
		long long* dstamps; hipMalloc(&dstamps, 8 * 16);
		for (int variant = 0; variant < 6; ++variant) {
			printf("narrow %d (%s):", variant, variant == 0 ? "poll, stamp->host" : variant == 1 ? "sleep 500us, stamp->host" : variant == 2 ? "poll, stamp->device" : variant == 3 ? "poll, 20 small + stream sync" : variant == 4 ? "occupant without host stamp, sleep 200us, stamp->device" : "3 ms occupants w/o host stamp, sleep 500us, 20 small + stream sync");
			for (int c = 0; c < 6; ++c) {
				for (int k = 0; k < 16; ++k) stamps[k] = 0;
				hipMemset(dstamps, 0, 8 * 16);
				hipDeviceSynchronize();
				const long long T = variant == 5 ? 300000LL : 40000LL;
				const bool host_stamp = variant < 4;
				hipLaunchKernelGGL(hog2, dim3(2048), dim3(64), 0, hm[0], T, host_stamp ? stamps : dstamps, (unsigned*)nullptr);
				hipLaunchKernelGGL(hog2, dim3(2048), dim3(64), 0, hm[1], T, (long long*)nullptr, (unsigned*)nullptr);
				if (variant == 1 || variant == 5) std::this_thread::sleep_for(std::chrono::microseconds(500));
				else if (variant == 4) std::this_thread::sleep_for(std::chrono::microseconds(200));
				else while (*(volatile long long*)stamps == 0) {}
				if (variant == 3 || variant == 5) {
					const auto t0 = std::chrono::steady_clock::now();
					for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(small, dim3(96), dim3(256), 0, ss[c], d_small, i);
					hipStreamSynchronize(ss[c]);
					printf(" [%.0f us for 20]", std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count());
					hipDeviceSynchronize();
				} else {
					const bool dev = variant == 2 || variant == 4;
					hipLaunchKernelGGL(stamp, dim3(8), dim3(256), 0, ss[c], dev ? dstamps + 1 : stamps + 1);
					hipDeviceSynchronize();
					long long h[2] = {stamps[0], stamps[1]};
					if (dev) { long long d2[2]; hipMemcpy(d2, dstamps, 16, hipMemcpyDeviceToHost); h[1] = d2[1]; if (!host_stamp) h[0] = d2[0]; }
					printf(" %.0f", (h[1] - h[0]) / 100.0);
				}
			}
			printf("\n");
		}
	}
	return 0;
}
