// dtrl_trainer.hip -- HIP backend of the native MACE trainer step (include/dtrl_trainer.h; sequencing in dtrl_trainer_core.h, operands in dtrl_trainer_ops.h).
//
// Every layer pass is one launch of ONE LDS-tiled GEMM kernel over implicit operands: a 32 x 64 output tile per 256-thread workgroup (4 wavefronts),
// 2 x 4 outputs per thread, K staged through LDS in chunks of 16; the operand tiles are gathered element-wise by load_a / load_b (im2col, transposes,
// ones column), with consecutive lanes walking whichever index is contiguous in memory. The products are tiny (a batch-32 step is 0.33 GFLOP in ~25
// products) and every one of them is sized for the batch, so the step is bound by launch latency and per-launch parallelism, not by FLOPs: the split
// over (tile, sample / head / K-slab) gives each launch 16 - 190 workgroups, and weights (2.3 MB), history, activations and the replay rows never leave
// the device. Accumulation runs in k order per output, so results are deterministic and equal the plain-loop check build up to fp32 contraction.
#include "dtrl_trainer_core.h"
#include <hip/hip_runtime.h>
#include <algorithm>
#include <string>

namespace dtrl_tr {

constexpr int kTM = 32, kTN = 64, kTK = 16, kThreads = 256;

__global__ void __launch_bounds__(kThreads) tr_gemm_kernel(NetDims d, Work wk, GemmDesc g)
{
	__shared__ float As[kTK][kTM + 1];
	__shared__ float Bs[kTK][kTN + 1];
	const int tid = static_cast<int>(threadIdx.x), ty = tid >> 4, tx = tid & 15;
	const int z = static_cast<int>(blockIdx.z), m0 = static_cast<int>(blockIdx.y) * kTM, n0 = static_cast<int>(blockIdx.x) * kTN;
	const int k_begin = g.k0_step ? z * g.k0_step : 0;
	const int k_end = g.k0_step ? (k_begin + g.k0_step < g.K ? k_begin + g.k0_step : g.K) : g.K;
	float acc[2][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
	for (int k0 = k_begin; k0 < k_end; k0 += kTK) {
		for (int e = tid; e < kTM * kTK; e += kThreads) {
			int mm, kk;
			if (g.a_kfast) { kk = e % kTK; mm = e / kTK; } else { mm = e % kTM; kk = e / kTM; }
			const int m = m0 + mm, k = k0 + kk;
			As[kk][mm] = (m < g.M && k < k_end) ? load_a(d, wk, g, z, m, k) : 0.0f;
		}
		for (int e = tid; e < kTK * kTN; e += kThreads) {
			int nn, kk;
			if (g.b_kfast) { kk = e % kTK; nn = e / kTK; } else { nn = e % kTN; kk = e / kTN; }
			const int n = n0 + nn, k = k0 + kk;
			Bs[kk][nn] = (n < g.N && k < k_end) ? load_b(d, wk, g, z, k, n) : 0.0f;
		}
		__syncthreads();
#pragma unroll
		for (int kk = 0; kk < kTK; ++kk) {
			const float a0 = As[kk][ty], a1 = As[kk][ty + 16];
			const float b0 = Bs[kk][tx], b1 = Bs[kk][tx + 16], b2 = Bs[kk][tx + 32], b3 = Bs[kk][tx + 48];
			acc[0][0] = fmaf(a0, b0, acc[0][0]); acc[0][1] = fmaf(a0, b1, acc[0][1]); acc[0][2] = fmaf(a0, b2, acc[0][2]); acc[0][3] = fmaf(a0, b3, acc[0][3]);
			acc[1][0] = fmaf(a1, b0, acc[1][0]); acc[1][1] = fmaf(a1, b1, acc[1][1]); acc[1][2] = fmaf(a1, b2, acc[1][2]); acc[1][3] = fmaf(a1, b3, acc[1][3]);
		}
		__syncthreads();
	}
#pragma unroll
	for (int i = 0; i < 2; ++i)
#pragma unroll
		for (int j = 0; j < 4; ++j) {
			const int m = m0 + ty + 16 * i, n = n0 + tx + 16 * j;
			if (m < g.M && n < g.N) store_c(d, wk, g, z, m, n, acc[i][j]);
		}
}

template <class F>
__global__ void tr_foreach_kernel(int64_t n, F f)
{
	for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += static_cast<int64_t>(gridDim.x) * blockDim.x) f(i);
}

struct HipTrainerBE {
	int device_id = -1;
	hipStream_t stream = nullptr;
	std::string err_;
	bool init(std::string& err)
	{
		int count = 0;
		hipError_t e = hipGetDeviceCount(&count);
		if (e != hipSuccess || count <= 0) { err = std::string("no usable HIP device (") + hipGetErrorString(e) + "); the trainer has no CPU fallback"; return false; }
		if (device_id >= 0 && hipSetDevice(device_id) != hipSuccess) { err = "hipSetDevice failed"; return false; }
		return true;
	}
	bool ok() const { return err_.empty(); }
	const std::string& error() const { return err_; }
	bool chk(hipError_t e, const char* what) { if (e == hipSuccess) return true; if (err_.empty()) err_ = std::string(what) + ": " + hipGetErrorString(e); return false; }
	void set_stream(void* s) { stream = static_cast<hipStream_t>(s); }
	void* alloc_dev(size_t bytes) { void* p = nullptr; if (!chk(hipMalloc(&p, bytes), "hipMalloc")) return nullptr; chk(hipMemset(p, 0, bytes), "hipMemset"); return p; }
	void free_dev(void* p) { hipFree(p); }
	void* alloc_host(size_t bytes) { void* p = nullptr; return chk(hipHostMalloc(&p, bytes, hipHostMallocMapped | hipHostMallocCoherent), "hipHostMalloc") ? p : nullptr; }
	void free_host(void* p) { hipHostFree(p); }
	void h2d(void* dst, const void* src, size_t n) { chk(hipMemcpyAsync(dst, src, n, hipMemcpyHostToDevice, stream), "hipMemcpy H2D"); chk(hipStreamSynchronize(stream), "sync"); }
	void d2h(void* dst, const void* src, size_t n) { chk(hipMemcpyAsync(dst, src, n, hipMemcpyDeviceToHost, stream), "hipMemcpy D2H"); chk(hipStreamSynchronize(stream), "sync"); }
	void d2d(void* dst, const void* src, size_t n) { chk(hipMemcpyAsync(dst, src, n, hipMemcpyDeviceToDevice, stream), "hipMemcpy D2D"); }
	void sync() { chk(hipStreamSynchronize(stream), "hipStreamSynchronize"); }
	void gemm(const NetDims& d, const Work& wk, const GemmDesc& g)
	{
		const dim3 grid((g.N + kTN - 1) / kTN, (g.M + kTM - 1) / kTM, g.Z);
		hipLaunchKernelGGL(tr_gemm_kernel, grid, dim3(kThreads), 0, stream, d, wk, g);
		chk(hipGetLastError(), "gemm launch");
	}
	template <class F>
	void for_each(int64_t n, const F& f)
	{
		if (n <= 0) return;
		const int blocks = static_cast<int>(std::min<int64_t>((n + 255) / 256, 2048));
		hipLaunchKernelGGL(tr_foreach_kernel<F>, dim3(blocks), dim3(n == 1 ? 1 : 256), 0, stream, n, f);
		chk(hipGetLastError(), "foreach launch");
	}
};

}  // namespace dtrl_tr

using Backend = dtrl_tr::HipTrainerBE;
#include "dtrl_trainer_capi.inc"
