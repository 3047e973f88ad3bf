// dtrl_backend_hip.hip -- the product backend: HIP runtime + the gfx950 frame kernel.
//
// Launch geometry: one 64-lane wavefront (one workgroup) per environment, so every __syncthreads() in the lane-phase
// code is a single-wave barrier; a 4096-env batch is 4096 workgroups (16 per CU), enough to fill all 256 CUs / 8 XCDs.
// Workgroup b lands on XCD b % 8 (observed dispatch order), i.e. consecutive envs spread across XCDs and each XCD's L2
// holds only its own envs' state/terrain records -- the per-env records are private, nothing is shared between XCDs
// except the read-only model and policy weights.
#include "dtrl_engine.h"
#include "dtrl_kernel_fast.h"
#include "dtrl_terrain_dev.h"
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <thread>
#include <cstdlib>
#include <cstring>
#include <map>
#include <vector>

namespace dtrl {

__global__ void __launch_bounds__(kGroup) dtrl_frame_kernel(const DevModel* __restrict__ gm, RunParams rp, DevBuffers buf, int n_envs, int n_steps, real dt, int frame_end)
{
	__shared__ WSRef ws;
	if (static_cast<int>(blockIdx.x) >= n_envs) return;
	const int env = buf.env_list ? buf.env_list[blockIdx.x] : static_cast<int>(blockIdx.x);
	env_frame<RefPath>(ws, *gm, rp, buf, env, n_steps, dt, frame_end != 0);
}

// register-resident fast path (dtrl_kernel_fast.h), one instantiation per skeleton of the shipped characters (dtrl_topo.h)
// Experiment builds (docs/EXPERIMENTS.md 13, tools/occupancy_ab.sh; never the shipped library): -DDTRL_DYN_LDS puts the workspace into DYNAMIC LDS, so that the compiler no
// longer derives "two waves per SIMD at most" from the 20 KB static allocation and honours -DDTRL_WAVES_PER_EU=3 (<= 168 registers per lane): the register diet a third wave
// per SIMD would need, priced at unchanged occupancy. Run-time knob of the shipped kernel: DTRL_LDS_PAD=<bytes> of dynamic LDS on top (fewer workgroups per CU: the
// throughput-vs-occupancy curve from the other side).
#ifndef DTRL_WAVES_PER_EU
#define DTRL_WAVES_PER_EU 2
#endif
#ifndef DTRL_WAVES_DOG
#define DTRL_WAVES_DOG DTRL_WAVES_PER_EU
#endif
#ifndef DTRL_WAVES_RAPTOR
#define DTRL_WAVES_RAPTOR DTRL_WAVES_PER_EU
#endif
template <class Topo> struct WavesPerEu { static constexpr int value = DTRL_WAVES_PER_EU; };
template <> struct WavesPerEu<TopoDog> { static constexpr int value = DTRL_WAVES_DOG; };          // (the fp32 build gives each skeleton's instance its own register budget:
template <> struct WavesPerEu<TopoRaptor> { static constexpr int value = DTRL_WAVES_RAPTOR; };    //  profiles/r06_fp32_physics.txt)
template <class Topo>
__global__ void __launch_bounds__(kGroup, WavesPerEu<Topo>::value) dtrl_frame_kernel_fast(const DevModel* __restrict__ gm, RunParams rp, DevBuffers buf, int n_envs, int n_steps, real dt, int frame_end)
{
#if defined(DTRL_DYN_LDS)
	extern __shared__ __align__(16) unsigned char dtrl_dyn_lds[];
	WSFast& ws = *reinterpret_cast<WSFast*>(dtrl_dyn_lds);
#else
	__shared__ WSFast ws;
#endif
	if (static_cast<int>(blockIdx.x) >= n_envs) return;
	const int env = buf.env_list ? buf.env_list[blockIdx.x] : static_cast<int>(blockIdx.x);
#if defined(__HIP_DEVICE_COMPILE__)
	env_frame<FastPath<Topo>>(ws, *gm, rp, buf, env, n_steps, dt, frame_end != 0);
#endif
}

// dst[i] = idx[i] >= 0 ? src[idx[i]] : 0: re-lays a policy blob handed over in device memory into the kernel's weight layout
// Calibration of the side streams (HipBackend::CalibrateSideStreams). dtrl_occupy: a stand-in for a frame launch's residency (one wavefront per workgroup,
// 20 KB of LDS -> 8 per compute unit, alive for `ticks` of the 100 MHz wall clock); block 0 stamps its start. dtrl_stamp: when did this stream's kernel get to run.
__global__ void __launch_bounds__(64) dtrl_occupy(long long ticks, long long* __restrict__ start_stamp)
{
	__shared__ float pad[5 * 1024];
	const long long t0 = wall_clock64();
	if (blockIdx.x == 0 && threadIdx.x == 0 && start_stamp) *start_stamp = t0;
	pad[threadIdx.x] = 1.0f;
	while (wall_clock64() - t0 < ticks) pad[threadIdx.x] += 1.0f;
	if (pad[threadIdx.x] < 0) __builtin_trap();
}
__global__ void __launch_bounds__(256) dtrl_stamp(long long* __restrict__ out)
{
	__shared__ float pad[4 * 1024];          // (a side kernel's typical footprint: does not fit beside a resident frame workgroup set)
	pad[threadIdx.x] = 1.0f;
	__syncthreads();
	if (blockIdx.x == 0 && threadIdx.x == 0) *out = wall_clock64() + (pad[1] > 2.0f ? 1 : 0);
}

__global__ void dtrl_gather_f32(float* __restrict__ dst, const float* __restrict__ src, const int32_t* __restrict__ idx, size_t n)
{
	for (size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += static_cast<size_t>(gridDim.x) * blockDim.x) {
		const int32_t k = idx[i];
		dst[i] = k >= 0 ? src[k] : 0.0f;
	}
}

// gr[ids[b]] = staged[b]: one workgroup per record (4 176 B as 16-byte words)
// -terrain_gen= device: one thread per env; only the few envs whose window must move (or that fell) do any work
__global__ void dtrl_terrain_boundary(DevBuffers buf, int e0, int n, int mode, const int32_t* __restrict__ env_list)
{
	const int k = static_cast<int>(blockIdx.x * blockDim.x + threadIdx.x);
	if (k >= n) return;
	const int e = env_list ? env_list[k] : e0 + k;
	tg_env_boundary(buf.gr[e], buf.gen[e], buf.status[e], *buf.tcfg, mode, e, buf.dist_ring, buf.dist_count, buf.dist_cap);
}
// launch order of a group's next frame: counting sort on cost / 16, costliest first (one workgroup; the order inside a bucket is whatever the
// atomics produce -- it only decides which wavefront starts first)
constexpr int kOrderBuckets = 1024;
__global__ void __launch_bounds__(1024) dtrl_order_by_cost(const EnvStatus* __restrict__ status, int e0, int n, int32_t* __restrict__ order)
{
	__shared__ int hist[kOrderBuckets];
	__shared__ int part[kOrderBuckets / 64];
	const int t = static_cast<int>(threadIdx.x);
	hist[t] = 0;
	__syncthreads();
	auto key = [&](int e) { int k = status[e].cost >> 4; k = k < 0 ? 0 : (k >= kOrderBuckets ? kOrderBuckets - 1 : k); return kOrderBuckets - 1 - k; };
	for (int i = t; i < n; i += kOrderBuckets) atomicAdd(&hist[key(e0 + i)], 1);
	__syncthreads();
	// exclusive scan of 1024 counters: per-wave serial prefix over 64 entries by lane 0 of each wave would idle; a two-level scan instead
	int v = hist[t];
	int incl = v;
	for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(incl, d, 64); if ((t & 63) >= d) incl += o; }
	if ((t & 63) == 63) part[t >> 6] = incl;
	__syncthreads();
	if (t == 0) { int run = 0; for (int w = 0; w < kOrderBuckets / 64; ++w) { const int c = part[w]; part[w] = run; run += c; } }
	__syncthreads();
	hist[t] = part[t >> 6] + incl - v;
	__syncthreads();
	for (int i = t; i < n; i += kOrderBuckets) { const int e = e0 + i; order[e0 + atomicAdd(&hist[key(e)], 1)] = e; }
}
// packed tuple drain (dtrl_drain_tuples_packed). The pending rows of the ring are put in (env id, ring position) order by a segmented counting sort on the
// env id -- O(rows + envs), one workgroup (round 2 ranked every row against every other: 36 M compares at a full 65 536-env ring) --, the first block_rows
// of them are copied into the caller's block, and whatever does not fit is CARRIED: moved to the front of the ring, where the next drain of this ring
// finds it in front of the newer rows (nothing is dropped because a block was small). Three launches on the drain stream:
//   dtrl_tuple_order   order[r] = ring position of the r-th row; meta = {take, lost to a full ring, carried, stored}
//   dtrl_tuple_pack    block row 1 + r <- ring row order[r] (r < take), carry staging row r - take <- ring row order[r] (r >= take)
//   dtrl_tuple_finish  ring rows [0, carried) <- staging, header row of the block, ring cursor = carried, drained / dropped totals
constexpr int kOrderThreads = 1024;
__global__ void __launch_bounds__(kOrderThreads) dtrl_tuple_order(DevBuffers buf, int block_rows, int n_envs, int32_t* __restrict__ order, int32_t* __restrict__ hist, int32_t* __restrict__ meta)
{
	__shared__ int part[kOrderThreads];
	const int t = static_cast<int>(threadIdx.x);
	const int cnt = buf.tuple_count[0];
	const int n = cnt < buf.tuple_cap ? cnt : buf.tuple_cap;
	for (int e = t; e <= n_envs; e += kOrderThreads) hist[e] = 0;
	__syncthreads();
	for (int k = t; k < n; k += kOrderThreads) atomicAdd(&hist[buf.tuple_env[k]], 1);
	__syncthreads();
	// exclusive scan over the envs: a contiguous chunk per thread, the chunk totals scanned across the workgroup
	const int chunk = (n_envs + kOrderThreads - 1) / kOrderThreads;
	const int c0 = t * chunk, c1 = (c0 + chunk < n_envs) ? c0 + chunk : n_envs;
	int sum = 0;
	for (int e = c0; e < c1; ++e) sum += hist[e];
	part[t] = sum;
	__syncthreads();
	for (int d = 1; d < kOrderThreads; d <<= 1) { const int o = t >= d ? part[t - d] : 0; __syncthreads(); part[t] += o; __syncthreads(); }
	int run = part[t] - sum;
	for (int e = c0; e < c1; ++e) { const int c = hist[e]; hist[e] = run; run += c; }
	__syncthreads();
	// scatter: the atomics hand out the slots of an env's segment in arbitrary order ...
	for (int k = t; k < n; k += kOrderThreads) order[atomicAdd(&hist[buf.tuple_env[k]], 1)] = k;
	__syncthreads();
	// ... so every segment with more than one row (an env that completed two cycles between drains, or carried rows) is put in ring order; after the
	// scatter hist[e] is the END of env e's segment
	for (int e = t; e < n_envs; e += kOrderThreads) {
		const int s0 = e ? hist[e - 1] : 0, s1 = hist[e];
		for (int i = s0 + 1; i < s1; ++i) { const int v = order[i]; int j = i - 1; while (j >= s0 && order[j] > v) { order[j + 1] = order[j]; --j; } order[j + 1] = v; }
	}
	if (t == 0) { const int take = n < block_rows ? n : block_rows; meta[0] = take; meta[1] = cnt - n; meta[2] = n - take; meta[3] = n; }
}
__global__ void dtrl_tuple_pack(DevBuffers buf, const int32_t* __restrict__ order, const int32_t* __restrict__ meta, float* __restrict__ block, int64_t env_id_base, float* __restrict__ c_rows, uint32_t* __restrict__ c_flags, int32_t* __restrict__ c_env)
{
	const int take = meta[0], n = meta[3], W = buf.W;
	for (int r = static_cast<int>(blockIdx.x); r < n; r += static_cast<int>(gridDim.x)) {
		const int k = order[r];
		const float* src = buf.tuple_rows + static_cast<size_t>(k) * W;
		if (r < take) {
			float* dst = block + static_cast<size_t>(1 + r) * (W + 2);
			for (int i = static_cast<int>(threadIdx.x); i < W; i += static_cast<int>(blockDim.x)) dst[i] = src[i];
			if (threadIdx.x == 0) { dst[W] = __int_as_float(static_cast<int>(buf.tuple_flags[k])); dst[W + 1] = __int_as_float(static_cast<int>(env_id_base + buf.tuple_env[k])); }
		} else {
			float* dst = c_rows + static_cast<size_t>(r - take) * W;
			for (int i = static_cast<int>(threadIdx.x); i < W; i += static_cast<int>(blockDim.x)) dst[i] = src[i];
			if (threadIdx.x == 0) { c_flags[r - take] = buf.tuple_flags[k]; c_env[r - take] = buf.tuple_env[k]; }
		}
	}
}
__global__ void dtrl_tuple_finish(DevBuffers buf, const int32_t* __restrict__ meta, float* __restrict__ block, const float* __restrict__ c_rows, const uint32_t* __restrict__ c_flags, const int32_t* __restrict__ c_env)
{
	const int take = meta[0], lost = meta[1], carry = meta[2], W = buf.W;
	const int t = static_cast<int>(threadIdx.x);
	for (int r = static_cast<int>(blockIdx.x); r < carry; r += static_cast<int>(gridDim.x)) {
		float* dst = buf.tuple_rows + static_cast<size_t>(r) * W;
		const float* src = c_rows + static_cast<size_t>(r) * W;
		for (int i = t; i < W; i += static_cast<int>(blockDim.x)) dst[i] = src[i];
		if (t == 0) { buf.tuple_flags[r] = c_flags[r]; buf.tuple_env[r] = c_env[r]; }
	}
	if (blockIdx.x == 0) {
		for (int i = t; i < W + 2; i += static_cast<int>(blockDim.x)) block[i] = (i == 0) ? __int_as_float(take) : (i == 1) ? __int_as_float(lost) : (i == 2) ? __int_as_float(carry) : 0.0f;
		if (t == 0) { buf.tuple_count[1] += take; buf.tuple_count[2] += lost; buf.tuple_count[0] = carry; }
	}
}
class HipBackend : public Backend {
public:
	~HipBackend() override
	{
		for (auto& ev : events_) { hipEventDestroy(ev.first); hipEventDestroy(ev.second); }
		for (auto& m : marks_) if (m.second) hipEventDestroy(m.second);
		if (policy_ready_) hipEventDestroy(policy_ready_);
		if (!owned_.empty()) { for (int i = kNumStreams / 2; i < kNumStreams; ++i) streams_[i] = nullptr; for (hipStream_t st : owned_) hipStreamDestroy(st); }
		for (hipStream_t st : streams_) if (st) hipStreamDestroy(st);
	}
	bool Init(int device_id, std::string& err) override
	{
		int count = 0;
		hipError_t e = hipGetDeviceCount(&count);
		if (e != hipSuccess || count <= 0) { err = std::string("no usable HIP device (") + hipGetErrorString(e) + "); the engine has no CPU fallback"; return false; }
		if (device_id >= 0) { e = hipSetDevice(device_id); if (e != hipSuccess) { err = std::string("hipSetDevice: ") + hipGetErrorString(e); return false; } }
		streams_.assign(kNumStreams, nullptr);
		// DTRL_RESERVE_CUS=k keeps k compute units per XCD out of the frame launches: a frame launch fills every wavefront slot of the CUs it may use for
		// milliseconds, and whatever else needs the GPU meanwhile -- RCCL's collective kernels, the tuple drain, a trainer's copies -- would wait for a wavefront
		// to retire (nothing does in the first ~2 ms of a frame). The env-group streams (the first kNumStreams / 2) get a CU mask without those units (mask
		// bit i = compute unit i / #XCD of XCD i % #XCD, so the top 8 k bits are k units on each of the 8 XCDs); the other streams, and every other stream of
		// the process, may use all of them. Costs k / 32 of the rollout rate; worth it when something has to overlap the rollout (multi-rank exchange).
		int reserve = reserve_arg_;
		if (reserve < 0) { reserve = 0; if (const char* env = std::getenv("DTRL_RESERVE_CUS")) reserve = std::atoi(env); }
		reserve = std::max(0, std::min(8, reserve));
		hipDeviceProp_t prop;
		int cus = 0;
		if (reserve > 0 && hipGetDeviceProperties(&prop, device_id >= 0 ? device_id : 0) == hipSuccess) cus = prop.multiProcessorCount;
		constexpr int kXcd = 8;
		for (int i = 0; i < kNumStreams; ++i) {
			if (reserve > 0 && cus >= 2 * kXcd * reserve && i < kNumStreams / 2) {
				std::vector<uint32_t> mask((cus + 31) / 32, 0u);
				for (int b = 0; b < cus - kXcd * reserve; ++b) mask[b / 32] |= 1u << (b % 32);
				e = hipExtStreamCreateWithCUMask(&streams_[i], static_cast<uint32_t>(mask.size()), mask.data());
			} else e = hipStreamCreateWithFlags(&streams_[i], hipStreamNonBlocking);
			if (e != hipSuccess) { err = std::string("hipStreamCreate: ") + hipGetErrorString(e); return false; }
		}
		stream_ = streams_[0];
		masked_ = reserve > 0 && cus >= 2 * kXcd * reserve;
		if (masked_ && !CalibrateSideStreams(err)) return false;
		return true;
	}
	// Which streams can actually USE the reserved compute units while frame launches are in flight? Measured on the MI355X (tools/microbench/cu_mask_probe.hip,
	// DESIGN 9): a launch that is waiting for wavefront slots (the second env group's, while the first fills the unmasked units) holds up OTHER hardware queues
	// too -- 3 to 5 of 10 plain streams of a process see their kernels start only when that launch has been placed or has finished (2.7 / 5.6 ms for a 3 ms
	// occupant), although the reserved units are idle; the rest start within 7 us. Which streams are affected depends on how the runtime mapped them onto
	// hardware queues, so it is measured here, once: two occupant launches on the env-group streams 0 and 1, a stamp kernel on every candidate stream, and the
	// candidates ordered by how long after the occupants' start their stamp ran (worst of 3 rounds). The plain engine streams (kNumStreams / 2 ..) are then
	// re-seated on the quickest candidates -- the tuple-drain stream first -- and the next two are handed out as side streams (dtrl_side_stream: the trainer's
	// launches, the exchange's collective).
	bool CalibrateSideStreams(std::string& err)
	{
		constexpr int kCand = 12, kBurst = 10;
		constexpr long long kTicks = 120000;         // 1.2 ms per occupant
		std::vector<hipStream_t> cand;
		for (int i = kNumStreams / 2; i < kNumStreams; ++i) cand.push_back(streams_[i]);
		while (static_cast<int>(cand.size()) < kCand) { hipStream_t st; if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) break; cand.push_back(st); extra_.push_back(st); }
		long long* stamps = nullptr;
		auto drop_extras = [&]() { for (hipStream_t st : extra_) hipStreamDestroy(st); extra_.clear(); };   // (a failed calibration keeps the engine's own streams only)
		// the assignment below hands out kNumStreams / 2 + 2 streams (the plain engine streams + two side streams): with fewer candidates (stream creation failed)
		// there is nothing to choose from -- keep the engine's own streams, hand out no side streams
		if (static_cast<int>(cand.size()) < kNumStreams - kNumStreams / 2 + 2) { drop_extras(); err = "side-stream calibration: could not create enough candidate streams"; return false; }
		if (!Check(hipHostMalloc(&stamps, sizeof(long long) * 4, hipHostMallocMapped | hipHostMallocCoherent), "hipHostMalloc")) { err = err_; drop_extras(); return false; }
		std::vector<long long> worst(cand.size(), 0);   // microseconds x 100 (the sort below only compares)
		for (int r = 0; r < 3; ++r) {                 // (round 0 warms the code objects and the streams' queues up; the worse of rounds 1 and 2 counts)
			for (size_t c = 0; c < cand.size(); ++c) {
				stamps[0] = 0;
				hipDeviceSynchronize();
				hipLaunchKernelGGL(dtrl_occupy, dim3(2048), dim3(64), 0, streams_[0], r == 0 ? 2000LL : kTicks, stamps);
				hipLaunchKernelGGL(dtrl_occupy, dim3(2048), dim3(64), 0, streams_[1], r == 0 ? 2000LL : kTicks, static_cast<long long*>(nullptr));
				{   // the first occupant is running, the second waits for slots (bounded wait: a launch that never starts must not hang the creation)
					const auto deadline = std::chrono::steady_clock::now() + std::chrono::seconds(2);
					while (*static_cast<volatile long long*>(stamps) == 0 && std::chrono::steady_clock::now() < deadline) {}
					if (*static_cast<volatile long long*>(stamps) == 0) { hipDeviceSynchronize(); hipHostFree(stamps); drop_extras(); err = "side-stream calibration: the occupant kernel did not start within 2 s"; return false; }
				}
				std::this_thread::sleep_for(std::chrono::microseconds(r == 0 ? 0 : 200));
				const auto t0 = std::chrono::steady_clock::now();
				for (int k = 0; k < kBurst; ++k) hipLaunchKernelGGL(dtrl_stamp, dim3(96), dim3(256), 0, cand[c], stamps + 1);
				if (!Check(hipStreamSynchronize(cand[c]), "side-stream calibration")) { err = err_; hipHostFree(stamps); drop_extras(); return false; }
				const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
				if (r > 0) worst[c] = std::max(worst[c], static_cast<long long>(us * 100.0));
			}
		}
		hipDeviceSynchronize();
		hipHostFree(stamps);
		std::vector<int> ord(cand.size());
		for (size_t c = 0; c < cand.size(); ++c) ord[c] = static_cast<int>(c);
		std::stable_sort(ord.begin(), ord.end(), [&](int a, int b) { return worst[a] < worst[b]; });
		side_delay_us_.clear();
		for (int c : ord) side_delay_us_.push_back(worst[c] / 100.0);
		// the drain stream (kNumStreams - 1) gets the quickest, then the two side streams, then the other plain engine streams
		std::vector<hipStream_t> pick;
		for (int c : ord) pick.push_back(cand[c]);
		streams_[kNumStreams - 1] = pick[0];
		side_[0] = pick[1]; side_[1] = pick[2];
		for (int i = kNumStreams / 2, k = 3; i < kNumStreams - 1; ++i, ++k) streams_[i] = pick[k];
		calibrated_ = true;
		owned_.assign(cand.begin(), cand.end());      // every candidate stays alive (destroying one could re-seat the others) and is destroyed with the backend
		if (std::getenv("DTRL_HOST_TIMING")) {
			std::fprintf(stderr, "[dtrl] side-stream calibration: 10-kernel burst beside the occupants, us per candidate (sorted):");
			for (double v : side_delay_us_) std::fprintf(stderr, " %.0f", v);
			std::fprintf(stderr, "\n");
		}
		return true;
	}
	void SetReserveCus(int k) override { reserve_arg_ = k; }
	// (without a reservation there is nothing to hand out: measured, a calibrated stream then did no better than any other -- A/B on one box, dog training loop:
	// 10.0 / 10.1 / 11.2 / 10.9 M with it, 11.2 / 11.3 / 11.2 / 11.2 M on the framework's own stream -- the occupants of the calibration leave room the frame kernel does not)
	void* SideStream(int k) override { return (masked_ && calibrated_ && k >= 0 && k < 2) ? static_cast<void*>(side_[k]) : nullptr; }
	double SideStreamDelayUs(int k) override { return (calibrated_ && k >= 0 && k + 1 < static_cast<int>(side_delay_us_.size())) ? side_delay_us_[k + 1] : -1.0; }
	void* Alloc(size_t bytes) override
	{
		void* p = nullptr;
		if (!Check(hipMalloc(&p, bytes), "hipMalloc")) return nullptr;
		if (!Check(hipMemsetAsync(p, 0, bytes, stream_), "hipMemset")) return nullptr;
		return p;
	}
	void Free(void* p) override { hipFree(p); }
	bool H2D(void* dst, const void* src, size_t n) override { return Check(hipMemcpyAsync(dst, src, n, hipMemcpyHostToDevice, stream_), "hipMemcpy H2D") && Check(hipStreamSynchronize(stream_), "sync"); }
	bool D2HAsync(void* dst, const void* src, size_t n) override { return Check(hipMemcpyAsync(dst, src, n, hipMemcpyDeviceToHost, stream_), "hipMemcpyAsync D2H"); }
	bool H2DAsync(void* dst, const void* src, size_t n) override { return Check(hipMemcpyAsync(dst, src, n, hipMemcpyHostToDevice, stream_), "hipMemcpyAsync H2D"); }
	void* HostStaging(size_t bytes) override { void* p = nullptr; return Check(hipHostMalloc(&p, bytes, hipHostMallocMapped | hipHostMallocCoherent), "hipHostMalloc") ? p : nullptr; }
	bool SyncSelected() override { return Check(hipStreamSynchronize(stream_), "hipStreamSynchronize"); }
	void FreeHostStaging(void* p) override { if (p) hipHostFree(p); }
	bool D2H(void* dst, const void* src, size_t n) override { return Check(hipMemcpyAsync(dst, src, n, hipMemcpyDeviceToHost, stream_), "hipMemcpy D2H") && Check(hipStreamSynchronize(stream_), "sync"); }
	bool D2D(void* dst, const void* src, size_t n) override { return Check(hipMemcpyAsync(dst, src, n, hipMemcpyDeviceToDevice, stream_), "hipMemcpy D2D") && Check(hipStreamSynchronize(stream_), "sync"); }
	bool GatherF32(float* dst, const float* src, const int32_t* idx, size_t n) override
	{
		hipLaunchKernelGGL(dtrl_gather_f32, dim3(1024), dim3(256), 0, stream_, dst, src, idx, n);
		return Check(hipGetLastError(), "gather launch") && Check(hipStreamSynchronize(stream_), "sync");
	}
	bool GatherF32On(void* stream, float* dst, const float* src, const int32_t* idx, size_t n) override
	{
		hipStream_t st = stream ? static_cast<hipStream_t>(stream) : stream_;
		hipLaunchKernelGGL(dtrl_gather_f32, dim3(1024), dim3(256), 0, st, dst, src, idx, n);
		return Check(hipGetLastError(), "gather launch") && Check(hipStreamSynchronize(st), "sync");
	}
	bool PackTuples(const DevBuffers& buf, float* block, int block_rows, int64_t env_id_base, int n_envs, const PackScratch& sc) override
	{
		const int grid = std::max(1, std::min<int>(buf.tuple_cap, 2048));
		hipLaunchKernelGGL(dtrl_tuple_order, dim3(1), dim3(kOrderThreads), 0, stream_, buf, block_rows, n_envs, sc.order, sc.hist, sc.meta);
		hipLaunchKernelGGL(dtrl_tuple_pack, dim3(grid), dim3(256), 0, stream_, buf, sc.order, sc.meta, block, env_id_base, sc.rows, sc.flags, sc.env);
		hipLaunchKernelGGL(dtrl_tuple_finish, dim3(grid), dim3(256), 0, stream_, buf, sc.meta, block, sc.rows, sc.flags, sc.env);
		return Check(hipGetLastError(), "tuple pack launch") && Check(hipStreamSynchronize(stream_), "tuple pack");
	}
	// frame marks: MarkFrame records an event behind the frame launch of a group; WaitFrames makes the selected stream wait for the marks of a slot
	bool MarkFrame(int group, int slot) override
	{
		hipEvent_t& ev = marks_[Key(group, slot)];
		if (!ev && !Check(hipEventCreateWithFlags(&ev, hipEventDisableTiming), "hipEventCreate")) return false;
		return Check(hipEventRecord(ev, streams_[group]), "hipEventRecord");
	}
	bool WaitFrames(int slot, int n_groups) override
	{
		for (int g = 0; g < n_groups; ++g) {
			auto it = marks_.find(Key(g, slot));
			if (it != marks_.end() && it->second && !Check(hipStreamWaitEvent(stream_, it->second, 0), "hipStreamWaitEvent")) return false;
		}
		return true;
	}
	bool GatherF32Async(void* stream, float* dst, const float* src, const int32_t* idx, size_t n) override
	{
		hipStream_t st = static_cast<hipStream_t>(stream);
		hipLaunchKernelGGL(dtrl_gather_f32, dim3(1024), dim3(256), 0, st, dst, src, idx, n);
		if (!Check(hipGetLastError(), "gather launch")) return false;
		if (!policy_ready_ && !Check(hipEventCreateWithFlags(&policy_ready_, hipEventDisableTiming), "hipEventCreate")) return false;
		return Check(hipEventRecord(policy_ready_, st), "hipEventRecord");
	}
	bool WaitPolicyReady(int group) override { return !policy_ready_ || Check(hipStreamWaitEvent(streams_[group], policy_ready_, 0), "hipStreamWaitEvent"); }
	bool SyncPolicyReady() override { return !policy_ready_ || Check(hipEventSynchronize(policy_ready_), "hipEventSynchronize"); }
	// latest reader of weight buffer `wbuf` per env group (the double-buffered policy hand-over, Engine::SetPolicyDevice)
	bool MarkWeightReader(int group, int wbuf) override
	{
		hipEvent_t& ev = marks_[Key(group, 8 + wbuf)];
		if (!ev && !Check(hipEventCreateWithFlags(&ev, hipEventDisableTiming), "hipEventCreate")) return false;
		return Check(hipEventRecord(ev, streams_[group]), "hipEventRecord");
	}
	bool WaitWeightReaders(void* stream, int wbuf, int n_groups) override
	{
		hipStream_t st = stream ? static_cast<hipStream_t>(stream) : stream_;
		for (int g = 0; g < n_groups; ++g) {
			auto it = marks_.find(Key(g, 8 + wbuf));
			if (it != marks_.end() && it->second && !Check(hipStreamWaitEvent(st, it->second, 0), "hipStreamWaitEvent")) return false;
		}
		return true;
	}
	bool TerrainBoundary(const DevBuffers& buf, int e0, int n, int mode, const int32_t* env_list) override
	{
		if (n <= 0) return true;
		hipLaunchKernelGGL(dtrl_terrain_boundary, dim3((n + 63) / 64), dim3(64), 0, stream_, buf, e0, n, mode, env_list);
		return Check(hipGetLastError(), "terrain boundary launch");
	}
	bool OrderByCost(const EnvStatus* status, int e0, int n, int32_t* order) override
	{
		if (n <= 0) return true;
		hipLaunchKernelGGL(dtrl_order_by_cost, dim3(1), dim3(kOrderBuckets), 0, stream_, status, e0, n, order);
		return Check(hipGetLastError(), "order launch");
	}
	// dynamic LDS of a fast-path launch: 0 in the shipped library; the workspace itself in a -DDTRL_DYN_LDS experiment build; plus DTRL_LDS_PAD bytes (occupancy experiments)
	static unsigned FastDynLds()
	{
		static const unsigned bytes = []() {
			unsigned b = 0;
#if defined(DTRL_DYN_LDS)
			b = static_cast<unsigned>(sizeof(WSFast));
#endif
			if (const char* e = std::getenv("DTRL_LDS_PAD")) b += static_cast<unsigned>(std::atoi(e));
			return b;
		}();
		return bytes;
	}
	bool Launch(const DevModel* gm, const RunParams& rp, const DevBuffers& buf, int n_envs, int n_steps, real dt, bool frame_end) override
	{
		// only stepping launches are timed (the compact 0-step reset launches would skew the per-frame average)
		const bool timed = n_steps > 0;
		std::pair<hipEvent_t, hipEvent_t> ev{};
		if (timed) {
			if (free_events_.empty()) { hipEventCreate(&ev.first); hipEventCreate(&ev.second); events_.push_back(ev); }
			else { ev = free_events_.back(); free_events_.pop_back(); }
			hipEventRecord(ev.first, stream_);
		}
		// DTRL_KERNEL=ref selects the LDS-phase reference kernel (A/B and bitwise cross-check); default is the fast path
		const char* sel = std::getenv("DTRL_KERNEL");
		const bool use_ref = sel && std::strcmp(sel, "ref") == 0;
		if (!use_ref && buf.model_topo == TopoDog::kId)
			hipLaunchKernelGGL(dtrl_frame_kernel_fast<TopoDog>, dim3(n_envs), dim3(kGroup), FastDynLds(), stream_, gm, rp, buf, n_envs, n_steps, dt, frame_end ? 1 : 0);
		else if (!use_ref && buf.model_topo == TopoRaptor::kId)
			hipLaunchKernelGGL(dtrl_frame_kernel_fast<TopoRaptor>, dim3(n_envs), dim3(kGroup), FastDynLds(), stream_, gm, rp, buf, n_envs, n_steps, dt, frame_end ? 1 : 0);
		else
			hipLaunchKernelGGL(dtrl_frame_kernel, dim3(n_envs), dim3(kGroup), 0, stream_, gm, rp, buf, n_envs, n_steps, dt, frame_end ? 1 : 0);
		if (timed) {
			hipEventRecord(ev.second, stream_); pending_.push_back(ev);
			// a long run never asks for the timing: fold finished pairs into the running sum so the event pool stays bounded
			if (pending_.size() > kMaxPendingEvents) FoldFinished(false);
		}
		return Check(hipGetLastError(), "kernel launch");
	}
	bool Sync() override { bool ok = true; for (hipStream_t st : streams_) ok = Check(hipStreamSynchronize(st), "hipStreamSynchronize") && ok; return ok; }
	int NumStreams() const override { return kNumStreams; }
	void SelectStream(int sid) override { stream_ = streams_[(sid >= 0 && sid < kNumStreams) ? sid : 0]; }
	bool StreamIdle(int sid) override { return hipStreamQuery(streams_[(sid >= 0 && sid < kNumStreams) ? sid : 0]) == hipSuccess; }
	void KernelTime(double* avg_ms, int64_t* launches) override
	{
		for (hipStream_t st : streams_) hipStreamSynchronize(st);
		FoldFinished(true);
		if (avg_ms) *avg_ms = time_n_ > 0 ? time_sum_ms_ / time_n_ : 0.0;
		if (launches) *launches = time_n_;
		time_sum_ms_ = 0; time_n_ = 0;
	}
	const char* Name() const override { return "hip"; }
private:
	// move the event pairs whose launch has completed (all of them when `all`: the streams were synchronised) into the running sum
	void FoldFinished(bool all)
	{
		size_t keep = 0;
		for (size_t i = 0; i < pending_.size(); ++i) {
			auto& ev = pending_[i];
			if (all || hipEventQuery(ev.second) == hipSuccess) {
				float ms = 0;
				if (hipEventElapsedTime(&ms, ev.first, ev.second) == hipSuccess) { time_sum_ms_ += ms; ++time_n_; }
				free_events_.push_back(ev);
			} else pending_[keep++] = ev;
		}
		pending_.resize(keep);
	}
	static constexpr size_t kMaxPendingEvents = 64;
	double time_sum_ms_ = 0; int64_t time_n_ = 0;
	bool Check(hipError_t e, const char* what) { if (e == hipSuccess) return true; err_ = std::string(what) + ": " + hipGetErrorString(e); return false; }
	static constexpr int kNumStreams = 8;
	int reserve_arg_ = -1;             // -reserve_cus= (else DTRL_RESERVE_CUS)
	bool masked_ = false, calibrated_ = false;
	hipStream_t side_[2] = {nullptr, nullptr};
	std::vector<hipStream_t> extra_, owned_;
	std::vector<double> side_delay_us_;
	std::vector<hipStream_t> streams_;
	hipStream_t stream_ = nullptr;   // the selected one
	std::vector<std::pair<hipEvent_t, hipEvent_t>> events_, free_events_, pending_;
	static int Key(int group, int slot) { return group * 16 + slot; }
	hipEvent_t policy_ready_ = nullptr;   // behind the latest asynchronous policy gather (GatherF32Async)
	std::map<int, hipEvent_t> marks_;   // (env group, tuple ring) -> event behind the group's latest frame launch that wrote that ring
};

Backend* MakeBackend() { return new HipBackend(); }

}  // namespace dtrl
