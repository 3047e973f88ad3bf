// dtrl_backend_emul.cpp -- TEST-ONLY backend: runs the very same lane-phase kernel source (dtrl_kernel.h) with the lane
// loop expanded on the host, so tests can check the kernel math and the engine's host logic on a box without a GPU.
// It is linked ONLY into libdtrl_emul.so, which nothing in the product loads (deepterrainrl_amd/__init__.py loads
// libdtrl.so and raises if the HIP library or device is missing). It is not a CPU fallback and is never benchmarked.
#include "dtrl_engine.h"
#include "dtrl_terrain_dev.h"
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <new>
#include <thread>
#include <vector>

namespace dtrl {

class EmulBackend : public Backend {
public:
	bool Init(int, std::string&) override { return true; }
	void* Alloc(size_t bytes) override { return std::calloc(1, bytes ? bytes : 1); }
	void Free(void* p) override { std::free(p); }
	bool H2D(void* dst, const void* src, size_t n) override { std::memcpy(dst, src, n); return true; }
	bool H2DAsync(void* dst, const void* src, size_t n) override { std::memcpy(dst, src, n); return true; }
	bool D2HAsync(void* dst, const void* src, size_t n) override { std::memcpy(dst, src, n); return true; }
	void* HostStaging(size_t bytes) override { return std::calloc(1, bytes ? bytes : 1); }
	void FreeHostStaging(void* p) override { std::free(p); }
	bool SyncSelected() override { return true; }
	bool D2H(void* dst, const void* src, size_t n) override { std::memcpy(dst, src, n); return true; }
	bool D2D(void* dst, const void* src, size_t n) override { std::memcpy(dst, src, n); return true; }
	bool GatherF32(float* dst, const float* src, const int32_t* idx, size_t n) override { for (size_t i = 0; i < n; ++i) dst[i] = idx[i] >= 0 ? src[idx[i]] : 0.0f; return true; }
	bool TerrainBoundary(const DevBuffers& buf, int e0, int n, int mode, const int32_t* env_list) override
	{
		for (int k = 0; k < n; ++k) { const int e = env_list ? env_list[k] : e0 + k; tg_env_boundary(buf.gr[e], buf.gen[e], buf.status[e], *buf.tcfg, mode, e, buf.dist_ring, buf.dist_count, buf.dist_cap); }
		return true;
	}
	bool OrderByCost(const EnvStatus* status, int e0, int n, int32_t* order) override
	{
		for (int k = 0; k < n; ++k) order[e0 + k] = e0 + k;
		std::stable_sort(order + e0, order + e0 + n, [&](int a, int b) { return (status[a].cost >> 4) > (status[b].cost >> 4); });
		return true;
	}
	bool PackTuples(const DevBuffers& buf, float* block, int block_rows, int64_t env_id_base, int, const PackScratch& sc) override
	{
		const int cnt = buf.tuple_count[0];
		const int n = std::min(cnt, static_cast<int>(buf.tuple_cap));
		const int take = std::min(n, block_rows), carry = n - take;
		const int W = buf.W, stride = W + 2;
		std::vector<int> ord(n);
		for (int k = 0; k < n; ++k) ord[k] = k;
		std::stable_sort(ord.begin(), ord.end(), [&](int a, int b) { return buf.tuple_env[a] < buf.tuple_env[b]; });
		for (int r = 0; r < n; ++r) {
			const int k = ord[r];
			const float* src = buf.tuple_rows + static_cast<size_t>(k) * W;
			if (r < take) {
				float* dst = block + static_cast<size_t>(1 + r) * stride;
				std::memcpy(dst, src, sizeof(float) * W);
				const int32_t fl = static_cast<int32_t>(buf.tuple_flags[k]), id = static_cast<int32_t>(env_id_base + buf.tuple_env[k]);
				std::memcpy(dst + W, &fl, 4); std::memcpy(dst + W + 1, &id, 4);
			} else {
				std::memcpy(sc.rows + static_cast<size_t>(r - take) * W, src, sizeof(float) * W);
				sc.flags[r - take] = buf.tuple_flags[k]; sc.env[r - take] = buf.tuple_env[k];
			}
		}
		for (int r = 0; r < carry; ++r) {
			std::memcpy(buf.tuple_rows + static_cast<size_t>(r) * W, sc.rows + static_cast<size_t>(r) * W, sizeof(float) * W);
			buf.tuple_flags[r] = sc.flags[r]; buf.tuple_env[r] = sc.env[r];
		}
		std::memset(block, 0, sizeof(float) * stride);
		const int32_t hdr[3] = {take, cnt - n, carry};
		std::memcpy(block, hdr, sizeof(hdr));
		buf.tuple_count[1] += take; buf.tuple_count[2] += cnt - n; buf.tuple_count[0] = carry;
		return true;
	}
	bool MarkFrame(int, int) override { return true; }
	bool WaitFrames(int, int) override { return true; }
	bool Launch(const DevModel* gm, const RunParams& rp, const DevBuffers& buf, int n_envs, int n_steps, real dt, bool frame_end) override
	{
		int nt = std::min<int>(n_envs, std::max(1u, std::thread::hardware_concurrency()));
		if (buf.tuple_count && nt > 1 && gm->scenario == kScnExp) nt = 1;   // the tuple cursor is a plain int on the host
		std::vector<std::thread> th;
		for (int t = 0; t < nt; ++t) th.emplace_back([=]() {
			// LDS is not cleared between workgroups on the device: poison the workspace before every env (all-ones bytes = NaN doubles,
			// -1 ints) so that a read of a never-written slot shows up in the CPU tests instead of as a flaky GPU result
			void* mem = ::operator new(sizeof(WSRef));
			for (int e = t; e < n_envs; e += nt) {
				std::memset(mem, 0xFF, sizeof(WSRef));
				WSRef* ws = new (mem) WSRef;
				env_frame<RefPath>(*ws, *gm, rp, buf, buf.env_list ? buf.env_list[e] : e, n_steps, dt, frame_end);
			}
			::operator delete(mem);
		});
		for (auto& x : th) x.join();
		++launches_;
		return true;
	}
	bool Sync() override { return true; }
	int NumStreams() const override { return 8; }
	void SelectStream(int) override {}
	bool StreamIdle(int) override { return true; }
	void KernelTime(double* avg_ms, int64_t* launches) override { if (avg_ms) *avg_ms = 0; if (launches) *launches = launches_; launches_ = 0; }
	const char* Name() const override { return "emul"; }
private:
	int64_t launches_ = 0;
};

Backend* MakeBackend() { return new EmulBackend(); }

}  // namespace dtrl
