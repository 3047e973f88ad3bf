// dtrl_engine.h -- batch engine: owns the device slabs, the per-env ground windows and the frame loop.
// The backend interface isolates the HIP runtime (dtrl_backend_hip.hip) from the engine logic so that the same logic can
// be unit-tested on a CPU-only box against the lane-loop build of the kernel math (tests/emul/, tests only).
#pragma once
#include "dtrl_host.h"
#include "dtrl_kernel.h"
#include <string>
#include <vector>

namespace dtrl {

// scratch of the packed tuple drain (device memory): order [cap], hist [n_envs + 1], meta [4], carry staging rows [cap][W] / flags [cap] / env [cap]
struct PackScratch { int32_t* order = nullptr; int32_t* hist = nullptr; int32_t* meta = nullptr; float* rows = nullptr; uint32_t* flags = nullptr; int32_t* env = nullptr; };

class Backend {
public:
	virtual ~Backend() {}
	virtual bool Init(int device_id, std::string& err) = 0;
	// -reserve_cus= k (before Init): keep k compute units per XCD out of the frame launches (HIP backend; see dtrl_side_stream in include/dtrl.h)
	virtual void SetReserveCus(int) {}
	// k = 0, 1: a stream (hipStream_t) whose kernels start on the reserved units while frame launches are in flight; nullptr without a reservation
	virtual void* SideStream(int) { return nullptr; }
	virtual double SideStreamDelayUs(int) { return -1.0; }
	virtual void* Alloc(size_t bytes) = 0;   // zero-filled
	virtual void Free(void* p) = 0;
	virtual bool H2D(void* dst, const void* src, size_t n) = 0;
	// stream-ordered upload without a host sync; src must come from HostStaging() and stay untouched until the next Sync()/D2H()
	virtual bool H2DAsync(void* dst, const void* src, size_t n) = 0;
	// stream-ordered download into HostStaging() memory without a host sync (valid after SyncSelected())
	virtual bool D2HAsync(void* dst, const void* src, size_t n) = 0;
	virtual void* HostStaging(size_t bytes) = 0;   // page-locked host memory the device can address directly through the same pointer (coherent; plain malloc in the test backend)
	virtual bool SyncSelected() = 0;               // wait for the selected stream
	virtual void FreeHostStaging(void* p) = 0;
	virtual bool D2H(void* dst, const void* src, size_t n) = 0;
	virtual bool D2D(void* dst, const void* src, size_t n) = 0;   // device-to-device on the selected stream, synchronised before returning
	// dst[i] = idx[i] >= 0 ? src[idx[i]] : 0 for i < n, all device pointers (policy hand-over without a host round trip); synchronised
	virtual bool GatherF32(float* dst, const float* src, const int32_t* idx, size_t n) = 0;
	// the same on a stream of the caller's (a hipStream_t; nullptr = the selected stream), synchronised: work the caller has queued there comes first
	virtual bool GatherF32On(void* stream, float* dst, const float* src, const int32_t* idx, size_t n) { (void)stream; return GatherF32(dst, src, idx, n); }
	// -terrain_gen= device: the frame-boundary terrain work of envs [e0, e0 + n) (or of env_list[0 .. n) when given), queued on the selected stream
	// (tg_env_boundary, dtrl_terrain_dev.h); mode 0 = after a frame, 1 = (re)initialise
	virtual bool TerrainBoundary(const DevBuffers& buf, int e0, int n, int mode, const int32_t* env_list) = 0;
	// order[e0 .. e0 + n) = the envs e0 .. e0 + n - 1 sorted by status[].cost, costliest first (launch order of the group's next frame), on the selected stream
	virtual bool OrderByCost(const EnvStatus* status, int e0, int n, int32_t* order) = 0;
	// pending tuples -> block [block_rows + 1][W + 2] (header row + rows sorted by env id, flag word and global env id as the two extra columns); rows
	// beyond block_rows are carried (moved to the front of the ring, in order), drained / lost-to-a-full-ring totals accumulated in tuple_count[1], [2];
	// device pointers; queued on the selected stream and synchronised. n_envs = number of local envs (tuple_env values are < n_envs).
	virtual bool PackTuples(const DevBuffers& buf, float* block, int block_rows, int64_t env_id_base, int n_envs, const PackScratch& sc) = 0;
	// MarkFrame: remember the point behind everything queued so far on env group `group`'s stream under (group, slot); WaitFrames: the selected stream
	// waits (on the device, not the host) for the marks of `slot` of groups 0 .. n_groups - 1. Lets the tuple drain follow the frame that wrote its ring
	// when the host never synchronised with that frame (-terrain_gen= device). No-ops on a synchronous backend.
	virtual bool MarkFrame(int group, int slot) = 0;
	virtual bool WaitFrames(int slot, int n_groups) = 0;
	// MarkWeightReader: env group `group`'s latest launch reads weight buffer `wbuf` (0 / 1: the double-buffered policy hand-over) -- remember the point behind it;
	// WaitWeightReaders: `stream` (nullptr = the selected stream) waits on the device for every group's latest reader of `wbuf`, so that a gather INTO that
	// buffer cannot overtake a frame kernel still reading it (dtrl_step_poll relaunches, -terrain_gen= device frames the host never waited for). No-ops on a
	// synchronous backend.
	// the gather WITHOUT a host wait, on a stream of the caller's (required): src must stay unchanged until the work queued on that stream has passed this
	// point. Records the "policy ready" point behind it; WaitPolicyReady makes env group `group`'s stream wait for it on the device (the next frame launch
	// of a group must not start before the weights it will read are complete); SyncPolicyReady waits for it on the host. The synchronous test backend
	// performs the gather at once.
	virtual bool GatherF32Async(void* stream, float* dst, const float* src, const int32_t* idx, size_t n) { (void)stream; return GatherF32(dst, src, idx, n); }
	virtual bool WaitPolicyReady(int group) { (void)group; return true; }
	virtual bool SyncPolicyReady() { return true; }
	virtual bool MarkWeightReader(int group, int wbuf) { (void)group; (void)wbuf; return true; }
	virtual bool WaitWeightReaders(void* stream, int wbuf, int n_groups) { (void)stream; (void)wbuf; (void)n_groups; return true; }
	virtual bool Launch(const DevModel* gm, const RunParams& rp, const DevBuffers& buf, int n_envs, int n_steps, real dt, bool frame_end) = 0;
	virtual bool Sync() = 0;                 // all streams
	// work queues: H2D / H2DAsync / D2H / Launch act on the selected stream (0 by default); D2H and H2D synchronise only that stream
	virtual int NumStreams() const = 0;
	virtual void SelectStream(int sid) = 0;
	virtual bool StreamIdle(int sid) = 0;    // everything queued on the stream has completed
	virtual void KernelTime(double* avg_ms, int64_t* launches) = 0;
	virtual const char* Name() const = 0;
	const std::string& error() const { return err_; }
protected:
	std::string err_;
};
Backend* MakeBackend();   // resolved at link time: HIP in libdtrl.so, the lane-loop test backend under tests/emul/

class Engine {
public:
	Engine() {}
	~Engine();
	int Create(const char* const* argv, int argc, int num_envs, int device_id);
	int Reset(const int32_t* env_ids, int n, const uint64_t* seeds);
	int Step(double dt);
	int StepBegin(double dt);
	int StepEnd();
	int StepUpdates(int n);
	int RunFrames(int frames, double dt);
	int SetPolicy(const float* w, size_t n, const double* io, const double* is, const double* oo, const double* os);
	int LoadScaleFile(const char* path);
	int WriteScaleFile(const char* path);
	int SetExplore(int enable, double rate, double temp, double base_rate);
	int SetTerrainLerp(double lerp);
	int DrainTuples(float* rows, uint32_t* flags, int32_t* env_ids, int cap, int* out_n, bool device_dst = false);
	int TupleStats(int64_t* pending, int64_t* drained, int64_t* dropped, int32_t* capacity);
	int GetDistLog(double* dist, int32_t* env_ids, int cap, int* out_n);
	int ResetAvgDist();
	int SetPolicyDevice(const float* w_dev, size_t n, const double* io_dev, const double* is_dev, const double* oo_dev, const double* os_dev, void* stream = nullptr);
	int SetPolicyDeviceAsync(const float* w_dev, size_t n, void* stream);
	int GetStates(const int32_t* env_ids, int n, std::vector<EnvState>& out);
	int SetPoseVel(const int32_t* env_ids, int n, const double* q, const double* qd);
	int GetContactCache(const int32_t* env_ids, int n, int32_t* count, int32_t* ids, double* lambda);
	int SetContactCache(const int32_t* env_ids, int n, const int32_t* count, const int32_t* ids, const double* lambda);
	int CommandAction(const int32_t* env_ids, int n, const int32_t* action_ids);
	void ApplyPendingPolicy();
	void* SideStream(int k, double* delay_us) { void* s = be_ ? be_->SideStream(k) : nullptr; if (delay_us) *delay_us = be_ ? be_->SideStreamDelayUs(k) : -1.0; return s; }
	int AddPerturb(const int32_t* env_ids, int n, const int32_t* link, const double* local_pos, const double* force, const double* duration);
	int ApplyRandForce(const int32_t* env_ids, int n, uint64_t seed);
	int GetPoliState(const int32_t* env_ids, int n, double* s);
	int GetPolicyOutput(const int32_t* env_ids, int n, double* y);
	int GroundWindowRec(int env, int32_t* w2, double* min_x2, double* max_x2, float* h0, float* h1, int cap, int64_t* num_builds);
	int DrainTuplesPacked(float* block_dev, int block_rows, int* out_n);
	int SetTuplePipelining(bool on);
	int StepEndBegin(double dt);
	int StepPoll(double dt, int* relaunched);
	int SampleGround(int env, int n, const double* x, double* h, int32_t* seg, int32_t* oi, int32_t* oj);
	int EvalStats(double* avg_dist, int64_t* episodes, int64_t* cycles, int64_t* resets);
	int KernelTime(double* avg_ms, int64_t* launches);
	int ProfileSections(unsigned long long* out, int cap);
	int ProfileEnv(int section, unsigned long long* out, int cap);   // summed s_memtime ticks per kernel section (DTRL_PROFILE builds)

	const ScenarioConfig& cfg() const { return cfg_; }
	int num_envs() const { return n_; }
	int S() const { return S_; }
	int A() const { return A_; }
	const std::string& error() const { return err_; }
	void set_error(const std::string& e) { err_ = e; }

private:
	int Fail(int code, const std::string& msg) { err_ = msg; return code; }
	int HostFrameWork(int group);
	int FoldTupleTotals();
	int DeviceFrameWork(int group);   // -terrain_gen= device: the same frame-boundary work queued as device kernels, no host sync
	int DrainDeviceDistLog();
	int UploadTerrainCfg(const double* params);
	int ApplyResets(const std::vector<int32_t>& ids, int group);
	int LaunchGroup(int group, int n_steps, double dt_step, bool frame_end);
	// env groups: contiguous env ranges, each with its own stream, launch order and staging slices. Envs are independent, so a group
	// starts its next frame as soon as ITS host work is done instead of waiting for the slowest wavefront of the whole batch
	struct Group { int e0 = 0, n = 0; };
	std::vector<Group> groups_;
	int32_t* d_env_list_ = nullptr;
	int32_t* d_order_ = nullptr;         // launch order of the full-batch frame launches (costliest env first)
	std::vector<int32_t> reset_ids_;
	bool UploadGround(int env);
	bool FetchGroundRec(int env);
	int EnvIndex(const int32_t* env_ids, int i) const { return env_ids ? env_ids[i] : i; }

	ScenarioConfig cfg_;
	Backend* be_ = nullptr;
	int n_ = 0, S_ = 0, A_ = 0, W_ = 0;
	bool policy_set_ = false;
	std::vector<char> early_; bool early_any_ = false;   // env groups dtrl_step_poll has relaunched ahead of the next dtrl_step_end_begin
	float* weights_alt_ = nullptr; const float* weights_buf0_ = nullptr; bool policy_flip_pending_ = false;
	std::vector<char> policy_wait_;   // per env group: its next launch waits (on the device) for the asynchronous hand-over's gather   // SetPolicyDevice during a frame: gathered here, switched in with the next launch
	bool step_pending_ = false;
	DevModel* d_model_ = nullptr;
	DevBuffers buf_{};
	std::vector<void*> allocs_;
	std::vector<GroundWindow> grounds_;
	EnvStatus* status_ = nullptr;   // page-locked; in host terrain mode the frame kernel writes it directly (zero_copy_), else the per-frame read-back lands here
	int32_t* stage_slot_ = nullptr;  // page-locked [n]: slot + 1 of the env's regenerated terrain record in pin_recs_, 0 = none (cleared by the env's wavefront)
	bool zero_copy_ = false;        // host terrain mode: status, launch order, reset lists and terrain records cross the boundary without a copy (dtrl_engine.cpp Init)
	GroundRec tmp_rec_;
	TerrainCfg* d_tcfg_ = nullptr;
	PackScratch pack_;                  // allocated by the first packed drain
	static constexpr int kDistRingCap = 1 << 20;
	bool D2HReal(double* dst, const real* src, size_t n);
	std::vector<double> in_off_, in_scale_, out_off_, out_scale_;   // host copies of the policy normalisers (identity until set)
	int UploadNormalizers();
	void BuildRelayoutMap(std::vector<int32_t>& map) const;
	std::vector<int32_t> relayout_;
	// page-locked staging for the per-frame uploads (terrain records, launch order, reset list): the copies are queued on the
	// stream without a host sync; the arena is recycled after the next frame's status read-back (a stream sync)
	char* pin_drain_ = nullptr; size_t pin_drain_bytes_ = 0;   // page-locked staging of dtrl_drain_tuples (count word + rows + flags + env ids)
	GroundRec* pin_recs_ = nullptr;
	int32_t* pin_order_ = nullptr; int32_t* pin_ids_ = nullptr;
	std::vector<int32_t> bucket_;
	std::vector<std::pair<int32_t, double>> dist_log_;   // (env, distance) of every recorded poli_eval episode, in completion order
	int64_t tuples_drained_ = 0, tuples_dropped_ = 0;
	// tuple pipelining (dtrl_set_tuple_pipelining): two tuple rings; every dtrl_step_begin switches the ring the kernels write, so the frame that
	// has just ended can be drained on its own stream while the next frame already runs
	struct TupleRing { float* rows = nullptr; uint32_t* flags = nullptr; int32_t* env = nullptr; int32_t* count = nullptr; };
	TupleRing ring_[2];
	std::vector<void*> host_allocs_;   // page-locked ring storage (-tuple_ring= host)
	bool AllocRing(TupleRing& r);
	bool RingRead(void* dst, const void* src, size_t n);
	bool RingWrite(void* dst, const void* src, size_t n);
	int wr_ring_ = 0;
	bool tuple_pipelining_ = false;
	void UseRing(DevBuffers& b, int r) const { b.tuple_rows = ring_[r].rows; b.tuple_flags = ring_[r].flags; b.tuple_env = ring_[r].env; b.tuple_count = ring_[r].count; }
	int DrainRing() const { return (tuple_pipelining_ && step_pending_) ? (wr_ring_ ^ 1) : wr_ring_; }   // the ring no kernel is writing
	int DrainFail(); bool drain_order_error_ = false;
	bool DrainSync();   // make the drain ring's contents final: all streams, or -- while a pipelined frame runs -- only the drain stream
	int PendingTuples(int32_t* stored, int32_t* overflow);
	std::vector<int32_t> work_;
	int32_t* d_relayout_ = nullptr;   // device weight index -> index into the caller's Caffe-order blob (-1 = padding), built at Create
	std::string err_;
};

}  // namespace dtrl
