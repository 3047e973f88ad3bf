// dtrl_engine.cpp -- batch engine logic (host side of the frame loop). See dtrl_engine.h.
//
// Frame loop = cScenarioExp::Update / cScenarioPoliEval::Update over the whole batch
// (scenarios/ScenarioExp.cpp:83-98, scenarios/ScenarioPoliEval.cpp:110-125):
//   1. one kernel launch advances every env by num_update_steps env-steps and runs the end-of-frame fall logic;
//   2. the host reads back 16 B per env (root x, reset request);
//   3. envs that fell get a fresh terrain window (cScenarioSimChar::ResetGround, scenarios/ScenarioSimChar.cpp:574-583)
//      and are flagged so the next launch performs the character/controller reset on device;
//   4. every other env's window slides if the character came within view distance of its end
//      (cScenarioSimChar::UpdateGround, :564-572). Doing this at frame boundaries instead of every env-step is
//      equivalent because the window is rebuilt 1 m before any sample can reach its end (DESIGN.md "Ground").
#include "dtrl_engine.h"
#include "dtrl_terrain_dev.h"
#include "dtrl_topo.h"
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstdio>
#include "../../include/dtrl.h"
#include <atomic>
#include <condition_variable>
#include <cstddef>
#include <cstring>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>

namespace dtrl {

namespace {
constexpr double kViewDist = 10.0;          // cCharController view distance (sim/TerrainRLCharController.cpp:12)
constexpr double kGroundSpawnOffset = -1.0; // scenarios/ScenarioSimChar.cpp:19
constexpr double kViewPad = 1.0;            // gCharViewDistPad, scenarios/ScenarioSimChar.cpp:18
}

// Persistent host workers for the frame-boundary terrain regeneration: every env owns its RNG stream and its window, so the rebuilds of one
// frame are independent (the reference runs one OS thread per scene, scenarios/ScenarioTrain.cpp:100-115). The pool is process-wide and
// created on first use; a frame with fewer than kMinParallel rebuilds stays on the calling thread.
namespace {
class WorkerPool {
public:
	static WorkerPool& Get() { static WorkerPool p; return p; }
	void ParallelFor(int count, const std::function<void(int)>& fn)
	{
		if (count <= 0) return;
		if (count < kMinParallel || threads_.empty()) { for (int i = 0; i < count; ++i) fn(i); return; }
		std::lock_guard<std::mutex> one_caller(call_m_);   // batches driven from different host threads take turns
		// one Job object per call: function, item count and the item cursor travel together, so a worker that wakes late from an earlier call still
		// holds THAT call's (exhausted) cursor and can neither run this call's function on a stale index nor count an item twice
		auto job = std::make_shared<Job>();
		job->fn = &fn; job->count = count;
		{ std::lock_guard<std::mutex> lk(m_); job_ = job; ++epoch_; }
		cv_.notify_all();
		Run(*job);   // the caller works too
		std::unique_lock<std::mutex> lk(m_);
		cv_done_.wait(lk, [&] { return job->done == job->count; });   // every item has RETURNED (done is bumped after the call): fn may go out of scope
		job_.reset();
	}
	int num_threads() const { return static_cast<int>(threads_.size()) + 1; }
private:
	struct Job { const std::function<void(int)>* fn = nullptr; int count = 0; std::atomic<int> next{0}; int done = 0; };
	static constexpr int kMinParallel = 8;
	WorkerPool()
	{
		// DTRL_HOST_THREADS overrides. Default: half the hardware threads, at most 16, shared between the ranks of this node (one process per GPU:
		// LOCAL_WORLD_SIZE as torch.distributed.run / torchrun export it), so that 8 ranks do not start 8 pools of 16 on the same cores
		int n = 0;
		if (const char* env = std::getenv("DTRL_HOST_THREADS")) n = std::atoi(env);
		else {
			int ranks = 1;
			if (const char* lw = std::getenv("LOCAL_WORLD_SIZE")) ranks = std::max(1, std::atoi(lw));
			n = std::max(1, std::min(16, static_cast<int>(std::thread::hardware_concurrency()) / (2 * ranks)));
		}
		for (int t = 1; t < n; ++t) threads_.emplace_back([this] { Loop(); });
	}
	~WorkerPool()
	{
		{ std::lock_guard<std::mutex> lk(m_); stop_ = true; ++epoch_; }
		cv_.notify_all();
		for (auto& t : threads_) t.join();
	}
	void Run(Job& j)
	{
		int finished = 0;
		for (;;) {
			const int i = j.next.fetch_add(1);
			if (i >= j.count) break;
			(*j.fn)(i); ++finished;
		}
		if (finished) { std::lock_guard<std::mutex> lk(m_); j.done += finished; if (j.done == j.count) cv_done_.notify_all(); }
	}
	void Loop()
	{
		long seen = 0;
		for (;;) {
			std::shared_ptr<Job> j;
			{
				std::unique_lock<std::mutex> lk(m_);
				cv_.wait(lk, [&] { return epoch_ != seen; });
				seen = epoch_;
				if (stop_) return;
				j = job_;   // the generation this wake-up belongs to (or a newer one; never a mix of two)
			}
			if (j) Run(*j);
		}
	}
	std::vector<std::thread> threads_;
	std::mutex m_, call_m_; std::condition_variable cv_, cv_done_;
	std::shared_ptr<Job> job_;
	long epoch_ = 0; bool stop_ = false;
};
}  // namespace

Engine::~Engine()
{
	if (be_) {
		be_->Sync();
		for (void* p : allocs_) be_->Free(p);
		for (void* p : host_allocs_) be_->FreeHostStaging(p);
		be_->FreeHostStaging(pin_drain_); be_->FreeHostStaging(pin_recs_); be_->FreeHostStaging(pin_order_); be_->FreeHostStaging(pin_ids_); be_->FreeHostStaging(status_); be_->FreeHostStaging(stage_slot_);
		delete be_;
	}
}

// number of floats of the device-side weight layout (InnerProduct blobs are padded to 4-input blocks)
static int64_t DevNumParams(const NetDesc& d)
{
	int64_t n = 0; int cin = 1, w = d.n_terrain;
	for (int l = 0; l < 3; ++l) { n += pad4(static_cast<int64_t>(d.conv_ch[l]) * cin * d.conv_k[l]) + pad4(d.conv_ch[l]); cin = d.conv_ch[l]; w = w - d.conv_k[l] + 1; }
	n += fc_dev_size(d.fc_terr, cin * w) + pad4(d.fc_terr);
	n += fc_dev_size(d.fc_trunk, d.fc_terr + d.n_char) + pad4(d.fc_trunk);
	n += (fc_dev_size(d.fc_head, d.fc_trunk) + pad4(d.fc_head)) * (1 + d.n_frags);
	n += fc_dev_size(d.n_frags, d.fc_head) + pad4(d.n_frags) + static_cast<int64_t>(d.n_frags) * (fc_dev_size(d.frag_size, d.fc_head) + pad4(d.frag_size));
	return n;
}

int Engine::Create(const char* const* argv, int argc, int num_envs, int device_id)
{
	if (num_envs <= 0) return Fail(DTRL_ERR_ARG, "num_envs must be positive");
	// optimizer/Main.cpp:19-32: command line first, then the arg file appended (first match wins -> command line overrides)
	ArgParser args(argv, argc);
	std::string arg_file;
	if (args.ParseString("arg_file", arg_file)) {
		std::string root; args.ParseString("data_root", root);
		std::string path = (arg_file.empty() || arg_file[0] == '/' || root.empty()) ? arg_file : root + "/" + arg_file;
		if (!args.AppendArgs(path)) return Fail(DTRL_ERR_IO, "Failed to load args from: " + path);
	}
	if (!LoadScenario(args, cfg_, err_)) return DTRL_ERR_IO;
	n_ = num_envs;
	const DevModel& m = cfg_.model;
	S_ = kNumGroundSamples + (2 * m.L - 1) + 2 * m.L;   // sim/TerrainRLCharController.cpp:308-342
	// sim/BaseControllerMACE.cpp:28-31; sim/BaseControllerCacla.cpp:13-16 (the parameters alone); sim/BaseControllerQ.cpp:12-23 (one-hot over the base actions)
	A_ = (m.ctrl_type == 2) ? m.n_opt : (m.ctrl_type == 0 ? m.n_actions : 1 + m.n_opt);
	W_ = 1 + 2 * S_ + A_;                               // learning/MACETrainer.cpp:373-376

	be_ = MakeBackend();
	{ int reserve = -1; if (args.ParseInt("reserve_cus", reserve)) be_->SetReserveCus(reserve); }   // (else the DTRL_RESERVE_CUS environment variable; include/dtrl.h: dtrl_side_stream)
	if (!be_->Init(device_id, err_)) return DTRL_ERR_NO_DEVICE;

	auto alloc = [&](size_t bytes) -> void* { void* p = be_->Alloc(bytes); if (p) allocs_.push_back(p); return p; };
	d_model_ = static_cast<DevModel*>(alloc(sizeof(DevModel)));
	buf_.st = static_cast<EnvState*>(alloc(sizeof(EnvState) * n_));
	buf_.gr = static_cast<GroundRec*>(alloc(sizeof(GroundRec) * n_));
	buf_.status = static_cast<EnvStatus*>(alloc(sizeof(EnvStatus) * n_));
	buf_.poli_state = static_cast<real*>(alloc(sizeof(real) * S_ * n_));
	buf_.tup_s0 = static_cast<real*>(alloc(sizeof(real) * S_ * n_));
	buf_.tup_a = static_cast<real*>(alloc(sizeof(real) * A_ * n_));
	buf_.S = S_; buf_.A = A_; buf_.W = W_; buf_.model_D = m.D; buf_.model_topo = match_topology(m.parent, m.L);
	// a tuple per env per cycle (~13 frames) at most; room for 2 per env between drains, at least the reference's ring size
	// (-tuple_ring_capacity= overrides; dtrl_tuple_stats reports how many rows were dropped because the ring was full)
	buf_.tuple_cap = std::max(2 * n_, cfg_.tuple_buffer_size);
	if (cfg_.tuple_ring_capacity > 0) buf_.tuple_cap = cfg_.tuple_ring_capacity;
	// (-tuple_ring= host: page-locked host memory the kernels address through the same pointers -- a tuple is 2.4 KB of fire-and-forget stores over the link and
	// one system-scope atomic on the cursor, ~0.6 MB per frame at 4096 dogs; dtrl_drain_tuples then reads the ring without queueing anything on the GPU)
	if (!AllocRing(ring_[0])) return Fail(DTRL_ERR_DEVICE, "tuple ring allocation failed: " + be_->error());
	UseRing(buf_, 0);
	d_env_list_ = static_cast<int32_t*>(alloc(sizeof(int32_t) * n_));
	d_order_ = static_cast<int32_t*>(alloc(sizeof(int32_t) * n_));
	if (cfg_.device_terrain) {
		buf_.gen = static_cast<GroundGen*>(alloc(sizeof(GroundGen) * n_));
		d_tcfg_ = static_cast<TerrainCfg*>(alloc(sizeof(TerrainCfg)));
		buf_.tcfg = d_tcfg_;
		buf_.dist_cap = kDistRingCap;
		buf_.dist_ring = static_cast<DistRec*>(alloc(sizeof(DistRec) * buf_.dist_cap));
		buf_.dist_count = static_cast<int32_t*>(alloc(sizeof(int32_t) * 4));
		if (!buf_.gen || !d_tcfg_ || !buf_.dist_ring || !buf_.dist_count) return Fail(DTRL_ERR_DEVICE, "device allocation failed: " + be_->error());
	}
	pin_recs_ = static_cast<GroundRec*>(be_->HostStaging(sizeof(GroundRec) * n_));
	pin_order_ = static_cast<int32_t*>(be_->HostStaging(sizeof(int32_t) * n_));
	pin_ids_ = static_cast<int32_t*>(be_->HostStaging(sizeof(int32_t) * n_));
	if (!pin_recs_ || !pin_order_ || !pin_ids_) return Fail(DTRL_ERR_DEVICE, "host staging allocation failed: " + be_->error());
	// env groups (one stream each): one group fills the resident-wavefront slots while another's frame-boundary host work runs. Round 1 (boundary work = copies
	// + scatter launches) measured 2 as the optimum (1: 10.1, 2: 13.7, 3: 10.9, 4: 12.6, 8: 6.3 M env-steps/s). Round 4, zero-copy boundary: a group's launch lasts
	// as long as ITS slowest env, so while the slowest envs (characters lying on 13-24 constraint rows) ran the slow Gauss-Seidel path, smaller groups were ahead
	// (2 / 3 / 4 groups: 16.7 / 16.8 / 17.0 M); with every row count on the register path the tail is gone and the count no longer matters for the rollout
	// (17.73 / 17.76 / 17.78 M) while the per-frame loops pay for every extra group's boundary work (exchange leg without a collective 16.9 / 16.6 / 15.8 M):
	// two it stays. DTRL_GROUPS overrides.
	{
		int G = n_ >= 1024 ? 2 : 1;
		if (const char* env = std::getenv("DTRL_GROUPS")) G = std::max(1, std::min(be_->NumStreams(), std::atoi(env)));
		G = std::min(G, n_);
		groups_.clear();
		for (int g = 0; g < G; ++g) { Group gr; gr.e0 = static_cast<int>(static_cast<int64_t>(n_) * g / G); gr.n = static_cast<int>(static_cast<int64_t>(n_) * (g + 1) / G) - gr.e0; groups_.push_back(gr); }
		for (int e = 0; e < n_; ++e) pin_order_[e] = e;
		if (!be_->H2D(d_order_, pin_order_, sizeof(int32_t) * n_)) return Fail(DTRL_ERR_DEVICE, be_->error());
	}
	buf_.prof = static_cast<unsigned long long*>(alloc(sizeof(unsigned long long) * kProfMax * n_));
	if (!d_model_ || !buf_.st || !buf_.gr || !buf_.status || !buf_.poli_state || !buf_.tup_s0 || !buf_.tup_a || !buf_.tuple_rows || !buf_.tuple_flags || !buf_.tuple_env || !buf_.tuple_count)
		return Fail(DTRL_ERR_DEVICE, "device allocation failed: " + be_->error());
	if (cfg_.has_policy_net) {
		const NetDesc& d = cfg_.net;
		buf_.net = d;
		buf_.nn_out = static_cast<real*>(alloc(sizeof(real) * static_cast<size_t>(d.out_size) * n_));
		float* w_dev = static_cast<float*>(alloc(sizeof(float) * static_cast<size_t>(DevNumParams(d))));
		weights_alt_ = static_cast<float*>(alloc(sizeof(float) * static_cast<size_t>(DevNumParams(d))));   // second weight buffer: hand-overs during a frame (SetPolicyDevice)
		weights_buf0_ = w_dev;
		real* io = static_cast<real*>(alloc(sizeof(real) * d.in_size)); real* is = static_cast<real*>(alloc(sizeof(real) * d.in_size));
		real* oo = static_cast<real*>(alloc(sizeof(real) * d.out_size)); real* os = static_cast<real*>(alloc(sizeof(real) * d.out_size));
		if (!buf_.nn_out || !w_dev || !weights_alt_ || !io || !is || !oo || !os) return Fail(DTRL_ERR_DEVICE, "device allocation failed: " + be_->error());
		{ std::vector<real> zeros(static_cast<size_t>(d.out_size) * n_, 0.0); if (!be_->H2D(buf_.nn_out, zeros.data(), sizeof(real) * zeros.size())) return Fail(DTRL_ERR_DEVICE, be_->error()); }   // dtrl_get_policy_output before the first forward
		buf_.weights = w_dev; buf_.in_off = io; buf_.in_scale = is; buf_.out_off = oo; buf_.out_scale = os;
		cfg_.model.has_net = 1;
		BuildRelayoutMap(relayout_);
		d_relayout_ = static_cast<int32_t*>(alloc(sizeof(int32_t) * relayout_.size()));
		if (!d_relayout_ || !be_->H2D(d_relayout_, relayout_.data(), sizeof(int32_t) * relayout_.size())) return Fail(DTRL_ERR_DEVICE, "device allocation failed: " + be_->error());
	}
	if (!be_->H2D(d_model_, &cfg_.model, sizeof(DevModel))) return Fail(DTRL_ERR_DEVICE, be_->error());

	// per-env grounds: cGroundVar2D seeded per env (terrain_seed + global env id) so a trajectory is shard-invariant
	grounds_.resize(n_);
	status_ = static_cast<EnvStatus*>(be_->HostStaging(sizeof(EnvStatus) * n_));
	if (!status_) return Fail(DTRL_ERR_DEVICE, "host staging allocation failed: " + be_->error());
	// host terrain mode: no copy at the frame boundary. The frame kernel writes its 24-byte status records straight into this page-locked host array,
	// and the launch order, the reset lists and the regenerated terrain records are read by the kernels from the page-locked arrays the host filled
	// (a copy is a blit kernel that queues behind 2048 resident wavefronts: ~160 us each, three to four per group-frame). In device terrain mode the status
	// stays in device memory, where the boundary kernels read it.
	zero_copy_ = !cfg_.device_terrain;
	if (zero_copy_) {
		buf_.status = status_;
		stage_slot_ = static_cast<int32_t*>(be_->HostStaging(sizeof(int32_t) * n_));
		if (!stage_slot_) return Fail(DTRL_ERR_DEVICE, "host staging allocation failed: " + be_->error());
		std::memset(stage_slot_, 0, sizeof(int32_t) * n_);
		buf_.stage_slot = stage_slot_; buf_.gr_stage = pin_recs_;
	}
	double params[kNumTerrainParams];
	LerpTerrainParams(cfg_, cfg_.terrain_blend, params);
	if (cfg_.device_terrain) {
		// every env's window is built by the GPU from its own counter stream (key: terrain seed + global env id)
		grounds_.clear();
		int rc = UploadTerrainCfg(params);
		if (rc != DTRL_OK) return rc;
		std::vector<GroundGen> gen(n_);
		for (int e = 0; e < n_; ++e) { gen[e].key = terrain_stream_key(cfg_.terrain_seed, cfg_.run.env_id_base + e); gen[e].ctr = 0; gen[e].builds = 0; gen[e].overflow = 0; }
		if (!be_->H2D(buf_.gen, gen.data(), sizeof(GroundGen) * n_)) return Fail(DTRL_ERR_DEVICE, be_->error());
		if (!be_->TerrainBoundary(buf_, 0, n_, 1, nullptr) || !be_->Sync()) return Fail(DTRL_ERR_DEVICE, be_->error());
	} else {
	std::vector<GroundRec> recs(n_);
	for (int e = 0; e < n_; ++e) {
		GroundWindow& g = grounds_[e];
		g.Configure(cfg_.terrain_type, params, cfg_.model.world_scale, 2 * kViewDist);
		g.SeedRand(static_cast<unsigned long>(cfg_.terrain_seed + static_cast<uint64_t>(cfg_.run.env_id_base) + e));
		g.InitSegments(-kViewDist + kGroundSpawnOffset, kViewDist + kGroundSpawnOffset);   // scenarios/ScenarioSimChar.cpp:344-369
		if (!g.FillRecord(recs[e], err_)) return DTRL_ERR_CAPACITY;
	}
	if (!be_->H2D(buf_.gr, recs.data(), sizeof(GroundRec) * n_)) return Fail(DTRL_ERR_DEVICE, be_->error());
	if (stage_slot_) std::memset(stage_slot_, 0, sizeof(int32_t) * n_);
	}
	std::vector<EnvState> st(n_);
	std::memset(st.data(), 0, sizeof(EnvState) * n_);
	for (int e = 0; e < n_; ++e) { st[e].do_init = 1; st[e].cmd_action = -1; st[e].pert_link = -1; }
	if (!be_->H2D(buf_.st, st.data(), sizeof(EnvState) * n_)) return Fail(DTRL_ERR_DEVICE, be_->error());
	// cScenarioSimChar::Init on every env now (a 0-step launch), so getters and dtrl_set_pose_vel see / act on the initial state
	// before the first step, as with the reference's Init()
	if (!be_->Launch(d_model_, cfg_.run, buf_, n_, 0, 0.0, false) || !be_->Sync()) return Fail(DTRL_ERR_DEVICE, be_->error());
	return DTRL_OK;
}

// tmp_rec_ = the env's current ground record: the device copy, or -- host terrain mode -- the regenerated record still waiting in page-locked
// memory for the env's next launch to copy it in (call after Sync())
bool Engine::FetchGroundRec(int env)
{
	if (zero_copy_ && stage_slot_[env] > 0) { std::memcpy(&tmp_rec_, &pin_recs_[stage_slot_[env] - 1], sizeof(GroundRec)); return true; }
	return be_->D2H(&tmp_rec_, &buf_.gr[env], sizeof(GroundRec));
}

bool Engine::UploadGround(int env)
{
	if (!grounds_[env].FillRecord(tmp_rec_, err_)) return false;
	if (zero_copy_) stage_slot_[env] = 0;   // a record waiting for the env's next launch is superseded
	if (!be_->H2D(&buf_.gr[env], &tmp_rec_, sizeof(GroundRec))) { err_ = be_->error(); return false; }
	return true;
}
// device-side half of a reset (cSimCharacter::Reset + controller reset + InitCharacterPos), on the listed envs only:
// a compact 0-step launch, so getters observe the reset state right after Update() as with the reference.
// group >= 0: the group's stream and its slices of the staging / list buffers; group < 0: whole batch on stream 0 (user resets).
int Engine::ApplyResets(const std::vector<int32_t>& ids, int group)
{
	if (ids.empty()) return DTRL_OK;
	const int off = group >= 0 ? groups_[group].e0 : 0;
	std::memcpy(pin_ids_ + off, ids.data(), sizeof(int32_t) * ids.size());
	if (!zero_copy_ && !be_->H2DAsync(d_env_list_ + off, pin_ids_ + off, sizeof(int32_t) * ids.size())) return Fail(DTRL_ERR_DEVICE, be_->error());
	DevBuffers b = buf_;
	b.env_list = (zero_copy_ ? pin_ids_ : d_env_list_) + off;
	b.reset_listed = 1;
	if (!be_->Launch(d_model_, cfg_.run, b, static_cast<int>(ids.size()), 0, 0.0, false)) return Fail(DTRL_ERR_DEVICE, be_->error());
	return DTRL_OK;
}

// DTRL_HOST_TIMING=1: print where the frame-boundary host work goes when the engine is destroyed (diagnosis only)
namespace {
struct HostTiming {
	bool on = std::getenv("DTRL_HOST_TIMING") != nullptr;
	double t_sync = 0, t_loop = 0, t_sort = 0, t_reset = 0, t_launch = 0; long n = 0, regen = 0;
	double d_wait = 0, d_count = 0, d_copy = 0; long d_n = 0;
	~HostTiming() { if (on && d_n) std::fprintf(stderr, "[dtrl host] per tuple drain: wait for the ring's frame %.1f us, count read-back %.1f us, rows + reset %.1f us (n=%ld)\n", 1e6 * d_wait / d_n, 1e6 * d_count / d_n, 1e6 * d_copy / d_n, d_n);
		if (on && n) std::fprintf(stderr, "[dtrl host] per group-frame: status read-back %.1f us, env loop %.1f us (%.2f regen), order %.1f us, resets %.1f us, launch %.1f us (n=%ld)\n", 1e6 * t_sync / n, 1e6 * t_loop / n, double(regen) / n, 1e6 * t_sort / n, 1e6 * t_reset / n, 1e6 * t_launch / n, n); }
} g_ht;
inline double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
}
int Engine::LaunchGroup(int group, int n_steps, double dt_step, bool frame_end)
{
	const Group& g = groups_[group];
	be_->SelectStream(group);
	DevBuffers b = buf_;
	b.env_list = (zero_copy_ ? pin_order_ : d_order_) + g.e0;   // the group's launch order (global env ids), costliest first
	const double lt0 = g_ht.on ? now_s() : 0;
	bool ok = true;
	if (group < static_cast<int>(policy_wait_.size()) && policy_wait_[group]) { ok = be_->WaitPolicyReady(group); policy_wait_[group] = 0; }   // dtrl_set_policy_device_async
	ok = ok && be_->Launch(d_model_, cfg_.run, b, g.n, n_steps, dt_step, frame_end);
	// tuple pipelining: the drain of this frame's ring runs on a stream of its own one frame later and must follow THIS launch on the device -- the host
	// has not necessarily waited for it by then (-terrain_gen= device queues the boundary work and the next frame without a sync)
	if (ok && tuple_pipelining_ && n_steps > 0) ok = be_->MarkFrame(group, wr_ring_);
	// the policy hand-over is double-buffered: remember which weight buffer this launch reads, so that a later gather INTO it can wait for the launch on the device
	if (ok && weights_alt_ && n_steps > 0) ok = be_->MarkWeightReader(group, b.weights == weights_buf0_ ? 0 : 1);
	if (g_ht.on) g_ht.t_launch += now_s() - lt0;
	be_->SelectStream(0);
	return ok ? DTRL_OK : Fail(DTRL_ERR_DEVICE, be_->error());
}

int Engine::UploadTerrainCfg(const double* params)
{
	TerrainCfg c{};
	c.type = cfg_.terrain_type;
	std::memcpy(c.params, params, sizeof(c.params));
	c.world_scale = cfg_.model.world_scale; c.segment_width = 2 * kViewDist;
	c.view_min = -2; c.view_max = kViewDist + kViewPad;
	c.spawn_min = -kViewDist + kGroundSpawnOffset; c.spawn_max = kViewDist + kGroundSpawnOffset;
	be_->Sync();
	if (!be_->H2D(d_tcfg_, &c, sizeof(c))) return Fail(DTRL_ERR_DEVICE, be_->error());
	return DTRL_OK;
}

// -terrain_gen= device: the frame boundary of one env group as three launches on the group's stream, behind its frame kernel and in front of its next
// one -- the host neither waits nor loops over envs:
//   dtrl_terrain_boundary  fresh windows for the envs that fell, slid windows where the character got close to an edge, episode distances logged
//   dtrl_order_by_cost     launch order of the next frame (costliest wavefronts first)
//   0-step frame launch    the device half of the reset, taken by the envs that fell, skipped by the others (reset_listed = 2)
int Engine::DeviceFrameWork(int group)
{
	const Group& grp = groups_[group];
	be_->SelectStream(group);
	struct Restore { Backend* b; ~Restore() { b->SelectStream(0); } } restore{be_};
	DevBuffers b = buf_;
	b.env_list = nullptr; b.reset_listed = 2;
	if (!be_->TerrainBoundary(buf_, grp.e0, grp.n, 0, nullptr) || !be_->OrderByCost(buf_.status, grp.e0, grp.n, d_order_)) return Fail(DTRL_ERR_DEVICE, be_->error());
	b.env_list = d_order_ + grp.e0;
	if (!be_->Launch(d_model_, cfg_.run, b, grp.n, 0, 0.0, false)) return Fail(DTRL_ERR_DEVICE, be_->error());
	return DTRL_OK;
}
// device distance ring -> dist_log_ (completion order inside the ring is whatever the atomics produced; GetDistLog groups by env, and one env
// finishes at most one episode per frame, so the per-env time order is kept by draining between frames or by the ring order of separate launches)
int Engine::DrainDeviceDistLog()
{
	if (!cfg_.device_terrain) return DTRL_OK;
	be_->Sync();
	int32_t cnt = 0;
	if (!be_->D2H(&cnt, buf_.dist_count, sizeof(cnt))) return Fail(DTRL_ERR_DEVICE, be_->error());
	const bool overflow = cnt > buf_.dist_cap;   // the ring keeps the first dist_cap records of the interval; the rest were counted, not stored
	if (overflow) cnt = buf_.dist_cap;
	if (cnt > 0) {
		std::vector<DistRec> tmp(cnt);
		if (!be_->D2H(tmp.data(), buf_.dist_ring, sizeof(DistRec) * cnt)) return Fail(DTRL_ERR_DEVICE, be_->error());
		for (const DistRec& r : tmp) dist_log_.emplace_back(r.env, r.dist);
		const int32_t zero = 0;
		if (!be_->H2D(buf_.dist_count, &zero, sizeof(zero))) return Fail(DTRL_ERR_DEVICE, be_->error());
	}
	if (overflow) return Fail(DTRL_ERR_CAPACITY, "episode distance ring overflowed (records beyond its capacity were dropped): call dtrl_get_dist_log more often");
	return DTRL_OK;
}

// frame-boundary host work of one env group, on the group's stream
int Engine::HostFrameWork(int group)
{
	if (cfg_.device_terrain) return DeviceFrameWork(group);
	const double ht0 = g_ht.on ? now_s() : 0;
	const Group& grp = groups_[group];
	const int e0 = grp.e0, e1 = grp.e0 + grp.n;
	be_->SelectStream(group);
	struct Restore { Backend* b; ~Restore() { b->SelectStream(0); } } restore{be_};
	// the status read-back synchronises the group's stream: its frame kernel and every upload queued during its previous frame
	// have completed, so its slice of the staging arena can be reused from the start
	if (!be_->SyncSelected()) return Fail(DTRL_ERR_DEVICE, be_->error());   // (the kernel wrote status_ itself: host terrain mode moves no copies, see Init)
	const double ht1 = g_ht.on ? now_s() : 0;
	reset_ids_.clear();
	work_.clear();
	for (int e = e0; e < e1; ++e) {
		const EnvStatus& s = status_[e];
		if (s.need_reset & 2) dist_log_.emplace_back(e, s.episode_dist);   // cScenarioPoliEval::RecordDistTraveled -> mDistLog
		if (s.need_reset) { reset_ids_.push_back(e); work_.push_back(e); }
		else if (grounds_[e].NeedsUpdate(s.root_x - 2, s.root_x + kViewDist + kViewPad)) work_.push_back(e);
	}
	const int used = static_cast<int>(work_.size());
	if (used > 0) {
		// the rebuilds of one frame are independent (own RNG stream, own window per env): host workers share them; every rebuilt record goes into
		// its slot of the group's page-locked slice, from which the env's own wavefront copies it in when its next launch starts (stage_slot_)
		std::atomic<int> failed{0};
		WorkerPool::Get().ParallelFor(used, [&](int k) {
			const int e = work_[k];
			GroundWindow& g = grounds_[e];
			const EnvStatus& s = status_[e];
			if (s.need_reset) {
				// cScenarioSimChar::ResetGround: Clear + Update around the spawn point -> InitSegments with the SAME rng stream
				g.Clear();
				g.Update(-kViewDist + kGroundSpawnOffset, kViewDist + kGroundSpawnOffset);
			} else g.Update(s.root_x - 2, s.root_x + kViewDist + kViewPad);
			std::string err;
			if (!g.FillRecord(pin_recs_[e0 + k], err)) failed.store(1);
			stage_slot_[e] = e0 + k + 1;   // the env's wavefront copies the record in at the start of its next launch (the reset launch below, or the next frame)
		});
		if (failed.load()) return Fail(DTRL_ERR_CAPACITY, "terrain segment exceeds kSegCap vertices");
	}
	const double ht2 = g_ht.on ? now_s() : 0;
	// longest-processing-time-first: a launch is as long as its slowest wavefront (a stumbling character with ~20 constraint
	// rows per substep costs 3x a running one), so the envs that were costliest last frame are dispatched first.
	// Counting sort on cost / 16 (stable, O(n)).
	constexpr int kBuckets = 1024;
	bucket_.assign(kBuckets + 1, 0);
	auto key = [&](int e) { int k = status_[e].cost >> 4; if (k < 0) k = 0; if (k >= kBuckets) k = kBuckets - 1; return kBuckets - 1 - k; };
	for (int e = e0; e < e1; ++e) ++bucket_[key(e) + 1];
	for (int k = 0; k < kBuckets; ++k) bucket_[k + 1] += bucket_[k];
	for (int e = e0; e < e1; ++e) pin_order_[e0 + bucket_[key(e)]++] = e;
	const double ht3 = g_ht.on ? now_s() : 0;
	const int rc = ApplyResets(reset_ids_, group);
	if (g_ht.on) { const double ht4 = now_s(); g_ht.t_sync += ht1 - ht0; g_ht.t_loop += ht2 - ht1; g_ht.t_sort += ht3 - ht2; g_ht.t_reset += ht4 - ht3; g_ht.regen += used; ++g_ht.n; }
	return rc;
}

int Engine::StepBegin(double dt)
{
	if (step_pending_) return Fail(DTRL_ERR_ARG, early_any_ ? "dtrl_step_begin after dtrl_step_poll relaunched a group: call dtrl_step_end_begin" : "dtrl_step_begin called twice without dtrl_step_end");
	if (dt <= 0) return DTRL_OK;   // cScenarioSimChar::Update returns early (scenarios/ScenarioSimChar.cpp:148-151)
	if (cfg_.model.has_net && !policy_set_) return Fail(DTRL_ERR_ARG, "policy_net was given but dtrl_set_policy has not been called");
	const int steps = cfg_.model.num_update_steps;
	ApplyPendingPolicy();
	if (tuple_pipelining_) { wr_ring_ ^= 1; UseRing(buf_, wr_ring_); }   // this frame's tuples go to the ring that is not being drained
	for (size_t g = 0; g < groups_.size(); ++g) { int rc = LaunchGroup(static_cast<int>(g), steps, dt / steps, true); if (rc != DTRL_OK) return rc; }
	step_pending_ = true;
	return DTRL_OK;
}

int Engine::StepEnd()
{
	if (!step_pending_) return DTRL_OK;
	if (early_any_) return Fail(DTRL_ERR_ARG, "dtrl_step_end after dtrl_step_poll relaunched a group: the groups are a frame apart, call dtrl_step_end_begin first");
	step_pending_ = false;
	for (size_t g = 0; g < groups_.size(); ++g) { int rc = HostFrameWork(static_cast<int>(g)); if (rc != DTRL_OK) return rc; }
	return DTRL_OK;
}

// dtrl_step_end + dtrl_step_begin without the barrier between them: every env group gets its frame-boundary host work and its next launch as soon as
// ITS frame is done (the scheduling of RunFrames, one frame at a time), so one group's stragglers are covered by the other groups' next launches.
// With tuple pipelining on, the new launches write the other tuple ring and the frame that has just ended can be drained when this returns.
int Engine::StepEndBegin(double dt)
{
	if (!step_pending_) return StepBegin(dt);
	if (dt <= 0) return StepEnd();
	const int G = static_cast<int>(groups_.size());
	const int steps = cfg_.model.num_update_steps;
	ApplyPendingPolicy();   // (every group's NEXT launch runs with the weights handed over during the frame that is ending)
	if (tuple_pipelining_) { wr_ring_ ^= 1; UseRing(buf_, wr_ring_); }
	std::vector<char> done(static_cast<size_t>(G), 0);
	int remaining = G;
	for (int c = 0; c < G; ++c) if (c < static_cast<int>(early_.size()) && early_[c]) { done[c] = 1; --remaining; }   // relaunched by dtrl_step_poll already (into the ring just switched to)
	early_.assign(static_cast<size_t>(G), 0); early_any_ = false;
	for (; remaining > 0; --remaining) {
		int g = -1;
		for (int c = 0; c < G; ++c) if (!done[c] && be_->StreamIdle(c)) { g = c; break; }
		if (g < 0) for (int c = 0; c < G; ++c) if (!done[c]) { g = c; break; }
		int rc = HostFrameWork(g);
		if (rc == DTRL_OK) rc = LaunchGroup(g, steps, dt / steps, true);
		if (rc != DTRL_OK) return rc;
		done[g] = 1;
	}
	return DTRL_OK;
}

// Between two dtrl_step_end_begin calls, with tuple pipelining on and the idle ring drained: every env group whose frame has ALREADY ended gets its
// boundary work and its next launch now (into the idle ring) instead of at the next dtrl_step_end_begin -- a caller that spends milliseconds between the two
// calls (a trainer working through the drained tuples) would otherwise leave a finished group's half of the GPU idle until it comes back. Never blocks.
// The next dtrl_step_end_begin handles the remaining groups only. Until then the tuple rings must not be touched (both are being written).
int Engine::StepPoll(double dt, int* relaunched)
{
	if (relaunched) *relaunched = 0;
	if (!step_pending_ || !tuple_pipelining_ || dt <= 0 || cfg_.device_terrain) return DTRL_OK;
	const int G = static_cast<int>(groups_.size());
	if (static_cast<int>(early_.size()) != G) early_.assign(static_cast<size_t>(G), 0);
	if (!early_any_) {
		// the ring the early launches will write must be empty: it is the one the caller has just drained
		DevBuffers o = buf_; UseRing(o, wr_ring_ ^ 1);
		int32_t cnt = 0;
		if (cfg_.tuple_ring_host) cnt = *o.tuple_count;
		else return DTRL_OK;   // (a device ring's cursor cannot be read without queueing a copy behind the frame: the early relaunch is a host-ring feature)
		if (cnt != 0) return DTRL_OK;
	}
	const int steps = cfg_.model.num_update_steps;
	for (int g = 0; g < G; ++g) {
		if (early_[g] || !be_->StreamIdle(g)) continue;
		int rc = HostFrameWork(g);
		if (rc != DTRL_OK) return rc;
		DevBuffers keep = buf_;
		UseRing(buf_, wr_ring_ ^ 1);
		const int ring_now = wr_ring_; wr_ring_ ^= 1;          // (LaunchGroup marks the frame under the ring it writes)
		rc = LaunchGroup(g, steps, dt / steps, true);
		wr_ring_ = ring_now; buf_ = keep;
		if (rc != DTRL_OK) return rc;
		early_[g] = 1; early_any_ = true;
		if (relaunched) ++*relaunched;
	}
	return DTRL_OK;
}

int Engine::Step(double dt)
{
	int rc = StepBegin(dt);
	return rc != DTRL_OK ? rc : StepEnd();
}

int Engine::StepUpdates(int n)
{
	if (n <= 0) return DTRL_OK;
	if (early_any_ || step_pending_) return Fail(DTRL_ERR_ARG, "dtrl_step_updates while a frame is in flight (dtrl_step_begin / dtrl_step_poll): call dtrl_step_end_begin / dtrl_step_end first");
	if (cfg_.model.has_net && !policy_set_) return Fail(DTRL_ERR_ARG, "policy_net was given but dtrl_set_policy has not been called");
	const double dt = (1.0 / 30.0) / cfg_.model.num_update_steps;
	ApplyPendingPolicy();
	for (size_t g = 0; g < groups_.size(); ++g) { int rc = LaunchGroup(static_cast<int>(g), n, dt, false); if (rc != DTRL_OK) return rc; }
	for (size_t g = 0; g < groups_.size(); ++g) { int rc = HostFrameWork(static_cast<int>(g)); if (rc != DTRL_OK) return rc; }
	return DTRL_OK;
}

// `frames` outer frames for every env. The envs are independent, so there is no frame barrier across groups: each group starts its
// next frame as soon as its own frame-boundary host work is done, and the tail of one group's launch (its slowest wavefronts) is
// covered by the other groups' next launches. Same results as `frames` calls of Step(dt).
int Engine::RunFrames(int frames, double dt)
{
	if (frames <= 0 || dt <= 0) return DTRL_OK;
	if (early_any_ || step_pending_) return Fail(DTRL_ERR_ARG, "dtrl_run_frames while a frame is in flight (dtrl_step_begin / dtrl_step_poll): call dtrl_step_end_begin / dtrl_step_end first");
	if (cfg_.model.has_net && !policy_set_) return Fail(DTRL_ERR_ARG, "policy_net was given but dtrl_set_policy has not been called");
	const int G = static_cast<int>(groups_.size());
	const int steps = cfg_.model.num_update_steps;
	ApplyPendingPolicy();
	if (cfg_.device_terrain) {
		// nothing on the host depends on a frame's outcome: queue everything; the streams run ahead of the host by as much as the HIP queues hold
		for (int f = 0; f < frames; ++f) for (int g = 0; g < G; ++g) {
			int rc = LaunchGroup(g, steps, dt / steps, true);
			if (rc == DTRL_OK) rc = DeviceFrameWork(g);
			if (rc != DTRL_OK) return rc;
		}
		return DTRL_OK;
	}
	std::vector<int> done(G, 0);
	for (int g = 0; g < G; ++g) { int rc = LaunchGroup(g, steps, dt / steps, true); if (rc != DTRL_OK) return rc; }
	int next = 0;   // oldest outstanding launch (launch order is round-robin)
	for (int remaining = G * frames; remaining > 0;) {
		// serve whichever group has finished its frame; if none has, block on the oldest launch
		int g = -1;
		for (int k = 0; k < G; ++k) { const int c = (next + k) % G; if (done[c] < frames && be_->StreamIdle(c)) { g = c; break; } }
		if (g < 0) { for (int k = 0; k < G; ++k) { const int c = (next + k) % G; if (done[c] < frames) { g = c; break; } } }
		int rc = HostFrameWork(g);
		if (rc != DTRL_OK) return rc;
		--remaining;
		if (++done[g] < frames) { rc = LaunchGroup(g, steps, dt / steps, true); if (rc != DTRL_OK) return rc; }
		if (g == next) next = (next + 1) % G;
	}
	return DTRL_OK;
}

int Engine::Reset(const int32_t* env_ids, int n, const uint64_t* seeds)
{
	if (env_ids && n < 0) return Fail(DTRL_ERR_ARG, "negative env count");
	be_->Sync();
	const int cnt = env_ids ? n : n_;
	reset_ids_.clear();
	// an env listed twice is reset once (its first seed wins): the reset list feeds buffers sized for num_envs entries
	std::vector<char> seen(static_cast<size_t>(n_), 0);
	for (int i = 0; i < cnt; ++i) {
		int e = EnvIndex(env_ids, i);
		if (e < 0 || e >= n_) return Fail(DTRL_ERR_ARG, "env id out of range");
		if (seen[e]) continue;
		seen[e] = 1;
		if (cfg_.device_terrain) {
			if (seeds) {   // a fresh stream for this env (cGroundVar2D::SeedRand)
				GroundGen gg; gg.key = terrain_stream_key(seeds[i], 0); gg.ctr = 0; gg.builds = 0; gg.overflow = 0;
				if (!be_->H2D(&buf_.gen[e], &gg, sizeof(uint64_t) * 2)) return Fail(DTRL_ERR_DEVICE, be_->error());
			}
			reset_ids_.push_back(e);
			continue;
		}
		GroundWindow& g = grounds_[e];
		if (seeds) g.SeedRand(static_cast<unsigned long>(seeds[i]));
		g.Clear();
		g.Update(-kViewDist + kGroundSpawnOffset, kViewDist + kGroundSpawnOffset);
		if (!UploadGround(e)) return DTRL_ERR_CAPACITY;
		reset_ids_.push_back(e);
	}
	if (cfg_.device_terrain && !reset_ids_.empty()) {
		std::memcpy(pin_ids_, reset_ids_.data(), sizeof(int32_t) * reset_ids_.size());
		if (!be_->H2DAsync(d_env_list_, pin_ids_, sizeof(int32_t) * reset_ids_.size()) || !be_->TerrainBoundary(buf_, 0, static_cast<int>(reset_ids_.size()), 1, d_env_list_)) return Fail(DTRL_ERR_DEVICE, be_->error());
	}
	int rc = ApplyResets(reset_ids_, -1);
	if (rc != DTRL_OK) return rc;
	if (!be_->Sync()) return Fail(DTRL_ERR_DEVICE, be_->error());
	return DTRL_OK;
}

// device weight layout (dtrl_kernel.h conv_tile / fc_partial / fc_layer) as an index map into the caller's blob (Caffe blob order of the deploy net, W then b
// per layer): conv blobs [cout][cin][k] -> [cin][k][cout]; InnerProduct blobs [nout][nin] -> [ceil(nin/4)][nout][4] (zero padded); biases
// unchanged; same blob order. -1 = a padding / zero entry. CACLA actor: the all-zero critic head (val_ip0, val_ip1) is inserted behind the
// trunk so that the device sees a one-fragment MACE net; the boundary keeps the actor's blob order and sizes.
void Engine::BuildRelayoutMap(std::vector<int32_t>& map) const
{
	const NetDesc& d = cfg_.net;
	map.assign(static_cast<size_t>(DevNumParams(d)), -1);
	const size_t n_user = static_cast<size_t>(cfg_.user_num_params);
	const size_t head = static_cast<size_t>(d.fc_head) * d.fc_trunk + d.fc_head + d.fc_head + 1;
	const size_t tail = static_cast<size_t>(d.fc_head) * d.fc_trunk + d.fc_head + static_cast<size_t>(d.frag_size) * d.fc_head + d.frag_size;   // ip2, output
	// index into the MACE-ordered (padded) blob -> index into the caller's blob
	auto user = [&](size_t p) -> int32_t {
		if (!cfg_.actor_only) return static_cast<int32_t>(p);
		if (p < n_user - tail) return static_cast<int32_t>(p);
		if (p < n_user - tail + head) return -1;
		return static_cast<int32_t>(p - head);
	};
	size_t src = 0, dst = 0; int cin = 1, wdt = d.n_terrain;
	for (int l = 0; l < 3; ++l) {
		const int co = d.conv_ch[l], k = d.conv_k[l];
		for (int o = 0; o < co; ++o) for (int c = 0; c < cin; ++c) for (int u = 0; u < k; ++u)
			map[dst + (static_cast<size_t>(c) * k + u) * co + o] = user(src + (static_cast<size_t>(o) * cin + c) * k + u);
		src += static_cast<size_t>(co) * cin * k; dst += static_cast<size_t>(pad4(static_cast<int64_t>(co) * cin * k));
		for (int o = 0; o < co; ++o) map[dst + o] = user(src + o);
		src += co; dst += static_cast<size_t>(pad4(co)); cin = co; wdt = wdt - k + 1;
	}
	auto block = [&](int nout, int nin) {
		for (int o = 0; o < nout; ++o) for (int i = 0; i < nin; ++i)
			map[dst + (static_cast<size_t>(i / 4) * nout + o) * 4 + (i % 4)] = user(src + static_cast<size_t>(o) * nin + i);
		src += static_cast<size_t>(nout) * nin; dst += static_cast<size_t>(fc_dev_size(nout, nin));
		for (int o = 0; o < nout; ++o) map[dst + o] = user(src + o);
		src += nout; dst += static_cast<size_t>(pad4(nout));
	};
	{
		// terr_ip0 consumes conv2's output tile by tile (dtrl_kernel.h nn_eval): input (c, t) of Caffe's channel-major flattening sits at
		// tile * cin * V + c * vt + (t - tile * V), V positions per tile (vt in the last one)
		const int nout = d.fc_terr, V = kConvTile - (d.conv_k[1] - 1) - (d.conv_k[2] - 1);
		for (int o = 0; o < nout; ++o) for (int c = 0; c < cin; ++c) for (int t = 0; t < wdt; ++t) {
			const int tile = t / V, vt = std::min(V, wdt - tile * V);
			const size_t i = static_cast<size_t>(tile) * cin * V + static_cast<size_t>(c) * vt + (t - tile * V);
			map[dst + ((i / 4) * nout + o) * 4 + (i % 4)] = user(src + static_cast<size_t>(o) * cin * wdt + static_cast<size_t>(c) * wdt + t);
		}
		src += static_cast<size_t>(nout) * cin * wdt; dst += static_cast<size_t>(fc_dev_size(nout, cin * wdt));
		for (int o = 0; o < nout; ++o) map[dst + o] = user(src + o);
		src += nout; dst += static_cast<size_t>(pad4(nout));
	}
	block(d.fc_trunk, d.fc_terr + d.n_char);
	block(d.fc_head, d.fc_trunk); block(d.n_frags, d.fc_head);
	for (int f = 0; f < d.n_frags; ++f) { block(d.fc_head, d.fc_trunk); block(d.frag_size, d.fc_head); }
}

int Engine::SetPolicy(const float* w, size_t n, const double* io, const double* is, const double* oo, const double* os)
{
	if (!cfg_.has_policy_net) return Fail(DTRL_ERR_ARG, "no -policy_net= in the arguments: this batch has no network");
	const NetDesc& d = cfg_.net;
	if (!w || n != static_cast<size_t>(cfg_.user_num_params)) return Fail(DTRL_ERR_ARG, "weight count does not match the deploy prototxt");
	be_->Sync();
	if (!policy_wait_.empty()) { if (!be_->SyncPolicyReady()) return Fail(DTRL_ERR_DEVICE, be_->error()); policy_wait_.assign(policy_wait_.size(), 0); }   // (an asynchronous hand-over still on its way: let it land, then supersede it)
	policy_flip_pending_ = false;   // (a deferred device hand-over is superseded)
	std::vector<float> dev_w(relayout_.size());
	for (size_t i = 0; i < relayout_.size(); ++i) dev_w[i] = relayout_[i] >= 0 ? w[relayout_[i]] : 0.0f;
	// CACLA actor: a neutral critic slot in front of the output normalisers (the device net's one zero critic output)
	const int pad = d.out_size - cfg_.user_out_size;
	// cNeuralNet without a scale file: identity normalisation (learning/NeuralNet.cpp:925-933) for every vector passed as NULL
	if (io) in_off_.assign(io, io + d.in_size); else in_off_.assign(d.in_size, 0.0);
	if (is) in_scale_.assign(is, is + d.in_size); else in_scale_.assign(d.in_size, 1.0);
	out_off_.assign(d.out_size, 0.0); out_scale_.assign(d.out_size, 1.0);
	if (oo) std::copy(oo, oo + cfg_.user_out_size, out_off_.begin() + pad);
	if (os) std::copy(os, os + cfg_.user_out_size, out_scale_.begin() + pad);
	if (!be_->H2D(const_cast<float*>(buf_.weights), dev_w.data(), sizeof(float) * dev_w.size())) return Fail(DTRL_ERR_DEVICE, be_->error());
	int rc = UploadNormalizers();
	if (rc != DTRL_OK) return rc;
	policy_set_ = true;
	return DTRL_OK;
}

// dtrl_set_policy with every pointer in DEVICE memory (the trainer's own tensors): no host round trip. Weights are gathered into the
// device layout by a kernel; NULL normalisers keep their current values.
// While a frame is in flight (between dtrl_step_begin and dtrl_step_end) a weights-only hand-over does NOT wait for it: the weights are gathered into a
// second buffer on the drain stream and the kernels switch to it with the next frame launch (the same moment the synchronous path would have taken effect).
void Engine::ApplyPendingPolicy()
{
	if (!policy_flip_pending_) return;
	const float* cur = buf_.weights; buf_.weights = weights_alt_; weights_alt_ = const_cast<float*>(cur);
	policy_flip_pending_ = false;
}
// dtrl_set_policy_device_async: the weights-only hand-over with NO host wait at all. The gather into the second weight buffer is queued on the caller's
// stream (behind whatever produced w_dev there: a broadcast, the trainer's last step), every env group's NEXT launch waits for it on the device and switches
// to the buffer. Valid at any time (frame in flight or not); w_dev must stay unchanged until the caller's stream has passed this point.
int Engine::SetPolicyDeviceAsync(const float* w_dev, size_t n, void* stream)
{
	if (!cfg_.has_policy_net) return Fail(DTRL_ERR_ARG, "no -policy_net= in the arguments: this batch has no network");
	if (!w_dev || n != static_cast<size_t>(cfg_.user_num_params)) return Fail(DTRL_ERR_ARG, "weight count does not match the deploy prototxt");
	if (!policy_set_ || !weights_alt_) return Fail(DTRL_ERR_ARG, "dtrl_set_policy_device_async needs a policy (normalisers) installed by dtrl_set_policy / dtrl_set_policy_device first");
	if (!stream && std::string(be_->Name()) == "hip") return Fail(DTRL_ERR_ARG, "dtrl_set_policy_device_async: a stream (hipStream_t) of the caller's is required");
	// a second asynchronous hand-over before any launch consumed the first: the first gather may still be writing the same buffer from another stream
	if (policy_flip_pending_ && !policy_wait_.empty() && !be_->SyncPolicyReady()) return Fail(DTRL_ERR_DEVICE, be_->error());
	if (!be_->WaitWeightReaders(stream, weights_alt_ == weights_buf0_ ? 0 : 1, static_cast<int>(groups_.size()))) return Fail(DTRL_ERR_DEVICE, be_->error());
	if (!be_->GatherF32Async(stream, weights_alt_, w_dev, d_relayout_, relayout_.size())) return Fail(DTRL_ERR_DEVICE, be_->error());
	policy_flip_pending_ = true;
	policy_wait_.assign(groups_.size(), 1);
	return DTRL_OK;
}
int Engine::SetPolicyDevice(const float* w_dev, size_t n, const double* io_dev, const double* is_dev, const double* oo_dev, const double* os_dev, void* stream)
{
	if (!cfg_.has_policy_net) return Fail(DTRL_ERR_ARG, "no -policy_net= in the arguments: this batch has no network");
	const NetDesc& d = cfg_.net;
	if (!w_dev || n != static_cast<size_t>(cfg_.user_num_params)) return Fail(DTRL_ERR_ARG, "weight count does not match the deploy prototxt");
	if (step_pending_ && policy_set_ && weights_alt_ && !io_dev && !is_dev && !oo_dev && !os_dev) {
		struct Restore { Backend* b; ~Restore() { b->SelectStream(0); } } restore{be_};
		be_->SelectStream(be_->NumStreams() - 1);
		// an ASYNCHRONOUS hand-over (dtrl_set_policy_device_async) may still be gathering into this very buffer on another stream: order behind it, and drop the
		// frame launches' wait on its event -- this gather is the newer one
		if (!policy_wait_.empty()) { if (!be_->SyncPolicyReady()) return Fail(DTRL_ERR_DEVICE, be_->error()); policy_wait_.assign(policy_wait_.size(), 0); }
		// (synchronised: the caller may change w_dev when this returns. On the caller's stream -- the trainer's -- the gather follows the steps queued there and
		// ONE wait covers both; on the drain stream it would wait for a wavefront slot of its own behind the frame in flight)
		// a frame kernel may still be READING the buffer about to be overwritten: a group dtrl_step_poll relaunched ran its launch on the buffer that the
		// flip in the following dtrl_step_end_begin turned into weights_alt_ (that call skips the group without a sync), and in -terrain_gen= device mode no
		// frame is ever waited for on the host. The gather waits on the device for every group's latest reader of this buffer (events of launches that have
		// completed cost nothing; in host terrain mode without dtrl_step_poll they always have).
		if (!be_->WaitWeightReaders(stream, weights_alt_ == weights_buf0_ ? 0 : 1, static_cast<int>(groups_.size()))) return Fail(DTRL_ERR_DEVICE, be_->error());
		if (!be_->GatherF32On(stream, weights_alt_, w_dev, d_relayout_, relayout_.size())) return Fail(DTRL_ERR_DEVICE, be_->error());
		policy_flip_pending_ = true;
		return DTRL_OK;
	}
	be_->Sync();
	if (!policy_wait_.empty()) { if (!be_->SyncPolicyReady()) return Fail(DTRL_ERR_DEVICE, be_->error()); policy_wait_.assign(policy_wait_.size(), 0); }   // an asynchronous hand-over still on its way
	ApplyPendingPolicy();
	if (!be_->GatherF32(const_cast<float*>(buf_.weights), w_dev, d_relayout_, relayout_.size())) return Fail(DTRL_ERR_DEVICE, be_->error());
	if (in_off_.empty()) { in_off_.assign(d.in_size, 0.0); in_scale_.assign(d.in_size, 1.0); out_off_.assign(d.out_size, 0.0); out_scale_.assign(d.out_size, 1.0); int rc = UploadNormalizers(); if (rc != DTRL_OK) return rc; }
	const int pad = d.out_size - cfg_.user_out_size;
	bool ok = true;
	if (sizeof(real) != sizeof(double) && (io_dev || is_dev || oo_dev || os_dev)) return Fail(DTRL_ERR_ARG, "the fp32 build takes device-resident WEIGHTS only; hand the normalisers over through dtrl_set_policy (host doubles)");
	if (io_dev) ok = ok && be_->D2D(const_cast<real*>(buf_.in_off), io_dev, sizeof(real) * d.in_size) && be_->D2H(in_off_.data(), buf_.in_off, sizeof(real) * d.in_size);
	if (is_dev) ok = ok && be_->D2D(const_cast<real*>(buf_.in_scale), is_dev, sizeof(real) * d.in_size) && be_->D2H(in_scale_.data(), buf_.in_scale, sizeof(real) * d.in_size);
	if (oo_dev) ok = ok && be_->D2D(const_cast<real*>(buf_.out_off) + pad, oo_dev, sizeof(real) * cfg_.user_out_size) && be_->D2H(out_off_.data(), buf_.out_off, sizeof(real) * d.out_size);
	if (os_dev) ok = ok && be_->D2D(const_cast<real*>(buf_.out_scale) + pad, os_dev, sizeof(real) * cfg_.user_out_size) && be_->D2H(out_scale_.data(), buf_.out_scale, sizeof(real) * d.out_size);
	if (!ok) return Fail(DTRL_ERR_DEVICE, be_->error());
	policy_set_ = true;
	return DTRL_OK;
}

int Engine::UploadNormalizers()
{
	const NetDesc& d = cfg_.net;
	// (the host copies are double whatever the kernel's arithmetic type: dtrl_types.h `real`)
	const std::vector<real> a(in_off_.begin(), in_off_.end()), b(in_scale_.begin(), in_scale_.end()), c(out_off_.begin(), out_off_.end()), e(out_scale_.begin(), out_scale_.end());
	bool ok = be_->H2D(const_cast<real*>(buf_.in_off), a.data(), sizeof(real) * d.in_size)
		&& be_->H2D(const_cast<real*>(buf_.in_scale), b.data(), sizeof(real) * d.in_size)
		&& be_->H2D(const_cast<real*>(buf_.out_off), c.data(), sizeof(real) * d.out_size)
		&& be_->H2D(const_cast<real*>(buf_.out_scale), e.data(), sizeof(real) * d.out_size);
	return ok ? DTRL_OK : Fail(DTRL_ERR_DEVICE, be_->error());
}

// cNeuralNet::LoadScale, learning/NeuralNet.cpp:137-215
int Engine::LoadScaleFile(const char* path)
{
	if (!cfg_.has_policy_net) return Fail(DTRL_ERR_ARG, "no -policy_net= in the arguments: this batch has no network");
	if (!path) return Fail(DTRL_ERR_ARG, "null path");
	const NetDesc& d = cfg_.net;
	Json root; std::string err;
	if (!Json::parse_file(path, root, err)) return Fail(DTRL_ERR_IO, std::string("Failed to read scale file ") + path + ": " + err);
	if (in_off_.empty()) { in_off_.assign(d.in_size, 0.0); in_scale_.assign(d.in_size, 1.0); out_off_.assign(d.out_size, 0.0); out_scale_.assign(d.out_size, 1.0); }
	struct Slot { const char* key; std::vector<double>* vec; int size; const char* what; };
	const Slot slots[4] = {{"InputOffset", &in_off_, d.in_size, "input offset"}, {"InputScale", &in_scale_, d.in_size, "input scale"},
		{"OutputOffset", &out_off_, cfg_.user_out_size, "output offset"}, {"OutputScale", &out_scale_, cfg_.user_out_size, "output scale"}};
	const int pad = d.out_size - cfg_.user_out_size;   // CACLA actor: 1 (the unused critic slot of the device net), else 0
	std::vector<double> tmp[4];
	for (int k = 0; k < 4; ++k) {
		const Json* j = root.find(slots[k].key);
		if (!j || j->type == Json::kNull) continue;
		if (j->type != Json::kArr) return Fail(DTRL_ERR_IO, std::string(slots[k].key) + " is not an array in " + path);
		if (static_cast<int>(j->arr.size()) != slots[k].size)
			return Fail(DTRL_ERR_IO, std::string("Invalid ") + slots[k].what + " size, expecting " + std::to_string(slots[k].size) + ", but got " + std::to_string(j->arr.size()));
		for (const Json& v : j->arr) tmp[k].push_back(v.num);
	}
	for (int k = 0; k < 4; ++k) if (!tmp[k].empty()) { if (k >= 2 && pad) tmp[k].insert(tmp[k].begin(), pad, k == 2 ? 0.0 : 1.0); *slots[k].vec = tmp[k]; }
	be_->Sync();
	return UploadNormalizers();
}

// cNeuralNet::WriteOffsetScale, learning/NeuralNet.cpp:1182-1205 (cJsonUtil::BuildVectorJson: std::to_string per element)
int Engine::WriteScaleFile(const char* path)
{
	if (!cfg_.has_policy_net) return Fail(DTRL_ERR_ARG, "no -policy_net= in the arguments: this batch has no network");
	const NetDesc& d = cfg_.net;
	if (in_off_.empty()) { in_off_.assign(d.in_size, 0.0); in_scale_.assign(d.in_size, 1.0); out_off_.assign(d.out_size, 0.0); out_scale_.assign(d.out_size, 1.0); }
	FILE* f = path ? std::fopen(path, "w") : nullptr;
	if (!f) return Fail(DTRL_ERR_IO, std::string("Failed to write offset and scale to ") + (path ? path : "(null)"));
	auto vec_json = [](const std::vector<double>& v) { std::string s = "["; for (size_t i = 0; i < v.size(); ++i) { if (i) s += ", "; s += std::to_string(v[i]); } return s + "]"; };
	std::fprintf(f, "{\n\"InputOffset\": %s,\n\"InputScale\": %s,\n\"OutputOffset\": %s,\n\"OutputScale\": %s\n}",
		vec_json(in_off_).c_str(), vec_json(in_scale_).c_str(),
		vec_json(std::vector<double>(out_off_.begin() + (d.out_size - cfg_.user_out_size), out_off_.end())).c_str(),
		vec_json(std::vector<double>(out_scale_.begin() + (d.out_size - cfg_.user_out_size), out_scale_.end())).c_str());
	std::fclose(f);
	return DTRL_OK;
}

int Engine::SetExplore(int enable, double rate, double temp, double base_rate)
{
	cfg_.run.enable_exp = enable ? 1 : 0; cfg_.run.exp_rate = rate; cfg_.run.exp_temp = temp; cfg_.run.exp_base_rate = base_rate;
	return DTRL_OK;
}

int Engine::SetTerrainLerp(double lerp)
{
	double params[kNumTerrainParams];
	LerpTerrainParams(cfg_, lerp, params);
	if (cfg_.device_terrain) return UploadTerrainCfg(params);
	for (GroundWindow& g : grounds_) g.SetParams(params);   // takes effect at the next segment build, as in the reference
	return DTRL_OK;
}

// ---- tuple rings. Without pipelining there is one ring and every drain waits for all streams. With it (SetTuplePipelining) dtrl_step_begin switches
// the ring the kernels write; between dtrl_step_begin and dtrl_step_end the drains below work on the OTHER ring (the frame that has ended), on a stream of
// their own, and do not wait for the frame in flight.
bool Engine::AllocRing(TupleRing& r)
{
	const size_t rb = sizeof(float) * static_cast<size_t>(W_) * buf_.tuple_cap, fb = sizeof(uint32_t) * static_cast<size_t>(buf_.tuple_cap), eb = sizeof(int32_t) * static_cast<size_t>(buf_.tuple_cap), cb = sizeof(int32_t) * 4;
	auto get = [&](size_t bytes) -> void* {
		void* p = cfg_.tuple_ring_host ? be_->HostStaging(bytes) : be_->Alloc(bytes);
		if (!p) return nullptr;
		if (cfg_.tuple_ring_host) { std::memset(p, 0, bytes); host_allocs_.push_back(p); } else allocs_.push_back(p);
		return p;
	};
	r.rows = static_cast<float*>(get(rb)); r.flags = static_cast<uint32_t*>(get(fb)); r.env = static_cast<int32_t*>(get(eb));
	r.count = static_cast<int32_t*>(get(cb));   // [0] ring cursor, [1] / [2] rows drained / dropped by packed drains since the last fold
	return r.rows && r.flags && r.env && r.count;
}
// the ring's words as the host sees them: a plain read / write when the ring is host memory (no kernel is using it: callers hold DrainSync), else a copy
bool Engine::RingRead(void* dst, const void* src, size_t n) { if (cfg_.tuple_ring_host) { std::memcpy(dst, src, n); return true; } return be_->D2H(dst, src, n); }
bool Engine::RingWrite(void* dst, const void* src, size_t n) { if (cfg_.tuple_ring_host) { std::memcpy(dst, src, n); return true; } return be_->H2D(dst, src, n); }
int Engine::SetTuplePipelining(bool on)
{
	if (step_pending_) return Fail(DTRL_ERR_ARG, "dtrl_set_tuple_pipelining between dtrl_step_begin and dtrl_step_end");
	if (!be_->Sync()) return Fail(DTRL_ERR_DEVICE, be_->error());
	if (on && !ring_[1].rows) {
		if (!AllocRing(ring_[1])) return Fail(DTRL_ERR_DEVICE, "tuple ring allocation failed: " + be_->error());
	}
	if (!on && tuple_pipelining_) {
		// back to one ring: whatever waits in the idle ring would be stranded
		DevBuffers o = buf_; UseRing(o, wr_ring_ ^ 1);
		int32_t cnt = 0;
		if (!RingRead(&cnt, o.tuple_count, sizeof(cnt))) return Fail(DTRL_ERR_DEVICE, be_->error());
		if (cnt != 0) return Fail(DTRL_ERR_ARG, "drain the pending tuples before switching tuple pipelining off");
	}
	tuple_pipelining_ = on;
	return DTRL_OK;
}
// DrainSync failed: an API-order error (a drain while dtrl_step_poll's early relaunches hold both rings) is DTRL_ERR_ARG, anything else a device error
int Engine::DrainFail()
{
	if (drain_order_error_) { drain_order_error_ = false; return Fail(DTRL_ERR_ARG, err_); }
	return Fail(DTRL_ERR_DEVICE, be_->error());
}
bool Engine::DrainSync()
{
	if (early_any_) { err_ = "tuple rings are both in use: dtrl_step_poll relaunched a group; call dtrl_step_end_begin first"; drain_order_error_ = true; return false; }
	if (tuple_pipelining_ && step_pending_) {   // the drain ring's frame ended with dtrl_step_end; the frame in flight writes the other ring
		// "ended" is a host-side fact only in host terrain mode (HostFrameWork synchronises the group's stream). In device terrain mode the frame may
		// still be running: the drain stream waits on the device for the mark behind every group's launch of that frame -- a torn row (the cursor is
		// bumped before the row is written) or a cursor zeroed under a running kernel would lose, duplicate or reorder tuples without an error.
		be_->SelectStream(be_->NumStreams() - 1);
		return be_->WaitFrames(DrainRing(), static_cast<int>(groups_.size())) && be_->SyncSelected();
	}
	be_->SelectStream(0);
	return be_->Sync();
}
// rows the kernel could not store because the ring was full are counted, never silently lost: the cursor keeps counting past the capacity
int Engine::PendingTuples(int32_t* stored, int32_t* overflow)
{
	int32_t cnt = 0;
	DevBuffers d = buf_; UseRing(d, DrainRing());
	struct Restore { Backend* b; ~Restore() { b->SelectStream(0); } } restore{be_};
	if (!DrainSync()) return DrainFail();
	if (!RingRead(&cnt, d.tuple_count, sizeof(cnt))) return Fail(DTRL_ERR_DEVICE, be_->error());
	*overflow = cnt > buf_.tuple_cap ? cnt - buf_.tuple_cap : 0;
	*stored = cnt - *overflow;
	return DTRL_OK;
}
int Engine::DrainTuples(float* rows, uint32_t* flags, int32_t* env_ids, int cap, int* out_n, bool device_dst)
{
	if (cap < 0 || !out_n || (cap > 0 && !rows)) return Fail(DTRL_ERR_ARG, "bad arguments");
	DevBuffers d = buf_; UseRing(d, DrainRing());
	struct Restore { Backend* b; ~Restore() { b->SelectStream(0); } } restore{be_};
	// Host destination: everything goes through ONE page-locked staging area with two synchronisations (the count; then rows + flags + ids + the cursor's reset
	// queued together) -- a copy into pageable caller memory is a staged, synchronous affair per call in the runtime (five of them before; with frames in
	// flight each could wait for a wavefront slot).
	if (cfg_.tuple_ring_host && !device_dst) {
		const double ht0 = g_ht.on ? now_s() : 0;
		if (!DrainSync()) return DrainFail();
		const double ht1 = g_ht.on ? now_s() : 0;
		const int32_t cnt_all = *d.tuple_count;
		const int32_t over = cnt_all > buf_.tuple_cap ? cnt_all - buf_.tuple_cap : 0;
		const int n = std::min<int>(cnt_all - over, cap);
		if (n < cnt_all - over) return Fail(DTRL_ERR_CAPACITY, "caller buffer smaller than the number of pending tuples");
		if (n > 0) {
			std::memcpy(rows, d.tuple_rows, sizeof(float) * static_cast<size_t>(W_) * n);
			if (flags) std::memcpy(flags, d.tuple_flags, sizeof(uint32_t) * static_cast<size_t>(n));
			if (env_ids) std::memcpy(env_ids, d.tuple_env, sizeof(int32_t) * static_cast<size_t>(n));
		}
		*d.tuple_count = 0;
		tuples_drained_ += n; tuples_dropped_ += over;
		*out_n = n;
		if (g_ht.on) { g_ht.d_wait += ht1 - ht0; g_ht.d_copy += now_s() - ht1; ++g_ht.d_n; }
		return DTRL_OK;
	}
	if (!pin_drain_) {
		pin_drain_bytes_ = 64 + (sizeof(float) * static_cast<size_t>(W_) + sizeof(uint32_t) + sizeof(int32_t)) * static_cast<size_t>(buf_.tuple_cap);
		pin_drain_ = static_cast<char*>(be_->HostStaging(pin_drain_bytes_));
		if (!pin_drain_) return Fail(DTRL_ERR_DEVICE, "host staging allocation failed: " + be_->error());
		std::memset(pin_drain_, 0, 64);
	}
	int32_t* p_cnt = reinterpret_cast<int32_t*>(pin_drain_); int32_t* p_zero = p_cnt + 1;   // (p_zero stays 0)
	const double ht0 = g_ht.on ? now_s() : 0;
	if (!DrainSync()) return DrainFail();
	const double ht1 = g_ht.on ? now_s() : 0;
	if (cfg_.tuple_ring_host) *p_cnt = *d.tuple_count;      // (device destination, host ring)
	else if (!be_->D2HAsync(p_cnt, d.tuple_count, sizeof(int32_t)) || !be_->SyncSelected()) return Fail(DTRL_ERR_DEVICE, be_->error());
	const double ht2 = g_ht.on ? now_s() : 0;
	const int32_t cnt_all = *p_cnt;
	const int32_t over = cnt_all > buf_.tuple_cap ? cnt_all - buf_.tuple_cap : 0;
	const int32_t cnt = cnt_all - over;
	const int n = std::min<int>(cnt, cap);
	if (n < cnt) return Fail(DTRL_ERR_CAPACITY, "caller buffer smaller than the number of pending tuples");
	const size_t rb = sizeof(float) * static_cast<size_t>(W_) * n, fb = sizeof(uint32_t) * static_cast<size_t>(n), eb = sizeof(int32_t) * static_cast<size_t>(n);
	char* p_rows = pin_drain_ + 64; char* p_flags = p_rows + sizeof(float) * static_cast<size_t>(W_) * buf_.tuple_cap; char* p_env = p_flags + sizeof(uint32_t) * static_cast<size_t>(buf_.tuple_cap);
	bool ok = true;
	if (n > 0) {
		if (device_dst) {
			if (cfg_.tuple_ring_host) ok = be_->H2D(rows, d.tuple_rows, rb) && (!flags || be_->H2D(flags, d.tuple_flags, fb)) && (!env_ids || be_->H2D(env_ids, d.tuple_env, eb));
			else ok = be_->D2D(rows, d.tuple_rows, rb) && (!flags || be_->D2D(flags, d.tuple_flags, fb)) && (!env_ids || be_->D2D(env_ids, d.tuple_env, eb));
		} else {
			ok = be_->D2HAsync(p_rows, d.tuple_rows, rb) && (!flags || be_->D2HAsync(p_flags, d.tuple_flags, fb)) && (!env_ids || be_->D2HAsync(p_env, d.tuple_env, eb));
		}
	}
	if (cfg_.tuple_ring_host) { ok = ok && be_->SyncSelected(); *d.tuple_count = 0; }
	else ok = ok && be_->H2DAsync(d.tuple_count, p_zero, sizeof(int32_t)) && be_->SyncSelected();
	if (!ok) return Fail(DTRL_ERR_DEVICE, be_->error());
	if (n > 0 && !device_dst) {
		std::memcpy(rows, p_rows, rb);
		if (flags) std::memcpy(flags, p_flags, fb);
		if (env_ids) std::memcpy(env_ids, p_env, eb);
	}
	tuples_drained_ += n; tuples_dropped_ += over;
	*out_n = n;
	if (g_ht.on) { g_ht.d_wait += ht1 - ht0; g_ht.d_count += ht2 - ht1; g_ht.d_copy += now_s() - ht2; ++g_ht.d_n; }
	return DTRL_OK;
}
// totals of the device-side (packed) drains -> host counters
int Engine::FoldTupleTotals()
{
	struct Restore { Backend* b; ~Restore() { b->SelectStream(0); } } restore{be_};
	if (!DrainSync()) return DrainFail();
	for (int r = 0; r < 2; ++r) {
		if (!ring_[r].count || (tuple_pipelining_ && step_pending_ && r == wr_ring_)) continue;   // (the ring in flight folds after its frame)
		int32_t c[4] = {0, 0, 0, 0};
		if (!RingRead(c, ring_[r].count, sizeof(c))) return Fail(DTRL_ERR_DEVICE, be_->error());
		if (c[1] || c[2]) {
			tuples_drained_ += c[1]; tuples_dropped_ += c[2];
			const int32_t zero[2] = {0, 0};
			if (!RingWrite(ring_[r].count + 1, zero, sizeof(zero))) return Fail(DTRL_ERR_DEVICE, be_->error());
		}
	}
	return DTRL_OK;
}
int Engine::DrainTuplesPacked(float* block_dev, int block_rows, int* out_n)
{
	if (!block_dev || block_rows < 0) return Fail(DTRL_ERR_ARG, "bad arguments");
	DevBuffers d = buf_; UseRing(d, DrainRing());
	struct Restore { Backend* b; ~Restore() { b->SelectStream(0); } } restore{be_};
	if (!pack_.order) {
		auto alloc = [&](size_t bytes) -> void* { void* p = be_->Alloc(bytes); if (p) allocs_.push_back(p); return p; };
		const size_t cap = static_cast<size_t>(buf_.tuple_cap);
		pack_.order = static_cast<int32_t*>(alloc(sizeof(int32_t) * cap));
		pack_.hist = static_cast<int32_t*>(alloc(sizeof(int32_t) * (static_cast<size_t>(n_) + 1)));
		pack_.meta = static_cast<int32_t*>(alloc(sizeof(int32_t) * 4));
		pack_.rows = static_cast<float*>(alloc(sizeof(float) * static_cast<size_t>(W_) * cap));
		pack_.flags = static_cast<uint32_t*>(alloc(sizeof(uint32_t) * cap));
		pack_.env = static_cast<int32_t*>(alloc(sizeof(int32_t) * cap));
		if (!pack_.order || !pack_.hist || !pack_.meta || !pack_.rows || !pack_.flags || !pack_.env || !be_->Sync()) { pack_ = PackScratch(); return Fail(DTRL_ERR_DEVICE, "device allocation failed: " + be_->error()); }
	}
	if (!DrainSync()) return DrainFail();   // the frame kernels that wrote this ring have finished
	if (!be_->PackTuples(d, block_dev, block_rows, cfg_.run.env_id_base, n_, pack_)) return Fail(DTRL_ERR_DEVICE, be_->error());
	if (out_n) { int32_t n = 0; if (!be_->D2H(&n, block_dev, sizeof(n))) return Fail(DTRL_ERR_DEVICE, be_->error()); *out_n = n; }
	return DTRL_OK;
}
int Engine::TupleStats(int64_t* pending, int64_t* drained, int64_t* dropped, int32_t* capacity)
{
	int32_t cnt = 0, over = 0;
	int rc = FoldTupleTotals();
	if (rc == DTRL_OK) rc = PendingTuples(&cnt, &over);
	if (rc != DTRL_OK) return rc;
	if (pending) *pending = cnt;
	if (drained) *drained = tuples_drained_;
	if (dropped) *dropped = tuples_dropped_ + over;
	if (capacity) *capacity = buf_.tuple_cap;
	return DTRL_OK;
}

// cScenarioPoliEval::GetDistLog for the batch: every recorded episode distance since Init / Clear, grouped by env (the order in which
// cOptScenarioPoliEval::OutputResults walks its pool, optimizer/scenarios/OptScenarioPoliEval.cpp:213-239), episodes of one env in time order
int Engine::GetDistLog(double* dist, int32_t* env_ids, int cap, int* out_n)
{
	if (!out_n || cap < 0) return Fail(DTRL_ERR_ARG, "bad arguments");
	{ int rc = DrainDeviceDistLog(); if (rc != DTRL_OK) return rc; }
	std::vector<std::pair<int32_t, double>> v = dist_log_;
	std::stable_sort(v.begin(), v.end(), [](const std::pair<int32_t, double>& a, const std::pair<int32_t, double>& b) { return a.first < b.first; });
	*out_n = static_cast<int>(v.size());
	for (int i = 0; i < *out_n && i < cap; ++i) { if (dist) dist[i] = v[i].second; if (env_ids) env_ids[i] = v[i].first; }
	return *out_n <= cap || (!dist && !env_ids) ? DTRL_OK : Fail(DTRL_ERR_CAPACITY, "caller buffer smaller than the dist log");
}
// cScenarioPoliEval::ResetAvgDist on every env: mAvgDist = 0, mEpisodeCount = 0 (the dist log and the cycle counters stay)
int Engine::ResetAvgDist()
{
	std::vector<EnvState> st;
	int rc = GetStates(nullptr, n_, st);
	if (rc != DTRL_OK) return rc;
	for (EnvState& s : st) { s.avg_dist = 0; s.num_episodes = 0; }
	if (!be_->H2D(buf_.st, st.data(), sizeof(EnvState) * n_)) return Fail(DTRL_ERR_DEVICE, be_->error());
	return DTRL_OK;
}

int Engine::GetStates(const int32_t* env_ids, int n, std::vector<EnvState>& out)
{
	if (n < 0) return Fail(DTRL_ERR_ARG, "negative env count");
	be_->Sync();
	out.resize(n);
	if (!env_ids) {
		if (n > n_) return Fail(DTRL_ERR_ARG, "env id out of range");
		if (!be_->D2H(out.data(), buf_.st, sizeof(EnvState) * n)) return Fail(DTRL_ERR_DEVICE, be_->error());
		return DTRL_OK;
	}
	for (int i = 0; i < n; ++i) {
		int e = env_ids[i];
		if (e < 0 || e >= n_) return Fail(DTRL_ERR_ARG, "env id out of range");
		if (!be_->D2H(&out[i], &buf_.st[e], sizeof(EnvState))) return Fail(DTRL_ERR_DEVICE, be_->error());
	}
	return DTRL_OK;
}

int Engine::SetPoseVel(const int32_t* env_ids, int n, const double* q, const double* qd)
{
	if (n < 0 || !q || !qd) return Fail(DTRL_ERR_ARG, "bad arguments");
	be_->Sync();
	const int D = cfg_.model.D;
	EnvState st;
	for (int i = 0; i < n; ++i) {
		int e = EnvIndex(env_ids, i);
		if (e < 0 || e >= n_) return Fail(DTRL_ERR_ARG, "env id out of range");
		if (!be_->D2H(&st, &buf_.st[e], sizeof(EnvState))) return Fail(DTRL_ERR_DEVICE, be_->error());
		for (int k = 0; k < D; ++k) { st.q[k] = q[i * D + k]; st.qd[k] = qd[i * D + k]; }
		// a teleported character drops its persistent contact rows (Bullet's refreshContactPoints removes manifold points that moved out of the breaking
		// threshold): cached impulses of the old pose would warm-start whichever sample points happen to touch in the new one. A caller that restores a
		// saved state calls dtrl_set_contact_cache AFTER this (include/dtrl.h)
		st.ws_R = 0;
		for (int k = 0; k < kMaxRows; ++k) { st.ws_id[k] = 0xffff; st.ws_lam[k] = 0.0; }
		if (!be_->H2D(&buf_.st[e], &st, sizeof(EnvState))) return Fail(DTRL_ERR_DEVICE, be_->error());
	}
	return DTRL_OK;
}

// the persistent contact points' identities and applied impulses (EnvState::ws_*; include/dtrl.h dtrl_get_contact_cache)
int Engine::GetContactCache(const int32_t* env_ids, int n, int32_t* count, int32_t* ids, double* lambda)
{
	if (n < 0 || !count || !ids || !lambda) return Fail(DTRL_ERR_ARG, "bad arguments");
	std::vector<EnvState> st; int rc = GetStates(env_ids, n, st);
	if (rc != DTRL_OK) return rc;
	const int cnt = static_cast<int>(st.size());
	for (int i = 0; i < cnt; ++i) {
		count[i] = st[i].ws_R;
		for (int k = 0; k < kMaxRows; ++k) { ids[i * kMaxRows + k] = k < st[i].ws_R ? st[i].ws_id[k] : 0xffff; lambda[i * kMaxRows + k] = k < st[i].ws_R ? st[i].ws_lam[k] : 0.0; }
	}
	return DTRL_OK;
}
int Engine::SetContactCache(const int32_t* env_ids, int n, const int32_t* count, const int32_t* ids, const double* lambda)
{
	if (n < 0 || !count || !ids || !lambda) return Fail(DTRL_ERR_ARG, "bad arguments");
	if (!env_ids && n > n_) return Fail(DTRL_ERR_ARG, "more rows than envs");
	be_->Sync();
	const int cnt = n;                       // env_ids == NULL: envs 0 .. n - 1, like dtrl_get_contact_cache and dtrl_set_pose_vel
	EnvState st;
	for (int i = 0; i < cnt; ++i) {
		const int e = EnvIndex(env_ids, i);
		if (e < 0 || e >= n_) return Fail(DTRL_ERR_ARG, "env id out of range");
		if (count[i] < 0 || count[i] > kMaxRows) return Fail(DTRL_ERR_ARG, "contact cache: row count out of range");
		if (!be_->D2H(&st, &buf_.st[e], sizeof(EnvState))) return Fail(DTRL_ERR_DEVICE, be_->error());
		st.ws_R = count[i];
		for (int k = 0; k < kMaxRows; ++k) {
			const bool on = k < count[i];
			if (on && (ids[i * kMaxRows + k] < 0 || ids[i * kMaxRows + k] > 0xffff)) return Fail(DTRL_ERR_ARG, "contact cache: row id out of range");
			st.ws_id[k] = on ? static_cast<uint16_t>(ids[i * kMaxRows + k]) : 0xffff; st.ws_lam[k] = on ? lambda[i * kMaxRows + k] : 0.0;
		}
		if (!be_->H2D(&buf_.st[e], &st, sizeof(EnvState))) return Fail(DTRL_ERR_DEVICE, be_->error());
	}
	return DTRL_OK;
}

// cCharController::CommandAction (sim/DogController.cpp:309-320, sim/RaptorController.cpp): the base action the controller takes at its next cycle instead
// of asking the policy. The reference keeps a STACK of commands (the latest is served first, older ones at later cycles); the engine keeps the top of
// that stack only -- one pending command per env, a new one replaces it (cScenarioExp::Reset's random first action, scenarios/ScenarioExp.cpp:63-73,
// sits in the same slot).
int Engine::CommandAction(const int32_t* env_ids, int n, const int32_t* action_ids)
{
	if ((env_ids && n < 0) || !action_ids) return Fail(DTRL_ERR_ARG, "bad arguments");
	be_->Sync();
	const int cnt = env_ids ? n : n_;
	EnvState st;
	for (int i = 0; i < cnt; ++i) {
		const int e = EnvIndex(env_ids, i);
		if (e < 0 || e >= n_) return Fail(DTRL_ERR_ARG, "env id out of range");
		if (action_ids[i] < 0 || action_ids[i] >= cfg_.model.n_actions) return Fail(DTRL_ERR_ARG, "action id out of range");   // the reference asserts (and ignores the command in release builds)
		if (!be_->D2H(&st, &buf_.st[e], sizeof(EnvState))) return Fail(DTRL_ERR_DEVICE, be_->error());
		st.cmd_action = action_ids[i];
		if (!be_->H2D(&buf_.st[e], &st, sizeof(EnvState))) return Fail(DTRL_ERR_DEVICE, be_->error());
	}
	return DTRL_OK;
}

// cScenarioSimChar::AddPerturb -> cWorld::AddPerturb (scenarios/ScenarioSimChar.cpp:204-207): a world-frame force on a body part at a
// body-local offset for `duration` seconds of simulated time. Planar characters: the in-plane components. One slot per env.
int Engine::AddPerturb(const int32_t* env_ids, int n, const int32_t* link, const double* local_pos, const double* force, const double* duration)
{
	if (env_ids && n < 0) return Fail(DTRL_ERR_ARG, "negative env count");
	be_->Sync();
	const DevModel& m = cfg_.model;
	EnvState st;
	const int cnt = env_ids ? n : n_;
	for (int i = 0; i < cnt; ++i) {
		const int e = EnvIndex(env_ids, i);
		if (e < 0 || e >= n_) return Fail(DTRL_ERR_ARG, "env id out of range");
		const int l = link[i];
		if (l < 0 || l >= m.L) return Fail(DTRL_ERR_ARG, "perturbation link out of range");
		if (!(duration[i] >= 0)) return Fail(DTRL_ERR_ARG, "perturbation duration must be non-negative");
		if (!be_->D2H(&st, &buf_.st[e], sizeof(EnvState))) return Fail(DTRL_ERR_DEVICE, be_->error());
		// the body frame is the joint frame turned by the body's attach angle: store the offset in the joint frame (the kernel has cos/sin of that)
		const double c = std::cos(m.body_theta[l]), sn = std::sin(m.body_theta[l]);
		const double lx = local_pos ? local_pos[2 * i] : 0.0, ly = local_pos ? local_pos[2 * i + 1] : 0.0;
		st.pert_link = l; st.pert_on = 0;
		st.pert_lp[0] = c * lx - sn * ly; st.pert_lp[1] = sn * lx + c * ly;
		st.pert_f[0] = force[2 * i]; st.pert_f[1] = force[2 * i + 1];
		st.pert_torque = 0; st.pert_time = 0; st.pert_dur = duration[i];
		if (!be_->H2D(&buf_.st[e], &st, sizeof(EnvState))) return Fail(DTRL_ERR_DEVICE, be_->error());
	}
	return DTRL_OK;
}

// cScenarioSimChar::ApplyRandForce (scenarios/ScenarioSimChar.cpp:209-235): a random body part, a random 3-D direction scaled to a random
// magnitude in [min_perturb, max_perturb] (the planar character keeps x, y), a random duration. The reference draws from its racy,
// time-seeded global RNG; here the stream is a counter-based hash of (seed, global env id), so a run is reproducible.
int Engine::ApplyRandForce(const int32_t* env_ids, int n, uint64_t seed)
{
	if (env_ids && n < 0) return Fail(DTRL_ERR_ARG, "negative env count");
	const DevModel& m = cfg_.model;
	const int cnt = env_ids ? n : n_;
	std::vector<int32_t> link(cnt); std::vector<double> f(2 * static_cast<size_t>(cnt)), dur(cnt);
	for (int i = 0; i < cnt; ++i) {
		const int e = EnvIndex(env_ids, i);
		if (e < 0 || e >= n_) return Fail(DTRL_ERR_ARG, "env id out of range");
		uint64_t ctr = 0;
		const uint64_t key = rng_mix(seed ^ rng_mix(static_cast<uint64_t>(cfg_.run.env_id_base + e) + 0x5851F42D4C957F2DULL));
		auto uni = [&]() { const uint64_t z = rng_mix(key + (ctr++) * 0xD1342543DE82EF95ULL); return static_cast<double>(z >> 11) * (1.0 / 9007199254740992.0); };
		int part = static_cast<int>(uni() * m.L); if (part >= m.L) part = m.L - 1;        // every link of the shipped characters has a body
		double d[3];
		for (int k = 0; k < 3; ++k) { const double sgn = uni() < 0.5 ? -1.0 : 1.0; d[k] = sgn * uni(); }
		double nrm = std::sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]); if (nrm == 0) { d[0] = 1; nrm = 1; }
		const double mag = cfg_.min_perturb + uni() * (cfg_.max_perturb - cfg_.min_perturb);
		link[i] = part; f[2 * i] = mag * d[0] / nrm; f[2 * i + 1] = mag * d[1] / nrm;
		dur[i] = cfg_.min_perturb_duration + uni() * (cfg_.max_perturb_duration - cfg_.min_perturb_duration);
	}
	return AddPerturb(env_ids, cnt, link.data(), nullptr, f.data(), dur.data());
}

// n values of the kernel's arithmetic type from device memory into the ABI's doubles
bool Engine::D2HReal(double* dst, const real* src, size_t n)
{
	if (sizeof(real) == sizeof(double)) return be_->D2H(dst, src, sizeof(real) * n);
	std::vector<real> tmp(n);
	if (!be_->D2H(tmp.data(), src, sizeof(real) * n)) return false;
	for (size_t k = 0; k < n; ++k) dst[k] = static_cast<double>(tmp[k]);
	return true;
}

int Engine::GetPoliState(const int32_t* env_ids, int n, double* s)
{
	if (n < 0 || !s) return Fail(DTRL_ERR_ARG, "bad arguments");
	be_->Sync();
	for (int i = 0; i < n; ++i) {
		int e = EnvIndex(env_ids, i);
		if (e < 0 || e >= n_) return Fail(DTRL_ERR_ARG, "env id out of range");
		if (!D2HReal(s + static_cast<size_t>(i) * S_, buf_.poli_state + static_cast<size_t>(e) * S_, S_)) return Fail(DTRL_ERR_DEVICE, be_->error());
	}
	return DTRL_OK;
}

int Engine::GetPolicyOutput(const int32_t* env_ids, int n, double* y)
{
	if (n < 0 || !y) return Fail(DTRL_ERR_ARG, "bad arguments");
	if (!cfg_.has_policy_net) return Fail(DTRL_ERR_ARG, "no -policy_net= in the arguments: this batch has no network");
	be_->Sync();
	const int pad = cfg_.net.out_size - cfg_.user_out_size;   // the device net of a single-head controller carries an unused critic slot in front
	for (int i = 0; i < n; ++i) {
		int e = EnvIndex(env_ids, i);
		if (e < 0 || e >= n_) return Fail(DTRL_ERR_ARG, "env id out of range");
		if (!D2HReal(y + static_cast<size_t>(i) * cfg_.user_out_size, buf_.nn_out + static_cast<size_t>(e) * cfg_.net.out_size + pad, cfg_.user_out_size)) return Fail(DTRL_ERR_DEVICE, be_->error());
	}
	return DTRL_OK;
}

int Engine::SampleGround(int env, int n, const double* x, double* h, int32_t* seg, int32_t* oi, int32_t* oj)
{
	if (env < 0 || env >= n_) return Fail(DTRL_ERR_ARG, "env id out of range");
	be_->Sync();
	// reads back the env's ground record as the kernel will see it and applies the same sampling routine the kernel uses
	if (!FetchGroundRec(env)) return Fail(DTRL_ERR_DEVICE, be_->error());
	for (int i = 0; i < n; ++i) {
		int a, b, s;
		h[i] = sample_ground(tmp_rec_, x[i], nullptr, &a, &b, &s);
		if (seg) seg[i] = s;
		if (oi) oi[i] = a;
		if (oj) oj[i] = b;
	}
	return DTRL_OK;
}

int Engine::GroundWindowRec(int env, int32_t* w2, double* min_x2, double* max_x2, float* h0, float* h1, int cap, int64_t* num_builds)
{
	if (env < 0 || env >= n_) return Fail(DTRL_ERR_ARG, "env id out of range");
	be_->Sync();
	if (!FetchGroundRec(env)) return Fail(DTRL_ERR_DEVICE, be_->error());
	float* dst[2] = {h0, h1};
	for (int s = 0; s < 2; ++s) {
		if (w2) w2[s] = tmp_rec_.w[s];
		if (min_x2) min_x2[s] = tmp_rec_.min_x[s];
		if (max_x2) max_x2[s] = tmp_rec_.max_x[s];
		if (dst[s]) std::memcpy(dst[s], tmp_rec_.data[s], sizeof(float) * std::max(0, std::min<int>(cap, tmp_rec_.w[s])));
	}
	if (num_builds) {
		*num_builds = -1;
		if (cfg_.device_terrain) { GroundGen g; if (!be_->D2H(&g, &buf_.gen[env], sizeof(g))) return Fail(DTRL_ERR_DEVICE, be_->error()); *num_builds = g.builds; if (g.overflow) return Fail(DTRL_ERR_CAPACITY, "terrain segment exceeds kSegCap vertices"); }
	}
	return DTRL_OK;
}

int Engine::EvalStats(double* avg_dist, int64_t* episodes, int64_t* cycles, int64_t* resets)
{
	std::vector<EnvState> st;
	int rc = GetStates(nullptr, n_, st);
	if (rc != DTRL_OK) return rc;
	double sum = 0; int64_t ep = 0, cy = 0, rs = 0;
	for (const EnvState& s : st) { sum += s.avg_dist * s.num_episodes; ep += s.num_episodes; cy += s.num_cycles; rs += s.num_resets; }
	if (avg_dist) *avg_dist = ep > 0 ? sum / ep : 0.0;
	if (episodes) *episodes = ep;
	if (cycles) *cycles = cy;
	if (resets) *resets = rs;
	return DTRL_OK;
}

int Engine::KernelTime(double* avg_ms, int64_t* launches) { be_->KernelTime(avg_ms, launches); return DTRL_OK; }

int Engine::ProfileSections(unsigned long long* out, int cap)
{
	be_->Sync();
	std::vector<unsigned long long> all(static_cast<size_t>(kProfMax) * n_);
	if (!be_->D2H(all.data(), buf_.prof, all.size() * sizeof(unsigned long long))) return Fail(DTRL_ERR_DEVICE, be_->error());
	for (int k = 0; k < kProfMax && k < cap; ++k) { unsigned long long s = 0; for (int e = 0; e < n_; ++e) s += all[static_cast<size_t>(e) * kProfMax + k]; out[k] = s; }
	return DTRL_OK;
}

// per-env values of one profile section (developer builds)
int Engine::ProfileEnv(int section, unsigned long long* out, int cap)
{
	if (section < 0 || section >= kProfMax || !out || cap < n_) return Fail(DTRL_ERR_ARG, "bad arguments");
	be_->Sync();
	std::vector<unsigned long long> all(static_cast<size_t>(kProfMax) * n_);
	if (!be_->D2H(all.data(), buf_.prof, all.size() * sizeof(unsigned long long))) return Fail(DTRL_ERR_DEVICE, be_->error());
	for (int e = 0; e < n_; ++e) out[e] = all[static_cast<size_t>(e) * kProfMax + section];
	return DTRL_OK;
}

}  // namespace dtrl
