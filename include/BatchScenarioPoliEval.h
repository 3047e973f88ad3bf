// BatchScenarioPoliEval.h -- header-only C++ shim that puts the MI355X batched rollout engine (libdtrl.so, include/dtrl.h) behind the reference's
// policy-evaluation scenario interface: ONE cBatchScenarioPoliEval stands for the whole pool of cScenarioPoliEval objects that
// cOptScenarioPoliEval builds and drives (optimizer/scenarios/OptScenarioPoliEval.cpp:135-162 BuildScenePool, :169-198 EvalHelper,
// :213-239 OutputResults), with unmodified args/*.txt files (args/opt_poli_eval*.txt, args/dog_slopes_mixed_args.txt ...).
//
// Written against the reference's headers as they are (a maintainer compiles it inside the reference tree):
//   scenarios/Scenario.h:7-32          cScenario {ParseArgs, Init, Reset, Clear, Update, Shutdown, GetName}                  (base class)
//   scenarios/ScenarioPoliEval.h:20-26 GetAvgDist / ResetAvgDist / GetNumEpisodes / GetNumCycles / GetDistLog / SetRandSeed  (same names)
//   util/ArgParser.h:6-34              cArgParser
//   util/Rand.h                        cRand (SetRandSeed draws the pool's per-scene seeds the way BuildScenePool does)
// The counters are POOL aggregates, i.e. what cOptScenarioPoliEval::UpdateRecord folds the per-scene values into
// (cMathUtil::AddAverage over scenes, optimizer/scenarios/OptScenarioPoliEval.cpp:200-211):
//   GetNumEpisodes = sum of the scenes' mEpisodeCount since the last ResetAvgDist, GetAvgDist = episode-weighted mean of their mAvgDist,
//   GetNumCycles   = sum of the scenes' mCycleCount (never reset, scenarios/ScenarioPoliEval.cpp:132-136),
//   GetDistLog     = the scenes' mDistLog concatenated in pool order (the order OutputResults prints), each scene's episodes in time order.
// Error behaviour follows the reference's: methods print the engine's message and return a neutral value; no exceptions.
//
// tests/shim/drive_shim_eval.cpp compiles this header unchanged against /root/reference's headers and runs the EvalHelper loop through it.
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "dtrl.h"
#include "scenarios/Scenario.h"
#include "util/ArgParser.h"
#include "util/Rand.h"

class cBatchScenarioPoliEval : public cScenario
{
public:
	explicit cBatchScenarioPoliEval(int pool_size, int device_id = -1) : mPoolSize(pool_size), mDeviceID(device_id), mBatch(nullptr), mHasSeeds(false) {}
	virtual ~cBatchScenarioPoliEval() { Clear(); }

	// cScenarioSimChar::ParseArgs + cScenarioPoliEval::ParseArgs (scenarios/ScenarioSimChar.cpp:76-108, scenarios/ScenarioPoliEval.cpp:38-59): the engine
	// parses the same keys itself, so they are forwarded as "-key= value" pairs; the scenario is pinned to poli_eval (a pool of evaluation scenes)
	virtual void ParseArgs(const cArgParser& parser)
	{
		static const char* const keys[] = {"character_file", "state_file", "char_type", "char_ctrl", "terrain_file", "terrain_blend",
			"world_scale", "num_update_steps", "num_sim_substeps", "char_init_pos_x", "policy_net", "policy_model", "critic_net", "critic_model",
			"min_perturb", "max_perturb", "min_pertrub_duration", "max_perturb_duration",
			"data_root", "terrain_seed", "rand_seed", "global_env_offset", "terrain_gen", "link_contacts"};
		mArgs.clear();
		mArgs.push_back("-scenario="); mArgs.push_back("poli_eval");
		for (size_t i = 0; i < sizeof(keys) / sizeof(keys[0]); ++i) {
			std::string val;
			if (parser.ParseString(keys[i], val)) { mArgs.push_back(std::string("-") + keys[i] + "="); mArgs.push_back(val); }
		}
	}

	// cScenarioPoliEval::Init on every scene of the pool (counters and dist log cleared, scenarios/ScenarioPoliEval.cpp:61-92)
	virtual void Init()
	{
		Clear();
		std::vector<const char*> argv;
		for (size_t i = 0; i < mArgs.size(); ++i) argv.push_back(mArgs[i].c_str());
		dtrl_status rc = dtrl_create(argv.empty() ? nullptr : &argv[0], static_cast<int>(argv.size()), mPoolSize, mDeviceID, &mBatch);
		if (rc != DTRL_OK) { printf("cBatchScenarioPoliEval: dtrl_create failed (%d): %s\n", static_cast<int>(rc), dtrl_last_error(nullptr)); mBatch = nullptr; }
	}
	// cScenarioPoliEval::Reset on every scene; after SetRandSeed it rebuilds every scene's ground from its new seed ("rebuild ground",
	// optimizer/scenarios/OptScenarioPoliEval.cpp:155-158)
	virtual void Reset()
	{
		if (mBatch) {
			Check(dtrl_reset(mBatch, nullptr, mHasSeeds ? mPoolSize : 0, mHasSeeds ? &mSeeds[0] : nullptr), "dtrl_reset");
			mHasSeeds = false;   // cGroundVar2D::SeedRand happens once; later resets continue the scene's stream
		}
		cScenario::Reset();
	}
	virtual void Clear()
	{
		if (mBatch) { dtrl_destroy(mBatch); mBatch = nullptr; }
		mDistLog.clear();
	}
	virtual void Shutdown() { Clear(); }

	// cScenarioPoliEval::Update(dt) on every scene (scenarios/ScenarioPoliEval.cpp:110-125): step, and scenes that fell record their episode and reset
	virtual void Update(double time_elapsed) { if (mBatch) Check(dtrl_step(mBatch, time_elapsed), "dtrl_step"); }

	virtual double GetAvgDist() const { double d = 0; Stats(&d, nullptr, nullptr); return d; }
	virtual void ResetAvgDist() { if (mBatch) Check(dtrl_reset_avg_dist(mBatch), "dtrl_reset_avg_dist"); }
	virtual int GetNumEpisodes() const { int64_t e = 0; Stats(nullptr, &e, nullptr); return static_cast<int>(e); }
	virtual int GetNumCycles() const { int64_t c = 0; Stats(nullptr, nullptr, &c); return static_cast<int>(c); }
	virtual const std::vector<double>& GetDistLog() const
	{
		mDistLog.clear();
		int n = 0;
		if (mBatch && Check(dtrl_get_dist_log(mBatch, nullptr, nullptr, 0, &n), "dtrl_get_dist_log") && n > 0) {
			mDistLog.resize(n);
			if (!Check(dtrl_get_dist_log(mBatch, &mDistLog[0], nullptr, n, &n), "dtrl_get_dist_log")) mDistLog.clear();
		}
		return mDistLog;
	}
	// the pool's dist log with the pool member (scene index) of every entry
	virtual void GetDistLog(std::vector<double>& out_dist, std::vector<int>& out_scene) const
	{
		out_dist.clear(); out_scene.clear();
		int n = 0;
		if (!mBatch || !Check(dtrl_get_dist_log(mBatch, nullptr, nullptr, 0, &n), "dtrl_get_dist_log") || n <= 0) return;
		out_dist.resize(n); std::vector<int32_t> ids(n);
		if (!Check(dtrl_get_dist_log(mBatch, &out_dist[0], &ids[0], n, &n), "dtrl_get_dist_log")) { out_dist.clear(); return; }
		out_scene.assign(ids.begin(), ids.end());
	}
	// cOptScenarioPoliEval::OutputResults: one appended line, every logged distance in pool order, std::to_string, ", "
	virtual bool OutputResults(const std::string& out_file) const
	{
		const bool succ = mBatch && Check(dtrl_write_dist_log(mBatch, out_file.c_str()), "dtrl_write_dist_log");
		if (!succ) printf("Failed to output results to %s\n", out_file.c_str());
		return succ;
	}

	// cOptScenarioPoliEval::BuildScenePool seeds scene i with the i-th value of  curr = abs(rand.RandInt())  from a cRand seeded with `seed`
	// (optimizer/scenarios/OptScenarioPoliEval.cpp:139-160) and calls cScenarioPoliEval::SetRandSeed(curr) + Reset() on it. One call here does that
	// for the whole pool: the per-scene seeds are drawn from the reference's own cRand in the same order; the next Reset() rebuilds the grounds.
	virtual void SetRandSeed(unsigned long seed)
	{
		cRand rand;
		rand.Seed(seed);
		mSeeds.resize(mPoolSize);
		for (int i = 0; i < mPoolSize; ++i) mSeeds[i] = static_cast<uint64_t>(static_cast<unsigned long>(std::abs(rand.RandInt())));
		mHasSeeds = mPoolSize > 0;
	}
	// explicit per-scene seeds (a driver that keeps its own seed list)
	virtual void SetRandSeeds(const std::vector<unsigned long>& seeds)
	{
		mSeeds.assign(mPoolSize, 0);
		for (int i = 0; i < mPoolSize && i < static_cast<int>(seeds.size()); ++i) mSeeds[i] = seeds[i];
		mHasSeeds = static_cast<int>(seeds.size()) >= mPoolSize && mPoolSize > 0;
	}
	virtual const std::vector<uint64_t>& GetSceneSeeds() const { return mSeeds; }

	// what the drivers reach through GetNNController(): LoadNet / LoadModel on every scene's controller = one weight push
	virtual int GetPoolSize() const { return mPoolSize; }
	virtual size_t GetNumPolicyParams() const { size_t n = 0; if (mBatch) dtrl_policy_num_params(mBatch, &n); return n; }
	virtual bool SetPolicy(const float* weights, size_t n, const double* in_off, const double* in_scale, const double* out_off, const double* out_scale)
	{
		return mBatch && Check(dtrl_set_policy(mBatch, weights, n, in_off, in_scale, out_off, out_scale), "dtrl_set_policy");
	}
	virtual bool LoadScale(const std::string& scale_file) { return mBatch && Check(dtrl_load_scale_file(mBatch, scale_file.c_str()), "dtrl_load_scale_file"); }
	virtual void GetDims(int& out_state_size, int& out_nn_out) const
	{
		out_state_size = 0; out_nn_out = 0;
		if (mBatch) dtrl_dims(mBatch, nullptr, nullptr, &out_state_size, nullptr, nullptr, &out_nn_out, nullptr, nullptr);
	}
	virtual dtrl_batch* GetBatch() { return mBatch; }
	virtual bool IsValid() const { return mBatch != nullptr; }

	virtual std::string GetName() const { return "Batch Policy Evaluation"; }

protected:
	int mPoolSize, mDeviceID;
	dtrl_batch* mBatch;
	std::vector<std::string> mArgs;
	std::vector<uint64_t> mSeeds;
	bool mHasSeeds;
	mutable std::vector<double> mDistLog;

	void Stats(double* avg_dist, int64_t* episodes, int64_t* cycles) const
	{
		if (mBatch) Check(dtrl_eval_stats(mBatch, avg_dist, episodes, cycles, nullptr), "dtrl_eval_stats");
	}
	bool Check(dtrl_status rc, const char* what) const
	{
		if (rc == DTRL_OK) return true;
		printf("cBatchScenarioPoliEval: %s failed (%d): %s\n", what, static_cast<int>(rc), dtrl_last_error(mBatch));
		return false;
	}
};
