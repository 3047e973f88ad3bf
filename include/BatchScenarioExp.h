// BatchScenarioExp.h -- header-only C++ shim that puts the MI355X batched rollout engine (libdtrl.so, include/dtrl.h) behind the reference's
// own scenario interface, so that a cScenarioTrain-shaped driver (scenarios/ScenarioTrain.cpp:197-222, 376-410, 467-475) keeps working with
// unmodified args/*.txt files: ONE cBatchScenarioExp stands for a pool of num_envs cScenarioExp objects.
//
// It is written against the reference's headers as they are (a maintainer compiles it inside the reference tree):
//   scenarios/Scenario.h:7-32   cScenario {ParseArgs, Init, Reset, Clear, Update, Shutdown, GetName}          (base class)
//   scenarios/ScenarioExp.h:16-38   IsTupleBufferFull / GetTuples / ResetTupleBuffer / SetBufferSize, EnableExplore / Set+GetExpRate /
//                                   ExpTemp / ExpBaseActionRate                                               (same names, batch-wide)
//   scenarios/ScenarioSimChar.h     SetTerrainParamsLerp
//   learning/ExpTuple.h:5-23    tExpTuple {mID, mReward, mFlags, mStateBeg, mAction, mStateEnd}                (what GetTuples returns)
//   util/ArgParser.h:6-34       cArgParser (ParseArgs reads the keys the engine understands and rebuilds "-key= value" pairs)
// Error behaviour follows the reference's: methods print the engine's message and return false / leave the object unusable; no exceptions.
//
// tests/test_boundary.py compiles this header unchanged against /root/reference's headers (with the stand-in Eigen of oracle/_ref_build)
// and drives frames through it; tests/shim/drive_shim.cpp is the driver.
#pragma once
#include <cstdio>
#include <string>
#include <vector>

#include "dtrl.h"
#include "learning/ExpTuple.h"
#include "scenarios/Scenario.h"
#include "util/ArgParser.h"

class cBatchScenarioExp : public cScenario
{
public:
	explicit cBatchScenarioExp(int num_envs, int device_id = -1)
		: mNumEnvs(num_envs), mDeviceID(device_id), mBatch(nullptr), mTupleBufferSize(32), mEnableExplore(true),
		  mExpRate(0.1), mExpTemp(1), mExpBaseActionRate(0.01), mStateSize(0), mActionSize(0)
	{
	}
	virtual ~cBatchScenarioExp() { Clear(); }

	// cScenarioSimChar::ParseArgs + cScenarioExp::ParseArgs (scenarios/ScenarioSimChar.cpp:76-108, scenarios/ScenarioExp.cpp:31-45): the engine parses
	// the same keys itself, so they are forwarded as "-key= value" pairs. Extra keys (data_root, terrain_seed, rand_seed, global_env_offset,
	// tuple_ring_capacity, terrain_gen, link_contacts) pass through when present.
	virtual void ParseArgs(const cArgParser& parser)
	{
		static const char* const keys[] = {"scenario", "character_file", "state_file", "char_type", "char_ctrl", "terrain_file", "terrain_blend",
			"world_scale", "num_update_steps", "num_sim_substeps", "char_init_pos_x", "policy_net", "policy_model", "critic_net", "critic_model",
			"tuple_buffer_size", "exp_rate", "exp_temp", "exp_base_rate", "min_perturb", "max_perturb", "min_pertrub_duration", "max_perturb_duration",
			"data_root", "terrain_seed", "rand_seed", "global_env_offset", "tuple_ring_capacity", "terrain_gen", "link_contacts"};
		mArgs.clear();
		for (size_t i = 0; i < sizeof(keys) / sizeof(keys[0]); ++i) {
			std::string val;
			if (parser.ParseString(keys[i], val)) { mArgs.push_back(std::string("-") + keys[i] + "="); mArgs.push_back(val); }
		}
		parser.ParseInt("tuple_buffer_size", mTupleBufferSize);
		parser.ParseDouble("exp_rate", mExpRate);
		parser.ParseDouble("exp_temp", mExpTemp);
		parser.ParseDouble("exp_base_rate", mExpBaseActionRate);
	}

	// cScenarioExp::Init on every scene of the pool (cScenarioTrain::BuildScenePool)
	virtual void Init()
	{
		Clear();
		std::vector<const char*> argv;
		for (size_t i = 0; i < mArgs.size(); ++i) argv.push_back(mArgs[i].c_str());
		dtrl_status rc = dtrl_create(argv.empty() ? nullptr : &argv[0], static_cast<int>(argv.size()), mNumEnvs, mDeviceID, &mBatch);
		if (rc != DTRL_OK) { printf("cBatchScenarioExp: dtrl_create failed (%d): %s\n", static_cast<int>(rc), dtrl_last_error(nullptr)); mBatch = nullptr; return; }
		int A = 0;
		dtrl_dims(mBatch, nullptr, nullptr, &mStateSize, &A, nullptr, nullptr, nullptr, nullptr);
		mActionSize = A;
		mRows.resize(static_cast<size_t>(RowCapacity()) * RowWidth());
		mRowFlags.resize(RowCapacity());
		mRowEnvs.resize(RowCapacity());
		Check(dtrl_set_explore(mBatch, mEnableExplore ? 1 : 0, mExpRate, mExpTemp, mExpBaseActionRate), "dtrl_set_explore");
	}
	virtual void Reset()
	{
		if (mBatch) Check(dtrl_reset(mBatch, nullptr, 0, nullptr), "dtrl_reset");
		cScenario::Reset();   // reset callback
	}
	virtual void Clear()
	{
		if (mBatch) { dtrl_destroy(mBatch); mBatch = nullptr; }
		mTupleBuffer.clear();
	}
	virtual void Shutdown() { Clear(); }

	// cScenarioExp::Update(dt) on every scene (scenarios/ScenarioExp.cpp:83-98), then the completed tuples of the frame join the buffer
	virtual void Update(double time_elapsed)
	{
		if (!mBatch) return;
		if (!Check(dtrl_step(mBatch, time_elapsed), "dtrl_step")) return;
		int n = 0;
		if (!Check(dtrl_drain_tuples(mBatch, &mRows[0], &mRowFlags[0], &mRowEnvs[0], RowCapacity(), &n), "dtrl_drain_tuples")) return;
		const int S = mStateSize, A = mActionSize, W = RowWidth();
		for (int i = 0; i < n; ++i) {
			// MACE replay row [r | s | a | s'] (learning/MACETrainer.cpp:373-401) -> tExpTuple
			const float* r = &mRows[static_cast<size_t>(i) * W];
			tExpTuple t(S, A);
			t.mID = mRowEnvs[i];
			t.mReward = r[0];
			t.mFlags = mRowFlags[i];
			for (int k = 0; k < S; ++k) { t.mStateBeg[k] = r[1 + k]; t.mStateEnd[k] = r[1 + S + A + k]; }
			for (int k = 0; k < A; ++k) t.mAction[k] = r[1 + S + k];
			mTupleBuffer.push_back(t);
		}
	}

	virtual void SetBufferSize(int size) { mTupleBufferSize = size; }
	virtual bool IsTupleBufferFull() const { return static_cast<int>(mTupleBuffer.size()) >= mTupleBufferSize; }
	virtual void ResetTupleBuffer() { mTupleBuffer.clear(); }
	virtual const std::vector<tExpTuple>& GetTuples() const { return mTupleBuffer; }

	virtual void EnableExplore(bool enable) { mEnableExplore = enable; PushExplore(); }
	virtual void SetExpRate(double rate) { mExpRate = rate; PushExplore(); }
	virtual void SetExpTemp(double temp) { mExpTemp = temp; PushExplore(); }
	virtual void SetExpBaseActionRate(double rate) { mExpBaseActionRate = rate; PushExplore(); }
	virtual double GetExpRate() const { return mExpRate; }
	virtual double GetExpTemp() const { return mExpTemp; }
	virtual double GetExpBaseActionRate() const { return mExpBaseActionRate; }
	virtual void SetTerrainParamsLerp(double lerp) { if (mBatch) Check(dtrl_set_terrain_lerp(mBatch, lerp), "dtrl_set_terrain_lerp"); }

	// what the drivers reach through GetNNController(): sizes, weight push (cNeuralNet::CopyModel from the trainer), output normaliser construction
	virtual int GetNumEnvs() const { return mNumEnvs; }
	virtual int GetPoliStateSize() const { return mStateSize; }
	virtual int GetPoliActionSize() const { return mActionSize; }
	virtual size_t GetNumPolicyParams() const { size_t n = 0; if (mBatch) dtrl_policy_num_params(mBatch, &n); return n; }
	virtual bool SetPolicy(const float* weights, size_t n, const double* in_off, const double* in_scale, const double* out_off, const double* out_scale)
	{
		return mBatch && Check(dtrl_set_policy(mBatch, weights, n, in_off, in_scale, out_off, out_scale), "dtrl_set_policy");
	}
	virtual bool BuildNNOutputOffsetScale(Eigen::VectorXd& out_offset, Eigen::VectorXd& out_scale) const
	{
		if (!mBatch) return false;
		int nn_out = 0;
		dtrl_dims(mBatch, nullptr, nullptr, nullptr, nullptr, nullptr, &nn_out, nullptr, nullptr);
		std::vector<double> off(nn_out), sc(nn_out);
		if (dtrl_build_output_offset_scale(mBatch, nn_out ? &off[0] : nullptr, nn_out ? &sc[0] : nullptr) != DTRL_OK) return false;
		out_offset.resize(nn_out); out_scale.resize(nn_out);
		for (int i = 0; i < nn_out; ++i) { out_offset[i] = off[i]; out_scale[i] = sc[i]; }
		return true;
	}
	virtual dtrl_batch* GetBatch() { return mBatch; }
	virtual bool IsValid() const { return mBatch != nullptr; }

	virtual std::string GetName() const { return "Batch Exploration"; }

protected:
	int mNumEnvs, mDeviceID;
	dtrl_batch* mBatch;
	std::vector<std::string> mArgs;
	int mTupleBufferSize;
	bool mEnableExplore;
	double mExpRate, mExpTemp, mExpBaseActionRate;
	int mStateSize, mActionSize;
	std::vector<tExpTuple> mTupleBuffer;
	std::vector<float> mRows;
	std::vector<uint32_t> mRowFlags;
	std::vector<int32_t> mRowEnvs;

	int RowWidth() const { return 1 + 2 * mStateSize + mActionSize; }
	int RowCapacity() const { int32_t cap = 0; if (mBatch) dtrl_tuple_stats(mBatch, nullptr, nullptr, nullptr, &cap); return cap > 0 ? cap : 2 * mNumEnvs; }
	void PushExplore() { if (mBatch) Check(dtrl_set_explore(mBatch, mEnableExplore ? 1 : 0, mExpRate, mExpTemp, mExpBaseActionRate), "dtrl_set_explore"); }
	bool Check(dtrl_status rc, const char* what) const
	{
		if (rc == DTRL_OK) return true;
		printf("cBatchScenarioExp: %s failed (%d): %s\n", what, static_cast<int>(rc), dtrl_last_error(mBatch));
		return false;
	}
};
