// ref_learn_api.cpp -- TEST INFRASTRUCTURE (oracle/_ref_build -> oracle/_ref/libref_learn.so): a C ABI over the reference's OWN trainers (cMACETrainer, cQNetTrainer,
// cCaclaTrainer: /root/reference/learning, compiled unchanged) so that tests can feed them the tuple stream the product's trainers get and compare, iteration by
// iteration: replay rows and flags, critic / actor / actor-batch index buffers, stage, iteration counters, normalisers, and -- through the harness' networks
// (ref_learn_net.cpp, oracle/reflearn.py) -- every minibatch and label the reference builds and the weights after N iterations.
// The probe subclasses below add NO behaviour: they only open protected members for reading.
#include <cstring>
#include <memory>
#include <vector>

#include "learning/MACETrainer.h"
#include "learning/QNetTrainer.h"
#include "learning/CaclaTrainer.h"
#include "learning/NeuralNetLearner.h"
#include "learning/ACLearner.h"
#include "util/MathUtil.h"
#include "util/Rand.h"
#include "ref_learn_harness.h"

namespace {
int NetId(const std::unique_ptr<cNeuralNet>& n) { return n ? static_cast<int>(n->GetParams().size()) - 1 : -1; }

struct Probe {
	virtual ~Probe() {}
	virtual cNeuralNetTrainer* T() = 0;
	virtual int PoolNet(int i) = 0;
	virtual int NumPool() = 0;
	virtual int ActorNet() { return -1; }
	virtual const std::vector<int>* Buffer(int which) { return nullptr; }   // 0 critic, 1 actor (exp / off-policy), 2 actor batch
	virtual int ActorIter() { return 0; }
	virtual const Eigen::MatrixXf& Mem() = 0;
	virtual const std::vector<unsigned int>& Flags() = 0;
	virtual int Head() = 0; virtual int NumStored() = 0; virtual int Stage() = 0;
};
#define PROBE_COMMON \
	cNeuralNetTrainer* T() override { return this; } \
	int PoolNet(int i) override { return (i >= 0 && i < static_cast<int>(mNetPool.size())) ? NetId(mNetPool[i]) : -1; } \
	int NumPool() override { return static_cast<int>(mNetPool.size()); } \
	const Eigen::MatrixXf& Mem() override { return mPlaybackMem; } \
	const std::vector<unsigned int>& Flags() override { return mFlagBuffer; } \
	int Head() override { return mBufferHead; } int NumStored() override { return mNumTuples; } int Stage() override { return static_cast<int>(mStage); }

struct MaceProbe : public cMACETrainer, public Probe {
	PROBE_COMMON
	const std::vector<int>* Buffer(int w) override { return w == 0 ? &mCriticBuffer : w == 1 ? &mActorBuffer : &mActorBatchBuffer; }
	int ActorIter() override { return mActorIter; }
};
struct QProbe : public cQNetTrainer, public Probe {
	PROBE_COMMON
};
struct CaclaProbe : public cCaclaTrainer, public Probe {
	PROBE_COMMON
	int ActorNet() override { return NetId(mActorNet); }
	const std::vector<int>* Buffer(int w) override { return w == 1 ? &mOffPolicyBuffer : w == 2 ? &mActorBatchBuffer : nullptr; }
	int ActorIter() override { return mActorIter; }
};
struct Handle {
	std::shared_ptr<cNeuralNetTrainer> trainer; Probe* probe = nullptr;
	// the env side of the reference (scenarios/ScenarioExp*.cpp hand their tuple buffer to a cNeuralNetLearner; learning/NeuralNetLearner.cpp:33-46): a learner the
	// trainer itself hands out (RequestLearner: cNeuralNetLearner, or cACLearner from cACTrainer), with nets of its own standing for the controller's
	std::shared_ptr<cNeuralNetLearner> learner; std::unique_ptr<cNeuralNet> learner_net, learner_critic;
};
Eigen::VectorXd Vec(const double* p, int n) { Eigen::VectorXd v(n); for (int i = 0; i < n; ++i) v[i] = p[i]; return v; }
}  // namespace

extern "C" {
// cMathUtil::gRand (util/MathUtil.cpp:4) is what every minibatch draw of the trainers reads (cMathUtil::RandInt); independent cRand streams (util/Rand.cpp) with the
// same seed give the product's trainer the same draws
void ref_learn_seed_rand(unsigned long seed) { cMathUtil::SeedRand(seed); }
void* ref_learn_rand_new(unsigned long seed) { cRand* r = new cRand(); r->Seed(seed); return r; }
void ref_learn_rand_free(void* r) { delete static_cast<cRand*>(r); }
int ref_learn_rand_int(void* r, int lo, int hi) { return static_cast<cRand*>(r)->RandInt(lo, hi); }
double ref_learn_rand_double(void* r, double lo, double hi) { return static_cast<cRand*>(r)->RandDouble(lo, hi); }

struct RefLearnParams {
	const char* net_file; const char* solver_file; const char* actor_net_file; const char* actor_solver_file;
	int playback_mem_size, pool_size, num_init_samples, num_steps_per_iter, freeze_target_iters, init_input_offset_scale;
	double discount;
	int num_action_frags, action_frag_size;     // cMACETrainer::SetNumActionFrags / SetActionFragSize (scenarios/ScenarioTrainMACE... via the controller)
};
// kind: 0 = cMACETrainer, 1 = cQNetTrainer, 2 = cCaclaTrainer
void* ref_learn_trainer_create(int kind, const RefLearnParams* p)
{
	Handle* h = new Handle();
	cTrainerInterface::tParams tp;
	tp.mNetFile = p->net_file; tp.mSolverFile = p->solver_file;
	tp.mPlaybackMemSize = p->playback_mem_size; tp.mPoolSize = p->pool_size; tp.mNumInitSamples = p->num_init_samples;
	tp.mNumStepsPerIter = p->num_steps_per_iter; tp.mFreezeTargetIters = p->freeze_target_iters; tp.mDiscount = p->discount;
	tp.mInitInputOffsetScale = p->init_input_offset_scale != 0;
	if (kind == 0) {
		auto t = std::make_shared<MaceProbe>();
		t->SetNumActionFrags(p->num_action_frags); t->SetActionFragSize(p->action_frag_size);
		h->trainer = t; h->probe = t.get();
	} else if (kind == 1) {
		auto t = std::make_shared<QProbe>(); h->trainer = t; h->probe = t.get();
	} else {
		auto t = std::make_shared<CaclaProbe>();
		t->SetActorFiles(p->actor_solver_file, p->actor_net_file);
		h->trainer = t; h->probe = t.get();
	}
	h->trainer->Init(tp);
	return h;
}
void ref_learn_trainer_destroy(void* hv) { delete static_cast<Handle*>(hv); }
int ref_learn_add_tuple(void* hv, double reward, unsigned int flags, const double* s_beg, const double* s_end, const double* action, int S, int A)
{
	Handle* h = static_cast<Handle*>(hv);
	tExpTuple t;
	t.mReward = reward; t.mFlags = flags;
	t.mStateBeg = Vec(s_beg, S); t.mStateEnd = Vec(s_end, S); t.mAction = Vec(action, A);
	return h->trainer->AddTuple(t);
}
void ref_learn_train(void* hv) { static_cast<Handle*>(hv)->trainer->Train(); }
// cNeuralNetLearner::Train(tuples) (learning/NeuralNetLearner.cpp:33-46, unchanged): Lock, UpdateTrainer, AddTuples, Train, SyncNet (CopyModel of the trainer's net into the
// learner's, i.e. the controller's), Unlock -- the call cScenarioExp makes when its tuple buffer is full. n rows of [reward | flags], states and actions as separate arrays.
// Returns the harness id of the learner's (actor) net, whose parameters after the call are what the env threads would run with.
int ref_learn_learner_train(void* hv, int n, const double* reward, const unsigned int* flags, const double* s_beg, const double* s_end, const double* action, int S, int A)
{
	Handle* h = static_cast<Handle*>(hv);
	if (!h->learner) {
		h->trainer->RequestLearner(h->learner);
		h->learner_net.reset(new cNeuralNet());
		h->learner->SetNet(h->learner_net.get());
		if (auto ac = std::dynamic_pointer_cast<cACLearner>(h->learner)) { h->learner_critic.reset(new cNeuralNet()); ac->SetCriticNet(h->learner_critic.get()); }
		h->learner->Init();        // LoadNet(trainer's net file) + SyncNet
	}
	std::vector<tExpTuple> tuples(static_cast<size_t>(n));
	for (int i = 0; i < n; ++i) {
		tExpTuple& t = tuples[static_cast<size_t>(i)];
		t.mReward = reward[i]; t.mFlags = flags[i];
		t.mStateBeg = Vec(s_beg + static_cast<size_t>(i) * S, S); t.mStateEnd = Vec(s_end + static_cast<size_t>(i) * S, S); t.mAction = Vec(action + static_cast<size_t>(i) * A, A);
	}
	h->learner->Train(tuples);
	return NetId(h->learner_net);
}
int ref_learn_learner_iter(void* hv) { Handle* h = static_cast<Handle*>(hv); return h->learner ? h->learner->GetIter() : -1; }
int ref_learn_learner_num_tuples(void* hv) { Handle* h = static_cast<Handle*>(hv); return h->learner ? h->learner->GetNumTuples() : -1; }
int ref_learn_iter(void* hv) { return static_cast<Handle*>(hv)->trainer->GetIter(); }
int ref_learn_actor_iter(void* hv) { return static_cast<Handle*>(hv)->probe->ActorIter(); }
int ref_learn_stage(void* hv) { return static_cast<Handle*>(hv)->probe->Stage(); }
int ref_learn_head(void* hv) { return static_cast<Handle*>(hv)->probe->Head(); }
int ref_learn_num_stored(void* hv) { return static_cast<Handle*>(hv)->probe->NumStored(); }
int ref_learn_num_tuples(void* hv) { return static_cast<Handle*>(hv)->trainer->GetNumTuples(); }
int ref_learn_batch_size(void* hv) { return static_cast<Handle*>(hv)->trainer->GetBatchSize(); }
int ref_learn_state_size(void* hv) { return static_cast<Handle*>(hv)->trainer->GetStateSize(); }
int ref_learn_action_size(void* hv) { return static_cast<Handle*>(hv)->trainer->GetActionSize(); }
int ref_learn_num_pool(void* hv) { return static_cast<Handle*>(hv)->probe->NumPool(); }
int ref_learn_pool_net(void* hv, int i) { return static_cast<Handle*>(hv)->probe->PoolNet(i); }
int ref_learn_actor_net(void* hv) { return static_cast<Handle*>(hv)->probe->ActorNet(); }
// index buffers: copies up to cap entries, returns the buffer's length (-1: this trainer has no such buffer)
int ref_learn_buffer(void* hv, int which, int* out, int cap)
{
	const std::vector<int>* b = static_cast<Handle*>(hv)->probe->Buffer(which);
	if (!b) return -1;
	for (int i = 0; i < static_cast<int>(b->size()) && i < cap; ++i) out[i] = (*b)[i];
	return static_cast<int>(b->size());
}
int ref_learn_mem_cols(void* hv) { return static_cast<int>(static_cast<Handle*>(hv)->probe->Mem().cols()); }
void ref_learn_mem_row(void* hv, int t, float* out, unsigned int* flags)
{
	const Eigen::MatrixXf& m = static_cast<Handle*>(hv)->probe->Mem();
	for (int j = 0; j < static_cast<int>(m.cols()); ++j) out[j] = m(t, j);
	*flags = static_cast<Handle*>(hv)->probe->Flags()[t];
}
void ref_learn_set_input_offset_scale(void* hv, const double* off, const double* scale, int n) { static_cast<Handle*>(hv)->trainer->SetInputOffsetScale(Vec(off, n), Vec(scale, n)); }
void ref_learn_set_output_offset_scale(void* hv, const double* off, const double* scale, int n) { static_cast<Handle*>(hv)->trainer->SetOutputOffsetScale(Vec(off, n), Vec(scale, n)); }
void ref_learn_set_actor_output_offset_scale(void* hv, const double* off, const double* scale, int n) { static_cast<Handle*>(hv)->trainer->SetActorOutputOffsetScale(Vec(off, n), Vec(scale, n)); }
void ref_learn_set_critic_output_offset_scale(void* hv, const double* off, const double* scale, int n) { static_cast<Handle*>(hv)->trainer->SetCriticOutputOffsetScale(Vec(off, n), Vec(scale, n)); }
// the current net's input normaliser (after cNeuralNetTrainer::UpdateOffsetScale)
void ref_learn_get_input_offset_scale(void* hv, double* off, double* scale, int n)
{
	const auto& net = static_cast<Handle*>(hv)->trainer->GetNet();
	for (int i = 0; i < n; ++i) { off[i] = net->GetInputOffset()[i]; scale[i] = net->GetInputScale()[i]; }
}
}
