// ref_learn_net_native.cpp -- TEST INFRASTRUCTURE (oracle/_ref_build -> oracle/_ref/libref_learn_native.so / libref_learn_native_hip.so).
//
// The reference's trainers (learning/NeuralNetTrainer.cpp, MACETrainer.cpp, QNetTrainer.cpp, ACTrainer.cpp, CaclaTrainer.cpp, ... compiled unchanged) with the
// PRODUCT's networks behind them: every member function of learning/NeuralNet.h's cNeuralNet is a forward to a cBatchNeuralNet (include/BatchNeuralNet.h, the shim
// a maintainer would drop into the reference tree) -- i.e. exactly the substitution INTEGRATION.md 4b describes: cMACETrainer keeps its replay memory, index buffers,
// minibatch draws, labels, stages and target refreshes, and only its nets run on the native step (the plain-loop check build on the CPU box, the HIP kernels on the
// MI355X). tests/test_reference_learn.py compares such a run with hip_trainer.HipMACETrainer on the same tuples and the same random stream.
// Sibling of ref_learn_net.cpp (same class, networks from the numpy harness instead); the two are never linked together.
#include <cassert>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <vector>

#include "learning/NeuralNet.h"
#include "BatchNeuralNet.h"

namespace {
std::vector<cBatchNeuralNet*> gNets;      // id -> net (ids are never reused: the C API names nets by them)
std::string gDataRoot;
int gDevice = -1;
cBatchNeuralNet* Impl(const caffe::Net<double>* n) { return n ? static_cast<cBatchNeuralNet*>(n->mImpl.get()) : nullptr; }
}
extern "C" {
void ref_learn_native_config(const char* data_root, int device_id) { gDataRoot = data_root ? data_root : ""; gDevice = device_id; }
long long ref_learn_native_num_params(int id) { return (id >= 0 && id < static_cast<int>(gNets.size()) && gNets[id]) ? gNets[id]->CalcNumParams() : 0; }
int ref_learn_native_get_params(int id, float* out, long long n)
{
	if (id < 0 || id >= static_cast<int>(gNets.size()) || !gNets[id]) return 1;
	std::vector<float> w; gNets[id]->GetParamsFlat(w);
	if (static_cast<long long>(w.size()) != n) return 1;
	for (long long i = 0; i < n; ++i) out[i] = w[static_cast<size_t>(i)];
	return 0;
}
int ref_learn_native_set_params(int id, const float* in, long long n)
{
	if (id < 0 || id >= static_cast<int>(gNets.size()) || !gNets[id] || n != gNets[id]->CalcNumParams()) return 1;
	gNets[id]->SetParamsFlat(std::vector<float>(in, in + n));
	return 0;
}
// the harness entry point of the sibling library: nothing to install here
void ref_learn_set_harness(const void*) {}
}

class cNNSolver { public: int mUnused = 0; };

std::mutex cNeuralNet::gOutputLock;
cNeuralNet::tProblem::tProblem() { mX.resize(0, 0); mY.resize(0, 0); mPassesPerStep = 100; }
bool cNeuralNet::tProblem::HasData() const { return mX.size() > 0; }

cNeuralNet::cCaffeNetWrapper::cCaffeNetWrapper(const std::string& net_file, caffe::Phase phase) : caffe::Net<tNNData>(net_file, phase)
{
	auto net = std::make_shared<cBatchNeuralNet>(gDevice, gDataRoot);
	net->LoadNet(net_file);
	mHarnessId = static_cast<int>(gNets.size());
	gNets.push_back(net.get());
	mImpl = net;
}
cNeuralNet::cCaffeNetWrapper::~cCaffeNetWrapper() { if (mHarnessId >= 0) gNets[mHarnessId] = nullptr; }
int cNeuralNet::cCaffeNetWrapper::GetLayerIdx(const std::string&) const { return -1; }

cNeuralNet::cNeuralNet() { Clear(); mAsync = false; }
cNeuralNet::~cNeuralNet() {}
void cNeuralNet::LoadNet(const std::string& net_file)
{
	if (net_file != "") {
		Clear();
		mNet = std::unique_ptr<cCaffeNetWrapper>(new cCaffeNetWrapper(net_file, caffe::TEST));
		if (!ValidOffsetScale()) InitOffsetScale();
	}
}
void cNeuralNet::LoadModel(const std::string& model_file) { if (model_file != "" && HasNet()) { Impl(mNet.get())->LoadModel(model_file); mValidModel = true; } }
void cNeuralNet::LoadSolver(const std::string& solver_file, bool async)
{
	if (solver_file != "") {
		assert(HasNet());
		mSolverFile = solver_file; mAsync = async;
		Impl(mNet.get())->LoadSolver(solver_file, false);
		mSolver = std::make_shared<cNNSolver>();
		if (!ValidOffsetScale()) InitOffsetScale();
	}
}
void cNeuralNet::LoadScale(const std::string&) {}
void cNeuralNet::Clear()
{
	mNet.reset(); mSolver.reset(); mValidModel = false;
	mInputOffset.resize(0); mInputScale.resize(0); mOutputOffset.resize(0); mOutputScale.resize(0);
}
void cNeuralNet::Train(const tProblem& prob)
{
	if (!HasSolver()) { printf("Solver has not been initialized\n"); assert(false); return; }
	cBatchNeuralNet::tProblem p; p.mX = prob.mX; p.mY = prob.mY; p.mPassesPerStep = prob.mPassesPerStep;
	Impl(mNet.get())->Train(p);
	mValidModel = true;
}
double cNeuralNet::ForwardBackward(const tProblem&) { fprintf(stderr, "libref_learn_native: ForwardBackward (asynchronous trainers) is not part of the shim\n"); abort(); }
void cNeuralNet::StepSolver(int) { abort(); }
void cNeuralNet::ResetSolver() { if (HasNet()) Impl(mNet.get())->ResetSolver(); }
void cNeuralNet::CalcOffsetScale(const Eigen::MatrixXd& X, Eigen::VectorXd& out_offset, Eigen::VectorXd& out_scale) const { Impl(mNet.get())->CalcOffsetScale(X, out_offset, out_scale); }
void cNeuralNet::SetInputOffsetScale(const Eigen::VectorXd& offset, const Eigen::VectorXd& scale) { mInputOffset = offset; mInputScale = scale; Impl(mNet.get())->SetInputOffsetScale(offset, scale); }
void cNeuralNet::SetOutputOffsetScale(const Eigen::VectorXd& offset, const Eigen::VectorXd& scale) { mOutputOffset = offset; mOutputScale = scale; Impl(mNet.get())->SetOutputOffsetScale(offset, scale); }
const Eigen::VectorXd& cNeuralNet::GetInputOffset() const { return mInputOffset; }
const Eigen::VectorXd& cNeuralNet::GetInputScale() const { return mInputScale; }
const Eigen::VectorXd& cNeuralNet::GetOutputOffset() const { return mOutputOffset; }
const Eigen::VectorXd& cNeuralNet::GetOutputScale() const { return mOutputScale; }
void cNeuralNet::Eval(const Eigen::VectorXd& x, Eigen::VectorXd& out_y) const { Impl(mNet.get())->Eval(x, out_y); }
void cNeuralNet::EvalBatch(const Eigen::MatrixXd& X, Eigen::MatrixXd& out_Y) const { Impl(mNet.get())->EvalBatch(X, out_Y); }
void cNeuralNet::Backward(const Eigen::VectorXd&, Eigen::VectorXd&) const { abort(); }
void cNeuralNet::EvalBatchNet(const Eigen::MatrixXd& X, Eigen::MatrixXd& out_Y) const { EvalBatch(X, out_Y); }
void cNeuralNet::EvalBatchSolver(const Eigen::MatrixXd& X, Eigen::MatrixXd& out_Y) const { EvalBatch(X, out_Y); }
int cNeuralNet::GetInputSize() const { return HasNet() ? Impl(mNet.get())->GetInputSize() : 0; }
int cNeuralNet::GetOutputSize() const { return HasNet() ? Impl(mNet.get())->GetOutputSize() : 0; }
int cNeuralNet::GetBatchSize() const { return HasSolver() ? Impl(mNet.get())->GetBatchSize() : 0; }
int cNeuralNet::CalcNumParams() const { return HasNet() ? Impl(mNet.get())->CalcNumParams() : 0; }
void cNeuralNet::OutputModel(const std::string& f) const { if (HasNet()) Impl(mNet.get())->OutputModel(f); }
void cNeuralNet::PrintParams() const {}
bool cNeuralNet::HasNet() const { return mNet != nullptr; }
bool cNeuralNet::HasSolver() const { return mSolver != nullptr; }
bool cNeuralNet::HasLayer(const std::string) const { return false; }
bool cNeuralNet::HasValidModel() const { return mValidModel; }
void cNeuralNet::NormalizeInput(Eigen::MatrixXd& X) const { if (ValidOffsetScale()) for (int i = 0; i < X.rows(); ++i) for (int j = 0; j < X.cols(); ++j) X(i, j) = (X(i, j) + mInputOffset[j]) * mInputScale[j]; }
void cNeuralNet::NormalizeInput(Eigen::VectorXd& x) const { Impl(mNet.get())->NormalizeInput(x); }
void cNeuralNet::NormalizeInputDiff(Eigen::VectorXd&) const { abort(); }
void cNeuralNet::UnnormalizeInput(Eigen::VectorXd& x) const { Impl(mNet.get())->UnnormalizeInput(x); }
void cNeuralNet::UnnormalizeInputDiff(Eigen::VectorXd&) const { abort(); }
void cNeuralNet::NormalizeOutput(Eigen::VectorXd& y) const { Impl(mNet.get())->NormalizeOutput(y); }
void cNeuralNet::NormalizeOutputDiff(Eigen::VectorXd&) const { abort(); }
void cNeuralNet::UnnormalizeOutput(Eigen::VectorXd& y) const { Impl(mNet.get())->UnnormalizeOutput(y); }
void cNeuralNet::UnnormalizeOutputDiff(Eigen::VectorXd&) const { abort(); }
void cNeuralNet::CopyModel(const cNeuralNet& other)
{
	assert(HasNet() && other.HasNet());
	Impl(mNet.get())->CopyModel(*Impl(other.mNet.get()));
	mInputOffset = other.GetInputOffset(); mInputScale = other.GetInputScale(); mOutputOffset = other.GetOutputOffset(); mOutputScale = other.GetOutputScale();
	mValidModel = true;
}
void cNeuralNet::LerpModel(const cNeuralNet&, double) { abort(); }
void cNeuralNet::BlendModel(const cNeuralNet&, double, double) { abort(); }
void cNeuralNet::BuildNetParams(caffe::NetParameter&) const {}
bool cNeuralNet::CompareModel(const cNeuralNet&) const { return false; }
void cNeuralNet::ForwardInjectNoisePrefilled(double, double, const std::string&, Eigen::VectorXd&) const { abort(); }
void cNeuralNet::GetLayerState(const std::string&, Eigen::VectorXd& s) const { s.resize(0); }
void cNeuralNet::SetLayerState(const Eigen::VectorXd&, const std::string&) const {}
const std::vector<caffe::Blob<cNeuralNet::tNNData>*>& cNeuralNet::GetParams() const
{
	static thread_local std::vector<caffe::Blob<tNNData>*> ids;      // (size = id + 1: how ref_learn_api.cpp names a net, as in the sibling library)
	ids.assign(static_cast<size_t>(HasNet() ? mNet->mHarnessId + 1 : 0), nullptr);
	return ids;
}
void cNeuralNet::SyncSolverParams() {}
void cNeuralNet::SyncNetParams() {}
void cNeuralNet::CopyGrad(const cNeuralNet&) { abort(); }
bool cNeuralNet::ValidOffsetScale() const { return mInputOffset.size() > 0 && mInputScale.size() > 0 && mOutputOffset.size() > 0 && mOutputScale.size() > 0; }
void cNeuralNet::InitOffsetScale()
{
	mInputOffset = Eigen::VectorXd::Zero(GetInputSize()); mInputScale = Eigen::VectorXd::Ones(GetInputSize());
	mOutputOffset = Eigen::VectorXd::Zero(GetOutputSize()); mOutputScale = Eigen::VectorXd::Ones(GetOutputSize());
}
void cNeuralNet::FetchOutput(const std::vector<caffe::Blob<tNNData>*>&, Eigen::VectorXd&) const {}
void cNeuralNet::FetchInput(Eigen::VectorXd&) const {}
boost::shared_ptr<caffe::Net<cNeuralNet::tNNData>> cNeuralNet::GetTrainNet() const { return nullptr; }
boost::shared_ptr<caffe::MemoryDataLayer<cNeuralNet::tNNData>> cNeuralNet::GetTrainDataLayer() const { return nullptr; }
void cNeuralNet::LoadTrainData(const Eigen::MatrixXd&, const Eigen::MatrixXd&) {}
bool cNeuralNet::WriteData(const Eigen::MatrixXd&, const Eigen::MatrixXd&, const std::string&) { return false; }
std::string cNeuralNet::GetOffsetScaleFile(const std::string& f) const { return f; }
void cNeuralNet::WriteOffsetScale(const std::string&) const {}
const std::string& cNeuralNet::GetInputLayerName() const { static const std::string s = "data"; return s; }
const std::string& cNeuralNet::GetOutputLayerName() const { static const std::string s = "output"; return s; }
