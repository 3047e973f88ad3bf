// ref_learn_net.cpp -- TEST INFRASTRUCTURE (oracle/_ref_build -> oracle/_ref/libref_learn.so), never part of the product.
//
// libref_learn.so = the reference's TRAINERS compiled unchanged from /root/reference/learning (NeuralNetTrainer.cpp, MACETrainer.cpp, QNetTrainer.cpp,
// ACTrainer.cpp, CaclaTrainer.cpp, NeuralNetLearner.cpp, ACLearner.cpp, ParamServer.cpp, TrainerInterface.cpp, ExpTuple.cpp) against the reference's OWN
// learning/NeuralNet.h. The one translation unit that cannot be compiled is learning/NeuralNet.cpp: it IS Caffe code (caffe::Net, caffe::Solver, HDF5), and Caffe is
// absent. This file defines cNeuralNet's member functions instead, over a network the test harness supplies through callbacks (oracle/reflearn.py: the numpy fp64 nets
// and the Caffe SGD rule of oracle/trainer_ref.py). RESTATED here from learning/NeuralNet.cpp, because everything around the forward / solver step is plain
// arithmetic the trainers depend on:
//   * the normalisation around every pass: x' = (x + InputOffset) * InputScale, label' = (y + OutputOffset) * OutputScale, y = y' / OutputScale - OutputOffset
//     (learning/NeuralNet.cpp:352-375, 443-512, 964-1054, 1077-1122);
//   * CalcOffsetScale: offset = -mean, scale = 1 / population standard deviation, 0 where the deviation is 0, accumulated row by row (:280-313);
//   * Train = LoadTrainData + StepSolver(mPassesPerStep x rows / batch) (:229-245), EvalBatch through the solver's net when there is one and its batch is > 1,
//     sample by sample through Eval otherwise (:377-387, 427-441); CopyModel = parameters + the four normaliser vectors (:722-733); LoadNet / LoadSolver initialise the
//     normalisers to (0, 1) when none are set (:62-79, 110-136, 919-928).
// A cNeuralNet holds ONE parameter vector in the harness: Caffe keeps a deploy net and the solver's train net and copies one into the other after every change
// (SyncNetParams / SyncSolverParams), which is the same thing seen from outside.
#include <cassert>
#include <cmath>
#include <cstdio>
#include <cstdlib>

#include "learning/NeuralNet.h"
#include "ref_learn_harness.h"

static RefLearnHarness gHarness;
extern "C" void ref_learn_set_harness(const RefLearnHarness* h) { gHarness = *h; }
const RefLearnHarness& RefLearnGetHarness() { return gHarness; }

// learning/NeuralNet.h only forward-declares cNNSolver (learning/NNSolver.h wraps caffe::SGDSolver and is not compiled): here the solver is the harness' step
class cNNSolver
{
public:
	int mBatchSize = 0;
	std::string mFile;
};

std::mutex cNeuralNet::gOutputLock;

cNeuralNet::tProblem::tProblem()
{
	mX.resize(0, 0);
	mY.resize(0, 0);
	mPassesPerStep = 100;
}
bool cNeuralNet::tProblem::HasData() const { return mX.size() > 0; }

cNeuralNet::cCaffeNetWrapper::cCaffeNetWrapper(const std::string& net_file, caffe::Phase phase) : caffe::Net<tNNData>(net_file, phase)
{
	mHarnessId = gHarness.net_new ? gHarness.net_new(net_file.c_str()) : -1;
}
cNeuralNet::cCaffeNetWrapper::~cCaffeNetWrapper()
{
	if (mHarnessId >= 0 && gHarness.net_free) gHarness.net_free(mHarnessId);
}
int cNeuralNet::cCaffeNetWrapper::GetLayerIdx(const std::string&) const { return -1; }


cNeuralNet::cNeuralNet() { Clear(); mAsync = false; }
cNeuralNet::~cNeuralNet() {}

void cNeuralNet::LoadNet(const std::string& net_file)
{
	if (net_file != "") {
		Clear();
		mNet = std::unique_ptr<cCaffeNetWrapper>(new cCaffeNetWrapper(net_file, caffe::TEST));
		if (!ValidOffsetScale()) InitOffsetScale();
		if (HasSolver()) SyncNetParams();
	}
}
void cNeuralNet::LoadModel(const std::string& model_file)
{
	if (model_file != "") { assert(HasNet()); mValidModel = true; }   // (HDF5 blobs: the harness sets weights directly)
}
void cNeuralNet::LoadSolver(const std::string& solver_file, bool async)
{
	if (solver_file != "") {
		assert(HasNet());            // every trainer of the reference calls LoadNet first (BuildNetPool, BuildActor)
		mSolverFile = solver_file;
		mAsync = async;
		mSolver = std::make_shared<cNNSolver>();
		mSolver->mFile = solver_file;
		mSolver->mBatchSize = gHarness.solver_load(mNet->mHarnessId, solver_file.c_str());
		if (!ValidOffsetScale()) InitOffsetScale();
		if (HasNet()) SyncSolverParams();
	}
}
void cNeuralNet::LoadScale(const std::string&) {}
void cNeuralNet::Clear()
{
	mNet.reset();
	mSolver.reset();
	mValidModel = false;
	mInputOffset.resize(0); mInputScale.resize(0); mOutputOffset.resize(0); mOutputScale.resize(0);
}

void cNeuralNet::Train(const tProblem& prob)
{
	if (!HasSolver()) { printf("Solver has not been initialized\n"); assert(false); return; }
	const int batch_size = GetBatchSize();
	const int num_batches = static_cast<int>(prob.mX.rows()) / batch_size;
	// LoadTrainData (one batch: learning/NeuralNet.cpp:1083-1085 asserts it) + StepSolver(passes x batches)
	const int n = batch_size, dx = static_cast<int>(prob.mX.cols()), dy = static_cast<int>(prob.mY.cols());
	std::vector<double> data(static_cast<size_t>(n) * dx), labels(static_cast<size_t>(n) * dy);
	for (int i = 0; i < n; ++i) {
		for (int j = 0; j < dx; ++j) { double v = prob.mX(i, j); if (ValidOffsetScale()) { v += mInputOffset[j]; v = v * mInputScale[j]; } data[static_cast<size_t>(i) * dx + j] = v; }
		for (int j = 0; j < dy; ++j) { double v = prob.mY(i, j); if (ValidOffsetScale()) { v += mOutputOffset[j]; v = v * mOutputScale[j]; } labels[static_cast<size_t>(i) * dy + j] = v; }
	}
	gHarness.net_step(mNet->mHarnessId, data.data(), labels.data(), n, prob.mPassesPerStep * num_batches);
	if (HasNet()) SyncNetParams();
	mValidModel = true;
}
double cNeuralNet::ForwardBackward(const tProblem&) { fprintf(stderr, "libref_learn: cNeuralNet::ForwardBackward (asynchronous trainers) is outside the harness\n"); abort(); }
void cNeuralNet::StepSolver(int) { fprintf(stderr, "libref_learn: cNeuralNet::StepSolver without data is outside the harness\n"); abort(); }
void cNeuralNet::ResetSolver()
{
	mSolver.reset();
	LoadSolver(mSolverFile, mAsync);
	if (gHarness.solver_reset) gHarness.solver_reset(mNet->mHarnessId);
}

void cNeuralNet::CalcOffsetScale(const Eigen::MatrixXd& X, Eigen::VectorXd& out_offset, Eigen::VectorXd& out_scale) const
{
	const int num_pts = static_cast<int>(X.rows());
	assert(num_pts > 1);
	const double norm = 1.0 / num_pts;
	const int input_size = GetInputSize();
	out_offset = Eigen::VectorXd::Zero(input_size);
	out_scale = Eigen::VectorXd::Zero(input_size);
	for (int i = 0; i < num_pts; ++i) for (int j = 0; j < input_size; ++j) out_offset[j] += norm * X(i, j);
	for (int i = 0; i < num_pts; ++i) for (int j = 0; j < input_size; ++j) { const double c = X(i, j) - out_offset[j]; out_scale[j] += norm * (c * c); }
	for (int j = 0; j < input_size; ++j) {
		out_offset[j] = -out_offset[j];
		double val = std::sqrt(out_scale[j]);
		out_scale[j] = (val == 0) ? 0 : (1 / val);
	}
}
void cNeuralNet::SetInputOffsetScale(const Eigen::VectorXd& offset, const Eigen::VectorXd& scale)
{
	assert(offset.size() == GetInputSize() && scale.size() == GetInputSize());
	mInputOffset = offset; mInputScale = scale;
}
void cNeuralNet::SetOutputOffsetScale(const Eigen::VectorXd& offset, const Eigen::VectorXd& scale)
{
	assert(offset.size() == GetOutputSize() && scale.size() == GetOutputSize());
	mOutputOffset = offset; mOutputScale = scale;
}
const Eigen::VectorXd& cNeuralNet::GetInputOffset() const { return mInputOffset; }
const Eigen::VectorXd& cNeuralNet::GetInputScale() const { return mInputScale; }
const Eigen::VectorXd& cNeuralNet::GetOutputOffset() const { return mOutputOffset; }
const Eigen::VectorXd& cNeuralNet::GetOutputScale() const { return mOutputScale; }

void cNeuralNet::Eval(const Eigen::VectorXd& x, Eigen::VectorXd& out_y) const
{
	assert(HasNet() && x.size() == GetInputSize());
	Eigen::VectorXd norm_x = x;
	NormalizeInput(norm_x);
	const int in = GetInputSize(), out = GetOutputSize();
	std::vector<double> xi(in), yo(out);
	for (int i = 0; i < in; ++i) xi[i] = norm_x[i];
	gHarness.net_forward(mNet->mHarnessId, xi.data(), 1, yo.data());
	out_y.resize(out);
	for (int i = 0; i < out; ++i) out_y[i] = yo[i];
	UnnormalizeOutput(out_y);     // FetchOutput (learning/NeuralNet.cpp:930-945)
}
void cNeuralNet::EvalBatch(const Eigen::MatrixXd& X, Eigen::MatrixXd& out_Y) const
{
	if (HasSolver() && GetBatchSize() > 1) EvalBatchSolver(X, out_Y); else EvalBatchNet(X, out_Y);
}
void cNeuralNet::Backward(const Eigen::VectorXd&, Eigen::VectorXd&) const { fprintf(stderr, "libref_learn: cNeuralNet::Backward (cCaclaTrainer's PTD mode) is outside the harness\n"); abort(); }
void cNeuralNet::EvalBatchNet(const Eigen::MatrixXd& X, Eigen::MatrixXd& out_Y) const
{
	assert(HasNet());
	const int num_data = static_cast<int>(X.rows());
	Eigen::VectorXd x, y;
	out_Y.resize(num_data, GetOutputSize());
	for (int i = 0; i < num_data; ++i) { x = X.row(i); Eval(x, y); out_Y.row(i) = y; }
}
void cNeuralNet::EvalBatchSolver(const Eigen::MatrixXd& X, Eigen::MatrixXd& out_Y) const
{
	assert(HasSolver());
	const int input_size = GetInputSize(), output_size = GetOutputSize();
	assert(X.cols() == input_size);
	const int num_data = static_cast<int>(X.rows());
	out_Y.resize(num_data, output_size);
	std::vector<double> data(static_cast<size_t>(num_data) * input_size), res(static_cast<size_t>(num_data) * output_size);
	for (int i = 0; i < num_data; ++i) for (int j = 0; j < input_size; ++j) {
		double val = X(i, j);
		if (ValidOffsetScale()) { val += mInputOffset[j]; val = val * mInputScale[j]; }
		data[static_cast<size_t>(i) * input_size + j] = val;
	}
	gHarness.net_forward(mNet->mHarnessId, data.data(), num_data, res.data());     // (the reference feeds it batch by batch; rows are independent)
	for (int i = 0; i < num_data; ++i) for (int j = 0; j < output_size; ++j) {
		double val = res[static_cast<size_t>(i) * output_size + j];
		if (ValidOffsetScale()) { val /= mOutputScale[j]; val -= mOutputOffset[j]; }
		out_Y(i, j) = val;
	}
}

int cNeuralNet::GetInputSize() const { int in = 0, out = 0; if (HasNet()) gHarness.net_dims(mNet->mHarnessId, &in, &out); return in; }
int cNeuralNet::GetOutputSize() const { int in = 0, out = 0; if (HasNet()) gHarness.net_dims(mNet->mHarnessId, &in, &out); return out; }
int cNeuralNet::GetBatchSize() const { return HasSolver() ? mSolver->mBatchSize : 0; }
int cNeuralNet::CalcNumParams() const { return 0; }
void cNeuralNet::OutputModel(const std::string&) const {}
void cNeuralNet::PrintParams() const {}
bool cNeuralNet::HasNet() const { return mNet != nullptr; }
bool cNeuralNet::HasSolver() const { return mSolver != nullptr; }
bool cNeuralNet::HasLayer(const std::string) const { return false; }
bool cNeuralNet::HasValidModel() const { return mValidModel; }

void cNeuralNet::NormalizeInput(Eigen::MatrixXd& X) const
{
	if (ValidOffsetScale()) for (int i = 0; i < X.rows(); ++i) for (int j = 0; j < X.cols(); ++j) X(i, j) = (X(i, j) + mInputOffset[j]) * mInputScale[j];
}
void cNeuralNet::NormalizeInput(Eigen::VectorXd& x) const { if (ValidOffsetScale()) for (int i = 0; i < x.size(); ++i) x[i] = (x[i] + mInputOffset[i]) * mInputScale[i]; }
void cNeuralNet::NormalizeInputDiff(Eigen::VectorXd& d) const { if (ValidOffsetScale()) for (int i = 0; i < d.size(); ++i) d[i] = d[i] * mInputScale[i]; }
void cNeuralNet::UnnormalizeInput(Eigen::VectorXd& x) const { if (ValidOffsetScale()) for (int i = 0; i < x.size(); ++i) x[i] = x[i] / mInputScale[i] - mInputOffset[i]; }
void cNeuralNet::UnnormalizeInputDiff(Eigen::VectorXd& d) const { if (ValidOffsetScale()) for (int i = 0; i < d.size(); ++i) d[i] = d[i] / mInputScale[i]; }
void cNeuralNet::NormalizeOutput(Eigen::VectorXd& y) const { if (ValidOffsetScale()) for (int i = 0; i < y.size(); ++i) y[i] = (y[i] + mOutputOffset[i]) * mOutputScale[i]; }
void cNeuralNet::NormalizeOutputDiff(Eigen::VectorXd& d) const { if (ValidOffsetScale()) for (int i = 0; i < d.size(); ++i) d[i] = d[i] * mOutputScale[i]; }
void cNeuralNet::UnnormalizeOutput(Eigen::VectorXd& y) const { if (ValidOffsetScale()) for (int i = 0; i < y.size(); ++i) y[i] = y[i] / mOutputScale[i] - mOutputOffset[i]; }
void cNeuralNet::UnnormalizeOutputDiff(Eigen::VectorXd& d) const { if (ValidOffsetScale()) for (int i = 0; i < d.size(); ++i) d[i] = d[i] / mOutputScale[i]; }

void cNeuralNet::CopyModel(const cNeuralNet& other)
{
	assert(HasNet() && other.HasNet());
	gHarness.net_copy(mNet->mHarnessId, other.mNet->mHarnessId);      // CopyParams(other.GetParams(), GetParams())
	mInputOffset = other.GetInputOffset(); mInputScale = other.GetInputScale();
	mOutputOffset = other.GetOutputOffset(); mOutputScale = other.GetOutputScale();
	SyncSolverParams();
	mValidModel = true;
}
void cNeuralNet::LerpModel(const cNeuralNet&, double) { abort(); }
void cNeuralNet::BlendModel(const cNeuralNet&, double, double) { abort(); }
void cNeuralNet::BuildNetParams(caffe::NetParameter&) const {}
bool cNeuralNet::CompareModel(const cNeuralNet&) const { return false; }
void cNeuralNet::ForwardInjectNoisePrefilled(double, double, const std::string&, Eigen::VectorXd&) const { abort(); }
void cNeuralNet::GetLayerState(const std::string&, Eigen::VectorXd& out_state) const { out_state.resize(0); }
void cNeuralNet::SetLayerState(const Eigen::VectorXd&, const std::string&) const {}
// the parameter blobs live in the harness; the one caller in the compiled sources is CopyModel above. The returned vector's SIZE carries the harness id so that the
// C API (ref_learn_api.cpp) can name a net without touching cNeuralNet's protected members
const std::vector<caffe::Blob<cNeuralNet::tNNData>*>& cNeuralNet::GetParams() const
{
	static thread_local std::vector<caffe::Blob<tNNData>*> ids;
	ids.assign(static_cast<size_t>(HasNet() ? mNet->mHarnessId + 1 : 0), nullptr);
	return ids;
}
void cNeuralNet::SyncSolverParams() {}
void cNeuralNet::SyncNetParams() {}
void cNeuralNet::CopyGrad(const cNeuralNet&) { abort(); }
bool cNeuralNet::ValidOffsetScale() const { return mInputOffset.size() > 0 && mInputScale.size() > 0 && mOutputOffset.size() > 0 && mOutputScale.size() > 0; }
void cNeuralNet::InitOffsetScale()
{
	const int input_size = GetInputSize();
	mInputOffset = Eigen::VectorXd::Zero(input_size); mInputScale = Eigen::VectorXd::Ones(input_size);
	const int output_size = GetOutputSize();
	mOutputOffset = Eigen::VectorXd::Zero(output_size); mOutputScale = Eigen::VectorXd::Ones(output_size);
}
void cNeuralNet::FetchOutput(const std::vector<caffe::Blob<tNNData>*>&, Eigen::VectorXd&) const {}
void cNeuralNet::FetchInput(Eigen::VectorXd&) const {}
boost::shared_ptr<caffe::Net<cNeuralNet::tNNData>> cNeuralNet::GetTrainNet() const { return nullptr; }
boost::shared_ptr<caffe::MemoryDataLayer<cNeuralNet::tNNData>> cNeuralNet::GetTrainDataLayer() const { return nullptr; }
void cNeuralNet::LoadTrainData(const Eigen::MatrixXd&, const Eigen::MatrixXd&) {}
bool cNeuralNet::WriteData(const Eigen::MatrixXd&, const Eigen::MatrixXd&, const std::string&) { return false; }
std::string cNeuralNet::GetOffsetScaleFile(const std::string& model_file) const { return model_file; }
void cNeuralNet::WriteOffsetScale(const std::string&) const {}
const std::string& cNeuralNet::GetInputLayerName() const { static const std::string s = "data"; return s; }
const std::string& cNeuralNet::GetOutputLayerName() const { static const std::string s = "output"; return s; }
