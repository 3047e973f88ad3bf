// BatchNeuralNet.h -- header-only C++ shim that puts the MI355X-native trainer step (libdtrl.so, include/dtrl_trainer.h) behind the reference's own
// network interface, so that cNeuralNetTrainer / cMACETrainer / cQNetTrainer / cCaclaTrainer -- which never touch Caffe themselves, only cNeuralNet
// (learning/NeuralNetTrainer.cpp:696-784, learning/MACETrainer.cpp:163-250, 346-372, 577-633) -- keep ALL their bookkeeping and swap only their nets.
//
// cBatchNeuralNet carries cNeuralNet's method names, argument types and error behaviour (learning/NeuralNet.h:10-140):
//   LoadNet / LoadSolver / LoadModel / Clear            the two prototxt paths go to dtrl_trainer_create_from_files (topology, MemoryData batch size, per-blob
//                                                        lr_mult / decay_mult, solver constants); a fresh net holds Caffe's xavier fill
//   Eval / EvalBatch                                     learning/NeuralNet.cpp:352-387, 427-512: normalise, forward, un-normalise -- on the device
//   Train(tProblem)                                      :229-245, 1077-1122: LoadTrainData + StepSolver(mPassesPerStep x rows / batch) -> that many Caffe SGD steps
//   CalcOffsetScale, Set / Get{Input, Output}OffsetScale :280-350
//   CopyModel                                            :722-733: parameters + normalisers, device to device
//   GetInputSize / GetOutputSize / GetBatchSize, HasNet / HasSolver / HasValidModel, ResetSolver, OutputModel (raw float32 blob + <file>_scale.txt)
// and two additions a caller of the rollout engine needs: GetTrainer() (the dtrl_trainer handle, e.g. for dtrl_trainer_params_device -> dtrl_set_policy_device)
// and GetParamsFlat / SetParamsFlat (the Caffe-blob-order weight vector).
// Not provided (they are Caffe objects): GetParams() blobs, BuildNetParams, ForwardBackward / CopyGrad / StepSolver of the asynchronous trainers (their
// data-parallel counterpart is dtrl_trainer_*_grad / dtrl_trainer_apply_grad), layer-state accessors, Backward (only cCaclaTrainer's TD / PTD modes would need it, and nothing in the reference selects them: dead code).
//
// Written against the reference's headers as they are (util/MathUtil.h brings Eigen in); a maintainer compiles it inside the reference tree. Two ways to use it:
//   (a) new code holds cBatchNeuralNet objects directly;
//   (b) the reference's trainers unchanged: build learning/NeuralNet.h's cNeuralNet member functions as forwards to a cBatchNeuralNet (INTEGRATION.md 4b;
//       oracle/_ref_build/ref_learn_net_native.cpp does exactly that for the tests, and tests/test_reference_learn.py runs the reference's own cMACETrainer on it).
// tests/shim/drive_shim_net.cpp compiles this header inside /root/reference's header tree and drives it on the plain-loop check build and on the HIP library.
#pragma once
#include <cassert>
#include <cmath>
#include <cstdio>
#include <string>
#include <vector>

#include "dtrl_trainer.h"
#include "util/MathUtil.h"

class cBatchNeuralNet
{
public:
	typedef double tNNData;
	struct tProblem
	{
		tProblem() : mPassesPerStep(100) { mX.resize(0, 0); mY.resize(0, 0); }
		Eigen::MatrixXd mX;
		Eigen::MatrixXd mY;
		int mPassesPerStep;
		bool HasData() const { return mX.size() > 0; }
	};

	explicit cBatchNeuralNet(int device_id = -1, const std::string& data_root = "")
		: mDeviceID(device_id), mDataRoot(data_root), mTrainer(nullptr), mHasSolver(false), mValidModel(false), mDiscount(0.9), mFreezeTarget(0), mSeed(0x5eed) {}
	virtual ~cBatchNeuralNet() { Clear(); }
	cBatchNeuralNet(const cBatchNeuralNet&) = delete;
	cBatchNeuralNet& operator=(const cBatchNeuralNet&) = delete;

	// what the MACE step calls need beyond the two files (cTrainerInterface::tParams::mDiscount, mFreezeTargetIters); set before LoadNet / LoadSolver
	virtual void SetTrainerParams(double discount, bool freeze_target) { mDiscount = discount; mFreezeTarget = freeze_target ? 1 : 0; }
	virtual void SetInitSeed(unsigned long long seed) { mSeed = seed; }

	virtual void LoadNet(const std::string& net_file)
	{
		if (net_file == "") return;
		Clear();
		mNetFile = net_file;
		Build();
	}
	virtual void LoadSolver(const std::string& solver_file, bool async = false)
	{
		if (solver_file == "") return;
		if (async) { printf("cBatchNeuralNet: asynchronous solvers are not part of the native step\n"); assert(false); return; }
		mSolverFile = solver_file;
		// the trainer is rebuilt with the solver's constants and the train net's batch size / multipliers; weights and normalisers are carried over
		std::vector<float> w; Eigen::VectorXd io = mInputOffset, is = mInputScale, oo = mOutputOffset, os = mOutputScale;
		const bool had = mTrainer != nullptr;
		if (had) GetParamsFlat(w);
		Build();
		if (had && mTrainer && static_cast<long long>(w.size()) == dtrl_trainer_num_params(mTrainer)) { SetParamsFlat(w); if (io.size() > 0) { SetInputOffsetScale(io, is); SetOutputOffsetScale(oo, os); } }
		mHasSolver = mTrainer != nullptr;
	}
	// the trained blobs of the reference are Caffe HDF5 files; here: a raw little-endian float32 vector in Caffe blob order (what OutputModel writes), plus <file>_scale.txt
	virtual void LoadModel(const std::string& model_file)
	{
		if (model_file == "" || !HasNet()) return;
		FILE* f = fopen(model_file.c_str(), "rb");
		if (!f) { printf("cBatchNeuralNet: cannot open %s\n", model_file.c_str()); return; }
		std::vector<float> w(static_cast<size_t>(dtrl_trainer_num_params(mTrainer)));
		const size_t got = fread(w.data(), sizeof(float), w.size(), f);
		fclose(f);
		if (got != w.size()) { printf("cBatchNeuralNet: %s holds %zu parameters, the net %zu\n", model_file.c_str(), got, w.size()); return; }
		SetParamsFlat(w);
		mValidModel = true;
	}
	virtual void Clear()
	{
		if (mTrainer) dtrl_trainer_destroy(mTrainer);
		mTrainer = nullptr; mHasSolver = false; mValidModel = false;
		mInputOffset.resize(0); mInputScale.resize(0); mOutputOffset.resize(0); mOutputScale.resize(0);
	}

	virtual void Train(const tProblem& prob)
	{
		if (!HasSolver()) { printf("Solver has not been initialized\n"); assert(false); return; }
		const int batch = GetBatchSize();
		const int num_batches = static_cast<int>(prob.mX.rows()) / batch;
		assert(num_batches == 1);                                   // learning/NeuralNet.cpp:1083-1085
		std::vector<double> x, y;
		Flatten(prob.mX, batch, x); Flatten(prob.mY, batch, y);
		for (int it = 0; it < prob.mPassesPerStep * num_batches; ++it) Check(dtrl_trainer_step_host(mTrainer, x.data(), y.data(), &mLastLoss), "Train");
		mValidModel = true;
	}
	virtual double GetLastLoss() const { return mLastLoss; }
	virtual void ResetSolver()
	{
		if (!mTrainer) return;
		std::vector<float> zero(static_cast<size_t>(dtrl_trainer_num_params(mTrainer)), 0.0f);
		Check(dtrl_trainer_set_params(mTrainer, 2, zero.data(), static_cast<int64_t>(zero.size())), "ResetSolver");
	}
	virtual void CalcOffsetScale(const Eigen::MatrixXd& X, Eigen::VectorXd& out_offset, Eigen::VectorXd& out_scale) const
	{
		const int num_pts = static_cast<int>(X.rows());
		assert(num_pts > 1);
		const double norm = 1.0 / num_pts;
		const int input_size = GetInputSize();
		out_offset = Eigen::VectorXd::Zero(input_size);
		out_scale = Eigen::VectorXd::Zero(input_size);
		for (int i = 0; i < num_pts; ++i) for (int j = 0; j < input_size; ++j) out_offset[j] += norm * X(i, j);
		for (int i = 0; i < num_pts; ++i) for (int j = 0; j < input_size; ++j) { const double c = X(i, j) - out_offset[j]; out_scale[j] += norm * (c * c); }
		for (int j = 0; j < input_size; ++j) { out_offset[j] = -out_offset[j]; const double v = std::sqrt(out_scale[j]); out_scale[j] = (v == 0) ? 0 : (1 / v); }
	}
	virtual void SetInputOffsetScale(const Eigen::VectorXd& offset, const Eigen::VectorXd& scale)
	{
		assert(offset.size() == GetInputSize() && scale.size() == GetInputSize());
		mInputOffset = offset; mInputScale = scale;
		PushNorm();
	}
	virtual void SetOutputOffsetScale(const Eigen::VectorXd& offset, const Eigen::VectorXd& scale)
	{
		assert(offset.size() == GetOutputSize() && scale.size() == GetOutputSize());
		mOutputOffset = offset; mOutputScale = scale;
		PushNorm();
	}
	virtual const Eigen::VectorXd& GetInputOffset() const { return mInputOffset; }
	virtual const Eigen::VectorXd& GetInputScale() const { return mInputScale; }
	virtual const Eigen::VectorXd& GetOutputOffset() const { return mOutputOffset; }
	virtual const Eigen::VectorXd& GetOutputScale() const { return mOutputScale; }

	virtual void Eval(const Eigen::VectorXd& x, Eigen::VectorXd& out_y) const
	{
		assert(HasNet() && x.size() == GetInputSize());
		std::vector<double> xi(static_cast<size_t>(x.size())), yo(static_cast<size_t>(GetOutputSize()));
		for (int i = 0; i < x.size(); ++i) xi[i] = x[i];
		Check(dtrl_trainer_eval_host(mTrainer, 0, xi.data(), 1, yo.data()), "Eval");
		out_y.resize(GetOutputSize());
		for (int i = 0; i < out_y.size(); ++i) out_y[i] = yo[i];
	}
	virtual void EvalBatch(const Eigen::MatrixXd& X, Eigen::MatrixXd& out_Y) const
	{
		assert(HasNet() && X.cols() == GetInputSize());
		const int n = static_cast<int>(X.rows()), out = GetOutputSize();
		std::vector<double> x, y(static_cast<size_t>(n) * out);
		Flatten(X, n, x);
		Check(dtrl_trainer_eval_host(mTrainer, 0, x.data(), n, y.data()), "EvalBatch");
		out_Y.resize(n, out);
		for (int i = 0; i < n; ++i) for (int j = 0; j < out; ++j) out_Y(i, j) = y[static_cast<size_t>(i) * out + j];
	}

	virtual int GetInputSize() const { int v = 0; if (mTrainer) dtrl_trainer_dims(mTrainer, &v, nullptr, nullptr, nullptr); return v; }
	virtual int GetOutputSize() const { int v = 0; if (mTrainer) dtrl_trainer_dims(mTrainer, nullptr, &v, nullptr, nullptr); return v; }
	virtual int GetBatchSize() const { int v = 0; if (mTrainer && mHasSolver) dtrl_trainer_dims(mTrainer, nullptr, nullptr, &v, nullptr); return v; }
	virtual int CalcNumParams() const { return mTrainer ? static_cast<int>(dtrl_trainer_num_params(mTrainer)) : 0; }

	virtual void OutputModel(const std::string& out_file) const
	{
		if (!mTrainer) return;
		std::vector<float> w; GetParamsFlat(w);
		FILE* f = fopen(out_file.c_str(), "wb");
		if (!f) { printf("cBatchNeuralNet: cannot write %s\n", out_file.c_str()); return; }
		fwrite(w.data(), sizeof(float), w.size(), f); fclose(f);
		// the sibling scale file of cNeuralNet::WriteOffsetScale (learning/NeuralNet.cpp:1182-1205): JSON {InputOffset, InputScale, OutputOffset, OutputScale}
		const size_t dot = out_file.find_last_of('.');
		const std::string scale_file = (dot == std::string::npos ? out_file : out_file.substr(0, dot)) + "_scale.txt";
		FILE* g = fopen(scale_file.c_str(), "w");
		if (!g) return;
		auto arr = [&](const char* key, const Eigen::VectorXd& v, bool last) { fprintf(g, "\"%s\": [", key); for (int i = 0; i < v.size(); ++i) fprintf(g, "%s%.17g", i ? ", " : "", v[i]); fprintf(g, "]%s\n", last ? "" : ","); };
		fprintf(g, "{\n"); arr("InputOffset", mInputOffset, false); arr("InputScale", mInputScale, false); arr("OutputOffset", mOutputOffset, false); arr("OutputScale", mOutputScale, true); fprintf(g, "}\n");
		fclose(g);
	}
	virtual bool HasNet() const { return mTrainer != nullptr; }
	virtual bool HasSolver() const { return mTrainer != nullptr && mHasSolver; }
	virtual bool HasLayer(const std::string) const { return false; }
	virtual bool HasValidModel() const { return mValidModel; }

	virtual void NormalizeInput(Eigen::VectorXd& x) const { if (ValidOffsetScale()) for (int i = 0; i < x.size(); ++i) x[i] = (x[i] + mInputOffset[i]) * mInputScale[i]; }
	virtual void UnnormalizeInput(Eigen::VectorXd& x) const { if (ValidOffsetScale()) for (int i = 0; i < x.size(); ++i) x[i] = x[i] / mInputScale[i] - mInputOffset[i]; }
	virtual void NormalizeOutput(Eigen::VectorXd& y) const { if (ValidOffsetScale()) for (int i = 0; i < y.size(); ++i) y[i] = (y[i] + mOutputOffset[i]) * mOutputScale[i]; }
	virtual void UnnormalizeOutput(Eigen::VectorXd& y) const { if (ValidOffsetScale()) for (int i = 0; i < y.size(); ++i) y[i] = y[i] / mOutputScale[i] - mOutputOffset[i]; }

	virtual void CopyModel(const cBatchNeuralNet& other)
	{
		assert(HasNet() && other.HasNet());
		Check(dtrl_trainer_copy_model(mTrainer, other.mTrainer), "CopyModel");
		mInputOffset = other.GetInputOffset(); mInputScale = other.GetInputScale();
		mOutputOffset = other.GetOutputOffset(); mOutputScale = other.GetOutputScale();
		mValidModel = true;
	}

	// ---- additions ----
	virtual dtrl_trainer* GetTrainer() const { return mTrainer; }
	virtual void GetParamsFlat(std::vector<float>& out_w) const
	{
		out_w.assign(mTrainer ? static_cast<size_t>(dtrl_trainer_num_params(mTrainer)) : 0, 0.0f);
		if (mTrainer) Check(dtrl_trainer_get_params(mTrainer, 0, out_w.data(), static_cast<int64_t>(out_w.size())), "GetParamsFlat");
	}
	virtual void SetParamsFlat(const std::vector<float>& w)
	{
		if (!mTrainer) return;
		Check(dtrl_trainer_set_params(mTrainer, 0, w.data(), static_cast<int64_t>(w.size())), "SetParamsFlat");
		mValidModel = true;
	}

protected:
	int mDeviceID;
	std::string mDataRoot, mNetFile, mSolverFile;
	dtrl_trainer* mTrainer;
	bool mHasSolver, mValidModel;
	double mDiscount;
	int mFreezeTarget;
	unsigned long long mSeed;
	mutable double mLastLoss = 0;
	Eigen::VectorXd mInputOffset, mInputScale, mOutputOffset, mOutputScale;

	virtual bool ValidOffsetScale() const { return mInputOffset.size() > 0 && mInputScale.size() > 0 && mOutputOffset.size() > 0 && mOutputScale.size() > 0; }
	virtual void Build()
	{
		if (mTrainer) { dtrl_trainer_destroy(mTrainer); mTrainer = nullptr; }
		const int rc = dtrl_trainer_create_from_files(mNetFile.c_str(), mSolverFile.empty() ? nullptr : mSolverFile.c_str(), mDataRoot.empty() ? nullptr : mDataRoot.c_str(),
		                                             static_cast<float>(mDiscount), mFreezeTarget, mDeviceID, &mTrainer);
		if (rc != 0) { printf("cBatchNeuralNet: %s\n", dtrl_trainer_last_error(nullptr)); mTrainer = nullptr; return; }
		Check(dtrl_trainer_init_xavier(mTrainer, mSeed), "init");
		if (!ValidOffsetScale()) {     // cNeuralNet::InitOffsetScale (learning/NeuralNet.cpp:919-928)
			mInputOffset = Eigen::VectorXd::Zero(GetInputSize()); mInputScale = Eigen::VectorXd::Ones(GetInputSize());
			int out = 0; dtrl_trainer_dims(mTrainer, nullptr, &out, nullptr, nullptr);
			mOutputOffset = Eigen::VectorXd::Zero(out); mOutputScale = Eigen::VectorXd::Ones(out);
		}
		PushNorm();
	}
	virtual void PushNorm()
	{
		if (!mTrainer || !ValidOffsetScale()) return;
		std::vector<double> a(mInputOffset.size()), b(mInputScale.size()), c(mOutputOffset.size()), d(mOutputScale.size());
		for (int i = 0; i < mInputOffset.size(); ++i) { a[i] = mInputOffset[i]; b[i] = mInputScale[i]; }
		for (int i = 0; i < mOutputOffset.size(); ++i) { c[i] = mOutputOffset[i]; d[i] = mOutputScale[i]; }
		Check(dtrl_trainer_set_normalizers(mTrainer, a.data(), b.data(), c.data(), d.data()), "normalisers");
	}
	static void Flatten(const Eigen::MatrixXd& M, int rows, std::vector<double>& out)
	{
		const int cols = static_cast<int>(M.cols());
		out.resize(static_cast<size_t>(rows) * cols);
		for (int i = 0; i < rows; ++i) for (int j = 0; j < cols; ++j) out[static_cast<size_t>(i) * cols + j] = M(i, j);
	}
	void Check(int rc, const char* what) const
	{
		if (rc != 0) printf("cBatchNeuralNet::%s failed (%d): %s\n", what, rc, dtrl_trainer_last_error(mTrainer));   // the reference's convention: print, no exception
		assert(rc == 0);
	}
};
