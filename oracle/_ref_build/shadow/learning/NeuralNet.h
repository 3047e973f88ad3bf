// Stand-in for the reference's learning/NeuralNet.h -- TEST INFRASTRUCTURE (oracle/_ref_build), never part of the product.
//
// The reference's cNeuralNet wraps Caffe (learning/NeuralNet.cpp includes caffe/caffe.hpp; Caffe is absent). The controllers built into
// libref_sim.so only need a network they can ask for its sizes, evaluate and normalise with, so this header -- found BEFORE the reference's own
// by the include path order of oracle/_ref_build/Makefile -- declares a cNeuralNet with the same method names whose forward pass is a callback
// the test harness installs (the oracle's forward, or a table of precomputed outputs). RESTATED here, because their home translation unit
// cannot be built: the normalisation around the forward (learning/NeuralNet.cpp:352-375, 977-986, 1027-1036): x' = (x + InputOffset) * InputScale,
// y = y' / OutputScale - OutputOffset.
#pragma once
#include <functional>
#include <string>
#include "util/MathUtil.h"

class cNeuralNet
{
public:
	typedef double tNNData;
	// harness hooks: sizes of the net "loaded" by LoadNet and its raw forward (normalised input -> normalised output)
	struct tHarness { int mInputSize = 0; int mOutputSize = 0; std::function<void(const Eigen::VectorXd&, Eigen::VectorXd&)> mForward; };
	static tHarness& Harness() { static tHarness h; return h; }

	cNeuralNet() : mHasNet(false), mValidModel(false) {}
	virtual ~cNeuralNet() {}

	virtual void LoadNet(const std::string& net_file)
	{
		mHasNet = (net_file != "") && Harness().mInputSize > 0;
		if (mHasNet) InitOffsetScale();
	}
	virtual void LoadModel(const std::string& model_file) { mValidModel = HasNet(); }
	virtual void LoadScale(const std::string& scale_file) {}
	virtual void Clear() { mHasNet = false; mValidModel = false; mInputOffset.resize(0); mInputScale.resize(0); mOutputOffset.resize(0); mOutputScale.resize(0); }

	virtual void SetInputOffsetScale(const Eigen::VectorXd& offset, const Eigen::VectorXd& scale) { mInputOffset = offset; mInputScale = scale; }
	virtual void SetOutputOffsetScale(const Eigen::VectorXd& offset, const Eigen::VectorXd& scale) { mOutputOffset = offset; mOutputScale = scale; }
	virtual const Eigen::VectorXd& GetInputOffset() const { return mInputOffset; }
	virtual const Eigen::VectorXd& GetInputScale() const { return mInputScale; }
	virtual const Eigen::VectorXd& GetOutputOffset() const { return mOutputOffset; }
	virtual const Eigen::VectorXd& GetOutputScale() const { return mOutputScale; }

	virtual void Eval(const Eigen::VectorXd& x, Eigen::VectorXd& out_y) const
	{
		Eigen::VectorXd norm_x = x;
		NormalizeInput(norm_x);
		Harness().mForward(norm_x, out_y);
		UnnormalizeOutput(out_y);
	}
	virtual int GetInputSize() const { return HasNet() ? Harness().mInputSize : 0; }
	virtual int GetOutputSize() const { return HasNet() ? Harness().mOutputSize : 0; }
	virtual void OutputModel(const std::string& out_file) const {}
	virtual bool HasNet() const { return mHasNet; }
	virtual bool HasLayer(const std::string layer_name) const { return false; }
	virtual bool HasValidModel() const { return mValidModel; }

	virtual void NormalizeInput(Eigen::VectorXd& x) const { for (int i = 0; i < static_cast<int>(x.size()); ++i) x[i] = (x[i] + mInputOffset[i]) * mInputScale[i]; }
	virtual void UnnormalizeInput(Eigen::VectorXd& x) const { for (int i = 0; i < static_cast<int>(x.size()); ++i) x[i] = x[i] / mInputScale[i] - mInputOffset[i]; }
	virtual void NormalizeOutput(Eigen::VectorXd& y) const { for (int i = 0; i < static_cast<int>(y.size()); ++i) y[i] = (y[i] + mOutputOffset[i]) * mOutputScale[i]; }
	virtual void UnnormalizeOutput(Eigen::VectorXd& y) const { for (int i = 0; i < static_cast<int>(y.size()); ++i) y[i] = y[i] / mOutputScale[i] - mOutputOffset[i]; }

	virtual void CopyModel(const cNeuralNet& other) { mHasNet = other.mHasNet; mValidModel = other.mValidModel; mInputOffset = other.mInputOffset; mInputScale = other.mInputScale; mOutputOffset = other.mOutputOffset; mOutputScale = other.mOutputScale; }
	virtual void ForwardInjectNoisePrefilled(double mean, double stdev, const std::string& layer_name, Eigen::VectorXd& out_y) const {}
	virtual void GetLayerState(const std::string& layer_name, Eigen::VectorXd& out_state) const { out_state.resize(0); }
	virtual void SetLayerState(const Eigen::VectorXd& state, const std::string& layer_name) const {}

protected:
	bool mHasNet, mValidModel;
	Eigen::VectorXd mInputOffset, mInputScale, mOutputOffset, mOutputScale;
	virtual void InitOffsetScale()
	{
		mInputOffset = Eigen::VectorXd::Zero(Harness().mInputSize); mInputScale = Eigen::VectorXd::Ones(Harness().mInputSize);
		mOutputOffset = Eigen::VectorXd::Zero(Harness().mOutputSize); mOutputScale = Eigen::VectorXd::Ones(Harness().mOutputSize);
	}
};
