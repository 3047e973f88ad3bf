// Stand-in for <caffe/caffe.hpp> -- TEST INFRASTRUCTURE (oracle/_ref_build), never part of the product.
//
// Caffe (fork niuzhiheng/caffe@7b3e6f2 + caffe_mods/, README.md:15) is an un-vendored external of the reference and is absent from this image. The reference's
// trainers (learning/NeuralNetTrainer.cpp, MACETrainer.cpp, QNetTrainer.cpp, ACTrainer.cpp, CaclaTrainer.cpp, NeuralNetLearner.cpp, ...) never touch Caffe
// themselves: they speak to learning/NeuralNet.h's cNeuralNet. That header names a handful of Caffe types in its declarations; this file declares exactly those,
// empty, so that the reference's OWN learning/NeuralNet.h (and with it every trainer translation unit) compiles unchanged. The member functions of cNeuralNet live in
// learning/NeuralNet.cpp, which IS Caffe code and is not compiled; oracle/_ref_build/ref_learn_net.cpp defines them over a network the test harness supplies.
#pragma once
#include <memory>
#include <string>
#include <vector>

namespace boost {
template <class T> using shared_ptr = std::shared_ptr<T>;
}

namespace caffe {
enum Phase { TRAIN = 0, TEST = 1 };
template <typename T> class Blob {};
class NetParameter {};
template <typename T> class MemoryDataLayer {};
// what learning/NeuralNet.h's cCaffeNetWrapper derives from: here a handle into the harness' table of networks
template <typename T> class Net {
public:
	Net(const std::string& net_file, Phase phase) : mFile(net_file), mPhase(phase) {}
	virtual ~Net() {}
	std::string mFile;
	Phase mPhase;
	int mHarnessId = -1;
	std::shared_ptr<void> mImpl;      // (libref_learn_native.so: the cBatchNeuralNet the wrapper stands for)
};
}  // namespace caffe
