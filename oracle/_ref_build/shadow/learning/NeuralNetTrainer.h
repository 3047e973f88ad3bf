// Stand-in for the reference's learning/NeuralNetTrainer.h -- TEST INFRASTRUCTURE (oracle/_ref_build). The trainers are Caffe-backed and are not
// built; the rollout-side sources compiled into libref_sim.so include the trainer headers only for their tuple FLAG enums.
#pragma once
#include "learning/ExpTuple.h"
#include "learning/NeuralNet.h"
class cNeuralNetTrainer { public: virtual ~cNeuralNetTrainer() {} };
