// Stand-in (test infrastructure, see NeuralNetTrainer.h in this directory).
#pragma once
#include "learning/NeuralNetTrainer.h"
class cACTrainer : public cNeuralNetTrainer {};
