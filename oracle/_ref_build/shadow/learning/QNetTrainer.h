// Stand-in (test infrastructure, see NeuralNetTrainer.h in this directory): learning/QNetTrainer.h:8-12 flag enum only.
#pragma once
#include "learning/NeuralNetTrainer.h"
class cQNetTrainer : public cNeuralNetTrainer { public: enum eFlag { eFlagFail, eFlagMax }; };
