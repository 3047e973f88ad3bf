// Stand-in (test infrastructure, see NeuralNetTrainer.h in this directory): learning/CaclaTrainer.h:8-21 enums only.
#pragma once
#include "learning/ACTrainer.h"
class cCaclaTrainer : public cACTrainer { public: enum eFlag { eFlagFail, eFlagOffPolicy, eFlagMax }; enum eMode { eModeCacla, eModeTD, eModePTD, eModeMax }; };
