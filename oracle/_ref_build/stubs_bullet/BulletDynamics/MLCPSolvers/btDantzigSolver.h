// Stand-in (test infrastructure, see ../../btBulletDynamicsCommon.h): included by sim/World.cpp, not used by the code paths built here.
#pragma once
#include "btBulletDynamicsCommon.h"
class btMLCPSolverInterface { public: virtual ~btMLCPSolverInterface() {} };
class btDantzigSolver : public btMLCPSolverInterface {};
