// Stand-in (test infrastructure, see ../../btBulletDynamicsCommon.h).
#pragma once
#include "btDantzigSolver.h"
class btMLCPSolver : public btSequentialImpulseConstraintSolver { public: explicit btMLCPSolver(btMLCPSolverInterface*) {} };
