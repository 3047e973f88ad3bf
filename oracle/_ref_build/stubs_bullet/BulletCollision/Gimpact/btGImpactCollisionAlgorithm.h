// Stand-in (test infrastructure, see ../../btBulletDynamicsCommon.h): the reference only registers the algorithm with the dispatcher.
#pragma once
#include "btBulletDynamicsCommon.h"
class btGImpactCollisionAlgorithm { public: static void registerAlgorithm(btCollisionDispatcher*) {} };
