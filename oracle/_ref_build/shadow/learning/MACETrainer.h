// Stand-in for the reference's learning/MACETrainer.h -- TEST INFRASTRUCTURE (oracle/_ref_build), never part of the product.
//
// sim/BaseControllerMACE.cpp uses a dozen static index helpers of cMACETrainer (learning/MACETrainer.cpp:9-65); that translation unit needs the
// Caffe-backed trainer classes and cannot be built here, so the helpers are RESTATED below (each is a one-line slice of the parameter vector
// [num_frags critic values | num_frags x frag_size actor parameters]) together with the tuple flag enum (learning/MACETrainer.h:11-17).
#pragma once
#include "learning/NeuralNet.h"

class cMACETrainer
{
public:
	enum eFlag { eFlagFail, eFlagExpCritic, eFlagExpActor, eFlagMax };

	static int GetMaxFragIdx(const Eigen::VectorXd& params, int num_frags) { int a = 0; for (int i = 1; i < num_frags; ++i) if (params[i] > params[a]) a = i; return a; }
	static double GetMaxFragVal(const Eigen::VectorXd& params, int num_frags) { return params[GetMaxFragIdx(params, num_frags)]; }
	static void GetFrag(const Eigen::VectorXd& params, int num_frags, int frag_size, int a_idx, Eigen::VectorXd& out_params) { out_params = params.segment(num_frags + a_idx * frag_size, frag_size); }
	static void SetFrag(const Eigen::VectorXd& frag, int a_idx, int num_frags, int frag_size, Eigen::VectorXd& out_params) { out_params.segment(num_frags + a_idx * frag_size, frag_size) = frag; }
	static double GetVal(const Eigen::VectorXd& params, int a_idx) { return params[a_idx]; }
	static void SetVal(double val, int a_idx, Eigen::VectorXd& out_params) { out_params[a_idx] = val; }
	static int CalcNumFrags(int param_size, int frag_size) { return param_size / (frag_size + 1); }
	static int GetActionFragIdx(const Eigen::VectorXd& action_params) { return static_cast<int>(action_params[0]); }
	static void SetActionFragIdx(int a_idx, Eigen::VectorXd& out_action_params) { out_action_params[0] = a_idx; }
	static void GetActionFrag(const Eigen::VectorXd& action_params, Eigen::VectorXd& out_frag_params) { out_frag_params = action_params.segment(1, action_params.size() - 1); }
	static void SetActionFrag(const Eigen::VectorXd& frag_params, Eigen::VectorXd& out_action_params) { out_action_params.segment(1, out_action_params.size() - 1) = frag_params; }
};
