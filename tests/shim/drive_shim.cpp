// drive_shim.cpp -- TEST driver for include/BatchScenarioExp.h: builds the shim inside the reference's header tree (unchanged
// scenarios/Scenario.h, learning/ExpTuple.h, util/ArgParser.h; stand-in Eigen of oracle/_ref_build) and drives outer frames through it
// the way cScenarioTrain::UpdateExpScene does (scenarios/ScenarioTrain.cpp:376-410). Prints one summary line per drained tuple buffer.
//   drive_shim <data_root> <arg_file> <num_envs> <frames> <policy.bin> [extra "-key= value" tokens...]
// policy.bin = [int64 n][float32 w[n]][float64 in_off[S]][in_scale[S]][out_off[O]][out_scale[O]] written by the test.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "BatchScenarioExp.h"

int main(int argc, char** argv)
{
	if (argc < 6) { fprintf(stderr, "usage\n"); return 2; }
	std::vector<char*> args;
	std::string a0 = "-data_root=", a1 = argv[1];
	args.push_back(&a0[0]); args.push_back(&a1[0]);
	for (int i = 6; i < argc; ++i) args.push_back(argv[i]);
	cArgParser parser(args.data(), static_cast<int>(args.size()));   // command line first, then the file (optimizer/Main.cpp:19-32)
	parser.AppendArgs(std::string(argv[1]) + "/" + argv[2]);
	const int n_envs = std::atoi(argv[3]), frames = std::atoi(argv[4]);

	cBatchScenarioExp scene(n_envs);
	scene.ParseArgs(parser);
	scene.Init();
	if (!scene.IsValid()) return 3;
	const int S = scene.GetPoliStateSize(), A = scene.GetPoliActionSize();
	Eigen::VectorXd off, sc;
	if (!scene.BuildNNOutputOffsetScale(off, sc)) return 4;
	const int O = static_cast<int>(off.size());
	FILE* f = std::fopen(argv[5], "rb");
	if (!f) return 5;
	int64_t n = 0;
	if (std::fread(&n, sizeof(n), 1, f) != 1 || static_cast<size_t>(n) != scene.GetNumPolicyParams()) return 6;
	std::vector<float> w(n); std::vector<double> io(S), is(S), oo(O), os(O);
	if (std::fread(w.data(), 4, n, f) != static_cast<size_t>(n) || std::fread(io.data(), 8, S, f) != static_cast<size_t>(S) || std::fread(is.data(), 8, S, f) != static_cast<size_t>(S)
		|| std::fread(oo.data(), 8, O, f) != static_cast<size_t>(O) || std::fread(os.data(), 8, O, f) != static_cast<size_t>(O)) return 7;
	std::fclose(f);
	if (!scene.SetPolicy(w.data(), w.size(), io.data(), is.data(), oo.data(), os.data())) return 8;
	scene.SetExpRate(0.2); scene.SetExpTemp(0.025); scene.SetExpBaseActionRate(0.002);   // args/opt_args_train_mace.txt:25-27 initial rates
	scene.Reset();                                                                         // BuildScenePool: "rebuild ground"
	printf("name=%s envs=%d S=%d A=%d O=%d buffer=%d\n", scene.GetName().c_str(), scene.GetNumEnvs(), S, A, O, 32);
	long total = 0;
	for (int fr = 0; fr < frames; ++fr) {
		scene.Update(1.0 / 30.0);
		if (scene.IsTupleBufferFull()) {
			const std::vector<tExpTuple>& tuples = scene.GetTuples();
			double sr = 0, ss = 0, sa = 0; unsigned fl = 0;
			for (const tExpTuple& t : tuples) {
				sr += t.mReward; fl ^= t.mFlags * 2654435761u + static_cast<unsigned>(t.mID);
				for (int k = 0; k < S; ++k) ss += t.mStateBeg[k] - 0.5 * t.mStateEnd[k];
				for (int k = 0; k < A; ++k) sa += t.mAction[k];
			}
			total += static_cast<long>(tuples.size());
			printf("frame=%d tuples=%zu reward_sum=%.9g state_sum=%.9g action_sum=%.9g flags_hash=%u\n", fr, tuples.size(), sr, ss, sa, fl);
			scene.ResetTupleBuffer();
		}
	}
	printf("total=%ld\n", total);
	scene.Shutdown();
	return 0;
}
