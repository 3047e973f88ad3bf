// drive_shim_eval.cpp -- TEST driver for include/BatchScenarioPoliEval.h: builds the shim inside the reference's header tree (unchanged
// scenarios/Scenario.h, util/ArgParser.h, util/Rand.h; stand-in Eigen of oracle/_ref_build) and runs what cOptScenarioPoliEval does with its pool
// (optimizer/scenarios/OptScenarioPoliEval.cpp): BuildScenePool (ParseArgs, Init, SetRandSeed + Reset), the EvalHelper loop (Update until
// max_episodes / max_cycles, folding GetAvgDist / GetNumEpisodes / GetNumCycles into the record and calling ResetAvgDist), OutputResults.
//   drive_shim_eval <data_root> <arg_file> <pool_size> <max_episodes> <max_cycles> <policy.bin> <rand_seed> <out_file> [extra "-key= value" tokens...]
// policy.bin = [int64 n][float32 w[n]][float64 in_off[S]][in_scale[S]][out_off[O]][out_scale[O]] written by the test.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "BatchScenarioPoliEval.h"
#include "util/MathUtil.h"

int main(int argc, char** argv)
{
	if (argc < 9) { fprintf(stderr, "usage\n"); return 2; }
	std::vector<char*> args;
	std::string a0 = "-data_root=", a1 = argv[1];
	args.push_back(&a0[0]); args.push_back(&a1[0]);
	for (int i = 9; i < argc; ++i) args.push_back(argv[i]);
	cArgParser parser(args.data(), static_cast<int>(args.size()));   // command line first, then the file (optimizer/Main.cpp:19-32)
	parser.AppendArgs(std::string(argv[1]) + "/" + argv[2]);
	const int pool = std::atoi(argv[3]), max_episodes = std::atoi(argv[4]), max_cycles = std::atoi(argv[5]);
	const unsigned long seed = std::strtoul(argv[7], nullptr, 10);

	cBatchScenarioPoliEval scene(pool);
	scene.ParseArgs(parser);
	scene.Init();
	if (!scene.IsValid()) return 3;
	int S = 0, O = 0;
	scene.GetDims(S, O);
	FILE* f = std::fopen(argv[6], "rb");
	if (!f) return 5;
	int64_t n = 0;
	if (std::fread(&n, sizeof(n), 1, f) != 1 || static_cast<size_t>(n) != scene.GetNumPolicyParams()) return 6;
	std::vector<float> w(n); std::vector<double> io(S), is(S), oo(O), os(O);
	if (std::fread(w.data(), 4, n, f) != static_cast<size_t>(n) || std::fread(io.data(), 8, S, f) != static_cast<size_t>(S) || std::fread(is.data(), 8, S, f) != static_cast<size_t>(S)
		|| std::fread(oo.data(), 8, O, f) != static_cast<size_t>(O) || std::fread(os.data(), 8, O, f) != static_cast<size_t>(O)) return 7;
	std::fclose(f);
	if (!scene.SetPolicy(w.data(), w.size(), io.data(), is.data(), oo.data(), os.data())) return 8;
	if (seed != 0) {   // BuildScenePool: valid_seed -> SetRandSeed + Reset ("rebuild ground")
		scene.SetRandSeed(seed);
		printf("seeds=");
		for (uint64_t s : scene.GetSceneSeeds()) printf("%llu ", static_cast<unsigned long long>(s));
		printf("\n");
		scene.Reset();
	}
	printf("name=%s pool=%d S=%d O=%d\n", scene.GetName().c_str(), scene.GetPoolSize(), S, O);

	// cOptScenarioPoliEval::EvalHelper over the pool as one object (the per-scene thresholds scale with the pool size), UpdateRecord's arithmetic
	const int num_episodes_per_update = 10 * pool;
	int num_episodes = 0, num_cycles = 0, prev_cycles = 0, rec_episodes = 0, rec_cycles = 0, frames = 0;
	double rec_avg = 0;
	while (num_episodes < max_episodes && num_cycles < max_cycles) {
		scene.Update(1.0 / 30.0); ++frames;
		num_cycles = scene.GetNumCycles();
		const int curr = scene.GetNumEpisodes();
		if (curr >= num_episodes_per_update || curr + num_episodes >= max_episodes) {
			const double avg = scene.GetAvgDist();
			rec_avg = cMathUtil::AddAverage(rec_avg, rec_episodes, avg, curr);
			rec_episodes += curr; rec_cycles += num_cycles - prev_cycles;
			printf("fold frame=%d episodes=%d cycles=%d avg_dist=%.9f\n", frames, rec_episodes, rec_cycles, rec_avg);
			scene.ResetAvgDist();
			num_episodes += curr; prev_cycles = num_cycles;
		}
	}
	const std::vector<double>& log = scene.GetDistLog();
	double sum = 0;
	for (double d : log) sum += d;
	printf("frames=%d dist_log=%zu dist_sum=%.9f\n", frames, log.size(), sum);
	if (!scene.OutputResults(argv[8])) return 9;
	scene.Shutdown();
	return 0;
}
