// ref_learn_harness.h -- TEST INFRASTRUCTURE (oracle/_ref_build): the callbacks through which libref_learn.so's cNeuralNet (ref_learn_net.cpp) reaches the network
// the test harness supplies (oracle/reflearn.py). All arrays are row-major doubles in NORMALISED coordinates (the normalisation is cNeuralNet's, restated in C++).
#pragma once
extern "C" {
struct RefLearnHarness {
	int (*net_new)(const char* net_file);                                  // cNeuralNet::LoadNet: a net of that prototxt's topology; returns its id
	void (*net_free)(int id);
	void (*net_dims)(int id, int* in_size, int* out_size);
	int (*solver_load)(int id, const char* solver_file);                   // cNeuralNet::LoadSolver: returns the train net's batch size
	void (*solver_reset)(int id);                                          // cNeuralNet::ResetSolver: solver history cleared
	void (*net_forward)(int id, const double* x, int n, double* y);        // y[n, out] = net(x[n, in])
	void (*net_step)(int id, const double* x, const double* y, int n, int iters);   // LoadTrainData + StepSolver(iters): EuclideanLoss + the solver's update
	void (*net_copy)(int dst, int src);                                    // cNeuralNet::CopyParams
};
void ref_learn_set_harness(const RefLearnHarness* h);
}
