// TESTS ONLY: include/BatchNeuralNet.h compiled inside the reference's header tree (util/MathUtil.h from $(REF), stand-in Eigen) and driven through the calls
// cNeuralNetTrainer / cMACETrainer make on a cNeuralNet. Usage: drive_shim_net <data_root>   (data_root holds data/policies/dog/nets/dog_mace3_*.prototxt)
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <string>

#include "BatchNeuralNet.h"

#define REQUIRE(c) do { if (!(c)) { printf("FAILED: %s (line %d)\n", #c, __LINE__); return 1; } } while (0)

int main(int argc, char** argv)
{
	const std::string root = argc > 1 ? argv[1] : ".";
	const std::string deploy = root + "/data/policies/dog/nets/dog_mace3_deploy.prototxt", solver = root + "/data/policies/dog/nets/dog_mace3_solver.prototxt";
	cBatchNeuralNet net(-1, root);
	net.SetTrainerParams(0.9, false);
	net.LoadNet(deploy);
	REQUIRE(net.HasNet() && !net.HasSolver() && net.GetInputSize() == 283 && net.GetOutputSize() == 90 && net.GetBatchSize() == 0 && net.CalcNumParams() == 570474);
	std::vector<float> w0; net.GetParamsFlat(w0);
	net.LoadSolver(solver);
	REQUIRE(net.HasSolver() && net.GetBatchSize() == 32);
	std::vector<float> w1; net.GetParamsFlat(w1);
	REQUIRE(w0 == w1);                                          // LoadSolver keeps the deploy net's parameters (SyncSolverParams)
	std::mt19937 rng(5); std::normal_distribution<double> nd(0, 1); std::uniform_real_distribution<double> ud(0.5, 2.0);
	const int S = 283, O = 90, B = 32;
	Eigen::VectorXd io(S), is(S), oo(O), os(O);
	for (int i = 0; i < S; ++i) { io[i] = 0.3 * nd(rng); is[i] = ud(rng); }
	for (int i = 0; i < O; ++i) { oo[i] = 0.2 * nd(rng); os[i] = ud(rng); }
	net.SetInputOffsetScale(io, is); net.SetOutputOffsetScale(oo, os);
	Eigen::MatrixXd X(B, S), Y;
	for (int i = 0; i < B; ++i) for (int j = 0; j < S; ++j) X(i, j) = nd(rng);
	net.EvalBatch(X, Y);
	REQUIRE(Y.rows() == B && Y.cols() == O);
	Eigen::VectorXd x = X.row(7), y;
	net.Eval(x, y);
	double dmax = 0, ymax = 0;
	for (int j = 0; j < O; ++j) { dmax = std::max(dmax, std::fabs(y[j] - Y(7, j))); ymax = std::max(ymax, std::fabs(y[j])); }
	REQUIRE(dmax <= 1e-6 * std::max(1.0, ymax) && ymax > 0);   // one row evaluated alone = the same row inside a batch
	// Train: repeated passes over one batch drive its loss down (EuclideanLoss + Caffe SGD with momentum)
	cBatchNeuralNet::tProblem prob;
	prob.mX = X; prob.mY.resize(B, O); prob.mPassesPerStep = 1;
	for (int i = 0; i < B; ++i) for (int j = 0; j < O; ++j) prob.mY(i, j) = Y(i, j) + 0.5 * nd(rng);
	net.Train(prob);
	const double loss0 = net.GetLastLoss();
	for (int k = 0; k < 40; ++k) net.Train(prob);
	const double loss1 = net.GetLastLoss();
	REQUIRE(std::isfinite(loss0) && loss0 > 0 && loss1 < 0.7 * loss0 && net.HasValidModel());
	// CopyModel: parameters and normalisers
	cBatchNeuralNet other(-1, root);
	other.LoadNet(deploy); other.LoadSolver(solver);
	Eigen::VectorXd y0, y1, y2;
	other.Eval(x, y0);
	other.CopyModel(net);
	other.Eval(x, y1); net.Eval(x, y2);
	double diff01 = 0, diff12 = 0;
	for (int j = 0; j < O; ++j) { diff01 = std::max(diff01, std::fabs(y0[j] - y1[j])); diff12 = std::max(diff12, std::fabs(y1[j] - y2[j])); }
	REQUIRE(diff12 == 0 && diff01 > 1e-6 && other.GetInputOffset()[3] == io[3] && other.GetOutputScale()[5] == os[5]);
	// CalcOffsetScale = -mean, 1 / population standard deviation
	Eigen::VectorXd off, sc;
	net.CalcOffsetScale(X, off, sc);
	double m = 0, v = 0;
	for (int i = 0; i < B; ++i) m += X(i, 11) / B;
	for (int i = 0; i < B; ++i) v += (X(i, 11) - m) * (X(i, 11) - m) / B;
	REQUIRE(std::fabs(off[11] + m) < 1e-12 && std::fabs(sc[11] - 1.0 / std::sqrt(v)) < 1e-9);
	// OutputModel / LoadModel round trip
	const std::string tmp = (argc > 2 ? std::string(argv[2]) : std::string("/tmp")) + "/shim_net_model.bin";
	net.OutputModel(tmp);
	cBatchNeuralNet third(-1, root);
	third.LoadNet(deploy); third.LoadModel(tmp);
	std::vector<float> wa, wb; net.GetParamsFlat(wa); third.GetParamsFlat(wb);
	REQUIRE(wa == wb && third.HasValidModel());
	printf("shim net ok: params %d, batch %d, loss %.5f -> %.5f over 41 steps, CopyModel exact\n", net.CalcNumParams(), net.GetBatchSize(), loss0, loss1);
	return 0;
}
