// dtrl_trainer_core.h -- host-side sequencing of the native trainer step (which GEMMs and element-wise passes, in which order), shared by the HIP
// backend (dtrl_trainer.hip, the product) and the plain-loop check build (tests/emul/dtrl_trainer_emul.cpp, TESTS ONLY). See dtrl_trainer_ops.h.
//
// What one call does and what it replaces in the reference (pool size 1, synchronous mode):
//   Eval         cNeuralNet::EvalBatch (normalise, forward, un-normalise)                                     learning/NeuralNet.cpp:352-375, 964-1036
//   Step         cNeuralNet::Train on one batch: LoadTrainData (normalise data and labels), Caffe SGD step     learning/NeuralNet.cpp:1077-1122
//   CriticStep   cMACETrainer::BuildProblemX / BuildProblemY (CalcNewCumulativeRewardBatch) + the solver step  learning/MACETrainer.cpp:163-250, 478-515
//   ActorFilter  cMACETrainer::UpdateActorBatchBuffer's test  new_q > Q_target(s)                              learning/MACETrainer.cpp:577-609
//   ActorStep    cMACETrainer::BuildActorProblemY + StepActor                                                  learning/MACETrainer.cpp:285-305, 611-633
#pragma once
#include "dtrl_trainer_ops.h"
#include "dtrl_trainer_fused.h"
#include <cstring>
#include <string>
#include <vector>

namespace dtrl_tr {

// ---- element-wise functors (index space = [0, n)) ----
struct FGatherNorm { int S; Norm nm; const float* mem; int W; const int64_t* idx; int col0; float* xin;
	TR_HD void operator()(int64_t i) const { gather_norm_elem(S, nm, mem, W, idx, col0, xin, i); } };
// up to four windows of rows gathered in ONE launch: window v covers rows [v n, (v + 1) n) of xin and reads column col0[v] of the replay rows idx[v][.]
struct FGatherMulti { int S; Norm nm; const float* mem; int W; int n; const int64_t* idx[4]; int col0[4]; float* xin;
	TR_HD void operator()(int64_t i) const
	{
		const int64_t per = static_cast<int64_t>(n) * S;
		const int v = static_cast<int>(i / per);
		gather_norm_elem(S, nm, mem, W, idx[v], col0[v], xin + v * per, i - v * per);
	} };
struct FNormIn { int S; Norm nm; const float* X; float* xin;     // xin = (X + in_off) * in_scale for caller-supplied rows
	TR_HD void operator()(int64_t i) const { const int j = static_cast<int>(i % S); xin[i] = (X[i] + nm.in_off[j]) * nm.in_scale[j]; } };
// (dims / work descriptors are read through pointers to their device-resident copies: a by-value struct whose arrays are indexed with a run-time layer or
// head number would be spilled to scratch memory by the compiler, and every operand load would then pay a scratch access)
struct FTerrReduce { const NetDims* d; const Work* wk; TR_HD void operator()(int64_t i) const { terr_reduce_elem(*d, *wk, static_cast<int>(i)); } };
struct FDhSum { const NetDims* d; const Work* wk; TR_HD void operator()(int64_t i) const { dh_sum_elem(*d, *wk, static_cast<int>(i)); } };
struct FConvGrad { const NetDims* d; const Work* wk; int n0, n1;   // the three conv layers' weight + bias gradients in one index space
	TR_HD void operator()(int64_t i) const { const int l = i < n0 ? 0 : (i < n1 ? 1 : 2); conv_grad_elem(*d, *wk, l, static_cast<int>(i - (l == 0 ? 0 : (l == 1 ? n0 : n1)))); } };
struct FSgd { float* w; float* hist; const float* g; const float* rate_mult; const float* decay_mult; float rate, momentum, weight_decay;
	TR_HD void operator()(int64_t i) const { sgd_elem(w, hist, g, rate_mult, decay_mult, rate, momentum, weight_decay, i); } };
struct FUnnorm { Norm nm; const float* out; float* Y; int out_size;
	TR_HD void operator()(int64_t i) const { Y[i] = unnorm_out(nm, out[i], static_cast<int>(i % out_size)); } };
// new_q[m] = r (1 - discount), + discount max_f Q_target(s')[f] unless the tuple ended in a fall (CalcNewCumulativeRewardBatch); tout = the target net's
// normalised outputs of the s' rows
struct FNewQ { Norm nm; const float* mem; int W; const int64_t* idx; const int64_t* flags; const float* tout; int out_size, n_frags; float discount; float* newq;
	TR_HD void operator()(int64_t m) const
	{
		const size_t row = static_cast<size_t>(idx[m]);
		const float r = mem[row * W] * (1.0f - discount);
		float q = unnorm_out(nm, tout[m * out_size], 0);
		for (int f = 1; f < n_frags; ++f) { const float v = unnorm_out(nm, tout[m * out_size + f], f); q = v > q ? v : q; }
		newq[m] = (flags[row] & 1) ? r : r + discount * q;
	} };
// label (normalised) and the loss gradient of one output element. mode 0: caller's labels Y (un-normalised); 1: critic (y = the net's own output,
// entry a[m] replaced by new_q); 2: actor (entries of fragment a[m] replaced by the tuple's action parameters); 3-5: the single-head trainers (below)
struct FLabelDout { int out_size; Norm nm; int mode; const float* Yext; const float* mem; int W, S; const int64_t* idx; const float* newq; int n_frags, frag_size; int rows;
	const float* out; float* dout; float* sq;
	TR_HD void operator()(int64_t i) const
	{
		const int m = static_cast<int>(i / out_size), j = static_cast<int>(i % out_size);
		float y;
		if (mode == 0) y = Yext[i];
		else {
			y = unnorm_out(nm, out[i], j);
			const size_t row = static_cast<size_t>(idx[m]);
			if (mode >= 3) {
				// single-head trainers. 3: cQNetTrainer::BuildProblemY (learning/QNetTrainer.cpp:57-83): the net's own outputs with the entry of the action taken
				// (tuple.mAction.maxCoeff: the FIRST maximum of the one-hot block) replaced by new_q; 4: cCaclaTrainer's critic (learning/CaclaTrainer.cpp:234-277):
				// the label is new_v; 5: its actor (BuildTupleActorY, :342-387): the action parameters the tuple carries
				if (mode == 3) {
					const float* act = mem + row * W + 1 + S;
					int a = 0; for (int k = 1; k < out_size; ++k) if (act[k] > act[a]) a = k;
					if (j == a) y = newq[m];
				} else if (mode == 4) y = newq[m];
				else y = mem[row * W + 1 + S + j];
			} else {
			const int a = static_cast<int>(mem[row * W + 1 + S]);
			if (mode == 1) { if (j == a) y = newq[m]; }
			else { const int c0 = n_frags + a * frag_size; if (j >= c0 && j < c0 + frag_size) y = mem[row * W + 2 + S + (j - c0)]; }
			}
		}
		const float label = (y + nm.out_off[j]) * nm.out_scale[j];
		const float e = out[i] - label;
		dout[i] = e / static_cast<float>(rows);
		sq[i] = e * e;
	} };
// better[m] = new_q(s') > max_f Q_target(s)[f]; tout rows [0, n) = s, rows [n, 2n) = s'
struct FActorFilter { Norm nm; const float* tout; int out_size, n_frags, n; const float* newq; int32_t* better; float* td = nullptr;
	TR_HD void operator()(int64_t m) const
	{
		float q = unnorm_out(nm, tout[m * out_size], 0);
		for (int f = 1; f < n_frags; ++f) { const float v = unnorm_out(nm, tout[m * out_size + f], f); q = v > q ? v : q; }
		better[m] = newq[m] > q ? 1 : 0;
		if (td) td[m] = newq[m] - q;       // the TD error cCaclaTrainer keeps beside the slot (learning/CaclaTrainer.cpp:365-378)
	} };

// the fused pass's targets in one launch: thread m < n -> new_q of critic row m; n <= m < 2 n -> candidate m - n: its new_q and the test against Q_target(s)
struct FFusedTargets { Norm nm; const float* mem; int W; const int64_t* idx; const int64_t* cand; const int64_t* flags; const float* tout; int out_size, n_frags, n; float discount; float* newq; int32_t* better;
	TR_HD void operator()(int64_t t) const
	{
		if (t < n) { FNewQ{nm, mem, W, idx, flags, tout, out_size, n_frags, discount, newq}(t); return; }
		const int64_t m = t - n;
		FNewQ{nm, mem, W, cand, flags, tout + static_cast<size_t>(2 * n) * out_size, out_size, n_frags, discount, newq + n}(m);
		FActorFilter{nm, tout + static_cast<size_t>(n) * out_size, out_size, n_frags, n, newq + n, better}(m);
	} };
// SGD step with the conv layers' weight / bias gradients reduced from their per-sample partials on the fly (one launch instead of two)
struct FSgdConv { const NetDims* d; const Work* wk; int64_t conv_end; float* w; float* hist; float* g; const float* rate_mult; const float* decay_mult; float rate, momentum, weight_decay;
	TR_HD void operator()(int64_t i) const
	{
		if (i < conv_end) {
			const NetDims& D = *d;
			int l = 0; while (l < 2 && i >= D.wo_conv[l + 1]) ++l;
			const int Kc = D.C[l] * D.Kw[l], N = Kc + 1, M = D.C[l + 1];
			int m, n;
			if (i >= D.bo_conv[l]) { m = static_cast<int>(i - D.bo_conv[l]); n = Kc; } else { const int64_t o = i - D.wo_conv[l]; m = static_cast<int>(o / Kc); n = static_cast<int>(o % Kc); }
			float s = 0;
			for (int z = 0; z < wk->rows; ++z) s += wk->pw[l][(static_cast<size_t>(z) * M + m) * N + n];
			g[i] = s;
		}
		sgd_elem(w, hist, g, rate_mult, decay_mult, rate, momentum, weight_decay, i);
	} };

// the same reduction WITHOUT the update (data-parallel step: the flat gradient leaves the trainer for an all-reduce): g[i] for the conv range, and the sample
// count the gradient was averaged over in the slot behind the last parameter
struct FConvReduce { const NetDims* d; const Work* wk; int64_t conv_end; float* g; int64_t num_params; float count;
	TR_HD void operator()(int64_t i) const
	{
		if (i == conv_end) { g[num_params] = count; return; }
		const NetDims& D = *d;
		int l = 0; while (l < 2 && i >= D.wo_conv[l + 1]) ++l;
		const int Kc = D.C[l] * D.Kw[l], N = Kc + 1, M = D.C[l + 1];
		int m, n;
		if (i >= D.bo_conv[l]) { m = static_cast<int>(i - D.bo_conv[l]); n = Kc; } else { const int64_t o = i - D.wo_conv[l]; m = static_cast<int>(o / Kc); n = static_cast<int>(o % Kc); }
		float s = 0;
		for (int z = 0; z < wk->rows; ++z) s += wk->pw[l][(static_cast<size_t>(z) * M + m) * N + n];
		g[i] = s;
	} };
// Caffe SGD step on a gradient that was SUMMED over ranks (all-reduce): g[num_params] = total number of samples behind it; every rank's share was the mean over
// its own `batch` rows, so the mean over all samples is g[i] * batch / g[num_params]. No samples anywhere (count 0): no update, the history stays
struct FSgdScaled { float* w; float* hist; const float* g; const float* rate_mult; const float* decay_mult; float rate, momentum, weight_decay; int64_t num_params; float batch; float* count_out;
	TR_HD void operator()(int64_t i) const
	{
		const float count = g[num_params];
		if (i == 0 && count_out) *count_out = count;
		if (!(count > 0)) return;
		const float gi = g[i] * (batch / count);
		const float diff = gi + weight_decay * decay_mult[i] * w[i];
		const float hv = momentum * hist[i] + rate * rate_mult[i] * diff;
		hist[i] = hv; w[i] = w[i] - hv;
	} };
struct FZero { float* p; TR_HD void operator()(int64_t i) const { p[i] = 0.0f; } };

// staged rows -> replay slots: element i = (row i / W, column i % W) of the page-locked staging area goes to slot (head + row) % mem_size; the flag word with column 0
struct FAddStaged { const float* src; const int64_t* src_flags; float* mem; int64_t* flags; int W; int64_t head, mem_size;
	TR_HD void operator()(int64_t i) const
	{
		const int64_t r = i / W; const int c = static_cast<int>(i - r * W);
		const int64_t slot = (head + r) % mem_size;
		mem[slot * W + c] = src[i];
		if (c == 0) flags[slot] = src_flags[r];
	} };

struct TrainerConfig {
	NetDims dims;
	int batch = 32, max_eval = 64;
	int n_frags = 0, frag_size = 0;    // MACE head layout (n_frags == 0: single-head net, CriticStep / Actor* unavailable)
	float base_lr = 0.001f, momentum = 0.9f, weight_decay = 0.0005f, discount = 0.9f;
};

// BE: alloc / free / h2d / d2h / d2d / host_alloc / host_free / gemm(d, wk, g) / template for_each(n, functor) / sync() / begin_graph(key) ... see the backends
template <class BE>
class TrainerCore {
public:
	explicit TrainerCore(const TrainerConfig& c) : cfg(c) {}
	~TrainerCore()
	{
		for (void* p : dev_) be.free_dev(p);
		for (void* p : host_) be.free_host(p);
	}
	bool Init(std::string& err)
	{
		if (!be.init(err)) return false;
		const NetDims& d = cfg.dims;
		be.setup_fused(d);
		const size_t P = static_cast<size_t>(d.num_params);
		w_cur = F(P); w_tgt = F(P); hist = F(P); grad = F(P + 1); grad_own = grad; rate_mult = F(P); decay_mult = F(P);   // (grad[P]: sample count of the data-parallel step)
		in_off = F(d.S); in_scale = F(d.S); out_off = F(d.out_size); out_scale = F(d.out_size);
		MakeWork(train, cfg.batch, true);
		MakeWork(eval, cfg.max_eval, false);
		// device-resident descriptors (see FTerrReduce): the dims, and the work set as the three passes see it (current net + gradient; evaluation with the
		// current / the target net)
		train.w = w_cur; train.g = grad;
		d_dims = static_cast<NetDims*>(Dev(sizeof(NetDims))); d_train = static_cast<Work*>(Dev(sizeof(Work))); d_eval_cur = static_cast<Work*>(Dev(sizeof(Work))); d_eval_tgt = static_cast<Work*>(Dev(sizeof(Work)));
		if (!be.ok()) { err = be.error(); return false; }
		{ Work e = eval; e.w = w_cur; be.h2d(d_eval_cur, &e, sizeof(Work)); e.w = w_tgt; be.h2d(d_eval_tgt, &e, sizeof(Work)); }
		be.h2d(d_dims, &cfg.dims, sizeof(NetDims)); be.h2d(d_train, &train, sizeof(Work));
		newq = F(2 * cfg.max_eval); sq = F(static_cast<size_t>(cfg.batch) * d.out_size);
		// page-locked, device-visible: the host writes indices / reads the mask and the loss without a copy being queued
		idx_host = static_cast<int64_t*>(HostAlloc(sizeof(int64_t) * 2 * cfg.max_eval));
		better_host = static_cast<int32_t*>(HostAlloc(sizeof(int32_t) * cfg.max_eval));
		td_host = static_cast<float*>(HostAlloc(sizeof(float) * cfg.max_eval));
		loss_host = static_cast<float*>(HostAlloc(sizeof(float) * 4));
		if (!be.ok()) { err = be.error(); return false; }
		std::vector<float> ones(d.S > d.out_size ? d.S : d.out_size, 1.0f);
		be.h2d(in_scale, ones.data(), sizeof(float) * d.S); be.h2d(out_scale, ones.data(), sizeof(float) * d.out_size);
		std::vector<float> onesP(P, 1.0f);
		be.h2d(rate_mult, onesP.data(), sizeof(float) * P); be.h2d(decay_mult, onesP.data(), sizeof(float) * P);
		be.sync();
		if (!be.ok()) { err = be.error(); return false; }
		return true;
	}
	Norm norm() const { return Norm{in_off, in_scale, out_off, out_scale}; }

	// ---- passes ----
	// wk: one of the device-resident descriptors (d_train, d_eval_cur, d_eval_tgt)
	void Forward(const Work* wk, int rows)
	{
		const NetDims& d = cfg.dims;
		if (be.fused_forward(d_dims, wk, rows, wk == d_train)) return;     // one launch, one workgroup per sample (dtrl_trainer_fused.h); the check build takes the layer-by-layer form below
		if (be.fused_forward_part(d_dims, wk, rows, wk == d_train, 1)) {   // DTRL_TRAINER_FUSED=3: conv stack per sample -> terr_ip0 as one split-K GEMM over all rows -> FC chain per sample
			be.gemm(d_dims, wk, make_gemm(d, rows, kTerrFwd));
			be.fused_forward_part(d_dims, wk, rows, wk == d_train, 3);
			return;
		}
		for (int l = 0; l < 3; ++l) be.gemm(d_dims, wk, make_gemm(d, rows, kConvFwd, l));
		be.gemm(d_dims, wk, make_gemm(d, rows, kTerrFwd));
		be.terr_reduce(d_dims, wk, rows * d.fc_terr, FTerrReduce{d_dims, wk});
		be.gemm(d_dims, wk, make_gemm(d, rows, kIp0Fwd));
		be.gemm(d_dims, wk, make_gemm(d, rows, kHead0Fwd));
		be.gemm(d_dims, wk, make_gemm(d, rows, kHead1Fwd));
	}
	// gradient of the current net on the rows of `train` (dout filled), then the Caffe SGD update
	void BackwardAndUpdate()
	{
		const NetDims& d = cfg.dims;
		if (be.fused_backward(d_dims, d_train, d, cfg.batch, SgdArgs{w_cur, hist, grad, rate_mult, decay_mult, cfg.base_lr, cfg.momentum, cfg.weight_decay, 1, static_cast<float>(cfg.batch)})) return;
		Backward();
		be.for_each(d.num_params, FSgdConv{d_dims, d_train, d.wo_terr, w_cur, hist, grad, rate_mult, decay_mult, cfg.base_lr, cfg.momentum, cfg.weight_decay});
	}
	// the data-parallel form: the flat gradient (conv partials reduced, sample count behind it) is left in `grad` for the caller's all-reduce; ApplyGrad updates
	void BackwardOnly()
	{
		const NetDims& d = cfg.dims;
		if (be.fused_backward(d_dims, d_train, d, cfg.batch, SgdArgs{w_cur, hist, grad, rate_mult, decay_mult, cfg.base_lr, cfg.momentum, cfg.weight_decay, 0, static_cast<float>(cfg.batch)})) return;
		Backward();
		be.for_each(d.wo_terr + 1, FConvReduce{d_dims, d_train, d.wo_terr, grad, d.num_params, static_cast<float>(cfg.batch)});
	}
	void Backward()
	{
		const NetDims& d = cfg.dims;
		const Work* wk = d_train;
		const int rows = cfg.batch;
		if (be.fused_backward_fc(d_dims, wk, rows)) {
			// DTRL_TRAINER_FUSED=4: dhz / dhs / dt3 of every sample from ONE launch; the FC layers' weight gradients then depend on nothing downstream and run beside the
			// terr_ip0 / conv launches (a second branch of the recorded graph), joined in front of the update
			be.fork();
			be.gemm2(d_dims, wk, make_gemm(d, rows, kHead1Bw), make_gemm(d, rows, kHead0Bw));
			be.gemm(d_dims, wk, make_gemm(d, rows, kIp0Bw));
			be.resume();
			be.gemm2(d_dims, wk, make_gemm(d, rows, kTerrBw), make_gemm(d, rows, kTerrBx));
			for (int l = 2; l >= 0; --l) {
				GemmDesc g = make_gemm(d, rows, kConvBw, l); g.b_kfast = 1;
				if (l > 0) be.gemm2(d_dims, wk, g, make_gemm(d, rows, kConvBx, l)); else be.gemm(d_dims, wk, g);
			}
			be.join();
			return;
		}
		// a layer's weight gradient and data gradient read the same incoming gradient and are independent of each other: one launch per pair
		be.gemm2(d_dims, wk, make_gemm(d, rows, kHead1Bw), make_gemm(d, rows, kHead1Bx));
		be.gemm2(d_dims, wk, make_gemm(d, rows, kHead0Bw), make_gemm(d, rows, kHead0Bx));
		be.for_each(static_cast<int64_t>(rows) * d.fc_trunk, FDhSum{d_dims, wk});
		be.gemm2(d_dims, wk, make_gemm(d, rows, kIp0Bw), make_gemm(d, rows, kIp0Bx));
		be.gemm2(d_dims, wk, make_gemm(d, rows, kTerrBw), make_gemm(d, rows, kTerrBx));
		for (int l = 2; l >= 0; --l) {
			GemmDesc g = make_gemm(d, rows, kConvBw, l); g.b_kfast = 1;
			if (l > 0) be.gemm2(d_dims, wk, g, make_gemm(d, rows, kConvBx, l)); else be.gemm(d_dims, wk, g);
		}
	}

	// ---- API ----
	// Y[n][out] = un-normalised outputs of net `which` (0 current, 1 target) on X[n][S] (device pointers)
	bool Eval(int which, const float* X, int n, float* Y)
	{
		if (n <= 0 || n > cfg.max_eval) return false;
		const NetDims& d = cfg.dims;
		be.for_each(static_cast<int64_t>(n) * d.S, FNormIn{d.S, norm(), X, eval.xin});
		Forward(which ? d_eval_tgt : d_eval_cur, n);
		be.for_each(static_cast<int64_t>(n) * d.out_size, FUnnorm{norm(), eval.out, Y, d.out_size});
		return be.ok();
	}
	// one solver iteration on caller-supplied (X, Y) of exactly `batch` rows; the loss lands in loss_host[0] (valid after Sync)
	bool Step(const float* X, const float* Y)
	{
		const NetDims& d = cfg.dims;
		const int n = cfg.batch;
		be.for_each(static_cast<int64_t>(n) * d.S, FNormIn{d.S, norm(), X, train.xin});
		Forward(d_train, n);
		be.label_loss(n * d.out_size, FLabelDout{d.out_size, norm(), 0, Y, nullptr, 0, d.S, nullptr, nullptr, cfg.n_frags, cfg.frag_size, n, train.out, train.dout, sq}, sq, 0.5f / static_cast<float>(n), loss_host);
		BackwardAndUpdate();
		return be.ok();
	}
	// replay memory binding: rows [mem_size][W] float32 in the MACE layout [r | s | a | s'], flag words int64 (device pointers, fixed for the trainer's life)
	void BindReplay(const float* mem, const int64_t* flags, int W) { mem_ = mem; flags_ = flags; W_ = W; }
	// Staging area for new tuples: page-locked, device-visible rows [kStageRows][W] + flag words. The host writes a frame's rows there ONCE; AddStaged then moves
	// rows [first, first + n) into the replay slots (head + i) % mem_size on the trainer's stream -- no copy is queued, no framework call, and the write is
	// ordered with the steps around it (cExpBuffer / cNeuralNetTrainer::AddTuple, learning/NeuralNetTrainer.cpp:145-165: SetTuple at the head, advance the head).
	static constexpr int kStageRows = 4096;
	bool EnsureStage()
	{
		if (stage_rows || !mem_) return stage_rows != nullptr;
		stage_rows = static_cast<float*>(HostAlloc(sizeof(float) * static_cast<size_t>(kStageRows) * W_));
		stage_flags = static_cast<int64_t*>(HostAlloc(sizeof(int64_t) * kStageRows));
		return stage_rows && stage_flags;
	}
	bool AddStaged(int first, int n, int64_t head, int64_t mem_size)
	{
		if (!EnsureStage() || first < 0 || n < 0 || first + n > kStageRows || mem_size <= 0 || head < 0 || head >= mem_size || n > mem_size) return false;
		be.for_each(static_cast<int64_t>(n) * W_, FAddStaged{stage_rows + static_cast<size_t>(first) * W_, stage_flags + first, const_cast<float*>(mem_), const_cast<int64_t*>(flags_), W_, head, mem_size});
		return be.ok();
	}

	// idx_host[0 .. batch) = the critic minibatch's slots. loss -> loss_host[0]
	bool CriticStep() { if (!mem_ || cfg.n_frags <= 0) return false; be.run_graph(0, [this] { CriticStepBody(); }); return be.ok(); }
	// frozen target only (cMACETrainer::EnableTargetNet()): the critic step and the actor candidates' test in ONE pass -- Q_target is needed on s' of the critic
	// batch and on s, s' of the candidates, none of which depends on the critic update, so the target net runs once over [s'_critic | s_cand | s'_cand].
	// idx_host[batch .. 2 batch) must hold `batch` valid slots (pad a shorter candidate list by repeating a slot; the caller ignores the padded answers).
	bool CriticStepAndFilter()
	{
		if (!mem_ || cfg.n_frags <= 0 || !cfg_target_frozen || cfg.max_eval < 3 * cfg.batch) return false;
		be.run_graph(3, [this] {
			const NetDims& d = cfg.dims;
			const int n = cfg.batch, S = d.S, A = 1 + cfg.frag_size;
			const int64_t* cand = idx_host + n;
			// the target net's pass over the 96 rows and the current net's forward over the batch share nothing but read-only inputs: two branches
			be.fork();
			be.for_each(static_cast<int64_t>(3 * n) * S, FGatherMulti{S, norm(), mem_, W_, n, {idx_host, cand, cand, nullptr}, {1 + S + A, 1, 1 + S + A, 0}, eval.xin});
			Forward(d_eval_tgt, 3 * n);
			be.resume();
			be.for_each(static_cast<int64_t>(n) * S, FGatherNorm{S, norm(), mem_, W_, idx_host, 1, train.xin});
			Forward(d_train, n);
			be.join();                     // the targets need the target net's outputs, the labels the targets: ONE launch for targets, labels and loss (round 6)
			be.pre_label_loss(2 * n, FFusedTargets{norm(), mem_, W_, idx_host, cand, flags_, eval.out, d.out_size, cfg.n_frags, n, cfg.discount, newq, better_host},
				n * d.out_size, FLabelDout{d.out_size, norm(), 1, nullptr, mem_, W_, S, idx_host, newq, cfg.n_frags, cfg.frag_size, n, train.out, train.dout, sq}, sq, 0.5f / static_cast<float>(n), loss_host);
			BackwardAndUpdate();
		});
		return be.ok();
	}
	void CriticStepBody()
	{
		const NetDims& d = cfg.dims;
		const int n = cfg.batch, S = d.S, A = 1 + cfg.frag_size;
		// Q_target(s') first (its activations live in `eval`), then the current net on s: the forward of the solver step IS the evaluation BuildProblemY needs
		be.for_each(static_cast<int64_t>(n) * S, FGatherNorm{S, norm(), mem_, W_, idx_host, 1 + S + A, eval.xin});
		Forward(cfg_target_frozen ? d_eval_tgt : d_eval_cur, n);
		be.for_each(n, FNewQ{norm(), mem_, W_, idx_host, flags_, eval.out, d.out_size, cfg.n_frags, cfg.discount, newq});
		be.for_each(static_cast<int64_t>(n) * S, FGatherNorm{S, norm(), mem_, W_, idx_host, 1, train.xin});
		Forward(d_train, n);
		be.label_loss(n * d.out_size, FLabelDout{d.out_size, norm(), 1, nullptr, mem_, W_, S, idx_host, newq, cfg.n_frags, cfg.frag_size, n, train.out, train.dout, sq}, sq, 0.5f / static_cast<float>(n), loss_host);
		BackwardAndUpdate();
	}
	// idx_host[batch .. batch + n) = candidate slots (a window of their own: the critic step queued before may not have read its indices yet);
	// better_host[0 .. n) = 1 where new_q > Q_target(s) (valid after Sync)
	bool ActorFilter(int n)
	{
		if (!mem_ || cfg.n_frags <= 0 || n <= 0 || 2 * n > cfg.max_eval || n > cfg.batch) return false;
		const NetDims& d = cfg.dims;
		const int S = d.S, A = 1 + cfg.frag_size;
		const int64_t* idx = idx_host + cfg.batch;
		be.for_each(static_cast<int64_t>(n) * S, FGatherNorm{S, norm(), mem_, W_, idx, 1, eval.xin});
		be.for_each(static_cast<int64_t>(n) * S, FGatherNorm{S, norm(), mem_, W_, idx, 1 + S + A, eval.xin + static_cast<size_t>(n) * S});
		Forward(cfg_target_frozen ? d_eval_tgt : d_eval_cur, 2 * n);
		be.for_each(n, FNewQ{norm(), mem_, W_, idx, flags_, eval.out + static_cast<size_t>(n) * d.out_size, d.out_size, cfg.n_frags, cfg.discount, newq});
		be.for_each(n, FActorFilter{norm(), eval.out, d.out_size, cfg.n_frags, n, newq, better_host});
		return be.ok();
	}
	// idx_host[max_eval .. max_eval + batch) = the actor batch's slots (a second window, so that a filter's candidates stay intact). loss -> loss_host[1]
	bool ActorStep() { if (!mem_ || cfg.n_frags <= 0) return false; be.run_graph(2, [this] { ActorStepBody(); }); return be.ok(); }
	void ActorStepBody()
	{
		const NetDims& d = cfg.dims;
		const int n = cfg.batch, S = d.S;
		const int64_t* idx = idx_host + cfg.max_eval;
		be.for_each(static_cast<int64_t>(n) * S, FGatherNorm{S, norm(), mem_, W_, idx, 1, train.xin});
		Forward(d_train, n);
		be.label_loss(n * d.out_size, FLabelDout{d.out_size, norm(), 2, nullptr, mem_, W_, S, idx, nullptr, cfg.n_frags, cfg.frag_size, n, train.out, train.dout, sq}, sq, 0.5f / static_cast<float>(n), loss_host + 1);
		BackwardAndUpdate();
	}
	// ---- single-head trainers on replay rows [r | s | a (A entries) | s'] (n_frags == 0): the whole iteration in one recorded launch sequence ----
	int ActionWidth() const { return W_ - 1 - 2 * cfg.dims.S; }
	// kind 0 = cQNetTrainer::Step's solver iteration (learning/QNetTrainer.cpp:27-83, 142-163): new_q = r (1 - g) [+ g max_a' Q(s')[a'] unless the tuple failed] with the
	//          net itself as reference (pool of one), label = own outputs with the taken action's entry replaced;
	// kind 1 = cCaclaTrainer's critic iteration (learning/CaclaTrainer.cpp:149-157, 234-277): new_v = r (1 - g) [+ g V_target(s')], label = new_v.
	// idx_host[0 .. batch) = the minibatch's slots; loss -> loss_host[0]
	bool ValueStep(int kind)
	{
		if (!mem_ || cfg.n_frags != 0 || kind < 0 || kind > 1 || ActionWidth() < 1 || (kind == 0 && ActionWidth() != cfg.dims.out_size) || (kind == 1 && cfg.dims.out_size != 1)) return false;
		be.run_graph(4 + kind, [this, kind] {
			const NetDims& d = cfg.dims;
			const int n = cfg.batch, S = d.S, A = ActionWidth();
			be.for_each(static_cast<int64_t>(n) * S, FGatherNorm{S, norm(), mem_, W_, idx_host, 1 + S + A, eval.xin});
			Forward((kind == 1 && cfg_target_frozen) ? d_eval_tgt : d_eval_cur, n);
			be.for_each(n, FNewQ{norm(), mem_, W_, idx_host, flags_, eval.out, d.out_size, d.out_size, cfg.discount, newq});   // (the maximum over ALL outputs = the entry argmax picks)
			be.for_each(static_cast<int64_t>(n) * S, FGatherNorm{S, norm(), mem_, W_, idx_host, 1, train.xin});
			Forward(d_train, n);
			be.label_loss(n * d.out_size, FLabelDout{d.out_size, norm(), 3 + kind, nullptr, mem_, W_, S, idx_host, newq, 0, 0, n, train.out, train.dout, sq}, sq, 0.5f / static_cast<float>(n), loss_host);
			BackwardAndUpdate();
		});
		return be.ok();
	}
	// cCaclaTrainer::UpdateActorBatchBuffer's test (learning/CaclaTrainer.cpp:342-387) on the CRITIC's trainer: candidates idx_host[batch .. batch + n);
	// td_host[m] = new_v(s') - V_target(s), better_host[m] = td > 0 (valid after Sync)
	bool TdFilter(int n)
	{
		if (!mem_ || cfg.n_frags != 0 || cfg.dims.out_size != 1 || n <= 0 || 2 * n > cfg.max_eval || n > cfg.batch) return false;
		const NetDims& d = cfg.dims;
		const int S = d.S, A = ActionWidth();
		const int64_t* idx = idx_host + cfg.batch;
		const Work* tgt = cfg_target_frozen ? d_eval_tgt : d_eval_cur;
		be.for_each(static_cast<int64_t>(2 * n) * S, FGatherMulti{S, norm(), mem_, W_, n, {idx, idx, nullptr, nullptr}, {1, 1 + S + A, 0, 0}, eval.xin});
		Forward(tgt, 2 * n);
		be.for_each(n, FNewQ{norm(), mem_, W_, idx, flags_, eval.out + static_cast<size_t>(n), 1, 1, cfg.discount, newq});
		be.for_each(n, FActorFilter{norm(), eval.out, 1, 1, n, newq, better_host, td_host});
		return be.ok();
	}
	// one solver iteration towards the action parameters the tuples carry (cCaclaTrainer's actor, BuildTupleActorY): slots idx_host[max_eval .. max_eval + batch)
	// of the replay memory this trainer is bound to (the critic's); loss -> loss_host[0]
	bool ActionStep()
	{
		if (!mem_ || cfg.n_frags != 0 || ActionWidth() != cfg.dims.out_size) return false;
		be.run_graph(6, [this] {
			const NetDims& d = cfg.dims;
			const int n = cfg.batch, S = d.S;
			const int64_t* idx = idx_host + cfg.max_eval;
			be.for_each(static_cast<int64_t>(n) * S, FGatherNorm{S, norm(), mem_, W_, idx, 1, train.xin});
			Forward(d_train, n);
			be.label_loss(n * d.out_size, FLabelDout{d.out_size, norm(), 5, nullptr, mem_, W_, S, idx, nullptr, 0, 0, n, train.out, train.dout, sq}, sq, 0.5f / static_cast<float>(n), loss_host);
			BackwardAndUpdate();
		});
		return be.ok();
	}
	void UpdateTarget() { be.d2d(w_tgt, w_cur, sizeof(float) * cfg.dims.num_params); }

	// ---- data-parallel step (SURVEY 5 last row; learning/ParamServer.cpp:65-90 is what it stands in for): gradient here, all-reduce at the caller, update here ----
	// The gradient buffer [num_params + 1] may be the caller's (device memory it can hand to a collective): BindGrad; nullptr = the trainer's own again
	void BindGrad(float* g) { grad = g ? g : grad_own; train.g = grad; be.h2d(d_train, &train, sizeof(Work)); be.drop_graphs(); }   // (the recorded update launches carry the old pointer)
	bool GradStep(const float* X, const float* Y)
	{
		const NetDims& d = cfg.dims;
		const int n = cfg.batch;
		be.for_each(static_cast<int64_t>(n) * d.S, FNormIn{d.S, norm(), X, train.xin});
		Forward(d_train, n);
		be.label_loss(n * d.out_size, FLabelDout{d.out_size, norm(), 0, Y, nullptr, 0, d.S, nullptr, nullptr, cfg.n_frags, cfg.frag_size, n, train.out, train.dout, sq}, sq, 0.5f / static_cast<float>(n), loss_host);
		BackwardOnly();
		return be.ok();
	}
	// cMACETrainer's critic batch idx_host[0 .. batch): everything of CriticStep but the update
	bool CriticGrad()
	{
		if (!mem_ || cfg.n_frags <= 0) return false;
		be.run_graph(7, [this] {      // (recorded like the fused steps; BindGrad drops the recordings, they carry the gradient buffer's address)
			const NetDims& d = cfg.dims;
			const int n = cfg.batch, S = d.S, A = 1 + cfg.frag_size;
			be.for_each(static_cast<int64_t>(n) * S, FGatherNorm{S, norm(), mem_, W_, idx_host, 1 + S + A, eval.xin});
			Forward(cfg_target_frozen ? d_eval_tgt : d_eval_cur, n);
			be.for_each(n, FNewQ{norm(), mem_, W_, idx_host, flags_, eval.out, d.out_size, cfg.n_frags, cfg.discount, newq});
			be.for_each(static_cast<int64_t>(n) * S, FGatherNorm{S, norm(), mem_, W_, idx_host, 1, train.xin});
			Forward(d_train, n);
			be.label_loss(n * d.out_size, FLabelDout{d.out_size, norm(), 1, nullptr, mem_, W_, S, idx_host, newq, cfg.n_frags, cfg.frag_size, n, train.out, train.dout, sq}, sq, 0.5f / static_cast<float>(n), loss_host);
			BackwardOnly();
		});
		return be.ok();
	}
	// the actor batch idx_host[max_eval .. max_eval + batch): everything of ActorStep but the update
	bool ActorGrad()
	{
		if (!mem_ || cfg.n_frags <= 0) return false;
		be.run_graph(8, [this] {
			const NetDims& d = cfg.dims;
			const int n = cfg.batch, S = d.S;
			const int64_t* idx = idx_host + cfg.max_eval;
			be.for_each(static_cast<int64_t>(n) * S, FGatherNorm{S, norm(), mem_, W_, idx, 1, train.xin});
			Forward(d_train, n);
			be.label_loss(n * d.out_size, FLabelDout{d.out_size, norm(), 2, nullptr, mem_, W_, S, idx, nullptr, cfg.n_frags, cfg.frag_size, n, train.out, train.dout, sq}, sq, 0.5f / static_cast<float>(n), loss_host + 1);
			BackwardOnly();
		});
		return be.ok();
	}
	// a rank without a batch this round contributes nothing: gradient and count zero
	bool ZeroGrad() { be.for_each(cfg.dims.num_params + 1, FZero{grad}); return be.ok(); }
	// the update from the (all-reduced) gradient; the total sample count lands in loss_host[slot] (2: critic round, 3: actor round; valid after Sync)
	bool ApplyGrad(int slot)
	{
		if (slot < 2 || slot > 3) return false;
		const NetDims& d = cfg.dims;
		be.for_each(d.num_params, FSgdScaled{w_cur, hist, grad, rate_mult, decay_mult, cfg.base_lr, cfg.momentum, cfg.weight_decay, d.num_params, static_cast<float>(cfg.batch), loss_host + slot});
		return be.ok();
	}

	TrainerConfig cfg;
	bool cfg_target_frozen = false;   // cMACETrainer::EnableTargetNet(): freeze_target_iters > 0, else the "target" is the current net
	BE be;
	float *w_cur = nullptr, *w_tgt = nullptr, *hist = nullptr, *grad = nullptr, *grad_own = nullptr, *rate_mult = nullptr, *decay_mult = nullptr;
	float *in_off = nullptr, *in_scale = nullptr, *out_off = nullptr, *out_scale = nullptr, *newq = nullptr, *sq = nullptr;
	Work train{}, eval{};                 // host copies (pointers into device memory)
	NetDims* d_dims = nullptr; Work* d_train = nullptr; Work* d_eval_cur = nullptr; Work* d_eval_tgt = nullptr;   // device-resident descriptors
	int64_t* idx_host = nullptr; int32_t* better_host = nullptr; float* loss_host = nullptr; float* td_host = nullptr;
	float* stage_rows = nullptr; int64_t* stage_flags = nullptr;

private:
	void* Dev(size_t bytes) { void* p = be.alloc_dev(bytes); if (p) dev_.push_back(p); return p; }
	float* F(size_t n) { void* p = be.alloc_dev(sizeof(float) * (n ? n : 1)); if (p) dev_.push_back(p); return static_cast<float*>(p); }
	void* HostAlloc(size_t bytes) { void* p = be.alloc_host(bytes); if (p) { std::memset(p, 0, bytes); host_.push_back(p); } return p; }
	void MakeWork(Work& wk, int rows, bool with_grad)
	{
		const NetDims& d = cfg.dims;
		std::memset(&wk, 0, sizeof(wk));
		wk.max_rows = rows; wk.rows = rows;
		wk.xin = F(static_cast<size_t>(rows) * d.S);
		for (int l = 0; l < 3; ++l) wk.act[l] = F(static_cast<size_t>(rows) * d.C[l + 1] * d.T[l + 1]);
		wk.tp = F(static_cast<size_t>(d.n_slabs) * rows * d.fc_terr);
		wk.t3 = F(static_cast<size_t>(rows) * d.fc_terr); wk.h = F(static_cast<size_t>(rows) * d.fc_trunk);
		wk.hz = F(static_cast<size_t>(d.n_heads) * rows * d.fc_head); wk.out = F(static_cast<size_t>(rows) * d.out_size);
		if (!with_grad) return;
		wk.dout = F(static_cast<size_t>(rows) * d.out_size); wk.dhz = F(static_cast<size_t>(d.n_heads) * rows * d.fc_head);
		wk.dh = F(static_cast<size_t>(d.n_heads) * rows * d.fc_trunk); wk.dhs = F(static_cast<size_t>(rows) * d.fc_trunk); wk.dt3 = F(static_cast<size_t>(rows) * d.fc_terr);
		for (int l = 0; l < 3; ++l) { wk.dy[l] = F(static_cast<size_t>(rows) * d.C[l + 1] * d.T[l + 1]); wk.pw[l] = F(static_cast<size_t>(rows) * d.C[l + 1] * (d.C[l] * d.Kw[l] + 1)); }
	}
	const float* mem_ = nullptr; const int64_t* flags_ = nullptr; int W_ = 0;
	std::vector<void*> dev_, host_;
};

}  // namespace dtrl_tr
