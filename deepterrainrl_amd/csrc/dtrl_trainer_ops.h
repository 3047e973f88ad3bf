// dtrl_trainer_ops.h -- the batch-32 trainer step of the MACE-family nets as a sequence of small GEMMs.
//
// Reference: cNeuralNet::Eval / Train -> Caffe Forward / SGDSolver::Step on the *_mace3 / *_actor / *_q nets (learning/NeuralNet.cpp:352-375, 1077-1122;
// learning/NeuralNetTrainer.cpp:696-784) as driven by cMACETrainer (learning/MACETrainer.cpp:163-241 BuildProblemY / CalcNewCumulativeRewardBatch,
// :346-372 Step, :577-633 UpdateActorBatchBuffer / StepActor). The Caffe solver rule (SGD: L2 regularise with weight_decay x decay_mult, history =
// momentum x history + base_lr x lr_mult x diff, w -= history; EuclideanLoss = 1 / (2N) sum ||y - label||^2) is written out here.
//
// Every layer's forward, data gradient and weight gradient is ONE small GEMM C[z] = A[z] B[z] over implicit operands (im2col for the 1-D convolutions,
// the transposed weight blobs, a ones column that yields the bias gradient, the ReLU mask in the store): `GemmDesc` names the shape, `load_a / load_b /
// store_c` below define the operands element by element. The same definitions run on the device (dtrl_trainer.hip: an LDS-tiled kernel per GEMM, the
// whole iteration captured as one HIP graph) and, expanded into plain loops by g++, in the CPU-side check build (tests/emul/, TESTS ONLY), so the index
// arithmetic is verified against an independent restatement on a box without a GPU.
#pragma once
#include <cstddef>
#include <cstdint>

#if defined(__HIPCC__)
#define TR_HD __host__ __device__
#else
#define TR_HD
#endif

namespace dtrl_tr {

constexpr int kMaxHeads = 8;

// topology of the net family: slice -> 3 conv1d (+ReLU) -> terr_ip0 (+ReLU) -> concat with the character slice -> trunk (+ReLU) -> n_heads x
// [head0 (+ReLU) -> head1]. MACE: head 0 = val_ip0 / val_ip1 (n_frags values), heads 1.. = a{f}_ip0 / a{f}_ip1. Single-head nets (Q, CACLA actor):
// trunk = ip1, head0 = ip2, head1 = output. Blob order of the flat weight vector = Caffe's layer order, weight then bias per layer.
struct NetDims {
	int S, n_terr, n_char;
	int C[4], Kw[3], T[4];            // conv layer l: C[l] x T[l] -> C[l + 1] x T[l + 1], kernel width Kw[l]
	int n_flat, fc_terr, fc_trunk, fc_head, n_heads;
	int head_out[kMaxHeads], out_off[kMaxHeads], out_size;
	int64_t wo_conv[3], bo_conv[3], wo_terr, bo_terr, wo_ip0, bo_ip0, wo_h0[kMaxHeads], bo_h0[kMaxHeads], wo_h1[kMaxHeads], bo_h1[kMaxHeads];
	int64_t num_params;
	int n_slabs;                      // terr_ip0 forward: split-K slabs of kTerrSlab inputs
};
constexpr int kTerrSlab = 64;

inline void finish_dims(NetDims& d)
{
	d.n_char = d.S - d.n_terr;
	d.C[0] = 1; d.T[0] = d.n_terr;
	for (int l = 0; l < 3; ++l) d.T[l + 1] = d.T[l] - d.Kw[l] + 1;
	d.n_flat = d.C[3] * d.T[3];
	int64_t o = 0;
	for (int l = 0; l < 3; ++l) { d.wo_conv[l] = o; o += static_cast<int64_t>(d.C[l + 1]) * d.C[l] * d.Kw[l]; d.bo_conv[l] = o; o += d.C[l + 1]; }
	d.wo_terr = o; o += static_cast<int64_t>(d.fc_terr) * d.n_flat; d.bo_terr = o; o += d.fc_terr;
	d.wo_ip0 = o; o += static_cast<int64_t>(d.fc_trunk) * (d.fc_terr + d.n_char); d.bo_ip0 = o; o += d.fc_trunk;
	int off = 0;
	for (int f = 0; f < d.n_heads; ++f) {
		d.wo_h0[f] = o; o += static_cast<int64_t>(d.fc_head) * d.fc_trunk; d.bo_h0[f] = o; o += d.fc_head;
		d.wo_h1[f] = o; o += static_cast<int64_t>(d.head_out[f]) * d.fc_head; d.bo_h1[f] = o; o += d.head_out[f];
		d.out_off[f] = off; off += d.head_out[f];
	}
	d.out_size = off; d.num_params = o;
	d.n_slabs = (d.n_flat + kTerrSlab - 1) / kTerrSlab;
}

// device-resident working set of one forward / backward pass over up to max_rows rows
struct Work {
	const float* w;      // weights the pass reads (current or target net)
	float* g;            // gradient, same layout as the weights
	float* xin;          // [rows][S] normalised input
	float* act[3];       // conv outputs, post-ReLU: [rows][C[l + 1]][T[l + 1]]
	float* tp;           // [rows][fc_terr][n_slabs] terr_ip0 partial sums
	float* t3;           // [rows][fc_terr]
	float* h;            // [rows][fc_trunk]
	float* hz;           // [n_heads][rows][fc_head]
	float* out;          // [rows][out_size] (normalised output space)
	float* dout;         // [rows][out_size]
	float* dhz;          // [n_heads][rows][fc_head]
	float* dh;           // [n_heads][rows][fc_trunk] per-head partial gradients wrt the trunk output (summed and ReLU-masked where they are read: dh_at)
	float* dhs;          // [rows][fc_trunk] the heads' contributions summed and masked by the trunk's ReLU (dh_sum_elem)
	float* dt3;          // [rows][fc_terr]
	float* dy[3];        // gradient wrt the conv layers' pre-activations
	float* pw[3];        // per-sample partial weight gradients of the conv layers: [rows][C[l + 1]][C[l] Kw[l] + 1]
	int rows;            // rows of this pass (the M or Z extent)
	int max_rows;        // allocation (row stride of hz / dhz / tp planes)
};

enum Op : int { kConvFwd, kTerrFwd, kIp0Fwd, kHead0Fwd, kHead1Fwd, kHead1Bw, kHead1Bx, kHead0Bw, kHead0Bx, kIp0Bw, kIp0Bx, kTerrBw, kTerrBx, kConvBw, kConvBx };

struct GemmDesc {
	int op, layer;
	int M, N, K, Z;      // C[z] is M x N, reduction length K, Z independent products
	int k0_step;         // split-K: product z covers k in [z k0_step, min(K, (z + 1) k0_step)) when > 0
	int a_kfast, b_kfast;   // which index of the operand is contiguous in memory (tile loads walk it with consecutive lanes)
};

// k = q Kw + r. The shipped nets' kernel widths are powers of two (8, 4, 4): shift and mask instead of an integer division by a run-time value (~30 instructions on the
// device) in front of every operand element of the conv layers' GEMMs (round 6)
TR_HD inline void div_kw(int k, int Kw, int& q, int& r)
{
	if ((Kw & (Kw - 1)) == 0) { const int sh = __builtin_ctz(static_cast<unsigned>(Kw)); q = k >> sh; r = k & (Kw - 1); }
	else { q = k / Kw; r = k - q * Kw; }
}
TR_HD inline float conv_in(const NetDims& d, const Work& wk, int l, int z, int ci, int t)
{
	return l == 0 ? wk.xin[static_cast<size_t>(z) * d.S + t] : wk.act[l - 1][(static_cast<size_t>(z) * d.C[l] + ci) * d.T[l] + t];
}
// gradient wrt the trunk's pre-activation: sum of the heads' contributions, masked by the trunk's ReLU
TR_HD inline float dh_at(const NetDims& d, const Work& wk, int m, int n)
{
	const size_t i = static_cast<size_t>(m) * d.fc_trunk + n;
	if (!(wk.h[i] > 0)) return 0.0f;
	float s = 0;
	for (int f = 0; f < d.n_heads; ++f) s += wk.dh[static_cast<size_t>(f) * wk.max_rows * d.fc_trunk + i];
	return s;
}
TR_HD inline float concat_in(const NetDims& d, const Work& wk, int m, int k)
{
	return k < d.fc_terr ? wk.t3[static_cast<size_t>(m) * d.fc_terr + k] : wk.xin[static_cast<size_t>(m) * d.S + d.n_terr + (k - d.fc_terr)];
}

inline GemmDesc make_gemm(const NetDims& d, int rows, int op, int layer = 0)
{
	GemmDesc g{}; g.op = op; g.layer = layer; g.Z = 1; g.k0_step = 0; g.a_kfast = 1; g.b_kfast = 0;
	const int l = layer;
	switch (op) {
	case kConvFwd: g.Z = rows; g.M = d.C[l + 1]; g.N = d.T[l + 1]; g.K = d.C[l] * d.Kw[l]; break;
	case kTerrFwd: g.Z = d.n_slabs; g.M = rows; g.N = d.fc_terr; g.K = d.n_flat; g.k0_step = kTerrSlab; g.b_kfast = 1; break;
	case kIp0Fwd: g.M = rows; g.N = d.fc_trunk; g.K = d.fc_terr + d.n_char; g.b_kfast = 1; break;
	case kHead0Fwd: g.Z = d.n_heads; g.M = rows; g.N = d.fc_head; g.K = d.fc_trunk; g.b_kfast = 1; break;
	case kHead1Fwd: g.Z = d.n_heads; g.M = rows; g.N = 0; for (int f = 0; f < d.n_heads; ++f) g.N = g.N > d.head_out[f] ? g.N : d.head_out[f]; g.K = d.fc_head; g.b_kfast = 1; break;
	case kHead1Bw: g.Z = d.n_heads; g.M = 0; for (int f = 0; f < d.n_heads; ++f) g.M = g.M > d.head_out[f] ? g.M : d.head_out[f]; g.N = d.fc_head + 1; g.K = rows; g.a_kfast = 0; break;
	case kHead1Bx: g.Z = d.n_heads; g.M = rows; g.N = d.fc_head; g.K = 0; for (int f = 0; f < d.n_heads; ++f) g.K = g.K > d.head_out[f] ? g.K : d.head_out[f]; break;
	case kHead0Bw: g.Z = d.n_heads; g.M = d.fc_head; g.N = d.fc_trunk + 1; g.K = rows; g.a_kfast = 0; break;
	case kHead0Bx: g.Z = d.n_heads; g.M = rows; g.N = d.fc_trunk; g.K = d.fc_head; break;   // one product per head (summed by dh_at): n_heads x the workgroups, a quarter of the K loop
	case kIp0Bw: g.M = d.fc_trunk; g.N = d.fc_terr + d.n_char + 1; g.K = rows; g.a_kfast = 0; break;
	case kIp0Bx: g.M = rows; g.N = d.fc_terr; g.K = d.fc_trunk; break;
	case kTerrBw: g.M = d.fc_terr; g.N = d.n_flat + 1; g.K = rows; g.a_kfast = 0; break;
	case kTerrBx: g.M = rows; g.N = d.n_flat; g.K = d.fc_terr; break;
	case kConvBw: g.Z = rows; g.M = d.C[l + 1]; g.N = d.C[l] * d.Kw[l] + 1; g.K = d.T[l + 1]; break;
	case kConvBx: g.Z = rows; g.M = d.C[l]; g.N = d.T[l]; g.K = d.C[l + 1] * d.Kw[l]; break;
	}
	return g;
}

// A[z](m, k); callers guarantee m < M, k < K (rows of a shorter head read as 0)
TR_HD inline float load_a(const NetDims& d, const Work& wk, const GemmDesc& g, int z, int m, int k)
{
	const int l = g.layer;
	switch (g.op) {
	case kConvFwd: return wk.w[d.wo_conv[l] + static_cast<int64_t>(m) * g.K + k];
	case kTerrFwd: return wk.act[2][static_cast<size_t>(m) * d.n_flat + k];
	case kIp0Fwd: return concat_in(d, wk, m, k);
	case kHead0Fwd: return wk.h[static_cast<size_t>(m) * d.fc_trunk + k];
	case kHead1Fwd: return wk.hz[(static_cast<size_t>(z) * wk.max_rows + m) * d.fc_head + k];
	case kHead1Bw: return m < d.head_out[z] ? wk.dout[static_cast<size_t>(k) * d.out_size + d.out_off[z] + m] : 0.0f;
	case kHead1Bx: return k < d.head_out[z] ? wk.dout[static_cast<size_t>(m) * d.out_size + d.out_off[z] + k] : 0.0f;
	case kHead0Bw: return wk.dhz[(static_cast<size_t>(z) * wk.max_rows + k) * d.fc_head + m];
	case kHead0Bx: return wk.dhz[(static_cast<size_t>(z) * wk.max_rows + m) * d.fc_head + k];
	case kIp0Bw: return wk.dhs[static_cast<size_t>(k) * d.fc_trunk + m];   // (round 6 tried dh_at() here to drop the 5 us dh_sum launch: five loads per operand element made the pair of products 40 us slower)
	case kIp0Bx: return wk.dhs[static_cast<size_t>(m) * d.fc_trunk + k];
	case kTerrBw: return wk.dt3[static_cast<size_t>(k) * d.fc_terr + m];
	case kTerrBx: return wk.dt3[static_cast<size_t>(m) * d.fc_terr + k];
	case kConvBw: return wk.dy[l][(static_cast<size_t>(z) * d.C[l + 1] + m) * d.T[l + 1] + k];
	case kConvBx: { int co, u; div_kw(k, d.Kw[l], co, u); return wk.w[d.wo_conv[l] + (static_cast<int64_t>(co) * d.C[l] + m) * d.Kw[l] + u]; }
	}
	return 0.0f;
}
// B[z](k, n)
TR_HD inline float load_b(const NetDims& d, const Work& wk, const GemmDesc& g, int z, int k, int n)
{
	const int l = g.layer;
	switch (g.op) {
	case kConvFwd: { int ci, u; div_kw(k, d.Kw[l], ci, u); return conv_in(d, wk, l, z, ci, n + u); }
	case kTerrFwd: return wk.w[d.wo_terr + static_cast<int64_t>(n) * d.n_flat + k];
	case kIp0Fwd: return wk.w[d.wo_ip0 + static_cast<int64_t>(n) * g.K + k];
	case kHead0Fwd: return wk.w[d.wo_h0[z] + static_cast<int64_t>(n) * d.fc_trunk + k];
	case kHead1Fwd: return n < d.head_out[z] ? wk.w[d.wo_h1[z] + static_cast<int64_t>(n) * d.fc_head + k] : 0.0f;
	case kHead1Bw: return n < d.fc_head ? wk.hz[(static_cast<size_t>(z) * wk.max_rows + k) * d.fc_head + n] : 1.0f;
	case kHead1Bx: return k < d.head_out[z] ? wk.w[d.wo_h1[z] + static_cast<int64_t>(k) * d.fc_head + n] : 0.0f;
	case kHead0Bw: return n < d.fc_trunk ? wk.h[static_cast<size_t>(k) * d.fc_trunk + n] : 1.0f;
	case kHead0Bx: return wk.w[d.wo_h0[z] + static_cast<int64_t>(k) * d.fc_trunk + n];
	case kIp0Bw: return n < g.N - 1 ? concat_in(d, wk, k, n) : 1.0f;
	case kIp0Bx: return wk.w[d.wo_ip0 + static_cast<int64_t>(k) * (d.fc_terr + d.n_char) + n];
	case kTerrBw: return n < d.n_flat ? wk.act[2][static_cast<size_t>(k) * d.n_flat + n] : 1.0f;
	case kTerrBx: return wk.w[d.wo_terr + static_cast<int64_t>(k) * d.n_flat + n];
	case kConvBw: { if (!(n < g.N - 1)) return 1.0f; int ci, u; div_kw(n, d.Kw[l], ci, u); return conv_in(d, wk, l, z, ci, k + u); }
	case kConvBx: { int co, u; div_kw(k, d.Kw[l], co, u); const int t = n - u; return (t >= 0 && t < d.T[l + 1]) ? wk.dy[l][(static_cast<size_t>(z) * d.C[l + 1] + co) * d.T[l + 1] + t] : 0.0f; }
	}
	return 0.0f;
}
// C[z](m, n) = acc
TR_HD inline void store_c(const NetDims& d, const Work& wk, const GemmDesc& g, int z, int m, int n, float acc)
{
	const int l = g.layer;
	switch (g.op) {
	case kConvFwd: { const float v = acc + wk.w[d.bo_conv[l] + m]; wk.act[l][(static_cast<size_t>(z) * d.C[l + 1] + m) * d.T[l + 1] + n] = v > 0 ? v : 0.0f; break; }
	case kTerrFwd: wk.tp[(static_cast<size_t>(m) * d.fc_terr + n) * d.n_slabs + z] = acc; break;   // slab index fastest: the reduction reads one contiguous run per output
	case kIp0Fwd: { const float v = acc + wk.w[d.bo_ip0 + n]; wk.h[static_cast<size_t>(m) * d.fc_trunk + n] = v > 0 ? v : 0.0f; break; }
	case kHead0Fwd: { const float v = acc + wk.w[d.bo_h0[z] + n]; wk.hz[(static_cast<size_t>(z) * wk.max_rows + m) * d.fc_head + n] = v > 0 ? v : 0.0f; break; }
	case kHead1Fwd: if (n < d.head_out[z]) wk.out[static_cast<size_t>(m) * d.out_size + d.out_off[z] + n] = acc + wk.w[d.bo_h1[z] + n]; break;
	case kHead1Bw: if (m < d.head_out[z]) { if (n < d.fc_head) wk.g[d.wo_h1[z] + static_cast<int64_t>(m) * d.fc_head + n] = acc; else wk.g[d.bo_h1[z] + m] = acc; } break;
	case kHead1Bx: wk.dhz[(static_cast<size_t>(z) * wk.max_rows + m) * d.fc_head + n] = wk.hz[(static_cast<size_t>(z) * wk.max_rows + m) * d.fc_head + n] > 0 ? acc : 0.0f; break;
	case kHead0Bw: if (n < d.fc_trunk) wk.g[d.wo_h0[z] + static_cast<int64_t>(m) * d.fc_trunk + n] = acc; else wk.g[d.bo_h0[z] + m] = acc; break;
	case kHead0Bx: wk.dh[(static_cast<size_t>(z) * wk.max_rows + m) * d.fc_trunk + n] = acc; break;
	case kIp0Bw: if (n < g.N - 1) wk.g[d.wo_ip0 + static_cast<int64_t>(m) * (g.N - 1) + n] = acc; else wk.g[d.bo_ip0 + m] = acc; break;
	case kIp0Bx: wk.dt3[static_cast<size_t>(m) * d.fc_terr + n] = wk.t3[static_cast<size_t>(m) * d.fc_terr + n] > 0 ? acc : 0.0f; break;
	case kTerrBw: if (n < d.n_flat) wk.g[d.wo_terr + static_cast<int64_t>(m) * d.n_flat + n] = acc; else wk.g[d.bo_terr + m] = acc; break;
	case kTerrBx: wk.dy[2][static_cast<size_t>(m) * d.n_flat + n] = wk.act[2][static_cast<size_t>(m) * d.n_flat + n] > 0 ? acc : 0.0f; break;
	case kConvBw: wk.pw[l][(static_cast<size_t>(z) * g.M + m) * g.N + n] = acc; break;
	case kConvBx: wk.dy[l - 1][(static_cast<size_t>(z) * d.C[l] + m) * d.T[l] + n] = conv_in(d, wk, l, z, m, n) > 0 ? acc : 0.0f; break;
	}
}

// ---- element-wise pieces (one call = the whole array; the device versions are grid-stride kernels over the same index space) ----
// t3 = relu(sum of the split-K partials + bias)
TR_HD inline void terr_reduce_elem(const NetDims& d, const Work& wk, int i)
{
	const int m = i / d.fc_terr, n = i % d.fc_terr;
	float s = 0;
	const float* p = wk.tp + (static_cast<size_t>(m) * d.fc_terr + n) * d.n_slabs;
	for (int z = 0; z < d.n_slabs; ++z) s += p[z];
	s += wk.w[d.bo_terr + n];
	wk.t3[i] = s > 0 ? s : 0.0f;
}
TR_HD inline void dh_sum_elem(const NetDims& d, const Work& wk, int i) { wk.dhs[i] = dh_at(d, wk, i / d.fc_trunk, i % d.fc_trunk); }
// conv weight / bias gradients: sum of the per-sample partials
TR_HD inline void conv_grad_elem(const NetDims& d, const Work& wk, int l, int i)
{
	const int N = d.C[l] * d.Kw[l] + 1, M = d.C[l + 1];
	const int m = i / N, n = i % N;
	float s = 0;
	for (int z = 0; z < wk.rows; ++z) s += wk.pw[l][(static_cast<size_t>(z) * M + m) * N + n];
	if (n < N - 1) wk.g[d.wo_conv[l] + static_cast<int64_t>(m) * (N - 1) + n] = s; else wk.g[d.bo_conv[l] + m] = s;
}
// Caffe SGDSolver: Regularize (L2) -> ComputeUpdateValue -> Net::Update
TR_HD inline void sgd_elem(float* w, float* hist, const float* g, const float* rate_mult, const float* decay_mult, float rate, float momentum, float weight_decay, int64_t i)
{
	const float diff = g[i] + weight_decay * decay_mult[i] * w[i];
	const float hv = momentum * hist[i] + rate * rate_mult[i] * diff;
	hist[i] = hv; w[i] = w[i] - hv;
}

// ---- the trainer's own arithmetic around the net (cMACETrainer), per row ----
struct Norm { const float* in_off; const float* in_scale; const float* out_off; const float* out_scale; };

// xin[row][j] = (mem[idx[row]][col0 + j] + in_off[j]) * in_scale[j]   (cNeuralNet::NormalizeInput)
TR_HD inline void gather_norm_elem(int S, const Norm& nm, const float* mem, int W, const int64_t* idx, int col0, float* xin, int64_t i)
{
	const int row = static_cast<int>(i / S), j = static_cast<int>(i % S);
	xin[i] = (mem[static_cast<size_t>(idx[row]) * W + col0 + j] + nm.in_off[j]) * nm.in_scale[j];
}
TR_HD inline float unnorm_out(const Norm& nm, float y, int j) { return y / nm.out_scale[j] - nm.out_off[j]; }

}  // namespace dtrl_tr
