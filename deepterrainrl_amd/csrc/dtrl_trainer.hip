// dtrl_trainer.hip -- HIP backend of the native MACE trainer step (include/dtrl_trainer.h; sequencing in dtrl_trainer_core.h, operands in dtrl_trainer_ops.h).
//
// Every layer pass is one launch of ONE LDS-tiled GEMM kernel over implicit operands: a 32 x 64 output tile per 256-thread workgroup (4 wavefronts),
// 2 x 4 outputs per thread, K staged through LDS in chunks of 16; the operand tiles are gathered element-wise by load_a / load_b (im2col, transposes,
// ones column), with consecutive lanes walking whichever index is contiguous in memory. The products are tiny (a batch-32 step is 0.33 GFLOP in ~25
// products) and every one of them is sized for the batch, so the step is bound by launch latency and per-launch parallelism, not by FLOPs: the split
// over (tile, sample / head / K-slab) gives each launch 16 - 190 workgroups, and weights (2.3 MB), history, activations and the replay rows never leave
// the device. Accumulation runs in k order per output, so results are deterministic and equal the plain-loop check build up to fp32 contraction.
#include "dtrl_trainer_core.h"
#include <hip/hip_runtime.h>
#include "dtrl_trainer_fused.h"
#include <algorithm>
#include <cstdlib>
#include <string>

namespace dtrl_tr {

constexpr int kTM = 32, kTN = 64, kTK = 64, kThreads = 256;   // (a K chunk costs a global-memory round trip whatever its size: few, long chunks)

// OP is a compile-time constant so that the operand switch of load_a / load_b / store_c folds away in each instantiation; dims and work descriptors are
// read through pointers (scalar loads from the device-resident copies; a by-value struct with run-time indexed arrays would live in scratch memory)
template <int OP>
__device__ __forceinline__ void tr_gemm_tile(const NetDims& d, const Work& wk, GemmDesc g, int z, float (&As)[kTK][kTM + 1], float (&Bs)[kTK][kTN + 1])
{
	g.op = OP;
	const int tid = static_cast<int>(threadIdx.x);
	const int m0 = static_cast<int>(blockIdx.y) * kTM, n0 = static_cast<int>(blockIdx.x) * kTN;
	if (m0 >= g.M || n0 >= g.N) return;   // (a fused launch is sized for the larger of its two products)
	const int k_begin = g.k0_step ? z * g.k0_step : 0;
	const int k_end = g.k0_step ? (k_begin + g.k0_step < g.K ? k_begin + g.k0_step : g.K) : g.K;
	constexpr int kRA = kTM * kTK / kThreads, kRB = kTK * kTN / kThreads;
	float ra[kRA], rb[kRB];
	// the operand elements of one K chunk, gathered into registers (the next chunk's loads are in flight while the current one is multiplied)
	auto fetch = [&](int k0) {
#pragma unroll
		for (int i = 0; i < kRA; ++i) {
			const int e = tid + i * kThreads;
			int mm, kk;
			if (g.a_kfast) { kk = e % kTK; mm = e / kTK; } else { mm = e % kTM; kk = e / kTM; }
			const int m = m0 + mm, k = k0 + kk;
			ra[i] = (m < g.M && k < k_end) ? load_a(d, wk, g, z, m, k) : 0.0f;
		}
#pragma unroll
		for (int i = 0; i < kRB; ++i) {
			const int e = tid + i * kThreads;
			int nn, kk;
			if (g.b_kfast) { kk = e % kTK; nn = e / kTK; } else { nn = e % kTN; kk = e / kTN; }
			const int n = n0 + nn, k = k0 + kk;
			rb[i] = (n < g.N && k < k_end) ? load_b(d, wk, g, z, k, n) : 0.0f;
		}
	};
	// tile product on the fp32 matrix pipe: wavefront w owns the 32 x 32 sub-tile of columns 32 (w & 1) .. and the half (w >> 1) of every K chunk; one
	// v_mfma_f32_32x32x2f32 per pair of k values (operands: A[i = lane % 32][k = lane / 32], B[k = lane / 32][j = lane % 32], one float each), the two K halves
	// are added through LDS at the end. Per chunk and wave: 16 matrix instructions and 32 LDS reads (the VALU form took 512 FMAs and 384 reads)
	typedef float v16f_t __attribute__((ext_vector_type(16)));
	const int wave = tid >> 6, lane = tid & 63, nsub = wave & 1, khalf = wave >> 1, li = lane & 31, lk = lane >> 5;
	v16f_t acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
	fetch(k_begin);
	for (int k0 = k_begin; k0 < k_end; k0 += kTK) {
#pragma unroll
		for (int i = 0; i < kRA; ++i) { const int e = tid + i * kThreads; if (g.a_kfast) As[e % kTK][e / kTK] = ra[i]; else As[e / kTM][e % kTM] = ra[i]; }
#pragma unroll
		for (int i = 0; i < kRB; ++i) { const int e = tid + i * kThreads; if (g.b_kfast) Bs[e % kTK][e / kTK] = rb[i]; else Bs[e / kTN][e % kTN] = rb[i]; }
		__syncthreads();
		if (k0 + kTK < k_end) fetch(k0 + kTK);
		const int kb = khalf * (kTK / 2);
#pragma unroll
		for (int kk = 0; kk < kTK / 2; kk += 2) {
			const float a = As[kb + kk + lk][li];
			const float b = Bs[kb + kk + lk][nsub * 32 + li];
			acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
		}
		__syncthreads();
	}
	// K halves: waves 2, 3 park their partial tiles in LDS (the operand tiles are dead), waves 0, 1 add and store. Result register r of lane l is
	// C[8 (r / 4) + 4 (l / 32) + r % 4][l % 32]
	float* part = &Bs[0][0];   // 2 x 32 x 32 floats = 8 KB of the 16.6 KB
	if (khalf == 1) {
#pragma unroll
		for (int r = 0; r < 16; ++r) part[(nsub * 16 + r) * 64 + lane] = acc[r];
	}
	__syncthreads();
	if (khalf == 0) {
#pragma unroll
		for (int r = 0; r < 16; ++r) {
			const float v = acc[r] + part[(nsub * 16 + r) * 64 + lane];
			const int m = m0 + 8 * (r >> 2) + 4 * lk + (r & 3), n = n0 + nsub * 32 + li;
			if (m < g.M && n < g.N) store_c(d, wk, g, z, m, n, v);
		}
	}
}
template <int OP>
__global__ void __launch_bounds__(kThreads) tr_gemm_kernel(const NetDims* __restrict__ dp, const Work* __restrict__ wp, GemmDesc g)
{
	__shared__ float As[kTK][kTM + 1];
	__shared__ float Bs[kTK][kTN + 1];
	tr_gemm_tile<OP>(*dp, *wp, g, static_cast<int>(blockIdx.z), As, Bs);
}
// two independent products in one launch: blockIdx.z < ga.Z belongs to the first
template <int OPA, int OPB>
__global__ void __launch_bounds__(kThreads) tr_gemm2_kernel(const NetDims* __restrict__ dp, const Work* __restrict__ wp, GemmDesc ga, GemmDesc gb)
{
	__shared__ float As[kTK][kTM + 1];
	__shared__ float Bs[kTK][kTN + 1];
	const int z = static_cast<int>(blockIdx.z);
	if (z < ga.Z) tr_gemm_tile<OPA>(*dp, *wp, ga, z, As, Bs); else tr_gemm_tile<OPB>(*dp, *wp, gb, z - ga.Z, As, Bs);
}

__global__ void __launch_bounds__(256) tr_sum_kernel(const float* __restrict__ x, int n, float scale, float* __restrict__ out)
{
	__shared__ float part[256];
	const int t = static_cast<int>(threadIdx.x);
	float s = 0;
	for (int i = t; i < n; i += 256) s += x[i];
	part[t] = s;
	__syncthreads();
	for (int d = 128; d > 0; d >>= 1) { if (t < d) part[t] += part[t + d]; __syncthreads(); }
	if (t == 0) *out = scale * part[0];
}

// t3 = relu(sum of the terr_ip0 split-K partials + bias): 16 lanes per output read the output's contiguous run of partials together (coalesced) and
// combine with a butterfly -- a thread per output walks 94 strided addresses alone
__global__ void __launch_bounds__(256) tr_terr_reduce_kernel(const NetDims* __restrict__ dp, const Work* __restrict__ wp, int n_out)
{
	const NetDims& d = *dp; const Work& wk = *wp;
	const int lane16 = static_cast<int>(threadIdx.x) & 15;
	const int o = (static_cast<int>(blockIdx.x) * 256 + static_cast<int>(threadIdx.x)) >> 4;
	float s = 0;
	if (o < n_out) { const float* p = wk.tp + static_cast<size_t>(o) * d.n_slabs; for (int z = lane16; z < d.n_slabs; z += 16) s += p[z]; }
	for (int off = 8; off > 0; off >>= 1) s += __shfl_xor(s, off, 16);
	if (o < n_out && lane16 == 0) { s += wk.w[d.bo_terr + o % d.fc_terr]; wk.t3[o] = s > 0 ? s : 0.0f; }
}

// one workgroup: an optional pre-pass (the critic step's targets: new_q of the batch and the candidates' test), then the labels / loss gradient of every output
// element and the loss = scale x sum of the squared errors -- three dependent launches of 5-7 us each (targets, labels, sum) as one (round 6). The sum runs in a fixed
// order (thread-strided partial sums, then a tree): deterministic.
template <class FP, class FL>
__global__ void __launch_bounds__(1024) tr_pre_label_loss_kernel(int n_pre, FP fp, int n_lab, FL fl, const float* sq, float scale, float* out)   // (no __restrict__: fl writes sq[i] through its own copy of the pointer)
{
	__shared__ float part[1024];
	const int t = static_cast<int>(threadIdx.x);
	for (int i = t; i < n_pre; i += 1024) fp(i);
	if (n_pre > 0) __syncthreads();           // the labels read new_q: block-wide visibility of the pre-pass's global writes
	float s = 0.0f;
	for (int i = t; i < n_lab; i += 1024) { fl(i); s += sq[i]; }
	part[t] = s;
	__syncthreads();
	for (int d = 512; d > 0; d >>= 1) { if (t < d) part[t] += part[t + d]; __syncthreads(); }
	if (t == 0) *out = scale * part[0];
}

template <class F>
__global__ void tr_foreach_kernel(int64_t n, F f)
{
	for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += static_cast<int64_t>(gridDim.x) * blockDim.x) f(i);
}

struct HipTrainerBE {
	int device_id = -1;
	hipStream_t stream = nullptr;
	std::string err_;
	bool init(std::string& err)
	{
		int count = 0;
		hipError_t e = hipGetDeviceCount(&count);
		if (e != hipSuccess || count <= 0) { err = std::string("no usable HIP device (") + hipGetErrorString(e) + "); the trainer has no CPU fallback"; return false; }
		if (device_id >= 0 && hipSetDevice(device_id) != hipSuccess) { err = "hipSetDevice failed"; return false; }
		return true;
	}
	bool ok() const { return err_.empty(); }
	const std::string& error() const { return err_; }
	bool chk(hipError_t e, const char* what) { if (e == hipSuccess) return true; if (err_.empty()) err_ = std::string(what) + ": " + hipGetErrorString(e); return false; }
	// fork(): what is queued next runs on a second stream, ordered behind everything queued so far; resume(): back on the trainer's stream WITHOUT waiting
	// (what is queued next runs beside the second stream's work); join(): the trainer's stream waits for the second one. Lets two independent passes of a step (the target net on 96 rows, the current net on the batch) share the GPU instead of queueing their
	// ~8 small launches each one after the other. Inside a graph capture the two become parallel branches of the graph. No-ops without a stream of the trainer's own.
	void fork()
	{
		if (!stream || forked_) return;
		if (!side_ && (hipStreamCreateWithFlags(&side_, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&ev_fork_, hipEventDisableTiming) != hipSuccess
				|| hipEventCreateWithFlags(&ev_join_, hipEventDisableTiming) != hipSuccess)) { side_ = nullptr; (void)hipGetLastError(); return; }
		if (!chk(hipEventRecord(ev_fork_, stream), "fork record") || !chk(hipStreamWaitEvent(side_, ev_fork_, 0), "fork wait")) return;
		main_ = stream; stream = side_; forked_ = true;
	}
	void resume() { if (forked_) stream = main_; }
	void join()
	{
		if (!forked_) return;
		chk(hipEventRecord(ev_join_, side_), "join record");
		stream = main_; forked_ = false;
		chk(hipStreamWaitEvent(stream, ev_join_, 0), "join wait");
	}
	hipStream_t side_ = nullptr, main_ = nullptr; hipEvent_t ev_fork_ = nullptr, ev_join_ = nullptr; bool forked_ = false;
	void drop_graphs() { for (hipGraphExec_t& e : exec_) if (e) { hipGraphExecDestroy(e); e = nullptr; } }   // recorded launches carry pointers by value: re-record
	void set_stream(void* s)
	{
		if (static_cast<hipStream_t>(s) == stream) return;
		drop_graphs();   // recorded on the old stream's behalf
		stream = static_cast<hipStream_t>(s);
	}
	void* alloc_dev(size_t bytes) { void* p = nullptr; if (!chk(hipMalloc(&p, bytes), "hipMalloc")) return nullptr; chk(hipMemset(p, 0, bytes), "hipMemset"); return p; }
	void free_dev(void* p) { hipFree(p); }
	void* alloc_host(size_t bytes) { void* p = nullptr; return chk(hipHostMalloc(&p, bytes, hipHostMallocMapped | hipHostMallocCoherent), "hipHostMalloc") ? p : nullptr; }
	void free_host(void* p) { hipHostFree(p); }
	void h2d(void* dst, const void* src, size_t n) { chk(hipMemcpyAsync(dst, src, n, hipMemcpyHostToDevice, stream), "hipMemcpy H2D"); chk(hipStreamSynchronize(stream), "sync"); }
	void d2h(void* dst, const void* src, size_t n) { chk(hipMemcpyAsync(dst, src, n, hipMemcpyDeviceToHost, stream), "hipMemcpy D2H"); chk(hipStreamSynchronize(stream), "sync"); }
	void d2d(void* dst, const void* src, size_t n) { chk(hipMemcpyAsync(dst, src, n, hipMemcpyDeviceToDevice, stream), "hipMemcpy D2D"); }
	void sync() { chk(hipStreamSynchronize(stream), "hipStreamSynchronize"); }
	template <int OP> void launch(const NetDims* d, const Work* wk, const GemmDesc& g)
	{
		const dim3 grid((g.N + kTN - 1) / kTN, (g.M + kTM - 1) / kTM, g.Z);
		hipLaunchKernelGGL(tr_gemm_kernel<OP>, grid, dim3(kThreads), 0, stream, d, wk, g);
	}
	void gemm(const NetDims* d, const Work* wk, const GemmDesc& g)
	{
		switch (g.op) {
		case kConvFwd: launch<kConvFwd>(d, wk, g); break; case kTerrFwd: launch<kTerrFwd>(d, wk, g); break; case kIp0Fwd: launch<kIp0Fwd>(d, wk, g); break;
		case kHead0Fwd: launch<kHead0Fwd>(d, wk, g); break; case kHead1Fwd: launch<kHead1Fwd>(d, wk, g); break; case kHead1Bw: launch<kHead1Bw>(d, wk, g); break;
		case kHead1Bx: launch<kHead1Bx>(d, wk, g); break; case kHead0Bw: launch<kHead0Bw>(d, wk, g); break; case kHead0Bx: launch<kHead0Bx>(d, wk, g); break;
		case kIp0Bw: launch<kIp0Bw>(d, wk, g); break; case kIp0Bx: launch<kIp0Bx>(d, wk, g); break; case kTerrBw: launch<kTerrBw>(d, wk, g); break;
		case kTerrBx: launch<kTerrBx>(d, wk, g); break; case kConvBw: launch<kConvBw>(d, wk, g); break; case kConvBx: launch<kConvBx>(d, wk, g); break;
		}
		chk(hipGetLastError(), "gemm launch");
	}
	template <int OPA, int OPB> void launch2(const NetDims* d, const Work* wk, const GemmDesc& ga, const GemmDesc& gb)
	{
		const dim3 grid((std::max(ga.N, gb.N) + kTN - 1) / kTN, (std::max(ga.M, gb.M) + kTM - 1) / kTM, ga.Z + gb.Z);
		hipLaunchKernelGGL((tr_gemm2_kernel<OPA, OPB>), grid, dim3(kThreads), 0, stream, d, wk, ga, gb);
	}
	void gemm2(const NetDims* d, const Work* wk, const GemmDesc& ga, const GemmDesc& gb)
	{
		if (ga.op == kHead1Bw && gb.op == kHead1Bx) launch2<kHead1Bw, kHead1Bx>(d, wk, ga, gb);
		else if (ga.op == kHead1Bw && gb.op == kHead0Bw) launch2<kHead1Bw, kHead0Bw>(d, wk, ga, gb);
		else if (ga.op == kHead0Bw && gb.op == kHead0Bx) launch2<kHead0Bw, kHead0Bx>(d, wk, ga, gb);
		else if (ga.op == kIp0Bw && gb.op == kIp0Bx) launch2<kIp0Bw, kIp0Bx>(d, wk, ga, gb);
		else if (ga.op == kTerrBw && gb.op == kTerrBx) launch2<kTerrBw, kTerrBx>(d, wk, ga, gb);
		else if (ga.op == kConvBw && gb.op == kConvBx) launch2<kConvBw, kConvBx>(d, wk, ga, gb);
		else { gemm(d, wk, ga); gemm(d, wk, gb); return; }
		chk(hipGetLastError(), "gemm2 launch");
	}
	// a fixed launch sequence (pointers and shapes never change: the callers' inputs arrive through fixed page-locked / device buffers) recorded once as a
	// HIP graph and replayed: ~35 launches of a few microseconds each are bound by the gaps between them, not by their work. Needs a stream of the
	// trainer's own (the legacy default stream cannot be captured); DTRL_TRAINER_GRAPHS=0 keeps plain launches.
	template <class Fn> void run_graph(int key, Fn fn)
	{
		static const bool enabled = []() { const char* e = std::getenv("DTRL_TRAINER_GRAPHS"); return !e || std::atoi(e) != 0; }();
		if (!enabled || !stream || key < 0 || key >= kMaxGraphs || graph_failed_) { fn(); return; }
		if (!exec_[key]) {
			hipGraph_t graph = nullptr;
			if (hipStreamBeginCapture(stream, hipStreamCaptureModeThreadLocal) != hipSuccess) { graph_failed_ = true; (void)hipGetLastError(); fn(); return; }
			fn();
			const hipError_t e1 = hipStreamEndCapture(stream, &graph);
			if (e1 != hipSuccess || !graph || hipGraphInstantiate(&exec_[key], graph, nullptr, nullptr, 0) != hipSuccess) {
				graph_failed_ = true; exec_[key] = nullptr; (void)hipGetLastError(); err_.clear();
				if (graph) hipGraphDestroy(graph);
				fn();   // nothing was executed during the failed capture
				return;
			}
			hipGraphDestroy(graph);
		}
		chk(hipGraphLaunch(exec_[key], stream), "hipGraphLaunch");
	}
	static constexpr int kMaxGraphs = 12;
	hipGraphExec_t exec_[kMaxGraphs] = {};
	bool graph_failed_ = false;
	~HipTrainerBE() { for (hipGraphExec_t e : exec_) if (e) hipGraphExecDestroy(e); if (side_) { hipStreamDestroy(side_); hipEventDestroy(ev_fork_); hipEventDestroy(ev_join_); } }
	template <class F> void terr_reduce(const NetDims* d, const Work* wk, int n_out, const F&)
	{
		hipLaunchKernelGGL(tr_terr_reduce_kernel, dim3((n_out * 16 + 255) / 256), dim3(256), 0, stream, d, wk, n_out);
		chk(hipGetLastError(), "terr reduce launch");
	}
	// labels + loss gradient (functor f writes dout and sq), then *out = scale * sum(sq)
	template <class F> void label_loss(int n, const F& f, const float* sq, float scale, float* out)
	{
		// two launches: measured against a single-workgroup fused form (2035-2050 vs 2063-2071 Train()/s): the 12-block label pass + one reduction wins
		// (the loss reduction on the second stream beside the backward pass: measured slower, 2190-2280 vs 2330/s -- the fork / join events cost more than its 6 us)
		for_each(n, f); loss_sum(sq, n, scale, out);
	}
	template <class FP, class FL> void pre_label_loss(int n_pre, const FP& fp, int n_lab, const FL& fl, const float* sq, float scale, float* out)
	{
		hipLaunchKernelGGL((tr_pre_label_loss_kernel<FP, FL>), dim3(1), dim3(1024), 0, stream, n_pre, fp, n_lab, fl, sq, scale, out);
		chk(hipGetLastError(), "pre/label/loss launch");
	}
	// *out = scale * sum(x[0 .. n)): one workgroup, tree reduction in LDS
	void loss_sum(const float* x, int n, float scale, float* out)
	{
		hipLaunchKernelGGL(tr_sum_kernel, dim3(1), dim3(256), 0, stream, x, n, scale, out);
		chk(hipGetLastError(), "loss launch");
	}
	// ---- per-sample fused passes (dtrl_trainer_fused.h): 1 + 1 + 1 launches instead of 8 + 8 + 1. DTRL_TRAINER_FUSED=0 keeps the layer-by-layer form (A/B) ----
	FusedPlan plan_;
	bool fused_bwd_ = false;
	// DTRL_TRAINER_FUSED: 0 = layer-by-layer everywhere, 1 = per-sample fused FORWARD passes (default: measured ahead), 2 = fused forward and backward (measured behind: a
	// sample's backward chain is bound by ONE compute unit's vector issue rate, 32 of 256 units busy -- profiles/r04_trainer.txt)
	void setup_fused(const NetDims& d)
	{
		plan_ = fused_plan(d);
		const char* e = std::getenv("DTRL_TRAINER_FUSED");
		const int mode = e ? std::atoi(e) : 1;
		if (mode == 0) plan_.ok = false;
		fused_bwd_ = mode == 2;
		split_fwd_ = mode == 3;
		fc_bwd_ = mode == 4;
	}
	bool fc_bwd_ = false;
	// DTRL_TRAINER_FUSED=4 (round 6): fused forward + the FC part of the data-gradient chain per sample in one launch (tr_fused_backward_x_kernel, fc_only)
	bool fused_backward_fc(const NetDims* d, const Work* wk, int rows)
	{
		if (!plan_.ok || !fc_bwd_ || rows <= 0) return false;
		static const int stage = []() { const char* e = std::getenv("DTRL_TRAINER_DBG"); const int v = e ? std::atoi(e) : 0; return (v == 2 || v == 3) ? v : 1; }();   // (2 / 3: timing experiments, results garbage)
		hipLaunchKernelGGL(tr_fused_backward_x_kernel, dim3(rows), dim3(kFT), plan_.lds_bytes, stream, d, wk, plan_.size_a, plan_.size_b | (stage << 24));
		chk(hipGetLastError(), "fused FC backward launch");
		return true;
	}
	bool split_fwd_ = false;
	// DTRL_TRAINER_FUSED=3 (round 6): the forward pass as conv stack (one workgroup per sample) -> terr_ip0 as ONE split-K GEMM over all rows -> FC chain (per sample)
	bool fused_forward_part(const NetDims* d, const Work* wk, int rows, bool store, int part)
	{
		if (!plan_.ok || !split_fwd_ || rows <= 0) return false;
		if (part == 1) {
			static const int dbg = []() { const char* e = std::getenv("DTRL_TRAINER_DBG"); return e ? std::atoi(e) : 0; }();   // timing experiments only (results are garbage)
			const int sb = plan_.size_b | (dbg << 24);
			if (store) hipLaunchKernelGGL((tr_fused_forward_kernel<true, 1>), dim3(rows), dim3(kFT), plan_.lds_bytes, stream, d, wk, plan_.size_a, sb);
			else hipLaunchKernelGGL((tr_fused_forward_kernel<false, 1>), dim3(rows), dim3(kFT), plan_.lds_bytes, stream, d, wk, plan_.size_a, sb);
		} else {
			if (store) hipLaunchKernelGGL((tr_fused_forward_kernel<true, 3>), dim3(rows), dim3(kFT), plan_.lds_bytes, stream, d, wk, plan_.size_a, plan_.size_b);
			else hipLaunchKernelGGL((tr_fused_forward_kernel<false, 3>), dim3(rows), dim3(kFT), plan_.lds_bytes, stream, d, wk, plan_.size_a, plan_.size_b);
		}
		chk(hipGetLastError(), "fused forward part launch");
		return true;
	}
	bool fused_forward(const NetDims* d, const Work* wk, int rows, bool store)
	{
		if (!plan_.ok || split_fwd_ || rows <= 0) return false;
		if (store) hipLaunchKernelGGL(tr_fused_forward_kernel<true>, dim3(rows), dim3(kFT), plan_.lds_bytes, stream, d, wk, plan_.size_a, plan_.size_b);
		else hipLaunchKernelGGL(tr_fused_forward_kernel<false>, dim3(rows), dim3(kFT), plan_.lds_bytes, stream, d, wk, plan_.size_a, plan_.size_b);
		chk(hipGetLastError(), "fused forward launch");
		return true;
	}
	bool fused_backward(const NetDims* d, const Work* wk, const NetDims& hd, int rows, const SgdArgs& a)
	{
		if (!plan_.ok || !fused_bwd_ || rows <= 0) return false;
		hipLaunchKernelGGL(tr_fused_backward_x_kernel, dim3(rows), dim3(kFT), plan_.lds_bytes, stream, d, wk, plan_.size_a, plan_.size_b);
		const int conv_blocks = static_cast<int>((hd.wo_terr + 3) / 4);
		const int fc_blocks = static_cast<int>((hd.num_params - hd.wo_terr + 255) / 256);
		hipLaunchKernelGGL(tr_fused_grad_kernel, dim3(conv_blocks + fc_blocks), dim3(256), 0, stream, d, wk, conv_blocks, a);
		chk(hipGetLastError(), "fused backward launch");
		return true;
	}
	template <class F>
	void for_each(int64_t n, const F& f)
	{
		if (n <= 0) return;
		const int blocks = static_cast<int>(std::min<int64_t>((n + 255) / 256, 2048));
		hipLaunchKernelGGL(tr_foreach_kernel<F>, dim3(blocks), dim3(n == 1 ? 1 : 256), 0, stream, n, f);
		chk(hipGetLastError(), "foreach launch");
	}
};

}  // namespace dtrl_tr

using Backend = dtrl_tr::HipTrainerBE;
#include "dtrl_trainer_capi.inc"
