// dtrl_trainer_emul.cpp -- TESTS ONLY: the native trainer step's operand definitions and sequencing (deepterrainrl_amd/csrc/dtrl_trainer_ops.h,
// dtrl_trainer_core.h) executed by plain host loops, so that the CPU box can check every GEMM's index arithmetic, the label construction and the
// Caffe SGD rule against the torch / numpy restatements without a GPU. Linked into tests/emul/libdtrl_trainer_emul.so, which nothing in the product
// loads (deepterrainrl_amd/hip_trainer.py binds lib/libdtrl.so and raises if the HIP library or device is missing).
#include "dtrl_trainer_core.h"
#include <cstdlib>
#include <string>

namespace dtrl_tr {
struct EmulTrainerBE {
	int device_id = -1;
	std::string err_;
	bool init(std::string&) { return true; }
	bool ok() const { return err_.empty(); }
	const std::string& error() const { return err_; }
	void set_stream(void*) {}
	void drop_graphs() {}
	// (the per-sample fused passes are device code: the check build always runs the layer-by-layer form, which is what the fused kernels are tested against)
	void setup_fused(const NetDims&) {}
	bool fused_forward(const NetDims*, const Work*, int, bool) { return false; }
	bool fused_forward_part(const NetDims*, const Work*, int, bool, int) { return false; }
	bool fused_backward_fc(const NetDims*, const Work*, int) { return false; }
	template <class A> bool fused_backward(const NetDims*, const Work*, const NetDims&, int, const A&) { return false; }
	void fork() {}
	void resume() {}
	void join() {}
	void* alloc_dev(size_t bytes) { return std::calloc(1, bytes ? bytes : 1); }
	void free_dev(void* p) { std::free(p); }
	void* alloc_host(size_t bytes) { return std::calloc(1, bytes ? bytes : 1); }
	void free_host(void* p) { std::free(p); }
	void h2d(void* dst, const void* src, size_t n) { std::memcpy(dst, src, n); }
	void d2h(void* dst, const void* src, size_t n) { std::memcpy(dst, src, n); }
	void d2d(void* dst, const void* src, size_t n) { std::memcpy(dst, src, n); }
	void sync() {}
	void gemm2(const NetDims* d, const Work* wk, const GemmDesc& ga, const GemmDesc& gb) { gemm(d, wk, ga); gemm(d, wk, gb); }
	template <class Fn> void run_graph(int, Fn fn) { fn(); }
	template <class F> void terr_reduce(const NetDims*, const Work*, int n, const F& f) { for_each(n, f); }
	template <class F> void label_loss(int n, const F& f, const float* sq, float scale, float* out) { for_each(n, f); loss_sum(sq, n, scale, out); }
	template <class FP, class FL> void pre_label_loss(int n_pre, const FP& fp, int n_lab, const FL& fl, const float* sq, float scale, float* out) { for_each(n_pre, fp); for_each(n_lab, fl); loss_sum(sq, n_lab, scale, out); }
	void loss_sum(const float* x, int n, float scale, float* out) { float s = 0; for (int i = 0; i < n; ++i) s += x[i]; *out = scale * s; }
	// (the operand switch is folded per instantiation, as in the HIP kernel: the check build is what the CPU suite spends its trainer time in)
	template <int OP> static void gemm_t(const NetDims& d, const Work& wk, GemmDesc g)
	{
		g.op = OP;
		for (int z = 0; z < g.Z; ++z) {
			const int k_begin = g.k0_step ? z * g.k0_step : 0;
			const int k_end = g.k0_step ? (k_begin + g.k0_step < g.K ? k_begin + g.k0_step : g.K) : g.K;
			for (int m = 0; m < g.M; ++m) for (int n = 0; n < g.N; ++n) {
				float acc = 0;
				for (int k = k_begin; k < k_end; ++k) acc = __builtin_fmaf(load_a(d, wk, g, z, m, k), load_b(d, wk, g, z, k, n), acc);
				store_c(d, wk, g, z, m, n, acc);
			}
		}
	}
	void gemm(const NetDims* dp, const Work* wp, const GemmDesc& g)
	{
		switch (g.op) {
#define DTRL_TR_CASE(k) case k: gemm_t<k>(*dp, *wp, g); break;
		DTRL_TR_CASE(0) DTRL_TR_CASE(1) DTRL_TR_CASE(2) DTRL_TR_CASE(3) DTRL_TR_CASE(4) DTRL_TR_CASE(5) DTRL_TR_CASE(6) DTRL_TR_CASE(7)
		DTRL_TR_CASE(8) DTRL_TR_CASE(9) DTRL_TR_CASE(10) DTRL_TR_CASE(11) DTRL_TR_CASE(12) DTRL_TR_CASE(13) DTRL_TR_CASE(14)
#undef DTRL_TR_CASE
		default: break;
		}
	}
	template <class F> void for_each(int64_t n, const F& f) { for (int64_t i = 0; i < n; ++i) f(i); }
};
}  // namespace dtrl_tr

using Backend = dtrl_tr::EmulTrainerBE;
#include "dtrl_trainer_capi.inc"
