// dtrl_trainer_fused.h -- the trainer step's per-SAMPLE fused passes for gfx950 (device code, included by dtrl_trainer.hip only).
//
// Why: the layer-by-layer form (dtrl_trainer_core.h Forward / Backward over tr_gemm_kernel) is ~19 dependent launches per solver step of 5-20 us each, i.e. bound
// by launch count and the dependent chain, not by arithmetic (profiles/r03_trainer.txt: 0.78 TFLOP/s on a 157 TFLOP/s pipe). What makes fusion possible WITHOUT
// device-wide barriers: every layer of the net family maps a sample to a sample -- conv stack, terr_ip0, trunk and heads never mix rows of the batch; only the
// WEIGHT gradients sum over the batch. So
//   tr_fused_forward_kernel     one 1024-thread workgroup per sample runs the whole net with the activations in LDS (53 KB for the dog nets): conv0 -> conv1 ->
//                               conv2 -> terr_ip0 -> concat -> trunk -> heads, workgroup barriers only; writes what the backward pass needs (train pass) or only
//                               the outputs (evaluation passes: target net, actor filter)
//   tr_fused_backward_x_kernel  one workgroup per sample runs the data-gradient chain the same way: dout -> heads -> trunk -> terr_ip0 -> conv2 -> conv1
//   tr_fused_grad_kernel        one thread per FC parameter (32-term sums over the batch, coalesced along the layer's input index), one wavefront per conv
//                               parameter (sum over batch x positions), and the Caffe SGD update in the same thread (or the flat gradient + sample count for the
//                               data-parallel step)
// replace 8 + 8 + 1 launches by 1 + 1 + 1. MEASURED (profiles/r04_trainer.txt, same box): layer-by-layer 2294 Train()/s; fused FORWARD passes 2618/s (70 us per launch
// against ~95 us for the eight layer launches and their gaps) -- the default (DTRL_TRAINER_FUSED=1); fused forward AND backward 2030/s (DTRL_TRAINER_FUSED=2): the backward
// pair costs 132 + 68 us against ~120 us for the paired GEMM launches, because a sample's chain of ~1.7 M multiply-adds runs on ONE compute unit (64 FMA per clock, >= 11 us at
// perfect issue, ~3x that with its addressing / select / LDS instructions) while a batch of 32 occupies 32 of the 256 units; the layer-by-layer form spreads every layer over
// the part. What made the forward kernel go from 136 to 70 us was memory-level parallelism, not arithmetic: conv weights as wave-uniform SCALAR loads, and every dot product
// with all of a lane's weight loads issued before its first FMA (one load per loop trip pays the L2 latency once per element).
// The arithmetic is fp32 FMA on the vector pipe: with one sample per workgroup the FC products are matrix-VECTOR shaped (M = 1) and the conv layers' weights are scalars against
// LDS-resident activations -- there is no dense tile to hand the matrix pipe without giving up the per-sample independence that removes the launches. Summation orders differ
// from the layer-by-layer form (the check build, tests/emul): agreement is at float32 rounding (the -m gpu tests hold unchanged at 2e-5 / 2e-4).
#pragma once
#include "dtrl_trainer_ops.h"

namespace dtrl_tr {

constexpr int kFT = 1024;        // threads per sample workgroup
constexpr int kFMaxPos = 8;      // output positions of a conv layer one thread accumulates

TR_HD inline int pad4(int n) { return (n + 3) & ~3; }   // LDS sub-buffers start on 16-byte boundaries (float4 reads)

struct FusedPlan {               // host-side: is the net inside what the fused kernels are written for, and how much LDS they need
	bool ok = false;
	int lds_bytes = 0;
	int size_a = 0, size_b = 0;  // bufA: act0 / act2 / dy2; bufB: act1 / dy1
};
inline FusedPlan fused_plan(const NetDims& d)
{
	FusedPlan p;
	for (int l = 0; l < 3; ++l) {
		const int cout = d.C[l + 1];
		if (cout <= 0 || cout % 16 != 0 || cout > 32 || d.T[l + 1] > 256) return p;       // a wavefront owns cout / 16 channels, its lanes 4 positions each
		if (l > 0) { const int cin = d.C[l]; if (cin % 16 != 0 || cin > 32 || d.T[l] > 256) return p; }   // backward: by INPUT channel
	}
	if (d.n_flat > kFMaxPos * kFT || d.fc_terr % 2 != 0) return p;
	if (d.fc_terr + d.n_char > 16 * 12 || d.fc_trunk > 16 * 16 || d.fc_head > 8 * 16) return p;
	for (int f = 0; f < d.n_heads; ++f) if (d.head_out[f] > 32) return p;
	if (d.n_flat % 4 != 0 || d.fc_terr <= 0 || kFT % d.fc_terr != 0 || kFT / d.fc_terr < 1) return p;
	if (kFT / d.fc_terr > 64) return p;   // terr_ip0 sums kFT / fc_terr lanes per output with a shuffle reduction: the group must fit in one wavefront; narrower nets take the layer-by-layer path
	if (d.fc_trunk <= 0 || kFT % d.fc_trunk != 0 || kFT / d.fc_trunk > 4 || d.wo_terr % 4 != 0) return p;
	const int nhz = d.n_heads * d.fc_head;
	if (nhz <= 0 || nhz > kFT || d.out_size * 8 > kFT || d.fc_terr > 64 || (d.fc_terr & (d.fc_terr - 1)) != 0) return p;
	p.size_a = pad4(d.C[1] * d.T[1] > d.n_flat ? d.C[1] * d.T[1] : d.n_flat);
	p.size_b = pad4(d.C[2] * d.T[2]);
	// forward: x[S] bufA bufB v[fc_terr + n_char] h[fc_trunk] hz[nhz]; backward: dout[out] dhz[nhz] part[4 fc_trunk] dhs[fc_trunk] dt3[fc_terr] part2[16 fc_terr] bufA bufB
	const int fwd = pad4(d.S) + p.size_a + p.size_b + pad4(d.fc_terr + d.n_char) + pad4(d.fc_trunk) + pad4(nhz);
	const int bwd = pad4(d.out_size) + pad4(nhz) + pad4(4 * d.fc_trunk) + pad4(d.fc_trunk) + pad4(d.fc_terr) + pad4(16 * d.fc_terr) + p.size_a + p.size_b;
	p.lds_bytes = 4 * (fwd > bwd ? fwd : bwd);
	p.ok = p.lds_bytes <= 64 * 1024;
	return p;
}

struct SgdArgs { float* w; float* hist; float* g; const float* rate_mult; const float* decay_mult; float rate, momentum, weight_decay; int apply; float count; };

#if defined(__HIPCC__)
}  // namespace dtrl_tr
#include <hip/hip_runtime.h>
namespace dtrl_tr {
__device__ __forceinline__ float group_sum(float s, int width)   // sum over `width` consecutive lanes (power of two <= 64)
{
	for (int off = width >> 1; off > 0; off >>= 1) s += __shfl_xor(s, off, width);
	return s;
}

// Dot product of a weight row (global, contiguous) with a vector in LDS by kLanes consecutive lanes: lane r takes elements r, r + kLanes, ...; ALL of its (<= kPer) weight
// loads are issued before the first FMA -- a one-load-per-trip loop pays the L2 latency once per element, which is what bounds a one-workgroup-per-sample pass
template <int kLanes, int kPer>
__device__ __forceinline__ float row_dot(const float* __restrict__ W, const float* __restrict__ x, int K, int r)
{
	float w[kPer];
#pragma unroll
	for (int q = 0; q < kPer; ++q) { const int k = r + kLanes * q; w[q] = k < K ? W[k] : 0.0f; }
	float s = 0.0f;
#pragma unroll
	for (int q = 0; q < kPer; ++q) { const int k = r + kLanes * q; s = fmaf(w[q], x[k < K ? k : 0], s); }
	return group_sum(s, kLanes);
}

// A fully connected layer of one sample with kRows weight rows per lane group IN FLIGHT: group g (kLanes consecutive lanes) owns output rows g, g + G, g + 2 G, ... (G groups
// per workgroup) and handles kRows of them per trip -- all their weight loads are issued before the first FMA, so a layer pays the L2 latency once per kRows rows instead of
// once per row (round 6: the FC chain was 24 us of the 70 us forward, ~13 dependent trips of one row each). Per output the summation order is row_dot()'s: bit-identical.
template <int kLanes, int kPer, int kRows, class RowPtr, class Store>
__device__ __forceinline__ void fc_rows(int n_rows, int K, const float* __restrict__ x, RowPtr row_ptr, Store store)
{
	const int tid = static_cast<int>(threadIdx.x), g = tid / kLanes, r = tid % kLanes;
	constexpr int G = kFT / kLanes;
	float xv[kPer];
#pragma unroll
	for (int q = 0; q < kPer; ++q) { const int k = r + kLanes * q; xv[q] = x[k < K ? k : 0]; }
	for (int n0 = g; n0 < n_rows; n0 += G * kRows) {
		float w[kRows][kPer];
#pragma unroll
		for (int t = 0; t < kRows; ++t) {
			const int n = n0 + t * G;
			const float* __restrict__ Wr = row_ptr(n < n_rows ? n : n0);
#pragma unroll
			for (int q = 0; q < kPer; ++q) { const int k = r + kLanes * q; w[t][q] = k < K ? Wr[k] : 0.0f; }
		}
#pragma unroll
		for (int t = 0; t < kRows; ++t) {
			const int n = n0 + t * G;
			float acc = 0.0f;
#pragma unroll
			for (int q = 0; q < kPer; ++q) acc = fmaf(w[t][q], xv[q], acc);
			acc = group_sum(acc, kLanes);
			if (r == 0 && n < n_rows) store(n, acc);
		}
	}
}

// one conv layer of one sample: in [Cin][Tin] (LDS or the input row), out [Cout][Tout] (LDS) = relu(W * in + b); optionally mirrored to global for the backward pass.
// A wavefront owns Cout / 16 output channels (wave-uniform -> their weights arrive as SCALAR loads through the constant cache and feed the FMAs as SGPR operands;
// a per-lane weight address would put one vector load with an L1 / L2 round trip into every step of the dependent chain), its lanes own the positions lane + 64 j;
// an LDS read of the input serves every channel of the wave
constexpr int kFMaxCh = 2, kFMaxJ = 4;
// Round 6: the kernel width is a template parameter. The round-4 form walked (input channel, tap) in a doubly nested run-time loop with ONE scalar weight load and four LDS
// reads in front of every eight FMAs -- each trip paid the scalar-load and LDS latencies in sequence (rocprofv3: 40 us of the 70 us forward were the conv stack, 80+ cycles
// per multiply-add and thread). Now all taps of an input channel are in flight together: 2 KW scalar weights (they arrive as s_load_dwordx4 / x8) and 4 KW LDS values per
// lane, then 8 KW FMAs; the channel loop is unrolled by two so that the next channel's loads overlap the current one's FMAs.
template <int KW, int J>
__device__ __forceinline__ void fused_conv_t(const NetDims& d, const Work& wk, int l, const float* __restrict__ in, float* __restrict__ out, float* __restrict__ gout)
{
	const int tid = static_cast<int>(threadIdx.x), lane = tid & 63;
	const int Cin = d.C[l], Cout = d.C[l + 1], Tin = d.T[l], Tout = d.T[l + 1];
	const int cpw = Cout / (kFT / 64);                                        // channels per wave (1 or 2)
	const int co0 = __builtin_amdgcn_readfirstlane((tid >> 6) * cpw);
	const int K = Cin * KW;
	const float* __restrict__ W = wk.w + d.wo_conv[l] + static_cast<int64_t>(co0) * K;
	float acc[kFMaxCh][J];
	int tt[J];
#pragma unroll
	for (int j = 0; j < J; ++j) { const int t = lane + 64 * j; tt[j] = t < Tout ? t : Tout - 1; acc[0][j] = 0.0f; acc[1][j] = 0.0f; }   // clamped: a surplus slot recomputes the last position and is not stored
	const bool two = cpw > 1;
	const float* __restrict__ W1 = two ? W + K : W;
	// software pipeline over the input channels: channel ci + 1's weights (scalar loads) and inputs (LDS) are requested before channel ci's FMAs are issued
	float w0[KW], w1[KW], a[J][KW], nw0[KW], nw1[KW], na[J][KW];
	// The weights are wave-uniform, but they are fetched with VECTOR loads on purpose (an opaque zero in a VGPR hides the uniformity): scalar loads and LDS reads share
	// one completion counter (lgkmcnt) and return out of order with respect to each other, so a wave that has both in flight can only wait for ALL of them -- the prefetch
	// of channel ci + 1 would be waited for in front of channel ci's FMAs. Vector loads count on vmcnt: the two streams pipeline independently. All lanes read the same
	// 16 bytes (one cache line, a broadcast).
	int vzero;
	asm volatile("v_mov_b32 %0, 0" : "=v"(vzero));
	static_assert(KW % 4 == 0, "float4 weight loads");
	auto fetch = [&](int ci, float (&x0)[KW], float (&x1)[KW], float (&xa)[J][KW]) {
		const float* __restrict__ row = in + ci * Tin;
		const float4* __restrict__ p0 = reinterpret_cast<const float4*>(W + ci * KW) + vzero;
		const float4* __restrict__ p1 = reinterpret_cast<const float4*>(W1 + ci * KW) + vzero;
#pragma unroll
		for (int q = 0; q < KW / 4; ++q) {
			const float4 v0 = p0[q], v1 = p1[q];
			x0[4 * q] = v0.x; x0[4 * q + 1] = v0.y; x0[4 * q + 2] = v0.z; x0[4 * q + 3] = v0.w;
			x1[4 * q] = v1.x; x1[4 * q + 1] = v1.y; x1[4 * q + 2] = v1.z; x1[4 * q + 3] = v1.w;
		}
#pragma unroll
		for (int j = 0; j < J; ++j)
#pragma unroll
			for (int u = 0; u < KW; ++u) xa[j][u] = row[tt[j] + u];
	};
	fetch(0, w0, w1, a);
	for (int ci = 0; ci < Cin; ++ci) {
		fetch(ci + 1 < Cin ? ci + 1 : ci, nw0, nw1, na);   // (the last trip re-reads itself: no branch around the loads)
#pragma unroll
		for (int u = 0; u < KW; ++u)
#pragma unroll
			for (int j = 0; j < J; ++j) { acc[0][j] = fmaf(w0[u], a[j][u], acc[0][j]); acc[1][j] = fmaf(w1[u], a[j][u], acc[1][j]); }
#pragma unroll
		for (int u = 0; u < KW; ++u) { w0[u] = nw0[u]; w1[u] = nw1[u]; }
#pragma unroll
		for (int j = 0; j < J; ++j)
#pragma unroll
			for (int u = 0; u < KW; ++u) a[j][u] = na[j][u];
	}
	for (int c = 0; c < cpw; ++c) {
		const int co = co0 + c;
		const float b = wk.w[d.bo_conv[l] + co];
#pragma unroll
		for (int j = 0; j < J; ++j) {
			const int t = lane + 64 * j;
			if (t < Tout) { float v = (c == 0 ? acc[0][j] : acc[1][j]) + b; v = v > 0 ? v : 0.0f; out[co * Tout + t] = v; if (gout) gout[co * Tout + t] = v; }
		}
	}
}
__device__ __forceinline__ void fused_conv(const NetDims& d, const Work& wk, int l, const float* __restrict__ in, float* __restrict__ out, float* __restrict__ gout)
{
	const int Kw = d.Kw[l];   // wave-uniform: 8 / 4 / 4 in every shipped net; other widths take the general loop
	const int slots = (d.T[l + 1] + 63) / 64;   // position slots a lane owns: 3 for conv1 / conv2 of the shipped nets (190 / 187 outputs), 4 for conv0 (193)
	if (Kw == 4 && slots <= 3) { fused_conv_t<4, 3>(d, wk, l, in, out, gout); return; }
	if (Kw == 4) { fused_conv_t<4, 4>(d, wk, l, in, out, gout); return; }
	if (Kw == 8 && slots <= 3) { fused_conv_t<8, 3>(d, wk, l, in, out, gout); return; }
	if (Kw == 8) { fused_conv_t<8, 4>(d, wk, l, in, out, gout); return; }
	const int tid = static_cast<int>(threadIdx.x), lane = tid & 63;
	const int Cin = d.C[l], Cout = d.C[l + 1], Tin = d.T[l], Tout = d.T[l + 1];
	const int cpw = Cout / (kFT / 64);
	const int co0 = __builtin_amdgcn_readfirstlane((tid >> 6) * cpw);
	const int K = Cin * Kw;
	const float* __restrict__ W = wk.w + d.wo_conv[l] + static_cast<int64_t>(co0) * K;
	float acc[kFMaxCh][kFMaxJ];
	int tt[kFMaxJ];
#pragma unroll
	for (int j = 0; j < kFMaxJ; ++j) { const int t = lane + 64 * j; tt[j] = t < Tout ? t : Tout - 1; acc[0][j] = 0.0f; acc[1][j] = 0.0f; }
	const bool two = cpw > 1;
	for (int ci = 0; ci < Cin; ++ci) {
		const float* __restrict__ row = in + ci * Tin;
		for (int u = 0; u < Kw; ++u) {
			const float w0 = W[ci * Kw + u];
			const float w1 = two ? W[K + ci * Kw + u] : 0.0f;
#pragma unroll
			for (int j = 0; j < kFMaxJ; ++j) { const float a = row[tt[j] + u]; acc[0][j] = fmaf(w0, a, acc[0][j]); acc[1][j] = fmaf(w1, a, acc[1][j]); }
		}
	}
	for (int c = 0; c < cpw; ++c) {
		const int co = co0 + c;
		const float b = wk.w[d.bo_conv[l] + co];
#pragma unroll
		for (int j = 0; j < kFMaxJ; ++j) {
			const int t = lane + 64 * j;
			if (t < Tout) { float v = (c == 0 ? acc[0][j] : acc[1][j]) + b; v = v > 0 ? v : 0.0f; out[co * Tout + t] = v; if (gout) gout[co * Tout + t] = v; }
		}
	}
}

// kStore: the train pass (activations kept for the backward pass); else only `out`.
// kPart (round 6): 0 = the whole net in one launch (rounds 4-5); 1 = the conv stack alone -- conv2's output always goes to global memory, where 2 = the split-K GEMM of
// terr_ip0 over ALL rows of the pass (tr_gemm_kernel<kTerrFwd>: the 1.5 MB matrix is read once per 32-row tile instead of once per SAMPLE through one compute unit) finds
// it; 3 = the FC chain: t3 = relu(sum of the GEMM's partials + bias), concat, trunk, heads. DTRL_TRAINER_FUSED=3 selects 1 + GEMM + 3.
template <bool kStore, int kPart = 0>
__global__ void __launch_bounds__(kFT) tr_fused_forward_kernel(const NetDims* __restrict__ dp, const Work* __restrict__ wp, int size_a, int size_b_dbg)
{
	extern __shared__ float sm[];
	const NetDims& d = *dp; const Work& wk = *wp;
	const int size_b = size_b_dbg & 0xffffff, dbg = size_b_dbg >> 24;   // dbg (timing experiments only, DTRL_TRAINER_DBG): 1..3 = leave after conv layer dbg - 1; 4 = after loading x
	const int z = static_cast<int>(blockIdx.x), tid = static_cast<int>(threadIdx.x);
	const int nhz = d.n_heads * d.fc_head, kc = d.fc_terr + d.n_char;
	float* x = sm; float* bufA = x + pad4(d.S); float* bufB = bufA + size_a; float* v = bufB + size_b; float* h = v + pad4(kc); float* hz = h + pad4(d.fc_trunk);
	for (int i = tid; i < d.S; i += kFT) x[i] = wk.xin[static_cast<size_t>(z) * d.S + i];
	__syncthreads();
	if (dbg == 4) return;
	if (kPart != 3) {
	fused_conv(d, wk, 0, x, bufA, kStore ? wk.act[0] + static_cast<size_t>(z) * d.C[1] * d.T[1] : nullptr);
	__syncthreads();
	if (dbg == 1) return;
	fused_conv(d, wk, 1, bufA, bufB, kStore ? wk.act[1] + static_cast<size_t>(z) * d.C[2] * d.T[2] : nullptr);
	__syncthreads();
	if (dbg == 2) return;
	fused_conv(d, wk, 2, bufB, bufA, (kStore || kPart == 1) ? wk.act[2] + static_cast<size_t>(z) * d.n_flat : nullptr);
	if (kPart == 1) return;
	__syncthreads();
	}
	if (kPart == 3) {   // t3 from the split-K partials [row][fc_terr][n_slabs]: 16 lanes per output read its contiguous run of partials
		const int per = kFT / 16;
		for (int n = tid >> 4; n < d.fc_terr; n += per) {
			const float* __restrict__ pp = wk.tp + (static_cast<size_t>(z) * d.fc_terr + n) * d.n_slabs;
			float s = 0.0f;
			for (int q = tid & 15; q < d.n_slabs; q += 16) s += pp[q];
			s = group_sum(s, 16);
			if ((tid & 15) == 0) { float t = s + wk.w[d.bo_terr + n]; t = t > 0 ? t : 0.0f; v[n] = t; if (kStore) wk.t3[static_cast<size_t>(z) * d.fc_terr + n] = t; }
		}
	} else
	{   // terr_ip0: fc_terr outputs x n_flat inputs; kFT / fc_terr lanes per output walk the weight row in float4 steps (coalesced), the flattened conv2 output sits in LDS
		const int tpo = kFT / d.fc_terr, n = tid / tpo, p = tid - n * tpo;
		const float4* __restrict__ W4 = reinterpret_cast<const float4*>(wk.w + d.wo_terr + static_cast<int64_t>(n) * d.n_flat);
		const float4* __restrict__ a4 = reinterpret_cast<const float4*>(bufA);
		// eight independent weight loads in flight per lane (a row of 1.5 MB comes from L2: one load per trip would pay the full latency ~94 times over)
		float s = 0.0f;
		const int n4 = d.n_flat / 4;
		for (int k4 = p; k4 < n4; k4 += 8 * tpo) {
			float4 w[8], a[8];
#pragma unroll
			for (int q = 0; q < 8; ++q) { const int kk = k4 + q * tpo; const bool in_range = kk < n4; w[q] = W4[in_range ? kk : p]; a[q] = a4[in_range ? kk : p]; if (!in_range) w[q] = make_float4(0, 0, 0, 0); }
#pragma unroll
			for (int q = 0; q < 8; ++q) { s = fmaf(w[q].x, a[q].x, s); s = fmaf(w[q].y, a[q].y, s); s = fmaf(w[q].z, a[q].z, s); s = fmaf(w[q].w, a[q].w, s); }
		}
		s = group_sum(s, tpo);
		if (p == 0) { float t = s + wk.w[d.bo_terr + n]; t = t > 0 ? t : 0.0f; v[n] = t; if (kStore) wk.t3[static_cast<size_t>(z) * d.fc_terr + n] = t; }
	}
	for (int j = tid; j < d.n_char; j += kFT) v[d.fc_terr + j] = x[d.n_terr + j];
	__syncthreads();
	// trunk, head0, head1: 16 (8) lanes per output row, 64 (128) rows per pass
	// trunk (K = fc_terr + n_char <= 16 x 12) and head0 (K = fc_trunk <= 16 x 16): 16 lanes per output row, four rows of a lane group in flight (fc_rows)
	fc_rows<16, 12, 4>(d.fc_trunk, kc, v,
		[&](int n) { return wk.w + d.wo_ip0 + static_cast<int64_t>(n) * kc; },
		[&](int n, float s) { float t = s + wk.w[d.bo_ip0 + n]; t = t > 0 ? t : 0.0f; h[n] = t; if (kStore) wk.h[static_cast<size_t>(z) * d.fc_trunk + n] = t; });
	__syncthreads();
	fc_rows<16, 16, 4>(nhz, d.fc_trunk, h,
		[&](int o) { const int f = o / d.fc_head, n = o - f * d.fc_head; return wk.w + d.wo_h0[f] + static_cast<int64_t>(n) * d.fc_trunk; },
		[&](int o, float s) { const int f = o / d.fc_head, n = o - f * d.fc_head; float t = s + wk.w[d.bo_h0[f] + n]; t = t > 0 ? t : 0.0f; hz[o] = t;
		                       if (kStore) wk.hz[(static_cast<size_t>(f) * wk.max_rows + z) * d.fc_head + n] = t; });
	__syncthreads();
	{   // head1: out_size outputs x fc_head (<= 8 x 16), 8 lanes each
		const int o = tid >> 3, p = tid & 7;
		if (o < d.out_size) {
			int f = 0; while (f + 1 < d.n_heads && o >= d.out_off[f + 1]) ++f;
			const int j = o - d.out_off[f];
			const float s = row_dot<8, 16>(wk.w + d.wo_h1[f] + static_cast<int64_t>(j) * d.fc_head, hz + f * d.fc_head, d.fc_head, p);
			if (p == 0) wk.out[static_cast<size_t>(z) * d.out_size + o] = s + wk.w[d.bo_h1[f] + j];
		}
	}
}

// data gradients of one sample, dout -> dy0. Everything the weight-gradient pass reads is written to global: dhz, dhs, dt3, dy[2], dy[1], dy[0]
// fc_only (round 6, DTRL_TRAINER_FUSED=4): stop behind dt3 -- the sample's data-gradient chain through heads and trunk (dhz, dhs, dt3: 0.18 M multiply-adds) in ONE launch
// instead of the four dependent GEMM launches head1 / head0 / dh_sum / trunk; terr_ip0 and the conv layers stay layer-by-layer (whole-part GEMMs)
__global__ void __launch_bounds__(kFT) tr_fused_backward_x_kernel(const NetDims* __restrict__ dp, const Work* __restrict__ wp, int size_a, int size_b_fc)
{
	extern __shared__ float sm[];
	const NetDims& d = *dp; const Work& wk = *wp;
	const int size_b = size_b_fc & 0xffffff; const int fc_stage = size_b_fc >> 24; const bool fc_only = fc_stage != 0;   // (fc_stage 2 / 3: timing experiments, leave after head1^T / head0^T)
	const int z = static_cast<int>(blockIdx.x), tid = static_cast<int>(threadIdx.x);
	const int nhz = d.n_heads * d.fc_head, kc = d.fc_terr + d.n_char;
	float* dout = sm; float* dhz = dout + pad4(d.out_size); float* part = dhz + pad4(nhz); float* dhs = part + pad4(4 * d.fc_trunk); float* dt3 = dhs + pad4(d.fc_trunk);
	float* part2 = dt3 + pad4(d.fc_terr); float* bufA = part2 + pad4(16 * d.fc_terr); float* bufB = bufA + size_a;
	for (int i = tid; i < d.out_size; i += kFT) dout[i] = wk.dout[static_cast<size_t>(z) * d.out_size + i];
	// per-head offsets as LDS tables (round 6): indexed by a per-lane head number they were global loads IN FRONT of every weight load (a load waiting for a load)
	__shared__ long long s_wo_h1[kMaxHeads];
	__shared__ int s_head_out[kMaxHeads], s_out_off[kMaxHeads];
	if (tid < d.n_heads) { s_wo_h1[tid] = d.wo_h1[tid]; s_head_out[tid] = d.head_out[tid]; s_out_off[tid] = d.out_off[tid]; }
	__syncthreads();
	if (tid < nhz) {   // head1^T: dhz[f][n] = relu'(hz) sum_j W1_f[j][n] dout[off_f + j]   (threads along n: coalesced weight reads)
		const int f = tid / d.fc_head, n = tid - f * d.fc_head;
		const float* __restrict__ W = wk.w + s_wo_h1[f] + n;
		const int nj = s_head_out[f];                 // <= 32 (plan)
		float w[32];
#pragma unroll
		for (int j = 0; j < 32; ++j) w[j] = j < nj ? W[static_cast<int64_t>(j) * d.fc_head] : 0.0f;
		float s = 0.0f;
		const int o0 = s_out_off[f];
#pragma unroll
		for (int j = 0; j < 32; ++j) s = fmaf(w[j], dout[o0 + (j < nj ? j : 0)], s);
		const size_t gi = (static_cast<size_t>(f) * wk.max_rows + z) * d.fc_head + n;
		s = wk.hz[gi] > 0 ? s : 0.0f;
		dhz[tid] = s; wk.dhz[gi] = s;
	}
	if (fc_stage == 2) return;
	__syncthreads();
	{   // head0^T, summed over the heads: dhs[n] = relu'(h) sum_{f, m} W0_f[m][n] dhz[f][m]; kFT / fc_trunk partial sums per n (threads along n)
		// Round 6: the weight row of flat index q = f fc_head + m is found with a shift (fc_head is a power of two in every shipped net) and a head-offset table in LDS -- the
		// round-4 form divided by fc_head and fetched d.wo_h0[f] from global memory per element, i.e. every weight load waited for ANOTHER load (46 us for this kernel's FC
		// part); and a thread's whole share of the sum (nhz / parts weights, <= 128) is requested in two trips of 64 loads instead of eight trips of 16.
		__shared__ long long s_wo_h0[kMaxHeads];
		if (tid < d.n_heads) s_wo_h0[tid] = d.wo_h0[tid];
		__syncthreads();
		const int parts = kFT / d.fc_trunk, n = tid % d.fc_trunk, p = tid / d.fc_trunk;
		const bool pow2 = (d.fc_head & (d.fc_head - 1)) == 0;
		const int sh = pow2 ? __builtin_ctz(static_cast<unsigned>(d.fc_head)) : 0;
		float s = 0.0f;
		constexpr int kIn = 64;                                 // weight loads in flight per trip
		for (int q0 = p; q0 < nhz; q0 += kIn * parts) {
			float w[kIn];
#pragma unroll
			for (int r = 0; r < kIn; ++r) {
				const int q = q0 + r * parts; const int qq = q < nhz ? q : p;
				const int f = pow2 ? (qq >> sh) : (qq / d.fc_head), m = qq - f * d.fc_head;
				w[r] = q < nhz ? wk.w[s_wo_h0[f] + static_cast<int64_t>(m) * d.fc_trunk + n] : 0.0f;
			}
#pragma unroll
			for (int r = 0; r < kIn; ++r) { const int q = q0 + r * parts; s = fmaf(w[r], dhz[q < nhz ? q : p], s); }
		}
		part[p * d.fc_trunk + n] = s;
		__syncthreads();
		if (tid < d.fc_trunk) {
			float t = 0.0f;
			for (int q = 0; q < parts; ++q) t += part[q * d.fc_trunk + tid];
			t = wk.h[static_cast<size_t>(z) * d.fc_trunk + tid] > 0 ? t : 0.0f;
			dhs[tid] = t; wk.dhs[static_cast<size_t>(z) * d.fc_trunk + tid] = t;
		}
	}
	if (fc_stage == 3) return;
	__syncthreads();
	{   // trunk^T (terrain part only): dt3[k] = relu'(t3) sum_n Wip0[n][k] dhs[n]; 16 partial sums per k
		const int k = tid % d.fc_terr, p = tid / d.fc_terr;      // p < kFT / fc_terr (>= 16)
		if (p < 16) {
			float s = 0.0f;
			for (int n0 = p; n0 < d.fc_trunk; n0 += 16 * 16) {   // sixteen weight loads in flight
				float w[16];
#pragma unroll
				for (int r = 0; r < 16; ++r) { const int n = n0 + 16 * r; w[r] = n < d.fc_trunk ? wk.w[d.wo_ip0 + static_cast<int64_t>(n) * kc + k] : 0.0f; }
#pragma unroll
				for (int r = 0; r < 16; ++r) { const int n = n0 + 16 * r; s = fmaf(w[r], dhs[n < d.fc_trunk ? n : p], s); }
			}
			part2[p * d.fc_terr + k] = s;
		}
		__syncthreads();
		if (tid < d.fc_terr) {
			float t = 0.0f;
			for (int q = 0; q < 16; ++q) t += part2[q * d.fc_terr + tid];
			t = wk.t3[static_cast<size_t>(z) * d.fc_terr + tid] > 0 ? t : 0.0f;
			dt3[tid] = t; wk.dt3[static_cast<size_t>(z) * d.fc_terr + tid] = t;
		}
	}
	if (fc_only) return;
	__syncthreads();
	{   // terr_ip0^T: dy2[i] = relu'(act2) sum_k Wt[k][i] dt3[k]  (threads along i: coalesced); a thread's outputs i = tid + 1024 j advance together, so that
		// kFMaxPos x 2 weight loads are in flight instead of one
		float acc[kFMaxPos];
		int ii[kFMaxPos];
#pragma unroll
		for (int j = 0; j < kFMaxPos; ++j) { acc[j] = 0.0f; const int i = tid + kFT * j; ii[j] = i < d.n_flat ? i : tid; }
		const float* __restrict__ Wt = wk.w + d.wo_terr;
		for (int k = 0; k < d.fc_terr; k += 2) {
			const float g0 = dt3[k], g1 = dt3[k + 1];
			float w0[kFMaxPos], w1[kFMaxPos];
#pragma unroll
			for (int j = 0; j < kFMaxPos; ++j) { w0[j] = Wt[static_cast<int64_t>(k) * d.n_flat + ii[j]]; w1[j] = Wt[static_cast<int64_t>(k + 1) * d.n_flat + ii[j]]; }
#pragma unroll
			for (int j = 0; j < kFMaxPos; ++j) { acc[j] = fmaf(w0[j], g0, acc[j]); acc[j] = fmaf(w1[j], g1, acc[j]); }
		}
#pragma unroll
		for (int j = 0; j < kFMaxPos; ++j) {
			const int i = tid + kFT * j;
			if (i < d.n_flat) { const size_t gi = static_cast<size_t>(z) * d.n_flat + i; const float v = wk.act[2][gi] > 0 ? acc[j] : 0.0f; bufA[i] = v; wk.dy[2][gi] = v; }
		}
	}
	__syncthreads();
	// conv l^T for l = 2, 1: dy_{l-1}[ci][t] = relu'(act_{l-1}[ci][t]) sum_{co, u} w_l[co][ci][u] dy_l[co][t - u]; a wavefront owns Cin / 16 INPUT channels (scalar weights)
	for (int l = 2; l >= 1; --l) {
		const float* __restrict__ src = (l == 2) ? bufA : bufB;
		float* __restrict__ dst = (l == 2) ? bufB : nullptr;
		const int Cin = d.C[l], Cout = d.C[l + 1], Kw = d.Kw[l], Tin = d.T[l], Tout = d.T[l + 1];
		const int lane = tid & 63, cpw = Cin / (kFT / 64);
		const int ci0 = __builtin_amdgcn_readfirstlane((tid >> 6) * cpw);
		const bool two = cpw > 1;
		float acc[kFMaxCh][kFMaxJ];
#pragma unroll
		for (int j = 0; j < kFMaxJ; ++j) { acc[0][j] = 0.0f; acc[1][j] = 0.0f; }
		for (int co = 0; co < Cout; ++co) {
			const float* __restrict__ row = src + co * Tout;
			const float* __restrict__ W = wk.w + d.wo_conv[l] + (static_cast<int64_t>(co) * Cin + ci0) * Kw;
			for (int u = 0; u < Kw; ++u) {
				const float w0 = W[u];
				const float w1 = two ? W[Kw + u] : 0.0f;
#pragma unroll
				for (int j = 0; j < kFMaxJ; ++j) {
					const int t = lane + 64 * j - u;
					const float g = (t >= 0 && t < Tout) ? row[t] : 0.0f;
					acc[0][j] = fmaf(w0, g, acc[0][j]); acc[1][j] = fmaf(w1, g, acc[1][j]);
				}
			}
		}
		for (int c = 0; c < cpw; ++c) {
			const int ci = ci0 + c;
			const size_t base = (static_cast<size_t>(z) * Cin + ci) * Tin;
#pragma unroll
			for (int j = 0; j < kFMaxJ; ++j) {
				const int t = lane + 64 * j;
				if (t < Tin) {
					const float v = wk.act[l - 1][base + t] > 0 ? (c == 0 ? acc[0][j] : acc[1][j]) : 0.0f;
					if (dst) dst[ci * Tin + t] = v;
					wk.dy[l - 1][base + t] = v;
				}
			}
		}
		__syncthreads();
	}
}

// sum over the batch rows of a[z sa] b[z sb] (b == nullptr: of a alone), eight rows' loads in flight
__device__ __forceinline__ float rows_dot(const float* __restrict__ a, size_t sa, const float* __restrict__ b, size_t sb, int rows)
{
	float s = 0.0f;
	for (int z0 = 0; z0 < rows; z0 += 8) {
		float x[8], y[8];
#pragma unroll
		for (int r = 0; r < 8; ++r) { const int zz = z0 + r; const bool ok = zz < rows; x[r] = ok ? a[static_cast<size_t>(zz) * sa] : 0.0f; y[r] = b ? (ok ? b[static_cast<size_t>(zz) * sb] : 0.0f) : 1.0f; }
#pragma unroll
		for (int r = 0; r < 8; ++r) s = fmaf(x[r], y[r], s);
	}
	return s;
}

// weight gradients (sums over the batch) + the Caffe SGD rule, or (apply == 0) the flat gradient with the sample count behind it.
// blocks [0, conv_blocks): one wavefront per conv parameter; the rest: one thread per parameter from wo_terr on
__global__ void __launch_bounds__(256) tr_fused_grad_kernel(const NetDims* __restrict__ dp, const Work* __restrict__ wp, int conv_blocks, SgdArgs a)
{
	const NetDims& d = *dp; const Work& wk = *wp;
	const int rows = wk.rows;
	float g = 0.0f;
	int64_t i = -1;
	if (static_cast<int>(blockIdx.x) < conv_blocks) {
		const int64_t e = static_cast<int64_t>(blockIdx.x) * 4 + (threadIdx.x >> 6);
		const int lane = static_cast<int>(threadIdx.x) & 63;
		if (e < d.wo_terr) {
			int l = 0; while (l < 2 && e >= d.wo_conv[l + 1]) ++l;
			const int Cin = d.C[l], Cout = d.C[l + 1], Kw = d.Kw[l], Tin = d.T[l], Tout = d.T[l + 1];
			const bool bias = e >= d.bo_conv[l];
			int co, ci = 0, u = 0;
			if (bias) co = static_cast<int>(e - d.bo_conv[l]);
			else { const int64_t o = e - d.wo_conv[l]; co = static_cast<int>(o / (Cin * Kw)); const int r = static_cast<int>(o - static_cast<int64_t>(co) * Cin * Kw); ci = r / Kw; u = r - ci * Kw; }
			float s = 0.0f;
			const int total = rows * Tout;
			for (int q = lane; q < total; q += 64) {
				const int zz = q / Tout, t = q - zz * Tout;
				const float dyv = wk.dy[l][(static_cast<size_t>(zz) * Cout + co) * Tout + t];
				const float in = bias ? 1.0f : (l == 0 ? wk.xin[static_cast<size_t>(zz) * d.S + t + u] : wk.act[l - 1][(static_cast<size_t>(zz) * Cin + ci) * Tin + t + u]);
				s = fmaf(dyv, in, s);
			}
			s = group_sum(s, 64);
			if (lane == 0) { g = s; i = e; }
		}
	} else {
		const int64_t e = d.wo_terr + (static_cast<int64_t>(blockIdx.x) - conv_blocks) * 256 + threadIdx.x;
		if (e < d.num_params) {
			i = e;
			const int kc = d.fc_terr + d.n_char;
			float s = 0.0f;
			if (e < d.wo_ip0) {            // terr_ip0: delta dt3 [z][fc_terr], input act2 [z][n_flat]
				if (e < d.bo_terr) { const int m = static_cast<int>((e - d.wo_terr) / d.n_flat), n = static_cast<int>((e - d.wo_terr) - static_cast<int64_t>(m) * d.n_flat);
					s = rows_dot(wk.dt3 + m, d.fc_terr, wk.act[2] + n, d.n_flat, rows); }
				else { const int m = static_cast<int>(e - d.bo_terr); s = rows_dot(wk.dt3 + m, d.fc_terr, nullptr, 0, rows); }
			} else if (e < d.wo_h0[0]) {   // trunk: delta dhs [z][fc_trunk], input concat(t3, x_char)
				if (e < d.bo_ip0) { const int m = static_cast<int>((e - d.wo_ip0) / kc), n = static_cast<int>((e - d.wo_ip0) - static_cast<int64_t>(m) * kc);
					const bool terr = n < d.fc_terr;
					s = rows_dot(wk.dhs + m, d.fc_trunk, terr ? wk.t3 + n : wk.xin + d.n_terr + (n - d.fc_terr), terr ? d.fc_terr : d.S, rows); }
				else { const int m = static_cast<int>(e - d.bo_ip0); s = rows_dot(wk.dhs + m, d.fc_trunk, nullptr, 0, rows); }
			} else {
				int f = 0; while (f + 1 < d.n_heads && e >= d.wo_h0[f + 1]) ++f;
				if (e < d.wo_h1[f]) {       // head0_f: delta dhz[f] [z][fc_head], input h [z][fc_trunk]
					if (e < d.bo_h0[f]) { const int m = static_cast<int>((e - d.wo_h0[f]) / d.fc_trunk), n = static_cast<int>((e - d.wo_h0[f]) - static_cast<int64_t>(m) * d.fc_trunk);
						s = rows_dot(wk.dhz + static_cast<size_t>(f) * wk.max_rows * d.fc_head + m, d.fc_head, wk.h + n, d.fc_trunk, rows); }
					else { const int m = static_cast<int>(e - d.bo_h0[f]); s = rows_dot(wk.dhz + static_cast<size_t>(f) * wk.max_rows * d.fc_head + m, d.fc_head, nullptr, 0, rows); }
				} else {                    // head1_f: delta dout [z][off_f + m], input hz[f] [z][fc_head]
					if (e < d.bo_h1[f]) { const int m = static_cast<int>((e - d.wo_h1[f]) / d.fc_head), n = static_cast<int>((e - d.wo_h1[f]) - static_cast<int64_t>(m) * d.fc_head);
						s = rows_dot(wk.dout + d.out_off[f] + m, d.out_size, wk.hz + static_cast<size_t>(f) * wk.max_rows * d.fc_head + n, d.fc_head, rows); }
					else { const int m = static_cast<int>(e - d.bo_h1[f]); s = rows_dot(wk.dout + d.out_off[f] + m, d.out_size, nullptr, 0, rows); }
				}
			}
			g = s;
		}
	}
	if (i < 0) return;
	if (a.apply) { a.g[i] = g; sgd_elem(a.w, a.hist, a.g, a.rate_mult, a.decay_mult, a.rate, a.momentum, a.weight_decay, i); }
	else { a.g[i] = g; if (i == 0) a.g[d.num_params] = a.count; }
}
#endif

}  // namespace dtrl_tr
