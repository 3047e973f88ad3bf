// dtrl_kernel_fast.h -- gfx950 fast path of the per-env dense linear algebra (device code only).
//
// The lane-phase reference path (dtrl_kernel.h) keeps the joint-space inertia matrix in LDS and pays one LDS round trip
// + barrier per elimination step (rocprof/s_memtime: factorisation 20 %, forward substitution 17 %, serial row-list
// construction 15 % of an env-step). Here lane i keeps ROW i of H in VGPRs (23 doubles) and the elimination uses
// v_readlane broadcasts of the pivot column, so an LDL^T factorisation is ~250 broadcast+FMA pairs with no LDS traffic
// and no barrier; triangular solves are 22 broadcast+FMA steps each. Constraint rows are compacted with wave ballots.
// The arithmetic (operation order included) is IDENTICAL to the reference path, so both produce the same bits; the
// reference kernel stays in the library (DTRL_KERNEL=ref) and tests/test_gpu_parity.py compares the two.
//
// The factorisation is H = U D U^T, eliminating the last DoF first (leaf-to-root: no fill-in, factorize() in dtrl_kernel.h); the
// kernel is instantiated per skeleton (dtrl_topo.h) and emits only the structurally non-zero updates.
// Row layout after factorize_regs(): h[k] for k > lane = U_{lane,k}; h[lane] = d_lane; h[k] for k < lane = U_{k,lane}
// (the transpose copy, so the second substitution also only needs the lane's own registers).
#pragma once
#include "dtrl_kernel.h"
#include "dtrl_topo.h"
#include <type_traits>

#if defined(__HIP_DEVICE_COMPILE__)

#if !DTRL_WAVE_SYNC
#error "dtrl_kernel_fast.h: the register-resident path is ONE wavefront per env by construction (v_readlane broadcasts, ballots, and LDS read-then-write sequences such as warm_match_fast() that rely on a wave's LDS operations completing in issue order). -DDTRL_WAVE_SYNC=0 builds the lane-phase reference path only (ADVICE r5)"
#endif

namespace dtrl {

__device__ __forceinline__ int lane_id() { return static_cast<int>(threadIdx.x); }
// the EXEC-window helpers below are spelled per arithmetic type (dtrl_types.h: fp64 shipped, fp32 opt-in build)
#if defined(DTRL_REAL_F32)
#define DTRL_VFMA "v_fma_f32"
#define DTRL_VMOV "v_mov_b32"
#else
#define DTRL_VFMA "v_fma_f64"
#define DTRL_VMOV "v_mov_b64"
#endif
__device__ __forceinline__ real bcast(real v, int src)   // src must be wave-uniform
{
#if defined(DTRL_REAL_F32)
	return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src));
#else
	int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
	int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
	return __hiloint2double(hi, lo);
#endif
}

// row `lane` of the joint-space inertia matrix from the composite quantities in LDS (same formulas and operand order as
// mass_matrix()). Only the upper triangle (column >= lane) is produced: the leaf-first factorisation never reads the rest.
// The (link, ancestor) pairs are dealt round-robin to all 64 lanes, each evaluates the closed form once and parks the value in a
// packed table in LDS (the storage of the Delassus matrix, dead at that point):
//   entries of link l (DoF l + 2) at base_l = l (l + 5) / 2: [hx, hy, H(l, link 0), ..., H(l, link l)], non-ancestor slots = 0
// i.e. the slot of DoF pair (c, d), d <= c, is base_{c-2} + d: a lane assembles its row with D loads at compile-time offsets from its
// own lane id (no per-column index arithmetic, no branches: the loads pipeline). Lanes >= D copy row 0 (finite, never read).
template <int D>
__device__ __forceinline__ void mass_row(WSFast& ws, real (&h)[D])
{
	constexpr int Lk = D - 2;                                  // links of the character this kernel is instantiated for
	constexpr int kTab = Lk * (Lk + 5) / 2;
	constexpr int kFill = (kTab + 1) & ~1;                     // even count (16-byte stores)
	static_assert(kFill <= static_cast<int>(sizeof(ws.Apk) / sizeof(real)), "mass table does not fit the packed-matrix storage");
	const int d = static_cast<int>(threadIdx.x);
	const int l = d >= 2 ? d - 2 : 0;
	const bool hinge = d >= 2 && d < D;
	real* T = ws.Apk;
	{
		struct alignas(2 * sizeof(real)) real2 { real x, y; };
		real2* T2 = reinterpret_cast<real2*>(T);
		const real2 z2 = {0.0, 0.0};
#pragma unroll
		for (int e = 0; e < (kFill / 2 + kGroup - 1) / kGroup; ++e) if (d + e * kGroup < kFill / 2) T2[d + e * kGroup] = z2;
	}
	env_sync();
	if (hinge) {
		const int base = l * (l + 5) / 2;
		const real m = ws.sm[l], mx = ws.smx[l], my = ws.smy[l];
		T[base] = -(my - m * ws.py[l]);
		T[base + 1] = (mx - m * ws.px[l]);
	}
	const int n_pairs = ws.M.n_pairs;
#pragma unroll 2
	for (int e = lane_id(); e < n_pairs; e += kGroup) {
		const int pl = ws.M.pair_l[e], pk = ws.M.pair_k[e];
		const int a = ws.M.path[pl][pk];
		const real m = ws.sm[pl], mx = ws.smx[pl], my = ws.smy[pl], I = ws.sI[pl];
		const real plx = ws.px[pl], ply = ws.py[pl];
		const real pax = ws.px[a], pay = ws.py[a];
		T[pl * (pl + 5) / 2 + 2 + a] = I - ((plx + pax) * mx + (ply + pay) * my) + m * (plx * pax + ply * pay);
	}
	env_sync();
	const real M0 = ws.sm[0];
	const real* Tl = T + (d < D ? d : 0);
	h[0] = (d == 0) ? M0 : 0.0;
	h[1] = (d == 1) ? M0 : 0.0;
#pragma unroll
	for (int c = 2; c < D; ++c) h[c] = Tl[(c - 2) * (c + 3) / 2];   // base_{c-2} + lane
	env_sync();   // T is dead; the storage may be reused
}

// ---- lane-predicated updates through EXEC ----
// The elimination steps below update "the lanes below the pivot". Written as `if (lane > k) x = fma(...)` the compiler emits the FMA
// for every lane plus two v_cndmask per double (and keeps 23 lane masks alive in SGPRs that spill to VGPR lanes and come back by
// v_readlane): 5+ VALU instructions per step for one useful FMA. The frame kernel is VALU-issue-bound, while the SALU
// port is nearly idle, so these helpers narrow EXEC with three SALU instructions around the FMA instead. The lane mask is built from an
// inline constant (no SGPR pressure). EXEC is restored inside the same asm block, so the compiler never sees a changed EXEC.
// Same operation, same operands, same rounding as the predicated C++ form: the bits do not change (fast-vs-ref parity tests).
template <int B, int E, class F>
__device__ __forceinline__ void static_for(F&& f)
{
	if constexpr (B < E) { f(std::integral_constant<int, B>{}); static_for<B + 1, E>(f); }
}
template <int B, int E, class F>
__device__ __forceinline__ void static_for_down(F&& f)   // E-1, E-2, ..., B
{
	if constexpr (B < E) { f(std::integral_constant<int, E - 1>{}); static_for_down<B, E - 1>(f); }
}
// acc = fma(-a, b, acc) on lanes >= J; b is wave-uniform (SGPR pair)
template <int J>
__device__ __forceinline__ void fnma_lanes_ge(real& acc, real a, real b)
{
	unsigned long long sv, m;
	asm("s_lshl_b64 %2, -1, %5\n\ts_and_saveexec_b64 %1, %2\n\t" DTRL_VFMA " %0, -%3, %4, %0\n\ts_mov_b64 exec, %1"
	    : "+v"(acc), "=&s"(sv), "=&s"(m) : "v"(a), "s"(b), "n"(J) : "scc");
}
template <int J>
__device__ __forceinline__ void fnma_lanes_ge(real (&acc)[1], real a, const real (&b)[1]) { fnma_lanes_ge<J>(acc[0], a, b[0]); }
template <int J>
__device__ __forceinline__ void fnma_lanes_ge(real (&acc)[2], real a, const real (&b)[2])
{
	unsigned long long sv, m;
	asm("s_lshl_b64 %3, -1, %7\n\ts_and_saveexec_b64 %2, %3\n\t" DTRL_VFMA " %0, -%4, %5, %0\n\t" DTRL_VFMA " %1, -%4, %6, %1\n\ts_mov_b64 exec, %2"
	    : "+v"(acc[0]), "+v"(acc[1]), "=&s"(sv), "=&s"(m) : "v"(a), "s"(b[0]), "s"(b[1]), "n"(J) : "scc");
}
template <int J>
__device__ __forceinline__ void fnma_lanes_ge(real (&acc)[4], real a, const real (&b)[4])
{
	unsigned long long sv, m;
	asm("s_lshl_b64 %5, -1, %11\n\ts_and_saveexec_b64 %4, %5\n\t" DTRL_VFMA " %0, -%6, %7, %0\n\t" DTRL_VFMA " %1, -%6, %8, %1\n\t"
	    "" DTRL_VFMA " %2, -%6, %9, %2\n\t" DTRL_VFMA " %3, -%6, %10, %3\n\ts_mov_b64 exec, %4"
	    : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "=&s"(sv), "=&s"(m)
	    : "v"(a), "s"(b[0]), "s"(b[1]), "s"(b[2]), "s"(b[3]), "n"(J) : "scc");
}
// acc = fma(-a, b, acc) on lanes < I
template <int I>
__device__ __forceinline__ void fnma_lanes_lt(real& acc, real a, real b)
{
	unsigned long long sv, m;
	asm("s_bfm_b64 %2, %5, 0\n\ts_and_saveexec_b64 %1, %2\n\t" DTRL_VFMA " %0, -%3, %4, %0\n\ts_mov_b64 exec, %1"
	    : "+v"(acc), "=&s"(sv), "=&s"(m) : "v"(a), "s"(b), "n"(I) : "scc");
}
// dst = src on lane K only / on lanes < I
template <int K>
__device__ __forceinline__ void mov_lane_eq(real& dst, real src)
{
	unsigned long long sv, m;
	asm("s_lshl_b64 %2, 1, %4\n\ts_and_saveexec_b64 %1, %2\n\t" DTRL_VMOV " %0, %3\n\ts_mov_b64 exec, %1"
	    : "+v"(dst), "=&s"(sv), "=&s"(m) : "v"(src), "n"(K) : "scc");
}
template <int I>
__device__ __forceinline__ void mov_lanes_lt(real& dst, real src)
{
	unsigned long long sv, m;
	asm("s_bfm_b64 %2, %4, 0\n\ts_and_saveexec_b64 %1, %2\n\t" DTRL_VMOV " %0, %3\n\ts_mov_b64 exec, %1"
	    : "+v"(dst), "=&s"(sv), "=&s"(m) : "v"(src), "n"(I) : "scc");
}
// dst = src on lanes >= J
template <int J>
__device__ __forceinline__ void mov_lanes_ge(real& dst, real src)
{
	unsigned long long sv, m;
	asm("s_lshl_b64 %2, -1, %4\n\ts_and_saveexec_b64 %1, %2\n\t" DTRL_VMOV " %0, %3\n\ts_mov_b64 exec, %1"
	    : "+v"(dst), "=&s"(sv), "=&s"(m) : "v"(src), "n"(J) : "scc");
}

template <int I>
__device__ __forceinline__ void fnma_lanes_lt(real (&acc)[1], real a, const real (&b)[1]) { fnma_lanes_lt<I>(acc[0], a, b[0]); }
template <int I>
__device__ __forceinline__ void fnma_lanes_lt(real (&acc)[2], real a, const real (&b)[2])
{
	unsigned long long sv, m;
	asm("s_bfm_b64 %3, %7, 0\n\ts_and_saveexec_b64 %2, %3\n\t" DTRL_VFMA " %0, -%4, %5, %0\n\t" DTRL_VFMA " %1, -%4, %6, %1\n\ts_mov_b64 exec, %2"
	    : "+v"(acc[0]), "+v"(acc[1]), "=&s"(sv), "=&s"(m) : "v"(a), "s"(b[0]), "s"(b[1]), "n"(I) : "scc");
}
template <int I>
__device__ __forceinline__ void fnma_lanes_lt(real (&acc)[4], real a, const real (&b)[4])
{
	unsigned long long sv, m;
	asm("s_bfm_b64 %5, %11, 0\n\ts_and_saveexec_b64 %4, %5\n\t" DTRL_VFMA " %0, -%6, %7, %0\n\t" DTRL_VFMA " %1, -%6, %8, %1\n\t"
	    "" DTRL_VFMA " %2, -%6, %9, %2\n\t" DTRL_VFMA " %3, -%6, %10, %3\n\ts_mov_b64 exec, %4"
	    : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "=&s"(sv), "=&s"(m)
	    : "v"(a), "s"(b[0]), "s"(b[1]), "s"(b[2]), "s"(b[3]), "n"(I) : "scc");
}

// in-register H = U D U^T, last DoF first; returns 1/d_lane. Same elimination order and operations as factorize(), minus the
// updates that are structurally zero for the skeleton Topo (exact no-ops in the dense form). The transposed copy of U (needed by
// utsolve_regs) is produced with one pass through LDS (packed triangle in the Apk storage, which is dead between mass_row() and the
// Delassus build) instead of per-entry lane selects.
// Only the upper triangle (column >= lane) is live, so the per-entry predicate of the textbook form is dropped: the other lanes
// update entries nobody reads (they are overwritten by the transposed copy), the lanes that matter execute exactly the same
// operations. One predicated move per pivot keeps the diagonal.
template <class Topo>
__device__ __forceinline__ real factorize_regs(WSFast& ws, real (&h)[Topo::L + 2])
{
	constexpr int D = Topo::L + 2;
	const int lane = static_cast<int>(threadIdx.x);
	real dinv = 0;
	static_for_down<1, D>([&](auto kc) {             // k = D-1 ... 1
		constexpr int k = decltype(kc)::value;
		const real dk = bcast(h[k], k);
		const real ak = h[k];
		const real rk = fast_recip(dk);
		mov_lane_eq<k>(dinv, rk);                       // if (lane == k) dinv = rk
		const real lik = ak * rk;
		static_for_down<0, k>([&](auto jc) {
			constexpr int j = decltype(jc)::value;
			if constexpr (dof_coupled<Topo>(j, k)) h[j] = fmadd(-lik, bcast(ak, j), h[j]);
		});
		mov_lanes_lt<k>(h[k], lik);                     // if (lane < k) h[k] = lik
	});
	{ const real r0 = fast_recip(h[0]); if (lane == 0) dinv = r0; }
	real* S = ws.Apk;
#pragma unroll
	for (int k = 1; k < D; ++k) if (lane < k) S[k * (k - 1) / 2 + lane] = h[k];
	env_sync();
	const int base = lane * (lane - 1) / 2;
#pragma unroll
	for (int k = 0; k < D - 1; ++k) if (lane > k && lane < D) h[k] = S[base + k];
	env_sync();
	return dinv;
}
// z = U^-1 rhs (lane i holds component i): from the last DoF up
template <int D>
__device__ __forceinline__ real usolve_regs(const real (&h)[D], real z)
{
	static_for_down<1, D>([&](auto kc) {
		constexpr int k = decltype(kc)::value;
		const real zk = bcast(z, k);
		fnma_lanes_lt<k>(z, h[k], zk);                  // if (lane < k) z = fma(-h[k], zk, z)
	});
	return z;
}
// NR right-hand sides at once: the substitution is a 22-step dependent chain per right-hand side, so interleaving independent
// chains divides the exposed latency at the same instruction count (and the NR updates of a step share one EXEC window)
template <int D, int NR>
__device__ __forceinline__ void usolve_regs_n(const real (&h)[D], real (&z)[NR])
{
	static_for_down<1, D>([&](auto kc) {
		constexpr int k = decltype(kc)::value;
		real zk[NR];
#pragma unroll
		for (int j = 0; j < NR; ++j) zk[j] = bcast(z[j], k);
		fnma_lanes_lt<k>(z, h[k], zk);
	});
}
// x = U^-T u: from DoF 0 down
template <int D>
__device__ __forceinline__ real utsolve_regs(const real (&h)[D], real u)
{
	static_for<0, D - 1>([&](auto kc) {
		constexpr int k = decltype(kc)::value;
		const real uk = bcast(u, k);
		fnma_lanes_ge<k + 1>(u, h[k], uk);              // if (lane > k) u = fma(-h[k], uk, u)
	});
	return u;
}

// contact sample points in registers: lane owns points lane, lane + 64, lane + 128; wave ballots give the per-link contact
// flags and the ordered constraint-row list (same order as detect_contacts()/build_rows()) without touching LDS.
// (three named values, not an array: the struct must stay in VGPRs, never in scratch)
static_assert((kMaxPts + kGroup - 1) / kGroup == 3, "written for 3 sample points per lane");
struct ContactPts { PtVal p0, p1, p2; unsigned long long m0, m1, m2; /* constraint-carrying points */ unsigned long long f0, f1, f2; /* points near the surface: contact flags */ };
// lane id the optimiser cannot see through: address / index arithmetic derived from it is recomputed where it is used instead of being
// hoisted out of the 4000-instruction physics loop as a loop invariant and spilled to scratch across it
__device__ __forceinline__ int opaque_lane()
{
	int lane = static_cast<int>(threadIdx.x);
	asm volatile("" : "+v"(lane));
	return lane;
}
// the kPtsPerLink ballot bits of link `link`'s sample points (points link * 6 .. link * 6 + 5 of the three 64-point ballots)
__device__ __forceinline__ unsigned link_field(unsigned long long m0, unsigned long long m1, unsigned long long m2, int link)
{
	const int b = link * kPtsPerLink, word = b >> 6, off = b & 63;
	unsigned long long lo = m0, hi = m1;   // (plain selects on by-value scalars: an indexable aggregate here would be demoted to scratch)
	if (word == 1) { lo = m1; hi = m2; }
	if (word == 2) { lo = m2; hi = 0ull; }
	unsigned long long bits = lo >> off;
	if (off + kPtsPerLink > 64) bits |= hi << (64 - off);
	return static_cast<unsigned>(bits & ((1ull << kPtsPerLink) - 1ull));
}
// at most kMaxPtsPerLink constraint-carrying points per link, the deepest ones (ties: lower sample-point index). Cold path (a box lying in the
// ground with five or six of its sample points below the surface): the depths go once through LDS (the packed-matrix storage is dead between
// the factorisation and the Delassus build, and before the controller's mass rows) so that a point sees the other sample points of its link
// Takes and returns scalars only (an aggregate passed by reference to a non-inlined function would live in scratch memory on the hot path):
// d0..d2 = depth of the lane's three sample points or kNoPoint when the point carries no rows (a row-carrying point within the breaking threshold above
// the surface has a small negative depth); returns bit k set when point k is dropped
constexpr real kNoPoint = -1.0e30;
__device__ __noinline__ unsigned link_cap_drop_mask(WSFast& ws, real d0, real d1, real d2)
{
	const int lane = opaque_lane();
	const int npts = ws.M.L * kPtsPerLink;
	real* S = ws.Apk;
	if (lane < npts) S[lane] = d0;
	if (lane + kGroup < npts) S[lane + kGroup] = d1;
	if (lane + 2 * kGroup < npts) S[lane + 2 * kGroup] = d2;
	env_sync();
	auto over = [&](real d, int pt) -> unsigned {
		if (!(d > kNoPoint)) return 0u;
		const int base = (pt / kPtsPerLink) * kPtsPerLink;
		int rank = 0;
		for (int k = 0; k < kPtsPerLink; ++k) { const int o = base + k; const real od = S[o]; rank += (o != pt && (od > d || (od == d && o < pt))) ? 1 : 0; }
		return rank >= kMaxPtsPerLink ? 1u : 0u;
	};
	const unsigned m = over(d0, lane) | (over(d1, lane + kGroup) << 1) | (over(d2, lane + 2 * kGroup) << 2);
	env_sync();
	return m;
}
// kFlags: also find the points within contact_tol of the surface (the per-link contact flags the controller reads: only the post-step pass needs them)
template <bool kFlags>
__device__ __forceinline__ ContactPts eval_points(WSFast& ws, const DevModel& gm, const GroundRec& g)
{
	const int lane = opaque_lane();
	const int npts = ws.M.L * kPtsPerLink;
	ContactPts c;
	PtVal z; z.x = 0; z.y = 0; z.depth = 0; z.nx = 0; z.ny = 0; z.active = 0; z.near = 0;
	c.p0 = z; c.p1 = z; c.p2 = z;
	if (lane < npts) c.p0 = contact_point_eval<kFlags>(ws, gm, g, lane);
	if (lane + kGroup < npts) c.p1 = contact_point_eval<kFlags>(ws, gm, g, lane + kGroup);
	if (lane + 2 * kGroup < npts) c.p2 = contact_point_eval<kFlags>(ws, gm, g, lane + 2 * kGroup);
	c.f0 = 0; c.f1 = 0; c.f2 = 0;
	if (kFlags) { c.f0 = __ballot(c.p0.near); c.f1 = __ballot(c.p1.near); c.f2 = __ballot(c.p2.near); }
	c.m0 = __ballot(c.p0.active); c.m1 = __ballot(c.p1.active); c.m2 = __ballot(c.p2.active);
	if ((c.m0 | c.m1 | c.m2) != 0ull) {   // wave-uniform: somebody penetrates; does any link have more than kMaxPtsPerLink such points?
		const int many = (lane < ws.M.L) && __popc(link_field(c.m0, c.m1, c.m2, lane)) > kMaxPtsPerLink;
		if (__builtin_expect(__ballot(many) != 0ull, 0)) {
			const unsigned drop = link_cap_drop_mask(ws, c.p0.active ? c.p0.depth : kNoPoint, c.p1.active ? c.p1.depth : kNoPoint, c.p2.active ? c.p2.depth : kNoPoint);
			if (drop & 1u) c.p0.active = 0;
			if (drop & 2u) c.p1.active = 0;
			if (drop & 4u) c.p2.active = 0;
			c.m0 = __ballot(c.p0.active); c.m1 = __ballot(c.p1.active); c.m2 = __ballot(c.p2.active);
		}
	}
	return c;
}
// more penetrating points than constraint rows: the deepest `cap` points overall keep theirs (ties: lower sample-point index). Cold path
// (a character lying on the ground with a dozen points in it)
__device__ __noinline__ unsigned row_cap_drop_mask(WSFast& ws, real d0, real d1, real d2, int cap)
{
	const int lane = opaque_lane();
	const int npts = ws.M.L * kPtsPerLink;
	real* S = ws.Apk;
	if (lane < npts) S[lane] = d0;
	if (lane + kGroup < npts) S[lane + kGroup] = d1;
	if (lane + 2 * kGroup < npts) S[lane + 2 * kGroup] = d2;
	env_sync();
	auto over = [&](real d, int pt) -> unsigned {
		if (!(d > kNoPoint)) return 0u;
		int rank = 0;
		for (int o = 0; o < npts; ++o) { const real od = S[o]; rank += (o != pt && (od > d || (od == d && o < pt))) ? 1 : 0; }
		return rank >= cap ? 1u : 0u;
	};
	const unsigned m = over(d0, lane) | (over(d1, lane + kGroup) << 1) | (over(d2, lane + 2 * kGroup) << 2);
	env_sync();
	return m;
}
// ---- link--link contacts on the wave: one lane per pair decides whether the two boxes can touch at all (hot path: a dozen instructions, almost always
// "no"); the pairs in reach are then taken one at a time, twelve lanes evaluating the twelve candidates (cold path, out of line: nothing of it may
// occupy registers in the physics loop). Same candidates, ranking and row order as append_pair_rows_serial().
__device__ __forceinline__ unsigned long long pairs_in_reach(const WSFast& ws)
{
	const int lane = static_cast<int>(threadIdx.x);
	return __ballot(lane < ws.M.n_cpairs && pair_in_reach(ws, lane));
}
// twelve candidates per pair, five pairs per pass (60 of the 64 lanes): the table reads of a pass travel together, the lane order (pair ascending, candidate
// ascending) is the serial order, so the ballots give the same row positions and the same cut when the row budget runs out.
// When a pair has more than kMaxPtsPerPair penetrating candidates the depth ranking (two barriers, an LDS round trip) only runs when one does.
// (Link--link contacts carry constraint rows but never set a link's contact flag: scenarios/ScenarioSimChar.cpp:321, sim/ContactManager.cpp:169-175.)
constexpr int kPairCands = 2 * kPtsPerLink, kPairSlots = kGroup / kPairCands;
__device__ __noinline__ int append_pair_rows_fast(WSFast& ws, const DevModel& gm, int R, unsigned long long reach)
{
	const int lane = opaque_lane();
	const unsigned long long below = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
	const int slot = lane / kPairCands, cand = lane - slot * kPairCands;
	while (reach != 0ull && R + 2 <= kMaxRows) {
		int mypr = -1;
#pragma unroll
		for (int sidx = 0; sidx < kPairSlots; ++sidx) {
			if (reach != 0ull) { const int pr = __ffsll(static_cast<long long>(reach)) - 1; reach &= reach - 1ull; if (slot == sidx) mypr = pr; }
		}
		int P = 0, Q = 0, k = 0;
		PairHit hit; hit.active = 0; hit.depth = 0; hit.x = 0; hit.y = 0; hit.nx = 0; hit.ny = 0;
		if (mypr >= 0) { pair_candidate(ws.M.cp_a[mypr], ws.M.cp_b[mypr], cand, &P, &Q, &k); hit = pair_point_eval(ws, gm, P, Q, k); }
		const unsigned long long am = __ballot(hit.active);
		if (am == 0ull) continue;
		int keep = hit.active;
		const int mine_cnt = __popcll((am >> (slot * kPairCands)) & ((1ull << kPairCands) - 1ull));
		if (__builtin_expect(__ballot(lane < kPairSlots * kPairCands && mine_cnt > kMaxPtsPerPair) != 0ull, 0)) {
			real* S = ws.Apk;   // (dead between the factorisation and the Delassus build, like in link_cap_drop_mask())
			S[lane] = hit.active ? hit.depth : -1.0;
			env_sync();
			if (hit.active) {
				int rank = 0;
				for (int o = 0; o < kPairCands; ++o) { const real od = S[slot * kPairCands + o]; rank += (o != cand && od > 0 && (od > hit.depth || (od == hit.depth && o < cand))) ? 1 : 0; }
				keep = rank < kMaxPtsPerPair;
			}
			env_sync();
		}
		const unsigned long long km = __ballot(keep);
		const int room = (kMaxRows - R) / 2;
		const int idx = __popcll(km & below);
		if (keep && idx < room) {
			const int Rw = R + 2 * idx;
			ws.row_kind[Rw] = 1; ws.row_link[Rw] = P; ws.row_link2[Rw] = static_cast<int8_t>(Q); ws.row_x[Rw] = hit.x; ws.row_y[Rw] = hit.y;
			ws.row_dx[Rw] = hit.nx; ws.row_dy[Rw] = hit.ny; ws.row_tgt[Rw] = 0;   // velocity-level only (append_pair_rows_serial)
			ws.row_id[Rw] = pair_row_id(mypr, cand, 0); ws.row_id[Rw + 1] = pair_row_id(mypr, cand, 1);
			ws.row_kind[Rw + 1] = 2; ws.row_link[Rw + 1] = P; ws.row_link2[Rw + 1] = static_cast<int8_t>(Q); ws.row_x[Rw + 1] = hit.x; ws.row_y[Rw + 1] = hit.y;
			ws.row_dx[Rw + 1] = hit.ny; ws.row_dy[Rw + 1] = -hit.nx; ws.row_tgt[Rw + 1] = 0;
		}
		int n = __popcll(km); if (n > room) n = room;
		R += 2 * n;
	}
	return R;
}
__device__ __forceinline__ void contact_bits_fast(WSFast& ws, unsigned long long m0, unsigned long long m1, unsigned long long m2)
{
	const int lane = static_cast<int>(threadIdx.x);
	int any = 0;
	if (lane < ws.M.L) any = link_field(m0, m1, m2, lane) != 0u;
	const unsigned long long lm = __ballot(any);
	if (lane == 0) ws.st.contact_bits = static_cast<uint32_t>(lm);
}
__device__ __forceinline__ void detect_contacts_fast(WSFast& ws, const DevModel& gm, const GroundRec& g)
{
	const ContactPts c = eval_points<true>(ws, gm, g);
	contact_bits_fast(ws, c.f0, c.f1, c.f2);
	env_sync();
}
__device__ __forceinline__ void emit_contact_rows(WSFast& ws, const PtVal& p, int pt, int rank, int cap, int R0, real inv_h)
{
	if (p.active && rank < cap) {
		const int R = R0 + 2 * rank;
		const int j = pt / kPtsPerLink;
		ws.row_kind[R] = 1; ws.row_link[R] = j; ws.row_link2[R] = -1; ws.row_x[R] = p.x; ws.row_y[R] = p.y;
		ws.row_dx[R] = p.nx; ws.row_dy[R] = p.ny; ws.row_tgt[R] = normal_row_target(p.depth, inv_h);
		ws.row_id[R] = ground_row_id(pt, 0); ws.row_id[R + 1] = ground_row_id(pt, 1);
		ws.row_kind[R + 1] = 2; ws.row_link[R + 1] = j; ws.row_link2[R + 1] = -1; ws.row_x[R + 1] = p.x; ws.row_y[R + 1] = p.y;
		ws.row_dx[R + 1] = p.ny; ws.row_dy[R + 1] = -p.nx; ws.row_tgt[R + 1] = 0;
	}
}
__device__ __forceinline__ void build_rows_fast(WSFast& ws, const DevModel& gm, const ContactPts& c_in, real h)
{
	ContactPts c = c_in;
	const int lane = static_cast<int>(threadIdx.x);
	const unsigned long long below = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
	const real inv_h = 1.0 / h;
	// joint limits, ordered by joint id
	int lim = 0; real tgt = 0;
	if (lane >= 1 && lane < ws.M.L && !(ws.M.lim_lo[lane] > ws.M.lim_hi[lane])) {
		const real th = ws.st.q[lane + 2];
		if (th <= ws.M.lim_lo[lane] + kLimitSlop) { lim = 1; tgt = kLimitErp * fmax(ws.M.lim_lo[lane] - th, 0.0) * inv_h; }
		else if (th >= ws.M.lim_hi[lane] - kLimitSlop) { lim = -1; tgt = kLimitErp * fmax(th - ws.M.lim_hi[lane], 0.0) * inv_h; }
	}
	const unsigned long long ml = __ballot(lim != 0);
	const int rl = __popcll(ml & below);
	int R0 = __popcll(ml); if (R0 > kMaxRows) R0 = kMaxRows;
	if (lim != 0 && rl < kMaxRows) { ws.row_kind[rl] = 0; ws.row_link[rl] = lane; ws.row_link2[rl] = -1; ws.row_dx[rl] = lim; ws.row_tgt[rl] = tgt; ws.row_id[rl] = kNoRowId; }
	// contacts, ordered by sample-point index
	const int cap = (kMaxRows - R0) / 2;
	if (__builtin_expect(__popcll(c.m0) + __popcll(c.m1) + __popcll(c.m2) > cap, 0)) {   // wave-uniform
		const unsigned drop = row_cap_drop_mask(ws, c.p0.active ? c.p0.depth : kNoPoint, c.p1.active ? c.p1.depth : kNoPoint, c.p2.active ? c.p2.depth : kNoPoint, cap);
		if (drop & 1u) c.p0.active = 0;
		if (drop & 2u) c.p1.active = 0;
		if (drop & 4u) c.p2.active = 0;
		c.m0 = __ballot(c.p0.active); c.m1 = __ballot(c.p1.active); c.m2 = __ballot(c.p2.active);
	}
	const int n0 = __popcll(c.m0), n1 = __popcll(c.m1), n2 = __popcll(c.m2);
	emit_contact_rows(ws, c.p0, lane, __popcll(c.m0 & below), cap, R0, inv_h);
	emit_contact_rows(ws, c.p1, lane + kGroup, n0 + __popcll(c.m1 & below), cap, R0, inv_h);
	emit_contact_rows(ws, c.p2, lane + 2 * kGroup, n0 + n1 + __popcll(c.m2 & below), cap, R0, inv_h);
	int nc = n0 + n1 + n2; if (nc > cap) nc = cap;
	int R = R0 + 2 * nc;
	// link--link contacts take what is left of the row budget
	const unsigned long long reach = pairs_in_reach(ws);
	if (__builtin_expect(reach != 0ull, 0)) R = append_pair_rows_fast(ws, gm, R, reach);
	if (lane == 0) ws.R = R;
	env_sync();
}

// Delassus matrix A = Z D^-1 Z^T, entry-parallel: the R (R + 1) / 2 lower-triangle entries are dealt round-robin to all 64
// lanes (entry e = packed index), so a 16-row system is 3 passes of 23-long dot products instead of 16; the rows of Z are
// read from LDS (odd row stride: lanes on different rows hit different banks, lanes on the same row broadcast).
// Per-entry operation order is the one of build_delassus(), so the bits are the same.
template <int D>
__device__ __forceinline__ void build_delassus_fast(WSFast& ws, real h, real dinv_mine)
{
	const int lane = static_cast<int>(threadIdx.x);
	const int R = ws.R;
	const int n_ent = R * (R + 1) / 2;
	if (R < 16) {
		// up to 15 rows (all but 0.2 % of the substeps): [A | zz] = (Z D^-1) Z^T for the R constraint rows plus the free right-hand side as
		// column R is ONE 16x16 tile of the fp64 matrix pipe, six k-steps of v_mfma_f64_16x16x4_f64 (k = DoF, padded to 24). The MFMA
		// accumulates in k order with fused multiply-adds, i.e. the sequence of the scalar loops (tools/microbench/mfma_f64_check.hip).
		// Operand layout: A[i = lane % 16][k = lane / 16], B[k = lane / 16][n = lane % 16], D register r = D[4 r + lane / 16][lane % 16].
		typedef v4r_t v4d_t;
		const int g = lane >> 4, c = lane & 15;
		// row residual's velocity term, on the lanes that finish wv below (independent of the product: overlaps it)
		real jv = 0;
		if (lane < R) {
			const int sw = lane;
			if (ws.row_kind[sw] == 0) jv = ws.row_dx[sw] * ws.st.qd[ws.row_link[sw] + 2];
			else jv = row_point_jv(ws, sw);
		}
		const int zrow = c <= R ? c : 0;                        // rows above R are never stored; any finite-or-not value will do
		v4d_t acc = {0, 0, 0, 0};
#pragma unroll
		for (int s4 = 0; s4 < (D + 3) / 4; ++s4) {
			const int k = 4 * s4 + g;
			const bool live = k < D;
			const real z = ws.Z[zrow][live ? k : 0], dk = ws.dinv[live ? k : 0];
			const real b = live ? z : 0.0;
			const real a = live ? z * dk : 0.0;
			acc = mfma_16x16x4(a, b, acc);
		}
#pragma unroll
		for (int r = 0; r < 4; ++r) {
			const int i = mfma_row(r, g);                          // D[i][c]
			if (i < R && c <= i) ws.Apk[i * (i + 1) / 2 + c] = acc[r];
			if (i < R && c == R) ws.wv[i] = acc[r];               // zz_i, finished below
		}
		env_sync();
		if (lane < R) ws.wv[lane] = jv + h * ws.wv[lane] - ws.row_tgt[lane];
		env_sync();
		return;
	}
	const real* di = ws.dinv;   // general path (16+ rows, 0.2 % of the substeps): entry-parallel dot products
	auto entry_rows = [](int e, int& s, int& r) {
		s = static_cast<int>((sqrtf(8.0f * e + 1.0f) - 1.0f) * 0.5f);
		while (s * (s + 1) / 2 > e) --s;
		while ((s + 1) * (s + 2) / 2 <= e) ++s;
		r = e - s * (s + 1) / 2;
	};
	// first pass: the lane's matrix entry (all of them for R <= 10) and, on lanes < R, the row's initial residual w = J v_free - target.
	// The two 23-term dot products are independent dependent chains; evaluated in one loop they cost the latency of one
	{
		const bool has_e = lane < n_ent, has_w = lane < R;
		int s = 0, r = 0;
		if (has_e) entry_rows(lane, s, r);
		const int sw = has_w ? lane : 0;
		real jv = 0;
		if (has_w) {
			if (ws.row_kind[sw] == 0) jv = ws.row_dx[sw] * ws.st.qd[ws.row_link[sw] + 2];
			else jv = row_point_jv(ws, sw);
		}
		const real* zs = ws.Z[s];
		const real* zr = ws.Z[r];
		const real* zw = ws.Z[sw];
		const real* z0 = ws.Z[R];
		real a = 0, zz = 0;
#pragma unroll 4
		for (int i = 0; i < D; ++i) { a = fmadd(zs[i] * di[i], zr[i], a); zz = fmadd(zw[i] * di[i], z0[i], zz); }
		if (has_e) ws.Apk[lane] = a;
		if (has_w) ws.wv[lane] = jv + h * zz - ws.row_tgt[lane];
	}
	for (int e = lane + kGroup; e < n_ent; e += kGroup) {
		int s, r; entry_rows(e, s, r);
		const real* zs = ws.Z[s];
		const real* zr = ws.Z[r];
		real a = 0;
#pragma unroll 4
		for (int i = 0; i < D; ++i) a = fmadd(zs[i] * di[i], zr[i], a);
		ws.Apk[e] = a;
	}
	env_sync();
}

// projected Gauss-Seidel in lambda space with the row state in registers: lane s owns row s (w_s, lambda_s, 1/A_ss, kind).
// The solve is one long dependent chain (10 sweeps x R row updates): the number of DEPENDENT instructions per update matters -- and, with two waves per SIMD, so does the
// instruction count (round 5: the 27-instruction generic step was issue-bound; the per-pass steps below are 9-11). Every lane evaluates its own candidate update each step (lane r's is the one that
// counts): the projection bounds [lo, hi] (normal rows [0, inf), tangent rows +-mu * lambda of the row before it, fetched with
// a wave_shr:1 DPP move) are off the chain, leaving w -> mul -> sub -> max -> min -> sub -> readlane -> mul -> add.
// Same operations on the same operands as pgs_solve(), hence the same bits; rows with a vanishing effective mass are skipped.
__device__ __forceinline__ real wave_shr1(real v)
{
#if defined(DTRL_REAL_F32)
	return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x138, 0xf, 0xf, false));   // DPP wave_shr:1
#else
	const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x138, 0xf, 0xf, false);   // DPP wave_shr:1
	const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x138, 0xf, 0xf, false);
	return __hiloint2double(hi, lo);
#endif
}
template <int r, int K>
__device__ __forceinline__ void pgs_rows_load(real (&a)[K], const WSFast& ws, int lane, bool mine, int R)
{
	if constexpr (r < K) {
		if (r < R) {
			const int mx = lane > r ? lane : r, mn = lane < r ? lane : r;
			a[r] = ws.Apk[mine ? mx * (mx + 1) / 2 + mn : 0];
			pgs_rows_load<r + 1, K>(a, ws, lane, mine, R);
		}
	}
}
// one sweep over rows r, r + 1, ...: rows below K read the lane's Delassus entry from registers, rows K .. kMaxRows - 1 (the tail only characters lying on the
// ground reach) from LDS, fetched one row update ahead (a_pref) so that the load's latency sits under the previous row's dependent chain
// The generic row step of rounds 1-4: ONE interleaved sweep over all rows, the bounds of row r evaluated per step (normal rows [0, inf), tangent rows +-mu * lambda of the
// row before it). Only -warm_start= 0 (the round-4 contact model, an ablation) still runs it; the shipped model's sweep is the two passes below.
template <int r, int K, int kEnd>
__device__ __forceinline__ void pgs_rows_sweep(const real (&a)[K], const WSFast& ws, real& w, real& lam, real rinv, bool tang, bool mine, unsigned long long actR, int lane, int R, real a_pref)
{
	if constexpr (r < kEnd) {
		if (r < R) {
			real a_sr;
			if constexpr (r < K) a_sr = a[r]; else a_sr = a_pref;
			real a_nx = 0.0;
			if constexpr (r + 1 >= K && r + 1 < kEnd) {
				constexpr int rn = r + 1;
				const int mx = lane > rn ? lane : rn, mn = lane < rn ? lane : rn;
				a_nx = ws.Apk[(mine && rn < R) ? mx * (mx + 1) / 2 + mn : 0];
			}
			if ((actR >> r) & 1ull) {
				// (a wave-uniform branch on the kind of row r -- normal rows need neither the friction bound nor the upper clamp -- was measured twice, round 1 and
				// round 4 under the ILP scheduler: -6 % / -3 %. The selects are cheaper than the branch.)
				const real lim = kMu * wave_shr1(lam);
				const real lo = tang ? -lim : 0.0, hi = tang ? lim : __builtin_huge_val();
				const real nl = fmin(fmax(fmadd(-w, rinv, lam), lo), hi);
				const real dl = bcast(nl - lam, r);
				if (lane == r) lam = nl;
				w = fmadd(a_sr, dl, w);
			}
			pgs_rows_sweep<r + 1, K, kEnd>(a, ws, w, lam, rinv, tang, mine, actR, lane, R, a_nx);
		}
	}
}
// The two passes of a sweep under Bullet's contact persistence as their own row steps (round 5). kFric = false: limit and normal rows, projection onto [0, inf) -- fma, max,
// sub, readlane, fma. kFric = true: the friction rows that are not held; their bounds +-mu lambda_n are LANE REGISTERS set once per pass (every normal row is final for
// the sweep by then) -- fma, max, min, sub, readlane, fma. Same operations on the same operands as the generic step above, without its per-step DPP move, multiply,
// compare and selects.
template <bool kFric, int r, int K, int kEnd>
__device__ __forceinline__ void pgs_rows_pass(const real (&a)[K], const WSFast& ws, real& w, real& lam, real rinv, real lo, real hi, bool mine, unsigned long long rows, int lane, int R, real a_pref)
{
	if constexpr (r < kEnd) {
		if (r < R) {
			real a_sr;
			if constexpr (r < K) a_sr = a[r]; else a_sr = a_pref;
			real a_nx = 0.0;
			if constexpr (r + 1 >= K && r + 1 < kEnd) {
				constexpr int rn = r + 1;
				const int mx = lane > rn ? lane : rn, mn = lane < rn ? lane : rn;
				a_nx = ws.Apk[(mine && rn < R) ? mx * (mx + 1) / 2 + mn : 0];
			}
			if ((rows >> r) & 1ull) {
				real nl = fmadd(-w, rinv, lam);
				if constexpr (kFric) nl = fmin(fmax(nl, lo), hi); else nl = fmax(nl, 0.0);   // (pgs_solve(): normal and limit rows are projected onto [0, inf))
				const real dl = bcast(nl - lam, r);
				if (lane == r) lam = nl;
				w = fmadd(a_sr, dl, w);
			}
			pgs_rows_pass<kFric, r + 1, K, kEnd>(a, ws, w, lam, rinv, lo, hi, mine, rows, lane, R, a_nx);
		}
	}
}
// w += A lambda_0 (the warm-started impulses), column by column in row order: the same fused operations as pgs_solve()'s first loop
template <int r, int K, int kEnd>
__device__ __forceinline__ void pgs_rows_warm(const real (&a)[K], const WSFast& ws, real& w, real lam, bool mine, unsigned long long nz, int lane, int R)
{
	if constexpr (r < kEnd) {
		if (r < R) {
			if ((nz >> r) & 1ull) {   // (rows that start from zero -- limit rows, link--link rows, new contacts -- add exactly nothing)
				real a_sr;
				if constexpr (r < K) a_sr = a[r];
				else { const int mx = lane > r ? lane : r, mn = lane < r ? lane : r; a_sr = ws.Apk[mine ? mx * (mx + 1) / 2 + mn : 0]; }
				w = fmadd(a_sr, bcast(lam, r), w);
			}
			pgs_rows_warm<r + 1, K, kEnd>(a, ws, w, lam, mine, nz, lane, R);
		}
	}
}
// Bullet's persistent contact points on the wave (warm_match() of dtrl_kernel.h): lane r looks the identity of fresh row r up among the rows of the last solved
// substep (<= 24 broadcast reads of 16-bit ids) and starts from kWarmFactor x that row's impulse; the fresh list then becomes the cache
__device__ __forceinline__ void warm_match_fast(WSFast& ws)
{
	const int lane = opaque_lane();
	const int R = ws.R, Rp = ws.st.ws_R;   // wave-uniform
	if (R == 0) { if (Rp != 0 && lane == 0) ws.st.ws_R = 0; return; }   // (a fifth of the substeps: airborne)
	int id = kNoRowId;
	if (lane < R) id = ws.row_id[lane];
	real l0 = 0.0;
	if (ws.M.warm_start != 0 && Rp > 0) {
		int hit = -1;
		for (int p = 0; p < Rp; ++p) hit = (ws.st.ws_id[p] == id) ? p : hit;
		if (id < kFirstPairRowId && hit >= 0) l0 = kWarmFactor * ws.st.ws_lam[hit];   // ground contact rows only
	}
	// (no barrier between the reads above and the writes below: one wavefront, and its LDS operations complete in issue order)
	if (lane < R) { ws.st.ws_lam[lane] = l0; ws.st.ws_id[lane] = static_cast<uint16_t>(id); }
	if (lane == 0) ws.st.ws_R = R;
	// (no fence behind the writes either: their next reader is the Gauss-Seidel solve, several phase boundaries further on)
}
template <int kPgsRegRows, bool kTailInSweep>
__device__ __forceinline__ void pgs_solve_fast(WSFast& ws)
{
	const int lane = opaque_lane();
	const int R = ws.R;
	const bool mine = lane < R;
	real w = mine ? ws.wv[lane] : 0.0;
	real rinv = 0.0;
	if (mine) { const real ass = ws.Apk[lane * (lane + 3) / 2]; rinv = (ass >= 1e-12) ? 1.0 / ass : 0.0; }
	real lam = (mine && rinv != 0.0) ? ws.st.ws_lam[lane] : 0.0;   // warm_match_fast(): kWarmFactor x the row's previous impulse; rows with a vanishing effective mass stay at zero
	const bool tang = mine && ws.row_kind[lane] == 2;
	const unsigned long long act = __ballot(rinv != 0.0);
	// Bullet's contact persistence: a sweep resolves the limit and normal rows (list order), then the friction rows; a friction row only under a loaded normal row
	const bool warm = ws.M.warm_start != 0;   // wave-uniform
	const unsigned long long tmask = __ballot(tang);
	const unsigned long long nz_l0 = __ballot(lam != 0.0);
	const bool any_l0 = nz_l0 != 0ull;
	// The lane's Delassus row: the first kPgsRegRows entries (per skeleton, dtrl_topo.h; eight until round 3, twelve in round 3) live in registers for all sweeps
	// -- no LDS read and no packed-index arithmetic per row update -- and the entries of the tail rows come from LDS one update ahead. Same operations on the
	// same values in the same order as pgs_solve() of dtrl_kernel.h. The rare substeps with many rows matter out of proportion: they are what the slowest envs of
	// a launch do in EVERY substep (a character lying on the ground), and a launch lasts as long as its slowest env.
	// The row sequences are NESTED (row r + 1 sits inside `if (r < R)` of row r), so a substep leaves them at its first absent row with one scalar compare + branch
	// per row that exists: tested row by row, the 24 - R absent rows cost 4 dependent SALU instructions each, ten sweeps per substep -- +4 k cycles on the typical
	// substep with 3 rows (round 4).
	// Per skeleton (dtrl_topo.h, same-box A/B in profiles/r04_pgs_rows_ab.txt): the dog's instance keeps 20 rows in registers and runs the tail rows inside the same
	// unrolled sweep (kTailInSweep); the raptor's pays for that longer sweep body with spills, keeps 16 and sends a substep with more rows through the plain loop below.
	if (kTailInSweep || R <= kPgsRegRows) {
		real a[kPgsRegRows];
		pgs_rows_load<0, kPgsRegRows>(a, ws, lane, mine, R);
		const unsigned long long actR = act & ((R < 64) ? ((1ull << R) - 1ull) : ~0ull);
		constexpr int kEnd = kTailInSweep ? kMaxRows : kPgsRegRows;
		if (any_l0) pgs_rows_warm<0, kPgsRegRows, kEnd>(a, ws, w, lam, mine, nz_l0, lane, R);
		const unsigned long long pass0 = warm ? (actR & ~tmask) : actR, pass1 = actR & tmask;
		if (warm) {
			// a pass leaves the nested row sequence behind its LAST row (not at R): the rows of the other pass that follow it are not even looked at
			// (with tail rows in the sweep the row count also guards the LDS prefetch of the next column: that instance keeps R)
			constexpr bool kPrefetch = kTailInSweep && kPgsRegRows < kMaxRows;
			const int end0 = (kPrefetch || pass0 == 0ull) ? R : 64 - __builtin_clzll(pass0);
#pragma unroll 1
			for (int it = 0; it < kPgsIters; ++it) {
				if (pass0 != 0ull) pgs_rows_pass<false, 0, kPgsRegRows, kEnd>(a, ws, w, lam, rinv, 0.0, 0.0, mine, pass0, lane, end0, 0.0);
				// friction pass: every normal row is final for this sweep, so the bounds +-mu lambda_n are fixed for the pass, and which friction rows Bullet's rule HOLDS
				// (normal row without impulse: their update is exactly zero) is known up front -- they leave the pass instead of walking through a row step each
				const real ln = wave_shr1(lam);   // (all lanes: a DPP move under a narrowed EXEC does not see the lanes that are switched off)
				const real lim = kMu * ln;
				const unsigned long long rows = pass1 & __ballot(tang && ln > kHoldEps);
				if (rows != 0ull) pgs_rows_pass<true, 0, kPgsRegRows, kEnd>(a, ws, w, lam, rinv, -lim, lim, mine, rows, lane, kPrefetch ? R : 64 - __builtin_clzll(rows), 0.0);
			}
		} else {
#pragma unroll 1
			for (int it = 0; it < kPgsIters; ++it) pgs_rows_sweep<0, kPgsRegRows, kEnd>(a, ws, w, lam, rinv, tang, mine, actR, lane, R, 0.0);   // (-warm_start= 0: one interleaved sweep, rounds 1-4)
		}
		if (mine) ws.st.ws_lam[lane] = lam;
		env_sync();
		return;
	}
	const int tri = lane * (lane + 1) / 2;
	const real inf = __builtin_huge_val();
	if (any_l0) for (int r = 0; r < R; ++r) if ((nz_l0 >> r) & 1ull) { const int mx = lane > r ? lane : r, mn = lane < r ? lane : r; w = fmadd(ws.Apk[mine ? mx * (mx + 1) / 2 + mn : 0], bcast(lam, r), w); }
	real a_nx = mine ? ws.Apk[tri] : 0.0;   // column 0; the column of the next row update is fetched one update ahead
	const unsigned long long actR = act & ((R < 64) ? ((1ull << R) - 1ull) : ~0ull);
	if (warm) {
		// the two passes with their own row steps, as in the unrolled form above (pgs_rows_pass)
		const unsigned long long pass0 = actR & ~tmask, pass1 = actR & tmask;
		for (int it = 0; it < kPgsIters; ++it) {
			for (int r = 0; r < R; ++r) {
				const real a_sr = a_nx;
				const int rn = (r + 1 < R) ? r + 1 : 0;
				{ const int mx = lane > rn ? lane : rn, mn = lane < rn ? lane : rn; a_nx = ws.Apk[mine ? mx * (mx + 1) / 2 + mn : 0]; }   // branch-free packed index
				if (!((pass0 >> r) & 1ull)) continue;
				const real nl = fmax(fmadd(-w, rinv, lam), 0.0);
				const real dl = bcast(nl - lam, r);
				if (lane == r) lam = nl;
				w = fmadd(a_sr, dl, w);
			}
			const real ln = wave_shr1(lam);
			const real lim = kMu * ln;
			const unsigned long long rows = pass1 & __ballot(tang && ln > kHoldEps);
			for (int r = 0; r < R; ++r) {
				const real a_sr = a_nx;
				const int rn = (r + 1 < R) ? r + 1 : 0;
				{ const int mx = lane > rn ? lane : rn, mn = lane < rn ? lane : rn; a_nx = ws.Apk[mine ? mx * (mx + 1) / 2 + mn : 0]; }
				if (!((rows >> r) & 1ull)) continue;
				const real nl = fmin(fmax(fmadd(-w, rinv, lam), -lim), lim);
				const real dl = bcast(nl - lam, r);
				if (lane == r) lam = nl;
				w = fmadd(a_sr, dl, w);
			}
		}
	} else {
		for (int it = 0; it < kPgsIters; ++it) {
			for (int r = 0; r < R; ++r) {   // (-warm_start= 0: one interleaved sweep, rounds 1-4)
				const real a_sr = a_nx;
				const int rn = (r + 1 < R) ? r + 1 : 0;
				{ const int mx = lane > rn ? lane : rn, mn = lane < rn ? lane : rn; a_nx = ws.Apk[mine ? mx * (mx + 1) / 2 + mn : 0]; }   // branch-free packed index
				if (!((act >> r) & 1ull)) continue;
				const real lim = kMu * wave_shr1(lam);
				const real lo = tang ? -lim : 0.0, hi = tang ? lim : inf;
				const real nl = fmin(fmax(fmadd(-w, rinv, lam), lo), hi);
				const real dl = bcast(nl - lam, r);
				if (lane == r) lam = nl;
				w = fmadd(a_sr, dl, w);
			}
		}
	}
	if (mine) ws.st.ws_lam[lane] = lam;
	env_sync();
}

template <class Topo>
struct FastPath {
	static constexpr int D = Topo::L + 2;
	static __device__ void substep(WSFast& ws, const DevModel& gm, const GroundRec& g, real h, bool kin_valid)
	{
		const int lane = static_cast<int>(threadIdx.x);
#if defined(DTRL_PROFILE)
		const unsigned long long prof_sub_t0 = __builtin_readcyclecounter();
#endif
		if (!kin_valid) { PROF_T0(); kin_dyn_terms(ws); PROF_ADD(ws, kProfFK); }
		// constraint rows first, factorisation second: the two are independent (both read the kinematics only), and in this order neither the
		// 23 doubles of the lane's matrix row are live across the contact pass (and its out-of-line calls) nor the sample points across the elimination.
		// The post-step contact pass of the previous env-step (contacts() below) ran at this very configuration and left its constraint
		// row list in LDS: the first substep of an env-step takes it over instead of sampling the heightfield again
		const int rows_ready = kin_valid ? ws.n_pts_active : -1;   // wave-uniform
		if (rows_ready >= 0) {
			if (lane == 0) { ws.R = rows_ready; ws.n_pts_active = -1; }
			env_sync();
		} else {
			ContactPts cp;
			{ PROF_T0(); cp = eval_points<false>(ws, gm, g); PROF_ADD(ws, kProfDetect); }   // the per-link contact flags are the post-step pass's business (contacts() below)
			{ PROF_T0(); build_rows_fast(ws, gm, cp, h); PROF_ADD(ws, kProfRows); }
		}
		{ PROF_T0(); warm_match_fast(ws); PROF_ADD(ws, kProfRows); }
		real hrow[D];
		{ PROF_T0(); mass_row<D>(ws, hrow); PROF_ADD(ws, kProfMass); }
		real dinv;
		{ PROF_T0(); dinv = factorize_regs<Topo>(ws, hrow); PROF_ADD(ws, kProfFact); }
		const int R = ws.R;
		if (lane == 0) ws.cost += 8 + R;
		{
			PROF_T0();
			// J_r[lane] from wave-uniform row descriptors and the lane's own joint position (registers): same value as row_jac()
			const bool hinge = lane >= 2 && lane < D;
			const real mypx = hinge ? ws.px[lane - 2] : 0.0, mypy = hinge ? ws.py[lane - 2] : 0.0;
			const uint32_t mysub = hinge ? ws.M.sub_mask[lane - 2] : 0u;
			real rhs0 = (lane < D) ? (ws.st.tau[lane] - ws.b[lane]) : 0.0;
			if (__builtin_expect(ws.st.pert_on != 0, 0)) { if (lane < D) rhs0 += perturb_gen_force(ws, lane); }
			// J_r^T for row r (r < R), the free right-hand side for r == R
			auto rhs_of = [&](int r) -> real {
				if (r >= R) return rhs0;
				const int kind = ws.row_kind[r], link = ws.row_link[r];
				const real dx = ws.row_dx[r];
				if (kind == 0) return (lane == link + 2) ? dx : 0.0;
				const real dy = ws.row_dy[r], x = ws.row_x[r], y = ws.row_y[r];
				const int link2 = ws.row_link2[r];
				if (lane == 0) return link2 < 0 ? dx : 0.0;
				if (lane == 1) return link2 < 0 ? dy : 0.0;
				const int on = static_cast<int>((mysub >> link) & 1u) - (link2 < 0 ? 0 : static_cast<int>((mysub >> link2) & 1u));
				if (on == 0) return 0.0;
				const real jr = dx * (-(y - mypy)) + dy * (x - mypx);
				return on > 0 ? jr : -jr;
			};
			int r0 = 0;
			for (; r0 + 4 <= R + 1; r0 += 4) {
				real z[4] = {rhs_of(r0), rhs_of(r0 + 1), rhs_of(r0 + 2), rhs_of(r0 + 3)};
				usolve_regs_n<D, 4>(hrow, z);
#pragma unroll
				for (int j = 0; j < 4; ++j) if (lane < D) ws.Z[r0 + j][lane] = z[j];
			}
			if (r0 + 2 <= R + 1) {
				real z[2] = {rhs_of(r0), rhs_of(r0 + 1)};
				usolve_regs_n<D, 2>(hrow, z);
				if (lane < D) { ws.Z[r0][lane] = z[0]; ws.Z[r0 + 1][lane] = z[1]; }
				r0 += 2;
			}
			if (r0 <= R) {
				real z[1] = {rhs_of(r0)};
				usolve_regs_n<D, 1>(hrow, z);
				if (lane < D) ws.Z[r0][lane] = z[0];
			}
			if (lane < D) ws.dinv[lane] = dinv;   // the Delassus product reads 1/d per DoF from LDS
			env_sync();
			PROF_ADD(ws, kProfFsub);
		}
		if (R > 0) {
			{ PROF_T0(); build_delassus_fast<D>(ws, h, dinv); PROF_ADD(ws, kProfDelassus); }
			{ PROF_T0(); pgs_solve_fast<Topo::kPgsRegRows, Topo::kPgsTailInSweep>(ws); PROF_ADD(ws, kProfPgs); }
		}
		{
			PROF_T0();
			real u = 0;
			if (lane < D) {
				real s = h * ws.Z[R][lane];
				for (int r = 0; r < R; ++r) s = fmadd(ws.Z[r][lane], ws.st.ws_lam[r], s);
				u = s * dinv;
			}
			u = utsolve_regs<D>(hrow, u);
			if (lane < D) { const real v = clamp_turn_rate(ws.st.qd[lane] + u, lane, h); ws.st.qd[lane] = v; ws.st.q[lane] += h * v; }
			env_sync();
			PROF_ADD(ws, kProfFinish);
		}
#if defined(DTRL_PROFILE)
		if (threadIdx.x == 0) { ws.prof[kProfRowsSum] += R; ws.prof[kProfSubsteps] += 1; const int bk = R == 0 ? 0 : (R <= 6 ? 1 : (R <= 12 ? 2 : (R <= 18 ? 3 : 4))); ws.prof[kProfR0 + bk] += 1; ws.prof[kProfT0 + bk] += __builtin_readcyclecounter() - prof_sub_t0; }
#endif
	}
	static __device__ void pd_solve(WSFast& ws, real dt)
	{
		const int lane = static_cast<int>(threadIdx.x);
		real hrow[D];
		mass_row<D>(ws, hrow);   // composite inertias come from kin_dyn_terms() in env_step
		const real add = (lane < D) ? dt * ws.kdm[lane] : 0.0;
#pragma unroll
		for (int k = 0; k < D; ++k) if (lane == k) hrow[k] += add;
		const real dinv = factorize_regs<Topo>(ws, hrow);
		real z = (lane < D) ? ws.u[lane] : 0.0;
		z = usolve_regs<D>(hrow, z);
		real u = z * dinv;
		u = utsolve_regs<D>(hrow, u);
		env_sync();
		if (lane < D) ws.u[lane] = u;
		if (lane == 0) ws.R = 0;
		env_sync();
	}
	// cContactManager::Update at the post-step configuration: the controller needs the per-link flags; the same pass builds the row list of
	// the next env-step's first substep (same q, same heightfield window within a launch)
	static __device__ void contacts(WSFast& ws, const DevModel& gm, const GroundRec& g, real h)
	{
		const ContactPts cp = eval_points<true>(ws, gm, g);
		contact_bits_fast(ws, cp.f0, cp.f1, cp.f2);
		build_rows_fast(ws, gm, cp, h);
		if (threadIdx.x == 0) ws.n_pts_active = ws.R;
		env_sync();
	}
};

}  // namespace dtrl
#endif  // __HIP_DEVICE_COMPILE__
