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
// Row layout after factorize_regs(): h[k] for k < lane = L_{lane,k}; h[lane] = d_lane; h[k] for k > lane = L_{k,lane}
// (the transpose copy, so back substitution also only needs the lane's own registers).
#pragma once
#include "dtrl_kernel.h"

#if defined(__HIP_DEVICE_COMPILE__)

namespace dtrl {

__device__ __forceinline__ real bcast(real v, int src)   // src must be wave-uniform
{
	int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
	int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
	return __hiloint2double(hi, lo);
}

// row `lane` of the joint-space inertia matrix from the composite quantities in LDS (same formulas and operand order as
// mass_matrix()). Each hinge lane evaluates the closed form once per ancestor and parks the values in LDS (the storage of
// the unused LDS copy of H); every lane then assembles its row with one LDS read per column: the entry (d, c) lives in
// the table row of the DEEPER link at the path position of the shallower one.
template <int D>
__device__ __forceinline__ void mass_row(WS& ws, real (&h)[D])
{
	const int d = static_cast<int>(threadIdx.x);
	const int l = d >= 2 ? d - 2 : 0;
	const bool valid = d < D;
	real (*T)[kMaxDepth + 2] = reinterpret_cast<real (*)[kMaxDepth + 2]>(&ws.H[0][0]);   // T[link][0..depth] + hx, hy at [kMaxDepth], [kMaxDepth+1]
	if (valid && d >= 2) {
		const real m = ws.sm[l], mx = ws.smx[l], my = ws.smy[l], I = ws.sI[l];
		const real plx = ws.px[l], ply = ws.py[l];
		T[l][kMaxDepth] = -(my - m * ply);
		T[l][kMaxDepth + 1] = (mx - m * plx);
		const int dep = ws.M.depth[l];
		for (int k = 0; k <= dep; ++k) {
			const int a = ws.M.path[l][k];
			const real pax = ws.px[a], pay = ws.py[a];
			T[l][k] = I - ((plx + pax) * mx + (ply + pay) * my) + m * (plx * pax + ply * pay);
		}
	}
	__syncthreads();
	const real M0 = ws.sm[0];
	const uint32_t my_sub = valid ? ws.M.sub_mask[l] : 0u, my_anc = valid ? ws.M.anc_mask[l] : 0u;
	const int my_depth = ws.M.depth[l];
#pragma unroll
	for (int c = 0; c < D; ++c) {
		real v = 0;
		if (valid) {
			if (d < 2) {
				if (c == d) v = M0;
				else if (c >= 2) v = T[c - 2][kMaxDepth + d];
			} else if (c < 2) {
				v = T[l][kMaxDepth + c];
			} else {
				const int lc = c - 2;
				const int dc = __builtin_amdgcn_readlane(my_depth, c);          // depth of link lc (lane c = lc + 2 holds it)
				if ((my_anc >> lc) & 1u) v = T[l][dc];                           // lc is an ancestor of (or is) my link
				else if ((my_sub >> lc) & 1u) v = T[lc][my_depth];               // lc is a descendant
			}
		}
		h[c] = v;
	}
	__syncthreads();   // T is dead; the storage may be reused
}

// in-register LDL^T; returns 1/d_lane. Same elimination order and operations as factorize().
template <int D>
__device__ __forceinline__ real factorize_regs(real (&h)[D])
{
	const int lane = static_cast<int>(threadIdx.x);
#pragma unroll
	for (int k = 0; k < D - 1; ++k) {
		const real dk = bcast(h[k], k);
		const real ak = h[k];
		const real lik = ak / dk;
#pragma unroll
		for (int j = k + 1; j < D; ++j) {
			const real ajk = bcast(ak, j);
			const real ljk = bcast(lik, j);
			if (lane >= j) h[j] -= lik * ajk;
			if (lane == k) h[j] = ljk;
		}
		if (lane > k) h[k] = lik;
	}
	real dinv = 0;
#pragma unroll
	for (int k = 0; k < D; ++k) if (lane == k) dinv = 1.0 / h[k];
	return dinv;
}
// z = L^-1 rhs (lane i holds component i)
template <int D>
__device__ __forceinline__ real fsub_regs(const real (&h)[D], real z)
{
	const int lane = static_cast<int>(threadIdx.x);
#pragma unroll
	for (int k = 0; k < D - 1; ++k) {
		const real zk = bcast(z, k);
		if (lane > k) z -= h[k] * zk;
	}
	return z;
}
// x = L^-T u
template <int D>
__device__ __forceinline__ real bsub_regs(const real (&h)[D], real u)
{
	const int lane = static_cast<int>(threadIdx.x);
#pragma unroll
	for (int i = D - 1; i >= 1; --i) {
		const real ui = bcast(u, i);
		if (lane < i) u -= h[i] * ui;
	}
	return u;
}

// contact flags + ordered constraint-row list with wave ballots (same order as detect_contacts()/build_rows())
__device__ __forceinline__ void contact_bits_fast(WS& ws)
{
	const int lane = static_cast<int>(threadIdx.x);
	const int npts = ws.M.L * kPtsPerLink;
	const int a0 = (lane < npts) ? ws.pt_active[lane] : 0;
	const int a1 = (lane + kGroup < npts) ? ws.pt_active[lane + kGroup] : 0;
	const unsigned long long m0 = __ballot(a0), m1 = __ballot(a1);
	int any = 0;
	if (lane < ws.M.L) {
		const int b = lane * kPtsPerLink;
		unsigned long long bits = (b < 64) ? (m0 >> b) : 0ull;
		if (b < 64 && b + kPtsPerLink > 64) bits |= m1 << (64 - b);
		if (b >= 64) bits = m1 >> (b - 64);
		any = (bits & ((1ull << kPtsPerLink) - 1ull)) != 0;
	}
	const unsigned long long lm = __ballot(any);
	if (lane == 0) ws.st.contact_bits = static_cast<uint32_t>(lm);
}
__device__ __forceinline__ void detect_contacts_fast(WS& ws, const DevModel& gm, const GroundRec& g)
{
	// first phase of detect_contacts() (sample points), then ballots instead of the serial flag scan
	const int lane = static_cast<int>(threadIdx.x);
	for (int pt = lane; pt < ws.M.L * kPtsPerLink; pt += kGroup) sample_contact_point(ws, gm, g, pt);
	__syncthreads();
	contact_bits_fast(ws);
	__syncthreads();
}
__device__ __forceinline__ void build_rows_fast(WS& ws, real h)
{
	const int lane = static_cast<int>(threadIdx.x);
	const unsigned long long below = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
	// joint limits, ordered by joint id
	int lim = 0; real tgt = 0;
	if (lane >= 1 && lane < ws.M.L && !(ws.M.lim_lo[lane] > ws.M.lim_hi[lane])) {
		const real th = ws.st.q[lane + 2];
		if (th <= ws.M.lim_lo[lane] + kLimitSlop) { lim = 1; tgt = kLimitErp * fmax(ws.M.lim_lo[lane] - th, 0.0) / h; }
		else if (th >= ws.M.lim_hi[lane] - kLimitSlop) { lim = -1; tgt = kLimitErp * fmax(th - ws.M.lim_hi[lane], 0.0) / h; }
	}
	const unsigned long long ml = __ballot(lim != 0);
	const int rl = __popcll(ml & below);
	int R0 = __popcll(ml); if (R0 > kMaxRows) R0 = kMaxRows;
	if (lim != 0 && rl < kMaxRows) { ws.row_kind[rl] = 0; ws.row_link[rl] = lane; ws.row_dx[rl] = lim; ws.row_tgt[rl] = tgt; }
	// contacts, ordered by sample-point index
	const int cap = (kMaxRows - R0) / 2;
	const int npts = ws.M.L * kPtsPerLink;
	const int a0 = (lane < npts) ? ws.pt_active[lane] : 0;
	const int a1 = (lane + kGroup < npts) ? ws.pt_active[lane + kGroup] : 0;
	const unsigned long long m0 = __ballot(a0), m1 = __ballot(a1);
	const int n0 = __popcll(m0), n1 = __popcll(m1);
#pragma unroll
	for (int half = 0; half < 2; ++half) {
		const int a = half ? a1 : a0;
		const int pt = lane + half * kGroup;
		const int rank = half ? (n0 + __popcll(m1 & below)) : __popcll(m0 & below);
		if (a && rank < cap) {
			const int R = R0 + 2 * rank;
			const int j = pt / kPtsPerLink;
			const real t = kErp * fmax(ws.pt_depth[pt] - kSlop, 0.0) / h;
			// NOTE: the point arrays alias the Delassus matrix, not the row arrays, so reading them here is safe
			ws.row_kind[R] = 1; ws.row_link[R] = j; ws.row_x[R] = ws.pt_x[pt]; ws.row_y[R] = ws.pt_y[pt];
			ws.row_dx[R] = ws.pt_nx[pt]; ws.row_dy[R] = ws.pt_ny[pt]; ws.row_tgt[R] = fmin(t, kVDepenMax);
			ws.row_kind[R + 1] = 2; ws.row_link[R + 1] = j; ws.row_x[R + 1] = ws.pt_x[pt]; ws.row_y[R + 1] = ws.pt_y[pt];
			ws.row_dx[R + 1] = ws.pt_ny[pt]; ws.row_dy[R + 1] = -ws.pt_nx[pt]; ws.row_tgt[R + 1] = 0;
		}
	}
	int nc = n0 + n1; if (nc > cap) nc = cap;
	if (lane == 0) ws.R = R0 + 2 * nc;
	__syncthreads();
}

// projected Gauss-Seidel in lambda space with the row state in registers: lane s owns row s (w_s, lambda_s, 1/A_ss, kind);
// a row update is a handful of v_readlane broadcasts + one FMA per lane against column r of the Delassus matrix (LDS,
// conflict-free 8-byte reads). Same operations and order as pgs_solve().
__device__ __forceinline__ void pgs_solve_fast(WS& ws)
{
	const int lane = static_cast<int>(threadIdx.x);
	const int R = ws.R;
	const bool mine = lane < R;
	real w = mine ? ws.wv[lane] : 0.0, lam = 0.0;
	const real rinv = mine ? ws.rinv[lane] : 0.0;
	const int kind = mine ? ws.row_kind[lane] : 0;
	for (int it = 0; it < kPgsIters; ++it) {
		for (int r = 0; r < R; ++r) {
			const real a_sr = mine ? ws.A[lane][r] : 0.0;
			const real ri = bcast(rinv, r);
			if (ri != 0.0) {
				const real lam_r = bcast(lam, r);
				real nl = lam_r - bcast(w, r) * ri;
				if (__builtin_amdgcn_readlane(kind, r) == 2) { const real lim = kMu * bcast(lam, r - 1); nl = fmin(fmax(nl, -lim), lim); }
				else nl = fmax(nl, 0.0);
				const real dl = nl - lam_r;
				if (lane == r) lam = nl;
				w += a_sr * dl;
			}
		}
	}
	if (mine) ws.lam[lane] = lam;
	__syncthreads();
}

template <int D>
struct FastPath {
	static __device__ void substep(WS& ws, const DevModel& gm, const GroundRec& g, real h)
	{
		const int lane = static_cast<int>(threadIdx.x);
		{ PROF_T0(); kin_dyn_terms(ws, false); PROF_ADD(ws, kProfFK); }
		real hrow[D];
		{ PROF_T0(); mass_row<D>(ws, hrow); PROF_ADD(ws, kProfMass); }
		real dinv;
		{ PROF_T0(); dinv = factorize_regs<D>(hrow); PROF_ADD(ws, kProfFact); }
		{ PROF_T0(); detect_contacts_fast(ws, gm, g); PROF_ADD(ws, kProfDetect); }
		{ PROF_T0(); build_rows_fast(ws, h); PROF_ADD(ws, kProfRows); }
		const int R = ws.R;
		{
			PROF_T0();
			// J_r[lane] from wave-uniform row descriptors and the lane's own joint position (registers): same value as row_jac()
			const bool hinge = lane >= 2 && lane < D;
			const real mypx = hinge ? ws.px[lane - 2] : 0.0, mypy = hinge ? ws.py[lane - 2] : 0.0;
			const uint32_t mysub = hinge ? ws.M.sub_mask[lane - 2] : 0u;
			const real rhs0 = (lane < D) ? (ws.st.tau[lane] - ws.b[lane]) : 0.0;
			for (int r = 0; r <= R; ++r) {
				real z = rhs0;
				if (r < R) {
					const int kind = ws.row_kind[r], link = ws.row_link[r];
					const real dx = ws.row_dx[r];
					if (kind == 0) z = (lane == link + 2) ? dx : 0.0;
					else {
						const real dy = ws.row_dy[r], x = ws.row_x[r], y = ws.row_y[r];
						z = 0.0;
						if (lane == 0) z = dx;
						else if (lane == 1) z = dy;
						else if ((mysub >> link) & 1u) z = dx * (-(y - mypy)) + dy * (x - mypx);
					}
				}
				z = fsub_regs<D>(hrow, z);
				if (lane < D) ws.Z[r][lane] = z;
			}
			if (lane < D) ws.dinv[lane] = dinv;
			__syncthreads();
			PROF_ADD(ws, kProfFsub);
		}
		if (R > 0) {
			{ PROF_T0(); build_delassus(ws, h); PROF_ADD(ws, kProfDelassus); }
			{ PROF_T0(); pgs_solve_fast(ws); PROF_ADD(ws, kProfPgs); }
		}
		{
			PROF_T0();
			real u = 0;
			if (lane < D) {
				real s = h * ws.Z[R][lane];
				for (int r = 0; r < R; ++r) s += ws.Z[r][lane] * ws.lam[r];
				u = s * dinv;
			}
			u = bsub_regs<D>(hrow, u);
			if (lane < D) { const real v = ws.st.qd[lane] + u; ws.st.qd[lane] = v; ws.st.q[lane] += h * v; }
			__syncthreads();
			PROF_ADD(ws, kProfFinish);
		}
#if defined(DTRL_PROFILE)
		if (threadIdx.x == 0) { ws.prof[kProfRowsSum] += R; ws.prof[kProfSubsteps] += 1; }
#endif
	}
	static __device__ void pd_solve(WS& ws, real dt)
	{
		const int lane = static_cast<int>(threadIdx.x);
		real hrow[D];
		mass_row<D>(ws, hrow);   // composite inertias come from kin_dyn_terms(ws, true) in env_step
		const real add = (lane < D) ? dt * ws.kdm[lane] : 0.0;
#pragma unroll
		for (int k = 0; k < D; ++k) if (lane == k) hrow[k] += add;
		const real dinv = factorize_regs<D>(hrow);
		real z = (lane < D) ? ws.u[lane] : 0.0;
		z = fsub_regs<D>(hrow, z);
		real u = z * dinv;
		u = bsub_regs<D>(hrow, u);
		__syncthreads();
		if (lane < D) ws.u[lane] = u;
		if (lane == 0) ws.R = 0;
		__syncthreads();
	}
	static __device__ void contacts(WS& ws, const DevModel& gm, const GroundRec& g) { detect_contacts_fast(ws, gm, g); }
};

}  // namespace dtrl
#endif  // __HIP_DEVICE_COMPILE__
