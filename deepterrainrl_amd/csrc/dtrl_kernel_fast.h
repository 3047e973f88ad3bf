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

// row `lane` of the joint-space inertia matrix from the composite quantities in LDS (same formulas as mass_matrix())
template <int D>
__device__ __forceinline__ void mass_row(const WS& ws, real (&h)[D])
{
	const int d = static_cast<int>(threadIdx.x);
	const int l = d >= 2 ? d - 2 : 0;                       // my link (lanes >= D compute garbage that is never used)
	const bool valid = d < D;
	const real M0 = ws.sm[0];
#pragma unroll
	for (int c = 0; c < D; ++c) {
		real v = 0;
		if (valid) {
			if (d < 2) {
				// translation rows: H[d][d] = M, H[x][c>=2] / H[y][c>=2] from the composite of link(c)
				if (c == d) v = M0;
				else if (c >= 2) {
					const int lc = c - 2;
					v = (d == 0) ? -(ws.smy[lc] - ws.sm[lc] * ws.py[lc]) : (ws.smx[lc] - ws.sm[lc] * ws.px[lc]);
				}
			} else if (c < 2) {
				v = (c == 0) ? -(ws.smy[l] - ws.sm[l] * ws.py[l]) : (ws.smx[l] - ws.sm[l] * ws.px[l]);
			} else {
				const int lc = c - 2;
				// deeper link of the pair carries the composite; pairs that are not ancestor-related are zero
				const bool c_anc_of_me = (ws.M.sub_mask[lc] >> l) & 1u;
				const bool me_anc_of_c = (ws.M.sub_mask[l] >> lc) & 1u;
				if (c_anc_of_me || me_anc_of_c) {
					const int deep = c_anc_of_me ? l : lc, anc = c_anc_of_me ? lc : l;
					const real m = ws.sm[deep], mx = ws.smx[deep], my = ws.smy[deep], I = ws.sI[deep];
					const real plx = ws.px[deep], ply = ws.py[deep], pax = ws.px[anc], pay = ws.py[anc];
					v = I - ((plx + pax) * mx + (ply + pay) * my) + m * (plx * pax + ply * pay);
				}
			}
		}
		h[c] = v;
	}
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
__device__ __forceinline__ void detect_contacts_fast(WS& ws, const GroundRec& g)
{
	// first phase of detect_contacts() (sample points), then ballot instead of the serial flag scan
	const int lane = static_cast<int>(threadIdx.x);
	for (int pt = lane; pt < ws.M.L * kPtsPerLink; pt += kGroup) {
		const int j = pt / kPtsPerLink, k = pt - j * kPtsPerLink;
		int active = 0;
		if (ws.M.col[j] != 0) {
			real hx = ws.M.body_half[j][0], hy = ws.M.body_half[j][1];
			real sx, sy;
			switch (k) {
			case 0: sx = -hx; sy = -hy; break;
			case 1: sx = hx; sy = -hy; break;
			case 2: sx = hx; sy = hy; break;
			case 3: sx = -hx; sy = hy; break;
			case 4: if (hx >= hy) { sx = 0; sy = -hy; } else { sx = -hx; sy = 0; } break;
			default: if (hx >= hy) { sx = 0; sy = hy; } else { sx = hx; sy = 0; } break;
			}
			real s, c; sincos(ws.psi[j], &s, &c);
			real x = ws.cx[j] + c * sx - s * sy;
			real y = ws.cy[j] + s * sx + c * sy;
			real slope;
			real h = sample_ground(g, ws.st.q[0] + x, &slope, nullptr, nullptr, nullptr);
			real inv = 1.0 / sqrt(1.0 + slope * slope);
			real nx = -slope * inv, ny = inv;
			real depth = (h - (ws.st.q[1] + y)) * ny;
			if (depth > 0) { active = 1; ws.pt_x[pt] = x; ws.pt_y[pt] = y; ws.pt_depth[pt] = depth; ws.pt_nx[pt] = nx; ws.pt_ny[pt] = ny; }
		}
		ws.pt_active[pt] = active;
	}
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

template <int D>
struct FastPath {
	static __device__ void substep(WS& ws, const GroundRec& g, real h)
	{
		const int lane = static_cast<int>(threadIdx.x);
		{ PROF_T0(); forward_kinematics(ws); PROF_ADD(ws, kProfFK); }
		real hrow[D];
		{ PROF_T0(); composite_inertia(ws, false); mass_row<D>(ws, hrow); PROF_ADD(ws, kProfMass); }
		{ PROF_T0(); bias_force(ws, false); PROF_ADD(ws, kProfBias); }
		real dinv;
		{ PROF_T0(); dinv = factorize_regs<D>(hrow); PROF_ADD(ws, kProfFact); }
		{ PROF_T0(); detect_contacts_fast(ws, g); PROF_ADD(ws, kProfDetect); }
		{ PROF_T0(); build_rows_fast(ws, h); PROF_ADD(ws, kProfRows); }
		const int R = ws.R;
		{
			PROF_T0();
			for (int r = 0; r <= R; ++r) {
				real z = 0;
				if (lane < D) z = (r < R) ? row_jac(ws, r, lane) : (ws.st.tau[lane] - ws.b[lane]);
				z = fsub_regs<D>(hrow, z);
				if (lane < D) ws.Z[r][lane] = z;
			}
			if (lane < D) ws.dinv[lane] = dinv;
			__syncthreads();
			PROF_ADD(ws, kProfFsub);
		}
		if (R > 0) {
			{ PROF_T0(); build_delassus(ws, h); PROF_ADD(ws, kProfDelassus); }
			{ PROF_T0(); pgs_solve(ws); PROF_ADD(ws, kProfPgs); }
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
		mass_row<D>(ws, hrow);   // composite_inertia() already ran at the top of controller_update
		const real add = (lane < D) ? dt * ws.kdv[lane] : 0.0;
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
	static __device__ void contacts(WS& ws, const GroundRec& g) { detect_contacts_fast(ws, g); }
};

}  // namespace dtrl
#endif  // __HIP_DEVICE_COMPILE__
