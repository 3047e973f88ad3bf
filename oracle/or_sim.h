// ORACLE (test infrastructure, NOT product code).
// The documented planar articulated-body + contact integrator that stands in for Bullet
// (cWorld::Update -> btDiscreteDynamicsWorld::stepSimulation, /root/reference/sim/World.cpp:96-105).
// Bullet is an un-vendored, un-pinned external of the reference and is absent here, so this part is
// "by documented model" (DESIGN.md section "Integrator v1"), PARITY UNPINNED against Bullet. The reference-visible
// parameters are preserved: gravity (0,-9.8,0) util/MathUtil.h:19; fixed substep dt/n (sim/World.cpp:101-102);
// friction 0.9*0.9 (sim/SimCharacter.cpp:17, sim/GroundVar2D.cpp:10); restitution 0; no damping (sim/World.cpp:52-53);
// joint limits with lo > hi => free (sim/World.cpp:28-29,625); per-joint torque clamp (sim/Joint.cpp:257-264,
// sim/PDController.cpp:98-100); collision groups (sim/SimDog.cpp:5-33); torques held over all substeps.
//
// Model (one substep of length h):
//   H(q) v+ = H v + h (tau - b(q,v)) + J^T lambda,   q+ = q + h v+
//   rows of J: joint limits within limit_slop of a stop (unit rows) and, per penetrating contact sample point, a normal and a
//   tangent row; lambda solved by 10 sweeps of projected Gauss-Seidel in a fixed row order; since round 5 with Bullet's contact persistence (or_model.h:
//   warm_start, link_brk): contact rows start from 0.85 x their previous impulse, a sweep takes limits -> normals -> friction rows, a friction row is resolved only
//   while its normal row carries an impulse, and a sample point keeps its rows while it is within the manifold's breaking threshold above the surface;
//   normal target velocity = min(ERP * max(depth - slop, 0) / h, v_depen_max); friction box |lt| <= mu * ln.
//   Contact sample points per box link: 4 corners + midpoints of the two long edges, tested against the
//   heightfield polyline; depth measured along the cell normal.
// Reference-visible semantics carried by the model (round 2):
//   * hinge limits act on theta + ref_theta (sim/World.cpp:543-553: theta = -getHingeAngle() - ref_theta; :624-626 setLimit(-LimHigh, -LimLow)),
//     ref_theta as cSimCharacter::BuildConstraints computes it (sim/SimCharacter.cpp:846-865): LimLow - ref_theta <= theta <= LimHigh - ref_theta;
//   * a link is "in contact" when a manifold point is within dist_tol = 0.001 in WORLD-SCALED units of the surface
//     (sim/ContactManager.cpp:74-75: pt.getDistance() <= 0.001f), i.e. separation <= 0.001 / world_scale, not only when it penetrates;
//   * at most 4 contact points per link and ground (Bullet's persistent manifold holds 4 points per pair): the deepest four of the link's
//     penetrating sample points; when the row budget is exceeded the DEEPEST points overall get rows (not the lowest link ids).
// Round 3: the box links carry Bullet's collision margin against the ground (M.contact_margin = 0.04 / world_scale): sample points sit on the margin-shrunk
//   box and the separation is measured to the rounded surface. tools/a2_deviation.py found the margin to be what separated this model from the Bullet-shaped
//   sequential-impulse integrator on the goat scene (world scale 1: a 4 cm margin on 3-5 cm thick links).
#pragma once
#include "or_rbd.h"
#include "or_terrain.h"

namespace orc {

struct SimConst {
	static constexpr double erp = 0.2;
	static constexpr double slop = 0.001;
	static constexpr double mu = 0.9 * 0.9;
	static constexpr double v_depen_max = 1.0;
	static constexpr double limit_erp = 0.2;
	static constexpr double limit_slop = 0.005;  // a limit row is active within this band of the stop (robust activation)
	static constexpr int pgs_iters = 10;
	static constexpr int max_rows = 24;
	static constexpr int pts_per_link = 6;
	static constexpr int max_pts_per_link = 4;      // points of one link--ground manifold
	static constexpr int max_pts_per_pair = 2;      // points of one link--link manifold: two convex boxes in the plane touch along a segment at most (Bullet's four 3-D manifold points project onto two)
	static constexpr double contact_dist_tol = 0.001;   // world-scaled units (sim/ContactManager.cpp:74)
	static constexpr double max_turn_per_substep = 1.5707963267948966;
	static constexpr double warmstart_factor = 0.85;   // btContactSolverInfo::m_warmstartingFactor
	static constexpr double hold_eps = 1e-9;           // a normal impulse below this counts as none for the friction rule (see dtrl_types.h kHoldEps)
};

struct Bodies {
	// per-link planar kinematics derived from (q, qd)
	double phi[ORC_MAXL];          // world angle of the joint frame
	double px[ORC_MAXL], py[ORC_MAXL];  // joint world position
	double cx[ORC_MAXL], cy[ORC_MAXL];  // body COM world position
	double psi[ORC_MAXL];          // body world angle = phi + body_theta
	double w[ORC_MAXL];            // angular velocity
	double vpx[ORC_MAXL], vpy[ORC_MAXL];  // velocity of the joint origin
	double vcx[ORC_MAXL], vcy[ORC_MAXL];  // velocity of the COM
};

inline void ForwardKin(const OrcModel& M, const double* q, const double* qd, Bodies& B)
{
	for (int j = 0; j < M.L; ++j) {
		int p = M.parent[j];
		if (p < 0) {
			B.phi[j] = q[2]; B.px[j] = q[0]; B.py[j] = q[1];
			B.w[j] = qd[2]; B.vpx[j] = qd[0]; B.vpy[j] = qd[1];
		} else {
			double c = std::cos(B.phi[p]), s = std::sin(B.phi[p]);
			double ax = M.attach[j][0], ay = M.attach[j][1];
			double rx = c * ax - s * ay, ry = s * ax + c * ay;
			B.px[j] = B.px[p] + rx; B.py[j] = B.py[p] + ry;
			B.phi[j] = B.phi[p] + q[j + 2];
			B.vpx[j] = B.vpx[p] - B.w[p] * ry; B.vpy[j] = B.vpy[p] + B.w[p] * rx;
			B.w[j] = B.w[p] + qd[j + 2];
		}
		double c = std::cos(B.phi[j]), s = std::sin(B.phi[j]);
		double bx = M.body_attach[j][0], by = M.body_attach[j][1];
		double rx = c * bx - s * by, ry = s * bx + c * by;
		B.cx[j] = B.px[j] + rx; B.cy[j] = B.py[j] + ry;
		B.psi[j] = B.phi[j] + M.body_theta[j];
		B.vcx[j] = B.vpx[j] - B.w[j] * ry; B.vcy[j] = B.vpy[j] + B.w[j] * rx;
	}
}

// body-frame sample points of link j (documented contact model). shrink > 0: the points of the MARGIN-SHRUNK box (Bullet's btBoxShape keeps half extents -
// margin as its core and adds the margin back as a rounding radius: the ground test below measures core point -> surface and subtracts the margin; a half
// extent smaller than the margin goes negative, as Bullet's implicitShapeDimensions do)
inline void LinkSamplePoint(const OrcModel& M, int j, int k, double& sx, double& sy, double shrink = 0.0)
{
	double hx = 0.5 * M.body_size[j][0] - shrink, hy = 0.5 * M.body_size[j][1] - shrink;
	switch (k) {
	case 0: sx = -hx; sy = -hy; break;
	case 1: sx = hx; sy = -hy; break;
	case 2: sx = hx; sy = hy; break;
	case 3: sx = -hx; sy = hy; break;
	case 4: if (M.body_size[j][0] >= M.body_size[j][1]) { sx = 0; sy = -hy; } else { sx = -hx; sy = 0; } break;
	default: if (M.body_size[j][0] >= M.body_size[j][1]) { sx = 0; sy = hy; } else { sx = hx; sy = 0; } break;
	}
}

struct ContactPoint { int link; double x, y, depth, nx, ny; int pt; };

// separation of a world point from box `q` (negative inside): max over the two face pairs of (|local coordinate| - half extent); *lx, *ly = local coordinates
inline void PointInBox(const OrcModel& M, const Bodies& B, int q, double x, double y, double& pen_x, double& pen_y, double& lx, double& ly)
{
	const double c = std::cos(B.psi[q]), s = std::sin(B.psi[q]);
	const double dx = x - B.cx[q], dy = y - B.cy[q];
	lx = c * dx + s * dy; ly = -s * dx + c * dy;
	pen_x = 0.5 * M.body_size[q][0] - std::fabs(lx);
	pen_y = 0.5 * M.body_size[q][1] - std::fabs(ly);
}

// signed separation (negative = penetration) of every contact sample point from the heightfield, along the cell normal; +inf for links that collide with nothing
inline void ContactDistances(const OrcModel& M, const Bodies& B, const Ground& g, double* out)
{
	for (int j = 0; j < M.L; ++j) {
		double c = std::cos(B.psi[j]), s = std::sin(B.psi[j]);
		for (int k = 0; k < SimConst::pts_per_link; ++k) {
			double& d = out[j * SimConst::pts_per_link + k];
			d = 1e30;
			if (M.col_group[j] == 0) continue;
			double sx, sy; LinkSamplePoint(M, j, k, sx, sy, M.link_margin[j]);
			double x = B.cx[j] + c * sx - s * sy, y = B.cy[j] + s * sx + c * sy;
			double slope = g.SampleSlope(x);
			d = -(g.SampleHeight(x) - y) / std::sqrt(1.0 + slope * slope) - M.link_margin[j];
		}
	}
}

// contact detection at the current configuration; fills in-contact flags per link and the list of points that get constraint rows
// (ordered by link, then sample point): per link the deepest max_pts_per_link penetrating points; overall the deepest `cap`
// diag (may be null; test diagnostics only): [0] += 1 when a link had more active points than max_pts_per_link, [1] += 1 when the row budget dropped points
inline int DetectContacts(const OrcModel& M, const Bodies& B, const Ground& g, ContactPoint* out, int cap, bool* flags, long long* diag = nullptr)
{
	const int npts = M.L * SimConst::pts_per_link;
	ContactPoint all[ORC_MAXL * SimConst::pts_per_link];
	bool active[ORC_MAXL * SimConst::pts_per_link];
	const double tol = SimConst::contact_dist_tol / M.world_scale;
	for (int j = 0; j < M.L; ++j) {
		flags[j] = false;
		for (int k = 0; k < SimConst::pts_per_link; ++k) active[j * SimConst::pts_per_link + k] = false;
		if (M.col_group[j] == 0) continue;
		double c = std::cos(B.psi[j]), s = std::sin(B.psi[j]);
		for (int k = 0; k < SimConst::pts_per_link; ++k) {
			double sx, sy; LinkSamplePoint(M, j, k, sx, sy, M.link_margin[j]);
			double x = B.cx[j] + c * sx - s * sy;
			double y = B.cy[j] + s * sx + c * sy;
			double h = g.SampleHeight(x);
			double slope = g.SampleSlope(x);
			double inv = 1.0 / std::sqrt(1.0 + slope * slope);
			double nx = -slope * inv, ny = inv;
			double depth = std::fma(h - y, ny, M.link_margin[j]);   // the rounded corner reaches a margin beyond the core point (fused, as the kernel does)
			if (depth >= -tol) flags[j] = true;                 // cContactManager::Update: distance <= dist_tol
			ContactPoint& p = all[j * SimConst::pts_per_link + k];
			p.link = j; p.x = x; p.y = y; p.depth = depth; p.nx = nx; p.ny = ny; p.pt = j * SimConst::pts_per_link + k;
			active[j * SimConst::pts_per_link + k] = depth > -M.link_brk[j];   // penetrating, or within the manifold's breaking threshold above the surface
		}
	}
	// a point outranks another when it is deeper (ties: lower sample-point index)
	auto outranks = [&](int a, int b) { return all[a].depth > all[b].depth || (all[a].depth == all[b].depth && a < b); };
	bool keep[ORC_MAXL * SimConst::pts_per_link];
	int n_keep = 0;
	for (int pt = 0; pt < npts; ++pt) {
		keep[pt] = false;
		if (!active[pt]) continue;
		const int j = pt / SimConst::pts_per_link;
		int rank = 0;
		for (int k = 0; k < SimConst::pts_per_link; ++k) { const int o = j * SimConst::pts_per_link + k; if (o != pt && active[o] && outranks(o, pt)) ++rank; }
		keep[pt] = rank < SimConst::max_pts_per_link;
		n_keep += keep[pt];
		if (diag && !keep[pt] && rank == SimConst::max_pts_per_link) ++diag[0];
	}
	if (diag && n_keep > cap) ++diag[1];
	if (n_keep > cap) {
		bool keep2[ORC_MAXL * SimConst::pts_per_link];
		for (int pt = 0; pt < npts; ++pt) {
			keep2[pt] = false;
			if (!keep[pt]) continue;
			int rank = 0;
			for (int o = 0; o < npts; ++o) if (o != pt && keep[o] && outranks(o, pt)) ++rank;
			keep2[pt] = rank < cap;
		}
		for (int pt = 0; pt < npts; ++pt) keep[pt] = keep2[pt];
	}
	int n = 0;
	for (int pt = 0; pt < npts; ++pt) if (keep[pt] && n < cap) out[n++] = all[pt];
	return n;
}

// ---- link--link contacts (same collision group, no hinge between them): the sample points of either box tested against the other box ----
struct PairContact { int a, b; double x, y, depth, nx, ny; int cand, pair; };   // the normal pushes link a along +n and link b along -n
// candidates of pair (a, b) in the order: a's six sample points against b's box, then b's six against a's. Returns the number of penetrating candidates
// written to cand (<= 12); *min_sep (may be null) = the smallest separation of any candidate point from the partner's box (what a narrowphase would
// report as the pair manifold's distance). Link--link contacts never set a link's contact FLAG: cScenarioSimChar registers the character's parts with
// filter eContactFlagEnvironment (scenarios/ScenarioSimChar.cpp:321), so cContactManager::IsValidContact (sim/ContactManager.cpp:169-175) drops them
inline int PairCandidates(const OrcModel& M, const Bodies& B, int a, int b, PairContact* cand, double* min_sep)
{
	int n = 0;
	for (int side = 0; side < 2; ++side) {
		const int P = side == 0 ? a : b, Q = side == 0 ? b : a;
		const double cp = std::cos(B.psi[P]), sp = std::sin(B.psi[P]);
		const double cq = std::cos(B.psi[Q]), sq = std::sin(B.psi[Q]);
		for (int k = 0; k < SimConst::pts_per_link; ++k) {
			double sx, sy; LinkSamplePoint(M, P, k, sx, sy);
			const double x = B.cx[P] + cp * sx - sp * sy, y = B.cy[P] + sp * sx + cp * sy;
			double px, py, lx, ly; PointInBox(M, B, Q, x, y, px, py, lx, ly);
			if (min_sep) *min_sep = std::min(*min_sep, std::max(-px, -py));
			if (!(px > 0 && py > 0)) continue;
			// push the point out through the nearest face of Q
			double nlx = 0, nly = 0, depth;
			if (px <= py) { nlx = lx >= 0 ? 1.0 : -1.0; depth = px; } else { nly = ly >= 0 ? 1.0 : -1.0; depth = py; }
			PairContact& c = cand[n++];
			c.a = P; c.b = Q; c.x = x; c.y = y; c.depth = depth; c.cand = side * SimConst::pts_per_link + k;
			c.nx = cq * nlx - sq * nly; c.ny = sq * nlx + cq * nly;
		}
	}
	return n;
}
// smallest candidate separation of every pair [n_cpairs] (lock-step tests hand these to the reference as link--link manifolds)
inline void PairDistances(const OrcModel& M, const Bodies& B, double* out)
{
	for (int pr = 0; pr < M.n_cpairs; ++pr) {
		PairContact cand[2 * SimConst::pts_per_link];
		out[pr] = 1e30;
		PairCandidates(M, B, M.cpair_a[pr], M.cpair_b[pr], cand, &out[pr]);
	}
}
// Link--link rows are velocity constraints WITHOUT a penetration-recovery term (v_n >= 0, friction as usual): the two links of a pair hang on one or two
// hinges, a contact point can sit millimetres from the only axis that could separate them, and a Baumgarte target of 0.2 depth / h there asks for
// thousands of rad/s (observed: |qd| > 2000 rad/s, then NaN). Bullet recovers penetration by split impulse, which adds no momentum either.
// pair contacts in pair order; per pair the deepest max_pts_per_pair candidates (ties: the earlier candidate), emitted in candidate order; at most `cap`
inline int DetectPairContacts(const OrcModel& M, const Bodies& B, PairContact* out, int cap)
{
	if (!M.link_contacts) return 0;
	int n = 0;
	for (int pr = 0; pr < M.n_cpairs; ++pr) {
		PairContact cand[2 * SimConst::pts_per_link];
		const int nc = PairCandidates(M, B, M.cpair_a[pr], M.cpair_b[pr], cand, nullptr);
		for (int i = 0; i < nc; ++i) {
			int rank = 0;
			for (int o = 0; o < nc; ++o) if (o != i && (cand[o].depth > cand[i].depth || (cand[o].depth == cand[i].depth && o < i))) ++rank;
			if (rank < SimConst::max_pts_per_pair && n < cap) { out[n] = cand[i]; out[n].pair = pr; ++n; }
		}
	}
	return n;
}

// one substep. H and b (true bias, fix_cj) come from the caller's RBDModel evaluated at (q, qd).
struct Integrator {
	double Jr[SimConst::max_rows][ORC_MAXD];
	double Yr[SimConst::max_rows][ORC_MAXD];
	double Arr[SimConst::max_rows], tgt[SimConst::max_rows], lam[SimConst::max_rows];
	int kind[SimConst::max_rows];  // 0 = limit (lambda >= 0), 1 = contact normal, 2 = contact tangent (paired with previous row)
	// warm starting (M.warm_start, or_model.h): a row keeps its identity across substeps -- 16-bit ids shared with the kernels (dtrl_kernel.h row_id): ground contact
	// 2 x sample point + (0 normal, 1 tangent); link--link contact 512 + 2 x (pair x 12 + candidate) + (0, 1); limit rows 32768 + 2 x joint + side (never matched under
	// Bullet's rule) -- and starts the sweeps from warmstart_factor x the impulse it ended the previous substep with (btContactSolverInfo::m_warmstartingFactor 0.85 on
	// the persistent manifold points' m_appliedImpulse / m_appliedImpulseLateral1)
	int id[SimConst::max_rows], prev_id[SimConst::max_rows], prev_R = 0;
	double prev_lam[SimConst::max_rows];
	void ResetWarmStart() { prev_R = 0; }

	void PointJacobian(const OrcModel& M, const Bodies& B, int link, double x, double y, double dx, double dy, double* row, int D) const
	{
		for (int i = 0; i < D; ++i) row[i] = 0;
		row[0] = dx; row[1] = dy;
		int j = link;
		while (j >= 0) {
			// d(point)/d(theta_j) = z x (point - p_j)
			double rx = x - B.px[j], ry = y - B.py[j];
			row[j + 2] = dx * (-ry) + dy * rx;
			j = M.parent[j];
		}
	}

	// external perturbation applied during the env-step (sim/Perturb.cpp:52-79, sim/World.cpp:445-470): Bullet accumulates the force at
	// the body's COM plus the torque rel_pos x force evaluated when cWorld::Update applies it, and holds both over the substeps
	struct PerturbForce { int link = -1; double fx = 0, fy = 0, torque = 0; bool on = false; };

	// diagnostics for the full-width parity record (tests only; no effect on the arithmetic): [0] substeps with a link over its point cap, [1] substeps over the row
	// budget, [2] substeps with R >= 16 (the kernels' general Delassus path), [3] substeps with link--link rows, [4] substeps, [5] max R, [6] sum of R
	long long diag[7] = {0, 0, 0, 0, 0, 0, 0};
	int sub_ix = 0;                       // index of the substep within its env-step
	double Hm[ORC_MAXD * ORC_MAXD];       // joint-space inertia the substep solves with
	void Substep(const OrcModel& M, RBDModel& rbd, const Ground& ground, double h, double* q, double* qd, const double* tau, const PerturbForce* pf = nullptr)
	{
		const int D = rbd.D;
		rbd.Update(q, qd, /*fix_cj=*/true);
		Bodies B; ForwardKin(M, q, qd, B);
		// -mass_matrix_every= N: H of the env-step's substeps 0, N, 2N ... is held for the substeps in between (sub_ix is set by the caller)
		if (M.mass_matrix_every <= 1 || sub_ix % M.mass_matrix_every == 0)
			for (int i = 0; i < D; ++i) for (int k = 0; k < D; ++k) Hm[i * D + k] = rbd.H[i][k];
		double rhs[ORC_MAXD], dv[ORC_MAXD], v[ORC_MAXD];
		for (int i = 0; i < D; ++i) rhs[i] = tau[i] - rbd.C[i];
		if (pf && pf->on) {
			double jx[ORC_MAXD], jy[ORC_MAXD];
			PointJacobian(M, B, pf->link, B.cx[pf->link], B.cy[pf->link], 1, 0, jx, D);
			PointJacobian(M, B, pf->link, B.cx[pf->link], B.cy[pf->link], 0, 1, jy, D);
			double add[ORC_MAXD];
			for (int i = 0; i < D; ++i) add[i] = jx[i] * pf->fx + jy[i] * pf->fy;
			for (int j = pf->link; j >= 0; j = M.parent[j]) add[j + 2] += pf->torque;
			for (int i = 0; i < D; ++i) rhs[i] += add[i];
		}
		SolveLDLT(D, Hm, D, rhs, dv);
		for (int i = 0; i < D; ++i) v[i] = qd[i] + h * dv[i];

		int R = 0;
		for (int j = 1; j < M.L; ++j) {
			if (M.lim_lo[j] > M.lim_hi[j]) continue;
			double th = q[j + 2];
			const double lo = M.lim_lo[j] - M.ref_theta[j], hi = M.lim_hi[j] - M.ref_theta[j];   // limits act on theta + ref_theta
			if (th <= lo + SimConst::limit_slop && R < SimConst::max_rows) {
				for (int i = 0; i < D; ++i) Jr[R][i] = 0;
				Jr[R][j + 2] = 1; kind[R] = 0; id[R] = 32768 + 2 * j; tgt[R] = SimConst::limit_erp * std::max(lo - th, 0.0) / h; ++R;
			} else if (th >= hi - SimConst::limit_slop && R < SimConst::max_rows) {
				for (int i = 0; i < D; ++i) Jr[R][i] = 0;
				Jr[R][j + 2] = -1; kind[R] = 0; id[R] = 32768 + 2 * j + 1; tgt[R] = SimConst::limit_erp * std::max(th - hi, 0.0) / h; ++R;
			}
		}
		ContactPoint cps[SimConst::max_rows / 2];
		bool flags[ORC_MAXL];
		int cap = (SimConst::max_rows - R) / 2;
		int nc = DetectContacts(M, B, ground, cps, cap, flags, diag);
		for (int c = 0; c < nc; ++c) {
			const ContactPoint& cp = cps[c];
			PointJacobian(M, B, cp.link, cp.x, cp.y, cp.nx, cp.ny, Jr[R], D);
			kind[R] = 1; id[R] = 2 * cp.pt;
			double t = SimConst::erp * std::max(cp.depth - SimConst::slop, 0.0) / h;
			tgt[R] = cp.depth < 0 ? cp.depth / h : std::min(t, SimConst::v_depen_max); ++R;   // a point still above the surface may approach by its distance per substep
			PointJacobian(M, B, cp.link, cp.x, cp.y, cp.ny, -cp.nx, Jr[R], D);
			kind[R] = 2; id[R] = 2 * cp.pt + 1; tgt[R] = 0; ++R;
		}
		// link--link contacts take what is left of the row budget, in pair order
		PairContact pcs[SimConst::max_rows / 2];
		const int npc = DetectPairContacts(M, B, pcs, (SimConst::max_rows - R) / 2);
		for (int c = 0; c < npc; ++c) {
			const PairContact& pc = pcs[c];
			double jb[ORC_MAXD];
			for (int t = 0; t < 2; ++t) {
				const double dx = t == 0 ? pc.nx : pc.ny, dy = t == 0 ? pc.ny : -pc.nx;
				PointJacobian(M, B, pc.a, pc.x, pc.y, dx, dy, Jr[R], D);
				PointJacobian(M, B, pc.b, pc.x, pc.y, dx, dy, jb, D);
				for (int i = 0; i < D; ++i) Jr[R][i] -= jb[i];
				kind[R] = 1 + t; id[R] = 512 + 2 * (pc.pair * 2 * SimConst::pts_per_link + pc.cand) + t;
				tgt[R] = 0.0;   // velocity-level non-penetration only (see DetectPairContacts)
				++R;
			}
		}
		diag[2] += R >= 16; diag[3] += npc > 0; ++diag[4]; diag[5] = std::max<long long>(diag[5], R); diag[6] += R;
		for (int r = 0; r < R; ++r) {
			SolveLDLT(D, Hm, D, Jr[r], Yr[r]);
			double a = 0; for (int i = 0; i < D; ++i) a += Jr[r][i] * Yr[r][i];
			Arr[r] = a; lam[r] = 0;
		}
		const int ws = M.warm_start;
		const bool rule = ws == 1 || ws == 2 || ws == 3;   // Bullet's contact persistence (or_model.h)
		if (ws == 3) {
			// (ablation) Bullet's friction direction: along the relative tangential velocity before the solve, else the plane-space vector (-n_y, n_x)
			for (int r = 0; r < R; ++r) if (kind[r] == 2) {
				double w = 0; for (int i = 0; i < D; ++i) w += Jr[r][i] * v[i];
				if (!(w * w > 1.1920929e-7) || w < 0) for (int i = 0; i < D; ++i) { Jr[r][i] = -Jr[r][i]; Yr[r][i] = -Yr[r][i]; }
			}
		}
		if (ws) {
			for (int r = 0; r < R; ++r) {
				if (Arr[r] < 1e-12) continue;
				if (rule && (kind[r] == 0 || id[r] >= 512)) continue;   // Bullet's rule here: ground contact rows only (limit rows: the solver zeroes typed constraints' rows; link--link rows: see or_model.h)
				for (int p = 0; p < prev_R; ++p) if (prev_id[p] == id[r]) { lam[r] = SimConst::warmstart_factor * prev_lam[p]; break; }
			}
			for (int r = 0; r < R; ++r) if (lam[r] != 0) for (int i = 0; i < D; ++i) v[i] += Yr[r][i] * lam[r];
		}
		const int passes = (ws == 1 || ws == 3) ? 2 : 1;   // pass 0: limit + normal rows, pass 1: friction rows (btSequentialImpulseConstraintSolver::solveSingleIteration)
		for (int it = 0; it < SimConst::pgs_iters; ++it) {
			for (int pass = 0; pass < passes; ++pass)
			for (int r = 0; r < R; ++r) {
				if (passes == 2 && ((pass == 0) == (kind[r] == 2))) continue;
				if (Arr[r] < 1e-12) continue;
				// a friction row is resolved only while its normal row carries an impulse: the cached friction impulse of a contact without normal force stays applied
				if (rule && kind[r] == 2 && !(lam[r - 1] > SimConst::hold_eps)) continue;
				double w = 0; for (int i = 0; i < D; ++i) w += Jr[r][i] * v[i];
				double nl = lam[r] + (tgt[r] - w) / Arr[r];
				if (kind[r] == 2) { double lim = SimConst::mu * lam[r - 1]; nl = std::min(std::max(nl, -lim), lim); }
				else nl = std::max(nl, 0.0);
				double dl = nl - lam[r];
				lam[r] = nl;
				for (int i = 0; i < D; ++i) v[i] += Yr[r][i] * dl;
			}
		}
		prev_R = R;
		for (int r = 0; r < R; ++r) { prev_id[r] = id[r]; prev_lam[r] = lam[r]; }
		// Bullet clamps a body's angular velocity to MAX_ANGVEL = pi / 2 per internal step (btRigidBody::integrateVelocities); in joint coordinates: every hinge
		// rate and the root's spin. 4712 rad/s at 1/3000 s: only the whipping tail of a crashed character gets there, and unclamped the explicit Coriolis terms overflow
		const double vmax = SimConst::max_turn_per_substep / h;
		for (int i = 2; i < D; ++i) v[i] = std::min(std::max(v[i], -vmax), vmax);
		for (int i = 0; i < D; ++i) { qd[i] = v[i]; q[i] += h * v[i]; }
	}
};

}  // namespace orc
