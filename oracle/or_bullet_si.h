// or_bullet_si.h -- TEST INFRASTRUCTURE (oracle/): a maximal-coordinate SEQUENTIAL-IMPULSE rigid-body step in the shape of Bullet 2.8x's
// btDiscreteDynamicsWorld::internalSingleStepSimulation + btSequentialImpulseConstraintSolver, restated from Bullet's published algorithm and
// defaults -- Bullet's source is absent from /root/reference and from this image, so this is "parity unpinned" like Integrator v1.
// Its purpose is the OTHER end of the modelling choice: Integrator v1 (oracle/or_sim.h, the product kernel) is reduced-coordinate (exact hinges,
// Delassus-space PGS, no margins / warm start / split impulse); this file keeps Bullet's structure -- one rigid body per link, hinges as velocity
// constraints with ERP drift correction, 10 Gauss-Seidel sweeps over joint rows -> contact normals -> friction rows, warm-started persistent contact
// points, collision margins with a contact breaking threshold, split-impulse penetration recovery, angular-limit rows with bias / relaxation -- so
// that running the REFERENCE'S OWN controllers (oracle/_ref/libref_sim.so) on both integrators measures what the reduced-coordinate model costs
// behaviourally (tools/a2_deviation.py, tests/test_reference_sim.py). Only oracle/_ref_build/ref_sim_api.cpp includes it.
//
// It operates directly on the Bullet stand-in's world (oracle/_ref_build/stubs_bullet): the rigid bodies, hinge constraints, collision filter groups,
// box shapes with their margins, heightfield / plane ground shapes that the reference's unchanged sim/World.cpp, sim/SimCharacter.cpp and
// sim/GroundVar2D.cpp created -- in Bullet's scaled units (world scale 4), with the forces and torques the reference applied.
//
// What follows Bullet (call sites: sim/World.cpp:61-77 solver / world construction with default btContactSolverInfo, :96-105 stepSimulation(dt, n, dt / n),
// :600-631 btHingeConstraint(A = parent, B = child, pivots, axis z) + setLimit(-high, -low), :157-183 linear factor (1, 1, 0) / angular factor (0, 0, 1)):
//   * per substep: v += h (F / m + g), w += h I^-1 tau (forces persist over the substeps of one stepSimulation call; zero damping, sim/World.cpp:52-53);
//     angular speed clamped to (pi / 2) / h; collision detection at the current transforms; constraint setup; solve; x += h v (+ push), rotation likewise;
//   * solver info defaults: 10 iterations, erp 0.2 (joints, shallow contacts), erp2 0.8 with split impulse below -0.04 penetration, warm-starting
//     factor 0.85 on contact normal and friction impulses, global CFM 0, linear slop 0, restitution 0, one friction direction (the relative
//     tangential velocity, else the plane-space vector), friction mu = mu_A mu_B, friction limits +- mu x the normal row's accumulated impulse;
//   * hinge: point-to-point rows at the pivots (rhs = erp / h x pivot separation, no warm start) + the angular-limit row of btAngularLimit
//     (active only while the angle is outside [low, high]; rhs = max(erp / h x violation, -relaxation x approach velocity) x bias factor 0.3;
//     relaxation 1, softness unused by the row) with one-sided impulse bounds; low > high = free;
//   * contact row: rel_vel along the normal, positional error -penetration x erp / h (or, for a point still `dist` above the surface, the allowed
//     approach -dist / h), split impulse: penetration below the threshold is recovered through push / turn velocities that move the transform but
//     never enter the velocity (turn x 0.1); contact points persist between steps with their applied impulses while they stay within the breaking
//     threshold (0.02 x the pair's smaller angular-motion disc: btCollisionDispatcher::getNewManifold with its default flag
//     CD_USE_RELATIVE_CONTACT_BREAKING_THRESHOLD; 1.35 mm for a dog toe), at most 4 per body pair (the deepest are kept); box margin = min(0.04, 0.1 x the
//     smallest half extent) (CONVEX_DISTANCE_MARGIN through btBoxShape's setSafeMargin; concave ground shapes: 0). Round 3 ran this comparator with 0.04 / 0.02
//     on every box (the `safe_margin = 0, relative_breaking = 0` ablation reproduces it); at world scale 1 that inverted the goat's 2.5 cm boxes.
// What is a stand-in here (Bullet's narrowphase cannot be restated without its source): contact GENERATION. Box vs ground: the four corners of the
// margin-shrunk box against the terrain polyline (distance along the cell normal, minus the margin) plus terrain vertices against the box; box vs box
// (same collision group, not hinge-linked, sim/SimDog.cpp:73-81): the corners of either margin-shrunk box against the other. The hinge's
// frame-offset variant of the point-to-point rows (mass-weighted anchor) is not modelled; both variants constrain the same two in-plane freedoms.
// Planar reduction: with the reference's factors every body has the freedoms (x, y, rotation about z); rows along z or about x / y have zero effective
// mass and Bullet skips their impulses, so the 2-D system below is the 3-D one.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>

#include "btBulletDynamicsCommon.h"
#include <BulletCollision/CollisionShapes/btHeightfieldTerrainShape.h>

namespace bsi {

struct Params {
	int iterations = 10;             // btContactSolverInfo::m_numIterations
	double erp = 0.2, erp2 = 0.8;    // m_erp, m_erp2
	int split_impulse = 1;           // m_splitImpulse
	double split_threshold = -0.04;  // m_splitImpulsePenetrationThreshold
	double split_turn_erp = 0.1;     // m_splitImpulseTurnErp
	int warmstarting = 1;            // SOLVER_USE_WARMSTARTING
	double warmstart_factor = 0.85;  // m_warmstartingFactor
	double breaking = 0.02;          // gContactBreakingThreshold
	int relative_breaking = 1;       // CD_USE_RELATIVE_CONTACT_BREAKING_THRESHOLD (btCollisionDispatcher's default): a pair's threshold = 0.02 x the smaller angular-motion disc
	int vertex_contacts = 1;         // terrain vertices against the box faces (part of the stand-in contact generation); 0 = corners only, as Integrator v1 samples
	int friction_warmstart = 1;      // warm start the friction rows as well (SOLVER_USE_WARMSTARTING covers them in Bullet 2.8x's setFrictionConstraintImpulse); 0 = normals only
	int safe_margin = 1;             // btBoxShape's setSafeMargin: margin = min(0.04, 0.1 x smallest half extent); 0 = 0.04 on every box (the round-3 comparator)
	int max_points = 4;              // MANIFOLD_CACHE_SIZE
	int use_margin = 1;              // box margins (0 = sharp boxes, distance without margin)
	int link_contacts = 1;           // box-box contacts between same-group, not hinge-linked links
	int friction_skip = 1;           // diagnostic: Bullet resolves a friction row only while its normal row carries an impulse (solveSingleIteration: `if (totalImpulse > 0)`); 0 = always (the box clamps to +-mu x 0)
	int friction_dir = 1;            // diagnostic: 1 = Bullet's direction (along the relative tangential velocity, else plane space); 0 = always the plane-space vector
	int friction_ws_lifted = 1;      // diagnostic: which contact points warm start their FRICTION row: 1 = all (Bullet), 0 = only points that penetrate (dist <= 0), 2 = only points above the
	                                 // surface, 3 = ground contacts only, 4 = link--link contacts only (3 reproduces the default, 4 the comparator without friction warm start: profiles/r05_a2_deviation.txt)
	int interleave = 0;              // diagnostic: 1 = each contact's normal row followed by its friction row (Integrator v1's order) instead of all normals, then all friction rows
	double limit_bias = 0.3, limit_relax = 1.0;   // btHingeConstraint::setLimit defaults (_biasFactor, _relaxationFactor)
};

struct Body {
	btRigidBody* rb = nullptr;
	double x = 0, y = 0, th = 0, vx = 0, vy = 0, w = 0;
	double inv_m = 0, inv_i = 0;
	double dvx = 0, dvy = 0, dw = 0;     // solver body: delta velocities
	double px = 0, py = 0, pw = 0;       // push / turn velocities (split impulse)
	double hx = 0, hy = 0, margin = 0;   // box half extents (margin included, as btBoxShape stores them) and margin
	double brk = 0;                      // contact breaking threshold of a manifold this body is the smaller partner of
	bool is_box = false;
	short group = 0, mask = 0;
};

struct Contact {
	int a = -1, b = -1;                  // body indices; b = -1: static ground
	const btCollisionObject* obj_b = nullptr;
	double ax = 0, ay = 0, bx = 0, by = 0;   // world points on A and on B
	double nx = 0, ny = 1;               // normal on B, towards A
	double dist = 0, mu = 0;
	uint64_t key = 0;
	double jn = 0, jt = 0;               // cached applied impulses (normal, friction)
};

struct Row {
	int a = -1, b = -1;
	double nax = 0, nay = 0, aa = 0, nbx = 0, nby = 0, ab = 0;   // Jacobian: linear + angular parts on A and B
	double dinv = 0, rhs = 0, rhs_pen = 0, cfm = 0, lo = 0, hi = 0, imp = 0, imp_push = 0;
	int normal_row = -1;                 // friction rows: index of their normal row
	double mu = 0;
	int contact = -1;
};

class Solver {
public:
	Params prm;
	std::vector<Contact> contacts;       // of the last substep (published to the dispatcher by the harness)
	long substeps = 0;
	bool diag_on = std::getenv("BSI_DIAG") != nullptr; double diag[2][5][5] = {};
	~Solver() { if (diag_on) for (int g = 0; g < 2; ++g) { std::fprintf(stderr, "BSI_DIAG %s substeps %ld\n", g ? "link-link" : "ground", substeps); const char* nm[5] = {"skipped,imp!=0 count", "skipped sum|imp|", "resolved count", "resolved sum|imp|", "skipped,imp==0 count"}; for (int k = 0; k < 5; ++k) { std::fprintf(stderr, "  %-22s", nm[k]); for (int b = 0; b < 5; ++b) std::fprintf(stderr, " %12.4g", diag[g][k][b]); std::fprintf(stderr, "\n"); } } }

	void Reset() { cache_.clear(); contacts.clear(); }
	const btRigidBody* BodyOf(int i) const { return (i >= 0 && i < static_cast<int>(bodies_.size())) ? bodies_[i].rb : nullptr; }

	// one fixed substep of size h on the stand-in's world
	void Step(btDiscreteDynamicsWorld* world, double h)
	{
		Gather(world);
		const btVector3 g = world->getGravity();
		for (Body& b : bodies_) {
			if (b.inv_m == 0) continue;
			const btVector3 lf = b.rb->getLinearFactor(), af = b.rb->getAngularFactor();
			b.vx += h * (b.rb->getTotalForce().x() * b.inv_m + g.x() * lf.x());
			b.vy += h * (b.rb->getTotalForce().y() * b.inv_m + g.y() * lf.y());
			b.w += h * b.rb->getTotalTorque().z() * b.inv_i * af.z();
			const double wmax = 0.5 * M_PI / h;   // MAX_ANGVEL (btRigidBody::integrateVelocities)
			if (std::fabs(b.w) > wmax) b.w = b.w > 0 ? wmax : -wmax;
		}
		Collide(world);
		rows_.clear(); n_joint_rows_ = 0;
		SetupJoints(world, h);
		n_joint_rows_ = static_cast<int>(rows_.size());
		SetupContacts(h);
		// split-impulse iterations, then the velocity iterations (btSequentialImpulseConstraintSolver::solveGroupCacheFriendlyIterations)
		if (prm.split_impulse) for (int it = 0; it < prm.iterations; ++it) for (size_t r = n_joint_rows_; r < rows_.size(); ++r) if (rows_[r].normal_row < 0 && rows_[r].contact >= 0) ResolvePush(rows_[r]);
		for (int it = 0; it < prm.iterations; ++it) {
			for (int r = 0; r < n_joint_rows_; ++r) Resolve(rows_[r]);
			auto friction = [&](Row& f) {
				const double tot = rows_[f.normal_row].imp;
				if (tot > 0 || !prm.friction_skip) { f.lo = -f.mu * tot; f.hi = f.mu * tot; Resolve(f); }
			};
			if (prm.interleave) {
				for (size_t r = n_joint_rows_; r < rows_.size(); ++r) { if (rows_[r].normal_row < 0) Resolve(rows_[r]); else friction(rows_[r]); }
				continue;
			}
			for (size_t r = n_joint_rows_; r < rows_.size(); ++r) if (rows_[r].normal_row < 0) Resolve(rows_[r]);
			for (size_t r = n_joint_rows_; r < rows_.size(); ++r) if (rows_[r].normal_row >= 0) friction(rows_[r]);
		}
		if (diag_on) for (size_t r = n_joint_rows_; r < rows_.size(); ++r) if (rows_[r].normal_row >= 0) {
			const Row& f = rows_[r]; const Contact& c = contacts[f.contact];
			const int g = c.b >= 0 ? 1 : 0;
			const bool skipped = !(rows_[f.normal_row].imp > 0);
			const int bin = c.dist < -1e-3 ? 0 : c.dist < 0 ? 1 : c.dist < 1e-4 ? 2 : c.dist < 1e-3 ? 3 : 4;
			if (skipped) { if (f.imp != 0) { diag[g][0][bin] += 1; diag[g][1][bin] += std::fabs(f.imp); } else diag[g][4][bin] += 1; }
			else { diag[g][2][bin] += 1; diag[g][3][bin] += std::fabs(f.imp); }
		}
		// write back: velocities, cached impulses, transforms
		for (size_t r = n_joint_rows_; r < rows_.size(); ++r) {
			const Row& row = rows_[r];
			Contact& c = contacts[row.contact];
			if (row.normal_row < 0) c.jn = row.imp; else c.jt = row.imp;
		}
		cache_.clear();
		for (const Contact& c : contacts) cache_[CacheKey(c)] = std::make_pair(c.jn, c.jt);
		for (Body& b : bodies_) {
			if (b.inv_m == 0) continue;
			b.vx += b.dvx; b.vy += b.dvy; b.w += b.dw;
			b.x += h * (b.vx + b.px); b.y += h * (b.vy + b.py);
			double wr = b.w;
			if (std::fabs(wr) * h > 0.25 * M_PI) wr = (wr > 0 ? 1 : -1) * 0.25 * M_PI / h;   // ANGULAR_MOTION_THRESHOLD (btTransformUtil::integrateTransform)
			b.th += h * (wr + prm.split_turn_erp * b.pw);
			WriteBack(b);
		}
		++substeps;
	}

private:
	std::vector<Body> bodies_;
	std::vector<Row> rows_;
	int n_joint_rows_ = 0;
	std::map<std::pair<const void*, uint64_t>, std::pair<double, double>> cache_;   // (body A, feature key) -> applied (normal, friction) impulse
	std::map<const btRigidBody*, int> index_;

	std::pair<const void*, uint64_t> CacheKey(const Contact& c) const { return std::make_pair(static_cast<const void*>(bodies_[c.a].rb), c.key ^ (c.b >= 0 ? reinterpret_cast<uint64_t>(bodies_[c.b].rb) * 0x9E3779B97F4A7C15ULL : 0)); }

	void Gather(btDiscreteDynamicsWorld* world)
	{
		bodies_.clear(); index_.clear();
		for (btRigidBody* rb : world->bodies()) {
			Body b; b.rb = rb;
			const btTransform& t = rb->getCenterOfMassTransform();
			b.x = t.getOrigin().x(); b.y = t.getOrigin().y();
			b.th = std::atan2(static_cast<double>(t.getBasis()[1][0]), static_cast<double>(t.getBasis()[0][0]));
			b.vx = rb->getLinearVelocity().x(); b.vy = rb->getLinearVelocity().y(); b.w = rb->getAngularVelocity().z();
			b.inv_m = rb->getInvMass(); b.inv_i = rb->getInvInertiaDiagLocal().z() * rb->getAngularFactor().z();
			if (const btBoxShape* box = dynamic_cast<const btBoxShape*>(rb->getCollisionShape())) {
				const btVector3 he = box->getHalfExtentsWithMargin();
				b.is_box = true; b.hx = he.x(); b.hy = he.y(); b.margin = prm.use_margin ? (prm.safe_margin ? static_cast<double>(box->getMargin()) : 0.04) : 0.0;
				b.brk = prm.relative_breaking ? static_cast<double>(box->getContactBreakingThreshold(static_cast<btScalar>(prm.breaking))) : prm.breaking;
			}
			b.group = rb->getBroadphaseHandle()->m_collisionFilterGroup; b.mask = rb->getBroadphaseHandle()->m_collisionFilterMask;
			index_[rb] = static_cast<int>(bodies_.size());
			bodies_.push_back(b);
		}
	}
	static void WriteBack(Body& b)
	{
		btTransform t = b.rb->getCenterOfMassTransform();
		const double z = t.getOrigin().z();
		t.getBasis().setEulerZYX(0, 0, static_cast<btScalar>(b.th));
		t.setOrigin(btVector3(static_cast<btScalar>(b.x), static_cast<btScalar>(b.y), static_cast<btScalar>(z)));
		b.rb->setCenterOfMassTransform(t);
		b.rb->setLinearVelocity(btVector3(static_cast<btScalar>(b.vx), static_cast<btScalar>(b.vy), 0));
		b.rb->setAngularVelocity(btVector3(0, 0, static_cast<btScalar>(b.w)));
		if (b.rb->getMotionState()) b.rb->getMotionState()->setWorldTransform(t);
	}

	// ---- ground description: polyline pieces (heightfields) and planes, in world (scaled) coordinates ----
	struct Ground { const btCollisionObject* obj = nullptr; bool plane = false; double pnx = 0, pny = 1, pc = 0; double x0 = 0, dx = 1; int n = 0; const float* h = nullptr; double mu = 0; };
	std::vector<Ground> grounds_;

	static bool Filter(short ga, short ma, short gb, short mb) { return (ga & mb) != 0 && (gb & ma) != 0; }

	void Collide(btDiscreteDynamicsWorld* world)
	{
		grounds_.clear();
		for (btRigidBody* rb : world->bodies()) {
			if (rb->getInvMass() != 0) continue;
			Ground gnd; gnd.obj = rb; gnd.mu = rb->getFriction();
			if (const btHeightfieldTerrainShape* hf = dynamic_cast<const btHeightfieldTerrainShape*>(rb->getCollisionShape())) {
				gnd.n = hf->width(); gnd.h = static_cast<const float*>(hf->data()); gnd.dx = hf->getLocalScaling().x();
				gnd.x0 = rb->getCenterOfMassTransform().getOrigin().x() - 0.5 * (gnd.n - 1) * gnd.dx;   // Bullet centres a heightfield on its local origin
				// heights: Bullet centres the height range on (min + max) / 2 and the reference places the body at that mid height: world y = stored (scaled) height
			} else if (const btStaticPlaneShape* pl = dynamic_cast<const btStaticPlaneShape*>(rb->getCollisionShape())) {
				gnd.plane = true; gnd.pnx = pl->getPlaneNormal().x(); gnd.pny = pl->getPlaneNormal().y(); gnd.pc = pl->getPlaneConstant();
				const btVector3 o = rb->getCenterOfMassTransform().getOrigin();
				gnd.pc += gnd.pnx * o.x() + gnd.pny * o.y();
			} else continue;
			grounds_.push_back(gnd);
		}
		std::vector<Contact> fresh;
		for (size_t i = 0; i < bodies_.size(); ++i) {
			const Body& A = bodies_[i];
			if (A.inv_m == 0 || !A.is_box) continue;
			for (const Ground& gnd : grounds_) {
				const btRigidBody* grb = static_cast<const btRigidBody*>(gnd.obj);
				if (!Filter(A.group, A.mask, grb->getBroadphaseHandle()->m_collisionFilterGroup, grb->getBroadphaseHandle()->m_collisionFilterMask)) continue;
				std::vector<Contact> cand;
				BoxGround(static_cast<int>(i), gnd, cand);
				Keep(cand, fresh);
			}
		}
		if (prm.link_contacts) {
			// hinge-linked pairs are excluded (addConstraint(c, true) -> disableCollisionsBetweenLinkedBodies)
			std::vector<std::pair<const btRigidBody*, const btRigidBody*>> linked;
			for (int c = 0; c < world->getNumConstraints(); ++c) { btTypedConstraint* tc = world->getConstraint(c); if (tc->hasBodyB()) linked.emplace_back(&tc->getRigidBodyA(), &tc->getRigidBodyB()); }
			for (size_t i = 0; i < bodies_.size(); ++i) for (size_t j = i + 1; j < bodies_.size(); ++j) {
				const Body& A = bodies_[i]; const Body& B = bodies_[j];
				if (A.inv_m == 0 || B.inv_m == 0 || !A.is_box || !B.is_box || !Filter(A.group, A.mask, B.group, B.mask)) continue;
				bool skip = false;
				for (const auto& l : linked) if ((l.first == A.rb && l.second == B.rb) || (l.first == B.rb && l.second == A.rb)) { skip = true; break; }
				if (skip) continue;
				// boxes must overlap in z as well (the planar characters' legs sit at different depths)
				const btBoxShape* ba = static_cast<const btBoxShape*>(A.rb->getCollisionShape()); const btBoxShape* bb = static_cast<const btBoxShape*>(B.rb->getCollisionShape());
				const double za = A.rb->getCenterOfMassTransform().getOrigin().z(), zb = B.rb->getCenterOfMassTransform().getOrigin().z();
				if (std::fabs(za - zb) > ba->getHalfExtentsWithMargin().z() + bb->getHalfExtentsWithMargin().z()) continue;
				const double rr = std::hypot(A.hx, A.hy) + std::hypot(B.hx, B.hy) + std::min(A.brk, B.brk);
				if (std::hypot(A.x - B.x, A.y - B.y) > rr) continue;
				std::vector<Contact> cand;
				BoxBox(static_cast<int>(i), static_cast<int>(j), cand);
				Keep(cand, fresh);
			}
		}
		for (Contact& c : fresh) {
			auto it = cache_.find(CacheKey(c));
			if (it != cache_.end()) { c.jn = it->second.first; c.jt = it->second.second; }
		}
		contacts.swap(fresh);
	}
	// at most max_points per pair: the deepest
	void Keep(std::vector<Contact>& cand, std::vector<Contact>& out) const
	{
		std::stable_sort(cand.begin(), cand.end(), [](const Contact& p, const Contact& q) { return p.dist < q.dist; });
		for (size_t k = 0; k < cand.size() && static_cast<int>(k) < prm.max_points; ++k) out.push_back(cand[k]);
	}
	static void Corner(const Body& b, int k, double shrink, double& x, double& y)
	{
		const double sx = (k & 1) ? 1 : -1, sy = (k & 2) ? 1 : -1;
		const double lx = sx * (b.hx - shrink), ly = sy * (b.hy - shrink), c = std::cos(b.th), s = std::sin(b.th);
		x = b.x + c * lx - s * ly; y = b.y + s * lx + c * ly;
	}
	void BoxGround(int ia, const Ground& gnd, std::vector<Contact>& out) const
	{
		const Body& A = bodies_[ia];
		const double m = A.margin;
		for (int k = 0; k < 4; ++k) {
			double px, py; Corner(A, k, m, px, py);
			Contact c; c.a = ia; c.b = -1; c.obj_b = gnd.obj; c.mu = A.rb->getFriction() * gnd.mu; c.key = static_cast<uint64_t>(k);
			if (gnd.plane) {
				const double d = gnd.pnx * px + gnd.pny * py - gnd.pc;
				c.nx = gnd.pnx; c.ny = gnd.pny; c.dist = d - m;
				c.bx = px - d * c.nx; c.by = py - d * c.ny;
			} else {
				const double u = (px - gnd.x0) / gnd.dx;
				if (u < 0 || u > gnd.n - 1) continue;
				int i = static_cast<int>(u); if (i > gnd.n - 2) i = gnd.n - 2;
				const double x0 = gnd.x0 + i * gnd.dx, y0 = gnd.h[i], y1 = gnd.h[i + 1];
				double tx = gnd.dx, ty = y1 - y0; const double tl = std::hypot(tx, ty); tx /= tl; ty /= tl;
				c.nx = -ty; c.ny = tx;   // upward normal of the cell
				const double d = c.nx * (px - x0) + c.ny * (py - y0);
				c.dist = d - m;
				c.bx = px - d * c.nx; c.by = py - d * c.ny;
			}
			if (c.dist > A.brk) continue;
			c.ax = px - m * c.nx; c.ay = py - m * c.ny;   // the point of the rounded box closest to the surface
			out.push_back(c);
		}
		if (gnd.plane || !prm.vertex_contacts) return;
		// terrain vertices against the box (a crest poking into a face between two corners)
		const double r = std::hypot(A.hx, A.hy) + A.brk;
		int i0 = static_cast<int>(std::floor((A.x - r - gnd.x0) / gnd.dx)), i1 = static_cast<int>(std::ceil((A.x + r - gnd.x0) / gnd.dx));
		i0 = std::max(i0, 0); i1 = std::min(i1, gnd.n - 1);
		const double cth = std::cos(A.th), sth = std::sin(A.th);
		for (int i = i0; i <= i1; ++i) {
			// only convex crests can touch a face first
			if (i > 0 && i < gnd.n - 1 && 2.0 * gnd.h[i] <= gnd.h[i - 1] + gnd.h[i + 1]) continue;
			const double vx = gnd.x0 + i * gnd.dx, vy = gnd.h[i];
			const double lx = cth * (vx - A.x) + sth * (vy - A.y), ly = -sth * (vx - A.x) + cth * (vy - A.y);   // vertex in the box frame
			const double ex = A.hx - m, ey = A.hy - m;
			const double qx = std::max(-ex, std::min(ex, lx)), qy = std::max(-ey, std::min(ey, ly));          // closest point of the core box
			double nlx, nly, d;
			if (qx == lx && qy == ly) {   // inside the core: out through the nearest face
				const double dxp = ex - std::fabs(lx), dyp = ey - std::fabs(ly);
				if (dxp < dyp) { nlx = lx > 0 ? -1 : 1; nly = 0; d = -dxp; } else { nlx = 0; nly = ly > 0 ? -1 : 1; d = -dyp; }
			} else {
				const double ddx = qx - lx, ddy = qy - ly; d = std::hypot(ddx, ddy); nlx = ddx / d; nly = ddy / d;
			}
			Contact c; c.a = ia; c.b = -1; c.obj_b = gnd.obj; c.mu = A.rb->getFriction() * gnd.mu; c.key = 0x100u + static_cast<uint64_t>(i) + (reinterpret_cast<uint64_t>(gnd.h) << 20);
			c.nx = cth * nlx - sth * nly; c.ny = sth * nlx + cth * nly;   // from the vertex (ground) towards the box
			if (c.ny < 0.2) continue;                                    // a vertex can only push upwards-ish (it is ground)
			c.dist = d - m;
			if (c.dist > A.brk) continue;
			c.bx = vx; c.by = vy; c.ax = vx + c.dist * c.nx; c.ay = vy + c.dist * c.ny;
			out.push_back(c);
		}
	}
	// corners of the margin-shrunk box P against box Q (and vice versa); normal on B (= Q side as stored) towards A
	void BoxBox(int ia, int ib, std::vector<Contact>& out) const
	{
		for (int side = 0; side < 2; ++side) {
			const int ip = side ? ib : ia, iq = side ? ia : ib;
			const Body& P = bodies_[ip]; const Body& Q = bodies_[iq];
			const double m = P.margin + Q.margin;
			const double cq = std::cos(Q.th), sq = std::sin(Q.th), ex = Q.hx - Q.margin, ey = Q.hy - Q.margin;
			for (int k = 0; k < 4; ++k) {
				double px, py; Corner(P, k, P.margin, px, py);
				const double lx = cq * (px - Q.x) + sq * (py - Q.y), ly = -sq * (px - Q.x) + cq * (py - Q.y);
				const double qx = std::max(-ex, std::min(ex, lx)), qy = std::max(-ey, std::min(ey, ly));
				double nlx, nly, d;
				if (qx == lx && qy == ly) {
					const double dxp = ex - std::fabs(lx), dyp = ey - std::fabs(ly);
					if (dxp < dyp) { nlx = lx > 0 ? 1 : -1; nly = 0; d = -dxp; } else { nlx = 0; nly = ly > 0 ? 1 : -1; d = -dyp; }
				} else { const double ddx = lx - qx, ddy = ly - qy; d = std::hypot(ddx, ddy); nlx = ddx / d; nly = ddy / d; }
				const double dist = d - m;
				if (dist > std::min(P.brk, Q.brk)) continue;
				// normal from Q towards P in world coordinates
				const double nwx = cq * nlx - sq * nly, nwy = sq * nlx + cq * nly;
				Contact c; c.a = ia; c.b = ib; c.obj_b = bodies_[ib].rb; c.mu = P.rb->getFriction() * Q.rb->getFriction(); c.key = 0x40u + static_cast<uint64_t>(side * 4 + k);
				c.dist = dist;
				const double qwx = Q.x + cq * qx - sq * qy, qwy = Q.y + sq * qx + cq * qy;   // closest point on Q's core
				if (side == 0) { c.nx = nwx; c.ny = nwy; c.ax = px - P.margin * nwx; c.ay = py - P.margin * nwy; c.bx = qwx + Q.margin * nwx; c.by = qwy + Q.margin * nwy; }
				else { c.nx = -nwx; c.ny = -nwy; c.bx = px - P.margin * nwx; c.by = py - P.margin * nwy; c.ax = qwx + Q.margin * nwx; c.ay = qwy + Q.margin * nwy; }
				out.push_back(c);
			}
		}
	}

	// ---- rows ----
	void Finish(Row& r)
	{
		const Body& A = bodies_[r.a];
		double s = A.inv_m * (r.nax * r.nax + r.nay * r.nay) + A.inv_i * r.aa * r.aa;
		if (r.b >= 0) { const Body& B = bodies_[r.b]; s += B.inv_m * (r.nbx * r.nbx + r.nby * r.nby) + B.inv_i * r.ab * r.ab; }
		r.dinv = s > 1e-300 ? 1.0 / s : 0.0;
	}
	double RelVel(const Row& r) const
	{
		const Body& A = bodies_[r.a];
		double v = r.nax * A.vx + r.nay * A.vy + r.aa * A.w;
		if (r.b >= 0) { const Body& B = bodies_[r.b]; v += r.nbx * B.vx + r.nby * B.vy + r.ab * B.w; }
		return v;
	}
	void Apply(const Row& r, double d)
	{
		Body& A = bodies_[r.a];
		A.dvx += r.nax * A.inv_m * d; A.dvy += r.nay * A.inv_m * d; A.dw += r.aa * A.inv_i * d;
		if (r.b >= 0) { Body& B = bodies_[r.b]; B.dvx += r.nbx * B.inv_m * d; B.dvy += r.nby * B.inv_m * d; B.dw += r.ab * B.inv_i * d; }
	}
	// btSequentialImpulseConstraintSolver::resolveSingleConstraintRowGeneric
	void Resolve(Row& r)
	{
		const Body& A = bodies_[r.a];
		double dv = r.nax * A.dvx + r.nay * A.dvy + r.aa * A.dw;
		if (r.b >= 0) { const Body& B = bodies_[r.b]; dv += r.nbx * B.dvx + r.nby * B.dvy + r.ab * B.dw; }
		double d = r.rhs - r.imp * r.cfm - dv * r.dinv;
		const double sum = r.imp + d;
		if (sum < r.lo) { d = r.lo - r.imp; r.imp = r.lo; } else if (sum > r.hi) { d = r.hi - r.imp; r.imp = r.hi; } else r.imp = sum;
		Apply(r, d);
	}
	// resolveSplitPenetrationImpulseCacheFriendly
	void ResolvePush(Row& r)
	{
		if (r.rhs_pen == 0) return;
		Body& A = bodies_[r.a];
		double dv = r.nax * A.px + r.nay * A.py + r.aa * A.pw;
		if (r.b >= 0) { const Body& B = bodies_[r.b]; dv += r.nbx * B.px + r.nby * B.py + r.ab * B.pw; }
		double d = r.rhs_pen - r.imp_push * r.cfm - dv * r.dinv;
		const double sum = r.imp_push + d;
		if (sum < r.lo) { d = r.lo - r.imp_push; r.imp_push = r.lo; } else r.imp_push = sum;
		A.px += r.nax * A.inv_m * d; A.py += r.nay * A.inv_m * d; A.pw += r.aa * A.inv_i * d;
		if (r.b >= 0) { Body& B = bodies_[r.b]; B.px += r.nbx * B.inv_m * d; B.py += r.nby * B.inv_m * d; B.pw += r.ab * B.inv_i * d; }
	}

	void SetupJoints(btDiscreteDynamicsWorld* world, double h)
	{
		const double inf = 1e300;
		for (int ci = 0; ci < world->getNumConstraints(); ++ci) {
			btHingeConstraint* hc = dynamic_cast<btHingeConstraint*>(world->getConstraint(ci));
			if (!hc || !hc->isEnabled()) continue;
			const int ia = index_.at(&hc->getRigidBodyA());
			const int ib = hc->hasBodyB() ? index_.at(&hc->getRigidBodyB()) : -1;
			const Body& A = bodies_[ia];
			const btVector3 pa = hc->getAFrame().getOrigin();
			const double ca = std::cos(A.th), sa = std::sin(A.th);
			const double rax = ca * pa.x() - sa * pa.y(), ray = sa * pa.x() + ca * pa.y();
			double rbx = 0, rby = 0, pbx, pby;
			if (ib >= 0) {
				const Body& B = bodies_[ib];
				const btVector3 pb = hc->getBFrame().getOrigin();
				const double cb = std::cos(B.th), sb = std::sin(B.th);
				rbx = cb * pb.x() - sb * pb.y(); rby = sb * pb.x() + cb * pb.y();
				pbx = B.x + rbx; pby = B.y + rby;
			} else { pbx = hc->getBFrame().getOrigin().x(); pby = hc->getBFrame().getOrigin().y(); }
			const double k = prm.erp / h;
			// point-to-point rows along world x and y: J_A = [e | r_A x e], J_B = -[e | r_B x e], rhs = k (pivot_B - pivot_A) . e
			for (int ax = 0; ax < 2; ++ax) {
				Row r; r.a = ia; r.b = ib; r.lo = -inf; r.hi = inf;
				if (ax == 0) { r.nax = 1; r.aa = -ray; r.nbx = -1; r.ab = rby; } else { r.nay = 1; r.aa = rax; r.nby = -1; r.ab = -rbx; }
				Finish(r);
				const double err = ax == 0 ? (pbx - (A.x + rax)) : (pby - (A.y + ray));
				r.rhs = (k * err - RelVel(r)) * r.dinv;
				rows_.push_back(r);
			}
			// angular limit (btAngularLimit::test on the hinge angle; angle rate = w_A - w_B)
			const double lo = hc->getLowerLimit(), hi = hc->getUpperLimit();
			if (lo <= hi && ib >= 0) {
				const double center = 0.5 * (lo + hi), half = 0.5 * (hi - lo);
				double dev = hc->getHingeAngle() - center;
				dev = std::fmod(dev + M_PI, 2 * M_PI); if (dev < 0) dev += 2 * M_PI; dev -= M_PI;   // btNormalizeAngle
				double corr = 0;
				if (dev < -half) corr = -(dev + half); else if (dev > half) corr = half - dev;
				if (corr != 0) {
					Row r; r.a = ia; r.b = ib; r.aa = 1; r.ab = -1;
					Finish(r);
					double c = k * corr;
					const double vel = RelVel(r);
					if (corr > 0) { r.lo = 0; r.hi = inf; if (vel < 0) { const double nc = -prm.limit_relax * vel; if (nc > c) c = nc; } }
					else { r.lo = -inf; r.hi = 0; if (vel > 0) { const double nc = -prm.limit_relax * vel; if (nc < c) c = nc; } }
					c *= prm.limit_bias;
					r.rhs = (c - vel) * r.dinv;
					rows_.push_back(r);
				}
			}
		}
	}

	void SetupContacts(double h)
	{
		for (size_t ci = 0; ci < contacts.size(); ++ci) {
			Contact& c = contacts[ci];
			const Body& A = bodies_[c.a];
			const double rax = c.ax - A.x, ray = c.ay - A.y;
			double rbx = 0, rby = 0;
			if (c.b >= 0) { rbx = c.bx - bodies_[c.b].x; rby = c.by - bodies_[c.b].y; }
			Row n; n.a = c.a; n.b = c.b; n.contact = static_cast<int>(ci);
			n.nax = c.nx; n.nay = c.ny; n.aa = rax * c.ny - ray * c.nx;
			n.nbx = -c.nx; n.nby = -c.ny; n.ab = -(rbx * c.ny - rby * c.nx);
			Finish(n);
			n.lo = 0; n.hi = 1e10;
			const double rel = RelVel(n);
			double pos_err = 0, vel_err = -rel;   // restitution 0
			const double pen = c.dist;
			double erp = prm.erp2;
			if (!prm.split_impulse || pen > prm.split_threshold) erp = prm.erp;
			if (pen > 0) vel_err -= pen / h; else pos_err = -pen * erp / h;
			if (!prm.split_impulse || pen > prm.split_threshold) { n.rhs = (pos_err + vel_err) * n.dinv; n.rhs_pen = 0; }
			else { n.rhs = vel_err * n.dinv; n.rhs_pen = pos_err * n.dinv; }
			n.imp = prm.warmstarting ? c.jn * prm.warmstart_factor : 0.0;
			if (n.imp != 0) Apply(n, n.imp);
			const int ni = static_cast<int>(rows_.size());
			rows_.push_back(n);
			// friction: along the relative tangential velocity, else the plane-space vector (-n_y, n_x)
			const double vax = A.vx - A.w * ray, vay = A.vy + A.w * rax;
			double vbx = 0, vby = 0;
			if (c.b >= 0) { const Body& B = bodies_[c.b]; vbx = B.vx - B.w * rby; vby = B.vy + B.w * rbx; }
			const double rvx = vax - vbx, rvy = vay - vby, rn = rvx * c.nx + rvy * c.ny;
			double tx = rvx - rn * c.nx, ty = rvy - rn * c.ny;
			const double t2 = tx * tx + ty * ty;
			if (t2 > SIMD_EPSILON && prm.friction_dir) { const double tl = std::sqrt(t2); tx /= tl; ty /= tl; } else { tx = -c.ny; ty = c.nx; }
			Row f; f.a = c.a; f.b = c.b; f.contact = static_cast<int>(ci); f.normal_row = ni; f.mu = c.mu;
			f.nax = tx; f.nay = ty; f.aa = rax * ty - ray * tx; f.nbx = -tx; f.nby = -ty; f.ab = -(rbx * ty - rby * tx);
			Finish(f);
			f.rhs = -RelVel(f) * f.dinv;
			f.lo = 0; f.hi = 0;   // set from the normal impulse inside the iterations
			f.imp = (prm.warmstarting && prm.friction_warmstart && (prm.friction_ws_lifted == 1 || (prm.friction_ws_lifted == 0 && !(c.dist > 0)) || (prm.friction_ws_lifted == 2 && c.dist > 0) || (prm.friction_ws_lifted == 3 && c.b < 0) || (prm.friction_ws_lifted == 4 && c.b >= 0))) ? c.jt * prm.warmstart_factor : 0.0;
			if (f.imp != 0) Apply(f, f.imp);
			rows_.push_back(f);
		}
	}
};

}  // namespace bsi
