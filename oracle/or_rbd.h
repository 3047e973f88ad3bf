// ORACLE (test infrastructure, NOT product code).
// CPU fp64 restatement of the reference's kinematic tree and rigid-body-dynamics model used by the
// controllers (implicit PD, gravity compensation, virtual forces).
//
// Follows (file:line relative to /root/reference):
//   anim/KinTree.cpp:726-757   GetParamOffset/GetParamSize (planar root = 3 params, revolute = 1)
//   anim/KinTree.cpp:1025-1048 ChildParentTrans            :1117-1149 Revolute / Planar child->parent matrices
//   sim/RBDModel.cpp:39-55     cRBDModel::Update (subspace, child-parent mats, world transforms, H, C)
//   sim/RBDUtil.cpp:4-84       SolveInvDyna (RNEA)         :110-176 BuildMassMat (CRBA)
//   sim/RBDUtil.cpp:250-269    BuildJacobian               :562-583 BuildMomentInertiaBox
//   sim/RBDUtil.cpp:614-623    BuildInertiaSpatialMat      :625-649 CalcWorldJointTransforms
//   sim/RBDUtil.cpp:742-772    BuildJointSubspacePlanar    :809-836 BuildCjPlanar (quirk kept, see fix_cj)
//   sim/RBDUtil.cpp:850-895    CalcGravityForce
#pragma once
#include "or_math.h"
#include "or_model.h"
#include <vector>

namespace orc {

struct KinTree {
	const OrcModel* M = nullptr;
	int L = 0, D = 0;
	int off[ORC_MAXL];
	int dim[ORC_MAXL];

	void Init(const OrcModel* m)
	{
		M = m; L = m->L;
		int o = 0;
		for (int j = 0; j < L; ++j) {
			int sz = 0;
			switch (m->joint_type[j]) { case 0: sz = 1; break; case 1: sz = 3; break; case 2: sz = 1; break; default: sz = (m->parent[j] < 0) ? 3 : 0; }
			off[j] = o; dim[j] = sz; o += sz;
		}
		D = o;
	}
	double JointTheta(const double* pose, int j) const { return (M->joint_type[j] == 1) ? pose[off[j] + 2] : pose[off[j]]; }
	// anim/KinTree.cpp:1117-1149 (root attach is zeroed by PostProcessJointMat, :1016-1019)
	M4 ChildParentMat(const double* pose, int j) const
	{
		if (M->parent[j] < 0) {
			V3 offset{pose[off[j]], pose[off[j] + 1], 0};
			return TranslateMat(offset) * RotateMatZ(JointTheta(pose, j));
		}
		V3 a{M->attach[j][0], M->attach[j][1], M->attach[j][2]};
		return TranslateMat(a) * RotateMatZ(JointTheta(pose, j));
	}
	// anim/KinTree.cpp:1057-1068
	M4 JointWorldMat(const double* pose, int j) const
	{
		M4 m;
		int c = j;
		while (c >= 0) { m = ChildParentMat(pose, c) * m; c = M->parent[c]; }
		return m;
	}
	// anim/KinTree.cpp:1087-1098
	M4 BodyJointMat(int j) const
	{
		V3 a{M->body_attach[j][0], M->body_attach[j][1], M->body_attach[j][2]};
		return TranslateMat(a) * RotateMatZ(M->body_theta[j]);
	}
};

struct RBDModel {
	KinTree kt;
	const OrcModel* M = nullptr;
	int L = 0, D = 0;
	V3 gravity{0, -9.8, 0};  // util/MathUtil.h:19
	std::vector<double> pose, vel;
	SV S[ORC_MAXD];                 // joint subspace columns (6 x D)
	M4 child_parent[ORC_MAXL];
	SpTrans world_joint[ORC_MAXL];  // mSpWorldJointTransArr
	SM Ispatial[ORC_MAXL];          // constant per-link spatial inertia (reference recomputes it per call)
	double H[ORC_MAXD][ORC_MAXD];
	double C[ORC_MAXD];
	SV J[ORC_MAXD];                 // world-frame Jacobian columns (cRBDUtil::BuildJacobian)

	void Init(const OrcModel* m)
	{
		M = m; kt.Init(m); L = kt.L; D = kt.D;
		pose.assign(D, 0); vel.assign(D, 0);
		for (int j = 0; j < L; ++j) Ispatial[j] = BuildInertiaSpatialMat(j);
	}
	// sim/RBDUtil.cpp:562-583 + 614-623
	SM BuildInertiaSpatialMat(int j) const
	{
		double mass = M->body_mass[j];
		double sx = M->body_size[j][0], sy = M->body_size[j][1], sz = M->body_size[j][2];
		SM Ic = SM::Zero();
		Ic.m[0][0] = mass / 12.0 * (sy * sy + sz * sz);
		Ic.m[1][1] = mass / 12.0 * (sx * sx + sz * sz);
		Ic.m[2][2] = mass / 12.0 * (sx * sx + sy * sy);
		Ic.m[3][3] = Ic.m[4][4] = Ic.m[5][5] = mass;
		SpTrans X = BuildTrans(V3{-M->body_attach[j][0], -M->body_attach[j][1], -M->body_attach[j][2]});
		return BuildSpatialMatF(X) * Ic * BuildSpatialMatM(InvTrans(X));
	}
	SpTrans SpChildParent(int j) const { return MatToTrans(child_parent[j]); }
	SpTrans SpParentChild(int j) const { return MatToTrans(InvRigidMat(child_parent[j])); }
	V3 JointWorldPos(int j) const { return world_joint[j].r; }

	// cRBDModel::Update, sim/RBDModel.cpp:39-55
	void Update(const double* q, const double* qd, bool fix_cj = false)
	{
		for (int i = 0; i < D; ++i) { pose[i] = q[i]; vel[i] = qd[i]; }
		// joint subspaces: sim/RBDUtil.cpp:728-772
		for (int j = 0; j < L; ++j) {
			int o = kt.off[j];
			if (M->joint_type[j] == 1) {
				double theta = kt.JointTheta(q, j);
				double c = std::cos(theta), s = std::sin(theta);
				// E = RotateMat(z, -theta): [[c, s], [-s, c]];  S.block(3,0,2,2) = E.block(0,0,2,2);  S(2,2) = 1
				S[o + 0] = SV{{0, 0, 0}, {c, -s, 0}};
				S[o + 1] = SV{{0, 0, 0}, {s, c, 0}};
				S[o + 2] = SV{{0, 0, 1}, {0, 0, 0}};
			} else {
				S[o] = SV{{0, 0, 1}, {0, 0, 0}};
			}
		}
		for (int j = 0; j < L; ++j) child_parent[j] = kt.ChildParentMat(q, j);
		// sim/RBDUtil.cpp:625-649
		for (int j = 0; j < L; ++j) {
			SpTrans world_parent;
			if (M->parent[j] >= 0) world_parent = world_joint[M->parent[j]];
			world_joint[j] = CompTrans(SpParentChild(j), world_parent);
		}
		BuildMassMat();
		std::vector<double> acc(D, 0.0);
		SolveInvDyna(acc.data(), C, fix_cj);
		// sim/RBDUtil.cpp:250-269
		for (int j = 0; j < L; ++j) for (int k = 0; k < kt.dim[j]; ++k) J[kt.off[j] + k] = ApplyInvTransM(world_joint[j], S[kt.off[j] + k]);
	}

	// sim/RBDUtil.cpp:110-176 (composite rigid body algorithm, 6x6 blocks exactly as the reference)
	void BuildMassMat()
	{
		for (int i = 0; i < D; ++i) for (int k = 0; k < D; ++k) H[i][k] = 0;
		SM Is[ORC_MAXL], cpF[ORC_MAXL], pcM[ORC_MAXL];
		for (int j = 0; j < L; ++j) {
			Is[j] = Ispatial[j];
			SpTrans cpt = SpChildParent(j);
			cpF[j] = BuildSpatialMatF(cpt);
			pcM[j] = BuildSpatialMatM(InvTrans(cpt));
		}
		for (int j = L - 1; j >= 0; --j) {
			const SM& curr_I = Is[j];
			int parent = M->parent[j];
			if (parent >= 0) AddTo(Is[parent], cpF[j] * curr_I * pcM[j]);
			int o = kt.off[j], dm = kt.dim[j];
			SV F[3];
			for (int a = 0; a < dm; ++a) F[a] = curr_I * S[o + a];
			for (int a = 0; a < dm; ++a) for (int b = 0; b < dm; ++b) H[o + a][o + b] = dot(S[o + a], F[b]);
			int curr = j;
			while (M->parent[curr] >= 0) {
				for (int a = 0; a < dm; ++a) F[a] = cpF[curr] * F[a];
				curr = M->parent[curr];
				int co = kt.off[curr], cd = kt.dim[curr];
				for (int a = 0; a < dm; ++a) for (int b = 0; b < cd; ++b) { double v = dot(F[a], S[co + b]); H[o + a][co + b] = v; H[co + b][o + a] = v; }
			}
		}
	}

	// sim/RBDUtil.cpp:809-836. fix_cj=false reproduces the reference verbatim: theta/offset are read from
	// q_dot and s = cos(theta). fix_cj=true is the textbook d/dt(S) q_dot used by the
	// documented integrator that stands in for Bullet.
	SV BuildCjRoot(const double* q, const double* qd, bool fix_cj) const
	{
		if (!fix_cj) {
			double x = qd[0], y = qd[1], theta = qd[2];
			double c = std::cos(theta);
			double s = std::cos(theta);
			return SV{{0, 0, 0}, {(-s * x + c * y) * theta, (-c * x - s * y) * theta, 0}};
		}
		double x = qd[0], y = qd[1], w = qd[2];
		double c = std::cos(q[2]), s = std::sin(q[2]);
		return SV{{0, 0, 0}, {(-s * x + c * y) * w, (-c * x - s * y) * w, 0}};
	}

	// sim/RBDUtil.cpp:4-84 (RNEA; acc0 = -gravity)
	void SolveInvDyna(const double* acc, double* out_tau, bool fix_cj) const
	{
		SV vel0{{0, 0, 0}, {0, 0, 0}};
		SV acc0{{0, 0, 0}, -gravity};
		SV vels[ORC_MAXL], accs[ORC_MAXL], fs[ORC_MAXL];
		for (int j = 0; j < L; ++j) {
			SpTrans parent_child = SpParentChild(j);
			int o = kt.off[j], dm = kt.dim[j];
			SV cj{{0, 0, 0}, {0, 0, 0}};
			if (M->joint_type[j] == 1 && M->parent[j] < 0) cj = BuildCjRoot(pose.data(), vel.data(), fix_cj);
			SV vj{{0, 0, 0}, {0, 0, 0}}, Sddq{{0, 0, 0}, {0, 0, 0}};
			for (int a = 0; a < dm; ++a) { vj = vj + vel[o + a] * S[o + a]; Sddq = Sddq + acc[o + a] * S[o + a]; }
			SV vel_p = vel0, acc_p = acc0;
			if (M->parent[j] >= 0) { vel_p = vels[M->parent[j]]; acc_p = accs[M->parent[j]]; }
			SV curr_vel = ApplyTransM(parent_child, vel_p) + vj;
			SV curr_acc = ApplyTransM(parent_child, acc_p) + Sddq + cj + CrossM(curr_vel, vj);
			SV curr_f = Ispatial[j] * curr_acc + CrossF(curr_vel, Ispatial[j] * curr_vel);
			vels[j] = curr_vel; accs[j] = curr_acc; fs[j] = curr_f;
		}
		for (int i = 0; i < D; ++i) out_tau[i] = 0;
		for (int j = L - 1; j >= 0; --j) {
			int o = kt.off[j], dm = kt.dim[j];
			for (int a = 0; a < dm; ++a) out_tau[o + a] = dot(S[o + a], fs[j]);
			if (M->parent[j] >= 0) fs[M->parent[j]] = fs[M->parent[j]] + ApplyTransF(SpChildParent(j), fs[j]);
		}
	}

	// sim/RBDUtil.cpp:850-895
	void CalcGravityForce(double* out) const
	{
		SV acc0{{0, 0, 0}, gravity};
		SV fs[ORC_MAXL];
		for (int j = 0; j < L; ++j) fs[j] = Ispatial[j] * ApplyTransM(world_joint[j], acc0);
		for (int i = 0; i < D; ++i) out[i] = 0;
		for (int j = L - 1; j >= 0; --j) {
			int o = kt.off[j], dm = kt.dim[j];
			for (int a = 0; a < dm; ++a) out[o + a] = dot(S[o + a], fs[j]);
			if (M->parent[j] >= 0) fs[M->parent[j]] = fs[M->parent[j]] + ApplyTransF(SpChildParent(j), fs[j]);
		}
	}
};

}  // namespace orc
