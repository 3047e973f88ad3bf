// ref_api.cpp -- TEST INFRASTRUCTURE. C entry points over the REFERENCE'S OWN sources, compiled unchanged from /root/reference
// by oracle/_ref_build/Makefile into oracle/_ref/libref_core.so (never linked or loaded by the product).
//
// What is the reference's and what is ours in that library:
//   reference (unchanged translation units): util/Rand.cpp, util/ArgParser.cpp, util/FileUtil.cpp, util/MathUtil.cpp, util/JsonUtil.cpp,
//       sim/TerrainGen2D.cpp, sim/SpAlg.cpp, sim/RBDModel.cpp, sim/RBDUtil.cpp, anim/KinTree.cpp
//   ours: this file (argument marshalling only) and the stand-ins for the two absent third-party headers the sources include
//       (stubs/Eigen/Dense = eager dense linear algebra, stubs/json/json.h = JSON reader). Bullet and Caffe are not needed by these files.
// tests/test_reference_pin.py checks oracle/ (the restatement) and the product's host code against these entry points, and
// tests/golden/make_ref_golden.py freezes their outputs as fixtures for boxes without /root/reference.
#include <cstdint>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>

#include "anim/KinTree.h"
#include "sim/RBDModel.h"
#include "sim/RBDUtil.h"
#include "sim/SpAlg.h"
#include "sim/TerrainGen2D.h"
#include "util/ArgParser.h"
#include "util/MathUtil.h"
#include "util/Rand.h"

namespace {
struct RefChar {
	Eigen::MatrixXd joint_mat, body_defs;
	cRBDModel model;
	bool inited = false;
};
int copy_str(const std::string& s, char* out, int cap)
{
	if (out && cap > 0) { std::strncpy(out, s.c_str(), cap - 1); out[cap - 1] = '\0'; }
	return static_cast<int>(s.size());
}
}  // namespace

extern "C" {

// ---- util/Rand.cpp ------------------------------------------------------------------------------------------------
// kind: 0 RandDouble(a, b), 1 RandInt(int a, int b), 2 RandDoubleNorm(a, b), 3 FlipCoin(a), 4 RandSign
void ref_rand_stream(unsigned long seed, int kind, double a, double b, int n, double* out)
{
	cRand r; r.Seed(seed);
	for (int i = 0; i < n; ++i) {
		switch (kind) {
		case 0: out[i] = r.RandDouble(a, b); break;
		case 1: out[i] = r.RandInt(static_cast<int>(a), static_cast<int>(b)); break;
		case 2: out[i] = r.RandDoubleNorm(a, b); break;
		case 3: out[i] = r.FlipCoin(a) ? 1 : 0; break;
		default: out[i] = r.RandSign(); break;
		}
	}
}

// ---- util/ArgParser.cpp -------------------------------------------------------------------------------------------
void* ref_args_load(const char* file)
{
	cArgParser* p = new cArgParser();
	p->AppendArgs(std::string(file));
	return p;
}
// command line first, then the file appended: optimizer/Main.cpp:19-32
void* ref_args_load_argv(char** argv, int argc, const char* file)
{
	cArgParser* p = new cArgParser(argv, argc);
	if (file) p->AppendArgs(std::string(file));
	return p;
}
void ref_args_free(void* h) { delete static_cast<cArgParser*>(h); }
int ref_args_count(void* h) { return static_cast<cArgParser*>(h)->GetNumArgs(); }
int ref_args_string(void* h, const char* key, char* out, int cap)
{
	std::string s;
	if (!static_cast<cArgParser*>(h)->ParseString(key, s)) return -1;
	return copy_str(s, out, cap);
}
int ref_args_int(void* h, const char* key, int* out) { return static_cast<cArgParser*>(h)->ParseInt(key, *out) ? 1 : 0; }
int ref_args_double(void* h, const char* key, double* out) { return static_cast<cArgParser*>(h)->ParseDouble(key, *out) ? 1 : 0; }
int ref_args_bool(void* h, const char* key, int* out) { bool b = false; bool ok = static_cast<cArgParser*>(h)->ParseBool(key, b); *out = b ? 1 : 0; return ok ? 1 : 0; }

// ---- sim/TerrainGen2D.cpp -----------------------------------------------------------------------------------------
int ref_terrain_num_params() { return cTerrainGen2D::eParamsMax; }
int ref_terrain_param_name(int i, char* out, int cap) { return copy_str(cTerrainGen2D::gParamDefs[i].mName, out, cap); }
void ref_terrain_default_params(double* out)
{
	cTerrainGen2D::tParams p = cTerrainGen2D::GetDefaultParams();
	for (int i = 0; i < cTerrainGen2D::eParamsMax; ++i) out[i] = p[i];
}
double ref_terrain_vert_spacing() { return cTerrainGen2D::gVertSpacing; }
// terrain file -> type name + every parameter vector of its "Params" array (scenarios/ScenarioSimChar.cpp:670-706 reads the same keys)
int ref_terrain_load_file(const char* file, char* type_out, int cap, double* params_out, int max_sets)
{
	std::ifstream f(file);
	if (!f.is_open()) return -1;
	Json::Value root; Json::Reader reader;
	if (!reader.parse(f, root)) return -2;
	copy_str(root[cTerrainGen2D::gTypeKey].asString(), type_out, cap);
	const Json::Value& ps = root[cTerrainGen2D::gParamsKey];
	int n = 0;
	if (ps.isArray()) for (; n < static_cast<int>(ps.size()) && n < max_sets; ++n) {
		Eigen::VectorXd v;
		cTerrainGen2D::LoadParams(ps.get(n, 0), v);
		for (int i = 0; i < cTerrainGen2D::eParamsMax; ++i) params_out[n * cTerrainGen2D::eParamsMax + i] = v[i];
	}
	return n;
}
// one strip: ParseType(name) -> GetTerrainFunc -> func(width, params, rand seeded with `seed`, data); returns the vertex count, the
// function's return value (width added) through out_width. `prefix` vertices of height prefix_h are placed in the vector first (the
// generators continue an existing profile from its last vertex, sim/GroundVar2D.cpp:318-322)
int ref_terrain_build(const char* type_name, const double* params40, unsigned long seed, double width, int prefix, float prefix_h, float* out, int cap, double* out_width)
{
	cTerrainGen2D::eType type = cTerrainGen2D::eTypeFlat;
	cTerrainGen2D::ParseType(type_name, type);
	cTerrainGen2D::tTerrainFunc func = cTerrainGen2D::GetTerrainFunc(type);
	cTerrainGen2D::tParams p;
	for (int i = 0; i < cTerrainGen2D::eParamsMax; ++i) p[i] = params40[i];
	cRand r; r.Seed(seed);
	std::vector<float> data(prefix, prefix_h);
	const double w = func(width, p, r, data);
	if (out_width) *out_width = w;
	const int n = static_cast<int>(data.size());
	for (int i = 0; i < n && i < cap; ++i) out[i] = data[i];
	return n;
}

// ---- anim/KinTree.cpp + sim/RBDModel.cpp + sim/RBDUtil.cpp + sim/SpAlg.cpp -------------------------------------------
// cCharacter::LoadSkeleton (anim/Character.cpp:264-271) = cKinTree::Load(root["Skeleton"]); cSimCharacter::LoadBodyDefs = cKinTree::LoadBodyDefs(file)
void* ref_char_load(const char* char_file)
{
	std::ifstream f(char_file);
	if (!f.is_open()) return nullptr;
	Json::Value root; Json::Reader reader;
	if (!reader.parse(f, root)) return nullptr;
	RefChar* c = new RefChar();
	if (root["Skeleton"].isNull() || !cKinTree::Load(root["Skeleton"], c->joint_mat)) { delete c; return nullptr; }
	if (!cKinTree::LoadBodyDefs(char_file, c->body_defs)) { delete c; return nullptr; }
	c->model.Init(c->joint_mat, c->body_defs, gGravity);   // sim/DogController.cpp:179-208 (InitRBDModel)
	c->inited = true;
	return c;
}
void ref_char_free(void* h) { delete static_cast<RefChar*>(h); }
void ref_char_dims(void* h, int* num_joints, int* num_dof, double* total_mass)
{
	RefChar* c = static_cast<RefChar*>(h);
	*num_joints = cKinTree::GetNumJoints(c->joint_mat); *num_dof = cKinTree::GetNumDof(c->joint_mat);
	*total_mass = cKinTree::CalcTotalMass(c->body_defs);
}
// per joint: [type, parent, param offset, param size, attach x, y, z, lim low, lim high]; per body: [mass, attach x, y, z, theta, size x, y, z]
void ref_char_tables(void* h, double* joints9, double* bodies8)
{
	RefChar* c = static_cast<RefChar*>(h);
	const int n = cKinTree::GetNumJoints(c->joint_mat);
	for (int j = 0; j < n; ++j) {
		double* o = joints9 + 9 * j;
		o[0] = cKinTree::GetJointType(c->joint_mat, j); o[1] = cKinTree::GetParent(c->joint_mat, j);
		o[2] = cKinTree::GetParamOffset(c->joint_mat, j); o[3] = cKinTree::GetParamSize(c->joint_mat, j);
		tVector a = cKinTree::GetScaledAttachPt(c->joint_mat, j);
		o[4] = a[0]; o[5] = a[1]; o[6] = a[2];
		o[7] = cKinTree::GetJointLimLow(c->joint_mat, j); o[8] = cKinTree::GetJointLimHigh(c->joint_mat, j);
		double* b = bodies8 + 8 * j;
		tVector ba = cKinTree::GetBodyAttachPt(c->body_defs, j), bs = cKinTree::GetBodySize(c->body_defs, j);
		tVector axis; double theta; cKinTree::GetBodyRotation(c->body_defs, j, axis, theta);
		b[0] = cKinTree::GetBodyMass(c->body_defs, j); b[1] = ba[0]; b[2] = ba[1]; b[3] = ba[2]; b[4] = theta; b[5] = bs[0]; b[6] = bs[1]; b[7] = bs[2];
	}
}
// cRBDModel::Update(pose, vel) (sim/RBDModel.cpp:39-55), then the quantities the controllers read:
//   H [D*D row-major] = GetMassMat (cRBDUtil::BuildMassMat), C [D] = GetBiasForce (BuildBiasForce, BuildCjPlanar as shipped),
//   grav [D] = cRBDUtil::CalcGravityForce, J [6*D row-major] = cRBDUtil::BuildJacobian, com / com_vel [3] = cRBDUtil::CalcCoM,
//   joint_pos [L*3] = cRBDModel::CalcJointWorldPos. Any output may be NULL.
void ref_rbd(void* h, const double* q, const double* qd, double* H, double* C, double* grav, double* J, double* com, double* com_vel, double* joint_pos)
{
	RefChar* c = static_cast<RefChar*>(h);
	const int D = c->model.GetNumDof(), L = c->model.GetNumJoints();
	Eigen::VectorXd pose(D), vel(D);
	for (int i = 0; i < D; ++i) { pose[i] = q[i]; vel[i] = qd[i]; }
	c->model.Update(pose, vel);
	if (H) { const Eigen::MatrixXd& m = c->model.GetMassMat(); for (int i = 0; i < D; ++i) for (int k = 0; k < D; ++k) H[i * D + k] = m(i, k); }
	if (C) { const Eigen::VectorXd& b = c->model.GetBiasForce(); for (int i = 0; i < D; ++i) C[i] = b[i]; }
	if (grav) { Eigen::VectorXd g; cRBDUtil::CalcGravityForce(c->model, g); for (int i = 0; i < D; ++i) grav[i] = g[i]; }
	if (J) { Eigen::MatrixXd jac; cRBDUtil::BuildJacobian(c->model, jac); for (int r = 0; r < 6; ++r) for (int k = 0; k < D; ++k) J[r * D + k] = jac(r, k); }
	if (com || com_vel) {
		tVector p, v; cRBDUtil::CalcCoM(c->model, p, v);
		for (int k = 0; k < 3; ++k) { if (com) com[k] = p[k]; if (com_vel) com_vel[k] = v[k]; }
	}
	if (joint_pos) for (int j = 0; j < L; ++j) { tVector p = c->model.CalcJointWorldPos(j); for (int k = 0; k < 3; ++k) joint_pos[3 * j + k] = p[k]; }
}
// cWorld::tJointParams::mRefTheta of every joint, evaluated with the reference's own primitives in the order cSimCharacter::BuildConstraints uses them
// (sim/SimCharacter.cpp:838-853; that translation unit itself needs Bullet): BodyJointTrans, ParentChildTrans at the zero pose, InvRigidMat, RotMatToAxisAngle
void ref_char_ref_theta(void* h, double* out)
{
	RefChar* c = static_cast<RefChar*>(h);
	const int L = cKinTree::GetNumJoints(c->joint_mat);
	Eigen::VectorXd default_pose = Eigen::VectorXd::Zero(cKinTree::GetNumDof(c->joint_mat));
	for (int j = 0; j < L; ++j) {
		out[j] = 0;
		const int parent_id = cKinTree::GetParent(c->joint_mat, j);
		if (parent_id == cKinTree::gInvalidJointID) continue;
		tMatrix curr_body_mat = cKinTree::BodyJointTrans(c->body_defs, j);
		tMatrix child_parent_mat = cKinTree::ParentChildTrans(c->joint_mat, default_pose, j);
		tMatrix parent_body_mat = cMathUtil::InvRigidMat(cKinTree::BodyJointTrans(c->body_defs, parent_id));
		tMatrix child_parent_body_mat = cMathUtil::InvRigidMat(parent_body_mat) * child_parent_mat * curr_body_mat;
		tVector ref_axis; double ref_theta;
		cMathUtil::RotMatToAxisAngle(child_parent_body_mat, ref_axis, ref_theta);
		out[j] = -ref_theta;
	}
}
// inverse dynamics for a given acceleration: cRBDUtil::SolveInvDyna (sim/RBDUtil.cpp:4-84) after Update(pose, vel)
void ref_inv_dyna(void* h, const double* q, const double* qd, const double* acc, double* tau)
{
	RefChar* c = static_cast<RefChar*>(h);
	const int D = c->model.GetNumDof();
	Eigen::VectorXd pose(D), vel(D), a(D), t;
	for (int i = 0; i < D; ++i) { pose[i] = q[i]; vel[i] = qd[i]; a[i] = acc[i]; }
	c->model.Update(pose, vel);
	cRBDUtil::SolveInvDyna(c->model, a, t);
	for (int i = 0; i < D; ++i) tau[i] = t[i];
}
// kinematics by cKinTree alone (what cCharacter / cKinCharacter and the body-part bookkeeping use): body COM world position
// (CalcBodyPartPos, anim/KinTree.cpp:286-295), body world angle about z (BodyWorldTrans), joint world position and angle
void ref_kin_bodies(void* h, const double* q, double* body_pos3, double* body_theta, double* joint_pos3, double* joint_theta)
{
	RefChar* c = static_cast<RefChar*>(h);
	const int D = cKinTree::GetNumDof(c->joint_mat), L = cKinTree::GetNumJoints(c->joint_mat);
	Eigen::VectorXd pose(D);
	for (int i = 0; i < D; ++i) pose[i] = q[i];
	for (int j = 0; j < L; ++j) {
		tVector p = cKinTree::CalcBodyPartPos(c->joint_mat, pose, c->body_defs, j);
		tMatrix m = cKinTree::BodyWorldTrans(c->joint_mat, pose, c->body_defs, j);
		tVector jp = cKinTree::CalcJointWorldPos(c->joint_mat, pose, j);
		tVector axis; double th; cKinTree::CalcJointWorldTheta(c->joint_mat, pose, j, axis, th);
		for (int k = 0; k < 3; ++k) { if (body_pos3) body_pos3[3 * j + k] = p[k]; if (joint_pos3) joint_pos3[3 * j + k] = jp[k]; }
		if (body_theta) body_theta[j] = std::atan2(m(1, 0), m(0, 0));
		if (joint_theta) joint_theta[j] = th * axis[2];
	}
}
// world velocity of a point attached to joint `parent_id`'s frame: cKinTree::CalcWorldVel (used by cSimCharacter::SetVel, sim/SimCharacter.cpp:227-315)
void ref_kin_world_vel(void* h, const double* q, const double* qd, int parent_id, const double* attach3, double* out3)
{
	RefChar* c = static_cast<RefChar*>(h);
	const int D = cKinTree::GetNumDof(c->joint_mat);
	Eigen::VectorXd pose(D), vel(D);
	for (int i = 0; i < D; ++i) { pose[i] = q[i]; vel[i] = qd[i]; }
	tVector v = cKinTree::CalcWorldVel(c->joint_mat, pose, vel, parent_id, tVector(attach3[0], attach3[1], attach3[2], 0));
	for (int k = 0; k < 3; ++k) out3[k] = v[k];
}

}  // extern "C"
