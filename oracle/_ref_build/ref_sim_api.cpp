// ref_sim_api.cpp -- TEST INFRASTRUCTURE. C entry points over the reference's OWN scenario / character / controller / ground sources, compiled
// unchanged from /root/reference by oracle/_ref_build/Makefile into oracle/_ref/libref_sim.so (never linked or loaded by the product).
//
// What is the reference's and what is ours in that library:
//   reference (unchanged translation units): scenarios/{Scenario, ScenarioSimChar, ScenarioExp, ScenarioExpMACE, ScenarioExpCacla, ScenarioPoliEval}.cpp,
//       sim/{World, SimObj, SimBox, SimCapsule, SimPlane, Joint, ContactManager, Perturb, PerturbManager, Ground, GroundFlat, GroundVar2D, TerrainGen2D,
//       SimCharacter, SimCharSoftFall, SimDog, SimRaptor, Controller, CharController, NNController, TerrainRLCharController, PDController,
//       ImpPDController, DogController(+Q, Cacla, MACE), GoatControllerMACE, RaptorController(+Q, Cacla, MACE), BaseController{Q, Cacla, MACE}, SpAlg,
//       RBDModel, RBDUtil}.cpp, anim/{KinTree, Character}.cpp, learning/ExpTuple.cpp, util/{Rand, ArgParser, FileUtil, MathUtil, JsonUtil}.cpp
//   ours: stand-ins for the absent third-party headers (stubs/: Eigen, jsoncpp; stubs_bullet/: Bullet as a state container WITHOUT a physics step),
//       shadow/learning/*.h (the Caffe-backed network / trainer classes reduced to what the rollout side touches: a forward callback, the
//       normalisation formula, a dozen index helpers, flag enums) and this file.
// So everything the reference does AROUND Bullet's stepSimulation and Caffe's Forward runs as the reference wrote it: the step loop and its
// ordering, generalised <-> maximal coordinate conversion, torque accumulation / clamping / application, contact flags from manifold
// distances, fall and stumble logic, ground windows and sampling, terrain features, FSM controllers, implicit PD, gravity compensation,
// virtual forces, action selection, rewards, tuples. The physics step is a hook: tests/test_reference_sim.py installs one that advances
// the state with the ORACLE's documented integrator and writes it back through cSimCharacter::SetPose / SetVel, then compares what the
// reference computes from that state with what the oracle (and, through the oracle, the HIP kernel) computes.
// every standard / stand-in header first, with normal access control ...
#include <algorithm>
#include <fstream>
#include <functional>
#include <iostream>
#include <map>
#include <memory>
#include <mutex>
#include <random>
#include <sstream>
#include <stack>
#include <string>
#include <vector>
#include <json/json.h>
#include "Eigen/Dense"
#include "btBulletDynamicsCommon.h"
#include <BulletCollision/CollisionShapes/btHeightfieldTerrainShape.h>
// ... then the reference's headers with `protected` opened, in THIS translation unit only: the harness reads controller internals (current
// action, PD targets, counters, the tuple buffer). The reference's own translation units are compiled untouched.
#define protected public
#include "scenarios/ScenarioExpMACE.h"
#include "scenarios/ScenarioPoliEval.h"
#include "scenarios/ScenarioSimChar.h"
#include "sim/DogController.h"
#include "sim/GroundVar2D.h"
#include "sim/RaptorController.h"
#include "sim/SimCharSoftFall.h"
#undef protected

#include <unistd.h>
#include <cstring>
#include <memory>

#include "../or_bullet_si.h"   // TEST INFRASTRUCTURE: the maximal-coordinate sequential-impulse step (Bullet 2.8x's published algorithm restated) for ref_scn_use_bullet_si

namespace {
typedef void (*nn_forward_fn)(const double* x_norm, double* y_norm);
typedef void (*step_hook_fn)(void* user, double dt, int substeps);
typedef void (*post_substep_fn)(void* user, double dt);

struct RefScn {
	std::shared_ptr<cScenarioSimChar> scn;
	int kind = 0;
	step_hook_fn hook = nullptr; void* hook_user = nullptr;
	post_substep_fn post = nullptr; void* post_user = nullptr;
	std::shared_ptr<bsi::Solver> si;
};
cTerrainRLCharController* Ctrl(RefScn* s) { return dynamic_cast<cTerrainRLCharController*>(s->scn->GetCharacter()->GetController().get()); }
cImpPDController* ImpPD(RefScn* s)
{
	cCharController* c = s->scn->GetCharacter()->GetController().get();
	if (cDogController* d = dynamic_cast<cDogController*>(c)) return &d->mImpPDCtrl;
	if (cRaptorController* r = dynamic_cast<cRaptorController*>(c)) return &r->mImpPDCtrl;
	return nullptr;
}
}  // namespace

extern "C" {

// sizes and raw forward (normalised input -> normalised output) of the network every controller of this process "loads" (shadow/learning/NeuralNet.h)
void ref_nn_config(int in_size, int out_size, nn_forward_fn fwd)
{
	cNeuralNet::tHarness& h = cNeuralNet::Harness();
	h.mInputSize = in_size; h.mOutputSize = out_size;
	h.mForward = [fwd, out_size](const Eigen::VectorXd& x, Eigen::VectorXd& y) { y.resize(out_size); fwd(x.data(), y.data()); };
}

// kind: 0 cScenarioSimChar, 1 cScenarioExpMACE, 2 cScenarioPoliEval, 3 cScenarioExp (what -scenario= train builds its pool from: the Q controllers). argv = "-key= value" tokens (command line first, then -arg_file= is appended like
// optimizer/Main.cpp:19-32 does); cwd = the directory relative data paths resolve against (the reference is run from its repo root).
// global_seed seeds cMathUtil's global RNG before Init (the reference seeds it from the clock, util/Rand.cpp:8).
void* ref_scn_create(int kind, char** argv, int argc, const char* cwd, unsigned long global_seed)
{
	char old[4096]; if (!getcwd(old, sizeof(old))) old[0] = '\0';
	if (cwd && chdir(cwd) != 0) return nullptr;
	cArgParser parser(argv, argc);
	std::string arg_file;
	if (parser.ParseString("arg_file", arg_file)) parser.AppendArgs(arg_file);
	cMathUtil::SeedRand(global_seed);
	RefScn* s = new RefScn(); s->kind = kind;
	if (kind == 1) s->scn = std::shared_ptr<cScenarioSimChar>(new cScenarioExpMACE());
	else if (kind == 2) s->scn = std::shared_ptr<cScenarioSimChar>(new cScenarioPoliEval());
	else if (kind == 3) s->scn = std::shared_ptr<cScenarioSimChar>(new cScenarioExp());
	else s->scn = std::shared_ptr<cScenarioSimChar>(new cScenarioSimChar());
	s->scn->ParseArgs(parser);
	s->scn->Init();
	if (old[0]) { if (chdir(old) != 0) {} }
	if (!s->scn->GetCharacter()) { delete s; return nullptr; }
	return s;
}
void ref_scn_free(void* h) { delete static_cast<RefScn*>(h); }

// cScenarioPoliEval::SetRandSeed + Reset ("rebuild ground", optimizer/scenarios/OptScenarioPoliEval.cpp:152-156): the ground's own cRand restarts from `seed`
void ref_scn_seed_ground_and_reset(void* h, unsigned long seed)
{
	RefScn* s = static_cast<RefScn*>(h);
	if (cGroundVar2D* g = dynamic_cast<cGroundVar2D*>(s->scn->GetGround().get())) g->SeedRand(seed);
	s->scn->Reset();
}
void ref_scn_reset(void* h) { static_cast<RefScn*>(h)->scn->Reset(); }

// the physics hook: called from inside cWorld::Update in place of Bullet's integration, once per cWorld::Update (= once per env-step)
void ref_scn_set_step_hook(void* h, step_hook_fn fn, void* user)
{
	RefScn* s = static_cast<RefScn*>(h);
	s->hook = fn; s->hook_user = user;
	if (s->si) return;   // the sequential-impulse integrator is installed: it calls the hook as an OBSERVER in front of its substeps (ref_scn_use_bullet_si)
	s->scn->GetWorld()->GetInternalWorld()->m_stepHook = [s](btScalar dt, int substeps, btScalar) { if (s->hook) s->hook(s->hook_user, dt, substeps); };
}
// called after every iteration of the loop at scenarios/ScenarioSimChar.cpp:162-173 (after the scenario's own PostSubstepUpdate)
void ref_scn_set_post_substep(void* h, post_substep_fn fn, void* user)
{
	RefScn* s = static_cast<RefScn*>(h);
	s->post = fn; s->post_user = user;
	cScenarioSimChar* scn = s->scn.get();
	// the scenario's own callback (cScenarioExp / cScenarioPoliEval register theirs in their constructors via virtual PostSubstepUpdate, not through this slot)
	scn->SetPostSubstepCallback([s](double dt) { if (s->post) s->post(s->post_user, dt); });
}
void ref_scn_update(void* h, double dt) { static_cast<RefScn*>(h)->scn->Update(dt); }

// Physics = oracle/or_bullet_si.h: the stand-in world's stepSimulation(dt, n, dt / n) runs n sequential-impulse substeps on the reference's rigid bodies,
// hinges and ground shapes (Bullet 2.8x's published algorithm and defaults), and the contact points it worked with are handed to the dispatcher as
// manifolds, so that the reference's cContactManager derives its flags from them as it would from Bullet's. opts (NULL = defaults), in order:
// iterations, erp, erp2, split_impulse, split_threshold, warmstarting, warmstart_factor, breaking, max_points, use_margin, link_contacts, safe_margin, relative_breaking, vertex_contacts, friction_warmstart, friction_skip, friction_dir, interleave, friction_ws_lifted (the last four: diagnostics, see bsi::Params)
void ref_scn_use_bullet_si(void* h, const double* opts, int n_opts)
{
	RefScn* s = static_cast<RefScn*>(h);
	s->si = std::make_shared<bsi::Solver>();
	bsi::Params& p = s->si->prm;
	auto opt = [&](int i, double def) { return (opts && i < n_opts) ? opts[i] : def; };
	p.iterations = static_cast<int>(opt(0, p.iterations)); p.erp = opt(1, p.erp); p.erp2 = opt(2, p.erp2); p.split_impulse = static_cast<int>(opt(3, p.split_impulse));
	p.split_threshold = opt(4, p.split_threshold); p.warmstarting = static_cast<int>(opt(5, p.warmstarting)); p.warmstart_factor = opt(6, p.warmstart_factor);
	p.breaking = opt(7, p.breaking); p.max_points = static_cast<int>(opt(8, p.max_points)); p.use_margin = static_cast<int>(opt(9, p.use_margin)); p.link_contacts = static_cast<int>(opt(10, p.link_contacts));
	p.safe_margin = static_cast<int>(opt(11, p.safe_margin)); p.relative_breaking = static_cast<int>(opt(12, p.relative_breaking));
	p.vertex_contacts = static_cast<int>(opt(13, p.vertex_contacts)); p.friction_warmstart = static_cast<int>(opt(14, p.friction_warmstart));
	p.friction_skip = static_cast<int>(opt(15, p.friction_skip)); p.friction_dir = static_cast<int>(opt(16, p.friction_dir)); p.interleave = static_cast<int>(opt(17, p.interleave)); p.friction_ws_lifted = static_cast<int>(opt(18, p.friction_ws_lifted));
	btDiscreteDynamicsWorld* world = s->scn->GetWorld()->GetInternalWorld().get();
	world->getConstraintSolver()->m_resetHook = [s]() { if (s->si) s->si->Reset(); };
	world->m_stepHook = [s, world](btScalar dt, int substeps, btScalar fixed) {
		if (s->hook) s->hook(s->hook_user, dt, substeps);   // observer: the state as the previous env-step left it (controller update included)
		for (int i = 0; i < substeps; ++i) s->si->Step(world, fixed);
		btDispatcher* d = world->getDispatcher();
		d->m_manifolds.clear();
		for (const bsi::Contact& c : s->si->contacts) {
			btPersistentManifold m;
			btManifoldPoint pt; pt.m_distance1 = static_cast<btScalar>(c.dist);
			pt.m_positionWorldOnA = btVector3(static_cast<btScalar>(c.ax), static_cast<btScalar>(c.ay), 0); pt.m_positionWorldOnB = btVector3(static_cast<btScalar>(c.bx), static_cast<btScalar>(c.by), 0);
			pt.m_normalWorldOnB = btVector3(static_cast<btScalar>(c.nx), static_cast<btScalar>(c.ny), 0);
			m.m_points.push_back(pt);
			m.m_body0 = s->si->BodyOf(c.a); m.m_body1 = c.obj_b;
			d->m_manifolds.push_back(m);
		}
	};
}
// contacts of the last substep: n, then per contact link index of A (-1 if not a character part), of B (-1 = ground), distance (scaled units), applied normal impulse
int ref_scn_si_contacts(void* h, int* link_a, int* link_b, double* dist, double* jn, int cap)
{
	RefScn* s = static_cast<RefScn*>(h);
	if (!s->si) return 0;
	const auto& c = s->scn->GetCharacter();
	auto link_of = [&](const btCollisionObject* o) { for (int j = 0; j < c->GetNumBodyParts(); ++j) if (c->GetBodyPart(j)->GetRigidBody().get() == o) return j; return -1; };
	int n = 0;
	for (const bsi::Contact& ct : s->si->contacts) {
		if (n >= cap) break;
		link_a[n] = link_of(s->si->BodyOf(ct.a)); link_b[n] = ct.b >= 0 ? link_of(ct.obj_b) : -1; dist[n] = ct.dist; jn[n] = ct.jn; ++n;
	}
	return n;
}

void ref_scn_dims(void* h, int* num_joints, int* num_dof, int* poli_state, int* poli_action, int* num_params)
{
	RefScn* s = static_cast<RefScn*>(h);
	const auto& c = s->scn->GetCharacter();
	*num_joints = c->GetNumJoints(); *num_dof = c->GetNumDof();
	cTerrainRLCharController* ctrl = Ctrl(s);
	*poli_state = ctrl ? ctrl->GetPoliStateSize() : 0; *poli_action = ctrl ? ctrl->GetPoliActionSize() : 0; *num_params = ctrl ? ctrl->GetNumParams() : 0;
}
// cSimCharacter::BuildPose / BuildVel (sim/SimCharacter.cpp:166-225): maximal (rigid bodies) -> generalised coordinates
void ref_scn_get_pose_vel(void* h, double* q, double* qd)
{
	RefScn* s = static_cast<RefScn*>(h);
	Eigen::VectorXd pose, vel;
	s->scn->GetCharacter()->BuildPose(pose); s->scn->GetCharacter()->BuildVel(vel);
	for (int i = 0; i < static_cast<int>(pose.size()); ++i) { q[i] = pose[i]; qd[i] = vel[i]; }
}
// cSimCharacter::SetPose / SetVel (sim/SimCharacter.cpp:665-683, 227-315): generalised -> maximal
void ref_scn_set_pose_vel(void* h, const double* q, const double* qd)
{
	RefScn* s = static_cast<RefScn*>(h);
	const int D = s->scn->GetCharacter()->GetNumDof();
	Eigen::VectorXd pose(D), vel(D);
	for (int i = 0; i < D; ++i) { pose[i] = q[i]; vel[i] = qd[i]; }
	s->scn->GetCharacter()->SetPose(pose); s->scn->GetCharacter()->SetVel(vel);
}
// per body part: world COM position (x, y), angle about z, linear velocity (x, y), angular velocity z -- straight from the rigid bodies
void ref_scn_get_bodies(void* h, double* pos2, double* angle, double* vel2, double* omega)
{
	RefScn* s = static_cast<RefScn*>(h);
	const auto& c = s->scn->GetCharacter();
	for (int j = 0; j < c->GetNumBodyParts(); ++j) {
		const auto& p = c->GetBodyPart(j);
		tVector x = p->GetPos(), v = p->GetLinearVelocity(), w = p->GetAngularVelocity(), axis; double th;
		p->GetRotation(axis, th);
		pos2[2 * j] = x[0]; pos2[2 * j + 1] = x[1]; angle[j] = th * axis[2]; vel2[2 * j] = v[0]; vel2[2 * j + 1] = v[1]; omega[j] = w[2];
	}
}
// joint torques as accumulated by the controller (cJoint::AddTorque) and clamped by cJoint::ApplyTorque (sim/Joint.cpp:171-201, 257-264) in the last
// cSimCharacter::Update: z component per joint (0 for the root); plus the torque Bullet received per body (sum over its joints, world frame, x scale^2)
void ref_scn_get_torques(void* h, double* joint_tau, double* body_torque_z)
{
	RefScn* s = static_cast<RefScn*>(h);
	const auto& c = s->scn->GetCharacter();
	for (int j = 0; j < c->GetNumJoints(); ++j) {
		joint_tau[j] = c->GetJoint(j).IsValid() ? c->GetJoint(j).GetTorque()[2] : 0.0;
		if (body_torque_z) { const btRigidBody* rb = c->GetBodyPart(j)->GetRigidBody().get(); body_torque_z[j] = rb ? rb->getTotalTorque().z() : 0.0; }
	}
}
// Replace the dispatcher's manifolds: one manifold per (link, ground) with `n_pts[k]` points at the given signed distances (metres; the harness
// multiplies by the world scale as Bullet would report them). cContactManager::Update then derives the per-link flags itself (distance <= 0.001).
void ref_scn_set_contacts(void* h, int n, const int* link, const double* dist)
{
	RefScn* s = static_cast<RefScn*>(h);
	const auto& c = s->scn->GetCharacter();
	btDispatcher* d = s->scn->GetWorld()->GetInternalWorld()->getDispatcher();
	const double scale = s->scn->GetWorld()->GetScale();
	d->m_manifolds.clear();
	cGroundVar2D* g = dynamic_cast<cGroundVar2D*>(s->scn->GetGround().get());
	const btCollisionObject* ground_obj = nullptr;
	if (g) ground_obj = g->GetMinSegment()->GetRigidBody().get();
	for (int k = 0; k < n; ++k) {
		btPersistentManifold m;
		m.m_body0 = c->GetBodyPart(link[k])->GetRigidBody().get();
		m.m_body1 = ground_obj;
		btManifoldPoint p; p.m_distance1 = static_cast<btScalar>(dist[k] * scale);
		m.m_points.push_back(p);
		d->m_manifolds.push_back(m);
	}
}
// appends link--link manifolds (body0 = part a, body1 = part b) behind the ground manifolds of ref_scn_set_contacts: what Bullet's narrowphase reports for
// links of one collision group. The reference's cContactManager must not turn them into contact flags (parts are registered with filter
// eContactFlagEnvironment, scenarios/ScenarioSimChar.cpp:321; cContactManager::IsValidContact, sim/ContactManager.cpp:169-175)
void ref_scn_add_pair_contacts(void* h, int n, const int* link_a, const int* link_b, const double* dist)
{
	RefScn* s = static_cast<RefScn*>(h);
	const auto& c = s->scn->GetCharacter();
	btDispatcher* d = s->scn->GetWorld()->GetInternalWorld()->getDispatcher();
	const double scale = s->scn->GetWorld()->GetScale();
	for (int k = 0; k < n; ++k) {
		btPersistentManifold m;
		m.m_body0 = c->GetBodyPart(link_a[k])->GetRigidBody().get();
		m.m_body1 = c->GetBodyPart(link_b[k])->GetRigidBody().get();
		btManifoldPoint p; p.m_distance1 = static_cast<btScalar>(dist[k] * scale);
		m.m_points.push_back(p);
		d->m_manifolds.push_back(m);
	}
}
void ref_scn_get_contact_flags(void* h, int* flags)
{
	RefScn* s = static_cast<RefScn*>(h);
	const auto& c = s->scn->GetCharacter();
	for (int j = 0; j < c->GetNumBodyParts(); ++j) flags[j] = c->IsInContact(j) ? 1 : 0;
}
// bit 0 fallen, bit 1 stumbled, bit 2 new cycle, bits 8.. FSM state
unsigned ref_scn_get_flags(void* h)
{
	RefScn* s = static_cast<RefScn*>(h);
	const auto& c = s->scn->GetCharacter();
	const auto& ctrl = c->GetController();
	return (c->HasFallen() ? 1u : 0u) | (c->HasStumbled() ? 2u : 0u) | (ctrl->IsNewCycle() ? 4u : 0u) | (static_cast<unsigned>(ctrl->GetState()) << 8);
}
// controller internals: FSM state, phase, current action id + parameters (mCurrAction), PD target angles, PD active flags
void ref_scn_get_ctrl(void* h, int* state, double* phase, int* action_id, double* params, double* pd_targets, int* pd_active)
{
	RefScn* s = static_cast<RefScn*>(h);
	const auto& c = s->scn->GetCharacter();
	cTerrainRLCharController* ctrl = Ctrl(s);
	*state = ctrl->GetState(); *phase = ctrl->GetPhase(); *action_id = ctrl->GetCurrActionID();
	for (int i = 0; i < static_cast<int>(ctrl->mCurrAction.mParams.size()); ++i) params[i] = ctrl->mCurrAction.mParams[i];
	cImpPDController* pd = ImpPD(s);
	for (int j = 0; j < c->GetNumJoints(); ++j) {
		const bool valid = pd && pd->IsValidPDCtrl(j);
		pd_targets[j] = valid ? pd->GetTargetTheta(j) : 0.0;
		if (pd_active) pd_active[j] = (valid && pd->GetPDCtrl(j).IsActive()) ? 1 : 0;
	}
}
// cNNController::RecordPoliState / RecordPoliAction (sim/TerrainRLCharController.cpp:120-123; sim/BaseControllerMACE.cpp:58-68)
void ref_scn_get_poli_state(void* h, double* out) { Eigen::VectorXd v; Ctrl(static_cast<RefScn*>(h))->RecordPoliState(v); for (int i = 0; i < static_cast<int>(v.size()); ++i) out[i] = v[i]; }
void ref_scn_get_poli_action(void* h, double* out) { Eigen::VectorXd v; Ctrl(static_cast<RefScn*>(h))->RecordPoliAction(v); for (int i = 0; i < static_cast<int>(v.size()); ++i) out[i] = v[i]; }
double ref_scn_calc_reward(void* h)
{
	cNNController* c = dynamic_cast<cNNController*>(static_cast<RefScn*>(h)->scn->GetCharacter()->GetController().get());
	return c ? c->CalcReward() : 0.0;
}
void ref_scn_build_output_offset_scale(void* h, double* off, double* scale)
{
	Eigen::VectorXd o, sc; Ctrl(static_cast<RefScn*>(h))->BuildNNOutputOffsetScale(o, sc);
	for (int i = 0; i < static_cast<int>(o.size()); ++i) { off[i] = o[i]; scale[i] = sc[i]; }
}
// the controller's network normalisers (what cNeuralNet::LoadScale would read from <model>_scale.txt)
void ref_scn_set_net_scale(void* h, const double* in_off, const double* in_scale, const double* out_off, const double* out_scale)
{
	cNNController* c = dynamic_cast<cNNController*>(static_cast<RefScn*>(h)->scn->GetCharacter()->GetController().get());
	if (!c) return;
	cNeuralNet& net = c->GetNet();
	const int ni = net.GetInputSize(), no = net.GetOutputSize();
	Eigen::VectorXd io(ni), is(ni), oo(no), os(no);
	for (int i = 0; i < ni; ++i) { io[i] = in_off[i]; is[i] = in_scale[i]; }
	for (int i = 0; i < no; ++i) { oo[i] = out_off[i]; os[i] = out_scale[i]; }
	net.SetInputOffsetScale(io, is); net.SetOutputOffsetScale(oo, os);
}
// cCharController::CommandAction (sim/CharController.cpp; what cScenarioExp::Reset does with a random id, scenarios/ScenarioExp.cpp:63-73)
// replaces whatever is queued (cScenarioExp::Reset queues a random first action drawn from the clock-seeded global RNG, scenarios/ScenarioExp.cpp:62-72)
void ref_scn_command_action(void* h, int action_id)
{
	cCharController* c = static_cast<RefScn*>(h)->scn->GetCharacter()->GetController().get();
	if (cDogController* d = dynamic_cast<cDogController*>(c)) d->ClearCommands();
	if (cRaptorController* r = dynamic_cast<cRaptorController*>(c)) r->ClearCommands();
	c->CommandAction(action_id);
}
void ref_scn_enable_explore(void* h, int enable) { static_cast<RefScn*>(h)->scn->GetCharacter()->GetController()->EnableExp(enable != 0); }
// cGround::SampleHeight (sim/GroundVar2D.cpp:98-128) and cGroundVar2D::CalcGridCoord (:559-619: the continuous grid coordinate whose integer part is
// the cell index the north star wants bit-exact); valid = the sample lies inside the two-segment window
double ref_scn_sample_ground(void* h, double x, int* valid, double* grid_coord)
{
	RefScn* s = static_cast<RefScn*>(h);
	bool v = false;
	const double y = s->scn->GetGround()->SampleHeight(tVector(x, 0, 0, 0), v);
	if (valid) *valid = v ? 1 : 0;
	if (grid_coord) { cGroundVar2D* g = dynamic_cast<cGroundVar2D*>(s->scn->GetGround().get()); *grid_coord = g ? g->CalcGridCoord(tVector(x, 0, 0, 0))[0] : 0.0; }
	return y;
}
// the two segments of the window in logical order (0 = min segment): vertex count, min x, heights
int ref_scn_ground_segment(void* h, int slot, float* out, int cap, double* min_x, double* max_x)
{
	RefScn* s = static_cast<RefScn*>(h);
	cGroundVar2D* g = dynamic_cast<cGroundVar2D*>(s->scn->GetGround().get());
	if (!g) return 0;
	const auto& seg = g->GetSegment(slot);
	const int n = static_cast<int>(seg->mData.size()) / std::max(1, seg->GetGridLength());
	for (int i = 0; i < n && i < cap; ++i) out[i] = seg->mData[i];
	*min_x = seg->GetMinX(); *max_x = seg->GetMaxX();
	return n;
}
double ref_scn_time(void* h) { return static_cast<RefScn*>(h)->scn->mTime; }
void ref_scn_com(void* h, double* com2, double* com_vel2)
{
	const auto& c = static_cast<RefScn*>(h)->scn->GetCharacter();
	tVector p = c->CalcCOM(), v = c->CalcCOMVel();
	com2[0] = p[0]; com2[1] = p[1]; com_vel2[0] = v[0]; com_vel2[1] = v[1];
}
// cScenarioExp: tuples completed so far (IsTupleBufferFull / GetTuples / ResetTupleBuffer); rows [reward | stateBeg | action | stateEnd], flags
int ref_scn_drain_tuples(void* h, double* rows, unsigned* flags, int cap)
{
	RefScn* s = static_cast<RefScn*>(h);
	cScenarioExp* e = dynamic_cast<cScenarioExp*>(s->scn.get());
	if (!e) return 0;
	const int n_avail = e->mTupleCount;
	int n = 0;
	for (; n < n_avail && n < cap; ++n) {
		const tExpTuple& t = e->mTupleBuffer[n];
		const int S = static_cast<int>(t.mStateBeg.size()), A = static_cast<int>(t.mAction.size()), W = 1 + 2 * S + A;
		double* r = rows + static_cast<size_t>(n) * W;
		r[0] = t.mReward;
		for (int i = 0; i < S; ++i) { r[1 + i] = t.mStateBeg[i]; r[1 + S + A + i] = t.mStateEnd[i]; }
		for (int i = 0; i < A; ++i) r[1 + S + i] = t.mAction[i];
		flags[n] = t.mFlags;
	}
	e->ResetTupleBuffer();
	return n;
}
// cScenarioPoliEval bookkeeping
void ref_scn_eval_stats(void* h, double* avg_dist, int* episodes, int* cycles, int* dist_log_n, double* dist_log, int cap)
{
	cScenarioPoliEval* e = dynamic_cast<cScenarioPoliEval*>(static_cast<RefScn*>(h)->scn.get());
	if (!e) { *avg_dist = 0; *episodes = 0; *cycles = 0; *dist_log_n = 0; return; }
	*avg_dist = e->GetAvgDist(); *episodes = e->GetNumEpisodes(); *cycles = e->GetNumCycles();
	const std::vector<double>& log = e->GetDistLog();
	*dist_log_n = static_cast<int>(log.size());
	for (int i = 0; i < *dist_log_n && i < cap; ++i) dist_log[i] = log[i];
}

}  // extern "C"
