// dtrl_types.h -- plain-old-data layouts shared by the host loader, the HIP kernels and the lane-loop unit-test
// build of the kernel math. Everything here is trivially copyable so the host can hipMemcpy it verbatim.
//
// Data layout in HBM (one record per env, records contiguous; a 64-lane wavefront owns one env and its lanes read
// consecutive elements of that env's record, so every load of a record field is a coalesced burst):
//   EnvState   persistent simulation + controller state              (~4.5 KB, fp64)
//   GroundRec  the env's two sliding heightfield segments (float)    (~4.2 KB)
//   policy-state / tuple scratch and the net's outputs live in separate per-env slabs (DevBuffers in the engine); the forward's activations stay in LDS
#pragma once
#include <cstdint>

namespace dtrl {

// Arithmetic type of the whole frame kernel. The shipped library (libdtrl.so) computes in fp64: the reference's controller / RBD / network precision.
// -DDTRL_REAL_F32 builds the OPT-IN fp32 library (libdtrl_f32.so, `-physics_precision= f32`; Bullet's own state is float: premake4.lua:115-124): same source, same
// operation order, half the registers and half the LDS per env -> more wave slots per CU. Trajectories then agree with the fp64 build in DISTRIBUTION only.
#if defined(DTRL_REAL_F32)
typedef float real;
#else
typedef double real;
#endif

constexpr int kGroup = 64;       // lanes per env = one CDNA wavefront
constexpr int kMaxL = 24;        // links
constexpr int kMaxD = 24;        // generalised coordinates (planar root = 3, one per hinge)
constexpr int kMaxP = 40;        // controller params per action (dog 30, raptor 37)
constexpr int kMaxSets = 8;
constexpr int kMaxAct = 16;
constexpr int kMaxDepth = 12;    // longest root->link path (dog: 10)
constexpr int kMaxRows = 24;     // constraint rows per substep (joint limits + 2 per contact point)
constexpr int kPtsPerLink = 6;   // contact sample points per box link (4 corners + 2 long-edge midpoints)
constexpr int kMaxPtsPerLink = 4;  // constraint-carrying points of one link--ground pair (Bullet's persistent manifold holds 4): the deepest ones
constexpr int kMaxPtsPerPair = 2;  // constraint-carrying points of one link--link pair: in the plane two convex boxes touch along a segment at most, i.e. two points (the four points of Bullet's 3-D box--box manifold project onto two)
constexpr int kMaxPts = kMaxL * kPtsPerLink;
constexpr int kSegCap = 512;     // floats per heightfield segment slot (51 m at 0.1 m spacing)
constexpr int kNumGroundSamples = 200;
constexpr int kMaxFrags = 4;
constexpr int kMaxPairs = kMaxL * kMaxDepth;
constexpr int kMaxCP = 64;      // link--link collision pairs (dog 34, raptor 51); one lane per pair in the wave-wide proximity test

// Integrator v1 constants (DESIGN.md "Integrator v1"; mirrored by oracle/or_sim.h SimConst)
constexpr real kErp = 0.2;
constexpr real kSlop = 0.001;
constexpr real kMu = 0.9 * 0.9;
constexpr real kVDepenMax = 1.0;
constexpr real kLimitErp = 0.2;
constexpr real kLimitSlop = 0.005;
constexpr int kPgsIters = 10;
constexpr real kWarmFactor = 0.85;   // btContactSolverInfo::m_warmstartingFactor
// Bullet's `if (totalImpulse > 0)` in front of a friction row is a float compare. Redundant contact points (three collinear sample points of a foot flat on the ground) leave a
// normal row whose exact impulse is zero at +-1e-17 N s by rounding, and the rule would then decide between KEEPING a cached friction impulse and clamping it to mu x 1e-17 on the
// sign of that rounding error. An impulse below kHoldEps = 1e-9 N s (a resting dog foot carries 6e-2 per substep: this is below float resolution of that) counts as none
constexpr real kHoldEps = 1e-9;
constexpr real kGravityY = -9.8;
constexpr real kMaxTurnPerSubstep = 1.5707963267948966;   // Bullet clamps a body's angular velocity so that it turns at most MAX_ANGVEL = pi / 2 per internal step (btRigidBody::integrateVelocities); here: every hinge rate and the root's spin

enum Scenario { kScnSimChar = 0, kScnExp = 1, kScnPoliEval = 2 };

struct DevModel {
	int32_t L, D, P, n_opt;
	int32_t char_type, ctrl_type, scenario;
	int32_t num_update_steps, num_sim_substeps;
	int32_t n_sets, n_actions, default_action, enable_grav_comp, enable_vf;
	int32_t valid_init_pos_x, has_net;
	int32_t parent[kMaxL];
	int32_t depth[kMaxL];
	int8_t path[kMaxL][kMaxDepth];   // path[j][0..depth[j]] = root .. j
	uint32_t sub_mask[kMaxL];        // bit k set <=> link k is in the subtree rooted at j (incl. j)
	uint32_t anc_mask[kMaxL];        // bit a set <=> link a is an ancestor of j or j itself
	int32_t col[kMaxL];
	int32_t use_world[kMaxL];
	int32_t n_pairs;                 // (link, path position) pairs = nonzero hinge-hinge entries of the lower triangle of H
	int8_t pair_l[kMaxPairs], pair_k[kMaxPairs];
	int32_t act_idx0[kMaxAct], act_idx1[kMaxAct], act_cyclic[kMaxAct];
	int32_t opt_index[kMaxP];        // opt_index[k] = param index of the k-th optimisable param
	real attach[kMaxL][2];
	real lim_lo[kMaxL], lim_hi[kMaxL];
	real ref_theta[kMaxL];           // hinge reference angle: the controller reads theta through atan2(theta + ref_theta) - ref_theta (sim/World.cpp:543-553)
	real body_attach[kMaxL][2];
	real body_theta[kMaxL];
	real body_half[kMaxL][2];        // half extents
	real mass[kMaxL];
	real sub_mass[kMaxL];            // mass of the subtree rooted at j (summed in link order, as the kernel's subtree loop would)
	real inertia[kMaxL];             // Izz about the COM: m/12 (sx^2 + sy^2)
	real kp[kMaxL], kd[kMaxL], torque_lim[kMaxL], target_theta[kMaxL];
	real act_blend[kMaxAct];
	real ctrl_params[kMaxSets][kMaxP];
	real pose0[kMaxD], vel0[kMaxD];
	real pt_joint[kMaxL][kPtsPerLink][2];   // contact sample points in the JOINT frame: body_attach + R(body_theta) * corner (link--link tests: sharp boxes)
	real pt_ground[kMaxL][kPtsPerLink][2];  // the same points of the MARGIN-SHRUNK box (ground test: core point -> surface, minus contact_margin)
	real contact_margin;                    // Bullet's CONVEX_DISTANCE_MARGIN 0.04 in world-scaled units = 0.04 / world_scale metres (-collision_margin= overrides; 0 = sharp boxes)
	// the margin a link's box actually carries, metres: btBoxShape's constructor calls setSafeMargin(halfExtents) -> min(CONVEX_DISTANCE_MARGIN, 0.1 x the smallest
	// half extent) in world-scaled units (Bullet >= 2.80; the reference needs >= 2.82: sim/World.cpp:4-5 includes MLCPSolvers), e.g. 2.5 mm for a dog / goat toe and
	// 7.5 mm for the torso at ANY world scale. -safe_margin= 0 gives every link contact_margin (the round-3 model)
	real link_margin[kMaxL];
	// Bullet's contact persistence (round 5; sim/World.cpp:61-77 builds a default btSequentialImpulseConstraintSolver: SOLVER_USE_WARMSTARTING, factor 0.85 on the
	// persistent manifold points' normal and friction impulses). warm_start 1 (default): GROUND contact rows keep their identity across substeps and env-steps and start
	// the sweeps from kWarmFactor x the impulse they ended the last substep with, a sweep takes limits -> normals -> friction rows, and a friction row is resolved
	// only while its normal row carries an impulse (solveSingleIteration's `if (totalImpulse > 0)`); 0: every row from zero, one interleaved sweep (rounds 1-4).
	// Link--link contact rows start from zero: the comparator's ablations show its friction warm start acts through the ground contacts alone (ground-only
	// reproduces it, pair-only changes nothing: DESIGN 4), and a cached friction impulse between two links that no longer press on each other is 200 N of free force.
	// link_brk[j]: a ground sample point of link j carries rows while it is within contact_breaking (0.02, gContactBreakingThreshold, relative by the dispatcher's
	// default CD_USE_RELATIVE_CONTACT_BREAKING_THRESHOLD) x |half extents| above the surface; -contact_breaking= 0: only while it penetrates
	real link_brk[kMaxL];
	int32_t warm_start, pad_ws_;
	real eff_joint[kMaxL][2];               // body-local (0, -size_y/2) in the joint frame (end-effector contact position)
	real init_pos_x, target_vel_x, total_mass;
	real world_scale;
	real contact_tol;                       // cContactManager::Update: a link is in contact when a point is within 0.001 (world-scaled units) of the surface = 0.001 / world_scale
	// link--link collisions: links of one collision group that no hinge joins and whose boxes overlap in z collide with each other in the reference
	// (GetPartColGroup == GetPartColMask, sim/SimDog.cpp:73-81, sim/SimRaptor.cpp; only constraint-linked bodies are excluded: sim/World.cpp:626 with
	// sim/SimCharacter.cpp:864). cp_a < cp_b; cp_half = the box half extents the wave-wide overlap pre-test uses, grown by contact_tol. The pre-test
	// takes a box's orientation from its joint frame; the ROOT's box may be rotated against it (dog / goat: 0.61 rad) by cp_root_bt = (cos, sin); any
	// other link with a rotated box is replaced by the square around its bounding circle (conservative)
	int32_t link_contacts, n_cpairs;
	int8_t cp_a[kMaxCP], cp_b[kMaxCP];
	float cp_half[kMaxL][2];
	float cp_root_bt[2];
	real bt_cs[kMaxL], bt_sn[kMaxL];        // cos / sin of body_theta (box frame against the joint frame)
};

// exploration knobs + seeds: may change between launches (dtrl_set_explore)
struct RunParams {
	int32_t enable_exp;
	real exp_rate, exp_temp, exp_base_rate, exp_noise;
	uint64_t rng_seed;
	int64_t env_id_base;   // global env id of local env 0 (multi-GPU sharding keeps per-env streams shard-invariant)
};

struct EnvState {
	real q[kMaxD], qd[kMaxD];
	real tau[kMaxD];        // clamped joint torques held during the next world update
	real tau_ctrl[kMaxD];   // controller output before clamping (observability / parity tests)
	real pd_target[kMaxL];
	real params[kMaxP];     // current action parameters (mCurrAction.mParams)
	real phase, curr_cycle_time, prev_cycle_time, prev_stumble, curr_stumble;
	real prev_com[2], prev_dist[2];
	real fall_dist_counter, fall_contact_counter, sum_fall_contact, prev_check[2];
	real sample_origin[2];
	real time;
	real pos_start_x, avg_dist;
	uint64_t rng_ctr;
	int64_t num_cycles, num_resets, num_episodes;
	int32_t action_id, state, first_cycle, is_off_policy;
	int32_t exp_actor, exp_critic, cmd_action, fail_fall_dist;
	int32_t stance;          // raptor: 0 = right leg is the stance leg (gDefaultStance), 1 = left
	uint32_t pd_active_bits; // raptor: cPDController active flags per joint (dog: all active)
	uint32_t contact_bits;
	int32_t cycle_count, tuple_flags;
	int32_t need_reset;      // set by the kernel at frame end (fall); host regenerates terrain, then sets do_reset
	int32_t do_reset;        // consumed by the kernel at frame start
	int32_t do_init;         // first launch: full cScenario::Init ordering
	// external perturbation (sim/Perturb.cpp, sim/PerturbManager.cpp; one slot per env, a new one replaces the old): a world-frame force
	// on link pert_link at its COM plus the constant torque of its application offset, while pert_time < pert_dur
	int32_t pert_link;       // -1 = none
	int32_t pert_on;         // applied during the current env-step (set at the env-step's start)
	int32_t pad_;
	real pert_f[2], pert_lp[2], pert_torque, pert_time, pert_dur;   // pert_lp: application point relative to the COM, in the link's joint frame
	// Bullet's persistent contact points (DevModel::warm_start): the rows of the last solved substep by identity, with the impulses they ended with. Row ids
	// (16 bit, shared with oracle/or_sim.h): ground contact 2 x sample point + (0 normal, 1 tangent); link--link contact 512 + 2 x (pair x 12 + candidate) + (0, 1);
	// limit rows 0xffff (never matched). Carried across env-steps and frames; emptied by a reset
	real ws_lam[kMaxRows];
	uint16_t ws_id[kMaxRows];
	int32_t ws_R, pad_ws_;
};
static_assert(sizeof(EnvState) % 8 == 0, "EnvState is copied as 64-bit words");

struct GroundRec {
	// logical order: slot 0 = min segment, slot 1 = max segment (the host resolves the reference's mFlipSeg)
	double origin_x[2], scale_x[2], min_x[2], max_x[2];
	int32_t w[2];
	int32_t pad_[2];
	float data[2][kSegCap];
};

// ---- on-device terrain generation (-terrain_gen= device; dtrl_terrain_dev.h) ----
// generator state of one env: a counter-based random stream (key from terrain seed + GLOBAL env id, so windows do not depend on sharding or on
// scheduling) and bookkeeping
struct GroundGen {
	uint64_t key, ctr;
	int32_t builds;      // segments built so far
	int32_t overflow;    // a segment did not fit kSegCap vertices (the host path reports DTRL_ERR_CAPACITY; here the strip is cut and this is counted)
};
// what cGroundVar2D / cTerrainGen2D are configured with (one record per batch; the curriculum rewrites params)
struct TerrainCfg {
	int32_t type, pad_;
	double params[40];
	double world_scale, segment_width;
	double view_min, view_max;      // window the character must see: root_x + view_min .. root_x + view_max (scenarios/ScenarioSimChar.cpp:328-342)
	double spawn_min, spawn_max;    // bounds cScenarioSimChar::ResetGround / BuildGround build around (:344-369)
};
// poli_eval episode distances recorded by the device-side frame boundary (cScenarioPoliEval::RecordDistTraveled -> mDistLog)
struct DistRec { int32_t env, pad_; double dist; };

// per-env status the host reads back after each frame (24 B, coalesced)
struct EnvStatus {
	double root_x;
	int32_t need_reset;   // bit 0: the env fell, the host regenerates its terrain; bit 1: a poli_eval episode ended with a valid cycle, episode_dist holds its distance
	int32_t cost;   // work estimate for the env's next frame: sum over the last frame's substeps of (8 + constraint rows) + 400 if a policy forward is due
	double episode_dist;  // cScenarioPoliEval::RecordDistTraveled's dist (scenarios/ScenarioPoliEval.cpp:202-217) when need_reset & 2
};

// the policy forward keeps every activation in the env's LDS workspace (dtrl_kernel.h nn_eval): conv layers run on tiles of kConvTile
// output positions inside a kNNTileBuf-double buffer, the normalised input sits in a kNNSideBuf-double side buffer
constexpr int kConvTile = 16, kFcChunk = 32;
constexpr int kNNTileBuf = (kMaxRows + 1) * (kMaxD + 1);
constexpr int kNNSideBuf = kMaxL * (kMaxDepth + 2);
// MACE network family (data/policies/*/nets/*_mace3_deploy.prototxt)
struct NetDesc {
	int32_t n_terrain, n_char;
	int32_t conv_ch[3], conv_k[3];
	int32_t fc_terr, fc_trunk, fc_head, n_frags, frag_size;
	int32_t in_size, out_size;
	int64_t num_params;
};

}  // namespace dtrl
