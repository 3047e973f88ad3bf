// ORACLE (test infrastructure, NOT product code).
// Flat character / scenario description handed to the oracle by oracle/model.py (which reads the reference's
// JSON + arg files with Python's json module, independently of the product's C++ loader).
// Field meanings follow anim/KinTree.h:24-64 (joint + body tables), sim/PDController.h (eParam*),
// sim/DogController.h:12-44 (param layout), scenarios/ScenarioSimChar.cpp:76-108 (scenario args).
#pragma once
#include <cstdint>

#define ORC_MAXL 24
#define ORC_MAXD 26
#define ORC_MAXP 40
#define ORC_MAXSETS 8
#define ORC_MAXACT 16
#define ORC_MAXTP 4
#define ORC_MAXCP 64

extern "C" {

struct OrcModel {
	int32_t char_type;   // 0 = dog (also goat), 1 = raptor
	int32_t ctrl_type;   // 0 = FSM without net ("dog"/"raptor": cDogControllerQ w/o net), 1 = MACE, 2 = CACLA (actor as a one-fragment MACE-family net with a zero critic head)
	int32_t L, D;
	int32_t parent[ORC_MAXL];
	int32_t joint_type[ORC_MAXL];          // cKinTree::eJointType (0 revolute, 1 planar)
	double attach[ORC_MAXL][3];            // joint attach point in parent joint frame
	double lim_lo[ORC_MAXL], lim_hi[ORC_MAXL];
	double ref_theta[ORC_MAXL];            // cWorld::tJointParams::mRefTheta as cSimCharacter::BuildConstraints computes it (sim/SimCharacter.cpp:846-865)
	double body_attach[ORC_MAXL][3];       // body COM in joint frame
	double body_theta[ORC_MAXL];
	double body_size[ORC_MAXL][3];
	double body_mass[ORC_MAXL];
	int32_t col_group[ORC_MAXL];           // 0 = collides with nothing (sim/SimDog.cpp:5-33)
	double kp[ORC_MAXL], kd[ORC_MAXL], torque_lim[ORC_MAXL], target_theta[ORC_MAXL];
	int32_t use_world[ORC_MAXL];
	int32_t n_sets, n_params;
	double ctrl_params[ORC_MAXSETS][ORC_MAXP];
	int32_t opt_mask[ORC_MAXP];            // gParamInfo / gOptParamsMasks: 1 = optimisable (part of the action fragment)
	double exp_noise;                      // mExpNoise: dog 0.2 (sim/DogControllerMACE.cpp:7), raptor 0.15 (sim/RaptorControllerMACE.cpp:7)
	int32_t n_actions;
	int32_t act_idx0[ORC_MAXACT], act_idx1[ORC_MAXACT];
	double act_blend[ORC_MAXACT];
	int32_t act_cyclic[ORC_MAXACT];
	int32_t default_action;
	int32_t enable_grav_comp;
	int32_t enable_virtual_forces;
	double target_vel_x;
	double pose0[ORC_MAXD], vel0[ORC_MAXD];
	int32_t valid_init_pos_x;
	double init_pos_x;
	int32_t num_update_steps, num_sim_substeps;
	double world_scale;
	int32_t terrain_type;
	int32_t n_terrain_sets;
	double terrain_params[ORC_MAXTP][40];
	double terrain_blend;
	int32_t scenario;     // 0 = sim_char (no auto reset), 1 = exp (tuples, reset on fall), 2 = poli_eval (reset on fall, dist log)
	int32_t tuple_buffer_size;
	int32_t enable_explore;
	double exp_rate, exp_temp, exp_base_rate;
	// link--link collision pairs: links of one collision group that no hinge joins and whose boxes overlap in z collide with each other in the
	// reference (cSimDog::GetPartColGroup == GetPartColMask, sim/SimDog.cpp:73-81; only constraint-linked bodies are excluded, sim/World.cpp:626,
	// sim/SimCharacter.cpp:864). a < b. link_contacts = 0 switches the pair contacts off (the round-1 model)
	int32_t link_contacts, n_cpairs;
	int32_t cpair_a[ORC_MAXCP], cpair_b[ORC_MAXCP];
	// collision margin of the box links against the GROUND, metres: Bullet's CONVEX_DISTANCE_MARGIN 0.04 in world-scaled units = 0.04 / world_scale (1 cm for the
	// dog at world scale 4, 4 cm for the goat scene whose arg file leaves the scale at 1). 0 = sharp boxes (the round-2 model, -collision_margin= 0)
	double contact_margin;
	// per-link margin actually used, metres. Bullet's btBoxShape constructor calls setSafeMargin(halfExtents): margin = min(CONVEX_DISTANCE_MARGIN, 0.1 x the
	// smallest half extent) in world-scaled units (btConvexInternalShape::setSafeMargin, Bullet >= 2.80; the reference needs >= 2.82, sim/World.cpp:4-5), so a
	// thin link carries a thin margin and the box core never inverts: dog / goat toe 2.5 mm, torso 7.5 mm at ANY world scale. -safe_margin= 0 restores the
	// uniform round-3 value contact_margin for every link
	double link_margin[ORC_MAXL];
	// Bullet's contact persistence (round 5; sim/World.cpp:61-77 builds a default btSequentialImpulseConstraintSolver, whose btContactSolverInfo has
	// SOLVER_USE_WARMSTARTING with factor 0.85 on the persistent manifold points' applied NORMAL and FRICTION impulses):
	//   warm_start 1 (default, what the product kernels run): a GROUND contact row (sample point) keeps its identity across substeps
	//     and env-steps and starts the sweeps from 0.85 x the impulse it ended the previous substep with (limit rows start from zero: the solver zeroes the rows of
	//     typed constraints; link--link contact rows too: the comparator's friction warm start acts through the ground contacts alone -- ground-only reproduces it,
	//     pair-only changes nothing, DESIGN 4 -- and a cached friction impulse between two links that no longer press on each other is free force); a sweep resolves the limit rows, then every normal row, then every friction row (solveSingleIteration's order), and a friction row
	//     only while its normal row carries an impulse (`if (totalImpulse > 0)`), so that the cached friction impulse of a contact without normal force stays applied;
	//   warm_start 0: every row from zero, one interleaved sweep (rounds 1-4).
	//   Oracle-only ablations (tools/a2_deviation.py): 2 = as 1 with the interleaved sweep; 3 = as 1 with Bullet's friction direction (along the pre-solve
	//     tangential velocity, else plane space); 4 = plain warm start of every row without the friction rule (round 4's -warm_start= 1).
	int32_t warm_start;
	// contact_breaking (default 0.02 = gContactBreakingThreshold, relative: btCollisionDispatcher's default CD_USE_RELATIVE_CONTACT_BREAKING_THRESHOLD): a ground
	// sample point carries rows while it is within link_brk[j] = contact_breaking x |half extents| ABOVE the surface (the persistent manifold keeps such a point;
	// its normal row lets it approach by its distance per substep: Bullet's `velocityError -= penetration / dt` for positive distance). 0 = rows only while penetrating
	double link_brk[ORC_MAXL];
	// -mass_matrix_every= N: the joint-space inertia H(q) is rebuilt (and factorised) at every N-th substep of an env-step and held in between; bias forces, contact
	// geometry and constraint Jacobians are evaluated at the current configuration in every substep. 1 = every substep (rounds 1-3)
	int32_t mass_matrix_every;
};

// MACE network family of data/policies/*/nets/*_mace3_deploy.prototxt
struct OrcNetDesc {
	int32_t n_terrain;      // 200 (slice_point)
	int32_t n_char;         // input_dim - 200
	int32_t conv_ch[3];     // 16, 32, 32
	int32_t conv_k[3];      // 8, 4, 4
	int32_t fc_terr;        // 64
	int32_t fc_trunk;       // 256
	int32_t fc_head;        // 128
	int32_t n_frags;        // 3
	int32_t frag_size;      // 29 / 28
};

}  // extern "C"
