/* dtrl.h -- C ABI of the MI355X batched rollout engine (libdtrl.so).
 *
 * The reference (xbpeng/DeepTerrainRL) has no FFI; its seam is the C++ virtual interface between the scenario
 * drivers (cScenarioTrain::ExpHelper, cOptScenarioPoliEval::EvalHelper) and ONE environment object. This header is
 * that seam for a BATCH of environments: every entry point names the reference interface it replaces
 * (file:line relative to the reference repo root). All buffers are caller-owned host memory; no torch types.
 * A batch handle is driven from one host thread (like one reference scenario object, scenarios/ScenarioTrain.cpp:467-475).
 *
 * Error behaviour mirrors the reference's bool-return + message convention (no exceptions): every call returns a
 * dtrl_status; dtrl_last_error() gives the message. There is NO CPU fallback: dtrl_create fails with
 * DTRL_ERR_NO_DEVICE when no HIP device is usable.
 */
#ifndef DTRL_H_
#define DTRL_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct dtrl_batch dtrl_batch;

typedef enum {
	DTRL_OK = 0,
	DTRL_ERR_ARG = 1,        /* bad argument / arg file */
	DTRL_ERR_IO = 2,         /* missing or malformed data file */
	DTRL_ERR_NO_DEVICE = 3,  /* no usable HIP device (product never falls back to CPU) */
	DTRL_ERR_DEVICE = 4,     /* HIP runtime error */
	DTRL_ERR_UNSUPPORTED = 5,
	DTRL_ERR_CAPACITY = 6
} dtrl_status;

/* flags returned by dtrl_get_flags */
#define DTRL_FLAG_FALLEN 1u
#define DTRL_FLAG_STUMBLED 2u
#define DTRL_FLAG_NEW_CYCLE 4u
#define DTRL_FLAG_STATE_SHIFT 8

/* tuple flag bits: learning/MACETrainer.h:11-17 (eFlagFail, eFlagExpCritic, eFlagExpActor) */
#define DTRL_TUPLE_FAIL 1u
#define DTRL_TUPLE_EXP_CRITIC 2u
#define DTRL_TUPLE_EXP_ACTOR 4u

/* Build a batch of num_envs environments from reference-format arguments.
 * Replaces: cArgParser(argv) + AppendArgs(-arg_file) (optimizer/Main.cpp:19-32, util/ArgParser.cpp:42-108) and, per env,
 * cScenarioExp/cScenarioPoliEval::ParseArgs + Init (scenarios/ScenarioSimChar.cpp:76-119, scenarios/ScenarioExp.cpp:31-61),
 * i.e. cScenarioTrain::BuildScenePool (scenarios/ScenarioTrain.cpp:197-222).
 * Relative paths inside the arg file resolve against the value of "-data_root=" if given, else the current directory
 * (the reference is run from its repo root). Extra keys understood: -data_root=, -terrain_seed= (env i uses seed+i),
 * -rand_seed= (exploration streams), -global_env_offset= (first global env id of this shard),
 * -physics_precision= f64 | f32: a CHECK, not a switch -- libdtrl.so computes in fp64 (default; the parity-tested product), libdtrl_f32.so is the same
 * source built with float arithmetic (opt-in; Bullet's own state is float, premake4.lua:115-124; distribution-level parity only) behind this same ABI
 * (doubles in, doubles out); each library fails dtrl_create with DTRL_ERR_ARG when asked for the other precision.
 * device_id < 0 selects the current HIP device. */
dtrl_status dtrl_create(const char* const* argv, int argc, int num_envs, int device_id, dtrl_batch** out);

/* Replaces: cScenario::Clear/Shutdown + destructor (scenarios/Scenario.h:15-23). */
dtrl_status dtrl_destroy(dtrl_batch* b);

/* Replaces: cScenarioExp::Reset / cScenarioPoliEval::Reset on the listed envs (scenarios/ScenarioSimChar.cpp:121-132,
 * scenarios/ScenarioExp.cpp:63-73). env_ids == NULL resets all. terrain_seeds != NULL re-seeds those envs' ground RNG
 * first (cScenarioPoliEval::SetRandSeed, scenarios/ScenarioPoliEval.cpp:153-160). */
dtrl_status dtrl_reset(dtrl_batch* b, const int32_t* env_ids, int n, const uint64_t* terrain_seeds);

/* Replaces: cScenarioExp::Update(dt) / cScenarioPoliEval::Update(dt) on every env
 * (scenarios/ScenarioExp.cpp:83-98, scenarios/ScenarioPoliEval.cpp:110-125): num_update_steps iterations of the loop at
 * scenarios/ScenarioSimChar.cpp:162-173, then fall handling (tuple + reset). dt is normally 1/30. */
dtrl_status dtrl_step(dtrl_batch* b, double dt);

/* dtrl_step split in two, so the caller can overlap its own GPU work (e.g. the trainer: the reference's env threads and trainer
 * run concurrently, scenarios/ScenarioTrain.cpp:100-115) with the frame kernel: dtrl_step_begin queues the frame launch on the
 * engine's stream and returns; dtrl_step_end waits for it and performs the frame-boundary host work (terrain windows, resets).
 * dtrl_step(dt) == dtrl_step_begin(dt); dtrl_step_end(). No other call on the batch is allowed between the two. */
dtrl_status dtrl_step_begin(dtrl_batch* b, double dt);
dtrl_status dtrl_step_end(dtrl_batch* b);

/* Finer grain: n iterations of the loop body only (no end-of-frame fall handling); used by parity tests and to count in
 * env-steps. The step length is (1/30)/num_update_steps. */
dtrl_status dtrl_step_updates(dtrl_batch* b, int n);

/* Asynchronous variant of dtrl_step for throughput runs: enqueue `frames` outer frames back-to-back on the batch's HIP
 * stream, doing the per-frame host work (terrain window slides, fall resets) between launches. Returns after the last
 * frame completed. */
dtrl_status dtrl_run_frames(dtrl_batch* b, int frames, double dt);

/* Replaces: cNNController::LoadNet + LoadModel + LoadScale (sim/NNController.cpp:49-91; learning/NeuralNet.cpp:81-215)
 * and cNeuralNet::CopyModel pushes from the trainer (learning/NeuralNetLearner.cpp:85-89). weights: flat float32 in Caffe
 * blob order of the deploy prototxt named by -policy_net= (W then b per layer). Offsets/scales follow
 * learning/NeuralNet.cpp:977-986,1027-1036. n must equal dtrl_policy_num_params().  * With -char_ctrl= dog_cacla the net is the CACLA ACTOR (cBaseControllerCacla::CopyActorNet, sim/BaseControllerCacla.cpp:67-75): weights in the actor
 * deploy net's blob order, output normalisers of its 29 outputs. */
dtrl_status dtrl_set_policy(dtrl_batch* b, const float* weights, size_t n, const double* in_off, const double* in_scale, const double* out_off, const double* out_scale);
dtrl_status dtrl_policy_num_params(const dtrl_batch* b, size_t* n);

/* Replaces: cNNController::BuildNNOutputOffsetScale (sim/BaseControllerMACE.cpp:75-113, sim/DogControllerMACE.cpp:93-99);
 * used by cScenarioTrain::SetupTrainerOutputOffsetScale (scenarios/ScenarioTrain.cpp:322-338). */
dtrl_status dtrl_build_output_offset_scale(const dtrl_batch* b, double* out_off, double* out_scale);

/* Replaces: cNeuralNet::LoadScale (learning/NeuralNet.cpp:137-215) and cNeuralNet::WriteOffsetScale (:1182-1205): the
 * "<model>_scale.txt" normaliser files ({"InputOffset": [...], "InputScale": [...], "OutputOffset": [...], "OutputScale": [...]},
 * values printed with std::to_string). Loading keeps the weights, replaces the vectors present in the file (absent keys keep their
 * current value, a wrong length is an error, as in the reference); the batch must have been built with -policy_net=. */
dtrl_status dtrl_load_scale_file(dtrl_batch* b, const char* path);
dtrl_status dtrl_write_scale_file(dtrl_batch* b, const char* path);

/* Replaces: cScenarioExp::EnableExplore / SetExpRate / SetExpTemp / SetExpBaseActionRate (scenarios/ScenarioExp.cpp:161-206). */
dtrl_status dtrl_set_explore(dtrl_batch* b, int enable, double rate, double temp, double base_rate);

/* Replaces: cScenarioSimChar::SetTerrainParamsLerp (scenarios/ScenarioSimChar.cpp:255-272). */
dtrl_status dtrl_set_terrain_lerp(dtrl_batch* b, double lerp);

/* Replaces: cScenarioExp::IsTupleBufferFull/GetTuples/ResetTupleBuffer (scenarios/ScenarioExp.h:16-38) for the whole batch.
 * rows: [cap][1 + 2S + A] float32 in the MACE replay row layout [r | s | a | s'] (learning/MACETrainer.cpp:373-401).
 * Where the rings live is a creation argument: `-tuple_ring= device` (default; the drain is a count read-back plus three copies through a page-locked staging
 * area, two synchronisations) or `-tuple_ring= host` (page-locked host memory the kernels write directly: the drain queues NOTHING on the GPU -- with frames
 * in flight every queued copy can wait milliseconds for a wavefront slot, measured 1.2-2.3 ms for the 4-byte count alone). Same rows either way. */
dtrl_status dtrl_drain_tuples(dtrl_batch* b, float* rows, uint32_t* flags, int32_t* env_ids, int cap, int* out_n);

/* dtrl_drain_tuples with DEVICE destination buffers (e.g. the trainer's replay tensors on the same GPU, or the send buffer of the RCCL tuple
 * gather): device-to-device copies on the batch's stream, completed on return; no host staging. The reference hands tuples to the trainer by
 * reference under its lock (learning/NeuralNetLearner.cpp:33-46). flags_dev / env_ids_dev may be NULL. */
dtrl_status dtrl_drain_tuples_device(dtrl_batch* b, float* rows_dev, uint32_t* flags_dev, int32_t* env_ids_dev, int cap, int* out_n);
/* Replaces: the same three calls as dtrl_drain_tuples_device, plus the per-rank packing the reference's learner thread does before it hands tuples to
 * the trainer (learning/NeuralNetLearner.cpp:33-46) -- for a consumer that wants ONE device block it can put on the wire as it is (one RCCL
 * gather per frame). block_dev: [block_rows + 1][W + 2] float32 in DEVICE memory. Row 0 is a header (int32 bit patterns: [0] = number of rows that
 * follow, [1] = rows lost because the RING was full since the last drain -- counted in dtrl_tuple_stats --, [2] = rows carried); rows 1..n are the
 * pending tuples sorted by env id (stable: an env's tuples stay in time order), each [r | s | a | s' | flag word | GLOBAL env id], the last two as int32
 * bit patterns. Rows that do not fit block_rows are CARRIED, not dropped: they move to the front of the ring, in order, and the next drain of that ring
 * hands them out in front of the newer rows -- so a consumer can size its block for the steady state (~0.08 rows per env and frame) instead of for the
 * worst case. Everything runs on the device (segmented counting sort by env id, copy and header kernels); out_n, when not NULL, costs one 4-byte read-back. */
dtrl_status dtrl_drain_tuples_packed(dtrl_batch* b, float* block_dev, int block_rows, int* out_n);
/* Replaces: the concurrency of the reference's learner threads -- an env thread hands its tuples to the trainer under the trainer's lock while the other
 * env threads keep stepping (scenarios/ScenarioTrain.cpp:322-338, 376-410; learning/NeuralNetLearner.cpp:33-46). Batched equivalent: with pipelining on,
 * every dtrl_step_begin switches between two tuple rings, and a drain issued between dtrl_step_begin(f + 1) and dtrl_step_end(f + 1) returns the tuples of
 * frame f from the ring that frame wrote, on a stream of its own, without waiting for frame f + 1 (dtrl_step_end(f); dtrl_step_begin(f + 1);
 * dtrl_drain_tuples_packed(...) -> frame f's tuples while f + 1 runs). Outside a pending step a drain returns the ring of the last frame, as without
 * pipelining. Off by default; the caller drains every frame while it is on (an undrained ring is not lost: it is appended to two frames later). Switching it off
 * requires the idle ring to be empty. */
dtrl_status dtrl_set_tuple_pipelining(dtrl_batch* b, int on);
/* dtrl_step_end followed by dtrl_step_begin(dt) without the barrier between them (same results): each env group receives its frame-boundary host
 * work and its next launch as soon as its own frame is done, so the slowest wavefronts of one group are covered by the other groups' next launches --
 * what dtrl_run_frames does inside, one frame at a time, for callers that act between frames (drain tuples, scenarios/ScenarioTrain.cpp:376-410).
 * Without a pending step it is dtrl_step_begin. */
dtrl_status dtrl_step_end_begin(dtrl_batch* b, double dt);
/* For a caller that works for milliseconds between two dtrl_step_end_begin calls (a trainer going through the drained tuples, one Train() per 32 of them --
 * the reference's env threads keep stepping meanwhile, scenarios/ScenarioTrain.cpp:376-410). Never blocks: every env group whose frame has ALREADY ended gets its
 * frame-boundary work and its next launch now (writing the ring the caller has just drained) instead of waiting for the caller to come back; *relaunched (may be
 * NULL) = how many groups that was. The next dtrl_step_end_begin then handles the other groups only. Needs tuple pipelining, `-tuple_ring= host` (the idle ring's
 * cursor is checked without queueing a copy) and host terrain mode; otherwise, or when the idle ring still holds rows, it does nothing. After a relaunch both rings
 * are being written: tuple drains and dtrl_step_end are refused until the next dtrl_step_end_begin. A group relaunched here runs its frame with the policy of the
 * last hand-over that had taken effect (one frame staler than the groups relaunched later). */
dtrl_status dtrl_step_poll(dtrl_batch* b, double dt, int* relaunched);
/* The reference never drops a tuple (scenarios/ScenarioTrain.cpp:376-410 trains whenever a scene's buffer is full). Here the ring holds
 * max(2 num_envs, -tuple_buffer_size=) rows (-tuple_ring_capacity= overrides); rows completed while it is full are COUNTED, not stored:
 * pending = rows waiting in the ring, drained = rows handed out so far, dropped = rows lost to a full ring since creation (stays 0 when
 * the caller drains at least every ~2 gait cycles), capacity = ring size. Any output may be NULL. */
dtrl_status dtrl_tuple_stats(dtrl_batch* b, int64_t* pending, int64_t* drained, int64_t* dropped, int32_t* capacity);
/* dtrl_set_policy with every pointer in DEVICE memory (cNeuralNet::CopyModel is a memcpy per blob between two nets of one process,
 * learning/NeuralNet.cpp:636-658): the trainer's weight blob -- same Caffe blob order -- is re-laid into the kernel's layout by a gather kernel;
 * NULL normaliser pointers keep the current vectors. Between dtrl_step_begin and dtrl_step_end a weights-only call does not wait for the frame in flight: the
 * weights are gathered into a second buffer (weights_dev may be changed when the call returns) and every env's NEXT frame launch runs with them -- the moment the
 * waiting form takes effect, too. */
dtrl_status dtrl_set_policy_device(dtrl_batch* b, const float* weights_dev, size_t n, const double* in_off_dev, const double* in_scale_dev, const double* out_off_dev, const double* out_scale_dev);
/* The weights-only form with the re-layout kernel queued on a stream of the CALLER's (a hipStream_t; the trainer's): it follows whatever the caller has queued
 * there -- the trainer's last step -- and the call returns when it has run, so ONE host wait covers the trainer's pending work and the hand-over (during a frame the
 * engine's own stream would have to find a wavefront slot of its own). Same semantics as dtrl_set_policy_device otherwise. */
dtrl_status dtrl_set_policy_device_on(dtrl_batch* b, const float* weights_dev, size_t n, void* stream);
/* The weights-only form with NO host wait: the re-layout kernel is queued on `stream` (a hipStream_t of the caller's, required) behind whatever produced
 * weights_dev there -- an RCCL broadcast of cNeuralNetLearner::SyncNet's payload (learning/NeuralNetLearner.cpp:85-89), the trainer's last step -- and every env's
 * NEXT frame launch waits for it ON THE DEVICE and runs with the new weights. Valid with or without a frame in flight. weights_dev must not change until the
 * work queued on `stream` has passed this point (the caller's next write to it on the same stream is ordered by the stream). Needs normalisers installed by an
 * earlier dtrl_set_policy / dtrl_set_policy_device. */
dtrl_status dtrl_set_policy_device_async(dtrl_batch* b, const float* weights_dev, size_t n, void* stream);

/* No counterpart in the reference (its trainer and its env threads share CPU cores under the OS scheduler; scenarios/ScenarioTrain.cpp runs them as threads
 * of one process). On the GPU a frame launch fills every wavefront slot of the compute units it may use for milliseconds, so work that should run BESIDE the
 * rollout -- the trainer's kernels, the exchange's collective -- needs (1) compute units the frame launches leave alone: `-reserve_cus= k` at creation (or
 * DTRL_RESERVE_CUS=k) keeps k units per XCD out of them (k / 32 of the rollout rate), and (2) a hardware queue that is not held up by a frame launch waiting
 * for slots: the engine measures its candidate streams at creation and hands out the two quickest. Returns a hipStream_t (k = 0, 1) owned by the batch, or
 * NULL without a reservation; *start_delay_us (may be NULL) = how long a burst of 10 small kernels on that stream took beside two frame-sized occupant
 * launches during the calibration (microseconds: about 70 on a free queue, a thousand or more on a held-up one). */
void* dtrl_side_stream(dtrl_batch* b, int k, double* start_delay_us);

/* Replaces: cSimCharacter::BuildPose / BuildVel (sim/SimCharacter.cpp:166-225). env_ids == NULL -> envs 0..n-1. */
dtrl_status dtrl_get_pose_vel(dtrl_batch* b, const int32_t* env_ids, int n, double* q, double* qd);
/* Replaces: cCharController::CommandAction (sim/CharController.h:23, sim/DogController.cpp:309-320): action_ids[i] (an index into the controller's
 * action table, dtrl_get_action_table) is the base action env i takes at its next cycle instead of asking the policy. env_ids == NULL -> envs 0..n-1 =
 * all envs, action_ids then holds num_envs entries. The reference keeps a stack of commands; the engine keeps its top only (one pending command per env,
 * a new one replaces it -- also the random first action cScenarioExp::Reset queues, scenarios/ScenarioExp.cpp:63-73). */
dtrl_status dtrl_command_action(dtrl_batch* b, const int32_t* env_ids, int n, const int32_t* action_ids);
/* Replaces: cSimCharacter::SetPose / SetVel (sim/SimCharacter.cpp:665-683, 227-315). A teleported character drops its persistent contact rows (what Bullet's
 * refreshContactPoints does to manifold points that moved out of the breaking threshold): restoring a saved state = dtrl_set_pose_vel, THEN dtrl_set_contact_cache. */
dtrl_status dtrl_set_pose_vel(dtrl_batch* b, const int32_t* env_ids, int n, const double* q, const double* qd);
/* Replaces: the state Bullet's collision world keeps BETWEEN stepSimulation calls besides the bodies' poses and velocities: the persistent contact points of
 * the dispatcher's manifolds with their applied normal / friction impulses (btManifoldPoint::m_appliedImpulse, m_appliedImpulseLateral1), which the default
 * btSequentialImpulseConstraintSolver the reference builds (sim/World.cpp:61-77) warm-starts from with factor 0.85. A character's dynamic state is (q, qd) PLUS this
 * cache: whoever saves, restores or transplants a character mid-run (checkpointing, the side-by-side parity tests) moves both. Per env i: count[i] rows
 * (<= DTRL_MAX_CONTACT_ROWS), ids[i][k] = identity of row k (ground contact 2 x sample point + (0 normal, 1 tangent); link--link contact 512 + 2 x (pair x 12 +
 * candidate) + (0, 1); 65535 = a joint-limit row, never matched) and lambda[i][k] = the impulse it ended the last substep with. cWorld::Reset empties it. */
#define DTRL_MAX_CONTACT_ROWS 24
dtrl_status dtrl_get_contact_cache(dtrl_batch* b, const int32_t* env_ids, int n, int32_t* count, int32_t* ids, double* lambda);
dtrl_status dtrl_set_contact_cache(dtrl_batch* b, const int32_t* env_ids, int n, const int32_t* count, const int32_t* ids, const double* lambda);

/* Replaces: cScenarioSimChar::AddPerturb -> cWorld::AddPerturb (scenarios/ScenarioSimChar.cpp:204-207, sim/World.cpp:256-259) with a
 * tPerturb of type ePerturbForce (sim/Perturb.cpp:52-79, sim/World.cpp:445-470): a world-frame force[n][2] on body part link[n] at the
 * body-local offset local_pos[n][2] (NULL = the COM) for duration[n] seconds of simulated time, advanced and applied at the start of
 * every env-step like cPerturbManager::UpdatePerturbs. One slot per env (a new perturbation replaces the old one); reset clears it. */
dtrl_status dtrl_add_perturb(dtrl_batch* b, const int32_t* env_ids, int n, const int32_t* link, const double* local_pos, const double* force, const double* duration);
/* Replaces: cScenarioSimChar::ApplyRandForce() (scenarios/ScenarioSimChar.cpp:209-235; ranges -min_perturb= -max_perturb=
 * -min_pertrub_duration= -max_perturb_duration= as the reference spells them): a random body part, direction, magnitude and duration per
 * env. The reference draws from its time-seeded global RNG; here the draw is a function of (seed, global env id). */
dtrl_status dtrl_apply_rand_force(dtrl_batch* b, const int32_t* env_ids, int n, uint64_t seed);
/* Replaces: cNNController::RecordPoliState (sim/TerrainRLCharController.cpp:120-123). */
dtrl_status dtrl_get_poli_state(dtrl_batch* b, const int32_t* env_ids, int n, double* s);
/* Replaces: cNeuralNet::GetLayerState("output", y) (learning/NeuralNet.cpp:814-834) after the controller's last cNeuralNet::Eval
 * (learning/NeuralNet.cpp:352-375; what -record_nn_activation= true -nn_activation_layer= output writes, scenarios/ScenarioPoliEval.cpp:271-286)
 * with the output un-normalisation of Eval applied: y[n][nn_out of dtrl_dims] = the net's outputs of each env's most recent action decision
 * that evaluated the net (zeros before the first one). */
dtrl_status dtrl_get_policy_output(dtrl_batch* b, const int32_t* env_ids, int n, double* y);
/* fallen | stumbled<<1 | new_cycle<<2 | fsm_state<<8: cSimCharacter::HasFallen/HasStumbled, cCharController::IsNewCycle/GetState. */
dtrl_status dtrl_get_flags(dtrl_batch* b, const int32_t* env_ids, int n, uint32_t* bits);
/* Replaces: cSimCharacter::GetBodyPart(i)->GetPos() / GetLinearVelocity() / GetRotation() (sim/SimCharacter.cpp:317-352, sim/SimObj.cpp:
 * 60-130), what the reference's features, recorders and draw code read per link: world position and velocity of every link's body
 * COM ([n][L][2] each) and the body's world angle ([n][L]); derived from (q, qd) with the same planar kinematics the kernel uses.
 * Any output may be NULL. */
dtrl_status dtrl_get_link_states(dtrl_batch* b, const int32_t* env_ids, int n, double* com_xy, double* com_vel_xy, double* angle);

/* observability for parity tests: controller torque before / after the cJoint clamp (sim/Joint.cpp:171-201), per-link contact flags */
dtrl_status dtrl_get_torques(dtrl_batch* b, const int32_t* env_ids, int n, double* tau_ctrl, double* tau_applied);
dtrl_status dtrl_get_contacts(dtrl_batch* b, const int32_t* env_ids, int n, int32_t* flags);
/* current action: cTerrainRLCharController::GetCurrActionID + mCurrAction.mParams, PD targets */
dtrl_status dtrl_get_ctrl(dtrl_batch* b, const int32_t* env_ids, int n, int32_t* state, double* phase, int32_t* action_id, double* params, double* pd_targets);
/* What cScenarioPoliEval's per-cycle recorders read (scenarios/ScenarioPoliEval.cpp:234-404: RecordAction, RecordVel, RecordActionIDState): the
 * env's cycle counter (mCycleCount; like the reference's it survives resets) and reset counter, the COM and simulated time at the start of the current cycle (what mChar->CalcCOM() / mTime returned when
 * the cycle began), and the optimisable parameters of the current action (cTerrainRLCharController::BuildOptParams, [n][frag_size] as dtrl_dims reports it). The action id is in
 * dtrl_get_ctrl, the policy state of the current action in dtrl_get_poli_state. deepterrainrl_amd.recorders.PoliEvalRecorder writes the
 * reference's files from these at frame boundaries (all of them are constant over a cycle, so nothing is lost). Any output may be NULL. */
dtrl_status dtrl_get_cycle_info(dtrl_batch* b, const int32_t* env_ids, int n, int64_t* num_cycles, int64_t* num_resets, double* cycle_start_com, double* cycle_start_time, double* opt_params);
/* Replaces: cTerrainRLCharController::BuildActionOptParams(a) for every action a (what cScenarioPoliEval::InitActionRecord writes): table[n_actions][frag_size];
 * returns the number of actions through n_actions (table may be NULL to query it). */
dtrl_status dtrl_get_action_table(dtrl_batch* b, int* n_actions, double* table);
/* ground observability: cGround::SampleHeight (sim/GroundVar2D.cpp:98-114) with the grid cell it used (terrain-index parity) */
dtrl_status dtrl_sample_ground(dtrl_batch* b, int env, int n, const double* x, double* h, int32_t* seg, int32_t* i, int32_t* j);
/* Replaces: cGroundVar2D::GetSegment(s) / tSegment::mData, GetMinX / GetMaxX (sim/GroundVar2D.cpp:279-290, 392-455, 504-520) of one env's two-segment
 * window in logical order (slot 0 = min segment): vertex counts, x ranges, and the heights (metres, float) of each slot, heights0 / heights1 holding
 * up to cap values each (may be NULL). num_builds = segments built for this env so far (-terrain_gen= device; -1 in host mode) */
dtrl_status dtrl_get_ground_window(dtrl_batch* b, int env, int32_t* w2, double* min_x2, double* max_x2, float* heights0, float* heights1, int cap, int64_t* num_builds);

/* Replaces: cScenarioPoliEval::GetAvgDist / GetNumEpisodes / GetNumCycles (scenarios/ScenarioPoliEval.h:20-26), batch aggregate. */
dtrl_status dtrl_eval_stats(dtrl_batch* b, double* avg_dist, int64_t* episodes, int64_t* cycles, int64_t* resets);

/* Replaces: cScenarioPoliEval::GetDistLog (scenarios/ScenarioPoliEval.cpp:147-150, filled by RecordDistTraveled :202-217) for the batch: the distance of
 * every recorded episode since creation, grouped by env id (= pool member, the order cOptScenarioPoliEval::OutputResults walks,
 * optimizer/scenarios/OptScenarioPoliEval.cpp:213-239), each env's episodes in time order. *out_n = number of entries (call with cap 0 and
 * NULL buffers to size them). */
dtrl_status dtrl_get_dist_log(dtrl_batch* b, double* dist, int32_t* env_ids, int cap, int* out_n);
/* Replaces: cScenarioPoliEval::ResetAvgDist (scenarios/ScenarioPoliEval.cpp:132-136) on every env: average distance and episode count restart,
 * cycle counters and the dist log stay (cOptScenarioPoliEval::EvalHelper calls it after folding a batch of episodes into its record, :184-196). */
dtrl_status dtrl_reset_avg_dist(dtrl_batch* b);
/* Replaces: cOptScenarioPoliEval::OutputResults (optimizer/scenarios/OptScenarioPoliEval.cpp:213-239): appends ONE line to `path`, every logged
 * distance in dtrl_get_dist_log order, printed with std::to_string and separated by ", ". */
dtrl_status dtrl_write_dist_log(dtrl_batch* b, const char* path);

/* sizes: L links, D dofs, S policy-state, A policy-action (1 + frag), P controller params, nn_out, num_frags, frag_size */
dtrl_status dtrl_dims(const dtrl_batch* b, int* L, int* D, int* S, int* A, int* P, int* nn_out, int* num_frags, int* frag_size);

/* HIP stream the batch launches on (so callers can bracket it with their own events), and the last kernel timing:
 * average duration in ms of the frame kernel over the launches since the previous call, measured with hipEvents on that stream. */
dtrl_status dtrl_kernel_time_ms(dtrl_batch* b, double* avg_ms, int64_t* launches);

/* ---- host-side pieces of the path that need no batch and no device (the reference's static / utility entry points) ---- */

/* Replaces: cTerrainGen2D::ParseType + GetTerrainFunc(type) (sim/TerrainGen2D.cpp:90-181) called as func(width, params, rand, data) on a
 * cRand seeded with `seed` (util/Rand.cpp:89-92), i.e. what cGroundVar2D::BuildSegment runs per segment (sim/GroundVar2D.cpp:312-342).
 * params40 in cTerrainGen2D::eParams order (sim/TerrainGen2D.h:31-81). The strip (float heights, 0.1 m vertex spacing) is written to
 * out[0..min(n, cap)); *out_n = vertex count, *out_width = the function's return value (metres added). Bit-identical to the reference
 * on the same libstdc++. */
dtrl_status dtrl_terrain_build(const char* type_name, const double* params40, uint64_t seed, double width, float* out, int cap, int* out_n, double* out_width);
/* Replaces: the terrain-file reader of cScenarioSimChar::ParseTerrainParams (scenarios/ScenarioSimChar.cpp:670-706) + cTerrainGen2D::LoadParams
 * (sim/TerrainGen2D.cpp:69-81): "Type" string and every 40-vector of the "Params" array (defaults sim/TerrainGen2D.cpp:8-56). */
dtrl_status dtrl_terrain_load_file(const char* path, char* type_out, int type_cap, double* params_out, int max_sets, int* out_sets);
/* Replaces: cArgParser(argv, argc) + AppendArgs(-arg_file) + ParseString(key) (optimizer/Main.cpp:19-32, util/ArgParser.cpp:42-108, 131-150) exactly as
 * dtrl_create resolves its arguments (-data_root= prefixes a relative -arg_file=). *found = 0 when the key is absent or followed by another key.
 * Returns the number of tokens through n_tokens (may be NULL). */
dtrl_status dtrl_args_parse_string(const char* const* argv, int argc, const char* key, char* out, int cap, int* found, int* n_tokens);

const char* dtrl_last_error(const dtrl_batch* b);
const char* dtrl_version(void);

#ifdef __cplusplus
}
#endif
#endif /* DTRL_H_ */
