// dtrl_c_api.cpp -- extern "C" surface declared in include/dtrl.h; thin wrappers over dtrl::Engine.
#include "../../include/dtrl.h"
#include "dtrl_engine.h"
#include <cmath>
#include <cstdio>
#include <new>
#include <stdexcept>
#include <string>
#include <vector>

using dtrl::Engine;
using dtrl::EnvState;

struct dtrl_batch { Engine eng; };

static thread_local std::string g_create_error;

// nothing may propagate across the C boundary (the reference's convention is bool + message, no exceptions): allocation failures and
// anything else thrown below the ABI become a status code and a message
static int dtrl_on_exception(const dtrl_batch* b)
{
	std::string msg = "internal error";
	int code = DTRL_ERR_DEVICE;
	try { throw; }
	catch (const std::bad_alloc&) { msg = "out of host memory"; code = DTRL_ERR_CAPACITY; }
	catch (const std::length_error& e) { msg = std::string("invalid size: ") + e.what(); code = DTRL_ERR_ARG; }
	catch (const std::exception& e) { msg = std::string("internal error: ") + e.what(); }
	catch (...) {}
	if (b) const_cast<dtrl_batch*>(b)->eng.set_error(msg); else g_create_error = msg;
	return code;
}

extern "C" {

const char* dtrl_version(void) { return sizeof(dtrl::real) == 4 ? "dtrl-mi355x 0.6 (round 6), fp32 build (opt-in: -physics_precision= f32)" : "dtrl-mi355x 0.6 (round 6), fp64"; }

const char* dtrl_last_error(const dtrl_batch* b) { return b ? b->eng.error().c_str() : g_create_error.c_str(); }

dtrl_status dtrl_create(const char* const* argv, int argc, int num_envs, int device_id, dtrl_batch** out)
try {
	if (!out) return DTRL_ERR_ARG;
	*out = nullptr;
	dtrl_batch* b = new (std::nothrow) dtrl_batch();
	if (!b) return DTRL_ERR_DEVICE;
	int rc = b->eng.Create(argv, argc, num_envs, device_id);
	if (rc != DTRL_OK) { g_create_error = b->eng.error(); delete b; return static_cast<dtrl_status>(rc); }
	*out = b;
	return DTRL_OK;
} catch (...) { return static_cast<dtrl_status>(dtrl_on_exception(nullptr)); }
dtrl_status dtrl_destroy(dtrl_batch* b) { delete b; return DTRL_OK; }

#define CHECK_B() do { if (!b) return DTRL_ERR_ARG; } while (0)

dtrl_status dtrl_reset(dtrl_batch* b, const int32_t* env_ids, int n, const uint64_t* seeds) try { CHECK_B(); return static_cast<dtrl_status>(b->eng.Reset(env_ids, n, seeds)); } catch (...) { return static_cast<dtrl_status>(dtrl_on_exception(b)); }
dtrl_status dtrl_step(dtrl_batch* b, double dt) try { CHECK_B(); return static_cast<dtrl_status>(b->eng.Step(dt)); } catch (...) { return static_cast<dtrl_status>(dtrl_on_exception(b)); }
dtrl_status dtrl_step_begin(dtrl_batch* b, double dt) try { CHECK_B(); return static_cast<dtrl_status>(b->eng.StepBegin(dt)); } catch (...) { return static_cast<dtrl_status>(dtrl_on_exception(b)); }
dtrl_status dtrl_step_end(dtrl_batch* b) try { CHECK_B(); return static_cast<dtrl_status>(b->eng.StepEnd()); } catch (...) { return static_cast<dtrl_status>(dtrl_on_exception(b)); }
dtrl_status dtrl_step_updates(dtrl_batch* b, int n) try { CHECK_B(); return static_cast<dtrl_status>(b->eng.StepUpdates(n)); } catch (...) { return static_cast<dtrl_status>(dtrl_on_exception(b)); }
dtrl_status dtrl_run_frames(dtrl_batch* b, int frames, double dt) try { CHECK_B(); return static_cast<dtrl_status>(b->eng.RunFrames(frames, dt)); } catch (...) { return static_cast<dtrl_status>(dtrl_on_exception(b)); }
dtrl_status dtrl_set_policy(dtrl_batch* b, const float* w, size_t n, const double* io, const double* is, const double* oo, const double* os)
try {
	CHECK_B(); return static_cast<dtrl_status>(b->eng.SetPolicy(w, n, io, is, oo, os));
} catch (...) { return static_cast<dtrl_status>(dtrl_on_exception(b)); }
dtrl_status dtrl_policy_num_params(const dtrl_batch* b, size_t* n)
try {
	CHECK_B(); if (!b->eng.cfg().has_policy_net) { *n = 0; return DTRL_OK; }
	*n = static_cast<size_t>(b->eng.cfg().user_num_params); return DTRL_OK;
} catch (...) { return static_cast<dtrl_status>(dtrl_on_exception(b)); }
dtrl_status dtrl_build_output_offset_scale(const dtrl_batch* b, double* out_off, double* out_scale)
try {
	CHECK_B();
	if (!b->eng.cfg().has_policy_net) return DTRL_ERR_ARG;
	std::vector<double> off, sc;
	dtrl::BuildOutputOffsetScale(b->eng.cfg().model, b->eng.cfg().net, off, sc);
	for (size_t i = 0; i < off.size(); ++i) { out_off[i] = off[i]; out_scale[i] = sc[i]; }
	return DTRL_OK;
} catch (...) { return static_cast<dtrl_status>(dtrl_on_exception(b)); }
dtrl_status dtrl_load_scale_file(dtrl_batch* b, const char* path) try { CHECK_B(); return static_cast<dtrl_status>(b->eng.LoadScaleFile(path)); } catch (...) { return static_cast<dtrl_status>(dtrl_on_exception(b)); }
dtrl_status dtrl_write_scale_file(dtrl_batch* b, const char* path) try { CHECK_B(); return static_cast<dtrl_status>(b->eng.WriteScaleFile(path)); } catch (...) { return static_cast<dtrl_status>(dtrl_on_exception(b)); }
dtrl_status dtrl_set_explore(dtrl_batch* b, int enable, double rate, double temp, double base_rate) try { CHECK_B(); return static_cast<dtrl_status>(b->eng.SetExplore(enable, rate, temp, base_rate)); } catch (...) { return static_cast<dtrl_status>(dtrl_on_exception(b)); }
dtrl_status dtrl_set_terrain_lerp(dtrl_batch* b, double lerp) try { CHECK_B(); return static_cast<dtrl_status>(b->eng.SetTerrainLerp(lerp)); } catch (...) { return static_cast<dtrl_status>(dtrl_on_exception(b)); }
dtrl_status dtrl_drain_tuples(dtrl_batch* b, float* rows, uint32_t* flags, int32_t* env_ids, int cap, int* out_n)
try {
	CHECK_B(); return static_cast<dtrl_status>(b->eng.DrainTuples(rows, flags, env_ids, cap, out_n));
} catch (...) { return static_cast<dtrl_status>(dtrl_on_exception(b)); }

dtrl_status dtrl_get_pose_vel(dtrl_batch* b, const int32_t* env_ids, int n, double* q, double* qd)
try {
	CHECK_B();
	std::vector<EnvState> st; int rc = b->eng.GetStates(env_ids, n, st);
	if (rc != DTRL_OK) return static_cast<dtrl_status>(rc);
	const int D = b->eng.cfg().model.D;
	for (int i = 0; i < n; ++i) for (int k = 0; k < D; ++k) { if (q) q[i * D + k] = st[i].q[k]; if (qd) qd[i * D + k] = st[i].qd[k]; }
	return DTRL_OK;
} catch (...) { return static_cast<dtrl_status>(dtrl_on_exception(b)); }
void* dtrl_side_stream(dtrl_batch* b, int k, double* start_delay_us) try { if (!b) return nullptr; return b->eng.SideStream(k, start_delay_us); } catch (...) { return nullptr; }
dtrl_status dtrl_command_action(dtrl_batch* b, const int32_t* env_ids, int n, const int32_t* action_ids) try { CHECK_B(); return static_cast<dtrl_status>(b->eng.CommandAction(env_ids, n, action_ids)); } catch (...) { return static_cast<dtrl_status>(dtrl_on_exception(b)); }
dtrl_status dtrl_set_pose_vel(dtrl_batch* b, const int32_t* env_ids, int n, const double* q, const double* qd) try { CHECK_B(); return static_cast<dtrl_status>(b->eng.SetPoseVel(env_ids, n, q, qd)); } catch (...) { return static_cast<dtrl_status>(dtrl_on_exception(b)); }
dtrl_status dtrl_get_contact_cache(dtrl_batch* b, const int32_t* env_ids, int n, int32_t* count, int32_t* ids, double* lambda) try { CHECK_B(); return static_cast<dtrl_status>(b->eng.GetContactCache(env_ids, n, count, ids, lambda)); } catch (...) { return static_cast<dtrl_status>(dtrl_on_exception(b)); }
dtrl_status dtrl_set_contact_cache(dtrl_batch* b, const int32_t* env_ids, int n, const int32_t* count, const int32_t* ids, const double* lambda) try { CHECK_B(); return static_cast<dtrl_status>(b->eng.SetContactCache(env_ids, n, count, ids, lambda)); } catch (...) { return static_cast<dtrl_status>(dtrl_on_exception(b)); }
dtrl_status dtrl_add_perturb(dtrl_batch* b, const int32_t* env_ids, int n, const int32_t* link, const double* local_pos, const double* force, const double* duration)
try {
	CHECK_B();
	if (!link || !force || !duration) { b->eng.set_error("dtrl_add_perturb: link, force and duration are required"); return DTRL_ERR_ARG; }
	return static_cast<dtrl_status>(b->eng.AddPerturb(env_ids, n, link, local_pos, force, duration));
} catch (...) { return static_cast<dtrl_status>(dtrl_on_exception(b)); }
dtrl_status dtrl_apply_rand_force(dtrl_batch* b, const int32_t* env_ids, int n, uint64_t seed) try { CHECK_B(); return static_cast<dtrl_status>(b->eng.ApplyRandForce(env_ids, n, seed)); } catch (...) { return static_cast<dtrl_status>(dtrl_on_exception(b)); }
dtrl_status dtrl_get_policy_output(dtrl_batch* b, const int32_t* env_ids, int n, double* y) try { CHECK_B(); return static_cast<dtrl_status>(b->eng.GetPolicyOutput(env_ids, n, y)); } catch (...) { return static_cast<dtrl_status>(dtrl_on_exception(b)); }
dtrl_status dtrl_get_poli_state(dtrl_batch* b, const int32_t* env_ids, int n, double* s) try { CHECK_B(); return static_cast<dtrl_status>(b->eng.GetPoliState(env_ids, n, s)); } catch (...) { return static_cast<dtrl_status>(dtrl_on_exception(b)); }

dtrl_status dtrl_get_flags(dtrl_batch* b, const int32_t* env_ids, int n, uint32_t* bits)
try {
	CHECK_B();
	std::vector<EnvState> st; int rc = b->eng.GetStates(env_ids, n, st);
	if (rc != DTRL_OK) return static_cast<dtrl_status>(rc);
	const int L = b->eng.cfg().model.L;
	for (int i = 0; i < n; ++i) {
		const EnvState& s = st[i];
		bool flipped = std::fabs(dtrl::wrap_pi(s.q[2])) > 3.14159265358979323846 * 0.8;
		bool fallen = s.sum_fall_contact > 0.25 || s.fail_fall_dist != 0 || flipped;
		uint32_t stumble_mask = (b->eng.cfg().model.char_type == 1)
			? ~((1u << dtrl::rRightToe) | (1u << dtrl::rLeftToe) | (1u << dtrl::rRightAnkle) | (1u << dtrl::rLeftAnkle))
			: ~((1u << dtrl::jToe) | (1u << dtrl::jFinger) | (1u << dtrl::jAnkle) | (1u << dtrl::jWrist));
		bool stumbled = (s.contact_bits & stumble_mask & ((1u << L) - 1u)) != 0;
		bool new_cycle = s.state == 0 && s.phase == 0;
		bits[i] = (fallen ? DTRL_FLAG_FALLEN : 0u) | (stumbled ? DTRL_FLAG_STUMBLED : 0u) | (new_cycle ? DTRL_FLAG_NEW_CYCLE : 0u) | (static_cast<uint32_t>(s.state) << DTRL_FLAG_STATE_SHIFT);
	}
	return DTRL_OK;
} catch (...) { return static_cast<dtrl_status>(dtrl_on_exception(b)); }
dtrl_status dtrl_get_torques(dtrl_batch* b, const int32_t* env_ids, int n, double* tau_ctrl, double* tau_applied)
try {
	CHECK_B();
	std::vector<EnvState> st; int rc = b->eng.GetStates(env_ids, n, st);
	if (rc != DTRL_OK) return static_cast<dtrl_status>(rc);
	const int D = b->eng.cfg().model.D;
	for (int i = 0; i < n; ++i) for (int k = 0; k < D; ++k) { if (tau_ctrl) tau_ctrl[i * D + k] = st[i].tau_ctrl[k]; if (tau_applied) tau_applied[i * D + k] = st[i].tau[k]; }
	return DTRL_OK;
} catch (...) { return static_cast<dtrl_status>(dtrl_on_exception(b)); }
dtrl_status dtrl_get_link_states(dtrl_batch* b, const int32_t* env_ids, int n, double* com_xy, double* com_vel_xy, double* angle)
try {
	CHECK_B();
	std::vector<EnvState> st; int rc = b->eng.GetStates(env_ids, n, st);
	if (rc != DTRL_OK) return static_cast<dtrl_status>(rc);
	const dtrl::DevModel& m = b->eng.cfg().model;
	const int L = m.L;
	for (int e = 0; e < n; ++e) {
		const EnvState& s = st[e];
		double phi[dtrl::kMaxL], w[dtrl::kMaxL], px[dtrl::kMaxL], py[dtrl::kMaxL], vx[dtrl::kMaxL], vy[dtrl::kMaxL];
		for (int j = 0; j < L; ++j) {   // parents precede children (cKinTree joint order)
			const int pa = m.parent[j];
			if (pa < 0) { phi[j] = s.q[2]; w[j] = s.qd[2]; px[j] = s.q[0]; py[j] = s.q[1]; vx[j] = s.qd[0]; vy[j] = s.qd[1]; }
			else {
				const double c = std::cos(phi[pa]), sn = std::sin(phi[pa]);
				const double rx = c * m.attach[j][0] - sn * m.attach[j][1], ry = sn * m.attach[j][0] + c * m.attach[j][1];
				phi[j] = phi[pa] + s.q[j + 2]; w[j] = w[pa] + s.qd[j + 2];
				px[j] = px[pa] + rx; py[j] = py[pa] + ry;
				vx[j] = vx[pa] - w[pa] * ry; vy[j] = vy[pa] + w[pa] * rx;
			}
			const double c = std::cos(phi[j]), sn = std::sin(phi[j]);
			const double bx = c * m.body_attach[j][0] - sn * m.body_attach[j][1], by = sn * m.body_attach[j][0] + c * m.body_attach[j][1];
			const size_t o = static_cast<size_t>(e) * L + j;
			if (com_xy) { com_xy[2 * o] = px[j] + bx; com_xy[2 * o + 1] = py[j] + by; }
			if (com_vel_xy) { com_vel_xy[2 * o] = vx[j] - w[j] * by; com_vel_xy[2 * o + 1] = vy[j] + w[j] * bx; }
			if (angle) angle[o] = phi[j] + m.body_theta[j];
		}
	}
	return DTRL_OK;
} catch (...) { return static_cast<dtrl_status>(dtrl_on_exception(b)); }
dtrl_status dtrl_get_contacts(dtrl_batch* b, const int32_t* env_ids, int n, int32_t* flags)
try {
	CHECK_B();
	std::vector<EnvState> st; int rc = b->eng.GetStates(env_ids, n, st);
	if (rc != DTRL_OK) return static_cast<dtrl_status>(rc);
	const int L = b->eng.cfg().model.L;
	for (int i = 0; i < n; ++i) for (int j = 0; j < L; ++j) flags[i * L + j] = (st[i].contact_bits >> j) & 1u;
	return DTRL_OK;
} catch (...) { return static_cast<dtrl_status>(dtrl_on_exception(b)); }
dtrl_status dtrl_get_ctrl(dtrl_batch* b, const int32_t* env_ids, int n, int32_t* state, double* phase, int32_t* action_id, double* params, double* pd_targets)
try {
	CHECK_B();
	std::vector<EnvState> st; int rc = b->eng.GetStates(env_ids, n, st);
	if (rc != DTRL_OK) return static_cast<dtrl_status>(rc);
	const int L = b->eng.cfg().model.L, P = b->eng.cfg().model.P;
	for (int i = 0; i < n; ++i) {
		if (state) state[i] = st[i].state;
		if (phase) phase[i] = st[i].phase;
		if (action_id) action_id[i] = st[i].action_id;
		if (params) for (int k = 0; k < P; ++k) params[i * P + k] = st[i].params[k];
		if (pd_targets) for (int j = 0; j < L; ++j) pd_targets[i * L + j] = st[i].pd_target[j];
	}
	return DTRL_OK;
} catch (...) { return static_cast<dtrl_status>(dtrl_on_exception(b)); }
dtrl_status dtrl_get_cycle_info(dtrl_batch* b, const int32_t* env_ids, int n, int64_t* num_cycles, int64_t* num_resets, double* cycle_start_com, double* cycle_start_time, double* opt_params)
try {
	CHECK_B();
	std::vector<EnvState> st; int rc = b->eng.GetStates(env_ids, n, st);
	if (rc != DTRL_OK) return static_cast<dtrl_status>(rc);
	const dtrl::DevModel& m = b->eng.cfg().model;
	for (int i = 0; i < n; ++i) {
		if (num_cycles) num_cycles[i] = st[i].num_cycles;
		if (num_resets) num_resets[i] = st[i].num_resets;
		if (cycle_start_com) { cycle_start_com[2 * i] = st[i].prev_com[0]; cycle_start_com[2 * i + 1] = st[i].prev_com[1]; }
		if (cycle_start_time) cycle_start_time[i] = st[i].time - st[i].curr_cycle_time;
		if (opt_params) for (int k = 0; k < m.n_opt; ++k) opt_params[static_cast<size_t>(i) * m.n_opt + k] = st[i].params[m.opt_index[k]];
	}
	return DTRL_OK;
} catch (...) { return static_cast<dtrl_status>(dtrl_on_exception(b)); }
dtrl_status dtrl_get_action_table(dtrl_batch* b, int* n_actions, double* table)
try {
	CHECK_B();
	const dtrl::DevModel& m = b->eng.cfg().model;
	if (n_actions) *n_actions = m.n_actions;
	if (table) for (int a = 0; a < m.n_actions; ++a) {
		const dtrl::real* p0 = m.ctrl_params[m.act_idx0[a]]; const dtrl::real* p1 = m.ctrl_params[m.act_idx1[a]]; const double bl = m.act_blend[a];
		for (int k = 0; k < m.n_opt; ++k) { const int i = m.opt_index[k]; table[static_cast<size_t>(a) * m.n_opt + k] = (1 - bl) * p0[i] + bl * p1[i]; }
	}
	return DTRL_OK;
} catch (...) { return static_cast<dtrl_status>(dtrl_on_exception(b)); }
dtrl_status dtrl_sample_ground(dtrl_batch* b, int env, int n, const double* x, double* h, int32_t* seg, int32_t* i, int32_t* j)
try {
	CHECK_B(); return static_cast<dtrl_status>(b->eng.SampleGround(env, n, x, h, seg, i, j));
} catch (...) { return static_cast<dtrl_status>(dtrl_on_exception(b)); }
dtrl_status dtrl_step_end_begin(dtrl_batch* b, double dt)
try {
	CHECK_B(); return static_cast<dtrl_status>(b->eng.StepEndBegin(dt));
} catch (...) { return static_cast<dtrl_status>(dtrl_on_exception(b)); }
dtrl_status dtrl_step_poll(dtrl_batch* b, double dt, int* relaunched)
try {
	CHECK_B(); return static_cast<dtrl_status>(b->eng.StepPoll(dt, relaunched));
} catch (...) { return static_cast<dtrl_status>(dtrl_on_exception(b)); }
dtrl_status dtrl_set_tuple_pipelining(dtrl_batch* b, int on)
try {
	CHECK_B(); return static_cast<dtrl_status>(b->eng.SetTuplePipelining(on != 0));
} catch (...) { return static_cast<dtrl_status>(dtrl_on_exception(b)); }
dtrl_status dtrl_drain_tuples_packed(dtrl_batch* b, float* block_dev, int block_rows, int* out_n)
try {
	CHECK_B(); return static_cast<dtrl_status>(b->eng.DrainTuplesPacked(block_dev, block_rows, out_n));
} catch (...) { return static_cast<dtrl_status>(dtrl_on_exception(b)); }
dtrl_status dtrl_get_ground_window(dtrl_batch* b, int env, int32_t* w2, double* min_x2, double* max_x2, float* heights0, float* heights1, int cap, int64_t* num_builds)
try {
	CHECK_B(); return static_cast<dtrl_status>(b->eng.GroundWindowRec(env, w2, min_x2, max_x2, heights0, heights1, cap, num_builds));
} catch (...) { return static_cast<dtrl_status>(dtrl_on_exception(b)); }
dtrl_status dtrl_eval_stats(dtrl_batch* b, double* avg_dist, int64_t* episodes, int64_t* cycles, int64_t* resets) try { CHECK_B(); return static_cast<dtrl_status>(b->eng.EvalStats(avg_dist, episodes, cycles, resets)); } catch (...) { return static_cast<dtrl_status>(dtrl_on_exception(b)); }
dtrl_status dtrl_dims(const dtrl_batch* b, int* L, int* D, int* S, int* A, int* P, int* nn_out, int* num_frags, int* frag_size)
try {
	CHECK_B();
	const dtrl::ScenarioConfig& c = b->eng.cfg();
	if (L) *L = c.model.L;
	if (D) *D = c.model.D;
	if (S) *S = b->eng.S();
	if (A) *A = b->eng.A();
	if (P) *P = c.model.P;
	if (nn_out) *nn_out = c.has_policy_net ? c.user_out_size : 0;
	if (num_frags) *num_frags = (c.has_policy_net && !c.actor_only) ? c.net.n_frags : 0;
	if (frag_size) *frag_size = c.model.n_opt;
	return DTRL_OK;
} catch (...) { return static_cast<dtrl_status>(dtrl_on_exception(b)); }
dtrl_status dtrl_kernel_time_ms(dtrl_batch* b, double* avg_ms, int64_t* launches) try { CHECK_B(); return static_cast<dtrl_status>(b->eng.KernelTime(avg_ms, launches)); } catch (...) { return static_cast<dtrl_status>(dtrl_on_exception(b)); }

dtrl_status dtrl_drain_tuples_device(dtrl_batch* b, float* rows_dev, uint32_t* flags_dev, int32_t* env_ids_dev, int cap, int* out_n)
try {
	CHECK_B(); return static_cast<dtrl_status>(b->eng.DrainTuples(rows_dev, flags_dev, env_ids_dev, cap, out_n, true));
} catch (...) { return static_cast<dtrl_status>(dtrl_on_exception(b)); }
dtrl_status dtrl_tuple_stats(dtrl_batch* b, int64_t* pending, int64_t* drained, int64_t* dropped, int32_t* capacity) try { CHECK_B(); return static_cast<dtrl_status>(b->eng.TupleStats(pending, drained, dropped, capacity)); } catch (...) { return static_cast<dtrl_status>(dtrl_on_exception(b)); }
dtrl_status dtrl_set_policy_device(dtrl_batch* b, const float* w_dev, size_t n, const double* io_dev, const double* is_dev, const double* oo_dev, const double* os_dev)
try {
	CHECK_B(); return static_cast<dtrl_status>(b->eng.SetPolicyDevice(w_dev, n, io_dev, is_dev, oo_dev, os_dev));
} catch (...) { return static_cast<dtrl_status>(dtrl_on_exception(b)); }
dtrl_status dtrl_set_policy_device_on(dtrl_batch* b, const float* w_dev, size_t n, void* stream)
try {
	CHECK_B(); return static_cast<dtrl_status>(b->eng.SetPolicyDevice(w_dev, n, nullptr, nullptr, nullptr, nullptr, stream));
} catch (...) { return static_cast<dtrl_status>(dtrl_on_exception(b)); }
dtrl_status dtrl_set_policy_device_async(dtrl_batch* b, const float* w_dev, size_t n, void* stream)
try {
	CHECK_B(); return static_cast<dtrl_status>(b->eng.SetPolicyDeviceAsync(w_dev, n, stream));
} catch (...) { return static_cast<dtrl_status>(dtrl_on_exception(b)); }
dtrl_status dtrl_get_dist_log(dtrl_batch* b, double* dist, int32_t* env_ids, int cap, int* out_n) try { CHECK_B(); return static_cast<dtrl_status>(b->eng.GetDistLog(dist, env_ids, cap, out_n)); } catch (...) { return static_cast<dtrl_status>(dtrl_on_exception(b)); }
dtrl_status dtrl_reset_avg_dist(dtrl_batch* b) try { CHECK_B(); return static_cast<dtrl_status>(b->eng.ResetAvgDist()); } catch (...) { return static_cast<dtrl_status>(dtrl_on_exception(b)); }
// cOptScenarioPoliEval::OutputResults (optimizer/scenarios/OptScenarioPoliEval.cpp:213-239): one line appended to `path`, the distances of every
// recorded episode pool member by pool member, std::to_string formatting, ", " separated
dtrl_status dtrl_write_dist_log(dtrl_batch* b, const char* path)
try {
	CHECK_B();
	if (!path) { b->eng.set_error("null path"); return DTRL_ERR_ARG; }
	int n = 0;
	int rc = b->eng.GetDistLog(nullptr, nullptr, 0, &n);
	if (rc != DTRL_OK) return static_cast<dtrl_status>(rc);
	std::vector<double> d(static_cast<size_t>(n));
	rc = b->eng.GetDistLog(d.data(), nullptr, n, &n);
	if (rc != DTRL_OK) return static_cast<dtrl_status>(rc);
	std::string str;
	for (int i = 0; i < n; ++i) { if (!str.empty()) str += ", "; str += std::to_string(d[i]); }
	str += "\n";
	FILE* f = std::fopen(path, "a");   // cFileUtil::AppendText
	if (!f) { b->eng.set_error(std::string("Failed to output results to ") + path); return DTRL_ERR_IO; }
	std::fputs(str.c_str(), f); std::fclose(f);
	return DTRL_OK;
} catch (...) { return static_cast<dtrl_status>(dtrl_on_exception(b)); }

dtrl_status dtrl_terrain_build(const char* type_name, const double* params40, uint64_t seed, double width, float* out, int cap, int* out_n, double* out_width)
try {
	if (!type_name || !params40 || !out_n || cap < 0 || (cap > 0 && !out)) return DTRL_ERR_ARG;
	std::string name = type_name;
	if (name.empty()) name = "flat";   // cTerrainGen2D::ParseType: "" == flat
	int type = -1;
	for (int i = 0; i < dtrl::kTerrTypeMax; ++i) if (name == dtrl::kTerrainTypeNames[i]) type = i;
	if (type < 0) { g_create_error = "unsupported terrain type " + name; return DTRL_ERR_ARG; }
	dtrl::TerrainRand rnd; rnd.Seed(static_cast<unsigned long>(seed));
	std::vector<float> data;
	const double w = dtrl::BuildTerrain(type, width, params40, rnd, data);
	*out_n = static_cast<int>(data.size());
	if (out_width) *out_width = w;
	for (int i = 0; i < *out_n && i < cap; ++i) out[i] = data[i];
	return *out_n <= cap ? DTRL_OK : DTRL_ERR_CAPACITY;
} catch (...) { return static_cast<dtrl_status>(dtrl_on_exception(nullptr)); }
dtrl_status dtrl_terrain_load_file(const char* path, char* type_out, int type_cap, double* params_out, int max_sets, int* out_sets)
try {
	if (!path || !out_sets || max_sets < 0 || (max_sets > 0 && !params_out)) return DTRL_ERR_ARG;
	dtrl::Json tf; std::string err;
	if (!dtrl::Json::parse_file(path, tf, err)) { g_create_error = err; return DTRL_ERR_IO; }
	const dtrl::Json* ty = tf.find("Type");
	const std::string tname = ty ? ty->str : "";
	if (type_out && type_cap > 0) { std::snprintf(type_out, static_cast<size_t>(type_cap), "%s", tname.c_str()); }
	int n = 0;
	const dtrl::Json* ps = tf.find("Params");
	if (ps) for (const dtrl::Json& obj : ps->arr) {
		if (n >= max_sets) break;
		for (int k = 0; k < dtrl::kNumTerrainParams; ++k) params_out[n * dtrl::kNumTerrainParams + k] = obj.get_num(dtrl::kTerrainParamNames[k], dtrl::kTerrainParamDefaults[k]);
		++n;
	}
	*out_sets = n;
	return DTRL_OK;
} catch (...) { return static_cast<dtrl_status>(dtrl_on_exception(nullptr)); }
dtrl_status dtrl_args_parse_string(const char* const* argv, int argc, const char* key, char* out, int cap, int* found, int* n_tokens)
try {
	if (!key || !found || argc < 0 || (argc > 0 && !argv)) return DTRL_ERR_ARG;
	dtrl::ArgParser args(argv, argc);
	std::string arg_file;
	if (args.ParseString("arg_file", arg_file)) {
		std::string root; args.ParseString("data_root", root);
		const std::string path = (arg_file.empty() || arg_file[0] == '/' || root.empty()) ? arg_file : root + "/" + arg_file;
		if (!args.AppendArgs(path)) { g_create_error = "Failed to load args from: " + path; return DTRL_ERR_IO; }
	}
	std::string v;
	*found = args.ParseString(key, v) ? 1 : 0;
	if (out && cap > 0) std::snprintf(out, static_cast<size_t>(cap), "%s", *found ? v.c_str() : "");
	if (n_tokens) *n_tokens = args.GetNumArgs();
	return DTRL_OK;
} catch (...) { return static_cast<dtrl_status>(dtrl_on_exception(nullptr)); }

// not part of include/dtrl.h: developer hook used by tools/gpu_sections.py with the DTRL_PROFILE build
int dtrlx_profile_env(dtrl_batch* b, int section, unsigned long long* out, int cap) try { return b ? b->eng.ProfileEnv(section, out, cap) : 1; } catch (...) { return static_cast<dtrl_status>(dtrl_on_exception(b)); }
int dtrlx_profile_sections(dtrl_batch* b, unsigned long long* out, int cap) try { return b ? b->eng.ProfileSections(out, cap) : 1; } catch (...) { return static_cast<dtrl_status>(dtrl_on_exception(b)); }

}  // extern "C"
