// ORACLE (test infrastructure, NOT product code) -- C entry points for ctypes.
//
// CPU fp64 restatement of the rollout hot path of xbpeng/DeepTerrainRL (see or_*.h headers for the
// reference file:line each function follows). PARITY STATUS: the controller / terrain / feature / MACE code is a
// function-by-function restatement checked against every known-answer the reference ships for this path
// (tests/test_oracle_kat.py: structure counts, masses, policy I/O sizes, the shipped *_scale.txt output
// offset/scale); the rigid-body integration behind cWorld::Update lives in Bullet, an un-vendored, un-pinned
// external that is absent here, so that part is a documented model (or_sim.h) and is "parity unpinned".
//
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
#include "or_ctrl.h"
#include <atomic>
#include <chrono>
#include <thread>

using namespace orc;

struct OrcHandle {
	Env env;
	PolicyNet net;
};

extern "C" {

void* orc_create(const OrcModel* m, uint64_t terrain_seed, uint64_t rng_seed, uint64_t env_id)
{
	OrcHandle* h = new OrcHandle();
	h->env.net = &h->net;
	h->env.Init(*m, terrain_seed, rng_seed, env_id);
	return h;
}
void orc_destroy(void* p) { delete static_cast<OrcHandle*>(p); }

// policy must be set BEFORE orc_create's Init consumed RNG? No: AssignFragID depends on the net being present,
// so a policy-carrying env is built with orc_create_with_policy.
void* orc_create_with_policy(const OrcModel* m, uint64_t terrain_seed, uint64_t rng_seed, uint64_t env_id, const OrcNetDesc* d,
							 const float* weights, const double* in_off, const double* in_scale, const double* out_off, const double* out_scale)
{
	OrcHandle* h = new OrcHandle();
	h->net.d = *d;
	size_t n = PolicyNet::NumParams(*d);
	h->net.w.assign(weights, weights + n);
	int ni = h->net.InSize(), no = h->net.OutSize();
	h->net.in_off.assign(in_off, in_off + ni); h->net.in_scale.assign(in_scale, in_scale + ni);
	h->net.out_off.assign(out_off, out_off + no); h->net.out_scale.assign(out_scale, out_scale + no);
	h->net.valid = true;
	h->env.net = &h->net;
	h->env.Init(*m, terrain_seed, rng_seed, env_id);
	return h;
}
uint64_t orc_net_num_params(const OrcNetDesc* d) { return PolicyNet::NumParams(*d); }

void orc_reset(void* p) { static_cast<OrcHandle*>(p)->env.Reset(); }
void orc_update(void* p, double dt) { static_cast<OrcHandle*>(p)->env.Update(dt); }
void orc_step(void* p, int n)
{
	Env& e = static_cast<OrcHandle*>(p)->env;
	double dt = (1.0 / 30.0) / e.M.num_update_steps;
	for (int i = 0; i < n; ++i) e.EnvStep(dt);
}
void orc_dims(void* p, int* L, int* D, int* S, int* A, int* P)
{
	Env& e = static_cast<OrcHandle*>(p)->env;
	*L = e.L; *D = e.D; *S = e.PoliStateSize(); *A = e.PoliActionSize(); *P = e.P;
}
void orc_get_pose_vel(void* p, double* q, double* qd)
{
	Env& e = static_cast<OrcHandle*>(p)->env;
	for (int i = 0; i < e.D; ++i) { q[i] = e.q[i]; qd[i] = e.qd[i]; }
}
void orc_set_pose_vel(void* p, const double* q, const double* qd)
{
	Env& e = static_cast<OrcHandle*>(p)->env;
	for (int i = 0; i < e.D; ++i) { e.q[i] = q[i]; e.qd[i] = qd[i]; }
	ForwardKin(e.M, e.q, e.qd, e.B);
	e.integ.ResetWarmStart();   // a teleported character drops its persistent contact rows, as dtrl_set_pose_vel does (orc_set_warm afterwards restores a saved set)
}
// the integrator's persistent contact rows (or_sim.h Integrator::prev_*): identities and the impulses the last substep ended with; 24 slots
int orc_get_warm(void* p, int32_t* ids, double* lam)
{
	Env& e = static_cast<OrcHandle*>(p)->env;
	for (int k = 0; k < SimConst::max_rows; ++k) { ids[k] = k < e.integ.prev_R ? e.integ.prev_id[k] : 0xffff; lam[k] = k < e.integ.prev_R ? e.integ.prev_lam[k] : 0.0; }
	return e.integ.prev_R;
}
void orc_set_warm(void* p, int R, const int32_t* ids, const double* lam)
{
	Env& e = static_cast<OrcHandle*>(p)->env;
	e.integ.prev_R = R;
	for (int k = 0; k < R && k < SimConst::max_rows; ++k) { e.integ.prev_id[k] = ids[k]; e.integ.prev_lam[k] = lam[k]; }
}
void orc_add_perturb(void* p, int link, double lx, double ly, double fx, double fy, double dur) { static_cast<OrcHandle*>(p)->env.AddPerturb(link, lx, ly, fx, fy, dur); }
void orc_get_tau(void* p, double* tau_ctrl, double* tau_applied)
{
	Env& e = static_cast<OrcHandle*>(p)->env;
	for (int i = 0; i < e.D; ++i) { tau_ctrl[i] = e.tau_ctrl[i]; tau_applied[i] = e.tau_applied[i]; }
}
void orc_get_contacts(void* p, int32_t* flags)
{
	Env& e = static_cast<OrcHandle*>(p)->env;
	for (int j = 0; j < e.L; ++j) flags[j] = e.contact[j] ? 1 : 0;
}
// bits: 0 fallen, 1 stumbled, 2 new_cycle, 8.. fsm state
uint32_t orc_get_flags(void* p)
{
	Env& e = static_cast<OrcHandle*>(p)->env;
	return (e.HasFallen() ? 1u : 0u) | (e.HasStumbled() ? 2u : 0u) | (e.IsNewCycle() ? 4u : 0u) | (static_cast<uint32_t>(e.state) << 8);
}
void orc_get_ctrl(void* p, int* state, double* phase, int* action_id, double* params, double* pd_targets)
{
	Env& e = static_cast<OrcHandle*>(p)->env;
	*state = e.state; *phase = e.phase; *action_id = e.curr.id;
	for (int i = 0; i < e.P; ++i) params[i] = e.curr.params[i];
	for (int j = 0; j < e.L; ++j) pd_targets[j] = e.pd_target[j];
}
void orc_get_poli_state(void* p, double* s)
{
	Env& e = static_cast<OrcHandle*>(p)->env;
	for (size_t i = 0; i < e.poli_state.size(); ++i) s[i] = e.poli_state[i];
}
void orc_get_bodies(void* p, double* com_xy, double* com_vel_xy, double* psi)
{
	Env& e = static_cast<OrcHandle*>(p)->env;
	for (int j = 0; j < e.L; ++j) { com_xy[2 * j] = e.B.cx[j]; com_xy[2 * j + 1] = e.B.cy[j]; com_vel_xy[2 * j] = e.B.vcx[j]; com_vel_xy[2 * j + 1] = e.B.vcy[j]; psi[j] = e.B.psi[j]; }
}
void orc_stats(void* p, int64_t* resets, int64_t* cycles, int64_t* episodes, double* avg_dist, int64_t* terrain_builds)
{
	Env& e = static_cast<OrcHandle*>(p)->env;
	*resets = e.num_resets; *cycles = e.num_cycles; *episodes = e.num_episodes; *avg_dist = e.avg_dist; *terrain_builds = e.ground.num_builds;
}
// cCharController::CommandAction: replaces whatever is queued (the reference keeps a stack; the scenarios only ever queue one)
void orc_command_action(void* p, int a) { Env& e = static_cast<OrcHandle*>(p)->env; e.commands.clear(); e.commands.push_back(a); }
void orc_pair_distances(void* p, double* out)
{
	Env& e = static_cast<OrcHandle*>(p)->env;
	ForwardKin(e.M, e.q, e.qd, e.B);
	PairDistances(e.M, e.B, out);
}
void orc_frame_end(void* p) { static_cast<OrcHandle*>(p)->env.FrameEnd(); }
// signed separation of every contact sample point [L * 6] at the current configuration (what a narrowphase would report as manifold distances)
void orc_contact_distances(void* p, double* out)
{
	Env& e = static_cast<OrcHandle*>(p)->env;
	ForwardKin(e.M, e.q, e.qd, e.B);
	ContactDistances(e.M, e.B, e.ground, out);
}
double orc_time(void* p) { return static_cast<OrcHandle*>(p)->env.time; }
// cScenarioPoliEval::GetDistLog (scenarios/ScenarioPoliEval.cpp:147-150)
int orc_dist_log(void* p, double* out, int cap)
{
	Env& e = static_cast<OrcHandle*>(p)->env;
	const int n = static_cast<int>(e.dist_log.size());
	for (int i = 0; i < n && i < cap; ++i) out[i] = e.dist_log[i];
	return n;
}
// rows in the MACE replay layout [r | s | a | s'] (learning/MACETrainer.cpp:373-401)
int orc_drain_tuples(void* p, float* rows, uint32_t* flags, int cap)
{
	Env& e = static_cast<OrcHandle*>(p)->env;
	int S = e.PoliStateSize(), A = e.PoliActionSize(), W = 1 + 2 * S + A;
	int n = 0;
	for (; n < static_cast<int>(e.tuples.size()) && n < cap; ++n) {
		const Tuple& t = e.tuples[n];
		float* r = rows + static_cast<size_t>(n) * W;
		r[0] = static_cast<float>(t.reward);
		for (int i = 0; i < S; ++i) r[1 + i] = static_cast<float>(i < static_cast<int>(t.s0.size()) ? t.s0[i] : 0);
		for (int i = 0; i < A; ++i) r[1 + S + i] = static_cast<float>(t.a[i]);
		for (int i = 0; i < S; ++i) r[1 + S + A + i] = static_cast<float>(t.s1[i]);
		flags[n] = t.flags;
	}
	e.tuples.erase(e.tuples.begin(), e.tuples.begin() + n);
	return n;
}
int orc_drain_tuples_f64(void* p, double* rows, uint32_t* flags, int cap)
{
	Env& e = static_cast<OrcHandle*>(p)->env;
	int S = e.PoliStateSize(), A = e.PoliActionSize(), W = 1 + 2 * S + A;
	int n = 0;
	for (; n < static_cast<int>(e.tuples.size()) && n < cap; ++n) {
		const Tuple& t = e.tuples[n];
		double* r = rows + static_cast<size_t>(n) * W;
		r[0] = t.reward;
		for (int i = 0; i < S; ++i) r[1 + i] = (i < static_cast<int>(t.s0.size()) ? t.s0[i] : 0);
		for (int i = 0; i < A; ++i) r[1 + S + i] = t.a[i];
		for (int i = 0; i < S; ++i) r[1 + S + A + i] = t.s1[i];
		flags[n] = t.flags;
	}
	e.tuples.erase(e.tuples.begin(), e.tuples.begin() + n);
	return n;
}

// ---- known-answer helpers -----------------------------------------------------------------------
// H (D*D row-major), C_quirk (reference BuildBiasForce), C_true (textbook bias used by the integrator), tau_g (CalcGravityForce)
void orc_rbd(void* p, const double* q, const double* qd, double* H, double* C_quirk, double* C_true, double* grav)
{
	Env& e = static_cast<OrcHandle*>(p)->env;
	RBDModel rbd; rbd.Init(&e.M);
	int D = rbd.D;
	rbd.Update(q, qd, false);
	for (int i = 0; i < D; ++i) { for (int k = 0; k < D; ++k) H[i * D + k] = rbd.H[i][k]; C_quirk[i] = rbd.C[i]; }
	rbd.CalcGravityForce(grav);
	rbd.Update(q, qd, true);
	for (int i = 0; i < D; ++i) C_true[i] = rbd.C[i];
}
// controller torque for a given (q, qd, contacts, state) without advancing anything: used for per-function parity
void orc_ctrl_eval(void* p, const double* q, const double* qd, const int32_t* contacts, int fsm_state, double phase, double* tau_ctrl, double* tau_applied)
{
	OrcHandle* h = static_cast<OrcHandle*>(p);
	Env e = h->env;  // copy
	e.net = nullptr;
	for (int i = 0; i < e.D; ++i) { e.q[i] = q[i]; e.qd[i] = qd[i]; }
	ForwardKin(e.M, e.q, e.qd, e.B);
	for (int j = 0; j < e.L; ++j) e.contact[j] = contacts[j] != 0;
	e.first_cycle = false;
	e.state = fsm_state; e.phase = phase; e.SetStateParams();
	// phase advances by dt/trans_time inside ControllerUpdate; callers pass a phase that does not trigger a transition
	e.ControllerUpdate((1.0 / 30.0) / e.M.num_update_steps);
	for (int i = 0; i < e.D; ++i) { tau_ctrl[i] = e.tau_ctrl[i]; tau_applied[i] = e.tau_applied[i]; }
}
int orc_terrain_build(int type, const double* params40, uint64_t seed, double width, float* out, int cap)
{
	Rand r; r.Seed(static_cast<unsigned long>(seed));
	std::vector<float> d;
	TerrainGen::Build(type, width, params40, r, d);
	int n = static_cast<int>(d.size());
	for (int i = 0; i < n && i < cap; ++i) out[i] = d[i];
	return n;
}
// sample the env's ground: height, valid flag, segment slot (0 = min segment), grid indices i, j
double orc_sample_ground(void* p, double x, int32_t* valid, int32_t* seg, int32_t* oi, int32_t* oj)
{
	Env& e = static_cast<OrcHandle*>(p)->env;
	bool v; int s, i, j;
	double h = e.ground.SampleHeight(x, &v, &s, &i, &j);
	*valid = v; *seg = s; *oi = i; *oj = j;
	return h;
}
int orc_ground_segment(void* p, int slot, float* out, int cap, double* min_x, double* origin_x, double* scale_x)
{
	Env& e = static_cast<OrcHandle*>(p)->env;
	const Segment& s = e.ground.Seg(slot);
	int n = s.W();
	for (int i = 0; i < n && i < cap; ++i) out[i] = s.data[i];
	*min_x = s.min_x; *origin_x = s.origin_x; *scale_x = s.scale_x;
	return n;
}
void orc_nn_eval(void* p, const double* x, double* y) { static_cast<OrcHandle*>(p)->net.Eval(x, y); }
void orc_rng_draw(uint64_t seed, uint64_t env_id, int n, double* out)
{
	EnvRng r; r.Seed(seed, env_id);
	for (int i = 0; i < n; ++i) out[i] = r.RandDouble();
}

// ---- CPU baseline: T threads x (N/T envs each), mirroring cScenarioTrain::Run's one-thread-per-scene
// (scenarios/ScenarioTrain.cpp:100-115). Returns env-steps per second. ------------------------------
double orc_batch_run(const OrcModel* m, int n_envs, int n_threads, int n_frames, uint64_t terrain_seed0, uint64_t rng_seed,
					 const OrcNetDesc* d, const float* weights, const double* in_off, const double* in_scale, const double* out_off, const double* out_scale,
					 int64_t* out_resets, int64_t* out_cycles)
{
	std::vector<OrcHandle*> hs(n_envs);
	for (int i = 0; i < n_envs; ++i) {
		hs[i] = static_cast<OrcHandle*>(d ? orc_create_with_policy(m, terrain_seed0 + i, rng_seed, i, d, weights, in_off, in_scale, out_off, out_scale)
										  : orc_create(m, terrain_seed0 + i, rng_seed, i));
	}
	auto t0 = std::chrono::steady_clock::now();
	std::vector<std::thread> th;
	for (int t = 0; t < n_threads; ++t) {
		th.emplace_back([&, t]() {
			for (int i = t; i < n_envs; i += n_threads) for (int f = 0; f < n_frames; ++f) hs[i]->env.Update(1.0 / 30.0);
		});
	}
	for (auto& x : th) x.join();
	double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
	int64_t resets = 0, cycles = 0;
	for (int i = 0; i < n_envs; ++i) { resets += hs[i]->env.num_resets; cycles += hs[i]->env.num_cycles; delete hs[i]; }
	if (out_resets) *out_resets = resets;
	if (out_cycles) *out_cycles = cycles;
	return static_cast<double>(n_envs) * n_frames * m->num_update_steps / sec;
}

// distribution-level statistics of a poli_eval / exp run on n_envs oracle envs (n_threads host threads): out[0] resets, [1] cycles, [2] episodes,
// [3] sum of episode distances, [4] sum of squared episode distances, [5] env-steps; returns wall seconds
double orc_batch_eval(const OrcModel* m, int n_envs, int n_threads, int n_frames, uint64_t terrain_seed0, uint64_t rng_seed, uint64_t env_id0,
					  const OrcNetDesc* d, const float* weights, const double* in_off, const double* in_scale, const double* out_off, const double* out_scale, double* out)
{
	std::vector<OrcHandle*> hs(n_envs);
	for (int i = 0; i < n_envs; ++i)
		hs[i] = static_cast<OrcHandle*>(d ? orc_create_with_policy(m, terrain_seed0 + i, rng_seed, env_id0 + i, d, weights, in_off, in_scale, out_off, out_scale)
										  : orc_create(m, terrain_seed0 + i, rng_seed, env_id0 + i));
	auto t0 = std::chrono::steady_clock::now();
	std::vector<std::thread> th;
	for (int t = 0; t < n_threads; ++t) th.emplace_back([&, t]() { for (int i = t; i < n_envs; i += n_threads) for (int f = 0; f < n_frames; ++f) hs[i]->env.Update(1.0 / 30.0); });
	for (auto& x : th) x.join();
	const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
	for (int k = 0; k < 6; ++k) out[k] = 0;
	for (int i = 0; i < n_envs; ++i) {
		const Env& e = hs[i]->env;
		out[0] += e.num_resets; out[1] += e.num_cycles; out[2] += e.num_episodes;
		for (double x : e.dist_log) { out[3] += x; out[4] += x * x; }
		delete hs[i];
	}
	out[5] = static_cast<double>(n_envs) * n_frames * m->num_update_steps;
	return sec;
}

// full-width parity record (tests/test_gpu_parity.py::test_config*_full_width_*): n_envs free-running oracle envs on n_threads host threads; after every outer
// frame f the pose and velocity of env i go to out_q / out_qd [n_frames][n_envs][D]; out_diag [n_envs][8] = the integrator's diagnostic counters (or_sim.h
// Integrator::diag, 7 entries) + the env's reset count.
// out_ws_n [n_frames][n_envs], out_ws_id / out_ws_lam [n_frames][n_envs][24] (may be
// null): the persistent contact rows after every frame (orc_get_warm); out_resets [n_frames][n_envs] (may be null): the env's reset count so far. nudge != 0: every
// env starts with its first joint angle moved by that much (the oracle's own sensitivity to a perturbation below rounding: the chaos floor). Returns wall seconds.
double orc_batch_trace(const OrcModel* m, int n_envs, int n_threads, int n_frames, uint64_t terrain_seed0, uint64_t rng_seed, uint64_t env_id0,
					   const OrcNetDesc* d, const float* weights, const double* in_off, const double* in_scale, const double* out_off, const double* out_scale,
					   double* out_q, double* out_qd, int64_t* out_diag, int32_t* out_ws_n, int32_t* out_ws_id, double* out_ws_lam, int32_t* out_resets, double nudge)
{
	std::vector<OrcHandle*> hs(n_envs);
	for (int i = 0; i < n_envs; ++i)
		hs[i] = static_cast<OrcHandle*>(d ? orc_create_with_policy(m, terrain_seed0 + i, rng_seed, env_id0 + i, d, weights, in_off, in_scale, out_off, out_scale)
										  : orc_create(m, terrain_seed0 + i, rng_seed, env_id0 + i));
	const int D = hs.empty() ? 0 : hs[0]->env.D;
	auto t0 = std::chrono::steady_clock::now();
	std::vector<std::thread> th;
	for (int t = 0; t < n_threads; ++t) th.emplace_back([&, t]() {
		for (int i = t; i < n_envs; i += n_threads) {
			Env& e = hs[i]->env;
			if (nudge != 0) { e.q[3] += nudge; ForwardKin(e.M, e.q, e.qd, e.B); }   // sensitivity probe: the first joint angle moved by `nudge` (1e-13: below the last bit that matters anywhere else)
			for (int f = 0; f < n_frames; ++f) {
				e.Update(1.0 / 30.0);
				double* q = out_q + (static_cast<size_t>(f) * n_envs + i) * D;
				double* qd = out_qd + (static_cast<size_t>(f) * n_envs + i) * D;
				for (int k = 0; k < D; ++k) { q[k] = e.q[k]; qd[k] = e.qd[k]; }
				const size_t fi = static_cast<size_t>(f) * n_envs + i;
				if (out_ws_n) out_ws_n[fi] = orc_get_warm(hs[i], out_ws_id + fi * SimConst::max_rows, out_ws_lam + fi * SimConst::max_rows);
				if (out_resets) out_resets[fi] = static_cast<int32_t>(e.num_resets);
			}
			for (int k = 0; k < 7; ++k) out_diag[i * 8 + k] = e.integ.diag[k];
			out_diag[i * 8 + 7] = e.num_resets;
		}
	});
	for (auto& x : th) x.join();
	const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
	for (int i = 0; i < n_envs; ++i) delete hs[i];
	return sec;
}

}  // extern "C"
