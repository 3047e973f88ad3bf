// ORACLE (test infrastructure, NOT product code).
// CPU fp64 restatement of the reference's character controller stack for the dog/goat and the raptor
// (sim/RaptorController.cpp:195-233 Update, :804-849 UpdateState, :899-983 stance hip / swing + stance feedback,
//  :985-1075 gravity compensation + virtual forces, :561-600 reward, :1414-1488 stance-mirrored policy state,
//  :1434-1446 FlipStance/SetStance; sim/SimRaptor.cpp:78-153 stumble / fall parts):
//   sim/DogController.cpp:229-268 Update, :805-845 UpdateState, :847-868 UpdateAction, :894-945 feedback,
//   :947-995 gravity compensation, :997-1029 virtual forces, :1042-1054 SetStateParams, :1120-1175 contact basis,
//   :594-628 CalcReward, :1323-1352 NewCycleUpdate/BlendCtrlParams/PostProcessParams, :1372-1393 contact pos/dist
//   sim/ImpPDController.cpp:234-310 stable PD,  sim/PDController.cpp:181-224 CalcTheta/CalcVel
//   sim/TerrainRLCharController.cpp:168-285 ParseGround/BuildPoliState, :120-146 ApplyAction
//   sim/BaseControllerMACE.cpp:58-68,254-318,339-396,437-518 MACE action selection
//   sim/DogControllerMACE.cpp:26-91 MACE overrides, AssignFragID
//   sim/SimDog.cpp:83-162, sim/SimCharSoftFall.cpp:46-125 stumble / fall detection
//   learning/NeuralNet.cpp:352-375,977-986,1027-1036 Eval = normalise -> forward -> unnormalise
//   data/policies/dog/nets/dog_mace3_deploy.prototxt network topology (Caffe itself is an absent external;
//   convolution/inner-product/ReLU are restated from their definitions, cross-correlation as in Caffe).
#pragma once
#include "or_sim.h"
#include <cstdint>

namespace orc {

// ---------------------------------------------------------------------------------------------
// Exploration RNG. The reference uses one unsynchronised, time-seeded global std::default_random_engine
// shared by all env threads (util/MathUtil.cpp:4), so its streams are not reproducible
// even by itself; the engine (oracle and product alike) uses a counter-based per-env stream instead.
struct EnvRng {
	uint64_t key = 0, ctr = 0;
	static uint64_t Mix(uint64_t x)
	{
		x += 0x9E3779B97F4A7C15ULL;
		x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ULL;
		x = (x ^ (x >> 27)) * 0x94D049BB133111EBULL;
		return x ^ (x >> 31);
	}
	void Seed(uint64_t seed, uint64_t env_id) { key = Mix(seed ^ Mix(env_id)); ctr = 0; }
	double RandDouble() { uint64_t z = Mix(key + ctr * 0xD1342543DE82EF95ULL); ++ctr; return static_cast<double>(z >> 11) * (1.0 / 9007199254740992.0); }
	double RandDouble(double mn, double mx) { return mn + RandDouble() * (mx - mn); }
	int RandInt(int mn, int mx) { if (mn == mx) return mn; int r = mn + static_cast<int>(RandDouble() * (mx - mn)); return r >= mx ? mx - 1 : r; }
	bool FlipCoin() { return RandDouble() < 0.5; }
	double RandNorm(double mean, double stdev)
	{
		double u1 = 1.0 - RandDouble(), u2 = RandDouble();
		return mean + stdev * std::sqrt(-2.0 * std::log(u1)) * std::cos(6.283185307179586476925286766559 * u2);
	}
	// util/Rand.cpp:63-87
	int RandIntExclude(int mn, int mx, int exc)
	{
		if (exc < mn || exc >= mx) return RandInt(mn, mx);
		int new_max = mx - 1;
		if (new_max <= mn) return mn;
		int r = RandInt(mn, new_max);
		if (r >= exc) ++r;
		return r;
	}
};

// ---------------------------------------------------------------------------------------------
struct PolicyNet {
	bool valid = false;
	OrcNetDesc d{};
	std::vector<double> w;  // flat, Caffe blob order (see DESIGN.md "policy weight layout")
	std::vector<double> in_off, in_scale, out_off, out_scale;
	int InSize() const { return d.n_terrain + d.n_char; }
	int OutSize() const { return d.n_frags + d.n_frags * d.frag_size; }
	static size_t NumParams(const OrcNetDesc& d)
	{
		size_t n = 0; int cin = 1, wdt = d.n_terrain;
		for (int l = 0; l < 3; ++l) { n += static_cast<size_t>(d.conv_ch[l]) * cin * d.conv_k[l] + d.conv_ch[l]; cin = d.conv_ch[l]; wdt = wdt - d.conv_k[l] + 1; }
		n += static_cast<size_t>(d.fc_terr) * cin * wdt + d.fc_terr;
		n += static_cast<size_t>(d.fc_trunk) * (d.fc_terr + d.n_char) + d.fc_trunk;
		n += static_cast<size_t>(d.fc_head) * d.fc_trunk + d.fc_head + static_cast<size_t>(d.n_frags) * d.fc_head + d.n_frags;
		for (int f = 0; f < d.n_frags; ++f) n += static_cast<size_t>(d.fc_head) * d.fc_trunk + d.fc_head + static_cast<size_t>(d.frag_size) * d.fc_head + d.frag_size;
		return n;
	}
	static void FC(const double* W, const double* b, int nout, int nin, const double* x, double* y, bool relu)
	{
		for (int o = 0; o < nout; ++o) {
			double s = b[o];
			const double* wr = W + static_cast<size_t>(o) * nin;
			for (int i = 0; i < nin; ++i) s += wr[i] * x[i];
			y[o] = (relu && s < 0) ? 0 : s;
		}
	}
	void Eval(const double* x_raw, double* y) const
	{
		const int nin = InSize();
		std::vector<double> x(nin);
		for (int i = 0; i < nin; ++i) x[i] = (x_raw[i] + in_off[i]) * in_scale[i];
		const double* p = w.data();
		std::vector<double> a(x.begin(), x.begin() + d.n_terrain), bvec;
		int cin = 1, wdt = d.n_terrain;
		for (int l = 0; l < 3; ++l) {
			int co = d.conv_ch[l], k = d.conv_k[l], wo = wdt - k + 1;
			const double* W = p; const double* b = p + static_cast<size_t>(co) * cin * k;
			bvec.assign(static_cast<size_t>(co) * wo, 0.0);
			for (int o = 0; o < co; ++o) for (int t = 0; t < wo; ++t) {
				double s = b[o];
				for (int c = 0; c < cin; ++c) for (int u = 0; u < k; ++u) s += W[(static_cast<size_t>(o) * cin + c) * k + u] * a[static_cast<size_t>(c) * wdt + t + u];
				bvec[static_cast<size_t>(o) * wo + t] = s < 0 ? 0 : s;
			}
			p = b + co; a.swap(bvec); cin = co; wdt = wo;
		}
		int nflat = cin * wdt;
		std::vector<double> trunk_in(d.fc_terr + d.n_char);
		FC(p, p + static_cast<size_t>(d.fc_terr) * nflat, d.fc_terr, nflat, a.data(), trunk_in.data(), true);
		p += static_cast<size_t>(d.fc_terr) * nflat + d.fc_terr;
		for (int i = 0; i < d.n_char; ++i) trunk_in[d.fc_terr + i] = x[d.n_terrain + i];
		int ntr = d.fc_terr + d.n_char;
		std::vector<double> trunk(d.fc_trunk), head(d.fc_head);
		FC(p, p + static_cast<size_t>(d.fc_trunk) * ntr, d.fc_trunk, ntr, trunk_in.data(), trunk.data(), true);
		p += static_cast<size_t>(d.fc_trunk) * ntr + d.fc_trunk;
		// critic head -> outputs [0, n_frags)
		FC(p, p + static_cast<size_t>(d.fc_head) * d.fc_trunk, d.fc_head, d.fc_trunk, trunk.data(), head.data(), true);
		p += static_cast<size_t>(d.fc_head) * d.fc_trunk + d.fc_head;
		FC(p, p + static_cast<size_t>(d.n_frags) * d.fc_head, d.n_frags, d.fc_head, head.data(), y, false);
		p += static_cast<size_t>(d.n_frags) * d.fc_head + d.n_frags;
		for (int f = 0; f < d.n_frags; ++f) {
			FC(p, p + static_cast<size_t>(d.fc_head) * d.fc_trunk, d.fc_head, d.fc_trunk, trunk.data(), head.data(), true);
			p += static_cast<size_t>(d.fc_head) * d.fc_trunk + d.fc_head;
			FC(p, p + static_cast<size_t>(d.frag_size) * d.fc_head, d.frag_size, d.fc_head, head.data(), y + d.n_frags + f * d.frag_size, false);
			p += static_cast<size_t>(d.frag_size) * d.fc_head + d.frag_size;
		}
		int nout = OutSize();
		for (int i = 0; i < nout; ++i) y[i] = y[i] / out_scale[i] - out_off[i];
	}
};

inline double WrapPi(double a)
{
	// Bullet quaternion axis-angle / btHingeConstraint::getHingeAngle report angles in (-pi, pi]
	const double two_pi = 6.283185307179586476925286766559;
	double r = std::fmod(a + 3.14159265358979323846, two_pi);
	if (r < 0) r += two_pi;
	return r - 3.14159265358979323846;
}

// dog joint ids: sim/SimDog.h:11-36
enum DogJoint { jRoot, jSpine0, jSpine1, jSpine2, jSpine3, jTorso, jNeck0, jNeck1, jHead, jTail0, jTail1, jTail2, jTail3,
	jShoulder, jElbow, jWrist, jFinger, jHip, jKnee, jAnkle, jToe, jDogMax };
// sim/DogController.h:12-44
enum { spSpineCurve, spShoulder, spElbow, spHip, spKnee, spAnkle, spMax };
enum { mpTransTime, mpCv, mpBackForceX, mpBackForceY, mpFrontForceX, mpFrontForceY, mpMax };
enum { stBackStance, stExtend, stFrontStance, stGather, stMax, stInvalid };

// raptor joint ids: sim/SimRaptor.h:11-33; params: sim/RaptorController.h (5 misc + 4 states x 8)
enum RaptorJoint { rRoot, rSpine0, rSpine1, rSpine2, rSpine3, rHead, rTail0, rTail1, rTail2, rTail3, rTail4,
	rRightHip, rRightKnee, rRightAnkle, rRightToe, rLeftHip, rLeftKnee, rLeftAnkle, rLeftToe, rRaptorMax };
enum { rspRootPitch, rspSpineCurve, rspStanceHip, rspStanceKnee, rspStanceAnkle, rspSwingHip, rspSwingKnee, rspSwingAnkle, rspMax };
enum { rmpTransTime, rmpCv, rmpCd, rmpForceX, rmpForceY, rmpMax };
enum { rstContact, rstDown, rstPassing, rstUp, rstMax, rstInvalid };

struct Action { int id = -1; double params[ORC_MAXP]; };

struct Tuple { double reward; unsigned flags; std::vector<double> s0, a, s1; };

struct Env {
	OrcModel M;
	RBDModel rbd;
	Integrator integ;
	Ground ground;
	EnvRng rng;
	PolicyNet* net = nullptr;
	int L = 0, D = 0, P = 0, nOpt = 0;

	// --- simulated state ("Bullet world") ---
	double q[ORC_MAXD], qd[ORC_MAXD];
	double tau_applied[ORC_MAXD];   // clamped joint torques held during the next world update
	double tau_ctrl[ORC_MAXD];      // controller output before clamping (for tests)
	bool contact[ORC_MAXL];
	Bodies B;
	double time = 0;

	// --- cSimCharSoftFall ---
	double fall_dist_counter = 5, fall_contact_counter = 0.1, sum_fall_contact = 0;
	double prev_check_x = 0, prev_check_y = 0;
	bool fail_fall_dist = false;

	// --- controller (cDogController + cTerrainRLCharController + cBaseControllerMACE) ---
	int state = 0; double phase = 0; bool first_cycle = true; bool is_off_policy = false;
	Action curr;
	double pd_target[ORC_MAXL];
	double prev_cycle_time = 0, curr_cycle_time = 0;
	double prev_com[2] = {0, 0}, prev_dist[2] = {0, 0};
	double prev_stumble = 0, curr_stumble = 0;
	std::vector<int> commands;
	bool exp_actor = false, exp_critic = false;
	bool enable_exp = false; double exp_rate = 0.2, exp_temp = 1, exp_base_rate = 0.2; double exp_noise = 0.2;
	// raptor: stance leg (0 = right = gDefaultStance, 1 = left) and per-joint cPDController active flags
	int stance = 0; bool pd_active[ORC_MAXL];
	bool IsRaptor() const { return M.char_type == 1; }
	std::vector<double> ground_samples, poli_state;
	double sample_origin[2] = {0, 0};
	std::vector<double> nn_out;

	// --- scenario (cScenarioExp / cScenarioPoliEval) ---
	int cycle_count = 0;
	Tuple cur_tuple;
	std::vector<Tuple> tuples;
	long num_resets = 0, num_cycles = 0, num_episodes = 0;
	double avg_dist = 0, pos_start_x = 0;
	std::vector<double> dist_log;

	bool IsOptParam(int i) const { return M.opt_mask[i] != 0; }  // sim/DogController.cpp:75-115 gParamInfo, sim/RaptorController.cpp:71-114 gOptParamsMasks

	void Init(const OrcModel& m, uint64_t terrain_seed, uint64_t rng_seed, uint64_t env_id)
	{
		M = m; rbd.Init(&M); L = rbd.L; D = rbd.D; P = M.n_params;
		nOpt = 0; for (int i = 0; i < P; ++i) if (IsOptParam(i)) ++nOpt;
		exp_noise = M.exp_noise; stance = 0; for (int j = 0; j < ORC_MAXL; ++j) pd_active[j] = true;
		rng.Seed(rng_seed, env_id);
		ground.type = M.terrain_type; ground.world_scale = M.world_scale;
		SetTerrainParamsLerp(M.terrain_blend);
		ground.rand.Seed(static_cast<unsigned long>(terrain_seed));
		ground_samples.assign(200, 0.0);
		poli_state.assign(PoliStateSize(), 0.0);
		enable_exp = M.enable_explore != 0; exp_rate = M.exp_rate; exp_temp = M.exp_temp; exp_base_rate = M.exp_base_rate;
		// cScenarioSimChar::Init: BuildGround (InitSegments around mid = -1), BuildCharacter
		ground.InitSegments(-10 + -1.0, 10 + -1.0);  // scenarios/ScenarioSimChar.cpp:344-369, gGroundSpawnOffset = -1
		for (int j = 0; j < L; ++j) pd_target[j] = M.target_theta[j];
		for (int i = 0; i < D; ++i) tau_applied[i] = tau_ctrl[i] = 0;
		ResetCharacter();
		InitCharacterPos();   // BuildCharacter: InitCharacterPos precedes BuildController (scenarios/ScenarioSimChar.cpp:320-330)
		CalcCOM(prev_com);    // cDogController::Init: mPrevCOM = character->CalcCOM()
		if (M.scenario == 1) { cycle_count = 0; CommandRandAction(); }
		if (M.scenario == 2) { pos_start_x = q[0]; }
	}

	// scenarios/ScenarioSimChar.cpp:255-272
	void SetTerrainParamsLerp(double lerp)
	{
		int n = M.n_terrain_sets;
		if (n <= 0) {  // cTerrainGen2D::GetDefaultParams is supplied by model.py as set 0 when the file has none
			return;
		}
		lerp = std::min(std::max(lerp, 0.0), n - 1.0);
		int i0 = static_cast<int>(lerp);
		int i1 = std::min(i0 + 1, n - 1);
		lerp -= i0;
		for (int k = 0; k < 40; ++k) ground.params[k] = (1 - lerp) * M.terrain_params[i0][k] + lerp * M.terrain_params[i1][k];
	}

	int PoliStateSize() const { return 200 + (L * 2 - 1) + L * 2; }  // sim/TerrainRLCharController.cpp:308-342
	// sim/BaseControllerMACE.cpp:28-31; sim/BaseControllerCacla.cpp:13-16; sim/BaseControllerQ.cpp:12-15 ("dog" / "raptor" ARE the Q controllers,
	// scenarios/ScenarioSimChar.cpp:421-430, 469-478)
	int PoliActionSize() const { return M.ctrl_type == 2 ? nOpt : (M.ctrl_type == 0 ? M.n_actions : 1 + nOpt); }

	// ---- character reset: cSimCharacter::Reset + cScenarioSimChar::InitCharacterPos -----------------
	void ResetCharacter()
	{
		integ.ResetWarmStart();
		for (int i = 0; i < D; ++i) { q[i] = M.pose0[i]; qd[i] = M.vel0[i]; }
		ForwardKin(M, q, qd, B);
		// controller Reset (cTerrainRLCharController::Reset -> ApplyAction(default); cDogController::Reset)
		ResetController();
		for (int i = 0; i < D; ++i) tau_applied[i] = 0;
		// cSimCharSoftFall::Reset
		fall_dist_counter = 5; prev_check_x = q[0]; prev_check_y = q[1]; fail_fall_dist = false;
		fall_contact_counter = 0.1; sum_fall_contact = 0;
		for (int j = 0; j < L; ++j) contact[j] = false;  // cWorld::Reset -> cContactManager::Reset
	}
	void InitCharacterPos()  // scenarios/ScenarioSimChar.cpp:539-552
	{
		if (M.valid_init_pos_x) q[0] = M.init_pos_x;
		q[1] += ground.SampleHeight(q[0]);
		ForwardKin(M, q, qd, B);
	}
	void ResetController()
	{
		exp_actor = exp_critic = false;                       // cBaseControllerMACE::Reset
		Action a; BuildBaseAction(M.default_action, a);        // cTerrainRLCharController::Reset -> ApplyAction(default)
		ApplyAction(a);
		state = 0; phase = 0;                                 // cCharController::Reset
		phase = 0; first_cycle = true; is_off_policy = false; // ResetParams
		sample_origin[0] = sample_origin[1] = 0;
		prev_cycle_time = 0; prev_dist[0] = prev_dist[1] = 0; curr_cycle_time = 0; prev_stumble = curr_stumble = 0;
		std::fill(ground_samples.begin(), ground_samples.end(), 0.0);
		commands.clear();
		if (IsRaptor()) { for (int j = 0; j < L; ++j) pd_active[j] = true; stance = 0; SetStance(0); }  // ResetParams + mImpPDCtrl.Reset + SetStance(default)
		CalcCOM(prev_com);                                    // cDogController::Reset: mPrevCOM = CalcCOM()
	}
	// scenario-level Reset: scenarios/ScenarioSimChar.cpp:121-132 (+ ScenarioExp.cpp:63-73 / ScenarioPoliEval.cpp:72-78)
	void Reset()
	{
		time = 0;
		pert.link = -1; pert.on = false;   // cWorld::Reset clears the perturbation manager
		ResetCharacter();
		ground.Clear();
		ground.Update(-10 + -1.0, 10 + -1.0);
		InitCharacterPos();
		++num_resets;
		if (M.scenario == 1) { cycle_count = 0; CommandRandAction(); }
		if (M.scenario == 2) { pos_start_x = q[0]; }
	}
	void CommandRandAction() { commands.push_back(rng.RandInt(0, M.n_actions)); }

	// ---- helpers -----------------------------------------------------------------------------------
	void CalcCOM(double* out) const
	{
		double sx = 0, sy = 0, m = 0;
		for (int j = 0; j < L; ++j) { sx += M.body_mass[j] * B.cx[j]; sy += M.body_mass[j] * B.cy[j]; m += M.body_mass[j]; }
		out[0] = sx / m; out[1] = sy / m;
	}
	void CalcCOMVel(double* out) const
	{
		double sx = 0, sy = 0, m = 0;
		for (int j = 0; j < L; ++j) { sx += M.body_mass[j] * B.vcx[j]; sy += M.body_mass[j] * B.vcy[j]; m += M.body_mass[j]; }
		out[0] = sx / m; out[1] = sy / m;
	}
	void BlendCtrlParams(int a, double* out) const
	{
		const double* p0 = M.ctrl_params[M.act_idx0[a]]; const double* p1 = M.ctrl_params[M.act_idx1[a]];
		double b = M.act_blend[a];
		for (int i = 0; i < P; ++i) out[i] = (1 - b) * p0[i] + b * p1[i];
	}
	void PostProcessParams(double* p) const
	{
		p[0] = std::fabs(p[0]); p[1] = std::fabs(p[1]);                       // TransTime, Cv (both characters)
		if (IsRaptor()) p[rmpCd] = std::fabs(p[rmpCd]);                         // sim/RaptorController.cpp:1401-1406
	}
	int NumFrags() const { return (net && net->valid) ? net->d.n_frags : 0; }
	// sim/DogControllerMACE.cpp:44-91
	int AssignFragID(int a_id)
	{
		int frag_id = 0, num_frags = NumFrags();
		if (num_frags > 0) {
			int id0 = M.act_idx0[a_id], id1 = M.act_idx1[a_id];
			if (id0 >= num_frags && id1 >= num_frags) frag_id = rng.RandInt(0, num_frags);
			else if (id0 >= num_frags) frag_id = id1;
			else if (id1 >= num_frags) frag_id = id0;
			else {
				frag_id = rng.FlipCoin() ? id0 : id1;
				int num_ctrl_params = M.n_sets;
				int num_copies = num_frags / num_ctrl_params;
				if (frag_id < num_frags % num_ctrl_params) ++num_copies;
				int offset = rng.RandInt(0, num_copies);
				frag_id += offset * num_ctrl_params;
			}
		}
		return frag_id;
	}
	void BuildBaseAction(int a_id, Action& out)
	{
		out.id = a_id;
		BlendCtrlParams(a_id, out.params);
		if (M.ctrl_type == 1) out.id = AssignFragID(a_id);
	}
	int StanceJ(int k) const { return (stance == 0 ? rRightHip : rLeftHip) + k; }   // k: 0 hip, 1 knee, 2 ankle, 3 toe
	int SwingJ(int k) const { return (stance == 0 ? rLeftHip : rRightHip) + k; }
	void SetStateParams()  // sim/DogController.cpp:1042-1054, sim/RaptorController.cpp:1108-1126 (ENABLE_SPINE_CURVE is not defined)
	{
		if (IsRaptor()) {
			const double* sp = curr.params + rmpMax + state * rspMax;
			pd_target[StanceJ(0)] = sp[rspStanceHip]; pd_target[StanceJ(1)] = sp[rspStanceKnee]; pd_target[StanceJ(2)] = sp[rspStanceAnkle];
			pd_target[SwingJ(0)] = sp[rspSwingHip]; pd_target[SwingJ(1)] = sp[rspSwingKnee]; pd_target[SwingJ(2)] = sp[rspSwingAnkle];
			return;
		}
		const double* sp = curr.params + mpMax + state * spMax;
		const int spine[5] = {jSpine0, jSpine1, jSpine2, jSpine3, jTorso};
		for (int i = 0; i < 5; ++i) pd_target[spine[i]] = sp[spSpineCurve];
		pd_target[jShoulder] = sp[spShoulder]; pd_target[jElbow] = sp[spElbow];
		pd_target[jHip] = sp[spHip]; pd_target[jKnee] = sp[spKnee]; pd_target[jAnkle] = sp[spAnkle];
	}
	void TransitionState(int s) { state = s; phase = 0; SetStateParams(); }
	// cRaptorController::SetStance, sim/RaptorController.cpp:1439-1446
	void SetStance(int st) { stance = st; pd_active[StanceJ(0)] = false; pd_active[SwingJ(0)] = true; SetStateParams(); }
	bool IsActiveVFEffector(int j) const { return j == StanceJ(3) && (state == rstContact || state == rstDown) && contact[j]; }  // :1157-1163
	void NewCycleUpdateCtrl()  // sim/DogController.cpp:1323-1333
	{
		prev_cycle_time = curr_cycle_time; curr_cycle_time = 0;
		prev_stumble = curr_stumble; curr_stumble = 0;
		double com[2]; CalcCOM(com);
		prev_dist[0] = com[0] - prev_com[0]; prev_dist[1] = com[1] - prev_com[1];
		prev_com[0] = com[0]; prev_com[1] = com[1];
	}
	void ApplyAction(const Action& a)  // sim/TerrainRLCharController.cpp:133-146 + DogController.cpp:1318-1322
	{
		curr = a;
		PostProcessParams(curr.params);
		NewCycleUpdateCtrl();
		TransitionState(stBackStance);
	}
	bool IsNewCycle() const { return state == 0 && phase == 0; }  // sim/CharController.cpp:72-75

	// sim/SimDog.cpp:83-105
	bool HasStumbled() const
	{
		if (IsRaptor()) { for (int j = 0; j < L; ++j) if (j != rRightToe && j != rLeftToe && j != rRightAnkle && j != rLeftAnkle && contact[j]) return true; return false; }
		for (int j = 0; j < L; ++j) if (j != jToe && j != jFinger && j != jAnkle && j != jWrist && contact[j]) return true;
		return false;
	}
	bool CheckFallContact() const { int last = IsRaptor() ? rHead : jHead; for (int j = 0; j <= last; ++j) if (contact[j]) return true; return false; }  // sim/SimDog.cpp:112-141
	bool HasFallen() const  // sim/SimCharSoftFall.cpp:53-61, sim/SimDog.cpp:143-161
	{
		bool fall_contact = sum_fall_contact > 0.25;
		bool flipped = std::fabs(WrapPi(q[2])) > 3.14159265358979323846 * 0.8;
		return fall_contact || fail_fall_dist || flipped;
	}

	// ---- policy state --------------------------------------------------------------------------------
	void ParseGround()  // sim/TerrainRLCharController.cpp:168-213
	{
		sample_origin[0] = q[0];
		sample_origin[1] = ground.SampleHeight(q[0]);
		for (int s = 0; s < 200; ++s) {
			double dist = ((10.0 - (-0.5)) * s) / (200 - 1) + (-0.5);
			double x = dist + sample_origin[0];
			ground_samples[s] = ground.SampleHeight(x) - sample_origin[1];
		}
	}
	void BuildPoliState()  // sim/TerrainRLCharController.cpp:215-285
	{
		double ground_h = ground.SampleHeight(q[0]);
		int idx = 0;
		for (int s = 0; s < 200; ++s) poli_state[idx++] = ground_samples[s];
		poli_state[idx++] = q[1] - ground_h;
		for (int j = 1; j < L; ++j) { poli_state[idx++] = B.cx[j] - q[0]; poli_state[idx++] = B.cy[j] - q[1]; }
		for (int j = 0; j < L; ++j) { poli_state[idx++] = B.vcx[j]; poli_state[idx++] = B.vcy[j]; }
		if (IsRaptor() && stance != 0) {
			// FlipPoliPoseStance on the pose block and on the vel block: swap the trailing right-leg / left-leg blocks
			const int nleg = 4 * 2, pose_end = 200 + (2 * L - 1), vel_end = pose_end + 2 * L;
			for (int i = 0; i < nleg; ++i) { std::swap(poli_state[pose_end - 1 - i], poli_state[pose_end - nleg - 1 - i]); std::swap(poli_state[vel_end - 1 - i], poli_state[vel_end - nleg - 1 - i]); }
		}
	}
	void GetOptParams(const double* p, double* out) const { int k = 0; for (int i = 0; i < P; ++i) if (IsOptParam(i)) out[k++] = p[i]; }
	void SetOptParams(const double* opt, double* p) const { int k = 0; for (int i = 0; i < P; ++i) if (IsOptParam(i)) p[i] = opt[k++]; PostProcessParams(p); }

	// ---- MACE action selection: sim/BaseControllerMACE.cpp:254-318 -------------------------------------
	void BuildActorAction(const double* y, int a_id, Action& out) const
	{
		out.id = a_id;
		for (int i = 0; i < P; ++i) out.params[i] = curr.params[i];
		SetOptParams(y + net->d.n_frags + a_id * net->d.frag_size, out.params);
	}
	void DecideActionBoltzmann(Action& out)
	{
		is_off_policy = false;
		double base_rand = rng.RandDouble();
		if (enable_exp && base_rand < exp_base_rate) {
			int a = rng.RandInt(0, M.n_actions);  // BuildRandBaseAction
			BuildBaseAction(a, out);
			is_off_policy = true; exp_actor = true; exp_critic = true;
			return;
		}
		nn_out.assign(net->OutSize(), 0.0);
		net->Eval(poli_state.data(), nn_out.data());
		const double* y = nn_out.data();
		int nf = net->d.n_frags;
		int a_max = 0; for (int i = 1; i < nf; ++i) if (y[i] > y[a_max]) a_max = i;
		int a = a_max;
		if (enable_exp && exp_temp != 0) {
			double vb[16]; double max_val = y[a_max], sum = 0;
			for (int i = 0; i < nf; ++i) { vb[i] = std::exp((y[i] - max_val) / exp_temp); sum += vb[i]; }
			double r = rng.RandDouble(0, sum);
			for (int i = 0; i < nf; ++i) { r -= vb[i]; if (r <= 0) { a = i; break; } }
		}
		BuildActorAction(y, a, out);
		if (enable_exp) {
			double rand_noise = rng.RandDouble();
			if (rand_noise < exp_rate) {
				// ApplyExpNoiseAction, :437-518 (non-covariance branch); noise scale = 1 / OutputScale of frag 0
				int k = 0;
				for (int i = 0; i < P; ++i) if (IsOptParam(i)) {
					double noise = rng.RandNorm(0, exp_noise);
					double scale = 1.0 / net->out_scale[nf + k];
					out.params[i] += noise * scale; ++k;
				}
				exp_actor = true;
			}
			exp_critic = (a != a_max);
			is_off_policy = exp_actor || exp_critic;
		}
	}
	// ---- Q action selection: sim/BaseControllerQ.cpp:32-87 (ShouldExplore / DecideAction / ExploitPolicy / ExploreAction). The net (trunk -> ip1 -> ip2
	// -> one output per base action, data/policies/dog/nets/dog_q_deploy.prototxt) is held like the CACLA actor: one fragment of n_actions outputs
	// behind an unused critic slot. Eigen's maxCoeff(&a) returns the FIRST maximum
	void DecideActionQ(Action& out)
	{
		bool explore = false;
		if (enable_exp) explore = rng.RandDouble() < exp_rate;
		is_off_policy = explore;
		if (explore) { BuildBaseAction(rng.RandInt(0, M.n_actions), out); return; }   // BuildRandBaseAction
		nn_out.assign(net->OutSize(), 0.0);
		net->Eval(poli_state.data(), nn_out.data());
		const double* y = nn_out.data() + 1;
		int a = 0; for (int i = 1; i < M.n_actions; ++i) if (y[i] > y[a]) a = i;
		BuildBaseAction(a, out);
	}
	// ---- CACLA action selection: sim/BaseControllerCacla.cpp:124-151 (ShouldExplore / DecideAction), 205-234 (ExploitPolicy / ExploreAction),
	// 262-296 (ApplyExpNoiseAction). The actor is held as a one-fragment net of the MACE family with an all-zero critic head, so its
	// parameter outputs sit behind one (unused) critic slot
	void DecideActionCacla(Action& out)
	{
		bool explore = false;
		if (enable_exp) explore = rng.RandDouble() < exp_rate;
		is_off_policy = explore;
		if (explore && rng.RandDouble() < exp_base_rate) {
			int a = rng.RandInt(0, M.n_actions);  // BuildRandBaseAction
			BuildBaseAction(a, out);
			return;
		}
		nn_out.assign(net->OutSize(), 0.0);
		net->Eval(poli_state.data(), nn_out.data());
		BuildActorAction(nn_out.data(), 0, out);
		out.id = -1;   // gInvalidIdx
		if (explore) {
			int k = 0;
			for (int i = 0; i < P; ++i) if (IsOptParam(i)) {
				double noise = rng.RandNorm(0, exp_noise);
				out.params[i] += noise * (1.0 / net->out_scale[1 + k]); ++k;
			}
		}
	}
	void UpdateAction()  // sim/DogControllerMACE.cpp:26-30 + sim/DogController.cpp:847-868
	{
		if (M.ctrl_type == 1) { exp_actor = false; exp_critic = false; }
		ParseGround();
		BuildPoliState();
		is_off_policy = true;
		Action a = curr;
		if (!commands.empty()) {
			int cmd = commands.back(); commands.pop_back();
			if (M.ctrl_type == 1) { exp_actor = true; exp_critic = true; }
			BuildBaseAction(cmd, a);
		} else if (net && net->valid) {
			if (M.ctrl_type == 2) DecideActionCacla(a); else if (M.ctrl_type == 0) DecideActionQ(a); else DecideActionBoltzmann(a);
		} else {
			bool cyclic = (M.ctrl_type >= 1) ? false : (M.act_cyclic[curr.id] != 0);  // MACE / CACLA: IsCurrActionCyclic() == false
			if (!cyclic) BuildBaseAction(M.default_action, a);
		}
		ApplyAction(a);
	}

	// ---- PD error terms: sim/PDController.cpp:181-224 ---------------------------------------------------
	// relative joint: theta = -btHingeConstraint::getHingeAngle() - ref_theta (sim/World.cpp:543-553), the hinge angle an atan2: the value the controller
	// reads lives in [-pi - ref_theta, pi - ref_theta). World-coordinate joint (dog shoulder / hip): cJoint::GetChildRotation -> cWorld::GetRotation
	// (sim/World.cpp:374-384) = btQuaternion::getAngle() * (axis . z) of the quaternion btMatrix3x3::getRotation extracts from the child's world basis:
	// for a rotation phi about z that is phi itself while trace = 1 + 2 cos(phi) > 0, and otherwise (largest-diagonal branch, z kept positive,
	// w = sin(phi) * s) phi for phi in [2 pi / 3, pi] but phi + 2 pi for phi in (-pi, -2 pi / 3]: 2 acos(w) with w < 0. The PD error of a shoulder or hip
	// whose link points more than 120 degrees clockwise is therefore off by a full turn in the reference, and here.
	double CalcTheta(int j) const
	{
		if (!M.use_world[j]) return WrapPi(q[j + 2] + M.ref_theta[j]) - M.ref_theta[j];
		const double phi = WrapPi(B.psi[j]);
		return phi <= -2.0943951023931954923084289221863 ? phi + 6.283185307179586476925286766559 : phi;
	}

	// ---- cDogController::Update, sim/DogController.cpp:229-268 -------------------------------------------
	void ImpPD(double dt, double* tau)  // cImpPDController::CalcControlForces, sim/ImpPDController.cpp:234-278
	{
		double Mm[ORC_MAXD * ORC_MAXD], rhs[ORC_MAXD], acc[ORC_MAXD], kp[ORC_MAXD], kd[ORC_MAXD], kdm[ORC_MAXD], perr[ORC_MAXD], verr[ORC_MAXD];
		for (int i = 0; i < D; ++i) { kp[i] = kd[i] = kdm[i] = perr[i] = verr[i] = 0; }
		for (int j = 1; j < L; ++j) {  // root has no cJoint -> its cPDController is never Init'ed (invalid)
			kdm[j + 2] = M.kd[j];        // M.diagonal() += t * mKd uses the raw gains even for inactive controllers (:258)
			perr[j + 2] = pd_target[j] - CalcTheta(j);
			verr[j + 2] = 0 - qd[j + 2];
			if (pd_active[j]) { kp[j + 2] = M.kp[j]; kd[j + 2] = M.kd[j]; }   // inactive -> Kp_mat / Kd_mat rows zeroed (:244-255)
		}
		for (int i = 0; i < D; ++i) for (int k = 0; k < D; ++k) Mm[i * D + k] = rbd.H[i][k];
		for (int i = 0; i < D; ++i) Mm[i * D + i] += dt * kdm[i];
		for (int i = 0; i < D; ++i) rhs[i] = kp[i] * (perr[i] - dt * qd[i]) + kd[i] * verr[i] - rbd.C[i];
		SolveLDLT(D, Mm, D, rhs, acc);
		for (int i = 0; i < D; ++i) tau[i] += kp[i] * (perr[i] - dt * qd[i]) + kd[i] * (verr[i] - dt * acc[i]);
	}
	void ClampAndApply(const double* tau)  // cSimCharacter::ApplyControlForces + cJoint::ApplyTorque, sim/Joint.cpp:171-201,257-264
	{
		for (int i = 0; i < D; ++i) { tau_ctrl[i] = tau[i]; tau_applied[i] = 0; }
		for (int j = 1; j < L; ++j) {
			double t = tau[j + 2], lim = M.torque_lim[j];
			if (std::fabs(t) > lim) t *= lim / std::fabs(t);
			tau_applied[j + 2] = t;
		}
	}
	// cRaptorController::Update, sim/RaptorController.cpp:195-233
	void RaptorControllerUpdate(double dt)
	{
		double tau[ORC_MAXD];
		for (int i = 0; i < D; ++i) tau[i] = 0;
		curr_cycle_time += dt;
		if (HasStumbled()) curr_stumble += dt;
		rbd.Update(q, qd, /*fix_cj=*/false);
		{   // UpdateState :804-849
			bool advance = first_cycle;
			phase += dt / curr.params[rmpTransTime];
			if (state != rstUp && phase >= 1) advance = true;
			if (state == rstUp && contact[SwingJ(3)]) advance = true;
			if (advance) {
				int next = first_cycle ? rstContact : ((state == rstUp) ? rstInvalid : state + 1);
				bool end_step = (next == rstInvalid) || first_cycle;
				if (end_step) { if (!first_cycle) SetStance(stance == 0 ? 1 : 0); UpdateAction(); first_cycle = false; }
				else TransitionState(next);
			}
		}
		pd_active[StanceJ(0)] = !IsActiveVFEffector(StanceJ(3));   // UpdateStanceHip :899-905
		{   // ApplySwingFeedback :907-931
			double cv = curr.params[rmpCv], cd = curr.params[rmpCd];
			bool first_half = state == rstContact || state == rstDown;
			cd = first_half ? 0 : cd; cv = first_half ? cv : 0;
			double com[2], com_vel[2]; CalcCOM(com); CalcCOMVel(com_vel);
			double d_theta = cd * (com[0] - B.cx[StanceJ(3)]) + cv * com_vel[0];
			pd_target[SwingJ(0)] = curr.params[rmpMax + state * rspMax + rspSwingHip] + d_theta;
		}
		ImpPD(dt, tau);
		if (M.enable_grav_comp) {   // :985-1028 (weighted ridge LS, support = active stance effector only)
			double basis[ORC_MAXD][4];
			for (int i = 0; i < D; ++i) for (int k = 0; k < 4; ++k) basis[i][k] = 0;
			bool has_support = false;
			const int effs[2] = {rRightToe, rLeftToe};
			for (int e = 0; e < 2; ++e) {
				int jid = effs[e];
				if (!IsActiveVFEffector(jid)) continue;
				has_support = true;
				double pos[2]; EffectorContactPos(jid, pos);
				const double fb[2][2] = {{0, 1}, {1, 0}};
				for (int c = jid; c >= 0; c = M.parent[c]) for (int a = 0; a < rbd.kt.dim[c]; ++a) for (int b = 0; b < 2; ++b) basis[rbd.kt.off[c] + a][e * 2 + b] = JtF(rbd.kt.off[c] + a, pos, fb[b]);
			}
			if (has_support) {
				double tau_g[ORC_MAXD]; rbd.CalcGravityForce(tau_g);
				for (int i = 0; i < D; ++i) tau_g[i] = -tau_g[i];
				const double W[3] = {0.0001, 0.0001, 1};
				double AtA[16], Atb[4], x[4];
				for (int a = 0; a < 4; ++a) {
					for (int b = 0; b < 4; ++b) { double s = 0; for (int r = 0; r < 3; ++r) s += basis[r][a] * W[r] * basis[r][b]; AtA[a * 4 + b] = s; }
					double s = 0; for (int r = 0; r < 3; ++r) s += basis[r][a] * W[r] * tau_g[r];
					Atb[a] = s; AtA[a * 4 + a] += 0.0001;
				}
				SolveGE(4, AtA, Atb, x);
				for (int i = 0; i < D; ++i) { double s = 0; for (int k = 0; k < 4; ++k) s += basis[i][k] * x[k]; tau[i] += tau_g[i] - s; }
			}
		}
		if (IsActiveVFEffector(StanceJ(3))) {   // ApplyStanceFeedback :933-983
			double hip_tau = -tau[SwingJ(0) + 2];
			const int sh = StanceJ(0);
			double root_tau = M.kp[sh] * (curr.params[rmpMax + state * rspMax + rspRootPitch] - WrapPi(q[2])) + M.kd[sh] * (-qd[2]);
			hip_tau += -root_tau;
			tau[sh + 2] += hip_tau;
		}
		if (M.enable_virtual_forces) {   // :1030-1075
			const int effs[2] = {rRightToe, rLeftToe};
			for (int e = 0; e < 2; ++e) {
				int jid = effs[e];
				if (!IsActiveVFEffector(jid)) continue;
				double f[2] = {-curr.params[rmpForceX], -curr.params[rmpForceY]};
				double pos[2]; EffectorContactPos(jid, pos);
				for (int c = jid; c != rRoot; c = M.parent[c]) {
					double t = JtF(c + 2, pos, f);
					tau[c + 2] += t;
					if (c == StanceJ(0)) tau[SwingJ(0) + 2] += -t;
				}
			}
		}
		ClampAndApply(tau);
	}
	void ControllerUpdate(double dt)
	{
		if (IsRaptor()) { RaptorControllerUpdate(dt); return; }
		double tau[ORC_MAXD];
		for (int i = 0; i < D; ++i) tau[i] = 0;
		curr_cycle_time += dt;
		if (HasStumbled()) curr_stumble += dt;
		rbd.Update(q, qd, /*fix_cj=*/false);  // UpdateRBDModel (reference quirk kept)
		// UpdateState :805-845
		{
			bool advance = first_cycle;
			double trans_time = curr.params[mpTransTime];
			phase += dt / trans_time;
			bool trans_time_state = (state == stBackStance || state == stFrontStance);
			if (trans_time_state && phase >= 1) advance = true;
			int trans_contact = (state == stExtend) ? jFinger : ((state == stGather) ? jToe : -1);
			if (trans_contact >= 0 && contact[trans_contact]) advance = true;
			if (advance) {
				int next = first_cycle ? stBackStance : ((state == stGather) ? stInvalid : state + 1);
				bool end_step = (next == stInvalid) || first_cycle;
				if (end_step) { UpdateAction(); first_cycle = false; }
				else TransitionState(next);
			}
		}
		// ApplyFeedback :903-945
		{
			double com_vel[2]; CalcCOMVel(com_vel);
			const int joints[2] = {jHip, jShoulder}, effs[2] = {jToe, jFinger}, prm[2] = {spHip, spShoulder};
			for (int k = 0; k < 2; ++k) if (!contact[effs[k]]) {
				double default_theta = curr.params[mpMax + state * spMax + prm[k]];
				pd_target[joints[k]] = default_theta + com_vel[0] * curr.params[mpCv];
			}
		}
		ImpPD(dt, tau);   // UpdatePDCtrls
		if (M.enable_grav_comp) ApplyGravityCompensation(tau);
		if (M.enable_virtual_forces) ApplyVirtualForces(tau);
		ClampAndApply(tau);
	}
	void EffectorContactPos(int j, double* out) const  // sim/DogController.cpp:1372-1387
	{
		double c = std::cos(B.psi[j]), s = std::sin(B.psi[j]);
		double ly = -0.5 * M.body_size[j][1];
		out[0] = B.cx[j] - s * ly; out[1] = B.cy[j] + c * ly;
	}
	// tau_j += J_j^T * ApplyTransF(BuildTrans(-pos), (0; f)) for a world-frame force f applied at pos
	double JtF(int dof, const double* pos, const double* f) const
	{
		SV sp = ApplyTransF(BuildTrans(V3{-pos[0], -pos[1], 0}), SV{{0, 0, 0}, {f[0], f[1], 0}});
		return dot(rbd.J[dof], sp);
	}
	void ApplyGravityCompensation(double* tau)  // sim/DogController.cpp:947-995 + :1120-1175
	{
		const double lambda = 0.0001;
		const int effs[2] = {jToe, jFinger};
		double basis[ORC_MAXD][4];
		for (int i = 0; i < D; ++i) for (int k = 0; k < 4; ++k) basis[i][k] = 0;
		bool has_support = false;
		for (int e = 0; e < 2; ++e) {
			int jid = effs[e];
			if (!contact[jid]) continue;
			has_support = true;
			double pos[2]; EffectorContactPos(jid, pos);
			const double fb[2][2] = {{0, 1}, {1, 0}};  // force_svs columns: (0,0,0,0,1,0) then (0,0,0,1,0,0)
			int c = jid;
			while (c >= 0) {
				for (int a = 0; a < rbd.kt.dim[c]; ++a) for (int b = 0; b < 2; ++b) basis[rbd.kt.off[c] + a][e * 2 + b] = JtF(rbd.kt.off[c] + a, pos, fb[b]);
				c = M.parent[c];
			}
		}
		if (!has_support) return;
		double tau_g[ORC_MAXD];
		rbd.CalcGravityForce(tau_g);
		for (int i = 0; i < D; ++i) tau_g[i] = -tau_g[i];
		double AtA[16], Atb[4], x[4];
		for (int a = 0; a < 4; ++a) {
			for (int b = 0; b < 4; ++b) { double s = 0; for (int r = 0; r < 3; ++r) s += basis[r][a] * basis[r][b]; AtA[a * 4 + b] = s; }
			double s = 0; for (int r = 0; r < 3; ++r) s += basis[r][a] * tau_g[r];
			Atb[a] = s; AtA[a * 4 + a] += lambda;
		}
		SolveGE(4, AtA, Atb, x);
		for (int i = 0; i < D; ++i) { double s = 0; for (int k = 0; k < 4; ++k) s += basis[i][k] * x[k]; tau_g[i] -= s; }
		tau_g[0] = tau_g[1] = tau_g[2] = 0;
		for (int i = 0; i < D; ++i) tau[i] += tau_g[i];
	}
	void ApplyVirtualForces(double* tau)  // sim/DogController.cpp:997-1029, :1091-1118
	{
		const int effs[2] = {jToe, jFinger};
		for (int e = 0; e < 2; ++e) {
			int jid = effs[e];
			bool valid = ((state == stBackStance || state == stExtend) && jid == jToe) || ((state == stFrontStance || state == stGather) && jid == jFinger);
			if (!(valid && contact[jid])) continue;
			double f[2];
			if (jid == jToe) { f[0] = -curr.params[mpBackForceX]; f[1] = -curr.params[mpBackForceY]; }
			else { f[0] = -curr.params[mpFrontForceX]; f[1] = -curr.params[mpFrontForceY]; }
			double pos[2]; EffectorContactPos(jid, pos);
			int c = jid;
			while (c != jRoot && c != jTorso) { tau[c + 2] += JtF(c + 2, pos, f); c = M.parent[c]; }
		}
	}

	// ---- reward: sim/DogController.cpp:594-628 -------------------------------------------------------------
	double CalcReward() const
	{
		double vel_reward = 0, stumble_reward = 0;
		if (!HasFallen()) {
			double cycle_time = prev_cycle_time;
			double avg_vel = prev_dist[0] / cycle_time;
			double vel_err = M.target_vel_x - avg_vel;
			vel_reward = std::exp(-0.5 * vel_err * vel_err);
			double avg_stumble = prev_stumble / cycle_time;
			stumble_reward = 1.0 / (1 + 10 * avg_stumble);
			if (IsRaptor() && avg_vel < 0) { vel_reward = 0; stumble_reward = 0; }   // sim/RaptorController.cpp:584-588
		}
		return 0.8 * vel_reward + 0.2 * stumble_reward;
	}

	// ---- scenario NewCycleUpdate: scenarios/ScenarioExp.cpp:209-243, ScenarioExpMACE.cpp:16-28 ----------------
	void ScenarioNewCycleUpdate()
	{
		++num_cycles;
		if (M.scenario != 1) return;
		cur_tuple.s1 = poli_state;
		bool fail = HasFallen();
		cur_tuple.flags = (cur_tuple.flags & ~1u) | (fail ? 1u : 0u);
		cur_tuple.reward = CalcReward();
		if (cycle_count > 1) tuples.push_back(cur_tuple);  // gNumWarmupCycles = 1
		cur_tuple.s0 = cur_tuple.s1;
		cur_tuple.a.assign(PoliActionSize(), 0.0);
		if (M.ctrl_type == 2) GetOptParams(curr.params, cur_tuple.a.data());   // cBaseControllerCacla::RecordPoliAction
		else if (M.ctrl_type == 0) { if (curr.id >= 0 && curr.id < M.n_actions) cur_tuple.a[curr.id] = 1; }   // cBaseControllerQ::RecordPoliAction: one-hot
		else { cur_tuple.a[0] = curr.id; GetOptParams(curr.params, cur_tuple.a.data() + 1); }
		cur_tuple.flags = 0;
		if (M.ctrl_type == 1) cur_tuple.flags |= (exp_critic ? 2u : 0u) | (exp_actor ? 4u : 0u);
		if (M.ctrl_type == 2) cur_tuple.flags |= is_off_policy ? 2u : 0u;       // cScenarioExpCacla::RecordFlagsBeg (cCaclaTrainer::eFlagOffPolicy)
		++cycle_count;
	}

	// ---- one iteration of the loop at scenarios/ScenarioSimChar.cpp:162-173 ---------------------------------
	// cWorld::AddPerturb with one slot per env; local position in the body frame
	Integrator::PerturbForce pert; double pert_lp[2] = {0, 0}, pert_time = 0, pert_dur = 0;
	void AddPerturb(int link, double lx, double ly, double fx, double fy, double dur)
	{
		pert.link = link; pert.fx = fx; pert.fy = fy; pert.torque = 0; pert.on = false;
		pert_lp[0] = lx; pert_lp[1] = ly; pert_time = 0; pert_dur = dur;
	}
	void EnvStep(double dt)
	{
		double h = dt / M.num_sim_substeps;
		if (pert.link >= 0) {   // cPerturbManager::UpdatePerturbs at the start of cWorld::Update (sim/PerturbManager.cpp:41-56)
			if (pert_time >= pert_dur) { pert.link = -1; pert.on = false; }
			else {
				pert_time += dt; pert.on = true;
				const double c = std::cos(B.psi[pert.link]), sn = std::sin(B.psi[pert.link]);
				const double rx = c * pert_lp[0] - sn * pert_lp[1], ry = sn * pert_lp[0] + c * pert_lp[1];
				pert.torque = rx * pert.fy - ry * pert.fx;
			}
		}
		for (int s = 0; s < M.num_sim_substeps; ++s) { integ.sub_ix = s; integ.Substep(M, rbd, ground, h, q, qd, tau_applied, &pert); }  // UpdateWorld
		ForwardKin(M, q, qd, B);
		{   // cContactManager::Update at the post-step configuration
			ContactPoint tmp[1];
			DetectContacts(M, B, ground, tmp, 0, contact);
		}
		ground.Update(q[0] - 2, q[0] + 10 + 1);  // UpdateGround: scenarios/ScenarioSimChar.cpp:564-572
		ControllerUpdate(dt);                     // UpdateCharacter -> cSimCharacter::Update
		// cSimCharSoftFall::UpdateFallDistCheck / UpdateFallContactCheck, sim/SimCharSoftFall.cpp:74-125
		fall_dist_counter -= dt;
		if (fall_dist_counter <= 0) {
			double dx = q[0] - prev_check_x, dy = q[1] - prev_check_y;
			if (dx * dx + dy * dy < 0.5 * 0.5) fail_fall_dist = true;
			prev_check_x = q[0]; prev_check_y = q[1]; fall_dist_counter = 5;
		}
		fall_contact_counter -= dt;
		if (fall_contact_counter <= 0) {
			const double discount = 0.9, norm = (1 + 1 / (1 - discount));
			double val = CheckFallContact() ? 1 : 0;
			sum_fall_contact = val / norm + discount * sum_fall_contact;
			fall_contact_counter = 0.1;
		}
		if (IsNewCycle()) ScenarioNewCycleUpdate();  // PostSubstepUpdate
		time += dt;
	}
	// cScenarioExp::Update / cScenarioPoliEval::Update (scenarios/ScenarioExp.cpp:83-98, ScenarioPoliEval.cpp:110-125)
	void Update(double time_elapsed)
	{
		if (time_elapsed <= 0) return;
		double step = time_elapsed / M.num_update_steps;
		for (int i = 0; i < M.num_update_steps; ++i) EnvStep(step);
		FrameEnd();
	}
	// the scenario logic behind the step loop of one outer frame (fall -> tuple / distance record -> Reset)
	void FrameEnd()
	{
		if (M.scenario == 1) {
			if (!IsNewCycle() && HasFallen()) { ScenarioNewCycleUpdate(); Reset(); }
		} else if (M.scenario == 2) {
			if (HasFallen()) {
				if (num_cycles >= 1) {   // IsValidCycle(): mCycleCount >= gNumWarmupCycles (= 1), scenarios/ScenarioPoliEval.cpp:7, 406-410
					double dist = q[0] - pos_start_x;
					avg_dist = (num_episodes * avg_dist + dist) / (num_episodes + 1.0);
					++num_episodes; dist_log.push_back(dist);
				}
				Reset();
			}
		}
	}
};

}  // namespace orc
