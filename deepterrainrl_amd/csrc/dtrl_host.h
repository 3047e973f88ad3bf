// dtrl_host.h -- host-side data layer of the engine: reference-format argument files, JSON data files, flattening of the
// character / controller / terrain descriptions into the POD tables the kernels read, and the per-env heightfield windows.
//
// Mirrors the reference's interfaces for this path (names and argument meaning kept):
//   cArgParser                 util/ArgParser.h:6-34, util/ArgParser.cpp:42-334
//   cKinTree::Load / LoadBodyDefs, cPDController::LoadParams, cDogController::LoadControllers / ReadParams
//   cTerrainGen2D (terrain functions + params), cGroundVar2D (two sliding segments)
#pragma once
#include "dtrl_types.h"
#include <map>
#include <memory>
#include <random>
#include <string>
#include <vector>

namespace dtrl {

// ---- minimal JSON value (the data files are plain JSON; jsoncpp is not available in this image) ----
struct Json {
	enum Type { kNull, kBool, kNum, kStr, kArr, kObj } type = kNull;
	double num = 0; bool b = false; std::string str;
	std::vector<Json> arr;
	std::vector<std::pair<std::string, Json>> obj;
	const Json* find(const std::string& key) const;
	bool has(const std::string& key) const { const Json* j = find(key); return j && j->type != kNull; }
	double get_num(const std::string& key, double dflt) const { const Json* j = find(key); return (j && (j->type == kNum)) ? j->num : ((j && j->type == kBool) ? (j->b ? 1.0 : 0.0) : dflt); }
	bool get_bool(const std::string& key, bool dflt) const { const Json* j = find(key); if (!j) return dflt; if (j->type == kBool) return j->b; if (j->type == kNum) return j->num != 0; return dflt; }
	static bool parse(const std::string& text, Json& out, std::string& err);
	static bool parse_file(const std::string& path, Json& out, std::string& err);
};

// ---- cArgParser ----
class ArgParser {
public:
	ArgParser() {}
	ArgParser(const char* const* args, int n) { AppendArgs(args, n); }
	void AppendArgs(const char* const* args, int n);
	bool AppendArgs(const std::string& file);   // util/ArgParser.cpp:42-108
	bool ParseString(const std::string& key, std::string& out) const;
	bool ParseInt(const std::string& key, int& out) const;
	bool ParseDouble(const std::string& key, double& out) const;
	bool ParseBool(const std::string& key, bool& out) const;
	int GetNumArgs() const { return static_cast<int>(mArgs.size()); }
private:
	std::vector<std::string> mArgs;
	static bool IsKey(const std::string& s);
	int FindKeyIndex(const std::string& key) const;
};

// ---- terrain ----
// cRand restated over libstdc++ <random> (util/Rand.cpp): identical streams to the reference on the same libstdc++.
class TerrainRand {
public:
	void Seed(unsigned long s) { gen_.seed(s); }
	double RandDouble(double mn, double mx) { if (mn == mx) return mn; double r = real_(gen_); return mn + (r * (mx - mn)); }
	int RandInt(int mn, int mx) { if (mn == mx) return mn; int r = int_(gen_); return mn + r % (mx - mn); }
	bool FlipCoin() { return RandDouble(0, 1) < 0.5; }
	int RandSign() { return FlipCoin() ? -1 : 1; }
private:
	std::default_random_engine gen_;
	std::uniform_real_distribution<double> real_{0, 1};
	std::uniform_int_distribution<int> int_{0, std::numeric_limits<int>::max()};
};

enum TerrainType { kTerrFlat, kTerrGaps, kTerrSteps, kTerrWalls, kTerrBumps, kTerrMixed, kTerrNarrowGaps, kTerrSlopes, kTerrSlopesGaps,
	kTerrSlopesSteps, kTerrSlopesWalls, kTerrSlopesMixed, kTerrSlopesNarrowGaps, kTerrCliffs, kTerrTypeMax };
constexpr int kNumTerrainParams = 40;
extern const char* const kTerrainTypeNames[kTerrTypeMax];
extern const char* const kTerrainParamNames[kNumTerrainParams];
extern const double kTerrainParamDefaults[kNumTerrainParams];

// appends one terrain strip of (at least) `width` metres to `out`; returns the width added (cTerrainGen2D::tTerrainFunc)
double BuildTerrain(int type, double width, const double* params, TerrainRand& rand, std::vector<float>& out);

// cGroundVar2D: two sliding heightfield segments of one env, host-resident mirror of the device GroundRec
class GroundWindow {
public:
	void Configure(int type, const double* params, double world_scale, double segment_width);
	void SetParams(const double* params);
	void SeedRand(unsigned long seed) { rand_.Seed(seed); }
	void Clear();
	// returns true when a segment was (re)built, i.e. the device record must be refreshed
	bool Update(double bound_min_x, double bound_max_x);   // sim/GroundVar2D.cpp:43-91
	bool NeedsUpdate(double bound_min_x, double bound_max_x) const;   // Update() would (re)build a segment
	void InitSegments(double bound_min_x, double bound_max_x);
	bool FillRecord(GroundRec& rec, std::string& err) const;
	long num_builds() const { return builds_; }
private:
	struct Seg { std::vector<float> data; double min_x = 0, origin_x = 0, scale_x = 0.1; double MaxX() const; double MinX() const; };
	void BuildSegment(int seg_id, double bound_min, double bound_max, bool align_min, double fix_y);
	int SegID(int s) const { return flip_ ? (s == 0 ? 1 : 0) : s; }
	int type_ = 0; double params_[kNumTerrainParams]; double world_scale_ = 1, segment_width_ = 20;
	TerrainRand rand_;
	Seg segs_[2];
	bool flip_ = false;
	long builds_ = 0;
};

// ---- scenario configuration flattened from the arg file + data files ----
struct ScenarioConfig {
	DevModel model{};
	RunParams run{};
	NetDesc net{};
	bool has_policy_net = false;
	// CACLA (sim/BaseControllerCacla.cpp): the policy net is the ACTOR alone (slice / 3 conv / terr_ip0 / ip1 / ip2 / output). It runs on the
	// device as a MACE-family net with one fragment whose critic head is all zeros; the boundary keeps the actor's own sizes and blob order
	bool actor_only = false;
	int64_t user_num_params = 0;   // what dtrl_set_policy expects (== net.num_params unless actor_only)
	int user_out_size = 0;         // normaliser length at the boundary (== net.out_size unless actor_only)
	int terrain_type = 0;
	std::vector<std::vector<double>> terrain_param_sets;
	double terrain_blend = 0;
	uint64_t terrain_seed = 0;
	bool device_terrain = false;   // -terrain_gen= device: windows are generated and slid by the GPU at the frame boundary (dtrl_terrain_dev.h)
	int tuple_buffer_size = 16;
	bool tuple_ring_host = false;  // -tuple_ring= host: the tuple rings live in page-locked host memory the kernels write directly (drains without a device copy)
	int tuple_ring_capacity = 0;   // -tuple_ring_capacity=: rows of the device tuple ring (0 = max(2 num_envs, tuple_buffer_size))
	// cScenarioSimChar::ApplyRandForce ranges (scenarios/ScenarioSimChar.cpp:60-63, 88-91; the duration key's typo is the reference's)
	double min_perturb = 50, max_perturb = 100, min_perturb_duration = 0.1, max_perturb_duration = 0.5;
	std::string data_root;
	std::string policy_net_file, policy_model_file;
};

bool LoadScenario(const ArgParser& args, ScenarioConfig& cfg, std::string& err);
bool ParseDeployPrototxt(const std::string& path, NetDesc& d, std::string& err, bool* actor_only = nullptr);
// cBaseControllerMACE::BuildNNOutputOffsetScale + cDogControllerMACE::BuildActorBias
void BuildOutputOffsetScale(const DevModel& m, const NetDesc& d, std::vector<double>& off, std::vector<double>& scale);
void LerpTerrainParams(const ScenarioConfig& cfg, double lerp, double* out);

}  // namespace dtrl
