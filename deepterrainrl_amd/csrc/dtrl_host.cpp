// dtrl_host.cpp -- see dtrl_host.h. Reference citations are file:line relative to the reference repo root.
#include "dtrl_host.h"
#include "dtrl_terrain_gen.h"
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <regex>
#include <sstream>

namespace dtrl {

// =====================================================================================================================
// JSON
namespace {
struct JsonParser {
	const std::string& s; size_t p = 0; std::string err;
	explicit JsonParser(const std::string& t) : s(t) {}
	void ws() { while (p < s.size() && (s[p] == ' ' || s[p] == '\t' || s[p] == '\n' || s[p] == '\r')) ++p; }
	bool fail(const char* m) { if (err.empty()) { err = std::string(m) + " at offset " + std::to_string(p); } return false; }
	bool value(Json& out)
	{
		ws();
		if (p >= s.size()) return fail("unexpected end");
		char c = s[p];
		if (c == '{') {
			out.type = Json::kObj; ++p; ws();
			if (p < s.size() && s[p] == '}') { ++p; return true; }
			while (true) {
				ws(); Json key;
				if (p >= s.size() || s[p] != '"' || !string(key.str)) return fail("expected key");
				ws(); if (p >= s.size() || s[p] != ':') return fail("expected ':'");
				++p; Json v; if (!value(v)) return false;
				out.obj.emplace_back(key.str, std::move(v));
				ws(); if (p < s.size() && s[p] == ',') { ++p; continue; }
				if (p < s.size() && s[p] == '}') { ++p; return true; }
				return fail("expected ',' or '}'");
			}
		}
		if (c == '[') {
			out.type = Json::kArr; ++p; ws();
			if (p < s.size() && s[p] == ']') { ++p; return true; }
			while (true) {
				Json v; if (!value(v)) return false;
				out.arr.push_back(std::move(v));
				ws(); if (p < s.size() && s[p] == ',') { ++p; continue; }
				if (p < s.size() && s[p] == ']') { ++p; return true; }
				return fail("expected ',' or ']'");
			}
		}
		if (c == '"') { out.type = Json::kStr; return string(out.str); }
		if (s.compare(p, 4, "true") == 0) { out.type = Json::kBool; out.b = true; p += 4; return true; }
		if (s.compare(p, 5, "false") == 0) { out.type = Json::kBool; out.b = false; p += 5; return true; }
		if (s.compare(p, 4, "null") == 0) { out.type = Json::kNull; p += 4; return true; }
		char* end = nullptr;
		double v = std::strtod(s.c_str() + p, &end);
		if (end == s.c_str() + p) return fail("bad value");
		out.type = Json::kNum; out.num = v; p = static_cast<size_t>(end - s.c_str());
		return true;
	}
	bool string(std::string& out)
	{
		++p; out.clear();
		while (p < s.size() && s[p] != '"') {
			if (s[p] == '\\' && p + 1 < s.size()) { char e = s[p + 1]; out += (e == 'n' ? '\n' : e == 't' ? '\t' : e); p += 2; }
			else out += s[p++];
		}
		if (p >= s.size()) return fail("unterminated string");
		++p; return true;
	}
};
}  // namespace

const Json* Json::find(const std::string& key) const
{
	for (const auto& kv : obj) if (kv.first == key) return &kv.second;
	return nullptr;
}
bool Json::parse(const std::string& text, Json& out, std::string& err)
{
	JsonParser jp(text);
	if (!jp.value(out)) { err = jp.err; return false; }
	return true;
}
bool Json::parse_file(const std::string& path, Json& out, std::string& err)
{
	std::ifstream f(path);
	if (!f.is_open()) { err = "cannot open " + path; return false; }
	std::stringstream ss; ss << f.rdbuf();
	if (!parse(ss.str(), out, err)) { err = path + ": " + err; return false; }
	return true;
}

// =====================================================================================================================
// cArgParser (util/ArgParser.cpp)
void ArgParser::AppendArgs(const char* const* args, int n) { for (int i = 0; i < n; ++i) mArgs.emplace_back(args[i]); }

bool ArgParser::AppendArgs(const std::string& file)
{
	// util/ArgParser.cpp:42-108: character scanner; whitespace separates tokens, "//" starts a comment up to end of line,
	// a single '/' followed by another character is kept as part of the token.
	FILE* fp = std::fopen(file.c_str(), "r");
	if (!fp) return false;
	std::string tok; bool comment = false; int slashes = 0; int ch;
	while ((ch = std::fgetc(fp)) != EOF) {
		char c = static_cast<char>(ch);
		if (!comment) {
			if (c != '/' && slashes == 1) { tok += '/'; slashes = 0; }
			if (c == ' ' || c == '\t' || c == '\n' || c == '\r') { if (!tok.empty()) { mArgs.push_back(tok); tok.clear(); } }
			else if (c == '/') { if (++slashes >= 2) comment = true; }
			else tok += c;
		}
		if (c == '\n') { comment = false; slashes = 0; }
	}
	if (!tok.empty()) mArgs.push_back(tok);
	std::fclose(fp);
	return true;
}
bool ArgParser::IsKey(const std::string& s) { return s.size() >= 3 && s[0] == '-' && s[s.size() - 1] == '='; }
int ArgParser::FindKeyIndex(const std::string& key) const
{
	std::string k = key;
	if (k.empty()) return -1;
	if (k[0] != '-') k = "-" + k;
	if (k[k.size() - 1] != '=') k += "=";
	for (size_t i = 0; i < mArgs.size(); ++i) if (mArgs[i] == k) return static_cast<int>(i);
	return -1;
}
bool ArgParser::ParseString(const std::string& key, std::string& out) const
{
	int i = FindKeyIndex(key);
	if (i < 0 || i + 1 >= static_cast<int>(mArgs.size())) return false;
	const std::string& v = mArgs[i + 1];
	if (IsKey(v)) return false;
	out = v; return true;
}
bool ArgParser::ParseInt(const std::string& key, int& out) const { std::string s; if (!ParseString(key, s)) return false; out = std::atoi(s.c_str()); return true; }
bool ArgParser::ParseDouble(const std::string& key, double& out) const { std::string s; if (!ParseString(key, s)) return false; out = std::atof(s.c_str()); return true; }
bool ArgParser::ParseBool(const std::string& key, bool& out) const
{
	std::string s; if (!ParseString(key, s)) return false;
	if (s == "true" || s == "1" || s == "True" || s == "T" || s == "t") { out = true; return true; }
	if (s == "false" || s == "0" || s == "False" || s == "F" || s == "f") { out = false; return true; }
	return false;
}

// =====================================================================================================================
// terrain generator (cTerrainGen2D). Strips are emitted through a small builder; the sequence of RNG draws and the
// float/double conversions follow sim/TerrainGen2D.cpp:185-706 so profiles are bit-identical for a given seed.
const char* const kTerrainTypeNames[kTerrTypeMax] = {"flat", "gaps", "steps", "walls", "bumps", "mixed", "narrow_gaps", "slopes", "slopes_gaps",
	"slopes_steps", "slopes_walls", "slopes_mixed", "slopes_narrow_gaps", "cliffs"};
const char* const kTerrainParamNames[kNumTerrainParams] = {
	"GapSpacingMin", "GapSpacingMax", "GapWMin", "GapWMax", "GapHMin", "GapHMax",
	"WallSpacingMin", "WallSpacingMax", "WallWMin", "WallWMax", "WallHMin", "WallHMax",
	"StepSpacingMin", "StepSpacingMax", "StepH0Min", "StepH0Max", "StepH1Min", "StepH1Max",
	"BumpHMin", "BumpHMax",
	"NarrowGapSpacingMin", "NarrowGapSpacingMax", "NarrowGapDistMin", "NarrowGapDistMax", "NarrowGapWMin", "NarrowGapWMax",
	"NarrowGapDepthMin", "NarrowGapDepthMax", "NarrowGapCountMin", "NarrowGapCountMax",
	"CliffSpacingMin", "CliffSpacingMax", "CliffH0Min", "CliffH0Max", "CliffH1Min", "CliffH1Max", "CliffMiniCountMax",
	"SlopeDeltaRange", "SlopeDeltaMin", "SlopeDeltaMax"};
const double kTerrainParamDefaults[kNumTerrainParams] = {4, 7, 0.5, 2, -2, -2, 6, 8, 0.2, 0.2, 0.25, 0.5, 5, 7, 0.1, 0.4, -0.4, -0.1, 0, 0.03,
	3, 6, 0.1, 0.4, 0.15, 0.5, -2, -2, 1, 4, 5, 7, 0.1, 0.4, -0.4, -0.1, 0, 0.25, -0.35, 0.35};

double BuildTerrain(int type, double width, const double* p, TerrainRand& rnd, std::vector<float>& out)
{
	return tgen::build_terrain(type, width, p, rnd, out);   // dtrl_terrain_gen.h: the generator shared with the on-device path
}

// =====================================================================================================================
// cGroundVar2D window (sim/GroundVar2D.cpp)
double GroundWindow::Seg::MinX() const { return data.empty() ? std::numeric_limits<double>::infinity() : min_x; }
double GroundWindow::Seg::MaxX() const { return data.empty() ? -std::numeric_limits<double>::infinity() : min_x + (data.size() - 1) * static_cast<double>(tgen::kSpacing); }

void GroundWindow::Configure(int type, const double* params, double world_scale, double segment_width)
{
	type_ = type; world_scale_ = world_scale; segment_width_ = segment_width; SetParams(params);
}
void GroundWindow::SetParams(const double* params) { std::memcpy(params_, params, sizeof(params_)); }
void GroundWindow::Clear() { segs_[0].data.clear(); segs_[1].data.clear(); flip_ = false; }

void GroundWindow::BuildSegment(int seg_id, double bmin, double bmax, bool align_min, double fix_y)
{
	// sim/GroundVar2D.cpp:312-355: optional flat padding around x = 0, terrain strip, C0 alignment at the seam, then
	// tSegment::Init (:392-455) whose Bullet round trips fix the float-rounded origin / x scaling used by CalcGridCoord.
	Seg& seg = segs_[seg_id];
	seg.data.clear();
	if (bmin <= 0 && bmax >= 0) { tgen::Strip<std::vector<float>> s(seg.data); s.flat(std::min(bmax - bmin, 1 - bmin)); }
	BuildTerrain(type_, bmax - bmin, params_, rand_, seg.data);
	const int n = static_cast<int>(seg.data.size());
	float end_h = n > 0 ? (align_min ? seg.data[0] : seg.data[n - 1]) : 0.f;
	float off = static_cast<float>(fix_y - end_h);
	for (float& v : seg.data) v += off;
	const double sp = static_cast<double>(tgen::kSpacing);
	seg.min_x = align_min ? bmin : (bmax - (n - 1) * sp);
	double centre = 0.5 * (seg.min_x + (seg.min_x + (seg.data.size() - 1) * sp));
	float bt_origin = static_cast<float>(world_scale_) * static_cast<float>(centre);
	seg.origin_x = static_cast<double>(bt_origin) / world_scale_;
	seg.scale_x = static_cast<double>(static_cast<float>(sp * world_scale_)) / world_scale_;
	++builds_;
}
void GroundWindow::InitSegments(double bound_min_x, double bound_max_x)
{
	Clear();
	double mid = 0.5 * (bound_max_x + bound_min_x);
	for (int i = 0; i < 2; ++i) {
		bool align_min = flip_ ? (i == 0) : (i != 0);
		double w = segment_width_;
		BuildSegment(SegID(i), (align_min ? 0 : -w) + mid, (align_min ? w : 0) + mid, align_min, 0.0);
	}
}
bool GroundWindow::NeedsUpdate(double bmin, double bmax) const
{
	const Seg& lo = segs_[SegID(0)]; const Seg& hi = segs_[SegID(1)];
	return !(bmax < hi.MaxX() && bmin > lo.MinX());
}
bool GroundWindow::Update(double bmin, double bmax)
{
	const Seg& lo = segs_[SegID(0)]; const Seg& hi = segs_[SegID(1)];
	double min_x = lo.MinX(), max_x = hi.MaxX();
	if (bmax < max_x && bmin > min_x) return false;
	if (bmax <= min_x || bmin >= max_x) { InitSegments(bmin, bmax); return true; }
	bool unflipped = (SegID(0) == 0);
	if (bmax >= max_x) BuildSegment(SegID(0), max_x, max_x + segment_width_, true, hi.data.back());
	else BuildSegment(SegID(1), min_x - segment_width_, min_x, false, lo.data.front());
	flip_ = unflipped;
	return true;
}
bool GroundWindow::FillRecord(GroundRec& rec, std::string& err) const
{
	for (int s = 0; s < 2; ++s) {
		const Seg& seg = segs_[SegID(s)];
		if (static_cast<int>(seg.data.size()) > kSegCap) { err = "terrain segment exceeds kSegCap vertices"; return false; }
		rec.origin_x[s] = seg.origin_x; rec.scale_x[s] = seg.scale_x; rec.min_x[s] = seg.MinX(); rec.max_x[s] = seg.MaxX();
		rec.w[s] = static_cast<int32_t>(seg.data.size());
		std::memcpy(rec.data[s], seg.data.data(), seg.data.size() * sizeof(float));
	}
	return true;
}

// =====================================================================================================================
// scenario loading
namespace {
std::string JoinPath(const std::string& root, const std::string& rel)
{
	if (rel.empty() || rel[0] == '/' || root.empty()) return rel;
	return root + (root.back() == '/' ? "" : "/") + rel;
}
const char* const kDogMisc[6] = {"TransTime", "Cv", "BackForceX", "BackForceY", "FrontForceX", "FrontForceY"};
const char* const kDogStates[4] = {"BackStance", "Extend", "FrontStance", "Gather"};
const char* const kDogStateParams[6] = {"SpineCurve", "Shoulder", "Elbow", "Hip", "Knee", "Ankle"};
// sim/SimDog.cpp:5-33
const int kDogCol[21] = {2, 2, 2, 2, 2, 2, 2, 2, 2, 0, 0, 0, 0, 4, 4, 4, 4, 8, 8, 8, 8};
// raptor: sim/RaptorController.cpp:37-55 parameter names, :71-114 gOptParamsMasks, sim/SimRaptor.cpp:5-29 collision groups
const char* const kRaptorMisc[5] = {"TransTime", "Cv", "Cd", "ForceX", "ForceY"};
const char* const kRaptorStates[4] = {"Contact", "Down", "Passing", "Up"};
const char* const kRaptorStateParams[8] = {"RootPitch", "SpineCurve", "StanceHip", "StanceKnee", "StanceAnkle", "SwingHip", "SwingKnee", "SwingAnkle"};
const int kRaptorOptMask[37] = {0, 1, 1, 0, 0, 1, 0, 1, 1, 1, 1, 1, 1, 1, 0, 1, 1, 1, 1, 1, 1, 0, 0, 1, 1, 1, 1, 1, 1, 0, 0, 1, 1, 1, 1, 1, 1};
const int kRaptorCol[19] = {2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 4, 4, 4, 4, 4, 4, 4, 4};
}  // namespace

bool ParseDeployPrototxt(const std::string& path, NetDesc& d, std::string& err, bool* actor_only)
{
	std::ifstream f(path);
	if (!f.is_open()) { err = "cannot open " + path; return false; }
	std::stringstream ss; ss << f.rdbuf();
	const std::string txt = ss.str();
	std::vector<int> dims;
	{
		std::regex re("input_dim:\\s*(\\d+)");
		for (auto it = std::sregex_iterator(txt.begin(), txt.end(), re); it != std::sregex_iterator(); ++it) dims.push_back(std::stoi((*it)[1]));
	}
	if (dims.empty()) { err = path + ": no input_dim"; return false; }
	struct Layer { std::string name, type; int num_output = -1, kernel_w = -1, slice_point = -1; };
	std::vector<Layer> layers;
	{
		std::regex split("\\blayer\\s*\\{");
		std::sregex_token_iterator it(txt.begin(), txt.end(), split, -1), end;
		bool first = true;
		for (; it != end; ++it) {
			if (first) { first = false; continue; }
			const std::string blk = *it;
			Layer l; std::smatch m;
			if (std::regex_search(blk, m, std::regex("name:\\s*\"([^\"]+)\""))) l.name = m[1];
			if (std::regex_search(blk, m, std::regex("type:\\s*\"([^\"]+)\""))) l.type = m[1];
			if (std::regex_search(blk, m, std::regex("num_output:\\s*(\\d+)"))) l.num_output = std::stoi(m[1]);
			if (std::regex_search(blk, m, std::regex("kernel_w:\\s*(\\d+)"))) l.kernel_w = std::stoi(m[1]);
			if (std::regex_search(blk, m, std::regex("slice_point:\\s*(\\d+)"))) l.slice_point = std::stoi(m[1]);
			layers.push_back(l);
		}
	}
	std::map<std::string, int> ips; int nconv = 0; d.n_terrain = -1;
	for (const Layer& l : layers) {
		if (l.type == "Slice") d.n_terrain = l.slice_point;
		else if (l.type == "Convolution") { if (nconv < 3) { d.conv_ch[nconv] = l.num_output; d.conv_k[nconv] = l.kernel_w; } ++nconv; }
		else if (l.type == "InnerProduct") ips[l.name] = l.num_output;
	}
	// the CACLA actor (data/policies/dog/nets/dog_actor_deploy.prototxt): same trunk, ONE head ip1 -> ip2 -> output and no critic outputs
	const bool actor = d.n_terrain >= 0 && nconv == 3 && ips.count("terr_ip0") && ips.count("ip1") && ips.count("ip2") && ips.count("output") && !ips.count("ip0");
	if (actor_only) *actor_only = actor;
	if (!actor && (d.n_terrain < 0 || nconv != 3 || !ips.count("terr_ip0") || !ips.count("ip0") || !ips.count("val_ip0") || !ips.count("val_ip1") || !ips.count("a0_ip1"))) {
		err = path + ": not a MACE (slice/3 conv/terr_ip0/ip0/val/a*) or actor (slice/3 conv/terr_ip0/ip1/ip2/output) deploy net"; return false;
	}
	d.n_char = dims.back() - d.n_terrain;
	if (actor) { d.fc_terr = ips["terr_ip0"]; d.fc_trunk = ips["ip1"]; d.fc_head = ips["ip2"]; d.n_frags = 1; d.frag_size = ips["output"]; }
	else { d.fc_terr = ips["terr_ip0"]; d.fc_trunk = ips["ip0"]; d.fc_head = ips["val_ip0"]; d.n_frags = ips["val_ip1"]; d.frag_size = ips["a0_ip1"]; }
	if (d.n_frags > kMaxFrags) { err = "too many actor fragments"; return false; }
	for (int l = 0; l < 3; ++l) if ((d.conv_ch[l] != 16 && d.conv_ch[l] != 32) || (d.conv_k[l] != 4 && d.conv_k[l] != 8)) { err = path + ": conv layer outside the supported family (16 or 32 channels, kernel width 4 or 8)"; return false; }
	{
		// the forward keeps its activations in the env's LDS workspace (dtrl_kernel.h nn_eval): tiles of kConvTile conv0 positions, of which V survive conv2
		const int s0 = kConvTile + d.conv_k[1] - 1, s1 = kConvTile + d.conv_k[2] - 1, V = kConvTile - (d.conv_k[1] - 1) - (d.conv_k[2] - 1);
		const bool fits = V >= 1 && d.conv_ch[0] * s0 <= kNNTileBuf && d.conv_ch[1] * s1 <= kNNTileBuf && d.conv_ch[2] * V <= kNNTileBuf && d.n_terrain + d.n_char <= kNNSideBuf
			&& d.fc_terr <= kGroup && 2 * kFcChunk + d.fc_terr + d.n_char + d.fc_trunk + d.fc_head <= kNNTileBuf;
		if (!fits) { err = path + ": layer sizes exceed the per-env on-chip workspace of the policy forward"; return false; }
	}
	d.in_size = d.n_terrain + d.n_char; d.out_size = d.n_frags + d.n_frags * d.frag_size;
	int64_t n = 0; int cin = 1, w = d.n_terrain;
	for (int l = 0; l < 3; ++l) { n += static_cast<int64_t>(d.conv_ch[l]) * cin * d.conv_k[l] + d.conv_ch[l]; cin = d.conv_ch[l]; w = w - d.conv_k[l] + 1; }
	n += static_cast<int64_t>(d.fc_terr) * cin * w + d.fc_terr;
	n += static_cast<int64_t>(d.fc_trunk) * (d.fc_terr + d.n_char) + d.fc_trunk;
	n += (static_cast<int64_t>(d.fc_head) * d.fc_trunk + d.fc_head) * (1 + d.n_frags);
	n += static_cast<int64_t>(d.n_frags) * d.fc_head + d.n_frags + static_cast<int64_t>(d.n_frags) * (static_cast<int64_t>(d.frag_size) * d.fc_head + d.frag_size);
	d.num_params = n;
	return true;
}

void LerpTerrainParams(const ScenarioConfig& cfg, double lerp, double* out)
{
	// scenarios/ScenarioSimChar.cpp:255-272
	const int n = static_cast<int>(cfg.terrain_param_sets.size());
	if (n == 0) { std::memcpy(out, kTerrainParamDefaults, sizeof(kTerrainParamDefaults)); return; }
	lerp = std::min(std::max(lerp, 0.0), n - 1.0);
	int i0 = static_cast<int>(lerp), i1 = std::min(i0 + 1, n - 1);
	lerp -= i0;
	for (int k = 0; k < kNumTerrainParams; ++k) out[k] = (1 - lerp) * cfg.terrain_param_sets[i0][k] + lerp * cfg.terrain_param_sets[i1][k];
}

void BuildOutputOffsetScale(const DevModel& m, const NetDesc& d, std::vector<double>& off, std::vector<double>& scale)
{
	// sim/BaseControllerMACE.cpp:75-168 with cDogControllerMACE::BuildActorBias (sim/DogControllerMACE.cpp:93-99)
	const int frag = m.n_opt, nf = d.n_frags;
	auto action_opt = [&](int a, std::vector<double>& out) {
		out.resize(frag);
		const real* p0 = m.ctrl_params[m.act_idx0[a]]; const real* p1 = m.ctrl_params[m.act_idx1[a]]; double b = m.act_blend[a];
		for (int k = 0; k < frag; ++k) { int i = m.opt_index[k]; out[k] = (1 - b) * p0[i] + b * p1[i]; }
	};
	std::vector<double> f_off, f_scale(frag, 1.0), tmp;
	int da = m.default_action < 0 ? 0 : m.default_action;
	action_opt(da, f_off);
	for (double& v : f_off) v = -v;
	if (m.n_actions > 1) {
		std::fill(f_scale.begin(), f_scale.end(), 0.0);
		for (int a = 0; a < m.n_actions; ++a) if (a != da) { action_opt(a, tmp); for (int k = 0; k < frag; ++k) f_scale[k] = std::max(f_scale[k], std::fabs(tmp[k] + f_off[k])); }
		for (double& v : f_scale) v = 1.0 / v;
	}
	if (m.ctrl_type == 2) {   // cBaseControllerCacla::BuildNNOutputOffsetScale (sim/BaseControllerCacla.cpp:88-122): the parameter block alone
		off = f_off; scale = f_scale; return;
	}
	if (m.ctrl_type == 0) {   // cBaseControllerQ::BuildNNOutputOffsetScale (sim/BaseControllerQ.cpp:25-30): one value per base action
		off.assign(m.n_actions, -0.5); scale.assign(m.n_actions, 2.0); return;
	}
	off.assign(nf + nf * frag, 0.0); scale.assign(nf + nf * frag, 1.0);
	for (int f = 0; f < nf; ++f) {
		off[f] = -0.5; scale[f] = 2;
		const real* bias = m.ctrl_params[f % m.n_sets];
		for (int k = 0; k < frag; ++k) { off[nf + f * frag + k] = -bias[m.opt_index[k]]; scale[nf + f * frag + k] = f_scale[k]; }
	}
}

bool LoadScenario(const ArgParser& args, ScenarioConfig& cfg, std::string& err)
{
	DevModel& m = cfg.model;
	std::string root; args.ParseString("data_root", root); cfg.data_root = root;
	std::string char_file, state_file, char_type, char_ctrl, terrain_file, scenario;
	if (!args.ParseString("character_file", char_file)) { err = "No character file specified."; return false; }
	args.ParseString("state_file", state_file);
	args.ParseString("char_type", char_type); args.ParseString("char_ctrl", char_ctrl);
	args.ParseString("terrain_file", terrain_file); args.ParseString("scenario", scenario);
	double world_scale = 1; int num_update_steps = 20, num_sim_substeps = 1;   // scenarios/ScenarioSimChar.cpp:51-53
	args.ParseDouble("world_scale", world_scale); args.ParseInt("num_update_steps", num_update_steps); args.ParseInt("num_sim_substeps", num_sim_substeps);
	m.world_scale = world_scale; m.num_update_steps = num_update_steps; m.num_sim_substeps = num_sim_substeps;

	// scenarios/ScenarioSimChar.cpp:19-36 controller names
	// "dog" / "raptor" build cDogControllerQ / cRaptorControllerQ (scenarios/ScenarioSimChar.cpp:421-430, 469-478): the plain FSM without a net,
	// the Q head (sim/BaseControllerQ.cpp) once -policy_net= names a net with one output per base action
	if (char_ctrl == "dog") { m.char_type = 0; m.ctrl_type = 0; }
	else if (char_ctrl == "dog_mace" || char_ctrl == "goat_mace") { m.char_type = 0; m.ctrl_type = 1; }
	else if (char_ctrl == "raptor") { m.char_type = 1; m.ctrl_type = 0; }
	else if (char_ctrl == "raptor_mace") { m.char_type = 1; m.ctrl_type = 1; }
	else if (char_ctrl == "dog_cacla") { m.char_type = 0; m.ctrl_type = 2; }   // cDogControllerCacla (sim/DogControllerCacla.cpp)
	// cRaptorControllerCacla (sim/RaptorControllerCacla.cpp; built by scenarios/ScenarioSimChar.cpp:407, 480-483): the raptor FSM with the CACLA head,
	// mExpNoise 0.15 (set below by character type). The reference ships no raptor actor net: -policy_net= must name a single-head deploy net 275 -> 28
	else if (char_ctrl == "raptor_cacla") { m.char_type = 1; m.ctrl_type = 2; }
	else { err = "char_ctrl '" + char_ctrl + "' is not supported by this build (dog, dog_mace, dog_cacla, goat_mace, raptor, raptor_mace, raptor_cacla)"; return false; }
	if (!char_type.empty() && char_type != (m.char_type == 0 ? "dog" : "raptor")) { err = "char_type '" + char_type + "' does not match the controller"; return false; }
	m.target_vel_x = (char_ctrl == "goat_mace") ? 2.0 : 4.0;   // sim/GoatControllerMACE.cpp:11-14, sim/DogController.cpp:625-628
	if (scenario == "train" || scenario == "train_mace" || scenario == "exp" || scenario == "exp_mace" || scenario == "train_cacla" || scenario == "exp_cacla") m.scenario = kScnExp;
	else if (scenario == "poli_eval") m.scenario = kScnPoliEval;
	else m.scenario = kScnSimChar;

	Json chr;
	if (!Json::parse_file(JoinPath(root, char_file), chr, err)) return false;
	const Json* skel = chr.find("Skeleton"); const Json* joints = skel ? skel->find("Joints") : nullptr;
	const Json* bodies = chr.find("BodyDefs"); const Json* pds = chr.find("PDControllers"); const Json* ctrls = chr.find("Controllers");
	if (!joints || !bodies || !pds || !ctrls) { err = char_file + ": missing Skeleton/BodyDefs/PDControllers/Controllers"; return false; }
	const int L = static_cast<int>(joints->arr.size());
	if (L > kMaxL || L != static_cast<int>(bodies->arr.size()) || L != static_cast<int>(pds->arr.size())) { err = char_file + ": inconsistent joint/body/PD counts"; return false; }
	if (m.char_type == 0 && L != 21) { err = "dog controller expects 21 joints"; return false; }
	if (m.char_type == 1 && L != 19) { err = "raptor controller expects 19 joints"; return false; }
	m.L = L;
	int D = 0;
	for (int j = 0; j < L; ++j) {
		const Json& jd = joints->arr[j];
		int type = static_cast<int>(jd.get_num("Type", 0)); int parent = static_cast<int>(jd.get_num("Parent", -1));
		if (parent >= j) { err = "Parent id must be < child id"; return false; }     // anim/KinTree.cpp:441-447
		if ((j == 0) != (type == 1) || (j > 0 && type != 0)) { err = "only planar root + revolute joints are supported"; return false; }
		m.parent[j] = parent;
		bool is_root = parent < 0;
		m.attach[j][0] = is_root ? 0 : jd.get_num("AttachX", 0); m.attach[j][1] = is_root ? 0 : jd.get_num("AttachY", 0);
		m.lim_lo[j] = jd.get_num("LimLow", 1); m.lim_hi[j] = jd.get_num("LimHigh", 0);                // anim/KinTree.cpp:990-1006 (shifted by ref_theta below)
		D += (type == 1) ? 3 : 1;
	}
	m.D = D;
	if (D != L + 2 || D > kMaxD) { err = "unexpected DoF count"; return false; }
	m.total_mass = 0;
	for (int j = 0; j < L; ++j) {
		const Json& bd = bodies->arr[j];
		const Json* shape = bd.find("Shape");
		if (!shape || shape->str != "box") { err = "only box bodies are supported"; return false; }
		m.mass[j] = bd.get_num("Mass", 0); m.total_mass += m.mass[j];
		m.body_attach[j][0] = bd.get_num("AttachX", 0); m.body_attach[j][1] = bd.get_num("AttachY", 0);
		m.body_theta[j] = bd.get_num("Theta", 0);
		double sx = bd.get_num("Param0", 0), sy = bd.get_num("Param1", 0);
		m.body_half[j][0] = 0.5 * sx; m.body_half[j][1] = 0.5 * sy;
		m.inertia[j] = m.mass[j] / 12.0 * (sx * sx + sy * sy);                                     // sim/RBDUtil.cpp:562-583 (zz term)
		m.col[j] = (m.char_type == 0) ? kDogCol[j] : kRaptorCol[j];
		const Json& pd = pds->arr[j];
		m.kp[j] = pd.get_num("Kp", 0); m.kd[j] = pd.get_num("Kd", 0); m.torque_lim[j] = pd.get_num("TorqueLim", 0);
		m.target_theta[j] = pd.get_num("TargetTheta", 0); m.use_world[j] = pd.get_num("UseWorldCoord", 0) != 0;
	}
	// hinge limits act on theta + ref_theta (sim/World.cpp:543-553: theta = -getHingeAngle() - ref_theta; :624-626 setLimit(-LimHigh, -LimLow)) with
	// ref_theta = -angle(BodyJointTrans(parent) * ParentChildTrans(zero pose) * BodyJointTrans(child)) as cSimCharacter::BuildConstraints computes it
	// (sim/SimCharacter.cpp:846-865; RotMatToAxisAngle returns acos of the cosine, util/MathUtil.cpp:128-149): the device tables hold the limits on theta itself
	m.ref_theta[0] = 0;
	for (int j = 1; j < L; ++j) {
		const double cth = std::cos(m.body_theta[m.parent[j]] + m.body_theta[j]);
		const double ref_theta = -std::acos(std::min(1.0, std::max(-1.0, cth)));
		m.ref_theta[j] = ref_theta;                // also the window the controller reads the angle in (dtrl_kernel.h, PD error)
		if (m.lim_lo[j] > m.lim_hi[j]) continue;   // free joint
		m.lim_lo[j] -= ref_theta; m.lim_hi[j] -= ref_theta;
	}
	m.contact_tol = 0.001 / world_scale;   // sim/ContactManager.cpp:74-75, dist_tol in world-scaled units
	{ double margin = 0.04; args.ParseDouble("collision_margin", margin); m.contact_margin = margin / world_scale; }   // CONVEX_DISTANCE_MARGIN (world-scaled units)
	{
		// btBoxShape::btBoxShape -> setSafeMargin(halfExtents, 0.1) on the world-scaled half extents cWorld::BuildBoxShape hands over (sim/World.cpp:475-482): in metres the
		// scale cancels out of the 0.1 x half-extent branch
		int safe = 1; args.ParseInt("safe_margin", safe);
		for (int j = 0; j < L; ++j) {
			const double he = std::min(std::min(static_cast<double>(m.body_half[j][0]), static_cast<double>(m.body_half[j][1])), 0.5 * bodies->arr[j].get_num("Param2", 0));
			m.link_margin[j] = safe ? std::min(static_cast<double>(m.contact_margin), 0.1 * he) : static_cast<double>(m.contact_margin);
		}
	}
	{
		// Bullet's contact persistence (DevModel::warm_start, link_brk): breaking threshold = gContactBreakingThreshold x the box's angular-motion disc |half extents|
		// (btCollisionShape::getContactBreakingThreshold under the dispatcher's default relative flag; world-scaled on both sides, so the scale cancels)
		int ws = 1; args.ParseInt("warm_start", ws);
		if (ws != 0 && ws != 1) { err = "-warm_start= takes 0 or 1"; return false; }
		m.warm_start = ws;
		double brk = 0.02; args.ParseDouble("contact_breaking", brk);
		if (!(brk >= 0)) { err = "-contact_breaking= must be >= 0"; return false; }
		for (int j = 0; j < L; ++j) {
			const double hx = m.body_half[j][0], hy = m.body_half[j][1], hz = 0.5 * bodies->arr[j].get_num("Param2", 0);
			m.link_brk[j] = brk * std::sqrt(hx * hx + hy * hy + hz * hz);
		}
	}
	// link--link collision pairs: same non-zero collision group, no hinge between the two, boxes overlapping in z (joint AttachZ accumulated down the
	// chain + body AttachZ against the box depth Param2: the raptor's legs share a group but sit 0.16 m apart in z with 0.065 m deep boxes)
	{
		double zj[kMaxL], zc[kMaxL], zs[kMaxL];
		for (int j = 0; j < L; ++j) {
			zj[j] = (m.parent[j] >= 0 ? zj[m.parent[j]] : 0.0) + joints->arr[j].get_num("AttachZ", 0);
			zc[j] = zj[j] + bodies->arr[j].get_num("AttachZ", 0);
			zs[j] = bodies->arr[j].get_num("Param2", 0);
			m.bt_cs[j] = std::cos(m.body_theta[j]); m.bt_sn[j] = std::sin(m.body_theta[j]);
		}
		int n = 0;
		for (int a = 0; a < L; ++a) for (int b = a + 1; b < L; ++b) {
			if (m.col[a] == 0 || m.col[a] != m.col[b] || m.parent[b] == a || m.parent[a] == b) continue;
			if (std::fabs(zc[a] - zc[b]) >= 0.5 * (zs[a] + zs[b])) continue;
			if (n >= kMaxCP) { err = "too many link--link collision pairs"; return false; }
			m.cp_a[n] = static_cast<int8_t>(a); m.cp_b[n] = static_cast<int8_t>(b);
			++n;
		}
		for (int j = 0; j < L; ++j) {
			double hx = m.body_half[j][0], hy = m.body_half[j][1];
			if (m.body_theta[j] != 0 && j != 0) hx = hy = std::sqrt(hx * hx + hy * hy);
			const double grow = m.contact_tol + 1e-5;   // float rounding of the extents and of the test itself stays inside this
			m.cp_half[j][0] = static_cast<float>((hx + grow) * 1.00001); m.cp_half[j][1] = static_cast<float>((hy + grow) * 1.00001);
		}
		m.n_cpairs = n;
		m.cp_root_bt[0] = static_cast<float>(std::cos(m.body_theta[0])); m.cp_root_bt[1] = static_cast<float>(std::sin(m.body_theta[0]));
		int lc = 1; args.ParseInt("link_contacts", lc);
		m.link_contacts = lc != 0;
	}
	// contact sample points (4 corners + long-edge midpoints, DESIGN.md "Integrator v1") and end-effector points, joint frame
	for (int j = 0; j < L; ++j) {
		const double hx = m.body_half[j][0], hy = m.body_half[j][1];
		const double c = std::cos(m.body_theta[j]), s = std::sin(m.body_theta[j]);
		const double loc[kPtsPerLink][2] = {{-hx, -hy}, {hx, -hy}, {hx, hy}, {-hx, hy},
			{hx >= hy ? 0.0 : -hx, hx >= hy ? -hy : 0.0}, {hx >= hy ? 0.0 : hx, hx >= hy ? hy : 0.0}};
		// against the ground the boxes carry Bullet's collision margin (btBoxShape: core = half extents - margin, rounded by the margin; with the safe margin
		// the core never inverts): the ground test measures core point -> surface and subtracts the margin
		const double gx = hx - m.link_margin[j], gy = hy - m.link_margin[j];
		const double locg[kPtsPerLink][2] = {{-gx, -gy}, {gx, -gy}, {gx, gy}, {-gx, gy},
			{hx >= hy ? 0.0 : -gx, hx >= hy ? -gy : 0.0}, {hx >= hy ? 0.0 : gx, hx >= hy ? gy : 0.0}};
		for (int k = 0; k < kPtsPerLink; ++k) {
			m.pt_joint[j][k][0] = m.body_attach[j][0] + c * loc[k][0] - s * loc[k][1];
			m.pt_joint[j][k][1] = m.body_attach[j][1] + s * loc[k][0] + c * loc[k][1];
			m.pt_ground[j][k][0] = m.body_attach[j][0] + c * locg[k][0] - s * locg[k][1];
			m.pt_ground[j][k][1] = m.body_attach[j][1] + s * locg[k][0] + c * locg[k][1];
		}
		m.eff_joint[j][0] = m.body_attach[j][0] - s * (-hy);
		m.eff_joint[j][1] = m.body_attach[j][1] + c * (-hy);
	}
	// tree tables: root->link paths and subtree masks
	for (int j = 0; j < L; ++j) {
		int chain[kMaxL]; int n = 0;
		for (int c = j; c >= 0; c = m.parent[c]) chain[n++] = c;
		if (n > kMaxDepth) { err = "kinematic chain too deep"; return false; }
		m.depth[j] = n - 1;
		for (int k = 0; k < n; ++k) m.path[j][k] = static_cast<int8_t>(chain[n - 1 - k]);
		m.sub_mask[j] = 0;
	}
	for (int k = 0; k < L; ++k) { m.anc_mask[k] = 0; for (int c = k; c >= 0; c = m.parent[c]) { m.sub_mask[c] |= (1u << k); m.anc_mask[k] |= (1u << c); } }
	for (int j = 0; j < L; ++j) { double sm = 0; for (int k = 0; k < L; ++k) if ((m.sub_mask[j] >> k) & 1u) sm += m.mass[k]; m.sub_mass[j] = sm; }
	m.n_pairs = 0;
	for (int l = 0; l < L; ++l) for (int k = 0; k <= m.depth[l]; ++k) { m.pair_l[m.n_pairs] = static_cast<int8_t>(l); m.pair_k[m.n_pairs] = static_cast<int8_t>(k); ++m.n_pairs; }

	// controllers (sim/DogController.cpp:629-700, 399-454)
	const Json* files = ctrls->find("Files"); const Json* acts = ctrls->find("Actions");
	if (!files || !acts) { err = "Controllers block needs Files and Actions"; return false; }
	const bool raptor = m.char_type == 1;
	m.P = raptor ? 37 : 30; m.n_opt = 0;
	for (int i = 0; i < m.P; ++i) if (raptor ? kRaptorOptMask[i] != 0 : i != 0) m.opt_index[m.n_opt++] = i;   // dog gParamInfo: only TransTime is not optimisable
	m.n_sets = static_cast<int>(files->arr.size());
	if (m.n_sets > kMaxSets) { err = "too many controller files"; return false; }
	for (int s = 0; s < m.n_sets; ++s) {
		Json cf;
		if (!Json::parse_file(JoinPath(root, files->arr[s].str), cf, err)) return false;
		const Json* misc = cf.find("MiscParams"); const Json* sps = cf.find("StateParams");
		if (!misc || !sps) { err = files->arr[s].str + ": missing MiscParams/StateParams"; return false; }
		int idx = 0;
		if (raptor) {   // sim/RaptorController.cpp:442-495
			for (int i = 0; i < 5; ++i) m.ctrl_params[s][idx++] = misc->get_num(kRaptorMisc[i], 0);
			for (int st = 0; st < 4; ++st) { const Json* sp = sps->find(kRaptorStates[st]); for (int i = 0; i < 8; ++i) m.ctrl_params[s][idx++] = sp ? sp->get_num(kRaptorStateParams[i], 0) : 0; }
			m.ctrl_params[s][2] = std::fabs(m.ctrl_params[s][2]);
		} else {
			for (int i = 0; i < 6; ++i) m.ctrl_params[s][idx++] = misc->get_num(kDogMisc[i], 0);
			for (int st = 0; st < 4; ++st) { const Json* sp = sps->find(kDogStates[st]); for (int i = 0; i < 6; ++i) m.ctrl_params[s][idx++] = sp ? sp->get_num(kDogStateParams[i], 0) : 0; }
		}
		m.ctrl_params[s][0] = std::fabs(m.ctrl_params[s][0]); m.ctrl_params[s][1] = std::fabs(m.ctrl_params[s][1]);   // PostProcessParams
	}
	m.n_actions = static_cast<int>(acts->arr.size());
	if (m.n_actions > kMaxAct) { err = "too many actions"; return false; }
	for (int a = 0; a < m.n_actions; ++a) {
		const Json& ad = acts->arr[a];
		if (!ad.has("ParamIdx0") || !ad.has("ParamIdx1") || !ad.has("Blend") || !ad.has("Cyclic")) { err = "failed to parse actions"; return false; }
		m.act_idx0[a] = static_cast<int>(ad.get_num("ParamIdx0", 0)); m.act_idx1[a] = static_cast<int>(ad.get_num("ParamIdx1", 0));
		m.act_blend[a] = ad.get_num("Blend", 0); m.act_cyclic[a] = ad.get_bool("Cyclic", false);
	}
	m.default_action = m.n_actions > 0 ? static_cast<int>(ctrls->get_num("DefaultAction", 0)) : -1;
	m.enable_grav_comp = ctrls->get_bool("EnableGravityCompensation", true);   // ctor default sim/DogController.cpp:172
	m.enable_vf = raptor ? ctrls->get_bool("EnableVirtualForces", true) : 1;   // sim/RaptorController.cpp:705-708; the dog always applies them

	// initial state (anim/Character.cpp:239-262)
	for (int i = 0; i < D; ++i) { m.pose0[i] = 0; m.vel0[i] = 0; }
	if (!state_file.empty()) {
		Json st;
		if (!Json::parse_file(JoinPath(root, state_file), st, err)) return false;
		const Json* pose = st.find("Pose"); const Json* vel = st.find("Vel");
		if (!pose || !vel || static_cast<int>(pose->arr.size()) != D || static_cast<int>(vel->arr.size()) != D) { err = state_file + ": Pose/Vel size mismatch"; return false; }
		for (int i = 0; i < D; ++i) { m.pose0[i] = pose->arr[i].num; m.vel0[i] = vel->arr[i].num; }
	}
	double init_x = 0; m.valid_init_pos_x = args.ParseDouble("char_init_pos_x", init_x); m.init_pos_x = init_x;

	// terrain (scenarios/ScenarioSimChar.cpp:670-706, sim/TerrainGen2D.cpp:58-146)
	cfg.terrain_type = kTerrFlat; cfg.terrain_param_sets.clear();
	if (!terrain_file.empty()) {
		Json tf;
		if (!Json::parse_file(JoinPath(root, terrain_file), tf, err)) return false;
		const Json* ty = tf.find("Type");
		std::string tname = ty ? ty->str : "";
		if (tname.empty()) tname = "flat";
		int found = -1;
		for (int i = 0; i < kTerrTypeMax; ++i) if (tname == kTerrainTypeNames[i]) found = i;
		if (found < 0) { err = "unsupported terrain type " + tname; return false; }
		cfg.terrain_type = found;
		const Json* ps = tf.find("Params");
		if (ps) for (const Json& obj : ps->arr) {
			std::vector<double> v(kNumTerrainParams);
			for (int k = 0; k < kNumTerrainParams; ++k) v[k] = obj.get_num(kTerrainParamNames[k], kTerrainParamDefaults[k]);
			cfg.terrain_param_sets.push_back(v);
		}
	}
	args.ParseDouble("terrain_blend", cfg.terrain_blend);
	args.ParseDouble("min_perturb", cfg.min_perturb); args.ParseDouble("max_perturb", cfg.max_perturb);
	args.ParseDouble("min_pertrub_duration", cfg.min_perturb_duration); args.ParseDouble("max_perturb_duration", cfg.max_perturb_duration);
	int seed = 0; if (args.ParseInt("terrain_seed", seed)) cfg.terrain_seed = static_cast<uint64_t>(seed);
	{
		{
			// -physics_precision= f64 | f32: WHICH LIBRARY must be loaded, checked here: libdtrl.so computes in fp64 (the default and the parity-tested product),
			// libdtrl_f32.so is the same source built with `real` = float (dtrl_types.h) -- an opt-in mode with distribution-level parity only
			const std::string built = sizeof(real) == 4 ? "f32" : "f64";
			std::string prec = built; args.ParseString("physics_precision", prec);
			if (prec != built) { err = "-physics_precision= " + prec + ": this library computes in " + built + " (fp64 = libdtrl.so, fp32 = libdtrl_f32.so)"; return false; }
		}
		std::string gen = "host"; args.ParseString("terrain_gen", gen);
		if (gen != "host" && gen != "device") { err = "-terrain_gen= must be host (the reference's generator streams, bit-exact) or device (counter-based streams, generated on the GPU)"; return false; }
		cfg.device_terrain = gen == "device";
	}

	// exploration (scenarios/ScenarioExp.cpp:16-45)
	cfg.tuple_buffer_size = 16; args.ParseInt("tuple_buffer_size", cfg.tuple_buffer_size);
	cfg.tuple_ring_capacity = 0; args.ParseInt("tuple_ring_capacity", cfg.tuple_ring_capacity);
	{
		std::string where = "device"; args.ParseString("tuple_ring", where);
		if (where != "device" && where != "host") { err = "-tuple_ring= must be device or host (page-locked host memory written by the kernels, read by dtrl_drain_tuples without a copy being queued)"; return false; }
		cfg.tuple_ring_host = where == "host";
	}
	double exp_rate = 0.1, exp_temp = 1, exp_base = 0.01;   // cScenarioExp ctor defaults; mExpTemp is uninitialised there -> controller default 1
	args.ParseDouble("exp_rate", exp_rate); args.ParseDouble("exp_temp", exp_temp); args.ParseDouble("exp_base_rate", exp_base);
	cfg.run.enable_exp = (m.scenario == kScnExp) ? 1 : 0;
	cfg.run.exp_rate = exp_rate; cfg.run.exp_temp = exp_temp; cfg.run.exp_base_rate = exp_base; cfg.run.exp_noise = (m.char_type == 1) ? 0.15 : 0.2;   // mExpNoise, sim/DogControllerMACE.cpp:7, sim/RaptorControllerMACE.cpp:7
	int rseed = 0; args.ParseInt("rand_seed", rseed); cfg.run.rng_seed = static_cast<uint64_t>(rseed);
	int goff = 0; args.ParseInt("global_env_offset", goff); cfg.run.env_id_base = goff;

	// policy net topology (weights arrive through dtrl_set_policy: the shipped *.h5 blobs are not in the reference checkout)
	cfg.has_policy_net = false; m.has_net = 0;
	if (args.ParseString("policy_net", cfg.policy_net_file)) {
		if (!ParseDeployPrototxt(JoinPath(root, cfg.policy_net_file), cfg.net, err, &cfg.actor_only)) return false;
		if (cfg.actor_only != (m.ctrl_type != 1)) { err = "policy_net topology does not match char_ctrl (MACE nets for *_mace, the single-head actor / Q net for dog_cacla, raptor_cacla, dog, raptor)"; return false; }
		cfg.user_num_params = cfg.net.num_params; cfg.user_out_size = cfg.net.out_size;
		if (cfg.actor_only) {   // minus the (zero) critic head val_ip0 / val_ip1 that only exists on the device
			cfg.user_num_params -= static_cast<int64_t>(cfg.net.fc_head) * cfg.net.fc_trunk + cfg.net.fc_head + cfg.net.fc_head + 1;
			cfg.user_out_size = cfg.net.frag_size;
		}
		const int S = kNumGroundSamples + (2 * L - 1) + 2 * L;
		if (cfg.net.in_size != S) { err = "Network input dimension does not match expected input size"; return false; }      // sim/NNController.cpp:58-75
		if (cfg.net.frag_size != (m.ctrl_type == 0 ? m.n_actions : m.n_opt)) { err = "Network output dimension does not match expected output size"; return false; }
		cfg.has_policy_net = true;
	}
	args.ParseString("policy_model", cfg.policy_model_file);
	return true;
}

}  // namespace dtrl
