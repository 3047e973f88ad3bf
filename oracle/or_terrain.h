// ORACLE (test infrastructure, NOT product code).
// CPU restatement of the reference's terrain generator and 2-segment sliding heightfield.
//
// Follows (file:line relative to /root/reference):
//   util/Rand.cpp:6-102            cRand = std::default_random_engine + uniform_real/int/normal (libstdc++)
//   sim/TerrainGen2D.cpp:4-56      gVertSpacing (float 0.1f), 40 default params
//   sim/TerrainGen2D.cpp:148-518   Build* terrain functions
//   sim/TerrainGen2D.cpp:520-706   CalcNumVerts / AddFlat / AddBox / AddStep / AddSlope / OverlaySlopes / OverlayBumps
//   sim/GroundVar2D.cpp:43-91      Update (window slide)        :98-114 SampleHeight (segment pick)
//   sim/GroundVar2D.cpp:239-259    InitSegments                 :312-355 BuildSegment / AddPadding
//   sim/GroundVar2D.cpp:392-455    tSegment::Init (float data x world scale, origin via Bullet float transform)
//   sim/GroundVar2D.cpp:559-633    tSegment::SampleHeight / CalcGridCoord / OutsideGrid / ClampCoord
// The precision sequence (float heights, float vertex spacing, double grid coords, float Bullet origin and
// local scaling divided back by the world scale) is kept so grid indices are bit-exact with the reference.
#pragma once
#include <algorithm>
#include <cmath>
#include <limits>
#include <random>
#include <vector>

namespace orc {

// util/Rand.cpp
struct Rand {
	std::default_random_engine gen;
	std::uniform_real_distribution<double> dreal{0, 1};
	std::normal_distribution<double> dnorm{0, 1};
	std::uniform_int_distribution<int> dint{0, std::numeric_limits<int>::max()};
	void Seed(unsigned long s) { gen.seed(s); }
	double RandDouble() { return dreal(gen); }
	double RandDouble(double mn, double mx) { if (mn == mx) return mn; double r = dreal(gen); return mn + (r * (mx - mn)); }
	int RandInt(int mn, int mx) { if (mn == mx) return mn; int delta = mx - mn; int r = dint(gen); return mn + r % delta; }
	bool FlipCoin(double p = 0.5) { return RandDouble(0, 1) < p; }
	int RandSign() { return FlipCoin() ? -1 : 1; }
};

enum TerrainType { // sim/TerrainGen2D.h:14-29
	eTypeFlat, eTypeGaps, eTypeSteps, eTypeWalls, eTypeBumps, eTypeMixed, eTypeNarrowGaps, eTypeSlopes,
	eTypeSlopesGaps, eTypeSlopesSteps, eTypeSlopesWalls, eTypeSlopesMixed, eTypeSlopesNarrowGaps, eTypeCliffs, eTypeMax
};
enum TerrainParam { // sim/TerrainGen2D.h:31-81
	pGapSpacingMin, pGapSpacingMax, pGapWidthMin, pGapWidthMax, pGapDepthMin, pGapDepthMax,
	pWallSpacingMin, pWallSpacingMax, pWallWidthMin, pWallWidthMax, pWallHeightMin, pWallHeightMax,
	pStepSpacingMin, pStepSpacingMax, pStepHeight0Min, pStepHeight0Max, pStepHeight1Min, pStepHeight1Max,
	pBumpHeightMin, pBumpHeightMax,
	pNarrowGapSpacingMin, pNarrowGapSpacingMax, pNarrowGapDistMin, pNarrowGapDistMax, pNarrowGapWidthMin, pNarrowGapWidthMax,
	pNarrowGapDepthMin, pNarrowGapDepthMax, pNarrowGapCountMin, pNarrowGapCountMax,
	pCliffSpacingMin, pCliffSpacingMax, pCliffHeight0Min, pCliffHeight0Max, pCliffHeight1Min, pCliffHeight1Max, pCliffMiniCountMax,
	pSlopeDeltaRange, pSlopeDeltaMin, pSlopeDeltaMax, pMax
};

struct TerrainGen {
	static constexpr float gVertSpacing = 0.1f;  // sim/TerrainGen2D.cpp:4

	static int CalcNumVerts(double w) { return static_cast<int>(std::ceil(w / gVertSpacing)) + 1; }

	static double AddFlat(double width, std::vector<float>& out)
	{
		int num_verts = CalcNumVerts(width);
		int size0 = static_cast<int>(out.size());
		bool start_empty = size0 == 0;
		float base_h = 0;
		if (!start_empty) { --num_verts; base_h = out[size0 - 1]; }
		for (int i = 0; i < num_verts; ++i) out.push_back(base_h);
		int added = static_cast<int>(out.size()) - size0;
		if (start_empty) --added;
		double width_added = added * gVertSpacing;  // int * float -> float -> double, as in the reference
		return width_added;
	}
	static double AddBox(double spacing, double width, double depth, std::vector<float>& out)
	{
		int num_verts = CalcNumVerts(spacing);
		int size0 = static_cast<int>(out.size());
		bool start_empty = size0 == 0;
		float base_h = 0;
		if (!start_empty) { --num_verts; base_h = out[size0 - 1]; }
		for (int i = 0; i < num_verts; ++i) out.push_back(base_h);
		num_verts = CalcNumVerts(width) - 1;
		float gap_h = static_cast<float>(base_h + depth);
		for (int i = 0; i < num_verts; ++i) out.push_back(gap_h);
		out.push_back(base_h);
		int added = static_cast<int>(out.size()) - size0;
		if (start_empty) --added;
		double width_added = added * gVertSpacing;
		return width_added;
	}
	static double AddStep(double width, double height, std::vector<float>& out)
	{
		int num_verts = CalcNumVerts(width);
		int size0 = static_cast<int>(out.size());
		bool start_empty = size0 == 0;
		float base_h = 0;
		if (!start_empty) { --num_verts; base_h = out[size0 - 1]; }
		for (int i = 0; i < num_verts; ++i) out.push_back(base_h);
		out.push_back(static_cast<float>(base_h + height));
		int added = static_cast<int>(out.size()) - size0;
		if (start_empty) --added;
		double width_added = added * gVertSpacing;
		return width_added;
	}
	static void OverlaySlopes(double delta_range, double delta_min, double delta_max, double init_slope, int beg, int end, Rand& rand, std::vector<float>& out)
	{
		double curr_slope = init_slope;
		double curr_delta_h = 0;
		double delta_mean = 0.5 * (delta_min + delta_max);
		double delta_diff = 0.5 * (delta_max - delta_min);
		for (int i = beg; i < end; ++i) {
			double delta = rand.RandDouble(0, delta_range);
			double sign_rand = rand.RandDouble(-1, 1);
			double sign_threshold = (curr_slope - delta_mean) / delta_diff;
			bool neg = sign_rand < sign_threshold;
			delta = (neg) ? -delta : delta;
			curr_slope += delta;
			curr_delta_h += curr_slope * gVertSpacing;
			out[i] += static_cast<float>(curr_delta_h);
		}
	}
	static void OverlayBumps(double mn, double mx, int beg, int end, Rand& rand, std::vector<float>& out)
	{
		for (int i = beg; i < end - 1; ++i) {
			double delta = rand.RandSign() * rand.RandDouble(mn, mx);
			out[i] += static_cast<float>(delta);
		}
	}

	static double BuildFlat(double width, const double* p, Rand& rand, std::vector<float>& out) { (void)p; (void)rand; return AddFlat(width, out); }
	static double BuildGaps(double width, const double* p, Rand& rand, std::vector<float>& out)
	{
		double total_w = 0;
		while (total_w < width) {
			double spacing = rand.RandDouble(p[pGapSpacingMin], p[pGapSpacingMax]);
			double w = rand.RandDouble(p[pGapWidthMin], p[pGapWidthMax]);
			double d = rand.RandDouble(p[pGapDepthMin], p[pGapDepthMax]);
			total_w += AddBox(spacing, w, d, out);
		}
		return total_w;
	}
	static void PickStepRange(double h0mn, double h0mx, double h1mn, double h1mx, Rand& rand, double& mn, double& mx)
	{
		bool valid_h0 = (h0mn != 0 || h0mx != 0);
		bool valid_h1 = (h1mn != 0 || h1mx != 0);
		if (valid_h0 && valid_h1) { bool heads = rand.FlipCoin(); mn = heads ? h0mn : h1mn; mx = heads ? h0mx : h1mx; }
		else if (valid_h0) { mn = h0mn; mx = h0mx; }
		else { mn = h1mn; mx = h1mx; }
	}
	static double BuildSteps(double width, const double* p, Rand& rand, std::vector<float>& out)
	{
		double total_w = 0;
		while (total_w < width) {
			double mn = 0, mx = 0;
			PickStepRange(p[pStepHeight0Min], p[pStepHeight0Max], p[pStepHeight1Min], p[pStepHeight1Max], rand, mn, mx);
			double w = rand.RandDouble(p[pStepSpacingMin], p[pStepSpacingMax]);
			double h = rand.RandDouble(mn, mx);
			total_w += AddStep(w, h, out);
		}
		return total_w;
	}
	static double BuildWalls(double width, const double* p, Rand& rand, std::vector<float>& out)
	{
		double total_w = 0;
		while (total_w < width) {
			double spacing = rand.RandDouble(p[pWallSpacingMin], p[pWallSpacingMax]);
			double w = rand.RandDouble(p[pWallWidthMin], p[pWallWidthMax]);
			double h = rand.RandDouble(p[pWallHeightMin], p[pWallHeightMax]);
			total_w += AddBox(spacing, w, h, out);
		}
		return total_w;
	}
	static double BuildBumps(double width, const double* p, Rand& rand, std::vector<float>& out)
	{
		int beg = static_cast<int>(out.size());
		double total_w = BuildFlat(width, p, rand, out);
		int end = static_cast<int>(out.size());
		OverlayBumps(p[pBumpHeightMin], p[pBumpHeightMax], beg, end, rand, out);
		return total_w;
	}
	static double BuildMixed(double width, const double* p, Rand& rand, std::vector<float>& out)
	{
		double total_w = 0;
		const int num_types = 3;
		const double dummy_w = gVertSpacing;
		while (total_w < width) {
			double curr_w = 0;
			int rand_type = rand.RandInt(0, num_types);
			if (rand_type == 0) curr_w = BuildGaps(dummy_w, p, rand, out);
			else if (rand_type == 1) curr_w = BuildSteps(dummy_w, p, rand, out);
			else if (rand_type == 2) curr_w = BuildWalls(dummy_w, p, rand, out);
			total_w += curr_w;
		}
		return total_w;
	}
	static double BuildNarrowGaps(double width, const double* p, Rand& rand, std::vector<float>& out)
	{
		int count_min = std::max(1, static_cast<int>(p[pNarrowGapCountMin]));
		int count_max = std::max(1, static_cast<int>(p[pNarrowGapCountMax]));
		double total_w = 0;
		while (total_w < width) {
			double spacing = rand.RandDouble(p[pNarrowGapSpacingMin], p[pNarrowGapSpacingMax]);
			int count = rand.RandInt(count_min, count_max + 1);
			for (int i = 0; i < count; ++i) {
				double w = rand.RandDouble(p[pNarrowGapWidthMin], p[pNarrowGapWidthMax]);
				double d = rand.RandDouble(p[pNarrowGapDepthMin], p[pNarrowGapDepthMax]);
				total_w += AddBox(spacing, w, d, out);
				spacing = rand.RandDouble(p[pNarrowGapDistMin], p[pNarrowGapDistMax]);
			}
		}
		return total_w;
	}
	typedef double (*Func)(double, const double*, Rand&, std::vector<float>&);
	static double WithSlopes(Func f, double width, const double* p, Rand& rand, std::vector<float>& out)
	{
		const double delta_range = std::abs(p[pSlopeDeltaRange]);
		int beg = static_cast<int>(out.size());
		double total_w = f(width, p, rand, out);
		int end = static_cast<int>(out.size());
		OverlaySlopes(delta_range, p[pSlopeDeltaMin], p[pSlopeDeltaMax], 0, beg, end, rand, out);
		return total_w;
	}
	static double BuildCliffs(double width, const double* p, Rand& rand, std::vector<float>& out)
	{
		int mini_count_max = static_cast<int>(p[pCliffMiniCountMax]);
		const double delta_range = std::abs(p[pSlopeDeltaRange]);
		int beg = static_cast<int>(out.size());
		double total_w = 0;
		while (total_w < width) {
			double mn = 0, mx = 0;
			PickStepRange(p[pCliffHeight0Min], p[pCliffHeight0Max], p[pCliffHeight1Min], p[pCliffHeight1Max], rand, mn, mx);
			double w = rand.RandDouble(p[pCliffSpacingMin], p[pCliffSpacingMax]);
			double h = rand.RandDouble(mn, mx);
			double curr_w = 0;
			double curr_delta_h = 0;
			int num_mini = rand.RandInt(0, mini_count_max + 1);
			for (int i = 0; i < num_mini + 1; ++i) {
				const double mini_w = (i == 0) ? w : 0.1;
				double mini_h = rand.RandDouble(curr_delta_h, h);
				mini_h = (i == num_mini) ? h : mini_h;
				double dh = mini_h - curr_delta_h;
				curr_w += AddStep(mini_w, dh, out);
				curr_delta_h = mini_h;
			}
			total_w += curr_w;
		}
		int end = static_cast<int>(out.size());
		OverlaySlopes(delta_range, p[pSlopeDeltaMin], p[pSlopeDeltaMax], 0, beg, end, rand, out);
		OverlayBumps(p[pBumpHeightMin], p[pBumpHeightMax], beg, end, rand, out);
		return total_w;
	}
	// sim/TerrainGen2D.cpp:148-181 GetTerrainFunc
	static double Build(int type, double width, const double* p, Rand& rand, std::vector<float>& out)
	{
		switch (type) {
		case eTypeGaps: return BuildGaps(width, p, rand, out);
		case eTypeSteps: return BuildSteps(width, p, rand, out);
		case eTypeWalls: return BuildWalls(width, p, rand, out);
		case eTypeBumps: return BuildBumps(width, p, rand, out);
		case eTypeMixed: return BuildMixed(width, p, rand, out);
		case eTypeNarrowGaps: return BuildNarrowGaps(width, p, rand, out);
		case eTypeSlopes: return WithSlopes(BuildFlat, width, p, rand, out);
		case eTypeSlopesGaps: return WithSlopes(BuildGaps, width, p, rand, out);
		case eTypeSlopesSteps: return WithSlopes(BuildSteps, width, p, rand, out);
		case eTypeSlopesWalls: return WithSlopes(BuildWalls, width, p, rand, out);
		case eTypeSlopesMixed: return WithSlopes(BuildMixed, width, p, rand, out);
		case eTypeSlopesNarrowGaps: return WithSlopes(BuildNarrowGaps, width, p, rand, out);
		case eTypeCliffs: return BuildCliffs(width, p, rand, out);
		default: return BuildFlat(width, p, rand, out);
		}
	}
};

// One heightfield segment: sim/GroundVar2D.cpp tSegment. Heights are kept UN-scaled (the reference stores
// data*world_scale as float and divides by the scale on every read; x4 and /4 are exact in binary fp).
struct Segment {
	std::vector<float> data;
	double min_x = 0;     // requested min x (double)
	double origin_x = 0;  // what cWorld::GetPos returns: float(scale * centre) / scale
	double scale_x = 0.1; // GetScaling()[0]: float(0.1f * scale) / scale
	bool Empty() const { return data.empty(); }
	int W() const { return static_cast<int>(data.size()); }
	void Init(double mn, double world_scale)
	{
		min_x = mn;
		double aabb_min = min_x;
		double aabb_max = min_x + (data.size() - 1) * static_cast<double>(TerrainGen::gVertSpacing);
		double origin = 0.5 * (aabb_min + aabb_max);
		float bt_origin = static_cast<float>(world_scale) * static_cast<float>(origin);  // cWorld::SetPos: scale * btVector3(float(pos))
		origin_x = static_cast<double>(bt_origin) / world_scale;                        // cWorld::GetPos
		double x_scale = static_cast<double>(TerrainGen::gVertSpacing) * world_scale;   // tSegment::Init x_scale
		scale_x = static_cast<double>(static_cast<float>(x_scale)) / world_scale;       // GetScaling()
	}
	// CalcAABB of the Bullet body is not reproducible without Bullet; GetMinX/GetMaxX are restated from the
	// construction values (min_x, min_x + (W-1)*0.1f) which is what the AABB spans up to Bullet's margin.
	double MinX() const { return Empty() ? std::numeric_limits<double>::infinity() : min_x; }
	double MaxX() const { return Empty() ? -std::numeric_limits<double>::infinity() : min_x + (data.size() - 1) * static_cast<double>(TerrainGen::gVertSpacing); }
	double StartH() const { return data[0]; }
	double EndH() const { return data[data.size() - 1]; }
	// sim/GroundVar2D.cpp:590-619 (x only; z coordinate is always 0 -> coord[1] = 1, inside [0, 2])
	double CalcGridCoord(double x) const
	{
		const double tol = 0.0001;
		int w = W();
		double c = x - origin_x;
		c /= scale_x;
		c += ((w - 1) * 0.5);
		if (c > -tol && c < w - 1 + tol) c = std::min(std::max(c, 0.0), w - 1.0);
		return c;
	}
	// sim/GroundVar2D.cpp:559-577
	double SampleHeight(double x, bool& valid, int& oi, int& oj) const
	{
		double c = CalcGridCoord(x);
		valid = !(c < 0 || c > W() - 1);
		c = std::min(std::max(c, 0.0), W() - 1.0);
		int i = static_cast<int>(c);
		int j = std::min(W() - 1, i + 1);
		double lerp = c - i;
		double a = data[i];
		double b = data[j];
		oi = i; oj = j;
		return (1 - lerp) * a + lerp * b;
	}
};

struct Ground {
	int type = eTypeFlat;
	double params[pMax];
	double world_scale = 1;
	double segment_width = 20;  // scenarios/ScenarioSimChar.cpp:351-352 (2 * char_view_dist)
	Rand rand;
	Segment segs[2];
	bool flip = false;
	long num_builds = 0;

	int SegID(int s) const { return flip ? (s == 0 ? 1 : 0) : s; }
	const Segment& Seg(int s) const { return segs[SegID(s)]; }
	double MinX() const { return Seg(0).MinX(); }
	double MidX() const { return Seg(0).MaxX(); }
	double MaxX() const { return Seg(1).MaxX(); }
	void Clear() { segs[0].data.clear(); segs[1].data.clear(); flip = false; }

	// sim/GroundVar2D.cpp:312-342 (+ AddPadding :344-355)
	void BuildSegment(int seg_id, double bound_min, double bound_max, bool align_min, double fix_y)
	{
		Segment& seg = segs[seg_id];
		seg.data.clear();
		bool contains_origin = (bound_min <= 0) && (bound_max >= 0);
		if (contains_origin) {
			double flat_w = std::min(bound_max - bound_min, 1 - bound_min);
			TerrainGen::BuildFlat(flat_w, params, rand, seg.data);
		}
		TerrainGen::Build(type, bound_max - bound_min, params, rand, seg.data);
		int num_verts = static_cast<int>(seg.data.size());
		float end_h = 0;
		if (num_verts > 0) end_h = align_min ? seg.data[0] : seg.data[num_verts - 1];
		float h_offset = static_cast<float>(fix_y - end_h);
		for (int i = 0; i < num_verts; ++i) seg.data[i] += h_offset;
		double new_bound_min = align_min ? bound_min : (bound_max - (num_verts - 1) * static_cast<double>(TerrainGen::gVertSpacing));
		seg.Init(new_bound_min, world_scale);
		++num_builds;
	}
	// sim/GroundVar2D.cpp:239-259
	void InitSegments(double bound_min_x, double bound_max_x)
	{
		Clear();
		double mid = 0.5 * (bound_max_x + bound_min_x);
		for (int i = 0; i < 2; ++i) {
			int seg_id = SegID(i);
			bool align_min = flip ? (i == 0) : (i != 0);  // GetSegAlignMode: seg 0 -> eAlignMax unless flipped
			double w = segment_width;
			double bmin = (!align_min) ? -w : 0;
			double bmax = (!align_min) ? 0 : w;
			BuildSegment(seg_id, bmin + mid, bmax + mid, align_min, 0.0);
		}
	}
	// sim/GroundVar2D.cpp:43-91
	void Update(double bound_min_x, double bound_max_x)
	{
		double min_x = MinX();
		double max_x = MaxX();
		if (bound_max_x < max_x && bound_min_x > min_x) {
		} else if (bound_max_x <= min_x || bound_min_x >= max_x) {
			InitSegments(bound_min_x, bound_max_x);
		} else {
			if (bound_max_x >= max_x) {
				int seg_id = SegID(0);
				double fix_y = Seg(1).EndH();
				bool was_unflipped = (SegID(0) == 0);
				BuildSegment(seg_id, max_x, max_x + segment_width, true, fix_y);
				flip = was_unflipped;
			} else {
				int seg_id = SegID(1);
				double fix_y = Seg(0).StartH();
				bool was_unflipped = (SegID(0) == 0);
				BuildSegment(seg_id, min_x - segment_width, min_x, false, fix_y);
				flip = was_unflipped;
			}
		}
	}
	// sim/GroundVar2D.cpp:98-114
	double SampleHeight(double x, bool* valid = nullptr, int* seg = nullptr, int* oi = nullptr, int* oj = nullptr) const
	{
		int seg_idx = 0;
		if (x >= Seg(0).MaxX()) seg_idx = 1;
		bool v; int i, j;
		double h = Seg(seg_idx).SampleHeight(x, v, i, j);
		if (valid) *valid = v;
		if (seg) *seg = seg_idx;
		if (oi) *oi = i;
		if (oj) *oj = j;
		return h;
	}
	// slope of the cell containing x (used by the documented contact model, not a reference function)
	double SampleSlope(double x) const
	{
		int seg_idx = 0;
		if (x >= Seg(0).MaxX()) seg_idx = 1;
		const Segment& s = Seg(seg_idx);
		bool v; int i, j;
		s.SampleHeight(x, v, i, j);
		if (j == i) return 0;
		return (static_cast<double>(s.data[j]) - static_cast<double>(s.data[i])) / (s.scale_x * (j - i));
	}
};

}  // namespace orc
