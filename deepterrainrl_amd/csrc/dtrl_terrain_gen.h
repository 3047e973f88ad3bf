// Terrain strip generator, shared by the host path (libstdc++ random streams: bit-identical with the reference for a given seed) and the
// on-device path (counter-based streams, dtrl_terrain_dev.h). Templates over the random source R (RandDouble / RandInt / FlipCoin / RandSign) and the
// vertex container V (size / empty / back / push_back / operator[]). Arithmetic (float / double conversions, draw order) follows
// sim/TerrainGen2D.cpp:185-706 statement by statement, so R = TerrainRand, V = std::vector<float> reproduces the reference's profiles.
#pragma once
#include <cmath>
#include <cstddef>
#include <cstdint>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define DTRL_TG_HD __host__ __device__
#else
#define DTRL_TG_HD
#endif

namespace dtrl {
namespace tgen {

enum { GSmin, GSmax, GWmin, GWmax, GHmin, GHmax, WSmin, WSmax, WWmin, WWmax, WHmin, WHmax, SSmin, SSmax, SH0min, SH0max, SH1min, SH1max,
	BHmin, BHmax, NSmin, NSmax, NDmin, NDmax, NWmin, NWmax, NDpmin, NDpmax, NCmin, NCmax, CSmin, CSmax, CH0min, CH0max, CH1min, CH1max, CMini,
	SlRange, SlMin, SlMax };
enum { kTypeFlat, kTypeGaps, kTypeSteps, kTypeWalls, kTypeBumps, kTypeMixed, kTypeNarrowGaps, kTypeSlopes, kTypeSlopesGaps, kTypeSlopesSteps,
	kTypeSlopesWalls, kTypeSlopesMixed, kTypeSlopesNarrowGaps, kTypeCliffs, kTypeMax };

constexpr float kSpacing = 0.1f;  // cTerrainGen2D::gVertSpacing (float)

// a run of vertices appended to a height strip; knows whether the strip started empty (first vertex is shared otherwise)
template <class V>
struct Strip {
	V& h;
	DTRL_TG_HD explicit Strip(V& v) : h(v) {}
	DTRL_TG_HD static int verts_for(double w) { return static_cast<int>(ceil(w / kSpacing)) + 1; }
	// common prologue of AddFlat/AddBox/AddStep: hold the current height for `w` metres; returns (n0, was_empty, base)
	DTRL_TG_HD float hold(double w, size_t& n0, bool& empty)
	{
		int n = verts_for(w);
		n0 = h.size(); empty = h.empty();
		float base = 0;
		if (!empty) { --n; base = h.back(); }
		for (int i = 0; i < n; ++i) h.push_back(base);
		return base;
	}
	DTRL_TG_HD double added(size_t n0, bool empty) const
	{
		int verts = static_cast<int>(h.size() - n0);
		if (empty) --verts;
		return verts * kSpacing;   // int * float -> float, widened on return (as in the reference)
	}
	DTRL_TG_HD double flat(double w) { size_t n0; bool e; hold(w, n0, e); return added(n0, e); }
	DTRL_TG_HD double box(double spacing, double w, double depth)
	{
		size_t n0; bool e; float base = hold(spacing, n0, e);
		int n = verts_for(w) - 1;
		float lvl = static_cast<float>(base + depth);
		for (int i = 0; i < n; ++i) h.push_back(lvl);
		h.push_back(base);
		return added(n0, e);
	}
	DTRL_TG_HD double step(double w, double dh)
	{
		size_t n0; bool e; float base = hold(w, n0, e);
		h.push_back(static_cast<float>(base + dh));
		return added(n0, e);
	}
};

template <class R, class V>
DTRL_TG_HD inline void overlay_slopes(const double* p, size_t beg, size_t end, R& rnd, V& h)
{
	const double range = fabs(p[SlRange]), mean = 0.5 * (p[SlMin] + p[SlMax]), half = 0.5 * (p[SlMax] - p[SlMin]);
	double slope = 0, dh = 0;
	for (size_t i = beg; i < end; ++i) {
		double delta = rnd.RandDouble(0, range);
		double sign_rand = rnd.RandDouble(-1, 1);
		if (sign_rand < (slope - mean) / half) delta = -delta;
		slope += delta;
		dh += slope * kSpacing;
		h[i] += static_cast<float>(dh);
	}
}
template <class R, class V>
DTRL_TG_HD inline void overlay_bumps(double mn, double mx, size_t beg, size_t end, R& rnd, V& h)
{
	for (size_t i = beg; i + 1 < end; ++i) { double d = rnd.RandSign() * rnd.RandDouble(mn, mx); h[i] += static_cast<float>(d); }
}
template <class R>
DTRL_TG_HD inline void pick_range(double a0, double a1, double b0, double b1, R& rnd, double& mn, double& mx)
{
	bool va = (a0 != 0 || a1 != 0), vb = (b0 != 0 || b1 != 0);
	if (va && vb) { bool heads = rnd.FlipCoin(); mn = heads ? a0 : b0; mx = heads ? a1 : b1; }
	else if (va) { mn = a0; mx = a1; }
	else { mn = b0; mx = b1; }
}

template <class R, class V>
DTRL_TG_HD inline double gaps(double width, const double* p, R& rnd, V& h)
{
	Strip<V> s(h); double tot = 0;
	while (tot < width) { double sp = rnd.RandDouble(p[GSmin], p[GSmax]); double w = rnd.RandDouble(p[GWmin], p[GWmax]); double d = rnd.RandDouble(p[GHmin], p[GHmax]); tot += s.box(sp, w, d); }
	return tot;
}
template <class R, class V>
DTRL_TG_HD inline double walls(double width, const double* p, R& rnd, V& h)
{
	Strip<V> s(h); double tot = 0;
	while (tot < width) { double sp = rnd.RandDouble(p[WSmin], p[WSmax]); double w = rnd.RandDouble(p[WWmin], p[WWmax]); double d = rnd.RandDouble(p[WHmin], p[WHmax]); tot += s.box(sp, w, d); }
	return tot;
}
template <class R, class V>
DTRL_TG_HD inline double steps(double width, const double* p, R& rnd, V& h)
{
	Strip<V> s(h); double tot = 0;
	while (tot < width) {
		double mn, mx; pick_range(p[SH0min], p[SH0max], p[SH1min], p[SH1max], rnd, mn, mx);
		double w = rnd.RandDouble(p[SSmin], p[SSmax]); double dh = rnd.RandDouble(mn, mx);
		tot += s.step(w, dh);
	}
	return tot;
}
template <class R, class V>
DTRL_TG_HD inline double narrow_gaps(double width, const double* p, R& rnd, V& h)
{
	Strip<V> s(h); double tot = 0;
	int cmin = static_cast<int>(p[NCmin]), cmax = static_cast<int>(p[NCmax]);
	if (cmin < 1) cmin = 1;
	if (cmax < 1) cmax = 1;
	while (tot < width) {
		double sp = rnd.RandDouble(p[NSmin], p[NSmax]);
		int count = rnd.RandInt(cmin, cmax + 1);
		for (int i = 0; i < count; ++i) {
			double w = rnd.RandDouble(p[NWmin], p[NWmax]); double d = rnd.RandDouble(p[NDpmin], p[NDpmax]);
			tot += s.box(sp, w, d);
			sp = rnd.RandDouble(p[NDmin], p[NDmax]);
		}
	}
	return tot;
}
template <class R, class V>
DTRL_TG_HD inline double mixed(double width, const double* p, R& rnd, V& h)
{
	double tot = 0; const double dummy = kSpacing;
	while (tot < width) {
		int t = rnd.RandInt(0, 3);
		tot += (t == 0) ? gaps(dummy, p, rnd, h) : (t == 1) ? steps(dummy, p, rnd, h) : walls(dummy, p, rnd, h);
	}
	return tot;
}
template <class R, class V>
DTRL_TG_HD inline double cliffs(double width, const double* p, R& rnd, V& h)
{
	Strip<V> s(h); double tot = 0; size_t beg = h.size();
	int mini_max = static_cast<int>(p[CMini]);
	while (tot < width) {
		double mn, mx; pick_range(p[CH0min], p[CH0max], p[CH1min], p[CH1max], rnd, mn, mx);
		double w = rnd.RandDouble(p[CSmin], p[CSmax]); double dh = rnd.RandDouble(mn, mx);
		double cur_w = 0, cur_dh = 0;
		int n_mini = rnd.RandInt(0, mini_max + 1);
		for (int i = 0; i < n_mini + 1; ++i) {
			double mw = (i == 0) ? w : 0.1;
			double mh = rnd.RandDouble(cur_dh, dh);
			if (i == n_mini) mh = dh;
			cur_w += s.step(mw, mh - cur_dh);
			cur_dh = mh;
		}
		tot += cur_w;
	}
	size_t end = h.size();
	overlay_slopes(p, beg, end, rnd, h);
	overlay_bumps(p[BHmin], p[BHmax], beg, end, rnd, h);
	return tot;
}
// kind: 0 flat 1 gaps 2 steps 3 walls 4 mixed 5 narrow gaps
template <class R, class V>
DTRL_TG_HD inline double base_feature(int kind, double width, const double* p, R& rnd, V& h)
{
	switch (kind) {
	case 1: return gaps(width, p, rnd, h);
	case 2: return steps(width, p, rnd, h);
	case 3: return walls(width, p, rnd, h);
	case 4: return mixed(width, p, rnd, h);
	case 5: return narrow_gaps(width, p, rnd, h);
	default: { Strip<V> s(h); return s.flat(width); }
	}
}

// appends one terrain strip of (at least) `width` metres to `out`; returns the width added (cTerrainGen2D::tTerrainFunc).
// (base feature, slopes overlay, bumps overlay) per terrain type: sim/TerrainGen2D.cpp:148-181 + the Build* bodies
template <class R, class V>
DTRL_TG_HD inline double build_terrain(int type, double width, const double* p, R& rnd, V& out)
{
	if (type == kTypeCliffs) return cliffs(width, p, rnd, out);
	if (type < 0 || type >= kTypeMax) type = kTypeFlat;
	const int base = (type == kTypeGaps || type == kTypeSlopesGaps) ? 1 : (type == kTypeSteps || type == kTypeSlopesSteps) ? 2 : (type == kTypeWalls || type == kTypeSlopesWalls) ? 3
		: (type == kTypeMixed || type == kTypeSlopesMixed) ? 4 : (type == kTypeNarrowGaps || type == kTypeSlopesNarrowGaps) ? 5 : 0;
	const bool slopes = type >= kTypeSlopes && type <= kTypeSlopesNarrowGaps;
	const bool bumps = type == kTypeBumps;
	size_t beg = out.size();
	double tot = base_feature(base, width, p, rnd, out);
	size_t end = out.size();
	if (slopes) overlay_slopes(p, beg, end, rnd, out);
	if (bumps) overlay_bumps(p[BHmin], p[BHmax], beg, end, rnd, out);
	return tot;
}

}  // namespace tgen
}  // namespace dtrl
