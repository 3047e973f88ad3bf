// On-device terrain: the sliding two-segment window of cGroundVar2D (sim/GroundVar2D.cpp:43-91 Update, :312-355 BuildSegment, :392-455 tSegment::Init)
// and the strip generator of cTerrainGen2D (dtrl_terrain_gen.h) run by the GPU at the frame boundary, one thread per env, straight into the env's
// GroundRec -- no status read-back, no per-env host loop, no upload. Same generator code as the host path; the random source is a counter-based
// stream per env instead of a libstdc++ engine (a std::default_random_engine + std::uniform_*_distribution cannot be reproduced bit-for-bit on the
// device without shipping libstdc++'s algorithms), so windows are equal IN DISTRIBUTION to the host mode's, deterministic, and shard-invariant.
// The window logic is a template over the random source: tests/terrain_dev instantiates it with the HOST stream and checks it record-for-record
// against GroundWindow.
#pragma once
#include "dtrl_types.h"
#include "dtrl_terrain_gen.h"

namespace dtrl {

// vertex container over one GroundRec slot
struct SegBuf {
	float* d; int n, cap; int overflow;
	DTRL_TG_HD size_t size() const { return static_cast<size_t>(n); }
	DTRL_TG_HD bool empty() const { return n == 0; }
	DTRL_TG_HD float back() const { return d[n - 1]; }
	DTRL_TG_HD void push_back(float v) { if (n < cap) d[n++] = v; else overflow = 1; }
	DTRL_TG_HD float& operator[](size_t i) { return d[i]; }
};

DTRL_TG_HD inline uint64_t tg_mix(uint64_t x)
{
	x += 0x9E3779B97F4A7C15ULL;
	x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ULL;
	x = (x ^ (x >> 27)) * 0x94D049BB133111EBULL;
	return x ^ (x >> 31);
}
// the draws cTerrainGen2D makes (util/Rand.cpp: RandDouble / RandInt / FlipCoin / RandSign, degenerate ranges consume nothing), from a counter stream
struct CtrRand {
	uint64_t key; uint64_t* ctr;
	DTRL_TG_HD double u01() { const uint64_t z = tg_mix(key + (*ctr) * 0xD1342543DE82EF95ULL); ++(*ctr); return static_cast<double>(z >> 11) * (1.0 / 9007199254740992.0); }
	DTRL_TG_HD double RandDouble(double mn, double mx) { if (mn == mx) return mn; const double r = u01(); return mn + (r * (mx - mn)); }
	DTRL_TG_HD int RandInt(int mn, int mx) { if (mn == mx) return mn; const int r = mn + static_cast<int>(u01() * (mx - mn)); return r >= mx ? mx - 1 : r; }
	DTRL_TG_HD bool FlipCoin() { return RandDouble(0, 1) < 0.5; }
	DTRL_TG_HD int RandSign() { return FlipCoin() ? -1 : 1; }
};
DTRL_TG_HD inline uint64_t terrain_stream_key(uint64_t terrain_seed, int64_t global_env) { return tg_mix(tg_mix(terrain_seed) ^ (0x7E44A1ULL + static_cast<uint64_t>(global_env))); }

// cGroundVar2D::BuildSegment into logical slot `slot` of the record (0 = min segment, 1 = max segment)
template <class R>
DTRL_TG_HD inline void tg_build_segment(GroundRec& rec, int slot, double bmin, double bmax, bool align_min, double fix_y, const TerrainCfg& c, R& rnd, GroundGen* gen)
{
	SegBuf v{rec.data[slot], 0, kSegCap, 0};
	if (bmin <= 0 && bmax >= 0) { tgen::Strip<SegBuf> s(v); const double a = bmax - bmin, b = 1 - bmin; s.flat(a < b ? a : b); }   // flat padding around x = 0
	tgen::build_terrain(c.type, bmax - bmin, c.params, rnd, v);
	const int n = v.n;
	const float end_h = n > 0 ? (align_min ? v.d[0] : v.d[n - 1]) : 0.f;
	const float off = static_cast<float>(fix_y - end_h);
	for (int i = 0; i < n; ++i) v.d[i] += off;
	const double sp = static_cast<double>(tgen::kSpacing);
	const double min_x = align_min ? bmin : (bmax - (n - 1) * sp);
	const double max_x = min_x + (n - 1) * sp;
	const double centre = 0.5 * (min_x + max_x);
	const float bt_origin = static_cast<float>(c.world_scale) * static_cast<float>(centre);       // tSegment::Init: Bullet keeps float origins / scalings
	rec.origin_x[slot] = static_cast<double>(bt_origin) / c.world_scale;
	rec.scale_x[slot] = static_cast<double>(static_cast<float>(sp * c.world_scale)) / c.world_scale;
	rec.min_x[slot] = min_x; rec.max_x[slot] = max_x; rec.w[slot] = n;
	if (gen) { gen->builds += 1; gen->overflow += v.overflow; }
}
DTRL_TG_HD inline void tg_copy_slot(GroundRec& rec, int dst, int src)
{
	rec.origin_x[dst] = rec.origin_x[src]; rec.scale_x[dst] = rec.scale_x[src]; rec.min_x[dst] = rec.min_x[src]; rec.max_x[dst] = rec.max_x[src]; rec.w[dst] = rec.w[src];
	for (int i = 0; i < rec.w[src]; ++i) rec.data[dst][i] = rec.data[src][i];
}
// cGroundVar2D::InitSegments after Clear(): [mid - w, mid] ending at height 0, then [mid, mid + w] starting at height 0 (this draw order)
template <class R>
DTRL_TG_HD inline void tg_init_segments(GroundRec& rec, double bmin, double bmax, const TerrainCfg& c, R& rnd, GroundGen* gen)
{
	const double mid = 0.5 * (bmax + bmin), w = c.segment_width;
	tg_build_segment(rec, 0, -w + mid, mid, false, 0.0, c, rnd, gen);
	tg_build_segment(rec, 1, mid, w + mid, true, 0.0, c, rnd, gen);
}
// cGroundVar2D::Update: slide the window so that it covers [bmin, bmax]; the record stays in logical order (the reference flips a segment index)
template <class R>
DTRL_TG_HD inline bool tg_window_update(GroundRec& rec, double bmin, double bmax, const TerrainCfg& c, R& rnd, GroundGen* gen)
{
	const double min_x = rec.min_x[0], max_x = rec.max_x[1];
	if (bmax < max_x && bmin > min_x) return false;
	if (bmax <= min_x || bmin >= max_x) { tg_init_segments(rec, bmin, bmax, c, rnd, gen); return true; }
	if (bmax >= max_x) {
		const double fix_y = rec.data[1][rec.w[1] - 1];
		tg_copy_slot(rec, 0, 1);
		tg_build_segment(rec, 1, max_x, max_x + c.segment_width, true, fix_y, c, rnd, gen);
	} else {
		const double fix_y = rec.data[0][0];
		tg_copy_slot(rec, 1, 0);
		tg_build_segment(rec, 0, min_x - c.segment_width, min_x, false, fix_y, c, rnd, gen);
	}
	return true;
}

// the frame-boundary work of one env: what Engine::HostFrameWork does on the host in the default mode.
//   mode 0: after a frame -- a fallen env (status.need_reset) gets a fresh window around the spawn point (cScenarioSimChar::ResetGround: Clear + Update),
//           any other env has its window slid along with the character; a finished poli_eval episode (bit 1) is appended to the distance log
//   mode 1: (re)initialise unconditionally (Init, user resets)
DTRL_TG_HD inline void tg_env_boundary(GroundRec& rec, GroundGen& gen, const EnvStatus& st, const TerrainCfg& c, int mode, int env, DistRec* dist_ring, int32_t* dist_count, int32_t dist_cap)
{
	CtrRand rnd{gen.key, &gen.ctr};
	if (mode == 1) { tg_init_segments(rec, c.spawn_min, c.spawn_max, c, rnd, &gen); return; }
	if (st.need_reset & 2) {
		if (dist_ring) {
#if defined(__HIP_DEVICE_COMPILE__)
			const int32_t k = atomicAdd(dist_count, 1);
#else
			const int32_t k = (*dist_count)++;
#endif
			if (k < dist_cap) { dist_ring[k].env = env; dist_ring[k].dist = st.episode_dist; }
		}
	}
	if (st.need_reset) tg_init_segments(rec, c.spawn_min, c.spawn_max, c, rnd, &gen);
	else tg_window_update(rec, st.root_x + c.view_min, st.root_x + c.view_max, c, rnd, &gen);
}

}  // namespace dtrl
