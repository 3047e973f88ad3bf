// TEST program. The window logic that runs on the GPU in -terrain_gen= device mode (tg_build_segment / tg_init_segments / tg_window_update,
// dtrl_terrain_dev.h) is a template over the random source; here it is instantiated with TerrainRand, the libstdc++ stream the host mode uses (and
// which tests/test_reference_pin.py holds bit-exact against the reference's cTerrainGen2D), and walked side by side with GroundWindow:
// every record must be identical.
#include "dtrl_host.h"
#include "dtrl_terrain_dev.h"
#include <cstdio>
#include <cstring>
#include <random>
#include <string>

using namespace dtrl;

static bool same(const GroundRec& a, const GroundRec& b)
{
	for (int s = 0; s < 2; ++s) {
		if (a.w[s] != b.w[s] || a.min_x[s] != b.min_x[s] || a.max_x[s] != b.max_x[s] || a.origin_x[s] != b.origin_x[s] || a.scale_x[s] != b.scale_x[s]) return false;
		if (std::memcmp(a.data[s], b.data[s], sizeof(float) * a.w[s]) != 0) return false;
	}
	return true;
}

int main()
{
	long compared = 0, mismatches = 0, rebuilt = 0;
	for (int type = 0; type < kTerrTypeMax; ++type) {
		for (unsigned seed = 1; seed <= 6; ++seed) {
			double params[kNumTerrainParams];
			std::memcpy(params, kTerrainParamDefaults, sizeof(params));
			if (seed % 2 == 0) { params[0] = 2; params[1] = 3; params[37] = 0.1; }   // denser gaps, gentler slopes
			TerrainCfg c{};
			c.type = type; std::memcpy(c.params, params, sizeof(params)); c.world_scale = 4; c.segment_width = 20;
			c.view_min = -2; c.view_max = 11; c.spawn_min = -11; c.spawn_max = 9;
			GroundWindow host; host.Configure(type, params, 4.0, 20.0); host.SeedRand(seed);
			TerrainRand rnd; rnd.Seed(seed);
			GroundRec dev_rec{}, host_rec{};
			GroundGen gen{};
			host.InitSegments(c.spawn_min, c.spawn_max);
			tg_init_segments(dev_rec, c.spawn_min, c.spawn_max, c, rnd, &gen);
			std::string err;
			std::mt19937 walk(seed * 7919u + type);
			double x = 0;
			for (int step = 0; step < 400; ++step) {
				if (!host.FillRecord(host_rec, err)) { std::printf("capacity: %s\n", err.c_str()); return 2; }
				++compared;
				if (!same(host_rec, dev_rec)) { ++mismatches; if (mismatches < 5) std::printf("mismatch type %d seed %u step %d\n", type, seed, step); }
				const int r = static_cast<int>(walk() % 100);
				if (r < 70) x += 0.8; else if (r < 90) x -= 1.1; else if (r < 95) x += 60; else if (r < 97) x -= 75; else {
					// a fall: Clear + Update around the spawn point (the stream continues)
					host.Clear(); host.Update(c.spawn_min, c.spawn_max);
					tg_init_segments(dev_rec, c.spawn_min, c.spawn_max, c, rnd, &gen);
					x = 0; continue;
				}
				const bool a = host.Update(x + c.view_min, x + c.view_max);
				const bool b = tg_window_update(dev_rec, x + c.view_min, x + c.view_max, c, rnd, &gen);
				if (a != b) { ++mismatches; std::printf("update decision differs: type %d seed %u step %d\n", type, seed, step); }
				rebuilt += a;
			}
			if (gen.overflow) { std::printf("overflow\n"); return 2; }
		}
	}
	std::printf("%ld records compared, %ld window moves, %ld mismatches\n", compared, rebuilt, mismatches);
	return mismatches == 0 ? 0 : 1;
}
