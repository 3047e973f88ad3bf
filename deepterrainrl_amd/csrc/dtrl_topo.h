// dtrl_topo.h -- skeleton topologies the register fast path is compiled for (dtrl_kernel_fast.h).
// The fast path eliminates the joint-space inertia matrix leaf-to-root; with the parent table known at compile time the updates that
// are structurally zero (DoF pairs that are not on one root->leaf path) are simply not emitted. A character whose parent table
// matches none of these runs on the generic LDS-phase kernel (same results, slower).
// Parent tables: data/characters/dog.txt and goat.txt (same skeleton), raptor.txt of the reference.
#pragma once
#include <cstdint>

namespace dtrl {

struct TopoDog {
	static constexpr int kId = 1;
	static constexpr int L = 21;
	// constraint rows the projected Gauss-Seidel keeps in registers (dtrl_kernel_fast.h pgs_solve_fast). A launch lasts as long as its slowest env, and the slowest
	// envs of a frame are characters lying on 13-24 rows per substep: with every row count on the unrolled path the dog's 2048-env launch went 4.71 -> 4.45 ms
	// (16.7 -> 17.8 M env-steps/s), with nested row sequences -> 4.33 ms (18.1 M), with 20 register rows + the tail rows from LDS -> 4.26 ms (18.4 M): same-box A/Bs in
	// profiles/r04_pgs_rows_ab.txt
#ifndef DTRL_PGS_ROWS_DOG
#define DTRL_PGS_ROWS_DOG 24   // (round 5, with the per-pass row steps: 17.22 M at 24 against 16.9 M at 20 + tail, 16.8 M at 16 + tail, 14.4 M at 20 + plain loop: profiles/r05_pgs_rows_ab.txt)
#endif
	static constexpr int kPgsRegRows = DTRL_PGS_ROWS_DOG;
#ifndef DTRL_PGS_TAIL_DOG
#define DTRL_PGS_TAIL_DOG 1
#endif
	static constexpr bool kPgsTailInSweep = DTRL_PGS_TAIL_DOG != 0;    // rows 20-23 inside the same unrolled sweep, their entries from LDS one update ahead (18.26 -> 18.42 M against 20 + plain loop)
	static constexpr int parent(int l) { constexpr int p[L] = {-1, 0, 1, 2, 3, 4, 5, 6, 7, 0, 9, 10, 11, 5, 13, 14, 15, 0, 17, 18, 19}; return p[l]; }
};
struct TopoRaptor {
	static constexpr int kId = 2;
	static constexpr int L = 19;
#ifndef DTRL_PGS_ROWS_RAPTOR
#define DTRL_PGS_ROWS_RAPTOR 16
#endif
#ifndef DTRL_PGS_TAIL_RAPTOR
#define DTRL_PGS_TAIL_RAPTOR 0
#endif
	static constexpr bool kPgsTailInSweep = DTRL_PGS_TAIL_RAPTOR != 0;   // (18.85 M with 16 + plain loop, 18.33 M with the tail rows in the sweep: spills)
	static constexpr int kPgsRegRows = DTRL_PGS_ROWS_RAPTOR;   // (nested row sequences, round 4: 18.65 M at 12, 18.88 M at 16, 18.82 M at 24 -- the raptor's instance pays for the registers of 24; flat sequences had 18.4 / 18.1 / 17.2 M at 12 / 18 / 24)
	static constexpr int parent(int l) { constexpr int p[L] = {-1, 0, 1, 2, 3, 4, 0, 6, 7, 8, 9, 0, 11, 12, 13, 0, 15, 16, 17}; return p[l]; }
};

// link a is l or an ancestor of l
template <class T>
constexpr bool link_on_path(int a, int l)
{
	while (l >= 0) { if (l == a) return true; l = T::parent(l); }
	return false;
}
// entry (j, k), j < k, of the joint-space inertia matrix is structurally non-zero during the leaf-first elimination. DoFs 0, 1 are the
// root translations: every hinge hangs below them (H(0,1) itself starts at zero but fills in as the hinges are eliminated);
// DoF d >= 2 is the hinge of link d - 2
template <class T>
constexpr bool dof_coupled(int j, int k)
{
	if (j < 2) return true;
	return link_on_path<T>(j - 2, k - 2);
}
template <class T>
inline bool topology_matches(const int32_t* parent, int L)
{
	if (L != T::L) return false;
	for (int l = 0; l < L; ++l) if (parent[l] != T::parent(l)) return false;
	return true;
}
// 0 = no compiled-in topology (generic kernel)
inline int match_topology(const int32_t* parent, int L)
{
	if (topology_matches<TopoDog>(parent, L)) return TopoDog::kId;
	if (topology_matches<TopoRaptor>(parent, L)) return TopoRaptor::kId;
	return 0;
}

}  // namespace dtrl
