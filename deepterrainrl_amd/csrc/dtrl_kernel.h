// dtrl_kernel.h -- the rollout hot path as wavefront-cooperative code: ONE 64-lane wavefront owns ONE environment.
//
// What runs here is the body of the loop at /root/reference/scenarios/ScenarioSimChar.cpp:162-173 for a batch of envs:
//   cWorld::Update            sim/World.cpp:96-105          -> substep(): planar articulated dynamics + contact (Integrator v1)
//   cContactManager::Update   sim/ContactManager.cpp:57-102 -> detect_contacts() link flags
//   cSimCharacter::Update     sim/SimCharacter.cpp:91-107   -> controller_update(): cDogController::Update
//     cRBDModel::Update / cRBDUtil::BuildMassMat / BuildBiasForce   sim/RBDModel.cpp:39-55, sim/RBDUtil.cpp:4-84,110-176
//     cDogController::UpdateState / ApplyFeedback                    sim/DogController.cpp:805-845, 903-945
//     cImpPDController::CalcControlForces                            sim/ImpPDController.cpp:234-278
//     ApplyGravityCompensation / ApplyVirtualForces                  sim/DogController.cpp:947-1029
//     cTerrainRLCharController::ParseGround / BuildPoliState         sim/TerrainRLCharController.cpp:168-285
//     cBaseControllerMACE::DecideActionBoltzmann + cNeuralNet::Eval  sim/BaseControllerMACE.cpp:254-318, learning/NeuralNet.cpp:352-375
//   cSimCharSoftFall fall checks sim/SimCharSoftFall.cpp:74-125, cScenarioExp::NewCycleUpdate scenarios/ScenarioExp.cpp:209-243
//
// MI355X mapping (not a translation of the reference's per-object C++):
//   * all characters are planar, so the reference's 6-D spatial algebra collapses to 3-D planar twists in
//     WORLD coordinates: H_ij = I_o - (p_i + p_j).mc + m p_i.p_j for ancestor pairs, RNEA = one path walk + one subtree sum;
//   * per-env working set (kinematics, H, constraint rows, Delassus matrix) is staged in LDS; lanes map to links / DoFs /
//     constraint rows / contact sample points; tree sweeps become path walks and subtree-mask sums with no level barriers;
//   * the contact solve is projected Gauss-Seidel in lambda space on the dense Delassus matrix so a row update is one
//     broadcast + one FMA per lane (see pgs_solve; the register/readlane form is pgs_solve_fast in dtrl_kernel_fast.h); since round 5 with Bullet's contact
//     persistence -- ground contact rows warm-started from the cache in EnvState (warm_match), a sweep = limit + normal rows, then the friction rows, a friction
//     row held while its normal row carries no impulse, rows for sample points within the manifold's breaking threshold (DevModel::warm_start, link_brk);
//   * one kernel launch advances a whole outer frame (20 env-steps); state touches HBM only at frame boundaries.
//
// The code is written once in "lane-phase" form: LANES_BEGIN/LANES_END delimit a phase executed by every lane, with a
// workgroup barrier at the end. Under hipcc a phase body runs once per thread; under g++ (tests only, see
// tests/emul/README) the same body runs in a for-loop over lanes so the math can be checked on a CPU-only box. The
// C-ABI never dispatches to the lane-loop build: the product path is the HIP kernel or an error.
#pragma once
#include "dtrl_types.h"
#include <cmath>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define DTRL_HD __host__ __device__
#define DTRL_HD_INLINE __host__ __device__ __forceinline__
#define DTRL_HD_NOINLINE __host__ __device__ __attribute__((noinline))
#else
#define DTRL_HD
#define DTRL_HD_INLINE inline
#define DTRL_HD_NOINLINE inline
#endif

#if defined(__HIP_DEVICE_COMPILE__)
#define LANES_BEGIN { const int lane = static_cast<int>(threadIdx.x);
#define LANES_END } ::dtrl::env_sync();
#else
#define LANES_BEGIN for (int lane = 0; lane < ::dtrl::kGroup; ++lane) {
#define LANES_END }
#endif

// optional per-section cycle accounting (device builds with -DDTRL_PROFILE only; s_memtime ticks of lane 0)
#if defined(__HIP_DEVICE_COMPILE__) && defined(DTRL_PROFILE)
#define PROF_T0() const unsigned long long prof_t0_ = __builtin_readcyclecounter()
#define PROF_ADD(ws, id) do { if (threadIdx.x == 0) (ws).prof[id] += __builtin_readcyclecounter() - prof_t0_; } while (0)
#define PROF_NOW() __builtin_readcyclecounter()
#define PROF_ADD_SINCE(ws, id, t0) do { if (threadIdx.x == 0) (ws).prof[id] += __builtin_readcyclecounter() - (t0); } while (0)
#define PROF_COUNT(ws, id) do { if (threadIdx.x == 0) (ws).prof[id] += 1; } while (0)
#else
#define PROF_T0() do {} while (0)
#define PROF_ADD(ws, id) do {} while (0)
#define PROF_NOW() 0ull
#define PROF_ADD_SINCE(ws, id, t0) do { (void)(t0); } while (0)
#define PROF_COUNT(ws, id) do {} while (0)
#endif

namespace dtrl {

#if defined(__HIP_DEVICE_COMPILE__)
// The synchronisation point between two lane phases of ONE env. An env is one 64-lane wavefront = one whole workgroup (dtrl_backend_hip.hip), so there is nobody
// to wait for: a wavefront's LDS (and memory) instructions are executed in program order, and what the phases need is only that the COMPILER keeps the accesses on
// their side of the point. __syncthreads() asks for more -- its workgroup-scope fences become `s_waitcnt vmcnt(0) lgkmcnt(0)` (every outstanding LDS and memory
// operation drained, a scheduling wall) although hipcc already drops the s_barrier of a one-wave workgroup -- so the env uses WAVEFRONT-scope fences around a wave
// barrier (AMDGPU memory model: no instruction required at wavefront scope); the s_waitcnt a value actually needs is placed by the compiler where it is used.
// -DDTRL_WAVE_SYNC=0 restores __syncthreads().
#ifndef DTRL_WAVE_SYNC
#define DTRL_WAVE_SYNC 1
#endif
#if DTRL_WAVE_SYNC && !defined(__GFX9__)
#error "env_sync(): wavefront-scope fences are only a barrier when the 64-thread workgroup is ONE wavefront (GFX9 / CDNA: wave64). On a wave32 target the workgroup is two waves: build with -DDTRL_WAVE_SYNC=0 (ADVICE r4)"
#endif
__device__ __forceinline__ void env_sync()
{
#if DTRL_WAVE_SYNC
	static_assert(kGroup == 64, "env_sync(): one wavefront per env");
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
	__builtin_amdgcn_wave_barrier();
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#else
	__syncthreads();
#endif
}
#else
inline void env_sync() {}      // (the lane-loop build runs a phase to its end before the next one starts)
#endif

enum ProfSection { kProfFK, kProfMass, kProfBias, kProfFact, kProfDetect, kProfRows, kProfFsub, kProfDelassus, kProfPgs, kProfFinish, kProfCtrl, kProfAction, kProfFrameIO, kProfTotal, kProfRowsSum, kProfSubsteps, kProfP1, kProfP2, kProfP3, kProfP4, kProfR0, kProfR1_6, kProfR7_12, kProfR13_18, kProfR19_24, kProfT0, kProfT1_6, kProfT7_12, kProfT13_18, kProfT19_24, kProfNNConv, kProfNNFcTerr, kProfNNRest, kProfNNEvals, kProfC_Fsm, kProfC_Feedback, kProfC_PdSetup, kProfC_PdSolve, kProfC_Grav, kProfC_Tail, kProfMax };

// hot, read-mostly model fields staged in LDS (per-substep readers only; the controller's once-per-env-step gains, torque
// limits and body angles stay in the HBM/L2-resident DevModel)
struct HotModel {
	int32_t L, D, char_type, n_pairs;
	int32_t warm_start;   // DevModel::warm_start (Bullet's contact persistence)
	int32_t parent[kMaxL], depth[kMaxL], col[kMaxL];
	int8_t pair_l[kMaxPairs], pair_k[kMaxPairs];
	alignas(4) int8_t path[kMaxL][kMaxDepth];   // read four entries at a time (path_word())
	uint32_t sub_mask[kMaxL];
	uint32_t anc_mask[kMaxL];   // bit a set <=> link a is an ancestor of j or j itself
	real attach[kMaxL][2], lim_lo[kMaxL], lim_hi[kMaxL];
	real body_attach[kMaxL][2];
	real mass[kMaxL], inertia[kMaxL], sub_mass[kMaxL];
	// link--link collision pairs (DevModel::cp_*): one lane per pair tests the two boxes for overlap every substep
	int32_t n_cpairs;
	int8_t cp_a[kMaxCP], cp_b[kMaxCP];
	float cp_half[kMaxL][2];
	float cp_root_bt[2];
};

struct DevBuffers {
	EnvState* st;
	GroundRec* gr;
	EnvStatus* status;
	real* poli_state;     // [N][S]
	real* tup_s0;         // [N][S]
	real* tup_a;          // [N][A]
	real* nn_out;         // [N][out_size]
	float* tuple_rows;    // [cap][W]  MACE replay row layout [r | s | a | s'] (learning/MACETrainer.cpp:373-401)
	uint32_t* tuple_flags;
	int32_t* tuple_env;
	int32_t* tuple_count; // device-wide atomic cursor
	int32_t tuple_cap;
	int32_t S, A, W;
	int32_t model_D;           // host-known DoF count
	int32_t model_topo;        // compiled-in skeleton id (dtrl_topo.h; selects the register-resident kernel instantiation), 0 = none
	unsigned long long* prof;  // [N][kProfMax] cycle counters (DTRL_PROFILE builds), else null
	const int32_t* env_list;   // optional indirection: workgroup b handles env_list[b] (launch order, compact reset launches)
	int32_t reset_listed;      // 1: every env of this launch performs the device half of a reset (compact reset launches);
	                           // 2: (-terrain_gen= device) a 0-step launch over a whole group in which the envs that fell (st.need_reset) do, the others return
	// on-device terrain generation (dtrl_terrain_dev.h); null / 0 in the default (host generator) mode
	GroundGen* gen;
	const TerrainCfg* tcfg;
	DistRec* dist_ring; int32_t* dist_count; int32_t dist_cap;
	// host terrain mode: regenerated terrain windows wait in page-locked HOST memory (gr_stage[slot]); stage_slot[env] = slot + 1 tells the env's own
	// wavefront to copy its record in at the start of its next launch (no upload, no scatter launch at the frame boundary); null otherwise
	const GroundRec* gr_stage; int32_t* stage_slot;
	const float* weights;
	const real* in_off; const real* in_scale; const real* out_off; const real* out_scale;
	NetDesc net;
};

// per-env LDS workspace. WSRef (reference / lane-loop path) keeps the joint-space inertia matrix, the square Delassus matrix
// and the contact sample points in LDS; the register-resident gfx950 path (WSFast) never materialises H or the sample
// points and stores the Delassus matrix packed, which is what lets 8 workgroups (2 waves per SIMD) share a CU's 160 KB.
constexpr int kZStride = kMaxD + 1;   // odd row stride: lane s walking row s of Z is LDS-bank-conflict free
struct WSBase {
	HotModel M;
	EnvState st;
	// kinematics (positions relative to the root joint origin; world x = st.q[0] + px)
	real phi[kMaxL], cs[kMaxL], sn[kMaxL], w[kMaxL];
	real px[kMaxL], py[kMaxL], cx[kMaxL], cy[kMaxL];
	real vpx[kMaxL], vpy[kMaxL], vcx[kMaxL], vcy[kMaxL];
	real fx[kMaxL], fy[kMaxL], fn[kMaxL];       // per-link inertial force / moment and first moments / inertia about the root origin:
	real mcx[kMaxL], mcy[kMaxL], Io[kMaxL];     // six contiguous arrays = the [6][kMaxL] right-hand matrix of the subtree-sum product
	real sfs[3][kMaxL];                         // subtree sums of fx, fy, fn
	// time-multiplexed: the bone vectors (kin_dyn_terms P2 -> P3) are dead when P4 writes the composite (subtree) sums
	union {
		struct { real bx[kMaxL], by[kMaxL], ux[kMaxL], uy[kMaxL], gx[kMaxL], gy[kMaxL]; };   // bone vectors, their velocity / centripetal terms
		struct { real sm[kMaxL], smx[kMaxL], smy[kMaxL], sI[kMaxL]; };                         // composite quantities about the root origin
	};
	real dinv[kMaxD];
	real b[kMaxD];
	real u[kMaxD];
	// constraint rows
	int32_t R, n_pts_active;   // n_pts_active (fast path): row count of the list built by the post-step contact pass for the next substep, -1 = none
	int32_t row_kind[kMaxRows], row_link[kMaxRows];
	int8_t row_link2[kMaxRows];   // link--link contact rows: the partner link the row pushes the other way (-1: the ground)
	real row_x[kMaxRows], row_y[kMaxRows], row_dx[kMaxRows], row_dy[kMaxRows], row_tgt[kMaxRows];
	uint16_t row_id[kMaxRows];    // identity of the row across substeps (EnvState::ws_id); the impulses live in st.ws_lam (the solve's lambda array AND the persistent cache)
	real wv[kMaxRows], rinv[kMaxRows];
	real dl;
	// time-multiplexed: the controller scratch is only live outside substep(), where Z rows >= 1 are unused
	union {
		real Z[kMaxRows + 1][kZStride];
		struct {
			real z0_[kZStride];
			real basis[kMaxD][4];
			real tau_g[kMaxD];
			real kpv[kMaxD], kdv[kMaxD], kdm[kMaxD], perr[kMaxD], verr[kMaxD];
		};
	};
	real red[8];
	int32_t flag_update_action, flag_new_cycle, flag_misc;
	int32_t cost;   // see EnvStatus::cost (the host launches the costliest envs first: longest-processing-time-first)
#if defined(DTRL_PROFILE)
	unsigned long long prof[kProfMax];
#endif
};
struct WSRef : WSBase {
	real H[kMaxD][kMaxD + 1];   // after factorisation: diag = d_k, upper H[k][i] = L_ik (i > k)
	// the contact sample points are dead once build_rows() has consumed them, which is before the Delassus matrix is written
	union {
		real A[kMaxRows][kMaxRows + 1];
		struct {
			real pt_x[kMaxPts], pt_y[kMaxPts], pt_depth[kMaxPts], pt_nx[kMaxPts], pt_ny[kMaxPts];
			int32_t pt_active[kMaxPts];   // bit 0: gets constraint rows, bit 1: near the surface (contact flag), bit 2: scratch (dropped by a rank filter)
		};
	};
};
constexpr int kPackedA = kMaxRows * (kMaxRows + 1) / 2;
constexpr int kMassTab = kMaxL * (kMaxDepth + 2);
struct WSFast : WSBase {
	// lower triangle of the Delassus matrix, row-major packed (entry (s, r), r <= s, at s (s + 1) / 2 + r: the triangular
	// numbers are distinct mod 32, so 24 lanes reading column r hit distinct banks); the same storage holds mass_row()'s
	// table before the rows are built
	real Apk[kPackedA > kMassTab ? kPackedA : kMassTab];
};

// ------------------------------------------------------------------------------------------------------------------
// small helpers

// fused multiply-add a*b + c, spelled out (the build runs with -ffp-contract=off): the SAME fused operations in the lane-loop,
// reference and register-resident builds keep the three bit-identical, and the solver's dependent chains are one op shorter per step
DTRL_HD_INLINE real fmadd(real a, real b, real c) { return __builtin_fma(a, b, c); }
DTRL_HD_INLINE void sincos_r(real x, real* s, real* c)
{
#if defined(DTRL_REAL_F32)
	sincosf(x, s, c);
#else
	sincos(x, s, c);
#endif
}
#if defined(__HIP_DEVICE_COMPILE__)
// one 16 x 16 x 4 step of the matrix pipe in the kernel's arithmetic type. Operands: A[i = lane % 16][k = lane / 16], B[k = lane / 16][n = lane % 16] in both types;
// the RESULT registers differ: register r of lane group g = lane / 16 holds row 4 r + g of the tile in fp64 (v_mfma_f64_16x16x4_f64) and row 4 g + r in fp32
// (v_mfma_f32_16x16x4_f32) -- mfma_row(). Both accumulate in k order with fused multiply-adds (tools/microbench/mfma_f64_check.hip).
typedef real v4r_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ v4r_t mfma_16x16x4(real a, real b, v4r_t acc)
{
#if defined(DTRL_REAL_F32)
	return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
#else
	return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
#endif
}
__device__ __forceinline__ constexpr int mfma_row(int r, int g)
{
#if defined(DTRL_REAL_F32)
	return 4 * g + r;
#else
	return 4 * r + g;
#endif
}
#endif
// 1/x and 1/sqrt(x) for the solver's pivots and the contact normals. An IEEE fp64 division on gfx950 is a 14-instruction VALU sequence
// (div_scale x2, rcp, Newton, div_fmas, div_fixup) and the frame kernel is bound by VALU issue; the hardware seed + two Newton steps
// (5 / 9 instructions) is within an ulp or two for the normal, positive arguments that occur here (inertias, 1 + slope^2).
// Both device kernels use these, so they stay bit-identical to each other; the CPU builds keep the exact division (tests compare
// against the oracle with a tolerance, not bitwise).
DTRL_HD_INLINE real fast_recip(real x)
{
#if defined(__HIP_DEVICE_COMPILE__)
	real r = __builtin_amdgcn_rcp(x);
	r = __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
	r = __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
	return r;
#else
	return 1.0 / x;
#endif
}
DTRL_HD_INLINE real fast_rsqrt(real x)
{
#if defined(__HIP_DEVICE_COMPILE__)
	real r = __builtin_amdgcn_rsq(x);
	const real hx = 0.5 * x;
	r = __builtin_fma(__builtin_fma(-hx * r, r, 0.5), r, r);   // r += r (1/2 - x r^2 / 2)
	r = __builtin_fma(__builtin_fma(-hx * r, r, 0.5), r, r);
	return r;
#else
	return 1.0 / sqrt(x);
#endif
}

// angular rates are clamped to a quarter turn per substep (kMaxTurnPerSubstep): 4712 rad/s at 1/3000 s -- only a whipping tail of a crashed character gets
// there; unclamped, the explicit Coriolis terms then overflow within a few substeps
DTRL_HD_INLINE real clamp_turn_rate(real v, int dof, real h)
{
	if (dof < 2) return v;
	const real vmax = kMaxTurnPerSubstep / h;
	return v > vmax ? vmax : (v < -vmax ? -vmax : v);
}
DTRL_HD inline real wrap_pi(real a)
{
	const real pi = 3.14159265358979323846, two_pi = 6.283185307179586476925286766559;
	real r = fmod(a + pi, two_pi);
	if (r < 0) r += two_pi;
	return r - pi;
}

// counter-based per-env RNG (same stream definition as oracle/or_ctrl.h EnvRng; the reference's global RNG is racy and
// time-seeded, SURVEY Appendix B.10, so only distributions are preserved)
DTRL_HD inline uint64_t rng_mix(uint64_t x)
{
	x += 0x9E3779B97F4A7C15ULL;
	x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ULL;
	x = (x ^ (x >> 27)) * 0x94D049BB133111EBULL;
	return x ^ (x >> 31);
}
struct Rng {
	uint64_t key; uint64_t* ctr;
	DTRL_HD real uniform() { uint64_t z = rng_mix(key + (*ctr) * 0xD1342543DE82EF95ULL); ++(*ctr); return static_cast<real>(z >> 11) * (1.0 / 9007199254740992.0); }
	DTRL_HD real uniform(real mn, real mx) { return mn + uniform() * (mx - mn); }
	DTRL_HD int rand_int(int mn, int mx) { if (mn == mx) return mn; int r = mn + static_cast<int>(uniform() * (mx - mn)); return r >= mx ? mx - 1 : r; }
	DTRL_HD bool flip() { return uniform() < 0.5; }
	DTRL_HD real normal(real mean, real stdev)
	{
		real u1 = 1.0 - uniform(), u2 = uniform();
		return mean + stdev * sqrt(-2.0 * log(u1)) * cos(6.283185307179586476925286766559 * u2);
	}
};
DTRL_HD inline Rng make_rng(const RunParams& rp, int env, uint64_t* ctr)
{
	Rng r; r.key = rng_mix(rp.rng_seed ^ rng_mix(static_cast<uint64_t>(rp.env_id_base + env))); r.ctr = ctr; return r;
}

// ------------------------------------------------------------------------------------------------------------------
// heightfield sampling: /root/reference/sim/GroundVar2D.cpp:98-114 (segment pick) + :559-619 (grid coord, clamp, lerp).
// Grid coordinates are computed in double with the float-rounded Bullet origin / scaling the host stored in GroundRec,
// so the cell indices i, j are bit-exact with the reference's arithmetic.
struct GroundHdr { real mx0, o0, o1, sc0, sc1; int w0, w1; };
// both segment headers, fetched up front (independent of the sample position, so the loads travel with the caller's other loads)
// and selected afterwards: loads indexed by the segment would be a second dependent round trip to memory in front of the heights
DTRL_HD_INLINE GroundHdr ground_header(const GroundRec& g)
{
	GroundHdr h;
	h.mx0 = g.max_x[0]; h.w0 = g.w[0]; h.w1 = g.w[1];
	h.o0 = g.origin_x[0]; h.o1 = g.origin_x[1]; h.sc0 = g.scale_x[0]; h.sc1 = g.scale_x[1];
	return h;
}
DTRL_HD_INLINE real sample_ground(const GroundRec& g, const GroundHdr& gh, real x, real* slope, int* oi, int* oj, int* oseg)
{
	const real mx0 = gh.mx0;
	const int w0 = gh.w0, w1 = gh.w1;
	const real o0 = gh.o0, o1 = gh.o1, sc0 = gh.sc0, sc1 = gh.sc1;
	const int seg = (x >= mx0) ? 1 : 0;
	const int w = seg ? w1 : w0;
	const real scale = seg ? sc1 : sc0;
	const real tol = 0.0001;
	real c = x - (seg ? o1 : o0);
	c /= scale;
	c += ((w - 1) * 0.5);
	if (c > -tol && c < w - 1 + tol) { c = c < 0.0 ? 0.0 : (c > w - 1.0 ? w - 1.0 : c); }
	c = !(c > 0.0) ? 0.0 : (c > w - 1.0 ? w - 1.0 : c);   // (written so that a NaN coordinate -- a state that has already blown up -- still indexes inside the segment)
	int i = static_cast<int>(c);
	int j = (i + 1 < w - 1) ? i + 1 : w - 1;
	real lerp = c - i;
	const real inv_run = fast_recip(scale * ((j - i) > 0 ? (j - i) : 1));   // ready before the height samples arrive
	real a = g.data[seg][i];
	real b = g.data[seg][j];
	if (slope) *slope = (j == i) ? 0.0 : (b - a) * inv_run;
	if (oi) *oi = i;
	if (oj) *oj = j;
	if (oseg) *oseg = seg;
	return (1 - lerp) * a + lerp * b;
}
DTRL_HD_INLINE real sample_ground(const GroundRec& g, real x, real* slope, int* oi, int* oj, int* oseg)
{
	const GroundHdr gh = ground_header(g);
	return sample_ground(g, gh, x, slope, oi, oj, oseg);
}

// ------------------------------------------------------------------------------------------------------------------
// kinematics + composite inertias + generalised bias force in four lane phases.
//   P1  phi_j, w_j = sums of q, qd along the root->j path; cos/sin(phi_j)
//   P2  per link: world "bone" vector r_j = R(phi_parent) attach_j and its velocity / centripetal contributions
//   P3  per link: joint position, velocity, path part of the acceleration = sums of the P2 quantities along the path
//       (one LDS read per quantity and path element, fixed-trip predicated loops so the loads pipeline); COM, inertial force
//   P4  per link: subtree sums (mass, first moments, inertia about the root origin, force, moment) with one masked loop
// then b_d (planar RNEA in world coordinates) falls out without further loops.
// The bias computed here is the textbook one (what the integrator needs). The reference's implicit-PD controller sees the bias of
// cRBDUtil::BuildCjPlanar as shipped (sim/RBDUtil.cpp:809-836: theta read from q_dot and s = cos(theta)), which differs from the
// textbook bias by a uniform extra base acceleration (dax, day); with subtree sums that is a closed-form correction per DoF
// (quirk_bias() below), so ONE evaluation at the post-step configuration serves the controller AND the next substep.
// four consecutive entries of link j's root->j path in one 32-bit LDS read. The path loops below fetch the whole index list of a lane
// up front (three independent loads) instead of one dependent byte load per path element in front of every data load
static_assert(kMaxDepth % 4 == 0, "path rows are read as 32-bit words");
template <class W>
DTRL_HD_INLINE uint32_t path_word(const W& ws, int j, int t)
{
#if defined(__HIP_DEVICE_COMPILE__)
	return reinterpret_cast<const uint32_t*>(&ws.M.path[j][0])[t];
#else
	uint32_t w; __builtin_memcpy(&w, &ws.M.path[j][4 * t], 4); return w;
#endif
}
template <class W>
DTRL_HD inline void kin_dyn_terms(W& ws)
{
	PROF_T0();
	LANES_BEGIN
	if (lane < ws.M.L) {
		const int j = lane;
		const int dep = ws.M.depth[j];
		real phi = 0, w = 0;
		const uint32_t pw[kMaxDepth / 4] = {path_word(ws, j, 0), path_word(ws, j, 1), path_word(ws, j, 2)};
#pragma unroll
		for (int k = 0; k < kMaxDepth; ++k) {
			const int a = (k <= dep) ? static_cast<int>((pw[k / 4] >> (8 * (k % 4))) & 0xffu) : 0;
			const real qa = ws.st.q[a + 2], wa = ws.st.qd[a + 2];
			const real on = (k <= dep) ? 1.0 : 0.0;   // predicated accumulation as fma(x, 1|0, acc): same sum (x*1 is exact), but branch-free, so the LDS loads of the whole loop pipeline
			phi = fmadd(qa, on, phi); w = fmadd(wa, on, w);
		}
		ws.phi[j] = phi; ws.w[j] = w;
		real s, c; sincos_r(phi, &s, &c);
		ws.cs[j] = c; ws.sn[j] = s;
	}
	LANES_END
	PROF_ADD(ws, kProfP1);
	LANES_BEGIN
	if (lane < ws.M.L) {
		const int j = lane;
		const int pa = ws.M.parent[j];
		real rx = 0, ry = 0, ux = 0, uy = 0, gx = 0, gy = 0;
		if (pa >= 0) {
			const real c = ws.cs[pa], s = ws.sn[pa], wp = ws.w[pa];
			rx = c * ws.M.attach[j][0] - s * ws.M.attach[j][1];
			ry = s * ws.M.attach[j][0] + c * ws.M.attach[j][1];
			ux = -(wp * ry); uy = wp * rx;
			const real w2 = wp * wp;
			gx = -(w2 * rx); gy = -(w2 * ry);
		}
		ws.bx[j] = rx; ws.by[j] = ry; ws.ux[j] = ux; ws.uy[j] = uy; ws.gx[j] = gx; ws.gy[j] = gy;
	}
	LANES_END
	PROF_ADD(ws, kProfP2);
	LANES_BEGIN
	if (lane < ws.M.L) {
		const int j = lane;
		const int dep = ws.M.depth[j];
		const real ax0 = 0, ay0 = -kGravityY;
		real px = 0, py = 0, vx = ws.st.qd[0], vy = ws.st.qd[1], ax = 0, ay = 0;
		const uint32_t pw[kMaxDepth / 4] = {path_word(ws, j, 0), path_word(ws, j, 1), path_word(ws, j, 2)};
#pragma unroll
		for (int k = 1; k < kMaxDepth; ++k) {
			const int a = (k <= dep) ? static_cast<int>((pw[k / 4] >> (8 * (k % 4))) & 0xffu) : 0;
			const real rx = ws.bx[a], ry = ws.by[a], dux = ws.ux[a], duy = ws.uy[a], dgx = ws.gx[a], dgy = ws.gy[a];
			const real on = (k <= dep) ? 1.0 : 0.0;
			px = fmadd(rx, on, px); py = fmadd(ry, on, py); vx = fmadd(dux, on, vx); vy = fmadd(duy, on, vy); ax = fmadd(dgx, on, ax); ay = fmadd(dgy, on, ay);
		}
		ws.px[j] = px; ws.py[j] = py; ws.vpx[j] = vx; ws.vpy[j] = vy;
		const real c = ws.cs[j], s = ws.sn[j], wj = ws.w[j];
		const real rx = c * ws.M.body_attach[j][0] - s * ws.M.body_attach[j][1];
		const real ry = s * ws.M.body_attach[j][0] + c * ws.M.body_attach[j][1];
		const real cx = px + rx, cy = py + ry;
		ws.cx[j] = cx; ws.cy[j] = cy;
		ws.vcx[j] = vx - wj * ry; ws.vcy[j] = vy + wj * rx;
		const real w2 = wj * wj, m = ws.M.mass[j];
		const real fx = m * ((ax0 + ax) - w2 * rx), fy = m * ((ay0 + ay) - w2 * ry);
		ws.fx[j] = fx; ws.fy[j] = fy; ws.fn[j] = cx * fy - cy * fx;
		ws.mcx[j] = m * cx; ws.mcy[j] = m * cy; ws.Io[j] = ws.M.inertia[j] + m * (cx * cx + cy * cy);
	}
	LANES_END
	PROF_ADD(ws, kProfP3);
	// P4: subtree sums S[j][q] = sum_k sub(j, k) V[k][q] of the six per-link quantities. This is a (0/1 matrix) x (matrix) product, and
	// on the device it runs on the fp64 matrix pipe: two 16-row tiles x six k-steps of v_mfma_f64_16x16x4_f64 (12 instructions and
	// 6 LDS reads per lane instead of 126 FMAs and 126 LDS reads). The MFMA accumulates in k order with fused multiply-adds, i.e. exactly
	// the sequence of the lane loop below (verified bit for bit by tools/microbench/mfma_f64_check.hip), so the CPU builds agree.
	// Operand layout (same check): A[i = lane % 16][k = lane / 16], B[k = lane / 16][n = lane % 16], D register r = D[4 r + lane / 16][lane % 16].
#if defined(__HIP_DEVICE_COMPILE__)
	{
		typedef v4r_t v4d_t;
		const int l = static_cast<int>(threadIdx.x), g = l >> 4, c = l & 15;
		const int nL = ws.M.L;
		const real* V = &ws.fx[0];                                              // [6][kMaxL]: fx, fy, fn, mcx, mcy, Io
		const uint32_t m0 = (c < nL) ? ws.M.sub_mask[c < nL ? c : 0] : 0u;      // rows c and 16 + c of the mask matrix
		const uint32_t m1 = (16 + c < nL) ? ws.M.sub_mask[16 + c < nL ? 16 + c : 0] : 0u;
		v4d_t acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
#pragma unroll
		for (int s4 = 0; s4 < kMaxL / 4; ++s4) {
			const int k = 4 * s4 + g;
			const bool live = c < 6 && k < nL;                                   // never multiply an unwritten LDS slot, not even by zero
			const real bv = V[live ? c * kMaxL + k : 0];
			const real b = live ? bv : 0.0;
			const real a0 = ((m0 >> k) & 1u) ? 1.0 : 0.0, a1 = ((m1 >> k) & 1u) ? 1.0 : 0.0;
			acc0 = mfma_16x16x4(a0, b, acc0);
			acc1 = mfma_16x16x4(a1, b, acc1);
		}
		if (c < 6) {
			real* out = (c < 3) ? &ws.sfs[c][0] : (&ws.smx[0] + (c - 3) * kMaxL);   // smx, smy, sI are contiguous
#pragma unroll
			for (int r = 0; r < 4; ++r) {
				const int j0 = mfma_row(r, g), j1 = 16 + mfma_row(r, g);
				if (j0 < nL) out[j0] = acc0[r];
				if (j1 < nL) out[j1] = acc1[r];
			}
		}
	}
	env_sync();
#else
	LANES_BEGIN
	if (lane < ws.M.L) {
		const int j = lane;
		const uint32_t mask = ws.M.sub_mask[j];
		real mx = 0, my = 0, I = 0, sfx = 0, sfy = 0, sfn = 0;
		const int nL = ws.M.L;
		for (int k = 0; k < nL; ++k) {
			const real on = ((mask >> k) & 1u) ? 1.0 : 0.0;
			sfx = fmadd(ws.fx[k], on, sfx); sfy = fmadd(ws.fy[k], on, sfy); sfn = fmadd(ws.fn[k], on, sfn);
			mx = fmadd(ws.mcx[k], on, mx); my = fmadd(ws.mcy[k], on, my); I = fmadd(ws.Io[k], on, I);
		}
		ws.sfs[0][j] = sfx; ws.sfs[1][j] = sfy; ws.sfs[2][j] = sfn;
		ws.smx[j] = mx; ws.smy[j] = my; ws.sI[j] = I;
	}
	LANES_END
#endif
	LANES_BEGIN
	if (lane < ws.M.L) {
		const int j = lane;
		ws.sm[j] = ws.M.sub_mass[j];   // subtree mass is a model constant (summed on the host in the same order)
		const real sfx = ws.sfs[0][j], sfy = ws.sfs[1][j], sfn = ws.sfs[2][j];
		// generalised bias: translations see the total force (root subtree = everything), hinge l the subtree moment about p_l
		const real bl = sfn - ws.px[j] * sfy + ws.py[j] * sfx;
		ws.b[j + 2] = bl;
		if (j == 0) { ws.b[0] = sfx; ws.b[1] = sfy; }
	}
	LANES_END
	PROF_ADD(ws, kProfP4);
}
template <class W>
DTRL_HD inline void forward_kinematics(W& ws) { kin_dyn_terms(ws); }

// bias of DoF i as the reference's controller sees it: textbook bias + the BuildCjPlanar discrepancy as a base acceleration
// (dax, day) acting on the subtree: translations M (dax, day); hinge l: day (smx_l - sm_l px_l) - dax (smy_l - sm_l py_l)
template <class W>
DTRL_HD inline real quirk_bias(const W& ws, int i)
{
	const real vx0 = ws.st.qd[0], vy0 = ws.st.qd[1], om = ws.st.qd[2];
	const real c = ws.cs[0], s = ws.sn[0];
	const real cq = cos(om);
	const real tx = (-s * vx0 + c * vy0) * om, ty = (-c * vx0 - s * vy0) * om;     // textbook cj (body frame)
	const real qx = (-cq * vx0 + cq * vy0) * om, qy = (-cq * vx0 - cq * vy0) * om; // shipped cj
	const real dx = qx - tx, dy = qy - ty;
	const real dax = c * dx - s * dy, day = s * dx + c * dy;                        // back to world frame
	if (i == 0) return ws.b[0] + ws.sm[0] * dax;
	if (i == 1) return ws.b[1] + ws.sm[0] * day;
	const int l = i - 2;
	return ws.b[i] + (day * (ws.smx[l] - ws.sm[l] * ws.px[l]) - dax * (ws.smy[l] - ws.sm[l] * ws.py[l]));
}

// joint-space inertia matrix in closed form from the composite quantities (LDS copy, reference path)
template <class W>
DTRL_HD inline void mass_matrix(W& ws)
{
	LANES_BEGIN
	for (int e = lane; e < kMaxD * (kMaxD + 1); e += kGroup) (&ws.H[0][0])[e] = 0;
	LANES_END
	LANES_BEGIN
	const int D = ws.M.D;
	if (lane < D) {
		const int d = lane;
		if (d < 2) { ws.H[d][d] = ws.sm[0]; }
		else {
			const int l = d - 2;
			real m = ws.sm[l], mx = ws.smx[l], my = ws.smy[l], I = ws.sI[l];
			real plx = ws.px[l], ply = ws.py[l];
			real hx = -(my - m * ply), hy = (mx - m * plx);
			ws.H[d][0] = hx; ws.H[0][d] = hx; ws.H[d][1] = hy; ws.H[1][d] = hy;
			for (int k = 0; k <= ws.M.depth[l]; ++k) {
				int a = ws.M.path[l][k], da = a + 2;
				real pax = ws.px[a], pay = ws.py[a];
				real v = I - ((plx + pax) * mx + (ply + pay) * my) + m * (plx * pax + ply * pay);
				ws.H[d][da] = v; ws.H[da][d] = v;
			}
		}
	}
	LANES_END
}

// in-place H = U D U^T of ws.H (U unit upper triangular; lane i owns row i), eliminating the LAST DoF first. Joints are numbered
// parents-first, so this is the leaf-to-root order of Featherstone's LTDL: a pivot only couples to its ancestors and the
// factorisation creates no fill-in -- U(i,k) != 0 only if DoF i is an ancestor of DoF k. This path runs the dense loops (the skipped
// updates are exact no-ops: fma(-0, x, h) = h); the register fast path compiles the sparsity of the shipped skeletons in.
// Afterwards: diag = d_k, H[k][i] (i < k) = U_ik (stored transposed, in the lower triangle), dinv = 1/d.
template <class W>
DTRL_HD inline void factorize(W& ws)
{
	const int D = ws.M.D;
	for (int k = D - 1; k >= 1; --k) {
		LANES_BEGIN
		if (lane < k) {
			const int i = lane;
			real lik = ws.H[i][k] * fast_recip(ws.H[k][k]);
			for (int j = k - 1; j >= i; --j) ws.H[i][j] = fmadd(-lik, ws.H[j][k], ws.H[i][j]);
			ws.H[k][i] = lik;
		}
		LANES_END
	}
	LANES_BEGIN
	if (lane < D) ws.dinv[lane] = fast_recip(ws.H[lane][lane]);
	LANES_END
}

// J_r[i] for a point row (link, x, y rel. root, direction d): translation DoFs see d, hinge a on the path sees d . z x (pt - p_a)
template <class W>
DTRL_HD inline real row_jac(const W& ws, int r, int i)
{
	if (ws.row_kind[r] == 0) return (i == ws.row_link[r] + 2) ? ws.row_dx[r] : 0.0;
	const int l2 = ws.row_link2[r];
	if (i == 0) return l2 < 0 ? ws.row_dx[r] : 0.0;   // a link--link row is the difference of two point rows: the root translation drops out
	if (i == 1) return l2 < 0 ? ws.row_dy[r] : 0.0;
	const int a = i - 2;
	const int on = static_cast<int>((ws.M.sub_mask[a] >> ws.row_link[r]) & 1u) - (l2 < 0 ? 0 : static_cast<int>((ws.M.sub_mask[a] >> l2) & 1u));
	if (on == 0) return 0.0;
	const real jv = ws.row_dx[r] * (-(ws.row_y[r] - ws.py[a])) + ws.row_dy[r] * (ws.row_x[r] - ws.px[a]);
	return on > 0 ? jv : -jv;
}
// velocity of the material point of link l that currently sits at (x, y), along (dx, dy)
template <class W>
DTRL_HD_INLINE real link_point_vel(const W& ws, int l, real x, real y, real dx, real dy)
{
	const real vx = ws.vpx[l] - ws.w[l] * (y - ws.py[l]);
	const real vy = ws.vpy[l] + ws.w[l] * (x - ws.px[l]);
	return dx * vx + dy * vy;
}
// J v of a point row at the current velocities (ground rows: the link's point; link--link rows: relative velocity of the two links' points)
template <class W>
DTRL_HD_INLINE real row_point_jv(const W& ws, int s)
{
	real jv = link_point_vel(ws, ws.row_link[s], ws.row_x[s], ws.row_y[s], ws.row_dx[s], ws.row_dy[s]);
	const int l2 = ws.row_link2[s];
	if (l2 >= 0) jv -= link_point_vel(ws, l2, ws.row_x[s], ws.row_y[s], ws.row_dx[s], ws.row_dy[s]);
	return jv;
}

// ---- link--link contacts: sample point k of link P against the box of link Q (both kernels and the lane-loop build share this) ----
struct PairHit { real x, y, depth, nx, ny; int active; };   // normal: pushes P out of Q
template <class W>
DTRL_HD_INLINE PairHit pair_point_eval(const W& ws, const DevModel& gm, int P, int Q, int k)
{
	PairHit r; r.x = 0; r.y = 0; r.depth = 0; r.nx = 0; r.ny = 0; r.active = 0;
	const real lx = gm.pt_joint[P][k][0], ly = gm.pt_joint[P][k][1];
	const real x = ws.px[P] + ws.cs[P] * lx - ws.sn[P] * ly;
	const real y = ws.py[P] + ws.sn[P] * lx + ws.cs[P] * ly;
	const real cq = ws.cs[Q] * gm.bt_cs[Q] - ws.sn[Q] * gm.bt_sn[Q], sq = ws.sn[Q] * gm.bt_cs[Q] + ws.cs[Q] * gm.bt_sn[Q];   // box frame of Q
	const real dx = x - ws.cx[Q], dy = y - ws.cy[Q];
	const real qx = cq * dx + sq * dy, qy = -sq * dx + cq * dy;
	const real pen_x = gm.body_half[Q][0] - fabs(qx), pen_y = gm.body_half[Q][1] - fabs(qy);
	if (!(pen_x > 0 && pen_y > 0)) return r;
	real nlx = 0, nly = 0;
	if (pen_x <= pen_y) { nlx = qx >= 0 ? 1.0 : -1.0; r.depth = pen_x; } else { nly = qy >= 0 ? 1.0 : -1.0; r.depth = pen_y; }
	r.nx = cq * nlx - sq * nly; r.ny = sq * nlx + cq * nly;
	r.x = x; r.y = y; r.active = 1;
	return r;
}
// can pair pr touch at all? Separating-axis test of the two (slightly grown) boxes: second neighbours along a limb are always within each other's
// bounding circles, so only an oriented test keeps the out-of-line contact code off the common path. Conservative by construction (cp_half)
template <class W>
DTRL_HD_INLINE bool pair_in_reach(const W& ws, int pr)
{
	const int a = ws.M.cp_a[pr], b = ws.M.cp_b[pr];
	const real tx = ws.cx[b] - ws.cx[a], ty = ws.cy[b] - ws.cy[a];
	real ca = ws.cs[a], sa = ws.sn[a];
	const real cb = ws.cs[b], sb = ws.sn[b];
	if (a == 0) { const real bc = ws.M.cp_root_bt[0], bs = ws.M.cp_root_bt[1]; const real c0 = ca * bc - sa * bs; sa = sa * bc + ca * bs; ca = c0; }   // (a < b: only a can be the root; float cos / sin: the grown extents absorb 1e-7)
	const real hxa = ws.M.cp_half[a][0], hya = ws.M.cp_half[a][1], hxb = ws.M.cp_half[b][0], hyb = ws.M.cp_half[b][1];
	const real c = fabs(ca * cb + sa * sb), s = fabs(ca * sb - sa * cb);
	const bool sep = fabs(tx * ca + ty * sa) > hxa + hxb * c + hyb * s || fabs(-tx * sa + ty * ca) > hya + hxb * s + hyb * c
		|| fabs(tx * cb + ty * sb) > hxb + hxa * c + hya * s || fabs(-tx * sb + ty * cb) > hyb + hxa * s + hya * c;
	return !sep;
}
// row identities across substeps (EnvState::ws_id; the same numbers as oracle/or_sim.h): ground contact 2 x sample point + (0 normal, 1 tangent), link--link contact
// 512 + 2 x (pair x 12 + candidate) + (0, 1), limit rows kNoRowId
constexpr uint16_t kNoRowId = 0xffffu;
constexpr int kFirstPairRowId = 512;
DTRL_HD_INLINE uint16_t ground_row_id(int pt, int t) { return static_cast<uint16_t>(2 * pt + t); }
DTRL_HD_INLINE uint16_t pair_row_id(int pr, int cand, int t) { return static_cast<uint16_t>(kFirstPairRowId + 2 * (pr * 2 * kPtsPerLink + cand) + t); }
static_assert(2 * kMaxPts <= kFirstPairRowId && kFirstPairRowId + 2 * kMaxCP * 2 * kPtsPerLink < 0xffff, "row id ranges");
// candidate c (0..11) of a pair: a's six sample points against b, then b's six against a
DTRL_HD_INLINE void pair_candidate(int a, int b, int c, int* P, int* Q, int* k) { const int side = c >= kPtsPerLink ? 1 : 0; *P = side ? b : a; *Q = side ? a : b; *k = c - side * kPtsPerLink; }
// append the link--link contact rows behind the ground rows (serial form): per pair the deepest kMaxPtsPerPair penetrating candidates (ties: the earlier
// candidate), in candidate order, while rows are left
template <class W>
DTRL_HD inline int append_pair_rows_serial(W& ws, const DevModel& gm, int R)
{
	for (int pr = 0; pr < ws.M.n_cpairs && R + 2 <= kMaxRows; ++pr) {
		if (!pair_in_reach(ws, pr)) continue;
		const int a = ws.M.cp_a[pr], b = ws.M.cp_b[pr];
		PairHit hit[2 * kPtsPerLink];
		for (int c = 0; c < 2 * kPtsPerLink; ++c) { int P, Q, k; pair_candidate(a, b, c, &P, &Q, &k); hit[c] = pair_point_eval(ws, gm, P, Q, k); }
		for (int c = 0; c < 2 * kPtsPerLink && R + 2 <= kMaxRows; ++c) {
			if (!hit[c].active) continue;
			int rank = 0;
			for (int o = 0; o < 2 * kPtsPerLink; ++o) if (o != c && hit[o].active && (hit[o].depth > hit[c].depth || (hit[o].depth == hit[c].depth && o < c))) ++rank;
			if (rank >= kMaxPtsPerPair) continue;
			int P, Q, k; pair_candidate(a, b, c, &P, &Q, &k);
			// (velocity-level non-penetration only, no recovery term: a contact point can sit millimetres from the only hinge axis that could separate
			// the two links, where 0.2 depth / h asks for thousands of rad/s; Bullet recovers penetration by split impulse, momentum-free as well)
			ws.row_kind[R] = 1; ws.row_link[R] = P; ws.row_link2[R] = static_cast<int8_t>(Q); ws.row_x[R] = hit[c].x; ws.row_y[R] = hit[c].y;
			ws.row_dx[R] = hit[c].nx; ws.row_dy[R] = hit[c].ny; ws.row_tgt[R] = 0; ws.row_id[R] = pair_row_id(pr, c, 0); ++R;
			ws.row_kind[R] = 2; ws.row_link[R] = P; ws.row_link2[R] = static_cast<int8_t>(Q); ws.row_x[R] = hit[c].x; ws.row_y[R] = hit[c].y;
			ws.row_dx[R] = hit[c].ny; ws.row_dy[R] = -hit[c].nx; ws.row_tgt[R] = 0; ws.row_id[R] = pair_row_id(pr, c, 1); ++R;
		}
	}
	return R;
}

// world position of contact sample point pt (relative to the root origin) + ground test; shared by both kernel paths
struct PtVal { real x, y, depth, nx, ny; int active; int near; };   // active: penetrating or within the breaking threshold above the surface (gets constraint rows; depth < 0 for the latter); near: within contact_tol of the surface (sets the link's contact flag)
template <bool kNear = true, class W>
DTRL_HD_INLINE PtVal contact_point_eval(const W& ws, const DevModel& gm, const GroundRec& g, int pt)
{
	PtVal r; r.x = 0; r.y = 0; r.depth = 0; r.nx = 0; r.ny = 0; r.active = 0; r.near = 0;
	const int j = pt / kPtsPerLink, k = pt - j * kPtsPerLink;
	const GroundHdr gh = ground_header(g);   // issued with the sample point's local coordinates below: one trip to memory for both
	if (ws.M.col[j] == 0) return r;
	const real lx = gm.pt_ground[j][k][0], ly = gm.pt_ground[j][k][1];   // point of the margin-shrunk box
	const real c = ws.cs[j], s = ws.sn[j];
	const real x = ws.px[j] + c * lx - s * ly;
	const real y = ws.py[j] + s * lx + c * ly;
	real slope;
	const real h = sample_ground(g, gh, ws.st.q[0] + x, &slope, nullptr, nullptr, nullptr);
	const real gap = h - (ws.st.q[1] + y);
	// inside a substep only the points that carry rows matter: penetrating ones and -- Bullet's persistent manifold -- those within the breaking threshold above the surface
	const real margin = gm.link_margin[j], brk = gm.link_brk[j];
	// (rows: depth = gap * ny + margin > -brk needs gap > -(margin + brk) / ny, and 1 / ny = sqrt(1 + slope^2) <= 1 + |slope|)
	if (!kNear && !(gap + (margin + brk) * (1.0 + fabs(slope)) > 0)) return r;
	const real inv = fast_rsqrt(1.0 + slope * slope);
	const real depth = fmadd(gap, inv, margin);   // along the cell normal (ny = inv > 0), to the ROUNDED surface of the box (Bullet's collision margin)
	if (kNear) r.near = depth >= -gm.contact_tol ? 1 : 0;   // cContactManager::Update: getDistance() <= dist_tol
	if (!(depth > -brk)) return r;
	r.nx = -slope * inv; r.ny = inv;
	r.depth = depth;
	r.x = x; r.y = y;
	r.active = 1;
	return r;
}
template <class W>
DTRL_HD inline int sample_contact_point(W& ws, const DevModel& gm, const GroundRec& g, int pt)
{
	const PtVal v = contact_point_eval(ws, gm, g, pt);
	if (v.active) { ws.pt_x[pt] = v.x; ws.pt_y[pt] = v.y; ws.pt_depth[pt] = v.depth; ws.pt_nx[pt] = v.nx; ws.pt_ny[pt] = v.ny; }
	ws.pt_active[pt] = v.active | (v.near << 1);
	return v.active;
}
// contact sample points of every colliding link against the env's heightfield (2 points per lane)
template <class W>
DTRL_HD inline void detect_contacts(W& ws, const DevModel& gm, const GroundRec& g)
{
	LANES_BEGIN
	for (int pt = lane; pt < ws.M.L * kPtsPerLink; pt += kGroup) sample_contact_point(ws, gm, g, pt);
	LANES_END
	LANES_BEGIN
	if (lane == 0) {
		uint32_t bits = 0;
		for (int j = 0; j < ws.M.L; ++j) { int any = 0; for (int k = 0; k < kPtsPerLink; ++k) any |= ws.pt_active[j * kPtsPerLink + k] & 2; if (any) bits |= (1u << j); }
		ws.st.contact_bits = bits;   // (link--link contacts never set a flag: the character's parts are registered with filter eContactFlagEnvironment, scenarios/ScenarioSimChar.cpp:321)
	}
	LANES_END
	// at most kMaxPtsPerLink constraint-carrying points per link: the deepest ones (ties: lower sample-point index). Two phases: every lane ranks
	// its points against the unfiltered flags, then the flags are rewritten
	LANES_BEGIN
	for (int pt = lane; pt < ws.M.L * kPtsPerLink; pt += kGroup) {
		int drop = 0;
		if (ws.pt_active[pt] & 1) {
			const int base = (pt / kPtsPerLink) * kPtsPerLink;
			const real d = ws.pt_depth[pt];
			int rank = 0;
			for (int k = 0; k < kPtsPerLink; ++k) { const int o = base + k; if (o != pt && (ws.pt_active[o] & 1)) { const real od = ws.pt_depth[o]; rank += (od > d || (od == d && o < pt)) ? 1 : 0; } }
			drop = rank >= kMaxPtsPerLink;
		}
		if (drop) ws.pt_active[pt] |= 4;
	}
	LANES_END
	LANES_BEGIN
	for (int pt = lane; pt < ws.M.L * kPtsPerLink; pt += kGroup) { const int f = ws.pt_active[pt]; ws.pt_active[pt] = (f & 4) ? (f & 2) : (f & 3); }
	LANES_END
}

// target normal velocity of a ground contact row: Baumgarte recovery of the penetration beyond the slop, capped; a point still above the surface (depth < 0, kept by the
// manifold's breaking threshold) may approach by its distance per substep (Bullet: `velocityError -= penetration / dt` for positive distance)
DTRL_HD_INLINE real normal_row_target(real depth, real inv_h)
{
	const real t = kErp * fmax(depth - kSlop, 0.0) * inv_h;
	return depth < 0 ? depth * inv_h : fmin(t, kVDepenMax);
}
// build the ordered row list: violated joint limits first (by joint id), then normal+tangent per active contact point
template <class W>
DTRL_HD inline void build_rows(W& ws, const DevModel& gm, real h)
{
	LANES_BEGIN
	if (lane == 0) {
		const real inv_h = 1.0 / h;   // one division per call instead of one per row (same value in both kernels)
		int R = 0;
		for (int j = 1; j < ws.M.L; ++j) {
			if (ws.M.lim_lo[j] > ws.M.lim_hi[j]) continue;
			real th = ws.st.q[j + 2];
			if (th <= ws.M.lim_lo[j] + kLimitSlop && R < kMaxRows) { ws.row_kind[R] = 0; ws.row_link[R] = j; ws.row_link2[R] = -1; ws.row_dx[R] = 1; ws.row_tgt[R] = kLimitErp * fmax(ws.M.lim_lo[j] - th, 0.0) * inv_h; ws.row_id[R] = kNoRowId; ++R; }
			else if (th >= ws.M.lim_hi[j] - kLimitSlop && R < kMaxRows) { ws.row_kind[R] = 0; ws.row_link[R] = j; ws.row_link2[R] = -1; ws.row_dx[R] = -1; ws.row_tgt[R] = kLimitErp * fmax(th - ws.M.lim_hi[j], 0.0) * inv_h; ws.row_id[R] = kNoRowId; ++R; }
		}
		int cap = (kMaxRows - R) / 2, nc = 0;
		// more penetrating points than rows: the deepest `cap` points overall get rows (ties: lower sample-point index)
		int n_act = 0;
		for (int pt = 0; pt < ws.M.L * kPtsPerLink; ++pt) n_act += ws.pt_active[pt] & 1;
		if (n_act > cap) {
			for (int pt = 0; pt < ws.M.L * kPtsPerLink; ++pt) if (ws.pt_active[pt] & 1) {
				const real d = ws.pt_depth[pt];
				int rank = 0;
				for (int o = 0; o < ws.M.L * kPtsPerLink; ++o) if (o != pt && (ws.pt_active[o] & 1)) { const real od = ws.pt_depth[o]; rank += (od > d || (od == d && o < pt)) ? 1 : 0; }
				if (rank >= cap) ws.pt_active[pt] |= 4;
			}
			for (int pt = 0; pt < ws.M.L * kPtsPerLink; ++pt) if (ws.pt_active[pt] & 4) ws.pt_active[pt] &= 2;
		}
		for (int pt = 0; pt < ws.M.L * kPtsPerLink && nc < cap; ++pt) if (ws.pt_active[pt] & 1) {
			const int j = pt / kPtsPerLink;
			ws.row_kind[R] = 1; ws.row_link[R] = j; ws.row_link2[R] = -1; ws.row_x[R] = ws.pt_x[pt]; ws.row_y[R] = ws.pt_y[pt];
			ws.row_dx[R] = ws.pt_nx[pt]; ws.row_dy[R] = ws.pt_ny[pt]; ws.row_tgt[R] = normal_row_target(ws.pt_depth[pt], inv_h); ws.row_id[R] = ground_row_id(pt, 0); ++R;
			ws.row_kind[R] = 2; ws.row_link[R] = j; ws.row_link2[R] = -1; ws.row_x[R] = ws.pt_x[pt]; ws.row_y[R] = ws.pt_y[pt];
			ws.row_dx[R] = ws.pt_ny[pt]; ws.row_dy[R] = -ws.pt_nx[pt]; ws.row_tgt[R] = 0; ws.row_id[R] = ground_row_id(pt, 1); ++R;
			++nc;
		}
		R = append_pair_rows_serial(ws, gm, R);   // link--link contacts take what is left of the row budget
		ws.R = R;
	}
	LANES_END
}

// Z_r = U^-1 J_r^T for every row (lane r) and z_0 = U^-1 rhs (lane R): substitution from the last DoF up, per lane
template <class W>
DTRL_HD inline void forward_subst_rows(W& ws, const real* rhs)
{
	const int D = ws.M.D, R = ws.R;
	LANES_BEGIN
	if (lane <= R) {
		const int r = lane;
		real* z = ws.Z[r];
		for (int i = D - 1; i >= 0; --i) {
			real s = (r < R) ? row_jac(ws, r, i) : rhs[i];
			for (int k = D - 1; k > i; --k) s = fmadd(-ws.H[k][i], z[k], s);
			z[i] = s;
		}
	}
	LANES_END
}

// Delassus matrix A = Z D^-1 Z^T (lane s owns row s), initial w = J v_free - target
template <class W>
DTRL_HD inline void build_delassus(W& ws, real h)
{
	const int D = ws.M.D, R = ws.R;
	LANES_BEGIN
	if (lane < R) {
		const int s = lane;
		for (int r = 0; r <= s; ++r) {
			real a = 0;
#pragma unroll 13
			for (int i = 0; i < kMaxD; ++i) { const real zs = ws.Z[s][i], zr = ws.Z[r][i], di = ws.dinv[i]; if (i < D) a = fmadd(zs * di, zr, a); }   // (Z_s D^-1) Z_r^T: the left factor is scaled first, as the matrix-pipe form of the fast path does
			ws.A[s][r] = a;
		}
		real jv;
		if (ws.row_kind[s] == 0) jv = ws.row_dx[s] * ws.st.qd[ws.row_link[s] + 2];
		else jv = row_point_jv(ws, s);
		real zz = 0;
#pragma unroll 13
		for (int i = 0; i < kMaxD; ++i) { const real zs = ws.Z[s][i], di = ws.dinv[i], z0 = ws.Z[R][i]; if (i < D) zz = fmadd(zs * di, z0, zz); }
		ws.wv[s] = jv + h * zz - ws.row_tgt[s];
		ws.rinv[s] = (ws.A[s][s] >= 1e-12) ? 1.0 / ws.A[s][s] : 0.0;   // rows with a vanishing effective mass are skipped
	}
	LANES_END
	LANES_BEGIN
	if (lane < R) { const int s = lane; for (int r = s + 1; r < R; ++r) ws.A[s][r] = ws.A[r][s]; }
	LANES_END
}

// Bullet's persistent contact points: every row of the fresh list looks its identity up among the rows of the last solved substep and starts from kWarmFactor x the
// impulse that row ended with (limit rows and new contacts from zero); the list then becomes the cache. Runs once per substep, after the rows exist (built in this
// substep or taken over from the post-step contact pass), in both kernels
template <class W>
DTRL_HD inline void warm_match(W& ws)
{
	const int R = ws.R;
	LANES_BEGIN
	if (lane < R) {
		real l0 = 0.0;
		const int id = ws.row_id[lane];
		if (ws.M.warm_start != 0 && id < kFirstPairRowId) {   // ground contact rows only: limit rows and link--link rows start from zero (DevModel::warm_start)
			int hit = -1;
			for (int p = 0; p < ws.st.ws_R; ++p) hit = (ws.st.ws_id[p] == id) ? p : hit;
			if (hit >= 0) l0 = kWarmFactor * ws.st.ws_lam[hit];
		}
		ws.wv[lane] = l0;   // (wv is free until the Delassus build)
	}
	LANES_END
	LANES_BEGIN
	if (lane < R) { ws.st.ws_lam[lane] = ws.wv[lane]; ws.st.ws_id[lane] = ws.row_id[lane]; }
	if (lane == 0) ws.st.ws_R = R;
	LANES_END
}

// projected Gauss-Seidel in lambda space, kPgsIters sweeps. LDS form (reference for the register/readlane form used by the tuned kernel).
// With Bullet's contact persistence (DevModel::warm_start): the rows start from the impulses warm_match() left in st.ws_lam (w += A lambda_0 first), a sweep takes the
// limit and normal rows in list order, then the friction rows, and a friction row is resolved only while its normal row (the row before it) carries an impulse.
template <class W>
DTRL_HD inline void pgs_solve(W& ws)
{
	const int R = ws.R;
	const bool warm = ws.M.warm_start != 0;
	for (int r = 0; r < R; ++r) {
		LANES_BEGIN
		if (lane == 0) { const real l0 = (ws.rinv[r] != 0.0) ? ws.st.ws_lam[r] : 0.0; ws.st.ws_lam[r] = l0; ws.dl = l0; }   // rows with a vanishing effective mass stay at zero
		LANES_END
		LANES_BEGIN
		if (lane < R) ws.wv[lane] = fmadd(ws.A[lane][r], ws.dl, ws.wv[lane]);
		LANES_END
	}
	for (int it = 0; it < kPgsIters; ++it) {
		for (int pass = 0; pass < (warm ? 2 : 1); ++pass) {
			for (int r = 0; r < R; ++r) {
				if (warm && ((pass == 0) == (ws.row_kind[r] == 2))) continue;
				LANES_BEGIN
				if (lane == 0) {
					real ri = ws.rinv[r], dl = 0;
					const bool hold = warm && ws.row_kind[r] == 2 && !(ws.st.ws_lam[r - 1] > kHoldEps);
					if (ri != 0.0 && !hold) {
						real nl = fmadd(-ws.wv[r], ri, ws.st.ws_lam[r]);
						if (ws.row_kind[r] == 2) { real lim = kMu * ws.st.ws_lam[r - 1]; nl = fmin(fmax(nl, -lim), lim); }
						else nl = fmax(nl, 0.0);
						dl = nl - ws.st.ws_lam[r];
						ws.st.ws_lam[r] = nl;
					}
					ws.dl = dl;
				}
				LANES_END
				LANES_BEGIN
				if (lane < R) ws.wv[lane] = fmadd(ws.A[lane][r], ws.dl, ws.wv[lane]);
				LANES_END
			}
		}
	}
}

// v+ = qd + U^-T D^-1 (h z_0 + sum_r Z_r lambda_r); q+ = q + h v+
template <class W>
DTRL_HD inline void finish_substep(W& ws, real h)
{
	const int D = ws.M.D, R = ws.R;
	LANES_BEGIN
	if (lane < D) {
		const int i = lane;
		real s = h * ws.Z[R][i];
		for (int r = 0; r < R; ++r) s = fmadd(ws.Z[r][i], ws.st.ws_lam[r], s);
		ws.u[i] = s * ws.dinv[i];
	}
	LANES_END
	for (int k = 0; k < D - 1; ++k) {   // U^-T: x_i -= U_ki x_k for i > k (U_ki sits at H[i][k])
		LANES_BEGIN
		if (lane > k && lane < D) ws.u[lane] = fmadd(-ws.H[lane][k], ws.u[k], ws.u[lane]);
		LANES_END
	}
	LANES_BEGIN
	if (lane < D) { const int i = lane; real v = clamp_turn_rate(ws.st.qd[i] + ws.u[i], i, h); ws.st.qd[i] = v; ws.st.q[i] += h * v; }
	LANES_END
}

// generalised force of the env's active perturbation on DoF d (tPerturb::ApplyForce -> cWorld::ApplyForce, sim/World.cpp:445-470: Bullet
// accumulates the force at the body's COM plus the torque rel_pos x force evaluated when it is applied, once per cWorld::Update)
template <class W>
DTRL_HD_INLINE real perturb_gen_force(const W& ws, int d)
{
	const int l = ws.st.pert_link;
	if (d == 0) return ws.st.pert_f[0];
	if (d == 1) return ws.st.pert_f[1];
	const int a = d - 2;
	if (!((ws.M.sub_mask[a] >> l) & 1u)) return 0.0;
	return (ws.cx[l] - ws.px[a]) * ws.st.pert_f[1] - (ws.cy[l] - ws.py[a]) * ws.st.pert_f[0] + ws.st.pert_torque;
}
// start of cWorld::Update: cPerturbManager::UpdatePerturbs (sim/PerturbManager.cpp:41-56) -- expired perturbations are dropped, the others
// advance their clock and are applied for this env-step. Needs the kinematics of the current configuration (cs, sn)
template <class W>
DTRL_HD inline void perturb_begin_step(W& ws, real dt)
{
	LANES_BEGIN
	if (lane == 0 && ws.st.pert_link >= 0) {
		if (ws.st.pert_time >= ws.st.pert_dur) { ws.st.pert_link = -1; ws.st.pert_on = 0; }
		else {
			ws.st.pert_time += dt; ws.st.pert_on = 1;
			const int l = ws.st.pert_link;
			const real rx = ws.cs[l] * ws.st.pert_lp[0] - ws.sn[l] * ws.st.pert_lp[1], ry = ws.sn[l] * ws.st.pert_lp[0] + ws.cs[l] * ws.st.pert_lp[1];
			ws.st.pert_torque = rx * ws.st.pert_f[1] - ry * ws.st.pert_f[0];
		}
	}
	LANES_END
}

// one physics substep (stand-in for one Bullet internal step of sim/World.cpp:101-102)
template <class W>
DTRL_HD inline void substep_ref(W& ws, const DevModel& gm, const GroundRec& g, real h, bool kin_valid)
{
	if (!kin_valid) { PROF_T0(); kin_dyn_terms(ws); PROF_ADD(ws, kProfFK); }
	{ PROF_T0(); mass_matrix(ws); PROF_ADD(ws, kProfMass); }
	{ PROF_T0(); factorize(ws); PROF_ADD(ws, kProfFact); }
	{ PROF_T0(); detect_contacts(ws, gm, g); PROF_ADD(ws, kProfDetect); }
	{ PROF_T0(); build_rows(ws, gm, h); warm_match(ws);
	LANES_BEGIN
	if (lane < ws.M.D) { ws.u[lane] = ws.st.tau[lane] - ws.b[lane]; if (__builtin_expect(ws.st.pert_on != 0, 0)) ws.u[lane] += perturb_gen_force(ws, lane); }
	if (lane == 0) ws.cost += 8 + ws.R;
	LANES_END
	PROF_ADD(ws, kProfRows); }
	{ PROF_T0(); forward_subst_rows(ws, ws.u); PROF_ADD(ws, kProfFsub); }
	if (ws.R > 0) {
		{ PROF_T0(); build_delassus(ws, h); PROF_ADD(ws, kProfDelassus); }
		{ PROF_T0(); pgs_solve(ws); PROF_ADD(ws, kProfPgs); }
	}
	{ PROF_T0(); finish_substep(ws, h); PROF_ADD(ws, kProfFinish); }
#if defined(__HIP_DEVICE_COMPILE__) && defined(DTRL_PROFILE)
	if (threadIdx.x == 0) { ws.prof[kProfRowsSum] += ws.R; ws.prof[kProfSubsteps] += 1; }
#endif
}

// ------------------------------------------------------------------------------------------------------------------
// controller (cDogController family)

enum { jRoot, jSpine0, jSpine1, jSpine2, jSpine3, jTorso, jNeck0, jNeck1, jHead, jTail0, jTail1, jTail2, jTail3,
	jShoulder, jElbow, jWrist, jFinger, jHip, jKnee, jAnkle, jToe };
enum { spSpineCurve, spShoulder, spElbow, spHip, spKnee, spAnkle, spMax };
enum { mpTransTime, mpCv, mpBackForceX, mpBackForceY, mpFrontForceX, mpFrontForceY, mpMax };
enum { stBackStance, stExtend, stFrontStance, stGather, stMax, stInvalid };

template <class W>
DTRL_HD inline bool in_contact(const W& ws, int j) { return (ws.st.contact_bits >> j) & 1u; }

template <class W>
DTRL_HD inline void calc_com(const W& ws, real* out)
{
	real sx = 0, sy = 0, m = 0;
	for (int j = 0; j < ws.M.L; ++j) { sx += ws.M.mass[j] * (ws.st.q[0] + ws.cx[j]); sy += ws.M.mass[j] * (ws.st.q[1] + ws.cy[j]); m += ws.M.mass[j]; }
	out[0] = sx / m; out[1] = sy / m;
}
// raptor joint ids (sim/SimRaptor.h:11-33), params (5 misc + 4 states x 8) and FSM states (sim/RaptorController.h)
enum { rRoot, rSpine0, rSpine1, rSpine2, rSpine3, rHead, rTail0, rTail1, rTail2, rTail3, rTail4,
	rRightHip, rRightKnee, rRightAnkle, rRightToe, rLeftHip, rLeftKnee, rLeftAnkle, rLeftToe };
enum { rspRootPitch, rspSpineCurve, rspStanceHip, rspStanceKnee, rspStanceAnkle, rspSwingHip, rspSwingKnee, rspSwingAnkle, rspMax };
enum { rmpTransTime, rmpCv, rmpCd, rmpForceX, rmpForceY, rmpMax };
enum { rstContact, rstDown, rstPassing, rstUp, rstMax, rstInvalid };
template <class W>
DTRL_HD inline int stance_joint(const W& ws, int k) { return (ws.st.stance == 0 ? rRightHip : rLeftHip) + k; }   // k: 0 hip .. 3 toe
template <class W>
DTRL_HD inline int swing_joint(const W& ws, int k) { return (ws.st.stance == 0 ? rLeftHip : rRightHip) + k; }

template <class W>
DTRL_HD inline void set_state_params(W& ws)  // sim/DogController.cpp:1042-1054, sim/RaptorController.cpp:1108-1126
{
	if (ws.M.char_type == 1) {
		const real* sp = ws.st.params + rmpMax + ws.st.state * rspMax;
		const int sh = stance_joint(ws, 0), wh = swing_joint(ws, 0);
		ws.st.pd_target[sh] = sp[rspStanceHip]; ws.st.pd_target[sh + 1] = sp[rspStanceKnee]; ws.st.pd_target[sh + 2] = sp[rspStanceAnkle];
		ws.st.pd_target[wh] = sp[rspSwingHip]; ws.st.pd_target[wh + 1] = sp[rspSwingKnee]; ws.st.pd_target[wh + 2] = sp[rspSwingAnkle];
		return;
	}
	const real* sp = ws.st.params + mpMax + ws.st.state * spMax;
	ws.st.pd_target[jSpine0] = sp[spSpineCurve]; ws.st.pd_target[jSpine1] = sp[spSpineCurve]; ws.st.pd_target[jSpine2] = sp[spSpineCurve];
	ws.st.pd_target[jSpine3] = sp[spSpineCurve]; ws.st.pd_target[jTorso] = sp[spSpineCurve];
	ws.st.pd_target[jShoulder] = sp[spShoulder]; ws.st.pd_target[jElbow] = sp[spElbow];
	ws.st.pd_target[jHip] = sp[spHip]; ws.st.pd_target[jKnee] = sp[spKnee]; ws.st.pd_target[jAnkle] = sp[spAnkle];
}
template <class W>
DTRL_HD inline void transition_state(W& ws, int s) { ws.st.state = s; ws.st.phase = 0; set_state_params(ws); }
// cRaptorController::SetStance, sim/RaptorController.cpp:1439-1446
template <class W>
DTRL_HD inline void set_stance(W& ws, int st)
{
	ws.st.stance = st;
	ws.st.pd_active_bits &= ~(1u << stance_joint(ws, 0));
	ws.st.pd_active_bits |= (1u << swing_joint(ws, 0));
	set_state_params(ws);
}
// cRaptorController::IsActiveVFEffector, :1157-1163
template <class W>
DTRL_HD inline bool raptor_active_effector(const W& ws, int j)
{
	return j == stance_joint(ws, 3) && (ws.st.state == rstContact || ws.st.state == rstDown) && ((ws.st.contact_bits >> j) & 1u);
}
template <class W>
DTRL_HD inline bool has_stumbled(const W& ws)  // sim/SimDog.cpp:83-105, sim/SimRaptor.cpp:78-101
{
	uint32_t mask = (ws.M.char_type == 1) ? ~((1u << rRightToe) | (1u << rLeftToe) | (1u << rRightAnkle) | (1u << rLeftAnkle))
										   : ~((1u << jToe) | (1u << jFinger) | (1u << jAnkle) | (1u << jWrist));
	return (ws.st.contact_bits & mask & ((1u << ws.M.L) - 1u)) != 0;
}
template <class W>
DTRL_HD inline bool check_fall_contact(const W& ws)  // sim/SimDog.cpp:112-141 (root..head), sim/SimRaptor.cpp:108-137 (root..head)
{
	const int last = (ws.M.char_type == 1) ? int(rHead) : int(jHead);
	return (ws.st.contact_bits & ((1u << (last + 1)) - 1u)) != 0;
}
template <class W>
DTRL_HD inline bool has_fallen(const W& ws)  // sim/SimCharSoftFall.cpp:53-61, sim/SimDog.cpp:143-161
{
	bool flipped = fabs(wrap_pi(ws.st.q[2])) > 3.14159265358979323846 * 0.8;
	return ws.st.sum_fall_contact > 0.25 || ws.st.fail_fall_dist != 0 || flipped;
}
template <class W>
DTRL_HD inline bool is_new_cycle(const W& ws) { return ws.st.state == 0 && ws.st.phase == 0; }  // sim/CharController.cpp:72-75

DTRL_HD inline void blend_ctrl_params(const DevModel& gm, int a, real* out)  // sim/DogController.cpp:1335-1341
{
	const real* p0 = gm.ctrl_params[gm.act_idx0[a]]; const real* p1 = gm.ctrl_params[gm.act_idx1[a]];
	real b = gm.act_blend[a];
	for (int i = 0; i < gm.P; ++i) out[i] = (1 - b) * p0[i] + b * p1[i];
}
DTRL_HD inline void post_process_params(real* p, int char_type)
{
	p[0] = fabs(p[0]); p[1] = fabs(p[1]);                 // TransTime, Cv
	if (char_type == 1) p[rmpCd] = fabs(p[rmpCd]);         // sim/RaptorController.cpp:1401-1406
}
DTRL_HD inline int assign_frag_id(const DevModel& gm, int num_frags, int a_id, Rng& rng)  // sim/DogControllerMACE.cpp:44-91
{
	int frag_id = 0;
	if (num_frags > 0) {
		int id0 = gm.act_idx0[a_id], id1 = gm.act_idx1[a_id];
		if (id0 >= num_frags && id1 >= num_frags) frag_id = rng.rand_int(0, num_frags);
		else if (id0 >= num_frags) frag_id = id1;
		else if (id1 >= num_frags) frag_id = id0;
		else {
			frag_id = rng.flip() ? id0 : id1;
			int num_copies = num_frags / gm.n_sets;
			if (frag_id < num_frags % gm.n_sets) ++num_copies;
			int offset = rng.rand_int(0, num_copies);
			frag_id += offset * gm.n_sets;
		}
	}
	return frag_id;
}
DTRL_HD inline void build_base_action(const DevModel& gm, int num_frags, int a_id, Rng& rng, int* out_id, real* out_params)
{
	*out_id = a_id;
	blend_ctrl_params(gm, a_id, out_params);
	if (gm.ctrl_type == 1) *out_id = assign_frag_id(gm, num_frags, a_id, rng);
}
// cTerrainRLCharController::ApplyAction + cDogController::NewCycleUpdate + TransitionState(BackStance); lane-0 code
template <class W>
DTRL_HD inline void apply_action(W& ws, int id, const real* params, int P)
{
	ws.st.action_id = id;
	for (int i = 0; i < P; ++i) ws.st.params[i] = params[i];
	post_process_params(ws.st.params, ws.M.char_type);
	ws.st.prev_cycle_time = ws.st.curr_cycle_time; ws.st.curr_cycle_time = 0;
	ws.st.prev_stumble = ws.st.curr_stumble; ws.st.curr_stumble = 0;
	real com[2]; calc_com(ws, com);
	ws.st.prev_dist[0] = com[0] - ws.st.prev_com[0]; ws.st.prev_dist[1] = com[1] - ws.st.prev_com[1];
	ws.st.prev_com[0] = com[0]; ws.st.prev_com[1] = com[1];
	transition_state(ws, stBackStance);
}

// ---- policy network: learning/NeuralNet.cpp:352-375 Eval on the MACE topology, all 64 lanes of the env's wavefront.
// No activation leaves the CU. The three conv layers and terr_ip0 are fused over TILES of output positions: a tile is 16 positions
// of conv0's output (= one N-tile of the fp64 matrix pipe), of which 16 - (k1 - 1) are valid after conv1 and V = 16 - (k1 - 1) - (k2 - 1)
// after conv2 (10 for the reference's 8/4/4 kernels; the halo is recomputed: 19 tiles instead of 12, +55 % multiply-adds on layers that
// run on the matrix pipe at a fraction of the old vector cost). Each layer's tile lives in the Z storage (dead while the controller picks
// an action): a layer accumulates its whole tile in registers (co x 16 outputs = 8 doubles per lane), and only then overwrites its
// input tile. terr_ip0 consumes the V valid positions of a conv2 tile right away (lanes <-> its 64 outputs, running sums in a
// register across tiles), so the 32 x 187 conv2 blob never exists anywhere; the engine lays terr_ip0's weights out tile-major to match
// (BuildRelayoutMap). The normalised input sits in the side buffer (packed-Delassus storage on the fast path, H on the reference path),
// the trunk / head activations (531 doubles) in Z.
//   conv tile: out[o][j] = relu(b[o] + sum_{c, u} W[c][u][o] x[c][j + u]) as D = A B on v_mfma_f64_16x16x4_f64 with M = output channel,
//              N = position, K = (c, u) flattened, A straight from the [cin][k][cout] weight blob (L2), B from the LDS tile. The MFMA
//              accumulates in K order with fused multiply-adds (tools/microbench/mfma_f64_check.hip), i.e. the reference order
//              (channel-major, then tap) of the scalar loop that the CPU builds run: bit-identical.
//   FC:        lanes <-> outputs; the engine re-lays InnerProduct blobs as [nin/4][nout][4] so a lane fetches the weights of
//              4 inputs with one 16-byte load; weights of chunk b+1 are in flight while chunk b is consumed.
// Accumulation order per conv output is the reference order; terr_ip0 sums tile-major (tile, channel, position) instead of Caffe's
// gemv order, which is unspecified anyway (the reference computes in fp32 through BLAS) -- the oracle sums channel-major and the
// parity tests hold to their fp64 tolerance.
constexpr int kMaxConvCh = 32;
static_assert(kNNTileBuf == (kMaxRows + 1) * kZStride, "the tile buffer of the policy forward is the Z storage");
static_assert(kMaxConvCh * kConvTile % kGroup == 0, "lane-loop form of conv_tile");
#if defined(__HIP_DEVICE_COMPILE__)
#define LANE_LOCAL(type, name, n) type name##_ll[1][n]
#define LL(name) name##_ll[0]
#define LL_IN(arr) arr[0]
#else
#define LANE_LOCAL(type, name, n) type name##_ll[::dtrl::kGroup][n]
#define LL(name) name##_ll[lane]
#define LL_IN(arr) arr[lane]
#endif
#define LL_ARR(name) name##_ll   // the whole lane-local object, to hand it to a callee (which indexes it with LL_IN)
constexpr int kFcBatch = 8;
#if defined(__HIP_DEVICE_COMPILE__)
// scheduling hint: the batch's LDS reads (kFcBatch doubles) as one group ahead of its multiply-adds
#define FC_SCHED_BATCH() do { __builtin_amdgcn_sched_group_barrier(0x100, kFcBatch / 2, 0); __builtin_amdgcn_sched_group_barrier(0x002, 2 * kFcBatch, 0); } while (0)
#else
#define FC_SCHED_BATCH() do {} while (0)
#endif
DTRL_HD inline int64_t fc_dev_size(int nout, int nin) { return static_cast<int64_t>((nin + 3) / 4) * 4 * nout; }
DTRL_HD inline int64_t pad4(int64_t n) { return (n + 3) / 4 * 4; }   // every device blob starts 16-byte aligned
struct alignas(16) F4 { float v[4]; };
DTRL_HD_INLINE real* nn_side_buf(WSRef& ws) { return &ws.H[0][0]; }
DTRL_HD_INLINE real* nn_side_buf(WSFast& ws) { return &ws.Apk[0]; }
static_assert(sizeof(WSRef::H) >= sizeof(real) * kNNSideBuf && sizeof(WSFast::Apk) >= sizeof(real) * kNNSideBuf, "side buffer of the policy forward");

// one position tile of a conv layer. x: LDS, channel stride xs, position index clamped to xlast (conv0 reads the input vector itself
// and runs off its end in the last tile; the outputs that see a clamped or never-written input are the halo the next layer discards).
// out (may alias x: everything is accumulated before anything is written): [o][os] for nv < 0, else the compact [o][nv] block of the
// first nv positions that terr_ip0 consumes.
// NB = K-steps (of 4 flattened (c, u) indices) per batch on the device: the batch's 2 NB weight loads are issued together, one batch ahead of
// the matrix instructions that consume them (a one-wave forward is latency-bound otherwise); cin * k must be a multiple of 4 NB.
template <int NB, class W>
DTRL_HD inline void conv_tile(W& ws, const float* Wd, const float* bias, int co, int cin, int k, const real* x, int xs, int xlast, real* out, int os, int nv)
{
	(void)ws;
#if defined(__HIP_DEVICE_COMPILE__)
	// operand layout: A[i = lane % 16][k = lane / 16], B[k = lane / 16][n = lane % 16], D register r = D[4 r + lane / 16][lane % 16]
	typedef v4r_t v4d_t;
	const int l = static_cast<int>(threadIdx.x), g = l >> 4, j = l & 15;
	const bool two = co > kConvTile;                 // a 16-channel layer runs the second tile on the first one's weights and drops the result
	v4d_t acc0, acc1;
#pragma unroll
	for (int r = 0; r < 4; ++r) { acc0[r] = static_cast<real>(bias[mfma_row(r, g)]); acc1[r] = static_cast<real>(bias[two ? 16 + mfma_row(r, g) : 0]); }
	const int nk = cin * k, ksh = (k == 8) ? 3 : 2;  // kernel widths are 4 or 8 (host check)
	const float* w0 = Wd + g * co + j;              // this lane's A entries: W[kk + g][j] and W[kk + g][16 + j]
	const float* w1 = w0 + (two ? kConvTile : 0);
	float a0[NB], a1[NB];
#pragma unroll
	for (int q = 0; q < NB; ++q) { a0[q] = w0[4 * q * co]; a1[q] = w1[4 * q * co]; }
	for (int kk = 0; kk < nk; kk += 4 * NB) {
		const int kn = (kk + 4 * NB < nk) ? kk + 4 * NB : kk;   // next batch (the last one re-reads itself: no branch around the loads)
		float n0[NB], n1[NB]; real b[NB];
#pragma unroll
		for (int q = 0; q < NB; ++q) { n0[q] = w0[(kn + 4 * q) * co]; n1[q] = w1[(kn + 4 * q) * co]; }
#pragma unroll
		for (int q = 0; q < NB; ++q) {
			const int e = kk + 4 * q + g, c = e >> ksh, xi = j + (e & (k - 1));
			b[q] = x[c * xs + (xi < xlast ? xi : xlast)];
		}
#pragma unroll
		for (int q = 0; q < NB; ++q) {
			acc0 = mfma_16x16x4(static_cast<real>(a0[q]), b[q], acc0);
			acc1 = mfma_16x16x4(static_cast<real>(a1[q]), b[q], acc1);
		}
#pragma unroll
		for (int q = 0; q < NB; ++q) { a0[q] = n0[q]; a1[q] = n1[q]; }
	}
	env_sync();
#pragma unroll
	for (int r = 0; r < 4; ++r) {
		const int o0 = mfma_row(r, g), o1 = 16 + o0;
		const real v0 = acc0[r] < 0 ? 0 : acc0[r], v1 = acc1[r] < 0 ? 0 : acc1[r];
		if (nv < 0) { out[o0 * os + j] = v0; if (two) out[o1 * os + j] = v1; }
		else if (j < nv) { out[o0 * nv + j] = v0; if (two) out[o1 * nv + j] = v1; }
	}
	env_sync();
#else
	constexpr int kPer = kMaxConvCh * kConvTile / kGroup;
	LANE_LOCAL(real, acc, kPer);
	LANES_BEGIN
	for (int q = 0; q < kPer; ++q) {
		const int idx = q * kGroup + lane, o = idx / kConvTile, j = idx % kConvTile;
		if (o >= co) continue;
		real a = static_cast<real>(bias[o]);
		for (int c = 0; c < cin; ++c) for (int u = 0; u < k; ++u) {
			const int xi = j + u;
			a = fmadd(static_cast<real>(Wd[(c * k + u) * co + o]), x[c * xs + (xi < xlast ? xi : xlast)], a);
		}
		LL(acc)[q] = a < 0 ? 0 : a;
	}
	LANES_END
	LANES_BEGIN
	for (int q = 0; q < kPer; ++q) {
		const int idx = q * kGroup + lane, o = idx / kConvTile, j = idx % kConvTile;
		if (o >= co) continue;
		if (nv < 0) out[o * os + j] = LL(acc)[q];
		else if (j < nv) out[o * nv + j] = LL(acc)[q];
	}
	LANES_END
#endif
}
// s[o] += sum_{i < nin} W[i][o] x[i] for the lane's output o (nout <= 64), x in LDS, weights in [nin/4][nout][4] blocks; input order ascending
template <class W, class S>
DTRL_HD inline void fc_partial(W& ws, const float* Wb, int nout, int nin, const real* x, S& s_ll)
{
	(void)ws;
	const int nblk = (nin + 3) / 4;
	constexpr int kQ = kFcChunk / 4;   // 16-byte weight loads per lane and chunk
	LANE_LOCAL(float, wc, kFcChunk);
	LANES_BEGIN
	const int oc = lane < nout ? lane : nout - 1;
#pragma unroll
	for (int q = 0; q < kQ; ++q) {
		const int bq = q < nblk ? q : nblk - 1;
		const F4 w4 = *reinterpret_cast<const F4*>(Wb + (static_cast<int64_t>(bq) * nout + oc) * 4);
#pragma unroll
		for (int r = 0; r < 4; ++r) LL(wc)[4 * q + r] = w4.v[r];
	}
	LANES_END
	for (int i0 = 0; i0 < nin; i0 += kFcChunk) {
		LANES_BEGIN
		const int oc = lane < nout ? lane : nout - 1;
		const bool more = i0 + kFcChunk < nin;
		float wn[kFcChunk];
		const int b1 = (i0 + kFcChunk) / 4;
#pragma unroll
		for (int q = 0; q < kQ; ++q) {   // next chunk's weights: clamped addresses, no selects (the loads must not wait for anything)
			const int bq = b1 + q < nblk ? b1 + q : nblk - 1;
			const F4 w4 = *reinterpret_cast<const F4*>(Wb + (static_cast<int64_t>(bq) * nout + oc) * 4);
#pragma unroll
			for (int r = 0; r < 4; ++r) wn[4 * q + r] = w4.v[r];
		}
		real acc = LL_IN(s_ll)[0];
		const int nhere = (nin - i0 < kFcChunk) ? nin - i0 : kFcChunk;
		if (nhere == kFcChunk) {
#pragma unroll
			for (int e0 = 0; e0 < kFcChunk; e0 += kFcBatch) {   // the batch's LDS reads in flight together, one wait per batch
				real xv[kFcBatch];
#pragma unroll
				for (int r = 0; r < kFcBatch; ++r) xv[r] = x[i0 + e0 + r];
				FC_SCHED_BATCH();
#pragma unroll
				for (int r = 0; r < kFcBatch; ++r) acc = fmadd(static_cast<real>(LL(wc)[e0 + r]), xv[r], acc);
			}
		} else {
#pragma unroll
			for (int e = 0; e < kFcChunk; ++e) if (e < nhere) acc = fmadd(static_cast<real>(LL(wc)[e]), x[i0 + e], acc);
		}
		LL_IN(s_ll)[0] = acc;
		if (more) {
#pragma unroll
			for (int e = 0; e < kFcChunk; ++e) LL(wc)[e] = wn[e];
		}
		LANES_END
	}
}
template <class W>
DTRL_HD inline void fc_layer(W& ws, const float* Wb, const float* b, int nout, int nin, const real* x, real* y, bool relu)
{
	real* lds = &ws.Z[0][0];   // two chunks of kFcChunk inputs
	const int nblk = (nin + 3) / 4;
	constexpr int kQ = kFcChunk / 4;   // 16-byte weight loads per lane and chunk
	for (int o0 = 0; o0 < nout; o0 += kGroup) {
		LANE_LOCAL(real, s, 1);
		LANE_LOCAL(float, wc, kFcChunk);
		LANES_BEGIN
		const int o = o0 + lane;
		if (lane < kFcChunk) lds[lane] = (lane < nin) ? x[lane] : 0.0;
		LL(s)[0] = (o < nout) ? static_cast<real>(b[o]) : 0.0;
#pragma unroll
		for (int q = 0; q < kQ; ++q) {
			const bool ok = o < nout && q < nblk;
			const F4 w4 = *reinterpret_cast<const F4*>(Wb + (static_cast<int64_t>(ok ? q : 0) * nout + (ok ? o : 0)) * 4);
#pragma unroll
			for (int r = 0; r < 4; ++r) LL(wc)[4 * q + r] = ok ? w4.v[r] : 0.0f;
		}
		LANES_END
		for (int i0 = 0; i0 < nin; i0 += kFcChunk) {
			LANES_BEGIN
			const int o = o0 + lane;
			const int cur = ((i0 / kFcChunk) & 1) * kFcChunk, nxt = kFcChunk - cur;
			const bool more = i0 + kFcChunk < nin;
			// next chunk: one input per lane, kQ 16-byte weight loads per lane; clamped addresses, no selects (the loads must not wait for anything)
			const int in = i0 + kFcChunk + (lane < kFcChunk ? lane : 0);
			const real xn = x[in < nin ? in : nin - 1];
			float wn[kFcChunk];
			const int b1 = (i0 + kFcChunk) / 4;
			const int oc = o < nout ? o : nout - 1;
#pragma unroll
			for (int q = 0; q < kQ; ++q) {
				const int bq = b1 + q < nblk ? b1 + q : nblk - 1;
				const F4 w4 = *reinterpret_cast<const F4*>(Wb + (static_cast<int64_t>(bq) * nout + oc) * 4);
#pragma unroll
				for (int r = 0; r < 4; ++r) wn[4 * q + r] = w4.v[r];
			}
			real acc = LL(s)[0];
			const int nhere = (nin - i0 < kFcChunk) ? nin - i0 : kFcChunk;
			if (nhere == kFcChunk) {   // straight-line block for full chunks
#pragma unroll
				for (int e0 = 0; e0 < kFcChunk; e0 += kFcBatch) {   // the batch's LDS reads in flight together, one wait per batch
				real xv[kFcBatch];
#pragma unroll
				for (int r = 0; r < kFcBatch; ++r) xv[r] = lds[cur + e0 + r];
				FC_SCHED_BATCH();
#pragma unroll
				for (int r = 0; r < kFcBatch; ++r) acc = fmadd(static_cast<real>(LL(wc)[e0 + r]), xv[r], acc);
			}
			} else {
#pragma unroll
				for (int e = 0; e < kFcChunk; ++e) if (e < nhere) acc = fmadd(static_cast<real>(LL(wc)[e]), lds[cur + e], acc);
			}
			LL(s)[0] = acc;
			if (more) {
				if (lane < kFcChunk) lds[nxt + lane] = xn;
#pragma unroll
				for (int e = 0; e < kFcChunk; ++e) LL(wc)[e] = wn[e];
			}
			LANES_END
		}
		LANES_BEGIN
		const int o = o0 + lane;
		if (o < nout) y[o] = (relu && LL(s)[0] < 0) ? 0 : LL(s)[0];
		LANES_END
	}
}
template <class W>
DTRL_HD inline void nn_eval(W& ws, const DevBuffers& buf, int env)
{
	const NetDesc& d = buf.net;
	const real* xin = buf.poli_state + static_cast<int64_t>(env) * buf.S;
	real* y = buf.nn_out + static_cast<int64_t>(env) * d.out_size;
	real* side = nn_side_buf(ws);        // [n_terrain | n_char] normalised input
	real* tile = &ws.Z[0][0];
	const float* p = buf.weights;
	const float* Wc[3]; const float* bc[3];
	int cin = 1, wdt = d.n_terrain;
	for (int l = 0; l < 3; ++l) {
		Wc[l] = p; bc[l] = p + pad4(static_cast<int64_t>(d.conv_ch[l]) * cin * d.conv_k[l]);
		p = bc[l] + pad4(d.conv_ch[l]); cin = d.conv_ch[l]; wdt = wdt - d.conv_k[l] + 1;
	}
	const int wo = wdt, nflat = cin * wo;                                   // conv2's width; terr_ip0's fan-in
	const float* Wt = p; const float* bt = p + fc_dev_size(d.fc_terr, nflat);
	p = bt + pad4(d.fc_terr);
	const int s0 = kConvTile + d.conv_k[1] - 1, s1 = kConvTile + d.conv_k[2] - 1;   // tile row strides: 16 outputs + the next layer's reach
	const int V = kConvTile - (d.conv_k[1] - 1) - (d.conv_k[2] - 1);
	LANE_LOCAL(real, sterr, 1);
	LANES_BEGIN
	for (int i = lane; i < d.in_size; i += kGroup) side[i] = (xin[i] + buf.in_off[i]) * buf.in_scale[i];
	LL(sterr)[0] = static_cast<real>(bt[lane < d.fc_terr ? lane : 0]);
	LANES_END
	PROF_T0();
	for (int p0 = 0; p0 < wo; p0 += V) {
		const int vt = (wo - p0 < V) ? wo - p0 : V;
		conv_tile<1>(ws, Wc[0], bc[0], d.conv_ch[0], 1, d.conv_k[0], side + p0, 0, d.n_terrain - 1 - p0, tile, s0, -1);
		conv_tile<8>(ws, Wc[1], bc[1], d.conv_ch[1], d.conv_ch[0], d.conv_k[1], tile, s0, s0 - 1, tile, s1, -1);
		conv_tile<8>(ws, Wc[2], bc[2], d.conv_ch[2], d.conv_ch[1], d.conv_k[2], tile, s1, s1 - 1, tile, 0, vt);
		const unsigned long long prof_fc_t0 = PROF_NOW();
		fc_partial(ws, Wt + static_cast<int64_t>(p0) * cin * d.fc_terr, d.fc_terr, cin * vt, tile, LL_ARR(sterr));
		PROF_ADD_SINCE(ws, kProfNNFcTerr, prof_fc_t0);
	}
	PROF_ADD(ws, kProfNNConv);   // conv tiles + terr_ip0 (kProfNNFcTerr is the terr_ip0 share)
	const unsigned long long prof_rest_t0 = PROF_NOW();
	// Z from here on: [0, 64) fc_layer's staging, then the trunk input [terr_ip0 | char features], the trunk and one head's hidden layer
	const int ntr = d.fc_terr + d.n_char;
	real* trunk_in = tile + 2 * kFcChunk;
	real* trunk = trunk_in + ntr;
	real* head = trunk + d.fc_trunk;
	LANES_BEGIN
	if (lane < d.fc_terr) trunk_in[lane] = LL(sterr)[0] < 0 ? 0 : LL(sterr)[0];
	for (int i = lane; i < d.n_char; i += kGroup) trunk_in[d.fc_terr + i] = side[d.n_terrain + i];
	LANES_END
	fc_layer(ws, p, p + fc_dev_size(d.fc_trunk, ntr), d.fc_trunk, ntr, trunk_in, trunk, true);
	p += fc_dev_size(d.fc_trunk, ntr) + pad4(d.fc_trunk);
	fc_layer(ws, p, p + fc_dev_size(d.fc_head, d.fc_trunk), d.fc_head, d.fc_trunk, trunk, head, true);
	p += fc_dev_size(d.fc_head, d.fc_trunk) + pad4(d.fc_head);
	fc_layer(ws, p, p + fc_dev_size(d.n_frags, d.fc_head), d.n_frags, d.fc_head, head, y, false);
	p += fc_dev_size(d.n_frags, d.fc_head) + pad4(d.n_frags);
	for (int f = 0; f < d.n_frags; ++f) {
		fc_layer(ws, p, p + fc_dev_size(d.fc_head, d.fc_trunk), d.fc_head, d.fc_trunk, trunk, head, true);
		p += fc_dev_size(d.fc_head, d.fc_trunk) + pad4(d.fc_head);
		fc_layer(ws, p, p + fc_dev_size(d.frag_size, d.fc_head), d.frag_size, d.fc_head, head, y + d.n_frags + f * d.frag_size, false);
		p += fc_dev_size(d.frag_size, d.fc_head) + pad4(d.frag_size);
	}
	LANES_BEGIN
	for (int i = lane; i < d.out_size; i += kGroup) y[i] = y[i] / buf.out_scale[i] - buf.out_off[i];
	LANES_END
	PROF_ADD_SINCE(ws, kProfNNRest, prof_rest_t0);
	PROF_COUNT(ws, kProfNNEvals);
}

// cDogController(MACE)::UpdateAction: ParseGround + BuildPoliState + action decision + ApplyAction
template <class W>
DTRL_HD inline void update_action(W& ws, const DevModel& gm, const RunParams& rp, const DevBuffers& buf, const GroundRec& g, int env)
{
	real* ps = buf.poli_state + static_cast<int64_t>(env) * buf.S;
	const int L = ws.M.L;
	// ParseGround, sim/TerrainRLCharController.cpp:168-213
	real origin_x = ws.st.q[0];
	real origin_y = sample_ground(g, origin_x, nullptr, nullptr, nullptr, nullptr);
	LANES_BEGIN
	for (int s = lane; s < kNumGroundSamples; s += kGroup) {
		real dist = ((10.0 - (-0.5)) * s) / (kNumGroundSamples - 1) + (-0.5);
		ps[s] = sample_ground(g, dist + origin_x, nullptr, nullptr, nullptr, nullptr) - origin_y;
	}
	// BuildPoliState, :215-285 (ENABLE_MAX_COORD_POSE)
	if (lane == 0) { ps[kNumGroundSamples] = ws.st.q[1] - origin_y; ws.st.sample_origin[0] = origin_x; ws.st.sample_origin[1] = origin_y; }
	if (lane >= 1 && lane < L) { ps[kNumGroundSamples + 1 + 2 * (lane - 1)] = ws.cx[lane]; ps[kNumGroundSamples + 2 + 2 * (lane - 1)] = ws.cy[lane]; }
	if (lane < L) { ps[kNumGroundSamples + 2 * L - 1 + 2 * lane] = ws.vcx[lane]; ps[kNumGroundSamples + 2 * L + 2 * lane] = ws.vcy[lane]; }
	LANES_END
	if (gm.char_type == 1 && ws.st.stance != 0) {
		// cRaptorController::BuildPoliStatePose/Vel -> FlipPoliPoseStance (sim/RaptorController.cpp:1414-1488): mirror the two
		// trailing leg blocks (4 links x 2) of the pose and of the velocity features when the left leg is the stance leg
		LANES_BEGIN
		if (lane < 16) {
			const int nleg = 8;
			const int end = (lane < 8) ? kNumGroundSamples + (2 * L - 1) : kNumGroundSamples + (2 * L - 1) + 2 * L;
			const int i = lane & 7;
			real a = ps[end - 1 - i], b2 = ps[end - nleg - 1 - i];
			ps[end - 1 - i] = b2; ps[end - nleg - 1 - i] = a;
		}
		LANES_END
	}
	// decide which branch of UpdateAction runs (lane 0 draws the random numbers; the branch flag is broadcast via LDS)
	LANES_BEGIN
	if (lane == 0) {
		if (gm.ctrl_type == 1) { ws.st.exp_actor = 0; ws.st.exp_critic = 0; }
		ws.st.is_off_policy = 1;
		int mode = 0;  // 0: keep / default action, 1: command, 2: net, 3: random base action (MACE / CACLA exploration), 4: random base action (Q exploration)
		if (ws.st.cmd_action >= 0) mode = 1;
		else if (gm.has_net && gm.ctrl_type == 2) {
			// cBaseControllerCacla::DecideAction / ShouldExplore / ExploreAction (sim/BaseControllerCacla.cpp:124-151, 219-234): explore with
			// probability exp_rate; an exploring step is a random base action with probability exp_base_rate, the actor's output plus noise otherwise
			Rng rng = make_rng(rp, env, &ws.st.rng_ctr);
			const bool explore = rp.enable_exp && rng.uniform() < rp.exp_rate;
			ws.st.is_off_policy = explore ? 1 : 0;
			ws.st.exp_actor = 0;
			mode = 2;
			if (explore) { if (rng.uniform() < rp.exp_base_rate) mode = 3; else ws.st.exp_actor = 1; }   // exp_actor: add parameter noise below
		}
		else if (gm.has_net && gm.ctrl_type == 0) {
			// cBaseControllerQ::DecideAction / ShouldExplore (sim/BaseControllerQ.cpp:32-57): a random base action with probability exp_rate
			Rng rng = make_rng(rp, env, &ws.st.rng_ctr);
			const bool explore = rp.enable_exp && rng.uniform() < rp.exp_rate;
			ws.st.is_off_policy = explore ? 1 : 0;
			mode = explore ? 4 : 2;
		}
		else if (gm.has_net) {
			Rng rng = make_rng(rp, env, &ws.st.rng_ctr);
			ws.st.is_off_policy = 0;
			real base_rand = rng.uniform();
			mode = (rp.enable_exp && base_rand < rp.exp_base_rate) ? 3 : 2;
		}
		ws.flag_misc = mode;
	}
	LANES_END
	if (ws.flag_misc == 2) nn_eval(ws, buf, env);
	LANES_BEGIN
	if (lane == 0) {
		Rng rng = make_rng(rp, env, &ws.st.rng_ctr);
		const int P = gm.P, nf = buf.net.n_frags;
		const int num_frags = gm.has_net ? nf : 0;
		int id = ws.st.action_id; real prm[kMaxP];
		for (int i = 0; i < P; ++i) prm[i] = ws.st.params[i];
		const int mode = ws.flag_misc;
		if (mode == 1) {
			int cmd = ws.st.cmd_action; ws.st.cmd_action = -1;
			if (gm.ctrl_type == 1) { ws.st.exp_actor = 1; ws.st.exp_critic = 1; }
			build_base_action(gm, num_frags, cmd, rng, &id, prm);
		} else if (mode == 3) {
			int a = rng.rand_int(0, gm.n_actions);
			build_base_action(gm, num_frags, a, rng, &id, prm);
			ws.st.is_off_policy = 1; ws.st.exp_actor = 1; ws.st.exp_critic = 1;
		} else if (mode == 4) {
			build_base_action(gm, num_frags, rng.rand_int(0, gm.n_actions), rng, &id, prm);   // cBaseControllerQ::ExploreAction
		} else if (mode == 2 && gm.ctrl_type == 0) {
			// cBaseControllerQ::ExploitPolicy (sim/BaseControllerQ.cpp:59-82): the base action with the largest value (Eigen maxCoeff: the first one);
			// like the CACLA actor, the device net carries an unused critic slot in front of its outputs
			const real* y = buf.nn_out + static_cast<int64_t>(env) * buf.net.out_size + 1;
			int a = 0; for (int i = 1; i < gm.n_actions; ++i) if (y[i] > y[a]) a = i;
			build_base_action(gm, num_frags, a, rng, &id, prm);
		} else if (mode == 2 && gm.ctrl_type == 2) {
			// cBaseControllerCacla::ExploitPolicy (:205-217) + ApplyExpNoiseAction (:262-296). The device net carries a (zero) critic slot in
			// front of the actor's outputs (dtrl_engine.cpp SetPolicy), hence the offset of one
			const real* y = buf.nn_out + static_cast<int64_t>(env) * buf.net.out_size;
			id = -1;                                                                     // gInvalidIdx
			for (int k = 0; k < gm.n_opt; ++k) prm[gm.opt_index[k]] = y[1 + k];
			post_process_params(prm, gm.char_type);
			if (ws.st.exp_actor) for (int k = 0; k < gm.n_opt; ++k) { real noise = rng.normal(0, rp.exp_noise); prm[gm.opt_index[k]] += noise * (1.0 / buf.out_scale[1 + k]); }
		} else if (mode == 2) {
			// sim/BaseControllerMACE.cpp:267-296
			const real* y = buf.nn_out + static_cast<int64_t>(env) * buf.net.out_size;
			int a_max = 0; for (int i = 1; i < nf; ++i) if (y[i] > y[a_max]) a_max = i;
			int a = a_max;
			if (rp.enable_exp && rp.exp_temp != 0) {
				real vb[kMaxFrags]; real max_val = y[a_max], sum = 0;
				for (int i = 0; i < nf; ++i) { vb[i] = exp((y[i] - max_val) / rp.exp_temp); sum += vb[i]; }
				real r = rng.uniform(0, sum);
				for (int i = 0; i < nf; ++i) { r -= vb[i]; if (r <= 0) { a = i; break; } }
			}
			id = a;
			const real* frag = y + nf + a * buf.net.frag_size;
			for (int k = 0; k < gm.n_opt; ++k) prm[gm.opt_index[k]] = frag[k];
			post_process_params(prm, gm.char_type);
			if (rp.enable_exp) {
				real rand_noise = rng.uniform();
				if (rand_noise < rp.exp_rate) {
					for (int k = 0; k < gm.n_opt; ++k) { real noise = rng.normal(0, rp.exp_noise); prm[gm.opt_index[k]] += noise * (1.0 / buf.out_scale[nf + k]); }
					ws.st.exp_actor = 1;
				}
				ws.st.exp_critic = (a != a_max) ? 1 : 0;
				ws.st.is_off_policy = (ws.st.exp_actor || ws.st.exp_critic) ? 1 : 0;
			}
		} else {
			bool cyclic = (gm.ctrl_type >= 1) ? false : (gm.act_cyclic[ws.st.action_id] != 0);
			if (!cyclic) build_base_action(gm, num_frags, gm.default_action, rng, &id, prm);
		}
		apply_action(ws, id, prm, P);
	}
	LANES_END
}

// effector contact position (body-local (0, -size_y/2)), relative to the root origin; sim/DogController.cpp:1372-1387
template <class W>
DTRL_HD inline void effector_pos(const W& ws, const DevModel& gm, int j, real* out)
{
	const real lx = gm.eff_joint[j][0], ly = gm.eff_joint[j][1];
	out[0] = ws.px[j] + ws.cs[j] * lx - ws.sn[j] * ly;
	out[1] = ws.py[j] + ws.sn[j] * lx + ws.cs[j] * ly;
}
// J_d^T applied to a world force f acting at pos (relative to root): translation DoFs see f, hinge a sees (pos - p_a) x f
template <class W>
DTRL_HD inline real jt_force(const W& ws, int d, const real* pos, const real* f)
{
	if (d == 0) return f[0];
	if (d == 1) return f[1];
	const int a = d - 2;
	return (pos[0] - ws.px[a]) * f[1] - (pos[1] - ws.py[a]) * f[0];
}

// reference (LDS-phase) implicit-PD solve: ws.u holds the right-hand side on entry and the acceleration on exit
template <class W>
DTRL_HD inline void pd_solve_ref(W& ws, real dt)
{
	const int D = ws.M.D;
	mass_matrix(ws);
	LANES_BEGIN
	if (lane < D) ws.H[lane][lane] += dt * ws.kdm[lane];
	if (lane == 0) { ws.R = 0; }
	LANES_END
	factorize(ws);
	forward_subst_rows(ws, ws.u);   // R = 0: only z_0 = U^-1 rhs (lane 0)
	LANES_BEGIN
	if (lane < D) ws.u[lane] = ws.Z[0][lane] * ws.dinv[lane];
	LANES_END
	for (int k = 0; k < D - 1; ++k) {   // U^-T: x_i -= U_ki x_k for i > k (U_ki sits at H[i][k])
		LANES_BEGIN
		if (lane > k && lane < D) ws.u[lane] = fmadd(-ws.H[lane][k], ws.u[k], ws.u[lane]);
		LANES_END
	}
}
struct RefPath {
	template <class W> static DTRL_HD void substep(W& ws, const DevModel& gm, const GroundRec& g, real h, bool kin_valid) { substep_ref(ws, gm, g, h, kin_valid); }
	template <class W> static DTRL_HD void pd_solve(W& ws, real dt) { pd_solve_ref(ws, dt); }
	template <class W> static DTRL_HD void contacts(W& ws, const DevModel& gm, const GroundRec& g, real) { detect_contacts(ws, gm, g); }
};

// 4x4 ridge solve of the contact-basis least squares (partial-pivot elimination), shared by both characters
DTRL_HD inline void solve_ls4(const real (*basis)[4], const real* tau_g, const real* W, real* x)
{
	real M4[4][5], ipiv[4];
#pragma unroll
	for (int a = 0; a < 4; ++a) {
#pragma unroll
		for (int c = 0; c < 4; ++c) { real s = 0; for (int r = 0; r < 3; ++r) s += basis[r][a] * W[r] * basis[r][c]; M4[a][c] = s; }
		real s = 0; for (int r = 0; r < 3; ++r) s += basis[r][a] * W[r] * tau_g[r];
		M4[a][4] = s; M4[a][a] += 0.0001;
	}
	// every index below is a compile-time constant after unrolling (the pivot row is applied as predicated swaps against each candidate
	// row, not as M4[p]): a dynamically indexed private array would live in scratch memory and cost a write-back to HBM per env-step
#pragma unroll
	for (int c = 0; c < 4; ++c) {
		int p = c; real best = fabs(M4[c][c]);
#pragma unroll
		for (int r = c + 1; r < 4; ++r) { const real v = fabs(M4[r][c]); if (v > best) { best = v; p = r; } }
#pragma unroll
		for (int r = c + 1; r < 4; ++r) {
			const bool sw = (p == r);
#pragma unroll
			for (int k = 0; k < 5; ++k) { const real a = M4[c][k], b = M4[r][k]; M4[c][k] = sw ? b : a; M4[r][k] = sw ? a : b; }
		}
		ipiv[c] = fast_recip(M4[c][c]);   // one reciprocal per pivot, reused by the back substitution (10 divisions -> 4 reciprocals)
#pragma unroll
		for (int r = c + 1; r < 4; ++r) {
			const real f = M4[r][c] * ipiv[c];
#pragma unroll
			for (int k = c; k < 5; ++k) M4[r][k] -= f * M4[c][k];
		}
	}
#pragma unroll
	for (int i = 3; i >= 0; --i) {
		real s = M4[i][4];
#pragma unroll
		for (int k = i + 1; k < 4; ++k) s -= M4[i][k] * x[k];
		x[i] = s * ipiv[i];
	}
}

// cDogController::Update (sim/DogController.cpp:229-268) / cRaptorController::Update (sim/RaptorController.cpp:195-233)
template <class Path, class W>
DTRL_HD inline void controller_update(W& ws, const DevModel& gm, const RunParams& rp, const DevBuffers& buf, const GroundRec& g, int env, real dt)
{
	const int D = ws.M.D, L = ws.M.L;
	const bool raptor = gm.char_type == 1;
	unsigned long long pc_t = PROF_NOW();
	// UpdateRBDModel: kinematics, composite inertias and the textbook bias at the post-step configuration were produced by
	// kin_dyn_terms() in env_step (quirk_bias() turns it into the reference's C); H itself is assembled inside the PD solve
	LANES_BEGIN
	if (lane == 0) {
		ws.st.curr_cycle_time += dt;
		if (has_stumbled(ws)) ws.st.curr_stumble += dt;
		// UpdateState: dog :805-845, raptor :804-849
		bool advance = ws.st.first_cycle != 0;
		real trans_time = ws.st.params[0];
		ws.st.phase += dt / trans_time;
		const int state = ws.st.state;
		int do_update = 0;
		if (raptor) {
			if (state != rstUp && ws.st.phase >= 1) advance = true;
			if (state == rstUp && in_contact(ws, swing_joint(ws, 3))) advance = true;
			if (advance) {
				int next = ws.st.first_cycle ? rstContact : ((state == rstUp) ? rstInvalid : state + 1);
				bool end_step = (next == rstInvalid) || ws.st.first_cycle;
				if (end_step) { if (!ws.st.first_cycle) set_stance(ws, ws.st.stance == 0 ? 1 : 0); do_update = 1; }   // FlipStance precedes UpdateAction
				else transition_state(ws, next);
			}
		} else {
			if ((state == stBackStance || state == stFrontStance) && ws.st.phase >= 1) advance = true;
			int trans_contact = (state == stExtend) ? jFinger : ((state == stGather) ? jToe : -1);
			if (trans_contact >= 0 && in_contact(ws, trans_contact)) advance = true;
			if (advance) {
				int next = ws.st.first_cycle ? stBackStance : ((state == stGather) ? stInvalid : state + 1);
				bool end_step = (next == stInvalid) || ws.st.first_cycle;
				if (end_step) do_update = 1; else transition_state(ws, next);
			}
		}
		ws.flag_update_action = do_update;
	}
	LANES_END
	PROF_ADD_SINCE(ws, kProfC_Fsm, pc_t); pc_t = PROF_NOW();
	if (__builtin_expect(ws.flag_update_action != 0, 0)) {   // once per gait cycle: cold, keep its register pressure out of the step loop
		PROF_T0();
		update_action(ws, gm, rp, buf, g, env);
		PROF_ADD(ws, kProfAction);
		LANES_BEGIN
		if (lane == 0) ws.st.first_cycle = 0;
		LANES_END
	}
	LANES_BEGIN
	if (lane == 0) {
		real sx = 0, m = 0;
		for (int j = 0; j < L; ++j) { sx += ws.M.mass[j] * ws.vcx[j]; m += ws.M.mass[j]; }
		real com_vx = sx / m;
		if (raptor) {
			// UpdateStanceHip :899-905
			const int sh = stance_joint(ws, 0), st_toe = stance_joint(ws, 3);
			if (raptor_active_effector(ws, st_toe)) ws.st.pd_active_bits &= ~(1u << sh); else ws.st.pd_active_bits |= (1u << sh);
			// ApplySwingFeedback :907-931 (SIMBICON-style: cd * d + cv * v on the swing hip, every step)
			real cv = ws.st.params[rmpCv], cd = ws.st.params[rmpCd];
			const bool first_half = ws.st.state == rstContact || ws.st.state == rstDown;
			cd = first_half ? 0 : cd; cv = first_half ? cv : 0;
			real com[2]; calc_com(ws, com);
			real d_theta = cd * (com[0] - (ws.st.q[0] + ws.cx[st_toe])) + cv * com_vx;
			ws.st.pd_target[swing_joint(ws, 0)] = ws.st.params[rmpMax + ws.st.state * rspMax + rspSwingHip] + d_theta;
		} else {
			// ApplyFeedback :903-945 (COM velocity feedback on hip / shoulder while the matching effector is airborne)
			const int joints[2] = {jHip, jShoulder}, effs[2] = {jToe, jFinger}, prm[2] = {spHip, spShoulder};
			for (int k = 0; k < 2; ++k) if (!in_contact(ws, effs[k])) {
				real default_theta = ws.st.params[mpMax + ws.st.state * spMax + prm[k]];
				ws.st.pd_target[joints[k]] = default_theta + com_vx * ws.st.params[mpCv];
			}
		}
	}
	LANES_END
	PROF_ADD_SINCE(ws, kProfC_Feedback, pc_t); pc_t = PROF_NOW();
	// cImpPDController::CalcControlForces: (H + dt Kd) acc = Kp (e - dt qd) + Kd e_dot - C;  tau = Kp (e - dt qd) + Kd (e_dot - dt acc)
	// inactive controllers (raptor stance hip) drop out of Kp/Kd but their raw Kd stays on the diagonal (sim/ImpPDController.cpp:244-258)
	LANES_BEGIN
	if (lane < D) {
		const int i = lane;
		real kp = 0, kd = 0, kdm = 0, pe = 0, ve = 0;
		if (i >= 3) {
			const int j = i - 2;
			kdm = gm.kd[j];
			if ((ws.st.pd_active_bits >> j) & 1u) { kp = gm.kp[j]; kd = kdm; }
			// relative joint: -getHingeAngle() - ref_theta, an atan2 window (sim/World.cpp:543-553). World-coordinate joint (dog shoulder / hip):
			// btQuaternion::getAngle() * (axis . z) of the quaternion btMatrix3x3::getRotation extracts from the link's world basis (sim/World.cpp:374-384):
			// phi while 1 + 2 cos(phi) > 0 or phi > 0, phi + 2 pi for phi in (-pi, -2 pi / 3] (w < 0 in the largest-diagonal branch)
			real theta;
			if (gm.use_world[j]) {
				theta = wrap_pi(ws.phi[j] + gm.body_theta[j]);
				if (theta <= real(-2.0943951023931954923084289221863)) theta += real(6.283185307179586476925286766559);
			} else {
				theta = wrap_pi(ws.st.q[i] + gm.ref_theta[j]) - gm.ref_theta[j];
			}
			pe = ws.st.pd_target[j] - theta;
			ve = 0 - ws.st.qd[i];
		}
		ws.kpv[i] = kp; ws.kdv[i] = kd; ws.kdm[i] = kdm; ws.perr[i] = pe; ws.verr[i] = ve;
		ws.u[i] = kp * (pe - dt * ws.st.qd[i]) + kd * ve - quirk_bias(ws, i);
	}
	LANES_END
	PROF_ADD_SINCE(ws, kProfC_PdSetup, pc_t); pc_t = PROF_NOW();
	Path::pd_solve(ws, dt);
	PROF_ADD_SINCE(ws, kProfC_PdSolve, pc_t); pc_t = PROF_NOW();
	LANES_BEGIN
	if (lane < D) { const int i = lane; ws.tau_g[i] = 0; ws.st.tau_ctrl[i] = ws.kpv[i] * (ws.perr[i] - dt * ws.st.qd[i]) + ws.kdv[i] * (ws.verr[i] - dt * ws.u[i]); }
	LANES_END
	// gravity compensation: dog :947-995 (+ BuildContactBasis :1120-1175), raptor :985-1028 (+ :1170-1240)
	const int eff0 = raptor ? int(rRightToe) : int(jToe), eff1 = raptor ? int(rLeftToe) : int(jFinger);
	const bool sup0 = raptor ? raptor_active_effector(ws, eff0) : in_contact(ws, eff0);
	const bool sup1 = raptor ? raptor_active_effector(ws, eff1) : in_contact(ws, eff1);
	if (gm.enable_grav_comp && (sup0 || sup1)) {
		LANES_BEGIN
		if (lane < D) {
			const int d = lane;
			// -Q_g: torque that counters gravity
			real tg;
			const real gy = kGravityY;
			if (d == 0) tg = 0;
			else if (d == 1) tg = -(ws.sm[0] * gy);
			else { const int l = d - 2; tg = -((ws.smx[l] - ws.sm[l] * ws.px[l]) * gy); }
			ws.tau_g[d] = tg;
			for (int e = 0; e < 2; ++e) {
				const int jid = e ? eff1 : eff0;
				real b0 = 0, b1 = 0;
				bool on_path = (d < 3) || ((ws.M.sub_mask[d - 2] >> jid) & 1u);
				if ((e ? sup1 : sup0) && on_path) {
					real pos[2]; effector_pos(ws, gm, jid, pos);
					const real fy[2] = {0, 1}, fxv[2] = {1, 0};
					b0 = jt_force(ws, d, pos, fy); b1 = jt_force(ws, d, pos, fxv);
				}
				ws.basis[d][e * 2 + 0] = b0; ws.basis[d][e * 2 + 1] = b1;
			}
		}
		LANES_END
		LANES_BEGIN
		if (lane == 0) {
			// ridge least squares on the 3 root rows: (A^T W A + 1e-4 I) x = A^T W b; W = I (dog), diag(1e-4, 1e-4, 1) (raptor)
			const real Wd[3] = {1, 1, 1}, Wr[3] = {0.0001, 0.0001, 1};
			real x[4];
			solve_ls4(ws.basis, ws.tau_g, raptor ? Wr : Wd, x);
			for (int k = 0; k < 4; ++k) ws.red[k] = x[k];
		}
		LANES_END
		LANES_BEGIN
		if (lane < D && (raptor || lane >= 3)) {   // the dog zeroes the root rows; the raptor keeps them (they are never applied)
			const int d = lane;
			real s = 0; for (int k = 0; k < 4; ++k) s += ws.basis[d][k] * ws.red[k];
			ws.st.tau_ctrl[d] += ws.tau_g[d] - s;
		}
		LANES_END
	}
	PROF_ADD_SINCE(ws, kProfC_Grav, pc_t); pc_t = PROF_NOW();
	if (raptor) {
		// ApplyStanceFeedback :933-983: the stance hip balances the swing hip torque and servoes the root pitch
		LANES_BEGIN
		if (lane == 0 && raptor_active_effector(ws, stance_joint(ws, 3))) {
			const int sh = stance_joint(ws, 0), wh = swing_joint(ws, 0);
			real hip_tau = -ws.st.tau_ctrl[wh + 2];
			real target_pitch = ws.st.params[rmpMax + ws.st.state * rspMax + rspRootPitch];
			real root_tau = gm.kp[sh] * (target_pitch - wrap_pi(ws.st.q[2])) + gm.kd[sh] * (-ws.st.qd[2]);
			hip_tau += -root_tau;
			ws.st.tau_ctrl[sh + 2] += hip_tau;
		}
		LANES_END
	}
	// virtual forces: dog :997-1029, raptor :1030-1075
	if (gm.enable_vf) {
		LANES_BEGIN
		if (lane >= 3 && lane < D) {
			const int d = lane, a = d - 2;
			if (raptor) {
				const int jid = stance_joint(ws, 3);
				if (raptor_active_effector(ws, jid)) {
					const real f[2] = {-ws.st.params[rmpForceX], -ws.st.params[rmpForceY]};
					real pos[2]; effector_pos(ws, gm, jid, pos);
					if ((ws.M.sub_mask[a] >> jid) & 1u) ws.st.tau_ctrl[d] += jt_force(ws, d, pos, f);   // toe .. hip (root excluded: d >= 3)
					if (a == swing_joint(ws, 0)) ws.st.tau_ctrl[d] += -jt_force(ws, stance_joint(ws, 0) + 2, pos, f);
				}
			} else {
				const int effs[2] = {jToe, jFinger};
				const int state = ws.st.state;
				for (int e = 0; e < 2; ++e) {
					const int jid = effs[e];
					bool valid = ((state == stBackStance || state == stExtend) && jid == jToe) || ((state == stFrontStance || state == stGather) && jid == jFinger);
					if (!(valid && in_contact(ws, jid))) continue;
					// chain: effector up to (excluding) root / torso
					bool on_chain = ((ws.M.sub_mask[a] >> jid) & 1u) && a != jRoot && a != jTorso;
					if (on_chain && jid == jFinger) on_chain = !((ws.M.sub_mask[a] >> jTorso) & 1u);  // strictly below the torso
					if (!on_chain) continue;
					real f[2];
					if (jid == jToe) { f[0] = -ws.st.params[mpBackForceX]; f[1] = -ws.st.params[mpBackForceY]; }
					else { f[0] = -ws.st.params[mpFrontForceX]; f[1] = -ws.st.params[mpFrontForceY]; }
					real pos[2]; effector_pos(ws, gm, jid, pos);
					ws.st.tau_ctrl[d] += jt_force(ws, d, pos, f);
				}
			}
		}
		LANES_END
	}
	// cSimCharacter::ApplyControlForces + cJoint::ApplyTorque clamp (sim/Joint.cpp:171-201, 257-264)
	LANES_BEGIN
	if (lane < D) {
		const int d = lane;
		real t = 0;
		if (d >= 3) { t = ws.st.tau_ctrl[d]; real lim = gm.torque_lim[d - 2]; if (fabs(t) > lim) t *= lim / fabs(t); }
		ws.st.tau[d] = t;
	}
	LANES_END
	PROF_ADD_SINCE(ws, kProfC_Tail, pc_t);
}

// dog reward, sim/DogController.cpp:594-628
template <class W>
DTRL_HD inline real calc_reward(const W& ws, const DevModel& gm)
{
	real vel_reward = 0, stumble_reward = 0;
	if (!has_fallen(ws)) {
		real cycle_time = ws.st.prev_cycle_time;
		real avg_vel = ws.st.prev_dist[0] / cycle_time;
		real vel_err = gm.target_vel_x - avg_vel;
		vel_reward = exp(-0.5 * vel_err * vel_err);
		real avg_stumble = ws.st.prev_stumble / cycle_time;
		stumble_reward = 1.0 / (1 + 10 * avg_stumble);
		if (gm.char_type == 1 && avg_vel < 0) { vel_reward = 0; stumble_reward = 0; }   // sim/RaptorController.cpp:584-588
	}
	return 0.8 * vel_reward + 0.2 * stumble_reward;
}

// cScenarioExp::NewCycleUpdate (+ cScenarioExpMACE flags): finish the running tuple, start the next one
template <class W>
DTRL_HD inline void scenario_new_cycle(W& ws, const DevModel& gm, const DevBuffers& buf, int env)
{
	LANES_BEGIN
	if (lane == 0) { ws.st.num_cycles += 1; ws.flag_new_cycle = -1; }
	LANES_END
	if (gm.scenario != kScnExp) return;
	const real* ps = buf.poli_state + static_cast<int64_t>(env) * buf.S;
	real* s0 = buf.tup_s0 + static_cast<int64_t>(env) * buf.S;
	real* ta = buf.tup_a + static_cast<int64_t>(env) * buf.A;
	LANES_BEGIN
	if (lane == 0) {
		int slot = -1;
		if (ws.st.cycle_count > 1) {  // gNumWarmupCycles = 1 (scenarios/ScenarioExp.cpp:12, 314-318)
#if defined(__HIP_DEVICE_COMPILE__)
			slot = __hip_atomic_fetch_add(buf.tuple_count, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);   // (system scope: the ring may live in page-locked host memory, -tuple_ring= host)
#else
			slot = (*buf.tuple_count)++;
#endif
			if (slot >= buf.tuple_cap) slot = -1;
		}
		ws.flag_new_cycle = slot;
		if (slot >= 0) {
			bool fail = has_fallen(ws);
			uint32_t flags = (static_cast<uint32_t>(ws.st.tuple_flags) & ~1u) | (fail ? 1u : 0u);
			buf.tuple_flags[slot] = flags;
			buf.tuple_env[slot] = env;
			buf.tuple_rows[static_cast<int64_t>(slot) * buf.W] = static_cast<float>(calc_reward(ws, gm));
		}
	}
	LANES_END
	LANES_BEGIN
	const int slot = ws.flag_new_cycle;
	if (slot >= 0) {
		float* row = buf.tuple_rows + static_cast<int64_t>(slot) * buf.W;
		for (int i = lane; i < buf.S; i += kGroup) { row[1 + i] = static_cast<float>(s0[i]); row[1 + buf.S + buf.A + i] = static_cast<float>(ps[i]); }
		for (int i = lane; i < buf.A; i += kGroup) row[1 + buf.S + i] = static_cast<float>(ta[i]);
	}
	LANES_END
	LANES_BEGIN
	for (int i = lane; i < buf.S; i += kGroup) s0[i] = ps[i];
	if (lane == 0) {
		if (gm.ctrl_type == 2) { for (int k = 0; k < gm.n_opt; ++k) ta[k] = ws.st.params[gm.opt_index[k]]; }   // cBaseControllerCacla::RecordPoliAction: the parameters alone
		else if (gm.ctrl_type == 0) { for (int k = 0; k < gm.n_actions; ++k) ta[k] = (k == ws.st.action_id) ? 1 : 0; }   // cBaseControllerQ::RecordPoliAction: one-hot
		else {
			ta[0] = ws.st.action_id;
			for (int k = 0; k < gm.n_opt; ++k) ta[1 + k] = ws.st.params[gm.opt_index[k]];
		}
		int flags = 0;
		if (gm.ctrl_type == 1) flags |= (ws.st.exp_critic ? 2 : 0) | (ws.st.exp_actor ? 4 : 0);
		if (gm.ctrl_type == 2) flags |= ws.st.is_off_policy ? 2 : 0;   // cScenarioExpCacla::RecordFlagsBeg: cCaclaTrainer::eFlagOffPolicy
		ws.st.tuple_flags = flags;
		ws.st.cycle_count += 1;
	}
	LANES_END
}

// one iteration of scenarios/ScenarioSimChar.cpp:162-173
template <class Path, class W>
DTRL_HD inline void env_step(W& ws, const DevModel& gm, const RunParams& rp, const DevBuffers& buf, const GroundRec& g, int env, real dt)
{
	const real h = dt / gm.num_sim_substeps;
	if (__builtin_expect(ws.st.pert_link >= 0, 0)) perturb_begin_step(ws, dt);   // wave-uniform (LDS)
	// UpdateWorld. The kinematics / composites / bias of the configuration at entry are already in the workspace (frame start, reset,
	// or the post-step evaluation of the previous env-step), so the first substep does not recompute them
	for (int s = 0; s < gm.num_sim_substeps; ++s) Path::substep(ws, gm, g, h, s == 0);
	kin_dyn_terms(ws);                                                      // post-step kinematics: controller's RBD terms AND the next substep's
	Path::contacts(ws, gm, g, h);                                               // cContactManager::Update
	// UpdateGround is host-side at frame boundaries (the 1 m look-ahead margin makes that equivalent; DESIGN.md "Ground")
	{ PROF_T0(); controller_update<Path>(ws, gm, rp, buf, g, env, dt); PROF_ADD(ws, kProfCtrl); }   // UpdateCharacter
	LANES_BEGIN
	if (lane == 0) {
		// cSimCharSoftFall::UpdateFallDistCheck / UpdateFallContactCheck
		ws.st.fall_dist_counter -= dt;
		if (ws.st.fall_dist_counter <= 0) {
			real dx = ws.st.q[0] - ws.st.prev_check[0], dy = ws.st.q[1] - ws.st.prev_check[1];
			if (dx * dx + dy * dy < 0.5 * 0.5) ws.st.fail_fall_dist = 1;
			ws.st.prev_check[0] = ws.st.q[0]; ws.st.prev_check[1] = ws.st.q[1]; ws.st.fall_dist_counter = 5;
		}
		ws.st.fall_contact_counter -= dt;
		if (ws.st.fall_contact_counter <= 0) {
			const real discount = 0.9, norm = (1 + 1 / (1 - discount));
			real val = check_fall_contact(ws) ? 1 : 0;
			ws.st.sum_fall_contact = val / norm + discount * ws.st.sum_fall_contact;
			ws.st.fall_contact_counter = 0.1;
		}
		ws.st.time += dt;
		ws.flag_misc = is_new_cycle(ws) ? 1 : 0;
	}
	LANES_END
	if (__builtin_expect(ws.flag_misc != 0, 0)) scenario_new_cycle(ws, gm, buf, env);   // PostSubstepUpdate
}

// cSimCharacter::Reset + controller Reset; init=true additionally follows cScenarioSimChar::Init ordering
template <class W>
DTRL_HD inline void reset_env(W& ws, const DevModel& gm, const RunParams& rp, const DevBuffers& buf, const GroundRec& g, int env, bool init)
{
	LANES_BEGIN
	if (lane < gm.D) { ws.st.q[lane] = gm.pose0[lane]; ws.st.qd[lane] = gm.vel0[lane]; ws.st.tau[lane] = 0; ws.st.tau_ctrl[lane] = 0; }
	if (init && lane < gm.L) ws.st.pd_target[lane] = gm.target_theta[lane];
	LANES_END
	forward_kinematics(ws);
	LANES_BEGIN
	if (lane == 0) {
		Rng rng = make_rng(rp, env, &ws.st.rng_ctr);
		const int num_frags = gm.has_net ? buf.net.n_frags : 0;
		ws.st.exp_actor = 0; ws.st.exp_critic = 0;
		int id; real prm[kMaxP];
		build_base_action(gm, num_frags, gm.default_action, rng, &id, prm);
		apply_action(ws, id, prm, gm.P);
		ws.st.state = 0; ws.st.phase = 0; ws.st.first_cycle = 1; ws.st.is_off_policy = 0;
		ws.st.sample_origin[0] = 0; ws.st.sample_origin[1] = 0;
		ws.st.prev_cycle_time = 0; ws.st.prev_dist[0] = 0; ws.st.prev_dist[1] = 0; ws.st.curr_cycle_time = 0;
		ws.st.prev_stumble = 0; ws.st.curr_stumble = 0;
		ws.st.cmd_action = -1;
		ws.st.pd_active_bits = 0xffffffffu; ws.st.stance = 0;
		if (gm.char_type == 1) set_stance(ws, 0);   // ResetParams + mImpPDCtrl.Reset + SetStance(gDefaultStance)
		calc_com(ws, ws.st.prev_com);
		ws.st.fall_dist_counter = 5; ws.st.prev_check[0] = ws.st.q[0]; ws.st.prev_check[1] = ws.st.q[1]; ws.st.fail_fall_dist = 0;
		ws.st.fall_contact_counter = 0.1; ws.st.sum_fall_contact = 0;
		ws.st.contact_bits = 0;
		ws.st.ws_R = 0;   // cWorld::Reset / a re-created character: no persistent contact points
		ws.st.time = 0;
		ws.st.pert_link = -1; ws.st.pert_on = 0;   // cWorld::Reset clears the perturbation manager (sim/World.cpp:76-82)
		// cScenarioSimChar::InitCharacterPos
		if (gm.valid_init_pos_x) ws.st.q[0] = gm.init_pos_x;
		ws.st.q[1] += sample_ground(g, ws.st.q[0], nullptr, nullptr, nullptr, nullptr);
		if (init) { real com[2]; calc_com(ws, com); ws.st.prev_com[0] = com[0]; ws.st.prev_com[1] = com[1]; }
		else ws.st.num_resets += 1;
		if (gm.scenario == kScnExp) { ws.st.cycle_count = 0; ws.st.cmd_action = rng.rand_int(0, gm.n_actions); }
		if (gm.scenario == kScnPoliEval) ws.st.pos_start_x = ws.st.q[0];
		ws.st.do_reset = 0; ws.st.do_init = 0; ws.st.need_reset = 0;
	}
	LANES_END
	forward_kinematics(ws);
}

// end-of-frame scenario logic: cScenarioExp::Update / cScenarioPoliEval::Update tail (fall -> NewCycleUpdate, request reset)
template <class W>
DTRL_HD inline void frame_end(W& ws, const DevModel& gm, const DevBuffers& buf, int env)
{
	LANES_BEGIN
	if (lane == 0) {
		int mode = 0;
		if (gm.scenario == kScnExp) { if (!is_new_cycle(ws) && has_fallen(ws)) mode = 1; }
		else if (gm.scenario == kScnPoliEval) {
			if (has_fallen(ws)) {
				if (ws.st.num_cycles >= 1) {   // IsValidCycle(): mCycleCount >= gNumWarmupCycles (= 1), scenarios/ScenarioPoliEval.cpp:7, 406-410
					real dist = ws.st.q[0] - ws.st.pos_start_x;
					ws.st.avg_dist = (ws.st.num_episodes * ws.st.avg_dist + dist) / (ws.st.num_episodes + 1.0);
					ws.st.num_episodes += 1;
					ws.red[7] = dist; mode = 3;   // episode recorded: the host appends dist to the batch's dist log (mDistLog)
				}
				else mode = 2;
			}
		}
		ws.flag_misc = mode;
	}
	LANES_END
	if (ws.flag_misc == 1) scenario_new_cycle(ws, gm, buf, env);
	LANES_BEGIN
	if (lane == 0 && ws.flag_misc != 0) ws.st.need_reset = (ws.flag_misc == 3) ? 3 : 1;
	LANES_END
}

template <class W>
DTRL_HD inline void load_hot_model(W& ws, const DevModel& gm)
{
	LANES_BEGIN
	if (lane == 0) { ws.M.L = gm.L; ws.M.D = gm.D; ws.M.char_type = gm.char_type; ws.M.n_pairs = gm.n_pairs; ws.M.n_cpairs = gm.link_contacts ? gm.n_cpairs : 0; ws.M.warm_start = gm.warm_start; }
	for (int e = lane; e < gm.n_pairs; e += kGroup) { ws.M.pair_l[e] = gm.pair_l[e]; ws.M.pair_k[e] = gm.pair_k[e]; }
	if (lane < gm.n_cpairs) { ws.M.cp_a[lane] = gm.cp_a[lane]; ws.M.cp_b[lane] = gm.cp_b[lane]; }
	if (lane < gm.L) { ws.M.cp_half[lane][0] = gm.cp_half[lane][0]; ws.M.cp_half[lane][1] = gm.cp_half[lane][1]; }
	if (lane < 2) ws.M.cp_root_bt[lane] = gm.cp_root_bt[lane];
	if (lane < gm.L) {
		const int j = lane;
		ws.M.parent[j] = gm.parent[j]; ws.M.depth[j] = gm.depth[j]; ws.M.col[j] = gm.col[j];
		for (int k = 0; k < kMaxDepth; ++k) ws.M.path[j][k] = gm.path[j][k];
		ws.M.sub_mask[j] = gm.sub_mask[j]; ws.M.anc_mask[j] = gm.anc_mask[j];
		ws.M.attach[j][0] = gm.attach[j][0]; ws.M.attach[j][1] = gm.attach[j][1];
		ws.M.lim_lo[j] = gm.lim_lo[j]; ws.M.lim_hi[j] = gm.lim_hi[j];
		ws.M.body_attach[j][0] = gm.body_attach[j][0]; ws.M.body_attach[j][1] = gm.body_attach[j][1];
		ws.M.mass[j] = gm.mass[j]; ws.M.inertia[j] = gm.inertia[j]; ws.M.sub_mass[j] = gm.sub_mass[j];
	}
	LANES_END
}

// the whole per-env frame: load -> (reset) -> n_steps env-steps -> frame-end logic -> store
#ifndef DTRL_COST_MODE
#define DTRL_COST_MODE 1   // (round 6: the last substep's rows; +1.0 % dog, +0.3 % raptor in 6 of 6 same-box pairs, results unchanged bit for bit -- profiles/r06_launch_order_ab.txt)
#endif
template <class Path, class W>
DTRL_HD inline void env_frame(W& ws, const DevModel& gm, const RunParams& rp, const DevBuffers& buf, int env, int n_steps, real dt, bool do_frame_end)
{
#if defined(__HIP_DEVICE_COMPILE__) && defined(DTRL_PROFILE)
	const unsigned long long prof_frame_t0 = __builtin_readcyclecounter();
	if (threadIdx.x < kProfMax) ws.prof[threadIdx.x] = 0;
	env_sync();
#endif
	load_hot_model(ws, gm);
	LANES_BEGIN
	if (lane == 0) { ws.cost = 0; ws.n_pts_active = -1; }
	LANES_END
	{
		const uint64_t* src = reinterpret_cast<const uint64_t*>(&buf.st[env]);
		uint64_t* dst = reinterpret_cast<uint64_t*>(&ws.st);
		LANES_BEGIN
		for (int i = lane; i < static_cast<int>(sizeof(EnvState) / 8); i += kGroup) dst[i] = src[i];
		LANES_END
	}
	if (buf.stage_slot != nullptr) {
		const int slot = buf.stage_slot[env] - 1;   // wave-uniform
		if (__builtin_expect(slot >= 0, 0)) {
			static_assert(sizeof(GroundRec) % 16 == 0, "GroundRec is copied as 16-byte words");
			const F4* src = reinterpret_cast<const F4*>(&buf.gr_stage[slot]);
			F4* dst = reinterpret_cast<F4*>(&buf.gr[env]);
			LANES_BEGIN
			for (int i = lane; i < static_cast<int>(sizeof(GroundRec) / 16); i += kGroup) dst[i] = src[i];
			if (lane == 0) buf.stage_slot[env] = 0;
			LANES_END
		}
	}
	const GroundRec& g = buf.gr[env];
	if (buf.reset_listed == 2 && ws.st.need_reset == 0 && ws.st.do_init == 0) return;   // wave-uniform: this env did not fall, nothing to do (state, status untouched)
	if (__builtin_expect(ws.st.do_init != 0, 0)) reset_env(ws, gm, rp, buf, g, env, true);
	else if (__builtin_expect(ws.st.do_reset != 0 || buf.reset_listed != 0, 0)) reset_env(ws, gm, rp, buf, g, env, false);
	else forward_kinematics(ws);
	for (int s = 0; s < n_steps; ++s) env_step<Path>(ws, gm, rp, buf, g, env, dt);
	if (do_frame_end) frame_end(ws, gm, buf, env);
	{
		uint64_t* dst = reinterpret_cast<uint64_t*>(&buf.st[env]);
		const uint64_t* src = reinterpret_cast<const uint64_t*>(&ws.st);
		LANES_BEGIN
		for (int i = lane; i < static_cast<int>(sizeof(EnvState) / 8); i += kGroup) dst[i] = src[i];
		if (lane == 0) { buf.status[env].root_x = ws.st.q[0]; buf.status[env].need_reset = ws.st.need_reset; buf.status[env].episode_dist = (ws.st.need_reset & 2) ? ws.red[7] : 0.0;
			// work estimate for the NEXT frame (the host launches the costliest envs first): the constraint-row part is persistent (a stumbling
			// character stays expensive), a policy forward is not -- it is predicted from the gait clock: the running cycle will reach the
			// length of the previous one within the next frame (one forward ~ 400 row-substep units, tools/gpu_sections.py)
			const real frame_t = n_steps * dt;
			const bool forward_due = gm.has_net != 0 && (ws.st.first_cycle != 0 || ws.st.curr_cycle_time + frame_t >= 0.9 * ws.st.prev_cycle_time);
			// DTRL_COST_MODE (A/B, profiles/r06_launch_order_ab.txt): 0 = the frame's row sum (shipped until round 5: finds 43 % of the next frame's slowest 5 %, rank
			// correlation 0.49); 1 = the row count of the frame's LAST substep x the substeps of a frame (50 %, 0.66: a character that just went down is expensive for the
			// WHOLE next frame, one that just got up is not); 2 = the mean of both
#if DTRL_COST_MODE == 1
			const int row_cost = n_steps * gm.num_sim_substeps * (8 + ws.st.ws_R);
#elif DTRL_COST_MODE == 2
			const int row_cost = (ws.cost + n_steps * gm.num_sim_substeps * (8 + ws.st.ws_R)) / 2;
#else
			const int row_cost = ws.cost;
#endif
			buf.status[env].cost = row_cost + (forward_due ? 400 : 0); }
		LANES_END
	}
#if defined(__HIP_DEVICE_COMPILE__) && defined(DTRL_PROFILE)
	if (buf.prof && n_steps > 0) {
		if (threadIdx.x == 0) ws.prof[kProfTotal] += __builtin_readcyclecounter() - prof_frame_t0;
		env_sync();
		if (threadIdx.x < kProfMax) buf.prof[static_cast<int64_t>(env) * kProfMax + threadIdx.x] += ws.prof[threadIdx.x];
	}
#endif
}

}  // namespace dtrl
