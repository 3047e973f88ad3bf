// Microbenchmark: the row-parallel algebra of an env-step -- sparse leaf-first U D U^T of the dog's 23 x 23 mass matrix plus the two
// triangular substitutions, lane i holding row i in registers, pivots broadcast with v_readlane (dtrl_kernel_fast.h) -- in two forms:
//   A  today's form: one env per wavefront, fp64 (v_fma_f64), 23 of 64 lanes busy;
//   B  TWO envs per wavefront packed into the two halves of a 64-bit register: float2 entries, v_pk_fma_f32, and ONE readlane pair broadcasts BOTH envs'
//      pivot (the broadcast moves 64 bits either way), so the instruction stream is the same length and serves two envs.
// (Two envs in the two HALF-WAVES -- the form DESIGN 3 analysed and rejected -- doubles the broadcasts instead; the packed form does not.)
// Reported: ticks per (factorisation + forward + back substitution) per WAVE and per ENV, at 1 and 2 waves per SIMD, plus the fp32 form's error against
// the fp64 one on a well-conditioned SPD matrix with the dog's sparsity. Build: hipcc --offload-arch=gfx950 -O3 -o pk_f32_ltdl pk_f32_ltdl.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>

constexpr int D = 23;
// DoF-level parents of the dog (data/characters/dog.txt: joints root, spine0-3, torso, neck0-1, head, tail0-3, shoulder, elbow, wrist, finger, hip, knee,
// ankle, toe with parents [-1,0,1,2,3,4,5,6,7,0,9,10,11,5,13,14,15,0,17,18,19]; DoFs 0..2 = planar root x, y, theta chained, joint j -> DoF j + 2)
__host__ __device__ constexpr int PD(int k)
{
	constexpr int pj[21] = {-1, 0, 1, 2, 3, 4, 5, 6, 7, 0, 9, 10, 11, 5, 13, 14, 15, 0, 17, 18, 19};
	return k == 0 ? -1 : (k <= 2 ? k - 1 : pj[k - 2] + 2);
}

typedef float f2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ double bcast(double v, int src)
{
	return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), src), __builtin_amdgcn_readlane(__double2loint(v), src));
}
__device__ __forceinline__ f2 bcast(f2 v, int src)
{
	f2 r;
	r.x = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v.x), src));
	r.y = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v.y), src));
	return r;
}
__device__ __forceinline__ double fnma(double a, double b, double c) { return __builtin_fma(-a, b, c); }
__device__ __forceinline__ f2 fnma(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(-a, b, c); }   // v_pk_fma_f32
__device__ __forceinline__ double recip(double d) { return 1.0 / d; }
__device__ __forceinline__ f2 recip(f2 d) { f2 r; r.x = 1.0f / d.x; r.y = 1.0f / d.y; return r; }

// ancestors I of pivot K, walked at compile time
template <int K, int I> struct Anc {
	template <class T> static __device__ __forceinline__ void elim(T (&h)[D], T inv)
	{
		const T f = bcast(h[K], I) * inv;          // U(I, K) = H[I][K] / d_K
		h[I] = fnma(h[K], f, h[I]);                // every lane j: H[j][I] -= H[j][K] U(I, K)  (only ancestors of K hold a non-zero H[j][K])
		Anc<K, PD(I)>::elim(h, inv);
	}
};
template <int K> struct Anc<K, -1> { template <class T> static __device__ __forceinline__ void elim(T (&)[D], T) {} };

template <int K> struct Piv {
	template <class T> static __device__ __forceinline__ void run(T (&h)[D], T (&dinv)[D])
	{
		const T inv = recip(bcast(h[K], K));
		dinv[K] = inv;
		Anc<K, PD(K)>::elim(h, inv);
		h[K] = h[K] * inv;                           // column K of U (the pivot lane's own entry becomes 1)
		Piv<K - 1>::run(h, dinv);
	}
};
template <> struct Piv<-1> { template <class T> static __device__ __forceinline__ void run(T (&)[D], T (&)[D]) {} };

// forward substitution y = U^-1 b (leaf first: b_i -= U(i, k) b_k for the ancestors i of k -- all of them in one FMA), scale by D^-1
template <int K> struct Fwd {
	template <class T> static __device__ __forceinline__ void run(const T (&h)[D], T& b, int lane)
	{
		const T bk = bcast(b, K);
		const T u = lane == K ? T(0) : h[K];
		b = fnma(u, bk, b);
		Fwd<K - 1>::run(h, b, lane);
	}
};
template <> struct Fwd<-1> { template <class T> static __device__ __forceinline__ void run(const T (&)[D], T&, int) {} };
// back substitution x = U^-T y (root first: x_k -= U(i, k) x_i over the ancestors i of k; lane k holds column k of U in ut[])
template <int I> struct Bwd {
	template <class T> static __device__ __forceinline__ void run(const T (&ut)[D], T& x)
	{
		const T xi = bcast(x, I);
		x = fnma(ut[I], xi, x);
		Bwd<I + 1>::run(ut, x);
	}
};
template <> struct Bwd<D> { template <class T> static __device__ __forceinline__ void run(const T (&)[D], T&) {} };

template <class T> struct Scalar;
template <> struct Scalar<double> { static __device__ double make(double a, double) { return a; } static __device__ double lo(double v) { return v; } static __device__ double hi(double v) { return v; } };
template <> struct Scalar<f2> { static __device__ f2 make(double a, double b) { f2 r; r.x = static_cast<float>(a); r.y = static_cast<float>(b); return r; }
	static __device__ double lo(f2 v) { return v.x; } static __device__ double hi(f2 v) { return v.y; } };

// H = SPD with the tree's sparsity: H[i][j] != 0 iff one of i, j is an ancestor of the other. Built as sum over DoFs k of w_k a_k a_k^T with a_k supported on
// the ancestors-or-self of k (what a kinematic tree's mass matrix is), two variants (env A / env B) through the seed.
__device__ double hentry(int i, int j, double seed)
{
	double s = 0;
	for (int k = 0; k < D; ++k) {
		bool ai = false, aj = false;
		for (int a = k; a >= 0; a = PD(a)) { ai |= (a == i); aj |= (a == j); }
		if (ai && aj) s += (1.0 + 0.37 * k + seed) * (1.0 + 0.05 * ((i * 7 + k * 3) % 5)) * (1.0 + 0.05 * ((j * 7 + k * 3) % 5));
	}
	return s;
}

constexpr int N_IT = 64;
template <class T>
__global__ void __launch_bounds__(64) k_ltdl(double* out, unsigned long long* cyc, double seed)
{
	const int lane = static_cast<int>(threadIdx.x);
	T h0[D], h[D], dinv[D], ut[D];
#pragma unroll
	for (int j = 0; j < D; ++j) h0[j] = lane < D ? Scalar<T>::make(hentry(lane, j, seed), hentry(lane, j, seed + 0.5)) : Scalar<T>::make(j == 0 ? 1 : 0, j == 0 ? 1 : 0);
	T acc = Scalar<T>::make(0, 0), x = Scalar<T>::make(0, 0);
	__shared__ T tr[D][D + 1];
	const unsigned long long t0 = __builtin_readcyclecounter();
	for (int it = 0; it < N_IT; ++it) {
#pragma unroll
		for (int j = 0; j < D; ++j) h[j] = h0[j] + acc * static_cast<float>(1e-9);   // a dependence on the previous round, as the substep loop has
		Piv<D - 1>::run(h, dinv);
		// the transposed copy of U goes once through LDS (as in the kernel)
		if (lane < D) {
#pragma unroll
			for (int j = 0; j < D; ++j) tr[lane][j] = h[j];
		}
		__syncthreads();
#pragma unroll
		for (int j = 0; j < D; ++j) ut[j] = (lane < D && j != lane) ? tr[j][lane] : Scalar<T>::make(0, 0);
		__syncthreads();
		T b = Scalar<T>::make(1.0 + lane, 2.0 - 0.1 * lane);
		Fwd<D - 1>::run(h, b, lane);
		T dl = Scalar<T>::make(1, 1);
#pragma unroll
		for (int j = 0; j < D; ++j) if (lane == j) dl = dinv[j];
		x = b * dl;
		Bwd<0>::run(ut, x);
		acc = acc + x;
	}
	const unsigned long long t1 = __builtin_readcyclecounter();
	if (lane < D) { out[(blockIdx.x * 64 + lane) * 2] = Scalar<T>::lo(x); out[(blockIdx.x * 64 + lane) * 2 + 1] = Scalar<T>::hi(x); }
	if (lane == 0) cyc[blockIdx.x] = t1 - t0;
}

template <class T>
double run(const char* name, int blocks, std::vector<double>* sol)
{
	double* out; unsigned long long* cyc;
	hipMalloc(&out, sizeof(double) * 128 * blocks); hipMalloc(&cyc, sizeof(unsigned long long) * blocks);
	hipMemset(out, 0, sizeof(double) * 128 * blocks);
	k_ltdl<T><<<blocks, 64>>>(out, cyc, 0.25); hipDeviceSynchronize();
	k_ltdl<T><<<blocks, 64>>>(out, cyc, 0.25); hipDeviceSynchronize();
	std::vector<unsigned long long> hc(blocks);
	hipMemcpy(hc.data(), cyc, sizeof(unsigned long long) * blocks, hipMemcpyDeviceToHost);
	double avg = 0; for (auto v : hc) avg += v; avg /= blocks;
	const int envs = sizeof(T) == 8 && std::is_same<T, f2>::value ? 2 : 1;
	printf("%-40s blocks=%5d: %8.0f ticks per wave per (factorise + 2 substitutions), %8.0f per env\n", name, blocks, avg / N_IT, avg / N_IT / envs);
	if (sol) { sol->resize(128); hipMemcpy(sol->data(), out, sizeof(double) * 128, hipMemcpyDeviceToHost); }
	hipFree(out); hipFree(cyc);
	return avg / N_IT / envs;
}

int main()
{
	std::vector<double> s64, s32;
	for (int blocks : {256, 1024, 2048}) {   // 1 wave per CU, 1 wave per SIMD, 2 waves per SIMD
		const double a = run<double>("A: fp64, one env per wave", blocks, &s64);
		const double b = run<f2>("B: packed fp32, two envs per wave", blocks, &s32);
		printf("   -> per-env cost ratio B / A = %.3f\n", b / a);
	}
	// accuracy of the fp32 form: env A of the packed run solved the same system as the fp64 run
	double err = 0, mag = 0;
	for (int i = 0; i < D; ++i) { err = std::fmax(err, std::fabs(s32[2 * i] - s64[2 * i])); mag = std::fmax(mag, std::fabs(s64[2 * i])); }
	printf("fp32 vs fp64 solution of H x = b (dog sparsity, cond ~1e3): max |dx| = %.3e, max |x| = %.3e, relative %.2e\n", err, mag, err / mag);
	return 0;
}
