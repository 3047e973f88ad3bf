// Checks the operand layout and the accumulation order of v_mfma_f64_16x16x4_f64 on gfx950 against a sequential fma loop
// (used by the subtree-sum phase of kin_dyn_terms). Build: hipcc --offload-arch=gfx950 -O3 -o mfma_f64_check mfma_f64_check.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
typedef double v4d __attribute__((ext_vector_type(4)));
// S[j][q] = sum_k M[j][k] V[k][q], j < 32, k < 24, q < 16
__global__ void k(const double* M, const double* V, double* S_mfma, double* S_seq)
{
	const int l = threadIdx.x;
	for (int t = 0; t < 2; ++t) {
		v4d acc = {0, 0, 0, 0};
		for (int s = 0; s < 6; ++s) {
			const double a = M[(16 * t + l % 16) * 24 + 4 * s + l / 16];   // A[i = l % 16][k = l / 16]
			const double b = V[(4 * s + l / 16) * 16 + l % 16];            // B[k = l / 16][j = l % 16]
			acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
		}
		for (int r = 0; r < 4; ++r) S_mfma[(16 * t + 4 * r + l / 16) * 16 + l % 16] = acc[r];   // D[i = 4 r + l / 16][j = l % 16]
	}
	for (int e = l; e < 32 * 16; e += 64) {
		const int j = e / 16, q = e % 16;
		double s = 0;
		for (int kk = 0; kk < 24; ++kk) s = __builtin_fma(M[j * 24 + kk], V[kk * 16 + q], s);
		S_seq[e] = s;
	}
}
int main()
{
	std::vector<double> M(32 * 24), V(24 * 16), A(32 * 16), B(32 * 16);
	unsigned x = 12345;
	auto rnd = [&]() { x = x * 1664525u + 1013904223u; return (x >> 8) / 16777216.0; };
	for (auto& m : M) m = rnd() < 0.4 ? 1.0 : 0.0;
	for (auto& v : V) v = (rnd() - 0.5) * 1000.0 * rnd();
	double *dM, *dV, *dA, *dB;
	hipMalloc(&dM, M.size() * 8); hipMalloc(&dV, V.size() * 8); hipMalloc(&dA, A.size() * 8); hipMalloc(&dB, B.size() * 8);
	hipMemcpy(dM, M.data(), M.size() * 8, hipMemcpyHostToDevice); hipMemcpy(dV, V.data(), V.size() * 8, hipMemcpyHostToDevice);
	k<<<1, 64>>>(dM, dV, dA, dB); hipDeviceSynchronize();
	hipMemcpy(A.data(), dA, A.size() * 8, hipMemcpyDeviceToHost); hipMemcpy(B.data(), dB, B.size() * 8, hipMemcpyDeviceToHost);
	int bitdiff = 0; double maxrel = 0;
	for (size_t i = 0; i < A.size(); ++i) { if (memcmp(&A[i], &B[i], 8)) ++bitdiff; double d = A[i] - B[i]; if (d < 0) d = -d; double sc = B[i] < 0 ? -B[i] : B[i]; if (sc > 0 && d / sc > maxrel) maxrel = d / sc; }
	printf("entries %zu, bitwise different %d, max relative difference %.3e\n", A.size(), bitdiff, maxrel);
	printf("sample: mfma %.17g seq %.17g\n", A[37], B[37]);
	return 0;
}
