// ORACLE (test infrastructure, NOT product code).
// CPU fp64 restatement of the reference's small-matrix / spatial-algebra helpers.
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use anything under oracle/.
//
// Follows (file:line relative to /root/reference):
//   util/MathUtil.cpp:67-149   TranslateMat / RotateMat / CrossMat / InvRigidMat / RotMatToAxisAngle
//   sim/SpAlg.cpp:50-345       6-D spatial vectors, tSpTrans = [E | r], ApplyTransM/F, inverses, CompTrans
// Eigen is not available here, so fixed-size structs stand in for tVector/tMatrix/tSpVec/tSpMat.
#pragma once
#include <cmath>
#include <cstring>

namespace orc {

struct V3 { double x = 0, y = 0, z = 0; };
inline V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 operator-(V3 a) { return {-a.x, -a.y, -a.z}; }
inline V3 operator*(double s, V3 a) { return {s * a.x, s * a.y, s * a.z}; }
inline double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }

struct M3 {
	double m[3][3];
	static M3 Identity() { M3 r; std::memset(r.m, 0, sizeof(r.m)); r.m[0][0] = r.m[1][1] = r.m[2][2] = 1; return r; }
	static M3 Zero() { M3 r; std::memset(r.m, 0, sizeof(r.m)); return r; }
};
inline V3 operator*(const M3& A, V3 v)
{
	return {A.m[0][0] * v.x + A.m[0][1] * v.y + A.m[0][2] * v.z,
			A.m[1][0] * v.x + A.m[1][1] * v.y + A.m[1][2] * v.z,
			A.m[2][0] * v.x + A.m[2][1] * v.y + A.m[2][2] * v.z};
}
inline M3 operator*(const M3& A, const M3& B)
{
	M3 C = M3::Zero();
	for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) for (int k = 0; k < 3; ++k) C.m[i][j] += A.m[i][k] * B.m[k][j];
	return C;
}
inline M3 transpose(const M3& A)
{
	M3 C;
	for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) C.m[i][j] = A.m[j][i];
	return C;
}
// util/MathUtil.cpp:109-117 (3x3 block)
inline M3 CrossMat(V3 a)
{
	M3 r = M3::Zero();
	r.m[0][1] = -a.z; r.m[0][2] = a.y;
	r.m[1][0] = a.z;  r.m[1][2] = -a.x;
	r.m[2][0] = -a.y; r.m[2][1] = a.x;
	return r;
}

// rigid 4x4 (rotation R, translation t): x_parent = R x_child + t
struct M4 {
	M3 R = M3::Identity();
	V3 t;
};
inline M4 operator*(const M4& A, const M4& B) { M4 C; C.R = A.R * B.R; C.t = A.R * B.t + A.t; return C; }
inline V3 xform(const M4& A, V3 p) { return A.R * p + A.t; }
// util/MathUtil.cpp:67-74
inline M4 TranslateMat(V3 t) { M4 r; r.t = t; return r; }
// util/MathUtil.cpp:91-107 specialised to axis = +z (the only axis the reference ever passes)
inline M4 RotateMatZ(double theta)
{
	M4 r;
	double c = std::cos(theta), s = std::sin(theta);
	r.R.m[0][0] = c; r.R.m[0][1] = -s;
	r.R.m[1][0] = s; r.R.m[1][1] = c;
	return r;
}
// util/MathUtil.cpp:119-126
inline M4 InvRigidMat(const M4& A) { M4 r; r.R = transpose(A.R); r.t = -(r.R * A.t); return r; }
// util/MathUtil.cpp:128-149; returns theta in [0, pi] and axis
inline void RotMatToAxisAngle(const M3& R, V3& axis, double& theta)
{
	double c = (R.m[0][0] + R.m[1][1] + R.m[2][2] - 1) * 0.5;
	c = c < -1 ? -1 : (c > 1 ? 1 : c);
	theta = std::acos(c);
	if (std::fabs(theta) < 0.00001) { axis = {0, 0, 1}; }
	else {
		double m21 = R.m[2][1] - R.m[1][2];
		double m02 = R.m[0][2] - R.m[2][0];
		double m10 = R.m[1][0] - R.m[0][1];
		double d = std::sqrt(m21 * m21 + m02 * m02 + m10 * m10);
		axis = {m21 / d, m02 / d, m10 / d};
	}
}

// ---- spatial algebra (sim/SpAlg.cpp) -------------------------------------------------------
struct SV { V3 o, v; };  // tSpVec = [omega; v]
inline SV operator+(SV a, SV b) { return {a.o + b.o, a.v + b.v}; }
inline SV operator*(double s, SV a) { return {s * a.o, s * a.v}; }
inline double dot(SV a, SV b) { return dot(a.o, b.o) + dot(a.v, b.v); }
inline double get6(const SV& a, int i) { return i < 3 ? (&a.o.x)[i] : (&a.v.x)[i - 3]; }
inline void set6(SV& a, int i, double x) { if (i < 3) (&a.o.x)[i] = x; else (&a.v.x)[i - 3] = x; }

struct SpTrans { M3 E = M3::Identity(); V3 r; };  // tSpTrans

// sim/SpAlg.cpp:155-169
inline SpTrans MatToTrans(const M4& mat) { SpTrans X; X.E = mat.R; X.r = -(transpose(mat.R) * mat.t); return X; }
// sim/SpAlg.cpp:171-179
inline M4 TransToMat(const SpTrans& X) { M4 m; m.R = X.E; m.t = -(X.E * X.r); return m; }
// sim/SpAlg.cpp:207-213
inline SpTrans InvTrans(const SpTrans& X) { SpTrans Y; Y.E = transpose(X.E); Y.r = -(X.E * X.r); return Y; }
inline SpTrans BuildTrans(V3 r) { SpTrans X; X.r = r; return X; }
// sim/SpAlg.cpp:233-243
inline SV ApplyTransM(const SpTrans& X, SV sv) { SV n; n.o = X.E * sv.o; n.v = X.E * (sv.v - cross(X.r, sv.o)); return n; }
// sim/SpAlg.cpp:245-256
inline SV ApplyTransF(const SpTrans& X, SV sv) { SV n; n.o = X.E * (sv.o - cross(X.r, sv.v)); n.v = X.E * sv.v; return n; }
// sim/SpAlg.cpp:285-295
inline SV ApplyInvTransM(const SpTrans& X, SV sv)
{
	M3 Et = transpose(X.E);
	SV n; n.o = Et * sv.o; n.v = Et * sv.v + cross(X.r, Et * sv.o); return n;
}
// sim/SpAlg.cpp:297-308
inline SV ApplyInvTransF(const SpTrans& X, SV sv)
{
	M3 Et = transpose(X.E);
	SV n; n.o = Et * sv.o + cross(X.r, Et * sv.v); n.v = Et * sv.v; return n;
}
// sim/SpAlg.cpp:336-345
inline SpTrans CompTrans(const SpTrans& X0, const SpTrans& X1)
{
	SpTrans X; X.E = X0.E * X1.E; X.r = X1.r + transpose(X1.E) * X0.r; return X;
}
// sim/SpAlg.cpp:50-60
inline SV CrossM(SV sv, SV m) { SV r; r.o = cross(sv.o, m.o); r.v = cross(sv.v, m.o) + cross(sv.o, m.v); return r; }
// sim/SpAlg.cpp:75-86
inline SV CrossF(SV sv, SV f) { SV r; r.o = cross(sv.o, f.o) + cross(sv.v, f.v); r.v = cross(sv.o, f.v); return r; }

struct SM { double m[6][6]; static SM Zero() { SM r; std::memset(r.m, 0, sizeof(r.m)); return r; } };
inline SM operator*(const SM& A, const SM& B)
{
	SM C = SM::Zero();
	for (int i = 0; i < 6; ++i) for (int k = 0; k < 6; ++k) { double a = A.m[i][k]; if (a == 0) continue; for (int j = 0; j < 6; ++j) C.m[i][j] += a * B.m[k][j]; }
	return C;
}
inline SV operator*(const SM& A, SV x)
{
	SV y;
	for (int i = 0; i < 6; ++i) { double s = 0; for (int j = 0; j < 6; ++j) s += A.m[i][j] * get6(x, j); set6(y, i, s); }
	return y;
}
inline void AddTo(SM& A, const SM& B) { for (int i = 0; i < 6; ++i) for (int j = 0; j < 6; ++j) A.m[i][j] += B.m[i][j]; }
// sim/SpAlg.cpp:181-192
inline SM BuildSpatialMatM(const SpTrans& X)
{
	SM m = SM::Zero();
	M3 Er = X.E * CrossMat(X.r);
	for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { m.m[i][j] = X.E.m[i][j]; m.m[3 + i][3 + j] = X.E.m[i][j]; m.m[3 + i][j] = -Er.m[i][j]; }
	return m;
}
// sim/SpAlg.cpp:194-205
inline SM BuildSpatialMatF(const SpTrans& X)
{
	SM m = SM::Zero();
	M3 Er = X.E * CrossMat(X.r);
	for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { m.m[i][j] = X.E.m[i][j]; m.m[3 + i][3 + j] = X.E.m[i][j]; m.m[i][3 + j] = -Er.m[i][j]; }
	return m;
}

// dense symmetric solve A x = b via LDL^T without pivoting (Eigen's ldlt() pivots, which only changes
// rounding for the SPD matrices on this path: H + dt*Kd and H). n <= 32.
inline bool SolveLDLT(int n, const double* A, int lda, const double* b, double* x)
{
	double L[32][32]; double d[32];
	for (int j = 0; j < n; ++j) {
		double dj = A[j * lda + j];
		for (int k = 0; k < j; ++k) dj -= L[j][k] * L[j][k] * d[k];
		d[j] = dj;
		if (dj == 0) return false;
		for (int i = j + 1; i < n; ++i) {
			double s = A[i * lda + j];
			for (int k = 0; k < j; ++k) s -= L[i][k] * L[j][k] * d[k];
			L[i][j] = s / dj;
		}
	}
	double y[32];
	for (int i = 0; i < n; ++i) { double s = b[i]; for (int k = 0; k < i; ++k) s -= L[i][k] * y[k]; y[i] = s; }
	for (int i = 0; i < n; ++i) y[i] /= d[i];
	for (int i = n - 1; i >= 0; --i) { double s = y[i]; for (int k = i + 1; k < n; ++k) s -= L[k][i] * x[k]; x[i] = s; }
	return true;
}

// general small solve (partial-pivot Gaussian elimination); stands in for Eigen's HouseholderQR solve of
// the 4x4 ridge system in sim/DogController.cpp:976-979 (well-conditioned: AtA + 1e-4 I).
inline void SolveGE(int n, const double* A_in, const double* b_in, double* x)
{
	double A[8][9];
	for (int i = 0; i < n; ++i) { for (int j = 0; j < n; ++j) A[i][j] = A_in[i * n + j]; A[i][n] = b_in[i]; }
	for (int c = 0; c < n; ++c) {
		int p = c; for (int r = c + 1; r < n; ++r) if (std::fabs(A[r][c]) > std::fabs(A[p][c])) p = r;
		if (p != c) for (int j = 0; j <= n; ++j) { double t = A[c][j]; A[c][j] = A[p][j]; A[p][j] = t; }
		for (int r = c + 1; r < n; ++r) { double f = A[r][c] / A[c][c]; for (int j = c; j <= n; ++j) A[r][j] -= f * A[c][j]; }
	}
	for (int i = n - 1; i >= 0; --i) { double s = A[i][n]; for (int j = i + 1; j < n; ++j) s -= A[i][j] * x[j]; x[i] = s / A[i][i]; }
}

}  // namespace orc
