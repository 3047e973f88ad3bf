// Stand-in for Bullet Physics' <btBulletDynamicsCommon.h> -- TEST INFRASTRUCTURE (oracle/_ref_build), never part of the product.
//
// Bullet (an un-vendored, un-pinned external of the reference) is absent from this image. This header provides the part of Bullet's API
// that the reference's sim/ sources use so that sim/World.cpp, sim/SimObj.cpp, sim/Joint.cpp, sim/ContactManager.cpp, sim/SimCharacter.cpp,
// sim/GroundVar2D.cpp and the controllers compile UNCHANGED (oracle/_ref_build/Makefile, libref_sim.so). It is a STATE CONTAINER, not a
// physics engine:
//   * rigid bodies hold a world transform, velocities, mass properties and force / torque accumulators (btScalar = float, as in the
//     reference's Bullet build); btTransform / btMatrix3x3 / btQuaternion / btVector3 implement the documented algebra;
//   * btHingeConstraint builds its two frames the way Bullet's constructor documents (frame x axis = body A's world x axis at
//     construction, swing axis in B) and getHingeAngle() = atan2(swing . ref0, swing . ref1); limits are stored, not enforced;
//   * btDiscreteDynamicsWorld::stepSimulation() does NOT integrate: it calls a hook the test harness installs (oracle/_ref_build/ref_sim_api.cpp
//     advances the state with the oracle's documented integrator or leaves it alone). Contact manifolds are whatever the harness injects.
// Everything the reference computes AROUND the physics step -- pose / velocity conversion, torque application and clamping, contact flags,
// fall detection, ground windows, terrain features, the FSM controllers, implicit PD, gravity compensation, virtual forces, action
// selection, rewards, tuples -- therefore runs as the reference wrote it. Written from Bullet's public API documentation; no Bullet source.
#pragma once
#include <cmath>
#include <cstddef>
#include <functional>
#include <vector>

#ifndef BT_SCALAR
#define BT_SCALAR float   // the reference's Bullet build (no BT_USE_DOUBLE_PRECISION in premake4.lua:115-124); the Makefile selects double for logic-parity tests
#endif
typedef BT_SCALAR btScalar;
// Bullet: SIMD_EPSILON = FLT_EPSILON, or DBL_EPSILON in a BT_USE_DOUBLE_PRECISION build. It matters in one place on this path: btQuaternion::getAxis()
// returns an ARBITRARY axis (1, 0, 0) when 1 - w^2 < 10 SIMD_EPSILON, i.e. for rotations below 2.2e-3 rad in the reference's float build, which makes
// cPDController::CalcTheta() of a world-coordinate joint (shoulder, hip) read 0 for the one or two env-steps in which the link's world angle crosses
// zero (sim/PDController.cpp:181-198 multiplies the angle by axis . z). The double build used for the logic-parity tests does not show it.
#define SIMD_EPSILON (sizeof(btScalar) == 8 ? btScalar(2.2204460492503131e-16) : btScalar(1.1920928955078125e-7))
#define SIMD_PI 3.1415926535897932384626433832795029f
#define BT_LARGE_FLOAT 1e18f
#define DISABLE_DEACTIVATION 4
#define ACTIVE_TAG 1
#define btAssert(x)
typedef int PHY_ScalarType;
#define PHY_FLOAT 0

inline btScalar btSqrt(btScalar x) { return std::sqrt(x); }
inline btScalar btFabs(btScalar x) { return std::fabs(x); }
inline btScalar btAtan2(btScalar y, btScalar x) { return std::atan2(y, x); }

class btVector3 {
public:
	btScalar m_floats[4];
	btVector3() { m_floats[0] = m_floats[1] = m_floats[2] = m_floats[3] = 0; }
	btVector3(btScalar x, btScalar y, btScalar z) { m_floats[0] = x; m_floats[1] = y; m_floats[2] = z; m_floats[3] = 0; }
	btScalar x() const { return m_floats[0]; }
	btScalar y() const { return m_floats[1]; }
	btScalar z() const { return m_floats[2]; }
	btScalar getX() const { return m_floats[0]; }
	btScalar getY() const { return m_floats[1]; }
	btScalar getZ() const { return m_floats[2]; }
	void setX(btScalar v) { m_floats[0] = v; }
	void setY(btScalar v) { m_floats[1] = v; }
	void setZ(btScalar v) { m_floats[2] = v; }
	void setValue(btScalar x, btScalar y, btScalar z) { m_floats[0] = x; m_floats[1] = y; m_floats[2] = z; m_floats[3] = 0; }
	void setZero() { setValue(0, 0, 0); }
	btScalar& operator[](int i) { return m_floats[i]; }
	const btScalar& operator[](int i) const { return m_floats[i]; }
	operator btScalar*() { return m_floats; }
	operator const btScalar*() const { return m_floats; }
	btVector3& operator+=(const btVector3& v) { m_floats[0] += v[0]; m_floats[1] += v[1]; m_floats[2] += v[2]; return *this; }
	btVector3& operator-=(const btVector3& v) { m_floats[0] -= v[0]; m_floats[1] -= v[1]; m_floats[2] -= v[2]; return *this; }
	btVector3& operator*=(btScalar s) { m_floats[0] *= s; m_floats[1] *= s; m_floats[2] *= s; return *this; }
	btVector3& operator*=(const btVector3& v) { m_floats[0] *= v[0]; m_floats[1] *= v[1]; m_floats[2] *= v[2]; return *this; }
	btVector3& operator/=(btScalar s) { return *this *= (btScalar(1) / s); }
	btScalar dot(const btVector3& v) const { return m_floats[0] * v[0] + m_floats[1] * v[1] + m_floats[2] * v[2]; }
	btVector3 cross(const btVector3& v) const { return btVector3(m_floats[1] * v[2] - m_floats[2] * v[1], m_floats[2] * v[0] - m_floats[0] * v[2], m_floats[0] * v[1] - m_floats[1] * v[0]); }
	btScalar length2() const { return dot(*this); }
	btScalar length() const { return btSqrt(length2()); }
	btVector3& normalize() { return *this /= length(); }
	btVector3 normalized() const { btVector3 r = *this; return r.normalize(); }
	bool isZero() const { return m_floats[0] == 0 && m_floats[1] == 0 && m_floats[2] == 0; }
};
inline btVector3 operator+(const btVector3& a, const btVector3& b) { return btVector3(a[0] + b[0], a[1] + b[1], a[2] + b[2]); }
inline btVector3 operator-(const btVector3& a, const btVector3& b) { return btVector3(a[0] - b[0], a[1] - b[1], a[2] - b[2]); }
inline btVector3 operator-(const btVector3& a) { return btVector3(-a[0], -a[1], -a[2]); }
inline btVector3 operator*(const btVector3& a, btScalar s) { return btVector3(a[0] * s, a[1] * s, a[2] * s); }
inline btVector3 operator*(btScalar s, const btVector3& a) { return a * s; }
inline btVector3 operator*(const btVector3& a, const btVector3& b) { return btVector3(a[0] * b[0], a[1] * b[1], a[2] * b[2]); }
inline btVector3 operator/(const btVector3& a, btScalar s) { return a * (btScalar(1) / s); }

class btQuaternion {
public:
	btScalar m_floats[4];   // x, y, z, w
	btQuaternion() { m_floats[0] = m_floats[1] = m_floats[2] = 0; m_floats[3] = 1; }
	btQuaternion(btScalar x, btScalar y, btScalar z, btScalar w) { m_floats[0] = x; m_floats[1] = y; m_floats[2] = z; m_floats[3] = w; }
	btQuaternion(const btVector3& axis, btScalar angle) { setRotation(axis, angle); }
	void setRotation(const btVector3& axis, btScalar angle)
	{
		const btScalar d = axis.length();
		const btScalar s = std::sin(angle * btScalar(0.5)) / d;
		m_floats[0] = axis.x() * s; m_floats[1] = axis.y() * s; m_floats[2] = axis.z() * s; m_floats[3] = std::cos(angle * btScalar(0.5));
	}
	btScalar x() const { return m_floats[0]; }
	btScalar y() const { return m_floats[1]; }
	btScalar z() const { return m_floats[2]; }
	btScalar w() const { return m_floats[3]; }
	btScalar getX() const { return m_floats[0]; }
	btScalar getY() const { return m_floats[1]; }
	btScalar getZ() const { return m_floats[2]; }
	btScalar getW() const { return m_floats[3]; }
	btScalar getAngle() const { btScalar w = m_floats[3]; w = w > 1 ? 1 : (w < -1 ? -1 : w); return btScalar(2) * std::acos(w); }
	btVector3 getAxis() const
	{
		const btScalar s2 = btScalar(1) - m_floats[3] * m_floats[3];
		if (s2 < btScalar(10) * SIMD_EPSILON) return btVector3(1, 0, 0);   // arbitrary, as in Bullet
		const btScalar s = btScalar(1) / btSqrt(s2);
		return btVector3(m_floats[0] * s, m_floats[1] * s, m_floats[2] * s);
	}
	btQuaternion inverse() const { return btQuaternion(-m_floats[0], -m_floats[1], -m_floats[2], m_floats[3]); }
	btScalar length() const { return btSqrt(m_floats[0] * m_floats[0] + m_floats[1] * m_floats[1] + m_floats[2] * m_floats[2] + m_floats[3] * m_floats[3]); }
	btQuaternion& normalize() { const btScalar l = length(); for (int i = 0; i < 4; ++i) m_floats[i] /= l; return *this; }
	static btQuaternion getIdentity() { return btQuaternion(0, 0, 0, 1); }
};
inline btQuaternion operator*(const btQuaternion& q1, const btQuaternion& q2)
{
	return btQuaternion(q1.w() * q2.x() + q1.x() * q2.w() + q1.y() * q2.z() - q1.z() * q2.y(),
		q1.w() * q2.y() + q1.y() * q2.w() + q1.z() * q2.x() - q1.x() * q2.z(),
		q1.w() * q2.z() + q1.z() * q2.w() + q1.x() * q2.y() - q1.y() * q2.x(),
		q1.w() * q2.w() - q1.x() * q2.x() - q1.y() * q2.y() - q1.z() * q2.z());
}

class btMatrix3x3 {
public:
	btVector3 m_el[3];   // rows
	btMatrix3x3() {}
	btMatrix3x3(const btQuaternion& q) { setRotation(q); }
	btMatrix3x3(btScalar xx, btScalar xy, btScalar xz, btScalar yx, btScalar yy, btScalar yz, btScalar zx, btScalar zy, btScalar zz) { setValue(xx, xy, xz, yx, yy, yz, zx, zy, zz); }
	void setValue(btScalar xx, btScalar xy, btScalar xz, btScalar yx, btScalar yy, btScalar yz, btScalar zx, btScalar zy, btScalar zz)
	{
		m_el[0].setValue(xx, xy, xz); m_el[1].setValue(yx, yy, yz); m_el[2].setValue(zx, zy, zz);
	}
	void setIdentity() { setValue(1, 0, 0, 0, 1, 0, 0, 0, 1); }
	static btMatrix3x3 getIdentity() { btMatrix3x3 m; m.setIdentity(); return m; }
	void setRotation(const btQuaternion& q)
	{
		const btScalar d = q.x() * q.x() + q.y() * q.y() + q.z() * q.z() + q.w() * q.w();
		const btScalar s = btScalar(2) / d;
		const btScalar xs = q.x() * s, ys = q.y() * s, zs = q.z() * s;
		const btScalar wx = q.w() * xs, wy = q.w() * ys, wz = q.w() * zs;
		const btScalar xx = q.x() * xs, xy = q.x() * ys, xz = q.x() * zs;
		const btScalar yy = q.y() * ys, yz = q.y() * zs, zz = q.z() * zs;
		setValue(btScalar(1) - (yy + zz), xy - wz, xz + wy, xy + wz, btScalar(1) - (xx + zz), yz - wx, xz - wy, yz + wx, btScalar(1) - (xx + yy));
	}
	// rotation about z by eulerX last ... Bullet: setEulerZYX(eulerX, eulerY, eulerZ) = Rz(eulerZ) Ry(eulerY) Rx(eulerX)
	void setEulerZYX(btScalar eulerX, btScalar eulerY, btScalar eulerZ)
	{
		const btScalar ci = std::cos(eulerX), cj = std::cos(eulerY), ch = std::cos(eulerZ);
		const btScalar si = std::sin(eulerX), sj = std::sin(eulerY), sh = std::sin(eulerZ);
		const btScalar cc = ci * ch, cs = ci * sh, sc = si * ch, ss = si * sh;
		setValue(cj * ch, sj * sc - cs, sj * cc + ss, cj * sh, sj * ss + cc, sj * cs - sc, -sj, cj * si, cj * ci);
	}
	const btVector3& getRow(int i) const { return m_el[i]; }
	btVector3 getColumn(int i) const { return btVector3(m_el[0][i], m_el[1][i], m_el[2][i]); }
	btVector3& operator[](int i) { return m_el[i]; }
	const btVector3& operator[](int i) const { return m_el[i]; }
	btMatrix3x3 transpose() const { return btMatrix3x3(m_el[0][0], m_el[1][0], m_el[2][0], m_el[0][1], m_el[1][1], m_el[2][1], m_el[0][2], m_el[1][2], m_el[2][2]); }
	btMatrix3x3 inverse() const { return transpose(); }   // rotations only
	void getRotation(btQuaternion& q) const
	{
		const btScalar trace = m_el[0][0] + m_el[1][1] + m_el[2][2];
		btScalar t[4];
		if (trace > 0) {
			btScalar s = btSqrt(trace + btScalar(1));
			t[3] = s * btScalar(0.5); s = btScalar(0.5) / s;
			t[0] = (m_el[2][1] - m_el[1][2]) * s; t[1] = (m_el[0][2] - m_el[2][0]) * s; t[2] = (m_el[1][0] - m_el[0][1]) * s;
		} else {
			const int i = m_el[0][0] < m_el[1][1] ? (m_el[1][1] < m_el[2][2] ? 2 : 1) : (m_el[0][0] < m_el[2][2] ? 2 : 0);
			const int j = (i + 1) % 3, k = (i + 2) % 3;
			btScalar s = btSqrt(m_el[i][i] - m_el[j][j] - m_el[k][k] + btScalar(1));
			t[i] = s * btScalar(0.5); s = btScalar(0.5) / s;
			t[3] = (m_el[k][j] - m_el[j][k]) * s; t[j] = (m_el[j][i] + m_el[i][j]) * s; t[k] = (m_el[k][i] + m_el[i][k]) * s;
		}
		q = btQuaternion(t[0], t[1], t[2], t[3]);
	}
};
inline btVector3 operator*(const btMatrix3x3& m, const btVector3& v) { return btVector3(m[0].dot(v), m[1].dot(v), m[2].dot(v)); }
inline btMatrix3x3 operator*(const btMatrix3x3& a, const btMatrix3x3& b)
{
	btMatrix3x3 r;
	for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) r[i][j] = a[i][0] * b[0][j] + a[i][1] * b[1][j] + a[i][2] * b[2][j];
	return r;
}

class btTransform {
public:
	btMatrix3x3 m_basis; btVector3 m_origin;
	btTransform() {}
	btTransform(const btQuaternion& q, const btVector3& c = btVector3(0, 0, 0)) : m_basis(q), m_origin(c) {}
	btTransform(const btMatrix3x3& b, const btVector3& c = btVector3(0, 0, 0)) : m_basis(b), m_origin(c) {}
	void setIdentity() { m_basis.setIdentity(); m_origin.setZero(); }
	static btTransform getIdentity() { btTransform t; t.setIdentity(); return t; }
	btMatrix3x3& getBasis() { return m_basis; }
	const btMatrix3x3& getBasis() const { return m_basis; }
	btVector3& getOrigin() { return m_origin; }
	const btVector3& getOrigin() const { return m_origin; }
	void setOrigin(const btVector3& o) { m_origin = o; }
	void setBasis(const btMatrix3x3& b) { m_basis = b; }
	void setRotation(const btQuaternion& q) { m_basis.setRotation(q); }
	btQuaternion getRotation() const { btQuaternion q; m_basis.getRotation(q); return q; }
	btVector3 operator*(const btVector3& v) const { return m_basis * v + m_origin; }
	btVector3 operator()(const btVector3& v) const { return (*this) * v; }
	btTransform operator*(const btTransform& t) const { return btTransform(m_basis * t.m_basis, (*this) * t.m_origin); }
	btTransform inverse() const { const btMatrix3x3 inv = m_basis.transpose(); return btTransform(inv, inv * (-m_origin)); }
	void getOpenGLMatrix(btScalar* m) const
	{
		for (int c = 0; c < 3; ++c) { m[4 * c] = m_basis[0][c]; m[4 * c + 1] = m_basis[1][c]; m[4 * c + 2] = m_basis[2][c]; m[4 * c + 3] = 0; }
		m[12] = m_origin.x(); m[13] = m_origin.y(); m[14] = m_origin.z(); m[15] = 1;
	}
};
inline btVector3 quatRotate(const btQuaternion& q, const btVector3& v) { return btMatrix3x3(q) * v; }

// ---- collision shapes ----
class btCollisionShape {
public:
	virtual ~btCollisionShape() {}
	virtual void calculateLocalInertia(btScalar mass, btVector3& inertia) const { inertia.setValue(0, 0, 0); }
	virtual void setLocalScaling(const btVector3& s) { m_scaling = s; }
	virtual const btVector3& getLocalScaling() const { return m_scaling; }
	virtual void getAabb(const btTransform& t, btVector3& aabbMin, btVector3& aabbMax) const { aabbMin = t.getOrigin(); aabbMax = t.getOrigin(); }
	virtual void setMargin(btScalar m) { m_margin = m; }
	virtual btScalar getMargin() const { return m_margin; }
	void setUserPointer(void* p) { m_user = p; }
	void* getUserPointer() const { return m_user; }
protected:
	btVector3 m_scaling{1, 1, 1};
	btScalar m_margin = 0.04f;   // Bullet's CONVEX_DISTANCE_MARGIN
	void* m_user = nullptr;
};
class btConvexShape : public btCollisionShape {
public:
	// btConvexInternalShape::setSafeMargin (Bullet >= 2.80, "issue 349"): a convex shape never carries a margin above a tenth of its smallest dimension
	void setSafeMargin(btScalar minDimension, btScalar defaultMarginMultiplier = btScalar(0.1))
	{
		const btScalar safe = defaultMarginMultiplier * minDimension;
		if (safe < getMargin()) setMargin(safe);
	}
};
class btConcaveShape : public btCollisionShape {};
class btBoxShape : public btConvexShape {
public:
	// btBoxShape::btBoxShape: setSafeMargin(boxHalfExtents) before the implicit (core) dimensions are derived -> margin = min(0.04, 0.1 x smallest half extent)
	explicit btBoxShape(const btVector3& half) : m_half(half) { setSafeMargin(std::min(half.x(), std::min(half.y(), half.z()))); }
	// btCollisionShape::getAngularMotionDisc for a box centred on its origin = |half extents|; getContactBreakingThreshold(0.02) = disc x 0.02
	// (btCollisionDispatcher::getNewManifold with CD_USE_RELATIVE_CONTACT_BREAKING_THRESHOLD, the dispatcher's default flag)
	btScalar getAngularMotionDisc() const { return m_half.length(); }
	btScalar getContactBreakingThreshold(btScalar defaultContactThreshold) const { return getAngularMotionDisc() * defaultContactThreshold; }
	btVector3 getHalfExtentsWithMargin() const { return m_half; }
	btVector3 getHalfExtentsWithoutMargin() const { return m_half - btVector3(m_margin, m_margin, m_margin); }
	void calculateLocalInertia(btScalar mass, btVector3& inertia) const override
	{
		const btScalar lx = btScalar(2) * m_half.x(), ly = btScalar(2) * m_half.y(), lz = btScalar(2) * m_half.z();
		inertia.setValue(mass / btScalar(12) * (ly * ly + lz * lz), mass / btScalar(12) * (lx * lx + lz * lz), mass / btScalar(12) * (lx * lx + ly * ly));
	}
	void getAabb(const btTransform& t, btVector3& mn, btVector3& mx) const override
	{
		btVector3 e(0, 0, 0);
		for (int i = 0; i < 3; ++i) e[i] = btFabs(t.getBasis()[i][0]) * m_half[0] + btFabs(t.getBasis()[i][1]) * m_half[1] + btFabs(t.getBasis()[i][2]) * m_half[2];
		mn = t.getOrigin() - e; mx = t.getOrigin() + e;
	}
private:
	btVector3 m_half;
};
class btCapsuleShape : public btConvexShape {
public:
	btCapsuleShape(btScalar radius, btScalar height) : m_r(radius), m_h(height) {}
	btScalar getRadius() const { return m_r; }
	btScalar getHalfHeight() const { return btScalar(0.5) * m_h; }
	void calculateLocalInertia(btScalar mass, btVector3& inertia) const override
	{
		const btScalar lx = btScalar(2) * m_r, ly = m_h + btScalar(2) * m_r, lz = btScalar(2) * m_r;
		inertia.setValue(mass / btScalar(12) * (ly * ly + lz * lz), mass / btScalar(12) * (lx * lx + lz * lz), mass / btScalar(12) * (lx * lx + ly * ly));
	}
private:
	btScalar m_r, m_h;
};
class btStaticPlaneShape : public btConcaveShape {
public:
	btStaticPlaneShape(const btVector3& n, btScalar c) : m_n(n), m_c(c) {}
	const btVector3& getPlaneNormal() const { return m_n; }
	btScalar getPlaneConstant() const { return m_c; }
private:
	btVector3 m_n; btScalar m_c;
};

// ---- collision objects / rigid bodies ----
class btMotionState {
public:
	virtual ~btMotionState() {}
	virtual void getWorldTransform(btTransform& t) const = 0;
	virtual void setWorldTransform(const btTransform& t) = 0;
};
class btDefaultMotionState : public btMotionState {
public:
	btTransform m_graphicsWorldTrans;
	explicit btDefaultMotionState(const btTransform& start = btTransform::getIdentity()) : m_graphicsWorldTrans(start) {}
	void getWorldTransform(btTransform& t) const override { t = m_graphicsWorldTrans; }
	void setWorldTransform(const btTransform& t) override { m_graphicsWorldTrans = t; }
};
struct btBroadphaseProxy { short m_collisionFilterGroup = 0, m_collisionFilterMask = 0; };
class btCollisionObject {
public:
	virtual ~btCollisionObject() {}
	btTransform& getWorldTransform() { return m_worldTransform; }
	const btTransform& getWorldTransform() const { return m_worldTransform; }
	void setWorldTransform(const btTransform& t) { m_worldTransform = t; }
	void setUserPointer(void* p) { m_user = p; }
	void* getUserPointer() const { return m_user; }
	void setFriction(btScalar f) { m_friction = f; }
	btScalar getFriction() const { return m_friction; }
	void setRestitution(btScalar r) { m_restitution = r; }
	btScalar getRestitution() const { return m_restitution; }
	void setActivationState(int s) { m_activation = s; }
	void forceActivationState(int s) { m_activation = s; }
	int getActivationState() const { return m_activation; }
	void activate(bool = false) {}
	btCollisionShape* getCollisionShape() { return m_shape; }
	const btCollisionShape* getCollisionShape() const { return m_shape; }
	void setCollisionShape(btCollisionShape* s) { m_shape = s; }
	void setCollisionFlags(int f) { m_flags = f; }
	int getCollisionFlags() const { return m_flags; }
	btBroadphaseProxy* getBroadphaseHandle() { return &m_proxy; }
	const btBroadphaseProxy* getBroadphaseHandle() const { return &m_proxy; }
protected:
	btTransform m_worldTransform = btTransform::getIdentity();
	void* m_user = nullptr;
	btScalar m_friction = 0.5f, m_restitution = 0;
	int m_activation = ACTIVE_TAG, m_flags = 0;
	btCollisionShape* m_shape = nullptr;
	btBroadphaseProxy m_proxy;
};
class btTypedConstraint;
class btRigidBody : public btCollisionObject {
public:
	struct btRigidBodyConstructionInfo {
		btScalar m_mass; btMotionState* m_motionState; btCollisionShape* m_collisionShape; btVector3 m_localInertia;
		btScalar m_friction = 0.5f, m_restitution = 0, m_linearDamping = 0, m_angularDamping = 0;
		btRigidBodyConstructionInfo(btScalar mass, btMotionState* ms, btCollisionShape* shape, const btVector3& inertia = btVector3(0, 0, 0))
			: m_mass(mass), m_motionState(ms), m_collisionShape(shape), m_localInertia(inertia) {}
	};
	explicit btRigidBody(const btRigidBodyConstructionInfo& ci) : m_motion(ci.m_motionState)
	{
		m_shape = ci.m_collisionShape; m_friction = ci.m_friction; m_restitution = ci.m_restitution;
		if (m_motion) m_motion->getWorldTransform(m_worldTransform);
		setMassProps(ci.m_mass, ci.m_localInertia);
	}
	static const btRigidBody* upcast(const btCollisionObject* o) { return dynamic_cast<const btRigidBody*>(o); }
	static btRigidBody* upcast(btCollisionObject* o) { return dynamic_cast<btRigidBody*>(o); }
	void setMassProps(btScalar mass, const btVector3& inertia)
	{
		m_invMass = mass == 0 ? 0 : btScalar(1) / mass;
		m_invInertiaLocal.setValue(inertia.x() != 0 ? btScalar(1) / inertia.x() : 0, inertia.y() != 0 ? btScalar(1) / inertia.y() : 0, inertia.z() != 0 ? btScalar(1) / inertia.z() : 0);
	}
	btScalar getInvMass() const { return m_invMass; }
	const btVector3& getInvInertiaDiagLocal() const { return m_invInertiaLocal; }
	btMotionState* getMotionState() { return m_motion; }
	const btMotionState* getMotionState() const { return m_motion; }
	const btTransform& getCenterOfMassTransform() const { return m_worldTransform; }
	void setCenterOfMassTransform(const btTransform& t) { m_worldTransform = t; }
	const btVector3& getCenterOfMassPosition() const { return m_worldTransform.getOrigin(); }
	btQuaternion getOrientation() const { return m_worldTransform.getRotation(); }
	const btVector3& getLinearVelocity() const { return m_linVel; }
	const btVector3& getAngularVelocity() const { return m_angVel; }
	void setLinearVelocity(const btVector3& v) { m_linVel = v; }
	void setAngularVelocity(const btVector3& v) { m_angVel = v; }
	btVector3 getVelocityInLocalPoint(const btVector3& rel_pos) const { return m_linVel + m_angVel.cross(rel_pos); }
	void setLinearFactor(const btVector3& f) { m_linFactor = f; }
	void setAngularFactor(const btVector3& f) { m_angFactor = f; }
	const btVector3& getLinearFactor() const { return m_linFactor; }
	const btVector3& getAngularFactor() const { return m_angFactor; }
	void setDamping(btScalar lin, btScalar ang) { m_linDamping = lin; m_angDamping = ang; }
	void setGravity(const btVector3& g) { m_gravity = g; }
	const btVector3& getGravity() const { return m_gravity; }
	void applyCentralForce(const btVector3& f) { m_totalForce += f * m_linFactor; }
	void applyTorque(const btVector3& t) { m_totalTorque += t * m_angFactor; }
	void applyForce(const btVector3& f, const btVector3& rel_pos) { applyCentralForce(f); applyTorque(rel_pos.cross(f * m_linFactor)); }
	void clearForces() { m_totalForce.setZero(); m_totalTorque.setZero(); }
	const btVector3& getTotalForce() const { return m_totalForce; }
	const btVector3& getTotalTorque() const { return m_totalTorque; }
	void getAabb(btVector3& mn, btVector3& mx) const { if (m_shape) m_shape->getAabb(m_worldTransform, mn, mx); else { mn = m_worldTransform.getOrigin(); mx = mn; } }
	int getNumConstraintRefs() const { return static_cast<int>(m_consRefs.size()); }
	btTypedConstraint* getConstraintRef(int i) { return m_consRefs[i]; }
	void addConstraintRef(btTypedConstraint* c) { m_consRefs.push_back(c); }
	void removeConstraintRef(btTypedConstraint* c) { for (size_t i = 0; i < m_consRefs.size(); ++i) if (m_consRefs[i] == c) { m_consRefs.erase(m_consRefs.begin() + i); break; } }
private:
	btMotionState* m_motion;
	btScalar m_invMass = 0, m_linDamping = 0, m_angDamping = 0;
	btVector3 m_invInertiaLocal, m_linVel, m_angVel, m_linFactor{1, 1, 1}, m_angFactor{1, 1, 1}, m_gravity, m_totalForce, m_totalTorque;
	std::vector<btTypedConstraint*> m_consRefs;
};
typedef btRigidBody::btRigidBodyConstructionInfo btRigidBodyConstructionInfo;

// ---- constraints ----
class btTypedConstraint {
public:
	btTypedConstraint(btRigidBody& a, btRigidBody& b) : m_rbA(&a), m_rbB(&b) {}
	explicit btTypedConstraint(btRigidBody& a) : m_rbA(&a), m_rbB(nullptr) {}
	virtual ~btTypedConstraint() {}
	btRigidBody& getRigidBodyA() { return *m_rbA; }
	btRigidBody& getRigidBodyB() { return *m_rbB; }
	const btRigidBody& getRigidBodyA() const { return *m_rbA; }
	const btRigidBody& getRigidBodyB() const { return *m_rbB; }
	bool hasBodyB() const { return m_rbB != nullptr; }
	void setEnabled(bool e) { m_enabled = e; }
	bool isEnabled() const { return m_enabled; }
protected:
	btRigidBody* m_rbA; btRigidBody* m_rbB; bool m_enabled = true;
};
class btHingeConstraint : public btTypedConstraint {
public:
	// frames as Bullet's two-body constructor documents them: the hinge frame's x axis in A is body A's WORLD x axis at construction taken as a local
	// axis (made orthogonal to the hinge axis), the same local axis is used in B (axes A and B coincide in the reference: both (0, 0, 1))
	btHingeConstraint(btRigidBody& a, btRigidBody& b, const btVector3& pivotA, const btVector3& pivotB, const btVector3& axisA, const btVector3& axisB, bool useReferenceFrameA = false)
		: btTypedConstraint(a, b), m_sign(useReferenceFrameA ? btScalar(-1) : btScalar(1))
	{
		buildFrame(a, axisA, pivotA, m_frameA);
		// rotationArc(axisA, axisB) is the identity when the axes coincide; the general case rotates A's x axis into B's frame
		btVector3 x1 = m_frameA.getBasis().getColumn(0);
		const btScalar d = axisA.dot(axisB);
		if (d < btScalar(1) - SIMD_EPSILON) {
			btVector3 c = axisA.cross(axisB);
			const btScalar s = btSqrt((btScalar(1) + d) * btScalar(2));
			btQuaternion arc(c.x() / s, c.y() / s, c.z() / s, s * btScalar(0.5));
			x1 = quatRotate(arc, x1);
		}
		const btVector3 y1 = axisB.cross(x1);
		m_frameB.getBasis().setValue(x1.x(), y1.x(), axisB.x(), x1.y(), y1.y(), axisB.y(), x1.z(), y1.z(), axisB.z());
		m_frameB.setOrigin(pivotB);
	}
	btHingeConstraint(btRigidBody& a, const btVector3& pivotA, const btVector3& axisA, bool useReferenceFrameA = false)
		: btTypedConstraint(a), m_sign(useReferenceFrameA ? btScalar(-1) : btScalar(1))
	{
		buildFrame(a, axisA, pivotA, m_frameA);
		m_frameB = a.getCenterOfMassTransform() * m_frameA;   // world frame at construction
	}
	void setLimit(btScalar low, btScalar high, btScalar = 0.9f, btScalar = 0.3f, btScalar = 1.0f) { m_low = low; m_high = high; }
	btScalar getLowerLimit() const { return m_low; }
	btScalar getUpperLimit() const { return m_high; }
	btScalar getHingeAngle() const
	{
		const btTransform ta = m_rbA->getCenterOfMassTransform();
		const btTransform tb = m_rbB ? m_rbB->getCenterOfMassTransform() : btTransform::getIdentity();
		return getHingeAngle(ta, tb);
	}
	btScalar getHingeAngle(const btTransform& ta, const btTransform& tb) const
	{
		const btVector3 ref0 = ta.getBasis() * m_frameA.getBasis().getColumn(0);
		const btVector3 ref1 = ta.getBasis() * m_frameA.getBasis().getColumn(1);
		const btVector3 swing = tb.getBasis() * m_frameB.getBasis().getColumn(1);
		return m_sign * btAtan2(swing.dot(ref0), swing.dot(ref1));
	}
	const btTransform& getAFrame() const { return m_frameA; }
	const btTransform& getBFrame() const { return m_frameB; }
private:
	static void buildFrame(const btRigidBody& a, const btVector3& axis, const btVector3& pivot, btTransform& frame)
	{
		btVector3 x = a.getCenterOfMassTransform().getBasis().getColumn(0), y;
		const btScalar p = axis.dot(x);
		if (p >= btScalar(1) - SIMD_EPSILON) { x = -a.getCenterOfMassTransform().getBasis().getColumn(2); y = a.getCenterOfMassTransform().getBasis().getColumn(1); }
		else if (p <= btScalar(-1) + SIMD_EPSILON) { x = a.getCenterOfMassTransform().getBasis().getColumn(2); y = a.getCenterOfMassTransform().getBasis().getColumn(1); }
		else { y = axis.cross(x); x = y.cross(axis); }
		frame.getBasis().setValue(x.x(), y.x(), axis.x(), x.y(), y.y(), axis.y(), x.z(), y.z(), axis.z());
		frame.setOrigin(pivot);
	}
	btTransform m_frameA, m_frameB;
	btScalar m_low = 1, m_high = -1, m_sign;
};
class btSliderConstraint : public btTypedConstraint {
public:
	btSliderConstraint(btRigidBody& a, btRigidBody& b, const btTransform& fa, const btTransform& fb, bool) : btTypedConstraint(a, b), m_fa(fa), m_fb(fb) {}
	btSliderConstraint(btRigidBody& b, const btTransform& fb, bool) : btTypedConstraint(b), m_fb(fb) {}
	void setLowerLinLimit(btScalar v) { m_ll = v; }
	void setUpperLinLimit(btScalar v) { m_ul = v; }
	void setLowerAngLimit(btScalar v) { m_la = v; }
	void setUpperAngLimit(btScalar v) { m_ua = v; }
	btScalar getLinearPos() const { return 0; }
private:
	btTransform m_fa, m_fb; btScalar m_ll = 0, m_ul = 0, m_la = 0, m_ua = 0;
};

// ---- manifolds / dispatcher / broadphase / solver ----
class btManifoldPoint {
public:
	btVector3 m_positionWorldOnA, m_positionWorldOnB, m_normalWorldOnB; btScalar m_distance1 = 0;
	btScalar getDistance() const { return m_distance1; }
	const btVector3& getPositionWorldOnA() const { return m_positionWorldOnA; }
	const btVector3& getPositionWorldOnB() const { return m_positionWorldOnB; }
};
class btPersistentManifold {
public:
	const btCollisionObject* m_body0 = nullptr; const btCollisionObject* m_body1 = nullptr;
	std::vector<btManifoldPoint> m_points;
	const btCollisionObject* getBody0() const { return m_body0; }
	const btCollisionObject* getBody1() const { return m_body1; }
	int getNumContacts() const { return static_cast<int>(m_points.size()); }
	btManifoldPoint& getContactPoint(int i) { return m_points[i]; }
	const btManifoldPoint& getContactPoint(int i) const { return m_points[i]; }
};
class btCollisionConfiguration { public: virtual ~btCollisionConfiguration() {} };
class btDefaultCollisionConfiguration : public btCollisionConfiguration {};
class btDispatcher {
public:
	virtual ~btDispatcher() {}
	int getNumManifolds() const { return static_cast<int>(m_manifolds.size()); }
	btPersistentManifold* getManifoldByIndexInternal(int i) { return &m_manifolds[i]; }
	std::vector<btPersistentManifold> m_manifolds;   // filled by the test harness
};
class btCollisionDispatcher : public btDispatcher { public: explicit btCollisionDispatcher(btCollisionConfiguration*) {} };
struct btBroadphasePair {};
class btBroadphasePairArray { public: int size() const { return 0; } btBroadphasePair& operator[](int) { static btBroadphasePair p; return p; } };
class btOverlappingPairCache {
public:
	btBroadphasePairArray& getOverlappingPairArray() { return m_pairs; }
	void cleanOverlappingPair(btBroadphasePair&, btDispatcher*) {}
private:
	btBroadphasePairArray m_pairs;
};
class btBroadphaseInterface {
public:
	virtual ~btBroadphaseInterface() {}
	virtual void resetPool(btDispatcher*) {}
	btOverlappingPairCache* getOverlappingPairCache() { return &m_cache; }
private:
	btOverlappingPairCache m_cache;
};
class btDbvtBroadphase : public btBroadphaseInterface {};
class btConstraintSolver { public: virtual ~btConstraintSolver() {} virtual void reset() { if (m_resetHook) m_resetHook(); } std::function<void()> m_resetHook; };   // (cWorld::Reset calls reset(): the harness's integrator drops its cached contact impulses there)
class btSequentialImpulseConstraintSolver : public btConstraintSolver {};

// ---- world ----
class btCollisionWorld {
public:
	struct RayResultCallback { virtual ~RayResultCallback() {} bool hasHit() const { return false; } short m_collisionFilterGroup = 1, m_collisionFilterMask = -1; };
	struct ClosestRayResultCallback : public RayResultCallback {
		ClosestRayResultCallback(const btVector3& from, const btVector3& to) : m_rayFromWorld(from), m_rayToWorld(to) {}
		btVector3 m_rayFromWorld, m_rayToWorld, m_hitPointWorld, m_hitNormalWorld; const btCollisionObject* m_collisionObject = nullptr;
	};
	struct AllHitsRayResultCallback : public RayResultCallback {
		AllHitsRayResultCallback(const btVector3& from, const btVector3& to) : m_rayFromWorld(from), m_rayToWorld(to) {}
		btVector3 m_rayFromWorld, m_rayToWorld; std::vector<const btCollisionObject*> m_collisionObjects; std::vector<btVector3> m_hitPointWorld;
	};
	virtual ~btCollisionWorld() {}
	void rayTest(const btVector3&, const btVector3&, RayResultCallback&) const {}
};
class btDiscreteDynamicsWorld : public btCollisionWorld {
public:
	btDiscreteDynamicsWorld(btDispatcher* d, btBroadphaseInterface* b, btConstraintSolver* s, btCollisionConfiguration*) : m_dispatcher(d), m_broadphase(b), m_solver(s) {}
	void setGravity(const btVector3& g) { m_gravity = g; for (btRigidBody* rb : m_bodies) rb->setGravity(g); }
	btVector3 getGravity() const { return m_gravity; }
	void addRigidBody(btRigidBody* rb) { rb->setGravity(m_gravity); m_bodies.push_back(rb); }
	void addRigidBody(btRigidBody* rb, short group, short mask) { rb->getBroadphaseHandle()->m_collisionFilterGroup = group; rb->getBroadphaseHandle()->m_collisionFilterMask = mask; addRigidBody(rb); }
	void removeRigidBody(btRigidBody* rb) { for (size_t i = 0; i < m_bodies.size(); ++i) if (m_bodies[i] == rb) { m_bodies.erase(m_bodies.begin() + i); break; } }
	void addCollisionObject(btCollisionObject* o, short group = 1, short mask = -1) { o->getBroadphaseHandle()->m_collisionFilterGroup = group; o->getBroadphaseHandle()->m_collisionFilterMask = mask; m_objects.push_back(o); }
	void removeCollisionObject(btCollisionObject* o) { for (size_t i = 0; i < m_objects.size(); ++i) if (m_objects[i] == o) { m_objects.erase(m_objects.begin() + i); break; } if (btRigidBody* rb = btRigidBody::upcast(o)) removeRigidBody(rb); }
	void addConstraint(btTypedConstraint* c, bool disableCollisionsBetweenLinkedBodies = false)
	{
		m_constraints.push_back(c); m_noCollide.push_back(disableCollisionsBetweenLinkedBodies);
		c->getRigidBodyA().addConstraintRef(c); if (c->hasBodyB()) c->getRigidBodyB().addConstraintRef(c);
	}
	void removeConstraint(btTypedConstraint* c)
	{
		for (size_t i = 0; i < m_constraints.size(); ++i) if (m_constraints[i] == c) { m_constraints.erase(m_constraints.begin() + i); m_noCollide.erase(m_noCollide.begin() + i); break; }
		c->getRigidBodyA().removeConstraintRef(c); if (c->hasBodyB()) c->getRigidBodyB().removeConstraintRef(c);
	}
	int getNumConstraints() const { return static_cast<int>(m_constraints.size()); }
	btTypedConstraint* getConstraint(int i) { return m_constraints[i]; }
	btDispatcher* getDispatcher() { return m_dispatcher; }
	btBroadphaseInterface* getBroadphase() { return m_broadphase; }
	btConstraintSolver* getConstraintSolver() { return m_solver; }
	void clearForces() { for (btRigidBody* rb : m_bodies) rb->clearForces(); }
	// NO physics here: the harness decides what a step does (see the header comment). Forces are cleared afterwards as Bullet does.
	int stepSimulation(btScalar timeStep, int maxSubSteps = 1, btScalar fixedTimeStep = btScalar(1) / btScalar(60))
	{
		if (m_stepHook) m_stepHook(timeStep, maxSubSteps, fixedTimeStep);
		clearForces();
		return maxSubSteps;
	}
	std::function<void(btScalar, int, btScalar)> m_stepHook;
	const std::vector<btRigidBody*>& bodies() const { return m_bodies; }
private:
	btDispatcher* m_dispatcher; btBroadphaseInterface* m_broadphase; btConstraintSolver* m_solver;
	btVector3 m_gravity;
	std::vector<btRigidBody*> m_bodies; std::vector<btCollisionObject*> m_objects;
	std::vector<btTypedConstraint*> m_constraints; std::vector<bool> m_noCollide;
};
