// Stand-in (test infrastructure, see ../../btBulletDynamicsCommon.h): the heightfield keeps the caller's data pointer, dimensions, height range
// and local scaling; getAabb() returns the scaled box centred on the origin the way Bullet centres heightfields (half extents = (width-1)/2 etc.
// times the scaling, vertical range centred on (min+max)/2) plus the collision margin -- which is 0 for concave shapes: btConcaveShape keeps its
// own m_collisionMargin, initialised to 0 (unlike the convex shapes' CONVEX_DISTANCE_MARGIN 0.04), and btHeightfieldTerrainShape never sets it.
#pragma once
#include "btBulletDynamicsCommon.h"
class btHeightfieldTerrainShape : public btConcaveShape {
public:
	btHeightfieldTerrainShape(int heightStickWidth, int heightStickLength, const void* data, btScalar heightScale, btScalar minHeight, btScalar maxHeight, int upAxis, PHY_ScalarType, bool flipQuadEdges)
		: m_w(heightStickWidth), m_l(heightStickLength), m_data(data), m_hscale(heightScale), m_min(minHeight), m_max(maxHeight), m_up(upAxis), m_flip(flipQuadEdges) { m_margin = 0; }
	void getAabb(const btTransform& t, btVector3& mn, btVector3& mx) const override
	{
		btVector3 half(btScalar(0.5) * (m_w - 1), btScalar(0.5) * (m_max - m_min), btScalar(0.5) * (m_l - 1));
		if (m_up == 0) half = btVector3(btScalar(0.5) * (m_max - m_min), btScalar(0.5) * (m_w - 1), btScalar(0.5) * (m_l - 1));
		if (m_up == 2) half = btVector3(btScalar(0.5) * (m_w - 1), btScalar(0.5) * (m_l - 1), btScalar(0.5) * (m_max - m_min));
		half = half * m_scaling + btVector3(m_margin, m_margin, m_margin);
		btVector3 e(0, 0, 0);
		for (int i = 0; i < 3; ++i) e[i] = btFabs(t.getBasis()[i][0]) * half[0] + btFabs(t.getBasis()[i][1]) * half[1] + btFabs(t.getBasis()[i][2]) * half[2];
		mn = t.getOrigin() - e; mx = t.getOrigin() + e;
	}
	void setUseDiamondSubdivision(bool = true) {}
	int width() const { return m_w; }
	int length() const { return m_l; }
	const void* data() const { return m_data; }
private:
	int m_w, m_l; const void* m_data; btScalar m_hscale, m_min, m_max; int m_up; bool m_flip;
};
