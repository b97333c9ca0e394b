// kdbuild.cpp -- host-side SAH kd-tree construction for libmtsgpu.
//
// Needed because the standalone driver / bench run where no Mitsuba exists
// (inside Mitsuba the plugin flattens the tree Scene::initialize already built).
// Same cost model and rules as the reference builder
// (include/mitsuba/render/gkdtree.h:913-1214, :1735-1867, :1898-2345, :2350-2608;
// SAH include/mitsuba/render/sahkdtree3.h:35-79; clipping src/libcore/triangle.cpp:59-158):
//   > exactPrimThreshold primitives : 128-bin min-max binning, tight child boxes
//   <= exactPrimThreshold           : exact O(n log n) sweep over sorted edge events with
//                                     perfect splits (re-clipping), empty-space bonus,
//                                     "bad refines" and retraction of subtrees that did not pay off
// Subtrees below the binning phase are independent jobs and are built by a pool
// of host threads (the reference hands them to its TreeBuilder threads, :1668-1704;
// like there, such a subtree reports cost -inf so it is never retracted from above).
// Edge events that compare equal are additionally ordered by primitive index so that
// the result does not depend on the sort implementation.
#include "host.h"
#include <hip/hip_runtime.h>
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/iterator/transform_iterator.hpp>
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstring>
#include <exception>
#include <functional>
#include <limits>
#include <mutex>
#include <thread>
#include <chrono>
#include <cstdio>
#include <cstdlib>

namespace mg {
namespace {

constexpr float kInf = std::numeric_limits<float>::infinity();
constexpr float kEps = 1e-4f;

struct Box {
	float mn[3], mx[3];
	void reset() { for (int i = 0; i < 3; ++i) { mn[i] = kInf; mx[i] = -kInf; } }
	void expand(const float *p) { for (int i = 0; i < 3; ++i) { mn[i] = std::min(mn[i], p[i]); mx[i] = std::max(mx[i], p[i]); } }
	void expand(const Box &b) { for (int i = 0; i < 3; ++i) { mn[i] = std::min(mn[i], b.mn[i]); mx[i] = std::max(mx[i], b.mx[i]); } }
	void clip(const Box &b) { for (int i = 0; i < 3; ++i) { mn[i] = std::max(mn[i], b.mn[i]); mx[i] = std::min(mx[i], b.mx[i]); } }
	bool valid() const { for (int i = 0; i < 3; ++i) if (mx[i] < mn[i]) return false; return true; }
	float area() const {            // aabb.h:326-329
		const float dx = mx[0] - mn[0], dy = mx[1] - mn[1], dz = mx[2] - mn[2];
		return (float) 2.0 * (dx * dy + dx * dz + dy * dz);
	}
};

// std::min(a, b) = (b < a) ? b : a and std::max(a, b) = (a < b) ? b : a, usable on the device
__host__ __device__ inline float hdMin(float a, float b) { return (b < a) ? b : a; }
__host__ __device__ inline float hdMax(float a, float b) { return (a < b) ? b : a; }

// Sutherland-Hodgman against one plane, double precision (triangle.cpp:61-106)
__host__ __device__ inline int clipPlane(const double (*in)[3], int inCount, double (*out)[3], int axis, double splitPos, bool isMinimum) {
	if (inCount < 3)
		return 0;
	double cur[3] = { in[0][0], in[0][1], in[0][2] };
	const double sign = isMinimum ? 1.0f : -1.0f;
	double distance = sign * (cur[axis] - splitPos);
	bool curIsInside = (distance >= 0);
	int outCount = 0;
	for (int i = 0; i < inCount; ++i) {
		const int nextIdx = (i + 1 == inCount) ? 0 : i + 1;
		const double next[3] = { in[nextIdx][0], in[nextIdx][1], in[nextIdx][2] };
		distance = sign * (next[axis] - splitPos);
		const bool nextIsInside = (distance >= 0);
		if (curIsInside && nextIsInside) {
			for (int c = 0; c < 3; ++c) out[outCount][c] = next[c];
			outCount++;
		} else if (curIsInside != nextIsInside) {
			const double t = (splitPos - cur[axis]) / (next[axis] - cur[axis]);
			for (int c = 0; c < 3; ++c) out[outCount][c] = cur[c] + (next[c] - cur[c]) * t;
			out[outCount][axis] = splitPos;
			outCount++;
			if (nextIsInside) {
				for (int c = 0; c < 3; ++c) out[outCount][c] = next[c];
				outCount++;
			}
		}
		for (int c = 0; c < 3; ++c) cur[c] = next[c];
		curIsInside = nextIsInside;
	}
	return outCount;
}

// Triangle::getClippedAABB (triangle.cpp:108-158): clip in double, round outward to float
__host__ __device__ inline bool clippedTriangleBoxHD(const float *p0, const float *p1, const float *p2, const float *bmin, const float *bmax,
                                                     float *omin, float *omax) {
	const float inf = INFINITY;
	double a[10][3], b[10][3];
	for (int c = 0; c < 3; ++c) { a[0][c] = p0[c]; a[1][c] = p1[c]; a[2][c] = p2[c]; }
	int n = 3;
	for (int axis = 0; axis < 3; ++axis) {
		n = clipPlane(a, n, b, axis, (double) bmin[axis], true);
		n = clipPlane(b, n, a, axis, (double) bmax[axis], false);
	}
	for (int c = 0; c < 3; ++c) { omin[c] = inf; omax[c] = -inf; }
	for (int i = 0; i < n; ++i)
		for (int j = 0; j < 3; ++j) {
			const double pos_d = a[i][j];
			const float pos_f = (float) pos_d;
			float lo, hi;
			if (pos_f < pos_d) { lo = pos_f; hi = nextafterf(pos_f, inf); }
			else if (pos_f > pos_d) { hi = pos_f; lo = nextafterf(pos_f, -inf); }
			else lo = hi = pos_f;
			omin[j] = hdMin(omin[j], lo);
			omax[j] = hdMax(omax[j], hi);
		}
	for (int c = 0; c < 3; ++c) { omin[c] = hdMax(omin[c], bmin[c]); omax[c] = hdMin(omax[c], bmax[c]); }
	for (int c = 0; c < 3; ++c)
		if (omax[c] < omin[c])
			return false;
	return true;
}

} // namespace

bool clippedTriangleBox(const float *p0, const float *p1, const float *p2, const float *bmin, const float *bmax,
                        float *omin, float *omax) {
	return clippedTriangleBoxHD(p0, p1, p2, bmin, bmax, omin, omax);
}

namespace {

// SurfaceAreaHeuristic (sahkdtree3.h:35-79)
struct SAH {
	float t0[3], t1[3];
	explicit SAH(const Box &b) {
		const float e[3] = { b.mx[0] - b.mn[0], b.mx[1] - b.mn[1], b.mx[2] - b.mn[2] };
		const float temp = 1.0f / (e[0] * e[1] + e[1] * e[2] + e[0] * e[2]);
		t0[0] = (e[1] * e[2]) * temp; t0[1] = (e[0] * e[2]) * temp; t0[2] = (e[0] * e[1]) * temp;
		t1[0] = (e[1] + e[2]) * temp; t1[1] = (e[0] + e[2]) * temp; t1[2] = (e[0] + e[1]) * temp;
	}
	void operator()(int axis, float leftWidth, float rightWidth, float &pl, float &pr) const {
		pl = t0[axis] + t1[axis] * leftWidth;
		pr = t0[axis] + t1[axis] * rightWidth;
	}
};

enum : uint16_t { kEnd = 0, kPlanar = 1, kStart = 2 };
struct Event { float pos; uint32_t index; uint16_t type, axis; };
struct EventLess {
	bool operator()(const Event &a, const Event &b) const {
		if (a.axis != b.axis) return a.axis < b.axis;
		if (a.pos != b.pos) return a.pos < b.pos;
		if (a.type != b.type) return a.type < b.type;
		return a.index < b.index;
	}
};

struct Split { float cost = kInf, pos = 0; int axis = 0; uint32_t numLeft = 0, numRight = 0; bool planarLeft = false; };
// a candidate split plane of the exact phase: how many primitives end on it, lie in it, start on it (indexed by event type)
struct Candidate { float pos; int axis; uint32_t count[3]; };

// preliminary node: leaf {start, end} into the owning context's index list, inner {axis, children, split},
// or a reference to a subtree built by a job
struct PNode { uint8_t kind; uint32_t a, b; float split; };   // kind: 0 inner, 1 leaf, 2 job reference
struct Params {
	float traversalCost, queryCost, emptySpaceBonus;
	uint32_t stopPrims, maxBadRefines, exactPrimThreshold, maxDepth;
	int minMaxBins; bool clip, retract;
};

struct Geometry {
	const float *vtx; const uint32_t *tri;
	const float *genBox;     // [n][6] boxes of the non-triangle primitives (tri row = {NONE, NONE, NONE})
	void box(uint32_t i, Box &b) const {
		const uint32_t *t = tri + 3 * (size_t) i;
		if (t[0] == MTSGPU_KNOTRIANGLE) {            // shape->getAABB() (skdtree.h:203-213)
			for (int a = 0; a < 3; ++a) { b.mn[a] = genBox[6 * (size_t) i + a]; b.mx[a] = genBox[6 * (size_t) i + 3 + a]; }
			return;
		}
		b.reset(); b.expand(vtx + 3 * (size_t) t[0]); b.expand(vtx + 3 * (size_t) t[1]); b.expand(vtx + 3 * (size_t) t[2]);
	}
	bool clipped(uint32_t i, const Box &to, Box &b) const {
		const uint32_t *t = tri + 3 * (size_t) i;
		if (t[0] == MTSGPU_KNOTRIANGLE) {            // Shape::getClippedAABB (shape.cpp:59-63): getAABB().clip(box)
			box(i, b);
			b.clip(to);
			return b.valid();
		}
		return clippedTriangleBox(vtx + 3 * (size_t) t[0], vtx + 3 * (size_t) t[1], vtx + 3 * (size_t) t[2], to.mn, to.mx, b.mn, b.mx);
	}
};

// One build context = one independently growing piece of the tree (BuildContext, gkdtree.h)
struct Context {
	std::vector<PNode> nodes;
	std::vector<uint32_t> indices;
	uint32_t leafCount = 0, nonemptyLeafCount = 0, innerCount = 0, primIndexCount = 0, retracted = 0, pruned = 0;
	uint32_t allocNodes(uint32_t n) { const uint32_t r = (uint32_t) nodes.size(); nodes.resize(nodes.size() + n); return r; }
	std::vector<Candidate> planes;      // scratch of Builder::sweep: the candidate planes of the node being split
};

struct Job { uint32_t depth; Box nodeBox; std::vector<uint32_t> prims; uint32_t badRefines; Context ctx; uint32_t root; double ms = 0; };

// (int64_t) of a bin coordinate as the x86 conversion the reference compiles to defines it: values outside the
// int64 range (a node that is flat along the axis divides by zero) give INT64_MIN; host and device agree on this
__host__ __device__ inline int64_t binIndex(float v) {
	return (v >= -9223372036854775808.0f && v < 9223372036854775808.0f) ? (int64_t) v : (int64_t) 0x8000000000000000ull;
}

// Record of the min-max binning phase when it ran on the device: per node what minimizeCost chose, the unions of the
// boxes sent to either side and, for nodes the phase ends at, their primitive list
struct PlanNode {
	uint32_t count = 0;
	Split best;
	Box leftRaw, rightRaw;
	std::vector<uint32_t> prims;
	int child[2] = { -1, -1 };
};
struct Plan { std::vector<PlanNode> nodes; };

class Builder {
public:
	Builder(const Geometry &g, const Params &p, uint32_t nPrims) : m_g(g), m_p(p), m_nPrims(nPrims) {}

	// SAH cost of splitting at a plane with nBelow / nAbove primitives on its sides (gkdtree.h:1974-1980: the empty-space
	// bonus applies when one side is empty)
	float planeCost(float probBelow, float probAbove, uint32_t nBelow, uint32_t nAbove) const {
		float cost = m_p.traversalCost + m_p.queryCost * (probBelow * (float) nBelow + probAbove * (float) nAbove);
		if (nBelow == 0 || nAbove == 0)
			cost *= m_p.emptySpaceBonus;
		return cost;
	}

	// The cheapest of the candidate planes strictly inside the node.  Primitives lying IN a plane go to the side that is
	// cheaper, to the right one when both cost the same (gkdtree.h:1996-2014)
	Split cheapestPlane(const std::vector<Candidate> &planes, const Box &nodeBox, uint32_t primCount) const {
		Split best;
		const SAH area(nodeBox);
		uint32_t below[3] = { 0, 0, 0 }, above[3] = { primCount, primCount, primCount };
		for (const Candidate &pl : planes) {
			const int k = pl.axis;
			const uint32_t inPlane = pl.count[kPlanar];
			above[k] -= inPlane + pl.count[kEnd];
			if (pl.pos > nodeBox.mn[k] && pl.pos < nodeBox.mx[k]) {
				float probBelow, probAbove;
				area(k, pl.pos - nodeBox.mn[k], nodeBox.mx[k] - pl.pos, probBelow, probAbove);
				Split s;
				s.pos = pl.pos; s.axis = k; s.numLeft = below[k]; s.numRight = above[k];
				s.cost = planeCost(probBelow, probAbove, below[k], above[k]);
				if (inPlane != 0) {
					const float withLeft = planeCost(probBelow, probAbove, below[k] + inPlane, above[k]);
					const float withRight = planeCost(probBelow, probAbove, below[k], above[k] + inPlane);
					s.planarLeft = withLeft < withRight;
					s.cost = s.planarLeft ? withLeft : withRight;
					(s.planarLeft ? s.numLeft : s.numRight) += inPlane;
				}
				if (s.cost < best.cost)
					best = s;
			}
			below[k] += pl.count[kStart] + inPlane;
		}
		return best;
	}

	static void pushEvents(std::vector<Event> &out, const Box &b, uint32_t index) {
		for (int axis = 0; axis < 3; ++axis) {
			const float mn = b.mn[axis], mx = b.mx[axis];
			if (mn == mx) {
				out.push_back(Event{ mn, index, kPlanar, (uint16_t) axis });
			} else {
				out.push_back(Event{ mn, index, kStart, (uint16_t) axis });
				out.push_back(Event{ mx, index, kEnd, (uint16_t) axis });
			}
		}
	}

	void leafFromEvents(Context &c, uint32_t node, const Event *es, const Event *ee, uint32_t primCount) const {
		PNode &n = c.nodes[node];
		n.kind = 1; n.a = (uint32_t) c.indices.size(); n.b = n.a + primCount;
		if (primCount > 0) {
			c.nonemptyLeafCount++;
			for (const Event *e = es; e != ee && e->axis == 0; ++e)
				if (e->type == kStart || e->type == kPlanar)
					c.indices.push_back(e->index);
			c.primIndexCount += primCount;
		}
		c.leafCount++;
	}

	void leafFromIndices(Context &c, uint32_t node, const uint32_t *idx, uint32_t primCount) const {
		PNode &n = c.nodes[node];
		n.kind = 1; n.a = (uint32_t) c.indices.size(); n.b = n.a + primCount;
		if (primCount > 0) {
			c.nonemptyLeafCount++;
			c.indices.insert(c.indices.end(), idx, idx + primCount);
			c.primIndexCount += primCount;
		}
		c.leafCount++;
	}

	// createLeafAfterRetraction (gkdtree.h:1603-1637)
	void leafAfterRetraction(Context &c, uint32_t node, uint32_t start) const {
		const uint32_t indexCount = (uint32_t) c.indices.size() - start;
		std::sort(c.indices.begin() + start, c.indices.end());
		auto last = std::unique(c.indices.begin() + start, c.indices.end());
		const uint32_t nSeen = (uint32_t) (last - (c.indices.begin() + start));
		c.indices.erase(last, c.indices.end());
		c.primIndexCount = c.primIndexCount - indexCount + nSeen;
		PNode &n = c.nodes[node];
		n.kind = 1; n.a = start; n.b = start + nSeen;
		c.nonemptyLeafCount++;
		c.leafCount++;
	}

	// buildTree: exact greedy sweep (gkdtree.h:1898-2345)
	float sweep(Context &c, uint8_t *cls, uint32_t depth, uint32_t node, const Box &nodeBox,
	            std::vector<Event> &events, uint32_t primCount, uint32_t badRefines) const {
		const Event *eventStart = events.data(), *eventEnd = events.data() + events.size();
		const float leafCost = primCount * m_p.queryCost;
		if (primCount <= m_p.stopPrims || depth >= m_p.maxDepth) {
			leafFromEvents(c, node, eventStart, eventEnd, primCount);
			return leafCost;
		}
		// The greedy SAH step (gkdtree.h:1936-2021) in two passes over the sorted events, the way the device phase does it with
		// prefix sums (exactOnDevice): first every distinct (axis, position) becomes one candidate plane with the numbers of
		// primitives that end on it, lie in it and start on it; then a running count per axis turns those into the populations
		// left and right of each plane and the cheapest plane is kept -- the first one among equals, in (axis, position) order.
		const Event *axisStart[3] = { eventEnd, eventEnd, eventEnd };
		std::vector<Candidate> &planes = c.planes;
		planes.clear();
		for (const Event *ev = eventStart; ev < eventEnd; ++ev) {
			if (planes.empty() || planes.back().axis != ev->axis || planes.back().pos != ev->pos) {
				if (planes.empty() || planes.back().axis != ev->axis) axisStart[ev->axis] = ev;
				planes.push_back(Candidate{ ev->pos, (int) ev->axis, { 0, 0, 0 } });
			}
			planes.back().count[ev->type]++;
		}
		const Split best = cheapestPlane(planes, nodeBox, primCount);

		if (best.cost >= leafCost) {
			if ((best.cost > 4 * leafCost && primCount < 16) || badRefines >= m_p.maxBadRefines || best.cost == kInf) {
				leafFromEvents(c, node, eventStart, eventEnd, primCount);
				return leafCost;
			}
			++badRefines;
		}

		// classification wrt. the chosen plane (gkdtree.h:2053-2103)
		enum : uint8_t { kBoth = 0, kLeft = 1, kRight = 2, kBothDone = 3 };
		for (const Event *e = axisStart[best.axis]; e < eventEnd && e->axis == best.axis; ++e)
			cls[e->index] = kBoth;
		uint32_t primsLeft = 0, primsRight = 0, primsBoth = primCount;
		for (const Event *e = axisStart[best.axis]; e < eventEnd && e->axis == best.axis; ++e) {
			if (e->type == kEnd && e->pos <= best.pos) {
				cls[e->index] = kLeft; primsBoth--; primsLeft++;
			} else if (e->type == kStart && e->pos >= best.pos) {
				cls[e->index] = kRight; primsBoth--; primsRight++;
			} else if (e->type == kPlanar) {
				if (e->pos < best.pos || (e->pos == best.pos && best.planarLeft)) {
					cls[e->index] = kLeft; primsBoth--; primsLeft++;
				} else if (e->pos > best.pos || (e->pos == best.pos && !best.planarLeft)) {
					cls[e->index] = kRight; primsBoth--; primsRight++;
				}
			}
		}

		Box leftBox = nodeBox, rightBox = nodeBox;
		leftBox.mx[best.axis] = best.pos;
		rightBox.mn[best.axis] = best.pos;
		uint32_t prunedLeft = 0, prunedRight = 0;
		std::vector<Event> leftEvents, rightEvents;
		if (m_p.clip) {
			std::vector<Event> lt, rt, nl, nr;
			lt.reserve(6 * (size_t) primsLeft); rt.reserve(6 * (size_t) primsRight);
			nl.reserve(6 * (size_t) primsBoth); nr.reserve(6 * (size_t) primsBoth);
			for (const Event *e = eventStart; e < eventEnd; ++e) {
				const uint8_t k = cls[e->index];
				if (k == kLeft) lt.push_back(*e);
				else if (k == kRight) rt.push_back(*e);
				else if (k == kBoth) {
					Box cl, cr;
					const bool vl = m_g.clipped(e->index, leftBox, cl), vr = m_g.clipped(e->index, rightBox, cr);
					if (vl && cl.area() > 0) pushEvents(nl, cl, e->index); else prunedLeft++;
					if (vr && cr.area() > 0) pushEvents(nr, cr, e->index); else prunedRight++;
					cls[e->index] = kBothDone;
				}
			}
			c.pruned += prunedLeft + prunedRight;
			std::sort(nl.begin(), nl.end(), EventLess());
			std::sort(nr.begin(), nr.end(), EventLess());
			leftEvents.resize(lt.size() + nl.size());
			rightEvents.resize(rt.size() + nr.size());
			std::merge(lt.begin(), lt.end(), nl.begin(), nl.end(), leftEvents.begin(), EventLess());
			std::merge(rt.begin(), rt.end(), nr.begin(), nr.end(), rightEvents.begin(), EventLess());
		} else {
			for (const Event *e = eventStart; e < eventEnd; ++e) {
				const uint8_t k = cls[e->index];
				if (k == kLeft) leftEvents.push_back(*e);
				else if (k == kRight) rightEvents.push_back(*e);
				else if (k == kBoth) { leftEvents.push_back(*e); rightEvents.push_back(*e); }
			}
		}
		std::vector<Event>().swap(events);      // the parent's list is no longer needed

		const uint32_t children = c.allocNodes(2);
		const uint32_t nodePosBefore = (uint32_t) c.nodes.size(), indexPosBefore = (uint32_t) c.indices.size();
		const uint32_t leafBefore = c.leafCount, nonemptyBefore = c.nonemptyLeafCount, innerBefore = c.innerCount;
		{ PNode &n = c.nodes[node]; n.kind = 0; n.a = (uint32_t) best.axis; n.b = children; n.split = best.pos; }
		c.innerCount++;

		const float leftCost = sweep(c, cls, depth + 1, children, leftBox, leftEvents, best.numLeft - prunedLeft, badRefines);
		const float rightCost = sweep(c, cls, depth + 1, children + 1, rightBox, rightEvents, best.numRight - prunedRight, badRefines);

		float pl, pr;
		const SAH area(nodeBox);
		area(best.axis, best.pos - nodeBox.mn[best.axis], nodeBox.mx[best.axis] - best.pos, pl, pr);
		const float finalCost = m_p.traversalCost + (pl * leftCost + pr * rightCost);
		if (!m_p.retract || finalCost < primCount * m_p.queryCost)
			return finalCost;
		c.nodes.resize(nodePosBefore);
		c.retracted++;
		c.leafCount = leafBefore; c.nonemptyLeafCount = nonemptyBefore; c.innerCount = innerBefore;
		leafAfterRetraction(c, node, indexPosBefore);
		return leafCost;
	}

	// transitionToNLogN + createEventList (gkdtree.h:1668-1704, :1490-1530)
	float runExact(Context &c, uint8_t *cls, uint32_t depth, uint32_t node, const Box &nodeBox,
	               const std::vector<uint32_t> &prims, uint32_t badRefines) const {
		std::vector<Event> events;
		events.reserve(6 * prims.size());
		uint32_t actual = 0;
		for (uint32_t index : prims) {
			Box b;
			if (m_p.clip) {
				if (!m_g.clipped(index, nodeBox, b) || b.area() == 0)
					continue;
			} else {
				m_g.box(index, b);
			}
			pushEvents(events, b, index);
			++actual;
		}
		std::sort(events.begin(), events.end(), EventLess());
		return sweep(c, cls, depth, node, nodeBox, events, actual, badRefines);
	}

	// MinMaxBins::minimizeCost (gkdtree.h:2405-2510)
	Split minimize(const Box &tight, const float *binSize, const float *invBinSize, const uint32_t *minBins,
	               const uint32_t *maxBins, uint32_t primCount) const {
		Split cand;
		int binIdx = 0, leftBin = 0;
		const int nb = m_p.minMaxBins;
		const SAH tch(tight);
		for (int axis = 0; axis < 3; ++axis) {
			uint32_t numLeft = 0, numRight = primCount;
			float leftWidth = 0, rightWidth = tight.mx[axis] - tight.mn[axis];
			const float bs = binSize[axis];
			for (int i = 0; i < nb - 1; ++i) {
				numLeft += minBins[binIdx];
				numRight -= maxBins[binIdx];
				leftWidth += bs;
				rightWidth -= bs;
				float pl, pr;
				tch(axis, leftWidth, rightWidth, pl, pr);
				const float cost = m_p.traversalCost + m_p.queryCost * (pl * (float) numLeft + pr * (float) numRight);
				if (cost < cand.cost) { cand.cost = cost; cand.axis = axis; cand.numLeft = numLeft; cand.numRight = numRight; leftBin = i; }
				binIdx++;
			}
			binIdx++;
		}
		const int axis = cand.axis;
		const float mn = tight.mn[axis], invBS = invBinSize[axis];
		const float fmax = std::numeric_limits<float>::max();
		float split = mn + (leftBin + 1) * binSize[axis];
		float splitNext = nextafterf(split, fmax);
		int idx = (int) ((split - mn) * invBS), idxNext = (int) ((splitNext - mn) * invBS);
		if (!(idx == leftBin && idxNext == leftBin + 1)) {
			float left = tight.mn[axis], right = tight.mx[axis];
			int it = 0;
			while (true) {
				split = left + (right - left) / 2;
				splitNext = nextafterf(split, fmax);
				idx = (int) ((split - mn) * invBS);
				idxNext = (int) ((splitNext - mn) * invBS);
				if (idx == leftBin && idxNext == leftBin + 1)
					break;
				if (std::abs(idx - idxNext) > 1 || ++it > 50) { cand.cost = kInf; break; }
				if (idx <= leftBin) left = split; else right = split;
			}
		}
		if (split <= tight.mn[axis] || split >= tight.mx[axis])
			cand.cost = kInf;
		cand.pos = split;
		return cand;
	}

	// --- decisions of buildTreeMinMax shared by the host loop and the device binning phase ---
	enum Early { kMakeLeaf, kDefer, kBin };
	Early early(uint32_t primCount, uint32_t depth) const {
		if (primCount <= m_p.stopPrims || depth >= m_p.maxDepth) return kMakeLeaf;
		if (primCount <= m_p.exactPrimThreshold) return kDefer;
		return kBin;
	}
	// what follows MinMaxBins::minimizeCost (gkdtree.h:1779-1800): 0 split, 1 leaf, 2 hand over to the exact method
	int afterMinimize(const Split &best, uint32_t primCount, uint32_t &badRefines) const {
		const float leafCost = primCount * m_p.queryCost;
		if (best.cost == kInf) return 2;
		if (best.cost >= leafCost) {
			if ((best.cost > 4 * leafCost && primCount < 16) || badRefines >= m_p.maxBadRefines) return 1;
			++badRefines;
		}
		return 0;
	}
	void binSetup(const Box &tight, float *binSize, float *invBinSize) const {
		const float recip = 1.0f / (float) m_p.minMaxBins;
		for (int a = 0; a < 3; ++a) { binSize[a] = (tight.mx[a] - tight.mn[a]) * recip; invBinSize[a] = 1 / binSize[a]; }
	}
	// tail of MinMaxBins::partition (gkdtree.h:2560-2596): tight child boxes, split moved onto the nearer of them
	void finishSplit(const Box &tight, Split &best, Box &leftBounds, Box &rightBounds, uint32_t numLeft, uint32_t numRight) const {
		const int axis = best.axis;
		const float splitPos = best.pos;
		leftBounds.clip(tight); rightBounds.clip(tight);
		leftBounds.mx[axis] = std::min(leftBounds.mx[axis], splitPos);
		rightBounds.mn[axis] = std::max(rightBounds.mn[axis], splitPos);
		if (leftBounds.mx[axis] != rightBounds.mn[axis]) {
			const SAH tch(tight);
			const float nL = (float) numLeft, nR = (float) numRight;
			float p1l, p1r, p2l, p2r;
			tch(axis, leftBounds.mx[axis] - tight.mn[axis], tight.mx[axis] - leftBounds.mx[axis], p1l, p1r);
			tch(axis, rightBounds.mn[axis] - tight.mn[axis], tight.mx[axis] - rightBounds.mn[axis], p2l, p2r);
			const float cost1 = m_p.traversalCost + m_p.queryCost * (p1l * nL + p1r * nR);
			const float cost2 = m_p.traversalCost + m_p.queryCost * (p2l * nL + p2r * nR);
			if (cost1 <= cost2) { best.cost = cost1; best.pos = leftBounds.mx[axis]; }
			else { best.cost = cost2; best.pos = rightBounds.mn[axis]; }
			leftBounds.mx[axis] = std::min(leftBounds.mx[axis], best.pos);
			rightBounds.mn[axis] = std::max(rightBounds.mn[axis], best.pos);
		}
	}

	// buildTreeMinMax (gkdtree.h:1735-1867).  Nodes that drop to the exact method become jobs.
	// With a plan (device binning phase, kdbuild_gpu below) the histogram / partition loops are replaced by its records.
	float binned(Context &c, uint32_t depth, uint32_t node, const Box &nodeBox, const Box &tight,
	             std::vector<uint32_t> &prims, uint32_t badRefines, Plan *plan = nullptr, int planNode = -1) {
		PlanNode *pn = plan ? &plan->nodes[planNode] : nullptr;
		if (pn && pn->child[0] < 0)
			prims.swap(pn->prims);                         // terminal record: its primitive list was read back
		const uint32_t primCount = (pn && pn->child[0] >= 0) ? pn->count : (uint32_t) prims.size();
		const float leafCost = primCount * m_p.queryCost;
		switch (early(primCount, depth)) {
			case kMakeLeaf: leafFromIndices(c, node, prims.data(), primCount); return leafCost;
			case kDefer: return defer(c, depth, node, nodeBox, prims, badRefines);
			default: break;
		}

		Split best;
		Box leftBounds, rightBounds; leftBounds.reset(); rightBounds.reset();
		std::vector<uint32_t> leftPrims, rightPrims;
		if (!pn) {
			const int nb = m_p.minMaxBins;
			float binSize[3], invBinSize[3];
			binSetup(tight, binSize, invBinSize);
			std::vector<uint32_t> minBins(3 * (size_t) nb, 0u), maxBins(3 * (size_t) nb, 0u);
			const int64_t maxBin = nb - 1;
			for (uint32_t i = 0; i < primCount; ++i) {
				Box b; m_g.box(prims[i], b);
				for (int a = 0; a < 3; ++a) {
					const int64_t minIdx = binIndex((b.mn[a] - tight.mn[a]) * invBinSize[a]);
					const int64_t maxIdx = binIndex((b.mx[a] - tight.mn[a]) * invBinSize[a]);
					maxBins[a * nb + std::max((int64_t) 0, std::min(maxIdx, maxBin))]++;
					minBins[a * nb + std::max((int64_t) 0, std::min(minIdx, maxBin))]++;
				}
			}
			best = minimize(tight, binSize, invBinSize, minBins.data(), maxBins.data(), primCount);
		} else {
			best = pn->best;
		}
		switch (afterMinimize(best, primCount, badRefines)) {
			case 2: return defer(c, depth, node, nodeBox, prims, badRefines);
			case 1: leafFromIndices(c, node, prims.data(), primCount); return leafCost;
			default: break;
		}

		// MinMaxBins::partition (gkdtree.h:2517-2596)
		if (!pn) {
			const float splitPos = best.pos;
			const int axis = best.axis;
			leftPrims.reserve(best.numLeft); rightPrims.reserve(best.numRight);
			for (uint32_t i = 0; i < primCount; ++i) {
				const uint32_t p = prims[i];
				Box b; m_g.box(p, b);
				if (b.mx[axis] <= splitPos) { leftBounds.expand(b); leftPrims.push_back(p); }
				else if (b.mn[axis] > splitPos) { rightBounds.expand(b); rightPrims.push_back(p); }
				else { leftBounds.expand(b); rightBounds.expand(b); leftPrims.push_back(p); rightPrims.push_back(p); }
			}
			if (leftPrims.size() != best.numLeft || rightPrims.size() != best.numRight)
				throw std::runtime_error("kd-tree build: min-max binning and partition disagree");
			std::vector<uint32_t>().swap(prims);
		} else {
			leftBounds = pn->leftRaw; rightBounds = pn->rightRaw;
		}
		finishSplit(tight, best, leftBounds, rightBounds, best.numLeft, best.numRight);

		const uint32_t children = c.allocNodes(2);
		const uint32_t nodePosBefore = (uint32_t) c.nodes.size(), indexPosBefore = (uint32_t) c.indices.size();
		const uint32_t leafBefore = c.leafCount, nonemptyBefore = c.nonemptyLeafCount, innerBefore = c.innerCount;
		const size_t jobsBefore = m_jobs.size();
		{ PNode &n = c.nodes[node]; n.kind = 0; n.a = (uint32_t) best.axis; n.b = children; n.split = best.pos; }
		c.innerCount++;

		Box childBox = nodeBox;
		childBox.mx[best.axis] = best.pos;
		const float leftCost = binned(c, depth + 1, children, childBox, leftBounds, leftPrims, badRefines, plan, pn ? pn->child[0] : -1);
		childBox.mn[best.axis] = best.pos;
		childBox.mx[best.axis] = nodeBox.mx[best.axis];
		const float rightCost = binned(c, depth + 1, children + 1, childBox, rightBounds, rightPrims, badRefines, plan, pn ? pn->child[1] : -1);

		const SAH tch(nodeBox);
		float pl, pr;
		tch(best.axis, best.pos - nodeBox.mn[best.axis], nodeBox.mx[best.axis] - best.pos, pl, pr);
		const float finalCost = m_p.traversalCost + (pl * leftCost + pr * rightCost);
		if (!m_p.retract || finalCost < primCount * m_p.queryCost)
			return finalCost;
		// only reachable when no job hangs below (their cost is -inf)
		(void) jobsBefore;
		c.nodes.resize(nodePosBefore);
		c.retracted++;
		c.leafCount = leafBefore; c.nonemptyLeafCount = nonemptyBefore; c.innerCount = innerBefore;
		leafAfterRetraction(c, node, indexPosBefore);
		return leafCost;
	}

	// Hand a subtree to the job pool (parallel build) or build it right here (<= threshold scenes)
	float defer(Context &c, uint32_t depth, uint32_t node, const Box &nodeBox, std::vector<uint32_t> &prims, uint32_t badRefines) {
		if (!m_parallel) {
			// scratch classification per primitive id: every entry is written (kBoth) before the sweep reads it, so it
			// is left uninitialised -- zero-filling nPrims bytes per worker page-faults 2 GB at 10 M primitives
			std::unique_ptr<uint8_t[]> cls(new uint8_t[m_nPrims]);
			return runExact(c, cls.get(), depth, node, nodeBox, prims, badRefines);
		}
		m_jobs.emplace_back(new Job());
		Job &j = *m_jobs.back();
		j.depth = depth; j.nodeBox = nodeBox; j.prims.swap(prims); j.badRefines = badRefines;
		PNode &n = c.nodes[node];
		n.kind = 2; n.a = (uint32_t) m_jobs.size() - 1; n.b = 0;
		return -kInf;       // "Never tear down this subtree" (gkdtree.h:1691-1692)
	}

	void runJobs(int nThreads) {
		std::atomic<size_t> next(0);
		// an exception that leaves a std::thread ends the process: the first one is kept and rethrown on the calling
		// thread after the join, where mtsgpu_flatten turns it into an error code
		std::exception_ptr firstError;
		std::mutex errorLock;
		auto worker = [&]() {
			try {
				std::unique_ptr<uint8_t[]> clsBuf(new uint8_t[m_nPrims]);
				uint8_t *cls = clsBuf.get();
				for (;;) {
					const size_t k = next.fetch_add(1);
					if (k >= m_jobs.size())
						break;
					Job &j = *m_jobs[k];
					j.root = j.ctx.allocNodes(1);
					const auto tj0 = std::chrono::steady_clock::now();
					runExact(j.ctx, cls, j.depth, j.root, j.nodeBox, j.prims, j.badRefines);
					j.ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tj0).count();
					std::vector<uint32_t>().swap(j.prims);
				}
			} catch (...) {
				std::lock_guard<std::mutex> guard(errorLock);
				if (!firstError) firstError = std::current_exception();
				next.store(m_jobs.size());           // the other workers stop after their current job
			}
		};
		nThreads = std::max(1, std::min<int>(nThreads, (int) m_jobs.size()));
		std::vector<std::thread> pool;
		for (int t = 1; t < nThreads; ++t) pool.emplace_back(worker);
		worker();
		for (auto &t : pool) t.join();
		if (firstError) std::rethrow_exception(firstError);
	}

	const Geometry &m_g;
	const Params &m_p;
	uint32_t m_nPrims;
	bool m_parallel = false;
	std::vector<std::unique_ptr<Job>> m_jobs;
};


// ---------------------------------------------------------------------------------------------------------------
// Device binning phase.  buildTreeMinMax walks every primitive of a node twice (histogram, partition) on ONE host
// thread, and at >= 10 M triangles that serial walk is most of the build on a many-core host.  Both walks are integer /
// compare work over 24-byte boxes, so they run here level by level: all nodes of one level share three launches
// (histogram, count + child bounds, stable scatter); the 768 counters of each node come back to the host, which runs
// the reference's minimizeCost / split bookkeeping unchanged.  The result is a Plan that binned() replays, so the tree
// is the one the host loop builds, bit for bit (tests/test_gpu_parity.py::test_gpu_binning_builds_the_same_tree).
// ---------------------------------------------------------------------------------------------------------------
constexpr int kKdBlock = 256;
constexpr uint32_t kKdChunk = 2048;          // entries of one node handled by one workgroup
struct DevNode { float tmn[3], inv[3]; int32_t axis; float split; };
struct DevChunk { uint32_t start, count, node, pad; };
struct DevChunkOut { uint32_t nLeft, nRight; float lmn[3], lmx[3], rmn[3], rmx[3]; uint32_t pad[2]; };

#define KDHIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) throw std::runtime_error(std::string("kd-tree build (device binning): ") + hipGetErrorString(e_)); } while (0)

__global__ void k_kd_iota(uint32_t *p, uint32_t n) {
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) p[i] = i;
}

// MinMaxBins::bin (gkdtree.h:2379-2403): hist[node][0..3nb) = minBins, [3nb..6nb) = maxBins
__global__ __launch_bounds__(kKdBlock) void k_kd_bin(const float *__restrict__ boxes, const uint32_t *__restrict__ cur,
		const DevChunk *__restrict__ chunks, const DevNode *__restrict__ nodes, uint32_t *__restrict__ hist, int nb) {
	extern __shared__ uint32_t s_h[];
	const DevChunk ch = chunks[blockIdx.x];
	const DevNode nd = nodes[ch.node];
	for (int i = threadIdx.x; i < 6 * nb; i += kKdBlock) s_h[i] = 0u;
	__syncthreads();
	const int64_t maxBin = nb - 1;
	for (uint32_t e = threadIdx.x; e < ch.count; e += kKdBlock) {
		const float *b = boxes + 6 * (size_t) cur[ch.start + e];
		for (int a = 0; a < 3; ++a) {
			const int64_t minIdx = binIndex((b[a] - nd.tmn[a]) * nd.inv[a]);
			const int64_t maxIdx = binIndex((b[3 + a] - nd.tmn[a]) * nd.inv[a]);
			atomicAdd(&s_h[3 * nb + a * nb + (int) max((int64_t) 0, min(maxIdx, maxBin))], 1u);
			atomicAdd(&s_h[a * nb + (int) max((int64_t) 0, min(minIdx, maxBin))], 1u);
		}
	}
	__syncthreads();
	uint32_t *out = hist + (size_t) ch.node * 6 * nb;
	for (int i = threadIdx.x; i < 6 * nb; i += kKdBlock)
		if (s_h[i]) atomicAdd(&out[i], s_h[i]);
}

// key of a bound for an order-independent reduction that still returns what the sequential std::min / std::max
// chain returns: ties (+0 / -0) go to the earliest entry, whose bits are read back afterwards
__device__ inline uint32_t monoKey(float v) {
	const uint32_t u = __float_as_uint(v + 0.0f);
	return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// MinMaxBins::partition, first half: how many entries of the chunk go left / right, and the boxes they span
__global__ __launch_bounds__(kKdBlock) void k_kd_count(const float *__restrict__ boxes, const uint32_t *__restrict__ cur,
		const DevChunk *__restrict__ chunks, const DevNode *__restrict__ nodes, DevChunkOut *__restrict__ out) {
	__shared__ unsigned long long s_key[12];
	__shared__ uint32_t s_cnt[2];
	const DevChunk ch = chunks[blockIdx.x];
	const DevNode nd = nodes[ch.node];
	if (threadIdx.x < 12) s_key[threadIdx.x] = (threadIdx.x % 6 < 3) ? ~0ull : 0ull;     // mins start high, maxes low
	if (threadIdx.x < 2) s_cnt[threadIdx.x] = 0u;
	__syncthreads();
	unsigned long long key[12];
	for (int k = 0; k < 12; ++k) key[k] = (k % 6 < 3) ? ~0ull : 0ull;
	uint32_t nL = 0, nR = 0;
	for (uint32_t e = threadIdx.x; e < ch.count; e += kKdBlock) {
		const float *b = boxes + 6 * (size_t) cur[ch.start + e];
		const bool leftOnly = b[3 + nd.axis] <= nd.split;
		const bool goesLeft = leftOnly || !(b[nd.axis] > nd.split), goesRight = !leftOnly;
		nL += goesLeft; nR += goesRight;
		for (int k = 0; k < 6; ++k) {
			const unsigned long long hi = (unsigned long long) monoKey(b[k]) << 32;
			const unsigned long long kmin = hi | e, kmax = hi | (0xFFFFFFFFu - e);
			if (k < 3) {
				if (goesLeft && kmin < key[k]) key[k] = kmin;
				if (goesRight && kmin < key[6 + k]) key[6 + k] = kmin;
			} else {
				if (goesLeft && kmax > key[k]) key[k] = kmax;
				if (goesRight && kmax > key[6 + k]) key[6 + k] = kmax;
			}
		}
	}
	for (int k = 0; k < 12; ++k) {
		if (k % 6 < 3) atomicMin(&s_key[k], key[k]); else atomicMax(&s_key[k], key[k]);
	}
	atomicAdd(&s_cnt[0], nL); atomicAdd(&s_cnt[1], nR);
	__syncthreads();
	DevChunkOut &o = out[blockIdx.x];
	if (threadIdx.x < 12) {
		const int k = threadIdx.x, comp = k % 6;
		const unsigned long long kk = s_key[k];
		const bool empty = (comp < 3) ? (kk == ~0ull) : (kk == 0ull);
		float v = (comp < 3) ? INFINITY : -INFINITY;             // Box::reset()
		if (!empty) {
			const uint32_t e = (comp < 3) ? (uint32_t) kk : 0xFFFFFFFFu - (uint32_t) kk;
			v = boxes[6 * (size_t) cur[ch.start + e] + comp];
		}
		float *dst = (k < 6) ? (comp < 3 ? o.lmn : o.lmx) : (comp < 3 ? o.rmn : o.rmx);
		dst[comp % 3] = v;
	}
	if (threadIdx.x == 0) { o.nLeft = s_cnt[0]; o.nRight = s_cnt[1]; }
}

// MinMaxBins::partition, second half: stable scatter (the lists keep the parent's order, as push_back gives it)
__global__ __launch_bounds__(kKdBlock) void k_kd_scatter(const float *__restrict__ boxes, const uint32_t *__restrict__ cur,
		const DevChunk *__restrict__ chunks, const DevNode *__restrict__ nodes, const uint2 *__restrict__ dst, uint32_t *__restrict__ next) {
	__shared__ uint32_t s_w[2][kKdBlock / 64];
	const DevChunk ch = chunks[blockIdx.x];
	const DevNode nd = nodes[ch.node];
	uint32_t baseL = dst[blockIdx.x].x, baseR = dst[blockIdx.x].y;
	const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
	for (uint32_t r = 0; r < ch.count; r += kKdBlock) {
		const uint32_t e = r + threadIdx.x;
		bool goesLeft = false, goesRight = false;
		uint32_t p = 0;
		if (e < ch.count) {
			p = cur[ch.start + e];
			const float *b = boxes + 6 * (size_t) p;
			const bool leftOnly = b[3 + nd.axis] <= nd.split;
			goesLeft = leftOnly || !(b[nd.axis] > nd.split); goesRight = !leftOnly;
		}
		const unsigned long long mL = __ballot(goesLeft), mR = __ballot(goesRight);
		const unsigned long long below = (1ull << lane) - 1ull;
		if (lane == 0) { s_w[0][wave] = (uint32_t) __popcll(mL); s_w[1][wave] = (uint32_t) __popcll(mR); }
		__syncthreads();
		uint32_t offL = 0, offR = 0, totL = 0, totR = 0;
		for (uint32_t w = 0; w < kKdBlock / 64; ++w) {
			if (w < wave) { offL += s_w[0][w]; offR += s_w[1][w]; }
			totL += s_w[0][w]; totR += s_w[1][w];
		}
		if (goesLeft) next[baseL + offL + (uint32_t) __popcll(mL & below)] = p;
		if (goesRight) next[baseR + offR + (uint32_t) __popcll(mR & below)] = p;
		baseL += totL; baseR += totR;
		__syncthreads();
	}
}

template <typename T> struct DevBuf {
	T *p = nullptr; size_t cap = 0;
	~DevBuf() { if (p) (void) hipFree(p); }
	void reserve(size_t n) {
		if (n <= cap) return;
		if (p) { (void) hipFree(p); p = nullptr; cap = 0; }
		KDHIP(hipMalloc((void **) &p, n * sizeof(T)));
		cap = n;
	}
};

// Runs the binning phase of the whole tree on the current HIP device and records it as a Plan
void planOnDevice(const Builder &b, const Params &p, const Geometry &g, uint32_t nPrims, const Box &scene, int nThreads, Plan &plan) {
	int ndev = 0;
	if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
		throw std::runtime_error("kd-tree build: gpu_binning was requested but there is no HIP device");
	const int nb = p.minMaxBins;
	if ((size_t) 6 * nb * sizeof(uint32_t) > 60 * 1024)
		throw std::runtime_error("kd-tree build: gpu_binning supports at most 2560 bins");
	hipStream_t st;
	KDHIP(hipStreamCreate(&st));
	struct StreamGuard { hipStream_t s; ~StreamGuard() { (void) hipStreamDestroy(s); } } guard{ st };

	// the boxes of all primitives, once (host threads), then resident
	std::vector<float> boxes(6 * (size_t) nPrims);
	{
		auto fill = [&](uint32_t lo, uint32_t hi) {
			for (uint32_t i = lo; i < hi; ++i) { Box t; g.box(i, t); std::memcpy(&boxes[6 * (size_t) i], t.mn, 12); std::memcpy(&boxes[6 * (size_t) i + 3], t.mx, 12); }
		};
		const int nt = std::max(1, std::min(nThreads, 16));
		std::vector<std::thread> pool;
		for (int t = 1; t < nt; ++t) pool.emplace_back(fill, (uint32_t) ((uint64_t) nPrims * t / nt), (uint32_t) ((uint64_t) nPrims * (t + 1) / nt));
		fill(0, (uint32_t) ((uint64_t) nPrims / nt));
		for (auto &t : pool) t.join();
	}
	DevBuf<float> dBoxes; dBoxes.reserve(boxes.size());
	KDHIP(hipMemcpyAsync(dBoxes.p, boxes.data(), boxes.size() * sizeof(float), hipMemcpyHostToDevice, st));
	DevBuf<uint32_t> dCur, dNext, dHist;
	dCur.reserve((size_t) nPrims + nPrims / 4 + 1024);
	k_kd_iota<<<(nPrims + 255) / 256, 256, 0, st>>>(dCur.p, nPrims);
	DevBuf<DevNode> dNodes; DevBuf<DevChunk> dChunks; DevBuf<DevChunkOut> dOut; DevBuf<uint2> dDst;

	struct Active { int plan; uint32_t off, count, depth, badRefines; Box tight; };
	std::vector<Active> level, nextLevel;
	plan.nodes.clear();
	plan.nodes.emplace_back();
	plan.nodes[0].count = nPrims;
	auto readBack = [&](int pn, const uint32_t *src, uint32_t off, uint32_t count) {
		plan.nodes[pn].prims.resize(count);
		if (count) KDHIP(hipMemcpyAsync(plan.nodes[pn].prims.data(), src + off, (size_t) count * 4, hipMemcpyDeviceToHost, st));
	};
	if (b.early(nPrims, 1) == Builder::kBin) level.push_back(Active{ 0, 0, nPrims, 1, 0, scene });
	else readBack(0, dCur.p, 0, nPrims);

	std::vector<DevNode> hNodes; std::vector<DevChunk> hChunks; std::vector<uint32_t> hHist;
	std::vector<DevChunkOut> hOut; std::vector<uint2> hDst;
	auto makeChunks = [&](const std::vector<uint32_t> &which) {
		hChunks.clear();
		for (uint32_t k : which)
			for (uint32_t o = 0; o < level[k].count; o += kKdChunk)
				hChunks.push_back(DevChunk{ level[k].off + o, std::min(kKdChunk, level[k].count - o), k, 0 });
		dChunks.reserve(hChunks.size());
		KDHIP(hipMemcpyAsync(dChunks.p, hChunks.data(), hChunks.size() * sizeof(DevChunk), hipMemcpyHostToDevice, st));
	};

	while (!level.empty()) {
		const uint32_t nAct = (uint32_t) level.size();
		// 1. histograms of every node of the level
		hNodes.assign(nAct, DevNode());
		std::vector<float> binSize(3 * (size_t) nAct), invBinSize(3 * (size_t) nAct);
		std::vector<uint32_t> all(nAct);
		for (uint32_t k = 0; k < nAct; ++k) {
			all[k] = k;
			b.binSetup(level[k].tight, &binSize[3 * k], &invBinSize[3 * k]);
			for (int a = 0; a < 3; ++a) { hNodes[k].tmn[a] = level[k].tight.mn[a]; hNodes[k].inv[a] = invBinSize[3 * k + a]; }
		}
		makeChunks(all);
		dNodes.reserve(nAct);
		KDHIP(hipMemcpyAsync(dNodes.p, hNodes.data(), nAct * sizeof(DevNode), hipMemcpyHostToDevice, st));
		dHist.reserve((size_t) nAct * 6 * nb);
		KDHIP(hipMemsetAsync(dHist.p, 0, (size_t) nAct * 6 * nb * 4, st));
		k_kd_bin<<<(uint32_t) hChunks.size(), kKdBlock, 6 * nb * sizeof(uint32_t), st>>>(dBoxes.p, dCur.p, dChunks.p, dNodes.p, dHist.p, nb);
		KDHIP(hipGetLastError());
		hHist.resize((size_t) nAct * 6 * nb);
		KDHIP(hipMemcpyAsync(hHist.data(), dHist.p, hHist.size() * 4, hipMemcpyDeviceToHost, st));
		KDHIP(hipStreamSynchronize(st));

		// 2. minimizeCost per node (host, 3 x 127 candidates each); nodes that stop here hand their list back
		std::vector<uint32_t> splitting;
		size_t nextSize = 0;
		std::vector<uint32_t> childOff(2 * (size_t) nAct, 0u);
		for (uint32_t k = 0; k < nAct; ++k) {
			Active &a = level[k];
			const uint32_t *h = &hHist[(size_t) k * 6 * nb];
			PlanNode &pn = plan.nodes[a.plan];
			pn.best = b.minimize(a.tight, &binSize[3 * k], &invBinSize[3 * k], h, h + 3 * nb, a.count);
			if (b.afterMinimize(pn.best, a.count, a.badRefines) != 0) {
				readBack(a.plan, dCur.p, a.off, a.count);
				continue;
			}
			hNodes[k].axis = pn.best.axis; hNodes[k].split = pn.best.pos;
			childOff[2 * k] = (uint32_t) nextSize; childOff[2 * k + 1] = (uint32_t) nextSize + pn.best.numLeft;
			nextSize += (size_t) pn.best.numLeft + pn.best.numRight;
			if (nextSize > 0xFFFFFFFFull) throw std::runtime_error("kd-tree build: a level of the binning phase exceeds 2^32 entries");
			splitting.push_back(k);
		}
		if (splitting.empty()) { KDHIP(hipStreamSynchronize(st)); break; }

		// 3. count + child bounds per chunk, offsets by a host scan over the chunks, stable scatter
		makeChunks(splitting);
		KDHIP(hipMemcpyAsync(dNodes.p, hNodes.data(), nAct * sizeof(DevNode), hipMemcpyHostToDevice, st));
		dOut.reserve(hChunks.size());
		k_kd_count<<<(uint32_t) hChunks.size(), kKdBlock, 0, st>>>(dBoxes.p, dCur.p, dChunks.p, dNodes.p, dOut.p);
		KDHIP(hipGetLastError());
		hOut.resize(hChunks.size());
		KDHIP(hipMemcpyAsync(hOut.data(), dOut.p, hOut.size() * sizeof(DevChunkOut), hipMemcpyDeviceToHost, st));
		KDHIP(hipStreamSynchronize(st));
		hDst.resize(hChunks.size());
		std::vector<uint32_t> gotL(nAct, 0u), gotR(nAct, 0u);
		for (uint32_t k : splitting) { PlanNode &pn = plan.nodes[level[k].plan]; pn.leftRaw.reset(); pn.rightRaw.reset(); }
		for (size_t cidx = 0; cidx < hChunks.size(); ++cidx) {
			const uint32_t k = hChunks[cidx].node;
			PlanNode &pn = plan.nodes[level[k].plan];
			const DevChunkOut &o = hOut[cidx];
			hDst[cidx] = make_uint2(childOff[2 * k] + gotL[k], childOff[2 * k + 1] + gotR[k]);
			gotL[k] += o.nLeft; gotR[k] += o.nRight;
			// the chunks of a node arrive in list order, so this is the sequential expand() chain of the host loop
			Box l, r;
			std::memcpy(l.mn, o.lmn, 12); std::memcpy(l.mx, o.lmx, 12); std::memcpy(r.mn, o.rmn, 12); std::memcpy(r.mx, o.rmx, 12);
			pn.leftRaw.expand(l); pn.rightRaw.expand(r);
		}
		for (uint32_t k : splitting) {
			const PlanNode &pn = plan.nodes[level[k].plan];
			if (gotL[k] != pn.best.numLeft || gotR[k] != pn.best.numRight)
				throw std::runtime_error("kd-tree build: min-max binning and partition disagree");
		}
		dDst.reserve(hDst.size());
		KDHIP(hipMemcpyAsync(dDst.p, hDst.data(), hDst.size() * sizeof(uint2), hipMemcpyHostToDevice, st));
		dNext.reserve(nextSize + 1024);
		k_kd_scatter<<<(uint32_t) hChunks.size(), kKdBlock, 0, st>>>(dBoxes.p, dCur.p, dChunks.p, dNodes.p, dDst.p, dNext.p);
		KDHIP(hipGetLastError());

		// 4. children: tight boxes as the host loop computes them; those that leave the phase read their list back
		nextLevel.clear();
		for (uint32_t k : splitting) {
			const Active a = level[k];
			Split best = plan.nodes[a.plan].best;
			Box lb = plan.nodes[a.plan].leftRaw, rb = plan.nodes[a.plan].rightRaw;
			b.finishSplit(a.tight, best, lb, rb, best.numLeft, best.numRight);
			const uint32_t counts[2] = { best.numLeft, best.numRight };
			const Box *boxes2[2] = { &lb, &rb };
			for (int side = 0; side < 2; ++side) {
				const int child = (int) plan.nodes.size();
				plan.nodes.emplace_back();
				plan.nodes[child].count = counts[side];
				plan.nodes[a.plan].child[side] = child;
				if (b.early(counts[side], a.depth + 1) == Builder::kBin)
					nextLevel.push_back(Active{ child, childOff[2 * k + side], counts[side], a.depth + 1, a.badRefines, *boxes2[side] });
				else
					readBack(child, dNext.p, childOff[2 * k + side], counts[side]);
			}
		}
		KDHIP(hipStreamSynchronize(st));              // read-backs done before the buffers swap roles
		std::swap(dCur.p, dNext.p); std::swap(dCur.cap, dNext.cap);
		level.swap(nextLevel);
	}
	KDHIP(hipStreamSynchronize(st));
}

// ---------------------------------------------------------------------------------------------------------------
// Device exact phase: buildTree (gkdtree.h:1898-2345), the O(n log n) sweep for nodes of at most exactPrimThreshold
// primitives, for ALL such subtrees ("jobs") at once, one tree level per round.
//
// The reference keeps one event list per node, sorted by (axis, position, type), and re-uses it down the tree with
// stable partitions and merges.  Here a node is the contiguous run of its primitive instances (primitive id + the box
// the primitive has inside the node), always in ascending primitive order, and every level
//   1. emits the edge events of all instances (createEventList, :1490-1530) with a 58-bit key
//      node | axis | position | type and sorts them with ONE stable radix sort (rocPRIM): events of equal key keep
//      the instance order, i.e. the primitive order, which is the tie rule of EventLess above;
//   2. turns the sweep (:1936-2021) into prefix sums over the sorted events: the counts in front of and up to a group
//      of events at one position give numLeft / numRight / numPlanar of that candidate plane, its cost is evaluated by
//      the group's last event with the reference's expressions, and an atomic minimum over (cost, event index) per node
//      finds the candidate the sequential loop keeps (the first of the cheapest);
//   3. lets the HOST take the decisions of :2023-2051 (leaf, bad refines) from that record with the same code path;
//   4. classifies every instance against the plane (:2053-2103), re-clips the straddlers in binary64
//      (Triangle::getClippedAABB, :2140-2219) and scatters the instances to the children with prefix sums (stable).
// Leaves get their primitive order on the host from the boxes of their instances.  When the last level is done the
// host replays the nodes depth first with the bookkeeping of buildTree (cost of the subtree, retraction :2327-2341,
// counters) into the Context of every job, so everything downstream -- and the tree -- is what runExact produces.
// ---------------------------------------------------------------------------------------------------------------
struct XInst { uint32_t prim; float mn[3], mx[3]; uint32_t node; };        // 32 B; node = index inside its level
// A node of the exact phase, resident on the device.  The nodes of one level are contiguous, the children of a level's
// split nodes form the next level (left, right, in node order).
struct XNodeG {
	float mn[3], mx[3];              // the node's box
	uint32_t primCount;              // what the sweep counts numRight down from
	uint32_t instBegin, instCount;   // its instances in the level's instance array
	uint32_t depth, badRefines;
	uint32_t sweep, isSplit, isLeaf;
	int32_t axis; float split; uint32_t planarLeft;
	uint32_t numLeft, numRight;      // of the chosen plane (gkdtree.h:1950-2016)
	uint32_t child;                  // global index of the left child (the right one follows)
	uint32_t childBase[2];           // where the children's instances start in the next level's array
	uint32_t pruned[2];              // straddlers whose clipped box in the left / right child is empty
	uint32_t leafOffset;             // a leaf's primitives in the leaf item buffer
	float cost; uint32_t retract;    // filled bottom-up when all levels are done
};
// what the host needs of a node for the replay
struct XNodeH { uint32_t primCount; int32_t kind; int32_t axis; float split; uint32_t child, pruned, leafOffset, leafCount, retract; };
struct XTotals { uint32_t nSplit, nNextInst, nLeafItems, pad; };
struct XParams { float traversalCost, queryCost, emptySpaceBonus; int clip; uint32_t stopPrims, maxDepth, maxBadRefines; int retract; };

constexpr uint64_t kXInvalid = ~0ull;
constexpr uint32_t kXMaxNodes = 1u << 22;       // nodes per level (22 bits of the event key)

#define KXHIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) throw std::runtime_error(std::string("kd-tree build (device exact phase): ") + hipGetErrorString(e_)); } while (0)

__device__ inline uint32_t xMono(float v) {
	const uint32_t u = __float_as_uint(v + 0.0f);         // -0 and +0 compare equal in EventLess
	return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ inline uint32_t xOrderable(float v) {
	const uint32_t u = __float_as_uint(v);
	return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// the box a primitive has inside `to` (Geometry::clipped / Geometry::box); triPos = 9 floats per primitive, a
// non-triangle primitive has NaN in its first slot and its box in genBox
__device__ inline bool xClipped(const float *triPos, const float *genBox, uint32_t prim, const float *tmn, const float *tmx, float *omn, float *omx) {
	const float *t = triPos + 9 * (size_t) prim;
	if (t[0] != t[0]) {
		const float *g = genBox + 6 * (size_t) prim;
		for (int a = 0; a < 3; ++a) { omn[a] = hdMax(g[a], tmn[a]); omx[a] = hdMin(g[3 + a], tmx[a]); }
		for (int a = 0; a < 3; ++a) if (omx[a] < omn[a]) return false;
		return true;
	}
	return clippedTriangleBoxHD(t, t + 3, t + 6, tmn, tmx, omn, omx);
}
__device__ inline void xBox(const float *triPos, const float *genBox, uint32_t prim, float *omn, float *omx) {
	const float *t = triPos + 9 * (size_t) prim;
	if (t[0] != t[0]) {
		const float *g = genBox + 6 * (size_t) prim;
		for (int a = 0; a < 3; ++a) { omn[a] = g[a]; omx[a] = g[3 + a]; }
		return;
	}
	for (int a = 0; a < 3; ++a) { omn[a] = INFINITY; omx[a] = -INFINITY; }
	for (int v = 0; v < 3; ++v)
		for (int a = 0; a < 3; ++a) { omn[a] = hdMin(omn[a], t[3 * v + a]); omx[a] = hdMax(omx[a], t[3 * v + a]); }
}
__device__ inline float xArea(const float *mn, const float *mx) {
	const float dx = mx[0] - mn[0], dy = mx[1] - mn[1], dz = mx[2] - mn[2];
	return (float) 2.0 * (dx * dy + dx * dz + dy * dz);
}

// transitionToNLogN (gkdtree.h:1668-1704): the instances of the jobs' primitives; keep[i] = 0 drops a primitive whose
// clipped box is empty or has no area
__global__ void k_x_init(const float *triPos, const float *genBox, const uint32_t *prims, const uint32_t *primNode, uint32_t n,
                         const XNodeG *nodes, int clip, XInst *out, uint32_t *keep) {
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	XInst x; x.prim = prims[i]; x.node = primNode[i];
	const XNodeG &nd = nodes[x.node];
	bool ok = true;
	if (clip) ok = xClipped(triPos, genBox, x.prim, nd.mn, nd.mx, x.mn, x.mx) && xArea(x.mn, x.mx) != 0;
	else xBox(triPos, genBox, x.prim, x.mn, x.mx);
	out[i] = x; keep[i] = ok ? 1u : 0u;
}
__global__ void k_x_compact(const XInst *in, const uint32_t *keep, const uint32_t *scan, uint32_t n, XInst *out) {
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n && keep[i]) out[scan[i] - 1u] = in[i];
}
// the roots after the compaction: their instance ranges from the inclusive scan of the keep flags
__global__ void k_x_roots(XNodeG *nodes, uint32_t n, const uint32_t *scan, XTotals *totals) {
	const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
	if (k >= n) return;
	XNodeG &nd = nodes[k];
	const uint32_t b = nd.instBegin, c = nd.instCount;
	const uint32_t before = b ? scan[b - 1] : 0u, kept = c ? scan[b + c - 1] - before : 0u;
	nd.instBegin = before; nd.instCount = kept; nd.primCount = kept;
	if (k == n - 1) { totals->nNextInst = before + kept; totals->nSplit = 0; totals->nLeafItems = 0; }
}

// the leaves that need no sweep (gkdtree.h:1903-1906)
__global__ void k_x_prep(XNodeG *nodes, uint32_t n, XParams prm) {
	const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
	if (k >= n) return;
	XNodeG &nd = nodes[k];
	const bool preLeaf = nd.primCount <= prm.stopPrims || nd.depth >= prm.maxDepth;
	nd.sweep = preLeaf ? 0u : 1u; nd.isLeaf = preLeaf ? 1u : 0u; nd.isSplit = 0u;
	nd.pruned[0] = nd.pruned[1] = 0u;
}

// createEventList: two key slots per instance and axis (a planar primitive uses one)
__global__ void k_x_emit(const XInst *inst, uint32_t n, const XNodeG *nodes, unsigned long long *keys, uint32_t *vals) {
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	const XInst x = inst[i];
	const bool sweep = nodes[x.node].sweep != 0u;
	for (int a = 0; a < 3; ++a) {
		unsigned long long k0 = kXInvalid, k1 = kXInvalid;
		if (sweep) {
			const unsigned long long hi = (unsigned long long) (x.node * 4u + (uint32_t) a) << 34;
			if (x.mn[a] == x.mx[a]) k0 = hi | ((unsigned long long) xMono(x.mn[a]) << 2) | (unsigned long long) kPlanar;
			else {
				k0 = hi | ((unsigned long long) xMono(x.mn[a]) << 2) | (unsigned long long) kStart;
				k1 = hi | ((unsigned long long) xMono(x.mx[a]) << 2) | (unsigned long long) kEnd;
			}
		}
		keys[6 * (size_t) i + 2 * a] = k0; keys[6 * (size_t) i + 2 * a + 1] = k1;
		vals[6 * (size_t) i + 2 * a] = i; vals[6 * (size_t) i + 2 * a + 1] = i;
	}
}
// per sorted event: the (index + 1) of the event that opens its group, and where the segments (node, axis) start; the
// three per-type counters of the sweep are read off the keys by the scans themselves (XIsType)
__global__ void k_x_flags(const unsigned long long *keys, uint32_t n, uint32_t *head, uint32_t *segStart) {
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	const unsigned long long k = keys[i];
	const bool valid = k != kXInvalid;
	const unsigned long long prev = i ? keys[i - 1] : kXInvalid;
	head[i] = (valid && (i == 0 || (prev >> 2) != (k >> 2))) ? i + 1u : 0u;
	if (valid && (i == 0 || (prev >> 34) != (k >> 34))) segStart[(uint32_t) (k >> 34)] = i;
}

struct XIsType {
	uint32_t type;
	__host__ __device__ uint32_t operator()(unsigned long long k) const { return (k != kXInvalid && (uint32_t) (k & 3ull) == type) ? 1u : 0u; }
};

// the candidate plane of the group of events that ends at sorted position i (gkdtree.h:1950-2016)
struct XCand { bool valid; float cost, pos; uint32_t numLeft, numRight, planarLeft; int axis; uint32_t node; };
__device__ inline XCand xCandidate(uint32_t i, const unsigned long long *keys, const uint32_t *vals, const uint32_t *E, const uint32_t *P,
                                   const uint32_t *S, const uint32_t *H, const uint32_t *segStart, const XInst *inst, const XNodeG *nodes,
                                   const XParams prm) {
	XCand c; c.valid = false;
	const unsigned long long k = keys[i];
	const uint32_t seg = (uint32_t) (k >> 34);
	c.node = seg >> 2; c.axis = (int) (seg & 3u);
	const XNodeG &nd = nodes[c.node];
	const uint32_t h = H[i] - 1u, s0 = segStart[seg];
	const uint32_t Eh = h ? E[h - 1] : 0u, Ph = h ? P[h - 1] : 0u, Sh = h ? S[h - 1] : 0u;
	const uint32_t Es = s0 ? E[s0 - 1] : 0u, Ps = s0 ? P[s0 - 1] : 0u, Ss = s0 ? S[s0 - 1] : 0u;
	(void) Eh;
	const uint32_t numPlanar = P[i] - Ph;
	const uint32_t nL = (Sh - Ss) + (Ph - Ps);                     // starts and planars of the groups in front
	const uint32_t nR = nd.primCount - ((P[i] - Ps) + (E[i] - Es)); // minus planars and ends up to and including this group
	// the position is that of the group's first event (ev->pos), taken from the box: the key has lost the sign of zero
	const XInst &hx = inst[vals[h]];
	const uint32_t htype = (uint32_t) (keys[h] & 3ull);
	const float pos = htype == kEnd ? hx.mx[c.axis] : hx.mn[c.axis];
	c.pos = pos;
	const float nmn = nd.mn[c.axis], nmx = nd.mx[c.axis];
	if (!(pos > nmn && pos < nmx)) return c;
	// SurfaceAreaHeuristic (sahkdtree3.h:35-79)
	const float e0 = nd.mx[0] - nd.mn[0], e1 = nd.mx[1] - nd.mn[1], e2 = nd.mx[2] - nd.mn[2];
	const float temp = 1.0f / (e0 * e1 + e1 * e2 + e0 * e2);
	const float t0 = (c.axis == 0 ? (e1 * e2) : c.axis == 1 ? (e0 * e2) : (e0 * e1)) * temp;
	const float t1 = (c.axis == 0 ? (e1 + e2) : c.axis == 1 ? (e0 + e2) : (e0 + e1)) * temp;
	const float pl = t0 + t1 * (pos - nmn), pr = t0 + t1 * (nmx - pos);
	const float nLF = (float) nL, nRF = (float) nR;
	if (numPlanar == 0) {
		float cost = prm.traversalCost + prm.queryCost * (pl * nLF + pr * nRF);
		if (nL == 0 || nR == 0) cost *= prm.emptySpaceBonus;
		c.cost = cost; c.numLeft = nL; c.numRight = nR; c.planarLeft = 0;
	} else {
		float costPlanarLeft = prm.traversalCost + prm.queryCost * (pl * (float) (nL + numPlanar) + pr * nRF);
		float costPlanarRight = prm.traversalCost + prm.queryCost * (pl * nLF + pr * (float) (nR + numPlanar));
		if (nL + numPlanar == 0 || nR == 0) costPlanarLeft *= prm.emptySpaceBonus;
		if (nL == 0 || nR + numPlanar == 0) costPlanarRight *= prm.emptySpaceBonus;
		if (costPlanarLeft < costPlanarRight) { c.cost = costPlanarLeft; c.numLeft = nL + numPlanar; c.numRight = nR; c.planarLeft = 1; }
		else { c.cost = costPlanarRight; c.numLeft = nL; c.numRight = nR + numPlanar; c.planarLeft = 0; }
	}
	c.valid = c.cost == c.cost;       // a NaN cost never wins "cost < best.cost"
	return c;
}
__global__ void k_x_cost(const unsigned long long *keys, const uint32_t *vals, uint32_t n, const uint32_t *E, const uint32_t *P, const uint32_t *S,
                         const uint32_t *H, const uint32_t *segStart, const XInst *inst, const XNodeG *nodes, XParams prm,
                         unsigned long long *nodeBest) {
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	const unsigned long long k = keys[i];
	if (k == kXInvalid) return;
	const unsigned long long next = (i + 1 < n) ? keys[i + 1] : kXInvalid;
	if (next != kXInvalid && (next >> 2) == (k >> 2)) return;          // not the last event of its group
	const XCand c = xCandidate(i, keys, vals, E, P, S, H, segStart, inst, nodes, prm);
	if (!c.valid) return;
	atomicMin(&nodeBest[c.node], ((unsigned long long) xOrderable(c.cost) << 32) | (unsigned long long) i);
}
// what follows the sweep (gkdtree.h:2023-2051): a leaf after all, a "bad refine", or the split
__global__ void k_x_decide(XNodeG *nodes, uint32_t nNodes, const unsigned long long *nodeBest, const unsigned long long *keys, const uint32_t *vals,
                           const uint32_t *E, const uint32_t *P, const uint32_t *S, const uint32_t *H, const uint32_t *segStart, const XInst *inst,
                           XParams prm) {
	const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
	if (k >= nNodes) return;
	XNodeG &nd = nodes[k];
	if (!nd.sweep) return;
	float cost = INFINITY, pos = 0; int axis = 0; uint32_t numLeft = 0, numRight = 0, planarLeft = 0;
	const unsigned long long v = nodeBest[k];
	if (v != ~0ull) {
		const XCand c = xCandidate((uint32_t) v, keys, vals, E, P, S, H, segStart, inst, nodes, prm);
		cost = c.cost; pos = c.pos; axis = c.axis; numLeft = c.numLeft; numRight = c.numRight; planarLeft = c.planarLeft;
	}
	const float leafCost = nd.primCount * prm.queryCost;
	bool leaf = false;
	if (cost >= leafCost) {
		if ((cost > 4 * leafCost && nd.primCount < 16) || nd.badRefines >= prm.maxBadRefines || cost == INFINITY) leaf = true;
		else nd.badRefines++;
	}
	if (leaf) { nd.isLeaf = 1u; return; }
	nd.isSplit = 1u; nd.axis = axis; nd.split = pos; nd.planarLeft = planarLeft; nd.numLeft = numLeft; nd.numRight = numRight;
}

// classification wrt. the chosen plane (gkdtree.h:2053-2103) and the boxes of the straddlers inside the children
// (perfect splits, :2140-2219)
__global__ void k_x_classify(const XInst *inst, uint32_t n, XNodeG *nodes, const float *triPos, const float *genBox, int clip,
                             uint32_t *goesL, uint32_t *goesR, uint32_t *isLeafInst, float *boxL, float *boxR) {
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	const XInst x = inst[i];
	XNodeG &nd = nodes[x.node];
	uint32_t L = 0, R = 0;
	isLeafInst[i] = nd.isLeaf;
	if (nd.isSplit) {
		const int a = nd.axis;
		const float split = nd.split, mn = x.mn[a], mx = x.mx[a];
		bool both = false;
		if (mn == mx) {
			if (mn < split || (mn == split && nd.planarLeft)) L = 1;
			else if (mn > split || (mn == split && !nd.planarLeft)) R = 1;
			else both = true;
		} else if (mx <= split) L = 1;
		else if (mn >= split) R = 1;
		else both = true;
		for (int c = 0; c < 3; ++c) { boxL[6 * (size_t) i + c] = x.mn[c]; boxL[6 * (size_t) i + 3 + c] = x.mx[c]; boxR[6 * (size_t) i + c] = x.mn[c]; boxR[6 * (size_t) i + 3 + c] = x.mx[c]; }
		if (both) {
			L = R = 1;
			if (clip) {
				float lmn[3], lmx[3], rmn[3], rmx[3], cmn[3], cmx[3];
				for (int c = 0; c < 3; ++c) { lmn[c] = rmn[c] = nd.mn[c]; lmx[c] = rmx[c] = nd.mx[c]; }
				lmx[a] = split; rmn[a] = split;
				if (xClipped(triPos, genBox, x.prim, lmn, lmx, cmn, cmx) && xArea(cmn, cmx) > 0) { for (int c = 0; c < 3; ++c) { boxL[6 * (size_t) i + c] = cmn[c]; boxL[6 * (size_t) i + 3 + c] = cmx[c]; } }
				else { L = 0; atomicAdd(&nd.pruned[0], 1u); }
				if (xClipped(triPos, genBox, x.prim, rmn, rmx, cmn, cmx) && xArea(cmn, cmx) > 0) { for (int c = 0; c < 3; ++c) { boxR[6 * (size_t) i + c] = cmn[c]; boxR[6 * (size_t) i + 3 + c] = cmx[c]; } }
				else { R = 0; atomicAdd(&nd.pruned[1], 1u); }
			}
		}
	}
	goesL[i] = L; goesR[i] = R;
}
// per node: how many instances go to either child (from the inclusive scans of the flags)
__global__ void k_x_counts(const XNodeG *nodes, uint32_t n, const uint32_t *sL, const uint32_t *sR, uint32_t *splitFlag, uint32_t *c2) {
	const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
	if (k >= n) return;
	const XNodeG &nd = nodes[k];
	const uint32_t b = nd.instBegin, c = nd.instCount;
	uint32_t nL = 0, nR = 0;
	if (nd.isSplit && c) { nL = sL[b + c - 1] - (b ? sL[b - 1] : 0u); nR = sR[b + c - 1] - (b ? sR[b - 1] : 0u); }
	splitFlag[k] = nd.isSplit; c2[2 * k] = nL; c2[2 * k + 1] = nR;
}
// the children of the split nodes = the next level (left, right, in node order); leaves learn where their items go
__global__ void k_x_children(XNodeG *nodes, uint32_t n, XNodeG *next, uint32_t nextGlobal, const uint32_t *sSplit, const uint32_t *c2, const uint32_t *sC2,
                             const uint32_t *sLeafInst, uint32_t nInst, uint32_t leafBase, XTotals *totals) {
	const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
	if (k >= n) return;
	XNodeG &nd = nodes[k];
	if (nd.isSplit) {
		const uint32_t r = sSplit[k] - 1u;
		nd.child = nextGlobal + 2u * r;
		for (int side = 0; side < 2; ++side) {
			XNodeG c;
			for (int a = 0; a < 3; ++a) { c.mn[a] = nd.mn[a]; c.mx[a] = nd.mx[a]; }
			if (side == 0) c.mx[nd.axis] = nd.split; else c.mn[nd.axis] = nd.split;
			c.primCount = (side == 0 ? nd.numLeft : nd.numRight) - nd.pruned[side];
			c.instCount = c2[2 * k + side];
			c.instBegin = sC2[2 * k + side] - c.instCount;
			c.depth = nd.depth + 1u; c.badRefines = nd.badRefines;
			c.sweep = c.isSplit = c.isLeaf = 0u; c.axis = 0; c.split = 0; c.planarLeft = 0; c.numLeft = c.numRight = 0;
			c.child = 0; c.childBase[0] = c.childBase[1] = 0; c.pruned[0] = c.pruned[1] = 0; c.leafOffset = 0; c.cost = 0; c.retract = 0;
			nd.childBase[side] = c.instBegin;
			next[2u * r + side] = c;
		}
	} else {
		// a leaf: its instances are contiguous in the level's compaction of leaf instances
		nd.leafOffset = leafBase + (nd.instBegin ? sLeafInst[nd.instBegin - 1] : 0u);
	}
	if (k == n - 1) {
		totals->nSplit = sSplit[n - 1]; totals->nNextInst = sC2[2 * n - 1];
		totals->nLeafItems = nInst ? sLeafInst[nInst - 1] : 0u;
	}
}
// instances to the children (stable: ascending primitive order is kept), leaf instances to the item buffer as
// (leaf, axis-0 position, type) keys in the order leafFromEvents needs after ONE stable sort at the end
__global__ void k_x_scatter(const XInst *inst, uint32_t n, const XNodeG *nodes, uint32_t levelGlobal, const uint32_t *goesL, const uint32_t *goesR,
                            const uint32_t *scanL, const uint32_t *scanR, const uint32_t *isLeafInst, const uint32_t *scanLeaf, const float *boxL,
                            const float *boxR, XInst *next, uint32_t leafBase, unsigned long long *leafKeys, uint32_t *leafPrims) {
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	const XInst x = inst[i];
	const XNodeG &nd = nodes[x.node];
	const uint32_t b = nd.instBegin;
	if (goesL[i]) {
		XInst o; o.prim = x.prim; o.node = 2u * (nd.child - (levelGlobal + 0u)) ; // placeholder, fixed below
		o.node = 0;
		for (int c = 0; c < 3; ++c) { o.mn[c] = boxL[6 * (size_t) i + c]; o.mx[c] = boxL[6 * (size_t) i + 3 + c]; }
		o.node = nd.child - levelGlobal;          // index inside the next level (levelGlobal = global index of its first node)
		next[nd.childBase[0] + (scanL[i] - 1u - (b ? scanL[b - 1] : 0u))] = o;
	}
	if (goesR[i]) {
		XInst o; o.prim = x.prim;
		for (int c = 0; c < 3; ++c) { o.mn[c] = boxR[6 * (size_t) i + c]; o.mx[c] = boxR[6 * (size_t) i + 3 + c]; }
		o.node = nd.child + 1u - levelGlobal;
		next[nd.childBase[1] + (scanR[i] - 1u - (b ? scanR[b - 1] : 0u))] = o;
	}
	if (isLeafInst[i]) {
		const uint32_t pos = leafBase + scanLeaf[i] - 1u;
		const uint32_t type = x.mn[0] == x.mx[0] ? (uint32_t) kPlanar : (uint32_t) kStart;
		// (global node id of the leaf) | position | type: node ids grow with the buffer position
		leafKeys[pos] = ((unsigned long long) (nd.leafOffset) << 34) | ((unsigned long long) xMono(x.mn[0]) << 2) | (unsigned long long) type;
		leafPrims[pos] = x.prim;
	}
}
// cost of the subtrees and retraction (gkdtree.h:2321-2341), one level at a time from the deepest one up
__global__ void k_x_cost_up(XNodeG *all, uint32_t levelBase, uint32_t n, XParams prm) {
	const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
	if (k >= n) return;
	XNodeG &nd = all[levelBase + k];
	const float leafCost = nd.primCount * prm.queryCost;
	nd.retract = 0u;
	if (!nd.isSplit) { nd.cost = leafCost; return; }
	const float leftCost = all[nd.child].cost, rightCost = all[nd.child + 1u].cost;
	const float e0 = nd.mx[0] - nd.mn[0], e1 = nd.mx[1] - nd.mn[1], e2 = nd.mx[2] - nd.mn[2];
	const float temp = 1.0f / (e0 * e1 + e1 * e2 + e0 * e2);
	const int a = nd.axis;
	const float t0 = (a == 0 ? (e1 * e2) : a == 1 ? (e0 * e2) : (e0 * e1)) * temp;
	const float t1 = (a == 0 ? (e1 + e2) : a == 1 ? (e0 + e2) : (e0 + e1)) * temp;
	const float pl = t0 + t1 * (nd.split - nd.mn[a]), pr = t0 + t1 * (nd.mx[a] - nd.split);
	const float finalCost = prm.traversalCost + (pl * leftCost + pr * rightCost);
	if (!prm.retract || finalCost < leafCost) { nd.cost = finalCost; return; }
	nd.cost = leafCost; nd.retract = 1u;
}
__global__ void k_x_host_view(const XNodeG *all, uint32_t n, XNodeH *out) {
	const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
	if (k >= n) return;
	const XNodeG &nd = all[k];
	XNodeH h;
	h.primCount = nd.primCount; h.kind = nd.isSplit ? 0 : 1; h.axis = nd.axis; h.split = nd.split; h.child = nd.child;
	h.pruned = nd.pruned[0] + nd.pruned[1]; h.leafOffset = nd.leafOffset; h.leafCount = nd.isSplit ? 0u : nd.instCount; h.retract = nd.retract;
	out[k] = h;
}

struct XScan {
	DevBuf<unsigned char> tmp;
	void sum(const uint32_t *in, uint32_t *out, size_t n, hipStream_t st) {
		size_t bytes = 0;
		KXHIP(rocprim::inclusive_scan(nullptr, bytes, in, out, n, rocprim::plus<uint32_t>(), st));
		tmp.reserve(bytes + 16);
		KXHIP(rocprim::inclusive_scan((void *) tmp.p, bytes, in, out, n, rocprim::plus<uint32_t>(), st));
	}
	// inclusive count of the events of one type, straight from the sorted keys
	void countType(const unsigned long long *keys, uint32_t type, uint32_t *out, size_t n, hipStream_t st) {
		auto in = rocprim::make_transform_iterator(keys, XIsType{ type });
		size_t bytes = 0;
		KXHIP(rocprim::inclusive_scan(nullptr, bytes, in, out, n, rocprim::plus<uint32_t>(), st));
		tmp.reserve(bytes + 16);
		KXHIP(rocprim::inclusive_scan((void *) tmp.p, bytes, in, out, n, rocprim::plus<uint32_t>(), st));
	}
	void max(const uint32_t *in, uint32_t *out, size_t n, hipStream_t st) {
		size_t bytes = 0;
		KXHIP(rocprim::inclusive_scan(nullptr, bytes, in, out, n, rocprim::maximum<uint32_t>(), st));
		tmp.reserve(bytes + 16);
		KXHIP(rocprim::inclusive_scan((void *) tmp.p, bytes, in, out, n, rocprim::maximum<uint32_t>(), st));
	}
};
// a device buffer that keeps its contents when it grows
template <typename T> void growKeep(DevBuf<T> &b, size_t used, size_t need, hipStream_t st) {
	if (need <= b.cap) return;
	DevBuf<T> n;
	n.reserve(std::max(need, 2 * b.cap));
	if (used) KXHIP(hipMemcpyAsync(n.p, b.p, used * sizeof(T), hipMemcpyDeviceToDevice, st));
	KXHIP(hipStreamSynchronize(st));
	std::swap(b.p, n.p); std::swap(b.cap, n.cap);
}

// Builds the subtrees of all jobs on the current HIP device; fills job.ctx (root = node 0 of the context)
void exactOnDevice(const Builder &b, const Params &p, const Geometry &g, uint32_t nPrims, std::vector<std::unique_ptr<Job>> &jobs) {
	int ndev = 0;
	if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
		throw std::runtime_error("kd-tree build: the device exact phase was requested but there is no HIP device");
	if (jobs.empty()) return;
	hipStream_t st;
	KXHIP(hipStreamCreate(&st));
	struct StreamGuard { hipStream_t s; ~StreamGuard() { (void) hipStreamDestroy(s); } } guard{ st };
	const int B = 256;
	auto grid = [&](size_t n) { return dim3((unsigned) ((n + B - 1) / B)); };
	const XParams prm{ p.traversalCost, p.queryCost, p.emptySpaceBonus, p.clip ? 1 : 0, p.stopPrims, p.maxDepth, p.maxBadRefines, p.retract ? 1 : 0 };
	const bool timing = std::getenv("MTSGPU_KDTIMING") != nullptr;
	double tm[4] = { 0, 0, 0, 0 };      // setup, levels, download, replay
	auto now = []() { return std::chrono::steady_clock::now(); };
	auto since = [](std::chrono::steady_clock::time_point t) { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t).count(); };
	auto tPhase = now();
	const int nThreads = std::max(1, std::min<int>(16, (int) std::thread::hardware_concurrency()));

	// geometry: nine floats per triangle (NaN marks a non-triangle primitive, whose box is in genBox)
	DevBuf<float> dTri, dGen;
	{
		std::unique_ptr<float[]> tpBuf(new float[9 * (size_t) nPrims]);
		float *tp = tpBuf.get();
		std::atomic<bool> anyGenA(false);
		{
			std::vector<std::thread> pool;
			for (int t = 0; t < nThreads; ++t) pool.emplace_back([&, t]() {
				for (uint32_t i = (uint32_t) ((uint64_t) nPrims * t / nThreads); i < (uint32_t) ((uint64_t) nPrims * (t + 1) / nThreads); ++i) {
					const uint32_t *tr = g.tri + 3 * (size_t) i;
					if (tr[0] == MTSGPU_KNOTRIANGLE) { for (int q = 0; q < 9; ++q) tp[9 * (size_t) i + q] = std::numeric_limits<float>::quiet_NaN(); anyGenA = true; continue; }
					for (int v = 0; v < 3; ++v) std::memcpy(&tp[9 * (size_t) i + 3 * v], g.vtx + 3 * (size_t) tr[v], 12);
				}
			});
			for (auto &th : pool) th.join();
		}
		const bool anyGen = anyGenA;
		dTri.reserve(9 * (size_t) nPrims);
		KXHIP(hipMemcpyAsync(dTri.p, tp, 9 * (size_t) nPrims * sizeof(float), hipMemcpyHostToDevice, st));
		dGen.reserve(anyGen ? 6 * (size_t) nPrims : 6);
		if (anyGen) KXHIP(hipMemcpyAsync(dGen.p, g.genBox, 6 * (size_t) nPrims * sizeof(float), hipMemcpyHostToDevice, st));
		KXHIP(hipStreamSynchronize(st));
	}

	// ---- the roots: one per job ----
	size_t nInit = 0;
	std::vector<XNodeG> roots(jobs.size());
	for (size_t j = 0; j < jobs.size(); ++j) {
		XNodeG n; std::memset(&n, 0, sizeof(n));
		for (int a = 0; a < 3; ++a) { n.mn[a] = jobs[j]->nodeBox.mn[a]; n.mx[a] = jobs[j]->nodeBox.mx[a]; }
		n.depth = jobs[j]->depth; n.badRefines = jobs[j]->badRefines;
		n.instBegin = (uint32_t) nInit; n.instCount = (uint32_t) jobs[j]->prims.size();
		nInit += jobs[j]->prims.size();
		roots[j] = n;
	}
	if (jobs.size() > kXMaxNodes) throw std::runtime_error("kd-tree build (device exact phase): too many subtrees");
	if (nInit >= (1ull << 31)) throw std::runtime_error("kd-tree build (device exact phase): too many primitives");

	DevBuf<XNodeG> dAll;                            // every node, level after level
	size_t nAll = jobs.size();
	dAll.reserve(jobs.size() + nInit + nInit / 4 + 1024);      // about 0.55 nodes per primitive in practice; grows if needed
	KXHIP(hipMemcpyAsync(dAll.p, roots.data(), roots.size() * sizeof(XNodeG), hipMemcpyHostToDevice, st));
	DevBuf<XInst> dInstA, dInstB;
	DevBuf<uint32_t> dU[12], dV[8];
	DevBuf<unsigned long long> dKeysA, dKeysB, dNodeBest, dLeafKeys, dLeafKeys2;
	DevBuf<uint32_t> dValsA, dValsB, dSegStart, dLeafPrims, dLeafPrims2;
	DevBuf<float> dBoxL, dBoxR;
	DevBuf<XTotals> dTotals;
	DevBuf<unsigned char> dSortTmp;
	XScan scan;
	dTotals.reserve(1);
	XTotals totals{};
	auto fetchTotals = [&]() {
		KXHIP(hipGetLastError());        // a failed launch is not sticky: catch it before its outputs are trusted
		KXHIP(hipMemcpyAsync(&totals, dTotals.p, sizeof(XTotals), hipMemcpyDeviceToHost, st));
		KXHIP(hipStreamSynchronize(st));
	};

	// ---- level 0: the instances of the jobs' primitive lists, clipped to the jobs' boxes ----
	size_t nInst = 0;
	{
		std::vector<uint32_t> hp(nInit), hn(nInit);
		size_t o = 0;
		for (size_t j = 0; j < jobs.size(); ++j)
			for (uint32_t prim : jobs[j]->prims) { hp[o] = prim; hn[o] = (uint32_t) j; ++o; }
		// every buffer gets its size once, with room for the straddlers that are duplicated on the way down (a
		// reallocation inside the level loop costs a device synchronisation; it still happens if a level outgrows this)
		const size_t instCap = nInit + nInit / 4 + 1024;
		for (int k = 0; k < 6; ++k) dU[k].reserve(instCap);
		for (int k = 7; k < 12; ++k) dU[k].reserve(6 * instCap);
		dInstA.reserve(instCap); dInstB.reserve(instCap);
		dKeysA.reserve(6 * instCap); dKeysB.reserve(6 * instCap); dValsA.reserve(6 * instCap); dValsB.reserve(6 * instCap);
		dBoxL.reserve(6 * instCap); dBoxR.reserve(6 * instCap);
		for (int k = 0; k < 4; ++k) dV[k].reserve(2 * std::min<size_t>(instCap, kXMaxNodes) + 1);
		dSegStart.reserve(4 * std::min<size_t>(instCap, kXMaxNodes)); dNodeBest.reserve(std::min<size_t>(instCap, kXMaxNodes));
		KXHIP(hipMemcpyAsync(dU[0].p, hp.data(), nInit * 4, hipMemcpyHostToDevice, st));
		KXHIP(hipMemcpyAsync(dU[1].p, hn.data(), nInit * 4, hipMemcpyHostToDevice, st));
		if (nInit) {
			hipLaunchKernelGGL(k_x_init, grid(nInit), dim3(B), 0, st, dTri.p, dGen.p, dU[0].p, dU[1].p, (uint32_t) nInit, dAll.p, prm.clip, dInstB.p, dU[2].p);
			scan.sum(dU[2].p, dU[3].p, nInit, st);
			hipLaunchKernelGGL(k_x_compact, grid(nInit), dim3(B), 0, st, dInstB.p, dU[2].p, dU[3].p, (uint32_t) nInit, dInstA.p);
			hipLaunchKernelGGL(k_x_roots, grid(jobs.size()), dim3(B), 0, st, dAll.p, (uint32_t) jobs.size(), dU[3].p, dTotals.p);
			fetchTotals();
			nInst = totals.nNextInst;
		} else {
			KXHIP(hipStreamSynchronize(st));
		}
	}
	tm[0] = since(tPhase); tPhase = now();

	// ---- the levels ----
	std::vector<std::pair<size_t, size_t>> levels;       // (global index of the first node, nodes)
	size_t levelBase = 0, nA = jobs.size(), leafTotal = 0;
	dLeafKeys.reserve(nInit + nInit / 4 + 1024); dLeafPrims.reserve(nInit + nInit / 4 + 1024);
	while (nA > 0) {
		if (nA > kXMaxNodes) throw std::runtime_error("kd-tree build (device exact phase): more than 2^22 nodes in one level");
		levels.emplace_back(levelBase, nA);
		XNodeG *nodes = dAll.p + levelBase;
		hipLaunchKernelGGL(k_x_prep, grid(nA), dim3(B), 0, st, nodes, (uint32_t) nA, prm);
		const size_t nEv = 6 * nInst;
		if (nEv >= (1ull << 32)) throw std::runtime_error("kd-tree build (device exact phase): more than 2^32 edge events in one level");
		if (nInst) {
			dKeysA.reserve(nEv); dKeysB.reserve(nEv); dValsA.reserve(nEv); dValsB.reserve(nEv);
			for (int k = 7; k < 12; ++k) dU[k].reserve(nEv + 1);
			dSegStart.reserve(4 * nA); dNodeBest.reserve(nA);
			hipLaunchKernelGGL(k_x_emit, grid(nInst), dim3(B), 0, st, dInstA.p, (uint32_t) nInst, nodes, dKeysA.p, dValsA.p);
			{
				size_t bytes = 0;
				// node | axis | position | type: 34 + 2 + the bits the node ids of this level need (unused slots are all ones,
				// which no event is because axis 3 does not exist: they sort last)
				unsigned endBit = 36;
				while (endBit < 58 && ((nA - 1) >> (endBit - 36))) ++endBit;
				KXHIP(rocprim::radix_sort_pairs(nullptr, bytes, dKeysA.p, dKeysB.p, dValsA.p, dValsB.p, nEv, 0u, endBit, st));
				dSortTmp.reserve(bytes + 16);
				KXHIP(rocprim::radix_sort_pairs((void *) dSortTmp.p, bytes, dKeysA.p, dKeysB.p, dValsA.p, dValsB.p, nEv, 0u, endBit, st));
			}
			uint32_t *hd = dU[7].p, *sE = dU[8].p, *sP = dU[9].p, *sS = dU[10].p, *sH = dU[11].p;
			KXHIP(hipMemsetAsync(dSegStart.p, 0, 4 * nA * sizeof(uint32_t), st));
			KXHIP(hipMemsetAsync(dNodeBest.p, 0xFF, nA * sizeof(unsigned long long), st));
			hipLaunchKernelGGL(k_x_flags, grid(nEv), dim3(B), 0, st, dKeysB.p, (uint32_t) nEv, hd, dSegStart.p);
			scan.countType(dKeysB.p, kEnd, sE, nEv, st); scan.countType(dKeysB.p, kPlanar, sP, nEv, st); scan.countType(dKeysB.p, kStart, sS, nEv, st);
			scan.max(hd, sH, nEv, st);
			hipLaunchKernelGGL(k_x_cost, grid(nEv), dim3(B), 0, st, dKeysB.p, dValsB.p, (uint32_t) nEv, sE, sP, sS, sH, dSegStart.p, dInstA.p, nodes, prm, dNodeBest.p);
			hipLaunchKernelGGL(k_x_decide, grid(nA), dim3(B), 0, st, nodes, (uint32_t) nA, dNodeBest.p, dKeysB.p, dValsB.p, sE, sP, sS, sH, dSegStart.p, dInstA.p, prm);
		}
		// classification, clipping, children
		for (int k = 0; k < 6; ++k) dU[k].reserve(nInst + 1);
		for (int k = 0; k < 4; ++k) dV[k].reserve(2 * nA + 1);
		dBoxL.reserve(6 * nInst + 6); dBoxR.reserve(6 * nInst + 6);
		uint32_t *gL = dU[0].p, *gR = dU[1].p, *lf = dU[2].p, *sL = dU[3].p, *sR = dU[4].p, *sLf = dU[5].p;
		uint32_t *splitFlag = dV[0].p, *sSplit = dV[1].p, *c2 = dV[2].p, *sC2 = dV[3].p;
		if (nInst) {
			hipLaunchKernelGGL(k_x_classify, grid(nInst), dim3(B), 0, st, dInstA.p, (uint32_t) nInst, nodes, dTri.p, dGen.p, prm.clip, gL, gR, lf, dBoxL.p, dBoxR.p);
			scan.sum(gL, sL, nInst, st); scan.sum(gR, sR, nInst, st); scan.sum(lf, sLf, nInst, st);
		}
		hipLaunchKernelGGL(k_x_counts, grid(nA), dim3(B), 0, st, nodes, (uint32_t) nA, sL, sR, splitFlag, c2);
		scan.sum(splitFlag, sSplit, nA, st); scan.sum(c2, sC2, 2 * nA, st);
		// room for the next level's nodes (at most two per node)
		if (nAll + 2 * nA > dAll.cap) { growKeep(dAll, nAll, nAll + 2 * nA, st); nodes = dAll.p + levelBase; }
		hipLaunchKernelGGL(k_x_children, grid(nA), dim3(B), 0, st, nodes, (uint32_t) nA, dAll.p + nAll, (uint32_t) nAll, sSplit, c2, sC2, sLf, (uint32_t) nInst,
		                   (uint32_t) leafTotal, dTotals.p);
		fetchTotals();
		const size_t nNext = totals.nNextInst, nLeafItems = totals.nLeafItems;
		if (leafTotal + nLeafItems >= (1ull << 30)) throw std::runtime_error("kd-tree build (device exact phase): too many leaf entries");
		if (nInst) {
			dInstB.reserve(nNext + 1);
			growKeep(dLeafKeys, leafTotal, leafTotal + nLeafItems + 1, st); growKeep(dLeafPrims, leafTotal, leafTotal + nLeafItems + 1, st);
			hipLaunchKernelGGL(k_x_scatter, grid(nInst), dim3(B), 0, st, dInstA.p, (uint32_t) nInst, nodes, (uint32_t) nAll, gL, gR, sL, sR, lf, sLf,
			                   dBoxL.p, dBoxR.p, dInstB.p, (uint32_t) leafTotal, dLeafKeys.p, dLeafPrims.p);
		}
		std::swap(dInstA.p, dInstB.p); std::swap(dInstA.cap, dInstB.cap);
		leafTotal += nLeafItems;
		nInst = nNext;
		levelBase = nAll;
		nA = 2 * (size_t) totals.nSplit;
		nAll += nA;
	}
	tm[1] = since(tPhase); tPhase = now();

	// ---- leaf order, subtree costs, download ----
	// leafFromEvents lists the axis-0 start / planar events in EventLess order: one stable sort of all leaf items by
	// (leaf, position, type); equal keys keep the ascending primitive order of the instances
	std::vector<uint32_t> leafStore(leafTotal);
	if (leafTotal) {
		dLeafKeys2.reserve(leafTotal); dLeafPrims2.reserve(leafTotal);
		size_t bytes = 0;
		KXHIP(rocprim::radix_sort_pairs(nullptr, bytes, dLeafKeys.p, dLeafKeys2.p, dLeafPrims.p, dLeafPrims2.p, leafTotal, 0u, 64u, st));
		dSortTmp.reserve(bytes + 16);
		KXHIP(rocprim::radix_sort_pairs((void *) dSortTmp.p, bytes, dLeafKeys.p, dLeafKeys2.p, dLeafPrims.p, dLeafPrims2.p, leafTotal, 0u, 64u, st));
		KXHIP(hipMemcpyAsync(leafStore.data(), dLeafPrims2.p, leafTotal * 4, hipMemcpyDeviceToHost, st));
	}
	for (size_t l = levels.size(); l-- > 0;)
		hipLaunchKernelGGL(k_x_cost_up, grid(levels[l].second), dim3(B), 0, st, dAll.p, (uint32_t) levels[l].first, (uint32_t) levels[l].second, prm);
	DevBuf<XNodeH> dView;
	dView.reserve(nAll);
	hipLaunchKernelGGL(k_x_host_view, grid(nAll), dim3(B), 0, st, dAll.p, (uint32_t) nAll, dView.p);
	std::vector<XNodeH> all(nAll);
	KXHIP(hipGetLastError());
	KXHIP(hipMemcpyAsync(all.data(), dView.p, nAll * sizeof(XNodeH), hipMemcpyDeviceToHost, st));
	KXHIP(hipStreamSynchronize(st));
	// what the replay indexes with: child ids inside the node table, leaf ranges inside the leaf store
	for (size_t i = 0; i < nAll; ++i) {
		const XNodeH &n = all[i];
		if (n.kind == 1 ? (n.primCount > 0 && (size_t) n.leafOffset + n.leafCount > leafTotal)
		                : ((size_t) n.child + 1 >= nAll || (size_t) n.child <= i))
			throw std::runtime_error("kd-tree build (device exact phase): inconsistent node table read back from the device");
	}
	tm[2] = since(tPhase); tPhase = now();

	// ---- replay: the bookkeeping of buildTree per job, depth first (nodes, indices, counters, retraction) ----
	struct Emit {
		const Builder &b; const Params &p; const std::vector<XNodeH> &all; const std::vector<uint32_t> &leafStore;
		void run(Context &c, uint32_t id, uint32_t node) {
			const XNodeH &n = all[id];
			if (n.kind == 1) {
				PNode &pn = c.nodes[node];
				pn.kind = 1; pn.a = (uint32_t) c.indices.size(); pn.b = pn.a + n.primCount;
				if (n.primCount > 0) {
					c.nonemptyLeafCount++;
					c.indices.insert(c.indices.end(), leafStore.begin() + (ptrdiff_t) n.leafOffset, leafStore.begin() + (ptrdiff_t) (n.leafOffset + n.leafCount));
					c.primIndexCount += n.primCount;
				}
				c.leafCount++;
				return;
			}
			c.pruned += n.pruned;
			const uint32_t children = c.allocNodes(2);
			const uint32_t nodePosBefore = (uint32_t) c.nodes.size(), indexPosBefore = (uint32_t) c.indices.size();
			const uint32_t leafBefore = c.leafCount, nonemptyBefore = c.nonemptyLeafCount, innerBefore = c.innerCount;
			{ PNode &pn = c.nodes[node]; pn.kind = 0; pn.a = (uint32_t) n.axis; pn.b = children; pn.split = n.split; }
			c.innerCount++;
			run(c, n.child, children);
			run(c, n.child + 1u, children + 1);
			if (!n.retract)
				return;
			c.nodes.resize(nodePosBefore);
			c.retracted++;
			c.leafCount = leafBefore; c.nonemptyLeafCount = nonemptyBefore; c.innerCount = innerBefore;
			b.leafAfterRetraction(c, node, indexPosBefore);
		}
	} emit{ b, p, all, leafStore };
	{
		std::atomic<size_t> nextJob(0);
		std::exception_ptr firstError;       // an exception must not leave a std::thread (see runJobs)
		std::mutex errorLock;
		auto worker = [&]() {
			try {
				for (;;) {
					const size_t j = nextJob.fetch_add(1);
					if (j >= jobs.size()) break;
					Job &job = *jobs[j];
					job.root = job.ctx.allocNodes(1);
					emit.run(job.ctx, (uint32_t) j, job.root);
					std::vector<uint32_t>().swap(job.prims);
				}
			} catch (...) {
				std::lock_guard<std::mutex> guard(errorLock);
				if (!firstError) firstError = std::current_exception();
				nextJob.store(jobs.size());
			}
		};
		const int T = (int) std::min<size_t>((size_t) nThreads, jobs.size());
		std::vector<std::thread> pool;
		for (int t = 1; t < T; ++t) pool.emplace_back(worker);
		worker();
		for (auto &th : pool) th.join();
		if (firstError) std::rethrow_exception(firstError);
	}
	tm[3] = since(tPhase);
	if (timing)
		std::fprintf(stderr, "[kdbuild] device exact phase: %zu levels, %zu nodes, %zu leaf entries; setup %.1f, levels %.1f, leaf sort + costs + download %.1f, replay %.1f ms\n",
		             levels.size(), nAll, leafTotal, tm[0], tm[1], tm[2], tm[3]);
}

int log2i(uint32_t v) { int r = 0; while (v >>= 1) r++; return r; }

} // namespace

void buildKdTree(const float *vtx, const uint32_t *tri, uint32_t nTris, const float *genBox, const mtsgpu_kd_params *kp, KdTree &out) {
	Params p;
	p.traversalCost = (kp && kp->traversal_cost > 0) ? kp->traversal_cost : 15;      // gkdtree.h:711-724
	p.queryCost = (kp && kp->query_cost > 0) ? kp->query_cost : 20;
	p.emptySpaceBonus = (kp && kp->empty_space_bonus > 0) ? kp->empty_space_bonus : 0.9f;
	p.stopPrims = (kp && kp->stop_prims > 0) ? (uint32_t) kp->stop_prims : 6;
	p.maxBadRefines = (kp && kp->max_bad_refines > 0) ? (uint32_t) kp->max_bad_refines : 3;
	p.exactPrimThreshold = (kp && kp->exact_prim_threshold > 0) ? (uint32_t) kp->exact_prim_threshold : 65536;
	p.minMaxBins = (kp && kp->min_max_bins > 1) ? kp->min_max_bins : 128;
	p.clip = !(kp && kp->clip < 0);
	p.retract = !(kp && kp->retract < 0);
	p.maxDepth = (kp && kp->max_depth > 0) ? (uint32_t) kp->max_depth : 0;
	int nThreads = (kp && kp->n_threads > 0) ? kp->n_threads : (int) std::thread::hardware_concurrency();

	out = KdTree();
	if (nTris == 0) {
		out.nodes = { 0x80000000u, 0u };
		return;
	}
	if (p.maxDepth == 0)
		p.maxDepth = (uint32_t) (int) (8 + 1.3f * log2i(nTris));                       // gkdtree.h:945-947
	p.maxDepth = std::min(p.maxDepth, 48u);

	const Geometry g{ vtx, tri, genBox };
	Builder b(g, p, nTris);
	// gpu_binning is a bit set: 1 = min-max binning phase on the device, 2 = exact phase on the device
	const bool exactOnDev = kp && (kp->gpu_binning & 2) != 0;
	b.m_parallel = nTris > p.exactPrimThreshold || exactOnDev;      // the device builds the exact subtrees as jobs
	Box scene; scene.reset();
	std::vector<uint32_t> prims(nTris);
	for (uint32_t i = 0; i < nTris; ++i) { Box t; g.box(i, t); scene.expand(t); prims[i] = i; }

	Context root;
	const uint32_t prelimRoot = root.allocNodes(1);
	const bool timing = std::getenv("MTSGPU_KDTIMING") != nullptr;
	const auto t0 = std::chrono::steady_clock::now();
	const bool onDevice = kp && (kp->gpu_binning & 1) != 0 && nTris > p.exactPrimThreshold;
	if (onDevice) {
		Plan plan;
		std::vector<uint32_t>().swap(prims);
		planOnDevice(b, p, g, nTris, scene, nThreads, plan);
		b.binned(root, 1, prelimRoot, scene, scene, prims, 0, &plan, 0);
	} else {
		b.binned(root, 1, prelimRoot, scene, scene, prims, 0);
	}
	const auto t1 = std::chrono::steady_clock::now();
	if (exactOnDev) exactOnDevice(b, p, g, nTris, b.m_jobs);
	else b.runJobs(nThreads);
	const auto t2 = std::chrono::steady_clock::now();
	if (timing) {
		double mx = 0, sum = 0;
		for (auto &j : b.m_jobs) { mx = std::max(mx, j->ms); sum += j->ms; }
		std::fprintf(stderr, "[kdbuild] jobs: longest %.1f ms, mean %.1f ms\n", mx, b.m_jobs.empty() ? 0.0 : sum / b.m_jobs.size());
	}
	if (timing)
		std::fprintf(stderr, "[kdbuild] %u prims: binning phase (%s) %.1f ms (%zu jobs), exact phase %.1f ms on %d threads\n", nTris, onDevice ? "device" : "host",
		             std::chrono::duration<double, std::milli>(t1 - t0).count(), b.m_jobs.size(),
		             std::chrono::duration<double, std::milli>(t2 - t1).count(), nThreads);

	// final layout (gkdtree.h:1042-1138): depth first, left child first, siblings adjacent
	uint32_t inner = root.innerCount, leaves = root.leafCount, nIdx = root.primIndexCount;
	for (auto &j : b.m_jobs) { inner += j->ctx.innerCount; leaves += j->ctx.leafCount; nIdx += j->ctx.primIndexCount; }
	const uint32_t nodeCount = inner + leaves;
	out.nodes.assign(2 * (size_t) nodeCount, 0u);
	out.indices.assign(nIdx, 0u);
	struct Item { const Context *ctx; uint32_t node, target; Box box; };
	std::vector<Item> stack;
	uint32_t nodePtr = 0, indexPtr = 0;
	float expTraversalSteps = 0, expLeavesVisited = 0, expPrimitivesIntersected = 0;
	stack.push_back(Item{ &root, prelimRoot, nodePtr++, scene });
	while (!stack.empty()) {
		Item it = stack.back(); stack.pop_back();
		const PNode *n = &it.ctx->nodes[it.node];
		if (n->kind == 2) {
			const Job &j = *b.m_jobs[n->a];
			it.ctx = &j.ctx; it.node = j.root;
			n = &it.ctx->nodes[it.node];
		}
		uint32_t *target = &out.nodes[2 * (size_t) it.target];
		if (n->kind == 1) {
			const uint32_t primsInLeaf = n->b - n->a;
			target[0] = 0x80000000u | indexPtr;
			target[1] = indexPtr + primsInLeaf;
			const float quantity = it.box.area();
			expLeavesVisited += quantity;
			expPrimitivesIntersected += quantity * primsInLeaf;
			for (uint32_t k = n->a; k < n->b; ++k)
				out.indices[indexPtr++] = it.ctx->indices[k];
		} else {
			expTraversalSteps += it.box.area();
			const uint32_t children = nodePtr;
			nodePtr += 2;
			const int axis = (int) n->a;
			target[0] = (uint32_t) axis | ((children - it.target) << 2);
			std::memcpy(&target[1], &n->split, 4);
			Box box = it.box;
			const float tmp = box.mn[axis];
			box.mn[axis] = n->split;
			stack.push_back(Item{ it.ctx, n->b + 1, children + 1, box });
			box.mn[axis] = tmp;
			box.mx[axis] = n->split;
			stack.push_back(Item{ it.ctx, n->b, children, box });
		}
	}
	if (nodePtr != nodeCount || indexPtr != nIdx)
		throw std::runtime_error("kd-tree build: layout pass is inconsistent");
	const float rootQuantity = scene.area();
	out.stats[0] = inner; out.stats[1] = leaves; out.stats[2] = nIdx;
	out.stats[3] = expTraversalSteps / rootQuantity;
	out.stats[4] = expLeavesVisited / rootQuantity;
	out.stats[5] = expPrimitivesIntersected / rootQuantity;

	// slightly enlarge the box (gkdtree.h:1170-1176); max uses the already-moved min
	for (int a = 0; a < 3; ++a) out.aabbMin[a] = scene.mn[a] - ((scene.mx[a] - scene.mn[a]) * kEps + kEps);
	for (int a = 0; a < 3; ++a) out.aabbMax[a] = scene.mx[a] + ((scene.mx[a] - out.aabbMin[a]) * kEps + kEps);
}

} // namespace mg
