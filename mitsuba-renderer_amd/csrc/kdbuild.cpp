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
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstring>
#include <limits>
#include <thread>

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

// Sutherland-Hodgman against one plane, double precision (triangle.cpp:61-106)
int clipPlane(const double (*in)[3], int inCount, double (*out)[3], int axis, double splitPos, bool isMinimum) {
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
			std::memcpy(out[outCount++], next, sizeof(next));
		} else if (curIsInside != nextIsInside) {
			const double t = (splitPos - cur[axis]) / (next[axis] - cur[axis]);
			for (int c = 0; c < 3; ++c) out[outCount][c] = cur[c] + (next[c] - cur[c]) * t;
			out[outCount][axis] = splitPos;
			outCount++;
			if (nextIsInside)
				std::memcpy(out[outCount++], next, sizeof(next));
		}
		std::memcpy(cur, next, sizeof(next));
		curIsInside = nextIsInside;
	}
	return outCount;
}

} // namespace

// Triangle::getClippedAABB (triangle.cpp:108-158): clip in double, round outward to float
bool clippedTriangleBox(const float *p0, const float *p1, const float *p2, const float *bmin, const float *bmax,
                        float *omin, float *omax) {
	double a[10][3], b[10][3];
	for (int c = 0; c < 3; ++c) { a[0][c] = p0[c]; a[1][c] = p1[c]; a[2][c] = p2[c]; }
	int n = 3;
	for (int axis = 0; axis < 3; ++axis) {
		n = clipPlane(a, n, b, axis, (double) bmin[axis], true);
		n = clipPlane(b, n, a, axis, (double) bmax[axis], false);
	}
	for (int c = 0; c < 3; ++c) { omin[c] = kInf; omax[c] = -kInf; }
	for (int i = 0; i < n; ++i)
		for (int j = 0; j < 3; ++j) {
			const double pos_d = a[i][j];
			const float pos_f = (float) pos_d;
			float lo, hi;
			if (pos_f < pos_d) { lo = pos_f; hi = nextafterf(pos_f, kInf); }
			else if (pos_f > pos_d) { hi = pos_f; lo = nextafterf(pos_f, -kInf); }
			else lo = hi = pos_f;
			omin[j] = std::min(omin[j], lo);
			omax[j] = std::max(omax[j], hi);
		}
	for (int c = 0; c < 3; ++c) { omin[c] = std::max(omin[c], bmin[c]); omax[c] = std::min(omax[c], bmax[c]); }
	for (int c = 0; c < 3; ++c)
		if (omax[c] < omin[c])
			return false;
	return true;
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
};

struct Job { uint32_t depth; Box nodeBox; std::vector<uint32_t> prims; uint32_t badRefines; Context ctx; uint32_t root; };

class Builder {
public:
	Builder(const Geometry &g, const Params &p, uint32_t nPrims) : m_g(g), m_p(p), m_nPrims(nPrims) {}

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
	float sweep(Context &c, std::vector<uint8_t> &cls, uint32_t depth, uint32_t node, const Box &nodeBox,
	            std::vector<Event> &events, uint32_t primCount, uint32_t badRefines) const {
		const Event *eventStart = events.data(), *eventEnd = events.data() + events.size();
		const float leafCost = primCount * m_p.queryCost;
		if (primCount <= m_p.stopPrims || depth >= m_p.maxDepth) {
			leafFromEvents(c, node, eventStart, eventEnd, primCount);
			return leafCost;
		}
		Split best;
		uint32_t numLeft[3] = { 0, 0, 0 }, numRight[3] = { primCount, primCount, primCount };
		const Event *axisStart[3] = { eventStart, eventEnd, eventEnd };
		int axisCtr = 1;
		const SAH tch(nodeBox);
		for (const Event *ev = eventStart; ev < eventEnd;) {
			const int axis = ev->axis;
			const float pos = ev->pos;
			uint32_t numStart = 0, numEnd = 0, numPlanar = 0;
			while (ev < eventEnd && ev->pos == pos && ev->axis == axis && ev->type == kEnd) { ++numEnd; ++ev; }
			while (ev < eventEnd && ev->pos == pos && ev->axis == axis && ev->type == kPlanar) { ++numPlanar; ++ev; }
			while (ev < eventEnd && ev->pos == pos && ev->axis == axis && ev->type == kStart) { ++numStart; ++ev; }
			if (ev < eventEnd && ev->axis != axis)
				axisStart[axisCtr++] = ev;
			numRight[axis] -= numPlanar + numEnd;
			if (pos > nodeBox.mn[axis] && pos < nodeBox.mx[axis]) {
				const uint32_t nL = numLeft[axis], nR = numRight[axis];
				const float nLF = (float) nL, nRF = (float) nR;
				float pl, pr;
				tch(axis, pos - nodeBox.mn[axis], nodeBox.mx[axis] - pos, pl, pr);
				if (numPlanar == 0) {
					float cost = m_p.traversalCost + m_p.queryCost * (pl * nLF + pr * nRF);
					if (nL == 0 || nR == 0)
						cost *= m_p.emptySpaceBonus;
					if (cost < best.cost) { best.pos = pos; best.axis = axis; best.cost = cost; best.numLeft = nL; best.numRight = nR; }
				} else {
					float costPlanarLeft = m_p.traversalCost + m_p.queryCost * (pl * (float) (nL + numPlanar) + pr * nRF);
					float costPlanarRight = m_p.traversalCost + m_p.queryCost * (pl * nLF + pr * (float) (nR + numPlanar));
					if (nL + numPlanar == 0 || nR == 0) costPlanarLeft *= m_p.emptySpaceBonus;
					if (nL == 0 || nR + numPlanar == 0) costPlanarRight *= m_p.emptySpaceBonus;
					if (costPlanarLeft < best.cost || costPlanarRight < best.cost) {
						best.pos = pos; best.axis = axis;
						if (costPlanarLeft < costPlanarRight) {
							best.cost = costPlanarLeft; best.numLeft = nL + numPlanar; best.numRight = nR; best.planarLeft = true;
						} else {
							best.cost = costPlanarRight; best.numLeft = nL; best.numRight = nR + numPlanar; best.planarLeft = false;
						}
					}
				}
			}
			numLeft[axis] += numStart + numPlanar;
		}

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
		tch(best.axis, best.pos - nodeBox.mn[best.axis], nodeBox.mx[best.axis] - best.pos, pl, pr);
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
	float runExact(Context &c, std::vector<uint8_t> &cls, uint32_t depth, uint32_t node, const Box &nodeBox,
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
	Split minimize(const Box &tight, const float *binSize, const float *invBinSize, const std::vector<uint32_t> &minBins,
	               const std::vector<uint32_t> &maxBins, uint32_t primCount) const {
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

	// buildTreeMinMax (gkdtree.h:1735-1867).  Nodes that drop to the exact method become jobs.
	float binned(Context &c, uint32_t depth, uint32_t node, const Box &nodeBox, const Box &tight,
	             std::vector<uint32_t> &prims, uint32_t badRefines) {
		const uint32_t primCount = (uint32_t) prims.size();
		const float leafCost = primCount * m_p.queryCost;
		if (primCount <= m_p.stopPrims || depth >= m_p.maxDepth) {
			leafFromIndices(c, node, prims.data(), primCount);
			return leafCost;
		}
		if (primCount <= m_p.exactPrimThreshold)
			return defer(c, depth, node, nodeBox, prims, badRefines);

		const int nb = m_p.minMaxBins;
		float binSize[3], invBinSize[3];
		const float recip = 1.0f / (float) nb;
		for (int a = 0; a < 3; ++a) { binSize[a] = (tight.mx[a] - tight.mn[a]) * recip; invBinSize[a] = 1 / binSize[a]; }
		std::vector<uint32_t> minBins(3 * (size_t) nb, 0u), maxBins(3 * (size_t) nb, 0u);
		const int64_t maxBin = nb - 1;
		for (uint32_t i = 0; i < primCount; ++i) {
			Box b; m_g.box(prims[i], b);
			for (int a = 0; a < 3; ++a) {
				const int64_t minIdx = (int64_t) ((b.mn[a] - tight.mn[a]) * invBinSize[a]);
				const int64_t maxIdx = (int64_t) ((b.mx[a] - tight.mn[a]) * invBinSize[a]);
				maxBins[a * nb + std::max((int64_t) 0, std::min(maxIdx, maxBin))]++;
				minBins[a * nb + std::max((int64_t) 0, std::min(minIdx, maxBin))]++;
			}
		}
		Split best = minimize(tight, binSize, invBinSize, minBins, maxBins, primCount);
		if (best.cost == kInf)
			return defer(c, depth, node, nodeBox, prims, badRefines);
		if (best.cost >= leafCost) {
			if ((best.cost > 4 * leafCost && primCount < 16) || badRefines >= m_p.maxBadRefines) {
				leafFromIndices(c, node, prims.data(), primCount);
				return leafCost;
			}
			++badRefines;
		}

		// MinMaxBins::partition (gkdtree.h:2517-2596)
		const float splitPos = best.pos;
		const int axis = best.axis;
		Box leftBounds, rightBounds; leftBounds.reset(); rightBounds.reset();
		std::vector<uint32_t> leftPrims, rightPrims;
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
		leftBounds.clip(tight); rightBounds.clip(tight);
		leftBounds.mx[axis] = std::min(leftBounds.mx[axis], splitPos);
		rightBounds.mn[axis] = std::max(rightBounds.mn[axis], splitPos);
		if (leftBounds.mx[axis] != rightBounds.mn[axis]) {
			const SAH tch(tight);
			const float nL = (float) leftPrims.size(), nR = (float) rightPrims.size();
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

		const uint32_t children = c.allocNodes(2);
		const uint32_t nodePosBefore = (uint32_t) c.nodes.size(), indexPosBefore = (uint32_t) c.indices.size();
		const uint32_t leafBefore = c.leafCount, nonemptyBefore = c.nonemptyLeafCount, innerBefore = c.innerCount;
		const size_t jobsBefore = m_jobs.size();
		{ PNode &n = c.nodes[node]; n.kind = 0; n.a = (uint32_t) best.axis; n.b = children; n.split = best.pos; }
		c.innerCount++;

		Box childBox = nodeBox;
		childBox.mx[best.axis] = best.pos;
		const float leftCost = binned(c, depth + 1, children, childBox, leftBounds, leftPrims, badRefines);
		childBox.mn[best.axis] = best.pos;
		childBox.mx[best.axis] = nodeBox.mx[best.axis];
		const float rightCost = binned(c, depth + 1, children + 1, childBox, rightBounds, rightPrims, badRefines);

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
			std::vector<uint8_t> cls(m_nPrims, 0);
			return runExact(c, cls, depth, node, nodeBox, prims, badRefines);
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
		auto worker = [&]() {
			std::vector<uint8_t> cls(m_nPrims, 0);
			for (;;) {
				const size_t k = next.fetch_add(1);
				if (k >= m_jobs.size())
					break;
				Job &j = *m_jobs[k];
				j.root = j.ctx.allocNodes(1);
				runExact(j.ctx, cls, j.depth, j.root, j.nodeBox, j.prims, j.badRefines);
				std::vector<uint32_t>().swap(j.prims);
			}
		};
		nThreads = std::max(1, std::min<int>(nThreads, (int) m_jobs.size()));
		std::vector<std::thread> pool;
		for (int t = 1; t < nThreads; ++t) pool.emplace_back(worker);
		worker();
		for (auto &t : pool) t.join();
	}

	const Geometry &m_g;
	const Params &m_p;
	uint32_t m_nPrims;
	bool m_parallel = false;
	std::vector<std::unique_ptr<Job>> m_jobs;
};

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
	b.m_parallel = nTris > p.exactPrimThreshold;
	Box scene; scene.reset();
	std::vector<uint32_t> prims(nTris);
	for (uint32_t i = 0; i < nTris; ++i) { Box t; g.box(i, t); scene.expand(t); prims[i] = i; }

	Context root;
	const uint32_t prelimRoot = root.allocNodes(1);
	b.binned(root, 1, prelimRoot, scene, scene, prims, 0);
	b.runJobs(nThreads);

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
