/*
 * orc_kdtree.c -- oracle (TEST INFRASTRUCTURE ONLY): SAH kd-tree construction.
 *
 * Restates GenericKDTree::buildInternal and helpers
 * (include/mitsuba/render/gkdtree.h:913-1214 buildInternal + final layout,
 *  :1735-1867 buildTreeMinMax, :1668-1704 transitionToNLogN, :1490-1530
 *  createEventList, :1898-2345 buildTree, :2350-2608 MinMaxBins) with the
 * SurfaceAreaHeuristic of include/mitsuba/render/sahkdtree3.h:35-79.
 *
 * Single-threaded (the reference's parallel build only changes which thread
 * builds a subtree).  One documented deviation: edge events that compare equal
 * under EdgeEventOrdering (gkdtree.h:1281-1289) are additionally ordered by
 * primitive index, because std::sort leaves their order unspecified; this only
 * affects the order of primitives inside a leaf.
 */
#include "orc_internal.h"

typedef struct { float min[3], max[3]; } aabb_t;

enum { EV_END = 0, EV_PLANAR = 1, EV_START = 2 };
typedef struct { float pos; uint32_t index; uint16_t type; uint16_t axis; } event_t;

enum { C_BOTH = 0, C_LEFT = 1, C_RIGHT = 2, C_BOTH_DONE = 3 };

typedef struct { uint32_t leaf, a, b; float split; } pnode_t;

typedef struct {
	float cost, pos; int axis; uint32_t numLeft, numRight; int planarLeft;
} split_t;

typedef struct {
	const float *vtx; const uint32_t *tri; uint32_t primCount;
	const float *genAABB;   /* [primCount][6]: boxes of the non-triangle primitives (tri row = {NONE,NONE,NONE}) */
	/* parameters (gkdtree.h:711-724) */
	float traversalCost, queryCost, emptySpaceBonus;
	uint32_t stopPrims, maxBadRefines, exactPrimThreshold, maxDepth; int minMaxBins, clip, retract;
	/* build context */
	pnode_t *nodes; size_t nNodes, capNodes;
	uint32_t *indices; size_t nIndices, capIndices;
	uint8_t *cls;
	uint32_t leafNodeCount, nonemptyLeafNodeCount, innerNodeCount, primIndexCount, retractedSplits, pruned;
	uint32_t *minBins, *maxBins;
} ctx_t;

static void aabb_reset(aabb_t *b) {
	for (int i = 0; i < 3; ++i) { b->min[i] = INFINITY; b->max[i] = -INFINITY; }
}
static void aabb_expand_pt(aabb_t *b, const float p[3]) {
	for (int i = 0; i < 3; ++i) { b->min[i] = fminf_(b->min[i], p[i]); b->max[i] = fmaxf_(b->max[i], p[i]); }
}
static void aabb_expand(aabb_t *b, const aabb_t *o) {
	for (int i = 0; i < 3; ++i) { b->min[i] = fminf_(b->min[i], o->min[i]); b->max[i] = fmaxf_(b->max[i], o->max[i]); }
}
static void aabb_clip(aabb_t *b, const aabb_t *o) {
	for (int i = 0; i < 3; ++i) { b->min[i] = fmaxf_(b->min[i], o->min[i]); b->max[i] = fminf_(b->max[i], o->max[i]); }
}
/* aabb.h:326-329 */
static float aabb_surface_area(const aabb_t *b) {
	float dx = b->max[0]-b->min[0], dy = b->max[1]-b->min[1], dz = b->max[2]-b->min[2];
	return (float) 2.0 * (dx*dy + dx*dz + dy*dz);
}

/* Triangle::getAABB (include/mitsuba/core/triangle.h:39-44) */
static void prim_aabb(const ctx_t *c, uint32_t idx, aabb_t *out) {
	const uint32_t *t = c->tri + 3 * (size_t) idx;
	if (t[0] == MTSGPU_KNOTRIANGLE) {       /* shape->getAABB() (skdtree.h:203-213) */
		for (int i = 0; i < 3; ++i) { out->min[i] = c->genAABB[6 * (size_t) idx + i]; out->max[i] = c->genAABB[6 * (size_t) idx + 3 + i]; }
		return;
	}
	aabb_reset(out);
	aabb_expand_pt(out, c->vtx + 3 * (size_t) t[0]);
	aabb_expand_pt(out, c->vtx + 3 * (size_t) t[1]);
	aabb_expand_pt(out, c->vtx + 3 * (size_t) t[2]);
}

static int prim_clipped_aabb(const ctx_t *c, uint32_t idx, const aabb_t *box, aabb_t *out) {
	const uint32_t *t = c->tri + 3 * (size_t) idx;
	if (t[0] == MTSGPU_KNOTRIANGLE) {       /* Shape::getClippedAABB (shape.cpp:59-63): getAABB().clip(box) */
		prim_aabb(c, idx, out);
		aabb_clip(out, box);
		for (int i = 0; i < 3; ++i) if (out->max[i] < out->min[i]) return 0;     /* AABB::isValid */
		return 1;
	}
	return orc_clipped_aabb(c->vtx + 3 * (size_t) t[0], c->vtx + 3 * (size_t) t[1], c->vtx + 3 * (size_t) t[2],
	                        box->min, box->max, out->min, out->max);
}

/* SurfaceAreaHeuristic (sahkdtree3.h:35-79) */
typedef struct { float temp0[3], temp1[3]; } sah_t;
static void sah_init(sah_t *s, const aabb_t *b) {
	float e[3] = { b->max[0]-b->min[0], b->max[1]-b->min[1], b->max[2]-b->min[2] };
	const float temp = 1.0f / (e[0] * e[1] + e[1]*e[2] + e[0]*e[2]);
	s->temp0[0] = (e[1] * e[2]) * temp; s->temp0[1] = (e[0] * e[2]) * temp; s->temp0[2] = (e[0] * e[1]) * temp;
	s->temp1[0] = (e[1] + e[2]) * temp; s->temp1[1] = (e[0] + e[2]) * temp; s->temp1[2] = (e[0] + e[1]) * temp;
}
static inline void sah_prob(const sah_t *s, int axis, float leftWidth, float rightWidth, float *pl, float *pr) {
	*pl = s->temp0[axis] + s->temp1[axis] * leftWidth;
	*pr = s->temp0[axis] + s->temp1[axis] * rightWidth;
}

static uint32_t alloc_nodes(ctx_t *c, uint32_t n) {
	if (c->nNodes + n > c->capNodes) {
		c->capNodes = (c->capNodes ? c->capNodes * 2 : 1024) + n;
		c->nodes = (pnode_t *) realloc(c->nodes, c->capNodes * sizeof(pnode_t));
	}
	uint32_t r = (uint32_t) c->nNodes;
	c->nNodes += n;
	return r;
}
static void push_index(ctx_t *c, uint32_t v) {
	if (c->nIndices + 1 > c->capIndices) {
		c->capIndices = c->capIndices ? c->capIndices * 2 : 4096;
		c->indices = (uint32_t *) realloc(c->indices, c->capIndices * sizeof(uint32_t));
	}
	c->indices[c->nIndices++] = v;
}

/* EdgeEventOrdering (gkdtree.h:1281-1289) + index tie-break (see header comment) */
static int event_cmp(const void *pa, const void *pb) {
	const event_t *a = (const event_t *) pa, *b = (const event_t *) pb;
	if (a->axis != b->axis) return a->axis < b->axis ? -1 : 1;
	if (a->pos != b->pos) return a->pos < b->pos ? -1 : 1;
	if (a->type != b->type) return a->type < b->type ? -1 : 1;
	if (a->index != b->index) return a->index < b->index ? -1 : 1;
	return 0;
}

static event_t *merge_events(const event_t *a, size_t na, const event_t *b, size_t nb, event_t *out) {
	size_t i = 0, j = 0;
	while (i < na && j < nb) {
		if (event_cmp(&b[j], &a[i]) < 0) *out++ = b[j++];
		else *out++ = a[i++];
	}
	while (i < na) *out++ = a[i++];
	while (j < nb) *out++ = b[j++];
	return out;
}

/* createLeaf from an index list (gkdtree.h:1578-1588) */
static void create_leaf_indices(ctx_t *c, uint32_t node, const uint32_t *idx, uint32_t primCount) {
	c->nodes[node].leaf = 1; c->nodes[node].a = (uint32_t) c->nIndices; c->nodes[node].b = (uint32_t) c->nIndices + primCount;
	if (primCount > 0) {
		c->nonemptyLeafNodeCount++;
		for (uint32_t i = 0; i < primCount; ++i) push_index(c, idx[i]);
		c->primIndexCount += primCount;
	}
	c->leafNodeCount++;
}

/* createLeaf from an event list (gkdtree.h:1545-1564): axis-0 start/planar events */
static void create_leaf_events(ctx_t *c, uint32_t node, const event_t *es, const event_t *ee, uint32_t primCount) {
	c->nodes[node].leaf = 1; c->nodes[node].a = (uint32_t) c->nIndices; c->nodes[node].b = (uint32_t) c->nIndices + primCount;
	if (primCount > 0) {
		c->nonemptyLeafNodeCount++;
		for (const event_t *e = es; e != ee && e->axis == 0; ++e)
			if (e->type == EV_START || e->type == EV_PLANAR)
				push_index(c, e->index);
		c->primIndexCount += primCount;
	}
	c->leafNodeCount++;
}

static int u32_cmp(const void *a, const void *b) {
	uint32_t x = *(const uint32_t *) a, y = *(const uint32_t *) b;
	return x < y ? -1 : (x > y ? 1 : 0);
}

/* createLeafAfterRetraction (gkdtree.h:1603-1637) */
static void create_leaf_after_retraction(ctx_t *c, uint32_t node, uint32_t start) {
	uint32_t indexCount = (uint32_t) (c->nIndices - start);
	qsort(c->indices + start, indexCount, sizeof(uint32_t), u32_cmp);
	uint32_t idx = start, p = start, end = start + indexCount;
	while (p < end) {
		c->indices[idx] = c->indices[p++];
		while (p < end && c->indices[p] == c->indices[idx])
			++p;
		idx++;
	}
	uint32_t nSeen = idx - start;
	c->primIndexCount = c->primIndexCount - indexCount + nSeen;
	c->nIndices = idx;
	c->nodes[node].leaf = 1; c->nodes[node].a = start; c->nodes[node].b = start + nSeen;
	c->nonemptyLeafNodeCount++;
	c->leafNodeCount++;
}

/* emit the 3 x (planar | start+end) events of one clipped box (gkdtree.h:1508-1522) */
static event_t *emit_events(event_t *out, const aabb_t *b, uint32_t index) {
	for (int axis = 0; axis < 3; ++axis) {
		float mn = b->min[axis], mx = b->max[axis];
		if (mn == mx) {
			out->pos = mn; out->index = index; out->type = EV_PLANAR; out->axis = (uint16_t) axis; out++;
		} else {
			out->pos = mn; out->index = index; out->type = EV_START; out->axis = (uint16_t) axis; out++;
			out->pos = mx; out->index = index; out->type = EV_END; out->axis = (uint16_t) axis; out++;
		}
	}
	return out;
}

static float build_tree(ctx_t *c, uint32_t depth, uint32_t node, const aabb_t *nodeAABB,
                        event_t *eventStart, event_t *eventEnd, uint32_t primCount, uint32_t badRefines);

/* gkdtree.h:1898-2345 */
static float build_tree(ctx_t *c, uint32_t depth, uint32_t node, const aabb_t *nodeAABB,
                        event_t *eventStart, event_t *eventEnd, uint32_t primCount, uint32_t badRefines) {
	float leafCost = primCount * c->queryCost;
	if (primCount <= c->stopPrims || depth >= c->maxDepth) {
		create_leaf_events(c, node, eventStart, eventEnd, primCount);
		return leafCost;
	}

	split_t best; best.cost = INFINITY; best.pos = 0; best.axis = 0; best.numLeft = best.numRight = 0; best.planarLeft = 0;
	uint32_t numLeft[3] = { 0, 0, 0 }, numRight[3] = { primCount, primCount, primCount };
	event_t *eventsByAxis[3] = { eventStart, eventEnd, eventEnd };
	int eventsByAxisCtr = 1;
	sah_t tch; sah_init(&tch, nodeAABB);

	for (event_t *event = eventStart; event < eventEnd; ) {
		int axis = event->axis;
		float pos = event->pos;
		uint32_t numStart = 0, numEnd = 0, numPlanar = 0;
		while (event < eventEnd && event->pos == pos && event->axis == axis && event->type == EV_END) { ++numEnd; ++event; }
		while (event < eventEnd && event->pos == pos && event->axis == axis && event->type == EV_PLANAR) { ++numPlanar; ++event; }
		while (event < eventEnd && event->pos == pos && event->axis == axis && event->type == EV_START) { ++numStart; ++event; }
		if (event < eventEnd && event->axis != axis)
			eventsByAxis[eventsByAxisCtr++] = event;

		numRight[axis] -= numPlanar + numEnd;

		if (pos > nodeAABB->min[axis] && pos < nodeAABB->max[axis]) {
			const uint32_t nL = numLeft[axis], nR = numRight[axis];
			const float nLF = (float) nL, nRF = (float) nR;
			float pl, pr;
			sah_prob(&tch, axis, pos - nodeAABB->min[axis], nodeAABB->max[axis] - pos, &pl, &pr);
			if (numPlanar == 0) {
				float cost = c->traversalCost + c->queryCost * (pl * nLF + pr * nRF);
				if (nL == 0 || nR == 0)
					cost *= c->emptySpaceBonus;
				if (cost < best.cost) {
					best.pos = pos; best.axis = axis; best.cost = cost; best.numLeft = nL; best.numRight = nR;
				}
			} else {
				float costPlanarLeft  = c->traversalCost + c->queryCost * (pl * (float) (nL+numPlanar) + pr * nRF);
				float costPlanarRight = c->traversalCost + c->queryCost * (pl * nLF + pr * (float) (nR+numPlanar));
				if (nL + numPlanar == 0 || nR == 0)
					costPlanarLeft *= c->emptySpaceBonus;
				if (nL == 0 || nR + numPlanar == 0)
					costPlanarRight *= c->emptySpaceBonus;
				if (costPlanarLeft < best.cost || costPlanarRight < best.cost) {
					best.pos = pos; best.axis = axis;
					if (costPlanarLeft < costPlanarRight) {
						best.cost = costPlanarLeft; best.numLeft = nL + numPlanar; best.numRight = nR; best.planarLeft = 1;
					} else {
						best.cost = costPlanarRight; best.numLeft = nL; best.numRight = nR + numPlanar; best.planarLeft = 0;
					}
				}
			}
		}
		numLeft[axis] += numStart + numPlanar;
	}

	/* "Bad refines" heuristic from PBRT (gkdtree.h:2038-2047) */
	if (best.cost >= leafCost) {
		if ((best.cost > 4 * leafCost && primCount < 16) || badRefines >= c->maxBadRefines || best.cost == INFINITY) {
			create_leaf_events(c, node, eventStart, eventEnd, primCount);
			return leafCost;
		}
		++badRefines;
	}

	/* classification (gkdtree.h:2053-2103) */
	uint8_t *storage = c->cls;
	for (event_t *e = eventsByAxis[best.axis]; e < eventEnd && e->axis == best.axis; ++e)
		storage[e->index] = C_BOTH;
	uint32_t primsLeft = 0, primsRight = 0, primsBoth = primCount;
	for (event_t *e = eventsByAxis[best.axis]; e < eventEnd && e->axis == best.axis; ++e) {
		if (e->type == EV_END && e->pos <= best.pos) {
			storage[e->index] = C_LEFT; primsBoth--; primsLeft++;
		} else if (e->type == EV_START && e->pos >= best.pos) {
			storage[e->index] = C_RIGHT; primsBoth--; primsRight++;
		} else if (e->type == EV_PLANAR) {
			if (e->pos < best.pos || (e->pos == best.pos && best.planarLeft)) {
				storage[e->index] = C_LEFT; primsBoth--; primsLeft++;
			} else if (e->pos > best.pos || (e->pos == best.pos && !best.planarLeft)) {
				storage[e->index] = C_RIGHT; primsBoth--; primsRight++;
			}
		}
	}

	aabb_t leftNodeAABB = *nodeAABB, rightNodeAABB = *nodeAABB;
	leftNodeAABB.max[best.axis] = best.pos;
	rightNodeAABB.min[best.axis] = best.pos;
	uint32_t prunedLeft = 0, prunedRight = 0;

	event_t *leftEventsStart = (event_t *) malloc(sizeof(event_t) * (6 * (size_t) best.numLeft + 1));
	event_t *rightEventsStart = (event_t *) malloc(sizeof(event_t) * (6 * (size_t) best.numRight + 1));
	event_t *leftEventsEnd = leftEventsStart, *rightEventsEnd = rightEventsStart;

	if (c->clip) {
		event_t *lt = (event_t *) malloc(sizeof(event_t) * (6 * (size_t) primsLeft + 1)), *lte = lt;
		event_t *rt = (event_t *) malloc(sizeof(event_t) * (6 * (size_t) primsRight + 1)), *rte = rt;
		event_t *nl = (event_t *) malloc(sizeof(event_t) * (6 * (size_t) primsBoth + 1)), *nle = nl;
		event_t *nr = (event_t *) malloc(sizeof(event_t) * (6 * (size_t) primsBoth + 1)), *nre = nr;
		for (event_t *e = eventStart; e < eventEnd; ++e) {
			int cl = storage[e->index];
			if (cl == C_LEFT) {
				*lte++ = *e;
			} else if (cl == C_RIGHT) {
				*rte++ = *e;
			} else if (cl == C_BOTH) {
				const uint32_t index = e->index;
				aabb_t cl_, cr_;
				int vl = prim_clipped_aabb(c, index, &leftNodeAABB, &cl_);
				int vr = prim_clipped_aabb(c, index, &rightNodeAABB, &cr_);
				if (vl && aabb_surface_area(&cl_) > 0) nle = emit_events(nle, &cl_, index);
				else prunedLeft++;
				if (vr && aabb_surface_area(&cr_) > 0) nre = emit_events(nre, &cr_, index);
				else prunedRight++;
				storage[index] = C_BOTH_DONE;
			}
		}
		c->pruned += prunedLeft + prunedRight;
		qsort(nl, (size_t) (nle - nl), sizeof(event_t), event_cmp);
		qsort(nr, (size_t) (nre - nr), sizeof(event_t), event_cmp);
		leftEventsEnd = merge_events(lt, (size_t) (lte - lt), nl, (size_t) (nle - nl), leftEventsStart);
		rightEventsEnd = merge_events(rt, (size_t) (rte - rt), nr, (size_t) (nre - nr), rightEventsStart);
		free(lt); free(rt); free(nl); free(nr);
	} else {
		for (event_t *e = eventStart; e < eventEnd; ++e) {
			int cl = storage[e->index];
			if (cl == C_LEFT) *leftEventsEnd++ = *e;
			else if (cl == C_RIGHT) *rightEventsEnd++ = *e;
			else if (cl == C_BOTH) { *leftEventsEnd++ = *e; *rightEventsEnd++ = *e; }
		}
	}

	/* recursion (gkdtree.h:2279-2312) */
	uint32_t children = alloc_nodes(c, 2);
	uint32_t nodePosBeforeSplit = (uint32_t) c->nNodes;
	uint32_t indexPosBeforeSplit = (uint32_t) c->nIndices;
	uint32_t leafNodeCountBeforeSplit = c->leafNodeCount;
	uint32_t nonemptyLeafNodeCountBeforeSplit = c->nonemptyLeafNodeCount;
	uint32_t innerNodeCountBeforeSplit = c->innerNodeCount;
	c->nodes[node].leaf = 0; c->nodes[node].a = (uint32_t) best.axis; c->nodes[node].b = children; c->nodes[node].split = best.pos;
	c->innerNodeCount++;

	float leftCost = build_tree(c, depth+1, children, &leftNodeAABB, leftEventsStart, leftEventsEnd,
	                            best.numLeft - prunedLeft, badRefines);
	float rightCost = build_tree(c, depth+1, children+1, &rightNodeAABB, rightEventsStart, rightEventsEnd,
	                             best.numRight - prunedRight, badRefines);
	free(leftEventsStart); free(rightEventsStart);

	float pl, pr;
	sah_prob(&tch, best.axis, best.pos - nodeAABB->min[best.axis], nodeAABB->max[best.axis] - best.pos, &pl, &pr);
	float finalCost = c->traversalCost + (pl * leftCost + pr * rightCost);

	if (!c->retract || finalCost < primCount * c->queryCost) {
		return finalCost;
	} else {
		c->nNodes = nodePosBeforeSplit;
		c->retractedSplits++;
		c->leafNodeCount = leafNodeCountBeforeSplit;
		c->nonemptyLeafNodeCount = nonemptyLeafNodeCountBeforeSplit;
		c->innerNodeCount = innerNodeCountBeforeSplit;
		create_leaf_after_retraction(c, node, indexPosBeforeSplit);
		return leafCost;
	}
}

/* transitionToNLogN (gkdtree.h:1668-1704) + createEventList (:1490-1530) */
static float transition_to_nlogn(ctx_t *c, uint32_t depth, uint32_t node, const aabb_t *nodeAABB,
                                 const uint32_t *prims, uint32_t primCount, uint32_t badRefines) {
	event_t *es = (event_t *) malloc(sizeof(event_t) * (6 * (size_t) primCount + 1)), *ee = es;
	uint32_t actualPrimCount = 0;
	for (uint32_t i = 0; i < primCount; ++i) {
		uint32_t index = prims[i];
		aabb_t box;
		if (c->clip) {
			int valid = prim_clipped_aabb(c, index, nodeAABB, &box);
			if (!valid || aabb_surface_area(&box) == 0)
				continue;
		} else {
			prim_aabb(c, index, &box);
		}
		ee = emit_events(ee, &box, index);
		++actualPrimCount;
	}
	qsort(es, (size_t) (ee - es), sizeof(event_t), event_cmp);
	float cost = build_tree(c, depth, node, nodeAABB, es, ee, actualPrimCount, badRefines);
	free(es);
	/* m_parallelBuild stays on above exactPrimThreshold primitives (gkdtree.h:939-940): the
	 * subtree is built by a worker and reports -inf, "never tear down this subtree" (:1691-1692) */
	if (c->primCount > c->exactPrimThreshold)
		return -INFINITY;
	return cost;
}

/* MinMaxBins::minimizeCost (gkdtree.h:2405-2510) */
static split_t minmax_minimize(ctx_t *c, const aabb_t *m_aabb, const float binSize[3], const float invBinSize[3],
                               uint32_t m_primCount) {
	split_t cand; cand.cost = INFINITY; cand.pos = 0; cand.axis = 0; cand.numLeft = cand.numRight = 0; cand.planarLeft = 0;
	int binIdx = 0, leftBin = 0;
	const int m_binCount = c->minMaxBins;
	sah_t tch; sah_init(&tch, m_aabb);
	for (int axis = 0; axis < 3; ++axis) {
		float extents[3] = { m_aabb->max[0]-m_aabb->min[0], m_aabb->max[1]-m_aabb->min[1], m_aabb->max[2]-m_aabb->min[2] };
		uint32_t numLeft = 0, numRight = m_primCount;
		float leftWidth = 0, rightWidth = extents[axis];
		const float bs = binSize[axis];
		for (int i = 0; i < m_binCount-1; ++i) {
			numLeft += c->minBins[binIdx];
			numRight -= c->maxBins[binIdx];
			leftWidth += bs;
			rightWidth -= bs;
			float pl, pr;
			sah_prob(&tch, axis, leftWidth, rightWidth, &pl, &pr);
			float cost = c->traversalCost + c->queryCost * (pl * (float) numLeft + pr * (float) numRight);
			if (cost < cand.cost) {
				cand.cost = cost; cand.axis = axis; cand.numLeft = numLeft; cand.numRight = numRight; leftBin = i;
			}
			binIdx++;
		}
		binIdx++;
	}
	const int axis = cand.axis;
	const float min = m_aabb->min[axis];
	float invBS = invBinSize[axis];
	float split = min + (leftBin + 1) * binSize[axis];
	float splitNext = nextafterf(split, 3.402823466e+38F);
	int idx = (int) ((split - min) * invBS);
	int idxNext = (int) ((splitNext - min) * invBS);
	if (!(idx == leftBin && idxNext == leftBin+1)) {
		float left = m_aabb->min[axis];
		float right = m_aabb->max[axis];
		int it = 0;
		while (1) {
			split = left + (right-left)/2;
			splitNext = nextafterf(split, 3.402823466e+38F);
			idx = (int) ((split - min) * invBS);
			idxNext = (int) ((splitNext - min) * invBS);
			if (idx == leftBin && idxNext == leftBin+1) {
				break;
			} else if (abs(idx-idxNext) > 1 || ++it > 50) {
				cand.cost = INFINITY;
				break;
			}
			if (idx <= leftBin) left = split;
			else right = split;
		}
	}
	if (split <= m_aabb->min[axis] || split >= m_aabb->max[axis])
		cand.cost = INFINITY;
	cand.pos = split;
	return cand;
}

/* buildTreeMinMax (gkdtree.h:1735-1867) */
static float build_tree_minmax(ctx_t *c, uint32_t depth, uint32_t node, const aabb_t *nodeAABB,
                               const aabb_t *tightAABB, uint32_t *indices, uint32_t primCount, uint32_t badRefines) {
	float leafCost = primCount * c->queryCost;
	if (primCount <= c->stopPrims || depth >= c->maxDepth) {
		create_leaf_indices(c, node, indices, primCount);
		return leafCost;
	}
	if (primCount <= c->exactPrimThreshold)
		return transition_to_nlogn(c, depth, node, nodeAABB, indices, primCount, badRefines);

	/* MinMaxBins::setAABB + bin (gkdtree.h:2363-2403) */
	const int nb = c->minMaxBins;
	float binSize[3], invBinSize[3];
	{
		float recip = 1.0f / (float) nb;   /* Vector / Float == * (1/f) */
		for (int a = 0; a < 3; ++a) {
			binSize[a] = (tightAABB->max[a] - tightAABB->min[a]) * recip;
			invBinSize[a] = 1 / binSize[a];
		}
	}
	memset(c->minBins, 0, sizeof(uint32_t) * 3 * (size_t) nb);
	memset(c->maxBins, 0, sizeof(uint32_t) * 3 * (size_t) nb);
	const int64_t maxBin = nb - 1;
	for (uint32_t i = 0; i < primCount; ++i) {
		aabb_t b; prim_aabb(c, indices[i], &b);
		for (int a = 0; a < 3; ++a) {
			int64_t minIdx = (int64_t) ((b.min[a] - tightAABB->min[a]) * invBinSize[a]);
			int64_t maxIdx = (int64_t) ((b.max[a] - tightAABB->min[a]) * invBinSize[a]);
			int64_t mx = maxIdx < maxBin ? maxIdx : maxBin; if (mx < 0) mx = 0;
			int64_t mn = minIdx < maxBin ? minIdx : maxBin; if (mn < 0) mn = 0;
			c->maxBins[a * nb + mx]++;
			c->minBins[a * nb + mn]++;
		}
	}
	split_t best = minmax_minimize(c, tightAABB, binSize, invBinSize, primCount);
	if (best.cost == INFINITY)
		return transition_to_nlogn(c, depth, node, nodeAABB, indices, primCount, badRefines);

	if (best.cost >= leafCost) {
		if ((best.cost > 4 * leafCost && primCount < 16) || badRefines >= c->maxBadRefines) {
			create_leaf_indices(c, node, indices, primCount);
			return leafCost;
		}
		++badRefines;
	}

	/* MinMaxBins::partition (gkdtree.h:2517-2596) */
	const float splitPos = best.pos;
	const int axis = best.axis;
	uint32_t nL = 0, nR = 0;
	aabb_t leftBounds, rightBounds; aabb_reset(&leftBounds); aabb_reset(&rightBounds);
	uint32_t *leftIndices = (uint32_t *) malloc(sizeof(uint32_t) * ((size_t) best.numLeft + 1));
	uint32_t *rightIndices = (uint32_t *) malloc(sizeof(uint32_t) * ((size_t) best.numRight + 1));
	for (uint32_t i = 0; i < primCount; ++i) {
		const uint32_t primIndex = indices[i];
		aabb_t b; prim_aabb(c, primIndex, &b);
		if (b.max[axis] <= splitPos) {
			aabb_expand(&leftBounds, &b); leftIndices[nL++] = primIndex;
		} else if (b.min[axis] > splitPos) {
			aabb_expand(&rightBounds, &b); rightIndices[nR++] = primIndex;
		} else {
			aabb_expand(&leftBounds, &b); aabb_expand(&rightBounds, &b);
			leftIndices[nL++] = primIndex; rightIndices[nR++] = primIndex;
		}
	}
	if (nL != best.numLeft || nR != best.numRight) {
		fprintf(stderr, "orc_kd_build: min-max partition inconsistent (%u/%u vs %u/%u)\n", nL, nR, best.numLeft, best.numRight);
		abort();
	}
	aabb_clip(&leftBounds, tightAABB);
	aabb_clip(&rightBounds, tightAABB);
	leftBounds.max[axis] = fminf_(leftBounds.max[axis], splitPos);
	rightBounds.min[axis] = fmaxf_(rightBounds.min[axis], splitPos);
	if (leftBounds.max[axis] != rightBounds.min[axis]) {
		sah_t tch; sah_init(&tch, tightAABB);
		float p1l, p1r, p2l, p2r;
		sah_prob(&tch, axis, leftBounds.max[axis] - tightAABB->min[axis], tightAABB->max[axis] - leftBounds.max[axis], &p1l, &p1r);
		sah_prob(&tch, axis, rightBounds.min[axis] - tightAABB->min[axis], tightAABB->max[axis] - rightBounds.min[axis], &p2l, &p2r);
		float cost1 = c->traversalCost + c->queryCost * (p1l * (float) nL + p1r * (float) nR);
		float cost2 = c->traversalCost + c->queryCost * (p2l * (float) nL + p2r * (float) nR);
		if (cost1 <= cost2) { best.cost = cost1; best.pos = leftBounds.max[axis]; }
		else { best.cost = cost2; best.pos = rightBounds.min[axis]; }
		leftBounds.max[axis] = fminf_(leftBounds.max[axis], best.pos);
		rightBounds.min[axis] = fmaxf_(rightBounds.min[axis], best.pos);
	}

	uint32_t children = alloc_nodes(c, 2);
	uint32_t nodePosBeforeSplit = (uint32_t) c->nNodes;
	uint32_t indexPosBeforeSplit = (uint32_t) c->nIndices;
	uint32_t leafNodeCountBeforeSplit = c->leafNodeCount;
	uint32_t nonemptyLeafNodeCountBeforeSplit = c->nonemptyLeafNodeCount;
	uint32_t innerNodeCountBeforeSplit = c->innerNodeCount;
	c->nodes[node].leaf = 0; c->nodes[node].a = (uint32_t) best.axis; c->nodes[node].b = children; c->nodes[node].split = best.pos;
	c->innerNodeCount++;

	aabb_t childAABB = *nodeAABB;
	childAABB.max[best.axis] = best.pos;
	float leftCost = build_tree_minmax(c, depth+1, children, &childAABB, &leftBounds, leftIndices, best.numLeft, badRefines);
	childAABB.min[best.axis] = best.pos;
	childAABB.max[best.axis] = nodeAABB->max[best.axis];
	float rightCost = build_tree_minmax(c, depth+1, children + 1, &childAABB, &rightBounds, rightIndices, best.numRight, badRefines);
	free(leftIndices); free(rightIndices);

	sah_t tch; sah_init(&tch, nodeAABB);
	float pl, pr;
	sah_prob(&tch, best.axis, best.pos - nodeAABB->min[best.axis], nodeAABB->max[best.axis] - best.pos, &pl, &pr);
	float finalCost = c->traversalCost + (pl * leftCost + pr * rightCost);

	if (!c->retract || finalCost < primCount * c->queryCost) {
		return finalCost;
	} else {
		c->nNodes = nodePosBeforeSplit;
		c->retractedSplits++;
		c->leafNodeCount = leafNodeCountBeforeSplit;
		c->nonemptyLeafNodeCount = nonemptyLeafNodeCountBeforeSplit;
		c->innerNodeCount = innerNodeCountBeforeSplit;
		create_leaf_after_retraction(c, node, indexPosBeforeSplit);
		return leafCost;
	}
}

/* log2i (include/mitsuba/core/util.h): floor(log2(v)) */
static int log2i_(uint32_t v) { int r = 0; while (v >>= 1) r++; return r; }

int orc_kd_build(const float *vtx_pos, const uint32_t *tri_idx, uint32_t n_tris, const float *gen_aabb,
                 const mtsgpu_kd_params *kp, orc_kdtree *out) {
	ctx_t c; memset(&c, 0, sizeof(c));
	memset(out, 0, sizeof(*out));
	c.vtx = vtx_pos; c.tri = tri_idx; c.primCount = n_tris; c.genAABB = gen_aabb;
	/* defaults: gkdtree.h:711-724 */
	c.traversalCost = (kp && kp->traversal_cost > 0) ? kp->traversal_cost : 15;
	c.queryCost = (kp && kp->query_cost > 0) ? kp->query_cost : 20;
	c.emptySpaceBonus = (kp && kp->empty_space_bonus > 0) ? kp->empty_space_bonus : 0.9f;
	c.stopPrims = (kp && kp->stop_prims > 0) ? (uint32_t) kp->stop_prims : 6;
	c.maxBadRefines = (kp && kp->max_bad_refines > 0) ? (uint32_t) kp->max_bad_refines : 3;
	c.exactPrimThreshold = (kp && kp->exact_prim_threshold > 0) ? (uint32_t) kp->exact_prim_threshold : 65536;
	c.minMaxBins = (kp && kp->min_max_bins > 1) ? kp->min_max_bins : 128;
	c.clip = (kp && kp->clip < 0) ? 0 : 1;
	c.retract = (kp && kp->retract < 0) ? 0 : 1;
	c.maxDepth = (kp && kp->max_depth > 0) ? (uint32_t) kp->max_depth : 0;

	if (n_tris == 0) {
		out->n_nodes = 1; out->nodes = (uint32_t *) calloc(2, sizeof(uint32_t));
		out->nodes[0] = 0x80000000u; out->nodes[1] = 0;
		out->indices = (uint32_t *) calloc(1, sizeof(uint32_t));
		return 0;
	}
	/* gkdtree.h:945-947 */
	if (c.maxDepth == 0)
		c.maxDepth = (uint32_t) (int) (8 + 1.3f * log2i_(n_tris));
	if (c.maxDepth > 48) c.maxDepth = 48;

	c.cls = (uint8_t *) calloc(n_tris, 1);
	c.minBins = (uint32_t *) malloc(sizeof(uint32_t) * 3 * (size_t) c.minMaxBins);
	c.maxBins = (uint32_t *) malloc(sizeof(uint32_t) * 3 * (size_t) c.minMaxBins);
	uint32_t *indices = (uint32_t *) malloc(sizeof(uint32_t) * n_tris);
	aabb_t aabb; aabb_reset(&aabb);
	for (uint32_t i = 0; i < n_tris; ++i) {
		aabb_t b; prim_aabb(&c, i, &b);
		aabb_expand(&aabb, &b);
		indices[i] = i;
	}
	uint32_t prelimRoot = alloc_nodes(&c, 1);
	build_tree_minmax(&c, 1, prelimRoot, &aabb, &aabb, indices, n_tris, 0);
	free(indices);

	/* final layout (gkdtree.h:1042-1138): DFS, left child first, siblings adjacent */
	uint32_t nodeCount = c.innerNodeCount + c.leafNodeCount;
	out->n_nodes = nodeCount; out->n_indices = c.primIndexCount;
	out->nodes = (uint32_t *) malloc(sizeof(uint32_t) * 2 * (size_t) nodeCount);
	out->indices = (uint32_t *) malloc(sizeof(uint32_t) * ((size_t) c.primIndexCount + 1));
	typedef struct { uint32_t node, target; aabb_t box; } item_t;
	item_t *stack = (item_t *) malloc(sizeof(item_t) * 256);
	int sp = 0;
	uint32_t nodePtr = 0, indexPtr = 0;
	float expTraversalSteps = 0, expLeavesVisited = 0, expPrimitivesIntersected = 0, heuristicCost = 0;
	stack[sp].node = prelimRoot; stack[sp].target = nodePtr++; stack[sp].box = aabb; sp++;
	while (sp > 0) {
		item_t it = stack[--sp];
		const pnode_t *n = &c.nodes[it.node];
		uint32_t *target = out->nodes + 2 * (size_t) it.target;
		if (n->leaf) {
			uint32_t primsInLeaf = n->b - n->a;
			target[0] = 0x80000000u | indexPtr;
			target[1] = indexPtr + primsInLeaf;
			float quantity = aabb_surface_area(&it.box), weightedQuantity = quantity * primsInLeaf;
			expLeavesVisited += quantity;
			expPrimitivesIntersected += weightedQuantity;
			heuristicCost += weightedQuantity * c.queryCost;
			for (uint32_t k = n->a; k < n->b; ++k)
				out->indices[indexPtr++] = c.indices[k];
		} else {
			float quantity = aabb_surface_area(&it.box);
			expTraversalSteps += quantity;
			heuristicCost += quantity * c.traversalCost;
			uint32_t children = nodePtr;
			nodePtr += 2;
			int axis = (int) n->a;
			target[0] = (uint32_t) axis | ((children - it.target) << 2);
			memcpy(&target[1], &n->split, 4);
			aabb_t box = it.box;
			float tmp = box.min[axis];
			box.min[axis] = n->split;
			stack[sp].node = n->b + 1; stack[sp].target = children + 1; stack[sp].box = box; sp++;
			box.min[axis] = tmp;
			box.max[axis] = n->split;
			stack[sp].node = n->b; stack[sp].target = children; stack[sp].box = box; sp++;
		}
	}
	free(stack);
	if (nodePtr != nodeCount || indexPtr != c.primIndexCount) {
		fprintf(stderr, "orc_kd_build: layout mismatch (%u/%u nodes, %u/%u indices)\n", nodePtr, nodeCount, indexPtr, c.primIndexCount);
		abort();
	}
	float rootQuantity = aabb_surface_area(&aabb);
	out->stats[0] = c.innerNodeCount; out->stats[1] = c.leafNodeCount; out->stats[2] = c.primIndexCount;
	out->stats[3] = expTraversalSteps / rootQuantity;
	out->stats[4] = expLeavesVisited / rootQuantity;
	out->stats[5] = expPrimitivesIntersected / rootQuantity;
	(void) heuristicCost;

	/* enlarge (gkdtree.h:1170-1176); note max uses the already-moved min */
	for (int a = 0; a < 3; ++a) { out->tight_min[a] = aabb.min[a]; out->tight_max[a] = aabb.max[a]; }
	float nmin[3], nmax[3];
	for (int a = 0; a < 3; ++a) nmin[a] = aabb.min[a] - ((aabb.max[a] - aabb.min[a]) * ORC_EPS + ORC_EPS);
	for (int a = 0; a < 3; ++a) nmax[a] = aabb.max[a] + ((aabb.max[a] - nmin[a]) * ORC_EPS + ORC_EPS);
	for (int a = 0; a < 3; ++a) { out->aabb_min[a] = nmin[a]; out->aabb_max[a] = nmax[a]; }

	free(c.nodes); free(c.indices); free(c.cls); free(c.minBins); free(c.maxBins);
	return 0;
}

void orc_kd_free(orc_kdtree *t) {
	free(t->nodes); free(t->indices);
	memset(t, 0, sizeof(*t));
}
