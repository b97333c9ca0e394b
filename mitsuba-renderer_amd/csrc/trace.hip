// trace.hip -- K2/K4: kd-tree traversal (closest hit with the material sort, any hit), and k_prep.
#include "kdevice.h"

namespace mg {

// ===========================================================================
// K2/K4: kd-tree traversal.
// ShapeKDTree::rayIntersect (src/librender/skdtree.cpp:108-132, :180-199) +
// rayIntersectHavran<shadow> (include/mitsuba/render/sahkdtree3.h:170-300) +
// TriAccel::rayIntersect (include/mitsuba/render/triaccel.h:98-159).
//
// Havran's stack of exit points is a LIFO once the entry point is kept in
// registers.  An exit point is (far child, t, axis, split)
// and all four are functions of (parent node, ray), so the stack stores ONE dword
// per level: parent index * 2 + "far child is the right one".  The top exit
// point lives in registers; deeper levels sit in LDS ([level][lane], conflict
// free), levels beyond kStackLDS spill to HBM.  The 8-entry hashed mailbox
// (sahkdtree3.h:130-144) is kept (LDS) because it decides equal-t ties.
// ===========================================================================

// MG_TAIL_FILTER: the exact record-tail filter (api.cpp: tailFilterFlag).  The stack words then carry the split axis of their node
// in bits 30-31 (node index * 2 + side stays below bit 30), so that the leaf loop knows through which face the ray leaves the
// leaf -- axis, plane, side -- without fetching anything.  Exact (bit-identical results: the GPU suite and the film A/B ran on it),
// it removes a quarter of the tail requests of a C3 frame -- and is NEUTRAL in time (profiles/r06g_exp_trace_exact_tail_filter.txt:
// the skipped tails are L1 hits on their head's line, the test costs six vector instructions per candidate), so the product is
// built without it.
#ifndef MG_TAIL_FILTER
#define MG_TAIL_FILTER 0
#endif
// How many lanes below this one have their bit set in a wave mask (ranks of the material sort, of the refill).  The hardware counts
// them (v_mbcnt_lo / _hi); the portable form popcount(mask & ((1 << lane) - 1)) keeps a 64-bit per-lane mask and its complement alive
// across the whole kernel -- four VGPRs in a kernel that sits on its register ceiling, which the compiler paid for with 20 bytes of
// scratch reloaded at every retirement (round 6: 79 VGPRs / 20 B -> 79 / 0 B closest-hit, 63 -> 61 VGPRs any-hit).
#ifndef MG_RANK_MBCNT
#define MG_RANK_MBCNT 1
#endif
__device__ __forceinline__ uint32_t lanes_below(uint64_t mask, uint32_t lane) {
	if (MG_RANK_MBCNT) return __builtin_amdgcn_mbcnt_hi((uint32_t) (mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t) mask, 0u));
	return (uint32_t) __popcll(mask & ((1ull << lane) - 1ull));
}
#ifndef MG_EXP_EXTRA_MISS
#define MG_EXP_EXTRA_MISS 0
#endif
size_t trace_spill_levels() { return kSpillLevels; }      // in dwords per thread
int trace_tail_filter() { return MG_TAIL_FILTER; }          // 0: the kernels were built without the record-tail filter (the default), 1: both kernels, 2: any-hit only
size_t trace_stack_levels() { return kStackLDS + kSpillLevels; }
uint32_t trace_top_nodes() { return 2u * kTopPairsMax; }

// Persistent waves: the grid is sized to fill the chip once and every wave walks its own 64-ray
// batches of the queue with a private cursor (no work-queue atomic: a single head word saturates at
// ~88 fetches/us, MI355X_MICROARCH.md "dequeue").  The kernel is bound by instruction issue under
// SIMD divergence (measured lane utilisation in DESIGN.md section 6), which three scheduling rules
// attack: idle lanes are refilled from the wave's next batch once q.refill_min of them are idle, and
// the descent / primitive loops are left as soon as fewer than q.desc_min / q.leaf_min lanes still
// need them (the others stop waiting; stragglers resume in the next round).  None of this changes
// what is computed for a ray.
// `first` is the queue index of the wave's first batch, `stride` the distance to its next one, `static_n` the
// statically dealt prefix of the queue.
// leaf record e (48 bytes at a stride of kLeafStride x 16): its head (k | flags | primitive, n_u, n_v, n_d) and the two halves of its tail
__device__ __forceinline__ const uint4 *leaf_head(const DTraceScene &sc, uint32_t e) { return &sc.leaf_ta[kLeafStride * (size_t) e]; }
__device__ __forceinline__ const uint4 *leaf_tail(const DTraceScene &sc, uint32_t e, uint32_t half) { return &sc.leaf_ta[kLeafStride * (size_t) e + 1 + half]; }

template <int MODE, bool COUNT, bool BIN, bool TIE = false>
__device__ __forceinline__ void trace_body(const DTraceScene &sc, const DPaths &ps, const DQueues &q, const TracePlan &plan,
                                           const uint32_t *queue, uint32_t n, const uint32_t first, const uint32_t stride,
                                           uint32_t (*s_stack)[kTraceBlock], uint32_t (*s_mbox)[kTraceBlock], const uint4 *s_top) {
	// node fetches: from the LDS copy of the top of the tree when the index lies inside it
	// COUNT: the requests this lane issued (global / served by the LDS copy) and, when q.rec is set, the list of them
	uint32_t g_pair = 0, l_pair = 0, g_node = 0, l_node = 0, g_tail = 0, g_spill = 0, g_head = 0;
	uint32_t rec_n = 0, rec_slot = 0;
	auto rec_add = [&](uint32_t kind, uint32_t idx) {
		if (COUNT && q.rec) {
			if (rec_n < q.rec_cap) q.rec[(size_t) rec_slot * q.rec_cap + rec_n] = (kind << 29) | idx;
			rec_n++;
		}
	};
	constexpr uint32_t kTop = TIE ? kTopPairsTie : kTopPairs;      // sibling pairs of the LDS copy this instantiation was given
	auto load_node = [&](uint32_t i) -> uint2 {
		if (kTop && i < 2u * kTop) { if (COUNT) l_node++; return reinterpret_cast<const uint2 *>(s_top)[i]; }
		if (COUNT) { g_node++; rec_add(kReqNode, i); }
		return sc.nodes[i];
	};
	auto load_pair = [&](uint32_t left) -> uint4 {
		if (kTop && left < 2u * kTop) { if (COUNT) l_pair++; return s_top[left >> 1]; }
		if (COUNT) { g_pair++; rec_add(kReqPair, left >> 1); }
		return reinterpret_cast<const uint4 *>(sc.nodes)[left >> 1];
	};
	// The hashed mailbox decides which of two primitives with equal t is reported (sahkdtree3.h:130-144, :278-283), so
	// closest-hit rays keep it.  For any-hit rays it only saves repeated tests of a primitive that spans several leaves --
	// the answer is a disjunction over the same primitives either way -- and its 8 dwords per lane are better spent on
	// the LDS copy of the tree: the shadow kernels run without it (counting builds keep it: the oracle counts with it).
	// TIE: closest-hit rays WITHOUT the mailbox.  Skipping a primitive that was tested before only saves work -- its test gives the
	// same answer again (rejected: the interval has only shrunk; accepted before and still the best: t == maxt is accepted again and
	// changes nothing) -- EXCEPT when two different primitives tie in t: then the later test wins and which tests run depends on the
	// mailbox's eight hashed slots.  Such a ray (an accepted hit with t == best_t on another primitive) is flagged, listed in q.redo
	// instead of being binned, and traced again by the kernel with the mailbox; everything else is bit-identical by the argument above.
	static_assert(!TIE || (MODE == 0 && BIN && !COUNT), "the mailbox-free form exists for binned closest-hit launches");
	constexpr bool kMbox = (MODE == 0 && !TIE) || COUNT;
	const uint32_t tid = threadIdx.x;
	const uint32_t gtid = blockIdx.x * kTraceBlock + tid;     // spill slot of this lane
	const uint32_t lane = lane_id();
	const uint32_t shard = blockIdx.x % kBinShards;     // contention on a bin counter is spread over kBinShards words

	uint32_t c_inner = 0, c_leaf = 0, c_idx = 0, c_tri = 0;
	// COUNT: candidates whose record tail was fetched although their plane distance lies outside the interval the ray spends
	// in the leaf being visited (the reference tests against the ray's whole interval, sahkdtree3.h:262-288); c_ten = stack[enPt].t
	uint32_t c_tail_out = 0; float c_ten = 0;
	uint32_t w_inner = 0, w_leaf = 0, w_outer = 0, w_batch = 0;   // COUNT: lane slots issued per loop (64 per wave iteration)
#define MG_WSLOT(w) do { if (COUNT && lane == (uint32_t) __builtin_ctzll(__builtin_amdgcn_ballot_w64(true))) (w) += 64u; } while (0)

	// Ray supply of a wave.  The first plan.static_n queue entries are dealt out statically: the wave owns the batches
	// (64 consecutive entries) wave_id, wave_id + n_waves, ... and walks them without any atomic.  The rest of the
	// queue (about a quarter) is claimed batch by batch through ONE counter once the static share is used up, which
	// evens out the finishing times of the waves (measured: the mean wave used to live 0.90-0.94 of the kernel).
	// A lane whose ray has finished stays idle until at least q.refill_min lanes of the wave are idle; then the
	// finished rays are retired together (one hit store + one binning step) and the idle lanes take new rays.
	// A launch with fewer rays than the chip has lanes runs B < 64 rays per wave (lanes >= B stay idle): a wave lasts
	// as long as its slowest ray, and with the wave slots to spare narrower batches shorten that critical path.
	const uint32_t B = plan.batch, static_n = plan.static_n;
	const uint64_t limitMask = (B >= 64u) ? ~0ull : ((1ull << B) - 1ull);
	uint32_t next_static = first;           // queue index of this wave's next static batch (uniform)
	uint32_t sup_base = 0, sup_left = 0;    // the chunk being handed out: queue[sup_base .. sup_base + sup_left)
	bool dyn_done = static_n >= n;          // nothing (left) to claim dynamically
	const uint32_t refill_min = plan.refill_min, desc_min = plan.desc_min, leaf_min = plan.leaf_min;

	uint32_t id = 0;
	float ox = 0, oy = 0, oz = 0, dx = 1, dy = 1, dz = 1, rx = 1, ry = 1, rz = 1;
	float mint = 0, maxt = 0, tmax0 = 0;
	float enx = 0, eny = 0, enz = 0, exx = 0, exy = 0, exz = 0, ex_t = 0;   // stack[enPt].p, stack[exPt].p, stack[exPt].t
	int sp = 0;
	// the current exit point: ex_node = its far child, ex_ref = its stack word (parent index * 2 + "far child is the right one")
	uint32_t ex_node = kNullNode, ex_ref = kSentinel, cur = 0;
	float best_t = MG_INF, best_u = 0, best_v = 0;
	uint32_t best_prim = kNoPrim, best_shape = 0;
	// best_shape: shape index of the accepted hit (dword 10 of its record), read while the record is at hand
	uint32_t e_cont = kNoPrim;              // position inside an interrupted leaf
	uint2 nd = make_uint2(0u, 0u);          // sc.nodes[cur], fetched as soon as cur is known
	bool found = false;
	bool has = false;                       // this lane is traversing a ray
	bool done = false;                      // this lane holds a finished ray that has not been retired yet

	while (true) {
		const uint64_t liveMask = __builtin_amdgcn_ballot_w64(has);
		const uint32_t nlive = (uint32_t) __popcll(liveMask);
		const bool wantRays = nlive == 0u || B - nlive >= refill_min;
		if (wantRays && sup_left == 0u) {
			// next chunk: a batch of the static share, or one claimed from the shared tail of the queue
			if (next_static < static_n) {
				sup_base = next_static; sup_left = (static_n - next_static < B) ? static_n - next_static : B; next_static += stride;
			} else if (!dyn_done) {
				uint32_t b = 0;
				if (lane == 0) b = atomicAdd(&q.counters[(MODE == 0 ? kCntDynClosest : kCntDynShadow) * kCounterStride], 1u);
				b = (uint32_t) __builtin_amdgcn_readfirstlane((int) b);
				const unsigned long long base = (unsigned long long) static_n + (unsigned long long) B * b;
				if (base < n) { sup_base = (uint32_t) base; sup_left = (n - sup_base < B) ? n - sup_base : B; }
				else dyn_done = true;
			}
		}
		const uint32_t remaining = sup_left;
		if (nlive == 0u || (remaining != 0u && wantRays)) {
			// ---- retire the finished rays (all lanes take part in the ballots) ----
			if (MODE == 0) {
				int bin = -1;
				constexpr bool hitsToBins = BIN;      // a binned path's hit travels with its id (DQueues::bin_hits)
				if (done) {
					if (!hitsToBins) st_stream<1>(&ps.hit(id), make_uint4(__float_as_uint(best_t), __float_as_uint(best_u), __float_as_uint(best_v), best_prim));
					if (BIN) {
						bin = kNumBins - 1;
						if (found) {
							bin = (int) sc.shape_bin[TIE ? (best_shape & 0x7FFFFFFFu) : best_shape];      // BSDF type of the hit shape, or the terminal bin (one lookup)
						}
						if (TIE && (best_shape & 0x80000000u)) {
							// two primitives tied on this ray: the mailbox decides (rare: coincident geometry, hits on shared edges)
							q.redo[atomicAdd(&q.counters[kCntRedo * kCounterStride], 1u)] = id;
							bin = -1;
						}
					}
				}
				if (BIN) {
					// material sort: one ballot + prefix popcount per bin; lane b reserves the slots of bin b,
					// so the wave issues ONE returning atomic instruction for all bins
					uint32_t cnt = 0, rank = 0;
					#pragma unroll
					for (int b = 0; b < kNumBins; ++b) {
						const uint64_t m = __builtin_amdgcn_ballot_w64(bin == b);
						if (lane == (uint32_t) b) cnt = (uint32_t) __popcll(m);
						if (bin == b) rank = lanes_below(m, lane);
					}
					uint32_t base = 0;
					if (lane < (uint32_t) kNumBins && cnt != 0u)
						base = atomicAdd(&q.counters[(lane * kBinShards + shard) * kCounterStride], cnt);
					base = __shfl(base, bin < 0 ? 0 : bin);
					// a shard's segment holds its share of a statically dealt queue (api.cpp: ensurePaths); dynamic claims
					// can in principle exceed it: the entry is dropped then, the counter still counts it, and the host
					// repeats the launch with static dealing when it sees a count above the capacity
					const uint32_t pos = base + rank;
					if (bin >= 0 && pos < q.bin_seg_cap) {
						const size_t at = (size_t) bin * q.bin_stride + (size_t) shard * q.bin_seg_cap + pos;
						st_stream<1>(&q.bins_base[at], id);
						if (hitsToBins) st_stream<1>(&q.bin_hits[at], make_uint4(__float_as_uint(best_t), __float_as_uint(best_u), __float_as_uint(best_v), best_prim));
					}
				}
			} else if (MODE == 1) {
				// Scene::sampleLuminaire's visibility test passed: add the pending contribution (path.cpp:124)
				if (done && !found) {
					const float4 c = ps.shq_nee[id];            // id = position in the shadow queue; c.w = path id
					const uint32_t pid = __float_as_uint(c.w);
					if (q.nee_parked) {
						// parked in the record (DQueues::nee_parked): its next reader adds it
						ps.slot(pid, 2) = make_float4(c.x, c.y, c.z, __uint_as_float(kNeeTag));
					} else {
						float4 L = ps.Li(pid);
						L.x += c.x; L.y += c.y; L.z += c.z;
						ps.Li(pid) = L;
					}
				}
			} else {
				if (done)
					ps.hit(id) = make_uint4(0u, 0u, 0u, found ? 1u : 0u);
			}
			done = false;
			if (remaining == 0u)
				break;          // nlive == 0 and nothing left: the wave is finished

			// ---- refill: idle lane number r takes ray r of the current chunk ----
			const uint32_t r = lanes_below(~liveMask & limitMask, lane);
			const bool take = !has && lane < B && r < remaining;
			const uint32_t taken = (B - nlive < remaining) ? B - nlive : remaining;
			const uint32_t my = sup_base + r;
			sup_base += taken; sup_left -= taken;
			MG_WSLOT(w_batch);
			if (take) {
				id = (MODE == 1) ? my : ld_stream<1>(&queue[my]);       // shadow rays are addressed by their queue position
				// (rays that arrive in queue order are streamed, like shadow rays: not part of the recorded request list)
				if (COUNT && q.rec) { rec_slot = my; rec_n = 0; if (MODE != 1 && !(MODE == 0 && ps.rq_o)) { rec_add(kReqRay, id * kPathSlots); rec_add(kReqRay, id * kPathSlots + 1); } }
				float4 a, b;
				float rmint, rmaxt;
				if (MODE == 1) {
					a = ld_stream<1>(&ps.shq_o[my]); b = ld_stream<1>(&ps.shq_d[my]);
					rmint = kShadowEpsilon; rmaxt = 1 - kShadowEpsilon;      // Scene::isOccluded, scene.h:241-246
				} else {
					if (MODE == 0 && ps.rq_o) { a = ld_stream<1>(&ps.rq_o[my]); b = ld_stream<1>(&ps.rq_d[my]); }    // in queue order: no trip behind the id
					else { a = ld_stream<1>(&ps.ray_o(id)); b = ld_stream<1>(&ps.ray_d(id)); }
					rmint = a.w; rmaxt = b.w;
				}
				ox = a.x; oy = a.y; oz = a.z; dx = b.x; dy = b.y; dz = b.z;
				rx = 1.0f / dx; ry = 1.0f / dy; rz = 1.0f / dz;          // Ray::dRcp (ray.h:63-74)
				// AABB::rayIntersect (aabb.h:349-382) + adaptive epsilon (skdtree.cpp:114-122)
				bool go = true;
				mint = -MG_INF; maxt = MG_INF;
				#pragma unroll
				for (int i = 0; i < 3; ++i) {
					const float direction = sel3(dx, dy, dz, i), origin = sel3(ox, oy, oz, i);
					const float minVal = sc.aabb_min[i], maxVal = sc.aabb_max[i];
					if (direction == 0) {
						if (origin < minVal || origin > maxVal) go = false;
					} else {
						const float rc = sel3(rx, ry, rz, i);
						float t1 = (minVal - origin) * rc, t2 = (maxVal - origin) * rc;
						if (t1 > t2) { const float tmp = t1; t1 = t2; t2 = tmp; }
						mint = smax(mint, t1);
						maxt = smin(maxt, t2);
						if (mint > maxt) go = false;
					}
				}
				float rayMinT = rmint;
				if (rayMinT == kEpsilon) {
					float m = smax(smax(fabsf(ox), fabsf(oy)), fabsf(oz));
					if (MODE == 0) m = smax(m, kEpsilon);    // only the (ray, its) variant has the inner max
					rayMinT *= m;
				}
				if (rayMinT > mint) mint = rayMinT;
				if (rmaxt < maxt) maxt = rmaxt;
				if (!(maxt > mint)) go = false;
				best_t = MG_INF; best_u = 0; best_v = 0; best_prim = kNoPrim; best_shape = 0;
				found = false;
				done = !go;       // a ray that misses the scene's box is finished at once
				has = go;
				if (COUNT && q.rec && !go) { if (MODE != 1 && !(MODE == 0 && BIN)) rec_add(kReqHit, id * kPathSlots + 2); q.rec_len[rec_slot] = rec_n; }
				if (go) {
					if (kMbox) {
						#pragma unroll
						for (int i = 0; i < 8; ++i) s_mbox[i][tid] = 0xFFFFFFFFu;
					}
					// entry point (stack[enPt]) and current exit point (stack[exPt]) in registers
					enx = ox + mint * dx; eny = oy + mint * dy; enz = oz + mint * dz;      // stack[enPt].p = ray(mint)
					tmax0 = maxt;
					if (COUNT) c_ten = mint;
					ex_t = maxt; exx = ox + maxt * dx; exy = oy + maxt * dy; exz = oz + maxt * dz;
					ex_node = kNullNode; ex_ref = kSentinel;
					sp = 0; cur = 0; e_cont = kNoPrim;
					nd = load_node(0u);
				}
			}
		}

		// ---- one leaf visit of every live lane: descend, test the leaf, pop ----
		if (has) {
			{
				bool inner = !(nd.x & 0x80000000u);     // nd = sc.nodes[cur] is part of the lane's state
				// The descent stops as soon as fewer than q.desc_min lanes are still on inner nodes: the lanes
				// that wait in a leaf go on, the few stragglers resume their descent in the next round.
				do { if (inner) {
					// One step of rayIntersectHavran's inner loop (sahkdtree3.h:196-252), written without
					// branches: this loop is bound by instruction issue (exec-mask bookkeeping of a branchy
					// version costs more than the arithmetic), not by memory.  The entry / exit points are
					// kept as the 3-vectors the reference stores (ray(t) with the split axis overwritten).
					const float split = __uint_as_float(nd.y);
					const int axis = (int) (nd.x & 3u);
					const uint32_t left = nd.x >> 2;            // device nodes hold the absolute index of the left child
					// both children in one 16-byte load (sibling pairs are 16-byte aligned in the device order), issued
					// before the case logic below instead of after it: the step is a chain of dependent fetches
					const uint4 pair = load_pair(left);
					if (COUNT) c_inner++;
					MG_WSLOT(w_inner);
					const float pen = sel3(enx, eny, enz, axis), pex = sel3(exx, exy, exz, axis);
					const bool A = pen <= split, B = pex <= split, C = pen == split, D = split < pex;
					//   A &&  B        : left only            (N1-N3, P5, Z2, Z3)
					//   A && !B &&  C  : right only           (Z1)
					//   A && !B && !C  : near left, far right (N4)  -> push
					//  !A &&  D        : right only           (P1-P3, N5)
					//  !A && !D        : near right, far left (P4)  -> push
					// the case logic is done on wave masks (SALU) to keep it off the vector pipe
					const uint64_t mA = __builtin_amdgcn_ballot_w64(A), mB = __builtin_amdgcn_ballot_w64(B);
					const uint64_t mC = __builtin_amdgcn_ballot_w64(C), mD = __builtin_amdgcn_ballot_w64(D);
					const bool side1 = __builtin_amdgcn_inverse_ballot_w64(~mA | (~mB & mC));   // go to the right child now
					const bool push = __builtin_amdgcn_inverse_ballot_w64((mA & ~mB & ~mC) | (~mA & ~mD));
					const uint32_t side = side1 ? 1u : 0u;
					if (push) {
						// push the current exit point's reference; (cur, far child) becomes the exit point
						if (sp < kStackLDS) s_stack[sp][tid] = ex_ref;
						else { q.spill[(size_t) (sp - kStackLDS) * q.spill_stride + gtid] = ex_ref; if (COUNT) g_spill++; }
						++sp;
						const uint32_t farRight = A ? 1u : 0u;
						const float distToSplit = (split - sel3(ox, oy, oz, axis)) * sel3(rx, ry, rz, axis);
						ex_ref = (cur << 1) | farRight | (MG_TAIL_FILTER ? ((uint32_t) axis << 30) : 0u);
						ex_t = distToSplit;
						const float px = ox + distToSplit * dx, py = oy + distToSplit * dy, pz = oz + distToSplit * dz;
						exx = (axis == 0) ? split : px; exy = (axis == 1) ? split : py; exz = (axis == 2) ? split : pz;   // selects, not branches
						ex_node = left + farRight;
					}
					cur = left + side;
					nd = side1 ? make_uint2(pair.z, pair.w) : make_uint2(pair.x, pair.y);
				}
				// evaluated for all lanes after the step (a lane that did not step sits on a leaf): the flag then is one
				// compare on the merged register instead of a value carried through the branch
				inner = !(nd.x & 0x80000000u);
				} while ((uint32_t) __popcll(__builtin_amdgcn_ballot_w64(inner)) >= desc_min);

				if (!inner) {
				// --- leaf: test the primitives (sahkdtree3.h:262-288, skdtree.h:244-336) ---
				if (COUNT && e_cont == kNoPrim) c_leaf++;      // a resumed leaf was counted already
				MG_WSLOT(w_outer);
				bool hitShadow = false, more = false;
#if MG_EXP_EXTRA_MISS
				// sensitivity probe (profiles/r06n_*): ONE more 16-byte request per leaf visit whose line is not in the L2 -- a pseudo-random
				// line of the path records, inside a 128 MB window (1: served by the Infinity Cache) or anywhere in 8 GB of them (2: from HBM);
				// the value is not used
				if (e_cont == kNoPrim) {
					uint32_t hsh = (id * 0x9E3779B9u) ^ (nd.x * 0x85EBCA6Bu); hsh ^= hsh >> 15; hsh *= 0x2C1B3C6Du; hsh ^= hsh >> 12;
					const uint32_t lines = MG_EXP_EXTRA_MISS == 1 ? (1u << 20) : (1u << 26);
					// 3: the same request to a line that IS in the L2 (the first 256 KB of the node array): what the request itself costs
					const float4 probe = MG_EXP_EXTRA_MISS == 3 ? reinterpret_cast<const float4 *>(sc.nodes)[(hsh & 0x3FFFu) * 1u] : ps.base[(size_t) (hsh & (lines - 1u)) * 8u];
					if (__float_as_uint(probe.x) == 0xDEADBEEFu && __float_as_uint(probe.w) == 0x12345u) best_u = 0.0f;
				}
#endif
				{
#if MG_TAIL_FILTER
					// The face through which the ray leaves this leaf: the plane of the current exit point (sahkdtree3.h:233,248-249) on its
					// axis; the leaf lies below the plane when the exit point's far child is the right one.  Kept with the sign that turns
					// "beyond the face" into "greater than": fO + t fD is the plane point's coordinate on that axis -- the very expression
					// o_u + t d_u (or o_v + t d_v) of triaccel.h:151-152 when the axis is one of the triangle's projection axes, negated
					// exactly when the leaf lies above the plane -- and fS the plane.  The end of the ray itself is no face (fS = inf).
					const uint32_t fAxisHi = ex_ref & 0xC0000000u;            // the axis where the records keep k; 3 for the sentinel
					const int fAxis = (int) (ex_ref >> 30);
					const uint32_t fFlip = (ex_ref & 1u) ? 0u : 0x80000000u;
					const float fO = __uint_as_float(__float_as_uint(sel3(ox, oy, oz, fAxis)) ^ fFlip);
					const float fD = __uint_as_float(__float_as_uint(sel3(dx, dy, dz, fAxis)) ^ fFlip);
					const float fS = ex_ref == kSentinel ? MG_INF : __uint_as_float(__float_as_uint(sel3(exx, exy, exz, fAxis)) ^ fFlip);
#endif
					uint32_t e = (e_cont != kNoPrim) ? e_cont : (nd.x & 0x7FFFFFFFu);     // resume an interrupted leaf
					const uint32_t last = nd.y;
					// record = 3 x 16 B: A = (k<<30 | non-occluder<<29 | prim, n_u, n_v, n_d), B = (a_u, a_v, b_nu, b_nv),
					// C = (c_nu, c_nv, shape, -).  A alone decides the mailbox test and the plane distance t; B and C
					// are only fetched for primitives whose t lies inside [mint, maxt] (triaccel.h:141-149).
					uint4 A;
					more = e != last;
					if (more) { A = ld_stream<2>(leaf_head(sc, e)); if (COUNT) { g_head++; rec_add(kReqLeaf, kLeafStride * e); } }
					// like the descent, the primitive loop stops when fewer than q.leaf_min lanes have entries left;
					// those lanes keep their position (e_cont) and go on in the next round
					// the primitive loop runs at a raised wave priority: a wave in it holds the lanes of the others back the
					// shortest (1.55 entries per visit), and its record fetches go out ahead of the descent steps of the waves it
					// shares the SIMD with: 195.0 -> 192.5 ms of traversal per C3 frame (profiles/r04u_exp_trace_wave_priority.txt)
					__builtin_amdgcn_s_setprio(2);
					do { if (more) {
						uint4 An = A;
						if (e + 1 != last) { An = ld_stream<2>(leaf_head(sc, e + 1)); if (COUNT) { g_head++; rec_add(kReqLeaf, kLeafStride * (e + 1)); } }      // next record's head in flight
						const uint32_t prim = A.x & 0x0FFFFFFFu, k = A.x >> 30;
						if (COUNT) c_idx++;
						MG_WSLOT(w_leaf);
						// Flat form of the mailbox test + TriAccel::rayIntersect: the plane distance t is computed for
						// every entry (selects, no branches) and masked afterwards; only the barycentric part, which
						// needs the rest of the record, is conditional.
						uint32_t *mslot = &s_mbox[kMbox ? (prim & 7u) : 0u][tid];
						const bool fresh = !kMbox || *mslot != prim;              // not in the mailbox
						const bool occl = !(MODE != 0 && (A.x & 0x20000000u));   // shape->isOccluder() (skdtree.h:318-333)
						const bool ok = fresh && occl && (k != 3u);               // k == 3: degenerate triangle or another shape
						if (COUNT && fresh) c_tri++;
						if (sc.has_shapes && k == 3u && A.y != 0u && fresh && occl) {     // has_shapes is uniform: one scalar branch
							// a non-triangle shape (skdtree.h:287-296 / :328-332); A.y = shape type, B = centre + radius
							const uint4 B = *leaf_tail(sc, e, 0);
							if (COUNT) { g_tail++; rec_add(kReqLeaf, kLeafStride * e + 1u); }
							const V3 ctr(__uint_as_float(B.x), __uint_as_float(B.y), __uint_as_float(B.z));
							const float rad = __uint_as_float(B.w);
							if (MODE != 0) {
								if (sphere_occludes(ctr, rad, V3(ox, oy, oz), V3(dx, dy, dz), mint, maxt)) hitShadow = true;
							} else {
								float ts;
								if (sphere_intersect(ctr, rad, V3(ox, oy, oz), V3(dx, dy, dz), mint, maxt, ts)) {
									const uint32_t tied = TIE ? ((ts == best_t && prim != best_prim) ? 0x80000000u : (best_shape & 0x80000000u)) : 0u;
									maxt = ts;
									best_t = ts; best_u = 0.0f; best_v = 0.0f; best_prim = prim;
									best_shape = leaf_tail(sc, e, 1)->z | tied;
								}
							}
						}
						const bool k0 = k == 0u, k1 = k == 1u;
						const float n_u = __uint_as_float(A.y), n_v = __uint_as_float(A.z), n_d = __uint_as_float(A.w);
						const float o_u = k0 ? oy : (k1 ? oz : ox), o_v = k0 ? oz : (k1 ? ox : oy), o_k = k0 ? ox : (k1 ? oy : oz);
						const float d_u = k0 ? dy : (k1 ? dz : dx), d_v = k0 ? dz : (k1 ? dx : dy), d_k = k0 ? dx : (k1 ? dy : dz);
						const float recip = 1.0f / (d_u * n_u + d_v * n_v + d_k);
						const float t = (n_d - o_u * n_u - o_v * n_v - o_k) * recip;
#if MG_TAIL_FILTER
						// flagged entry (api.cpp: tailFilterFlag), exit axis one of the triangle's projection axes (not its k), plane point
						// beyond the face by more than the margin: TriAccel::rayIntersect is certain to reject, the tail is not fetched
						const bool beyond = (MG_TAIL_FILTER == 1 || MODE != 0) && (A.x & 0x10000000u) && ((A.x ^ fAxisHi) & 0xC0000000u) != 0u && (fO + t * fD) - fS > sc.tail_margin;
						if (ok && !(t < mint || t > maxt) && !beyond) {
#else
						if (ok && !(t < mint || t > maxt)) {
#endif
							const uint4 B = ld_stream<2>(leaf_tail(sc, e, 0));
							const uint4 C = ld_stream<2>(leaf_tail(sc, e, 1));         // c_nu, c_nv, shape index, -
							if (COUNT) { g_tail += 2u; rec_add(kReqLeaf, kLeafStride * e + 1u); rec_add(kReqLeaf, kLeafStride * e + 2u); }
						if (COUNT && (t < c_ten - 1e-4f * fabsf(c_ten) || t > ex_t + 1e-4f * fabsf(ex_t))) c_tail_out++;
							const float a_u = __uint_as_float(B.x), a_v = __uint_as_float(B.y);
							const float b_nu = __uint_as_float(B.z), b_nv = __uint_as_float(B.w);
							const float c_nu = __uint_as_float(C.x), c_nv = __uint_as_float(C.y);
							const float hu = o_u + t * d_u - a_u;
							const float hv = o_v + t * d_v - a_v;
							const float u = hv * b_nu + hu * b_nv;
							const float v = hu * c_nu + hv * c_nv;
							if (u >= 0 && v >= 0 && u + v <= 1.0f) {
								if (MODE != 0) hitShadow = true;
								const uint32_t tied = TIE ? ((t == best_t && prim != best_prim) ? 0x80000000u : (best_shape & 0x80000000u)) : 0u;
								maxt = t;      // a later hit with equal t replaces this one (t > maxt rejects)
								best_t = t; best_u = u; best_v = v; best_prim = prim; best_shape = C.z | tied;
							}
						}
						if (kMbox) *mslot = prim;         // (re)writing an entry that is already there changes nothing
						A = An;
						++e;
					}
					more = (e != last) && !hitShadow;      // for all lanes: those that did not step have e == last or hitShadow
					} while ((uint32_t) __popcll(__builtin_amdgcn_ballot_w64(more)) >= leaf_min);
					__builtin_amdgcn_s_setprio(0);
					e_cont = more ? e : kNoPrim;
				}
				bool finished = false;
				if (hitShadow) finished = true;
				else if (more) { /* leaf not finished yet */ }
				else if (ex_t > maxt) finished = true;
				else {
					// --- pop: the exit point becomes the entry point ---
					enx = exx; eny = exy; enz = exz;
					if (COUNT) c_ten = ex_t;
					cur = ex_node;
					if (cur == kNullNode) {
						finished = true;
					} else {
						--sp;
						nd = load_node(cur);         // in flight together with the parent's node below
						const uint32_t ref = (sp < kStackLDS) ? s_stack[sp][tid] : q.spill[(size_t) (sp - kStackLDS) * q.spill_stride + gtid];
						if (ref == kSentinel) {
							ex_t = tmax0; exx = ox + tmax0 * dx; exy = oy + tmax0 * dy; exz = oz + tmax0 * dz;
							ex_node = kNullNode; ex_ref = kSentinel;
						} else {
							// the exit point is a function of (parent node, ray): rebuilt with the reference's formulas (sahkdtree3.h:233,248-249)
							const uint2 pn = load_node(MG_TAIL_FILTER ? ((ref & 0x3FFFFFFFu) >> 1) : (ref >> 1));
							const int axis = (int) (pn.x & 3u);
							const float split = __uint_as_float(pn.y);
							ex_node = (pn.x >> 2) + (ref & 1u);
							ex_t = (split - sel3(ox, oy, oz, axis)) * sel3(rx, ry, rz, axis);
							const float px = ox + ex_t * dx, py = oy + ex_t * dy, pz = oz + ex_t * dz;
							exx = (axis == 0) ? split : px; exy = (axis == 1) ? split : py; exz = (axis == 2) ? split : pz;
							ex_ref = ref;
						}
					}
				}
				if (finished) {
					has = false; done = true; found = (MODE == 0) ? (best_prim != kNoPrim) : hitShadow;
					if (COUNT && q.rec) { if (MODE != 1 && !(MODE == 0 && BIN)) rec_add(kReqHit, id * kPathSlots + 2); q.rec_len[rec_slot] = rec_n; }      // binned hits are streamed
				}
				}
			}
		}
	}

	if (COUNT) {
		// wave reduction, then one atomic per wave and counter
		unsigned long long v[16] = { c_inner, c_leaf, c_idx, c_tri, w_inner, w_leaf, w_outer, w_batch, g_pair, l_pair, g_node, l_node, g_tail, g_spill, g_head, c_tail_out };
		#pragma unroll
		for (int k = 0; k < 16; ++k) {
			unsigned long long x = v[k];
			for (int off = 32; off > 0; off >>= 1)
				x += __shfl_down(x, off);
			if (lane == 0)
				atomicAdd(&q.trace_counts[k], x);
		}
	}
}

template <int MODE, bool COUNT, bool BIN, bool TIE = false>
__global__ __launch_bounds__(kTraceBlock, trace_waves_per_simd(MODE)) void k_trace(DTraceScene sc, DPaths ps, DQueues q,
                                                          const uint32_t *queue, uint32_t n_host, const uint32_t *n_dev) {
	constexpr uint32_t kTop = TIE ? kTopPairsTie : kTopPairs;
	__shared__ uint32_t s_stack[kStackLDS][kTraceBlock];
	__shared__ uint32_t s_mbox[((MODE == 0 && !TIE) || COUNT) ? 8 : 1][kTraceBlock];
	__shared__ uint4 s_top[kTop ? kTop : 1];
	// the number of rays: known to the host, or left in device memory by the kernel that filled the queue
	const uint32_t n = n_dev ? (uint32_t) __builtin_amdgcn_readfirstlane((int) *n_dev) : n_host;
	const TracePlan plan = trace_plan(n, MODE, q);
	if (blockIdx.x >= plan.blocks)
		return;                            // a grid sized for the worst case: nothing left for this workgroup
	if (q.dev_stats && blockIdx.x == 0 && threadIdx.x == 0) {
		atomicAdd(&q.dev_stats[MODE == 0 ? kStatClosest : kStatShadow], (unsigned long long) n);
		atomicAdd(&q.dev_stats[kStatLaunches], 1ull);
	}
	const uint32_t first = (blockIdx.x * (kTraceBlock / 64u) + (threadIdx.x >> 6)) * plan.batch;
	const uint32_t stride = plan.blocks * (kTraceBlock / 64u) * plan.batch;      // queue entries per round of the grid
	if (kTop) {
		// the device tree is padded to at least 2 * kTopPairsMax nodes (mtsgpu_upload_scene)
		for (uint32_t t = threadIdx.x; t < kTop; t += kTraceBlock) s_top[t] = reinterpret_cast<const uint4 *>(sc.nodes)[t];
		__syncthreads();
	}
	trace_body<MODE, COUNT, BIN, TIE>(sc, ps, q, plan, queue, n, first, stride, s_stack, s_mbox, s_top);
}

// ONE persistent launch per bounce (host-driven bounces, knob "merged"): its waves drain the closest-hit
// queue of bounce b + 1 and then the any-hit queue of bounce b -- legal because the any-hit kernel only parks direct-light
// terms, so neither queue depends on the other -- so that the waves that run out of closest-hit rays go on with shadow rays
// instead of idling through the 0.3-0.7 ms in which the launch's longest rays finish alone.  One footprint: the closest-hit
// kernel's (80 VGPRs, 52 KB of LDS, 3 workgroups per CU), which the any-hit phase then runs at too (6 waves per SIMD
// instead of 8, no mailbox).  Each phase deals its queue exactly as the separate launches do (own plan, own counter set).
__global__ __launch_bounds__(kTraceBlock, trace_waves_per_simd(0)) void k_trace_pair(DTraceScene sc, DPaths ps, DQueues qc, DQueues qs,
                                                                                    const uint32_t *queue_c, uint32_t n_c, const uint32_t *queue_s, uint32_t n_s) {
	__shared__ uint32_t s_stack[kStackLDS][kTraceBlock];
	__shared__ uint32_t s_mbox[8][kTraceBlock];
	__shared__ uint4 s_top[kTopPairs ? kTopPairs : 1];
	const TracePlan plan_c = trace_plan(n_c, 0, qc), plan_s = trace_plan(n_s, 1, qs);
	if (kTopPairs) {
		for (uint32_t t = threadIdx.x; t < kTopPairs; t += kTraceBlock) s_top[t] = reinterpret_cast<const uint4 *>(sc.nodes)[t];
		__syncthreads();
	}
	const uint32_t wave = blockIdx.x * (kTraceBlock / 64u) + (threadIdx.x >> 6);
	if (blockIdx.x < plan_c.blocks)
		trace_body<0, false, true>(sc, ps, qc, plan_c, queue_c, n_c, wave * plan_c.batch, plan_c.blocks * (kTraceBlock / 64u) * plan_c.batch, s_stack, s_mbox, s_top);
	if (blockIdx.x < plan_s.blocks)
		trace_body<1, false, false>(sc, ps, qs, plan_s, queue_s, n_s, wave * plan_s.batch, plan_s.blocks * (kTraceBlock / 64u) * plan_s.batch, s_stack, s_mbox, s_top);
}

// Device-driven bounces: the per-bin views k_shade needs, from the shard counters the closest-hit launch left in `cur`
// (what the host computes from a read-back otherwise), and the counter set of the NEXT bounce cleared.  One workgroup.
__global__ __launch_bounds__(256) void k_prep(const uint32_t *cur, uint32_t *next_set, BinView *views, uint32_t bin_seg_cap,
                                              unsigned long long *dev_stats) {
	__shared__ uint32_t s_cnt[kNumBins * kBinShards];
	const uint32_t t = threadIdx.x;
	if (t < (uint32_t) (kNumBins * kBinShards)) {
		const uint32_t c = cur[t * kCounterStride];
		s_cnt[t] = c;
		if (c > bin_seg_cap && dev_stats) atomicAdd(&dev_stats[kStatOverflow], 1ull);
	}
	if (t < (uint32_t) kNumCounters) next_set[t * kCounterStride] = 0u;
	__syncthreads();
	if (t < (uint32_t) kNumBins) {
		uint32_t acc = 0;
		for (int k = 0; k < kBinShards; ++k) {
			views[t].prefix[k] = acc;
			const uint32_t c = s_cnt[t * kBinShards + k];
			acc += c < bin_seg_cap ? c : bin_seg_cap;      // entries beyond the capacity were dropped (and flagged)
		}
		views[t].prefix[kBinShards] = acc;
	}
}

template <int MODE, bool COUNT, bool BIN, bool TIE = false>
static void launch_trace_t(hipStream_t s, const DScene &sc, const DPaths &ps, const DQueues &q, const uint32_t *queue, uint32_t n,
                           const uint32_t *n_dev) {
	// persistent grid: enough workgroups to fill every CU, never more than there are rays (trace_plan); when only the
	// device knows the count, the grid is sized for the upper bound n and the surplus workgroups exit at once
	unsigned blocks = trace_plan(n, MODE, q).blocks;
	if (n_dev) {
		// any count up to n: the narrowest batches need the most workgroups
		const unsigned minBatch = (q.tune_batch >= 1 && q.tune_batch <= 64) ? q.tune_batch : (q.coherent ? 64u : 8u);
		unsigned perCu = trace_blocks_per_cu(MODE);
		if (q.tune_blocks_per_cu && q.tune_blocks_per_cu < perCu) perCu = q.tune_blocks_per_cu;
		blocks = std::min<unsigned>(blocks_for(n, minBatch * (kTraceBlock / 64)), q.n_cus * perCu);
	}
	if (!blocks) return;
	hipLaunchKernelGGL((k_trace<MODE, COUNT, BIN, TIE>), dim3(blocks), dim3(kTraceBlock), 0, s, trace_scene(sc), ps, q, queue, n, n_dev);
}

void launch_trace(hipStream_t s, int mode, bool count, bool bin, const DScene &sc, const DPaths &ps,
                  const DQueues &q, const uint32_t *queue, uint32_t n, bool coherent, const uint32_t *n_dev, bool tie) {
	if (!n) return;
	DQueues qq = q;
	qq.coherent = coherent ? 1u : 0u;
	if (n_dev)
		qq.force_static = 1u;        // no dynamically claimed batches: the material-queue segments cannot overflow then
	if (mode == 0) {
		if (bin) { if (count) launch_trace_t<0, true, true>(s, sc, ps, qq, queue, n, n_dev); else if (tie) launch_trace_t<0, false, true, true>(s, sc, ps, qq, queue, n, n_dev); else launch_trace_t<0, false, true>(s, sc, ps, qq, queue, n, n_dev); }
		else     { if (count) launch_trace_t<0, true, false>(s, sc, ps, qq, queue, n, n_dev); else launch_trace_t<0, false, false>(s, sc, ps, qq, queue, n, n_dev); }
	} else if (mode == 1) {
		if (count) launch_trace_t<1, true, false>(s, sc, ps, qq, queue, n, n_dev); else launch_trace_t<1, false, false>(s, sc, ps, qq, queue, n, n_dev);
	} else {
		if (count) launch_trace_t<2, true, false>(s, sc, ps, qq, queue, n, n_dev); else launch_trace_t<2, false, false>(s, sc, ps, qq, queue, n, n_dev);
	}
}

void launch_trace_pair(hipStream_t s, const DScene &sc, const DPaths &ps, const DQueues &qc, const uint32_t *queue_c, uint32_t n_c, bool coherent_c,
                       const DQueues &qs, const uint32_t *queue_s, uint32_t n_s, bool coherent_s) {
	DQueues a = qc, b = qs;
	a.coherent = coherent_c ? 1u : 0u; b.coherent = coherent_s ? 1u : 0u;
	// the any-hit phase lives in the closest-hit kernel's footprint: 3 workgroups per CU
	const uint32_t perCu = trace_blocks_per_cu(0);
	if (!b.tune_blocks_per_cu || b.tune_blocks_per_cu > perCu) b.tune_blocks_per_cu = perCu;
	const unsigned blocks = std::max(trace_plan(n_c, 0, a).blocks, trace_plan(n_s, 1, b).blocks);
	if (!blocks) return;
	hipLaunchKernelGGL(k_trace_pair, dim3(blocks), dim3(kTraceBlock), 0, s, trace_scene(sc), ps, a, b, queue_c, n_c, queue_s, n_s);
}

void launch_prep(hipStream_t s, const uint32_t *cur, uint32_t *next_set, BinView *views_dev, uint32_t bin_seg_cap,
                 unsigned long long *dev_stats) {
	hipLaunchKernelGGL(k_prep, dim3(1), dim3(256), 0, s, cur, next_set, views_dev, bin_seg_cap, dev_stats);
}

} // namespace mg
