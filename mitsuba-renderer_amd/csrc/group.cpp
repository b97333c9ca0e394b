// group.cpp -- several GPUs behind ONE host process (include/mtsgpu.h "several GPUs in one process").
//
// The reference merges the work results of its workers by per-pixel summation under a mutex
// (BlockedRenderProcess::processResult -> Film::putImageBlock, src/librender/renderproc.cpp:123-130,
// src/films/mfilm.cpp:118-143).  Here every member context renders its share of the ImageBlock tiles into a
// full-frame film on its own GPU and the films are summed once per frame: one ncclReduce over RCCL/xGMI, or peer
// copies added in member order when a reproducible sum order is asked for (or RCCL is not usable).
// librccl is loaded with dlopen the first time a group needs a collective: the library itself has no
// load-time dependency on it, and a process that already carries an RCCL (PyTorch does) keeps using that copy.
#include "ctx.h"
#include <rccl/rccl.h>
#include <dlfcn.h>
#include <atomic>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <set>
#include <thread>

using namespace mg;

namespace {

struct Rccl {
	void *handle = nullptr;
	decltype(&ncclCommInitAll) commInitAll = nullptr;
	decltype(&ncclCommDestroy) commDestroy = nullptr;
	decltype(&ncclGroupStart) groupStart = nullptr;
	decltype(&ncclGroupEnd) groupEnd = nullptr;
	decltype(&ncclReduce) reduce = nullptr;
	decltype(&ncclGetErrorString) errorString = nullptr;
	decltype(&ncclCommCount) commCount = nullptr;            // optional: the self-check asks every communicator for its size

	bool load(std::string &why) {
		// MTSGPU_RCCL_LIB names the library to load (a site's own build; tests point it at a file that does not exist to
		// exercise the fallback); otherwise the usual sonames
		const char *forced = getenv("MTSGPU_RCCL_LIB");
		if (forced && *forced) {
			handle = dlopen(forced, RTLD_NOW | RTLD_LOCAL);
		} else {
			for (const char *name : { "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so" }) {
				handle = dlopen(name, RTLD_NOW | RTLD_LOCAL);
				if (handle) break;
			}
		}
		if (!handle) { const char *e = dlerror(); why = std::string("dlopen(") + (forced && *forced ? forced : "librccl") + ") failed: " + (e ? e : "?"); return false; }
		commInitAll = (decltype(commInitAll)) dlsym(handle, "ncclCommInitAll");
		commDestroy = (decltype(commDestroy)) dlsym(handle, "ncclCommDestroy");
		groupStart = (decltype(groupStart)) dlsym(handle, "ncclGroupStart");
		groupEnd = (decltype(groupEnd)) dlsym(handle, "ncclGroupEnd");
		reduce = (decltype(reduce)) dlsym(handle, "ncclReduce");
		errorString = (decltype(errorString)) dlsym(handle, "ncclGetErrorString");
		commCount = (decltype(commCount)) dlsym(handle, "ncclCommCount");
		if (!commInitAll || !commDestroy || !groupStart || !groupEnd || !reduce || !errorString) {
			why = "librccl lacks one of ncclCommInitAll / ncclCommDestroy / ncclGroupStart / ncclGroupEnd / ncclReduce";
			dlclose(handle); handle = nullptr;
			return false;
		}
		return true;
	}
};

} // namespace

struct mtsgpu_group {
	std::vector<mtsgpu_ctx *> members;
	std::vector<int> devices;
	bool distinct = false;                 // every member on its own GPU
	Rccl rccl;
	std::vector<ncclComm_t> comms;         // one per member when RCCL is usable
	std::string rcclNote;                  // why RCCL is not used, if it is not
	float *staging = nullptr; size_t stagingFloats = 0;   // on members[0]'s device: a peer's film (ordered sum) / the RCCL result
	int lastReduceKind = 0;
	int rcclRanks = 0;                     // ranks of the communicator that passed the self-check (0: none did)
	std::string reduceNote;                // why the last frame fell back to the ordered sum ("" when it did not)
	bool testFailReduce = false;           // tests: the next collective reports a failure (mtsgpu_group_set_tuning "rccl_fail")
	std::string error;
};

namespace {

int gfail(mtsgpu_group *g, int code, const char *fmt, ...) {
	char buf[512];
	va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof(buf), fmt, ap); va_end(ap);
	g_lastError = buf;
	if (g) g->error = buf;
	return code;
}

// the same call on every member, concurrently (scene upload to 8 GPUs is 8 PCIe transfers)
template <typename F> int forAll(mtsgpu_group *g, const char *what, F &&f) {
	const size_t n = g->members.size();
	std::vector<int> rc(n, 0);
	std::vector<std::thread> th;
	for (size_t i = 1; i < n; ++i) th.emplace_back([&, i] { rc[i] = f(g->members[i], (int) i); });
	rc[0] = f(g->members[0], 0);
	for (auto &t : th) t.join();
	for (size_t i = 0; i < n; ++i)
		if (rc[i]) return gfail(g, rc[i], "%s, member %zu (device %d): %s", what, i, g->devices[i], mtsgpu_last_error(g->members[i]));
	return 0;
}

bool selfCheckComms(mtsgpu_group *g, std::string &why) {
	const int n = (int) g->members.size();
	for (int i = 0; i < n; ++i) {
		if (!g->comms[i]) { why = "ncclCommInitAll left member " + std::to_string(i) + " without a communicator"; return false; }
		if (g->rccl.commCount) {
			int cnt = -1;
			const ncclResult_t r = g->rccl.commCount(g->comms[i], &cnt);
			if (r != ncclSuccess || cnt != n) {
				why = "communicator of member " + std::to_string(i) + " reports " + std::to_string(cnt) + " ranks, the group has " + std::to_string(n);
				return false;
			}
		}
	}
	std::vector<float *> buf(n, nullptr);
	auto release = [&]() { for (int i = 0; i < n; ++i) if (buf[i]) { (void) hipSetDevice(g->devices[i]); (void) hipFree(buf[i]); } };
	hipError_t he = hipSuccess;
	const float one = 1.0f, zero = 0.0f;
	for (int i = 0; i < n && he == hipSuccess; ++i) {
		if ((he = hipSetDevice(g->devices[i])) != hipSuccess) break;
		if ((he = hipMalloc((void **) &buf[i], 2 * sizeof(float))) != hipSuccess) break;
		if ((he = hipMemcpy(buf[i], &one, sizeof(float), hipMemcpyHostToDevice)) != hipSuccess) break;
		he = hipMemcpy(buf[i] + 1, &zero, sizeof(float), hipMemcpyHostToDevice);
	}
	if (he != hipSuccess) { why = std::string("buffers of the probe: ") + hipGetErrorString(he); release(); return false; }
	ncclResult_t r = g->testFailReduce ? ncclInternalError : g->rccl.groupStart();
	if (r == ncclSuccess) {
		for (int i = 0; i < n && r == ncclSuccess && he == hipSuccess; ++i) {
			if ((he = hipSetDevice(g->devices[i])) != hipSuccess) break;
			r = g->rccl.reduce(buf[i], buf[i] + 1, 1, ncclFloat, ncclSum, 0, g->comms[i], g->members[i]->stream);
		}
		const ncclResult_t r2 = g->rccl.groupEnd();
		if (r == ncclSuccess) r = r2;
	}
	for (int i = 0; i < n && he == hipSuccess && r == ncclSuccess; ++i)
		if ((he = hipSetDevice(g->devices[i])) == hipSuccess) he = hipStreamSynchronize(g->members[i]->stream);
	float sum = -1.0f;
	if (he == hipSuccess && r == ncclSuccess && (he = hipSetDevice(g->devices[0])) == hipSuccess)
		he = hipMemcpy(&sum, buf[0] + 1, sizeof(float), hipMemcpyDeviceToHost);
	release();
	if (r != ncclSuccess) { why = std::string("ncclReduce of the probe: ") + (g->testFailReduce ? "failure injected by the rccl_fail test knob" : g->rccl.errorString(r)); return false; }
	if (he != hipSuccess) { why = std::string("probe: ") + hipGetErrorString(he); return false; }
	if (sum != (float) n) { why = "a sum of ones over " + std::to_string(n) + " members arrived as " + std::to_string(sum); return false; }
	return true;
}

// librccl and the communicator over the group's devices, created the first time a collective is wanted
bool ensureComms(mtsgpu_group *g) {
	if (!g->comms.empty()) return true;
	if (!g->distinct || !g->rcclNote.empty()) return false;
	if (!g->rccl.load(g->rcclNote)) return false;
	g->comms.assign(g->members.size(), nullptr);
	const ncclResult_t r = g->rccl.commInitAll(g->comms.data(), (int) g->members.size(), g->devices.data());
	if (r != ncclSuccess) {
		g->rcclNote = std::string("ncclCommInitAll: ") + g->rccl.errorString(r);
		g->comms.clear();
		return false;
	}
	// Self-check before the first frame depends on it: every member got a communicator, every communicator reports the
	// group's size, and a one-float sum of ones arrives at the root as the number of members.  A group that fails it adds its
	// films up in member order and says why (mtsgpu_group_reduce_note).
	std::string why;
	if (!selfCheckComms(g, why)) {
		for (ncclComm_t cm : g->comms) if (cm) (void) g->rccl.commDestroy(cm);
		g->comms.clear();
		g->rcclNote = "RCCL self-check: " + why;
		(void) hipGetLastError();
		return false;
	}
	g->rcclRanks = (int) g->members.size();
	return true;
}

#define GHIP(g, expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) \
	return gfail(g, MTSGPU_EHIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); } while (0)

} // namespace

extern "C" {

int mtsgpu_create_multi(int ndev, const int *devs, mtsgpu_group **out) {
	if (!out) return gfail(nullptr, MTSGPU_EINVAL, "out is null");
	*out = nullptr;
	if (ndev <= 0 || ndev > 64 || !devs) return gfail(nullptr, MTSGPU_EINVAL, "bad device list");
	mtsgpu_group *g = new mtsgpu_group();
	for (int i = 0; i < ndev; ++i) {
		mtsgpu_ctx *c = nullptr;
		const int rc = mtsgpu_create(devs[i], &c);
		if (rc) {
			const std::string msg = mtsgpu_last_error(nullptr);
			mtsgpu_group_destroy(g);
			return gfail(nullptr, rc, "member %d (device %d): %s", i, devs[i], msg.c_str());
		}
		g->members.push_back(c);
		g->devices.push_back(devs[i]);
	}
	g->distinct = std::set<int>(g->devices.begin(), g->devices.end()).size() == g->devices.size();
	if (ndev > 1 && g->distinct) {
		// member 0 sums the films: let it reach its peers' memory directly over xGMI (not fatal if refused)
		(void) hipSetDevice(g->devices[0]);
		for (int i = 1; i < ndev; ++i) {
			int can = 0;
			if (hipDeviceCanAccessPeer(&can, g->devices[0], g->devices[i]) == hipSuccess && can)
				(void) hipDeviceEnablePeerAccess(g->devices[i], 0);
		}
		(void) hipGetLastError();
	}
	if (!g->distinct)
		g->rcclNote = "members share a device: an RCCL communicator needs one GPU per rank";
	*out = g;
	return 0;
}

void mtsgpu_group_destroy(mtsgpu_group *g) {
	if (!g) return;
	for (ncclComm_t c : g->comms) if (c) (void) g->rccl.commDestroy(c);
	if (g->staging && !g->devices.empty()) { (void) hipSetDevice(g->devices[0]); (void) hipFree(g->staging); }
	for (mtsgpu_ctx *c : g->members) mtsgpu_destroy(c);
	// the RCCL handle stays loaded: unloading a library that owns device state is not safe at this point
	delete g;
}

int mtsgpu_group_size(const mtsgpu_group *g) { return g ? (int) g->members.size() : 0; }

mtsgpu_ctx *mtsgpu_group_ctx(mtsgpu_group *g, int i) {
	return (g && i >= 0 && i < (int) g->members.size()) ? g->members[i] : nullptr;
}

const char *mtsgpu_group_last_error(const mtsgpu_group *g) { return g ? g->error.c_str() : g_lastError.c_str(); }

int mtsgpu_group_last_reduce_kind(const mtsgpu_group *g) { return g ? g->lastReduceKind : -1; }

int mtsgpu_group_rccl_ranks(const mtsgpu_group *g) { return g ? g->rcclRanks : 0; }

int mtsgpu_group_upload_scene(mtsgpu_group *g, const mtsgpu_scene *scene) {
	if (!g) return gfail(nullptr, MTSGPU_EINVAL, "null group");
	return forAll(g, "upload_scene", [&](mtsgpu_ctx *c, int) { return mtsgpu_upload_scene(c, scene); });
}

int mtsgpu_group_set_camera(mtsgpu_group *g, const mtsgpu_camera *cam) {
	if (!g) return gfail(nullptr, MTSGPU_EINVAL, "null group");
	for (size_t i = 0; i < g->members.size(); ++i)
		if (int rc = mtsgpu_set_camera(g->members[i], cam)) return gfail(g, rc, "set_camera: %s", mtsgpu_last_error(g->members[i]));
	return 0;
}

int mtsgpu_group_set_integrator(mtsgpu_group *g, int max_depth, int rr_depth, int strict_normals) {
	if (!g) return gfail(nullptr, MTSGPU_EINVAL, "null group");
	for (size_t i = 0; i < g->members.size(); ++i)
		if (int rc = mtsgpu_set_integrator(g->members[i], max_depth, rr_depth, strict_normals)) return gfail(g, rc, "set_integrator: %s", mtsgpu_last_error(g->members[i]));
	return 0;
}

int mtsgpu_group_set_sampler(mtsgpu_group *g, int kind, uint32_t spp, int ld_depth, uint64_t seed) {
	if (!g) return gfail(nullptr, MTSGPU_EINVAL, "null group");
	for (size_t i = 0; i < g->members.size(); ++i)
		if (int rc = mtsgpu_set_sampler(g->members[i], kind, spp, ld_depth, seed)) return gfail(g, rc, "set_sampler: %s", mtsgpu_last_error(g->members[i]));
	return 0;
}

int mtsgpu_group_set_rfilter(mtsgpu_group *g, float size_x, float size_y, const float *values) {
	if (!g) return gfail(nullptr, MTSGPU_EINVAL, "null group");
	for (size_t i = 0; i < g->members.size(); ++i)
		if (int rc = mtsgpu_set_rfilter(g->members[i], size_x, size_y, values)) return gfail(g, rc, "set_rfilter: %s", mtsgpu_last_error(g->members[i]));
	return 0;
}

namespace {

// keeps the calling thread's current HIP device: the Mitsuba plugin calls from its RenderJob thread, which owns other state
struct DeviceGuard {
	int saved = -1;
	DeviceGuard() { if (hipGetDevice(&saved) != hipSuccess) saved = -1; }
	~DeviceGuard() { if (saved >= 0) (void) hipSetDevice(saved); }
};

int ensureStaging(mtsgpu_group *g, size_t count) {
	if (g->stagingFloats >= count) return 0;
	GHIP(g, hipSetDevice(g->devices[0]));
	if (g->staging) (void) hipFree(g->staging);
	g->staging = nullptr; g->stagingFloats = 0;
	GHIP(g, hipMalloc((void **) &g->staging, count * sizeof(float)));
	g->stagingFloats = count;
	return 0;
}

// ONE ncclReduce(sum, f32, root 0) over the members' films.  The result is received in the staging buffer and copied
// over member 0's film only when every step succeeded, so that a collective that fails half way leaves the films as the
// members rendered them and the caller can still add them up in member order.  false + *why on any failure.
bool reduceWithRccl(mtsgpu_group *g, size_t count, std::string &why) {
	const int n = (int) g->members.size();
	mtsgpu_ctx *root = g->members[0];
	const bool inject = g->testFailReduce;                   // fault injection for tests (mtsgpu_group_set_tuning "rccl_fail")
	ncclResult_t r = g->rccl.groupStart();
	hipError_t he = hipSuccess;
	if (r == ncclSuccess) {
		for (int i = 0; i < n && r == ncclSuccess && he == hipSuccess; ++i) {
			he = hipSetDevice(g->devices[i]);
			if (he != hipSuccess) break;
			mtsgpu_ctx *c = g->members[i];
			if (inject) { r = ncclInternalError; break; }
			r = g->rccl.reduce(c->film, i == 0 ? g->staging : c->film, count, ncclFloat, ncclSum, 0, g->comms[i], c->stream);
		}
		const ncclResult_t r2 = g->rccl.groupEnd();           // always: an open group would swallow the next frame's calls
		if (r == ncclSuccess) r = r2;
	}
	if (he != hipSuccess) { why = std::string("hipSetDevice inside the RCCL group: ") + hipGetErrorString(he); return false; }
	if (r != ncclSuccess) {
		why = std::string("ncclReduce: ") + (inject ? "failure injected by the rccl_fail test knob" : g->rccl.errorString(r));
		return false;
	}
	for (int i = 0; i < n; ++i) {
		if ((he = hipSetDevice(g->devices[i])) != hipSuccess || (he = hipStreamSynchronize(g->members[i]->stream)) != hipSuccess) {
			why = std::string("waiting for the RCCL reduce on member ") + std::to_string(i) + ": " + hipGetErrorString(he);
			return false;
		}
	}
	if ((he = hipSetDevice(g->devices[0])) != hipSuccess
	    || (he = hipMemcpyAsync(root->film, g->staging, count * sizeof(float), hipMemcpyDeviceToDevice, root->stream)) != hipSuccess
	    || (he = hipStreamSynchronize(root->stream)) != hipSuccess) {
		why = std::string("copying the reduced film: ") + hipGetErrorString(he);
		return false;
	}
	return true;
}

} // namespace

int mtsgpu_group_render(mtsgpu_group *g, int block_size, int ordered_reduce, volatile const int *cancel) {
	if (!g) return gfail(nullptr, MTSGPU_EINVAL, "null group");
	DeviceGuard keepDevice;
	const int n = (int) g->members.size();
	g->reduceNote.clear();
	// 1 + 2: every member clears its film and renders its part of the tiles, one host thread per GPU
	int rc = forAll(g, "render", [&](mtsgpu_ctx *c, int i) {
		if (int r = mtsgpu_set_tiles(c, block_size, i, n)) return r;
		if (int r = mtsgpu_clear_film(c)) return r;
		return mtsgpu_render(c, cancel);          // returns with the member's stream idle
	});
	if (rc) return rc;
	if (n == 1 && ordered_reduce != 2) return 0;
	// 3: Film::putImageBlock -- the per-GPU films are summed into member 0's film
	mtsgpu_ctx *root = g->members[0];
	const size_t count = (size_t) root->cam.width * root->cam.height * 5;
	for (int i = 1; i < n; ++i)
		if ((size_t) g->members[i]->cam.width * g->members[i]->cam.height * 5 != count)
			return gfail(g, MTSGPU_ESTATE, "member %d has a different film size", i);
	if (ordered_reduce == 2 && !ensureComms(g))
		return gfail(g, MTSGPU_EHIP, "RCCL is not usable: %s", g->rcclNote.c_str());
	// the staging buffer first: failing to allocate it says nothing about RCCL.  It receives the collective's result and the
	// films of peers on other devices; members that all share one device and add their films in order need none
	bool distinctPeer = false;
	for (int i = 1; i < n; ++i) distinctPeer = distinctPeer || g->devices[i] != g->devices[0];
	if (distinctPeer || ordered_reduce == 2 || (ordered_reduce != 1 && n > 1))
		if (int r = ensureStaging(g, count)) return r;
	if (ordered_reduce != 1) {
		if (ensureComms(g)) {
			std::string why;
			if (reduceWithRccl(g, count, why)) { g->lastReduceKind = 1; return 0; }
			// The collective failed: the films are untouched (see reduceWithRccl).  Give RCCL up for this group -- a
			// communicator that has failed once is not trusted again -- and add the films up in member order instead.
			for (ncclComm_t cm : g->comms) if (cm) (void) g->rccl.commDestroy(cm);
			g->comms.clear();
			g->rcclRanks = 0;
			g->rcclNote = why;
			(void) hipGetLastError();
		}
		if (n > 1 || ordered_reduce == 2) g->reduceNote = g->rcclNote;
	}
	// ordered sum on member 0's GPU: film_0 += film_1, += film_2, ... (a fixed order, so filters wider than a pixel
	// give the same bits on every run); a peer's film travels over xGMI into a staging buffer first
	GHIP(g, hipSetDevice(g->devices[0]));
	for (int i = 1; i < n; ++i) {
		const float *src = g->members[i]->film;
		if (g->devices[i] != g->devices[0]) {
			GHIP(g, hipMemcpyPeerAsync(g->staging, g->devices[0], src, g->devices[i], count * sizeof(float), root->stream));
			src = g->staging;
		}
		launch_add_film(root->stream, root->film, src, count);
		GHIP(g, hipGetLastError());
	}
	GHIP(g, hipStreamSynchronize(root->stream));
	g->lastReduceKind = 0;
	return 0;
}

const char *mtsgpu_group_reduce_note(const mtsgpu_group *g) { return g ? g->reduceNote.c_str() : ""; }

int mtsgpu_group_set_tuning(mtsgpu_group *g, const char *key, long value) {
	if (!g || !key) return gfail(g, MTSGPU_EINVAL, "null argument");
	if (!strcmp(key, "rccl_fail")) { g->testFailReduce = value != 0; return 0; }
	for (size_t i = 0; i < g->members.size(); ++i)
		if (int r = mtsgpu_set_tuning(g->members[i], key, value))
			return gfail(g, r, "member %zu: %s", i, mtsgpu_last_error(g->members[i]));
	return 0;
}

int mtsgpu_hbm_triad(int device, size_t bytes, int iters, double *gbs) {
	if (!gbs || bytes < 4096 || iters <= 0) return gfail(nullptr, MTSGPU_EINVAL, "bad triad arguments");
	*gbs = 0;
	int nd = 0;
	if (hipGetDeviceCount(&nd) != hipSuccess || device < 0 || device >= nd) return gfail(nullptr, MTSGPU_ENODEV, "no such HIP device");
	GHIP(nullptr, hipSetDevice(device));
	const size_t n = bytes / sizeof(float4);
	float4 *a = nullptr, *b = nullptr, *c = nullptr;
	hipEvent_t e0 = nullptr, e1 = nullptr;
	hipStream_t s = nullptr;
	int rc = 0;
	do {
		if (hipMalloc((void **) &a, n * sizeof(float4)) != hipSuccess || hipMalloc((void **) &b, n * sizeof(float4)) != hipSuccess
		    || hipMalloc((void **) &c, n * sizeof(float4)) != hipSuccess || hipStreamCreate(&s) != hipSuccess
		    || hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) { rc = gfail(nullptr, MTSGPU_EHIP, "triad: allocation failed"); break; }
		(void) hipMemsetAsync(b, 0, n * sizeof(float4), s); (void) hipMemsetAsync(c, 0, n * sizeof(float4), s);
		launch_triad(s, a, b, c, 0.5f, n);                                  // warm-up
		// The rate depends on how many workgroups stream at once (fewer concurrent streams keep more DRAM pages open:
		// profiles/r02b_cumask_triad.txt has 4.6 TB/s with 16 workgroups of 256 per CU and 6.0 TB/s with 2), so the
		// practical roof is the best of a few grid shapes
		hipDeviceProp_t prop;
		const unsigned cus = hipGetDeviceProperties(&prop, device) == hipSuccess ? (unsigned) std::max(1, prop.multiProcessorCount) : 256u;
		float best = 1e30f;
		for (unsigned perCu : { 16u, 4u, 2u, 1u }) {
			for (int i = 0; i < iters && !rc; ++i) {
				(void) hipEventRecord(e0, s);
				launch_triad(s, a, b, c, 0.5f, n, cus * perCu);
				(void) hipEventRecord(e1, s);
				if (hipEventSynchronize(e1) != hipSuccess) { rc = gfail(nullptr, MTSGPU_EHIP, "triad: kernel failed"); break; }
				float ms = 0; (void) hipEventElapsedTime(&ms, e0, e1);
				if (ms > 0 && ms < best) best = ms;
			}
		}
		if (!rc && best < 1e29f) *gbs = 3.0 * (double) (n * sizeof(float4)) / (best * 1e-3) / 1e9;
	} while (false);
	if (e0) (void) hipEventDestroy(e0);
	if (e1) (void) hipEventDestroy(e1);
	if (s) (void) hipStreamDestroy(s);
	if (a) (void) hipFree(a);
	if (b) (void) hipFree(b);
	if (c) (void) hipFree(c);
	return rc;
}

int mtsgpu_gather_roof(int device, size_t footprint_bytes, double *lane_requests_per_s) {
	if (!lane_requests_per_s || footprint_bytes < 4096 || (footprint_bytes & (footprint_bytes - 1)) != 0 || footprint_bytes > (1ull << 32))
		return gfail(nullptr, MTSGPU_EINVAL, "footprint must be a power of two between 4 KiB and 4 GiB");
	*lane_requests_per_s = 0;
	int nd = 0;
	if (hipGetDeviceCount(&nd) != hipSuccess || device < 0 || device >= nd) return gfail(nullptr, MTSGPU_ENODEV, "no such HIP device");
	GHIP(nullptr, hipSetDevice(device));
	hipDeviceProp_t prop;
	GHIP(nullptr, hipGetDeviceProperties(&prop, device));
	const unsigned blocks = (unsigned) prop.multiProcessorCount * 7u;
	const int iters = 1000;
	uint4 *data = nullptr; uint32_t *sink = nullptr;
	hipEvent_t e0 = nullptr, e1 = nullptr; hipStream_t s = nullptr;
	int rc = 0;
	do {
		if (hipMalloc((void **) &data, footprint_bytes) != hipSuccess || hipMalloc((void **) &sink, 4) != hipSuccess || hipStreamCreate(&s) != hipSuccess
		    || hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) { rc = gfail(nullptr, MTSGPU_EHIP, "gather roof: allocation failed"); break; }
		(void) hipMemsetAsync(data, 1, footprint_bytes, s);
		const uint32_t mask = (uint32_t) (footprint_bytes / 16 - 1);
		launch_gather_roof(s, data, mask, iters, blocks, sink);           // warm-up
		float best = 1e30f;
		for (int i = 0; i < 3; ++i) {
			(void) hipEventRecord(e0, s);
			launch_gather_roof(s, data, mask, iters, blocks, sink);
			(void) hipEventRecord(e1, s);
			if (hipEventSynchronize(e1) != hipSuccess) { rc = gfail(nullptr, MTSGPU_EHIP, "gather roof: kernel failed"); break; }
			float ms = 0; (void) hipEventElapsedTime(&ms, e0, e1);
			if (ms > 0 && ms < best) best = ms;
		}
		if (!rc && best < 1e29f) *lane_requests_per_s = (double) blocks * 256.0 * iters / (best * 1e-3);
	} while (false);
	if (e0) (void) hipEventDestroy(e0);
	if (e1) (void) hipEventDestroy(e1);
	if (s) (void) hipStreamDestroy(s);
	if (data) (void) hipFree(data);
	if (sink) (void) hipFree(sink);
	return rc;
}

int mtsgpu_make_camera_crop(const float origin[3], const float target[3], const float up[3], float fov_deg,
                            int film_width, int film_height, int crop_x, int crop_y, int crop_width, int crop_height,
                            mtsgpu_camera *out) {
	if (!origin || !target || !up || !out || film_width <= 0 || film_height <= 0) return gfail(nullptr, MTSGPU_EINVAL, "bad camera arguments");
	if (crop_x < 0 || crop_y < 0 || crop_width <= 0 || crop_height <= 0 || crop_x + crop_width > film_width || crop_y + crop_height > film_height)
		return gfail(nullptr, MTSGPU_EINVAL, "Invalid crop window specification! (film.cpp:41-45)");
	makeCamera(origin, target, up, fov_deg, film_width, film_height, *out);     // raster space of the FULL film
	out->width = crop_width; out->height = crop_height;
	out->crop_offset_x = crop_x; out->crop_offset_y = crop_y;
	out->film_width = film_width; out->film_height = film_height;
	return 0;
}

} // extern "C"
