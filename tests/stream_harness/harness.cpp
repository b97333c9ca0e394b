// C entry points around integration/streamparse.h for tests/test_stream_parsers.py (built with g++ by the test; no Mitsuba).
// prec = 4: Float is float (SINGLE_PRECISION, the reference's default build), 8: double.
#include "streamparse.h"
#include <cstdio>

namespace {
template <typename F> int finish(bool ok, const std::string &err, char *msg, size_t cap) {
	if (msg && cap) snprintf(msg, cap, "%s", ok ? "" : err.c_str());
	return ok ? 0 : 1;
}
}
#define DISPATCH(call_f, call_d) std::string err; const bool ok = (prec == 8) ? (call_d) : (call_f); return finish<float>(ok, err, msg, cap);

extern "C" {
int sp_parse_bsdf(const uint8_t *d, size_t n, int prec, uint32_t *type, float *P, char *msg, size_t cap) {
	DISPATCH(mtsgpu_stream::parseBSDF<float>(d, n, type, P, &err), mtsgpu_stream::parseBSDF<double>(d, n, type, P, &err))
}
int sp_parse_directional(const uint8_t *d, size_t n, int prec, float *P, char *msg, size_t cap) {
	DISPATCH(mtsgpu_stream::parseDirectional<float>(d, n, P, &err), mtsgpu_stream::parseDirectional<double>(d, n, P, &err))
}
int sp_parse_spot(const uint8_t *d, size_t n, int prec, float *P, char *msg, size_t cap) {
	DISPATCH(mtsgpu_stream::parseSpot<float>(d, n, P, &err), mtsgpu_stream::parseSpot<double>(d, n, P, &err))
}
int sp_parse_collimated(const uint8_t *d, size_t n, int prec, float *P, char *msg, size_t cap) {
	DISPATCH(mtsgpu_stream::parseCollimated<float>(d, n, P, &err), mtsgpu_stream::parseCollimated<double>(d, n, P, &err))
}
int sp_parse_envmap(const uint8_t *d, size_t n, int prec, float *P, uint64_t *off, uint32_t *size, char *msg, size_t cap) {
	size_t o = 0;
	std::string err;
	const bool ok = (prec == 8) ? mtsgpu_stream::parseEnvMapHeader<double>(d, n, P, &o, size, &err) : mtsgpu_stream::parseEnvMapHeader<float>(d, n, P, &o, size, &err);
	*off = o;
	return finish<float>(ok, err, msg, cap);
}
int sp_parse_sphere(const uint8_t *d, size_t n, int prec, float *SP, char *msg, size_t cap) {
	DISPATCH(mtsgpu_stream::parseSphere<float>(d, n, SP, &err), mtsgpu_stream::parseSphere<double>(d, n, SP, &err))
}
}
