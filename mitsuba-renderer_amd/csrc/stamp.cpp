// stamp.cpp -- which sources this libmtsgpu.so was built from.  The Makefile hashes every source and header of the
// library (sha256 of their concatenation in the order of SRCS then HDRS, first 16 hex digits) into MTSGPU_SOURCE_HASH
// and rebuilds this file whenever one of them changes; build() (mitsuba-renderer_amd/__init__.py) recomputes the hash
// and forces a full rebuild when the loaded library reports another one, so a stale binary cannot ship silently.
#include "../../include/mtsgpu.h"
#ifndef MTSGPU_SOURCE_HASH
#define MTSGPU_SOURCE_HASH "unstamped"
#endif
extern "C" const char *mtsgpu_source_hash(void) { return MTSGPU_SOURCE_HASH; }
