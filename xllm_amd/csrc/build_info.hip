// build_info.hip -- which sources this library was built from. The Makefile passes -DXM_BUILD_DIGEST="<sha256>" computed
// over $(DIGEST_SRCS) (every kernel source, the private headers, the Makefile and include/xllm_mi355.h, byte-sorted by
// name) and rebuilds this file whenever one of them changes; tools/source_digest.py --lib computes the same value from the
// tree, and the first GPU test compares the two (a stale prebuilt .so shipped next to newer sources fails there).
#include "common.h"

#ifndef XM_BUILD_DIGEST
#define XM_BUILD_DIGEST "unknown"
#endif

extern "C" {
XM_API const char* xllm_mi355_build_digest(void) { return XM_BUILD_DIGEST; }
}
