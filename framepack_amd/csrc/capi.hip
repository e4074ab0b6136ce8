// Library-level entry points of libframepack_hip.so (see include/framepack_hip.h).
#include "fp_common.h"

extern "C" int fp_abi_version(void) { return 5; }
