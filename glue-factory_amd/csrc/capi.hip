// ABI bookkeeping for libgf_amd.so (include/gf_amd.h).
#include "gf_amd.h"

extern "C" int gf_abi_version(void) { return GF_AMD_ABI_VERSION; }
