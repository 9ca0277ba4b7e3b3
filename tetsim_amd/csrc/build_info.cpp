// build_info.cpp -- tetsim_library_info: which sources / flags this libtetsim_hip.so was built from (stamped by
// tetsim_amd/build.py through -DTETSIM_SOURCE_SHA / -DTETSIM_KERNEL_SHA) and whether it is the ablation build.
#include <cstdlib>
#include <cstring>

#include "../../include/tetsim.h"

#ifndef TETSIM_SOURCE_SHA
#define TETSIM_SOURCE_SHA "unstamped"
#endif
#ifndef TETSIM_KERNEL_SHA
#define TETSIM_KERNEL_SHA "unstamped"
#endif

extern "C" int tetsim_library_info(TetSimLibraryInfo* out) {
    if (!out) return TETSIM_EINVAL;
    std::memset(out, 0, sizeof(*out));
    out->abi = TETSIM_ABI_VERSION;
#ifdef TETSIM_ABLATION
    out->ablation = 1;
#endif
    static const char* const kEnv[] = {"TETSIM_DEBUG_LOOPBACK_HALO", "TETSIM_DEBUG_LOOPBACK_COPY", "TETSIM_DEBUG_ONE_STREAM", "TETSIM_DEBUG_GROUP_SYNC",
                                       "TETSIM_DEBUG_HOSTPROF", "TETSIM_DEBUG_TRACE", "TETSIM_HALO_SYNC", "TETSIM_HALO_GRAPH", "TETSIM_DEBUG_LOOPBACK_DELAY_US", "TETSIM_NH_QUADS", "TETSIM_FUSED_PARTICLE_PASS",
                   "TETSIM_FRAME_KERNEL", "TETSIM_FRAME_LOCAL", "TETSIM_NH_FOLD", "TETSIM_HALO_ALIGNED_TILES", "TETSIM_HALO_FOLD_WAIT", "TETSIM_QUAD", "TETSIM_QUAD_POLL_DELAY", "TETSIM_NH_FRAME"};
    for (unsigned i = 0; i < sizeof(kEnv) / sizeof(kEnv[0]); i++)
        if (std::getenv(kEnv[i])) out->debug_env |= 1u << i;
    std::strncpy(out->source_sha, TETSIM_SOURCE_SHA, sizeof(out->source_sha) - 1);
    std::strncpy(out->kernel_sha, TETSIM_KERNEL_SHA, sizeof(out->kernel_sha) - 1);
    return TETSIM_OK;
}
