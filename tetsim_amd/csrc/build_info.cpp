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
    // bit i = the i-th name is set in the environment AND this build reads it.  The positions are part of the ABI (tetsim_amd/_capi.py,
    // the N-API addon); names marked lab are A/B switches of settled choices that only the development build looks at (body.h: lab_env).
    static const struct { const char* name; bool lab; } kEnv[] = {
        {"TETSIM_DEBUG_LOOPBACK_HALO", false}, {"TETSIM_DEBUG_LOOPBACK_COPY", false}, {"TETSIM_DEBUG_ONE_STREAM", false}, {"TETSIM_DEBUG_GROUP_SYNC", false},
        {"TETSIM_DEBUG_HOSTPROF", false}, {"TETSIM_DEBUG_TRACE", true}, {"TETSIM_HALO_SYNC", false}, {"TETSIM_HALO_GRAPH", false}, {"TETSIM_DEBUG_LOOPBACK_DELAY_US", false},
        {"TETSIM_NH_QUADS", true}, {"TETSIM_FUSED_PARTICLE_PASS", false}, {"TETSIM_FRAME_KERNEL", false}, {"TETSIM_FRAME_LOCAL", true}, {"TETSIM_NH_FOLD", true},
        {"TETSIM_HALO_ALIGNED_TILES", true}, {"TETSIM_HALO_FOLD_WAIT", false}, {"TETSIM_QUAD", false}, {"TETSIM_QUAD_POLL_DELAY", true}, {"TETSIM_NH_FRAME", true},
        {"TETSIM_NH_ONE_LAUNCH", false}, {"TETSIM_PJ_ONE_LAUNCH", false}};
    for (unsigned i = 0; i < sizeof(kEnv) / sizeof(kEnv[0]); i++) {
#ifndef TETSIM_ABLATION
        if (kEnv[i].lab) continue;
#endif
        if (std::getenv(kEnv[i].name)) out->debug_env |= 1u << i;
    }
    std::strncpy(out->source_sha, TETSIM_SOURCE_SHA, sizeof(out->source_sha) - 1);
    std::strncpy(out->kernel_sha, TETSIM_KERNEL_SHA, sizeof(out->kernel_sha) - 1);
    return TETSIM_OK;
}
