// pj_lab.h -- the LABORATORY of the polar kernels (pj_blocked.hip, pj_math.inc), kept out of the product's sources.
//
// The product kernels use the names below as if they were part of the language: in the product build every one of them is nothing, or
// the product's compile-time constant (nine rotation iterations, the first one peeled, every store).  Only the separate development
// build (python -m tetsim_amd.build --ablation: -DTETSIM_ABLATION -> libtetsim_hip_ablation.so, never shipped, `library.ablation` in
// bench.py's line) gives them bodies: a run-time mode word (TETSIM_DEBUG_ITERS / _SKIP_REST_STORE / _NO_PEEL / _STAGGER), per-tile
// phase stamps (TETSIM_DEBUG_TRACE; tools/attic/trace_tet.py, frame_trace.py) and the rotation-iteration histogram (TETSIM_DEBUG_ITER_HIST;
// tools/attic/rotation_iterations.py).  Macros only -- the functions they call live in pj_blocked_lab.inc.
#pragma once

// the product's rotation-iteration count (tools/mutation_check.sh and tools/iteration_floor.sh mutate THIS line in a copy of the tree)
#define TETSIM_ROTATION_ITERATIONS 9

#ifndef TETSIM_ABLATION
// ---- the product --------------------------------------------------------------------------------------------------------------
#define TETSIM_DBG_PARAM                      /* no mode word among the kernel arguments */
#define TETSIM_DBG_ARG
#define TETSIM_DBG_LAUNCH
#define TETSIM_DBG_STORE_REST true
#define TETSIM_LAB_FRAME_ITERS TETSIM_ROTATION_ITERATIONS
#define TETSIM_W2LOG_PARAM
#define TETSIM_W2LOG(i, v) do { } while (0)
#define TETSIM_LAB_TILE_BEGIN() do { } while (0)
#define TETSIM_STAMP(i) do { } while (0)
#define TETSIM_LAB_SOLVE_TET(cur, rest, q_old, q_new, goal, cc) \
    pj_solve_tet(cur, rest, q_old, q_new, goal, TETSIM_ROTATION_ITERATIONS, true, kLean, !kLean, &cc, d.rot_exit_w2)
#define TETSIM_LAB_FRAME_BEGIN() do { } while (0)
#define FRAME_STAMP(i) do { } while (0)
#define FRAME_POLL() do { } while (0)
#define TETSIM_LAB_FRAME_END() do { } while (0)
#else
// ---- the development build ----------------------------------------------------------------------------------------------------
// kernels take a run-time `dbg` word (bits 0-3: iterations, bit 4: skip the rest-shape write-back, bit 6: no peel, bits 8-15: stagger,
// bit 16: stagger map); the physics is wrong for anything but the defaults
#define TETSIM_DBG_PARAM , uint32_t dbg
#define TETSIM_DBG_ARG , dbg
#define TETSIM_DBG_LAUNCH , tet_mode()
#define TETSIM_DBG_STORE_REST (!(dbg & 16u))
#define TETSIM_LAB_FRAME_ITERS 9              /* (the ablation knobs belong to the per-substep kernel) */
#define TETSIM_W2LOG_PARAM , float* w2log = nullptr
#define TETSIM_W2LOG(i, v) do { if (w2log) w2log[(i)] = (v); } while (0)
// per-tile phase timestamps + the hardware id of the tile's first wave; TETSIM_DEBUG_STAGGER=<s_sleep units of 64 cycles>: workgroups of
// the first round delay their loads by (slot index) x that much, slot index guessed from the dispatch order in two ways
#define TETSIM_STAMP(i) do { if (d.trace && tid == 0) d.trace[8ull * b + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#define TETSIM_LAB_TILE_BEGIN() do {                                                                                              \
        if (d.trace && tid == 0) d.trace[8ull * b + 7] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)); /* HW_ID */ \
        TETSIM_STAMP(0);                                                                                                           \
        if (const uint32_t st_ = (dbg >> 8) & 0xffu; st_ != 0u && blockIdx.x < 2048u) {                                            \
            const uint32_t k_ = (dbg & 0x10000u) ? (blockIdx.x >> 3) & 7u : (blockIdx.x >> 8) & 7u;                                \
            for (uint32_t i_ = 0; i_ < k_ * st_; i_++) __builtin_amdgcn_s_sleep(1);                                                \
        }                                                                                                                          \
    } while (0)
#define TETSIM_LAB_SOLVE_TET(cur, rest, q_old, q_new, goal, cc) do {                                                              \
        float w2log_[9];                                                                                                           \
        for (int i_ = 0; i_ < 9; i_++) w2log_[i_] = -1.0f;   /* -1: iteration not executed (the wave had left the loop) */          \
        pj_solve_tet(cur, rest, q_old, q_new, goal, static_cast<int>(dbg & 15u), !(dbg & 64u), kLean, !kLean, &cc, d.rot_exit_w2,   \
                     d.iter_hist ? w2log_ : nullptr);                                                                              \
        if (d.iter_hist) pjb_log_iterations(d.iter_hist, w2log_);                                                                  \
    } while (0)
// the frame kernel: thread 0 adds up the cycles of each phase over the call (tools/attic/frame_trace.py)
#define TETSIM_LAB_FRAME_BEGIN() unsigned long long fr_acc[5] = {0, 0, 0, 0, 0}, fr_last = 0, fr_polls = 0
#define FRAME_STAMP(i) do { if (d.trace && tid == 0) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); if ((i) > 0) fr_acc[(i) > 0 ? (i) - 1 : 0] += now_ - fr_last; fr_last = now_; } } while (0)
#define FRAME_POLL() do { if (tid == 0) fr_polls++; } while (0)
#define TETSIM_LAB_FRAME_END() do {                                                                                                \
        if (d.trace && tid == 0) {                                                                                                 \
            for (int i_ = 0; i_ < 5; i_++) d.trace[8ull * b + i_] = fr_acc[i_];                                                    \
            d.trace[8ull * b + 5] = fr_polls; d.trace[8ull * b + 6] = n;                                                           \
            d.trace[8ull * b + 7] = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11));   /* XCC_ID */                           \
        }                                                                                                                          \
    } while (0)
#endif
