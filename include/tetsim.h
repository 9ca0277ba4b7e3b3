/*
 * tetsim.h -- C ABI of libtetsim_hip.so: the MI355X-native XPBD tetrahedral soft-body hot path.
 *
 * This is the drop-in boundary for the per-substep solve of zalo/TetSim.  Every entry point replaces
 * one piece of the reference's `SoftBody` / `SoftBodyGPU` surface (citations are file:line under the
 * reference tree); the N-API shim (tetsim_amd/node/tetsim_napi.cc) and the ctypes host
 * (tetsim_amd/softbody.py) bind these 1:1.  Plain pointers and sizes only; no C++/torch types.
 *
 * Conventions
 *   - every function returns 0 on success or a TETSIM_E* code; tetsim_last_error() gives the text
 *     (the reference reports init problems as an error *string*, SoftbodyGPU.js:379-380);
 *   - all inputs are copied at create; device memory is owned by the handle;
 *   - step calls enqueue work on the handle's HIP stream and return WITHOUT synchronising;
 *     reads synchronise;
 *   - a handle is not thread-safe (the reference is single-threaded JS, World.js:73).
 *   - numbers that are JS `number`s in the reference (dt, physicsParams) are doubles here so the
 *     Neo-Hookean path can reproduce Softbody.js's f64-arithmetic/f32-store results bit for bit.
 */
#ifndef TETSIM_H
#define TETSIM_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TETSIM_ABI_VERSION 5

typedef struct tetsim_body *tetsim_handle;

/* error codes */
enum {
    TETSIM_OK = 0,
    TETSIM_EINVAL = 1,    /* bad argument (null pointer, index out of range, repeated vertex in a tet ...) */
    TETSIM_ENODEVICE = 2, /* no usable HIP device / HIP extension not functional: never falls back to CPU */
    TETSIM_EHIP = 3,      /* a HIP runtime call failed; text in tetsim_last_error */
    TETSIM_ENOMEM = 4,
    TETSIM_ECOMM = 5,     /* RCCL load/init/transfer failure */
    TETSIM_ESTATE = 6     /* call not valid for this handle (e.g. quats of a Neo-Hookean body) */
};

/* which of the reference's two solvers */
enum {
    TETSIM_SOLVER_POLAR_JACOBI = 0,  /* SoftbodyGPU.js: shape-matching / polar decomposition, Jacobi (7 GLSL passes :59-376) */
    TETSIM_SOLVER_NEOHOOKEAN_GS = 1  /* Softbody.js:  Neo-Hookean XPBD, Gauss-Seidel (:91-240) */
};

/* arithmetic mode */
enum {
    /* POLAR_JACOBI: IEEE f32, no FMA contraction, correctly rounded div/sqrt, libm-grade sin -- op-for-op the
     *               GLSL source order.  NEOHOOKEAN_GS: f64 arithmetic with f32 stores exactly where
     *               Softbody.js rounds (bit-exact with the reference CPU solver). */
    TETSIM_PRECISE = 0,
    /* f32 throughout, FMA contraction, hardware rcp/rsq/sin.  Tolerance-level parity (tests state it). */
    TETSIM_FAST = 1
};

/* Gauss-Seidel element order (NEOHOOKEAN_GS only) */
enum {
    /* exactly the sequential order of the caller's tetIds (Softbody.js:207-208), parallelised by
     * dependency levels: two tets run concurrently only if no earlier/later pair shares a vertex. */
    TETSIM_ORDER_ORIGINAL = 0,
    /* greedy graph colouring, tets stably sorted by colour (BASELINE config 4).  The result equals the
     * reference CPU solver run on tetIds permuted by tetsim_get_tet_order(). */
    TETSIM_ORDER_COLOURED = 1,
    /* clusters of <= 8 tets over <= 8 vertices (the 6 tets of a cell on a cell-major lattice), clusters coloured; a GPU lane
     * sweeps its cluster sequentially out of LDS, one launch per cluster colour (8 on the lattice instead of 31 tet colours).
     * Same contract as COLOURED: the result equals the reference CPU solver run on tetIds permuted by
     * tetsim_get_tet_order().  TetSimInfo.num_levels = launches per substep; tet_colour is ignored. */
    TETSIM_ORDER_CLUSTERED = 2
};

/* flags */
enum {
    /* build the particle->(tet,vertex) scatter table exactly as SoftbodyGPU.js:563-577 does, including
     * its `<= 0.0` empty-slot test (:568), which drops tet 0 / vertex 0's contribution, and the 36-slot
     * cap.  Without it every incident tet contributes. */
    TETSIM_FLAG_REF_SLOT_TABLE = 1u << 0,
    /* ignore params.worldBounds and clamp to the constants hard-coded in the collision pass
     * (SoftbodyGPU.js:347).  POLAR_JACOBI only; on by default through tetsim_default_options(). */
    TETSIM_FLAG_REF_FIXED_BOUNDS = 1u << 1,
    /* POLAR_JACOBI + FAST normally runs the blocked formulation (workgroup tiles, LDS-staged particles,
     * per-tile partial sums).  This flag keeps the gather formulation (reference slot order). */
    TETSIM_FLAG_GATHER_FORMULATION = 1u << 2,
    /* POLAR_JACOBI + FAST + blocked: do not carry the rotated rest shape in world space (48 B/tet read + 48 B/tet written
     * per substep, as the reference's textureElem does) but re-derive it as R(q) * rest0 from the quaternion and a constant
     * centred rest shape (SURVEY.md 8(a) design note): identical in exact arithmetic, ~30% less tet-kernel traffic;
     * rounding differs (tolerance-level).  Measured +4% tet-solves/s (the tet kernel is issue-bound).  Off by default: the benchmark measures the reference formulation. */
    TETSIM_FLAG_CONSTANT_REST_SHAPE = 1u << 3,
    /* POLAR_JACOBI: pin the particle(s) the reference's collision pass actually pins.  Its indexFromUV
     * (SoftbodyGPU.js:335-338, "This isn't quite correct") maps texel (px,py) of the R x R position texture to
     * int(uv.x*(R-1)) + int(uv.y*(R-1)*R), not to px + R*py, so `grabId` selects zero, one or two OTHER particles.
     * Off by default (the library pins exactly grabId); on for bit-faithful replays of the reference's grab. */
    TETSIM_FLAG_REF_GRAB_TEXEL = 1u << 4,
    /* partitioned POLAR_JACOBI + TETSIM_FAST (blocked): a ghost region TWO layers deep.  The partition advances its first ghost layer
     * itself and its neighbours' particles cross only every other substep -- half the hand-overs on the substep's critical chain
     * (DESIGN.md 7).  Needs the peer-to-peer halo (tetsim_halo_p2p_connect) before the first step; dt must stay fixed. */
    TETSIM_FLAG_DEEP_GHOSTS = 1u << 5,
    /* POLAR_JACOBI + TETSIM_FAST: end a tet's rotation iterations only where the reference does (|omega| < 1e-9,
     * SoftbodyGPU.js:131 -- unreachable in f32 unless the tet is exactly rigid, so all nine run).  Without the flag FAST ends the
     * CORRECTION iterations 2..9 of a tet (of a wavefront, when all its tets agree) once |omega| < 1e-6 rad: an order of magnitude
     * below the rotation that f32 position rounding alone induces in a centimetre-sized tet a metre from the origin (iteration 1 of
     * a rigid free fall reads 1e-6..3e-5, profiles/r04_rotation_iterations.txt).  Iteration 1 always keeps the reference's test.
     * PRECISE ignores this flag: it always is the reference. */
    TETSIM_FLAG_REF_ROTATION_EXIT = 1u << 6,
    /* POLAR_JACOBI + FAST + blocked: the LEAN tet record (since ABI 5).  The reference streams per tet and substep the carried
     * ("last rotated") rest shape in and out (SoftbodyGPU.js:253-262: 4 corners, 48 B each way) and its quaternion in and out
     * (:181, 16 B each way): 148 B with the staged positions and the weight.  Two of those items are redundant in FAST arithmetic:
     * (a) the quaternion is PURE OUTPUT of a substep -- the rotation applied to the shape is this substep's `rel` alone -- and the
     * carried shape IS the accumulated rotation applied to the rest shape, so the quaternion is recovered from the shape when it
     * is asked for (tetsim_read_quats, the visual mesh, tetsim_save_state) instead of being multiplied up every substep;
     * (b) the shape is kept relative to its own centroid, so its fourth corner is minus the sum of the other three.
     * 92 B per tet instead of 148.  Rounding differs from the default FAST path at tolerance level (the fourth corner is rebuilt,
     * not carried; the read-out quaternion is R = C S0^-1 -> q, equal to the multiplied-up one to ~1e-6, and equal in SIGN as long as
     * a tet turns by less than pi between two read-outs); a zero-volume tet reads back the quaternion it was created with.  Small
     * bodies take the 256-tet-tile kernels (TetSimInfo.fused_particle_pass 2, not 3).  Excludes _CONSTANT_REST_SHAPE and _DEEP_GHOSTS.
     * Off by default: the benchmark's headline measures the reference's formulation, `value_lean` this one (bench.py). */
    TETSIM_FLAG_LEAN_STATE = 1u << 7
};

/* physicsParams (main.js:22-36) -- the keys the hot path reads each substep. */
typedef struct TetSimParams {
    double gravity;        /* Softbody.js:199 ; SoftbodyGPU.js:371 */
    double friction;       /* Softbody.js:224-225 ; SoftbodyGPU.js:352 */
    double devCompliance;  /* Softbody.js:130,161 (NEOHOOKEAN_GS) */
    double volCompliance;  /* Softbody.js:161,165 (NEOHOOKEAN_GS) */
    double worldBounds[6]; /* lo xyz, hi xyz ; Softbody.js:215-216 */
} TetSimParams;

typedef struct TetSimOptions {
    int32_t solver;     /* TETSIM_SOLVER_* */
    int32_t precision;  /* TETSIM_PRECISE / TETSIM_FAST */
    int32_t order;      /* TETSIM_ORDER_* (NEOHOOKEAN_GS) */
    uint32_t flags;     /* TETSIM_FLAG_* */
    int32_t device;     /* HIP device ordinal */
    double density;     /* physicsParams.density, consumed at construction (Softbody.js:32,74) */
    /* Domain decomposition (POLAR_JACOBI).  part_count <= 1: whole mesh on this handle.  Otherwise this
     * handle owns the vertices v with vert_owner[v] == part_index (vert_owner == NULL: the built-in
     * partitioner, tetsim_prep_partition without coordinates) plus a ghost layer; see DESIGN.md 7. */
    int32_t part_count;
    int32_t part_index;
    const int32_t *vert_owner;
    /* NEOHOOKEAN_GS + TETSIM_ORDER_COLOURED: a caller-supplied colour per tet, [num_elems], or NULL for the built-in greedy
     * colouring.  ANY labelling is safe: the solve order is "stable sort by colour", and the parallel levels are derived
     * from that order's true dependencies, so a poor colouring costs speed, never correctness (since ABI 2). */
    const int32_t *tet_colour;
} TetSimOptions;

typedef struct TetSimInfo {
    uint32_t num_particles;      /* vertices of the caller's (global) mesh  -- SoftBody.numParticles */
    uint32_t num_elems;          /* tets of the caller's (global) mesh      -- SoftBody.numElems */
    uint32_t owned_particles;    /* vertices this handle integrates (== num_particles when unpartitioned) */
    uint32_t local_particles;    /* owned + ghost */
    uint32_t local_elems;        /* tets this handle solves (owned + ghost tets) */
    uint32_t owned_elems;        /* tets counted once across partitions (lowest-owner rule) */
    uint32_t num_levels;         /* Gauss-Seidel dependency levels / colours (0 for POLAR_JACOBI) */
    uint32_t max_valence;        /* max incident tets per vertex used by the scatter table */
    uint32_t dropped_slots;      /* (tet,vertex) contributions dropped by TETSIM_FLAG_REF_SLOT_TABLE */
    uint32_t num_neighbours;     /* partitions this handle exchanges a halo with */
    uint64_t device_bytes;       /* HBM allocated by this handle */
    int32_t solver, precision, order, device;
    uint32_t flags;
    uint32_t num_vis_verts;      /* visual vertices attached by tetsim_set_visual_mesh / a .tetsim file (0 = none); since ABI 3 */
    uint32_t num_bodies;         /* independent bodies behind this handle (tetsim_create_batch), 1 otherwise; since ABI 3 */
    uint32_t fused_particle_pass; /* 1: tetsim_step_n runs ONE kernel per substep (particle update fused into the tet kernel's staging,
                                     HISTORY.md 5.4: unpartitioned POLAR_JACOBI + FAST blocked bodies); tetsim_profile then times that kernel.
                                     2: ... and the body is small enough for tetsim_step_n to run ONE persistent kernel per CALL (every
                                     tile's workgroup resident for all n substeps); tetsim_profile still
                                     uses the per-substep kernels, whose results are the same bit for bit.
                                     3: as 2, on 64-tet tiles with one tet and one particle on FOUR lanes (pj_quad.hip: the default for
                                     small carried-rest-shape bodies); tetsim_profile runs the same substep as two launches.  (2 and 3:
                                     tetsim_step is the persistent kernel for ONE substep -- one launch per call.)
                                     4: NEOHOOKEAN_GS, level schedules (TETSIM_ORDER_ORIGINAL / _COLOURED), at most 4,096 particles (PRECISE: and 12,288 tets): they all
                                     fit one CU's LDS and tetsim_step_n -- and tetsim_step -- run a whole call as ONE launch of one workgroup (one per body of a batch)
                                     (nh_kernels.inc); tetsim_profile keeps one launch per level, same arithmetic, same results bit for bit
                                     5: (since ABI 5) POLAR_JACOBI + FAST blocked bodies that keep a tet and a particle kernel per substep (0: >= 2,048 tiles, or a
                                     particle no tile sums): tetsim_step_n runs a whole CALL as ONE launch -- per substep the tiles' workgroups, then the particles',
                                     substep after substep in one grid, partial sums and predictions handed on with the substep's sequence number in their fourth
                                     float (pj_blocked.hip: pjb_call_kernel); tetsim_step and tetsim_profile keep the two kernels, same results bit for bit
                                     (0 remains: partitioned bodies, PRECISE / gather bodies, TETSIM_PJ_ONE_LAUNCH=0) */
    uint32_t total_vis_verts;    /* rows of the visVerts the caller attached (num_vis_verts of them are this handle's: all, unless partitioned); since ABI 5 */
} TetSimInfo;

/* per-kernel HIP-event timing of eagerly launched substeps (tetsim_profile) */
enum { TETSIM_K_TET = 0, TETSIM_K_VERTEX = 1, TETSIM_K_HALO = 2, TETSIM_K_COUNT = 3 };
typedef struct TetSimProfile {
    double total_ms;                 /* wall on the stream for all substeps */
    double kernel_ms[TETSIM_K_COUNT]; /* summed duration per kernel class */
    uint32_t launches[TETSIM_K_COUNT];
    uint32_t substeps;
    uint32_t tets_per_tet_launch;    /* tets one timed TETSIM_K_TET launch processes: all of them, or on a partitioned body with a
                                        halo transport the INTERIOR tiles' (the halo-side tiles -- those touching a ghost or a boundary particle --
                                        run beside them on the halo stream, as do the boundary particles: TETSIM_K_VERTEX then times the
                                        interior particles' kernel) */
} TetSimProfile;

/* --- lifecycle ------------------------------------------------------------------------------- */

/* Fill `o` with the defaults that mirror the reference's GPU demo path
 * (POLAR_JACOBI, PRECISE, REF_SLOT_TABLE|REF_FIXED_BOUNDS, density 1000, device 0, unpartitioned). */
void tetsim_default_options(TetSimOptions *o);
/* physicsParams defaults of main.js:22-36. */
void tetsim_default_params(TetSimParams *p);

/* Replaces `new SoftBody(vertices, tetIds, ...)` (Softbody.js:4-58 + initPhysics :60-87) and
 * `new SoftBodyGPU(...)` (SoftbodyGPU.js:5-56 + initPhysics :487-608): copies the mesh, builds rest
 * data / tables on the host, uploads.  verts = [3*nv] xyz, tets = [4*nt] vertex ids. */
int tetsim_create(const float *verts, uint32_t nv, const int32_t *tets, uint32_t nt,
                  const TetSimOptions *opts, tetsim_handle *out);
/* Several INDEPENDENT bodies behind one handle (the reference steps `softBodies[]` one after the other, main.js:80-84): body b is
 * (verts[b], nv[b], tets[b], nt[b]); the handle's particles / tets are their concatenation (tetsim_get_batch_layout gives the
 * ranges) and every other entry point works on that concatenation -- one tetsim_step_n steps all bodies with ONE launch per
 * kernel: a Dragon-sized body alone fills 15 of the chip's 2,048 workgroup slots, sixty-four of them 960.  All bodies share
 * physicsParams; tetsim_set_grab addresses a particle of the concatenation.  Each body's results equal its solo run BIT FOR BIT
 * (both solvers, both precisions): tiles never span two bodies and are cut per body exactly as for a body alone, the
 * reference's slot-table quirk applies to every body's own first tet, Gauss-Seidel schedules of disjoint bodies are independent
 * (tetsim_get_tet_order then refers to the concatenation).  Unpartitioned only. */
int tetsim_create_batch(const float *const *verts, const uint32_t *nv, const int32_t *const *tets, const uint32_t *nt,
                        uint32_t count, const TetSimOptions *opts, tetsim_handle *out);
/* first_particle / first_elem [num_bodies + 1]: body b owns particles [first_particle[b], first_particle[b+1]) of the handle. */
int tetsim_get_batch_layout(tetsim_handle h, uint32_t *first_particle, uint32_t *first_elem);
void tetsim_destroy(tetsim_handle h);

/* Text of the last error on this handle.  h == NULL: the last error, on this thread, of a failed create or of an entry point that takes
 * SEVERAL handles (tetsim_group_step_n, tetsim_group_refresh_final, tetsim_halo_exchange_local: "partition <i>: <text>"). */
const char *tetsim_last_error(tetsim_handle h);
int tetsim_get_info(tetsim_handle h, TetSimInfo *info);

/* --- the hot path ---------------------------------------------------------------------------- */

/* Replaces `simulate(dt, physicsParams)` (Softbody.js:195-240 ; SoftbodyGPU.js:610-641): ONE substep,
 * enqueued on the handle's stream, no synchronisation. */
int tetsim_step(tetsim_handle h, double dt, const TetSimParams *params);
/* n substeps with the same dt/params/grab as one host call and one launch: the body of the caller's substep loop (main.js:79-84).
 * One replay of a captured HIP graph -- or, where the whole call is ONE kernel (TetSimInfo.fused_particle_pass 5: large
 * unpartitioned POLAR_JACOBI FAST bodies), that kernel launched directly with the call's parameters among its arguments. */
int tetsim_step_n(tetsim_handle h, uint32_t n, double dt, const TetSimParams *params);
/* Block until everything enqueued on the handle's stream(s) has finished.  Partitioned bodies: TETSIM_ECOMM if a device-side
 * halo wait gave up (TETSIM_HALO_TIMEOUT_MS, 30 s by default: a stuck peer, or the two chains of a graph replay sharing one
 * hardware queue).  The substeps since then used stale data: go back to the last checkpoint -- every rank loads the blob it saved
 * with tetsim_save_state at the same substep count (partitions save and load their own, since ABI 4), or rebuilds its partition
 * (same owner map), loads it there and re-attaches the transport -- and the trajectory continues bit for bit.  The handle itself
 * stays usable and steps eagerly (no graph replay of the halo chains) from then on. */
int tetsim_sync(tetsim_handle h);

/* --- state access (synchronising) ------------------------------------------------------------ */

/* Replaces reading `.pos` (Softbody.js:12) / readToCPU (SoftbodyGPU.js:649-653): xyz of the OWNED
 * particles after the last completed substep, [3*owned_particles], in tetsim_get_owned_ids order. */
int tetsim_read_positions(tetsim_handle h, float *out);
/* Zero-copy variant: positions are packed to xyz on the device and copied straight into a pinned
 * host buffer owned by the handle ([3*owned_particles] floats, valid until tetsim_destroy, overwritten by the next
 * call).  *out receives its address: a host wraps it once (N-API external ArrayBuffer, numpy view) and every
 * endFrame() is one device pack kernel + one DMA, no intermediate host copy.  Synchronises. */
int tetsim_read_positions_pinned(tetsim_handle h, const float **out);
int tetsim_read_prev_positions(tetsim_handle h, float *out); /* .prevPos */
int tetsim_read_velocities(tetsim_handle h, float *out);     /* .vel */
/* POLAR_JACOBI: per-tet rotation quaternion xyzw (textureQuat, SoftbodyGPU.js:55,181), [4*local_elems]
 * in tetsim_get_local_tets order. */
int tetsim_read_quats(tetsim_handle h, float *out);
/* Zero-copy variant of tetsim_read_quats (SURVEY.md §8(f)-2 "position/quaternion"; the normal path of SoftbodyGPU.js:440
 * consumes textureQuat every frame): one DMA straight into a pinned host buffer owned by the handle ([4*local_elems]
 * floats in tetsim_get_local_tets order, valid until tetsim_destroy, overwritten by the next call).  Synchronises. */
int tetsim_read_quats_pinned(tetsim_handle h, const float **out);
/* NEOHOOKEAN_GS: `.volError` of the last substep (Softbody.js:163,206,209), summed in the caller's
 * tet order in f64. */
int tetsim_read_vol_error(tetsim_handle h, double *out);
/* Overwrite positions and velocities of the owned particles.  NEOHOOKEAN_GS: that is the whole solver state.
 * POLAR_JACOBI also carries per-tet state (quaternions and the rotated rest shape, SoftbodyGPU.js:49-55 `elems`/`quats`),
 * which this call leaves untouched: to continue a trajectory use tetsim_save_state / tetsim_load_state. */
int tetsim_write_state(tetsim_handle h, const float *pos, const float *vel);
/* Checkpoint / resume of the COMPLETE solver state (both solvers): positions, velocities and, for POLAR_JACOBI, every tet's
 * quaternion and carried rest shape -- everything the reference keeps in its ping-pong render targets (SoftbodyGPU.js:49-55).
 * The blob is only meaningful for a body created from the same mesh with the same options by the same library build family (it
 * starts with a header that tetsim_load_state validates: magic, ABI, solver, precision, flags, counts, and a digest of the mesh --
 * vertices, tets, density, batch layout -- so that a blob of ANOTHER mesh with the same counts is rejected too).  A body restored
 * from a blob continues the original trajectory bit for bit.  Both calls synchronise (both streams; for the partitions of an
 * in-process group the streams of every member: a neighbour's transfer lands in this body's ghost range).
 * PARTITIONED bodies (since ABI 4): every partition saves and loads ITS OWN blob -- its owned particles, its ghosts (the predictions
 * its neighbours sent for the next substep are state too) and its local tets incl. ghost tets; the digest also covers the cut
 * (part_count, part_index, the owner map as this partition sees it), so a blob of another decomposition is rejected.  All ranks save
 * at the same substep count (after tetsim_sync) and restore together: a multi-GPU run that lost a rank rebuilds its partitions with
 * the same owner map, loads the last set of blobs, re-attaches the transport and continues bit for bit.  Transport state is not in
 * the blob (semaphore words are at rest after a sync; the peer-to-peer halo's substep parity stays the restored body's own).
 * ONE RANK PER PROCESS with the peer-to-peer halo (since ABI 5): a neighbour stores into this rank's ghost range from ITS queues, which
 * this rank cannot drain.  tetsim_save_state therefore waits (bounded: TETSIM_HALO_TIMEOUT_MS, TETSIM_ECOMM) until every neighbour's
 * predictions of the last substep have arrived, and never writes into a receive buffer -- a rank may save while its neighbours are
 * already stepping on; no barrier is needed around a save beyond "every rank saves after the same number of substeps".
 * tetsim_load_state overwrites both ghost buffers: the CALLER brackets the ranks' loads with barriers -- no rank loads before every
 * rank has synchronised (tetsim_sync), no rank steps before every rank has loaded (tests/test_gpu_p2p_halo.py shows the sequence).
 * TETSIM_FLAG_LEAN_STATE bodies: the blob holds three carried corners per tet and the quaternions as recovered at the save.  Not
 * supported: bodies with a two-layer ghost region (TETSIM_FLAG_DEEP_GHOSTS). */
int tetsim_state_size(tetsim_handle h, uint64_t *bytes_out);
int tetsim_save_state(tetsim_handle h, void *blob, uint64_t bytes);
int tetsim_load_state(tetsim_handle h, const void *blob, uint64_t bytes);

/* global vertex id of each owned particle, [owned_particles] (identity when unpartitioned) */
int tetsim_get_owned_ids(tetsim_handle h, int32_t *out);
/* global tet id of each local tet, [local_elems] */
int tetsim_get_local_tets(tetsim_handle h, int32_t *out);
/* NEOHOOKEAN_GS: the order in which tets are solved, as indices into the caller's tetIds, [num_elems]. */
int tetsim_get_tet_order(tetsim_handle h, int32_t *out);
/* NEOHOOKEAN_GS: first solve-order position of every level, [num_levels + 1]. */
int tetsim_get_level_offsets(tetsim_handle h, int32_t *out);
/* rest data as the reference computes it (Softbody.js:60-87): invMass [nv] */
int tetsim_read_inv_mass(tetsim_handle h, float *out);

/* --- embedded visual mesh --------------------------------------------------------- */

/* Attach the embedded visual mesh: vis_verts = [4*nvis] rows (tetNr, b0, b1, b2) exactly as the reference's
 * `visVerts` (Dragon.js:1705; Softbody.js:46,263-267).  rest_normals = [3*nvis] object-space normals or NULL.
 * PARTITIONED bodies (since ABI 4) take the same, global list on every rank and keep the rows whose tet they OWN (lowest-owner rule,
 * TetSimInfo.owned_elems: exactly one partition per tet, and that partition solves the tet); TetSimInfo.num_vis_verts is the number
 * kept, tetsim_get_visual_ids their row numbers -- the union over the partitions is the whole visual mesh, each row once. */
int tetsim_set_visual_mesh(tetsim_handle h, const float *vis_verts, uint32_t nvis, const float *rest_normals);
/* row of the caller's vis_verts behind each attached visual vertex, [num_vis_verts] ascending (identity when unpartitioned) */
int tetsim_get_visual_ids(tetsim_handle h, int32_t *out);
/* Skin on the device and read back.  positions_out [3*num_vis_verts]: sum_k b_k * pos[tet corner k] with b3 = 1-b0-b1-b2,
 * evaluated like updateVisMesh (Softbody.js:259-277: f64 accumulate, f32 store per step; NEOHOOKEAN_GS + PRECISE is
 * bit-exact with it) or like the vertex shader of SoftbodyGPU.js:429-435 (POLAR_JACOBI, f32).
 * normals_out [3*num_vis_verts] (may be NULL): Rotate(rest_normal, quat[tetNr]) as SoftbodyGPU.js:440 -- POLAR_JACOBI only.
 * A PARTITION returns its own rows (tetsim_get_visual_ids order); scattered by row number, the partitions' outputs equal the
 * unpartitioned body's bit for bit (PRECISE).  Corners it does not own need their owner's END-OF-SUBSTEP position, which the
 * per-substep halo does not carry (it carries predictions): every rank calls tetsim_halo_refresh_final (RCCL; in-process groups:
 * tetsim_group_refresh_final) after the frame's last substep, THEN reads.  The read itself never communicates (since ABI 5; before,
 * RCCL bodies ran the exchange from inside it, and ranks that disagreed on whether it was due hung each other): with stale ghosts
 * it fails with TETSIM_ESTATE.  A freshly created partition is fresh (its ghost range holds the rest positions). */
int tetsim_read_visual_mesh(tetsim_handle h, float *positions_out, float *normals_out);
/* The end-of-substep positions of this partition's ghost particles, from their owners, into the ghost range behind
 * tetsim_read_positions' owned range (what tetsim_read_visual_mesh needs; valid until the next step).  RCCL bodies: a collective of
 * all ranks (one grouped send / recv per neighbour on the main stream, after a sync).  Unpartitioned: nothing to do. */
int tetsim_halo_refresh_final(tetsim_handle h);
/* The same for the partitions of an in-process group (device copies). */
int tetsim_group_refresh_final(tetsim_handle *handles, uint32_t count);
/* The visual mesh's triangle list, as the reference hands it to the geometry index (Dragon.js `dragonAttachedTriIds`;
 * Softbody.js:48-50, SoftbodyGPU.js:455-457): [3*ntri] ids of visual vertices.  Needed by tetsim_read_visual_vertex_normals. */
int tetsim_set_visual_triangles(tetsim_handle h, const int32_t *tri_ids, uint32_t ntri);
/* `visMesh.geometry.computeVertexNormals()` (Softbody.js:273, every frame -- 37 ms of the CPU path's frame, SURVEY.md §8(f)-1;
 * SoftbodyGPU.js:687 when physicsParams.computeNormals) on the device, for the positions tetsim_read_visual_mesh returns:
 * three.js r160 BufferGeometry.computeVertexNormals (indexed) + normalizeNormals -- face normal (pC-pB) x (pA-pB) in f64 from
 * the f32 positions, added to the triangle's three vertices in triangle order with an f32 store per add, then scaled by
 * 1/(length || 1) in f64 and stored f32.  Each vertex gathers its triangles in that order, so the result equals the
 * reference's bit for bit (golden recorded from three.js).  normals_out [3*nvis].  Synchronises. */
int tetsim_read_visual_vertex_normals(tetsim_handle h, float *normals_out);
/* PARTITIONED bodies (since ABI 5): tetsim_set_visual_triangles takes the same, GLOBAL triangle list on every rank (ids = rows of the
 * caller's visVerts) and keeps, for each row the partition skins, that row's triangles in triangle order.  A triangle's corners may be
 * skinned by different ranks, so the normals are computed from the ranks' skins put together: the host scatters every rank's
 * tetsim_read_visual_mesh rows by tetsim_get_visual_ids into one [3 * rows of visVerts] array (it needs that array to draw anyway) and
 * every rank calls tetsim_visual_vertex_normals_from with it -> normals_out [3*num_vis_verts], this rank's rows.  Per vertex the same
 * face normals are added in the same order as in the unpartitioned body: scattered by row, the ranks' normals equal
 * tetsim_read_visual_vertex_normals of the unpartitioned body bit for bit (Softbody.js:259-277).  tetsim_read_visual_vertex_normals
 * itself refuses partitions.  An unpartitioned body may call it too (normals of any positions of its visual mesh). */
int tetsim_visual_vertex_normals_from(tetsim_handle h, const float *all_positions, float *normals_out);
/* The partitions of one process, all at once: every member skins its rows (after tetsim_group_refresh_final), the rows are put
 * together, every member computes its rows' normals from the whole.  positions_out / normals_out [3 * rows of visVerts]; either may be
 * NULL. */
int tetsim_group_read_visual_vertex_normals(tetsim_handle *handles, uint32_t count, float *positions_out, float *normals_out);

/* --- grab (Softbody.js:279-298 ; SoftbodyGPU.js:692-712) --------------------------------------- */

/* Pin global particle `id` (-1 = none, endGrab) at xyz: consumed by the next substeps
 * (Softbody.js:233-235 ; SoftbodyGPU.js:345, with the exact particle index -- see DESIGN.md). */
int tetsim_set_grab(tetsim_handle h, int32_t id, const float xyz[3]);
/* startGrab: nearest particle to xyz among the latest positions -- argmin on the device (f64 distances evaluated as
 * Softbody.js:284-288 does, first minimum wins), only one candidate per 256 particles travels to the host; sets and
 * returns the particle (no full read-back, unlike GPUGrabber.start, SoftbodyGPU.js:790-795). */
int tetsim_start_grab(tetsim_handle h, const float xyz[3], int32_t *id_out);
/* The owned particle nearest to xyz (same f64 distance and first-minimum rule) WITHOUT grabbing it: global id and squared
 * distance.  Partitioned bodies: take the minimum over the partitions on the host (ties: lowest id), then call
 * tetsim_set_grab(id, xyz) on every partition -- only the owner pins it. */
int tetsim_nearest_particle(tetsim_handle h, const float xyz[3], int32_t *global_id, double *dist2);

/* --- measurement ----------------------------------------------------------------------------- */

/* Run n substeps eagerly on the handle's own stream; every POLAR_JACOBI kernel carries its own begin/end HIP
 * events (hipExtLaunchKernelGGL), so kernel_ms[] sums the kernels' own durations in the real launch sequence.  Bodies with
 * TetSimInfo.fused_particle_pass run the sequence of tetsim_step_n (tet | fused x (n-1) | particle): TETSIM_K_TET then holds the
 * n-1 fused kernels, TETSIM_K_VERTEX the one particle kernel that ends the call. */
int tetsim_profile(tetsim_handle h, uint32_t n, double dt, const TetSimParams *params, TetSimProfile *out);
/* Kernel-only timing: `reps` back-to-back launches of the per-tet kernel(s) of one substep inside ONE HIP-event
 * pair on the handle's stream, then the same for the per-particle kernel(s); kernel_ms[] = total / reps.  No event
 * sits between kernels, so the figure is comparable with rocprofv3's per-kernel average.  The body is left in a
 * finite but NON-PHYSICAL state (kernels are repeated out of sequence): call it on a scratch body. */
int tetsim_time_kernels(tetsim_handle h, uint32_t reps, double dt, const TetSimParams *params, TetSimProfile *out);
/* n substeps through the production path (tetsim_step_n), bracketed by HIP events on the handle's
 * stream; returns elapsed milliseconds.  Synchronises. */
int tetsim_time_step_n(tetsim_handle h, uint32_t n, double dt, const TetSimParams *params, double *ms_out);
/* Device-to-device stream copy of `bytes` on the handle's stream, `reps` times: measured copy
 * bandwidth in GB/s (read+write bytes / time) -- the "measured HBM peak" of SURVEY.md §8(d). */
int tetsim_measure_copy_bandwidth(int32_t device, uint64_t bytes, uint32_t reps, double *gbps_out);
/* The same probe by kind: 0 = copy (read + write bytes / time, what tetsim_measure_copy_bandwidth returns), 1 = read only, 2 = write
 * only (bytes / time).  Every lane keeps four independent 16-byte accesses in flight; at its first use per (device, kind, size class)
 * the probe times plain and non-temporal accesses at five grid sizes and keeps the fastest -- a yardstick has to be the best the
 * chip does, not one guess at it (MI355X_MICROARCH.md "HBM": ~6.3 TB/s achievable for a float4 copy). */
int tetsim_measure_stream_bandwidth(int32_t device, uint64_t bytes, uint32_t reps, int32_t kind, double *gbps_out);

/* --- multi-GPU halo (POLAR_JACOBI, part_count > 1) ---------------------------------------------- */

/* RCCL transport: rank 0 calls tetsim_comm_unique_id, the host distributes the 128 bytes, every rank
 * calls tetsim_comm_init.  Afterwards tetsim_step/_step_n exchange ghost positions with neighbouring
 * partitions every substep (grouped ncclSend/ncclRecv on a dedicated stream).
 * Deployment assumption: ONE PROCESS PER GPU stepping ONE partitioned body.  tetsim_step_n replays the two queues' kernel
 * chains from captured graphs that wait for each other through device words, which is only live while the handle's two
 * streams are served by independent hardware queues.  That is probed per body, again whenever this library has created
 * another stream in the process, and a timed-out wait disables replay for good (tetsim_sync) -- but streams that other code
 * creates in the same process are invisible to the probe.  TETSIM_HALO_GRAPH=0 keeps the halo path eager, which is live on
 * any queue mapping. */
int tetsim_comm_unique_id(void *id128);
int tetsim_comm_init(tetsim_handle h, const void *id128, int32_t rank, int32_t nranks);
/* What RCCL itself says about this handle's communicator (ncclCommCount / ncclCommUserRank) -- not what the caller passed to
 * tetsim_comm_init -- plus the halo this partition moves per substep: neighbours, bytes sent to and received from all of them,
 * and the largest single message.  bench.py puts it in its JSON line and refuses to report a run whose ranks != --gpus. */
typedef struct TetSimCommInfo {
    int32_t rccl_ranks, rccl_rank;
    uint32_t neighbours;
    uint64_t send_bytes_per_substep, recv_bytes_per_substep, max_message_bytes;
    int32_t loopback;            /* TETSIM_DEBUG_LOOPBACK_HALO: the halo partner is this rank itself (measurement only) */
    int32_t p2p;                 /* 1: the per-substep halo is stored straight into the neighbours' ghost ranges (tetsim_halo_p2p_connect);
                                    occupies what was tail padding: the struct's size is unchanged */
} TetSimCommInfo;
int tetsim_comm_info(tetsim_handle h, TetSimCommInfo *out);
/* Send 1 KiB to this handle's own rank and receive it back through the initialised communicator, on the halo
 * stream, and verify the bytes: exercises the run-time-resolved RCCL entry points on hosts with a single GPU. */
int tetsim_comm_selftest(tetsim_handle h);
/* Measurement helper: `reps` grouped ncclSend+ncclRecv of `bytes` to this rank itself on the halo stream, issued eagerly
 * (use_graph = 0) or captured per_graph at a time into a HIP graph and replayed (use_graph = 1); checks the bytes.
 * host_us = host time to issue one group, total_us = wall time per group.  Design input for DESIGN.md 7. */
int tetsim_comm_probe(tetsim_handle h, uint64_t bytes, uint32_t reps, int32_t use_graph, uint32_t per_graph,
                      double *host_us, double *total_us);
/* What one halo exchange costs this rank with ITS neighbours and message sizes: `reps` grouped send / recv of the current predictions
 * into the neighbours' ghost ranges (idempotent), each bracketed by events on the halo stream; min / median / max in microseconds.
 * A collective of all ranks (same reps), between steps.  The transfer term of the substep's halo chain, measured on the real wire
 * (bench.py --gpus N reports it per rank).  Since ABI 4. */
int tetsim_halo_probe(tetsim_handle h, uint32_t reps, double *min_us, double *median_us, double *max_us);
/* Peer-to-peer halo (opt-in; POLAR_JACOBI + TETSIM_FAST blocked partitions).  The per-substep ghost exchange without a transfer
 * kernel: each rank's boundary-particle kernel stores its new predictions straight into the neighbours' ghost ranges (peer memory
 * over xGMI, mapped through HIP IPC; double buffered by substep parity) and a word per neighbour says "arrived"; the halo-side
 * tiles wait for those words.  RCCL's grouped send/recv kernel (~15 us even to itself) leaves the substep's critical chain.
 *   every rank:  tetsim_halo_p2p_export(h, blob)            -> TETSIM_P2P_BLOB_BYTES bytes describing its buffers
 *   the host:    gathers the blobs of all part_count ranks (any transport: torch.distributed, MPI, files)
 *   every rank:  tetsim_halo_p2p_connect(h, blobs, part_count)   [blobs + r * TETSIM_P2P_BLOB_BYTES = rank r's]
 * Ranks of the same process connect through plain pointers and must be stepped TOGETHER with tetsim_group_step_n (after its first
 * call): their kernels wait for each other on the device, which is only live if no wait is submitted in front of the kernel that
 * satisfies it -- the group call orders the submissions, independent tetsim_step_n calls of two bodies of one process do not.  A
 * loopback body (TETSIM_DEBUG_LOOPBACK_HALO) connects to itself with its own blob (count 1).  Call between steps, and before ANY rank
 * steps again (the host synchronises the ranks around it).  On top of an RCCL communicator (connect after tetsim_comm_init) RCCL keeps
 * carrying the one refresh exchange after a dt change; without one the peer-to-peer halo is the only transport and dt must stay fixed.
 * Device-side waits are bounded by TETSIM_HALO_TIMEOUT_MS as read at this call.  Results are the RCCL transport's, bit for bit. */
#define TETSIM_P2P_BLOB_BYTES 512
int tetsim_halo_p2p_export(tetsim_handle h, void *blob);
int tetsim_halo_p2p_connect(tetsim_handle h, const void *blobs, uint32_t count);
/* tetsim_halo_probe's counterpart for the peer-to-peer halo (since ABI 5): `reps` hand-overs with all neighbours at once -- one store
 * into each neighbour's inbox word, one wait on the own ones, on the halo stream -- timed by the device's wall clock; min / median / max
 * microseconds per hand-over (one one-way signal latency of this wire; the boundary predictions ride with the same kind of stores).
 * A collective of all ranks (same reps, the same number of calls), between steps; one-layer ghost regions only. */
int tetsim_halo_p2p_probe(tetsim_handle h, uint32_t reps, double *min_us, double *median_us, double *max_us);
/* All partitions of one decomposition living in ONE process (one or several devices): n substeps with the SAME
 * stream/event choreography as the RCCL path -- interior tiles, wait for the previous halo, boundary tiles, boundary
 * particles, start the halo on a second stream, interior particles -- with asynchronous device copies standing in
 * for ncclSend/ncclRecv.  handles[i] must be partition i.  Asynchronous; reads synchronise.  The first call wires the members to
 * each other (plain pointers); tetsim_destroy of a member first drains its siblings' queues (their transfers land in its ghost ranges)
 * and the siblings forget it: they can still be read, saved and destroyed, a later tetsim_group_step_n with them is TETSIM_ESTATE
 * (create the partitions anew).  On failure tetsim_last_error(NULL) has the failing member's text. */
int tetsim_group_step_n(tetsim_handle *handles, uint32_t count, uint32_t n, double dt, const TetSimParams *params);
/* In-process transport for partitions living on one device (tests; "multi-GPU without a cluster"):
 * after every handle has been stepped ONE substep, copy owned interface positions into the
 * neighbours' ghost ranges.  handles[i] must be partition i of the same mesh. */
int tetsim_halo_exchange_local(tetsim_handle *handles, uint32_t count);
/* Host-visible halo plan of this handle (for transports implemented by the caller and for tests):
 * neighbour ranks, and per neighbour the global ids sent and received, concatenated. */
int tetsim_get_halo_plan(tetsim_handle h, int32_t *neigh /*[num_neighbours]*/,
                         int32_t *send_counts, int32_t *recv_counts,
                         int32_t *send_ids, int32_t *recv_ids);
/* Export this handle's packed send buffer for neighbour slot `n` (device->host) / import a received
 * one (host->device): the slow, transport-agnostic path used by the CPU `gloo` tests. */
int tetsim_halo_export(tetsim_handle h, uint32_t n, float *out_xyzw);
int tetsim_halo_import(tetsim_handle h, uint32_t n, const float *in_xyzw);

/* --- host-side preprocessing, callable without a GPU (unit-tested on CPU) ------------------------ */

/* Dependency levels of sequential Gauss-Seidel over `tets` in the given order: level[e] in [0,L).
 * Tets of one level are vertex-disjoint and every pair sharing a vertex keeps its relative order. */
int tetsim_prep_levels(const int32_t *tets, uint32_t nt, uint32_t nv, int32_t *level, uint32_t *num_levels);
/* Greedy colouring in tet order (smallest colour not used by any tet sharing a vertex). */
int tetsim_prep_colours(const int32_t *tets, uint32_t nt, uint32_t nv, int32_t *colour, uint32_t *num_colours);
/* The TETSIM_ORDER_CLUSTERED schedule of a mesh.  order[i] = caller's tet id at sequential position i (what
 * tetsim_get_tet_order returns for such a body); launch[i] / lane[i] / step[i] = the kernel launch (cluster colour), the lane
 * (cluster) and the lane's step that solve position i.  Correct iff any two positions a < b whose tets share a vertex have
 * launch[a] < launch[b], or the same launch AND lane with step[a] < step[b]. */
int tetsim_prep_clusters(const int32_t *tets, uint32_t nt, uint32_t nv, int32_t *order, int32_t *launch, int32_t *lane,
                         int32_t *step, uint32_t *num_launches, uint32_t *num_clusters);
/* The tile plan of the blocked polar kernels for a mesh or a batch of meshes (body_first_tet / body_first_vert
 * [bodies + 1] as tetsim_create_batch lays them out; NULL = one body): tets in tile order (tile_tets[i] = input tet at position
 * i), tile offsets into it (tile_off [tiles + 1]; pass NULL arrays to query *num_tiles first: at most nt of them), and per
 * position the tile-local particle slot of each corner (corner_slot [4 * nt], < 256).  Host only; what the property tests check:
 * <= 256 tets and <= 256 particles per tile, every tet in exactly one tile, no tile spanning two bodies, a body tiled in a batch
 * exactly as alone. */
int tetsim_prep_tiles(const float *verts, uint32_t nv, const int32_t *tets, uint32_t nt, const uint32_t *body_first_tet,
                      const uint32_t *body_first_vert, uint32_t bodies, int32_t *tile_tets, uint32_t *tile_off,
                      uint8_t *corner_slot, uint32_t *num_tiles);
/* Scatter table of SoftbodyGPU.js:563-577: slots[v*36 + s] = 4*tet + corner, -1 = empty.
 * ref_quirk != 0 reproduces the `<= 0.0` test.  Returns the number of dropped contributions. */
int tetsim_prep_slot_table(const int32_t *tets, uint32_t nt, uint32_t nv, int32_t ref_quirk,
                           int32_t *slots /*[nv*36]*/, uint32_t *dropped);
/* The particle(s) SoftbodyGPU.js:335-338,345 pins for `grab_id` on a mesh with num_elems tets (texture side
 * ceil(sqrt(num_elems))): out[0..1], -1 = none.  What TETSIM_FLAG_REF_GRAB_TEXEL uses. */
int tetsim_prep_ref_grab_texels(int32_t grab_id, uint32_t num_elems, uint32_t num_particles, int32_t out[2]);
/* Rest data of Softbody.js:60-87 in JS number semantics: invMass[nv], invRestPose[9*nt] (column-major),
 * invRestVolume[nt]. */
int tetsim_prep_rest(const float *verts, uint32_t nv, const int32_t *tets, uint32_t nt, double density,
                     float *inv_mass, float *inv_rest_pose, float *inv_rest_volume);

/* The built-in vertex partitioner for general meshes (SURVEY.md 8(e) "General meshes: host-side graph partition"; the coupling a cut
 * must respect is the particle -> incident tets table of SoftbodyGPU.js:563-577 that the Jacobi average :306-319 reads): which
 * partition owns each particle, vert_owner_out [nv], values in [0, part_count).  Recursive bisection at the weighted median
 * (weights 1 + valence: a part's tets ~ the corners it owns / 4) of the key that cuts the fewest tets -- breadth-first distances
 * from the two ends of a pseudo-diameter and their difference (topology only), plus x / y / z when `verts` is not NULL -- then
 * k-way boundary refinement (a particle moves to the part holding more of its tet-mates) which keeps every part within +-3% of the
 * mean weight of what the cuts gave it (a cut lands on a particle boundary: one particle's weight per bisection level on top, which
 * only shows on meshes of a few dozen particles).  Deterministic.  tetsim_create and tetsim_plan_create* with part_count > 1 and vert_owner == NULL use this WITHOUT
 * coordinates (verts == NULL: the plan entry points have none, and a plan must equal what tetsim_create builds); pass the
 * result of a call WITH coordinates as vert_owner to both for planar cuts on lattice-like meshes, or store it in a .tetsim
 * container (TetSimMeshArrays.vert_owner). */
int tetsim_prep_partition(const float *verts /* [3*nv] or NULL */, uint32_t nv, const int32_t *tets, uint32_t nt,
                          int32_t part_count, int32_t *vert_owner_out);
/* What a particle -> partition map costs, per partition (out [part_count]; the counts the partition plans would have, without
 * building them).  vert_owner == NULL: the map tetsim_create would pick itself.  Ghost-particle fraction of partition r =
 * ghost_particles / (owned_particles + ghost_particles) -- what crosses the wire per substep against what is integrated --,
 * ghost-tet fraction = (local_elems - owned_elems summed over r) / num_elems -- the tets solved twice. */
typedef struct TetSimPartQuality {
    uint32_t owned_particles;    /* particles this partition integrates */
    uint32_t ghost_particles;    /* particles it reads but does not own (received every substep) */
    uint32_t boundary_particles; /* owned particles some other partition reads (sent every substep) */
    uint32_t local_elems;        /* tets it solves: every tet touching an owned particle */
    uint32_t owned_elems;        /* tets counted once across partitions (lowest-owner rule) */
    uint32_t num_neighbours;
} TetSimPartQuality;
int tetsim_prep_partition_quality(const int32_t *tets, uint32_t nt, uint32_t nv, int32_t part_count,
                                  const int32_t *vert_owner, TetSimPartQuality *out);

/* Domain-decomposition plan of one partition, computed on the host exactly as tetsim_create does for
 * part_count > 1 (local numbering: owned boundary | owned interior | ghosts grouped by owner).  Lets a host
 * drive its own transport and lets the multi-process path be tested without GPUs. */
typedef struct tetsim_plan_s *tetsim_plan;
typedef struct TetSimPlanSizes {
    uint32_t owned_particles, boundary_particles, local_particles, local_elems, owned_elems, num_neighbours;
} TetSimPlanSizes;
int tetsim_plan_create(const int32_t *tets, uint32_t nt, uint32_t nv, int32_t part_count, int32_t part_index,
                       const int32_t *vert_owner, tetsim_plan *out);
/* The same plan with a ghost region `depth` layers deep (1 or 2).  Depth 2: the second layer holds the particles that share a tet
 * with a first-layer ghost, the local tets include the tets of the first-layer ghosts that touch no owned particle, and every
 * neighbour has a second pair of lists -- what lets a partition advance its first ghost layer itself and exchange ghosts only every
 * other substep (exercised on the CPU by tests/test_partition_gloo.py).  Local numbering: owned boundary | owned
 * interior | first-layer ghosts by owner | second-layer ghosts by owner. */
int tetsim_plan_create_deep(const int32_t *tets, uint32_t nt, uint32_t nv, int32_t part_count, int32_t part_index,
                            const int32_t *vert_owner, int32_t depth, tetsim_plan *out);
/* first_layer_ghosts: ghosts [owned, owned + that) are the first layer; tet_layer [local_elems]: 0 = touches an owned particle, 1 = not */
int tetsim_plan_layers(tetsim_plan p, uint32_t *first_layer_ghosts, uint8_t *tet_layer);
/* neighbour i's SECOND-layer lists (disjoint from the first-layer ones of tetsim_plan_neighbour) */
int tetsim_plan_neighbour_layer2(tetsim_plan p, uint32_t i, uint32_t *send_count, uint32_t *recv_start, uint32_t *recv_count);
int tetsim_plan_neighbour_layer2_ids(tetsim_plan p, uint32_t i, int32_t *send_local, int32_t *send_global, int32_t *recv_global);
void tetsim_plan_destroy(tetsim_plan p);
int tetsim_plan_sizes(tetsim_plan p, TetSimPlanSizes *out);
/* local_to_global_vert [local_particles], local_to_global_tet [local_elems], local_tets [4*local_elems] */
int tetsim_plan_arrays(tetsim_plan p, int32_t *local_to_global_vert, int32_t *local_to_global_tet, int32_t *local_tets);
/* neighbour i: its rank, how many owned particles are sent to it, and the contiguous local ghost range
 * [recv_start, recv_start+recv_count) its particles are received into. */
int tetsim_plan_neighbour(tetsim_plan p, uint32_t i, int32_t *rank, uint32_t *send_count, uint32_t *recv_start,
                          uint32_t *recv_count, int32_t *send_is_contiguous);
/* send_local / send_global [send_count] (ascending global id), recv_global [recv_count] */
int tetsim_plan_neighbour_ids(tetsim_plan p, uint32_t i, int32_t *send_local, int32_t *send_global, int32_t *recv_global);

int tetsim_abi_version(void);
/* What exactly is loaded.  source_sha: first 16 hex digits of the SHA-256 over the library's sources (csrc/, include/) and
 * build flags, stamped by tetsim_amd/build.py; kernel_sha: the same over the files that determine the polar tet kernel and
 * its tiling (what profiles/pmc_traffic.json is keyed by); ablation != 0: the development build (-DTETSIM_ABLATION) whose
 * tet kernel obeys TETSIM_DEBUG_ITERS / _SKIP_REST_STORE / _NO_PEEL -- never the product; debug_env: bit i set = the i-th of
 * {TETSIM_DEBUG_LOOPBACK_HALO, TETSIM_DEBUG_LOOPBACK_COPY, TETSIM_DEBUG_ONE_STREAM, TETSIM_DEBUG_GROUP_SYNC,
 * TETSIM_DEBUG_HOSTPROF, TETSIM_DEBUG_TRACE*, TETSIM_HALO_SYNC, TETSIM_HALO_GRAPH, TETSIM_DEBUG_LOOPBACK_DELAY_US, TETSIM_NH_QUADS*,
 * TETSIM_FUSED_PARTICLE_PASS, TETSIM_FRAME_KERNEL, TETSIM_FRAME_LOCAL*, TETSIM_NH_FOLD*, TETSIM_HALO_ALIGNED_TILES*, TETSIM_HALO_FOLD_WAIT, TETSIM_QUAD,
 * TETSIM_QUAD_POLL_DELAY*, TETSIM_NH_FRAME*} is set in the environment and READ by this build -- the names marked * are A/B switches
 * of choices settled by measurement, which only the development build looks at (the product library reads the 12 others and
 * TETSIM_RCCL_LIB, TETSIM_HALO_TIMEOUT_MS, TETSIM_DEBUG_STREAM_PROBE: INTEGRATION.md 4).  bench.py records all of it in its JSON line. */
typedef struct TetSimLibraryInfo {
    int32_t abi;
    int32_t ablation;
    uint32_t debug_env;
    char source_sha[20];
    char kernel_sha[20];
} TetSimLibraryInfo;
int tetsim_library_info(TetSimLibraryInfo *out);

/* --- .tetsim mesh container (GPU-free except tetsim_create_from_file) ---------------------------
 * The five arrays of the reference's Dragon.js (:1 verts, :311 tetIds, :1080 tetEdgeIds, :1705 attachedVerts
 * [tetNr,b0,b1,b2], :11640 attachedTriIds) as raw little-endian sections of one mmap-able file, plus optional
 * preprocessing: a tet colouring and a vertex->partition map.  Layout in tetsim_amd/csrc/mesh_file.cpp. */
typedef struct TetSimMeshArrays {
    uint32_t num_particles;      /* verts        [3 * num_particles] f32  (required) */
    uint32_t num_elems;          /* tets         [4 * num_elems]     i32  (required, may be empty) */
    uint32_t num_edges;          /* edge_ids     [2 * num_edges]     i32  or NULL */
    uint32_t num_vis_verts;      /* vis_verts    [4 * num_vis_verts] f32  or NULL */
    uint32_t num_vis_tris;       /* vis_tri_ids  [3 * num_vis_tris]  i32  or NULL */
    uint32_t part_count;         /* partitions addressed by vert_owner (0 when absent) */
    const float *verts;
    const int32_t *tets;
    const int32_t *edge_ids;
    const float *vis_verts;
    const int32_t *vis_tri_ids;
    const int32_t *tet_colour;   /* [num_elems] or NULL */
    const int32_t *vert_owner;   /* [num_particles], values in [0, part_count), or NULL */
} TetSimMeshArrays;
typedef struct tetsim_mesh_file *tetsim_mesh;

/* Write `a` to `path` (atomically: temp file + rename). */
int tetsim_mesh_write(const char *path, const TetSimMeshArrays *a);
/* Map a file read-only and validate it (magic, version, bounds, index ranges).  The arrays returned by
 * tetsim_mesh_arrays point INTO the mapping and stay valid until tetsim_mesh_close. */
int tetsim_mesh_open(const char *path, tetsim_mesh *out);
int tetsim_mesh_arrays(tetsim_mesh m, TetSimMeshArrays *out);
int tetsim_mesh_close(tetsim_mesh m);
/* tetsim_create from a file: vertices/tets from the mapping; a stored colouring is used when opts->tet_colour is NULL
 * and the solver/order take one; a stored partition map is used when opts->part_count > 1, opts->vert_owner is NULL
 * and the stored part_count matches; a stored visual mesh is attached (tetsim_set_visual_mesh; partitions keep their rows). */
int tetsim_create_from_file(const char *path, const TetSimOptions *opts, tetsim_handle *out);

#ifdef __cplusplus
}
#endif
#endif /* TETSIM_H */
