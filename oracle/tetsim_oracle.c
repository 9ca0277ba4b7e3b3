/*
 * tetsim_oracle.c -- TEST INFRASTRUCTURE ONLY.  CPU restatement of the two reference solvers.
 *
 * This file is the parity checker for the HIP path.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load it; the product (tetsim_amd/csrc, libtetsim_hip.so) never
 * links or calls it.  Nothing here is copied from the reference: it is a from-scratch C restatement
 * of the reference's *arithmetic*, written so that every rounding step happens where the reference's
 * JavaScript (f64 arithmetic, f32 typed-array stores) or GLSL (f32) performs it.
 *
 * Section A  "nh": Neo-Hookean XPBD, sequential Gauss-Seidel -- follows
 *            /root/reference/src/Softbody.js:60-87 (initPhysics), :91-166 (solveElem),
 *            :168-193 (applyToElem), :195-240 (simulate), helpers :300-410.
 *            PINNED: bit-exact against golden vectors produced by importing Softbody.js under
 *            Node 12 (tests/golden/make_golden.mjs; tests/test_oracle_golden.py).
 *
 * Section G  "pj": shape-matching / polar-decomposition Jacobi -- follows the 7 GLSL passes
 *            /root/reference/src/SoftbodyGPU.js:59-376, the tables of :487-608 and the ping-pong
 *            semantics of /root/reference/src/MultiTargetGPUComputationRenderer.js:192-202,272-306.
 *            PINNED against the reference RUN HERE: the reference's own SoftbodyGPU.js, pass scheduler and vendored
 *            three.js execute under Node 12 with their GL calls served by Mesa's software rasteriser (softpipe,
 *            IEEE f32; oracle/glsl_ref/, tests/golden/make_golden_gpu.sh).  Against those golden vectors this
 *            restatement is bit-exact for the host tables and the first substep and tracks later substeps to
 *            4e-5 m at 200-300 substeps (sin/rsqrt/division ulps of the GLSL implementation vs glibc feeding back);
 *            tests/test_oracle_golden_glsl.py states the tolerance per horizon.
 *
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math -fopenmp (see oracle/Makefile).
 * -ffp-contract=off matters: JS never fuses a*b+c, and the f32 section is meant to be plain IEEE.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

typedef struct {
    double gravity, friction, devCompliance, volCompliance;
    double worldBounds[6]; /* lo xyz, hi xyz -- Softbody.js:215-216 via vecSetClamped :350-355 */
} OrcParams;

/* ------------------------------------------------------------------------------------------------
 * JS number semantics helpers
 * ---------------------------------------------------------------------------------------------- */
static inline float fround(double x) { return (float)x; }

/* Math.max / Math.min: NaN-propagating, and -0 < +0 (ECMA-262 21.3.2.24/25). */
static inline double js_max(double a, double b) {
    if (isnan(a) || isnan(b)) return NAN;
    if (a == 0.0 && b == 0.0) return signbit(a) ? b : a;
    return a > b ? a : b;
}
static inline double js_min(double a, double b) {
    if (isnan(a) || isnan(b)) return NAN;
    if (a == 0.0 && b == 0.0) return signbit(a) ? a : b;
    return a < b ? a : b;
}

/* ================================================================================================
 * Section A: Neo-Hookean XPBD Gauss-Seidel   (Softbody.js)
 * ============================================================================================== */
typedef struct {
    int nv, nt;
    float *pos, *prev, *vel, *invMass, *invRestPose, *invRestVolume;
    int32_t *tet;
    double volError;
    int grabId;
    float grabPos[3];
    float P[9], F[9], dF[9], g[12]; /* shared scratch, Softbody.js:27-30 */
} OrcNH;

/* vecAdd, Softbody.js:316-321: a[i] += b[i]*scale; f64 multiply, f64 add, f32 store. */
static inline void nh_vec_add(float *a, int anr, const float *b, int bnr, double scale) {
    anr *= 3; bnr *= 3;
    a[anr]     = fround((double)a[anr]     + (double)b[bnr]     * scale);
    a[anr + 1] = fround((double)a[anr + 1] + (double)b[bnr + 1] * scale);
    a[anr + 2] = fround((double)a[anr + 2] + (double)b[bnr + 2] * scale);
}
/* vecSetDiff, Softbody.js:323-328 */
static inline void nh_vec_set_diff(float *d, int dnr, const float *a, int anr, const float *b, int bnr, double scale) {
    dnr *= 3; anr *= 3; bnr *= 3;
    d[dnr]     = fround(((double)a[anr]     - (double)b[bnr])     * scale);
    d[dnr + 1] = fround(((double)a[anr + 1] - (double)b[bnr + 1]) * scale);
    d[dnr + 2] = fround(((double)a[anr + 2] - (double)b[bnr + 2]) * scale);
}
static inline void nh_vec_zero(float *a, int anr) { anr *= 3; a[anr] = a[anr + 1] = a[anr + 2] = 0.0f; }
/* vecLengthSquared, Softbody.js:330-334 */
static inline double nh_vec_len2(const float *a, int anr) {
    anr *= 3;
    double a0 = a[anr], a1 = a[anr + 1], a2 = a[anr + 2];
    return a0 * a0 + a1 * a1 + a2 * a2;
}
/* vecSetCross, Softbody.js:343-348 */
static inline void nh_vec_set_cross(float *a, int anr, const float *b, int bnr, const float *c, int cnr) {
    anr *= 3; bnr *= 3; cnr *= 3;
    double b0 = b[bnr], b1 = b[bnr + 1], b2 = b[bnr + 2];
    double c0 = c[cnr], c1 = c[cnr + 1], c2 = c[cnr + 2];
    a[anr]     = fround(b1 * c2 - b2 * c1);
    a[anr + 1] = fround(b2 * c0 - b0 * c2);
    a[anr + 2] = fround(b0 * c1 - b1 * c0);
}
/* matGetDeterminant, Softbody.js:381-387 (column-major, term order as written) */
static inline double nh_mat_det(const float *A, int anr) {
    anr *= 9;
    double a11 = A[anr + 0], a12 = A[anr + 3], a13 = A[anr + 6];
    double a21 = A[anr + 1], a22 = A[anr + 4], a23 = A[anr + 7];
    double a31 = A[anr + 2], a32 = A[anr + 5], a33 = A[anr + 8];
    return a11 * a22 * a33 + a12 * a23 * a31 + a13 * a21 * a32 - a13 * a22 * a31 - a12 * a21 * a33 - a11 * a23 * a32;
}
/* matSetInverse, Softbody.js:389-410, including the zero-determinant branch that indexes
 * without the *9 (it clears elements [e, e+9) of the whole array, not tet e's matrix). */
static void nh_mat_set_inverse(float *A, int anr, int total_floats) {
    double det = nh_mat_det(A, anr);
    if (det == 0.0) {
        for (int i = 0; i < 9; i++)
            if (anr + i < total_floats) A[anr + i] = 0.0f; /* typed-array OOB writes are dropped in JS */
        return;
    }
    double invDet = 1.0 / det;
    anr *= 9;
    double a11 = A[anr + 0], a12 = A[anr + 3], a13 = A[anr + 6];
    double a21 = A[anr + 1], a22 = A[anr + 4], a23 = A[anr + 7];
    double a31 = A[anr + 2], a32 = A[anr + 5], a33 = A[anr + 8];
    A[anr + 0] = fround((a22 * a33 - a23 * a32) * invDet);
    A[anr + 3] = fround(-(a12 * a33 - a13 * a32) * invDet);
    A[anr + 6] = fround((a12 * a23 - a13 * a22) * invDet);
    A[anr + 1] = fround(-(a21 * a33 - a23 * a31) * invDet);
    A[anr + 4] = fround((a11 * a33 - a13 * a31) * invDet);
    A[anr + 7] = fround(-(a11 * a23 - a13 * a21) * invDet);
    A[anr + 2] = fround((a21 * a32 - a22 * a31) * invDet);
    A[anr + 5] = fround(-(a11 * a32 - a12 * a31) * invDet);
    A[anr + 8] = fround((a11 * a22 - a12 * a21) * invDet);
}
/* matSetVecProduct + matSetMatProduct, Softbody.js:363-379, specialised to the only call shape the
 * solver uses: Dst = F (matrix 0), A = P (matrix 0), B = invRestPose matrix e.  Column j of F is
 * accumulated as ((0 + P0*b0) + P1*b1) + P2*b2 with an f32 store after every add. */
static inline void nh_F_from_P(float *F, const float *P, const float *ir, int e) {
    for (int j = 0; j < 3; j++) {
        int bnr = (3 * e + j) * 3;
        double b0 = ir[bnr], b1 = ir[bnr + 1], b2 = ir[bnr + 2];
        nh_vec_zero(F, j);
        nh_vec_add(F, j, P, 0, b0);
        nh_vec_add(F, j, P, 1, b1);
        nh_vec_add(F, j, P, 2, b2);
    }
}
static inline double nh_mat_ij(const float *A, int anr, int row, int col) { return A[9 * anr + 3 * col + row]; }

/* initPhysics, Softbody.js:60-87 */
static void nh_init_physics(OrcNH *s, double density) {
    for (int i = 0; i < s->nv; i++) s->invMass[i] = 0.0f;
    for (int i = 0; i < s->nt; i++) {
        int id0 = s->tet[4 * i], id1 = s->tet[4 * i + 1], id2 = s->tet[4 * i + 2], id3 = s->tet[4 * i + 3];
        nh_vec_set_diff(s->invRestPose, 3 * i, s->pos, id1, s->pos, id0, 1.0);
        nh_vec_set_diff(s->invRestPose, 3 * i + 1, s->pos, id2, s->pos, id0, 1.0);
        nh_vec_set_diff(s->invRestPose, 3 * i + 2, s->pos, id3, s->pos, id0, 1.0);
        double V = nh_mat_det(s->invRestPose, i) / 6.0;
        nh_mat_set_inverse(s->invRestPose, i, 9 * s->nt);
        double pm = V / 4.0 * density;
        s->invMass[id0] = fround((double)s->invMass[id0] + pm);
        s->invMass[id1] = fround((double)s->invMass[id1] + pm);
        s->invMass[id2] = fround((double)s->invMass[id2] + pm);
        s->invMass[id3] = fround((double)s->invMass[id3] + pm);
        s->invRestVolume[i] = fround(1.0 / V);
    }
    for (int i = 0; i < s->nv; i++)
        if (s->invMass[i] != 0.0f) s->invMass[i] = fround(1.0 / (double)s->invMass[i]);
}

/* applyToElem, Softbody.js:168-193 */
static void nh_apply_to_elem(OrcNH *s, int e, double C, double compliance, double dt) {
    if (C == 0.0) return;
    float *g = s->g;
    nh_vec_zero(g, 0);
    nh_vec_add(g, 0, g, 1, -1.0);
    nh_vec_add(g, 0, g, 2, -1.0);
    nh_vec_add(g, 0, g, 3, -1.0);
    double w = 0.0;
    for (int i = 0; i < 4; i++) {
        int id = s->tet[4 * e + i];
        w += nh_vec_len2(g, i) * (double)s->invMass[id];
    }
    if (w == 0.0) return;
    double alpha = compliance / dt / dt * (double)s->invRestVolume[e];
    double dlambda = -C / (w + alpha);
    for (int i = 0; i < 4; i++) {
        int id = s->tet[4 * e + i];
        nh_vec_add(s->pos, id, g, i, dlambda * (double)s->invMass[id]);
    }
}

/* solveElem, Softbody.js:91-166 */
static void nh_solve_elem(OrcNH *s, int e, double dt, const OrcParams *pp) {
    float *g = s->g;
    const float *ir = s->invRestPose;
    int id0 = s->tet[4 * e], id1 = s->tet[4 * e + 1], id2 = s->tet[4 * e + 2], id3 = s->tet[4 * e + 3];

    /* deviatoric: C = sqrt(tr(F^T F)) */
    nh_vec_set_diff(s->P, 0, s->pos, id1, s->pos, id0, 1.0);
    nh_vec_set_diff(s->P, 1, s->pos, id2, s->pos, id0, 1.0);
    nh_vec_set_diff(s->P, 2, s->pos, id3, s->pos, id0, 1.0);
    nh_F_from_P(s->F, s->P, ir, e);
    double r_s = sqrt(nh_vec_len2(s->F, 0) + nh_vec_len2(s->F, 1) + nh_vec_len2(s->F, 2));
    double r_s_inv = 1.0 / r_s;
    for (int k = 0; k < 3; k++) {
        nh_vec_zero(g, k + 1);
        nh_vec_add(g, k + 1, s->F, 0, r_s_inv * nh_mat_ij(ir, e, k, 0));
        nh_vec_add(g, k + 1, s->F, 1, r_s_inv * nh_mat_ij(ir, e, k, 1));
        nh_vec_add(g, k + 1, s->F, 2, r_s_inv * nh_mat_ij(ir, e, k, 2));
    }
    nh_apply_to_elem(s, e, r_s, pp->devCompliance, dt);

    /* hydrostatic: C = det F - 1 - volCompliance/devCompliance */
    nh_vec_set_diff(s->P, 0, s->pos, id1, s->pos, id0, 1.0);
    nh_vec_set_diff(s->P, 1, s->pos, id2, s->pos, id0, 1.0);
    nh_vec_set_diff(s->P, 2, s->pos, id3, s->pos, id0, 1.0);
    nh_F_from_P(s->F, s->P, ir, e);
    nh_vec_set_cross(s->dF, 0, s->F, 1, s->F, 2);
    nh_vec_set_cross(s->dF, 1, s->F, 2, s->F, 0);
    nh_vec_set_cross(s->dF, 2, s->F, 0, s->F, 1);
    for (int k = 0; k < 3; k++) {
        nh_vec_zero(g, k + 1);
        nh_vec_add(g, k + 1, s->dF, 0, nh_mat_ij(ir, e, k, 0));
        nh_vec_add(g, k + 1, s->dF, 1, nh_mat_ij(ir, e, k, 1));
        nh_vec_add(g, k + 1, s->dF, 2, nh_mat_ij(ir, e, k, 2));
    }
    double vol = nh_mat_det(s->F, 0);
    double C = vol - 1.0 - pp->volCompliance / pp->devCompliance;
    s->volError += vol - 1.0;
    nh_apply_to_elem(s, e, C, pp->volCompliance, dt);
}

OrcNH *orc_nh_create(const float *verts, int nv, const int32_t *tets, int nt, double density) {
    OrcNH *s = (OrcNH *)calloc(1, sizeof(OrcNH));
    s->nv = nv; s->nt = nt;
    s->pos = (float *)malloc(sizeof(float) * 3 * (nv + 1));
    s->prev = (float *)malloc(sizeof(float) * 3 * (nv + 1));
    s->vel = (float *)calloc(3 * (nv + 1), sizeof(float));
    s->invMass = (float *)calloc(nv + 1, sizeof(float));
    s->invRestPose = (float *)calloc(9 * (nt + 1), sizeof(float));
    s->invRestVolume = (float *)calloc(nt + 1, sizeof(float));
    s->tet = (int32_t *)malloc(sizeof(int32_t) * 4 * (nt + 1));
    memcpy(s->pos, verts, sizeof(float) * 3 * nv);
    memcpy(s->prev, verts, sizeof(float) * 3 * nv);
    memcpy(s->tet, tets, sizeof(int32_t) * 4 * nt);
    s->grabId = -1;
    nh_init_physics(s, density);
    return s;
}
void orc_nh_destroy(OrcNH *s) {
    if (!s) return;
    free(s->pos); free(s->prev); free(s->vel); free(s->invMass);
    free(s->invRestPose); free(s->invRestVolume); free(s->tet); free(s);
}

/* simulate, Softbody.js:195-240 */
void orc_nh_simulate(OrcNH *s, double dt, const OrcParams *pp) {
    for (int i = 0; i < s->nv; i++) {
        /* vecAdd(vel, i, [0, gravity, 0], 0, dt): the gravity triple is a plain JS array (f64) */
        s->vel[3 * i]     = fround((double)s->vel[3 * i]     + 0.0 * dt);
        s->vel[3 * i + 1] = fround((double)s->vel[3 * i + 1] + pp->gravity * dt);
        s->vel[3 * i + 2] = fround((double)s->vel[3 * i + 2] + 0.0 * dt);
        s->prev[3 * i] = s->pos[3 * i]; s->prev[3 * i + 1] = s->pos[3 * i + 1]; s->prev[3 * i + 2] = s->pos[3 * i + 2];
        nh_vec_add(s->pos, i, s->vel, i, dt);
    }
    s->volError = 0.0;
    for (int e = 0; e < s->nt; e++) nh_solve_elem(s, e, dt, pp);
    s->volError /= (double)s->nt;

    const double *wb = pp->worldBounds;
    for (int i = 0; i < s->nv; i++) {
        for (int c = 0; c < 3; c++)
            s->pos[3 * i + c] = fround(js_max(wb[c], js_min(wb[3 + c], (double)s->pos[3 * i + c])));
        if (s->pos[3 * i + 1] < 0.0f) {
            s->pos[3 * i + 1] = 0.0f;
            nh_vec_set_diff(s->F, 0, s->prev, i, s->pos, i, 1.0);
            double m = js_min(1.0, dt * pp->friction);
            s->pos[3 * i]     = fround((double)s->pos[3 * i]     + (double)s->F[0] * m);
            s->pos[3 * i + 2] = fround((double)s->pos[3 * i + 2] + (double)s->F[2] * m);
        }
    }
    if (s->grabId >= 0 && s->grabId < s->nv) {
        s->pos[3 * s->grabId] = s->grabPos[0];
        s->pos[3 * s->grabId + 1] = s->grabPos[1];
        s->pos[3 * s->grabId + 2] = s->grabPos[2];
    }
    /* The reference loops to pos.length (=3 Nv) with a *3 index; iterations >= Nv are out-of-range
     * typed-array accesses that JS drops (Softbody.js:238-239), so only i < Nv has an effect. */
    double inv_dt = 1.0 / dt;
    for (int i = 0; i < s->nv; i++) nh_vec_set_diff(s->vel, i, s->pos, i, s->prev, i, inv_dt);
}

void orc_nh_set_grab(OrcNH *s, int id, const float *xyz) {
    s->grabId = id;
    if (xyz) { s->grabPos[0] = xyz[0]; s->grabPos[1] = xyz[1]; s->grabPos[2] = xyz[2]; }
}
/* startGrab, Softbody.js:279-291: argmin of squared distance (f64), first minimum wins. */
int orc_nh_start_grab(OrcNH *s, double x, double y, double z) {
    double minD2 = 1.7976931348623157e308;
    int id = -1;
    for (int i = 0; i < s->nv; i++) {
        double a0 = x - (double)s->pos[3 * i], a1 = y - (double)s->pos[3 * i + 1], a2 = z - (double)s->pos[3 * i + 2];
        double d2 = a0 * a0 + a1 * a1 + a2 * a2;
        if (d2 < minD2) { minD2 = d2; id = i; }
    }
    s->grabId = id;
    s->grabPos[0] = fround(x); s->grabPos[1] = fround(y); s->grabPos[2] = fround(z);
    return id;
}
float *orc_nh_pos(OrcNH *s) { return s->pos; }
float *orc_nh_prev(OrcNH *s) { return s->prev; }
float *orc_nh_vel(OrcNH *s) { return s->vel; }
float *orc_nh_inv_mass(OrcNH *s) { return s->invMass; }
float *orc_nh_inv_rest_pose(OrcNH *s) { return s->invRestPose; }
float *orc_nh_inv_rest_volume(OrcNH *s) { return s->invRestVolume; }
double orc_nh_vol_error(OrcNH *s) { return s->volError; }

/* ================================================================================================
 * Section G: shape-matching polar-decomposition Jacobi   (SoftbodyGPU.js GLSL passes)
 * All arithmetic is IEEE f32, evaluated in the order the GLSL source writes it.
 * ============================================================================================== */
typedef struct { float x, y, z; } v3;
typedef struct { float x, y, z, w; } v4;

static inline v3 V3(float x, float y, float z) { v3 r = {x, y, z}; return r; }
static inline v3 v3add(v3 a, v3 b) { return V3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 v3sub(v3 a, v3 b) { return V3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 v3scale(v3 a, float s) { return V3(a.x * s, a.y * s, a.z * s); }
static inline float v3dot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline v3 v3cross(v3 a, v3 b) { /* GLSL ES 3.0 spec 8.5 */
    return V3(a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y);
}
static inline v4 v4normalize(v4 q) { /* x / length(x) */
    float l = sqrtf(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w);
    v4 r = {q.x / l, q.y / l, q.z / l, q.w / l};
    return r;
}
/* Rotate, SoftbodyGPU.js:111-113 */
static inline v3 pj_rotate(v3 p, v4 q) {
    v3 qv = V3(q.x, q.y, q.z);
    v3 inner = v3add(v3cross(qv, p), v3scale(p, q.w));
    return v3add(p, v3scale(v3cross(qv, inner), 2.0f));
}
/* quat_mult, SoftbodyGPU.js:114-121 */
static inline v4 pj_quat_mult(v4 q1, v4 q2) {
    v4 r;
    r.x = (q1.w * q2.x) + (q1.x * q2.w) + (q1.y * q2.z) - (q1.z * q2.y);
    r.y = (q1.w * q2.y) - (q1.x * q2.z) + (q1.y * q2.w) + (q1.z * q2.x);
    r.z = (q1.w * q2.z) + (q1.x * q2.y) - (q1.y * q2.x) + (q1.z * q2.w);
    r.w = (q1.w * q2.w) - (q1.x * q2.x) - (q1.y * q2.y) - (q1.z * q2.z);
    return r;
}
/* extractRotation, SoftbodyGPU.js:122-139 (A given as its three columns) */
static v4 pj_extract_rotation(v3 A0, v3 A1, v3 A2, v4 q, int *iters_out) {
    int iter = 0;
    for (; iter < 9; iter++) {
        v3 X = pj_rotate(V3(1.0f, 0.0f, 0.0f), q);
        v3 Y = pj_rotate(V3(0.0f, 1.0f, 0.0f), q);
        v3 Z = pj_rotate(V3(0.0f, 0.0f, 1.0f), q);
        v3 num = v3add(v3add(v3cross(X, A0), v3cross(Y, A1)), v3cross(Z, A2));
        float den = fabsf(v3dot(X, A0) + v3dot(Y, A1) + v3dot(Z, A2) + 0.000000001f);
        v3 omega = v3scale(num, 1.0f / den);
        float w = sqrtf(v3dot(omega, omega));
        if (w < 0.000000001f) break;
        /* RotationToQuaternion, :106-110 -- cos is sin(x + 1.57), deliberately not pi/2 */
        v3 axis = V3(omega.x / w, omega.y / w, omega.z / w);
        float half = w * 0.5f;
        float sx = sinf(half), sy = sinf(half + 1.57f);
        v4 dq = {axis.x * sx, axis.y * sx, axis.z * sx, sy};
        q = pj_quat_mult(dq, q);
    }
    if (iters_out) *iters_out = iter;
    return q;
}

#define PJ_SLOTS 36 /* 9 RGBA tables, SoftbodyGPU.js:29-37 */

typedef struct {
    int nv, nt;
    /* ping-pong variables: [2] buffers + current index (MultiTargetGPUComputationRenderer.js:144-162) */
    v3 *pos[2], *prev[2], *vel[2];
    v4 *elem[2][4]; /* xyz = last rotated rest vertex / goal, w = volume */
    v4 *quat[2];
    int cur_pos, cur_prev, cur_vel, cur_elem, cur_quat;
    int32_t *tet;
    float *invRestVolume; /* invRestVolumeAndColor.x */
    float *invMass;       /* built by the reference, unused by its shaders */
    int32_t *slots;       /* [nv][36] particleToElemVertsTable, -1 = empty */
    int grabId;
    float grabPos[3];
    int ref_grab_texel;   /* 1: pin the texel(s) the reference's indexFromUV selects (SoftbodyGPU.js:335-338,345) */
    int biggestT;
    long long iter_hist[10];
} OrcPJ;

/* JS determinant of the rest edge matrix, SoftbodyGPU.js:579-582,752-758 (f64 on f32-stored diffs) */
static double pj_rest_volume(const float *v, int id0, int id1, int id2, int id3) {
    float m[9];
    nh_vec_set_diff(m, 0, v, id1, v, id0, 1.0);
    nh_vec_set_diff(m, 1, v, id2, v, id0, 1.0);
    nh_vec_set_diff(m, 2, v, id3, v, id0, 1.0);
    return nh_mat_det(m, 0) / 6.0;
}

/* slot_quirk != 0 reproduces SoftbodyGPU.js:568 (`<= 0.0` treats the encoded value 0 -- tet 0,
 * vertex 0 -- as an empty slot, so it is overwritten by the particle's next incident tet).
 * Valence beyond 36 is silently dropped either way, as in the reference. */
OrcPJ *orc_pj_create(const float *verts, int nv, const int32_t *tets, int nt, double density, int slot_quirk) {
    OrcPJ *s = (OrcPJ *)calloc(1, sizeof(OrcPJ));
    s->nv = nv; s->nt = nt;
    for (int b = 0; b < 2; b++) {
        s->pos[b] = (v3 *)calloc(nv + 1, sizeof(v3));
        s->prev[b] = (v3 *)calloc(nv + 1, sizeof(v3));
        s->vel[b] = (v3 *)calloc(nv + 1, sizeof(v3));
        s->quat[b] = (v4 *)calloc(nt + 1, sizeof(v4));
        for (int k = 0; k < 4; k++) s->elem[b][k] = (v4 *)calloc(nt + 1, sizeof(v4));
    }
    s->tet = (int32_t *)malloc(sizeof(int32_t) * 4 * (nt + 1));
    memcpy(s->tet, tets, sizeof(int32_t) * 4 * nt);
    s->invRestVolume = (float *)calloc(nt + 1, sizeof(float));
    s->invMass = (float *)calloc(nv + 1, sizeof(float));
    s->slots = (int32_t *)malloc(sizeof(int32_t) * PJ_SLOTS * (nv + 1));
    for (int i = 0; i < PJ_SLOTS * nv; i++) s->slots[i] = -1;
    s->grabId = -1;

    /* both copies of every variable start equal: MultiTargetGPUComputationRenderer.js:192-202 */
    for (int b = 0; b < 2; b++)
        for (int i = 0; i < nv; i++) {
            s->pos[b][i] = V3(verts[3 * i], verts[3 * i + 1], verts[3 * i + 2]);
            s->prev[b][i] = s->pos[b][i];
        }
    for (int e = 0; e < nt; e++) {
        const int32_t *id = &tets[4 * e];
        for (int b = 0; b < 2; b++) {
            for (int k = 0; k < 4; k++) {
                v4 r = {verts[3 * id[k]], verts[3 * id[k] + 1], verts[3 * id[k] + 2], 0.0f}; /* w starts 0 */
                s->elem[b][k][e] = r;
            }
            v4 qi = {0.0f, 0.0f, 0.0f, 1.0f};
            s->quat[b][e] = qi;
        }
        for (int k = 0; k < 4; k++) { /* SoftbodyGPU.js:563-577 */
            int32_t *row = &s->slots[PJ_SLOTS * id[k]];
            for (int sl = 0; sl < PJ_SLOTS; sl++) {
                int empty = slot_quirk ? (row[sl] <= 0) : (row[sl] < 0);
                if (empty) {
                    row[sl] = 4 * e + k;
                    if (sl / 4 > s->biggestT) s->biggestT = sl / 4;
                    break;
                }
            }
        }
        double V = pj_rest_volume(verts, id[0], id[1], id[2], id[3]);
        double pm = V / 4.0 * density;
        for (int k = 0; k < 4; k++) s->invMass[id[k]] = fround((double)s->invMass[id[k]] + pm);
        s->invRestVolume[e] = fround(1.0 / V);
    }
    for (int i = 0; i < nv; i++)
        if (s->invMass[i] != 0.0f) s->invMass[i] = fround(1.0 / (double)s->invMass[i]);
    return s;
}
void orc_pj_destroy(OrcPJ *s) {
    if (!s) return;
    for (int b = 0; b < 2; b++) {
        free(s->pos[b]); free(s->prev[b]); free(s->vel[b]); free(s->quat[b]);
        for (int k = 0; k < 4; k++) free(s->elem[b][k]);
    }
    free(s->tet); free(s->invRestVolume); free(s->invMass); free(s->slots); free(s);
}

/* One substep = passes P1..P7 in insertion order (MultiTargetGPUComputationRenderer.js:272-306).
 * Each pass reads every dependency at its current index, writes the variable's other buffer, flips. */
void orc_pj_simulate(OrcPJ *s, double dt_js, const OrcParams *pp) {
    const float dt = (float)dt_js;               /* uniforms are f32 */
    const float friction = (float)pp->friction;
    const float gravity = (float)pp->gravity;
    const int nv = s->nv, nt = s->nt;

    { /* P1 copyPrevPos, SoftbodyGPU.js:59-64 */
        const v3 *pos = s->pos[s->cur_pos];
        v3 *out = s->prev[s->cur_prev ^ 1];
#pragma omp parallel for schedule(static)
        for (int i = 0; i < nv; i++) out[i] = pos[i];
        s->cur_prev ^= 1;
    }
    { /* P2 xpbdIntegrate, :67-74 */
        const v3 *pos = s->pos[s->cur_pos], *vel = s->vel[s->cur_vel];
        v3 *out = s->pos[s->cur_pos ^ 1];
#pragma omp parallel for schedule(static)
        for (int i = 0; i < nv; i++) out[i] = v3add(pos[i], v3scale(vel[i], dt));
        s->cur_pos ^= 1;
    }
    long long hist[10] = {0};
    { /* P3 solveElem, :80-182 */
        const v3 *pos = s->pos[s->cur_pos];
        v4 *const *el = s->elem[s->cur_elem];
        const v4 *qin = s->quat[s->cur_quat];
        v4 *qout = s->quat[s->cur_quat ^ 1];
#pragma omp parallel for schedule(static) reduction(+ : hist[:10])
        for (int e = 0; e < nt; e++) {
            v3 cur[4], rest[4];
            for (int k = 0; k < 4; k++) {
                cur[k] = pos[s->tet[4 * e + k]];
                rest[k] = V3(el[k][e].x, el[k][e].y, el[k][e].z);
            }
            v3 cc = v3scale(v3add(v3add(v3add(cur[0], cur[1]), cur[2]), cur[3]), 0.25f);
            v3 rc = v3scale(v3add(v3add(v3add(rest[0], rest[1]), rest[2]), rest[3]), 0.25f);
            for (int k = 0; k < 4; k++) { cur[k] = v3sub(cur[k], cc); rest[k] = v3sub(rest[k], rc); }
            /* TransposeMult(lastRest, current), :90-105: column a of A = sum_k rest_k[a] * cur_k */
            v3 A0 = V3(0, 0, 0), A1 = V3(0, 0, 0), A2 = V3(0, 0, 0);
            for (int k = 0; k < 4; k++) {
                v3 l = rest[k], r = cur[k];
                A0.x += l.x * r.x; A1.x += l.y * r.x; A2.x += l.z * r.x;
                A0.y += l.x * r.y; A1.y += l.y * r.y; A2.y += l.z * r.y;
                A0.z += l.x * r.z; A1.z += l.y * r.z; A2.z += l.z * r.z;
            }
            v4 ident = {0.0f, 0.0f, 0.0f, 1.0f};
            int it;
            v4 rot = pj_extract_rotation(A0, A1, A2, ident, &it);
            hist[it]++;
            qout[e] = v4normalize(pj_quat_mult(rot, qin[e]));
        }
        s->cur_quat ^= 1;
    }
    for (int i = 0; i < 10; i++) s->iter_hist[i] += hist[i];
    { /* P4 gatherElem, :188-263 */
        const v3 *pos = s->pos[s->cur_pos];
        v4 *const *el = s->elem[s->cur_elem];
        v4 *const *eo = s->elem[s->cur_elem ^ 1];
        const v4 *qnew = s->quat[s->cur_quat], *qold = s->quat[s->cur_quat ^ 1]; /* prev_textureQuat */
#pragma omp parallel for schedule(static)
        for (int e = 0; e < nt; e++) {
            float invVolume = 1.0f / s->invRestVolume[e];
            v3 cur[4], rest[4];
            for (int k = 0; k < 4; k++) {
                cur[k] = pos[s->tet[4 * e + k]];
                rest[k] = V3(el[k][e].x, el[k][e].y, el[k][e].z);
            }
            v4 qo = qold[e];
            v4 conj = {-qo.x, -qo.y, -qo.z, qo.w};
            v4 rel = v4normalize(pj_quat_mult(qnew[e], v4normalize(conj)));
            v3 cc = v3scale(v3add(v3add(v3add(cur[0], cur[1]), cur[2]), cur[3]), 0.25f);
            v3 rc = v3scale(v3add(v3add(v3add(rest[0], rest[1]), rest[2]), rest[3]), 0.25f);
            for (int k = 0; k < 4; k++) {
                v3 g = v3add(pj_rotate(v3sub(rest[k], rc), rel), cc);
                v4 o = {g.x, g.y, g.z, invVolume};
                eo[k][e] = o;
            }
        }
        s->cur_elem ^= 1;
    }
    { /* P5 applyElem, :272-320: volume-weighted average over the slots, in slot order */
        v4 *const *el = s->elem[s->cur_elem];
        v3 *out = s->pos[s->cur_pos ^ 1];
#pragma omp parallel for schedule(static)
        for (int i = 0; i < nv; i++) {
            v3 sum = V3(0, 0, 0);
            float wsum = 0.0f;
            const int32_t *row = &s->slots[PJ_SLOTS * i];
            for (int sl = 0; sl < PJ_SLOTS; sl++) {
                if (!(row[sl] > -1)) break;
                v4 ev = el[row[sl] % 4][row[sl] / 4];
                sum = v3add(sum, v3scale(V3(ev.x, ev.y, ev.z), ev.w));
                wsum += ev.w;
            }
            out[i] = V3(sum.x / wsum, sum.y / wsum, sum.z / wsum); /* 0/0 = NaN for slot-less particles */
        }
        s->cur_pos ^= 1;
    }
    { /* P6 collision, :326-355.  The reference's indexFromUV (:335-338, "This isn't quite correct") maps a
       * texel to int(uv.x*(R-1)) + int(uv.y*(R-1)*R), which is not the particle's index, so a different texel (or
       * several, or none) is pinned.  Default: pin exactly particle grabId (what the library does, documented
       * divergence).  ref_grab_texel: the reference's mapping, f32 operation by operation, R = texDim (:13). */
        const v3 *pos = s->pos[s->cur_pos], *prev = s->prev[s->cur_prev];
        v3 *out = s->pos[s->cur_pos ^ 1];
        const float fr = fminf(1.0f, dt * friction);
        const int R = (int)ceil(sqrt((double)nt));
        const float Rf = (float)R, Rm1 = Rf - 1.0f;
#pragma omp parallel for schedule(static)
        for (int i = 0; i < nv; i++) {
            v3 p = pos[i];
            int grabbed = (i == s->grabId);
            if (s->ref_grab_texel) {
                const float ux = ((float)(i % R) + 0.5f) / Rf, uy = ((float)(i / R) + 0.5f) / Rf; /* gl_FragCoord.xy / resolution.xy */
                const int idx = (int)(ux * Rm1) + (int)((uy * Rm1) * Rf);
                grabbed = ((float)idx == (float)s->grabId);
            }
            if (grabbed) p = V3(s->grabPos[0], s->grabPos[1], s->grabPos[2]);
            p.x = fminf(fmaxf(p.x, -2.5f), 2.5f);
            p.y = fminf(fmaxf(p.y, -1.0f), 10.0f);
            p.z = fminf(fmaxf(p.z, -2.5f), 2.5f);
            if (p.y < 0.0f) {
                p.y = 0.0f;
                v3 F = v3sub(prev[i], p);
                p.x += F.x * fr;
                p.z += F.z * fr;
            }
            out[i] = p;
        }
        s->cur_pos ^= 1;
    }
    { /* P7 xpbdVelocity, :364-372 */
        const v3 *pos = s->pos[s->cur_pos], *prev = s->prev[s->cur_prev];
        v3 *out = s->vel[s->cur_vel ^ 1];
        const v3 gdt = v3scale(V3(0.0f, gravity, 0.0f), dt);
#pragma omp parallel for schedule(static)
        for (int i = 0; i < nv; i++) {
            v3 d = v3sub(pos[i], prev[i]);
            out[i] = v3add(V3(d.x / dt, d.y / dt, d.z / dt), gdt);
        }
        s->cur_vel ^= 1;
    }
}

void orc_pj_set_ref_grab_texel(OrcPJ *s, int on) { s->ref_grab_texel = on; }
void orc_pj_set_grab(OrcPJ *s, int id, const float *xyz) {
    s->grabId = id;
    if (xyz) { s->grabPos[0] = xyz[0]; s->grabPos[1] = xyz[1]; s->grabPos[2] = xyz[2]; }
}
/* Overwrite position and velocity of the listed particles (partitioned tests: ghost particles take the
 * owner's state before the next substep). */
void orc_pj_write_particles(OrcPJ *s, int n, const int32_t *idx, const float *pos3, const float *vel3) {
    for (int i = 0; i < n; i++) {
        s->pos[s->cur_pos][idx[i]] = V3(pos3[3 * i], pos3[3 * i + 1], pos3[3 * i + 2]);
        s->vel[s->cur_vel][idx[i]] = V3(vel3[3 * i], vel3[3 * i + 1], vel3[3 * i + 2]);
    }
}
/* Overwrite the per-tet state (quaternion xyzw, the 4 carried rest corners as xyzw each) of the listed tets (partitioned tests with a
 * two-layer ghost region: second-layer ghost tets are restored / evolved after the fact, tests/test_partition_gloo.py). */
void orc_pj_write_tets(OrcPJ *s, int n, const int32_t *idx, const float *quat4, const float *elem16) {
    for (int i = 0; i < n; i++) {
        const int e = idx[i];
        v4 q = {quat4[4 * i], quat4[4 * i + 1], quat4[4 * i + 2], quat4[4 * i + 3]};
        s->quat[s->cur_quat][e] = q;
        for (int k = 0; k < 4; k++) {
            v4 c = {elem16[16 * i + 4 * k], elem16[16 * i + 4 * k + 1], elem16[16 * i + 4 * k + 2], elem16[16 * i + 4 * k + 3]};
            s->elem[s->cur_elem][k][e] = c;
        }
    }
}
void orc_pj_read_pos(OrcPJ *s, float *out) { memcpy(out, s->pos[s->cur_pos], sizeof(v3) * s->nv); }
void orc_pj_read_prev(OrcPJ *s, float *out) { memcpy(out, s->prev[s->cur_prev], sizeof(v3) * s->nv); }
void orc_pj_read_vel(OrcPJ *s, float *out) { memcpy(out, s->vel[s->cur_vel], sizeof(v3) * s->nv); }
void orc_pj_read_quat(OrcPJ *s, float *out) { memcpy(out, s->quat[s->cur_quat], sizeof(v4) * s->nt); }
void orc_pj_read_elem(OrcPJ *s, int k, float *out) { memcpy(out, s->elem[s->cur_elem][k], sizeof(v4) * s->nt); }
const int32_t *orc_pj_slots(OrcPJ *s) { return s->slots; }
const float *orc_pj_inv_rest_volume(OrcPJ *s) { return s->invRestVolume; }
const float *orc_pj_inv_mass(OrcPJ *s) { return s->invMass; }
int orc_pj_biggest_table(OrcPJ *s) { return s->biggestT; }
void orc_pj_iter_hist(OrcPJ *s, long long *out10) { memcpy(out10, s->iter_hist, sizeof(s->iter_hist)); }

int orc_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
void orc_set_threads(int n) {
#ifdef _OPENMP
    omp_set_num_threads(n);
#else
    (void)n;
#endif
}
