/* The drop-in boundary used from plain C -- no Python, no Node, no C++ on the caller's side: build a small Kuhn lattice, drop it
 * on the floor through libtetsim_hip's C ABI (include/tetsim.h) and print a checksum of the particle positions.
 *
 *   gcc -std=c99 -Iinclude examples/lattice_demo.c -Ltetsim_amd -ltetsim_hip -Wl,-rpath,$PWD/tetsim_amd -o lattice_demo
 *   ./lattice_demo [cells=8] [frames=5] [solver: polar|neohookean]
 *
 * What the reference's host loop does (main.js:74-96: numSubsteps x simulate(dt) per frame, endFrame) is ONE tetsim_step_n per frame
 * here.  Without a usable HIP device tetsim_create fails with TETSIM_ENODEVICE and this program exits 3: there is no CPU path.
 * tests/test_c_example.py builds and runs it (CPU: the failure; GPU: the checksum against the Python host's). */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "tetsim.h"

/* Kuhn split of a unit cell into six tets along the six monotone paths (0,0,0) -> (1,1,1) */
static void make_lattice(uint32_t n, double y0, float **verts, uint32_t *nv, int32_t **tets, uint32_t *nt) {
    static const int perm[6][3] = {{0, 1, 2}, {0, 2, 1}, {1, 0, 2}, {1, 2, 0}, {2, 0, 1}, {2, 1, 0}};
    const uint32_t m = n + 1;
    const double h = 1.0 / (double)n;
    uint32_t i, j, k, e = 0;
    int p, s;
    *nv = m * m * m;
    *nt = 6u * n * n * n;
    *verts = (float *)malloc(sizeof(float) * 3u * *nv);
    *tets = (int32_t *)malloc(sizeof(int32_t) * 4u * *nt);
    for (k = 0; k < m; k++)
        for (j = 0; j < m; j++)
            for (i = 0; i < m; i++) {
                float *v = *verts + 3u * (i + m * (j + m * k));
                v[0] = (float)(((double)i - (double)n / 2.0) * h);   /* (in double, then rounded: tetsim_amd/lattice.py's vertices bit for bit) */
                v[1] = (float)(y0 + (double)j * h);
                v[2] = (float)(((double)k - (double)n / 2.0) * h);
            }
    for (k = 0; k < n; k++)
        for (j = 0; j < n; j++)
            for (i = 0; i < n; i++)
                for (p = 0; p < 6; p++) {
                    uint32_t c[3];
                    int32_t *t = *tets + 4u * e++;
                    double a[3][3], det;
                    c[0] = i; c[1] = j; c[2] = k;
                    t[0] = (int32_t)(c[0] + m * (c[1] + m * c[2]));
                    for (s = 0; s < 3; s++) {
                        c[perm[p][s]]++;
                        t[s + 1] = (int32_t)(c[0] + m * (c[1] + m * c[2]));
                    }
                    for (s = 0; s < 3; s++) {   /* positive rest volume: swap two corners where the path is left-handed */
                        a[s][0] = (*verts)[3 * t[s + 1] + 0] - (*verts)[3 * t[0] + 0];
                        a[s][1] = (*verts)[3 * t[s + 1] + 1] - (*verts)[3 * t[0] + 1];
                        a[s][2] = (*verts)[3 * t[s + 1] + 2] - (*verts)[3 * t[0] + 2];
                    }
                    det = a[0][0] * (a[1][1] * a[2][2] - a[1][2] * a[2][1]) - a[0][1] * (a[1][0] * a[2][2] - a[1][2] * a[2][0]) +
                          a[0][2] * (a[1][0] * a[2][1] - a[1][1] * a[2][0]);
                    if (det < 0.0) { const int32_t x = t[2]; t[2] = t[3]; t[3] = x; }
                }
}

int main(int argc, char **argv) {
    const uint32_t cells = argc > 1 ? (uint32_t)atoi(argv[1]) : 8u, frames = argc > 2 ? (uint32_t)atoi(argv[2]) : 5u;
    const int neo = argc > 3 && strcmp(argv[3], "neohookean") == 0;
    const uint32_t substeps = 20;
    const double dt = (1.0 * (1.0 / 60.0)) / (double)substeps; /* (timeScale * timeStep) / numSubsteps, main.js:79 */
    float *verts, *pos;
    int32_t *tets;
    uint32_t nv, nt, f, i;
    TetSimOptions opt;
    TetSimParams pp;
    TetSimInfo info;
    tetsim_handle h = NULL;
    double sum = 0.0, ymin = 1e30;
    int rc;

    if (tetsim_abi_version() != TETSIM_ABI_VERSION) {
        fprintf(stderr, "libtetsim_hip has ABI %d, this program was built against %d\n", tetsim_abi_version(), TETSIM_ABI_VERSION);
        return 2;
    }
    make_lattice(cells, 0.05, &verts, &nv, &tets, &nt);
    tetsim_default_options(&opt);
    tetsim_default_params(&pp);
    opt.solver = neo ? TETSIM_SOLVER_NEOHOOKEAN_GS : TETSIM_SOLVER_POLAR_JACOBI;
    opt.precision = TETSIM_FAST;
    opt.order = TETSIM_ORDER_CLUSTERED;
    rc = tetsim_create(verts, nv, tets, nt, &opt, &h);
    if (rc != TETSIM_OK) {
        fprintf(stderr, "tetsim_create failed (%d): %s\n", rc, tetsim_last_error(NULL));
        return rc == TETSIM_ENODEVICE ? 3 : 4;
    }
    for (f = 0; f < frames; f++)
        if ((rc = tetsim_step_n(h, substeps, dt, &pp)) != TETSIM_OK) {
            fprintf(stderr, "tetsim_step_n failed (%d): %s\n", rc, tetsim_last_error(h));
            return 4;
        }
    pos = (float *)malloc(sizeof(float) * 3u * nv);
    if ((rc = tetsim_read_positions(h, pos)) != TETSIM_OK || (rc = tetsim_get_info(h, &info)) != TETSIM_OK) {
        fprintf(stderr, "read-back failed (%d): %s\n", rc, tetsim_last_error(h));
        return 4;
    }
    for (i = 0; i < nv; i++) {
        sum += (double)pos[3 * i] + (double)pos[3 * i + 1] + (double)pos[3 * i + 2];
        if (pos[3 * i + 1] < ymin) ymin = pos[3 * i + 1];
    }
    printf("%s %u tets %u particles %u frames x %u substeps: sum %.9g ymin %.6g\n", neo ? "neohookean" : "polar", info.num_elems, info.num_particles,
           frames, substeps, sum, ymin);
    tetsim_destroy(h);
    free(pos); free(verts); free(tets);
    return 0;
}
