/* nso_selftest.c -- TEST INFRASTRUCTURE.  Drives every entry point of the CPU oracle (nso.c, included below) on a small seeded scene;
 * built with -fsanitize=address,undefined by `make -C oracle asan` and run by tests/test_oracle.py so that an out-of-bounds access,
 * a use of uninitialised stack data flagged by UBSan, a signed overflow or a misaligned access in the oracle fails a CPU test
 * ("CPU oracle under -fsanitize=address,undefined").  Prints a checksum; exit code 0 = clean. */
#include "nso.c"

#include <stdio.h>

static uint64_t lcg_state = 88172645463325252ull;
static double urand(void) { lcg_state ^= lcg_state << 13; lcg_state ^= lcg_state >> 7; lcg_state ^= lcg_state << 17; return (double)(lcg_state >> 11) / 9007199254740992.0; }
static double nrand(void) { double s = 0; for (int i = 0; i < 12; ++i) s += urand(); return s - 6.0; }

int main(void)
{
    const real bound[6] = { (real)-4.5, (real)3.82, (real)-1.5, (real)2.02, (real)-3.0, (real)2.76 };
    const int shapes[4][3] = { {3, 2, 4}, {6, 5, 7}, {9, 8, 11}, {9, 8, 11} };      /* Z,Y,X */
    nso_grid grids[4];
    real* gv[4]; real* gg[4];
    for (int l = 0; l < 4; ++l) {
        size_t n = (size_t)32 * shapes[l][0] * shapes[l][1] * shapes[l][2];
        gv[l] = (real*)malloc(n * sizeof(real)); gg[l] = (real*)calloc(n, sizeof(real));
        for (size_t i = 0; i < n; ++i) gv[l][i] = (real)(0.3 * nrand());
        grids[l].C = 32; grids[l].Z = shapes[l][0]; grids[l].Y = shapes[l][1]; grids[l].X = shapes[l][2]; grids[l].v = gv[l];
    }
    real* P[4]; real* gP[4];
    for (int w = 0; w < 4; ++w) {
        long n = nso_decoder_param_count(w);
        P[w] = (real*)malloc(n * sizeof(real)); gP[w] = (real*)calloc(n, sizeof(real));
        for (long i = 0; i < n; ++i) P[w][i] = (real)(0.25 * nrand());
        if (w) for (int i = 0; i < 3 * E_DIM; ++i) P[w][i] = (real)(25.0 * nrand());
    }
    enum { N = 70 };
    real ro[N * 3], rd[N * 3], gd[N], gc[N * 3];
    for (int n = 0; n < N; ++n) {
        ro[3 * n] = (real)(-0.3 + 0.5 * urand()); ro[3 * n + 1] = (real)(0.2 + 0.3 * urand()); ro[3 * n + 2] = (real)(0.1 * urand());
        rd[3 * n] = (real)(urand() - 0.5); rd[3 * n + 1] = (real)(0.6 * (urand() - 0.5)); rd[3 * n + 2] = (real)-1;
        gd[n] = n % 9 == 0 ? 0 : (real)(0.8 + 2.0 * urand());
        for (int k = 0; k < 3; ++k) gc[3 * n + k] = (real)urand();
    }
    nso_opts o; memset(&o, 0, sizeof(o));
    memcpy(o.bound, bound, sizeof(bound)); o.n_samples = 32; o.n_surface = 16;
    double sum = 0;
    for (int variant = 0; variant < 4; ++variant) {
        o.occupancy = variant == 1; o.perturb = variant == 2 ? (real)1 : 0; o.lindisp = 0; o.seed = 99;
        for (int stage = 0; stage < 4; ++stage) {
            for (int with_gt = 0; with_gt < 2; ++with_gt) {
                const int S = 32 + (with_gt ? 16 : 0);
                real rgb[N * 3], depth[N], var[N];
                real* w = (real*)malloc(sizeof(real) * N * S); real* z = (real*)malloc(sizeof(real) * N * S); real* raw = (real*)malloc(sizeof(real) * N * S * 4);
                if (nso_render_forward(&o, grids, (const real* const*)P, stage, N, ro, rd, with_gt ? gd : NULL, (real)-1, rgb, depth, var, w, z, raw)) return 2;
                real g_rgb[N * 3], g_d[N], g_v[N], g_ro[N * 3], g_rd[N * 3];
                real l = nso_loss_map(N, depth, rgb, gd, gc, (real)0.5, stage == 3, g_d, g_rgb);
                real l2 = nso_loss_track(N, depth, rgb, var, gd, gc, (real)0.5, 1, variant != 3, variant == 0, g_d, g_rgb, g_v);
                if (nso_render_backward(&o, grids, (const real* const*)P, stage, N, ro, rd, with_gt ? gd : NULL, (real)-1, g_rgb, g_d, variant == 0 ? NULL : g_v,
                                        (real* const*)gg, (real* const*)gP, g_ro, g_rd)) return 3;
                real frag[N];
                if (nso_ray_fragility(&o, grids, (const real* const*)P, stage, N, ro, rd, with_gt ? gd : NULL, (real)-1, frag)) return 4;
                for (int n = 0; n < N; ++n) sum += depth[n] + var[n] + g_ro[3 * n] + g_rd[3 * n + 1] + (frag[n] < 1e30 ? frag[n] : 0);
                sum += l + l2;
                free(w); free(z); free(raw);
            }
        }
    }
    /* optimiser, pose chain, filters, N1-N3 restatements */
    {
        size_t n = (size_t)32 * 9 * 8 * 11;
        real* m = (real*)calloc(n, sizeof(real)); real* v = (real*)calloc(n, sizeof(real));
        unsigned char* mask = (unsigned char*)malloc(n);
        for (size_t i = 0; i < n; ++i) mask[i] = urand() < 0.7;
        for (int step = 1; step <= 3; ++step) nso_adam_step((long)n, gv[2], gg[2], m, v, step == 2 ? NULL : mask, (real)0.005, (real)0.9, (real)0.999, (real)1e-8, step);
        for (size_t i = 0; i < n; i += 97) sum += gv[2][i];
        free(m); free(v); free(mask);
        real cam[7] = { (real)0.9, (real)0.1, (real)-0.2, (real)0.3, (real)0.5, (real)-0.4, (real)0.3 }, c2w[12], g_c2w[12], g_cam[7];
        nso_camera_from_tensor(cam, c2w);
        int pi[N], pj[N];
        nso_sample_pixels(7, N, 4, 44, 4, 60, pi, pj);
        real r_o[N * 3], r_d[N * 3];
        for (int mode = 0; mode < 4; ++mode) {
            nso_rays_from_pixels(N, pi, pj, (real)40.5, (real)40.5, (real)31.5, (real)23.5, c2w, mode, r_o, r_d);
            nso_rays_backward(N, pi, pj, (real)40.5, (real)40.5, (real)31.5, (real)23.5, mode, ro, rd, g_c2w);
            nso_camera_backward(cam, g_c2w, g_cam);
            for (int k = 0; k < 7; ++k) sum += g_cam[k];
        }
        unsigned char keep[N];
        sum += nso_inside_filter(bound, N, r_o, r_d, gd, keep);
        enum { H = 48, W = 64 };
        real* img = (real*)malloc(sizeof(real) * H * W); real* col = (real*)malloc(sizeof(real) * H * W * 3);
        for (int i = 0; i < H * W; ++i) { img[i] = i % 11 == 0 ? 0 : (real)(1.0 + 2.0 * urand()); col[3 * i] = col[3 * i + 1] = col[3 * i + 2] = (real)urand(); }
        real g_d2[N], g_c2[N * 3];
        nso_gather_pixels(N, pi, pj, W, img, col, g_d2, g_c2);
        real c2w4[16] = { 1, 0, 0, (real)-0.3, 0, 1, 0, (real)0.2, 0, 0, 1, (real)0.1, 0, 0, 0, 1 }, w2c[16];
        nso_world_to_camera(c2w4, w2c);
        for (int l = 0; l < 4; ++l) {
            size_t nv = (size_t)shapes[l][0] * shapes[l][1] * shapes[l][2];
            unsigned char* mk = (unsigned char*)malloc(nv);
            nso_frustum_mask(bound, shapes[l][0], shapes[l][1], shapes[l][2], img, H, W, (real)40, (real)40, (real)32, (real)24, c2w4, l == 0, mk);
            for (size_t i = 0; i < nv; ++i) sum += mk[i];
            free(mk);
        }
        real poses[3 * 16], pct[3];
        for (int k = 0; k < 3; ++k) { memcpy(poses + 16 * k, c2w4, sizeof(c2w4)); poses[16 * k + 3] += (real)(0.2 * k); }
        nso_keyframe_overlap(N, r_o, r_d, g_d2, 16, H, W, (real)40, (real)40, (real)32, (real)24, 3, poses, pct);
        sum += pct[0] + pct[1] + pct[2] + nso_depth_max(N, gd) + w2c[3];
        free(img); free(col);
    }
    for (int l = 0; l < 4; ++l) { free(gv[l]); free(gg[l]); }
    for (int w = 0; w < 4; ++w) { free(P[w]); free(gP[w]); }
    printf("nso_selftest ok (real = %d bytes) checksum %.6e\n", (int)sizeof(real), sum);
    return sum == sum ? 0 : 5;      /* NaN checksum = failure */
}
