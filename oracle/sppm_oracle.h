/*
 * sppm_oracle.h — C interface of the CPU oracle (liboracle.so).
 *
 * TEST INFRASTRUCTURE ONLY.  The oracle is a CPU restatement of the reference
 * renderer's hot path (wondergo2017/raytrace3, the headers under raytracer/).  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it; the
 * product library (libsppm_hip.so) never does.
 *
 * PARITY UNPINNED IN VALUE: the reference ships no tests, golden vectors or runnable
 * build (it needs OpenCV, FLANN, Eigen and MSVC), so this restatement is pinned
 * by analytic known-answer tests, by reading the source and, in structure only, by
 * the correlation of its picture of the stock scene with the one picture the
 * reference's checkout holds (a lossy JPEG of another state of the program).
 */
#ifndef SPPM_ORACLE_H
#define SPPM_ORACLE_H

#include "../include/sppm_abi.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct oracle_ctx oracle_ctx;

/* newton_fp32: run Bezier3::newton in float (the device's Newton type) instead of
 *              double (the reference's type).
 * rng_literal: 1 = one global cv::RNG-style MWC stream drawn in source order
 *              (what the reference does, serial); 0 = counter-based streams keyed
 *              by (seed, replica, path, vertex, purpose, index) — the mode the GPU
 *              path reproduces draw for draw. */
oracle_ctx* oracle_create(int newton_fp32, int rng_literal);
void oracle_destroy(oracle_ctx* ctx);
/* The radius-update knob of sppm_set_radius_update restated (initial_n = 0: the reference's dead update()). */
int oracle_set_radius_update(oracle_ctx* ctx, int32_t initial_n, uint32_t rounds_per_update);

/* A measuring switch (not the reference's arithmetic): ray directions and hit normals rounded to float before use,
 * everything else as it is.  Off by default. */
int oracle_set_narrow_rays(oracle_ctx* ctx, int on);

int oracle_set_scene(oracle_ctx* ctx, const sppm_scene* scene);
int oracle_eye_pass(oracle_ctx* ctx, const sppm_camera* cam, uint64_t seed, uint32_t replica);
/* The rows [row_begin, row_end) of GetHitPoint's outer loop (Raytracer.h:105): with counter-based streams a pixel's
 * HitPoints do not depend on the pixels before it, so strips made by several contexts on several host threads,
 * concatenated in row order and handed to oracle_set_hitpoints, are the HitPoints of oracle_eye_pass. */
int oracle_eye_pass_rows(oracle_ctx* ctx, const sppm_camera* cam, uint64_t seed, uint32_t replica,
                         int row_begin, int row_end);
/* Replace the context's HitPoints by the given records (as oracle_get_hitpoints writes them) and reset its tallies:
 * lets each of several contexts run a range of the photons (photon_begin / photon_count) over ALL HitPoints; the
 * pass is linear in tao and newN, so the contexts' resolved images and tallies add up to the serial pass. */
int oracle_set_hitpoints(oracle_ctx* ctx, const sppm_camera* cam, const sppm_hitpoint* in, uint64_t n);

#define ORACLE_GATHER_GRID 0    /* exact uniform-grid radius search */
#define ORACLE_GATHER_BRUTE 1   /* all pairs */
#define ORACLE_GATHER_KDTREE 2  /* kd-tree radius search + the per-round update() sweep (Raytracer.h:237-245) */
int oracle_photon_pass(oracle_ctx* ctx, uint64_t rounds, uint32_t per_round,
                       uint64_t photon_begin, uint64_t photon_count,
                       uint64_t seed, uint32_t replica, int gather_mode);
int oracle_resolve(oracle_ctx* ctx, double* pic_hw3, int accumulate);
int oracle_get_hitpoints(oracle_ctx* ctx, sppm_hitpoint* out, uint64_t cap, uint64_t* n);
int oracle_get_counters(oracle_ctx* ctx, sppm_counters* out);

int oracle_near_collide(oracle_ctx* ctx, uint64_t n, const double* org, const double* dir,
                        uint64_t seed, uint32_t replica, sppm_collision* out);
int oracle_newton(oracle_ctx* ctx, uint64_t n, const double* org, const double* dir,
                  const int32_t* patch, uint64_t seed, uint32_t replica, double* out_tuv);

/* Deposit records (diffuse photon hits) of the photons [photon_begin, +photon_count):
 * out is cap x 11 doubles: photon index, bounce, pos[3], n[3], flux[3]. */
int oracle_photon_deposits(oracle_ctx* ctx, uint64_t rounds, uint32_t per_round,
                           uint64_t photon_begin, uint64_t photon_count,
                           uint64_t seed, uint32_t replica,
                           double* out, uint64_t cap, uint64_t* n);

/* ---- unit-level hooks for the known-answer tests ---- */
void oracle_bezier_eval(const double* P48, double u, double v, double* out9); /* S, dS/du, dS/dv */
int oracle_aabb_intersect(const double* pmin, const double* pmax, const double* org, const double* dir);
void oracle_refracted(const double* I, const double* N, double n, double* out3);
void oracle_reflected_unit(const double* I, const double* N, double* out3);
void oracle_rotated(const double* v, const double* axis, double angle, double* out3);
void oracle_rand_refl(const double* N, double u1, double u2, double* out3);
int oracle_to_int(double x);
void oracle_tonemap(const double* pic_hw3, int w, int h, double n_images, int rgb, uint8_t* out_hw3);
void oracle_hitpoint_update(double* R2, int32_t* N, double* tao3, int32_t* newN);
void oracle_mwc_kat(int n, uint32_t* out_u32, double* out_uniform01);
uint32_t oracle_rng_u32(uint64_t seed, uint32_t stream, uint32_t path, uint32_t sub, uint32_t idx);
void oracle_texture_color_uv(const sppm_texture* tex, double u, double v, double* out3);
void oracle_camera_lookat(const double* pos, const double* look, int w, int h, sppm_camera* out);
void oracle_camera_emit(const sppm_camera* cam, int row, int col, double* org3, double* dir3);
int oracle_plane_intersect(const sppm_object* o, const double* org, const double* dir, double* dist);
int oracle_sphere_intersect(const sppm_object* o, const double* org, const double* dir, double* dist, int* inside);

#ifdef __cplusplus
}
#endif
#endif
