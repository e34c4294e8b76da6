// sppm_device.h — device-side building blocks of the SPPM hot path (gfx950, wave64).
//
// Precision split: ray state, analytic plane / sphere hits, AABB tests, shading
// directions and gather positions are fp64 (the reference's type, so the 1e-4
// self-intersection guards of Obj.h:69-76,117-128 behave as they do on the CPU);
// the Bezier Newton solve — the dominant ALU cost — and all colours, flux and
// accumulators are fp32.
//
// Random numbers are counter-based: a draw is murmur3_32 of
// (seed_lo, seed_hi, stream, path, (vertex << 8) | purpose, idx) >> 8, times 2^-24,
// so any lane can evaluate any draw in any order (this is what lets one wavefront
// spread the 50 Newton restarts of one ray over its lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace sppm {

// ---- constants of the reference: Vec3.h:6-14, Raytracer.h:11-13, Bezier.h:6
#define SPPM_EPS 1e-4
#define SPPM_PI 3.1415926
#define SPPM_PI_INV 0.3183098862
#define SPPM_MAX_NUM 1e20
#define SPPM_MAX_DEP 13
#define SPPM_INIT_R2 2.0
#define SPPM_ALPHA 0.7          // Raytracer.h:45
#define SPPM_NEWTON_ITERS 10
#define SPPM_NEWTON_RESTARTS 50

enum { P_EMIT = 0, P_NEWTON = 1, P_ROULETTE = 2, P_DIFFUSE = 3 };
enum { D_EYE = 0, D_PHOTON = 1 };
enum { OBJ_PLANE = 0, OBJ_SPHERE = 1, OBJ_BEZIER = 2 };
enum { F_DIFF = 1, F_REFL = 2, F_REFR = 4 };
enum { A_PLANE_X = 3, A_PLANE_Y = 4, A_PLANE_Z = 5 };     // DevObject::akind beyond the object types

struct DevObject {
    double p0[3];
    double n[3];
    double radius;
    double refrn;
    double bmin[3];
    double bmax[3];
    double diffr, reflr, refrr, allr;   // Obj.h:11-16
    float color[3];
    float diff[3];
    float refl[3];
    float refr[3];
    int type, texture, first_patch, n_patches;
    int flags;
    int tex_rows, tex_cols;     // the object's texture, copied here so that a lookup starts from LDS (texture < 0: unused)
    unsigned tex_off;           // first texel in the packed uchar4 array
    int lobe;                   // F_DIFF / F_REFL / F_REFR if exactly one of diffr, reflr, refrr is positive (Obj::Roulette then has one outcome), else 0
    int akind;                  // what the plane / sphere loop runs for this object: its type, or A_PLANE_X + k for a plane whose normal is +-1 along axis k
    int tex_udex, tex_vdex;     // plane: the coordinates its texture is laid along (Obj.h:86-100)
    int pad0;
    double inv_radius;          // sphere: 1 / radius
    double inv_refrn;           // 1 / refrn (Raytracer.h:187, :332)
};

struct TexDesc {
    int rows, cols;
    unsigned offset;   // first texel in the packed uchar4 array
    int pad;
};

struct DevLight {
    double pos[3];
    float flux[3];     // color * PI * 4, Light.h:10
    float pad;
};

// Scene tables staged in LDS by every workgroup: objects | patch AABBs | control points.
// a pointer into LDS, typed as one: a pointer that may be LDS or global memory costs flat loads
#define SPPM_LDS(T) const __attribute__((address_space(3))) T
struct AObj;
struct SceneView {
    const DevObject* objs;      // LDS
    const double* patch_box;    // LDS, n_patches x 6 (pmin, pmax)
    const float* patch_cp;      // LDS, n_patches x 48
    // Newton tables in GLOBAL memory (read at wave-uniform addresses): per patch the 48
    // power-basis coefficients of S(u,v) - centre, and the centre (see newton_coop)
    const float* pw_q;
    const double* pw_c;
    const int* patch_obj;       // LDS, n_patches: owning object of each patch
    const struct AObj* aobj;    // LDS, n_objs: what the plane / sphere tests read of each object (analytic_nearest)
    // Per patch the boxes of its 4 x 4 sub-patches (global memory, n_patches x 16 x {lo xyz, hi xyz}, fp32 rounded
    // outwards), each grown by SPPM_SUBBOX_MARGIN; null: no such table (see subbox_prune)
    const float* sub_box;
    int n_objs;
    int n_patches;
    const uchar4* texels;       // global
    // The Bezier objects by index, one byte each, so that the loops over them do not read the type of every object from LDS
    // in every turn of the trace loop; n_bez < 0: not listed (more than eight, or an index beyond 255), the loops scan the objects
    int n_bez;
    uint64_t bez_idx;
    // The objects in index order as runs of one DevObject::akind, a byte each (kind | length << 4, length <= 15), for the
    // plane / sphere loop; n_runs < 0: more than sixteen runs, the loop reads every object's kind from its record
    int n_runs;
    uint64_t runs_lo, runs_hi;
};

// ------------------------------------------------------------------ RNG
__device__ __forceinline__ uint32_t rotl32(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }
__device__ __forceinline__ uint32_t mm3(uint32_t h, uint32_t k) {
    k *= 0xcc9e2d51u; k = rotl32(k, 15); k *= 0x1b873593u;
    h ^= k; h = rotl32(h, 13); h = h * 5u + 0xe6546b64u;
    return h;
}
__device__ __forceinline__ uint32_t fmix3(uint32_t h) {
    h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16;
    return h;
}
// hash state after (seed_lo, seed_hi, stream)
__device__ __forceinline__ uint32_t rng_base(uint64_t seed, uint32_t stream) {
    uint32_t h = 0x9747b28cu;
    h = mm3(h, (uint32_t)seed);
    h = mm3(h, (uint32_t)(seed >> 32));
    return mm3(h, stream);
}
// hash state after (.., path, (vertex << 8) | purpose)
__device__ __forceinline__ uint32_t rng_sub(uint32_t base, uint32_t path, uint32_t vertex, uint32_t purpose) {
    return mm3(mm3(base, path), (vertex << 8) | purpose);
}
// the same in two steps, so that the draws of one path vertex share the hash of the path: rng_sub(base, path, v, p) ==
// rng_vertex(rng_path(base, path), v, p)
__device__ __forceinline__ uint32_t rng_path(uint32_t base, uint32_t path) { return mm3(base, path); }
__device__ __forceinline__ uint32_t rng_vertex(uint32_t path_hash, uint32_t vertex, uint32_t purpose) { return mm3(path_hash, (vertex << 8) | purpose); }
__device__ __forceinline__ uint32_t rng_bits(uint32_t sub, uint32_t idx) { return fmix3(mm3(sub, idx) ^ 24u) >> 8; }
__device__ __forceinline__ double rng_u01(uint32_t sub, uint32_t idx) { return (double)rng_bits(sub, idx) * (1.0 / 16777216.0); }
__device__ __forceinline__ float rng_u01f(uint32_t sub, uint32_t idx) { return (float)rng_bits(sub, idx) * (1.0f / 16777216.0f); }

// ------------------------------------------------------------------ wave helpers
__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63); }
__device__ __forceinline__ uint64_t lanemask_lt() { return (1ull << lane_id()) - 1ull; }
__device__ __forceinline__ int bcast_i(int v, int src) { return __builtin_amdgcn_readlane(v, src); }
__device__ __forceinline__ uint32_t bcast_u(uint32_t v, int src) { return (uint32_t)__builtin_amdgcn_readlane((int)v, src); }
__device__ __forceinline__ float bcast_f(float v, int src) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src)); }
__device__ __forceinline__ double bcast_d(double v, int src) {
    int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
    int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ uint64_t bcast_u64(uint64_t v, int src) {
    uint32_t lo = bcast_u((uint32_t)v, src), hi = bcast_u((uint32_t)(v >> 32), src);
    return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ float wave_sum_f(float v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ uint32_t wave_sum_u(uint32_t v) {
    for (int o = 32; o > 0; o >>= 1) v += (uint32_t)__shfl_xor((int)v, o, 64);
    return v;
}

// ------------------------------------------------------------------ Vec3.h in fp64
struct d3 { double x, y, z; };
__device__ __forceinline__ d3 mk(double x, double y, double z) { d3 r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ d3 operator+(d3 a, d3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ d3 operator-(d3 a, d3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ d3 operator*(d3 a, double s) { return mk(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ d3 neg(d3 a) { return mk(-a.x, -a.y, -a.z); }
__device__ __forceinline__ double dot(d3 a, d3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ d3 cross(d3 a, d3 b) { return mk(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
__device__ __forceinline__ d3 ld3(const double* p) { return mk(p[0], p[1], p[2]); }
__device__ __forceinline__ double comp(d3 a, int i) { return i == 0 ? a.x : (i == 1 ? a.y : a.z); }
__device__ __forceinline__ bool is_zero(double v) { return fabs(v) < SPPM_EPS; }           // Vec3.h:23
__device__ __forceinline__ d3 get_unit(d3 v) {                                             // Vec3.h:56 (and Normalize, :48-55)
    double mo = sqrt(dot(v, v));
    return is_zero(mo) ? v : v * (1.0 / mo);
}
// 1 / sqrt(s) and sqrt(s) to the last place or two from the hardware's reciprocal square root (26 bits) and two Newton
// steps: a third of the instructions of the correctly rounded sqrt followed by a division.  The reference's results (sqrt
// and 1 / mo, each rounded) are themselves that far from the exact value, so this is rounding-level, like the FMA
// contraction of the other expressions.
#ifndef SPPM_FAST_UNIT
#define SPPM_FAST_UNIT 1
#endif
__device__ __forceinline__ double rsqrt_nr(double s) {
    double y = __builtin_amdgcn_rsq(s);
    y = y * fma(-0.5 * s, y * y, 1.5);
    y = y * fma(-0.5 * s, y * y, 1.5);
    return y;
}
// 1 / x to the last place or two: the hardware's reciprocal and two Newton steps
__device__ __forceinline__ double rcp_nr(double x) {
    double y = __builtin_amdgcn_rcp(x);
    y = fma(y, fma(-x, y, 1.0), y);
    y = fma(y, fma(-x, y, 1.0), y);
    return y;
}
#ifndef SPPM_FAST_DIV
#define SPPM_FAST_DIV 1
#endif
__device__ __forceinline__ double sqrt_nr(double s) {
#if SPPM_FAST_UNIT
    return s > 0.0 ? s * rsqrt_nr(s) : 0.0;
#else
    return sqrt(s);
#endif
}
// Vec3::GetUnit (Vec3.h:56) for a vector of any length: |v| < 1e-4 (i.e. |v|^2 < 1e-8) is returned as it is, as there.
__device__ __forceinline__ d3 get_unit_any(d3 v) {
#if SPPM_FAST_UNIT
    const double s = dot(v, v);
    return (s < 1e-8) ? v : v * rsqrt_nr(s);
#else
    return get_unit(v);
#endif
}
// Vec3::GetUnit for a vector that is of unit length but for rounding — the reflection, refraction or rotation of unit
// vectors, a sphere's radius vector divided by the radius: v / |v| = v (1.5 - 0.5 |v|^2) up to (3/8)(|v|^2 - 1)^2, which
// is below 4e-19 inside the guard; any other input takes the general path.
__device__ __forceinline__ d3 get_unit_near1(d3 v) {
#if SPPM_FAST_UNIT
    const double s = dot(v, v);
    if (fabs(s - 1.0) < 1e-9) return v * fma(-0.5, s, 1.5);
    return get_unit_any(v);
#else
    return get_unit(v);
#endif
}
__device__ __forceinline__ d3 get_refl(d3 v, d3 n) { return v - n * (2 * dot(v, n)); }     // Vec3.h:80-83
__device__ __forceinline__ d3 get_anormal(d3 v) {                                          // Vec3.h:85-89
    return (v.x == 0 && v.y == 0) ? mk(1, 0, 0) : get_unit_any(mk(v.y, -v.x, 0));
}
__device__ __forceinline__ d3 rotated(d3 v, d3 a, double angle) {                          // Vec3.h:99-115
    if (fabs(angle) < SPPM_EPS) return v;
    double cost = cos(angle), sint = sin(angle);
    d3 r;
    r.x = v.x * (a.x * a.x + (1 - a.x * a.x) * cost) + v.y * (a.x * a.y * (1 - cost) - a.z * sint) +
          v.z * (a.x * a.z * (1 - cost) + a.y * sint);
    r.y = v.x * (a.y * a.x * (1 - cost) + a.z * sint) + v.y * (a.y * a.y + (1 - a.y * a.y) * cost) +
          v.z * (a.y * a.z * (1 - cost) - a.x * sint);
    r.z = v.x * (a.z * a.x * (1 - cost) - a.y * sint) + v.y * (a.z * a.y * (1 - cost) + a.x * sint) +
          v.z * (a.z * a.z + (1 - a.z * a.z) * cost);
    return r;
}
// Vec3::GetUnitRandRefl, Vec3.h:90-98: rotate n about a = GetAnormal(n) by theta = acos(sqrt(u1)), the
// result about n by phi = 2 PI u2, normalise.  With a perpendicular to n and both of unit length the two
// Rodrigues rotations (Vec3::rotated, Vec3.h:99-115) collapse to
//     n cos(theta) + sin(theta) (b cos(phi) + a sin(phi)),   b = a x n,
// and cos(theta) = sqrt(u1), sin(theta) = sqrt(1 - u1): the same direction as the reference's matrix
// form up to fp64 rounding, for ~1/4 of its operations and without acos.  Vec3::rotated returns its input
// when |angle| < 1e-4: for theta that needs u1 > 1 - 1e-8, which a 24-bit uniform never reaches; for phi
// the early return is kept (cos = 1, sin = 0).
__device__ __forceinline__ d3 rand_refl(d3 n, double u1, double u2) {
    const double cost = sqrt_nr(u1), sint = sqrt_nr(1.0 - u1);
    const double phi = 2 * SPPM_PI * u2;
    double sp = 0.0, cp = 1.0;
    if (!(fabs(phi) < SPPM_EPS)) sincos(phi, &sp, &cp);
    const d3 a = get_anormal(n);
    const d3 b = cross(a, n);
    return get_unit_near1(n * cost + (b * cp + a * sp) * sint);
}
// sin and cos of phi = 2 PI k / 2^24 for the 24-bit draw k behind a uniform u2 = k / 2^24 (PI: the reference's constant,
// SPPM_PI): the top eight bits pick (sin, cos) of 2 PI j / 256 from a table of 256 pairs, correct to the last place (made on
// the host in extended precision, copied to LDS by the photon kernel), the low sixteen an angle x < 2 PI / 256 whose Taylor series stop at x^7 / x^6 (the
// next terms are below 4e-18), and the two meet by the addition theorems.  About 25 instructions and one LDS read where the
// library's sincos, with its argument reduction for any double, takes 156; the results differ from sincos(fl(2 PI u2)) by
// a few 1e-16, which is what rounding phi to a double already moves them by.
#define SPPM_SINCOS_TAB 256
__device__ __forceinline__ void sincos_2pi_u24(SPPM_LDS(double)* tab, uint32_t k, double& s, double& c) {
    const uint32_t hi = k >> 16, lo = k & 0xffffu;
    const double sh = tab[2 * hi], ch = tab[2 * hi + 1];
    const double x = (double)lo * (2.0 * SPPM_PI / 16777216.0), x2 = x * x;
    const double sl = x * fma(x2, fma(x2, fma(x2, -1.0 / 5040.0, 1.0 / 120.0), -1.0 / 6.0), 1.0);
    const double cl = fma(x2, fma(x2, fma(x2, -1.0 / 720.0, 1.0 / 24.0), -0.5), 1.0);
    s = fma(sh, cl, ch * sl);
    c = fma(ch, cl, -(sh * sl));
}
// rand_refl with phi's sine and cosine from the table: k2 is the draw behind u2.
__device__ __forceinline__ d3 rand_refl_tab(d3 n, double u1, uint32_t k2, SPPM_LDS(double)* tab) {
    const double cost = sqrt_nr(u1), sint = sqrt_nr(1.0 - u1);
    const double phi = 2 * SPPM_PI * ((double)k2 * (1.0 / 16777216.0));
    double sp, cp;
    sincos_2pi_u24(tab, k2, sp, cp);
    if (fabs(phi) < SPPM_EPS) { sp = 0.0; cp = 1.0; }      // Vec3::rotated returns its input (Vec3.h:101)
    const d3 a = get_anormal(n);
    const d3 b = cross(a, n);
    return get_unit_near1(n * cost + (b * cp + a * sp) * sint);
}
__device__ __forceinline__ d3 refracted(d3 I, d3 N, double n) {                            // Vec3.h:124-134
    double cosI = -dot(N, I);
    double cosR2 = 1 - (1 - cosI * cosI) * (n * n);
    return (cosR2 > SPPM_EPS) ? I * n + N * (n * cosI - sqrt_nr(cosR2)) : get_refl(I, N);
}

__device__ __forceinline__ TexDesc tex_of(const DevObject& ob) {
    TexDesc t; t.rows = ob.tex_rows; t.cols = ob.tex_cols; t.offset = ob.tex_off; t.pad = 0;
    return t;
}

// ------------------------------------------------------------------ Element.h:61-72 Texture::colorUV
// The lookup in two halves, so that a caller can put independent work between asking for the four texels and
// blending them: tex_fetch does the index arithmetic and issues the loads, tex_blend the bilinear weights.
struct TexTap {
    uchar4 a, b, c, d;      // texels (r1,c1) (r1,c2) (r2,c1) (r2,c2)
    double detR, detC;
    int on;                 // 0: no texture on this hit, the object's own colour stands
};
__device__ __forceinline__ void tex_fetch(const TexDesc t, const uchar4* __restrict__ texels, double u, double v, TexTap& tap) {
    int rows = t.rows, cols = t.cols;
    double row = (u - floor(u)) * rows, col = (v - floor(v)) * cols;
    int r1 = (int)floor(row + 1e-10), r2 = r1 + 1;
    int c1 = (int)floor(col + 1e-10), c2 = c1 + 1;
    tap.detR = r2 - row; tap.detC = c2 - col;
    r1 = (r1 >= 0) ? (r1 >= rows ? 0 : r1) : (rows - 1);
    c1 = (c1 >= 0) ? (c1 >= cols ? 0 : c1) : (cols - 1);
    r2 = (r2 < rows) ? r2 : 0;
    c2 = (c2 < cols) ? c2 : 0;
    r2 = r2 < 0 ? 0 : r2;  // memory-safety guard for NaN coordinates (the reference would index out of bounds)
    c2 = c2 < 0 ? 0 : c2;
    const uchar4* base = texels + t.offset;
    tap.a = base[(size_t)r1 * cols + c1]; tap.b = base[(size_t)r1 * cols + c2];
    tap.c = base[(size_t)r2 * cols + c1]; tap.d = base[(size_t)r2 * cols + c2];
    tap.on = 1;
}
__device__ __forceinline__ void tex_blend(const TexTap& tap, float out[3]) {
    const double detR = tap.detR, detC = tap.detC;
    double wa = detR * detC, wb = detR * (1 - detC), wc = (1 - detR) * detC, wd = (1 - detR) * (1 - detC);
    const double k = 1.0 / 255.0;
    out[0] = (float)((tap.a.x * wa + tap.b.x * wb + tap.c.x * wc + tap.d.x * wd) * k);
    out[1] = (float)((tap.a.y * wa + tap.b.y * wb + tap.c.y * wc + tap.d.y * wd) * k);
    out[2] = (float)((tap.a.z * wa + tap.b.z * wb + tap.c.z * wc + tap.d.z * wd) * k);
}
__device__ __forceinline__ void color_uv(const TexDesc t, const uchar4* __restrict__ texels, double u, double v, float out[3]) {
    TexTap tap;
    tex_fetch(t, texels, u, v, tap);
    tex_blend(tap, out);
}

// ------------------------------------------------------------------ Bezier.h:20-51 AABBbox::intersect
// Entry faces only, t > 0: an origin inside the box reports a miss.
__device__ __forceinline__ bool aabb_hit(const double* __restrict__ box, d3 o, d3 d) {
    d3 pmin = ld3(box), pmax = ld3(box + 3);
#pragma unroll
    for (int i = 0; i < 3; i++) {
        double di = comp(d, i);
        if (!is_zero(di)) {
            double t = ((di > 0 ? comp(pmin, i) : comp(pmax, i)) - comp(o, i)) / di;   // (a reciprocal here costs 12 more registers: measured slower)
            if (t > 0) {
                d3 c = o + d * t;
                bool ok = true;
#pragma unroll
                for (int j = 0; j < 3; j++)
                    if (j != i)
                        if (comp(pmax, j) < comp(c, j) || comp(pmin, j) > comp(c, j)) ok = false;
                if (ok) return true;
            }
        }
    }
    return false;
}

// Lower bound of the ray parameter of every ray point that lies within `margin` of the box (slab test on
// the box grown by margin; 0 if the ray starts inside it).  A Newton candidate of a patch is a ray point
// whose distance to S(u,v), u,v in [0,1], is below sqrt(1e-4) = 0.01 (Bezier.h:135), and S(u,v) lies in the
// convex hull of the control points, hence in the patch box: its t is >= this bound for margin > 0.01.
// A patch whose bound exceeds a hit already found can therefore be skipped without changing the result of
// Bezier3Obj::GetIntersect / Scene::NearCollide (both keep the nearest hit).
// The bound only orders and prunes, it decides no hit: it is computed in fp32 (coordinates and ray parameters of a
// few hundred units: errors below 1e-4), and the slack of the comparison absorbs that.
#define SPPM_PRUNE_MARGIN 0.011f
#define SPPM_PRUNE_SLACK 2e-3f
__device__ __forceinline__ float aabb_entry_lb(const double* __restrict__ box, d3 o, d3 d) {
    float t0 = 0.0f;
#pragma unroll
    for (int i = 0; i < 3; i++) {
        const float di = (float)comp(d, i);
        if (fabsf(di) > 1e-12f) {
            const float inv = __frcp_rn(di);
            const float lo = (float)(box[i] - comp(o, i)) - SPPM_PRUNE_MARGIN, hi = (float)(box[3 + i] - comp(o, i)) + SPPM_PRUNE_MARGIN;
            t0 = fmaxf(t0, fminf(lo * inv, hi * inv));
        }
    }
    return t0;
}

// AABBbox::intersect answered in fp32 wherever fp32 can be sure of the answer.  With no direction component that IsZero,
// the reference's test says yes exactly when T = max_i entry_i is positive and T <= X = min_i exit_i (entry_i, exit_i: the
// ray parameters at which the ray crosses into and out of slab i): the face it enters through is the one of the largest
// entry parameter, and its hit point lies within the other two slabs exactly when their entry parameters are not larger
// and their exit parameters not smaller.  Each parameter is computed here with a relative error below 1e-6 (two
// conversions, a reciprocal, a product) and bracketed by that much and by 1e-5 absolute (the reference's own o + t * d
// carries 1e-16 of the coordinates, divided by a direction component of at least 1e-4): `yes` and a false `open` are
// certain, and only a ray for which `open` comes back true needs the fp64 test.  With a component that IsZero the reference
// does not enter through that axis' faces but still holds the hit point to its slab, so it says yes for a subset of the
// rays the slab test passes: then `yes` is never claimed and `open` is the slab test (a ray exactly parallel gives
// +-inf or the NaN of 0 * inf, which fminf / fmaxf drop: no constraint).  lb: the lower bound of aabb_entry_lb, the
// entry parameters moved out by the margin.
__device__ __forceinline__ void aabb_classify(const double* __restrict__ box, d3 o, d3 d, bool& yes, bool& open, float& lb) {
    const float W = 1e-6f, A = 1e-5f;
    float tl = -3.0e38f, th = -3.0e38f, xl = 3.0e38f, xh = 3.0e38f, g = 0.0f, dmin = 3.0e38f;
#pragma unroll
    for (int i = 0; i < 3; i++) {
        const float df = (float)comp(d, i);
        const float inv = __builtin_amdgcn_rcpf(df);
        dmin = fminf(dmin, fabsf(df));
        const float a = (float)(box[i] - comp(o, i)) * inv, b = (float)(box[3 + i] - comp(o, i)) * inv;
        const float en = fminf(a, b), ex = fmaxf(a, b);
        const float enl = fmaf(-fabsf(en), W, en);
        tl = fmaxf(tl, enl);
        th = fmaxf(th, fmaf(fabsf(en), W, en));
        xl = fminf(xl, fmaf(-fabsf(ex), W, ex));
        xh = fminf(xh, fmaf(fabsf(ex), W, ex));
        g = fmaxf(g, fmaf(-fabsf(inv), SPPM_PRUNE_MARGIN * 1.001f, enl));
    }
    const bool no = (th + A <= 0.0f) || (tl - A > xh + A);
    yes = (dmin >= 1.001e-4f) && (tl - A > 0.0f) && (th + A <= xl - A);
    open = !no && !yes;
    lb = g;
}
// The fp64 test for the rays aabb_classify leaves open: rare, so the ray goes through an opaque copy, which keeps the
// compiler from computing the test's direction predicates ahead of the branch, on every turn of the loops around it.
__device__ __forceinline__ bool aabb_hit_cold(const double* __restrict__ box, d3 o, d3 d) {
    asm volatile("" : "+v"(d.x), "+v"(d.y), "+v"(d.z));
    return aabb_hit(box, o, d);
}

// ------------------------------------------------------------------ bicubic Bernstein evaluation (Bezier.h:69-111), fp32
// S(u,v) = sum_i sum_k P[4i+k] b_k(u) b_i(v): u runs along the fast index.
struct f3 { float x, y, z; };
__device__ __forceinline__ void bern(float t, float b[4], float d[4]) {
    float m = 1.0f - t;
    b[0] = m * m * m; b[1] = 3.0f * t * m * m; b[2] = 3.0f * t * t * m; b[3] = t * t * t;
    d[0] = -3.0f * m * m; d[1] = 3.0f * m * m - 6.0f * t * m; d[2] = 6.0f * t * m - 3.0f * t * t; d[3] = 3.0f * t * t;
}
// P: 48 floats (wave-uniform values held in registers).  Returns S, dS/du, dS/dv.
__device__ __forceinline__ void patch_eval(const float* P, float u, float v, f3& S, f3& Su, f3& Sv) {
    float bu[4], du[4], bv[4], dv[4];
    bern(u, bu, du);
    bern(v, bv, dv);
    S = {0, 0, 0}; Su = {0, 0, 0}; Sv = {0, 0, 0};
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const float* r = P + 12 * i;
        float ax = r[0] * bu[0] + r[3] * bu[1] + r[6] * bu[2] + r[9] * bu[3];
        float ay = r[1] * bu[0] + r[4] * bu[1] + r[7] * bu[2] + r[10] * bu[3];
        float az = r[2] * bu[0] + r[5] * bu[1] + r[8] * bu[2] + r[11] * bu[3];
        float dx = r[0] * du[0] + r[3] * du[1] + r[6] * du[2] + r[9] * du[3];
        float dy = r[1] * du[0] + r[4] * du[1] + r[7] * du[2] + r[10] * du[3];
        float dz = r[2] * du[0] + r[5] * du[1] + r[8] * du[2] + r[11] * du[3];
        S.x += ax * bv[i]; S.y += ay * bv[i]; S.z += az * bv[i];
        Sv.x += ax * dv[i]; Sv.y += ay * dv[i]; Sv.z += az * dv[i];
        Su.x += dx * bv[i]; Su.y += dy * bv[i]; Su.z += dz * bv[i];
    }
}
__device__ __forceinline__ f3 patch_point(const float* P, float u, float v) {
    float bu[4], du[4], bv[4], dv[4];
    bern(u, bu, du);
    bern(v, bv, dv);
    f3 S = {0, 0, 0};
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const float* r = P + 12 * i;
        S.x += (r[0] * bu[0] + r[3] * bu[1] + r[6] * bu[2] + r[9] * bu[3]) * bv[i];
        S.y += (r[1] * bu[0] + r[4] * bu[1] + r[7] * bu[2] + r[10] * bu[3]) * bv[i];
        S.z += (r[2] * bu[0] + r[5] * bu[1] + r[8] * bu[2] + r[11] * bu[3]) * bv[i];
    }
    return S;
}

// The patch in the form the Newton solver evaluates: S(u,v) - c = sum_j sum_i Q[j][i] u^i v^j, the bicubic
// of Bezier.h:69-111 expanded from the Bernstein to the power basis around the patch centre c (done once per
// scene on the host, in fp64).  Coordinates relative to the centre are a few units instead of ~100, so the fp32
// residual org + dir t - S(u,v) is formed from small numbers.  Q: 48 wave-uniform values.
// Every polynomial in Horner form, the derivatives as q1 + 2u (q2 + 1.5u q3): 3 + 2 fused operations per (row, coordinate)
// and 3 + 3 + 2 per coordinate, 96 in all for S, dS/du, dS/dv (~160 in Bernstein form).
__device__ __forceinline__ void patch_eval_pw(const float* Q, float u, float v, f3& S, f3& Su, f3& Sv) {
    const float ua = 1.5f * u, ub = 2.0f * u, va = 1.5f * v, vb = 2.0f * v;
    float A[4][3], D[4][3];
#pragma unroll
    for (int j = 0; j < 4; j++)
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const float q0 = Q[(j * 4 + 0) * 3 + c], q1 = Q[(j * 4 + 1) * 3 + c], q2 = Q[(j * 4 + 2) * 3 + c], q3 = Q[(j * 4 + 3) * 3 + c];
            A[j][c] = fmaf(fmaf(fmaf(q3, u, q2), u, q1), u, q0);
            D[j][c] = fmaf(fmaf(q3, ua, q2), ub, q1);
        }
    float s[3], su[3], sv[3];
#pragma unroll
    for (int c = 0; c < 3; c++) {
        s[c] = fmaf(fmaf(fmaf(A[3][c], v, A[2][c]), v, A[1][c]), v, A[0][c]);
        su[c] = fmaf(fmaf(fmaf(D[3][c], v, D[2][c]), v, D[1][c]), v, D[0][c]);
        sv[c] = fmaf(fmaf(A[3][c], va, A[2][c]), vb, A[1][c]);
    }
    S = {s[0], s[1], s[2]}; Su = {su[0], su[1], su[2]}; Sv = {sv[0], sv[1], sv[2]};
}
__device__ __forceinline__ f3 patch_point_pw(const float* Q, float u, float v) {
    float A[4][3];
#pragma unroll
    for (int j = 0; j < 4; j++)
#pragma unroll
        for (int c = 0; c < 3; c++)
            A[j][c] = fmaf(fmaf(fmaf(Q[(j * 4 + 3) * 3 + c], u, Q[(j * 4 + 2) * 3 + c]), u, Q[(j * 4 + 1) * 3 + c]), u, Q[(j * 4 + 0) * 3 + c]);
    f3 S;
    S.x = fmaf(fmaf(fmaf(A[3][0], v, A[2][0]), v, A[1][0]), v, A[0][0]);
    S.y = fmaf(fmaf(fmaf(A[3][1], v, A[2][1]), v, A[1][1]), v, A[0][1]);
    S.z = fmaf(fmaf(fmaf(A[3][2], v, A[2][2]), v, A[1][2]), v, A[0][2]);
    return S;
}

// One Newton restart, Bezier.h:117-140: 10 steps of x -= J^-1 (org + dir t - S(u,v)),
// J = [dir | -Su | -Sv], from (0, u0, v0); every iterate with residual^2 < 1e-4 and
// u, v in [0,1] is a candidate; returns this restart's first minimum-t candidate.
// Q: the patch in power form around its centre; o: ray origin relative to that centre.
__device__ __forceinline__ void newton_restart(const float* Q, f3 o, f3 d, float u0, float v0, float& best_t, float& best_u, float& best_v) {
    float t = 0.0f, u = u0, v = v0;
    best_t = (float)SPPM_MAX_NUM; best_u = 0; best_v = 0;
#pragma unroll 1
    for (int it = 0; it < SPPM_NEWTON_ITERS; it++) {
        f3 S, Su, Sv;
        patch_eval_pw(Q, u, v, S, Su, Sv);
        float rx = o.x + d.x * t - S.x, ry = o.y + d.y * t - S.y, rz = o.z + d.z * t - S.z;
        if (it > 0) {  // candidate test of the iterate produced by the previous step (Bezier.h:133-139)
            float res2 = rx * rx + ry * ry + rz * rz;
            if (res2 < (float)SPPM_EPS && u <= 1.0f && u >= 0.0f && v >= 0.0f && v <= 1.0f && t < best_t) {
                best_t = t; best_u = u; best_v = v;
            }
        }
        // columns a = d, b = -Su, c = -Sv; inverse rows = (b x c, c x a, a x b) / det; det == 0 -> zero matrix
        // (the other two rows as triple products of r x d, one cross product fewer, were measured slower)
        float c0x = Su.y * Sv.z - Su.z * Sv.y, c0y = Su.z * Sv.x - Su.x * Sv.z, c0z = Su.x * Sv.y - Su.y * Sv.x;  // b x c
        float c1x = d.y * Sv.z - d.z * Sv.y, c1y = d.z * Sv.x - d.x * Sv.z, c1z = d.x * Sv.y - d.y * Sv.x;        // c x a = d x Sv
        float c2x = Su.y * d.z - Su.z * d.y, c2y = Su.z * d.x - Su.x * d.z, c2z = Su.x * d.y - Su.y * d.x;        // a x b = Su x d
        float det = d.x * c0x + d.y * c0y + d.z * c0z;
        float inv = (det == 0.0f) ? 0.0f : __builtin_amdgcn_rcpf(det);   // the hardware's reciprocal (1 ulp): a Newton step does not need the division's last bit
        t -= (c0x * rx + c0y * ry + c0z * rz) * inv;
        u -= (c1x * rx + c1y * ry + c1z * rz) * inv;
        v -= (c2x * rx + c2y * ry + c2z * rz) * inv;
    }
    f3 S = patch_point_pw(Q, u, v);
    float rx = o.x + d.x * t - S.x, ry = o.y + d.y * t - S.y, rz = o.z + d.z * t - S.z;
    float res2 = rx * rx + ry * ry + rz * rz;
    if (res2 < (float)SPPM_EPS && u <= 1.0f && u >= 0.0f && v >= 0.0f && v <= 1.0f && t < best_t) {
        best_t = t; best_u = u; best_v = v;
    }
}

// Bezier3::newton (Bezier.h:112-159) for ONE (ray, patch), all 64 lanes cooperating:
// lane r < 50 runs restart r; the result is the minimum-t candidate over all
// restarts, ties to the lowest restart (the reference keeps the first pushed
// minimum).  so, sd (the ray, fp64), sub (the ray's NEWTON hash state) and patch are wave-uniform.
// Returns a wave-uniform (t,u,v); t = 1e20 when nothing converged.
// pw_q / pw_c are GLOBAL memory, read at a wave-uniform address: in k_newton the compiler turns the reads into scalar
// loads and the 48 coefficients live in SGPRs; in k_trace, which stores to global memory between solves, it reads them
// with vector loads of one address (a broadcast) into 48 VGPRs.  Forcing scalar loads there (constant address space)
// was measured slower: the scalar registers they take are spilled elsewhere in the loop.
__device__ __forceinline__ void newton_coop(const float* __restrict__ pw_q, const double* __restrict__ pw_c, int patch, d3 so, d3 sd,
                                            uint32_t sub, float& rt, float& ru, float& rv) {
    const int lane = lane_id();
    patch = __builtin_amdgcn_readfirstlane(patch);
    float Q[48];
    const float* src = pw_q + patch * 48;
#pragma unroll
    for (int k = 0; k < 48; k++) Q[k] = src[k];
    const double* c = pw_c + patch * 3;
    const f3 o = {(float)(so.x - c[0]), (float)(so.y - c[1]), (float)(so.z - c[2])};   // origin relative to the patch centre, rounded once
    const f3 d = {(float)sd.x, (float)sd.y, (float)sd.z};
    float bt = (float)SPPM_MAX_NUM, bu = 0, bv = 0;
    if (lane < SPPM_NEWTON_RESTARTS) {
        float u0 = rng_u01f(sub, (uint32_t)patch * 128u + (uint32_t)lane * 2u + 0u);
        float v0 = rng_u01f(sub, (uint32_t)patch * 128u + (uint32_t)lane * 2u + 1u);
        newton_restart(Q, o, d, u0, v0, bt, bu, bv);
    }
    // min over lanes of (t, lane)
    int who = lane;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        float ot = __shfl_xor(bt, off, 64);
        int ow = __shfl_xor(who, off, 64);
        if (ot < bt || (ot == bt && ow < who)) { bt = ot; who = ow; }
    }
    who = bcast_i(who, 0);
    rt = bcast_f(bt, 0);
    ru = bcast_f(bu, who);
    rv = bcast_f(bv, who);
}

// Minimum over the 16 lanes of each DPP row, left in every lane of the row.
__device__ __forceinline__ float row_min_f(float v) {
    v = fminf(v, __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), 0xB1, 0xf, 0xf, false)));    // quad_perm [1,0,3,2]
    v = fminf(v, __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), 0x4E, 0xf, 0xf, false)));    // quad_perm [2,3,0,1]
    v = fminf(v, __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), 0x141, 0xf, 0xf, false)));   // row_half_mirror
    v = fminf(v, __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), 0x140, 0xf, 0xf, false)));   // row_mirror
    return v;
}

// Which of the patches whose box a ray enters can hold a Newton candidate at all?  A candidate (Bezier.h:133-139) is a
// ray point within sqrt(1e-4) = 0.01 of S(u,v), (u,v) in [0,1]^2.  S(u,v) lies in the convex hull of the control points
// of the one of the patch's 4 x 4 sub-patches that (u,v) falls into, hence in that sub-patch's box, and the ray point in
// the box grown by 0.01.  A patch none of whose 16 grown sub-boxes the forward ray (t >= 0) meets therefore has no
// candidate with t >= 0: its solve would return nothing, or a negative t, and either way the patch gives no hit
// (Bezier.h:251-259) — it is taken off the list without being solved.  Three in five solves at the reference's scene are
// of this kind (the box of a curved patch is mostly empty).  For the patches that stay, the smallest entry distance into
// a sub-box met is a lower bound of every candidate's t, tighter than the patch box's, for the nearest-first order.
// Lane p holds patch (chunk base + p): ph = its box is entered, lbb = bits of its lower bound.  Sixteen lanes test the
// sixteen sub-boxes of one patch, four patches per round.  The slab test runs in fp32 on differences taken in fp64; every
// slab distance is moved outwards by more than its rounding error (conversion of the difference and of the direction,
// reciprocal to 1 ulp, product: under 4e-7 relative), and the boxes are stored with margin to spare.
#define SPPM_SUBBOX_MARGIN 0.012
__device__ __forceinline__ void subbox_prune(const float* __restrict__ sub_box, int chunk_first_patch, d3 so, d3 sd, bool& ph, uint32_t& lbb) {
    const int lane = lane_id();
    const int row = lane >> 4, sidx = lane & 15;
    // +-inf for a ray parallel to an axis: a slab the origin is outside of then gives an empty interval, one it is
    // inside of gives (-inf, inf), and the NaN of 0 * inf or inf - inf is dropped by fminf / fmaxf (no constraint)
    const float ix = __builtin_amdgcn_rcpf((float)sd.x), iy = __builtin_amdgcn_rcpf((float)sd.y), iz = __builtin_amdgcn_rcpf((float)sd.z);
    const float W = 1e-6f;
    uint64_t rem = __ballot(ph);
    while (rem) {
        int b0 = -1, b1 = -1, b2 = -1, b3 = -1;
        b0 = __ffsll((unsigned long long)rem) - 1; rem &= rem - 1;
        if (rem) { b1 = __ffsll((unsigned long long)rem) - 1; rem &= rem - 1; }
        if (rem) { b2 = __ffsll((unsigned long long)rem) - 1; rem &= rem - 1; }
        if (rem) { b3 = __ffsll((unsigned long long)rem) - 1; rem &= rem - 1; }
        const int mine = row == 0 ? b0 : (row == 1 ? b1 : (row == 2 ? b2 : b3));   // the patch this row of 16 lanes looks at
        bool hit = false;
        float entry = 3.0e38f;
        if (mine >= 0) {
            const float2* bx = (const float2*)(sub_box + ((size_t)(chunk_first_patch + mine) * 16 + (size_t)sidx) * 6);
            const float2 q0 = bx[0], q1 = bx[1], q2 = bx[2];      // lo.x lo.y | lo.z hi.x | hi.y hi.z
            float t0 = 0.0f, t1 = 3.0e38f;
            {
                const float a = (float)((double)q0.x - so.x) * ix, b = (float)((double)q1.y - so.x) * ix;
                const float lo = fminf(a, b), hi = fmaxf(a, b);
                t0 = fmaxf(t0, fmaf(-fabsf(lo), W, lo)); t1 = fminf(t1, fmaf(fabsf(hi), W, hi));
            }
            {
                const float a = (float)((double)q0.y - so.y) * iy, b = (float)((double)q2.x - so.y) * iy;
                const float lo = fminf(a, b), hi = fmaxf(a, b);
                t0 = fmaxf(t0, fmaf(-fabsf(lo), W, lo)); t1 = fminf(t1, fmaf(fabsf(hi), W, hi));
            }
            {
                const float a = (float)((double)q1.x - so.z) * iz, b = (float)((double)q2.y - so.z) * iz;
                const float lo = fminf(a, b), hi = fmaxf(a, b);
                t0 = fmaxf(t0, fmaf(-fabsf(lo), W, lo)); t1 = fminf(t1, fmaf(fabsf(hi), W, hi));
            }
            hit = t0 <= t1;
            if (hit) entry = t0;
        }
        const uint64_t hm = __ballot(hit);
        const float mn = row_min_f(entry);
        const int bs[4] = {b0, b1, b2, b3};
#pragma unroll
        for (int k = 0; k < 4; k++) {
            if (bs[k] < 0) break;
            const bool any = ((hm >> (16 * k)) & 0xffffull) != 0;
            const uint32_t lb = __float_as_uint(fmaxf(bcast_f(mn, 16 * k), 0.0f));
            if (lane == bs[k]) {
                if (!any) { ph = false; lbb = 0xffffffffu; }
                else if (lb > lbb) lbb = lb;                   // both are bit patterns of non-negative floats
            }
        }
    }
}

// ------------------------------------------------------------------ Scene::NearCollide (Scene.h:165-182)
// Bezier3Obj::GetIntersect (Bezier.h:240-282) for the lanes of one wave.  `need` marks
// lanes whose ray passed the object AABB; `bound` is the distance of the nearest hit the lane's ray has
// so far (other objects).  For each such ray in turn the wave tests the patch AABBs one per lane, then
// solves the hit patches cooperatively, nearest box first, and stops as soon as no remaining box can hold
// a hit nearer than the best one (aabb_entry_lb).  On return the ray's lane holds (t,u,v,patch) of the
// nearest accepted patch or patch = -1.
// path_hash / vertex: the ray's random stream (rng_path of its pixel or photon, its path vertex); the hash state of its
// Newton draws is only formed for the one ray being solved, on the scalar unit.
__device__ __forceinline__ void bezier_coop(const SceneView& S, const DevObject& ob, bool need, d3 o, d3 d, uint32_t path_hash, uint32_t vertex, float bound,
                                            float& out_t, float& out_u, float& out_v, int& out_patch, uint32_t& newton_calls) {
    const int lane = lane_id();
    out_patch = -1; out_t = (float)SPPM_MAX_NUM; out_u = 0; out_v = 0;
    const int n_patches = __builtin_amdgcn_readfirstlane(ob.n_patches), first_patch = __builtin_amdgcn_readfirstlane(ob.first_patch);
    uint64_t m = __ballot(need);
    while (m) {
        int src = __ffsll((unsigned long long)m) - 1;
        m &= m - 1;
        d3 so = mk(bcast_d(o.x, src), bcast_d(o.y, src), bcast_d(o.z, src));
        d3 sd = mk(bcast_d(d.x, src), bcast_d(d.y, src), bcast_d(d.z, src));
        const uint32_t ssub = rng_vertex(bcast_u(path_hash, src), bcast_u(vertex, src), P_NEWTON);
        const float sbound = bcast_f(bound, src);
        float pt = (float)SPPM_MAX_NUM, pu = (float)SPPM_MAX_NUM, pv = (float)SPPM_MAX_NUM;  // para = MAX_VEC3
        int pidx = -1;
        for (int base = 0; base < n_patches; base += 64) {
            int p = first_patch + base + lane;
            float lbf = 0.0f;
            bool ph, ph_open;
            aabb_classify(S.patch_box + 6 * p, so, sd, ph, ph_open, lbf);                      // Bezier3::intersect, Bezier.h:176-186
            ph = ph && (base + lane < n_patches);
            ph_open = ph_open && (base + lane < n_patches);
            if (__ballot(ph_open)) ph = ph || (ph_open && aabb_hit_cold(S.patch_box + 6 * p, so, sd));
            // lb >= 0, so the bit patterns order like the values and the selection below runs on scalars
            uint32_t lbb = ph ? __float_as_uint(lbf) : 0xffffffffu;
            if (S.sub_box && __ballot(ph)) subbox_prune(S.sub_box, first_patch + base, so, sd, ph, lbb);
            uint64_t pm = __ballot(ph);
            while (pm) {
                int sel = 0;
                uint32_t selb = 0xffffffffu;
                for (uint64_t q = pm; q; q &= q - 1) {
                    const int bit = __ffsll((unsigned long long)q) - 1;
                    const uint32_t v = bcast_u(lbb, bit);
                    if (v < selb) { selb = v; sel = bit; }
                }
                if (__uint_as_float(selb) - SPPM_PRUNE_SLACK > fminf(pt, sbound)) break;   // nor can any box after it in this order
                pm &= ~(1ull << sel);
                const int pi = first_patch + base + sel;
                float t, u, v;
                newton_coop(S.pw_q, S.pw_c, pi, so, sd, ssub, t, u, v);
                newton_calls += 1u;            // wave-uniform tally
                const float e = (float)SPPM_EPS;
                // Bezier.h:251-259 (a patch whose minimum-t candidate is negative is dropped as a whole); the
                // reference walks the patches in index order and keeps the first minimum
                if (t < e || t > pt || u < -e || u > 1 + e || v < -e || v > 1 + e) continue;
                if (t < pt || (t == pt && pi < pidx)) { pt = t; pu = u; pv = v; pidx = pi; }
            }
        }
        if (lane == src) { out_t = pt; out_u = pu; out_v = pv; out_patch = pidx; }
    }
}

// Running nearest-hit record of Scene::NearCollide (Scene.h:165-182).  The reference walks
// objvec in order and replaces on strictly smaller dist, i.e. the winner is the minimum of
// (dist, object index); consider() applies that rule so analytic and Bezier candidates can be
// merged in any order.
struct Near {
    double dist;
    int obj, inside, patch;
    float t, u, v;
};
__device__ __forceinline__ void near_init(Near& n) {
    n.dist = SPPM_MAX_NUM; n.obj = -1; n.inside = 0; n.patch = -1; n.t = 0; n.u = 0; n.v = 0;
}
__device__ __forceinline__ void near_consider(Near& n, double dist, int obj, int inside, int patch, float t, float u, float v) {
    if (dist < n.dist || (dist == n.dist && obj < n.obj)) {
        n.dist = dist; n.obj = obj; n.inside = inside; n.patch = patch; n.t = t; n.u = u; n.v = v;
    }
}

// Planes and spheres (Obj.h:65-85, :111-139), per lane, fp64.
//
// Shaped for a wave that is alone (or nearly) on its SIMD, which is how the trace kernels run: such a wave pays about 24
// cycles for a taken branch, 12 for one not taken, as much again for every compare -> exec-mask hand-over and 60 and more
// for an LDS round trip, against 4 to 5 for a vector instruction.  So (1) the bodies are straight-line code whose guards
// (Obj.h:68-76, :117-128) are conjoined into ONE select: what a guard protects against cannot reach the record, since a
// quotient by a projection that IsZero comes out finite, infinite or NaN and is deselected by its own guard bit, and the
// square root is taken of max(det2, tiny) and deselected where det2 < EPS; (2) the objects are walked in index order
// (the order decides ties, Scene.h:171-178) as RUNS of one kind: the kind and length of each run come packed in scalar
// registers (a kernel argument), the dispatch happens once per run with no lane value behind it, and a run is a tight loop
// around one body; (3) the loop reads a table of 32-byte records (AObj) that holds just what the tests read, the next
// object's record asked for before the current one is worked on.
struct __attribute__((aligned(16))) AObj { double a, b, c, r; };   // axis plane k: a = P0[k]; sphere: centre (a, b, c), radius r; else unused
#define SPPM_MAX_RUNS 16
__device__ __forceinline__ void analytic_nearest(const SceneView& S, bool active, d3 o, d3 d, Near& nr) {
#ifdef ABLATE_ANALYTIC
    if (active) near_consider(nr, 10.0 + d.x, d.y > 0 ? 3 : 0, 0, -1, 0, 0, 0);
    return;
#endif
    // Objects are visited in index order and a hit replaces the record only when strictly nearer, which is the
    // (dist, object index) rule of near_consider without its tie test; the Bezier fields of the record stay as
    // near_init left them, so three values are carried through the loop instead of eight.
    double best = nr.dist;
    int bobj = nr.obj, binside = nr.inside;
#if SPPM_FAST_DIV
    const double ix = rcp_nr(d.x), iy = rcp_nr(d.y), iz = rcp_nr(d.z);   // for the axis planes; a zero component gives inf / NaN, deselected below
#endif
    const bool ax_ok = active & !is_zero(d.x), ay_ok = active & !is_zero(d.y), az_ok = active & !is_zero(d.z);
    // Two objects that are the same object listed twice must get the same bits wherever they are worked on (the first
    // listed wins a tie): no contraction is left to the compiler in these bodies, every fused operation is written out.
    auto dotf = [](const d3 u, const d3 v) { return fma(u.z, v.z, fma(u.y, v.y, u.x * v.x)); };
    // A plane whose normal is +-1 along one axis k (every wall of the reference's scene): the two dot products of
    // Obj.h:65-84 have one non-zero term each, dot(d, n) = d[k] n[k] and dot(P0 - o, n) = (P0[k] - o[k]) n[k], both
    // exact, and the sign n[k] cancels in the quotient (rcp_nr is odd): the same bits as the general form gives.
    auto axis_plane = [&](const int a, const double pk, const double ok_, const double dk, const double ik, const bool dir_ok) {
#pragma clang fp contract(off)
#if SPPM_FAST_DIV
        const double dist = (pk - ok_) * ik;
#else
        const double dist = (pk - ok_) / dk;
#endif
        const bool ok = dir_ok & (dist > SPPM_EPS) & (dist < best);
        best = ok ? dist : best; bobj = ok ? a : bobj; binside = ok ? 0 : binside;
    };
    auto sphere = [&](const int a, const AObj& q) {
#pragma clang fp contract(off)
        const d3 L = mk(q.a, q.b, q.c) - o;
        const double project = dotf(L, d);
        const double det2 = fma(q.r, q.r, -fma(-project, project, dotf(L, L)));    // r^2 - (|L|^2 - project^2)
        const double pos = fmax(det2, 1e-300);
        const double det = pos * rsqrt_nr(pos);                                    // sqrt_nr without its branch
        const double dist1 = project - det, dist2 = project + det;
        const double dist = (dist1 > SPPM_EPS) ? dist1 : dist2;
        const bool ok = active & !(det2 < SPPM_EPS) & !(dist2 < SPPM_EPS) & (dist < best);
        best = ok ? dist : best; bobj = ok ? a : bobj; binside = ok ? ((dist1 < SPPM_EPS) ? 1 : 0) : binside;
    };
    auto plane = [&](const int a) {   // a plane in general position: read where it lies, none of the reference's scene is one
#pragma clang fp contract(off)
        const d3 p0 = ld3(S.objs[a].p0), n = ld3(S.objs[a].n);
        const double project = dotf(d, n);
#if SPPM_FAST_DIV
        const double dist = dotf(p0 - o, n) * rcp_nr(project);      // the quotient to the last place or two, in half the instructions
#else
        const double dist = dotf(p0 - o, n) / project;
#endif
        const bool ok = active & !is_zero(project) & (dist > SPPM_EPS) & (dist < best);
        best = ok ? dist : best; bobj = ok ? a : bobj; binside = ok ? 0 : binside;
    };
#if !SPPM_FAST_DIV
    const double ix = 0, iy = 0, iz = 0;
#endif
    const int n_runs = S.n_runs;
    if (n_runs >= 0) {
        AObj q = S.aobj[0];                      // the table ends with a spare record: the look-ahead never leaves it
        int a = 0;
        for (int r = 0; r < n_runs; r++) {
            const uint32_t code = (uint32_t)((r < 8 ? S.runs_lo >> (8 * r) : S.runs_hi >> (8 * (r - 8))) & 0xffu);
            const int kind = (int)(code & 15u), end = a + (int)(code >> 4);
            if (kind == A_PLANE_X) {
#pragma unroll 1
                for (; a < end; a++) { const double pk = q.a; q = S.aobj[a + 1]; axis_plane(a, pk, o.x, d.x, ix, ax_ok); }
            } else if (kind == A_PLANE_Y) {
#pragma unroll 1
                for (; a < end; a++) { const double pk = q.a; q = S.aobj[a + 1]; axis_plane(a, pk, o.y, d.y, iy, ay_ok); }
            } else if (kind == A_PLANE_Z) {
#pragma unroll 1
                for (; a < end; a++) { const double pk = q.a; q = S.aobj[a + 1]; axis_plane(a, pk, o.z, d.z, iz, az_ok); }
            } else if (kind == OBJ_SPHERE) {
#pragma unroll 1
                for (; a < end; a++) { const AObj c = q; q = S.aobj[a + 1]; sphere(a, c); }
            } else {
                if (kind == OBJ_PLANE)
                    for (; a < end; a++) plane(a);
                a = end;
                q = S.aobj[a];
            }
        }
    } else {   // more runs than the packed form holds: one object at a time, its kind from its record
        const int n_objs = S.n_objs;
        for (int a = 0; a < n_objs; a++) {
            const int kind = __builtin_amdgcn_readfirstlane(S.objs[a].akind);
            const AObj q = S.aobj[a];
            if (kind == A_PLANE_X) axis_plane(a, q.a, o.x, d.x, ix, ax_ok);
            else if (kind == A_PLANE_Y) axis_plane(a, q.a, o.y, d.y, iy, ay_ok);
            else if (kind == A_PLANE_Z) axis_plane(a, q.a, o.z, d.z, iz, az_ok);
            else if (kind == OBJ_SPHERE) sphere(a, q);
            else if (kind == OBJ_PLANE) plane(a);
        }
    }
    if (bobj != nr.obj || best != nr.dist) { nr.dist = best; nr.obj = bobj; nr.inside = binside; nr.patch = -1; nr.t = 0; nr.u = 0; nr.v = 0; }
}

// Bezier objects, solved inside the calling wave (Bezier.h:240-282).
__device__ __forceinline__ void bezier_nearest_coop(const SceneView& S, bool active, d3 o, d3 d, uint32_t path_hash, uint32_t vertex, Near& nr, uint32_t& newton_calls) {
#ifdef ABLATE_BEZIER
    return;
#endif
    const int n_loop = S.n_bez >= 0 ? S.n_bez : S.n_objs;
    for (int k = 0; k < n_loop; k++) {
        const int a = S.n_bez >= 0 ? (int)((S.bez_idx >> (8 * k)) & 0xffu) : k;
        const DevObject& ob = S.objs[a];
        if (S.n_bez < 0 && __builtin_amdgcn_readfirstlane(ob.type) != OBJ_BEZIER) continue;
        // The object's box (Bezier.h:244) is tested by every lane in every turn of the trace loop, and three rays in a
        // thousand enter it: the fp32 test answers wherever it is sure (aabb_classify), and the reference's fp64 test with its
        // three divisions runs only in a wave where some lane's ray grazes the box.  bmin and bmax lie side by side: one box
        // of six.
        float obj_lb;
        bool need, need_open;
        aabb_classify(ob.bmin, o, d, need, need_open, obj_lb);
        need = need && active;
        need_open = need_open && active;
        if (__ballot(need_open)) need = need || (need_open && aabb_hit_cold(ob.bmin, o, d));
        float t, u, v;
        int patch;
        bezier_coop(S, ob, need, o, d, path_hash, vertex, (float)nr.dist, t, u, v, patch, newton_calls);
        if (need && patch >= 0) near_consider(nr, (double)t, a, 0, patch, t, u, v);
    }
}

// Bezier objects, deferred: only the AABB culls (object box per lane, patch boxes one per lane
// for one ray at a time, Bezier.h:176-186, :244).  words[] receives a bit per patch (global
// patch index, up to 256) whose box the lane's ray enters; the Newton solves run in k_newton.
#define SPPM_MASK_WORDS 4
__device__ __forceinline__ void bezier_patch_masks(const SceneView& S, bool active, d3 o, d3 d, uint64_t words[SPPM_MASK_WORDS]) {
    const int lane = lane_id();
#pragma unroll
    for (int w = 0; w < SPPM_MASK_WORDS; w++) words[w] = 0;
    const int n_loop = S.n_bez >= 0 ? S.n_bez : S.n_objs;
    for (int k = 0; k < n_loop; k++) {
        const int a = S.n_bez >= 0 ? (int)((S.bez_idx >> (8 * k)) & 0xffu) : k;
        const DevObject& ob = S.objs[a];
        if (S.n_bez < 0 && __builtin_amdgcn_readfirstlane(ob.type) != OBJ_BEZIER) continue;
        const int n_patches = __builtin_amdgcn_readfirstlane(ob.n_patches), first_patch = __builtin_amdgcn_readfirstlane(ob.first_patch);
        float obj_lb;
        bool need, need_open;
        aabb_classify(ob.bmin, o, d, need, need_open, obj_lb);                         // as in bezier_nearest_coop
        need = need && active;
        need_open = need_open && active;
        if (__ballot(need_open)) need = need || (need_open && aabb_hit_cold(ob.bmin, o, d));
        uint64_t m = __ballot(need);
        while (m) {
            int src = __ffsll((unsigned long long)m) - 1;
            m &= m - 1;
            d3 so = mk(bcast_d(o.x, src), bcast_d(o.y, src), bcast_d(o.z, src));
            d3 sd = mk(bcast_d(d.x, src), bcast_d(d.y, src), bcast_d(d.z, src));
            for (int base = 0; base < n_patches; base += 64) {
                int p = first_patch + base + lane;
                bool ph = (base + lane < n_patches) && aabb_hit(S.patch_box + 6 * p, so, sd);
                uint64_t pm = __ballot(ph);
                if (pm == 0) continue;
                const int g = first_patch + base, wi = g >> 6, sh = g & 63;
                const uint64_t lo = pm << sh, hi = sh ? (pm >> (64 - sh)) : 0ull;
                if (lane == src) {
#pragma unroll
                    for (int w = 0; w < SPPM_MASK_WORDS; w++) {
                        if (w == wi) words[w] |= lo;
                        if (w == wi + 1) words[w] |= hi;
                    }
                }
            }
        }
    }
}

struct Hit {
    d3 pos, n;
    float color[3];   // the object's colour; superseded by the texture where tap.on (hit_color gives the final one)
    TexTap tap;       // texture lookup in flight
    double dist;
    int obj;      // -1 = none
    int inside;
};
// GetColor of the hit: the blended texels if the object is textured, its plain colour otherwise.  Called as late as
// the caller can afford: the texel loads issued by finish_hit travel meanwhile.
__device__ __forceinline__ void hit_color(const Hit& h, float out[3]) {
    out[0] = h.color[0]; out[1] = h.color[1]; out[2] = h.color[2];
    if (h.tap.on) tex_blend(h.tap, out);
}

// Collision of the winner only (GetColor is a pure function, so evaluating it for the nearest
// object alone gives what the reference's per-object evaluation gives).
__device__ __forceinline__ Hit finish_hit(const SceneView& S, d3 o, d3 d, const Near& nr) {
    Hit h;
    h.obj = nr.obj; h.dist = nr.dist; h.inside = nr.inside;
    h.pos = mk(0, 0, 0); h.n = mk(0, 0, 0); h.color[0] = h.color[1] = h.color[2] = 0;
    h.tap.on = 0; h.tap.detR = h.tap.detC = 0;
    h.tap.a = h.tap.b = h.tap.c = h.tap.d = make_uchar4(0, 0, 0, 0);
    if (nr.obj >= 0) {
        const DevObject& ob = S.objs[nr.obj];
        h.pos = o + d * nr.dist;
        h.color[0] = ob.color[0]; h.color[1] = ob.color[1]; h.color[2] = ob.color[2];
#ifdef ABLATE_TEX
        if (ob.type == OBJ_PLANE) { h.n = ld3(ob.n); return h; }
        if (ob.type == OBJ_SPHERE) { h.n = get_unit(h.pos - ld3(ob.p0)); return h; }
#endif
        if (ob.type == OBJ_PLANE) {
            h.n = ld3(ob.n);
            if (ob.texture >= 0) {                                                        // Obj.h:86-100
                const d3 p0 = ld3(ob.p0);
                const int vdex = ob.tex_vdex, udex = ob.tex_udex;   // (ndir + 2) % 3, (ndir + 1) % 3, ndir = the last non-zero component of the normal
#if SPPM_FAST_DIV
                double v = 0.5 + (comp(h.pos, vdex) - comp(p0, vdex)) * (1.0 / 400.0);      // Module(Vec3(400, 0, 0)), Obj.h:95
                double u = 0.5 + (comp(h.pos, udex) - comp(p0, udex)) * (1.0 / 300.0);
#else
                double v = 0.5 + (comp(h.pos, vdex) - comp(p0, vdex)) / 400.0;
                double u = 0.5 + (comp(h.pos, udex) - comp(p0, udex)) / 300.0;
#endif
                tex_fetch(tex_of(ob), S.texels, u, v, h.tap);
            }
        } else if (ob.type == OBJ_SPHERE) {
            d3 c = ld3(ob.p0);
#if SPPM_FAST_UNIT
            h.n = get_unit_near1((h.pos - c) * ob.inv_radius);       // |pos - c| is the radius but for the rounding of pos
#else
            h.n = get_unit(h.pos - c);
#endif
            if (ob.texture >= 0) {                                                        // Obj.h:140-153
                const d3 texU = mk(0, 0.70710678118654757, -0.70710678118654757), texV = mk(1, 0, 0);
                d3 N = h.n;
                double theta = acos(dot(N, texV));
                double t = dot(N, texU) / sin(theta);
                t = (t >= -1) ? t : -1;
                t = (t <= 1) ? t : 1;
                double phi = acos(t);
                double u = theta / SPPM_PI, v = phi / (2 * SPPM_PI);
                v = (dot(N, cross(texU, texV)) < 0) ? (1 - v) : v;
                tex_fetch(tex_of(ob), S.texels, u, v, h.tap);
            }
        } else {
            // normal = Su x Sv at the winning (u, v), turned against the ray (Bezier.h:267-273)
            f3 Sp, Su, Sv;
            patch_eval(S.patch_cp + nr.patch * 48, nr.u, nr.v, Sp, Su, Sv);   // control points read straight from LDS
            d3 N = cross(mk(Su.x, Su.y, Su.z), mk(Sv.x, Sv.y, Sv.z));
            if (dot(N, d) > 0) N = neg(N);
            h.n = get_unit_any(N);
            // GetColor(u, t): the second texture coordinate is the ray parameter (Bezier.h:278)
            if (ob.texture >= 0) tex_fetch(tex_of(ob), S.texels, (double)nr.u, (double)nr.t, h.tap);
        }
    }
    return h;
}

// Scene::NearCollide with the Bezier solves done inside the wave.  All lanes of the wave must
// call this together; lanes with active = false take part but get obj = -1.
__device__ __forceinline__ Hit near_collide(const SceneView& S, bool active, d3 o, d3 d, uint32_t path_hash, uint32_t vertex, uint32_t& newton_calls) {
    Near nr;
    near_init(nr);
    analytic_nearest(S, active, o, d, nr);
    bezier_nearest_coop(S, active, o, d, path_hash, vertex, nr, newton_calls);
    return finish_hit(S, o, d, nr);
}

}  // namespace sppm
