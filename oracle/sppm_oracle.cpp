/*
 * sppm_oracle.cpp — CPU oracle of the SPPM hot path.
 *
 * TEST INFRASTRUCTURE ONLY (see sppm_oracle.h).  PARITY UNPINNED IN VALUE: the reference
 * has no tests or golden vectors and cannot be built here; this file restates
 * its algorithm function by function, in the reference's own operation order
 * and in double (optionally float inside Bezier3::newton), and is pinned by the
 * analytic known-answer tests under tests/ and, in structure only (correlation),
 * by the one rendered picture the reference's checkout holds
 * (tests/test_reference_picture.py).
 *
 * Each function cites the reference code it follows (paths relative to the
 * reference checkout, raytracer/...).
 *
 * Random numbers.  Two modes:
 *  - literal: one global multiply-with-carry generator restating OpenCV's
 *    cv::RNG (Vec3.h:5,15-27; OpenCV is not in the reference tree, the
 *    algorithm is the published one: state = (u32)state * 4164903690 + (state >> 32),
 *    seed 0xffffffff, double = ((u64)next << 32 | next) * 2^-64), drawn in source
 *    order on one thread.
 *  - parity: counter-based.  A draw is a pure function
 *        u = (murmur3_32(seed_lo, seed_hi, stream, path, sub, idx) >> 8) * 2^-24
 *    with stream = replica * 4 + domain (0 eye, 1 photon), path = pixel index
 *    (row * w + col) or photon index, sub = (vertex << 8) | purpose, where vertex
 *    is the eye-tree node (root 0, reflection child 2n+1, refraction child 2n+2) or
 *    the photon's bounce count, and
 *        purpose 0 EMIT    idx = attempt * 3 + axis          (Vec3.h:57-65)
 *        purpose 1 NEWTON  idx = patch * 128 + restart * 2 + {0: u, 1: v}  (Bezier.h:117)
 *        purpose 2 ROULETTE idx = 0                           (Obj.h:33)
 *        purpose 3 DIFFUSE idx = {0: theta draw, 1: phi draw} (Vec3.h:93-94)
 *    so the GPU can evaluate any draw from any lane in any order.
 */
#include "sppm_oracle.h"

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <unordered_map>
#include <vector>

namespace {

// ---- constants, Vec3.h:6-14, Raytracer.h:11-13,45, Bezier.h:6 ----
constexpr double M_EPS_ = 1e-4;
constexpr double PI_ = 3.1415926;
constexpr double PI_INV_ = 0.3183098862;
constexpr double MAX_NUM_ = 1e20;
constexpr int MAX_DEP_ = 13;
constexpr double INIT_R2_ = 2.0;
constexpr int MAX_ITER_ = 10;
constexpr double ALPHA_ = 0.7;

// ---------------------------------------------------------------- Vec3.h
struct V3 {
    double x, y, z;
    V3() : x(0), y(0), z(0) {}
    V3(double a, double b, double c) : x(a), y(b), z(c) {}
    double operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
    double& at(int i) { return i == 0 ? x : (i == 1 ? y : z); }
};
inline V3 operator+(const V3& a, const V3& b) { return V3(a.x + b.x, a.y + b.y, a.z + b.z); }
inline V3 operator-(const V3& a, const V3& b) { return V3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline V3 operator*(const V3& a, double s) { return V3(a.x * s, a.y * s, a.z * s); }
inline V3 operator*(double s, const V3& a) { return V3(a.x * s, a.y * s, a.z * s); }
inline V3 operator-(const V3& a) { return V3(-a.x, -a.y, -a.z); }
inline double dot(const V3& a, const V3& b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline V3 cross(const V3& a, const V3& b) {
    return V3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
inline V3 mul(const V3& a, const V3& b) { return V3(a.x * b.x, a.y * b.y, a.z * b.z); }
inline bool IsZero(double i) { return std::fabs(i) < M_EPS_; }           // Vec3.h:23
inline bool IsZeroV(const V3& v) { return IsZero(v.x) || IsZero(v.y) || IsZero(v.z); }  // Vec3.h:72-79 (any component)
inline double Module2(const V3& v) { return v.x * v.x + v.y * v.y + v.z * v.z; }     // Vec3.h:44
inline double Module(const V3& v) { return std::sqrt(Module2(v)); }
inline V3 GetUnit(const V3& v) {                                         // Vec3.h:56
    double mo = Module(v);
    return IsZero(mo) ? v : v * (1.0 / mo);
}
inline V3 Normalized(V3 v) {                                             // Vec3.h:48-55
    double mo = Module(v);
    if (!IsZero(mo)) {
        double mo_ = 1 / mo;
        v.x *= mo_; v.y *= mo_; v.z *= mo_;
    }
    return v;
}
inline V3 GetRefl(const V3& v, const V3& norm) { return v - 2 * dot(v, norm) * norm; }  // Vec3.h:80-83
inline V3 GetUnitRefl(const V3& v, const V3& norm) { return GetUnit(GetRefl(v, norm)); }
inline V3 GetAnormal(const V3& v) {                                      // Vec3.h:85-89
    return (v.x == 0 && v.y == 0) ? V3(1, 0, 0) : GetUnit(V3(v.y, -v.x, 0));
}
inline V3 rotated(const V3& v, const V3& axis, double angle) {           // Vec3.h:99-115
    if (std::fabs(angle) < M_EPS_) return v;
    V3 ret;
    double cost = std::cos(angle);
    double sint = std::sin(angle);
    ret.x += v.x * (axis.x * axis.x + (1 - axis.x * axis.x) * cost);
    ret.x += v.y * (axis.x * axis.y * (1 - cost) - axis.z * sint);
    ret.x += v.z * (axis.x * axis.z * (1 - cost) + axis.y * sint);
    ret.y += v.x * (axis.y * axis.x * (1 - cost) + axis.z * sint);
    ret.y += v.y * (axis.y * axis.y + (1 - axis.y * axis.y) * cost);
    ret.y += v.z * (axis.y * axis.z * (1 - cost) - axis.x * sint);
    ret.z += v.x * (axis.z * axis.x * (1 - cost) - axis.y * sint);
    ret.z += v.y * (axis.z * axis.y * (1 - cost) + axis.x * sint);
    ret.z += v.z * (axis.z * axis.z + (1 - axis.z * axis.z) * cost);
    return ret;
}
inline V3 RandReflFrom(const V3& norm, double u1, double u2) {           // Vec3.h:90-98
    double theta = std::acos(std::sqrt(u1));
    double phi = 2 * PI_ * u2;
    V3 ret = rotated(norm, GetAnormal(norm), theta);
    return GetUnit(rotated(ret, norm, phi));
}
inline V3 refracted(const V3& I, const V3& N, double n) {                // Vec3.h:124-134
    double cosI = -dot(N, I);
    double cosR2 = 1 - (1 - std::pow(cosI, 2)) * std::pow(n, 2);
    return (cosR2 > M_EPS_) ? I * n + N * (n * cosI - std::sqrt(cosR2)) : GetRefl(I, N);
}
inline double GetPower(const V3& v) { return (v.x + v.y + v.z) / 3; }   // Vec3.h:116-119

// ---------------------------------------------------------------- RNG
struct MWC {                                                             // cv::RNG behind Vec3.h:5
    uint64_t state = 0xffffffffULL;
    uint32_t next() {
        state = (uint64_t)(uint32_t)state * 4164903690U + (uint32_t)(state >> 32);
        return (uint32_t)state;
    }
    double uniform01() {
        uint32_t t = next();
        return (double)(((uint64_t)t << 32) | next()) * 5.4210108624275221700372640043497e-20;
    }
};

inline uint32_t rotl32(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }
inline uint32_t mm3(uint32_t h, uint32_t k) {
    k *= 0xcc9e2d51u; k = rotl32(k, 15); k *= 0x1b873593u;
    h ^= k; h = rotl32(h, 13); h = h * 5u + 0xe6546b64u;
    return h;
}
inline uint32_t fmix3(uint32_t h) {
    h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16;
    return h;
}
inline uint32_t rng_u32(uint64_t seed, uint32_t stream, uint32_t path, uint32_t sub, uint32_t idx) {
    uint32_t h = 0x9747b28cu;
    h = mm3(h, (uint32_t)seed);
    h = mm3(h, (uint32_t)(seed >> 32));
    h = mm3(h, stream);
    h = mm3(h, path);
    h = mm3(h, sub);
    h = mm3(h, idx);
    return fmix3(h ^ 24u);
}
enum { P_EMIT = 0, P_NEWTON = 1, P_ROULETTE = 2, P_DIFFUSE = 3 };
enum { D_EYE = 0, D_PHOTON = 1 };

// One random-number source as seen by one path vertex.
struct Rand {
    MWC* mwc;          // non-null = literal mode
    uint64_t seed;
    uint32_t stream, path, vertex;
    double draw(int purpose, uint32_t idx) const {
        if (mwc) return mwc->uniform01();
        uint32_t x = rng_u32(seed, stream, path, (vertex << 8) | (uint32_t)purpose, idx);
        return (double)(x >> 8) * (1.0 / 16777216.0);
    }
    // GetRandu(r1, r2), Vec3.h:15-18 -> cv::RNG::uniform(double,double) = u * (b - a) + a
    double randu(double a, double b, int purpose, uint32_t idx) const { return draw(purpose, idx) * (b - a) + a; }
};

inline V3 GetUnitRandVec(const Rand& r) {                                // Vec3.h:57-65
    double x = 0, y = 0, z = 0;
    for (uint32_t k = 0; k < 64; k++) {  // rejection loop; 64 attempts bound it (p(reject)^64 ~ 1e-21)
        x = r.randu(-1., 1., P_EMIT, 3 * k + 0);
        y = r.randu(-1., 1., P_EMIT, 3 * k + 1);
        z = r.randu(-1., 1., P_EMIT, 3 * k + 2);
        double m = x * x + y * y + z * z;
        if (!(m > 1 || m < M_EPS_)) break;
    }
    return GetUnit(V3(x, y, z));
}

// ---------------------------------------------------------------- Element.h
struct Material {                                                        // Element.h:7-19
    V3 diff, refl, refr;
    double refrn = 1.5, refln = 1;
    bool IsDiff() const { return !IsZeroV(diff); }
    bool IsRefl() const { return !IsZeroV(refl); }
    bool IsRefr() const { return !IsZeroV(refr); }
};
struct Texture {                                                         // Element.h:43-75
    int rows = 0, cols = 0;
    std::vector<V3> m_map;  // rows x cols
    V3 colorUV(double u, double v) const {                               // Element.h:61-72
        double row = (u - std::floor(u)) * rows, col = (v - std::floor(v)) * cols;
        int r1 = (int)std::floor(row + 1e-10), r2 = r1 + 1;
        int c1 = (int)std::floor(col + 1e-10), c2 = c1 + 1;
        double detR = r2 - row, detC = c2 - col;
        r1 = (r1 >= 0) ? (r1 >= rows ? 0 : r1) : (rows - 1);
        c1 = (c1 >= 0) ? (c1 >= cols ? 0 : c1) : (cols - 1);
        r2 = (r2 < rows) ? r2 : 0;
        c2 = (c2 < cols) ? c2 : 0;
        const V3& a = m_map[(size_t)r1 * cols + c1];
        const V3& b = m_map[(size_t)r1 * cols + c2];
        const V3& c = m_map[(size_t)r2 * cols + c1];
        const V3& d = m_map[(size_t)r2 * cols + c2];
        return a * detR * detC + b * detR * (1 - detC) + c * (1 - detR) * detC + d * (1 - detR) * (1 - detC);
    }
};
struct Collision {                                                       // Element.h:20-38
    V3 Pos, N, From, color;
    double dist = 0;
    int obj = -1;
    bool inside = false;
    bool IsValid() const { return obj >= 0; }
    bool IsFront() const { return !inside; }
};

// ---------------------------------------------------------------- Bezier.h
struct AABB {                                                            // Bezier.h:7-57
    V3 pmin = V3(MAX_NUM_, MAX_NUM_, MAX_NUM_), pmax = V3(-MAX_NUM_, -MAX_NUM_, -MAX_NUM_);
    void addpoint(const V3& p) {
        for (int i = 0; i < 3; i++) {
            if (pmin[i] > p[i]) pmin.at(i) = p[i];
            if (pmax[i] < p[i]) pmax.at(i) = p[i];
        }
    }
    bool intersect(const V3& org, const V3& dir) const {                 // Bezier.h:20-51
        double t = 0;
        for (int i = 0; i < 3; i++) {
            if (!IsZero(dir[i])) {
                if (dir[i] > 0) t = (pmin[i] - org[i]) / dir[i];
                else t = (pmax[i] - org[i]) / dir[i];
                if (t > 0) {
                    V3 Colp = org + t * dir;
                    bool flag = true;
                    for (int j = 0; j < 3; j++)
                        if (j != i)
                            if (pmax[j] < Colp[j] || pmin[j] > Colp[j]) flag = false;
                    if (flag) return true;
                }
            }
        }
        return false;
    }
};

template <class R>
struct T3 { R x, y, z; };
template <class R> inline T3<R> tadd(T3<R> a, T3<R> b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
template <class R> inline T3<R> tsub(T3<R> a, T3<R> b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
template <class R> inline T3<R> tmul(T3<R> a, R s) { return {a.x * s, a.y * s, a.z * s}; }

template <class R>
inline T3<R> evalCurvePoint(const T3<R>* P, R t) {                       // Bezier.h:69-76
    R c0 = (1 - t) * (1 - t) * (1 - t);
    R c1 = 3 * t * (1 - t) * (1 - t);
    R c2 = 3 * t * t * (1 - t);
    R c3 = t * t * t;
    return tadd(tadd(tadd(tmul(P[0], c0), tmul(P[1], c1)), tmul(P[2], c2)), tmul(P[3], c3));
}
template <class R>
inline T3<R> evalCurveDeriv(const T3<R>* P, R t) {                       // Bezier.h:77-84
    R c0 = -3 * (1 - t) * (1 - t);
    R c1 = 3 * (1 - t) * (1 - t) - 6 * t * (1 - t);
    R c2 = 6 * t * (1 - t) - 3 * t * t;
    R c3 = 3 * t * t;
    return tadd(tadd(tadd(tmul(P[0], c0), tmul(P[1], c1)), tmul(P[2], c2)), tmul(P[3], c3));
}
template <class R>
inline T3<R> evalPatchPoint(const T3<R>* P, R u, R v) {                  // Bezier.h:85-90
    T3<R> curve[4];
    for (int i = 0; i < 4; i++) curve[i] = evalCurvePoint(P + 4 * i, u);
    return evalCurvePoint(curve, v);
}
template <class R>
inline T3<R> evalPatchDerivU(const T3<R>* P, R u, R v) {                 // Bezier.h:91-105
    T3<R> Pv[4], curve[4];
    for (int i = 0; i < 4; i++) {
        for (int j = 0; j < 4; j++) Pv[j] = P[i + 4 * j];
        curve[i] = evalCurvePoint(Pv, v);
    }
    return evalCurveDeriv(curve, u);
}
template <class R>
inline T3<R> evalPatchDerivV(const T3<R>* P, R u, R v) {                 // Bezier.h:106-111
    T3<R> curve[4];
    for (int i = 0; i < 4; i++) curve[i] = evalCurvePoint(P + 4 * i, u);
    return evalCurveDeriv(curve, v);
}

// cv::Matx33d::inv() (Bezier.h:130), DECOMP_LU on a 3x3: determinant by cofactor
// expansion; det == 0 -> the zero matrix; else adjugate * (1/det).
template <class R>
inline void inv33(const R a[3][3], R b[3][3]) {
    R d = a[0][0] * (a[1][1] * a[2][2] - a[1][2] * a[2][1]) - a[0][1] * (a[1][0] * a[2][2] - a[1][2] * a[2][0]) +
          a[0][2] * (a[1][0] * a[2][1] - a[1][1] * a[2][0]);
    if (d == 0) {
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) b[i][j] = 0;
        return;
    }
    d = 1 / d;
    b[0][0] = (a[1][1] * a[2][2] - a[1][2] * a[2][1]) * d;
    b[0][1] = (a[0][2] * a[2][1] - a[0][1] * a[2][2]) * d;
    b[0][2] = (a[0][1] * a[1][2] - a[0][2] * a[1][1]) * d;
    b[1][0] = (a[1][2] * a[2][0] - a[1][0] * a[2][2]) * d;
    b[1][1] = (a[0][0] * a[2][2] - a[0][2] * a[2][0]) * d;
    b[1][2] = (a[0][2] * a[1][0] - a[0][0] * a[1][2]) * d;
    b[2][0] = (a[1][0] * a[2][1] - a[1][1] * a[2][0]) * d;
    b[2][1] = (a[0][1] * a[2][0] - a[0][0] * a[2][1]) * d;
    b[2][2] = (a[0][0] * a[1][1] - a[0][1] * a[1][0]) * d;
}

struct Patch {                                                           // Bezier3, Bezier.h:59-187
    V3 P[16];
    T3<float> Pf[16];
    AABB aabb;
};

// Bezier3::newton, Bezier.h:112-159.  Returns (t,u,v) or MAX_VEC3.
template <class R>
V3 newton(const T3<R>* P, const V3& org_, const V3& dir_, int maxIter, const Rand& rnd, uint32_t patch_id) {
    T3<R> org = {(R)org_.x, (R)org_.y, (R)org_.z}, dir = {(R)dir_.x, (R)dir_.y, (R)dir_.z};
    bool found = false;
    R best[3] = {0, 0, 0};
    R mindist = (R)MAX_NUM_;
    for (int o = 0; o < maxIter * 5; o++) {
        // Vec3 x(0, GetRandu(0.,1.), GetRandu(0.,1.)): u is drawn before v here (the
        // reference leaves the order to the compiler).
        R x[3];
        x[0] = 0;
        x[1] = (R)rnd.randu(0., 1., P_NEWTON, patch_id * 128u + (uint32_t)o * 2u + 0u);
        x[2] = (R)rnd.randu(0., 1., P_NEWTON, patch_id * 128u + (uint32_t)o * 2u + 1u);
        int iter = 0;
        while (iter < maxIter) {
            R prevX[3] = {x[0], x[1], x[2]};
            T3<R> L = tadd(org, tmul(dir, prevX[0]));
            T3<R> P_ = evalPatchPoint(P, prevX[1], prevX[2]);
            T3<R> dU = evalPatchDerivU(P, prevX[1], prevX[2]);
            T3<R> dV = evalPatchDerivV(P, prevX[1], prevX[2]);
            R m[3][3], mi[3][3];
            m[0][0] = dir.x; m[1][0] = dir.y; m[2][0] = dir.z;
            m[0][1] = -dU.x; m[1][1] = -dU.y; m[2][1] = -dU.z;
            m[0][2] = -dV.x; m[1][2] = -dV.y; m[2][2] = -dV.z;
            inv33(m, mi);
            T3<R> r = tsub(L, P_);
            for (int i = 0; i < 3; i++) x[i] = prevX[i] - (mi[i][0] * r.x + mi[i][1] * r.y + mi[i][2] * r.z);
            iter++;
            L = tadd(org, tmul(dir, x[0]));
            P_ = evalPatchPoint(P, x[1], x[2]);
            T3<R> e = tsub(L, P_);
            R res2 = e.x * e.x + e.y * e.y + e.z * e.z;
            if (res2 < (R)M_EPS_ && x[1] <= 1 && x[1] >= 0 && x[2] >= 0 && x[2] <= 1) {
                // xvec.push_back(x) followed by the min-t scan of Bezier.h:146-158:
                // strict '<' keeps the first pushed minimum, negative t included.
                if (x[0] < mindist) {
                    mindist = x[0];
                    best[0] = x[0]; best[1] = x[1]; best[2] = x[2];
                    found = true;
                }
            }
        }
    }
    if (!found) return V3(MAX_NUM_, MAX_NUM_, MAX_NUM_);
    return V3((double)best[0], (double)best[1], (double)best[2]);
}

// ---------------------------------------------------------------- Obj.h / Scene.h
struct Object {
    int type = 0, texture = -1;
    V3 color;
    Material material;
    double diffr = 0, reflr = 0, refrr = 0, allr = 0;                    // Obj.h:11-16
    V3 P0, norm;                                                         // plane
    double radius = 0;                                                   // sphere (center = P0)
    int first_patch = 0, n_patches = 0;                                  // bezier
    AABB aabb;
};

struct HitPoint {                                                        // Raytracer.h:47-80
    V3 pos, n, v;
    int obj = -1, x = -1, y = -1, node = 0;
    V3 wgt;
    double R2 = INIT_R2_;
    int N = 0;
    V3 tao;
    int newN = 0;
    void update() {                                                      // Raytracer.h:69-79
        if (N <= 0 || newN <= 0) return;
        double k = (N + ALPHA_ * newN) / (N + newN);
        R2 *= k; tao = tao * k;
        N += ALPHA_ * newN;
        newN = 0;
    }
};

struct KdNode { int left, right, begin, end, dim; double split; };

}  // namespace

struct oracle_ctx {
    bool newton_fp32 = false, rng_literal = false;
    MWC mwc;
    std::vector<Object> objs;
    std::vector<Patch> patches;
    std::vector<Texture> textures;
    struct Light { V3 pos, color; };
    std::vector<Light> lights;
    sppm_camera cam{};
    bool have_cam = false;
    std::vector<HitPoint> Hvec;
    sppm_counters cnt{};
    uint64_t last_rounds = 0, last_per_round = 0;
    // The radius-update knob of the GPU library (sppm_set_radius_update), restated: HitPoints start with initial_n
    // instead of 0, and the sweep of Raytracer.h:237-245 runs after every rounds_per_update rounds.
    int initial_n = 0;
    uint32_t rounds_per_update = 1;
    // A measuring switch, not the reference's arithmetic: every ray direction and every hit normal is rounded to float
    // before it is used (positions, distances and the 1e-4 guards stay double).  It answers what a device path that kept
    // directions and normals in float would do to HitPoints, tallies and the image (tests/report_narrow_ray_bias.py).
    bool narrow_rays = false;
    V3 narrow(const V3& v) const { return narrow_rays ? V3((double)(float)v.x, (double)(float)v.y, (double)(float)v.z) : v; }
    // exact uniform grid over HitPoints
    std::unordered_map<uint64_t, std::vector<int>> grid;
    // kd-tree
    std::vector<KdNode> kd;
    std::vector<int> kd_idx;

    Rand make_rand(uint64_t seed, uint32_t replica, int domain, uint32_t path, uint32_t vertex) {
        Rand r;
        r.mwc = rng_literal ? &mwc : nullptr;
        r.seed = seed; r.stream = replica * 4u + (uint32_t)domain; r.path = path; r.vertex = vertex;
        return r;
    }

    // ---- Obj.h:65-100 PlaneObj
    V3 plane_color(const Object& o, const V3& P) const {
        if (o.texture < 0) return o.color;
        int ndir = 0;
        for (int i = 0; i < 3; i++) if (o.norm[i] != 0) ndir = i;
        int vdex = (ndir + 2) % 3;
        int udex = (ndir + 1) % 3;
        double v = 0.5 + (P[vdex] - o.P0[vdex]) / Module(V3(400, 0, 0));
        double u = 0.5 + (P[udex] - o.P0[udex]) / Module(V3(0, 0, 300));
        return textures[o.texture].colorUV(u, v);
    }
    bool plane_hit(const Object& o, const V3& org, const V3& dir, double& dist) const {
        double project = dot(dir, o.norm);
        if (IsZero(project)) return false;
        dist = dot(o.P0 - org, o.norm) / project;
        return dist > M_EPS_;
    }
    // ---- Obj.h:111-153 SphereObj
    V3 sphere_color(const Object& o, const V3& P) const {
        if (o.texture < 0) return o.color;
        const V3 texU = GetUnit(V3(0, 3, -3)), texV = GetUnit(V3(1, 0, 0));
        V3 N = GetUnit(P - o.P0);
        double theta = std::acos(dot(N, texV));
        double t = dot(N, texU) / std::sin(theta);
        t = (t >= -1) ? t : -1;
        t = (t <= 1) ? t : 1;
        double phi = std::acos(t);
        double u = theta / PI_, v = phi / (2 * PI_);
        v = (dot(N, cross(texU, texV)) < 0) ? (1 - v) : v;
        return textures[o.texture].colorUV(u, v);
    }
    bool sphere_hit(const Object& o, const V3& org, const V3& d, double& dist, bool& inside) const {
        V3 L = o.P0 - org;
        double project = dot(L, d);
        double det2 = o.radius * o.radius - (Module2(L) - project * project);
        if (det2 < M_EPS_) return false;
        double det = std::sqrt(det2);
        double dist1 = project - det, dist2 = project + det;
        if (dist2 < M_EPS_) return false;
        dist = (dist1 > M_EPS_) ? dist1 : dist2;
        inside = (dist1 < M_EPS_);
        return true;
    }
    // ---- Bezier.h:240-282 Bezier3Obj::GetIntersect
    bool bezier_hit(const Object& o, int obj_id, const V3& org, const V3& d, const Rand& rnd, Collision& c) {
        if (!o.aabb.intersect(org, d)) return false;
        V3 para(MAX_NUM_, MAX_NUM_, MAX_NUM_);
        int index = -1;
        for (int i = 0; i < o.n_patches; i++) {
            const Patch& pt = patches[o.first_patch + i];
            V3 paratmp(MAX_NUM_, MAX_NUM_, MAX_NUM_);
            if (pt.aabb.intersect(org, d)) {                             // Bezier3::intersect, Bezier.h:176-186
                cnt.newton_calls++;
                if (newton_fp32) paratmp = newton<float>(pt.Pf, org, d, MAX_ITER_, rnd, (uint32_t)(o.first_patch + i));
                else {
                    T3<double> Pd[16];
                    for (int k = 0; k < 16; k++) Pd[k] = {pt.P[k].x, pt.P[k].y, pt.P[k].z};
                    paratmp = newton<double>(Pd, org, d, MAX_ITER_, rnd, (uint32_t)(o.first_patch + i));
                }
            }
            if (paratmp.x < M_EPS_ || paratmp.x > para.x || paratmp.y < -M_EPS_ || paratmp.y > 1 + M_EPS_ ||
                paratmp.z < -M_EPS_ || paratmp.z > 1 + M_EPS_)
                continue;
            if (paratmp.x < para.x) { para = paratmp; index = i; }
        }
        if (index == -1) return false;
        const Patch& pt = patches[o.first_patch + index];
        T3<double> Pd[16];
        for (int k = 0; k < 16; k++) Pd[k] = {pt.P[k].x, pt.P[k].y, pt.P[k].z};
        c.Pos = org + d * para.x;
        c.From = d;
        T3<double> U = evalPatchDerivU(Pd, para.y, para.z);
        T3<double> V = evalPatchDerivV(Pd, para.y, para.z);
        V3 N = cross(V3(U.x, U.y, U.z), V3(V.x, V.y, V.z));
        if (dot(N, d) > 0) N = -N;
        N = Normalized(N);
        c.N = N;
        c.obj = obj_id;
        c.dist = para.x;
        c.inside = false;
        // GetColor(para.val[1], para.val[0]): the second texture coordinate is the ray parameter t (Bezier.h:278)
        c.color = (o.texture < 0) ? o.color : textures[o.texture].colorUV(para.y, para.x);
        return true;
    }
    // ---- Scene.h:165-182 NearCollide
    Collision NearCollide(const V3& org, const V3& dir, const Rand& rnd) {
        double mindist = MAX_NUM_;
        Collision ctmp;
        for (size_t a = 0; a < objs.size(); a++) {
            const Object& o = objs[a];
            Collision t;
            bool valid = false;
            if (o.type == SPPM_OBJ_PLANE) {
                double dist;
                if (plane_hit(o, org, dir, dist)) {
                    V3 P = org + dir * dist;
                    t.Pos = P; t.From = dir; t.dist = dist; t.obj = (int)a; t.N = o.norm; t.color = plane_color(o, P);
                    valid = true;
                }
            } else if (o.type == SPPM_OBJ_SPHERE) {
                double dist; bool inside;
                if (sphere_hit(o, org, dir, dist, inside)) {
                    t.Pos = org + dir * dist;
                    t.From = dir;
                    t.N = GetUnit(t.Pos - o.P0);
                    t.obj = (int)a; t.dist = dist; t.inside = inside;
                    t.color = sphere_color(o, t.Pos);
                    valid = true;
                }
            } else {
                valid = bezier_hit(o, (int)a, org, dir, rnd, t);
            }
            if (valid && t.dist < mindist) { mindist = t.dist; ctmp = t; }
        }
        return ctmp;
    }

    // ---- Obj.h:30-45 Roulette
    int Roulette(const Object& o, const Rand& rnd) const {
        double r = rnd.randu(0., o.allr, P_ROULETTE, 0);
        if (o.diffr > r) return 0;
        else if (o.diffr + o.reflr > r) return 1;
        else return 2;
    }

    // ---- Raytracer.h:296-357 ViewTrace (recursive tree)
    void ViewTrace(const V3& org, const V3& dir_in, int dep, V3 wgt, int x, int y, uint32_t node,
                   uint64_t seed, uint32_t replica) {
        const V3 dir = narrow(dir_in);
        if (dep <= MAX_DEP_) cnt.eye_segments++;   // like `segments`: the call whose result the depth test discards is not tallied
        Rand rnd = make_rand(seed, replica, D_EYE, (uint32_t)(x * cam.w + y), node);
        Collision c1 = NearCollide(org, dir, rnd);
        c1.N = narrow(c1.N);
        if (!c1.IsValid()) return;
        if (dep > MAX_DEP_) return;
        dep++;
        const Object& obj = objs[c1.obj];
        if (obj.material.IsDiff()) {
            HitPoint hp;
            hp.pos = c1.Pos; hp.n = c1.N; hp.v = c1.From; hp.obj = c1.obj; hp.x = x; hp.y = y; hp.node = (int)node;
            hp.wgt = mul(mul(c1.color, wgt), obj.material.diff);
            hp.N = initial_n;
            Hvec.push_back(hp);
        }
        if (obj.material.IsRefl()) {
            ViewTrace(c1.Pos, GetUnitRefl(c1.From, c1.N), dep, mul(mul(c1.color, wgt), obj.material.refl), x, y,
                      2 * node + 1, seed, replica);
        }
        if (obj.material.IsRefr()) {
            double refrn = c1.IsFront() ? 1 / obj.material.refrn : obj.material.refrn;
            V3 refr = refracted(c1.From, c1.IsFront() ? c1.N : -c1.N, refrn);
            refr = Normalized(refr);
            ViewTrace(c1.Pos, refr, dep, mul(mul(c1.color, wgt), obj.material.refr), x, y, 2 * node + 2, seed, replica);
        }
    }

    // ---- gather structures
    static uint64_t cell_key(const V3& p, double cs) {
        int64_t ix = (int64_t)std::floor(p.x / cs), iy = (int64_t)std::floor(p.y / cs), iz = (int64_t)std::floor(p.z / cs);
        auto c = [](int64_t v) { return (uint64_t)(std::min<int64_t>(std::max<int64_t>(v, -1000000), 1000000) + 1048576); };
        return (c(ix) << 42) | (c(iy) << 21) | c(iz);
    }
    void build_grid() {
        grid.clear();
        for (size_t i = 0; i < Hvec.size(); i++) grid[cell_key(Hvec[i].pos, 1.5)].push_back((int)i);
    }
    int kd_build(int begin, int end) {
        KdNode nd{-1, -1, begin, end, 0, 0};
        int id = (int)kd.size();
        kd.push_back(nd);
        if (end - begin <= 10) return id;  // KDTreeSingleIndexParams(): leaf_max_size 10
        V3 lo(MAX_NUM_, MAX_NUM_, MAX_NUM_), hi(-MAX_NUM_, -MAX_NUM_, -MAX_NUM_);
        for (int i = begin; i < end; i++)
            for (int k = 0; k < 3; k++) {
                double v = Hvec[kd_idx[i]].pos[k];
                if (v < lo[k]) lo.at(k) = v;
                if (v > hi[k]) hi.at(k) = v;
            }
        int dim = 0;
        for (int k = 1; k < 3; k++) if (hi[k] - lo[k] > hi[dim] - lo[dim]) dim = k;
        int mid = (begin + end) / 2;
        std::nth_element(kd_idx.begin() + begin, kd_idx.begin() + mid, kd_idx.begin() + end,
                         [&](int a, int b) { return Hvec[a].pos[dim] < Hvec[b].pos[dim]; });
        double split = Hvec[kd_idx[mid]].pos[dim];
        int l = kd_build(begin, mid), r = kd_build(mid, end);
        kd[id].left = l; kd[id].right = r; kd[id].dim = dim; kd[id].split = split;
        return id;
    }
    void build_kd() {
        kd.clear();
        kd_idx.resize(Hvec.size());
        for (size_t i = 0; i < Hvec.size(); i++) kd_idx[i] = (int)i;
        if (!Hvec.empty()) kd_build(0, (int)Hvec.size());
    }
    void kd_search(int node, const V3& q, double r2, std::vector<int>& out) const {
        const KdNode& nd = kd[node];
        if (nd.left < 0) {
            for (int i = nd.begin; i < nd.end; i++) {
                V3 v = Hvec[kd_idx[i]].pos - q;
                if (dot(v, v) < r2) out.push_back(kd_idx[i]);
            }
            return;
        }
        double d = q[nd.dim] - nd.split;
        int first = d < 0 ? nd.left : nd.right, second = d < 0 ? nd.right : nd.left;
        kd_search(first, q, r2, out);
        if (d * d <= r2) kd_search(second, q, r2, out);
    }

    // One candidate returned by the radius search (flann keeps dist < radius strictly,
    // with radius passed as float, Raytracer.h:146): the body of Raytracer.h:151-158.
    inline void gather_one(HitPoint& hp, const V3& x, const V3& N, const V3& flux) {
        V3 v = hp.pos - x;
        double d2 = dot(v, v);
        if (!(d2 < (double)(float)INIT_R2_)) return;
        cnt.candidates++;
        if ((dot(hp.n, N) > 1e-3) && (d2 <= hp.R2)) {
            hp.newN++;
            hp.tao = hp.tao + mul(hp.wgt, flux) * PI_INV_;
            cnt.accepted++;
        }
    }
    std::vector<int> kd_result;
    void gather(const V3& x, const V3& N, const V3& flux, int mode) {
        cnt.deposits++;
        if (mode == ORACLE_GATHER_BRUTE) {
            for (auto& hp : Hvec) gather_one(hp, x, N, flux);
        } else if (mode == ORACLE_GATHER_KDTREE) {
            kd_result.clear();
            if (!kd.empty()) kd_search(0, x, (double)(float)INIT_R2_, kd_result);
            for (int e : kd_result) gather_one(Hvec[e], x, N, flux);
        } else {
            const double cs = 1.5;
            int64_t bx = (int64_t)std::floor(x.x / cs), by = (int64_t)std::floor(x.y / cs), bz = (int64_t)std::floor(x.z / cs);
            for (int64_t dx = -1; dx <= 1; dx++)
                for (int64_t dy = -1; dy <= 1; dy++)
                    for (int64_t dz = -1; dz <= 1; dz++) {
                        V3 q((bx + dx + 0.5) * cs, (by + dy + 0.5) * cs, (bz + dz + 0.5) * cs);
                        auto it = grid.find(cell_key(q, cs));
                        if (it == grid.end()) continue;
                        for (int e : it->second) gather_one(Hvec[e], x, N, flux);
                    }
        }
    }

    // ---- Raytracer.h:117-209 PhotonTrace (tail recursion written as a loop)
    template <class F>
    void PhotonTrace(V3 org, V3 dir, V3 flux, uint64_t seed, uint32_t replica, uint32_t photon, F&& deposit) {
        int dep = 0;
        for (;;) {
            Rand rnd = make_rand(seed, replica, D_PHOTON, photon, (uint32_t)dep);
            if (dep <= MAX_DEP_) cnt.segments++;
            dir = narrow(dir);
            Collision c1 = NearCollide(org, dir, rnd);
            c1.N = narrow(c1.N);
            if (!c1.IsValid()) return;
            if (dep > MAX_DEP_) return;
            int vertex = dep;
            dep++;
            const Object& obj = objs[c1.obj];
            if (obj.material.IsDiff()) deposit(c1.Pos, c1.N, flux, vertex);
            int Next = Roulette(obj, rnd);
            V3 ndir;
            if (Next == 0) {
                double u1 = rnd.randu(0., 1., P_DIFFUSE, 0);
                double u2 = rnd.randu(0., 1., P_DIFFUSE, 1);
                ndir = RandReflFrom(c1.N, u1, u2);
            } else if (Next == 1) {
                ndir = GetUnitRefl(c1.From, c1.N);
            } else {
                double refrn = c1.IsFront() ? 1 / obj.material.refrn : obj.material.refrn;
                V3 refr = refracted(c1.From, c1.IsFront() ? c1.N : -c1.N, refrn);
                ndir = Normalized(refr);
            }
            flux = mul(c1.color, flux);
            org = c1.Pos;
            dir = ndir;
        }
    }

    // Light::emit, Light.h:8-13
    void emit(const Light& l, const Rand& rnd, V3& org, V3& dir, V3& flux) const {
        flux = l.color * PI_ * 4;
        dir = GetUnitRandVec(rnd);
        org = l.pos;
    }

    template <class F>
    void run_photons(uint64_t rounds, uint32_t per_round, uint64_t begin, uint64_t count, uint64_t seed,
                     uint32_t replica, bool sweep, F&& deposit) {
        uint64_t nl = lights.size();
        uint64_t total = rounds * nl * per_round;
        uint64_t end = (count == 0) ? total : std::min(total, begin + count);
        uint64_t per = nl * per_round;
        for (uint64_t g = begin; g < end; g++) {
            const Light& l = lights[(g / per_round) % nl];
            Rand rnd = make_rand(seed, replica, D_PHOTON, (uint32_t)g, 0);
            V3 org, dir, flux;
            emit(l, rnd, org, dir, flux);
            cnt.photons++;
            PhotonTrace(org, dir, flux, seed, replica, (uint32_t)g, deposit);
            const uint64_t cadence = initial_n > 0 ? per * rounds_per_update : per;
            if ((sweep || initial_n > 0) && cadence && ((g + 1) % cadence == 0 || (initial_n > 0 && g + 1 == end))) {   // Raytracer.h:237-245
                double maxr2 = 0;
                for (auto& hp : Hvec) {
                    hp.update();
                    if (hp.R2 > maxr2) maxr2 = hp.R2;
                }
                (void)maxr2;  // updateR2(maxr2) is commented out in the reference (Raytracer.h:246)
            }
        }
    }
};

// =================================================================== C interface
static V3 v3(const double* p) { return V3(p[0], p[1], p[2]); }
static void put3(double* o, const V3& v) { o[0] = v.x; o[1] = v.y; o[2] = v.z; }

extern "C" {

int oracle_set_radius_update(oracle_ctx* ctx, int32_t initial_n, uint32_t rounds_per_update) {
    if (!ctx || initial_n < 0 || (initial_n > 0 && rounds_per_update == 0)) return SPPM_E_INVALID;
    ctx->initial_n = initial_n;
    ctx->rounds_per_update = rounds_per_update ? rounds_per_update : 1;
    return SPPM_OK;
}

int oracle_set_narrow_rays(oracle_ctx* ctx, int on) {
    if (!ctx) return SPPM_E_INVALID;
    ctx->narrow_rays = on != 0;
    return SPPM_OK;
}

oracle_ctx* oracle_create(int newton_fp32, int rng_literal) {
    oracle_ctx* c = new oracle_ctx();
    c->newton_fp32 = newton_fp32 != 0;
    c->rng_literal = rng_literal != 0;
    return c;
}
void oracle_destroy(oracle_ctx* ctx) { delete ctx; }

int oracle_set_scene(oracle_ctx* ctx, const sppm_scene* s) {
    if (!ctx || !s) return SPPM_E_INVALID;
    ctx->objs.clear(); ctx->patches.clear(); ctx->textures.clear(); ctx->lights.clear();
    for (int i = 0; i < s->n_patches; i++) {
        Patch p;
        for (int k = 0; k < 16; k++) {
            p.P[k] = v3(s->patches + (size_t)i * 48 + k * 3);
            p.Pf[k] = {(float)p.P[k].x, (float)p.P[k].y, (float)p.P[k].z};
            p.aabb.addpoint(p.P[k]);                                     // Bezier3::read, Bezier.h:167-175
        }
        ctx->patches.push_back(p);
    }
    for (int i = 0; i < s->n_textures; i++) {
        Texture t;
        t.rows = s->textures[i].rows; t.cols = s->textures[i].cols;
        t.m_map.resize((size_t)t.rows * t.cols);
        for (size_t k = 0; k < t.m_map.size(); k++) {                    // Element.h:55-57
            const uint8_t* px = s->textures[i].texels + 3 * k;
            t.m_map[k] = V3(px[0] / 255.0, px[1] / 255.0, px[2] / 255.0);
        }
        ctx->textures.push_back(std::move(t));
    }
    for (int i = 0; i < s->n_objects; i++) {
        const sppm_object& so = s->objects[i];
        Object o;
        o.type = so.type; o.texture = so.texture; o.color = v3(so.color);
        o.material.diff = v3(so.material.diff); o.material.refl = v3(so.material.refl); o.material.refr = v3(so.material.refr);
        o.material.refrn = so.material.refrn; o.material.refln = so.material.refln;
        o.diffr = GetPower(o.material.diff); o.reflr = GetPower(o.material.refl); o.refrr = GetPower(o.material.refr);
        o.allr = o.diffr + o.reflr + o.refrr;
        o.P0 = v3(so.p0); o.norm = v3(so.n); o.radius = so.radius;
        o.first_patch = so.first_patch; o.n_patches = so.n_patches;
        if (o.type == SPPM_OBJ_BEZIER) {
            if (o.first_patch < 0 || o.first_patch + o.n_patches > s->n_patches) return SPPM_E_INVALID;
            for (int p = 0; p < o.n_patches; p++)
                for (int k = 0; k < 16; k++) o.aabb.addpoint(ctx->patches[o.first_patch + p].P[k]);  // Bezier.h:226-231
        }
        if (o.texture >= s->n_textures) return SPPM_E_INVALID;
        ctx->objs.push_back(o);
    }
    for (int i = 0; i < s->n_lights; i++) ctx->lights.push_back({v3(s->lights[i].pos), v3(s->lights[i].color)});
    return SPPM_OK;
}

void oracle_camera_emit(const sppm_camera* cam, int row, int col, double* org3, double* dir3) {
    // Camera::emit(y = row, x = col), Camera.h:18-22
    V3 du = v3(cam->du), dv = v3(cam->dv), dir = v3(cam->dir);
    V3 d = du * ((col + 0.5) / cam->w - 0.5) + dv * ((row + 0.5) / cam->h - 0.5) + dir;
    put3(org3, v3(cam->pos));
    put3(dir3, GetUnit(d));
}

int oracle_eye_pass_rows(oracle_ctx* ctx, const sppm_camera* cam, uint64_t seed, uint32_t replica, int row_begin,
                         int row_end) {
    if (!ctx || !cam || row_begin < 0 || row_end > cam->h || row_begin > row_end) return SPPM_E_INVALID;
    ctx->cam = *cam; ctx->have_cam = true;
    ctx->Hvec.clear();
    ctx->cnt = sppm_counters{};
    for (int i = row_begin; i < row_end; i++)                            // GetHitPoint, Raytracer.h:102-116
        for (int j = 0; j < cam->w; j++) {
            double o[3], d[3];
            oracle_camera_emit(cam, i, j, o, d);
            ctx->ViewTrace(v3(o), v3(d), 0, V3(1, 1, 1), i, j, 0, seed, replica);
        }
    ctx->cnt.hitpoints = ctx->Hvec.size();
    ctx->grid.clear(); ctx->kd.clear();
    return SPPM_OK;
}

int oracle_eye_pass(oracle_ctx* ctx, const sppm_camera* cam, uint64_t seed, uint32_t replica) {
    if (!ctx || !cam) return SPPM_E_INVALID;
    return oracle_eye_pass_rows(ctx, cam, seed, replica, 0, cam->h);
}

int oracle_set_hitpoints(oracle_ctx* ctx, const sppm_camera* cam, const sppm_hitpoint* in, uint64_t n) {
    if (!ctx || !cam || (n && !in)) return SPPM_E_INVALID;
    ctx->cam = *cam; ctx->have_cam = true;
    ctx->cnt = sppm_counters{};
    ctx->Hvec.resize(n);
    for (uint64_t i = 0; i < n; i++) {
        const sppm_hitpoint& s = in[i];
        HitPoint& h = ctx->Hvec[i];
        h.pos = v3(s.pos); h.n = v3(s.n); h.v = v3(s.v); h.wgt = v3(s.wgt); h.tao = v3(s.tao);
        h.R2 = s.R2; h.obj = s.obj; h.x = s.x; h.y = s.y; h.node = s.node; h.N = s.N; h.newN = s.newN;
    }
    ctx->cnt.hitpoints = n;
    ctx->grid.clear(); ctx->kd.clear();
    return SPPM_OK;
}

int oracle_photon_pass(oracle_ctx* ctx, uint64_t rounds, uint32_t per_round, uint64_t begin, uint64_t count,
                       uint64_t seed, uint32_t replica, int mode) {
    if (!ctx || !ctx->have_cam) return SPPM_E_INVALID;
    if (mode == ORACLE_GATHER_GRID && ctx->grid.empty()) ctx->build_grid();
    if (mode == ORACLE_GATHER_KDTREE && ctx->kd.empty()) ctx->build_kd();  // Raytracer.h:370-380
    ctx->last_rounds = rounds; ctx->last_per_round = per_round;
    ctx->run_photons(rounds, per_round, begin, count, seed, replica, mode == ORACLE_GATHER_KDTREE,
                     [&](const V3& x, const V3& N, const V3& flux, int) { ctx->gather(x, N, flux, mode); });
    return SPPM_OK;
}

int oracle_photon_deposits(oracle_ctx* ctx, uint64_t rounds, uint32_t per_round, uint64_t begin, uint64_t count,
                           uint64_t seed, uint32_t replica, double* out, uint64_t cap, uint64_t* n) {
    if (!ctx) return SPPM_E_INVALID;
    uint64_t k = 0;
    uint32_t cur = 0;
    // run_photons passes the photon index only implicitly; recover it from the emission order
    uint64_t nl = ctx->lights.size();
    uint64_t total = rounds * nl * per_round;
    uint64_t end = (count == 0) ? total : std::min(total, begin + count);
    for (uint64_t g = begin; g < end; g++) {
        cur = (uint32_t)g;
        ctx->run_photons(rounds, per_round, g, 1, seed, replica, false,
                         [&](const V3& x, const V3& N, const V3& flux, int vertex) {
                             if (k < cap) {
                                 double* o = out + k * 11;
                                 o[0] = (double)cur; o[1] = (double)vertex;
                                 put3(o + 2, x); put3(o + 5, N); put3(o + 8, flux);
                             }
                             k++;
                         });
    }
    *n = k;
    return k <= cap ? SPPM_OK : SPPM_E_CAPACITY;
}

int oracle_resolve(oracle_ctx* ctx, double* pic, int accumulate) {
    if (!ctx || !ctx->have_cam) return SPPM_E_INVALID;
    size_t npx = (size_t)ctx->cam.w * ctx->cam.h;
    if (!accumulate) std::fill(pic, pic + npx * 3, 0.0);                 // Raytracer.h:281-287
    double cnt = (double)ctx->last_rounds, onetime = (double)ctx->last_per_round;
    for (auto& hp : ctx->Hvec) {                                         // Raytracer.h:288-293
        V3 add = hp.tao * (1.0 / PI_ / (hp.R2 * cnt * onetime));
        double* p = pic + ((size_t)hp.x * ctx->cam.w + hp.y) * 3;
        p[0] += add.x; p[1] += add.y; p[2] += add.z;
    }
    return SPPM_OK;
}

int oracle_get_hitpoints(oracle_ctx* ctx, sppm_hitpoint* out, uint64_t cap, uint64_t* n) {
    if (!ctx) return SPPM_E_INVALID;
    *n = ctx->Hvec.size();
    for (uint64_t i = 0; i < std::min<uint64_t>(cap, *n); i++) {
        const HitPoint& h = ctx->Hvec[i];
        sppm_hitpoint& o = out[i];
        put3(o.pos, h.pos); put3(o.n, h.n); put3(o.v, h.v); put3(o.wgt, h.wgt); put3(o.tao, h.tao);
        o.R2 = h.R2; o.obj = h.obj; o.x = h.x; o.y = h.y; o.node = h.node; o.N = h.N; o.newN = h.newN;
    }
    return SPPM_OK;
}
int oracle_get_counters(oracle_ctx* ctx, sppm_counters* out) {
    if (!ctx) return SPPM_E_INVALID;
    *out = ctx->cnt;
    out->hitpoints = ctx->Hvec.size();
    return SPPM_OK;
}

int oracle_near_collide(oracle_ctx* ctx, uint64_t n, const double* org, const double* dir, uint64_t seed,
                        uint32_t replica, sppm_collision* out) {
    if (!ctx) return SPPM_E_INVALID;
    for (uint64_t i = 0; i < n; i++) {
        Rand rnd = ctx->make_rand(seed, replica, D_EYE, (uint32_t)i, 0);
        Collision c = ctx->NearCollide(v3(org + 3 * i), v3(dir + 3 * i), rnd);
        put3(out[i].pos, c.Pos); put3(out[i].n, c.N); put3(out[i].color, c.color);
        out[i].dist = c.dist; out[i].obj = c.obj; out[i].inside = c.inside ? 1 : 0;
    }
    return SPPM_OK;
}

int oracle_newton(oracle_ctx* ctx, uint64_t n, const double* org, const double* dir, const int32_t* patch,
                  uint64_t seed, uint32_t replica, double* out_tuv) {
    if (!ctx) return SPPM_E_INVALID;
    for (uint64_t i = 0; i < n; i++) {
        if (patch[i] < 0 || patch[i] >= (int)ctx->patches.size()) return SPPM_E_INVALID;
        const Patch& pt = ctx->patches[patch[i]];
        Rand rnd = ctx->make_rand(seed, replica, D_EYE, (uint32_t)i, 0);
        V3 r;
        if (ctx->newton_fp32) r = newton<float>(pt.Pf, v3(org + 3 * i), v3(dir + 3 * i), MAX_ITER_, rnd, (uint32_t)patch[i]);
        else {
            T3<double> Pd[16];
            for (int k = 0; k < 16; k++) Pd[k] = {pt.P[k].x, pt.P[k].y, pt.P[k].z};
            r = newton<double>(Pd, v3(org + 3 * i), v3(dir + 3 * i), MAX_ITER_, rnd, (uint32_t)patch[i]);
        }
        put3(out_tuv + 3 * i, r);
    }
    return SPPM_OK;
}

// ---- unit hooks
void oracle_bezier_eval(const double* P48, double u, double v, double* out9) {
    T3<double> P[16];
    for (int k = 0; k < 16; k++) P[k] = {P48[3 * k], P48[3 * k + 1], P48[3 * k + 2]};
    T3<double> s = evalPatchPoint(P, u, v), su = evalPatchDerivU(P, u, v), sv = evalPatchDerivV(P, u, v);
    out9[0] = s.x; out9[1] = s.y; out9[2] = s.z;
    out9[3] = su.x; out9[4] = su.y; out9[5] = su.z;
    out9[6] = sv.x; out9[7] = sv.y; out9[8] = sv.z;
}
int oracle_aabb_intersect(const double* pmin, const double* pmax, const double* org, const double* dir) {
    AABB b; b.pmin = v3(pmin); b.pmax = v3(pmax);
    return b.intersect(v3(org), v3(dir)) ? 1 : 0;
}
void oracle_refracted(const double* I, const double* N, double n, double* out3) { put3(out3, refracted(v3(I), v3(N), n)); }
void oracle_reflected_unit(const double* I, const double* N, double* out3) { put3(out3, GetUnitRefl(v3(I), v3(N))); }
void oracle_rotated(const double* v, const double* axis, double angle, double* out3) { put3(out3, rotated(v3(v), v3(axis), angle)); }
void oracle_rand_refl(const double* N, double u1, double u2, double* out3) { put3(out3, RandReflFrom(v3(N), u1, u2)); }
int oracle_to_int(double x) { return int(std::pow(1 - std::exp(-x), 1 / 2.2) * 255 + .5); }  // Raytracer.h:24-26
// The byte image SPPMRayTracer::render writes (Raytracer.h:460-466): m1.ptr<uchar>(h - 1 - i1)[j * 3 + k] =
// toInt(pic[i1][j][k] / n_images); rgb != 0 stores channel k at byte 2 - k instead (the same picture as a PPM).
void oracle_tonemap(const double* pic_hw3, int w, int h, double n_images, int rgb, uint8_t* out_hw3) {
    for (int i1 = 0; i1 < h; i1++)
        for (int j = 0; j < w; j++)
            for (int k = 0; k < 3; k++)
                out_hw3[((size_t)(h - 1 - i1) * w + j) * 3 + (rgb ? 2 - k : k)] =
                    (uint8_t)oracle_to_int(pic_hw3[((size_t)i1 * w + j) * 3 + k] / n_images);
}
void oracle_hitpoint_update(double* R2, int32_t* N, double* tao3, int32_t* newN) {
    HitPoint h; h.R2 = *R2; h.N = *N; h.tao = v3(tao3); h.newN = *newN;
    h.update();
    *R2 = h.R2; *N = h.N; put3(tao3, h.tao); *newN = h.newN;
}
void oracle_mwc_kat(int n, uint32_t* out_u32, double* out_uniform01) {
    MWC a, b;
    for (int i = 0; i < n; i++) { out_u32[i] = a.next(); out_uniform01[i] = b.uniform01(); }
}
uint32_t oracle_rng_u32(uint64_t seed, uint32_t stream, uint32_t path, uint32_t sub, uint32_t idx) {
    return rng_u32(seed, stream, path, sub, idx);
}
void oracle_texture_color_uv(const sppm_texture* tex, double u, double v, double* out3) {
    Texture t; t.rows = tex->rows; t.cols = tex->cols; t.m_map.resize((size_t)t.rows * t.cols);
    for (size_t k = 0; k < t.m_map.size(); k++) t.m_map[k] = V3(tex->texels[3 * k] / 255.0, tex->texels[3 * k + 1] / 255.0, tex->texels[3 * k + 2] / 255.0);
    put3(out3, t.colorUV(u, v));
}
void oracle_camera_lookat(const double* pos, const double* look, int w, int h, sppm_camera* out) {
    // Camera::lookAt(P, 1.), Camera.h:32-45
    V3 p = v3(pos), l = v3(look);
    const V3 up(0, 0, 1);
    V3 dir = GetUnit(l - p);
    V3 du = Normalized(cross(dir, up));
    V3 dv = Normalized(-cross(dir, du));
    double fov = 50. * PI_ / 180.;
    dir = dir * (.5 / std::tan(fov / 2.));
    put3(out->pos, p); put3(out->dir, dir); put3(out->du, du); put3(out->dv, dv);
    out->w = w; out->h = h;
}
int oracle_plane_intersect(const sppm_object* o, const double* org, const double* dir, double* dist) {
    oracle_ctx c; Object ob; ob.P0 = v3(o->p0); ob.norm = v3(o->n);
    return c.plane_hit(ob, v3(org), v3(dir), *dist) ? 1 : 0;
}
int oracle_sphere_intersect(const sppm_object* o, const double* org, const double* dir, double* dist, int* inside) {
    oracle_ctx c; Object ob; ob.P0 = v3(o->p0); ob.radius = o->radius;
    bool in = false;
    bool hit = c.sphere_hit(ob, v3(org), v3(dir), *dist, in);
    *inside = in ? 1 : 0;
    return hit ? 1 : 0;
}

}  // extern "C"
