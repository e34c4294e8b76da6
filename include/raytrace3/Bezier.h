// Bezier.h — drop-in for the reference's raytracer/Bezier.h: AABBbox, Bezier3 (one bicubic
// patch) and Bezier3Obj (a .bpt file of patches) with the reference's constructor
// (raytracer/Bezier.h:199).  The patches are held as control points; the AABB culls and the
// Newton ray-patch solve (Bezier.h:20-51, :112-159) run on the GPU.  The members a host may call itself are here as
// well: AABBbox::intersect and the Bernstein evaluators are a few lines of host arithmetic, Bezier3::newton / intersect
// hand the solve to the GPU library.
#pragma once
#include <atomic>
#include <fstream>
#include <stdexcept>
#include <string>
#include <vector>

#include "Bridge.h"
#include "Element.h"
#include "Obj.h"
#include "Vec3.h"
#define MAX_ITER 10

// Minimal 3x3 matrix standing in for cv::Matx33d (zero-initialised, (i,j) access, products).
struct Mat33 {
    double m[3][3];
    Mat33() : m{{0, 0, 0}, {0, 0, 0}, {0, 0, 0}} {}
    static Mat33 eye() { Mat33 r; r.m[0][0] = r.m[1][1] = r.m[2][2] = 1; return r; }
    double& operator()(int i, int j) { return m[i][j]; }
    double operator()(int i, int j) const { return m[i][j]; }
    Mat33 operator*(const Mat33& o) const {
        Mat33 r;
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) for (int k = 0; k < 3; k++) r.m[i][j] += m[i][k] * o.m[k][j];
        return r;
    }
    Vec3 operator*(const Vec3& v) const {
        return Vec3(m[0][0] * v.val[0] + m[0][1] * v.val[1] + m[0][2] * v.val[2], m[1][0] * v.val[0] + m[1][1] * v.val[1] + m[1][2] * v.val[2],
                    m[2][0] * v.val[0] + m[2][1] * v.val[1] + m[2][2] * v.val[2]);
    }
};
namespace cv { typedef ::Mat33 Matx33d; }

class AABBbox {
public:
    Vec3 pmin = MAX_VEC3, pmax = MIN_VEC3;
    void addpoint(const Vec3& p) {
        for (int i = 0; i < 3; i++) {
            pmin.val[i] = std::min(pmin.val[i], p.val[i]);
            pmax.val[i] = std::max(pmax.val[i], p.val[i]);
        }
    }
    void clear() { pmin = MAX_VEC3; pmax = MIN_VEC3; }
    // The reference's box test (raytracer/Bezier.h:20-51): a ray meets the box if, on some axis along which it moves
    // (|dir| >= 1e-4 there), it reaches the face it approaches from outside (pmin going up, pmax going down) at a
    // positive parameter, at a point that lies within the box's extent on the two other axes.  Only ENTRY faces are
    // asked, so a ray that starts inside the box misses it.
    bool intersect(const Vec3& org, const Vec3& dir) const {
        for (int axis = 0; axis < 3; axis++) {
            const double d = dir.val[axis];
            if (IsZero(d)) continue;
            const double face = d > 0 ? pmin.val[axis] : pmax.val[axis];
            const double t = (face - org.val[axis]) / d;
            if (!(t > 0)) continue;
            const Vec3 at = org + t * dir;
            bool within = true;
            for (int other = 0; other < 3; other++)
                if (other != axis && (at.val[other] > pmax.val[other] || at.val[other] < pmin.val[other])) within = false;
            if (within) return true;
        }
        return false;
    }
};

class Bezier3 {
    // cubic Bernstein weights and their derivatives (raytracer/Bezier.h:71-74, :79-82)
    static void weights(double t, double w[4]) {
        const double s = 1 - t;
        w[0] = s * s * s; w[1] = 3 * t * s * s; w[2] = 3 * t * t * s; w[3] = t * t * t;
    }
    static void weights_deriv(double t, double w[4]) {
        const double s = 1 - t;
        w[0] = -3 * s * s; w[1] = 3 * s * s - 6 * t * s; w[2] = 6 * t * s - 3 * t * t; w[3] = 3 * t * t;
    }
    static Vec3 blend(const Vec3& a, const Vec3& b, const Vec3& c, const Vec3& d, const double w[4]) {
        return a * w[0] + b * w[1] + c * w[2] + d * w[3];
    }

public:
    Vec3 P[16];
    AABBbox aabb;
    void read(const Vec3 p[16]) {
        aabb.clear();
        for (int i = 0; i < 16; i++) { P[i] = p[i]; aabb.addpoint(P[i]); }
    }

    // One cubic Bezier curve and its derivative (raytracer/Bezier.h:69-84).
    static Vec3 evalCurvePoint(const Vec3 P[4], double t) { double w[4]; weights(t, w); return blend(P[0], P[1], P[2], P[3], w); }
    static Vec3 evalCurveDeriv(const Vec3 P[4], double t) { double w[4]; weights_deriv(t, w); return blend(P[0], P[1], P[2], P[3], w); }
    // The patch S(u, v) and its partial derivatives (raytracer/Bezier.h:85-111): u runs along the four points of a row
    // P[4 i .. 4 i + 3], v across the rows.  The order of evaluation is the reference's (rows in u first for S and S_v,
    // columns in v first for S_u), so the values are the reference's to the last bit.
    static Vec3 evalPatchPoint(const Vec3 P[16], double u, double v) {
        double wu[4], wv[4];
        weights(u, wu); weights(v, wv);
        Vec3 row[4];
        for (int i = 0; i < 4; i++) row[i] = blend(P[4 * i], P[4 * i + 1], P[4 * i + 2], P[4 * i + 3], wu);
        return blend(row[0], row[1], row[2], row[3], wv);
    }
    static Vec3 evalPatchDerivU(const Vec3 P[16], double u, double v) {
        double du[4], wv[4];
        weights_deriv(u, du); weights(v, wv);
        Vec3 col[4];
        for (int i = 0; i < 4; i++) col[i] = blend(P[i], P[i + 4], P[i + 8], P[i + 12], wv);
        return blend(col[0], col[1], col[2], col[3], du);
    }
    static Vec3 evalPatchDerivV(const Vec3 P[16], double u, double v) {
        double wu[4], dv[4];
        weights(u, wu); weights_deriv(v, dv);
        Vec3 row[4];
        for (int i = 0; i < 4; i++) row[i] = blend(P[4 * i], P[4 * i + 1], P[4 * i + 2], P[4 * i + 3], wu);
        return blend(row[0], row[1], row[2], row[3], dv);
    }

    // Bezier3::newton (raytracer/Bezier.h:112-159): 5 maxIter restarts of maxIter Newton steps on (t, u, v), every iterate
    // within 0.01 of the surface with u, v in [0, 1] a candidate, the candidate of smallest t returned, MAX_VEC3 if none.
    // The solve runs on the GPU (sppm_debug_newton, on the context of slot 0, which is handed a scene of this one patch
    // and gets the caller's scene back at its next pass).  The reference draws the restarts' starting points from its
    // global generator, so two calls never use the same ones; here every call takes the counter-based streams of a call
    // number of its own.  The GPU solver is built for the reference's MAX_ITER = 10, the only value the reference passes.
    Vec3 newton(const Vec3& org, const Vec3& dir, int maxIter) const {
        if (maxIter != MAX_ITER) throw std::invalid_argument("Bezier3::newton: the GPU solver runs MAX_ITER = 10 iterations of 50 restarts");
        static std::atomic<uint32_t> calls{0};
        const uint32_t call = calls.fetch_add(1);
        sppm_object o;
        std::memset(&o, 0, sizeof(o));
        o.type = SPPM_OBJ_BEZIER; o.texture = -1; o.first_patch = 0; o.n_patches = 1;
        for (int k = 0; k < 3; k++) { o.color[k] = 1; o.material.diff[k] = 0.75; }
        o.material.refrn = 1.5; o.material.refln = 1;
        double cp[48];
        uint64_t tag = 0x62657a33ull;                        // not a tag Scene's flatten() could produce by chance: other salt, other length
        for (int k = 0; k < 16; k++)
            for (int c = 0; c < 3; c++) {
                cp[3 * k + c] = P[k].val[c];
                uint64_t bits;
                std::memcpy(&bits, &cp[3 * k + c], 8);
                tag = (tag ^ bits) * 1099511628211ull;
            }
        sppm_light light = {{0, 0, 0}, {1, 1, 1}};
        sppm_scene sc;
        std::memset(&sc, 0, sizeof(sc));
        sc.n_objects = 1; sc.objects = &o; sc.n_patches = 1; sc.patches = cp; sc.n_lights = 1; sc.lights = &light;
        sppm_host::Device& dev = sppm_host::Device::get(0);
        std::lock_guard<std::mutex> g(dev.mu);
        dev.use_scene(sc, tag | 1);
        const int32_t patch = 0;
        double tuv[3];
        dev.check(sppm_debug_newton(dev.ctx, 1, org.val, dir.val, &patch, 0x6e6577746f6eull, call, tuv), "sppm_debug_newton");
        return Vec3(tuv[0], tuv[1], tuv[2]);
    }
    // Bezier3::intersect (raytracer/Bezier.h:176-186): the solve only for a ray that meets the patch's box.
    Vec3 intersect(const Vec3& org, const Vec3& dir) const {
        return aabb.intersect(org, dir) ? newton(org, dir, MAX_ITER) : MAX_VEC3;
    }
};

class Bezier3Obj : public Obj {
    std::vector<Bezier3> Bvec;
    AABBbox aabb;

public:
    // File format (.bpt): patch count, then per patch "m n" and (m+1)(n+1) points; each
    // point becomes Trans * (scale * p) + C.  The reference copies the points into Bezier3::P[16] as they come
    // (raytracer/Bezier.h:205-224), which is a bicubic patch only for m = n = 3.  A patch of lower degree in u or v
    // is raised to degree 3 here (degree elevation: the same surface, exactly); a higher degree has no bicubic form
    // and the patch is skipped with a message.
    static void elevate(Vec3 q[4], int& deg) {       // one Bezier curve, degree deg -> deg + 1
        Vec3 r[4];
        r[0] = q[0];
        for (int i = 1; i <= deg; i++) {
            const double a = (double)i / (deg + 1);
            r[i] = q[i - 1] * a + q[i] * (1 - a);
        }
        r[deg + 1] = q[deg];
        deg++;
        for (int i = 0; i <= deg; i++) q[i] = r[i];
    }
    Bezier3Obj(const Vec3& C, double scale, const std::string& file, const Color& cc, const Material& mm,
               const Mat33& Trans = Mat33::eye())
        : Obj(mm, cc) {
        std::cerr << "read bezier3::" << file << std::endl;
        std::ifstream in(file);
        if (!in) in.open(sppm_host::data_dir() + "/" + file);
        if (!in) { std::cerr << "Bezier3Obj: cannot open " << file << std::endl; return; }
        int size = 0;
        in >> size;
        for (int i = 0; i < size && in; i++) {
            int m = 0, n = 0;
            in >> m >> n;
            if (m < 0 || n < 0) { std::cerr << "Bezier3Obj: bad degrees in patch " << i << std::endl; return; }
            if (m > 3 || n > 3) {
                std::cerr << "Bezier3Obj: patch " << i << " has degree " << m << " x " << n << " (more than bicubic), skipped" << std::endl;
                double skip; for (long k = 0; k < 3L * (m + 1) * (n + 1); k++) in >> skip; continue; }
            Vec3 g[4][4];                               // g[j][k]: k runs along u (the fast index of P[16])
            for (int j = 0; j <= m; j++)
                for (int k = 0; k <= n; k++) {
                    Vec3 p;
                    for (int l = 0; l < 3; l++) { in >> p.val[l]; p.val[l] *= scale; }
                    g[j][k] = Trans * p;
                }
            for (int j = 0; j <= m; j++) { int d = n; while (d < 3) elevate(g[j], d); }
            for (int k = 0; k < 4; k++) {
                Vec3 col[4];
                for (int j = 0; j <= m; j++) col[j] = g[j][k];
                int d = m; while (d < 3) elevate(col, d);
                for (int j = 0; j < 4; j++) g[j][k] = col[j];
            }
            Vec3 pts[16];
            for (int j = 0; j < 4; j++) for (int k = 0; k < 4; k++) pts[j * 4 + k] = g[j][k];
            for (int k = 0; k < 16; k++) { pts[k] += C; aabb.addpoint(pts[k]); }
            Bezier3 b;
            b.read(pts);
            Bvec.push_back(b);
        }
        std::cerr << "done!size::" << Bvec.size() << std::endl;
        std::cerr << "Bezier::aabb" << aabb.pmax << aabb.pmin << std::endl;
    }
    size_t patch_count() const { return Bvec.size(); }
    const Bezier3& patch(size_t i) const { return Bvec[i]; }
    void Describe(sppm_object& o, std::vector<double>& patches) const override {
        describe_common(o, SPPM_OBJ_BEZIER);
        o.first_patch = (int)(patches.size() / 48);
        o.n_patches = (int)Bvec.size();
        for (const Bezier3& b : Bvec)
            for (int k = 0; k < 16; k++)
                for (int c = 0; c < 3; c++) patches.push_back(b.P[k].val[c]);
    }
};
