// Bezier.h — drop-in for the reference's raytracer/Bezier.h: AABBbox, Bezier3 (one bicubic
// patch) and Bezier3Obj (a .bpt file of patches) with the reference's constructor
// (raytracer/Bezier.h:199).  The patches are held as control points; the AABB culls and the
// Newton ray-patch solve (Bezier.h:20-51, :112-159) run on the GPU.
#pragma once
#include <fstream>
#include <string>
#include <vector>

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
};

class Bezier3 {
public:
    Vec3 P[16];
    AABBbox aabb;
    void read(const Vec3 p[16]) {
        aabb.clear();
        for (int i = 0; i < 16; i++) { P[i] = p[i]; aabb.addpoint(P[i]); }
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
