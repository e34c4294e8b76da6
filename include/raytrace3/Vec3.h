// Vec3.h — drop-in for the reference's raytracer/Vec3.h (host side, no OpenCV).
//
// Same names and meaning as the reference's fp64 vector type and helpers
// (raytracer/Vec3.h:5-155): Vec3 with val[3], dot/cross/mul, Module/Normalize/GetUnit,
// reflection, Rodrigues rotation, Snell refraction, GetRandu backed by a
// multiply-with-carry generator equal to cv::RNG (seed 0xffffffff).  The host only
// uses these for scene set-up and camera jitter; the per-ray work runs on the GPU.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <iostream>

#define M_EPS 1e-4
#define RAND_R 10000000
#define PI 3.1415926
#define PI_INV 0.3183098862
#define Color Vec3
#define MAX_NUM 1e20
#define MIN_NUM -1e20
#define MAX_VEC3 (Vec3(MAX_NUM, MAX_NUM, MAX_NUM))
#define MIN_VEC3 (Vec3(MIN_NUM, MIN_NUM, MIN_NUM))

namespace sppm_host {
// cv::RNG's recurrence: state = (u32)state * 4164903690 + (state >> 32).
struct MwcRng {
    uint64_t state = 0xffffffffULL;
    uint32_t next() {
        state = (uint64_t)(uint32_t)state * 4164903690U + (uint32_t)(state >> 32);
        return (uint32_t)state;
    }
    double uniform(double a, double b) {
        uint32_t hi = next();
        uint64_t bits = ((uint64_t)hi << 32) | next();
        return (double)bits * 5.4210108624275221700372640043497e-20 * (b - a) + a;
    }
    int uniform(int a, int b) { return a == b ? a : (int)(next() % (unsigned)(b - a) + a); }
};
inline MwcRng& global_rng() {
    static MwcRng r;
    return r;
}
}  // namespace sppm_host

inline double GetRandu(double r1, double r2) { return sppm_host::global_rng().uniform(r1, r2); }
inline double GetRandu() { return GetRandu((double)-RAND_R, (double)RAND_R); }
inline int GetRandu(int i, int j) { return sppm_host::global_rng().uniform(i, j); }
inline bool IsZero(double i) { return std::fabs(i) < M_EPS; }

class Vec3 {
public:
    double val[3];
    Vec3() : val{0, 0, 0} {}
    Vec3(double x, double y, double z) : val{x, y, z} {}
    double& operator()(int i) { return val[i]; }
    double operator()(int i) const { return val[i]; }
    double& operator[](int i) { return val[i]; }
    double operator[](int i) const { return val[i]; }

    double dot(const Vec3& o) const { return val[0] * o.val[0] + val[1] * o.val[1] + val[2] * o.val[2]; }
    Vec3 cross(const Vec3& o) const {
        return Vec3(val[1] * o.val[2] - val[2] * o.val[1], val[2] * o.val[0] - val[0] * o.val[2],
                    val[0] * o.val[1] - val[1] * o.val[0]);
    }
    Vec3 mul(const Vec3& o) const { return Vec3(val[0] * o.val[0], val[1] * o.val[1], val[2] * o.val[2]); }

    Vec3& operator+=(const Vec3& o) { for (int i = 0; i < 3; i++) val[i] += o.val[i]; return *this; }
    Vec3& operator-=(const Vec3& o) { for (int i = 0; i < 3; i++) val[i] -= o.val[i]; return *this; }
    Vec3& operator*=(double s) { for (int i = 0; i < 3; i++) val[i] *= s; return *this; }
    Vec3& operator+=(double s) { for (int i = 0; i < 3; i++) val[i] += s; return *this; }
    Vec3& operator-=(double s) { for (int i = 0; i < 3; i++) val[i] -= s; return *this; }
    Vec3 operator+(double s) const { return Vec3(val[0] + s, val[1] + s, val[2] + s); }
    Vec3 operator-(double s) const { return Vec3(val[0] - s, val[1] - s, val[2] - s); }
    Vec3 operator-() const { return Vec3(-val[0], -val[1], -val[2]); }
    bool operator<(const Vec3& v) const { return val[0] < v.val[0] && val[1] < v.val[1] && val[2] < v.val[2]; }
    bool operator<(const double& k) const { return val[0] < k && val[1] < k && val[2] < k; }
    bool operator>(const Vec3& v) const { return val[0] > v.val[0] && val[1] > v.val[1] && val[2] > v.val[2]; }

    double Module2() const { return dot(*this); }
    double Module() const { return std::sqrt(Module2()); }
    Vec3& Normalize() {
        double m = Module();
        if (!IsZero(m)) *this *= 1 / m;
        return *this;
    }
    Vec3 GetUnit() const {
        double m = Module();
        return IsZero(m) ? *this : Vec3(val[0] * (1.0 / m), val[1] * (1.0 / m), val[2] * (1.0 / m));
    }
    static Vec3 GetUnitRandVec() {
        for (;;) {
            double x = GetRandu(-1., 1.), y = GetRandu(-1., 1.), z = GetRandu(-1., 1.);
            double m = x * x + y * y + z * z;
            if (m > 1 || m < M_EPS) continue;
            return Vec3(x, y, z).GetUnit();
        }
    }
    Vec3 GetRefl(const Vec3& norm) const;
    Vec3 GetUnitRefl(const Vec3& norm) const;
    Vec3 GetAnormal() const { return (val[0] == 0 && val[1] == 0) ? Vec3(1, 0, 0) : Vec3(val[1], -val[0], 0).GetUnit(); }
    Vec3 rotated(const Vec3& axis, double angle) const;
    static Vec3 GetUnitRandRefl(const Vec3& norm);
    double GetPower() const { return (val[0] + val[1] + val[2]) / 3; }
    Vec3 refracted(const Vec3& N, double n) const;
    Vec3 refracted(const Vec3& N, double n_from, double n_to) const { return refracted(N, n_from / n_to); }
    double GetInfNorm() const { return std::max(std::fabs(val[0]), std::max(std::fabs(val[1]), std::fabs(val[2]))); }
    bool isnan() const { return std::isnan(val[0]) || std::isnan(val[1]) || std::isnan(val[2]); }
};

inline Vec3 operator+(const Vec3& a, const Vec3& b) { return Vec3(a.val[0] + b.val[0], a.val[1] + b.val[1], a.val[2] + b.val[2]); }
inline Vec3 operator-(const Vec3& a, const Vec3& b) { return Vec3(a.val[0] - b.val[0], a.val[1] - b.val[1], a.val[2] - b.val[2]); }
inline Vec3 operator*(const Vec3& a, double s) { return Vec3(a.val[0] * s, a.val[1] * s, a.val[2] * s); }
inline Vec3 operator*(double s, const Vec3& a) { return a * s; }
inline Vec3 operator/(const Vec3& a, double s) { return Vec3(a.val[0] / s, a.val[1] / s, a.val[2] / s); }
inline std::ostream& operator<<(std::ostream& os, const Vec3& v) { return os << "[" << v.val[0] << ", " << v.val[1] << ", " << v.val[2] << "]"; }
inline double dist2(const Vec3& a, const Vec3& b) { return (a - b).Module2(); }
inline double dist(const Vec3& a, const Vec3& b) { return std::sqrt(dist2(a, b)); }
inline void Confine(Vec3& v) { for (int i = 0; i < 3; i++) if (v.val[i] > 1) v.val[i] = 1; }
// True when ANY component is ~0 (this is what makes Material::IsDiff need all three channels).
inline bool IsZero(const Vec3& v) { return IsZero(v.val[0]) || IsZero(v.val[1]) || IsZero(v.val[2]); }

inline Vec3 Vec3::GetRefl(const Vec3& norm) const { return *this - norm * (2 * dot(norm)); }
inline Vec3 Vec3::GetUnitRefl(const Vec3& norm) const { return GetRefl(norm).GetUnit(); }
inline Vec3 Vec3::rotated(const Vec3& a, double angle) const {
    if (std::fabs(angle) < M_EPS) return *this;
    const double c = std::cos(angle), s = std::sin(angle), k = 1 - c;
    const double x = a.val[0], y = a.val[1], z = a.val[2];
    return Vec3(val[0] * (x * x + (1 - x * x) * c) + val[1] * (x * y * k - z * s) + val[2] * (x * z * k + y * s),
                val[0] * (y * x * k + z * s) + val[1] * (y * y + (1 - y * y) * c) + val[2] * (y * z * k - x * s),
                val[0] * (z * x * k - y * s) + val[1] * (z * y * k + x * s) + val[2] * (z * z + (1 - z * z) * c));
}
inline Vec3 Vec3::GetUnitRandRefl(const Vec3& norm) {
    double theta = std::acos(std::sqrt(GetRandu(0., 1.)));
    double phi = 2 * PI * GetRandu(0., 1.);
    return norm.rotated(norm.GetAnormal(), theta).rotated(norm, phi).GetUnit();
}
inline Vec3 Vec3::refracted(const Vec3& N, double n) const {
    double cosI = -N.dot(*this);
    double cosR2 = 1 - (1 - cosI * cosI) * n * n;
    return (cosR2 > M_EPS) ? *this * n + N * (n * cosI - std::sqrt(cosR2)) : GetRefl(N);
}
