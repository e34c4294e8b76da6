// Light.h — drop-in for the reference's raytracer/Light.h: isotropic point lights.  The
// photon emission of Light::emit (raytracer/Light.h:8-13) happens on the GPU from pos / color.
#pragma once
#include "Element.h"
#include "Vec3.h"

class Light {
public:
    Light(Vec3 p, Color c) : pos(p), color(c) {}
    virtual ~Light() {}
    virtual void emit(Ray& ray, Vec3& flux) {
        flux = color * PI * 4;
        ray.dir = Vec3::GetUnitRandVec();
        ray.org = pos;
    }
    const Vec3& position() const { return pos; }
    const Color& power() const { return color; }

private:
    Vec3 pos;
    Color color;
};

class SpotLight : public Light {
public:
    SpotLight(Vec3 p, Color c) : Light(p, c) {}
};
