// Obj.h — drop-in for the reference's raytracer/Obj.h: Obj / PlaneObj / SphereObj with the
// reference's constructors and public members (raytracer/Obj.h:5-154).  The geometry lives
// here only as data: every intersection of the render loop runs on the GPU, where
// Scene::NearCollide is evaluated from the flattened objects (Describe()).  GetIntersect on
// a single object is kept for API compatibility and is also answered by the GPU.
#pragma once
#include <vector>

#include "../sppm_abi.h"
#include "Element.h"
#include "Vec3.h"

class Obj;
namespace sppm_host {
Collision intersect_objects(const std::vector<Obj*>& objs, const Ray& ray);
}

class Obj {
protected:
    Texture* texture = nullptr;

public:
    Obj(const Material& mm, const Color& cc) : material(mm), color(cc) {
        diffr = mm.diff.GetPower();
        reflr = mm.refl.GetPower();
        refrr = mm.refr.GetPower();
        allr = diffr + reflr + refrr;
    }
    Material material;
    bool IsDiff() const { return material.IsDiff(); }
    bool IsRefl() const { return material.IsRefl(); }
    bool IsRefr() const { return material.IsRefr(); }
    void GetDiff(Vec3& v) const { v = material.diff; }
    void GetRefl(Vec3& v) const { v = material.refl; }
    void GetRefr(Vec3& v) const { v = material.refr; }
    void GetRefl(Vec3& v, double& rln) const { v = material.refl; rln = material.refln; }
    void GetRefr(Vec3& v, double& rrn) const { v = material.refr; rrn = material.refrn; }

    enum React { DIFF, REFL, REFR };
    double diffr, reflr, refrr, allr;
    React Roulette() {
        double r = GetRandu(0., allr);
        if (diffr > r) return DIFF;
        if (diffr + reflr > r) return REFL;
        return REFR;
    }
    Color color;
    virtual ~Obj() = 0;
    virtual Collision GetIntersect(const Ray& ray) {
        std::vector<Obj*> one{this};
        return sppm_host::intersect_objects(one, ray);
    }
    void setTexture(Texture* texture_) { texture = texture_; }
    Texture* getTexture() const { return texture; }

    // Fill the C-ABI record of this object; Bezier objects append their control points to `patches`.
    virtual void Describe(sppm_object& o, std::vector<double>& patches) const = 0;

protected:
    void describe_common(sppm_object& o, int type) const {
        o = sppm_object{};
        o.type = type;
        o.texture = -1;
        for (int k = 0; k < 3; k++) {
            o.color[k] = color.val[k];
            o.material.diff[k] = material.diff.val[k];
            o.material.refl[k] = material.refl.val[k];
            o.material.refr[k] = material.refr.val[k];
        }
        o.material.refrn = material.refrn;
        o.material.refln = material.refln;
    }
};
inline Obj::~Obj() {}

class PlaneObj : public Obj {
public:
    Vec3 P0;
    Vec3 norm;
    PlaneObj(const Vec3& pp, const Vec3& nn, const Color& cc, const Material& mm) : Obj(mm, cc), P0(pp), norm(nn) { norm.Normalize(); }
    Vec3 texU = Vec3(400, 0, 0);
    Vec3 texV = Vec3(0, 0, 300);
    void Describe(sppm_object& o, std::vector<double>&) const override {
        describe_common(o, SPPM_OBJ_PLANE);
        for (int k = 0; k < 3; k++) { o.p0[k] = P0.val[k]; o.n[k] = norm.val[k]; }
    }
};

class SphereObj : public Obj {
public:
    Vec3 org;
    double radius;
    Vec3 texU = Vec3(0, 3, -3).GetUnit(), texV = Vec3(1, 0, 0).GetUnit();
    SphereObj(double rradius, const Vec3& orgg, const Color& cc, const Material& mm) : Obj(mm, cc), org(orgg), radius(rradius) {}
    void Describe(sppm_object& o, std::vector<double>&) const override {
        describe_common(o, SPPM_OBJ_SPHERE);
        for (int k = 0; k < 3; k++) o.p0[k] = org.val[k];
        o.radius = radius;
    }
};
