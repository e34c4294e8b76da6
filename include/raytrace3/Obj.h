// Obj.h — drop-in for the reference's raytracer/Obj.h: Obj / PlaneObj / SphereObj with the
// reference's constructors and public members (raytracer/Obj.h:5-154).  The geometry lives
// here only as data: every intersection of the render loop runs on the GPU, where
// Scene::NearCollide is evaluated from the flattened objects (Describe()).  GetIntersect on
// a single object is kept for API compatibility and is also answered by the GPU.
#pragma once
#include <vector>

#include "../sppm_abi.h"
#include "Bridge.h"
#include "Element.h"
#include "Light.h"
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

// ------------------------------------------------------------------ the host bridge: objects and lights -> C-ABI scene
namespace sppm_host {

struct FlatScene {
    std::vector<sppm_object> objs;
    std::vector<double> patches;
    std::vector<sppm_texture> tex;
    std::vector<sppm_light> lights;
    sppm_scene c{};
    uint64_t tag = 0;
};

// The tag decides whether a context's copy of the scene is still current (Device::ensure_scene).  The reference reads the
// live objects on every ray, so a host may change a material, a colour, a position or a texture between two passes: the
// tag therefore hashes the flattened CONTENT (every field of every object, the control points, the lights, each
// texture's size and texels), not the addresses of the objects.  A texture's share of it is cached in the Texture
// (Texture::content_hash: recomputed when the texel array moves or changes size, or after Texture::touch()).
inline void flatten(const std::vector<Obj*>& objvec, const std::vector<Light*>& plvec, FlatScene& f) {
    f = FlatScene();
    std::vector<const Texture*> seen;
    uint64_t tag = 1469598103934665603ull;
    auto mix = [&](uint64_t v) { tag = (tag ^ v) * 1099511628211ull; };
    auto mix_bytes = [&](const void* p, size_t n) {
        const unsigned char* b = (const unsigned char*)p;
        size_t i = 0;
        for (; i + 8 <= n; i += 8) { uint64_t w; std::memcpy(&w, b + i, 8); mix(w); }
        for (; i < n; i++) mix(b[i]);
    };
    for (Obj* o : objvec) {
        sppm_object so;
        std::memset(&so, 0, sizeof(so));               // padding bytes, if a compiler adds any, hash alike
        o->Describe(so, f.patches);
        if (const Texture* t = o->getTexture()) {
            size_t k = 0;
            while (k < seen.size() && seen[k] != t) k++;
            if (k == seen.size()) {
                seen.push_back(t);
                sppm_texture st;
                st.rows = t->rows; st.cols = t->cols; st.texels = t->texels.data();
                f.tex.push_back(st);
                mix(t->content_hash());
            }
            so.texture = (int)k;
        }
        f.objs.push_back(so);
        mix_bytes(&so, sizeof(so));
    }
    mix_bytes(f.patches.data(), f.patches.size() * sizeof(double));
    for (Light* l : plvec) {
        sppm_light sl;
        for (int k = 0; k < 3; k++) { sl.pos[k] = l->position().val[k]; sl.color[k] = l->power().val[k]; }
        f.lights.push_back(sl);
        mix_bytes(&sl, sizeof(sl));
    }
    mix(objvec.size()); mix(plvec.size());
    f.tag = tag;
    f.c.n_objects = (int)f.objs.size(); f.c.objects = f.objs.data();
    f.c.n_patches = (int)(f.patches.size() / 48); f.c.patches = f.patches.data();
    f.c.n_textures = (int)f.tex.size(); f.c.textures = f.tex.data();
    f.c.n_lights = (int)f.lights.size(); f.c.lights = f.lights.data();
}

// Device::ensure_scene (declared in Bridge.h): upload the scene if the context's copy is not this one.
inline void Device::ensure_scene(const std::vector<Obj*>& objvec, const std::vector<Light*>& plvec) {
    FlatScene f;
    flatten(objvec, plvec, f);
    use_scene(f.c, f.tag);
}

inline Collision intersect_objects(const std::vector<Obj*>& objs, const Ray& ray) {
    static SpotLight dummy(Vec3(), Vec3(1, 1, 1));
    std::vector<Light*> lights{&dummy};
    Device& dev = Device::get(0);
    std::lock_guard<std::mutex> g(dev.mu);
    dev.ensure_scene(objs, lights);
    sppm_collision c;
    dev.check(sppm_debug_near_collide(dev.ctx, 1, ray.org.val, ray.dir.val, 0, 0, &c), "sppm_debug_near_collide");
    if (c.obj < 0) return NoneColl;
    Collision out(Vec3(c.pos[0], c.pos[1], c.pos[2]), ray.dir, c.dist, objs[c.obj], Vec3(c.n[0], c.n[1], c.n[2]),
                  Vec3(c.color[0], c.color[1], c.color[2]));
    out.inside = c.inside != 0;
    return out;
}

}  // namespace sppm_host
