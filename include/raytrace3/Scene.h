// Scene.h — drop-in for the reference's raytracer/Scene.h: Scene{objvec, plvec, init(),
// NearCollide} (raytracer/Scene.h:93-183), plus the host bridge that flattens a scene into
// the C-ABI structs of sppm_abi.h and keeps one GPU context per device.
#pragma once
#include <map>
#include <sstream>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <vector>

#include "../sppm_abi.h"
#include "Bezier.h"
#include "Element.h"
#include "Light.h"
#include "Obj.h"
#include "Vec3.h"

namespace sppm_host {

struct FlatScene {
    std::vector<sppm_object> objs;
    std::vector<double> patches;
    std::vector<sppm_texture> tex;
    std::vector<sppm_light> lights;
    sppm_scene c{};
    uint64_t tag = 0;
};

inline void flatten(const std::vector<Obj*>& objvec, const std::vector<Light*>& plvec, FlatScene& f) {
    f = FlatScene();
    std::vector<const Texture*> seen;
    uint64_t tag = 1469598103934665603ull;
    auto mix = [&](uint64_t v) { tag = (tag ^ v) * 1099511628211ull; };
    for (Obj* o : objvec) {
        sppm_object so;
        o->Describe(so, f.patches);
        if (const Texture* t = o->getTexture()) {
            size_t k = 0;
            while (k < seen.size() && seen[k] != t) k++;
            if (k == seen.size()) {
                seen.push_back(t);
                sppm_texture st;
                st.rows = t->rows; st.cols = t->cols; st.texels = t->texels.data();
                f.tex.push_back(st);
            }
            so.texture = (int)k;
        }
        f.objs.push_back(so);
        mix((uint64_t)(uintptr_t)o);
        mix((uint64_t)(uintptr_t)o->getTexture());
    }
    for (Light* l : plvec) {
        sppm_light sl;
        for (int k = 0; k < 3; k++) { sl.pos[k] = l->position().val[k]; sl.color[k] = l->power().val[k]; }
        f.lights.push_back(sl);
        mix((uint64_t)(uintptr_t)l);
    }
    mix(objvec.size()); mix(plvec.size());
    f.tag = tag;
    f.c.n_objects = (int)f.objs.size(); f.c.objects = f.objs.data();
    f.c.n_patches = (int)(f.patches.size() / 48); f.c.patches = f.patches.data();
    f.c.n_textures = (int)f.tex.size(); f.c.textures = f.tex.data();
    f.c.n_lights = (int)f.lights.size(); f.c.lights = f.lights.data();
}

// SPPM_DEVICES: comma-separated GPU ids, one context slot per entry (an id may repeat).
inline std::vector<int> env_devices() {
    std::vector<int> d;
    const char* e = std::getenv("SPPM_DEVICES");
    std::stringstream ss(e && *e ? e : "0,0");
    std::string tok;
    while (std::getline(ss, tok, ',')) if (!tok.empty()) d.push_back(std::atoi(tok.c_str()));
    if (d.empty()) d.push_back(0);
    return d;
}

// One library context per slot of SPPM_DEVICES, created on first use on that slot's GPU; `mu`
// serialises callers because a context serves one host thread at a time.
struct Device {
    sppm_ctx* ctx = nullptr;
    std::mutex mu;
    uint64_t scene_tag = 0;
    ~Device() { if (ctx) sppm_destroy(ctx); }
    void check(int rc, const char* what) const {
        if (rc != SPPM_OK) throw std::runtime_error(std::string(what) + " failed: " + (ctx ? sppm_last_error(ctx) : "no context"));
    }
    void ensure_scene(const std::vector<Obj*>& objvec, const std::vector<Light*>& plvec) {
        FlatScene f;
        flatten(objvec, plvec, f);
        if (f.tag == scene_tag) return;
        check(sppm_set_scene(ctx, &f.c), "sppm_set_scene");
        scene_tag = f.tag;
    }
    static Device& get(int slot_index) {
        static std::mutex reg_mu;
        static std::map<int, std::unique_ptr<Device>> reg;
        static const std::vector<int> devices = env_devices();
        std::lock_guard<std::mutex> g(reg_mu);
        const int key = (int)((size_t)(slot_index < 0 ? 0 : slot_index) % devices.size());
        const int id = devices[(size_t)key];
        auto& slot = reg[key];
        if (!slot) {
            slot.reset(new Device());
            int rc = sppm_create(&slot->ctx, id);
            if (rc != SPPM_OK) {
                reg.erase(key);
                throw std::runtime_error("sppm_create failed: no usable GPU " + std::to_string(id) + " (there is no CPU fallback)");
            }
        }
        return *slot;
    }
};

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

class Scene {
public:
    std::vector<Obj*> objvec;
    std::vector<Light*> plvec;

    // The reference's hard-coded scene (raytracer/Scene.h:98-164): five planes (the back one
    // a mirror), a mirror, a glass and a textured diffuse sphere, the teapot of teapot.bpt
    // scaled by 4 at (20,0,120), four textures and one light of power 5000.
    void init() {
        const Material WHITE_DIFF(Vec3(.75, .75, .75), Vec3(), Vec3(), 0);
        const Material MIRROR(Vec3(), Vec3(1, 1, 1) * .999, Vec3(), 0);
        const Material REFR0(Vec3(), Vec3(), Vec3(.999, .999, .999), 1.5);
        objvec.push_back(new PlaneObj(Vec3(1, 40.8, 81.6), Vec3(1, 0, 0), Vec3(.75, .25, .25), WHITE_DIFF));   // left
        objvec.push_back(new PlaneObj(Vec3(99, 40.8, 81.6), Vec3(1, 0, 0), Vec3(.25, .25, .75), WHITE_DIFF));  // right
        objvec.push_back(new PlaneObj(Vec3(50, 40.8, 0), Vec3(0, 0, 1), Vec3(.75, .75, .75), MIRROR));          // back
        objvec.push_back(new PlaneObj(Vec3(50, 0, 81.6), Vec3(0, -1, 0), Vec3(.75, .75, .75), WHITE_DIFF));     // bottom
        objvec.push_back(new PlaneObj(Vec3(50, 81.6, 81.6), Vec3(0, 1, 0), Vec3(.75, .75, .75), WHITE_DIFF));   // top
        objvec.push_back(new SphereObj(16.5, Vec3(27, 16.5, 47), Vec3(1, 1, 1) * .999, MIRROR));
        objvec.push_back(new SphereObj(16.5, Vec3(73, 16.5, 88), Vec3(1, 1, 1) * .999, REFR0));
        objvec.push_back(new SphereObj(8.5, Vec3(50, 8.5, 60), Vec3(1, 1, 1) * .999, WHITE_DIFF));
        Texture* timg = new Texture("timg.jpg");
        Texture* stone = new Texture("wall.jpg");
        objvec[0]->setTexture(stone);
        objvec[1]->setTexture(stone);
        objvec[4]->setTexture(stone);
        objvec[7]->setTexture(new Texture("planet.jpg"));
        objvec[3]->setTexture(timg);
        Mat33 Trans;
        Trans(0, 0) = 1; Trans(1, 2) = 1; Trans(2, 1) = 1;
        Mat33 Trans2;
        const double theta = (90. / 180.) * PI;
        Trans2(0, 0) = std::cos(theta); Trans2(0, 2) = std::sin(theta);
        Trans2(1, 1) = 1;
        Trans2(2, 0) = -std::sin(theta); Trans2(2, 2) = std::cos(theta);
        objvec.push_back(new Bezier3Obj(Vec3(20, 0, 120), 4, "teapot.bpt", Vec3(0, 1, 1) * .999, WHITE_DIFF, Trans2 * Trans));
        objvec[8]->setTexture(new Texture("blue.jpg"));
        plvec.push_back(new SpotLight(Vec3(50, 60, 85), 5000 * Color(1, 1, 1)));
        std::cout << "scene::init()!" << std::endl;
    }

    // Nearest hit over all objects (raytracer/Scene.h:165-182), answered by the GPU.
    Collision NearCollide(const Ray& ray) {
        return sppm_host::intersect_objects(objvec, ray);
    }
};
