// Scene.h — drop-in for the reference's raytracer/Scene.h: Scene{objvec, plvec, init(),
// NearCollide} (raytracer/Scene.h:93-183), plus the host bridge that flattens a scene into
// the C-ABI structs of sppm_abi.h and keeps one GPU context per device.
#pragma once
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
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

// The tag decides whether a context's copy of the scene is still current (Device::ensure_scene).  The reference reads the
// live objects on every ray, so a host may change a material, a colour, a position or a texture between two passes: the
// tag therefore hashes the flattened CONTENT (every field of every object, the control points, the lights, each
// texture's size and texels), not the addresses of the objects.
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
                mix((uint64_t)t->rows); mix((uint64_t)t->cols);
                mix_bytes(t->texels.data(), t->texels.size());
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

// A hardware queue per context: HIP's default of 4 queues per process folds the streams of four contexts and a film
// onto shared queues, whose kernels then run one after the other (measured: 789 instead of 1015 Mphotons/s).  The HIP
// runtime reads GPU_MAX_HW_QUEUES when it starts, so the value is set when the program is loaded, before any GPU call
// (SPPM_HW_QUEUES picks another value, 0 leaves the environment alone).
inline int set_hw_queues() {
    const char* e = std::getenv("SPPM_HW_QUEUES");
    const std::string v = (e && *e) ? e : "8";
    return v == "0" ? 0 : setenv("GPU_MAX_HW_QUEUES", v.c_str(), 1);
}
inline const int hw_queues_set = set_hw_queues();

// SPPM_DEVICES: comma-separated GPU ids, one context slot per entry (an id may repeat).
inline std::vector<int> env_devices() {
    std::vector<int> d;
    const char* e = std::getenv("SPPM_DEVICES");
    std::stringstream ss(e && *e ? e : "0,0,0,0");
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

    // Reads a scene description file instead of the hard-coded init(): lines of `keyword key=value ...`, vectors as
    // comma lists, '#' starts a comment (the format is documented in raytrace3_amd/scenes.py, which reads the same files):
    //   material NAME [diff=] [refl=] [refr=] [refrn=] [refln=]   |  texture NAME file=
    //   plane p0= n= color= material= [texture=]                  |  sphere radius= center= color= material= [texture=]
    //   bezier file= center= scale= color= material= [texture=] [trans=m00,...,m22]   |  light pos= color=
    // Files are looked up next to the scene file, then as Texture / Bezier3Obj do (working directory, SPPM_DATA_DIR).
    void load(const std::string& path) {
        std::ifstream in(path);
        if (!in) throw std::runtime_error("Scene::load: cannot open " + path);
        const size_t slash = path.find_last_of('/');
        const std::string here = slash == std::string::npos ? std::string(".") : path.substr(0, slash);
        auto locate = [&](const std::string& name) {
            std::ifstream probe(here + "/" + name);
            return probe ? here + "/" + name : name;
        };
        std::map<std::string, Material> mats;
        std::map<std::string, Texture*> texs;
        std::string line;
        for (int ln = 1; std::getline(in, line); ln++) {
            line = line.substr(0, line.find('#'));
            std::istringstream ls(line);
            std::string kw, tok;
            if (!(ls >> kw)) continue;
            std::vector<std::string> names;
            std::map<std::string, std::string> a;
            while (ls >> tok) {
                const size_t eq = tok.find('=');
                if (eq == std::string::npos) names.push_back(tok); else a[tok.substr(0, eq)] = tok.substr(eq + 1);
            }
            auto fail = [&](const std::string& what) -> std::runtime_error {
                return std::runtime_error(path + ":" + std::to_string(ln) + ": " + what);
            };
            auto need = [&](const char* key) -> const std::string& {
                auto it = a.find(key);
                if (it == a.end()) throw fail(kw + " needs " + key + "=");
                return it->second;
            };
            auto nums = [&](const std::string& text, size_t n, const char* what) {
                std::vector<double> v;
                std::stringstream ss(text);
                std::string item;
                while (std::getline(ss, item, ',')) v.push_back(std::atof(item.c_str()));
                if (v.size() != n) throw fail(std::string(what) + " needs " + std::to_string(n) + " numbers");
                return v;
            };
            auto vec3 = [&](const std::string& text, const char* what) { auto v = nums(text, 3, what); return Vec3(v[0], v[1], v[2]); };
            auto opt3 = [&](const char* key) { return a.count(key) ? vec3(a[key], key) : Vec3(); };
            auto mat_of = [&]() -> const Material& {
                auto it = mats.find(need("material"));
                if (it == mats.end()) throw fail("unknown material " + a["material"]);
                return it->second;
            };
            auto dress = [&](Obj* o) {
                if (a.count("texture")) {
                    auto it = texs.find(a["texture"]);
                    if (it == texs.end()) { delete o; throw fail("unknown texture " + a["texture"]); }
                    o->setTexture(it->second);
                }
                objvec.push_back(o);
            };
            if (kw == "material") {
                if (names.size() != 1) throw fail("material needs a name");
                mats.erase(names[0]);
                mats.emplace(names[0], Material(opt3("diff"), opt3("refl"), opt3("refr"), a.count("refrn") ? std::atof(a["refrn"].c_str()) : 1.5,
                                                a.count("refln") ? std::atof(a["refln"].c_str()) : 1.0));
            } else if (kw == "texture") {
                if (names.size() != 1) throw fail("texture needs a name");
                texs[names[0]] = new Texture(locate(need("file")));
            } else if (kw == "plane") {
                dress(new PlaneObj(vec3(need("p0"), "p0"), vec3(need("n"), "n"), vec3(need("color"), "color"), mat_of()));
            } else if (kw == "sphere") {
                dress(new SphereObj(std::atof(need("radius").c_str()), vec3(need("center"), "center"), vec3(need("color"), "color"), mat_of()));
            } else if (kw == "bezier") {
                Mat33 T = Mat33::eye();
                if (a.count("trans")) {
                    auto v = nums(a["trans"], 9, "trans");
                    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) T(i, j) = v[(size_t)i * 3 + j];
                }
                Bezier3Obj* b = new Bezier3Obj(vec3(need("center"), "center"), std::atof(need("scale").c_str()), locate(need("file")),
                                               vec3(need("color"), "color"), mat_of(), T);
                if (b->patch_count() == 0) { delete b; throw fail("cannot find " + a["file"] + " (or it holds no bicubic patch)"); }
                dress(b);
            } else if (kw == "light") {
                plvec.push_back(new SpotLight(vec3(need("pos"), "pos"), vec3(need("color"), "color")));
            } else {
                throw fail("unknown keyword " + kw);
            }
        }
        std::cout << "scene::load(" << path << "): " << objvec.size() << " objects, " << plvec.size() << " lights" << std::endl;
    }

    // Nearest hit over all objects (raytracer/Scene.h:165-182), answered by the GPU.
    Collision NearCollide(const Ray& ray) {
        return sppm_host::intersect_objects(objvec, ray);
    }
};
