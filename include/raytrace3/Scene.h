// Scene.h — drop-in for the reference's raytracer/Scene.h: Scene{objvec, plvec, init(),
// NearCollide} (raytracer/Scene.h:93-183).  The host bridge that flattens a scene into the C-ABI structs of
// sppm_abi.h is at the end of Obj.h, the GPU contexts it talks to in Bridge.h.
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
#include "Bridge.h"
#include "Element.h"
#include "Light.h"
#include "Obj.h"
#include "Vec3.h"

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
