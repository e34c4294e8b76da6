// render_main.cpp — a C++ host using the drop-in classes the way the reference's main.cpp
// does (camera set-up, scene of Obj subclasses, RayTracer::render), on a texture-free copy
// of the reference scene.  Writes the resolved flux image as raw doubles ([h][w][3]) so the
// tests can compare it with the Python-driven path and the oracle.
//
//   render_main <out.f64> [w h rounds replica [stock | file.scene]]
// With "stock" as sixth argument the scene is the reference's own Scene::init(), with a file name the scene file
// is read by Scene::load() (textures: <name>.ppm next to the working
// directory or in SPPM_DATA_DIR, procedural stand-ins otherwise).
#include <cstdio>
#include <cstdlib>
#include <string>

#include "Camera.h"
#include "Element.h"
#include "Light.h"
#include "Obj.h"
#include "Raytracer.h"
#include "Scene.h"
#include "Vec3.h"

int main(int argc, char** argv) {
    if (argc < 2) { std::fprintf(stderr, "usage: %s out.f64 [w h rounds replica [stock]]\n", argv[0]); return 2; }
    const int w = argc > 2 ? std::atoi(argv[2]) : 64, h = argc > 3 ? std::atoi(argv[3]) : 64;
    const int rounds = argc > 4 ? std::atoi(argv[4]) : 100;
    const unsigned replica = argc > 5 ? (unsigned)std::atoi(argv[5]) : 0;

    Camera camera;
    Vec3 org = Vec3(50, 35, 230);
    camera.init(w, h);
    camera.MoveTo(org);
    camera.lookAt(org + Vec3(0, 0.042612, -1), 1 / 1.);

    Scene scene;
    if (argc > 6 && std::string(argv[6]) == "stock") {
        scene.init();
    } else if (argc > 6) {
        try {
            scene.load(argv[6]);                   // a scene description file
        } catch (const std::exception& e) {
            std::fprintf(stderr, "%s\n", e.what());
            return 1;
        }
    } else {
    const Material WHITE_DIFF(Vec3(.75, .75, .75), Vec3(), Vec3(), 0);
    const Material MIRROR(Vec3(), Vec3(1, 1, 1) * .999, Vec3(), 0);
    const Material REFR0(Vec3(), Vec3(), Vec3(.999, .999, .999), 1.5);
    scene.objvec.push_back(new PlaneObj(Vec3(1, 40.8, 81.6), Vec3(1, 0, 0), Vec3(.75, .25, .25), WHITE_DIFF));
    scene.objvec.push_back(new PlaneObj(Vec3(99, 40.8, 81.6), Vec3(1, 0, 0), Vec3(.25, .25, .75), WHITE_DIFF));
    scene.objvec.push_back(new PlaneObj(Vec3(50, 40.8, 0), Vec3(0, 0, 1), Vec3(.75, .75, .75), MIRROR));
    scene.objvec.push_back(new PlaneObj(Vec3(50, 0, 81.6), Vec3(0, -1, 0), Vec3(.75, .75, .75), WHITE_DIFF));
    scene.objvec.push_back(new PlaneObj(Vec3(50, 81.6, 81.6), Vec3(0, 1, 0), Vec3(.75, .75, .75), WHITE_DIFF));
    scene.objvec.push_back(new SphereObj(16.5, Vec3(27, 16.5, 47), Vec3(.999, .999, .999), MIRROR));
    scene.objvec.push_back(new SphereObj(16.5, Vec3(73, 16.5, 88), Vec3(.999, .999, .999), REFR0));
    scene.objvec.push_back(new SphereObj(8.5, Vec3(50, 8.5, 60), Vec3(.999, .999, .999), WHITE_DIFF));
    Mat33 Trans, Trans2;
    Trans(0, 0) = 1; Trans(1, 2) = 1; Trans(2, 1) = 1;
    const double theta = (90. / 180.) * PI;
    Trans2(0, 0) = std::cos(theta); Trans2(0, 2) = std::sin(theta); Trans2(1, 1) = 1;
    Trans2(2, 0) = -std::sin(theta); Trans2(2, 2) = std::cos(theta);
    scene.objvec.push_back(new Bezier3Obj(Vec3(20, 0, 120), 4, "teapot.bpt", Vec3(0, .999, .999), WHITE_DIFF, Trans2 * Trans));
    scene.plvec.push_back(new SpotLight(Vec3(50, 60, 85), 5000 * Color(1, 1, 1)));
    }

    RayTracer r1;
    r1.camera = &camera;
    r1.scene = &scene;
    r1.replica = replica;
    r1.GetHitPoint();
    r1.PhotonMap(rounds);

    FILE* f = std::fopen(argv[1], "wb");
    if (!f) return 1;
    for (int i = 0; i < camera.h; i++)
        for (int j = 0; j < camera.w; j++) std::fwrite(camera.pic[i][j].val, sizeof(double), 3, f);
    std::fclose(f);
    std::printf("hitpoints %zu\n", r1.Hvec.size());
    return 0;
}
