// trace_members.cpp — RayTracer::ViewTrace and RayTracer::PhotonTrace used the way the reference's own GetHitPoint() and
// PhotonMap() use them (raytracer/Raytracer.h:102-116, :220-233): one path per call, through the drop-in classes.
//
//   trace_members <scene file> [w h photons]
// Traces every eye path of a w x h image by ViewTrace and checks the HitPoints against those of GetHitPoint(); then sends
// `photons` photons of Light::emit through PhotonTrace and checks that HitPoints collected them.  Prints "OK ..." on success.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <string>

#include "Camera.h"
#include "Light.h"
#include "Raytracer.h"
#include "Scene.h"

int main(int argc, char** argv) {
    if (argc < 2) { std::fprintf(stderr, "usage: %s file.scene [w h photons]\n", argv[0]); return 2; }
    const int w = argc > 2 ? std::atoi(argv[2]) : 24, h = argc > 3 ? std::atoi(argv[3]) : 18, photons = argc > 4 ? std::atoi(argv[4]) : 64;
    Camera camera;
    const Vec3 org(50, 35, 230);
    camera.init(w, h);
    camera.MoveTo(org);
    camera.lookAt(org + Vec3(0, 0.042612, -1), 1.);
    Scene scene;
    try {
        scene.load(argv[1]);
    } catch (const std::exception& e) {
        std::fprintf(stderr, "%s\n", e.what());
        return 1;
    }
    RayTracer whole, single;
    whole.camera = single.camera = &camera;
    whole.scene = single.scene = &scene;
    whole.GetHitPoint();
    for (int i = 0; i < camera.h; i++)
        for (int j = 0; j < camera.w; j++) single.ViewTrace(camera.emit(i, j), 0, Color(1, 1, 1), i, j);
    if (single.Hvec.size() != whole.Hvec.size() || whole.Hvec.empty()) {
        std::fprintf(stderr, "ViewTrace left %zu HitPoints, GetHitPoint %zu\n", single.Hvec.size(), whole.Hvec.size());
        return 1;
    }
    // the same HitPoints, whatever the order: matched by pixel and position
    std::multimap<long long, const HitPoint*> by_pixel;
    for (const HitPoint* hp : whole.Hvec) by_pixel.insert({(long long)hp->x * camera.w + hp->y, hp});
    double worst = 0;
    for (const HitPoint* hp : single.Hvec) {
        auto range = by_pixel.equal_range((long long)hp->x * camera.w + hp->y);
        double best = 1e30;
        for (auto it = range.first; it != range.second; ++it) {
            const Vec3 d = it->second->pos - hp->pos;
            best = std::min(best, std::sqrt(d.dot(d)));
        }
        worst = std::max(worst, best);
    }
    if (!(worst < 1e-6)) { std::fprintf(stderr, "ViewTrace HitPoints differ from GetHitPoint's by %g\n", worst); return 1; }
    long long collected = 0;
    for (int k = 0; k < photons; k++) {
        Ray ray;
        Vec3 flux;
        scene.plvec[0]->emit(ray, flux);
        whole.PhotonTrace(ray, 0, flux);
    }
    double tao = 0;
    for (const HitPoint* hp : whole.Hvec) { collected += hp->newN; tao += hp->tao.val[0] + hp->tao.val[1] + hp->tao.val[2]; }
    if (collected <= 0 || !(tao > 0)) { std::fprintf(stderr, "PhotonTrace left no photon in any HitPoint\n"); return 1; }
    std::printf("OK %zu hitpoints, worst distance %.3g, %lld photon-hitpoint pairs from %d photons\n", whole.Hvec.size(), worst, collected, photons);
    return 0;
}
