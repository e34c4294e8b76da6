// sppm_main.cpp — the outer SPPM loop through the drop-in classes: the stock scene of Scene::init,
// four jittered replicas per iteration (SPPMRayTracer::render), running mean written as
// aaanew<i/10>.ppm into the working directory.  Resolution, iterations and photon rounds come from
// the environment (SPPM_W, SPPM_H, SPPM_ITERS, SPPM_ROUNDS); SPPM_DEVICES picks the contexts.
#include <cstdlib>

#include "Camera.h"
#include "Raytracer.h"
#include "Scene.h"
#include "Vec3.h"

int main() {
    Camera camera;
    const Vec3 eye(50, 35, 230);
    camera.init((int)sppm_host::env_ll("SPPM_W", 1024), (int)sppm_host::env_ll("SPPM_H", 1024));
    camera.MoveTo(eye);
    camera.lookAt(eye + Vec3(0, 0.042612, -1), 1.);
    Scene scene;
    scene.init();
    SPPMRayTracer tracer;
    tracer.render(scene, camera);
    return 0;
}
