// Raytracer.h — drop-in for the reference's raytracer/Raytracer.h: HitPoint, RayTracer
// (one fixed-radius photon-mapping pass) and SPPMRayTracer (the outer loop over jittered
// replicas), with the reference's member names (raytracer/Raytracer.h:47-477).  render(),
// GetHitPoint() and PhotonMap() call the GPU library through sppm_abi.h; nothing is traced
// on the CPU.
//
// Environment knobs (all optional, defaults = the reference's literals):
//   SPPM_ROUNDS   rounds per PhotonMap in render()      (10000, Raytracer.h:384)
//   SPPM_ONETIME  photons per round and light           (100,   Raytracer.h:218)
//   SPPM_ITERS    outer SPPM iterations                 (100000, Raytracer.h:425)
//   SPPM_SEED     seed of the counter-based streams     (20260101)
//   SPPM_DEVICES  comma-separated GPU ids, one library context (= HIP stream) per entry; the four
//                 replicas of an iteration are dealt over them.  Default "0,0": two contexts on GPU 0,
//                 so the eye pass of one replica overlaps the photon pass of another.
#pragma once
#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <sstream>
#include <string>
#include <thread>
#include <vector>

#include "../sppm_abi.h"
#include "Camera.h"
#include "Scene.h"
using namespace std;

#define DEBUG
#define MIN_WGT 1e-3
#define MAX_DEP 13
#define INIT_R2 2.0

inline int toInt(double x) { return int(pow(1 - exp(-x), 1 / 2.2) * 255 + .5); }
const double alpha = 0.7;

namespace sppm_host {
inline long long env_ll(const char* name, long long dflt) {
    const char* e = std::getenv(name);
    return (e && *e) ? std::atoll(e) : dflt;
}
// Tone-mapped running mean as a binary PPM: rows flipped, and val[0] written as blue, which
// is what the reference's BGR cv::Mat shows (Raytracer.h:460-470).
inline bool write_ppm(const std::string& path, const Camera& cam, double n_images) {
    FILE* f = std::fopen(path.c_str(), "wb");
    if (!f) return false;
    std::fprintf(f, "P6\n%d %d\n255\n", cam.w, cam.h);
    std::vector<unsigned char> row((size_t)cam.w * 3);
    for (int i1 = cam.h - 1; i1 >= 0; i1--) {
        for (int j = 0; j < cam.w; j++)
            for (int k = 0; k < 3; k++) {
                int v = toInt(cam.pic[i1][j].val[k] / n_images);
                row[(size_t)j * 3 + (2 - k)] = (unsigned char)std::min(255, std::max(0, v));
            }
        std::fwrite(row.data(), 1, row.size(), f);
    }
    std::fclose(f);
    return true;
}
}  // namespace sppm_host

struct HitPoint {
    Vec3 pos;
    Vec3 n;
    Vec3 v;
    Obj* obj;
    int x, y;  // x = image row, y = image column
    Color wgt;
    double R2;
    int N;
    Color tao;
    int newN = 0;
    friend ostream& operator<<(ostream& out, const HitPoint& hp) {
        out << "pos:" << hp.pos << "x,y:" << hp.x << " " << hp.y << endl << "R2:" << hp.R2 << " tao" << hp.tao << endl;
        return out;
    }
    explicit HitPoint(const Vec3& pos_ = Vec3(), const Vec3& n_ = Vec3(), const Vec3& v_ = Vec3(), Obj* o = nullptr, int xx = -1,
                      int yy = -1, const Color& wgt_ = Color())
        : pos(pos_), n(n_), v(v_), obj(o), x(xx), y(yy), wgt(wgt_), R2(INIT_R2), N(0), tao(Color()) {}
    // The reference's radius shrink; with N starting at 0 it never gets past the guard.
    void update() {
        if (N <= 0 || newN <= 0) return;
        double k = (N + alpha * newN) / (N + newN);
        R2 *= k;
        tao *= k;
        N += alpha * newN;
        newN = 0;
    }
};

class RayTracer {
public:
    Camera* camera = nullptr;
    Scene* scene = nullptr;
    vector<HitPoint*> Hvec;
    int device = 0;                       // context slot this tracer renders on (entry of SPPM_DEVICES)
    uint64_t seed = (uint64_t)sppm_host::env_ll("SPPM_SEED", 20260101);
    uint32_t replica = 0;                 // selects the random streams of this pass

    RayTracer() {}

    // Pass 1 on the GPU; Hvec receives a copy of the HitPoints.
    void GetHitPoint() {
        sppm_host::Device& dev = sppm_host::Device::get(device);
        std::lock_guard<std::mutex> g(dev.mu);
        eye_locked(dev);
        fetch_hitpoints(dev);
    }
    // Pass 2 on the GPU: cnt rounds x lights x onetime photons, then the resolve into camera->pic.
    void PhotonMap(int cnt) {
        sppm_host::Device& dev = sppm_host::Device::get(device);
        std::lock_guard<std::mutex> g(dev.mu);
        photon_locked(dev, cnt);
        fetch_hitpoints(dev);
    }
    void render() {
        sppm_host::Device& dev = sppm_host::Device::get(device);
        std::lock_guard<std::mutex> g(dev.mu);
        eye_locked(dev);
        photon_locked(dev, (int)sppm_host::env_ll("SPPM_ROUNDS", 10000));
    }
    void debugHitPoint() {
        cout << "HitPoint:size:" << Hvec.size() << endl;
        for (auto e : Hvec) cout << *e;
    }
    ~RayTracer() {
        clear_hvec();
        camera = nullptr;
        scene = nullptr;
    }

private:
    void clear_hvec() {
        for (auto hp : Hvec) delete hp;
        vector<HitPoint*>().swap(Hvec);
    }
    void eye_locked(sppm_host::Device& dev) {
        if (!camera || !scene) throw std::runtime_error("RayTracer: camera and scene must be set");
        dev.ensure_scene(scene->objvec, scene->plvec);
        sppm_camera c;
        for (int k = 0; k < 3; k++) { c.pos[k] = camera->pos.val[k]; c.dir[k] = camera->dir.val[k]; c.du[k] = camera->du.val[k]; c.dv[k] = camera->dv.val[k]; }
        c.w = camera->w; c.h = camera->h;
        cout << "GetHitPoint():" << camera->pos << endl;
        dev.check(sppm_eye_pass(dev.ctx, &c, seed, replica), "sppm_eye_pass");
    }
    void photon_locked(sppm_host::Device& dev, int cnt) {
        const int onetime = (int)sppm_host::env_ll("SPPM_ONETIME", 100);
        cout << "PhotonMap:" << cnt << endl << "onetime:" << onetime << endl;
        dev.check(sppm_photon_pass(dev.ctx, (uint64_t)cnt, (uint32_t)onetime, 0, 0, seed, replica), "sppm_photon_pass");
        std::vector<double> pic((size_t)camera->w * camera->h * 3);
        dev.check(sppm_resolve(dev.ctx, pic.data(), 0), "sppm_resolve");
        for (int i = 0; i < camera->h; i++)
            for (int j = 0; j < camera->w; j++) {
                const double* p = &pic[((size_t)i * camera->w + j) * 3];
                camera->pic[i][j] = Vec3(p[0], p[1], p[2]);
            }
        cout << (long long)cnt * onetime << endl;
    }
    void fetch_hitpoints(sppm_host::Device& dev) {
        clear_hvec();
        uint64_t n = 0;
        dev.check(sppm_get_hitpoints(dev.ctx, nullptr, 0, &n), "sppm_get_hitpoints");
        std::vector<sppm_hitpoint> hp(n);
        if (n) dev.check(sppm_get_hitpoints(dev.ctx, hp.data(), n, &n), "sppm_get_hitpoints");
        Hvec.reserve(n);
        for (const sppm_hitpoint& h : hp) {
            HitPoint* p = new HitPoint(Vec3(h.pos[0], h.pos[1], h.pos[2]), Vec3(h.n[0], h.n[1], h.n[2]), Vec3(h.v[0], h.v[1], h.v[2]),
                                       scene->objvec[h.obj], h.x, h.y, Vec3(h.wgt[0], h.wgt[1], h.wgt[2]));
            p->R2 = h.R2; p->N = h.N; p->newN = h.newN; p->tao = Vec3(h.tao[0], h.tao[1], h.tao[2]);
            Hvec.push_back(p);
        }
        cout << "hitpoint::" << Hvec.size() << endl;
    }
};

class SPPMRayTracer {
public:
    // The outer SPPM loop: per iteration four replicas with the camera moved by
    // 0.00015 * a random unit vector, their images summed, the running mean tone-mapped and
    // written out (aaanew<i/10>.ppm).  Replicas are spread over SPPM_DEVICES.
    void render(Scene scene, Camera camera) {
        const long long sppmtime = sppm_host::env_ll("SPPM_ITERS", 100000);
        const std::vector<int> devices = sppm_host::env_devices();
        Camera c[4];
        for (long long i = 0; i < sppmtime; i++) {
            RayTracer rt[4];
            for (int k = 0; k < 4; k++) {
                c[k] = camera;
                c[k].MoveTo(camera.pos + 0.00015 * Vec3::GetUnitRandVec());
                c[k].lookAt(camera.look, 1.);
                rt[k].camera = &c[k];
                rt[k].scene = &scene;
                rt[k].device = (int)((size_t)k % devices.size());   // context slot, see sppm_host::Device::get
                rt[k].replica = (uint32_t)(i * 4 + k);
            }
            std::vector<std::thread> th;
            std::vector<std::string> errors(4);
            for (int j = 0; j < 4; j++)
                th.emplace_back([&, j]() {
                    try { rt[j].render(); } catch (const std::exception& e) { errors[j] = e.what(); }
                    std::cerr << "Thread " << j << " done" << std::endl;
                });
            for (auto& t : th) t.join();
            for (const std::string& e : errors) if (!e.empty()) throw std::runtime_error(e);
            for (int k = 0; k < 4; k++)
                for (int i1 = 0; i1 < camera.h; i1++)
                    for (int j = 0; j < camera.w; j++) camera.pic[i1][j] += c[k].pic[i1][j];
            std::string pic = std::string("aaanew") + std::to_string(i / 10) + std::string(".ppm");
            std::cerr << "save:" << pic << std::endl;
            sppm_host::write_ppm(pic, camera, (i + 1) * 4.);
        }
    }
};
