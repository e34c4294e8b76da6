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
//   SPPM_DEVICES  comma-separated GPU ids, one library context (= HIP stream) per entry; the replicas go
//                 round them.  Default "0,0,0,0": four contexts on GPU 0, whose kernels share the CUs (the trace
//                 of one replica beside the gather of another).  Two GPUs: "0,1,0,1,0,1,0,1".
//   SPPM_INITIAL_N / SPPM_UPDATE_ROUNDS   HitPoints start with this N (0: the reference, whose update() then never runs) and,
//                 if positive, update() runs after every SPPM_UPDATE_ROUNDS rounds (1) — sppm_set_radius_update
//   SPPM_SAVE_EVERY        write the running mean every n-th outer iteration (1, as the reference) and after the last
//   SPPM_CHECKPOINT        file that receives the fp64 image sum, the iteration count and the jitter generator's state
//   SPPM_CHECKPOINT_EVERY  ... every n-th outer iteration (10) and after the last
//   SPPM_RESUME            checkpoint file to continue from
#pragma once
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <map>
#include <mutex>
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
// A byte image [h][w][3] in PPM channel order as a binary PPM.
inline bool write_ppm(const std::string& path, int w, int h, const std::vector<unsigned char>& rgb) {
    FILE* f = std::fopen(path.c_str(), "wb");
    if (!f) return false;
    std::fprintf(f, "P6\n%d %d\n255\n", w, h);
    const bool ok = std::fwrite(rgb.data(), 1, rgb.size(), f) == rgb.size();
    return std::fclose(f) == 0 && ok;
}

// The image sums of the outer SPPM loop, one film per GPU in use (camera.pic of Raytracer.h:449-458 lives on the GPUs).
struct Films {
    int w = 0, h = 0;
    std::map<int, sppm_film*> by_gpu;
    ~Films() { for (auto& kv : by_gpu) sppm_film_destroy(kv.second); }
    static void check(int rc, sppm_film* f, const char* what) {
        if (rc != SPPM_OK) throw std::runtime_error(std::string(what) + " failed: " + sppm_film_last_error(f));
    }
    sppm_film* on(int gpu) {
        auto it = by_gpu.find(gpu);
        if (it != by_gpu.end()) return it->second;
        sppm_film* f = nullptr;
        sppm_host::init();                 // before the first GPU call of the process, whichever comes first: a film or a context
        if (sppm_film_create(&f, gpu, w, h) != SPPM_OK) throw std::runtime_error("sppm_film_create failed on GPU " + std::to_string(gpu));
        by_gpu[gpu] = f;
        return f;
    }
    // The sum over all films.  Several GPUs: ONE all-reduce over xGMI (sppm_film_allreduce: RCCL, fp64) leaves it on every
    // GPU, instead of each film coming back over PCIe to be added on the host (SPPM_FILM_SUM=host keeps that older way,
    // for a machine without librccl).
    bool reduce_on_host() const { const char* e = std::getenv("SPPM_FILM_SUM"); return e && std::string(e) == "host"; }
    std::vector<sppm_film*> all() { std::vector<sppm_film*> v; for (auto& kv : by_gpu) v.push_back(kv.second); return v; }
    std::vector<double> sum() {
        std::vector<double> total((size_t)w * h * 3, 0.0);
        if (by_gpu.empty()) return total;
        if (by_gpu.size() == 1) {
            check(sppm_film_read(by_gpu.begin()->second, total.data()), by_gpu.begin()->second, "sppm_film_read");
        } else if (!reduce_on_host()) {
            std::vector<sppm_film*> v = all();
            check(sppm_film_allreduce(v.data(), (int)v.size()), v[0], "sppm_film_allreduce");
            check(sppm_film_read_total(v[0], total.data()), v[0], "sppm_film_read_total");
        } else {
            std::vector<double> part(total.size());
            for (auto& kv : by_gpu) {
                check(sppm_film_read(kv.second, part.data()), kv.second, "sppm_film_read");
                for (size_t e = 0; e < total.size(); e++) total[e] += part[e];
            }
        }
        return total;
    }
    // The picture the reference writes (Raytracer.h:460-470: toInt of the running mean, rows flipped), channels in PPM order.
    std::vector<unsigned char> tonemapped(double n_images) {
        std::vector<unsigned char> out((size_t)w * h * 3);
        if (by_gpu.size() == 1) {
            sppm_film* f = by_gpu.begin()->second;
            check(sppm_film_tonemap(f, n_images, 1, out.data()), f, "sppm_film_tonemap");
            return out;
        }
        if (!reduce_on_host()) {                 // several GPUs: their films meet by one all-reduce, tone-mapped where it landed
            std::vector<sppm_film*> v = all();
            check(sppm_film_allreduce(v.data(), (int)v.size()), v[0], "sppm_film_allreduce");
            check(sppm_film_tonemap_total(v[0], n_images, 1, out.data()), v[0], "sppm_film_tonemap_total");
            return out;
        }
        const std::vector<double> total = sum();
        for (int i1 = 0; i1 < h; i1++)
            for (int j = 0; j < w; j++)
                for (int k = 0; k < 3; k++)
                    out[((size_t)(h - 1 - i1) * w + j) * 3 + (2 - k)] = (unsigned char)toInt(total[((size_t)i1 * w + j) * 3 + k] / n_images);
        return out;
    }
};

// Checkpoint of the outer loop: "SPPMACC1", w, h (i32), iterations done (i64), jitter generator state (u64), w*h*3 doubles.
inline bool save_checkpoint(const std::string& path, Films& films, long long iterations) {
    const std::vector<double> total = films.sum();
    const std::string tmp = path + ".tmp";
    FILE* f = std::fopen(tmp.c_str(), "wb");
    if (!f) return false;
    const int32_t wh[2] = {films.w, films.h};
    const int64_t it = iterations;
    const uint64_t st = global_rng().state;
    bool ok = std::fwrite("SPPMACC1", 1, 8, f) == 8 && std::fwrite(wh, sizeof(wh), 1, f) == 1 && std::fwrite(&it, sizeof(it), 1, f) == 1 &&
              std::fwrite(&st, sizeof(st), 1, f) == 1 && std::fwrite(total.data(), sizeof(double), total.size(), f) == total.size();
    ok = std::fclose(f) == 0 && ok;
    return ok && std::rename(tmp.c_str(), path.c_str()) == 0;   // a reader never sees half a file
}
// Returns the number of iterations the checkpoint holds; the sum goes into the film of `gpu`.
inline long long load_checkpoint(const std::string& path, Films& films, int gpu) {
    FILE* f = std::fopen(path.c_str(), "rb");
    if (!f) throw std::runtime_error("SPPM_RESUME: cannot open " + path);
    char magic[8]; int32_t wh[2]; int64_t it = 0; uint64_t st = 0;
    std::vector<double> total((size_t)films.w * films.h * 3);
    const bool ok = std::fread(magic, 1, 8, f) == 8 && std::memcmp(magic, "SPPMACC1", 8) == 0 && std::fread(wh, sizeof(wh), 1, f) == 1 &&
                    std::fread(&it, sizeof(it), 1, f) == 1 && std::fread(&st, sizeof(st), 1, f) == 1 && wh[0] == films.w && wh[1] == films.h &&
                    std::fread(total.data(), sizeof(double), total.size(), f) == total.size();
    std::fclose(f);
    if (!ok || it < 0) throw std::runtime_error("SPPM_RESUME: " + path + " is not a checkpoint of a " + std::to_string(films.w) + "x" + std::to_string(films.h) + " image");
    sppm_film* film = films.on(gpu);
    Films::check(sppm_film_write(film, total.data()), film, "sppm_film_write");
    global_rng().state = st;
    return it;
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
    int PhotonCnt = 0;                    // photons handed to PhotonTrace so far (raytracer/Raytracer.h:84)
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
    // render() for the outer loop: the resolved image is added to `film` on the GPU instead of coming back into
    // camera->pic.  Only queues the photon pass and the add; sppm_synchronize reports what they ran into.
    void render_into(sppm_film* film) {
        sppm_host::Device& dev = sppm_host::Device::get(device);
        std::lock_guard<std::mutex> g(dev.mu);
        eye_locked(dev);
        const int cnt = (int)sppm_host::env_ll("SPPM_ROUNDS", 10000), onetime = (int)sppm_host::env_ll("SPPM_ONETIME", 100);
        dev.check(sppm_photon_pass(dev.ctx, (uint64_t)cnt, (uint32_t)onetime, 0, 0, seed, replica), "sppm_photon_pass");
        dev.check(sppm_film_add(dev.ctx, film), "sppm_film_add");
    }
    // One eye path through the library (raytracer/Raytracer.h:296-357): the ray is traced on the GPU from depth `dep` with
    // weight `wgt` for pixel (x = row, y = column), and the HitPoints it leaves are appended to Hvec, as the reference's
    // recursion pushes them.  They live on the host only: the photon pass of the GPU works on the HitPoints of GetHitPoint().
    void ViewTrace(const Ray& ray, int dep, Color wgt, int x, int y) {
        sppm_host::Device& dev = sppm_host::Device::get(device);
        std::lock_guard<std::mutex> g(dev.mu);
        if (!camera || !scene) throw std::runtime_error("RayTracer: camera and scene must be set");
        dev.ensure_scene(scene->objvec, scene->plvec);
        const uint32_t pixel = (uint32_t)(x * camera->w + y);
        const int32_t d = dep;
        sppm_hitpoint out[16];
        uint64_t n = 0;
        dev.check(sppm_view_trace(dev.ctx, 1, ray.org.val, ray.dir.val, &d, wgt.val, &pixel, camera->w, seed, replica, out, 16, &n), "sppm_view_trace");
        for (uint64_t i = 0; i < n; i++) {
            const sppm_hitpoint& h = out[i];
            Hvec.push_back(new HitPoint(Vec3(h.pos[0], h.pos[1], h.pos[2]), Vec3(h.n[0], h.n[1], h.n[2]), Vec3(h.v[0], h.v[1], h.v[2]),
                                        scene->objvec[h.obj], h.x, h.y, Vec3(h.wgt[0], h.wgt[1], h.wgt[2])));
        }
    }
    // One photon path through the library (raytracer/Raytracer.h:117-209): traced on the GPU from depth `dep` with `flux`, its
    // diffuse hits gathered into the HitPoints of the last GetHitPoint() there; Hvec is then read back, so that tao / newN of
    // its HitPoints show the photon as they do in the reference (costly per call: the reference's loop is PhotonMap()).
    // Photons are numbered in the order of the calls, which selects their roulette and bounce streams.
    void PhotonTrace(const Ray& ray, int dep, Color flux) {
        sppm_host::Device& dev = sppm_host::Device::get(device);
        std::lock_guard<std::mutex> g(dev.mu);
        const int32_t d = dep;
        const uint32_t id = (uint32_t)PhotonCnt++;
        dev.check(sppm_photon_trace(dev.ctx, 1, ray.org.val, ray.dir.val, &d, flux.val, &id, seed, replica), "sppm_photon_trace");
        fetch_hitpoints(dev);
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
        dev.check(sppm_set_radius_update(dev.ctx, (int32_t)sppm_host::env_ll("SPPM_INITIAL_N", 0), (uint32_t)sppm_host::env_ll("SPPM_UPDATE_ROUNDS", 1)),
                  "sppm_set_radius_update");
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
    // The outer SPPM loop (Raytracer.h:421-477): per iteration four replicas with the camera moved by
    // 0.00015 * a random unit vector, their images summed, the running mean tone-mapped and written out
    // (aaanew<i/10>.ppm).  The reference runs the four replicas on four OpenMP threads.  Here a pass is queued on a
    // GPU stream and the host does not wait for it, so one host thread hands the replicas to the library contexts
    // (SPPM_DEVICES) in turn: while it waits for the HitPoints of one context's eye pass, the photon passes queued on
    // the other contexts run.  Going round the contexts in a fixed order also keeps their passes a third of an
    // iteration apart, which is what lets the trace of one and the gather of another share the CUs (threads racing to
    // submit were measured 10 % slower).  The image sum stays on the GPUs (one film per GPU).
    void render(Scene scene, Camera camera) {
        const long long sppmtime = sppm_host::env_ll("SPPM_ITERS", 100000);
        const long long save_every = std::max(1LL, sppm_host::env_ll("SPPM_SAVE_EVERY", 1));
        const long long ckpt_every = std::max(1LL, sppm_host::env_ll("SPPM_CHECKPOINT_EVERY", 10));
        const char* ckpt = std::getenv("SPPM_CHECKPOINT");
        const char* resume = std::getenv("SPPM_RESUME");
        const std::vector<int> devices = sppm_host::env_devices();
        sppm_host::Films films;
        films.w = camera.w; films.h = camera.h;
        long long first = 0;
        if (resume && *resume) {
            first = sppm_host::load_checkpoint(resume, films, devices[0]);
            std::cerr << "resume:" << resume << " at iteration " << first << std::endl;
        }
        std::vector<Camera> cams(devices.size(), camera);   // one per context, assigned to again and again (their images are allocated once)
        for (long long i = first; i < sppmtime; i++) {
            for (int k = 0; k < 4; k++) {
                const size_t slot = (size_t)(i * 4 + k) % devices.size();   // context slot, see sppm_host::Device::get
                Camera& c = cams[slot];
                c = camera;
                c.MoveTo(camera.pos + 0.00015 * Vec3::GetUnitRandVec());
                c.lookAt(camera.look, 1.);
                RayTracer rt;
                rt.camera = &c;
                rt.scene = &scene;
                rt.device = (int)slot;
                rt.replica = (uint32_t)(i * 4 + k);
                rt.render_into(films.on(devices[slot]));
                std::cerr << "Thread " << k << " done" << std::endl;
            }
            const bool last = i + 1 == sppmtime;
            const bool save = (i + 1) % save_every == 0 || last, keep = ckpt && *ckpt && ((i + 1) % ckpt_every == 0 || last);
            if (!save && !keep) continue;
            // A picture or checkpoint: the host waits for the contexts and hears what the queued passes ran into (a queue
            // that overflowed is remembered until then).  In between, the next iterations' passes queue up behind and the
            // GPU never drains.
            for (size_t slot = 0; slot < devices.size(); slot++) {
                sppm_host::Device& dev = sppm_host::Device::get((int)slot);
                std::lock_guard<std::mutex> g(dev.mu);
                dev.check(sppm_synchronize(dev.ctx), "sppm_synchronize");
            }
            if (save) {
                std::string pic = std::string("aaanew") + std::to_string(i / 10) + std::string(".ppm");
                std::cerr << "save:" << pic << std::endl;
                if (!sppm_host::write_ppm(pic, camera.w, camera.h, films.tonemapped((i + 1) * 4.))) std::cerr << "cannot write " << pic << std::endl;
            }
            if (keep)
                if (!sppm_host::save_checkpoint(ckpt, films, i + 1)) std::cerr << "cannot write " << ckpt << std::endl;
        }
    }
};
