// Bridge.h — the host side's way to the GPU library (no counterpart in the reference, whose classes compute on the host):
// sppm_host::init() and one library context per slot of SPPM_DEVICES.  Included by the drop-in headers whose members
// call the library (Bezier.h, Scene.h, Raytracer.h).
#pragma once
#include <cstdlib>
#include <map>
#include <memory>
#include <mutex>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

#include "../sppm_abi.h"

class Obj;
class Light;

namespace sppm_host {

// sppm_host::init(): what the host side sets up before the first GPU call of the process.  At present one thing: a
// hardware queue per context.  HIP's default of 4 queues per process folds the streams of four contexts and a film onto
// shared queues, whose kernels then run one after the other (measured: 789 instead of 1015 Mphotons/s), so
// GPU_MAX_HW_QUEUES is set to 8 (SPPM_HW_QUEUES picks another value, 0 leaves the environment alone).  The HIP runtime
// reads the variable when it initialises, i.e. at the first HIP call of the process: init() must come before that.
// The drop-in classes call it themselves on their way to the first context (Device::get below), which is in time for
// a program like the reference's main.cpp, whose first GPU work is theirs; a host that makes HIP calls of its own
// earlier calls sppm_host::init() first, or exports GPU_MAX_HW_QUEUES itself.  Nothing happens when a header is merely
// included.  Returns the number of queues asked for (0: environment left alone).
inline int init() {
    static const int queues = [] {
        const char* e = std::getenv("SPPM_HW_QUEUES");
        const std::string v = (e && *e) ? e : "8";
        if (v == "0") return 0;
        setenv("GPU_MAX_HW_QUEUES", v.c_str(), 1);
        return std::atoi(v.c_str());
    }();
    return queues;
}

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
    // The context's copy of the scene is the one with this tag, or becomes it (a tag is a hash of a flattened scene's content).
    void use_scene(const sppm_scene& scene, uint64_t tag) {
        if (tag == scene_tag) return;
        check(sppm_set_scene(ctx, &scene), "sppm_set_scene");
        scene_tag = tag;
    }
    // Flattens the live objects and lights (defined at the end of Obj.h, after the classes it reads) and calls use_scene.
    void ensure_scene(const std::vector<Obj*>& objvec, const std::vector<Light*>& plvec);
    static Device& get(int slot_index) {
        static std::mutex reg_mu;
        static std::map<int, std::unique_ptr<Device>> reg;
        static const std::vector<int> devices = env_devices();

        init();
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

}  // namespace sppm_host
