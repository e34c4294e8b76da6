// Element.h — drop-in for the reference's raytracer/Element.h (Ray, Material, Collision,
// Texture; raytracer/Element.h:6-75) without OpenCV.  Texture reads binary PPM (P6) files;
// for any other file name (the reference's .jpg names) it looks for a .ppm of the same
// base name, in the working directory and then in SPPM_DATA_DIR, and otherwise builds a seeded procedural image, because JPEG decoding is not
// part of the accelerated path.  Texel storage is rows x cols x 3 bytes in the reference's
// channel order (OpenCV BGR), value = byte / 255.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <string>
#include <vector>

#include "Vec3.h"

namespace sppm_host {
// Directory searched for .bpt and texture files that are not found relative to the working directory.
inline std::string data_dir() {
    if (const char* e = std::getenv("SPPM_DATA_DIR")) return e;
#ifdef SPPM_DEFAULT_DATA_DIR
    return SPPM_DEFAULT_DATA_DIR;
#else
    return ".";
#endif
}
}  // namespace sppm_host

class Obj;
struct Ray {
    Vec3 org, dir;
    Ray() {}
    Ray(Vec3 o, Vec3 d) : org(o), dir(d) {}
};

class Material {
public:
    Material(Vec3 d, Vec3 s, Vec3 r, double rr = 1.5, double rl = 1) : diff(d), refl(s), refr(r), refrn(rr), refln(rl) {}
    Vec3 diff, refl, refr;
    double refrn, refln;
    bool IsDiff() const { return !IsZero(diff); }
    bool IsRefl() const { return !IsZero(refl); }
    bool IsRefr() const { return !IsZero(refr); }
};

class Collision {
public:
    bool IsValid() const { return obj != nullptr; }
    bool IsFront() const { return !inside; }
    Vec3 Pos, N, From;
    Obj* obj;
    Color color = Vec3();
    double dist;
    explicit Collision(const Vec3& pos = Vec3(), const Vec3& from = Vec3(), double d = 0, Obj* o = nullptr,
                       const Vec3& n = Vec3(), const Vec3& c = Vec3())
        : Pos(pos), N(n), From(from), obj(o), color(c), dist(d) {}
    bool inside = false;
};
const Collision NoneColl = Collision();
enum OBJ_REL { INSIDE, OUTSIDE, ON, UNKNOWN };

struct Texture {
    int rows = 0, cols = 0;
    std::vector<uint8_t> texels;  // rows * cols * 3, BGR

    explicit Texture(const std::string& fileName) {
        if (!load_ppm(fileName)) {
            std::string alt = fileName;
            size_t dot = alt.find_last_of('.');
            if (dot != std::string::npos) alt = alt.substr(0, dot);
            alt += ".ppm";
            if (!load_ppm(alt) && !load_ppm(sppm_host::data_dir() + "/" + alt)) {
                procedural(fileName);
                std::cerr << "Texture: " << fileName << " not readable as PPM, using a procedural stand-in" << std::endl;
            }
        }
        std::cout << "Texture:" << fileName << "::row:" << rows << " cols:" << cols << std::endl;
    }
    Texture(int r, int c, const uint8_t* bgr) : rows(r), cols(c), texels(bgr, bgr + (size_t)r * c * 3) {}

    // A hash of the size and the texels, by which the host bridge sees whether a GPU context's copy of the scene is still
    // current.  It is computed once and kept while the texel array stays where it is and as large as it is (several
    // million texels per stock texture, asked for once per pass or per RayTracer::ViewTrace call); a host that edits
    // texels IN PLACE says so with touch().
    uint64_t content_hash() const {
        if (!(hash_valid_ && hashed_at_ == texels.data() && hashed_n_ == texels.size() && hashed_rows_ == rows && hashed_cols_ == cols)) {
            uint64_t h = 1469598103934665603ull;
            auto mix = [&](uint64_t v) { h = (h ^ v) * 1099511628211ull; };
            mix((uint64_t)rows); mix((uint64_t)cols);
            const uint8_t* b = texels.data();
            size_t i = 0, n = texels.size();
            for (; i + 8 <= n; i += 8) { uint64_t w; std::memcpy(&w, b + i, 8); mix(w); }
            for (; i < n; i++) mix(b[i]);
            hash_ = h; hashed_at_ = texels.data(); hashed_n_ = n; hashed_rows_ = rows; hashed_cols_ = cols; hash_valid_ = true;
        }
        return hash_;
    }
    void touch() { hash_valid_ = false; }

    // Wrap-around bilinear lookup with the reference's index arithmetic (Element.h:61-72).
    Color colorUV(double u, double v) const {
        double row = (u - std::floor(u)) * rows, col = (v - std::floor(v)) * cols;
        int r1 = (int)std::floor(row + 1e-10), r2 = r1 + 1;
        int c1 = (int)std::floor(col + 1e-10), c2 = c1 + 1;
        double fr = r2 - row, fc = c2 - col;
        r1 = r1 < 0 ? rows - 1 : (r1 >= rows ? 0 : r1);
        c1 = c1 < 0 ? cols - 1 : (c1 >= cols ? 0 : c1);
        r2 = r2 < rows ? r2 : 0;
        c2 = c2 < cols ? c2 : 0;
        return texel(r1, c1) * fr * fc + texel(r1, c2) * fr * (1 - fc) + texel(r2, c1) * (1 - fr) * fc +
               texel(r2, c2) * (1 - fr) * (1 - fc);
    }

private:
    mutable uint64_t hash_ = 0;
    mutable const uint8_t* hashed_at_ = nullptr;
    mutable size_t hashed_n_ = 0;
    mutable int hashed_rows_ = 0, hashed_cols_ = 0;
    mutable bool hash_valid_ = false;
    Vec3 texel(int r, int c) const {
        const uint8_t* p = &texels[((size_t)r * cols + c) * 3];
        return Vec3(p[0] / 255.0, p[1] / 255.0, p[2] / 255.0);
    }
    bool load_ppm(const std::string& path) {
        std::ifstream f(path, std::ios::binary);
        if (!f) return false;
        std::string magic;
        f >> magic;
        if (magic != "P6") return false;
        int w = 0, h = 0, maxv = 0;
        auto skip = [&]() {
            for (;;) {
                f >> std::ws;
                if (f.peek() == '#') { std::string line; std::getline(f, line); } else break;
            }
        };
        skip(); f >> w; skip(); f >> h; skip(); f >> maxv;
        f.get();
        if (!f || w <= 0 || h <= 0 || maxv != 255) return false;
        std::vector<uint8_t> rgb((size_t)w * h * 3);
        f.read((char*)rgb.data(), (std::streamsize)rgb.size());
        if (!f) return false;
        rows = h; cols = w;
        texels.resize(rgb.size());
        for (size_t k = 0; k < (size_t)w * h; k++) {  // PPM is RGB, the reference's map is BGR
            texels[3 * k + 0] = rgb[3 * k + 2];
            texels[3 * k + 1] = rgb[3 * k + 1];
            texels[3 * k + 2] = rgb[3 * k + 0];
        }
        return true;
    }
    void procedural(const std::string& name) {
        uint32_t h = 2166136261u;
        for (char ch : name) h = (h ^ (uint8_t)ch) * 16777619u;
        rows = cols = 256;
        texels.resize((size_t)rows * cols * 3);
        int base[3] = {60 + (int)(h % 120), 60 + (int)((h >> 8) % 120), 60 + (int)((h >> 16) % 120)};
        for (int r = 0; r < rows; r++)
            for (int c = 0; c < cols; c++) {
                int checker = ((r / 32) + (c / 32)) & 1;
                double wave = 0.5 + 0.5 * std::sin(2 * 3.14159265358979 * (r * 3 + c * 5) / 256.0);
                for (int k = 0; k < 3; k++) {
                    int v = base[k] + 40 * checker + (int)(30 * wave);
                    texels[((size_t)r * cols + c) * 3 + k] = (uint8_t)std::min(255, std::max(0, v));
                }
            }
    }
};
