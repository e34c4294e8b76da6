// Camera.h — drop-in for the reference's raytracer/Camera.h: pinhole camera with the image
// `pic` (h rows of w Color).  lookAt builds dir / du / dv exactly as the reference does
// (up = (0,0,1), fov 50 degrees, no aspect ratio; raytracer/Camera.h:32-45); copies take the
// parameters and a fresh zeroed image, not the pixels (Camera.h:55-95).
// SPPM_W / SPPM_H in the environment override the resolution given to init(), so an
// unmodified main.cpp can be run at other sizes.
#pragma once
#include <cstdlib>

#include "Element.h"
#include "Vec3.h"

class Camera {
public:
    Vec3 pos;
    Vec3 look;
    double len_w = 0, len_h = 0;
    int w = 0, h = 0;
    Vec3 dir;
    Vec3 du, dv;
    Color** pic = nullptr;
    Camera() {}
    static const int DEFAULT_LENS = 100;
    static const int DEFAULT_HEIGHT = 1024;
    static const int DEFAULT_WIDTH = 1024;

    Ray emit(int y, int x) const {
        Vec3 d = du * ((x + 0.5) / w - 0.5) + dv * ((y + 0.5) / h - 0.5) + dir;
        return Ray(pos, d.GetUnit());
    }
    void init(int w_ = DEFAULT_WIDTH, int h_ = DEFAULT_HEIGHT) {
        w = w_;
        h = h_;
        if (const char* e = std::getenv("SPPM_W")) if (std::atoi(e) > 0) w = std::atoi(e);
        if (const char* e = std::getenv("SPPM_H")) if (std::atoi(e) > 0) h = std::atoi(e);
    }
    void MoveTo(const Vec3& pos_) { pos = pos_; }
    void lookAt(const Vec3& P, double scale) {
        look = P;
        const Vec3 up(0, 0, 1);
        w = (int)(w * scale);
        h = (int)(h * scale);
        dir = (look - pos).GetUnit();
        du = dir.cross(up);
        du.Normalize();
        dv = -(dir.cross(du));
        dv.Normalize();
        double fov = 50. * PI / 180.;
        dir *= .5 / std::tan(fov / 2.);
        if (!pic) alloc();
    }
    Camera(const Camera& c) { copy_params(c); alloc(); }
    Camera& operator=(const Camera& c) {
        if (this == &c) return *this;
        if (pic && (w != c.w || h != c.h)) release();
        copy_params(c);
        if (!pic) alloc();
        return *this;
    }
    void clear() {
        for (int i = 0; i < h; i++) for (int j = 0; j < w; j++) pic[i][j] = Color();
    }
    ~Camera() { release(); }

private:
    int alloc_h = 0;
    void copy_params(const Camera& c) {
        look = c.look; pos = c.pos; len_w = c.len_w; len_h = c.len_h;
        w = c.w; h = c.h; dir = c.dir; du = c.du; dv = c.dv;
    }
    void alloc() {
        if (w <= 0 || h <= 0) return;
        pic = new Color*[h];
        for (int i = 0; i < h; i++) pic[i] = new Color[w];
        alloc_h = h;
    }
    void release() {
        if (!pic) return;
        for (int i = 0; i < alloc_h; i++) delete[] pic[i];
        delete[] pic;
        pic = nullptr;
        alloc_h = 0;
    }
};
