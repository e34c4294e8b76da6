// bezier_members.cpp — the members of AABBbox and Bezier3 a host may call itself (raytracer/Bezier.h:20-51, :69-186),
// through the drop-in header.
//
//   bezier_members <out file> [gpu]
// Loads teapot.bpt as the reference's scene places it (scale 4 at (20, 0, 120)) and writes doubles to <out file>:
//   header   n_patches, n_uv, n_rays
//   for every patch, its 48 control-point coordinates
//   for every patch and every (u, v) of an n_uv x n_uv grid over [0, 1]^2: S, S_u, S_v (evalPatchPoint / DerivU / DerivV)
//   for the first four control points of every patch as a curve, at the same u: evalCurvePoint, evalCurveDeriv
//   for every ray (origin, direction: 6 doubles): AABBbox::intersect of the ray with each patch's box (0 / 1)
//   with "gpu": for every ray and the patch it was aimed at, Bezier3::intersect's (t, u, v)   [needs a GPU]
// A test compares the file with the CPU oracle's evaluators and box test, and checks the solves' own conditions.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "Bezier.h"

int main(int argc, char** argv) {
    if (argc < 2) { std::fprintf(stderr, "usage: %s out.f64 [gpu]\n", argv[0]); return 2; }
    const bool gpu = argc > 2 && std::string(argv[2]) == "gpu";
    Mat33 Trans, Trans2;
    Trans(0, 0) = 1; Trans(1, 2) = 1; Trans(2, 1) = 1;
    const double theta = (90. / 180.) * PI;
    Trans2(0, 0) = std::cos(theta); Trans2(0, 2) = std::sin(theta); Trans2(1, 1) = 1;
    Trans2(2, 0) = -std::sin(theta); Trans2(2, 2) = std::cos(theta);
    Bezier3Obj teapot(Vec3(20, 0, 120), 4, "teapot.bpt", Vec3(0, 1, 1) * .999, Material(Vec3(.75, .75, .75), Vec3(), Vec3(), 0), Trans2 * Trans);
    const int np = (int)teapot.patch_count(), n_uv = 9, n_rays = 400;
    if (np == 0) { std::fprintf(stderr, "teapot.bpt not found\n"); return 1; }
    std::vector<double> out = {(double)np, (double)n_uv, (double)n_rays};
    auto put = [&](const Vec3& v) { for (int k = 0; k < 3; k++) out.push_back(v.val[k]); };
    for (int p = 0; p < np; p++) for (int k = 0; k < 16; k++) put(teapot.patch(p).P[k]);
    for (int p = 0; p < np; p++)
        for (int a = 0; a < n_uv; a++)
            for (int b = 0; b < n_uv; b++) {
                const double u = a / (n_uv - 1.0), v = b / (n_uv - 1.0);
                const Vec3* P = teapot.patch(p).P;
                put(Bezier3::evalPatchPoint(P, u, v)); put(Bezier3::evalPatchDerivU(P, u, v)); put(Bezier3::evalPatchDerivV(P, u, v));
            }
    for (int p = 0; p < np; p++)
        for (int a = 0; a < n_uv; a++) {
            const double u = a / (n_uv - 1.0);
            put(Bezier3::evalCurvePoint(teapot.patch(p).P, u)); put(Bezier3::evalCurveDeriv(teapot.patch(p).P, u));
        }
    // rays from around the camera of the reference's main.cpp towards points of the patches, a few from inside a box and a
    // few along an axis (a zero direction component is skipped by the box test)
    unsigned long long lcg = 12345;
    auto rnd = [&]() { lcg = lcg * 6364136223846793005ull + 1442695040888963407ull; return (double)(lcg >> 11) / 9007199254740992.0; };
    std::vector<Vec3> orgs, dirs;
    std::vector<int> aimed;
    for (int r = 0; r < n_rays; r++) {
        const int p = r % np;
        const Bezier3& b = teapot.patch(p);
        Vec3 target = Bezier3::evalPatchPoint(b.P, rnd(), rnd()) + Vec3(rnd() - .5, rnd() - .5, rnd() - .5) * (r % 3 == 0 ? 4.0 : 0.0);
        Vec3 org = r % 17 == 5 ? (b.aabb.pmin + b.aabb.pmax) * 0.5 : Vec3(50 + 40 * (rnd() - .5), 35 + 30 * (rnd() - .5), 230 - 60 * rnd());
        Vec3 dir = (target - org).GetUnit();
        if (r % 23 == 7) { dir = Vec3(0, 0, -1); org = Vec3(target.val[0], target.val[1], 200); }
        orgs.push_back(org); dirs.push_back(dir); aimed.push_back(p);
        put(org); put(dir);
        for (int q = 0; q < np; q++) out.push_back(teapot.patch(q).aabb.intersect(org, dir) ? 1.0 : 0.0);
    }
    if (gpu) {
        try {
            for (int r = 0; r < n_rays; r++) put(teapot.patch(aimed[r]).intersect(orgs[r], dirs[r]));
            bool refused = false;
            try { (void)teapot.patch(0).newton(orgs[0], dirs[0], 7); } catch (const std::invalid_argument&) { refused = true; }
            if (!refused) { std::fprintf(stderr, "newton accepted an iteration count the GPU solver does not run\n"); return 1; }
        } catch (const std::exception& e) {
            std::fprintf(stderr, "%s\n", e.what());
            return 1;
        }
    }
    FILE* f = std::fopen(argv[1], "wb");
    if (!f || std::fwrite(out.data(), sizeof(double), out.size(), f) != out.size()) { std::fprintf(stderr, "cannot write %s\n", argv[1]); return 1; }
    std::fclose(f);
    std::printf("OK %d patches, %zu doubles\n", np, out.size());
    return 0;
}
