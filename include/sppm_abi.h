/*
 * sppm_abi.h — C-ABI of the MI355X SPPM hot path (libsppm_hip.so).
 *
 * The reference renderer (wondergo2017/raytrace3) has no FFI: everything is
 * inlined C++ in one translation unit.  This header is the lower boundary that
 * the drop-in C++ classes in include/raytrace3/ (RayTracer, SPPMRayTracer) call
 * instead of running the reference's CPU loops.  Every entry point names the
 * reference code it replaces (paths relative to the reference checkout).
 *
 * Conventions
 *   - extern "C", plain pointers and sizes, caller-owned host buffers.
 *   - every call returns 0 on success, <0 on error (SPPM_E_*); the message is
 *     available from sppm_last_error(ctx) and is owned by the context.
 *   - a context is bound to ONE GPU and must be used from one host thread at a
 *     time.  Multi-GPU = one context per GPU (one process per GPU under
 *     torch.distributed / RCCL, or one context per device in a C++ host).
 *   - geometry crosses the boundary as double (the reference's type); colours
 *     and material coefficients as double too.  The library converts to its
 *     device types (fp64 ray state / analytic hits / gather positions, fp32
 *     Newton core, colours, flux and accumulators).
 *   - no exceptions cross the boundary.
 */
#ifndef SPPM_ABI_H
#define SPPM_ABI_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SPPM_OK 0
#define SPPM_E_INVALID (-1)   /* bad argument / call order */
#define SPPM_E_DEVICE (-2)    /* HIP runtime error */
#define SPPM_E_CAPACITY (-3)  /* an internal queue or a caller buffer is too small */
#define SPPM_E_NOMEM (-4)

/* Object kinds: raytracer/Obj.h:55 (PlaneObj), :102 (SphereObj), raytracer/Bezier.h:188 (Bezier3Obj). */
#define SPPM_OBJ_PLANE 0
#define SPPM_OBJ_SPHERE 1
#define SPPM_OBJ_BEZIER 2

/* raytracer/Element.h:7-19 (Material).  refln is carried for completeness; the
 * reference never reads it on the hot path. */
typedef struct sppm_material {
    double diff[3];
    double refl[3];
    double refr[3];
    double refrn;
    double refln;
} sppm_material;

/* One entry of Scene::objvec (raytracer/Scene.h:96), flattened.
 *   plane : p0 = PlaneObj::P0, n = PlaneObj::norm (already normalised, Obj.h:60-62)
 *   sphere: p0 = SphereObj::org, radius
 *   bezier: first_patch/n_patches index sppm_scene::patches
 * texture = index into sppm_scene::textures or -1 (Obj::texture, Obj.h:9,50). */
typedef struct sppm_object {
    int32_t type;
    int32_t texture;
    double color[3];
    sppm_material material;
    double p0[3];
    double n[3];
    double radius;
    int32_t first_patch;
    int32_t n_patches;
} sppm_object;

/* raytracer/Element.h:43-75 (Texture): rows x cols texels, 3 bytes each in the
 * channel order of the reference's m_map (OpenCV BGR), value = byte / 255. */
typedef struct sppm_texture {
    int32_t rows;
    int32_t cols;
    const uint8_t* texels;
} sppm_texture;

/* raytracer/Light.h:4-17 (Light / SpotLight): isotropic point light. */
typedef struct sppm_light {
    double pos[3];
    double color[3];
} sppm_light;

/* Scene::objvec + Scene::plvec (raytracer/Scene.h:96-97) flattened.
 * patches: n_patches x 16 x 3 doubles, P[16] of Bezier3 (raytracer/Bezier.h:165)
 * in world space (after scale / Trans / +C of Bezier.h:213-229). */
typedef struct sppm_scene {
    int32_t n_objects;
    const sppm_object* objects;
    int32_t n_patches;
    const double* patches;
    int32_t n_textures;
    const sppm_texture* textures;
    int32_t n_lights;
    const sppm_light* lights;
} sppm_scene;

/* Camera state after Camera::lookAt (raytracer/Camera.h:32-54): pos, the scaled
 * view vector dir (= unit * .5/tan(fov/2)), the unit image axes du, dv and the
 * resolution.  Camera::emit (Camera.h:18-22) is evaluated on the device. */
typedef struct sppm_camera {
    double pos[3];
    double dir[3];
    double du[3];
    double dv[3];
    int32_t w;
    int32_t h;
} sppm_camera;

/* raytracer/Raytracer.h:47-80 (HitPoint) as returned for parity / debugging.
 * x = image row, y = image column (Raytracer.h:105-111).  node is the position
 * of the eye-path vertex in the ViewTrace recursion tree (root 0, reflection
 * child 2n+1, refraction child 2n+2); (x, y, node) identifies a HitPoint
 * independently of the order the device emitted it in. */
typedef struct sppm_hitpoint {
    double pos[3];
    double n[3];
    double v[3];
    double wgt[3];
    double tao[3];
    double R2;
    int32_t obj;
    int32_t x;
    int32_t y;
    int32_t node;
    int32_t N;
    int32_t newN;
} sppm_hitpoint;

/* Work counters of everything run since sppm_eye_pass.  candidates / accepted /
 * segments are the C / A / S of the gather byte model (40*C + 32*A + 64*S). */
typedef struct sppm_counters {
    uint64_t photons;        /* emitted photons */
    uint64_t segments;       /* photon path segments traced (NearCollide calls that were kept) */
    uint64_t deposits;       /* diffuse photon hits (radius searches in the reference) */
    uint64_t candidates;     /* (deposit, HitPoint) pairs with dist^2 < searchR2 */
    uint64_t accepted;       /* pairs that updated a HitPoint */
    uint64_t hitpoints;      /* |Hvec| */
    uint64_t newton_calls;   /* (ray, patch) Newton solves, eye + photon */
    uint64_t eye_segments;   /* eye path segments traced */
} sppm_counters;

/* HIP-event timings (ms), measured on the context's stream, of everything queued since
 * the last sppm_eye_pass call (sppm_get_timings waits for the stream). */
typedef struct sppm_timings {
    double eye_trace_ms;
    double grid_build_ms;
    double photon_trace_ms;
    double deposit_sort_ms;
    double gather_ms;
    double resolve_ms;
    int32_t gather_launches;
    int32_t trace_launches;
} sppm_timings;

typedef struct sppm_ctx sppm_ctx;

/* Create a context on one GPU.  Fails (SPPM_E_DEVICE) if no HIP device is usable:
 * there is no CPU fallback. */
int sppm_create(sppm_ctx** out, int device_id);
void sppm_destroy(sppm_ctx* ctx);
const char* sppm_last_error(const sppm_ctx* ctx);

/* Upload the flattened scene.  Replaces Scene::init's object graph as seen by
 * Scene::NearCollide (raytracer/Scene.h:165-182); patch and object AABBs are
 * rebuilt as Bezier3::read / Bezier3Obj do (raytracer/Bezier.h:167-175, :226-231).
 * Waits for the passes still queued on the context.  Scene tables of up to
 * 64 KiB (296 B per object: a 264-B record and the 32 B the plane / sphere tests
 * read, plus one spare such record; 244 B per Bezier patch; the reference's scene:
 * 9 objects, 32 patches, 10 KiB) are staged in LDS by the kernels, larger ones are
 * read from global memory (the photon kernel keeps 16.6 KiB more there — emission
 * slices, sixteen lights, a sine table — and on a device whose workgroups have
 * less than 80.6 KiB of LDS the staging area shrinks by what is missing).  Every patch is also cut into 4 x 4 sub-patches whose
 * boxes (384 B per patch) let the kernels skip solves that cannot find anything;
 * they ride in the staged tables while those stay within the 64 KiB (the
 * reference's scene: 22 KiB in all).  Capacity (SPPM_E_CAPACITY beyond it):
 * 4096 objects, 65536 patches, 2^30 texels. */
int sppm_set_scene(sppm_ctx* ctx, const sppm_scene* scene);

/* Pass 1: RayTracer::GetHitPoint + ViewTrace (raytracer/Raytracer.h:102-116,
 * :296-357) for every pixel, then the hash grid over HitPoints that replaces the
 * flann kd-tree build (Raytracer.h:370-380).  Clears tao / newN and the counters. */
int sppm_eye_pass(sppm_ctx* ctx, const sppm_camera* cam, uint64_t seed, uint32_t replica);

/* Pass 2: RayTracer::PhotonMap + PhotonTrace + the radius gather
 * (raytracer/Raytracer.h:117-209, :210-247).  The pass is rounds x n_lights x
 * per_round photons (Raytracer.h:220-233); photon g of that sequence is traced
 * iff photon_begin <= g < photon_begin + photon_count (photon_count 0 = all),
 * which is how ranks shard one pass.  Accumulates into tao / newN. */
int sppm_photon_pass(sppm_ctx* ctx, uint64_t rounds, uint32_t per_round,
                     uint64_t photon_begin, uint64_t photon_count,
                     uint64_t seed, uint32_t replica);

/* Resolve, raytracer/Raytracer.h:281-293: pic[x][y] (+)= tao / (PI * R2 * rounds * per_round)
 * with the rounds / per_round of the last photon pass, row-major [h][w][3].
 * accumulate = 0 overwrites pic, 1 adds to it (the image sum of Raytracer.h:449-458). */
int sppm_resolve(sppm_ctx* ctx, double* pic_hw3, int accumulate);

/* Same, into a caller-provided DEVICE buffer of h*w*3 floats (overwritten) so a
 * collective (RCCL all-reduce of the flux image) can run on it in place. */
int sppm_resolve_device(sppm_ctx* ctx, float* dev_pic_hw3);

/* The image sum and its tone-mapped copy, SPPMRayTracer::render (raytracer/Raytracer.h:449-470), kept on
 * the GPU: a film is the fp64 `camera.pic` of that loop, [h][w][3], zero when created. */
typedef struct sppm_film sppm_film;
int sppm_film_create(sppm_film** out, int device_id, int w, int h);
void sppm_film_destroy(sppm_film* film);
const char* sppm_film_last_error(const sppm_film* film);
int sppm_film_clear(sppm_film* film);

/* film += resolve of the context's last photon pass (Raytracer.h:281-293 into :449-458), queued on the
 * context's stream; context and film must be on the same GPU and of the same resolution.  Several
 * contexts may add to one film, each from its own host thread. */
int sppm_film_add(sppm_ctx* ctx, sppm_film* film);

/* out[h - 1 - i][j][c] = toInt(film[i][j][k] / n_images) with c = k (rgb = 0: the cv::Mat the reference
 * passes to cv::imwrite, Raytracer.h:460-470, toInt of :11-13) or c = 2 - k (rgb = 1: the same picture
 * with the channels in PPM order).  Waits for every add queued before the call.  Bytes are those the
 * host's exp / pow give: elements the device library could round the other way are recomputed on the host. */
int sppm_film_tonemap(sppm_film* film, double n_images, int rgb, uint8_t* out_hw3);

/* The fp64 sum itself, out of and into the film (a checkpoint of the running mean and its resume). */
int sppm_film_read(sppm_film* film, double* pic_hw3);
int sppm_film_write(sppm_film* film, const double* pic_hw3);

/* The image sum of SPPMRayTracer::render (raytracer/Raytracer.h:449-458) when the replicas of an iteration ran on several GPUs
 * of this process: ONE all-reduce (RCCL, fp64 sum, over xGMI) over n films, one per GPU (distinct devices, equal
 * resolutions), instead of reading every film back and adding on the host.  Each film keeps its own sum and receives the
 * sum over all films in a second buffer, which sppm_film_tonemap_total / sppm_film_read_total read (on any of the films:
 * an all-reduce leaves the total on every GPU).  n = 1 is allowed (the total is then the film's own sum, still through
 * RCCL).  RCCL is opened at run time (librccl.so.1); SPPM_E_DEVICE if it cannot be.  Waits for the adds queued before the
 * call.  The communicators are made on first use for a set of GPUs (ncclCommInitAll) and kept for the process. */
int sppm_film_allreduce(sppm_film* const* films, int n);
int sppm_film_tonemap_total(sppm_film* film, double n_images, int rgb, uint8_t* out_hw3);
int sppm_film_read_total(sppm_film* film, double* pic_hw3);

/* sppm_photon_pass, sppm_resolve_device and sppm_film_add only QUEUE their kernels on the context's stream
 * (the reference is synchronous; a GPU host wants to keep queueing).  sppm_synchronize waits
 * for everything queued and returns the error, if any, that a queued pass ran into (a ray /
 * task / deposit queue overflow: SPPM_E_CAPACITY).  sppm_resolve, sppm_get_counters and the
 * next sppm_eye_pass report such an error as well.  No counterpart in the reference. */
int sppm_synchronize(sppm_ctx* ctx);

/* Counters and timings normally restart at every sppm_eye_pass.  accumulate = 1 clears them now and
 * lets them run on over the following passes (a benchmark reads them once after K iterations instead
 * of waiting for the stream after each); accumulate = 0 clears them and restores the default. */
int sppm_stats_mode(sppm_ctx* ctx, int accumulate);

/* The progressive radius update, HitPoint::update (raytracer/Raytracer.h:69-79) swept over all HitPoints after
 * every round of PhotonMap (:237-245).  As written in the reference it never runs: N starts at 0 and the guard
 * `if (N <= 0 || newN <= 0) return;` leaves at once, so radii stay at INIT_R2 — and that is this library's
 * default (initial_n = 0).  initial_n > 0 is a DEVIATION offered as a knob: HitPoints of the following eye passes
 * start with that N, and sppm_photon_pass then runs update() exactly as written after every
 * rounds_per_update rounds (the reference's cadence is 1; larger values trade fidelity for larger batches).
 * R2 only shrinks, the search radius stays INIT_R2 (updateR2 is commented out, Raytracer.h:246).  A pass with the
 * update on cannot be sharded by photon range: each update needs all photons of its rounds. */
int sppm_set_radius_update(sppm_ctx* ctx, int32_t initial_n, uint32_t rounds_per_update);

/* A context queues its work on a HIP stream of its own.  sppm_set_stream makes it use the caller's stream
 * (a hipStream_t of the context's GPU; NULL: back to its own) from the next call on, after waiting for what it
 * had queued so far, so that a host can order the library's passes with its own GPU work (a collective on the
 * image sppm_resolve_device wrote) by stream order instead of waiting on the host.  The stream stays the
 * caller's: it must outlive the context or be replaced first.  No counterpart in the reference. */
int sppm_set_stream(sppm_ctx* ctx, void* hip_stream);
/* The hipStream_t the context queues on at present (its own unless sppm_set_stream replaced it), for a host
 * that wants to put its own work behind the library's on the same stream.  No counterpart in the reference. */
void* sppm_get_stream(const sppm_ctx* ctx);

/* Copy out up to cap HitPoints; *n receives |Hvec|. */
int sppm_get_hitpoints(sppm_ctx* ctx, sppm_hitpoint* out, uint64_t cap, uint64_t* n);
int sppm_get_counters(sppm_ctx* ctx, sppm_counters* out);
int sppm_get_timings(sppm_ctx* ctx, sppm_timings* out);

/* Tallies of the library's own making, over the same span as sppm_counters (no counterpart in the reference; they
 * depend on how the passes are batched and laid out, which sppm_counters does not):
 *   stored_deposits  deposit records written for the gather: diffuse photon hits that some HitPoint can reach
 *   pair_tests       (deposit, HitPoint) pairs the gather evaluated; candidates / pair_tests is its hit rate
 *   device_bytes     device memory the context holds now (scene tables, HitPoints, grid, deposit buffers, queues)
 *   deposit_room     deposit slots of the direct-placement layout in use (0: none) */
typedef struct sppm_device_stats {
    uint64_t stored_deposits;
    uint64_t pair_tests;
    uint64_t device_bytes;
    uint64_t deposit_room;
} sppm_device_stats;
int sppm_get_device_stats(sppm_ctx* ctx, sppm_device_stats* out);

/* Collision record of Scene::NearCollide for the debug entry point below
 * (raytracer/Element.h:20-38).  obj = -1 when nothing was hit. */
typedef struct sppm_collision {
    double pos[3];
    double n[3];
    double color[3];
    double dist;
    int32_t obj;
    int32_t inside;
} sppm_collision;

/* Parity hook: Scene::NearCollide (raytracer/Scene.h:165-182) for n arbitrary
 * rays (org, dir: n x 3 doubles each).  Ray i uses the Newton random stream of
 * (seed, replica, path = i, node = 0) of the eye pass. */
int sppm_debug_near_collide(sppm_ctx* ctx, uint64_t n, const double* org, const double* dir,
                            uint64_t seed, uint32_t replica, sppm_collision* out);

/* sppm_debug_near_collide for callers that cannot allocate GPU memory (code under HIP graph capture rules, tight serving
 * loops, harnesses that forbid hipMalloc): NO COUNTERPART IN THE REFERENCE.  Same arguments, same kernel, same random
 * streams and bit-identical results, but the call makes no hipMalloc / hipFree (the allocating entry point makes three
 * of each per call): rays and collision records are staged through a static device array of the library (2.2 MiB, which
 * every GPU the library runs on holds until the process ends), whose launch tallies a fill kernel clears first.
 * Limit: n <= SPPM_NOALLOC_MAX_RAYS per call, SPPM_E_CAPACITY beyond it.  Every call on a GPU shares that GPU's array:
 * calls from several host threads of one process take turns on a mutex; the array keeps the last call's rays and records. */
#define SPPM_NOALLOC_MAX_RAYS 16384
int sppm_near_collide_noalloc(sppm_ctx* ctx, uint64_t n, const double* org, const double* dir,
                              uint64_t seed, uint32_t replica, sppm_collision* out);

/* RayTracer::ViewTrace(ray, dep, wgt, x, y) (raytracer/Raytracer.h:296-357) for n eye rays of the caller's making instead
 * of Camera::emit's: ray i (org, dir: n x 3 doubles) starts at depth dep[i] (null: 0) with weight wgt[i] (n x 3) for pixel
 * pixel[i] = row * w + column, whose random streams its Newton solves draw from.  The HitPoints the paths leave are
 * returned (x = row, y = column, node = position in the path's tree under the given ray); the context's own HitPoints,
 * those of sppm_eye_pass, are not touched.  At most 2^22 rays per call. */
int sppm_view_trace(sppm_ctx* ctx, uint64_t n, const double* org, const double* dir, const int32_t* dep, const double* wgt,
                    const uint32_t* pixel, int32_t w, uint64_t seed, uint32_t replica,
                    sppm_hitpoint* out, uint64_t cap, uint64_t* n_out);

/* RayTracer::PhotonTrace(ray, dep, flux) (raytracer/Raytracer.h:117-209) for n photons of the caller's making instead of
 * Light::emit's: photon i starts as the ray (org, dir) at depth dep[i] (null: 0) with flux[i] (n x 3); its roulette and
 * bounce draws come from the streams of photon index id[i].  The diffuse hits of the paths are gathered into the HitPoints
 * of the last sppm_eye_pass (tao, newN) exactly as a photon pass gathers its own; sppm_counters::photons does not count
 * them.  Queued like a photon pass.  At most 2^22 photons per call; refused while the radius update is on. */
int sppm_photon_trace(sppm_ctx* ctx, uint64_t n, const double* org, const double* dir, const int32_t* dep, const double* flux,
                      const uint32_t* id, uint64_t seed, uint32_t replica);

/* Parity hook: Bezier3::newton (raytracer/Bezier.h:112-159) for n (ray, patch)
 * pairs; out_tuv receives (t, u, v) or (1e20, 1e20, 1e20). */
int sppm_debug_newton(sppm_ctx* ctx, uint64_t n, const double* org, const double* dir,
                      const int32_t* patch, uint64_t seed, uint32_t replica, double* out_tuv);

/* Parity hook: the deposit records (diffuse photon hits = the radius-search queries of
 * raytracer/Raytracer.h:136-146) of the photons [photon_begin, +photon_count) of a pass, in
 * device order.  out: cap x 11 doubles = photon index, bounce, pos[3], n[3], flux[3] (flux is
 * the photon's flux before the hit's colour is applied).  Needs a prior sppm_eye_pass; does not
 * touch the HitPoints or the counters. */
int sppm_debug_photon_deposits(sppm_ctx* ctx, uint64_t rounds, uint32_t per_round,
                               uint64_t photon_begin, uint64_t photon_count,
                               uint64_t seed, uint32_t replica,
                               double* out, uint64_t cap, uint64_t* n);

#ifdef __cplusplus
}
#endif
#endif /* SPPM_ABI_H */
