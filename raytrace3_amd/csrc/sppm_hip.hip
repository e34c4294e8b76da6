// sppm_hip.hip — HIP kernels (gfx950) and the C-ABI of include/sppm_abi.h.
//
// Pipeline of one replica (everything is queued on the context's stream; the only host wait is the one of the
// eye pass, which needs the HitPoint count):
//   eye pass   : k_trace<MODE_EYE> (persistent waves, one lane per eye path, pixels dealt out in a scattered
//                order, Bezier patches solved by the whole wave) -> raw HitPoints, histogrammed by grid bucket
//                as they are written -> k_scan_* -> k_hp_scatter (HitPoints sorted by bucket) -> k_hp_occupancy
//                (which buckets a HitPoint can reach)
//   photon pass: per batch  k_trace<MODE_PHOTON> (lanes refilled from a photon counter; a diffuse hit that some
//                HitPoint can reach becomes a deposit record) -> deposits grouped by bucket, either by
//                k_scan_* + k_dep_scatter or, once a layout exists, written in place by the trace kernel itself
//                -> k_gather (+ k_gather_split for heavy cells and for deposits that found their bucket
//                full): the HitPoints of a cell share one sweep over the deposits of the 27 cells
//                around them, rows of 64 deposits held in registers; no atomics on HitPoints in the main kernel
//   resolve    : k_resolve (tao -> flux image) or k_film_add (-> fp64 image sum), k_tonemap
// Several contexts on one GPU (SPPM iterations in flight) launch their persistent grids small enough for a trace
// workgroup of one and gather workgroups of another to be resident on a CU together (sppm_ctx::trace_blocks_per_cu).
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>      // types and prototypes only: the library is opened at run time (film_rccl), not linked
#include <dlfcn.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstddef>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/sppm_abi.h"
#include "sppm_device.h"

using namespace sppm;

// ------------------------------------------------------------------ device records
struct __attribute__((aligned(32))) PosKey { double x, y, z; uint64_t key; };
// bucket: grid bucket in bits 0..23, path vertex (bounce) in bits 24..31; photon: global photon index
struct __attribute__((aligned(32))) DepB { float nx, ny, nz, fr, fg, fb; uint32_t bucket, photon; };
// One deposit = one 64-byte record (position + cell key | normal, flux, bucket, photon): one scattered
// write when it is sorted, four consecutive 16-byte loads per lane when a row of them is gathered.
struct __attribute__((aligned(64))) Deposit { PosKey a; DepB b; };
struct RawHit { double pos[3]; float n[3]; float v[3]; float wgt[3]; int pix, node, obj; };
struct HpMeta { float v[3]; int node, obj, N; };
// A parked ray: either one whose Bezier patch solves were deferred to k_newton (task_count > 0;
// adist / aobj / ainside hold the nearest plane-or-sphere hit) or a fresh ray still to be traced
// (task_count == 0: the refraction branch of an eye path that also reflects).
struct __attribute__((aligned(16))) ParkRec {
    double o[3], d[3];
    double adist;
    float carry[3];        // eye: path weight; photon: flux
    uint32_t id;           // pixel index or global photon index
    int dep, node, aobj, ainside;
    uint32_t task_first, task_count, sub, pad;
};
struct NewtonTask { uint32_t park, patch; };

struct SceneParams {
    const char* blob;
    int blob_bytes, off_box, off_cp, off_pobj, off_sub, off_aobj, n_objs, n_patches;   // off_sub < 0: the sub-patch boxes are not in the blob
    const uchar4* texels;
    const float* pw_q;      // Newton tables: power-basis coefficients around the patch centre, n_patches x 48
    const double* pw_c;     // and the centres, n_patches x 3
    const float* sub_box;   // boxes of the 4 x 4 sub-patches of every patch (SceneView::sub_box) where the blob has no room for them, or null
    int n_bez;              // SceneView::n_bez, bez_idx
    unsigned long long bez_idx;
    int n_runs;             // SceneView::n_runs, runs_lo, runs_hi
    unsigned long long runs_lo, runs_hi;
};

struct DevCounters {
    unsigned long long photons, segments, deposits, candidates, accepted, newton_calls, eye_segments;
    unsigned long long stored;       // deposits written for the gather (diffuse hits that some HitPoint can reach)
    unsigned long long row_tests;    // (HitPoint, row of 64 deposits) sweeps of the gather: x 64 = pairs evaluated
    unsigned long long stamps[8];   // cycle sums per trace-loop section, filled by -DTRACE_STAMPS diagnostic builds only
    unsigned int overflow;   // bit0 hitpoint capacity, bit1 eye task queue, bit2 deposit capacity (sticky: see overflow_code)
    unsigned int pad;
};

struct GridParams {
    double inv_cs;       // 1 / cell size
    uint32_t mask;       // bucket count - 1 (power of two)
};

// ------------------------------------------------------------------ grid helpers
#define CELL_OFF 1048576   // 2^20
__device__ __forceinline__ uint32_t cell_coord(double p, double inv_cs) {
    double f = floor(p * inv_cs);
    f = fmin(fmax(f, -1048574.0), 1048574.0);   // keeps every 21-bit field in [2, 2^21 - 2]: +-1 never carries
    return (uint32_t)((int)f + CELL_OFF);
}
__device__ __forceinline__ uint64_t cell_key(d3 p, double inv_cs) {
    return ((uint64_t)cell_coord(p.x, inv_cs) << 42) | ((uint64_t)cell_coord(p.y, inv_cs) << 21) | (uint64_t)cell_coord(p.z, inv_cs);
}
// Bucket of a cell: cells are grouped in 4 x 4 x 4 blocks; the block is hashed, the position inside the
// block is the low 6 bits.  The 64 cells of a block are therefore 64 consecutive buckets, so after the
// counting sort spatial neighbours are mostly memory neighbours (what the gather's L2 reuse rests on).
__device__ __forceinline__ uint32_t key_bucket(uint64_t key, uint32_t mask) {
    uint32_t ix = (uint32_t)(key >> 42) & 0x1fffffu, iy = (uint32_t)(key >> 21) & 0x1fffffu, iz = (uint32_t)key & 0x1fffffu;
    uint32_t h = ((ix >> 2) * 73856093u) ^ ((iy >> 2) * 19349663u) ^ ((iz >> 2) * 83492791u);
    uint32_t local = ((ix & 3u) << 4) | ((iy & 3u) << 2) | (iz & 3u);
    return ((fmix3(h) << 6) | local) & mask;
}

// ------------------------------------------------------------------ LDS scene staging
extern __shared__ double lds_scene[];
// Scene tables of up to SCENE_LDS_MAX bytes are copied to LDS by every workgroup (IN_LDS); larger ones (hundreds of
// Bezier patches) are read where they lie in global memory, through the same SceneView.
#define SCENE_LDS_MAX (64 * 1024)
template <bool IN_LDS>
__device__ __forceinline__ SceneView stage_scene(const SceneParams& sp) {
    const char* base = sp.blob;
    if (IN_LDS) {
        const uint32_t* src = (const uint32_t*)sp.blob;
        uint32_t* dst = (uint32_t*)lds_scene;
        for (int i = threadIdx.x; i < sp.blob_bytes / 4; i += blockDim.x) dst[i] = src[i];
        __syncthreads();
        base = (const char*)lds_scene;
    }
    SceneView S;
    S.objs = (const DevObject*)base;
    S.patch_box = (const double*)(base + sp.off_box);
    S.patch_cp = (const float*)(base + sp.off_cp);
    S.pw_q = sp.pw_q;
    S.pw_c = sp.pw_c;
    S.sub_box = sp.off_sub >= 0 ? (const float*)(base + sp.off_sub) : sp.sub_box;
    S.patch_obj = (const int*)(base + sp.off_pobj);
    S.aobj = (const AObj*)(base + sp.off_aobj);
    S.n_objs = sp.n_objs;
    S.n_patches = sp.n_patches;
    S.texels = sp.texels;
    S.n_bez = sp.n_bez;
    S.bez_idx = sp.bez_idx;
    S.n_runs = sp.n_runs;
    S.runs_lo = sp.runs_lo;
    S.runs_hi = sp.runs_hi;
    return S;
}

// wave-aggregated slot allocation: every lane calls it; lanes with want = true get
// consecutive slots from *counter.
__device__ __forceinline__ unsigned long long wave_alloc(bool want, unsigned long long* counter) {
    uint64_t m = __ballot(want);
    if (m == 0) return 0;
    int leader = __ffsll((unsigned long long)m) - 1;
    unsigned long long base = 0;
    if (lane_id() == leader) base = atomicAdd(counter, (unsigned long long)__popcll(m));
    base = bcast_u64(base, leader);
    return base + (unsigned long long)__popcll(m & lanemask_lt());
}

__device__ __forceinline__ void flush_counter(unsigned long long* dst, uint32_t v) {
    uint32_t s = wave_sum_u(v);
    if (lane_id() == 0 && s) atomicAdd(dst, (unsigned long long)s);
}
// the same for a tally that is wave-uniform already (kept in a scalar register, counted with ballots)
__device__ __forceinline__ void flush_uniform(unsigned long long* dst, uint32_t v) {
    if (lane_id() == 0 && v) atomicAdd(dst, (unsigned long long)v);
}
__device__ __forceinline__ uint32_t count_lanes(bool p) { return (uint32_t)__popcll(__ballot(p)); }

// ------------------------------------------------------------------ path tracing: eye (pass 1) and photon (pass 2)
// wave-aggregated allocation of n slots per lane (n may be 0); returns the lane's first slot
__device__ __forceinline__ unsigned long long wave_alloc_n(uint32_t n, unsigned long long* counter) {
    uint32_t inc = n;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        uint32_t v = (uint32_t)__shfl_up((int)inc, off, 64);
        if (lane_id() >= off) inc += v;
    }
    const uint32_t total = bcast_u(inc, 63);
    if (total == 0) return 0;
    unsigned long long base = 0;
    if (lane_id() == 0) base = atomicAdd(counter, (unsigned long long)total);
    base = bcast_u64(base, 0);
    return base + (inc - n);
}

// Output slots (deposits, raw HitPoints) are claimed from the global counter a chunk at a time and
// handed out inside the wave: [next, end) is the wave's open chunk (wave-uniform).  Every lane calls
// this; lanes with want = true get distinct slots.  A claim that does not fit the open chunk uses up
// its rest and continues in a fresh chunk, so only the last chunk of a wave slot is left partly unused.
__device__ __forceinline__ unsigned long long chunk_alloc(bool want, unsigned long long& next, unsigned long long& end,
                                                          unsigned long long* counter, unsigned long long chunk) {
    const uint64_t m = __ballot(want);
    if (m == 0) return 0;
    const unsigned long long need = (unsigned long long)__popcll(m), left = end - next;
    const unsigned long long rank = (unsigned long long)__popcll(m & lanemask_lt());
    unsigned long long slot = next + rank;
    if (need > left) {
        unsigned long long base = 0;
        if (lane_id() == 0) base = atomicAdd(counter, chunk);
        base = bcast_u64(base, 0);
        if (rank >= left) slot = base + (rank - left);
        next = base + (need - left);
        end = base + chunk;
    } else {
        next += need;
    }
    return slot;
}

#ifndef TRACE_EYE_OCC
#define TRACE_EYE_OCC 2
#endif
#ifndef TRACE_COOP_OCC
#define TRACE_COOP_OCC 2
#endif
#ifdef TRACE_STAMPS
#define STAMP(i) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_s_waitcnt(0xC07F); \
                      st_sum[i] += t_ - st_last; st_last = t_; } while (0)
#else
#define STAMP(i) do { } while (0)
#endif
enum { MODE_EYE = 0, MODE_PHOTON = 1 };
#ifndef TRACE_WORK_CHUNK_EYE
#define TRACE_WORK_CHUNK_EYE 128u
#endif
// work items a wave claims per atomic on the work counter
#define TRACE_WORK_CHUNK (MODE == MODE_EYE ? TRACE_WORK_CHUNK_EYE : 128u)
// output slots (deposits / raw HitPoints) a wave claims per atomic on the slot counter
#ifndef TRACE_SLOT_CHUNK
#define TRACE_SLOT_CHUNK 256u
#endif
#define TRACE_LDS_LIGHTS 16        // lights kept in LDS by the photon kernel (40 bytes each); more are read from global memory
// what the photon kernel keeps in LDS beside the scene tables: 128 emission directions per wave, the first lights, the sine table
#define TRACE_PHOTON_LDS(waves) ((waves) * 128 * 3 * sizeof(double) + TRACE_LDS_LIGHTS * sizeof(DevLight) + 2 * SPPM_SINCOS_TAB * sizeof(double))
#define TRACE_PHOTON_LDS_EXTRA TRACE_PHOTON_LDS(4)
#define DEP_HOLE 0x00ffffffu      // DepB::bucket of a claimed but unused slot (buckets are < 2^22)

struct TraceParams {
    // work items: [0, n_resume) = park records of the previous round, then fresh_count fresh
    // items (pixels for the eye pass, photon indices for the photon pass)
    const ParkRec* resume; const unsigned long long* n_resume_dev;   // park records of the previous round and their count (device)
    const NewtonTask* tasks_in; const float4* res_in;
    unsigned long long fresh_first, fresh_count;
    unsigned long long* next;
    ParkRec* park_out; unsigned long long* n_park; unsigned long long park_cap;
    NewtonTask* tasks_out; unsigned long long* n_tasks; unsigned long long task_cap;
    uint32_t rng;                             // rng_base(seed, replica * 4 + domain)
    DevCounters* counters;
    // eye pass
    double pos[3], dir[3], du[3], dv[3];
    int w, h;
    // Work item k of the eye pass is pixel (k * pix_mul) mod (w * h), pix_mul coprime to w * h: a wave's 64
    // pixels are spread over the image, so the rays that meet a Bezier object are a few lanes of every wave
    // (as photons are) instead of all lanes of a few waves, and can be solved in place without idling the rest.
    unsigned long long pix_mul;
    // floor(2^64 / (w * h)) and floor(2^32 / w): the pixel of a work item and its row and column come from multiplications
    // by these and a correction step instead of 64-bit divisions (a few hundred instructions per refill otherwise)
    unsigned long long npx_magic; uint32_t w_magic;
    RawHit* hits; unsigned long long* n_hits; unsigned long long hit_cap;   // n_hits: claimed slots
    unsigned long long* n_real_hits;                                          // HitPoints written
    uint32_t* hcount;                                                         // HitPoints per grid bucket (the histogram of their counting sort)
    // photon pass
    const DevLight* lights; int n_lights; uint32_t per_round;
    const double* sincos_tab;                 // SPPM_SINCOS_TAB pairs (sin, cos) of 2 PI j / SPPM_SINCOS_TAB, copied to LDS by the photon kernel
    Deposit* dep; unsigned long long* n_dep; unsigned long long dep_cap;
    uint32_t* dcount; GridParams grid;
    const uint8_t* occ;                       // one byte per bucket: some HitPoint lies within one cell of a cell of that bucket
                                              // (null: every deposit is kept)
    unsigned long long* wave_chunks;          // per wave slot {next, end} of its open deposit-slot chunk
    // Direct placement (null: off).  place[b] .. place[b + 1] is the room the sorted deposit array holds for bucket b,
    // laid out from the histogram of an earlier batch of the same size; a deposit goes straight to
    // sorted[place[b] + (its rank among the deposits of b)], the rank being the value the histogram atomic returns,
    // and the counting sort is not needed.  A deposit that finds its bucket full goes to the list `over`.
    uint32_t* dcount_all;                     // histogram of ALL diffuse hits, kept or not (null: not wanted); a layout is made from it
    const uint32_t* place;
    Deposit* sorted;
    Deposit* over; unsigned long long* n_over; unsigned long long over_cap;
};

// One kernel for both passes, on persistent wavefronts: a lane that finishes (or parks) its
// path is refilled from the work counter, so lanes stay busy across paths of different length.
//   MODE_EYE    RayTracer::GetHitPoint + ViewTrace (Raytracer.h:102-116, :296-357)
//   MODE_PHOTON PhotonMap's emission loop + PhotonTrace (Raytracer.h:117-209, :220-233); a diffuse
//               hit appends one deposit record (the reference's radius-search query)
// Bezier objects: with COOP the wave solves them in place (bezier_coop).  Without, a ray whose
// box culls leave patches to solve is PARKED: its state and one task per patch go to queues,
// k_newton solves all tasks of the round at full occupancy, and the next launch resumes the ray
// from its record.  This keeps the 50 x 10 Newton iterations out of the divergent path loop.
// BEZ = false: a scene without Bezier objects; the kernel then holds no Bezier stage (its registers and spilled scalars cost
// the loop time even where it never runs).
template <int MODE, bool COOP, bool IN_LDS, bool BEZ = true>
__device__ __forceinline__ void trace_body(const SceneParams& sp, const TraceParams& tp) {
    const unsigned long long n_resume = tp.resume ? min(*tp.n_resume_dev, tp.park_cap) : 0ull;   // records beyond the capacity were never written
    const unsigned long long total = n_resume + tp.fresh_count;
    // the host sizes a resume launch without knowing its count: workgroups beyond the work leave at once
    if (blockIdx.x > 0 && (unsigned long long)blockIdx.x * blockDim.x >= total) return;
    SceneView S = stage_scene<IN_LDS>(sp);
#ifdef TRACE_SETPRIO
    __builtin_amdgcn_s_setprio(TRACE_SETPRIO);   // experiment: the trace wave's issue priority over the gather waves that share its SIMD
#endif
    const int lane = lane_id();
    bool active = false, pending = false, exhausted = (total == 0);
    d3 o = mk(0, 0, 0), d = mk(0, 0, 1);
    float carry[3] = {0, 0, 0};
    uint32_t id = 0, tfirst = 0, tcount = 0;
    int dep = 0, node = 0;
    Near nr;
    near_init(nr);
    uint32_t ncalls = 0, nseg = 0, ndep = 0, nphot = 0, nhit = 0, nstored = 0;   // wave-uniform tallies (scalar registers): lanes are counted with ballots
    // Work items and deposit slots are claimed from the global counters a chunk at a time (all waves
    // hammering one address with a returned atomic per loop iteration serialises the whole grid).
    unsigned long long work_next = 0, work_end = 0;     // this wave's claimed work items, wave-uniform
    unsigned long long chunk_base = 0;                  // first item of the chunk the emission directions in LDS belong to
    // per-wave LDS slice behind the scene tables: emission directions of the claimed photon chunk
    double* emit_dir = (double*)((char*)lds_scene + (IN_LDS ? ((sp.blob_bytes + 15) & ~15) : 0)) + (threadIdx.x >> 6) * (3 * 128);
    // Up to TRACE_LDS_LIGHTS lights are copied behind the emission slices: a refill then reads its light from LDS instead of
    // waiting for a vector load from global memory at the head of a turn of the loop.
    DevLight* lds_lights = (DevLight*)((double*)((char*)lds_scene + (IN_LDS ? ((sp.blob_bytes + 15) & ~15) : 0)) + (blockDim.x >> 6) * (3 * 128));
    const bool lights_in_lds = MODE == MODE_PHOTON && tp.n_lights <= TRACE_LDS_LIGHTS;
    // and behind the lights the table of sincos_2pi_u24 (the diffuse bounce's angle about the normal)
    double* lds_sincos = (double*)(lds_lights + TRACE_LDS_LIGHTS);
    if (MODE == MODE_PHOTON) {
        for (uint32_t i = threadIdx.x; i < 2u * SPPM_SINCOS_TAB; i += blockDim.x) lds_sincos[i] = tp.sincos_tab[i];
        if (lights_in_lds) {
            const uint32_t* src = (const uint32_t*)tp.lights;
            uint32_t* dst = (uint32_t*)lds_lights;
            for (uint32_t i = threadIdx.x; i < (uint32_t)tp.n_lights * (uint32_t)(sizeof(DevLight) / 4); i += blockDim.x) dst[i] = src[i];
        }
        __syncthreads();
    }
    unsigned long long slot_next = 0, slot_end = 0;     // this wave's open chunk of output slots (deposits / raw HitPoints), wave-uniform
    const uint32_t wave_slot = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    slot_next = tp.wave_chunks[2 * wave_slot];
    slot_end = tp.wave_chunks[2 * wave_slot + 1];
#ifdef TRACE_STAMPS
    unsigned long long st_sum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long st_last = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_s_waitcnt(0xC07F);
#endif
    for (;;) {
        STAMP(7);
#ifdef TRACE_STAMP_ITERS
        st_sum[3] -= 1000000000ull;   // diagnostic: the loop's turns, counted in the merge / park slot (a billion off per turn)
#endif
        // ---- refill idle lanes
        const uint64_t idle = __ballot(!active);
        if (idle && work_next >= work_end && !exhausted) {
            unsigned long long base = 0;
            if (lane == 0) base = atomicAdd(tp.next, (unsigned long long)TRACE_WORK_CHUNK);
            base = bcast_u64(base, 0);
            work_next = base < total ? base : total;
            work_end = base + TRACE_WORK_CHUNK < total ? base + TRACE_WORK_CHUNK : total;
            if (base + TRACE_WORK_CHUNK >= total) exhausted = true;
            if (MODE == MODE_PHOTON) {
                // Light::emit (Light.h:8-13) for the whole chunk now, all lanes busy, into this wave's LDS
                // slice: the rejection loop of Vec3::GetUnitRandVec (Vec3.h:57-65) then runs once per 128
                // photons instead of in every bounce iteration that happens to refill a lane.
                chunk_base = work_next;
#pragma unroll 1
                for (uint32_t e = (uint32_t)lane; e < TRACE_WORK_CHUNK; e += 64) {
                    const unsigned long long item = chunk_base + e;
                    if (item >= n_resume && item < work_end) {
                        const uint32_t es = rng_sub(tp.rng, (uint32_t)(tp.fresh_first + (item - n_resume)), 0u, P_EMIT);
                        double x = 0, y = 0, z = 0;
                        for (uint32_t k = 0; k < 64; k++) {
                            x = rng_u01(es, 3 * k + 0) * 2.0 + -1.0;
                            y = rng_u01(es, 3 * k + 1) * 2.0 + -1.0;
                            z = rng_u01(es, 3 * k + 2) * 2.0 + -1.0;
                            const double m = x * x + y * y + z * z;
                            if (!(m > 1 || m < SPPM_EPS)) break;
                        }
                        const d3 u = get_unit_any(mk(x, y, z));
                        emit_dir[3 * e + 0] = u.x; emit_dir[3 * e + 1] = u.y; emit_dir[3 * e + 2] = u.z;
                    }
                }
            }
        }
        if (idle && work_next < work_end) {
            const unsigned long long avail = work_end - work_next;
            const unsigned long long rank = (unsigned long long)__popcll(idle & lanemask_lt());
            const unsigned long long my = work_next + rank;
            const unsigned long long n = (unsigned long long)__popcll(idle);
            work_next += n < avail ? n : avail;
            bool took_photon = false;      // this lane starts a freshly emitted photon in this refill
            if (!active && rank < avail) {
                active = true;
                pending = false;
                if (my < n_resume) {
                    const ParkRec r = tp.resume[my];
                    o = ld3(r.o); d = ld3(r.d);
                    carry[0] = r.carry[0]; carry[1] = r.carry[1]; carry[2] = r.carry[2];
                    id = r.id; dep = r.dep; node = r.node;
                    if (r.task_count) {
                        pending = true;
                        near_init(nr);
                        if (r.aobj >= 0) { nr.dist = r.adist; nr.obj = r.aobj; nr.inside = r.ainside; }
                        tfirst = r.task_first; tcount = r.task_count;
                    }
                } else {
                    unsigned long long g = tp.fresh_first + (my - n_resume);
                    uint32_t prow = 0, pcol = 0;
                    if (MODE == MODE_EYE) {   // pixel (g * pix_mul) mod (w * h), see TraceParams::pix_mul; g, pix_mul < w * h <= 2^28
                        const unsigned long long npx = (unsigned long long)tp.w * (unsigned long long)tp.h, x = g * tp.pix_mul;
                        unsigned long long r = x - __umul64hi(x, tp.npx_magic) * npx;      // the quotient estimate is at most 2 short
                        if (r >= npx) r -= npx;
                        if (r >= npx) r -= npx;
                        g = r;
                        const uint32_t p = (uint32_t)r, w32 = (uint32_t)tp.w;
                        prow = __umulhi(p, tp.w_magic);
                        pcol = p - prow * w32;
                        if (pcol >= w32) { pcol -= w32; prow++; }
                        if (pcol >= w32) { pcol -= w32; prow++; }
                    }
                    id = (uint32_t)g; dep = 0; node = 0;
                    if (MODE == MODE_EYE) {
                        const int row = (int)prow, col = (int)pcol;
                        o = ld3(tp.pos);
                        // Camera::emit(y = row, x = col), Camera.h:18-22
                        d = get_unit_any(ld3(tp.du) * ((col + 0.5) / tp.w - 0.5) + ld3(tp.dv) * ((row + 0.5) / tp.h - 0.5) + ld3(tp.dir));
                        carry[0] = carry[1] = carry[2] = 1.0f;
                    } else {
                        // Raytracer.h:220-233: round by round, light by light; one light (the reference's scene) needs no division
                        const uint32_t li = tp.n_lights == 1 ? 0u : (uint32_t)(((uint32_t)g / tp.per_round) % (uint32_t)tp.n_lights);
                        const uint32_t e = (uint32_t)(my - chunk_base);                   // the direction drawn when the chunk was claimed
                        d = mk(emit_dir[3 * e + 0], emit_dir[3 * e + 1], emit_dir[3 * e + 2]);
                        if (lights_in_lds) {                                              // Light::emit, Light.h:8-13
                            SPPM_LDS(DevLight)* L = (SPPM_LDS(DevLight)*)lds_lights + li;   // typed as LDS: no flat load
                            o = mk(L->pos[0], L->pos[1], L->pos[2]); carry[0] = L->flux[0]; carry[1] = L->flux[1]; carry[2] = L->flux[2];
                        } else {
                            const DevLight& L = tp.lights[li];
                            o = ld3(L.pos); carry[0] = L.flux[0]; carry[1] = L.flux[1]; carry[2] = L.flux[2];
                        }
                        took_photon = true;
                    }
                }
            }
            if (MODE == MODE_PHOTON) nphot += count_lanes(took_photon);
        }
        STAMP(0);
        if (!__any(active)) {
            if (exhausted && work_next >= work_end) break;
            continue;   // the chunk ran dry mid-refill: claim the next one
        }

        // ---- Scene::NearCollide for lanes that carry an untraced ray
        const bool tracing = active && !pending;
        const uint32_t path_hash = rng_path(tp.rng, id);        // shared by the vertex's Newton, roulette and bounce draws
        const uint32_t nvertex = (uint32_t)(MODE == MODE_EYE ? node : dep);
        bool park = false;
        uint64_t words[SPPM_MASK_WORDS] = {0, 0, 0, 0};
        if (__any(tracing)) {
            Near a;
            near_init(a);
            analytic_nearest(S, tracing, o, d, a);
            STAMP(1);
            if (COOP) {
                if (BEZ) bezier_nearest_coop(S, tracing, o, d, path_hash, nvertex, a, ncalls);
                STAMP(2);
            } else {
                bezier_patch_masks(S, tracing, o, d, words);
                park = tracing && ((words[0] | words[1] | words[2] | words[3]) != 0);
            }
            if (tracing) nr = a;
        }
        // ---- lanes resumed from a park record: merge the solved patches, object by object, with
        // the per-patch filter of Bezier.h:251-259 (tasks are in patch order)
        if (pending) {
            float pt = (float)SPPM_MAX_NUM, pu = 0, pv = 0;
            int pidx = -1, cur = -1;
            const float e = (float)SPPM_EPS;
            for (uint32_t k = 0; k < tcount; k++) {
                const NewtonTask tk = tp.tasks_in[tfirst + k];
                const float4 r = tp.res_in[tfirst + k];
                const int ob = S.patch_obj[tk.patch];
                if (ob != cur) {
                    if (pidx >= 0) near_consider(nr, (double)pt, cur, 0, pidx, pt, pu, pv);
                    cur = ob; pt = (float)SPPM_MAX_NUM; pidx = -1;
                }
                if (r.x < e || r.x > pt || r.y < -e || r.y > 1 + e || r.z < -e || r.z > 1 + e) continue;
                if (r.x < pt) { pt = r.x; pu = r.y; pv = r.z; pidx = (int)tk.patch; }
            }
            if (pidx >= 0) near_consider(nr, (double)pt, cur, 0, pidx, pt, pu, pv);
            pending = false;
        }
        // ---- park rays that still need their Bezier solves
        if (!COOP) {
            const uint32_t my_tasks = park ? (uint32_t)(__popcll(words[0]) + __popcll(words[1]) + __popcll(words[2]) + __popcll(words[3])) : 0u;
            const unsigned long long slot = wave_alloc(park, tp.n_park);
            const unsigned long long tb = wave_alloc_n(my_tasks, tp.n_tasks);
            if (park) {
                const bool tasks_fit = tb + my_tasks <= tp.task_cap;
                if (!tasks_fit || slot >= tp.park_cap) atomicOr(&tp.counters->overflow, 2u);
                if (slot < tp.park_cap) {   // a claimed record is always well formed (later launches read it whatever the flags say)
                    ParkRec r;
                    r.o[0] = o.x; r.o[1] = o.y; r.o[2] = o.z; r.d[0] = d.x; r.d[1] = d.y; r.d[2] = d.z;
                    r.adist = nr.dist; r.aobj = nr.obj; r.ainside = nr.inside;
                    r.carry[0] = carry[0]; r.carry[1] = carry[1]; r.carry[2] = carry[2];
                    r.id = id; r.dep = dep; r.node = node;
                    r.task_first = (uint32_t)tb; r.task_count = tasks_fit ? my_tasks : 0u; r.sub = rng_vertex(path_hash, nvertex, P_NEWTON); r.pad = 0;
                    tp.park_out[slot] = r;
                    uint32_t k = 0;
#pragma unroll
                    for (int w = 0; w < SPPM_MASK_WORDS; w++) {
                        uint64_t m = tasks_fit ? words[w] : 0ull;
                        while (m) {
                            const int bit = __ffsll((unsigned long long)m) - 1;
                            m &= m - 1;
                            NewtonTask t;
                            t.park = (uint32_t)slot; t.patch = (uint32_t)(w * 64 + bit);
                            tp.tasks_out[tb + k++] = t;
                        }
                    }
                } else {
                    atomicOr(&tp.counters->overflow, 2u);
                }
                active = false;
            }
        }

        // ---- the hit (or miss) of every lane still active
        Near fin = nr;
        if (!active) near_init(fin);
        STAMP(3);
        const Hit hit = finish_hit(S, o, d, fin);
        STAMP(4);
        nseg += count_lanes(active);
        if (active && hit.obj < 0) active = false;

        if (MODE == MODE_EYE) {
            bool emit_hp = false, push = false;
            RawHit rh;
            ParkRec br;
            if (active) {
                dep++;
                const DevObject& ob = S.objs[hit.obj];
                const bool has_refl = ob.flags & F_REFL, has_refr = ob.flags & F_REFR;
                d3 refl_d = mk(0, 0, 0), refr_d = mk(0, 0, 0);
                if (has_refl) refl_d = get_unit_near1(get_refl(d, hit.n));                // Raytracer.h:320-326
                if (has_refr) {                                                           // Raytracer.h:327-336
                    const double refrn = hit.inside ? ob.refrn : ob.inv_refrn;
                    refr_d = get_unit_near1(refracted(d, hit.inside ? neg(hit.n) : hit.n, refrn));
                }
                float hc[3];
                hit_color(hit, hc);                                                       // the texels were asked for in finish_hit
                const float bw[3] = {hc[0] * carry[0], hc[1] * carry[1], hc[2] * carry[2]};
                if (ob.flags & F_DIFF) {                                                  // Raytracer.h:312-319
                    emit_hp = true;
                    rh.pos[0] = hit.pos.x; rh.pos[1] = hit.pos.y; rh.pos[2] = hit.pos.z;
                    rh.n[0] = (float)hit.n.x; rh.n[1] = (float)hit.n.y; rh.n[2] = (float)hit.n.z;
                    rh.v[0] = (float)d.x; rh.v[1] = (float)d.y; rh.v[2] = (float)d.z;
                    rh.wgt[0] = bw[0] * ob.diff[0]; rh.wgt[1] = bw[1] * ob.diff[1]; rh.wgt[2] = bw[2] * ob.diff[2];
                    rh.pix = (int)id; rh.node = node; rh.obj = hit.obj;
                }
                if (dep > SPPM_MAX_DEP) {
                    active = false;  // the next NearCollide result would be discarded (Raytracer.h:303-309)
                } else if (has_refl) {
                    if (has_refr) {  // both lobes: the refraction branch becomes a fresh parked ray
                        push = true;
                        br.o[0] = hit.pos.x; br.o[1] = hit.pos.y; br.o[2] = hit.pos.z;
                        br.d[0] = refr_d.x; br.d[1] = refr_d.y; br.d[2] = refr_d.z;
                        br.carry[0] = bw[0] * ob.refr[0]; br.carry[1] = bw[1] * ob.refr[1]; br.carry[2] = bw[2] * ob.refr[2];
                        br.id = id; br.dep = dep; br.node = 2 * node + 2;
                        br.adist = 0; br.aobj = -1; br.ainside = 0; br.task_first = 0; br.task_count = 0; br.sub = 0; br.pad = 0;
                    }
                    carry[0] = bw[0] * ob.refl[0]; carry[1] = bw[1] * ob.refl[1]; carry[2] = bw[2] * ob.refl[2];
                    o = hit.pos; d = refl_d; node = 2 * node + 1;
                } else if (has_refr) {
                    carry[0] = bw[0] * ob.refr[0]; carry[1] = bw[1] * ob.refr[1]; carry[2] = bw[2] * ob.refr[2];
                    o = hit.pos; d = refr_d; node = 2 * node + 2;
                } else {
                    active = false;
                }
            }
            const unsigned long long hs = chunk_alloc(emit_hp, slot_next, slot_end, tp.n_hits, TRACE_SLOT_CHUNK);
            nhit += count_lanes(emit_hp);
            if (emit_hp) {
                if (hs < tp.hit_cap) {
                    tp.hits[hs] = rh;
                    atomicAdd(&tp.hcount[key_bucket(cell_key(hit.pos, tp.grid.inv_cs), tp.grid.mask)], 1u);
                } else atomicOr(&tp.counters->overflow, 1u);
            }
            const unsigned long long bs = wave_alloc(push, tp.n_park);
            if (push) {
                if (bs < tp.park_cap) tp.park_out[bs] = br;
                else atomicOr(&tp.counters->overflow, 2u);
            }
        } else {
            bool dp = false, keep = false;
            uint32_t pbase = 0, pend = 0, prank = 0, pocc = 0;
            PosKey da;
            DepB db;
            if (active) {
                const uint32_t vertex = (uint32_t)dep;
                dep++;
                const DevObject& ob = S.objs[hit.obj];
                if (ob.flags & F_DIFF) {                                                  // Raytracer.h:136-160: the gather query
                    dp = true;
                    da.x = hit.pos.x; da.y = hit.pos.y; da.z = hit.pos.z;
                    da.key = cell_key(hit.pos, tp.grid.inv_cs);
                    db.nx = (float)hit.n.x; db.ny = (float)hit.n.y; db.nz = (float)hit.n.z;
                    db.fr = carry[0]; db.fg = carry[1]; db.fb = carry[2];
                    const uint32_t bk = key_bucket(da.key, tp.grid.mask);
                    db.bucket = bk | (vertex << 24); db.photon = id;
                    if (tp.dcount_all) atomicAdd(&tp.dcount_all[bk], 1u);
                    if (tp.place) {
                        // A placed deposit asks for everything its store will need right here, in one go and before the bounce
                        // is worked out: the occupancy byte, the bucket's room and — the returned atomic — its rank in the
                        // bucket.  The answers are looked at where the record is stored (the asm statement below keeps the
                        // compiler from waiting for them any earlier).  The rank is asked for before the occupancy byte is
                        // known, so the fill count of a bucket out of every HitPoint's reach also counts the deposits that
                        // are then dropped: nobody reads that count (the gather only visits buckets within reach).
                        pbase = tp.place[bk]; pend = tp.place[bk + 1];
                        pocc = tp.occ ? (uint32_t)tp.occ[bk] : 1u;
                        prank = atomicAdd(&tp.dcount[bk], 1u);
                    } else {
                        // no HitPoint in the 27 cells around: the gather would never look at this deposit
                        keep = !tp.occ || tp.occ[bk] != 0;
                    }
                }
                // Obj::Roulette, Obj.h:30-45
                // A material with one lobe (every one of the reference's scene) has one outcome whatever r is: the draw, which
                // no other draw depends on (counter-based streams), is then not even computed.
                const int lobe = ob.lobe;
                double r = 0;
                if (__any(lobe == 0)) r = rng_u01(rng_vertex(path_hash, vertex, P_ROULETTE), 0) * ob.allr;
                d3 nd;
                if (lobe == F_DIFF || (lobe == 0 && ob.diffr > r)) {                      // DIFF, Raytracer.h:163-166
                    const uint32_t ds = rng_vertex(path_hash, vertex, P_DIFFUSE);
#ifdef ABLATE_BOUNCE
                    nd = get_unit(hit.n + mk(rng_u01(ds, 0) - 0.5, rng_u01(ds, 1) - 0.5, 0.1));
#else
                    if (MODE == MODE_PHOTON) nd = rand_refl_tab(hit.n, rng_u01(ds, 0), rng_bits(ds, 1), (SPPM_LDS(double)*)lds_sincos);
                    else nd = rand_refl(hit.n, rng_u01(ds, 0), rng_u01(ds, 1));
#endif
                } else if (lobe == F_REFL || (lobe == 0 && ob.diffr + ob.reflr > r)) {    // REFL, Raytracer.h:177-181
                    nd = get_unit_near1(get_refl(d, hit.n));
                } else {                                                                  // REFR, Raytracer.h:182-190
                    const double refrn = hit.inside ? ob.refrn : ob.inv_refrn;
                    nd = get_unit_near1(refracted(d, hit.inside ? neg(hit.n) : hit.n, refrn));
                }
                float hc[3];
                hit_color(hit, hc);                                                       // the texels were asked for in finish_hit
                carry[0] *= hc[0]; carry[1] *= hc[1]; carry[2] *= hc[2];
                o = hit.pos; d = nd;
                if (dep > SPPM_MAX_DEP) active = false;                                   // Raytracer.h:126-133
            }
            STAMP(6);
            ndep += count_lanes(dp);
            if (tp.place) {
                asm volatile("" : "+v"(pocc), "+v"(pbase), "+v"(pend), "+v"(prank));     // the answers: not before this point
                keep = pocc != 0;
            }
            dp = dp && keep;
            nstored += count_lanes(dp);
            if (tp.place) {
                if (dp) {
                    const uint32_t base = pbase, room = pend - pbase, rank = prank;
                    Deposit rec; rec.a = da; rec.b = db;
                    if (rank < room) {
                        tp.sorted[base + rank] = rec;
                    } else {
                        const unsigned long long s = atomicAdd(tp.n_over, 1ull);
                        if (s < tp.over_cap) tp.over[s] = rec;
                        else atomicOr(&tp.counters->overflow, 4u);
                    }
                }
            } else {
                const unsigned long long slot = chunk_alloc(dp, slot_next, slot_end, tp.n_dep, TRACE_SLOT_CHUNK);
                if (dp) {
                    if (slot < tp.dep_cap) {
                        Deposit rec; rec.a = da; rec.b = db;
                        tp.dep[slot] = rec;
                        atomicAdd(&tp.dcount[db.bucket & 0xffffffu], 1u);
                    } else atomicOr(&tp.counters->overflow, 4u);
                }
            }
        }
        STAMP(5);
    }
#ifdef TRACE_STAMPS
    if (lane == 0) for (int i = 0; i < 8; i++) atomicAdd(&tp.counters->stamps[i], st_sum[i]);
#endif
    if (lane == 0) {   // the open chunk stays with this wave slot for the next launch of the pass / batch
        tp.wave_chunks[2 * wave_slot] = slot_next;
        tp.wave_chunks[2 * wave_slot + 1] = slot_end;
    }
    flush_uniform(&tp.counters->newton_calls, ncalls);
    if (MODE == MODE_EYE) {
        flush_uniform(&tp.counters->eye_segments, nseg);
        flush_uniform(tp.n_real_hits, nhit);
    } else {
        flush_uniform(&tp.counters->segments, nseg);
        flush_uniform(&tp.counters->deposits, ndep);
        flush_uniform(&tp.counters->photons, nphot);
        flush_uniform(&tp.counters->stored, nstored);
    }
}

template <int MODE, bool COOP, bool IN_LDS, bool BEZ = true>
__global__ void __launch_bounds__(256, COOP ? (MODE == MODE_EYE ? TRACE_EYE_OCC : TRACE_COOP_OCC) : 3) k_trace(SceneParams sp, TraceParams tp) {
    trace_body<MODE, COOP, IN_LDS, BEZ>(sp, tp);
}
// The two kernels the reference's scene runs, under names of their own so that each can be held to a register count
// (the attribute counts register PAIRS of the unified file here: 104 means 208 VGPRs).  Left alone the photon kernel takes
// 244, i.e. 248 allocated: beside three gather workgroups of 80 that leaves 24 registers per lane of a SIMD, and the small
// kernels between the passes (scans, HitPoint scatter, occupancy, resolve: 32 to 40 registers) wait for a workgroup to
// leave.  At 232 nothing spills and they fit in; at 208 twenty values spill, the kernel takes the same time and leaves 64.
#ifndef TRACE_PHOTON_VGPRS
#define TRACE_PHOTON_VGPRS 104
#endif
#ifndef TRACE_EYE_VGPRS
#define TRACE_EYE_VGPRS 0
#endif
__global__ void __launch_bounds__(256, TRACE_COOP_OCC) __attribute__((amdgpu_num_vgpr(TRACE_PHOTON_VGPRS))) k_trace_photon_bez(SceneParams sp, TraceParams tp) {
    trace_body<MODE_PHOTON, true, true, true>(sp, tp);
}
__global__ void __launch_bounds__(256, TRACE_EYE_OCC) __attribute__((amdgpu_num_vgpr(TRACE_EYE_VGPRS))) k_trace_eye_bez(SceneParams sp, TraceParams tp) {
    trace_body<MODE_EYE, true, true, true>(sp, tp);
}
// The photon kernel of a scene without Bezier objects takes 198 registers; held to 160 it spills 31 values and runs 3 %
// longer by itself, but one such wave leaves a SIMD room for four gather waves (or two of them for two): the glass-ball
// caustic scene, 48 photon batches and a long gather per iteration, gains 8 % as a whole.
#ifndef TRACE_PLAIN_VGPRS
#define TRACE_PLAIN_VGPRS 80
#endif
__global__ void __launch_bounds__(256, TRACE_COOP_OCC) __attribute__((amdgpu_num_vgpr(TRACE_PLAIN_VGPRS))) k_trace_photon_plain(SceneParams sp, TraceParams tp) {
    trace_body<MODE_PHOTON, true, true, false>(sp, tp);
}

// Bezier3::newton (Bezier.h:112-159) for the parked rays of a round: one wave per ray at a time, on
// long-lived waves.  The lanes look up the boxes of the ray's tasks (one task = one patch whose box the ray
// enters), then the wave solves them nearest box first, lane r < 50 = restart r (newton_coop), and stops as
// soon as no remaining box can hold a hit nearer than the best one found or than the plane / sphere hit the
// ray already has (aabb_entry_lb).  Every task gets a result; (1e20, 0, 0) = nothing, or not needed.
__global__ void __launch_bounds__(256) k_newton(const ParkRec* __restrict__ park, const unsigned long long* __restrict__ n_park_dev,
                                                unsigned long long park_cap, const NewtonTask* __restrict__ tasks, unsigned long long task_cap,
                                                const double* __restrict__ patch_box, const float* __restrict__ pw_q,
                                                const double* __restrict__ pw_c, float4* __restrict__ res, DevCounters* counters) {
    const uint32_t n_park = (uint32_t)min(*n_park_dev, park_cap);
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = (gridDim.x * blockDim.x) >> 6;
    const int lane = lane_id();
    uint32_t done = 0;
    for (uint32_t r = wave; r < n_park; r += n_waves) {
        const ParkRec& rec = park[r];
        const uint32_t first = __builtin_amdgcn_readfirstlane(rec.task_first), cnt = __builtin_amdgcn_readfirstlane(rec.task_count);
        if (cnt == 0 || (unsigned long long)first + cnt > task_cap) continue;
        const d3 so = ld3(rec.o), sd = ld3(rec.d);
        const uint32_t sub = rec.sub;
        float best = rec.aobj >= 0 ? (float)rec.adist : (float)SPPM_MAX_NUM;   // nearest hit so far; solved patches lower it
        const float e = (float)SPPM_EPS;
        for (uint32_t base = 0; base < cnt; base += 64) {
            const uint32_t i = base + lane;
            const bool valid = i < cnt;
            const uint32_t patch = valid ? tasks[first + i].patch : 0u;
            const uint32_t lbb = valid ? __float_as_uint(aabb_entry_lb(patch_box + 6 * (size_t)patch, so, sd)) : 0xffffffffu;
            float rt = (float)SPPM_MAX_NUM, ru = 0, rv = 0;
            uint64_t pm = __ballot(valid);
            while (pm) {
                int sel = 0;
                uint32_t selb = 0xffffffffu;
                for (uint64_t q = pm; q; q &= q - 1) {
                    const int bit = __ffsll((unsigned long long)q) - 1;
                    const uint32_t v = bcast_u(lbb, bit);
                    if (v < selb) { selb = v; sel = bit; }
                }
                if (__uint_as_float(selb) - SPPM_PRUNE_SLACK > best) break;
                pm &= ~(1ull << sel);
                float t, u, v;
                newton_coop(pw_q, pw_c, (int)bcast_u(patch, sel), so, sd, sub, t, u, v);
                done += 1u;
                if (lane == sel) { rt = t; ru = u; rv = v; }
                // what the per-patch filter of Bezier.h:251-259 would accept can lower the bound
                if (!(t < e || u < -e || u > 1 + e || v < -e || v > 1 + e)) best = fminf(best, t);
            }
            if (valid) res[first + i] = make_float4(rt, ru, rv, 0.0f);
        }
    }
    flush_uniform(&counters->newton_calls, done);
}

// ------------------------------------------------------------------ hashed grid over HitPoints
// (replaces the flann kd-tree build, Raytracer.h:370-380)
// The histogram of the HitPoints over the buckets is taken by the eye kernel as it writes them (TraceParams::hcount).

// Exclusive prefix sum start[i] = sum_{j<i} cnt[j], i in [0, n], in two launches:
// per-tile sums, then per-tile scan + the sum of the tiles before.
#define SCAN_THREADS 256
#define SCAN_PER_THREAD 8
#define SCAN_TILE (SCAN_THREADS * SCAN_PER_THREAD)
__device__ __forceinline__ uint32_t block_exclusive_scan_256(uint32_t v, uint32_t* total) {
    __shared__ uint32_t wsum[4];
    const int lane = lane_id(), w = threadIdx.x >> 6;
    uint32_t inc = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        uint32_t o = (uint32_t)__shfl_up((int)inc, off, 64);
        if (lane >= off) inc += o;
    }
    if (lane == 63) wsum[w] = inc;
    __syncthreads();
    uint32_t base = 0, tot = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) { uint32_t s = wsum[k]; if (k < w) base += s; tot += s; }
    __syncthreads();
    if (total) *total = tot;
    return base + inc - v;
}
__device__ __forceinline__ void scan_reduce_tile(const uint32_t* __restrict__ cnt, uint32_t n, uint32_t* __restrict__ partial, uint32_t tile) {
    const uint32_t base = tile * SCAN_TILE + threadIdx.x * SCAN_PER_THREAD;
    uint32_t s = 0;
#pragma unroll
    for (int k = 0; k < SCAN_PER_THREAD; k++) if (base + k < n) s += cnt[base + k];
    uint32_t tot;
    block_exclusive_scan_256(s, &tot);
    if (threadIdx.x == 0) partial[tile] = tot;
}
__global__ void __launch_bounds__(SCAN_THREADS) k_scan_reduce(const uint32_t* __restrict__ cnt, uint32_t n, uint32_t* __restrict__ partial) {
    scan_reduce_tile(cnt, n, partial, blockIdx.x);
}
// partial[]: the tile sums of k_scan_reduce (nb <= SCAN_TILE of them).  Every workgroup adds up the sums of the tiles
// before its own (at most 8 loads per thread), so no third launch is needed to scan them.
__global__ void __launch_bounds__(SCAN_THREADS) k_scan_final(const uint32_t* __restrict__ cnt, uint32_t n, const uint32_t* __restrict__ partial,
                                                             uint32_t nb, uint32_t* __restrict__ start) {
    uint32_t before = 0;
    for (uint32_t j = threadIdx.x; j < blockIdx.x; j += SCAN_THREADS) before += partial[j];
    uint32_t offset;
    block_exclusive_scan_256(before, &offset);           // offset = sum over the workgroup
    const uint32_t base = blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_PER_THREAD;
    uint32_t v[SCAN_PER_THREAD], s = 0;
#pragma unroll
    for (int k = 0; k < SCAN_PER_THREAD; k++) { v[k] = (base + k < n) ? cnt[base + k] : 0; s += v[k]; }
    uint32_t tile_total;
    uint32_t run = block_exclusive_scan_256(s, &tile_total) + offset;
#pragma unroll
    for (int k = 0; k < SCAN_PER_THREAD; k++) { if (base + k < n) start[base + k] = run; run += v[k]; }
    if (blockIdx.x == nb - 1 && threadIdx.x == 0) start[n] = offset + tile_total;
}

// What the gather reads of a HitPoint, one 64-byte record: position and cell, normal, weight and pixel.
struct HpRec { PosKey a; float4 nr; float4 wgt; };   // nr.w unused; wgt.w: the pixel index (bits of an int)
static_assert(sizeof(HpRec) == 64, "one memory line per HitPoint");
// A HitPoint lands at a place of the sorted arrays that depends on its bucket: every array written here costs a
// partly filled memory line per HitPoint.  Only the record the gather reads goes this way, as one line; the fields every HitPoint starts with alike (R2, tao, newN) are filled in order by k_hp_occupancy, and the
// ones only a host that asks for the HitPoints (or the radius update) reads are written by k_hp_meta when asked for, through
// the place noted here in the order of the raw array.
// n_hp: the HitPoint count the sorted arrays were sized for (the eye kernel's tally).  A record whose place lies beyond it
// would mean that tally and the histogram disagree: it is not written, and the pass reports a device error.
__global__ void k_hp_scatter(const RawHit* hits, uint32_t n, GridParams g, const uint32_t* start,
                             uint32_t* cursor, HpRec* hp, uint32_t* sorted_of_raw, uint32_t n_hp, unsigned int* overflow) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const RawHit h = hits[i];
    if (h.pix < 0) { sorted_of_raw[i] = 0xffffffffu; return; }   // a claimed but unused slot
    PosKey pk; pk.x = h.pos[0]; pk.y = h.pos[1]; pk.z = h.pos[2]; pk.key = cell_key(ld3(h.pos), g.inv_cs);
    const uint32_t b = key_bucket(pk.key, g.mask);
    uint32_t s = start[b] + atomicAdd(&cursor[b], 1u);
    if (s >= n_hp) { atomicOr(overflow, 16u); sorted_of_raw[i] = 0xffffffffu; return; }
    HpRec r;
    r.a = pk;
    r.nr = make_float4(h.n[0], h.n[1], h.n[2], 0.f);
    r.wgt = make_float4(h.wgt[0], h.wgt[1], h.wgt[2], __int_as_float(h.pix));
    hp[s] = r;
    sorted_of_raw[i] = s;
}
// View direction, tree node, object and N of every HitPoint, in the sorted order (see k_hp_scatter).
__global__ void k_hp_meta(const RawHit* hits, uint32_t n, const uint32_t* __restrict__ sorted_of_raw, HpMeta* hp_meta, int initial_n) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t s = sorted_of_raw[i];
    if (s == 0xffffffffu) return;
    const RawHit h = hits[i];
    HpMeta m; m.v[0] = h.v[0]; m.v[1] = h.v[1]; m.v[2] = h.v[2]; m.node = h.node; m.obj = h.obj; m.N = initial_n;   // HitPoint(): N = 0, Raytracer.h:66
    hp_meta[s] = m;
}

// Which buckets can hold a deposit that some HitPoint gathers: the cells of the HitPoints, widened by one cell
// in every direction (the gather of a cell looks at exactly those 27).  One thread per HitPoint of the sorted
// array; only the first of a run of equal cells does the marking.  A byte per bucket, set by plain stores:
// neighbouring cells mark the same few buckets over and over, which read-modify-write atomics would serialise.
// The same threads give every HitPoint its starting radius and sums (HitPoint(): R2 = INIT_R2, tao = 0, newN = 0,
// Raytracer.h:65), in order.
__global__ void k_hp_occupancy(const HpRec* __restrict__ hp, uint32_t n, GridParams g, uint8_t* occ, double* __restrict__ hp_r2,
                               float4* __restrict__ hp_tao) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    hp_r2[i] = SPPM_INIT_R2;
    hp_tao[i] = make_float4(0, 0, 0, __int_as_float(0));
    const uint64_t key = hp[i].a.key;
    if (i > 0 && hp[i - 1].a.key == key) return;
#pragma unroll
    for (int c = 0; c < 27; c++) {
        const int dx = c % 3 - 1, dy = (c / 3) % 3 - 1, dz = c / 9 - 1;
        const uint64_t nb_key = key + ((uint64_t)(int64_t)dx << 42) + ((uint64_t)(int64_t)dy << 21) + (uint64_t)(int64_t)dz;
        occ[key_bucket(nb_key, g.mask)] = 1;
    }
}

// HitPoint::update (Raytracer.h:69-79) for every HitPoint, the sweep of Raytracer.h:237-245.  With N = 0, the
// reference's initial value, the guard returns at once; the sweep is only launched when a host asked for another
// initial N (sppm_set_radius_update).
__global__ void k_hp_update(double* hp_r2, float4* hp_tao, HpMeta* hp_meta, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float4 t = hp_tao[i];
    const int N = hp_meta[i].N, newN = __float_as_int(t.w);
    if (N <= 0 || newN <= 0) return;
    const double k = (N + SPPM_ALPHA * newN) / (N + newN);
    hp_r2[i] *= k;
    t.x = (float)(t.x * k); t.y = (float)(t.y * k); t.z = (float)(t.z * k);
    t.w = __int_as_float(0);
    hp_tao[i] = t;
    hp_meta[i].N = (int)(N + SPPM_ALPHA * newN);          // `N += alpha * newN` on an int
}

__global__ void k_dep_scatter(const Deposit* __restrict__ dep, const unsigned long long* n_dep, unsigned long long cap,
                              const uint32_t* __restrict__ start, uint32_t* cursor, Deposit* __restrict__ sorted) {
    unsigned long long n = min(*n_dep, cap);
    for (unsigned long long i = blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x; i < n;
         i += (unsigned long long)gridDim.x * blockDim.x) {
        const Deposit d = dep[i];
        const uint32_t bk = d.b.bucket & 0xffffffu;
        if (bk == DEP_HOLE) continue;
        const uint32_t s = start[bk] + atomicAdd(&cursor[bk], 1u);
        sorted[s] = d;
    }
}

// Marks the unused tail of every wave slot's open chunk as holes (run once per eye pass / photon batch,
// after the last trace launch, before the records are sorted).  Exactly one of dep / hits is set.
__device__ __forceinline__ void close_chunks_block(const unsigned long long* __restrict__ wave_chunks, uint32_t n_slots, Deposit* dep, RawHit* hits,
                                                   unsigned long long cap, uint32_t block) {
    const uint32_t w = (block * blockDim.x + threadIdx.x) >> 6;
    if (w >= n_slots) return;
    const unsigned long long a = wave_chunks[2 * w], b = wave_chunks[2 * w + 1];
    for (unsigned long long h = a + lane_id(); h < b; h += 64)
        if (h < cap) {
            if (dep) dep[h].b.bucket = DEP_HOLE;
            else hits[h].pix = -1;
        }
}
__global__ void k_close_chunks(const unsigned long long* __restrict__ wave_chunks, uint32_t n_slots, Deposit* dep, RawHit* hits, unsigned long long cap) {
    close_chunks_block(wave_chunks, n_slots, dep, hits, cap, blockIdx.x);
}
// The first launch of a scan and the closing of the open chunks, which have nothing to do with each other but the
// place in the queue: the first nb workgroups sum histogram tiles, the others mark chunk tails.
__global__ void __launch_bounds__(SCAN_THREADS) k_scan_reduce_close(const uint32_t* __restrict__ cnt, uint32_t n, uint32_t* __restrict__ partial, uint32_t nb,
                                                                    const unsigned long long* __restrict__ wave_chunks, uint32_t n_slots, Deposit* dep,
                                                                    RawHit* hits, unsigned long long cap) {
    if (blockIdx.x < nb) scan_reduce_tile(cnt, n, partial, blockIdx.x);
    else close_chunks_block(wave_chunks, n_slots, dep, hits, cap, blockIdx.x - nb);
}

// ------------------------------------------------------------------ gather
// The radius search + HitPoint update of Raytracer.h:146-159, turned around: HitPoints own the
// accumulation, and the HitPoints of one grid cell share one sweep over the deposits around them.
//   dist^2 < searchR2  &&  hp.n . N > 1e-3  &&  dist^2 <= hp.R2   ->   newN++, tao += wgt * flux * PI_INV
// A wave owns a tile of 64 grid-sorted HitPoints and walks the distinct cells in it.  For a cell,
// lanes 0..26 look up the deposit ranges of the 27 neighbour cells; the ranges are cut into rows
// of 64 deposits (contiguous after the sort: one coalesced 32 B + 32 B load per lane) and
// GATHER_ROWS rows at a time are held in registers while every HitPoint of the cell is broadcast
// across the wave and tested against them.  Per (HitPoint, pass) the accepted flux is summed over
// the 16 lanes of each DPP row, the counts with ballots; one lane per row keeps that row's running
// sums of the HitPoint, the four meet when the tile is done and the owner lane writes its HitPoint
// once: no atomics on HitPoints, and a deposit is fetched once per cell instead of once per HitPoint.
#ifndef GATHER_ROWS
#define GATHER_ROWS 3        // 78 VGPRs: three gather workgroups and one trace workgroup fit a CU's register files together
#endif
#ifndef GATHER_DEFER_WGT
#define GATHER_DEFER_WGT 1   // sum the deposits' flux and multiply by the HitPoint's weight once (Raytracer.h:156 summed over the photons)
#endif
#ifndef GATHER_TILE
#define GATHER_TILE 16u      // HitPoints per tile (<= 16: one accumulator lane per HitPoint in every DPP row)
#endif

// Sum over the 16 lanes of each DPP row, left in every lane of the row.
__device__ __forceinline__ float row_total_f(float v) {
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xf, 0xf, true));    // quad_perm [1,0,3,2]
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xf, 0xf, true));    // quad_perm [2,3,0,1]
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xf, 0xf, true));   // row_half_mirror
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xf, 0xf, true));   // row_mirror
    return v;
}

// One tile = GATHER_TILE consecutive grid-sorted HitPoints, one per lane of the first GATHER_TILE lanes.
// A cell whose neighbourhood holds very many deposits (a caustic focus) would keep one wave busy long after
// the others have finished: the wave that owns the cell's HitPoints sweeps only the first GATHER_SPLIT_ROWS rows
// itself and queues the rest as pieces of GATHER_SPLIT_ROWS rows.  k_gather_split, launched behind k_gather,
// spreads the pieces over all waves (PIECE = true: one cell of the tile, rows [row_begin, row_end), sums
// added to the HitPoints with atomics; the owners' plain read-modify-write is over by then).
#ifndef GATHER_SPLIT_ROWS
#define GATHER_SPLIT_ROWS 24u
#endif
#ifndef GATHER_OWN_ROWS
#define GATHER_OWN_ROWS 24u     // rows the owner of a cell sweeps itself before it queues the rest
#endif
static_assert(GATHER_SPLIT_ROWS % GATHER_ROWS == 0 && GATHER_OWN_ROWS % GATHER_ROWS == 0, "a sweep is cut between passes of GATHER_ROWS rows");
struct GatherPiece { uint32_t tile_base, lead, row_begin, row_end; };   // lead: lane of one HitPoint of the cell in its tile

// FIXED_R2: every HitPoint still has R2 = INIT_R2 (the reference's case: update() never runs), so `dist^2 <= R2` is
// implied by the search radius test and the radii are not even loaded.
template <bool PIECE, bool FIXED_R2>
__device__ __forceinline__ void gather_tile(const uint32_t base, const HpRec* __restrict__ hp,
                                            const double* __restrict__ hp_r2,
                                            float4* __restrict__ hp_tao, const uint32_t n_hp,
                                            const uint32_t* __restrict__ dstart, const uint32_t* __restrict__ dlen,
                                            const Deposit* __restrict__ deps,
                                            const GridParams g, GatherPiece* __restrict__ pieces, unsigned long long* n_pieces,
                                            const uint32_t piece_cap, unsigned int* overflow, const GatherPiece piece,
                                            uint32_t& cand, uint32_t& acc_n, uint32_t& row_tests) {
    const int lane = lane_id();
    const uint32_t mine = base + lane;
    const bool valid = (uint32_t)lane < GATHER_TILE && mine < n_hp;
    const HpRec mrec = hp[valid ? mine : base];
    const PosKey mp = mrec.a;
    const float4 mnr = mrec.nr;
    const double mr2 = FIXED_R2 ? SPPM_INIT_R2 : hp_r2[valid ? mine : base];
    const float4 mw = mrec.wgt;
    // Running sums of HitPoint j: its flux in lanes j, 16 + j, 32 + j, 48 + j, one partial sum per DPP row (j < GATHER_TILE <= 16),
    // so that a sweep only reduces inside the rows of 16 lanes (four DPP adds per channel instead of a whole-wave reduction
    // and a broadcast); the four partial sums meet once, when the tile is done.  The count, which comes from ballots, is kept
    // in all four lanes alike.
    static_assert(GATHER_TILE <= 16, "one accumulator lane per DPP row and HitPoint");
    float tr = 0, tg = 0, tb = 0;
    uint32_t tn = 0;
    const int lane15 = lane & 15;
    uint64_t todo = __ballot(valid);
    if (PIECE) todo &= 1ull << (piece.lead & 63u);               // only the cell the piece belongs to
    while (todo) {
        const int lead = __ffsll((unsigned long long)todo) - 1;
        const uint64_t key = bcast_u64(mp.key, lead);
        const uint64_t grp = __ballot(valid && mp.key == key);   // the HitPoints of this cell in the tile
        todo &= ~grp;
        uint32_t nb_s = 0, nb_len = 0;
        uint64_t nb_key = 0;
        if (lane < 27) {
            const int dx = lane % 3 - 1, dy = (lane / 3) % 3 - 1, dz = lane / 9 - 1;
            nb_key = key + ((uint64_t)(int64_t)dx << 42) + ((uint64_t)(int64_t)dy << 21) + (uint64_t)(int64_t)dz;
            const uint32_t b = key_bucket(nb_key, g.mask);
            // three loads asked for together, one wait: the bucket's room and how much of it is filled (directly placed deposits
            // fill their bucket's room only partly; after a counting sort the fill count IS the room)
            const uint32_t s0 = dstart[b], s1 = dstart[b + 1], filled = dlen[b];
            nb_s = s0;
            nb_len = min(s1 - s0, filled);
        }
        // The non-empty ranges, concatenated, form one flat list; a row is 64 consecutive entries of it
        // (lane l of the row at flat offset F holds entry F + l), so rows are full except the last.
        uint64_t cells = __ballot(nb_len > 0);
        uint32_t F = 0;                     // flat offset of the next row
        uint32_t cur_end = 0, cur_base = 0; // current range: flat end, (deposit index - flat index)
        uint32_t cur_klo = 0, cur_khi = 0;  // its cell key
        uint32_t F_limit = GATHER_OWN_ROWS * 64u;   // this sweep stops here
        if (PIECE) {
            F = piece.row_begin * 64u; F_limit = piece.row_end * 64u;
            while (cur_end <= F && cells) {           // walk to the range that holds the first entry wanted
                const int k = __ffsll((unsigned long long)cells) - 1;
                cells &= cells - 1;
                const uint32_t start = cur_end;
                cur_base = bcast_u(nb_s, k) - start;
                cur_end = start + bcast_u(nb_len, k);
                cur_klo = bcast_u((uint32_t)nb_key, k);
                cur_khi = bcast_u((uint32_t)(nb_key >> 32), k);
            }
        }
        while (cells || F < cur_end) {
            double ax[GATHER_ROWS], ay[GATHER_ROWS], az[GATHER_ROWS];
            float bnx[GATHER_ROWS], bny[GATHER_ROWS], bnz[GATHER_ROWS], bfr[GATHER_ROWS], bfg[GATHER_ROWS], bfb[GATHER_ROWS];
            int rows = 0;
#pragma unroll
            for (int r = 0; r < GATHER_ROWS; r++) {
                ax[r] = ay[r] = az[r] = 1e30;
                bnx[r] = bny[r] = bnz[r] = bfr[r] = bfg[r] = bfb[r] = 0.0f;
                if (cells || F < cur_end) {
                    rows = r + 1;
                    const uint32_t f = F + lane;
                    uint32_t src = cur_base + f, klo = cur_klo, khi = cur_khi;
                    bool in = f < cur_end;
                    while (cur_end < F + 64 && cells) {   // ranges that begin inside this row
                        const int k = __ffsll((unsigned long long)cells) - 1;
                        cells &= cells - 1;
                        const uint32_t start = cur_end;
                        cur_base = bcast_u(nb_s, k) - start;
                        cur_end = start + bcast_u(nb_len, k);
                        cur_klo = bcast_u((uint32_t)nb_key, k);
                        cur_khi = bcast_u((uint32_t)(nb_key >> 32), k);
                        if (f >= start) { src = cur_base + f; klo = cur_klo; khi = cur_khi; in = f < cur_end; }
                    }
                    F += 64;
                    if (in) {
                        const PosKey a = deps[src].a;
                        const DepB b = deps[src].b;   // fetched with the position, not after the key check: one memory latency per row
                        if (a.key == (((uint64_t)khi << 32) | klo)) {   // else: another cell hashed into this bucket
                            ax[r] = a.x; ay[r] = a.y; az[r] = a.z;
                            bnx[r] = b.nx; bny[r] = b.ny; bnz[r] = b.nz;
                            bfr[r] = b.fr * (float)SPPM_PI_INV; bfg[r] = b.fg * (float)SPPM_PI_INV; bfb[r] = b.fb * (float)SPPM_PI_INV;
                        }
                    }
                }
            }
            uint64_t gm = grp;
            while (gm) {
                const int j = __ffsll((unsigned long long)gm) - 1;
                gm &= gm - 1;
                const double hx = bcast_d(mp.x, j), hy = bcast_d(mp.y, j), hz = bcast_d(mp.z, j);
                const float nx = bcast_f(mnr.x, j), ny = bcast_f(mnr.y, j), nz = bcast_f(mnr.z, j);
                const double R2 = FIXED_R2 ? SPPM_INIT_R2 : bcast_d(mr2, j);
#if !GATHER_DEFER_WGT
                const float wr = bcast_f(mw.x, j), wg = bcast_f(mw.y, j), wb = bcast_f(mw.z, j);
#endif
                float pr = 0, pg = 0, pb = 0;
                uint32_t pn = 0;
                row_tests += (uint32_t)rows;
#pragma unroll
                for (int r = 0; r < GATHER_ROWS; r++) {
                    if (r < rows) {
                        const double vx = hx - ax[r], vy = hy - ay[r], vz = hz - az[r];
                        const double d2 = vx * vx + vy * vy + vz * vz;
                        const bool c = d2 < SPPM_INIT_R2;      // flann radiusSearch(radius = searchR2): strict '<' on the squared distance
                        // ballots of the bare compares, combined as scalars (a ballot of their conjunction costs two more vector instructions)
                        const uint64_t cm = __builtin_amdgcn_ballot_w64(c);
                        const float dn = nx * bnx[r] + ny * bny[r] + nz * bnz[r];
                        const bool facing = dn > 1e-3f, inr = FIXED_R2 ? true : (d2 <= R2);
                        const bool ok = c & facing & inr;
                        cand += (uint32_t)__popcll(cm);
                        pn += FIXED_R2 ? (uint32_t)__popcll(cm & __builtin_amdgcn_ballot_w64(facing))
                                       : (uint32_t)__popcll(cm & __builtin_amdgcn_ballot_w64(facing) & __builtin_amdgcn_ballot_w64(inr));
                        if (ok) {   // three operations under the lane mask (no per-channel select)
#if GATHER_DEFER_WGT
                            pr += bfr[r]; pg += bfg[r]; pb += bfb[r];
#else
                            pr = fmaf(wr, bfr[r], pr);
                            pg = fmaf(wg, bfg[r], pg);
                            pb = fmaf(wb, bfb[r], pb);
#endif
                        }
                    }
                }
                if (pn) {
                    pr = row_total_f(pr); pg = row_total_f(pg); pb = row_total_f(pb);
                    if (lane15 == j) { tr += pr; tg += pg; tb += pb; tn += pn; }
                    acc_n += pn;
                }
            }
            if (F >= F_limit) break;   // checked once per pass: GATHER_SPLIT_ROWS is a multiple of GATHER_ROWS
        }
        if (!PIECE && (cells || F < cur_end)) {   // rows are left: queue them in pieces
            uint32_t left = cur_end - F;           // entries left in the current range ...
            for (uint64_t q = cells; q; q &= q - 1) left += bcast_u(nb_len, __ffsll((unsigned long long)q) - 1);   // ... and in those not yet opened
            const uint32_t n_p = (left + GATHER_SPLIT_ROWS * 64u - 1u) / (GATHER_SPLIT_ROWS * 64u);
            unsigned long long first = 0;
            if (lane == 0) first = atomicAdd(n_pieces, (unsigned long long)n_p);
            first = bcast_u64(first, 0);
            if (first + n_p <= piece_cap) {
                for (uint32_t i = lane; i < n_p; i += 64) {
                    GatherPiece pc;
                    pc.tile_base = base; pc.lead = (uint32_t)lead;
                    pc.row_begin = GATHER_OWN_ROWS + i * GATHER_SPLIT_ROWS; pc.row_end = pc.row_begin + GATHER_SPLIT_ROWS;
                    pieces[first + i] = pc;
                }
            } else {
                atomicOr(overflow, 8u);             // the piece queue is sized for 2^20 pieces of 6144 deposits: report, do not drop silently
            }
        }
    }
    tr += __shfl_xor(tr, 16, 64); tg += __shfl_xor(tg, 16, 64); tb += __shfl_xor(tb, 16, 64);
    tr += __shfl_xor(tr, 32, 64); tg += __shfl_xor(tg, 32, 64); tb += __shfl_xor(tb, 32, 64);
    if (valid && tn) {
#if GATHER_DEFER_WGT
        tr *= mw.x; tg *= mw.y; tb *= mw.z;        // the HitPoint's weight, the same for every deposit, multiplies the flux sums once
#endif
        if (PIECE) {
            float* t = (float*)&hp_tao[mine];
            atomicAdd(t + 0, tr); atomicAdd(t + 1, tg); atomicAdd(t + 2, tb);
            atomicAdd((int*)(t + 3), (int)tn);
        } else {
            float4 t = hp_tao[mine];
            t.x += tr; t.y += tg; t.z += tb;
            t.w = __int_as_float(__float_as_int(t.w) + (int)tn);
            hp_tao[mine] = t;
        }
    }
}

// Persistent waves; tiles are handed out by eight counters, one per eighth of the tile range.  A
// workgroup starts on the eighth of its XCD (workgroups are dealt round-robin to the 8 XCDs), so the
// tiles in flight on one XCD are neighbours in bucket order and share its L2, and moves on to the
// other eighths when its own is drained (tile cost varies with the photon density around it).
template <bool FIXED_R2>
__global__ void __launch_bounds__(256) k_gather(const HpRec* __restrict__ hp,
                                                const double* __restrict__ hp_r2, float4* __restrict__ hp_tao, uint32_t n_hp,
                                                const uint32_t* __restrict__ dstart, const uint32_t* __restrict__ dlen,
                                                const Deposit* __restrict__ deps,
                                                GridParams g, uint32_t* tile_ctr, GatherPiece* __restrict__ pieces,
                                                unsigned long long* n_pieces, uint32_t piece_cap, DevCounters* counters) {
    const uint32_t n_tiles = (n_hp + GATHER_TILE - 1u) / GATHER_TILE;
    const uint32_t per = (n_tiles + 7u) >> 3;
    uint32_t cand = 0, acc_n = 0, row_tests = 0;   // wave-uniform tallies
    for (uint32_t s = 0; s < 8; s++) {
        const uint32_t q = (blockIdx.x + s) & 7u;
        const uint32_t q_end = min(n_tiles, (q + 1) * per);
        // A wave asks for its next tile before it works on the one it has, so that the counter's answer travels while the
        // tile's HitPoints are loaded (the asm statement keeps the compiler from waiting for it where it is asked for).  Every
        // wave thus claims one tile number beyond the eighth's end before it moves on, which nobody minds.  Only where every
        // wave has several tiles to come, though: a tile reserved ahead is a tile another wave cannot take, and with about as
        // many tiles as waves (a 256 x 256 picture) that leaves a third of the waves idle while the others do two.
        const bool ahead = n_tiles >= 4u * ((gridDim.x * blockDim.x) >> 6);
        uint32_t claimed = 0;
        if (lane_id() == 0) claimed = atomicAdd(&tile_ctr[q * 16], 1u);
        for (;;) {
            asm volatile("" : "+v"(claimed));
            const uint32_t t = bcast_u(claimed, 0) + q * per;
            if (t >= q_end) break;
            if (ahead && lane_id() == 0) claimed = atomicAdd(&tile_ctr[q * 16], 1u);
            gather_tile<false, FIXED_R2>(t * GATHER_TILE, hp, hp_r2, hp_tao, n_hp, dstart, dlen, deps, g, pieces, n_pieces, piece_cap,
                               &counters->overflow, GatherPiece{0, 0, 0, 0}, cand, acc_n, row_tests);
            if (!ahead && lane_id() == 0) claimed = atomicAdd(&tile_ctr[q * 16], 1u);
        }
    }
    if (lane_id() == 0) {
        if (cand) atomicAdd(&counters->candidates, (unsigned long long)cand);
        if (acc_n) atomicAdd(&counters->accepted, (unsigned long long)acc_n);
        if (row_tests) atomicAdd(&counters->row_tests, (unsigned long long)row_tests);
    }
}

// The same update as the gather's, from the deposit's side, for the few deposits that found their bucket full: one
// wave per deposit walks the HitPoints of the 27 cells around it (the HitPoint array is sorted by bucket: hstart).
// Runs at the end of k_gather_split, whose waves use atomics on the HitPoints as well.
__device__ __forceinline__ void gather_overflow(const Deposit* __restrict__ over, const unsigned long long* n_over, unsigned long long over_cap,
                                                unsigned long long* over_max, const uint32_t* __restrict__ hstart, const HpRec* __restrict__ hp_rec,
                                                const double* __restrict__ hp_r2,
                                                float4* __restrict__ hp_tao, GridParams g,
                                                DevCounters* counters) {
    const unsigned long long n = min(*n_over, over_cap);
    if (blockIdx.x == 0 && threadIdx.x == 0) atomicMax(over_max, n);   // the host drops a layout that overflows much
    const unsigned long long n_waves = ((unsigned long long)gridDim.x * blockDim.x) >> 6;
    const int lane = lane_id();
    unsigned long long cand = 0, acc = 0;
    for (unsigned long long i = ((unsigned long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6; i < n; i += n_waves) {
        const PosKey a = over[i].a;
        const DepB b = over[i].b;
        // lanes 0..26 look up the HitPoint ranges of the 27 cells together (one round of loads instead of 27 in a row); the
        // ranges, laid end to end, are then walked 64 HitPoints at a time
        uint32_t hs = 0, hl = 0;
        uint64_t ck = 0;
        if (lane < 27) {
            const int dx = lane % 3 - 1, dy = (lane / 3) % 3 - 1, dz = lane / 9 - 1;
            ck = a.key + ((uint64_t)(int64_t)dx << 42) + ((uint64_t)(int64_t)dy << 21) + (uint64_t)(int64_t)dz;
            const uint32_t bk = key_bucket(ck, g.mask);
            hs = hstart[bk];
            hl = hstart[bk + 1] - hs;
        }
        const uint64_t cells = __ballot(hl > 0);
        uint32_t total = 0;
        for (uint64_t q = cells; q; q &= q - 1) total += bcast_u(hl, __ffsll((unsigned long long)q) - 1);
        for (uint32_t f0 = 0; f0 < total; f0 += 64) {
            const uint32_t f = f0 + (uint32_t)lane;
            uint32_t h = 0, klo = 0, khi = 0, fs = 0;
            bool in = false;
            for (uint64_t q = cells; q; q &= q - 1) {       // which range holds flat entry f
                const int k = __ffsll((unsigned long long)q) - 1;
                const uint32_t len = bcast_u(hl, k), start = bcast_u(hs, k);
                const uint32_t lo = bcast_u((uint32_t)ck, k), hi = bcast_u((uint32_t)(ck >> 32), k);
                if (f >= fs && f < fs + len) { h = start + (f - fs); klo = lo; khi = hi; in = true; }
                fs += len;
            }
            if (!in) continue;
            const PosKey hp = hp_rec[h].a;
            if (hp.key != (((uint64_t)khi << 32) | klo)) continue;   // another cell hashed into this bucket
            const double vx = hp.x - a.x, vy = hp.y - a.y, vz = hp.z - a.z;
            const double d2 = vx * vx + vy * vy + vz * vz;
            if (!(d2 < SPPM_INIT_R2)) continue;
            cand++;
            const float4 nr = hp_rec[h].nr;
            if (!(nr.x * b.nx + nr.y * b.ny + nr.z * b.nz > 1e-3f) || !(d2 <= hp_r2[h])) continue;
            acc++;
            const float4 w = hp_rec[h].wgt;
            float* t = (float*)&hp_tao[h];
            atomicAdd(t + 0, w.x * (b.fr * (float)SPPM_PI_INV));
            atomicAdd(t + 1, w.y * (b.fg * (float)SPPM_PI_INV));
            atomicAdd(t + 2, w.z * (b.fb * (float)SPPM_PI_INV));
            atomicAdd((int*)(t + 3), 1);
        }
    }
    if (cand) atomicAdd(&counters->candidates, cand);
    if (acc) atomicAdd(&counters->accepted, acc);
}

// The queued pieces of heavy cells (see GATHER_SPLIT_ROWS), one wave per piece at a time.  SHARE: held to 80 registers (a
// few values spill) like k_gather, so that three of its workgroups fit a CU's register files beside a trace workgroup:
// the variant for contexts that share their GPU; a context that is alone launches the one that spills nothing.
template <bool FIXED_R2, bool SHARE>
__global__ void __launch_bounds__(256, SHARE ? 6 : 1) k_gather_split(const HpRec* __restrict__ hp,
                                                      const double* __restrict__ hp_r2, float4* __restrict__ hp_tao, uint32_t n_hp,
                                                      const uint32_t* __restrict__ dstart, const uint32_t* __restrict__ dlen,
                                                      const Deposit* __restrict__ deps, GridParams g,
                                                      const GatherPiece* __restrict__ pieces, const unsigned long long* __restrict__ n_pieces,
                                                      uint32_t piece_cap, unsigned long long* next_piece, DevCounters* counters,
                                                      const Deposit* __restrict__ over, const unsigned long long* n_over,
                                                      unsigned long long over_cap, unsigned long long* over_max,
                                                      const uint32_t* __restrict__ hstart) {
    const unsigned long long n = min(*n_pieces, (unsigned long long)piece_cap);
    const unsigned long long n_waves = ((unsigned long long)gridDim.x * blockDim.x) >> 6;
    uint32_t cand = 0, acc_n = 0, row_tests = 0;
    // wave w starts with piece w; only when there are more pieces than waves does the shared counter come into
    // play (a launch with nothing queued costs no atomic at all)
    for (unsigned long long i = ((unsigned long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6; i < n;) {
        const GatherPiece pc = pieces[i];
        gather_tile<true, FIXED_R2>(pc.tile_base, hp, hp_r2, hp_tao, n_hp, dstart, dlen, deps, g, nullptr, nullptr, 0u, nullptr, pc, cand, acc_n, row_tests);
        if (n <= n_waves) break;
        unsigned long long nx = 0;
        if (lane_id() == 0) nx = atomicAdd(next_piece, 1ull);
        i = n_waves + bcast_u64(nx, 0);
    }
    if (lane_id() == 0) {
        if (cand) atomicAdd(&counters->candidates, (unsigned long long)cand);
        if (acc_n) atomicAdd(&counters->accepted, (unsigned long long)acc_n);
        if (row_tests) atomicAdd(&counters->row_tests, (unsigned long long)row_tests);
    }
    if (over) gather_overflow(over, n_over, over_cap, over_max, hstart, hp, hp_r2, hp_tao, g, counters);
}

// Room per bucket for direct placement, from the histogram of ALL diffuse hits of a batch, reachable by a HitPoint
// or not: which buckets are reachable changes with every camera jitter, where the photons land does not.
// A bucket some HitPoint reaches now (occ) gets the count plus a margin that a Poisson count of the same mean all but
// never exceeds, and a few slots if it had no hit (far down the open tunnel every batch hits other cells, one photon
// each).  A bucket out of reach stores nothing as long as it stays so: it gets the count and the small margin `far`
// (one slot if it had no hit), enough for the buckets at the rim of the HitPoints' reach that a jitter brings in; what
// does not fit goes to the overflow list either way.  slack_sigma = 0 leaves no margin at all (tests use it to send
// deposits down the overflow path).
__global__ void k_bucket_room(const uint32_t* __restrict__ count, const uint8_t* __restrict__ occ, uint32_t n, float slack_sigma, uint32_t slack_add,
                              uint32_t slack_empty, float far_sigma, uint32_t far_add, uint32_t far_empty, uint32_t* __restrict__ room) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t c = count[i];
    if (!occ || occ[i]) room[i] = c ? c + (uint32_t)ceilf(slack_sigma * sqrtf((float)c)) + slack_add : slack_empty;
    else room[i] = c ? c + (uint32_t)ceilf(far_sigma * sqrtf((float)c)) + far_add : far_empty;
}

// ------------------------------------------------------------------ resolve (Raytracer.h:281-293)
// ONE_PER_PIXEL: no material of the scene lets a pixel leave more than one HitPoint (the reference's scene), so a pixel's
// value is its HitPoint's and is stored, not added with atomics.
template <bool ONE_PER_PIXEL>
__global__ void k_resolve(const double* hp_r2, const HpRec* hp, const float4* hp_tao, uint32_t n_hp, double inv_norm, float* pic) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_hp) return;
    float4 t = hp_tao[i];
    int pix = __float_as_int(hp[i].wgt.w);
    double s = inv_norm / hp_r2[i];   // 1 / PI / (R2 * cnt * onetime)
    float* p = pic + 3 * (size_t)pix;
    if (ONE_PER_PIXEL) {
        p[0] = (float)(t.x * s); p[1] = (float)(t.y * s); p[2] = (float)(t.z * s);
    } else {
        atomicAdd(p + 0, (float)(t.x * s));
        atomicAdd(p + 1, (float)(t.y * s));
        atomicAdd(p + 2, (float)(t.z * s));
    }
}

// ------------------------------------------------------------------ film: image sum and tone-map (Raytracer.h:449-470)
// film += resolve: the image sum of SPPMRayTracer::render (Raytracer.h:449-458) taken straight from the HitPoints, in fp64
// as the reference's camera.pic.  Contexts on other streams add to the same film, hence the atomics.
__global__ void k_film_add(const double* hp_r2, const HpRec* hp, const float4* hp_tao, uint32_t n_hp, double inv_norm, double* film) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_hp) return;
    float4 t = hp_tao[i];
    if (t.x == 0.f && t.y == 0.f && t.z == 0.f) return;
    int pix = __float_as_int(hp[i].wgt.w);
    double s = inv_norm / hp_r2[i];   // 1 / PI / (R2 * cnt * onetime)
    unsafeAtomicAdd(&film[3 * (size_t)pix + 0], (double)t.x * s);
    unsafeAtomicAdd(&film[3 * (size_t)pix + 1], (double)t.y * s);
    unsafeAtomicAdd(&film[3 * (size_t)pix + 2], (double)t.z * s);
}

// toInt(film / n_images) of Raytracer.h:11-13 into the byte image the reference hands to cv::imwrite (:460-466): rows
// flipped, channel k at byte k (rgb = 0) or at byte 2 - k (rgb = 1, what a PPM wants).  exp / pow of the device library
// may differ from the host's in the last place, which matters only when the value sits on a rounding step of the
// truncation: those elements are listed in `flagged` ([0] = count) and recomputed by the host.
__global__ void k_tonemap(const double* film, int w, int h, double n_images, int rgb, uint8_t* out, uint32_t* flagged, uint32_t flag_cap) {
    const size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (e >= (size_t)w * h * 3) return;
    const int k = (int)(e % 3);
    const size_t p = e / 3;
    const int j = (int)(p % (size_t)w), i1 = (int)(p / (size_t)w);
    const double v = pow(1.0 - exp(-(film[e] / n_images)), 1.0 / 2.2) * 255.0 + 0.5;
    if (!(fabs(v - rint(v)) >= 1e-6)) {   // also NaN
        uint32_t s = atomicAdd(&flagged[0], 1u);
        if (s < flag_cap) flagged[1 + s] = (uint32_t)e;
    }
    const int iv = (int)v;
    out[((size_t)(h - 1 - i1) * w + j) * 3 + (rgb ? 2 - k : k)] = (uint8_t)iv;
}

__global__ void k_fill_u32(uint32_t* p, size_t n, uint32_t v) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}
// Zeroes up to three ranges of 32-bit words in one launch (the control block, the tallies and the histograms of a pass).
__global__ void k_zero3(uint32_t* p0, size_t n0, uint32_t* p1, size_t n1, uint32_t* p2, size_t n2) {
    const size_t stride = (size_t)gridDim.x * blockDim.x, t = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    for (size_t i = t; i < n0; i += stride) p0[i] = 0u;
    for (size_t i = t; i < n1; i += stride) p1[i] = 0u;
    for (size_t i = t; i < n2; i += stride) p2[i] = 0u;
}

// ------------------------------------------------------------------ parity hooks
template <bool IN_LDS>
__global__ void __launch_bounds__(256) k_debug_near_collide(SceneParams sp, uint32_t n, const double* org, const double* dir, uint32_t rng,
                                                            sppm_collision* out, DevCounters* counters) {
    SceneView S = stage_scene<IN_LDS>(sp);
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool active = i < n;
    d3 o = active ? ld3(org + 3 * (size_t)i) : mk(0, 0, 0), d = active ? ld3(dir + 3 * (size_t)i) : mk(0, 0, 1);
    uint32_t ncalls = 0;
    Hit h = near_collide(S, active, o, d, rng_path(rng, i), 0u, ncalls);
    if (active) {
        sppm_collision c;
        c.pos[0] = h.pos.x; c.pos[1] = h.pos.y; c.pos[2] = h.pos.z;
        c.n[0] = h.n.x; c.n[1] = h.n.y; c.n[2] = h.n.z;
        float hc[3];
        hit_color(h, hc);
        c.color[0] = hc[0]; c.color[1] = hc[1]; c.color[2] = hc[2];
        c.dist = h.obj >= 0 ? h.dist : 0.0;
        c.obj = h.obj; c.inside = h.inside;
        out[i] = c;
    }
    flush_uniform(&counters->newton_calls, ncalls);
}

__global__ void __launch_bounds__(256) k_debug_newton(SceneParams sp, uint32_t n, const double* org, const double* dir, const int* patch,
                                                      uint32_t rng, double* out_tuv) {
    SceneView S = stage_scene<false>(sp);   // only the Newton tables are used, and those are in global memory anyway
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool active = i < n;
    const int lane = lane_id();
    double ox = 0, oy = 0, oz = 0, dx = 0, dy = 0, dz = 1;
    int p = 0;
    if (active) {
        ox = org[3 * (size_t)i]; oy = org[3 * (size_t)i + 1]; oz = org[3 * (size_t)i + 2];
        dx = dir[3 * (size_t)i]; dy = dir[3 * (size_t)i + 1]; dz = dir[3 * (size_t)i + 2];
        p = patch[i];
    }
    uint32_t sub = rng_sub(rng, i, 0u, P_NEWTON);
    float mt = (float)SPPM_MAX_NUM, mu = (float)SPPM_MAX_NUM, mv = (float)SPPM_MAX_NUM;
    uint64_t m = __ballot(active);
    while (m) {
        int src = __ffsll((unsigned long long)m) - 1;
        m &= m - 1;
        const d3 so = mk(bcast_d(ox, src), bcast_d(oy, src), bcast_d(oz, src));
        const d3 sd = mk(bcast_d(dx, src), bcast_d(dy, src), bcast_d(dz, src));
        float t, u, v;
        newton_coop(S.pw_q, S.pw_c, bcast_i(p, src), so, sd, bcast_u(sub, src), t, u, v);
        if (lane == src && t < (float)SPPM_MAX_NUM) { mt = t; mu = u; mv = v; }
    }
    if (active) {
        const bool none = !(mt < (float)SPPM_MAX_NUM);
        out_tuv[3 * (size_t)i + 0] = none ? SPPM_MAX_NUM : (double)mt;
        out_tuv[3 * (size_t)i + 1] = none ? SPPM_MAX_NUM : (double)mu;
        out_tuv[3 * (size_t)i + 2] = none ? SPPM_MAX_NUM : (double)mv;
    }
}

// =================================================================== host side
namespace {

template <class T>
struct DevBuf {
    T* p = nullptr;
    size_t n = 0;
    hipError_t reserve(size_t want) {
        if (want <= n) return hipSuccess;
        if (p) (void)hipFree(p);
        p = nullptr; n = 0;
        hipError_t e = hipMalloc((void**)&p, want * sizeof(T));
        if (e == hipSuccess) n = want;
        return e;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; n = 0; }
    // For sizes that wander from call to call (the HitPoint count moves with the camera jitter): growing a buffer frees
    // the old one, which waits for every stream of the GPU, so a buffer that must grow grows past what is asked for.
    hipError_t reserve_with_room(size_t want) { return want <= n ? hipSuccess : reserve(want + want / 16 + 4096); }
};

uint32_t next_pow2(uint64_t v) { uint32_t p = 1; while (p < v) p <<= 1; return p; }

}  // namespace

// ------------------------------------------------------------------ static scratch of sppm_near_collide_noalloc
// The staging areas (rays in, collision records out) and the tally block of the entry point that allocates nothing live
// in one file-scope device array: every GPU the library runs on holds its own copy, at its own address, until the
// process ends.  Host code never passes the array's NAME to a kernel or a copy (on the host that name is a stand-in, not
// a device address): sppm_create asks the runtime for the address once per device (hipGetSymbolAddress, with that device
// current) and keeps it in g_scratch_ptr; every use below goes through that pointer.
#define SPPM_NOALLOC_RAYS SPPM_NOALLOC_MAX_RAYS
struct NoAllocScratch {
    DevCounters counters;                              // tallies of the launch; must start at zero: cleared by a fill kernel
    double org[3 * SPPM_NOALLOC_RAYS];
    double dir[3 * SPPM_NOALLOC_RAYS];
    sppm_collision out[SPPM_NOALLOC_RAYS];
};
static_assert(sizeof(NoAllocScratch) < (4u << 20), "the static scratch stays modest: under 4 MiB per GPU");
static __device__ __align__(256) unsigned char scratch[sizeof(NoAllocScratch)];
static void* g_scratch_ptr[64];                        // per device: the address of that GPU's copy of `scratch`
static std::mutex g_scratch_mu[64];                    // calls that use a GPU's copy take turns
static std::mutex g_scratch_init_mu;

// adds a launch's Newton tally to the context's counters (what k_debug_near_collide does directly in the allocating entry point)
__global__ void k_add_newton_calls(DevCounters* dst, const DevCounters* src) {
    if (blockIdx.x == 0 && threadIdx.x == 0 && src->newton_calls) atomicAdd(&dst->newton_calls, src->newton_calls);
}

// contexts alive per GPU of this process (see sppm_ctx::trace_blocks_per_cu)
static std::atomic<int> g_live_contexts[64];
static bool shares_gpu(int device) { return device >= 0 && device < 64 && g_live_contexts[device].load(std::memory_order_relaxed) > 1; }

#define CTL_SCALARS 16
struct sppm_ctx {
    int device = 0;
    hipStream_t stream = nullptr;         // the stream every pass is queued on: own_stream, or the caller's (sppm_set_stream)
    hipStream_t own_stream = nullptr;
    std::string err;
    // scene
    bool have_scene = false;
    DevBuf<char> blob; int blob_bytes = 0, off_box = 0, off_cp = 0, off_pobj = 0, off_aobj = 0, off_sub = -1, n_objs = 0, n_patches = 0;
    bool scene_in_lds = true;             // the scene tables fit the LDS staging area (else: read from global memory)
    size_t scene_lds_max = SCENE_LDS_MAX; // that area's size on this device (sppm_create)
    bool coop_only = false;               // more patches than a parked ray's patch mask can name: rays are never parked
    DevBuf<uchar4> texels;
    DevBuf<float> pw_q; DevBuf<double> pw_c;   // Newton tables (power basis around the patch centres)
    DevBuf<float> sub_box;                     // boxes of the sub-patches (subbox_prune); SPPM_SUBBOX=0: not used
    bool use_sub_box = true;
    DevBuf<DevLight> lights; int n_lights = 0;
    DevBuf<double> sincos_tab;            // table of sincos_2pi_u24, filled once in sppm_create
    // eye pass
    bool have_eye = false;
    sppm_camera cam{};
    DevBuf<RawHit> raw;
    DevBuf<ParkRec> park[2]; DevBuf<NewtonTask> ntask[2]; DevBuf<float4> nres;
    uint64_t park_scale = 1;          // the eye pass's branch queue has been doubled this many times over (run_trace)
    uint64_t branch_queue_min = 1u << 16;   // its smallest size (SPPM_BRANCH_QUEUE_MIN: tests shrink it to see it grow)
    double dep_per_photon = 4.0;      // deposit slots per photon of a batch (measured 3.7 diffuse hits per photon, 1.9 stored, in the reference's
                                      // open tunnel; at most 14): grows when a counting-sorted batch overflows them (sppm_photon_pass)
    uint64_t coop_threshold = 4096;   // rounds with fewer parked rays finish inside the trace kernel
    bool force_coop = false;
    bool photon_coop = true;
    bool eye_scatter = true;              // eye work items visit the pixels in a scattered order (TraceParams::pix_mul)
    int newton_blocks_per_cu = 8;
    // Persistent grids, in workgroups per CU; 0 = chosen at every launch: a context that is alone on its GPU fills the
    // CUs (trace 2 per CU, gather 5), contexts that share a GPU (several iterations in flight) leave each other room: one
    // trace workgroup (<= 256 VGPRs per lane) and three gather workgroups (78) fit the register file of every SIMD
    // together, and a trace kernel of one context then runs beside the gather of another instead of after it.
    int gather_blocks_per_cu = 0;
    int trace_blocks_per_cu = 0;
    int trace_wg_threads = 256;           // threads of a trace workgroup (SPPM_TRACE_WG: 64, 128 or 256)
    int eye_blocks = 0;                   // at most so many workgroups for the eye trace (SPPM_EYE_BLOCKS; 0: by the per-CU rule)
    int eye_blocks_per_cu = 0;            // the eye pass's own value of trace_blocks_per_cu (0: the same rule)
    int scatter_blocks_per_cu = 32;
    bool verbose = false;
    DevBuf<uint32_t> hstart;
    int grid_w = 0, grid_h = 0;           // resolution of the eye pass the bucket count was chosen for
    // bucket occupancy bytes (M / 4 words) | HitPoint histogram | scatter cursors | deposit histogram | histogram of all
    // diffuse hits ((M + 1) words each)
    DevBuf<uint32_t> hist;
    size_t occ_words() const { return ((size_t)grid.mask + 1) / 4; }
    uint8_t* occ() const { return (uint8_t*)hist.p; }
    uint32_t* hcount() const { return hist.p + occ_words(); }
    uint32_t* cursor() const { return hcount() + ((size_t)grid.mask + 2); }
    uint32_t* dcount() const { return hcount() + 2 * ((size_t)grid.mask + 2); }
    uint32_t* dcount_all() const { return hcount() + 3 * ((size_t)grid.mask + 2); }
    bool dep_filter = true;               // deposits no HitPoint can reach are not stored (SPPM_DEP_FILTER=0 keeps them)
    bool has_dual_lobe = false;           // some material reflects and refracts: eye paths branch
    bool one_hp_per_pixel = false;        // no material both stores a HitPoint and sends the eye path on, none branches: a pixel leaves one HitPoint at most
    DevBuf<HpRec> hp_rec; DevBuf<float4> hp_tao; DevBuf<HpMeta> hp_meta; DevBuf<double> hp_r2;
    DevBuf<uint32_t> sorted_of_raw;       // place of every raw HitPoint slot in the sorted arrays (0xffffffff: an unused slot)
    bool meta_valid = false;              // hp_meta holds the HitPoints of the last eye pass (written on demand: ensure_meta)
    uint64_t raw_claimed = 0;             // slots of the raw array the last eye pass claimed
    bool radii_changed = false;           // an update() sweep has run on the HitPoints of the last eye pass: R2 != INIT_R2 for some
    int initial_n = 0;                    // HitPoint::N at creation (the reference: 0, which keeps update() from ever running)
    uint32_t rounds_per_update = 1;       // with initial_n > 0: rounds of a photon pass between two update() sweeps
    uint64_t n_hp = 0;
    GridParams grid{};
    // photon pass
    DevBuf<Deposit> dep, sdep;   // deposits of a batch as traced / sorted by bucket
    DevBuf<GatherPiece> pieces;  // queued row ranges of heavy cells (k_gather -> k_gather_split)
    // Direct placement of deposits (see TraceParams::place): the layout made from the histogram of an earlier batch
    bool place_enabled = true;            // SPPM_PLACE=0: every batch is counting-sorted
    // room of a bucket = count + sigma * sqrt(count) + add, and `empty` slots for a bucket without a hit in the batch the
    // layout was made from.  Measured at the bench scene (1 M photons, 2^20 buckets; room in deposits / deposits per pass
    // that find their bucket full and take the slower overflow path): 6, 16, 8: 24.1 M / 40; 6, 4, 4: 15.2 M / 900;
    // 5, 4, 4: 14.1 M / 1 000 (no measurable cost); 4, 8, 2 or 6, 4, 2: 14.2 M / 5 200 (gather + 4 %); 4, 4, 1: 11.4 M / 14 000 (+ 9 %).
    float place_sigma = 5.0f; uint32_t place_add = 4;
    uint32_t place_empty = 4;
    float place_far_sigma = 1.0f; uint32_t place_far_add = 1, place_far_empty = 1;   // margins of the buckets no HitPoint reaches when the layout is made; sigma < 0: as the others
    bool place_valid = false;
    uint32_t place_M = 0; uint64_t place_photons = 0, place_total = 0;
    DevBuf<uint32_t> place_room, place_start;
    DevBuf<unsigned long long> place_over_max;   // largest overflow list since the host last looked (one word)
    DevBuf<uint32_t> dstart, scan_partial;
    // Control block, one allocation so that one memset resets a pass: queue scalars (CTL_SCALARS x u64) |
    // gather tile counters (8 x 16 u32, 64 B apart) | {next, end} of the open output chunk of every trace wave slot
    DevBuf<unsigned long long> ctl;
    unsigned long long* scalars() const { return ctl.p; }
    uint32_t* tile_ctr() const { return (uint32_t*)(ctl.p + CTL_SCALARS); }
    unsigned long long* wave_chunks() const { return ctl.p + CTL_SCALARS + 64; }
    size_t ctl_words() const { return CTL_SCALARS + 64 + 2 * (size_t)num_cu * 4 * 4; }
    DevBuf<DevCounters> counters;
    bool accumulate_stats = false;        // counters and timing spans run on over eye passes (sppm_stats_mode)
    bool check_pending = false;           // launches since the last look at the overflow flags
    DevBuf<float> pic;
    float* host_pic = nullptr; size_t host_pic_n = 0;   // pinned staging area of sppm_resolve
    void* host_q = nullptr;               // pinned landing area of read_queues (copies into pageable memory would each block the host)
    uint64_t last_rounds = 0, last_per_round = 0;
    uint64_t batch = 1u << 22;            // photons traced, sorted and gathered together (128 B of deposit buffers per photon x 14)
    // HIP-event spans on the stream, read back (and only then waited for) by sppm_get_timings
    struct Span { int phase; hipEvent_t a, b; };
    std::vector<hipEvent_t> ev_pool;
    size_t ev_used = 0;
    std::vector<Span> spans;
    int gather_launches = 0, trace_launches = 0;
    int num_cu = 256;
    int n_bez = -1;                       // SceneView::n_bez, bez_idx
    unsigned long long bez_idx = 0;
    int n_runs = -1;                      // SceneView::n_runs, runs_lo, runs_hi
    unsigned long long runs_lo = 0, runs_hi = 0;

    SceneParams scene_params() const {
        SceneParams sp;
        sp.blob = blob.p; sp.blob_bytes = blob_bytes; sp.off_box = off_box; sp.off_cp = off_cp; sp.off_pobj = off_pobj; sp.off_aobj = off_aobj;
        sp.n_objs = n_objs; sp.n_patches = n_patches; sp.texels = texels.p;
        sp.pw_q = pw_q.p; sp.pw_c = pw_c.p;
        sp.off_sub = off_sub;
        sp.sub_box = (use_sub_box && n_patches > 0 && off_sub < 0) ? sub_box.p : nullptr;
        sp.n_bez = n_bez; sp.bez_idx = bez_idx;
        sp.n_runs = n_runs; sp.runs_lo = runs_lo; sp.runs_hi = runs_hi;
        return sp;
    }
};

#define HIPCHK(expr)                                                                         \
    do {                                                                                     \
        hipError_t e_ = (expr);                                                              \
        if (e_ != hipSuccess) {                                                              \
            ctx->err = std::string(#expr) + ": " + hipGetErrorString(e_);                    \
            return SPPM_E_DEVICE;                                                            \
        }                                                                                    \
    } while (0)

static uint32_t* hist_words(sppm_ctx* ctx) { return ctx->hist.p; }
static int fail(sppm_ctx* ctx, int code, const std::string& msg) {
    if (ctx) ctx->err = msg;
    return code;
}

// same mixing as rng_base() on the device
static uint32_t host_rng_base(uint64_t seed, uint32_t stream) {
    auto rotl = [](uint32_t x, int r) { return (x << r) | (x >> (32 - r)); };
    auto mm = [&](uint32_t h, uint32_t k) { k *= 0xcc9e2d51u; k = rotl(k, 15); k *= 0x1b873593u; h ^= k; h = rotl(h, 13); return h * 5u + 0xe6546b64u; };
    uint32_t h = 0x9747b28cu;
    h = mm(h, (uint32_t)seed); h = mm(h, (uint32_t)(seed >> 32)); h = mm(h, stream);
    return h;
}

enum { PH_EYE = 0, PH_GRID, PH_TRACE, PH_SORT, PH_GATHER, PH_RESOLVE, PH_COUNT };
static hipEvent_t next_event(sppm_ctx* ctx) {
    if (ctx->ev_used == ctx->ev_pool.size()) {
        hipEvent_t e = nullptr;
        if (hipEventCreate(&e) != hipSuccess) return nullptr;
        ctx->ev_pool.push_back(e);
    }
    return ctx->ev_pool[ctx->ev_used++];
}
// Opens / closes a timed span on the stream.  Nothing waits here: the events are read in sppm_get_timings.
static void tic(sppm_ctx* ctx, int phase) {
    sppm_ctx::Span sp{phase, next_event(ctx), nullptr};
    if (sp.a) (void)hipEventRecord(sp.a, ctx->stream);
    ctx->spans.push_back(sp);
}
static void toc(sppm_ctx* ctx) {
    if (ctx->spans.empty()) return;
    sppm_ctx::Span& sp = ctx->spans.back();
    sp.b = next_event(ctx);
    if (sp.b) (void)hipEventRecord(sp.b, ctx->stream);
}
static void drop_spans(sppm_ctx* ctx) { ctx->spans.clear(); ctx->ev_used = 0; ctx->gather_launches = ctx->trace_launches = 0; }

extern "C" {

int sppm_create(sppm_ctx** out, int device_id) {
    if (!out) return SPPM_E_INVALID;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device_id < 0 || device_id >= ndev) return SPPM_E_DEVICE;
    if (hipSetDevice(device_id) != hipSuccess) return SPPM_E_DEVICE;
    if (device_id < 64) {
        // this GPU's copy of the static scratch array: its address is asked for once per device, with the device current
        std::lock_guard<std::mutex> g(g_scratch_init_mu);
        if (!g_scratch_ptr[device_id]) {
            void* p = nullptr;
            if (hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)) != hipSuccess || !p) return SPPM_E_DEVICE;
            g_scratch_ptr[device_id] = p;
        }
    }
    sppm_ctx* ctx = new (std::nothrow) sppm_ctx();
    if (!ctx) return SPPM_E_NOMEM;
    ctx->device = device_id;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device_id) == hipSuccess) {
        ctx->num_cu = prop.multiProcessorCount;
        // the staging area for the scene tables: 64 KiB, or what a workgroup's LDS leaves beside the photon kernel's own
        // slices (emission, lights, sine table) on a part with less than the 160 KiB of gfx950
        const size_t group = prop.sharedMemPerBlock > 0 ? prop.sharedMemPerBlock : 64 * 1024;
        ctx->scene_lds_max = group > TRACE_PHOTON_LDS_EXTRA ? std::min<size_t>(SCENE_LDS_MAX, group - TRACE_PHOTON_LDS_EXTRA) : 0;
    }
    if (hipStreamCreate(&ctx->own_stream) != hipSuccess) {
        delete ctx;
        return SPPM_E_DEVICE;
    }
    ctx->stream = ctx->own_stream;
    if (const char* b = getenv("SPPM_BATCH")) {
        long long v = atoll(b);
        if (v > 0) ctx->batch = (uint64_t)v;
    }
    if (const char* b = getenv("SPPM_COOP_THRESHOLD")) ctx->coop_threshold = (uint64_t)std::max(0ll, atoll(b));
    if (const char* b = getenv("SPPM_FORCE_COOP")) ctx->force_coop = atoi(b) != 0;
    if (const char* b = getenv("SPPM_EYE_SCATTER")) ctx->eye_scatter = atoi(b) != 0;
    if (const char* b = getenv("SPPM_PHOTON_COOP")) ctx->photon_coop = atoi(b) != 0;
    if (const char* b = getenv("SPPM_GATHER_BPC")) ctx->gather_blocks_per_cu = std::max(0, atoi(b));
    if (const char* b = getenv("SPPM_SCATTER_BPC")) ctx->scatter_blocks_per_cu = std::max(1, atoi(b));
    if (const char* b = getenv("SPPM_TRACE_BPC")) ctx->trace_blocks_per_cu = std::max(0, atoi(b));
    if (const char* b = getenv("SPPM_EYE_BPC")) ctx->eye_blocks_per_cu = std::max(0, atoi(b));
    if (const char* b = getenv("SPPM_EYE_BLOCKS")) ctx->eye_blocks = std::max(0, atoi(b));
    if (const char* b = getenv("SPPM_TRACE_WG")) { const int t = atoi(b); if (t == 64 || t == 128 || t == 256) ctx->trace_wg_threads = t; }
    if (const char* b = getenv("SPPM_NEWTON_BPC")) ctx->newton_blocks_per_cu = std::max(1, atoi(b));
    if (const char* b = getenv("SPPM_DEP_FILTER")) ctx->dep_filter = atoi(b) != 0;
    if (const char* b = getenv("SPPM_PLACE")) ctx->place_enabled = atoi(b) != 0;
    if (const char* b = getenv("SPPM_PLACE_SIGMA")) { ctx->place_sigma = std::max(0.0f, (float)atof(b)); if (ctx->place_sigma == 0.0f) ctx->place_add = 0; }
    if (const char* b = getenv("SPPM_PLACE_ADD")) ctx->place_add = (uint32_t)std::max(0, atoi(b));
    if (const char* b = getenv("SPPM_PLACE_EMPTY")) ctx->place_empty = (uint32_t)std::max(0, atoi(b));
    if (const char* b = getenv("SPPM_PLACE_FAR_SIGMA")) ctx->place_far_sigma = (float)atof(b);
    if (const char* b = getenv("SPPM_PLACE_FAR_ADD")) ctx->place_far_add = (uint32_t)std::max(0, atoi(b));
    if (const char* b = getenv("SPPM_PLACE_FAR_EMPTY")) ctx->place_far_empty = (uint32_t)std::max(0, atoi(b));
    if (const char* b = getenv("SPPM_BRANCH_QUEUE_MIN")) ctx->branch_queue_min = (uint64_t)std::max(1, atoi(b));
    if (const char* b = getenv("SPPM_DEP_PER_PHOTON")) ctx->dep_per_photon = std::min(14.0, std::max(0.01, atof(b)));
    if (const char* b = getenv("SPPM_VERBOSE")) ctx->verbose = atoi(b) != 0;
    if (const char* b = getenv("SPPM_SUBBOX")) ctx->use_sub_box = atoi(b) != 0;
    if (ctx->ctl.reserve(ctx->ctl_words()) != hipSuccess || ctx->counters.reserve(1) != hipSuccess) {
        (void)hipStreamDestroy(ctx->own_stream);
        delete ctx;
        return SPPM_E_DEVICE;
    }
    (void)hipMemset(ctx->ctl.p, 0, ctx->ctl_words() * sizeof(unsigned long long));
    (void)hipMemset(ctx->counters.p, 0, sizeof(DevCounters));
    {
        // sin and cos of 2 PI j / 256 in the host's extended precision, rounded once to double; PI is the reference's
        // seven-digit constant (Vec3.h:7), so a full turn of phi is not quite a full circle and no entry is an exact zero
        double tab[2 * SPPM_SINCOS_TAB];
        const long double two_pi = (long double)(2.0 * SPPM_PI);
        for (int j = 0; j < SPPM_SINCOS_TAB; j++) {
            const long double a = two_pi * (long double)j / (long double)SPPM_SINCOS_TAB;
            tab[2 * j] = (double)sinl(a);
            tab[2 * j + 1] = (double)cosl(a);
        }
        if (ctx->sincos_tab.reserve(2 * SPPM_SINCOS_TAB) != hipSuccess ||
            hipMemcpy(ctx->sincos_tab.p, tab, sizeof(tab), hipMemcpyHostToDevice) != hipSuccess) {
            ctx->ctl.release(); ctx->counters.release(); ctx->sincos_tab.release();
            (void)hipStreamDestroy(ctx->own_stream);
            delete ctx;
            return SPPM_E_DEVICE;
        }
    }
    if (device_id < 64) g_live_contexts[device_id].fetch_add(1);
    *out = ctx;
    return SPPM_OK;
}

void sppm_destroy(sppm_ctx* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    ctx->pw_q.release(); ctx->pw_c.release(); ctx->sub_box.release();
    ctx->blob.release(); ctx->texels.release(); ctx->lights.release(); ctx->sincos_tab.release();
    ctx->raw.release(); ctx->park[0].release(); ctx->park[1].release(); ctx->ntask[0].release(); ctx->ntask[1].release(); ctx->nres.release();
    ctx->hist.release(); ctx->hstart.release();
    ctx->hp_rec.release(); ctx->hp_tao.release(); ctx->hp_meta.release(); ctx->hp_r2.release(); ctx->sorted_of_raw.release();
    ctx->dep.release(); ctx->sdep.release(); ctx->pieces.release();
    ctx->place_room.release(); ctx->place_start.release(); ctx->place_over_max.release();
    ctx->dstart.release(); ctx->scan_partial.release(); ctx->ctl.release(); ctx->counters.release(); ctx->pic.release();
    if (ctx->host_pic) (void)hipHostFree(ctx->host_pic);
    if (ctx->host_q) (void)hipHostFree(ctx->host_q);
    for (hipEvent_t e : ctx->ev_pool) (void)hipEventDestroy(e);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    if (ctx->device < 64) g_live_contexts[ctx->device].fetch_sub(1);
    delete ctx;
}

const char* sppm_last_error(const sppm_ctx* ctx) { return ctx ? ctx->err.c_str() : "null context"; }

static bool any_zero3(const double* v) { return std::fabs(v[0]) < 1e-4 || std::fabs(v[1]) < 1e-4 || std::fabs(v[2]) < 1e-4; }

int sppm_set_scene(sppm_ctx* ctx, const sppm_scene* s) {
    if (!ctx || !s) return SPPM_E_INVALID;
    if (s->n_objects <= 0 || !s->objects) return fail(ctx, SPPM_E_INVALID, "scene has no objects");
    if (s->n_lights <= 0 || !s->lights) return fail(ctx, SPPM_E_INVALID, "scene has no lights");
    if (s->n_patches < 0 || (s->n_patches > 0 && !s->patches)) return fail(ctx, SPPM_E_INVALID, "bad patch array");
    if (s->n_textures < 0 || (s->n_textures > 0 && !s->textures)) return fail(ctx, SPPM_E_INVALID, "bad texture array");
    if (s->n_patches > (1 << 16)) return fail(ctx, SPPM_E_CAPACITY, "more than 65536 Bezier patches in one scene");
    if (s->n_objects > (1 << 12)) return fail(ctx, SPPM_E_CAPACITY, "more than 4096 objects in one scene");
    HIPCHK(hipSetDevice(ctx->device));
    HIPCHK(hipStreamSynchronize(ctx->stream));   // passes still queued read the tables replaced below
    ctx->have_scene = false;                     // an upload that fails half way leaves no scene rather than a mixed one
    ctx->have_eye = false;
    const int no = s->n_objects, np = s->n_patches;
    // Boxes of the 4 x 4 sub-patches of every patch (subbox_prune): the control points of the sub-patch over
    // [iu/4, (iu+1)/4] x [iv/4, (iv+1)/4] are blossom values of the patch (de Casteljau with one parameter per level), their
    // box is grown by the margin and rounded outwards to fp32.
    std::vector<float> sb;
    if (ctx->use_sub_box && np > 0) {
        auto blossom = [](const double c[4], double t1, double t2, double t3) {
            const double a0 = c[0] + (c[1] - c[0]) * t1, a1 = c[1] + (c[2] - c[1]) * t1, a2 = c[2] + (c[3] - c[2]) * t1;
            const double b0 = a0 + (a1 - a0) * t2, b1 = a1 + (a2 - a1) * t2;
            return b0 + (b1 - b0) * t3;
        };
        auto segment = [&](const double c[4], double a, double b, double out[4]) {
            out[0] = blossom(c, a, a, a); out[1] = blossom(c, a, a, b); out[2] = blossom(c, a, b, b); out[3] = blossom(c, b, b, b);
        };
        sb.assign((size_t)np * 16 * 6, 0.0f);
        for (int p = 0; p < np; p++)
            for (int iv = 0; iv < 4; iv++)
                for (int iu = 0; iu < 4; iu++) {
                    double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
                    for (int k = 0; k < 3; k++) {
                        double R[4][4];                                   // rows cut to the u segment: R[j][.] for row j (v index)
                        for (int j = 0; j < 4; j++) {
                            double c[4];
                            for (int i = 0; i < 4; i++) c[i] = s->patches[(size_t)p * 48 + (j * 4 + i) * 3 + k];   // P[4 j + i], i along u
                            segment(c, iu / 4.0, (iu + 1) / 4.0, R[j]);
                        }
                        for (int i = 0; i < 4; i++) {                     // columns cut to the v segment
                            const double c[4] = {R[0][i], R[1][i], R[2][i], R[3][i]};
                            double q[4];
                            segment(c, iv / 4.0, (iv + 1) / 4.0, q);
                            for (int j = 0; j < 4; j++) { lo[k] = std::min(lo[k], q[j]); hi[k] = std::max(hi[k], q[j]); }
                        }
                    }
                    float* o = sb.data() + ((size_t)p * 16 + (size_t)(iv * 4 + iu)) * 6;
                    for (int k = 0; k < 3; k++) {
                        o[k] = std::nextafterf((float)(lo[k] - SPPM_SUBBOX_MARGIN), -INFINITY);
                        o[3 + k] = std::nextafterf((float)(hi[k] + SPPM_SUBBOX_MARGIN), INFINITY);
                    }
                }
    }
    const int off_box = (int)(((size_t)no * sizeof(DevObject) + 15) & ~(size_t)15);
    const int off_cp = off_box + np * 6 * (int)sizeof(double);
    const int off_pobj = off_cp + np * 48 * (int)sizeof(float);
    const int off_aobj = (off_pobj + np * (int)sizeof(int) + 15) & ~15;
    int bytes = (off_aobj + (no + 1) * (int)sizeof(AObj) + 15) & ~15;   // one spare record: the plane / sphere loop looks one ahead
    // the sub-patch boxes ride in the blob (and so in LDS) where that keeps it within the staging area
    int off_sub = -1;
    if (!sb.empty() && (size_t)bytes + sb.size() * sizeof(float) <= ctx->scene_lds_max) { off_sub = bytes; bytes += (int)(sb.size() * sizeof(float)); }
    std::vector<char> blob((size_t)bytes, 0);
    if (off_sub >= 0) memcpy(blob.data() + off_sub, sb.data(), sb.size() * sizeof(float));
    else if (!sb.empty()) {
        HIPCHK(ctx->sub_box.reserve(sb.size()));
        HIPCHK(hipMemcpy(ctx->sub_box.p, sb.data(), sb.size() * sizeof(float), hipMemcpyHostToDevice));
    }
    DevObject* objs = (DevObject*)blob.data();
    double* box = (double*)(blob.data() + off_box);
    float* cp = (float*)(blob.data() + off_cp);
    int* pobj = (int*)(blob.data() + off_pobj);
    for (int p = 0; p < np; p++) pobj[p] = -1;
    for (int p = 0; p < np; p++) {                                                       // Bezier3::read, Bezier.h:167-175
        double* b = box + 6 * p;
        for (int k = 0; k < 3; k++) { b[k] = 1e20; b[3 + k] = -1e20; }
        for (int k = 0; k < 16; k++)
            for (int c = 0; c < 3; c++) {
                double v = s->patches[(size_t)p * 48 + k * 3 + c];
                cp[(size_t)p * 48 + k * 3 + c] = (float)v;
                if (b[c] > v) b[c] = v;
                if (b[3 + c] < v) b[3 + c] = v;
            }
    }
    // textures -> packed uchar4
    std::vector<TexDesc> td((size_t)std::max(1, s->n_textures));
    size_t total = 0;
    for (int t = 0; t < s->n_textures; t++) {
        if (s->textures[t].rows <= 0 || s->textures[t].cols <= 0 || !s->textures[t].texels)
            return fail(ctx, SPPM_E_INVALID, "empty texture");
        td[t].rows = s->textures[t].rows; td[t].cols = s->textures[t].cols; td[t].offset = (unsigned)total; td[t].pad = 0;
        total += (size_t)td[t].rows * td[t].cols;
        if (total > (1ull << 30)) return fail(ctx, SPPM_E_CAPACITY, "more than 2^30 texels in one scene");
    }
    bool dual_lobe = false, hp_then_on = false;
    for (int i = 0; i < no; i++) {
        const sppm_object& so = s->objects[i];
        DevObject& o = objs[i];
        if (so.type < SPPM_OBJ_PLANE || so.type > SPPM_OBJ_BEZIER) return fail(ctx, SPPM_E_INVALID, "unknown object type");
        if (so.texture >= s->n_textures) return fail(ctx, SPPM_E_INVALID, "texture index out of range");
        o.type = so.type; o.texture = so.texture < 0 ? -1 : so.texture;
        for (int k = 0; k < 3; k++) {
            o.p0[k] = so.p0[k]; o.n[k] = so.n[k];
            o.color[k] = (float)so.color[k];
            o.diff[k] = (float)so.material.diff[k]; o.refl[k] = (float)so.material.refl[k]; o.refr[k] = (float)so.material.refr[k];
            o.bmin[k] = 1e20; o.bmax[k] = -1e20;
        }
        o.radius = so.radius; o.refrn = so.material.refrn;
        o.inv_radius = so.radius != 0 ? 1.0 / so.radius : 0.0;
        o.inv_refrn = so.material.refrn != 0 ? 1.0 / so.material.refrn : 0.0;
        o.diffr = (so.material.diff[0] + so.material.diff[1] + so.material.diff[2]) / 3;  // Obj.h:11-16, Vec3.h:116-119
        o.reflr = (so.material.refl[0] + so.material.refl[1] + so.material.refl[2]) / 3;
        o.refrr = (so.material.refr[0] + so.material.refr[1] + so.material.refr[2]) / 3;
        o.allr = o.diffr + o.reflr + o.refrr;
        {   // Obj::Roulette (Obj.h:30-45) draws r in [0, allr): with one positive power the outcome does not depend on r
            const int lobes = (o.diffr > 0) + (o.reflr > 0) + (o.refrr > 0);
            o.lobe = 0;
            if (lobes == 1) o.lobe = o.diffr > 0 ? F_DIFF : (o.reflr > 0 ? F_REFL : F_REFR);
        }
        // Material::IsDiff etc.: !IsZero(vec), where IsZero(vec) is true if ANY component is ~0 (Element.h:16-18, Vec3.h:72-79)
        o.flags = (any_zero3(so.material.diff) ? 0 : F_DIFF) | (any_zero3(so.material.refl) ? 0 : F_REFL) |
                  (any_zero3(so.material.refr) ? 0 : F_REFR);
        o.first_patch = so.first_patch; o.n_patches = so.n_patches;
        o.akind = o.type; o.pad0 = 0;
        if (o.type == OBJ_PLANE)   // normal exactly +-1 along one axis: the one-term form of the plane test
            for (int k = 0; k < 3; k++)
                if (std::fabs(o.n[k]) == 1.0 && o.n[(k + 1) % 3] == 0.0 && o.n[(k + 2) % 3] == 0.0) o.akind = A_PLANE_X + k;
        {   // PlaneObj::GetColor, Obj.h:86-100
            int ndir = 0;
            for (int k = 0; k < 3; k++) if (o.n[k] != 0) ndir = k;
            o.tex_vdex = (ndir + 2) % 3; o.tex_udex = (ndir + 1) % 3;
        }
        o.tex_rows = o.tex_cols = 0; o.tex_off = 0;
        if (o.texture >= 0) { o.tex_rows = td[o.texture].rows; o.tex_cols = td[o.texture].cols; o.tex_off = td[o.texture].offset; }
        if ((o.flags & F_REFL) && (o.flags & F_REFR)) dual_lobe = true;
        if ((o.flags & F_DIFF) && (o.flags & (F_REFL | F_REFR))) hp_then_on = true;
        if (so.type == SPPM_OBJ_BEZIER) {
            if (so.first_patch < 0 || so.n_patches < 0 || so.first_patch + so.n_patches > np)
                return fail(ctx, SPPM_E_INVALID, "bezier object indexes patches out of range");
            for (int p = so.first_patch; p < so.first_patch + so.n_patches; p++) {         // Bezier.h:226-231
                if (pobj[p] >= 0) return fail(ctx, SPPM_E_INVALID, "two bezier objects share a patch");
                pobj[p] = i;
                for (int k = 0; k < 3; k++) {
                    o.bmin[k] = std::min(o.bmin[k], box[6 * p + k]);
                    o.bmax[k] = std::max(o.bmax[k], box[6 * p + 3 + k]);
                }
            }
        }
    }
    {
        AObj* ao = (AObj*)(blob.data() + off_aobj);     // what the plane / sphere tests read (analytic_nearest)
        for (int i = 0; i < no; i++) {
            ao[i].a = ao[i].b = ao[i].c = ao[i].r = 0;
            if (objs[i].akind >= A_PLANE_X) ao[i].a = objs[i].p0[objs[i].akind - A_PLANE_X];
            else if (objs[i].type == OBJ_SPHERE) { ao[i].a = objs[i].p0[0]; ao[i].b = objs[i].p0[1]; ao[i].c = objs[i].p0[2]; ao[i].r = objs[i].radius; }
        }
    }
    // Newton tables: Bernstein -> power basis around the patch centre, in fp64 (patch_eval_pw)
    {
        static const double Mb[4][4] = {{1, 0, 0, 0}, {-3, 3, 0, 0}, {3, -6, 3, 0}, {-1, 3, -3, 1}};   // u^i = sum_a Mb[i][a] P_a
        std::vector<float> q((size_t)std::max(1, np) * 48, 0.0f);
        std::vector<double> cen((size_t)std::max(1, np) * 3, 0.0);
        for (int p = 0; p < np; p++) {
            double c[3];
            for (int k = 0; k < 3; k++) c[k] = 0.5 * (box[6 * p + k] + box[6 * p + 3 + k]);
            for (int j = 0; j < 4; j++)
                for (int i = 0; i < 4; i++)
                    for (int k = 0; k < 3; k++) {
                        double acc = 0;
                        for (int b2 = 0; b2 < 4; b2++)
                            for (int a2 = 0; a2 < 4; a2++)
                                acc += Mb[j][b2] * Mb[i][a2] * (s->patches[(size_t)p * 48 + (b2 * 4 + a2) * 3 + k] - c[k]);
                        q[(size_t)p * 48 + (j * 4 + i) * 3 + k] = (float)acc;
                    }
            for (int k = 0; k < 3; k++) cen[(size_t)p * 3 + k] = c[k];
        }
        HIPCHK(ctx->pw_q.reserve(q.size()));
        HIPCHK(hipMemcpy(ctx->pw_q.p, q.data(), q.size() * sizeof(float), hipMemcpyHostToDevice));
        HIPCHK(ctx->pw_c.reserve(cen.size()));
        HIPCHK(hipMemcpy(ctx->pw_c.p, cen.data(), cen.size() * sizeof(double), hipMemcpyHostToDevice));
    }
    HIPCHK(ctx->blob.reserve((size_t)bytes));
    HIPCHK(hipMemcpy(ctx->blob.p, blob.data(), (size_t)bytes, hipMemcpyHostToDevice));
    {   // the Bezier objects by index, for the loops over them (SceneView::bez_idx)
        int nb = 0;
        unsigned long long packed = 0;
        for (int i = 0; i < no && nb >= 0; i++)
            if (objs[i].type == OBJ_BEZIER) {
                if (nb == 8 || i > 255) { nb = -1; break; }
                packed |= (unsigned long long)i << (8 * nb++);
            }
        ctx->n_bez = nb; ctx->bez_idx = nb >= 0 ? packed : 0ull;
        // the objects in index order as runs of one kind (SceneView::n_runs)
        int nr = 0;
        unsigned long long rl[2] = {0, 0};
        for (int i = 0; i < no && nr >= 0;) {
            int len = 1;
            while (i + len < no && len < 15 && objs[i + len].akind == objs[i].akind) len++;
            if (nr == SPPM_MAX_RUNS) { nr = -1; break; }
            rl[nr >> 3] |= (unsigned long long)((objs[i].akind & 15) | (len << 4)) << (8 * (nr & 7));
            nr++; i += len;
        }
        ctx->n_runs = nr; ctx->runs_lo = nr >= 0 ? rl[0] : 0ull; ctx->runs_hi = nr >= 0 ? rl[1] : 0ull;
    }
    ctx->blob_bytes = bytes; ctx->off_box = off_box; ctx->off_cp = off_cp; ctx->off_pobj = off_pobj; ctx->off_aobj = off_aobj; ctx->off_sub = off_sub; ctx->n_objs = no; ctx->n_patches = np;
    // Large scenes: tables beyond the LDS staging area are read from global memory, and with more patches than the patch
    // mask of a parked ray holds every Bezier solve happens in the tracing wave (the default anyway).
    ctx->scene_in_lds = (size_t)bytes <= ctx->scene_lds_max;
    ctx->coop_only = np > 64 * SPPM_MASK_WORDS || !ctx->scene_in_lds;
    std::vector<uchar4> tx(std::max<size_t>(1, total));
    for (int t = 0; t < s->n_textures; t++) {
        const uint8_t* src = s->textures[t].texels;
        uchar4* dst = tx.data() + td[t].offset;
        for (size_t k = 0; k < (size_t)td[t].rows * td[t].cols; k++) dst[k] = make_uchar4(src[3 * k], src[3 * k + 1], src[3 * k + 2], 0);
    }
    HIPCHK(ctx->texels.reserve(tx.size()));
    HIPCHK(hipMemcpy(ctx->texels.p, tx.data(), tx.size() * sizeof(uchar4), hipMemcpyHostToDevice));
    std::vector<DevLight> dl((size_t)s->n_lights);
    for (int l = 0; l < s->n_lights; l++)
        for (int k = 0; k < 3; k++) {
            dl[l].pos[k] = s->lights[l].pos[k];
            dl[l].flux[k] = (float)(s->lights[l].color[k] * SPPM_PI * 4);                  // Light.h:10
        }
    HIPCHK(ctx->lights.reserve(dl.size()));
    HIPCHK(hipMemcpy(ctx->lights.p, dl.data(), dl.size() * sizeof(DevLight), hipMemcpyHostToDevice));
    ctx->n_lights = s->n_lights;
    ctx->has_dual_lobe = dual_lobe;
    ctx->one_hp_per_pixel = !dual_lobe && !hp_then_on;
    ctx->place_valid = false;
    ctx->have_scene = true;
    ctx->have_eye = false;
    return SPPM_OK;
}

static int launch_fill(sppm_ctx* ctx, uint32_t* p, size_t n, uint32_t v) {
    if (n == 0) return SPPM_OK;
    int blocks = (int)std::min<size_t>((n + 255) / 256, 4096);
    hipLaunchKernelGGL(k_fill_u32, dim3(blocks), dim3(256), 0, ctx->stream, p, n, v);
    HIPCHK(hipGetLastError());
    return SPPM_OK;
}

// The control block, optionally the tallies, and a range of histogram words, zeroed by one launch at the start of a pass.
static int launch_pass_reset(sppm_ctx* ctx, bool tallies, uint32_t* hist, size_t hist_words) {
    const size_t ctl_w = ctx->ctl_words() * 2;                        // 64-bit scalars
    const size_t tally_w = tallies ? offsetof(DevCounters, overflow) / 4 : 0;
    const size_t most = std::max(std::max(ctl_w, tally_w), hist_words);
    const int blocks = (int)std::min<size_t>((most + 255) / 256, 4096);
    hipLaunchKernelGGL(k_zero3, dim3(blocks), dim3(256), 0, ctx->stream, (uint32_t*)ctx->ctl.p, ctl_w, (uint32_t*)ctx->counters.p, tally_w,
                       hist, hist_words);
    HIPCHK(hipGetLastError());
    return SPPM_OK;
}

// start[0..n] = exclusive prefix sums of cnt[0..n-1]; n <= SCAN_TILE^2.
static int trace_max_blocks(const sppm_ctx* ctx);
// close_dep / close_hits: also turn the unused tails of the open chunks of that array into holes (see close_chunks).
static int launch_scan(sppm_ctx* ctx, const uint32_t* cnt, uint32_t* start, uint32_t n, Deposit* close_dep = nullptr, RawHit* close_hits = nullptr,
                       uint64_t close_cap = 0) {
    const uint32_t nb = (n + SCAN_TILE - 1) / SCAN_TILE;
    if (nb > SCAN_TILE) return fail(ctx, SPPM_E_INVALID, "scan too large");
    HIPCHK(ctx->scan_partial.reserve(SCAN_TILE + 1));
    if (close_dep || close_hits) {
        const uint32_t n_slots = (uint32_t)trace_max_blocks(ctx) * 4u;
        hipLaunchKernelGGL(k_scan_reduce_close, dim3(nb + (n_slots + 3) / 4), dim3(SCAN_THREADS), 0, ctx->stream, cnt, n, ctx->scan_partial.p, nb,
                           ctx->wave_chunks(), n_slots, close_dep, close_hits, (unsigned long long)close_cap);
    } else
    hipLaunchKernelGGL(k_scan_reduce, dim3(nb), dim3(SCAN_THREADS), 0, ctx->stream, cnt, n, ctx->scan_partial.p);
    hipLaunchKernelGGL(k_scan_final, dim3(nb), dim3(SCAN_THREADS), 0, ctx->stream, cnt, n, ctx->scan_partial.p, nb, start);
    HIPCHK(hipGetLastError());
    return SPPM_OK;
}

}  // extern "C" (templates below need C++ linkage)

// queue scalars of the control block
enum { SC_HITS = 0, SC_DEP = 1, SC_NEXT0 = 2 /* .. SC_NEXT0 + 3 */, SC_PARK0 = 6, SC_TASK0 = 7, SC_PARK1 = 8, SC_TASK1 = 9, SC_REAL_HITS = 10,
       SC_PIECES = 11, SC_NEXT_PIECE = 12, SC_OVER = 13 };

// Smallest m >= want (>= 1) with gcd(m, n) = 1.
static uint64_t coprime_near(uint64_t n, uint64_t want) {
    auto gcd = [](uint64_t a, uint64_t b) { while (b) { uint64_t t = a % b; a = b; b = t; } return a; };
    uint64_t m = std::max<uint64_t>(1, want);
    while (gcd(m, n) != 1) m++;
    return m;
}

static int trace_max_blocks(const sppm_ctx* ctx) { return ctx->num_cu * 4; }
// output slots that can be left unused in open chunks (one chunk per wave slot of the largest trace grid)
// items: work items of the largest launch (a grid never has more workgroups than items / 256)
static uint64_t dep_slack(const sppm_ctx* ctx, uint64_t items = ~0ull) {
    return std::min<uint64_t>((uint64_t)trace_max_blocks(ctx), (items + 255) / 256 + 1) * 4u * TRACE_SLOT_CHUNK;
}

static int reset_ctl(sppm_ctx* ctx) {
    HIPCHK(hipMemsetAsync(ctx->ctl.p, 0, ctx->ctl_words() * sizeof(unsigned long long), ctx->stream));
    return SPPM_OK;
}
// Turns the unused tails of the open chunks into holes once the trace launches of a batch / pass are done.
static int close_chunks(sppm_ctx* ctx, Deposit* dep, RawHit* hits, uint64_t cap) {
    const uint32_t n_slots = (uint32_t)trace_max_blocks(ctx) * 4u;
    hipLaunchKernelGGL(k_close_chunks, dim3((n_slots + 3) / 4), dim3(256), 0, ctx->stream, ctx->wave_chunks(), n_slots, dep, hits,
                       (unsigned long long)cap);
    HIPCHK(hipGetLastError());
    return SPPM_OK;
}

template <int MODE>
static void launch_trace(sppm_ctx* ctx, bool coop, int blocks, const SceneParams& sp, const TraceParams& tp) {
    // dynamic LDS: the scene tables, then (photon pass) 4 waves x 128 emission directions, the first lights and the sine table
    // A trace workgroup is four waves unless SPPM_TRACE_WG says otherwise (64 or 128 threads: a workgroup's registers and LDS
    // are given back when its LAST wave is done, so smaller workgroups free a SIMD's slot for another context's kernel sooner
    // at the end of a launch; the grid has as many waves either way).
    const int threads = ctx->trace_wg_threads;
    blocks *= 256 / threads;
    const size_t lds = (ctx->scene_in_lds ? (((size_t)ctx->blob_bytes + 15) & ~(size_t)15) : 0) + (MODE == MODE_PHOTON ? TRACE_PHOTON_LDS(threads / 64) : 0);
    if (!ctx->scene_in_lds) hipLaunchKernelGGL((k_trace<MODE, true, false>), dim3(blocks), dim3(threads), lds, ctx->stream, sp, tp);   // coop_only
    else if (coop && sp.n_patches == 0 && MODE == MODE_PHOTON) hipLaunchKernelGGL(k_trace_photon_plain, dim3(blocks), dim3(threads), lds, ctx->stream, sp, tp);
    else if (coop && sp.n_patches == 0) hipLaunchKernelGGL((k_trace<MODE, true, true, false>), dim3(blocks), dim3(threads), lds, ctx->stream, sp, tp);
    else if (coop && MODE == MODE_PHOTON) hipLaunchKernelGGL(k_trace_photon_bez, dim3(blocks), dim3(threads), lds, ctx->stream, sp, tp);
    else if (coop) hipLaunchKernelGGL(k_trace_eye_bez, dim3(blocks), dim3(threads), lds, ctx->stream, sp, tp);
    else hipLaunchKernelGGL((k_trace<MODE, false, true>), dim3(blocks), dim3(threads), lds, ctx->stream, sp, tp);
}
static int launch_trace_mode(sppm_ctx* ctx, int mode, bool coop, int blocks, const TraceParams& tp) {
    const SceneParams sp = ctx->scene_params();
    if (mode == MODE_EYE) launch_trace<MODE_EYE>(ctx, coop, blocks, sp, tp);
    else launch_trace<MODE_PHOTON>(ctx, coop, blocks, sp, tp);
    HIPCHK(hipGetLastError());
    ctx->trace_launches++;
    ctx->check_pending = true;
    return SPPM_OK;
}

// What the host reads back when it has to look at the queues.
struct QueueState { unsigned long long sc[CTL_SCALARS]; DevCounters dc; };
struct QueueLanding { QueueState q; unsigned long long over_max; };
// The queue scalars, the tallies and the overflow high-water mark, written straight into the host's pinned landing area
// by one small kernel (three copies of their own were three launches on the stream the host is waiting for).
__global__ void k_queue_snapshot(const unsigned long long* __restrict__ scalars, const DevCounters* __restrict__ dc,
                                 const unsigned long long* __restrict__ over_max, QueueLanding* out) {
    const uint32_t i = threadIdx.x;
    if (i < CTL_SCALARS) out->q.sc[i] = scalars[i];
    const uint32_t* src = (const uint32_t*)dc;
    uint32_t* dst = (uint32_t*)&out->q.dc;
    for (uint32_t k = i; k < sizeof(DevCounters) / 4; k += blockDim.x) dst[k] = src[k];
    if (i == 0) out->over_max = over_max ? *over_max : 0ull;
}
static int read_queues(sppm_ctx* ctx, QueueState& q) {
    typedef QueueLanding Landing;
    if (!ctx->host_q) HIPCHK(hipHostMalloc(&ctx->host_q, sizeof(Landing), hipHostMallocDefault));
    Landing* L = (Landing*)ctx->host_q;
    const bool look = ctx->place_valid && ctx->place_over_max.p;
    hipLaunchKernelGGL(k_queue_snapshot, dim3(1), dim3(64), 0, ctx->stream, ctx->scalars(), ctx->counters.p,
                       look ? ctx->place_over_max.p : (const unsigned long long*)nullptr, L);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(ctx->stream));
    q = L->q;
    const unsigned long long over_max = L->over_max;
    ctx->check_pending = false;
    // a deposit layout that sent many deposits to the overflow list no longer fits what is traced (another scene
    // behind the same grid, other materials): the next batch is counting-sorted again and lays out a new one
    if (look && over_max > 4096 && over_max > ctx->place_total / 64) {
        ctx->place_valid = false;
        HIPCHK(hipMemsetAsync(ctx->place_over_max.p, 0, sizeof(unsigned long long), ctx->stream));
    }
    return SPPM_OK;
}
#define SPPM_RETRY_HITS (SPPM_E_CAPACITY + 100)
#define SPPM_RETRY_PARK (SPPM_E_CAPACITY + 101)
#define SPPM_RETRY_DEP (SPPM_E_CAPACITY + 102)
// Overflow bits: 1 HitPoint buffer, 2 ray / task queue, 4 deposit buffer or overflow list, 8 gather piece queue, 16 a HitPoint
// placed beyond the sorted arrays (k_hp_scatter: never expected).  A pass whose
// host is still there to run it again (the eye pass; a counting-sorted photon batch) retries with more room, told which by
// the RETRY codes; everything else is an error.
static int overflow_code(sppm_ctx* ctx, const DevCounters& dc, bool can_retry_queues = false) {
    if (dc.overflow == 0) return SPPM_OK;
    (void)hipMemsetAsync(&ctx->counters.p->overflow, 0, sizeof(unsigned int), ctx->stream);   // seen: clear
    if (can_retry_queues && (dc.overflow & ~7u) == 0) {
        if (dc.overflow & 4u) return SPPM_RETRY_DEP;
        if (dc.overflow & 2u) return SPPM_RETRY_PARK;
    }
    if (dc.overflow & 4u) {   // a queued pass ran out of deposit slots: its result is lost, the next pass gets the worst-case room
        ctx->dep_per_photon = SPPM_MAX_DEP + 1;
        ctx->place_valid = false;
    }
    if (dc.overflow & 16u) return fail(ctx, SPPM_E_DEVICE, "the HitPoint tally and the HitPoint histogram of an eye pass disagree");
    if (dc.overflow & 14u) return fail(ctx, SPPM_E_CAPACITY, "a ray / task / deposit / gather-piece queue overflowed");
    return SPPM_RETRY_HITS;   // HitPoint buffer too small: the eye pass retries with more room
}

// Records the ray queues of run_trace must hold.  With in-wave solves nothing is parked for a solve: only the refraction
// branches of eye paths through a material that both reflects and refracts are queued (as fresh rays), at most one per
// path vertex, and their queue grows by doubling when a pass overflows it (park_scale).  Parked solves (the non-default
// modes) can queue every work item.  Seed rays (n_seed) are records of the queue themselves.
static uint64_t trace_park_cap(const sppm_ctx* ctx, int mode, bool first_coop, bool has_dual_lobe, uint64_t fresh_count, uint64_t n_seed) {
    const bool branches = mode == MODE_EYE && has_dual_lobe;
    return std::max<uint64_t>(n_seed, !first_coop ? std::max<uint64_t>(fresh_count, 1u << 16)
                                      : branches  ? std::max<uint64_t>(std::max(fresh_count, n_seed) / 4, ctx->branch_queue_min) * ctx->park_scale
                                                  : (1u << 12));
}

// Traces one set of fresh work items (pixels or photons) to completion.  The control block must have
// been reset.  The launches are queued without looking at their results wherever the schedule is known
// in advance:
//   in-wave Bezier solves (photons by default): one launch, nothing can be parked;
//   parked solves (eye rays by default): trace -> k_newton -> resume launch with in-wave solves, the
//     counts travelling in device memory.  Only a material with both a reflective and a refractive lobe
//     can leave rays queued after that (the refraction branches of Raytracer.h:327-336); then, and in
//     the non-default modes, the host reads the queue counters and keeps launching until they are empty.
// *synced is set when the host has waited for the stream (q then holds the final queue state).
// n_seed > 0: no fresh items; the launch starts from n_seed ray records the caller has put into queue 1 (ctx->park[1],
// count in the control block), traced with in-wave solves: rays of the caller's own making (sppm_view_trace, sppm_photon_trace).
static int run_trace(sppm_ctx* ctx, int mode, TraceParams tp, uint64_t fresh_first, uint64_t fresh_count, bool has_dual_lobe,
                     bool need_final_state, QueueState& q, bool* synced, uint64_t n_seed = 0) {
    *synced = false;
    // Photons meet a Bezier object at random, a few lanes per wave, and so do eye rays once the pixels are
    // dealt to the waves in a scattered order (TraceParams::pix_mul): solving in place costs less than the
    // extra launches of a park round.  With pixels in row order the rays that meet the object are whole
    // waves, which would serialise 64 solves each while the other waves idle: then they are parked and
    // solved by k_newton (SPPM_EYE_SCATTER=0, or SPPM_PHOTON_COOP=0 for photons).
    const bool first_coop = n_seed > 0 || ctx->force_coop || ctx->coop_only || (mode == MODE_PHOTON && ctx->photon_coop) || (mode == MODE_EYE && ctx->eye_scatter);
    // The ray and task queues are sized by what can reach them (trace_park_cap; the eye pass retries with a doubled
    // park_scale like it does for the HitPoint buffer); parked solves bring up to 16 patch tasks per ray.
    const uint64_t park_cap = trace_park_cap(ctx, mode, first_coop, has_dual_lobe, fresh_count, n_seed);
    const uint64_t task_cap = first_coop ? 16 : 16 * park_cap;
    if (task_cap > 0xffffffffull) return fail(ctx, SPPM_E_CAPACITY, "too many work items for one round");
    // Seed rays already sit in queue 1 (upload_seed_rays sized it with trace_park_cap too): growing the queue here would
    // free them, since DevBuf::reserve does not keep a buffer's contents.
    if (n_seed > 0 && ctx->park[1].n < park_cap) return fail(ctx, SPPM_E_DEVICE, "seed queue smaller than the trace rounds need");
    const ParkRec* const seeds_at = ctx->park[1].p;
    HIPCHK(ctx->park[0].reserve(park_cap)); HIPCHK(ctx->park[1].reserve(park_cap));
    if (n_seed > 0 && ctx->park[1].p != seeds_at) return fail(ctx, SPPM_E_DEVICE, "seed queue moved after the upload");
    HIPCHK(ctx->ntask[0].reserve(task_cap)); HIPCHK(ctx->ntask[1].reserve(task_cap));
    HIPCHK(ctx->nres.reserve(task_cap));
    unsigned long long* sc = ctx->scalars();
    const int max_blocks = trace_max_blocks(ctx);
    // persistent workgroups: as many as stay resident (2 per CU with in-wave solves, 3 without), not more; one per CU
    // when other contexts share the GPU
    auto blocks_for = [&](uint64_t items, bool coop) {
        const int per_cu = (mode == MODE_EYE && ctx->eye_blocks_per_cu > 0) ? ctx->eye_blocks_per_cu
                           : ctx->trace_blocks_per_cu > 0 ? ctx->trace_blocks_per_cu
                           : shares_gpu(ctx->device)    ? 1
                                                        : (coop ? (mode == MODE_EYE ? TRACE_EYE_OCC : TRACE_COOP_OCC) : 3);
        const uint64_t cap = std::min<uint64_t>((uint64_t)max_blocks, (uint64_t)ctx->num_cu * per_cu);
        return (int)std::max<uint64_t>(1, std::min<uint64_t>(cap, (items + 255) / 256));
    };
    auto park_ctr = [&](int i) { return sc + (i ? SC_PARK1 : SC_PARK0); };
    auto task_ctr = [&](int i) { return sc + (i ? SC_TASK1 : SC_TASK0); };
    int next_slot = 0;   // work counters SC_NEXT0.. are zero after the reset; later launches clear theirs first
    auto set_round = [&](int cur, bool fresh) -> int {   // cur: queue the launch reads (ignored when fresh); writes the other one
        const int out = fresh ? 0 : (cur ^ 1);
        if (next_slot < 4) tp.next = sc + SC_NEXT0 + next_slot++;
        else { tp.next = sc + SC_NEXT0; HIPCHK(hipMemsetAsync(tp.next, 0, sizeof(unsigned long long), ctx->stream)); }
        if (fresh) {
            tp.resume = nullptr; tp.n_resume_dev = nullptr; tp.tasks_in = nullptr; tp.res_in = nullptr;
            tp.fresh_first = fresh_first; tp.fresh_count = fresh_count;
        } else {
            tp.resume = ctx->park[cur].p; tp.n_resume_dev = park_ctr(cur); tp.tasks_in = ctx->ntask[cur].p; tp.res_in = ctx->nres.p;
            tp.fresh_first = 0; tp.fresh_count = 0;
        }
        tp.park_out = ctx->park[out].p; tp.n_park = park_ctr(out); tp.tasks_out = ctx->ntask[out].p; tp.n_tasks = task_ctr(out);
        return SPPM_OK;
    };
    auto launch_newton = [&](int cur) -> int {
        hipLaunchKernelGGL(k_newton, dim3(ctx->num_cu * ctx->newton_blocks_per_cu), dim3(256), 0, ctx->stream, ctx->park[cur].p,
                           park_ctr(cur), (unsigned long long)park_cap, ctx->ntask[cur].p, (unsigned long long)task_cap,
                           (const double*)(ctx->blob.p + ctx->off_box), ctx->pw_q.p, ctx->pw_c.p, ctx->nres.p, ctx->counters.p);
        HIPCHK(hipGetLastError());
        return SPPM_OK;
    };
    tp.park_cap = park_cap; tp.task_cap = task_cap;
    const bool legacy = (mode == MODE_PHOTON && !first_coop);   // photons with parked solves: every round looks at the counters
    if (int rc = set_round(n_seed > 0 ? 1 : 0, n_seed == 0)) return rc;       // seeds: read queue 1, write queue 0
    int first_blocks = blocks_for(n_seed > 0 ? n_seed : fresh_count, first_coop);
    if (mode == MODE_EYE && ctx->eye_blocks > 0) first_blocks = std::max(1, std::min(first_blocks, ctx->eye_blocks));
    if (int rc = launch_trace_mode(ctx, mode, first_coop, first_blocks, tp)) return rc;
    int cur = 0;   // queue written by the last launch
    if (!legacy) {
        if (!first_coop) {
            if (int rc = launch_newton(0)) return rc;
            if (int rc = set_round(0, false)) return rc;
            // the count is unknown here: a generous grid, of which the workgroups beyond the work leave at once
            if (int rc = launch_trace_mode(ctx, mode, true, std::max(1, blocks_for(fresh_count, true) / 2), tp)) return rc;
            cur = 1;
        }
        const bool may_queue = (mode == MODE_EYE && has_dual_lobe);
        if (!may_queue && !need_final_state) return SPPM_OK;
    }
    for (int round = 0; round < 256; round++) {
        if (int rc = read_queues(ctx, q)) return rc;
        *synced = true;
        if (int rc = overflow_code(ctx, q.dc, first_coop || mode == MODE_PHOTON)) return rc;    // photons: the caller retries a full deposit buffer
        const unsigned long long n_park = q.sc[cur ? SC_PARK1 : SC_PARK0], n_tasks = q.sc[cur ? SC_TASK1 : SC_TASK0];
        if (ctx->verbose) fprintf(stderr, "[sppm] mode %d round %d: %llu rays queued, %llu newton tasks\n", mode, round, n_park, n_tasks);
        if (n_park == 0) return SPPM_OK;
        if (n_tasks > 0) if (int rc = launch_newton(cur)) return rc;
        const int nxt = cur ^ 1;
        HIPCHK(hipMemsetAsync(park_ctr(nxt), 0, sizeof(unsigned long long), ctx->stream));
        HIPCHK(hipMemsetAsync(task_ctr(nxt), 0, sizeof(unsigned long long), ctx->stream));
        if (int rc = set_round(cur, false)) return rc;
        const bool coop = !legacy || ctx->force_coop || ctx->coop_only || n_park < ctx->coop_threshold;
        if (int rc = launch_trace_mode(ctx, mode, coop, blocks_for(n_park, coop), tp)) return rc;
        cur = nxt;
    }
    return fail(ctx, SPPM_E_DEVICE, "trace rounds did not drain");
}

extern "C" {

// The HitPoints' view direction, tree node, object and N, which only a host that asks for the HitPoints and the radius
// update read: written from the raw array of the last eye pass the first time they are wanted (and before anything else
// reuses that array).
static int ensure_meta(sppm_ctx* ctx) {
    if (ctx->meta_valid) return SPPM_OK;
    if (ctx->raw_claimed > 0) {
        hipLaunchKernelGGL(k_hp_meta, dim3((unsigned)((ctx->raw_claimed + 255) / 256)), dim3(256), 0, ctx->stream, ctx->raw.p, (uint32_t)ctx->raw_claimed,
                           ctx->sorted_of_raw.p, ctx->hp_meta.p, ctx->initial_n);
        HIPCHK(hipGetLastError());
    }
    ctx->meta_valid = true;
    return SPPM_OK;
}

int sppm_eye_pass(sppm_ctx* ctx, const sppm_camera* cam, uint64_t seed, uint32_t replica) {
    if (!ctx || !cam) return SPPM_E_INVALID;
    if (!ctx->have_scene) return fail(ctx, SPPM_E_INVALID, "sppm_set_scene must come first");
    if (cam->w <= 0 || cam->h <= 0 || (uint64_t)cam->w * cam->h > (1ull << 28)) return fail(ctx, SPPM_E_INVALID, "bad resolution");
    HIPCHK(hipSetDevice(ctx->device));
    ctx->cam = *cam;
    ctx->have_eye = false;
    const uint64_t npx = (uint64_t)cam->w * cam->h;
    // raw HitPoint slots: what the pixels can leave if that is known (the reference's scene: one each), else a guess that
    // the retry below doubles
    uint64_t hit_cap = std::max<uint64_t>(ctx->one_hp_per_pixel ? npx : 2 * npx, 1 << 16) + dep_slack(ctx, npx);
    if (!ctx->accumulate_stats) drop_spans(ctx);
    // ---- the hashed grid the HitPoints will be sorted into.  The eye kernel histograms them as it writes them, so the
    // bucket count is chosen BEFORE the trace, from the HitPoint count of the last eye pass at this resolution (a
    // few per cent off at worst: the count only changes with the camera jitter) or else from the pixel count.  The
    // bucket count of the last pass is kept while it still suits (1.5 to 4 buckets per HitPoint), so the deposit
    // layout made for it stays usable when the count wanders across a power of two.  A count that suits badly costs
    // hash collisions, not results.
    const double cs = std::sqrt(SPPM_INIT_R2) * 1.0001;   // cell >= search radius: neighbours within +-1 cell
    ctx->grid.inv_cs = 1.0 / cs;
    {
        const bool same = ctx->grid_w == cam->w && ctx->grid_h == cam->h && ctx->n_hp > 0;
        const uint64_t n_est = same ? ctx->n_hp : npx;
        const uint32_t M_prev = ctx->grid.mask + 1;
        uint32_t M = std::min<uint32_t>(1u << 22, std::max<uint32_t>(1u << 16, next_pow2(2 * std::max<uint64_t>(n_est, 1))));
        if (same && M_prev >= (1u << 16) && 2 * (uint64_t)M_prev >= 3 * n_est && (uint64_t)M_prev <= 4 * n_est) M = M_prev;
        ctx->grid.mask = M - 1;
        ctx->grid_w = cam->w; ctx->grid_h = cam->h;
    }
    const uint32_t M = ctx->grid.mask + 1;
    HIPCHK(ctx->hist.reserve(ctx->occ_words() + 4 * ((size_t)M + 1)));
    HIPCHK(ctx->hstart.reserve(M + 1)); HIPCHK(ctx->dstart.reserve(M + 1));
    QueueState q{};
    for (int attempt = 0;; attempt++) {
        HIPCHK(ctx->raw.reserve(hit_cap));
        // the tallies restart; the overflow flags stay until the host has looked at them
        if (int rc = launch_pass_reset(ctx, !ctx->accumulate_stats, hist_words(ctx), ctx->occ_words() + 2 * ((size_t)M + 1))) return rc;   // + occupancy, hcount, cursor
        tic(ctx, PH_EYE);
        TraceParams tp{};
        tp.hcount = ctx->hcount(); tp.grid = ctx->grid;
        for (int k = 0; k < 3; k++) { tp.pos[k] = cam->pos[k]; tp.dir[k] = cam->dir[k]; tp.du[k] = cam->du[k]; tp.dv[k] = cam->dv[k]; }
        tp.w = cam->w; tp.h = cam->h;
        tp.pix_mul = ctx->eye_scatter ? coprime_near(npx, (uint64_t)((double)npx * 0.6180339887)) : 1;
        tp.npx_magic = npx > 1 ? (unsigned long long)((((unsigned __int128)1) << 64) / npx) : ~0ull;
        tp.w_magic = cam->w > 1 ? (uint32_t)((1ull << 32) / (uint64_t)cam->w) : 0xffffffffu;
        tp.hits = ctx->raw.p; tp.n_hits = ctx->scalars() + SC_HITS; tp.hit_cap = hit_cap;
        tp.n_real_hits = ctx->scalars() + SC_REAL_HITS;
        tp.counters = ctx->counters.p;
        tp.wave_chunks = ctx->wave_chunks();
        tp.rng = host_rng_base(seed, replica * 4u + D_EYE);
        bool synced = false;
        int rc = run_trace(ctx, MODE_EYE, tp, 0, npx, ctx->has_dual_lobe, true, q, &synced);
        toc(ctx);
        if (rc == SPPM_OK) break;
        if (rc != SPPM_RETRY_HITS && rc != SPPM_RETRY_PARK) return rc < 0 ? rc : fail(ctx, SPPM_E_CAPACITY, "a queue the eye pass does not use reported an overflow");
        if (attempt == 7) return fail(ctx, SPPM_E_CAPACITY, "HitPoint buffer or branch queue overflowed after 7 capacity doublings");
        if (rc == SPPM_RETRY_HITS) hit_cap *= 2; else ctx->park_scale *= 2;
        if (!ctx->accumulate_stats) drop_spans(ctx);
    }
    // q is the state after the last launch: claimed slots of the raw array (with holes) and the HitPoints in them
    const unsigned long long claimed = std::min<unsigned long long>(q.sc[SC_HITS], hit_cap);
    ctx->n_hp = q.sc[SC_REAL_HITS];
    // ---- counting sort of the HitPoints by bucket (the histogram is the eye kernel's)
    const uint64_t n = ctx->n_hp;
    tic(ctx, PH_GRID);
    HIPCHK(ctx->hp_rec.reserve_with_room(n)); HIPCHK(ctx->hp_tao.reserve_with_room(n));
    HIPCHK(ctx->hp_meta.reserve_with_room(n)); HIPCHK(ctx->hp_r2.reserve_with_room(n));
    if (n > 0) {
        const int blocks = (int)((claimed + 255) / 256);
        if (int rc = launch_scan(ctx, ctx->hcount(), ctx->hstart.p, M, nullptr, ctx->raw.p, hit_cap)) return rc;   // + chunk tails -> holes
        HIPCHK(ctx->sorted_of_raw.reserve(std::max<uint64_t>(hit_cap, 1)));   // as many as the raw array holds: the claimed count wanders with the jitter, and growing a buffer waits for the GPU
        hipLaunchKernelGGL(k_hp_scatter, dim3(blocks), dim3(256), 0, ctx->stream, ctx->raw.p, (uint32_t)claimed, ctx->grid,
                           ctx->hstart.p, ctx->cursor(), ctx->hp_rec.p, ctx->sorted_of_raw.p, (uint32_t)n, &ctx->counters.p->overflow);
        hipLaunchKernelGGL(k_hp_occupancy, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, ctx->hp_rec.p, (uint32_t)n, ctx->grid, ctx->occ(),
                           ctx->hp_r2.p, ctx->hp_tao.p);
        HIPCHK(hipGetLastError());
    }
    ctx->raw_claimed = n > 0 ? claimed : 0;
    ctx->meta_valid = false;
    if (ctx->initial_n > 0)                      // the radius update keeps N in the record
        if (int rc = ensure_meta(ctx)) return rc;
    toc(ctx);
    ctx->have_eye = true;
    ctx->radii_changed = false;
    ctx->last_rounds = 0; ctx->last_per_round = 0;
    return SPPM_OK;
}

static void fill_photon_params(sppm_ctx* ctx, TraceParams& tp, uint32_t per_round, uint64_t dep_cap, uint64_t seed, uint32_t replica,
                               DevCounters* counters) {
    tp.lights = ctx->lights.p; tp.n_lights = ctx->n_lights; tp.per_round = per_round;
    tp.sincos_tab = ctx->sincos_tab.p;
    tp.dep = ctx->dep.p; tp.n_dep = ctx->scalars() + SC_DEP; tp.dep_cap = dep_cap;
    tp.dcount = ctx->dcount(); tp.grid = ctx->grid; tp.counters = counters;
    tp.occ = ctx->dep_filter ? ctx->occ() : nullptr;
    tp.place = nullptr; tp.sorted = nullptr; tp.over = nullptr; tp.n_over = nullptr; tp.over_cap = 0; tp.dcount_all = nullptr;
    tp.wave_chunks = ctx->wave_chunks();
    tp.rng = host_rng_base(seed, replica * 4u + D_PHOTON);
}

// Deferred look at the overflow flags of the launches queued since the last look.
static int check_deferred(sppm_ctx* ctx) {
    if (!ctx->check_pending) return SPPM_OK;
    DevCounters dc;
    HIPCHK(hipMemcpyAsync(&dc, ctx->counters.p, sizeof(dc), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    ctx->check_pending = false;
    if (overflow_code(ctx, dc) != SPPM_OK) return fail(ctx, SPPM_E_CAPACITY, "a ray / task / deposit queue overflowed in an earlier pass");
    return SPPM_OK;
}

int sppm_stats_mode(sppm_ctx* ctx, int accumulate) {
    if (!ctx) return SPPM_E_INVALID;
    HIPCHK(hipSetDevice(ctx->device));
    ctx->accumulate_stats = accumulate != 0;
    drop_spans(ctx);
    HIPCHK(hipMemsetAsync(ctx->counters.p, 0, offsetof(DevCounters, overflow), ctx->stream));
    return SPPM_OK;
}

int sppm_set_stream(sppm_ctx* ctx, void* hip_stream) {
    if (!ctx) return SPPM_E_INVALID;
    HIPCHK(hipSetDevice(ctx->device));
    HIPCHK(hipStreamSynchronize(ctx->stream));     // nothing of this context is left behind on the old stream
    ctx->stream = hip_stream ? (hipStream_t)hip_stream : ctx->own_stream;
    return SPPM_OK;
}

int sppm_set_radius_update(sppm_ctx* ctx, int32_t initial_n, uint32_t rounds_per_update) {
    if (!ctx) return SPPM_E_INVALID;
    if (initial_n < 0 || (initial_n > 0 && rounds_per_update == 0)) return fail(ctx, SPPM_E_INVALID, "initial_n >= 0 and rounds_per_update >= 1");
    if (ctx->have_eye && initial_n != ctx->initial_n) {   // the HitPoints of the last eye pass keep the N they were made with
        HIPCHK(hipSetDevice(ctx->device));
        if (int rc = ensure_meta(ctx)) return rc;
    }
    ctx->initial_n = initial_n;
    ctx->rounds_per_update = std::max<uint32_t>(1, rounds_per_update);
    return SPPM_OK;
}

void* sppm_get_stream(const sppm_ctx* ctx) { return ctx ? (void*)ctx->stream : nullptr; }

int sppm_synchronize(sppm_ctx* ctx) {
    if (!ctx) return SPPM_E_INVALID;
    HIPCHK(hipSetDevice(ctx->device));
    if (ctx->check_pending) return check_deferred(ctx);
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return SPPM_OK;
}

// k_gather + k_gather_split over the grid-sorted deposits [dstart[b], + min(dlen[b], dstart[b + 1] - dstart[b])); `over`: the deposits
// that found their bucket full under direct placement.  fixed_r2: every radius is still INIT_R2.
static int launch_gather(sppm_ctx* ctx, const uint32_t* dstart, const uint32_t* dlen, const Deposit* over, uint64_t dep_cap, bool fixed_r2) {
    tic(ctx, PH_GATHER);
    const uint64_t tiles = (ctx->n_hp + GATHER_TILE - 1) / GATHER_TILE;
    // beside a trace wave of 208 (photons) or 256 (eye) registers a SIMD has room for three gather waves, beside one of 160 (no Bezier objects) for four
    const int gather_bpc = ctx->gather_blocks_per_cu > 0 ? ctx->gather_blocks_per_cu : (shares_gpu(ctx->device) ? (ctx->n_patches == 0 ? 4 : 3) : 5);
    const int blocks = (int)std::max<uint64_t>(8, std::min<uint64_t>((uint64_t)ctx->num_cu * gather_bpc, ((tiles + 3) / 4 + 7) / 8 * 8));
    const uint32_t piece_cap = 1u << 20;
    HIPCHK(ctx->pieces.reserve(piece_cap));
#define SPPM_LAUNCH_GATHER(FIXED, SHARE)                                                                                                                        \
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_gather<FIXED>), dim3(blocks), dim3(256), 0, ctx->stream, ctx->hp_rec.p, ctx->hp_r2.p,                          \
                       ctx->hp_tao.p, (uint32_t)ctx->n_hp, dstart, dlen, ctx->sdep.p, ctx->grid, ctx->tile_ctr(), ctx->pieces.p,                        \
                       ctx->scalars() + SC_PIECES, piece_cap, ctx->counters.p);                                                                          \
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_gather_split<FIXED, SHARE>), dim3(blocks), dim3(256), 0, ctx->stream, ctx->hp_rec.p, ctx->hp_r2.p, \
                       ctx->hp_tao.p, (uint32_t)ctx->n_hp, dstart, dlen, ctx->sdep.p, ctx->grid, ctx->pieces.p,                                         \
                       ctx->scalars() + SC_PIECES, piece_cap, ctx->scalars() + SC_NEXT_PIECE, ctx->counters.p, over, ctx->scalars() + SC_OVER,           \
                       (unsigned long long)dep_cap, ctx->place_over_max.p, ctx->hstart.p)
    const bool share = shares_gpu(ctx->device);
    if (fixed_r2) { if (share) { SPPM_LAUNCH_GATHER(true, true); } else { SPPM_LAUNCH_GATHER(true, false); } }
    else          { if (share) { SPPM_LAUNCH_GATHER(false, true); } else { SPPM_LAUNCH_GATHER(false, false); } }
#undef SPPM_LAUNCH_GATHER
    HIPCHK(hipGetLastError());
    toc(ctx);
    ctx->gather_launches++;
    return SPPM_OK;
}

int sppm_photon_pass(sppm_ctx* ctx, uint64_t rounds, uint32_t per_round, uint64_t photon_begin, uint64_t photon_count,
                     uint64_t seed, uint32_t replica) {
    if (!ctx) return SPPM_E_INVALID;
    if (!ctx->have_eye) return fail(ctx, SPPM_E_INVALID, "sppm_eye_pass must come first");
    if (rounds == 0 || per_round == 0) return fail(ctx, SPPM_E_INVALID, "rounds and per_round must be positive");
    HIPCHK(hipSetDevice(ctx->device));
    if (rounds > 0xffffffffull) return fail(ctx, SPPM_E_INVALID, "more than 2^32 photons in one pass");
    const uint64_t total = rounds * (uint64_t)ctx->n_lights * per_round;   // < 2^32 * 2^31 * 2^32: the product may wrap, the factors are checked first
    if (total / per_round / (uint64_t)ctx->n_lights != rounds || total > 0xffffffffull) return fail(ctx, SPPM_E_INVALID, "more than 2^32 photons in one pass");
    if (photon_begin > total) return fail(ctx, SPPM_E_INVALID, "photon_begin beyond the pass");
    const uint64_t end = photon_count == 0 ? total : std::min(total, photon_begin + photon_count);
    ctx->last_rounds = rounds; ctx->last_per_round = per_round;
    const uint32_t M = ctx->grid.mask + 1;
    uint64_t batch = std::max<uint64_t>(1, std::min<uint64_t>(ctx->batch, end > photon_begin ? end - photon_begin : 1));
    const bool sweeps = ctx->initial_n > 0;      // HitPoint::update() after every rounds_per_update rounds (Raytracer.h:237-245)
    if (sweeps) {
        if (photon_begin != 0 || end != total)
            return fail(ctx, SPPM_E_INVALID, "with the radius update on, a photon pass cannot be sharded: every update needs all photons of its rounds");
        batch = (uint64_t)ctx->rounds_per_update * ctx->n_lights * per_round;
        if (batch > (1ull << 24)) return fail(ctx, SPPM_E_INVALID, "more than 2^24 photons between two radius updates");
    }
    // Deposit slots: dep_per_photon per photon of a batch, not the 14 a path could leave in a closed all-diffuse room (the
    // reference's open tunnel leaves 3.7 diffuse hits per photon, of which 1.9 are stored).  A counting-sorted batch that
    // overflows them is traced again with more (below); a placed batch only uses this buffer as its overflow list.
    auto cap_for = [&](double per_photon) { return (uint64_t)std::ceil((double)batch * std::min<double>(per_photon, SPPM_MAX_DEP + 1)) + dep_slack(ctx, batch); };
    // With the radius update on every batch is counting-sorted and batches are a few rounds: they get the worst-case room
    // while that stays under 256 MiB, so that no batch has to be looked at before the next one is queued.
    const bool worst_case_room = sweeps && batch * (SPPM_MAX_DEP + 1) * sizeof(Deposit) <= (256ull << 20);
    uint64_t dep_cap = cap_for(worst_case_room ? SPPM_MAX_DEP + 1 : ctx->dep_per_photon);
    // The buffer of deposits as traced is reserved batch by batch: a counting-sorted batch needs dep_cap slots of it, a
    // placed one only an overflow list.
    const uint64_t over_room = std::max<uint64_t>(1u << 16, batch / 4);
    HIPCHK(ctx->sdep.reserve(std::max<uint64_t>(dep_cap, ctx->place_valid ? ctx->place_total : 0)));
    HIPCHK(ctx->place_over_max.reserve(1));
    TraceParams tp{};
    fill_photon_params(ctx, tp, per_round, dep_cap, seed, replica, ctx->counters.p);
    QueueState q{};
    for (uint64_t first = photon_begin; first < end; first += batch) {
        const uint64_t count = std::min(batch, end - first);
        // Direct placement needs a layout made from a batch at least as large, on the same grid, and deposits
        // filtered by reach (a bucket no HitPoint reaches has no room).
        const bool placed = ctx->place_enabled && ctx->dep_filter && !sweeps && ctx->place_valid && ctx->place_M == M &&
                            count <= ctx->place_photons && ctx->n_hp > 0;
        bool lay_out = false;     // this batch is counting-sorted and leaves a layout behind
        const uint32_t* dstart = placed ? ctx->place_start.p : ctx->dstart.p;
        const uint32_t* dlen = ctx->dcount();      // placed: the buckets' fill counts; counting-sorted: the histogram the ranges were made from
        uint64_t over_cap = 0;
        if (placed) {
            if (int rc = launch_pass_reset(ctx, false, ctx->dcount(), (size_t)M + 1)) return rc;
            tp.place = ctx->place_start.p; tp.sorted = ctx->sdep.p; tp.dcount_all = nullptr;
            // the overflow list lives in the buffer the counting sort reads (all of it, if that is still there)
            HIPCHK(ctx->dep.reserve(over_room));
            over_cap = ctx->dep.n;
            tp.over = ctx->dep.p; tp.n_over = ctx->scalars() + SC_OVER; tp.over_cap = over_cap;
            tic(ctx, PH_TRACE);
            bool synced = false;
            if (int rc = run_trace(ctx, MODE_PHOTON, tp, first, count, false, false, q, &synced)) {
                if (rc < 0) return rc;
                ctx->place_valid = false;          // parked-solve rounds saw the overflow list full: the next batch lays the room out anew
                return fail(ctx, SPPM_E_CAPACITY, "the overflow list of a placed photon batch is full");
            }
            toc(ctx);
        } else {
            HIPCHK(ctx->dep.reserve(dep_cap));
            fill_photon_params(ctx, tp, per_round, dep_cap, seed, replica, ctx->counters.p);
            lay_out = ctx->place_enabled && ctx->dep_filter && !sweeps && ctx->n_hp > 0 && count == batch;
            tp.dcount_all = lay_out ? ctx->dcount_all() : nullptr;
            DevCounters first_failed{}, before{};
            // Parked-solve rounds (SPPM_PHOTON_COOP=0) stop at the round that sees the buffer full, so the failed attempt's
            // tallies are partial: that mode, which waits for the stream every round anyway, notes the tallies before the batch.
            const bool rounds_on_host = !(ctx->force_coop || ctx->coop_only || ctx->photon_coop);
            if (rounds_on_host) { if (int rc = read_queues(ctx, q)) return rc; before = q.dc; }
            bool tallies_restored = false;
            for (int attempt = 0;; attempt++) {
                if (int rc = launch_pass_reset(ctx, false, ctx->cursor(), (lay_out ? 3 : 2) * ((size_t)M + 1))) return rc;   // + cursor, dcount (and dcount_all)
                tic(ctx, PH_TRACE);
                bool synced = false;
                const int trc = run_trace(ctx, MODE_PHOTON, tp, first, count, false, false, q, &synced);
                toc(ctx);
                if (trc < 0) return trc;
                // A counting-sorted batch is where the deposit buffer can overflow with the host still at hand to trace it
                // again: one look at the flags before the deposits are sorted and gathered (placed batches, the steady
                // state, have no such wait; nor has a batch whose buffer holds the 14 deposits a path can leave — every
                // batch of a pass with the radius update on — which goes on asynchronously like them).
                int oc = trc;              // parked-solve rounds look at the flags themselves and hand a full buffer back here
                bool partial = trc != SPPM_OK;
                if (trc == SPPM_OK) {
                    if (attempt == 0 && dep_cap >= cap_for(SPPM_MAX_DEP + 1)) break;
                    if (int rc = read_queues(ctx, q)) return rc;
                    oc = overflow_code(ctx, q.dc, true);
                }
                if (oc == SPPM_OK) {
                    if (attempt > 0 && !tallies_restored) {
                        // the photons of the failed attempts were tallied as well: back to the tallies after the first of them
                        // (the stream is idle: read_queues waited for it)
                        DevCounters fix = q.dc;
                        fix.photons = first_failed.photons; fix.segments = first_failed.segments; fix.deposits = first_failed.deposits;
                        fix.stored = first_failed.stored; fix.newton_calls = first_failed.newton_calls;
                        HIPCHK(hipMemcpy(ctx->counters.p, &fix, offsetof(DevCounters, overflow), hipMemcpyHostToDevice));
                    }
                    break;
                }
                if (oc != SPPM_RETRY_DEP || attempt == 3) return oc < 0 ? oc : fail(ctx, SPPM_E_CAPACITY, "a ray / deposit queue overflowed");
                if (attempt == 0) first_failed = q.dc;
                if (partial) {             // the stream is idle: the rounds waited for it
                    HIPCHK(hipMemcpy(ctx->counters.p, &before, offsetof(DevCounters, overflow), hipMemcpyHostToDevice));
                    tallies_restored = true;
                }
                // the slot counter ran on past the capacity: it says what the batch needs
                const double need = partial ? SPPM_MAX_DEP + 1 : (double)q.sc[SC_DEP] / (double)count;      // rounds cut short say nothing
                ctx->dep_per_photon = std::min<double>(SPPM_MAX_DEP + 1, std::max(2.0 * ctx->dep_per_photon, need * 1.25));
                dep_cap = cap_for(ctx->dep_per_photon);
                if (ctx->verbose) fprintf(stderr, "[sppm] deposit buffer overflowed (%.2f slots per photon needed): retrying with %.2f\n", need, ctx->dep_per_photon);
                HIPCHK(ctx->dep.reserve(dep_cap)); HIPCHK(ctx->sdep.reserve(std::max<uint64_t>(dep_cap, ctx->place_valid ? ctx->place_total : 0)));
                fill_photon_params(ctx, tp, per_round, dep_cap, seed, replica, ctx->counters.p);
                tp.dcount_all = lay_out ? ctx->dcount_all() : nullptr;
            }
            tic(ctx, PH_SORT);
            if (int rc = launch_scan(ctx, ctx->dcount(), ctx->dstart.p, M, ctx->dep.p, nullptr, dep_cap)) return rc;   // + chunk tails -> holes
            hipLaunchKernelGGL(k_dep_scatter, dim3(ctx->num_cu * ctx->scatter_blocks_per_cu), dim3(256), 0, ctx->stream, ctx->dep.p, ctx->scalars() + SC_DEP,
                               dep_cap, ctx->dstart.p, ctx->cursor(), ctx->sdep.p);
            HIPCHK(hipGetLastError());
            toc(ctx);
        }
        if (ctx->n_hp > 0) {
            // the deposits that found their bucket full (normally a handful) ride along with the pieces
            if (int rc = launch_gather(ctx, dstart, dlen, placed ? (const Deposit*)ctx->dep.p : (const Deposit*)nullptr, placed ? over_cap : dep_cap,
                                       !sweeps && !ctx->radii_changed)) return rc;
            if (sweeps) {
                hipLaunchKernelGGL(k_hp_update, dim3((unsigned)((ctx->n_hp + 255) / 256)), dim3(256), 0, ctx->stream, ctx->hp_r2.p, ctx->hp_tao.p,
                                   ctx->hp_meta.p, (uint32_t)ctx->n_hp);
                ctx->radii_changed = true;
                HIPCHK(hipGetLastError());
            }
        }
        // A counting-sorted batch of full size leaves its histogram behind: lay out the room for the batches to come
        // (one wait for the stream, for the total; later batches and passes of this size are placed directly).
        if (lay_out) {
            HIPCHK(ctx->place_room.reserve((size_t)M + 1)); HIPCHK(ctx->place_start.reserve((size_t)M + 1));
            // First with the full margin for every bucket; if that room is more than the buffer of sorted deposits holds
            // anyway (it is as large as the buffer of deposits as traced), again with the small margins for the buckets out
            // of the HitPoints' reach: the tighter layout sends more deposits to the overflow list and is only worth its memory.
            uint32_t total_room = 0;
            HIPCHK(hipMemsetAsync(ctx->place_over_max.p, 0, sizeof(unsigned long long), ctx->stream));
            for (int tight = 0; tight < 2; tight++) {
                hipLaunchKernelGGL(k_bucket_room, dim3((M + 255) / 256), dim3(256), 0, ctx->stream, ctx->dcount_all(),
                                   tight ? (const uint8_t*)ctx->occ() : (const uint8_t*)nullptr, M, ctx->place_sigma, ctx->place_add,
                                   ctx->place_add ? ctx->place_empty : 0u, ctx->place_far_sigma, ctx->place_far_add, ctx->place_far_empty, ctx->place_room.p);
                HIPCHK(hipGetLastError());
                if (int rc = launch_scan(ctx, ctx->place_room.p, ctx->place_start.p, M)) return rc;
                HIPCHK(hipMemcpyAsync(&total_room, ctx->place_start.p + M, sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
                HIPCHK(hipStreamSynchronize(ctx->stream));
                const bool may_tighten = ctx->place_far_sigma >= 0.0f && ctx->place_sigma > 0.0f;
                if (tight || !may_tighten || total_room <= std::max<uint64_t>(dep_cap, ctx->sdep.n)) break;
            }
            ctx->place_total = total_room; ctx->place_M = M; ctx->place_photons = count; ctx->place_valid = true;
            HIPCHK(ctx->sdep.reserve(std::max<uint64_t>(dep_cap, ctx->place_total)));
            tp.sorted = ctx->sdep.p;
            // the batches to come are placed: the counting sort's input buffer goes back (the stream is idle), an overflow
            // list takes its place; a batch that has to be counting-sorted again reserves it anew
            if (ctx->dep.n > over_room) ctx->dep.release();
        }
    }
    return SPPM_OK;   // queued; a queue overflow, if any, is reported by the next call that waits for the stream
}

int sppm_resolve_device(sppm_ctx* ctx, float* dev_pic) {
    if (!ctx || !dev_pic) return SPPM_E_INVALID;
    if (!ctx->have_eye) return fail(ctx, SPPM_E_INVALID, "sppm_eye_pass must come first");
    if (ctx->last_rounds == 0) return fail(ctx, SPPM_E_INVALID, "sppm_photon_pass must come first");
    HIPCHK(hipSetDevice(ctx->device));
    const size_t npx = (size_t)ctx->cam.w * ctx->cam.h;
    tic(ctx, PH_RESOLVE);
    HIPCHK(hipMemsetAsync(dev_pic, 0, npx * 3 * sizeof(float), ctx->stream));
    if (ctx->n_hp > 0) {
        const double inv_norm = 1.0 / SPPM_PI / ((double)ctx->last_rounds * (double)ctx->last_per_round);
        const int blocks = (int)((ctx->n_hp + 255) / 256);
        if (ctx->one_hp_per_pixel)
            hipLaunchKernelGGL(k_resolve<true>, dim3(blocks), dim3(256), 0, ctx->stream, ctx->hp_r2.p, ctx->hp_rec.p, ctx->hp_tao.p, (uint32_t)ctx->n_hp,
                               inv_norm, dev_pic);
        else
            hipLaunchKernelGGL(k_resolve<false>, dim3(blocks), dim3(256), 0, ctx->stream, ctx->hp_r2.p, ctx->hp_rec.p, ctx->hp_tao.p, (uint32_t)ctx->n_hp,
                               inv_norm, dev_pic);
        HIPCHK(hipGetLastError());
    }
    toc(ctx);
    return SPPM_OK;
}

int sppm_resolve(sppm_ctx* ctx, double* pic, int accumulate) {
    if (!ctx || !pic) return SPPM_E_INVALID;
    if (!ctx->have_eye) return fail(ctx, SPPM_E_INVALID, "sppm_eye_pass must come first");
    const size_t n = (size_t)ctx->cam.w * ctx->cam.h * 3;
    HIPCHK(hipSetDevice(ctx->device));
    HIPCHK(ctx->pic.reserve(n));
    if (int rc = sppm_resolve_device(ctx, ctx->pic.p)) return rc;
    if (ctx->host_pic_n < n) {
        if (ctx->host_pic) (void)hipHostFree(ctx->host_pic);
        ctx->host_pic = nullptr; ctx->host_pic_n = 0;
        HIPCHK(hipHostMalloc((void**)&ctx->host_pic, n * sizeof(float), hipHostMallocDefault));
        ctx->host_pic_n = n;
    }
    const float* h = ctx->host_pic;
    HIPCHK(hipMemcpyAsync(ctx->host_pic, ctx->pic.p, n * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    if (int rc = check_deferred(ctx)) return rc;
    if (accumulate) for (size_t i = 0; i < n; i++) pic[i] += (double)h[i];
    else for (size_t i = 0; i < n; i++) pic[i] = (double)h[i];
    return SPPM_OK;
}

int sppm_get_hitpoints(sppm_ctx* ctx, sppm_hitpoint* out, uint64_t cap, uint64_t* n) {
    if (!ctx || !n) return SPPM_E_INVALID;
    if (!ctx->have_eye) return fail(ctx, SPPM_E_INVALID, "sppm_eye_pass must come first");
    *n = ctx->n_hp;
    const uint64_t m = std::min<uint64_t>(cap, ctx->n_hp);
    if (m == 0 || !out) return SPPM_OK;
    HIPCHK(hipSetDevice(ctx->device));
    if (int rc = ensure_meta(ctx)) return rc;
    HIPCHK(hipStreamSynchronize(ctx->stream));   // the copies below go through the null stream, which a caller's stream need not wait for
    std::vector<HpRec> rec(m); std::vector<float4> tao(m); std::vector<HpMeta> meta(m); std::vector<double> r2(m);
    HIPCHK(hipMemcpy(r2.data(), ctx->hp_r2.p, m * sizeof(double), hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(rec.data(), ctx->hp_rec.p, m * sizeof(HpRec), hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(tao.data(), ctx->hp_tao.p, m * sizeof(float4), hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(meta.data(), ctx->hp_meta.p, m * sizeof(HpMeta), hipMemcpyDeviceToHost));
    for (uint64_t i = 0; i < m; i++) {
        sppm_hitpoint& h = out[i];
        const PosKey& pk = rec[i].a; const float4& nr_i = rec[i].nr; const float4& wg_i = rec[i].wgt;
        h.pos[0] = pk.x; h.pos[1] = pk.y; h.pos[2] = pk.z;
        h.n[0] = nr_i.x; h.n[1] = nr_i.y; h.n[2] = nr_i.z; h.R2 = r2[i];
        h.v[0] = meta[i].v[0]; h.v[1] = meta[i].v[1]; h.v[2] = meta[i].v[2];
        h.wgt[0] = wg_i.x; h.wgt[1] = wg_i.y; h.wgt[2] = wg_i.z;
        h.tao[0] = tao[i].x; h.tao[1] = tao[i].y; h.tao[2] = tao[i].z;
        int pix, newN;
        memcpy(&pix, &wg_i.w, 4); memcpy(&newN, &tao[i].w, 4);
        h.x = pix / ctx->cam.w; h.y = pix % ctx->cam.w;
        h.obj = meta[i].obj; h.node = meta[i].node; h.N = meta[i].N; h.newN = newN;
    }
    return SPPM_OK;
}

int sppm_get_counters(sppm_ctx* ctx, sppm_counters* out) {
    if (!ctx || !out) return SPPM_E_INVALID;
    HIPCHK(hipSetDevice(ctx->device));
    DevCounters dc;
    HIPCHK(hipMemcpyAsync(&dc, ctx->counters.p, sizeof(dc), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    out->photons = dc.photons; out->segments = dc.segments; out->deposits = dc.deposits;
    out->candidates = dc.candidates; out->accepted = dc.accepted; out->hitpoints = ctx->n_hp;
    out->newton_calls = dc.newton_calls; out->eye_segments = dc.eye_segments;
    if (ctx->verbose) {
        unsigned long long np = 0;
        (void)hipMemcpy(&np, ctx->scalars() + SC_PIECES, sizeof(np), hipMemcpyDeviceToHost);
        unsigned long long nov = 0, omax = 0;
        (void)hipMemcpy(&nov, ctx->scalars() + SC_OVER, sizeof(nov), hipMemcpyDeviceToHost);
        if (ctx->place_over_max.p) (void)hipMemcpy(&omax, ctx->place_over_max.p, sizeof(omax), hipMemcpyDeviceToHost);
        fprintf(stderr, "[sppm] gather pieces queued in the last batch: %llu; deposits on the overflow list: %llu (largest since the last look %llu), "
                        "layout %s, room for %llu deposits\n", np, nov, omax, ctx->place_valid ? "valid" : "none", (unsigned long long)ctx->place_total);
    }
    if (ctx->verbose && dc.stamps[0])
        fprintf(stderr, "[sppm] trace loop cycles: refill %llu analytic %llu bezier %llu merge/park %llu finish %llu bounce %llu (photons: %llu of them up to the new direction) loop %llu\n",
                dc.stamps[0], dc.stamps[1], dc.stamps[2], dc.stamps[3], dc.stamps[4], dc.stamps[5] + dc.stamps[6], dc.stamps[6], dc.stamps[7]);
    ctx->check_pending = false;
    if (overflow_code(ctx, dc) != SPPM_OK) return fail(ctx, SPPM_E_CAPACITY, "a ray / task / deposit queue overflowed in an earlier pass");
    return SPPM_OK;
}

int sppm_get_device_stats(sppm_ctx* ctx, sppm_device_stats* out) {
    if (!ctx || !out) return SPPM_E_INVALID;
    HIPCHK(hipSetDevice(ctx->device));
    DevCounters dc;
    HIPCHK(hipMemcpyAsync(&dc, ctx->counters.p, sizeof(dc), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    out->stored_deposits = dc.stored;
    out->pair_tests = dc.row_tests * 64ull;
    out->deposit_room = ctx->place_valid ? ctx->place_total : 0;
    size_t b = 0;
    auto add = [&](size_t n, size_t elem) { b += n * elem; };
    add(ctx->blob.n, 1); add(ctx->texels.n, sizeof(uchar4)); add(ctx->pw_q.n, sizeof(float)); add(ctx->pw_c.n, sizeof(double));
    add(ctx->sub_box.n, sizeof(float)); add(ctx->lights.n, sizeof(DevLight)); add(ctx->raw.n, sizeof(RawHit));
    for (int k = 0; k < 2; k++) { add(ctx->park[k].n, sizeof(ParkRec)); add(ctx->ntask[k].n, sizeof(NewtonTask)); }
    add(ctx->nres.n, sizeof(float4)); add(ctx->hstart.n, 4); add(ctx->hist.n, 4);
    add(ctx->hp_rec.n, sizeof(HpRec)); add(ctx->hp_tao.n, sizeof(float4));
    add(ctx->hp_meta.n, sizeof(HpMeta)); add(ctx->hp_r2.n, sizeof(double));
    add(ctx->dep.n, sizeof(Deposit)); add(ctx->sdep.n, sizeof(Deposit)); add(ctx->pieces.n, sizeof(GatherPiece));
    add(ctx->place_room.n, 4); add(ctx->place_start.n, 4); add(ctx->place_over_max.n, 8); add(ctx->dstart.n, 4); add(ctx->scan_partial.n, 4);
    add(ctx->ctl.n, 8); add(ctx->counters.n, sizeof(DevCounters)); add(ctx->pic.n, sizeof(float));
    out->device_bytes = b;
    return SPPM_OK;
}

// Waits for the stream and sums the spans recorded since the last sppm_eye_pass call.
int sppm_get_timings(sppm_ctx* ctx, sppm_timings* out) {
    if (!ctx || !out) return SPPM_E_INVALID;
    HIPCHK(hipSetDevice(ctx->device));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    double ms[PH_COUNT] = {0, 0, 0, 0, 0, 0};
    for (const sppm_ctx::Span& sp : ctx->spans) {
        float t = 0;
        if (sp.a && sp.b && hipEventElapsedTime(&t, sp.a, sp.b) == hipSuccess) ms[sp.phase] += (double)t;
    }
    out->eye_trace_ms = ms[PH_EYE]; out->grid_build_ms = ms[PH_GRID]; out->photon_trace_ms = ms[PH_TRACE];
    out->deposit_sort_ms = ms[PH_SORT]; out->gather_ms = ms[PH_GATHER]; out->resolve_ms = ms[PH_RESOLVE];
    out->gather_launches = ctx->gather_launches; out->trace_launches = ctx->trace_launches;
    return SPPM_OK;
}

int sppm_debug_near_collide(sppm_ctx* ctx, uint64_t n, const double* org, const double* dir, uint64_t seed, uint32_t replica,
                            sppm_collision* out) {
    if (!ctx || !org || !dir || !out) return SPPM_E_INVALID;
    if (!ctx->have_scene) return fail(ctx, SPPM_E_INVALID, "sppm_set_scene must come first");
    if (n == 0) return SPPM_OK;
    if (n > (1u << 28)) return fail(ctx, SPPM_E_INVALID, "too many rays");
    HIPCHK(hipSetDevice(ctx->device));
    DevBuf<double> o, d; DevBuf<sppm_collision> c;
    int rc = SPPM_OK;
    do {
        if (o.reserve(3 * n) != hipSuccess || d.reserve(3 * n) != hipSuccess || c.reserve(n) != hipSuccess) { rc = fail(ctx, SPPM_E_NOMEM, "hipMalloc failed"); break; }
        (void)hipMemcpy(o.p, org, 3 * n * sizeof(double), hipMemcpyHostToDevice);
        (void)hipMemcpy(d.p, dir, 3 * n * sizeof(double), hipMemcpyHostToDevice);
        if (ctx->scene_in_lds)
            hipLaunchKernelGGL(k_debug_near_collide<true>, dim3((unsigned)((n + 255) / 256)), dim3(256), (size_t)ctx->blob_bytes, ctx->stream,
                               ctx->scene_params(), (uint32_t)n, o.p, d.p, host_rng_base(seed, replica * 4u + D_EYE), c.p, ctx->counters.p);
        else
            hipLaunchKernelGGL(k_debug_near_collide<false>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream,
                               ctx->scene_params(), (uint32_t)n, o.p, d.p, host_rng_base(seed, replica * 4u + D_EYE), c.p, ctx->counters.p);
        hipError_t e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) { rc = fail(ctx, SPPM_E_DEVICE, hipGetErrorString(e)); break; }
        (void)hipMemcpy(out, c.p, n * sizeof(sppm_collision), hipMemcpyDeviceToHost);
    } while (0);
    o.release(); d.release(); c.release();
    return rc;
}

// sppm_debug_near_collide without a single allocation: the rays and the collision records are staged through this
// GPU's copy of the static scratch array (g_scratch_ptr[device], never the array's name), the launch's tally block sits
// there as well and is cleared by a fill kernel first (the array keeps the last call's contents).  Same kernel, same random
// streams, same results as the allocating entry point.  At most SPPM_NOALLOC_RAYS rays per call.
int sppm_near_collide_noalloc(sppm_ctx* ctx, uint64_t n, const double* org, const double* dir, uint64_t seed, uint32_t replica,
                              sppm_collision* out) {
    if (!ctx || !org || !dir || !out) return SPPM_E_INVALID;
    if (!ctx->have_scene) return fail(ctx, SPPM_E_INVALID, "sppm_set_scene must come first");
    if (n == 0) return SPPM_OK;
    if (n > SPPM_NOALLOC_RAYS)
        return fail(ctx, SPPM_E_CAPACITY, "sppm_near_collide_noalloc takes at most " + std::to_string(SPPM_NOALLOC_RAYS) +
                                              " rays per call (the static scratch array holds no more); split the batch or use sppm_debug_near_collide");
    if (ctx->device < 0 || ctx->device >= 64 || !g_scratch_ptr[ctx->device]) return fail(ctx, SPPM_E_DEVICE, "no static scratch on this device");
    HIPCHK(hipSetDevice(ctx->device));
    std::lock_guard<std::mutex> turn(g_scratch_mu[ctx->device]);     // the GPU's one array serves one call at a time
    NoAllocScratch* s = (NoAllocScratch*)g_scratch_ptr[ctx->device];
    // Work that was queued before a failing step may still read or write the array: the turn is only given up (the guard
    // above, on return) once the stream is idle, whatever the outcome.
    auto run = [&]() -> int {
        HIPCHK(hipMemcpyAsync(s->org, org, 3 * n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
        HIPCHK(hipMemcpyAsync(s->dir, dir, 3 * n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
        if (int rc = launch_fill(ctx, (uint32_t*)&s->counters, sizeof(DevCounters) / 4, 0u)) return rc;
        const uint32_t rng = host_rng_base(seed, replica * 4u + D_EYE);
        if (ctx->scene_in_lds)
            hipLaunchKernelGGL(k_debug_near_collide<true>, dim3((unsigned)((n + 255) / 256)), dim3(256), (size_t)ctx->blob_bytes, ctx->stream,
                               ctx->scene_params(), (uint32_t)n, (const double*)s->org, (const double*)s->dir, rng, s->out, &s->counters);
        else
            hipLaunchKernelGGL(k_debug_near_collide<false>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream,
                               ctx->scene_params(), (uint32_t)n, (const double*)s->org, (const double*)s->dir, rng, s->out, &s->counters);
        HIPCHK(hipGetLastError());
        hipLaunchKernelGGL(k_add_newton_calls, dim3(1), dim3(64), 0, ctx->stream, ctx->counters.p, (const DevCounters*)&s->counters);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(out, s->out, n * sizeof(sppm_collision), hipMemcpyDeviceToHost, ctx->stream));
        return SPPM_OK;
    };
    const int rc = run();
    const hipError_t idle = hipStreamSynchronize(ctx->stream);
    if (rc != SPPM_OK) return rc;
    if (idle != hipSuccess) return fail(ctx, SPPM_E_DEVICE, hipGetErrorString(idle));
    return SPPM_OK;
}

int sppm_debug_photon_deposits(sppm_ctx* ctx, uint64_t rounds, uint32_t per_round, uint64_t photon_begin, uint64_t photon_count,
                               uint64_t seed, uint32_t replica, double* out, uint64_t cap, uint64_t* n) {
    if (!ctx || !n || (!out && cap)) return SPPM_E_INVALID;
    if (!ctx->have_eye) return fail(ctx, SPPM_E_INVALID, "sppm_eye_pass must come first");
    if (rounds == 0 || per_round == 0 || photon_count == 0) return fail(ctx, SPPM_E_INVALID, "empty photon range");
    if (rounds > 0xffffffffull) return fail(ctx, SPPM_E_INVALID, "more than 2^32 photons in one pass");
    const uint64_t total = rounds * (uint64_t)ctx->n_lights * per_round;
    if (total / per_round / (uint64_t)ctx->n_lights != rounds || total > 0xffffffffull) return fail(ctx, SPPM_E_INVALID, "more than 2^32 photons in one pass");
    if (photon_begin >= total) return fail(ctx, SPPM_E_INVALID, "photon_begin beyond the pass");
    const uint64_t count = std::min(photon_count, total - photon_begin);
    if (count > (1u << 22)) return fail(ctx, SPPM_E_INVALID, "at most 2^22 photons per debug call");
    HIPCHK(hipSetDevice(ctx->device));
    const uint64_t dep_cap = count * (SPPM_MAX_DEP + 1) + dep_slack(ctx);
    HIPCHK(ctx->dep.reserve(dep_cap)); HIPCHK(ctx->sdep.reserve(dep_cap));
    if (int rc = check_deferred(ctx)) return rc;
    DevBuf<DevCounters> scratch_counters;   // keep the pass counters untouched
    HIPCHK(scratch_counters.reserve(1));
    HIPCHK(hipMemsetAsync(scratch_counters.p, 0, sizeof(DevCounters), ctx->stream));
    TraceParams tp{};
    fill_photon_params(ctx, tp, per_round, dep_cap, seed, replica, scratch_counters.p);
    tp.occ = nullptr;                       // the parity hook returns every deposit, reachable by a HitPoint or not
    const uint32_t M = ctx->grid.mask + 1;
    int rrc = reset_ctl(ctx);
    if (rrc == SPPM_OK) rrc = launch_fill(ctx, ctx->dcount(), (size_t)M + 1, 0);
    DevCounters* saved = ctx->counters.p;
    ctx->counters.p = scratch_counters.p;   // the trace driver reads the overflow bits of the counters in use
    QueueState q{};
    bool synced = false;
    if (rrc == SPPM_OK) rrc = run_trace(ctx, MODE_PHOTON, tp, photon_begin, count, false, true, q, &synced);
    ctx->counters.p = saved;
    if (rrc == SPPM_OK) rrc = close_chunks(ctx, ctx->dep.p, nullptr, dep_cap);
    unsigned long long nd = 0;
    hipError_t e = hipMemcpyAsync(&nd, ctx->scalars() + SC_DEP, sizeof(nd), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    scratch_counters.release();
    ctx->check_pending = false;
    if (rrc != SPPM_OK) return rrc;
    if (e != hipSuccess) return fail(ctx, SPPM_E_DEVICE, hipGetErrorString(e));
    // nd counts claimed slots; unused ones are holes
    const uint64_t claimed = std::min<uint64_t>(nd, dep_cap);
    uint64_t real = 0;
    if (claimed) {
        std::vector<Deposit> recs(claimed);
        HIPCHK(hipMemcpy(recs.data(), ctx->dep.p, claimed * sizeof(Deposit), hipMemcpyDeviceToHost));
        for (uint64_t i = 0; i < claimed; i++) {
            const PosKey& a = recs[i].a;
            const DepB& b = recs[i].b;
            if ((b.bucket & 0xffffffu) == DEP_HOLE) continue;
            if (real < cap) {
                double* o = out + 11 * real;
                o[0] = (double)b.photon; o[1] = (double)(b.bucket >> 24);
                o[2] = a.x; o[3] = a.y; o[4] = a.z;
                o[5] = b.nx; o[6] = b.ny; o[7] = b.nz;
                o[8] = b.fr; o[9] = b.fg; o[10] = b.fb;
            }
            real++;
        }
    }
    *n = real;
    return real <= cap ? SPPM_OK : SPPM_E_CAPACITY;
}

// Ray records of the caller's making, uploaded as the seed queue of run_trace (queue 1).
// Both queues are brought to the size the trace rounds will ask for BEFORE the copy: run_trace must find nothing to grow.
static int upload_seed_rays(sppm_ctx* ctx, int mode, uint64_t n, const double* org, const double* dir, const int32_t* dep, const double* carry, const uint32_t* id) {
    std::vector<ParkRec> recs((size_t)n);
    for (uint64_t i = 0; i < n; i++) {
        ParkRec& r = recs[(size_t)i];
        memset(&r, 0, sizeof(r));
        for (int k = 0; k < 3; k++) { r.o[k] = org[3 * i + k]; r.d[k] = dir[3 * i + k]; r.carry[k] = (float)carry[3 * i + k]; }
        r.id = id[i]; r.dep = dep ? dep[i] : 0; r.node = 0; r.aobj = -1;
    }
    const uint64_t cap = trace_park_cap(ctx, mode, true, mode == MODE_EYE && ctx->has_dual_lobe, 0, n);
    HIPCHK(ctx->park[0].reserve(cap)); HIPCHK(ctx->park[1].reserve(cap));
    HIPCHK(hipMemcpyAsync(ctx->park[1].p, recs.data(), (size_t)n * sizeof(ParkRec), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));     // recs goes out of scope
    return SPPM_OK;
}

int sppm_view_trace(sppm_ctx* ctx, uint64_t n, const double* org, const double* dir, const int32_t* dep, const double* wgt,
                    const uint32_t* pixel, int32_t w, uint64_t seed, uint32_t replica, sppm_hitpoint* out, uint64_t cap, uint64_t* n_out) {
    if (!ctx || !org || !dir || !wgt || !pixel || !n_out || (!out && cap)) return SPPM_E_INVALID;
    *n_out = 0;
    if (!ctx->have_scene) return fail(ctx, SPPM_E_INVALID, "sppm_set_scene must come first");
    if (w <= 0) return fail(ctx, SPPM_E_INVALID, "image width must be positive");
    if (n == 0) return SPPM_OK;
    if (n > (1u << 22)) return fail(ctx, SPPM_E_INVALID, "at most 2^22 rays per call");
    for (uint64_t i = 0; i < n; i++) if (dep && (dep[i] < 0 || dep[i] > SPPM_MAX_DEP + 1)) return fail(ctx, SPPM_E_INVALID, "depth out of range");
    HIPCHK(hipSetDevice(ctx->device));
    if (int rc = check_deferred(ctx)) return rc;
    if (ctx->have_eye) if (int rc = ensure_meta(ctx)) return rc;     // the raw array is about to be reused
    if (int rc = upload_seed_rays(ctx, MODE_EYE, n, org, dir, dep, wgt, pixel)) return rc;
    // the HitPoints land in the raw array of the eye pass (free between passes); their histogram goes to a throw-away table
    uint64_t hit_cap = (SPPM_MAX_DEP + 1) * n + dep_slack(ctx, n);
    DevBuf<uint32_t> sink; DevBuf<DevCounters> tallies;
    int rc = SPPM_OK;
    QueueState q{};
    do {
        if (ctx->raw.reserve(hit_cap) != hipSuccess || sink.reserve(1u << 16) != hipSuccess || tallies.reserve(1) != hipSuccess) { rc = fail(ctx, SPPM_E_NOMEM, "hipMalloc failed"); break; }
        if ((rc = launch_pass_reset(ctx, false, nullptr, 0)) != SPPM_OK) break;
        if ((rc = launch_fill(ctx, (uint32_t*)tallies.p, sizeof(DevCounters) / 4, 0u)) != SPPM_OK) break;
        if ((rc = launch_fill(ctx, (uint32_t*)(ctx->scalars() + SC_PARK1), 1, (uint32_t)n)) != SPPM_OK) break;
        TraceParams tp{};
        tp.hcount = sink.p; tp.grid.inv_cs = 1.0; tp.grid.mask = 0xffffu;
        tp.w = w; tp.h = 1; tp.pix_mul = 1; tp.npx_magic = ~0ull; tp.w_magic = 0xffffffffu;
        tp.hits = ctx->raw.p; tp.n_hits = ctx->scalars() + SC_HITS; tp.hit_cap = hit_cap;
        tp.n_real_hits = ctx->scalars() + SC_REAL_HITS;
        tp.counters = tallies.p;
        tp.wave_chunks = ctx->wave_chunks();
        tp.rng = host_rng_base(seed, replica * 4u + D_EYE);
        DevCounters* saved = ctx->counters.p;
        ctx->counters.p = tallies.p;             // the trace driver reads the overflow bits of the counters in use
        bool synced = false;
        rc = run_trace(ctx, MODE_EYE, tp, 0, 0, ctx->has_dual_lobe, true, q, &synced, n);
        ctx->counters.p = saved;
        if (rc != SPPM_OK) { if (rc > 0) rc = fail(ctx, SPPM_E_CAPACITY, "a queue overflowed while tracing the given rays"); break; }
        if ((rc = close_chunks(ctx, nullptr, ctx->raw.p, hit_cap)) != SPPM_OK) break;
        hipError_t e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) { rc = fail(ctx, SPPM_E_DEVICE, hipGetErrorString(e)); break; }
        const uint64_t claimed = std::min<uint64_t>(q.sc[SC_HITS], hit_cap);
        std::vector<RawHit> raw((size_t)claimed);
        if (claimed && hipMemcpy(raw.data(), ctx->raw.p, (size_t)claimed * sizeof(RawHit), hipMemcpyDeviceToHost) != hipSuccess) { rc = fail(ctx, SPPM_E_DEVICE, "copy of the HitPoints failed"); break; }
        uint64_t real = 0;
        for (const RawHit& h : raw) {
            if (h.pix < 0) continue;
            if (real < cap) {
                sppm_hitpoint& o = out[real];
                memset(&o, 0, sizeof(o));
                for (int k = 0; k < 3; k++) { o.pos[k] = h.pos[k]; o.n[k] = h.n[k]; o.v[k] = h.v[k]; o.wgt[k] = h.wgt[k]; }
                o.R2 = SPPM_INIT_R2; o.obj = h.obj; o.x = h.pix / w; o.y = h.pix % w; o.node = h.node; o.N = ctx->initial_n; o.newN = 0;
            }
            real++;
        }
        *n_out = real;
        if (real > cap) rc = fail(ctx, SPPM_E_CAPACITY, "the caller's HitPoint array is too small");
    } while (0);
    sink.release(); tallies.release();
    ctx->check_pending = false;
    return rc;
}

int sppm_photon_trace(sppm_ctx* ctx, uint64_t n, const double* org, const double* dir, const int32_t* dep, const double* flux,
                      const uint32_t* id, uint64_t seed, uint32_t replica) {
    if (!ctx || !org || !dir || !flux || !id) return SPPM_E_INVALID;
    if (!ctx->have_eye) return fail(ctx, SPPM_E_INVALID, "sppm_eye_pass must come first");
    if (ctx->initial_n > 0) return fail(ctx, SPPM_E_INVALID, "single photons are not traced with the radius update on");
    if (n == 0) return SPPM_OK;
    if (n > (1u << 22)) return fail(ctx, SPPM_E_INVALID, "at most 2^22 photons per call");
    for (uint64_t i = 0; i < n; i++) if (dep && (dep[i] < 0 || dep[i] > SPPM_MAX_DEP + 1)) return fail(ctx, SPPM_E_INVALID, "depth out of range");
    HIPCHK(hipSetDevice(ctx->device));
    if (int rc = check_deferred(ctx)) return rc;
    if (int rc = upload_seed_rays(ctx, MODE_PHOTON, n, org, dir, dep, flux, id)) return rc;
    const uint32_t M = ctx->grid.mask + 1;
    const uint64_t dep_cap = (SPPM_MAX_DEP + 1) * n + dep_slack(ctx, n);
    HIPCHK(ctx->dep.reserve(dep_cap)); HIPCHK(ctx->sdep.reserve(std::max<uint64_t>(dep_cap, ctx->place_valid ? ctx->place_total : 0)));
    HIPCHK(ctx->place_over_max.reserve(1));
    TraceParams tp{};
    fill_photon_params(ctx, tp, 1, dep_cap, seed, replica, ctx->counters.p);
    if (int rc = launch_pass_reset(ctx, false, ctx->cursor(), 2 * ((size_t)M + 1))) return rc;      // + cursor, dcount
    if (int rc = launch_fill(ctx, (uint32_t*)(ctx->scalars() + SC_PARK1), 1, (uint32_t)n)) return rc;
    QueueState q{};
    bool synced = false;
    tic(ctx, PH_TRACE);
    int rc = run_trace(ctx, MODE_PHOTON, tp, 0, 0, false, false, q, &synced, n);
    toc(ctx);
    if (rc != SPPM_OK) return rc > 0 ? fail(ctx, SPPM_E_CAPACITY, "a queue overflowed while tracing the given photons") : rc;
    tic(ctx, PH_SORT);
    if (int rc2 = launch_scan(ctx, ctx->dcount(), ctx->dstart.p, M, ctx->dep.p, nullptr, dep_cap)) return rc2;   // + chunk tails -> holes
    hipLaunchKernelGGL(k_dep_scatter, dim3(ctx->num_cu * ctx->scatter_blocks_per_cu), dim3(256), 0, ctx->stream, ctx->dep.p, ctx->scalars() + SC_DEP,
                       dep_cap, ctx->dstart.p, ctx->cursor(), ctx->sdep.p);
    HIPCHK(hipGetLastError());
    toc(ctx);
    if (ctx->n_hp > 0)
        if (int rc2 = launch_gather(ctx, ctx->dstart.p, ctx->dcount(), nullptr, dep_cap, !ctx->radii_changed)) return rc2;
    if (ctx->last_rounds == 0) { ctx->last_rounds = 1; ctx->last_per_round = 1; }     // a resolve with nothing but such photons: tao / (PI R2)
    return SPPM_OK;
}

int sppm_debug_newton(sppm_ctx* ctx, uint64_t n, const double* org, const double* dir, const int32_t* patch, uint64_t seed,
                      uint32_t replica, double* out_tuv) {
    if (!ctx || !org || !dir || !patch || !out_tuv) return SPPM_E_INVALID;
    if (!ctx->have_scene) return fail(ctx, SPPM_E_INVALID, "sppm_set_scene must come first");
    if (n == 0) return SPPM_OK;
    if (n > (1u << 28)) return fail(ctx, SPPM_E_INVALID, "too many rays");
    for (uint64_t i = 0; i < n; i++)
        if (patch[i] < 0 || patch[i] >= ctx->n_patches) return fail(ctx, SPPM_E_INVALID, "patch index out of range");
    HIPCHK(hipSetDevice(ctx->device));
    DevBuf<double> o, d, r; DevBuf<int> p;
    int rc = SPPM_OK;
    do {
        if (o.reserve(3 * n) != hipSuccess || d.reserve(3 * n) != hipSuccess || r.reserve(3 * n) != hipSuccess || p.reserve(n) != hipSuccess) {
            rc = fail(ctx, SPPM_E_NOMEM, "hipMalloc failed"); break;
        }
        (void)hipMemcpy(o.p, org, 3 * n * sizeof(double), hipMemcpyHostToDevice);
        (void)hipMemcpy(d.p, dir, 3 * n * sizeof(double), hipMemcpyHostToDevice);
        (void)hipMemcpy(p.p, patch, n * sizeof(int), hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k_debug_newton, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream,
                           ctx->scene_params(), (uint32_t)n, o.p, d.p, p.p, host_rng_base(seed, replica * 4u + D_EYE), r.p);
        hipError_t e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) { rc = fail(ctx, SPPM_E_DEVICE, hipGetErrorString(e)); break; }
        (void)hipMemcpy(out_tuv, r.p, 3 * n * sizeof(double), hipMemcpyDeviceToHost);
    } while (0);
    o.release(); d.release(); r.release(); p.release();
    return rc;
}

}  // extern "C"

// ------------------------------------------------------------------ film (Raytracer.h:449-470)
#define FILM_FLAG_CAP 4096u
struct sppm_film {
    int device = 0, w = 0, h = 0;
    hipStream_t stream = nullptr;
    DevBuf<double> acc;
    DevBuf<double> total;                       // sum over the films of all GPUs, left here by sppm_film_allreduce
    bool have_total = false;
    DevBuf<uint8_t> bytes;
    DevBuf<uint32_t> flagged;
    std::mutex mu;                              // contexts add from their own host threads
    std::map<sppm_ctx*, hipEvent_t> last_add;   // the latest add queued by each context
    std::string err;
    size_t elems() const { return (size_t)w * h * 3; }
};

static int film_fail(sppm_film* f, int code, const std::string& msg) {
    if (f) f->err = msg;
    return code;
}
#define FILMCHK(expr)                                                                                                   \
    do {                                                                                                                \
        hipError_t e_ = (expr);                                                                                         \
        if (e_ != hipSuccess) return film_fail(film, SPPM_E_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

// The film's stream waits for every add queued so far.
static int film_join(sppm_film* film) {
    for (auto& kv : film->last_add) FILMCHK(hipStreamWaitEvent(film->stream, kv.second, 0));
    return SPPM_OK;
}

static inline uint8_t host_to_int(double x) { return (uint8_t)(int)(std::pow(1 - std::exp(-x), 1 / 2.2) * 255 + .5); }

extern "C" {

int sppm_film_create(sppm_film** out, int device_id, int w, int h) {
    if (!out) return SPPM_E_INVALID;
    *out = nullptr;
    if (w <= 0 || h <= 0 || (uint64_t)w * (uint64_t)h * 3 >= (1ull << 32)) return SPPM_E_INVALID;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device_id < 0 || device_id >= ndev) return SPPM_E_DEVICE;
    if (hipSetDevice(device_id) != hipSuccess) return SPPM_E_DEVICE;
    sppm_film* film = new (std::nothrow) sppm_film();
    if (!film) return SPPM_E_NOMEM;
    film->device = device_id; film->w = w; film->h = h;
    if (hipStreamCreate(&film->stream) != hipSuccess) { delete film; return SPPM_E_DEVICE; }
    if (film->acc.reserve(film->elems()) != hipSuccess || film->bytes.reserve(film->elems()) != hipSuccess ||
        film->flagged.reserve(1 + FILM_FLAG_CAP) != hipSuccess || hipMemset(film->acc.p, 0, film->elems() * sizeof(double)) != hipSuccess) {
        film->acc.release(); film->bytes.release(); film->flagged.release();
        (void)hipStreamDestroy(film->stream);
        delete film;
        return SPPM_E_NOMEM;
    }
    *out = film;
    return SPPM_OK;
}

void sppm_film_destroy(sppm_film* film) {
    if (!film) return;
    (void)hipSetDevice(film->device);
    for (auto& kv : film->last_add) { (void)hipEventSynchronize(kv.second); (void)hipEventDestroy(kv.second); }
    (void)hipStreamSynchronize(film->stream);
    film->acc.release(); film->bytes.release(); film->flagged.release(); film->total.release();
    (void)hipStreamDestroy(film->stream);
    delete film;
}

const char* sppm_film_last_error(const sppm_film* film) { return film ? film->err.c_str() : "null film"; }

int sppm_film_add(sppm_ctx* ctx, sppm_film* film) {
    if (!ctx || !film) return SPPM_E_INVALID;
    if (!ctx->have_eye) return fail(ctx, SPPM_E_INVALID, "sppm_eye_pass must come first");
    if (ctx->last_rounds == 0) return fail(ctx, SPPM_E_INVALID, "sppm_photon_pass must come first");
    if (ctx->device != film->device) return fail(ctx, SPPM_E_INVALID, "the film lives on another GPU than the context");
    if (ctx->cam.w != film->w || ctx->cam.h != film->h) return fail(ctx, SPPM_E_INVALID, "the film has another resolution than the eye pass");
    HIPCHK(hipSetDevice(ctx->device));
    hipEvent_t ev = nullptr;
    {
        std::lock_guard<std::mutex> g(film->mu);
        auto it = film->last_add.find(ctx);
        if (it == film->last_add.end()) {
            HIPCHK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
            film->last_add[ctx] = ev;
        } else {
            ev = it->second;
        }
    }
    tic(ctx, PH_RESOLVE);
    if (ctx->n_hp > 0) {
        const double inv_norm = 1.0 / SPPM_PI / ((double)ctx->last_rounds * (double)ctx->last_per_round);
        hipLaunchKernelGGL(k_film_add, dim3((unsigned)((ctx->n_hp + 255) / 256)), dim3(256), 0, ctx->stream, ctx->hp_r2.p, ctx->hp_rec.p,
                           ctx->hp_tao.p, (uint32_t)ctx->n_hp, inv_norm, film->acc.p);
        HIPCHK(hipGetLastError());
    }
    toc(ctx);
    std::lock_guard<std::mutex> g(film->mu);   // a reader on another thread walks last_add
    HIPCHK(hipEventRecord(ev, ctx->stream));
    return SPPM_OK;
}

int sppm_film_clear(sppm_film* film) {
    if (!film) return SPPM_E_INVALID;
    std::lock_guard<std::mutex> g(film->mu);
    FILMCHK(hipSetDevice(film->device));
    if (int rc = film_join(film)) return rc;
    FILMCHK(hipMemsetAsync(film->acc.p, 0, film->elems() * sizeof(double), film->stream));
    FILMCHK(hipStreamSynchronize(film->stream));
    return SPPM_OK;
}

int sppm_film_read(sppm_film* film, double* pic_hw3) {
    if (!film || !pic_hw3) return SPPM_E_INVALID;
    std::lock_guard<std::mutex> g(film->mu);
    FILMCHK(hipSetDevice(film->device));
    if (int rc = film_join(film)) return rc;
    FILMCHK(hipMemcpyAsync(pic_hw3, film->acc.p, film->elems() * sizeof(double), hipMemcpyDeviceToHost, film->stream));
    FILMCHK(hipStreamSynchronize(film->stream));
    return SPPM_OK;
}

int sppm_film_write(sppm_film* film, const double* pic_hw3) {
    if (!film || !pic_hw3) return SPPM_E_INVALID;
    std::lock_guard<std::mutex> g(film->mu);
    FILMCHK(hipSetDevice(film->device));
    if (int rc = film_join(film)) return rc;
    FILMCHK(hipMemcpyAsync(film->acc.p, pic_hw3, film->elems() * sizeof(double), hipMemcpyHostToDevice, film->stream));
    FILMCHK(hipStreamSynchronize(film->stream));
    return SPPM_OK;
}

// toInt of src / n_images (src: the film's own sum or the total over GPUs), see sppm_film_tonemap
static int film_tonemap_from(sppm_film* film, const double* src, double n_images, int rgb, uint8_t* out_hw3) {
    const size_t n = film->elems();
    hipLaunchKernelGGL(k_fill_u32, dim3(1), dim3(64), 0, film->stream, film->flagged.p, (size_t)1, 0u);
    hipLaunchKernelGGL(k_tonemap, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, film->stream, src, film->w, film->h, n_images, rgb,
                       film->bytes.p, film->flagged.p, FILM_FLAG_CAP);
    FILMCHK(hipGetLastError());
    uint32_t n_flagged = 0;
    FILMCHK(hipMemcpyAsync(out_hw3, film->bytes.p, n, hipMemcpyDeviceToHost, film->stream));
    FILMCHK(hipMemcpyAsync(&n_flagged, film->flagged.p, sizeof(uint32_t), hipMemcpyDeviceToHost, film->stream));
    FILMCHK(hipStreamSynchronize(film->stream));
    if (n_flagged == 0) return SPPM_OK;
    // values on a rounding step of the truncation: decided by the host's exp / pow
    auto at = [&](size_t e) {
        const int k = (int)(e % 3);
        const size_t p = e / 3;
        const size_t j = p % (size_t)film->w, i1 = p / (size_t)film->w;
        return ((size_t)(film->h - 1 - (int)i1) * film->w + j) * 3 + (size_t)(rgb ? 2 - k : k);
    };
    if (n_flagged <= FILM_FLAG_CAP) {
        std::vector<uint32_t> idx(n_flagged);
        FILMCHK(hipMemcpy(idx.data(), film->flagged.p + 1, n_flagged * sizeof(uint32_t), hipMemcpyDeviceToHost));
        for (uint32_t e : idx) {
            double x = 0;
            FILMCHK(hipMemcpy(&x, src + e, sizeof(double), hipMemcpyDeviceToHost));
            out_hw3[at(e)] = host_to_int(x / n_images);
        }
    } else {
        std::vector<double> all(n);
        FILMCHK(hipMemcpy(all.data(), src, n * sizeof(double), hipMemcpyDeviceToHost));
        for (size_t e = 0; e < n; e++) out_hw3[at(e)] = host_to_int(all[e] / n_images);
    }
    return SPPM_OK;
}

int sppm_film_tonemap(sppm_film* film, double n_images, int rgb, uint8_t* out_hw3) {
    if (!film || !out_hw3) return SPPM_E_INVALID;
    if (!(n_images > 0)) return film_fail(film, SPPM_E_INVALID, "n_images must be positive");
    std::lock_guard<std::mutex> g(film->mu);
    FILMCHK(hipSetDevice(film->device));
    if (int rc = film_join(film)) return rc;
    return film_tonemap_from(film, film->acc.p, n_images, rgb, out_hw3);
}

int sppm_film_tonemap_total(sppm_film* film, double n_images, int rgb, uint8_t* out_hw3) {
    if (!film || !out_hw3) return SPPM_E_INVALID;
    if (!(n_images > 0)) return film_fail(film, SPPM_E_INVALID, "n_images must be positive");
    std::lock_guard<std::mutex> g(film->mu);
    if (!film->have_total) return film_fail(film, SPPM_E_INVALID, "sppm_film_allreduce must come first");
    FILMCHK(hipSetDevice(film->device));
    return film_tonemap_from(film, film->total.p, n_images, rgb, out_hw3);
}

int sppm_film_read_total(sppm_film* film, double* pic_hw3) {
    if (!film || !pic_hw3) return SPPM_E_INVALID;
    std::lock_guard<std::mutex> g(film->mu);
    if (!film->have_total) return film_fail(film, SPPM_E_INVALID, "sppm_film_allreduce must come first");
    FILMCHK(hipSetDevice(film->device));
    FILMCHK(hipMemcpyAsync(pic_hw3, film->total.p, film->elems() * sizeof(double), hipMemcpyDeviceToHost, film->stream));
    FILMCHK(hipStreamSynchronize(film->stream));
    return SPPM_OK;
}

}  // extern "C"

// ------------------------------------------------------------------ film sum over the GPUs of one process (RCCL over xGMI)
// The serial image sum of Raytracer.h:449-458 when the replicas ran on several GPUs: one all-reduce (fp64 sum) over the
// films, one per GPU.  RCCL is opened at run time: a copy the process already holds (a host that has loaded PyTorch has
// PyTorch's) is taken first, so that no second ROCm runtime comes in; otherwise librccl.so.1 from the library path.
namespace {
struct FilmRccl {
    void* lib = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    std::string err;
    std::map<std::vector<int>, std::vector<ncclComm_t>> comms;     // one communicator set per list of devices, kept for the process
    std::mutex mu;
    bool open() {
        if (lib) return true;
        for (const char* name : {"librccl.so.1", "librccl.so"}) if ((lib = dlopen(name, RTLD_NOW | RTLD_NOLOAD))) break;
        if (!lib) for (const char* name : {"librccl.so.1", "librccl.so"}) if ((lib = dlopen(name, RTLD_NOW | RTLD_LOCAL))) break;
        if (!lib) { err = std::string("cannot open librccl: ") + dlerror(); return false; }
        CommInitAll = (decltype(CommInitAll))dlsym(lib, "ncclCommInitAll");
        CommDestroy = (decltype(CommDestroy))dlsym(lib, "ncclCommDestroy");
        AllReduce = (decltype(AllReduce))dlsym(lib, "ncclAllReduce");
        GroupStart = (decltype(GroupStart))dlsym(lib, "ncclGroupStart");
        GroupEnd = (decltype(GroupEnd))dlsym(lib, "ncclGroupEnd");
        GetErrorString = (decltype(GetErrorString))dlsym(lib, "ncclGetErrorString");
        if (!CommInitAll || !CommDestroy || !AllReduce || !GroupStart || !GroupEnd || !GetErrorString) {
            err = "librccl lacks an expected symbol"; lib = nullptr; return false;
        }
        return true;
    }
};
FilmRccl& film_rccl() { static FilmRccl r; return r; }
}  // namespace

extern "C" int sppm_film_allreduce(sppm_film* const* films, int n) {
    if (!films || n <= 0 || n > 64) return SPPM_E_INVALID;
    sppm_film* first = films[0];
    if (!first) return SPPM_E_INVALID;
    std::vector<int> devs;
    for (int i = 0; i < n; i++) {
        if (!films[i]) return SPPM_E_INVALID;
        if (films[i]->w != first->w || films[i]->h != first->h) return film_fail(first, SPPM_E_INVALID, "films of different resolutions");
        for (int d : devs) if (d == films[i]->device) return film_fail(first, SPPM_E_INVALID, "two films on one GPU: one film per GPU takes part in the sum");
        devs.push_back(films[i]->device);
    }
    FilmRccl& r = film_rccl();
    std::lock_guard<std::mutex> g(r.mu);
    if (!r.open()) return film_fail(first, SPPM_E_DEVICE, r.err);
    auto it = r.comms.find(devs);
    if (it == r.comms.end()) {
        std::vector<ncclComm_t> c((size_t)n);
        ncclResult_t rc = r.CommInitAll(c.data(), n, devs.data());
        if (rc != ncclSuccess) return film_fail(first, SPPM_E_DEVICE, std::string("ncclCommInitAll: ") + r.GetErrorString(rc));
        it = r.comms.emplace(devs, c).first;
    }
    const size_t count = first->elems();
    for (int i = 0; i < n; i++) {            // each film's stream waits for the adds queued on it so far
        sppm_film* film = films[i];
        std::lock_guard<std::mutex> fg(film->mu);
        FILMCHK(hipSetDevice(film->device));
        FILMCHK(film->total.reserve(count));
        if (int rc = film_join(film)) return rc;
    }
    ncclResult_t rc = r.GroupStart();
    for (int i = 0; i < n && rc == ncclSuccess; i++)
        rc = r.AllReduce(films[i]->acc.p, films[i]->total.p, count, ncclDouble, ncclSum, it->second[(size_t)i], films[i]->stream);
    const ncclResult_t rc_end = r.GroupEnd();
    if (rc == ncclSuccess) rc = rc_end;
    if (rc != ncclSuccess) return film_fail(first, SPPM_E_DEVICE, std::string("ncclAllReduce: ") + r.GetErrorString(rc));
    for (int i = 0; i < n; i++) {
        sppm_film* film = films[i];
        FILMCHK(hipSetDevice(film->device));
        FILMCHK(hipStreamSynchronize(film->stream));
        film->have_total = true;
    }
    return SPPM_OK;
}
