"""The device-resident image sum and tone-map (SPPMRayTracer::render, raytracer/Raytracer.h:449-470)
against the CPU oracle: the fp64 sum over replicas within the flux-image tolerance of the parity
tests, the tone-mapped bytes bit-exact (byte work), the fp64 read / write round trip exact."""
import numpy as np
import pytest

from raytrace3_amd import Film, SppmContext, SppmError, scenes
from tests import oracle_lib as ol

pytestmark = pytest.mark.gpu

SEED = 20260101
W, H = 64, 48


def _step_values(n):
    """Image values that sit on the rounding steps of toInt: x with pow(1 - exp(-x), 1/2.2) * 255 + .5 == m up to
    the last place, m = 1..255, where the device's exp / pow may truncate to the other side than the host's."""
    m = (np.arange(n) % 255) + 1
    y = ((m - 0.5) / 255.0) ** 2.2
    x = -np.log1p(-y)
    return np.nextafter(x, np.where(np.arange(n) % 2 == 0, 0.0, 1e9))   # alternately just below / just above the step


def test_film_sums_replicas_like_the_host_loop():
    scene = scenes.scene_teapot(1)
    a, b = SppmContext(0), SppmContext(0)
    film = Film(W, H)
    try:
        for c in (a, b):
            c.set_scene(scene)
        rng = np.random.default_rng(3)
        host_sum = np.zeros((H, W, 3))
        oracle_sum = np.zeros((H, W, 3))
        o = ol.OracleContext()
        o.set_scene(scene)
        for replica in range(4):        # the four jittered replicas of one outer iteration, dealt over two contexts
            j = rng.normal(size=3)
            cam = scenes.default_camera(W, H, jitter=0.00015 * j / np.linalg.norm(j))
            c = (a, b)[replica % 2]
            c.eye_pass(cam, SEED, replica)
            c.photon_pass(40, 100, SEED, replica)
            film.add(c)
            host_sum += c.resolve()     # the float image of the same pass, summed on the host in fp64
            o.eye_pass(cam, SEED, replica)
            o.photon_pass(40, 100, SEED, replica)
            oracle_sum += o.resolve()
        got = film.read()
        assert got.sum() > 0
        # same HitPoints and tao; the film multiplies in fp64 where the float image rounds each pixel to fp32
        assert np.allclose(got, host_sum, rtol=1e-5, atol=1e-7 * host_sum.mean())
        # against the oracle: the flux-image tolerance of the parity tests (relative to the image mean)
        d = np.abs(got - oracle_sum).max(axis=2) / oracle_sum.mean()
        print(f"film vs oracle: within 1e-3 {(d <= 1e-3).mean():.5f}, outliers {int((d > 1e-3).sum())} of {d.size}, MAE/mean {d.mean():.3g}")
        assert (d <= 1e-3).mean() >= 0.999 and d.mean() <= 1e-6          # measured: no outlier among 3 072 pixels, MAE 2.3e-7
        # tone-mapped bytes of the device against the oracle's toInt on the same sum, both channel orders
        for rgb in (False, True):
            assert np.array_equal(film.tonemap(4.0, rgb=rgb), ol.tonemap(got, 4.0, rgb=rgb))
        # ... and against the bytes of the oracle's own sum: off by one level at most, on few pixels
        diff = np.abs(film.tonemap(4.0).astype(int) - ol.tonemap(oracle_sum, 4.0).astype(int))
        print(f"tone-mapped bytes vs the oracle's: differ on {(diff > 0).mean():.5f}, by more than one level on {(diff > 1).mean():.5f}, max {diff.max()}")
        assert diff.max() <= 1 and (diff > 0).mean() <= 0.001                # measured: every byte equal
        film.clear()
        assert not film.read().any()
    finally:
        film.close(); a.close(); b.close()


def test_tonemap_bytes_are_the_hosts_on_rounding_steps():
    film = Film(W, H)
    try:
        rng = np.random.default_rng(11)
        pic = rng.gamma(0.5, 0.6, size=(H, W, 3))
        pic[0, :5] = 0.0                         # black
        pic[1, :5] = 1e300                       # saturated
        pic[2].reshape(-1)[:] = _step_values(W * 3)   # a row of values on the steps: few enough for the listed recompute
        for n_images in (1.0, 4.0, 12.0):
            film.write(pic * n_images)
            assert np.array_equal(film.read(), pic * n_images)            # the fp64 sum round-trips exactly
            for rgb in (False, True):
                want = ol.tonemap(pic * n_images, n_images, rgb=rgb)
                assert np.array_equal(film.tonemap(n_images, rgb=rgb), want)
        # every element on a step: more than the device lists, the whole image is recomputed on the host
        allsteps = _step_values(H * W * 3).reshape(H, W, 3)
        film.write(allsteps)
        assert np.array_equal(film.tonemap(1.0), ol.tonemap(allsteps, 1.0))
        # layout: row flip and channel order
        ramp = np.zeros((H, W, 3)); ramp[0, 0] = (0.1, 0.5, 3.0)
        film.write(ramp)
        t = film.tonemap(1.0)
        assert tuple(t[H - 1, 0]) == tuple(int(ol.load().oracle_to_int(v)) for v in (0.1, 0.5, 3.0))
        assert tuple(film.tonemap(1.0, rgb=True)[H - 1, 0]) == tuple(t[H - 1, 0][::-1])
    finally:
        film.close()


def test_film_argument_checks():
    scene = scenes.scene_spheres_plane(False)
    c = SppmContext(0)
    film = Film(W, H)
    try:
        c.set_scene(scene)
        c.eye_pass(scenes.default_camera(W, H), SEED, 0)
        with pytest.raises(SppmError):
            film.add(c)                               # no photon pass yet
        c.photon_pass(5, 100, SEED, 0)
        film.add(c)
        other = Film(W + 1, H)
        with pytest.raises(SppmError):
            other.add(c)                              # another resolution
        other.close()
        with pytest.raises(SppmError):
            film.tonemap(0.0)
        with pytest.raises(ValueError):
            film.write(np.zeros((H, W + 1, 3)))
        with pytest.raises(SppmError):
            Film(0, 4)
    finally:
        film.close(); c.close()


def test_five_contexts_in_flight_add_up_like_one():
    """Five contexts on one GPU, used in turn with nothing but the eye pass's own wait between them (bench.py's loop:
    their persistent grids share the CUs, their adds meet in one film), against the same replicas run one after the
    other by a single context: the same HitPoints per replica, the same image sum up to the order of the additions."""
    scene = scenes.scene_teapot(1)
    w, h, n_iter = 160, 120, 15
    rng = np.random.default_rng(11)
    cams = []
    for _ in range(n_iter):
        j = rng.normal(size=3)
        cams.append(scenes.default_camera(w, h, jitter=0.00015 * j / np.linalg.norm(j)))
    ctxs = [SppmContext(0) for _ in range(5)]
    film = Film(w, h)
    try:
        for c in ctxs:
            c.set_scene(scene)
        for i, cam in enumerate(cams):
            c = ctxs[i % len(ctxs)]
            c.eye_pass(cam, SEED, i)
            c.photon_pass(300, 100, SEED, i)
            film.add(c)
        for c in ctxs:
            c.synchronize()
        together = film.read()
        counts = [c.counters() for c in ctxs]
    finally:
        film.close()
        for c in ctxs:
            c.close()
    one = SppmContext(0)                # alone on the GPU now: full grids, one pass after the other
    try:
        one.set_scene(scene)
        alone = np.zeros((h, w, 3))
        for i, cam in enumerate(cams):
            one.eye_pass(cam, SEED, i)
            one.photon_pass(300, 100, SEED, i)
            alone += one.resolve()
            if i >= n_iter - 5:         # the last pass of each of the five contexts: tallies equal to the lone context's
                ca, cb = counts[i % 5], one.counters()
                assert all(ca[k] == cb[k] for k in ("photons", "segments", "deposits", "candidates", "accepted", "hitpoints")), (i, ca, cb)
    finally:
        one.close()
    assert alone.sum() > 0
    assert np.allclose(together, alone, rtol=2e-4, atol=1e-6 * alone.mean())


def test_film_allreduce_through_rccl_with_the_one_rank_this_box_has():
    """sppm_film_allreduce sums the films of several GPUs with one RCCL all-reduce (fp64) instead of on the host.  On a
    one-GPU box the communicator has one rank: the path through librccl (opened at run time, ncclCommInitAll,
    ncclAllReduce on the film's stream, ordered after the adds queued on it) runs for real and the total must be the film's
    own sum, which stays untouched; tone-mapping the total gives the bytes of tone-mapping the film.  More than one rank is
    unmeasured here (no box with two GPUs)."""
    scene, cam = scenes.scene_teapot(1), scenes.default_camera(W, H)
    g = SppmContext(0)
    film, other = Film(W, H), Film(W, H)
    try:
        g.set_scene(scene)
        with pytest.raises(SppmError, match="sppm_film_allreduce must come first"):
            film.read_total()
        for replica in range(2):
            g.eye_pass(cam, SEED, replica)
            g.photon_pass(40, 100, SEED, replica)
            film.add(g)
            Film.allreduce([film])                    # queued behind the add, no host wait in between
            own = film.read()
            assert own.sum() > 0 and np.array_equal(film.read_total(), own)
        for rgb in (False, True):
            assert np.array_equal(film.tonemap(2.0, rgb=rgb, total=True), film.tonemap(2.0, rgb=rgb))
        # the film's own sum keeps accumulating after a reduction (the total is a second buffer)
        g.eye_pass(cam, SEED, 2)
        g.photon_pass(40, 100, SEED, 2)
        film.add(g)
        assert film.read().sum() > own.sum() and np.array_equal(film.read_total(), own)
        with pytest.raises(SppmError, match="two films on one GPU"):
            Film.allreduce([film, other])
    finally:
        film.close(); other.close(); g.close()
