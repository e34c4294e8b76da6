"""sppm_near_collide_noalloc: the variant of sppm_debug_near_collide that stages through the library's static device
array and allocates nothing.  Same kernel and random streams, so every record must equal the allocating entry point's
bit for bit; two calls in a row, the second with fewer rays, show that nothing of the first call's staged rays or records
leaks into the second."""
import numpy as np
import pytest

from raytrace3_amd import SppmContext, SppmError, abi, scenes

pytestmark = pytest.mark.gpu

SEED = 20260101


def _rays(n, rng):
    org = np.tile([50., 35., 230.], (n, 1)) + rng.normal(size=(n, 3)) * [8, 5, 3]
    tgt = np.array([20., 6., 120.]) + rng.normal(size=(n, 3)) * [25, 12, 25]       # around the teapot and past it
    d = tgt - org
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    return org, d


def _same(a, b):
    assert a.dtype == b.dtype and a.shape == b.shape
    assert a.tobytes() == b.tobytes()


def test_noalloc_near_collide_equals_the_allocating_entry_point():
    rng = np.random.default_rng(5)
    g = SppmContext(0)
    try:
        g.set_scene(scenes.scene_teapot(1, textured=True))
        org1, d1 = _rays(5000, rng)
        org2, d2 = _rays(1200, rng)
        want1, want2 = g.near_collide(org1, d1, SEED), g.near_collide(org2, d2, SEED, replica=3)
        c0 = g.counters()["newton_calls"]
        got1 = g.near_collide(org1, d1, SEED, noalloc=True)
        c1 = g.counters()["newton_calls"]
        got2 = g.near_collide(org2, d2, SEED, replica=3, noalloc=True)           # second call, smaller input
        c2 = g.counters()["newton_calls"]
        _same(got1, want1)
        _same(got2, want2)
        assert (want1["obj"] == 8).sum() > 200 and (want1["obj"] >= 0).sum() > 4000    # teapot and walls were hit
        assert c1 > c0 and c2 > c1                       # the Newton tally reaches the context's counters as before
        # the largest batch the static array holds, and one ray more
        orgm, dm = _rays(abi.NOALLOC_MAX_RAYS, rng)
        _same(g.near_collide(orgm, dm, SEED, noalloc=True), g.near_collide(orgm, dm, SEED))
        with pytest.raises(SppmError, match="at most 16384 rays"):
            g.near_collide(*_rays(abi.NOALLOC_MAX_RAYS + 1, rng), SEED, noalloc=True)
        _same(g.near_collide(org2, d2, SEED, replica=3, noalloc=True), want2)    # still usable after the refusal
        # another context on the same GPU shares the array
        h = SppmContext(0)
        try:
            h.set_scene(scenes.scene_spheres_plane(True))
            a = h.near_collide(org2, d2, SEED, noalloc=True)
            _same(a, h.near_collide(org2, d2, SEED))
            _same(g.near_collide(org1, d1, SEED, noalloc=True), want1)
        finally:
            h.close()
    finally:
        g.close()
