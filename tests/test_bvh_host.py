"""Host acceleration-structure builder (replacement of the reference's Embree commit, src/scene.cpp:20-31):
gdpt_bvh_check builds the BVH2 and its collapsed wide form and verifies coverage / enclosure itself; these tests
drive it with random, degenerate and real-scene inputs. CPU only."""
import os

import numpy as np
import pytest

import gdpt_amd as G

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def rand_boxes(rng, n, spread=10.0, size=0.5):
    c = rng.uniform(-spread, spread, size=(n, 3))
    h = rng.uniform(0, size, size=(n, 3))
    return np.concatenate([c - h, c + h], axis=1).astype(np.float32)


@pytest.mark.parametrize("n", [1, 2, 3, 4, 5, 7, 64, 1000, 20000])
def test_random_boxes(n):
    st = G.bvh_check(rand_boxes(np.random.default_rng(n), n))
    assert st["max_leaf_prims"] <= 4
    assert 1 <= st["wide_arity"] <= 4
    assert st["wide_stack_need"] <= 32 and st["bvh2_depth"] <= 32
    if n > 4:
        assert st["wide_nodes"] < st["bvh2_nodes"]
    if n > 64:
        assert st["bvh8_nodes"] < st["wide_nodes"]        # the quantised 8-wide form of the same tree (verified inside the check)


def test_empty():
    st = G.bvh_check(np.zeros((0, 6), np.float32))
    assert st["bvh2_nodes"] == 0 and st["wide_nodes"] == 0


def test_identical_centroids_and_flat_boxes():
    # all centroids coincide (SAH has nothing to split on) and zero-thickness boxes
    b = np.tile(np.array([[0, 0, 0, 1, 1, 0]], np.float32), (37, 1))
    st = G.bvh_check(b)
    assert st["bvh2_depth"] <= 32 and st["leaves"] >= 10


def test_long_diagonal_chain():
    # boxes on a line with geometrically growing sizes: drives the builder towards deep, unbalanced trees
    n = 4000
    t = (1.01 ** np.arange(n)).astype(np.float64)
    b = np.stack([t, t, t, t * 1.001, t * 1.001, t * 1.001], axis=1).astype(np.float32)
    st = G.bvh_check(b)
    assert st["bvh2_depth"] <= 32 and st["wide_stack_need"] <= 32


@pytest.mark.parametrize("scene", ["cbox/cbox_gdpt.xml", "sponza/sponza.xml", "disney_bsdf_test/disney_bsdf.xml"])
def test_scene_geometry(scene):
    sd = G.parse_scene(os.path.join(ROOT, "scenes", scene))
    b = G.shape_triangle_bounds(sd)
    st = G.bvh_check(b)
    print(scene, st)
    assert st["leaves"] > 0 and st["wide_stack_need"] <= 32
    assert 0 < st["bvh8_nodes"] <= st["wide_nodes"]


def soup(rng, n, long_ones):
    c = rng.uniform(-10, 10, (n, 1, 3))
    t = c + rng.normal(0, 0.3, (n, 3, 3))
    t[:long_ones] = rng.uniform(-10, 10, (long_ones, 3, 3))       # long diagonal triangles: what spatial splits are for
    return t.astype(np.float32)


@pytest.mark.parametrize("n, budget", [(1, 0.3), (2, 1.0), (7, 1.0), (300, 0.0), (300, 0.3), (5000, 0.3), (5000, 2.0)])
def test_spatial_split_build_covers_every_triangle(n, budget):
    """host/sbvh.cpp (knob `sbvh`): gdpt_sbvh_check builds the tree with spatial splits and verifies nesting, the reference budget,
    the depth / stack bounds and, by point sampling, that every part of every triangle lies in a leaf box that references it."""
    st = G.sbvh_check(soup(np.random.default_rng(n), n, min(n, 40)), budget=budget, samples_per_tri=24)
    assert n <= st["references"] <= (1 + budget) * n + 1
    assert st["bvh2_depth"] <= 32 and st["wide_stack_need"] <= 32
    if budget == 0.0:
        assert st["references"] == n
    if n >= 5000:
        assert st["references"] > n         # the long triangles do get split


def test_spatial_split_build_degenerate_inputs():
    # zero-area and axis-aligned triangles, duplicates, a triangle in a plane of a split candidate
    t = np.zeros((64, 3, 3), np.float32)
    t[:16, 1, 0] = 1.0                                                        # degenerate: two vertices coincide per row
    t[16:32] = np.array([[0, 0, 0], [4, 0, 0], [0, 4, 0]], np.float32)        # 16 copies of one triangle
    t[32:48] = np.array([[0, 0, 1], [8, 0, 1], [0, 0.001, 1]], np.float32)    # slivers
    t[48:] = np.array([[2, 2, -3], [2, 2, 3], [2, -2, 0]], np.float32)        # in the plane x = 2
    st = G.sbvh_check(t, budget=1.0, samples_per_tri=24)
    assert st["references"] >= 64 and st["bvh2_depth"] <= 32


@pytest.mark.parametrize("scene", ["sponza/sponza.xml", "disney_bsdf_test/disney_bsdf.xml"])
def test_spatial_split_build_on_scene_geometry(scene):
    sd = G.parse_scene(os.path.join(ROOT, "scenes", scene))
    st = G.sbvh_check(G.shape_triangles(sd), budget=0.3, samples_per_tri=10)
    print(scene, st)
    assert st["references"] > 0 and st["wide_stack_need"] <= 32
