"""Map compiler (sigmarl_amd/mapc.py) against the tables the reference's own parser produced, and on a hand-made map."""
import os

import numpy as np
import pytest

from sigmarl_amd import mapc
from sigmarl_amd.maps import MapTable

OSM_SCENARIOS = sorted(k for k, v in mapc.scenario_specs().items() if v["map_path"].endswith(".osm"))
ALL_SCENARIOS = sorted(mapc.scenario_specs())
EXACT = ("center", "left", "right", "n_center", "n_left", "n_right", "n_yaw", "is_loop", "lanelet_ids", "n_lanelet_ids", "list_id", "local_id")


def _ulp_diff(a, b):
    return np.abs(a.view(np.int32).astype(np.int64) - b.view(np.int32).astype(np.int64))


@pytest.mark.parametrize("scen", ALL_SCENARIOS)
def test_compiled_table_equals_reference_parser_output(scen):
    """Polylines (incl. the CPM map's smoothed shared boundaries), counts, loop flags, lanelet ids, path lists and world size
    bit-identical to ParseOSM's / ParseXML's output (assets/maps/<scen>.npz, written by tests/golden/gen/gen_maps.py from the
    reference); yaw within 1 ulp (correctly rounded atan2 here, SLEEF in the reference)."""
    out = mapc.compile_scenario(scen, lane_width=0.25)
    ref = np.load(os.path.join(os.path.dirname(mapc.__file__), "assets", "maps", scen + ".npz"))
    for k in EXACT:
        assert out[k].shape == ref[k].shape and np.array_equal(out[k], ref[k]), k
    assert _ulp_diff(out["yaw"], ref["yaw"]).max() <= 1
    assert float(out["world_x_dim"]) == float(ref["world_x_dim"]) and float(out["world_y_dim"]) == float(ref["world_y_dim"])
    assert float(out["lane_width"]) == float(ref["lane_width"]) and int(out["default_n_agents"]) == int(ref["default_n_agents"])
    assert int(out["n_lanelets_all"]) == int(ref["n_lanelets_all"]) if "n_lanelets_all" in out else True
    mt = MapTable(scen, table=out)   # and it is accepted where the shipped table is
    assert mt.n_paths == ref["center"].shape[0] and sum(mt.list_count.values()) == mt.n_paths


@pytest.mark.skipif(not os.path.isdir("/root/reference/sigmarl/scenarios/assets/maps"), reason="raw .osm files only exist next to the reference")
@pytest.mark.parametrize("scen", OSM_SCENARIOS[:4])
def test_osm_reader_matches_the_extracted_sources(scen):
    spec = mapc.scenario_specs()[scen]
    raw = mapc.read_osm(os.path.join("/root/reference/sigmarl/scenarios/assets/maps", spec["map_path"]))
    src = mapc.load_source(scen)
    for f in ("node_id", "node_latlon", "way_lanes", "way_off", "way_nodes"):
        assert np.array_equal(getattr(raw, f), getattr(src, f)), f


OSM_TEXT = """<?xml version='1.0' encoding='UTF-8'?>
<osm version='0.6'>
  <node id='-1' lat='0.00000' lon='0.00000' />
  <node id='-2' lat='0.00002' lon='0.00000' />
  <node id='-3' lat='0.00004' lon='0.00000' />
  <node id='-4' lat='0.00004' lon='0.00003' />
  <node id='-5' lat='0.00009' lon='0.00009' />
  <way id='-10'><nd ref='-1' /><nd ref='-2' /><nd ref='-3' /><tag k='lanes' v='1' /></way>
  <way id='-11'><nd ref='-3' /><nd ref='-4' /><tag k='lanes' v='2' /></way>
  <way id='-12'><nd ref='-4' /><nd ref='-5' /></way>
</osm>
"""


def test_hand_made_map(tmp_path):
    """A two-lanelet L-shaped road written as an .osm file: the corner node is shared, the untagged way is ignored, boundaries sit half a
    lane width to the left / right of the direction of travel, the last point reuses the last normal."""
    p = tmp_path / "toy.osm"
    p.write_text(OSM_TEXT)
    src = mapc.read_osm(str(p))
    assert list(src.way_lanes) == [1, 2, -1]
    w, scale = 0.25, 1e5
    t = mapc.compile_osm(src, [[1, 2], [1]], lane_width=w, scale=scale)
    assert list(t["n_center"]) == [4, 3] and list(t["is_loop"]) == [0, 0] and list(t["lanelet_ids"][0]) == [0, 1]
    c = t["center"][0, :4]
    m = w * 1.2
    np.testing.assert_allclose(c, np.array([[0, 0], [2, 0], [4, 0], [4, 3]], np.float64) + m, rtol=0, atol=1e-5)
    # travelling in +x: left is +y; travelling in +y (last segment and last point): left is -x
    np.testing.assert_allclose(t["left"][0, :2] - c[:2], [[0, w / 2]] * 2, atol=1e-6)
    np.testing.assert_allclose(t["right"][0, :2] - c[:2], [[0, -w / 2]] * 2, atol=1e-6)
    np.testing.assert_allclose(t["left"][0, 2:4] - c[2:4], [[-w / 2, 0]] * 2, atol=1e-6)
    np.testing.assert_allclose(t["yaw"][0, :3], [0, 0, np.pi / 2], atol=1e-6)
    assert abs(float(t["world_x_dim"]) - ((4 + m + w / 2) + m)) < 1e-4  # max x (right boundary of the vertical leg) + min x


@pytest.mark.skipif(not os.path.exists("/root/reference/sigmarl/scenarios/assets/maps/cpm.xml"), reason="cpm.xml only exists next to the reference")
def test_commonroad_reader_matches_the_extracted_source():
    raw = mapc.read_commonroad_xml("/root/reference/sigmarl/scenarios/assets/maps/cpm.xml")
    src = mapc.load_lanelet_source("cpm")
    for f in ("lanelet_id", "left_off", "right_off", "left", "right"):
        assert np.array_equal(getattr(raw, f), getattr(src, f)), f


def test_lanelet_tables_match_the_reference_parser():
    """The tables behind the lanelet-relation mask (map_manager.py:41-118): `parser.lanelets_all` centre lines stacked and zero-padded as
    `determine_current_lanelet` does, and `parser.neighboring_lanelets_idx` -- compiled by sigmarl_amd.mapc from the shipped map sources, bit-identical to
    the reference parser's for all 16 OSM scenarios (tests/golden/lanelets.npz); and the reference's own lanelet lookup on seeded positions (incl. one
    next to the origin, where the zero padding sits) is reproduced by the restated formula the oracle and the HIP kernel implement."""
    from sigmarl_amd.maps import load_map

    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "lanelets.npz"))
    assert len(z["names"]) == 16
    for name in (str(n) for n in z["names"]):
        t = mapc.compile_scenario(name)
        assert np.array_equal(t["lanelet_centers"], z[name + "_centers"]), name
        nb = z[name + "_neighbors"]
        masks = np.zeros(len(nb), np.uint64)
        for i, row in enumerate(nb):
            for n in row[row >= 0]:
                masks[i] |= np.uint64(1) << np.uint64(int(n))
        assert np.array_equal(t["lanelet_neighbors"][: len(nb)], masks), name  # (interchange_3 lists 20 of its 22 lanelets: the rest see nobody here, raise there)
        c, pos = z[name + "_centers"], z[name + "_pos"]
        d = (pos[:, :, None, None, :] - c[None, None]) ** 2
        got = (d[..., 0] + d[..., 1]).astype(np.float32).min(3).argmin(2)
        assert np.array_equal(got, z[name + "_lanelet"]), name
        if name != "pseudo_distance_example":
            centers, neigh = load_map(name).lanelet_tables()
            assert np.array_equal(centers, t["lanelet_centers"]) and np.array_equal(neigh, t["lanelet_neighbors"])
    assert load_map("cpm_entire").lanelet_tables() is None  # the CPM parser has no neighbour table: the mask by lanelets is empty there
