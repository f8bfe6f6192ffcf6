"""Dump the reference map parsers' OUTPUT (reference-path polylines) as data assets.

Runs only in the build container (needs /root/reference).  The product never
imports the reference parsers; it consumes these ``.npz`` tables
(``sigmarl_amd/assets/maps/<scenario_type>.npz``).  Writing an own parser for
``cpm.xml`` / ``*.osm`` is SURVEY.md section 8(f) rank 2 ("next").

Per scenario type the table holds every reference path of
``MapManager(...).parser`` (reference ``sigmarl/map_manager.py:13-40``):
list 0 = ``reference_paths`` and, for the CPM map, lists 1..3 =
``reference_paths_intersection / _merge_in / _merge_out``
(``sigmarl/parse_xml.py:579-603``), flattened in that order.
"""
from __future__ import annotations

import contextlib
import io
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import refshim  # noqa: E402

refshim.install()

from sigmarl.constants import SCENARIOS  # noqa: E402
from sigmarl.map_manager import MapManager  # noqa: E402

OUT = os.path.abspath(os.path.join(HERE, "..", "..", "..", "sigmarl_amd", "assets", "maps"))


def dump(scenario_type: str) -> None:
    sc = SCENARIOS[scenario_type]
    # ScenarioRoadTraffic builds MapManager(lane_width=parameters.lane_width) (road_traffic.py:463-467); for the OSM maps the
    # parser offsets the boundaries (and shifts all coordinates) by that width (parse_osm.py:97-98,300-304), so the table is
    # the one of Parameters' default lane_width = 0.25 (helper_common.py:119).  The CPM parser ignores it.
    parser_lane_width = 0.25
    with contextlib.redirect_stdout(io.StringIO()):
        m = MapManager(scenario_type=scenario_type, device="cpu", lane_width=parser_lane_width)
    p = m.parser
    lists = [p.reference_paths, p.reference_paths_intersection, p.reference_paths_merge_in, p.reference_paths_merge_out]
    paths, list_id, local_id = [], [], []
    for li, lst in enumerate(lists):
        for k, rp in enumerate(lst):
            paths.append(rp)
            list_id.append(li)
            local_id.append(k)
    n = len(paths)
    max_c = max(r["center_line"].shape[0] for r in paths)
    max_l = max(r["left_boundary_shared"].shape[0] for r in paths)
    max_r = max(r["right_boundary_shared"].shape[0] for r in paths)
    max_ids = max(len(r["lanelet_IDs"]) for r in paths)
    center = np.zeros((n, max_c, 2), np.float32)
    yaw = np.zeros((n, max_c), np.float32)
    left = np.zeros((n, max_l, 2), np.float32)
    right = np.zeros((n, max_r, 2), np.float32)
    n_center = np.zeros(n, np.int32)
    n_yaw = np.zeros(n, np.int32)
    n_left = np.zeros(n, np.int32)
    n_right = np.zeros(n, np.int32)
    is_loop = np.zeros(n, np.uint8)
    lanelet_ids = np.zeros((n, max_ids), np.int32)
    n_lanelet_ids = np.zeros(n, np.int32)
    for i, r in enumerate(paths):
        c = r["center_line"].numpy().astype(np.float32)
        l = r["left_boundary_shared"].numpy().astype(np.float32)
        rr = r["right_boundary_shared"].numpy().astype(np.float32)
        y = r["center_line_yaw"].numpy().astype(np.float32).reshape(-1)
        n_center[i], n_left[i], n_right[i], n_yaw[i] = len(c), len(l), len(rr), len(y)
        center[i, : len(c)] = c
        left[i, : len(l)] = l
        right[i, : len(rr)] = rr
        yaw[i, : len(y)] = y
        is_loop[i] = bool(r["is_loop"])
        ids = list(r["lanelet_IDs"])
        lanelet_ids[i, : len(ids)] = ids
        n_lanelet_ids[i] = len(ids)
    os.makedirs(OUT, exist_ok=True)
    np.savez_compressed(
        os.path.join(OUT, f"{scenario_type}.npz"),
        center=center, n_center=n_center, yaw=yaw, n_yaw=n_yaw,
        left=left, n_left=n_left, right=right, n_right=n_right,
        is_loop=is_loop, lanelet_ids=lanelet_ids, n_lanelet_ids=n_lanelet_ids,
        list_id=np.asarray(list_id, np.int32), local_id=np.asarray(local_id, np.int32),
        world_x_dim=np.float64(p.bounds["world_x_dim"]), world_y_dim=np.float64(p.bounds["world_y_dim"]),
        lane_width=np.float64(sc["lane_width"]), parser_lane_width=np.float64(parser_lane_width), default_n_agents=np.int32(sc.get("n_agents", 4)),
        n_lanelets_all=np.int32(len(p.lanelets_all)),
    )
    print(scenario_type, n, "paths", max_c, max_l, max_r)


if __name__ == "__main__":
    for k in SCENARIOS:
        dump(k)
