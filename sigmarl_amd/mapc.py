"""Map compiler: OSM lanelet maps -> the reference-path tables the environment step consumes.

Replaces, for the ``*.osm`` scenarios, the reference's ``ParseOSM`` (``sigmarl/parse_osm.py:37-336``) + the part of
``MapManager`` the scenario uses (``sigmarl/map_manager.py:13-40``): the product no longer needs the reference's parser OUTPUT
for such a map, only the map itself (an ``.osm`` file in JOSM's format, or the node / way lists extracted from one) and the
scenario's specification (lane width, scale, lanelet id lists of the reference paths -- ``sigmarl/constants.py:SCENARIOS``, shipped
as ``assets/maps/scenarios.json``).

The arithmetic follows the reference's tensor code one fp32 operation at a time, so the compiled polylines are BIT-IDENTICAL to the
tables the reference's parser produced (``tests/test_mapc.py`` checks every shipped OSM scenario); the centre-line yaw goes through
``atan2``, where the reference's SLEEF kernel is 1-ulp accurate and this module is correctly rounded: <= 1 ulp apart.

Pure host code (numpy); nothing here runs in the step.
"""
from __future__ import annotations

import json
import os
import xml.etree.ElementTree as ET
from dataclasses import dataclass
from typing import Dict, List, Optional

import numpy as np

_ASSETS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "assets", "maps")
f32 = np.float32


@dataclass
class MapSource:
    """Node / way lists of a JOSM ``.osm`` lanelet map (``parse_osm.py:67-107``): nodes in file order, ways with their ``lanes`` tag
    (the lanelet id; -1 when the way has no tag and is ignored)."""
    node_id: np.ndarray      # [n] int64
    node_latlon: np.ndarray  # [n, 2] float64
    way_lanes: np.ndarray    # [w] int32
    way_off: np.ndarray      # [w + 1] int64 offsets into way_nodes
    way_nodes: np.ndarray    # [*] int64 node refs


def read_osm(path: str) -> MapSource:
    """Parse an ``.osm`` file (nodes with lat/lon, ways with <nd ref> lists and an optional <tag k='lanes'>)."""
    root = ET.parse(path).getroot()
    node_id, latlon = [], []
    for node in root.findall("node"):
        node_id.append(int(node.get("id")))
        latlon.append((float(node.get("lat")), float(node.get("lon"))))
    lanes, off, refs = [], [0], []
    for way in root.findall("way"):
        tag = way.find("tag[@k='lanes']")
        lanes.append(int(tag.get("v")) if tag is not None else -1)
        refs.extend(int(nd.get("ref")) for nd in way.findall("nd"))
        off.append(len(refs))
    return MapSource(np.array(node_id, np.int64), np.array(latlon, np.float64).reshape(-1, 2), np.array(lanes, np.int32), np.array(off, np.int64),
                     np.array(refs, np.int64))


def load_source(scenario_type: str) -> MapSource:
    """The extracted node / way lists shipped for the reference's own OSM scenarios (``assets/maps/src``)."""
    d = np.load(os.path.join(_ASSETS, "src", scenario_type + ".npz"))
    return MapSource(d["node_id"], d["node_latlon"], d["way_lanes"], d["way_off"], d["way_nodes"])


def scenario_specs() -> Dict[str, dict]:
    with open(os.path.join(_ASSETS, "scenarios.json")) as f:
        return json.load(f)


# ---- fp32 arithmetic of the reference's tensor code ------------------------------------------------------------------------------
def _norm2(x: np.float32, y: np.float32) -> np.float32:
    """``torch.norm`` of a 2-vector on PyTorch-CPU: sqrt(fma(y, y, x * x)) (measured; the same form the step kernel uses)."""
    xx = f32(x * x)
    s = f32(np.float64(y) * np.float64(y) + np.float64(xx))  # exact product and sum in double, one rounding: the fused multiply-add
    return f32(np.sqrt(s))


def _boundaries(center: np.ndarray, width: float):
    """``ParseOSM._compute_boundaries`` (parse_osm.py:283-306): every centre-line point is offset along the normal of the segment that
    STARTS there; the last point reuses the last segment's normal."""
    n = len(center)
    left = np.zeros((n, 2), f32)
    right = np.zeros((n, 2), f32)
    w = f32(width)
    perp = np.zeros(2, f32)
    for i in range(n - 1):
        d = center[i + 1] - center[i]                      # fp32
        perp = np.array([-d[1], d[0]], f32)
        nrm = _norm2(perp[0], perp[1])
        if nrm != 0:
            perp = (perp / nrm).astype(f32)
        off = ((perp * w).astype(f32) / f32(2)).astype(f32)  # perp_direction * self._width / 2
        left[i] = center[i] + off
        right[i] = center[i] - off
    off = ((perp * w).astype(f32) / f32(2)).astype(f32)
    left[n - 1] = center[n - 1] + off
    right[n - 1] = center[n - 1] - off
    return left, right


def _yaw(center: np.ndarray) -> np.ndarray:
    """``_compute_center_line_info`` (parse_osm.py:264-281): atan2 of the fp32 difference vectors, correctly rounded."""
    v = np.diff(center, axis=0).astype(f32)
    return np.arctan2(v[:, 1].astype(np.float64), v[:, 0].astype(np.float64)).astype(f32)


def compile_osm(src: MapSource, reference_paths_ids: List[List[int]], lane_width: float, scale: float) -> dict:
    """The reference-path table of an OSM scenario, with the keys of ``assets/maps/<scenario>.npz``.

    ``lane_width`` is the width ``MapManager`` is constructed with -- the scenario passes ``Parameters.lane_width`` (0.25 by default,
    road_traffic.py:463-467), NOT the scenario's own ``lane_width`` entry; it offsets the boundaries and shifts every coordinate."""
    lat, lon = src.node_latlon[:, 0], src.node_latlon[:, 1]
    min_x, min_y = float(lat.min()), float(lon.min())       # parse_osm.py:83-91
    margin = lane_width * 1.2
    nodes = {int(i): ((float(a) - min_x) * scale + margin, (float(b) - min_y) * scale + margin) for i, a, b in zip(src.node_id, lat, lon)}  # :93-99
    lanelets: Dict[int, np.ndarray] = {}
    for w in range(len(src.way_lanes)):
        lid = int(src.way_lanes[w])
        if lid < 0:
            continue                                         # ways without a `lanes` tag are not lanelets (:118-164)
        refs = src.way_nodes[src.way_off[w]: src.way_off[w + 1]]
        lanelets[lid] = np.array([nodes[int(r)] for r in refs], np.float64).astype(f32)  # torch.tensor(points, dtype=float32)
    paths = []
    for ids in reference_paths_ids:                          # parse_osm.py:197-254
        is_loop = len(ids) > 1 and ids[0] == ids[-1]
        pts = []
        for k, lid in enumerate(ids):
            c = lanelets[int(lid)]
            pts.append(c if k == 0 else c[1:])               # the first node of a lanelet repeats the last node of its predecessor
        center = np.concatenate(pts, axis=0)
        if is_loop and len(center):
            center = center[:-1]                             # ... and a loop's last node repeats its first
        left, right = _boundaries(center, lane_width)
        paths.append({"center": center, "yaw": _yaw(center), "left": left, "right": right, "is_loop": is_loop, "lanelet_ids": [int(x) - 1 for x in ids]})
    return _pack(paths, lane_width)


def _pack(paths: List[dict], parser_lane_width: float) -> dict:
    n = len(paths)
    mc = max(len(p["center"]) for p in paths)
    ml = max(len(p["left"]) for p in paths)
    mr = max(len(p["right"]) for p in paths)
    mi = max(len(p["lanelet_ids"]) for p in paths)
    out = {
        "center": np.zeros((n, mc, 2), f32), "yaw": np.zeros((n, mc), f32), "left": np.zeros((n, ml, 2), f32), "right": np.zeros((n, mr, 2), f32),
        "n_center": np.zeros(n, np.int32), "n_yaw": np.zeros(n, np.int32), "n_left": np.zeros(n, np.int32), "n_right": np.zeros(n, np.int32),
        "is_loop": np.zeros(n, np.uint8), "lanelet_ids": np.zeros((n, mi), np.int32), "n_lanelet_ids": np.zeros(n, np.int32),
        "list_id": np.zeros(n, np.int32), "local_id": np.arange(n, dtype=np.int32),
    }
    xs, ys = [], []
    for i, p in enumerate(paths):
        c, l, r, y = p["center"], p["left"], p["right"], p["yaw"]
        out["n_center"][i], out["n_left"][i], out["n_right"][i], out["n_yaw"][i] = len(c), len(l), len(r), len(y)
        out["center"][i, : len(c)] = c
        out["left"][i, : len(l)] = l
        out["right"][i, : len(r)] = r
        out["yaw"][i, : len(y)] = y
        out["is_loop"][i] = p["is_loop"]
        out["lanelet_ids"][i, : len(p["lanelet_ids"])] = p["lanelet_ids"]
        out["n_lanelet_ids"][i] = len(p["lanelet_ids"])
        for a in (c, l, r):                                  # _get_map_dimension, parse_osm.py:308-336
            xs.extend(float(v) for v in a[:, 0])
            ys.extend(float(v) for v in a[:, 1])
    out["world_x_dim"] = np.float64(max(xs) + min(xs))
    out["world_y_dim"] = np.float64(max(ys) + min(ys))
    out["parser_lane_width"] = np.float64(parser_lane_width)
    return out


def compile_scenario(scenario_type: str, lane_width: float = 0.25, osm_path: Optional[str] = None) -> dict:
    """Compile one of the named scenarios (``assets/maps/scenarios.json``) from its shipped node / way lists, or from ``osm_path``."""
    spec = scenario_specs()[scenario_type]
    if not spec["map_path"].endswith(".osm"):
        raise NotImplementedError(f"{scenario_type}: only OSM lanelet maps are compiled here; the CPM map ships as a table (assets/maps/{scenario_type}.npz)")
    src = read_osm(osm_path) if osm_path else load_source(scenario_type)
    out = compile_osm(src, spec["reference_paths_ids"], lane_width, float(spec["scale"]))
    out["lane_width"] = np.float64(spec["lane_width"])
    out["default_n_agents"] = np.int32(spec["n_agents"])
    return out
