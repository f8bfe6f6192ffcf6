"""Map compiler: lanelet maps -> the reference-path tables the environment step consumes.

Replaces the reference's ``ParseOSM`` (``sigmarl/parse_osm.py:37-336``, the 16 JOSM ``*.osm`` scenarios) and ``ParseXML``
(``sigmarl/parse_xml.py:24-907``, the CPM lab's CommonRoad map) + the part of ``MapManager`` the scenario uses
(``sigmarl/map_manager.py:13-40``): the product no longer needs the reference's parser OUTPUT for a map, only the map itself (the file,
or the point lists extracted from one) and the scenario's specification (lane width, scale, lanelet id lists of the reference paths,
groups of lanelets sharing their outer boundaries -- ``sigmarl/constants.py:SCENARIOS`` and the tables in ``parse_xml.py``, shipped as
``assets/maps/scenarios.json``).

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


def lanelet_tables(lanelets: Dict[int, np.ndarray], neighboring_lanelet_ids: Optional[List[List[int]]]) -> dict:
    """What ``MapManager.determine_current_lanelet`` / ``determine_masked_agents_by_lanelets`` (map_manager.py:41-118) work on: the centre lines of
    ``lanelets_all`` (lanelet ids 1 .. max in id order, parse_osm.py:166-171) stacked into ``[L, max_len, 2]`` and ZERO-padded -- the reference pads
    with ``torch.nn.functional.pad``, so the padded entries are real candidate points at the origin --, and ``neighboring_lanelets_idx``
    (parse_osm.py:256-262: 0-based index lists) as one bit mask per lanelet (bit j: lanelet j is visible from this lanelet)."""
    L = max(lanelets) if lanelets else 0
    ml = max((len(c) for c in lanelets.values()), default=1)
    centers = np.zeros((L, ml, 2), f32)
    n_pts = np.zeros(L, np.int32)
    for lid, c in lanelets.items():
        centers[lid - 1, : len(c)] = c
        n_pts[lid - 1] = len(c)
    masks = np.zeros(L, np.uint64)
    if neighboring_lanelet_ids:
        if len(neighboring_lanelet_ids) > 64 or L > 64:
            raise ValueError("more than 64 lanelets: the neighbour masks are 64-bit")
        for i, ids in enumerate(neighboring_lanelet_ids):
            for n in ids:
                masks[i] |= np.uint64(1) << np.uint64(int(n) - 1)
    return {"lanelet_centers": centers, "n_lanelet_points": n_pts, "lanelet_neighbors": masks, "has_lanelet_neighbors": np.int32(bool(neighboring_lanelet_ids))}


def compile_osm(src: MapSource, reference_paths_ids: List[List[int]], lane_width: float, scale: float, neighboring_lanelet_ids: Optional[List[List[int]]] = None) -> dict:
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
    out = _pack(paths, lane_width)
    out.update(lanelet_tables(lanelets, neighboring_lanelet_ids))
    return out


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


# ---- CPM map (CommonRoad XML lanelets), parse_xml.py ---------------------------------------------------------------------------------
@dataclass
class LaneletSource:
    """Left / right bound points of every lanelet of a CommonRoad XML map as parsed (float64); lanelet ids start at 1."""
    lanelet_id: np.ndarray  # [L] int32
    left_off: np.ndarray    # [L + 1]
    right_off: np.ndarray   # [L + 1]
    left: np.ndarray        # [*, 2] float64
    right: np.ndarray       # [*, 2] float64


def read_commonroad_xml(path: str) -> LaneletSource:
    root = ET.parse(path).getroot()
    ids, off_l, off_r, pl, pr = [], [0], [0], [], []
    for child in root:
        if child.tag != "lanelet":
            continue
        ids.append(int(child.get("id")))
        for tag, pts, off in (("leftBound", pl, off_l), ("rightBound", pr, off_r)):
            for point in child.find(tag).findall("point"):
                pts.append((float(point.find("x").text), float(point.find("y").text)))
            off.append(len(pts))
    return LaneletSource(np.array(ids, np.int32), np.array(off_l, np.int64), np.array(off_r, np.int64), np.array(pl, np.float64).reshape(-1, 2),
                         np.array(pr, np.float64).reshape(-1, 2))


def load_lanelet_source(name: str = "cpm") -> LaneletSource:
    d = np.load(os.path.join(_ASSETS, "src", name + ".npz"))
    return LaneletSource(d["lanelet_id"], d["left_off"], d["right_off"], d["left"], d["right"])


def _dist2(a: np.ndarray, b: np.ndarray) -> np.float32:
    d = (a - b).astype(f32)
    return _norm2(d[0], d[1])


def _linspace01(steps: int) -> np.ndarray:
    """``torch.linspace(0, 1, steps)`` in fp32: the first half counts up from the start, the second half down from the end."""
    step = f32(f32(1) - f32(0)) / f32(steps - 1)
    half = steps // 2
    return np.array([f32(0) + step * f32(i) if i < half else f32(1) - step * f32(steps - 1 - i) for i in range(steps)], f32)


def _lerp_block(start: np.ndarray, end: np.ndarray, overlap: int) -> np.ndarray:
    t = _linspace01(2 * overlap)[:, None]
    return ((f32(1) - t).astype(f32) * start[None, :]).astype(f32) + (t * end[None, :]).astype(f32)  # (1 - t) * start + t * end


def _smooth_concatenate(a: np.ndarray, b: np.ndarray, overlap: int = 4) -> np.ndarray:
    """``ParseXML.smooth_concatenate`` (parse_xml.py:832-871): the last `overlap` points of a and the first `overlap` points of b are
    replaced by a straight interpolation between a[-overlap] and b[overlap - 1]."""
    return np.concatenate([a[:-overlap], _lerp_block(a[-overlap], b[overlap - 1], overlap).astype(f32), b[overlap:]], axis=0)


def _smooth_loop_boundary(bd: np.ndarray, overlap: int = 4) -> np.ndarray:
    """``ParseXML.smooth_loop_boundary`` (parse_xml.py:873-907), incl. the closing point appended at the end."""
    inter = _lerp_block(bd[-overlap], bd[overlap - 1], overlap).astype(f32)
    out = bd.copy()
    out[:overlap] = inter[overlap:]
    out[-overlap:] = inter[:overlap]
    return np.concatenate([out, out[:1]], axis=0)


def compile_lanelet_paths(src: LaneletSource, paths: List[List[int]], list_id: List[int], shared_groups: List[List[int]]) -> dict:
    """Reference paths of a CommonRoad lanelet map (``ParseXML._get_reference_path``, parse_xml.py:605-797): per path the lanelets'
    own boundaries are chained (a repeated connection point is dropped), the centre line is their mean, and the SHARED boundaries --
    left bound of the leftmost and right bound of the rightmost lanelet of the road the lanelet belongs to -- are chained with a
    smoothed transition where they do not connect, and smoothed across the seam of a loop."""
    lane = {}
    for k, lid in enumerate(src.lanelet_id):
        lane[int(lid)] = (src.left[src.left_off[k]: src.left_off[k + 1]].astype(f32), src.right[src.right_off[k]: src.right_off[k + 1]].astype(f32))
    TH = f32(1e-4)
    out_paths = []
    for ids in paths:
        lb = rb = lbs = rbs = None
        for lid in ids:
            grp = next(g for g in shared_groups if lid in g)
            l, r = lane[lid]
            ls, rs = lane[grp[0]][0], lane[grp[-1]][1]
            if lb is None:
                lb, rb, lbs, rbs = l, r, ls, rs
                continue
            lb = np.concatenate([lb, l[1:] if _dist2(lb[-1], l[0]) < TH else l], axis=0)
            lbs = np.concatenate([lbs, ls[1:]], axis=0) if _dist2(lbs[-1], ls[0]) < TH else _smooth_concatenate(lbs, ls)
            rb = np.concatenate([rb, r[1:] if _dist2(rb[-1], r[0]) < TH else r], axis=0)
            rbs = np.concatenate([rbs, rs[1:]], axis=0) if _dist2(rbs[-1], rs[0]) < TH else _smooth_concatenate(rbs, rs)
        center = ((lb + rb).astype(f32) / f32(2)).astype(f32)
        is_loop = bool(_dist2(center[0], center[-1]) <= TH)
        if is_loop:
            if _dist2(lbs[0], lbs[-1]) > f32(0.1):
                lbs = _smooth_loop_boundary(lbs)
            if _dist2(rbs[0], rbs[-1]) > f32(0.1):
                rbs = _smooth_loop_boundary(rbs)
        out_paths.append({"center": center, "yaw": _yaw(center), "left": lbs.astype(f32), "right": rbs.astype(f32), "is_loop": is_loop,
                          "lanelet_ids": list(ids)})
    out = _pack(out_paths, 0.0)
    out["list_id"] = np.array(list_id, np.int32)
    local = np.zeros(len(paths), np.int32)
    seen: Dict[int, int] = {}
    for i, li in enumerate(list_id):
        local[i] = seen.get(li, 0)
        seen[li] = local[i] + 1
    out["local_id"] = local
    return out


def compile_scenario(scenario_type: str, lane_width: float = 0.25, osm_path: Optional[str] = None) -> dict:
    """Compile one of the named scenarios (``assets/maps/scenarios.json``) from its shipped node / way lists, or from ``osm_path``."""
    spec = scenario_specs()[scenario_type]
    if not spec["map_path"].endswith(".osm"):  # the CPM map: CommonRoad XML; the world size is fixed by the lab (parse_xml.py:51-58)
        src = read_commonroad_xml(osm_path) if osm_path else load_lanelet_source("cpm")
        out = compile_lanelet_paths(src, spec["paths"], spec["list_id"], spec["shared_boundary_groups"])
        out["world_x_dim"] = np.float64(spec["x_dim_min"] + spec["x_dim_max"])
        out["world_y_dim"] = np.float64(spec["y_dim_min"] + spec["y_dim_max"])
        out["parser_lane_width"] = np.float64(lane_width)
        out["lane_width"] = np.float64(spec["lane_width"])
        out["default_n_agents"] = np.int32(spec["n_agents"])
        out["n_lanelets_all"] = np.int32(len(src.lanelet_id))
        return out
    src = read_osm(osm_path) if osm_path else load_source(scenario_type)
    out = compile_osm(src, spec["reference_paths_ids"], lane_width, float(spec["scale"]), spec.get("neighboring_lanelet_ids"))
    out["lane_width"] = np.float64(spec["lane_width"])
    out["default_n_agents"] = np.int32(spec["n_agents"])
    return out
