"""Extract the INPUT side of the reference's map parsers as data: the node / way lists of every ``*.osm`` map and the scenario
specifications (lane width, scale, reference-path lanelet id lists) of ``sigmarl/constants.py:SCENARIOS``.

Runs only in the build container (needs /root/reference).  Output:
  sigmarl_amd/assets/maps/src/<scenario>.npz   nodes (id, lat, lon) in file order; ways (id, `lanes` tag, node refs)
  sigmarl_amd/assets/maps/scenarios.json       per scenario: map file name, lane_width, scale, n_agents, reference_paths_ids
``sigmarl_amd.mapc`` compiles these into the reference-path tables; ``tests/test_mapc.py`` holds the result against the tables the
reference's own parsers produced (``sigmarl_amd/assets/maps/<scenario>.npz``, written by gen_maps.py).
"""
from __future__ import annotations

import json
import os
import sys
import xml.etree.ElementTree as ET

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import refshim  # noqa: E402

refshim.install()
from sigmarl.constants import SCENARIOS  # noqa: E402

REF_MAPS = "/root/reference/sigmarl/scenarios/assets/maps"
OUT = os.path.abspath(os.path.join(HERE, "..", "..", "..", "sigmarl_amd", "assets", "maps"))


def main():
    specs = {}
    for name, sc in SCENARIOS.items():
        spec = {"map_path": sc["map_path"], "lane_width": sc["lane_width"], "scale": sc["scale"], "n_agents": sc["n_agents"]}
        for k in ("x_dim_min", "x_dim_max", "y_dim_min", "y_dim_max"):
            if k in sc:
                spec[k] = sc[k]
        if "reference_paths_ids" in sc:
            spec["reference_paths_ids"] = [[int(x) for x in path] for path in sc["reference_paths_ids"]]
        if "neighboring_lanelet_ids" in sc:  # constants.py: per lanelet id (1-based keys) the ids it may see (parse_osm.py:256-262 turns them into 0-based index lists)
            nb = sc["neighboring_lanelet_ids"]
            spec["neighboring_lanelet_ids"] = [[int(x) for x in nb[str(i + 1)]] for i in range(max(int(k) for k in nb))]
        specs[name] = spec
        if not sc["map_path"].endswith(".osm"):
            continue
        root = ET.parse(os.path.join(REF_MAPS, sc["map_path"])).getroot()
        node_id, latlon = [], []
        for node in root.findall("node"):
            node_id.append(int(node.get("id")))
            latlon.append((float(node.get("lat")), float(node.get("lon"))))
        way_id, way_lanes, way_off, way_nodes = [], [], [0], []
        for way in root.findall("way"):
            way_id.append(int(way.get("id")))
            tag = way.find("tag[@k='lanes']")
            way_lanes.append(int(tag.get("v")) if tag is not None else -1)
            way_nodes.extend(int(nd.get("ref")) for nd in way.findall("nd"))
            way_off.append(len(way_nodes))
        np.savez_compressed(os.path.join(OUT, "src", name + ".npz"), node_id=np.array(node_id, np.int64), node_latlon=np.array(latlon, np.float64),
                            way_id=np.array(way_id, np.int64), way_lanes=np.array(way_lanes, np.int32), way_off=np.array(way_off, np.int64),
                            way_nodes=np.array(way_nodes, np.int64))
    with open(os.path.join(OUT, "scenarios.json"), "w") as f:
        json.dump(specs, f, indent=1, sort_keys=True)
    print("wrote", len(specs), "scenario specs")


if __name__ == "__main__":
    main()


# Lanelets of the CPM map that share their outer boundaries (adjacent lanes of one road; first entry = leftmost lane, last entry =
# rightmost lane): map metadata of the reference's CPM parser (parse_xml.py:417-466), needed for the "shared" boundaries.
CPM_SHARED_BOUNDARY_GROUPS = [
    [4, 3, 22], [6, 5, 23], [8, 7], [60, 59], [58, 57, 75], [56, 55, 74], [54, 53], [80, 79], [82, 81, 100], [84, 83, 101], [86, 85],
    [34, 33], [32, 31, 49], [30, 29, 48], [28, 27], [2, 1],
    [13, 14], [15, 16], [9, 10], [11, 12], [63, 64], [61, 62], [67, 68], [65, 66], [91, 92], [93, 94], [87, 88], [89, 90],
    [37, 38], [35, 36], [41, 42], [39, 40],
    [25, 18], [26, 17], [52, 43], [72, 73], [51, 44], [50, 45], [102, 97], [20, 21], [103, 96], [104, 95], [78, 69], [46, 47],
    [77, 70], [76, 71], [24, 19], [98, 99],
]


def cpm_sources():
    """The CPM map (CommonRoad XML): left / right bound points of every lanelet as parsed (float64), plus the lanelet sequences of the
    72 reference paths as the reference's parser lists them (read back from its OUTPUT, assets/maps/cpm_entire.npz: `lanelet_IDs`)."""
    root = ET.parse(os.path.join(REF_MAPS, "cpm.xml")).getroot()
    ids, off_l, off_r, pl, pr = [], [0], [0], [], []
    for child in root:
        if child.tag != "lanelet":
            continue
        ids.append(int(child.get("id")))
        for tag, pts, off in (("leftBound", pl, off_l), ("rightBound", pr, off_r)):
            b = child.find(tag)
            for point in b.findall("point"):
                pts.append((float(point.find("x").text), float(point.find("y").text)))
            off.append(len(pts))
    np.savez_compressed(os.path.join(OUT, "src", "cpm.npz"), lanelet_id=np.array(ids, np.int32), left_off=np.array(off_l, np.int64),
                        right_off=np.array(off_r, np.int64), left=np.array(pl, np.float64), right=np.array(pr, np.float64))
    tab = np.load(os.path.join(OUT, "cpm_entire.npz"))
    paths = [[int(v) for v in tab["lanelet_ids"][i, : tab["n_lanelet_ids"][i]]] for i in range(tab["lanelet_ids"].shape[0])]
    with open(os.path.join(OUT, "scenarios.json")) as f:
        specs = json.load(f)
    for name in ("cpm_entire", "cpm_mixed"):
        specs[name]["paths"] = paths
        specs[name]["list_id"] = [int(v) for v in tab["list_id"]]
        specs[name]["shared_boundary_groups"] = CPM_SHARED_BOUNDARY_GROUPS
    with open(os.path.join(OUT, "scenarios.json"), "w") as f:
        json.dump(specs, f, indent=1, sort_keys=True)
    print("cpm:", len(ids), "lanelets,", len(paths), "paths")


if __name__ == "__main__":
    cpm_sources()
