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
