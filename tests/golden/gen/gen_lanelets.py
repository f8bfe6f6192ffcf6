"""Golden for the lanelet tables behind the lanelet-relation mask (map_manager.py:41-118): for every OSM scenario of the reference, what its own
parser holds -- ``parser.lanelets_all[*]["center_line"]`` stacked and zero-padded exactly as ``MapManager.determine_current_lanelet`` does, and
``parser.neighboring_lanelets_idx`` (ragged -> padded with -1) -- plus, for seeded positions, the lanelet that method assigns.
Runs only in the build container (needs /root/reference).  Output: tests/golden/lanelets.npz (held to ``sigmarl_amd.mapc`` by tests/test_mapc.py)."""
from __future__ import annotations

import contextlib
import io
import os
import sys

import numpy as np
import torch
from torch.nn.functional import pad

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import refshim  # noqa: E402

refshim.install()
from sigmarl.constants import SCENARIOS  # noqa: E402
from sigmarl.map_manager import MapManager  # noqa: E402

OUT = os.path.abspath(os.path.join(HERE, "..", "lanelets.npz"))


def main():
    out = {}
    names = []
    for name, sc in SCENARIOS.items():
        if not sc["map_path"].endswith(".osm"):
            continue
        with contextlib.redirect_stdout(io.StringIO()):
            m = MapManager(scenario_type=name, device="cpu", lane_width=0.25)
        p = m.parser
        ml = max(len(l["center_line"]) for l in p.lanelets_all)
        centers = torch.stack([pad(l["center_line"], (0, 0, 0, ml - len(l["center_line"]))) for l in p.lanelets_all])  # map_manager.py:53-66
        nb = p.neighboring_lanelets_idx
        mn = max(len(x) for x in nb)
        nbp = np.full((len(nb), mn), -1, np.int32)
        for i, x in enumerate(nb):
            nbp[i, : len(x)] = x
        # the reference's own lookup on seeded positions spread over (and a little beyond) the map, incl. the origin the padding sits on
        g = torch.Generator().manual_seed(len(name))
        lo = centers.reshape(-1, 2)[centers.reshape(-1, 2).abs().sum(1) > 0].min(0).values - 0.3
        hi = centers.reshape(-1, 2).max(0).values + 0.3
        pos = lo + (hi - lo) * torch.rand((6, 40, 2), generator=g)
        pos[0, 0] = torch.tensor([0.01, 0.02])
        m.determine_current_lanelet(pos)
        out[name + "_centers"] = centers.numpy()
        out[name + "_neighbors"] = nbp
        out[name + "_pos"] = pos.numpy()
        out[name + "_lanelet"] = m.current_lanelet_idx.squeeze(2).numpy().astype(np.int32)
        names.append(name)
    out["names"] = np.asarray(names)
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, len(names), "scenarios")


if __name__ == "__main__":
    main()
