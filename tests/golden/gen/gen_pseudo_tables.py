"""Dump the per-segment tables of the reference's pseudo-distance (map-only quantities) as data assets.

Runs only in the build container (needs /root/reference).  ``PseudoDistance.get_pseudo_distance``
(``sigmarl/pseudo_distance.py:123-202``) recomputes, on every call, for every segment of a boundary polyline:
the rotation into the segment frame (``atan2`` -> ``cos``/``sin``, :43-56), the slopes of the two pseudo tangent
vectors in that frame (:94-103) and the segment length (:177).  None of these depend on the query point.  The
result is rounded to fp16 (:118) and then differentiated numerically (``cbf_qp.py:624-644``), so a one-ulp
difference in a table entry can move a margin by 1e-3: the tables are therefore produced by the reference's own
torch ops, run here, and shipped as data (``sigmarl_amd/assets/maps/pseudo/<scenario_type>.npz``), exactly like
the polylines of ``gen_maps.py``.  ``sigmarl_amd/cbf.py:segment_tables`` computes the same tables with numpy for
maps that have no asset (this script prints how many entries differ).

Layout: ``left[n_paths, max_l-1, 5]`` / ``right[n_paths, max_r-1, 5]`` float32 = (cos, sin, m_b, m_t, length),
paths in the order of ``<scenario_type>.npz`` (all four path lists, flattened).
"""
from __future__ import annotations

import contextlib
import io
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import refshim  # noqa: E402

refshim.install()

from sigmarl.constants import SCENARIOS  # noqa: E402
from sigmarl.map_manager import MapManager  # noqa: E402
from sigmarl.pseudo_distance import PseudoDistance  # noqa: E402
from sigmarl.helper_scenario import compute_pseudo_tangent_vector  # noqa: E402

ROOT = os.path.abspath(os.path.join(HERE, "..", "..", ".."))
OUT = os.path.join(ROOT, "sigmarl_amd", "assets", "maps", "pseudo")
sys.path.insert(0, ROOT)


def reference_tables(pd: PseudoDistance, poly: torch.Tensor, tang: torch.Tensor) -> np.ndarray:
    pv, tv = poly.unsqueeze(0), tang.unsqueeze(0)
    p_b, p_t = pv[:, :-1], pv[:, 1:]
    S = p_b.shape[1]
    e1 = torch.tensor([1.0, 0.0]).view(1, 1, 2).expand(1, S, 2)
    r = pd.transform_from_global_to_line_coordiante(e1, p_b, p_t, True)  # R @ (1, 0) = (cos, -sin): exact products
    c, s = r[0, :, 0, 0], -r[0, :, 1, 0]
    tb = pd.transform_from_global_to_line_coordiante(tv[:, :-1], p_b, p_t, True)[0, :, :, 0]
    tt = pd.transform_from_global_to_line_coordiante(tv[:, 1:], p_b, p_t, True)[0, :, :, 0]
    m_b = torch.where(tb[:, 0] != 0, tb[:, 1] / tb[:, 0], torch.full_like(tb[:, 0], 1e-8))  # pseudo_distance.py:94-103
    m_t = torch.where(tt[:, 0] != 0, tt[:, 1] / tt[:, 0], torch.full_like(tt[:, 0], 1e-8))
    l = torch.norm(p_t - p_b, dim=-1)[0]  # :174-177
    return torch.stack([c, s, m_b, m_t, l], dim=-1).numpy().astype(np.float32)


def dump(scenario_type: str) -> None:
    from sigmarl_amd import cbf as own

    with contextlib.redirect_stdout(io.StringIO()):
        m = MapManager(scenario_type=scenario_type, device="cpu", lane_width=0.25)
    pd = PseudoDistance(scenario_type, m)
    p = m.parser
    paths = list(p.reference_paths) + list(p.reference_paths_intersection) + list(p.reference_paths_merge_in) + list(p.reference_paths_merge_out)
    n = len(paths)
    max_l = max(r["left_boundary_shared"].shape[0] for r in paths)
    max_r = max(r["right_boundary_shared"].shape[0] for r in paths)
    left = np.zeros((n, max_l - 1, 5), np.float32)
    right = np.zeros((n, max_r - 1, 5), np.float32)
    diff = tot = 0
    for i, r in enumerate(paths):
        for side, dst in (("left", left), ("right", right)):
            poly = r[f"{side}_boundary_shared"]
            # the parser attaches the vectors to the lists a scenario type uses (parse_map_base.py:102-157); same function otherwise
            tang = r.get(f"{side}_boundary_shared_pseudo_vector")
            if tang is None:
                tang = compute_pseudo_tangent_vector(poly)
            t = reference_tables(pd, poly, tang)
            dst[i, : len(t)] = t
            mine = own.segment_tables(poly.numpy().astype(np.float32))
            diff += int((mine.view(np.uint32) != t.view(np.uint32)).sum())
            tot += t.size
    os.makedirs(OUT, exist_ok=True)
    np.savez_compressed(os.path.join(OUT, f"{scenario_type}.npz"), left=left, right=right)
    print(scenario_type, n, "paths; own numpy tables differ in", diff, "of", tot, "entries")


if __name__ == "__main__":
    for k in (sys.argv[1:] or list(SCENARIOS)):
        dump(k)
