"""Reference-path tables (the map data the environment step consumes).

The tables under ``assets/maps/*.npz`` are the OUTPUT of the reference's map parsers
(``sigmarl/map_manager.py:13-40`` -> ``parse_xml.py`` / ``parse_osm.py``), dumped by
``tests/golden/gen/gen_maps.py``.  ``sigmarl_amd.mapc`` is the package's own compiler for ``*.osm`` lanelet maps (same polylines
bit for bit, ``tests/test_mapc.py``); pass its result as ``MapTable(name, table=...)``.  The CPM map (``cpm.xml``) ships as a table.

Path lists (``list_id``): 0 = ``parser.reference_paths``; for the CPM map 1/2/3 =
``reference_paths_intersection`` / ``_merge_in`` / ``_merge_out`` (``world_state_rt_sim.py:313-358`` selects by
``scenario_id``).  All lists are flattened into one table; the step addresses paths by their GLOBAL row.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import capi

ASSET_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "assets", "maps")


class MapTable:
    def __init__(self, scenario_type: str, table: dict | None = None):
        """``table``: a compiled table (``sigmarl_amd.mapc.compile_osm`` / ``compile_scenario``) instead of the shipped ``.npz``."""
        if table is not None:
            z = dict(table)
            z.setdefault("lane_width", z.get("parser_lane_width", 0.25))
            z.setdefault("default_n_agents", 1)
            z.setdefault("n_lanelets_all", int(np.max(z["lanelet_ids"])) + 1)
        else:
            path = os.path.join(ASSET_DIR, f"{scenario_type}.npz")
            if not os.path.exists(path):
                raise FileNotFoundError(f"no map table for scenario_type={scenario_type!r} ({path})")
            z = np.load(path)
        self.scenario_type = scenario_type
        self.name = scenario_type
        self.from_asset = table is None  # the shipped table (pseudo-distance segment assets exist for these only)
        self._table = dict(table) if table is not None else None
        self.n_paths = int(z["center"].shape[0])
        stride = max(z["center"].shape[1], z["left"].shape[1], z["right"].shape[1])
        self.stride = int(stride)

        def pad(a, width):
            out = np.zeros((a.shape[0], width) + a.shape[2:], a.dtype)
            out[:, : a.shape[1]] = a
            return np.ascontiguousarray(out)

        self.center = pad(z["center"].astype(np.float32), stride)
        self.left = pad(z["left"].astype(np.float32), stride)
        self.right = pad(z["right"].astype(np.float32), stride)
        self.yaw = pad(z["yaw"].astype(np.float32), stride)
        self.n_center = np.ascontiguousarray(z["n_center"].astype(np.int32))
        self.n_yaw = np.ascontiguousarray(z["n_yaw"].astype(np.int32))
        self.n_left = np.ascontiguousarray(z["n_left"].astype(np.int32))
        self.n_right = np.ascontiguousarray(z["n_right"].astype(np.int32))
        self.is_loop = np.ascontiguousarray(z["is_loop"].astype(np.uint8))
        self.lanelet_ids = z["lanelet_ids"].astype(np.int32)
        self.n_lanelet_ids = z["n_lanelet_ids"].astype(np.int32)
        self.list_id = z["list_id"].astype(np.int32)
        self.local_id = z["local_id"].astype(np.int32)
        self.world_x_dim = float(z["world_x_dim"])
        self.world_y_dim = float(z["world_y_dim"])
        self.lane_width = float(z["lane_width"])  # SCENARIOS[...]["lane_width"]: the normaliser
        self.parser_lane_width = float(z["parser_lane_width"])  # Parameters.lane_width the table was parsed with
        self.default_n_agents = int(z["default_n_agents"])
        self.n_lanelets_all = int(z["n_lanelets_all"])
        # first global row and length of every list
        self.list_first = {}
        self.list_count = {}
        for li in range(4):
            rows = np.nonzero(self.list_id == li)[0]
            self.list_first[li] = int(rows[0]) if len(rows) else 0
            self.list_count[li] = int(len(rows))

    def lanelet_tables(self):
        """(centers f32 [L, max_len, 2] zero-padded, neighbour bit masks u64 [L]) of the map's lanelets -- what the lanelet-relation mask of the
        bird-view observation works on (map_manager.py:41-118) -- or None when the map has no neighbour table (the CPM map: its parser leaves
        ``neighboring_lanelets_idx`` empty and the mask masks nobody, observation_provider_rt.py:647-665).  A compiled table carries them; for a
        shipped OSM table they are compiled from the shipped map source (``sigmarl_amd.mapc``)."""
        if getattr(self, "_lanelets", None) is None:
            t = self._table if self._table is not None and "lanelet_centers" in self._table else None
            if t is None and self.from_asset and "cpm" not in self.scenario_type:
                from . import mapc

                spec = mapc.scenario_specs().get(self.scenario_type)
                if spec is not None and spec["map_path"].endswith(".osm"):
                    t = mapc.compile_scenario(self.scenario_type, lane_width=self.parser_lane_width)
            if t is None or not int(t.get("has_lanelet_neighbors", 0)):
                self._lanelets = ()
            else:
                self._lanelets = (np.ascontiguousarray(t["lanelet_centers"], np.float32), np.ascontiguousarray(t["lanelet_neighbors"], np.uint64))
        return self._lanelets or None

    def global_path(self, scenario_id: int, path_id: int) -> int:
        """(scenario_id, list-local path_id) -> global row.  scenario_id 0 = all paths, 1..3 = CPM sub-scenarios."""
        return self.list_first[int(scenario_id)] + int(path_id)

    def as_struct(self) -> capi.Map:
        m = capi.Map()
        m.n_paths = self.n_paths
        m.stride_points = self.stride
        m.center = self.center.ctypes.data_as(C.c_void_p)
        m.left = self.left.ctypes.data_as(C.c_void_p)
        m.right = self.right.ctypes.data_as(C.c_void_p)
        m.yaw = self.yaw.ctypes.data_as(C.c_void_p)
        m.n_center = self.n_center.ctypes.data_as(C.c_void_p)
        m.n_left = self.n_left.ctypes.data_as(C.c_void_p)
        m.n_right = self.n_right.ctypes.data_as(C.c_void_p)
        m.is_loop = self.is_loop.ctypes.data_as(C.c_void_p)
        return m


def injected_start(mp: MapTable, n_agents: int, first_point: int = 3, stride: int = 3, jitter_seed: int = 4):
    """Deterministic start for more agents than the reference's rejection sampler can place (BASELINE config 4: 32 agents on the
    on-ramp map): agent i on path i mod n_paths at centre-line point first_point + stride * (i // n_paths) with the
    centre line's yaw there, plus a few millimetres / milliradians of jitter so that no two mutual distances tie exactly.  Returns
    ``(predefined_ref_path_idx, init_state)`` in the form ``Parameters`` takes them (the reference's world_state_rt_sim.py:99-126 path:
    zero speed / steering, same start in every env).  tests/golden/gen/gen_golden.py builds the config-4 fixture with the same rule."""
    jit = np.random.default_rng(jitter_seed).uniform(-1.0, 1.0, (n_agents, 3))
    first, count = mp.list_first[0], mp.list_count[0]
    idx, st = [], []
    for i in range(n_agents):
        pid = i % count
        gp = first + pid
        n, ny = int(mp.n_center[gp]), int(mp.n_yaw[gp])
        k = min(first_point + stride * (i // count), n - 8)
        idx.append(pid)
        st.append([float(mp.center[gp, k, 0]) + 4e-3 * jit[i, 0], float(mp.center[gp, k, 1]) + 4e-3 * jit[i, 1],
                   float(mp.yaw[gp, min(k, ny - 1)]) + 2e-2 * jit[i, 2]])
    return idx, st


_cache = {}


def load_map(scenario_type: str) -> MapTable:
    if scenario_type not in _cache:
        _cache[scenario_type] = MapTable(scenario_type)
    return _cache[scenario_type]
