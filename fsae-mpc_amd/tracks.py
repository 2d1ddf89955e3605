"""Track spline tables (the `kappa` closure of main.m:18 as data): M x 4 Bezier control points per axis,
arc-length parameterised (main.m:11-17).  Tables under tracks/*.json are produced by tools/make_track_tables.py."""
import json
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))


class Track:
    def __init__(self, xP, yP, dl, L, name="track"):
        self.name = name
        self.xP = np.asfortranarray(np.asarray(xP, dtype=np.float64))  # M x 4, column-major
        self.yP = np.asfortranarray(np.asarray(yP, dtype=np.float64))
        self.M = self.xP.shape[0]
        self.dl = float(dl)
        self.L = float(L)
        self._dev = {}

    @staticmethod
    def load(name_or_path):
        path = name_or_path if os.path.exists(name_or_path) else os.path.join(_HERE, "tracks", name_or_path + ".json")
        with open(path) as f:
            d = json.load(f)
        return Track(d["xP"], d["yP"], d["dl"], d["L"], d.get("name", "track"))

    def device(self, device):
        """(xP, yP) as device tensors holding the column-major M x 4 tables."""
        import torch
        key = str(device)
        if key not in self._dev:
            fx = torch.from_numpy(np.ascontiguousarray(self.xP.T)).to(device)  # memory = column-major M x 4
            fy = torch.from_numpy(np.ascontiguousarray(self.yP.T)).to(device)
            self._dev[key] = (fx, fy)
        return self._dev[key]
