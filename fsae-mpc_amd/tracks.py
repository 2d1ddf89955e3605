"""Track spline tables (the `kappa` closure of main.m:18 as data): M x 4 Bezier control points per axis,
arc-length parameterised (main.m:11-17).  Track.from_csv / from_points run the library's track pipeline (csrc/track.cpp:
race-line CSV -> periodic spline -> arc-length reparameterisation, host side); the tables under tracks/*.json were produced by
tools/make_track_tables.py (a numpy / scipy restatement of the same three reference files) and the two agree to 1e-13."""
import ctypes as C
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

    @staticmethod
    def _from_table(t, name):
        from ._lib import lib
        M = t.M
        xP = np.ctypeslib.as_array(t.xP, (4 * M,)).reshape(4, M).T.copy()
        yP = np.ctypeslib.as_array(t.yP, (4 * M,)).reshape(4, M).T.copy()
        tr = Track(xP, yP, t.dl, t.L, name)
        lib().fsaempc_track_free(C.byref(t))
        return tr

    @staticmethod
    def from_csv(path, M=100, name=None):
        """main.m:11-17 through the library: read_raceline_csv -> make_spline_periodic -> arclength_reparam(.., M, true)."""
        from ._lib import FsaempcError, TrackTable, lib
        t = TrackTable()
        rc = lib().fsaempc_track_from_csv(os.fsencode(path), M, C.byref(t))
        if rc != 0:
            raise FsaempcError("fsaempc_track_from_csv failed (%d): %s" % (rc, lib().fsaempc_track_last_error().decode()))
        return Track._from_table(t, name or os.path.splitext(os.path.basename(path))[0])

    @staticmethod
    def from_points(x, y, M=100, name="track"):
        from ._lib import FsaempcError, TrackTable, lib
        x = np.ascontiguousarray(x, dtype=np.float64); y = np.ascontiguousarray(y, dtype=np.float64)
        t = TrackTable()
        rc = lib().fsaempc_track_from_points(x.ctypes.data_as(C.c_void_p), y.ctypes.data_as(C.c_void_p), len(x), M, C.byref(t))
        if rc != 0:
            raise FsaempcError("fsaempc_track_from_points failed (%d): %s" % (rc, lib().fsaempc_track_last_error().decode()))
        return Track._from_table(t, name)

    def save_table(self, path):
        """binary FSTRK001 table (include/fsaempc.h)"""
        from ._lib import FsaempcError, TrackTable, lib
        xs = np.ascontiguousarray(self.xP.T).ravel(); ys = np.ascontiguousarray(self.yP.T).ravel()
        t = TrackTable(self.M, self.dl, self.L, xs.ctypes.data_as(C.POINTER(C.c_double)), ys.ctypes.data_as(C.POINTER(C.c_double)))
        if lib().fsaempc_track_save(C.byref(t), os.fsencode(path)) != 0:
            raise FsaempcError(lib().fsaempc_track_last_error().decode())

    @staticmethod
    def load_table(path, name="track"):
        from ._lib import FsaempcError, TrackTable, lib
        t = TrackTable()
        if lib().fsaempc_track_load(os.fsencode(path), C.byref(t)) != 0:
            raise FsaempcError(lib().fsaempc_track_last_error().decode())
        return Track._from_table(t, name)

    def device(self, device):
        """(xP, yP) as device tensors holding the column-major M x 4 tables."""
        import torch
        key = str(device)
        if key not in self._dev:
            fx = torch.from_numpy(np.ascontiguousarray(self.xP.T)).to(device)  # memory = column-major M x 4
            fy = torch.from_numpy(np.ascontiguousarray(self.yP.T)).to(device)
            self._dev[key] = (fx, fy)
        return self._dev[key]
