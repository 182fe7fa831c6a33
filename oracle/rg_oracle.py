"""ctypes binding of oracle/librg_oracle.so (the double-precision CPU restatement of mj_step).

TEST INFRASTRUCTURE: importable only from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force=False):
    so = os.path.join(_HERE, "librg_oracle.so")
    src = os.path.join(_HERE, "rg_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "librg_oracle.so"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = ctypes.CDLL(build())
        L.ro_model_load.restype = ctypes.c_void_p
        L.ro_model_load.argtypes = [ctypes.c_char_p, ctypes.c_size_t]
        L.ro_data_new.restype = ctypes.c_void_p
        L.ro_data_new.argtypes = [ctypes.c_void_p]
        L.ro_field.restype = ctypes.POINTER(ctypes.c_double)
        L.ro_field.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_char_p, ctypes.POINTER(ctypes.c_int)]
        L.ro_int.restype = ctypes.c_int
        L.ro_int.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_char_p]
        L.ro_efc_type.restype = ctypes.POINTER(ctypes.c_int)
        L.ro_efc_type.argtypes = [ctypes.c_void_p]
        L.ro_time.restype = ctypes.c_double
        L.ro_time.argtypes = [ctypes.c_void_p]
        L.ro_set_time.argtypes = [ctypes.c_void_p, ctypes.c_double]
        for fn in ("ro_forward", "ro_step", "ro_reset", "ro_fwd_position"):
            getattr(L, fn).argtypes = [ctypes.c_void_p, ctypes.c_void_p]
            getattr(L, fn).restype = None
        L.ro_sim_step.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
        L.ro_sim_step.restype = None
        L.ro_model_free.argtypes = [ctypes.c_void_p]
        L.ro_data_free.argtypes = [ctypes.c_void_p]
        L.ro_contact_get.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_double)]
        L.ro_stats.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_double)]
        L.ro_stats_reset.argtypes = [ctypes.c_void_p]
        L.ro_mpr_pair.restype = ctypes.c_int
        L.ro_mpr_pair.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_double,
                                  ctypes.POINTER(ctypes.c_double)]
        L.ro_set_mpr_libccd_tridist.argtypes = [ctypes.c_int]
        L.ro_set_boxbox_multipoint.argtypes = [ctypes.c_int]
        _LIB = L
    return _LIB


def set_kernel_variant(on: bool):
    """The oracle's default is the closest available statement of MuJoCo 2.0 (libccd's triangle-distance MPR depth,
    multi-point box-box).  `on` switches it to the HIP kernel's two documented deviations (portal-plane MPR depth,
    box-box through MPR), so that a test can separate "the kernel computes what it says" (tight, kernel variant)
    from "how far what it says is from the MuJoCo restatement" (measured, default)."""
    L = lib()
    L.ro_set_mpr_libccd_tridist(0 if on else 1)
    L.ro_set_boxbox_multipoint(0 if on else 1)


class OracleSim:
    """One env of the CPU oracle; numpy views alias the C arrays (writes go through)."""

    def __init__(self, model_blob: bytes):
        L = lib()
        self._blob = model_blob
        self.m = L.ro_model_load(model_blob, len(model_blob))
        if not self.m:
            raise ValueError("bad model blob")
        self.d = L.ro_data_new(self.m)

    def __del__(self):
        try:
            L = lib()
            L.ro_data_free(self.d)
            L.ro_model_free(self.m)
        except Exception:
            pass

    def field(self, name) -> np.ndarray:
        n = ctypes.c_int(0)
        p = lib().ro_field(self.m, self.d, name.encode(), ctypes.byref(n))
        if not p:
            raise KeyError(name)
        if n.value == 0:
            return np.zeros(0)
        return np.ctypeslib.as_array(p, shape=(n.value,))

    def __getattr__(self, name):
        if name.startswith("_") or name in ("m", "d"):
            raise AttributeError(name)
        v = lib().ro_int(self.m, self.d, name.encode())
        if v >= 0:
            return v
        return self.field(name)

    @property
    def time(self):
        return lib().ro_time(self.d)

    def efc_types(self):
        return np.ctypeslib.as_array(lib().ro_efc_type(self.d), shape=(600,))[: self.nefc].copy()

    def reset(self):
        lib().ro_reset(self.m, self.d)

    def forward(self):
        lib().ro_forward(self.m, self.d)

    def fwd_position(self):
        lib().ro_fwd_position(self.m, self.d)

    def step(self):
        lib().ro_step(self.m, self.d)

    def sim_step(self, nsubsteps):
        lib().ro_sim_step(self.m, self.d, nsubsteps)

    def contacts(self):
        out = []
        buf = (ctypes.c_double * 23)()
        for i in range(self.ncon):
            lib().ro_contact_get(self.d, i, buf)
            a = np.array(buf[:])
            out.append(dict(dist=a[0], pos=a[1:4], frame=a[4:13].reshape(3, 3), includemargin=a[13], friction=a[14:19],
                            dim=int(a[19]), geom1=int(a[20]), geom2=int(a[21]), efc_address=int(a[22])))
        return out

    def stats(self):
        buf = (ctypes.c_double * 6)()
        lib().ro_stats(self.d, buf)
        return dict(ncon=buf[0], nefc=buf[1], iters=buf[2], steps=buf[3], mpr_calls=buf[4], mpr_iters=buf[5])

    def stats_reset(self):
        lib().ro_stats_reset(self.d)

    def mpr_pair(self, g1, g2, margin=0.0):
        buf = (ctypes.c_double * 7)()
        rc = lib().ro_mpr_pair(self.m, self.d, g1, g2, margin, buf)
        a = np.array(buf[:])
        return rc, a[0], a[1:4], a[4:7]
