"""ctypes binding of oracle/librg_oracle.so (the double-precision CPU restatement of mj_step).

TEST INFRASTRUCTURE: importable only from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIBS = {}


def _host_signature():
    """The libraries are built -march=native, so a build only belongs to the CPU it was made on."""
    try:
        with open("/proc/cpuinfo") as f:
            txt = f.read()
        keep = [l for l in txt.split("\n") if l.startswith(("model name", "flags"))][:2]
        import hashlib
        return hashlib.sha256("\n".join(keep).encode()).hexdigest()[:16]
    except OSError:
        return "unknown"


def build(force=False, f32=False):
    name = "librg_oracle_f32.so" if f32 else "librg_oracle.so"
    so = os.path.join(_HERE, name)
    src = os.path.join(_HERE, "rg_oracle.c")
    stamp = so + ".host"
    sig = _host_signature()
    try:
        same_host = open(stamp).read().strip() == sig
    except OSError:
        same_host = False
    if force or not same_host or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B", name])
        with open(stamp, "w") as f:
            f.write(sig)
    return so


def lib(f32=False):
    """f32=False: THE oracle (double).  f32=True: the same source built in single precision (precision studies only)."""
    if f32 not in _LIBS:
        L = ctypes.CDLL(build(f32=f32))
        real_p = ctypes.POINTER(ctypes.c_float if f32 else ctypes.c_double)
        assert L.ro_real_size() == (4 if f32 else 8)
        L.ro_model_load.restype = ctypes.c_void_p
        L.ro_model_load.argtypes = [ctypes.c_char_p, ctypes.c_size_t]
        L.ro_data_new.restype = ctypes.c_void_p
        L.ro_data_new.argtypes = [ctypes.c_void_p]
        L.ro_field.restype = real_p
        L.ro_field.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_char_p, ctypes.POINTER(ctypes.c_int)]
        L.ro_int.restype = ctypes.c_int
        L.ro_int.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_char_p]
        L.ro_efc_type.restype = ctypes.POINTER(ctypes.c_int)
        L.ro_efc_type.argtypes = [ctypes.c_void_p]
        L.ro_solve_pgs.restype = ctypes.c_int
        L.ro_solve_pgs.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_double, ctypes.POINTER(ctypes.c_double)]
        L.ro_eq_active.restype = ctypes.POINTER(ctypes.c_int)
        L.ro_eq_active.argtypes = [ctypes.c_void_p]
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
        L.ro_bench_locked.restype = ctypes.c_int
        L.ro_bench_locked.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_double, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int),
                                      ctypes.c_int, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double), ctypes.c_int, ctypes.c_int,
                                      ctypes.c_uint64, ctypes.POINTER(ctypes.c_long), ctypes.POINTER(ctypes.c_double)]
        L.ro_f32 = f32
        _LIBS[f32] = L
    return _LIBS[f32]


_VARIANT = [False]


def set_kernel_variant(on: bool):
    """The oracle's default is the closest available statement of MuJoCo 2.0 (libccd's triangle-distance MPR depth,
    multi-point box-box).  `on` switches it to the HIP kernel's two documented deviations (portal-plane MPR depth,
    box-box through MPR), so that a test can separate "the kernel computes what it says" (tight, kernel variant)
    from "how far what it says is from the MuJoCo restatement" (measured, default)."""
    _VARIANT[0] = bool(on)
    for L in [lib()] + [l for k, l in _LIBS.items() if k]:
        L.ro_set_mpr_libccd_tridist(0 if on else 1)
        L.ro_set_boxbox_multipoint(0 if on else 1)


class OracleSim:
    """One env of the CPU oracle; numpy views alias the C arrays (writes go through)."""

    def __init__(self, model_blob: bytes, f32: bool = False):
        L = self._L = lib(f32)
        if f32:
            set_kernel_variant(_VARIANT[0])   # a library loaded later takes the current variant
        self._blob = model_blob
        self.m = L.ro_model_load(model_blob, len(model_blob))
        if not self.m:
            raise ValueError("bad model blob")
        self.d = L.ro_data_new(self.m)

    def __del__(self):
        try:
            L = self._L
            L.ro_data_free(self.d)
            L.ro_model_free(self.m)
        except Exception:
            pass

    def field(self, name) -> np.ndarray:
        n = ctypes.c_int(0)
        p = self._L.ro_field(self.m, self.d, name.encode(), ctypes.byref(n))
        if not p:
            raise KeyError(name)
        if n.value == 0:
            return np.zeros(0)
        return np.ctypeslib.as_array(p, shape=(n.value,))

    def __getattr__(self, name):
        if name.startswith("_") or name in ("m", "d"):  # (also keeps _L out of the C lookups)
            raise AttributeError(name)
        v = self._L.ro_int(self.m, self.d, name.encode())
        if v >= 0:
            return v
        return self.field(name)

    @property
    def time(self):
        return self._L.ro_time(self.d)

    def efc_types(self):
        return np.ctypeslib.as_array(self._L.ro_efc_type(self.d), shape=(2000,))[: self.nefc].copy()

    def solve_pgs(self, max_sweeps=200000, tol=1e-13):
        """The dual projected Gauss-Seidel solver on the rows of the last forward(): (qacc, sweeps).  Independent cross-check of the Newton solver."""
        out = (ctypes.c_double * self.field("qacc").shape[0])()
        n = self._L.ro_solve_pgs(self.m, self.d, int(max_sweeps), float(tol), out)
        return np.array(out[:]), n

    def eq_active(self):
        """mjModel.eq_active as a writable int view (run-time copy: the envs toggle it)."""
        n = self.neq
        return np.ctypeslib.as_array(self._L.ro_eq_active(self.d), shape=(max(n, 1),))[:n]

    def reset(self):
        self._L.ro_reset(self.m, self.d)

    def forward(self):
        self._L.ro_forward(self.m, self.d)

    def fwd_position(self):
        self._L.ro_fwd_position(self.m, self.d)

    def step(self):
        self._L.ro_step(self.m, self.d)

    def sim_step(self, nsubsteps):
        self._L.ro_sim_step(self.m, self.d, nsubsteps)

    def contacts(self):
        out = []
        buf = (ctypes.c_double * 23)()
        for i in range(self.ncon):
            self._L.ro_contact_get(self.d, i, buf)
            a = np.array(buf[:])
            out.append(dict(dist=a[0], pos=a[1:4], frame=a[4:13].reshape(3, 3), includemargin=a[13], friction=a[14:19],
                            dim=int(a[19]), geom1=int(a[20]), geom2=int(a[21]), efc_address=int(a[22])))
        return out

    def stats(self):
        buf = (ctypes.c_double * 6)()
        self._L.ro_stats(self.d, buf)
        return dict(ncon=buf[0], nefc=buf[1], iters=buf[2], steps=buf[3], mpr_calls=buf[4], mpr_iters=buf[5])

    def stats_reset(self):
        self._L.ro_stats_reset(self.d)

    def mpr_pair(self, g1, g2, margin=0.0):
        buf = (ctypes.c_double * 7)()
        rc = self._L.ro_mpr_pair(self.m, self.d, g1, g2, margin, buf)
        a = np.array(buf[:])
        return rc, a[0], a[1:4], a[4:7]
