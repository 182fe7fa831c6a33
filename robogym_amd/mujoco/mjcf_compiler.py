"""MJCF-subset model compiler (host side, double precision, runs once per model).

Turns a merged MJCF element tree (see `mujoco_xml.MujocoXML`) into the flat
arrays the batched HIP stepper consumes through the C ABI (`include/rgstep.h`).
It replaces, for the subset of MJCF the robogym hot path loads, what the
reference obtains from `mujoco_py.load_model_from_xml`
(/root/reference/robogym/mujoco/mujoco_xml.py:249-260) — MuJoCo itself is a
closed third-party dependency of the reference (`setup.py:16`,
mujoco-py==2.0.2.13 / MuJoCo 2.0) and is restated here from its published
model semantics.  Field names follow `mjModel` so that reference code touching
`sim.model.*` maps one-to-one.

Supported subset (everything dactyl/locked, dactyl/reach need; SURVEY §8a
"model features"): compiler(angle, eulerseq), option, nested default classes and
childclass, body/inertial/joint(free,ball,slide,hinge)/geom(plane,sphere,capsule,
cylinder,box,mesh)/site, mesh assets (binary STL) -> convex hull, geom-derived
inertia, contact excludes, fixed and spatial tendons (site / cylinder wrap with
sidesite), general actuators (joint or tendon transmission), touch sensors.
"""
import os
import struct
from typing import Dict, List, Optional

import numpy as np

# ----------------------------------------------------------------------------- enums (mjModel values)
JNT_FREE, JNT_BALL, JNT_SLIDE, JNT_HINGE = 0, 1, 2, 3
GEOM_PLANE, GEOM_HFIELD, GEOM_SPHERE, GEOM_CAPSULE, GEOM_ELLIPSOID, GEOM_CYLINDER, GEOM_BOX, GEOM_MESH = range(8)
WRAP_JOINT, WRAP_PULLEY, WRAP_SITE, WRAP_SPHERE, WRAP_CYLINDER = 1, 2, 3, 4, 5
TRN_JOINT, TRN_TENDON = 0, 3
EQ_WELD, EQ_JOINT = 1, 2   # mjtEq: connect 0, weld 1, joint 2
SENS_TOUCH, SENS_FORCE, SENS_TORQUE, SENS_JOINTPOS = 0, 4, 5, 8   # mjtSensor (MuJoCo 2.0)
GAIN_FIXED, GAIN_USER = 0, 2  # mjGAIN_FIXED, mjGAIN_USER (MuJoCo 2.0: fixed=0, user=1 … we only need "user or not")
MINVAL = 1e-15

_GEOM_TYPES = dict(plane=0, hfield=1, sphere=2, capsule=3, ellipsoid=4, cylinder=5, box=6, mesh=7)
_JNT_TYPES = dict(free=0, ball=1, slide=2, hinge=3)


# ----------------------------------------------------------------------------- quaternion helpers (w,x,y,z)
def qmul(a, b):
    aw, ax, ay, az = a
    bw, bx, by, bz = b
    return np.array(
        [
            aw * bw - ax * bx - ay * by - az * bz,
            aw * bx + ax * bw + ay * bz - az * by,
            aw * by - ax * bz + ay * bw + az * bx,
            aw * bz + ax * by - ay * bx + az * bw,
        ]
    )


def qconj(q):
    return np.array([q[0], -q[1], -q[2], -q[3]])


def qnorm(q):
    q = np.asarray(q, dtype=float)
    n = np.linalg.norm(q)
    if n < MINVAL:
        return np.array([1.0, 0, 0, 0])
    return q / n


def q2mat(q):
    w, x, y, z = q
    return np.array(
        [
            [w * w + x * x - y * y - z * z, 2 * (x * y - w * z), 2 * (x * z + w * y)],
            [2 * (x * y + w * z), w * w - x * x + y * y - z * z, 2 * (y * z - w * x)],
            [2 * (x * z - w * y), 2 * (y * z + w * x), w * w - x * x - y * y + z * z],
        ]
    )


def mat2q(m):
    """Rotation matrix -> unit quaternion (largest-pivot branch)."""
    m = np.asarray(m, dtype=float)
    tr = m[0, 0] + m[1, 1] + m[2, 2]
    if tr > 0:
        s = np.sqrt(tr + 1.0) * 2
        q = [0.25 * s, (m[2, 1] - m[1, 2]) / s, (m[0, 2] - m[2, 0]) / s, (m[1, 0] - m[0, 1]) / s]
    elif m[0, 0] > m[1, 1] and m[0, 0] > m[2, 2]:
        s = np.sqrt(1.0 + m[0, 0] - m[1, 1] - m[2, 2]) * 2
        q = [(m[2, 1] - m[1, 2]) / s, 0.25 * s, (m[0, 1] + m[1, 0]) / s, (m[0, 2] + m[2, 0]) / s]
    elif m[1, 1] > m[2, 2]:
        s = np.sqrt(1.0 + m[1, 1] - m[0, 0] - m[2, 2]) * 2
        q = [(m[0, 2] - m[2, 0]) / s, (m[0, 1] + m[1, 0]) / s, 0.25 * s, (m[1, 2] + m[2, 1]) / s]
    else:
        s = np.sqrt(1.0 + m[2, 2] - m[0, 0] - m[1, 1]) * 2
        q = [(m[1, 0] - m[0, 1]) / s, (m[0, 2] + m[2, 0]) / s, (m[1, 2] + m[2, 1]) / s, 0.25 * s]
    return qnorm(q)


def axisangle2q(axis, angle):
    axis = np.asarray(axis, dtype=float)
    n = np.linalg.norm(axis)
    if n < MINVAL:
        return np.array([1.0, 0, 0, 0])
    s = np.sin(angle / 2) / n
    return np.array([np.cos(angle / 2), axis[0] * s, axis[1] * s, axis[2] * s])


def euler2q(e, seq="xyz"):
    """MJCF euler: lower-case letters rotate about the moving frame (q <- q*r),
    upper-case about the fixed frame (q <- r*q)."""
    q = np.array([1.0, 0, 0, 0])
    for ang, ch in zip(e, seq):
        ax = np.zeros(3)
        ax["xyz".index(ch.lower())] = 1.0
        r = axisangle2q(ax, ang)
        q = qmul(q, r) if ch.islower() else qmul(r, q)
    return q


def z2vec_q(vec):
    """Quaternion rotating the z axis onto `vec`."""
    vec = np.asarray(vec, dtype=float)
    vec = vec / np.linalg.norm(vec)
    z = np.array([0.0, 0, 1])
    axis = np.cross(z, vec)
    s = np.linalg.norm(axis)
    if s < 1e-10:
        return np.array([1.0, 0, 0, 0]) if vec[2] > 0 else np.array([0.0, 1, 0, 0])
    return axisangle2q(axis / s, np.arctan2(s, vec[2]))


def qrot(q, v):
    return q2mat(q) @ np.asarray(v, dtype=float)


# ----------------------------------------------------------------------------- parsing helpers
def _floats(s, n=None, default=None):
    if s is None:
        return None if default is None else np.array(default, dtype=float)
    a = np.array([float(x) for x in s.split()], dtype=float)
    if n is not None and default is not None and len(a) < n:
        full = np.array(default, dtype=float)
        full[: len(a)] = a
        a = full
    return a


def _bool(s, default=False):
    if s is None:
        return default
    return s.strip().lower() == "true"


class _Defaults:
    """Nested `<default class=…>` trees; lookups return merged attribute dicts."""

    ACT_TAGS = ("general", "motor", "position", "velocity", "cylinder", "muscle")
    TEN_TAGS = ("tendon", "fixed", "spatial")

    def __init__(self):
        self.classes: Dict[str, Dict[str, Dict[str, str]]] = {"main": {}}

    def _norm(self, tag):
        if tag in self.ACT_TAGS:
            return "general"
        if tag in self.TEN_TAGS:
            return "tendon"
        return tag

    def load(self, elem, parent="main", top=True):
        name = "main" if top and elem.get("class") is None else elem.get("class", "main")
        if name not in self.classes:
            self.classes[name] = {t: dict(a) for t, a in self.classes[parent].items()}
        cur = self.classes[name]
        for child in elem:
            if child.tag == "default":
                continue
            cur.setdefault(self._norm(child.tag), {}).update(child.attrib)
        for child in elem:
            if child.tag == "default":
                self.load(child, parent=name, top=False)

    def resolve(self, elem, childclass):
        cls = elem.get("class") or childclass or "main"
        if cls not in self.classes:
            raise ValueError("unknown default class %r" % cls)
        merged = dict(self.classes[cls].get(self._norm(elem.tag), {}))
        merged.update(elem.attrib)
        return merged


# ----------------------------------------------------------------------------- meshes
def load_stl(path):
    """Binary STL -> (nfaces,3,3) float64 triangle soup."""
    with open(path, "rb") as f:
        raw = f.read()
    n = struct.unpack("<I", raw[80:84])[0]
    if 84 + 50 * n != len(raw):
        raise ValueError("%s: not a binary STL" % path)
    rec = np.dtype([("n", "<f4", 3), ("v", "<f4", (3, 3)), ("a", "<u2")])
    return np.frombuffer(raw, dtype=rec, count=n, offset=84)["v"].astype(float)


def load_msh(path):
    """MuJoCo's legacy binary mesh (.msh): int32 nvertex, nnormal, ntexcoord, nface, then float32 vertices [nvertex, 3], normals
    [nnormal, 3], texcoords [ntexcoord, 2], int32 faces [nface, 3]  ->  (nfaces, 3, 3) triangle soup."""
    with open(path, "rb") as f:
        raw = f.read()
    nv, nn, nt, nf = struct.unpack("<4i", raw[:16])
    if 16 + 4 * (3 * nv + 3 * nn + 2 * nt + 3 * nf) != len(raw):
        raise ValueError("%s: not a MuJoCo .msh file" % path)
    vert = np.frombuffer(raw, dtype="<f4", count=3 * nv, offset=16).reshape(nv, 3).astype(float)
    face = np.frombuffer(raw, dtype="<i4", count=3 * nf, offset=16 + 4 * (3 * nv + 3 * nn + 2 * nt)).reshape(nf, 3)
    return vert[face]


def _legacy_mesh_frame(tris):
    """Centre of mass and principal frame of a triangle mesh, MuJoCo-2.0 style:
    pyramids from the area-weighted face centroid with unsigned volumes."""
    a, b, c = tris[:, 0], tris[:, 1], tris[:, 2]
    area = 0.5 * np.linalg.norm(np.cross(b - a, c - a), axis=1)
    fc = (a + b + c) / 3
    cen = (fc * area[:, None]).sum(0) / area.sum()
    a, b, c = a - cen, b - cen, c - cen
    vol = np.abs(np.einsum("ij,ij->i", a, np.cross(b, c))) / 6
    com_local = ((a + b + c) / 4 * vol[:, None]).sum(0) / vol.sum()
    com = cen + com_local
    a, b, c = a - com_local, b - com_local, c - com_local
    # second moments of each tetra (apex at the com): covariance integral
    # ∫ x x^T dV = vol/20 * (Σ_i v_i v_i^T + (Σ v_i)(Σ v_i)^T) with v_0 = 0
    s = a + b + c
    P = np.einsum("i,ijk->jk", vol / 20.0,
                  np.einsum("ij,ik->ijk", a, a) + np.einsum("ij,ik->ijk", b, b)
                  + np.einsum("ij,ik->ijk", c, c) + np.einsum("ij,ik->ijk", s, s))
    inertia = np.trace(P) * np.eye(3) - P
    w, V = np.linalg.eigh(inertia)
    order = np.argsort(-w)  # descending, as mju_eig3
    V = V[:, order]
    if np.linalg.det(V) < 0:
        V[:, 2] = -V[:, 2]
    return com, V, vol.sum(), w[order]


def process_mesh(path, scale):
    from scipy.spatial import ConvexHull

    tris = (load_msh(path) if path.lower().endswith(".msh") else load_stl(path)) * np.asarray(scale, dtype=float)
    com, R, volume, inertia = _legacy_mesh_frame(tris)
    pts = np.unique(tris.reshape(-1, 3), axis=0)
    hull = ConvexHull(pts)
    verts = pts[np.sort(hull.vertices)]
    local = ((verts - com) @ R).astype(np.float32)  # stored as float, like mjModel.mesh_vert
    return dict(pos=com, quat=mat2q(R), vert=local, volume=volume, inertia=inertia)   # (inertia: principal moments at unit density)


# ----------------------------------------------------------------------------- the compiled model
class CompiledModel:
    """Flat, named arrays + name tables.  `arrays` is what crosses the C ABI."""

    def __init__(self):
        self.arrays: Dict[str, np.ndarray] = {}
        self.names: Dict[str, List[str]] = {}

    def __getattr__(self, key):
        arrays = self.__dict__.get("arrays", {})
        if key in arrays:
            return arrays[key]
        raise AttributeError(key)

    def name2id(self, kind, name):
        return self.names[kind].index(name)

    def copy_with(self, **overrides):
        """A copy of the model with some arrays replaced (per-env model variants: the oracle side of the per-env parameter
        tests, `refresh_constants`).  Shapes are taken from the original arrays."""
        m = CompiledModel()
        m.names = self.names
        m.arrays = {k: v.copy() for k, v in self.arrays.items()}
        for k, v in overrides.items():
            m.arrays[k] = np.asarray(v, dtype=self.arrays[k].dtype).reshape(self.arrays[k].shape).copy()
        return m

    def save(self, path):
        payload = dict(self.arrays)
        for kind, lst in self.names.items():
            payload["names_" + kind] = np.array(lst, dtype=object).astype(str) if lst else np.array([], dtype=str)
        np.savez_compressed(path, **payload)

    @classmethod
    def load(cls, path):
        m = cls()
        with np.load(path, allow_pickle=False) as z:
            for k in z.files:
                if k.startswith("names_"):
                    m.names[k[6:]] = [str(s) for s in z[k]]
                else:
                    m.arrays[k] = z[k]
        return m


def _orientation(attrs, eulerseq):
    if "quat" in attrs:
        return qnorm(_floats(attrs["quat"]))
    if "euler" in attrs:
        return euler2q(_floats(attrs["euler"]), eulerseq)
    if "axisangle" in attrs:
        aa = _floats(attrs["axisangle"])
        return axisangle2q(aa[:3], aa[3])
    if "zaxis" in attrs:
        return z2vec_q(_floats(attrs["zaxis"]))
    return np.array([1.0, 0, 0, 0])


def _geom_mass_inertia(gtype, size, density):
    if gtype == GEOM_SPHERE:
        r = size[0]
        m = density * 4 / 3 * np.pi * r ** 3
        return m, np.full(3, 0.4 * m * r * r)
    if gtype == GEOM_BOX:
        m = density * 8 * size[0] * size[1] * size[2]
        return m, m / 3 * np.array([size[1] ** 2 + size[2] ** 2, size[0] ** 2 + size[2] ** 2, size[0] ** 2 + size[1] ** 2])
    if gtype == GEOM_CYLINDER:
        r, h = size[0], size[1]
        m = density * np.pi * r * r * 2 * h
        ix = m * (3 * r * r + 4 * h * h) / 12
        return m, np.array([ix, ix, m * r * r / 2])
    if gtype == GEOM_CAPSULE:
        r, h = size[0], size[1]
        mc = density * np.pi * r * r * 2 * h
        ms = density * 4 / 3 * np.pi * r ** 3
        ix = mc * (3 * r * r + 4 * h * h) / 12 + ms * (0.4 * r * r + 0.75 * r * h + h * h)
        return mc + ms, np.array([ix, ix, mc * r * r / 2 + 0.4 * ms * r * r])
    raise NotImplementedError("mass of geom type %d" % gtype)


def compile_mjcf(root, meshdir: Optional[str] = None, verbose: bool = False) -> CompiledModel:
    # ------------------------------------------------------------ global sections
    eulerseq, angle_scale = "xyz", np.pi / 180
    for c in root.findall("compiler"):
        if c.get("angle") == "radian":
            angle_scale = 1.0
        elif c.get("angle") == "degree":
            angle_scale = np.pi / 180
        eulerseq = c.get("eulerseq", eulerseq)
        if c.get("meshdir") and meshdir is None:
            meshdir = c.get("meshdir")
        if c.get("coordinate", "local") != "local":
            raise NotImplementedError("global coordinates")
    if angle_scale != 1.0:
        raise NotImplementedError("degree angles (robogym always compiles with angle=radian)")

    opt = dict(timestep=0.002, gravity=[0, 0, -9.81], iterations=100, tolerance=1e-8, impratio=1.0,
               cone=0, ls_iterations=50, ls_tolerance=0.01, mpr_iterations=50, mpr_tolerance=1e-6,
               noslip_iterations=0)
    for o in root.findall("option"):
        for k in ("timestep", "tolerance", "impratio", "ls_tolerance", "mpr_tolerance"):
            if o.get(k) is not None:
                opt[k] = float(o.get(k))
        for k in ("iterations", "ls_iterations", "mpr_iterations", "noslip_iterations"):
            if o.get(k) is not None:
                opt[k] = int(o.get(k))
        if o.get("gravity") is not None:
            opt["gravity"] = list(_floats(o.get("gravity")))
        if o.get("cone") is not None:
            opt["cone"] = dict(pyramidal=0, elliptic=1)[o.get("cone")]
        for k in ("solver", "jacobian", "integrator"):
            if o.get(k) not in (None, "Newton", "auto", "dense", "Euler"):
                raise NotImplementedError("option %s=%s" % (k, o.get(k)))
    size = dict(njmax=-1, nconmax=-1, nuserdata=0, nuser_actuator=0)
    for s in root.findall("size"):
        for k in size:
            if s.get(k) is not None:
                size[k] = int(s.get(k))

    defaults = _Defaults()
    for d in root.findall("default"):
        defaults.load(d)

    # ------------------------------------------------------------ meshes
    mesh_defs: Dict[str, dict] = {}
    for a in root.findall("asset"):
        for me in a.findall("mesh"):
            at = defaults.resolve(me, None)
            fn = at["file"]
            name = at.get("name") or os.path.splitext(os.path.basename(fn))[0]
            mesh_defs[name] = dict(file=fn if os.path.isabs(fn) else os.path.join(meshdir, fn),
                                   scale=_floats(at.get("scale"), 3, [1, 1, 1]))
    used_meshes: Dict[str, int] = {}
    mesh_data: List[dict] = []

    def mesh_id(name):
        if name not in used_meshes:
            md = mesh_defs[name]
            used_meshes[name] = len(mesh_data)
            mesh_data.append(process_mesh(md["file"], md["scale"]))
        return used_meshes[name]

    # ------------------------------------------------------------ bodies
    B = dict(name=["world"], parentid=[0], pos=[np.zeros(3)], quat=[np.array([1.0, 0, 0, 0])],
             ipos=[np.zeros(3)], iquat=[np.array([1.0, 0, 0, 0])], mass=[0.0], inertia=[np.zeros(3)],
             jntadr=[-1], jntnum=[0], geomadr=[-1], geomnum=[0], mocap=[0])
    J = dict(name=[], type=[], bodyid=[], pos=[], axis=[], stiffness=[], range=[], limited=[], margin=[],
             armature=[], damping=[], frictionloss=[], ref=[], springref=[], solref_lim=[], solimp_lim=[],
             solref_fri=[], solimp_fri=[])
    G = dict(name=[], type=[], bodyid=[], dataid=[], size=[], pos=[], quat=[], friction=[], condim=[],
             contype=[], conaffinity=[], margin=[], gap=[], solmix=[], solref=[], solimp=[], rbound=[],
             mass=[], inertia=[])
    S = dict(name=[], bodyid=[], pos=[], quat=[], type=[], size=[])
    SOLREF, SOLIMP = [0.02, 1.0], [0.9, 0.95, 0.001, 0.5, 2.0]

    def add_geom(at, bodyid):
        gtype = _GEOM_TYPES[at.get("type", "sphere")]
        sz = _floats(at.get("size"), 3, [0, 0, 0])
        if sz is None:
            sz = np.zeros(3)
        pos = _floats(at.get("pos"), 3, [0, 0, 0])
        quat = _orientation(at, eulerseq)
        dataid = -1
        if at.get("mesh") is not None:
            gtype = GEOM_MESH
        if "fromto" in at and gtype in (GEOM_CAPSULE, GEOM_CYLINDER, GEOM_BOX, GEOM_ELLIPSOID):
            ft = _floats(at["fromto"])
            p0, p1 = ft[:3], ft[3:]
            pos = 0.5 * (p0 + p1)
            quat = z2vec_q(p1 - p0)
            sz = np.array([sz[0], 0.5 * np.linalg.norm(p1 - p0), 0.0])
        density = float(at.get("density", 1000.0))
        if gtype == GEOM_MESH:
            dataid = mesh_id(at["mesh"])
            md = mesh_data[dataid]
            # geom frame = user frame ∘ mesh inertial frame (vertices are stored in the latter)
            pos = pos + qrot(quat, md["pos"])
            quat = qmul(quat, md["quat"])
            v = md["vert"].astype(float)
            aabb = np.abs(v).max(0)
            sz = aabb
            rbound = float(np.linalg.norm(aabb))
            mass, inertia = density * md["volume"], density * np.asarray(md["inertia"], dtype=float)
        elif gtype == GEOM_PLANE:
            rbound, mass, inertia = 0.0, 0.0, np.zeros(3)
        else:
            rbound = {GEOM_SPHERE: lambda: sz[0], GEOM_CAPSULE: lambda: sz[0] + sz[1],
                      GEOM_CYLINDER: lambda: np.hypot(sz[0], sz[1]), GEOM_BOX: lambda: np.linalg.norm(sz),
                      GEOM_ELLIPSOID: lambda: max(sz)}[gtype]()
            mass, inertia = _geom_mass_inertia(gtype, sz, density)
        if at.get("mass") is not None:
            mnew = float(at["mass"])
            inertia = inertia * (mnew / mass) if mass > 0 else inertia
            mass = mnew
        G["name"].append(at.get("name", ""))
        G["type"].append(gtype); G["bodyid"].append(bodyid); G["dataid"].append(dataid)
        G["size"].append(sz); G["pos"].append(pos); G["quat"].append(quat)
        G["friction"].append(_floats(at.get("friction"), 3, [1, 0.005, 0.0001]))
        G["condim"].append(int(at.get("condim", 3)))
        G["contype"].append(int(at.get("contype", 1))); G["conaffinity"].append(int(at.get("conaffinity", 1)))
        G["margin"].append(float(at.get("margin", 0))); G["gap"].append(float(at.get("gap", 0)))
        G["solmix"].append(float(at.get("solmix", 1)))
        G["solref"].append(_floats(at.get("solref"), 2, SOLREF)); G["solimp"].append(_floats(at.get("solimp"), 5, SOLIMP))
        G["rbound"].append(rbound); G["mass"].append(mass); G["inertia"].append(inertia)

    def add_joint(at, bodyid):
        jt = _JNT_TYPES[at.get("type", "hinge")]
        axis = _floats(at.get("axis"), 3, [0, 0, 1])
        axis = axis / max(np.linalg.norm(axis), MINVAL)
        limited = _bool(at.get("limited"), False)
        J["name"].append(at.get("name", "")); J["type"].append(jt); J["bodyid"].append(bodyid)
        J["pos"].append(_floats(at.get("pos"), 3, [0, 0, 0])); J["axis"].append(axis)
        J["stiffness"].append(float(at.get("stiffness", 0)))
        J["range"].append(_floats(at.get("range"), 2, [0, 0])); J["limited"].append(int(limited))
        J["margin"].append(float(at.get("margin", 0)))
        J["armature"].append(float(at.get("armature", 0)))
        J["damping"].append(float(at.get("damping", 0)))
        J["frictionloss"].append(float(at.get("frictionloss", 0)))
        J["ref"].append(float(at.get("ref", 0))); J["springref"].append(float(at.get("springref", 0)))
        J["solref_lim"].append(_floats(at.get("solreflimit"), 2, SOLREF))
        J["solimp_lim"].append(_floats(at.get("solimplimit"), 5, SOLIMP))
        J["solref_fri"].append(_floats(at.get("solreffriction"), 2, SOLREF))
        J["solimp_fri"].append(_floats(at.get("solimpfriction"), 5, SOLIMP))

    def add_site(at, bodyid):
        S["name"].append(at.get("name", "")); S["bodyid"].append(bodyid)
        S["pos"].append(_floats(at.get("pos"), 3, [0, 0, 0])); S["quat"].append(_orientation(at, eulerseq))
        S["type"].append(_GEOM_TYPES[at.get("type", "sphere")])
        S["size"].append(_floats(at.get("size"), 3, [0.005, 0.005, 0.005]))

    def add_body(elem, parentid, childclass):
        bid = len(B["name"])
        childclass = elem.get("childclass") or childclass
        B["name"].append(elem.get("name", "")); B["parentid"].append(parentid)
        B["pos"].append(_floats(elem.get("pos"), 3, [0, 0, 0])); B["quat"].append(_orientation(elem.attrib, eulerseq))
        B["mocap"].append(int(_bool(elem.get("mocap"))))
        if B["mocap"][-1] and parentid != 0:
            raise ValueError("mocap body %r must be a child of the world" % elem.get("name"))
        for k in ("ipos", "iquat", "mass", "inertia"):
            B[k].append(None)
        B["jntadr"].append(len(J["name"])); B["geomadr"].append(len(G["name"]))
        for child in elem:
            if child.tag == "joint":
                add_joint(defaults.resolve(child, childclass), bid)
            elif child.tag == "freejoint":
                add_joint(dict(type="free", name=child.get("name", "")), bid)
            elif child.tag == "geom":
                add_geom(defaults.resolve(child, childclass), bid)
            elif child.tag == "site":
                add_site(defaults.resolve(child, childclass), bid)
        B["jntnum"].append(len(J["name"]) - B["jntadr"][bid])
        B["geomnum"].append(len(G["name"]) - B["geomadr"][bid])
        if B["jntnum"][bid] == 0:
            B["jntadr"][bid] = -1
        if B["geomnum"][bid] == 0:
            B["geomadr"][bid] = -1
        inert = elem.find("inertial")
        if inert is not None:
            B["ipos"][bid] = _floats(inert.get("pos"), 3, [0, 0, 0])
            B["mass"][bid] = float(inert.get("mass"))
            if inert.get("fullinertia") is not None:
                f = _floats(inert.get("fullinertia"))
                I = np.array([[f[0], f[3], f[4]], [f[3], f[1], f[5]], [f[4], f[5], f[2]]])
                w, V = np.linalg.eigh(I)
                order = np.argsort(-w)
                V = V[:, order]
                if np.linalg.det(V) < 0:
                    V[:, 2] = -V[:, 2]
                B["inertia"][bid] = w[order]
                B["iquat"][bid] = mat2q(V)
            else:
                B["inertia"][bid] = _floats(inert.get("diaginertia"), 3, [0, 0, 0])
                B["iquat"][bid] = _orientation(inert.attrib, eulerseq)
        else:
            # inertia from geoms (inertiafromgeom="auto")
            idx = range(B["geomadr"][bid], B["geomadr"][bid] + B["geomnum"][bid]) if B["geomnum"][bid] else []
            mtot = sum(G["mass"][g] for g in idx)
            if mtot <= 0:
                B["ipos"][bid], B["iquat"][bid] = np.zeros(3), np.array([1.0, 0, 0, 0])
                B["mass"][bid], B["inertia"][bid] = 0.0, np.zeros(3)
            else:
                com = sum(G["mass"][g] * G["pos"][g] for g in idx) / mtot
                I = np.zeros((3, 3))
                for g in idx:
                    R = q2mat(G["quat"][g])
                    d = G["pos"][g] - com
                    I += R @ np.diag(G["inertia"][g]) @ R.T + G["mass"][g] * (d @ d * np.eye(3) - np.outer(d, d))
                off = np.abs(I - np.diag(np.diag(I))).max()
                if off < 1e-14 * max(1.0, np.abs(I).max()) or off < MINVAL:
                    w, V = np.diag(I).copy(), np.eye(3)
                else:
                    w, V = np.linalg.eigh(I)
                    order = np.argsort(-w)
                    w, V = w[order], V[:, order]
                    if np.linalg.det(V) < 0:
                        V[:, 2] = -V[:, 2]
                B["ipos"][bid], B["iquat"][bid] = com, mat2q(V)
                B["mass"][bid], B["inertia"][bid] = mtot, w
        for child in elem:
            if child.tag == "body":
                add_body(child, bid, childclass)

    for wb in root.findall("worldbody"):
        g0 = len(G["name"])
        for child in wb:
            if child.tag == "geom":
                add_geom(defaults.resolve(child, None), 0)
            elif child.tag == "site":
                add_site(defaults.resolve(child, None), 0)
        if len(G["name"]) > g0 and B["geomadr"][0] < 0:
            B["geomadr"][0] = g0
        B["geomnum"][0] += len(G["name"]) - g0
        for child in wb:
            if child.tag == "body":
                add_body(child, 0, None)

    nbody, njnt, ngeom, nsite = len(B["name"]), len(J["name"]), len(G["name"]), len(S["name"])

    # geoms must be contiguous per body for body_geomadr/num: re-sort by body (stable)
    order = sorted(range(ngeom), key=lambda g: G["bodyid"][g])
    if order != list(range(ngeom)):
        for k in G:
            G[k] = [G[k][g] for g in order]
    for b in range(nbody):
        idx = [g for g in range(ngeom) if G["bodyid"][g] == b]
        B["geomadr"][b], B["geomnum"][b] = (idx[0], len(idx)) if idx else (-1, 0)

    # ------------------------------------------------------------ dofs / qpos layout
    jnt_qposadr, jnt_dofadr, dof_bodyid, dof_jntid, qpos0 = [], [], [], [], []
    nq = nv = 0
    for j in range(njnt):
        jnt_qposadr.append(nq); jnt_dofadr.append(nv)
        t = J["type"][j]
        nqj, nvj = {JNT_FREE: (7, 6), JNT_BALL: (4, 3), JNT_SLIDE: (1, 1), JNT_HINGE: (1, 1)}[t]
        if t == JNT_FREE:
            b = J["bodyid"][j]
            qpos0 += list(B["pos"][b]) + list(B["quat"][b])
        elif t == JNT_BALL:
            qpos0 += [1, 0, 0, 0]
        else:
            qpos0.append(J["ref"][j])
        dof_bodyid += [J["bodyid"][j]] * nvj; dof_jntid += [j] * nvj
        nq += nqj; nv += nvj
    body_dofadr, body_dofnum = [], []
    for b in range(nbody):
        d = [i for i in range(nv) if dof_bodyid[i] == b]
        body_dofadr.append(d[0] if d else -1); body_dofnum.append(len(d))
    # dof_parentid: previous dof in the same body, else last dof of nearest ancestor with dofs
    dof_parentid = []
    for i in range(nv):
        b = dof_bodyid[i]
        if i > body_dofadr[b]:
            dof_parentid.append(i - 1)
            continue
        p = B["parentid"][b]
        while p > 0 and body_dofnum[p] == 0:
            p = B["parentid"][p]
        dof_parentid.append(body_dofadr[p] + body_dofnum[p] - 1 if p > 0 and body_dofnum[p] else -1)
    body_weldid, body_rootid = [0] * nbody, [0] * nbody
    for b in range(1, nbody):
        p = B["parentid"][b]
        body_weldid[b] = b if B["jntnum"][b] > 0 else body_weldid[p]
        body_rootid[b] = b if p == 0 else body_rootid[p]

    m = CompiledModel()
    A = m.arrays
    f64 = lambda x, shape=None: np.asarray(x, dtype=np.float64).reshape(shape) if shape else np.asarray(x, dtype=np.float64)
    i32 = lambda x: np.asarray(x, dtype=np.int32)
    A["opt_timestep"] = f64([opt["timestep"]]); A["opt_gravity"] = f64(opt["gravity"])
    A["opt_tolerance"] = f64([opt["tolerance"]]); A["opt_impratio"] = f64([opt["impratio"]])
    A["opt_ls_tolerance"] = f64([opt["ls_tolerance"]]); A["opt_mpr_tolerance"] = f64([opt["mpr_tolerance"]])
    A["opt_int"] = i32([opt["iterations"], opt["cone"], opt["ls_iterations"], opt["mpr_iterations"], opt["noslip_iterations"]])
    A["size_int"] = i32([size["njmax"], size["nconmax"], size["nuserdata"], size["nuser_actuator"]])
    A["body_parentid"] = i32(B["parentid"]); A["body_rootid"] = i32(body_rootid); A["body_weldid"] = i32(body_weldid)
    A["body_jntadr"] = i32(B["jntadr"]); A["body_jntnum"] = i32(B["jntnum"])
    A["body_dofadr"] = i32(body_dofadr); A["body_dofnum"] = i32(body_dofnum)
    A["body_geomadr"] = i32(B["geomadr"]); A["body_geomnum"] = i32(B["geomnum"])
    A["body_pos"] = f64(B["pos"], (nbody, 3)); A["body_quat"] = f64(B["quat"], (nbody, 4))
    A["body_ipos"] = f64(B["ipos"], (nbody, 3)); A["body_iquat"] = f64([qnorm(q) for q in B["iquat"]], (nbody, 4))
    A["body_mass"] = f64(B["mass"]); A["body_inertia"] = f64(B["inertia"], (nbody, 3))
    # mocap bodies (mjModel.body_mocapid: index into data.mocap_pos / mocap_quat, -1 for ordinary bodies)
    mocapid, nmocap = [], 0
    for b in range(nbody):
        if B["mocap"][b]:
            if B["jntnum"][b]:
                raise ValueError("mocap body %r has joints" % B["name"][b])
            mocapid.append(nmocap); nmocap += 1
        else:
            mocapid.append(-1)
    A["body_mocapid"] = i32(mocapid)
    A["jnt_type"] = i32(J["type"]); A["jnt_qposadr"] = i32(jnt_qposadr); A["jnt_dofadr"] = i32(jnt_dofadr)
    A["jnt_bodyid"] = i32(J["bodyid"]); A["jnt_pos"] = f64(J["pos"], (njnt, 3)); A["jnt_axis"] = f64(J["axis"], (njnt, 3))
    A["jnt_stiffness"] = f64(J["stiffness"]); A["jnt_range"] = f64(J["range"], (njnt, 2))
    A["jnt_limited"] = i32(J["limited"]); A["jnt_margin"] = f64(J["margin"])
    A["jnt_solref"] = f64(J["solref_lim"], (njnt, 2)); A["jnt_solimp"] = f64(J["solimp_lim"], (njnt, 5))
    A["dof_bodyid"] = i32(dof_bodyid); A["dof_jntid"] = i32(dof_jntid); A["dof_parentid"] = i32(dof_parentid)
    A["dof_armature"] = f64([J["armature"][j] for j in dof_jntid]); A["dof_damping"] = f64([J["damping"][j] for j in dof_jntid])
    A["dof_frictionloss"] = f64([J["frictionloss"][j] for j in dof_jntid])
    A["dof_solref"] = f64([J["solref_fri"][j] for j in dof_jntid], (nv, 2))
    A["dof_solimp"] = f64([J["solimp_fri"][j] for j in dof_jntid], (nv, 5))
    A["qpos0"] = f64(qpos0)
    qspring = np.array(qpos0, dtype=float)
    for j in range(njnt):
        if J["type"][j] in (JNT_SLIDE, JNT_HINGE):
            qspring[jnt_qposadr[j]] = J["springref"][j]
    A["qpos_spring"] = qspring
    A["geom_type"] = i32(G["type"]); A["geom_bodyid"] = i32(G["bodyid"]); A["geom_dataid"] = i32(G["dataid"])
    A["geom_contype"] = i32(G["contype"]); A["geom_conaffinity"] = i32(G["conaffinity"]); A["geom_condim"] = i32(G["condim"])
    A["geom_size"] = f64(G["size"], (ngeom, 3)); A["geom_rbound"] = f64(G["rbound"])
    A["geom_pos"] = f64(G["pos"], (ngeom, 3)); A["geom_quat"] = f64([qnorm(q) for q in G["quat"]], (ngeom, 4))
    A["geom_friction"] = f64(G["friction"], (ngeom, 3)); A["geom_margin"] = f64(G["margin"]); A["geom_gap"] = f64(G["gap"])
    A["geom_solmix"] = f64(G["solmix"]); A["geom_solref"] = f64(G["solref"], (ngeom, 2)); A["geom_solimp"] = f64(G["solimp"], (ngeom, 5))
    A["site_bodyid"] = i32(S["bodyid"]); A["site_pos"] = f64(S["pos"], (nsite, 3)) if nsite else np.zeros((0, 3))
    A["site_quat"] = f64([qnorm(q) for q in S["quat"]], (nsite, 4)) if nsite else np.zeros((0, 4))
    A["site_type"] = i32(S["type"]); A["site_size"] = f64(S["size"], (nsite, 3)) if nsite else np.zeros((0, 3))
    vadr, verts = [], []
    for md in mesh_data:
        vadr.append(sum(len(v) for v in verts)); verts.append(md["vert"])
    A["mesh_vertadr"] = i32(vadr); A["mesh_vertnum"] = i32([len(v) for v in verts])
    A["mesh_vert"] = np.concatenate(verts).astype(np.float32) if verts else np.zeros((0, 3), np.float32)
    m.names = dict(body=B["name"], joint=J["name"], geom=G["name"], site=S["name"], mesh=list(used_meshes))

    # ------------------------------------------------------------ contact excludes
    excl = []
    for c in root.findall("contact"):
        for e in c.findall("exclude"):
            b1, b2 = m.name2id("body", e.get("body1")), m.name2id("body", e.get("body2"))
            excl.append((min(b1, b2) << 16) + max(b1, b2))
    A["exclude_signature"] = i32(excl)

    # ------------------------------------------------------------ tendons
    T = dict(name=[], adr=[], num=[], limited=[], range=[], margin=[], stiffness=[], damping=[], frictionloss=[],
             springlength=[], solref_lim=[], solimp_lim=[], solref_fri=[], solimp_fri=[])
    W = dict(type=[], objid=[], prm=[])
    for tsec in root.findall("tendon"):
        for te in tsec:
            at = defaults.resolve(te, None)
            T["name"].append(at.get("name", "")); T["adr"].append(len(W["type"]))
            T["limited"].append(int(_bool(at.get("limited")))); T["range"].append(_floats(at.get("range"), 2, [0, 0]))
            T["margin"].append(float(at.get("margin", 0))); T["stiffness"].append(float(at.get("stiffness", 0)))
            T["damping"].append(float(at.get("damping", 0))); T["frictionloss"].append(float(at.get("frictionloss", 0)))
            T["springlength"].append(float(at.get("springlength", -1)))
            T["solref_lim"].append(_floats(at.get("solreflimit"), 2, SOLREF)); T["solimp_lim"].append(_floats(at.get("solimplimit"), 5, SOLIMP))
            T["solref_fri"].append(_floats(at.get("solreffriction"), 2, SOLREF)); T["solimp_fri"].append(_floats(at.get("solimpfriction"), 5, SOLIMP))
            for w in te:
                if w.tag == "joint":
                    W["type"].append(WRAP_JOINT); W["objid"].append(m.name2id("joint", w.get("joint"))); W["prm"].append(float(w.get("coef")))
                elif w.tag == "site":
                    W["type"].append(WRAP_SITE); W["objid"].append(m.name2id("site", w.get("site"))); W["prm"].append(0.0)
                elif w.tag == "geom":
                    gid = m.name2id("geom", w.get("geom"))
                    gt = G["type"][gid]
                    if gt not in (GEOM_SPHERE, GEOM_CYLINDER):
                        raise ValueError("tendon can only wrap spheres and cylinders")
                    W["type"].append(WRAP_SPHERE if gt == GEOM_SPHERE else WRAP_CYLINDER); W["objid"].append(gid)
                    W["prm"].append(float(m.name2id("site", w.get("sidesite"))) if w.get("sidesite") else -1.0)
                elif w.tag == "pulley":
                    W["type"].append(WRAP_PULLEY); W["objid"].append(-1); W["prm"].append(float(w.get("divisor")))
            T["num"].append(len(W["type"]) - T["adr"][-1])
    ntendon = len(T["name"])
    A["tendon_adr"] = i32(T["adr"]); A["tendon_num"] = i32(T["num"]); A["tendon_limited"] = i32(T["limited"])
    A["tendon_range"] = f64(T["range"], (ntendon, 2)) if ntendon else np.zeros((0, 2))
    A["tendon_margin"] = f64(T["margin"]); A["tendon_stiffness"] = f64(T["stiffness"]); A["tendon_damping"] = f64(T["damping"])
    A["tendon_frictionloss"] = f64(T["frictionloss"]); A["tendon_lengthspring"] = f64(T["springlength"])
    A["tendon_solref_lim"] = f64(T["solref_lim"], (ntendon, 2)) if ntendon else np.zeros((0, 2))
    A["tendon_solimp_lim"] = f64(T["solimp_lim"], (ntendon, 5)) if ntendon else np.zeros((0, 5))
    A["tendon_solref_fri"] = f64(T["solref_fri"], (ntendon, 2)) if ntendon else np.zeros((0, 2))
    A["tendon_solimp_fri"] = f64(T["solimp_fri"], (ntendon, 5)) if ntendon else np.zeros((0, 5))
    A["wrap_type"] = i32(W["type"]); A["wrap_objid"] = i32(W["objid"]); A["wrap_prm"] = f64(W["prm"])
    m.names["tendon"] = T["name"]

    # ------------------------------------------------------------ actuators
    U = dict(name=[], trntype=[], trnid=[], gear=[], ctrllimited=[], ctrlrange=[], forcelimited=[], forcerange=[],
             gainprm=[], biasprm=[], gaintype=[], biastype=[], user=[])
    for asec in root.findall("actuator"):
        for ae in asec:
            at = defaults.resolve(ae, None)
            if ae.tag == "motor":        # MJCF shortcuts (MuJoCo XML reference, actuator/motor and actuator/position): a <general> with these settings
                at = dict(at, gaintype="fixed", biastype="none", gainprm="1")
            elif ae.tag == "position":
                kp = float(_floats(at.get("kp"), 1, [1])[0])
                at = dict(at, gaintype="fixed", biastype="affine", gainprm=repr(kp), biasprm="0 %r 0" % (-kp))
            elif ae.tag != "general":
                raise NotImplementedError("actuator shortcut <%s>" % ae.tag)
            U["name"].append(at.get("name", ""))
            if at.get("joint") is not None:
                U["trntype"].append(TRN_JOINT); U["trnid"].append(m.name2id("joint", at["joint"]))
            elif at.get("tendon") is not None:
                U["trntype"].append(TRN_TENDON); U["trnid"].append(m.name2id("tendon", at["tendon"]))
            else:
                raise NotImplementedError("actuator transmission")
            U["gear"].append(_floats(at.get("gear"), 6, [1, 0, 0, 0, 0, 0])[0])
            U["ctrllimited"].append(int(_bool(at.get("ctrllimited")))); U["ctrlrange"].append(_floats(at.get("ctrlrange"), 2, [0, 0]))
            U["forcelimited"].append(int(_bool(at.get("forcelimited")))); U["forcerange"].append(_floats(at.get("forcerange"), 2, [0, 0]))
            U["gainprm"].append(_floats(at.get("gainprm"), 10, [1] + [0] * 9)); U["biasprm"].append(_floats(at.get("biasprm"), 10, [0] * 10))
            U["gaintype"].append(dict(fixed=0, user=GAIN_USER)[at.get("gaintype", "fixed")])
            U["biastype"].append(dict(none=0, affine=1, user=GAIN_USER)[at.get("biastype", "none")])
            U["user"].append(_floats(at.get("user"), 1, [0])[0])
    nu = len(U["name"])
    A["actuator_trntype"] = i32(U["trntype"]); A["actuator_trnid"] = i32(U["trnid"]); A["actuator_gear"] = f64(U["gear"])
    A["actuator_ctrllimited"] = i32(U["ctrllimited"]); A["actuator_ctrlrange"] = f64(U["ctrlrange"], (nu, 2)) if nu else np.zeros((0, 2))
    A["actuator_forcelimited"] = i32(U["forcelimited"]); A["actuator_forcerange"] = f64(U["forcerange"], (nu, 2)) if nu else np.zeros((0, 2))
    A["actuator_gainprm"] = f64(U["gainprm"], (nu, 10)) if nu else np.zeros((0, 10))
    A["actuator_biasprm"] = f64(U["biasprm"], (nu, 10)) if nu else np.zeros((0, 10))
    A["actuator_gaintype"] = i32(U["gaintype"]); A["actuator_biastype"] = i32(U["biastype"]); A["actuator_user"] = f64(U["user"])
    m.names["actuator"] = U["name"]

    # ------------------------------------------------------------ equality constraints (mjEQ_WELD = 1, mjEQ_JOINT = 2)
    E = dict(name=[], type=[], obj1=[], obj2=[], active=[], solref=[], solimp=[], data=[])
    for esec in root.findall("equality"):
        for ee in esec:
            at = defaults.resolve(ee, None)
            data = np.zeros(7)
            if ee.tag == "weld":
                b1 = m.name2id("body", at["body1"])
                b2 = m.name2id("body", at["body2"]) if at.get("body2") is not None else 0
                rp = _floats(at.get("relpose"), 7, [0, 1, 0, 0, 0, 0, 0])
                if np.abs(rp[3:]).sum() > 0:
                    data = np.concatenate([rp[:3], qnorm(rp[3:])])
                else:
                    data = None   # the pose of body2 relative to body1 at qpos0: filled in below (needs the kinematics)
                E["type"].append(EQ_WELD); E["obj1"].append(b1); E["obj2"].append(b2)
            elif ee.tag == "joint":
                j1 = m.name2id("joint", at["joint1"])
                j2 = m.name2id("joint", at["joint2"]) if at.get("joint2") is not None else -1
                data[:5] = _floats(at.get("polycoef"), 5, [0, 1, 0, 0, 0])
                for j in (j1, j2):
                    if j >= 0 and J["type"][j] not in (JNT_SLIDE, JNT_HINGE):
                        raise ValueError("joint equality needs scalar joints")
                E["type"].append(EQ_JOINT); E["obj1"].append(j1); E["obj2"].append(j2)
            else:
                raise NotImplementedError("equality <%s>" % ee.tag)
            E["name"].append(at.get("name", "")); E["active"].append(int(_bool(at.get("active"), True)))
            E["solref"].append(_floats(at.get("solref"), 2, SOLREF)); E["solimp"].append(_floats(at.get("solimp"), 5, SOLIMP))
            E["data"].append(data)
    neq = len(E["name"])
    A["eq_type"] = i32(E["type"]); A["eq_obj1id"] = i32(E["obj1"]); A["eq_obj2id"] = i32(E["obj2"]); A["eq_active"] = i32(E["active"])
    A["eq_solref"] = f64(E["solref"], (neq, 2)) if neq else np.zeros((0, 2))
    A["eq_solimp"] = f64(E["solimp"], (neq, 5)) if neq else np.zeros((0, 5))
    m.names["equality"] = E["name"]

    # ------------------------------------------------------------ sensors (mjtSensor values of MuJoCo 2.0)
    sens_type, sens_objid, sens_names, sens_dim = [], [], [], []
    for ssec in root.findall("sensor"):
        for se in ssec:
            if se.tag in ("touch", "force", "torque"):
                sens_type.append(dict(touch=SENS_TOUCH, force=SENS_FORCE, torque=SENS_TORQUE)[se.tag])
                sens_objid.append(m.name2id("site", se.get("site"))); sens_dim.append(1 if se.tag == "touch" else 3)
            elif se.tag == "jointpos":
                sens_type.append(SENS_JOINTPOS); sens_objid.append(m.name2id("joint", se.get("joint"))); sens_dim.append(1)
            else:
                raise NotImplementedError("sensor <%s>" % se.tag)
            sens_names.append(se.get("name", ""))
    A["sensor_type"] = i32(sens_type); A["sensor_objid"] = i32(sens_objid); A["sensor_dim"] = i32(sens_dim)
    A["sensor_adr"] = i32(np.concatenate([[0], np.cumsum(sens_dim)[:-1]])) if sens_dim else i32([])
    m.names["sensor"] = sens_names

    A["dims"] = i32([nq, nv, nu, nbody, njnt, ngeom, nsite, ntendon, len(W["type"]), len(mesh_data),
                     len(A["mesh_vert"]), len(excl), len(sens_type)])

    from robogym_amd.mujoco.setconst import kinematics, set_constants

    # weld constraints without an explicit relpose: the pose of body2 relative to body1 in the reference configuration
    if neq:
        kin = kinematics(m, A["qpos0"])
        for e in range(neq):
            if E["data"][e] is None:
                b1, b2 = E["obj1"][e], E["obj2"][e]
                rel_q = qmul(qconj(kin["xquat"][b1]), kin["xquat"][b2])
                rel_p = kin["xmat"][b1].T @ (kin["xpos"][b2] - kin["xpos"][b1])
                E["data"][e] = np.concatenate([rel_p, qnorm(rel_q)])
    A["eq_data"] = f64(E["data"], (neq, 7)) if neq else np.zeros((0, 7))
    A["nmocap"] = i32([nmocap])
    set_constants(m)
    return m
