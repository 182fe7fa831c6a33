"""Batched quaternion helpers (torch) with the conventions of the reference's
`robogym/utils/rotation.py` (w,x,y,z; `quat_normalize` = sign normalisation to w >= 0,
rotation.py:281-286; `quat_difference` 271; `quat_magnitude` 275; `uniform_quat` 440)."""
import numpy as np
import torch


# Hamilton product (the reference's `quat_mul`, utils/rotation.py:234-257), out_k = ((t0 + t1) + t2) + t3 with t_j = sign[j][k] * q0[a[j][k]] * q1[b[j][k]] in the order of the component formula
#   w = w0 w1 - x0 x1 - y0 y1 - z0 z1,  x = w0 x1 + x0 w1 + y0 z1 - z0 y1,  y = w0 y1 + y0 w1 + z0 x1 - x0 z1,  z = w0 z1 + z0 w1 + x0 y1 - y0 x1
# as SIX tensor kernels instead of 29 (16 products, 12 sums, a stack): all 16 products at once, the 16 signed terms picked by one gather, three adds.
# The products and the order of the sums are those of the formula (a - b is a + (-b) exactly), so the result is bit-identical to it; the wrapper stack calls
# this three times per env.step on [B, 4] tensors, where every kernel is launch latency.
_QM_A = ((0, 0, 0, 0), (1, 1, 2, 3), (2, 2, 3, 1), (3, 3, 1, 2))      # [term j][component k]: index into q0
_QM_B = ((0, 1, 2, 3), (1, 0, 0, 0), (2, 3, 1, 2), (3, 2, 3, 1))      # ... into q1
_QM_S = ((1, 1, 1, 1), (-1, 1, 1, 1), (-1, 1, 1, 1), (-1, -1, -1, -1))
_qm_tables = {}


def _qm_table(device, dtype, conj=False):
    key = (str(device), dtype, conj)
    if key not in _qm_tables:
        idx = torch.tensor([4 * _QM_A[j][k] + _QM_B[j][k] for j in range(4) for k in range(4)], dtype=torch.long, device=device)
        # (conj: the second factor is conjugated -- its x, y, z enter with the opposite sign: the terms of q0 * conj(q1), as quat_mul(q0, quat_conjugate(q1)) forms them)
        sign = torch.tensor([float(_QM_S[j][k] * (-1 if conj and _QM_B[j][k] else 1)) for j in range(4) for k in range(4)], dtype=dtype, device=device)
        _qm_tables[key] = (idx, sign)
    return _qm_tables[key]


def quat_mul(q0: torch.Tensor, q1: torch.Tensor, conj: bool = False) -> torch.Tensor:
    """q0 * q1, or q0 * conj(q1) with `conj` (no kernel for the conjugate: its signs are in the term table)."""
    q0, q1 = torch.broadcast_tensors(q0, q1)
    idx, sign = _qm_table(q0.device, q0.dtype, conj)
    prod = (q0[..., :, None] * q1[..., None, :]).reshape(q0.shape[:-1] + (16,))
    t = (prod[..., idx] * sign).reshape(q0.shape[:-1] + (4, 4))
    return ((t[..., 0, :] + t[..., 1, :]) + t[..., 2, :]) + t[..., 3, :]


def quat_conjugate(q: torch.Tensor) -> torch.Tensor:
    return torch.cat([q[..., :1], -q[..., 1:]], dim=-1)   # (no host-built constant: a tensor made from a Python list is a blocking copy behind the queued kernels)


def quat_normalize(q: torch.Tensor) -> torch.Tensor:
    """Representative with w >= 0 (w == 0 keeps its sign), as the reference."""
    return torch.where(q[..., :1] < 0, -q, q)      # (three kernels; -q is q * -1 exactly)


def quat_difference(q: torch.Tensor, p: torch.Tensor) -> torch.Tensor:
    return quat_normalize(quat_mul(q, p, conj=True))


def quat_magnitude(q: torch.Tensor) -> torch.Tensor:
    return 2 * torch.arccos(torch.clamp(q[..., 0], -1.0, 1.0))


# ---------------------------------------------------------------- the 24 axis-aligned cube orientations
def parallel_quats_np() -> np.ndarray:
    """The rotation group of the cube as unit quaternions (w >= 0): what `cube_utils.PARALLEL_QUATS`
    (cube_utils.py:8-11, from rotation.get_parallel_rotations, rotation.py:393-408) enumerates through Euler angles.
    Built here from the group's structure: the identity, the 90/180/270 degree turns about the three face axes, the
    120/240 degree turns about the four body diagonals and the 180 degree turns about the six edge axes.  The ORDER is
    this module's own (lexicographic); LockedParallelGoal draws uniformly from the set, so only the set matters
    (tests/test_golden.py asserts set equality with the reference's table)."""
    h, r = 0.5, np.sqrt(0.5)
    quats = [(1.0, 0.0, 0.0, 0.0)]
    for axis in range(3):
        e = [0.0, 0.0, 0.0]; e[axis] = 1.0
        quats += [(r, r * e[0], r * e[1], r * e[2]), (r, -r * e[0], -r * e[1], -r * e[2]), (0.0, e[0], e[1], e[2])]
    quats += [(h, sx * h, sy * h, sz * h) for sx in (1, -1) for sy in (1, -1) for sz in (1, -1)]
    for a, b in ((0, 1), (0, 2), (1, 2)):
        for sb in (1, -1):
            e = [0.0, 0.0, 0.0]; e[a] = r; e[b] = sb * r
            quats.append((0.0, e[0], e[1], e[2]))
    out = np.array(sorted(quats), dtype=np.float64)
    assert out.shape == (24, 4)
    return out


def quat2mat(q: torch.Tensor) -> torch.Tensor:
    """rotation.quat2mat (utils/rotation.py:202-225) for quaternions [..., 4] -> [..., 3, 3] (identity where the quaternion is ~0)."""
    w, x, y, z = q.unbind(-1)
    nq = (q * q).sum(-1)
    s = torch.where(nq > 2.220446049250313e-16, 2.0 / nq.clamp_min(1.0e-30), torch.zeros_like(nq))
    X, Y, Z = x * s, y * s, z * s
    wX, wY, wZ, xX, xY, xZ, yY, yZ, zZ = w * X, w * Y, w * Z, x * X, x * Y, x * Z, y * Y, y * Z, z * Z
    m = torch.stack([1.0 - (yY + zZ), xY - wZ, xZ + wY, xY + wZ, 1.0 - (xX + zZ), yZ - wX, xZ - wY, yZ + wX, 1.0 - (xX + yY)], -1).reshape(q.shape[:-1] + (3, 3))
    eye = torch.eye(3, dtype=q.dtype, device=q.device).expand_as(m)
    return torch.where((nq > 2.220446049250313e-16)[..., None, None], m, eye)


def vectors2quat_to_z(v: torch.Tensor) -> torch.Tensor:
    """rotation.vectors2quat(v, ez) (utils/rotation.py:469-486), batched over the leading dimensions: the shortest-arc rotation of v onto the world's z;
    v = -|v| ez: half a turn about any_orthogonal(v) (:461-466)."""
    n = v.norm(dim=-1)
    w = n + v[..., 2]
    q = torch.stack([w, v[..., 1], -v[..., 0], torch.zeros_like(w)], -1)            # cross(v, ez) = (vy, -vx, 0)
    qn = q.norm(dim=-1, keepdim=True)
    # the antiparallel case: promising axis = the unit vector of v's smallest component (np.abs(vec).argmin()), orthogonal = cross(v, axis), normalised
    k = v.abs().argmin(dim=-1)
    axis = torch.nn.functional.one_hot(k, 3).to(v.dtype)
    orth = torch.cross(v, axis, dim=-1)
    orth = orth / orth.norm(dim=-1, keepdim=True).clamp_min(1.0e-30)
    flip = torch.cat([torch.zeros_like(w)[..., None], orth], -1)
    small = qn < 1.0e-6
    q = torch.where(small, flip, q)
    return quat_normalize(q / q.norm(dim=-1, keepdim=True).clamp_min(1.0e-30))


def normalize_angles(a: torch.Tensor) -> torch.Tensor:
    """rotation.normalize_angles (utils/rotation.py:372-378): into [-pi, pi)."""
    return (a + np.pi) % (2 * np.pi) - np.pi
