"""Batched quaternion helpers (torch) with the conventions of the reference's
`robogym/utils/rotation.py` (w,x,y,z; `quat_normalize` = sign normalisation to w >= 0,
rotation.py:281-286; `quat_difference` 271; `quat_magnitude` 275; `uniform_quat` 440)."""
import numpy as np
import torch


def quat_mul(q0: torch.Tensor, q1: torch.Tensor) -> torch.Tensor:
    w0, x0, y0, z0 = q0.unbind(-1)
    w1, x1, y1, z1 = q1.unbind(-1)
    return torch.stack(
        [
            w0 * w1 - x0 * x1 - y0 * y1 - z0 * z1,
            w0 * x1 + x0 * w1 + y0 * z1 - z0 * y1,
            w0 * y1 + y0 * w1 + z0 * x1 - x0 * z1,
            w0 * z1 + z0 * w1 + x0 * y1 - y0 * x1,
        ],
        dim=-1,
    )


def quat_conjugate(q: torch.Tensor) -> torch.Tensor:
    return torch.cat([q[..., :1], -q[..., 1:]], dim=-1)   # (no host-built constant: a tensor made from a Python list is a blocking copy behind the queued kernels)


def quat_normalize(q: torch.Tensor) -> torch.Tensor:
    """Representative with w >= 0 (w == 0 keeps its sign), as the reference."""
    sign = torch.where(q[..., :1] < 0, -torch.ones_like(q[..., :1]), torch.ones_like(q[..., :1]))
    return q * sign


def quat_difference(q: torch.Tensor, p: torch.Tensor) -> torch.Tensor:
    return quat_normalize(quat_mul(q, quat_conjugate(p)))


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
