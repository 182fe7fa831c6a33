"""Batched quaternion helpers (torch) with the conventions of the reference's
`robogym/utils/rotation.py` (w,x,y,z; `quat_normalize` = sign normalisation to w >= 0,
rotation.py:281-286; `quat_difference` 271; `quat_magnitude` 275; `uniform_quat` 440)."""
import itertools

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
    return q * q.new_tensor([1.0, -1.0, -1.0, -1.0])


def quat_normalize(q: torch.Tensor) -> torch.Tensor:
    """Representative with w >= 0 (w == 0 keeps its sign), as the reference."""
    sign = torch.where(q[..., :1] < 0, -torch.ones_like(q[..., :1]), torch.ones_like(q[..., :1]))
    return q * sign


def quat_difference(q: torch.Tensor, p: torch.Tensor) -> torch.Tensor:
    return quat_normalize(quat_mul(q, quat_conjugate(p)))


def quat_magnitude(q: torch.Tensor) -> torch.Tensor:
    return 2 * torch.arccos(torch.clamp(q[..., 0], -1.0, 1.0))


# ---------------------------------------------------------------- numpy helpers used at goal sampling time
def euler2quat_np(euler):
    ai, aj, ak = euler[2] / 2, -euler[1] / 2, euler[0] / 2
    si, sj, sk = np.sin(ai), np.sin(aj), np.sin(ak)
    ci, cj, ck = np.cos(ai), np.cos(aj), np.cos(ak)
    cc, cs, sc, ss = ci * ck, ci * sk, si * ck, si * sk
    return np.array([cj * cc + sj * ss, cj * cs - sj * sc, -(cj * ss + sj * cc), cj * sc - sj * cs])


def euler2mat_np(euler):
    ai, aj, ak = -euler[2], -euler[1], -euler[0]
    si, sj, sk = np.sin(ai), np.sin(aj), np.sin(ak)
    ci, cj, ck = np.cos(ai), np.cos(aj), np.cos(ak)
    cc, cs, sc, ss = ci * ck, ci * sk, si * ck, si * sk
    m = np.empty((3, 3))
    m[2, 2] = cj * ck; m[2, 1] = sj * sc - cs; m[2, 0] = sj * cc + ss
    m[1, 2] = cj * sk; m[1, 1] = sj * ss + cc; m[1, 0] = sj * cs - sc
    m[0, 2] = -sj; m[0, 1] = cj * si; m[0, 0] = cj * ci
    return m


def mat2euler_np(mat):
    cy = np.sqrt(mat[2, 2] * mat[2, 2] + mat[1, 2] * mat[1, 2])
    if cy > np.finfo(np.float64).eps * 4.0:
        return np.array([-np.arctan2(mat[1, 2], mat[2, 2]), -np.arctan2(-mat[0, 2], cy), -np.arctan2(mat[0, 1], mat[0, 0])])
    return np.array([0.0, -np.arctan2(-mat[0, 2], cy), -np.arctan2(-mat[1, 0], mat[1, 1])])


def parallel_quats_np():
    """The 24 axis-aligned orientations, enumerated as `rotation.get_parallel_rotations`
    (rotation.py:393-408) and converted as `cube_utils.PARALLEL_QUATS` (cube_utils.py:8-11)."""
    mult90 = [0, np.pi / 2, -np.pi / 2, np.pi]
    found = []
    for euler in itertools.product(mult90, repeat=3):
        canonical = mat2euler_np(euler2mat_np(np.array(euler)))
        canonical = np.round(canonical / (np.pi / 2))
        if canonical[0] == -2:
            canonical[0] = 2
        if canonical[2] == -2:
            canonical[2] = 2
        canonical *= np.pi / 2
        if all((canonical != r).any() for r in found):
            found.append(canonical)
    assert len(found) == 24
    quats = np.array([euler2quat_np(r) for r in found])
    quats[quats[:, 0] < 0] *= -1
    return quats
