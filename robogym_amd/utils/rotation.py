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
