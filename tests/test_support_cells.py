"""Host-side check of the hull support tables (kernel_tables.mesh_support_cells): for random directions the arg-max
over a direction cell's candidate list equals the arg-max over ALL hull vertices — in fp32, with the kernel's
cell arithmetic (dir_cell in rg_kernel.h) and its fused dot product, lowest index on ties."""
import numpy as np

from robogym_amd.mujoco.kernel_tables import CELL_N


def _dir_cell(ld):
    """numpy restatement of dir_cell() (rg_kernel.h), fp32."""
    ld = ld.astype(np.float32)
    a = np.abs(ld)
    axis = np.where(a[:, 0] >= a[:, 1], np.where(a[:, 0] >= a[:, 2], 0, 2), np.where(a[:, 1] >= a[:, 2], 1, 2))
    idx = np.arange(len(ld))
    mj, u, v = ld[idx, axis], ld[idx, (axis + 1) % 3], ld[idx, (axis + 2) % 3]
    inv = np.float32(0.5 * CELL_N) / np.maximum(np.abs(mj), np.float32(1e-30))
    iu = np.clip((u * inv + np.float32(0.5 * CELL_N)).astype(np.int32), 0, CELL_N - 1)
    iv = np.clip((v * inv + np.float32(0.5 * CELL_N)).astype(np.int32), 0, CELL_N - 1)
    return ((2 * axis + (mj < 0)) * CELL_N + iu) * CELL_N + iv


def _vdot(d, P):
    """fmaf(d.z, z, fmaf(d.y, y, d.x * x)) — emulated in float64 then rounded per step (exact for fp32 operands)."""
    t = (d[:, None, 0].astype(np.float64) * P[None, :, 0]).astype(np.float32)
    t = (d[:, None, 1].astype(np.float64) * P[None, :, 1] + t).astype(np.float32)
    return (d[:, None, 2].astype(np.float64) * P[None, :, 2] + t).astype(np.float32)


def test_cell_lists_reproduce_the_full_argmax(locked_model):
    A = locked_model.arrays
    adr, vidx = A["k_mesh_cell_adr"], A["k_mesh_cell_vidx"]
    ncell = 6 * CELL_N * CELL_N
    V = np.asarray(A["mesh_vert"], dtype=np.float32).reshape(-1, 3)
    rng = np.random.RandomState(0)
    lens = adr & 255
    assert lens.min() >= 1 and lens.mean() < 6       # a handful of candidates per cell instead of 40-909 vertices
    for mi in range(len(A["mesh_vertnum"])):
        P = V[int(A["mesh_vertadr"][mi]): int(A["mesh_vertadr"][mi]) + int(A["mesh_vertnum"][mi])].astype(np.float64)
        d = rng.randn(4000, 3)
        # a third of the directions sit on cell borders / face diagonals, where fp32 binning may go either way
        d[:1300] = np.round(d[:1300] * 2) / 2 + rng.randn(1300, 3) * 1e-7
        d = (d / np.maximum(np.linalg.norm(d, axis=1, keepdims=True), 1e-12)).astype(np.float32)
        d = d[np.abs(d).max(1) > 0]
        dots = _vdot(d, P)
        full = dots.argmax(1)                       # first maximum = lowest index on ties
        cell = _dir_cell(d)
        for k in range(len(d)):
            e = int(adr[mi * ncell + cell[k]])
            cand = vidx[(e >> 8): (e >> 8) + (e & 255)]
            assert (np.diff(cand) > 0).all()
            best = cand[dots[k, cand].argmax()]
            assert best == full[k], (mi, k, d[k], best, full[k])
