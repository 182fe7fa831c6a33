"""GPU tests written after round 4's last GPU call (their helpers live next to their CPU twins; what ran instead: the emulation harness, profiles/r04_emul_gpu_protocols.txt).
The file sorts last, and tests/conftest.py orders the GPU run dactyl -> rearrange -> this file, so that under `pytest -x` a first-run surprise here cannot hide
the tests of the workloads that were validated on the MI355X."""
import pytest

from tests.test_rearrange_env import _impulse_response_on_the_kernel
from tests.test_rearrange_ycb import _per_episode_checks


@pytest.mark.gpu
def test_mocap_ik_impulse_response_on_the_kernel_gpu():
    """The reference's impulse-response pin (envs/rearrange/tests/test_rearrange_sim.py:135-230) stepped on BatchedBlockRearrangeEnv itself, no oracle in the loop."""
    _impulse_response_on_the_kernel(None, "cuda:0", stabilize_steps=100)


@pytest.mark.gpu
def test_ycb_new_object_set_per_episode_gpu():
    """A new object set per episode by slot trading: every output equals a twin's with pinned slots, bit for bit, through reset, episode ends and the reset recipe."""
    B = 64
    env, ended, started, tables = _per_episode_checks(None, "cuda:0", B=B, steps=22, seed=3, stabilize_steps=4, n_random_initial_steps=1, settle_steps=4)
    assert ended >= B and started >= B                   # every env timed out at least once (8 steps) and came back (4 + 1 + 4 recipe steps)
    assert env.episodes_moved > B // 4 and len({tuple(t) for t in tables}) >= 2      # (dealt at reset, dealt again when the goals time out together on step 8)
