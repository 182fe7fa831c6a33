"""Round 6, CPU tier: tests that live in a file of their own so that the file-granular test distribution (tests/_parallel_cpu.py) runs them beside, not behind,
the other emulated ycb runs."""
from tests.test_rearrange_ycb import _per_episode_checks


def test_ycb_new_object_set_per_episode_device_trading_emul(emul_lib):
    """Round 6: with the reset recipe on the device (`device_reset=True`) the slots are traded on the device as well -- fixed-shape tensor ops on the
    recipe kernel's stage / ended arrays, no readback inside step() -- and the twin protocol holds the same way: every output is the pinned twin's through the env -> slot table."""
    env, ended, started, tables = _per_episode_checks(emul_lib, "cpu", B=2, steps=10, seed=8, n_substeps=1, stabilize_steps=1, n_random_initial_steps=0, settle_steps=0, device_reset=True)
    assert env._device_trade and ended == 2 and started == 2          # (B = 2 keeps the emulated run at the length of the host-trading twin test; B = 64 on the GPU)
    assert all(sorted(t.tolist()) == [0, 1] for t in tables)
