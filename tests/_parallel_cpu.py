"""Early-loaded pytest plugin (pytest.ini: `-p tests._parallel_cpu`): the CPU suite (`-m "not gpu"`) is dominated by the fiber-emulation
harness (one mj_step of the rearrange worlds takes seconds there), so it is spread over worker processes with pytest-xdist, one test FILE
per worker at a time (module-scoped fixtures are built once per file).  GPU runs (`-m gpu`) stay in ONE process: the native library must be
loaded by the process the driver watches, and the tests share one device.  No effect when pytest-xdist is missing or `-n` / `-p no:xdist`
is given."""
import os


def pytest_load_initial_conftests(early_config, parser, args):
    try:
        import xdist  # noqa: F401
    except ImportError:
        return
    if any(a == "-n" or a.startswith("-n") or a.startswith("--numprocesses") or a == "no:xdist" for a in args):
        return
    expr = ""
    for i, a in enumerate(args):
        if a == "-m" and i + 1 < len(args):
            expr = args[i + 1]
        elif a.startswith("-m") and len(a) > 2:
            expr = a[2:]
    if "not gpu" not in expr:
        return
    n = max(1, min(6, (os.cpu_count() or 2) - 1))
    if n > 1:
        args[:] = list(args) + ["-n", str(n), "--dist", "loadfile"]
