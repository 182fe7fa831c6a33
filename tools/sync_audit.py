"""Host/device synchronisation points inside one wrapped env.step (torch.cuda.set_sync_debug_mode("warn")): every one of them
stalls the launch queue, so the wrapper stack's ~150 small kernels stop hiding behind the physics kernel.
    python tools/sync_audit.py [B]"""
import collections
import os
import sys
import time
import traceback
import warnings

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from robogym_amd.envs.dactyl.locked import make_env  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
gen = torch.Generator(device="cuda:0"); gen.manual_seed(0)
AUTO = len(sys.argv) > 2 and sys.argv[2] == "auto"   # episode ends inside the loop: pipelined in-step resets + the wrappers' auto_reset
CASES = ((("make_env(pipelined_reset=True), goals time out after 60 steps", dict(constants={"max_timesteps_per_goal": 60}, pipelined_reset=True)),) if AUTO
         else (("make_env()", dict()), ("make_simple_env", dict(apply_wrappers=False))))
for label, kw in CASES:
    env = make_env(batch_size=B, device="cuda:0", starting_seed=1, **kw)
    env.reset()
    wrapped = kw.get("apply_wrappers", True)
    act = (lambda: torch.randint(0, 11, (B, 20), generator=gen, device="cuda:0")) if wrapped else (lambda: torch.rand((B, 20), generator=gen, device="cuda:0") * 2 - 1)
    for _ in range(70 if AUTO else 3):
        env.step(act())
    torch.cuda.synchronize()
    sites = collections.Counter()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

    def showwarning(message, category, filename, lineno, file=None, line=None):
        if "synchroniz" not in str(message):
            return
        for fr in reversed(traceback.extract_stack()):
            if fr.filename.startswith(root) and "sync_audit" not in fr.filename:
                sites["%s:%d %s" % (os.path.relpath(fr.filename, root), fr.lineno, (fr.line or "").strip()[:110])] += 1
                break
        else:
            sites["(outside the repo) %s:%d" % (filename, lineno)] += 1

    old = warnings.showwarning
    warnings.showwarning = showwarning
    warnings.simplefilter("always")
    torch.cuda.set_sync_debug_mode("warn")
    for _ in range(5 if AUTO else 1):
        env.step(act())
    torch.cuda.set_sync_debug_mode("default")
    warnings.showwarning = old
    torch.cuda.synchronize()
    # host time of a step (enqueue only) vs wall time per step
    t0 = time.perf_counter()
    for _ in range(10):
        env.step(act())
    t_host = (time.perf_counter() - t0) / 10
    torch.cuda.synchronize(); t_wall = (time.perf_counter() - t0) / 10
    print("%s B=%d: %d synchronisation points in the audited step(s); host returns after %.2f ms per step, wall %.2f ms per step" % (label, B, sum(sites.values()), 1e3 * t_host, 1e3 * t_wall))
    for k, v in sites.most_common():
        print("   %2d x %s" % (v, k))
    del env
