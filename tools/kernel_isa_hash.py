"""sha256 of the gfx950 machine code of every kernel in a built librgstep.so (the code object is taken out of the .hip_fatbin offload bundle, the bytes of each
kernel symbol out of its .text): tells whether an edit of one kernel's source changed another kernel's code.

    python tools/kernel_isa_hash.py [robogym_amd/csrc/librgstep.so] [substring of the kernel names to print]
    python tools/kernel_isa_hash.py --compare before.so after.so [substring]
    python tools/kernel_isa_hash.py --mix [substring]        static instruction mix per function (spill loads / stores, LDS, memory, waits)"""
import hashlib
import os
import struct
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def code_object(so_path):
    data = open(so_path, "rb").read()
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    at = data.find(magic)
    assert at >= 0, "no uncompressed offload bundle in " + so_path
    n = struct.unpack_from("<Q", data, at + len(magic))[0]
    p = at + len(magic) + 8
    for _ in range(n):
        off, size, tl = struct.unpack_from("<QQQ", data, p)
        triple = data[p + 24:p + 24 + tl].decode()
        p += 24 + tl
        if "gfx950" in triple:
            return data[at + off:at + off + size]
    raise RuntimeError("no gfx950 entry")


def kernel_hashes(so_path):
    co = code_object(so_path)
    with tempfile.NamedTemporaryFile(suffix=".co") as f:
        f.write(co); f.flush()
        syms = subprocess.check_output([LLVM + "/llvm-readelf", "-sW", f.name]).decode().splitlines()
        secs = subprocess.check_output([LLVM + "/llvm-readelf", "-SW", f.name]).decode().splitlines()
    text = [l.split() for l in secs if " .text " in l][0]
    i = text.index(".text")
    addr, off = int(text[i + 2], 16), int(text[i + 3], 16)
    out = {}
    for l in syms:
        w = l.split()
        if len(w) >= 8 and w[3] == "FUNC" and w[6] != "UND":
            a, size, name = int(w[1], 16), int(w[2]), w[7]
            out[name] = (size, hashlib.sha256(co[off + a - addr:off + a - addr + size]).hexdigest()[:16])
    return out


def disassembly(so_path, names):
    co = code_object(so_path)
    out = {}
    with tempfile.NamedTemporaryFile(suffix=".co") as f:
        f.write(co); f.flush()
        for n in names:
            txt = subprocess.check_output([LLVM + "/llvm-objdump", "-d", "--no-show-raw-insn", "--no-leading-addr", "--disassemble-symbols=" + n, f.name]).decode()
            out[n] = [l.split("//")[0].strip() for l in txt.splitlines() if l.startswith("\t")]
    return out


def compare(a_path, b_path, sub=""):
    """Per function: identical bytes / identical instructions except pc-relative literals (s_add_u32 / s_addc_u32 after s_getpc_b64: the distance to other
    functions and to constant data moves when ANOTHER function grows) / different."""
    a, b = kernel_hashes(a_path), kernel_hashes(b_path)
    names = [n for n in sorted(set(a) | set(b)) if sub in n]
    cand = [n for n in names if n in a and n in b and a[n] != b[n] and a[n][0] == b[n][0]]
    da, db = disassembly(a_path, cand), disassembly(b_path, cand)
    rows = []
    for n in names:
        if n not in a or n not in b:
            rows.append((n, "only in one build"))
        elif a[n] == b[n]:
            rows.append((n, "identical bytes (%d)" % a[n][0]))
        elif a[n][0] != b[n][0]:
            rows.append((n, "DIFFERENT: %d -> %d bytes" % (a[n][0], b[n][0])))
        else:
            diff = [(x, y) for x, y in zip(da[n], db[n]) if x != y]
            rel = all(x.split(",")[:2] == y.split(",")[:2] and x.split()[0] in ("s_add_u32", "s_addc_u32") for x, y in diff) and len(da[n]) == len(db[n])
            rows.append((n, ("identical instructions, %d pc-relative literals moved" % len(diff)) if rel else "DIFFERENT: %d instructions" % len(diff)))
    return rows


def static_mix(so_path, sub=""):
    """Per function: instruction counts by class from the disassembly (static, not executed counts): VALU, SALU, LDS (ds_*), global / buffer memory, scratch
    (spill) loads and stores, waits, branches, calls (s_swappc)."""
    names = [n for n in sorted(kernel_hashes(so_path)) if sub in n]
    rows = []
    for n, lines in disassembly(so_path, names).items():
        c = dict(valu=0, salu=0, lds=0, vmem=0, scratch_ld=0, scratch_st=0, wait=0, branch=0, call=0)
        for l in lines:
            op = l.split()[0] if l.split() else ""
            if op.startswith("scratch_load"): c["scratch_ld"] += 1
            elif op.startswith("scratch_store"): c["scratch_st"] += 1
            elif op.startswith("ds_"): c["lds"] += 1
            elif op.startswith(("global_", "buffer_", "flat_")): c["vmem"] += 1
            elif op.startswith("s_waitcnt"): c["wait"] += 1
            elif op.startswith("s_swappc"): c["call"] += 1
            elif op.startswith(("s_cbranch", "s_branch")): c["branch"] += 1
            elif op.startswith("v_"): c["valu"] += 1
            elif op.startswith("s_"): c["salu"] += 1
        rows.append((n, len(lines), c))
    return rows


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--mix":
        so = os.path.join(ROOT, "robogym_amd", "csrc", "librgstep.so")
        print("%-58s %7s %6s %6s %5s %5s %6s %6s %5s %6s %4s" % ("function", "instr", "valu", "salu", "lds", "vmem", "spl_ld", "spl_st", "wait", "branch", "call"))
        for n, total, c in static_mix(so, sys.argv[2] if len(sys.argv) > 2 else ""):
            print("%-58s %7d %6d %6d %5d %5d %6d %6d %5d %6d %4d" % (n[:58], total, c["valu"], c["salu"], c["lds"], c["vmem"], c["scratch_ld"], c["scratch_st"], c["wait"], c["branch"], c["call"]))
        sys.exit(0)
    if len(sys.argv) > 3 and sys.argv[1] == "--compare":
        for n, verdict in compare(sys.argv[2], sys.argv[3], sys.argv[4] if len(sys.argv) > 4 else ""):
            print("%-60s %s" % (verdict, n))
        sys.exit(0)
    so = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "robogym_amd", "csrc", "librgstep.so")
    sub = sys.argv[2] if len(sys.argv) > 2 else ""
    for name, (size, h) in sorted(kernel_hashes(so).items()):
        if sub in name:
            print("%-16s %8d  %s" % (h, size, name))
