"""Helper process of `rg_compile_mjcf` / `rb_compile_mjcf` (include/rgstep.h): the MJCF document on stdin -> the RGMODEL1 blob in a file.

This is what `mujoco_py.load_model_from_xml(xml_string)` is to the reference (/root/reference/robogym/mujoco/mujoco_xml.py:249-260: `MujocoXML.build`
hands the merged document's XML STRING to MuJoCo's compiler): the C-ABI library runs this module with the caller's string, reads the blob back and creates the
model from it, so a non-Python host crosses the boundary with an XML string exactly as the reference does.  The compiler itself is the package's
(`mjcf_compiler.compile_mjcf` + `setconst.set_constants` + the kernel's derived tables).

    python -m robogym_amd.mujoco.compile_cli --kind rg|rb --out model.blob [--xml model.xml] [--meshdir DIR] [--names names.json] < model.xml

Exit status 0 on success; on failure the message goes to stderr (the library hands its tail to the caller's `err` buffer) and the status is 1."""
import argparse
import json
import sys


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--kind", choices=["rg", "rb"], required=True, help="rg: the Shadow-hand layout (rg_model_create); rb: the large-model stepper (rb_model_create)")
    ap.add_argument("--out", required=True)
    ap.add_argument("--xml", default=None, help="file holding the MJCF document (default: stdin)")
    ap.add_argument("--meshdir", default=None, help="directory mesh file names are relative to (default: the document's <compiler meshdir>)")
    ap.add_argument("--names", default=None, help="also write the name tables (body / joint / geom / ... -> list of names) as JSON")
    args = ap.parse_args(argv)
    try:
        from robogym_amd.mujoco.model_blob import pack_model
        from robogym_amd.mujoco.mujoco_xml import MujocoXML

        xml = open(args.xml).read() if args.xml else sys.stdin.read()
        if not xml.strip():
            raise ValueError("empty MJCF document on stdin")
        model = MujocoXML.from_string(xml).build(meshdir=args.meshdir or None)
        if args.kind == "rg":
            from robogym_amd.mujoco.kernel_tables import derive_kernel_tables
            derive_kernel_tables(model)
        else:
            from robogym_amd.mujoco.big_tables import derive_big_tables
            derive_big_tables(model)
        blob = pack_model(model)
        with open(args.out, "wb") as f:
            f.write(blob)
        if args.names:
            with open(args.names, "w") as f:
                json.dump(model.names, f)
    except Exception as ex:      # the caller is a C program: one line it can show
        sys.stderr.write("compile_cli: %s: %s\n" % (type(ex).__name__, ex))
        return 1
    return 0


if __name__ == "__main__":
    sys.exit(main())
