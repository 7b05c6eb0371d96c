"""TEST INFRASTRUCTURE - recipe for `oracle/_ref/`: the *unmodified reference*, compiled where it lies.

The reference (MaxHalford/sorobn, /root/reference) is pure Python: "building" it means byte-compiling its own
source files - straight from /root/reference/sorobn/*.py, nothing edited, nothing copied into the repo - into
sourceless bytecode under the git-ignored `oracle/_ref/sorobn/` (`<module>.pyc` next to each other, which CPython
imports without sources).  Like the in-tree `.so` files the directory stays out of the history (.gitignore) but is not
gpurun-ignored, so it travels to the GPU box, where /root/reference does not exist, and `bench.py`'s `cpu_baseline`
leg (kind "reference") and the `-m gpu` drop-in tests can run the reference's own pandas path on that box's host
cores (same image, same CPython 3.10 bytecode magic; `refload.load()` checks the magic and refuses otherwise).

    python oracle/build_ref.py        # or: make -C oracle _ref      (a no-op with a note when /root/reference is absent)

Only the four modules `import sorobn` needs are compiled (`__init__`, `bayes_net`, `examples`, `structure`); the
Streamlit GUI and the reference's own test module are not.  `vose` (third-party, absent) is stubbed by refload.py.
"""
import importlib.util
import json
import os
import py_compile
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REFERENCE_ROOT = "/root/reference"
MODULES = ("__init__", "bayes_net", "examples", "structure")


def build(out_root=None, quiet=False):
    src_pkg = os.path.join(REFERENCE_ROOT, "sorobn")
    out_root = out_root or os.path.join(HERE, "_ref")
    if not os.path.isdir(src_pkg):
        if not quiet:
            state = "keeping the prebuilt one" if os.path.isdir(os.path.join(out_root, "sorobn")) else "none present"
            print(f"[oracle/_ref] {REFERENCE_ROOT} is not mounted here: nothing to compile ({state})")
        return False
    out_pkg = os.path.join(out_root, "sorobn")
    os.makedirs(out_pkg, exist_ok=True)
    manifest = {"python": sys.version.split()[0], "magic": importlib.util.MAGIC_NUMBER.hex(), "modules": {}}
    for m in MODULES:
        src = os.path.join(src_pkg, m + ".py")
        dst = os.path.join(out_pkg, m + ".pyc")
        # UNCHECKED_HASH: the .pyc is valid on its own (no source mtime to compare with on the GPU box)
        py_compile.compile(src, cfile=dst, dfile=f"<reference>/sorobn/{m}.py", doraise=True,
                           invalidation_mode=py_compile.PycInvalidationMode.UNCHECKED_HASH)
        manifest["modules"][m] = {"source": src, "source_bytes": os.path.getsize(src)}
    with open(os.path.join(out_root, "MANIFEST.json"), "w") as f:
        json.dump(manifest, f, indent=1)
    if not quiet:
        print(f"[oracle/_ref] byte-compiled {len(MODULES)} reference modules from {src_pkg} -> {out_pkg}")
    return True


if __name__ == "__main__":
    build()
