#!/usr/bin/env python3
"""`make install`: make `import bindings` / `import million_amd` resolve without PYTHONPATH.

The reference installs its extension with `python setup.py install` (reference makefile:1-4).  This repository's `bindings`
is a pure-Python shim over the in-tree C-ABI library (million_amd/libmillion_hip.so), so installing means telling the
interpreter where the checkout is - a one-line `.pth` file in site-packages, what `pip install -e .` does, with no network
and no copy of the library (the driver wants the `.so` loaded from the tree).

    python tools/install_pth.py              # user site-packages of the running interpreter
    python tools/install_pth.py --target D   # any directory on sys.path / passed to site.addsitedir
    python tools/install_pth.py --uninstall
"""
import argparse
import site
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
NAME = "million_hip.pth"


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--target", default=None, help="directory to write the .pth file into (default: user site-packages)")
    ap.add_argument("--uninstall", action="store_true")
    a = ap.parse_args()
    target = Path(a.target) if a.target else Path(site.getusersitepackages())
    pth = target / NAME
    if a.uninstall:
        if pth.exists():
            pth.unlink()
            print(f"removed {pth}")
        return 0
    if not (ROOT / "million_amd" / "libmillion_hip.so").exists():
        print("note: million_amd/libmillion_hip.so is not built yet - run `make bindings`", file=sys.stderr)
    target.mkdir(parents=True, exist_ok=True)
    pth.write_text(str(ROOT) + "\n")
    print(f"wrote {pth} -> {ROOT}")
    return 0


if __name__ == "__main__":
    sys.exit(main())
