#!/usr/bin/env python3
"""`make install`: make `import bindings` / `import million_amd` resolve without PYTHONPATH.

The reference installs its extension with `python setup.py install` (reference makefile:1-4).  This repository's `bindings`
is a pure-Python shim over the in-tree C-ABI library (million_amd/libmillion_hip.so), so installing means telling the
interpreter where the two packages are - a one-line `.pth` file in site-packages, what `pip install -e .` does, with no network
and no copy of the library (the driver wants the `.so` loaded from the tree).  The `.pth` does NOT name the checkout root (that
would make the generically named `tests`, `tools` and `oracle` directories importable everywhere, where they could shadow other
projects' modules): it names `build/site/`, a directory holding two symlinks, `bindings` and `million_amd`.

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
PACKAGES = ("bindings", "million_amd")      # nothing else of the checkout becomes importable


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
    site_dir = ROOT / "build" / "site"
    site_dir.mkdir(parents=True, exist_ok=True)
    for pkg in PACKAGES:
        link = site_dir / pkg
        if link.is_symlink() or link.exists():
            link.unlink()
        link.symlink_to(ROOT / pkg, target_is_directory=True)
    target.mkdir(parents=True, exist_ok=True)
    pth.write_text(str(site_dir) + "\n")
    print(f"wrote {pth} -> {site_dir} ({', '.join(PACKAGES)})")
    return 0


if __name__ == "__main__":
    sys.exit(main())
