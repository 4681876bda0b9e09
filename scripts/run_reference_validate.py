#!/usr/bin/env python3
"""Run the reference's fastervit/validate.py UNMODIFIED against the MI355X implementation.

validate.py does `from models.faster_vit import *` / `from models.gcvit import *` with cwd = fastervit/
(validate.py:24-25; models/gcvit.py does not exist upstream) and needs timm.  This launcher puts on sys.path
  1. a generated `models/` package that re-exports fastervit_amd's entrypoints (+ an empty gcvit stub),
  2. timm: the real one when installed, else the test-only shim (tests/golden/_shim) with a synthetic loader,
then executes the reference file with runpy.  Example (GPU box with the reference checked out):

  python scripts/run_reference_validate.py /path/to/FasterViT/fastervit/validate.py \
      --model faster_vit_0_224 --checkpoint ck.pth.tar -b 256 --amp --channels-last --device cuda
"""
import os
import runpy
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def make_models_package(dst):
    pkg = os.path.join(dst, "models")
    os.makedirs(pkg, exist_ok=True)
    open(os.path.join(pkg, "__init__.py"), "w").close()
    with open(os.path.join(pkg, "faster_vit.py"), "w") as f:
        f.write("from fastervit_amd.models.faster_vit import *  # noqa\nfrom fastervit_amd.models.faster_vit_any_res import *  # noqa\n")
    with open(os.path.join(pkg, "gcvit.py"), "w") as f:
        f.write("# stub: the reference imports models.gcvit (validate.py:25) but does not ship it\n")
    return dst


def main(argv):
    if not argv or not os.path.isfile(argv[0]):
        sys.exit(__doc__)
    script, rest = argv[0], argv[1:]
    tmp = tempfile.mkdtemp(prefix="fvit_validate_")
    paths = [make_models_package(tmp), ROOT]
    try:
        import timm  # noqa: F401
    except ImportError:
        paths.insert(0, os.path.join(ROOT, "tests", "golden", "_shim"))
    sys.path[:0] = paths
    import fastervit_amd  # noqa: F401
    from fastervit_amd.models.faster_vit import register_with_timm
    register_with_timm()  # make the entrypoints visible to timm.models.create_model (real timm or the shim)
    sys.argv = [script] + rest
    runpy.run_path(script, run_name="__main__")


if __name__ == "__main__":
    main(sys.argv[1:])
