#!/usr/bin/env python3
"""`python render_test.py --config <reference config>`: what `python test.py --config ...` is in the reference."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _pkg  # noqa: E402

_pkg.load()
from ibl_nerf_amd import render_views  # noqa: E402

if __name__ == "__main__":
    render_views.main()
