#!/usr/bin/env python3
"""Runs pytest with another build of libsfd2hip.so bound in this process (kernel A/B builds from build_lib(out=...)):
    python tools/run_tests_with_lib.py build/variants/libX.so -m gpu -k nms -q"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from sfd2_amd import _lib  # noqa: E402

_lib.use_library(sys.argv[1])
import pytest  # noqa: E402

sys.exit(pytest.main(["tests"] + sys.argv[2:]))
