"""CPU: the C++ drop-in for the extractor type-checks against the reference's own, unmodified include/ORBextractor.h.

OpenCV is not installed here, so the cv:: types come from a declaration-only test double
(tests/support/cv_typecheck_stub/, see its header).  Only this repository's adapter is compiled, with -fsyntax-only;
no reference source file is built.  Skipped where /root/reference is absent (e.g. on the GPU box)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_INC = "/root/reference/include"


@pytest.mark.skipif(not os.path.exists(os.path.join(REF_INC, "ORBextractor.h")) or shutil.which("g++") is None,
                    reason="reference headers or g++ not available")
def test_extractor_adapter_typechecks_against_reference_header():
    cmd = ["g++", "-std=c++11", "-fsyntax-only", "-Wall",
           "-I", os.path.join(ROOT, "tests", "support", "cv_typecheck_stub"), "-I", REF_INC, "-I", os.path.join(ROOT, "include"),
           os.path.join(ROOT, "3_orb_slam3_selfnote_amd", "csrc", "adapter", "ORBextractor_hip.cc")]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
