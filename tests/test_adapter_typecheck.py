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


@pytest.mark.skipif(not os.path.exists(os.path.join(REF_INC, "ORBmatcher.h")) or shutil.which("g++") is None,
                    reason="reference headers or g++ not available")
def test_matcher_adapter_typechecks_against_reference_headers():
    """ORBmatcher_hip.cc - the whole-TU replacement of src/ORBmatcher.cc - against the reference's unmodified ORBmatcher.h, Frame.h,
    KeyFrame.h, MapPoint.h (and the in-tree DBoW2 headers they include).  OpenCV, Eigen, boost and Pangolin are absent from this image:
    their headers are declaration-only doubles (tests/support/slam_typecheck_stub/).  -fsyntax-only, only this repository's file."""
    stub = os.path.join(ROOT, "tests", "support", "slam_typecheck_stub")
    cmd = ["g++", "-std=c++11", "-fsyntax-only", "-I", stub, "-I", REF_INC, "-I", os.path.join(REF_INC, "CameraModels"), "-I", "/root/reference",
           "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "3_orb_slam3_selfnote_amd", "csrc", "adapter", "ORBmatcher_hip.cc")]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-4000:]


@pytest.mark.skipif(not os.path.exists(os.path.join(REF_INC, "ORBmatcher.h")) or shutil.which("g++") is None,
                    reason="reference headers or g++ not available")
def test_matcher_adapter_defines_every_declared_member():
    """Every member function ORBmatcher.h declares has a definition in the adapter (a missing one would only show at link time)."""
    import re
    hdr = open(os.path.join(REF_INC, "ORBmatcher.h"), encoding="utf-8", errors="ignore").read()
    body = hdr[hdr.index("class ORBmatcher"):]
    names = set(re.findall(r"\b(\w+)\s*\(", " ".join(l.split("//")[0] for l in body.splitlines())))
    names -= {"ORBmatcher"}
    names = {n for n in names if n[0].isupper()}
    src = open(os.path.join(ROOT, "3_orb_slam3_selfnote_amd", "csrc", "adapter", "ORBmatcher_hip.cc")).read()
    src = re.sub(r'"(?:[^"\\\\]|\\\\.)*"', '""', "\n".join(l.split("//")[0] for l in src.splitlines()))   # no comments, no string literals
    decl_counts = {n: len(re.findall(r"\b%s\s*\(" % n, " ".join(l.split("//")[0] for l in body.splitlines()))) for n in names}
    for n, cnt in decl_counts.items():
        defs = len(re.findall(r"ORBmatcher::%s\s*\(" % n, src))
        assert defs == cnt, "%s: %d declarations, %d definitions" % (n, cnt, defs)
    assert "ORBmatcher::ORBmatcher(" in src and "ORBmatcher::TH_LOW" in src and "ORBmatcher::TH_HIGH" in src and "ORBmatcher::HISTO_LENGTH" in src


@pytest.mark.skipif(not os.path.exists(os.path.join(REF_INC, "Frame.h")) or shutil.which("g++") is None,
                    reason="reference headers or g++ not available")
@pytest.mark.parametrize("name", ["Frame_ComputeStereoMatches_hip.cc", "MapPoint_ComputeDistinctiveDescriptors_hip.cc"])
def test_member_snippets_typecheck(name):
    """The drop-in bodies of Frame::ComputeStereoMatches (Frame.cc:901-1079) and MapPoint::ComputeDistinctiveDescriptors
    (MapPoint.cc:350-436) against the reference's unmodified Frame.h / MapPoint.h / KeyFrame.h."""
    stub = os.path.join(ROOT, "tests", "support", "slam_typecheck_stub")
    cmd = ["g++", "-std=c++11", "-fsyntax-only", "-I", stub, "-I", REF_INC, "-I", os.path.join(REF_INC, "CameraModels"), "-I", "/root/reference",
           "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "3_orb_slam3_selfnote_amd", "csrc", "adapter", "snippets", name)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-4000:]
