"""CPU (not gpu): the kernels' single-pass left-feature mask (featMaskFast, kiwi_amd/csrc/feature.hpp) equals the thirteen predicate calls it
replaces (featMask: FeatureTestor::isMatched for every CondVowel / CondPolarity, /root/reference/src/FeatureTestor.cpp:6-78) on every single unit
of the Hangul blocks and on two million random strings."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))


def test_single_pass_feature_mask_equals_the_predicates(tmp_path):
    exe = str(tmp_path / "feature_mask_check")
    subprocess.check_call(["g++", "-O2", "-std=c++17", os.path.join(HERE, "cxx", "feature_mask_check.cpp"), "-o", exe])
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0 and " bad 0" in r.stdout, r.stdout + r.stderr
