"""The OpenCV pinning hook (scripts/opencv_ref/): OpenCV is absent from this image, so the dump tool is never built here -- this test keeps the
hook USABLE: the exporter writes every stage's inputs, the C++ tool names every stage the exporter writes, and the comparer reads, indexes
and bounds a complete set of outputs (filled in by the oracle itself, standing in for the tool: that proves the plumbing, not parity)."""
import importlib.util
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOOK = os.path.join(ROOT, "scripts", "opencv_ref")


def _mod():
    spec = importlib.util.spec_from_file_location("opencv_ref_compare", os.path.join(HOOK, "opencv_ref_compare.py"))
    m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
    return m


def test_export_dump_compare_plumbing(tmp_path, capsys):
    m = _mod()
    d = str(tmp_path)
    m.export(d)
    stems = {re.sub(r"_\d+\.", "_N.", f) for f in os.listdir(d)}
    cpp = open(os.path.join(HOOK, "opencv_ref_dump.cpp")).read()
    for stem in stems:                                                     # every input the exporter writes is read by the tool
        name, ext = stem.split(".", 1)
        assert ('"' + name.replace("_N", "_") in cpp or '"' + name + "." in cpp) and "." + ext + '"' in cpp, stem
    for call in ("cv::resize", "INTER_AREA", "FastFeatureDetector::create", "SparsePyrLKOpticalFlow::create", "cv::findHomography", "cv::UsacParams",
                 "estimateAffinePartial2D", "getGaussianKernel", "getPerspectiveTransform", "INTER_LINEAR_EXACT", "cv::ocl::setUseOpenCL(false)",
                 "LVK_WITH_EIGEN", "Eigen::LeastSquaresConjugateGradient", "solveWithGuess", "setFromTriplets"):
        assert call in cpp, call
    m.selftest_outputs(d)
    assert m.compare(d) == 0
    out = capsys.readouterr().out
    for stage in ("a3 cvtColor", "a4 INTER_AREA", "a5 FAST", "a7 PyrLK", "a9 findHomography", "a12 getGaussianKernel", "a14 getPerspectiveTransform",
                  "a14 mesh -> map", "f2 chroma INTER_LINEAR", "f2 chroma INTER_AREA", "a10 Eigen LSCG"):
        assert "ok   " + stage in out, stage
    assert "FAIL" not in out
    # one command for whoever has the libraries: export, build against pkg-config's opencv4 (+ eigen3), dump, compare
    run = open(os.path.join(HOOK, "run.sh")).read()
    for step in ("opencv_ref_compare.py\" export", "pkg-config --cflags --libs opencv4", "-DLVK_WITH_EIGEN", "opencv_ref_dump\" \"$WORK\"", "opencv_ref_compare.py\" compare"):
        assert step in run, step
