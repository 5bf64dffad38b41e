#!/usr/bin/env python3
"""Build oracle/_ref/libjdet_ref.so from the reference's OWN embedded CPU sources.

TEST INFRASTRUCTURE ONLY.  Runs only where /root/reference exists (the build
container); the resulting .so travels to the GPU box with the snapshot
(oracle/_ref/ is git-ignored but not gpurun-ignored).

What it does
------------
JDet keeps its native kernels as C++ source *strings* inside Python modules and
hands them to Jittor's ``jt.code`` JIT (e.g. python/jdet/ops/box_iou_rotated.py:507).
Jittor is not installable here, so this script

 1. parses the reference modules with ``ast`` (no import, no exec of reference
    code) and evaluates only the top-level string constants / concatenations;
 2. performs the two textual steps Jittor's JIT would perform -- drop
    ``#include <executor.h>`` (Jittor runtime header, nothing from it is used by
    the CPU bodies) and expand ``@alias(name,inK)`` into ``name_p`` /
    ``name_shapeD`` variables -- and wraps each body in an ``extern "C"``
    function; for NMS it prepends ``#define BOX_LENGTH`` and
    ``const float iou_threshold`` exactly as ops/nms_rotated.py:498-503 does;
 3. compiles with ``g++ -O2 -std=c++14`` (the flags a default Jittor CPU JIT
    amounts to for these sources: no -march, no fast-math).

The generated translation unit is written to a temporary directory and deleted;
only the .so lands in oracle/_ref/.  No reference source is copied into the repo.

Entry points of the .so (all arrays caller-allocated):
  ref_box_iou_rotated   (b1,n1,b2,n2,out)            ops/box_iou_rotated.py:312-326,487-500
  ref_box_iou_rotated_v1(b1,n1,b2,n2,out)            ops/box_iou_rotated_v1.py:317-331,492-505
  ref_nms_rotated5/6    (dets,n,order,thr,keep)      ops/nms_rotated.py:314-328,414-449
  ref_arf_forward/backward                            ops/orn.py:132-257 (uint16 index quirk included)
  ref_rie_forward/backward                            ops/orn.py:283-395 (forward: the embedded CPU_SRC body; backward: the
                                                      header's RIE_backward_cpu_kernel called directly, because the embedded
                                                      RIE_CPU_GRAD_SRC body misses a ';' at :383 and does not compile as written)
  ref_convex_sort       (x,y,m,start,order,nbs,npts,circular,out)  ops/convex_sort.py:93-154 (the Graham-scan loop of
                                                      convex_sort_cpu; the tensor code before it -- argmin / argsort --
                                                      is Jittor and is restated in NumPy by the caller)
"""
import ast
import os
import re
import shutil
import subprocess
import sys
import tempfile

REF_OPS = "/root/reference/python/jdet/ops"
HERE = os.path.dirname(os.path.abspath(__file__))
OUT_DIR = os.path.join(HERE, "_ref")


def module_strings(path):
    """Evaluate top-level NAME = <str | NAME | a + b> assignments of a module."""
    tree = ast.parse(open(path).read())
    env = {}

    def ev(node):
        if isinstance(node, ast.Constant) and isinstance(node.value, str):
            return node.value
        if isinstance(node, ast.Name) and node.id in env:
            return env[node.id]
        if isinstance(node, ast.BinOp) and isinstance(node.op, ast.Add):
            return ev(node.left) + ev(node.right)
        raise ValueError("not a plain string expression")

    for st in tree.body:
        if isinstance(st, ast.Assign) and len(st.targets) == 1 and isinstance(st.targets[0], ast.Name):
            try:
                env[st.targets[0].id] = ev(st.value)
            except ValueError:
                pass
    return env


def strip_jittor(src):
    return re.sub(r"#include\s*<\s*executor\.h\s*>", "", src)


def expand_alias(body, ndims):
    """@alias(name,inK) -> name_p / name_shapeD locals (what Jittor's JIT emits)."""
    def repl(m):
        name, tgt = m.group(1).strip(), m.group(2).strip()
        out = ["auto %s_p = %s_p;" % (name, tgt)]
        for d in range(ndims.get(tgt, 0)):
            out.append("auto %s_shape%d = %s_shape%d; (void)%s_shape%d;" % (name, d, tgt, d, name, d))
        return " ".join(out)
    return re.sub(r"@alias\(([^,]+),([^)]+)\)", repl, body)


def unit_iou(strings, ns, fn):
    hdr = strip_jittor(strings["IOU_ROTATED_CPU_HEADER"])
    body = expand_alias(strings["IOU_CPU_SRC"], {"in0": 2, "in1": 2})
    return """
namespace %s {
%s
}
extern "C" void %s(float* in0_p, int in0_shape0, float* in1_p, int in1_shape0, float* out0_p) {
  using namespace %s;
  int in0_shape1 = 5, in1_shape1 = 5; (void)in1_shape1;
  %s
}
""" % (ns, hdr, fn, ns, body)


def unit_nms(strings, box_len):
    ns = "nms%d" % box_len
    hdr = "#define BOX_LENGTH %d\n" % box_len + strip_jittor(strings["ML_NMS_ROTATED_CPU_HEADER"])
    body = strings["ML_NMS_ROTATED_CPU_SRC"].replace("keep_t->size", "(ndets*sizeof(bool))")
    body = expand_alias(body, {"in0": 1, "in1": 0, "in2": 0, "out0": 0})
    return """
namespace %s {
%s
}
#undef BOX_LENGTH
#define BOX_LENGTH %d
extern "C" void ref_nms_rotated%d(float* in0_p, int in0_shape0, int* in1_p, float thr, bool* out0_p) {
  using namespace %s;
  unsigned char* in2_p = new unsigned char[in0_shape0 > 0 ? in0_shape0 : 1]();
  const float iou_threshold = thr;
  %s
  delete[] in2_p;
}
#undef BOX_LENGTH
""" % (ns, hdr, box_len, box_len, ns, body)


def unit_arf(strings):
    hdr = strings["ARF_CPU_HEADER"]
    fwd = expand_alias(strings["ARF_CPU_SRC"], {"in0": 5, "in1": 4})
    bwd = expand_alias(strings["ARF_CPU_GRAD_SRC"], {"in0": 4, "in1": 4})
    return """
namespace arf {
%s
}
extern "C" void ref_arf_forward(float* in0_p, int in0_shape0, int in0_shape1, int in0_shape2,
                                int in0_shape3, int in0_shape4, unsigned char* in1_p,
                                int in1_shape3, float* out0_p) {
  using namespace arf;
  int in1_shape0 = in0_shape2, in1_shape1 = in0_shape3, in1_shape2 = in0_shape4;
  %s
}
extern "C" void ref_arf_backward(unsigned char* in0_p, int in0_shape0, int in0_shape1,
                                 int in0_shape2, int in0_shape3, float* in1_p, int in1_shape0,
                                 int in1_shape1, float* out0_p) {
  using namespace arf;
  int in1_shape2 = in0_shape1, in1_shape3 = in0_shape2;
  %s
}
""" % (hdr, fwd, bwd)


def unit_rie(strings):
    hdr = strings["RIE_CPU_HEADER"]
    fwd = expand_alias(strings["RIE_CPU_SRC"].replace("aligned->size", "((size_t)in0_shape0*in0_shape1*sizeof(float))"),
                       {"in0": 2})
    return """
namespace rie {
%s
}
extern "C" void ref_rie_forward(float* in0_p, int in0_shape0, int in0_shape1, int nOri, unsigned char* out0_p,
                                float* out1_p) {
  using namespace rie;
  const uint8 nOrientation = (uint8)nOri;
  %s
}
extern "C" void ref_rie_backward(unsigned char* dir_p, int nBatch, int nFeature, int nOri, float* grad_out_p,
                                 float* grad_in_p) {
  using namespace rie;
  std::memset(grad_in_p, 0, (size_t)nBatch * nFeature * nOri * sizeof(float));
  RIE_backward_cpu_kernel<float>(dir_p, grad_out_p, (uint8)nOri, (uint16)nBatch, (uint16)nFeature, grad_in_p);
}
""" % (hdr, fwd)


def function_raw_string(path, func, var):
    """The plain-string operand(s) of `var = f"..." + r"..."` inside function `func` (no evaluation of the f-string:
    its only content is the four scalar declarations, which the wrapper below supplies as parameters)."""
    tree = ast.parse(open(path).read())
    for st in tree.body:
        if isinstance(st, ast.FunctionDef) and st.name == func:
            for sub in ast.walk(st):
                if isinstance(sub, ast.Assign) and getattr(sub.targets[0], "id", None) == var:
                    parts = []

                    def walk(node):
                        if isinstance(node, ast.BinOp):
                            walk(node.left)
                            walk(node.right)
                        elif isinstance(node, ast.Constant) and isinstance(node.value, str):
                            parts.append(node.value)
                    walk(sub.value)
                    return "".join(parts)
    raise KeyError((func, var))


def unit_convex(path):
    body = function_raw_string(path, "convex_sort_cpu", "SRC")
    return """
extern "C" void ref_convex_sort(float* in0_p, float* in1_p, float* in2_p, int* in3_p, int* in4_p, int nbs, int npts,
                                int circular_i, int* out0_p) {
  const int index_size = circular_i ? npts + 1 : npts;
  const bool circular = circular_i != 0;
  %s
}
""" % body


def main():
    if not os.path.isdir(REF_OPS):
        print("build_ref: %s not present -- using prebuilt oracle/_ref if any" % REF_OPS)
        return 0
    s0 = module_strings(os.path.join(REF_OPS, "box_iou_rotated.py"))
    s1 = module_strings(os.path.join(REF_OPS, "box_iou_rotated_v1.py"))
    sn = module_strings(os.path.join(REF_OPS, "nms_rotated.py"))
    so = module_strings(os.path.join(REF_OPS, "orn.py"))
    tu = ("#include <cstring>\n#include <cstdint>\n#include <cassert>\n#include <cmath>\n"
          "#include <algorithm>\n")  # system headers first: the reference headers re-include them inside our namespaces
    tu += unit_iou(s0, "iou0", "ref_box_iou_rotated")
    tu += unit_iou(s1, "iou1", "ref_box_iou_rotated_v1")
    tu += unit_nms(sn, 5)
    tu += unit_nms(sn, 6)
    tu += unit_arf(so)
    tu += unit_rie(so)
    tu += unit_convex(os.path.join(REF_OPS, "convex_sort.py"))
    os.makedirs(OUT_DIR, exist_ok=True)
    tmp = tempfile.mkdtemp(prefix="jdet_ref_")
    try:
        cpp = os.path.join(tmp, "jdet_ref_tu.cpp")
        with open(cpp, "w") as f:
            f.write(tu)
        out = os.path.join(OUT_DIR, "libjdet_ref.so")
        cmd = ["g++", "-O2", "-std=c++14", "-fPIC", "-shared", "-w", "-o", out, cpp]
        subprocess.check_call(cmd)
        print("build_ref: built", out)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())
