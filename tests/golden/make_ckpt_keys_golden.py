"""Attribute names of the reference's model classes, read with `ast` from where the reference lies (no import, no text
copied): for every class of the files below, the names it assigns on ``self`` (``self.x = ...``, ``self.add_module('x',
...)``, ``setattr(self, f"x{i}", ...)`` -> the pattern ``x#``) in any of its methods.  These names are what a JDet
checkpoint's keys are made of (jt.save of model.state_dict(), runner.py:251-262); tests/test_checkpoint_cpu.py checks
that every child module / parameter name of the product's classes of the same name is one of them.
Run in the build container:  python tests/golden/make_ckpt_keys_golden.py  ->  tests/golden/ckpt_attrs.json"""
import ast
import json
import os
import sys

REF = "/root/reference/python/jdet"
FILES = ["models/backbones/resnet.py", "models/backbones/van.py", "models/necks/fpn.py", "models/roi_heads/s2anet_head.py",
         "models/roi_heads/oriented_rpn_head.py", "models/roi_heads/oriented_head.py", "models/utils/modules.py",
         "ops/dcn_v1.py", "ops/orn.py", "models/roi_extractors/oriented_single_level.py", "models/networks/rcnn.py",
         "models/networks/s2anet.py", "models/roi_heads/retina_head.py", "models/networks/retinanet.py"]


def _fstring_pattern(node):
    out = ""
    for v in node.values:
        out += v.value if isinstance(v, ast.Constant) else "#"
    return out


def class_attrs(cls):
    names = set()
    for node in ast.walk(cls):
        if isinstance(node, (ast.Assign, ast.AnnAssign, ast.AugAssign)):
            targets = node.targets if isinstance(node, ast.Assign) else [node.target]
            for t in targets:
                for el in (t.elts if isinstance(t, (ast.Tuple, ast.List)) else [t]):
                    if isinstance(el, ast.Attribute) and isinstance(el.value, ast.Name) and el.value.id == "self":
                        names.add(el.attr)
        elif isinstance(node, ast.Call):
            f = node.func
            if isinstance(f, ast.Attribute) and f.attr == "add_module" and node.args and isinstance(node.args[0], ast.Constant):
                names.add(str(node.args[0].value))
            if isinstance(f, ast.Name) and f.id == "setattr" and len(node.args) >= 2 and \
                    isinstance(node.args[0], ast.Name) and node.args[0].id == "self":
                a = node.args[1]
                if isinstance(a, ast.Constant):
                    names.add(str(a.value))
                elif isinstance(a, ast.JoinedStr):
                    names.add(_fstring_pattern(a))
    return sorted(names)


def main():
    out = {}
    for rel in FILES:
        with open(os.path.join(REF, rel)) as f:
            tree = ast.parse(f.read())
        for node in tree.body:
            if isinstance(node, ast.ClassDef):
                out.setdefault(node.name, {"file": rel, "attrs": []})
                out[node.name]["attrs"] = sorted(set(out[node.name]["attrs"]) | set(class_attrs(node)))
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ckpt_attrs.json")
    if "--check" in sys.argv:
        with open(path) as f:
            assert json.load(f) == out, "tests/golden/ckpt_attrs.json is stale: re-run this script"
        return
    with open(path, "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print("wrote", path, {k: len(v["attrs"]) for k, v in out.items()})


if __name__ == "__main__":
    main()
