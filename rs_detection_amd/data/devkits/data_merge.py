"""Test-time submission pipeline (/root/reference/python/jdet/data/devkits/data_merge.py:14-104,
dota_to_fair.py:6-36,102-118): predictions of the (possibly flipped) tiles -> per-class Task-1 files
(``before_nms``) -> tile merge + polygon NMS on the GPU (``after_nms``, devkits/result_merge.py) -> the
submission archive (a zip of the class files for DOTA; one csv for FAIR1M-1.5).  The reference shells out to ``zip`` /
``mv``; here the archive is written with ``zipfile`` / ``shutil`` and lands under ``<work_dir>/submit_zips``."""
import os
import shutil
import zipfile

from rs_detection_amd.config.constant import get_classes_by_name
from .result_merge import mergebypoly


def flip_box(box, target):
    """:14-27: undo a test-time flip of a tile on its 8 polygon coordinates."""
    ans = [float(box[i]) for i in range(8)]
    if "flip_mode" not in target:
        return ans
    mode = target["flip_mode"]
    w, h = target['ori_img_size'][0], target['ori_img_size'][1]
    if 'H' in mode:
        for i in (0, 2, 4, 6):
            ans[i] = w - ans[i]
    if 'V' in mode:
        for i in (1, 3, 5, 7):
            ans[i] = h - ans[i]
    return ans


def prepare_data(results, save_path, classes):
    """:29-48: [((polys (n,8), scores, labels), target)] -> ``save_path/<class>.txt`` Task-1 lines."""
    os.makedirs(save_path, exist_ok=True)
    data = {}
    for result, target in results:
        img_name = os.path.splitext(os.path.split(target["img_file"])[-1])[0]
        for bbox, score, label in zip(*result):
            b = flip_box(bbox, target)
            data.setdefault(classes[int(label)], []).append(
                '{} {:.4f} {:.4f} {:.4f} {:.4f} {:.4f} {:.4f} {:.4f} {:.4f} {:.4f}\n'.format(img_name, float(score), *b))
    for classname, lines in data.items():
        with open(os.path.join(save_path, classname + '.txt'), 'w') as f:
            f.writelines(lines)
    return data


def data_merge(results, save_path, final_path, dataset_type, device="cuda", nms_threshold_type=0):
    """:50-54."""
    classes = get_classes_by_name(dataset_type)
    prepare_data(results, save_path, classes)
    os.makedirs(final_path, exist_ok=True)
    mergebypoly(save_path, final_path, device=device, nms_threshold_type=nms_threshold_type)


def pick_res(path, images_dir, keep_underline=False):
    """dota_to_fair.py:6-36: merged class files -> {image: [{cls, p, box}]}, one entry per image of ``images_dir``."""
    res = {}
    for root, _, files in os.walk(images_dir):
        for f in files:
            if f.endswith(".png"):
                res[f.split("__")[0]] = []
    for root, _, files in os.walk(path):
        for f in sorted(files):
            cls = f[:-4] if keep_underline else f[:-4].replace("_", " ")
            with open(os.path.join(root, f)) as ff:
                for line in ff.read().split("\n"):
                    if len(line) < 5:
                        continue
                    d = line.split(" ")
                    assert d[0] in res, "detection for an image that is not in images_dir: %s" % d[0]
                    res[d[0]].append({"cls": cls, "p": float(d[1]), "box": [float(v) for v in d[2:10]]})
    return res


def dota_to_fair1m_1_5(src_path, tar_path, images_dir, name):
    """dota_to_fair.py:102-118: one csv, ``<id>.tif,class,x1..y4,score``."""
    data = pick_res(src_path, images_dir, keep_underline=True)
    os.makedirs(tar_path, exist_ok=True)
    out = os.path.join(tar_path, name + ".csv")
    with open(out, "w") as f:
        for i in data:
            for obj in data[i]:
                f.write('{},{},{:.4f},{:.4f},{:.4f},{:.4f},{:.4f},{:.4f},{:.4f},{:.4f},{:.4f}\n'.format(
                    str(int(i[1:])) + ".tif", obj["cls"], *obj["box"], obj["p"]))
    return out


SUPPORTED_MERGE_TYPES = ("FAIR", "DOTA", "DOTA1", "DOTA1_5", "DOTA2", "FAIR1M_1_5")


def check_dataset_type(dataset_type):
    """Raised BEFORE any inference is run (ImageDataset.__init__, Runner.test): a whole predicted test set must not be
    lost to a type the submission writer does not know (the reference asserts only after the run, data_merge.py:57)."""
    if dataset_type not in SUPPORTED_MERGE_TYPES:
        raise ValueError("dataset.test.dataset_type = %r: set it to one of %s" % (dataset_type, ", ".join(SUPPORTED_MERGE_TYPES)))


def dota_to_fair(src_path, tar_path, images_dir):
    """dota_to_fair.py:37-100: one FAIR1M (v1) submission XML per image, ``<int(id)>.xml``: source / research / size
    blocks with the fixed values of the challenge template (1000 x 1000 x 3, GF2/GF3, version 4.0) and one <object> per
    detection: class, probability, the four corners + the first one again (closed rectangle)."""
    data = pick_res(src_path, images_dir)
    os.makedirs(tar_path, exist_ok=True)
    for i in data:
        img = str(int(i[1:]))
        parts = ['<?xml version="1.0" encoding="utf-8"?>', "<annotation>",
                 "  <source>", "    <filename>%s.tif</filename>" % img, "    <origin>GF2/GF3</origin>", "  </source>",
                 "  <research>", "    <version>4.0</version>", "    <provider>placeholder_affiliation</provider>",
                 "    <author>placeholder_authorname</author>", "    <pluginname>placeholder_direction</pluginname>",
                 "    <pluginclass>placeholder_suject</pluginclass>", "    <time>2020-07-2020-11</time>", "  </research>",
                 "  <size>", "    <width>1000</width>", "    <height>1000</height>", "    <depth>3</depth>", "  </size>",
                 "  <objects>"]
        for obj in data[i]:
            b = obj["box"]
            pts = ["%s, %s" % (b[2 * k], b[2 * k + 1]) for k in (0, 1, 2, 3, 0)]
            parts += ["    <object>", "      <coordinate>pixel</coordinate>", "      <type>rectangle</type>",
                      "      <description>None</description>", "      <possibleresult>",
                      "        <name>%s</name>" % obj["cls"], "        <probability>%s</probability>" % obj["p"],
                      "      </possibleresult>", "      <points>"]
            parts += ["        <point>%s</point>" % q for q in pts]
            parts += ["      </points>", "    </object>"]
        parts += ["  </objects>", "</annotation>", ""]
        with open(os.path.join(tar_path, img + ".xml"), "w") as f:
            f.write("\n".join(parts))


def data_merge_result(results, work_dir, epoch, name, dataset_type, images_dir="", device="cuda",
                      nms_threshold_type=0):
    """:56-104 -> path of the submission file (zip; csv for FAIR1M_1_5; a zip of test/*.xml for FAIR)."""
    check_dataset_type(dataset_type)
    save_path = os.path.join(work_dir, "test", "submit_%s" % epoch, "before_nms")
    final_path = os.path.join(work_dir, "test", "submit_%s" % epoch, "after_nms")
    for p in (save_path, final_path):
        if os.path.exists(p):
            shutil.rmtree(p)
    zips = os.path.join(work_dir, "submit_zips")
    os.makedirs(zips, exist_ok=True)
    data_merge(results, save_path, final_path, dataset_type, device, nms_threshold_type)
    if dataset_type == 'FAIR1M_1_5':
        fair = os.path.join(work_dir, "test", "submit_%s" % epoch, "final_fair1m_1_5", "test")
        csv = dota_to_fair1m_1_5(final_path, fair, images_dir, name)
        out = os.path.join(zips, name + ".csv")
        shutil.move(csv, out)
        return out
    out = os.path.join(zips, name + ".zip")
    if os.path.exists(out):
        os.remove(out)
    if dataset_type == 'FAIR':
        fair = os.path.join(work_dir, "test", "submit_%s" % epoch, "final_fair", "test")
        if os.path.exists(fair):
            shutil.rmtree(fair)
        dota_to_fair(final_path, fair, images_dir)
        with zipfile.ZipFile(out, 'w', zipfile.ZIP_DEFLATED) as z:
            for f in sorted(os.listdir(fair)):
                z.write(os.path.join(fair, f), os.path.join("test", f))    # `zip -r name.zip test`
        return out
    with zipfile.ZipFile(out, 'w', zipfile.ZIP_DEFLATED) as z:
        for f in sorted(os.listdir(final_path)):
            z.write(os.path.join(final_path, f), f)          # `zip -rj`: junk the paths
    return out
