"""DOTADataset (/root/reference/python/jdet/data/dota.py:22-143): CustomDataset + the DOTA class tables, the
``balance_category`` re-sampling, Task-1 result files and mAP (polygon IoU on the GPU, data/devkits/voc_eval.py)."""
import os

import numpy as np

from rs_detection_amd.config.constant import get_classes_by_name
from rs_detection_amd.utils.registry import DATASETS
from .box_np import rotated_box_to_poly_np, rotated_box_to_poly_single
from .custom import CustomDataset


def s2anet_post(result):
    """:13-19."""
    dets, labels = result
    return rotated_box_to_poly_np(dets[:, :5], 'le135'), dets[:, 5], labels + 1


@DATASETS.register_module()
class DOTADataset(CustomDataset):
    # :46-57 (class name -> (whole copies, extra head copies))
    BALANCE = {"storage-tank": (1, 526), "baseball-diamond": (2, 202), "ground-track-field": (1, 575),
               "swimming-pool": (2, 104), "soccer-ball-field": (1, 962), "roundabout": (1, 711),
               "tennis-court": (1, 655), "basketball-court": (4, 0), "helicopter": (8, 0), "container-crane": (50, 0)}

    def __init__(self, *arg, balance_category=False, version='1', **kwargs):
        assert version in ['1', '1_5', '2']
        self.CLASSES = get_classes_by_name('DOTA' + version)
        super().__init__(*arg, **kwargs)
        if balance_category:
            self.img_infos = self._balance_categories()
            self.total_len = len(self.img_infos)

    def _balance_categories(self):
        cate_dict = {}
        for idx, img_info in enumerate(self.img_infos):
            for label in np.unique(img_info["ann"]["labels"]):
                cate_dict.setdefault(label, []).append(idx)
        new_idx = []
        for k, d in cate_dict.items():
            l1, l2 = self.BALANCE.get(self.CLASSES[k - 1], (1, 0))
            new_idx.extend(d * l1 + d[:l2])
        return [self.img_infos[i] for i in new_idx]

    def parse_result(self, results, save_path):
        """:64-81: [((dets (n,6), labels (n,)), image name)] -> one Task-1 file per class."""
        os.makedirs(save_path, exist_ok=True)
        data = {}
        for (dets, labels), img_name in results:
            img_name = os.path.splitext(img_name)[0]
            for det, label in zip(dets, labels):
                bbox = rotated_box_to_poly_single(det[:5])
                data.setdefault(self.CLASSES[label], []).append(
                    '{} {:.4f} {:.4f} {:.4f} {:.4f} {:.4f} {:.4f} {:.4f} {:.4f} {:.4f}\n'.format(img_name, det[5], *bbox))
        for classname, lines in data.items():
            with open(os.path.join(save_path, classname + '.txt'), 'w') as f:
                f.writelines(lines)

    def evaluate(self, results, work_dir=None, epoch=0, logger=None, save=False, device="cuda", pairwise=None):
        """:83-143: results = [((polys, scores, labels 0-based), target)] -> {"eval/<i>_<class>_AP", "eval/0_meanAP"}."""
        from .devkits import evaluate_dota
        return evaluate_dota(results, self.CLASSES, device=device, pairwise=pairwise)
