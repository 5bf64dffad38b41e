"""FAIR1M datasets of this fork (/root/reference/python/jdet/data/fair.py:9-150): DOTADataset with the FAIR1M class
tables and their own ``balance_category`` multipliers.  ``FAIR1M_1_5_Dataset`` (ten coarse classes; configs[3] and the
S2ANet-R101 ms config train on it) additionally drops boxes of area <= 1 px at load time (:99-107)."""
import numpy as np

from rs_detection_amd.config.constant import FAIR1M_1_5_CLASSES, FAIR_CLASSES_
from rs_detection_amd.utils.registry import DATASETS
from .dota import DOTADataset


class _FairBase(DOTADataset):
    def __init__(self, *arg, balance_category=False, **kwargs):
        # DOTADataset resolves CLASSES from its ``version``; these tables are fixed per class
        super().__init__(*arg, balance_category=False, **kwargs)
        self.CLASSES = type(self).CLASSES
        self._post_load()
        if balance_category:
            self.img_infos = self._balance_categories()
            self.total_len = len(self.img_infos)

    def _post_load(self):
        pass


@DATASETS.register_module()
class FAIR1M_1_5_Dataset(_FairBase):
    CLASSES = FAIR1M_1_5_CLASSES
    # :135-146 (the active table)
    BALANCE = {'Airplane': (1, 0), 'Ship': (2, 0), 'Vehicle': (1, 0), 'Basketball_Court': (2, 0),
               'Tennis_Court': (1, 0), 'Football_Field': (2, 0), 'Baseball_Field': (2, 0), 'Intersection': (4, 0),
               'Roundabout': (1, 0), 'Bridge': (8, 0)}

    def _post_load(self):
        for info in self.img_infos:
            boxes = info["ann"]["bboxes"]
            if boxes.shape[0] == 0:
                continue
            keep = boxes[:, 2] * boxes[:, 3] > 1.
            info["ann"]["bboxes"], info["ann"]["labels"] = boxes[keep], info["ann"]["labels"][keep]


@DATASETS.register_module()
class FAIRDataset(_FairBase):
    CLASSES = FAIR_CLASSES_
    BALANCE = {"C919": (8, 0), "ARJ21": (7, 0), "Tractor": (5, 0)}     # :79-83
