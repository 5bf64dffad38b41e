"""CustomDataset (/root/reference/python/jdet/data/custom.py:14-120): ``labels.pkl`` + an ``images/`` folder, the
reference's annotation schema, transforms, batching and collate.  Jittor's Dataset is its own loader (batch_size,
shuffle, drop_last live on the dataset); here the dataset is iterable in the same way -- ``for images, targets in ds``
-- and shards the index stream across ranks (one process per GPU, DistributedSampler-style split)."""
import os
import pickle

import numpy as np
from PIL import Image

from rs_detection_amd.utils.registry import DATASETS
from .box_np import rotated_box_to_bbox_np
from .transforms import Compose


@DATASETS.register_module()
class CustomDataset:
    CLASSES = None

    def __init__(self, images_dir=None, annotations_file=None, dataset_dir=None, transforms=None, batch_size=1,
                 num_workers=0, shuffle=False, drop_last=False, filter_empty_gt=True, filter_min_size=-1, seed=0):
        if dataset_dir is not None:
            assert images_dir is None and annotations_file is None
            images_dir = os.path.join(dataset_dir, "images")
            annotations_file = os.path.join(dataset_dir, "labels.pkl")
        assert images_dir is not None and annotations_file is not None
        self.images_dir, self.annotations_file = os.path.abspath(images_dir), os.path.abspath(annotations_file)
        self.batch_size, self.num_workers, self.shuffle, self.drop_last = batch_size, num_workers, shuffle, drop_last
        self.transforms = Compose(transforms)
        with open(self.annotations_file, "rb") as f:
            self.img_infos = pickle.load(f)      # jt.load of a .pkl is a pickle of plain dicts / ndarrays
        if filter_empty_gt:
            self.img_infos = self._filter_imgs(filter_min_size)
        self.total_len = len(self.img_infos)
        self.seed, self.epoch = seed, 0
        self.rank, self.world_size = 0, 1

    def _filter_imgs(self, min_size):
        return [i for i in self.img_infos
                if len(i["ann"]["bboxes"]) > 0 and min(i['width'], i['height']) >= min_size]

    def __len__(self):
        return self.total_len

    def set_epoch(self, epoch):
        self.epoch = epoch

    def set_shard(self, rank, world_size, keep_all=False):
        """``keep_all``: evaluation sharding -- every image exactly once, ranks may get unequal counts (no collective
        runs per batch there); training trims to whole global batches."""
        self.rank, self.world_size, self.shard_keep_all = rank, world_size, keep_all

    def _read_ann_info(self, idx):
        while True:
            img_info = self.img_infos[idx]
            if len(img_info["ann"]["bboxes"]) > 0:
                break
            idx = np.random.choice(np.arange(self.total_len))
        anno = img_info["ann"]
        img_path = os.path.join(self.images_dir, img_info["filename"])
        image = Image.open(img_path).convert("RGB")
        width, height = image.size
        assert width == img_info['width'] and height == img_info["height"], "image size is different from annotations"
        ignore = anno.get("bboxes_ignore", np.zeros((0, 5), np.float32))
        hboxes, polys = rotated_box_to_bbox_np(anno["bboxes"])
        hboxes_ignore, polys_ignore = rotated_box_to_bbox_np(ignore)
        ann = dict(rboxes=anno['bboxes'].astype(np.float32), hboxes=hboxes.astype(np.float32),
                   polys=polys.astype(np.float32), labels=anno['labels'].astype(np.int32),
                   rboxes_ignore=ignore.astype(np.float32), hboxes_ignore=hboxes_ignore, polys_ignore=polys_ignore,
                   classes=self.CLASSES, ori_img_size=(width, height), img_size=(width, height), scale_factor=1.0,
                   filename=img_info["filename"], img_file=img_path)
        return image, ann

    def __getitem__(self, idx):
        if "BATCH_IDX" in os.environ:
            idx = int(os.environ['BATCH_IDX'])
        image, anno = self._read_ann_info(idx)
        if self.transforms is not None:
            image, anno = self.transforms(image, anno)
        return image, anno

    def collate_batch(self, batch):
        """:92-108: zero-pad to the largest height / width of the batch.  With ``self.reuse_batch_buffers`` (switched on
        by data/loader.prefetch_to_device for the duration of its loop) the batch array is one of a ring of long-lived
        -- pinned, where a GPU is present -- buffers: valid until five further batches have been collated."""
        imgs, anns = [b[0] for b in batch], [b[1] for b in batch]
        max_h, max_w = max(i.shape[-2] for i in imgs), max(i.shape[-1] for i in imgs)
        if getattr(self, "reuse_batch_buffers", False):
            from .loader import reusable_batch_buffer
            batch_imgs = reusable_batch_buffer(self, (len(imgs), 3, max_h, max_w), np.float32)
            if any(i.shape[-2] != max_h or i.shape[-1] != max_w for i in imgs):
                batch_imgs.fill(0)
        else:
            batch_imgs = np.zeros((len(imgs), 3, max_h, max_w), dtype=np.float32)
        for i, image in enumerate(imgs):
            batch_imgs[i, :, :image.shape[-2], :image.shape[-1]] = image
        return batch_imgs, anns

    def _indices(self):
        idx = np.arange(self.total_len)
        if self.shuffle:
            np.random.default_rng(self.seed + self.epoch).shuffle(idx)
        per = self.batch_size * self.world_size
        if self.world_size > 1 and not getattr(self, "shard_keep_all", False):
            # every rank must run the same number of steps (DDP's all-reduce is collective): trim to whole global
            # batches, and refuse a dataset that cannot fill even one
            if len(idx) < per:
                raise ValueError("dataset of %d images cannot fill one batch of %d on each of %d ranks"
                                 % (len(idx), self.batch_size, self.world_size))
            idx = idx[:(len(idx) // per) * per]
        elif self.drop_last:
            idx = idx[:(len(idx) // per) * per]
        return idx[self.rank::self.world_size] if self.world_size > 1 else idx

    def __iter__(self):
        """Batches of this rank's share of the epoch.  ``num_workers > 0``: decode + transforms in worker processes
        (data/loader.py), a few batches ahead; the batches are the same either way (per-sample random state)."""
        from .loader import iterate_samples
        idx = self._indices()
        n_full = (len(idx) // self.batch_size) * self.batch_size
        if self.drop_last:
            idx = idx[:n_full]
        batch = []
        for sample in iterate_samples(self, idx):
            batch.append(sample)
            if len(batch) == self.batch_size:
                yield self.collate_batch(batch)
                batch = []
        if batch:
            yield self.collate_batch(batch)

    def __getstate__(self):
        from .loader import state_without_pool
        return state_without_pool(self)

    def evaluate(self, results, work_dir, epoch, logger=None):
        raise NotImplementedError
