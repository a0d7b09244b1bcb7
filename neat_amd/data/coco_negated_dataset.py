"""TTA datasets over a NegBench COCO-negated CSV (reference src/data/coco_negated_dataset.py:12-125).

CSV schema (sep ','): ``filepath`` and four columns holding Python-literal lists of equal length per row --
``captions``, ``pos_captions``, ``neg_captions``, ``inverted_captions`` (parsed with ``ast.literal_eval``).
Caption row r of image i gets text id = running count; ``txt2img[txt_id] = [img_id]``, ``img2txt[img_id] = [txt ids]``.

The image dataset hands out the DECODED image as a uint8 [H, W, 3] tensor; resize / ToTensor / Normalize run on the
GPU in the loader (neat_amd.data.GpuImageLoader -> neat_image_preprocess), not in the dataset.
"""
import ast
import csv

import numpy as np
import torch
from torch.utils.data import Dataset

from .utils import pre_caption


def _safe_eval(x):
    return ast.literal_eval(x) if isinstance(x, str) else x


def _read_csv(csv_file, sep):
    with open(csv_file, newline="") as f:
        return list(csv.DictReader(f, delimiter=sep))


def decode_rgb_u8(path) -> torch.Tensor:
    """File -> uint8 [H, W, 3] (PIL decode + convert('RGB'), like the reference's __getitem__)."""
    from PIL import Image
    with Image.open(path) as im:
        return torch.from_numpy(np.asarray(im.convert("RGB")).copy())


class coco_negated_retrieval_eval_image(Dataset):
    def __init__(self, transform, csv_file, sep=',', img_key='filepath', caption_key='captions',
                 caption_key_pos='pos_captions', caption_key_neg='neg_captions', max_words=77):
        rows = _read_csv(csv_file, sep)
        self.annotation = [dict(image=r[img_key], text=_safe_eval(r[caption_key])) for r in rows]
        self.transform = transform
        self.text, self.image, self.txt2img, self.img2txt = [], [], {}, {}
        txt_id = 0
        for img_id, ann in enumerate(self.annotation):
            self.image.append(ann['image'])
            self.img2txt[img_id] = []
            for caption in ann['text']:
                self.text.append(pre_caption(caption, max_words))
                self.img2txt[img_id].append(txt_id)
                self.txt2img[txt_id] = [img_id]
                txt_id += 1

    def __len__(self):
        return len(self.annotation)

    def __getitem__(self, index):
        image = decode_rgb_u8(self.annotation[index]['image'])
        if self.transform is not None:
            image = self.transform(image)
        return image, index


class coco_negated_retrieval_eval_text(Dataset):
    def __init__(self, transform, csv_file, sep=',', img_key='filepath', caption_key='captions',
                 caption_key_pos='pos_captions', caption_key_neg='neg_captions',
                 caption_key_inv='inverted_captions', caption_key_ori='ori_captions', max_words=77):
        rows = _read_csv(csv_file, sep)
        self.annotation = [dict(image=r[img_key], text=_safe_eval(r[caption_key]), pos_text=_safe_eval(r[caption_key_pos]),
                                neg_text=_safe_eval(r[caption_key_neg]), inv_text=_safe_eval(r[caption_key_inv]))
                           for r in rows]
        self.transform = transform
        self.text, self.pos_text, self.neg_text, self.inv_text, self.image = [], [], [], [], []
        self.txt2img, self.img2txt = {}, {}
        txt_id = 0
        for img_id, ann in enumerate(self.annotation):
            self.image.append(ann['image'])
            self.img2txt[img_id] = []
            for i, caption in enumerate(ann['text']):
                self.text.append(pre_caption(caption, max_words))
                self.pos_text.append(pre_caption(ann['pos_text'][i], max_words))
                self.neg_text.append(pre_caption(ann['neg_text'][i], max_words))
                self.inv_text.append(pre_caption(ann['inv_text'][i], max_words))
                self.img2txt[img_id].append(txt_id)
                self.txt2img[txt_id] = [img_id]
                txt_id += 1

    def __len__(self):
        return len(self.text)

    def __getitem__(self, index):
        return self.text[index], self.pos_text[index], self.neg_text[index], self.inv_text[index], index
