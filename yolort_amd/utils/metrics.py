"""Detection metrics on the host (numpy): COCO-style average precision without pycocotools.

`coco_ap` restates the published COCO detection protocol (AP averaged over IoU 0.50:0.05:0.95, 101-point interpolated
precision, greedy score-ordered matching per class and image) for box detections.  `DetectionEvaluator` offers the
update()/compute() surface of the reference's evaluator (yolort/data/coco_eval.py:28-120, which wraps pycocotools --
absent here) on in-memory ground truth: SURVEY.md 8f-4.  COCO's maxDets = 100 cap (the top-100 detections of an image by
score are evaluated) and its exact threshold grid linspace(.5, .95, 10) are applied; crowd boxes and the small / medium /
large area ranges are not modelled (no crowd boxes, all areas).
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence

import numpy as np


def _iou_matrix(a, b):
    x1 = np.maximum(a[:, None, 0], b[None, :, 0]); y1 = np.maximum(a[:, None, 1], b[None, :, 1])
    x2 = np.minimum(a[:, None, 2], b[None, :, 2]); y2 = np.minimum(a[:, None, 3], b[None, :, 3])
    inter = np.clip(x2 - x1, 0, None) * np.clip(y2 - y1, 0, None)
    aa = (a[:, 2] - a[:, 0]) * (a[:, 3] - a[:, 1]); ab = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    return inter / (aa[:, None] + ab[None, :] - inter + 1e-12)


COCO_IOU_THRS = np.linspace(0.5, 0.95, 10)   # pycocotools Params.iouThrs (np.arange(.5, .96, .05) drifts: 0.7000000000000001 ...)
COCO_MAX_DETS = 100


def _top_dets(d, max_dets):
    """the `max_dets` best detections of one image by score (stable), as pycocotools evaluates them (maxDets)"""
    if max_dets is None or len(d["scores"]) <= max_dets:
        return d
    keep = np.argsort(-d["scores"], kind="stable")[:max_dets]
    return {"boxes": d["boxes"][keep], "scores": d["scores"][keep], "labels": d["labels"][keep]}


def coco_ap(refs, dets, num_classes=80, thrs=None, max_dets=None):
    """COCO-style AP@[.5:.95] (101-point interpolation, greedy score-ordered matching per class and image) of `dets`
    with the ORACLE's detections `refs` as ground truth -- the 'mAP vs ref' of SURVEY.md 8d.  Lists of per-image dicts
    of numpy arrays {boxes (n,4), scores (n), labels (n)}.  `max_dets`: keep only the best N detections per image
    (COCO: 100); None scores everything handed in (the bench's mAP-vs-ref compares full 300-detection outputs)."""
    thrs = COCO_IOU_THRS if thrs is None else np.asarray(thrs)
    dets = [_top_dets(d, max_dets) for d in dets]
    aps = []
    for c in range(num_classes):
        n_gt = sum(int((r["labels"] == c).sum()) for r in refs)
        if n_gt == 0:
            continue
        recs = []   # (score, tp flags per threshold)
        for r, d in zip(refs, dets):
            gb = r["boxes"][r["labels"] == c]
            m = d["labels"] == c
            db, ds = d["boxes"][m], d["scores"][m]
            order = np.argsort(-ds, kind="stable")
            db, ds = db[order], ds[order]
            iou = _iou_matrix(db, gb) if len(db) and len(gb) else np.zeros((len(db), len(gb)))
            tp = np.zeros((len(db), len(thrs)), bool)
            for ti, t in enumerate(thrs):
                used = np.zeros(len(gb), bool)
                for i in range(len(db)):
                    cand = np.where(~used & (iou[i] >= t))[0]
                    if len(cand):
                        j = cand[np.argmax(iou[i, cand])]
                        used[j] = True
                        tp[i, ti] = True
            recs += [(float(ds[i]), tp[i]) for i in range(len(db))]
        if not recs:
            aps.append(0.0)
            continue
        recs.sort(key=lambda x: -x[0])
        tps = np.stack([x[1] for x in recs]).astype(np.float64)
        ap_t = []
        for ti in range(len(thrs)):
            ctp = np.cumsum(tps[:, ti]); cfp = np.cumsum(1.0 - tps[:, ti])
            rec = ctp / n_gt; prec = ctp / np.maximum(ctp + cfp, 1e-12)
            for i in range(len(prec) - 2, -1, -1):
                prec[i] = max(prec[i], prec[i + 1])
            q = np.zeros(101)
            inds = np.searchsorted(rec, np.linspace(0, 1, 101), side="left")
            ok = inds < len(prec)
            q[ok] = prec[inds[ok]]
            ap_t.append(q.mean())
        aps.append(float(np.mean(ap_t)))
    return float(np.mean(aps)) if aps else None




class DetectionEvaluator:
    """Accumulates detections and ground truth per image; compute() returns COCO-style numbers in the 0-100 range
    (like the reference: yolort/data/coco_eval.py:31-34), -1 when nothing can be scored."""

    def __init__(self, num_classes: int = 80, max_dets: Optional[int] = COCO_MAX_DETS):
        self.num_classes = num_classes
        self.max_dets = max_dets   # COCO evaluates the top-100 detections per image (the model's default keeps 300)
        self._preds: List[Dict[str, np.ndarray]] = []
        self._gts: List[Dict[str, np.ndarray]] = []

    @staticmethod
    def _np(d, keys) -> Dict[str, np.ndarray]:
        out = {}
        for k in keys:
            v = d[k]
            if hasattr(v, "detach"):
                v = v.detach().float().cpu().numpy() if k != "labels" else v.detach().cpu().numpy()
            out[k] = np.asarray(v)
        return out

    def update(self, preds: Sequence[Dict], targets: Sequence[Dict]) -> None:
        """preds: the model's List[Dict{boxes, scores, labels}]; targets: List[Dict{boxes, labels}] (xyxy, same image order)"""
        if len(preds) != len(targets):
            raise ValueError("one target per prediction is required")
        for p, t in zip(preds, targets):
            self._preds.append(self._np(p, ("boxes", "scores", "labels")))
            g = self._np(t, ("boxes", "labels"))
            g["scores"] = np.ones(len(g["labels"]), np.float32)
            self._gts.append(g)

    def merge(self, other: "DetectionEvaluator") -> None:
        """fold in another rank's accumulator (ranks own disjoint image shards)"""
        self._preds += other._preds
        self._gts += other._gts

    def compute(self) -> Dict[str, float]:
        if not self._gts:
            return {"AP": -1.0, "AP50": -1.0, "AP75": -1.0}
        ap = coco_ap(self._gts, self._preds, self.num_classes, max_dets=self.max_dets)
        ap50 = coco_ap(self._gts, self._preds, self.num_classes, thrs=np.array([0.5]), max_dets=self.max_dets)
        ap75 = coco_ap(self._gts, self._preds, self.num_classes, thrs=np.array([0.75]), max_dets=self.max_dets)
        f = lambda v: -1.0 if v is None else 100.0 * v  # noqa: E731
        return {"AP": f(ap), "AP50": f(ap50), "AP75": f(ap75)}
