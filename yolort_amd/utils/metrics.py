"""Detection metrics on the host (numpy): COCO-style average precision without pycocotools.

`coco_ap` restates the published COCO detection protocol (AP averaged over IoU 0.50:0.05:0.95, 101-point interpolated
precision, greedy score-ordered matching per class and image) for box detections.  `DetectionEvaluator` offers the
update()/compute() surface of the reference's evaluator (yolort/data/coco_eval.py:28-120, which wraps pycocotools --
absent here) on in-memory ground truth: SURVEY.md 8f-4.  COCO's maxDets = 100 cap -- per (image, category), as pycocotools
applies it -- its exact threshold grid linspace(.5, .95, 10) and its small / medium / large area ranges (ignore semantics of
COCOeval.evaluateImg) are applied; crowd boxes are not modelled (the in-memory ground truth has none).  `update_from_slab`
feeds the evaluator from the fixed-shape detection slab the ranks all-gather (yolort_amd/dist.py), so every rank scores the
GLOBAL batch without the reference's pickle exchange (yolort/data/coco_eval.py:225-226 -> data/distributed.py:6-49).
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
# pycocotools Params.areaRng / areaRngLbl (the reference's summary lines, yolort/data/coco_eval.py:86-218 -> COCOeval.summarize)
COCO_AREA_RNG = {"all": (0.0, 1e10), "small": (0.0, 32.0 ** 2), "medium": (32.0 ** 2, 96.0 ** 2), "large": (96.0 ** 2, 1e10)}


def _area(b):
    return (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])


def coco_ap(refs, dets, num_classes=80, thrs=None, max_dets=None, area_rng=None):
    """COCO-style AP@[.5:.95] (101-point interpolation, greedy score-ordered matching per class and image) of `dets`
    with the ORACLE's detections `refs` as ground truth -- the 'mAP vs ref' of SURVEY.md 8d.  Lists of per-image dicts
    of numpy arrays {boxes (n,4), scores (n), labels (n)}.

    `max_dets`: as pycocotools applies maxDets -- PER (image, category): `dt = _dts[imgId, catId]` sorted by score, cut at
    `[0:maxDet]` (COCO: 100); None scores everything handed in (the bench's mAP-vs-ref compares full 300-detection outputs).
    `area_rng` = (lo, hi): COCO's area ranges -- ground truth outside the range is IGNORED (a detection matched to it counts neither
    as true nor as false positive; non-ignored ground truth is preferred in the matching), and so is an unmatched detection whose
    own area is outside the range (pycocotools COCOeval.evaluateImg)."""
    thrs = COCO_IOU_THRS if thrs is None else np.asarray(thrs)
    aps = []
    for c in range(num_classes):
        n_gt = 0
        recs = []   # (score, tp flags per threshold, ignore flags per threshold)
        for r, d in zip(refs, dets):
            gb = r["boxes"][r["labels"] == c]
            g_ign = np.zeros(len(gb), bool)
            if area_rng is not None and len(gb):
                ga = _area(gb)
                g_ign = (ga < area_rng[0]) | (ga > area_rng[1])
            go = np.argsort(g_ign, kind="stable")            # non-ignored ground truth first
            gb, g_ign = gb[go], g_ign[go]
            n_gt += int((~g_ign).sum())
            m = d["labels"] == c
            db, ds = d["boxes"][m], d["scores"][m]
            order = np.argsort(-ds, kind="stable")
            if max_dets is not None:
                order = order[:max_dets]                     # maxDets per (image, category)
            db, ds = db[order], ds[order]
            iou = _iou_matrix(db, gb) if len(db) and len(gb) else np.zeros((len(db), len(gb)))
            d_out = np.zeros(len(db), bool)
            if area_rng is not None and len(db):
                da = _area(db)
                d_out = (da < area_rng[0]) | (da > area_rng[1])
            tp = np.zeros((len(db), len(thrs)), bool)
            ig = np.zeros((len(db), len(thrs)), bool)
            for ti, t in enumerate(thrs):
                used = np.zeros(len(gb), bool)
                for i in range(len(db)):
                    best, bj = min(t, 1 - 1e-10), -1
                    for j in range(len(gb)):
                        if used[j]:
                            continue
                        if bj > -1 and not g_ign[bj] and g_ign[j]:
                            break                             # matched to real ground truth: ignored boxes are not considered
                        if iou[i, j] < best:
                            continue
                        best, bj = iou[i, j], j
                    if bj >= 0:
                        used[bj] = True
                        tp[i, ti] = not g_ign[bj]
                        ig[i, ti] = g_ign[bj]
                    else:
                        ig[i, ti] = d_out[i]
            recs += [(float(ds[i]), tp[i], ig[i]) for i in range(len(db))]
        if n_gt == 0:
            continue
        if not recs:
            aps.append(0.0)
            continue
        recs.sort(key=lambda x: -x[0])   # stable: equal scores keep image order, like pycocotools' mergesort
        tps = np.stack([x[1] for x in recs]).astype(np.float64)
        igs = np.stack([x[2] for x in recs])
        ap_t = []
        for ti in range(len(thrs)):
            keep = ~igs[:, ti]
            t_ = tps[keep, ti]
            ctp = np.cumsum(t_); cfp = np.cumsum(1.0 - t_)
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
        self.max_dets = max_dets   # COCO evaluates the top-100 detections per (image, category)
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

    def update_from_slab(self, slab, targets: Sequence[Dict]) -> None:
        """slab = (boxes (N,K,4), scores (N,K), labels (N,K), counts (N)) of the GLOBAL batch in rank order -- what
        `PendingDetections.gathered()` / `dist.all_gather_slab` return; targets: ground truth of the same N images in the same order"""
        boxes, scores, labels, counts = [t.detach().cpu() if hasattr(t, "detach") else t for t in slab]
        cl = [int(c) for c in counts]
        if any(c < 0 for c in cl):
            raise ValueError("the slab still holds a stale shard: take PendingDetections.gathered() (it resolves the second round) first")
        self.update([{"boxes": boxes[i, :c], "scores": scores[i, :c], "labels": labels[i, :c]} for i, c in enumerate(cl)], targets)

    def merge(self, other: "DetectionEvaluator") -> None:
        """fold in another rank's accumulator (ranks own disjoint image shards)"""
        self._preds += other._preds
        self._gts += other._gts

    def compute(self) -> Dict[str, float]:
        if not self._gts:
            return {"AP": -1.0, "AP50": -1.0, "AP75": -1.0, "APs": -1.0, "APm": -1.0, "APl": -1.0}
        ap = coco_ap(self._gts, self._preds, self.num_classes, max_dets=self.max_dets)
        ap50 = coco_ap(self._gts, self._preds, self.num_classes, thrs=np.array([0.5]), max_dets=self.max_dets)
        ap75 = coco_ap(self._gts, self._preds, self.num_classes, thrs=np.array([0.75]), max_dets=self.max_dets)
        f = lambda v: -1.0 if v is None else 100.0 * v  # noqa: E731
        out = {"AP": f(ap), "AP50": f(ap50), "AP75": f(ap75)}
        for name, key in (("small", "APs"), ("medium", "APm"), ("large", "APl")):   # COCOeval.summarize lines 4-6
            out[key] = f(coco_ap(self._gts, self._preds, self.num_classes, max_dets=self.max_dets, area_rng=COCO_AREA_RNG[name]))
        return out
