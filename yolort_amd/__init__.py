"""yolort_amd: MI355X-native (gfx950) YOLOv5 inference hot path behind yolort's model API.

    from yolort_amd.models import yolov5s
    model = yolov5s(score_thresh=0.25).half().cuda().eval()
    detections = model.predict([img0, img1])      # List[Dict(scores, labels, boxes)]

Compute runs exclusively in hand-written HIP kernels (yolort_amd/csrc, C ABI in
include/yolort_amd.h); there is no CPU or eager-PyTorch fallback.
"""
__version__ = "0.1.0"
