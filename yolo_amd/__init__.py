"""yolo_amd: MI355X (gfx950) native YOLOv3 hot path -- Darknet-style backbone + 3-scale heads,
anchor decode, top-1 / NMS -- as hand-written HIP kernels behind a C ABI (include/yolo_amd.h)."""
from .spec import NetGraph, ConvSpec, darknet53_spec      # noqa: F401


def __getattr__(name):
    # torch-dependent pieces are imported lazily so `import yolo_amd` works in tooling contexts
    if name in ('CarNet', 'CarLPNet'):
        from . import net
        return getattr(net, name)
    if name in ('Detector', 'get_iou', 'cv_img_2_ndarray', 'make_grid', 'predict_LP', 'predict_LP_batch', 'default_ltrb'):
        from . import detect
        return getattr(detect, name)
    raise AttributeError(name)
