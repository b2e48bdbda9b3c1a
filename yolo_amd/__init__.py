"""yolo_amd - the detector inference hot path of Jex-y/yolo on AMD MI355X (gfx950).

Same module layout and names as the reference package (its __init__.py:1-8 exposes
``config`` narrowed to the ``yolo:`` section, the utils functions, ``backbones`` and
``detectors``); every tensor operation runs in hand-written HIP kernels behind the C ABI
declared in include/ymx.h.  There is no CPU fallback.
"""
from .config import config
from .utils import *          # noqa: F401,F403
from .utils import (decode, decode_train, decode_predict, filter_bboxes, bbox_nms, bbox_iou, bbox_giou,
                    combined_non_max_suppression, image_preprocess, draw_bbox, draw_boxes)
from . import layers
from . import backbones
from . import detectors
from . import losses
from . import dataset
from . import tfrecord
from . import engine
from . import dist
from .engine import Detector

config = config["yolo"]

__version__ = "1.0.0"
