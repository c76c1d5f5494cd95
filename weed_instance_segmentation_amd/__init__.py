"""MI355X-native Mask2Former hot path for marco-conciatori-public/weed_instance_segmentation.

    from weed_instance_segmentation_amd import Mask2FormerForUniversalSegmentation, Mask2FormerConfig

is the drop-in for `transformers.Mask2FormerForUniversalSegmentation` at the reference's call sites
(models/mask2former/train.py:7, :167-172; models/model_utils.py:5, :14).  The kernels live in
libwm2f.so (C ABI: include/wm2f.h); build it with `python -m weed_instance_segmentation_amd._build`.
"""
from .configuration import Mask2FormerConfig  # noqa: F401
from .modeling import Mask2FormerForUniversalSegmentation, Mask2FormerForUniversalSegmentationOutput  # noqa: F401

__version__ = "0.1.0"
from .postprocess import Mask2FormerInstancePostProcessor  # noqa: F401
