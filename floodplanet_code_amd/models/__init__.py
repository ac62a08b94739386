"""Model registry -- mirror of st_water_seg/models/__init__.py:5-20.

``build_model(model_name, input_channels, n_classes, lr, log_image_iter, to_rgb_fcn, ignore_index, **kwargs)``
forwards positionally exactly like the reference, so ``fit.py:66-73`` / ``predict.py:164-171`` /
``infer.py:86-93`` can import this module in place of ``st_water_seg.models``."""
from .ef_model import EarlyFusionModel
from .lf_model import LateFusionModel
from .water_seg_model import WaterSegmentationModel


MODELS = {
    'ms_model': WaterSegmentationModel,
    'ef_model': EarlyFusionModel,
    'lf_model': LateFusionModel,
}


def build_model(model_name, input_channels, n_classes, lr, log_image_iter, to_rgb_fcn, ignore_index, **kwargs):
    model_cls = MODELS.get(model_name)
    if model_cls is None:
        # the reference prints and then trips over its unbound local (models/__init__.py:18-20)
        print(f'Could not find model named: {model_name}')
        raise UnboundLocalError("local variable 'model' referenced before assignment")
    return model_cls(input_channels, n_classes, lr, log_image_iter, to_rgb_fcn, ignore_index, **kwargs)
