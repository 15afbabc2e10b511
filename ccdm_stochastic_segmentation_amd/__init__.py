"""MI355X-native conditional categorical diffusion sampler (hot path of
LarsDoorenbos/ccdm-stochastic-segmentation): HIP kernels behind the reference's build_model /
DenoisingModel interface.  See DESIGN.md and include/ccdm_hip.h."""
from .unet_spec import UNetSpec, make_unet_spec, make_synthetic_state_dict, default_channel_mult  # noqa: F401
from .models import (build_model, DiffusionModel, DenoisingModel, UNetModel, OneHotCategoricalBCHW,  # noqa: F401
                     linear_schedule, cosine_schedule, step_values)
from . import hip  # noqa: F401

__version__ = "0.1.0"
