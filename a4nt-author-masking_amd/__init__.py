"""a4nt-author-masking_amd: MI355X-native (gfx950) implementation of the A4NT adversarial-training
hot path behind the reference's own nn.Module API.

    from a4nt_amd import CharTranslator, CharLstm, CharCNN, get_classifier     # drop-in model layer
    from a4nt_amd import adv_forward_pass, d_update, g_update                  # restated training step

(the directory name has a hyphen; `a4nt_amd.py` at the repository root aliases it).
The compute path is liba4nt_hip.so (csrc/, C ABI in include/a4nt_hip.h); importing this package
without the built library, or running it without a gfx950 GPU, raises -- there is no CPU fallback.
"""
from ._lib import A4NTError, LIB_PATH, EXPORTS, load as load_library      # noqa: F401
from .gen_decode import SoftSamples                                       # noqa: F401
from .char_translator import CharTranslator                               # noqa: F401
from .char_lstm import CharLstm                                           # noqa: F401
from .char_cnn import CharCNN                                             # noqa: F401
from .model_utils import get_classifier                                   # noqa: F401
from .adv_step import (adv_forward_pass, d_update, g_update, fused_update, make_optimizers,    # noqa: F401
                       GradAllReducer, synthetic_batch)
from .data_provider import DataProvider                                   # noqa: F401
from . import eval_utils                                                  # noqa: F401
from .ops import check_timeouts, lstm_set_max_workgroups                  # noqa: F401

load_library()
