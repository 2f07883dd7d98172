"""xlstm_hved_amd: MI355X-native (gfx950) implementation of the XLSTM-HVED volumetric forward/backward hot path.

Import name `xlstm_hved_amd` (see /xlstm_hved_amd.py for the alias of this hyphenated directory).
Compute lives in csrc/*.hip behind the C ABI of include/xlstm_hved.h; this package is the host-side mirror of the
reference's nn.Module surface."""
from . import _lib, functional, ops, parallel  # noqa: F401
from .blocks import (AttenModule2, BasicConv, ChannelPool, Decoder, DoubleConv, DoubleConv_ViL, DuSEAttention, Encoder,  # noqa: F401
                     ProductOfExperts, ProductOfExperts2, ResBlock, SingleConv, SkipReturnAttention, SpacialAttention3D,
                     Upsampling, ViLLayer, number_of_features_per_level)
from .disc import Discriminator  # noqa: F401
from .model import (MODELS, SUBSETS_MODALITIES, AbstractFusion3DUNet, ReconDecoder, Seg_Recon_DuSFEDecoder,  # noqa: F401
                    U_HVEDConvDuSFEmViLNet3D, U_HVEDConvDuSFEmViLSkrNet3D, U_HVEDConvDuSFENet3D, U_HVEDConvDuSFESkrNet3D,
                    U_HVEDConvNet3D, U_HVEDConvXLSTMNet3D, XLSTM_HVED, XLSTM_HVED_woDuSFE, XLSTM_HVED_woSMVAE,
                    XLSTM_HVED_woViL, find_model_using_name)
from .utils import init_weights, seed_everything, subset_idx  # noqa: F401
from . import inference, losses  # noqa: F401
from .losses import (DiceCoefficient, DiceLoss, DiceRegion, GANLoss, MSELoss, combine, compute_KLD, compute_KLD_levels,  # noqa: F401
                     gan_pair_loss, mse_loss, nested_attention)
from .inference import eval_overlap_volume  # noqa: F401

TRAIN_KWARGS = dict(multi_stream=4, fusion_level=4, shared_recon=True, recon_skip=True, MVAE_reduction=True,
                    final_sigmoid=True, f_maps=4, layer_order="ilc")      # train.py:142-143
