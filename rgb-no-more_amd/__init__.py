"""rgb-no-more_amd: MI355X-native (gfx950) hot path of RGB-no-more -- DCT-domain augment + JPEG-ViT
forward/backward as hand-written HIP kernels behind a C-ABI library (include/rgbnm.h).

Host-side modules mirror the reference's operator surface for this path:
  dct_manip (read_coefficients), dct_ops / custom_transforms (DCT-domain augment),
  plainvit (ViT), cls_transforms (RandomMixup_DCT), custom_optims (WeightDecay, fused AdamW).
"""
from . import detfill  # noqa: F401
from . import lib, dct_ops, plainvit, cls_transforms, custom_optims  # noqa: F401,E402
from .plainvit import ViT  # noqa: F401,E402
from . import custom_transforms  # noqa: F401,E402
from . import dct_manip  # noqa: F401,E402
from . import parallel  # noqa: F401,E402
from . import swinv2  # noqa: F401,E402
from .swinv2 import SwinTransformerV2  # noqa: F401,E402
from . import eval  # noqa: F401,E402,A004
from . import datasets  # noqa: F401,E402
from . import loader  # noqa: F401,E402
from .loader import DCTBatchLoader  # noqa: F401,E402
