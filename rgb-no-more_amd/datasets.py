"""Mirror of the transform factory of the reference's `datasets.py` (get_transform, :305-385) for the DCT datasets.

    transform = get_transform('imagenet_dct', 'train', ops_list=cfg.TRAIN.AUGLIST, num_ops=2, ops_magnitude=3)
    Y, CbCr = transform((Y, CbCr))          # device int16 coefficients in, ToRange'd float tensors out

composes the device transform classes of `custom_transforms` exactly as the reference composes its CPU classes
(datasets.py:354-382), so `datasets.imagenet_dataset_indexing.__getitem__` (:274-297) needs an import swap only.  The
transforms accept one sample (C,H,W,8,8) -- the reference's calling convention -- or a whole batch (B,C,H,W,8,8);
`fused=True` returns the one-launch-pair batch transform (`TrainTransform_DCT` / `EvalTransform_DCT`) instead, which takes
the RAW quantised coefficients plus the quantisation tables and also does the de-quantisation of datasets.py:288-293.
"""
import torch

from . import custom_transforms as ctrans


def get_transform(dataset="imagenet_dct", type="train", ops_list=None, num_ops=2, ops_magnitude=10, dtype=torch.float32,
                  dtype_resize=torch.float32, fused=False):
    if dataset not in ("imagenet_dct", "imagenet_dct_swin"):
        raise NotImplementedError("rgb-no-more_amd implements the --domain DCT datasets ('imagenet_dct', 'imagenet_dct_swin')")
    size = 28 if dataset == "imagenet_dct" else 32
    if type == "train":
        if ops_list is None:          # one default for both forms (the fused class alone defaults to the JPEG-Ti config's list)
            ops_list = ctrans.DEFAULT_OPS
        if fused:
            return ctrans.TrainTransform_DCT(size=size, num_ops=num_ops, magnitude=ops_magnitude, num_magnitude_bins=11,
                                             ops_list=ops_list, out_dtype=dtype)
        return ctrans.Compose([
            ctrans.RandomResizedCrop_DCT(size, scale=(0.05, 1.0), ratio=(1, 1), dtype_resize=dtype_resize),
            ctrans.RandomFlip_DCT(p=0.5, direction="horizontal"),
            ctrans.RandAugment_dct(num_ops=num_ops, magnitude=ops_magnitude, num_magnitude_bins=11, ops_list=ops_list),
            ctrans.ToRange(val_min=-1, val_max=1, orig_min=-1024, orig_max=1016, dtype=dtype),
        ])
    if type in ("val", "test"):
        if fused:
            return ctrans.EvalTransform_DCT(size=size, out_dtype=dtype)
        first = (ctrans.ResizedCenterCrop_DCT(32, 28, dtype_resize=dtype_resize) if dataset == "imagenet_dct"
                 else ctrans.Resize_DCT(32))
        return ctrans.Compose([first, ctrans.ToRange(val_min=-1, val_max=1, orig_min=-1024, orig_max=1016, dtype=dtype)])
    print("Unrecognized dataset type! Returning 'None' transform")
    return None
