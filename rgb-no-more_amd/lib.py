"""ctypes binding of librgbnm.so (include/rgbnm.h).  There is NO fallback: if the HIP library is missing or
a call fails, an exception is raised (the product path never routes through a CPU/oracle implementation)."""
import ctypes as C
import os

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "librgbnm.so")

DT_F32, DT_BF16 = 0, 1
EPI_NONE, EPI_RES, EPI_GELU, EPI_POS, EPI_DGELU, EPI_TANH, EPI_DTANH = range(7)

_vp, _i, _f, _sz, _ll = C.c_void_p, C.c_int, C.c_float, C.c_size_t, C.c_longlong
# seconds the host spent blocked on the pinned-slot rings of the per-step records (augment parameters, mixup lambda): the host runs
# up to 16 steps ahead of the GPU and waits HERE; a value near zero over a timed loop means the host, not the GPU, paces the loop
HOST_WAIT = {"sec": 0.0}


class LinearDesc(C.Structure):
    _fields_ = [("w_off", _ll), ("b_off", _ll), ("ws_off", _ll), ("wst_off", _ll), ("bperm_off", _ll),
                ("N", _i), ("K", _i), ("perm_heads", _i), ("add_identity", _i), ("ldn", _i), ("pair", _i),
                ("chain_kind", _i), ("chain_off", _ll), ("bias_mode", _i), ("_pad2", _i), ("b2_off", _ll)]


class VitCfg(C.Structure):
    _fields_ = [("dtype", _i), ("B", _i), ("N", _i), ("E", _i), ("heads", _i), ("ln_eps", _f), ("attn_scale", _f)]


class BlockParams(C.Structure):
    _fields_ = [(n, _vp) for n in ("ln1_g", "ln1_b", "ln2_g", "ln2_b", "bqkv_perm", "bproj", "b1", "b2",
                                   "wqkv", "wqkv_t", "wproj", "wproj_t", "w1", "w1_t", "w2", "w2_t")]


class BlockActs(C.Structure):
    _fields_ = [(n, _vp) for n in ("x_in", "xn1", "mean1", "rstd1", "qkv", "lse", "attn", "x_mid", "xn2", "mean2",
                                   "rstd2", "u", "gl", "x_out")]


class BlockGrads(C.Structure):
    _fields_ = [(n, _vp) for n in ("dln1_g", "dln1_b", "dln2_g", "dln2_b", "dwqkv", "dbqkv", "dwproj", "dbproj",
                                   "dw1", "db1", "dw2", "db2")]


class BlockScratch(C.Structure):
    _fields_ = [("du", _vp), ("dxn", _vp), ("dx_mid", _vp), ("dattn", _vp), ("dqkv", _vp), ("ws", _vp),
                ("ws_bytes", _sz)]


class HeadParams(C.Structure):
    _fields_ = [(n, _vp) for n in ("ln_g", "ln_b", "b1", "b2", "w1", "w1_t", "w2", "w2_t")] + \
               [("n_classes", _i), ("_pad", _i)]


class HeadActs(C.Structure):
    _fields_ = [(n, _vp) for n in ("x", "mean", "rstd", "pooled", "h1", "logits")]


class HeadGrads(C.Structure):
    _fields_ = [(n, _vp) for n in ("dln_g", "dln_b", "dw1", "db1", "dw2", "db2")]


class CpbBlock(C.Structure):
    _fields_ = [(n, _vp) for n in ("w1", "b1", "w2", "ls", "bias", "scale", "dbias", "dscale", "dw1", "db1", "dw2", "dls")] + \
               [("heads", _i), ("_pad", _i)]


class ChainBlock(C.Structure):
    _fields_ = [(n, _vp) for n in ("wimg", "ln1_g", "ln1_b", "ln2_g", "ln2_b", "bqkv_perm", "bproj", "b1", "b2", "xn1", "mean1",
                                   "rstd1", "qkv", "lse", "attn", "x_mid", "xn2", "mean2", "rstd2", "u", "gl", "x_out")]


class ChainBwdBlock(C.Structure):
    _fields_ = [(n, _vp) for n in ("wimg", "ln1_g", "ln2_g", "x_in", "mean1", "rstd1", "qkv", "lse", "attn", "x_mid", "mean2",
                                   "rstd2", "u", "dy", "du", "dx_mid", "dqkv", "dx", "part2", "part1")]


ABI_VERSION = 3      # include/rgbnm.h RGBNM_ABI_VERSION this binding was written against

_P = C.POINTER
# name -> (restype, argtypes); every symbol include/rgbnm.h declares
PROTOTYPES = {
    "rgbnm_abi_version": (_i, []),
    "rgbnm_strerror": (C.c_char_p, [_i]),
    "rgbnm_set_option": (_i, [C.c_char_p, _i]),
    "rgbnm_gemm_tn_group_begin": (None, []),
    "rgbnm_gemm_tn_group_begin_n": (None, [_i]),
    "rgbnm_gemm_tn_group_begin_id": (None, [_i, C.c_ulonglong]),
    "rgbnm_gemm_tn_group_abort": (None, [C.c_ulonglong]),
    "rgbnm_gemm_tn_group_end": (_i, [_vp]),
    "rgbnm_get_option": (_i, [C.c_char_p]),
    "rgbnm_trace_collect": (_i, [_i, _vp, _vp, _vp, _vp]),
    "rgbnm_trace_reserve": (_i, [_i]),
    "rgbnm_gemm_nt": (_i, [_i, _i, _vp, _i, _vp, _i, _vp, _i, _vp, _vp, _i, _vp, _i, _vp, _i, _i, _i, _i, _i, _vp]),
    "rgbnm_gemm_tn_workspace": (_sz, [_i, _i, _i]),
    "rgbnm_gemm_tn_workspace_splits": (_sz, [_i, _i, _i]),
    "rgbnm_gemm_tn": (_i, [_i, _vp, _i, _vp, _i, _vp, _vp, _i, _i, _i, _i, _i, _vp, _sz, _vp]),
    "rgbnm_prep_weights": (_i, [_i, _vp, _i, _vp, _vp, _vp, _vp]),
    "rgbnm_prep_weights_chain": (_i, [_i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _vp]),
    "rgbnm_layernorm_fwd": (_i, [_i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _f, _vp]),
    "rgbnm_layernorm_bwd_workspace": (_sz, [_i, _i]),
    "rgbnm_layernorm_bwd": (_i, [_i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _sz, _vp]),
    "rgbnm_head_pool_fwd": (_i, [_i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _f, _vp]),
    "rgbnm_head_pool_bwd": (_i, [_i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _sz, _vp]),
    "rgbnm_attention_fwd": (_i, [_i, _vp, _vp, _vp, _i, _i, _i, _f, _vp]),
    "rgbnm_attention_bwd": (_i, [_i, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _f, _vp]),
    "rgbnm_subblock_embed": (_i, [_i, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "rgbnm_subblock_embed_mix": (_i, [_i, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "rgbnm_dct_augment_workspace": (_sz, [_i]),
    "rgbnm_dct_augment": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _sz,
                               _vp]),
    "rgbnm_dct_augment_workspace_ex": (_sz, [_i, _i]),
    "rgbnm_dct_augment_ex": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp,
                                  _sz, _vp]),
    "rgbnm_dct_augment_packed": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i,
                                      _i, _vp, _sz, _vp]),
    "rgbnm_softxent": (_i, [_i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _f, _vp]),
    "rgbnm_softxent_loss": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp]),
    "rgbnm_softxent_grad": (_i, [_i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _f, _vp]),
    "rgbnm_softxent_loss_mix": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp]),
    "rgbnm_softxent_grad_mix": (_i, [_i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _f, _vp]),
    "rgbnm_mixup": (_i, [_i, _i, _vp, _vp, _vp, _i, _ll, _vp]),
    "rgbnm_mixup_target": (_i, [_vp, _vp, _vp, _i, _i, _vp]),
    "rgbnm_clip_adamw_wd_workspace": (_sz, []),
    "rgbnm_clip_adamw_wd_step": (_i, [_vp, _vp, _vp, _vp, _vp, _ll, _f, _f, _f, _f, _i, _f, _f, _vp, _vp, _sz, _vp]),
    "rgbnm_vit_workspace": (_sz, [_P(VitCfg)]),
    "rgbnm_vit_workspace_ex": (_sz, [_P(VitCfg), _i]),
    "rgbnm_head_bwd_workspace": (_sz, [_vp, _i]),
    "rgbnm_gelu_table_init": (_i, [_vp]),
    "rgbnm_gelu_table_info": (_i, [_vp, _vp]),
    "rgbnm_reduce_hold_begin": (_i, []),
    "rgbnm_reduce_hold_end": (_i, [_vp, _vp, _sz, _vp]),
    "rgbnm_reduce_hold_cancel": (None, []),
    "rgbnm_reduce_hold_table_bytes": (_sz, []),
    "rgbnm_vit_block_fwd": (_i, [_P(VitCfg), _P(BlockParams), _P(BlockActs), _vp]),
    "rgbnm_vit_ln_chain": (_i, [_P(VitCfg)]),
    "rgbnm_swin_embed": (_i, [_i, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    "rgbnm_ln_generic_fwd": (_i, [_i, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _i, _i, _f, _vp]),
    "rgbnm_ln_generic_bwd_workspace": (_sz, [_i, _i]),
    "rgbnm_ln_generic_bwd": (_i, [_i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _i, _i, _i, _vp, _sz, _vp]),
    "rgbnm_window_attention_fwd": (_i, [_i, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "rgbnm_window_attention_bwd_workspace": (_sz, [_i, _i, _i]),
    "rgbnm_window_attention_bwd": (_i, [_i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _sz, _vp]),
    "rgbnm_swin_cpb_table_elems": (_sz, [_i]),
    "rgbnm_swin_cpb_fwd": (_i, [_vp, _i, _vp, _vp, _vp, _vp]),
    "rgbnm_swin_cpb_bwd": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _vp]),
    "rgbnm_merge_gather": (_i, [_i, _vp, _vp, _i, _i, _i, _i, _vp]),
    "rgbnm_token_mean": (_i, [_i, _vp, _vp, _i, _i, _i, _i, _vp]),
    "rgbnm_calib_mfma_bf16": (_i, [_i, _i, _vp, _vp]),
    "rgbnm_calib_stream": (_i, [_vp, _vp, _sz, _i, _i, _vp, _vp]),
    "rgbnm_calib_vmem_issue": (_i, [_i, _i, _i, _vp, _sz, _i, _vp, _vp]),
    "rgbnm_calib_l2": (_i, [_vp, _sz, _i, _i, _i, _i, _vp, _vp]),
    "rgbnm_calib_pipes": (_i, [_vp, _i, _i, _i, _vp, _vp, _vp]),
    "rgbnm_calib_occupy": (_i, [_vp, _sz, _i, _i, C.c_longlong, _i, _vp, _vp, _vp]),
    "rgbnm_calib_occupy_log": (_i, [_vp, _sz, _i, _i, C.c_longlong, _i, _vp, _vp, _vp, _vp]),
    "rgbnm_chain_block_bytes": (_sz, []),
    "rgbnm_chain_image_elems": (_ll, []),
    "rgbnm_chain_gather": (_i, [_vp, _vp, _vp, _ll, _vp]),
    "rgbnm_vit_chain_fwd": (_i, [_P(VitCfg), _P(ChainBlock), _i, _vp, _vp]),
    "rgbnm_chain_bwd_block_bytes": (_sz, []),
    "rgbnm_vit_chain_bwd": (_i, [_P(VitCfg), _P(ChainBwdBlock), _i, _vp, _vp]),
    "rgbnm_vit_block_bwd_dw": (_i, [_P(VitCfg), _P(BlockActs), _P(BlockGrads), _P(BlockScratch), _vp, _vp, _vp, _vp]),
    "rgbnm_vit_blocks_bwd_dw": (_i, [_P(VitCfg), _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "rgbnm_vit_blocks_bwd_dw_pe": (_i, [_P(VitCfg), _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "rgbnm_vit_block_fwd_chain": (_i, [_P(VitCfg), _P(BlockParams), _P(BlockActs), _i, _P(BlockParams), _P(BlockActs), _vp]),
    "rgbnm_vit_block_bwd": (_i, [_P(VitCfg), _P(BlockParams), _P(BlockActs), _P(BlockGrads), _P(BlockScratch), _vp,
                                 _vp, _vp]),
    "rgbnm_patch_embed_fwd": (_i, [_P(VitCfg), _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp]),
    "rgbnm_patch_embed_fwd_mix": (_i, [_P(VitCfg), _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp]),
    "rgbnm_patch_embed_bwd": (_i, [_P(VitCfg), _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "rgbnm_head_fwd": (_i, [_P(VitCfg), _P(HeadParams), _P(HeadActs), _vp]),
    "rgbnm_head_bwd": (_i, [_P(VitCfg), _P(HeadParams), _P(HeadActs), _P(HeadGrads), _vp, _vp, _vp, _vp, _vp, _sz,
                            _vp]),
}

_lib = None


class RgbnmError(RuntimeError):
    pass


def lib():
    """Load librgbnm.so (built by __graft_entry__.build() / rgb-no-more_amd/build.py).  Fails loudly."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RgbnmError(f"HIP extension missing: {LIB_PATH} (run `python -c 'import __graft_entry__ as g; g.build()'`). "
                             "There is no CPU fallback.")
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in PROTOTYPES.items():
            fn = getattr(L, name)       # AttributeError if the symbol is not exported
            fn.restype = res
            fn.argtypes = args
        if L.rgbnm_abi_version() != ABI_VERSION:
            raise RgbnmError(f"{LIB_PATH} has ABI version {L.rgbnm_abi_version()}, this binding expects {ABI_VERSION}: rebuild "
                             "(python rgb-no-more_amd/build.py)")
        if L.rgbnm_chain_block_bytes() != C.sizeof(ChainBlock) or L.rgbnm_chain_bwd_block_bytes() != C.sizeof(ChainBwdBlock):
            raise RgbnmError("rgbnm_chain_block layout mismatch between librgbnm.so and lib.py")
        _lib = L
    return _lib


def source_hash():
    """sha256 (first 16 hex digits) over the kernel sources the library is built from (csrc/*.hip, *.inc, *.h, reader.c and
    include/rgbnm.h, in name order): profiles/*.json files that hold counter figures measured in separate rocprofv3 passes carry the
    hash they were measured at, and bench.py marks them `stale` when it differs from the tree it runs from."""
    import hashlib
    h = hashlib.sha256()
    csrc = os.path.join(HERE, "csrc")
    files = sorted(os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith((".hip", ".inc", ".h", ".c")))
    files.append(os.path.join(os.path.dirname(HERE), "include", "rgbnm.h"))
    for f in files:
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def check(rc, what=""):
    if rc != 0:
        msg = lib().rgbnm_strerror(rc).decode()
        raise RgbnmError(f"{what}: rgbnm error {rc}: {msg}")


def dt_of(t):
    if t == torch.float32:
        return DT_F32
    if t == torch.bfloat16:
        return DT_BF16
    raise TypeError(f"unsupported dtype {t} (float32 or bfloat16)")


def ptr(t):
    return None if t is None else t.data_ptr()


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def stream():
    """Raw handle of torch's current stream on the current device.  Called once per C-ABI call (~150 per train step):
    torch._C._cuda_getCurrentRawStream is a plain C call, torch.cuda.current_stream() builds a Stream object (~10 us)."""
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


def require_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RgbnmError("rgb-no-more_amd kernels need device (HIP) tensors; there is no CPU fallback")
        if t is not None and not t.is_contiguous():
            raise RgbnmError("tensor must be contiguous")
