"""vm_asr_amd — MI355X-native (gfx950) implementation of the VM-ASR data-parallel hot path.

The compute path is libvmasr_hip.so (hand-written HIP, C ABI in include/vmasr_hip.h);
this package is the PyTorch-ROCm host side that mirrors the reference's operator /
module interfaces (selective_scan_cuda_core.fwd/bwd, CrossScan/CrossMerge, SS2D,
VSSBlock, wav2spectro/spectro2wav, DualStreamInteractiveMambaUNet, Trainer).

There is NO CPU fallback in this package: ops raise if the HIP library is missing.
"""
__version__ = "0.1.0"

from . import hip_env  # noqa: F401,E402  (runtime settings, before anything touches the GPU)


def get_model(config):
    """Build the models a config asks for — same contract as the reference's
    model/__init__.py:8-66 (`{"generator": ..., "mpd": ...}`)."""
    from .discriminator import MultiPeriodDiscriminator
    from .model import DualStreamInteractiveMambaUNet

    models = {"generator": None}
    if config.MODEL.NAME == "DualStreamInteractiveMambaUNet":
        v = config.MODEL.VSSM
        models["generator"] = DualStreamInteractiveMambaUNet(
            in_chans=v.IN_CHANS, patch_size=v.PATCH_SIZE, depths=list(v.DEPTHS), dims=v.DIMS,
            ssm_d_state=v.SSM_D_STATE, ssm_ratio=v.SSM_RATIO,
            ssm_dt_rank=("auto" if v.SSM_DT_RANK == "auto" else int(v.SSM_DT_RANK)),
            ssm_act_layer=v.SSM_ACT_LAYER, ssm_conv=v.SSM_CONV, ssm_conv_bias=v.SSM_CONV_BIAS,
            ssm_drop_rate=v.SSM_DROP_RATE, ssm_init=v.SSM_INIT, forward_type=v.SSM_FORWARDTYPE,
            mlp_ratio=v.MLP_RATIO, mlp_act_layer=v.MLP_ACT_LAYER, mlp_drop_rate=v.MLP_DROP_RATE, gmlp=v.GMLP,
            drop_path_rate=v.DROP_PATH_RATE, patch_norm=v.PATCH_NORM, norm_layer=v.NORM_LAYER,
            patchembed_version=v.PATCHEMBED, downsample_version=v.DOWNSAMPLE, upsample_version=v.UPSAMPLE,
            output_version=v.OUTPUT, concat_skip=v.CONCAT_SKIP, interact=v.INTERACT,
            n_fft=config.DATA.STFT.N_FFT, hop_length=config.DATA.STFT.HOP_LENGTH,
            win_length=config.DATA.STFT.WIN_LENGTH, spectro_scale=config.DATA.STFT.SCALE,
            low_freq_replacement=config.TRAIN.LOW_FREQ_REPLACEMENT)
    if config.TRAIN.ADVERSARIAL.ENABLE:
        if "mpd" in config.TRAIN.ADVERSARIAL.DISCRIMINATORS:
            models["mpd"] = MultiPeriodDiscriminator(hidden=config.TRAIN.ADVERSARIAL.MPD_HIDDEN)
        if "msd" in config.TRAIN.ADVERSARIAL.DISCRIMINATORS:
            raise NotImplementedError("MSD is not enabled by any shipped yaml and is not built")
    return models
