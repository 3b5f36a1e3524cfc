"""The reference's configs/vm_asr_*.yaml drop in unchanged (north_star): every shipped yaml loads through
vm_asr_amd.config.get_config and builds the models it names.  The yaml files are the reference's own and
are read from the reference checkout — present in the build container, absent on the GPU box (skipped)."""
import glob
import os

import pytest

REF_CONFIGS = "/root/reference/configs"
YAMLS = sorted(glob.glob(os.path.join(REF_CONFIGS, "vm_asr_*.yaml")))


@pytest.mark.skipif(not YAMLS, reason="reference checkout not present")
@pytest.mark.parametrize("path", YAMLS, ids=[os.path.basename(p)[:-5] for p in YAMLS])
def test_reference_yaml_loads_and_builds(path):
    import vm_asr_amd
    from vm_asr_amd.config import get_config
    cfg = get_config(path)
    name = os.path.basename(path)
    stft = cfg.DATA.STFT
    assert stft.N_FFT == (2048 if "nfft2048" in name else 1024) and stft.WIN_LENGTH == 1024   # config.py:55-57
    assert cfg.DATA.TARGET_SR == (16000 if name.startswith("vm_asr_16k") else 48000)
    # 512 frames per clip at every rate: hop = sr * 2.555 s / 511 (data_loader/data_loaders.py:482-513)
    assert stft.HOP_LENGTH == cfg.DATA.TARGET_SR // 200 and round(cfg.DATA.SEGMENT * cfg.DATA.TARGET_SR) == 511 * stft.HOP_LENGTH
    for tag, dims in (("VSSM8", 8), ("VSSM24", 24), ("VSSM32", 32)):
        if tag in name:
            assert cfg.MODEL.VSSM.DIMS == dims
    for tag, interact in (("M2P", "m2p"), ("P2M", "p2m"), ("SINGLE", "single")):
        if name.endswith(f"_{tag}.yaml"):
            assert cfg.MODEL.VSSM.INTERACT == interact
    models = vm_asr_amd.get_model(cfg)
    gen = models["generator"]
    assert gen is not None and gen.interact == cfg.MODEL.VSSM.INTERACT
    assert ("mpd" in models) == bool(cfg.TRAIN.ADVERSARIAL.ENABLE and "mpd" in cfg.TRAIN.ADVERSARIAL.DISCRIMINATORS)
    if cfg.MODEL.VSSM.DIMS == 16 and cfg.MODEL.VSSM.INTERACT == "dual":
        assert sum(p.numel() for p in gen.parameters()) == 3010352          # README.md:8
