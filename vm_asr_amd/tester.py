"""Evaluation harness: the reference's Tester on the HIP path (trainer/tester.py:15-240, base/base_tester.py:10-90,
utils/post_processing.py:4-34).

    Tester(models, metric_ftns, config, device, data_loader, logger).evaluate() -> dict

Same flow: generator in eval mode, batch-1 clips `(wave_in, wave_tgt, highcut, name, pad)`; clips longer than one
training segment are cut into overlapping segments (TEST.OVERLAP samples), enhanced one by one and cross-averaged
back (`unfold_audio` / `fold_audio`); metrics = config.TRAIN.METRICS (SNR, LSD, LSD-HF, LSD-LF on the HIP STFT) plus
RTF = processing time / clip duration; the summary row is appended to `results_{16,48}kHz.csv` with the reference's
columns (SAMPLE_RATE, SNR, LSD, LSD_HF, LSD_LF, RTF, RTF_RECIPROCAL).  Differences: the processing time is taken
with a device synchronise on both sides (the reference reads the host clock without one, so its RTF under-reports);
wav files are written with the standard library (`wave`, 16-bit PCM) because torchaudio is not a dependency here.
The checkpoint is the generator's `checkpoint-best-G.pth` (utils/utils.py:153-176), read through the yacs-compatible
loader of trainer.BaseTrainer.
"""
import csv
import os
import time
import wave

import torch

from .trainer import _Logger, unwrap

__all__ = ["unfold_audio", "fold_audio", "BaseTester", "Tester"]


def unfold_audio(audio, segment_length, overlap):
    """(B, C, T) -> (B, C, n_seg, segment_length) overlapping segments (utils/post_processing.py:4-9)."""
    return audio.unfold(dimension=-1, size=segment_length, step=segment_length - overlap)


def fold_audio(segments, total_length, segment_length, overlap):
    """Inverse of unfold_audio with the overlaps averaged; samples no segment covers stay 0
    (utils/post_processing.py:12-34)."""
    step = segment_length - overlap
    B, C, n, _ = segments.shape
    out = torch.zeros(B, C, total_length, dtype=segments.dtype, device=segments.device)
    cnt = torch.zeros(B, C, total_length, dtype=segments.dtype, device=segments.device)
    for i in range(n):
        out[:, :, i * step:i * step + segment_length] += segments[:, :, i]
        cnt[:, :, i * step:i * step + segment_length] += 1
    return out / cnt.clamp(min=1)


class BaseTester:
    def __init__(self, models, metric_ftns, config, logger=None):
        self.config, self.logger = config, logger or _Logger()
        self.models, self.metric_ftns = models, metric_ftns
        tag = str(config.TAG).split("_")
        if len(tag) != 2:
            raise ValueError("config.TAG must be '{input_sr}_{target_sr}' in evaluation mode (main.py:244-248)")
        self.input_sr, self.target_sr = int(tag[0]), int(tag[1])
        self.output_dir = config.OUTPUT
        if config.MODEL.RESUME_PATH is not None:
            self._resume_checkpoint()

    def _resume_checkpoint(self):
        from .config import yacs_pickle_compat
        path = self.config.MODEL.RESUME_PATH
        for kind in ("best", "latest"):
            f = os.path.join(path, f"checkpoint-{kind}-G.pth")
            if os.path.exists(f):
                with yacs_pickle_compat():
                    ck = torch.load(f, map_location="cpu", weights_only=False)
                unwrap(self.models["generator"]).load_state_dict(ck["state_dict"], strict=True)
                self.logger.info(f"Loaded generator from {f}")
                break
        else:
            raise FileNotFoundError(f"No generator checkpoint found in {path}")
        if self.target_sr != self.config.DATA.TARGET_SR:
            raise ValueError(f"Target sampling rate mismatch: {self.target_sr} vs {self.config.DATA.TARGET_SR}")


class Tester(BaseTester):
    def __init__(self, models, metric_ftns, config, device, data_loader, logger=None):
        super().__init__(models, metric_ftns, config, logger)
        self.device = device[0] if isinstance(device, (tuple, list)) else device
        self.test_loader = data_loader
        self.test_log = {}
        self.num_frames_per_seg = int(int(config.DATA.SEGMENT * config.DATA.FLAC2WAV.SRC_SR) * self.target_sr
                                      / config.DATA.FLAC2WAV.SRC_SR)
        for k, m in self.models.items():
            self.models[k] = m.to(self.device)

    def _sync(self):
        if self.device.type == "cuda":
            torch.cuda.synchronize(self.device)

    def _enhance(self, wave_input, highcut):
        gen = self.models["generator"]
        if wave_input.size(2) <= self.num_frames_per_seg:
            return gen(wave_input, highcut)
        segs = unfold_audio(wave_input, self.num_frames_per_seg, self.config.TEST.OVERLAP)
        done = torch.zeros_like(segs)
        for i in range(segs.size(2)):
            done[:, :, i] = gen(segs[:, :, i].contiguous(), highcut)
        return fold_audio(done, wave_input.size(2), self.num_frames_per_seg, self.config.TEST.OVERLAP)

    @torch.no_grad()
    def evaluate(self):
        for m in self.models.values():
            m.eval()
        sums, n = {}, 0
        for wave_input, wave_target, highcut, filename, pad_length in self.test_loader:
            wave_input, wave_target = wave_input.to(self.device), wave_target.to(self.device)
            self._sync()
            t0 = time.time()
            wave_out = self._enhance(wave_input, highcut)
            self._sync()
            run_time = time.time() - t0
            pad = int(pad_length[0]) if hasattr(pad_length, "__getitem__") else int(pad_length)
            rtf = run_time / ((wave_input.size(2) - pad) / self.config.DATA.TARGET_SR)
            vals = self._evaluate_batch(wave_out, wave_target, highcut)
            vals["rtf"], vals["rtf_reciprocal"] = rtf, 1.0 / rtf
            for k, v in vals.items():
                sums[k] = sums.get(k, 0.0) + float(v)
            n += 1
            if self.config.TEST.SAVE_RESULT:
                self._save_wavs(wave_input, wave_out, wave_target, filename, pad)
        self.test_log = {k: v / max(1, n) for k, v in sums.items()}
        self.test_log["sample_rate"] = self.input_sr
        self.logger.info("Summary of Evaluation: " + " | ".join(f"{k}={v:.4f}" for k, v in self.test_log.items()))
        self.save_results_to_csv(self.test_log, "results_16kHz.csv" if self.target_sr == 16000 else "results_48kHz.csv")
        return self.test_log

    def _evaluate_batch(self, wave_out, wave_target, highcut):
        return {m.__name__: m(wave_out.float().squeeze(1), wave_target.squeeze(1), hf=highcut) for m in self.metric_ftns}

    def _save_wavs(self, wave_input, wave_out, wave_target, filename, pad):
        os.makedirs(self.output_dir, exist_ok=True)
        stem = str(filename[0] if isinstance(filename, (list, tuple)) else filename).replace(".wav", "")
        keep = wave_input.size(2) - pad
        for tag, w in (("up", wave_out), ("orig", wave_target), ("down", wave_input)):
            pcm = (w[0, 0, :keep].float().clamp(-1, 1) * 32767.0).round().to(torch.int16).cpu().numpy()
            with wave.open(os.path.join(self.output_dir, f"{stem}_{tag}.wav"), "wb") as f:
                f.setnchannels(1)
                f.setsampwidth(2)
                f.setframerate(int(self.config.DATA.TARGET_SR))
                f.writeframes(pcm.tobytes())

    def save_results_to_csv(self, results, filename="results.csv"):
        """One row per evaluation, the reference's column order (trainer/tester.py:221-240)."""
        order = ["sample_rate", "snr", "lsd", "lsd_hf", "lsd_lf", "rtf", "rtf_reciprocal"]
        row = {k: results.get(k, "") for k in order}
        exists = os.path.isfile(filename)
        with open(filename, mode="a", newline="") as f:
            w = csv.writer(f)
            if not exists:
                w.writerow([k.upper() for k in row])
            w.writerow(row.values())
