"""Codec model description: architecture config, tensor names and seeded random-init weights.

The reference loads a third-party MagiCodec checkpoint by name
(audio_tokenizer.py:26-28); neither the package nor weights exist offline, so this
build defines its own "MagiCodec-style" stack: a strided
conv1d encoder (hop 320 = 2*4*5*8 at 16 kHz -> 50 Hz), ONE 131072-entry codebook
of dimension 16 behind a `codebook_proj` linear (audio_tokenizer.py:158,198;
codec_llama.py:18-19), and a mirrored transposed-conv decoder.

Tensor names are the contract between this file, include/rca.h and oracle/.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Tuple

import numpy as np


@dataclass
class CodecConfig:
    sample_rate: int = 16000
    strides: Tuple[int, ...] = (2, 4, 5, 8)
    channels: Tuple[int, ...] = (32, 64, 128, 256, 512)
    k_in: int = 7
    k_latent: int = 3
    latent_dim: int = 256
    codebook_size: int = 131072
    codebook_raw_dim: int = 32
    codebook_dim: int = 16
    leaky_slope: float = 0.1
    name: str = "MagiCodec-50Hz-Base"

    def __post_init__(self):
        if len(self.channels) != len(self.strides) + 1:
            raise ValueError("channels must have len(strides)+1 entries")
        if self.k_in % 2 != 1 or self.k_latent % 2 != 1:
            raise ValueError("k_in and k_latent must be odd")

    @property
    def n_stages(self) -> int:
        return len(self.strides)

    @property
    def hop(self) -> int:
        return int(np.prod(self.strides))

    @property
    def framerate(self) -> float:
        return self.sample_rate / self.hop

    def encoder_layers(self) -> List[dict]:
        """(name, Cin, Cout, k, stride, pre_act) for every encoder conv, in order."""
        out = [dict(name="enc.conv_in", cin=1, cout=self.channels[0], k=self.k_in, s=1, pre=False)]
        for i, s in enumerate(self.strides):
            out.append(dict(name=f"enc.down.{i}", cin=self.channels[i], cout=self.channels[i + 1], k=2 * s, s=s, pre=True))
        out.append(dict(name="enc.conv_out", cin=self.channels[-1], cout=self.latent_dim, k=self.k_latent, s=1, pre=True))
        return out

    def decoder_layers(self) -> List[dict]:
        n = self.n_stages
        out = [dict(name="dec.conv_in", cin=self.codebook_dim, cout=self.channels[n], k=self.k_latent, s=1, pre=False, tr=False)]
        for i in range(n):
            s = self.strides[n - 1 - i]
            out.append(dict(name=f"dec.up.{i}", cin=self.channels[n - i], cout=self.channels[n - 1 - i], k=2 * s, s=s, pre=True, tr=True))
        out.append(dict(name="dec.conv_out", cin=self.channels[0], cout=1, k=self.k_in, s=1, pre=True, tr=False))
        return out

    def receptive_field(self) -> Tuple[int, int]:
        """(encoder, decoder) whole frames a kept code / a kept sample can see to its LEFT, from the layer geometry
        (padL = (k - s + 1) // 2 everywhere).  Mirrors csrc/rca_codec.hip::codec_receptive_field: the streaming tail
        entry points recompute only this margin plus the kept frames."""
        lo = 0
        for layer in reversed(self.encoder_layers()):
            lo = lo * layer["s"] - (layer["k"] - layer["s"] + 1) // 2
        enc_left = -(lo // self.hop)          # ceil(-lo / hop)
        u = 0
        for layer in reversed(self.decoder_layers()):
            k, s = layer["k"], layer["s"]
            pad = (k - s + 1) // 2
            u = -((-(u + pad - (k - 1))) // s) if layer["tr"] else u - pad     # ceil division / plain shift
        return enc_left, -u

    def encoder_flops_per_sample(self) -> float:
        """Multiply-add FLOPs (2 per MAC) of the encoder stack per input sample."""
        rate = 1.0
        total = 0.0
        for layer in self.encoder_layers():
            rate /= layer["s"]
            total += 2.0 * layer["cin"] * layer["k"] * layer["cout"] * rate
        return total

    def vq_flops_per_frame(self) -> float:
        return 2.0 * self.codebook_size * self.codebook_dim + 2.0 * self.latent_dim * self.codebook_dim


def tiny_codec_config(**kw) -> CodecConfig:
    """A small config with the same structure (hop 320) for fast CPU tests."""
    base = dict(channels=(4, 8, 8, 16, 16), latent_dim=16, codebook_size=1024, codebook_raw_dim=8,
                codebook_dim=16, name="tiny")
    base.update(kw)
    return CodecConfig(**base)


def init_codec_weights(cfg: CodecConfig, seed: int = 0) -> Dict[str, np.ndarray]:
    """Seeded random weights (float32).  Fan-in scaled so activations keep O(1) scale and
    the latent z and the projected codebook have comparable spread (diverse codes)."""
    rng = np.random.default_rng(seed)
    w: Dict[str, np.ndarray] = {}

    def conv(name, cout, cin, k, gain):
        std = gain / np.sqrt(cin * k)
        w[f"{name}.weight"] = (rng.standard_normal((cout, cin, k)) * std).astype(np.float32)
        w[f"{name}.bias"] = (rng.standard_normal((cout,)) * 0.02).astype(np.float32)

    def convtr(name, cin, cout, k, s, gain):
        # each output sees k/s taps per input channel
        std = gain / np.sqrt(cin * (k // s))
        w[f"{name}.weight"] = (rng.standard_normal((cin, cout, k)) * std).astype(np.float32)
        w[f"{name}.bias"] = (rng.standard_normal((cout,)) * 0.02).astype(np.float32)

    g = 1.35  # ~ compensates LeakyReLU(0.1) variance loss
    for layer in cfg.encoder_layers():
        # raw PCM is ~0.1-0.2 rms: lift conv_in so the stack runs at O(1)
        conv(layer["name"], layer["cout"], layer["cin"], layer["k"], 6.0 if layer["name"] == "enc.conv_in" else g)
    cd, raw, D, N = cfg.codebook_dim, cfg.codebook_raw_dim, cfg.latent_dim, cfg.codebook_size
    w["quantizer.in_proj.weight"] = (rng.standard_normal((cd, D)) / np.sqrt(D)).astype(np.float32)
    w["quantizer.in_proj.bias"] = np.zeros((cd,), np.float32)
    w["quantizer.codebook.weight"] = rng.standard_normal((N, raw)).astype(np.float32)
    w["quantizer.codebook_proj.weight"] = (rng.standard_normal((cd, raw)) / np.sqrt(raw)).astype(np.float32)
    w["quantizer.codebook_proj.bias"] = np.zeros((cd,), np.float32)
    for layer in cfg.decoder_layers():
        if layer["tr"]:
            convtr(layer["name"], layer["cin"], layer["cout"], layer["k"], layer["s"], g)
        else:
            conv(layer["name"], layer["cout"], layer["cin"], layer["k"], 0.25 if layer["name"] == "dec.conv_out" else g)
    return w


def codec_tensor_names(cfg: CodecConfig) -> List[str]:
    names = []
    for layer in cfg.encoder_layers():
        names += [f"{layer['name']}.weight", f"{layer['name']}.bias"]
    names += ["quantizer.in_proj.weight", "quantizer.in_proj.bias", "quantizer.codebook.weight",
              "quantizer.codebook_proj.weight", "quantizer.codebook_proj.bias"]
    for layer in cfg.decoder_layers():
        names += [f"{layer['name']}.weight", f"{layer['name']}.bias"]
    return names
