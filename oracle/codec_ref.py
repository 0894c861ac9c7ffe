"""Plain PyTorch fp32 twin of the codec stack (TEST INFRASTRUCTURE ONLY).

Exposes exactly the attributes the reference AudioTokenizer touches on its
`codec_model` (audio_tokenizer.py:26-36,158,189-200):
codebook_size, sample_rate, pad_audio, encoder, quantizer.inference,
quantizer.codebook.weight, quantizer.codebook_proj, decoder.

It is used to (a) cross-check that oracle/codec_oracle.c computes the same network
as a conventional nn.Module (fp tolerance; accumulation order differs), and (b) stand
in as a CPU `codec_model` object when driving the AudioTokenizer wrapper logic.
"""
from __future__ import annotations

from typing import Dict

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F


class _StridedConv(nn.Module):
    """Conv1d with explicit asymmetric zero padding padL=(k-s+1)//2, padR=(k-s)//2 and optional
    LeakyReLU pre-activation."""

    def __init__(self, cin, cout, k, s, pre, slope):
        super().__init__()
        self.conv = nn.Conv1d(cin, cout, k, stride=s, padding=0)
        self.padL, self.padR = (k - s + 1) // 2, (k - s) // 2
        self.pre, self.slope = pre, slope

    def forward(self, x):
        if self.pre:
            x = F.leaky_relu(x, self.slope)
        return self.conv(F.pad(x, (self.padL, self.padR)))


class _UpConv(nn.Module):
    def __init__(self, cin, cout, k, s, slope):
        super().__init__()
        p = (k - s + 1) // 2
        self.conv = nn.ConvTranspose1d(cin, cout, k, stride=s, padding=p, output_padding=s - k + 2 * p)
        self.slope = slope

    def forward(self, x):
        return self.conv(F.leaky_relu(x, self.slope))


class _Encoder(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.layers = nn.ModuleList(
            [_StridedConv(l["cin"], l["cout"], l["k"], l["s"], l["pre"], cfg.leaky_slope) for l in cfg.encoder_layers()]
        )

    def forward(self, x):  # [B,T] -> z_e [B,F,D]
        h = x.unsqueeze(1)
        for layer in self.layers:
            h = layer(h)
        return h.transpose(1, 2)


class _Quantizer(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.in_proj = nn.Linear(cfg.latent_dim, cfg.codebook_dim)
        self.codebook = nn.Embedding(cfg.codebook_size, cfg.codebook_raw_dim)
        self.codebook_proj = nn.Linear(cfg.codebook_raw_dim, cfg.codebook_dim)

    def inference(self, z_e):  # [B,F,D] -> (z_q [B,F,cd], idx [B,F])
        z = self.in_proj(z_e)
        cb = self.codebook_proj(self.codebook.weight)
        score = z @ cb.t() - 0.5 * (cb * cb).sum(-1)
        idx = score.argmax(-1)
        return F.embedding(idx, cb), idx


class _Decoder(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        layers = []
        for l in cfg.decoder_layers():
            if l["tr"]:
                layers.append(_UpConv(l["cin"], l["cout"], l["k"], l["s"], cfg.leaky_slope))
            else:
                layers.append(_StridedConv(l["cin"], l["cout"], l["k"], 1, l["pre"], cfg.leaky_slope))
        self.layers = nn.ModuleList(layers)

    def forward(self, z_q):  # [B,F,cd] -> [B,1,T]
        h = z_q.transpose(1, 2)
        for layer in self.layers:
            h = layer(h)
        return h.clamp(-1.0, 1.0)


class MagiCodecStyleRef(nn.Module):
    def __init__(self, cfg, weights: Dict[str, np.ndarray]):
        super().__init__()
        self.cfg = cfg
        self.codebook_size = cfg.codebook_size
        self.sample_rate = cfg.sample_rate
        self.hop = cfg.hop
        self.encoder = _Encoder(cfg)
        self.quantizer = _Quantizer(cfg)
        self.decoder = _Decoder(cfg)
        with torch.no_grad():
            for layer, spec in zip(self.encoder.layers, cfg.encoder_layers()):
                layer.conv.weight.copy_(torch.from_numpy(weights[spec["name"] + ".weight"]))
                layer.conv.bias.copy_(torch.from_numpy(weights[spec["name"] + ".bias"]))
            for layer, spec in zip(self.decoder.layers, cfg.decoder_layers()):
                layer.conv.weight.copy_(torch.from_numpy(weights[spec["name"] + ".weight"]))
                layer.conv.bias.copy_(torch.from_numpy(weights[spec["name"] + ".bias"]))
            q = self.quantizer
            q.in_proj.weight.copy_(torch.from_numpy(weights["quantizer.in_proj.weight"]))
            q.in_proj.bias.copy_(torch.from_numpy(weights["quantizer.in_proj.bias"]))
            q.codebook.weight.copy_(torch.from_numpy(weights["quantizer.codebook.weight"]))
            q.codebook_proj.weight.copy_(torch.from_numpy(weights["quantizer.codebook_proj.weight"]))
            q.codebook_proj.bias.copy_(torch.from_numpy(weights["quantizer.codebook_proj.bias"]))

    def pad_audio(self, x):  # [B,T] -> right-pad to a hop multiple
        T = x.shape[-1]
        pad = (-T) % self.hop
        return F.pad(x, (0, pad)) if pad else x
