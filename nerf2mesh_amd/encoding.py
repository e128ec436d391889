"""Encoder factory with the reference's names (encoding.py:71-106): None / frequency_torch / frequency / sh / hashgrid / tiledgrid."""
import torch
import torch.nn as nn


class FreqEncoder_torch(nn.Module):
    """sin/cos frequency encoding in plain torch (encoding.py:8-46)."""

    def __init__(self, input_dim, max_freq_log2, N_freqs, log_sampling=True, include_input=True):
        super().__init__()
        self.input_dim = input_dim
        self.include_input = include_input
        self.output_dim = (input_dim if include_input else 0) + input_dim * N_freqs * 2
        bands = 2.0 ** torch.linspace(0.0, max_freq_log2, N_freqs) if log_sampling else torch.linspace(1.0, 2.0 ** max_freq_log2, N_freqs)
        self.freq_bands = bands.tolist()

    def forward(self, x, **kwargs):
        parts = [x] if self.include_input else []
        for f in self.freq_bands:
            parts += [torch.sin(x * f), torch.cos(x * f)]
        return torch.cat(parts, dim=-1)


def get_encoder(encoding, input_dim=3, multires=6, degree=4, num_levels=16, level_dim=2, base_resolution=16,
                log2_hashmap_size=19, desired_resolution=2048, align_corners=False, interpolation="linear", **kwargs):
    if encoding == "None":
        return (lambda x, **kw: x), input_dim
    if encoding == "frequency_torch":
        enc = FreqEncoder_torch(input_dim, multires - 1, multires)
    elif encoding == "sh":
        from .shencoder import SHEncoder
        enc = SHEncoder(input_dim=input_dim, degree=degree)
    elif encoding in ("hashgrid", "tiledgrid"):
        from .gridencoder import GridEncoder
        enc = GridEncoder(input_dim=input_dim, num_levels=num_levels, level_dim=level_dim, base_resolution=base_resolution,
                          log2_hashmap_size=log2_hashmap_size, desired_resolution=desired_resolution,
                          gridtype="hash" if encoding == "hashgrid" else "tiled", align_corners=align_corners,
                          interpolation=interpolation)
    elif encoding == "frequency":
        from .freqencoder import FreqEncoder
        enc = FreqEncoder(input_dim=input_dim, degree=multires)                # encoding.py:84-86
    elif encoding == "hashgrid_tcnn":
        raise NotImplementedError("encoding 'hashgrid_tcnn' is outside this build's scope (SURVEY.md section 2, row 24: tiny-cuda-nn)")
    else:
        raise NotImplementedError("Unknown encoding mode, choose from [None, frequency_torch, sh, hashgrid, tiledgrid]")
    return enc, enc.output_dim
